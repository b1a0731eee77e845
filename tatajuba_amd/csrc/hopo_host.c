#define _GNU_SOURCE
/* hopo_host.c -- C host side of the drop-in boundary (include/tatajuba_hopo.h).
 *
 * Same function names, argument meaning and error behaviour as tatajuba's src/hopo_counter.c for the per-read scan and
 * the per-sample finalise, but the work is done by the HIP kernels behind the tjamd_* C-ABI (hopo_device.hip).  There
 * is NO CPU implementation of the scan or of the sort/reduce in this library: without an MI355X every entry that
 * needs results prints an error and exits, like the reference's biomcmc_error().
 */
#include "../../include/tatajuba_amd.h"
#include <pthread.h>
#include <sched.h>
#include <sys/stat.h>
#include <unistd.h>
#include "fastq_reader.h"
#include "feeder.h"
#include "tj_inflate.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <limits.h>

void tj_set_last_error (const char *msg);            /* hopo_device.hip: what tjamd_last_error () returns */

/* ---- tables (reference: src/hopo_counter.c:10-11,205-216; constant here, so no lazy-initialisation race) -------- */

#define TJ_OTHER {4, 4}
uint8_t dna_in_2_bits[256][2] = {
  [0 ... 255] = TJ_OTHER,
  ['A'] = {0, 3}, ['a'] = {0, 3}, ['C'] = {1, 2}, ['c'] = {1, 2}, ['G'] = {2, 1}, ['g'] = {2, 1},
  ['T'] = {3, 0}, ['t'] = {3, 0}, ['U'] = {3, 0}, ['u'] = {3, 0}
};
char bit_2_dna[] = {'A', 'C', 'G', 'T'};

/* ---- private state appended to the public struct ------------------------------------------------------------------ */

#define TJ_PRIV_MAGIC 0x746a616d64686f70ULL
#define TJ_BATCH_BYTES (64u << 20)

typedef struct
{
  uint64_t magic;
  tjamd_counter *dev;
  int n_host;            /* records in hc->elem[] that were produced by update_hopo_counter_from_seq */
  long n_device;         /* raw records held on the device */
} tj_private;

static inline tj_private *tj_priv (hopo_counter hc) { return (tj_private *) (hc + 1); }

static void
tj_fatal (const char *fmt, ...)
{ /* biomcmc_error(): message, then exit(EXIT_FAILURE) */
  va_list ap;
  fprintf (stderr, "tatajuba_amd error: ");
  va_start (ap, fmt); vfprintf (stderr, fmt, ap); va_end (ap);
  fprintf (stderr, "\n");
  exit (EXIT_FAILURE);
}

/* hc->n_elem is an int (reference src/hopo_counter.h:54): a sample with more raw tracts than that cannot be stated in
 * the drop-in struct -- the reference would run into undefined behaviour there; stop instead of wrapping quietly */
static int
tj_n_elem (long n_host, long n_device, const char *name)
{
  if (n_host + n_device > (long) INT_MAX)
    tj_fatal ("sample %s holds %ld homopolymer tracts, more than hopo_counter's int n_elem can state (%d)", name ? name : "?", n_host + n_device, INT_MAX);
  return (int) (n_host + n_device);
}

static void
tj_warning (const char *fmt, ...)
{
  va_list ap;
  fprintf (stderr, "tatajuba_amd warning: ");
  va_start (ap, fmt); vfprintf (stderr, fmt, ap); va_end (ap);
  fprintf (stderr, "\n");
}

static int tj_next_device = 0;

/* sample i -> device i mod N (the reference runs one OpenMP thread per sample: src/genome_set.c:66-94);
 * TATAJUBA_AMD_DEVICE pins every counter of the process to one device (one process per GPU under torchrun). */
static tjamd_counter *
tj_device_counter (hopo_counter hc)
{
  tj_private *pv = tj_priv (hc);
  if (!pv->dev) {
    int n = tjamd_device_count (), dev;
    const char *pin = getenv ("TATAJUBA_AMD_DEVICE");
    if (n <= 0) tj_fatal ("no HIP device is visible; the homopolymer scan runs on an MI355X only (there is no CPU fallback)");
    dev = pin ? atoi (pin) : (__atomic_fetch_add (&tj_next_device, 1, __ATOMIC_RELAXED) % n);
    pv->dev = tjamd_counter_create (dev, hc->kmer_size);
    if (!pv->dev) tj_fatal ("%s", tjamd_last_error ());
  }
  return pv->dev;
}

/* (context_host.c: the counter's device side, NULL if it never had one) */
tjamd_counter *tj_counter_device (hopo_counter hc) { return hc ? tj_priv (hc)->dev : NULL; }

/* ---- constructor / destructor (reference: src/hopo_counter.c:159-186) ---------------------------------------------- */

hopo_counter
new_hopo_counter (int kmer_size)
{
  hopo_counter hc = (hopo_counter) calloc (1, sizeof (struct hopo_counter_struct) + sizeof (tj_private));
  if (!hc) tj_fatal ("out of memory");
  hc->n_alloc = 32;
  hc->kmer_size = kmer_size;
  hc->n_idx = hc->n_elem = 0;
  hc->coverage = 0;
  hc->elem = (hopo_element *) malloc ((size_t) hc->n_alloc * sizeof (hopo_element));
  hc->ref_counter = 1;
  hc->name = NULL; hc->idx_initial = hc->idx_final = NULL;
  tj_priv (hc)->magic = TJ_PRIV_MAGIC;
  return hc;
}

void
del_hopo_counter (hopo_counter hc)
{
  if (!hc) return;
  if (--hc->ref_counter) return;
  if (tj_priv (hc)->magic == TJ_PRIV_MAGIC && tj_priv (hc)->dev) tjamd_counter_destroy (tj_priv (hc)->dev);
  free (hc->elem); free (hc->name); free (hc->idx_initial); free (hc->idx_final);
  free (hc);
}

/* ---- multi-threaded feeder for plain files (feeder.c): its sink is the device scan ---------------------------------- */

typedef struct { tjamd_counter *dev; int min_tract_size; } tj_gpu_sink;

/* Pinned batch buffers are kept for the next file: pinning and unpinning half a gigabyte costs as much as parsing it.
 * A small process-wide pool (the caller's sample threads share it); at most TJ_PIN_POOL_BYTES stay cached. */
#define TJ_PIN_POOL_SLOTS 128
#define TJ_PIN_POOL_BYTES (2l << 30)
static struct { void *p; size_t bytes; int in_use; } tj_pin_pool[TJ_PIN_POOL_SLOTS];
static pthread_mutex_t tj_pin_lock = PTHREAD_MUTEX_INITIALIZER;

static void *
tj_pinned_get (size_t bytes)
{
  int i;
  void *p = NULL;
  pthread_mutex_lock (&tj_pin_lock);
  for (i = 0; i < TJ_PIN_POOL_SLOTS; i++)
    if (tj_pin_pool[i].p && !tj_pin_pool[i].in_use && tj_pin_pool[i].bytes >= bytes && tj_pin_pool[i].bytes <= 2 * bytes) {
      tj_pin_pool[i].in_use = 1; p = tj_pin_pool[i].p; break;
    }
  pthread_mutex_unlock (&tj_pin_lock);
  if (p) return p;
  p = tjamd_host_alloc (bytes);
  if (!p) return NULL;
  pthread_mutex_lock (&tj_pin_lock);
  for (i = 0; i < TJ_PIN_POOL_SLOTS; i++) if (!tj_pin_pool[i].p) { tj_pin_pool[i].p = p; tj_pin_pool[i].bytes = bytes; tj_pin_pool[i].in_use = 1; break; }
  pthread_mutex_unlock (&tj_pin_lock);                  /* (no free slot: the buffer is simply not pooled) */
  return p;
}

static void
tj_pinned_put (void *p)
{
  int i, found = 0;
  size_t cached = 0;
  if (!p) return;
  pthread_mutex_lock (&tj_pin_lock);
  for (i = 0; i < TJ_PIN_POOL_SLOTS; i++) if (tj_pin_pool[i].p && !tj_pin_pool[i].in_use) cached += tj_pin_pool[i].bytes;
  for (i = 0; i < TJ_PIN_POOL_SLOTS; i++)
    if (tj_pin_pool[i].p == p) {
      found = 1;
      if (cached + tj_pin_pool[i].bytes <= (size_t) TJ_PIN_POOL_BYTES) { tj_pin_pool[i].in_use = 0; p = NULL; }
      else { tj_pin_pool[i].p = NULL; tj_pin_pool[i].bytes = 0; tj_pin_pool[i].in_use = 0; }
      break;
    }
  pthread_mutex_unlock (&tj_pin_lock);
  (void) found;
  if (p) tjamd_host_free (p);
}

static void *tj_sink_alloc (void *ctx, size_t bytes) { (void) ctx; return tj_pinned_get (bytes); }
static void tj_sink_release (void *ctx, void *p) { (void) ctx; tj_pinned_put (p); }
static int tj_sink_put (void *ctx, const unsigned char *stream, size_t n_bytes, long n_reads)
{
  tj_gpu_sink *g = (tj_gpu_sink *) ctx;
  (void) n_reads;
  return tjamd_scan_host (g->dev, stream, n_bytes, g->min_tract_size);
}
/* (tjamd_raw_count rather than tjamd_sync: it also brings the exact record counts back, so the device storage is sized
 * for what the batches really produced, not for the sum of their worst cases) */
static int tj_sink_sync (void *ctx) { return tjamd_raw_count (((tj_gpu_sink *) ctx)->dev) < 0; }
static long tj_sink_mark (void *ctx) { return tjamd_mark (((tj_gpu_sink *) ctx)->dev); }
static int tj_sink_wait (void *ctx, long mark) { return tjamd_wait_mark (((tj_gpu_sink *) ctx)->dev, (int) mark); }

/* CPUs this process may use: online processors, cut down to the affinity mask and to a cgroup CPU quota if there is one
 * (a container given 16 of a host's 256 cores reports 256 online) */
static long
tj_cpu_budget (void)
{
  long n = sysconf (_SC_NPROCESSORS_ONLN);
  cpu_set_t set;
  FILE *f;
  if (sched_getaffinity (0, sizeof set, &set) == 0) { const long a = CPU_COUNT (&set); if (a > 0 && a < n) n = a; }
  f = fopen ("/sys/fs/cgroup/cpu.max", "r");            /* cgroup v2: "<quota> <period>" or "max <period>" */
  if (f) {
    long quota = 0, period = 0;
    if (fscanf (f, "%ld %ld", &quota, &period) == 2 && quota > 0 && period > 0) { const long q = (quota + period - 1) / period; if (q < n) n = q; }
    fclose (f);
  }
  return n < 1 ? 1 : n;
}

static int tj_files_in_flight;                          /* new_or_append_hopo_counter_from_file calls that are parsing right now */

/* Feeder threads of one file: the CPUs of the process shared among the files being read at this moment (the reference's
 * caller runs one thread per sample, src/genome_set.c:66-68: eight samples at once get an eighth each, one sample alone
 * gets them all), at most TJF_MAX_THREADS.  TATAJUBA_AMD_FEEDER_THREADS overrides. */
static int
tj_feeder_threads (const tatajuba_options_t *opt)
{
  const char *e = getenv ("TATAJUBA_AMD_FEEDER_THREADS");
  long n;
  if (e) n = atol (e);
  else {
    /* The files being read now are a lower bound only: the reference's caller starts its per-sample threads together, and
     * the first file to get here would see itself alone and take the whole budget, the second half of it, ...  What the
     * caller says it will run at once -- min (n_samples, n_threads), the clipping of src/main.c:176-181 -- is the share
     * to plan for from the first file on. */
    long active = __atomic_load_n (&tj_files_in_flight, __ATOMIC_RELAXED), planned = 1;
    if (opt && opt->n_samples > 0 && opt->n_threads > 0) planned = opt->n_samples < opt->n_threads ? opt->n_samples : opt->n_threads;
    if (active < planned) active = planned;
    if (active < 1) active = 1;
    n = tj_cpu_budget () / active;
  }
  if (n < 1) n = 1;
  if (n > TJF_MAX_THREADS) n = TJF_MAX_THREADS;
  return (int) n;
}

#define TJ_FEEDER_MIN_BYTES (32l << 20)                  /* smaller files: one reader is as fast */
#define TJ_FEEDER_WINDOW    (128l << 20)
#define TJ_FEEDER_MIN_GZ_BYTES (4l << 20)                /* gzip: the inflate runs ahead of the parse from this size on */
#define TJ_FEEDER_GZ_WINDOW (64l << 20)                  /* inflated bytes per view */

/* ---- file -> device (reference: src/hopo_counter.c:135-157) -------------------------------------------------------- */

hopo_counter
new_or_append_hopo_counter_from_file (hopo_counter hc, const char *filename, tatajuba_options_t opt)
{
  hopo_counter h = hc;
  tjamd_counter *dev;
  tjr_reader *rd;
  unsigned char *buf[2] = {NULL, NULL};
  size_t fill = 0;
  int cur = 0, in_flight = 0;
  long len, n;
  const char *seq;

  if (!h) {
    h = new_hopo_counter (opt.kmer_size);
    h->name = (char *) malloc (strlen (filename) + 1);
    strcpy (h->name, filename);
    h->opt = opt;
  }
  if (h->idx_initial) tj_fatal ("This counter has been compared to another; cannot add more reads to it"); /* reference :152 */
  {                                                     /* big file: several readers / inflaters (feeder.c), same output */
    struct stat st;
    int threads;
    (void) __atomic_add_fetch (&tj_files_in_flight, 1, __ATOMIC_RELAXED);
    threads = tj_feeder_threads (&opt);
    const int plain = threads > 1 ? tjf_is_plain_file (filename) : -1;
    if (plain >= 0 && stat (filename, &st) == 0 && st.st_size >= (plain ? TJ_FEEDER_MIN_BYTES : TJ_FEEDER_MIN_GZ_BYTES)) {
      tj_gpu_sink g;
      tjf_sink sk;
      long got;
      g.dev = tj_device_counter (h); g.min_tract_size = opt.min_tract_size;
      sk.ctx = &g; sk.alloc = tj_sink_alloc; sk.release = tj_sink_release; sk.put = tj_sink_put; sk.sync = tj_sink_sync;
      sk.mark = tj_sink_mark; sk.wait = tj_sink_wait;
      /* about half of a FASTQ file is sequence; gzip packs it about four times smaller */
      (void) tjamd_reserve (g.dev, plain ? (size_t) st.st_size / 2 : (size_t) st.st_size * 2, opt.min_tract_size);
      got = plain ? tjf_parse_file (filename, threads, (size_t) TJ_FEEDER_WINDOW, &sk)
                  : tjf_parse_gz_file (filename, threads, (size_t) TJ_FEEDER_GZ_WINDOW, &sk);
      if (got == -2 || got == -3) tj_fatal ("%s", got == -2 ? "out of pinned host memory for the feeder" : tjamd_last_error ());
      if (got == -4) tj_fatal ("'%s': a gzip member does not match its own CRC-32 / size (damaged file); set TATAJUBA_AMD_FEEDER_INFLATE=zlib to read it the reference's way", filename);
      if (got >= 0) {
        n = tjamd_raw_count (g.dev);
        if (n < 0) tj_fatal ("%s", tjamd_last_error ());
        tj_priv (h)->n_device = n;
        h->n_elem = tj_n_elem (tj_priv (h)->n_host, n, h->name);
        (void) __atomic_sub_fetch (&tj_files_in_flight, 1, __ATOMIC_RELAXED);
        return h;
      }                                                 /* (-1: cannot open / map -- the one-reader path reports it) */
    }
    (void) __atomic_sub_fetch (&tj_files_in_flight, 1, __ATOMIC_RELAXED);
  }
  rd = tjr_open (filename);
  if (!rd) tj_fatal ("cannot open '%s' (the reference leaves gzopen unchecked, src/hopo_counter.c:142; this build stops)", filename);
  dev = tj_device_counter (h);
  buf[0] = (unsigned char *) tj_pinned_get (TJ_BATCH_BYTES);
  buf[1] = (unsigned char *) tj_pinned_get (TJ_BATCH_BYTES);
  if (!buf[0] || !buf[1]) tj_fatal ("%s", tjamd_last_error ());

  while ((len = tjr_next (rd, &seq)) >= 0) {           /* -2 (bad quality string) ends the file silently, as in the reference */
    if ((size_t) len + 1 > TJ_BATCH_BYTES) {            /* one read larger than a batch: send it on its own */
      unsigned char *big = (unsigned char *) malloc ((size_t) len + 1);
      memcpy (big, seq, (size_t) len); big[len] = '\n';
      if (fill) { if (tjamd_scan_host (dev, buf[cur], fill, opt.min_tract_size)) tj_fatal ("%s", tjamd_last_error ()); fill = 0; }
      if (tjamd_scan_host (dev, big, (size_t) len + 1, opt.min_tract_size) || tjamd_sync (dev)) tj_fatal ("%s", tjamd_last_error ());
      free (big);
      in_flight = 0;
      continue;
    }
    if (fill + (size_t) len + 1 > TJ_BATCH_BYTES) {     /* batch full: queue copy + scan, parse on into the other buffer */
      if (in_flight && tjamd_raw_count (dev) < 0) tj_fatal ("%s", tjamd_last_error ()); /* the other buffer's copy must be done (+ exact counts: see tj_sink_sync) */
      if (tjamd_scan_host (dev, buf[cur], fill, opt.min_tract_size)) tj_fatal ("%s", tjamd_last_error ());
      in_flight = 1; cur ^= 1; fill = 0;
    }
    memcpy (buf[cur] + fill, seq, (size_t) len);
    buf[cur][fill + (size_t) len] = '\n';
    fill += (size_t) len + 1;
  }
  if (fill && tjamd_scan_host (dev, buf[cur], fill, opt.min_tract_size)) tj_fatal ("%s", tjamd_last_error ());
  tjr_close (rd);
  n = tjamd_raw_count (dev);                            /* synchronises the stream */
  if (n < 0) tj_fatal ("%s", tjamd_last_error ());
  tj_pinned_put (buf[0]); tj_pinned_put (buf[1]);
  tj_priv (h)->n_device = n;
  h->n_elem = tj_n_elem (tj_priv (h)->n_host, n, h->name);
  return h;
}

long
tjamd_read_file_stream (const char *path, unsigned char *out, long capacity, long *n_reads)
{
  tjr_reader *rd = tjr_open (path);
  long total = 0, n = 0, len;
  const char *seq;
  if (!rd) return -1;
  while ((len = tjr_next (rd, &seq)) >= 0) {
    if (out && total + len + 1 <= capacity) { memcpy (out + total, seq, (size_t) len); out[total + len] = '\n'; }
    total += len + 1; n++;
  }
  tjr_close (rd);
  if (n_reads) *n_reads = n;
  return total;
}

/* the multi-threaded feeder into host memory (tests compare it with tjamd_read_file_stream) */
typedef struct { unsigned char *out; long cap, total, n_reads; } tj_mem_sink;
static void *tj_mem_alloc (void *ctx, size_t bytes) { (void) ctx; return malloc (bytes); }
static void tj_mem_release (void *ctx, void *p) { (void) ctx; free (p); }
static int tj_mem_put (void *ctx, const unsigned char *stream, size_t n_bytes, long n_reads)
{
  tj_mem_sink *m = (tj_mem_sink *) ctx;
  if (m->out && m->total + (long) n_bytes <= m->cap) memcpy (m->out + m->total, stream, n_bytes);
  m->total += (long) n_bytes; m->n_reads += n_reads;
  return 0;
}

long
tjamd_read_file_stream_mt (const char *path, unsigned char *out, long capacity, long *n_reads, int n_threads, long window_bytes)
{
  tj_mem_sink m = {out, capacity, 0, 0};
  tjf_sink sk = {&m, tj_mem_alloc, tj_mem_release, tj_mem_put, NULL, NULL, NULL};
  long got;
  const int plain = tjf_is_plain_file (path);
  if (plain < 0) return -1;
  got = plain ? tjf_parse_file (path, n_threads, (size_t) (window_bytes > 0 ? window_bytes : TJ_FEEDER_WINDOW), &sk)
              : tjf_parse_gz_file (path, n_threads, (size_t) (window_bytes > 0 ? window_bytes : TJ_FEEDER_GZ_WINDOW), &sk);
  if (got < 0) return -1;
  if (n_reads) *n_reads = m.n_reads;
  return m.total;
}

/* diagnostic (not in the public header): windows the feeder accepted in its last call, and whether it fell back */
long tjamd_debug_feeder_stats (long *fell_back) { long w = 0; tjf_last_stats (&w, fell_back); return w; }
long tjamd_debug_feeder_bgzf_blocks (void) { return tjf_last_bgzf_blocks (); }
/* stretches of a one-member gzip file that were decoded side by side and used / block starts that turned out not to be */
long tjamd_debug_feeder_gz_stretches (void) { return tjf_last_gz_stretches (); }
long tjamd_debug_feeder_gz_false_starts (void) { return tjf_last_gz_false_starts (); }

unsigned tjamd_debug_crc32 (const unsigned char *p, long n) { return tji_crc32 (0u, p, (size_t) n); }

/* diagnostic (tests): the feeder's DEFLATE decoder on a raw stream, its output produced `chunk` bytes at a time in a
 * buffer that is reused -- the last 32 KiB moved in front of it each time, as the feeder does between two windows.
 * Returns the inflated size (-1: error, -2: truncated, -3: out too small); *in_used = bytes of `in` consumed. */
long
tjamd_debug_inflate (const unsigned char *in, long in_len, unsigned char *out, long out_cap, long chunk, long *in_used)
{
  tji_state *st = (tji_state *) malloc (sizeof (tji_state));
  unsigned char *buf = (unsigned char *) malloc (32768 + (size_t) chunk);
  size_t in_pos = 0, hist = 0;
  long total = 0, rc;
  tji_init (st);
  for (;;) {
    size_t out_pos = 0;
    const int r = tji_inflate (st, in, (size_t) in_len, &in_pos, buf + 32768, (size_t) chunk, &out_pos, hist);
    if (total + (long) out_pos > out_cap) { rc = -3; break; }
    memcpy (out + total, buf + 32768, out_pos);
    total += (long) out_pos;
    if (r == TJI_DONE) { rc = total; break; }
    if (r == TJI_ERROR) { rc = -1; break; }
    if (r == TJI_MORE_INPUT) { rc = -2; break; }
    /* output full: the last 32 KiB of everything written so far go in front of the buffer */
    {
      const size_t have = hist + out_pos, keep = have < 32768 ? have : 32768;
      memmove (buf + 32768 - keep, buf + 32768 + out_pos - keep, keep);
      hist = keep;
    }
  }
  if (in_used) *in_used = (long) in_pos;
  free (buf); free (st);
  return rc;
}

/* ---- one sequence, synchronously (reference: src/hopo_counter.c:219-258) ------------------------------------------ */

static void tj_scan_seq (hopo_counter hc, char *seq, int seq_length, int min_tract_size);

void
update_hopo_counter_from_seq (hopo_counter hc, char *seq, int seq_length, int min_tract_size)
{
  tj_scan_seq (hc, seq, seq_length, min_tract_size < 1 ? 1 : min_tract_size);
}

/* reference: src/hopo_counter.c:260-283 */
void
update_hopo_counter_from_seq_all_monomers (hopo_counter hc, char *seq, int seq_length)
{
  tj_scan_seq (hc, seq, seq_length, 0);
}

/* Device context for the synchronous string scans.  The reference creates a counter per reference window
 * (src/genome_set.c:530-555: thousands of them); giving each its own stream, events and device buffers would cost
 * more than the scans.  They share one context per calling thread and k-mer size, bound to the device that thread's
 * first counter was given (counters that scan files keep a context of their own: tj_device_counter). */
#define TJ_SCRATCH_SLOTS 4
typedef struct { int k; tjamd_counter *dev; } tj_scratch_slot;
static __thread tj_scratch_slot tj_scratch[TJ_SCRATCH_SLOTS];
static __thread int tj_scratch_next;                    /* the slot that is recycled next when all are taken */
static pthread_key_t tj_scratch_key;                    /* its destructor releases a thread's contexts when the thread ends */
static pthread_once_t tj_scratch_once = PTHREAD_ONCE_INIT;

static int tj_process_exiting;                          /* set by an atexit handler: the HIP runtime may be gone by the time a late thread ends */
static void tj_mark_exit (void) { __atomic_store_n (&tj_process_exiting, 1, __ATOMIC_RELAXED); }

static void
tj_scratch_release (void *slots)
{
  tj_scratch_slot *sl = (tj_scratch_slot *) slots;
  int i;
  if (!sl) return;
  /* a thread that ends while the process is being taken down: the runtime's own destructors may have run -- leave the
   * contexts to the operating system rather than call into it */
  if (__atomic_load_n (&tj_process_exiting, __ATOMIC_RELAXED)) return;
  for (i = 0; i < TJ_SCRATCH_SLOTS; i++) if (sl[i].dev) { tjamd_counter_destroy (sl[i].dev); sl[i].dev = NULL; sl[i].k = 0; }
}
static void tj_scratch_make_key (void) { (void) pthread_key_create (&tj_scratch_key, tj_scratch_release); (void) atexit (tj_mark_exit); }

/* release the calling thread's shared scan contexts now (a thread pool that outlives its work; the main thread before
 * exit).  They are released by themselves when a thread ends. */
void
tjamd_thread_cleanup (void)
{
  tj_scratch_release (tj_scratch);
}

static tjamd_counter *
tj_scratch_counter (int kmer_size)
{
  int i, n, dev;
  const char *pin = getenv ("TATAJUBA_AMD_DEVICE");
  for (i = 0; i < TJ_SCRATCH_SLOTS; i++) if (tj_scratch[i].dev && tj_scratch[i].k == kmer_size) return tj_scratch[i].dev;
  n = tjamd_device_count ();
  if (n <= 0) tj_fatal ("no HIP device is visible; the homopolymer scan runs on an MI355X only (there is no CPU fallback)");
  dev = pin ? atoi (pin) : (__atomic_fetch_add (&tj_next_device, 1, __ATOMIC_RELAXED) % n);
  (void) pthread_once (&tj_scratch_once, tj_scratch_make_key);
  (void) pthread_setspecific (tj_scratch_key, tj_scratch);       /* (the thread-local table itself: valid until the destructor has run) */
  for (i = 0; i < TJ_SCRATCH_SLOTS && tj_scratch[i].dev; i++) ;
  if (i == TJ_SCRATCH_SLOTS) {                          /* (more than four k-mer sizes in one thread: recycle them in turn) */
    i = tj_scratch_next;
    tj_scratch_next = (tj_scratch_next + 1) % TJ_SCRATCH_SLOTS;
    tjamd_counter_destroy (tj_scratch[i].dev);
    tj_scratch[i].dev = NULL;
  }
  tj_scratch[i].k = kmer_size;
  tj_scratch[i].dev = tjamd_counter_create (dev, kmer_size);
  if (!tj_scratch[i].dev) tj_fatal ("%s", tjamd_last_error ());
  return tj_scratch[i].dev;
}

static void
tj_scan_seq (hopo_counter hc, char *seq, int seq_length, int min_tract_size)
{
  tj_private *pv = tj_priv (hc);
  tjamd_counter *dev;
  tjamd_located_record *rec;
  unsigned char *stream;
  long i, n, cap;

  if (seq_length <= hc->kmer_size) return;              /* reference loop bound :226 */
  dev = pv->dev ? pv->dev : tj_scratch_counter (hc->kmer_size);
  cap = (min_tract_size ? seq_length / 2 : seq_length) + 2;
  stream = (unsigned char *) malloc ((size_t) seq_length + 1);
  rec = (tjamd_located_record *) malloc ((size_t) cap * sizeof (tjamd_located_record));
  if (!stream || !rec) tj_fatal ("out of memory (a string of %d bases)", seq_length);
  memcpy (stream, seq, (size_t) seq_length);
  stream[seq_length] = '\n';
  n = tjamd_scan_host_located (dev, stream, (size_t) seq_length + 1, min_tract_size, rec, cap);
  if (n < 0) tj_fatal ("%s", tjamd_last_error ());
  for (i = 0; i < n; i++) {                             /* reference add_kmer_to_hopo_counter :285-307 */
    hopo_element *e;
    if (pv->n_host == hc->n_alloc) {
      hc->n_alloc *= 2;
      hc->elem = (hopo_element *) realloc (hc->elem, (size_t) hc->n_alloc * sizeof (hopo_element));
      if (!hc->elem) tj_fatal ("out of memory (%d tract records)", hc->n_alloc);
    }
    e = hc->elem + pv->n_host++;
    e->context[0] = rec[i].ctx0; e->context[1] = rec[i].ctx1;
    memcpy ((char *) e + 16, &rec[i].meta, 8);
    e->read_offset = (int32_t) ((long) rec[i].pos - hc->kmer_size);
    e->loc_ref_id = e->loc_pos = e->loc_last = -1;
  }
  hc->n_elem = tj_n_elem (pv->n_host, pv->n_device, hc->name);
  free (rec); free (stream);
}

/* ---- finalise (reference: src/hopo_counter.c:339-417) ------------------------------------------------------------- */

void
finalise_hopo_counter (hopo_counter hc)
{
  tj_private *pv = tj_priv (hc);
  tjamd_counter *dev;
  int status = 0;
  long n1;

  if (!hc->n_elem) {                                    /* reference :345-349 */
    hc->ref_start = hc->n_elem = 0;
    tj_warning ("No HTs were found in sample %s%s, not even before QC. This sample will be excluded.", hc->name, (hc->opt.paired_end ? " together with its pair." : "."));
    return;
  }
  dev = tj_device_counter (hc);
  if (pv->n_host && tjamd_upload_raw (dev, hc->elem, pv->n_host)) tj_fatal ("%s", tjamd_last_error ());
  if (tjamd_finalise (dev, hc->opt.remove_biased, hc->opt.min_coverage, &status)) tj_fatal ("%s", tjamd_last_error ());
  pv->n_host = 0; pv->n_device = 0;

  if (status == 1) { hc->ref_start = hc->n_elem = 0; return; }
  if (status == 2) {                                    /* reference :376-381 */
    hc->ref_start = hc->n_elem = 0;
    if (hc->opt.remove_biased) tj_warning ("No HTs found in sample %s after excluding those present only in one strand. This sample will be excluded.", hc->name);
    else tj_warning ("No HTs found in sample %s %s after excluding those seen only once. This sample will be excluded.", hc->name, (hc->opt.paired_end ? "(and pair)" : ""));
    return;
  }
  n1 = tjamd_kept_count (dev);                          /* reference :382-386 */
  free (hc->elem);
  hc->elem = (hopo_element *) malloc ((size_t) n1 * sizeof (hopo_element));
  if (tjamd_download_kept (dev, hc->elem, n1) != n1) tj_fatal ("%s", tjamd_last_error ());
  hc->n_alloc = hc->n_elem = (int) n1;

  if (status == 3) {                                    /* reference :406-411 (index arrays stay allocated, n_idx = 0) */
    hc->idx_initial = (int *) malloc ((size_t) n1 * sizeof (int));
    hc->idx_final = (int *) malloc ((size_t) n1 * sizeof (int));
    hc->n_idx = 0;
    hc->ref_start = hc->n_elem = 0;
    tj_warning ("From the HTs found in sample %s%s, not a single one has coverage higher than %d. This sample will be excluded. Try decreasing the 'Min depth of tract lengths'.",
                hc->name, (hc->opt.paired_end ? " (together with its pair)" : ""), hc->opt.min_coverage);
    return;
  }
  hc->n_idx = tjamd_n_idx (dev);                        /* reference :412-413 */
  hc->idx_initial = (int *) malloc ((size_t) hc->n_idx * sizeof (int));
  hc->idx_final = (int *) malloc ((size_t) hc->n_idx * sizeof (int));
  if (tjamd_download_idx (dev, hc->idx_initial, hc->idx_final, hc->n_idx) != hc->n_idx) tj_fatal ("%s", tjamd_last_error ());
  hc->coverage = tjamd_coverage (dev);                  /* reference :415 */
  if (find_reference_location_and_sort_hopo_counter) find_reference_location_and_sort_hopo_counter (hc); /* :416, if linked */
}

/* ---- host helpers kept for callers (not on the accelerated path) --------------------------------------------------- */

int
compare_hopo_element_decreasing (const void *a, const void *b)
{ /* reference: src/hopo_counter.c:28-38 */
  const hopo_element *x = (const hopo_element *) a, *y = (const hopo_element *) b;
  int d = y->base - x->base;
  if (d) return d;
  if (y->context[0] != x->context[0]) return y->context[0] > x->context[0] ? 1 : -1;
  if (y->context[1] != x->context[1]) return y->context[1] > x->context[1] ? 1 : -1;
  return y->length - x->length;
}

int
compare_hopo_context (hopo_element a, hopo_element b)
{ /* reference: src/hopo_counter.c:48-58 */
  int d = b.base - a.base;
  if (d) return d;
  if (b.context[0] != a.context[0]) return b.context[0] > a.context[0] ? 1 : -1;
  if (b.context[1] != a.context[1]) return b.context[1] > a.context[1] ? 1 : -1;
  return 0;
}

char *
generate_tract_as_string (uint64_t *context, int8_t base, int kmer_size, int tract_length, bool neg_strand)
{ /* reference: src/hopo_counter.c:447-469: left flank, tract, right flank; reverse-complemented when neg_strand */
  int n = 2 * kmer_size + tract_length, i;
  char *s = (char *) malloc ((size_t) n + 1);
  for (i = 0; i < kmer_size; i++) s[i] = bit_2_dna[(context[0] >> (2 * i)) & 3];
  for (i = 0; i < tract_length; i++) s[kmer_size + i] = bit_2_dna[base & 3];
  for (i = 0; i < kmer_size; i++) s[kmer_size + tract_length + i] = bit_2_dna[(context[1] >> (2 * i)) & 3];
  s[n] = '\0';
  if (neg_strand)
    for (i = 0; i < (n + 1) / 2; i++) {
      char a = s[i], b = s[n - 1 - i];
      s[i] = bit_2_dna[3 - dna_in_2_bits[(unsigned char) b][0]];
      s[n - 1 - i] = bit_2_dna[3 - dna_in_2_bits[(unsigned char) a][0]];
    }
  return s;
}

char *
generate_name_from_flanking_contexts (uint64_t *context, int8_t base, int kmer_size, bool neg_strand)
{ /* reference: src/hopo_counter.c:471-493: "left.B.right" */
  char *t = generate_tract_as_string (context, base, kmer_size, 1, neg_strand);
  char *s = (char *) malloc ((size_t) (2 * kmer_size + 4));
  memcpy (s, t, (size_t) kmer_size);
  s[kmer_size] = '.'; s[kmer_size + 1] = t[kmer_size]; s[kmer_size + 2] = '.';
  memcpy (s + kmer_size + 3, t + kmer_size + 1, (size_t) kmer_size);
  s[2 * kmer_size + 3] = '\0';
  free (t);
  return s;
}

/* ---- distances between packed contexts (reference: src/hopo_counter.c:61-113; callers: src/context_histogram.c) ---- */

static int
tj_mismatches (uint64_t d, int dist, int stop)
{ /* bases (2-bit groups) in which d is non-zero, added to dist, counting no further than `stop` */
  for (; d && dist < stop; d >>= 2) dist += (d & 3) != 0;
  return dist;
}

int
distance_between_single_context_kmer (uint64_t *c1, uint64_t *c2, int max_dist)
{ /* reference :61-68: never more than max_dist */
  return tj_mismatches (*c1 ^ *c2, 0, max_dist);
}

int
distance_between_context_kmer_pair (uint64_t *c1, uint64_t *c2)
{ /* reference :70-79: both flanks, uncapped */
  return tj_mismatches (c1[1] ^ c2[1], tj_mismatches (c1[0] ^ c2[0], 0, 65), 130);
}

int
distance_between_context_kmer_pair_with_edit_shift (uint64_t *c1, uint64_t *c2, int *best_shift)
{ /* reference :81-113: per flank, the cheapest of "no shift", c2 shifted by 1-3 bases, c1 shifted by 1-3 bases (a shift
   * of s bases costs s and drops the s top bases from the comparison); tried in that order, first minimum wins,
   * a zero ends the search */
  int f, total = 0;
  for (f = 0; f < 2; f++) {
    int best = 0xffffff, t;
    for (t = 0; t < 7 && best > 0; t++) {
      const int s = (t <= 3) ? t : t - 3, s1 = (t <= 3) ? 0 : s, s2 = (t <= 3) ? s : 0;
      const int dist = tj_mismatches (((c1[f] >> (2 * s1)) ^ (c2[f] >> (2 * s2))) & (~0ULL >> (2 * s)), s, 1000);
      if (dist < best) {
        best = dist;
        if (best_shift) { best_shift[2 * f] = s1; best_shift[2 * f + 1] = s2; }
      }
    }
    total += best;
  }
  return total;
}

/* reference: src/hopo_counter.c:188-203: name of the first tract of a (reference) string, NULL and length 0 if none */
char *
leftmost_hopo_name_and_length_from_string (char *seq, size_t len, int kmer_size, int min_tract_size, int *tract_length)
{
  hopo_counter hc = new_hopo_counter (kmer_size);
  char *name = NULL;
  update_hopo_counter_from_seq (hc, seq, (int) len, min_tract_size);
  *tract_length = 0;
  if (hc->n_elem) {
    *tract_length = hc->elem[0].length;
    name = generate_name_from_flanking_contexts (hc->elem[0].context, (int8_t) hc->elem[0].base, kmer_size, false);
  }
  del_hopo_counter (hc);
  return name;
}

/* reference: src/hopo_counter.c:440-445 (marked obsolete there): depth of the start-th indexed context */
int
hopo_counter_histogram_integral (hopo_counter hc, int start)
{
  int i, depth = 0;
  for (i = hc->idx_initial[start]; i < hc->idx_final[start]; i++) depth += hc->elem[i].count;
  return depth;
}

/* ---- many short strings at once (the reference's per-tract rescans, src/genome_set.c:525-577) ----------------------
 * One stream, one launch, one synchronisation for all windows instead of a device round trip per window.  Records come
 * back in window order, then in read order; read_offset is relative to the window (start of the left flank, as in
 * update_hopo_counter_from_seq).  window_of[i] = window of record i.  Returns the number of records, -1 on error. */
long
tjamd_scan_windows (int kmer_size, const char *const *seqs, const int *lens, int n_windows, int min_tract_size,
                    hopo_element *out, int *window_of, long capacity)
{
  tjamd_counter *dev;
  tjamd_located_record *rec;
  unsigned char *stream;
  long *start;
  long total = 0, n, i, cap, w = 0;
  if (n_windows < 0 || (n_windows && (!seqs || !lens))) { tj_set_last_error ("tjamd_scan_windows: bad arguments"); return -1; }
  start = (long *) malloc (((size_t) n_windows + 1) * sizeof (long));
  if (!start) { tj_set_last_error ("tjamd_scan_windows: out of memory"); return -1; }
  for (i = 0; i < n_windows; i++) { start[i] = total; total += (lens[i] > 0 ? lens[i] : 0) + 1; }
  start[n_windows] = total;
  if (!total) { free (start); return 0; }
  cap = (min_tract_size ? total / 2 : total) + 2;
  stream = (unsigned char *) malloc ((size_t) total);
  rec = (tjamd_located_record *) malloc ((size_t) cap * sizeof (tjamd_located_record));
  if (!stream || !rec) { free (stream); free (rec); free (start); tj_set_last_error ("tjamd_scan_windows: out of memory"); return -1; }
  for (i = 0; i < n_windows; i++) {
    if (lens[i] > 0) memcpy (stream + start[i], seqs[i], (size_t) lens[i]);
    stream[start[i + 1] - 1] = '\n';
  }
  dev = tj_scratch_counter (kmer_size);
  n = tjamd_scan_host_located (dev, stream, (size_t) total, min_tract_size, rec, cap);
  if (n > capacity) {
    char msg[160];
    snprintf (msg, sizeof msg, "tjamd_scan_windows: %ld tract records, caller capacity %ld", n, capacity);
    tj_set_last_error (msg);
    n = -1;
  }
  for (i = 0; i < n; i++) {
    hopo_element *e = out + i;
    while ((long) rec[i].pos >= start[w + 1]) w++;      /* records come sorted by position */
    e->context[0] = rec[i].ctx0; e->context[1] = rec[i].ctx1;
    memcpy ((char *) e + 16, &rec[i].meta, 8);
    e->read_offset = (int32_t) ((long) rec[i].pos - start[w] - kmer_size);
    e->loc_ref_id = e->loc_pos = e->loc_last = -1;
    if (window_of) window_of[i] = (int) w;
  }
  free (rec); free (stream); free (start);
  return n;
}

void
print_tatajuba_options (tatajuba_options_t opt)
{ /* reference: src/hopo_counter.c:115-133 (the GFF3 prefix lives in biomcmc's struct, which is opaque here) */
  fprintf (stderr, "%s\n", tjamd_version ());
  fprintf (stderr, "Reference genome fasta file: %s\n", opt.reference_fasta_filename ? opt.reference_fasta_filename : "(none)");
  fprintf (stderr, "Output directory:            %s\n", opt.outdir ? opt.outdir : "(none)");
  fprintf (stderr, "Number of samples:           %5d (%s)\n", opt.n_samples, (opt.paired_end ? "paired-end" : "single-end"));
  fprintf (stderr, "Max distance per flanking k-mer:  %6d\n", opt.max_distance_per_flank);
  fprintf (stderr, "Levenshtein distance for merging: %6d\n", opt.levenshtein_distance);
  fprintf (stderr, "Flanking k-mer size (context):    %6d\n", opt.kmer_size);
  fprintf (stderr, "Min tract length to consider:     %6d\n", opt.min_tract_size);
  fprintf (stderr, "Min depth of tract lengths:       %6d\n", opt.min_coverage);
  fprintf (stderr, "Remove biased tracts:             %s\n", (opt.remove_biased ? "yes" : "no"));
  if (opt.n_threads) fprintf (stderr, "Number of threads (requested or optimised): %3d\n", opt.n_threads);
  fprintf (stderr, "HIP devices visible: %d\n", tjamd_device_count ());
}
