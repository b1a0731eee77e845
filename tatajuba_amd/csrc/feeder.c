/* feeder.c -- multi-threaded host feeder for plain (uncompressed) FASTA/FASTQ files.
 *
 * The reference parses with kseq, a byte-at-a-time state machine (src/kseq.h:172-212, used at
 * src/hopo_counter.c:142-155); the single-threaded restatement of it is fastq_reader.c and stays the definition of
 * what a file contains.  This file runs SEVERAL of those readers over one memory-mapped file and proves, window by
 * window, that their concatenated output is what one reader would have produced:
 *
 *   - the file is cut into windows, a window into one range per thread; range starts inside the window are GUESSES
 *     (a line that starts with '@' or '>' and, for '@', is followed by a sequence line, a '+' line and a quality line
 *     of the same length);
 *   - every thread runs the ordinary reader from its start (fresh state) and stops at the first record that begins
 *     at or after the next range's start; tjr_record_start() tells where each record began;
 *   - the window is accepted iff every thread stopped EXACTLY at the next thread's start.  A fresh reader placed on
 *     the first byte of a record is in the same state as the reader that arrived there (see fastq_reader.h), so by
 *     induction over the ranges the concatenation equals the sequential parse.  Anything else (a wrong guess, a
 *     record longer than a range, ...) discards the results from the first inconsistent range on and the rest of
 *     the file is parsed by one reader.
 *
 * Output goes to a sink in file order (the device scan does not care about order, the tests do).
 */
#include "feeder.h"
#include "fastq_reader.h"
#include <fcntl.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <stdio.h>
#include <time.h>

static double tjf_now (void) { struct timespec t; clock_gettime (CLOCK_MONOTONIC, &t); return (double) t.tv_sec + 1e-9 * (double) t.tv_nsec; }

static long tjf_stat_windows = 0, tjf_stat_fallback = 0;   /* of the last tjf_parse_file call (diagnostics, tests) */
void tjf_last_stats (long *windows, long *fell_back) { if (windows) *windows = tjf_stat_windows; if (fell_back) *fell_back = tjf_stat_fallback; }

enum { TJF_LANDED = 0, TJF_MISMATCH = 1, TJF_BADQUAL = 2, TJF_EOF = 3, TJF_OVERFLOW = 4 };

typedef struct
{
  const unsigned char *data;
  size_t n, start, stop;
  int is_last;
  unsigned char *out;
  size_t out_cap, out_len, end_pos;
  long n_reads;
  int status;
} tjf_job;

static void *
tjf_run (void *arg)
{
  tjf_job *j = (tjf_job *) arg;
  tjr_reader *r = tjr_open_mem (j->data, j->n, j->start);
  const char *seq;
  long len;
  j->out_len = 0; j->n_reads = 0; j->end_pos = j->n; j->status = TJF_EOF;
  for (;;) {
    len = tjr_next (r, &seq);
    if (len == -1) { j->status = (j->is_last || j->stop >= j->n) ? TJF_EOF : TJF_MISMATCH; break; }
    {
      const size_t q = tjr_record_start (r);
      if (q >= j->stop) {                               /* the next range's business -- if it starts exactly there */
        j->end_pos = q;
        j->status = (j->is_last || q == j->stop) ? TJF_LANDED : TJF_MISMATCH;
        break;
      }
    }
    if (len == -2) { j->status = TJF_BADQUAL; break; }  /* the reference's read loop ends here, silently */
    if (j->out_len + (size_t) len + 1 > j->out_cap) { j->status = TJF_OVERFLOW; break; }
    memcpy (j->out + j->out_len, seq, (size_t) len);
    j->out[j->out_len + (size_t) len] = '\n';
    j->out_len += (size_t) len + 1;
    j->n_reads++;
  }
  tjr_close (r);
  return NULL;
}

/* first plausible record start at or after `from` (and before `limit`), or `limit` if there is none */
static size_t
tjf_guess_start (const unsigned char *d, size_t n, size_t from, size_t limit)
{
  size_t p = from;
  int tries;
  for (tries = 0; tries < 256 && p < limit; tries++) {
    const unsigned char *nl = (const unsigned char *) memchr (d + p, '\n', limit - p);
    size_t q, l1, l2, l3, e1, e2, e3, e4;
    if (!nl) break;
    q = (size_t) (nl - d) + 1;                          /* a line start */
    if (q >= limit) break;
    p = q;
    if (d[q] == '>') return q;
    if (d[q] != '@') continue;
    /* '@' also opens quality lines: ask for header / sequence / '+' / quality of the sequence's length */
    nl = (const unsigned char *) memchr (d + q, '\n', n - q); if (!nl) continue; e1 = (size_t) (nl - d); l1 = e1 + 1;
    nl = (const unsigned char *) memchr (d + l1, '\n', n - l1); if (!nl) continue; e2 = (size_t) (nl - d); l2 = e2 + 1;
    if (l2 >= n || d[l2] != '+') continue;
    nl = (const unsigned char *) memchr (d + l2, '\n', n - l2); if (!nl) continue; e3 = (size_t) (nl - d); l3 = e3 + 1;
    nl = (const unsigned char *) memchr (d + l3, '\n', n - l3); e4 = nl ? (size_t) (nl - d) : n;
    if (e4 - l3 != e2 - l1) continue;
    if (e4 + 1 < n && d[e4 + 1] != '@' && d[e4 + 1] != '>') continue;
    return q;
  }
  return limit;
}

int
tjf_is_plain_file (const char *path)
{
  unsigned char magic[2] = {0, 0};
  int fd = open (path, O_RDONLY);
  ssize_t got;
  if (fd < 0) return -1;
  got = read (fd, magic, 2);
  close (fd);
  return !(got == 2 && magic[0] == 0x1f && magic[1] == 0x8b);
}

long
tjf_parse_file (const char *path, int n_threads, size_t window_bytes, const tjf_sink *sink)
{
  struct stat st;
  const unsigned char *data;
  size_t n, pos = 0, cap;
  long total_reads = 0;
  int fd, i, set = 0, done = 0, fell_back = 0;
  unsigned char *buf[2][TJF_MAX_THREADS];
  tjf_job job[TJF_MAX_THREADS];
  pthread_t th[TJF_MAX_THREADS];
  int started[TJF_MAX_THREADS];
  long set_mark[2] = {-1, -1};
  long n_windows = 0;

  const int trace = getenv ("TATAJUBA_AMD_FEEDER_TRACE") != NULL;
  double t0 = tjf_now (), t_alloc = 0, t_parse = 0, t_put = 0, t_sync = 0, tt;
  if (n_threads < 1) n_threads = 1;
  if (n_threads > TJF_MAX_THREADS) n_threads = TJF_MAX_THREADS;
  fd = open (path, O_RDONLY);
  if (fd < 0) return -1;
  if (fstat (fd, &st) != 0 || st.st_size <= 0) { close (fd); return st.st_size == 0 ? 0 : -1; }
  n = (size_t) st.st_size;
  data = (const unsigned char *) mmap (NULL, n, PROT_READ, MAP_PRIVATE, fd, 0);
  close (fd);
  if (data == (const unsigned char *) MAP_FAILED) return -1;
  (void) madvise ((void *) data, n, MADV_SEQUENTIAL);
  if (window_bytes < 4096) window_bytes = 4096;
  cap = 2 * (window_bytes / (size_t) n_threads) + (1u << 16);
  tt = tjf_now ();
  for (set = 0; set < 2; set++) for (i = 0; i < n_threads; i++) {
    buf[set][i] = (unsigned char *) sink->alloc (sink->ctx, cap);
    if (!buf[set][i]) { munmap ((void *) data, n); return -2; }
  }

  t_alloc = tjf_now () - tt;
  set = 0;
  tjf_stat_windows = 0; tjf_stat_fallback = 0;
  while (!done && pos < n) {
    const size_t wend = (n - pos > window_bytes) ? pos + window_bytes : n;
    size_t starts[TJF_MAX_THREADS + 1];
    int nj = 1, accepted;
    starts[0] = pos;
    for (i = 1; i < n_threads; i++) {                   /* guessed range starts, strictly increasing */
      const size_t want = pos + (size_t) ((double) (wend - pos) * i / n_threads);
      const size_t g = tjf_guess_start (data, n, want > starts[nj - 1] ? want : starts[nj - 1], wend);
      if (g > starts[nj - 1] && g < wend) starts[nj++] = g;
    }
    starts[nj] = wend;
    tt = tjf_now ();
    /* this buffer set was sent two windows ago: with marks only ITS batches are waited for (the window sent last is
     * still in flight); every fourth window a full synchronisation lets the sink refresh what it knows (exact counts) */
    if (sink->mark && sink->wait) {
      if (set_mark[set] >= 0 && sink->wait (sink->ctx, set_mark[set])) { done = 1; total_reads = -3; break; }
      if ((n_windows & 3) == 3 && sink->sync && sink->sync (sink->ctx)) { done = 1; total_reads = -3; break; }
    }
    else if (sink->sync && sink->sync (sink->ctx)) { done = 1; total_reads = -3; break; }
    n_windows++;
    t_sync += tjf_now () - tt; tt = tjf_now ();
    for (i = 0; i < nj; i++) {
      job[i].data = data; job[i].n = n; job[i].start = starts[i]; job[i].stop = starts[i + 1]; job[i].is_last = (i == nj - 1);
      job[i].out = buf[set][i]; job[i].out_cap = cap;
      started[i] = 0;
      if (i) { if (pthread_create (&th[i], NULL, tjf_run, &job[i]) == 0) started[i] = 1; else tjf_run (&job[i]); }
    }
    tjf_run (&job[0]);
    for (i = 1; i < nj; i++) if (started[i]) pthread_join (th[i], NULL);
    t_parse += tjf_now () - tt; tt = tjf_now ();

    accepted = 0;
    for (i = 0; i < nj; i++) {                          /* the chain of ranges: each must end where the next begins */
      if (job[i].status == TJF_MISMATCH || job[i].status == TJF_OVERFLOW) break;
      accepted++;
      if (job[i].status == TJF_BADQUAL || job[i].status == TJF_EOF) { done = 1; break; }
    }
    if (accepted == nj || done) tjf_stat_windows++;      /* (the whole window came from the parallel readers) */
    for (i = 0; i < accepted; i++) {
      if (job[i].out_len && sink->put (sink->ctx, job[i].out, job[i].out_len, job[i].n_reads)) { done = 1; total_reads = -3; break; }
      total_reads += job[i].n_reads;
    }
    if (total_reads >= 0 && sink->mark) { set_mark[set] = sink->mark (sink->ctx); if (set_mark[set] < 0) total_reads = -3; }
    t_put += tjf_now () - tt;
    if (total_reads < 0) break;
    if (done) break;
    if (accepted < nj) {                                /* inconsistent guess: one reader takes the rest of the file */
      tjr_reader *r = tjr_open_mem (data, n, job[accepted].start);
      const char *seq;
      long len;
      unsigned char *b = buf[set ^ 1][0];
      size_t fill = 0;
      long nr = 0;
      fell_back = 1; tjf_stat_fallback = 1;
      if (sink->sync && sink->sync (sink->ctx)) { total_reads = -3; tjr_close (r); break; }
      while ((len = tjr_next (r, &seq)) >= 0) {
        if ((size_t) len + 1 > cap) {                   /* longer than a buffer: on its own */
          unsigned char *big = (unsigned char *) malloc ((size_t) len + 1);
          memcpy (big, seq, (size_t) len); big[len] = '\n';
          if (fill && sink->put (sink->ctx, b, fill, nr)) { total_reads = -3; free (big); break; }
          total_reads += nr; fill = 0; nr = 0;
          if (sink->put (sink->ctx, big, (size_t) len + 1, 1) || (sink->sync && sink->sync (sink->ctx))) { total_reads = -3; free (big); break; }
          total_reads += 1;
          free (big);
          continue;
        }
        if (fill + (size_t) len + 1 > cap) {
          if (sink->put (sink->ctx, b, fill, nr) || (sink->sync && sink->sync (sink->ctx))) { total_reads = -3; break; }
          total_reads += nr; fill = 0; nr = 0;
        }
        memcpy (b + fill, seq, (size_t) len); b[fill + (size_t) len] = '\n';
        fill += (size_t) len + 1; nr++;
      }
      if (total_reads >= 0 && fill) { if (sink->put (sink->ctx, b, fill, nr)) total_reads = -3; else total_reads += nr; }
      tjr_close (r);
      break;
    }
    pos = job[nj - 1].end_pos;
    set ^= 1;
  }
  if (sink->sync) (void) sink->sync (sink->ctx);
  for (set = 0; set < 2; set++) for (i = 0; i < n_threads; i++) sink->release (sink->ctx, buf[set][i]);
  munmap ((void *) data, n);
  if (trace) fprintf (stderr, "[feeder] %s: %d threads, %ld windows%s, alloc %.1f ms, sync %.1f ms, parse %.1f ms, put %.1f ms, total %.1f ms\n", path, n_threads,
                      tjf_stat_windows, fell_back ? " + fallback" : "", t_alloc * 1e3, t_sync * 1e3, t_parse * 1e3, t_put * 1e3, (tjf_now () - t0) * 1e3);
  (void) fell_back;
  return total_reads;
}
