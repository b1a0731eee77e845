/* feeder.c -- multi-threaded host feeder for FASTA/FASTQ files, plain or gzip.
 *
 * The reference parses with kseq, a byte-at-a-time state machine (src/kseq.h:172-212, used at
 * src/hopo_counter.c:142-155); the single-threaded restatement of it is fastq_reader.c and stays the definition of
 * what a file contains.  This file runs SEVERAL of those readers over one stretch of file bytes in memory and proves,
 * window by window, that their concatenated output is what one reader would have produced:
 *
 *   - the bytes are cut into windows, a window into one range per thread; range starts inside the window are GUESSES
 *     (a line that starts with '@' or '>' and, for '@', is followed by a sequence line, a '+' line and a quality line
 *     of the same length);
 *   - every thread runs the ordinary reader from its start (fresh state) and stops at the first record that begins
 *     at or after the next range's start; tjr_record_start() tells where each record began;
 *   - the window is accepted iff every thread stopped EXACTLY at the next thread's start.  A fresh reader placed on
 *     the first byte of a record is in the same state as the reader that arrived there (see fastq_reader.h), so by
 *     induction over the ranges the concatenation equals the sequential parse.  Where a range did not land (a wrong
 *     guess, a record longer than a range, ...) the results of the ranges after it are discarded and one reader
 *     takes the rest of that window; the next window is cut afresh.
 *
 * A plain file is mapped whole and the windows walk over it.  A gzip file (the usual input: the reference opens
 * everything through zlib, src/hopo_counter.c:142) is inflated a VIEW at a time by a producer thread that runs one
 * view ahead of the parse: BGZF files (bgzip: independent members that state their compressed and uncompressed
 * sizes) by all threads at once, any other gzip member by all threads as well -- entered at block starts found by trial,
 * every stretch checked by the decoder of the stretch in front (tjz_round) -- or by one thread when that finds nothing.  A
 * view ends wherever the inflated bytes happened to end; the last reader of a view gives back the record it was in
 * when it touched the end (tjr_at_end), and those bytes open the next view.
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
#include <zlib.h>
#include "tj_inflate.h"

static double tjf_now (void) { struct timespec t; clock_gettime (CLOCK_MONOTONIC, &t); return (double) t.tv_sec + 1e-9 * (double) t.tv_nsec; }

static long tjf_stat_windows = 0, tjf_stat_fallback = 0, tjf_stat_bgzf = 0;   /* of the last parse call (diagnostics, tests) */
void tjf_last_stats (long *windows, long *fell_back) { if (windows) *windows = tjf_stat_windows; if (fell_back) *fell_back = tjf_stat_fallback; }
long tjf_last_bgzf_blocks (void) { return tjf_stat_bgzf; }

enum { TJF_LANDED = 0, TJF_MISMATCH = 1, TJF_BADQUAL = 2, TJF_EOF = 3, TJF_OVERFLOW = 4, TJF_CUT = 5 };

typedef struct
{
  const unsigned char *data;
  size_t n, start, stop;
  int is_last;
  int open_end;         /* the bytes are a view of an unfinished file: give back the record that is cut by their end */
  unsigned char *out;
  size_t out_cap, out_len, end_pos;
  long n_reads;
  int status;
} tjf_job;

/* One range.  Its start is a record start that is proven by the time the output is used, so what it emits is right
 * whatever happens to the ranges after it; status says how it ended: LANDED on the next range's start, MISMATCH (the
 * first record at or after `stop` starts somewhere else, at end_pos: the next start was a wrong guess), CUT (view only:
 * end_pos is the record to read again when more bytes are there), EOF / BADQUAL (the file ends here for any reader),
 * OVERFLOW (the output buffer is too small: nothing of this range is used). */
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
    if (j->open_end && tjr_at_end (r)) {
      j->end_pos = tjr_record_open (r) ? tjr_record_start (r) : j->n;
      j->status = TJF_CUT;
      break;
    }
    if (len == -1) { j->status = TJF_EOF; break; }
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
    if (d[q] == '>') {                                  /* a FASTA header -- or a quality line (Phred 29), which a header line follows */
      nl = (const unsigned char *) memchr (d + q, '\n', n - q);
      if (nl && (size_t) (nl - d) + 1 < n && nl[1] != '@' && nl[1] != '>' && nl[1] != '+') return q;
      continue;
    }
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

/* ---- the windowed parse over bytes in memory ---------------------------------------------------------------------- */

typedef struct
{
  const tjf_sink *sink;
  int n_threads;
  size_t cap;
  unsigned char *buf[2][TJF_MAX_THREADS];
  long set_mark[2];
  int set;
  long n_windows;
  long total_reads;     /* < 0: -2 out of memory, -3 the sink failed */
  int done;             /* the file ends here for the reference too (bad quality string), or an error */
  double t_alloc, t_parse, t_put, t_sync;
} tjf_state;

static int
tjf_state_init (tjf_state *s, const tjf_sink *sink, int n_threads, size_t window_bytes)
{
  int set, i;
  const double tt = tjf_now ();
  memset (s, 0, sizeof *s);
  if (n_threads < 1) n_threads = 1;
  if (n_threads > TJF_MAX_THREADS) n_threads = TJF_MAX_THREADS;
  s->sink = sink; s->n_threads = n_threads;
  s->cap = 2 * (window_bytes / (size_t) n_threads) + (1u << 16);
  s->set_mark[0] = s->set_mark[1] = -1;
  for (set = 0; set < 2; set++) for (i = 0; i < n_threads; i++) {
    s->buf[set][i] = (unsigned char *) sink->alloc (sink->ctx, s->cap);
    if (!s->buf[set][i]) return -2;
  }
  s->t_alloc = tjf_now () - tt;
  return 0;
}

static void
tjf_state_free (tjf_state *s)
{
  int set, i;
  for (set = 0; set < 2; set++) for (i = 0; i < s->n_threads; i++) if (s->buf[set][i]) s->sink->release (s->sink->ctx, s->buf[set][i]);
}

/* the buffer set about to be written was sent two windows ago: with marks only ITS batches are waited for (the window
 * sent last is still in flight); every fourth window a full synchronisation lets the sink refresh what it knows (exact
 * counts).  Returns != 0 if the sink failed. */
static int
tjf_claim_set (tjf_state *s, int set)
{
  const tjf_sink *sink = s->sink;
  const double tt = tjf_now ();
  int bad = 0;
  if (sink->mark && sink->wait) {
    if (s->set_mark[set] >= 0 && sink->wait (sink->ctx, s->set_mark[set])) bad = 1;
    else if ((s->n_windows & 3) == 3 && sink->sync && sink->sync (sink->ctx)) bad = 1;
  }
  else if (sink->sync && sink->sync (sink->ctx)) bad = 1;
  s->n_windows++;
  s->t_sync += tjf_now () - tt;
  if (bad) { s->done = 1; s->total_reads = -3; }
  return bad;
}

/* One reader over data[start, ...) up to the first record that begins at or after `stop`: the repair after a wrong
 * guess or a range too big for its buffer.  Records go out through buffer 0 of the current set, which is flushed (and
 * waited for) whenever it is full.  open_end as in tjf_job.  Returns the position reached. */
static size_t
tjf_sequential (tjf_state *s, const unsigned char *data, size_t n, size_t start, size_t stop, int open_end)
{
  const tjf_sink *sink = s->sink;
  tjr_reader *r = tjr_open_mem (data, n, start);
  unsigned char *b = s->buf[s->set][0];
  const size_t cap = s->cap;
  const char *seq;
  size_t fill = 0, end_pos = n;
  long len, nr = 0;
  for (;;) {
    len = tjr_next (r, &seq);
    if (open_end && tjr_at_end (r)) { end_pos = tjr_record_open (r) ? tjr_record_start (r) : n; break; }
    if (len == -1) { s->done = 1; break; }
    if (tjr_record_start (r) >= stop) { end_pos = tjr_record_start (r); break; }
    if (len == -2) { s->done = 1; break; }
    if ((size_t) len + 1 > cap) {                       /* longer than a buffer: on its own */
      unsigned char *big = (unsigned char *) malloc ((size_t) len + 1);
      if (!big) { s->total_reads = -2; break; }
      memcpy (big, seq, (size_t) len); big[len] = '\n';
      if (fill && sink->put (sink->ctx, b, fill, nr)) { s->total_reads = -3; free (big); break; }
      s->total_reads += nr; fill = 0; nr = 0;
      if (sink->put (sink->ctx, big, (size_t) len + 1, 1) || (sink->sync && sink->sync (sink->ctx))) { s->total_reads = -3; free (big); break; }
      s->total_reads += 1;
      free (big);
      continue;
    }
    if (fill + (size_t) len + 1 > cap) {
      if (sink->put (sink->ctx, b, fill, nr) || (sink->sync && sink->sync (sink->ctx))) { s->total_reads = -3; break; }
      s->total_reads += nr; fill = 0; nr = 0;
    }
    memcpy (b + fill, seq, (size_t) len); b[fill + (size_t) len] = '\n';
    fill += (size_t) len + 1; nr++;
  }
  if (s->total_reads >= 0 && fill) {
    if (sink->put (sink->ctx, b, fill, nr) || (sink->sync && sink->sync (sink->ctx))) s->total_reads = -3; else s->total_reads += nr;
  }
  tjr_close (r);
  if (s->total_reads < 0) s->done = 1;
  return end_pos;
}

/* One window: data[pos, wend) of data[0, n).  whole_file: the bytes are the whole rest of the file (the last range may
 * read past wend); otherwise wend == n is where a view of an unfinished file ends.  In three steps, so that a caller with
 * the whole file in memory can have the next window parsed while this one's batches are handed to the sink
 * (tjf_parse_file): tjf_win_start (buffer set claimed by the caller, range starts guessed, one reader per range),
 * tjf_win_join, tjf_win_finish (the chain of ranges checked, their output sent, a wrong guess repaired; returns the
 * position reached). */
typedef struct
{
  const unsigned char *data;
  size_t n, pos, wend;
  int whole_file, set, nj;
  tjf_job job[TJF_MAX_THREADS];
  pthread_t th[TJF_MAX_THREADS];
  int started[TJF_MAX_THREADS];
  double t_start;
} tjf_win;

/* all_threads: every range gets a thread of its own (the caller has something else to do until tjf_win_join) */
static void
tjf_win_start (tjf_state *s, tjf_win *w, const unsigned char *data, size_t n, size_t pos, size_t wend, int whole_file, int set, int all_threads)
{
  size_t starts[TJF_MAX_THREADS + 1];
  int nj = 1, i;
  starts[0] = pos;
  for (i = 1; i < s->n_threads; i++) {                  /* guessed range starts, strictly increasing */
    const size_t want = pos + (size_t) ((double) (wend - pos) * i / s->n_threads);
    const size_t g = tjf_guess_start (data, n, want > starts[nj - 1] ? want : starts[nj - 1], wend);
    if (g > starts[nj - 1] && g < wend) starts[nj++] = g;
  }
  starts[nj] = wend;
  w->data = data; w->n = n; w->pos = pos; w->wend = wend; w->whole_file = whole_file; w->set = set; w->nj = nj;
  w->t_start = tjf_now ();
  for (i = 0; i < nj; i++) {
    tjf_job *j = &w->job[i];
    j->data = data; j->n = n; j->start = starts[i]; j->stop = starts[i + 1]; j->is_last = (i == nj - 1);
    j->open_end = !whole_file;
    j->out = s->buf[set][i]; j->out_cap = s->cap;
    w->started[i] = 0;
    if (i || all_threads) { if (pthread_create (&w->th[i], NULL, tjf_run, j) == 0) w->started[i] = 1; else tjf_run (j); }
  }
  if (!all_threads) tjf_run (&w->job[0]);
}

static void
tjf_win_join (tjf_state *s, tjf_win *w)
{
  int i;
  for (i = 0; i < w->nj; i++) if (w->started[i]) { pthread_join (w->th[i], NULL); w->started[i] = 0; }
  s->t_parse += tjf_now () - w->t_start;
}

/* every range landed on the start of the next and the last one inside the data: the window after this one begins at the
 * last range's end, whatever the sink makes of this one's batches (a sink that fails ends the file anyway) */
static int
tjf_win_plain (const tjf_win *w)
{
  int i;
  for (i = 0; i < w->nj; i++) if (w->job[i].status != TJF_LANDED) return 0;
  return 1;
}

static size_t
tjf_win_finish (tjf_state *s, tjf_win *w)
{
  const tjf_sink *sink = s->sink;
  tjf_job *job = w->job;
  const unsigned char *data = w->data;
  const size_t n = w->n, wend = w->wend;
  size_t pos = w->pos;
  const int nj = w->nj, whole_file = w->whole_file;
  int accepted, i, cut = 0, repair = 0;
  size_t repair_from = 0;
  double tt = tjf_now ();

  accepted = 0;
  for (i = 0; i < nj; i++) {                            /* the chain of ranges: each must end where the next begins */
    const int st = job[i].status;
    if (st == TJF_OVERFLOW) { repair = 1; repair_from = job[i].start; break; }
    accepted++;                                         /* (it began on a proven record start: its records are right) */
    if (st == TJF_BADQUAL || st == TJF_EOF) { s->done = 1; break; }
    if (st == TJF_CUT) { cut = 1; break; }
    if (st == TJF_MISMATCH) { repair = 1; repair_from = job[i].end_pos; break; }
  }
  if (!repair) tjf_stat_windows++;                      /* (the whole window came from the parallel readers) */
  for (i = 0; i < accepted; i++) {
    if (job[i].out_len && sink->put (sink->ctx, job[i].out, job[i].out_len, job[i].n_reads)) { s->done = 1; s->total_reads = -3; break; }
    s->total_reads += job[i].n_reads;
  }
  if (s->total_reads >= 0 && sink->mark) { s->set_mark[w->set] = sink->mark (sink->ctx); if (s->set_mark[w->set] < 0) s->total_reads = -3; }
  s->t_put += tjf_now () - tt;
  if (s->total_reads < 0) { s->done = 1; return pos; }
  s->set = w->set ^ 1;                                  /* (the set just sent is in flight) */
  if (s->done) return n;
  if (cut) return job[accepted - 1].end_pos;
  if (repair) {                                         /* a wrong guess: one reader for the rest of this window */
    tjf_stat_fallback++;
    if (sink->sync && sink->sync (sink->ctx)) { s->done = 1; s->total_reads = -3; return pos; }
    s->set_mark[0] = s->set_mark[1] = -1;
    tt = tjf_now ();
    pos = tjf_sequential (s, data, n, repair_from, wend, !whole_file);
    s->t_parse += tjf_now () - tt;
    return pos;
  }
  return job[nj - 1].end_pos;
}

static size_t
tjf_window (tjf_state *s, const unsigned char *data, size_t n, size_t pos, size_t wend, int whole_file)
{
  tjf_win w;
  if (tjf_claim_set (s, s->set)) return pos;
  tjf_win_start (s, &w, data, n, pos, wend, whole_file, s->set, 0);
  tjf_win_join (s, &w);
  return tjf_win_finish (s, &w);
}

static void
tjf_trace (const tjf_state *s, const char *path, const char *kind, double t0, double t_inflate)
{
  if (!getenv ("TATAJUBA_AMD_FEEDER_TRACE")) return;
  fprintf (stderr, "[feeder] %s (%s): %d threads, %ld windows%s, alloc %.1f ms, sync %.1f ms, parse %.1f ms, put %.1f ms, waiting for inflate %.1f ms, total %.1f ms\n",
           path, kind, s->n_threads, tjf_stat_windows, tjf_stat_fallback ? " + one-reader repairs" : "", s->t_alloc * 1e3, s->t_sync * 1e3, s->t_parse * 1e3,
           s->t_put * 1e3, t_inflate * 1e3, (tjf_now () - t0) * 1e3);
}

static const unsigned char *
tjf_map (const char *path, size_t *n)
{
  struct stat st;
  const unsigned char *data;
  const int fd = open (path, O_RDONLY);
  *n = 0;
  if (fd < 0) return NULL;
  if (fstat (fd, &st) != 0 || st.st_size < 0) { close (fd); return NULL; }
  if (st.st_size == 0) { close (fd); return (const unsigned char *) ""; }
  *n = (size_t) st.st_size;
  data = (const unsigned char *) mmap (NULL, *n, PROT_READ, MAP_PRIVATE, fd, 0);
  close (fd);
  if (data == (const unsigned char *) MAP_FAILED) { *n = 0; return NULL; }
  (void) madvise ((void *) data, *n, MADV_SEQUENTIAL);
  return data;
}

long
tjf_parse_file (const char *path, int n_threads, size_t window_bytes, const tjf_sink *sink)
{
  const unsigned char *data;
  size_t n, pos = 0;
  tjf_state s;
  const double t0 = tjf_now ();
  long total;

  data = tjf_map (path, &n);
  if (!data) return -1;
  if (n == 0) return 0;
  if (window_bytes < 4096) window_bytes = 4096;
  tjf_stat_windows = 0; tjf_stat_fallback = 0; tjf_stat_bgzf = 0;
  if (tjf_state_init (&s, sink, n_threads, window_bytes)) { tjf_state_free (&s); munmap ((void *) data, n); return -2; }
  {
    /* Two windows in turn: while the batches of one go to the sink (copies and kernel launches queued by this thread,
     * a fifth of the file's time), the readers are already on the next -- whose start is the end of this one's last range
     * as soon as every range has landed on the next one's start, which is known before anything is sent.  Anything else
     * (a wrong guess to repair, the end of the file, a bad quality line) is finished first, as in tjf_window. */
    tjf_win *w = (tjf_win *) malloc (2 * sizeof (tjf_win));
    int cur = 0, have = 0;
    if (!w) { tjf_state_free (&s); munmap ((void *) data, n); return -2; }
    while (!s.done && (have || pos < n)) {
      int ahead = 0;
      if (!have) {
        const size_t wend = (n - pos > window_bytes) ? pos + window_bytes : n;
        if (tjf_claim_set (&s, s.set)) break;
        tjf_win_start (&s, &w[cur], data, n, pos, wend, 1, s.set, 1);
      }
      tjf_win_join (&s, &w[cur]);
      have = 0;
      if (tjf_win_plain (&w[cur])) {
        const size_t npos = w[cur].job[w[cur].nj - 1].end_pos;
        if (npos < n) {
          const size_t wend = (n - npos > window_bytes) ? npos + window_bytes : n;
          if (tjf_claim_set (&s, w[cur].set ^ 1)) break;                  /* (waits for the batches sent from that set two windows ago) */
          tjf_win_start (&s, &w[cur ^ 1], data, n, npos, wend, 1, w[cur].set ^ 1, 1);
          ahead = 1;
        }
      }
      pos = tjf_win_finish (&s, &w[cur]);
      if (ahead) { have = 1; cur ^= 1; }
    }
    if (have) tjf_win_join (&s, &w[cur]);               /* (the sink failed: the readers that are under way are waited for, their output dropped) */
    free (w);
  }
  { const double tt = tjf_now (); if (sink->sync) (void) sink->sync (sink->ctx); s.t_sync += tjf_now () - tt; }
  tjf_state_free (&s);
  /* (taking the mapping apart: 2.4 ms for 627 MB, reported with the buffers' time.  Tried: a detached thread for it -- the
   * feeder returned 2.4 ms earlier and the caller's finalise, which needs the address-space lock for its own allocations,
   * waited that much longer) */
  { const double tt = tjf_now (); munmap ((void *) data, n); s.t_alloc += tjf_now () - tt; }
  tjf_trace (&s, path, "plain", t0, 0.0);
  total = s.total_reads;
  return total;
}

/* ---- gzip: inflate a view at a time --------------------------------------------------------------------------------
 * gzip member (RFC 1952): 1f 8b, CM = 8, FLG, MTIME[4], XFL, OS, then optional FEXTRA / FNAME / FCOMMENT / FHCRC, the
 * deflate stream, CRC32[4], ISIZE[4].  BGZF (the SAM specification, section 4.1; what bgzip writes) is a series of such members of
 * at most 64 KiB of data each, every one with an FEXTRA subfield 'B','C',2,0 that holds the member's total size - 1:
 * the members can be found without inflating anything and inflated independently. */

typedef struct { size_t data_off, data_len, out_off; unsigned isize, crc; } tjz_block;

typedef struct
{
  const unsigned char *z;
  size_t zn, zpos;
  int bgzf;             /* the members at zpos are BGZF blocks (until one is not) */
  int ended;            /* nothing more will come (end of input, trailing garbage, damaged stream) */
  int damaged;
  int crc_failed;       /* a member inflated by tj_inflate.c did not match its own CRC-32 / size: results cannot be trusted */
  int use_zlib;         /* TATAJUBA_AMD_FEEDER_INFLATE=zlib: zlib's inflate() instead of tj_inflate.c (tests, comparison) */
  z_stream strm;        /* (use_zlib) the one-thread inflater, gzip wrapper, live across views */
  tji_state *inf;       /* the one-thread inflater of the member at hand, live across views */
  int strm_open;
  unsigned long crc;    /* of the member's output so far */
  unsigned long long isize;
  unsigned char hist[32768];    /* the member's last output: goes in front of the next view's payload (LZ77 window) */
  size_t hist_len;
  int n_threads;
  tjz_block *blk;
  size_t blk_cap;
  /* a member that is not BGZF on several threads (tjz_round): the stream is entered at block starts found by trial
   * (tj_inflate.c: tjp_*), a round of stretches at a time */
  int par;              /* allowed: more than one thread, the feeder's own inflater */
  int par_active;       /* the member at hand is being decoded that way: pbit is a block boundary, hist the output in front of it */
  int par_final;        /* ... and its final block has been decoded (pout holds the rest of its bytes) */
  int par_giveup;       /* ... and the last round said that one decoder does this file better: it takes over once pout is handed out */
  size_t pbit;
  unsigned char *pout;  /* a round's bytes, handed out view by view */
  size_t pout_cap, pout_len, pout_pos;
  unsigned long pout_crc;       /* CRC-32 of pout[0, pout_len) (worked out by the threads that made the bytes) */
  tjp_segment seg[TJF_MAX_THREADS];
  size_t seg_bytes;     /* compressed bytes per stretch */
  long par_rounds, par_stretches, par_false;    /* diagnostics */
  double t_decode, t_resolve, t_handout;        /* (TATAJUBA_AMD_FEEDER_TRACE) */
} tjz_source;

static long tjf_stat_gz_rounds = 0, tjf_stat_gz_stretches = 0, tjf_stat_gz_false = 0;   /* of the last gzip file (tests) */
long tjf_last_gz_stretches (void) { return tjf_stat_gz_stretches; }
long tjf_last_gz_false_starts (void) { return tjf_stat_gz_false; }

/* a BGZF block header at p (avail bytes there)?  -> its total size, 0 if it is not one / not whole */
static size_t
tjz_bgzf_block (const unsigned char *p, size_t avail, size_t *data_off)
{
  size_t xlen, x, bsize = 0;
  if (avail < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || p[3] != 4) return 0;   /* FLG == FEXTRA only, as bgzip writes it */
  xlen = (size_t) p[10] | ((size_t) p[11] << 8);
  if (12 + xlen > avail) return 0;
  for (x = 12; x + 4 <= 12 + xlen; ) {
    const size_t slen = (size_t) p[x + 2] | ((size_t) p[x + 3] << 8);
    if (p[x] == 'B' && p[x + 1] == 'C' && slen == 2 && x + 6 <= 12 + xlen) bsize = ((size_t) p[x + 4] | ((size_t) p[x + 5] << 8)) + 1;
    x += 4 + slen;
  }
  if (!bsize || bsize < 12 + xlen + 8 || bsize > avail) return 0;
  *data_off = 12 + xlen;
  return bsize;
}

typedef struct { tjz_source *src; unsigned char *out; size_t first, last; int bad_at_set; size_t bad_at; } tjz_job;

static void *
tjz_inflate_blocks (void *arg)
{
  tjz_job *j = (tjz_job *) arg;
  z_stream zs;
  tji_state *ti = j->src->use_zlib ? NULL : (tji_state *) malloc (sizeof (tji_state));
  int zs_open = 0;
  size_t b;
  j->bad_at_set = 0;
  for (b = j->first; b < j->last; b++) {
    const tjz_block *k = &j->src->blk[b];
    int ok = 0;
    if (k->isize == 0) continue;                        /* (the empty block bgzip ends a file with) */
    if (ti) {
      size_t ip = 0, op = 0;
      tji_init (ti);
      ok = tji_inflate (ti, j->src->z + k->data_off, k->data_len, &ip, j->out + k->out_off, k->isize, &op, 0) == TJI_DONE && op == k->isize
           && tji_crc32 (0u, j->out + k->out_off, k->isize) == k->crc;
    }
    if (!ok) {                                          /* zlib's word on it (always, with use_zlib) */
      int rc;
      if (!zs_open) { memset (&zs, 0, sizeof zs); if (inflateInit2 (&zs, -15) != Z_OK) { j->bad_at_set = 1; j->bad_at = b; break; } zs_open = 1; }
      else inflateReset (&zs);
      zs.next_in = (Bytef *) (j->src->z + k->data_off); zs.avail_in = (uInt) k->data_len;
      zs.next_out = j->out + k->out_off; zs.avail_out = k->isize;
      rc = inflate (&zs, Z_FINISH);
      if (rc != Z_STREAM_END || zs.avail_out != 0 || (unsigned) crc32 (crc32 (0L, Z_NULL, 0), j->out + k->out_off, k->isize) != k->crc) {
        j->bad_at_set = 1; j->bad_at = b; break;
      }
    }
  }
  if (zs_open) inflateEnd (&zs);
  free (ti);
  return NULL;
}

/* CRC-32 of a stretch of output by several threads (zlib's crc32 does 1 GB/s, the inflater should not wait for it) */
typedef struct { const unsigned char *p; size_t n; unsigned long crc; } tjz_crc_job;
static void *tjz_crc_run (void *arg) { tjz_crc_job *c = (tjz_crc_job *) arg; c->crc = tji_crc32 (0u, c->p, c->n); return NULL; }
static unsigned long
tjz_crc_extend (unsigned long crc, const unsigned char *p, size_t n, int n_threads)
{
  tjz_crc_job job[TJF_MAX_THREADS];
  pthread_t th[TJF_MAX_THREADS];
  int started[TJF_MAX_THREADS], t, nt = n_threads;
  if (nt > TJF_MAX_THREADS) nt = TJF_MAX_THREADS;
  if (n < (4u << 20) || nt < 2) { tjz_crc_job one = {p, n, 0}; tjz_crc_run (&one); return crc32_combine (crc, one.crc, (z_off_t) n); }
  for (t = 0; t < nt; t++) {
    const size_t a = n * (size_t) t / (size_t) nt, b = n * (size_t) (t + 1) / (size_t) nt;
    job[t].p = p + a; job[t].n = b - a; started[t] = 0;
    if (t) { if (pthread_create (&th[t], NULL, tjz_crc_run, &job[t]) == 0) started[t] = 1; else tjz_crc_run (&job[t]); }
  }
  tjz_crc_run (&job[0]);
  for (t = 1; t < nt; t++) if (started[t]) pthread_join (th[t], NULL);
  for (t = 0; t < nt; t++) crc = crc32_combine (crc, job[t].crc, (z_off_t) job[t].n);
  return crc;
}

/* length of the gzip member header at p (RFC 1952), 0 if it is none that zlib would take (or cut short) */
static size_t
tjz_member_header (const unsigned char *p, size_t avail)
{
  size_t x = 10;
  unsigned flg;
  if (avail < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8) return 0;
  flg = p[3];
  if (flg & 0xe0u) return 0;                            /* reserved flag bits */
  if (flg & 4u) { if (x + 2 > avail) return 0; x += 2 + ((size_t) p[x] | ((size_t) p[x + 1] << 8)); }
  if (flg & 8u) { while (x < avail && p[x]) x++; x++; }
  if (flg & 16u) { while (x < avail && p[x]) x++; x++; }
  if (flg & 2u) x += 2;
  return x < avail ? x : 0;
}

/* ---- one member on several threads ----------------------------------------------------------------------------------
 * A round: n stretches of seg_bytes compressed bytes from the known block boundary pbit on.  Stretch 0 starts there;
 * every other one at the first bit position behind its nominal start that passes for the start of a block of text
 * (tjp_find_block), if there is one.  All are decoded side by side, the window in front of a stretch unknown (16-bit
 * symbols), each up to the first block boundary at or behind the NEXT stretch's start.  Then the check that makes the
 * result exact whatever the search believed: a stretch is used only if the decoder of the stretch before it stopped
 * precisely on its first bit -- the concatenation is then what one decoder running through would have produced; at the
 * first stretch that was not reached that way the round ends and the next one starts where the last good decoder
 * stopped.  The symbols become bytes window by window (the 32 KiB tails one after the other, the bulk side by side). */
typedef struct
{
  tjz_source *s;
  int j, n;
  size_t nominal_bit, round_end_bit;
  size_t start_bit;             /* (size_t) -1: no block start found */
  size_t *starts;               /* all stretches' start_bit, complete when the decoding begins */
  int rc;
} tjz_stretch_job;

static void *
tjz_stretch_find (void *arg)
{
  tjz_stretch_job *k = (tjz_stretch_job *) arg;
  if (k->j > 0) k->start_bit = tjp_find_block (k->s->z, k->s->zn, k->nominal_bit, k->nominal_bit + k->s->seg_bytes * 4);   /* (half a stretch) */
  k->starts[k->j] = k->start_bit;
  return NULL;
}

static void *
tjz_stretch_decode (void *arg)
{
  tjz_stretch_job *k = (tjz_stretch_job *) arg;
  size_t stop = k->round_end_bit;
  int i;
  k->rc = -2;
  if (k->start_bit == (size_t) -1) return NULL;
  for (i = k->j + 1; i < k->n; i++) if (k->starts[i] != (size_t) -1) { stop = k->starts[i]; break; }
  k->rc = tjp_decode (k->s->z, k->s->zn, k->start_bit, stop, &k->s->seg[k->j]);
  return NULL;
}

/* jobs 1 .. n - 1 on threads of their own, job 0 here; a thread that cannot be had: its job here as well */
static void
tjz_run_all (void *(*fn) (void *), tjz_stretch_job *job, int n)
{
  pthread_t th[TJF_MAX_THREADS];
  int started[TJF_MAX_THREADS], j;
  for (j = 1; j < n; j++) started[j] = pthread_create (&th[j], NULL, fn, &job[j]) == 0;
  (void) fn (&job[0]);
  for (j = 1; j < n; j++) { if (started[j]) pthread_join (th[j], NULL); else (void) fn (&job[j]); }
}

typedef struct { const unsigned short *sym; size_t n; const unsigned char *win; size_t win_valid; unsigned char *out; int rc; unsigned crc; } tjz_resolve_job;
static void *
tjz_resolve_run (void *arg)
{ /* symbols -> bytes, and the bytes' CRC-32 while they are in the cache (combined in order by the caller) */
  tjz_resolve_job *r = (tjz_resolve_job *) arg;
  size_t i;
  r->rc = 0; r->crc = 0;
  for (i = 0; i < r->n; i += 1u << 20) {
    const size_t m = r->n - i < (1u << 20) ? r->n - i : (1u << 20);
    if (tjp_resolve (r->sym + i, m, r->win, r->win_valid, r->out + i)) r->rc = -1;
    r->crc = tji_crc32 (r->crc, r->out + i, m);
  }
  return NULL;
}

/* 0: pout holds the round's bytes (possibly none, when par_final is set); -1: the stream is damaged at pbit; 1: nothing
 * came of it that a single decoder would not do better (no block start found in any stretch): the caller switches over */
static int
tjz_round (tjz_source *s, unsigned char *direct, size_t direct_room, size_t *direct_got)
{ /* direct: where the caller wants the bytes (a view with direct_room bytes free); the round's first *direct_got bytes go
   * there, whole stretches only, the rest to pout -- no copy for what fits */
  tjz_stretch_job job[TJF_MAX_THREADS];
  tjz_resolve_job res[TJF_MAX_THREADS];
  pthread_t th[TJF_MAX_THREADS];
  int started[TJF_MAX_THREADS], chain[TJF_MAX_THREADS];
  size_t starts[TJF_MAX_THREADS];
  unsigned char (*win)[32768] = NULL;
  size_t win_valid[TJF_MAX_THREADS + 1];
  const size_t base = s->pbit >> 3, end_bit = s->zn * 8;
  size_t total = 0, off;
  int n = s->n_threads, j, nc = 0, rc = 0, n_direct = 0;
  if (n > TJF_MAX_THREADS) n = TJF_MAX_THREADS;
  while (n > 1 && base + (size_t) (n - 1) * s->seg_bytes + 65536 >= s->zn) n--;     /* stretches that would begin at the file's end */
  for (j = 0; j < n; j++) {
    job[j].s = s; job[j].j = j; job[j].n = n; job[j].starts = starts; job[j].rc = -2;
    job[j].nominal_bit = (base + (size_t) j * s->seg_bytes) * 8;
    job[j].round_end_bit = (base + (size_t) n * s->seg_bytes) * 8;
    if (job[j].round_end_bit > end_bit) job[j].round_end_bit = end_bit;
    job[j].start_bit = j ? (size_t) -1 : s->pbit;
  }
  {
    /* first every stretch's start, then -- each one's end being the next one's start -- the decoding */
    const double t_ = tjf_now ();
    tjz_run_all (tjz_stretch_find, job, n);
    tjz_run_all (tjz_stretch_decode, job, n);
    s->t_decode += tjf_now () - t_;
  }
  s->par_rounds++;
  /* the chain of stretches each of which was reached by the decoder of the one before */
  *direct_got = 0;
  if (job[0].rc != 0 && s->seg[0].n == 0) { s->pout_len = s->pout_pos = 0; return 1; }   /* no block decoded at pbit: the single decoder says what is wrong, if anything */
  chain[nc++] = 0;
  for (;;) {
    const int a = chain[nc - 1];
    int b;
    if (job[a].rc != 0 || s->seg[a].is_final) break;   /* (an error: the blocks in front of it are good, the next round meets it again at its start) */
    for (b = a + 1; b < n && job[b].start_bit == (size_t) -1; b++) ;
    if (b >= n) break;
    if (s->seg[a].end_bit != job[b].start_bit) { s->par_false++; break; }
    chain[nc++] = b;
  }
  s->par_stretches += nc;
  for (j = 0; j < nc; j++) total += s->seg[chain[j]].n;
  {
    /* stretches that fit the caller's view whole go there, the others to pout */
    size_t fit = 0, rest;
    for (j = 0; j < nc && fit + s->seg[chain[j]].n <= direct_room; j++) fit += s->seg[chain[j]].n;
    n_direct = j; *direct_got = fit; rest = total - fit;
    if (rest > s->pout_cap) {
      free (s->pout);
      s->pout_cap = rest + (rest >> 2) + 65536;
      s->pout = (unsigned char *) malloc (s->pout_cap);
      if (!s->pout) { s->pout_cap = 0; return -1; }
    }
  }
  /* windows: the bytes in front of every stretch of the chain (the member's output so far for the first) */
  const double t_res0 = tjf_now ();
  win = (unsigned char (*)[32768]) malloc ((size_t) (nc + 1) * 32768);
  if (!win) return -1;
  memset (win[0], 0, 32768);
  memcpy (win[0] + 32768 - s->hist_len, s->hist, s->hist_len);
  win_valid[0] = s->hist_len;
  for (j = 0; j < nc; j++) {
    const tjp_segment *g = &s->seg[chain[j]];
    const size_t tail = g->n < 32768 ? g->n : 32768;
    if (tail < 32768) memcpy (win[j + 1], win[j] + tail, 32768 - tail);          /* (what stays of the window in front) */
    if (tjp_resolve (TJP_SYMBOLS (g) + (g->n - tail), tail, win[j], win_valid[j], win[j + 1] + 32768 - tail)) rc = -1;
    win_valid[j + 1] = win_valid[j] + tail > 32768 ? 32768 : win_valid[j] + tail;
  }
  /* the bulk, side by side */
  off = 0;
  for (j = 0; j < nc; j++) {
    const tjp_segment *g = &s->seg[chain[j]];
    if (j == n_direct) off = 0;                         /* (from here on: pout) */
    res[j].sym = TJP_SYMBOLS (g); res[j].n = g->n; res[j].win = win[j]; res[j].win_valid = win_valid[j];
    res[j].out = (j < n_direct ? direct : s->pout) + off; res[j].rc = 0;
    off += g->n;
    started[j] = 0;
    if (j && pthread_create (&th[j], NULL, tjz_resolve_run, &res[j]) == 0) started[j] = 1;
  }
  for (j = 0; j < nc; j++) { if (!started[j]) tjz_resolve_run (&res[j]); }
  for (j = 1; j < nc; j++) if (started[j]) pthread_join (th[j], NULL);
  for (j = 0; j < nc; j++) if (res[j].rc) rc = -1;      /* a match that reaches in front of the member's first byte */
  /* the member's CRC-32 and size so far: the direct part now, pout's part as it is handed out (crc of the whole of pout is known: kept) */
  for (j = 0; j < n_direct; j++) { s->crc = crc32_combine (s->crc, res[j].crc, (z_off_t) res[j].n); s->isize += res[j].n; }
  s->pout_crc = 0;
  for (j = n_direct; j < nc; j++) s->pout_crc = crc32_combine (s->pout_crc, res[j].crc, (z_off_t) res[j].n);
  if (!rc) {
    const tjp_segment *last = &s->seg[chain[nc - 1]];
    memcpy (s->hist, win[nc], 32768); s->hist_len = win_valid[nc];
    if (s->hist_len < 32768) memmove (s->hist, s->hist + 32768 - s->hist_len, s->hist_len);    /* (hist holds its bytes from index 0) */
    s->pbit = last->end_bit;
    s->par_final = last->is_final;
    s->pout_len = total - *direct_got; s->pout_pos = 0;
    /* nothing but stretch 0 and no other block start found anywhere: a file this does nothing for (not text, or one block) */
    if (!rc && nc == 1 && n > 1 && !last->is_final) { int any = 0; for (j = 1; j < n; j++) any |= job[j].start_bit != (size_t) -1; if (!any) rc = 1; }
  }
  free (win);
  s->t_resolve += tjf_now () - t_res0;
  return rc;
}

/* the next inflated bytes of the file into out[0, cap) (cap >= 64 KiB); 0 only when nothing is left */
static size_t
tjz_fill (tjz_source *s, unsigned char *out, size_t cap)
{
  size_t got = 0;
  while (!s->ended && got < cap) {
    if (s->zpos >= s->zn) { s->ended = 1; if (s->strm_open) s->damaged = 1; break; }   /* (inside a member: truncated) */
    if (s->bgzf && !s->strm_open) {                     /* gather whole blocks that fit, inflate them side by side */
      size_t nb = 0, zp = s->zpos, off = got;
      int nt, t;
      pthread_t th[TJF_MAX_THREADS];
      tjz_job job[TJF_MAX_THREADS];
      int started[TJF_MAX_THREADS];
      for (;;) {
        size_t doff, bsize;
        unsigned isize;
        if (zp >= s->zn) break;
        bsize = tjz_bgzf_block (s->z + zp, s->zn - zp, &doff);
        if (!bsize) { if (!nb) s->bgzf = 0; break; }    /* something else from here on: the general inflater takes it */
        isize = (unsigned) s->z[zp + bsize - 4] | ((unsigned) s->z[zp + bsize - 3] << 8) | ((unsigned) s->z[zp + bsize - 2] << 16) | ((unsigned) s->z[zp + bsize - 1] << 24);
        if (isize > 65536u) { if (!nb) s->bgzf = 0; break; }
        if (off + isize > cap) break;
        if (nb == s->blk_cap) {
          s->blk_cap = s->blk_cap ? 2 * s->blk_cap : 4096;
          s->blk = (tjz_block *) realloc (s->blk, s->blk_cap * sizeof (tjz_block));
        }
        s->blk[nb].data_off = zp + doff; s->blk[nb].data_len = bsize - doff - 8; s->blk[nb].out_off = off; s->blk[nb].isize = isize;
        s->blk[nb].crc = (unsigned) s->z[zp + bsize - 8] | ((unsigned) s->z[zp + bsize - 7] << 8) | ((unsigned) s->z[zp + bsize - 6] << 16) | ((unsigned) s->z[zp + bsize - 5] << 24);
        nb++; off += isize; zp += bsize;
      }
      if (!nb) { if (s->bgzf) break; else continue; }   /* (bgzf still set: the next block does not fit -- the view is full) */
      nt = s->n_threads; if ((size_t) nt > nb) nt = (int) nb;
      for (t = 0; t < nt; t++) {
        job[t].src = s; job[t].out = out; job[t].first = nb * (size_t) t / (size_t) nt; job[t].last = nb * (size_t) (t + 1) / (size_t) nt;
        started[t] = 0;
        if (t) { if (pthread_create (&th[t], NULL, tjz_inflate_blocks, &job[t]) == 0) started[t] = 1; else tjz_inflate_blocks (&job[t]); }
      }
      tjz_inflate_blocks (&job[0]);
      for (t = 1; t < nt; t++) if (started[t]) pthread_join (th[t], NULL);
      tjf_stat_bgzf += (long) nb;
      for (t = 0; t < nt; t++) if (job[t].bad_at_set) {  /* a damaged block: the file ends in front of it */
        const size_t b = job[t].bad_at;
        off = s->blk[b].out_off; s->ended = 1; s->damaged = 1;
        break;
      }
      got = off; s->zpos = zp;
      continue;
    }
    /* any other gzip stream: one inflater, members one after the other (what gzread does: zlib's gzread.c gz_look /
     * gz_decomp -- another member if the next two bytes are 1f 8b, anything else after a member is ignored) */
    if (!s->use_zlib) {
      size_t ip, op = 0;
      int rc;
      if (!s->strm_open) {
        const size_t h = tjz_member_header (s->z + s->zpos, s->zn - s->zpos);
        if (!h) { s->ended = 1; s->damaged = 1; break; }
        if (!s->inf) s->inf = (tji_state *) malloc (sizeof (tji_state));
        if (!s->inf) { s->ended = 1; s->damaged = 1; break; }
        tji_init (s->inf);
        s->zpos += h; s->strm_open = 1; s->crc = crc32 (0L, Z_NULL, 0); s->isize = 0; s->hist_len = 0;
        /* a member worth several stretches: on all threads (tjz_round) */
        s->par_active = s->par && s->zn - s->zpos > 4 * s->seg_bytes;
        s->par_final = 0; s->par_giveup = 0; s->pbit = s->zpos * 8; s->pout_len = s->pout_pos = 0;
      }
      if (s->par_active) {
        if (s->pout_pos < s->pout_len) {                /* a round's bytes, as many as the view takes */
          size_t m = s->pout_len - s->pout_pos;
          const double t_ = tjf_now ();
          if (m > cap - got) m = cap - got;
          memcpy (out + got, s->pout + s->pout_pos, m);
          if (s->pout_pos == 0 && m == s->pout_len) s->crc = crc32_combine (s->crc, s->pout_crc, (z_off_t) m);   /* (the usual case: known already) */
          else s->crc = tjz_crc_extend (s->crc, out + got, m, s->n_threads);
          s->isize += m; s->pout_pos += m; got += m;
          s->t_handout += tjf_now () - t_;
          continue;
        }
        if (s->par_final) {                             /* the member's last byte has been handed out: its trailer */
          s->zpos = (s->pbit + 7) >> 3;
          s->par_active = 0; s->strm_open = 0; s->hist_len = 0;
          rc = TJI_DONE;
          goto member_done;
        }
        {
          size_t dg = 0;
          rc = s->par_giveup ? 1 : tjz_round (s, out + got, cap - got, &dg);
          got += dg;
        }
        if (rc < 0) { s->ended = 1; s->damaged = 1; break; }
        if (rc == 1 && s->pout_pos < s->pout_len) { s->par_giveup = 1; continue; }     /* (what the round did decode goes out first) */
        if (rc == 1) {                                  /* no use here (not text, one huge block, ...): the single decoder from pbit on */
          s->par_active = 0; s->par_giveup = 0;
          tji_init (s->inf);
          s->zpos = s->pbit >> 3;
          if (s->pbit & 7u) { s->inf->bitbuf = (unsigned long long) s->z[s->zpos] >> (s->pbit & 7u); s->inf->bitcnt = 8u - (unsigned) (s->pbit & 7u); s->zpos++; }
          /* (it wants the member's last 32 KiB right in front of what it writes, which tjz_fill arranges at the start of a
           * view only: what this view holds goes out first) */
          if (got) break;
        }
        continue;
      }
      if (got == 0 && s->hist_len) memcpy (out - s->hist_len, s->hist, s->hist_len);   /* (the room is there: feeder views keep >= 64 KiB in front) */
      ip = s->zpos;
      rc = tji_inflate (s->inf, s->z, s->zn, &ip, out + got, cap - got, &op, got == 0 ? s->hist_len : 0);
      s->zpos = ip;
      s->crc = tjz_crc_extend (s->crc, out + got, op, s->n_threads);
      s->isize += op;
      if (rc == TJI_DONE) {
      member_done:
        s->strm_open = 0; s->hist_len = 0;
        if (s->zn - s->zpos < 8) { s->ended = 1; s->damaged = 1; }
        else {
          const unsigned char *t = s->z + s->zpos;
          const unsigned long c = (unsigned long) t[0] | ((unsigned long) t[1] << 8) | ((unsigned long) t[2] << 16) | ((unsigned long) t[3] << 24);
          const unsigned long n = (unsigned long) t[4] | ((unsigned long) t[5] << 8) | ((unsigned long) t[6] << 16) | ((unsigned long) t[7] << 24);
          s->zpos += 8;
          if (c != (s->crc & 0xffffffffUL) || n != (unsigned long) (s->isize & 0xffffffffULL)) { s->ended = 1; s->crc_failed = 1; }
          else if (s->zn - s->zpos < 2 || s->z[s->zpos] != 0x1f || s->z[s->zpos + 1] != 0x8b) s->ended = 1;
          else { size_t d; s->bgzf = tjz_bgzf_block (s->z + s->zpos, s->zn - s->zpos, &d) != 0; }
        }
      }
      else if (rc == TJI_OUTPUT_FULL) {                 /* the view is full: keep the window for the next one */
        if (op >= sizeof s->hist) { memcpy (s->hist, out + got + op - sizeof s->hist, sizeof s->hist); s->hist_len = sizeof s->hist; }
        else {
          const size_t keep = s->hist_len < sizeof s->hist - op ? s->hist_len : sizeof s->hist - op;
          memmove (s->hist, s->hist + s->hist_len - keep, keep);
          memcpy (s->hist + keep, out + got, op);
          s->hist_len = keep + op;
        }
      }
      else { s->ended = 1; s->damaged = 1; }            /* truncated or not DEFLATE */
      got += op;
      continue;
    }
    if (!s->strm_open) {
      memset (&s->strm, 0, sizeof s->strm);
      if (inflateInit2 (&s->strm, 15 + 16) != Z_OK) { s->ended = 1; s->damaged = 1; break; }
      s->strm_open = 1;
    }
    {
      const size_t in_now = (s->zn - s->zpos > (1u << 30)) ? (1u << 30) : s->zn - s->zpos;
      const size_t out_now = (cap - got > (1u << 30)) ? (1u << 30) : cap - got;
      int rc;
      s->strm.next_in = (Bytef *) (s->z + s->zpos); s->strm.avail_in = (uInt) in_now;
      s->strm.next_out = out + got; s->strm.avail_out = (uInt) out_now;
      rc = inflate (&s->strm, Z_NO_FLUSH);
      s->zpos += in_now - s->strm.avail_in;
      got += out_now - s->strm.avail_out;
      if (rc == Z_STREAM_END) {
        inflateEnd (&s->strm); s->strm_open = 0;
        if (s->zn - s->zpos < 2 || s->z[s->zpos] != 0x1f || s->z[s->zpos + 1] != 0x8b) s->ended = 1;
        else { size_t d; s->bgzf = tjz_bgzf_block (s->z + s->zpos, s->zn - s->zpos, &d) != 0; }
      }
      else if (rc == Z_OK) { if (in_now == s->strm.avail_in && out_now == s->strm.avail_out) { s->ended = 1; s->damaged = 1; } }
      else { s->ended = 1; s->damaged = 1; }
    }
  }
  return got;
}

typedef struct { tjz_source *src; unsigned char *out; size_t cap, got; } tjz_fill_job;
static void *tjz_fill_thread (void *arg) { tjz_fill_job *f = (tjz_fill_job *) arg; f->got = tjz_fill (f->src, f->out, f->cap); return NULL; }

/* The buffers of a gzip file -- two views, the round's bytes, the stretches' symbols: some 300 MB -- are kept for the next
 * file instead of going back to the allocator: a sample is read as several files (R1, R2, and the caller runs one thread
 * per sample), and first-touch page faults on fresh buffers were a quarter of a 70 MB file's time.  Up to TJZ_KEEP sets (what more files at once use is freed when they are done). */
#define TJZ_KEEP 4
typedef struct { int used; unsigned char *view[2]; size_t view_cap[2], head[2]; unsigned char *pout; size_t pout_cap; tjp_segment seg[TJF_MAX_THREADS]; } tjz_buffers;
static tjz_buffers tjz_kept[TJZ_KEEP];
static pthread_mutex_t tjz_kept_lock = PTHREAD_MUTEX_INITIALIZER;

static int
tjz_take_buffers (tjz_buffers *b, size_t want_view_cap)
{
  int i, got = 0;
  memset (b, 0, sizeof *b);
  pthread_mutex_lock (&tjz_kept_lock);
  for (i = 0; i < TJZ_KEEP && !got; i++)
    if (tjz_kept[i].used && tjz_kept[i].view_cap[0] == want_view_cap && tjz_kept[i].view_cap[1] == want_view_cap) { *b = tjz_kept[i]; tjz_kept[i].used = 0; got = 1; }
  pthread_mutex_unlock (&tjz_kept_lock);
  return got;
}

static void
tjz_keep_buffers (tjz_buffers *b)
{
  int i, j, kept = 0;
  pthread_mutex_lock (&tjz_kept_lock);
  for (i = 0; i < TJZ_KEEP && !kept; i++) if (!tjz_kept[i].used) { tjz_kept[i] = *b; tjz_kept[i].used = 1; kept = 1; }
  pthread_mutex_unlock (&tjz_kept_lock);
  if (!kept) { free (b->view[0]); free (b->view[1]); free (b->pout); for (j = 0; j < TJF_MAX_THREADS; j++) free (b->seg[j].buf); }
}

long
tjf_parse_gz_file (const char *path, int n_threads, size_t window_bytes, const tjf_sink *sink)
{
  tjz_source src;
  tjf_state s;
  unsigned char *view[2] = {NULL, NULL};
  size_t view_cap[2] = {0, 0}, head[2];                 /* view[i] holds payload at [head[i], head[i] + got) */
  size_t reserve = 1u << 20, payload;
  tjz_fill_job fj;
  tjz_buffers kept;
  pthread_t fth;
  int cur = 0, fill_running = 0, fill_threaded = 0;
  size_t carry_len = 0;
  const double t0 = tjf_now ();
  double t_inflate = 0, tt;
  long total;

  memset (&src, 0, sizeof src);
  src.z = tjf_map (path, &src.zn);
  if (!src.z) return -1;
  if (src.zn < 2 || src.z[0] != 0x1f || src.z[1] != 0x8b) { if (src.zn) munmap ((void *) src.z, src.zn); return -1; }
  if (window_bytes < 65536) window_bytes = 65536;       /* a BGZF block must fit */
  payload = window_bytes;
  if (reserve > payload) reserve = payload;
  { size_t d; src.bgzf = tjz_bgzf_block (src.z, src.zn, &d) != 0; }
  src.n_threads = n_threads < 1 ? 1 : (n_threads > TJF_MAX_THREADS ? TJF_MAX_THREADS : n_threads);
  { const char *e = getenv ("TATAJUBA_AMD_FEEDER_INFLATE"); src.use_zlib = e && !strcmp (e, "zlib"); }
  {
    /* a member that is not BGZF: on all threads, 1 MiB of compressed bytes per stretch (TATAJUBA_AMD_GZ_STRETCH: tests;
     * TATAJUBA_AMD_GZ_PARALLEL=0: one decoder as before) */
    const char *e = getenv ("TATAJUBA_AMD_GZ_STRETCH"), *d = getenv ("TATAJUBA_AMD_GZ_PARALLEL");
    src.seg_bytes = e && atol (e) >= 65536 ? (size_t) atol (e) : ((size_t) 1 << 20);
    src.par = src.n_threads > 1 && !src.use_zlib && !(d && !strcmp (d, "0"));
  }
  tjf_stat_windows = 0; tjf_stat_fallback = 0; tjf_stat_bgzf = 0; tjf_stat_gz_rounds = 0; tjf_stat_gz_stretches = 0; tjf_stat_gz_false = 0;
  if (tjf_state_init (&s, sink, n_threads, window_bytes + reserve)) { tjf_state_free (&s); munmap ((void *) src.z, src.zn); return -2; }
  if (tjz_take_buffers (&kept, reserve + payload)) {
    int j;
    for (cur = 0; cur < 2; cur++) { view[cur] = kept.view[cur]; view_cap[cur] = kept.view_cap[cur]; head[cur] = reserve; }
    src.pout = kept.pout; src.pout_cap = kept.pout_cap;
    for (j = 0; j < TJF_MAX_THREADS; j++) src.seg[j] = kept.seg[j];
  }
  else for (cur = 0; cur < 2; cur++) {
    view_cap[cur] = reserve + payload; head[cur] = reserve;
    view[cur] = (unsigned char *) malloc (view_cap[cur]);
    if (!view[cur]) { s.total_reads = -2; s.done = 1; }
  }
  cur = 0;
  if (!s.done) { fj.src = &src; fj.out = view[0] + head[0]; fj.cap = payload; fj.got = tjz_fill (&src, fj.out, fj.cap); }
  while (!s.done) {
    /* view `cur`: carry_len bytes in front of head[cur] (already in place) + fj.got fresh bytes */
    unsigned char *v = view[cur] + head[cur] - carry_len;
    const size_t n = carry_len + fj.got;
    const int final = src.ended;                        /* (the producer is not running here) */
    const int nxt = cur ^ 1;
    size_t pos;
    if (n == 0) break;
    if (!final) {                                       /* the producer fills the other view while this one is parsed */
      fj.src = &src; fj.out = view[nxt] + head[nxt]; fj.cap = view_cap[nxt] - head[nxt]; fj.got = 0;
      fill_threaded = pthread_create (&fth, NULL, tjz_fill_thread, &fj) == 0;
      fill_running = 1;
    }
    pos = tjf_window (&s, v, n, 0, n, final);
    if (final) break;
    tt = tjf_now ();
    if (fill_threaded) pthread_join (fth, NULL); else tjz_fill_thread (&fj);
    fill_running = 0;
    t_inflate += tjf_now () - tt;
    if (s.done) break;
    carry_len = n - pos;                                /* the cut record: in front of the fresh bytes of the other view */
    if (carry_len > head[nxt]) {                        /* (a record longer than the room in front: a bigger view) */
      const size_t new_head = carry_len + reserve, new_cap = new_head + (view_cap[nxt] - head[nxt]) + carry_len;
      unsigned char *bigger = (unsigned char *) malloc (new_cap);
      if (!bigger) { s.total_reads = -2; break; }
      memcpy (bigger + new_head, view[nxt] + head[nxt], fj.got);
      free (view[nxt]); view[nxt] = bigger; view_cap[nxt] = new_cap; head[nxt] = new_head;
    }
    memcpy (view[nxt] + head[nxt] - carry_len, v + pos, carry_len);
    cur = nxt;
  }
  if (fill_running) { if (fill_threaded) pthread_join (fth, NULL); }
  if (sink->sync) (void) sink->sync (sink->ctx);
  if (src.damaged) fprintf (stderr, "tatajuba_amd: '%s' is truncated or damaged; reading stops where its gzip stream breaks\n", path);
  if (src.strm_open && src.use_zlib) inflateEnd (&src.strm);
  free (src.inf);
  free (src.blk);
  tjf_stat_gz_rounds = src.par_rounds; tjf_stat_gz_stretches = src.par_stretches; tjf_stat_gz_false = src.par_false;
  if (view[0] && view[1] && view_cap[0] == reserve + payload && view_cap[1] == reserve + payload && payload >= ((size_t) 1 << 20)) {
    int j;
    for (cur = 0; cur < 2; cur++) { kept.view[cur] = view[cur]; kept.view_cap[cur] = view_cap[cur]; kept.head[cur] = reserve; }
    kept.pout = src.pout; kept.pout_cap = src.pout_cap;
    for (j = 0; j < TJF_MAX_THREADS; j++) kept.seg[j] = src.seg[j];
    tjz_keep_buffers (&kept);
  }
  else {
    int j;
    free (src.pout);
    for (j = 0; j < TJF_MAX_THREADS; j++) free (src.seg[j].buf);
    free (view[0]); free (view[1]);
  }
  tjf_state_free (&s);
  munmap ((void *) src.z, src.zn);
  tjf_trace (&s, path, tjf_stat_bgzf ? "bgzf" : "gzip", t0, t_inflate);
  if (src.par_rounds && getenv ("TATAJUBA_AMD_FEEDER_TRACE"))
    fprintf (stderr, "[feeder]   one member on %d threads: %ld rounds, %ld stretches used, %ld false starts; search + decode %.1f ms, resolve %.1f ms, hand-out + CRC %.1f ms\n",
             src.n_threads, src.par_rounds, src.par_stretches, src.par_false, src.t_decode * 1e3, src.t_resolve * 1e3, src.t_handout * 1e3);
  total = s.total_reads;
  if (src.crc_failed && total >= 0) total = -4;         /* (what was handed to the sink cannot be trusted) */
  return total;
}
