/* Sanitizer harness (ASan + UBSan) for the mid-stream entry into DEFLATE (tj_inflate.c: tjp_find_block / tjp_decode /
 * tjp_resolve): random texts and byte soups, every zlib level, strategy and flush pattern, random stretch sizes, the
 * compressed bytes in an exact-size buffer (any over-read shows); the stretches are chained the way feeder.c chains them
 * (a stretch counts only if the decoder in front stopped on its first bit) and the result must be the input.  Then the same
 * functions on corrupted and truncated streams: any answer but a crash or an over-run will do. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include "tj_inflate.h"
static unsigned long long rs = 0x9E3779B97F4A7C15ull;
static unsigned rnd (void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (unsigned) (rs >> 16); }

static size_t
make_text (unsigned char *t, size_t n, int kind)
{
  size_t i = 0;
  if (kind == 0) {                                      /* FASTQ */
    unsigned r = 0;
    while (i + 400 < n) {
      int j, L = 50 + (int) (rnd () % 200);
      i += (size_t) sprintf ((char *) t + i, "@r%u/1\n", r++);
      for (j = 0; j < L; j++) t[i++] = (unsigned char) "ACGTN"[rnd () % (rnd () % 50 ? 4 : 5)];
      t[i++] = '\n'; t[i++] = '+'; t[i++] = '\n';
      for (j = 0; j < L; j++) t[i++] = (unsigned char) "FFFF:FF,#"[rnd () % 9];
      t[i++] = '\n';
    }
    return i;
  }
  for (i = 0; i < n; i++) t[i] = kind == 1 ? (unsigned char) rnd () : kind == 2 ? (unsigned char) "ACGT\n"[rnd () % 5] : (unsigned char) ('a' + (i / 700) % 20);
  return n;
}

static int
pipeline (const unsigned char *z, size_t zn, size_t seg_bytes, const unsigned char *want, size_t n_want, int must_match)
{ /* returns 0 if the chained result is `want` (or, for damaged input, if nothing crashed) */
  size_t nseg = (zn + seg_bytes - 1) / seg_bytes, j, op = 0;
  size_t *start = (size_t *) malloc ((nseg + 1) * sizeof (size_t));
  unsigned char *out = (unsigned char *) malloc (n_want + 1), win[32768];
  int rc = 0, guard = 0;
  start[0] = 0;
  for (j = 1; j < nseg; j++) start[j] = tjp_find_block (z, zn, j * seg_bytes * 8, (j * seg_bytes + seg_bytes / 2) * 8);
  j = 0;
  while (j < nseg && guard++ < 100000) {
    size_t jn = j + 1, stop, wv;
    tjp_segment sg;
    memset (&sg, 0, sizeof sg);
    while (jn < nseg && start[jn] == (size_t) -1) jn++;
    stop = jn < nseg ? start[jn] : zn * 8;
    if (tjp_decode (z, zn, start[j], stop, &sg)) { free (sg.buf); rc = 1; break; }
    if (jn < nseg && sg.end_bit != stop && !sg.is_final) { start[jn] = (size_t) -1; free (sg.buf); continue; }   /* not a block start after all */
    if (op + sg.n > n_want) { free (sg.buf); rc = 2; break; }
    wv = op < 32768 ? op : 32768;
    memset (win, 0, sizeof win);
    memcpy (win + 32768 - wv, out + op - wv, wv);
    if (tjp_resolve (TJP_SYMBOLS (&sg), sg.n, win, wv, out + op)) { free (sg.buf); rc = 3; break; }
    op += sg.n;
    if (sg.is_final) { free (sg.buf); break; }
    free (sg.buf);
    j = jn;
  }
  if (!rc && must_match && (op != n_want || memcmp (out, want, n_want))) rc = 4;
  free (out); free (start);
  return must_match ? rc : 0;
}

int main (void)
{
  int it, bad = 0;
  for (it = 0; it < 400; it++) {
    const size_t n0 = (size_t[]){3000, 70000, 400000, 3000000}[rnd () % 4];
    const int kind = rnd () % 4, level = rnd () % 10, strat = (int[]){Z_DEFAULT_STRATEGY, Z_FILTERED, Z_HUFFMAN_ONLY, Z_RLE, Z_FIXED}[rnd () % 5];
    const size_t seg = (size_t[]){4096, 20000, 65536, 300000}[rnd () % 4];
    unsigned char *txt = (unsigned char *) malloc (n0 + 512), *comp, *exact;
    const size_t n = make_text (txt, n0, kind);
    size_t cap = n + n / 2 + 4096, cn, fed = 0;
    z_stream zs;
    comp = (unsigned char *) malloc (cap);
    memset (&zs, 0, sizeof zs);
    deflateInit2 (&zs, level, Z_DEFLATED, -15, 1 + rnd () % 9, strat);
    zs.next_out = comp; zs.avail_out = (uInt) cap;
    while (fed < n) {                                   /* a flush now and then: empty stored blocks, byte-aligned boundaries */
      size_t piece = 1 + rnd () % (n / 3 + 1);
      if (piece > n - fed) piece = n - fed;
      zs.next_in = txt + fed; zs.avail_in = (uInt) piece;
      deflate (&zs, (int[]){Z_NO_FLUSH, Z_NO_FLUSH, Z_SYNC_FLUSH, Z_FULL_FLUSH, Z_BLOCK}[rnd () % 5]);
      fed += piece;
    }
    deflate (&zs, Z_FINISH); cn = zs.total_out; deflateEnd (&zs);
    exact = (unsigned char *) malloc (cn ? cn : 1); memcpy (exact, comp, cn);
    {
      const int rc = pipeline (exact, cn, seg, txt, n, 1);
      if (rc) { printf ("FAIL it %d n %zu kind %d level %d strat %d seg %zu rc %d\n", it, n, kind, level, strat, seg, rc); bad++; }
    }
    if (cn > 16) {                                      /* damaged: must not crash */
      size_t k;
      for (k = 0; k < 3; k++) exact[rnd () % cn] ^= (unsigned char) (1u << (rnd () % 8));
      (void) pipeline (exact, cn - rnd () % (cn / 2), seg, txt, n, 0);
    }
    free (exact); free (comp); free (txt);
  }
  printf ("tjp_fuzz: 400 streams, %d failures\n", bad);
  return bad != 0;
}
