#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include "tj_inflate.h"
static unsigned long long rs = 88172645463325252ull;
static unsigned rnd (void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (unsigned) (rs >> 16); }
int main (void)
{
  tji_state *st = malloc (sizeof *st);
  int it, bad = 0;
  for (it = 0; it < 3000; it++) {
    size_t n = (size_t[]){0, 1, 3, 100, 5000, 70000, 400000}[rnd () % 7], i, cn, cap;
    unsigned char *data = malloc (n + 1), *comp, *exact_in, *out, *buf;
    int kind = rnd () % 4, level = rnd () % 10, strat = (int[]){Z_DEFAULT_STRATEGY, Z_FILTERED, Z_HUFFMAN_ONLY, Z_RLE, Z_FIXED}[rnd () % 5];
    for (i = 0; i < n; i++) data[i] = kind == 0 ? (unsigned char) rnd () : kind == 1 ? "ACGT"[rnd () & 3] : kind == 2 ? (unsigned char) ("ab"[rnd () % 2] + (rnd () % 50 == 0)) : (unsigned char) (i / 300);
    cap = n + n / 2 + 1024; comp = malloc (cap);
    z_stream zs; memset (&zs, 0, sizeof zs);
    deflateInit2 (&zs, level, Z_DEFLATED, -15, 1 + rnd () % 9, strat);
    zs.next_in = data; zs.avail_in = n; zs.next_out = comp; zs.avail_out = cap;
    deflate (&zs, Z_FINISH); cn = zs.total_out; deflateEnd (&zs);
    /* exact-size input buffer (ASan sees any over-read), 8 trailer bytes as in gzip */
    exact_in = malloc (cn + 8); memcpy (exact_in, comp, cn); memset (exact_in + cn, 0x5a, 8);
    {
      size_t chunk = (size_t[]){1, 5, 300, 4096, 100000, 1u << 20}[rnd () % 6], ip = 0, total = 0, hist = 0;
      int rc;
      out = malloc (n + 1); buf = malloc (32768 + chunk);
      tji_init (st);
      for (;;) {
        size_t op = 0;
        rc = tji_inflate (st, exact_in, cn + 8, &ip, buf + 32768, chunk, &op, hist);
        if (total + op > n) { rc = -9; break; }
        memcpy (out + total, buf + 32768, op); total += op;
        if (rc != TJI_OUTPUT_FULL) break;
        { size_t have = hist + op, keep = have < 32768 ? have : 32768; memmove (buf + 32768 - keep, buf + 32768 + op - keep, keep); hist = keep; }
      }
      if (rc != TJI_DONE || total != n || memcmp (out, data, n) || ip != cn) { printf ("FAIL it %d n %zu kind %d level %d strat %d chunk %zu rc %d total %zu ip %zu cn %zu\n", it, n, kind, level, strat, chunk, rc, total, ip, cn); bad++; }
      /* corrupt a byte / truncate: must not crash or over-run */
      if (cn > 2) {
        size_t cut = rnd () % cn, ip2 = 0, op2 = 0;
        unsigned char *t = malloc (cut ? cut : 1); memcpy (t, comp, cut);
        tji_init (st); tji_inflate (st, t, cut, &ip2, buf + 32768, chunk, &op2, 0); free (t);
        exact_in[rnd () % cn] ^= (unsigned char) (1u << (rnd () % 8));
        ip2 = 0; tji_init (st);
        for (int k = 0; k < 2000; k++) { op2 = 0; if (tji_inflate (st, exact_in, cn + 8, &ip2, buf + 32768, chunk, &op2, 0) != TJI_OUTPUT_FULL) break; }
      }
      free (out); free (buf);
    }
    free (data); free (comp); free (exact_in);
  }
  printf ("done, %d failures\n", bad);
  return bad != 0;
}
