#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "feeder.h"
typedef struct { long total, n_reads; unsigned long sum; } msink;
static void *m_alloc (void *c, size_t b) { (void) c; return malloc (b); }
static void m_release (void *c, void *p) { (void) c; free (p); }
static int m_put (void *c, const unsigned char *s, size_t n, long r) { msink *m = c; size_t i; for (i = 0; i < n; i += 997) m->sum += s[i]; m->total += (long) n; m->n_reads += r; return 0; }
int main (int argc, char **argv)
{
  int i;
  for (i = 1; i < argc; i++) {
    size_t w;
    for (w = 65536; w <= (8u << 20); w *= 11) {
      msink m = {0, 0, 0};
      tjf_sink sk = {&m, m_alloc, m_release, m_put, NULL, NULL, NULL};
      long got = tjf_is_plain_file (argv[i]) ? tjf_parse_file (argv[i], 4, w, &sk) : tjf_parse_gz_file (argv[i], 4, w, &sk);
      printf ("%s window %zu: %ld reads, %ld bytes, sum %lu\n", argv[i], w, got, m.total, m.sum);
    }
  }
  return 0;
}
