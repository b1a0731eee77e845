/* synth.c -- deterministic synthetic inputs for parity tests and the benchmark: a random genome and
 * error-free reads sampled from both strands, written in the device path's batch format (each read followed by '\n').
 * Reads come from a genome (not i.i.d. bases) so that tracts recur at sequencing depth and survive the strand filter. */
#include "../../include/tatajuba_amd.h"
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

static inline uint64_t
splitmix64 (uint64_t x)
{
  x += 0x9E3779B97F4A7C15ULL;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
  return x ^ (x >> 31);
}

static unsigned char *
make_genome (uint64_t seed, uint64_t variant_seed, long g, long *g_out)
{
  static const char dna[4] = {'A', 'C', 'G', 'T'};
  unsigned char *gen = (unsigned char *) malloc ((size_t) g + 64);
  long i;
  for (i = 0; i < g; i += 32) {
    uint64_t w = splitmix64 (seed + (uint64_t) (i >> 5));
    int j;
    for (j = 0; j < 32 && i + j < g; j++) gen[i + j] = (unsigned char) dna[(w >> (2 * j)) & 3];
  }
  if (variant_seed) {                        /* 1% of the tracts >= 4 lose or gain one base */
    unsigned char *v = (unsigned char *) malloc ((size_t) g + g / 64 + 64);
    long o = 0, run = 0, s = 0;
    for (i = 0; i <= g; i++) {
      if (i < g && i > s && gen[i] == gen[s]) continue;
      if (i > s) {                           /* run [s, i) */
        long n = i - s, j;
        if (n >= 4) {
          uint64_t h = splitmix64 (variant_seed ^ (0xD1B54A32D192ED03ULL * (uint64_t) (++run)));
          if (h % 100 == 0) n += ((h >> 32) & 1) ? 1 : -1;
        }
        for (j = 0; j < n; j++) v[o++] = gen[s];
      }
      s = i;
    }
    free (gen);
    gen = v; g = o;
  }
  *g_out = g;
  return gen;
}

typedef struct
{
  const unsigned char *gen; long g;
  uint64_t seed_reads; long r0, r1; int len_min, len_max;
  const long *offset;        /* NULL for fixed length */
  unsigned char *out;
} synth_job;

static inline int
read_length (uint64_t seed, long r, int len_min, int len_max)
{
  if (len_max <= len_min) return len_min;
  return len_min + (int) (splitmix64 (seed ^ 0xA5A5A5A5ULL ^ ((uint64_t) r * 3 + 2)) % (uint64_t) (len_max - len_min + 1));
}

static void *
synth_worker (void *arg)
{
  synth_job *jb = (synth_job *) arg;
  long r;
  for (r = jb->r0; r < jb->r1; r++) {
    int L = read_length (jb->seed_reads, r, jb->len_min, jb->len_max), j;
    uint64_t a = splitmix64 (jb->seed_reads + (uint64_t) r * 3), b = splitmix64 (jb->seed_reads + (uint64_t) r * 3 + 1);
    long start = (long) (a % (uint64_t) (jb->g - L + 1));
    unsigned char *o = jb->out + (jb->offset ? jb->offset[r] : r * (long) (jb->len_min + 1));
    const unsigned char *src = jb->gen + start;
    if (b & 1) {
      for (j = 0; j < L; j++) {
        unsigned char c = src[L - 1 - j];
        o[j] = (unsigned char) (c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A');
      }
    }
    else memcpy (o, src, (size_t) L);
    o[L] = '\n';
  }
  return NULL;
}

long
tjamd_synth_stream (uint64_t seed_genome, uint64_t seed_reads, uint64_t variant_seed, long genome_len,
                    long n_reads, int read_len, int read_len_max, unsigned char *out, long capacity, int n_threads)
{
  long g = 0, total, r, *offset = NULL;
  unsigned char *gen;
  pthread_t th[64];
  synth_job jb[64];
  int t;

  if (n_reads < 0 || read_len < 1 || genome_len < 64) return 0;
  if (read_len_max > read_len) {
    offset = (long *) malloc ((size_t) (n_reads + 1) * sizeof (long));
    for (total = 0, r = 0; r < n_reads; r++) { offset[r] = total; total += read_length (seed_reads, r, read_len, read_len_max) + 1; }
    offset[n_reads] = total;
  }
  else total = n_reads * (long) (read_len + 1);
  if (total > capacity || !out) { free (offset); return -total; }

  gen = make_genome (seed_genome, variant_seed, genome_len, &g);
  if (g < (read_len_max > read_len ? read_len_max : read_len)) { free (gen); free (offset); return 0; }
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 64) n_threads = 64;
  for (t = 0; t < n_threads; t++) {
    jb[t].gen = gen; jb[t].g = g; jb[t].seed_reads = seed_reads; jb[t].len_min = read_len; jb[t].len_max = read_len_max;
    jb[t].offset = offset; jb[t].out = out;
    jb[t].r0 = n_reads * t / n_threads; jb[t].r1 = n_reads * (t + 1) / n_threads;
    if (n_threads == 1) synth_worker (&jb[t]);
    else pthread_create (&th[t], NULL, synth_worker, &jb[t]);
  }
  if (n_threads > 1) for (t = 0; t < n_threads; t++) pthread_join (th[t], NULL);
  free (gen); free (offset);
  return total;
}
