// micro-benchmark 6: what a host round trip costs -- (a) tiny kernel + 64-byte hipMemcpyAsync D2H + hipStreamSynchronize,
// (b) tiny kernel that writes its result and a sequence number into pinned host memory + host spin on it (diagnostic)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <time.h>
static double now (void) { struct timespec t; clock_gettime (CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
__global__ void work (unsigned *d, int spin) { unsigned x = threadIdx.x; for (int i = 0; i < spin; i++) x = x * 1664525u + 1013904223u; if (threadIdx.x == 0) d[0] = x | 1u; }
__global__ void publish (const unsigned *d, volatile unsigned *h, unsigned seq) { if (threadIdx.x < 15) h[threadIdx.x] = d[threadIdx.x]; __threadfence_system (); if (threadIdx.x == 0) h[15] = seq; }
int main ()
{
  unsigned *d, *hpin, *hmap; hipStream_t s;
  hipStreamCreate (&s); hipMalloc (&d, 256); hipHostMalloc (&hpin, 256); hipHostMalloc (&hmap, 256, hipHostMallocMapped | hipHostMallocCoherent);
  hipMemset (d, 0, 256); hmap[15] = 0;
  for (int spin = 0; spin <= 200000; spin += 100000) {
    const int N = 300;
    double ta = 0, tb = 0;
    for (int it = 0; it < N + 20; it++) {
      double t0 = now ();
      work<<<1, 64, 0, s>>> (d, spin);
      hipMemcpyAsync (hpin, d, 64, hipMemcpyDeviceToHost, s);
      hipStreamSynchronize (s);
      double t1 = now ();
      if (it >= 20) ta += t1 - t0;
    }
    for (int it = 0; it < N + 20; it++) {
      const unsigned seq = (unsigned) (it + 1 + spin);
      double t0 = now ();
      work<<<1, 64, 0, s>>> (d, spin);
      publish<<<1, 64, 0, s>>> (d, hmap, seq);
      while (__atomic_load_n (&hmap[15], __ATOMIC_ACQUIRE) != seq) __builtin_ia32_pause ();
      double t1 = now ();
      if (it >= 20) tb += t1 - t0;
    }
    printf ("kernel spin %6d: memcpy+sync %.1f us   publish+host spin %.1f us per round trip\n", spin, ta / N * 1e6, tb / N * 1e6);
  }
  hipStreamSynchronize (s);
  return 0;
}
