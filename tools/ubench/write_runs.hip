// micro-benchmark 7: what the write pattern of a hash partition costs -- 256 bucket frontiers, every workgroup writes one run
// of R eight-byte records to each of them per pass (sorted slots -> coalesced stores, as StageSink's copy-out does); runs of
// a bucket lie back to back.  R = 14: runs start and end inside 128-byte lines that another workgroup's run shares (what
// the partition writes); R = 16: whole lines; "14 of 16": line-aligned runs that leave the last two records of their line
// unwritten (partial lines, but nobody else's).  Reports GB/s of records written.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned long long u64;
typedef unsigned u32;
template <int R, int STRIDE>
__global__ __launch_bounds__ (512) void write_runs (u64 *pool, u64 region_words, u32 passes)
{
  constexpr int N = 256 * R;
  for (u32 p = blockIdx.x; p < passes; p += gridDim.x)
    for (int i = threadIdx.x; i < N; i += 512) {
      const u32 b = (u32) i / R, o = (u32) i % R;
      pool[(u64) b * region_words + (u64) p * STRIDE + o] = ((u64) p << 32) | (u32) i | 1u;
    }
}
// the same whole-line runs, every line written in two halves by two store instructions a pass apart (does the L2 put a line
// together before it leaves?): first the first 64 bytes of every line of the pass, then the second 64 bytes
template <int R, int STRIDE>
__global__ __launch_bounds__ (512) void write_runs_split (u64 *pool, u64 region_words, u32 passes)
{
  constexpr int N = 256 * R;
  for (u32 p = blockIdx.x; p < passes; p += gridDim.x)
    for (int half = 0; half < 2; half++)
      for (int i = threadIdx.x; i < N / 2; i += 512) {
        const u32 ii = ((u32) i / 8u) * 16u + (u32) half * 8u + ((u32) i & 7u);      // record ii of the pass: the half-line `half` of line i / 8
        const u32 b = ii / R, o = ii % R;
        pool[(u64) b * region_words + (u64) p * STRIDE + o] = ((u64) p << 32) | ii | 1u;
      }
}
// for reference: the same records read in a line and written in a line (what a partition moves, without the partition)
__global__ __launch_bounds__ (512) void copy_linear (const u64 *__restrict__ src, u64 *__restrict__ dst, u32 passes)
{
  for (u32 p = blockIdx.x; p < passes; p += gridDim.x) {
    u64 v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = src[(u64) p * 4096 + threadIdx.x + 512 * i];
#pragma unroll
    for (int i = 0; i < 8; i++) dst[(u64) p * 4096 + threadIdx.x + 512 * i] = v[i] | 1u;
  }
}
template <int R, int STRIDE>
static void run_split (const char *what, u64 *pool, u64 region_words, u64 records);
__global__ __launch_bounds__ (512) void write_linear (u64 *pool, u32 passes)
{
  for (u32 p = blockIdx.x; p < passes; p += gridDim.x)
    for (int i = threadIdx.x; i < 4096; i += 512) pool[(u64) p * 4096 + i] = ((u64) p << 32) | (u32) i | 1u;
}
static u64 g_pool_words;
template <int R, int STRIDE>
static void run (const char *what, u64 *pool, u64 region_words, u64 records)
{
  const u32 passes = (u32) (records / (256 * R));
  if (255ull * region_words + (u64) passes * STRIDE + R > g_pool_words) { printf ("%s: would write past the pool\n", what); exit (1); }
  hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
  float best = 1e9f;
  for (int it = 0; it < 5; it++) {
    hipEventRecord (e0);
    write_runs<R, STRIDE><<<768, 512>>> (pool, region_words, passes);
    hipEventRecord (e1); hipEventSynchronize (e1);
    float ms; hipEventElapsedTime (&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf ("%-34s R %3d stride %3d: %.3f ms for %.0f M records = %.0f GB/s of records\n", what, R, STRIDE, best, passes * 256.0 * R / 1e6, passes * 256.0 * R * 8 / best / 1e6);
}
template <int R, int STRIDE>
static void run_split (const char *what, u64 *pool, u64 region_words, u64 records)
{
  const u32 passes = (u32) (records / (256 * R));
  if (255ull * region_words + (u64) passes * STRIDE + R > g_pool_words) { printf ("%s: would write past the pool\n", what); exit (1); }
  hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
  float best = 1e9f;
  for (int it = 0; it < 5; it++) {
    hipEventRecord (e0);
    write_runs_split<R, STRIDE><<<768, 512>>> (pool, region_words, passes);
    hipEventRecord (e1); hipEventSynchronize (e1);
    float ms; hipEventElapsedTime (&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf ("%-34s R %3d stride %3d: %.3f ms for %.0f M records = %.0f GB/s of records\n", what, R, STRIDE, best, passes * 256.0 * R / 1e6, passes * 256.0 * R * 8 / best / 1e6);
}
int main ()
{
  /* 256 frontiers, `sp` words apart; region 0 starts on a 128-byte line (hipMalloc), so a run is line-aligned when both sp and
     the stride are multiples of 16 words */
  const u64 records = 60000000ull, region_words = records / 256 * 2 + 4096;
  u64 *pool; if (hipMalloc (&pool, 256 * region_words * 8) != hipSuccess) { printf ("alloc failed\n"); return 1; }
  hipMemset (pool, 0, 256 * region_words * 8);
  g_pool_words = 256 * region_words;
  const u64 al = 21ull * 12288;                         /* 21 chunks of 96 KB: line-aligned regions */
  printf ("-- line-aligned regions (%llu words apart)\n", al);
  run<8, 8> ("runs of 8 = half lines", pool, al, records);
  run<14, 14> ("runs of 14, back to back", pool, al, records);
  run<14, 16> ("14 of every 16 (aligned, partial)", pool, al, records);
  run<16, 16> ("runs of 16 = whole lines", pool, al, records);
  run<24, 24> ("runs of 24 (64-byte aligned)", pool, al, records);
  run<28, 32> ("28 of every 32 (aligned, partial)", pool, al, records);
  run<30, 30> ("runs of 30, back to back", pool, al, records);
  run<32, 32> ("runs of 32 = two whole lines", pool, al, records);
  run<40, 40> ("runs of 40 (64-byte aligned)", pool, al, records);
  run<60, 60> ("runs of 60, back to back", pool, al, records);
  run<64, 64> ("runs of 64", pool, al, records);
  run<128, 128> ("runs of 128", pool, al, records);
  run_split<32, 32> ("runs of 32 on lines, each line in two halves a pass apart", pool, al, records);
  run_split<16, 16> ("runs of 16 on lines, in two halves", pool, al, records);
  printf ("-- regions that start 112 bytes into a line (%llu words apart)\n", al + 14);
  run<16, 16> ("runs of 16, every one across two lines", pool, al + 14, records);
  run<32, 32> ("runs of 32, every one across three lines", pool, al + 14, records);
  run<64, 64> ("runs of 64, across five lines", pool, al + 14, records);
  printf ("-- regions 64 bytes into a line (%llu words apart)\n", al + 8);
  run<32, 32> ("runs of 32, 64-byte aligned", pool, al + 8, records);
  printf ("-- the first version's regions (%llu words apart: 112 bytes into a line, 968 MB in all)\n", region_words);
  run<32, 32> ("runs of 32", pool, region_words, records);
  run<32, 32> ("runs of 32, regions rounded to lines", pool, region_words + 2, records - 4096);
  {                                                     /* for reference: the same bytes written in a line (every workgroup 32 KB at a time) */
    hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
    float best = 1e9f;
    for (int it = 0; it < 5; it++) {
      hipEventRecord (e0);
      write_linear<<<768, 512>>> (pool, (u32) (records / 4096));
      hipEventRecord (e1); hipEventSynchronize (e1);
      float ms; hipEventElapsedTime (&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf ("%-34s               : %.3f ms for %.0f M records = %.0f GB/s of records\n", "in a line, 32 KB per workgroup", best, records / 4096 * 4096 / 1e6, records / 4096 * 4096.0 * 8 / best / 1e6);
  }
  {                                                     /* read in a line + written in a line: the first half of the pool to the second half (60 M records each way fit: 2 x 480 MB of 968) */
    hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
    const u32 passes = (u32) (records / 4096);
    if ((u64) passes * 4096 * 2 > g_pool_words) { printf ("copy: pool too small\n"); return 1; }
    float best = 1e9f;
    for (int it = 0; it < 5; it++) {
      hipEventRecord (e0);
      copy_linear<<<768, 512>>> (pool, pool + (u64) passes * 4096, passes);
      hipEventRecord (e1); hipEventSynchronize (e1);
      float ms; hipEventElapsedTime (&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf ("%-34s               : %.3f ms for %.0f M records = %.0f GB/s read + as much written\n", "copied in a line", best, passes * 4096.0 / 1e6, passes * 4096.0 * 8 / best / 1e6);
  }
  return 0;
}
