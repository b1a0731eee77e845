// micro-benchmark 2: cycles per wave-instruction of individual gfx950 VALU encodings at 8 waves/SIMD (diagnostic)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32;
typedef unsigned long long u64;

#define R8(X) X(0,1) X(1,2) X(2,3) X(3,4) X(4,5) X(5,6) X(6,7) X(7,0)

#define DEFK(NAME, ASM) \
__global__ void NAME (u32 *out, int iters, u32 seed, u64 *clk) { \
  u32 a[8]; for (int i = 0; i < 8; i++) a[i] = threadIdx.x * (2 * i + 3) + seed; \
  const u32 c = seed | 0x01010101u; \
  u64 t0 = __builtin_amdgcn_s_memtime (), r0 = __builtin_amdgcn_s_memrealtime (); \
  for (int i = 0; i < iters; i++) { \
    _Pragma ("unroll") for (int u = 0; u < 8; u++) { \
      asm volatile (ASM : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "s"(c), "v"(c) : "vcc", "s20"); } } \
  u64 t1 = __builtin_amdgcn_s_memtime (), r1 = __builtin_amdgcn_s_memrealtime (); \
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; } \
  u32 r = a[0] ^ a[1] ^ a[2] ^ a[3] ^ a[4] ^ a[5] ^ a[6] ^ a[7]; if (r == 0x12345u) out[threadIdx.x] = r; }

// %0..%7 = a0..a7, %8 = sgpr const, %9 = vgpr const
DEFK (k_allfast, "v_xor_b32 %0, %0, %1\n v_bitop3_b32 %1, %1, %2, %9 bitop3:0x96\n v_lshrrev_b32 %2, 1, %3\n v_and_b32 %3, 0x07070707, %4\n v_xor_b32 %4, %4, %5\n v_bitop3_b32 %5, %5, %6, %9 bitop3:0x96\n v_lshrrev_b32 %6, 1, %7\n v_and_b32 %7, 0x07070707, %0")
DEFK (k_allslow, "v_perm_b32 %0, %0, %1, %9\n v_dot4_u32_u8 %1, %1, %8, %2\n v_alignbit_b32 %2, %2, %3, 24\n v_perm_b32 %3, %3, %4, %9\n v_dot4_u32_u8 %4, %4, %8, %5\n v_alignbit_b32 %5, %5, %6, 24\n v_perm_b32 %6, %6, %7, %9\n v_dot4_u32_u8 %7, %7, %8, %0")
DEFK (k_alt_fs, "v_xor_b32 %0, %0, %1\n v_perm_b32 %1, %1, %2, %9\n v_bitop3_b32 %2, %2, %3, %9 bitop3:0x96\n v_dot4_u32_u8 %3, %3, %8, %4\n v_lshrrev_b32 %4, 1, %5\n v_alignbit_b32 %5, %5, %6, 24\n v_and_b32 %6, 0x07070707, %7\n v_perm_b32 %7, %7, %0, %9")
DEFK (k_ffss, "v_xor_b32 %0, %0, %1\n v_bitop3_b32 %1, %1, %2, %9 bitop3:0x96\n v_perm_b32 %2, %2, %3, %9\n v_dot4_u32_u8 %3, %3, %8, %4\n v_lshrrev_b32 %4, 1, %5\n v_and_b32 %5, 0x07070707, %6\n v_alignbit_b32 %6, %6, %7, 24\n v_perm_b32 %7, %7, %0, %9")
DEFK (k_ffffssss, "v_xor_b32 %0, %0, %1\n v_bitop3_b32 %1, %1, %2, %9 bitop3:0x96\n v_lshrrev_b32 %2, 1, %3\n v_and_b32 %3, 0x07070707, %4\n v_perm_b32 %4, %4, %5, %9\n v_dot4_u32_u8 %5, %5, %8, %6\n v_alignbit_b32 %6, %6, %7, 24\n v_perm_b32 %7, %7, %0, %9")
DEFK (k_fffs, "v_xor_b32 %0, %0, %1\n v_bitop3_b32 %1, %1, %2, %9 bitop3:0x96\n v_lshrrev_b32 %2, 1, %3\n v_perm_b32 %3, %3, %4, %9\n v_and_b32 %4, 0x07070707, %5\n v_xor_b32 %5, %5, %6\n v_bitop3_b32 %6, %6, %7, %9 bitop3:0x96\n v_dot4_u32_u8 %7, %7, %8, %0")
DEFK (k_fast_salu, "v_xor_b32 %0, %0, %1\n s_add_u32 s20, s20, 1\n v_bitop3_b32 %2, %2, %3, %9 bitop3:0x96\n s_add_u32 s20, s20, 1\n v_lshrrev_b32 %4, 1, %5\n s_add_u32 s20, s20, 1\n v_and_b32 %6, 0x07070707, %7\n s_add_u32 s20, s20, 1")
DEFK (k_slow_salu, "v_perm_b32 %0, %0, %1, %9\n s_add_u32 s20, s20, 1\n v_dot4_u32_u8 %2, %2, %8, %3\n s_add_u32 s20, s20, 1\n v_alignbit_b32 %4, %4, %5, 24\n s_add_u32 s20, s20, 1\n v_perm_b32 %6, %6, %7, %9\n s_add_u32 s20, s20, 1")
DEFK (k_p1like, "v_and_b32 %0, 0x07070707, %1\n v_perm_b32 %1, %1, %2, %9\n v_bitop3_b32 %2, %2, %3, %9 bitop3:0x96\n v_lshrrev_b32 %3, 1, %4\n v_bitop3_b32 %4, %4, %5, %9 bitop3:0x96\n v_and_b32 %5, 0x07070707, %6\n v_alignbit_b32 %6, %6, %7, 24\n v_bitop3_b32 %7, %7, %0, %9 bitop3:0x96")

typedef void (*kfn) (u32 *, int, u32, u64 *);
static int g_blocks = 2, g_threads = 1024;
static void run (const char *name, kfn f, u32 *d, u64 *clk)
{
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
  f<<<256 * g_blocks, g_threads>>> (d, 10, 1, clk);
  hipDeviceSynchronize ();
  hipEventRecord (e0);
  f<<<256 * g_blocks, g_threads>>> (d, iters, 1, clk);
  hipEventRecord (e1); hipEventSynchronize (e1);
  float ms; hipEventElapsedTime (&ms, e0, e1);
  u64 h[2]; hipMemcpy (h, clk, 16, hipMemcpyDeviceToHost);
  const double n_per_simd = (double) iters * 64 * (g_blocks * g_threads / 256.0);         // wave-instructions per SIMD (8 waves)
  const double ghz = (double) h[0] / ((double) h[1] * 10.0);   // memtime ticks per ns (memrealtime = 100 MHz)
  fflush (stdout); printf ("%-12s %8.3f ms  %6.3f ns/instr  clock %.2f GHz  -> %.2f cycles/instr (in-kernel: %.2f)\n", name, ms, ms * 1e6 / n_per_simd, ghz,
          ms * 1e6 / n_per_simd * ghz, (double) h[0] / n_per_simd);
}

int main ()
{
  u32 *d; u64 *clk; hipMalloc (&d, 4096); hipMalloc (&clk, 64);
  for (int cfg = 0; cfg < 3; cfg++) { g_blocks = cfg == 0 ? 1 : cfg == 1 ? 3 : 2; g_threads = cfg == 0 ? 512 : cfg == 1 ? 512 : 1024; printf ("--- %d waves per SIMD\n", g_blocks * g_threads / 256);
#define RUN(K) run (#K, K, d, clk)
  RUN (k_allfast); RUN (k_allslow); RUN (k_alt_fs); RUN (k_ffss); RUN (k_ffffssss); RUN (k_fffs); RUN (k_fast_salu); RUN (k_slow_salu); RUN (k_p1like);
  }
  return 0;
}
