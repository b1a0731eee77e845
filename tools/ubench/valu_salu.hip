// micro-benchmark 5: does scalar / LDS work of a wave slow down its vector instructions?  (diagnostic)
// Each kernel runs 8 vector instructions per unrolled step plus a number of scalar ones; cycles are per VECTOR instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32;
typedef unsigned long long u64;

#define DEFK(NAME, ASM) \
__global__ void NAME (u32 *out, int iters, u32 seed, u64 *clk) { \
  __shared__ u32 lds[1024]; lds[threadIdx.x & 1023] = seed; __syncthreads (); \
  u32 a[8]; for (int i = 0; i < 8; i++) a[i] = threadIdx.x * (2 * i + 3) + seed; \
  u32 s0 = seed, s1 = seed + 1, s2 = seed + 2, s3 = seed + 3; \
  const u32 c = seed | 0x01010101u; u32 la = (threadIdx.x & 255) * 4, lv = 0; \
  u64 t0 = __builtin_amdgcn_s_memtime (), r0 = __builtin_amdgcn_s_memrealtime (); \
  for (int i = 0; i < iters; i++) { \
    _Pragma ("unroll") for (int u = 0; u < 8; u++) { \
      asm volatile (ASM : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+v"(lv) : "v"(c), "v"(la) : "scc", "memory"); } } \
  u64 t1 = __builtin_amdgcn_s_memtime (), r1 = __builtin_amdgcn_s_memrealtime (); \
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; } \
  u32 r = a[0] ^ a[1] ^ a[2] ^ a[3] ^ a[4] ^ a[5] ^ a[6] ^ a[7] ^ s0 ^ s1 ^ s2 ^ s3 ^ lv; if (r == 0x12345u) out[threadIdx.x] = r; }

// %0..%7 vgprs, %8..%11 sgprs, %12 lds value, %13 vgpr const, %14 lds address
#define V8F "v_xor_b32 %0, %0, %1\n v_bitop3_b32 %1, %1, %2, %13 bitop3:0x96\n v_lshrrev_b32 %2, 1, %3\n v_and_b32 %3, 0x07070707, %4\n v_xor_b32 %4, %4, %5\n v_bitop3_b32 %5, %5, %6, %13 bitop3:0x96\n v_lshrrev_b32 %6, 1, %7\n v_and_b32 %7, 0x07070707, %0\n"
#define V8M "v_xor_b32 %0, %0, %1\n v_perm_b32 %1, %1, %2, %13\n v_lshrrev_b32 %2, 1, %3\n v_alignbit_b32 %3, %3, %4, 24\n v_xor_b32 %4, %4, %5\n v_bitop3_b32 %5, %5, %6, %13 bitop3:0x96\n v_perm_b32 %6, %6, %7, %13\n v_and_b32 %7, 0x07070707, %0\n"
#define S4 "s_add_u32 %8, %8, 1\n s_xor_b32 %9, %9, %8\n s_add_u32 %10, %10, 3\n s_and_b32 %11, %11, %10\n"
DEFK (k_v8, V8F)
DEFK (k_v8_s4, V8F S4)
DEFK (k_v8_s8, V8F S4 S4)
DEFK (k_v8_s16, V8F S4 S4 S4 S4)
DEFK (k_m8, V8M)
DEFK (k_m8_s4, V8M S4)
DEFK (k_m8_s8, V8M S4 S4)
DEFK (k_m8_s16, V8M S4 S4 S4 S4)
DEFK (k_v8_inter, "v_xor_b32 %0, %0, %1\n s_add_u32 %8, %8, 1\n v_bitop3_b32 %1, %1, %2, %13 bitop3:0x96\n s_xor_b32 %9, %9, %8\n v_lshrrev_b32 %2, 1, %3\n s_add_u32 %10, %10, 3\n v_and_b32 %3, 0x07070707, %4\n s_and_b32 %11, %11, %10\n v_xor_b32 %4, %4, %5\n s_add_u32 %8, %8, 1\n v_bitop3_b32 %5, %5, %6, %13 bitop3:0x96\n s_xor_b32 %9, %9, %8\n v_lshrrev_b32 %6, 1, %7\n s_add_u32 %10, %10, 3\n v_and_b32 %7, 0x07070707, %0\n s_and_b32 %11, %11, %10\n")
DEFK (k_v8_nop4, V8F "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n")
DEFK (k_v8_lds1, V8F "ds_read_b32 %12, %14\n")
DEFK (k_v8_lds2w, V8F "ds_read_b32 %12, %14\n s_waitcnt lgkmcnt(0)\n v_xor_b32 %0, %0, %12\n")
DEFK (k_v8_br, V8F "s_cmp_eq_u32 %8, 0x12345\n s_cbranch_scc1 1f\n s_add_u32 %8, %8, 1\n 1:\n")
DEFK (k_v8_exec, V8F "v_cmp_ne_u32 vcc, 0x12345, %0\n s_and_saveexec_b64 s[20:21], vcc\n v_add_u32 %1, 1, %1\n s_or_b64 exec, exec, s[20:21]\n")

typedef void (*kfn) (u32 *, int, u32, u64 *);
static int g_blocks = 2, g_threads = 512;
static void run (const char *name, kfn f, u32 *d, u64 *clk)
{
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
  f<<<256 * g_blocks, g_threads>>> (d, 10, 1, clk);
  if (hipDeviceSynchronize () != hipSuccess) { printf ("%s: launch failed\n", name); return; }
  hipEventRecord (e0);
  f<<<256 * g_blocks, g_threads>>> (d, iters, 1, clk);
  hipEventRecord (e1); hipEventSynchronize (e1);
  float ms; hipEventElapsedTime (&ms, e0, e1);
  u64 h[2]; hipMemcpy (h, clk, 16, hipMemcpyDeviceToHost);
  const double n_per_simd = (double) iters * 64 * (g_blocks * g_threads / 256.0);         // vector wave-instructions per SIMD
  const double ghz = (double) h[0] / ((double) h[1] * 10.0);
  fflush (stdout); printf ("%-12s %8.3f ms  clock %.2f GHz  -> %.2f cycles per vector instruction\n", name, ms, ghz, ms * 1e6 / n_per_simd * ghz);
}

int main ()
{
  u32 *d; u64 *clk; hipMalloc (&d, 4096); hipMalloc (&clk, 64);
  for (int cfg = 0; cfg < 3; cfg++) { g_blocks = cfg == 0 ? 1 : cfg == 1 ? 2 : 4; printf ("--- %d waves per SIMD\n", g_blocks * g_threads / 256);
#define RUN(K) run (#K, K, d, clk)
  RUN (k_v8); RUN (k_v8_s4); RUN (k_v8_s8); RUN (k_v8_s16); RUN (k_v8_inter); RUN (k_m8); RUN (k_m8_s4); RUN (k_m8_s8); RUN (k_m8_s16);
  RUN (k_v8_nop4); RUN (k_v8_lds1); RUN (k_v8_lds2w); RUN (k_v8_br); RUN (k_v8_exec);
  }
  return 0;
}
