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
      asm volatile (ASM : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "s"(c), "v"(c)); } } \
  u64 t1 = __builtin_amdgcn_s_memtime (), r1 = __builtin_amdgcn_s_memrealtime (); \
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; } \
  u32 r = a[0] ^ a[1] ^ a[2] ^ a[3] ^ a[4] ^ a[5] ^ a[6] ^ a[7]; if (r == 0x12345u) out[threadIdx.x] = r; }

// %0..%7 = a0..a7, %8 = sgpr const, %9 = vgpr const
DEFK (k_xor_vv,   "v_xor_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_xor_b32 %3, %3, %4\n v_xor_b32 %4, %4, %5\n v_xor_b32 %5, %5, %6\n v_xor_b32 %6, %6, %7\n v_xor_b32 %7, %7, %0")
DEFK (k_xor_sv,   "v_xor_b32 %0, %8, %0\n v_xor_b32 %1, %8, %1\n v_xor_b32 %2, %8, %2\n v_xor_b32 %3, %8, %3\n v_xor_b32 %4, %8, %4\n v_xor_b32 %5, %8, %5\n v_xor_b32 %6, %8, %6\n v_xor_b32 %7, %8, %7")
DEFK (k_xor_lit,  "v_xor_b32 %0, 0x12345678, %0\n v_xor_b32 %1, 0x12345678, %1\n v_xor_b32 %2, 0x12345678, %2\n v_xor_b32 %3, 0x12345678, %3\n v_xor_b32 %4, 0x12345678, %4\n v_xor_b32 %5, 0x12345678, %5\n v_xor_b32 %6, 0x12345678, %6\n v_xor_b32 %7, 0x12345678, %7")
DEFK (k_add_vv,   "v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %4\n v_add_u32 %4, %4, %5\n v_add_u32 %5, %5, %6\n v_add_u32 %6, %6, %7\n v_add_u32 %7, %7, %0")
DEFK (k_shr_imm,  "v_lshrrev_b32 %0, 1, %1\n v_lshrrev_b32 %1, 1, %2\n v_lshrrev_b32 %2, 1, %3\n v_lshrrev_b32 %3, 1, %4\n v_lshrrev_b32 %4, 1, %5\n v_lshrrev_b32 %5, 1, %6\n v_lshrrev_b32 %6, 1, %7\n v_lshrrev_b32 %7, 1, %0")
DEFK (k_mov,      "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0")
DEFK (k_xor_e64,  "v_xor_b32_e64 %0, %0, %1\n v_xor_b32_e64 %1, %1, %2\n v_xor_b32_e64 %2, %2, %3\n v_xor_b32_e64 %3, %3, %4\n v_xor_b32_e64 %4, %4, %5\n v_xor_b32_e64 %5, %5, %6\n v_xor_b32_e64 %6, %6, %7\n v_xor_b32_e64 %7, %7, %0")
DEFK (k_and_or,   "v_and_or_b32 %0, %0, %1, %2\n v_and_or_b32 %1, %1, %2, %3\n v_and_or_b32 %2, %2, %3, %4\n v_and_or_b32 %3, %3, %4, %5\n v_and_or_b32 %4, %4, %5, %6\n v_and_or_b32 %5, %5, %6, %7\n v_and_or_b32 %6, %6, %7, %0\n v_and_or_b32 %7, %7, %0, %1")
DEFK (k_and_or_s, "v_and_or_b32 %0, %0, %8, %1\n v_and_or_b32 %1, %1, %8, %2\n v_and_or_b32 %2, %2, %8, %3\n v_and_or_b32 %3, %3, %8, %4\n v_and_or_b32 %4, %4, %8, %5\n v_and_or_b32 %5, %5, %8, %6\n v_and_or_b32 %6, %6, %8, %7\n v_and_or_b32 %7, %7, %8, %0")
DEFK (k_lshl_or,  "v_lshl_or_b32 %0, %0, 2, %1\n v_lshl_or_b32 %1, %1, 2, %2\n v_lshl_or_b32 %2, %2, 2, %3\n v_lshl_or_b32 %3, %3, 2, %4\n v_lshl_or_b32 %4, %4, 2, %5\n v_lshl_or_b32 %5, %5, 2, %6\n v_lshl_or_b32 %6, %6, 2, %7\n v_lshl_or_b32 %7, %7, 2, %0")
DEFK (k_bitop3_3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96\n v_bitop3_b32 %1, %1, %2, %3 bitop3:0x96\n v_bitop3_b32 %2, %2, %3, %4 bitop3:0x96\n v_bitop3_b32 %3, %3, %4, %5 bitop3:0x96\n v_bitop3_b32 %4, %4, %5, %6 bitop3:0x96\n v_bitop3_b32 %5, %5, %6, %7 bitop3:0x96\n v_bitop3_b32 %6, %6, %7, %0 bitop3:0x96\n v_bitop3_b32 %7, %7, %0, %1 bitop3:0x96")
DEFK (k_bitop3_s, "v_bitop3_b32 %0, %0, %8, %1 bitop3:0x96\n v_bitop3_b32 %1, %1, %8, %2 bitop3:0x96\n v_bitop3_b32 %2, %2, %8, %3 bitop3:0x96\n v_bitop3_b32 %3, %3, %8, %4 bitop3:0x96\n v_bitop3_b32 %4, %4, %8, %5 bitop3:0x96\n v_bitop3_b32 %5, %5, %8, %6 bitop3:0x96\n v_bitop3_b32 %6, %6, %8, %7 bitop3:0x96\n v_bitop3_b32 %7, %7, %8, %0 bitop3:0x96")
DEFK (k_dot4_s,   "v_dot4_u32_u8 %0, %0, %8, %1\n v_dot4_u32_u8 %1, %1, %8, %2\n v_dot4_u32_u8 %2, %2, %8, %3\n v_dot4_u32_u8 %3, %3, %8, %4\n v_dot4_u32_u8 %4, %4, %8, %5\n v_dot4_u32_u8 %5, %5, %8, %6\n v_dot4_u32_u8 %6, %6, %8, %7\n v_dot4_u32_u8 %7, %7, %8, %0")
DEFK (k_dot4_0,   "v_dot4_u32_u8 %0, %1, %8, 0\n v_dot4_u32_u8 %1, %2, %8, 0\n v_dot4_u32_u8 %2, %3, %8, 0\n v_dot4_u32_u8 %3, %4, %8, 0\n v_dot4_u32_u8 %4, %5, %8, 0\n v_dot4_u32_u8 %5, %6, %8, 0\n v_dot4_u32_u8 %6, %7, %8, 0\n v_dot4_u32_u8 %7, %0, %8, 0")
DEFK (k_perm_s,   "v_perm_b32 %0, %0, %1, %8\n v_perm_b32 %1, %1, %2, %8\n v_perm_b32 %2, %2, %3, %8\n v_perm_b32 %3, %3, %4, %8\n v_perm_b32 %4, %4, %5, %8\n v_perm_b32 %5, %5, %6, %8\n v_perm_b32 %6, %6, %7, %8\n v_perm_b32 %7, %7, %0, %8")
DEFK (k_perm_tbl, "v_perm_b32 %0, %8, %8, %1\n v_perm_b32 %1, %8, %8, %2\n v_perm_b32 %2, %8, %8, %3\n v_perm_b32 %3, %8, %8, %4\n v_perm_b32 %4, %8, %8, %5\n v_perm_b32 %5, %8, %8, %6\n v_perm_b32 %6, %8, %8, %7\n v_perm_b32 %7, %8, %8, %0")
DEFK (k_mov_dpp,  "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 wave_shr:1 row_mask:0xf bank_mask:0xf")
DEFK (k_or_dpp,   "v_or_b32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_or_b32_dpp %1, %2, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_or_b32_dpp %2, %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_or_b32_dpp %3, %4, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_or_b32_dpp %4, %5, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_or_b32_dpp %5, %6, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_or_b32_dpp %6, %7, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_or_b32_dpp %7, %0, %7 row_shr:1 row_mask:0xf bank_mask:0xf")
DEFK (k_sdwa,     "v_or_b32_sdwa %0, %1, %0 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:DWORD\n v_or_b32_sdwa %1, %2, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:DWORD\n v_or_b32_sdwa %2, %3, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:DWORD\n v_or_b32_sdwa %3, %4, %3 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:DWORD\n v_or_b32_sdwa %4, %5, %4 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:DWORD\n v_or_b32_sdwa %5, %6, %5 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:DWORD\n v_or_b32_sdwa %6, %7, %6 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:DWORD\n v_or_b32_sdwa %7, %0, %7 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:DWORD")
DEFK (k_alignbit, "v_alignbit_b32 %0, %0, %1, 24\n v_alignbit_b32 %1, %1, %2, 24\n v_alignbit_b32 %2, %2, %3, 24\n v_alignbit_b32 %3, %3, %4, 24\n v_alignbit_b32 %4, %4, %5, 24\n v_alignbit_b32 %5, %5, %6, 24\n v_alignbit_b32 %6, %6, %7, 24\n v_alignbit_b32 %7, %7, %0, 24")
DEFK (k_bfe,      "v_bfe_u32 %0, %1, 3, 5\n v_bfe_u32 %1, %2, 3, 5\n v_bfe_u32 %2, %3, 3, 5\n v_bfe_u32 %3, %4, 3, 5\n v_bfe_u32 %4, %5, 3, 5\n v_bfe_u32 %5, %6, 3, 5\n v_bfe_u32 %6, %7, 3, 5\n v_bfe_u32 %7, %0, 3, 5")
DEFK (k_mul24,    "v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %1, %1, %2\n v_mul_u32_u24 %2, %2, %3\n v_mul_u32_u24 %3, %3, %4\n v_mul_u32_u24 %4, %4, %5\n v_mul_u32_u24 %5, %5, %6\n v_mul_u32_u24 %6, %6, %7\n v_mul_u32_u24 %7, %7, %0")
DEFK (k_mullo,    "v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %3, %3, %4\n v_mul_lo_u32 %4, %4, %5\n v_mul_lo_u32 %5, %5, %6\n v_mul_lo_u32 %6, %6, %7\n v_mul_lo_u32 %7, %7, %0")
DEFK (k_bcnt,     "v_bcnt_u32_b32 %0, %0, %1\n v_bcnt_u32_b32 %1, %1, %2\n v_bcnt_u32_b32 %2, %2, %3\n v_bcnt_u32_b32 %3, %3, %4\n v_bcnt_u32_b32 %4, %4, %5\n v_bcnt_u32_b32 %5, %5, %6\n v_bcnt_u32_b32 %6, %6, %7\n v_bcnt_u32_b32 %7, %7, %0")
DEFK (k_ffbl,     "v_ffbl_b32 %0, %1\n v_ffbl_b32 %1, %2\n v_ffbl_b32 %2, %3\n v_ffbl_b32 %3, %4\n v_ffbl_b32 %4, %5\n v_ffbl_b32 %5, %6\n v_ffbl_b32 %6, %7\n v_ffbl_b32 %7, %0")
DEFK (k_bfrev,    "v_bfrev_b32 %0, %1\n v_bfrev_b32 %1, %2\n v_bfrev_b32 %2, %3\n v_bfrev_b32 %3, %4\n v_bfrev_b32 %4, %5\n v_bfrev_b32 %5, %6\n v_bfrev_b32 %6, %7\n v_bfrev_b32 %7, %0")
DEFK (k_pk_add16, "v_pk_add_u16 %0, %0, %1\n v_pk_add_u16 %1, %1, %2\n v_pk_add_u16 %2, %2, %3\n v_pk_add_u16 %3, %3, %4\n v_pk_add_u16 %4, %4, %5\n v_pk_add_u16 %5, %5, %6\n v_pk_add_u16 %6, %6, %7\n v_pk_add_u16 %7, %7, %0")

typedef void (*kfn) (u32 *, int, u32, u64 *);
static void run (const char *name, kfn f, u32 *d, u64 *clk)
{
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
  f<<<256 * 2, 1024>>> (d, 10, 1, clk);
  hipDeviceSynchronize ();
  hipEventRecord (e0);
  f<<<256 * 2, 1024>>> (d, iters, 1, clk);
  hipEventRecord (e1); hipEventSynchronize (e1);
  float ms; hipEventElapsedTime (&ms, e0, e1);
  u64 h[2]; hipMemcpy (h, clk, 16, hipMemcpyDeviceToHost);
  const double n_per_simd = (double) iters * 64 * 8;         // wave-instructions per SIMD (8 waves)
  const double ghz = (double) h[0] / ((double) h[1] * 10.0);   // memtime ticks per ns (memrealtime = 100 MHz)
  printf ("%-12s %8.3f ms  %6.3f ns/instr  clock %.2f GHz  -> %.2f cycles/instr (in-kernel: %.2f)\n", name, ms, ms * 1e6 / n_per_simd, ghz,
          ms * 1e6 / n_per_simd * ghz, (double) h[0] / n_per_simd);
}

int main ()
{
  u32 *d; u64 *clk; hipMalloc (&d, 4096); hipMalloc (&clk, 64);
#define RUN(K) run (#K, K, d, clk)
  RUN (k_xor_vv); RUN (k_xor_sv); RUN (k_xor_lit); RUN (k_add_vv); RUN (k_shr_imm); RUN (k_mov); RUN (k_xor_e64); RUN (k_and_or); RUN (k_and_or_s); RUN (k_lshl_or);
  RUN (k_bitop3_3); RUN (k_bitop3_s); RUN (k_dot4_s); RUN (k_dot4_0); RUN (k_perm_s); RUN (k_perm_tbl); RUN (k_mov_dpp); RUN (k_or_dpp); RUN (k_sdwa); RUN (k_alignbit); RUN (k_bfe);
  return 0;
}
