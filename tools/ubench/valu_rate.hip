// micro-benchmark: integer VALU issue rate on gfx950 as a function of waves per SIMD (diagnostic, not part of the library)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32;

template <int OP>
__global__ void k (u32 *out, int iters, u32 seed)
{
  u32 a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
  const u32 c = seed | 0x01010101u;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (OP == 0) { a0 ^= c; a1 += c; a2 &= ~c; a3 |= c; a4 ^= a0; a5 += a1; a6 ^= a2; a7 += a3; asm volatile ("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
      if (OP == 1) { a0 = __builtin_amdgcn_udot4 (a0, c, a1, false); a1 = __builtin_amdgcn_udot4 (a1, c, a2, false); a2 = __builtin_amdgcn_udot4 (a2, c, a3, false); a3 = __builtin_amdgcn_udot4 (a3, c, a4, false);
                     a4 = __builtin_amdgcn_udot4 (a4, c, a5, false); a5 = __builtin_amdgcn_udot4 (a5, c, a6, false); a6 = __builtin_amdgcn_udot4 (a6, c, a7, false); a7 = __builtin_amdgcn_udot4 (a7, c, a0, false); }
      if (OP == 2) { a0 = __builtin_amdgcn_perm (a0, a1, c); a1 = __builtin_amdgcn_perm (a1, a2, c); a2 = __builtin_amdgcn_perm (a2, a3, c); a3 = __builtin_amdgcn_perm (a3, a4, c);
                     a4 = __builtin_amdgcn_perm (a4, a5, c); a5 = __builtin_amdgcn_perm (a5, a6, c); a6 = __builtin_amdgcn_perm (a6, a7, c); a7 = __builtin_amdgcn_perm (a7, a0, c); }
      if (OP == 3) { a0 = __builtin_amdgcn_alignbit (a0, a1, 7); a1 = __builtin_amdgcn_alignbit (a1, a2, 7); a2 = __builtin_amdgcn_alignbit (a2, a3, 7); a3 = __builtin_amdgcn_alignbit (a3, a4, 7);
                     a4 = __builtin_amdgcn_alignbit (a4, a5, 7); a5 = __builtin_amdgcn_alignbit (a5, a6, 7); a6 = __builtin_amdgcn_alignbit (a6, a7, 7); a7 = __builtin_amdgcn_alignbit (a7, a0, 7); }
      if (OP == 4) { // bitop3 via the (a ^ b) & c pattern
        a0 = (a0 ^ a1) & c; a1 = (a1 ^ a2) & c; a2 = (a2 ^ a3) & c; a3 = (a3 ^ a4) & c; a4 = (a4 ^ a5) & c; a5 = (a5 ^ a6) & c; a6 = (a6 ^ a7) & c; a7 = (a7 ^ a0) | c;
        asm volatile ("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
      if (OP == 5) { // 64-bit shift
        unsigned long long x = ((unsigned long long) a1 << 32) | a0, y = ((unsigned long long) a3 << 32) | a2, z = ((unsigned long long) a5 << 32) | a4, w = ((unsigned long long) a7 << 32) | a6;
        x >>= (c & 31); y >>= (c & 31); z <<= (c & 31); w <<= (c & 31);
        a0 = (u32) x; a1 = (u32) (x >> 32) ^ a0; a2 = (u32) y; a3 = (u32) (y >> 32) ^ a2; a4 = (u32) z; a5 = (u32) (z >> 32) ^ a4; a6 = (u32) w; a7 = (u32) (w >> 32) ^ a6;
        asm volatile ("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
      if (OP == 6) { // dpp mov
        a0 += (u32) __builtin_amdgcn_update_dpp (0, (int) a1, 0x138, 0xF, 0xF, false); a1 += (u32) __builtin_amdgcn_update_dpp (0, (int) a2, 0x138, 0xF, 0xF, false);
        a2 += (u32) __builtin_amdgcn_update_dpp (0, (int) a3, 0x138, 0xF, 0xF, false); a3 += (u32) __builtin_amdgcn_update_dpp (0, (int) a4, 0x138, 0xF, 0xF, false);
        a4 += (u32) __builtin_amdgcn_update_dpp (0, (int) a5, 0x138, 0xF, 0xF, false); a5 += (u32) __builtin_amdgcn_update_dpp (0, (int) a6, 0x138, 0xF, 0xF, false);
        a6 += (u32) __builtin_amdgcn_update_dpp (0, (int) a7, 0x138, 0xF, 0xF, false); a7 += (u32) __builtin_amdgcn_update_dpp (0, (int) a0, 0x138, 0xF, 0xF, false); }
      if (OP == 7) { a0 = __builtin_amdgcn_sad_u8 (a0, c, a1); a1 = __builtin_amdgcn_sad_u8 (a1, c, a2); a2 = __builtin_amdgcn_sad_u8 (a2, c, a3); a3 = __builtin_amdgcn_sad_u8 (a3, c, a4);
                     a4 = __builtin_amdgcn_sad_u8 (a4, c, a5); a5 = __builtin_amdgcn_sad_u8 (a5, c, a6); a6 = __builtin_amdgcn_sad_u8 (a6, c, a7); a7 = __builtin_amdgcn_sad_u8 (a7, c, a0); }
      if (OP == 8) { a0 = __popc (a0) + a1; a1 = __brev (a1) ^ a2; a2 = __ffs ((int) a2) + a3; a3 = __popc (a3) + a4; a4 = __brev (a4) ^ a5; a5 = __ffs ((int) a5) + a6; a6 = __popc (a6) + a7; a7 = __brev (a7) ^ a0; }
    }
  }
  u32 r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
  if (r == 0x12345u) out[threadIdx.x] = r;
}

template <int OP> static void run (const char *name, u32 *d)
{
  const int iters = 2000;
  const int cfg[][2] = {{256, 1}, {512, 1}, {1024, 1}, {1024, 2}, {512, 3}, {256, 8}};   // block, blocks per CU
  printf ("%-22s", name);
  for (auto &c : cfg) {
    hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
    k<OP><<<256 * c[1], c[0]>>> (d, 10, 1);
    hipDeviceSynchronize ();
    hipEventRecord (e0);
    k<OP><<<256 * c[1], c[0]>>> (d, iters, 1);
    hipEventRecord (e1); hipEventSynchronize (e1);
    float ms; hipEventElapsedTime (&ms, e0, e1);
    const double waves_per_simd = c[0] / 64.0 * c[1] / 4.0;
    // instructions per wave: iters * 8 * (ops per u-iteration); report ns per (wave-instruction-group of 8) per SIMD
    const double groups = (double) iters * 8 * waves_per_simd;       // per SIMD
    printf ("  w/simd %.0f: %7.3f ms (%.2f ns/grp)", waves_per_simd, ms, ms * 1e6 / groups);
  }
  printf ("\n");
}

int main ()
{
  u32 *d; hipMalloc (&d, 4096);
  run<0> ("xor/add/and/or x8", d);
  run<1> ("dot4 x8", d);
  run<2> ("perm x8", d);
  run<3> ("alignbit x8", d);
  run<4> ("bitop3 x8", d);
  run<5> ("shift64 x4 + 4 xor", d);
  run<6> ("add_dpp x8", d);
  run<7> ("sad_u8 x8", d);
  run<8> ("popc/brev/ffs x8(+)", d);
  return 0;
}
