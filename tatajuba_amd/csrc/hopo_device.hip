// hopo_device.hip -- HIP/CDNA4 (gfx950) kernels and the thin C-ABI layer (tjamd_*) of the homopolymer-tract engine.
//
// Replaces, on the device, tatajuba's per-read scan (reference: src/hopo_counter.c:219-258,285-307) and the per-sample
// sort / dedupe / filter / index / coverage (reference: src/hopo_counter.c:339-438).  Integer and byte work only:
// HBM-bound, no MFMA.  See DESIGN.md for the data layout and the roofline of each kernel.
//
// Scan kernel in one paragraph: a workgroup (512 threads = 8 wavefronts, three resident per CU) takes tiles of 8 KiB of
// the '\n'-delimited read stream from a work counter.  The next tile's bytes go from HBM straight into LDS
// (global_load_lds, 1 KiB per wave instruction) while the current one is worked on.  Every lane classifies its 16 bytes
// with SWAR bit logic into 2-bit base codes plus three bit-planes (run start, read delimiter, non-ACGTU) kept in LDS.
// Run starts that begin a tract of >= m bases are found with shifted ANDs of the run-start plane, compacted into a
// workgroup-wide candidate list, and then one lane per candidate finds the run end (count-trailing-zeros on the plane),
// checks both flanks against the delimiter plane, pulls the two k-mers out of the packed codes with funnel shifts,
// canonicalises (reverse complement = bit reverse + pair swap of the complemented word) and appends a packed 8/16/32
// byte record to an LDS staging buffer, which the workgroup partitions into 256 hash buckets in HBM with one counting
// sort per few thousand records.  The finalise step aggregates every bucket in an LDS hash table (one workgroup per
// bucket), orders the survivors with a bin partition + per-wavefront rank sort, and derives the index and the coverage
// in the same pass.  File map: helpers and tile scan (scan_tiles) -> sinks (located list, bucket staging) -> bucket
// storage -> aggregation kernels -> radix sort / scans (fallback paths, merge) -> bin kernels -> host layer (tjamd_*).

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>                  // types only: the library is looked up at run time (rccl_api)
#include <dlfcn.h>
#include <mutex>
#include <string>
#include <new>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <limits.h>
#include <algorithm>
#include <vector>

#include "../../include/tatajuba_amd.h"

typedef unsigned long long u64;
typedef unsigned int u32;

// ---------------------------------------------------------------------------------------------------------------
// error plumbing

static thread_local char g_err[512] = "";

static int set_err (int code, const char *fmt, ...)
{
  va_list ap;
  va_start (ap, fmt);
  vsnprintf (g_err, sizeof (g_err), fmt, ap);
  va_end (ap);
  return code;
}

#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
  return set_err (TJAMD_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString (e_), __FILE__, __LINE__); } while (0)
#define HIPCHK_NULL(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
  set_err (TJAMD_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString (e_), __FILE__, __LINE__); return NULL; } } while (0)

extern "C" const char *tjamd_last_error (void) { return g_err; }
// (for the C half of the library, hopo_host.c; not exported)
extern "C" void tj_set_last_error (const char *msg) { (void) set_err (TJAMD_ERR_ARG, "%s", msg ? msg : "error"); }

extern "C" int tjamd_device_count (void)
{
  int n = 0;
  if (hipGetDeviceCount (&n) != hipSuccess) return 0;
  return n;
}

// ---------------------------------------------------------------------------------------------------------------
// geometry of the scan tile

#define TJ_HL      64                   // left halo  (>= max k + 1, multiple of 32)
#define TJ_HR      192                  // right halo (tracts whose end + k stays inside are handled from LDS)

template <int BLOCK, int TILE, int CANDDIV = 2>
struct TileLds
{
  static constexpr int WIN = TJ_HL + TILE + TJ_HR;       // bytes of stream seen by one tile
  static constexpr int NCHUNK = WIN / 16;                // 16-byte chunks
  static constexpr int MASKW = WIN / 32 + 4;             // words per bit-plane (+ zeroed pad for 3-word funnel reads)
  static constexpr int CODEW = WIN / 16 + 4;             // words of 2-bit codes (+ pad)
  static constexpr int MAXCAND = TILE / CANDDIV;         // tracts have >= 2 bases: one candidate per 2 bytes at most
                                                         // (CANDDIV 1: the located kernel's all-monomers mode)
  static constexpr int NLOAD = (NCHUNK + BLOCK - 1) / BLOCK;
  static constexpr int NRAW = NCHUNK + NLOAD * (BLOCK / 64);   // landing zone: the chunks + per wave and load, the chunk in front of its first one
  u32 code[CODEW];
  u32 start[MASKW];
  u32 sent[MASKW];
  u32 inval[MASKW];
  unsigned short cand[MAXCAND];
  u32 ncand;
  u32 grp[2];                                            // first tile of the current / next group of this workgroup
  u32 odd[2];                                            // per tile parity: some byte of the tile is neither ACGT nor '\n'
};

struct DevCounters
{
  u64 n_rec;        // records appended to the located list
  u64 n_fix_unused;
  u64 n_undefined;  // qualifying non-ACGTU runs without an earlier tract in the read (reference: uninitialised memory)
  u64 n_null;       // padding records written into the buckets (reserved slots that stayed empty)
  u32 overflow;     // output list / a bucket too small
  u32 fix_overflow; // fix list too small
  u32 pad[2];
  // per-launch counters, double-buffered: launch i works on lc[i & 1] and zeroes lc[(i + 1) & 1] for the next launch
  // (no memset between launches)
  // work = tile counter of the main scan kernel of the launch; n_slow = tiles the fast kernel handed to the generic one
  // (stream edges, bytes outside ACGT\n, too many candidates); work_slow = the generic kernel's counter over that list
  struct { u64 n_fix; u32 work, n_slow, work_slow, ticket, pad[2]; } lc[2];   // (ticket: workgroups of the generic kernel that are through, see scan_bins_kernel)
};

#ifndef TJ_TILE_GROUP
#define TJ_TILE_GROUP 16
#endif

struct FixEntry { long long pos; long long len; };

// counts of the finalise step (device block read back by the host)
struct FinCounts { u32 n_seg, n_kept, n_ctx, n_idx; int coverage; u32 overflow, sort_fallback, pad; };
// The sizes the ordering kernels work with, derived on the device from the number of kept records so that the host does
// not have to fetch that number between the aggregation and them (plan_tail; ok = 0: nothing to do, or more records
// than the buffers were sized for -- every kernel then returns at once and the host takes the exact path).
struct FinPlan { long n1; long t; int nbits, nbins, log2t, ok; u32 kept_overflow, cov_direct; long n1_exact; u32 ovf_exact, pad2; };   // (n1_exact, ovf_exact: read at the kernel boundary after the aggregation, see clear_buckets_kernel)


// ---------------------------------------------------------------------------------------------------------------
// device helpers

__device__ __forceinline__ u32 zero_bytes (u32 t)
{ // 0x80 in every byte of t that is zero, exact (no borrow between bytes)
  return ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu);
}

__device__ __forceinline__ u32 gather_bit7 (u32 z)
{ // bit 7 of byte j -> bit j
  return ((z >> 7) | (z >> 14) | (z >> 21) | (z >> 28)) & 0xFu;
}

// Four stream bytes (x, byte 0 = lowest address) -> 8 bits of 2-bit codes, and 4-bit planes.
// Codes follow the reference's table (src/hopo_counter.c:205-216): A/a 0, C/c 1, G/g 2, T/t/U/u 3; every other byte is
// "other" and packs as 0 (reference :304-305 masks the table value 4 with 3).
__device__ __forceinline__ void classify_word (u32 x, u32 prev_byte, u32 &code8, u32 &start4, u32 &sent4, u32 &inval4)
{
  u32 a1 = x >> 1, a2 = x >> 2, a3 = x >> 3, a4 = x >> 4, a6 = x >> 6, a7 = x >> 7;
  // ACGTU in either case, as a boolean function of the byte's bits evaluated at bit 0 of every byte:
  // 0x41/43/47 (low nibble 1,3,7 with bit4 = 0) or 0x54/55 (low nibble 4,5 with bit4 = 1), bit5 free, bit6 = 1, bit7 = 0
  u32 v = ~a7 & a6 & ~a3 & ((~a4 & x & ~(a2 & ~a1)) | (a4 & a2 & ~a1));
  v &= 0x01010101u;
  u32 c = (a1 ^ a2) & (v * 3u);                       // ((b>>1)^(b>>2))&3 is the code of a valid byte
  code8 = (c | (c >> 6) | (c >> 12) | (c >> 18)) & 0xFFu;
  u32 iv = v ^ 0x01010101u;
  inval4 = (iv | (iv >> 7) | (iv >> 14) | (iv >> 21)) & 0xFu;
  sent4 = gather_bit7 (zero_bytes (x ^ 0x0A0A0A0Au));
  start4 = gather_bit7 (zero_bytes (x ^ ((x << 8) | prev_byte))) ^ 0xFu;   // run start: byte differs from its predecessor
}

// Fast path of classify_word for the bytes that make up almost all of a read stream: upper-case A C G T and the read
// delimiter.  Same code / start / delimiter planes (the code of a delimiter byte is never used); `bad` comes back non-zero if some byte is anything else (then the
// caller redoes the chunk with classify_word, which also produces the non-ACGTU plane).
// `hi` selects the weights 16..128 and the results are added to the incoming start4 / sent4: two words give one byte of
// each plane through the dot product's accumulator, without a merge instruction.
__device__ __forceinline__ void classify_word_fast (u32 x, u32 prev_word, bool hi, u32 &code8, u32 &start4, u32 &sent4, u32 &bad)
{
  const u32 wts = hi ? 0x80402010u : 0x08040201u;
  // 2-bit codes, packed with one dot product (byte j * 4^j)
  const u32 c = ((x >> 1) ^ (x >> 2)) & 0x03030303u;
  code8 = __builtin_amdgcn_udot4 (c, 0x40100401u, 0u, false);
  // delimiter: the only byte of the fast path without bit 6
  const u32 s = ~(x >> 6) & 0x01010101u;                 // (shift first: the and-not is then one instruction)
  sent4 = __builtin_amdgcn_udot4 (s, wts, sent4, false);
  // validation: every byte must be the letter its code stands for, or '\n' where bit 6 is clear -- one 8-entry byte
  // table look-up (v_perm_b32): entries 0..3 = 'A','C','G','T' by code, entries 4..7 = '\n'
  bad = x ^ __builtin_amdgcn_perm (0x0A0A0A0Au, 0x54474341u, c | (s << 2));
  // run start: byte differs from its predecessor
  const u32 d = x ^ __builtin_amdgcn_alignbit (x, prev_word, 24);
  // (the flag stays in bit 7 of its byte: the sum comes out 128 times too large and the caller's merge shift absorbs it)
  const u32 nz = (((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d) & 0x80808080u;
  start4 = __builtin_amdgcn_udot4 (nz, wts, start4, false);
}

// ---- diagnostic build only (-DTJ_STAMPS=1): where does a tile's time go?  Never enabled in the product library. ----
#ifndef TJ_STAMPS
#define TJ_STAMPS 0
#endif
#if TJ_STAMPS
__device__ unsigned long long tj_stamp_acc[32];
struct Stamper
{
  unsigned long long last, acc[16];
  __device__ __forceinline__ void begin () { for (int i = 0; i < 16; i++) acc[i] = 0; last = __builtin_amdgcn_s_memtime (); }
  __device__ __forceinline__ void mark (int i) { unsigned long long t = __builtin_amdgcn_s_memtime (); acc[i] += t - last; last = t; }
  __device__ __forceinline__ void flush () { if (threadIdx.x == 0) for (int i = 0; i < 16; i++) atomicAdd (&tj_stamp_acc[i], acc[i]); }
};
#define ASTAMP_DECL unsigned long long a_last = __builtin_amdgcn_s_memtime (), a_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define ASTAMP(i) { unsigned long long t_ = __builtin_amdgcn_s_memtime (); a_acc[i] += t_ - a_last; a_last = t_; }
#define ASTAMP_FLUSH if ((threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 8; i_++) atomicAdd (&tj_stamp_acc[16 + i_], a_acc[i_])
#if TJ_STAMPS == 2                                      // (the partition kernel's stamps alone)
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH
#define PLSTAMP_DECL Stamper stamper; stamper.begin (); sink.stp = &stamper
#define PLSTAMP(i) stamper.mark (i)
#define PLSTAMP_FLUSH stamper.flush ()
#else
#define STAMP_DECL Stamper stamper; stamper.begin (); sink.stp = &stamper
#define STAMP(i) stamper.mark (i)
#define STAMP_FLUSH stamper.flush ()
#define PLSTAMP_DECL
#define PLSTAMP(i)
#define PLSTAMP_FLUSH
#endif
#define PSTAMP(i) stp->mark (i)
#define STAMP_MEMBER Stamper *stp;
extern "C" int tjamd_debug_stamps (unsigned long long *out, int reset)
{
  unsigned long long z[32] = {0};
  if (hipMemcpyFromSymbol (out, HIP_SYMBOL (tj_stamp_acc), 32 * 8) != hipSuccess) return 1;
  if (reset && hipMemcpyToSymbol (HIP_SYMBOL (tj_stamp_acc), z, 32 * 8) != hipSuccess) return 1;
  return 0;
}
#else
#define ASTAMP_DECL
#define ASTAMP(i)
#define ASTAMP_FLUSH
#define STAMP_DECL
#define STAMP(i)
#define PSTAMP(i)
#define STAMP_FLUSH
#define STAMP_MEMBER
#define PLSTAMP_DECL
#define PLSTAMP(i)
#define PLSTAMP_FLUSH
#endif

// inclusive prefix sum over the 64 lanes of a wavefront with DPP adds (row shifts inside 16-lane rows, then the two
// row broadcasts): 6 VALU instructions instead of 6 ds_bpermute round trips
__device__ __forceinline__ u32 wave_inclusive_scan (u32 x)
{
  x += (u32) __builtin_amdgcn_update_dpp (0, (int) x, 0x111, 0xF, 0xF, true);   // row_shr:1
  x += (u32) __builtin_amdgcn_update_dpp (0, (int) x, 0x112, 0xF, 0xF, true);   // row_shr:2
  x += (u32) __builtin_amdgcn_update_dpp (0, (int) x, 0x114, 0xF, 0xF, true);   // row_shr:4
  x += (u32) __builtin_amdgcn_update_dpp (0, (int) x, 0x118, 0xF, 0xF, true);   // row_shr:8
  x += (u32) __builtin_amdgcn_update_dpp (0, (int) x, 0x142, 0xA, 0xF, true);   // row_bcast:15 -> rows 1 and 3
  x += (u32) __builtin_amdgcn_update_dpp (0, (int) x, 0x143, 0xC, 0xF, true);   // row_bcast:31 -> rows 2 and 3
  return x;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the wave's outstanding global stores and
// atomics (vmcnt(0), microseconds each under load); inside the scan nothing written to HBM is read back in the same
// launch, so the bucket writes and the cursor atomics are left in flight across barriers.
__device__ __forceinline__ void lds_barrier ()
{
  asm volatile ("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
// LDS atomic add of a wave-uniform value by the wave's first lane alone, result left in flight: *raw_out (lane 0) holds
// the old value once lgkmcnt has drained (lds_collect).  Hand-written because the compiler's version of "one lane adds
// for the wave" costs a dozen scalar instructions, which on this chip are no cheaper than vector ones.
__device__ __forceinline__ void lds_add_issue (u32 lds_addr, u32 value_uniform, u32 &raw_out)
{
  u64 saved; u32 tmp;
  asm volatile ("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tv_mov_b32 %1, %4\n\tds_add_rtn_u32 %2, %3, %1\n\ts_mov_b64 exec, %0"
                : "=&s"(saved), "=&v"(tmp), "=&v"(raw_out) : "v"(lds_addr), "s"(value_uniform) : "memory");
}
__device__ __forceinline__ u32 lds_collect (u32 raw)
{
  asm volatile ("s_waitcnt lgkmcnt(0)" : "+v"(raw) :: "memory");
  return (u32) __builtin_amdgcn_readfirstlane ((int) raw);
}

// 64 bits of a little-endian bit array starting at bit position `bitpos` (array padded by >= 2 words)
__device__ __forceinline__ u64 bits64 (const u32 *a, int bitpos)
{
  int w = bitpos >> 5, sh = bitpos & 31;
  u32 x0 = a[w], x1 = a[w + 1], x2 = a[w + 2];
  u32 lo = __funnelshift_r (x0, x1, sh);
  u32 hi = __funnelshift_r (x1, x2, sh);
  return ((u64) hi << 32) | lo;
}

// 32 bits of a bit array starting at bit position `bitpos` (array padded by >= 1 word)
__device__ __forceinline__ u32 bits32 (const u32 *a, int bitpos)
{
  const int w = bitpos >> 5;
  return __builtin_amdgcn_alignbit (a[w + 1], a[w], (u32) bitpos & 31u);
}

// reverse complement of a k-mer of at most 16 bases held in 32 bits
__device__ __forceinline__ u32 revcomp_k32 (u32 x, int k)
{
  const u32 y = __brev (~x);
  u32 z;                                                // pair swap = bit-field insert of y >> 1 (even bits) into y << 1; the
  asm ("v_bfi_b32 %0, %1, %2, %3" : "=v"(z) : "s"(0x55555555u), "v"(y >> 1), "v"(y << 1));   // compiler spends 5 instructions on it
  return z >> (32 - 2 * k);
}

__device__ __forceinline__ u64 kmask (int k) { return (k >= 32) ? ~0ull : ((1ull << (2 * k)) - 1ull); }

// reverse complement of a k-mer packed first-base-lowest (reference: src/hopo_counter.c:241-242 builds it base by base)
__device__ __forceinline__ u64 revcomp_k (u64 x, int k)
{
  const u64 y = __brevll (~x);
  const u32 lo = (u32) y, hi = (u32) (y >> 32);
  u32 zl, zh;                                           // the pair swap never crosses the 32-bit halves: one v_bfi_b32 each
  asm ("v_bfi_b32 %0, %1, %2, %3" : "=v"(zl) : "s"(0x55555555u), "v"(lo >> 1), "v"(lo << 1));
  asm ("v_bfi_b32 %0, %1, %2, %3" : "=v"(zh) : "s"(0x55555555u), "v"(hi >> 1), "v"(hi << 1));
  return (((u64) zh << 32) | zl) >> (64 - 2 * k);
}

__device__ __forceinline__ u32 stream_byte (const uint8_t *seq, long n, long p)
{ // outside the stream everything is a read delimiter
  return (p >= 0 && p < n) ? (u32) seq[p] : (u32) '\n';
}

__device__ __forceinline__ bool byte_is_acgtu (u32 b)
{
  u32 l = b | 0x20u;
  return l == 'a' || l == 'c' || l == 'g' || l == 't' || l == 'u';
}

__device__ __forceinline__ u32 byte_code (u32 b) { return byte_is_acgtu (b) ? (((b >> 1) ^ (b >> 2)) & 3u) : 0u; }

// bit i of x -> bits 2i and 2i+1 (both set)
__device__ __forceinline__ u64 spread_pairs (u32 x)
{
  u64 v = x;
  v = (v | (v << 16)) & 0x0000FFFF0000FFFFull;
  v = (v | (v << 8)) & 0x00FF00FF00FF00FFull;
  v = (v | (v << 4)) & 0x0F0F0F0F0F0F0F0Full;
  v = (v | (v << 2)) & 0x3333333333333333ull;
  v = (v | (v << 1)) & 0x5555555555555555ull;
  return v | (v << 1);
}

// canonical record fields from the two flanks as read (reference: src/hopo_counter.c:233-246).  linv / rinv mark the
// non-ACGTU flank positions (k-bit masks): the reference's table gives them 4 in BOTH columns, so they pack as 0 in the
// reverse-complemented orientation too -- force their forward code to 3 before complementing.
__device__ __forceinline__ void canonicalise (u64 left, u64 right, u32 linv, u32 rinv, u32 cb, int k,
                                              u64 &c0, u64 &c1, u32 &base, u32 &flag)
{
  if (cb < 2u) { c0 = left; c1 = right; base = cb; flag = 1u; }
  else {
    if (linv | rinv) { left |= spread_pairs (linv); right |= spread_pairs (rinv); }
    c0 = revcomp_k (right, k); c1 = revcomp_k (left, k); base = 3u - cb; flag = 2u;
  }
}

__device__ __forceinline__ u64 make_meta (u32 base, long len, u32 flag)
{ // reference: src/hopo_counter.c:293-302 (count 1, mismatches 0xffe, multi 0, neg_strand 0), 10-bit length store
  return (u64) base | (((u64) len & 0x3FFull) << TJ_META_LEN_SHIFT) | TJ_META_RAW_CONST | ((u64) flag << TJ_META_FLAG_SHIFT);
}

// Slow path from global memory for a run [gs, ge] of a valid base: both flanks present?  then build the flanks.
__device__ bool flanks_from_stream (const uint8_t *seq, long n, long gs, long ge, int k, u64 &left, u64 &right, u32 &linv, u32 &rinv)
{
  left = right = 0; linv = rinv = 0;
  for (int i = 0; i < k; i++) {
    u32 bl = stream_byte (seq, n, gs - k + i), br = stream_byte (seq, n, ge + 1 + i);
    if (bl == '\n' || br == '\n') return false;
    left |= (u64) byte_code (bl) << (2 * i);
    right |= (u64) byte_code (br) << (2 * i);
    linv |= (byte_is_acgtu (bl) ? 0u : 1u) << i;
    rinv |= (byte_is_acgtu (br) ? 0u : 1u) << i;
  }
  return true;
}

template <int W>
__device__ __forceinline__ void store_record (u64 *out, u64 idx, u64 c0, u64 c1, u64 meta, u64 pos)
{
  u64 *p = out + idx * W;
  p[0] = c0; p[1] = c1; p[2] = meta;
  if (W == 4) p[3] = pos;
}

// wave-aggregated append: one atomic per wavefront, lanes ranked by ballot prefix (must be called wave-uniformly)
template <int W>
__device__ __forceinline__ void emit_record (bool have, u64 c0, u64 c1, u64 meta, u64 pos, u64 *out, u64 cap, DevCounters *ctr)
{
  u64 mask = __ballot (have);
  if (!mask) return;
  int lane = threadIdx.x & 63;
  int leader = __ffsll ((long long) mask) - 1;
  u64 base = 0;
  if (lane == leader) base = atomicAdd (&ctr->n_rec, (u64) __popcll (mask));
  base = __shfl (base, leader);
  if (have) {
    u64 idx = base + (u64) __popcll (mask & ((1ull << lane) - 1ull));
    if (idx < cap) store_record<W> (out, idx, c0, c1, meta, pos);
    else ctr->overflow = 1u;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// tile scan, shared by both kernels.  The sink receives, once per round and from every thread of the workgroup
// (wave- and block-uniform call), at most one tract: (have, ctx0, ctx1, base, 10-bit length, strand flag, position).

// Prefetch of one 16-byte chunk per lane straight into LDS (global_load_lds_dwordx4: no VGPR destination, so nothing
// the register allocator can turn into a copy that waits for the data).  lds_wave_base is the slot of the wave's first
// lane; lane l's 16 bytes land at lds_wave_base + l.  A chunk that sticks out of the stream reads the stream's first
// bytes instead and is rebuilt bytewise when it is consumed.  Requires >= 16 readable bytes at seq (the host pads tiny
// streams).

// The LDS-DMA instruction itself, from inline assembly.  Through the builtin the compiler knows that a VMEM operation
// writes LDS and, unable to tell where, waits for it (s_waitcnt vmcnt(0): the prefetch AND every bucket store in
// flight) in front of LDS accesses all over phases 2 and 3 -- the prefetch was complete a few hundred cycles after its
// issue instead of a tile later.  Unknown to the compiler, it is waited for exactly once, by the explicit s_waitcnt in
// front of phase 1's reads of the landing zone.  (Hidden VMEM operations can only make the compiler's own vmcnt waits
// longer, never shorter: the counter returns in issue order.)  lds_wave_base must be wave-uniform.
__device__ __forceinline__ void lds_dma16 (const void *gptr, void *lds_wave_base)
{
  const u32 m0v = (u32) __builtin_amdgcn_readfirstlane ((int) (u32) (size_t) (lptr_t) lds_wave_base);
  u32 saved;                                            // (M0 is the compiler's: put it back)
#ifdef TJ_EXP_NT
  asm volatile ("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                : "=&s"(saved) : "v"(gptr), "s"(m0v) : "memory");
#else
  asm volatile ("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                : "=&s"(saved) : "v"(gptr), "s"(m0v) : "memory");
#endif
}
__device__ __forceinline__ void issue_chunk (const uint8_t *__restrict__ seq, long n_bytes, long g, uint4 *lds_wave_base)
{
  const bool inside = (g >= 0) && (g + 16 <= n_bytes);
  lds_dma16 (inside ? seq + g : seq, lds_wave_base);
}

// the same chunk, byte by byte, for chunks that are not entirely inside the stream (first and last tile only)
struct EdgeChunk { u32 x, y, z, w; };
__device__ __noinline__ EdgeChunk edge_chunk (const uint8_t *__restrict__ seq, long n_bytes, long g)
{
  u32 w[4] = {0, 0, 0, 0};
#pragma unroll
  for (int b = 0; b < 16; b++) w[b >> 2] |= stream_byte (seq, n_bytes, g + b) << (8 * (b & 3));
  EdgeChunk e = {w[0], w[1], w[2], w[3]};
  return e;
}

// `raw` is the LDS-DMA landing zone: next tile's bytes arrive there straight from HBM; slot c is read and refilled only
// by the lane that owns chunk c.  It must be a __shared__ variable of its own: the compiler orders every later LDS access
// that MAY alias an in-flight LDS-DMA behind vmcnt(0), and members of one struct all may alias -- as part of TileLds the
// "prefetch" was waited for at the first LDS instruction after its issue.
// Where the tiles of a launch come from.  list == nullptr: the stream cut into n_tiles pieces of TILE bytes, handed out in
// groups of TJ_TILE_GROUP.  Otherwise: the tiles the fast kernel (scan_fast_kernel) left to this one -- tile t of that
// kernel owns the tract starts in [t * fown, (t + 1) * fown); each is covered here by NSUB = ceil (fown / TILE) pieces,
// handed out two at a time (the list is short, or the stream is slow-path material anyway).
#define TJ_LIST_FOWN 16288               // = FK_OWN (static_assert below): tract starts per listed tile
struct TileSrc { const u32 *list; long fown; };

template <int BLOCK, int TILE, int CANDDIV, class Sink>
__device__ __forceinline__ void scan_tiles (const uint8_t *__restrict__ seq, long n_bytes, long n_tiles, int k, int mprime,
                                            TileLds<BLOCK, TILE, CANDDIV> &T, uint4 *raw, Sink &sink, DevCounters *ctr, FixEntry *fix, u32 fix_cap, int par,
                                            const TileSrc src = TileSrc {nullptr, 0})
{
  typedef TileLds<BLOCK, TILE, CANDDIV> G;
  const int tid = threadIdx.x;
  const u64 km = kmask (k);
  const u64 kbits = (1ull << k) - 1ull;                 // k <= 32
  const bool listed = src.list != nullptr;
  constexpr u32 nsub_listed = (u32) ((TJ_LIST_FOWN + TILE - 1) / TILE);    // (compile-time: no division per tile)
  const u32 nsub = listed ? nsub_listed : 1u;
  u32 *const work = listed ? &ctr->lc[par].work_slow : &ctr->lc[par].work;
  if (listed) n_tiles = (long) ctr->lc[par].n_slow * (long) nsub;
  // (a group is reserved while its predecessor's first tile is worked on and looked up during the predecessor's last
  // tile: with at least two tiles per group a workgroup barrier lies between the two.  A short list is handed out two
  // pieces at a time, for balance; a long one -- a stream full of N, say -- in groups like the whole stream.)
  const u32 tgroup = !listed ? (u32) TJ_TILE_GROUP : (n_tiles >= 8l * TJ_TILE_GROUP * (long) gridDim.x) ? (u32) TJ_TILE_GROUP : 2u;
  // first stream byte a work item owns, and how many (multiples of 16); the list entry of the piece looked at last is
  // kept (the pieces of one listed tile follow each other)
  u32 le_idx = 0xFFFFFFFFu; long le_val = 0;
  auto own_start = [&] (long w) -> long {
    if (!listed) return w * (long) TILE;
    const u32 li = (u32) w / nsub_listed;
    if (li != le_idx) { le_idx = li; le_val = (long) src.list[li] * (long) TJ_LIST_FOWN; }
    return le_val + (long) ((u32) w % nsub_listed) * TILE;
  };
  auto own_len = [&] (long w) -> int { return listed ? (int) min ((long) TILE, (long) TJ_LIST_FOWN - (long) ((u32) w % nsub_listed) * TILE) : TILE; };

  // Tiles are handed out dynamically in groups of TJ_TILE_GROUP (one global atomic per group): workgroups differ in
  // how many tracts their tiles hold, and a static split leaves the slow ones running alone at the end.
  if (tid == 0) {
    T.odd[0] = 0; T.odd[1] = 0; T.grp[0] = atomicAdd (work, tgroup);
    if (blockIdx.x == 0 && !listed) { ctr->lc[par ^ 1].n_fix = 0; ctr->lc[par ^ 1].work = 0; ctr->lc[par ^ 1].n_slow = 0; ctr->lc[par ^ 1].work_slow = 0; ctr->lc[par ^ 1].ticket = 0; }
  }
  lds_barrier ();
  long tile = (long) T.grp[0];
  long grp_end = tile + tgroup;
  u32 gpar = 0, it = 0;
  if (tile < n_tiles) {
    const long w0 = own_start (tile) - TJ_HL;
#pragma unroll
    for (int i = 0; i < G::NLOAD; i++) {
      const int c = tid + i * BLOCK;
      if (c < G::NCHUNK) {
        issue_chunk (seq, n_bytes, w0 + 16l * c, &raw[c - (tid & 63)]);
        if ((tid & 63) == 0) issue_chunk (seq, n_bytes, w0 + 16l * (c - 1), &raw[G::NCHUNK + (tid >> 6) + i * (BLOCK / 64)]);
      }
    }
  }

  STAMP_DECL;
  while (tile < n_tiles) {
    STAMP (0);
    if (tid == 0 && tile + tgroup == grp_end) T.grp[gpar ^ 1u] = atomicAdd (work, tgroup);  // first tile of a group: reserve the next
    const long g0 = own_start (tile) - TJ_HL;           // stream position of window byte 0 (may be negative)
    const int olen = own_len (tile);                    // tract starts in window positions [TJ_HL, TJ_HL + olen) belong to this tile

    // ---- phase 1: classify the prefetched chunks into LDS, then prefetch the next tile ------------------------
    const u32 tpar = it & 1u;
    if (tid == 0) { T.ncand = 0; T.odd[tpar ^ 1u] = 0; }
    if (tid < 4) { T.code[G::CODEW - 4 + tid] = 0; T.start[G::MASKW - 4 + tid] = 0; T.sent[G::MASKW - 4 + tid] = 0xFFFFFFFFu; T.inval[G::MASKW - 4 + tid] = 0; }
    const bool interior = (g0 >= 0) && (g0 + (long) G::WIN <= n_bytes);   // whole window inside the stream (uniform)
    asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");   // this lane's chunks of the tile have landed in raw
#pragma unroll
    for (int i = 0; i < G::NLOAD; i++) {
      const int c = tid + i * BLOCK;
      if (c < G::NCHUNK) {
        uint4 v = raw[c];
        if (!interior) {
          const long g = g0 + 16l * c;
          if (!((g >= 0) && (g + 16 <= n_bytes))) {
            const EdgeChunk e = edge_chunk (seq, n_bytes, g);
            v.x = e.x; v.y = e.y; v.z = e.z; v.w = e.w;
          }
        }
        const u32 w[4] = {v.x, v.y, v.z, v.w};
        u32 code32 = 0, st16 = 0, se16 = 0, iv16 = 0, bad = 0;
        // The byte in front of the chunk is the last byte of the lane before (one DPP move); the wave's first lane
        // finds it in the extra chunk its wave loaded for the purpose.
        u32 prevw = (u32) __builtin_amdgcn_update_dpp (0, (int) v.w, 0x138, 0xF, 0xF, false);      // wave_shr:1
        if ((tid & 63) == 0) {
          prevw = raw[G::NCHUNK + (tid >> 6) + i * (BLOCK / 64)].w;
          if (!interior) prevw = stream_byte (seq, n_bytes, g0 + 16l * c - 1) << 24;
        }
        const u32 prev0 = prevw >> 24;
#pragma unroll
        for (int j = 0; j < 4; j += 2) {                // two words per byte of the start / delimiter planes
          u32 c8a, c8b, s8 = 0, e8 = 0, ba, bb;
          classify_word_fast (w[j], prevw, false, c8a, s8, e8, ba);
          classify_word_fast (w[j + 1], w[j], true, c8b, s8, e8, bb);
          code32 |= (c8a | (c8b << 8)) << (16 * (j / 2));
          st16 |= j ? s8 << 1 : s8 >> 7;                // (s8 = 128 x the byte of 8 flags)
          se16 |= e8 << (4 * j); bad |= ba | bb;
          prevw = w[j + 1];
        }
        if (__builtin_expect (bad != 0u, 0)) {          // lower case, U, N, anything else: exact classification
          T.odd[tpar] = 1u;
          u32 prev = prev0;
          code32 = st16 = se16 = iv16 = 0;
          for (int j = 0; j < 4; j++) {
            u32 c8, s4, e4, i4;
            classify_word (w[j], prev, c8, s4, e4, i4);
            code32 |= c8 << (8 * j); st16 |= s4 << (4 * j); se16 |= e4 << (4 * j); iv16 |= i4 << (4 * j);
            prev = w[j] >> 24;
          }
        }
        T.code[c] = code32;
        reinterpret_cast<unsigned short *> (T.start)[c] = (unsigned short) st16;
        reinterpret_cast<unsigned short *> (T.sent)[c] = (unsigned short) se16;
        reinterpret_cast<unsigned short *> (T.inval)[c] = (unsigned short) (iv16 | se16);
      }
    }
    sink.tick ();
    STAMP (1);
    // the tile after this one: the next of the group, or the first of the next group (reserved TJ_TILE_GROUP - 1 tiles ago)
    const long nt = (tile + 1 < grp_end) ? tile + 1 : (long) T.grp[gpar ^ 1u];
    {
      if (nt < n_tiles) {
        const long ng0 = own_start (nt) - TJ_HL;
        if (ng0 >= 16 && ng0 + (long) G::WIN <= n_bytes) {      // (uniform) the whole window and the chunk in front are inside: no per-lane checks
          const uint8_t *pl = seq + ng0 + 16l * tid;
#pragma unroll
          for (int i = 0; i < G::NLOAD; i++) {
            const int c = tid + i * BLOCK;
            if (c < G::NCHUNK) {
              lds_dma16 (pl + 16l * i * BLOCK, &raw[c - (tid & 63)]);
              if ((tid & 63) == 0) lds_dma16 (pl + 16l * i * BLOCK - 16, &raw[G::NCHUNK + (tid >> 6) + i * (BLOCK / 64)]);
            }
          }
        }
        else {
#pragma unroll
          for (int i = 0; i < G::NLOAD; i++) {
            const int c = tid + i * BLOCK;
            if (c < G::NCHUNK) {
              issue_chunk (seq, n_bytes, ng0 + 16l * c, &raw[c - (tid & 63)]);
              if ((tid & 63) == 0) issue_chunk (seq, n_bytes, ng0 + 16l * (c - 1), &raw[G::NCHUNK + (tid >> 6) + i * (BLOCK / 64)]);
            }
          }
        }
      }
    }
    STAMP (2);
    lds_barrier ();
    STAMP (3);
#if defined(TJ_EXP_STOP_AFTER) && TJ_EXP_STOP_AFTER == 1       // experiment builds only (tools/exp_scan_pmc.sh)
    { if (tile + 1 >= grp_end) { gpar ^= 1u; grp_end = nt + tgroup; } tile = nt; it++; continue; }
#endif

    // ---- phase 2: candidate tract starts among this lane's 16 positions ------------------------------------
    {
      const int p0 = TJ_HL + 16 * tid;
      const u64 S = bits64 (T.start, p0);
      u32 cand = (u32) S & 0xFFFFu;
      if (mprime <= 0) cand &= (u32) (S >> 1);                      // monomer mode: the next position starts a run too
      for (int j = 1; j < mprime; j++) cand &= ~(u32) (S >> j);     // next m'-1 positions continue the run
      cand &= ~(u32) reinterpret_cast<unsigned short *> (T.sent)[p0 >> 4];  // a run of delimiters is not a tract
      if (16 * tid >= olen) cand = 0;                               // (a listed tile may own less than TILE positions)
      // one LDS atomic per wavefront (512 same-address atomics serialise): exclusive prefix of the lane counts
      const u32 n = (u32) __popc (cand);
      const u32 incl = wave_inclusive_scan (n);
      const u32 total = (u32) __builtin_amdgcn_readlane ((int) incl, 63);     // (a uniform operand keeps the compiler's atomic optimiser from looping over lanes)
      u32 wbase = 0;
      if ((tid & 63) == 63 && total) wbase = atomicAdd (&T.ncand, total);
      wbase = (u32) __builtin_amdgcn_readlane ((int) wbase, 63);
      u32 at = wbase + incl - n;
      // (never more than MAXCAND in a tile: a candidate takes m' >= 2 positions -- one in the monomer mode, where MAXCAND = TILE)
      while (cand) { int b = __ffs ((int) cand) - 1; cand &= cand - 1u; T.cand[at] = (unsigned short) (p0 + b); at++; }
    }
    STAMP (4);
    lds_barrier ();
    STAMP (5);
#if defined(TJ_EXP_STOP_AFTER) && TJ_EXP_STOP_AFTER == 2
    { if (tile + 1 >= grp_end) { gpar ^= 1u; grp_end = nt + tgroup; } tile = nt; it++; continue; }
#endif

    // ---- phase 3: one lane per candidate --------------------------------------------------------------------
    const int ncand = min ((int) T.ncand, G::MAXCAND);
    const bool tile_odd = T.odd[tpar] != 0u;            // some byte of this tile needed the exact classification
    for (int cb0 = 0; cb0 < ncand; cb0 += BLOCK) {
      const int ci = cb0 + tid;
      bool have = false;
      u64 c0 = 0, c1 = 0, pos = 0;
      u32 base = 0, flag = 0, len10 = 0;
      if (ci < ncand) {
        const int s = T.cand[ci];
        const long gs = g0 + s;
        int e = -1;                                   // run end: first run start after s
        const u32 ns32 = bits32 (T.start, s + 1);     // (within 32 positions for all but the longest tracts)
        if (ns32) e = s + __ffs ((int) ns32) - 1;
        else {
          const u64 ns = bits64 (T.start, s + 1);
          if (ns) e = s + __ffsll ((long long) ns) - 1;
          else for (int p = s + 65; p < G::WIN; p += 64) {
            const u64 n2 = bits64 (T.start, p);
            if (n2) { e = p + __ffsll ((long long) n2) - 2; break; }
          }
        }
        const bool inval = tile_odd && ((T.inval[s >> 5] >> (s & 31)) & 1u);
        bool ok;
        long len;
        u64 left = 0, right = 0;
        u32 cb = 0, linv = 0, rinv = 0;
        if (e >= 0 && e + k < G::WIN) {               // everything needed is in LDS
          len = e - s + 1;
          if (Sink::K32 || k <= 16) {                 // k <= 16: flanks, masks and k-mers fit 32-bit arithmetic (uniform branch)
            const u32 kb32 = (1u << k) - 1u, km32 = (k >= 16) ? 0xFFFFFFFFu : (1u << (2 * k)) - 1u;
            ok = (((bits32 (T.sent, s - k) | bits32 (T.sent, e + 1)) & kb32) == 0u);
            if (ok && !inval) {
              u32 l32 = bits32 (T.code, 2 * (s - k)) & km32, r32 = bits32 (T.code, 2 * (e + 1)) & km32;
              cb = (T.code[s >> 4] >> (2 * (s & 15))) & 3u;
              if (cb < 2u) { c0 = l32; c1 = r32; base = cb; flag = 1u; }
              else {
                if (tile_odd) {
                  linv = bits32 (T.inval, s - k) & kb32; rinv = bits32 (T.inval, e + 1) & kb32;
                  if (linv | rinv) { l32 |= (u32) spread_pairs (linv); r32 |= (u32) spread_pairs (rinv); }
                }
                c0 = revcomp_k32 (r32, k); c1 = revcomp_k32 (l32, k); base = 3u - cb; flag = 2u;
              }
              len10 = (u32) len & 0x3FFu;
              pos = (u64) gs;
              have = true;
              ok = false;                             // (record complete: skip the generic tail below)
            }
          }
          else {
            ok = ((bits64 (T.sent, s - k) & kbits) == 0ull) && ((bits64 (T.sent, e + 1) & kbits) == 0ull);
            if (ok && !inval) {
              left = bits64 (T.code, 2 * (s - k)) & km;
              right = bits64 (T.code, 2 * (e + 1)) & km;
              cb = (T.code[s >> 4] >> (2 * (s & 15))) & 3u;
              if (cb >= 2u && tile_odd) { linv = (u32) (bits64 (T.inval, s - k) & kbits); rinv = (u32) (bits64 (T.inval, e + 1) & kbits); }
            }
          }
        }
        else {                                        // tract runs past the window: walk the stream (rare)
          const u32 b = stream_byte (seq, n_bytes, gs);
          long ge = gs;
          while (stream_byte (seq, n_bytes, ge + 1) == b) ge++;
          len = ge - gs + 1;
          ok = flanks_from_stream (seq, n_bytes, gs, ge, k, left, right, linv, rinv);
          cb = byte_code (b);
        }
        if (ok) {
          if (!inval) {
            canonicalise (left, right, linv, rinv, cb, k, c0, c1, base, flag);
            len10 = (u32) ((u64) len & 0x3FFull);     // 10-bit store of the reference (src/hopo_counter.h:39)
            pos = (u64) gs;
            have = true;
          }
          else {                                      // non-ACGTU run: context comes from the previous tract of the read
            u64 at = atomicAdd ((unsigned long long *) &ctr->lc[par].n_fix, 1ull);
            if (at < fix_cap) { fix[at].pos = gs; fix[at].len = len; }
            else ctr->fix_overflow = 1u;
          }
        }
      }
      STAMP (6);
#if defined(TJ_EXP_STOP_AFTER) && TJ_EXP_STOP_AFTER == 3
      if (have) asm volatile ("" :: "v"(c0), "v"(c1), "v"(base), "v"(len10), "v"(flag), "v"(pos));
#else
      sink.put (have, c0, c1, base, len10, flag, pos, (u32) min (ncand - cb0, BLOCK));
#endif
      STAMP (7);
    }
    lds_barrier ();
    STAMP (8);
    if (tile + 1 >= grp_end) { gpar ^= 1u; grp_end = nt + tgroup; }
    tile = nt;
    it++;
  }
  STAMP_FLUSH;
}

// Non-ACGTU runs that qualify as tracts (reference: src/hopo_counter.c:246-248: add_kmer is called with whatever the
// previous tract of the same read left in context[], hopo_base_int and reverse_forward_flag).  Walks back through the
// read to the nearest earlier recorded tract of a valid base; false = there is none (undefined in the reference).
__device__ bool stale_context (const uint8_t *__restrict__ seq, long n_bytes, long gs, int k, int mprime,
                               u64 &c0, u64 &c1, u32 &base, u32 &flag)
{
  long p = gs - 1;
  while (p >= 0 && seq[p] != '\n') {
    const u32 b = seq[p];
    long q = p;
    while (q - 1 >= 0 && seq[q - 1] == b) q--;          // run [q, p]
    // a recorded tract: >= m' equal bases -- or, in monomer mode (mprime <= 0), exactly one
    if (byte_is_acgtu (b) && (mprime > 0 ? (p - q + 1) >= mprime : p == q)) {
      u64 left, right; u32 linv, rinv;
      // recorded iff k bases of the same read precede it (its right side is fine: it ends before our run does);
      // an earlier run would start even closer to the read start, so the search stops here either way
      if (!flanks_from_stream (seq, n_bytes, q, p, k, left, right, linv, rinv)) return false;
      canonicalise (left, right, linv, rinv, byte_code (b), k, c0, c1, base, flag);
      return true;
    }
    p = q - 1;
  }
  return false;
}

// ---- sink 1: one list of located records (CPU-entry and test aid; small inputs) ------------------------------------

struct ListSink
{
  static constexpr bool K32 = false;
  u64 *out; u64 cap; DevCounters *ctr;
  STAMP_MEMBER
  __device__ __forceinline__ void tick () {}
  __device__ __forceinline__ void put (bool have, u64 c0, u64 c1, u32 base, u32 len10, u32 flag, u64 pos, u32 round_max)
  {
    emit_record<4> (have, c0, c1, make_meta (base, len10, flag), pos, out, cap, ctr);
  }
};

__global__ __launch_bounds__ (256)
void scan_list_kernel (const uint8_t *__restrict__ seq, long n_bytes, long n_tiles, int k, int mprime,
                       u64 *__restrict__ out, u64 cap, DevCounters *ctr, FixEntry *fix, u32 fix_cap, int par)
{
  __shared__ TileLds<256, 4096, 1> T;
  __shared__ uint4 raw[TileLds<256, 4096, 1>::NRAW];
  ListSink sink = {out, cap, ctr};
  scan_tiles<256, 4096, 1> (seq, n_bytes, n_tiles, k, mprime, T, raw, sink, ctr, fix, fix_cap, par);
}

__global__ void nrun_fixup_list_kernel (const uint8_t *__restrict__ seq, long n_bytes, int k, int mprime,
                                        u64 *__restrict__ out, u64 cap, DevCounters *ctr, const FixEntry *fix, u32 fix_cap, int par)
{
  u64 n_fix = ctr->lc[par].n_fix;
  if (n_fix > fix_cap) n_fix = fix_cap;
  for (u64 i = blockIdx.x * (u64) blockDim.x + threadIdx.x; i < n_fix; i += (u64) gridDim.x * blockDim.x) {
    u64 c0, c1; u32 base, flag;
    if (stale_context (seq, n_bytes, fix[i].pos, k, mprime, c0, c1, base, flag)) {
      u64 idx = atomicAdd (&ctr->n_rec, 1ull);
      if (idx < cap) store_record<4> (out, idx, c0, c1, make_meta (base, fix[i].len, flag), (u64) fix[i].pos);
      else ctr->overflow = 1u;
    }
    else atomicAdd (&ctr->n_undefined, 1ull);
  }
}

// ---- compact raw records and their hash ---------------------------------------------------------------------------
// Between the scan and the aggregation a raw tract is W 64-bit words, W chosen from k so that nothing is wasted:
//   W = 1 (k <= 12): low half  = ctx1 | len[7:0] << 24
//                    high half = ctx0 | len[9:8] << 24 | base << 26 | flag << 27      (fields at fixed places: packing is
//                    two shift-or instructions on 32-bit halves whatever k is; the word without its flag is the reduction key)
//   W = 2 (k <= 28): { 1 | base << 1 | len[4:0] << 2 | ctx0 << 7 ,  ctx1 | len[9:5] << 56 | flag << 61 }
//                    (word 0 is never 0 and never uses bit 63: the aggregation claims table slots with it)
//   W = 4          : { ctx0, ctx1, base | len << 2 | flag << 12, 0 }   (k > 28; padded so that 4 records fill a 128-byte line)
// flag == 3 never occurs in a raw record (one tract, one strand): it marks a padding ("null") record.

template <int W> __device__ __forceinline__ void pack_raw (u64 c0, u64 c1, u32 base, u32 len10, u32 flag, int k, u64 *w);
#define R1_FLAG_SHIFT 59                // W = 1 record: strand flag in bits 59-60 (27-28 of the high half)
#define R1_KEY_MASK   (~(3ull << R1_FLAG_SHIFT))
__device__ __forceinline__ u64 pack_rec1 (u32 c0, u32 c1, u32 base, u32 len10, u32 flag)
{
  const u32 lo = c1 | (len10 << 24), hi = c0 | ((len10 >> 8) << 24) | (base << 26) | (flag << 27);
  return ((u64) hi << 32) | lo;
}
template <> __device__ __forceinline__ void pack_raw<1> (u64 c0, u64 c1, u32 base, u32 len10, u32 flag, int k, u64 *w)
{ w[0] = pack_rec1 ((u32) c0, (u32) c1, base, len10, flag); }
template <> __device__ __forceinline__ void pack_raw<2> (u64 c0, u64 c1, u32 base, u32 len10, u32 flag, int k, u64 *w)
{ w[0] = 1ull | ((u64) base << 1) | ((u64) (len10 & 31u) << 2) | (c0 << 7); w[1] = c1 | ((u64) (len10 >> 5) << 56) | ((u64) flag << 61); }
template <> __device__ __forceinline__ void pack_raw<4> (u64 c0, u64 c1, u32 base, u32 len10, u32 flag, int k, u64 *w)
{ w[0] = c0; w[1] = c1; w[2] = (u64) base | ((u64) len10 << 2) | ((u64) flag << 12); w[3] = 0; }
template <int W> __device__ __forceinline__ void pack_null (u64 *w) { pack_raw<W> (0, 0, 0, 0, 3u, 2, w); }

template <int W> __device__ __forceinline__ void unpack_raw (const u64 *w, int k, u64 &c0, u64 &c1, u32 &base, u32 &len10, u32 &flag);
template <> __device__ __forceinline__ void unpack_raw<1> (const u64 *w, int k, u64 &c0, u64 &c1, u32 &base, u32 &len10, u32 &flag)
{
  const u32 lo = (u32) w[0], hi = (u32) (w[0] >> 32);
  c1 = lo & 0xFFFFFFu; c0 = hi & 0xFFFFFFu; len10 = (lo >> 24) | (((hi >> 24) & 3u) << 8); base = (hi >> 26) & 1u; flag = (hi >> 27) & 3u;
}
template <> __device__ __forceinline__ void unpack_raw<2> (const u64 *w, int k, u64 &c0, u64 &c1, u32 &base, u32 &len10, u32 &flag)
{
  const u64 m56 = (1ull << 56) - 1ull;
  c0 = (w[0] >> 7) & m56; c1 = w[1] & m56;
  base = (u32) (w[0] >> 1) & 1u; len10 = ((u32) (w[0] >> 2) & 31u) | (((u32) (w[1] >> 56) & 31u) << 5); flag = (u32) (w[1] >> 61) & 3u;
}
template <> __device__ __forceinline__ void unpack_raw<4> (const u64 *w, int k, u64 &c0, u64 &c1, u32 &base, u32 &len10, u32 &flag)
{ c0 = w[0]; c1 = w[1]; base = (u32) (w[2] & 3ull); len10 = (u32) ((w[2] >> 2) & 0x3FFull); flag = (u32) ((w[2] >> 12) & 3ull); }

// hash of the reduction key (base, context, stored length): bits 0-7 pick the bucket, 8-19 the table slot, 32-63 the tag
__device__ __forceinline__ u64 hash_key (u64 c0, u64 c1, u32 base, u32 len10)
{ // 32-bit arithmetic only (64-bit multiplies cost several quarter-rate instructions each); the tag merely has to
  // differ from the slot bits: a tag collision costs a key comparison, never a wrong answer
  u32 a = (u32) c0 ^ __builtin_amdgcn_alignbit ((u32) (c0 >> 32), (u32) (c0 >> 32), 25) ^ __builtin_amdgcn_alignbit ((u32) c1, (u32) c1, 19)
          ^ __builtin_amdgcn_alignbit ((u32) (c1 >> 32), (u32) (c1 >> 32), 11) ^ ((base | (len10 << 2)) * 0x9E3779B1u);
  u32 h = a * 0x85EBCA6Bu;
  h ^= h >> 15; h *= 0xC2B2AE35u; h ^= h >> 16;
  const u32 tag = (a ^ (a >> 13)) * 0x27D4EB2Fu;
  return ((u64) tag << 32) | h;
}

// bucket of a key: byte-wise dot products (full-rate v_dot4_u32_u8) instead of multiplies; the flank k-mers are close to
// uniform, so a weighted byte sum spreads them evenly enough over 256 buckets (the tables inside a bucket use hash_key)
__device__ __forceinline__ u32 bucket_of_key (u64 c0, u64 c1, u32 base, u32 len10)
{
  u32 h = __builtin_amdgcn_udot4 ((u32) c0, 0x6D2B4F0Bu, base | (len10 << 2), false);
  h = __builtin_amdgcn_udot4 ((u32) (c0 >> 32), 0x1D59A735u, h, false);
  h = __builtin_amdgcn_udot4 ((u32) c1, 0xC5A34D17u, h, false);
  h = __builtin_amdgcn_udot4 ((u32) (c1 >> 32), 0x3B7F9165u, h, false);
  return (h ^ (h >> 8)) & 255u;
}

// bucket of a one-word record (k <= 12), straight from its two halves (the strand flag is not part of the key)
__device__ __forceinline__ u32 bucket_of_rec1 (u32 lo, u32 hi)
{
  u32 h = __builtin_amdgcn_udot4 (lo, 0x6D2B4F0Bu, 0u, false);
  h = __builtin_amdgcn_udot4 (hi & 0x07FFFFFFu, 0xC5A34D17u, h, false);
  return (h ^ (h >> 8)) & 255u;
}

#define TJ_P        256                 // hash buckets
#define TJ_PBITS    8
#ifndef TJ_STAGE_WORDS
#define TJ_STAGE_WORDS 2048
#endif
// TJ_STAGE_WORDS: 64-bit words of one-word records a workgroup stages in LDS between partition passes (twice that in the fast kernel)
#define TJ_CH0      1536                // chunk size unit in records; chunks are TJ_CH0 << ch_shift with ch_shift >= 2
#define TJ_EMPTY    0xFFFFFFFFu
// The bucket cursors take every reservation of every workgroup (millions of atomic adds per launch): each lives on a
// 256-byte line of its own, so that they spread over the memory channels instead of queueing up at the one or two that
// a packed 1 KB array maps to.  cursors[TJ_P * TJ_CSTRIDE] = the next free chunk.
#ifndef TJ_CSTRIDE
#define TJ_CSTRIDE  64
#endif

// Bucket storage.  A bucket is a sequence of records numbered by its cursor; a workgroup reserves a run of positions
// with one atomic add and record `pos` lives in the bucket's (pos / CH)-th chunk (runs are shorter than a chunk, so a
// run straddles at most one chunk boundary).  Chunks come from one pool: chunk 0 of every bucket is assigned by the
// host, and the thread whose reservation holds the FIRST record of chunk j takes chunk j + 1 from the pool and
// publishes it in the bucket's table -- a whole chunk before anybody needs it, so nobody waits in practice, nothing
// leaks, and a skewed hash distribution costs nothing: memory follows the data, not the fullest bucket.
struct Buckets
{
  u64 *pool;          // pool_chunks * CH * W words
  u32 *table;         // [TJ_P][maxj] chunk ids, TJ_EMPTY = not claimed yet
  u32 *cursors;       // records reserved per bucket b at [b * TJ_CSTRIDE]
  u32 *pool_next;     // next free chunk
  u32 pool_chunks, maxj, ch_shift;      // CH = TJ_CH0 << ch_shift
};

#define TJ_NOCHUNK  0xFFFFFFFEu         // published instead of a chunk id when the pool is exhausted
#define TJ_SPIN_MAX (1u << 20)

__device__ __forceinline__ u32 chunk_of_pos (const Buckets &B, u32 pos) { return (pos / TJ_CH0) >> B.ch_shift; }

// Called by whoever reserved records [p0, p0 + n) of bucket b (n smaller than a chunk): if the reservation holds the
// first record of chunk j, claim chunk j + 1 (chunk 0 is pre-assigned).
__device__ __forceinline__ void bucket_claim_ahead (const Buckets &B, u32 b, u32 p0, u32 n, DevCounters *ctr)
{
  const u32 j = chunk_of_pos (B, p0 + n - 1);
  if (((j * TJ_CH0) << B.ch_shift) < p0) return;        // chunk j was opened by an earlier reservation
  if (j + 1 >= B.maxj) { ctr->overflow = 1u; return; }
  const u32 mine = atomicAdd (B.pool_next, 1u);
  if (mine >= B.pool_chunks) ctr->overflow = 1u;
  __hip_atomic_store (B.table + (u64) b * B.maxj + j + 1, mine < B.pool_chunks ? mine : TJ_NOCHUNK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the same for a reservation of any length: every chunk whose first record lies in [p0, p0 + n) has its successor claimed
__device__ __forceinline__ void bucket_claim_ahead_range (const Buckets &B, u32 b, u32 p0, u32 n, DevCounters *ctr)
{
  const u32 ch = (u32) TJ_CH0 << B.ch_shift;
  for (u32 j = (p0 + ch - 1u) / ch; (u64) j * ch < (u64) p0 + n; j++) {
    if (j + 1 >= B.maxj) { ctr->overflow = 1u; return; }
    const u32 mine = atomicAdd (B.pool_next, 1u);
    if (mine >= B.pool_chunks) ctr->overflow = 1u;
    __hip_atomic_store (B.table + (u64) b * B.maxj + j + 1, mine < B.pool_chunks ? mine : TJ_NOCHUNK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// chunk id of the j-th chunk of bucket b (TJ_NOCHUNK if it has none).  wait = the entry may still be on its way from
// the thread that claims it (same launch): poll it, bounded.
__device__ __forceinline__ u32 bucket_chunk_id (const Buckets &B, u32 b, u32 j, bool wait, DevCounters *ctr)
{
  if (j >= B.maxj) return TJ_NOCHUNK;
  u32 *e = B.table + (u64) b * B.maxj + j;
  u32 ch = __hip_atomic_load (e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (wait) {
    for (u32 spins = 0; ch == TJ_EMPTY && spins < TJ_SPIN_MAX; spins++) {
      __builtin_amdgcn_s_sleep (4);
      ch = __hip_atomic_load (e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (ch == TJ_EMPTY && ctr) ctr->overflow = 2u;          // (2: waited in vain for a chunk id -- a liveness problem, not a full pool)
  }
  return ch >= TJ_NOCHUNK ? TJ_NOCHUNK : ch;
}

// index (in records) inside the pool of record `pos` of bucket b; ~0 if it has no storage
__device__ __forceinline__ u64 bucket_slot (const Buckets &B, u32 b, u32 pos, bool wait, DevCounters *ctr)
{
  const u32 j = chunk_of_pos (B, pos);
  const u32 ch = bucket_chunk_id (B, b, j, wait, ctr);
  if (ch == TJ_NOCHUNK) return ~0ull;
  return (((u64) ch * TJ_CH0) << B.ch_shift) + (pos - ((j * TJ_CH0) << B.ch_shift));
}

template <int W>
__device__ __forceinline__ void bucket_insert_slow (u64 c0, u64 c1, u32 base, u32 len10, u32 flag, int k, const Buckets &B, DevCounters *ctr)
{
  u64 w[W];
  pack_raw<W> (c0, c1, base, len10, flag, k, w);
  const u32 b = (W == 1) ? bucket_of_rec1 ((u32) w[0], (u32) (w[0] >> 32)) : bucket_of_key (c0, c1, base, len10);
  const u32 pos = atomicAdd (&B.cursors[b * TJ_CSTRIDE], 1u);
  bucket_claim_ahead (B, b, pos, 1u, ctr);
  const u64 at = bucket_slot (B, b, pos, true, ctr);
  if (at != ~0ull) { u64 *q = B.pool + at * W; for (int j = 0; j < W; j++) q[j] = w[j]; }
}

// ---- sink 2: hash-partition into TJ_P buckets, in bulk ---------------------------------------------------------------
// Tracts are appended to an LDS staging buffer as they are found (one LDS atomic per wavefront, no barrier).  When the
// next round might not fit, the workgroup partitions what it has staged with one counting sort: count per bucket, one
// global atomic per non-empty bucket to reserve a run, permute in place (through registers) into bucket order, and
// write every run out contiguously.  All the fixed costs of partitioning (barriers, reservations, chunk look-ups) are
// paid once per ~4096 records instead of once per tile.

template <int W, int BIG = 0>
struct StageLds
{
  // a staged record is WS words (the 4th word of a W = 4 record is padding and only exists in HBM); the two-word and
  // three-word formats get 21 KB instead of 16: fewer, fuller partition passes (their fixed cost is per pass), still three
  // workgroups per CU
  static constexpr int WS = (W == 4) ? 3 : W;
  // BIG = the fast kernel's sink (scan_fast_kernel, two workgroups per CU): 4096 words of records are staged at a time
  // (4096 one-word records -- twice the records per pass, runs twice as long -- or 2048 two-word ones: the fast kernel's
  // LDS budget is the same for every k), and every array has 64 spare entries so that partition_big can do without
  // per-record branches
  // BIG = 2: partition_log_kernel's (two workgroups per CU and nothing else in LDS): 8192 one-word records per pass
  static constexpr int WORDS = BIG == 2 ? 8192 : BIG ? TJ_STAGE_WORDS * 2 : (W == 1) ? TJ_STAGE_WORDS : 2688;
  static constexpr int S = WORDS / WS;                   // records
  static constexpr int SPARE = BIG ? 64 : 0;
  u64 rec[WORDS + SPARE * WS];                          // (BIG: slots S + lane take the writes of lanes without a record)
  alignas (16) u64 gbase[TJ_P];                         // pool index of the first record of the bucket's run (BIG: gbase + gbase2 = 256 entries of 16 bytes, see partition_pass)
  u64 gbase2[TJ_P];                                     // pool index of the part of the run that lies in the next chunk
  alignas (16) u32 hist[TJ_P + SPARE];                  // (BIG: entries TJ_P + lane are for the lanes without a record)
  alignas (16) u32 offs[TJ_P + SPARE];                               // start of the bucket's run in the sorted staging buffer
  u32 split[BIG == 2 ? 1 : TJ_P];                       // records of the run before the chunk boundary (BIG: not used, the threshold travels in gbase's entry)
  u32 wsum[TJ_P / 64];
  u32 n;
  unsigned char bin[S + SPARE];
};

typedef StageLds<1, true> StageLdsFast1;
static_assert (offsetof (StageLdsFast1, gbase2) == offsetof (StageLdsFast1, gbase) + sizeof (u64) * TJ_P, "gbase and gbase2 back to back");

template <int W, int BLOCK, int BIG = 0>
struct StageSink
{
  static constexpr bool K32 = (W == 1);                 // k <= 12: the scan may use 32-bit k-mer arithmetic
  static constexpr int S = StageLds<W, BIG>::S;
  static constexpr int WS = StageLds<W, BIG>::WS;
  static constexpr int R = (S + BLOCK - 1) / BLOCK;     // staged records per thread in a partition pass
  // The fast kernel's one-word sink partitions exactly PS records at a time whenever it holds that many (the usual case:
  // a pass is asked for when the next round of BLOCK candidates might not fit, i.e. with S - BLOCK .. S records staged):
  // every round of the pass is full, so it is compiled without the "does this round hold a record" scalar tests (some 130
  // scalar instructions per wave and pass) and without the per-slot "is there a record" selects; what is staged beyond PS
  // (fewer than BLOCK records) moves to the front for the next pass.  PS = 0: no such pass.
  static constexpr int PS = (BIG && S > BLOCK && (S - BLOCK) % BLOCK == 0) ? S - BLOCK : 0;
  StageLds<W, BIG> &L;
  Buckets B; DevCounters *ctr; int k;
  u32 bound;                                            // upper bound of the records staged (workgroup-uniform)
  u32 cur_j, cur_chunk;                                 // owner thread (tid < TJ_P): the chunk its bucket is being written to
  STAMP_MEMBER
#if defined(TJ_EXP_SINK) && TJ_EXP_SINK >= 4
  u32 exp_cur = 0;
#endif

  __device__ __forceinline__ void start ()
  {
    if (threadIdx.x == 0) L.n = 0;
    if (BIG && threadIdx.x < TJ_P) L.hist[threadIdx.x] = 0;   // (partition_big leaves the counts zeroed for the pass after it)
    bound = 0; cur_j = TJ_EMPTY; cur_chunk = TJ_NOCHUNK;
    lds_barrier ();
  }

  __device__ __forceinline__ void tick () {}

  // `round_max`: workgroup-uniform upper bound of the lanes that bring a record in this call
  __device__ __forceinline__ void put (bool have, u64 c0, u64 c1, u32 base, u32 len10, u32 flag, u64 pos, u32 round_max)
  {
    if (bound + round_max > (u32) S) bound = partition ();
    bound += round_max;
    if (have) {
      // (a uniform-address atomic under a divergent condition: the compiler's atomic optimiser makes it one LDS atomic
      // per wavefront plus a lane prefix -- and does it better than the hand-written ballot / shuffle version did)
      const u32 at = atomicAdd (&L.n, 1u);
      u64 w[W];
      pack_raw<W> (c0, c1, base, len10, flag, k, w);
#pragma unroll
      for (int j = 0; j < WS; j++) L.rec[at * WS + j] = w[j];
      L.bin[at] = (unsigned char) (W == 1 ? bucket_of_rec1 ((u32) w[0], (u32) (w[0] >> 32)) : bucket_of_key (c0, c1, base, len10));
    }
  }

  // a one-word record already packed (scan_fast_kernel); same contract as put
  __device__ __forceinline__ void put1 (bool have, u32 lo, u32 hi, u32 round_max)
  {
    if (bound + round_max > (u32) S) bound = partition ();
    bound += round_max;
    if (have) {
      const u32 at = atomicAdd (&L.n, 1u);
      L.rec[at * WS] = ((u64) hi << 32) | lo;             // (W == 1)
      L.bin[at] = (unsigned char) bucket_of_rec1 (lo, hi);
    }
  }

  // the same in two steps, so that the slot's atomic is in flight while the caller works the record out:
  // reserve1 (round_max) by everybody; then per wave: lds_add_issue (count_addr (), records of the wave, raw); ...;
  // at = lds_collect (raw) + rank in the wave; if (have) store1 (at, lo, hi);
  __device__ __forceinline__ void reserve1 (u32 round_max)
  {
    if (bound + round_max > (u32) S) bound = partition ();
    bound += round_max;
  }
  __device__ __forceinline__ u32 count_addr () const { return (u32) (size_t) (lptr_t) &L.n; }
  __device__ __forceinline__ void store1 (u32 at, u32 lo, u32 hi)
  {
    L.rec[at * WS] = ((u64) hi << 32) | lo;
    L.bin[at] = (unsigned char) bucket_of_rec1 (lo, hi);
  }
  // the same with the hash's weights in VGPRs (h0 = 0x6D2B4F0B, h1 = 0xC5A34D17, m27 = 0x07FFFFFF: see bucket_of_rec1) and no branch: a
  // lane without a record passes at = S + lane, a spare slot
  __device__ __forceinline__ void store1v (u32 at, u32 lo, u32 hi, u32 h0, u32 h1, u32 m27)
  {
    L.rec[at * WS] = ((u64) hi << 32) | lo;
    u32 h = __builtin_amdgcn_udot4 (lo, h0, 0u, false);
    h = __builtin_amdgcn_udot4 (hi & m27, h1, h, false);
    L.bin[at] = (unsigned char) (h ^ (h >> 8));
  }
  __device__ __forceinline__ void store_fields (u32 at, u64 c0, u64 c1, u32 base, u32 len10, u32 flag)
  {
    u64 w[W];
    pack_raw<W> (c0, c1, base, len10, flag, k, w);
#pragma unroll
    for (int j = 0; j < WS; j++) L.rec[at * WS + j] = w[j];
    L.bin[at] = (unsigned char) (W == 1 ? bucket_of_rec1 ((u32) w[0], (u32) (w[0] >> 32)) : bucket_of_key (c0, c1, base, len10));
  }

  // The partition of the fast kernel's sink (BIG), written without per-record branches: a lane whose slot holds no
  // record works on spare bucket TJ_P + lane and spare staging slot S + lane, which exist for that purpose, and only its
  // store to the pool is masked; rounds r with r * BLOCK >= n hold no record at all and are skipped by a scalar branch.
  // (A branch per record costs three scalar instructions and an exec round trip, and scalar instructions are as dear as
  // vector ones on this chip.)
  // returns the records that stay staged (workgroup-uniform)
  __device__ __forceinline__ u32 partition_big (const bool final_pass = false)
  {
    lds_barrier ();                                     // every append so far is in LDS
#if defined(TJ_EXP_SINK) && TJ_EXP_SINK == 1            // experiment builds only: records dropped
    if (threadIdx.x == 0) L.n = 0;
    lds_barrier ();
    return 0u;
#endif
    PSTAMP (9);
    const u32 n = (u32) __builtin_amdgcn_readfirstlane ((int) L.n);
    if constexpr (PS > 0) {
      // (fewer than PS records: the caller's bound had counted candidates that were not recorded -- the round it asked
      // room for fits, n + BLOCK <= S, and the exact count is what it gets back; only finish () empties a partial buffer,
      // so that the loop holds one copy of the pass, the one without tests)
      if (n >= (u32) PS || final_pass) return final_pass ? partition_pass<false> (n) : partition_pass<true> (n);
      lds_barrier ();                                   // (rare path; nobody appends before everybody has read n)
      return n;
    }
    else return partition_pass<false> (n);
  }

  // FULL: the pass takes exactly the first PS staged records (all RR rounds are full); otherwise all n of them
  template <bool FULL>
  __device__ __forceinline__ u32 partition_pass (const u32 n)
  {
    static_assert (!BIG || BLOCK == 2 * TJ_P, "two threads per bucket");
    constexpr int RR = FULL ? PS / BLOCK : R;
    constexpr int H = (RR < 4) ? RR : 4;                // records per thread whose LDS loads are in flight together (2 and 8: the same time)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane (tid >> 6);
    // (five barriers per pass: the bucket counts are zero already -- zeroed below for the next pass, the spare ones are
    // never looked at -- and each of the four waves that own buckets works out the whole prefix for itself)
    u64 w[RR][WS], wrem[WS];
    u32 rk[RR], bb[RR], brem = 0;
#pragma unroll
    for (int r = 0; r < RR; r++)                        // my records ...
      if (FULL || (u32) r * BLOCK < n) {
        const u32 i = (u32) tid + (u32) r * BLOCK;
        bb[r] = L.bin[i];
#pragma unroll
        for (int j = 0; j < WS; j++) w[r][j] = L.rec[i * WS + j];
      }
    if constexpr (FULL) {                               // (what is staged beyond PS: kept in registers until the buffer is free)
      brem = L.bin[(u32) PS + (u32) tid];
#pragma unroll
      for (int j = 0; j < WS; j++) wrem[j] = L.rec[((u32) PS + (u32) tid) * WS + j];
    }
#pragma unroll
    for (int r = 0; r < RR; r++)                        // ... and their rank inside their bucket
      if (FULL || (u32) r * BLOCK < n) {
        const u32 i = (u32) tid + (u32) r * BLOCK;
        if constexpr (!FULL) bb[r] = (i < n) ? bb[r] : (u32) (TJ_P + lane);
#if defined(TJ_EXP_SINK) && TJ_EXP_SINK == 5            // experiment: no rank atomics (wrong results; cost of the LDS atomics)
        rk[r] = 0; if (r == 0) L.hist[bb[r]] = 7u;
#else
        rk[r] = atomicAdd (&L.hist[bb[r]], 1u);
#endif
      }
    lds_barrier ();
    PSTAMP (10);
    if (tid == 0) L.n = FULL ? n - (u32) PS : 0u;       // (everybody has read it; the next appends come after the last barrier below)
    u32 cnt = 0, p0 = 0, off = 0;
    if (wave < TJ_P / 64) {
      // exclusive prefix of the 256 bucket counts, four per lane, in each of the waves 0..3 (no exchange between them);
      // the wave writes the quarter that holds its own buckets and reads its bucket's entry back
      const uint4 h4 = *reinterpret_cast<const uint4 *> (&L.hist[4 * lane]);
      const u32 tot = h4.x + h4.y + h4.z + h4.w;
      const u32 e0 = wave_inclusive_scan (tot) - tot;
      if ((lane >> 4) == wave) *reinterpret_cast<uint4 *> (&L.offs[4 * lane]) = make_uint4 (e0, e0 + h4.x, e0 + h4.x + h4.y, e0 + h4.x + h4.y + h4.z);
      cnt = L.hist[tid];
      off = L.offs[tid];
      // reserve the bucket's run: the global atomic's round trip runs under the LDS permutation below (its result is
      // first looked at after that)
#if defined(TJ_EXP_SINK) && TJ_EXP_SINK >= 4            // experiment builds only: a private cursor instead of the global atomic (results are wrong)
      if (cnt) { p0 = exp_cur; exp_cur = (p0 + cnt) & 255u; }
#else
      if (cnt) p0 = atomicAdd (&B.cursors[tid * TJ_CSTRIDE], cnt);
#endif
    }
    else if (!FULL && wave == TJ_P / 64) L.offs[tid] = (u32) S + (u32) lane;   // (the spare entries: rank 0 lands on staging slot S + lane)
    lds_barrier ();
    PSTAMP (11);
    if (tid < TJ_P) L.hist[tid] = 0;                    // (for the next pass)
    {
      u32 dst[RR];
#pragma unroll
      for (int r = 0; r < RR; r++) if (FULL || (u32) r * BLOCK < n) dst[r] = L.offs[bb[r]];
#pragma unroll
      for (int r = 0; r < RR; r++)                      // in-place permutation into bucket order (records are in registers)
        if (FULL || (u32) r * BLOCK < n) {
          const u32 d = FULL ? dst[r] + rk[r] : dst[r] + ((bb[r] < (u32) TJ_P) ? rk[r] : 0u);
#pragma unroll
          for (int j = 0; j < WS; j++) L.rec[d * WS + j] = w[r][j];
          L.bin[d] = (unsigned char) bb[r];
        }
    }
    PSTAMP (12);
    if (wave < TJ_P / 64) {                             // where the reserved run lives: byte addresses of "sorted slot 0" for both parts
      // (tried in round 3: the usual run -- inside the chunk its bucket was written to last time, opening no new one -- worked
      // out without a branch and the rest behind one wave-wide test: 1.5 M vector and 0.5 M scalar instructions MORE per
      // launch, 1 % slower; the exec regions below are cheaper than they look)
      u64 a1 = 0, a2 = 0;
      u32 thr = 0;
#if defined(TJ_EXP_SINK) && TJ_EXP_SINK >= 4            // (experiment: every workgroup writes to a 4 KB region of its own per bucket)
      if (cnt && cnt <= 256u) { a1 = (u64) (size_t) B.pool + (((((u64) tid * 512u + (blockIdx.x & 511u)) * 512u) + p0) << 3) - 8ull * off; thr = off + cnt; }
      if (0) {
#else
      if (cnt) {
#endif
        const u32 ch = (u32) TJ_CH0 << B.ch_shift;
        bucket_claim_ahead (B, (u32) tid, p0, cnt, ctr);
        const u32 j0 = chunk_of_pos (B, p0), j1 = chunk_of_pos (B, p0 + cnt - 1);
        if (j0 != cur_j) { cur_j = j0; cur_chunk = bucket_chunk_id (B, (u32) tid, j0, true, ctr); }
        if (cur_chunk != TJ_NOCHUNK) a1 = (u64) (size_t) (B.pool + ((u64) cur_chunk * ch + (p0 - j0 * ch)) * W) - (8ull * W) * off;
        thr = off + cnt;
        if (j1 != j0) {                                 // the run crosses into the next chunk
          thr = off + (j1 * ch - p0);
          cur_j = j1; cur_chunk = bucket_chunk_id (B, (u32) tid, j1, true, ctr);
          if (cur_chunk != TJ_NOCHUNK) a2 = (u64) (size_t) (B.pool + ((u64) cur_chunk * ch) * W) - (8ull * W) * thr;
        }
      }
      // one 16-byte entry per bucket, read with one LDS instruction per record in the copy-out: [a1 low | a1 high (16 bits:
      // device addresses have 48) + thr << 16 | a2]   (gbase and gbase2 lie back to back: 256 x 16 bytes)
      reinterpret_cast<uint4 *> (L.gbase)[tid] = make_uint4 ((u32) a1, (u32) (a1 >> 32) | (thr << 16), (u32) a2, (u32) (a2 >> 32));
    }
    PSTAMP (13);
    // Everything this wave has asked global memory for so far (the cursor's atomic, chunk ids, the tile's prefetch) is waited
    // for HERE, with the compiler's own instruction, before the first store below: from now on nothing is in flight that
    // anybody waits for but stores -- otherwise the compiler, unsure on which path a chunk id was fetched, puts a vmcnt(0)
    // where the tile loop's paths meet, in the classification of every tile, and that one waits for the stores too.
    __builtin_amdgcn_s_waitcnt (0x0F70);                // vmcnt(0) alone
    lds_barrier ();
    PSTAMP (14);
#pragma unroll
    for (int r0 = 0; r0 < RR; r0 += H)
      if (FULL || (u32) r0 * BLOCK < n) {
        u32 cb[H], cthr[H];
        u64 ca1[H], ca2[H], cw[H][WS];
#pragma unroll
        for (int h = 0; h < H; h++) if (r0 + h < RR) cb[h] = L.bin[(u32) tid + (u32) (r0 + h) * BLOCK];
#pragma unroll
        for (int h = 0; h < H; h++) if (r0 + h < RR) {
          const u32 i = (u32) tid + (u32) (r0 + h) * BLOCK;
          const u32 b = cb[h] & (u32) (TJ_P - 1);       // (slots past n hold stale bytes: any bucket will do, the store is masked)
          const uint4 e = reinterpret_cast<const uint4 *> (L.gbase)[b];
          cthr[h] = e.y >> 16; ca1[h] = ((u64) (e.y & 0xFFFFu) << 32) | e.x; ca2[h] = ((u64) e.w << 32) | e.z;
#pragma unroll
          for (int j = 0; j < WS; j++) cw[h][j] = L.rec[i * WS + j];
        }
#pragma unroll
        for (int h = 0; h < H; h++) if (r0 + h < RR) {  // sorted slot i -> its place in the bucket's run (coalesced per run)
          const u32 i = (u32) tid + (u32) (r0 + h) * BLOCK;
          const u64 a = (i < cthr[h]) ? ca1[h] : ca2[h];
#if defined(TJ_EXP_SINK) && TJ_EXP_SINK == 3
          const bool st = a == 0x123456789ull;
#else
          const bool st = (FULL || i < n) && a != 0ull;
#endif
          if (st) {
            // (a pointer into the global address space, said so: made from an integer it is a generic pointer to the compiler, the
            // stores become flat_store -- which counts in lgkmcnt as well, completes out of order, and makes every later
            // wait of the wave a full drain of both counters)
            typedef __attribute__((address_space(1))) u64 *gwords_t;
            gwords_t q = (gwords_t) (a + (8ull * W) * i);
#pragma unroll
            for (int j = 0; j < WS; j++) q[j] = cw[h][j];
            if (WS < W) q[W - 1] = 0;
          }
        }
      }
    lds_barrier ();                                     // the staging buffer is free again
    PSTAMP (15);
    if constexpr (FULL) {
      // what was staged beyond PS opens the next pass's buffer (the appends that follow go to slots n - PS and up: no
      // slot is written from both sides)
      if ((u32) PS + (u32) tid < n) {
#pragma unroll
        for (int j = 0; j < WS; j++) L.rec[(u32) tid * WS + j] = wrem[j];
        L.bin[tid] = (unsigned char) brem;
      }
      return n - (u32) PS;
    }
    return 0u;
  }

  __device__ __forceinline__ u32 partition (const bool final_pass = false)
  {
    if constexpr (BIG) return partition_big (final_pass);
    // (scan_bins_kernel, 80 registers: one record at a time)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    lds_barrier ();                                     // every append so far is in LDS
    const u32 n = L.n;
    if (tid < TJ_P) L.hist[tid] = 0;
    lds_barrier ();
    if (tid == 0) L.n = 0;                              // (everybody has read it; the next appends come after the last barrier below)
    u64 w[R][WS];
    u32 rk[R], bb[R];
#pragma unroll
    for (int r = 0; r < R; r++) {                       // my records and their rank inside their bucket
      const u32 i = (u32) tid + (u32) r * BLOCK;
      bb[r] = TJ_EMPTY;
      if (i < n) {
        bb[r] = L.bin[i];
#pragma unroll
        for (int j = 0; j < WS; j++) w[r][j] = L.rec[i * WS + j];
        rk[r] = atomicAdd (&L.hist[bb[r]], 1u);
      }
    }
    lds_barrier ();
    u32 cnt = 0, incl = 0;
    if (tid < TJ_P) {                                   // exclusive prefix of the bucket counts (waves 0..3)
      cnt = L.hist[tid];
      incl = wave_inclusive_scan (cnt);
      if (lane == 63) L.wsum[wave] = incl;
    }
    lds_barrier ();
    u32 p0 = 0;
    if (tid < TJ_P) {
      u32 wbase = 0;
      for (int v = 0; v < wave; v++) wbase += L.wsum[v];
      L.offs[tid] = wbase + incl - cnt;
      // reserve the bucket's run: the global atomic's round trip runs under the LDS permutation below (its result is
      // first looked at after that)
      if (cnt) p0 = atomicAdd (&B.cursors[tid * TJ_CSTRIDE], cnt);
    }
    lds_barrier ();
#pragma unroll
    for (int r = 0; r < R; r++)                         // in-place permutation into bucket order (records are in registers)
      if (bb[r] != TJ_EMPTY) {
        const u32 d = L.offs[bb[r]] + rk[r];
#pragma unroll
        for (int j = 0; j < WS; j++) L.rec[d * WS + j] = w[r][j];
        L.bin[d] = (unsigned char) bb[r];
      }
    if (tid < TJ_P && cnt) {                            // where the reserved run lives
      const u32 ch = (u32) TJ_CH0 << B.ch_shift;
      bucket_claim_ahead (B, (u32) tid, p0, cnt, ctr);
      const u32 j0 = chunk_of_pos (B, p0), j1 = chunk_of_pos (B, p0 + cnt - 1);
      if (j0 != cur_j) { cur_j = j0; cur_chunk = bucket_chunk_id (B, (u32) tid, j0, true, ctr); }
      L.gbase[tid] = (cur_chunk == TJ_NOCHUNK) ? ~0ull : (u64) cur_chunk * ch + (p0 - j0 * ch);
      u32 sp = cnt;
      u64 g2 = ~0ull;
      if (j1 != j0) {                                   // the run crosses into the next chunk
        sp = j1 * ch - p0;
        cur_j = j1; cur_chunk = bucket_chunk_id (B, (u32) tid, j1, true, ctr);
        if (cur_chunk != TJ_NOCHUNK) g2 = (u64) cur_chunk * ch;
      }
      L.split[tid] = sp; L.gbase2[tid] = g2;
    }
    lds_barrier ();
#pragma unroll
    for (int r = 0; r < R; r++) {                       // sorted slot i -> its place in the bucket's run (coalesced per run)
      const u32 i = (u32) tid + (u32) r * BLOCK;
      if (i < n) {
        const u32 b = L.bin[i], o = i - L.offs[b], sp = L.split[b];
        const u64 g = (o < sp) ? L.gbase[b] : L.gbase2[b];
        if (g != ~0ull) {
          u64 *q = B.pool + (g + (o < sp ? o : o - sp)) * W;
#pragma unroll
          for (int j = 0; j < WS; j++) q[j] = L.rec[i * WS + j];
          if (WS < W) q[W - 1] = 0;
        }
      }
    }
    lds_barrier ();                                     // the staging buffer is free again
    return 0u;
  }

  __device__ __forceinline__ void finish () { partition (true); }
};

#define TJ_SB_BLOCK 512
#define TJ_SB_TILE  8192
#ifndef TJ_SB_WG_PER_CU
#define TJ_SB_WG_PER_CU 3               // grid = 3 workgroups per CU: 2 are resident (65 KB of LDS each), the queued third evens out the tail
#endif

// the qualifying runs of a non-ACGTU byte that scan_tiles has listed: their context is the tract before them (reference
// src/hopo_counter.c:246-248).  Entries first, first + step, ...
template <int W>
__device__ __forceinline__ void nrun_fixup_entries (const uint8_t *__restrict__ seq, long n_bytes, int k, int mprime, Buckets BK, DevCounters *ctr,
                                                    const FixEntry *fix, u32 fix_cap, int par, u64 first, u64 step)
{
  u64 n_fix = __hip_atomic_load (&ctr->lc[par].n_fix, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (n_fix > fix_cap) n_fix = fix_cap;
  for (u64 i = first; i < n_fix; i += step) {
    u64 c0, c1; u32 base, flag;
    if (stale_context (seq, n_bytes, fix[i].pos, k, mprime, c0, c1, base, flag))
      bucket_insert_slow<W> (c0, c1, base, (u32) ((u64) fix[i].len & 0x3FFull), flag, k, BK, ctr);
    else atomicAdd (&ctr->n_undefined, 1ull);
  }
}

template <int W>
__global__ __launch_bounds__ (TJ_SB_BLOCK, 6)
void scan_bins_kernel (const uint8_t *__restrict__ seq, long n_bytes, long n_tiles, int k, int mprime,
                       Buckets BK, DevCounters *ctr, FixEntry *fix, u32 fix_cap, int par, TileSrc src)
{
  __shared__ TileLds<TJ_SB_BLOCK, TJ_SB_TILE> T;
  __shared__ uint4 raw[TileLds<TJ_SB_BLOCK, TJ_SB_TILE>::NRAW];
  __shared__ StageLds<W> SL;
  // (behind the fast kernel: nothing was left over -- the usual case -- or a handful of tiles: one workgroup per listed tile is
  // plenty, the others leave here instead of paying for a sink's start and finish and, below, a device-scope fence each: with
  // 24 tiles listed the launch took 120 us for its 768 fences)
  const u32 n_listed = src.list ? ctr->lc[par].n_slow : 0u;
  const u32 n_active = src.list ? min (n_listed, (u32) gridDim.x) : (u32) gridDim.x;
  if (src.list && blockIdx.x >= n_active) return;
  StageSink<W, TJ_SB_BLOCK> sink = {SL, BK, ctr, k, 0u, 0u, 0u};
  sink.start ();
  scan_tiles<TJ_SB_BLOCK, TJ_SB_TILE, 2> (seq, n_bytes, n_tiles, k, mprime, T, raw, sink, ctr, fix, fix_cap, par, src);
  sink.finish ();
  if (src.list) {
    // Behind the fast kernel the fix-up of the listed runs is this kernel's last act instead of a launch of its own (which
    // cost 4 us of stream time per scan for nothing: with no tile on the list, the usual case, every workgroup has left at
    // the top and there is nothing to fix up).  The workgroup that finishes last does it: everybody's entries are released
    // by a device-scope fence in front of the ticket and acquired by one behind it -- a path taken by streams with
    // lower case, IUPAC codes or countable runs of 'N', where a fence's price does not matter.
    __shared__ u32 s_last;
    __threadfence ();
    __syncthreads ();
    if (threadIdx.x == 0) s_last = (atomicAdd (&ctr->lc[par].ticket, 1u) == n_active - 1u) ? 1u : 0u;
    __syncthreads ();
    if (s_last) {
      __threadfence ();
      nrun_fixup_entries<W> (seq, n_bytes, k, mprime, BK, ctr, fix, fix_cap, par, threadIdx.x, blockDim.x);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Fast scan for the streams that make up almost all input: tiles whose bytes are all upper-case A C G T or the read
// delimiter.  Same results as scan_tiles (reference: src/hopo_counter.c:219-258,285-307); any tile it cannot vouch for --
// a window that touches the stream's ends, a byte outside the five, more candidates than its list holds -- goes on a
// list that scan_bins_kernel works through afterwards (TileSrc), so nothing here has to know about lower case, U, N,
// the stale-context rule or stream edges.
//
// What makes it fast is the instruction mix.  On gfx950 a wave-wide and / or / xor / add / sub / right shift / mov /
// v_bitop3 whose operands are VGPRs, inline constants or (VOP1/VOP2) literals issues every 2 cycles; everything else --
// any SGPR operand, left shifts, v_perm, v_dot4, v_alignbit, DPP, v_lshl_or, v_ffbl, compares -- takes 4
// (tools/ubench/valu_rate3.hip).  So masks live in VGPRs, boolean algebra goes through v_bitop3, and the 4-cycle
// instructions are kept for what only they can do (byte table look-ups, bit gathers, funnel shifts).
//
// Geometry: 512 lanes x 32 bytes = a 16 KiB window per tile; the first 32 bytes (>= max k) and the last 64 are halo, the
// 16288 between them are the tract starts the tile owns (a tract whose end + k leaves the window walks the stream in
// global memory, like scan_tiles does).  A byte is one of the five iff the table look-up on its low three bits (all
// different: A 1, '\n' 2, C 3, T 4, G 7) gives the byte back; those three bits are then all that run detection needs
// (equal bytes <=> equal low bits), and code / letter / run-start bits are gathered with one v_dot4 per word each.
// Each lane keeps the run-start and letter masks of its 32 positions in registers and finds its candidates from them
// (plus one DPP move for the next lane's bits); the rest is scan_tiles' phase 3 with 32-bit windows: one funnel read of
// the run-start plane for the run end, one of the letter plane for "k letters on both sides" (whenever 2k + length <= 32),
// one of the code plane per flank.

// ---- sink 3: the record log (one-word records: k <= 12) ---------------------------------------------------------------
// scan_fast_kernel<1> does not partition: a recorded tract goes straight from the lane that worked it out to the next
// free word of a linear log -- blocks of TJ_LOGB records that a workgroup takes from a counter one ahead of need (the
// atomic's result is picked up a tile later), slot = the workgroup's running count (one LDS atomic per wave and round, as
// the staging buffer's was).  partition_log_kernel then reads the log back at memory speed and distributes it over the
// hash buckets with the counting sort of StageSink::partition_pass -- a kernel of its own with six waves per SIMD, in
// which that sort's barriers and atomic round trips hide behind other workgroups, instead of a pass that stalls a
// scanning workgroup 32 times per launch with 37 KB of its LDS and 45 of its registers set aside for it.
#define TJ_LOGB_SHIFT 13
#define TJ_LOGB       (1u << TJ_LOGB_SHIFT)             // records per log block (a tile brings at most FK_MAXCAND = 4096)
#define TJ_LOG_AHEAD  (2u * 4096u + 1024u)              // a block is asked for when the count may reach it within two tiles
struct LogSpace
{
  u64 *log;             // n_blocks * TJ_LOGB records, then 64 scratch words per workgroup of the scan
  u32 *next;            // blocks handed out
  u32 *next_other;      // the next launch's counter (launches alternate between two): zeroed by this launch
  u32 *count;           // [n_blocks] records in each block (written by the workgroup that filled it)
  u32 n_blocks;
};

struct LogLds
{
  alignas (16) u64 blk[4];              // byte address of the block that holds the workgroup's records j * TJ_LOGB ..., j & 3
  u32 id[4];
  u32 n;                                // records appended by this workgroup
};

// W: words per record (1: k <= 12, 2: k <= 28).  A block is TJ_LOGB words whatever W is: 8192 one-word or 4096 two-word records.
template <int W>
struct LogSink
{
  static_assert (W == 1 || W == 2, "the log holds one-word and two-word records");
  static constexpr int S = 0x7FFFFFF0;                  // (no buffer to fill up: every tile "fits")
  static constexpr u32 RS = TJ_LOGB_SHIFT - (W == 2 ? 1 : 0), RB = 1u << RS;     // records per block
  LogLds &L;
  LogSpace G; DevCounters *ctr; int k;
  u32 bound;
  u32 upto, pend, pend_id;                              // thread 0: last block index with an address in L.blk; a reservation on its way
  u64 scratch;                                          // where lanes without a record write (byte address, 64 words per workgroup)
  STAMP_MEMBER

  __device__ __forceinline__ u64 block_addr (u32 id) const
  {
    // (cannot happen: the log is sized for a tract every m' bytes.  If it did: the flag makes the scan fail; until then the
    // records go to block 0, inside the log -- the scratch words are 64 per workgroup, not a block)
    if (id >= G.n_blocks) { ctr->overflow = 1u; return (u64) (size_t) G.log; }
    return (u64) (size_t) G.log + (((u64) id << TJ_LOGB_SHIFT) << 3);
  }
  __device__ __forceinline__ void start ()
  {
    bound = 0; pend = 0; pend_id = 0; upto = 1;
    scratch = (u64) (size_t) G.log + ((((u64) G.n_blocks << TJ_LOGB_SHIFT) + 64ull * blockIdx.x) << 3);
    if (threadIdx.x == 0) {
      L.n = 0;
      const u32 b0 = atomicAdd (G.next, 2u);
      L.id[0] = b0; L.id[1] = b0 + 1u; L.id[2] = ~0u; L.id[3] = ~0u;
      L.blk[0] = block_addr (b0); L.blk[1] = block_addr (b0 + 1u); L.blk[2] = scratch; L.blk[3] = scratch;
    }
    lds_barrier ();
  }
  // top of a tile, every outstanding load of the wave waited for: a block that was asked for a tile ago has its address now
  // (the ring slot it takes held block upto - 3, which is full -- or will be when the kernel ends: slots are handed out densely)
  __device__ __forceinline__ void collect ()
  {
    if (threadIdx.x == 0 && pend) {
      upto++;
      const u32 slot = upto & 3u, old = L.id[slot];
      if (old < G.n_blocks) G.count[old] = RB;
      L.id[slot] = pend_id; L.blk[slot] = block_addr (pend_id);
      pend = 0;
    }
  }
  // `n`: the workgroup's count, read while nobody appends.  (A tile brings at most FK_MAXCAND = 4096 records, a block
  // holds at least as many: at most one new block per tile, asked for TJ_LOG_AHEAD records -- two tiles -- ahead, so that the
  // tile after the asking one, which picks the address up at its top, is the first that can need it; the ring slot it
  // takes held block j - 4 while the appends are at j - 3 or later.)
  __device__ __forceinline__ void tile_top (u32 n)
  {
    if (threadIdx.x == 0 && !pend && ((n + TJ_LOG_AHEAD) >> RS) > upto) { pend_id = atomicAdd (G.next, 1u); pend = 1u; }
  }
  __device__ __forceinline__ void reserve1 (u32) {}
  __device__ __forceinline__ u32 count_addr () const { return (u32) (size_t) (lptr_t) &L.n; }
  __device__ __forceinline__ u64 place (u32 at) const { return L.blk[(at >> RS) & 3u] + (8ull * W) * (at & (RB - 1u)); }
  __device__ __forceinline__ void store1 (u32 at, u32 lo, u32 hi)
  {
    typedef __attribute__((address_space(1))) u64 *gwords_t;
    *(gwords_t) place (at) = ((u64) hi << 32) | lo;
  }
  // branch-free: a lane without a record writes to the workgroup's scratch words
  __device__ __forceinline__ void store1ok (bool ok, u32 at, u32 lo, u32 hi, u32 lane)
  {
    typedef __attribute__((address_space(1))) u64 *gwords_t;
    const u64 a = place (at);
    *(gwords_t) (ok ? a : scratch + 8ull * lane) = ((u64) hi << 32) | lo;
  }
  // a record from its fields (the two-word kernel's phase 3)
  __device__ __forceinline__ void store_fields (u32 at, u64 c0, u64 c1, u32 base, u32 len10, u32 flag)
  {
    typedef __attribute__((address_space(1))) u64 *gwords_t;
    u64 w[W];
    pack_raw<W> (c0, c1, base, len10, flag, k, w);
    gwords_t q = (gwords_t) place (at);
#pragma unroll
    for (int j = 0; j < W; j++) q[j] = w[j];
  }
  __device__ __forceinline__ void finish ()
  {
    lds_barrier ();
    collect ();
    if (threadIdx.x == 0) {
      const u32 n = L.n;
      for (u32 j = (upto >= 3u ? upto - 3u : 0u); j <= upto; j++) {
        const u32 id = L.id[j & 3u];
        if (id < G.n_blocks) G.count[id] = (n > (j << RS)) ? min (n - (j << RS), RB) : 0u;
      }
    }
  }
};

#define FK_BLOCK    512
#ifndef FK_WG_PER_CU
#define FK_WG_PER_CU 2
#endif
#ifndef FK_LOG_WG_PER_CU
#define FK_LOG_WG_PER_CU 3              // the log variant (no staging buffer: 35 KB of LDS)
#endif
// FK_WG_PER_CU: 73 KB of LDS each (the 4096-record staging buffer is worth more than a third workgroup)
#define FK_UNIT     32                  // stream bytes per lane
#define FK_WIN      (FK_BLOCK * FK_UNIT)
#define FK_HL       32
#define FK_HR       64
#define FK_OWN      (FK_WIN - FK_HL - FK_HR)
static_assert (FK_OWN == TJ_LIST_FOWN, "scan_tiles' list mode is compiled for the fast kernel's tile size");
#ifndef FK_MAXCAND
#define FK_MAXCAND  4096
#endif
// FK_MAXCAND: (a tile with more candidates goes to the generic kernel: more than one per 4 positions)
#ifndef FK_GROUP
#define FK_GROUP    4                   // tiles per work-counter atomic (8: 1.5 % slower, a longer tail; 2: 5 % slower)
#endif

struct FastLds
{
  u32 code[FK_WIN / 16 + 4];            // 2-bit codes, 16 positions per word (+ zeroed pad for funnel reads)
  u32 st[FK_WIN / 32 + 4];              // run starts (byte differs from its predecessor)
  u32 lt[FK_WIN / 32 + 4];              // letters (not a read delimiter)
  unsigned short cand[FK_MAXCAND + 2 + 64];         // (+ 64: a spare entry per lane, see FK_CAND_UNROLL)
  u32 ncand[4];                         // per tile, three in rotation (zeroed two tiles ahead)
  u32 bad[4];                           // per tile, likewise: some byte outside ACGT\n
  u32 grp[2];
  u32 sbuf[32];                         // tiles given up, on their way to the slow list (one global atomic per 32)
  u32 np[FK_WIN / 32 + 4];              // second chance of a tile (see the kernel): positions that hold an 'N'
  u32 ncand2;                           // its candidate count
};

// v_bitop3_b32 truth tables: bit (a << 2 | b << 1 | c) of the immediate = f (a, b, c)
#define BITOP3_XOR_AND   0x48           // (a ^ c) & b
#define BITOP3_OR_XOR    0xF6           // a | (b ^ c)

// reverse complement of a k-mer of at most 16 bases with the constants in VGPRs: bit reverse, then swap the two bits of
// every base and complement in one v_bitop3 (rs = 32 - 2k)
__device__ __forceinline__ u32 revcomp_v32 (u32 x, u32 m55, u32 rs)
{
  const u32 y = __brev (x);
  // ~(((y >> 1) & 0x5555...) | ((y << 1) & 0xAAAA...)): a = y >> 1, b = y + y, c = mask -> ~(c ? a : b)
  const u32 z = __builtin_amdgcn_bitop3_b32 (y >> 1, y + y, m55, 0x1B);
  return z >> rs;
}

// Phase 3 of scan_fast_kernel for any k: run end from the run-start plane (32 positions first, then 64-bit windows), the
// two flanks checked against the letter plane and pulled out of the code plane separately (32-bit arithmetic for
// k <= 16), and past the window's end a walk through the stream in global memory -- scan_tiles' logic on this kernel's
// planes.  Returns false if the tract is not recorded.
template <bool K32>
__device__ __forceinline__ bool fast_tract (const FastLds &T, const uint8_t *__restrict__ seq, long n_bytes, long g0, int s, int k, int mprime,
                                            u64 &c0, u64 &c1, u32 &base, u32 &len10, u32 &flag, bool n_tile = false)
{
  int e = -1;
  {
    const u32 ns32 = bits32 (T.st, s + 1);
    if (ns32) e = s + __ffs ((int) ns32) - 1;
    else {
      const u64 ns = bits64 (T.st, s + 1);
      if (ns) e = s + __ffsll ((long long) ns) - 1;
      else for (int p = s + 65; p < FK_WIN; p += 64) {
        const u64 n2 = bits64 (T.st, p);
        if (n2) { e = p + __ffsll ((long long) n2) - 2; break; }
      }
    }
  }
  const long gs = g0 + s;
  long len;
  if (e >= 0 && e + k < FK_WIN) {                       // everything needed is in LDS
    len = e - s + 1;
    const u32 kb32 = (k >= 32) ? 0xFFFFFFFFu : (1u << k) - 1u;
    if ((((~bits32 (T.lt, s - k) | ~bits32 (T.lt, e + 1)) & kb32) != 0u) || len < mprime) return false;
    const u32 cb = (T.code[s >> 4] >> (2 * (s & 15))) & 3u;
    // (a tile with 'N's: an 'N' in a flank has code 0 in the plane, which is what the reference packs for it read forwards;
    // read backwards it packs 0 as well, not the complement -- its forward code is made 3 before complementing, as
    // `canonicalise` does for every non-ACGTU byte)
    u64 nl2 = 0, nr2 = 0;
    if (n_tile) {
      const u32 nl = bits32 (T.np, s - k) & kb32, nr = bits32 (T.np, e + 1) & kb32;
      if (nl | nr) { nl2 = spread_pairs (nl); nr2 = spread_pairs (nr); }
    }
    if (K32 || k <= 16) {
      const u32 km32 = (k >= 16) ? 0xFFFFFFFFu : (1u << (2 * k)) - 1u;
      const u32 l32 = bits32 (T.code, 2 * (s - k)) & km32, r32 = bits32 (T.code, 2 * (e + 1)) & km32;
      if (cb < 2u) { c0 = l32; c1 = r32; base = cb; flag = 1u; }
      else { c0 = revcomp_k32 (r32 | (u32) nr2, k); c1 = revcomp_k32 (l32 | (u32) nl2, k); base = 3u - cb; flag = 2u; }
    }
    else {
      const u64 km = kmask (k);
      const u64 left = bits64 (T.code, 2 * (s - k)) & km, right = bits64 (T.code, 2 * (e + 1)) & km;
      if (cb < 2u) { c0 = left; c1 = right; base = cb; flag = 1u; }
      else { c0 = revcomp_k (right | nr2, k); c1 = revcomp_k (left | nl2, k); base = 3u - cb; flag = 2u; }
    }
  }
  else {                                                // the tract runs past the window: walk the stream (rare)
    const u32 b = stream_byte (seq, n_bytes, gs);
    long ge = gs;
    while (stream_byte (seq, n_bytes, ge + 1) == b) ge++;
    len = ge - gs + 1;
    u64 left, right; u32 linv, rinv;
    // (outside the window bytes may be anything: non-ACGTU flank bases pack as 0 in both orientations)
    if (!flanks_from_stream (seq, n_bytes, gs, ge, k, left, right, linv, rinv) || len < mprime) return false;
    canonicalise (left, right, linv, rinv, byte_code (b), k, c0, c1, base, flag);
  }
  len10 = (u32) ((u64) len & 0x3FFull);
  return true;
}

// the same, out of line, for the tracts that do not fit the one-word kernel's straight-line path (k + length + k > 32).
// Returns the packed record, 0 if the tract is not recorded (a record is never 0; by value: a reference parameter of a
// function that is not inlined would put the caller's variables into scratch memory, in every round).
__device__ __noinline__ u64 fast_general_tract (const FastLds &T, const uint8_t *__restrict__ seq, long n_bytes, long g0, int s, int k, int mprime, bool n_tile)
{
  u64 c0, c1; u32 base, len10, flag;
  if (!fast_tract<true> (T, seq, n_bytes, g0, s, k, mprime, c0, c1, base, len10, flag, n_tile)) return 0ull;
  return pack_rec1 ((u32) c0, (u32) c1, base, len10, flag);
}

// two LDS-DMA loads of 1 KiB each (lane l: 16 bytes at sbase + voff, voff = 16 l) with a wave-uniform base address in
// SGPRs and one M0 write: the instruction offset of the second load moves its global address and its LDS address alike
__device__ __forceinline__ void lds_dma32 (const uint8_t *sbase, u32 voff, void *lds_wave_base)
{
  const u32 m0v = (u32) (size_t) (lptr_t) lds_wave_base;
  // (M0 is the compiler's: as an operand bound to that register it writes it itself, one s_mov, and knows that it did)
  asm volatile ("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024"
                :: "v"(voff), "s"(sbase), "{m0}"(m0v) : "memory");
}

template <int W, bool LOG = false>
__global__ __launch_bounds__ (FK_BLOCK, FK_BLOCK * (LOG ? FK_LOG_WG_PER_CU : FK_WG_PER_CU) / 256)
void scan_fast_kernel (const uint8_t *__restrict__ seq, long n_bytes, long n_ftiles, int k, int mprime,
                       Buckets BK, DevCounters *ctr, u32 *__restrict__ slow_list, int par, int all_slow, LogSpace LG)
{
  static_assert (!LOG || W <= 2, "the record log holds one-word and two-word records");
  __shared__ FastLds T;
  __shared__ uint4 raw[FK_WIN / 16];
  __shared__ std::conditional_t<LOG, LogLds, StageLds<W, true>> SL;
  auto make_sink = [&] () {
    if constexpr (LOG) return LogSink<W> {SL, LG, ctr, k, 0u, 0u, 0u, 0u, 0ull};
    else return StageSink<W, FK_BLOCK, true> {SL, BK, ctr, k, 0u, 0u, 0u};
  };
  auto sink = make_sink ();
  sink.start ();

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane (tid >> 6);
  // masks in VGPRs (an SGPR or literal operand would halve the issue rate of the instructions that use them)
  u32 M07, M06, M40;
  asm ("v_mov_b32 %0, 0x07070707" : "=v"(M07));
  asm ("v_mov_b32 %0, 0x06060606" : "=v"(M06));
  asm ("v_mov_b32 %0, 0x40404040" : "=v"(M40));
  u32 M55, vk, vk2, vkm, vrs, v32, vmp;                 // likewise the uniform values that phase 3 combines with per-lane data
  asm ("v_mov_b32 %0, 0x55555555" : "=v"(M55));
  asm ("v_mov_b32 %0, %1" : "=v"(vk) : "s"(k));
  asm ("v_mov_b32 %0, %1" : "=v"(vk2) : "s"(2 * k));
  asm ("v_mov_b32 %0, %1" : "=v"(vkm) : "s"(W == 1 ? (1u << (2 * k)) - 1u : 0u));     // (the straight-line phase 3 is for k <= 12)
  asm ("v_mov_b32 %0, %1" : "=v"(vrs) : "s"(W == 1 ? 32 - 2 * k : 0));
  asm ("v_mov_b32 %0, 32" : "=v"(v32));
  asm ("v_mov_b32 %0, %1" : "=v"(vmp) : "s"(mprime));
  u32 vm3, vm4, vh0, vh1, vm27;                             // "mprime >= 3", "mprime >= 4" as masks; the bucket hash's weights
  asm ("v_mov_b32 %0, %1" : "=v"(vm3) : "s"(mprime >= 3 ? 0xFFFFFFFFu : 0u));
  asm ("v_mov_b32 %0, %1" : "=v"(vm4) : "s"(mprime >= 4 ? 0xFFFFFFFFu : 0u));
  asm ("v_mov_b32 %0, 0x6D2B4F0B" : "=v"(vh0));
  asm ("v_mov_b32 %0, 0xC5A34D17" : "=v"(vh1));
  asm ("v_mov_b32 %0, 0x07FFFFFF" : "=v"(vm27));
  u32 own = (tid == 0 || tid >= (FK_WIN - FK_HR) / FK_UNIT) ? 0u : 0xFFFFFFFFu;   // halo lanes own no tract start
  asm volatile ("" : "+v"(own));                        // (a VGPR, not a condition that is looked up in spilled SGPRs per tile)
  const u32 voff16 = 16u * (u32) lane;

  // Scalar instructions are not free on this chip (about 4.6 cycles each per SIMD, tools/ubench/valu_salu.hip), so the
  // loop's bookkeeping is kept to 32-bit scalars: a launch covers less than 2 GiB (scan_device_piece), tile numbers and
  // byte offsets fit an int.  Tiles 1 .. t_hi have their whole window (and the 16 bytes in front) inside the stream.
  const int nt_all = (int) n_ftiles, nb = (int) n_bytes;
  const int t_hi = (nb >= FK_WIN - FK_HL) ? (nb - (FK_WIN - FK_HL)) / FK_OWN : 0;
  if (all_slow) {                                       // (tests: every tile goes to the generic kernel)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      for (int t = 0; t < nt_all; t++) slow_list[t] = (u32) t;
      ctr->lc[par].n_slow = (u32) nt_all; ctr->lc[par].work = (u32) nt_all;
      ctr->lc[par ^ 1].n_fix = 0; ctr->lc[par ^ 1].work = 0; ctr->lc[par ^ 1].n_slow = 0; ctr->lc[par ^ 1].work_slow = 0; ctr->lc[par ^ 1].ticket = 0;
      if constexpr (LOG) *LG.next_other = 0u;
    }
    sink.finish ();
    return;
  }
  const u32 ncand_addr = (u32) (size_t) (lptr_t) &T.ncand[0];

  if (wave == 0) {
    if (lane == 0) {
      T.grp[0] = atomicAdd (&ctr->lc[par].work, (u32) FK_GROUP);
      if (blockIdx.x == 0) {
        ctr->lc[par ^ 1].n_fix = 0; ctr->lc[par ^ 1].work = 0; ctr->lc[par ^ 1].n_slow = 0; ctr->lc[par ^ 1].work_slow = 0; ctr->lc[par ^ 1].ticket = 0;
        if constexpr (LOG) *LG.next_other = 0u;
      }
    }
    if (lane < 4) { T.ncand[lane] = 0; T.bad[lane] = 0; T.code[FK_WIN / 16 + lane] = 0; T.st[FK_WIN / 32 + lane] = 0; T.lt[FK_WIN / 32 + lane] = 0; T.np[FK_WIN / 32 + lane] = 0; T.ncand2 = 0; }
  }
  // (phase 3 reads the candidate list with a clamped index instead of a branch: every entry must be a position)
  for (int i = tid; i < FK_MAXCAND + 2; i += FK_BLOCK) T.cand[i] = (unsigned short) FK_HL;
  lds_barrier ();
  int tile = __builtin_amdgcn_readfirstlane ((int) T.grp[0]);
  int grp_end = tile + FK_GROUP;
  u32 gpar = 0, it = 0;

  // next tile's bytes: HBM -> LDS, 2 KiB per wave (two instructions of 1 KiB), and the word in front of the wave's
  // first byte.  The stream's first and last tiles are done here too: their chunks that stick out are fetched from a
  // valid address instead and phase 1 overwrites what lies outside the stream with read delimiters (edge_chunk).
  u32 pred = 0;
  // (the interior test comes first and stands for "there is such a tile" as well: tiles 1 .. t_hi all exist, so the usual
  // tile pays one compare, not two; its byte offset is non-negative and is added as an unsigned number)
  auto prefetch = [&] (int t) {
    const int g0w = t * FK_OWN - FK_HL + 2048 * wave;   // stream position of the wave's first byte
    if ((u32) (t - 1) < (u32) t_hi) {
      const uint8_t *g = seq + (u32) g0w;
      lds_dma32 (g, voff16, &raw[128 * wave]);
      pred = *reinterpret_cast<const u32 *> (g - 4);
    }
    else if (t < nt_all) {
      issue_chunk (seq, n_bytes, (long) g0w + 16l * lane, &raw[128 * wave]);
      issue_chunk (seq, n_bytes, (long) g0w + 1024l + 16l * lane, &raw[128 * wave + 64]);
      pred = stream_byte (seq, n_bytes, (long) g0w - 1) << 24;
    }
  };
  prefetch (tile);
  u32 n_sbuf = 0;                                       // (workgroup-uniform)
  auto flush_slow = [&] () {
    if (wave == 0 && n_sbuf) {                          // (thread 0 wrote the entries: same wave, LDS keeps its order)
      u32 base = 0;
      if (lane == 0) base = atomicAdd (&ctr->lc[par].n_slow, n_sbuf);
      base = (u32) __builtin_amdgcn_readfirstlane ((int) base);
      if ((u32) lane < n_sbuf) slow_list[base + (u32) lane] = T.sbuf[lane];
    }
    n_sbuf = 0;
  };

  STAMP_DECL;
  while (tile < nt_all) {
    STAMP (0);
    const u32 slot = it & 3u;
    // (opaque: as a loop invariant the compiler keeps "wave == 0" as a 64-bit mask in a spilled pair of SGPRs and pays two
    // v_readlane and an s_and per tile to look at it; compared afresh it is one s_cmp)
    int wave_now = wave;
    asm volatile ("" : "+s"(wave_now));
    if (wave_now == 0) {
      if (tile + FK_GROUP == grp_end) { if (lane == 0) T.grp[gpar ^ 1u] = atomicAdd (&ctr->lc[par].work, (u32) FK_GROUP); }
      if (lane == 0) { T.ncand[(it + 2u) & 3u] = 0; T.ncand2 = 0; }      // (zeroed two tiles ahead; the second chance's: a barrier ahead)
    }
    u32 S32 = 0, L32 = 0, N32 = 0;
    const u32 pred_now = pred;
    u32 x[8];
    // ---- phase 1: 32 bytes per lane -> codes, run starts, letters ---------------------------------------------
    // (with_n: the tile's second chance, see below -- 'N' is let through and its positions are gathered as well)
    auto classify = [&] (auto with_n) -> u32 {
      constexpr bool NP = decltype (with_n)::value;
      // the word in front: the last word of the lane before (one DPP move); the wave's first lane has it from `pred`
      const u32 prevw = (u32) __builtin_amdgcn_update_dpp ((int) pred_now, (int) x[7], 0x138, 0xF, 0xF, false);   // wave_shr:1
      u32 selp = prevw & M07, bad = 0, r[8], sc[4] = {0, 0, 0, 0}, lc[4] = {0, 0, 0, 0}, nc[4] = {0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const u32 w = x[j];
        const u32 sel = w & M07;
        // every byte must be the table's entry for its low three bits: 1 'A', 2 '\n', 3 'C', 4 'T', 7 'G' (entries 0, 5, 6
        // hold bytes with other low bits: they can never match; the second chance has 'N' at 6)
        const u32 rec = __builtin_amdgcn_perm (NP ? 0x474E0054u : 0x47000054u, 0x430A4101u, sel);
        bad = __builtin_amdgcn_bitop3_b32 (bad, w, rec, BITOP3_OR_XOR);
        // 2-bit code (A 0, C 1, G 2, T 3; N 0; the delimiter 3, which no record ever packs) by table look-up on the low three
        // bits (one v_perm; round 2 took bits 1-2 of b ^ (b >> 1): a shift and a v_bitop3, and codes that sat one bit up),
        // gathered 4 bytes -> 8 bits
        const u32 cd = __builtin_amdgcn_perm (0x02000003u, 0x01030000u, sel);
        r[j] = __builtin_amdgcn_udot4 (cd, 0x40100401u, 0u, false);
        // letters have bit 6 ('\n' has not): 64 x the byte of 8 flags per pair of words
        const u32 wts = (j & 1) ? 0x80402010u : 0x08040201u;
        lc[j >> 1] = __builtin_amdgcn_udot4 (w & M40, wts, lc[j >> 1], false);
        // run start: low three bits differ from the byte before
        const u32 d = sel ^ __builtin_amdgcn_alignbit (sel, selp, 24);
        const u32 nz = __builtin_amdgcn_perm (0x01010101u, 0x01010100u, d);       // byte != 0 -> 1
        sc[j >> 1] = __builtin_amdgcn_udot4 (nz, wts, sc[j >> 1], false);
        if constexpr (NP) nc[j >> 1] = __builtin_amdgcn_udot4 (__builtin_amdgcn_perm (0x00010000u, 0u, sel), wts, nc[j >> 1], false);
        selp = sel;
      }
      // four gathered bytes -> one word with two byte permutes and an OR (as shifts and ORs the compiler spent seven
      // instructions per word here, 26 per tile and lane for the three planes; now 13)
      auto pack4 = [] (u32 b0, u32 b1, u32 b2, u32 b3) {
        return __builtin_amdgcn_perm (b1, b0, 0x0c0c0400u) | __builtin_amdgcn_perm (b3, b2, 0x04000c0cu);
      };
      const u32 code_lo = pack4 (r[0], r[1], r[2], r[3]);
      const u32 code_hi = pack4 (r[4], r[5], r[6], r[7]);
      S32 = pack4 (sc[0], sc[1], sc[2], sc[3]);
      {                                                   // (the letter flags come 64-fold: bit 6 of the bytes)
        const u32 u = lc[0] | (lc[1] << 8), v = lc[2] | (lc[3] << 8);
        L32 = (u >> 6) | (v << 10);
      }
      *reinterpret_cast<uint2 *> (&T.code[2 * tid]) = make_uint2 (code_lo, code_hi);
      T.st[tid] = S32;
      T.lt[tid] = L32;
      if constexpr (NP) { N32 = nc[0] | (nc[1] << 8) | (nc[2] << 16) | (nc[3] << 24); T.np[tid] = N32; }
      return bad;
    };
    // ---- phase 2: candidate tract starts among this lane's 32 positions ------------------------------------------
    // (the next lane's run starts come by DPP; the wave's last lane assumes none, i.e. that runs go on: phase 3 checks
    // the length anyway.)  count_addr: the LDS word that counts the tile's candidates.
    auto find_candidates = [&] (u32 count_addr, bool n_tile) {
      const u32 nx = (u32) __builtin_amdgcn_update_dpp (0, (int) S32, 0x130, 0xF, 0xF, false);   // wave_shl:1
      u32 cand = S32 & L32 & own;
      // no run start at the next mprime - 1 positions (the usual minimum lengths without a loop: its scalar bookkeeping
      // costs more than the vector work)
      {
        // (minimum lengths 2, 3, 4 through masks in VGPRs: a branch on mprime here cost a dozen scalar instructions per tile)
        const u32 a1 = __builtin_amdgcn_alignbit (nx, S32, 1u), a2 = __builtin_amdgcn_alignbit (nx, S32, 2u), a3 = __builtin_amdgcn_alignbit (nx, S32, 3u);
        const u32 t = __builtin_amdgcn_bitop3_b32 (a1, a2, vm3, 0xF8);      // a1 | (a2 & m3)
        cand = __builtin_amdgcn_bitop3_b32 (cand, t, a3 & vm4, 0x10);       // cand & ~t & ~(a3 & m4)
        int mp_now = mprime;
        asm volatile ("" : "+s"(mp_now));                 // (compared afresh: one s_cmp instead of a mask kept across the loop)
        if (mp_now > 4) for (int j = 4; j < mp_now; j++) cand &= ~__builtin_amdgcn_alignbit (nx, S32, (u32) j);
      }
      // (a run of 'N' long enough to count takes its context from the tract before it, reference src/hopo_counter.c:246-248:
      // the general kernel's business)
      if (n_tile && (cand & N32)) T.ncand2 = 0x40000000u;
      const u32 n = (u32) __popc (cand);
      // (tried: every lane reserves its own entries with one LDS atomic on the count word instead of the prefix sum -- 7 M
      // fewer vector instructions, but 64 adds on one address hold the LDS pipe for as many cycles: + 17 % time)
      const u32 incl = wave_inclusive_scan (n);
      const u32 total = (u32) __builtin_amdgcn_readlane ((int) incl, 63);
      u32 wraw;
      lds_add_issue (count_addr, total, wraw);
      const u32 p0 = (u32) (FK_UNIT * tid);
      const u32 wbase = lds_collect (wraw);
      // (a wave whose candidates would not all fit leaves its list alone: the tile is given up below)
      if (wbase + total <= (u32) FK_MAXCAND) {
        u32 at = wbase + incl - n;
        // (tried: a loop on "any lane has one left" as a scalar branch, with the idle lanes writing to spare entries -- no
        // exec mask to keep per iteration, 11 M fewer scalar instructions per launch, and 4 % slower)
#ifdef FK_CAND_UNROLL
        // The first FK_CAND_UNROLL candidates of every lane without a loop (a lane has 1.5 on average, 98 in 100 have at most
        // four): the loop's exec bookkeeping and branch are three scalar instructions per turn at the pace of the wave's
        // busiest lane; here a lane without a candidate left writes to a spare entry of its own instead
        const u32 spare_idx = (u32) FK_MAXCAND + 2u + (u32) lane;
#pragma unroll
        for (int j = 0; j < FK_CAND_UNROLL; j++) {
          const u32 b = (u32) __ffs ((int) cand) - 1u;
          T.cand[((u32) j < n) ? at + (u32) j : spare_idx] = (unsigned short) (p0 | b);
          cand &= cand - 1u;
        }
        at += (u32) FK_CAND_UNROLL;
#endif
        while (cand) {
          const u32 b = (u32) __ffs ((int) cand) - 1u;
          cand &= cand - 1u;
          T.cand[at] = (unsigned short) (p0 | b);
          at++;
        }
      }
    };
    {
      // This wave's part of the tile has landed in raw.  (Tried in round 3: a counted wait -- vmcnt(7) when a full partition
      // pass, whose last seven vector-memory operations are its stores, has run since the prefetch -- so that the stores
      // need not be waited for.  It needs the predecessor word loaded outside the compiler's view, into a register that
      // the compiler then copies before the word has arrived; through LDS it would cost what it gains, 1.7 %.)
      asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
      STAMP (1);
      if constexpr (LOG) sink.collect ();
      const uint4 va = raw[2 * tid], vb = raw[2 * tid + 1];
      x[0] = va.x; x[1] = va.y; x[2] = va.z; x[3] = va.w; x[4] = vb.x; x[5] = vb.y; x[6] = vb.z; x[7] = vb.w;
      if (!((u32) (tile - 1) < (u32) t_hi)) {            // (uniform; the stream's first and last tiles) chunks that are not wholly inside the stream: byte by byte
        const long p0 = tile * (long) FK_OWN - FK_HL + 32l * tid;
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const long g = p0 + 16l * h;
          if (!(g >= 0 && g + 16 <= n_bytes)) {
            const EdgeChunk e = edge_chunk (seq, n_bytes, g);      // (outside the stream everything is a read delimiter)
            // (the wait for whatever the call left in flight, here: left to the compiler it sits where the two paths meet,
            // in the classification of EVERY tile, as vmcnt(0) -- which would also wait for the stores of a partition pass)
            __builtin_amdgcn_s_waitcnt (0x0F70);          // vmcnt(0) only
            x[4 * h] = e.x; x[4 * h + 1] = e.y; x[4 * h + 2] = e.z; x[4 * h + 3] = e.w;
          }
        }
      }
      if (classify (std::false_type ())) T.ncand[slot] = 0x40000000u;   // (more candidates than any tile has: looked at below; later atomic adds keep it so)
    }
    STAMP (2);
    // the tile after this one
    int nt = tile + 1;
    if (nt >= grp_end) nt = __builtin_amdgcn_readfirstlane ((int) T.grp[gpar ^ 1u]);
    asm volatile ("s_waitcnt lgkmcnt(0)" ::: "memory");   // raw has been read: it may be refilled
    prefetch (nt);
    STAMP (3);

#if defined(FK_EXP_STOP) && FK_EXP_STOP == 1          // experiment builds only (tools/exp_fast_phases.sh)
    if (S32 == 0x12345u && L32 == 0x54321u) T.bad[3] = 1u;
    S32 = 0;
#endif
    find_candidates (ncand_addr + 4u * slot, false);
    STAMP (4);
    // the records staged so far, exactly: read BEFORE the barrier -- between the barrier that ended the tile before and
    // this one nobody appends, so every wave reads the same number; behind the barrier the quick waves are already
    // appending this tile's records while a slow one has yet to look (seen: one scan in fifty lost some eighty records
    // when a wave got a larger number, partitioned on its own schedule and its barriers paired up with the others' wrongly)
    const u32 staged_v = SL.n;
    if constexpr (LOG) sink.tile_top (staged_v);
#if !(defined(FK_EXP_NOBAR) && (FK_EXP_NOBAR & 2))
    lds_barrier ();
#endif
    STAMP (5);

    // ---- phase 3: one lane per candidate ----------------------------------------------------------------------
#if defined(FK_EXP_STOP) && FK_EXP_STOP <= 2
    const u32 ncand_all = (T.ncand[slot] == 0x7FFFFFFFu) ? 1u : 0u;
#else
    const u32 ncand_all = (u32) __builtin_amdgcn_readfirstlane ((int) T.ncand[slot]);
#endif
    // (`bound` has counted candidates, of which one in eight is not recorded -- with the true count the buffer is
    // partitioned when it is full)
    sink.bound = LOG ? 0u : (u32) __builtin_amdgcn_readfirstlane ((int) staged_v);
#if defined(FK_EXP_LINEAR)
    sink.bound = 0;
#endif
    u32 ncand_now = ncand_all;
    if (ncand_all >= 0x40000000u) {
      // Second chance for a tile with a byte outside the five: if all such bytes are 'N' (the no-call of every sequencer),
      // classify again with 'N' let through (it packs as code 0 like 'A' -- what the reference's table gives it in a flank
      // read forwards), gather where the 'N's are, find the candidates again, and let phase 3 read the tracts that have
      // one within their 2k + length positions from the stream.  Anything else -- lower case, U, IUPAC codes, a countable
      // run of 'N' -- still sends the tile to the general kernel.  (The bytes are still in registers; the tile costs
      // two more barriers and a second classification, a fraction of what the general kernel takes for it.)
      const u32 bad2 = classify (std::true_type ());
      if (bad2) T.ncand2 = 0x40000000u;
      lds_barrier ();
      find_candidates ((u32) (size_t) (lptr_t) &T.ncand2, true);
      lds_barrier ();
      ncand_now = (u32) __builtin_amdgcn_readfirstlane ((int) T.ncand2);
    }
    const bool give_up = ncand_now > (u32) FK_MAXCAND;    // (too many candidates, or a byte that is not for this kernel)
    if (give_up) {
      // (listed in batches: one atomic with a return value per tile, on one address for the whole grid, cost 4.6 us each
      // when every tile of a stream was given up -- more than scanning the tile)
      if (tid == 0) T.sbuf[n_sbuf] = (u32) tile;
      if (++n_sbuf == 32u) flush_slow ();
    }
    else {
      const int ncand = (int) ncand_now;
      // room for all of the tile's candidates (four tiles out of five): one reservation, none per round
      const bool all_fit = sink.bound + (u32) ncand <= (u32) sink.S;
      if (all_fit) sink.bound += (u32) ncand;
      if constexpr (W != 1) {                           // k > 12: two windows per plane (the flanks do not fit one with the tract)
        const long g0 = tile * (long) FK_OWN - FK_HL;
        const bool n_tile = ncand_all >= 0x40000000u;
        for (int cb0 = 0; cb0 < ncand; cb0 += FK_BLOCK) {
          if (!all_fit) sink.reserve1 ((u32) min (ncand - cb0, FK_BLOCK));   // (may partition)
          if (cb0 + 64 * wave >= ncand) continue;          // (whole waves without a candidate in this round)
          const int ci = cb0 + tid;
          bool have = false;
          u64 c0 = 0, c1 = 0;
          u32 base = 0, len10 = 0, flag = 0;
          if constexpr (W == 2) {
            // Straight-line path (round 3) for a tract whose k + length + k positions fit one 64-bit window of the planes
            // -- everything else, rare, goes through fast_tract below -- as in the one-word kernel: lanes past the list's
            // end redo its last entry and are masked at the end; the run starts from s + 1, the letters and the codes from
            // s - k are read together (three words each), the right flank's codes behind them once the run's end is known.
            const bool valid = ci < ncand;
            const u32 s = T.cand[min (ci, ncand - 1)];
            const u32 q = s + 1u, u = s - (u32) k;
            const u32 *ps = &T.st[q >> 5], *pl = &T.lt[u >> 5], *pc = &T.code[u >> 4];
            u32 s0 = ps[0], s1 = ps[1], s2 = ps[2], l0 = pl[0], l1 = pl[1], l2 = pl[2], w0 = pc[0], w1 = pc[1], w2 = pc[2];
            asm volatile ("" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(l0), "+v"(l1), "+v"(l2), "+v"(w0), "+v"(w1), "+v"(w2));
            const u64 ns = ((u64) __builtin_amdgcn_alignbit (s2, s1, q) << 32) | __builtin_amdgcn_alignbit (s1, s0, q);
            const u32 len = (u32) __builtin_ctzll (ns | (1ull << 63)) + 1u;     // (no run start in 64 positions: 64, which does not fit)
            const int room = 64 - (int) (len + 2u * (u32) k);
            const bool fits = room >= 0;
            const u32 rpos = s + (fits ? len : 0u);          // (first position behind the tract; kept inside the planes otherwise)
            const u32 *pr = &T.code[rpos >> 4];
            const u32 r0 = pr[0], r1 = pr[1], r2 = pr[2];
            const u64 notl = ~(((u64) __builtin_amdgcn_alignbit (l2, l1, u) << 32) | __builtin_amdgcn_alignbit (l1, l0, u));
            const u64 km = kmask (k);
            const u64 L64 = ((u64) __builtin_amdgcn_alignbit (w2, w1, 2u * u) << 32) | __builtin_amdgcn_alignbit (w1, w0, 2u * u);
            const u64 R64 = ((u64) __builtin_amdgcn_alignbit (r2, r1, 2u * rpos) << 32) | __builtin_amdgcn_alignbit (r1, r0, 2u * rpos);
            const u64 left = L64 & km, right = R64 & km;
            const u32 cb = (u32) (L64 >> (2 * k)) & 3u;      // the tract's base (2k <= 56: inside the window)
            u64 nl2 = 0, nr2 = 0;
            if (n_tile) {                                   // (uniform) 'N's in the flanks: forward code 3 where the strand is turned (see fast_tract)
              const u32 kb32 = (1u << k) - 1u;
              const u32 nl = bits32 (T.np, (int) u) & kb32, nr = bits32 (T.np, (int) rpos) & kb32;
              if (nl | nr) { nl2 = spread_pairs (nl); nr2 = spread_pairs (nr); }
            }
            const bool rev = cb >= 2u;
            const u64 rc0 = revcomp_k (right | nr2, k), rc1 = revcomp_k (left | nl2, k);
            c0 = rev ? rc0 : left; c1 = rev ? rc1 : right;
            base = rev ? 3u - cb : cb; flag = rev ? 2u : 1u; len10 = len;
            have = valid && fits && (notl << (room & 63)) == 0ull && (int) len >= mprime;
            if (valid && !fits) have = fast_tract<false> (T, seq, n_bytes, g0, (int) s, k, mprime, c0, c1, base, len10, flag, n_tile);
          }
          else if (ci < ncand) have = fast_tract<false> (T, seq, n_bytes, g0, (int) T.cand[ci], k, mprime, c0, c1, base, len10, flag, n_tile);
          const u64 okm = __builtin_amdgcn_ballot_w64 (have);
          u32 araw;
          lds_add_issue (sink.count_addr (), (u32) __builtin_popcountll (okm), araw);
          const u32 at = lds_collect (araw) + (u32) __builtin_amdgcn_mbcnt_hi ((u32) (okm >> 32), __builtin_amdgcn_mbcnt_lo ((u32) okm, 0u));
          if (have) sink.store_fields (at, c0, c1, base, len10, flag);
        }
      }
      else
      for (int cb0 = 0; cb0 < ncand; cb0 += FK_BLOCK) {
        // Straight-line path for a tract whose k + length + k positions fit one 32-bit window (everything else is
        // `general`), without branches: lanes past the list's end work on its last entry again and are masked out at the
        // end.  The windows depend on s alone, so all their LDS reads are in flight together: run starts from s + 1,
        // letters from s - k, and 64 bits of codes from s - k (left flank, tract, right flank).
#if !(defined(FK_EXP_STOP) && FK_EXP_STOP == 3)
        if (!all_fit) sink.reserve1 ((u32) min (ncand - cb0, FK_BLOCK));    // (may partition: before anything of this round is in registers)
#endif
        if (cb0 + 64 * wave >= ncand) continue;            // (whole waves without a candidate in this round)
        const int ci = cb0 + tid;
        const bool valid = ci < ncand;
        const u32 s = T.cand[min (ci, ncand - 1)];
        const u32 q = s + 1u, u = s - vk;
        // (tried: the word addresses as shift + v_lshl_add_u32 on the planes' LDS addresses, two instructions per plane
        // instead of the compiler's three -- the three base addresses in VGPRs were three registers too many at this
        // kernel's 128-register edge once the counted wait below needed one)
        const u32 *ps = &T.st[q >> 5], *pl = &T.lt[u >> 5], *pc = &T.code[u >> 4];
        u32 s0 = ps[0], s1 = ps[1], l0 = pl[0], l1 = pl[1], w0 = pc[0], w1 = pc[1], w2 = pc[2];
        // (pinned: left alone the compiler waits for the run starts before it asks for the codes, and reads the letters
        // only inside a branch it makes up -- three LDS round trips in a row instead of one)
        asm volatile ("" : "+v"(s0), "+v"(s1), "+v"(l0), "+v"(l1), "+v"(w0), "+v"(w1), "+v"(w2));
        // run end: first run start after s (bit 31 set: "none in 32 positions" reads as a run of 32, which is general)
        const u32 len = (u32) __builtin_ctz (__builtin_amdgcn_alignbit (s1, s0, q) | 0x80000000u) + 1u;
        const u32 need = len + vk2;                        // k letters, the tract, k letters
        const u32 room = v32 - need;                       // (negative: the window is too short -> `general`)
        const u32 notl = ~__builtin_amdgcn_alignbit (l1, l0, u);
        // recorded: fits the window, letters all the way, long enough -- as one integer so that it costs vector
        // instructions only (the compiler turns `a && b && c` into an exec region of six scalar instructions)
        // (lanes past the list's end: a minimum length that nothing reaches)
        const u32 bad = __builtin_amdgcn_bitop3_b32 (notl << (room & 31u), room >> 31, (len - (valid ? vmp : 0x1000u)) >> 31, 0xFE);
        const bool ok = bad == 0u;
        if (valid && (int) room < 0) {                     // (rare) the long tracts: appended on their own (before this wave's atomic is issued: its result register must not live across a call)
          const u64 rec = fast_general_tract (T, seq, n_bytes, tile * (long) FK_OWN - FK_HL, (int) s, k, mprime, (ncand_all >= 0x40000000u));
#if !(defined(FK_EXP_STOP) && FK_EXP_STOP == 3)
          if (rec != 0ull) sink.store1 (atomicAdd (&sink.L.n, 1u), (u32) rec, (u32) (rec >> 32));
#endif
        }
#if defined(FK_EXP_STOP) && FK_EXP_STOP == 3
        const u32 at = 0;
#else
        // whether the tract is recorded is known: its slot in the staging buffer is asked for now (an LDS atomic per
        // wave) and the record is worked out while that is on its way
        const u64 okm = __builtin_amdgcn_ballot_w64 (ok);
        u32 araw;
#if defined(FK_EXP_SCATTER)
        lds_add_issue (sink.count_addr (), 0u, araw);
#else
        lds_add_issue (sink.count_addr (), (u32) __builtin_popcountll (okm), araw);
#endif
#endif
        const u32 sh = u + u;
        const u32 clo = __builtin_amdgcn_alignbit (w1, w0, sh), chi = __builtin_amdgcn_alignbit (w2, w1, sh);
        const u32 l = clo & vkm;
        const u32 cb = (clo >> vk2) & 3u;                  // the tract's base
        const u32 r = (u32) ((((u64) chi << 32) | clo) >> (vk2 + len + len)) & vkm;
        // canonical orientation (reference: src/hopo_counter.c:233-246): T / G tracts store the reverse complement
        u32 rcl = revcomp_v32 (l, M55, vrs), rcr = revcomp_v32 (r, M55, vrs);
        if ((ncand_all >= 0x40000000u)) {                                       // (uniform) 'N's in the flanks: forward code 3 where the strand is turned (see fast_tract)
          const u32 nb = bits32 (T.np, (int) u);
          if ((nb << ((v32 - need) & 31u)) != 0u && need <= 32u) {
            rcl = revcomp_v32 (l | (u32) spread_pairs (nb & ((1u << k) - 1u)), M55, vrs);
            rcr = revcomp_v32 (r | (u32) spread_pairs ((nb >> ((u32) k + len)) & ((1u << k) - 1u)), M55, vrs);
          }
        }
        const bool rev = cb >= 2u;
        const u32 c0 = rev ? rcr : l, c1 = rev ? rcl : r;
        // base << 2 | flag << 3 by table look-up on cb: A (0, fwd) 8, C (1, fwd) 12, G (-> C, rev) 20, T (-> A, rev) 16
        // (looked up straight into the record's top byte: selector byte 3 = cb, the others "zero")
        const u32 fld24 = __builtin_amdgcn_perm (0u, 0x10140C08u, (cb << 24) | 0x000C0C0Cu);
        const u32 lo = c1 | (len << 24);
        const u32 hi = c0 | fld24;
        STAMP (6);
#if defined(FK_EXP_STOP) && FK_EXP_STOP == 3
        if (ok) asm volatile ("" :: "v"(lo), "v"(hi), "v"(at));
#else
#if defined(FK_EXP_LINEAR)                              // experiment: no staging, no partition: records to a linear log of the workgroup's own
        {
          const u32 at = lds_collect (araw) + (u32) __builtin_amdgcn_mbcnt_hi ((u32) (okm >> 32), __builtin_amdgcn_mbcnt_lo ((u32) okm, 0u));
          typedef __attribute__((address_space(1))) u64 *gwords_t;
          gwords_t q = (gwords_t) ((u64) (size_t) BK.pool + (((u64) blockIdx.x * 200000u + (at % 200000u)) << 3));
          if (ok) *q = ((u64) hi << 32) | lo;
        }
#elif defined(FK_EXP_SCATTER)                           // experiment: no staging, every record straight to a private place of its bucket (wrong results)
        (void) lds_collect (araw);
        {
          u32 h = __builtin_amdgcn_udot4 (lo, vh0, 0u, false);
          h = __builtin_amdgcn_udot4 (hi & vm27, vh1, h, false);
          const u32 bin = ok ? ((h ^ (h >> 8)) & 255u) : 256u + (u32) lane;
          const u32 r = atomicAdd (&SL.hist[bin], 1u);
          const uint4 e = reinterpret_cast<const uint4 *> (SL.gbase)[bin & 255u];
          asm volatile ("" :: "v"(e.x), "v"(e.y), "v"(e.z), "v"(e.w));
          typedef __attribute__((address_space(1))) u64 *gwords_t;
          gwords_t q = (gwords_t) ((u64) (size_t) BK.pool + ((((u64) (bin & 255u) * 512u + (blockIdx.x & 511u)) * 512u + (r & 511u)) << 3));
          if (ok) *q = ((u64) hi << 32) | lo;
        }
#else
        const u32 at = lds_collect (araw) + (u32) __builtin_amdgcn_mbcnt_hi ((u32) (okm >> 32), __builtin_amdgcn_mbcnt_lo ((u32) okm, 0u));
        if constexpr (LOG) sink.store1ok (ok, at, lo, hi, (u32) lane);
        else sink.store1v (ok ? at : (u32) sink.S + (u32) lane, lo, hi, vh0, vh1, vm27);   // (no record: a spare slot takes the write)
#endif
#endif
        STAMP (7);
      }
    }
#if !(defined(FK_EXP_NOBAR) && (FK_EXP_NOBAR & 1))     // (timing experiment only: results are wrong without it)
    lds_barrier ();
#endif
    STAMP (8);
    if (nt >= grp_end) { gpar ^= 1u; grp_end = nt + FK_GROUP; }
    tile = nt;
    it++;
  }
  STAMP_FLUSH;
  flush_slow ();
  sink.finish ();
}

// The record log of scan_fast_kernel<W, true> -> the hash buckets.  A workgroup takes log blocks in turn (static stride)
// and sorts one block per pass by bucket, a counting sort like StageSink's with the records held in registers from HBM
// to their sorted slot -- the same reservation protocol on the same bucket cursors and chunk table as every other
// producer of raw records.
#define PL_BLOCK 512
#ifndef PL_WG_PER_CU
#define PL_WG_PER_CU 2
#endif
template <int W>
__global__ __launch_bounds__ (PL_BLOCK, PL_BLOCK * PL_WG_PER_CU / 256)
void partition_log_kernel (LogSpace LG, Buckets BK, DevCounters *ctr, int k)
{
  // One log block (TJ_LOGB words: 8192 one-word or 4096 two-word records, 16 words per thread) per pass, the records in
  // registers from the load to their place in the sorted buffer: count per bucket (LDS atomics) -- prefix, one
  // reservation per bucket, where the runs go (the owners, as in StageSink::partition_pass) -- every record straight to
  // its sorted slot in LDS (its rank: an LDS atomic on the bucket's running position) -- the sorted buffer out in runs,
  // coalesced.  Four barriers per block; two workgroups per CU cover each other's waits.  What bounds it is the memory
  // system: for the 10 M-read sample's one-word records 0.48 GB read in a line, 0.48 GB written in runs of 32 records to
  // 256 places (tools/ubench/write_runs.hip: that write pattern alone takes 0.16 ms); loading the next block's records
  // while this one is sorted and written changed nothing (tried, with 32 more registers).
  static_assert (W == 1 || W == 2, "the log holds one-word and two-word records");
  typedef StageLds<W, 2> Lds;
  __shared__ Lds L;
  constexpr u32 RB = TJ_LOGB / W;                        // records per block
  static_assert (Lds::S == (int) RB && Lds::WS == W, "one pass per log block");
  constexpr u32 RR = RB / PL_BLOCK;                      // records per thread
  static_assert (RR % 4 == 0, "bins are kept four to a register");
  const u32 tid = threadIdx.x, lane = tid & 63u;
  const u32 wave = (u32) __builtin_amdgcn_readfirstlane ((int) (tid >> 6));
  if (tid < TJ_P) L.hist[tid] = 0;
  u32 cur_j = TJ_EMPTY, cur_chunk = TJ_NOCHUNK;         // owner thread (tid < TJ_P): the chunk its bucket was written to last
  lds_barrier ();
  const u32 n_blocks = min (*LG.next, LG.n_blocks);
#if TJ_STAMPS == 2
  Stamper stamper; stamper.begin ();
#endif
  // (blocks in turn, static: a shared work counter would be one more address that every workgroup of the grid adds to)
  u32 b = blockIdx.x;
  if (b >= n_blocks) return;
  u64 w[RR][W];
  u32 n = min (LG.count[b], RB), nn = 0;
  // a block's records, the part of it that was written: whole waves skip what lies behind the block's count (every scanning
  // workgroup leaves one block partly filled and one empty: a tenth of all blocks on the 10 M-read sample, nearly all of
  // them on a sparse one)
  auto load_block = [&] (u32 blk, u32 cnt) {
    const u64 *__restrict__ src = LG.log + ((u64) blk << TJ_LOGB_SHIFT);
    const u32 w0 = tid & ~63u;                            // (the wave's first thread: uniform)
#pragma unroll
    for (u32 r = 0; r < RR; r++) {
#pragma unroll
      for (int j = 0; j < W; j++) w[r][j] = 0;
      if (w0 + r * PL_BLOCK < cnt) {
        if constexpr (W == 1) w[r][0] = (src + r * PL_BLOCK)[tid];
        else { const ulonglong2 v = reinterpret_cast<const ulonglong2 *> (src + 2u * r * PL_BLOCK)[tid]; w[r][0] = v.x; w[r][1] = v.y; }
      }
    }
  };
  auto bucket_of = [&] (const u64 *rec) -> u32 {
    if constexpr (W == 1) return bucket_of_rec1 ((u32) rec[0], (u32) (rec[0] >> 32));
    else { u64 c0, c1; u32 base, len10, flag; unpack_raw<W> (rec, k, c0, c1, base, len10, flag); return bucket_of_key (c0, c1, base, len10); }
  };
  load_block (b, n);
  while (true) {
    const u32 bn = b + gridDim.x;
    const bool more = bn < n_blocks;                      // (uniform)
    if (more) nn = min (LG.count[bn], RB);                // (asked for now, looked at when this block is done)
    PLSTAMP (0);
    u32 tq = tid;
    asm volatile ("" : "+v"(tq));                         // (opaque: what the compiler can derive from tid alone it hoists out of the loop -- sixteen 64-bit offsets 8 i took 32 registers and went to scratch)
    u32 pk[RR / 4];
#pragma unroll
    for (u32 r = 0; r < RR / 4; r++) pk[r] = 0;
    if (n) {
      // ---- records per bucket
#pragma unroll
      for (u32 r = 0; r < RR; r++) {
        const u32 bin = bucket_of (w[r]);
        pk[r >> 2] |= bin << (8u * (r & 3u));
        if (tid + r * PL_BLOCK < n) atomicAdd (&L.hist[bin], 1u);
      }
      PLSTAMP (1);
      lds_barrier ();
      PLSTAMP (2);
      // ---- owners: prefix, reservation, where the run lives
      if (wave < TJ_P / 64) {
        const uint4 h4 = *reinterpret_cast<const uint4 *> (&L.hist[4 * lane]);
        const u32 tot = h4.x + h4.y + h4.z + h4.w;
        const u32 e0 = wave_inclusive_scan (tot) - tot;
        if ((lane >> 4) == wave) *reinterpret_cast<uint4 *> (&L.offs[4 * lane]) = make_uint4 (e0, e0 + h4.x, e0 + h4.x + h4.y, e0 + h4.x + h4.y + h4.z);
        const u32 cnt = L.hist[tid], off = L.offs[tid];
        u64 a1 = 0, a2 = 0;
        u32 thr = off + cnt;
        if (cnt) {
          const u32 p0 = atomicAdd (&BK.cursors[tid * TJ_CSTRIDE], cnt);
          const u32 ch = (u32) TJ_CH0 << BK.ch_shift;
          bucket_claim_ahead_range (BK, tid, p0, cnt, ctr);   // (a run of up to a block's records may hold more than one chunk's first record)
          const u32 j0 = chunk_of_pos (BK, p0), j1 = chunk_of_pos (BK, p0 + cnt - 1);
          if (j0 != cur_j) { cur_j = j0; cur_chunk = bucket_chunk_id (BK, tid, j0, true, ctr); }
          if (cur_chunk != TJ_NOCHUNK) a1 = (u64) (size_t) (BK.pool + ((u64) cur_chunk * ch + (p0 - j0 * ch)) * W) - (8ull * W) * off;
          if (j1 != j0) {                                 // the run crosses into the next chunk (chunks are longer than a log block: at most once)
            thr = off + ((j0 + 1u) * ch - p0);
            cur_j = j0 + 1u; cur_chunk = bucket_chunk_id (BK, tid, j0 + 1u, true, ctr);
            if (cur_chunk != TJ_NOCHUNK) a2 = (u64) (size_t) (BK.pool + ((u64) cur_chunk * ch) * W) - (8ull * W) * thr;
          }
        }
        reinterpret_cast<uint4 *> (L.gbase)[tid] = make_uint4 ((u32) a1, (u32) (a1 >> 32) | (thr << 16), (u32) a2, (u32) (a2 >> 32));
      }
      PLSTAMP (3);
      lds_barrier ();
      PLSTAMP (4);
      if (tid < TJ_P) L.hist[tid] = 0;                    // (for the next block; the ranks below count in offs)
      // ---- every record to its sorted slot (offs[bucket] runs from the bucket's first slot to its last)
#pragma unroll
      for (u32 r = 0; r < RR; r++)
        if (tid + r * PL_BLOCK < n) {
          const u32 bin = (pk[r >> 2] >> (8u * (r & 3u))) & 255u;
          const u32 d = atomicAdd (&L.offs[bin], 1u);
#pragma unroll
          for (int j = 0; j < W; j++) L.rec[d * W + j] = w[r][j];
          L.bin[d] = (unsigned char) bin;
        }
      PLSTAMP (5);
      lds_barrier ();
      PLSTAMP (6);
      // ---- copy-out: sorted slot i -> its place in the bucket's run
      for (u32 r0 = 0; r0 < RR; r0 += 4) {
        if (r0 * PL_BLOCK >= n) break;
        u32 cb[4]; uint4 e[4]; u64 cw[4][W];
#pragma unroll
        for (u32 h = 0; h < 4; h++) cb[h] = L.bin[tq + (r0 + h) * PL_BLOCK];
#pragma unroll
        for (u32 h = 0; h < 4; h++) {
          e[h] = reinterpret_cast<const uint4 *> (L.gbase)[cb[h]];
#pragma unroll
          for (int j = 0; j < W; j++) cw[h][j] = L.rec[(tq + (r0 + h) * PL_BLOCK) * W + j];
        }
#pragma unroll
        for (u32 h = 0; h < 4; h++) {
          const u32 i = tq + (r0 + h) * PL_BLOCK;
          const u64 a = (i < (e[h].y >> 16)) ? (((u64) (e[h].y & 0xFFFFu) << 32) | e[h].x) : (((u64) e[h].w << 32) | e[h].z);
          if (i < n && a != 0ull) {
            typedef __attribute__((address_space(1))) u64 *gwords_t;
            gwords_t q = (gwords_t) (a + (u64) (8u * W * i));
#pragma unroll
            for (int j = 0; j < W; j++) q[j] = cw[h][j];
          }
        }
      }
      PLSTAMP (7);
      lds_barrier ();                                     // the sorted buffer is free again
      PLSTAMP (8);
    }
    if (!more) break;
    load_block (bn, nn);
    n = nn; b = bn;
  }
  PLSTAMP_FLUSH;
}

template <int W>
__global__ void nrun_fixup_bins_kernel (const uint8_t *__restrict__ seq, long n_bytes, int k, int mprime,
                                        Buckets BK, DevCounters *ctr, const FixEntry *fix, u32 fix_cap, int par)
{
  nrun_fixup_entries<W> (seq, n_bytes, k, mprime, BK, ctr, fix, fix_cap, par, blockIdx.x * (u64) blockDim.x + threadIdx.x, (u64) gridDim.x * blockDim.x);
}

// host hopo_element array (40 B each) -> buckets
template <int W>
__global__ void bin_elems_kernel (const u64 *__restrict__ elems5, long n, int k, Buckets BK, DevCounters *ctr)
{
  for (long i = blockIdx.x * (long) blockDim.x + threadIdx.x; i < n; i += (long) gridDim.x * blockDim.x) {
    const u64 m = elems5[5 * i + 2];
    bucket_insert_slow<W> (elems5[5 * i], elems5[5 * i + 1], (u32) (m & 3ull), (u32) ((m >> TJ_META_LEN_SHIFT) & 0x3FFull),
                           (u32) ((m >> TJ_META_FLAG_SHIFT) & 3ull), k, BK, ctr);
  }
}

// buckets -> flat list of 24-byte raw records, padding skipped (test aid; order is irrelevant)
template <int W>
__global__ void unpack_buckets_kernel (Buckets BK, int k, u64 *__restrict__ out, u64 cap, u64 *n_out)
{
  const u32 b = blockIdx.x;
  const u32 n = BK.cursors[b * TJ_CSTRIDE];
  for (u32 i = threadIdx.x; i < n; i += blockDim.x) {
    u64 c0, c1; u32 base, len10, flag;
    const u64 at = bucket_slot (BK, b, i, false, nullptr);
    if (at == ~0ull) continue;
    unpack_raw<W> (BK.pool + at * W, k, c0, c1, base, len10, flag);
    if (flag == 3u) continue;
    const u64 o = atomicAdd (n_out, 1ull);
    if (o < cap) { u64 *q = out + 3 * o; q[0] = c0; q[1] = c1; q[2] = make_meta (base, len10, flag); }
  }
}

// chunk 0 of every bucket
__global__ void init_table_kernel (u32 *table, u32 maxj, u32 *pool_next)
{
  const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < TJ_P) table[(u64) b * maxj] = b;
  if (b == 0) *pool_next = TJ_P;
}

// empty buckets: cursors 0, every bucket owns chunk b as its chunk 0, the rest of the table unclaimed, pool_next = TJ_P.
// One workgroup per bucket; only the row entries the bucket can have claimed are rewritten (its records / chunk + the
// one claimed ahead), unless the whole row is asked for.
__global__ void clear_buckets_kernel (u32 *cursors, DevCounters *ctr, u32 *table, u32 maxj, int ch_shift, FinCounts *fin, u32 *bins, int nbins,
                                      DevCounters *snap_ctr, u32 *snap_cursors, FinPlan *plan, u32 *__restrict__ fine = nullptr, int fine_bits = 0)
{
  const u32 b = blockIdx.x;
  if (snap_ctr) {                                       // (every workgroup reads its own cursor, workgroup 0 the rest: before anything is zeroed)
    if (threadIdx.x == 0) {
      snap_cursors[b] = cursors[b * TJ_CSTRIDE];
      if (b == 0) {
        snap_cursors[TJ_P] = cursors[TJ_P * TJ_CSTRIDE];
        // The kept count and the overflow flag as a kernel boundary shows them.  plan_tail read them inside the
        // aggregation, from its last workgroup, through relaxed atomics and no fence (a fence there doubled the kernel's
        // time): the host compares the two readings and runs the ordering step again if they ever differ.
        if (plan) { plan->n1_exact = (long) fin->n_kept; plan->ovf_exact = fin->overflow; }
      }
    }
    if (b == 0) for (u32 i = threadIdx.x; i < sizeof (DevCounters) / 4; i += blockDim.x) reinterpret_cast<u32 *> (snap_ctr)[i] = reinterpret_cast<const u32 *> (ctr)[i];
    __syncthreads ();
  }
  // the scan counters as well (all but n_undefined, which counts from reset to reset): a reset right after a finalise
  // then needs no memset of its own
  if (b == 0) for (u32 i = threadIdx.x; i < sizeof (DevCounters) / 4; i += blockDim.x) if (i != offsetof (DevCounters, n_undefined) / 4 && i != offsetof (DevCounters, n_undefined) / 4 + 1) reinterpret_cast<u32 *> (ctr)[i] = 0;
  if (table) {
    u32 used = maxj;
    if (ch_shift >= 0) used = min (maxj, ((cursors[b * TJ_CSTRIDE] / TJ_CH0) >> ch_shift) + 3u);
    for (u32 j = threadIdx.x; j < used; j += blockDim.x) table[(u64) b * maxj + j] = j ? TJ_EMPTY : b;
  }
  if (fine) {
    // The aggregation before this kernel has counted the records it kept into 2^fine_bits bins (the finest partition the
    // ordering step can ask for); its plan says how many leading bits the step uses: a bin of the step is a run of
    // 2^shift fine ones (a bin is a prefix of the key).  Summed here, one run per thread, which also zeroes what it has
    // read for the next aggregation (loads first: a store to the array between two loads of it serialises them) -- this
    // replaces the step's counting pass over the kept records.  No usable plan: the counts go, the step counts again.
    const int gid = (int) (b * blockDim.x + threadIdx.x), gsz = (int) (gridDim.x * blockDim.x);
    if (plan->ok) {
      const int shift = fine_bits - plan->nbits, run = 1 << shift, nb = plan->nbins;
      for (int i = gid; i < nb; i += gsz) {
        u32 sum = 0;
        if (shift >= 2) {
          uint4 *f4 = reinterpret_cast<uint4 *> (fine) + (size_t) i * (run >> 2);
          for (int j = 0; j < (run >> 2); j++) { const uint4 v = f4[j]; sum += v.x + v.y + v.z + v.w; }
          for (int j = 0; j < (run >> 2); j++) f4[j] = make_uint4 (0, 0, 0, 0);
        }
        else {
          for (int j = 0; j < run; j++) sum += fine[(i << shift) + j];
          for (int j = 0; j < run; j++) fine[(i << shift) + j] = 0;
        }
        bins[i] = sum;
      }
    }
    else for (int i = gid; i < (1 << fine_bits); i += gsz) fine[i] = 0;
  }
  else if (bins) for (int i = (int) (b * blockDim.x + threadIdx.x); i < nbins; i += (int) (gridDim.x * blockDim.x)) bins[i] = 0;
  __syncthreads ();
  if (threadIdx.x == 0) { cursors[b * TJ_CSTRIDE] = 0; if (b == 0) { cursors[TJ_P * TJ_CSTRIDE] = TJ_P; fin->n_kept = 0; fin->overflow = 0; fin->pad = 0; } }
}

// chunk table with a longer row
__global__ void table_relayout_kernel (const u32 *__restrict__ old, u32 old_maxj, u32 *__restrict__ neu, u32 new_maxj)
{
  const u32 b = blockIdx.x;
  for (u32 j = threadIdx.x; j < old_maxj; j += blockDim.x) neu[(u64) b * new_maxj + j] = old[(u64) b * old_maxj + j];
}

// ---------------------------------------------------------------------------------------------------------------
// aggregation: one workgroup per bucket streams its raw records through an LDS hash table keyed by
// (base, context, stored length), counting the two strands separately (reference: the qsort + run-length pass of
// src/hopo_counter.c:351-365), then applies the strand / singleton filter (:367-374) and appends the survivors, as
// 24-byte records carrying count and canon_flag, to the kept list.  A table that fills up closes (no new keys; records
// of absent keys go to a second pool) and the leftovers are aggregated in further rounds; a key is always entirely in
// one round, so rounds never split a count.  One kernel per record width; aggregate1_kernel carries the commentary.

#define AG_BLOCK    1024
#define AG_NCH      512                 // chunk ids of a bucket cached in LDS (longer buckets look the rest up in the table)

// ---- W = 1 (k <= 12): the whole reduction key is the record word without its strand flag, so one 64-bit LDS
// compare-and-swap per probe decides "new key / same key / other key" -- no tag, no publish step, no key read-back.
#define AG1_S        8192
#define AG1_CLOSE_AT 4864                // typical overshoot: a few keys; worst case one per lane (1024): 72 % full
#define AG1_R        4                   // records in flight per lane
#define AG1_MARK     (~0ull)             // a key has bits 59-63 clear

#define BS_MAXBITS   16                  // (bins of the ordering step: see bin_count_kernel)
__host__ __device__ static inline int cov_table_bits (long n1) { int b = 10; while ((1l << b) < 4 * n1 && b < 31) b++; return b; }   // 2 n1 entries at most: half full
// The coverage table's keys are flanks cut to 31 bits (src/hopo_counter.c:419-438): with 2k <= 31 there are 4^k of them, and
// when that is no more than the hash table would have slots the table is addressed by the key itself -- a quarter of the
// memory to zero and to look through at k = 10, and a plain add instead of a compare-and-swap round trip per record.
__host__ __device__ static inline int cov_plan_bits (long n1, int k, int *direct)
{
  const int hb = cov_table_bits (n1), kb = 2 * k < 31 ? 2 * k : 31;
  *direct = kb <= hb ? 1 : 0;
  return kb <= hb ? kb : hb;
}
// bins of the finest partition the ordering step can ask for: the aggregation counts its kept records into them as it
// writes them (the sample's kept count, which decides how many leading bits the step really uses, is not known yet)
__host__ __device__ static inline int fine_bin_bits (int k) { return BS_MAXBITS < 1 + 4 * k ? BS_MAXBITS : 1 + 4 * k; }
__device__ __forceinline__ u32 bin_of_record (u64 c0, u64 c1, u64 meta, int k, int nbits);
__host__ __device__ static inline int bin_bits_for (long n1, int k)
{
  int nbits = 6;
  while (nbits < BS_MAXBITS && nbits < 1 + 4 * k && (48l << nbits) < n1) nbits++;    // ~24..48 records per bin: most of a wavefront's lanes busy
  return nbits < 1 + 4 * k ? nbits : 1 + 4 * k;
}
// The last workgroup of an aggregation to finish derives the sizes the ordering kernels work with (FinPlan) from the
// number of kept records: what a one-thread kernel launched behind the aggregation used to do, for 5 us of stream time.
// fin->pad counts the workgroups that are through (zero before every aggregation: here, and clear_buckets_kernel).
// the kept list is full: said with an atomic whose return the thread waits for (see plan_tail)
__device__ __forceinline__ void flag_kept_overflow (FinCounts *fin)
{
  u32 old = atomicOr (&fin->overflow, 1u);
  asm volatile ("" :: "v"(old));
}

__device__ __forceinline__ void plan_tail (FinCounts *fin, int k, long cap, FinPlan *plan)
{
  if (!plan) return;
  // (no __threadfence: at device scope it writes the XCD's L2 back, which doubled the aggregation's run time.  What the
  // last workgroup needs from the others went through atomics that their threads waited for -- the kept count, the
  // overflow flag -- before the barrier in front of their tick.)
  __syncthreads ();
  if (threadIdx.x == 0) {
    const u32 done = atomicAdd (&fin->pad, 1u);
    if (done == gridDim.x - 1u) {
      const long n1 = (long) __hip_atomic_load (&fin->n_kept, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const u32 ovf = __hip_atomic_load (&fin->overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      plan->n1 = n1; plan->kept_overflow = ovf;
      plan->nbits = bin_bits_for (n1, k); plan->nbins = 1 << plan->nbits;
      int direct;
      plan->log2t = cov_plan_bits (n1, k, &direct); plan->t = 1l << plan->log2t; plan->cov_direct = (u32) direct;
      plan->ok = (n1 > 0 && n1 <= cap && !ovf) ? 1 : 0;
      fin->pad = 0;
    }
  }
}

// The t-th batch a workgroup hands to its waves: the bucket is read FROM ITS END.  The order of the records does not matter to
// a hash aggregation, and what a bucket received last is what the 256 MB memory-side cache may still hold when the
// aggregation starts right behind the kernel that filled it: behind partition_log_kernel the waves' wait for their loads
// fell by more than half and the kernel from 0.160 to 0.137 ms (the time it takes behind the fused scan kernel).
// Returns a position >= n when the bucket is used up.
#ifndef AG_FROM_END
#define AG_FROM_END 1
#endif
__device__ __forceinline__ u32 batch_start (u32 t, u32 n, u32 batch)
{
#if AG_FROM_END
  const u32 nb = (n + batch - 1u) / batch;
  return t < nb ? (nb - 1u - t) * batch : n;
#else
  return t * batch;
#endif
}

struct Agg1Lds
{
  u64 key[AG1_S];                       // record without its strand flag (never 0: the base next to an A tract is not A)
  u32 cnt[2 * AG1_S];                   // per slot: records seen on the forward / on the reverse strand
  u32 chunk[AG_NCH];                    // the bucket's chunk ids
  u32 n_claimed, n_ovf, total;
  u32 next_batch;                       // batches of 64 * AG1_R records are handed to the waves as they come for one
  u32 wsum[AG_BLOCK / 64];
};

__global__ __launch_bounds__ (AG_BLOCK)
void aggregate1_kernel (Buckets BK, u64 *ovf, int k, int remove_biased, u64 *__restrict__ kept, u64 kept_cap, FinCounts *fin,
                        long plan_cap, FinPlan *plan, u32 *__restrict__ fine, int fine_bits)
{
  __shared__ Agg1Lds L;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const u32 bkt = blockIdx.x;
  u32 n = BK.cursors[bkt * TJ_CSTRIDE];
  if (n) for (u32 j = (u32) tid; j <= chunk_of_pos (BK, n - 1u) && j < AG_NCH; j += AG_BLOCK) L.chunk[j] = bucket_chunk_id (BK, bkt, j, false, nullptr);
  auto chunk_id = [&] (u32 j) { return j < AG_NCH ? L.chunk[j] : bucket_chunk_id (BK, bkt, j, false, nullptr); };

  // records that do not fit a round's table go to the same place in a second pool (`ovf`) and are read from there in
  // the next round, whose own leftovers go back to the first pool: no wave ever writes where another may still read,
  // so the waves of a round run free of each other (no barrier in the record loop).
  const u64 *src = BK.pool;
  u64 *dst = ovf;
  ASTAMP_DECL;
  while (n > 0) {
    ASTAMP (0);
    for (int i = tid; i < AG1_S; i += AG_BLOCK) { L.key[i] = 0; L.cnt[2 * i] = 0; L.cnt[2 * i + 1] = 0; }
    if (tid == 0) { L.n_claimed = 0; L.n_ovf = 0; L.next_batch = 0; }
    __syncthreads ();

    // AG1_R records per lane are in flight while the previous AG1_R are inserted: with one 8-byte load per lane the
    // kernel sat at latency x (8 KB per CU in flight) = 1.4 TB/s whatever the table did.
    u64 wn[AG1_R];
    u32 vn = 0;
    // A batch is 64 * AG1_R consecutive records of the bucket: less than a chunk, so it lies in at most two chunks and
    // everything about them is wave-uniform (chunk ids come from LDS: no global look-up whose wait would also wait
    // for the records in flight).
    auto fetch = [&] (u32 b0) {
      vn = 0;
#pragma unroll
      for (int r = 0; r < AG1_R; r++) wn[r] = 0;
      if (b0 >= n) return;
      const u32 ch = (u32) TJ_CH0 << BK.ch_shift;
      const u32 j0 = chunk_of_pos (BK, b0), bound = (j0 + 1u) * ch;
      if (b0 + 64u * AG1_R <= min (n, bound) && j0 < AG_NCH) {      // a whole batch inside one chunk (nearly all of them):
        const u32 c0s = (u32) __builtin_amdgcn_readfirstlane ((int) L.chunk[j0]);   // scalar address arithmetic, four plain loads
        if (c0s != TJ_NOCHUNK) {
          const u64 *q = src + ((u64) c0s * ch + (b0 - j0 * ch)) + (u32) lane;
#pragma unroll
          for (int r = 0; r < AG1_R; r++) wn[r] = q[64 * r];
          vn = (1u << AG1_R) - 1u;
          return;
        }
      }
      const u32 c0 = chunk_id (j0), c1 = (bound < n) ? chunk_id (j0 + 1u) : TJ_NOCHUNK;
      const u64 off0 = (u64) c0 * ch - (u64) j0 * ch, off1 = (u64) c1 * ch - (u64) bound;   // record index -> pool index
#pragma unroll
      for (int r = 0; r < AG1_R; r++) {
        const u32 idx = b0 + (u32) r * 64u + (u32) lane;
        const bool hi = idx >= bound;
        if (idx < n && (hi ? c1 : c0) != TJ_NOCHUNK) { vn |= 1u << r; wn[r] = src[(hi ? off1 : off0) + idx]; }
      }
    };
    // Waves take batches of 64 * AG1_R consecutive records as they become free (one LDS atomic per batch): with a fixed
    // share per wave the slower waves of a SIMD finished a third of the round after the faster ones had started to wait.
    auto next_batch = [&] () {
      u32 b = 0;
      if (lane == 0) b = atomicAdd (&L.next_batch, 1u);
      return batch_start ((u32) __builtin_amdgcn_readfirstlane ((int) b), n, 64u * AG1_R);
    };
    u32 b_next = next_batch ();
    fetch (b_next);
    ASTAMP (1);
    while (b_next < n) {
      const u32 b0 = b_next;
      (void) b0;
      // this round's records have arrived (fetched one round ago).  The registers pass through the asm so that the
      // compiler stops tracking them as pending loads: otherwise it waits for them again at their first use -- after
      // the next round's loads have been issued, i.e. for those as well (vmcnt counts in order).
      asm volatile ("s_waitcnt vmcnt(0)" : "+v"(wn[0]), "+v"(wn[1]), "+v"(wn[2]), "+v"(wn[3]) :: "memory");
      static_assert (AG1_R == 4, "asm operand list");
      ASTAMP (2);
      u64 w[AG1_R];
#pragma unroll
      for (int r = 0; r < AG1_R; r++) w[r] = wn[r];
      const u32 valid = vn;
      b_next = next_batch ();
      fetch (b_next);
      ASTAMP (3);
      ASTAMP (4);
#if defined(TJ_EXP_AGG) && TJ_EXP_AGG == 1      // experiment: loads only
      { u64 acc = 0;
#pragma unroll
        for (int r = 0; r < AG1_R; r++) acc ^= w[r];
        if (acc == 0x123456789ull) atomicAdd (&L.n_ovf, 1u);
        continue; }
#endif
      // Table protocol.  Slots come in aligned pairs read with one 16-byte LDS load; a key's probe chain is its home
      // pair, the next pair, ... and it lives in the first slot of the chain that was free when it arrived.  Almost
      // every record finds its key with that one load and adds to its counter (no compare-and-swap: a bucket holds
      // each key hundreds of times).  A new key is CASed into the first free slot of the chain while the table is
      // open; once it is closed (enough keys; a lane looks before each claim, so at most one more key per lane gets
      // in) that slot is MARKed instead, and a MARK ends the chain for everybody: either the key or a MARK wins the
      // slot, so a key is in the table for all of its records or for none (those go back to the bucket's front).
      // Lanes work through their unsettled records independently (a wave loops until all its lanes are through): the
      // iterations of a wave follow the lane with the most probes in total, not the worst probe of every record.
      {
        auto home = [] (u64 key) {
          u32 h = (u32) key ^ __builtin_amdgcn_alignbit ((u32) (key >> 32), (u32) (key >> 32), 17);
          h *= 0x9E3779B1u; h ^= h >> 15;
          return h & (AG1_S / 2 - 1);
        };
        // first, every record looks at its home pair (straight-line code: the four loads are in flight together) --
        // that settles all but a few per cent of them
        // (the loads are for every lane -- a slot without a record holds 0, whose home pair is as good as any -- and their
        // results pass through an asm statement: left alone the compiler reads the pair's first slot, waits, looks, and
        // reads the second slot inside a branch, record after record: eight LDS round trips in a row instead of one)
        u32 todo = 0;
        u32 hp[AG1_R];
        ulonglong2 hk[AG1_R];
#pragma unroll
        for (int r = 0; r < AG1_R; r++) {
          hp[r] = home (w[r] & R1_KEY_MASK);
          hk[r] = *reinterpret_cast<const ulonglong2 *> (&L.key[2 * hp[r]]);
        }
        asm volatile ("" : "+v"(hk[0].x), "+v"(hk[0].y), "+v"(hk[1].x), "+v"(hk[1].y), "+v"(hk[2].x), "+v"(hk[2].y), "+v"(hk[3].x), "+v"(hk[3].y));
#pragma unroll
        for (int r = 0; r < AG1_R; r++) {
          const u64 cur = w[r];
          if (((valid >> r) & 1u) && ((cur >> R1_FLAG_SHIFT) & 3ull) != 3ull) {
            const u64 key = cur & R1_KEY_MASK;
            const bool hx = hk[r].x == key, hy = hk[r].y == key;
            if (hx | hy) atomicAdd (&L.cnt[4 * hp[r] + (hx ? 0u : 2u) + ((u32) (cur >> (R1_FLAG_SHIFT + 1)) & 1u)], 1u);
            else todo |= 1u << r;
          }
        }
        // the rest (new keys, collisions, a closed table) one probe at a time, every lane through its own records
        u32 probes = 0;
        u32 r = todo ? (u32) __ffs ((int) todo) - 1u : 0u;
        u64 cur = (r == 0u) ? w[0] : (r == 1u) ? w[1] : (r == 2u) ? w[2] : w[3];
        u32 pair = home (cur & R1_KEY_MASK);
        while (todo) {
          bool adv = false, left = false;
          {
            const u64 key = cur & R1_KEY_MASK;
            const u32 strand = (u32) (cur >> (R1_FLAG_SHIFT + 1)) & 1u;
            const ulonglong2 kk = *reinterpret_cast<const ulonglong2 *> (&L.key[2 * pair]);
            if (kk.x == key || kk.y == key) { atomicAdd (&L.cnt[4 * pair + (kk.x == key ? 0u : 2u) + strand], 1u); adv = true; }
            else {
              const bool e0 = (kk.x == 0ull) | (kk.x == AG1_MARK), e1 = (kk.y == 0ull) | (kk.y == AG1_MARK);
              if (e0 | e1) {                            // the chain ends in this pair
                const u32 slot = 2 * pair + (e0 ? 0u : 1u);
                if ((e0 ? kk.x : kk.y) == AG1_MARK) { left = true; adv = true; }
                else {
                  const bool closed = __hip_atomic_load (&L.n_claimed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > AG1_CLOSE_AT;
                  const u64 old = atomicCAS ((unsigned long long *) &L.key[slot], 0ull, closed ? AG1_MARK : (unsigned long long) key);
                  if (old == 0ull) {
                    if (closed) left = true;
                    else { atomicAdd (&L.n_claimed, 1u); atomicAdd (&L.cnt[2 * slot + strand], 1u); }
                    adv = true;
                  }                                     // else somebody else took the slot: look at the pair again
                }
              }
              else if (++probes >= AG1_S / 2) { left = true; adv = true; }   // every slot holds another key
              else pair = (pair + 1u) & (AG1_S / 2 - 1);
            }
          }
          if (left) {                                   // to the second pool (chunk ids from LDS)
            const u32 o = atomicAdd (&L.n_ovf, 1u);
            const u32 j = chunk_of_pos (BK, o), oc = chunk_id (j);
            if (oc != TJ_NOCHUNK) dst[(((u64) oc * TJ_CH0) << BK.ch_shift) + (o - ((j * TJ_CH0) << BK.ch_shift))] = cur;
          }
          if (adv) {
            todo &= todo - 1u; probes = 0;
            if (todo) {
              r = (u32) __ffs ((int) todo) - 1u;
              cur = (r == 1u) ? w[1] : (r == 2u) ? w[2] : w[3];
              pair = home (cur & R1_KEY_MASK);
            }
          }
        }
      }
      ASTAMP (5);
    }
    __syncthreads ();
    ASTAMP (6);

    u32 mine = 0;
    u64 metas[AG1_S / AG_BLOCK];
#pragma unroll
    for (int r = 0; r < AG1_S / AG_BLOCK; r++) {
      const int slot = tid + r * AG_BLOCK;
      metas[r] = 0;
      if (L.key[slot] && L.key[slot] != AG1_MARK) {
        const u32 cf = L.cnt[2 * slot], cr = L.cnt[2 * slot + 1];
        const u64 flag = (cf ? 1ull : 0ull) | (cr ? 2ull : 0ull);
        const u64 cnt = ((u64) cf + (u64) cr) & 0xFFFFFull;
        const int scnt = (cnt & 0x80000ull) ? (int) cnt - 0x100000 : (int) cnt;
        if (remove_biased ? (flag == 3ull) : (scnt > 1)) {
          const u64 key = L.key[slot];                  // the record's fields (see pack_rec1)
          const u64 len10 = ((key >> 24) & 0xFFull) | (((key >> 56) & 3ull) << 8);
          metas[r] = ((key >> 58) & 1ull) | (len10 << TJ_META_LEN_SHIFT) | (cnt << TJ_META_COUNT_SHIFT) |
                     (0xffeull << TJ_META_MISM_SHIFT) | (flag << TJ_META_FLAG_SHIFT);
          mine++;
        }
      }
    }
    const u32 x = wave_inclusive_scan (mine);
    if (lane == 63) L.wsum[wave] = x;
    __syncthreads ();
    u32 wbase = 0, total = 0;
    for (int wv = 0; wv < AG_BLOCK / 64; wv++) { const u32 sm = L.wsum[wv]; if (wv < wave) wbase += sm; total += sm; }
    if (tid == 0) L.total = total ? atomicAdd (&fin->n_kept, total) : 0u;
    __syncthreads ();
    u64 at = (u64) L.total + wbase + x - mine;
#pragma unroll
    for (int r = 0; r < AG1_S / AG_BLOCK; r++)
      if (metas[r]) {
        const u64 key = L.key[tid + r * AG_BLOCK];
        if (at < kept_cap) {
          u64 *q = kept + 3 * at; q[0] = (key >> 32) & 0xFFFFFFull; q[1] = key & 0xFFFFFFull; q[2] = metas[r];
          if (fine) atomicAdd (&fine[bin_of_record ((key >> 32) & 0xFFFFFFull, key & 0xFFFFFFull, metas[r], k, fine_bits)], 1u);
        }
        else flag_kept_overflow (fin);
        at++;
      }
    __threadfence_block ();
    __syncthreads ();
    n = L.n_ovf;
    { const u64 *t = src; src = dst; dst = (u64 *) t; }
    __syncthreads ();
    ASTAMP (7);
  }
  ASTAMP_FLUSH;
  plan_tail (fin, k, plan_cap, plan);
}

// ---- W = 2 (k <= 28): word 0 of the record is never 0 and leaves bit 63 free, so it doubles as the slot's claim word:
// compare-and-swap 0 -> (word0 | PENDING), write word 1, then store word0.  A prober that meets its own word0 with
// PENDING set looks again; with it clear the second word is there to compare.  Same structure as aggregate1_kernel:
// records in flight per lane, read-first probing (a bucket holds each key many times), slots in aligned pairs read
// together (a probe chain is home pair, next pair, ...; a key lives in the first slot of the chain that was free when it
// arrived), a closed table MARKs the free slot at the end of a probe chain so that a key is in the table for all of its
// records or for none, leftovers go to the second pool, no barrier in the record loop.  6144 slots (144 KB of LDS with
// the counters): a bucket of the long-read configuration (4.6 k keys) fits one round, and at the usual 2-3 k keys the
// home pair settles nine records in ten.
#define AG2_S        6144
#define AG2_PAIRS    (AG2_S / 2)
#ifndef AG2_CLOSE_AT
#define AG2_CLOSE_AT 2560
#endif
                                        // (closing early pays: at 4.6 k keys per bucket -- the long-read configuration -- a table filled to 75 % costs more in probes than the second round costs in traffic)
#define AG2_R        4                   // records in flight per lane
#define AG2_VALID    (1ull << 63)        // set in the stored second key word (bits 61-63 of it are not key)
#define AG2_PENDING  (1ull << 63)
#define AG2_MARK     (1ull << 63)        // PENDING without a word0 (word0 always has bit 0 set)

struct Agg2Lds
{
  ulonglong2 kk[AG2_S];                 // x: claim word (record word 0), y: second key word | VALID
  u32 cnt[2 * AG2_S];                   // per slot: forward / reverse strand
  u32 chunk[AG_NCH];
  u32 n_claimed, n_ovf, total;
  u32 next_batch;                       // (see aggregate1_kernel: batches are handed to the waves as they come for one)
  u32 wsum[AG_BLOCK / 64];
};

__global__ __launch_bounds__ (AG_BLOCK)
void aggregate2_kernel (Buckets BK, u64 *ovf, int k, int remove_biased, u64 *__restrict__ kept, u64 kept_cap, FinCounts *fin,
                        long plan_cap, FinPlan *plan, u32 *__restrict__ fine, int fine_bits)
{
  __shared__ Agg2Lds L;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const u32 bkt = blockIdx.x;
  u32 n = BK.cursors[bkt * TJ_CSTRIDE];
  const u64 m56 = (1ull << 56) - 1ull, fmask = ~(3ull << 61);
  if (n) for (u32 j = (u32) tid; j <= chunk_of_pos (BK, n - 1u) && j < AG_NCH; j += AG_BLOCK) L.chunk[j] = bucket_chunk_id (BK, bkt, j, false, nullptr);
  auto chunk_id = [&] (u32 j) { return j < AG_NCH ? L.chunk[j] : bucket_chunk_id (BK, bkt, j, false, nullptr); };
  const u64 *src = BK.pool;
  u64 *dst = ovf;

  while (n > 0) {
    for (int i = tid; i < AG2_S; i += AG_BLOCK) { L.kk[i] = make_ulonglong2 (0ull, 0ull); L.cnt[2 * i] = 0; L.cnt[2 * i + 1] = 0; }
    if (tid == 0) { L.n_claimed = 0; L.n_ovf = 0; L.next_batch = 0; }
    __syncthreads ();

    u64 wn[2 * AG2_R];
    u32 vn = 0;
    auto fetch = [&] (u32 b0) {                         // see aggregate1_kernel: a round lies in at most two chunks
      vn = 0;
#pragma unroll
      for (int r = 0; r < 2 * AG2_R; r++) wn[r] = 0;
      if (b0 >= n) return;
      const u32 ch = (u32) TJ_CH0 << BK.ch_shift;
      const u32 j0 = chunk_of_pos (BK, b0), bound = (j0 + 1u) * ch;
      const u32 c0 = chunk_id (j0), c1 = (bound < n) ? chunk_id (j0 + 1u) : TJ_NOCHUNK;
      const u64 off0 = (u64) c0 * ch - (u64) j0 * ch, off1 = (u64) c1 * ch - (u64) bound;
#pragma unroll
      for (int r = 0; r < AG2_R; r++) {
        const u32 idx = b0 + (u32) r * 64u + (u32) lane;
        const bool hi = idx >= bound;
        if (idx < n && (hi ? c1 : c0) != TJ_NOCHUNK) {
          const ulonglong2 v = *reinterpret_cast<const ulonglong2 *> (src + 2 * ((hi ? off1 : off0) + idx));
          vn |= 1u << r; wn[2 * r] = v.x; wn[2 * r + 1] = v.y;
        }
      }
    };
    auto next_batch = [&] () {
      u32 b = 0;
      if (lane == 0) b = atomicAdd (&L.next_batch, 1u);
      return batch_start ((u32) __builtin_amdgcn_readfirstlane ((int) b), n, 64u * AG2_R);
    };
    u32 b_next = next_batch ();
    fetch (b_next);
    while (b_next < n) {
      asm volatile ("s_waitcnt vmcnt(0)" : "+v"(wn[0]), "+v"(wn[1]), "+v"(wn[2]), "+v"(wn[3]), "+v"(wn[4]), "+v"(wn[5]), "+v"(wn[6]), "+v"(wn[7]) :: "memory");
      static_assert (AG2_R == 4, "asm operand list");
      u64 w[2 * AG2_R];
#pragma unroll
      for (int r = 0; r < 2 * AG2_R; r++) w[r] = wn[r];
      const u32 valid = vn;
      b_next = next_batch ();
      fetch (b_next);
      {
        auto home = [] (u64 a, u64 b) {                 // home pair of a key
          const u64 key1 = b & ~(3ull << 61);
          u32 h = (u32) a ^ __builtin_amdgcn_alignbit ((u32) (a >> 32), (u32) (a >> 32), 19) ^
                  __builtin_amdgcn_alignbit ((u32) key1, (u32) key1, 11) ^ __builtin_amdgcn_alignbit ((u32) (key1 >> 32), (u32) (key1 >> 32), 25);
          h *= 0x9E3779B1u; h ^= h >> 15;
          return __umulhi (h, (u32) AG2_PAIRS);
        };
        // the two slots of a pair, loaded together (atomic loads keep the LDS address space)
        auto load_pair = [&] (u32 pair, ulonglong2 &s0, ulonglong2 &s1) {
          s0.x = __hip_atomic_load ((unsigned long long *) &L.kk[2 * pair].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          s0.y = __hip_atomic_load ((unsigned long long *) &L.kk[2 * pair].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          s1.x = __hip_atomic_load ((unsigned long long *) &L.kk[2 * pair + 1].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          s1.y = __hip_atomic_load ((unsigned long long *) &L.kk[2 * pair + 1].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        // First, every record walks the first pairs of its chain in straight-line code (the loads of all records of a
        // lane are in flight together): a published claim word with the record's own valid second word settles it, a full
        // pair of other keys sends it on to the next pair; anything else (a free slot: the key is new; a slot in the
        // middle of being published) is left to the general loop below, which starts at the pair reached here.
        u32 todo = 0, look = 0, cur[AG2_R];
#pragma unroll
        for (int q = 0; q < AG2_R; q++) {
          cur[q] = 0;
          if (((valid >> q) & 1u) && ((w[2 * q + 1] >> 61) & 3ull) != 3ull) { look |= 1u << q; cur[q] = home (w[2 * q], w[2 * q + 1]); }
        }
#pragma unroll
        for (int pass = 0; pass < 3; pass++) {
          if (pass && !look) break;
#pragma unroll
          for (int q = 0; q < AG2_R; q++)
            if ((look >> q) & 1u) {
              const u64 a0 = w[2 * q], a1 = w[2 * q + 1];
              ulonglong2 s0, s1;
              load_pair (cur[q], s0, s1);
              const u64 want = (a1 & fmask) | AG2_VALID;
              const bool m0 = (s0.x == a0) & (s0.y == want), m1 = (s1.x == a0) & (s1.y == want);
              if (m0 | m1) { atomicAdd (&L.cnt[2 * (2 * cur[q] + (m0 ? 0u : 1u)) + ((u32) (a1 >> 62) & 1u)], 1u); look &= ~(1u << q); }
              else {
                const u64 mine = a0 | AG2_PENDING;
                // (a slot holding another key: claim word published, not ours -- or ours with a valid second word that differs)
                const bool o0 = (s0.x != 0ull) & (s0.x != AG2_MARK) & (s0.x != mine) & ((s0.x & AG2_PENDING) == 0ull) & ((s0.x != a0) | ((s0.y & AG2_VALID) != 0ull));
                const bool o1 = (s1.x != 0ull) & (s1.x != AG2_MARK) & (s1.x != mine) & ((s1.x & AG2_PENDING) == 0ull) & ((s1.x != a0) | ((s1.y & AG2_VALID) != 0ull));
                if (o0 & o1) cur[q] = (cur[q] + 1u == (u32) AG2_PAIRS) ? 0u : cur[q] + 1u;
                else { look &= ~(1u << q); todo |= 1u << q; }
              }
            }
        }
        todo |= look;
        u32 probes = 0;
        u32 r = todo ? (u32) __ffs ((int) todo) - 1u : 0u;
        u64 w0 = (r == 0u) ? w[0] : (r == 1u) ? w[2] : (r == 2u) ? w[4] : w[6];
        u64 w1 = (r == 0u) ? w[1] : (r == 1u) ? w[3] : (r == 2u) ? w[5] : w[7];
        u32 pair = (r == 0u) ? cur[0] : (r == 1u) ? cur[1] : (r == 2u) ? cur[2] : cur[3];
        while (todo) {
          bool adv = false, left = false;
          {
            const u64 key1 = w1 & fmask, want = key1 | AG2_VALID;
            const u32 strand = (u32) (w1 >> 62) & 1u;
            ulonglong2 s0, s1;
            load_pair (pair, s0, s1);
            const bool m0 = (s0.x == w0) & (s0.y == want), m1 = (s1.x == w0) & (s1.y == want);
            // our own claim word still PENDING, or published next to a second word that is not valid yet (the loads
            // straddled the owner's writes): look at the pair again
            const bool again = (s0.x == (w0 | AG2_PENDING)) | (s1.x == (w0 | AG2_PENDING)) |
                               ((s0.x == w0) & !(s0.y & AG2_VALID)) | ((s1.x == w0) & !(s1.y & AG2_VALID));
            if (m0 | m1) { atomicAdd (&L.cnt[2 * (2 * pair + (m0 ? 0u : 1u)) + strand], 1u); adv = true; }
            else if (!again) {
              const bool e0 = (s0.x == 0ull) | (s0.x == AG2_MARK), e1 = (s1.x == 0ull) | (s1.x == AG2_MARK);
              if (e0 | e1) {                            // the chain ends in this pair
                const u32 slot = 2 * pair + (e0 ? 0u : 1u);
                if ((e0 ? s0.x : s1.x) == AG2_MARK) { left = true; adv = true; }
                else {                                  // claim the slot, or MARK it if the table is closed
                  const bool closed = __hip_atomic_load (&L.n_claimed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > AG2_CLOSE_AT;
                  const u64 old = atomicCAS ((unsigned long long *) &L.kk[slot].x, 0ull, closed ? AG2_MARK : (unsigned long long) (w0 | AG2_PENDING));
                  if (old == 0ull) {
                    if (closed) left = true;
                    else {                              // claimed: publish the second word, then the first
                      __hip_atomic_store ((unsigned long long *) &L.kk[slot].y, (unsigned long long) want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                      atomicAdd (&L.n_claimed, 1u);
                      __hip_atomic_store ((unsigned long long *) &L.kk[slot].x, (unsigned long long) w0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                      atomicAdd (&L.cnt[2 * slot + strand], 1u);
                    }
                    adv = true;
                  }                                     // else somebody else took the slot: look at the pair again
                }
              }
              else if (++probes >= AG2_PAIRS) { left = true; adv = true; }   // every slot holds another key
              else pair = (pair + 1u == (u32) AG2_PAIRS) ? 0u : pair + 1u;
            }
          }
          if (left) {
            const u32 o = atomicAdd (&L.n_ovf, 1u);
            const u32 j = chunk_of_pos (BK, o), oc = chunk_id (j);
            if (oc != TJ_NOCHUNK) {
              const u64 at = (((u64) oc * TJ_CH0) << BK.ch_shift) + (o - ((j * TJ_CH0) << BK.ch_shift));
              dst[2 * at] = w0; dst[2 * at + 1] = w1;
            }
          }
          if (adv) {
            todo &= todo - 1u; probes = 0;
            if (todo) {
              r = (u32) __ffs ((int) todo) - 1u;
              w0 = (r == 1u) ? w[2] : (r == 2u) ? w[4] : w[6];
              w1 = (r == 1u) ? w[3] : (r == 2u) ? w[5] : w[7];
              pair = (r == 1u) ? cur[1] : (r == 2u) ? cur[2] : cur[3];
            }
          }
        }
      }
    }
    __syncthreads ();

    u32 mine = 0;
    u64 metas[AG2_S / AG_BLOCK];
#pragma unroll
    for (int r = 0; r < AG2_S / AG_BLOCK; r++) {
      const int slot = tid + r * AG_BLOCK;
      metas[r] = 0;
      if (L.kk[slot].x && L.kk[slot].x != AG2_MARK) {
        const u32 cf = L.cnt[2 * slot], cr = L.cnt[2 * slot + 1];
        const u64 flag = (cf ? 1ull : 0ull) | (cr ? 2ull : 0ull);
        const u64 cnt = ((u64) cf + (u64) cr) & 0xFFFFFull;
        const int scnt = (cnt & 0x80000ull) ? (int) cnt - 0x100000 : (int) cnt;
        if (remove_biased ? (flag == 3ull) : (scnt > 1)) {
          const u64 a = L.kk[slot].x, b = L.kk[slot].y;
          const u64 len10 = ((a >> 2) & 31ull) | (((b >> 56) & 31ull) << 5);
          metas[r] = ((a >> 1) & 1ull) | (len10 << TJ_META_LEN_SHIFT) | (cnt << TJ_META_COUNT_SHIFT) |
                     (0xffeull << TJ_META_MISM_SHIFT) | (flag << TJ_META_FLAG_SHIFT);
          mine++;
        }
      }
    }
    const u32 x = wave_inclusive_scan (mine);
    if (lane == 63) L.wsum[wave] = x;
    __syncthreads ();
    u32 wbase = 0, total = 0;
    for (int wv = 0; wv < AG_BLOCK / 64; wv++) { const u32 sm = L.wsum[wv]; if (wv < wave) wbase += sm; total += sm; }
    if (tid == 0) L.total = total ? atomicAdd (&fin->n_kept, total) : 0u;
    __syncthreads ();
    u64 at = (u64) L.total + wbase + x - mine;
#pragma unroll
    for (int r = 0; r < AG2_S / AG_BLOCK; r++)
      if (metas[r]) {
        const int slot = tid + r * AG_BLOCK;
        if (at < kept_cap) {
          u64 *q = kept + 3 * at; q[0] = (L.kk[slot].x >> 7) & m56; q[1] = L.kk[slot].y & m56; q[2] = metas[r];
          if (fine) atomicAdd (&fine[bin_of_record ((L.kk[slot].x >> 7) & m56, L.kk[slot].y & m56, metas[r], k, fine_bits)], 1u);
        }
        else flag_kept_overflow (fin);
        at++;
      }
    __threadfence_block ();
    __syncthreads ();
    n = L.n_ovf;
    { const u64 *t = src; src = dst; dst = (u64 *) t; }
    __syncthreads ();
  }
  plan_tail (fin, k, plan_cap, plan);
}

// ---- W = 4 (k <= 32): the k-mers fill their 64-bit words, so the claim word is made of the small key part and a hash:
// 1 | (base | length << 2) << 1 | 50 hash bits << 13 (never 0, bit 63 free).  Same protocol as aggregate2_kernel; a match
// needs the claim word AND both k-mers, and because the k-mer words have no room for a VALID bit, a claim-word match
// whose k-mers differ -- or are both 0, the cleared state -- is confirmed with a second look before it counts.
#define AG4_S        4096
#define AG4_CLOSE_AT 2560
#define AG4_R        2
#define AG4_PENDING  (1ull << 63)
#define AG4_MARK     (1ull << 63)

struct Agg4Lds
{
  u64 cw[AG4_S], c0[AG4_S], c1[AG4_S];
  u32 cnt[2 * AG4_S];
  u32 chunk[AG_NCH];
  u32 n_claimed, n_ovf, total;
  u32 next_batch;                       // (see aggregate1_kernel)
  u32 wsum[AG_BLOCK / 64];
};

__global__ __launch_bounds__ (AG_BLOCK)
void aggregate4_kernel (Buckets BK, u64 *ovf, int k, int remove_biased, u64 *__restrict__ kept, u64 kept_cap, FinCounts *fin,
                        long plan_cap, FinPlan *plan, u32 *__restrict__ fine, int fine_bits)
{
  __shared__ Agg4Lds L;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const u32 bkt = blockIdx.x;
  u32 n = BK.cursors[bkt * TJ_CSTRIDE];
  if (n) for (u32 j = (u32) tid; j <= chunk_of_pos (BK, n - 1u) && j < AG_NCH; j += AG_BLOCK) L.chunk[j] = bucket_chunk_id (BK, bkt, j, false, nullptr);
  auto chunk_id = [&] (u32 j) { return j < AG_NCH ? L.chunk[j] : bucket_chunk_id (BK, bkt, j, false, nullptr); };
  const u64 *src = BK.pool;
  u64 *dst = ovf;
  auto ld = [] (u64 *p) { return (u64) __hip_atomic_load ((unsigned long long *) p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };

  while (n > 0) {
    for (int i = tid; i < AG4_S; i += AG_BLOCK) { L.cw[i] = 0; L.c0[i] = 0; L.c1[i] = 0; L.cnt[2 * i] = 0; L.cnt[2 * i + 1] = 0; }
    if (tid == 0) { L.n_claimed = 0; L.n_ovf = 0; L.next_batch = 0; }
    __syncthreads ();

    u64 wn[3 * AG4_R];                                  // (the fourth word of a record is padding)
    u32 vn = 0;
    auto fetch = [&] (u32 b0) {                         // see aggregate1_kernel: a round lies in at most two chunks
      vn = 0;
#pragma unroll
      for (int r = 0; r < 3 * AG4_R; r++) wn[r] = 0;
      if (b0 >= n) return;
      const u32 ch = (u32) TJ_CH0 << BK.ch_shift;
      const u32 j0 = chunk_of_pos (BK, b0), bound = (j0 + 1u) * ch;
      const u32 c0 = chunk_id (j0), c1 = (bound < n) ? chunk_id (j0 + 1u) : TJ_NOCHUNK;
      const u64 off0 = (u64) c0 * ch - (u64) j0 * ch, off1 = (u64) c1 * ch - (u64) bound;
#pragma unroll
      for (int r = 0; r < AG4_R; r++) {
        const u32 idx = b0 + (u32) r * 64u + (u32) lane;
        const bool hi = idx >= bound;
        if (idx < n && (hi ? c1 : c0) != TJ_NOCHUNK) {
          const u64 *p = src + 4 * ((hi ? off1 : off0) + idx);
          const ulonglong2 v = *reinterpret_cast<const ulonglong2 *> (p);
          vn |= 1u << r; wn[3 * r] = v.x; wn[3 * r + 1] = v.y; wn[3 * r + 2] = p[2];
        }
      }
    };
    auto next_batch = [&] () {
      u32 b = 0;
      if (lane == 0) b = atomicAdd (&L.next_batch, 1u);
      return batch_start ((u32) __builtin_amdgcn_readfirstlane ((int) b), n, 64u * AG4_R);
    };
    u32 b_next = next_batch ();
    fetch (b_next);
    while (b_next < n) {
      asm volatile ("s_waitcnt vmcnt(0)" : "+v"(wn[0]), "+v"(wn[1]), "+v"(wn[2]), "+v"(wn[3]), "+v"(wn[4]), "+v"(wn[5]) :: "memory");
      static_assert (AG4_R == 2, "asm operand list");
      u64 w[3 * AG4_R];
#pragma unroll
      for (int r = 0; r < 3 * AG4_R; r++) w[r] = wn[r];
      const u32 valid = vn;
      b_next = next_batch ();
      fetch (b_next);
      u32 todo = 0;
#pragma unroll
      for (int q = 0; q < AG4_R; q++) if (((valid >> q) & 1u) && ((w[3 * q + 2] >> 12) & 3ull) != 3ull) todo |= 1u << q;
      u32 probes = 0;
      while (todo) {
        const u32 r = (u32) __ffs ((int) todo) - 1u;
        const u64 c0 = r ? w[3] : w[0], c1 = r ? w[4] : w[1], m2 = r ? w[5] : w[2];
        const u32 k2 = (u32) m2 & 0xFFFu, strand = (u32) (m2 >> 13) & 1u;     // flag 1 = as read, 2 = reverse-complemented
        const u64 hk = hash_key (c0, c1, k2 & 3u, k2 >> 2);
        const u64 cw = 1ull | ((u64) k2 << 1) | ((hk >> 14) << 13);
        u32 slot = ((u32) hk + probes) & (AG4_S - 1);     // (linear probing: the probe count is the offset)
        bool adv = false, left = false;
        const u64 a = ld (&L.cw[slot]);
        u64 x0 = ld (&L.c0[slot]), x1 = ld (&L.c1[slot]);
        if (a == cw) {
          bool eq = (x0 == c0) & (x1 == c1);
          if (!eq || (c0 | c1) == 0ull) { x0 = ld (&L.c0[slot]); x1 = ld (&L.c1[slot]); eq = (x0 == c0) & (x1 == c1); }   // second look, after the claim word was seen published
          if (eq) { atomicAdd (&L.cnt[2 * slot + strand], 1u); adv = true; }
          else if (++probes >= AG4_S) { left = true; adv = true; }
        }
        else if (a == AG4_MARK) { left = true; adv = true; }
        else if (a == 0ull) {                           // the chain ends here: claim the slot, or MARK it if the table is closed
          const bool closed = __hip_atomic_load (&L.n_claimed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > AG4_CLOSE_AT;
          const u64 old = atomicCAS ((unsigned long long *) &L.cw[slot], 0ull, closed ? AG4_MARK : (unsigned long long) (cw | AG4_PENDING));
          if (old == 0ull) {
            if (closed) left = true;
            else {                                      // claimed: publish the k-mers, then the claim word
              __hip_atomic_store ((unsigned long long *) &L.c0[slot], (unsigned long long) c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              __hip_atomic_store ((unsigned long long *) &L.c1[slot], (unsigned long long) c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              atomicAdd (&L.n_claimed, 1u);
              __hip_atomic_store ((unsigned long long *) &L.cw[slot], (unsigned long long) cw, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
              atomicAdd (&L.cnt[2 * slot + strand], 1u);
            }
            adv = true;
          }                                             // else somebody else took the slot: look at it again
        }
        else if (a != (cw | AG4_PENDING)) {             // another key (PENDING with our claim word: its owner is still writing, look again)
          if (++probes >= AG4_S) { left = true; adv = true; }
        }
        if (left) {
          const u32 o = atomicAdd (&L.n_ovf, 1u);
          const u32 j = chunk_of_pos (BK, o), oc = chunk_id (j);
          if (oc != TJ_NOCHUNK) {
            u64 *q = dst + 4 * ((((u64) oc * TJ_CH0) << BK.ch_shift) + (o - ((j * TJ_CH0) << BK.ch_shift)));
            q[0] = c0; q[1] = c1; q[2] = m2; q[3] = 0;
          }
        }
        if (adv) { todo &= todo - 1u; probes = 0; }
      }
    }
    __syncthreads ();

    u32 mine = 0;
    u64 metas[AG4_S / AG_BLOCK];
#pragma unroll
    for (int r = 0; r < AG4_S / AG_BLOCK; r++) {
      const int slot = tid + r * AG_BLOCK;
      metas[r] = 0;
      const u64 cw = L.cw[slot];
      if (cw && cw != AG4_MARK) {
        const u32 cf = L.cnt[2 * slot], cr = L.cnt[2 * slot + 1];
        const u64 flag = (cf ? 1ull : 0ull) | (cr ? 2ull : 0ull);
        const u64 cnt = ((u64) cf + (u64) cr) & 0xFFFFFull;           // 20-bit store wraps (reference :361)
        const int scnt = (cnt & 0x80000ull) ? (int) cnt - 0x100000 : (int) cnt;
        if (remove_biased ? (flag == 3ull) : (scnt > 1)) {
          const u32 k2 = (u32) (cw >> 1) & 0xFFFu;
          metas[r] = (u64) (k2 & 3u) | ((u64) (k2 >> 2) << TJ_META_LEN_SHIFT) | (cnt << TJ_META_COUNT_SHIFT) |
                     (0xffeull << TJ_META_MISM_SHIFT) | (flag << TJ_META_FLAG_SHIFT);
          mine++;
        }
      }
    }
    const u32 x = wave_inclusive_scan (mine);
    if (lane == 63) L.wsum[wave] = x;
    __syncthreads ();
    u32 wbase = 0, total = 0;
    for (int wv = 0; wv < AG_BLOCK / 64; wv++) { const u32 sm = L.wsum[wv]; if (wv < wave) wbase += sm; total += sm; }
    if (tid == 0) L.total = total ? atomicAdd (&fin->n_kept, total) : 0u;
    __syncthreads ();
    u64 at = (u64) L.total + wbase + x - mine;
#pragma unroll
    for (int r = 0; r < AG4_S / AG_BLOCK; r++)
      if (metas[r]) {
        const int slot = tid + r * AG_BLOCK;
        if (at < kept_cap) {
          u64 *q = kept + 3 * at; q[0] = L.c0[slot]; q[1] = L.c1[slot]; q[2] = metas[r];
          if (fine) atomicAdd (&fine[bin_of_record (L.c0[slot], L.c1[slot], metas[r], k, fine_bits)], 1u);
        }
        else flag_kept_overflow (fin);
        at++;
      }
    __threadfence_block ();
    __syncthreads ();
    n = L.n_ovf;
    { const u64 *t = src; src = dst; dst = (u64 *) t; }
    __syncthreads ();
  }
  plan_tail (fin, k, plan_cap, plan);
}

// ---------------------------------------------------------------------------------------------------------------
// sort key.  Reference order (src/hopo_counter.c:28-38): base, context[0], context[1], length, all descending, with
// length compared as a signed 10-bit value.  The strand flag is appended as the least significant digit so that the
// first and last member of a run of equal keys carry the OR of all flags.  The key is a bit string
//   [flag:2][length^0x200:10][ctx1:2k][ctx0:2k][base:1]   (least significant first), cut into 8-bit digits;
// the sort is an ascending LSD radix sort on the complemented digits.

__device__ __forceinline__ u32 key_digit (u64 c0, u64 c1, u64 meta, int pass, int k)
{
  const int o = 8 * pass;
  const u32 v12 = (u32) ((((meta >> TJ_META_LEN_SHIFT) & 0x3FFull) ^ 0x200ull) << 2) | (u32) ((meta >> TJ_META_FLAG_SHIFT) & 3ull);
  const u64 base = meta & 1ull;
  u32 d = 0;
  int lo;
  lo = o;                      if (lo < 12)                       d |= (u32) (v12 >> lo);
  lo = o - 12;                 if (lo < 2 * k && lo > -8)         d |= (u32) (lo >= 0 ? (c1 >> lo) : (c1 << -lo));
  lo = o - 12 - 2 * k;         if (lo < 2 * k && lo > -8)         d |= (u32) (lo >= 0 ? (c0 >> lo) : (c0 << -lo));
  lo = o - 12 - 4 * k;         if (lo < 1 && lo > -8)             d |= (u32) (base << -lo);
  return (~d) & 0xFFu;
}

__host__ int key_passes (int k) { return (13 + 4 * k + 7) / 8; }

#define RS_ITEMS      1024           // per workgroup: the sort only ever sees the kept set (1e5..1e6 records), so favour many small blocks
#define RS_WAVE_ITEMS 256

// lanes of the wavefront holding the same 8-bit digit as this lane
__device__ __forceinline__ u64 match_digit (u32 d, bool active)
{
  u64 peers = __ballot (active);
#pragma unroll
  for (int b = 0; b < 8; b++) {
    u64 bal = __ballot ((d >> b) & 1u);
    peers &= ((d >> b) & 1u) ? bal : ~bal;
  }
  return active ? peers : 0ull;
}

__global__ __launch_bounds__ (256)
void radix_count_kernel (const u64 *__restrict__ in, long n, int pass, int k, u32 *__restrict__ hist, int nblk)
{
  __shared__ u32 h[256];
  const int tid = threadIdx.x, lane = tid & 63;
  const u64 lt = (1ull << lane) - 1ull;
  h[tid] = 0;
  __syncthreads ();
  const long base = (long) blockIdx.x * RS_ITEMS;
  for (int i = tid; i < RS_ITEMS; i += 256) {            // uniform trip count: ballots are wave-wide
    long r = base + i;
    bool active = r < n;
    u32 d = 0;
    if (active) { const u64 *p = in + 3 * r; d = key_digit (p[0], p[1], p[2], pass, k); }
    u64 peers = match_digit (d, active);
    if (active && (peers & lt) == 0ull) atomicAdd (&h[d], (u32) __popcll (peers));   // one LDS atomic per distinct digit
  }
  __syncthreads ();
  hist[(long) tid * nblk + blockIdx.x] = h[tid];
}

__global__ __launch_bounds__ (256)
void radix_scatter_kernel (const u64 *__restrict__ in, u64 *__restrict__ out, long n, int pass, int k,
                           const u32 *__restrict__ offs, int nblk)
{
  __shared__ u32 wcnt[4][256];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < 1024; i += 256) (&wcnt[0][0])[i] = 0;
  __syncthreads ();
  const long wbase = (long) blockIdx.x * RS_ITEMS + (long) wave * RS_WAVE_ITEMS;
  const u64 lt = (1ull << lane) - 1ull;

  for (int r = 0; r < RS_WAVE_ITEMS / 64; r++) {         // count this wave's digits
    long idx = wbase + r * 64 + lane;
    bool active = idx < n;
    u32 d = 0;
    if (active) { const u64 *p = in + 3 * idx; d = key_digit (p[0], p[1], p[2], pass, k); }
    u64 peers = match_digit (d, active);
    if (active && (peers & lt) == 0ull) atomicAdd (&wcnt[wave][d], (u32) __popcll (peers));
  }
  __syncthreads ();
  {                                                      // digit tid: global offset of each wave's first item
    u32 run = offs[(long) tid * nblk + blockIdx.x];
    for (int w = 0; w < 4; w++) { u32 c = wcnt[w][tid]; wcnt[w][tid] = run; run += c; }
  }
  __syncthreads ();
  for (int r = 0; r < RS_WAVE_ITEMS / 64; r++) {         // stable scatter, wave-synchronous
    long idx = wbase + r * 64 + lane;
    bool active = idx < n;
    u32 d = 0;
    u64 a = 0, b = 0, m = 0;
    if (active) { const u64 *p = in + 3 * idx; a = p[0]; b = p[1]; m = p[2]; d = key_digit (a, b, m, pass, k); }
    u64 peers = match_digit (d, active);
    int leader = active ? (__ffsll ((long long) peers) - 1) : 0;
    u32 old = 0;
    if (active && lane == leader) old = atomicAdd (&wcnt[wave][d], (u32) __popcll (peers));
    old = __shfl (old, leader);
    if (active) {
      u64 *q = out + 3 * ((u64) old + (u64) __popcll (peers & lt));
      q[0] = a; q[1] = b; q[2] = m;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Ordering the kept set (the usual case: 1e5..1e6 records, where seven stable radix passes are all launch overhead):
// one partition by the leading bits of the key into ~n/32 bins, then every bin is rank-sorted in LDS by one wavefront.
// Keys compare as the reference's qsort does (src/hopo_counter.c:28-38): base, ctx0, ctx1, signed length, descending.

#define BS_RANK_MAX  256                // records of one bin a wavefront sorts in LDS; a fuller bin -> radix sort instead

__device__ __forceinline__ u32 bin_of_record (u64 c0, u64 c1, u64 meta, int k, int nbits)
{ // leading nbits (<= 1 + 4k) of [base:1][ctx0:2k][ctx1:2k], complemented: ascending bins = descending keys
  const int kb = nbits - 1;
  const u64 v = (2 * k >= kb) ? (c0 >> (2 * k - kb)) : ((c0 << (kb - 2 * k)) | (c1 >> (4 * k - kb)));
  const u32 x = ((u32) (meta & 1ull) << kb) | (u32) v;
  return ((1u << nbits) - 1u) - x;
}

// does record a come before record b?  (equal keys -- only the merge ever has them -- keep their input order)
__device__ __forceinline__ bool record_before (u64 a0, u64 a1, u64 am, u32 ai, u64 b0, u64 b1, u64 bm, u32 bi)
{
  const u32 ab = (u32) am & 1u, bb = (u32) bm & 1u;
  if (ab != bb) return ab > bb;
  if (a0 != b0) return a0 > b0;
  if (a1 != b1) return a1 > b1;
  const u32 al = ((u32) (am >> TJ_META_LEN_SHIFT) & 0x3FFu) ^ 0x200u, bl = ((u32) (bm >> TJ_META_LEN_SHIFT) & 0x3FFu) ^ 0x200u;
  if (al != bl) return al > bl;
  return ai < bi;
}

// Sorted inputs (the merge: every sample's histogram is in key order) put a wavefront's 64 consecutive records into one or
// two bins, and the device's global-atomic rate (~24 G/s, whatever the addresses) is what these kernels run at: with
// `grouped` the lanes of a wavefront that share a bin send ONE atomic (peeling one distinct bin per round).
// Returns the lane's position in its bin (scatter) -- or nothing useful for a pure count.
__device__ __forceinline__ u32 bin_reserve (u32 *__restrict__ cursors, u32 bin, bool active, int grouped)
{
  if (!grouped) return active ? atomicAdd (&cursors[bin], 1u) : 0u;
  const int lane = threadIdx.x & 63;
  u64 todo = __ballot (active);
  u32 pos = 0;
  while (todo) {
    const int leader = __ffsll ((long long) todo) - 1;
    const u32 lb = (u32) __builtin_amdgcn_readlane ((int) bin, leader);
    const u64 grp = __ballot (active && bin == lb) & todo;
    u32 base = 0;
    if (lane == leader) base = atomicAdd (&cursors[lb], (u32) __popcll (grp));
    base = (u32) __builtin_amdgcn_readlane ((int) base, leader);
    if ((grp >> lane) & 1ull) pos = base + (u32) __popcll (grp & ((1ull << lane) - 1ull));
    todo &= ~grp;
  }
  return pos;
}

__global__ __launch_bounds__ (256)
void bin_count_kernel (const u64 *__restrict__ in, long n, int k, int nbits, u32 *__restrict__ bins, uint4 *__restrict__ cov, long cov_vec, int grouped,
                       const FinPlan *__restrict__ plan = nullptr)
{ // (also empties the coverage table, which the sort pass three launches later fills: 16 bytes per store)
  if (plan) { if (!plan->ok) return; n = plan->n1; nbits = plan->nbits; cov_vec = plan->t / 2; }
  for (long i = (long) blockIdx.x * 256 + threadIdx.x; i < cov_vec; i += (long) gridDim.x * 256) cov[i] = make_uint4 (0, 0, 0, 0);
  for (long i0 = (long) blockIdx.x * 256; i0 < n; i0 += (long) gridDim.x * 256) {     // (wave-uniform trip count: ballots inside)
    const long i = i0 + threadIdx.x;
    u32 bin = 0;
    if (i < n) { const u64 *p = in + 3 * i; bin = bin_of_record (p[0], p[1], p[2], k, nbits); }
    (void) bin_reserve (bins, bin, i < n, grouped);
  }
}

// Exclusive prefix of n values by one workgroup of 1024 threads, WGS_CHUNK at a time with a running carry: coalesced
// load into LDS, every thread sums a contiguous slice, one scan over the slice sums, coalesced store.  Returns the
// total; vmax = largest value.
#define BS_MAXBINS (1 << BS_MAXBITS)
#define WGS_CHUNK  16384

#define WGS_PAD(i) ((i) + ((i) >> 5))      // one pad word per 32: a thread's contiguous slice does not collide with its neighbours' banks
struct WgScanLds { u32 a[WGS_CHUNK + WGS_CHUNK / 32 + 1]; u32 wsum[16]; u32 vmax; };

__device__ __forceinline__ u32 wg_exclusive_scan (const u32 *__restrict__ in, int n_all, u32 *__restrict__ out, u32 *__restrict__ out2, WgScanLds &L, u32 &vmax)
{
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) L.vmax = 0;
  u32 carry = 0;
  for (int base = 0; base < n_all; base += WGS_CHUNK) {
    const int n = min (WGS_CHUNK, n_all - base);
    const int per = (n + 1023) / 1024;
    for (int i = tid; i < n; i += 1024) L.a[WGS_PAD (i)] = in[base + i];
    __syncthreads ();
    u32 sum = 0, mx = 0;
    for (int j = 0; j < per; j++) { const int b = tid * per + j; if (b < n) { const u32 v = L.a[WGS_PAD (b)]; sum += v; mx = max (mx, v); } }
    const u32 incl = wave_inclusive_scan (sum);
    if (lane == 63) L.wsum[wave] = incl;
    for (int o = 32; o > 0; o >>= 1) mx = max (mx, (u32) __shfl_down ((int) mx, o));
    if (lane == 0 && mx) atomicMax (&L.vmax, mx);
    __syncthreads ();
    u32 run = carry + incl - sum, total = 0;
    for (int w = 0; w < 16; w++) { const u32 x = L.wsum[w]; if (w < wave) run += x; total += x; }
    for (int j = 0; j < per; j++) { const int b = tid * per + j; if (b < n) { const u32 v = L.a[WGS_PAD (b)]; L.a[WGS_PAD (b)] = run; run += v; } }
    __syncthreads ();
    for (int i = tid; i < n; i += 1024) { const u32 v = L.a[WGS_PAD (i)]; out[base + i] = v; if (out2) out2[base + i] = v; }
    carry += total;
    __syncthreads ();
  }
  vmax = L.vmax;
  return carry;
}

// bin counts -> binstart[0..nbins] and the scatter cursors (bins[] itself); a bin above rank_max switches the whole
// sort to the radix path (flag in FinCounts)
__global__ __launch_bounds__ (1024)
void bin_scan_kernel (u32 *__restrict__ bins, int nbins, u32 *__restrict__ binstart, u32 rank_max, FinCounts *fin, const FinPlan *__restrict__ plan = nullptr)
{
  __shared__ WgScanLds L;
  u32 vmax;
  if (plan) { if (!plan->ok) return; nbins = plan->nbins; }
  const u32 total = wg_exclusive_scan (bins, nbins, binstart, bins, L, vmax);
  if (threadIdx.x == 0) { binstart[nbins] = total; fin->sort_fallback = vmax > rank_max ? 1u : 0u; }
}

__global__ __launch_bounds__ (256)
void bin_scatter_kernel (const u64 *__restrict__ in, u64 *__restrict__ out, long n, int k, int nbits, u32 *__restrict__ cursors,
                         const FinPlan *__restrict__ plan = nullptr, uint4 *__restrict__ cov = nullptr)
{
  if (plan) { if (!plan->ok) return; n = plan->n1; nbits = plan->nbits; }
  // (no bin_count_kernel ran, see finalise_binned: the coverage table, which the next launch fills, is emptied here)
  if (cov) for (long i = (long) blockIdx.x * 256 + threadIdx.x; i < plan->t / 2; i += (long) gridDim.x * 256) cov[i] = make_uint4 (0, 0, 0, 0);
  for (long i = (long) blockIdx.x * 256 + threadIdx.x; i < n; i += (long) gridDim.x * 256) {
    const u64 *p = in + 3 * i;
    const u64 a = p[0], b = p[1], m = p[2];
    u64 *q = out + 3 * (u64) atomicAdd (&cursors[bin_of_record (a, b, m, k, nbits)], 1u);
    q[0] = a; q[1] = b; q[2] = m;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// exclusive scan of u32 arrays (three-kernel, recursive on block sums)

#define SC_ITEMS 4096

__device__ __forceinline__ u32 block_exclusive_scan_256 (u32 v, u32 *s_tmp, u32 &total)
{ // exclusive scan of one value per thread over 256 threads
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  u32 x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { u32 y = __shfl_up (x, o); if (lane >= o) x += y; }
  if (lane == 63) s_tmp[wave] = x;
  __syncthreads ();
  u32 wbase = 0;
  for (int w = 0; w < wave; w++) wbase += s_tmp[w];
  total = s_tmp[0] + s_tmp[1] + s_tmp[2] + s_tmp[3];
  __syncthreads ();
  return wbase + x - v;
}

__global__ __launch_bounds__ (256)
void scan_reduce_kernel (const u32 *__restrict__ in, long n, u32 *__restrict__ bsum)
{
  __shared__ u32 s_tmp[4];
  const long base = (long) blockIdx.x * SC_ITEMS + (long) threadIdx.x * 16;
  u32 s = 0;
  for (int i = 0; i < 16; i++) if (base + i < n) s += in[base + i];
  u32 total;
  block_exclusive_scan_256 (s, s_tmp, total);
  if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}

__global__ __launch_bounds__ (256)
void scan_apply_kernel (const u32 *__restrict__ in, u32 *__restrict__ out, long n, const u32 *__restrict__ boff, u32 *total_out)
{
  __shared__ u32 s_tmp[4];
  const long base = (long) blockIdx.x * SC_ITEMS + (long) threadIdx.x * 16;
  u32 v[16], s = 0;
  for (int i = 0; i < 16; i++) { v[i] = (base + i < n) ? in[base + i] : 0u; s += v[i]; }
  u32 total;
  u32 ex = block_exclusive_scan_256 (s, s_tmp, total) + (boff ? boff[blockIdx.x] : 0u);
  for (int i = 0; i < 16; i++) { if (base + i < n) out[base + i] = ex; ex += v[i]; }
  if (total_out && base <= n - 1 && n - 1 < base + 16) *total_out = ex;   // thread holding the last element
}

// ---------------------------------------------------------------------------------------------------------------
// reduce runs of equal keys (reference: src/hopo_counter.c:356-374), context index (:388-404), coverage (:419-438)

__device__ __forceinline__ bool same_key (const u64 *a, const u64 *b)
{ // base, context, length (not the flag, not the count)
  const u64 km = (3ull << TJ_META_BASE_SHIFT) | (0x3FFull << TJ_META_LEN_SHIFT);
  return a[0] == b[0] && a[1] == b[1] && ((a[2] ^ b[2]) & km) == 0ull;
}

__device__ __forceinline__ bool same_context (const u64 *a, const u64 *b)
{
  return a[0] == b[0] && a[1] == b[1] && ((a[2] ^ b[2]) & 3ull) == 0ull;
}

__device__ __forceinline__ int meta_count (u64 meta)
{ // signed 20-bit read-back
  int c = (int) ((meta >> TJ_META_COUNT_SHIFT) & 0xFFFFFull);
  return (c & 0x80000) ? c - 0x100000 : c;
}

__global__ void seg_heads_kernel (const u64 *__restrict__ rec, long n, u32 *__restrict__ flags, int context_only)
{
  for (long i = blockIdx.x * (long) blockDim.x + threadIdx.x; i < n; i += (long) gridDim.x * blockDim.x) {
    bool head = (i == 0);
    if (!head) head = context_only ? !same_context (rec + 3 * i, rec + 3 * (i - 1)) : !same_key (rec + 3 * i, rec + 3 * (i - 1));
    flags[i] = head ? 1u : 0u;
  }
}

__global__ void seg_headpos_kernel (const u32 *__restrict__ flags, const u32 *__restrict__ segid, long n, u32 *__restrict__ headpos)
{
  for (long i = blockIdx.x * (long) blockDim.x + threadIdx.x; i < n; i += (long) gridDim.x * blockDim.x)
    if (flags[i]) headpos[segid[i]] = (u32) i;
}

// one thread per context of the kept array: depth, index decision
__global__ void ctx_decide_kernel (const u64 *__restrict__ kept, long n1, const u32 *__restrict__ ctxpos, const u32 *n_ctx_p,
                                   int min_coverage, u32 *__restrict__ keep)
{
  const u32 n_ctx = *n_ctx_p;
  for (u32 j = blockIdx.x * blockDim.x + threadIdx.x; j < n_ctx; j += gridDim.x * blockDim.x) {
    const long s = ctxpos[j], e = (j + 1 < n_ctx) ? (long) ctxpos[j + 1] : n1;
    int depth = 0;
    for (long i = s; i < e; i++) depth += meta_count (kept[3 * i + 2]);
    keep[j] = depth >= min_coverage;
  }
}

__global__ void ctx_write_kernel (long n1, const u32 *__restrict__ ctxpos, const u32 *n_ctx_p, const u32 *__restrict__ keep,
                                  const u32 *__restrict__ outpos, int *__restrict__ idx_initial, int *__restrict__ idx_final)
{
  const u32 n_ctx = *n_ctx_p;
  for (u32 j = blockIdx.x * blockDim.x + threadIdx.x; j < n_ctx; j += gridDim.x * blockDim.x)
    if (keep[j]) {
      idx_initial[outpos[j]] = (int) ctxpos[j];
      idx_final[outpos[j]] = (j + 1 < n_ctx) ? (int) ctxpos[j + 1] : (int) n1;
    }
}

// coverage: pooled 31-bit-truncated flanks weighted by count, largest pooled weight wins (reference
// src/hopo_counter.c:419-438).  Open-addressing table in HBM, one 64-bit word per slot: key + 1 in the high half (0 =
// empty), the pooled weight in the low half.  A new key and its first weight go in with ONE compare-and-swap (the
// memory-side atomics are what this step runs at); a key that is already there gets a 32-bit add on the low half
// (wraps like the reference's int, never carries into the key).
__device__ __forceinline__ u64 cov_word (u32 key31, int w) { return ((u64) (key31 + 1u) << 32) | (u64) (u32) w; }

// The table addressed by the key itself (cov_plan_bits): one 64-bit add per flank.  The low half is the pooled weight (it
// wraps like the reference's int: the sum of sign-extended weights modulo 2^32); 2^33 per add keeps the high half
// non-zero for a key that was seen, whatever the weights' signs (|weight| < 2^30: a context's depth, the sum of at most
// 1024 counts of 20 bits), which is all that cov_max_part asks of it (and cannot wrap the word: a flank value is shared by
// at most 2 x 1024 records' both sides, far from 2^31 adds).
__device__ __forceinline__ void cov_direct_add (u32 key31, int w, u64 *__restrict__ tab)
{
  atomicAdd ((unsigned long long *) &tab[key31], (unsigned long long) (long long) w + (1ull << 33));
}

__device__ __forceinline__ void cov_add_from (u32 key31, u32 slot, int w, u64 *__restrict__ tab, int log2t)
{
  const u32 tmask = (1u << log2t) - 1u;
  for (u32 probe = 0; probe <= tmask; probe++) {
    const u64 old = atomicCAS ((unsigned long long *) &tab[slot], 0ull, (unsigned long long) cov_word (key31, w));
    if (old == 0ull) break;
    if ((u32) (old >> 32) == key31 + 1u) { atomicAdd (reinterpret_cast<u32 *> (&tab[slot]), (u32) w); break; }   // (little endian: low half first)
    slot = (slot + 1u) & tmask;
  }
}

__device__ __forceinline__ void cov_add (u32 key31, int w, u64 *__restrict__ tab, int log2t)
{
  cov_add_from (key31, (key31 * 2654435761u) >> (32 - log2t), w, tab, log2t);
}

// two keys at once: both compare-and-swaps are in flight together
__device__ __forceinline__ void cov_add2 (u32 ka, u32 kb, int w, u64 *__restrict__ tab, int log2t)
{
  const u32 tmask = (1u << log2t) - 1u;
  const u32 pa = (ka * 2654435761u) >> (32 - log2t), pb = (kb * 2654435761u) >> (32 - log2t);
  const u64 oa = atomicCAS ((unsigned long long *) &tab[pa], 0ull, (unsigned long long) cov_word (ka, w));
  const u64 ob = atomicCAS ((unsigned long long *) &tab[pb], 0ull, (unsigned long long) cov_word (kb, w));
  if (oa != 0ull) {
    if ((u32) (oa >> 32) == ka + 1u) atomicAdd (reinterpret_cast<u32 *> (&tab[pa]), (u32) w);
    else cov_add_from (ka, (pa + 1u) & tmask, w, tab, log2t);
  }
  // (both keys equal and the slot empty: the second compare-and-swap has seen the first one's word and adds to it)
  if (ob != 0ull) {
    if ((u32) (ob >> 32) == kb + 1u) atomicAdd (reinterpret_cast<u32 *> (&tab[pb]), (u32) w);
    else cov_add_from (kb, (pb + 1u) & tmask, w, tab, log2t);
  }
}

// the same in two halves, so that the caller can work while the two compare-and-swaps make their round trip to memory
__device__ __forceinline__ void cov_add2_issue (u32 ka, u32 kb, int w, u64 *__restrict__ tab, int log2t, u64 &oa, u64 &ob)
{
  const u32 pa = (ka * 2654435761u) >> (32 - log2t), pb = (kb * 2654435761u) >> (32 - log2t);
  oa = atomicCAS ((unsigned long long *) &tab[pa], 0ull, (unsigned long long) cov_word (ka, w));
  ob = atomicCAS ((unsigned long long *) &tab[pb], 0ull, (unsigned long long) cov_word (kb, w));
}
__device__ __forceinline__ void cov_add2_finish (u32 ka, u32 kb, int w, u64 *__restrict__ tab, int log2t, u64 oa, u64 ob)
{
  const u32 tmask = (1u << log2t) - 1u;
  const u32 pa = (ka * 2654435761u) >> (32 - log2t), pb = (kb * 2654435761u) >> (32 - log2t);
  if (oa != 0ull) {
    if ((u32) (oa >> 32) == ka + 1u) atomicAdd (reinterpret_cast<u32 *> (&tab[pa]), (u32) w);
    else cov_add_from (ka, (pa + 1u) & tmask, w, tab, log2t);
  }
  if (ob != 0ull) {
    if ((u32) (ob >> 32) == kb + 1u) atomicAdd (reinterpret_cast<u32 *> (&tab[pb]), (u32) w);
    else cov_add_from (kb, (pb + 1u) & tmask, w, tab, log2t);
  }
}

__global__ void cov_insert_kernel (const u64 *__restrict__ kept, long n1, u64 *__restrict__ tab, int log2t)
{
  for (long i = blockIdx.x * (long) blockDim.x + threadIdx.x; i < 2 * n1; i += (long) gridDim.x * blockDim.x) {
    const long r = (i < n1) ? i : i - n1;
    cov_add ((u32) (kept[3 * r + (i < n1 ? 0 : 1)] & 0x7FFFFFFFull), meta_count (kept[3 * r + 2]), tab, log2t);
  }
}

// largest pooled weight of the table slice this workgroup looks at -> atomicMax (one atomic per workgroup)
__device__ __forceinline__ void cov_max_part (const u64 *__restrict__ tab, long t, int *result, int *s_best)
{
  int best = INT_MIN;
  for (long i = blockIdx.x * (long) blockDim.x + threadIdx.x; i < t; i += (long) gridDim.x * blockDim.x) {
    const u64 v = tab[i];
    if (v >> 32) best = max (best, (int) (u32) v);
  }
  for (int o = 32; o > 0; o >>= 1) best = max (best, __shfl_down (best, o));
  if ((threadIdx.x & 63) == 0) s_best[threadIdx.x >> 6] = best;
  __syncthreads ();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int) blockDim.x / 64; w++) best = max (best, s_best[w]);
    if (best != INT_MIN) atomicMax (result, best);
  }
}

__global__ __launch_bounds__ (256)
void cov_max_kernel (const u64 *__restrict__ tab, long t, int *result)
{
  __shared__ int s_best[4];
  cov_max_part (tab, t, result, s_best);
}

// One wavefront per bin: the bin's records go to LDS, every lane counts the records that come before its own, and that
// rank is the record's place in the output -- plus everything else the finalise step needs from the sorted order: a
// context (base, ctx0, ctx1) never straddles bins (the bin is a prefix of it), so its depth, its first record and its
// size are all found among the bin's records -- and the coverage table takes the records in any order.
//   binctx[bin]        contexts of the bin that reach min_coverage
//   tstart/tend[st+o]  index range of the o-th such context of the bin (st = first record of the bin)
#define BSI_WAVES 1

// the value that lane (lane ^ J) holds, without the LDS pipe: quad permutes for 1 and 2, a row shift either way for 4 and 8,
// gfx950's row and half swaps for 16 and 32 (a ds_bpermute per word and step kept the LDS pipe busy for half of the sort
// kernel's time: 168 of them per bin)
template <int J>
__device__ __forceinline__ u32 lane_xor (u32 v, int lane)
{
  if constexpr (J == 1) return (u32) __builtin_amdgcn_update_dpp (0, (int) v, 0xB1, 0xF, 0xF, false);        // quad_perm [1,0,3,2]
  else if constexpr (J == 2) return (u32) __builtin_amdgcn_update_dpp (0, (int) v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
  else if constexpr (J == 4 || J == 8) {
    const u32 up = (u32) __builtin_amdgcn_update_dpp (0, (int) v, 0x100 + J, 0xF, 0xF, false);               // row_shl: lane i gets lane i + J
    const u32 dn = (u32) __builtin_amdgcn_update_dpp (0, (int) v, 0x110 + J, 0xF, 0xF, false);               // row_shr: lane i gets lane i - J
    return (lane & J) ? dn : up;
  }
  else if constexpr (J == 16) { const auto r = __builtin_amdgcn_permlane16_swap (v, v, false, false); return (lane & 16) ? r[0] : r[1]; }
  else { const auto r = __builtin_amdgcn_permlane32_swap (v, v, false, false); return (lane & 32) ? r[0] : r[1]; }
}

__global__ __launch_bounds__ (64 * BSI_WAVES)
void bin_sort_index_kernel (const u64 *__restrict__ in, u64 *__restrict__ out, const u32 *__restrict__ binstart, int nbins, const FinCounts *fin,
                            int min_coverage, u64 *__restrict__ cov_tab, int log2t,
                            u32 *__restrict__ binctx, u32 *__restrict__ tstart, u32 *__restrict__ tend, const FinPlan *__restrict__ plan = nullptr,
                            int cov_direct = 0)
{
  // (2 KB of LDS per wavefront: the usual bin lives in registers, a fuller one reads its records where they lie -- with the
  // records of up to 256 staged in LDS, 8 KB, twenty wavefronts fitted a CU instead of twenty-eight, and a wavefront's
  // time is a chain of memory round trips that only other wavefronts can fill)
  __shared__ alignas (16) u32 hd[BSI_WAVES][BS_RANK_MAX];
  __shared__ u32 sz[BSI_WAVES][BS_RANK_MAX];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // (the first bin's bounds are asked for together with the plan: one round trip less in the chain; `nbins` as passed is
  // the layout's size, which the array has whatever the plan says)
  const int bin0 = (int) blockIdx.x * BSI_WAVES + wave;
  u32 st0 = 0, en0 = 0;
  if (bin0 < nbins) { st0 = binstart[bin0]; en0 = binstart[bin0 + 1]; }
  if (plan) { if (!plan->ok) return; nbins = plan->nbins; log2t = plan->log2t; cov_direct = (int) plan->cov_direct; }
  if (fin->sort_fallback) return;                       // some bin is too full: the caller takes the radix path
  u32 *H = hd[wave], *E = sz[wave];
  for (int bin = bin0; bin < nbins; bin += gridDim.x * BSI_WAVES) {
    const u32 st = (bin == bin0) ? st0 : binstart[bin], s = ((bin == bin0) ? en0 : binstart[bin + 1]) - st;
    if (s == 0) { if (lane == 0) binctx[bin] = 0; continue; }
    const u64 *R = in + 3 * (u64) st;                   // the bin's records (a fuller bin reads them from here, the same record in every lane)
    u32 nkeep = 0;
    if (s <= 64u) {
      // The usual bin (a record per lane at most): a bitonic sort of the records across the wavefront's lanes -- 21
      // compare-and-exchange steps of some thirty instructions -- instead of every lane comparing its record with every
      // record of the bin (s / 4 turns of 180); sorted, a record's place is its lane, a context's records are neighbours,
      // and its size, depth and place among the contexts that are kept come out of two ballots and one prefix sum.
      const bool valid = (u32) lane < s;
      u64 a0 = 0, a1 = 0, am = 0;
      if (valid) { a0 = R[3 * lane]; a1 = R[3 * lane + 1]; am = R[3 * lane + 2]; }   // (one record per lane: its three loads are in flight together)
      // (a hashed coverage table: the record's two compare-and-swaps go out now and are looked at after the sort -- their round
      // trip to memory, a few microseconds, runs under it, and they stay with the lane that sent them.  Tried: reading the two
      // home slots first and adding where the key is already there, one atomic per flank instead of two -- no change, 177
      // against 179 us on the long-read sample: it is the random 8-byte accesses to a 64 MB table that this kernel waits for
      // there, whatever their kind)
      const int w = meta_count (am);
      u64 cas_a = 0, cas_b = 0;
      if (!cov_direct && valid) cov_add2_issue ((u32) (a0 & 0x7FFFFFFFull), (u32) (a1 & 0x7FFFFFFFull), w, cov_tab, log2t, cas_a, cas_b);
      // order: records before empty lanes, then base, ctx0, ctx1, signed length, all descending (record_before); kh holds
      // what comes before the contexts, kl what comes after them (a lane's own number last: no two lanes compare equal)
      u32 kh = valid ? 2u | ((u32) am & 1u) : 0u;
      u32 kl = ((((u32) (am >> TJ_META_LEN_SHIFT) & 0x3FFu) ^ 0x200u) << 6) | (63u - (u32) lane);
      u32 x0 = (u32) a0, x1 = (u32) (a0 >> 32), y0 = (u32) a1, y1 = (u32) (a1 >> 32), m0 = (u32) am, m1 = (u32) (am >> 32);
      auto step = [&] (auto jc, int kk) {
        constexpr int J = decltype (jc)::value;
        const u32 ox0 = lane_xor<J> (x0, lane), ox1 = lane_xor<J> (x1, lane), oy0 = lane_xor<J> (y0, lane), oy1 = lane_xor<J> (y1, lane);
        const u32 om0 = lane_xor<J> (m0, lane), om1 = lane_xor<J> (m1, lane), okh = lane_xor<J> (kh, lane), okl = lane_xor<J> (kl, lane);
        const u64 mx = ((u64) x1 << 32) | x0, my = ((u64) y1 << 32) | y0, ox = ((u64) ox1 << 32) | ox0, oy = ((u64) oy1 << 32) | oy0;
        const bool mine_first = (kh > okh) | ((kh == okh) & ((mx > ox) | ((mx == ox) & ((my > oy) | ((my == oy) & (kl > okl))))));
        // the lower lane of a pair keeps the record that comes first where the run of kk lanes is to ascend, the other one where it is to descend
        const bool want_first = ((lane & kk) == 0) == ((lane & J) == 0);
        const bool take = mine_first != want_first;       // the partner's record
        x0 = take ? ox0 : x0; x1 = take ? ox1 : x1; y0 = take ? oy0 : y0; y1 = take ? oy1 : y1;
        m0 = take ? om0 : m0; m1 = take ? om1 : m1; kh = take ? okh : kh; kl = take ? okl : kl;
      };
#pragma unroll
      for (int kk = 2; kk <= 64; kk <<= 1) {
        if (kk >= 64) step (std::integral_constant<int, 32> (), kk);
        if (kk >= 32) step (std::integral_constant<int, 16> (), kk);
        if (kk >= 16) step (std::integral_constant<int, 8> (), kk);
        if (kk >= 8) step (std::integral_constant<int, 4> (), kk);
        if (kk >= 4) step (std::integral_constant<int, 2> (), kk);
        step (std::integral_constant<int, 1> (), kk);
      }
      // lane i holds the i-th record of the bin
      const bool have = kh >= 2u;                          // (the records are in lanes 0 .. s - 1 again)
      const u64 b0 = ((u64) x1 << 32) | x0, b1 = ((u64) y1 << 32) | y0, bm = ((u64) m1 << 32) | m0;
      if (have) { u64 *q = out + 3 * ((u64) st + (u32) lane); q[0] = b0; q[1] = b1; q[2] = bm; }
      // a context's first record: the lane before holds another context (or it is lane 0)
      const u32 px0 = (u32) __builtin_amdgcn_update_dpp (0, (int) x0, 0x138, 0xF, 0xF, false), px1 = (u32) __builtin_amdgcn_update_dpp (0, (int) x1, 0x138, 0xF, 0xF, false);
      const u32 py0 = (u32) __builtin_amdgcn_update_dpp (0, (int) y0, 0x138, 0xF, 0xF, false), py1 = (u32) __builtin_amdgcn_update_dpp (0, (int) y1, 0x138, 0xF, 0xF, false);
      const u32 pkh = (u32) __builtin_amdgcn_update_dpp (0, (int) kh, 0x138, 0xF, 0xF, false);
      const bool head = have & ((lane == 0) | (px0 != x0) | (px1 != x1) | (py0 != y0) | (py1 != y1) | (pkh != kh));
      const u64 heads = __ballot (head);
      const int cnt = have ? meta_count (bm) : 0;
      const u32 incl = wave_inclusive_scan ((u32) cnt);    // (wraps like the int sum it stands for)
      const u64 above = (lane == 63) ? 0ull : (heads >> (lane + 1));
      const u32 next_head = above ? (u32) lane + (u32) __ffsll ((long long) above) : s;   // first lane of the next context (or the end)
      const u32 last_incl = (u32) __builtin_amdgcn_ds_bpermute ((int) ((next_head - 1u) << 2), (int) incl);
      const int depth = (int) (last_incl - (incl - (u32) cnt));
      const bool keep = head & (depth >= min_coverage);
      const u64 keeps = __ballot (keep);
      if (keep) {
        const u32 o = (u32) __popcll (keeps & ((1ull << lane) - 1ull));
        tstart[st + o] = st + (u32) lane;
        tend[st + o] = st + next_head;
      }
      nkeep = (u32) __popcll (keeps);
      // The coverage table takes a context's records in one go: they pool under the same two flanks, so the context's depth
      // is added once for each (the sum of the counts, wrapping like the reference's int, whatever the order) -- the
      // memory-side atomics are what this kernel runs at (24 G/s whatever the addresses: 470 k adds were 20 of its 31 us on
      // the headline sample), and a context has 2.3 records on average.
      // (only where the table is addressed by the flank itself: a plain add needs no answer)
      if (cov_direct) {
        if (head) { cov_direct_add ((u32) (b0 & 0x7FFFFFFFull), depth, cov_tab); cov_direct_add ((u32) (b1 & 0x7FFFFFFFull), depth, cov_tab); }
      }
      else if (valid) {
        asm volatile ("" : "+v"(cas_a), "+v"(cas_b));      // (or the compiler tests them for zero, and waits, ahead of the sort)
        cov_add2_finish ((u32) (a0 & 0x7FFFFFFFull), (u32) (a1 & 0x7FFFFFFFull), w, cov_tab, log2t, cas_a, cas_b);
      }
      if (lane == 0) binctx[bin] = nkeep;
      asm volatile ("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier ();
      continue;
    }
    for (u32 t0 = 0; t0 < s; t0 += 64) {
      const u32 t = t0 + lane;
      bool keep = false;
      if (t < s) {
        const u64 a0 = R[3 * t], a1 = R[3 * t + 1], am = R[3 * t + 2];
        u32 rank = 0, ctx_before = 0, ctx_size = 0;
        int depth = 0;
        // (the bin's records four at a time, the same for every lane: twelve LDS reads in flight for one wait, and the
        // comparison -- record_before, spelt without branches -- is a chain of lane masks: as nested ifs, one record per turn,
        // it was four exec regions and a wait for the reads in every turn, 220 cycles for 30 instructions)
        const u32 abase = (u32) am & 1u, alen = ((u32) (am >> TJ_META_LEN_SHIFT) & 0x3FFu) ^ 0x200u;
        for (u32 j0 = 0; j0 < s; j0 += 4) {
          u64 b0[4], b1[4], bm[4];
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const u32 jj = (j0 + (u32) i < s) ? j0 + (u32) i : s - 1u;
            b0[i] = R[3 * jj]; b1[i] = R[3 * jj + 1]; bm[i] = R[3 * jj + 2];
          }
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const u32 j = j0 + (u32) i;
            const bool live = j < s;                        // (uniform)
            const u32 bbase = (u32) bm[i] & 1u, blen = ((u32) (bm[i] >> TJ_META_LEN_SHIFT) & 0x3FFu) ^ 0x200u;
            const bool e0 = b0[i] == a0, e1 = b1[i] == a1, eb = bbase == abase;
            const bool before = live & ((bbase > abase) | (eb & ((b0[i] > a0) | (e0 & ((b1[i] > a1) | (e1 & ((blen > alen) | ((blen == alen) & (j < t)))))))));
            const bool same = live & e0 & e1 & (((u32) (bm[i] ^ am) & 3u) == 0u);
            rank += before ? 1u : 0u;
            ctx_size += same ? 1u : 0u;
            depth += same ? meta_count (bm[i]) : 0;
            ctx_before += (same & before) ? 1u : 0u;
          }
        }
        u64 *q = out + 3 * ((u64) st + rank);
        q[0] = a0; q[1] = a1; q[2] = am;
        keep = (ctx_before == 0u) && depth >= min_coverage;      // first record of its context, context deep enough
        H[t] = rank | (keep ? 0x80000000u : 0u);
        E[t] = ctx_size;
        if (ctx_before == 0u) {                            // (the context's depth, once for each flank: see above)
          if (cov_direct) { cov_direct_add ((u32) (a0 & 0x7FFFFFFFull), depth, cov_tab); cov_direct_add ((u32) (a1 & 0x7FFFFFFFull), depth, cov_tab); }
          else cov_add2 ((u32) (a0 & 0x7FFFFFFFull), (u32) (a1 & 0x7FFFFFFFull), depth, cov_tab, log2t);
        }
      }
      nkeep += (u32) __popcll (__ballot (keep));
    }
    asm volatile ("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier ();
    for (u32 t = lane; t < s; t += 64) {
      const u32 h = H[t];
      if (h & 0x80000000u) {
        const u32 rank = h & 0x7FFFFFFFu;
        u32 o = 0;
        for (u32 j = 0; j < s; j += 4) {                  // (four entries per LDS read; entries past the bin's end hold an earlier bin's)
          const uint4 g = *reinterpret_cast<const uint4 *> (&H[j]);
          o += ((g.x & 0x80000000u) && (g.x & 0x7FFFFFFFu) < rank) ? 1u : 0u;
          o += (j + 1u < s && (g.y & 0x80000000u) && (g.y & 0x7FFFFFFFu) < rank) ? 1u : 0u;
          o += (j + 2u < s && (g.z & 0x80000000u) && (g.z & 0x7FFFFFFFu) < rank) ? 1u : 0u;
          o += (j + 3u < s && (g.w & 0x80000000u) && (g.w & 0x7FFFFFFFu) < rank) ? 1u : 0u;
        }
        tstart[st + o] = st + rank;
        tend[st + o] = st + rank + E[t];
      }
    }
    if (lane == 0) binctx[bin] = nkeep;
    asm volatile ("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier ();
  }
}

// one workgroup: exclusive prefix of binctx -> binout, number of index ranges, coverage start value
__global__ __launch_bounds__ (1024)
void bin_ctx_scan_kernel (const u32 *__restrict__ binctx, int nbins, u32 *__restrict__ binout, FinCounts *fin, const FinPlan *__restrict__ plan = nullptr)
{
  __shared__ WgScanLds L;
  u32 vmax;
  if (plan) { if (!plan->ok) return; nbins = plan->nbins; }
  const u32 total = wg_exclusive_scan (binctx, nbins, binout, nullptr, L, vmax);
  if (threadIdx.x == 0 && !fin->sort_fallback) { fin->n_idx = total; fin->coverage = INT_MIN; }
}

// index ranges in order (reference: idx_initial / idx_final, src/hopo_counter.c:388-404) + the coverage maximum
__global__ __launch_bounds__ (256)
void bin_ctx_write_kernel (const u32 *__restrict__ binstart, const u32 *__restrict__ binctx, const u32 *__restrict__ binout, int nbins,
                           const u32 *__restrict__ tstart, const u32 *__restrict__ tend, int *__restrict__ idx_initial, int *__restrict__ idx_final,
                           const u64 *__restrict__ cov_tab, long t, FinCounts *fin, const FinPlan *__restrict__ plan = nullptr)
{
  __shared__ int s_best[4];
  if (plan) { if (!plan->ok) return; nbins = plan->nbins; t = plan->t; }
  if (fin->sort_fallback) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int bin = blockIdx.x * 4 + wave; bin < nbins; bin += gridDim.x * 4) {
    const u32 cnt = binctx[bin], st = binstart[bin], o0 = binout[bin];
    for (u32 o = lane; o < cnt; o += 64) { idx_initial[o0 + o] = (int) tstart[st + o]; idx_final[o0 + o] = (int) tend[st + o]; }
  }
  cov_max_part (cov_tab, t, &fin->coverage, s_best);
}

// ---- cross-sample merge on the bins --------------------------------------------------------------------------------
// The samples' histograms, back to back, are partitioned by the same leading key bits; one wavefront per bin then finds
// the distinct keys of its bin (the lowest-numbered record of each key represents it), their order, and the depth of
// each key over all samples.  MG_RANK_MAX bounds a bin (a fuller one -> radix path).
#define MG_RANK_MAX 256
#define TJ_META_SAMPLE_SHIFT 52          // bits 52..63 of the bitfield word are unused by hopo_element

__global__ __launch_bounds__ (256)
void merge_scatter_kernel (const u64 *__restrict__ in, u64 *__restrict__ out, long n, int k, int nbits, u32 *__restrict__ cursors,
                           const long *__restrict__ starts, int n_samples)
{ // bin_scatter_kernel + the sample index of every record written into the spare bits of its meta word
  for (long i0 = (long) blockIdx.x * 256; i0 < n; i0 += (long) gridDim.x * 256) {
    const long i = i0 + threadIdx.x;
    const bool active = i < n;
    u64 a = 0, b = 0, m = 0;
    int lo = 0;
    if (active) {
      int hi = n_samples;                               // starts[s] <= i < starts[s + 1]
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (starts[mid] <= i) lo = mid; else hi = mid; }
      const u64 *p = in + 3 * i;
      a = p[0]; b = p[1]; m = p[2];
    }
    const u32 pos = bin_reserve (cursors, active ? bin_of_record (a, b, m, k, nbits) : 0u, active, 1);
    if (active) {
      u64 *q = out + 3 * (u64) pos;
      q[0] = a; q[1] = b; q[2] = (m & ((1ull << TJ_META_SAMPLE_SHIFT) - 1ull)) | ((u64) lo << TJ_META_SAMPLE_SHIFT);
    }
  }
}

__device__ __forceinline__ bool merge_same_key (u64 a0, u64 a1, u64 am, u64 b0, u64 b1, u64 bm)
{
  const u64 km = (3ull << TJ_META_BASE_SHIFT) | (0x3FFull << TJ_META_LEN_SHIFT);
  return a0 == b0 && a1 == b1 && ((am ^ bm) & km) == 0ull;
}

// per bin: binctx[bin] = distinct keys; per record (at its place st + t in the partitioned array): tpos = rank of its
// key among the bin's distinct keys, ttot = depth of the key over all samples if the record represents its key, else ~0.
// Inside a bin the base is the same (it is the bin's leading bit), so keys compare as (ctx0, ctx1, length ^ 0x200).
__global__ __launch_bounds__ (64)
void bin_merge_kernel (const u64 *__restrict__ rec, const u32 *__restrict__ binstart, int nbins, const FinCounts *fin,
                       u32 *__restrict__ binctx, u32 *__restrict__ tpos, u32 *__restrict__ ttot)
{
  __shared__ u64 R[3 * MG_RANK_MAX];
  __shared__ alignas (16) u32 H[MG_RANK_MAX];           // records in front of this one's key | representative << 31
  if (fin->sort_fallback) return;
  const int lane = threadIdx.x;
  for (int bin = blockIdx.x; bin < nbins; bin += gridDim.x) {
    const u32 st = binstart[bin], s = binstart[bin + 1] - st;
    if (s == 0) { if (lane == 0) binctx[bin] = 0; continue; }
    for (u32 t = lane; t < s; t += 64) {                // one record per lane: its three loads are in flight together
      const u64 *p = rec + 3 * (u64) (st + t);
      const u64 v0 = p[0], v1 = p[1], v2 = p[2];
      R[3 * t] = v0; R[3 * t + 1] = v1;
      R[3 * t + 2] = (v2 & ~(0x3FFull << TJ_META_LEN_SHIFT)) | ((((v2 >> TJ_META_LEN_SHIFT) & 0x3FFull) ^ 0x200ull) << TJ_META_LEN_SHIFT);   // length as an unsigned sort key
    }
    asm volatile ("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier ();
    const u64 lmask = 0x3FFull << TJ_META_LEN_SHIFT;
    u32 nrep = 0;
    for (u32 t0 = 0; t0 < s; t0 += 64) {                // pass 1: records in front, who represents its key, the key's depth
      const u32 t = t0 + lane;
      bool rep = false;
      if (t < s) {
        const u64 a0 = R[3 * t], a1 = R[3 * t + 1], al = R[3 * t + 2] & lmask;
        rep = true;
        u32 tot = 0, r = 0, fl = 0;
        for (u32 j0 = 0; j0 < s; j0 += 4) {               // (four records per turn of LDS reads, no branch: see bin_sort_index_kernel)
          u64 b0[4], b1[4], bm[4];
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const u32 jj = (j0 + (u32) i < s) ? j0 + (u32) i : s - 1u;
            b0[i] = R[3 * jj]; b1[i] = R[3 * jj + 1]; bm[i] = R[3 * jj + 2];
          }
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const u32 j = j0 + (u32) i;
            const bool live = j < s;                        // (uniform)
            const u64 bl = bm[i] & lmask;
            const bool same = live & (b0[i] == a0) & (b1[i] == a1) & (bl == al);
            const bool before = live & ((b0[i] > a0) | ((b0[i] == a0) & ((b1[i] > a1) | ((b1[i] == a1) & (bl > al)))));
            r += before ? 1u : 0u;
            rep = rep & !(same & (j < t));
            tot += same ? (u32) meta_count (bm[i]) : 0u;
            fl |= same ? (u32) (bm[i] >> TJ_META_FLAG_SHIFT) & 7u : 0u;
          }
        }
        H[t] = r | (rep ? 0x80000000u : 0u);
        // (the key's depth over all samples, 20-bit store, and the strands any sample saw it on)
        ttot[st + t] = rep ? ((tot & 0xFFFFFu) | (fl << 28)) : 0xFFFFFFFFu;
      }
      nrep += (u32) __popcll (__ballot (rep));
    }
    asm volatile ("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier ();
    for (u32 t = lane; t < s; t += 64) {                // pass 2: distinct keys in front = representatives with fewer records in front
      const u32 r = H[t] & 0x7FFFFFFFu;
      u32 d = 0;
      for (u32 j = 0; j < s; j += 4) {                    // (four entries per LDS read; entries past the bin's end hold an earlier bin's)
        const uint4 h = *reinterpret_cast<const uint4 *> (&H[j]);
        d += ((h.x & 0x80000000u) && (h.x & 0x7FFFFFFFu) < r) ? 1u : 0u;
        d += (j + 1u < s && (h.y & 0x80000000u) && (h.y & 0x7FFFFFFFu) < r) ? 1u : 0u;
        d += (j + 2u < s && (h.z & 0x80000000u) && (h.z & 0x7FFFFFFFu) < r) ? 1u : 0u;
        d += (j + 3u < s && (h.w & 0x80000000u) && (h.w & 0x7FFFFFFFu) < r) ? 1u : 0u;
      }
      tpos[st + t] = d;
    }
    if (lane == 0) binctx[bin] = nrep;
    asm volatile ("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier ();
  }
}

// union keys in order with the total depth in the count field, and the per-sample depth matrix
__global__ __launch_bounds__ (256)
void bin_merge_write_kernel (const u64 *__restrict__ rec, const u32 *__restrict__ binstart, const u32 *__restrict__ binout, int nbins, const FinCounts *fin,
                             const u32 *__restrict__ tpos, const u32 *__restrict__ ttot, int n_samples,
                             u64 *__restrict__ keys, int *__restrict__ counts, long cap)
{
  if (fin->sort_fallback) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int bin = blockIdx.x * 4 + wave; bin < nbins; bin += gridDim.x * 4) {
    const u32 st = binstart[bin], s = binstart[bin + 1] - st, o0 = binout[bin];
    for (u32 t0 = 0; t0 < s; t0 += 64) {                // representatives come first in a bin's index order: rows are
      const u32 t = t0 + lane;                          // zeroed (same wave, program order) before anybody fills them
      if (t < s) {
        const long u = (long) o0 + tpos[st + t];
        if (u < cap) {
          const u64 a0 = rec[3 * (u64) (st + t)], a1 = rec[3 * (u64) (st + t) + 1], am = rec[3 * (u64) (st + t) + 2];
          const u32 tot = ttot[st + t];
          if (tot != 0xFFFFFFFFu) {
            for (int q = 0; q < n_samples; q++) counts[u * n_samples + q] = 0;
            u64 m = am & ((1ull << TJ_META_SAMPLE_SHIFT) - 1ull);
            m = (m & ~(0xFFFFFull << TJ_META_COUNT_SHIFT)) | (((u64) tot & 0xFFFFFull) << TJ_META_COUNT_SHIFT);
            m = (m & ~(7ull << TJ_META_FLAG_SHIFT)) | ((u64) ((tot >> 28) & 7u) << TJ_META_FLAG_SHIFT);
            keys[3 * u] = a0; keys[3 * u + 1] = a1; keys[3 * u + 2] = m;
          }
        }
      }
      __builtin_amdgcn_wave_barrier ();
      asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
      if (t < s) {
        const long u = (long) o0 + tpos[st + t];
        if (u < cap) {
          const u64 am = rec[3 * (u64) (st + t) + 2];
          counts[u * n_samples + (int) (am >> TJ_META_SAMPLE_SHIFT)] = meta_count (am);
        }
      }
    }
  }
}

__global__ void set_int_kernel (int *p, int v) { *p = v; }

// ---- cross-sample merge, radix path (context-keyed union of the samples' histograms; reference precursor: src/genome_set.c:250-289)

__global__ void merge_tag_kernel (const u64 *__restrict__ in, u64 *__restrict__ out, long n, const long *__restrict__ starts, int n_samples)
{ // copy + write the sample index of every record into the spare bits of its meta word
  for (long i = blockIdx.x * (long) blockDim.x + threadIdx.x; i < n; i += (long) gridDim.x * blockDim.x) {
    int lo = 0, hi = n_samples;                         // starts[s] <= i < starts[s + 1]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (starts[mid] <= i) lo = mid; else hi = mid; }
    out[3 * i] = in[3 * i]; out[3 * i + 1] = in[3 * i + 1];
    out[3 * i + 2] = (in[3 * i + 2] & ((1ull << TJ_META_SAMPLE_SHIFT) - 1ull)) | ((u64) lo << TJ_META_SAMPLE_SHIFT);
  }
}

__global__ void merge_write_kernel (const u64 *__restrict__ rec, long n, const u32 *__restrict__ flags, const u32 *__restrict__ segid,
                                    int n_samples, u64 *__restrict__ keys, int *__restrict__ counts, u32 *__restrict__ totals, u32 *__restrict__ orflags, long cap)
{
  for (long i = blockIdx.x * (long) blockDim.x + threadIdx.x; i < n; i += (long) gridDim.x * blockDim.x) {
    const u32 seg = segid[i] + flags[i] - 1u;           // exclusive scan of the head flags: heads before i (+ itself) - 1
    if ((long) seg >= cap) continue;
    const u64 meta = rec[3 * i + 2];
    const int sample = (int) (meta >> TJ_META_SAMPLE_SHIFT);
    const int cnt = meta_count (meta);
    counts[(long) seg * n_samples + sample] = cnt;
    atomicAdd (&totals[seg], (u32) cnt);
    atomicOr (&orflags[seg], (u32) (meta >> TJ_META_FLAG_SHIFT) & 7u);
    if (flags[i]) { keys[3 * (long) seg] = rec[3 * i]; keys[3 * (long) seg + 1] = rec[3 * i + 1]; keys[3 * (long) seg + 2] = meta; }
  }
}

__global__ void merge_totals_kernel (u64 *__restrict__ keys, const u32 *__restrict__ totals, const u32 *__restrict__ orflags, const u32 *n_seg_p, long cap)
{ // count field := depth over all samples (20-bit store), canon_flag := the strands any sample saw the key on
  const long n_seg = min ((long) *n_seg_p, cap);
  for (long j = blockIdx.x * (long) blockDim.x + threadIdx.x; j < n_seg; j += (long) gridDim.x * blockDim.x) {
    u64 m = keys[3 * j + 2] & ((1ull << TJ_META_SAMPLE_SHIFT) - 1ull);
    m = (m & ~(0xFFFFFull << TJ_META_COUNT_SHIFT)) | (((u64) totals[j] & 0xFFFFFull) << TJ_META_COUNT_SHIFT);
    m = (m & ~(7ull << TJ_META_FLAG_SHIFT)) | ((u64) (orflags[j] & 7u) << TJ_META_FLAG_SHIFT);
    keys[3 * j + 2] = m;
  }
}


// ---------------------------------------------------------------------------------------------------------------
// host side of the thin layer

struct DevBuf
{
  void *p = nullptr;
  size_t cap = 0;
};

#define TJ_SCAN_PIECE_TARGET (1ull << 30)               // a long stream is scanned in pieces of 1 GiB; up to 1.5 GiB goes in one launch

// everything the host reads back: one device block, one pinned mirror
// (each part on cache lines of its own: the scan's workgroups hammer ctr and cursors with atomics)
// (snap_*: what ctr and cursors held when clear_buckets_kernel emptied them in a finalise -- the host reads the counts of
// the sample from there, with the one copy at the end of the finalise, instead of copying the block before the clear)
struct DevState { alignas (256) DevCounters ctr; alignas (256) FinCounts fin; alignas (256) u32 cursors[TJ_P + 1]; alignas (256) FinPlan plan;
                  alignas (256) DevCounters snap_ctr; alignas (256) u32 snap_cursors[TJ_P + 1]; alignas (256) u32 pad[4]; };

struct tjamd_counter
{
  int device = 0, k = 0, W = 4, n_cu = 256;
  hipStream_t own_stream = nullptr, stream = nullptr;
  DevCounters *d_ctr = nullptr, *h_ctr = nullptr;     // scan counters (device / pinned host mirror)
  DevCounters *d_lctr = nullptr;                      // located-list counters
  u32 *d_cursors = nullptr, *h_cursors = nullptr;     // [TJ_P] records per bucket, then [TJ_P] = next free chunk
  DevBuf pool, table, stage, fix, loc, prefix, rawlist, slow;
  DevBuf log, logmeta;        // the record log of scan_fast_kernel<1, true> and its block counters (2 words: blocks handed out, by launch parity; then a count per block)
  int log_mode = 1;           // widest record (words) that goes through the record log and partition_log_kernel: 1 = k <= 12 (default), 2 = k <= 28 as well
                              // (TATAJUBA_AMD_SINK=log), 0 = none, the scan kernels partition by themselves (TATAJUBA_AMD_SINK=fused)
  bool log_next_clean[2] = {false, false};
  hipEvent_t ev_p1 = nullptr; // after the partition kernel of the last scan call
  // a stream scanned in pieces runs scan, partition, scan, partition, ...: an event in front of every partition kernel but the
  // last and one behind it, so that the scan's and the partition's times can be told apart (TJ_PIECE_EVENTS pieces; beyond: lumped)
#define TJ_PIECE_EVENTS 64
  hipEvent_t ev_pa[TJ_PIECE_EVENTS] = {}, ev_pb[TJ_PIECE_EVENTS] = {};
  int n_piece_ev = 0;
  hipEvent_t ev_m0 = nullptr, ev_m1 = nullptr; bool merge_timed = false;   // around the kernels of the last tjamd_merge_samples
  bool part_timed = false;
  int fast_mode = 1;          // 1: scan_fast_kernel + the generic kernel on what it leaves; 0: generic kernel only; 2: fast kernel leaves everything (tests)
  u32 pool_chunks = 0, maxj = 0;
  int ch_shift = -1;          // chunk = TJ_CH0 << ch_shift records; fixed by the first scan after a reset
  u64 bucket_bound = 0;       // upper bound of the fullest bucket (exact after a synchronisation)
  u64 chunk_bound = 0;        // upper bound of the chunks handed out
  long n_raw_known = 0;       // exact after the last synchronisation
  u64 raw_bound = 0;          // upper bound of the raw records in the buckets (exact after a synchronisation)
  long n_undefined = 0;
  double slack = 1.0;
  DevBuf alt, hist, flags, segid, headpos, keep, outpos, scan_tmp, kept, idx_i, idx_f, cov, bins, binstart, binctx, ovf, grp_jt, grp_hist, fine;
  u32 bin_rank_max = BS_RANK_MAX;
  FinCounts *d_fin = nullptr, *h_fin = nullptr;
  struct DevState *d_state = nullptr, *h_state = nullptr;   // ctr, fin and cursors live in one block: one copy brings all three to the host
  void *h_kept = nullptr; size_t h_kept_cap = 0;   // pinned landing place of tjamd_download_kept (grown, never per call)
  bool bins_zeroed = false;
  bool fine_dirty = false;              // the fine bins hold an aggregation's counts that no clear_buckets_kernel has consumed yet
  bool bins_counted = false;            // clear_buckets_kernel has turned them into the ordering step's bin counts (finalise_binned: no counting pass)
  long n_kept = 0; int n_idx = 0, coverage = 0, status = -1;
  hipEvent_t ev_s0 = nullptr, ev_s1 = nullptr, ev_f0 = nullptr, ev_f1 = nullptr;
  hipEvent_t ev_done = nullptr;                         // tjamd_finalise_begin: the counts have reached the host
  hipEvent_t ev_agg = nullptr;                          // the aggregation and the clearing of the buckets are done (order_stream waits for it)
  hipStream_t order_stream = nullptr;                   // tjamd_counter_set_order_stream: where a finalise begun with tjamd_finalise_begin runs its ordering step
  int fin_pending = 0;                                  // 1: begun, results not looked at yet (tjamd_finalise_end)
  int fin_rb = 0, fin_mc = 0; bool fin_speculative = false, fin_plan_ahead = false; u64 fin_kept_cap = 0;
  hipEvent_t marks[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // tjamd_mark / tjamd_wait_mark
  unsigned mark_seq = 0;
  bool scan_timed = false, fin_timed = false;
  long last_scan_launches = 0;
  unsigned scan_seq = 0;
  size_t piece_target = TJ_SCAN_PIECE_TARGET;           // TATAJUBA_AMD_SCAN_PIECE (bytes) overrides it: tests
  bool buckets_clean = false;
  long plan_mismatches = 0;   // finalises whose in-kernel reading of the kept count differed from the kernel boundary's (expected: 0)
  bool ctr_clean = false;     // the scan counters are zero but for n_undefined (clear_buckets_kernel did it, no scan since)
};

static int ensure (DevBuf &b, size_t bytes, hipStream_t stream, size_t keep_bytes = 0)
{
  if (bytes <= b.cap) return TJAMD_OK;
  size_t want = std::max (bytes, b.cap + b.cap / 2);
  void *np = nullptr;
  hipError_t e = hipMalloc (&np, want);
  if (e != hipSuccess && want > bytes) { want = bytes; e = hipMalloc (&np, want); }
  if (e != hipSuccess) return set_err (TJAMD_ERR_HIP, "hipMalloc of %zu bytes failed: %s", want, hipGetErrorString (e));
  if (b.p) {
    if (keep_bytes) HIPCHK (hipMemcpyAsync (np, b.p, keep_bytes, hipMemcpyDeviceToDevice, stream));
    HIPCHK (hipStreamSynchronize (stream));
    HIPCHK (hipFree (b.p));
  }
  b.p = np; b.cap = want;
  return TJAMD_OK;
}

static void release (DevBuf &b) { if (b.p) (void) hipFree (b.p); b.p = nullptr; b.cap = 0; }

extern "C" tjamd_counter *tjamd_counter_create (int device, int kmer_size)
{
  int n = tjamd_device_count ();
  if (n <= 0) { set_err (TJAMD_ERR_NO_DEVICE, "no HIP device visible: the homopolymer-tract engine needs an MI355X (no CPU fallback)"); return NULL; }
  if (device < 0 || device >= n) { set_err (TJAMD_ERR_ARG, "device %d out of range [0,%d)", device, n); return NULL; }
  if (kmer_size < 2 || kmer_size > 32) { set_err (TJAMD_ERR_ARG, "kmer_size %d outside [2,32] (reference clamp: src/main.c:184-185)", kmer_size); return NULL; }
  HIPCHK_NULL (hipSetDevice (device));
  tjamd_counter *c = new tjamd_counter ();
  c->device = device; c->k = kmer_size;
  c->W = (kmer_size <= 12) ? 1 : (kmer_size <= 28) ? 2 : 4;
  hipDeviceProp_t prop;
  HIPCHK_NULL (hipGetDeviceProperties (&prop, device));
  c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  const char *bm = getenv ("TATAJUBA_AMD_BIN_MAX");      // test hook: a smaller limit forces the radix path
  if (bm && atoi (bm) >= 1 && atoi (bm) <= BS_RANK_MAX) c->bin_rank_max = (u32) atoi (bm);
  const char *pc = getenv ("TATAJUBA_AMD_SCAN_PIECE");
  if (pc && atol (pc) >= 4096) c->piece_target = (size_t) atol (pc);
  const char *sk = getenv ("TATAJUBA_AMD_SINK");          // "fused": scan_fast_kernel<1> partitions its records itself (rounds 1-3); default: record log + partition_log_kernel
  if (sk && !strcmp (sk, "fused")) c->log_mode = 0;
  if (sk && !strcmp (sk, "log")) c->log_mode = 2;
  const char *fm = getenv ("TATAJUBA_AMD_FAST");          // test hook: 0 = generic scan kernel only, 2 = fast kernel hands every tile over
  if (fm && atoi (fm) >= 0 && atoi (fm) <= 2) c->fast_mode = atoi (fm);
  const char *sl = getenv ("TATAJUBA_AMD_BUCKET_SLACK");
  if (sl && atof (sl) >= 1.0) c->slack = atof (sl);
  HIPCHK_NULL (hipStreamCreateWithFlags (&c->own_stream, hipStreamNonBlocking));
  c->stream = c->own_stream;
  HIPCHK_NULL (hipMalloc ((void **) &c->d_state, sizeof (DevState)));
  HIPCHK_NULL (hipMalloc ((void **) &c->d_lctr, sizeof (DevCounters)));
  HIPCHK_NULL (hipHostMalloc ((void **) &c->h_state, sizeof (DevState), hipHostMallocDefault));
  c->d_ctr = &c->d_state->ctr; c->d_fin = &c->d_state->fin;
  HIPCHK_NULL (hipMalloc ((void **) &c->d_cursors, (size_t) (TJ_P + 1) * TJ_CSTRIDE * 4));     // (strided: see TJ_CSTRIDE; d_state->cursors is the packed copy the host reads)
  HIPCHK_NULL (hipMemsetAsync (c->d_cursors, 0, (size_t) (TJ_P + 1) * TJ_CSTRIDE * 4, c->stream));
  c->h_ctr = &c->h_state->ctr; c->h_fin = &c->h_state->fin; c->h_cursors = c->h_state->cursors;
  memset (c->h_state, 0, sizeof (DevState));
  HIPCHK_NULL (hipMemsetAsync (c->d_state, 0, sizeof (DevState), c->stream));
  HIPCHK_NULL (hipMemsetAsync (c->d_lctr, 0, sizeof (DevCounters), c->stream));
  HIPCHK_NULL (hipEventCreate (&c->ev_s0)); HIPCHK_NULL (hipEventCreate (&c->ev_s1));
  HIPCHK_NULL (hipEventCreate (&c->ev_f0)); HIPCHK_NULL (hipEventCreate (&c->ev_f1));
  HIPCHK_NULL (hipEventCreate (&c->ev_p1));
  HIPCHK_NULL (hipEventCreate (&c->ev_m0)); HIPCHK_NULL (hipEventCreate (&c->ev_m1));
  HIPCHK_NULL (hipEventCreateWithFlags (&c->ev_done, hipEventDisableTiming));
  HIPCHK_NULL (hipEventCreateWithFlags (&c->ev_agg, hipEventDisableTiming));
  HIPCHK_NULL (hipStreamSynchronize (c->stream));
  return c;
}

extern "C" void tjamd_counter_destroy (tjamd_counter *c)
{
  if (!c) return;
  (void) hipSetDevice (c->device);
  (void) hipStreamSynchronize (c->stream);
  DevBuf *all[] = {&c->log, &c->logmeta, &c->pool, &c->table, &c->stage, &c->fix, &c->loc, &c->prefix, &c->rawlist, &c->slow, &c->alt, &c->hist, &c->flags, &c->segid, &c->headpos,
                   &c->keep, &c->outpos, &c->scan_tmp, &c->kept, &c->idx_i, &c->idx_f, &c->cov, &c->bins, &c->binstart, &c->binctx, &c->ovf, &c->grp_jt, &c->grp_hist, &c->fine};
  for (DevBuf *b : all) release (*b);
  for (hipEvent_t ev : c->marks) if (ev) (void) hipEventDestroy (ev);
  if (c->d_state) (void) hipFree (c->d_state);
  if (c->d_lctr) (void) hipFree (c->d_lctr);
  if (c->d_cursors) (void) hipFree (c->d_cursors);
  if (c->h_state) (void) hipHostFree (c->h_state);
  if (c->h_kept) (void) hipHostFree (c->h_kept);
  if (c->ev_s0) (void) hipEventDestroy (c->ev_s0);
  if (c->ev_s1) (void) hipEventDestroy (c->ev_s1);
  if (c->ev_p1) (void) hipEventDestroy (c->ev_p1);
  for (int i = 0; i < TJ_PIECE_EVENTS; i++) { if (c->ev_pa[i]) (void) hipEventDestroy (c->ev_pa[i]); if (c->ev_pb[i]) (void) hipEventDestroy (c->ev_pb[i]); }
  if (c->ev_m0) (void) hipEventDestroy (c->ev_m0);
  if (c->ev_m1) (void) hipEventDestroy (c->ev_m1);
  if (c->ev_f0) (void) hipEventDestroy (c->ev_f0);
  if (c->ev_f1) (void) hipEventDestroy (c->ev_f1);
  if (c->ev_done) (void) hipEventDestroy (c->ev_done);
  if (c->ev_agg) (void) hipEventDestroy (c->ev_agg);
  if (c->own_stream) (void) hipStreamDestroy (c->own_stream);
  delete c;
}

extern "C" int tjamd_counter_device (const tjamd_counter *c) { return c ? c->device : -1; }

extern "C" int tjamd_counter_set_stream (tjamd_counter *c, void *hip_stream)
{
  if (!c) return set_err (TJAMD_ERR_ARG, "null counter");
  HIPCHK (hipSetDevice (c->device));
  HIPCHK (hipStreamSynchronize (c->stream));
  c->stream = hip_stream ? (hipStream_t) hip_stream : c->own_stream;
  return TJAMD_OK;
}

// A second stream for the ordering step of a finalise that was begun with tjamd_finalise_begin (null: none).  The step is
// six small launches whose time is latency, not work (110 us for 234 k records): behind an event on the counter's stream
// it runs beside whatever that stream does next -- the scan of the next sample on another counter -- and
// tjamd_finalise_end waits for its last copy as before.  The counter itself must not be touched between the two calls.
extern "C" int tjamd_counter_set_order_stream (tjamd_counter *c, void *hip_stream)
{
  if (!c) return set_err (TJAMD_ERR_ARG, "null counter");
  HIPCHK (hipSetDevice (c->device));
  HIPCHK (hipStreamSynchronize (c->stream));
  if (c->order_stream) HIPCHK (hipStreamSynchronize (c->order_stream));
  c->order_stream = (hipStream_t) hip_stream;
  return TJAMD_OK;
}

static Buckets make_buckets (const tjamd_counter *c)
{
  Buckets B;
  B.pool = (u64 *) c->pool.p; B.table = (u32 *) c->table.p; B.cursors = c->d_cursors; B.pool_next = c->d_cursors + TJ_P * TJ_CSTRIDE;
  B.pool_chunks = c->pool_chunks; B.maxj = c->maxj; B.ch_shift = (u32) std::max (c->ch_shift, 0);
  return B;
}

// forget every raw record: cursors and chunk counter to zero, chunk table to "unclaimed"
static int clear_buckets (tjamd_counter *c, bool snapshot = false)
{
  if (!c->buckets_clean || snapshot) {
    // (behind an aggregation that counted its kept records into the fine bins: this kernel turns them into the ordering
    // step's bin counts, see there)
    const bool fused = snapshot && c->fine_dirty && c->bins.p != nullptr;
    hipLaunchKernelGGL (clear_buckets_kernel, dim3 (TJ_P), dim3 (256), 0, c->stream, c->d_cursors, c->d_ctr,
                        (u32 *) (c->maxj ? c->table.p : nullptr), c->maxj, c->ch_shift, c->d_fin, (u32 *) c->bins.p, c->bins.p ? BS_MAXBINS : 0,
                        snapshot ? &c->d_state->snap_ctr : (DevCounters *) nullptr, snapshot ? c->d_state->snap_cursors : (u32 *) nullptr,
                        snapshot ? &c->d_state->plan : (FinPlan *) nullptr, fused ? (u32 *) c->fine.p : (u32 *) nullptr, fine_bin_bits (c->k));
    HIPCHK (hipGetLastError ());
    c->buckets_clean = true;
    c->ctr_clean = true;
    c->bins_zeroed = c->bins.p != nullptr && !fused;
    if (fused) { c->fine_dirty = false; c->bins_counted = true; }   // (the fine bins are zeroed again, the step's bins hold its counts if the plan is usable)
  }
  c->n_raw_known = 0; c->raw_bound = 0; c->bucket_bound = 0; c->chunk_bound = TJ_P; c->ch_shift = -1;
  return TJAMD_OK;
}

extern "C" int tjamd_counter_reset (tjamd_counter *c)
{
  if (!c) return set_err (TJAMD_ERR_ARG, "null counter");
  HIPCHK (hipSetDevice (c->device));
  // (the memset is not needed right after a finalise whose counts the host has seen: clear_buckets_kernel zeroed the rest)
  if (!(c->ctr_clean && c->fin_pending == 0 && c->n_undefined == 0)) HIPCHK (hipMemsetAsync (c->d_ctr, 0, sizeof (DevCounters), c->stream));
  int rc = clear_buckets (c);
  c->ctr_clean = true;
  if (rc) return rc;
  c->n_undefined = 0;
  c->n_kept = 0; c->n_idx = 0; c->coverage = 0; c->status = -1;
  return TJAMD_OK;
}

extern "C" void *tjamd_host_alloc (size_t bytes)
{
  void *p = nullptr;
  if (hipHostMalloc (&p, bytes, hipHostMallocPortable) != hipSuccess) { set_err (TJAMD_ERR_HIP, "hipHostMalloc of %zu bytes failed", bytes); return NULL; }   // (portable: batch buffers are pooled across the counters of all devices)
  return p;
}

extern "C" void tjamd_host_free (void *p) { if (p) (void) hipHostFree (p); }

// device memory for callers that have no HIP headers of their own (outputs of tjamd_merge_samples and the like)
extern "C" void *tjamd_device_alloc (tjamd_counter *c, size_t bytes)
{
  void *p = nullptr;
  if (!c || hipSetDevice (c->device) != hipSuccess || hipMalloc (&p, bytes ? bytes : 1) != hipSuccess) { set_err (TJAMD_ERR_HIP, "hipMalloc of %zu bytes failed", bytes); return NULL; }
  return p;
}
extern "C" void tjamd_device_free (tjamd_counter *c, void *p) { if (c && p && hipSetDevice (c->device) == hipSuccess) (void) hipFree (p); }
extern "C" int tjamd_device_download (tjamd_counter *c, void *host, const void *dev, size_t bytes)
{
  if (!c || (bytes && (!host || !dev))) return set_err (TJAMD_ERR_ARG, "bad arguments");
  HIPCHK (hipSetDevice (c->device));
  HIPCHK (hipStreamSynchronize (c->stream));
  if (bytes) HIPCHK (hipMemcpy (host, dev, bytes, hipMemcpyDeviceToHost));
  return TJAMD_OK;
}

extern "C" int tjamd_sync (tjamd_counter *c)
{
  if (!c) return set_err (TJAMD_ERR_ARG, "null counter");
  HIPCHK (hipSetDevice (c->device));
  HIPCHK (hipStreamSynchronize (c->stream));
  return TJAMD_OK;
}

// A mark is a point in the counter's stream (everything queued before it); waiting for a mark does not wait for what was
// queued after it.  The feeder uses marks to re-use a pinned batch buffer as soon as ITS copy and scan are done while
// later batches are still in flight.  Eight marks are live at a time (the ninth re-uses the first one's slot).
extern "C" int tjamd_mark (tjamd_counter *c)
{
  if (!c) return -set_err (TJAMD_ERR_ARG, "null counter");
  if (hipSetDevice (c->device) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "hipSetDevice failed");
  const int slot = (int) (c->mark_seq++ & 7u);
  if (!c->marks[slot] && hipEventCreateWithFlags (&c->marks[slot], hipEventDisableTiming) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "hipEventCreate failed");
  if (hipEventRecord (c->marks[slot], c->stream) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "hipEventRecord failed");
  return slot;
}

extern "C" int tjamd_wait_mark (tjamd_counter *c, int mark)
{
  if (!c || mark < 0 || mark > 7 || !c->marks[mark]) return set_err (TJAMD_ERR_ARG, "bad mark");
  HIPCHK (hipSetDevice (c->device));
  HIPCHK (hipEventSynchronize (c->marks[mark]));
  return TJAMD_OK;
}

// stream synchronisation + exact counts
__global__ void pack_cursors_kernel (const u32 *__restrict__ cursors, u32 *__restrict__ packed)
{
  if (threadIdx.x <= TJ_P) packed[threadIdx.x] = cursors[threadIdx.x * TJ_CSTRIDE];
}

static int queue_counter_copies (tjamd_counter *c)
{
  hipLaunchKernelGGL (pack_cursors_kernel, dim3 (1), dim3 (320), 0, c->stream, (const u32 *) c->d_cursors, c->d_state->cursors);
  HIPCHK (hipGetLastError ());
  HIPCHK (hipMemcpyAsync (c->h_state, c->d_state, sizeof (DevState), hipMemcpyDeviceToHost, c->stream));
  return TJAMD_OK;
}

// the copies queued by queue_counter_copies() have arrived: errors, exact counts
static int apply_counter_copies (tjamd_counter *c, bool from_snapshot = false)
{
  long total = 0;
  u64 mx = 0;
  struct View { const DevCounters *h_ctr; const u32 *h_cursors; } v = {c->h_ctr, c->h_cursors};
  if (from_snapshot) { v.h_ctr = &c->h_state->snap_ctr; v.h_cursors = c->h_state->snap_cursors; }
  for (int b = 0; b < TJ_P; b++) { total += v.h_cursors[b]; mx = std::max<u64> (mx, v.h_cursors[b]); }
  if (v.h_ctr->overflow == 2u)
    return set_err (TJAMD_ERR_HIP, "a chunk of raw record storage was claimed but its id was never published (waited %u polls): "
                    "the claiming workgroup did not make progress -- please report; scanning again usually succeeds", (unsigned) TJ_SPIN_MAX);
  if (v.h_ctr->overflow)
    return set_err (TJAMD_ERR_CAPACITY, "raw record storage exhausted (%u of %u chunks handed out, fullest bucket %llu records): "
                    "raise TATAJUBA_AMD_BUCKET_SLACK (now %.1f) and scan again", v.h_cursors[TJ_P], c->pool_chunks, (unsigned long long) mx, c->slack);
  if (v.h_ctr->fix_overflow) return set_err (TJAMD_ERR_CAPACITY, "too many non-ACGTU tract candidates in one batch (%llu)",
                                             (unsigned long long) std::max (v.h_ctr->lc[0].n_fix, v.h_ctr->lc[1].n_fix));
  c->n_raw_known = total - (long) v.h_ctr->n_null;
  c->raw_bound = (u64) c->n_raw_known;
  c->bucket_bound = mx;
  c->chunk_bound = v.h_cursors[TJ_P];
  c->n_undefined = (long) v.h_ctr->n_undefined;
  return TJAMD_OK;
}

static int sync_counters (tjamd_counter *c)
{
  int rc = queue_counter_copies (c);
  if (rc) return rc;
  HIPCHK (hipStreamSynchronize (c->stream));
  return apply_counter_copies (c);
}

// room for `add` more raw records (an upper bound), wherever the hash sends them, written by `grid` workgroups
static void choose_chunk_size (tjamd_counter *c, u64 records)
{ // fixed by the first scan (or reservation) after a reset: at most ~16 k chunks for this many records
  if (c->ch_shift >= 0) return;
  const u64 units = records / (16384ull * TJ_CH0);
  int sft = 3;                                          // a chunk holds more than any one reservation: a partition pass stages 4096 records, partition_log_kernel's TJ_LOGB = 8192
  static_assert ((TJ_CH0 << 3) > (int) TJ_LOGB, "a run of one log block crosses at most one chunk boundary");
  while ((1ull << sft) < units) sft++;
  c->ch_shift = sft;
}

// chunk table with rows of at least need_maxj entries (rows keep their contents)
static int ensure_table (tjamd_counter *c, u64 need_maxj)
{
  if (need_maxj > c->maxj) {
    const u32 nmaxj = (u32) std::max<u64> (need_maxj, (u64) c->maxj + c->maxj / 2);
    void *np = nullptr;
    hipError_t e = hipMalloc (&np, (size_t) TJ_P * nmaxj * 4);
    if (e != hipSuccess) return set_err (TJAMD_ERR_HIP, "hipMalloc of the chunk table failed: %s", hipGetErrorString (e));
    HIPCHK (hipMemsetAsync (np, 0xFF, (size_t) TJ_P * nmaxj * 4, c->stream));
    if (!c->table.p || !c->maxj) {
      hipLaunchKernelGGL (init_table_kernel, dim3 (1), dim3 (TJ_P), 0, c->stream, (u32 *) np, nmaxj, c->d_cursors + TJ_P * TJ_CSTRIDE);
      HIPCHK (hipGetLastError ());
    }
    if (c->table.p) {
      if (c->maxj) {
        hipLaunchKernelGGL (table_relayout_kernel, dim3 (TJ_P), dim3 (256), 0, c->stream, (const u32 *) c->table.p, c->maxj, (u32 *) np, nmaxj);
        HIPCHK (hipGetLastError ());
      }
      HIPCHK (hipStreamSynchronize (c->stream));
      HIPCHK (hipFree (c->table.p));
    }
    c->table.p = np; c->table.cap = (size_t) TJ_P * nmaxj * 4; c->maxj = nmaxj;
  }
  return TJAMD_OK;
}

static int ensure_buckets (tjamd_counter *c, u64 add, int grid)
{
  choose_chunk_size (c, add);
  const u64 ch = (u64) TJ_CH0 << c->ch_shift;
  const u64 pad_bucket = 0;                             // (no padding records any more)
  (void) grid;
  // one chunk per bucket is always claimed ahead of the cursor
  const u64 need_chunks = c->chunk_bound + (add + ch - 1) / ch + 2 * TJ_P;
  c->buckets_clean = false; c->ctr_clean = false;
  const u64 need_maxj = (c->bucket_bound + add + ch - 1) / ch + 3;      // worst case: everything in one bucket
  if (need_chunks >= TJ_NOCHUNK || need_maxj >= (1ull << 31)) return set_err (TJAMD_ERR_CAPACITY, "batch too large for the chunk table");
  int rc = ensure (c->pool, (size_t) need_chunks * ch * c->W * 8, c->stream, std::min<size_t> ((size_t) (c->chunk_bound * ch * c->W * 8), c->pool.cap));
  if (rc) return rc;
  c->pool_chunks = (u32) std::min<u64> (c->pool.cap / (ch * c->W * 8), TJ_NOCHUNK - 1);
  rc = ensure_table (c, need_maxj);
  if (rc) return rc;
  c->chunk_bound = need_chunks;
  c->bucket_bound += add + pad_bucket;
  return TJAMD_OK;
}

#define TJ_FIX_CAP (1u << 20)

static int check_scan_args (tjamd_counter *c, int min_tract_size)
{
  if (!c) return set_err (TJAMD_ERR_ARG, "null counter");
  if (min_tract_size < 1 || min_tract_size > 32) return set_err (TJAMD_ERR_ARG, "min_tract_size %d outside [1,32] (reference clamp: src/main.c:186-187)", min_tract_size);
  return TJAMD_OK;
}

// Room for the raw records of about stream_bytes of reads, allocated in one piece instead of by repeated growth (each
// growth is an allocation, a device copy and a synchronisation).  A hint: scans allocate what they need anyway.
extern "C" int tjamd_reserve (tjamd_counter *c, size_t stream_bytes, int min_tract_size)
{
  int rc = check_scan_args (c, min_tract_size);
  if (rc) return rc;
  HIPCHK (hipSetDevice (c->device));
  const u64 records = std::min<u64> ((u64) (stream_bytes / (size_t) std::max (min_tract_size, 2) + 1), ((u64) 4 << 30) / ((u64) c->W * 8));   // (a hint is not worth more than 4 GB)
  choose_chunk_size (c, records);
  const u64 ch = (u64) TJ_CH0 << c->ch_shift;
  const size_t keep = std::min<size_t> (c->pool.cap, (size_t) (c->chunk_bound * ch * c->W * 8));
  rc = ensure (c->pool, (size_t) ((records / ch + 2 * TJ_P + 1) * ch * c->W * 8), c->stream, keep);
  if (rc) return rc;
  c->pool_chunks = (u32) std::min<u64> (c->pool.cap / (ch * c->W * 8), TJ_NOCHUNK - 1);
  c->buckets_clean = c->buckets_clean && c->table.p != nullptr;
  return ensure_table (c, (c->bucket_bound + records + ch - 1) / ch + 3);
}

// one launch of the scan over [d_stream, d_stream + n_bytes) (16-byte aligned, starts and ends on read boundaries)
static int scan_device_piece (tjamd_counter *c, const void *d_stream, size_t n_bytes, int min_tract_size, bool first, bool last)
{
  int rc = TJAMD_OK;
  if (n_bytes < 64 && d_stream != c->stage.p) {             // the kernels read whole 16-byte chunks: give tiny streams room
    rc = ensure (c->stage, 256, c->stream);
    if (rc) return rc;
    HIPCHK (hipMemcpyAsync (c->stage.p, d_stream, n_bytes, hipMemcpyDeviceToDevice, c->stream));
    d_stream = c->stage.p;
  }
  const int mprime = std::max (min_tract_size, 2);          // a tract needs two equal bytes: m = 1 behaves as m = 2
  // tracts are disjoint runs of >= m' bytes: at most n/m' records come out of this batch
  const u64 bound = (u64) ((double) (n_bytes / (size_t) mprime + 1) * c->slack);
  const long n_tiles = (long) ((n_bytes + TJ_SB_TILE - 1) / TJ_SB_TILE);
  const int grid = (int) std::min<long> (n_tiles, (long) c->n_cu * TJ_SB_WG_PER_CU);
  rc = ensure_buckets (c, bound, grid + 1);             // + the fix-up kernel's block-per-record inserts
  c->raw_bound += (u64) (n_bytes / (size_t) mprime) + 1u;
  if (!rc) rc = ensure (c->fix, (size_t) TJ_FIX_CAP * sizeof (FixEntry), c->stream);
  // fast kernel: tiles of FK_OWN tract starts, two workgroups resident per CU; the list of the tiles it leaves to the generic
  // kernel (every one of them at worst), which that kernel covers with ceil (FK_OWN / TJ_SB_TILE) of its own tiles each
  const long n_ftiles = (long) ((n_bytes + FK_OWN - 1) / FK_OWN);
  static const int fwg = getenv ("TATAJUBA_AMD_FGRID") ? atoi (getenv ("TATAJUBA_AMD_FGRID")) : FK_WG_PER_CU;     // (experiment hook: workgroups per CU)
  const int fgrid = (int) std::min<long> (n_ftiles, (long) c->n_cu * fwg);
  const int lgrid = (int) std::min<long> (n_ftiles * ((FK_OWN + TJ_SB_TILE - 1) / TJ_SB_TILE), (long) c->n_cu * TJ_SB_WG_PER_CU);
  if (!rc && c->fast_mode) rc = ensure (c->slow, (size_t) n_ftiles * 4 + 64, c->stream);
  if (rc) return rc;
  const uint8_t *seq = (const uint8_t *) d_stream;
  const Buckets BK = make_buckets (c);
  FixEntry *fix = (FixEntry *) c->fix.p;
  const int par = (int) (c->scan_seq++ & 1u);            // per-launch counters are double-buffered (DevCounters::lc)
  if (first) HIPCHK (hipEventRecord (c->ev_s0, c->stream));
  const TileSrc plain = {nullptr, 0};
  // (the fast kernel counts tiles and bytes in 32 bits: a piece that could not be cut below 2 GiB -- one read longer
  // than that -- is left to the general kernel)
  const bool use_fast = c->fast_mode && n_bytes < ((size_t) 1 << 31) - (1u << 20);
  // one-word records (k <= 12): the fast kernel appends to a log sized for the worst case (a tract every m' bytes; every
  // workgroup leaves at most two blocks partly filled), partition_log_kernel distributes it over the buckets
  const bool use_log = use_fast && c->W <= c->log_mode;
  LogSpace LG = {nullptr, nullptr, nullptr, 0u};
  const int fgrid_log = (int) std::min<long> (n_ftiles, (long) c->n_cu * FK_LOG_WG_PER_CU);
  if (use_log) {
    // (a workgroup with n records has taken at most n / TJ_LOGB + 3 blocks: the one being filled and up to two ahead of it)
    const u64 n_blocks = ((bound * (u64) c->W) >> TJ_LOGB_SHIFT) + 3ull * (u64) fgrid_log + 8ull;      // (a block is TJ_LOGB words: TJ_LOGB / W records)
    rc = ensure (c->log, (size_t) (((n_blocks << TJ_LOGB_SHIFT) + 64ull * (u64) fgrid_log) * 8ull), c->stream);
    if (!rc && (size_t) (n_blocks + 2) * 4 > c->logmeta.cap) {
      rc = ensure (c->logmeta, (size_t) (n_blocks + 2) * 4 * 2, c->stream);
      c->log_next_clean[0] = c->log_next_clean[1] = false;
    }
    if (rc) return rc;
    u32 *meta = (u32 *) c->logmeta.p;
    if (!c->log_next_clean[par]) HIPCHK (hipMemsetAsync (meta + par, 0, 4, c->stream));
    LG.log = (u64 *) c->log.p; LG.next = meta + par; LG.next_other = meta + (par ^ 1); LG.count = meta + 2; LG.n_blocks = (u32) n_blocks;
    c->log_next_clean[par] = false; c->log_next_clean[par ^ 1] = true;
  }
  else c->log_next_clean[par ^ 1] = c->log_next_clean[par ^ 1] && true;
#define TJ_LAUNCH_SCAN(WW) do { \
    if (use_fast) { \
      /* the fast kernel takes every tile it can vouch for and lists the others; the generic kernel then works through the list */ \
      if (WW <= 2 && use_log) \
        hipLaunchKernelGGL ((scan_fast_kernel<(WW <= 2 ? WW : 1), true>), dim3 (fgrid_log), dim3 (FK_BLOCK), 0, c->stream, seq, (long) n_bytes, n_ftiles, c->k, mprime, BK, c->d_ctr, \
                            (u32 *) c->slow.p, par, c->fast_mode == 2 ? 1 : 0, LG); \
      else \
      hipLaunchKernelGGL (scan_fast_kernel<WW>, dim3 (fgrid), dim3 (FK_BLOCK), 0, c->stream, seq, (long) n_bytes, n_ftiles, c->k, mprime, BK, c->d_ctr, \
                          (u32 *) c->slow.p, par, c->fast_mode == 2 ? 1 : 0, LG); \
      const TileSrc listed = {(const u32 *) c->slow.p, (long) FK_OWN}; \
      hipLaunchKernelGGL (scan_bins_kernel<WW>, dim3 (lgrid), dim3 (TJ_SB_BLOCK), 0, c->stream, seq, (long) n_bytes, 0l, c->k, mprime, BK, c->d_ctr, fix, (u32) TJ_FIX_CAP, par, listed); \
    } \
    else { \
      hipLaunchKernelGGL (scan_bins_kernel<WW>, dim3 (grid), dim3 (TJ_SB_BLOCK), 0, c->stream, seq, (long) n_bytes, n_tiles, c->k, mprime, BK, c->d_ctr, fix, (u32) TJ_FIX_CAP, par, plain); \
      hipLaunchKernelGGL (nrun_fixup_bins_kernel<WW>, dim3 (64), dim3 (256), 0, c->stream, seq, (long) n_bytes, c->k, mprime, BK, c->d_ctr, (const FixEntry *) fix, (u32) TJ_FIX_CAP, par); \
    } \
  } while (0)
  switch (c->W) {
    case 1: TJ_LAUNCH_SCAN (1); break;
    case 2: TJ_LAUNCH_SCAN (2); break;
    default: TJ_LAUNCH_SCAN (4); break;
  }
#undef TJ_LAUNCH_SCAN
  HIPCHK (hipGetLastError ());
  if (last) HIPCHK (hipEventRecord (c->ev_s1, c->stream));
  if (first) c->n_piece_ev = 0;
  const bool mid_events = use_log && !last && c->n_piece_ev < TJ_PIECE_EVENTS;
  if (mid_events) {
    const int i = c->n_piece_ev;
    if (!c->ev_pa[i]) { HIPCHK (hipEventCreate (&c->ev_pa[i])); HIPCHK (hipEventCreate (&c->ev_pb[i])); }
    HIPCHK (hipEventRecord (c->ev_pa[i], c->stream));
  }
  if (use_log) {
    if (c->W == 1) hipLaunchKernelGGL (partition_log_kernel<1>, dim3 ((unsigned) (c->n_cu * PL_WG_PER_CU)), dim3 (PL_BLOCK), 0, c->stream, LG, BK, c->d_ctr, c->k);
    else hipLaunchKernelGGL (partition_log_kernel<2>, dim3 ((unsigned) (c->n_cu * PL_WG_PER_CU)), dim3 (PL_BLOCK), 0, c->stream, LG, BK, c->d_ctr, c->k);
    HIPCHK (hipGetLastError ());
  }
  if (mid_events) { HIPCHK (hipEventRecord (c->ev_pb[c->n_piece_ev], c->stream)); c->n_piece_ev++; }
  if (last) { HIPCHK (hipEventRecord (c->ev_p1, c->stream)); c->part_timed = use_log; }
  c->scan_timed = true;
  c->last_scan_launches = first ? 1 : c->last_scan_launches + 1;
  c->status = -1;
  return TJAMD_OK;
}

// A stream is scanned in one launch up to 1.5 x TJ_SCAN_PIECE_TARGET bytes.  A longer one goes piece by piece, each piece cut
// after a read delimiter that sits on the last byte of a 16-byte line (so that the next piece starts aligned; with reads
// of any length one turns up within a few reads), and the exact record counts are fetched between pieces: the device
// storage then follows what the reads really contain instead of the worst case of the whole stream (n / m' records).

__global__ __launch_bounds__ (1024)
void find_cut_kernel (const uint8_t *__restrict__ seq, unsigned long long first_from, unsigned long long step, unsigned long long n, unsigned long long *out)
{ // out[b] = smallest p >= first_from + b * step with p % 16 == 15 and seq[p] == '\n', or ~0 (one workgroup per cut)
  __shared__ unsigned long long best;
  const unsigned long long from = first_from + (unsigned long long) blockIdx.x * step;
  if (threadIdx.x == 0) best = ~0ull;
  __syncthreads ();
  for (unsigned long long base = from & ~15ull; base < n; base += 1024ull * 16ull) {
    const unsigned long long p = base + 16ull * threadIdx.x + 15ull;
    if (p >= from && p < n && seq[p] == (uint8_t) '\n') atomicMin (&best, p);
    __syncthreads ();
    if (best != ~0ull) break;
    __syncthreads ();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = best;
}

extern "C" int tjamd_scan_device (tjamd_counter *c, const void *d_stream, size_t n_bytes, int min_tract_size)
{
  int rc = check_scan_args (c, min_tract_size);
  if (rc) return rc;
  if (n_bytes == 0) return TJAMD_OK;
  if (!d_stream || ((uintptr_t) d_stream & 15u)) return set_err (TJAMD_ERR_ARG, "device stream pointer must be non-null and 16-byte aligned");
  HIPCHK (hipSetDevice (c->device));
  const size_t piece_target = c->piece_target, piece_max = piece_target + piece_target / 2;
  if (n_bytes <= piece_max) return scan_device_piece (c, d_stream, n_bytes, min_tract_size, true, true);
  const uint8_t *seq = (const uint8_t *) d_stream;
  // every piece is cut near a multiple of the target: the cuts do not depend on each other, so they are all looked for
  // by one launch and fetched with one copy (per piece that would be a kernel, a copy and a synchronisation of 25 us)
  const size_t n_cuts = (n_bytes - 1) / piece_target;    // candidates at piece_target, 2 piece_target, ...
  std::vector<unsigned long long> cuts (n_cuts);
  rc = ensure (c->prefix, 64 + 8 * n_cuts, c->stream);
  if (rc) return rc;
  hipLaunchKernelGGL (find_cut_kernel, dim3 ((unsigned) n_cuts), dim3 (1024), 0, c->stream, seq, (unsigned long long) piece_target, (unsigned long long) piece_target,
                      (unsigned long long) n_bytes, (unsigned long long *) c->prefix.p);
  HIPCHK (hipMemcpyAsync (cuts.data (), c->prefix.p, 8 * n_cuts, hipMemcpyDeviceToHost, c->stream));
  HIPCHK (hipStreamSynchronize (c->stream));
  size_t off = 0, next_cut = 0;
  bool first = true;
  while (off < n_bytes) {
    size_t end = n_bytes;
    if (n_bytes - off > piece_max) {
      // the next cut that makes a piece of at least half the target (cuts come in increasing order, about a target apart)
      while (next_cut < n_cuts && (cuts[next_cut] == ~0ull || cuts[next_cut] + 1 < off + piece_target / 2)) next_cut++;
      if (next_cut < n_cuts && cuts[next_cut] + 1 < n_bytes) end = (size_t) cuts[next_cut] + 1;
    }
    rc = scan_device_piece (c, seq + off, end - off, min_tract_size, first, end == n_bytes);
    if (rc) return rc;
    if (end < n_bytes && (rc = sync_counters (c))) return rc;     // exact counts: the next piece adds its own worst case to them
    off = end;
    first = false;
  }
  return TJAMD_OK;
}

extern "C" int tjamd_scan_host (tjamd_counter *c, const void *h_stream, size_t n_bytes, int min_tract_size)
{
  if (!c) return set_err (TJAMD_ERR_ARG, "null counter");
  if (n_bytes == 0) return TJAMD_OK;
  if (!h_stream) return set_err (TJAMD_ERR_ARG, "null host stream");
  HIPCHK (hipSetDevice (c->device));
  int rc = ensure (c->stage, (n_bytes + 255) & ~(size_t) 255, c->stream);
  if (rc) return rc;
  HIPCHK (hipMemcpyAsync (c->stage.p, h_stream, n_bytes, hipMemcpyHostToDevice, c->stream));
  return tjamd_scan_device (c, c->stage.p, n_bytes, min_tract_size);
}

extern "C" long tjamd_scan_host_located (tjamd_counter *c, const void *h_stream, size_t n_bytes, int min_tract_size,
                                          tjamd_located_record *out, long capacity)
{ // min_tract_size == 0: every isolated base instead of tracts (reference: update_hopo_counter_from_seq_all_monomers)
  int rc = (min_tract_size == 0 && c) ? TJAMD_OK : check_scan_args (c, min_tract_size);
  if (rc) return -rc;
  if (n_bytes == 0) return 0;
  if (hipSetDevice (c->device) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "hipSetDevice failed");
  const int mprime = min_tract_size == 0 ? 0 : std::max (min_tract_size, 2);
  const long bound = (long) (n_bytes / (size_t) (mprime ? mprime : 1)) + 1;
  rc = ensure (c->stage, (n_bytes + 255) & ~(size_t) 255, c->stream);
  if (!rc) rc = ensure (c->loc, (size_t) bound * 32, c->stream);
  if (!rc) rc = ensure (c->fix, (size_t) TJ_FIX_CAP * sizeof (FixEntry), c->stream);
  if (rc) return -rc;
  if (hipMemcpyAsync (c->stage.p, h_stream, n_bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
      hipMemsetAsync (c->d_lctr, 0, sizeof (DevCounters), c->stream) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "copy to device failed");
  const long n_tiles = (long) ((n_bytes + 4095) / 4096);
  hipLaunchKernelGGL (scan_list_kernel, dim3 ((unsigned) std::min<long> (n_tiles, 2048)), dim3 (256), 0, c->stream, (const uint8_t *) c->stage.p, (long) n_bytes,
                      n_tiles, c->k, mprime, (u64 *) c->loc.p, (u64) bound, c->d_lctr, (FixEntry *) c->fix.p, (u32) TJ_FIX_CAP, 0);
  hipLaunchKernelGGL (nrun_fixup_list_kernel, dim3 (64), dim3 (256), 0, c->stream, (const uint8_t *) c->stage.p, (long) n_bytes, c->k, mprime,
                      (u64 *) c->loc.p, (u64) bound, c->d_lctr, (const FixEntry *) c->fix.p, (u32) TJ_FIX_CAP, 0);
  if (hipGetLastError () != hipSuccess) return -set_err (TJAMD_ERR_HIP, "located scan launch failed");
  if (hipMemcpyAsync (c->h_ctr, c->d_lctr, sizeof (DevCounters), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
      hipStreamSynchronize (c->stream) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "located scan failed: %s", hipGetErrorString (hipGetLastError ()));
  const DevCounters h = *c->h_ctr;
  if (h.overflow || h.fix_overflow) return -set_err (TJAMD_ERR_CAPACITY, "located scan overflow");
  c->n_undefined += (long) h.n_undefined;
  const long n = (long) h.n_rec;
  if (n > capacity) return -set_err (TJAMD_ERR_CAPACITY, "located scan produced %ld records, caller capacity %ld", n, capacity);
  if (n) {
    if (hipMemcpy (out, c->loc.p, (size_t) n * 32, hipMemcpyDeviceToHost) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "download failed");
    std::sort (out, out + n, [] (const tjamd_located_record &a, const tjamd_located_record &b) { return a.pos < b.pos; });
  }
  return n;
}

extern "C" long tjamd_raw_count (tjamd_counter *c)
{
  if (!c) return -set_err (TJAMD_ERR_ARG, "null counter");
  if (hipSetDevice (c->device) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "hipSetDevice failed");
  int rc = sync_counters (c);
  return rc ? -rc : c->n_raw_known;
}

// diagnostic: records per hash bucket after a synchronisation (not part of the public header)
extern "C" long tjamd_debug_bucket_counts (tjamd_counter *c, unsigned *out, int n)
{
  if (tjamd_raw_count (c) < 0) return -1;
  for (int b = 0; b < n && b < TJ_P; b++) out[b] = c->h_cursors[b];
  return TJ_P;
}

// diagnostic (tests, tools): tiles the fast kernel of the last scan launch handed over to the generic kernel
extern "C" long tjamd_debug_slow_tiles (tjamd_counter *c)
{
  if (tjamd_raw_count (c) < 0) return -1;
  return (long) c->h_ctr->lc[(c->scan_seq - 1u) & 1u].n_slow;
}

extern "C" long tjamd_undefined_runs (tjamd_counter *c)
{
  if (tjamd_raw_count (c) < 0) return -1;
  return c->n_undefined;
}

extern "C" long tjamd_download_raw (tjamd_counter *c, tjamd_record *out, long capacity)
{
  long n = tjamd_raw_count (c);
  if (n < 0) return n;
  if (n > capacity) return -set_err (TJAMD_ERR_CAPACITY, "%ld raw records, caller capacity %ld", n, capacity);
  if (n == 0) return 0;
  int rc = ensure (c->prefix, 64, c->stream);
  if (!rc) rc = ensure (c->rawlist, (size_t) n * 24, c->stream);
  if (rc) return -rc;
  if (hipMemsetAsync (c->prefix.p, 0, 8, c->stream) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "memset failed");
  const Buckets BK = make_buckets (c);
  switch (c->W) {
    case 1: hipLaunchKernelGGL (unpack_buckets_kernel<1>, dim3 (TJ_P), dim3 (256), 0, c->stream, BK, c->k, (u64 *) c->rawlist.p, (u64) n, (u64 *) c->prefix.p); break;
    case 2: hipLaunchKernelGGL (unpack_buckets_kernel<2>, dim3 (TJ_P), dim3 (256), 0, c->stream, BK, c->k, (u64 *) c->rawlist.p, (u64) n, (u64 *) c->prefix.p); break;
    default: hipLaunchKernelGGL (unpack_buckets_kernel<4>, dim3 (TJ_P), dim3 (256), 0, c->stream, BK, c->k, (u64 *) c->rawlist.p, (u64) n, (u64 *) c->prefix.p); break;
  }
  u64 n_out = 0;
  if (hipGetLastError () != hipSuccess || hipStreamSynchronize (c->stream) != hipSuccess ||
      hipMemcpy (&n_out, c->prefix.p, 8, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy (out, c->rawlist.p, (size_t) n * 24, hipMemcpyDeviceToHost) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "raw download failed");
  if ((long) n_out != n) return -set_err (TJAMD_ERR_STATE, "raw download found %llu records, cursors say %ld", (unsigned long long) n_out, n);
  return n;
}

extern "C" int tjamd_upload_raw (tjamd_counter *c, const hopo_element *elems, long n)
{
  if (!c || (n && !elems) || n < 0) return set_err (TJAMD_ERR_ARG, "bad arguments");
  if (n == 0) return TJAMD_OK;
  HIPCHK (hipSetDevice (c->device));
  int rc = sync_counters (c);
  if (!rc) rc = ensure_buckets (c, (u64) n, 0);
  c->raw_bound += (u64) n;
  if (!rc) rc = ensure (c->stage, (size_t) n * 40, c->stream);
  if (rc) return rc;
  HIPCHK (hipMemcpyAsync (c->stage.p, elems, (size_t) n * 40, hipMemcpyHostToDevice, c->stream));
  const unsigned grid = (unsigned) std::min<long> ((n + 255) / 256, 2048);
  const Buckets BK = make_buckets (c);
  switch (c->W) {
    case 1: hipLaunchKernelGGL (bin_elems_kernel<1>, dim3 (grid), dim3 (256), 0, c->stream, (const u64 *) c->stage.p, n, c->k, BK, c->d_ctr); break;
    case 2: hipLaunchKernelGGL (bin_elems_kernel<2>, dim3 (grid), dim3 (256), 0, c->stream, (const u64 *) c->stage.p, n, c->k, BK, c->d_ctr); break;
    default: hipLaunchKernelGGL (bin_elems_kernel<4>, dim3 (grid), dim3 (256), 0, c->stream, (const u64 *) c->stage.p, n, c->k, BK, c->d_ctr); break;
  }
  HIPCHK (hipGetLastError ());
  c->status = -1;
  return sync_counters (c);
}

// ---- device-wide exclusive scan ------------------------------------------------------------------------------------

static int exclusive_scan (tjamd_counter *c, const u32 *in, u32 *out, long n, u32 *total_out, u32 *tmp, size_t tmp_words)
{ // tmp: scratch for block sums of every level (n/4096 + n/4096^2 + ... + 2 words)
  if (n <= 0) return TJAMD_OK;
  long nblk = (n + SC_ITEMS - 1) / SC_ITEMS;
  if (nblk == 1) {
    hipLaunchKernelGGL (scan_apply_kernel, dim3 (1), dim3 (256), 0, c->stream, in, out, n, (const u32 *) nullptr, total_out);
    HIPCHK (hipGetLastError ());
    return TJAMD_OK;
  }
  if ((size_t) nblk > tmp_words) return set_err (TJAMD_ERR_STATE, "scan scratch too small");
  hipLaunchKernelGGL (scan_reduce_kernel, dim3 ((unsigned) nblk), dim3 (256), 0, c->stream, in, n, tmp);
  HIPCHK (hipGetLastError ());
  int rc = exclusive_scan (c, tmp, tmp, nblk, nullptr, tmp + nblk, tmp_words - (size_t) nblk);
  if (rc) return rc;
  hipLaunchKernelGGL (scan_apply_kernel, dim3 ((unsigned) nblk), dim3 (256), 0, c->stream, in, out, n, (const u32 *) tmp, total_out);
  HIPCHK (hipGetLastError ());
  return TJAMD_OK;
}

static size_t scan_tmp_words (long n)
{
  size_t w = 8;
  while (n > SC_ITEMS) { n = (n + SC_ITEMS - 1) / SC_ITEMS; w += (size_t) n; }
  return w;
}

static int radix_sort_records (tjamd_counter *c, u64 *&a, u64 *&b, long n)
{ // a: input (sorted result ends in a), b: scratch of the same size
  const int passes = key_passes (c->k);
  const int nblk = (int) ((n + RS_ITEMS - 1) / RS_ITEMS);
  const long nh = 256l * nblk;
  int rc = ensure (c->hist, (size_t) nh * 4, c->stream);
  if (!rc) rc = ensure (c->scan_tmp, std::max (scan_tmp_words (nh), scan_tmp_words (n)) * 4, c->stream);
  if (rc) return rc;
  for (int p = 0; p < passes; p++) {
    hipLaunchKernelGGL (radix_count_kernel, dim3 (nblk), dim3 (256), 0, c->stream, (const u64 *) a, n, p, c->k, (u32 *) c->hist.p, nblk);
    HIPCHK (hipGetLastError ());
    rc = exclusive_scan (c, (const u32 *) c->hist.p, (u32 *) c->hist.p, nh, nullptr, (u32 *) c->scan_tmp.p, c->scan_tmp.cap / 4);
    if (rc) return rc;
    hipLaunchKernelGGL (radix_scatter_kernel, dim3 (nblk), dim3 (256), 0, c->stream, (const u64 *) a, b, n, p, c->k, (const u32 *) c->hist.p, nblk);
    HIPCHK (hipGetLastError ());
    std::swap (a, b);
  }
  return TJAMD_OK;
}

static unsigned grid_for (long n) { return (unsigned) std::max<long> (1, std::min<long> ((n + 255) / 256, 4096)); }

// ---- finalise steps 3-4 + coverage, two ways --------------------------------------------------------------------


// radix path: stable LSD sort, then heads / scans / decisions as separate kernels (any bin occupancy, any size)
static int finalise_radix (tjamd_counter *c, long n1, int min_coverage)
{
  int rc = ensure (c->alt, (size_t) n1 * 24, c->stream);
  if (!rc) rc = ensure (c->flags, (size_t) n1 * 4, c->stream);
  if (!rc) rc = ensure (c->segid, (size_t) n1 * 4, c->stream);
  if (!rc) rc = ensure (c->headpos, (size_t) n1 * 4, c->stream);
  if (!rc) rc = ensure (c->keep, (size_t) n1 * 4, c->stream);
  if (!rc) rc = ensure (c->outpos, (size_t) n1 * 4, c->stream);
  if (!rc) rc = ensure (c->scan_tmp, scan_tmp_words (n1) * 4, c->stream);
  if (rc) return rc;
  u64 *a = (u64 *) c->kept.p, *b = (u64 *) c->alt.p;
  rc = radix_sort_records (c, a, b, n1);
  if (rc) return rc;
  if (a != (u64 *) c->kept.p) std::swap (c->kept, c->alt);

  // step 4: contexts deep enough get an index range (reference :388-404)
  const u64 *kept = (const u64 *) c->kept.p;
  u32 *flags = (u32 *) c->flags.p, *segid = (u32 *) c->segid.p, *headpos = (u32 *) c->headpos.p;
  u32 *keep = (u32 *) c->keep.p, *outpos = (u32 *) c->outpos.p;
  hipLaunchKernelGGL (seg_heads_kernel, dim3 (grid_for (n1)), dim3 (256), 0, c->stream, kept, n1, flags, 1);
  HIPCHK (hipGetLastError ());
  rc = exclusive_scan (c, flags, segid, n1, &c->d_fin->n_ctx, (u32 *) c->scan_tmp.p, c->scan_tmp.cap / 4);
  if (rc) return rc;
  hipLaunchKernelGGL (seg_headpos_kernel, dim3 (grid_for (n1)), dim3 (256), 0, c->stream, (const u32 *) flags, (const u32 *) segid, n1, headpos);
  HIPCHK (hipGetLastError ());
  hipLaunchKernelGGL (ctx_decide_kernel, dim3 (grid_for (n1)), dim3 (256), 0, c->stream, kept, n1, (const u32 *) headpos,
                      (const u32 *) &c->d_fin->n_ctx, min_coverage, keep);
  HIPCHK (hipGetLastError ());
  HIPCHK (hipMemcpyAsync (c->h_fin, c->d_fin, sizeof (FinCounts), hipMemcpyDeviceToHost, c->stream));
  HIPCHK (hipStreamSynchronize (c->stream));
  const long n_ctx = c->h_fin->n_ctx;
  rc = exclusive_scan (c, keep, outpos, n_ctx, &c->d_fin->n_idx, (u32 *) c->scan_tmp.p, c->scan_tmp.cap / 4);
  if (!rc) rc = ensure (c->idx_i, (size_t) n_ctx * 4, c->stream);
  if (!rc) rc = ensure (c->idx_f, (size_t) n_ctx * 4, c->stream);
  if (rc) return rc;
  hipLaunchKernelGGL (ctx_write_kernel, dim3 (grid_for (n_ctx)), dim3 (256), 0, c->stream, n1, (const u32 *) headpos, (const u32 *) &c->d_fin->n_ctx,
                      (const u32 *) keep, (const u32 *) outpos, (int *) c->idx_i.p, (int *) c->idx_f.p);
  HIPCHK (hipGetLastError ());

  // coverage (reference :419-438)
  const int log2t = cov_table_bits (n1);
  const long t = 1l << log2t;
  rc = ensure (c->cov, (size_t) t * 8, c->stream);
  if (rc) return rc;
  u64 *ctab = (u64 *) c->cov.p;
  HIPCHK (hipMemsetAsync (c->cov.p, 0, (size_t) t * 8, c->stream));
  hipLaunchKernelGGL (set_int_kernel, dim3 (1), dim3 (1), 0, c->stream, &c->d_fin->coverage, INT_MIN);
  hipLaunchKernelGGL (cov_insert_kernel, dim3 (grid_for (2 * n1)), dim3 (256), 0, c->stream, kept, n1, ctab, log2t);
  HIPCHK (hipGetLastError ());
  hipLaunchKernelGGL (cov_max_kernel, dim3 (std::min<unsigned> (grid_for (t), 256u)), dim3 (256), 0, c->stream, (const u64 *) ctab, t, &c->d_fin->coverage);
  HIPCHK (hipGetLastError ());
  HIPCHK (hipEventRecord (c->ev_f1, c->stream));
  c->fin_timed = true;
  HIPCHK (hipMemcpyAsync (c->h_fin, c->d_fin, sizeof (FinCounts), hipMemcpyDeviceToHost, c->stream));
  HIPCHK (hipStreamSynchronize (c->stream));
  return TJAMD_OK;
}

// binned path: kept (n1 records, any order) -> sorted kept, idx_i / idx_f, FinCounts{n_idx, coverage}; one host
// synchronisation at the end.  If a bin turns out too full every kernel after the bin scan does nothing, kept stays as
// it was and FinCounts::sort_fallback comes back set.
// n1 > 0: the kept count, known.  n1 == 0: not fetched yet -- the kernels take their sizes from the device-side plan
// (plan_tail has run), the buffers hold `cap` records, and the caller looks at what came of it after the one
// synchronisation at the end (finalise_speculative_ok).
static int finalise_binned (tjamd_counter *c, long n1, int min_coverage, long cap = 0, bool wait = true)
{
  const bool planned = n1 == 0;
  // (begun, not waited for, and a second stream is there: the step goes to it, behind what the counter's stream holds so far)
  const bool aside = planned && !wait && c->order_stream != nullptr && c->order_stream != c->stream;
  const hipStream_t st = aside ? c->order_stream : c->stream;
  const FinPlan *plan = planned ? &c->d_state->plan : nullptr;
  if (planned) n1 = cap;                                // (sizes everything below; the kernels use the plan's numbers)
  const int nbits = bin_bits_for (n1, c->k);
  const int nbins = planned ? BS_MAXBINS : 1 << nbits;  // (planned: the layout of binctx / binout must not depend on the count)
  // (planned: the table is allocated for the capacity's hash table, the kernels take their numbers from the plan -- a
  // table addressed by the key is never larger than that, see cov_plan_bits)
  int cov_direct = 0;
  const int log2t = planned ? cov_table_bits (n1) : cov_plan_bits (n1, c->k, &cov_direct);
  const long t = 1l << log2t;
  // the aggregation has counted the kept records into the finest bins (finalise_impl): no counting pass
  const bool fused = planned && c->bins_counted;
  c->bins_counted = false;
  int rc = ensure (c->alt, (size_t) n1 * 24, c->stream);
  if (!rc && !c->bins.p) { rc = ensure (c->bins, (size_t) (BS_MAXBINS + 1) * 4, c->stream); c->bins_zeroed = false; }
  if (!rc) rc = ensure (c->binstart, (size_t) (nbins + 1) * 4, c->stream);
  if (!rc) rc = ensure (c->binctx, (size_t) nbins * 8, c->stream);
  if (!rc) rc = ensure (c->headpos, (size_t) n1 * 4, c->stream);
  if (!rc) rc = ensure (c->outpos, (size_t) n1 * 4, c->stream);
  if (!rc) rc = ensure (c->idx_i, (size_t) n1 * 4, c->stream);
  if (!rc) rc = ensure (c->idx_f, (size_t) n1 * 4, c->stream);
  if (!rc) rc = ensure (c->cov, (size_t) t * 8, c->stream);
  if (rc) return rc;
  u32 *bins = (u32 *) c->bins.p, *binstart = (u32 *) c->binstart.p, *binctx = (u32 *) c->binctx.p, *binout = binctx + nbins;
  u32 *tstart = (u32 *) c->headpos.p, *tend = (u32 *) c->outpos.p;
  u64 *ctab = (u64 *) c->cov.p;
  if (!fused && !c->bins_zeroed) HIPCHK (hipMemsetAsync (bins, 0, (size_t) BS_MAXBINS * 4, c->stream));
  c->bins_zeroed = false;
  if (aside) { HIPCHK (hipEventRecord (c->ev_agg, c->stream)); HIPCHK (hipStreamWaitEvent (st, c->ev_agg, 0)); }
  // (planned: the grids are sized for a typical kept count, not for the buffers' capacity -- the kernels stride)
  const unsigned g1 = planned ? std::min<unsigned> (grid_for (n1), 2048u) : grid_for (n1);
  if (!fused) hipLaunchKernelGGL (bin_count_kernel, dim3 (g1), dim3 (256), 0, st, (const u64 *) c->kept.p, n1, c->k, nbits, bins,
                                  (uint4 *) c->cov.p, (long) (t / 2), 0, plan);
  hipLaunchKernelGGL (bin_scan_kernel, dim3 (1), dim3 (1024), 0, st, bins, nbins, binstart, c->bin_rank_max, c->d_fin, plan);
  hipLaunchKernelGGL (bin_scatter_kernel, dim3 (g1), dim3 (256), 0, st, (const u64 *) c->kept.p, (u64 *) c->alt.p, n1, c->k, nbits, bins, plan,
                      fused ? (uint4 *) c->cov.p : (uint4 *) nullptr);
  hipLaunchKernelGGL (bin_sort_index_kernel, dim3 ((unsigned) std::min (nbins / BSI_WAVES + 1, 16384)), dim3 (64 * BSI_WAVES), 0, st,
                      (const u64 *) c->alt.p, (u64 *) c->kept.p, (const u32 *) binstart, nbins, (const FinCounts *) c->d_fin, min_coverage,
                      ctab, log2t, binctx, tstart, tend, plan, cov_direct);
  hipLaunchKernelGGL (bin_ctx_scan_kernel, dim3 (1), dim3 (1024), 0, st, (const u32 *) binctx, nbins, binout, c->d_fin, plan);
  hipLaunchKernelGGL (bin_ctx_write_kernel, dim3 (256), dim3 (256), 0, st, (const u32 *) binstart, (const u32 *) binctx, (const u32 *) binout, nbins,
                      (const u32 *) tstart, (const u32 *) tend, (int *) c->idx_i.p, (int *) c->idx_f.p, (const u64 *) ctab, t, c->d_fin, plan);
  HIPCHK (hipGetLastError ());
  HIPCHK (hipEventRecord (c->ev_f1, st));
  c->fin_timed = true;
  if (planned) HIPCHK (hipMemcpyAsync (c->h_state, c->d_state, sizeof (DevState), hipMemcpyDeviceToHost, st));   // (counts of the sample, plan, results: one copy)
  else HIPCHK (hipMemcpyAsync (c->h_fin, c->d_fin, sizeof (FinCounts), hipMemcpyDeviceToHost, c->stream));
  if (wait) HIPCHK (hipStreamSynchronize (c->stream));
  else HIPCHK (hipEventRecord (c->ev_done, st));
  return TJAMD_OK;
}

// tjamd_finalise in two halves: _begin queues everything (one stream, no host synchronisation when the ordering step can be
// planned on the device) and _end waits for the counts -- an event, not the stream: what the caller has queued behind
// (the next sample's scan on another counter of the same stream) runs on -- and finishes the bookkeeping.
static int finalise_impl (tjamd_counter *c, int remove_biased, int min_coverage, int *status, int phase);

extern "C" int tjamd_finalise (tjamd_counter *c, int remove_biased, int min_coverage, int *status)
{
  int rc = finalise_impl (c, remove_biased, min_coverage, status, 0);
  return rc;
}
extern "C" int tjamd_finalise_begin (tjamd_counter *c, int remove_biased, int min_coverage)
{
  const int rc = finalise_impl (c, remove_biased, min_coverage, nullptr, 1);
  if (!rc && c->fin_pending == 0) c->fin_pending = 3;   // (the outcome was clear at once, e.g. no raw records: _end reports it)
  return rc;
}
extern "C" int tjamd_finalise_end (tjamd_counter *c, int *status)
{
  if (!c) return set_err (TJAMD_ERR_ARG, "null counter");
  if (!c->fin_pending) return set_err (TJAMD_ERR_STATE, "tjamd_finalise_end without tjamd_finalise_begin");
  if (c->fin_pending == 3) { c->fin_pending = 0; if (status) *status = c->status; return TJAMD_OK; }
  return finalise_impl (c, c->fin_rb, c->fin_mc, status, 2);
}

// phase 0: all of it; 1: up to the point where the host has to know the counts; 2: from there
static int finalise_impl (tjamd_counter *c, int remove_biased, int min_coverage, int *status, int phase)
{
  if (!c) return set_err (TJAMD_ERR_ARG, "null counter");
  HIPCHK (hipSetDevice (c->device));
  bool speculative = c->fin_speculative, plan_ahead = c->fin_plan_ahead;
  u64 kept_cap = c->fin_kept_cap;
  int rc = TJAMD_OK;
  if (phase == 2) goto second_half;
  c->fin_pending = 0;
  {
  // The raw count sizes the kept list.  A survivor stands for at least two raw records (both strands seen, or a count
  // above one), so half of the host's running upper bound of the raw count is a safe capacity: unless that is a lot of
  // memory, the aggregation is launched without first asking the device (one host round trip less) and the exact
  // counts and error flags, copied in stream order before the buckets are cleared, are looked at afterwards.
  c->n_kept = 0; c->n_idx = 0; c->coverage = 0; c->fin_timed = false;
  speculative = (c->raw_bound / 2 + 1) * 24 <= (8ull << 30);
  kept_cap = c->raw_bound / 2 + 1;
  if (!speculative) {
    rc = sync_counters (c);
    if (rc) return rc;
    if (c->n_raw_known == 0) { c->status = 1; if (status) *status = 1; return TJAMD_OK; }     // reference: src/hopo_counter.c:345-349
    if (c->n_raw_known >= (1l << 31)) return set_err (TJAMD_ERR_CAPACITY, "%ld raw records exceed the reference's int n_elem", c->n_raw_known);
    kept_cap = (u64) c->n_raw_known / 2 + 1;
  }

  // steps 1-2: per-bucket hash aggregation + filter (reference :351-374)
  rc = ensure (c->kept, (size_t) kept_cap * 24, c->stream);
  if (!rc) rc = ensure (c->ovf, c->pool.cap, c->stream);   // second pool for the aggregation's leftover rounds
  if (rc) return rc;
  HIPCHK (hipEventRecord (c->ev_f0, c->stream));
  const Buckets BK = make_buckets (c);
  // The ordering step is launched right behind the aggregation, sized on the device (plan_tail, by the aggregation's last
  // workgroup) for up to kept_cap / 8 records -- a sample keeps a per cent or so of its raw records -- so that the whole
  // finalise has one host round trip; if more were kept, its kernels do nothing and the step is run again below with
  // the count in hand.
  long plan_cap = (long) std::min<u64> (kept_cap, std::max<u64> (kept_cap / 8, 1u << 16));
  if (const char *pc = getenv ("TATAJUBA_AMD_PLAN_CAP")) plan_cap = std::max (1l, std::min (plan_cap, atol (pc)));     // (tests: make the second attempt happen)
  plan_ahead = speculative && getenv ("TATAJUBA_AMD_NO_PLAN") == nullptr;
  FinPlan *const d_plan = plan_ahead ? &c->d_state->plan : nullptr;
  // with the ordering step planned ahead the aggregation counts what it keeps into the finest bins of that step
  // (clear_buckets_kernel, next on the stream, sums them to the bins the plan asks for and zeroes them again)
  u32 *fine = nullptr;
  if (plan_ahead && getenv ("TATAJUBA_AMD_NO_FUSED_BINS") == nullptr) {
    const bool fresh = c->fine.p == nullptr;
    rc = ensure (c->fine, (size_t) BS_MAXBINS * 4, c->stream);
    if (rc) return rc;
    if (fresh || c->fine_dirty) HIPCHK (hipMemsetAsync (c->fine.p, 0, (size_t) BS_MAXBINS * 4, c->stream));
    if (!c->bins.p) { rc = ensure (c->bins, (size_t) (BS_MAXBINS + 1) * 4, c->stream); c->bins_zeroed = false; if (rc) return rc; }   // (clear_buckets_kernel writes the step's counts there)
    fine = (u32 *) c->fine.p;
    c->fine_dirty = true;
  }
  const int fbits = fine_bin_bits (c->k);
  switch (c->W) {
    case 1: hipLaunchKernelGGL (aggregate1_kernel, dim3 (TJ_P), dim3 (AG_BLOCK), 0, c->stream, BK, (u64 *) c->ovf.p, c->k, remove_biased, (u64 *) c->kept.p, kept_cap, c->d_fin, plan_cap, d_plan, fine, fbits); break;
    case 2: hipLaunchKernelGGL (aggregate2_kernel, dim3 (TJ_P), dim3 (AG_BLOCK), 0, c->stream, BK, (u64 *) c->ovf.p, c->k, remove_biased, (u64 *) c->kept.p, kept_cap, c->d_fin, plan_cap, d_plan, fine, fbits); break;
    default: hipLaunchKernelGGL (aggregate4_kernel, dim3 (TJ_P), dim3 (AG_BLOCK), 0, c->stream, BK, (u64 *) c->ovf.p, c->k, remove_biased, (u64 *) c->kept.p, kept_cap, c->d_fin, plan_cap, d_plan, fine, fbits); break;
  }
  HIPCHK (hipGetLastError ());
  // counters, bucket sizes and the aggregation's own counts, in one copy; then the buckets are emptied (the aggregation
  // consumed them: leftover rounds reuse their fronts) together with the counts the next aggregation adds to
  // (with the ordering step planned ahead the copy waits until that has run: clear_buckets_kernel keeps what it zeroes)
  rc = plan_ahead ? TJAMD_OK : queue_counter_copies (c);
  if (!rc) rc = clear_buckets (c, plan_ahead);
  if (rc) return rc;
  c->fin_rb = remove_biased; c->fin_mc = min_coverage; c->fin_speculative = speculative; c->fin_plan_ahead = plan_ahead; c->fin_kept_cap = kept_cap;
  if (plan_ahead) { rc = finalise_binned (c, 0, min_coverage, plan_cap, phase != 1); if (rc) return rc; }   // (ends with a copy of the counts and a synchronisation, or an event)
  else HIPCHK (hipStreamSynchronize (c->stream));
  }
  if (phase == 1) { c->fin_pending = plan_ahead ? 1 : 2; return TJAMD_OK; }    // (2: nothing left to wait for)
second_half:
  if (phase == 2) {
    if (c->fin_pending == 1) HIPCHK (hipEventSynchronize (c->ev_done));
    c->fin_pending = 0;
  }
  if (speculative) {
    rc = apply_counter_copies (c, plan_ahead);
    const long n = c->n_raw_known;
    c->n_raw_known = 0; c->raw_bound = 0; c->bucket_bound = 0; c->chunk_bound = TJ_P;      // (the buckets are empty again)
    if (rc) return rc;
    if (n == 0) { c->status = 1; if (status) *status = 1; return TJAMD_OK; }                // reference: src/hopo_counter.c:345-349
    if (n >= (1l << 31)) return set_err (TJAMD_ERR_CAPACITY, "%ld raw records exceed the reference's int n_elem", n);
  }
  if (plan_ahead) {
    // what the aggregation's last workgroup read against what the kernel boundary behind it shows (the
    // last-workgroup pattern has no fence); a difference is repaired by sizing the ordering step again, from the exact count
    FinPlan &pl = c->h_state->plan;
    if (pl.n1 != pl.n1_exact || pl.kept_overflow != pl.ovf_exact) {
      c->plan_mismatches++;
      pl.n1 = pl.n1_exact; pl.kept_overflow = pl.ovf_exact; pl.ok = 0;
    }
  }
  if (plan_ahead ? c->h_state->plan.kept_overflow : c->h_fin->overflow) return set_err (TJAMD_ERR_CAPACITY, "kept list overflow");
  const long n1 = plan_ahead ? c->h_state->plan.n1 : (long) c->h_fin->n_kept;
  if (n1 == 0) {                                                               // reference :376-381
    HIPCHK (hipEventRecord (c->ev_f1, c->stream)); c->fin_timed = true;
    c->status = 2; if (status) *status = 2; return TJAMD_OK;
  }

  // steps 3-4 + coverage: bin partition, then one pass per bin sorts it and derives the context index ranges and the
  // coverage table entries; the stable radix passes + separate index kernels take over if a bin is too full.
  c->n_kept = n1;
  if (!(plan_ahead && c->h_state->plan.ok)) rc = finalise_binned (c, n1, min_coverage);   // (not done yet, or more kept than planned for)
  if (!rc && c->h_fin->sort_fallback) rc = finalise_radix (c, n1, min_coverage);
  if (rc) return rc;
  c->n_idx = (int) c->h_fin->n_idx;
  if (c->n_idx == 0) { c->status = 3; if (status) *status = 3; return TJAMD_OK; }   // reference :406-411 (coverage not estimated)
  c->coverage = c->h_fin->coverage;
  c->status = 0;
  if (status) *status = 0;
  return TJAMD_OK;
}

extern "C" long tjamd_kept_count (tjamd_counter *c) { return c ? c->n_kept : -1; }
extern "C" int tjamd_n_idx (tjamd_counter *c) { return c ? c->n_idx : -1; }
extern "C" int tjamd_coverage (tjamd_counter *c) { return c ? c->coverage : -1; }
extern "C" const void *tjamd_kept_device_ptr (tjamd_counter *c) { return c ? c->kept.p : NULL; }

extern "C" long tjamd_download_kept (tjamd_counter *c, hopo_element *out, long capacity)
{
  if (!c) return -set_err (TJAMD_ERR_ARG, "null counter");
  if (c->status < 0) return -set_err (TJAMD_ERR_STATE, "counter not finalised");
  const long n1 = c->n_kept;
  if (n1 > capacity) return -set_err (TJAMD_ERR_CAPACITY, "%ld kept records, caller capacity %ld", n1, capacity);
  if (n1 == 0) return 0;
  if (hipSetDevice (c->device) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "hipSetDevice failed");
  // (into pinned memory that the counter keeps: a fresh pageable vector of n1 records cost its page faults and a staged
  // copy every time -- a third of what finalise_hopo_counter spends behind the device finalise)
  if ((size_t) n1 * 24 > c->h_kept_cap) {
    if (c->h_kept) (void) hipHostFree (c->h_kept);
    c->h_kept = nullptr; c->h_kept_cap = 0;
    const size_t want = (size_t) n1 * 24 + ((size_t) n1 * 24 >> 2) + 4096;
    if (hipHostMalloc (&c->h_kept, want, hipHostMallocDefault) != hipSuccess) { c->h_kept = nullptr; return -set_err (TJAMD_ERR_HIP, "hipHostMalloc of %zu bytes failed", want); }
    c->h_kept_cap = want;
  }
  const tjamd_record *tmp = (const tjamd_record *) c->h_kept;
  if (hipMemcpyAsync (c->h_kept, c->kept.p, (size_t) n1 * 24, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
      hipStreamSynchronize (c->stream) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "download failed");
  for (long i = 0; i < n1; i++) {
    out[i].context[0] = tmp[i].ctx0; out[i].context[1] = tmp[i].ctx1;
    memcpy ((char *) &out[i] + 16, &tmp[i].meta, 8);
    out[i].read_offset = -1;                    // state after reference src/hopo_counter.c:511
    out[i].loc_ref_id = out[i].loc_pos = out[i].loc_last = -1;
  }
  return n1;
}

extern "C" long tjamd_download_idx (tjamd_counter *c, int *idx_initial, int *idx_final, long capacity)
{
  if (!c) return -set_err (TJAMD_ERR_ARG, "null counter");
  if (c->status < 0) return -set_err (TJAMD_ERR_STATE, "counter not finalised");
  const long n = c->n_idx;
  if (n > capacity) return -set_err (TJAMD_ERR_CAPACITY, "%ld index ranges, caller capacity %ld", n, capacity);
  if (n == 0) return 0;
  if (hipSetDevice (c->device) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "hipSetDevice failed");
  if (hipMemcpy (idx_initial, c->idx_i.p, (size_t) n * 4, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy (idx_final, c->idx_f.p, (size_t) n * 4, hipMemcpyDeviceToHost) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "download failed");
  return n;
}

extern "C" double tjamd_last_scan_ms (tjamd_counter *c)
{
  if (!c || !c->scan_timed) return -1.0;
  float ms = 0.f;
  if (hipSetDevice (c->device) != hipSuccess || hipEventSynchronize (c->ev_s1) != hipSuccess || hipEventElapsedTime (&ms, c->ev_s0, c->ev_s1) != hipSuccess) return -1.0;
  // (a stream scanned in pieces: the partition kernels between the scans are not the scan's time)
  for (int i = 0; i < c->n_piece_ev; i++) { float p = 0.f; if (hipEventElapsedTime (&p, c->ev_pa[i], c->ev_pb[i]) == hipSuccess) ms -= p; }
  return (double) ms;
}

extern "C" double tjamd_last_merge_ms (tjamd_counter *c)
{ // the kernels of the last tjamd_merge_samples on this counter (bin path), HIP events on its stream
  if (!c || !c->merge_timed) return -1.0;
  float ms = 0.f;
  if (hipSetDevice (c->device) != hipSuccess || hipEventSynchronize (c->ev_m1) != hipSuccess || hipEventElapsedTime (&ms, c->ev_m0, c->ev_m1) != hipSuccess) return -1.0;
  return (double) ms;
}

// 1: one-word records go through the record log and partition_log_kernel (k <= 12, the default); 0: every scan kernel
// partitions its records itself
extern "C" int tjamd_counter_uses_log (const tjamd_counter *c) { return (c && c->W <= c->log_mode && c->fast_mode) ? 1 : 0; }

extern "C" double tjamd_last_partition_ms (tjamd_counter *c)
{ // partition_log_kernel behind the last scan launch (k <= 12, record log); 0 when the scan partitioned by itself
  if (!c || !c->scan_timed) return -1.0;
  if (!c->part_timed) return 0.0;
  float ms = 0.f;
  if (hipSetDevice (c->device) != hipSuccess || hipEventSynchronize (c->ev_p1) != hipSuccess || hipEventElapsedTime (&ms, c->ev_s1, c->ev_p1) != hipSuccess) return -1.0;
  for (int i = 0; i < c->n_piece_ev; i++) { float p = 0.f; if (hipEventElapsedTime (&p, c->ev_pa[i], c->ev_pb[i]) == hipSuccess) ms += p; }   // (the pieces before the last)
  return (double) ms;
}

extern "C" double tjamd_last_finalise_ms (tjamd_counter *c)
{
  if (!c || !c->fin_timed) return -1.0;
  float ms = 0.f;
  if (hipSetDevice (c->device) != hipSuccess || hipEventSynchronize (c->ev_f1) != hipSuccess || hipEventElapsedTime (&ms, c->ev_f0, c->ev_f1) != hipSuccess) return -1.0;
  return (double) ms;
}

extern "C" long tjamd_last_scan_launches (tjamd_counter *c) { return c ? c->last_scan_launches : -1; }
extern "C" long tjamd_plan_mismatches (tjamd_counter *c) { return c ? c->plan_mismatches : -1; }

// radix path of the merge: tag, stable sort, run heads, scans (any size, any skew)
static long merge_samples_radix (tjamd_counter *c, const void *d_records, long n, int n_samples, void *d_out_keys, void *d_out_counts, long capacity);

// ---------------------------------------------------------------------------------------------------------------
// "next" rows N3 / N1 on the device: grouping of near-identical contexts within a sample, tract ids on a union.
//
// Grouping (reference: new_genomic_context_list, src/context_histogram.c:245-270 with the distance of :25-48): the
// reference walks the elements in order and lets element i join the group being built iff it has the group's base and
// is closer than max_distance_per_flank (differing flank bases, both flanks together) to EVERY context already in it;
// otherwise i starts a new group.  (It walks them in BWA-location order and retries with a Levenshtein distance; here
// the order is the finalised array's own and there is no retry: the aligner and biomcmc-lib are absent.)
// On the device: back[i] = nearest j < i that element i could not share a group with (a thread walks back until it
// meets one); i can join iff the current group started after back[i].  Positions with back[i] == i - 1 start a group
// whatever came before, so the array falls into independent stretches, each walked by one thread.

__device__ __forceinline__ int flank_hamming (u64 a, u64 b)
{ // differing 2-bit positions of two packed k-mers (reference: src/hopo_counter.c:61-68 counts them one by one)
  const u64 d = a ^ b;
  return __popcll ((d | (d >> 1)) & 0x5555555555555555ull);
}

__global__ void group_back_kernel (const u64 *__restrict__ kept, long n, int maxd, int *__restrict__ back)
{
  for (long i = blockIdx.x * (long) blockDim.x + threadIdx.x; i < n; i += (long) gridDim.x * blockDim.x) {
    const u64 c0 = kept[3 * i], c1 = kept[3 * i + 1], base = kept[3 * i + 2] & 3ull;
    long j = i - 1;
    while (j >= 0 && (kept[3 * j + 2] & 3ull) == base && flank_hamming (kept[3 * j], c0) + flank_hamming (kept[3 * j + 1], c1) < maxd) j--;
    back[i] = (int) j;
  }
}

__global__ void group_resolve_kernel (const int *__restrict__ back, long n, u32 *__restrict__ head)
{
  for (long i = blockIdx.x * (long) blockDim.x + threadIdx.x; i < n; i += (long) gridDim.x * blockDim.x) {
    if (back[i] != (int) i - 1) continue;               // (i == 0: back = -1) only stretch starts walk
    head[i] = 1u;
    long start = i;
    for (long j = i + 1; j < n && back[j] != (int) j - 1; j++) {
      if (start > (long) back[j]) head[j] = 0u;
      else { head[j] = 1u; start = j; }
    }
  }
}

struct GroupOut { int first, n_elem, n_context, mode; long long integral; };

__global__ void group_summary_kernel (const u64 *__restrict__ kept, long n, const u32 *__restrict__ head, const u32 *__restrict__ gid_excl,
                                      int *__restrict__ group_of, GroupOut *__restrict__ groups)
{
  for (long i = blockIdx.x * (long) blockDim.x + threadIdx.x; i < n; i += (long) gridDim.x * blockDim.x) {
    if (!head[i]) continue;
    const int g = (int) gid_excl[i];                     // groups before this head
    GroupOut o = {(int) i, 0, 0, (int) i, 0};
    int mode_count = 0;
    for (long j = i; j < n && (j == i || !head[j]); j++) {   // (reference: context_histogram_add_hopo_elem, src/context_histogram.c:181-222)
      const u64 m = kept[3 * j + 2];
      int cnt = (int) ((m >> TJ_META_COUNT_SHIFT) & 0xFFFFFull);
      if (cnt & 0x80000) cnt -= 0x100000;               // signed 20-bit field
      if (j == i || kept[3 * j] != kept[3 * (j - 1)] || kept[3 * j + 1] != kept[3 * (j - 1) + 1]) o.n_context++;
      if (j == i || mode_count < cnt) { mode_count = cnt; o.mode = (int) j; }
      o.integral += cnt; o.n_elem++;
      group_of[j] = g;
    }
    groups[g] = o;
  }
}

extern "C" long tjamd_group_contexts (tjamd_counter *c, int max_distance_per_flank, int *group_of, tjamd_group *groups, long capacity)
{
  if (!c || max_distance_per_flank < 0) return -set_err (TJAMD_ERR_ARG, "bad arguments");
  if (c->status < 0) return -set_err (TJAMD_ERR_STATE, "tjamd_group_contexts needs a finalised counter");
  if (hipSetDevice (c->device) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "hipSetDevice failed");
  const long n = c->n_kept;
  if (n == 0) return 0;
  static_assert (sizeof (GroupOut) == sizeof (tjamd_group), "group layout");
  int rc = ensure (c->headpos, (size_t) n * 4, c->stream);          // back[]
  if (!rc) rc = ensure (c->flags, (size_t) n * 4, c->stream);      // head flags
  if (!rc) rc = ensure (c->outpos, (size_t) n * 4, c->stream);     // groups before each element
  if (!rc) rc = ensure (c->segid, (size_t) n * 4, c->stream);      // group_of
  if (!rc) rc = ensure (c->scan_tmp, scan_tmp_words (n) * 4 + 64, c->stream);
  if (!rc) rc = ensure (c->alt, (size_t) n * sizeof (GroupOut), c->stream);
  if (rc) return -rc;
  const u64 *kept = (const u64 *) c->kept.p;
  u32 *total = (u32 *) c->scan_tmp.p + scan_tmp_words (n);
  hipLaunchKernelGGL (group_back_kernel, dim3 (grid_for (n)), dim3 (256), 0, c->stream, kept, n, max_distance_per_flank, (int *) c->headpos.p);
  hipLaunchKernelGGL (group_resolve_kernel, dim3 (grid_for (n)), dim3 (256), 0, c->stream, (const int *) c->headpos.p, n, (u32 *) c->flags.p);
  rc = exclusive_scan (c, (const u32 *) c->flags.p, (u32 *) c->outpos.p, n, total, (u32 *) c->scan_tmp.p, scan_tmp_words (n));
  if (rc) return -rc;
  hipLaunchKernelGGL (group_summary_kernel, dim3 (grid_for (n)), dim3 (256), 0, c->stream, kept, n, (const u32 *) c->flags.p, (const u32 *) c->outpos.p,
                      (int *) c->segid.p, (GroupOut *) c->alt.p);
  if (hipGetLastError () != hipSuccess) return -set_err (TJAMD_ERR_HIP, "grouping launch failed");
  u32 ng = 0;
  if (hipMemcpyAsync (&ng, total, 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize (c->stream) != hipSuccess)
    return -set_err (TJAMD_ERR_HIP, "grouping failed: %s", hipGetErrorString (hipGetLastError ()));
  if ((long) ng > capacity && groups) return -set_err (TJAMD_ERR_CAPACITY, "%u groups, caller capacity %ld", ng, capacity);
  if (group_of && hipMemcpy (group_of, c->segid.p, (size_t) n * 4, hipMemcpyDeviceToHost) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "copy failed");
  if (groups && hipMemcpy (groups, c->alt.p, (size_t) ng * sizeof (GroupOut), hipMemcpyDeviceToHost) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "copy failed");
  return (long) ng;
}

// ---- the whole grouping step of new_genomic_context_list (reference: src/context_histogram.c:245-270): an element that
// fails the flank-distance test against the histogram being built is tried again with an edit distance between the two
// "left.B.right" names (:19-23, :255-261: joins if it is below opt.levenshtein_distance and marks the histogram `indel`),
// then every histogram gets its tract-length histogram (:278-286).
//
// The retry makes the step sequential in principle: whether element i joins depends on which element currently carries
// the histogram's name (the one with the highest count so far).  On the device it is speculated: (1) the grouping without
// the retry (group_back / group_resolve above) is computed in parallel; (2) every element that opens a group there is
// tested, in parallel, against the name the group before it ends up with -- which is what the sequential pass would test it
// against as long as no retry has succeeded since the last point where both agree; (3) where such a test succeeds (rare:
// unrelated contexts are a dozen edits apart) one thread replays the reference's loop from there until it opens a group at
// an element that opens one in (1) as well -- from that element on the two passes are in the same state again.
//
// The edit distance stands for biomcmc_levenshtein_distance (s1, n, s2, n, 1, 1, true) of biomcmc-lib, whose source is
// absent from the reference tree: unit-cost global edit distance (UNPINNED, see oracle/context_oracle.c).

__device__ __forceinline__ u32 name_symbol (u64 c0, u64 c1, u32 base, int k, int t)
{ // t-th character of generate_name_from_flanking_contexts (context, base, k, false) as a small integer ('.' = 4)
  if (t < k) return (u32) (c0 >> (2 * t)) & 3u;
  if (t == k + 1) return base;
  if (t < k + 3) return 4u;
  return (u32) (c1 >> (2 * (t - k - 3))) & 3u;
}

__device__ int name_edit_distance (const u64 *__restrict__ kept, long a, long b, int k, int free_end)
{
  const u64 a0 = kept[3 * a], a1 = kept[3 * a + 1], b0 = kept[3 * b], b1 = kept[3 * b + 1];
  const u32 ab = (u32) (kept[3 * a + 2] & 3ull), bb = (u32) (kept[3 * b + 2] & 3ull);
  const int n = 2 * k + 3;
  unsigned char row[2 * 32 + 4];
  for (int y = 0; y <= n; y++) row[y] = (unsigned char) y;
  int early = n;                                        // (free_end) a used up before b
  for (int x = 1; x <= n; x++) {
    const u32 sb = name_symbol (b0, b1, bb, k, x - 1);
    int diag = row[0];
    row[0] = (unsigned char) x;
    for (int y = 1; y <= n; y++) {
      const int up = row[y];
      int best = diag + (name_symbol (a0, a1, ab, k, y - 1) == sb ? 0 : 1);
      best = min (best, min ((int) row[y - 1], up) + 1);
      diag = up; row[y] = (unsigned char) best;
    }
    early = min (early, (int) row[n]);
  }
  int result = row[n];
  if (free_end) {                                       // the other reading of the absent function's last argument: one name may end early
    result = min (result, early);
    for (int y = 0; y <= n; y++) result = min (result, (int) row[y]);
  }
  return result;
}

// join types on the device: 0 opened its histogram, 1 joined within the flank distance and met its own context in the
// list, 5 joined within the flank distance and added a context, 2 taken in by the retry (always adds a context); the host
// sees type & 3
#define GJ_ADDS_CONTEXT(t) ((t) != 1)

// (2): elements that open a group in the grouping without the retry, tested against the name of the group before them
__global__ void group_speculate_kernel (const u64 *__restrict__ kept, long n, int k, int lev, int free_end, const u32 *__restrict__ head,
                                        u32 *__restrict__ cand, int *__restrict__ jt, u32 *__restrict__ n_cand)
{
  for (long i = blockIdx.x * (long) blockDim.x + threadIdx.x; i < n; i += (long) gridDim.x * blockDim.x) {
    u32 c = 0;
    // (without the retry every context of a histogram is within the flank distance of every other, the distance loop runs
    // through all of them, and an identical context -- the element before, in the sorted array -- is met)
    jt[i] = head[i] ? 0 : ((kept[3 * i] == kept[3 * (i - 1)] && kept[3 * i + 1] == kept[3 * (i - 1) + 1]) ? 1 : 5);
    if (head[i] && i > 0 && ((kept[3 * i + 2] ^ kept[3 * (i - 1) + 2]) & 3ull) == 0ull) {
      long ph = i - 1;
      while (!head[ph]) ph--;
      long mode = ph;
      int mc = meta_count (kept[3 * ph + 2]);
      for (long j = ph + 1; j < i; j++) { const int cj = meta_count (kept[3 * j + 2]); if (cj > mc) { mc = cj; mode = j; } }
      if (name_edit_distance (kept, mode, i, k, free_end) < lev) { c = 1u; atomicAdd (n_cand, 1u); }
    }
    cand[i] = c;
  }
}

// (3): one workgroup walks the candidates in order; thread 0 replays the reference's loop from each one that lies past
// the stretch the replay before it covered.  Once a histogram holds a context that came in through the retry its contexts
// are no longer all close to one another, and the reference's distance loop (src/context_histogram.c:36-46) matters as
// written: contexts in the order they were added, give up at the first one 2 * max_distance or more away, stop with
// distance 0 at an identical one (whatever comes after it), otherwise the largest distance met.
#define GR_CHUNK 4096
__global__ __launch_bounds__ (256)
void group_repair_kernel (const u64 *__restrict__ kept, long n, int k, int maxd, int lev, int free_end,
                          u32 *__restrict__ head, const u32 *__restrict__ cand, int *__restrict__ jt)
{
  __shared__ long s_pos, s_min;
  if (threadIdx.x == 0) s_pos = 0;
  __syncthreads ();
  for (;;) {
    const long start = s_pos;
    if (start >= n) break;
    if (threadIdx.x == 0) s_min = n;
    __syncthreads ();
    const long end = min (n, start + (long) GR_CHUNK);
    for (long i = start + threadIdx.x; i < end; i += blockDim.x)
      if (cand[i]) { atomicMin ((unsigned long long *) &s_min, (unsigned long long) i); break; }
    __syncthreads ();
    const long h = s_min;
    if (threadIdx.x == 0) {
      if (h >= n) s_pos = end;
      else {
        long hd = h - 1;
        while (!head[hd]) hd--;                           // (flags before h are final: h lies past every earlier replay)
        long mode = hd;
        int mc = meta_count (kept[3 * hd + 2]);
        for (long j = hd + 1; j < h; j++) { const int cj = meta_count (kept[3 * j + 2]); if (cj > mc) { mc = cj; mode = j; } }
        long i = h;
        while (i < n) {
          if ((kept[3 * i + 2] ^ kept[3 * hd + 2]) & 3ull) break;          // another base: opens a group in both passes, untouched
          const u64 c0 = kept[3 * i], c1 = kept[3 * i + 1];
          int this_max = 0;
          bool fail = false, matched = false;
          for (long j = hd; j < i && !fail && !matched; j++) {
            if (j > hd && !GJ_ADDS_CONTEXT (jt[j])) continue;
            int d = min (flank_hamming (kept[3 * j], c0), 2 * maxd);
            if (d >= 2 * maxd) { fail = true; break; }
            d += min (flank_hamming (kept[3 * j + 1], c1), 2 * maxd - d);
            if (d >= 2 * maxd) { fail = true; break; }
            this_max = max (this_max, d);
            if (d == 0) matched = true;
          }
          int type = 0;
          if (!fail && (matched ? 0 : this_max) < maxd) type = matched ? 1 : 5;
          else if (name_edit_distance (kept, mode, i, k, free_end) < lev) type = 2;  // the indel retry
          if (type) {
            head[i] = 0u; jt[i] = type;
            const int ci = meta_count (kept[3 * i + 2]);
            if (ci > mc) { mc = ci; mode = i; }
            i++;
          }
          else {
            const bool was_head = head[i] != 0u;
            head[i] = 1u; jt[i] = 0; hd = i; mode = i; mc = meta_count (kept[3 * i + 2]);
            i++;
            if (was_head) break;                          // both passes open a group here: the same state from now on
          }
        }
        s_pos = i;
      }
    }
    __syncthreads ();
  }
}

struct CtxGroupOut { int first, n_elem, n_context, mode, indel, n_len, modal_len, modal_freq; long long integral; };
struct LenFreq { int length, freq; };

__global__ void group_histogram_kernel (const u64 *__restrict__ kept, long n, const u32 *__restrict__ head, const u32 *__restrict__ gid_excl,
                                        int *__restrict__ jt, int *__restrict__ group_of, CtxGroupOut *__restrict__ groups, LenFreq *__restrict__ hist)
{
  for (long i = blockIdx.x * (long) blockDim.x + threadIdx.x; i < n; i += (long) gridDim.x * blockDim.x) {
    if (!head[i]) continue;
    const int g = (int) gid_excl[i];
    CtxGroupOut o = {(int) i, 0, 0, (int) i, 0, 0, 0, 0, 0};
    int mode_count = 0;
    LenFreq *h = hist + i;                              // (a histogram has at most as many lengths as its group has elements)
    for (long j = i; j < n && (j == i || !head[j]); j++) {
      const u64 m = kept[3 * j + 2];
      const int cnt = meta_count (m);
      int len = (int) ((m >> TJ_META_LEN_SHIFT) & 0x3FFull);
      if (len & 0x200) len -= 0x400;                    // signed 10-bit field
      // contexts of the histogram (reference: context_histogram_add_hopo_elem, src/context_histogram.c:184-190): one more unless
      // the distance loop met an identical context; an element taken in by the indel retry is appended whatever the list holds
      const int t = jt[j];
      if (j == i || GJ_ADDS_CONTEXT (t)) o.n_context++;
      if (t == 2) o.indel = 1;
      jt[j] = t & 3;
      if (j == i || mode_count < cnt) { mode_count = cnt; o.mode = (int) j; }
      o.integral += cnt; o.n_elem++;
      int q = 0;
      while (q < o.n_len && h[q].length != len) q++;
      if (q == o.n_len) { h[q].length = len; h[q].freq = 0; o.n_len++; }
      h[q].freq += cnt;
      group_of[j] = g;
    }
    for (int a = 1; a < o.n_len; a++) {                 // highest summed count first, then the larger length (insertion sort: a few entries)
      const LenFreq x = h[a];
      int b = a - 1;
      while (b >= 0 && (h[b].freq < x.freq || (h[b].freq == x.freq && h[b].length < x.length))) { h[b + 1] = h[b]; b--; }
      h[b + 1] = x;
    }
    o.modal_len = h[0].length; o.modal_freq = h[0].freq;
    groups[g] = o;
  }
}

extern "C" long tjamd_context_histograms (tjamd_counter *c, int max_distance_per_flank, int levenshtein_distance, int *group_of, int *join_type,
                                          tjamd_context_group *groups, tjamd_length_freq *hist, long capacity)
{
  if (!c || max_distance_per_flank < 0) return -set_err (TJAMD_ERR_ARG, "bad arguments");
  if (c->status < 0) return -set_err (TJAMD_ERR_STATE, "tjamd_context_histograms needs a finalised counter");
  if (hipSetDevice (c->device) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "hipSetDevice failed");
  const long n = c->n_kept;
  if (n == 0) return 0;
  static_assert (sizeof (CtxGroupOut) == sizeof (tjamd_context_group) && sizeof (LenFreq) == sizeof (tjamd_length_freq), "group layout");
  int rc = ensure (c->headpos, (size_t) n * 4, c->stream);          // back[]
  if (!rc) rc = ensure (c->flags, (size_t) n * 4, c->stream);      // head flags
  if (!rc) rc = ensure (c->outpos, (size_t) n * 4, c->stream);     // groups before each element
  if (!rc) rc = ensure (c->segid, (size_t) n * 4, c->stream);      // group_of
  if (!rc) rc = ensure (c->keep, (size_t) n * 4, c->stream);       // candidates of the retry
  if (!rc) rc = ensure (c->grp_jt, (size_t) n * 4, c->stream);
  if (!rc) rc = ensure (c->grp_hist, (size_t) n * sizeof (LenFreq), c->stream);
  if (!rc) rc = ensure (c->scan_tmp, scan_tmp_words (n) * 4 + 64, c->stream);
  if (!rc) rc = ensure (c->alt, (size_t) n * sizeof (CtxGroupOut), c->stream);
  if (rc) return -rc;
  const u64 *kept = (const u64 *) c->kept.p;
  u32 *total = (u32 *) c->scan_tmp.p + scan_tmp_words (n), *n_cand = total + 1;
  hipLaunchKernelGGL (group_back_kernel, dim3 (grid_for (n)), dim3 (256), 0, c->stream, kept, n, max_distance_per_flank, (int *) c->headpos.p);
  hipLaunchKernelGGL (group_resolve_kernel, dim3 (grid_for (n)), dim3 (256), 0, c->stream, (const int *) c->headpos.p, n, (u32 *) c->flags.p);
  if (hipMemsetAsync (n_cand, 0, 4, c->stream) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "memset failed");
  // (which reading of the absent edit distance's last argument: looked up per call -- tests switch it)
  const char *ed = getenv ("TATAJUBA_AMD_EDIT_DISTANCE");
  const int free_end = (ed && !strcmp (ed, "free_end")) ? 1 : 0;
  hipLaunchKernelGGL (group_speculate_kernel, dim3 (grid_for (n)), dim3 (256), 0, c->stream, kept, n, c->k, levenshtein_distance, free_end,
                      (const u32 *) c->flags.p, (u32 *) c->keep.p, (int *) c->grp_jt.p, n_cand);
  u32 nc = 0;
  if (hipMemcpyAsync (&nc, n_cand, 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize (c->stream) != hipSuccess)
    return -set_err (TJAMD_ERR_HIP, "grouping failed: %s", hipGetErrorString (hipGetLastError ()));
  if (nc) hipLaunchKernelGGL (group_repair_kernel, dim3 (1), dim3 (256), 0, c->stream, kept, n, c->k, max_distance_per_flank, levenshtein_distance, free_end,
                              (u32 *) c->flags.p, (const u32 *) c->keep.p, (int *) c->grp_jt.p);
  rc = exclusive_scan (c, (const u32 *) c->flags.p, (u32 *) c->outpos.p, n, total, (u32 *) c->scan_tmp.p, scan_tmp_words (n));
  if (rc) return -rc;
  hipLaunchKernelGGL (group_histogram_kernel, dim3 (grid_for (n)), dim3 (256), 0, c->stream, kept, n, (const u32 *) c->flags.p, (const u32 *) c->outpos.p,
                      (int *) c->grp_jt.p, (int *) c->segid.p, (CtxGroupOut *) c->alt.p, (LenFreq *) c->grp_hist.p);
  if (hipGetLastError () != hipSuccess) return -set_err (TJAMD_ERR_HIP, "grouping launch failed");
  u32 ng = 0;
  if (hipMemcpyAsync (&ng, total, 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize (c->stream) != hipSuccess)
    return -set_err (TJAMD_ERR_HIP, "grouping failed: %s", hipGetErrorString (hipGetLastError ()));
  if ((long) ng > capacity && groups) return -set_err (TJAMD_ERR_CAPACITY, "%u groups, caller capacity %ld", ng, capacity);
  if (group_of && hipMemcpy (group_of, c->segid.p, (size_t) n * 4, hipMemcpyDeviceToHost) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "copy failed");
  if (join_type && hipMemcpy (join_type, c->grp_jt.p, (size_t) n * 4, hipMemcpyDeviceToHost) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "copy failed");
  if (groups && hipMemcpy (groups, c->alt.p, (size_t) ng * sizeof (CtxGroupOut), hipMemcpyDeviceToHost) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "copy failed");
  if (hist && hipMemcpy (hist, c->grp_hist.p, (size_t) n * sizeof (LenFreq), hipMemcpyDeviceToHost) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "copy failed");
  return (long) ng;
}

// Tract ids on a merged union (reference: src/genome_set.c:207-221: the id goes up wherever two neighbours of the
// concatenated list do not overlap; context-keyed: wherever (base, ctx0, ctx1) changes).  d_keys: tjamd_record[n] in the
// reference's descending order, as tjamd_merge_samples writes them.
__global__ void tract_head_kernel (const u64 *__restrict__ keys, long n, u32 *__restrict__ head)
{
  for (long i = blockIdx.x * (long) blockDim.x + threadIdx.x; i < n; i += (long) gridDim.x * blockDim.x)
    head[i] = (i > 0 && (keys[3 * i] != keys[3 * (i - 1)] || keys[3 * i + 1] != keys[3 * (i - 1) + 1] || ((keys[3 * i + 2] ^ keys[3 * (i - 1) + 2]) & 3ull))) ? 1u : 0u;
}
__global__ void tract_id_kernel (const u32 *__restrict__ excl, const u32 *__restrict__ head, long n, int *__restrict__ id)
{
  for (long i = blockIdx.x * (long) blockDim.x + threadIdx.x; i < n; i += (long) gridDim.x * blockDim.x) id[i] = (int) (excl[i] + head[i]);
}

extern "C" long tjamd_tract_ids (tjamd_counter *c, const void *d_keys, long n, int *d_tract_id, int *h_tract_id)
{
  if (!c || n < 0 || (n && !d_keys)) return -set_err (TJAMD_ERR_ARG, "bad arguments");
  if (hipSetDevice (c->device) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "hipSetDevice failed");
  if (n == 0) return 0;
  int rc = ensure (c->flags, (size_t) n * 4, c->stream);
  if (!rc) rc = ensure (c->outpos, (size_t) n * 4, c->stream);
  if (!rc) rc = ensure (c->segid, (size_t) n * 4, c->stream);
  if (!rc) rc = ensure (c->scan_tmp, scan_tmp_words (n) * 4 + 64, c->stream);
  if (rc) return -rc;
  u32 *total = (u32 *) c->scan_tmp.p + scan_tmp_words (n);
  int *ids = d_tract_id ? d_tract_id : (int *) c->segid.p;
  hipLaunchKernelGGL (tract_head_kernel, dim3 (grid_for (n)), dim3 (256), 0, c->stream, (const u64 *) d_keys, n, (u32 *) c->flags.p);
  rc = exclusive_scan (c, (const u32 *) c->flags.p, (u32 *) c->outpos.p, n, total, (u32 *) c->scan_tmp.p, scan_tmp_words (n));
  if (rc) return -rc;
  hipLaunchKernelGGL (tract_id_kernel, dim3 (grid_for (n)), dim3 (256), 0, c->stream, (const u32 *) c->outpos.p, (const u32 *) c->flags.p, n, ids);
  if (hipGetLastError () != hipSuccess) return -set_err (TJAMD_ERR_HIP, "tract id launch failed");
  u32 nh = 0;
  if (hipMemcpyAsync (&nh, total, 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize (c->stream) != hipSuccess)
    return -set_err (TJAMD_ERR_HIP, "tract ids failed: %s", hipGetErrorString (hipGetLastError ()));
  if (h_tract_id && hipMemcpy (h_tract_id, ids, (size_t) n * 4, hipMemcpyDeviceToHost) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "copy failed");
  return (long) nh + 1;
}

// Peer access between two devices of this process, asked for once per ordered pair: with it hipMemcpyPeerAsync moves the
// bytes over xGMI directly, without it the runtime stages them through host memory.  Returns 1 direct, 0 staged;
// tjamd_peer_access_report () says which pairs are which.
static std::mutex g_peer_mutex;
static signed char g_peer[64][64];                      // 0 unknown, 1 direct, -1 refused
static int peer_access (int dst_dev, int src_dev)
{
  if (dst_dev == src_dev) return 1;
  if (dst_dev < 0 || src_dev < 0 || dst_dev >= 64 || src_dev >= 64) return 0;
  std::lock_guard<std::mutex> lock (g_peer_mutex);
  if (g_peer[dst_dev][src_dev]) return g_peer[dst_dev][src_dev] > 0;
  int can = 0, prev = -1;
  (void) hipGetDevice (&prev);
  if (hipDeviceCanAccessPeer (&can, dst_dev, src_dev) == hipSuccess && can && hipSetDevice (dst_dev) == hipSuccess) {
    const hipError_t e = hipDeviceEnablePeerAccess (src_dev, 0);
    can = (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled);
    (void) hipGetLastError ();
  }
  else can = 0;
  if (prev >= 0) (void) hipSetDevice (prev);
  g_peer[dst_dev][src_dev] = can ? 1 : -1;
  return can;
}

extern "C" int tjamd_peer_access_report (char *out, int capacity)
{ // "dst<-src:direct" / "dst<-src:staged" for every pair a gather has used so far; returns the number of staged pairs
  int staged = 0, at = 0;
  std::lock_guard<std::mutex> lock (g_peer_mutex);
  if (out && capacity > 0) out[0] = 0;
  for (int d = 0; d < 64; d++)
    for (int sdev = 0; sdev < 64; sdev++)
      if (g_peer[d][sdev]) {
        if (g_peer[d][sdev] < 0) staged++;
        if (out && at < capacity - 24) at += snprintf (out + at, (size_t) (capacity - at), "%d<-%d:%s ", d, sdev, g_peer[d][sdev] > 0 ? "direct" : "staged");
      }
  return staged;
}

// The exchange of the cross-sample merge for a caller that, like the reference, runs its samples as threads of ONE
// process (src/genome_set.c:66-94; merge at :195-229): the kept records of the finalised counters `samples` -- each on
// the device its thread bound it to -- are copied into one buffer on dst's device (peer copies over xGMI between
// devices, a plain device copy on the same one), back to back in sample order.  counts[i] = records of sample i;
// *d_records = the buffer (owned by dst, valid until its next gather).  Returns the total, ready for
// tjamd_merge_samples (dst, *d_records, counts, n_samples, ...).
extern "C" long tjamd_gather_histograms (tjamd_counter *dst, tjamd_counter *const *samples, int n_samples, const void **d_records, long *counts)
{
  if (!dst || !samples || n_samples < 1 || !d_records || !counts) return -set_err (TJAMD_ERR_ARG, "bad arguments");
  long n = 0;
  for (int i = 0; i < n_samples; i++) {
    if (!samples[i] || samples[i]->status < 0) return -set_err (TJAMD_ERR_STATE, "sample %d is not finalised", i);
    counts[i] = samples[i]->n_kept; n += counts[i];
  }
  if (hipSetDevice (dst->device) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "hipSetDevice failed");
  int rc = ensure (dst->rawlist, (size_t) std::max<long> (n, 1) * 24, dst->stream);
  if (rc) return -rc;
  size_t off = 0;
  for (int i = 0; i < n_samples; i++) {
    const size_t bytes = (size_t) counts[i] * 24;
    if (!bytes) continue;
    tjamd_counter *s = samples[i];
    if (s->device != dst->device) peer_access (dst->device, s->device);   // (direct xGMI copies; staged through the host by the runtime where it is refused)
    hipError_t e = (s->device == dst->device)
      ? hipMemcpyAsync ((char *) dst->rawlist.p + off, s->kept.p, bytes, hipMemcpyDeviceToDevice, dst->stream)
      : hipMemcpyPeerAsync ((char *) dst->rawlist.p + off, dst->device, s->kept.p, s->device, bytes, dst->stream);
    if (e != hipSuccess) return -set_err (TJAMD_ERR_HIP, "gather copy of sample %d failed: %s", i, hipGetErrorString (e));
    off += bytes;
  }
  if (hipStreamSynchronize (dst->stream) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "gather failed");
  *d_records = dst->rawlist.p;
  return n;
}


extern "C" long tjamd_merge_samples (tjamd_counter *c, const void *d_records, const long *counts, int n_samples,
                                      void *d_out_keys, void *d_out_counts, long capacity)
{
  if (!c || !counts || n_samples < 1 || n_samples > 4096 || !d_out_keys || !d_out_counts) return -set_err (TJAMD_ERR_ARG, "bad arguments");
  if (hipSetDevice (c->device) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "hipSetDevice failed");
  std::vector<long> starts ((size_t) n_samples + 1);
  long n = 0;
  for (int i = 0; i < n_samples; i++) { if (counts[i] < 0) return -set_err (TJAMD_ERR_ARG, "negative count"); starts[i] = n; n += counts[i]; }
  starts[n_samples] = n;
  if (n == 0) return 0;
  if (!d_records) return -set_err (TJAMD_ERR_ARG, "null records");
  if (n >= (1l << 31)) return -set_err (TJAMD_ERR_CAPACITY, "%ld records to merge", n);

  // bins of ~32..64 records: partition, then one wavefront per bin finds the distinct keys, their order and depths
  int nbits = 6;
  while (nbits < BS_MAXBITS && nbits < 1 + 4 * c->k && (24l << nbits) < n) nbits++;   // (the per-bin work is quadratic in the bin's size: bins of ~30 records)
  nbits = std::min (nbits, 1 + 4 * c->k);
  const int nbins = 1 << nbits;
  int rc = ensure (c->prefix, (size_t) (n_samples + 1) * 8, c->stream);
  if (!rc) rc = ensure (c->alt, (size_t) n * 24, c->stream);
  if (!rc && !c->bins.p) { rc = ensure (c->bins, (size_t) (BS_MAXBINS + 1) * 4, c->stream); c->bins_zeroed = false; }
  if (!rc) rc = ensure (c->binstart, (size_t) (BS_MAXBINS + 1) * 4, c->stream);
  if (!rc) rc = ensure (c->binctx, (size_t) BS_MAXBINS * 8, c->stream);
  if (!rc) rc = ensure (c->headpos, (size_t) n * 4, c->stream);
  if (!rc) rc = ensure (c->outpos, (size_t) n * 4, c->stream);
  if (rc) return -rc;
  if (hipMemcpyAsync (c->prefix.p, starts.data (), (size_t) (n_samples + 1) * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
      hipStreamSynchronize (c->stream) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "copy failed");
  u32 *bins = (u32 *) c->bins.p, *binstart = (u32 *) c->binstart.p, *binctx = (u32 *) c->binctx.p, *binout = binctx + nbins;
  u32 *tpos = (u32 *) c->headpos.p, *ttot = (u32 *) c->outpos.p;
  if (!c->bins_zeroed && hipMemsetAsync (bins, 0, (size_t) BS_MAXBINS * 4, c->stream) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "memset failed");
  c->bins_zeroed = false;
  (void) hipEventRecord (c->ev_m0, c->stream);
  hipLaunchKernelGGL (bin_count_kernel, dim3 (grid_for (n)), dim3 (256), 0, c->stream, (const u64 *) d_records, n, c->k, nbits, bins, (uint4 *) nullptr, 0l, 1);
  hipLaunchKernelGGL (bin_scan_kernel, dim3 (1), dim3 (1024), 0, c->stream, bins, nbins, binstart,
                      c->bin_rank_max < (u32) BS_RANK_MAX ? c->bin_rank_max : (u32) MG_RANK_MAX, c->d_fin);   // (test hook: see TATAJUBA_AMD_BIN_MAX)
  hipLaunchKernelGGL (merge_scatter_kernel, dim3 (grid_for (n)), dim3 (256), 0, c->stream, (const u64 *) d_records, (u64 *) c->alt.p, n, c->k, nbits, bins,
                      (const long *) c->prefix.p, n_samples);
  hipLaunchKernelGGL (bin_merge_kernel, dim3 ((unsigned) std::min (nbins, 16384)), dim3 (64), 0, c->stream, (const u64 *) c->alt.p, (const u32 *) binstart, nbins,
                      (const FinCounts *) c->d_fin, binctx, tpos, ttot);
  hipLaunchKernelGGL (bin_ctx_scan_kernel, dim3 (1), dim3 (1024), 0, c->stream, (const u32 *) binctx, nbins, binout, c->d_fin);
  hipLaunchKernelGGL (bin_merge_write_kernel, dim3 ((unsigned) std::min (nbins / 4 + 1, 4096)), dim3 (256), 0, c->stream, (const u64 *) c->alt.p, (const u32 *) binstart,
                      (const u32 *) binout, nbins, (const FinCounts *) c->d_fin, (const u32 *) tpos, (const u32 *) ttot, n_samples,
                      (u64 *) d_out_keys, (int *) d_out_counts, capacity);
  if (hipGetLastError () != hipSuccess) return -set_err (TJAMD_ERR_HIP, "merge launch failed");
  (void) hipEventRecord (c->ev_m1, c->stream);
  c->merge_timed = true;
  if (hipMemcpyAsync (c->h_fin, c->d_fin, sizeof (FinCounts), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
      hipStreamSynchronize (c->stream) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "merge failed: %s", hipGetErrorString (hipGetLastError ()));
  if (c->h_fin->sort_fallback) return merge_samples_radix (c, d_records, n, n_samples, d_out_keys, d_out_counts, capacity);
  const long n_union = c->h_fin->n_idx;
  if (n_union > capacity) return -set_err (TJAMD_ERR_CAPACITY, "%ld union keys, caller capacity %ld", n_union, capacity);
  return n_union;
}

static long merge_samples_radix (tjamd_counter *c, const void *d_records, long n, int n_samples, void *d_out_keys, void *d_out_counts, long capacity)
{
  int rc = ensure (c->rawlist, (size_t) n * 24, c->stream);
  if (!rc) rc = ensure (c->alt, (size_t) n * 24, c->stream);
  if (!rc) rc = ensure (c->flags, (size_t) n * 4, c->stream);
  if (!rc) rc = ensure (c->segid, (size_t) n * 4, c->stream);
  if (!rc) rc = ensure (c->keep, (size_t) n * 8, c->stream);          // totals per union key, then their flags
  if (!rc) rc = ensure (c->scan_tmp, scan_tmp_words (n) * 4, c->stream);
  if (rc) return -rc;
  hipLaunchKernelGGL (merge_tag_kernel, dim3 (grid_for (n)), dim3 (256), 0, c->stream, (const u64 *) d_records, (u64 *) c->rawlist.p, n, (const long *) c->prefix.p, n_samples);
  u64 *a = (u64 *) c->rawlist.p, *b = (u64 *) c->alt.p;
  rc = radix_sort_records (c, a, b, n);               // the reference's order; the sample tag does not take part
  if (rc) return -rc;
  if (a != (u64 *) c->rawlist.p) std::swap (c->rawlist, c->alt);
  u32 *flags = (u32 *) c->flags.p, *segid = (u32 *) c->segid.p;
  hipLaunchKernelGGL (seg_heads_kernel, dim3 (grid_for (n)), dim3 (256), 0, c->stream, (const u64 *) a, n, flags, 0);
  rc = exclusive_scan (c, flags, segid, n, &c->d_fin->n_seg, (u32 *) c->scan_tmp.p, c->scan_tmp.cap / 4);
  if (rc) return -rc;
  if (hipMemcpyAsync (c->h_fin, c->d_fin, sizeof (FinCounts), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
      hipStreamSynchronize (c->stream) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "merge failed: %s", hipGetErrorString (hipGetLastError ()));
  const long n_union = c->h_fin->n_seg;
  if (n_union > capacity) return -set_err (TJAMD_ERR_CAPACITY, "%ld union keys, caller capacity %ld", n_union, capacity);
  if (hipMemsetAsync (d_out_counts, 0, (size_t) n_union * n_samples * 4, c->stream) != hipSuccess ||
      hipMemsetAsync (c->keep.p, 0, (size_t) n * 8, c->stream) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "memset failed");
  hipLaunchKernelGGL (merge_write_kernel, dim3 (grid_for (n)), dim3 (256), 0, c->stream, (const u64 *) a, n, (const u32 *) flags, (const u32 *) segid,
                      n_samples, (u64 *) d_out_keys, (int *) d_out_counts, (u32 *) c->keep.p, (u32 *) c->keep.p + n, capacity);
  hipLaunchKernelGGL (merge_totals_kernel, dim3 (grid_for (n_union)), dim3 (256), 0, c->stream, (u64 *) d_out_keys, (const u32 *) c->keep.p,
                      (const u32 *) c->keep.p + n, (const u32 *) &c->d_fin->n_seg, capacity);
  if (hipGetLastError () != hipSuccess || hipStreamSynchronize (c->stream) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "merge kernels failed");
  return n_union;
}


// ---------------------------------------------------------------------------------------------------------------
// The exchange between PROCESSES, one per GPU: an all-gatherv of the finalised counters' kept records over RCCL (xGMI on
// the node), so that every rank holds every sample's histogram for the cross-sample merge (reference attach point:
// src/genome_set.c:195-229 -- there the samples are threads of one process and the "exchange" is a pointer).
// RCCL has no all-gatherv: every rank contributes one block of the same size -- a header with its record count, then its
// records -- to ONE ncclAllGather on the counter's stream, and a kernel packs the received blocks back to back.  The block
// size is agreed without talking: it is a function of the counts of the exchange before (the same numbers on every
// rank); the first exchange, and one in which some rank holds more than a block takes, learn the counts with an 8-byte
// all-gather first.  One host synchronisation per exchange (the counts the caller gets back), on the usual path.

struct tjamd_comm
{
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  long cap = 0;                                         // records per block, 0 = not agreed yet
  hipStream_t stream = nullptr;                         // tjamd_comm_set_stream; null: the stream of the counter being exchanged
  DevBuf send, recv, out, dcounts;
  long *h_counts = nullptr;                             // pinned, world entries
  long exchanges = 0, collectives = 0;
  // the last exchange, for whoever reports on it (bench.py): device time from the first pack to the last unpack (HIP events on
  // the exchange's stream), bytes received, collectives it took (1 when the block size was settled, 2 or more otherwise)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double last_ms = -1.0; long last_bytes = 0, last_collectives = 0;
};

// RCCL is looked up when the first communicator call needs it (dlopen), not when the library is loaded: the one-GPU drop-in
// has no use for it and must load on a machine without it.  <rccl/rccl.h> gives the types; the seven entry points used:
struct RcclApi
{
  const char *(*GetErrorString) (ncclResult_t);
  ncclResult_t (*GetUniqueId) (ncclUniqueId *);
  ncclResult_t (*CommInitRank) (ncclComm_t *, int, ncclUniqueId, int);
  ncclResult_t (*CommDestroy) (ncclComm_t);
  ncclResult_t (*CommCount) (const ncclComm_t, int *);
  ncclResult_t (*CommAbort) (ncclComm_t);
  ncclResult_t (*AllGather) (const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
};

static const RcclApi *rccl_api ()
{ // TATAJUBA_AMD_RCCL names the library; otherwise the loader's search path, then $ROCM_PATH/lib (default /opt/rocm/lib)
  static RcclApi api;
  static bool ok = false;
  static char why[512] = "";
  static std::once_flag once;
  std::call_once (once, [] () {
    void *h = nullptr;
    std::string tried;
    std::vector<std::string> names;
    if (const char *e = getenv ("TATAJUBA_AMD_RCCL")) names.push_back (e);
    else {
      names.push_back ("librccl.so.1"); names.push_back ("librccl.so");
      const char *rp = getenv ("ROCM_PATH");
      const std::string root = (rp && *rp) ? rp : "/opt/rocm";
      names.push_back (root + "/lib/librccl.so.1"); names.push_back (root + "/lib/librccl.so");
    }
    for (const std::string &nm : names) {
      h = dlopen (nm.c_str (), RTLD_NOW | RTLD_LOCAL);
      if (h) break;
      tried += (tried.empty () ? "" : ", ") + nm;
    }
    if (!h) { snprintf (why, sizeof why, "RCCL not found (tried %s): the exchange between processes needs it", tried.c_str ()); return; }
    struct { const char *name; void **to; } syms[] = {
      {"ncclGetErrorString", (void **) &api.GetErrorString}, {"ncclGetUniqueId", (void **) &api.GetUniqueId},
      {"ncclCommInitRank", (void **) &api.CommInitRank}, {"ncclCommDestroy", (void **) &api.CommDestroy},
      {"ncclCommCount", (void **) &api.CommCount}, {"ncclCommAbort", (void **) &api.CommAbort}, {"ncclAllGather", (void **) &api.AllGather}};
    for (auto &sy : syms) {
      *sy.to = dlsym (h, sy.name);
      if (!*sy.to) { snprintf (why, sizeof why, "RCCL: %s not found in the library that was loaded", sy.name); return; }
    }
    ok = true;
  });
  if (!ok) { set_err (TJAMD_ERR_HIP, "%s", why); return nullptr; }
  return &api;
}

#define NCCLCHK(name, call, ret) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { \
  set_err (TJAMD_ERR_HIP, "%s failed: %s", name, NC->GetErrorString (r_)); return ret; } } while (0)

extern "C" int tjamd_comm_unique_id (void *id_bytes)
{
  static_assert (TJAMD_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
  if (!id_bytes) return set_err (TJAMD_ERR_ARG, "tjamd_comm_unique_id: null buffer");
  const RcclApi *NC = rccl_api ();
  if (!NC) return TJAMD_ERR_HIP;
  ncclUniqueId id;
  NCCLCHK ("ncclGetUniqueId", NC->GetUniqueId (&id), TJAMD_ERR_HIP);
  memcpy (id_bytes, &id, sizeof id);
  return TJAMD_OK;
}

extern "C" tjamd_comm *tjamd_comm_create (tjamd_counter *c, const void *id_bytes, int rank, int world)
{
  if (!c || !id_bytes || world < 1 || world > 4096 || rank < 0 || rank >= world) { set_err (TJAMD_ERR_ARG, "tjamd_comm_create: bad arguments"); return NULL; }
  const RcclApi *NC = rccl_api ();
  if (!NC) return NULL;
  HIPCHK_NULL (hipSetDevice (c->device));
  tjamd_comm *m = new (std::nothrow) tjamd_comm;
  if (!m) { set_err (TJAMD_ERR_HIP, "out of memory"); return NULL; }
  m->rank = rank; m->world = world; m->device = c->device;
  ncclUniqueId id;
  memcpy (&id, id_bytes, sizeof id);
  ncclResult_t r = NC->CommInitRank (&m->comm, world, id, rank);
  if (r != ncclSuccess) { set_err (TJAMD_ERR_HIP, "ncclCommInitRank failed: %s", NC->GetErrorString (r)); delete m; return NULL; }
  if (hipHostMalloc ((void **) &m->h_counts, (size_t) world * sizeof (long), hipHostMallocDefault) != hipSuccess ||
      hipEventCreate (&m->ev0) != hipSuccess || hipEventCreate (&m->ev1) != hipSuccess) {
    set_err (TJAMD_ERR_HIP, "hipHostMalloc / hipEventCreate failed"); (void) NC->CommDestroy (m->comm); delete m; return NULL;
  }
  return m;
}

extern "C" void tjamd_comm_destroy (tjamd_comm *m)
{
  if (!m) return;
  (void) hipSetDevice (m->device);
  if (m->comm) if (const RcclApi *NC = rccl_api ()) (void) NC->CommDestroy (m->comm);
  release (m->send); release (m->recv); release (m->out); release (m->dcounts);
  if (m->h_counts) (void) hipHostFree (m->h_counts);
  if (m->ev0) (void) hipEventDestroy (m->ev0);
  if (m->ev1) (void) hipEventDestroy (m->ev1);
  delete m;
}

extern "C" int tjamd_comm_set_stream (tjamd_comm *m, void *hip_stream)
{
  if (!m) return set_err (TJAMD_ERR_ARG, "tjamd_comm_set_stream: null communicator");
  m->stream = (hipStream_t) hip_stream;
  return TJAMD_OK;
}
extern "C" int tjamd_comm_rank (const tjamd_comm *m) { return m ? m->rank : -1; }
extern "C" int tjamd_comm_world (const tjamd_comm *m) { return m ? m->world : -1; }
// the number of ranks RCCL itself says the communicator has (ncclCommCount), not the argument tjamd_comm_create was given
extern "C" int tjamd_comm_count (const tjamd_comm *m)
{
  int n = -1;
  const RcclApi *NC = (m && m->comm) ? rccl_api () : nullptr;
  if (!NC || NC->CommCount (m->comm, &n) != ncclSuccess) return -1;
  return n;
}
// the last exchange on this communicator: device milliseconds (pack, collective(s), unpack), bytes the data collective
// delivered to this rank, collectives it took.  Any pointer may be NULL.  Returns 0, or non-zero before the first exchange.
extern "C" int tjamd_comm_last_exchange (const tjamd_comm *m, double *ms, long *bytes, long *collectives)
{
  if (!m || m->exchanges == 0 || m->last_collectives == 0) return 1;
  if (ms) *ms = m->last_ms;
  if (bytes) *bytes = m->last_bytes;
  if (collectives) *collectives = m->last_collectives;
  return 0;
}
extern "C" long tjamd_comm_collectives (const tjamd_comm *m) { return m ? m->collectives : -1; }

#define GX_HEADER 16                                    // bytes in front of a block's records: the record count, then padding

__global__ void gx_pack_kernel (const u64 *__restrict__ kept, long n, long cap, u64 *__restrict__ block)
{ // header + the first min (n, cap) records
  const long words = 3 * min (n, cap);
  if (blockIdx.x == 0 && threadIdx.x == 0) { block[0] = (u64) n; block[1] = 0; }
  for (long i = blockIdx.x * (long) blockDim.x + threadIdx.x; i < words; i += (long) gridDim.x * blockDim.x) block[2 + i] = kept[i];
}

__global__ void gx_unpack_kernel (const u64 *__restrict__ recv, long block_words, int world, long cap, u64 *__restrict__ out, long *__restrict__ counts)
{ // blocks -> records back to back in rank order; counts[r] = what rank r holds (may exceed cap: the caller looks)
  for (int r = blockIdx.y; r < world; r += gridDim.y) {
    long before = 0;
    for (int q = 0; q < r; q++) before += min ((long) recv[(long) q * block_words], cap);
    const u64 *src = recv + (long) r * block_words;
    const long n = (long) src[0], words = 3 * min (n, cap);
    if (blockIdx.x == 0 && threadIdx.x == 0) counts[r] = n;
    for (long i = blockIdx.x * (long) blockDim.x + threadIdx.x; i < words; i += (long) gridDim.x * blockDim.x) out[3 * before + i] = src[2 + i];
  }
}

static long gx_block_cap (long max_count)
{ // what a block takes after an exchange whose fullest rank held max_count: a quarter more, in steps of 4096 records
  const long want = max_count + max_count / 4 + 1;
  return ((want + 4095) / 4096) * 4096;
}

// Test hook (not in the public header): the exchange's pack and unpack kernels on fabricated data -- what the ranks r > 0 of
// a communicator would have sent -- on one GPU.  samples[r] / n[r]: rank r's kept records (host, 3 words each); every rank's
// block is packed with gx_pack_kernel for a block of `cap` records and the blocks are unpacked as tjamd_allgather_histograms
// unpacks what ncclAllGather delivers.  out: world * cap records (host); counts[r] = what rank r says it holds (n[r], also
// when it exceeds cap: the caller's cue to agree on a larger block).  Returns the records unpacked (sum of min (n[r], cap)).
extern "C" long tjamd_debug_exchange_pack_unpack (tjamd_counter *c, const u64 *const *samples, const long *n, int world, long cap, u64 *out, long *counts)
{
  if (!c || !samples || !n || world < 1 || cap < 1 || !out || !counts) return -set_err (TJAMD_ERR_ARG, "bad arguments");
  if (hipSetDevice (c->device) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "hipSetDevice failed");
  const long block_words = GX_HEADER / 8 + 3 * cap;
  u64 *d_kept = nullptr, *d_recv = nullptr, *d_out = nullptr; long *d_counts = nullptr;
  long total = 0, mx = 1;
  for (int r = 0; r < world; r++) mx = std::max (mx, n[r]);
  if (hipMalloc ((void **) &d_kept, (size_t) mx * 24) != hipSuccess || hipMalloc ((void **) &d_recv, (size_t) block_words * 8 * (size_t) world) != hipSuccess ||
      hipMalloc ((void **) &d_out, (size_t) cap * world * 24) != hipSuccess || hipMalloc ((void **) &d_counts, (size_t) world * sizeof (long)) != hipSuccess)
    return -set_err (TJAMD_ERR_HIP, "hipMalloc failed");
  for (int r = 0; r < world; r++) {
    if (n[r] && hipMemcpyAsync (d_kept, samples[r], (size_t) n[r] * 24, hipMemcpyHostToDevice, c->stream) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "copy failed");
    hipLaunchKernelGGL (gx_pack_kernel, dim3 (grid_for (3 * std::min (n[r], cap) + 1)), dim3 (256), 0, c->stream, (const u64 *) d_kept, n[r], cap, d_recv + (size_t) r * block_words);
    if (hipStreamSynchronize (c->stream) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "pack failed");
    total += std::min (n[r], cap);
  }
  hipLaunchKernelGGL (gx_unpack_kernel, dim3 (grid_for (3 * cap + 1), (unsigned) std::min (world, 64)), dim3 (256), 0, c->stream, (const u64 *) d_recv, block_words, world, cap, d_out, d_counts);
  if (hipMemcpyAsync (out, d_out, (size_t) total * 24, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
      hipMemcpyAsync (counts, d_counts, (size_t) world * sizeof (long), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
      hipStreamSynchronize (c->stream) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "unpack failed: %s", hipGetErrorString (hipGetLastError ()));
  (void) hipFree (d_kept); (void) hipFree (d_recv); (void) hipFree (d_out); (void) hipFree (d_counts);
  return total;
}

extern "C" long tjamd_allgather_histograms (tjamd_counter *c, tjamd_comm *m, const void **d_records, long *counts)
{
  if (!c || !m || !d_records || !counts) return -set_err (TJAMD_ERR_ARG, "tjamd_allgather_histograms: bad arguments");
  if (c->status < 0) return -set_err (TJAMD_ERR_STATE, "tjamd_allgather_histograms needs a finalised counter");
  if (c->device != m->device) return -set_err (TJAMD_ERR_ARG, "the communicator was made for device %d, the counter lives on %d", m->device, c->device);
  if (hipSetDevice (c->device) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "hipSetDevice failed");
  if (!m->comm) return -set_err (TJAMD_ERR_STATE, "the communicator was aborted after a failed exchange");
  const RcclApi *NC = rccl_api ();                      // (there: the communicator was made through it)
  if (!NC) return -TJAMD_ERR_HIP;
  const int world = m->world;
  const long n_mine = c->n_kept;
  // A rank that gives up between two collectives would leave its peers waiting in theirs for ever: every failure from
  // here on aborts the communicator (ncclCommAbort: the peers' pending and later calls fail instead of hanging).
  auto give_up = [&] (int code) -> long { if (m->comm) { (void) NC->CommAbort (m->comm); m->comm = nullptr; } return -(long) code; };
  // (a stream of the communicator's own: the exchange of a finalised sample runs beside the scan of the next one; the
  // caller has seen this counter's finalise end -- tjamd_finalise / tjamd_finalise_end -- so its kept records are complete)
  const hipStream_t st = m->stream ? m->stream : c->stream;
  int rc = ensure (m->dcounts, (size_t) (world + 1) * sizeof (long), st);
  if (rc) return give_up (rc);
  long *d_counts = (long *) m->dcounts.p;
  m->exchanges++;
  const long collectives_before = m->collectives;
  (void) hipEventRecord (m->ev0, st);
  for (int attempt = 0; attempt < 3; attempt++) {
    if (m->cap == 0) {                                  // block size not agreed (first exchange, or the last one overflowed): the counts first
      if (hipMemcpyAsync (d_counts + world, &n_mine, sizeof (long), hipMemcpyHostToDevice, st) != hipSuccess) return give_up (set_err (TJAMD_ERR_HIP, "copy failed"));
      NCCLCHK ("ncclAllGather (counts)", NC->AllGather (d_counts + world, d_counts, sizeof (long), ncclChar, m->comm, st), -TJAMD_ERR_HIP);
      m->collectives++;
      if (hipMemcpyAsync (m->h_counts, d_counts, (size_t) world * sizeof (long), hipMemcpyDeviceToHost, st) != hipSuccess ||
          hipStreamSynchronize (st) != hipSuccess) return give_up (set_err (TJAMD_ERR_HIP, "exchange of the counts failed: %s", hipGetErrorString (hipGetLastError ())));
      long mx = 0;
      for (int r = 0; r < world; r++) mx = std::max (mx, m->h_counts[r]);
      m->cap = gx_block_cap (mx);
    }
    const long cap = m->cap, block_words = GX_HEADER / 8 + 3 * cap;
    rc = ensure (m->send, (size_t) block_words * 8, st);
    if (!rc) rc = ensure (m->recv, (size_t) block_words * 8 * (size_t) world, st);
    if (!rc) rc = ensure (m->out, (size_t) std::max<long> (cap * world, 1) * 24, st);
    if (rc) return give_up (rc);
    hipLaunchKernelGGL (gx_pack_kernel, dim3 (grid_for (3 * std::min (n_mine, cap) + 1)), dim3 (256), 0, st, (const u64 *) c->kept.p, n_mine, cap, (u64 *) m->send.p);
    NCCLCHK ("ncclAllGather", NC->AllGather (m->send.p, m->recv.p, (size_t) block_words * 8, ncclChar, m->comm, st), -TJAMD_ERR_HIP);
    m->collectives++;
    hipLaunchKernelGGL (gx_unpack_kernel, dim3 (grid_for (3 * cap + 1), (unsigned) std::min (world, 64)), dim3 (256), 0, st,
                        (const u64 *) m->recv.p, block_words, world, cap, (u64 *) m->out.p, d_counts);
    if (hipGetLastError () != hipSuccess) return give_up (set_err (TJAMD_ERR_HIP, "exchange launch failed"));
    (void) hipEventRecord (m->ev1, st);
    if (hipMemcpyAsync (m->h_counts, d_counts, (size_t) world * sizeof (long), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize (st) != hipSuccess) return give_up (set_err (TJAMD_ERR_HIP, "exchange failed: %s", hipGetErrorString (hipGetLastError ())));
    long mx = 0, total = 0;
    for (int r = 0; r < world; r++) { mx = std::max (mx, m->h_counts[r]); total += m->h_counts[r]; }
    if (mx > cap) { m->cap = 0; continue; }             // (every rank sees the same counts and takes the same turn)
    m->cap = gx_block_cap (mx);
    for (int r = 0; r < world; r++) counts[r] = m->h_counts[r];
    *d_records = m->out.p;
    {
      float ms = 0.f;
      m->last_ms = hipEventElapsedTime (&ms, m->ev0, m->ev1) == hipSuccess ? (double) ms : -1.0;
      m->last_bytes = block_words * 8 * (long) world;     // (what the data collective moved into this rank: every rank's max-padded block)
      m->last_collectives = m->collectives - collectives_before;
    }
    return total;
  }
  return -set_err (TJAMD_ERR_STATE, "the exchange did not settle on a block size");
}
