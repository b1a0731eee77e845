// hopo_device.hip -- HIP/CDNA4 (gfx950) kernels and the thin C-ABI layer (tjamd_*) of the homopolymer-tract engine.
//
// Replaces, on the device, tatajuba's per-read scan (reference: src/hopo_counter.c:219-258,285-307) and the per-sample
// sort / dedupe / filter / index / coverage (reference: src/hopo_counter.c:339-438).  Integer and byte work only:
// HBM-bound, no MFMA.  See DESIGN.md for the data layout and the roofline of each kernel.
//
// Scan kernel in one paragraph: a workgroup (256 threads = 4 wavefronts) owns a 4 KiB tile of the '\n'-delimited
// read stream.  Every lane loads 16 bytes (coalesced 1 KiB per wave instruction), classifies them with SWAR bit
// logic into 2-bit base codes plus three bit-planes (run start, read delimiter, non-ACGTU) and stores those in LDS
// (2 bits + 3 bits per base instead of 8).  Run starts that can begin a tract of >= m bases are found with shifted
// ANDs of the run-start plane, compacted into a workgroup-wide candidate list in LDS, and then one lane per
// candidate finds the run end (count-trailing-zeros on the plane), checks both flanks against the delimiter plane,
// pulls the two k-mers out of the packed codes with funnel shifts, canonicalises (reverse complement = bit reverse +
// pair swap of the complemented word) and appends a 24-byte record through a wave-aggregated atomic.

#include <hip/hip_runtime.h>
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
extern "C" const char *tjamd_version (void) { return "tatajuba_amd 0.1 (gfx950)"; }

extern "C" int tjamd_device_count (void)
{
  int n = 0;
  if (hipGetDeviceCount (&n) != hipSuccess) return 0;
  return n;
}

// ---------------------------------------------------------------------------------------------------------------
// geometry of the scan tile

#define TJ_BLOCK   256
#define TJ_TILE    4096                 // bytes of stream owned by one workgroup iteration
#define TJ_HL      64                   // left halo  (>= max k + 1, multiple of 32)
#define TJ_HR      192                  // right halo (tracts whose end + k stays inside are handled from LDS)
#define TJ_WIN     (TJ_HL + TJ_TILE + TJ_HR)     // 4352
#define TJ_NCHUNK  (TJ_WIN / 16)                 // 272 chunks of 16 bytes
#define TJ_MASKW   (TJ_WIN / 32 + 4)             // words per bit-plane (+ zeroed pad for 3-word funnel reads)
#define TJ_CODEW   (TJ_WIN / 16 + 4)             // words of 2-bit codes (+ pad)
#define TJ_MAXCAND (TJ_TILE / 2)                 // tracts have >= 2 bases, so at most one candidate per 2 bytes

struct DevCounters
{
  u64 n_rec;        // records appended to the output list
  u64 n_fix;        // pending non-ACGTU runs (see nrun_fixup_kernel)
  u64 n_undefined;  // qualifying non-ACGTU runs without an earlier tract in the read (reference: uninitialised memory)
  u32 overflow;     // output list too small
  u32 fix_overflow; // fix list too small
};

struct FixEntry { long long pos; long long len; };

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

// 64 bits of a little-endian bit array starting at bit position `bitpos` (array padded by >= 2 words)
__device__ __forceinline__ u64 bits64 (const u32 *a, int bitpos)
{
  int w = bitpos >> 5, sh = bitpos & 31;
  u32 x0 = a[w], x1 = a[w + 1], x2 = a[w + 2];
  u32 lo = __funnelshift_r (x0, x1, sh);
  u32 hi = __funnelshift_r (x1, x2, sh);
  return ((u64) hi << 32) | lo;
}

__device__ __forceinline__ u64 kmask (int k) { return (k >= 32) ? ~0ull : ((1ull << (2 * k)) - 1ull); }

// reverse complement of a k-mer packed first-base-lowest (reference: src/hopo_counter.c:241-242 builds it base by base)
__device__ __forceinline__ u64 revcomp_k (u64 x, int k)
{
  u64 y = __brevll (~x);
  y = ((y >> 1) & 0x5555555555555555ull) | ((y & 0x5555555555555555ull) << 1);
  return y >> (64 - 2 * k);
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
// scan kernel.  W = 3: 24-byte records; W = 4: located records (adds the stream position of the tract's first base).

template <int W>
__global__ __launch_bounds__ (TJ_BLOCK)
void scan_kernel (const uint8_t *__restrict__ seq, long n_bytes, long n_tiles, int k, int mprime,
                  u64 *__restrict__ out, u64 cap, DevCounters *ctr, FixEntry *fix, u32 fix_cap)
{
  __shared__ u32 s_code[TJ_CODEW];
  __shared__ u32 s_start[TJ_MASKW];
  __shared__ u32 s_sent[TJ_MASKW];
  __shared__ u32 s_inval[TJ_MASKW];
  __shared__ unsigned short s_cand[TJ_MAXCAND];
  __shared__ u32 s_ncand;

  const int tid = threadIdx.x;
  const u64 km = kmask (k);
  const u64 kbits = (k >= 64) ? ~0ull : ((1ull << k) - 1ull);   // k <= 32

  for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const long g0 = tile * (long) TJ_TILE - TJ_HL;      // stream position of window byte 0 (may be negative)

    // ---- phase 1: load + classify -------------------------------------------------------------------------
    if (tid == 0) s_ncand = 0;
    if (tid < 4) { s_code[TJ_CODEW - 4 + tid] = 0; s_start[TJ_MASKW - 4 + tid] = 0; s_sent[TJ_MASKW - 4 + tid] = 0xFFFFFFFFu; s_inval[TJ_MASKW - 4 + tid] = 0; }
    for (int c = tid; c < TJ_NCHUNK; c += TJ_BLOCK) {
      const long g = g0 + 16l * c;
      u32 w[4];
      if (g >= 0 && g + 16 <= n_bytes) {
        uint4 v = *reinterpret_cast<const uint4 *> (seq + g);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
      }
      else {
        for (int j = 0; j < 4; j++) {
          u32 x = 0;
          for (int b = 0; b < 4; b++) x |= stream_byte (seq, n_bytes, g + 4 * j + b) << (8 * b);
          w[j] = x;
        }
      }
      u32 prev = stream_byte (seq, n_bytes, g - 1);
      u32 code32 = 0, st16 = 0, se16 = 0, iv16 = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        u32 c8, s4, e4, i4;
        classify_word (w[j], prev, c8, s4, e4, i4);
        code32 |= c8 << (8 * j); st16 |= s4 << (4 * j); se16 |= e4 << (4 * j); iv16 |= i4 << (4 * j);
        prev = w[j] >> 24;
      }
      s_code[c] = code32;
      reinterpret_cast<unsigned short *> (s_start)[c] = (unsigned short) st16;
      reinterpret_cast<unsigned short *> (s_sent)[c] = (unsigned short) se16;
      reinterpret_cast<unsigned short *> (s_inval)[c] = (unsigned short) iv16;
    }
    __syncthreads ();

    // ---- phase 2: candidate tract starts among this lane's 16 positions ------------------------------------
    {
      const int p0 = TJ_HL + 16 * tid;
      const u64 S = bits64 (s_start, p0);
      u32 cand = (u32) S & 0xFFFFu;
      for (int j = 1; j < mprime; j++) cand &= ~(u32) (S >> j);     // next m'-1 positions continue the run
      cand &= ~(u32) reinterpret_cast<unsigned short *> (s_sent)[p0 >> 4];  // a run of delimiters is not a tract
      if (cand) {
        u32 at = atomicAdd (&s_ncand, (u32) __popc (cand));
        while (cand) { int b = __ffs ((int) cand) - 1; cand &= cand - 1u; if (at < TJ_MAXCAND) s_cand[at] = (unsigned short) (p0 + b); at++; }
      }
    }
    __syncthreads ();

    // ---- phase 3: one lane per candidate --------------------------------------------------------------------
    const int ncand = min ((int) s_ncand, TJ_MAXCAND);
    for (int cb0 = 0; cb0 < ncand; cb0 += TJ_BLOCK) {
      const int ci = cb0 + tid;
      bool have = false;
      u64 c0 = 0, c1 = 0, meta = 0, pos = 0;
      if (ci < ncand) {
        const int s = s_cand[ci];
        const long gs = g0 + s;
        // run end: first run start after s
        int e = -1;
        for (int p = s + 1; p < TJ_WIN; p += 64) {
          u64 ns = bits64 (s_start, p);
          if (ns) { e = p + __ffsll ((long long) ns) - 2; break; }
        }
        const bool inval = (s_inval[s >> 5] >> (s & 31)) & 1u;
        bool ok;
        long len;
        u64 left = 0, right = 0;
        u32 cb = 0, linv = 0, rinv = 0;
        if (e >= 0 && e + k < TJ_WIN) {             // everything needed is in LDS
          len = e - s + 1;
          ok = ((bits64 (s_sent, s - k) & kbits) == 0ull) && ((bits64 (s_sent, e + 1) & kbits) == 0ull);
          if (ok && !inval) {
            left = bits64 (s_code, 2 * (s - k)) & km;
            right = bits64 (s_code, 2 * (e + 1)) & km;
            cb = (s_code[s >> 4] >> (2 * (s & 15))) & 3u;
            if (cb >= 2u) { linv = (u32) (bits64 (s_inval, s - k) & kbits); rinv = (u32) (bits64 (s_inval, e + 1) & kbits); }
          }
        }
        else {                                      // tract runs past the window: walk the stream (rare)
          const u32 b = stream_byte (seq, n_bytes, gs);
          long ge = gs;
          while (stream_byte (seq, n_bytes, ge + 1) == b) ge++;
          len = ge - gs + 1;
          ok = flanks_from_stream (seq, n_bytes, gs, ge, k, left, right, linv, rinv);
          cb = byte_code (b);
        }
        if (ok) {
          if (!inval) {
            u32 base, flag;
            canonicalise (left, right, linv, rinv, cb, k, c0, c1, base, flag);
            meta = make_meta (base, len, flag);
            pos = (u64) gs;
            have = true;
          }
          else {                                    // non-ACGTU run: context comes from the previous tract of the read
            u64 at = atomicAdd (&ctr->n_fix, 1ull);
            if (at < fix_cap) { fix[at].pos = gs; fix[at].len = len; }
            else ctr->fix_overflow = 1u;
          }
        }
      }
      emit_record<W> (have, c0, c1, meta, pos, out, cap, ctr);
    }
    __syncthreads ();
  }
}

// Non-ACGTU runs that qualify as tracts (reference: src/hopo_counter.c:246-248: add_kmer is called with whatever the
// previous tract of the same read left in context[], hopo_base_int and reverse_forward_flag).  One thread per entry
// walks back through its read to the nearest earlier recorded tract of a valid base and re-uses its context.
template <int W>
__global__ void nrun_fixup_kernel (const uint8_t *__restrict__ seq, long n_bytes, int k, int mprime,
                                   u64 *__restrict__ out, u64 cap, DevCounters *ctr, const FixEntry *fix, u32 fix_cap)
{
  u64 n_fix = ctr->n_fix;
  if (n_fix > fix_cap) n_fix = fix_cap;
  for (u64 i = blockIdx.x * (u64) blockDim.x + threadIdx.x; i < n_fix; i += (u64) gridDim.x * blockDim.x) {
    const long gs = fix[i].pos;
    bool found = false;
    u64 left = 0, right = 0;
    u32 cb = 0, linv = 0, rinv = 0;
    long p = gs - 1;
    while (p >= 0 && seq[p] != '\n') {
      const u32 b = seq[p];
      long q = p;
      while (q - 1 >= 0 && seq[q - 1] == b) q--;       // run [q, p]
      if (byte_is_acgtu (b) && (p - q + 1) >= mprime) {
        // recorded iff k bases of the same read precede it (its right side is fine: it ends before our run does)
        if (flanks_from_stream (seq, n_bytes, q, p, k, left, right, linv, rinv)) { found = true; cb = byte_code (b); }
        break;                                          // an earlier run would start even closer to the read start
      }
      p = q - 1;
    }
    if (found) {
      u64 c0, c1; u32 base, flag;
      canonicalise (left, right, linv, rinv, cb, k, c0, c1, base, flag);
      u64 idx = atomicAdd (&ctr->n_rec, 1ull);
      if (idx < cap) store_record<W> (out, idx, c0, c1, make_meta (base, fix[i].len, flag), (u64) gs);
      else ctr->overflow = 1u;
    }
    else atomicAdd (&ctr->n_undefined, 1ull);
  }
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

#define RS_ITEMS      4096
#define RS_WAVE_ITEMS 1024

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

struct FinCounts { u32 n_seg, n_kept, n_ctx, n_idx; int coverage; u32 pad; };

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

// one thread per run of equal keys: multiplicity, OR of strand flags, filter decision
__global__ void seg_decide_kernel (const u64 *__restrict__ rec, long n, const u32 *__restrict__ headpos, const u32 *n_seg_p,
                                   int remove_biased, u32 *__restrict__ keep, u64 *__restrict__ seg_meta)
{
  const u32 n_seg = *n_seg_p;
  for (u32 j = blockIdx.x * blockDim.x + threadIdx.x; j < n_seg; j += gridDim.x * blockDim.x) {
    const long s = headpos[j], e = (j + 1 < n_seg) ? (long) headpos[j + 1] : n;
    const u64 mf = rec[3 * s + 2], ml = rec[3 * (e - 1) + 2];
    const u64 flag = ((mf | ml) >> TJ_META_FLAG_SHIFT) & 3ull;          // sorted by flag inside the run
    const u64 cnt = (u64) (e - s) & 0xFFFFFull;                          // 20-bit store wraps (reference :361)
    u64 meta = mf & ~((0xFFFFFull << TJ_META_COUNT_SHIFT) | (7ull << TJ_META_FLAG_SHIFT));
    meta |= (cnt << TJ_META_COUNT_SHIFT) | (flag << TJ_META_FLAG_SHIFT);
    seg_meta[j] = meta;
    keep[j] = remove_biased ? (flag == 3ull) : (meta_count (meta) > 1);
  }
}

__global__ void seg_write_kernel (const u64 *__restrict__ rec, const u32 *__restrict__ headpos, const u32 *n_seg_p,
                                  const u32 *__restrict__ keep, const u32 *__restrict__ outpos, const u64 *__restrict__ seg_meta,
                                  u64 *__restrict__ kept)
{
  const u32 n_seg = *n_seg_p;
  for (u32 j = blockIdx.x * blockDim.x + threadIdx.x; j < n_seg; j += gridDim.x * blockDim.x)
    if (keep[j]) {
      const long s = headpos[j];
      u64 *q = kept + 3 * (u64) outpos[j];
      q[0] = rec[3 * s]; q[1] = rec[3 * s + 1]; q[2] = seg_meta[j];
    }
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

// coverage: pooled 31-bit-truncated flanks weighted by count, largest pooled weight wins
__global__ void cov_insert_kernel (const u64 *__restrict__ kept, long n1, u32 *__restrict__ keys, int *__restrict__ sums, int log2t)
{
  const u32 tmask = (1u << log2t) - 1u;
  for (long i = blockIdx.x * (long) blockDim.x + threadIdx.x; i < 2 * n1; i += (long) gridDim.x * blockDim.x) {
    const long r = (i < n1) ? i : i - n1;
    const u32 key = (u32) (kept[3 * r + (i < n1 ? 0 : 1)] & 0x7FFFFFFFull);
    const int w = meta_count (kept[3 * r + 2]);
    u32 slot = (key * 2654435761u) >> (32 - log2t);
    for (u32 probe = 0; probe <= tmask; probe++) {
      u32 old = atomicCAS (&keys[slot], 0xFFFFFFFFu, key);
      if (old == 0xFFFFFFFFu || old == key) { atomicAdd (&sums[slot], w); break; }
      slot = (slot + 1u) & tmask;
    }
  }
}

__global__ void cov_max_kernel (const u32 *__restrict__ keys, const int *__restrict__ sums, long t, int *result)
{
  int best = INT_MIN;
  for (long i = blockIdx.x * (long) blockDim.x + threadIdx.x; i < t; i += (long) gridDim.x * blockDim.x)
    if (keys[i] != 0xFFFFFFFFu) best = max (best, sums[i]);
  for (int o = 32; o > 0; o >>= 1) best = max (best, __shfl_down (best, o));
  if ((threadIdx.x & 63) == 0 && best != INT_MIN) atomicMax (result, best);
}

__global__ void set_int_kernel (int *p, int v) { *p = v; }

// hopo_element (40 B, host layout) -> 24 B device record
__global__ void elem_to_record_kernel (const u64 *__restrict__ elems5, long n, u64 *__restrict__ rec)
{
  for (long i = blockIdx.x * (long) blockDim.x + threadIdx.x; i < n; i += (long) gridDim.x * blockDim.x) {
    rec[3 * i] = elems5[5 * i]; rec[3 * i + 1] = elems5[5 * i + 1]; rec[3 * i + 2] = elems5[5 * i + 2];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// host side of the thin layer

struct DevBuf
{
  void *p = nullptr;
  size_t cap = 0;
};

struct tjamd_counter
{
  int device = 0, k = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  DevCounters *d_ctr = nullptr, *h_ctr = nullptr;     // raw list counters (device / pinned host mirror)
  DevCounters *d_lctr = nullptr;                      // located list counters
  DevBuf raw, alt, stage, fix, loc;
  long n_raw_known = 0;       // exact after the last synchronisation
  long n_raw_bound = 0;       // known + worst case of the scans launched since
  long n_undefined = 0;
  DevBuf hist, flags, segid, headpos, keep, outpos, segmeta, scan_tmp, kept, idx_i, idx_f, cov_keys, cov_sums;
  FinCounts *d_fin = nullptr, *h_fin = nullptr;
  long n_kept = 0; int n_idx = 0, coverage = 0, status = -1;
  hipEvent_t ev_s0 = nullptr, ev_s1 = nullptr, ev_f0 = nullptr, ev_f1 = nullptr;
  bool scan_timed = false, fin_timed = false;
  long last_scan_launches = 0;
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
  HIPCHK_NULL (hipStreamCreateWithFlags (&c->own_stream, hipStreamNonBlocking));
  c->stream = c->own_stream;
  HIPCHK_NULL (hipMalloc ((void **) &c->d_ctr, sizeof (DevCounters)));
  HIPCHK_NULL (hipMalloc ((void **) &c->d_lctr, sizeof (DevCounters)));
  HIPCHK_NULL (hipMalloc ((void **) &c->d_fin, sizeof (FinCounts)));
  HIPCHK_NULL (hipHostMalloc ((void **) &c->h_ctr, sizeof (DevCounters), hipHostMallocDefault));
  HIPCHK_NULL (hipHostMalloc ((void **) &c->h_fin, sizeof (FinCounts), hipHostMallocDefault));
  HIPCHK_NULL (hipMemsetAsync (c->d_ctr, 0, sizeof (DevCounters), c->stream));
  HIPCHK_NULL (hipMemsetAsync (c->d_lctr, 0, sizeof (DevCounters), c->stream));
  HIPCHK_NULL (hipEventCreate (&c->ev_s0)); HIPCHK_NULL (hipEventCreate (&c->ev_s1));
  HIPCHK_NULL (hipEventCreate (&c->ev_f0)); HIPCHK_NULL (hipEventCreate (&c->ev_f1));
  HIPCHK_NULL (hipStreamSynchronize (c->stream));
  return c;
}

extern "C" void tjamd_counter_destroy (tjamd_counter *c)
{
  if (!c) return;
  (void) hipSetDevice (c->device);
  (void) hipStreamSynchronize (c->stream);
  DevBuf *all[] = {&c->raw, &c->alt, &c->stage, &c->fix, &c->loc, &c->hist, &c->flags, &c->segid, &c->headpos, &c->keep, &c->outpos,
                   &c->segmeta, &c->scan_tmp, &c->kept, &c->idx_i, &c->idx_f, &c->cov_keys, &c->cov_sums};
  for (DevBuf *b : all) release (*b);
  if (c->d_ctr) (void) hipFree (c->d_ctr);
  if (c->d_lctr) (void) hipFree (c->d_lctr);
  if (c->d_fin) (void) hipFree (c->d_fin);
  if (c->h_ctr) (void) hipHostFree (c->h_ctr);
  if (c->h_fin) (void) hipHostFree (c->h_fin);
  if (c->ev_s0) (void) hipEventDestroy (c->ev_s0);
  if (c->ev_s1) (void) hipEventDestroy (c->ev_s1);
  if (c->ev_f0) (void) hipEventDestroy (c->ev_f0);
  if (c->ev_f1) (void) hipEventDestroy (c->ev_f1);
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

extern "C" int tjamd_counter_reset (tjamd_counter *c)
{
  if (!c) return set_err (TJAMD_ERR_ARG, "null counter");
  HIPCHK (hipSetDevice (c->device));
  HIPCHK (hipMemsetAsync (c->d_ctr, 0, sizeof (DevCounters), c->stream));
  c->n_raw_known = c->n_raw_bound = 0; c->n_undefined = 0;
  c->n_kept = 0; c->n_idx = 0; c->coverage = 0; c->status = -1;
  return TJAMD_OK;
}

static int sync_counters (tjamd_counter *c)
{
  HIPCHK (hipMemcpyAsync (c->h_ctr, c->d_ctr, sizeof (DevCounters), hipMemcpyDeviceToHost, c->stream));
  HIPCHK (hipStreamSynchronize (c->stream));
  if (c->h_ctr->overflow) return set_err (TJAMD_ERR_CAPACITY, "raw record list overflowed (%llu records, capacity %zu)",
                                         (unsigned long long) c->h_ctr->n_rec, c->raw.cap / 24);
  if (c->h_ctr->fix_overflow) return set_err (TJAMD_ERR_CAPACITY, "too many non-ACGTU tract candidates in one batch (%llu)",
                                             (unsigned long long) c->h_ctr->n_fix);
  c->n_raw_known = c->n_raw_bound = (long) c->h_ctr->n_rec;
  c->n_undefined = (long) c->h_ctr->n_undefined;
  return TJAMD_OK;
}

#define TJ_FIX_CAP (1u << 20)

template <int W>
static int launch_scan (tjamd_counter *c, const uint8_t *d_seq, size_t n_bytes, int mprime, u64 *out, u64 cap, DevCounters *ctr)
{
  int rc = ensure (c->fix, (size_t) TJ_FIX_CAP * sizeof (FixEntry), c->stream);
  if (rc) return rc;
  long n_tiles = (long) ((n_bytes + TJ_TILE - 1) / TJ_TILE);
  if (n_tiles == 0) return TJAMD_OK;
  int grid = (int) std::min<long> (n_tiles, 256l * 8);
  hipLaunchKernelGGL (scan_kernel<W>, dim3 (grid), dim3 (TJ_BLOCK), 0, c->stream, d_seq, (long) n_bytes, n_tiles, c->k, mprime,
                      out, cap, ctr, (FixEntry *) c->fix.p, (u32) TJ_FIX_CAP);
  HIPCHK (hipGetLastError ());
  // non-ACGTU tracts (usually none): resolved against the stream, then the pending list is cleared
  hipLaunchKernelGGL (nrun_fixup_kernel<W>, dim3 (64), dim3 (256), 0, c->stream, d_seq, (long) n_bytes, c->k, mprime,
                      out, cap, ctr, (const FixEntry *) c->fix.p, (u32) TJ_FIX_CAP);
  HIPCHK (hipGetLastError ());
  HIPCHK (hipMemsetAsync (&ctr->n_fix, 0, sizeof (u64), c->stream));
  c->last_scan_launches++;
  return TJAMD_OK;
}

extern "C" int tjamd_scan_device (tjamd_counter *c, const void *d_stream, size_t n_bytes, int min_tract_size)
{
  if (!c) return set_err (TJAMD_ERR_ARG, "null counter");
  if (n_bytes == 0) return TJAMD_OK;
  if (!d_stream || ((uintptr_t) d_stream & 15u)) return set_err (TJAMD_ERR_ARG, "device stream pointer must be non-null and 16-byte aligned");
  if (min_tract_size < 1 || min_tract_size > 32) return set_err (TJAMD_ERR_ARG, "min_tract_size %d outside [1,32] (reference clamp: src/main.c:186-187)", min_tract_size);
  HIPCHK (hipSetDevice (c->device));
  const int mprime = std::max (min_tract_size, 2);          // a tract needs two equal bytes: m = 1 behaves as m = 2
  const long bound = (long) (n_bytes / (size_t) mprime) + 1; // tracts are disjoint runs of >= m' bytes
  int rc = ensure (c->raw, (size_t) (c->n_raw_bound + bound) * 24, c->stream, (size_t) c->n_raw_bound * 24);
  if (rc) return rc;
  c->n_raw_bound += bound;
  HIPCHK (hipEventRecord (c->ev_s0, c->stream));
  c->last_scan_launches = 0;
  rc = launch_scan<3> (c, (const uint8_t *) d_stream, n_bytes, mprime, (u64 *) c->raw.p, (u64) (c->raw.cap / 24), c->d_ctr);
  if (rc) return rc;
  HIPCHK (hipEventRecord (c->ev_s1, c->stream));
  c->scan_timed = true;
  c->status = -1;
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
{
  if (!c) return -set_err (TJAMD_ERR_ARG, "null counter");
  if (min_tract_size < 1) return -set_err (TJAMD_ERR_ARG, "min_tract_size %d < 1", min_tract_size);
  if (n_bytes == 0) return 0;
  if (hipSetDevice (c->device) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "hipSetDevice failed");
  const int mprime = std::max (min_tract_size, 2);
  const long bound = (long) (n_bytes / (size_t) mprime) + 1;
  int rc = ensure (c->stage, (n_bytes + 255) & ~(size_t) 255, c->stream);
  if (!rc) rc = ensure (c->loc, (size_t) bound * 32, c->stream);
  if (rc) return -rc;
  if (hipMemcpyAsync (c->stage.p, h_stream, n_bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
      hipMemsetAsync (c->d_lctr, 0, sizeof (DevCounters), c->stream) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "copy to device failed");
  rc = launch_scan<4> (c, (const uint8_t *) c->stage.p, n_bytes, mprime, (u64 *) c->loc.p, (u64) bound, c->d_lctr);
  if (rc) return -rc;
  DevCounters h;
  if (hipMemcpyAsync (c->h_ctr, c->d_lctr, sizeof (DevCounters), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
      hipStreamSynchronize (c->stream) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "located scan failed: %s", hipGetErrorString (hipGetLastError ()));
  h = *c->h_ctr;
  if (h.overflow || h.fix_overflow) return -set_err (TJAMD_ERR_CAPACITY, "located scan overflow");
  c->n_undefined += (long) h.n_undefined;
  long n = (long) h.n_rec;
  if (n > capacity) return -set_err (TJAMD_ERR_CAPACITY, "located scan produced %ld records, caller capacity %ld", n, capacity);
  if (n) {
    if (hipMemcpy (out, c->loc.p, (size_t) n * 32, hipMemcpyDeviceToHost) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "download failed");
    std::sort (out, out + n, [] (const tjamd_located_record &a, const tjamd_located_record &b) { return a.pos < b.pos; });
  }
  return n;
}

extern "C" void *tjamd_host_alloc (size_t bytes)
{
  void *p = nullptr;
  if (hipHostMalloc (&p, bytes, hipHostMallocDefault) != hipSuccess) { set_err (TJAMD_ERR_HIP, "hipHostMalloc of %zu bytes failed", bytes); return NULL; }
  return p;
}

extern "C" void tjamd_host_free (void *p) { if (p) (void) hipHostFree (p); }

extern "C" int tjamd_sync (tjamd_counter *c)
{
  if (!c) return set_err (TJAMD_ERR_ARG, "null counter");
  HIPCHK (hipSetDevice (c->device));
  HIPCHK (hipStreamSynchronize (c->stream));
  return TJAMD_OK;
}

extern "C" long tjamd_raw_count (tjamd_counter *c)
{
  if (!c) return -set_err (TJAMD_ERR_ARG, "null counter");
  if (hipSetDevice (c->device) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "hipSetDevice failed");
  int rc = sync_counters (c);
  return rc ? -rc : c->n_raw_known;
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
  if (n && hipMemcpy (out, c->raw.p, (size_t) n * 24, hipMemcpyDeviceToHost) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "download failed");
  return n;
}

extern "C" int tjamd_upload_raw (tjamd_counter *c, const hopo_element *elems, long n)
{
  if (!c || (n && !elems) || n < 0) return set_err (TJAMD_ERR_ARG, "bad arguments");
  if (n == 0) return TJAMD_OK;
  HIPCHK (hipSetDevice (c->device));
  int rc = sync_counters (c);
  if (rc) return rc;
  rc = ensure (c->raw, (size_t) (c->n_raw_known + n) * 24, c->stream, (size_t) c->n_raw_known * 24);
  if (!rc) rc = ensure (c->stage, (size_t) n * 40, c->stream);
  if (rc) return rc;
  HIPCHK (hipMemcpyAsync (c->stage.p, elems, (size_t) n * 40, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL (elem_to_record_kernel, dim3 ((unsigned) std::min<long> ((n + 255) / 256, 2048)), dim3 (256), 0, c->stream,
                      (const u64 *) c->stage.p, n, (u64 *) c->raw.p + 3 * c->n_raw_known);
  HIPCHK (hipGetLastError ());
  c->h_ctr->n_rec = (u64) (c->n_raw_known + n);
  HIPCHK (hipMemcpyAsync (&c->d_ctr->n_rec, &c->h_ctr->n_rec, sizeof (u64), hipMemcpyHostToDevice, c->stream));
  HIPCHK (hipStreamSynchronize (c->stream));
  c->n_raw_known += n; c->n_raw_bound = c->n_raw_known;
  c->status = -1;
  return TJAMD_OK;
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

extern "C" int tjamd_finalise (tjamd_counter *c, int remove_biased, int min_coverage, int *status)
{
  if (!c) return set_err (TJAMD_ERR_ARG, "null counter");
  HIPCHK (hipSetDevice (c->device));
  int rc = sync_counters (c);
  if (rc) return rc;
  const long n = c->n_raw_known;
  c->n_kept = 0; c->n_idx = 0; c->coverage = 0;
  if (n == 0) { c->status = 1; if (status) *status = 1; return TJAMD_OK; }     // reference: src/hopo_counter.c:345-349
  if (n >= (1l << 31)) return set_err (TJAMD_ERR_CAPACITY, "%ld raw records exceed the reference's int n_elem", n);

  HIPCHK (hipEventRecord (c->ev_f0, c->stream));
  rc = ensure (c->alt, (size_t) n * 24, c->stream);
  if (!rc) rc = ensure (c->flags, (size_t) n * 4, c->stream);
  if (!rc) rc = ensure (c->segid, (size_t) n * 4, c->stream);
  if (!rc) rc = ensure (c->headpos, (size_t) n * 4, c->stream);
  if (!rc) rc = ensure (c->scan_tmp, scan_tmp_words (n) * 4, c->stream);
  if (rc) return rc;

  // step 1: sort (reference :351)
  u64 *a = (u64 *) c->raw.p, *b = (u64 *) c->alt.p;
  rc = radix_sort_records (c, a, b, n);
  if (rc) return rc;
  if (a != (u64 *) c->raw.p) std::swap (c->raw, c->alt);      // sorted records are the raw list again

  // step 2: collapse equal keys, filter (reference :356-374)
  u32 *flags = (u32 *) c->flags.p, *segid = (u32 *) c->segid.p, *headpos = (u32 *) c->headpos.p;
  hipLaunchKernelGGL (seg_heads_kernel, dim3 (grid_for (n)), dim3 (256), 0, c->stream, (const u64 *) a, n, flags, 0);
  HIPCHK (hipGetLastError ());
  rc = exclusive_scan (c, flags, segid, n, &c->d_fin->n_seg, (u32 *) c->scan_tmp.p, c->scan_tmp.cap / 4);
  if (rc) return rc;
  hipLaunchKernelGGL (seg_headpos_kernel, dim3 (grid_for (n)), dim3 (256), 0, c->stream, (const u32 *) flags, (const u32 *) segid, n, headpos);
  HIPCHK (hipGetLastError ());
  // the number of runs is only known on the device: size by the bound n, kernels read the count
  rc = ensure (c->keep, (size_t) n * 4, c->stream);
  if (!rc) rc = ensure (c->outpos, (size_t) n * 4, c->stream);
  if (!rc) rc = ensure (c->segmeta, (size_t) n * 8, c->stream);
  if (rc) return rc;
  HIPCHK (hipMemcpyAsync (c->h_fin, c->d_fin, sizeof (FinCounts), hipMemcpyDeviceToHost, c->stream));
  HIPCHK (hipStreamSynchronize (c->stream));
  const long n_seg = c->h_fin->n_seg;
  u32 *keep = (u32 *) c->keep.p, *outpos = (u32 *) c->outpos.p;
  hipLaunchKernelGGL (seg_decide_kernel, dim3 (grid_for (n_seg)), dim3 (256), 0, c->stream, (const u64 *) a, n, (const u32 *) headpos,
                      (const u32 *) &c->d_fin->n_seg, remove_biased, keep, (u64 *) c->segmeta.p);
  HIPCHK (hipGetLastError ());
  rc = exclusive_scan (c, keep, outpos, n_seg, &c->d_fin->n_kept, (u32 *) c->scan_tmp.p, c->scan_tmp.cap / 4);
  if (rc) return rc;
  HIPCHK (hipMemcpyAsync (c->h_fin, c->d_fin, sizeof (FinCounts), hipMemcpyDeviceToHost, c->stream));
  HIPCHK (hipStreamSynchronize (c->stream));
  const long n1 = c->h_fin->n_kept;
  if (n1 == 0) {                                                               // reference :376-381
    HIPCHK (hipEventRecord (c->ev_f1, c->stream)); c->fin_timed = true;
    c->status = 2; if (status) *status = 2; return TJAMD_OK;
  }
  rc = ensure (c->kept, (size_t) n1 * 24, c->stream);
  if (rc) return rc;
  hipLaunchKernelGGL (seg_write_kernel, dim3 (grid_for (n_seg)), dim3 (256), 0, c->stream, (const u64 *) a, (const u32 *) headpos,
                      (const u32 *) &c->d_fin->n_seg, (const u32 *) keep, (const u32 *) outpos, (const u64 *) c->segmeta.p, (u64 *) c->kept.p);
  HIPCHK (hipGetLastError ());
  c->n_kept = n1;

  // step 4: contexts deep enough get an index range (reference :388-404)
  const u64 *kept = (const u64 *) c->kept.p;
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
  int log2t = 10;
  while ((1l << log2t) < 8 * n1 && log2t < 31) log2t++;
  const long t = 1l << log2t;
  rc = ensure (c->cov_keys, (size_t) t * 4, c->stream);
  if (!rc) rc = ensure (c->cov_sums, (size_t) t * 4, c->stream);
  if (rc) return rc;
  HIPCHK (hipMemsetAsync (c->cov_keys.p, 0xFF, (size_t) t * 4, c->stream));
  HIPCHK (hipMemsetAsync (c->cov_sums.p, 0, (size_t) t * 4, c->stream));
  hipLaunchKernelGGL (set_int_kernel, dim3 (1), dim3 (1), 0, c->stream, &c->d_fin->coverage, INT_MIN);
  hipLaunchKernelGGL (cov_insert_kernel, dim3 (grid_for (2 * n1)), dim3 (256), 0, c->stream, kept, n1, (u32 *) c->cov_keys.p, (int *) c->cov_sums.p, log2t);
  HIPCHK (hipGetLastError ());
  hipLaunchKernelGGL (cov_max_kernel, dim3 (grid_for (t)), dim3 (256), 0, c->stream, (const u32 *) c->cov_keys.p, (const int *) c->cov_sums.p, t, &c->d_fin->coverage);
  HIPCHK (hipGetLastError ());
  HIPCHK (hipEventRecord (c->ev_f1, c->stream));
  c->fin_timed = true;
  HIPCHK (hipMemcpyAsync (c->h_fin, c->d_fin, sizeof (FinCounts), hipMemcpyDeviceToHost, c->stream));
  HIPCHK (hipStreamSynchronize (c->stream));
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
  std::vector<tjamd_record> tmp ((size_t) n1);
  if (hipMemcpy (tmp.data (), c->kept.p, (size_t) n1 * 24, hipMemcpyDeviceToHost) != hipSuccess) return -set_err (TJAMD_ERR_HIP, "download failed");
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

extern "C" long tjamd_merge_samples (tjamd_counter *c, const void *d_records, const long *counts, int n_samples,
                                      void *d_out_keys, void *d_out_counts, long capacity)
{
  (void) c; (void) d_records; (void) counts; (void) n_samples; (void) d_out_keys; (void) d_out_counts; (void) capacity;
  return -set_err (TJAMD_ERR_STATE, "tjamd_merge_samples: not built yet");
}
