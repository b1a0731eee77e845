/* tj_inflate.c -- a raw DEFLATE (RFC 1951) decoder for the feeder, see tj_inflate.h.
 *
 * Why not zlib's inflate(): the reference reads its input through zlib (src/hopo_counter.c:142, kseq over gzread) and
 * on a FASTQ file zlib inflates about 320 MB/s per thread -- 1 M reads/s, four orders of magnitude below the scan
 * kernel.  This decoder does what the fast ones do (64-bit bit buffer refilled without branches, one table look-up per
 * literal / length symbol with sub-tables for the long codes, word-wise match copies) and is written to be resumable:
 * it stops wherever the output buffer ends and carries on in another one, given the last 32 KiB in front of it.
 * Every gzip member it inflates is checked against the member's CRC-32 and size by the caller (feeder.c); nothing
 * it produces reaches the scan unchecked in the BGZF path, and a mismatch at the end of a streamed member is fatal.
 */
#include "tj_inflate.h"
#include <string.h>
#include <stdlib.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <pthread.h>

#define LIT_TABLE_BITS   11
#define DIST_TABLE_BITS  8
#define MAX_CODE_LEN     15
#define N_LITLEN         288
#define N_DIST           32
#define N_PRECODE        19

/* table entry: bits 0-7 = bits to drop for this step (code length, or the primary index width for a sub-table link),
 * bits 8-15 = flags / extra-bit count, bits 16-31 = value (literal, base length / distance, or sub-table start) */
#define E_LITERAL   0x8000u
#define E_EOB       0x4000u
#define E_SUBTABLE  0x2000u
#define E_INVALID   0x1000u
#define E_EXTRA(e)  (((e) >> 8) & 0x0fu)
/* Primary literal entries of the fast loop's table (pair_literals) pack up to three literals:
 * bits 0-3 = code bits of all of them, bits 4-5 = how many - 1, bits 8-14 = the third (ASCII only), bits 16-31 = first | second << 8. */
#define E_LITN(e)   ((((e) >> 4) & 3u) + 1u)

static const unsigned short len_base[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
static const unsigned char  len_extra[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
static const unsigned short dist_base[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
static const unsigned char  dist_extra[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
static const unsigned char  precode_order[N_PRECODE] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};

static unsigned
bit_reverse (unsigned code, int len)
{
  unsigned r = 0;
  int i;
  for (i = 0; i < len; i++) { r = (r << 1) | (code & 1u); code >>= 1; }
  return r;
}

/* Canonical Huffman decode table from code lengths.  kind: 0 = precode (value = symbol), 1 = literal/length, 2 = distance.
 * Returns 0, or -1 if the lengths do not describe a usable code (over-subscribed; incomplete codes are accepted only in
 * the forms zlib accepts: a single code of length 1, or no code at all for distances). */
static int
build_table (unsigned *table, int table_bits, int table_cap, const unsigned char *lens, int n_sym, int kind)
{
  unsigned count[MAX_CODE_LEN + 1], next_code[MAX_CODE_LEN + 2];
  unsigned sub_start[1u << LIT_TABLE_BITS];            /* per primary index: start of its sub-table (0 = none yet) */
  unsigned char sub_bits_of[1u << LIT_TABLE_BITS];
  int len, s, max_len = 0, used = 1 << table_bits;
  unsigned code, left;
  for (len = 0; len <= MAX_CODE_LEN; len++) count[len] = 0;
  for (s = 0; s < n_sym; s++) count[lens[s]]++;
  count[0] = 0;
  for (len = 1; len <= MAX_CODE_LEN; len++) if (count[len]) max_len = len;
  for (s = 0; s < (1 << table_bits); s++) table[s] = E_INVALID | 1u;
  if (max_len == 0) return 0;                           /* no codes at all: every look-up is an error (as in zlib's inftrees.c) */
  left = 1;
  for (len = 1; len <= MAX_CODE_LEN; len++) {
    left <<= 1;
    if (count[len] > left) return -1;                   /* over-subscribed */
    left -= count[len];
  }
  if (left > 0 && (kind == 0 || max_len != 1)) return -1;   /* incomplete: only a lone 1-bit code is let through (zlib's rule) */
  code = 0; next_code[0] = 0;
  for (len = 1; len <= MAX_CODE_LEN; len++) { code = (code + count[len - 1]) << 1; next_code[len] = code; }
  /* widths of the sub-tables: for each primary prefix, the longest code under it */
  if (max_len > table_bits) {
    memset (sub_start, 0, sizeof (unsigned) << table_bits);
    memset (sub_bits_of, 0, (size_t) 1 << table_bits);
    {
      unsigned nc[MAX_CODE_LEN + 2];
      memcpy (nc, next_code, sizeof nc);
      for (s = 0; s < n_sym; s++) {
        const int l = lens[s];
        if (l > table_bits) {
          const unsigned rev = bit_reverse (nc[l], l);
          const unsigned prim = rev & ((1u << table_bits) - 1u);
          if ((unsigned) (l - table_bits) > sub_bits_of[prim]) sub_bits_of[prim] = (unsigned char) (l - table_bits);
        }
        if (l) nc[l]++;
      }
    }
  }
  for (s = 0; s < n_sym; s++) {
    const int l = lens[s];
    unsigned e, rev;
    if (!l) continue;
    rev = bit_reverse (next_code[l]++, l);
    if (kind == 0) e = ((unsigned) s << 16);
    else if (kind == 1) {
      if (s < 256) e = ((unsigned) s << 16) | E_LITERAL;
      else if (s == 256) e = E_EOB;
      else if (s < 286) e = ((unsigned) len_base[s - 257] << 16) | ((unsigned) len_extra[s - 257] << 8);
      else e = E_INVALID;
    }
    else {
      if (s < 30) e = ((unsigned) dist_base[s] << 16) | ((unsigned) dist_extra[s] << 8);
      else e = E_INVALID;
    }
    if (l <= table_bits) {
      unsigned i;
      e |= (unsigned) l;
      for (i = rev; i < (1u << table_bits); i += 1u << l) table[i] = e;
    }
    else {
      const unsigned prim = rev & ((1u << table_bits) - 1u);
      const int sb = sub_bits_of[prim];
      unsigned i;
      if (!sub_start[prim]) {
        if (used + (1 << sb) > table_cap) return -1;
        sub_start[prim] = (unsigned) used;
        for (i = 0; i < (1u << sb); i++) table[used + i] = E_INVALID | 1u;
        table[prim] = ((unsigned) used << 16) | E_SUBTABLE | ((unsigned) sb << 8) | (unsigned) table_bits;
        used += 1 << sb;
      }
      e |= (unsigned) (l - table_bits);
      for (i = rev >> table_bits; i < (1u << sb); i += 1u << (l - table_bits)) table[sub_start[prim] + i] = e;
    }
  }
  return 0;
}

/* Literal-heavy data (quality strings, random DNA) spends its time in the chain look-up -> shift -> look-up; primary entries
 * whose index also holds a whole second literal code are turned into two-literal entries.  `single` keeps the plain table. */
static void
pair_literals (unsigned *table, unsigned *single)
{
  unsigned i;
  memcpy (single, table, sizeof (unsigned) << LIT_TABLE_BITS);
  for (i = 0; i < (1u << LIT_TABLE_BITS); i++) {
    const unsigned e1 = single[i];
    if (e1 & E_LITERAL) {
      unsigned bits = e1 & 0xffu, n = 1, val = e1 >> 16, third = 0;
      const unsigned e2 = single[i >> bits];
      if ((e2 & E_LITERAL) && (e2 & 0xffu) + bits <= LIT_TABLE_BITS) {
        unsigned e3;
        val |= (e2 >> 16) << 8; bits += e2 & 0xffu; n = 2;
        e3 = single[i >> bits];
        if ((e3 & E_LITERAL) && (e3 & 0xffu) + bits <= LIT_TABLE_BITS && (e3 >> 16) < 128u) { third = e3 >> 16; bits += e3 & 0xffu; n = 3; }
      }
      table[i] = (val << 16) | E_LITERAL | (third << 8) | ((n - 1u) << 4) | bits;
    }
  }
}

void
tji_init (tji_state *s)
{
  memset (s, 0, sizeof *s);
}

#define NEED_BITS(n)  do { while (bitcnt < (unsigned) (n)) { if (in >= in_end) goto out_of_input; bitbuf |= (unsigned long long) *in++ << bitcnt; bitcnt += 8; } } while (0)
#define DROP_BITS(n)  do { bitbuf >>= (n); bitcnt -= (unsigned) (n); } while (0)
#define BITS(n)       ((unsigned) (bitbuf & ((1ull << (n)) - 1ull)))

static unsigned long long
load64 (const unsigned char *p)
{
  unsigned long long v;
  memcpy (&v, p, 8);
  return v;                                             /* (little-endian hosts only: x86-64, the feeder's platform) */
}

/* Decode until the output is full, the input is exhausted (TJI_MORE_INPUT: impossible when the whole member is in
 * memory -- it means a truncated stream), or the final block has ended (TJI_DONE).  History: the 32 KiB in front of `out`
 * must hold the previous output whenever *out_pos of earlier calls was not 0 (hist_avail bytes are addressable there). */
int
tji_inflate (tji_state *s, const unsigned char *in0, size_t in_len, size_t *in_pos, unsigned char *out0, size_t out_cap, size_t *out_pos,
             size_t hist_avail)
{
  const unsigned char *in = in0 + *in_pos, *const in_end = in0 + in_len;
  unsigned char *out = out0 + *out_pos, *const out_end = out0 + out_cap;
  unsigned long long bitbuf = s->bitbuf;
  unsigned bitcnt = s->bitcnt;
  int rc = TJI_ERROR;

  for (;;) {
    if (s->phase == 0) {                                /* block header */
      unsigned type;
      if (s->last_block_done) { rc = TJI_DONE; goto save; }
      NEED_BITS (3);
      s->is_final = (int) BITS (1);
      type = (bitbuf >> 1) & 3u;
      DROP_BITS (3);
      if (type == 0) {                                  /* stored: skip to a byte boundary, LEN, NLEN */
        unsigned len, nlen;
        DROP_BITS (bitcnt & 7u);
        NEED_BITS (32);
        len = BITS (16); nlen = (unsigned) ((bitbuf >> 16) & 0xffffu);
        DROP_BITS (32);
        if ((len ^ nlen) != 0xffffu) goto bad;
        s->stored_left = len;
        s->phase = 1;
      }
      else if (type == 1) {                             /* fixed codes */
        unsigned char lens[N_LITLEN + N_DIST];
        int i;
        for (i = 0; i < 144; i++) lens[i] = 8;
        for (; i < 256; i++) lens[i] = 9;
        for (; i < 280; i++) lens[i] = 7;
        for (; i < 288; i++) lens[i] = 8;
        for (i = 0; i < 32; i++) lens[N_LITLEN + i] = 5;
        if (build_table (s->lit_table, LIT_TABLE_BITS, TJI_LIT_TABLE_CAP, lens, 288, 1)) goto bad;
        if (build_table (s->dist_table, DIST_TABLE_BITS, TJI_DIST_TABLE_CAP, lens + N_LITLEN, 32, 2)) goto bad;
        pair_literals (s->lit_table, s->lit_single);
        s->phase = 2;
      }
      else if (type == 2) {                             /* dynamic codes */
        unsigned hlit, hdist, hclen, i, n;
        unsigned char pre_lens[N_PRECODE], lens[N_LITLEN + N_DIST + 137];
        unsigned pre_table[1u << 7];
        NEED_BITS (14);
        hlit = BITS (5) + 257; hdist = (unsigned) ((bitbuf >> 5) & 31u) + 1; hclen = (unsigned) ((bitbuf >> 10) & 15u) + 4;
        DROP_BITS (14);
        if (hlit > 286 || hdist > 30) goto bad;
        memset (pre_lens, 0, sizeof pre_lens);
        for (i = 0; i < hclen; i++) { NEED_BITS (3); pre_lens[precode_order[i]] = (unsigned char) BITS (3); DROP_BITS (3); }
        if (build_table (pre_table, 7, 1 << 7, pre_lens, N_PRECODE, 0)) goto bad;
        n = 0;
        while (n < hlit + hdist) {
          unsigned e, sym;
          NEED_BITS (7 + 7);
          e = pre_table[BITS (7)];
          if (e & E_INVALID) goto bad;
          DROP_BITS (e & 0xffu);
          sym = e >> 16;
          if (sym < 16) lens[n++] = (unsigned char) sym;
          else {
            unsigned rep, val = 0;
            if (sym == 16) { if (!n) goto bad; val = lens[n - 1]; rep = 3 + BITS (2); DROP_BITS (2); }
            else if (sym == 17) { rep = 3 + BITS (3); DROP_BITS (3); }
            else { rep = 11 + BITS (7); DROP_BITS (7); }
            if (n + rep > hlit + hdist) goto bad;
            while (rep--) lens[n++] = (unsigned char) val;
          }
        }
        if (lens[256] == 0) goto bad;                   /* no end-of-block code */
        {
          unsigned char ll[N_LITLEN], dl[N_DIST];
          memset (ll, 0, sizeof ll); memset (dl, 0, sizeof dl);
          memcpy (ll, lens, hlit); memcpy (dl, lens + hlit, hdist);
          if (build_table (s->lit_table, LIT_TABLE_BITS, TJI_LIT_TABLE_CAP, ll, N_LITLEN, 1)) goto bad;
          if (build_table (s->dist_table, DIST_TABLE_BITS, TJI_DIST_TABLE_CAP, dl, N_DIST, 2)) goto bad;
        }
        pair_literals (s->lit_table, s->lit_single);
        s->phase = 2;
      }
      else goto bad;
    }

    if (s->phase == 1) {                                /* stored bytes: first what is left in the bit buffer (whole bytes) */
      while (s->stored_left && bitcnt >= 8) {
        if (out >= out_end) { rc = TJI_OUTPUT_FULL; goto save; }
        *out++ = (unsigned char) bitbuf; DROP_BITS (8); s->stored_left--;
      }
      if (s->stored_left) {
        size_t n = s->stored_left;
        if (n > (size_t) (in_end - in)) n = (size_t) (in_end - in);
        if (n > (size_t) (out_end - out)) n = (size_t) (out_end - out);
        memcpy (out, in, n); out += n; in += n; s->stored_left -= (unsigned) n;
        if (s->stored_left) { rc = (out >= out_end) ? TJI_OUTPUT_FULL : TJI_MORE_INPUT; goto save; }
      }
      s->phase = 0;
      if (s->is_final) s->last_block_done = 1;
      continue;
    }

    /* phase 2: Huffman-coded symbols.  A match that did not fit the output is finished first. */
    if (s->pending_len) {
      unsigned n = s->pending_len;
      const size_t d = s->pending_dist;
      while (n && out < out_end) { *out = *(out - d); out++; n--; }
      s->pending_len = n;
      if (n) { rc = TJI_OUTPUT_FULL; goto save; }
    }
    {
      const unsigned *const lt = s->lit_table, *const dt = s->dist_table;
      for (;;) {
        unsigned e, len, dist;
        /* fast loop: at least 16 input bytes and 320 output bytes to spare (one literal run + one longest match + slack) */
        while ((size_t) (in_end - in) >= 16 && (size_t) (out_end - out) >= 320) {
          bitbuf |= load64 (in) << bitcnt;                /* branch-free refill to >= 56 bits */
          in += (63u - bitcnt) >> 3;
          bitcnt |= 56u;
          e = lt[bitbuf & ((1u << LIT_TABLE_BITS) - 1u)];
#define PUT_LITERALS(e) do { const unsigned v_ = ((e) >> 16) | ((((e) >> 8) & 0x7fu) << 16); memcpy (out, &v_, 4); out += E_LITN (e); } while (0)
          if (e & E_LITERAL) {                          /* up to three look-ups (nine literals) per refill (3 x 11 bits <= 56) */
            DROP_BITS (e & 0xfu); PUT_LITERALS (e);
            e = lt[bitbuf & ((1u << LIT_TABLE_BITS) - 1u)];
            if (e & E_LITERAL) {
              DROP_BITS (e & 0xfu); PUT_LITERALS (e);
              e = lt[bitbuf & ((1u << LIT_TABLE_BITS) - 1u)];
              if (e & E_LITERAL) { DROP_BITS (e & 0xfu); PUT_LITERALS (e); continue; }
            }
          }
          if (e & E_SUBTABLE) {
            DROP_BITS (e & 0xffu);
            e = lt[(e >> 16) + BITS (E_EXTRA (e))];
            if (e & E_LITERAL) { DROP_BITS (e & 0xffu); *out++ = (unsigned char) (e >> 16); continue; }
          }
          if (e & (E_EOB | E_INVALID)) { if (e & E_INVALID) goto bad; DROP_BITS (e & 0xffu); goto block_done; }
          DROP_BITS (e & 0xffu);
          len = (e >> 16) + BITS (E_EXTRA (e)); DROP_BITS (E_EXTRA (e));
          /* bits used so far: <= 15 + 15 + 15 (literals) or 15 + 5; refill before the distance if fewer than 28 are left */
          if (bitcnt < 32u) { bitbuf |= load64 (in) << bitcnt; in += (63u - bitcnt) >> 3; bitcnt |= 56u; }
          e = dt[bitbuf & ((1u << DIST_TABLE_BITS) - 1u)];
          if (e & E_SUBTABLE) { DROP_BITS (e & 0xffu); e = dt[(e >> 16) + BITS (E_EXTRA (e))]; }
          if (e & E_INVALID) goto bad;
          DROP_BITS (e & 0xffu);
          dist = (e >> 16) + BITS (E_EXTRA (e)); DROP_BITS (E_EXTRA (e));
          if (dist > (size_t) (out - out0) + hist_avail) goto bad;     /* reaches before anything ever written */
          {
            const unsigned char *src = out - dist;
            unsigned char *const end = out + len;
            if (dist >= 8) {                            /* word copies (may write up to 7 bytes past the match: slack is there) */
              do { memcpy (out, src, 8); out += 8; src += 8; } while (out < end);
            }
            else if (dist == 1) memset (out, *src, len);
            else { while (out < end) *out++ = *src++; }
            out = end;
          }
        }
        /* careful loop: one symbol at a time, every bound checked */
        {
          const unsigned char *in_sym = in; const unsigned long long bb_sym = bitbuf; const unsigned bc_sym = bitcnt;
          int refill_ok = 1;
          while (bitcnt < 48u && in < in_end) { bitbuf |= (unsigned long long) *in++ << bitcnt; bitcnt += 8; }
          e = s->lit_single[bitbuf & ((1u << LIT_TABLE_BITS) - 1u)];
          if (e & E_SUBTABLE) {
            if (bitcnt < (e & 0xffu) + E_EXTRA (e)) refill_ok = 0;
            else { DROP_BITS (e & 0xffu); e = lt[(e >> 16) + BITS (E_EXTRA (e))]; }
          }
          if (refill_ok && !(e & E_INVALID) && bitcnt < (e & 0xffu)) refill_ok = 0;
          if (!refill_ok) { in = in_sym; bitbuf = bb_sym; bitcnt = bc_sym; rc = TJI_MORE_INPUT; goto save_symbol; }
          if (e & E_INVALID) goto bad;
          if (e & E_LITERAL) {
            if (out >= out_end) { in = in_sym; bitbuf = bb_sym; bitcnt = bc_sym; rc = TJI_OUTPUT_FULL; goto save_symbol; }
            DROP_BITS (e & 0xffu); *out++ = (unsigned char) (e >> 16);
            continue;
          }
          if (e & E_EOB) { DROP_BITS (e & 0xffu); goto block_done; }
          DROP_BITS (e & 0xffu);
          if (bitcnt < E_EXTRA (e)) { in = in_sym; bitbuf = bb_sym; bitcnt = bc_sym; rc = TJI_MORE_INPUT; goto save_symbol; }
          len = (e >> 16) + BITS (E_EXTRA (e)); DROP_BITS (E_EXTRA (e));
          while (bitcnt < 48u && in < in_end) { bitbuf |= (unsigned long long) *in++ << bitcnt; bitcnt += 8; }
          e = dt[bitbuf & ((1u << DIST_TABLE_BITS) - 1u)];
          if (e & E_SUBTABLE) {
            if (bitcnt < (e & 0xffu) + E_EXTRA (e)) { in = in_sym; bitbuf = bb_sym; bitcnt = bc_sym; rc = TJI_MORE_INPUT; goto save_symbol; }
            DROP_BITS (e & 0xffu); e = dt[(e >> 16) + BITS (E_EXTRA (e))];
          }
          if (e & E_INVALID) { if (bitcnt < 15u && in >= in_end) { in = in_sym; bitbuf = bb_sym; bitcnt = bc_sym; rc = TJI_MORE_INPUT; goto save_symbol; } goto bad; }
          if (bitcnt < (e & 0xffu) + E_EXTRA (e)) { in = in_sym; bitbuf = bb_sym; bitcnt = bc_sym; rc = TJI_MORE_INPUT; goto save_symbol; }
          DROP_BITS (e & 0xffu);
          dist = (e >> 16) + BITS (E_EXTRA (e)); DROP_BITS (E_EXTRA (e));
          if (dist > (size_t) (out - out0) + hist_avail) goto bad;
          while (len && out < out_end) { *out = *(out - dist); out++; len--; }
          if (len) { s->pending_len = len; s->pending_dist = dist; rc = TJI_OUTPUT_FULL; goto save; }
        }
      }
    block_done:
      s->phase = 0;
      if (s->is_final) s->last_block_done = 1;
      continue;
    save_symbol:
      goto save;
    }
  }

out_of_input:
  /* (from the header paths: the caller hands whole members, so this is a truncated stream and there is no resuming) */
  rc = TJI_MORE_INPUT;
  goto save;
bad:
  rc = TJI_ERROR;
save:
  /* give whole bytes of look-ahead back to the input so that the byte position after a member is exact */
  while (bitcnt >= 8 && in > in0 && rc == TJI_DONE) { bitcnt -= 8; in--; bitbuf &= (1ull << bitcnt) - 1ull; }
  s->bitbuf = bitbuf; s->bitcnt = bitcnt;
  *in_pos = (size_t) (in - in0);
  *out_pos = (size_t) (out - out0);
  return rc;
}

/* ---- entering a DEFLATE stream in the middle (feeder.c: a one-member .gz file on more than one thread) -------------
 * A deflate stream can only be decoded from a block start, and from there only up to references into the 32 KiB of
 * output in front of it.  So: (1) tjp_find_block looks for a bit position at which a dynamic-Huffman block header
 * parses, its codes are complete, the block decodes to text and is followed by something that looks like a block
 * header again; (2) tjp_decode decodes from such a position with the window in front UNKNOWN: its output is 16-bit
 * symbols, a byte or 256 + i for "byte i of the unknown window" (a match that reaches into the window copies those
 * markers, a match into the segment's own output copies whatever stands there); (3) tjp_resolve turns the symbols
 * into bytes once the window is known.  Whether a position found by (1) really was a block start is not a matter of
 * trust: the decoder of the stretch in front must arrive at exactly that bit position at a block boundary -- then the
 * bytes are the bytes a decoder running from the member's start would have produced -- and otherwise the speculative
 * stretch is thrown away (feeder.c).  The member's CRC-32 and size are checked as for any other member. */
#define TJP_WIN 32768u

static unsigned long long
tjp_peek (const unsigned char *z, size_t zn, size_t bitpos)
{ /* >= 57 bits of the stream from bitpos on (zeros behind its end) */
  const size_t b = bitpos >> 3;
  unsigned long long v = 0;
  if (b + 8 <= zn) memcpy (&v, z + b, 8);
  else { size_t i; for (i = 0; b + i < zn && i < 8; i++) v |= (unsigned long long) z[b + i] << (8 * i); }
  return v >> (bitpos & 7u);
}

typedef struct { unsigned lit[TJI_LIT_TABLE_CAP], dist[TJI_DIST_TABLE_CAP], single[2048]; } tjp_tables;   /* lit: primary entries hold up to three literals (pair_literals), single: one symbol per entry */

/* block header at *pos: 0 = stored (then *stored_len, *pos at its first byte), 1 / 2 = coded (tables built, *pos at the first
 * symbol), -1 = not a valid header.  *is_final = BFINAL. */
static int
tjp_header (const unsigned char *z, size_t zn, size_t *pos, tjp_tables *t, int *is_final, unsigned *stored_len)
{
  size_t p = *pos;
  unsigned long long b;
  unsigned type;
  if ((p >> 3) >= zn) return -1;
  b = tjp_peek (z, zn, p);
  *is_final = (int) (b & 1u);
  type = (unsigned) (b >> 1) & 3u;
  p += 3;
  if (type == 0) {
    unsigned len, nlen;
    p = (p + 7u) & ~(size_t) 7u;
    if ((p >> 3) + 4 > zn) return -1;
    len = (unsigned) z[p >> 3] | ((unsigned) z[(p >> 3) + 1] << 8);
    nlen = (unsigned) z[(p >> 3) + 2] | ((unsigned) z[(p >> 3) + 3] << 8);
    if ((len ^ nlen) != 0xffffu) return -1;
    p += 32;
    if ((p >> 3) + len > zn) return -1;
    *stored_len = len; *pos = p;
    return 0;
  }
  if (type == 1) {
    unsigned char lens[N_LITLEN + N_DIST];
    int i;
    for (i = 0; i < 144; i++) lens[i] = 8;
    for (; i < 256; i++) lens[i] = 9;
    for (; i < 280; i++) lens[i] = 7;
    for (; i < 288; i++) lens[i] = 8;
    for (i = 0; i < 32; i++) lens[N_LITLEN + i] = 5;
    if (build_table (t->lit, LIT_TABLE_BITS, TJI_LIT_TABLE_CAP, lens, 288, 1)) return -1;
    if (build_table (t->dist, DIST_TABLE_BITS, TJI_DIST_TABLE_CAP, lens + N_LITLEN, 32, 2)) return -1;
    pair_literals (t->lit, t->single);
    *pos = p;
    return 1;
  }
  if (type == 2) {
    unsigned hlit, hdist, hclen, i, n;
    unsigned char pre_lens[N_PRECODE], lens[N_LITLEN + N_DIST + 137], ll[N_LITLEN], dl[N_DIST];
    unsigned pre_table[1u << 7];
    b = tjp_peek (z, zn, p);
    hlit = (unsigned) (b & 31u) + 257; hdist = (unsigned) ((b >> 5) & 31u) + 1; hclen = (unsigned) ((b >> 10) & 15u) + 4;
    p += 14;
    if (hlit > 286 || hdist > 30) return -1;
    memset (pre_lens, 0, sizeof pre_lens);
    b = tjp_peek (z, zn, p);                            /* 19 x 3 = 57 bits at most */
    for (i = 0; i < hclen; i++) { pre_lens[precode_order[i]] = (unsigned char) (b & 7u); b >>= 3; }
    p += 3 * hclen;
    if (build_table (pre_table, 7, 1 << 7, pre_lens, N_PRECODE, 0)) return -1;
    n = 0;
    while (n < hlit + hdist) {
      unsigned e, sym;
      if ((p >> 3) >= zn) return -1;
      b = tjp_peek (z, zn, p);
      e = pre_table[b & 127u];
      if (e & E_INVALID) return -1;
      p += e & 0xffu; b >>= e & 0xffu;
      sym = e >> 16;
      if (sym < 16) lens[n++] = (unsigned char) sym;
      else {
        unsigned rep, val = 0;
        if (sym == 16) { if (!n) return -1; val = lens[n - 1]; rep = 3 + (unsigned) (b & 3u); p += 2; }
        else if (sym == 17) { rep = 3 + (unsigned) (b & 7u); p += 3; }
        else { rep = 11 + (unsigned) (b & 127u); p += 7; }
        if (n + rep > hlit + hdist) return -1;
        while (rep--) lens[n++] = (unsigned char) val;
      }
    }
    if (lens[256] == 0) return -1;
    memset (ll, 0, sizeof ll); memset (dl, 0, sizeof dl);
    memcpy (ll, lens, hlit); memcpy (dl, lens + hlit, hdist);
    if (build_table (t->lit, LIT_TABLE_BITS, TJI_LIT_TABLE_CAP, ll, N_LITLEN, 1)) return -1;
    if (build_table (t->dist, DIST_TABLE_BITS, TJI_DIST_TABLE_CAP, dl, N_DIST, 2)) return -1;
    pair_literals (t->lit, t->single);
    *pos = p;
    return 2;
  }
  return -1;
}

/* The cheap part of tjp_header's test for a dynamic block, for the search: BFINAL = 0, BTYPE = 2, counts in range, the code
 * length code complete.  Nearly every wrong position fails here. */
static int
tjp_header_plausible (const unsigned char *z, size_t zn, size_t p)
{
  const unsigned long long b = tjp_peek (z, zn, p);
  unsigned hclen, i, left = 128;                        /* Kraft sum in units of 2^-7 */
  unsigned long long c;
  if ((b & 7u) != 4u) return 0;                         /* BFINAL 0, BTYPE 10 */
  if (((b >> 3) & 31u) > 29u || ((b >> 8) & 31u) > 29u) return 0;
  hclen = (unsigned) ((b >> 13) & 15u) + 4;
  c = tjp_peek (z, zn, p + 17);
  for (i = 0; i < hclen; i++) { const unsigned l = (unsigned) (c & 7u); c >>= 3; if (l) { const unsigned w = 128u >> l; if (w > left) return 0; left -= w; } }
  return left == 0;
}

/* one coded block from *pos (first symbol) to behind its end-of-block code, for the search: nothing is written; -2 at the
 * first literal that is no text byte, 1 once more than max_out bytes would have come out ("looks fine so far"), 0 at the
 * block's end (*n = bytes it holds), -1 invalid */
static int
tjp_block_text (const unsigned char *z, size_t zn, size_t *pos, const tjp_tables *t, size_t *n, size_t max_out)
{
  size_t p = *pos, k = 0;
  const size_t end_bit = zn * 8;
  for (;;) {
    unsigned long long b;
    unsigned e;
    if (p >= end_bit) return -1;
    b = tjp_peek (z, zn, p);
    e = t->single[b & ((1u << LIT_TABLE_BITS) - 1u)];
    if (e & E_SUBTABLE) { p += e & 0xffu; b >>= e & 0xffu; e = t->lit[(e >> 16) + (unsigned) (b & ((1u << E_EXTRA (e)) - 1u))]; }
    if (e & E_INVALID) return -1;
    p += e & 0xffu; b >>= e & 0xffu;
    if (e & E_LITERAL) {
      const unsigned c = e >> 16;
      if (!((c >= 32u && c < 127u) || c == '\n' || c == '\r' || c == '\t')) return -2;
      if (++k > max_out) { *pos = p; *n = k; return 1; }
      continue;
    }
    if (e & E_EOB) { *pos = p; *n = k; return 0; }
    k += (e >> 16) + (unsigned) (b & ((1u << E_EXTRA (e)) - 1u));
    p += E_EXTRA (e); b >>= E_EXTRA (e);
    e = t->dist[b & ((1u << DIST_TABLE_BITS) - 1u)];
    if (e & E_SUBTABLE) { p += e & 0xffu; b >>= e & 0xffu; e = t->dist[(e >> 16) + (unsigned) (b & ((1u << E_EXTRA (e)) - 1u))]; }
    if (e & E_INVALID) return -1;
    p += (e & 0xffu) + E_EXTRA (e);
    if (k > max_out) { *pos = p; *n = k; return 1; }
  }
}

/* the same block decoded: symbols appended at out[*n] (out[-TJP_WIN .. -1] is the window in front); `room` symbols fit.
 * 0 at the block's end, -1 invalid, -3 out of room (nothing is kept: the caller comes again with more) */
static int
tjp_block (const unsigned char *z, size_t zn, size_t *pos, const tjp_tables *t, unsigned short *out, size_t *n, size_t room)
{
  size_t p = *pos, k = *n;
  const size_t end_bit = zn * 8;
  for (;;) {
    unsigned long long b;
    unsigned e, len, dist;
    if (p >= end_bit) return -1;
    if (k + 264 > room) return -3;                      /* (three literals or the longest match, written without further checks) */
    b = tjp_peek (z, zn, p);
    e = t->lit[b & ((1u << LIT_TABLE_BITS) - 1u)];
    if (e & E_LITERAL) {                                /* a primary entry of literals: up to three, first | second << 8 in the value, the third in bits
                                                         * 8-14 -- which are the other flags' bits: E_LITERAL is looked at first, as in the serial loop */
      out[k] = (unsigned short) ((e >> 16) & 0xffu); out[k + 1] = (unsigned short) (e >> 24); out[k + 2] = (unsigned short) ((e >> 8) & 0x7fu);
      k += E_LITN (e); p += e & 0xfu; b >>= e & 0xfu;
      e = t->lit[b & ((1u << LIT_TABLE_BITS) - 1u)];    /* (57 bits were looked at: 11 are gone at most, 15 + 5 + 15 + 13 may follow) */
      if (e & E_LITERAL) {
        out[k] = (unsigned short) ((e >> 16) & 0xffu); out[k + 1] = (unsigned short) (e >> 24); out[k + 2] = (unsigned short) ((e >> 8) & 0x7fu);
        k += E_LITN (e); p += e & 0xfu;
        continue;
      }
    }
    if (e & E_SUBTABLE) { p += e & 0xffu; b >>= e & 0xffu; e = t->lit[(e >> 16) + (unsigned) (b & ((1u << E_EXTRA (e)) - 1u))]; }
    if (e & E_INVALID) return -1;
    p += e & 0xffu; b >>= e & 0xffu;
    if (e & E_LITERAL) { out[k++] = (unsigned short) (e >> 16); continue; }
    if (e & E_EOB) { *pos = p; *n = k; return 0; }
    len = (e >> 16) + (unsigned) (b & ((1u << E_EXTRA (e)) - 1u));
    p += E_EXTRA (e); b >>= E_EXTRA (e);
    e = t->dist[b & ((1u << DIST_TABLE_BITS) - 1u)];
    if (e & E_SUBTABLE) { p += e & 0xffu; b >>= e & 0xffu; e = t->dist[(e >> 16) + (unsigned) (b & ((1u << E_EXTRA (e)) - 1u))]; }
    if (e & E_INVALID) return -1;
    p += e & 0xffu; b >>= e & 0xffu;
    dist = (e >> 16) + (unsigned) (b & ((1u << E_EXTRA (e)) - 1u));
    p += E_EXTRA (e);
    {
      unsigned short *d = out + k;
      const unsigned short *src = d - dist;
      unsigned i;
      if (dist >= 4) { for (i = 0; i < len; i += 4) memcpy (d + i, src + i, 8); }       /* (four symbols at a time; up to three past the match: room is there) */
      else for (i = 0; i < len; i++) d[i] = src[i];
    }
    k += len;
  }
}

size_t
tjp_find_block (const unsigned char *z, size_t zn, size_t from_bit, size_t limit_bit)
{
  tjp_tables *t = (tjp_tables *) malloc (sizeof (tjp_tables));
  size_t p, found = (size_t) -1;
  if (!t) return found;
  if (limit_bit > zn * 8) limit_bit = zn * 8;
  for (p = from_bit; p + 64 < limit_bit; p++) {
    size_t q = p, n = 0;
    int fin, rc;
    unsigned sl;
    if (!tjp_header_plausible (z, zn, p)) continue;
    if (tjp_header (z, zn, &q, t, &fin, &sl) != 2) continue;
    rc = tjp_block_text (z, zn, &q, t, &n, 1u << 20);
    if (rc == 1) { found = p; break; }                  /* a megabyte of text out of one block: good enough */
    if (rc != 0 || n < 64) continue;
    /* the block ended: what follows must look like a block header too (a dynamic one: plausible; stored: LEN / NLEN agree;
     * a fixed one cannot be told from noise, so such a position is passed over -- the next dynamic block will do) */
    {
      const unsigned long long b = tjp_peek (z, zn, q);
      const unsigned type = (unsigned) (b >> 1) & 3u;
      size_t q2 = q;
      if (type == 2 && (b & 1u) == 0 && tjp_header_plausible (z, zn, q)) { found = p; break; }
      if ((type == 0 || (type == 2 && (b & 1u))) && tjp_header (z, zn, &q2, t, &fin, &sl) >= 0) { found = p; break; }
    }
  }
  free (t);
  return found;
}

int
tjp_decode (const unsigned char *z, size_t zn, size_t start_bit, size_t stop_bit, tjp_segment *seg)
{
  tjp_tables *t = (tjp_tables *) malloc (sizeof (tjp_tables));
  size_t p = start_bit, n = 0, cap = seg->cap;
  unsigned short *buf = seg->buf;                       /* buf[0 .. TJP_WIN) = the window's markers, symbols from buf[TJP_WIN] on */
  int rc = -1;
  seg->n = 0; seg->end_bit = start_bit; seg->is_final = 0;
  if (!t) return -1;
  if (!buf || cap < (1u << 20)) {
    cap = (size_t) 8 << 20;
    buf = (unsigned short *) realloc (buf, (TJP_WIN + cap) * sizeof (unsigned short));
    if (!buf) { free (t); seg->buf = NULL; seg->cap = 0; return -1; }
  }
  { unsigned i; for (i = 0; i < TJP_WIN; i++) buf[i] = (unsigned short) (256u + i); }
  for (;;) {
    int fin, kind;
    unsigned sl = 0;
    if (p >= stop_bit) { rc = 0; break; }               /* at a block boundary at or behind the target */
    if (n + 65536u + 258u + 1024u > cap) {               /* room for a stored block or a long stretch of coded output */
      unsigned short *nb;
      if (cap >= ((size_t) 1 << 29)) break;               /* (a stretch that will not end: give up, what was decoded so far stands) */
      cap *= 2;
      nb = (unsigned short *) realloc (buf, (TJP_WIN + cap) * sizeof (unsigned short));
      if (!nb) break;
      buf = nb;
    }
    kind = tjp_header (z, zn, &p, t, &fin, &sl);
    if (kind < 0) break;
    if (kind == 0) {
      unsigned i;
      const unsigned char *src = z + (p >> 3);
      for (i = 0; i < sl; i++) buf[TJP_WIN + n + i] = src[i];
      n += sl; p += (size_t) sl * 8;
    }
    else {
      /* (a coded block can be longer than the room at hand -- zlib ends one after 16 K symbols, others need not: then more
       * room, and the block again from its first symbol) */
      for (;;) {
        size_t q = p, k = n;
        const int r = tjp_block (z, zn, &q, t, buf + TJP_WIN, &k, cap);
        if (r == 0) { p = q; n = k; break; }
        if (r == -3 && cap < ((size_t) 1 << 28)) {       /* (a block of more than 256 M bytes: not for this decoder -- the caller's single one streams it) */
          unsigned short *nb;
          cap *= 2;
          nb = (unsigned short *) realloc (buf, (TJP_WIN + cap) * sizeof (unsigned short));
          if (!nb) { kind = -1; break; }
          buf = nb;
          continue;
        }
        kind = -1; break;
      }
      if (kind < 0) break;
    }
    seg->end_bit = p; seg->n = n;
    if (fin) { seg->is_final = 1; rc = 0; break; }
  }
  seg->buf = buf; seg->cap = cap;
  if (rc == 0) { seg->n = n; seg->end_bit = p; }        /* (on an error: the last block boundary reached, set in the loop) */
  free (t);
  return rc;
}

int
tjp_resolve (const unsigned short *sym, size_t n, const unsigned char *window, size_t win_valid, unsigned char *out)
{ /* window[TJP_WIN - win_valid .. TJP_WIN) = the bytes in front; marker 256 + i = window[i] */
  size_t i = 0;
  unsigned bad = 0;
  const size_t first_valid = TJP_WIN - win_valid;
  if (win_valid == TJP_WIN && n >= 65536u) {
    /* A whole window in front and plenty to do: one table for bytes and markers alike, one look-up per symbol and no
     * branch.  (Markers do not die out behind a stretch's first 32 KiB: a match copies them, and text with a period --
     * read names, quality strings -- keeps copying what it copied before, so most of a stretch's symbols may be markers.) */
    unsigned char *lut = (unsigned char *) malloc (256u + TJP_WIN);
    if (lut) {
      unsigned v;
      for (v = 0; v < 256u; v++) lut[v] = (unsigned char) v;
      memcpy (lut + 256, window, TJP_WIN);
      for (; i + 4 <= n; i += 4) { out[i] = lut[sym[i]]; out[i + 1] = lut[sym[i + 1]]; out[i + 2] = lut[sym[i + 2]]; out[i + 3] = lut[sym[i + 3]]; }
      for (; i < n; i++) out[i] = lut[sym[i]];
      free (lut);
      return 0;
    }
  }
#if defined(__SSE2__)
  /* sixteen symbols at a time while they are all plain bytes (nearly everything behind a stretch's first 32 KiB) */
  for (; i + 16 <= n; i += 16) {
    const __m128i a = _mm_loadu_si128 ((const __m128i *) (sym + i)), b = _mm_loadu_si128 ((const __m128i *) (sym + i + 8));
    const __m128i hi = _mm_or_si128 (_mm_srli_epi16 (a, 8), _mm_srli_epi16 (b, 8));
    if (_mm_movemask_epi8 (_mm_cmpeq_epi16 (hi, _mm_setzero_si128 ())) == 0xffff) { _mm_storeu_si128 ((__m128i *) (out + i), _mm_packus_epi16 (a, b)); continue; }
    {
      size_t j;
      for (j = i; j < i + 16; j++) {
        const unsigned v = sym[j];
        if (v < 256u) out[j] = (unsigned char) v;
        else { const unsigned w = v - 256u; bad |= (unsigned) (w < first_valid); out[j] = window[w]; }
      }
    }
  }
#endif
  for (; i < n; i++) {
    const unsigned v = sym[i];
    if (v < 256u) out[i] = (unsigned char) v;
    else { const unsigned w = v - 256u; bad |= (unsigned) (w < first_valid); out[i] = window[w]; }
  }
  return bad ? -1 : 0;
}

/* ---- CRC-32 (the gzip polynomial, reflected 0xEDB88320), slicing by 16: zlib 1.2.11's crc32() does 1 GB/s, which next to
 * this inflater is a third of the time per byte.  Same values as zlib's crc32 (tests compare them). ---- */
static unsigned tji_crc_tab[16][256];
static pthread_once_t tji_crc_once = PTHREAD_ONCE_INIT;

static void
tji_crc_init (void)
{
  unsigned i, j;
  for (i = 0; i < 256; i++) {
    unsigned c = i;
    for (j = 0; j < 8; j++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
    tji_crc_tab[0][i] = c;
  }
  for (i = 0; i < 256; i++) for (j = 1; j < 16; j++) tji_crc_tab[j][i] = (tji_crc_tab[j - 1][i] >> 8) ^ tji_crc_tab[0][tji_crc_tab[j - 1][i] & 0xffu];
}

#if defined(__x86_64__) && defined(__GNUC__)
#include <immintrin.h>
/* Carry-less-multiply folding (Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ Instruction",
 * Intel 2009; constants for the reflected gzip polynomial as every implementation of it uses them): 64 bytes per step
 * on four lanes, folded to 128 bits, then a Barrett reduction.  n is a multiple of 16 and at least 64; crc is the raw
 * (inverted) register value.  Checked against the table version at start-up (tji_crc_init) and left off if the CPU
 * lacks the instruction or the check fails. */
__attribute__ ((target ("pclmul,sse4.1")))
static unsigned
tji_crc_clmul (const unsigned char *buf, size_t len, unsigned crc)
{
  static const unsigned long long k1k2[2] __attribute__ ((aligned (16))) = {0x0154442bd4ull, 0x01c6e41596ull};
  static const unsigned long long k3k4[2] __attribute__ ((aligned (16))) = {0x01751997d0ull, 0x00ccaa009eull};
  static const unsigned long long k5k0[2] __attribute__ ((aligned (16))) = {0x0163cd6124ull, 0x0000000000ull};
  static const unsigned long long poly[2] __attribute__ ((aligned (16))) = {0x01db710641ull, 0x01f7011641ull};
  __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
  x1 = _mm_loadu_si128 ((const __m128i *) (buf + 0x00)); x2 = _mm_loadu_si128 ((const __m128i *) (buf + 0x10));
  x3 = _mm_loadu_si128 ((const __m128i *) (buf + 0x20)); x4 = _mm_loadu_si128 ((const __m128i *) (buf + 0x30));
  x1 = _mm_xor_si128 (x1, _mm_cvtsi32_si128 ((int) crc));
  x0 = _mm_load_si128 ((const __m128i *) k1k2);
  buf += 64; len -= 64;
  while (len >= 64) {
    x5 = _mm_clmulepi64_si128 (x1, x0, 0x00); x6 = _mm_clmulepi64_si128 (x2, x0, 0x00);
    x7 = _mm_clmulepi64_si128 (x3, x0, 0x00); x8 = _mm_clmulepi64_si128 (x4, x0, 0x00);
    x1 = _mm_clmulepi64_si128 (x1, x0, 0x11); x2 = _mm_clmulepi64_si128 (x2, x0, 0x11);
    x3 = _mm_clmulepi64_si128 (x3, x0, 0x11); x4 = _mm_clmulepi64_si128 (x4, x0, 0x11);
    y5 = _mm_loadu_si128 ((const __m128i *) (buf + 0x00)); y6 = _mm_loadu_si128 ((const __m128i *) (buf + 0x10));
    y7 = _mm_loadu_si128 ((const __m128i *) (buf + 0x20)); y8 = _mm_loadu_si128 ((const __m128i *) (buf + 0x30));
    x1 = _mm_xor_si128 (_mm_xor_si128 (x1, x5), y5); x2 = _mm_xor_si128 (_mm_xor_si128 (x2, x6), y6);
    x3 = _mm_xor_si128 (_mm_xor_si128 (x3, x7), y7); x4 = _mm_xor_si128 (_mm_xor_si128 (x4, x8), y8);
    buf += 64; len -= 64;
  }
  x0 = _mm_load_si128 ((const __m128i *) k3k4);
  x5 = _mm_clmulepi64_si128 (x1, x0, 0x00); x1 = _mm_clmulepi64_si128 (x1, x0, 0x11); x1 = _mm_xor_si128 (_mm_xor_si128 (x1, x2), x5);
  x5 = _mm_clmulepi64_si128 (x1, x0, 0x00); x1 = _mm_clmulepi64_si128 (x1, x0, 0x11); x1 = _mm_xor_si128 (_mm_xor_si128 (x1, x3), x5);
  x5 = _mm_clmulepi64_si128 (x1, x0, 0x00); x1 = _mm_clmulepi64_si128 (x1, x0, 0x11); x1 = _mm_xor_si128 (_mm_xor_si128 (x1, x4), x5);
  while (len >= 16) {
    x2 = _mm_loadu_si128 ((const __m128i *) buf);
    x5 = _mm_clmulepi64_si128 (x1, x0, 0x00); x1 = _mm_clmulepi64_si128 (x1, x0, 0x11); x1 = _mm_xor_si128 (_mm_xor_si128 (x1, x2), x5);
    buf += 16; len -= 16;
  }
  x2 = _mm_clmulepi64_si128 (x1, x0, 0x10);
  x3 = _mm_setr_epi32 (~0, 0, ~0, 0);
  x1 = _mm_srli_si128 (x1, 8);
  x1 = _mm_xor_si128 (x1, x2);
  x0 = _mm_loadl_epi64 ((const __m128i *) k5k0);
  x2 = _mm_srli_si128 (x1, 4);
  x1 = _mm_and_si128 (x1, x3);
  x1 = _mm_clmulepi64_si128 (x1, x0, 0x00);
  x1 = _mm_xor_si128 (x1, x2);
  x0 = _mm_load_si128 ((const __m128i *) poly);
  x2 = _mm_and_si128 (x1, x3);
  x2 = _mm_clmulepi64_si128 (x2, x0, 0x10);
  x2 = _mm_and_si128 (x2, x3);
  x2 = _mm_clmulepi64_si128 (x2, x0, 0x00);
  x1 = _mm_xor_si128 (x1, x2);
  return (unsigned) _mm_extract_epi32 (x1, 1);
}
#define TJI_HAVE_CLMUL 1
#else
#define TJI_HAVE_CLMUL 0
#endif
static int tji_crc_use_clmul = 0;

static unsigned
tji_crc_tables (unsigned crc, const unsigned char *p, size_t n)     /* crc: raw register value in and out */
{
  while (n && ((size_t) p & 7u)) { crc = (crc >> 8) ^ tji_crc_tab[0][(crc ^ *p++) & 0xffu]; n--; }
  while (n >= 16) {
    unsigned long long a = load64 (p), b = load64 (p + 8);
    const unsigned a0 = (unsigned) a ^ crc, a1 = (unsigned) (a >> 32), b0 = (unsigned) b, b1 = (unsigned) (b >> 32);
    crc = tji_crc_tab[15][a0 & 0xffu] ^ tji_crc_tab[14][(a0 >> 8) & 0xffu] ^ tji_crc_tab[13][(a0 >> 16) & 0xffu] ^ tji_crc_tab[12][a0 >> 24]
        ^ tji_crc_tab[11][a1 & 0xffu] ^ tji_crc_tab[10][(a1 >> 8) & 0xffu] ^ tji_crc_tab[9][(a1 >> 16) & 0xffu] ^ tji_crc_tab[8][a1 >> 24]
        ^ tji_crc_tab[7][b0 & 0xffu] ^ tji_crc_tab[6][(b0 >> 8) & 0xffu] ^ tji_crc_tab[5][(b0 >> 16) & 0xffu] ^ tji_crc_tab[4][b0 >> 24]
        ^ tji_crc_tab[3][b1 & 0xffu] ^ tji_crc_tab[2][(b1 >> 8) & 0xffu] ^ tji_crc_tab[1][(b1 >> 16) & 0xffu] ^ tji_crc_tab[0][b1 >> 24];
    p += 16; n -= 16;
  }
  while (n--) crc = (crc >> 8) ^ tji_crc_tab[0][(crc ^ *p++) & 0xffu];
  return crc;
}

static void
tji_crc_setup (void)
{
  unsigned i, j;
  tji_crc_init ();
  (void) i; (void) j;
#if TJI_HAVE_CLMUL
  if (__builtin_cpu_supports ("pclmul") && __builtin_cpu_supports ("sse4.1")) {   /* trusted only after it has agreed with the tables */
    unsigned char t[64 * 5 + 16];
    unsigned seed = 0x12345678u, ok = 1, len;
    for (i = 0; i < sizeof t; i++) { seed = seed * 1664525u + 1013904223u; t[i] = (unsigned char) (seed >> 24); }
    for (len = 64; len <= sizeof t && ok; len += 16)
      for (j = 0; j < 3 && ok; j++) {
        const unsigned start = j == 0 ? 0xffffffffu : j == 1 ? 0u : 0xdeadbeefu;
        if (tji_crc_clmul (t, len, start) != tji_crc_tables (start, t, len)) ok = 0;
      }
    tji_crc_use_clmul = (int) ok;
  }
#endif
}

unsigned
tji_crc32 (unsigned crc, const unsigned char *p, size_t n)
{
  pthread_once (&tji_crc_once, tji_crc_setup);
  crc = ~crc;
#if TJI_HAVE_CLMUL
  if (tji_crc_use_clmul && n >= 64) {
    const size_t m = n & ~(size_t) 15;
    crc = tji_crc_clmul (p, m, crc);
    p += m; n -= m;
  }
#endif
  return ~tji_crc_tables (crc, p, n);
}
