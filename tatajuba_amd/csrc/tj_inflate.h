/* tj_inflate.h -- raw DEFLATE (RFC 1951) decoder of the feeder (feeder.c); see tj_inflate.c. */
#ifndef TATAJUBA_AMD_TJ_INFLATE_H
#define TATAJUBA_AMD_TJ_INFLATE_H
#include <stddef.h>

#define TJI_LIT_TABLE_CAP  (2048 + 1024)
#define TJI_DIST_TABLE_CAP (256 + 512)

enum { TJI_DONE = 0, TJI_OUTPUT_FULL = 1, TJI_MORE_INPUT = 2, TJI_ERROR = -1 };

typedef struct
{
  unsigned long long bitbuf;
  unsigned bitcnt;
  int phase;                    /* 0 block header, 1 stored bytes, 2 coded symbols */
  int is_final, last_block_done;
  unsigned stored_left;
  unsigned pending_len;         /* rest of a match that did not fit the output buffer */
  size_t pending_dist;
  unsigned lit_table[TJI_LIT_TABLE_CAP];   /* primary entries may hold up to three literals (fast loop) */
  unsigned lit_single[2048];               /* the primary table with one symbol per entry (careful loop) */
  unsigned dist_table[TJI_DIST_TABLE_CAP];
} tji_state;

void tji_init (tji_state *s);

/* Inflate from in[*in_pos, in_len) to out[*out_pos, out_cap); both positions are advanced.  TJI_DONE: the final block
 * has ended, *in_pos is the first byte after the stream.  TJI_OUTPUT_FULL: call again with more room -- another buffer
 * will do if the 32 KiB (or all, if less) of output before it are copied in front of it; hist_avail = valid bytes in
 * front of out[0].  TJI_MORE_INPUT: the input ended inside the stream (truncated, when the caller handed all there is).
 * TJI_ERROR: not a DEFLATE stream. */
int tji_inflate (tji_state *s, const unsigned char *in, size_t in_len, size_t *in_pos, unsigned char *out, size_t out_cap, size_t *out_pos,
                 size_t hist_avail);

/* ---- entering a stream in the middle (see tj_inflate.c): find a block start, decode with the window in front unknown
 * (16-bit symbols: a byte, or 256 + i = byte i of that 32 KiB window), resolve once the window is known ---- */
typedef struct
{
  unsigned short *buf;          /* 32768 window markers, then the symbols; (re)allocated by tjp_decode, freed by the caller */
  size_t cap, n;                /* symbols that fit / decoded */
  size_t end_bit;               /* the block boundary the decoder stopped at */
  int is_final;                 /* ... which was the end of the final block */
} tjp_segment;
#define TJP_SYMBOLS(seg) ((seg)->buf + 32768)

/* first bit position in [from_bit, limit_bit) that passes for the start of a dynamic-Huffman block of text; (size_t) -1: none */
size_t tjp_find_block (const unsigned char *z, size_t zn, size_t from_bit, size_t limit_bit);
/* decode block after block from start_bit (a block start) and stop at the first block boundary >= stop_bit, or behind the
 * final block.  0: stopped at seg->end_bit; -1: invalid data (or out of memory) -- seg->n / end_bit then describe the
 * blocks that did decode. */
int tjp_decode (const unsigned char *z, size_t zn, size_t start_bit, size_t stop_bit, tjp_segment *seg);
/* symbols -> bytes; window[32768 - win_valid, 32768) are the bytes in front.  -1: a symbol points in front of them. */
int tjp_resolve (const unsigned short *sym, size_t n, const unsigned char *window, size_t win_valid, unsigned char *out);

/* CRC-32 as in gzip (same values as zlib's crc32(); start with crc = 0) */
unsigned tji_crc32 (unsigned crc, const unsigned char *p, size_t n);

#endif
