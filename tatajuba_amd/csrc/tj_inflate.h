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

/* CRC-32 as in gzip (same values as zlib's crc32(); start with crc = 0) */
unsigned tji_crc32 (unsigned crc, const unsigned char *p, size_t n);

#endif
