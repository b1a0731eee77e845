/* feeder.h -- multi-threaded host feeder for FASTA/FASTQ files (plain, gzip, BGZF); see feeder.c. */
#ifndef TATAJUBA_AMD_FEEDER_H
#define TATAJUBA_AMD_FEEDER_H
#include <stddef.h>

#define TJF_MAX_THREADS 32

typedef struct
{
  void *ctx;
  void *(*alloc) (void *ctx, size_t bytes);                 /* output buffers (pinned memory when the sink is the GPU) */
  void (*release) (void *ctx, void *p);
  int (*put) (void *ctx, const unsigned char *stream, size_t n_bytes, long n_reads);  /* reads + '\n' each, in file order; != 0 aborts */
  int (*sync) (void *ctx);                                  /* may be NULL; everything put so far has been consumed */
  long (*mark) (void *ctx);                                 /* may be NULL (then sync is used): a point after everything put so far */
  int (*wait) (void *ctx, long mark);                       /* everything put before that mark has been consumed */
} tjf_sink;

/* 1 = not gzip (first two bytes are not 1f 8b), 0 = gzip, -1 = cannot open */
int tjf_is_plain_file (const char *path);

/* Parse a plain file with n_threads readers, window_bytes of the file at a time.  Returns the number of reads handed to
 * the sink, -1 if the file cannot be opened / mapped, -2 out of memory, -3 if the sink failed. */
long tjf_parse_file (const char *path, int n_threads, size_t window_bytes, const tjf_sink *sink);

/* The same for a gzip file (first two bytes 1f 8b): a producer thread inflates one view of window_bytes ahead of the
 * parse -- BGZF members by n_threads threads side by side, any other gzip stream by one (tj_inflate.c; zlib's inflate
 * with TATAJUBA_AMD_FEEDER_INFLATE=zlib).  Same return values, and -4 if a member did not match its own CRC-32 / size
 * after it had been handed on (a damaged file, or a decoder fault: either way the counter's content is void). */
long tjf_parse_gz_file (const char *path, int n_threads, size_t window_bytes, const tjf_sink *sink);
long tjf_last_gz_stretches (void);                          /* stretches of other gzip members decoded side by side and used (tjz_round) */
long tjf_last_gz_false_starts (void);                       /* ... block starts found by trial that the decoder in front did not arrive at */
long tjf_last_bgzf_blocks (void);                           /* BGZF members the last call inflated side by side */

/* diagnostics of the last call in this process: windows that came whole from the parallel readers, windows that one reader had to finish */
void tjf_last_stats (long *windows, long *fell_back);

#endif
