/* feeder.h -- multi-threaded host feeder for plain FASTA/FASTQ files; see feeder.c. */
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

/* diagnostics of the last call in this process: windows accepted from the parallel readers, 1 if one reader had to take over */
void tjf_last_stats (long *windows, long *fell_back);

#endif
