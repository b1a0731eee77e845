/* fastq_reader.h -- host feeder: FASTA/FASTQ (plain or gzip) record reader with tatajuba's parsing semantics
 * (reference behaviour: src/kseq.h:172-212 as used at src/hopo_counter.c:142-155). */
#ifndef TATAJUBA_AMD_FASTQ_READER_H
#define TATAJUBA_AMD_FASTQ_READER_H
#include <stddef.h>

typedef struct tjr_reader tjr_reader;

tjr_reader *tjr_open (const char *path);                 /* NULL if the file cannot be opened */
/* next record: returns its sequence length (>= 0) and points *seq at the bytes (valid until the next call);
 * -1 at end of file; -2 for a FASTQ record whose quality string is missing or of a different length (the caller
 * stops reading the file there, as the reference's `>= 0` loop does). */
long tjr_next (tjr_reader *r, const char **seq);
void tjr_close (tjr_reader *r);

/* The same reader over bytes already in memory (a plain, uncompressed file that was mapped): no copies, positions are
 * offsets into the block.  tjr_record_start() = offset of the '>' / '@' that opened the record tjr_next() returned last;
 * a reader started exactly there (fresh state) returns that record and everything after it identically -- what the
 * multi-threaded feeder's consistency check rests on. */
tjr_reader *tjr_open_mem (const unsigned char *data, size_t n_bytes, size_t start);
size_t tjr_record_start (const tjr_reader *r);
/* For input that arrives in pieces (feeder.c, gzip): after a tjr_next(), tjr_at_end() says that the reader touched the
 * end of its bytes -- the record it returned (or failed to return) may be cut short and has to be read again from
 * tjr_record_start() once more bytes are there; tjr_record_open() = 0 if that call found no record marker at all. */
int tjr_at_end (const tjr_reader *r);
int tjr_record_open (const tjr_reader *r);

#endif
