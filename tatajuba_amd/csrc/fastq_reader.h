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

#endif
