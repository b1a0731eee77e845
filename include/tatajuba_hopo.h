/* tatajuba_hopo.h -- drop-in C boundary of the MI355X homopolymer-tract counting engine.
 *
 * This header declares, with C linkage and the SAME names, struct layouts and argument meaning, the part of
 * tatajuba's `src/hopo_counter.h` that sits on the accelerated path (reference: src/hopo_counter.h:15-80).  A
 * tatajuba tree that includes this header instead of its own hopo_counter.h and links libtatajuba_amd.so gets the
 * per-read scan and the per-sample sort/dedupe/filter executed on an MI355X.  Nothing here is a torch type; every
 * argument is a plain pointer, integer or by-value C struct.
 *
 * Layouts are ABI: callers index hc->elem[i] and read hc->n_elem, ref_start, name, opt, coverage directly
 * (reference: src/context_histogram.c:231-256).
 */
#ifndef TATAJUBA_AMD_HOPO_H
#define TATAJUBA_AMD_HOPO_H

#include <stdint.h>
#include <stdbool.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* biomcmc-lib's GFF3 handle; opaque here (only ever stored and handed back; reference: src/hopo_counter.h:24) */
#ifndef TATAJUBA_AMD_HAVE_GFF3_T
typedef struct gff3_struct *gff3_t;
#endif

/* ASCII -> 2-bit tables, reference: src/hopo_counter.c:10-11,205-216.
 * dna_in_2_bits[c][0] forward code (A0 C1 G2 T/U3), [c][1] complement code, 4 for any other byte. */
extern uint8_t dna_in_2_bits[256][2];
extern char bit_2_dna[];

typedef struct hopo_counter_struct *hopo_counter;

/* reference: src/hopo_counter.h:20-32 (passed and stored BY VALUE; 64 bytes on LP64) */
typedef struct
{
  char *reference_fasta_filename;   /* not used on this path (BWA index of the reference genome) */
  char *outdir;                     /* not used on this path */
  bool paired_end;                  /* R2 is appended to R1's counter by the caller (src/genome_set.c:72-73) */
  bool remove_biased;               /* keep a (context, length) only if it was seen on both strands (canon_flag == 3) */
  bool save_vcf;                    /* not used on this path */
  gff3_t gff;                       /* stored and handed back, never dereferenced here */
  int max_distance_per_flank;       /* not used on this path (context grouping) */
  int kmer_size;                    /* flanking context length k, 2..32 */
  int min_tract_size;               /* shortest tract recorded (1 behaves as 2) */
  int levenshtein_distance;         /* not used on this path */
  int min_coverage;                 /* a context is indexed only if its depth reaches this */
  int n_samples;
  int n_threads;
} tatajuba_options_t;

/* reference: src/hopo_counter.h:34-49 (40 bytes).  The bitfield word is laid out LSB-first by gcc/clang on x86-64:
 * base 0-1, length 2-11, count 12-31, mismatches 32-43, multi 44-46, neg_strand 47-48, canon_flag 49-51.
 * The device kernels build exactly this 64-bit word (see TJ_META_* in tatajuba_amd.h). */
typedef struct
{
  uint64_t context[2];    /* left / right flanking k-mers, 2 bits per base, first base in the two lowest bits */
  int64_t base:2;         /* 0 = A/T tract, 1 = C/G tract (canonical strand) */
  int64_t length:10;      /* tract length in bases (signed 10-bit store: 512 wraps to -512) */
  int64_t count:20;       /* depth of this (context, base, length) */
  int64_t mismatches:12;
  int64_t multi:3;
  int64_t neg_strand:2;
  int64_t canon_flag:3;   /* 1 = seen as A/C run, 2 = seen as T/G run (reverse-complemented), 3 = both */
  int32_t read_offset;    /* start of left flank within the read; -1 once finalised (reference: src/hopo_counter.c:511) */
  int32_t loc_ref_id;
  int32_t loc_pos;
  int32_t loc_last;
} hopo_element;

/* reference: src/hopo_counter.h:51-59 (136 bytes on LP64) */
struct hopo_counter_struct
{
  hopo_element *elem;       /* raw records (only those added through update_hopo_counter_from_seq) until finalised, then the histogram */
  char *name;               /* copy of the first file's name */
  int ref_start;
  int n_elem;               /* raw tracts so far; after finalise: entries of the histogram */
  int n_alloc;
  int kmer_size;
  int coverage;             /* estimate_coverage_hopo_counter's value (src/hopo_counter.c:419-438) */
  int *idx_initial;         /* per indexed context: first entry in elem ... */
  int *idx_final;           /* ... and one past its last */
  int n_idx;
  tatajuba_options_t opt;   /* by value, as given to the first new_or_append call */
  int ref_counter;
};

/* ---- accelerated path ------------------------------------------------------------------------------------- */

/* reference: src/hopo_counter.h:69, src/hopo_counter.c:159-173 */
hopo_counter new_hopo_counter (int kmer_size);

/* reference: src/hopo_counter.h:74, src/hopo_counter.c:175-186 (ref-counted; frees elem, name, idx_*, device state) */
void del_hopo_counter (hopo_counter hc);

/* reference: src/hopo_counter.h:70, src/hopo_counter.c:135-157; called at src/genome_set.c:72,73,87.
 * hc == NULL creates the counter (name = copy of filename, opt stored by value); otherwise the file's reads are
 * appended.  Reads are parsed on the host (FASTA/FASTQ, plain or gzip) into sentinel-delimited batches, copied to
 * HBM and scanned by the HIP scan kernel; the raw tract records stay device-resident until finalise_hopo_counter.
 * hc->n_elem is the number of raw records, as in the reference.  Aborts (exit) if the counter was already finalised
 * (reference :152) and -- documented divergence -- if the file cannot be opened (reference leaves gzopen unchecked). */
hopo_counter new_or_append_hopo_counter_from_file (hopo_counter hc, const char *filename, tatajuba_options_t opt);

/* reference: src/hopo_counter.h:71, src/hopo_counter.c:219-258; called at src/genome_set.c:539 with m = 2.
 * Synchronous: the string goes through the same HIP scan kernel and its records are appended, in read order, to the
 * host array hc->elem (so hc->elem[0..n_elem) is readable right after the call, as callers expect). */
void update_hopo_counter_from_seq (hopo_counter hc, char *seq, int seq_length, int min_tract_size);

/* reference: src/hopo_counter.h:73, src/hopo_counter.c:260-283; called at src/genome_set.c:543.  Same as above for every
 * base that differs from both of its neighbours (tract length 1), used when a reference window holds no tract. */
void update_hopo_counter_from_seq_all_monomers (hopo_counter hc, char *seq, int seq_length);

/* reference: src/hopo_counter.h:78, src/hopo_counter.c:339-417.  Device radix sort + segmented reduce of the raw
 * records, strand-bias/singleton filter, per-context depth index and coverage estimate; leaves elem, n_elem, n_alloc,
 * idx_initial, idx_final, n_idx, coverage, ref_start exactly as the reference has them when it reaches its BWA step
 * (src/hopo_counter.c:416; read_offset = -1 as after :511).  If the program defines
 * find_reference_location_and_sort_hopo_counter() (weak reference below) it is called last, as in the reference. */
void finalise_hopo_counter (hopo_counter hc);

/* Out of scope here (needs the BWA fork); resolved at link time if the host program provides it. */
void find_reference_location_and_sort_hopo_counter (hopo_counter hc) __attribute__((weak));

/* ---- small host helpers kept so the header is complete for callers (not on the accelerated path) ------------ */

int compare_hopo_element_decreasing (const void *a, const void *b);   /* reference: src/hopo_counter.c:28-38 */
int compare_hopo_context (hopo_element a, hopo_element b);            /* reference: src/hopo_counter.c:48-58 */
/* reference: src/hopo_counter.c:471-493; caller frees */
char *generate_name_from_flanking_contexts (uint64_t *context, int8_t base, int kmer_size, bool neg_strand);
/* reference: src/hopo_counter.c:447-469; caller frees */
char *generate_tract_as_string (uint64_t *context, int8_t base, int kmer_size, int tract_length, bool neg_strand);
void print_tatajuba_options (tatajuba_options_t opt);                /* reference: src/hopo_counter.c:115-133 */
/* distances between packed contexts: differing bases of one flank, counted up to max_dist (reference: src/hopo_counter.h:63,
 * src/hopo_counter.c:61-68); of both flanks (:64, :70-79); of both flanks allowing one of the two to be shifted by up to
 * three bases at a cost of one per base, best_shift[4] = bases shifted {c1 left, c2 left, c1 right, c2 right} (:66, :81-113) */
int distance_between_single_context_kmer (uint64_t *c1, uint64_t *c2, int max_dist);
int distance_between_context_kmer_pair (uint64_t *c1, uint64_t *c2);
int distance_between_context_kmer_pair_with_edit_shift (uint64_t *c1, uint64_t *c2, int *best_shift);
/* reference: src/hopo_counter.h:75, src/hopo_counter.c:188-203; caller frees; NULL (and *tract_length = 0) if the string holds no tract */
char *leftmost_hopo_name_and_length_from_string (char *seq, size_t len, int kmer_size, int min_tract_size, int *tract_length);
/* reference: src/hopo_counter.h:77, src/hopo_counter.c:440-445 (obsolete there) */
int hopo_counter_histogram_integral (hopo_counter hc, int start);

#ifdef __cplusplus
}
#endif
#endif
