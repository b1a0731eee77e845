/* hopo_oracle.h -- TEST INFRASTRUCTURE ONLY (see hopo_oracle.c header). */
#ifndef TATAJUBA_AMD_ORACLE_H
#define TATAJUBA_AMD_ORACLE_H

#include "../include/tatajuba_hopo.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct
{
  hopo_element *elem;
  int n_elem, n_alloc, kmer_size, coverage, ref_start;
  int *idx_initial, *idx_final, n_idx;
  long n_reads;           /* reads seen by orc_scan_file / orc_scan_stream (bookkeeping only) */
  long n_undefined;       /* qualifying non-ACGTU runs with no earlier tract in the read: reference reads
                             uninitialised memory there (src/hopo_counter.c:223,248); the oracle emits nothing */
  int status;             /* 0 ok; 1 = "no HTs before QC"; 2 = "none after filter"; 3 = "none reach min coverage" */
} orc_counter;

orc_counter *orc_new (int kmer_size);
void orc_free (orc_counter *oc);
void orc_scan_seq (orc_counter *oc, const char *seq, int seq_length, int min_tract_size);
void orc_scan_seq_all_monomers (orc_counter *oc, const char *seq, int seq_length);
long orc_scan_file (orc_counter *oc, const char *path, int min_tract_size);        /* reads parsed, -1 = cannot open */
long orc_scan_stream (orc_counter *oc, const char *buf, size_t n, int min_tract_size); /* '\n'-delimited reads */
void orc_finalise (orc_counter *oc, int remove_biased, int min_coverage);
int  orc_compare_decreasing (const void *a, const void *b);

/* parser only: concatenates every parsed read followed by '\n' into a malloc'ed buffer (caller frees) */
char *orc_parse_file_to_stream (const char *path, size_t *n_bytes, long *n_reads);

#ifdef __cplusplus
}
#endif

/* "next" rows N3 / N1 (see hopo_oracle.c): packed-context distances (reference: src/hopo_counter.c:61-113), greedy grouping
 * of the finalised elements (src/context_histogram.c:25-48,245-270) and tract ids (src/genome_set.c:207-221), context-keyed */
int orc_distance_single (const uint64_t *c1, const uint64_t *c2, int max_dist);
int orc_distance_pair (const uint64_t *c1, const uint64_t *c2);
int orc_distance_pair_shift (const uint64_t *c1, const uint64_t *c2, int *best_shift);
long orc_group_contexts (const hopo_element *elem, long n, int max_distance_per_flank, int *group_of,
                         int *g_first, int *g_n_elem, int *g_n_ctx, long *g_integral, int *g_mode);
long orc_tract_ids (const uint64_t *rec3, long n, int *tract_id);

/* context_oracle.c: the whole of new_genomic_context_list's grouping (Hamming test, indel retry, bookkeeping, length
 * histograms; src/context_histogram.c:19-48,131-222,224-286) and the order of the cross-sample merge
 * (src/genome_set.c:250-289) */
typedef struct
{
  int first, n_elem, n_context, mode, indel, n_len, modal_len, modal_freq, mode_context_id, mode_context_count, mode_context_length;
  int coverage, n_tracts;     /* genome-wide figures copied onto every histogram (src/context_histogram.c:302) */
  long integral;
} orc_group;
char *orc_name_from_contexts (const uint64_t *context, int base, int kmer_size, int neg_strand);
int orc_levenshtein (const char *s1, int n1, const char *s2, int n2, int cost_sub, int cost_indel);
int orc_levenshtein_mode (const char *s1, int n1, const char *s2, int n2, int cost_sub, int cost_indel, int free_end);
void orc_set_edit_free_end (int on);
long orc_genomic_context_list (const hopo_element *elem, long n, int kmer_size, int max_distance_per_flank, int levenshtein_distance,
                               int min_tract_size, int genome_coverage, int *group_of, int *join_type, orc_group *g, int *hist_len, int *hist_freq, uint64_t *contexts);
long orc_merge_samples (const uint64_t *rec3, const long *counts_in, int n_samples, int *cat_sample, int *cat_index, uint64_t *keys3, int *counts);

#endif
