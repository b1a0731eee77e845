/* tatajuba_context.h -- the consumer on the output side of the accelerated path: a sample's finalised tracts grouped
 * into context histograms.
 *
 * Declares, with the reference's names and argument meaning, the part of tatajuba's src/context_histogram.h that exists
 * without the BWA aligner and the GFF3 reader (reference: src/context_histogram.h:15-49, src/context_histogram.c:19-48,
 * 131-222,224-286).  new_genomic_context_list () finalises the counter and groups its elements exactly as the reference's
 * loop does -- flank distance first, then the retry with the edit distance between the names -- with the distances
 * computed on the device (tjamd_context_histograms, include/tatajuba_amd.h) and the structs assembled on the host; every
 * histogram gets its tract-length histogram `h`.  What needs locations stays out: the location sort, the same-location
 * merge (src/context_histogram.c:288-421) and the GFF3 feature of a histogram (`gffeature`, a biomcmc-lib struct, is the
 * one field of struct context_histogram_struct that is not mirrored: the struct here is source-compatible for every
 * other field, not layout-compatible).
 *
 * Two pieces stand for biomcmc-lib code that is absent from the reference tree (UNPINNED, see oracle/context_oracle.c):
 * the edit distance behind indel_distance_between_context_histogram_and_hopo_context (the global unit-cost distance; with
 * TATAJUBA_AMD_EDIT_DISTANCE=free_end the other reading of biomcmc_levenshtein_distance's last argument: one string may
 * end early) and the order of equal counts inside an empfreq. */
#ifndef TATAJUBA_AMD_CONTEXT_H
#define TATAJUBA_AMD_CONTEXT_H

#include "tatajuba_hopo.h"

#ifdef __cplusplus
extern "C" {
#endif

#define CH_MAX_DIST 0xffff              /* reference: src/context_histogram.h:13 */

/* biomcmc-lib's empirical frequency as far as tatajuba reads it (src/context_histogram.c:56-66: h->n, h->i[j].idx,
 * h->i[j].freq): distinct values with their summed weights, highest weight first */
#ifndef TATAJUBA_AMD_HAVE_EMPFREQ
typedef struct { int freq, idx; } empfreq_element;
struct empfreq_struct { empfreq_element *i; int n, min, max; };    /* min / max: smallest and largest idx (UNPINNED like the rest of empfreq: in
                                                                    * biomcmc-lib they may be positions in `i` instead; tatajuba reads them in a debug print only) */
typedef struct empfreq_struct *empfreq;
#endif

typedef struct context_histogram_struct *context_histogram_t;
typedef struct genomic_context_list_struct *genomic_context_list_t;

/* reference: src/context_histogram.h:18-43 (gffeature left out, see above) */
struct context_histogram_struct
{
  uint64_t *context;      /* 2 * n_context words: every context pair within distance, in the order they were added */
  int32_t base:2, multi:3, indel:2, neg_strand:1, mismatches:12;
  char *name;             /* "left.B.right" of the modal context */
  int n_context, integral, location, loc2d[3], coverage, n_tracts, mode_context_count, mode_context_length, mode_context_id;
  int *tmp_count, *tmp_length, index;   /* NULL / NULL / -1 once h exists (src/context_histogram.c:283-285) */
  empfreq h;              /* h.idx = tract length; h.freq = count */
  int tract_id;
  int ref_counter;
};

/* reference: src/context_histogram.h:45-51 */
struct genomic_context_list_struct
{
  context_histogram_t *hist;
  char *name;
  tatajuba_options_t opt;
  int n_hist, coverage, ref_start;
};

int indel_distance_between_context_histogram_and_hopo_context (context_histogram_t ch, char *name);
int distance_between_context_histogram_and_hopo_context (context_histogram_t ch, hopo_element he, int max_distance, int location_difference, int *idx_match);
void del_context_histogram (context_histogram_t ch);
genomic_context_list_t new_genomic_context_list (hopo_counter hc);
void del_genomic_context_list (genomic_context_list_t genome);

#ifdef __cplusplus
}
#endif
#endif
