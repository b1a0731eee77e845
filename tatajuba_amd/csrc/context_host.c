/* context_host.c -- host side of include/tatajuba_context.h: new_genomic_context_list () and what goes with it.
 *
 * The grouping decisions (which histogram an element joins, and how) come from the device (tjamd_context_histograms: flank
 * distances of every element against every context of the histogram being built, edit distances of the retry); this file
 * turns them into the reference's structs with the reference's bookkeeping (src/context_histogram.c:131-166,181-222,278-286).
 * The two distance functions are exported as well, as plain C: a caller may use them on its own histograms. */
#include "../../include/tatajuba_context.h"
#include "../../include/tatajuba_amd.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

tjamd_counter *tj_counter_device (hopo_counter hc);     /* hopo_host.c */
void tj_set_last_error (const char *msg);               /* hopo_device.hip */

static void
ctx_fatal (const char *msg)
{ /* biomcmc_error (): message, then exit */
  fprintf (stderr, "tatajuba_amd error: %s\n", msg);
  exit (EXIT_FAILURE);
}

/* unit-cost global edit distance (stands for biomcmc_levenshtein_distance (s1, n, s2, n, 1, 1, true): UNPINNED) */
static int
ctx_edit_distance (const char *s1, int n1, const char *s2, int n2)
{
  int x, y, result, *row = (int *) malloc ((size_t) (n1 + 1) * sizeof (int));
  if (!row) ctx_fatal ("out of memory");
  for (y = 0; y <= n1; y++) row[y] = y;
  for (x = 1; x <= n2; x++) {
    int diag = row[0];
    row[0] = x;
    for (y = 1; y <= n1; y++) {
      const int up = row[y];
      int best = diag + (s1[y - 1] != s2[x - 1]);
      if (row[y - 1] + 1 < best) best = row[y - 1] + 1;
      if (up + 1 < best) best = up + 1;
      diag = up; row[y] = best;
    }
  }
  result = row[n1];
  free (row);
  return result;
}

/* reference: src/context_histogram.c:19-23 */
int
indel_distance_between_context_histogram_and_hopo_context (context_histogram_t ch, char *name)
{
  const int len = (int) strlen (ch->name);
  return ctx_edit_distance (ch->name, len, name, len);
}

/* reference: src/context_histogram.c:25-48 */
int
distance_between_context_histogram_and_hopo_context (context_histogram_t ch, hopo_element he, int max_distance, int location_difference, int *idx_match)
{
  int distance = 0, loc_diff, this_max = 0, i;
  *idx_match = -1;
  if (ch->base != he.base) return CH_MAX_DIST;
  loc_diff = he.read_offset - ch->location;
  if (loc_diff < 0) loc_diff = -loc_diff;
  if (loc_diff > location_difference) return CH_MAX_DIST;
  for (i = 0; i < ch->n_context; i++) {
    distance = distance_between_single_context_kmer (&(ch->context[2 * i]), &(he.context[0]), 2 * max_distance);
    if (distance >= 2 * max_distance) return distance;
    distance += distance_between_single_context_kmer (&(ch->context[2 * i + 1]), &(he.context[1]), 2 * max_distance - distance);
    if (distance >= 2 * max_distance) return distance;
    if (distance > this_max) this_max = distance;
    if (distance == 0) { *idx_match = i; return 0; }
  }
  return this_max;
}

/* reference: new_context_histogram_from_hopo_elem, src/context_histogram.c:131-166 (tmp_count / tmp_length are not kept:
 * the length histogram comes from the device) */
static context_histogram_t
ctx_new (const hopo_element *he, char *name)
{
  context_histogram_t ch = (context_histogram_t) calloc (1, sizeof (struct context_histogram_struct));
  if (!ch) ctx_fatal ("out of memory");
  ch->context = (uint64_t *) malloc (2 * sizeof (uint64_t));
  if (!ch->context) ctx_fatal ("out of memory");
  ch->ref_counter = 1;
  ch->n_context = 1;
  ch->mode_context_id = 0;
  ch->base = he->base;
  ch->indel = 0;
  ch->multi = he->multi;
  ch->mode_context_count = he->count;
  ch->mode_context_length = he->length;
  ch->context[0] = he->context[0];
  ch->context[1] = he->context[1];
  ch->location = he->read_offset;
  ch->loc2d[0] = he->loc_ref_id; ch->loc2d[1] = he->loc_pos; ch->loc2d[2] = he->loc_last;
  ch->mismatches = he->mismatches;
  ch->neg_strand = he->neg_strand;
  ch->integral = he->count;
  ch->name = name;
  ch->h = NULL;
  ch->index = -1;
  ch->tmp_count = ch->tmp_length = NULL;
  ch->tract_id = -1;
  return ch;
}

/* reference: context_histogram_add_hopo_elem, src/context_histogram.c:181-222 */
static void
ctx_add (context_histogram_t ch, const hopo_element *he, char *name, int idx_match)
{
  if (idx_match < 0) {
    ch->context = (uint64_t *) realloc (ch->context, 2 * (size_t) (ch->n_context + 1) * sizeof (uint64_t));
    if (!ch->context) ctx_fatal ("out of memory");
    idx_match = ch->n_context++;
    ch->context[2 * idx_match] = he->context[0];
    ch->context[2 * idx_match + 1] = he->context[1];
  }
  if (ch->mode_context_count < he->count) {
    ch->mode_context_count = he->count;
    ch->mode_context_length = he->length;
    ch->mode_context_id = idx_match;
    ch->location = he->read_offset;
    ch->loc2d[0] = he->loc_ref_id; ch->loc2d[1] = he->loc_pos; ch->loc2d[2] = he->loc_last;
    ch->mismatches = he->mismatches;
    ch->neg_strand = he->neg_strand;
    free (ch->name);
    ch->name = name;
  }
  else free (name);
  if ((ch->multi ^ he->multi) == 1) ch->multi = 2;
  ch->integral += he->count;
}

void
del_context_histogram (context_histogram_t ch)
{ /* reference: src/context_histogram.c:168-179 */
  if (!ch) return;
  if (--ch->ref_counter) return;
  free (ch->context);
  free (ch->name);
  free (ch->tmp_count);
  free (ch->tmp_length);
  if (ch->h) { free (ch->h->i); free (ch->h); }
  free (ch);
}

void
del_genomic_context_list (genomic_context_list_t genome)
{
  int i;
  if (!genome) return;
  for (i = genome->n_hist - 1; i >= 0; i--) del_context_histogram (genome->hist[i]);
  free (genome->hist);
  free (genome->name);
  free (genome);
}

/* reference: src/context_histogram.c:224-272 + step 1 of finalise_genomic_context_hist (:278-286).  Steps 2-4 of the latter
 * (GFF3 features, location order, same-location merge) need the aligner's locations and are not done: ref_start = 0. */
genomic_context_list_t
new_genomic_context_list (hopo_counter hc)
{
  genomic_context_list_t genome;
  tjamd_counter *dev;
  tjamd_context_group *groups;
  tjamd_length_freq *lf;
  int *join_type;
  long n, ng, g, i;

  finalise_hopo_counter (hc);                           /* :231 */
  if (hc->ref_start == hc->n_elem) {                    /* :232-235 */
    fprintf (stderr, "tatajuba_amd warning: Sample %s doesn't contain any HT mapped to reference: it will be excluded from analysis\n", hc->name);
    return NULL;
  }
  n = hc->n_elem;
  dev = tj_counter_device (hc);
  if (!dev || tjamd_kept_count (dev) != n) ctx_fatal ("new_genomic_context_list: the counter's device histogram is gone (the counter was reused after finalise_hopo_counter)");
  groups = (tjamd_context_group *) malloc ((size_t) n * sizeof (tjamd_context_group));
  lf = (tjamd_length_freq *) malloc ((size_t) n * sizeof (tjamd_length_freq));
  join_type = (int *) malloc ((size_t) n * sizeof (int));
  genome = (genomic_context_list_t) malloc (sizeof (struct genomic_context_list_struct));
  if (!groups || !lf || !join_type || !genome) ctx_fatal ("out of memory");
  ng = tjamd_context_histograms (dev, hc->opt.max_distance_per_flank, hc->opt.levenshtein_distance, NULL, join_type, groups, lf, n);
  if (ng < 0) ctx_fatal (tjamd_last_error ());

  genome->hist = (context_histogram_t *) malloc ((size_t) ng * sizeof (context_histogram_t));
  if (!genome->hist) ctx_fatal ("out of memory");
  genome->n_hist = (int) ng;
  genome->opt = hc->opt;
  genome->coverage = hc->coverage;
  genome->name = hc->name;                              /* :241-242 */
  hc->name = NULL;
  genome->ref_start = 0;

  for (g = 0; g < ng; g++) {
    const long first = groups[g].first;
    context_histogram_t ch = NULL;
    int t;
    for (i = first; i < first + groups[g].n_elem; i++) {
      const hopo_element *he = &hc->elem[i];
      char *name = generate_name_from_flanking_contexts ((uint64_t *) he->context, (int8_t) he->base, genome->opt.kmer_size, he->neg_strand);
      if (i == first) { ch = ctx_new (he, name); continue; }
      {
        /* the device says how the element joined; which context of the list it met (idx_match) follows: within the flank
         * distance the loop of :36-46 stops at the first identical context, the retry never has a match */
        int idx_match = -1;
        if (join_type[i] == 1)
          for (t = 0; t < ch->n_context && idx_match < 0; t++)
            if (ch->context[2 * t] == he->context[0] && ch->context[2 * t + 1] == he->context[1]) idx_match = t;
        ctx_add (ch, he, name, idx_match);
        if (join_type[i] == 2) ch->indel = 1;           /* :260 (a bool stored in a 2-bit signed field) */
      }
    }
    /* :282 new_empfreq_from_int_weighted (lengths, n, counts): from the device */
    ch->h = (empfreq) malloc (sizeof (struct empfreq_struct));
    if (!ch->h) ctx_fatal ("out of memory");
    ch->h->n = groups[g].n_len;
    ch->h->i = (empfreq_element *) malloc ((size_t) ch->h->n * sizeof (empfreq_element));
    if (!ch->h->i) ctx_fatal ("out of memory");
    ch->h->min = ch->h->max = lf[first].length;
    for (t = 0; t < ch->h->n; t++) {
      ch->h->i[t].idx = lf[first + t].length; ch->h->i[t].freq = lf[first + t].freq;
      if (ch->h->i[t].idx < ch->h->min) ch->h->min = ch->h->i[t].idx;
      if (ch->h->i[t].idx > ch->h->max) ch->h->max = ch->h->i[t].idx;
    }
    if (ch->n_context != groups[g].n_context || ch->integral != (int) groups[g].integral)
      ctx_fatal ("new_genomic_context_list: host bookkeeping and device summary disagree");
    genome->hist[g] = ch;
  }
  free (groups); free (lf); free (join_type);
  return genome;
}
