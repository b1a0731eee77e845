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

/* Unit-cost edit distance: stands for biomcmc_levenshtein_distance (s1, n, s2, n, 1, 1, true), whose source is not in the
 * reference tree (UNPINNED).  Default: the global distance.  TATAJUBA_AMD_EDIT_DISTANCE=free_end: the other reading of
 * that last argument -- one of the strings may end early and the rest of the other costs nothing (oracle/context_oracle.c). */
static int
ctx_edit_free_end (void)
{
  static int mode = -1;
  if (mode < 0) { const char *e = getenv ("TATAJUBA_AMD_EDIT_DISTANCE"); mode = (e && !strcmp (e, "free_end")) ? 1 : 0; }
  return mode;
}

static int
ctx_edit_distance (const char *s1, int n1, const char *s2, int n2)
{
  int x, y, result, early, *row = (int *) malloc ((size_t) (n1 + 1) * sizeof (int));
  if (!row) ctx_fatal ("out of memory");
  for (y = 0; y <= n1; y++) row[y] = y;
  early = row[n1];                                      /* s1 used up before s2 */
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
    if (row[n1] < early) early = row[n1];
  }
  result = row[n1];
  if (ctx_edit_free_end ()) {
    if (early < result) result = early;
    for (y = 0; y <= n1; y++) if (row[y] < result) result = row[y];   /* s2 used up before s1 */
  }
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

/* Summed mismatches of both flanks between one stored context pair and an element's, counted only as far as `budget`:
 * the left flank is measured against the whole budget, the right flank against what the left one has left. */
static int
ctx_pair_mismatches (const uint64_t *pair, const hopo_element *he, int budget)
{
  int d = distance_between_single_context_kmer ((uint64_t *) &pair[0], (uint64_t *) &he->context[0], budget);
  if (d < budget) d += distance_between_single_context_kmer ((uint64_t *) &pair[1], (uint64_t *) &he->context[1], budget - d);
  return d;
}

/* How far an element is from a histogram (behaviour of src/context_histogram.c:25-48): CH_MAX_DIST for another base or a
 * start further than `location_difference` from the histogram's; otherwise the contexts are tried in the order they were
 * added -- the first one at or beyond 2 * max_distance ends the search with its distance, an identical one ends it with 0
 * and its position in *idx_match -- and a search that runs to the end returns the largest distance it met. */
int
distance_between_context_histogram_and_hopo_context (context_histogram_t ch, hopo_element he, int max_distance, int location_difference, int *idx_match)
{
  const int budget = 2 * max_distance;
  const int apart = (he.read_offset > ch->location) ? he.read_offset - ch->location : ch->location - he.read_offset;
  int c, worst = 0;
  *idx_match = -1;
  if (ch->base != he.base || apart > location_difference) return CH_MAX_DIST;
  for (c = 0; c < ch->n_context; c++) {
    const int d = ctx_pair_mismatches (ch->context + 2 * c, &he, budget);
    if (d >= budget) return d;
    if (d == 0) { *idx_match = c; return 0; }
    if (d > worst) worst = d;
  }
  return worst;
}

static void *
ctx_alloc (size_t bytes)
{
  void *p = calloc (1, bytes ? bytes : 1);
  if (!p) ctx_fatal ("out of memory");
  return p;
}

/* What a histogram takes from the element that represents it -- the first one, then every element with a count above all
 * before it (src/context_histogram.c:141-155,192-203): where it is, how it mapped, its count and tract length. */
static void
ctx_take_representative (context_histogram_t ch, const hopo_element *he, int context_id)
{
  ch->mode_context_id = context_id;
  ch->mode_context_count = he->count;
  ch->mode_context_length = he->length;
  ch->location = he->read_offset;
  ch->loc2d[0] = he->loc_ref_id;
  ch->loc2d[1] = he->loc_pos;
  ch->loc2d[2] = he->loc_last;
  ch->mismatches = he->mismatches;
  ch->neg_strand = he->neg_strand;
}

/* One histogram from the elements [first, first + n_elem) that the device put together (join_type: how each one joined).
 * Same outcome as the reference's element-by-element bookkeeping (src/context_histogram.c:131-166,181-222), built in one
 * go: the device has already said how many distinct contexts there are and which element is the modal one, so the
 * context list is allocated once and only the modal element's name is ever made. */
static context_histogram_t
ctx_build (const hopo_element *elem, const int *join_type, const tjamd_context_group *grp, const tjamd_length_freq *lf, int kmer_size)
{
  context_histogram_t ch = (context_histogram_t) ctx_alloc (sizeof (struct context_histogram_struct));
  const hopo_element *he = elem + grp->first;
  int e, t, n_ctx = 0, modal_context = 0, best = he->count;

  ch->context = (uint64_t *) ctx_alloc (2 * (size_t) grp->n_context * sizeof (uint64_t));
  ch->ref_counter = 1;
  ch->base = he->base;
  ch->multi = he->multi;
  ch->index = -1;
  ch->tract_id = -1;
  for (e = 0; e < grp->n_elem; e++, he++) {
    int at = -1;
    /* an element within the flank distance that met its own context in the list is counted there (the search of :36-46
     * stops at the first identical context); the opener, and an element taken in by the retry, always add theirs */
    if (e > 0 && join_type[grp->first + e] == 1)
      for (t = 0; t < n_ctx && at < 0; t++)
        if (ch->context[2 * t] == he->context[0] && ch->context[2 * t + 1] == he->context[1]) at = t;
    if (at < 0) {
      if (n_ctx >= grp->n_context) ctx_fatal ("new_genomic_context_list: host bookkeeping and device summary disagree (contexts)");
      at = n_ctx++;
      ch->context[2 * at] = he->context[0];
      ch->context[2 * at + 1] = he->context[1];
    }
    if (e == 0 || he->count > best) { best = he->count; modal_context = at; ctx_take_representative (ch, he, at); }
    if (e > 0) {
      if (join_type[grp->first + e] == 2) ch->indel = 1;  /* :260 (a bool in a 2-bit signed field) */
      if ((ch->multi ^ he->multi) == 1) ch->multi = 2;    /* :219 */
    }
    ch->integral += he->count;
  }
  if (n_ctx != grp->n_context || ch->integral != (int) grp->integral || ch->mode_context_id != modal_context)
    ctx_fatal ("new_genomic_context_list: host bookkeeping and device summary disagree");
  ch->n_context = n_ctx;
  he = elem + grp->mode;                                  /* the device's modal element: first one with the highest count */
  if (he->count != ch->mode_context_count || he->length != ch->mode_context_length)
    ctx_fatal ("new_genomic_context_list: host and device disagree on the modal element");
  ch->name = generate_name_from_flanking_contexts ((uint64_t *) he->context, (int8_t) he->base, kmer_size, he->neg_strand);
  /* tract lengths weighted by count, highest count first (:282 new_empfreq_from_int_weighted): from the device */
  ch->h = (empfreq) ctx_alloc (sizeof (struct empfreq_struct));
  ch->h->n = grp->n_len;
  ch->h->i = (empfreq_element *) ctx_alloc ((size_t) grp->n_len * sizeof (empfreq_element));
  ch->h->min = ch->h->max = lf[0].length;
  for (t = 0; t < grp->n_len; t++) {
    ch->h->i[t].idx = lf[t].length;
    ch->h->i[t].freq = lf[t].freq;
    if (lf[t].length < ch->h->min) ch->h->min = lf[t].length;
    if (lf[t].length > ch->h->max) ch->h->max = lf[t].length;
  }
  return ch;
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

/* reference: src/context_histogram.c:224-272 + steps 1 and 5 of finalise_genomic_context_hist (:278-286, :302).  Steps
 * 2-4 of the latter (GFF3 features, location order, same-location merge) need the aligner's locations and are not done:
 * ref_start = 0. */
genomic_context_list_t
new_genomic_context_list (hopo_counter hc)
{
  genomic_context_list_t genome;
  tjamd_counter *dev;
  tjamd_context_group *groups;
  tjamd_length_freq *lf;
  int *join_type;
  long n, ng, g;

  finalise_hopo_counter (hc);                           /* :231 */
  if (hc->ref_start != 0 && hc->ref_start != hc->n_elem) {
    /* Only a host program's own find_reference_location_and_sort_hopo_counter (the weak hook finalise_hopo_counter calls
     * when it is linked) sets ref_start: it has re-sorted hc->elem by location, and the grouping decisions held on the
     * device are in the finalised order, not that one.  Applying them would give wrong histograms: stop. */
    ctx_fatal ("new_genomic_context_list: hc->elem was re-ordered after finalise_hopo_counter (ref_start != 0: an aligner hook is linked); "
               "location-ordered grouping is outside this library -- see INTEGRATION.md, \"with an aligner\"");
  }
  if (hc->ref_start == hc->n_elem) {                    /* :232-235: with no aligner linked this is "nothing left after the filters" */
    fprintf (stderr, "tatajuba_amd warning: Sample %s holds no homopolymeric tract after the strand / depth filters: it will be excluded from analysis\n",
             hc->name ? hc->name : "(unnamed)");
    return NULL;
  }
  n = hc->n_elem;
  dev = tj_counter_device (hc);
  if (!dev || tjamd_kept_count (dev) != n) ctx_fatal ("new_genomic_context_list: the counter's device histogram is gone (the counter was reused after finalise_hopo_counter)");
  groups = (tjamd_context_group *) ctx_alloc ((size_t) n * sizeof (tjamd_context_group));
  lf = (tjamd_length_freq *) ctx_alloc ((size_t) n * sizeof (tjamd_length_freq));
  join_type = (int *) ctx_alloc ((size_t) n * sizeof (int));
  ng = tjamd_context_histograms (dev, hc->opt.max_distance_per_flank, hc->opt.levenshtein_distance, NULL, join_type, groups, lf, n);
  if (ng < 0) ctx_fatal (tjamd_last_error ());

  genome = (genomic_context_list_t) ctx_alloc (sizeof (struct genomic_context_list_struct));
  genome->hist = (context_histogram_t *) ctx_alloc ((size_t) ng * sizeof (context_histogram_t));
  genome->n_hist = (int) ng;
  genome->opt = hc->opt;
  genome->coverage = hc->coverage;
  genome->name = hc->name;                              /* :241-242: the list takes the counter's name */
  hc->name = NULL;
  genome->ref_start = 0;
  for (g = 0; g < ng; g++) {
    context_histogram_t ch = ctx_build (hc->elem, join_type, &groups[g], lf + groups[g].first, genome->opt.kmer_size);
    ch->coverage = genome->coverage;                    /* :302: genome-wide figures on every histogram */
    ch->n_tracts = genome->n_hist;
    genome->hist[g] = ch;
  }
  free (groups); free (lf); free (join_type);
  return genome;
}
