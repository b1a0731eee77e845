/* context_oracle.c -- TEST INFRASTRUCTURE ONLY: CPU restatement of the consumers either side of the hot path.
 *
 *   N3  within-sample grouping of the finalised elements into context histograms
 *       (reference: new_genomic_context_list, src/context_histogram.c:224-272, with the distance of :25-48, the indel
 *       retry of :19-23,259-265, the bookkeeping of :131-166,181-222 and the length histogram of :274-286)
 *   N1  the order in which the per-sample lists are merged into one (reference: src/genome_set.c:250-289,
 *       ties ":278-281": the new genome's histogram goes first)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the product never does.
 *
 * PARITY UNPINNED for two pieces whose source is absent from /root/reference (biomcmc-lib, an empty submodule, version
 * not recorded anywhere in the tree):
 *   - biomcmc_levenshtein_distance (s1, n1, s2, n2, 1, 1, true): restated as the plain global edit distance with
 *     substitution cost 1 and insertion / deletion cost 1 (the two cost arguments).  The last argument cannot be checked
 *     here.  Two readings: (a) "skip the borders" = strip the common prefix and suffix of the two strings before the
 *     dynamic programme -- the usual shortcut, which leaves the global distance unchanged (then this restatement is exact);
 *     (b) free end gaps (a semi-global distance), under which a one-base shift of a name costs 1 instead of 2 and the default
 *     `levenshtein_distance = max_distance_per_flank + 1` would merge contexts that this restatement keeps apart.  The
 *     restatement follows (a), which is what the function's name and its two cost arguments suggest;
 *     tests/test_cabi.py::test_edit_distance_readings_differ_on_a_shifted_name shows a pair on which (a) and (b) differ, so
 *     that whoever holds biomcmc-lib can settle it with one call.  Product (device, host) and oracle share reading (a) by
 *     default; reading (b) -- one string may end early, the rest of the other is free -- is there on both sides as well
 *     (product: TATAJUBA_AMD_EDIT_DISTANCE=free_end; here: orc_set_edit_free_end), for the day the call says (b).
 *   - new_empfreq_from_int_weighted (lengths, n, counts): restated as "distinct values with their summed weights, highest
 *     sum first" (context_histogram.h:42: "h.idx = tract length; h.freq = count"; src/context_histogram.c:282: "histogram,
 *     from high to low count"); the order among equal sums is not stated in the reference: here the larger value first.
 * Everything else follows lines of the reference that are present and cited next to each function.
 *
 * The reference runs the grouping on elements sorted by BWA location (out of scope: the aligner is absent); here, as in
 * the product, it runs on the finalised array in its own order.  The location test of :32-34 is restated as written and
 * passes whenever read_offset is -1 everywhere (the state after src/hopo_counter.c:511). */
#include <stdlib.h>
#include <string.h>
#include "hopo_oracle.h"

#define ORC_CH_MAX_DIST 0xffff          /* src/context_histogram.h:13 */

static const char orc_bit_2_dna[] = "ACGT";     /* src/hopo_counter.c:11 */

/* reference: generate_name_from_flanking_contexts, src/hopo_counter.c:470-493.  Returns a malloc'ed "left.B.right". */
char *
orc_name_from_contexts (const uint64_t *context, int base, int kmer_size, int neg_strand)
{
  int i, j, length = 2 * kmer_size + 4;
  char *s = (char *) malloc ((size_t) length);
  uint64_t ctx = context[0];
  if (neg_strand) {                                     /* :476-482 */
    s[length - 1] = '\0';
    for (j = length - 2, i = 0; i < kmer_size; i++, j--) s[j] = orc_bit_2_dna[(~(ctx >> (2 * i))) & 3ULL];
    s[j--] = '.'; s[j--] = orc_bit_2_dna[(~base) & 3]; s[j--] = '.';
    ctx = context[1];
    for (i = 0; i < kmer_size; j--, i++) s[j] = orc_bit_2_dna[(~(ctx >> (2 * i))) & 3ULL];
  }
  else {                                                /* :483-489 */
    for (i = 0; i < kmer_size; i++) s[i] = orc_bit_2_dna[(ctx >> (2 * i)) & 3ULL];
    s[i++] = '.'; s[i++] = orc_bit_2_dna[base & 3]; s[i++] = '.';
    ctx = context[1];
    for (j = 0; j < kmer_size; j++, i++) s[i] = orc_bit_2_dna[(ctx >> (2 * j)) & 3ULL];
    s[i] = '\0';
  }
  return s;
}

/* UNPINNED (see the header): edit distance, substitution `cost_sub`, insertion / deletion `cost_indel`.  free_end = 0: reading
 * (a), the global distance.  free_end = 1: reading (b) -- one of the two strings may end early: what is left of the other
 * costs nothing (the smallest entry of the dynamic programme's last row and last column); the starts are aligned. */
int
orc_levenshtein_mode (const char *s1, int n1, const char *s2, int n2, int cost_sub, int cost_indel, int free_end)
{
  int x, y, *row = (int *) malloc ((size_t) (n1 + 1) * sizeof (int)), result, ended;
  for (y = 0; y <= n1; y++) row[y] = y * cost_indel;
  ended = row[n1];                                      /* s1 used up against the first x characters of s2 */
  for (x = 1; x <= n2; x++) {
    int diag = row[0];
    row[0] = x * cost_indel;
    for (y = 1; y <= n1; y++) {
      int up = row[y], best = diag + ((s1[y - 1] == s2[x - 1]) ? 0 : cost_sub);
      if (row[y - 1] + cost_indel < best) best = row[y - 1] + cost_indel;
      if (up + cost_indel < best) best = up + cost_indel;
      diag = up; row[y] = best;
    }
    if (row[n1] < ended) ended = row[n1];
  }
  result = row[n1];
  if (free_end) {
    if (ended < result) result = ended;
    for (y = 0; y <= n1; y++) if (row[y] < result) result = row[y];   /* s2 used up against the first y characters of s1 */
  }
  free (row);
  return result;
}

int
orc_levenshtein (const char *s1, int n1, const char *s2, int n2, int cost_sub, int cost_indel)
{
  return orc_levenshtein_mode (s1, n1, s2, n2, cost_sub, cost_indel, 0);
}

/* which reading orc_genomic_context_list's retry uses (tests set it next to the product's TATAJUBA_AMD_EDIT_DISTANCE) */
static int orc_edit_free_end = 0;
void orc_set_edit_free_end (int on) { orc_edit_free_end = on ? 1 : 0; }

/* the part of struct context_histogram_struct (src/context_histogram.h:18-43) that exists without the aligner */
typedef struct
{
  uint64_t *context;
  int base, indel, n_context, integral, location, mode_context_count, mode_context_length, mode_context_id, mode_elem;
  char *name;
  int *tmp_count, *tmp_length, index;
  int first, n_elem;
} orc_ch;

static void
orc_ch_new (orc_ch *ch, const hopo_element *he, long i, char *name)
{ /* reference: new_context_histogram_from_hopo_elem, src/context_histogram.c:131-166 */
  ch->context = (uint64_t *) malloc (2 * sizeof (uint64_t));
  ch->n_context = 1;
  ch->mode_context_id = 0;
  ch->base = he->base;
  ch->indel = 0;
  ch->mode_context_count = he->count;
  ch->mode_context_length = he->length;
  ch->context[0] = he->context[0];
  ch->context[1] = he->context[1];
  ch->location = he->read_offset;
  ch->integral = he->count;
  ch->name = name;
  ch->index = 1;
  ch->tmp_count = (int *) malloc (sizeof (int));
  ch->tmp_length = (int *) malloc (sizeof (int));
  ch->tmp_count[0] = he->count;
  ch->tmp_length[0] = he->length;
  ch->first = (int) i; ch->n_elem = 1; ch->mode_elem = (int) i;
}

static void
orc_ch_add (orc_ch *ch, const hopo_element *he, long i, char *name, int idx_match)
{ /* reference: context_histogram_add_hopo_elem, src/context_histogram.c:181-222 */
  if (idx_match < 0) {                                  /* :184-190 */
    ch->context = (uint64_t *) realloc (ch->context, 2 * (size_t) (ch->n_context + 1) * sizeof (uint64_t));
    idx_match = ch->n_context++;
    ch->context[2 * idx_match] = he->context[0];
    ch->context[2 * idx_match + 1] = he->context[1];
  }
  if (ch->mode_context_count < he->count) {             /* :192-204 */
    ch->mode_context_count = he->count;
    ch->mode_context_length = he->length;
    ch->mode_context_id = idx_match;
    ch->location = he->read_offset;
    free (ch->name);
    ch->name = name;
    ch->mode_elem = (int) i;
  }
  else free (name);
  ch->integral += he->count;                            /* :214 */
  ch->tmp_count = (int *) realloc (ch->tmp_count, (size_t) (ch->index + 1) * sizeof (int));   /* :217-220 */
  ch->tmp_length = (int *) realloc (ch->tmp_length, (size_t) (ch->index + 1) * sizeof (int));
  ch->tmp_count[ch->index] = he->count;
  ch->tmp_length[ch->index++] = he->length;
  ch->n_elem++;
}

static int
orc_ch_distance (const orc_ch *ch, const hopo_element *he, int max_distance, int location_difference, int *idx_match)
{ /* reference: distance_between_context_histogram_and_hopo_context, src/context_histogram.c:25-48 */
  int distance = 0, loc_diff, this_max = 0, i;
  *idx_match = -1;
  if (ch->base != he->base) return ORC_CH_MAX_DIST;
  loc_diff = he->read_offset - ch->location;
  if (loc_diff < 0) loc_diff = -loc_diff;
  if (loc_diff > location_difference) return ORC_CH_MAX_DIST;
  for (i = 0; i < ch->n_context; i++) {
    distance = orc_distance_single (&ch->context[2 * i], &he->context[0], 2 * max_distance);
    if (distance >= 2 * max_distance) return distance;
    distance += orc_distance_single (&ch->context[2 * i + 1], &he->context[1], 2 * max_distance - distance);
    if (distance >= 2 * max_distance) return distance;
    if (distance > this_max) this_max = distance;
    if (distance == 0) { *idx_match = i; return 0; }
  }
  return this_max;
}

typedef struct { int idx, freq; } orc_ef;
static int
orc_ef_cmp (const void *a, const void *b)
{ /* UNPINNED (see the header): highest summed weight first, then the larger value */
  const orc_ef *x = (const orc_ef *) a, *y = (const orc_ef *) b;
  if (y->freq != x->freq) return (y->freq > x->freq) ? 1 : -1;
  return (y->idx > x->idx) - (y->idx < x->idx);
}

/* One sample's list of context histograms from its finalised elements (reference: new_genomic_context_list steps 2 and
 * 3.1, src/context_histogram.c:240-286).  Outputs (caller-allocated, n entries each unless noted):
 *   group_of[i]   histogram of element i;  join_type[i]  0 = opened it, 1 = joined within the flank distance, 2 = by the indel retry
 *   g[...]        one orc_group per histogram
 *   hist_len / hist_freq   histogram g's (length, summed count) pairs at [g.first, g.first + g.n_len), highest count first
 *   contexts      2 n words: histogram g's context pairs at [2 g.first, 2 (g.first + g.n_context))
 * Returns the number of histograms. */
long
orc_genomic_context_list (const hopo_element *elem, long n, int kmer_size, int max_distance_per_flank, int levenshtein_distance,
                          int min_tract_size, int genome_coverage, int *group_of, int *join_type, orc_group *g, int *hist_len, int *hist_freq, uint64_t *contexts)
{
  long n_hist = 0, i, j;
  orc_ch *hist = (orc_ch *) malloc ((size_t) (n > 0 ? n : 1) * sizeof (orc_ch));
  for (i = 0; i < n; i++) {
    char *histname = orc_name_from_contexts (elem[i].context, elem[i].base, kmer_size, elem[i].neg_strand);   /* :242,248 */
    int idx_match = -1, distance, joined = 0;
    if (n_hist > 0) {
      orc_ch *ch = &hist[n_hist - 1];                   /* :247 j = genome->n_hist - 1 */
      distance = orc_ch_distance (ch, &elem[i], max_distance_per_flank, min_tract_size, &idx_match);   /* :249-250 */
      if (distance < max_distance_per_flank) { orc_ch_add (ch, &elem[i], i, histname, idx_match); joined = 1; }   /* :251-253 */
      else {
        if (distance < ORC_CH_MAX_DIST) {               /* :255-256 try again, now using indels */
          int len = (int) strlen (ch->name);            /* :21-22 */
          distance = orc_levenshtein_mode (ch->name, len, histname, len, 1, 1, orc_edit_free_end);
        }
        if (distance < levenshtein_distance) {          /* :258-261 */
          orc_ch_add (ch, &elem[i], i, histname, idx_match);
          ch->indel = 1;
          joined = 2;
        }
      }
    }
    if (!joined) orc_ch_new (&hist[n_hist++], &elem[i], i, histname);   /* :243,262 */
    group_of[i] = (int) (n_hist - 1);
    join_type[i] = joined;
  }
  for (i = 0; i < n_hist; i++) {                        /* :278-286 finalise_genomic_context_hist, step 1 */
    orc_ch *ch = &hist[i];
    orc_ef *ef = (orc_ef *) malloc ((size_t) ch->index * sizeof (orc_ef));
    int n_ef = 0, t;
    for (j = 0; j < ch->index; j++) {
      for (t = 0; t < n_ef && ef[t].idx != ch->tmp_length[j]; t++) {}
      if (t == n_ef) { ef[n_ef].idx = ch->tmp_length[j]; ef[n_ef++].freq = 0; }
      ef[t].freq += ch->tmp_count[j];
    }
    qsort (ef, (size_t) n_ef, sizeof (orc_ef), orc_ef_cmp);
    for (t = 0; t < n_ef; t++) { hist_len[ch->first + t] = ef[t].idx; hist_freq[ch->first + t] = ef[t].freq; }
    for (t = 0; t < 2 * ch->n_context; t++) contexts[2 * ch->first + t] = ch->context[t];
    g[i].first = ch->first; g[i].n_elem = ch->n_elem; g[i].n_context = ch->n_context; g[i].mode = ch->mode_elem;
    g[i].integral = ch->integral; g[i].indel = ch->indel; g[i].n_len = n_ef; g[i].modal_len = ef[0].idx; g[i].modal_freq = ef[0].freq;
    g[i].mode_context_id = ch->mode_context_id; g[i].mode_context_count = ch->mode_context_count; g[i].mode_context_length = ch->mode_context_length;
    g[i].coverage = genome_coverage; g[i].n_tracts = (int) n_hist;     /* :302 (genome->coverage = hc->coverage, :240) */
    free (ef); free (ch->context); free (ch->name); free (ch->tmp_count); free (ch->tmp_length);
  }
  free (hist);
  return n_hist;
}

/* ------------------------------------------------------------------------------------------------------------ */
/* N1: order of the merged list.  The reference concatenates the samples' context histograms into one array sorted by
 * location with merge sort, sample after sample (src/genome_set.c:250-289): while both lists have entries, the one
 * with the smaller location goes first and on a tie the new genome's (:278-281).  Context-keyed (the location needs
 * the absent aligner), a sample's finalised array is already in the comparator's order (src/hopo_counter.c:28-38:
 * base, context[0], context[1], length, all descending), so the union is the same two-way merge on that key with the
 * same tie rule.  That list (cat_sample / cat_index, one entry per record: which sample, which of its records) is the
 * reference's; the product keeps it collapsed: neighbours with one key become one union record with a count per sample
 * -- count field = total over the samples (20-bit store), canon flag = the strands any sample saw the key on (OR of the
 * samples' flags), as tjamd_merge_samples documents.  rec3: tjamd_record triples {ctx0, ctx1, meta}; key = (base, ctx0,
 * ctx1, signed 10-bit length).  Outputs: cat_sample, cat_index (total records each, may be NULL), keys3 (3 words per
 * distinct key), counts (n_samples ints per key: signed 20-bit count of each sample, 0 if absent).
 * Returns the number of distinct keys. */
static int
orc_key_cmp_desc (const uint64_t *a, const uint64_t *b)
{ /* reference: compare_hopo_element_decreasing, src/hopo_counter.c:28-38 */
  int ba = (int) (a[2] & 3ULL), bb = (int) (b[2] & 3ULL);
  int la = (int) ((a[2] >> 2) & 0x3FFULL), lb = (int) ((b[2] >> 2) & 0x3FFULL);
  if (ba & 2) ba -= 4;
  if (bb & 2) bb -= 4;
  if (la & 0x200) la -= 0x400;
  if (lb & 0x200) lb -= 0x400;
  if (bb != ba) return bb - ba;
  if (b[0] > a[0]) return 1;
  if (b[0] < a[0]) return -1;
  if (b[1] > a[1]) return 1;
  if (b[1] < a[1]) return -1;
  return lb - la;
}

long
orc_merge_samples (const uint64_t *rec3, const long *counts_in, int n_samples, int *cat_sample, int *cat_index, uint64_t *keys3, int *counts)
{
  const uint64_t count_mask = 0xFFFFFULL << 12, field_mask = (1ULL << 52) - 1ULL;   /* (bits 52 and up of a device record are scratch) */
  long total = 0, s, n_out = 0, i, tot = 0;
  uint64_t fl = 0;
  for (s = 0; s < n_samples; s++) total += counts_in[s];
  /* `cur` = the concatenated list so far (records tagged with their sample), `nxt` = after adding one more genome */
  uint64_t *cur = (uint64_t *) malloc ((size_t) (total > 0 ? total : 1) * 4 * sizeof (uint64_t)), *nxt = (uint64_t *) malloc ((size_t) (total > 0 ? total : 1) * 4 * sizeof (uint64_t));
  long n_cur = 0, off = 0;
  for (s = 0; s < n_samples; s++) {                     /* :255 "for each genome" */
    const uint64_t *g = rec3 + 3 * off;
    long ng = counts_in[s], a = 0, b = 0, o = 0;
    while (a < n_cur && b < ng) {                       /* :270-283 */
      if (orc_key_cmp_desc (cur + 4 * a, g + 3 * b) < 0) { memcpy (nxt + 4 * o, cur + 4 * a, 32); a++; }
      else { memcpy (nxt + 4 * o, g + 3 * b, 24); nxt[4 * o + 3] = ((uint64_t) s << 32) | (uint64_t) b; b++; }   /* (:278-281 on a tie the new genome first) */
      o++;
    }
    for (; a < n_cur; a++, o++) memcpy (nxt + 4 * o, cur + 4 * a, 32);                     /* :284-285 */
    for (; b < ng; b++, o++) { memcpy (nxt + 4 * o, g + 3 * b, 24); nxt[4 * o + 3] = ((uint64_t) s << 32) | (uint64_t) b; }
    { uint64_t *t = cur; cur = nxt; nxt = t; }
    n_cur = o; off += ng;
  }
  for (i = 0; i < n_cur; i++) {                         /* neighbours with one key -> one record, a count per sample */
    const int smp = (int) (cur[4 * i + 3] >> 32);
    int cnt = (int) ((cur[4 * i + 2] >> 12) & 0xFFFFFULL);
    if (cnt & 0x80000) cnt -= 0x100000;
    if (cat_sample) cat_sample[i] = smp;
    if (cat_index) cat_index[i] = (int) (cur[4 * i + 3] & 0xFFFFFFFFULL);
    if (i == 0 || orc_key_cmp_desc (cur + 4 * (i - 1), cur + 4 * i) != 0) {
      for (s = 0; s < n_samples; s++) counts[n_out * n_samples + s] = 0;
      n_out++; tot = 0; fl = 0;
    }
    tot += cnt;
    fl |= cur[4 * i + 2] & (7ULL << 49);
    keys3[3 * (n_out - 1)] = cur[4 * i]; keys3[3 * (n_out - 1) + 1] = cur[4 * i + 1];
    keys3[3 * (n_out - 1) + 2] = ((cur[4 * i + 2] & field_mask) & ~count_mask & ~(7ULL << 49)) | (((uint64_t) tot & 0xFFFFFULL) << 12) | fl;
    counts[(n_out - 1) * n_samples + smp] += cnt;
  }
  free (cur); free (nxt);
  return n_out;
}
