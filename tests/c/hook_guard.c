/* Test program (tests/test_gpu_parity.py::test_context_list_refuses_elements_reordered_by_an_aligner_hook): a host program
 * that links its own find_reference_location_and_sort_hopo_counter () -- the weak hook finalise_hopo_counter () calls last,
 * where the reference calls its BWA step (src/hopo_counter.c:416,495-572) -- which re-orders hc->elem and sets ref_start
 * as an aligner would.  new_genomic_context_list () must then stop with a message instead of applying the device's
 * grouping (made in the finalised order) to a differently ordered array.
 * usage: hook_guard reads.fastq k m   (with a 4th argument the hook leaves the array alone: the list is built) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "tatajuba_context.h"

static int hook_calls = 0, hook_permutes = 1;

void
find_reference_location_and_sort_hopo_counter (hopo_counter hc)
{
  int i;
  hook_calls++;
  if (!hook_permutes || hc->n_elem < 2) return;
  for (i = 0; i < hc->n_elem / 2; i++) {                /* "sorted by location": here simply reversed */
    hopo_element t = hc->elem[i];
    hc->elem[i] = hc->elem[hc->n_elem - 1 - i];
    hc->elem[hc->n_elem - 1 - i] = t;
  }
  hc->ref_start = 1;                                    /* one element "not found in the reference" (src/hopo_counter.c:564) */
}

int
main (int argc, char **argv)
{
  tatajuba_options_t opt;
  hopo_counter hc;
  genomic_context_list_t g;
  if (argc < 4) return 2;
  if (argc > 4) hook_permutes = 0;
  memset (&opt, 0, sizeof (opt));
  opt.kmer_size = atoi (argv[2]); opt.min_tract_size = atoi (argv[3]);
  opt.min_coverage = 2; opt.remove_biased = false; opt.max_distance_per_flank = 1; opt.levenshtein_distance = 2;
  opt.n_samples = 1; opt.n_threads = 1;
  hc = new_or_append_hopo_counter_from_file (NULL, argv[1], opt);
  g = new_genomic_context_list (hc);
  printf ("hook calls %d, histograms %d\n", hook_calls, g ? g->n_hist : -1);
  del_genomic_context_list (g);
  del_hopo_counter (hc);
  return 0;
}
