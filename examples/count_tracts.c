/* count_tracts.c -- a plain C caller of the drop-in API (include/tatajuba_hopo.h), the way tatajuba's own
 * src/genome_set.c:66-94 uses it: one counter per sample, R1 (and R2) appended, then finalised.
 *
 *   gcc -O2 -I include examples/count_tracts.c -L tatajuba_amd -ltatajuba_amd -Wl,-rpath,$PWD/tatajuba_amd -o count_tracts
 *   ./count_tracts reads_R1.fastq.gz [reads_R2.fastq.gz] [-k 25] [-m 4] [-c 5] [-b 0|1]
 *
 * Prints the raw tract count, the size of the context histogram after the strand-bias / coverage filters, and the
 * deepest contexts by name (left flank . base . right flank, reference: src/hopo_counter.c:471-493). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <tatajuba_hopo.h>

int
main (int argc, char **argv)
{
  tatajuba_options_t opt;
  const char *files[2] = {NULL, NULL};
  int n_files = 0, i, shown = 0;
  hopo_counter hc;

  memset (&opt, 0, sizeof opt);
  opt.kmer_size = 25; opt.min_tract_size = 4; opt.min_coverage = 5; opt.remove_biased = true;   /* tatajuba's defaults */
  opt.max_distance_per_flank = 1; opt.levenshtein_distance = 1; opt.n_threads = 1;
  for (i = 1; i < argc; i++) {
    if (!strcmp (argv[i], "-k") && i + 1 < argc) opt.kmer_size = atoi (argv[++i]);
    else if (!strcmp (argv[i], "-m") && i + 1 < argc) opt.min_tract_size = atoi (argv[++i]);
    else if (!strcmp (argv[i], "-c") && i + 1 < argc) opt.min_coverage = atoi (argv[++i]);
    else if (!strcmp (argv[i], "-b") && i + 1 < argc) opt.remove_biased = atoi (argv[++i]) != 0;
    else if (n_files < 2) files[n_files++] = argv[i];
  }
  if (!n_files) { fprintf (stderr, "usage: %s R1.fastq[.gz] [R2.fastq[.gz]] [-k K] [-m M] [-c C] [-b 0|1]\n", argv[0]); return 2; }
  opt.paired_end = n_files == 2;
  opt.n_samples = 1;

  hc = new_or_append_hopo_counter_from_file (NULL, files[0], opt);
  if (n_files == 2) new_or_append_hopo_counter_from_file (hc, files[1], opt);        /* reference: src/genome_set.c:72-73 */
  printf ("sample %s: %d homopolymer tracts with both flanks (k=%d, min tract %d)\n", hc->name, hc->n_elem, hc->kmer_size, opt.min_tract_size);
  finalise_hopo_counter (hc);
  printf ("context histogram: %d (context, length) entries, %d contexts reach coverage %d, coverage estimate %d\n",
          hc->n_elem, hc->n_idx, opt.min_coverage, hc->coverage);
  for (i = 0; i < hc->n_idx && shown < 5; i++, shown++) {                             /* each indexed context: its lengths and depths */
    int j;
    char *name = generate_name_from_flanking_contexts (hc->elem[hc->idx_initial[i]].context, (int8_t) hc->elem[hc->idx_initial[i]].base, hc->kmer_size, false);
    printf ("  %s :", name);
    for (j = hc->idx_initial[i]; j < hc->idx_final[i]; j++) printf (" len %d x%d", (int) hc->elem[j].length, (int) hc->elem[j].count);
    printf ("\n");
    free (name);
  }
  del_hopo_counter (hc);
  return 0;
}
