/* merge_samples.c -- several samples in ONE process, the way tatajuba runs them (src/genome_set.c:66-94: a thread per
 * sample, then the cross-sample merge of :195-229), through the C ABI and nothing else: every sample gets a counter on
 * device (sample mod devices), is scanned and finalised there; tjamd_gather_histograms brings the per-sample
 * histograms to the first counter's device (peer copies between devices), tjamd_merge_samples builds the union with a
 * count per sample, tjamd_tract_ids numbers its contexts.
 *
 *   gcc -O2 -I include examples/merge_samples.c -L tatajuba_amd -ltatajuba_amd -Wl,-rpath,$PWD/tatajuba_amd -o merge_samples
 *   ./merge_samples [-k 10] [-m 3] [-c 5] sample1.fastq[.gz] sample2.fastq[.gz] ...                                         */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <tatajuba_amd.h>

#define MAX_SAMPLES 64

int
main (int argc, char **argv)
{
  tjamd_counter *ctr[MAX_SAMPLES];
  const char *files[MAX_SAMPLES];
  long counts[MAX_SAMPLES], total, n_union, n_tracts, shared = 0, i;
  int n = 0, k = 10, m = 3, cov = 5, a, ndev = tjamd_device_count (), status;
  const void *d_records = NULL;
  void *d_keys, *d_counts;
  int *h_counts, *h_ids;

  for (a = 1; a < argc; a++) {
    if (!strcmp (argv[a], "-k") && a + 1 < argc) k = atoi (argv[++a]);
    else if (!strcmp (argv[a], "-m") && a + 1 < argc) m = atoi (argv[++a]);
    else if (!strcmp (argv[a], "-c") && a + 1 < argc) cov = atoi (argv[++a]);
    else if (n < MAX_SAMPLES) files[n++] = argv[a];
  }
  if (n < 1) { fprintf (stderr, "usage: %s [-k K] [-m M] [-c C] sample.fastq[.gz] ...\n", argv[0]); return 2; }
  if (ndev < 1) { fprintf (stderr, "tatajuba_amd error: no HIP device is visible (there is no CPU fallback)\n"); return 1; }

  for (a = 0; a < n; a++) {                               /* reference: one OpenMP thread per sample; here one after the other */
    long n_reads = 0, bytes = tjamd_read_file_stream (files[a], NULL, 0, &n_reads);     /* size, then contents */
    unsigned char *buf;
    if (bytes < 0) { fprintf (stderr, "cannot read %s\n", files[a]); return 1; }
    buf = (unsigned char *) malloc ((size_t) bytes + 1);
    tjamd_read_file_stream (files[a], buf, bytes, &n_reads);
    ctr[a] = tjamd_counter_create (a % ndev, k);
    if (!ctr[a] || tjamd_scan_host (ctr[a], buf, (size_t) bytes, m) || tjamd_finalise (ctr[a], 1, cov, &status)) { fprintf (stderr, "%s\n", tjamd_last_error ()); return 1; }
    printf ("sample %d (%s) on device %d: %ld reads, %ld histogram bars, %d contexts indexed\n", a, files[a], tjamd_counter_device (ctr[a]), n_reads,
            tjamd_kept_count (ctr[a]), tjamd_n_idx (ctr[a]));
    free (buf);
  }

  total = tjamd_gather_histograms (ctr[0], ctr, n, &d_records, counts);
  if (total < 0) { fprintf (stderr, "%s\n", tjamd_last_error ()); return 1; }
  d_keys = tjamd_device_alloc (ctr[0], (size_t) (total ? total : 1) * 24);
  d_counts = tjamd_device_alloc (ctr[0], (size_t) (total ? total : 1) * (size_t) n * 4);
  n_union = tjamd_merge_samples (ctr[0], d_records, counts, n, d_keys, d_counts, total);
  if (n_union < 0) { fprintf (stderr, "%s\n", tjamd_last_error ()); return 1; }
  h_counts = (int *) malloc ((size_t) (n_union ? n_union : 1) * (size_t) n * sizeof (int));
  h_ids = (int *) malloc ((size_t) (n_union ? n_union : 1) * sizeof (int));
  n_tracts = tjamd_tract_ids (ctr[0], d_keys, n_union, NULL, h_ids);
  if (n_tracts < 0 || tjamd_device_download (ctr[0], h_counts, d_counts, (size_t) n_union * (size_t) n * 4)) { fprintf (stderr, "%s\n", tjamd_last_error ()); return 1; }
  for (i = 0; i < n_union; i++) {
    int present = 0;
    for (a = 0; a < n; a++) present += h_counts[i * n + a] > 0;
    shared += present == n;
  }
  printf ("merged: %ld bars gathered, %ld in the union, %ld seen in every sample, %ld tract ids\n", total, n_union, shared, n_tracts);
  tjamd_device_free (ctr[0], d_keys); tjamd_device_free (ctr[0], d_counts);
  free (h_counts); free (h_ids);
  for (a = 0; a < n; a++) tjamd_counter_destroy (ctr[a]);
  return 0;
}
