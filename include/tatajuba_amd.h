/* tatajuba_amd.h -- C-ABI extension of the drop-in boundary (tatajuba_hopo.h) for callers that hold their reads in
 * device memory, drive several GPUs, or want to inspect the device-side state.  Everything is extern "C", plain
 * pointers and sizes; `void *hip_stream` is a hipStream_t (NULL = the counter's own stream).
 *
 * A "stream of reads" (the batch format of the device path) is a byte buffer in which every read is followed by one
 * '\n' (0x0A) -- the one byte tatajuba's parser can never deliver inside a sequence (reference: src/kseq.h:105,189-192).
 * It carries read boundaries in-band, so the scan kernel needs no offset table.
 *
 * Reference interfaces replaced:
 *   tjamd_scan_*      : the loop `while (kseq_read) update_hopo_counter_from_seq(...)`   src/hopo_counter.c:153,219-258,285-307
 *   tjamd_finalise    : finalise_hopo_counter() steps 1-4 + coverage                     src/hopo_counter.c:339-415,419-438
 *   tjamd_download_*  : the caller's direct reads of hc->elem / idx_* / coverage          src/context_histogram.c:231-256
 */
#ifndef TATAJUBA_AMD_H
#define TATAJUBA_AMD_H

#include "tatajuba_hopo.h"

#ifdef __cplusplus
extern "C" {
#endif

/* bit positions inside the 64-bit word of hopo_element / of a device record (see tatajuba_hopo.h) */
#define TJ_META_BASE_SHIFT   0
#define TJ_META_LEN_SHIFT    2
#define TJ_META_COUNT_SHIFT  12
#define TJ_META_MISM_SHIFT   32
#define TJ_META_FLAG_SHIFT   49
#define TJ_META_RAW_CONST    ((1ULL << TJ_META_COUNT_SHIFT) | (0xffeULL << TJ_META_MISM_SHIFT)) /* count=1, mismatches=0xffe */

typedef struct tjamd_counter tjamd_counter;      /* device-side per-sample accumulator (opaque) */

/* 24-byte device record: context[0], context[1], bitfield word.  32-byte "located" record adds the global byte
 * position of the tract's first base in the scanned stream (test / CPU-entry aid; gives emission order and read_offset). */
typedef struct { uint64_t ctx0, ctx1, meta; } tjamd_record;
typedef struct { uint64_t ctx0, ctx1, meta, pos; } tjamd_located_record;

/* error codes (0 = ok) */
enum { TJAMD_OK = 0, TJAMD_ERR_NO_DEVICE = 1, TJAMD_ERR_HIP = 2, TJAMD_ERR_ARG = 3, TJAMD_ERR_CAPACITY = 4, TJAMD_ERR_STATE = 5 };

int  tjamd_device_count (void);                   /* number of visible HIP devices (0 => every entry below fails loudly) */
const char *tjamd_last_error (void);              /* thread-local message of the last failure */
const char *tjamd_version (void);        /* name, version and the hash of the sources the library was built from */
const char *tjamd_source_hash (void);    /* that hash alone (tatajuba_amd/build.py computes the same over the tree) */

tjamd_counter *tjamd_counter_create (int device, int kmer_size);   /* NULL on failure (see tjamd_last_error) */
void tjamd_counter_destroy (tjamd_counter *c);
int  tjamd_counter_reset (tjamd_counter *c);      /* forget raw and finalised records, keep buffers */
int  tjamd_counter_set_stream (tjamd_counter *c, void *hip_stream); /* run on the caller's stream (e.g. torch's) */
int  tjamd_counter_device (const tjamd_counter *c);

/* Scan a stream of reads already resident in this counter's device memory.  d_stream must be 16-byte aligned.
 * Asynchronous; appends to the counter's raw records. */
int  tjamd_scan_device (tjamd_counter *c, const void *d_stream, size_t n_bytes, int min_tract_size);
/* Same from host memory (copied to HBM first). */
int  tjamd_scan_host (tjamd_counter *c, const void *h_stream, size_t n_bytes, int min_tract_size);
/* Located variant: records go to a separate list with positions, sorted by position (emission order) on download.
 * min_tract_size == 0 selects the all-monomers scan (reference: src/hopo_counter.c:260-283). */
long tjamd_scan_host_located (tjamd_counter *c, const void *h_stream, size_t n_bytes, int min_tract_size,
                              tjamd_located_record *out, long capacity);

/* Many short strings in one launch: what the reference does with one counter and one update_hopo_counter_from_seq call per
 * reference window (src/genome_set.c:525-577), batched.  seqs[i] / lens[i]: the windows; min_tract_size as in
 * update_hopo_counter_from_seq (0: the all-monomers scan).  out: hopo_element[capacity] in window order, then read order,
 * with read_offset relative to the window; window_of[i] (may be NULL) = window of record i.  Uses the calling thread's
 * shared device context.  Returns the number of records, -1 on error (tjamd_last_error) or if capacity is too small. */
long tjamd_scan_windows (int kmer_size, const char *const *seqs, const int *lens, int n_windows, int min_tract_size,
                         hopo_element *out, int *window_of, long capacity);

/* Host-only: parse a FASTA/FASTQ file (plain or gzip; same record semantics as the reference's reader, src/kseq.h:172-212
 * as looped at src/hopo_counter.c:153) into a stream of reads.  Returns the stream's size in bytes and writes it to out
 * when capacity suffices (call with out = NULL to size); *n_reads = records parsed; -1 if the file cannot be opened. */
long tjamd_read_file_stream (const char *path, unsigned char *out, long capacity, long *n_reads);

/* The same through the multi-threaded feeder (tatajuba_amd/csrc/feeder.c): n_threads readers over window_bytes of
 * file bytes at a time (0 = default), their outputs accepted only where each reader ended exactly on the next one's
 * first record -- so the result is byte-identical to tjamd_read_file_stream.  A plain file is mapped; a gzip file is
 * inflated one window ahead of the parse, BGZF (bgzip) members by all threads side by side, any other gzip stream by
 * one thread.  new_or_append_hopo_counter_from_file takes this path for plain files of 32 MiB and gzip files of 4 MiB
 * and more (TATAJUBA_AMD_FEEDER_THREADS, default min(8, cores); 1 = the single reader). */
long tjamd_read_file_stream_mt (const char *path, unsigned char *out, long capacity, long *n_reads, int n_threads, long window_bytes);

/* pinned host memory, so that tjamd_scan_host overlaps the copy with the caller's parsing; tjamd_sync waits for
 * everything queued on the counter's stream */
void *tjamd_host_alloc (size_t bytes);
void tjamd_host_free (void *p);
int  tjamd_sync (tjamd_counter *c);

/* device memory on the counter's device for callers without HIP headers of their own (buffers for tjamd_merge_samples,
 * tjamd_tract_ids); tjamd_device_download waits for the counter's stream, then copies to the host */
void *tjamd_device_alloc (tjamd_counter *c, size_t bytes);
void tjamd_device_free (tjamd_counter *c, void *p);
int  tjamd_device_download (tjamd_counter *c, void *host, const void *dev, size_t bytes);

/* Hint: reads of about stream_bytes in total are coming.  Allocates the raw-record storage for them in one piece (up to
 * 4 GB) instead of by repeated growth. */
int  tjamd_reserve (tjamd_counter *c, size_t stream_bytes, int min_tract_size);

/* A mark is a point in the counter's stream: tjamd_mark() returns a small handle (>= 0; < 0 on error), tjamd_wait_mark()
 * returns once everything queued before the mark has finished -- without waiting for what was queued after it (how the
 * feeder re-uses a pinned batch buffer while later batches are in flight).  Eight marks are live at a time. */
int  tjamd_mark (tjamd_counter *c);
int  tjamd_wait_mark (tjamd_counter *c, int mark);

long tjamd_raw_count (tjamd_counter *c);          /* synchronises; number of raw records so far; <0 on error */
long tjamd_download_raw (tjamd_counter *c, tjamd_record *out, long capacity); /* unordered multiset */
long tjamd_undefined_runs (tjamd_counter *c);     /* qualifying non-ACGTU runs with no earlier tract in the read (dropped) */
/* append host-produced raw records (hopo_element array) to the device raw list */
int  tjamd_upload_raw (tjamd_counter *c, const hopo_element *elems, long n);

/* steps 1-4 + coverage on the device.  status: 0 ok, 1 no raw records, 2 nothing after filter, 3 nothing reaches
 * min_coverage (reference: src/hopo_counter.c:345-349,376-381,406-411). */
int  tjamd_finalise (tjamd_counter *c, int remove_biased, int min_coverage, int *status);
/* The same in two calls, for a caller with more samples than GPUs: _begin queues the whole device finalise on the counter's
 * stream and returns; _end waits until THIS counter's counts have reached the host (an event, not the stream: the next
 * sample's scan, queued on another counter of the same stream in between, runs on) and returns what tjamd_finalise
 * returns.  Between the two calls the counter must not be touched. */
int  tjamd_finalise_begin (tjamd_counter *c, int remove_biased, int min_coverage);
/* A second HIP stream for the ordering step (bin partition, sort, index, coverage: five small launches whose time is latency)
 * of a finalise begun with tjamd_finalise_begin: it then runs behind an event, beside whatever the counter's own stream
 * does next -- the next sample's scan on another counter.  NULL: none (everything on the counter's stream, the default;
 * measured on one MI355X: no gain while a scan fills the device). */
int  tjamd_counter_set_order_stream (tjamd_counter *c, void *hip_stream);
int  tjamd_finalise_end (tjamd_counter *c, int *status);
long tjamd_kept_count (tjamd_counter *c);
int  tjamd_n_idx (tjamd_counter *c);
int  tjamd_coverage (tjamd_counter *c);
long tjamd_download_kept (tjamd_counter *c, hopo_element *out, long capacity);      /* widened to 40-byte elements */
long tjamd_download_idx (tjamd_counter *c, int *idx_initial, int *idx_final, long capacity);
const void *tjamd_kept_device_ptr (tjamd_counter *c);   /* tjamd_record[kept_count] in HBM (for collectives) */

/* cross-sample merge on one device (reference precursor of src/genome_set.c:250-289, keyed by context instead of
 * BWA location): concatenation of n_samples kept arrays (d_records, counts[]) -> sorted union with per-sample counts.
 * out_keys: tjamd_record[n_union] (count field = total over the samples, canon_flag = OR of the samples' flags),
 * out_counts: int32[n_union * n_samples].  Returns n_union. */
long tjamd_merge_samples (tjamd_counter *c, const void *d_records, const long *counts, int n_samples,
                          void *d_out_keys, void *d_out_counts, long capacity);

/* The exchange of that merge for a caller that, like the reference, runs its samples as threads of one process
 * (reference: src/genome_set.c:66-94 OpenMP loop, merge at :195-229): the kept records of the finalised counters
 * `samples`, whatever devices they live on, copied back to back into a buffer on dst's device (peer copies over xGMI).
 * counts[i] = records of sample i, *d_records = the buffer (owned by dst until its next gather).  Returns the total. */
long tjamd_gather_histograms (tjamd_counter *dst, tjamd_counter *const *samples, int n_samples, const void **d_records, long *counts);

/* which device pairs the gathers of this process have used, and how: "dst<-src:direct" (peer access, xGMI) or
 * "dst<-src:staged" (peer access refused: the runtime copies through host memory).  Returns the number of staged pairs. */
int tjamd_peer_access_report (char *out, int capacity);

/* The same exchange between PROCESSES, one per GPU (north_star: "an RCCL all-gatherv over xGMI of the per-sample
 * histograms"; reference attach point src/genome_set.c:195-229, where the samples are threads and nothing moves): every
 * rank contributes its finalised counter's kept records and receives every rank's, in rank order, back to back in a device
 * buffer the communicator owns (valid until its next exchange), ready for tjamd_merge_samples.  Two calls set it up:
 *   rank 0:      tjamd_comm_unique_id (id)            -> TJAMD_COMM_ID_BYTES bytes to hand to the other ranks (MPI_Bcast, a file, ...)
 *   every rank:  tjamd_comm_create (c, id, rank, world)   collective; the communicator is bound to c's device
 * and tjamd_allgather_histograms (c, comm, &d_records, counts[world]) is the exchange: ncclAllGather on c's stream, one
 * block per rank (its count, then its records), block size agreed from the counts of the exchange before; returns the total
 * number of records, counts[r] = records of rank r.  Collective: every rank calls it, in the same order. */
#define TJAMD_COMM_ID_BYTES 128
typedef struct tjamd_comm tjamd_comm;
int  tjamd_comm_unique_id (void *id_bytes);
tjamd_comm *tjamd_comm_create (tjamd_counter *c, const void *id_bytes, int rank, int world);
void tjamd_comm_destroy (tjamd_comm *comm);
/* run the exchanges on this HIP stream instead of the exchanged counter's (null: back to the counter's): a finalised
 * sample's exchange then runs beside the next sample's scan.  The caller must have seen the counter's finalise end. */
int  tjamd_comm_set_stream (tjamd_comm *comm, void *hip_stream);
int  tjamd_comm_rank (const tjamd_comm *comm);
int  tjamd_comm_world (const tjamd_comm *comm);
int  tjamd_comm_count (const tjamd_comm *comm);            /* ranks RCCL itself reports for the communicator (ncclCommCount); -1 on failure */
/* the last exchange on this communicator: device milliseconds from the first pack to the last unpack (HIP events on the
 * exchange's stream), bytes the data collective delivered to this rank, collectives it took; any pointer may be NULL;
 * non-zero before the first exchange */
int  tjamd_comm_last_exchange (const tjamd_comm *comm, double *ms, long *bytes, long *collectives);
long tjamd_comm_collectives (const tjamd_comm *comm);    /* RCCL calls issued so far (diagnostic: one per exchange once the block size has settled) */
long tjamd_allgather_histograms (tjamd_counter *c, tjamd_comm *comm, const void **d_records, long *counts);

/* tract ids on a merged union (reference: src/genome_set.c:207-221, context-keyed: the id goes up wherever
 * (base, ctx0, ctx1) changes between neighbours of d_keys = tjamd_record[n] in the reference's descending order).
 * d_tract_id (device, may be NULL) and / or h_tract_id (host, may be NULL) receive int32[n].  Returns the number of ids. */
long tjamd_tract_ids (tjamd_counter *c, const void *d_keys, long n, int *d_tract_id, int *h_tract_id);

/* within-sample grouping of near-identical contexts on a finalised counter (reference: new_genomic_context_list,
 * src/context_histogram.c:245-270 with the Hamming distance of :25-48, on the finalised array's own order; no
 * Levenshtein retry).  group_of: int32[kept_count] (host, may be NULL); groups: one entry per group (host, may be NULL):
 * first element, elements, distinct contexts, element with the modal count, summed count.  Returns the number of groups. */
typedef struct { int first, n_elem, n_context, mode; long long integral; } tjamd_group;
long tjamd_group_contexts (tjamd_counter *c, int max_distance_per_flank, int *group_of, tjamd_group *groups, long capacity);

/* The whole grouping step (reference: new_genomic_context_list, src/context_histogram.c:245-270 and :278-286): the flank
 * distance test as above, then, for an element of the histogram's base that fails it, the retry with an edit distance
 * between the "left.B.right" names of the histogram's modal context and of the element (:19-23,255-261: joins if it is below
 * levenshtein_distance and marks the histogram `indel`), then every histogram's tract lengths weighted by count, highest
 * count first (:282 new_empfreq_from_int_weighted; modal_len / modal_freq = its first entry).  The edit distance stands
 * for biomcmc_levenshtein_distance (.., 1, 1, true) of biomcmc-lib, absent from the reference tree: unit-cost global edit
 * distance; likewise the order among equal counts (larger length first).
 *   group_of   int32[kept_count]   histogram of each element (host, may be NULL)
 *   join_type  int32[kept_count]   0 = the element opened its histogram, 1 = joined within the flank distance, 2 = by the retry
 *   groups     one entry per histogram (host, may be NULL; `capacity` entries)
 *   hist       tjamd_length_freq[kept_count]: histogram g's entries at [groups[g].first, groups[g].first + groups[g].n_len)
 * Returns the number of histograms. */
typedef struct { int first, n_elem, n_context, mode, indel, n_len, modal_len, modal_freq; long long integral; } tjamd_context_group;
typedef struct { int length, freq; } tjamd_length_freq;
long tjamd_context_histograms (tjamd_counter *c, int max_distance_per_flank, int levenshtein_distance, int *group_of, int *join_type,
                               tjamd_context_group *groups, tjamd_length_freq *hist, long capacity);

/* release the calling thread's shared device contexts of the synchronous string scans (update_hopo_counter_from_seq on a
 * counter that never read a file, tjamd_scan_windows) now; they are released by themselves when the thread ends */
void tjamd_thread_cleanup (void);

/* timing of the last operations on this counter, from HIP events on its stream (milliseconds) */
double tjamd_last_scan_ms (tjamd_counter *c);       /* scan kernel(s) of the last tjamd_scan_* call */
int    tjamd_counter_uses_log (const tjamd_counter *c);  /* 1: k <= 12 and the scan writes a record log that partition_log_kernel distributes (default); 0: the scan kernels partition by themselves (k > 12, or TATAJUBA_AMD_SINK=fused) */
double tjamd_last_partition_ms (tjamd_counter *c);  /* partition_log_kernel behind the last scan launch (k <= 12); 0 if the scan kernel partitioned by itself */
double tjamd_last_finalise_ms (tjamd_counter *c);
double tjamd_last_merge_ms (tjamd_counter *c);      /* kernels of the last tjamd_merge_samples on this counter */   /* whole device finalise of the last tjamd_finalise call */
long   tjamd_last_scan_launches (tjamd_counter *c);
/* finalises of this counter whose device-side sizing of the ordering step had read a stale kept count (checked against the
 * count at the next kernel boundary and repaired; expected to stay 0) */
long   tjamd_plan_mismatches (tjamd_counter *c);

/* synthetic inputs: genome of `genome_len` i.i.d. bases from splitmix64(seed_genome); n_reads reads of
 * length read_len (or uniform in [read_len, read_len_max] when read_len_max > read_len), uniform start, strand by coin,
 * no N, written as a stream of reads into out (capacity bytes).  variant_seed != 0 lengthens/shortens 1% of the
 * genome's tracts >= 4 by one base first.  Returns bytes written, or -(bytes needed) if capacity is too small. */
long tjamd_synth_stream (uint64_t seed_genome, uint64_t seed_reads, uint64_t variant_seed, long genome_len,
                         long n_reads, int read_len, int read_len_max, unsigned char *out, long capacity, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
