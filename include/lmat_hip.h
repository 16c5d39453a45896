/* lmat_hip.h -- C ABI of the MI355X-native read-labeling engine (liblmat_hip.so).
 *
 * The reference (LivGen/LMAT) has no plugin/FFI interface for this path; the
 * engine is a drop-in at the three seams SURVEY.md 8(b) names.  Each entry
 * point below cites the reference interface it replaces (paths relative to the
 * LMAT source tree).  Plain pointers and sizes only; every function returns 0
 * on success or a negative LMAT_E_* code and never throws; lmat_last_error()
 * gives the message.  One context per GPU; distinct contexts are independent.
 *
 * The engine has NO CPU fallback: if no HIP device is usable, lmat_ctx_create
 * fails with LMAT_E_DEVICE.
 */
#ifndef LMAT_HIP_H
#define LMAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LMAT_OK 0
#define LMAT_E_ARG (-1)       /* bad argument / call order */
#define LMAT_E_IO (-2)        /* file missing or malformed */
#define LMAT_E_DEVICE (-3)    /* HIP error / no device */
#define LMAT_E_CAPACITY (-4)  /* a fixed on-device capacity was exceeded (message says which) */
#define LMAT_E_TAXONOMY (-5)  /* inconsistent taxonomy inputs */
#define LMAT_E_NOMEM (-6)

typedef struct lmat_ctx lmat_ctx;
typedef struct lmat_reads lmat_reads;
typedef struct lmat_ingest lmat_ingest;
typedef struct lmat_stream lmat_stream;

/* ScoreOptions + the scalar thresholds proc_line takes
 * (src/read_label.cpp:487-497, :1211-1212, getopt cases :1353-1441). */
typedef struct {
    float sdiff;        /* -b  ScoreOptions::_diff_thresh   (default 1.0) */
    float hbias;        /* -l  ScoreOptions::_diff_thresh2  (default 3.0) */
    float min_score;    /* -x  min_label_score              (default 0)   */
    int32_t min_kmer;   /* -j  min valid k-mers             (default 35)  */
    int32_t min_fnd_kmer; /* -z                             (default 1)   */
    int32_t prn_all;    /* -p  ScoreOptions::_prn_all                      */
    int32_t screen_phix; /* 1 unless -h (screenPhiXGlobal)                 */
} lmat_params;

/* What one .out record needs (src/read_label.cpp:844-848,894-937,1218,1233,1271). */
enum {
    LMAT_ST_CALL = 0,           /* stats + candidates + call line              */
    LMAT_ST_PHIX = 1,           /* PhiX short-circuit record (:841-848)        */
    LMAT_ST_SHORT_LEN = 2,      /* ri_len < k      ReadTooShort (:1217-1218)   */
    LMAT_ST_SHORT_VALID = 3,    /* valid < min     ReadTooShort (:1232-1233)   */
    LMAT_ST_NODBHITS = 4,       /* no taxid registered (:1271)                 */
    LMAT_ST_SILENT = 5          /* construct_labels returned before writing (:727-733): no record text, tallied NoDbHits */
};
enum { LMAT_MT_DIRECT = 0, LMAT_MT_MULTI = 1, LMAT_MT_PARTIAL = 2, LMAT_MT_NOMATCH = 3, LMAT_MT_LCA_ERROR = 4 };

typedef struct {
    uint8_t status;        /* LMAT_ST_*                                         */
    uint8_t match_type;    /* LMAT_MT_* (status CALL only)                      */
    uint16_t cand_kmer_cnt; /* distinct valid k-mers with first>=0              */
    int32_t valid_kmers;   /* valid k-mer positions (dups included)             */
    int32_t read_len;
    float log_avg;         /* exact float bits; host formats with %g            */
    float stdev;
    uint32_t call_tid;     /* 32-bit NCBI id (0 => printed as -1 -1)            */
    float call_score;
    uint32_t cand_off;     /* first candidate of this read in the cands array   */
    uint32_t n_cand;       /* candidates to print, best first                   */
    int32_t bin_sel;       /* GC decile (read_label.cpp:1205-1206), the null-model input: 0 unless null models are loaded */
} lmat_read_result;

typedef struct {
    uint32_t tid;
    float score;
} lmat_cand;

/* ---- context ------------------------------------------------------------ */
int lmat_device_count(void);   /* usable HIP devices (0 when there is none) */
int lmat_ctx_create(int device, const lmat_params* params, lmat_ctx** out);
void lmat_ctx_destroy(lmat_ctx* ctx);
const char* lmat_last_error(const lmat_ctx* ctx);
int lmat_set_params(lmat_ctx* ctx, const lmat_params* params);

/* ---- taxonomy + id maps ---------------------------------------------------
 * Replaces TaxTree<uint32_t>(file) + getPathToRoot (src/kmerdb/TaxTree.hpp:24-91),
 * the depth map (read_label.cpp:1574-1582), gRank_table (:1560-1567), conv_map
 * (:1585-1602) and gLowNumPlasmid (:499-510).  rank_fn / plasmid_fn may be NULL.
 * idmap_fn NULL = a database of 32-bit taxids (TID_SIZE=32 build): see lmat_ingest_idmap_from_tree. */
int lmat_taxonomy_load_files(lmat_ctx* ctx, const char* tree_fn, const char* depth_fn, const char* rank_fn,
                             const char* idmap_fn, const char* plasmid_fn);

/* ---- k-mer database --------------------------------------------------------
 * Replaces the PERM-mapped SortedDb (read_label.cpp:1477-1491; lookup contract
 * SortedDb::begin_/next, src/kmerdb/SortedDb.hpp:188,366, wrapped by
 * TaxNodeStat::begin/next/taxid/taxidCount, src/kmerdb/TaxNodeStat.hpp:60,208,258,262)
 * by an open-addressed hash in HBM.  Ingest format = make_db_table's input
 * (tax_histo binary, SortedDb::add_data src/kmerdb/SortedDb.cpp:84-751, un-pruned path).
 * table_bytes == 0 sizes the table for a 0.8 load factor.
 * n_kmers_hint or table_bytes non-zero: the table is allocated at once and the files stream through the insert
 * kernel in chunks (only the distinct lists stay on the host); then n_kmers_hint must not be below the real
 * count, the label / rand modes must be set before lmat_db_begin, and lmat_db_save_image is unavailable.
 * Both zero: the k-mers are buffered on the host until lmat_db_finalize. */
int lmat_db_begin(lmat_ctx* ctx, int k, uint64_t n_kmers_hint, uint64_t table_bytes);
int lmat_db_add_taxhisto(lmat_ctx* ctx, const char* fn);
int lmat_db_finalize(lmat_ctx* ctx);
/* make_db_table's content-changing options (src/make_db_table.cpp:150-213,303-313): build-time pruning of
 * lists longer than tid_cutoff by the rank map (-g N -m file; SortedDb.cpp:296-409), human k-mer feed (-j;
 * SortedDb.cpp:170-233,475-530,664-715), adaptor k-mer feed (-u; SortedDb.cpp:190-204,275-292).  Call between
 * lmat_db_begin and the first lmat_db_add_taxhisto; NULL / 0 disables an option. */
int lmat_db_set_build_options(lmat_ctx* ctx, int tid_cutoff, const char* rank_map_fn, const char* human_kmers_fn,
                              const char* adaptor_kmers_fn, uint32_t adaptor_tid);
/* Engine-native image of the ingested database (replaces the PERM heap image make_db_table leaves on disk,
 * src/make_db_table.cpp:330-343,429): save between begin and finalize; load + finalize to use it. */
int lmat_db_save_image(lmat_ctx* ctx, const char* fn);
int lmat_db_load_image(lmat_ctx* ctx, const char* fn, uint64_t table_bytes);
/* Two image formats, told apart by their first 8 bytes:
 *   LMATIMG1  the ingest's view (sorted k-mers + canonical 16-bit lists; what make_db_image writes without a GPU): saved between
 *             lmat_db_begin and lmat_db_finalize; loading it sizes a table and inserts every k-mer on the GPU.
 *   LMATIMG2  the DEVICE layout itself (bucket array, overflow table, list arena, geometry): lmat_db_save_image on a FINALIZED
 *             database streams it out of HBM, lmat_db_load_image streams it back in through pinned buffers -- the counterpart of
 *             the reference mapping its database file (src/read_label.cpp:1477-1491); lmat_db_finalize is then a no-op.  Its list
 *             records hold internal taxid indices built under the context's label modes: loading checks fingerprints of both
 *             (LMAT_E_TAXONOMY / LMAT_E_ARG on a mismatch).  table_bytes is ignored for it. */
/* A replica of src's finalized database in dst (another GPU of the node, or the same one): the k-mer table, its
 * overflow table and the list arena are copied device to device (hipMemcpyPeer: over xGMI between GPUs), instead of
 * every GPU parsing the files again.  dst must hold the same taxonomy and label modes as src and no database yet. */
int lmat_db_clone(lmat_ctx* dst, lmat_ctx* src);
int lmat_db_kmer_length(const lmat_ctx* ctx);   /* SortedDb::get_kmer_length (SortedDb.hpp:433) */
uint64_t lmat_db_size(const lmat_ctx* ctx);      /* SortedDb::size (SortedDb.hpp:438)            */
uint64_t lmat_db_list_count(const lmat_ctx* ctx); /* distinct taxid-list records of a synthetic database (0: not counted) */
uint64_t lmat_db_arena_bytes(const lmat_ctx* ctx); /* size of the taxid-list arena on the device */
uint64_t lmat_db_table_bytes(const lmat_ctx* ctx);

/* TaxNodeStat-style lookup of n k-mers on the GPU: counts[i] = taxidCount (0 = miss),
 * tids[i*stride .. ] = taxid() sequence (32-bit, stored order), up to stride each. */
int lmat_db_lookup(lmat_ctx* ctx, const uint64_t* kmers, uint64_t n, uint32_t* counts, uint32_t* tids, uint32_t stride);

/* Synthetic DB generated on the device straight into the hash (bench configs;
 * SURVEY.md 8d).  Requires lmat_synth_taxonomy() first.  genome_len = bases per
 * strain genome. */
int lmat_synth_taxonomy(lmat_ctx* ctx, const uint32_t* branching6);
int lmat_synth_db_build(lmat_ctx* ctx, int k, uint64_t genome_len, uint64_t seed, uint64_t table_bytes);
/* ... with the share (in 1/1000) of every genome that is a block shared by the species of its genus (SURVEY 8d asks
 * for 10 %: lmat_synth_db_build passes 100) and the number of copies of the taxid-list records the payloads are spread
 * over (1 = every list once; more make the arena as large as a real database's, beyond the caches). */
int lmat_synth_db_build2(lmat_ctx* ctx, int k, uint64_t genome_len, uint64_t seed, uint64_t table_bytes,
                         uint32_t genus_block_permille, uint32_t list_replicas);
/* ... and with a HEAVY TAIL of taxid lists (real LMAT lists run to thousands of ids, doc/lmat-doc.txt:914-931): behind the genus
 * block three conserved blocks of conserved_permille3[0..2] thousandths of every genome, shared -- without strain substitutions --
 * by all species of a family, of a phylum, of a superkingdom: k-mers whose lists hold every strain, species and inner node of the
 * group (69 / 277 / 1109 taxids in the bench taxonomy).  {0, 0, 0} is lmat_synth_db_build2. */
int lmat_synth_db_build3(lmat_ctx* ctx, int k, uint64_t genome_len, uint64_t seed, uint64_t table_bytes,
                         uint32_t genus_block_permille, uint32_t list_replicas, const uint32_t* conserved_permille3);

/* Measurement hook: the 16 per-launch counters of the most recent launch, read after lmat_sync: [0] candidate cursor, [2] reads the
 * fast classes passed on to the E = 512 class, [3] reads that class passed on to the middle tier (T = 256), [10] reads the middle
 * tier passed on to the large LDS class (T = 1024), [7] reads that class passed on to the global-memory class. */
int lmat_debug_last_counters(lmat_ctx* ctx, uint32_t* out16);

/* Measurement hook: where the lookups of these (forward-encoded) k-mers end in the compact table -- out5[0] found in the home
 * bucket, [1] absent and the bucket never spilled (one 64-byte request in all), [2] found in the overflow table, [3] absent after
 * the overflow table was asked too, [4] overflow buckets read.  bench.py reports the share of a read sample's lookups that need
 * the second request (`overflow_probe_share`). */
int lmat_debug_probe_stats(lmat_ctx* ctx, const uint64_t* kmers, uint64_t n, uint64_t* out5);
/* Debug: the 4-instruction division the decision step on the wave uses for count / candidate-k-mers (integers below 1024,
 * read_label.cpp:821) against the IEEE division on every pair of its domain, and -- the two averages of the statistics, :806-880 --
 * on 2^26 drawn pairs (float >= 0, integer 1..64): out4 = {pairs that differ, pairs tried, drawn pairs that differ, drawn pairs tried}. */
int lmat_debug_div_check(lmat_ctx* ctx, uint64_t* out4);

/* Test hook for the synthetic database: what it must hold for the window of `species` (0-based) that starts at base `pos` of the
 * species ancestor -- the canonical k-mer and its taxid list (the strains that carry the window unmutated; with two or more
 * owners also their species, and the genus when they span species) -- computed on the HOST from the generator's functions, never
 * read from the device table.  *n = 0: no strain kept the window.  (The table itself may hold a different list for that k-mer
 * when another window yields the same 20-mer: about 1 % of them at 6.4 G k-mers; the smaller payload wins.) */
int lmat_synth_window(lmat_ctx* ctx, uint32_t species, uint64_t pos, uint64_t* kmer, uint32_t* tids, uint32_t cap, uint32_t* n);
/* ... and for a synthetic READ: the k-mers read r of lmat_reads_synth(lengths, n_lengths, seed) must find, with their lists, derived
 * on the host from the read generator and the genome generator alone -- every window that lies in the read's strain genome
 * without a substituted base: the ancestor window's list where the strain carries it unmutated, else the strain alone.  Windows
 * with an error, random and low-complexity reads yield nothing (what the table returns for those is a chance hit).
 * kmers[cap], tids[cap][stride], counts[cap]; *n = windows written. */
int lmat_synth_read_windows(lmat_ctx* ctx, const uint32_t* lengths, uint32_t n_lengths, uint64_t seed, uint64_t r, uint64_t* kmers,
                            uint32_t* tids, uint32_t* counts, uint32_t cap, uint32_t stride, uint32_t* n, uint32_t* read_len);

/* ---- gene databases (src/gene_label.cpp) ---------------------------------------------------
 * gene_label runs the same lookup against a database whose lists are 32-bit GENE ids (INDEXDB<uint32_t>, TaxNodeStat
 * <uint32_t>, gene_label.cpp:15-22,218-267) and has no taxonomy step.  A context opened with lmat_genedb_begin (instead of a
 * taxonomy + lmat_db_begin) ingests such a database -- same tax_histo record format, ids stored as read --; then
 * lmat_db_add_taxhisto / lmat_db_finalize / lmat_db_lookup work as usual, and every classification entry point (lmat_classify,
 * lmat_stream_*) returns gene_label's vote instead of the taxonomic call (proc_line, gene_label.cpp:269-313):
 *   status LMAT_ST_CALL: call_tid = the gene std::sort(Cmp) puts first, n_cand = its vote count, cand_kmer_cnt = distinct
 *   valid k-mers of the read, call_score = n_cand / cand_kmer_cnt;  LMAT_ST_NODBHITS / LMAT_ST_SHORT_LEN: no gene (upstream
 *   prints nothing for such a read). */
int lmat_genedb_begin(lmat_ctx* ctx, int k, uint64_t n_kmers_hint, uint64_t table_bytes);

/* ---- label modes -------------------------------------------------------------------
 * -s permissive match (gPERMISSIVE_MATCH, read_label.cpp:1050-1058,1075-1102,1143) and run-time pruning of
 * lists longer than tid_cutoff (-g N [-m numeric ranks]; TaxNodeStat::begin, TaxNodeStat.hpp:76-203).  Both are
 * functions of a k-mer's taxid list alone, so they are applied when the list records are built: call before
 * lmat_db_finalize.  rank_map_fn may be NULL (then only the first stored taxid survives, as upstream). */
int lmat_set_label_modes(lmat_ctx* ctx, int permissive, int tid_cutoff, const char* rank_map_fn);

/* ---- null models (-n) -----------------------------------------------------------
 * Replaces loadRandHits (read_label.cpp:512-678) and switches scoring to log(label_prob / null_prob)
 * (construct_labels :735-819, log_odds_score :680-690).  list_fn holds `<kmer_count> <gz file relative to
 * $LMAT_DIR>` lines.  A taxid met during classification without a null-model entry is an error
 * (LMAT_E_TAXONOMY), as upstream's assert is. */
int lmat_nullmodel_load(lmat_ctx* ctx, const char* list_fn);
int lmat_nullmodel_clear(lmat_ctx* ctx);

/* ---- reads ----------------------------------------------------------------
 * A batch of reads packed on the device (2-bit bases + validity bits).
 * Replaces the (read,hdr) queue hand-off of main() (read_label.cpp:1716-1746):
 * bases = concatenated ASCII, off[n+1] byte offsets. */
int lmat_reads_upload(lmat_ctx* ctx, const uint8_t* bases, const uint64_t* off, uint64_t n, lmat_reads** out);
int lmat_reads_synth(lmat_ctx* ctx, uint64_t n, const uint32_t* lengths, uint32_t n_lengths, uint64_t seed,
                     lmat_reads** out);
int lmat_reads_download_ascii(lmat_ctx* ctx, const lmat_reads* r, uint64_t first, uint64_t count, uint8_t* bases,
                              uint64_t* off);
uint64_t lmat_reads_count(const lmat_reads* r);
uint64_t lmat_reads_device_bytes(const lmat_reads* r);
void lmat_reads_free(lmat_ctx* ctx, lmat_reads* r);

/* ---- classification --------------------------------------------------------
 * Replaces proc_line (read_label.cpp:1211-1279) for reads [first, first+count).
 * results[count]; cands/cand_cap may be NULL/0 for calls-only output; *n_cands
 * receives the number of candidate pairs written.  Per-taxid tallies
 * (read_label.cpp:1241-1276) accumulate in the context until lmat_counts_reset; a call that fails (e.g.
 * LMAT_E_CAPACITY because cand_cap was too small) leaves them as they were, so it can simply be retried. */
int lmat_classify(lmat_ctx* ctx, const lmat_reads* reads, uint64_t first, uint64_t count, lmat_read_result* results,
                  lmat_cand* cands, uint64_t cand_cap, uint64_t* n_cands);

/* Same kernels, results left in device memory (throughput runs).  Asynchronous on
 * the context's stream; lmat_sync waits and reports an error raised by ANY launch since the last report
 * (device-side error flags are sticky until reported).  kernel_ms_total (may be NULL) = HIP-event time of
 * all kernels of those launches; lmat_last_timing splits it per kernel.  The device-side result
 * buffer holds the records of the most recent launch only (lmat_results_fetch reads those); the
 * tallies accumulate over all launches. */
int lmat_classify_async(lmat_ctx* ctx, const lmat_reads* reads, uint64_t first, uint64_t count);
/* The same with the candidate pairs of `-p` (read_label.cpp:898-910; bin/run_rl.sh:243 always passes -p) written to a
 * device-side buffer of cand_cap pairs (grown on demand, one per set of in-flight buffers); lmat_read_result.cand_off /
 * n_cand index it.  cand_cap = 0: calls only, as lmat_classify_async.  A buffer that proves too small raises
 * LMAT_E_CAPACITY at the next lmat_sync. */
int lmat_classify_async_cands(lmat_ctx* ctx, const lmat_reads* reads, uint64_t first, uint64_t count, uint64_t cand_cap);
int lmat_sync(lmat_ctx* ctx, float* kernel_ms_total, uint64_t* kernel_launches);
/* HIP-event times accumulated by the launches the last lmat_sync waited for, split per kernel:
 * classify_ms = classify_kernel (extract + probe + registration/closure, the HBM-bound kernel),
 * decide_ms = the K4 kernels (score + LCA decision) + the larger-class re-runs. */
int lmat_last_timing(const lmat_ctx* ctx, float* classify_ms, float* decide_ms, uint64_t* launches);
int lmat_results_fetch(lmat_ctx* ctx, uint64_t first, uint64_t count, lmat_read_result* results);

/* ---- streamed boundary -------------------------------------------------------
 * Replaces the reader thread + (read, hdr) queue of main() (src/read_label.cpp:1651-1746): a ring of n_slots batch slots
 * whose pinned host buffers and device buffers are allocated once.  Host-to-device copy, packing + classification and
 * the copy of the results back overlap across consecutive batches; nothing is allocated per batch.
 *   acquire  -> pointers to the next slot's pinned input buffers: bases (concatenated ASCII, up to max_bases) and
 *               off[n + 1] byte offsets; the caller fills them (LMAT_E_ARG when every slot is in use)
 *   submit   -> queues the slot: n reads, an opaque tag that comes back with the results; asynchronous
 *   next     -> waits for the OLDEST submitted batch; pointers into its pinned result buffers (cands in the order of
 *               lmat_read_result.cand_off), valid until lmat_stream_release; returns 1 when nothing is in flight
 *   release  -> the batch handed out by lmat_stream_next is consumed, its slot is free again
 * cands_per_read = 0: calls only; otherwise the slot holds cands_per_read x max_reads candidate pairs and grows itself
 * when a batch needs more.  Tallies accumulate in the context as with lmat_classify.  One thread drives a stream. */
int lmat_stream_create(lmat_ctx* ctx, uint64_t max_reads, uint64_t max_bases, uint32_t cands_per_read, int n_slots,
                       lmat_stream** out);
int lmat_stream_acquire(lmat_stream* st, uint8_t** bases, uint64_t** off);
int lmat_stream_submit(lmat_stream* st, uint64_t n_reads, uint64_t tag);
/* As lmat_stream_submit, but the reads are off[0..n_reads] inside the caller's OWN pinned buffer `bases` (from
 * lmat_host_alloc; off need not start at 0), which the device reads directly: no copy into the slot.  The buffer must
 * stay untouched until lmat_stream_next has returned this batch. */
int lmat_stream_submit_from(lmat_stream* st, const uint8_t* bases, const uint64_t* off, uint64_t n_reads, uint64_t tag);
/* Page-locked host memory for buffers handed to lmat_stream_submit_from. */
int lmat_host_alloc(uint64_t bytes, void** out);
void lmat_host_free(void* p);
int lmat_stream_next(lmat_stream* st, const lmat_read_result** results, const lmat_cand** cands, uint64_t* n_reads,
                     uint64_t* n_cands, uint64_t* tag);
int lmat_stream_release(lmat_stream* st);
void lmat_stream_destroy(lmat_stream* st);

/* ---- tallies (merge step read_label.cpp:1760-1800) --------------------------
 * Dense arrays indexed by the engine's internal taxid index; n_ids = number of
 * internal ids + 1.  counts_device_ptr exposes the device buffer
 * [u64 count[n_ids] | f64 score[n_ids] | u64 nomatch[3]] for an RCCL all-reduce
 * done by the caller (one process per GPU). */
int lmat_counts_reset(lmat_ctx* ctx);
int lmat_counts_layout(const lmat_ctx* ctx, uint32_t* n_ids, uint64_t* bytes);
void* lmat_counts_device_ptr(lmat_ctx* ctx);
int lmat_counts_get(lmat_ctx* ctx, uint32_t* tid32, uint64_t* count, double* score, uint32_t cap, uint32_t* n_nonzero,
                    uint64_t nomatch3[3]);
/* The same merge across the contexts of ONE process (one per GPU, read_label -t N on an N-GPU node): afterwards every
 * context holds the sum of all.  Contexts on distinct devices: RCCL over xGMI (ncclCommInitAll over their devices, kept
 * for the next call, + one grouped ncclAllReduce per array on each context's stream).  Contexts that share a device
 * cannot be ranks of one communicator: those are summed through the host. */
int lmat_counts_allreduce(lmat_ctx** ctxs, int n_ctx);

/* ... and across PROCESSES, one context (= one GPU) per rank, whatever launched them: rank 0 makes an id
 * (ncclGetUniqueId), the launcher's own channel hands its LMAT_COMM_ID_BYTES bytes to the other ranks, every rank calls
 * lmat_comm_init (ncclCommInitRank; collective), then lmat_comm_allreduce_counts sums the tallies in place on all ranks
 * (ncclAllReduce on the context's stream, u64 counts / f64 score sums / u64 nomatch; returns when it is complete).
 * librccl is opened on first use of these entry points, never by a one-GPU run. */
#define LMAT_COMM_ID_BYTES 128
/* 1 when librccl and every entry point the engine uses could be loaded in this process (a pre-flight every rank runs BEFORE
 * any rank enters lmat_comm_init, so that a rank without the library cannot leave the others waiting inside it), else 0. */
int lmat_comm_available(void);
int lmat_comm_unique_id(uint8_t* id /* [LMAT_COMM_ID_BYTES] */);
int lmat_comm_init(lmat_ctx* ctx, const uint8_t* id, int n_ranks, int rank);
int lmat_comm_allreduce_counts(lmat_ctx* ctx);
int lmat_comm_size(const lmat_ctx* ctx);   /* ranks of the communicator, 0 = none */
void lmat_comm_destroy(lmat_ctx* ctx);

/* ---- rand_read_label: the null-model generator (src/rand_read_label.cpp) on the same kernels -----------
 * Replaces its proc_line/construct_labels (:185-213, :372-398) over src/rkmer.hpp's retrieve_kmer_labels, which is
 * read_label's without the human folding.  Per (taxid, GC bucket): the largest fraction label_prob =
 * count / valid_kmers over the reads given, and the number of reads that hit the taxid -- the two columns of
 * the .rand_lst file (:741-754).  lmat_rand_mode(ctx, 1) before lmat_db_finalize; lmat_rand_reset sizes and
 * clears the tables; lmat_rand_label adds a range of reads, gc_bucket[i] (host) being the bucket of read
 * first + i; lmat_rand_get returns the rows with any hit, ascending taxid, row-major [n_rows][n_buckets]
 * (call with cap 0 for the row count). */
int lmat_rand_mode(lmat_ctx* ctx, int on);
int lmat_rand_reset(lmat_ctx* ctx, uint32_t n_buckets);
int lmat_rand_label(lmat_ctx* ctx, const lmat_reads* reads, uint64_t first, uint64_t count, const uint8_t* gc_bucket);
int lmat_rand_get(lmat_ctx* ctx, uint32_t* tid32, float* max_prob, uint32_t* count, uint32_t cap, uint32_t* n_rows);

/* ---- GPU-free ingest (the make_db_image tool; also usable without any device) ----------------------
 * Same parsing and options as above, producing the canonical (k-mer, 16-bit taxid list) form;
 * lmat_ingest_lookup returns the stored list of one k-mer (16-bit DB ids, stored order; 0 = absent). */
int lmat_ingest_create(int k, const char* idmap_fn, lmat_ingest** out);   /* idmap_fn NULL: call lmat_ingest_idmap_from_tree */
/* A database of 32-bit taxids (the reference's TID_SIZE=32 build, CMakeLists.txt:92-105; make_db_table without -f):
 * the storage code of a taxid is its rank among the taxonomy tree's node ids (at most 65534 nodes).
 * lmat_taxonomy_load_files with idmap_fn NULL derives the same codes. */
int lmat_ingest_idmap_from_tree(lmat_ingest* ing, const char* tree_fn);
void lmat_ingest_destroy(lmat_ingest* ing);
const char* lmat_ingest_error(const lmat_ingest* ing);
int lmat_ingest_set_options(lmat_ingest* ing, int tid_cutoff, const char* rank_map_fn, const char* human_kmers_fn,
                            const char* adaptor_kmers_fn, uint32_t adaptor_tid);
int lmat_ingest_add_taxhisto(lmat_ingest* ing, const char* fn);
int lmat_ingest_save_image(const lmat_ingest* ing, const char* fn);
int lmat_ingest_load_image(const char* fn, lmat_ingest** out);
uint64_t lmat_ingest_size(const lmat_ingest* ing);
int lmat_ingest_kmer_length(const lmat_ingest* ing);
int lmat_ingest_lookup(const lmat_ingest* ing, uint64_t kmer, uint16_t* tids16, int cap);
int lmat_db_from_ingest(lmat_ctx* ctx, lmat_ingest* ing, uint64_t table_bytes);

/* ---- test hook: the decision step on given candidate tables ---------------------------------
 * Runs the decision kernels' own code -- std::sort(TCmp) (src/read_label.cpp:475-485, 892-893) and findReadLabelVer2
 * (:284-419) -- on n candidate tables given from outside instead of computed from reads: table i = the (32-bit taxid,
 * score) pairs off[i] .. off[i+1] (1..64 of them, any order) and the read's standard deviation stdev[i] (the second
 * statistic read_label prints).  Needs the taxonomy only.  results[i]: call_tid, call_score, match_type.  This is how the
 * records of the reference's own example run (tests/golden/example_records.json) are put to the GPU kernels directly. */
int lmat_debug_decide(lmat_ctx* ctx, const uint32_t* tids, const float* scores, const uint64_t* off, const float* stdev,
                      uint64_t n, lmat_read_result* results);
/* The same step on (taxid, COUNT) tables and a candidate k-mer count per table, through either decision path of the engine:
 * on_the_wave = 0 the general path (k4_part1 / k4_part2: scores = count / cand, the statistics, std::sort(TCmp) replayed,
 * findReadLabelVer2 -- the code lmat_debug_decide pins to the reference's printed records), on_the_wave = 1 k4_wave, the
 * wave-parallel form the classify kernel runs for every read of the benchmark.  A table k4_wave declines (cand above 999, the
 * heapsort turn of introsort, a lineage beyond 64 entries, two lineage entries of one depth ...) comes back with status 255.
 * results[i]: status, match_type, cand_kmer_cnt, log_avg, stdev, call_tid, call_score. */
int lmat_debug_decide_counts(lmat_ctx* ctx, const uint32_t* tids, const uint32_t* counts, const uint64_t* off, const uint32_t* cand,
                             uint64_t n, int on_the_wave, lmat_read_result* results);

/* ---- measurement aid ---------------------------------------------------------
 * Random 64-byte bucket gather over the loaded table with the probe kernel's access shape;
 * reports the HIP-event time and the bytes it read (the practical ceiling of the probe). */
int lmat_gather_bench(lmat_ctx* ctx, uint64_t n_probes, uint64_t seed, float* ms, uint64_t* bytes);

/* Where the compact table layout files a k-mer (forward-encoded, either strand): bucket index and 16-bit slot tag
 * for a table of about want_buckets buckets; *n_buckets = the bucket count actually used (the 16-bit tag sets a
 * minimum).  (bucket, tag) <-> canonical k-mer is a bijection; tests check exactly that.  Needs no device.
 * LMAT_E_ARG when this k has no compact layout (k < 10). */
int lmat_table_address(int k, uint64_t want_buckets, uint64_t kmer, uint64_t* n_buckets, uint32_t* bucket, uint32_t* tag);

/* ---- record text -------------------------------------------------------------
 * The bytes read_label writes to a .out file for these results (prefix
 * read_label.cpp:1733-1738, body :844-848,894-937,1218,1233,1271).  Headers are
 * "r<first_index+i>"; bases/off may be NULL when prn_read == 0 ("X" is written, -a).
 * Returns the text length (excluding the terminating NUL); writes at most cap bytes. */
int64_t lmat_format_out(const lmat_ctx* ctx, const lmat_read_result* results, uint64_t n, const lmat_cand* cands,
                        const uint8_t* bases, const uint64_t* off, int prn_read, uint64_t first_index, char* buf,
                        uint64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* LMAT_HIP_H */
