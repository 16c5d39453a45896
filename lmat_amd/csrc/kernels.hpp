// kernels.hpp -- launch interface between the C-ABI host code and kernels.hip.
#pragma once
#include "lmat_internal.hpp"

namespace lmat {

struct ClassifyArgs {
    DeviceTables tb;
    KernelParams prm;
    const uint32_t* words;     // packed read records
    const uint64_t* rec_off;   // [n+1] word offsets
    const uint32_t* index;     // optional explicit read list (re-runs), else first+i
    uint32_t p_min = 0, p_max = 0xFFFFFFFFu;  // a re-run launch only takes the listed reads with this many k-mer positions
    uint64_t first, count;
    uint64_t result_base;      // results[r - result_base]
    lmat_read_result* results;
    lmat_cand* cands;          // may be null (calls-only)
    uint64_t cand_cap;
    uint32_t* cursor;          // per-batch counters: [0] candidate bump cursor, [2..] list lengths; behind the 16 words: kCandSubs sub-cursors (64 B apart)
    uint32_t cand_chunk = 0;   // pairs a sub-cursor takes from the bump cursor at a time (0: every read bumps the cursor itself)
    uint32_t cand_sub_mask = 0; // sub-cursors in use - 1 (a power of two, at most kCandSubs: fewer for smaller batches, whose slack would otherwise outweigh their pairs)
    uint32_t* err;             // sticky error flags: launches only OR into the word
    void* counts;              // u64 count[n_ids] | f64 score[n_ids] | u64 nomatch[3]
    uint32_t phix_call_idx;    // internal index of 32630
    uint32_t* ovf_list;        // reads that exceed this launch's capacities are appended here (count in cursor[ovf_slot])
    uint32_t ovf_slot;         // cursor word that counts ovf_list: 2 fast -> E=512 class, 3 -> middle tier, 10 -> large LDS class, 7 -> global-memory class
    const uint32_t* count_ptr; // when set, the number of `index` entries is read from device memory
    uint32_t* k4buf;           // per-read records handed from the fast classify kernel to the K4 kernels
    uint32_t* k4_small;        // read indices awaiting K4, small tables (count in cursor[4])
    uint32_t* k4_mid;          // read indices awaiting K4, up to 32 taxids (count in cursor[8])
    uint32_t* k4_large;        // read indices awaiting K4, large tables (count in cursor[5])
    uint32_t* k4_bail;         // reads the LDS K4 kernel could not hold after all (count in cursor[6])
    uint32_t* k4_row;          // read indices awaiting k4_row_kernel: up to 16 taxids, decision by rows (count in cursor[9])
    uint32_t k4_slot;          // which of the two lists a k4_kernel launch takes (5 or 6)
    // rand_read_label mode (null-model generation): per (taxid, GC bucket) the largest k-mer fraction over the reads and
    // the number of reads that hit the taxid; the decision step is skipped
    uint32_t* rand_max;        // [n_ids][rand_nb] float bits, or null
    uint32_t* rand_cnt;        // [n_ids][rand_nb]
    const uint8_t* rand_gc;    // GC bucket of read r - result_base
    uint32_t rand_nb;
    unsigned char* gscratch;   // per-workgroup tables of the global-memory class (reads beyond the LDS classes), or null
    NullModelDev nm;           // -n null models (active == 0: scores are plain k-mer fractions)
    uint32_t gene_mode = 0;    // the database holds 32-bit gene-id lists: gene_label's vote instead of the taxonomic call
    // Tail entries (tail_kernel): the k-mer positions of a read past its last full 64-lane chunk, looked up beforehand with the
    // tails of many reads side by side in a wave.  The 160-k-mer classes take them instead of running a third chunk for a
    // handful of lanes.  tail_lpr = entries per read (4, 8 or 16; 0 = off), indexed like `results`.
    const uint32_t* tail16 = nullptr;  // [reads][tail_lpr] x 16 B: k-mer lo | k-mer hi | bucket | payload, valid << 24, minimizer offset << 30
    const uint64_t* tail_u = nullptr;  // [reads][4]: the scrambled m-mers at positions 128..130 (what the k-mers at 125..127 need)
    uint32_t tail_lpr = 0;
};

// record handed to the K4 kernels: word0 = nT | cand << 16, word1 reserved, then K4T words reg | cnt << 16
static const int kK4T = 64;
static const int kK4RecWords = 2 + kK4T;

// Candidate space (-p): 8 M reads bumping ONE cursor word is 8 M atomics on one address, which the L2 serves one after the
// other (~8 ns each: the fast class took 88 ms instead of 24).  The fast classes therefore allocate through kCandSubs
// sub-cursors -- (next free pair | end of the chunk << 32), one per 64-byte line, picked by the workgroup's number -- each of
// which takes cand_chunk pairs from the bump cursor at a time: the read that crosses a chunk's end brings and installs the next one,
// reads that arrive while it does allocate from the bump cursor directly (what is lost is the crossed chunk's last few pairs).
static const int kCandSubs = 1024;
static const int kCursorWords = 16;   // the counter block proper
static const size_t kCursorBytes = kCursorWords * 4 + (size_t)kCandSubs * 64;

enum { kErrTidOverflow = 1, kErrReadTooLong = 2, kErrCandOverflow = 4, kErrLineageTrunc = 8, kErrNoNullModel = 16 };

// launchers (all asynchronous on `stream`)
void launch_fill_offsets(uint64_t* off, uint64_t* rec_off, uint64_t n, uint32_t len, hipStream_t stream);  // off[i] = i * len, rec_off[i] = i * rec_words(len), i = 0..n
void launch_pack_reads(const uint8_t* bases, const uint64_t* off, const uint64_t* rec_off, uint32_t* words, uint64_t n,
                       hipStream_t stream);
void launch_insert_pairs(const DeviceTables& tb, const uint64_t* kmers, const uint32_t* payload, uint64_t n,
                         uint32_t* fail, hipStream_t stream);
void launch_synth_db(const DeviceTables& tb, const SynthGeo& g, int k, const uint16_t* strain_idx, const uint32_t* list_payload,
                     uint64_t g_off, uint32_t rep, uint32_t rep_stride, uint32_t* fail, unsigned long long* inserted, hipStream_t stream);
void launch_table_count(const DeviceTables& tb, unsigned long long* out, hipStream_t stream);
void launch_synth_reads(uint32_t* words, const uint64_t* rec_off, const uint32_t* lengths, uint32_t n_lengths, uint64_t n,
                        uint64_t seed, const SynthGeo& g, hipStream_t stream);
void launch_lookup(const DeviceTables& tb, const uint64_t* kmers, uint64_t n, uint32_t* counts, uint32_t* tids,
                   uint32_t stride, hipStream_t stream);
void launch_div_check(unsigned long long* out2, hipStream_t stream);
void launch_probe_stats(const DeviceTables& tb, const uint64_t* kmers, uint64_t n, unsigned long long* out, hipStream_t stream);
// fills a.tail16 / a.tail_u for the reads of the launch (a.index / a.first, a.count; not for device-side counts)
void launch_tail(const ClassifyArgs& a, hipStream_t stream);
// tcap_class: 0 = fast (T=64, E=256), 2 = the same with E=512, 3 = middle (T=256, E=1024, reads up to 531 bp), 1 = large (T=1024).
// Returns false if max_len exceeds every U class of the tier.
bool launch_classify(const ClassifyArgs& a, uint32_t max_read_len, int tcap_class, hipStream_t stream);
void launch_k4_begin(const ClassifyArgs& a, hipStream_t stream, hipStream_t stream2, hipStream_t stream3, hipStream_t small_stream,
                     hipEvent_t forked);
void launch_k4_end(const ClassifyArgs& a, hipStream_t join_stream, hipStream_t stream2, hipStream_t stream3, hipStream_t small_stream,
                   hipEvent_t joined2, hipEvent_t joined3, hipEvent_t joined_small, hipEvent_t done);
void launch_k4_debug(const ClassifyArgs& a, const uint32_t* idx, const float* scores, const uint64_t* off, const float* stdevs, uint64_t n,
                     hipStream_t stream);
void launch_k4_debug_counts(const ClassifyArgs& a, const uint32_t* idx, const uint32_t* cnts, const uint64_t* off, const uint32_t* cands, uint64_t n,
                            bool on_the_wave, hipStream_t stream);
int classify_max_read_len();
size_t classify_gmem_scratch_bytes();
// issues ~n_probes random bucket reads (rounded up to 144 per wave x 4096 waves)
void launch_gather_bench(const uint64_t* slots, uint32_t nbuckets, uint64_t n_probes, uint64_t seed,
                         unsigned long long* sink, hipStream_t stream, int bytes_per_probe = 64);


// host-callable copies of the synthetic genome functions (tests / oracle cross-checks)
uint32_t synth_strain_base_host(const SynthGeo& g, uint32_t species, uint32_t strain_global, uint64_t pos);
uint32_t synth_read_windows_host(const SynthGeo& g, const uint32_t* lengths, uint32_t n_lengths, uint64_t seed, uint64_t r, int k,
                                 uint32_t* len_out, uint32_t* strain_global, uint64_t* gpos, uint32_t* rpos, uint32_t cap);
void synth_window_host(const SynthGeo& g, int k, uint32_t species, uint64_t pos, uint64_t* kmer, uint32_t* first_strain, uint32_t* mask, bool* inblk,
                       int* cons_level = nullptr);

}  // namespace lmat
