// lmat_internal.hpp -- context and host-side tables of the engine (not part of the ABI).
#pragma once
#include <hip/hip_runtime_api.h>
#include <map>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>
#include "../../include/lmat_hip.h"
#include "lmat_common.hpp"
#include "dbbuild.hpp"

namespace lmat {

// Taxonomy in the engine's internal index space: index 1..n in ascending 32-bit
// taxid order (so "ascending tid" loops of the reference become ascending index),
// 0 = none.  Covers every taxid the 16-bit map can produce plus all their ancestors.
struct HostTaxonomy {
    uint32_t n = 0;
    std::vector<uint32_t> tid32;      // [n+1]
    std::vector<uint16_t> fdepth;     // [n+1] depth from the -e file (read_label.cpp:1579-1582)
    std::vector<uint8_t> flags;       // [n+1] kFlag*
    std::vector<uint32_t> species_of; // [n+1] for strain-ranked ids: first ancestor ranked "species"
    std::vector<uint32_t> path_off;   // [n+1] into paths
    std::vector<uint16_t> path_len;   // [n+1] number of ancestors (tree depth)
    std::vector<uint32_t> paths;      // ancestors parent..root as internal indices (TaxTree.hpp:60-91)
    std::vector<uint32_t> tin, tout;  // [n+1] Euler-tour interval: a is a proper ancestor of b iff tin[a] < tin[b] && tout[b] <= tout[a]
    std::unordered_map<uint32_t, uint32_t> index_of;  // tid32 -> internal
    std::unordered_map<uint32_t, uint16_t> br;        // 32 -> 16 (make_db_table.cpp:259-273)
    std::vector<uint32_t> conv;                       // [65536] 16 -> 32 (read_label.cpp:1593-1598), 0 = unmapped
    uint32_t human_idx = 0;                           // internal index of 9606
    bool loaded = false;
    // More than 65534 ids (the closure of a database of 32-bit taxids: upstream's TID_SIZE=32 build, CMakeLists.txt:92-105):
    // internal ids and Euler ticks do not fit 16 bits.  The host keeps 32-bit tables either way; the device gets the 16-bit
    // packing of rounds 1-3 for a taxonomy that fits it (the fast classes are built on it) and the WIDE tables otherwise, which
    // the wide classes of the classify kernel read (kernels.hip: WIDE).  A wide database stores its lists as 32-bit taxids.
    bool wide = false;
};

struct DeviceTables {
    uint64_t* slots = nullptr;      // the k-mer table: compact buckets when cpt.nb != 0, else wide (8 x u64 slots per bucket)
    uint32_t nbuckets = 0;
    CptGeom cpt;                    // geometry of the compact layout (lmat_common.hpp)
    uint64_t* ovf_slots = nullptr;  // compact only: k-mers whose bucket was full, wide layout
    uint32_t ovf_nbuckets = 0;
    uint16_t* arena = nullptr;
    uint32_t* tid32 = nullptr;
    uint16_t* fdepth = nullptr;
    uint8_t* flags = nullptr;
    uint16_t* species_of = nullptr;
    uint32_t* path_off = nullptr;
    uint16_t* path_len = nullptr;
    uint16_t* paths = nullptr;
    uint16_t* tin = nullptr;
    uint16_t* tout = nullptr;
    // packed copies for the classify kernels: one load instead of several gathers per id
    uint64_t* paths8 = nullptr;   // parallel to paths: id | fdepth << 16 | tin << 32 | tout << 48
    uint8_t* paths_fl = nullptr;  // parallel to paths: the ancestor's flags (with paths8: everything the decision step wants of an ancestor the closure registers)
    uint32_t* facts16 = nullptr;  // [n+1][4]: path_off | path_len, species_of << 16 | tin, tout << 16 | fdepth, flags << 16
    // wide taxonomies (HostTaxonomy::wide): 32-bit ids and ticks; the 16-bit arrays above stay null
    uint32_t wide = 0;
    uint32_t* paths32 = nullptr;      // parallel to paths
    uint32_t* tin32 = nullptr;        // [n+1]
    uint32_t* tout32 = nullptr;
    uint32_t* species_of32 = nullptr;
    uint32_t* conv = nullptr;  // [65536] 16 -> 32, lookup API only
    uint32_t n_ids = 0;
    int k = 0;
    uint32_t list_shift = 0;   // list records lie on (16 << list_shift)-byte boundaries (lmat_common.hpp)
    uint32_t depth_consistent = 0;  // every node's -e depth is larger than its parent's: "fits the lineage" is then "related to every member"
};

// Null models on the device (loadRandHits, src/read_label.cpp:512-678).  One table per k-mer-count class
// that has a readable file; tables are dense over the internal taxid index.
struct NullModelDev {
    const float* val = nullptr;       // [(table * n_ids + id) * nb_max + bin]
    const uint8_t* cls = nullptr;     // [table * n_ids + id] class id, 0xFF = taxid has no null model
    const int* len_vec = nullptr;     // [n_len] sorted k-mer-count classes, first entry 0 (read_len_vec)
    const int* len_avg = nullptr;     // [n_len - 1] midpoints (read_len_avgs)
    const int* len_table = nullptr;   // [n_len] table index of a class or -1; [n_len] = table of class 80 or -1
    const int* nbins = nullptr;       // [n_tables]
    const uint8_t* cls_rank = nullptr;   // [n_cls] gRank2num of the class string (0 for unknown strings)
    const uint8_t* lower_cls = nullptr;  // [10] class id of gNum2rank[ti]
    int n_len = 0, n_tables = 0, nb_max = 0, n_cls = 0, active = 0;
};

// geometry of the synthetic genomes
struct SynthGeo {
    uint64_t seed = 0, G = 0, blk = 0;   // genome length; the first blk bases are shared by the species of a genus
    uint32_t n_species = 0, S = 0, spg = 1;  // strains per species, species per genus
    // Heavy tail (round 4): behind the genus block, three CONSERVED blocks -- [blk, cend[0]) shared by the csz[0] species of a
    // family, [cend[0], cend[1]) by those of a phylum, [cend[1], cend[2]) by those of a superkingdom, without strain
    // substitutions (think rRNA operons): k-mers whose taxid lists hold every strain, species and inner node of the group
    // (69 / 277 / 1109 taxids in the bench taxonomy).  coff[l]: where level l's list payloads start in the generator's table.
    uint64_t cend[3] = {0, 0, 0};
    uint32_t csz[3] = {1, 1, 1};
    uint64_t coff[3] = {0, 0, 0};
};

struct KernelParams {
    float sdiff, hbias, min_score;
    int min_kmer, min_fnd_kmer, prn_all, screen_phix;
    int permissive;  // -s: no representative-strain / lineage-closure pass (read_label.cpp:1143)
    int stop_after;  // debug/profiling only (env LMAT_STOP_AFTER): 0 = full path, n = return after phase n
    int k4_row;      // LMAT_K4_ROW=1: tables of up to 16 taxids are decided by k4_row_kernel instead of on the classify wave (measured slower)
    int k4_static;   // set by the launchers (kernels.hip k4_static_of): what the choice of decision path asks of the launch, not of the read
};

}  // namespace lmat

namespace lmat {
struct StreamBuild;
// k-mer capacities of the fast classify classes: a read of P k-mer positions runs in the smallest class that holds it
static const int kNCls = 4;
static const uint32_t kClsU[kNCls] = {160, 256, 320, 512};
inline int len_class(uint32_t P) { return P <= 160 ? 0 : (P <= 256 ? 1 : (P <= 320 ? 2 : 3)); }
}

struct lmat_reads {
    uint32_t* words = nullptr;    // device: packed records
    uint64_t* rec_off = nullptr;  // device: [n+1] word offsets
    uint64_t n = 0;
    uint64_t n_words = 0;
    uint32_t max_len = 0;
    uint32_t class_len = 0;       // length that all but the longest 1 % of the reads stay under
    // Length classes of the fast kernel (k-mer capacities 160 / 256 / 320 / 512: lmat::kNCls, lmat::len_class): read indices per class, ascending, so a
    // mixed-length batch runs each read in the smallest class that holds it.  Built on first use for the database's k.
    std::vector<uint32_t> lens;
    std::vector<uint32_t> cls_host[lmat::kNCls];
    uint32_t* cls_dev[lmat::kNCls] = {nullptr, nullptr, nullptr, nullptr};
    int cls_k = 0;
    // a batch slot of a lmat_stream: the class lists are filled by the submitter into buffers the slot owns (cls_dev, cls_n
    // entries each) and always cover the whole batch
    bool preset = false;
    uint64_t cls_n[lmat::kNCls] = {0, 0, 0, 0};
};

struct lmat_ingest {
    lmat::Ingest ing;
};

struct lmat_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    lmat_params params;
    std::string err;
    lmat::HostTaxonomy tax;
    lmat::DeviceTables dev;
    lmat::StreamBuild* sb = nullptr;     // device-side build in progress (lmat_api.cpp)
    int last_rc = 0;                     // code of the last set_err
    lmat::Ingest* ingest = nullptr;      // open between lmat_db_begin / lmat_db_load_image and lmat_db_finalize
    uint64_t ingest_table_bytes = 0;
    uint64_t n_kmers = 0;
    uint64_t arena_words = 0;  // u16 units
    bool db_ready = false;
    // label modes that shape the list records (set before the database is finalized)
    int rt_tid_cut = 0;                                  // -g: run-time pruning threshold (TaxNodeStat.hpp:76)
    std::unordered_map<uint32_t, uint32_t> rt_rank_map;  // -m: taxid -> numeric rank (read_label.cpp:1547-1553)
    int permissive = 0;                                  // -s (gPERMISSIVE_MATCH)
    int rand_mode = 0;                                   // list records as src/rkmer.hpp builds them (no human folding)
    int gene_mode = 0;                                   // gene database (gene_label): lists of 32-bit gene ids, no taxonomy involved
    uint32_t* d_rand_max = nullptr;                      // rand_read_label tables [n_ids][rand_nb]
    uint32_t* d_rand_cnt = nullptr;
    uint8_t* d_rand_gc = nullptr;
    uint64_t rand_gc_cap = 0;
    uint32_t rand_nb = 0;
    bool rand_launch = false;
    // synthetic generator state
    uint32_t synth_branching[6] = {0, 0, 0, 0, 0, 0};
    uint32_t synth_n_species = 0, synth_strains_per_species = 0;
    uint64_t synth_genome_len = 0, synth_seed = 0;
    lmat::SynthGeo synth_geo;
    uint64_t n_lists = 0;
    std::vector<uint16_t> synth_strain_idx;   // [species*S + s] internal index
    std::vector<uint16_t> synth_species_idx;  // [species]
    uint16_t* d_synth_strain_idx = nullptr;
    // results / tallies on device
    lmat_read_result* d_results = nullptr;
    uint64_t results_cap = 0;
    lmat_cand* d_cands = nullptr;
    uint64_t cands_cap = 0;
    uint32_t* d_cursor = nullptr;  // [0] cand cursor, [1] error flags, [2] overflow-list length
    uint32_t* d_ovf = nullptr;     // reads to re-run with the large-capacity kernel
    uint32_t* d_k4buf = nullptr;   // records handed from the fast classify kernel to the K4 kernels
    uint32_t* d_ovf2 = nullptr;    // second overflow list: reads beyond the E=512 class
    uint32_t* d_ovf3 = nullptr;    // third: reads beyond the large LDS class
    uint32_t* d_ovf4 = nullptr;    // reads beyond the middle tier (T = 256): the large LDS class takes them
    unsigned char* d_gscratch = nullptr;  // tables of the global-memory class, allocated on first use
    uint32_t* d_k4small = nullptr; // index lists of the reads awaiting K4 (k4_compact_kernel)
    uint32_t* d_k4large = nullptr;
    uint32_t* d_k4bail = nullptr;
    unsigned char* d_tail = nullptr;  // tail entries of the batch's reads (tail_kernel): [reads][lpr] x 16 B, then [reads][4] x 8 B
    uint64_t tail_bytes = 0;
    hipStream_t stream2 = nullptr;  // the scratch K4 kernel runs beside the LDS one
    hipStream_t stream3 = nullptr;  // ... and the LDS kernel of the largest tables beside both
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join3 = nullptr, ev_join_small = nullptr;
    uint32_t* d_err = nullptr;      // sticky error flags of every launch since the last report
    // Two sets of everything a batch's kernels write besides the tallies (result records, candidates, hand-off records,
    // lists, counters): the decision kernels of one batch run beside the classify kernel of the next.  The members above
    // (d_results ... d_k4bail, d_cursor) are the set in use; the other one is parked here.
    struct BatchSet {
        lmat_read_result* d_results = nullptr; uint64_t results_cap = 0;
        lmat_cand* d_cands = nullptr; uint64_t cands_cap = 0;
        uint32_t *d_cursor = nullptr, *d_ovf = nullptr, *d_ovf2 = nullptr, *d_ovf3 = nullptr, *d_ovf4 = nullptr, *d_k4buf = nullptr, *d_k4small = nullptr,
                 *d_k4large = nullptr, *d_k4bail = nullptr;
        uint64_t ovf_cap = 0;
        unsigned char* d_tail = nullptr; uint64_t tail_bytes = 0;
        hipEvent_t done = nullptr;   // recorded behind the last kernel that touches the set
        bool in_flight = false;
    } parked;
    hipEvent_t ev_done = nullptr;    // `done` of the set in use
    hipStream_t join_stream = nullptr;  // where the last launch's kernels joined (its `done` was recorded there)
    bool set_in_flight = false;
    lmat::NullModelDev nm;         // device pointers owned by the context
    std::vector<void*> nm_allocs;
    uint64_t ovf_cap = 0;
    void* d_counts = nullptr;      // u64 count[n_ids] | f64 score[n_ids] | u64 nomatch[3]
    void* d_counts_bak = nullptr;  // tallies as they were before the current blocking launch (restored when it fails)
    // where the next launch writes instead of the context's own buffers (set by the streamed boundary around its launches)
    lmat_read_result* out_results = nullptr;
    lmat_cand* out_cands = nullptr;
    void* out_counts = nullptr;
    bool batch_err = false;        // the next launch keeps its error flags in its own counter block (word 15) instead of the sticky word
    uint64_t counts_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending_events;   // around the classify kernel
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending_events2;  // around the K4 kernels + the larger-class re-runs
    float kernel_ms_total = 0, kernel2_ms_total = 0, last_classify_ms = 0, last_decide_ms = 0;
    uint64_t last_launches = 0;
    uint64_t kernel_launches = 0;
    // RCCL communicators (collective.cpp; opaque here): comm_ranks = this context as one rank of a multi-process job
    // (lmat_comm_init), comm_local = as one of the contexts of this process (lmat_counts_allreduce), keyed by the device list
    void* comm_ranks = nullptr;
    int comm_n = 0, comm_rank = 0;
    void* comm_local = nullptr;
    std::string comm_local_key;
};

namespace lmat {
int set_err(lmat_ctx* c, int code, const std::string& msg);
int load_taxonomy_files(lmat_ctx* c, const char* tree_fn, const char* depth_fn, const char* rank_fn,
                        const char* idmap_fn, const char* plasmid_fn);
int upload_taxonomy(lmat_ctx* c);
void build_euler_intervals(HostTaxonomy& T);
int load_null_models(lmat_ctx* c, const char* list_fn);
void free_null_models(lmat_ctx* c);
// compute the arena record of one raw list; returns false (err set) on invalid ids
bool build_list_record(lmat_ctx* c, const std::vector<uint16_t>& raw, std::vector<uint16_t>& rec);
// the same for a wide database: raw = the stored 32-bit taxids as (low, high) pairs of u16 (Ingest::raw32); the record holds pairs too
bool build_list_record_wide(lmat_ctx* c, const std::vector<uint16_t>& raw_pairs, std::vector<uint16_t>& rec);
void comm_free(lmat_ctx* c);
}  // namespace lmat
