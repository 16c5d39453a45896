// lmat_api.cpp -- C-ABI entry points (include/lmat_hip.h) over the HIP kernels.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <string>
#include <thread>
#include <atomic>
#include <fcntl.h>
#include <unistd.h>
#include "kernels.hpp"
#include "outfmt.hpp"

using namespace lmat;

#define HIPCHK(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e__ = (call);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return set_err(ctx, LMAT_E_DEVICE, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)

template <class T>
static int dev_upload(lmat_ctx* c, T** dst, const std::vector<T>& src) {
    if (*dst) { hipFree(*dst); *dst = nullptr; }
    size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
    HIPCHK(c, hipMalloc((void**)dst, bytes));
    if (!src.empty()) HIPCHK(c, hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return LMAT_OK;
}

namespace lmat {
int upload_taxonomy(lmat_ctx* c) {
    HostTaxonomy& T = c->tax;
    int rc;
    if ((rc = dev_upload(c, &c->dev.tid32, T.tid32))) return rc;
    if ((rc = dev_upload(c, &c->dev.fdepth, T.fdepth))) return rc;
    if ((rc = dev_upload(c, &c->dev.flags, T.flags))) return rc;
    if ((rc = dev_upload(c, &c->dev.path_off, T.path_off))) return rc;
    if ((rc = dev_upload(c, &c->dev.path_len, T.path_len))) return rc;
    if ((rc = dev_upload(c, &c->dev.conv, T.conv))) return rc;
    c->dev.wide = T.wide ? 1u : 0u;
    if (T.wide) {  // 32-bit ids and ticks as they are; the 16-bit packing does not exist for this taxonomy
        if ((rc = dev_upload(c, &c->dev.species_of32, T.species_of))) return rc;
        if ((rc = dev_upload(c, &c->dev.paths32, T.paths))) return rc;
        if ((rc = dev_upload(c, &c->dev.tin32, T.tin))) return rc;
        if ((rc = dev_upload(c, &c->dev.tout32, T.tout))) return rc;
    } else {
        auto narrow = [](const std::vector<uint32_t>& v) { return std::vector<uint16_t>(v.begin(), v.end()); };
        if ((rc = dev_upload(c, &c->dev.species_of, narrow(T.species_of)))) return rc;
        if ((rc = dev_upload(c, &c->dev.paths, narrow(T.paths)))) return rc;
        if ((rc = dev_upload(c, &c->dev.tin, narrow(T.tin)))) return rc;
        if ((rc = dev_upload(c, &c->dev.tout, narrow(T.tout)))) return rc;
        std::vector<uint64_t> p8(T.paths.size());
        std::vector<uint8_t> pfl(T.paths.size());
        for (size_t i = 0; i < T.paths.size(); ++i) {
            const uint32_t a = T.paths[i];
            p8[i] = (uint64_t)a | ((uint64_t)T.fdepth[a] << 16) | ((uint64_t)T.tin[a] << 32) | ((uint64_t)T.tout[a] << 48);
            pfl[i] = (uint8_t)T.flags[a];
        }
        if ((rc = dev_upload(c, &c->dev.paths_fl, pfl))) return rc;
        std::vector<uint32_t> f16((size_t)(T.n + 1) * 4);
        for (uint32_t i = 0; i <= T.n; ++i) {
            f16[4 * i + 0] = T.path_off[i];
            f16[4 * i + 1] = (uint32_t)T.path_len[i] | ((uint32_t)T.species_of[i] << 16);
            f16[4 * i + 2] = (uint32_t)T.tin[i] | ((uint32_t)T.tout[i] << 16);
            f16[4 * i + 3] = (uint32_t)T.fdepth[i] | ((uint32_t)T.flags[i] << 16);
        }
        if ((rc = dev_upload(c, &c->dev.paths8, p8))) return rc;
        if ((rc = dev_upload(c, &c->dev.facts16, f16))) return rc;
    }
    c->dev.n_ids = T.n + 1;
    c->dev.depth_consistent = 1;
    for (uint32_t i = 1; i <= T.n; ++i)
        if (T.path_len[i] && T.fdepth[i] <= T.fdepth[T.paths[T.path_off[i]]]) { c->dev.depth_consistent = 0; break; }
    // tallies: u64 count[n_ids] | f64 score[n_ids] | u64 nomatch[3]
    if (c->d_counts) { hipFree(c->d_counts); c->d_counts = nullptr; }
    if (c->d_counts_bak) { hipFree(c->d_counts_bak); c->d_counts_bak = nullptr; }
    c->counts_bytes = (uint64_t)c->dev.n_ids * 16 + 24;
    HIPCHK(c, hipMalloc(&c->d_counts, c->counts_bytes));
    HIPCHK(c, hipMemset(c->d_counts, 0, c->counts_bytes));
    return LMAT_OK;
}
}  // namespace lmat

static KernelParams kparams(const lmat_params& p) {
    KernelParams k;
    k.sdiff = p.sdiff; k.hbias = p.hbias; k.min_score = p.min_score;
    k.min_kmer = p.min_kmer; k.min_fnd_kmer = p.min_fnd_kmer; k.prn_all = p.prn_all; k.screen_phix = p.screen_phix;
    k.permissive = 0;
    const char* sa = getenv("LMAT_STOP_AFTER");
    k.stop_after = sa ? atoi(sa) : 0;
    // LMAT_K4_WAVE=0: every read takes the general decision path (hand-off record -> k4 kernels); 20 stops nothing
    if (const char* w = getenv("LMAT_K4_WAVE")) if (!k.stop_after && atoi(w) == 0) k.stop_after = 20;
    // LMAT_K4_ROW=1: tables of up to 16 taxids are decided by k4_row_kernel, four reads to a wave, instead of on the classify wave.
    // Measured (64 GiB table, 2 M reads per launch): that kernel needs 0.95 ms for the 1.5 M such reads of a launch -- its rows turn
    // every wave-uniform value of k4_wave into vector work, ~1400 instructions per wave of four reads, no cheaper per read than the
    // ~390 of k4_wave -- and the step takes 7.5 ms instead of 7.1.  Off by default; kept as the measured alternative.
    k.k4_row = getenv("LMAT_K4_ROW") && atoi(getenv("LMAT_K4_ROW")) != 0;
    k.k4_static = 0;
    return k;
}

extern "C" {

static void sb_free(lmat_ctx* c);
static int sb_begin(lmat_ctx* c, uint64_t n_kmers, uint64_t table_bytes, int k);
static int sb_push(lmat_ctx* c, Ingest& B);

int lmat_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int lmat_ctx_create(int device, const lmat_params* params, lmat_ctx** out) {
    if (!out) return LMAT_E_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return LMAT_E_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return LMAT_E_DEVICE;
    lmat_ctx* c = new lmat_ctx();
    c->device = device;
    lmat_params def = {1.0f, 3.0f, 0.0f, 35, 1, 0, 1};
    c->params = params ? *params : def;
    if (hipStreamCreate(&c->stream) != hipSuccess) { delete c; return LMAT_E_DEVICE; }
    if (hipMalloc((void**)&c->d_cursor, kCursorBytes) != hipSuccess || hipMalloc((void**)&c->parked.d_cursor, kCursorBytes) != hipSuccess ||
        hipMalloc((void**)&c->d_err, 64) != hipSuccess) { delete c; return LMAT_E_DEVICE; }
    hipMemset(c->d_cursor, 0, kCursorBytes);
    hipMemset(c->parked.d_cursor, 0, kCursorBytes);
    hipMemset(c->d_err, 0, 64);
    *out = c;
    return LMAT_OK;
}

void lmat_ctx_destroy(lmat_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    for (auto& e : c->pending_events) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    for (auto& e : c->pending_events2) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    void* ptrs[] = {c->dev.slots, c->dev.ovf_slots, c->dev.arena, c->dev.tid32, c->dev.fdepth, c->dev.flags, c->dev.species_of,
                    c->dev.path_off, c->dev.path_len, c->dev.paths, c->dev.paths8, c->dev.paths_fl, c->dev.facts16, c->dev.conv, c->dev.tin, c->dev.tout, c->dev.paths32, c->dev.tin32, c->dev.tout32, c->dev.species_of32, c->d_results, c->d_cands, c->d_cursor,
                    c->d_counts, c->d_counts_bak, c->d_synth_strain_idx, c->d_ovf, c->d_ovf2, c->d_ovf3, c->d_ovf4, c->parked.d_ovf4, c->d_k4buf, c->d_k4small, c->d_k4large, c->d_k4bail, c->d_tail, c->parked.d_tail, c->d_gscratch, c->d_rand_max, c->d_rand_cnt, c->d_rand_gc,
                    c->d_err, c->parked.d_results, c->parked.d_cands, c->parked.d_cursor, c->parked.d_ovf, c->parked.d_ovf2, c->parked.d_ovf3, c->parked.d_k4buf, c->parked.d_k4small,
                    c->parked.d_k4large, c->parked.d_k4bail};
    for (void* p : ptrs)
        if (p) hipFree(p);
    sb_free(c);
    free_null_models(c);
    comm_free(c);
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->ev_join) hipEventDestroy(c->ev_join);
    if (c->ev_join3) hipEventDestroy(c->ev_join3);
    if (c->ev_done) hipEventDestroy(c->ev_done);
    if (c->ev_join_small) hipEventDestroy(c->ev_join_small);
    if (c->parked.done) hipEventDestroy(c->parked.done);
    if (c->stream3) hipStreamDestroy(c->stream3);
    if (c->stream2) hipStreamDestroy(c->stream2);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c->ingest;
    delete c;
}

const char* lmat_last_error(const lmat_ctx* c) { return c ? c->err.c_str() : "null context"; }

int lmat_set_params(lmat_ctx* c, const lmat_params* p) {
    if (!c || !p) return LMAT_E_ARG;
    c->params = *p;
    return LMAT_OK;
}

int lmat_taxonomy_load_files(lmat_ctx* c, const char* tree_fn, const char* depth_fn, const char* rank_fn,
                             const char* idmap_fn, const char* plasmid_fn) {
    if (!c) return LMAT_E_ARG;
    hipSetDevice(c->device);
    return load_taxonomy_files(c, tree_fn, depth_fn, rank_fn, idmap_fn, plasmid_fn);
}

// ---------------------------------------------------------------------------------- DB
// GPU-free ingest objects (dbbuild.cpp): what make_db_table does up to the in-memory table
int lmat_ingest_create(int k, const char* idmap_fn, lmat_ingest** out) {
    if (!out || k < 1 || k > 20) return LMAT_E_ARG;
    lmat_ingest* g = new lmat_ingest();
    g->ing.k = k;
    if (idmap_fn && *idmap_fn && !g->ing.load_idmap(idmap_fn)) { delete g; *out = nullptr; return LMAT_E_IO; }
    *out = g;
    return LMAT_OK;
}
int lmat_ingest_idmap_from_tree(lmat_ingest* g, const char* tree_fn) {
    if (!g || !tree_fn) return LMAT_E_ARG;
    return g->ing.idmap_from_tree(tree_fn) ? LMAT_OK : LMAT_E_IO;
}
void lmat_ingest_destroy(lmat_ingest* g) { delete g; }
const char* lmat_ingest_error(const lmat_ingest* g) { return g ? g->ing.err.c_str() : "null ingest"; }
int lmat_ingest_set_options(lmat_ingest* g, int tid_cutoff, const char* rank_map_fn, const char* human_kmers_fn,
                            const char* adaptor_kmers_fn, uint32_t adaptor_tid) {
    if (!g) return LMAT_E_ARG;
    return g->ing.set_options(tid_cutoff, rank_map_fn, human_kmers_fn, adaptor_kmers_fn, adaptor_tid) ? LMAT_OK : LMAT_E_IO;
}
int lmat_ingest_add_taxhisto(lmat_ingest* g, const char* fn) {
    if (!g || !fn) return LMAT_E_ARG;
    return g->ing.add_taxhisto(fn) ? LMAT_OK : LMAT_E_IO;
}
int lmat_ingest_save_image(const lmat_ingest* g, const char* fn) {
    if (!g || !fn) return LMAT_E_ARG;
    return g->ing.save_image(fn) ? LMAT_OK : LMAT_E_IO;
}
int lmat_ingest_load_image(const char* fn, lmat_ingest** out) {
    if (!fn || !out) return LMAT_E_ARG;
    lmat_ingest* g = new lmat_ingest();
    if (!g->ing.load_image(fn)) { delete g; *out = nullptr; return LMAT_E_IO; }
    *out = g;
    return LMAT_OK;
}
uint64_t lmat_ingest_size(const lmat_ingest* g) { return g ? g->ing.kmers.size() : 0; }
int lmat_ingest_kmer_length(const lmat_ingest* g) { return g ? g->ing.k : 0; }
int lmat_ingest_lookup(const lmat_ingest* g, uint64_t kmer, uint16_t* tids16, int cap) {
    if (!g) return LMAT_E_ARG;
    std::vector<uint16_t> l;
    if (!g->ing.lookup(kmer, l)) return 0;
    for (int i = 0; i < (int)l.size() && i < cap; ++i) tids16[i] = l[i];
    return (int)l.size();
}

int lmat_genedb_begin(lmat_ctx* c, int k, uint64_t n_kmers_hint, uint64_t table_bytes) {
    if (!c) return LMAT_E_ARG;
    if (c->db_ready || c->ingest) return set_err(c, LMAT_E_ARG, "a context holds one database: use a fresh context for a gene database");
    const int had_gene_mode = c->gene_mode;
    c->gene_mode = 1;  // (lmat_db_begin reads it: no taxonomy is needed in this mode)
    if (!c->d_counts) {  // no taxonomy in this mode: the kernels still expect a (minimal) tally buffer
        c->dev.n_ids = 1;
        c->counts_bytes = 16 + 24;
        HIPCHK(c, hipMalloc(&c->d_counts, c->counts_bytes));
        HIPCHK(c, hipMemset(c->d_counts, 0, c->counts_bytes));
    }
    const int rc = lmat_db_begin(c, k, n_kmers_hint, table_bytes);
    if (rc == LMAT_OK) c->ingest->raw32 = true;
    else c->gene_mode = had_gene_mode;  // a refused call leaves the context as it was
    return rc;
}

int lmat_db_begin(lmat_ctx* c, int k, uint64_t n_kmers_hint, uint64_t table_bytes) {
    if (!c) return LMAT_E_ARG;
    if (!c->tax.loaded && !c->gene_mode) return set_err(c, LMAT_E_ARG, "load the taxonomy before the k-mer database");
    if (k < 1 || k > 20) return set_err(c, LMAT_E_ARG, "k must be in 1..20 (40-bit keys)");
    delete c->ingest;
    c->ingest = new Ingest();
    c->ingest->k = k;
    c->ingest->br = c->tax.br;
    // a wide taxonomy without a 16-bit map: the database holds 32-bit taxids, stored as they come (upstream's TID_SIZE=32 build)
    if (c->tax.wide && c->tax.br.empty() && !c->gene_mode) c->ingest->raw32 = true;
    c->ingest_table_bytes = table_bytes;
    c->db_ready = false;
    sb_free(c);
    if (n_kmers_hint || table_bytes) {
        // the caller sized the table: stream -- every 16 M k-mers go straight into the GPU table and leave the host
        int rc = sb_begin(c, n_kmers_hint, table_bytes, k);
        if (rc) return rc;
        c->ingest->flush_every = 1ull << 24;
        if (const char* e = getenv("LMAT_INGEST_CHUNK")) c->ingest->flush_every = std::max<uint64_t>(1, strtoull(e, nullptr, 10));
        c->ingest->flush = [c](Ingest& B) { return sb_push(c, B) == LMAT_OK; };
    }
    return LMAT_OK;
}

int lmat_db_set_build_options(lmat_ctx* c, int tid_cutoff, const char* rank_map_fn, const char* human_kmers_fn,
                              const char* adaptor_kmers_fn, uint32_t adaptor_tid) {
    if (!c || !c->ingest) return set_err(c, LMAT_E_ARG, "lmat_db_begin first");
    if (c->ingest->raw32 && !c->gene_mode && (tid_cutoff > 0 || (human_kmers_fn && *human_kmers_fn) || (adaptor_kmers_fn && *adaptor_kmers_fn)))
        return set_err(c, LMAT_E_ARG, "build-time pruning and the human / adaptor k-mer feeds work on 16-bit codes: a database of 32-bit taxids under a taxonomy beyond 65534 ids is taken as it is (run-time pruning, -g, applies)");
    if (!c->ingest->set_options(tid_cutoff, rank_map_fn, human_kmers_fn, adaptor_kmers_fn, adaptor_tid))
        return set_err(c, LMAT_E_IO, c->ingest->err);
    return LMAT_OK;
}

int lmat_db_add_taxhisto(lmat_ctx* c, const char* fn) {
    if (!c || !fn) return LMAT_E_ARG;
    if (!c->ingest) return set_err(c, LMAT_E_ARG, "lmat_db_begin first");
    if (!c->ingest->add_taxhisto(fn)) {
        const bool tax = c->ingest->err.compare(0, 3, "bad") == 0;
        return set_err(c, tax ? LMAT_E_TAXONOMY : LMAT_E_IO, c->ingest->err);
    }
    if (c->ingest->stream_failed) return c->last_rc ? c->last_rc : LMAT_E_DEVICE;  // the message is already set
    return LMAT_OK;
}

// ---- LMATIMG2: the database as it lies in HBM ------------------------------------------------------------------------------
// The reference opens its database by mapping it (src/read_label.cpp:1477-1491, make_db_table.cpp:330-343): the file IS the
// in-memory layout, and start-up costs what paging it in costs.  LMATIMG1 (dbbuild.cpp) is the ingest's view -- sorted k-mers
// and canonical lists -- and loading it re-inserts every k-mer.  LMATIMG2 is the device layout itself: a 4 KiB header (geometry
// of the compact table, list alignment, counts, and fingerprints of the taxonomy and the label modes the list records were
// built under), then the bucket array, the overflow table and the list arena, each on a 4 KiB boundary.  Saving streams them
// out of HBM, loading streams them in, through a few pinned buffers each with a stream and a thread of its own (file I/O of
// one chunk beside the copies of the others): a database starts at the rate of the file system or the PCIe link.
}  // extern "C"
namespace {
struct ImgHdr2 {
    char magic[8];            // "LMATIMG2"
    uint32_t version, k;
    uint64_t nb;              // compact buckets (0: wide layout)
    uint32_t W, lowbits, m; int32_t wshift;
    uint32_t nbuckets, ovf_nbuckets, list_shift, n_ids;
    uint64_t n_kmers, arena_words, n_lists;
    uint64_t tax_fingerprint, mode_fingerprint;
    uint64_t off_slots, bytes_slots, off_ovf, bytes_ovf, off_arena, bytes_arena;
};
static_assert(sizeof(ImgHdr2) <= 4096, "header page");
uint64_t fnv(uint64_t h, const void* p, size_t n) {
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 0x100000001B3ull; }
    return h;
}
template <class V> uint64_t fnv_vec(uint64_t h, const V& v) { return v.empty() ? h : fnv(h, v.data(), v.size() * sizeof(v[0])); }
// the list records hold internal taxid indices and were filtered / sorted under the context's label modes: an image only fits
// a context with the same taxonomy and the same modes
uint64_t tax_fingerprint(const lmat_ctx* c) {
    const HostTaxonomy& T = c->tax;
    uint64_t h = 0xCBF29CE484222325ull;
    h = fnv_vec(h, T.tid32); h = fnv_vec(h, T.fdepth); h = fnv_vec(h, T.flags); h = fnv_vec(h, T.paths); h = fnv_vec(h, T.path_len);
    return h;
}
uint64_t mode_fingerprint(const lmat_ctx* c) {
    uint64_t h = 0xCBF29CE484222325ull;
    const int32_t m[5] = {c->permissive, c->rt_tid_cut, c->rand_mode, c->gene_mode, c->tax.wide ? 1 : 0};
    h = fnv(h, m, sizeof m);
    std::vector<std::pair<uint32_t, uint32_t>> rm(c->rt_rank_map.begin(), c->rt_rank_map.end());
    std::sort(rm.begin(), rm.end());
    return fnv_vec(h, rm);
}
const uint64_t kImgChunk = 64ull << 20;
const int kImgThreads = 4;
// moves `bytes` between the device range and the file range in chunks, kImgThreads at a time (to_file: HBM -> file)
bool img_stream(lmat_ctx* c, int fd, uint64_t file_off, char* dev, uint64_t bytes, bool to_file, std::string& err) {
    if (!bytes) return true;
    const uint64_t nchunks = (bytes + kImgChunk - 1) / kImgChunk;
    const int nt = (int)std::min<uint64_t>(kImgThreads, nchunks);
    std::atomic<uint64_t> next{0};
    std::atomic<int> failed{0};
    std::vector<std::string> errs(nt);
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t)
        th.emplace_back([&, t]() {
            hipSetDevice(c->device);
            void* buf = nullptr;
            hipStream_t st = nullptr;
            if (hipHostMalloc(&buf, kImgChunk, hipHostMallocDefault) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
                errs[t] = "out of pinned memory for the image buffers"; failed = 1;
                if (buf) hipHostFree(buf);
                return;
            }
            for (uint64_t j; !failed && (j = next.fetch_add(1)) < nchunks;) {
                const uint64_t o = j * kImgChunk, n = std::min(kImgChunk, bytes - o);
                if (to_file) {
                    if (hipMemcpyAsync(buf, dev + o, n, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { errs[t] = "copy from the device failed"; failed = 1; break; }
                    for (uint64_t w = 0; w < n;) {
                        const ssize_t r = pwrite(fd, (char*)buf + w, n - w, (off_t)(file_off + o + w));
                        if (r <= 0) { errs[t] = "write failed (disk full?)"; failed = 1; break; }
                        w += (uint64_t)r;
                    }
                } else {
                    for (uint64_t w = 0; w < n;) {
                        const ssize_t r = pread(fd, (char*)buf + w, n - w, (off_t)(file_off + o + w));
                        if (r <= 0) { errs[t] = "image truncated"; failed = 1; break; }
                        w += (uint64_t)r;
                    }
                    if (failed) break;
                    if (hipMemcpyAsync(dev + o, buf, n, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { errs[t] = "copy to the device failed"; failed = 1; break; }
                }
            }
            hipStreamDestroy(st);
            hipHostFree(buf);
        });
    for (auto& x : th) x.join();
    for (auto& e : errs) if (!e.empty()) { err = e; return false; }
    return !failed;
}
int save_device_image(lmat_ctx* c, const char* fn) {
    if (c->gene_mode) return set_err(c, LMAT_E_ARG, "device images hold taxonomy databases: a gene database is built from its files");
    hipSetDevice(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const DeviceTables& D = c->dev;
    ImgHdr2 h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, "LMATIMG2", 8);
    h.version = 1; h.k = (uint32_t)D.k;
    h.nb = D.cpt.nb; h.W = D.cpt.W; h.lowbits = (uint32_t)D.cpt.lowbits; h.m = (uint32_t)D.cpt.m; h.wshift = D.cpt.wshift;
    h.nbuckets = D.nbuckets; h.ovf_nbuckets = D.ovf_nbuckets; h.list_shift = D.list_shift; h.n_ids = D.n_ids;
    h.n_kmers = c->n_kmers; h.arena_words = c->arena_words; h.n_lists = c->n_lists;
    h.tax_fingerprint = tax_fingerprint(c); h.mode_fingerprint = mode_fingerprint(c);
    auto up = [](uint64_t x) { return (x + 4095) / 4096 * 4096; };
    h.bytes_slots = (D.cpt.nb ? D.cpt.nb : (uint64_t)D.nbuckets) * 64;
    h.bytes_ovf = (uint64_t)D.ovf_nbuckets * 64;
    h.bytes_arena = c->arena_words * 2;
    h.off_slots = 4096; h.off_ovf = up(h.off_slots + h.bytes_slots); h.off_arena = up(h.off_ovf + h.bytes_ovf);
    const int fd = open(fn, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return set_err(c, LMAT_E_IO, std::string("cannot write ") + fn);
    std::vector<char> page(4096, 0);
    memcpy(page.data(), &h, sizeof h);
    std::string err;
    bool ok = pwrite(fd, page.data(), 4096, 0) == 4096;
    if (!ok) err = "write failed";
    ok = ok && img_stream(c, fd, h.off_slots, (char*)D.slots, h.bytes_slots, true, err) &&
         img_stream(c, fd, h.off_ovf, (char*)D.ovf_slots, h.bytes_ovf, true, err) &&
         img_stream(c, fd, h.off_arena, (char*)D.arena, h.bytes_arena, true, err);
    if (close(fd) != 0 && ok) { ok = false; err = "close failed"; }
    if (!ok) { unlink(fn); return set_err(c, LMAT_E_IO, std::string("saving the device image ") + fn + ": " + err); }
    return LMAT_OK;
}
int load_device_image(lmat_ctx* c, const char* fn) {
    if (c->gene_mode) return set_err(c, LMAT_E_ARG, "device images hold taxonomy databases");
    const int fd = open(fn, O_RDONLY);
    if (fd < 0) return set_err(c, LMAT_E_IO, std::string("cannot read database image ") + fn);
    struct Closer { int fd; ~Closer() { close(fd); } } closer{fd};
    std::vector<char> page(4096, 0);
    ImgHdr2 h;
    if (pread(fd, page.data(), 4096, 0) != 4096) return set_err(c, LMAT_E_IO, std::string("cannot read database image ") + fn);
    memcpy(&h, page.data(), sizeof h);
    if (memcmp(h.magic, "LMATIMG2", 8) != 0 || h.version != 1 || h.k < 1 || h.k > 20) return set_err(c, LMAT_E_IO, "not a device image of this engine version");
    if (h.tax_fingerprint != tax_fingerprint(c) || h.n_ids != c->dev.n_ids)
        return set_err(c, LMAT_E_TAXONOMY, "the image was saved under a different taxonomy (its list records hold that taxonomy's internal ids): load the files it was built with");
    if (h.mode_fingerprint != mode_fingerprint(c))
        return set_err(c, LMAT_E_ARG, "the image was saved under different label modes (-s / -g / -m shape the list records): set the same modes, or build from the tax_histo files");
    const off_t fsz = lseek(fd, 0, SEEK_END);
    if (fsz < 0 || (uint64_t)fsz < h.off_arena + h.bytes_arena || h.bytes_slots % 64 || h.bytes_ovf != (uint64_t)h.ovf_nbuckets * 64 ||
        h.bytes_slots != (h.nb ? h.nb : (uint64_t)h.nbuckets) * 64 || h.bytes_arena != h.arena_words * 2)
        return set_err(c, LMAT_E_IO, "database image truncated or inconsistent");
    hipSetDevice(c->device);
    DeviceTables& D = c->dev;
    if (D.slots) { hipFree(D.slots); D.slots = nullptr; }
    if (D.ovf_slots) { hipFree(D.ovf_slots); D.ovf_slots = nullptr; }
    if (D.arena) { hipFree(D.arena); D.arena = nullptr; }
    c->db_ready = false;
    HIPCHK(c, hipMalloc((void**)&D.slots, std::max<uint64_t>(h.bytes_slots, 64)));
    if (h.bytes_ovf) HIPCHK(c, hipMalloc((void**)&D.ovf_slots, h.bytes_ovf));
    HIPCHK(c, hipMalloc((void**)&D.arena, h.bytes_arena + 64));   // (a record's first 64 bytes are read whatever its length: slack behind the last)
    HIPCHK(c, hipMemset((char*)D.arena + h.bytes_arena, 0, 64));
    std::string err;
    if (!img_stream(c, fd, h.off_slots, (char*)D.slots, h.bytes_slots, false, err) ||
        !img_stream(c, fd, h.off_ovf, (char*)D.ovf_slots, h.bytes_ovf, false, err) ||
        !img_stream(c, fd, h.off_arena, (char*)D.arena, h.bytes_arena, false, err))
        return set_err(c, LMAT_E_IO, std::string("loading the device image ") + fn + ": " + err);
    D.cpt = CptGeom();
    D.cpt.nb = h.nb; D.cpt.W = h.W; D.cpt.invW = h.W ? 1.0 / (double)h.W : 0.0; D.cpt.k = (int)h.k; D.cpt.m = (int)h.m; D.cpt.lowbits = (int)h.lowbits; D.cpt.wshift = h.wshift;
    if (!h.nb) D.cpt = CptGeom();
    D.nbuckets = h.nbuckets; D.ovf_nbuckets = h.ovf_nbuckets; D.list_shift = h.list_shift; D.k = (int)h.k;
    c->n_kmers = h.n_kmers; c->arena_words = h.arena_words; c->n_lists = h.n_lists;
    c->db_ready = true;
    return LMAT_OK;
}
}  // namespace
extern "C" {

int lmat_db_save_image(lmat_ctx* c, const char* fn) {
    if (!c || !fn) return LMAT_E_ARG;
    if (!c->ingest && c->db_ready) return save_device_image(c, fn);   // a finalized database: the device layout itself (LMATIMG2)
    if (!c->ingest) return set_err(c, LMAT_E_ARG, "no database: save the ingest's image between lmat_db_begin and lmat_db_finalize, the device image after lmat_db_finalize");
    if (c->ingest->flush) return set_err(c, LMAT_E_ARG, "a streamed build (n_kmers_hint / table_bytes given) keeps no copy to save: use make_db_image");
    return c->ingest->save_image(fn) ? LMAT_OK : set_err(c, LMAT_E_IO, std::string("cannot write ") + fn);
}

int lmat_db_load_image(lmat_ctx* c, const char* fn, uint64_t table_bytes) {
    if (!c || !fn) return LMAT_E_ARG;
    if (!c->tax.loaded) return set_err(c, LMAT_E_ARG, "load the taxonomy before the k-mer database");
    {   // a device image (LMATIMG2) is copied in as it is; lmat_db_finalize then has nothing left to do
        char magic[8] = {0};
        FILE* f = fopen(fn, "rb");
        const bool img2 = f && fread(magic, 8, 1, f) == 1 && memcmp(magic, "LMATIMG2", 8) == 0;
        if (f) fclose(f);
        if (img2) {
            delete c->ingest;
            c->ingest = nullptr;
            sb_free(c);
            return load_device_image(c, fn);
        }
    }
    delete c->ingest;
    c->ingest = new Ingest();
    c->ingest_table_bytes = table_bytes;
    c->db_ready = false;
    sb_free(c);
    {   // header first: the table is sized from the image's k-mer count, then the arrays stream through in chunks
        FILE* f = fopen(fn, "rb");
        char magic[8];
        uint32_t kk = 0;
        uint64_t n = 0;
        const bool ok = f && fread(magic, 8, 1, f) == 1 && memcmp(magic, "LMATIMG1", 8) == 0 && fread(&kk, 4, 1, f) == 1 && fread(&n, 8, 1, f) == 1;
        if (f) fclose(f);
        if (!ok) return set_err(c, LMAT_E_IO, std::string("cannot read database image ") + fn);
        if (kk < 1 || kk > 20) return set_err(c, LMAT_E_IO, "image holds an unsupported k-mer length");
        int rc = sb_begin(c, n, table_bytes, (int)kk);
        if (rc) return rc;
    }
    c->ingest->flush_every = 1ull << 24;
    if (const char* e = getenv("LMAT_INGEST_CHUNK")) c->ingest->flush_every = std::max<uint64_t>(1, strtoull(e, nullptr, 10));
    c->ingest->flush = [c](Ingest& B) { return sb_push(c, B) == LMAT_OK; };
    if (!c->ingest->load_image_streaming(fn, nullptr)) {
        sb_free(c);
        return c->ingest->err.empty() ? (c->last_rc ? c->last_rc : LMAT_E_DEVICE) : set_err(c, LMAT_E_IO, c->ingest->err);
    }
    return LMAT_OK;
}

// Share of the k-mers that do not fit their 12-slot bucket.  Two things drive it:
//  * the average number of k-mers per bucket (`load`).  K-mers arrive in minimizer groups (~2.5 per genome, several times that
//    where many strains share a region), so the tail is far heavier than Poisson: scripts/minimizer_sim.py gives 6.1 % at 6.4
//    for three strains per species;
//  * how often the table reuses a minimizer VALUE for unrelated k-mers: r = k-mers per canonical m-mer (4^m / 2 of them:
//    8.6 G for k = 20).  The minimizer is the smallest of four m-mers, so the low-ranked values are chosen four times as often
//    as the average one, and once r passes ~1 the buckets that hold them fill up whatever the mean load says: a scaled model
//    (k = 14, same scramblers) gives 12 % displaced at load 4.4, r = 2.2 against 3.4 % from the load alone, and the term
//    0.019 r^1.8 fits it from r = 0.5 to 3.  This was the build that "did not finish" in round 2: 20 G 20-mers (r = 2.3) into
//    2^32 buckets with an overflow table sized for 7 % of them met ~12 %, the overflow table filled up, and every further insert
//    crawled along a 16 384-bucket probe chain before giving up.  (2^32 buckets as such are fine: 6 G k-mers build into them in
//    5 s.)  Builds now also stop at the first failed insert instead of crawling on.
// The overflow table is sized for the estimate times a safety factor (2, or 1.4 where the share is large anyway) at a load of
// 0.6, and the build refuses a table that ends up more than 85 % full.
static double cpt_displaced_share(double load, double kmers_per_mmer) {
    return std::min(0.8, 0.061 * std::pow(load / 6.4, 2.5) + 0.01 + 0.019 * std::pow(kmers_per_mmer, 1.8));
}

static int alloc_table(lmat_ctx* c, uint64_t n_kmers, uint64_t table_bytes, int k) {
    if (c->dev.slots) { hipFree(c->dev.slots); c->dev.slots = nullptr; }
    if (c->dev.ovf_slots) { hipFree(c->dev.ovf_slots); c->dev.ovf_slots = nullptr; }
    c->dev.cpt = CptGeom();
    c->dev.ovf_nbuckets = 0;
    const char* fmt = getenv("LMAT_TABLE_FORMAT");  // "wide": the 8 x 64-bit-slot layout for every table (A/B runs)
    if (!(fmt && !strcmp(fmt, "wide"))) {
        // compact layout, ~10 bytes per k-mer: 6.4 k-mers on average in a 12-slot bucket
        uint64_t want = table_bytes ? table_bytes / 64 : (uint64_t)((double)n_kmers / 6.4) + 1;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = ~(size_t)0;
        for (int attempt = 0; attempt < 24; ++attempt) {  // step down while table + overflow exceed the free memory
            static const bool frac_ok = !getenv("LMAT_FRAC_W") || atoi(getenv("LMAT_FRAC_W")) != 0;  // (0: integer widths only, as before round 4 -- A/B runs)
            CptGeom g = cpt_geometry(k, want, frac_ok);
            if (!g.nb) break;
            const double n_est = n_kmers ? (double)n_kmers : 6.4 * (double)g.nb;
            const double load = n_est / (double)g.nb;
            if (load > 10.0) return set_err(c, LMAT_E_CAPACITY, "table_bytes (or the free device memory) too small for the number of k-mers");
            const double s0 = cpt_displaced_share(load, n_est / (0.5 * std::pow(4.0, g.m)));
            double share = s0 * (s0 > 0.15 ? 1.4 : 2.0);
            if (const char* e = getenv("LMAT_OVERFLOW_SHARE")) share = atof(e);
            const uint64_t onb = (uint64_t)(n_est * share / (0.6 * kSlotsPerBucket)) + 1024;
            if (onb > 0xFFFFFFFFull) return set_err(c, LMAT_E_CAPACITY, "overflow table above 2^32 buckets");
            if ((double)(g.nb + onb) * 64.0 + 6.0 * (double)(1ull << 30) > (double)free_b && (g.wshift == -2 || g.W < (127u >> g.lowbits))) {
                want = g.wshift == -2 ? g.nb - g.nb / 16   // fractional widths: any size goes -- a sixteenth less
                                      : (1ull << (2 * g.m - g.lowbits)) / (g.W + 1);  // the next smaller table
                if (want < 1) break;
                continue;
            }
            HIPCHK(c, hipMalloc((void**)&c->dev.slots, g.nb * 64));
            HIPCHK(c, hipMalloc((void**)&c->dev.ovf_slots, onb * 64));
            HIPCHK(c, hipMemsetAsync(c->dev.slots, 0, g.nb * 64, c->stream));
            HIPCHK(c, hipMemsetAsync(c->dev.ovf_slots, 0, onb * 64, c->stream));
            c->dev.nbuckets = (uint32_t)std::min<uint64_t>(g.nb, 0xFFFFFFFFull);  // (the gather microbenchmark's range)
            c->dev.cpt = g;
            c->dev.ovf_nbuckets = (uint32_t)onb;
            if (getenv("LMAT_DEBUG"))
                fprintf(stderr, "[lmat] compact table: %llu buckets (W=%u, %.1f GiB), expected load %.2f, overflow table %llu buckets (%.1f GiB)\n",
                        (unsigned long long)g.nb, g.W, g.nb * 64.0 / (1ull << 30), load, (unsigned long long)onb, onb * 64.0 / (1ull << 30));
            return LMAT_OK;
        }
    }
    uint64_t nb;
    if (table_bytes) nb = table_bytes / 64;
    else nb = (uint64_t)((double)n_kmers / (0.8 * kSlotsPerBucket)) + 1;
    if (nb < 16) nb = 16;
    if (nb > 0xFFFFFFFFull) return set_err(c, LMAT_E_CAPACITY, "hash table above 2^32 buckets");
    if (nb * kSlotsPerBucket < n_kmers + n_kmers / 64)
        return set_err(c, LMAT_E_CAPACITY, "table_bytes too small for the number of k-mers");
    HIPCHK(c, hipMalloc((void**)&c->dev.slots, nb * 64));
    HIPCHK(c, hipMemsetAsync(c->dev.slots, 0, nb * 64, c->stream));
    c->dev.nbuckets = (uint32_t)nb;
    return LMAT_OK;
}

// canonical payloads (16-bit DB id / list number) -> device payloads (internal taxid index / arena offset), chunk by
// chunk: list records are built the first time a list is seen, so a database streams through without the whole
// (k-mer, payload) array ever being resident on the host.
// appends a list record on the next 16-byte boundary of the arena; returns its payload
static uint32_t arena_append(std::vector<uint16_t>& arena, const std::vector<uint16_t>& rec, uint32_t shift) {
    const size_t unit = (size_t)kListUnit << shift;
    arena.resize((arena.size() + unit - 1) / unit * unit, 0);
    const uint32_t pay = kListBase + (uint32_t)(arena.size() / unit);
    arena.insert(arena.end(), rec.begin(), rec.end());
    return pay;
}
// The alignment of a streamed build's list records has to be fixed before the first payload goes into the table.  16 bytes
// (256 MB of records) unless the database is large: from the k-mer count the caller announced -- the distinct lists of the
// reference's 64 GB-class databases stay well below 256 MB, a 460 GB one is not known to -- or LMAT_LIST_SHIFT=0..4.
static uint32_t pick_list_shift(uint64_t n_kmers_hint) {
    if (const char* e = getenv("LMAT_LIST_SHIFT")) return (uint32_t)std::min(std::max(atoi(e), 0), kListShiftMax);
    return n_kmers_hint > 32000000000ull ? 3u : (n_kmers_hint > 8000000000ull ? 2u : 0u);
}
namespace lmat {
struct StreamBuild {
    uint32_t shift = 0;                 // alignment of the list records (pick_list_shift)
    std::vector<uint16_t> arena;        // record 0 reserved (sb_begin)
    std::vector<uint32_t> list_pay;     // per canonical list: device payload (0 = not built yet)
    std::vector<uint32_t> single_pay = std::vector<uint32_t>(65536, 0);
    std::vector<uint32_t> pay;
    uint64_t* d_k = nullptr;
    uint32_t* d_p = nullptr;
    uint32_t* d_fail = nullptr;
    uint64_t cap = 0, inserted = 0;
};
}
static void sb_free(lmat_ctx* c) {
    if (!c->sb) return;
    if (c->sb->d_k) hipFree(c->sb->d_k);
    if (c->sb->d_p) hipFree(c->sb->d_p);
    if (c->sb->d_fail) hipFree(c->sb->d_fail);
    delete c->sb;
    c->sb = nullptr;
}
static int sb_begin(lmat_ctx* c, uint64_t n_kmers, uint64_t table_bytes, int k) {
    hipSetDevice(c->device);
    sb_free(c);
    c->sb = new StreamBuild();
    c->sb->shift = pick_list_shift(n_kmers);
    c->sb->arena.assign((size_t)kListUnit << c->sb->shift, 0);
    int rc = alloc_table(c, n_kmers, table_bytes, k);
    if (rc) return rc;
    c->dev.k = k;
    c->sb->cap = 1ull << 24;
    HIPCHK(c, hipMalloc((void**)&c->sb->d_k, c->sb->cap * 8));
    HIPCHK(c, hipMalloc((void**)&c->sb->d_p, c->sb->cap * 4));
    HIPCHK(c, hipMalloc((void**)&c->sb->d_fail, 4));
    HIPCHK(c, hipMemsetAsync(c->sb->d_fail, 0, 4, c->stream));
    return LMAT_OK;
}
// inserts B.kmers / B.payload (all of them) and clears both
static int sb_push(lmat_ctx* c, Ingest& B) {
    StreamBuild& S = *c->sb;
    const HostTaxonomy& T = c->tax;
    std::vector<uint16_t> rec, one(1);
    if (S.list_pay.size() < B.lists.size()) S.list_pay.resize(B.lists.size(), 0);
    const uint64_t n = B.kmers.size();
    for (uint64_t s = 0; s < n; s += S.cap) {
        const uint64_t m = std::min(S.cap, n - s);
        S.pay.resize(m);
        for (uint64_t i = 0; i < m; ++i) {
            const uint32_t p = B.payload[s + i];
            uint32_t dp;
            if (p >= kListBase) {
                uint32_t& lp = S.list_pay[p - kListBase];
                if (!lp) {
                    if (c->gene_mode) {  // [0x8000][n][0][id low, id high]...: the ids as stored, nothing derived from a taxonomy
                        const std::vector<uint16_t>& l = B.lists[p - kListBase];
                        if (l.size() / 2 > 0xFFFF) return set_err(c, LMAT_E_CAPACITY, "gene list above 65535 ids");
                        rec.assign(kListHdr, 0);
                        rec[0] = 0x8000;
                        rec[1] = (uint16_t)(l.size() / 2);
                        rec.insert(rec.end(), l.begin(), l.end());
                    } else if (!(B.raw32 ? build_list_record_wide(c, B.lists[p - kListBase], rec) : build_list_record(c, B.lists[p - kListBase], rec))) return LMAT_E_TAXONOMY;
                    lp = arena_append(S.arena, rec, S.shift);
                }
                dp = lp;
            } else if (c->gene_mode) {
                return set_err(c, LMAT_E_IO, "gene database record without a list");
            } else {
                uint32_t& sp = S.single_pay[p];
                if (!sp) {
                    // one-element lists: a plain taxid index unless the id needs per-list treatment (unmapped, human
                    // variants folded to 9606, ignored ids): those get a one-element list record
                    // (-s: even a lone taxid brings its lineage into the position set, so it needs a record too)
                    const uint32_t t32 = T.conv[p];
                    const bool special = c->permissive || t32 == 0 || t32 == 63221 || t32 == 741158 || t32 == 20999999 || t32 == 12721 || t32 == 693660;
                    auto it = T.index_of.find(t32);
                    if (!special && it != T.index_of.end() && it->second < kListBase) {   // (a plain payload is an internal index below 65536)
                        sp = it->second;
                    } else {
                        one[0] = (uint16_t)p;
                        if (!build_list_record(c, one, rec)) return LMAT_E_TAXONOMY;
                        sp = arena_append(S.arena, rec, S.shift);
                    }
                }
                dp = sp;
            }
            S.pay[i] = dp;
        }
        if (S.arena.size() / ((size_t)kListUnit << S.shift) + kListBase > kPayloadMask)
            return set_err(c, LMAT_E_CAPACITY, "the taxid-list records exceed " + std::to_string(256u << S.shift) +
                                               " MB, the range of the 24-bit payloads at this record alignment: set LMAT_LIST_SHIFT=" +
                                               std::to_string(std::min<int>(S.shift + 1, kListShiftMax)) + " (records on " +
                                               std::to_string(32u << S.shift) + "-byte boundaries) and build again");
        HIPCHK(c, hipMemcpyAsync(S.d_k, B.kmers.data() + s, m * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(S.d_p, S.pay.data(), m * 4, hipMemcpyHostToDevice, c->stream));
        launch_insert_pairs(c->dev, S.d_k, S.d_p, m, S.d_fail, c->stream);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        S.inserted += m;
    }
    B.flushed += n;
    B.kmers.clear();
    B.payload.clear();
    return LMAT_OK;
}
// an overflow table that is nearly full answers every miss with a probe chain of thousands of buckets: refuse it
static int check_overflow_fill(lmat_ctx* c, uint64_t entries) {
    const double fill = c->dev.ovf_nbuckets ? (double)entries / ((double)c->dev.ovf_nbuckets * kSlotsPerBucket) : 0.0;
    if (getenv("LMAT_DEBUG")) fprintf(stderr, "[lmat] overflow table: %llu k-mers, %.1f %% full\n", (unsigned long long)entries, 100.0 * fill);
    if (fill > 0.85)
        return set_err(c, LMAT_E_CAPACITY, "the overflow table of the compact layout is " + std::to_string((int)(100 * fill)) +
                                           " % full: give the table more room (table_bytes) or raise LMAT_OVERFLOW_SHARE");
    return LMAT_OK;
}
static int sb_finish(lmat_ctx* c) {
    StreamBuild& S = *c->sb;
    int rc;
    const size_t arena_words = S.arena.size();
    S.arena.resize(arena_words + 32, 0);  // the fast classes copy a record's first 64 bytes into LDS whatever its length (kernels.hip, K3b stage 1): 64 bytes of slack behind the last record, as every other producer of an arena leaves
    if ((rc = dev_upload(c, &c->dev.arena, S.arena))) return rc;
    c->dev.list_shift = S.shift;
    c->arena_words = arena_words;
    uint32_t fail = 0;
    HIPCHK(c, hipMemcpy(&fail, S.d_fail, 4, hipMemcpyDeviceToHost));
    uint64_t n = S.inserted;
    sb_free(c);
    if (fail) return set_err(c, LMAT_E_CAPACITY, "hash table (or its overflow table) full during insert (n_kmers_hint / table_bytes too small; LMAT_OVERFLOW_SHARE raises the overflow table's share)");
    if (c->dev.cpt.nb) {  // tidy the compact buckets (slot counts; a key fed twice keeps its smaller payload) and count
        unsigned long long* d_n = nullptr;
        unsigned long long cnt[2] = {0, 0};
        HIPCHK(c, hipMalloc((void**)&d_n, 16));
        HIPCHK(c, hipMemsetAsync(d_n, 0, 16, c->stream));
        launch_table_count(c->dev, d_n, c->stream);
        HIPCHK(c, hipMemcpyAsync(cnt, d_n, 16, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        hipFree(d_n);
        n = cnt[0];
        int rc2 = check_overflow_fill(c, cnt[1]);
        if (rc2) return rc2;
    }
    c->n_kmers = n;
    c->db_ready = true;
    return LMAT_OK;
}

static int build_device_db(lmat_ctx* c, Ingest& B, uint64_t table_bytes) {
    int rc = sb_begin(c, B.kmers.size(), table_bytes, B.k);
    if (rc) return rc;
    std::vector<uint64_t> keep_k;
    std::vector<uint32_t> keep_p;
    keep_k.swap(B.kmers);   // sb_push consumes the arrays; a caller-owned ingest keeps its contents
    keep_p.swap(B.payload);
    const uint64_t n = keep_k.size(), chunk = 1ull << 24;
    for (uint64_t s = 0; s < n && !rc; s += chunk) {
        const uint64_t m = std::min(chunk, n - s);
        B.kmers.assign(keep_k.begin() + s, keep_k.begin() + s + m);
        B.payload.assign(keep_p.begin() + s, keep_p.begin() + s + m);
        rc = sb_push(c, B);
    }
    B.kmers.swap(keep_k);
    B.payload.swap(keep_p);
    B.flushed = 0;
    if (rc) { sb_free(c); return rc; }
    return sb_finish(c);
}

int lmat_db_finalize(lmat_ctx* c) {
    if (!c) return LMAT_E_ARG;
    if (!c->ingest && c->db_ready) return LMAT_OK;   // a device image came in ready
    if (!c->ingest) return set_err(c, LMAT_E_ARG, "lmat_db_begin first");
    int rc;
    if (c->ingest->flush) {  // streamed build: the tail, then the list arena
        rc = c->ingest->stream_failed ? (c->last_rc ? c->last_rc : LMAT_E_DEVICE) : sb_push(c, *c->ingest);
        if (!rc) rc = sb_finish(c);
        else sb_free(c);
    } else {
        rc = build_device_db(c, *c->ingest, c->ingest_table_bytes);
    }
    delete c->ingest;
    c->ingest = nullptr;
    return rc;
}

int lmat_db_from_ingest(lmat_ctx* c, lmat_ingest* g, uint64_t table_bytes) {
    if (!c || !g) return LMAT_E_ARG;
    if (!c->tax.loaded) return set_err(c, LMAT_E_ARG, "load the taxonomy before the k-mer database");
    c->db_ready = false;
    return build_device_db(c, g->ing, table_bytes);
}

// A replica of a finalized database on another context: table, overflow table and arena copied device to device.
int lmat_db_clone(lmat_ctx* d, lmat_ctx* s) {
    if (!d || !s || d == s) return LMAT_E_ARG;
    if (!s->db_ready) return set_err(d, LMAT_E_ARG, "the source context holds no finalized database");
    if (d->db_ready || d->ingest) return set_err(d, LMAT_E_ARG, "the destination already holds a database");
    if (s->gene_mode)  // (a gene context is set up by lmat_genedb_begin, which opens an ingest: there is no empty gene context to clone into)
        return set_err(d, LMAT_E_ARG, "lmat_db_clone replicates taxonomy databases only: build a gene database in each context");
    if (s->gene_mode != d->gene_mode || (!s->gene_mode && (!d->tax.loaded || d->tax.n != s->tax.n || d->tax.tid32 != s->tax.tid32)))
        return set_err(d, LMAT_E_TAXONOMY, "load the same taxonomy into the destination first");
    if (d->permissive != s->permissive || d->rt_tid_cut != s->rt_tid_cut || d->rand_mode != s->rand_mode)
        return set_err(d, LMAT_E_ARG, "the label modes of the two contexts differ (they shape the list records)");
    hipSetDevice(s->device);
    HIPCHK(s, hipStreamSynchronize(s->stream));
    if (d->device != s->device) {  // direct copies over xGMI when the link allows it; staged by the runtime otherwise
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, d->device, s->device) == hipSuccess && can) {
            hipSetDevice(d->device);
            hipError_t e = hipDeviceEnablePeerAccess(s->device, 0);
            if (e != hipSuccess) (void)hipGetLastError();  // already enabled, or not allowed: the copy still works
        }
    }
    hipSetDevice(d->device);
    DeviceTables& D = d->dev;
    const DeviceTables& S = s->dev;
    if (D.slots) { hipFree(D.slots); D.slots = nullptr; }
    if (D.ovf_slots) { hipFree(D.ovf_slots); D.ovf_slots = nullptr; }
    if (D.arena) { hipFree(D.arena); D.arena = nullptr; }
    const uint64_t nb = S.cpt.nb ? S.cpt.nb : (uint64_t)S.nbuckets;
    const uint64_t arena_bytes = s->arena_words * 2 + 64;  // (every builder leaves 64 bytes of slack behind the records: a record's first 64 bytes are read whatever its length)
    HIPCHK(d, hipMalloc((void**)&D.slots, nb * 64));
    HIPCHK(d, hipMemcpyPeerAsync(D.slots, d->device, S.slots, s->device, nb * 64, d->stream));
    if (S.ovf_slots) {
        HIPCHK(d, hipMalloc((void**)&D.ovf_slots, (uint64_t)S.ovf_nbuckets * 64));
        HIPCHK(d, hipMemcpyPeerAsync(D.ovf_slots, d->device, S.ovf_slots, s->device, (uint64_t)S.ovf_nbuckets * 64, d->stream));
    }
    HIPCHK(d, hipMalloc((void**)&D.arena, arena_bytes));
    HIPCHK(d, hipMemsetAsync(D.arena, 0, arena_bytes, d->stream));
    HIPCHK(d, hipMemcpyPeerAsync(D.arena, d->device, S.arena, s->device, s->arena_words * 2, d->stream));
    HIPCHK(d, hipStreamSynchronize(d->stream));
    D.nbuckets = S.nbuckets; D.cpt = S.cpt; D.ovf_nbuckets = S.ovf_nbuckets; D.k = S.k; D.list_shift = S.list_shift;
    d->n_kmers = s->n_kmers; d->arena_words = s->arena_words; d->n_lists = s->n_lists;
    d->db_ready = true;
    return LMAT_OK;
}

int lmat_db_kmer_length(const lmat_ctx* c) { return c ? c->dev.k : 0; }
uint64_t lmat_db_size(const lmat_ctx* c) { return c ? c->n_kmers : 0; }
uint64_t lmat_db_list_count(const lmat_ctx* c) { return c ? c->n_lists : 0; }
uint64_t lmat_db_arena_bytes(const lmat_ctx* c) { return c ? c->arena_words * 2 : 0; }
uint64_t lmat_db_table_bytes(const lmat_ctx* c) {
    return c ? ((c->dev.cpt.nb ? c->dev.cpt.nb : (uint64_t)c->dev.nbuckets) + c->dev.ovf_nbuckets) * 64 : 0;
}

int lmat_db_lookup(lmat_ctx* c, const uint64_t* kmers, uint64_t n, uint32_t* counts, uint32_t* tids, uint32_t stride) {
    if (!c || !c->db_ready) return set_err(c, LMAT_E_ARG, "database not ready");
    if (!n) return LMAT_OK;
    hipSetDevice(c->device);
    uint64_t* d_k = nullptr;
    uint32_t *d_c = nullptr, *d_t = nullptr;
    HIPCHK(c, hipMalloc((void**)&d_k, n * 8));
    HIPCHK(c, hipMalloc((void**)&d_c, n * 4));
    HIPCHK(c, hipMalloc((void**)&d_t, std::max<uint64_t>(n * stride, 1) * 4));
    HIPCHK(c, hipMemsetAsync(d_t, 0, std::max<uint64_t>(n * stride, 1) * 4, c->stream));  // entries past a k-mer's count read as 0
    HIPCHK(c, hipMemcpyAsync(d_k, kmers, n * 8, hipMemcpyHostToDevice, c->stream));
    launch_lookup(c->dev, d_k, n, d_c, d_t, stride, c->stream);
    HIPCHK(c, hipMemcpyAsync(counts, d_c, n * 4, hipMemcpyDeviceToHost, c->stream));
    if (stride && tids) HIPCHK(c, hipMemcpyAsync(tids, d_t, n * stride * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    hipFree(d_k); hipFree(d_c); hipFree(d_t);
    return LMAT_OK;
}

// Measurement hook: where the lookups of these k-mers end -- out[0] home bucket, out[1] absent without a second request,
// out[2] found in the overflow table, out[3] absent after asking it too, out[4] overflow buckets read.  Compact layout only.
int lmat_debug_div_check(lmat_ctx* c, uint64_t* out4) {
    if (!c || !out4) return LMAT_E_ARG;
    hipSetDevice(c->device);
    unsigned long long* d_o = nullptr;
    HIPCHK(c, hipMalloc((void**)&d_o, 32));
    HIPCHK(c, hipMemsetAsync(d_o, 0, 32, c->stream));
    launch_div_check(d_o, c->stream);
    HIPCHK(c, hipMemcpyAsync(out4, d_o, 32, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    hipFree(d_o);
    return LMAT_OK;
}

int lmat_debug_probe_stats(lmat_ctx* c, const uint64_t* kmers, uint64_t n, uint64_t* out5) {
    if (!c || !out5 || (n && !kmers)) return LMAT_E_ARG;
    if (!c->db_ready) return set_err(c, LMAT_E_ARG, "database not ready");
    hipSetDevice(c->device);
    for (int i = 0; i < 5; ++i) out5[i] = 0;
    if (!n || !c->dev.cpt.nb) return LMAT_OK;
    uint64_t* d_k = nullptr;
    unsigned long long* d_o = nullptr;
    HIPCHK(c, hipMalloc((void**)&d_k, n * 8));
    HIPCHK(c, hipMalloc((void**)&d_o, 40));
    HIPCHK(c, hipMemsetAsync(d_o, 0, 40, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_k, kmers, n * 8, hipMemcpyHostToDevice, c->stream));
    launch_probe_stats(c->dev, d_k, n, d_o, c->stream);
    HIPCHK(c, hipMemcpyAsync(out5, d_o, 40, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    hipFree(d_k); hipFree(d_o);
    return LMAT_OK;
}

// ---------------------------------------------------------------------------------- synthetic
// Taxonomy of SURVEY 8(d): root(1) -> b0 superkingdoms -> .. -> species -> strains, 32-bit ids
// 1000 + 7*dense, 16-bit map = ascending rank from 2.  Built in memory (no files).
int lmat_synth_taxonomy(lmat_ctx* c, const uint32_t* br) {
    if (!c || !br) return LMAT_E_ARG;
    hipSetDevice(c->device);
    static const char* ranks[] = {"no_rank", "superkingdom", "phylum", "family", "genus", "species", "strain"};
    (void)ranks;
    HostTaxonomy& T = c->tax;
    T = HostTaxonomy();
    std::vector<uint32_t> ids{1}, parent{0}, depth{0}, level_of{0};
    std::vector<uint32_t> level{0};  // indices into ids
    uint32_t dense = 0;
    for (int lv = 0; lv < 6; ++lv) {
        std::vector<uint32_t> nxt;
        for (uint32_t p : level)
            for (uint32_t j = 0; j < br[lv]; ++j) {
                ++dense;
                ids.push_back(1000 + 7 * dense);
                parent.push_back(p);
                depth.push_back(lv + 1);
                level_of.push_back(lv + 1);
                nxt.push_back((uint32_t)ids.size() - 1);
            }
        level = nxt;
    }
    if (ids.size() > 65535) return set_err(c, LMAT_E_CAPACITY, "synthetic taxonomy above 65535 nodes");
    // ids are already ascending, so internal index = position + 1
    T.n = (uint32_t)ids.size();
    T.tid32.assign(T.n + 1, 0); T.fdepth.assign(T.n + 1, 0); T.flags.assign(T.n + 1, 0);
    T.species_of.assign(T.n + 1, 0); T.path_off.assign(T.n + 1, 0); T.path_len.assign(T.n + 1, 0);
    T.conv.assign(65536, 0);
    for (uint32_t i = 0; i < T.n; ++i) {
        const uint32_t ix = i + 1;
        T.tid32[ix] = ids[i];
        T.index_of[ids[i]] = ix;
        T.fdepth[ix] = (uint16_t)depth[i];
        if (level_of[i] == 6) T.flags[ix] |= kFlagStrain;
        const uint16_t t16 = (uint16_t)(i == 0 ? 1 : i + 1);
        T.conv[t16] = ids[i];
        T.br[ids[i]] = t16;
        T.path_off[ix] = (uint32_t)T.paths.size();
        uint32_t cur = i;
        while (parent[cur] != cur) { cur = parent[cur]; T.paths.push_back(cur + 1); }
        T.path_len[ix] = (uint16_t)(T.paths.size() - T.path_off[ix]);
        if (level_of[i] == 6) T.species_of[ix] = parent[i] + 1;
    }
    T.loaded = true;
    build_euler_intervals(T);
    for (int i = 0; i < 6; ++i) c->synth_branching[i] = br[i];
    c->synth_strains_per_species = br[5];
    c->synth_n_species = (uint32_t)level.size() / br[5];
    c->synth_strain_idx.clear();
    c->synth_species_idx.clear();
    for (uint32_t s : level) c->synth_strain_idx.push_back((uint16_t)(s + 1));
    for (uint32_t sp = 0; sp < c->synth_n_species; ++sp)
        c->synth_species_idx.push_back((uint16_t)(parent[level[sp * br[5]]] + 1));
    return upload_taxonomy(c);
}

int lmat_synth_db_build(lmat_ctx* c, int k, uint64_t G, uint64_t seed, uint64_t table_bytes) {
    return lmat_synth_db_build2(c, k, G, seed, table_bytes, 100, 1);
}

int lmat_synth_db_build2(lmat_ctx* c, int k, uint64_t G, uint64_t seed, uint64_t table_bytes, uint32_t genus_block_permille,
                         uint32_t list_replicas) {
    const uint32_t none[3] = {0, 0, 0};
    return lmat_synth_db_build3(c, k, G, seed, table_bytes, genus_block_permille, list_replicas, none);
}

// the taxid list of conserved block `level` (0 family, 1 phylum, 2 superkingdom) of group `grp`, as tax_histo's LCA closure gives
// it: every strain of the group, their species, and every inner node up to the group's own (src/kmerdb/TaxTree.hpp:160-260)
static void cons_group_ids(const lmat_ctx* c, int level, uint32_t grp, std::vector<uint16_t>& internal) {
    const HostTaxonomy& T = c->tax;
    const SynthGeo& g = c->synth_geo;
    const uint32_t gsz = g.csz[level], S = g.S, sp0 = grp * gsz;
    internal.clear();
    for (uint32_t s = 0; s < gsz * S; ++s) internal.push_back(c->synth_strain_idx[sp0 * S + s]);
    std::vector<uint16_t> inner;
    const uint32_t up = (uint32_t)level + 2;   // levels above the species that belong to the group: genus .. the group's root
    for (uint32_t q = 0; q < gsz; ++q) {
        const uint16_t sp = c->synth_species_idx[sp0 + q];
        internal.push_back(sp);
        for (uint32_t j = 0; j < up && j < T.path_len[sp]; ++j) inner.push_back((uint16_t)T.paths[T.path_off[sp] + j]);
    }
    std::sort(inner.begin(), inner.end());
    inner.erase(std::unique(inner.begin(), inner.end()), inner.end());
    internal.insert(internal.end(), inner.begin(), inner.end());
}

int lmat_synth_db_build3(lmat_ctx* c, int k, uint64_t G, uint64_t seed, uint64_t table_bytes, uint32_t genus_block_permille,
                         uint32_t list_replicas, const uint32_t* conserved_permille3) {
    if (!c || !c->tax.loaded || !c->synth_n_species) return set_err(c, LMAT_E_ARG, "lmat_synth_taxonomy first");
    if (!conserved_permille3) return LMAT_E_ARG;
    if (k < 1 || k > 20 || G < (uint64_t)k) return set_err(c, LMAT_E_ARG, "bad k / genome length");
    if (genus_block_permille > 1000 || list_replicas < 1) return set_err(c, LMAT_E_ARG, "bad genus block share / replica count");
    hipSetDevice(c->device);
    const uint32_t S = c->synth_strains_per_species, NS = c->synth_n_species;
    if (S > 8) return set_err(c, LMAT_E_ARG, "at most 8 strains per species");
    const uint32_t spg = std::max<uint32_t>(c->synth_branching[4], 1);
    const uint32_t GB = spg * S;  // strains of a genus
    SynthGeo geo;
    geo.seed = seed; geo.G = G; geo.n_species = NS; geo.S = S; geo.spg = spg;
    geo.blk = GB <= 12 && NS % spg == 0 ? G * genus_block_permille / 1000 : 0;  // the genus lists are a table over 2^GB strain subsets
    if (geo.blk < (uint64_t)k) geo.blk = 0;
    {   // the heavy tail: conserved blocks behind the genus block (SynthGeo); group sizes follow the taxonomy's branching
        uint64_t at = geo.blk;
        uint32_t gsz = spg;
        for (int l = 0; l < 3; ++l) {
            gsz *= std::max<uint32_t>(c->synth_branching[3 - l], 1);   // genera per family, families per phylum, phyla per superkingdom
            uint64_t len = conserved_permille3[l] > 1000 ? 0 : G * conserved_permille3[l] / 1000;
            if (len < (uint64_t)k || NS % gsz != 0) len = 0;
            at += len;
            geo.cend[l] = at;
            geo.csz[l] = gsz;
        }
        if (geo.cend[2] >= G) return set_err(c, LMAT_E_ARG, "the shared blocks leave no room for the species' own bases");
    }
    const uint32_t NG = NS / spg;
    const HostTaxonomy& T = c->tax;
    auto id16 = [&](uint16_t internal) { return T.br.at(T.tid32[internal]); };
    // one list per (species, non-empty strain subset): owners + the species when >= 2 owners, and, for the windows of the
    // genus block, per (genus, non-empty subset of its strains): owners + their species + the genus when the owners span
    // species -- what tax_histo's LCA closure yields (src/kmerdb/TaxTree.hpp:160-260).  Records are built on all host cores.
    const uint64_t g_off = (uint64_t)NS << S;
    const uint64_t c_off0 = g_off + (geo.blk ? (uint64_t)NG << GB : 0);
    uint64_t n_cons = 0;
    for (int l = 0; l < 3; ++l) { geo.coff[l] = c_off0 + n_cons; n_cons += geo.cend[l] > (l ? geo.cend[l - 1] : geo.blk) ? NS / geo.csz[l] : 0; }
    std::vector<uint32_t> list_payload(c_off0 + n_cons, 0);
    c->synth_geo = geo;   // (cons_group_ids reads the group sizes)
    struct Made { uint64_t slot; std::vector<uint16_t> rec; };
    const unsigned nthr = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::vector<std::vector<Made>> made(nthr);
    std::vector<int> bad(nthr, 0);
    {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nthr; ++t)
            th.emplace_back([&, t]() {
                std::vector<uint16_t> raw, rec;
                for (uint32_t sp = t; sp < NS; sp += nthr)
                    for (uint32_t mask = 1; mask < (1u << S); ++mask) {
                        if (__builtin_popcount(mask) == 1) continue;
                        raw.clear();
                        for (uint32_t s = 0; s < S; ++s) if (mask & (1u << s)) raw.push_back(id16(c->synth_strain_idx[sp * S + s]));
                        raw.push_back(id16(c->synth_species_idx[sp]));
                        if (!build_list_record(c, raw, rec)) { bad[t] = 1; return; }
                        made[t].push_back({((uint64_t)sp << S) + mask, rec});
                    }
                if (!geo.blk) return;
                for (uint32_t ge = t; ge < NG; ge += nthr)
                    for (uint32_t mask = 1; mask < (1u << GB); ++mask) {
                        uint32_t spmask = 0;
                        for (uint32_t s = 0; s < GB; ++s) if (mask & (1u << s)) spmask |= 1u << (s / S);
                        if (__builtin_popcount(spmask) < 2) continue;  // owners within one species: the species list (or a single strain)
                        raw.clear();
                        for (uint32_t s = 0; s < GB; ++s) if (mask & (1u << s)) raw.push_back(id16(c->synth_strain_idx[ge * GB + s]));
                        for (uint32_t q = 0; q < spg; ++q) if (spmask & (1u << q)) raw.push_back(id16(c->synth_species_idx[ge * spg + q]));
                        const uint16_t sp0 = c->synth_species_idx[ge * spg];
                        raw.push_back(id16(T.paths[T.path_off[sp0]]));  // the genus: parent of its species
                        if (!build_list_record(c, raw, rec)) { bad[t] = 1; return; }
                        made[t].push_back({g_off + ((uint64_t)ge << GB) + mask, rec});
                    }
            });
        for (auto& x : th) x.join();
    }
    for (int b_ : bad) if (b_) return LMAT_E_TAXONOMY;
    for (int l = 0; l < 3; ++l) {   // one list per conserved group
        if (geo.cend[l] <= (l ? geo.cend[l - 1] : geo.blk)) continue;
        std::vector<uint16_t> ids, raw, rec;
        for (uint32_t grp = 0; grp < NS / geo.csz[l]; ++grp) {
            cons_group_ids(c, l, grp, ids);
            raw.clear();
            for (uint16_t x : ids) raw.push_back(id16(x));
            if (!build_list_record(c, raw, rec)) return LMAT_E_TAXONOMY;
            made[0].push_back({geo.coff[l] + grp, rec});
        }
    }
    // records on the smallest alignment at which all copies fit the 24-bit payloads (16 bytes: 256 MB ... 256 bytes: 4 GB)
    std::vector<uint16_t> arena;
    uint64_t block_units = 0;
    uint32_t shift = 0;
    if (const char* e = getenv("LMAT_LIST_SHIFT")) shift = (uint32_t)std::min(std::max(atoi(e), 0), kListShiftMax);
    for (;; ++shift) {
        const size_t unit = (size_t)kListUnit << shift;
        arena.assign(unit, 0);  // record 0 reserved
        for (auto& v : made)
            for (auto& m : v) list_payload[m.slot] = arena_append(arena, m.rec, shift);
        arena.resize((arena.size() + unit - 1) / unit * unit, 0);
        block_units = arena.size() / unit;   // payload units of one copy of the records
        if (block_units * list_replicas + kListBase <= kPayloadMask) break;
        if (shift == (uint32_t)kListShiftMax) return set_err(c, LMAT_E_CAPACITY, "taxid-list arena exceeds the 24-bit payload range even at 256-byte records");
    }
    made.clear();
    for (uint32_t sp = 0; sp < NS; ++sp)  // single owners: plain strain payloads
        for (uint32_t s = 0; s < S; ++s) list_payload[((uint64_t)sp << S) + (1u << s)] = c->synth_strain_idx[sp * S + s];
    if (geo.blk)
        for (uint32_t ge = 0; ge < NG; ++ge)
            for (uint32_t mask = 1; mask < (1u << GB); ++mask) {
                uint32_t spmask = 0;
                for (uint32_t s = 0; s < GB; ++s) if (mask & (1u << s)) spmask |= 1u << (s / S);
                if (__builtin_popcount(spmask) >= 2) continue;
                const uint32_t q = (uint32_t)__builtin_ctz(spmask);  // all owners in species q of the genus: that species' list
                list_payload[g_off + ((uint64_t)ge << GB) + mask] = list_payload[((uint64_t)(ge * spg + q) << S) + ((mask >> (q * S)) & ((1u << S) - 1))];
            }
    c->dev.list_shift = shift;
    int rc;
    {   // the device arena: list_replicas copies back to back (+ 64 bytes of slack: a record's first 64 bytes are read whatever its length)
        if (c->dev.arena) { hipFree(c->dev.arena); c->dev.arena = nullptr; }
        const size_t bytes = arena.size() * 2;
        HIPCHK(c, hipMalloc((void**)&c->dev.arena, bytes * list_replicas + 64));
        HIPCHK(c, hipMemset((char*)c->dev.arena + bytes * list_replicas, 0, 64));
        HIPCHK(c, hipMemcpy(c->dev.arena, arena.data(), bytes, hipMemcpyHostToDevice));
        for (uint32_t r = 1; r < list_replicas; ++r)
            HIPCHK(c, hipMemcpy((char*)c->dev.arena + bytes * r, c->dev.arena, bytes, hipMemcpyDeviceToDevice));
        c->arena_words = arena.size() * list_replicas;
    }
    uint32_t* d_lp = nullptr;
    if ((rc = dev_upload(c, &d_lp, list_payload))) return rc;
    if ((rc = dev_upload(c, &c->d_synth_strain_idx, c->synth_strain_idx))) return rc;
    // expected distinct k-mers ~ windows * (1 + owners * P(window mutated)); size the table from that
    const double pm = 1.0 - std::pow(0.99, k);
    const double npos = (double)(G - k + 1), nblk = geo.blk ? (double)(geo.blk - k + 1) : 0.0;
    double ncons = 0, cons_kmers = 0;   // windows of the conserved blocks: one k-mer per group, no strain variants
    for (int l = 0; l < 3; ++l) {
        const double len = (double)(geo.cend[l] - (l ? geo.cend[l - 1] : geo.blk));
        ncons += len;
        cons_kmers += len * (double)(NS / geo.csz[l]);
    }
    const uint64_t est = (uint64_t)((double)NS * (npos - nblk - ncons) * (1.0 + S * pm) + (double)NG * nblk * (1.0 + GB * pm) + cons_kmers);
    if ((rc = alloc_table(c, est, table_bytes, k))) return rc;
    c->dev.k = k;
    uint32_t* d_fail = nullptr;
    unsigned long long* d_ins = nullptr;
    HIPCHK(c, hipMalloc((void**)&d_fail, 4));
    HIPCHK(c, hipMalloc((void**)&d_ins, 32));
    HIPCHK(c, hipMemsetAsync(d_fail, 0, 4, c->stream));
    HIPCHK(c, hipMemsetAsync(d_ins, 0, 32, c->stream));
    launch_synth_db(c->dev, geo, k, c->d_synth_strain_idx, d_lp, g_off, list_replicas, (uint32_t)block_units, d_fail, d_ins, c->stream);
    launch_table_count(c->dev, d_ins + 1, c->stream);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    uint32_t fail = 0;
    unsigned long long ins[4] = {0, 0, 0, 0};
    HIPCHK(c, hipMemcpy(&fail, d_fail, 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(ins, d_ins, 32, hipMemcpyDeviceToHost));
    hipFree(d_fail); hipFree(d_ins); hipFree(d_lp);
    if (fail) return set_err(c, LMAT_E_CAPACITY, "hash table (or its overflow table) full during the synthetic build: the build stopped at the first k-mer without a slot");
    if (c->dev.cpt.nb && (rc = check_overflow_fill(c, ins[2]))) return rc;
    c->n_kmers = ins[1];
    c->synth_genome_len = G;
    c->synth_seed = seed;
    c->synth_geo = geo;
    c->n_lists = (uint64_t)list_payload.size() * list_replicas;
    c->db_ready = true;
    return LMAT_OK;
}

// Test hook: what the synthetic database must hold for the ancestor window of `species` at `pos`, derived on the HOST from the
// generator's own functions (never from the device table): the canonical k-mer and the taxid list tax_histo's LCA closure gives
// its owners (include/lmat_hip.h).  *n = 0: every strain mutated the window, the k-mer was not filed.
int lmat_synth_window(lmat_ctx* c, uint32_t species, uint64_t pos, uint64_t* kmer, uint32_t* tids, uint32_t cap, uint32_t* n) {
    if (!c || !kmer || !n) return LMAT_E_ARG;
    if (!c->synth_genome_len) return set_err(c, LMAT_E_ARG, "lmat_synth_db_build first");
    const SynthGeo& g = c->synth_geo;
    const int k = c->dev.k;
    if (species >= g.n_species || pos + k > g.G) return set_err(c, LMAT_E_ARG, "window outside the synthetic genomes");
    uint32_t first = 0, mask = 0;
    bool inblk = false;
    int clv = -1;
    synth_window_host(g, k, species, pos, kmer, &first, &mask, &inblk, &clv);
    const HostTaxonomy& T = c->tax;
    if (clv >= 0) {  // a conserved block: the whole group's list
        std::vector<uint16_t> ids;
        cons_group_ids(c, clv, species / g.csz[clv], ids);
        *n = (uint32_t)ids.size();
        for (uint32_t i = 0; i < ids.size() && i < cap; ++i) if (tids) tids[i] = T.tid32[ids[i]];
        return LMAT_OK;
    }
    std::vector<uint32_t> out;
    const uint32_t S = g.S, ns = inblk ? g.spg * S : S;
    uint32_t spmask = 0;
    for (uint32_t s = 0; s < ns; ++s)
        if (mask & (1u << s)) { out.push_back(T.tid32[c->synth_strain_idx[first + s]]); spmask |= 1u << (s / S); }
    if (out.size() >= 2) {  // owners + their species, + the genus when they span species (src/kmerdb/TaxTree.hpp:160-260)
        const uint32_t sp0 = first / S;
        for (uint32_t q = 0; q < (inblk ? g.spg : 1u); ++q)
            if (spmask & (1u << q)) out.push_back(T.tid32[c->synth_species_idx[sp0 + q]]);
        if (__builtin_popcount(spmask) >= 2) out.push_back(T.tid32[T.paths[T.path_off[c->synth_species_idx[sp0]]]]);
    }
    *n = (uint32_t)out.size();
    for (uint32_t i = 0; i < out.size() && i < cap; ++i) if (tids) tids[i] = out[i];
    return LMAT_OK;
}

// Test hook: the k-mers a synthetic read must find in the database, derived on the HOST from the two generators (reads and
// genomes) alone.  For read r of lmat_reads_synth(lengths, seed): every window that lies in its strain's genome without a
// substituted base or the N -> the canonical k-mer and the list the database must hold for it: the ancestor window's list
// (lmat_synth_window) where the strain carries the window unmutated, else the strain alone (its own mutated copy was filed as a
// singleton).  Windows with an error, random and low-complexity reads give nothing: whatever the table returns for those is a
// chance hit.  tids: [cap][stride]; -> *n windows, *read_len.
int lmat_synth_read_windows(lmat_ctx* c, const uint32_t* lengths, uint32_t n_lengths, uint64_t seed, uint64_t r, uint64_t* kmers,
                            uint32_t* tids, uint32_t* counts, uint32_t cap, uint32_t stride, uint32_t* n, uint32_t* read_len) {
    if (!c || !lengths || !n_lengths || !kmers || !tids || !counts || !n || !stride) return LMAT_E_ARG;
    if (!c->synth_genome_len) return set_err(c, LMAT_E_ARG, "lmat_synth_db_build first");
    const SynthGeo& g = c->synth_geo;
    const int k = c->dev.k;
    const HostTaxonomy& T = c->tax;
    std::vector<uint64_t> gpos(cap);
    std::vector<uint32_t> rpos(cap);
    uint32_t len = 0, sg = 0;
    const uint32_t nw = synth_read_windows_host(g, lengths, n_lengths, seed, r, k, &len, &sg, gpos.data(), rpos.data(), cap);
    if (read_len) *read_len = len;
    const uint32_t S = g.S, sp = sg / S;
    for (uint32_t w = 0; w < nw; ++w) {
        uint64_t km = 0;
        uint32_t first = 0, mask = 0;
        bool inblk = false;
        int clv = -1;
        synth_window_host(g, k, sp, gpos[w], &km, &first, &mask, &inblk, &clv);
        uint32_t* out = tids + (size_t)w * stride;
        uint32_t cnt = 0;
        auto put = [&](uint32_t t) { if (cnt < stride) out[cnt] = t; ++cnt; };
        if (clv >= 0) {  // a conserved block: the whole group's list (longer than `stride` it is cut: counts[w] still says how long)
            std::vector<uint16_t> ids;
            cons_group_ids(c, clv, sp / g.csz[clv], ids);
            for (uint16_t x : ids) put(T.tid32[x]);
        } else if (mask & (1u << (sg - first))) {  // the strain carries the ancestor's window: its owners' list
            const uint32_t ns = inblk ? g.spg * S : S;
            uint32_t spmask = 0, owners = 0;
            for (uint32_t s_ = 0; s_ < ns; ++s_)
                if (mask & (1u << s_)) { put(T.tid32[c->synth_strain_idx[first + s_]]); spmask |= 1u << (s_ / S); ++owners; }
            if (owners >= 2) {
                const uint32_t sp0 = first / S;
                for (uint32_t q = 0; q < (inblk ? g.spg : 1u); ++q)
                    if (spmask & (1u << q)) put(T.tid32[c->synth_species_idx[sp0 + q]]);
                if (__builtin_popcount(spmask) >= 2) put(T.tid32[T.paths[T.path_off[c->synth_species_idx[sp0]]]]);
            }
        } else {  // its own copy of the window, filed under the strain alone
            uint64_t f = 0;
            for (int j = 0; j < k; ++j) f = (f << 2) | synth_strain_base_host(g, sp, sg, gpos[w] + j);
            const uint64_t rcw = revcomp_fwd(f, k);
            km = f < rcw ? f : rcw;
            put(T.tid32[c->synth_strain_idx[sg]]);
        }
        kmers[w] = km;
        counts[w] = cnt;
    }
    *n = nw;
    return LMAT_OK;
}

// ---------------------------------------------------------------------------------- reads
// The fast kernel's capacity class follows the bulk of the batch, not its longest read: the length below which
// 99 % of the reads fall; longer ones are re-run by a larger class through the device-side overflow list.
static uint32_t bulk_length(std::vector<uint32_t>& lens) {
    if (lens.empty()) return 0;
    const size_t kth = (size_t)((lens.size() - 1) * 0.99);
    std::nth_element(lens.begin(), lens.begin() + kth, lens.end());
    return lens[kth];
}

static int reads_alloc(lmat_ctx* c, const std::vector<uint64_t>& rec_off, uint32_t max_len, lmat_reads** out) {
    lmat_reads* r = new lmat_reads();
    r->n = rec_off.size() - 1;
    r->n_words = rec_off.back();
    r->max_len = max_len;
    if (hipMalloc((void**)&r->words, std::max<uint64_t>(r->n_words, 1) * 4 + 16384) != hipSuccess ||  // kernels read whole-record tiles: pad
        hipMalloc((void**)&r->rec_off, rec_off.size() * 8) != hipSuccess) {
        if (r->words) hipFree(r->words);
        delete r;
        return set_err(c, LMAT_E_NOMEM, "out of device memory for reads");
    }
    hipMemcpyAsync(r->rec_off, rec_off.data(), rec_off.size() * 8, hipMemcpyHostToDevice, c->stream);
    *out = r;
    return LMAT_OK;
}

int lmat_reads_upload(lmat_ctx* c, const uint8_t* bases, const uint64_t* off, uint64_t n, lmat_reads** out) {
    if (!c || !out || (n && (!bases || !off))) return LMAT_E_ARG;
    hipSetDevice(c->device);
    std::vector<uint64_t> rec_off(n + 1, 0);
    uint32_t max_len = 0;
    std::vector<uint32_t> lens(n);
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t len = off[i + 1] - off[i];
        if (len > 0x7FFFFFFF) return set_err(c, LMAT_E_ARG, "read too long");
        max_len = std::max<uint32_t>(max_len, (uint32_t)len);
        lens[i] = (uint32_t)len;
        rec_off[i + 1] = rec_off[i] + rec_words((uint32_t)len);
    }
    int rc = reads_alloc(c, rec_off, max_len, out);
    if (rc) return rc;
    (*out)->lens = lens;
    (*out)->class_len = bulk_length(lens);
    if (!n) return LMAT_OK;
    uint8_t* d_b = nullptr;
    uint64_t* d_o = nullptr;
    HIPCHK(c, hipMalloc((void**)&d_b, off[n] + 32));  // (the pack kernel reads whole dwords: the one holding the last base may end past it)
    HIPCHK(c, hipMalloc((void**)&d_o, (n + 1) * 8));
    HIPCHK(c, hipMemcpyAsync(d_b, bases, off[n], hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_o, off, (n + 1) * 8, hipMemcpyHostToDevice, c->stream));
    launch_pack_reads(d_b, d_o, (*out)->rec_off, (*out)->words, n, c->stream);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    hipFree(d_b); hipFree(d_o);
    return LMAT_OK;
}

int lmat_reads_synth(lmat_ctx* c, uint64_t n, const uint32_t* lengths, uint32_t n_lengths, uint64_t seed, lmat_reads** out) {
    if (!c || !out || !lengths || !n_lengths) return LMAT_E_ARG;
    if (!c->db_ready || !c->synth_genome_len) return set_err(c, LMAT_E_ARG, "synthetic reads need lmat_synth_db_build");
    hipSetDevice(c->device);
    // the kernel picks lengths[(h>>48) % n_lengths]; offsets must match, so mirror that choice here
    std::vector<uint64_t> rec_off(n + 1, 0);
    std::vector<uint32_t> lens(n);
    uint32_t max_len = 0;
    for (uint64_t r = 0; r < n; ++r) {
        const uint64_t h0 = splitmix(seed ^ (r * 0x9E3779B97F4A7C15ull));
        const uint32_t len = lengths[(uint32_t)((h0 >> 48) % n_lengths)];
        lens[r] = len;
        max_len = std::max(max_len, len);
        rec_off[r + 1] = rec_off[r] + rec_words(len);
    }
    int rc = reads_alloc(c, rec_off, max_len, out);
    if (rc) return rc;
    (*out)->class_len = max_len;  // a handful of configured lengths: no tail to cut
    (*out)->lens.swap(lens);
    uint32_t* d_len = nullptr;
    HIPCHK(c, hipMalloc((void**)&d_len, n_lengths * 4));
    HIPCHK(c, hipMemcpyAsync(d_len, lengths, n_lengths * 4, hipMemcpyHostToDevice, c->stream));
    launch_synth_reads((*out)->words, (*out)->rec_off, d_len, n_lengths, n, seed, c->synth_geo, c->stream);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    hipFree(d_len);
    return LMAT_OK;
}

int lmat_reads_download_ascii(lmat_ctx* c, const lmat_reads* r, uint64_t first, uint64_t count, uint8_t* bases,
                              uint64_t* off) {
    if (!c || !r || first + count > r->n) return LMAT_E_ARG;
    hipSetDevice(c->device);
    std::vector<uint64_t> ro(count + 1);
    HIPCHK(c, hipMemcpy(ro.data(), r->rec_off + first, (count + 1) * 8, hipMemcpyDeviceToHost));
    std::vector<uint32_t> w(ro[count] - ro[0]);
    if (!w.empty()) HIPCHK(c, hipMemcpy(w.data(), r->words + ro[0], w.size() * 4, hipMemcpyDeviceToHost));
    uint64_t o = 0;
    static const char A[4] = {'A', 'C', 'G', 'T'};
    for (uint64_t i = 0; i < count; ++i) {
        const uint32_t* rec = w.data() + (ro[i] - ro[0]);
        const uint32_t len = rec[0], nb = (len + 15) / 16;
        off[i] = o;
        if (bases)
            for (uint32_t p = 0; p < len; ++p) {
                const bool valid = (rec[1 + nb + (p >> 5)] >> (p & 31)) & 1u;
                bases[o + p] = valid ? A[(rec[1 + (p >> 4)] >> (2 * (p & 15))) & 3u] : 'N';
            }
        o += len;
    }
    off[count] = o;
    return LMAT_OK;
}

uint64_t lmat_reads_count(const lmat_reads* r) { return r ? r->n : 0; }
uint64_t lmat_reads_device_bytes(const lmat_reads* r) { return r ? r->n_words * 4 + (r->n + 1) * 8 : 0; }
void lmat_reads_free(lmat_ctx* c, lmat_reads* r) {
    if (!r) return;
    if (c) hipSetDevice(c->device);
    if (r->words) hipFree(r->words);
    if (r->rec_off) hipFree(r->rec_off);
    for (int j = 0; j < kNCls; ++j) if (r->cls_dev[j]) hipFree(r->cls_dev[j]);
    delete r;
}

// ---------------------------------------------------------------------------------- classify
static int ensure_results(lmat_ctx* c, uint64_t count, uint64_t cand_cap) {
    if (count > c->results_cap) {
        if (c->d_results) hipFree(c->d_results);
        c->d_results = nullptr;
        HIPCHK(c, hipMalloc((void**)&c->d_results, count * sizeof(lmat_read_result)));
        c->results_cap = count;
    }
    if (cand_cap > c->cands_cap) {
        if (c->d_cands) hipFree(c->d_cands);
        c->d_cands = nullptr;
        HIPCHK(c, hipMalloc((void**)&c->d_cands, cand_cap * sizeof(lmat_cand)));
        c->cands_cap = cand_cap;
    }
    return LMAT_OK;
}
// per-batch scratch of the kernels (overflow lists, the K4 hand-off records): sized by the largest batch so far
static int ensure_scratch(lmat_ctx* c, uint64_t count) {
    if (count > c->ovf_cap) {
        if (c->d_ovf) hipFree(c->d_ovf);
        if (c->d_k4buf) hipFree(c->d_k4buf);
        c->d_ovf = nullptr;
        c->d_k4buf = nullptr;
        HIPCHK(c, hipMalloc((void**)&c->d_k4buf, count * (uint64_t)kK4RecWords * sizeof(uint32_t)));
        if (c->d_k4small) hipFree(c->d_k4small);
        if (c->d_k4large) hipFree(c->d_k4large);
        c->d_k4small = c->d_k4large = nullptr;
        HIPCHK(c, hipMalloc((void**)&c->d_k4small, count * sizeof(uint32_t)));
        HIPCHK(c, hipMalloc((void**)&c->d_k4large, 3 * count * sizeof(uint32_t)));  // three lists: large tables | up to 32 taxids | up to 16, by rows
        if (c->d_k4bail) hipFree(c->d_k4bail);
        c->d_k4bail = nullptr;
        HIPCHK(c, hipMalloc((void**)&c->d_k4bail, count * sizeof(uint32_t)));
        HIPCHK(c, hipMalloc((void**)&c->d_ovf, count * sizeof(uint32_t)));
        if (c->d_ovf2) hipFree(c->d_ovf2);
        if (c->d_ovf3) hipFree(c->d_ovf3);
        if (c->d_ovf4) hipFree(c->d_ovf4);
        c->d_ovf2 = c->d_ovf3 = c->d_ovf4 = nullptr;
        HIPCHK(c, hipMalloc((void**)&c->d_ovf2, count * sizeof(uint32_t)));
        HIPCHK(c, hipMalloc((void**)&c->d_ovf3, count * sizeof(uint32_t)));
        HIPCHK(c, hipMalloc((void**)&c->d_ovf4, count * sizeof(uint32_t)));
        c->ovf_cap = count;
    }
    return LMAT_OK;
}

static ClassifyArgs make_args(lmat_ctx* c, const lmat_reads* reads, uint64_t first, uint64_t count, bool want_cands,
                              uint64_t cand_cap) {
    ClassifyArgs a;
    a.tb = c->dev;
    a.prm = kparams(c->params);
    a.prm.permissive = c->permissive;
    a.words = reads->words;
    a.rec_off = reads->rec_off;
    a.index = nullptr;
    a.first = first;
    a.count = count;
    a.result_base = first;
    a.results = c->out_results ? c->out_results : c->d_results;
    a.cands = want_cands ? (c->out_cands ? c->out_cands : c->d_cands) : nullptr;
    a.cand_cap = cand_cap;
    // Large candidate buffers are handed out through sub-cursors, a chunk at a time (kernels.hpp): at most kCandSubs chunks lie
    // partly unused at the end of a launch, an eighth of the buffer in the worst case.  Small ones -- few reads, no contention --
    // are bumped read by read as before.  (LMAT_CAND_CHUNK=0 turns the sub-cursors off.)
    // Batches of up to 128 K reads bump the cursor read by read (no contention to speak of, and no slack: the CLI's batches are
    // 50 K reads of an 8 MB piece); larger ones use one sub-cursor per 4096 reads, 2048 pairs a chunk: the slack -- chunks partly
    // used when the launch ends -- stays below half a pair per read.  (LMAT_CAND_CHUNK=0 turns the sub-cursors off.)
    a.cand_chunk = 0;
    a.cand_sub_mask = 0;
    if (want_cands && cand_cap <= 0x7FFFFFFFull && count > (128u << 10)) {   // (a sub-cursor's two halves are 32 bits wide each)
        static const int forced = getenv("LMAT_CAND_CHUNK") ? atoi(getenv("LMAT_CAND_CHUNK")) : -1;
        uint32_t subs = 32;
        while (subs < (uint32_t)kCandSubs && (uint64_t)subs * 4096 < count) subs *= 2;
        uint64_t ch = 2048;
        while (ch > 256 && ch * subs * 4 > cand_cap) ch /= 2;   // a tight buffer: smaller chunks rather than a quarter of it as slack
        if (ch * subs * 4 > cand_cap) ch = 0;
        if (forced >= 0) ch = forced >= 64 ? (uint64_t)forced : 0;
        a.cand_chunk = (uint32_t)ch;
        a.cand_sub_mask = subs - 1;
    }
    a.cursor = c->d_cursor;
    a.err = c->batch_err ? c->d_cursor + 15 : c->d_err;  // a streamed batch keeps its own flags (word 15 of the per-batch counter block)
    a.counts = c->out_counts ? c->out_counts : c->d_counts;
    auto it = c->tax.index_of.find(32630);
    a.phix_call_idx = it == c->tax.index_of.end() ? 0 : it->second;
    a.ovf_list = c->d_ovf;
    a.ovf_slot = 2;
    a.count_ptr = nullptr;
    a.k4buf = c->d_k4buf;
    a.gscratch = nullptr;
    a.rand_max = nullptr;
    a.rand_cnt = nullptr;
    a.rand_gc = nullptr;
    a.rand_nb = 0;
    if (c->rand_launch) { a.rand_max = c->d_rand_max; a.rand_cnt = c->d_rand_cnt; a.rand_gc = c->d_rand_gc; a.rand_nb = c->rand_nb; }
    a.k4_small = c->d_k4small;
    a.k4_large = c->d_k4large;
    a.k4_mid = c->d_k4large + c->ovf_cap;
    a.k4_row = c->d_k4large + 2 * c->ovf_cap;
    a.k4_bail = c->d_k4bail;
    a.k4_slot = 5;
    a.nm = c->nm;
    a.gene_mode = (uint32_t)c->gene_mode;
    if (c->gene_mode) a.prm.min_kmer = 0;  // gene_label looks every read of >= k bases up (gene_label.cpp:276-288)
    return a;
}

// the set of per-batch buffers in use <-> the parked one
static void swap_sets(lmat_ctx* c) {
    auto& p = c->parked;
    std::swap(c->d_results, p.d_results); std::swap(c->results_cap, p.results_cap);
    std::swap(c->d_cands, p.d_cands); std::swap(c->cands_cap, p.cands_cap);
    std::swap(c->d_cursor, p.d_cursor); std::swap(c->d_ovf, p.d_ovf); std::swap(c->d_ovf2, p.d_ovf2); std::swap(c->d_ovf3, p.d_ovf3); std::swap(c->d_ovf4, p.d_ovf4);
    std::swap(c->d_k4buf, p.d_k4buf); std::swap(c->d_k4small, p.d_k4small); std::swap(c->d_k4large, p.d_k4large);
    std::swap(c->d_k4bail, p.d_k4bail); std::swap(c->ovf_cap, p.ovf_cap);
    std::swap(c->d_tail, p.d_tail); std::swap(c->tail_bytes, p.tail_bytes);
    std::swap(c->ev_done, p.done); std::swap(c->set_in_flight, p.in_flight);
}

// Queued launches (lmat_classify_async, the streamed boundary) take the two sets of per-batch buffers in turn, and what follows
// a batch's classify kernel -- the re-run classes and the general decision path, each over the few reads the fast path passed
// on -- runs on the side streams beside the next batch's classify kernel.  Round 2 measured +3 % for this with the decision
// step of EVERY read in those kernels (they took LDS from the classify kernel: 8.2 -> 9.3 ms); now that the decision is made on
// the classify wave itself the tail is ~0.4 ms of nearly empty launches and hiding it is a plain gain (64 GiB table, 2 M reads
// per launch: 7.42 -> 7.17 ms per launch).  LMAT_PIPELINE=0 puts every launch back on the context's stream.
static bool pipeline_on() {
    static const bool on = !getenv("LMAT_PIPELINE") || atoi(getenv("LMAT_PIPELINE")) != 0;
    return on;
}
// pipelined: the launch takes the other set of per-batch buffers and returns with its decision kernels still running on the
// side streams (beside whatever the caller queues next on the context's stream); the set's `done` event marks their end.
// Otherwise the context's stream waits for them, as every caller that reads results right away needs.
static int run_classify(lmat_ctx* c, const lmat_reads* reads, uint64_t first, uint64_t count, bool want_cands,
                        uint64_t cand_cap, bool timed, bool pipelined = false) {
    if (!c->db_ready) return set_err(c, LMAT_E_ARG, "database not ready");
    if (first + count > reads->n) return set_err(c, LMAT_E_ARG, "read range out of bounds");
    if (count > 0xFFFFFFFFull) return set_err(c, LMAT_E_ARG, "batch above 2^32 reads");
    hipSetDevice(c->device);
    if (pipelined) swap_sets(c);
    if (!c->ev_done) HIPCHK(c, hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
    if (c->set_in_flight) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_done, 0));  // the set's previous batch has to be through
    int rc = ensure_results(c, c->out_results ? 0 : count, want_cands && !c->out_cands ? cand_cap : 0);
    if (!rc) rc = ensure_scratch(c, count);
    if (rc) return rc;
    if (reads->n > 0xFFFFFFFFull) return set_err(c, LMAT_E_ARG, "read set above 2^32 reads");
    if ((int)reads->max_len > classify_max_read_len())
        return set_err(c, LMAT_E_CAPACITY, "read longer than " + std::to_string(classify_max_read_len()) + " bases");
    if (!c->d_gscratch)  // per-read tables of the global-memory class (very long reads, very large taxid tables)
        HIPCHK(c, hipMalloc((void**)&c->d_gscratch, classify_gmem_scratch_bytes()));
    // per-launch counters: the candidate cursor [0] and the list lengths [2..]; the error word (d_err) is sticky -- launches
    // only OR into it and whoever reports it (lmat_sync, lmat_classify, lmat_rand_label) clears it
    HIPCHK(c, hipMemsetAsync(c->d_cursor, 0, want_cands ? kCursorBytes : (size_t)kCursorWords * 4, c->stream));  // (the sub-cursors start empty: next free = end = 0)
    ClassifyArgs a = make_args(c, reads, first, count, want_cands, cand_cap);
    // Tails (tail_kernel): where the reads of the 160-k-mer class end a few positions past their second chunk -- 150 bp reads
    // at k = 20 have 131 -- those positions are looked up beforehand, 4, 8 or 16 lanes per read.  Compact layout, no null
    // models (their GC accounting walks every chunk).  LMAT_TAIL=0 turns it off.
    auto setup_tail = [&](bool single_class) {  // single_class: the whole batch runs in the class its longest read asks for
        static const bool tail_on = !getenv("LMAT_TAIL") || atoi(getenv("LMAT_TAIL")) != 0;
        static const int tail_force = getenv("LMAT_TAIL_LPR") ? atoi(getenv("LMAT_TAIL_LPR")) : 0;  // experiments: 4, 8 or 16
        const uint32_t k = (uint32_t)c->dev.k;
        const uint32_t realP = reads->max_len >= k ? reads->max_len - k + 1 : 0;
        if (single_class && realP > 160u) return;
        const uint32_t maxP = std::min<uint32_t>(realP, 160u);
        uint32_t lpr = 0;
        if (tail_on && c->dev.cpt.nb && !c->nm.active && maxP > 128u) lpr = maxP <= 132u ? 4u : (maxP <= 136u ? 8u : 16u);
        if (lpr && (tail_force == 4 || tail_force == 8 || tail_force == 16) && 128u + (uint32_t)tail_force >= std::min(maxP, 144u)) lpr = (uint32_t)tail_force;
        if (lpr) {
            const uint64_t need = count * (16ull * lpr + 32ull);
            if (need > c->tail_bytes) {
                if (c->d_tail) hipFree(c->d_tail);
                c->d_tail = nullptr; c->tail_bytes = 0;
                if (hipMalloc((void**)&c->d_tail, need) == hipSuccess) c->tail_bytes = need;
                else { (void)hipGetLastError(); lpr = 0; }  // no room for them: the reads run their third chunk
            }
        }
        if (lpr) {
            a.tail_lpr = lpr;
            a.tail16 = (const uint32_t*)c->d_tail;
            a.tail_u = (const uint64_t*)(c->d_tail + count * 16ull * lpr);
        }
    };
    auto tail_launch = [&](const ClassifyArgs& s) { launch_tail(s, c->stream); };  // (on a stream of its own beside the batch before: measured, no gain -- it takes the same wave slots)
    struct Ev {  // timing events of this launch: handed to the context on success, destroyed on any early return
        hipEvent_t e[4] = {nullptr, nullptr, nullptr, nullptr};
        ~Ev() { for (hipEvent_t x : e) if (x) hipEventDestroy(x); }
    } ev;
    hipEvent_t &e0 = ev.e[0], &e1 = ev.e[1], &e2 = ev.e[2], &e3 = ev.e[3];
    if (timed) {
        HIPCHK(c, hipEventCreate(&e0));
        HIPCHK(c, hipEventCreate(&e1));
        HIPCHK(c, hipEventCreate(&e2));
        HIPCHK(c, hipEventCreate(&e3));
    }
    bool started = false;
    auto mark_start = [&]() {  // the first timing event brackets the classify kernels only: the tail kernel ahead of them is on the step's account
        if (timed && !started) hipEventRecord(e0, c->stream);
        started = true;
    };
    if (c->dev.wide) {
        // A wide taxonomy (more than 65534 ids): the classes with 32-bit ids in their tables, which decide in-kernel -- a first
        // tier of 128 taxids / 512 list elements (five waves per CU), a second of 512 / 2048, then tables in global memory.  The
        // 16-bit fast classes do not apply; this path is built for correctness on databases of 32-bit taxids, not for the headline.
        if (!c->dev.cpt.nb) return set_err(c, LMAT_E_ARG, "a wide taxonomy needs the compact table layout (k >= 10)");
        const uint32_t kk = (uint32_t)c->dev.k;
        mark_start();
        ClassifyArgs w = a;
        w.gscratch = c->d_gscratch;
        bool ok = true;
        if (reads->max_len <= 512 + kk - 1) {
            w.ovf_list = c->d_ovf; w.ovf_slot = 2;
            ok = launch_classify(w, reads->max_len, 4, c->stream);
            w.index = c->d_ovf; w.count_ptr = c->d_cursor + 2; w.count = 0;
        }
        if (ok && reads->max_len <= 2048 + kk - 1) {
            w.ovf_list = c->d_ovf2; w.ovf_slot = 3;
            ok = launch_classify(w, reads->max_len, 5, c->stream);
            w.index = c->d_ovf2; w.count_ptr = c->d_cursor + 3; w.count = 0;
        }
        w.ovf_list = nullptr; w.ovf_slot = 7;
        if (ok) ok = launch_classify(w, reads->max_len, 1, c->stream);
        if (!ok) return set_err(c, LMAT_E_CAPACITY, "read longer than " + std::to_string(classify_max_read_len()) + " bases");
        if (timed) { HIPCHK(c, hipEventRecord(e1, c->stream)); HIPCHK(c, hipEventRecord(e2, c->stream)); }
        c->join_stream = c->stream;
        HIPCHK(c, hipEventRecord(c->ev_done, c->stream));
        c->set_in_flight = true;
        if (timed) {
            HIPCHK(c, hipEventRecord(e3, c->stream));
            c->pending_events.push_back(std::make_pair(e0, e1));
            c->pending_events2.push_back(std::make_pair(e2, e3));
            e0 = e1 = e2 = e3 = nullptr;
        }
        HIPCHK(c, hipGetLastError());
        return LMAT_OK;
    }
    // Each read runs in the smallest fast class that holds it (160 / 256 / 512 k-mers; 512 = 531 bp at k = 20); longer
    // reads ride the overflow list to the wave-per-read classes behind it.  A batch of one class is one plain launch.
    {
        lmat_reads* rw = const_cast<lmat_reads*>(reads);
        const uint32_t k = (uint32_t)c->dev.k;
        if (!rw->preset && rw->cls_k != (int)k && rw->lens.size() == rw->n) {
            for (int j = 0; j < kNCls; ++j) { rw->cls_host[j].clear(); if (rw->cls_dev[j]) { hipFree(rw->cls_dev[j]); rw->cls_dev[j] = nullptr; } }
            for (uint64_t i = 0; i < rw->n; ++i) {
                const uint32_t P = rw->lens[i] >= k ? rw->lens[i] - k + 1 : 0;
                rw->cls_host[len_class(P)].push_back((uint32_t)i);
            }
            int used = 0;
            for (int j = 0; j < kNCls; ++j) used += !rw->cls_host[j].empty();
            if (used > 1)
                for (int j = 0; j < kNCls; ++j)
                    if (!rw->cls_host[j].empty()) {
                        HIPCHK(c, hipMalloc((void**)&rw->cls_dev[j], rw->cls_host[j].size() * 4));
                        HIPCHK(c, hipMemcpyAsync(rw->cls_dev[j], rw->cls_host[j].data(), rw->cls_host[j].size() * 4, hipMemcpyHostToDevice, c->stream));
                    }
            rw->cls_k = (int)k;
        }
        const uint32_t cls_len[kNCls] = {kClsU[0] + k - 1, kClsU[1] + k - 1, kClsU[2] + k - 1, kClsU[3] + k - 1};
        bool mixed = rw->cls_dev[0] || rw->cls_dev[1] || rw->cls_dev[2] || rw->cls_dev[3];
        if (rw->preset) {  // a stream slot: lists over the whole batch, already on the device
            const int used = (rw->cls_n[0] != 0) + (rw->cls_n[1] != 0) + (rw->cls_n[2] != 0) + (rw->cls_n[3] != 0);
            setup_tail(used <= 1);
            if (used <= 1) {
                tail_launch(a);
                mark_start();
                if (!launch_classify(a, std::min<uint32_t>(reads->max_len, 512 + k - 1), 0, c->stream))
                    return set_err(c, LMAT_E_CAPACITY, "read longer than " + std::to_string(classify_max_read_len()) + " bases");
            } else {
                for (int j = 0; j < kNCls; ++j) {
                    if (!rw->cls_n[j]) continue;
                    ClassifyArgs s = a;
                    s.index = rw->cls_dev[j];
                    s.count = rw->cls_n[j];
                    if (j == 0) tail_launch(s);
                    mark_start();
                    if (!launch_classify(s, cls_len[j], 0, c->stream))
                        return set_err(c, LMAT_E_CAPACITY, "read longer than " + std::to_string(classify_max_read_len()) + " bases");
                }
            }
        } else if (!mixed) {
            setup_tail(true);
            tail_launch(a);
            mark_start();
            if (!launch_classify(a, std::min<uint32_t>(reads->max_len, 512 + k - 1), 0, c->stream))
                return set_err(c, LMAT_E_CAPACITY, "read longer than " + std::to_string(classify_max_read_len()) + " bases");
        } else {
            setup_tail(false);
            for (int j = 0; j < kNCls; ++j) {
                const std::vector<uint32_t>& v = rw->cls_host[j];
                const size_t lo = std::lower_bound(v.begin(), v.end(), (uint32_t)first) - v.begin();
                const size_t hi = std::lower_bound(v.begin(), v.end(), (uint32_t)(first + count)) - v.begin();
                if (hi == lo) continue;
                ClassifyArgs s = a;
                s.index = rw->cls_dev[j] + lo;
                s.count = hi - lo;
                if (j == 0) tail_launch(s);
                mark_start();
                if (!launch_classify(s, cls_len[j], 0, c->stream))
                    return set_err(c, LMAT_E_CAPACITY, "read longer than " + std::to_string(classify_max_read_len()) + " bases");
            }
        }
    }
    mark_start();  // (no class had a read)
    if (timed) {
        HIPCHK(c, hipEventRecord(e1, c->stream));
        HIPCHK(c, hipEventRecord(e2, c->stream));
    }
    {   // reads the fast classes could not hold (length, taxids, list elements), listed on the device: first the fast
        // kernel with room for 512 list elements -- reads over k-mers shared by many strains --, whose own leftovers go on
        ClassifyArgs m = a;
        m.index = c->d_ovf;
        m.count_ptr = c->d_cursor + 2;
        m.count = 0;
        m.ovf_list = c->d_ovf2;
        m.ovf_slot = 3;
        m.p_max = 160;   // two launches over the one list: reads of up to 160 k-mers in the leaner variant, the rest in the 512 one
        launch_classify(m, 160 + (uint32_t)c->dev.k - 1, 2, c->stream);
        m.p_min = 161;
        m.p_max = 0xFFFFFFFFu;
        launch_classify(m, 512 + (uint32_t)c->dev.k - 1, 2, c->stream);
    }
    const bool k4 = a.prm.stop_after == 0 || a.prm.stop_after >= 10;  // 10..12: partial decision steps (timing experiments)
    if (k4) {  // score + LCA decision, one lane per read
        if (!c->stream2) {
            HIPCHK(c, hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
        }
        if (!c->stream3) {
            HIPCHK(c, hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_join3, hipEventDisableTiming));
        }
        launch_k4_begin(a, c->stream, c->stream2, c->stream3, pipelined ? c->stream3 : c->stream, c->ev_fork);
    }
    {   // ... to the large LDS class (1024 taxids, reads up to 2067 bp), or directly to the global-memory class when the
        // batch holds reads beyond that.  These kernels make their own decision step, so they run beside the K4 kernels.
        static const bool rerun_side = getenv("LMAT_RERUN_SIDE") && atoi(getenv("LMAT_RERUN_SIDE")) != 0;  // experiments
        hipStream_t rs = k4 && (pipelined || rerun_side) ? c->stream3 : c->stream;
        const bool lds_class = reads->max_len <= 2048 + 19;
        static const bool mid_on = !getenv("LMAT_MID_TIER") || atoi(getenv("LMAT_MID_TIER")) != 0;  // (0: as before this tier existed, for A/B runs)
        const bool mid_tier = mid_on && reads->max_len <= 512 + (uint32_t)c->dev.k - 1;
        ClassifyArgs b = a;
        b.index = c->d_ovf2;
        b.count_ptr = c->d_cursor + 3;
        b.count = 0;
        b.gscratch = c->d_gscratch;
        if (mid_tier) {  // first the middle tier (256 taxids, 1024 list elements; four waves per CU), whose leftovers go on
            b.ovf_list = c->d_ovf4;
            b.ovf_slot = 10;
            launch_classify(b, reads->max_len, 3, rs);
            b.index = c->d_ovf4;
            b.count_ptr = c->d_cursor + 10;
        }
        b.ovf_list = lds_class ? c->d_ovf3 : nullptr;  // the global-memory class is the last resort
        b.ovf_slot = 7;
        launch_classify(b, reads->max_len, 1, rs);
        if (lds_class) {  // what even that class cannot hold goes to the global-memory class
            ClassifyArgs g = b;
            g.index = c->d_ovf3;
            g.count_ptr = c->d_cursor + 7;
            g.ovf_list = nullptr;
            launch_classify(g, 2048 + 20, 1, rs);
        }
    }
    // the tiers join on the context's stream, or, when the next batch is to run beside them, on the second stream
    hipStream_t js = k4 && pipelined ? c->stream2 : c->stream;
    c->join_stream = js;
    if (k4) {
        if (!c->ev_join_small) HIPCHK(c, hipEventCreateWithFlags(&c->ev_join_small, hipEventDisableTiming));
        launch_k4_end(a, js, c->stream2, c->stream3, pipelined ? c->stream3 : c->stream, c->ev_join, c->ev_join3, c->ev_join_small, c->ev_done);
    } else HIPCHK(c, hipEventRecord(c->ev_done, c->stream));
    c->set_in_flight = true;
    if (timed) {
        HIPCHK(c, hipEventRecord(e3, js));
        c->pending_events.push_back(std::make_pair(e0, e1));
        c->pending_events2.push_back(std::make_pair(e2, e3));
        e0 = e1 = e2 = e3 = nullptr;
    }
    HIPCHK(c, hipGetLastError());
    return LMAT_OK;
}

// Reads and clears the sticky error word; maps it to the API's codes (the most specific message wins).
static int report_device_errors(lmat_ctx* c, uint32_t* cand_cursor) {
    uint32_t cur[2];
    HIPCHK(c, hipMemcpy(&cur[0], c->d_cursor, 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(&cur[1], c->d_err, 4, hipMemcpyDeviceToHost));
    if (cand_cursor) *cand_cursor = cur[0];
    if (!cur[1]) return LMAT_OK;
    HIPCHK(c, hipMemset(c->d_err, 0, 4));
    if (cur[1] & kErrReadTooLong) return set_err(c, LMAT_E_CAPACITY, "read longer than the kernel's k-mer capacity");
    if (cur[1] & kErrLineageTrunc) return set_err(c, LMAT_E_CAPACITY, "taxonomy deeper than the lineage scratch (72 levels)");
    if (cur[1] & kErrNoNullModel) return set_err(c, LMAT_E_TAXONOMY, "ERROR, ALL TAXIDS MUST HAVE NULL MODELS");
    if (cur[1] & kErrTidOverflow)
        return set_err(c, LMAT_E_CAPACITY, "a read exceeds the largest tables (4096 taxids / 16384 list elements)");
    if (cur[1] & kErrCandOverflow) return set_err(c, LMAT_E_CAPACITY, "candidate buffer too small (cand_cap)");
    return set_err(c, LMAT_E_DEVICE, "unknown device error flag");
}

int lmat_classify(lmat_ctx* c, const lmat_reads* reads, uint64_t first, uint64_t count, lmat_read_result* results,
                  lmat_cand* cands, uint64_t cand_cap, uint64_t* n_cands) {
    if (!c || !reads || (count && !results)) return LMAT_E_ARG;
    if (!count) { if (n_cands) *n_cands = 0; return LMAT_OK; }
    const bool want = cands != nullptr && cand_cap > 0;
    hipSetDevice(c->device);
    // a launch that ends in an error (say, cand_cap too small) has still tallied its reads: keep a copy of the tallies
    // so that the caller's retry does not count the batch twice
    if (c->d_counts && !c->d_counts_bak) HIPCHK(c, hipMalloc(&c->d_counts_bak, c->counts_bytes));
    if (c->d_counts) HIPCHK(c, hipMemcpyAsync(c->d_counts_bak, c->d_counts, c->counts_bytes, hipMemcpyDeviceToDevice, c->stream));
    int rc = run_classify(c, reads, first, count, want, cand_cap, false);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    uint32_t cand_cursor = 0;
    rc = report_device_errors(c, &cand_cursor);
    if (rc) {
        if (c->d_counts) HIPCHK(c, hipMemcpy(c->d_counts, c->d_counts_bak, c->counts_bytes, hipMemcpyDeviceToDevice));
        return rc;
    }
    HIPCHK(c, hipMemcpy(results, c->d_results, count * sizeof(lmat_read_result), hipMemcpyDeviceToHost));
    if (want) {
        const uint64_t used = std::min<uint64_t>(cand_cursor, cand_cap);
        if (used) HIPCHK(c, hipMemcpy(cands, c->d_cands, used * sizeof(lmat_cand), hipMemcpyDeviceToHost));
        if (n_cands) *n_cands = used;
    } else if (n_cands) *n_cands = 0;
    return LMAT_OK;
}

int lmat_classify_async(lmat_ctx* c, const lmat_reads* reads, uint64_t first, uint64_t count) {
    if (!c || !reads) return LMAT_E_ARG;
    if (!count) return LMAT_OK;
    return run_classify(c, reads, first, count, false, 0, true, pipeline_on());
}

int lmat_classify_async_cands(lmat_ctx* c, const lmat_reads* reads, uint64_t first, uint64_t count, uint64_t cand_cap) {
    if (!c || !reads) return LMAT_E_ARG;
    if (!count) return LMAT_OK;
    return run_classify(c, reads, first, count, cand_cap != 0, cand_cap, true, pipeline_on());
}

int lmat_sync(lmat_ctx* c, float* kernel_ms_total, uint64_t* kernel_launches) {
    if (!c) return LMAT_E_ARG;
    hipSetDevice(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->set_in_flight) { HIPCHK(c, hipEventSynchronize(c->ev_done)); c->set_in_flight = false; }  // decision kernels on the side streams
    if (c->parked.in_flight) { HIPCHK(c, hipEventSynchronize(c->parked.done)); c->parked.in_flight = false; }
    if (c->stream2) HIPCHK(c, hipStreamSynchronize(c->stream2));  // the timing events behind them
    for (auto& e : c->pending_events) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) { c->kernel_ms_total += ms; c->kernel_launches++; }
        hipEventDestroy(e.first);
        hipEventDestroy(e.second);
    }
    c->pending_events.clear();
    for (auto& e : c->pending_events2) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) c->kernel2_ms_total += ms;
        hipEventDestroy(e.first);
        hipEventDestroy(e.second);
    }
    c->pending_events2.clear();
    if (kernel_ms_total) *kernel_ms_total = c->kernel_ms_total + c->kernel2_ms_total;
    if (kernel_launches) *kernel_launches = c->kernel_launches;
    c->last_classify_ms = c->kernel_ms_total;
    c->last_decide_ms = c->kernel2_ms_total;
    c->last_launches = c->kernel_launches;
    c->kernel_ms_total = 0;
    c->kernel2_ms_total = 0;
    c->kernel_launches = 0;
    if (getenv("LMAT_DEBUG")) {
        uint32_t cur[16];
        uint32_t err = 0;
        HIPCHK(c, hipMemcpy(cur, c->d_cursor, 64, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(&err, c->d_err, 4, hipMemcpyDeviceToHost));
        fprintf(stderr, "[lmat] last launch: cand cursor %u, error flags (all launches since the last report) %u; reads passed on by the fast classes %u, "
                        "by the E=512 class %u, by the middle tier (T=256) %u, by the large LDS class %u; general decision path: small tables %u, large %u\n",
                cur[0], err, cur[2], cur[3], cur[10], cur[7], cur[4], cur[5]);
    }
    return report_device_errors(c, nullptr);  // flags of every launch since the last report, not just the last one
}

int lmat_set_label_modes(lmat_ctx* c, int permissive, int tid_cutoff, const char* rank_map_fn) {
    if (!c) return LMAT_E_ARG;
    if (c->db_ready) return set_err(c, LMAT_E_ARG, "label modes shape the database records: set them before lmat_db_finalize");
    c->permissive = permissive ? 1 : 0;
    c->rt_tid_cut = tid_cutoff > 0 ? tid_cutoff : 0;
    c->rt_rank_map.clear();
    if (tid_cutoff > 0 && rank_map_fn && *rank_map_fn) {  // read_label.cpp:1543-1557
        FILE* f = fopen(rank_map_fn, "r");
        if (!f) return set_err(c, LMAT_E_IO, std::string("cannot read rank map ") + rank_map_fn);
        int s, d;
        while (fscanf(f, "%d%d", &s, &d) > 0) c->rt_rank_map[(uint32_t)s] = (uint32_t)d;
        fclose(f);
    }
    return LMAT_OK;
}

// ---- rand_read_label (src/rand_read_label.cpp): the null-model generator on the same kernels ------------------
int lmat_rand_mode(lmat_ctx* c, int on) {
    if (!c) return LMAT_E_ARG;
    if (c->db_ready) return set_err(c, LMAT_E_ARG, "rand mode shapes the database records: set it before lmat_db_finalize");
    c->rand_mode = on ? 1 : 0;
    return LMAT_OK;
}
int lmat_rand_reset(lmat_ctx* c, uint32_t n_buckets) {
    if (!c || !n_buckets || n_buckets > 255) return LMAT_E_ARG;
    if (!c->tax.loaded) return set_err(c, LMAT_E_ARG, "load the taxonomy first");
    hipSetDevice(c->device);
    if (c->d_rand_max) { hipFree(c->d_rand_max); c->d_rand_max = nullptr; }
    if (c->d_rand_cnt) { hipFree(c->d_rand_cnt); c->d_rand_cnt = nullptr; }
    const size_t bytes = (size_t)c->dev.n_ids * n_buckets * 4;
    HIPCHK(c, hipMalloc((void**)&c->d_rand_max, bytes));
    HIPCHK(c, hipMalloc((void**)&c->d_rand_cnt, bytes));
    HIPCHK(c, hipMemset(c->d_rand_max, 0, bytes));
    HIPCHK(c, hipMemset(c->d_rand_cnt, 0, bytes));
    c->rand_nb = n_buckets;
    return LMAT_OK;
}
int lmat_rand_label(lmat_ctx* c, const lmat_reads* reads, uint64_t first, uint64_t count, const uint8_t* gc_bucket) {
    if (!c || !reads || (count && !gc_bucket)) return LMAT_E_ARG;
    if (!c->rand_mode) return set_err(c, LMAT_E_ARG, "lmat_rand_mode(ctx, 1) before the database is finalized");
    if (!c->d_rand_max) return set_err(c, LMAT_E_ARG, "lmat_rand_reset first");
    if (!count) return LMAT_OK;
    hipSetDevice(c->device);
    for (uint64_t i = 0; i < count; ++i)
        if (gc_bucket[i] >= c->rand_nb) return set_err(c, LMAT_E_ARG, "GC bucket out of range");
    if (count > c->rand_gc_cap) {
        if (c->d_rand_gc) hipFree(c->d_rand_gc);
        c->d_rand_gc = nullptr;
        HIPCHK(c, hipMalloc((void**)&c->d_rand_gc, count));
        c->rand_gc_cap = count;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_rand_gc, gc_bucket, count, hipMemcpyHostToDevice, c->stream));
    const lmat_params keep = c->params;
    c->params.min_kmer = 1;      // "if (valid_kmers > 0)", rand_read_label.cpp:382
    c->params.min_fnd_kmer = 0;
    c->rand_launch = true;
    int rc = run_classify(c, reads, first, count, false, 0, false);
    c->rand_launch = false;
    c->params = keep;
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return report_device_errors(c, nullptr);
}
int lmat_rand_get(lmat_ctx* c, uint32_t* tid32, float* max_prob, uint32_t* cnt, uint32_t cap, uint32_t* n_rows) {
    if (!c || !n_rows) return LMAT_E_ARG;
    if (!c->d_rand_max) return set_err(c, LMAT_E_ARG, "lmat_rand_reset first");
    hipSetDevice(c->device);
    const uint32_t nb = c->rand_nb, n_ids = c->dev.n_ids;
    std::vector<uint32_t> mx((size_t)n_ids * nb), ct((size_t)n_ids * nb);
    HIPCHK(c, hipMemcpy(mx.data(), c->d_rand_max, mx.size() * 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(ct.data(), c->d_rand_cnt, ct.size() * 4, hipMemcpyDeviceToHost));
    uint32_t n = 0;
    for (uint32_t i = 1; i < n_ids; ++i) {  // internal index order = ascending taxid = the std::map order of the output file
        bool any = false;
        for (uint32_t b = 0; b < nb; ++b) any |= ct[(size_t)i * nb + b] != 0;
        if (!any) continue;
        if (n < cap && tid32 && max_prob && cnt) {
            tid32[n] = c->tax.tid32[i];
            for (uint32_t b = 0; b < nb; ++b) {
                memcpy(&max_prob[(size_t)n * nb + b], &mx[(size_t)i * nb + b], 4);
                cnt[(size_t)n * nb + b] = ct[(size_t)i * nb + b];
            }
        }
        ++n;
    }
    *n_rows = n;
    return LMAT_OK;
}

int lmat_nullmodel_load(lmat_ctx* c, const char* list_fn) {
    if (!c || !list_fn) return LMAT_E_ARG;
    hipSetDevice(c->device);
    int rc = load_null_models(c, list_fn);
    if (rc == LMAT_OK && c->nm.n_cls > 64) {
        free_null_models(c);
        return set_err(c, LMAT_E_CAPACITY, "more than 64 distinct null-model class strings");
    }
    return rc;
}
int lmat_nullmodel_clear(lmat_ctx* c) {
    if (!c) return LMAT_E_ARG;
    hipSetDevice(c->device);
    free_null_models(c);
    return LMAT_OK;
}

// Test hook: the decision kernels' code on candidate tables given from outside (include/lmat_hip.h).
int lmat_debug_decide(lmat_ctx* c, const uint32_t* tids, const float* scores, const uint64_t* off, const float* stdev, uint64_t n,
                      lmat_read_result* results) {
    if (!c || !tids || !scores || !off || !stdev || !results) return LMAT_E_ARG;
    if (!c->tax.loaded) return set_err(c, LMAT_E_ARG, "load the taxonomy first");
    if (c->tax.wide) return set_err(c, LMAT_E_ARG, "the debug entries of the decision step take taxonomies of up to 65534 ids");
    if (!n) return LMAT_OK;
    hipSetDevice(c->device);
    const uint64_t total = off[n];
    std::vector<uint32_t> idx(total);
    for (uint64_t i = 0; i < total; ++i) {
        auto it = c->tax.index_of.find(tids[i]);
        if (it == c->tax.index_of.end()) return set_err(c, LMAT_E_TAXONOMY, "taxid " + std::to_string(tids[i]) + " is not in the taxonomy");
        idx[i] = it->second;
    }
    uint32_t* d_idx = nullptr; float *d_sc = nullptr, *d_sd = nullptr; uint64_t* d_off = nullptr; lmat_read_result* d_res = nullptr;
    HIPCHK(c, hipMalloc((void**)&d_idx, std::max<uint64_t>(total, 1) * 4));
    HIPCHK(c, hipMalloc((void**)&d_sc, std::max<uint64_t>(total, 1) * 4));
    HIPCHK(c, hipMalloc((void**)&d_sd, n * 4));
    HIPCHK(c, hipMalloc((void**)&d_off, (n + 1) * 8));
    HIPCHK(c, hipMalloc((void**)&d_res, n * sizeof(lmat_read_result)));
    HIPCHK(c, hipMemcpyAsync(d_idx, idx.data(), total * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_sc, scores, total * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_sd, stdev, n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_off, off, (n + 1) * 8, hipMemcpyHostToDevice, c->stream));
    ClassifyArgs a;
    a.tb = c->dev;
    a.prm = kparams(c->params);
    a.prm.stop_after = 0;
    a.results = d_res;
    a.nm = NullModelDev();
    launch_k4_debug(a, d_idx, d_sc, d_off, d_sd, n, c->stream);
    HIPCHK(c, hipMemcpyAsync(results, d_res, n * sizeof(lmat_read_result), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    hipFree(d_idx); hipFree(d_sc); hipFree(d_sd); hipFree(d_off); hipFree(d_res);
    for (uint64_t i = 0; i < n; ++i)
        if (results[i].status == 255) return set_err(c, LMAT_E_CAPACITY, "a candidate table outside 1..64 entries, or a lineage beyond the scratch");
    return LMAT_OK;
}

// The decision step on (taxid, count) tables: on_the_wave = 1 runs k4_wave -- the code the benchmarked path runs on the classify
// wave --, 0 the general path (k4_part1 / k4_part2, the one lmat_debug_decide pins to the reference's printed records).
int lmat_debug_decide_counts(lmat_ctx* c, const uint32_t* tids, const uint32_t* counts, const uint64_t* off, const uint32_t* cand, uint64_t n,
                             int on_the_wave, lmat_read_result* results) {
    if (!c || !tids || !counts || !off || !cand || !results) return LMAT_E_ARG;
    if (!c->tax.loaded) return set_err(c, LMAT_E_ARG, "load the taxonomy first");
    if (c->tax.wide) return set_err(c, LMAT_E_ARG, "the debug entries of the decision step take taxonomies of up to 65534 ids");
    if (!n) return LMAT_OK;
    hipSetDevice(c->device);
    const uint64_t total = off[n];
    std::vector<uint32_t> idx(total);
    for (uint64_t i = 0; i < total; ++i) {
        auto it = c->tax.index_of.find(tids[i]);
        if (it == c->tax.index_of.end()) return set_err(c, LMAT_E_TAXONOMY, "taxid " + std::to_string(tids[i]) + " is not in the taxonomy");
        idx[i] = it->second;
        if (counts[i] > 0xFFFFu) return set_err(c, LMAT_E_ARG, "a count above 65535");
    }
    uint32_t *d_idx = nullptr, *d_cn = nullptr, *d_cd = nullptr; uint64_t* d_off = nullptr; lmat_read_result* d_res = nullptr; void* d_tally = nullptr;
    HIPCHK(c, hipMalloc((void**)&d_idx, std::max<uint64_t>(total, 1) * 4));
    HIPCHK(c, hipMalloc((void**)&d_cn, std::max<uint64_t>(total, 1) * 4));
    HIPCHK(c, hipMalloc((void**)&d_cd, n * 4));
    HIPCHK(c, hipMalloc((void**)&d_off, (n + 1) * 8));
    HIPCHK(c, hipMalloc((void**)&d_res, n * sizeof(lmat_read_result)));
    HIPCHK(c, hipMalloc(&d_tally, c->counts_bytes));   // k4_wave tallies its calls: into a buffer nobody reads
    HIPCHK(c, hipMemsetAsync(d_tally, 0, c->counts_bytes, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_idx, idx.data(), total * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_cn, counts, total * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_cd, cand, n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_off, off, (n + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_cursor, 0, kCursorWords * 4, c->stream));
    ClassifyArgs a;
    a.tb = c->dev;
    a.prm = kparams(c->params);
    a.prm.stop_after = 0;
    a.results = d_res;
    a.cands = nullptr;
    a.cand_cap = 0;
    a.cursor = c->d_cursor;
    a.err = c->d_err;
    a.counts = d_tally;
    auto it = c->tax.index_of.find(32630);
    a.phix_call_idx = it == c->tax.index_of.end() ? 0 : it->second;
    a.nm = NullModelDev();
    launch_k4_debug_counts(a, d_idx, d_cn, d_off, d_cd, n, on_the_wave != 0, c->stream);
    HIPCHK(c, hipMemcpyAsync(results, d_res, n * sizeof(lmat_read_result), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    hipFree(d_idx); hipFree(d_cn); hipFree(d_cd); hipFree(d_off); hipFree(d_res); hipFree(d_tally);
    return LMAT_OK;
}

// Measurement hook: the per-launch counter block of the most recent launch (after lmat_sync): [0] candidate cursor, [2] reads the
// fast classes passed on, [3] reads the E = 512 class passed on, [10] the middle tier, [7] the large LDS class, [4] / [5] / [8] / [9]
// reads handed to the general decision path by table size.
int lmat_debug_last_counters(lmat_ctx* c, uint32_t* out16) {
    if (!c || !out16) return LMAT_E_ARG;
    hipSetDevice(c->device);
    HIPCHK(c, hipMemcpy(out16, c->d_cursor, kCursorWords * 4, hipMemcpyDeviceToHost));
    return LMAT_OK;
}

int lmat_last_timing(const lmat_ctx* c, float* classify_ms, float* decide_ms, uint64_t* launches) {
    if (!c) return LMAT_E_ARG;
    if (classify_ms) *classify_ms = c->last_classify_ms;
    if (decide_ms) *decide_ms = c->last_decide_ms;
    if (launches) *launches = c->last_launches;
    return LMAT_OK;
}

int lmat_results_fetch(lmat_ctx* c, uint64_t first, uint64_t count, lmat_read_result* results) {
    if (!c || !results || first + count > c->results_cap) return LMAT_E_ARG;
    hipSetDevice(c->device);
    HIPCHK(c, hipMemcpy(results, c->d_results + first, count * sizeof(lmat_read_result), hipMemcpyDeviceToHost));
    return LMAT_OK;
}

// ---------------------------------------------------------------------------------- tallies
int lmat_counts_reset(lmat_ctx* c) {
    if (!c || !c->d_counts) return LMAT_E_ARG;
    hipSetDevice(c->device);
    HIPCHK(c, hipMemsetAsync(c->d_counts, 0, c->counts_bytes, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return LMAT_OK;
}
int lmat_counts_layout(const lmat_ctx* c, uint32_t* n_ids, uint64_t* bytes) {
    if (!c) return LMAT_E_ARG;
    if (n_ids) *n_ids = c->dev.n_ids;
    if (bytes) *bytes = c->counts_bytes;
    return LMAT_OK;
}
void* lmat_counts_device_ptr(lmat_ctx* c) { return c ? c->d_counts : nullptr; }

int lmat_counts_get(lmat_ctx* c, uint32_t* tid32, uint64_t* count, double* score, uint32_t cap, uint32_t* n_nonzero,
                    uint64_t nomatch3[3]) {
    if (!c || !c->d_counts) return LMAT_E_ARG;
    hipSetDevice(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<unsigned char> buf(c->counts_bytes);
    HIPCHK(c, hipMemcpy(buf.data(), c->d_counts, c->counts_bytes, hipMemcpyDeviceToHost));
    const uint32_t n = c->dev.n_ids;
    const uint64_t* cn = (const uint64_t*)buf.data();
    const double* sc = (const double*)(cn + n);
    const uint64_t* nm = (const uint64_t*)(sc + n);
    uint32_t k = 0;
    for (uint32_t i = 0; i < n; ++i)
        if (cn[i]) {
            if (k < cap) {
                if (tid32) tid32[k] = c->tax.tid32[i];
                if (count) count[k] = cn[i];
                if (score) score[k] = sc[i];
            }
            ++k;
        }
    if (n_nonzero) *n_nonzero = k;
    if (nomatch3) { nomatch3[0] = nm[0]; nomatch3[1] = nm[1]; nomatch3[2] = nm[2]; }
    return LMAT_OK;
}

int lmat_gather_bench(lmat_ctx* c, uint64_t n_probes, uint64_t seed, float* ms, uint64_t* bytes) {
    if (!c || !c->db_ready) return set_err(c, LMAT_E_ARG, "database not ready");
    hipSetDevice(c->device);
    unsigned long long* sink = nullptr;
    HIPCHK(c, hipMalloc((void**)&sink, 8));
    HIPCHK(c, hipMemsetAsync(sink, 0, 8, c->stream));
    hipEvent_t e0, e1;
    HIPCHK(c, hipEventCreate(&e0));
    HIPCHK(c, hipEventCreate(&e1));
    HIPCHK(c, hipEventRecord(e0, c->stream));
    const int bpp = getenv("LMAT_GATHER_BYTES") && atoi(getenv("LMAT_GATHER_BYTES")) == 128 ? 128 : 64;  // experiment: aligned 128-byte requests
    launch_gather_bench(c->dev.slots, c->dev.nbuckets, n_probes, seed, sink, c->stream, bpp);
    HIPCHK(c, hipEventRecord(e1, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float t = 0;
    HIPCHK(c, hipEventElapsedTime(&t, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1); hipFree(sink);
    if (ms) *ms = t;
    const uint64_t per_wave = bpp == 128 ? (n_probes / 4096 + 71) / 72 * 72 : (n_probes / 4096 + 143) / 144 * 144;
    if (bytes) *bytes = per_wave * 4096 * (uint64_t)bpp;
    return LMAT_OK;
}

// ---------------------------------------------------------------------------------- streamed boundary
// A ring of batch slots with everything a batch needs allocated once: pinned host buffers (ASCII in, results out) and
// their device twins.  H2D, pack + classify and D2H of consecutive batches overlap on three HIP streams, which is the
// overlap upstream gets from its reader thread and (read, hdr) queue (src/read_label.cpp:1651-1746).  Nothing is
// allocated or freed on the per-batch path.
struct lmat_stream {
    struct Slot {
        uint8_t* h_bases = nullptr;  uint64_t* h_off = nullptr;      // pinned: filled by the caller
        uint64_t* h_rec_off = nullptr; uint32_t* h_cls[kNCls] = {nullptr, nullptr, nullptr, nullptr};
        lmat_read_result* h_results = nullptr; lmat_cand* h_cands = nullptr; uint32_t* h_cursor = nullptr;  // pinned: filled by the engine
        uint8_t* d_bases = nullptr; uint64_t* d_off = nullptr;
        lmat_read_result* d_results = nullptr; lmat_cand* d_cands = nullptr;
        uint64_t cand_cap = 0;
        lmat_reads reads;            // packed records + class lists (device)
        hipEvent_t ev_up = nullptr, ev_done = nullptr, ev_out = nullptr;
        uint64_t n = 0, tag = 0;
        int state = 0;               // 0 free, 1 acquired, 2 in flight, 3 handed to the caller
    };
    lmat_ctx* c = nullptr;
    std::vector<Slot> slots;
    uint64_t max_reads = 0, max_bases = 0;
    uint32_t cands_per_read = 0;
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    uint64_t head = 0, tail = 0;     // slot of the next acquire / of the oldest batch in flight (monotonic counters)
    void* d_scratch_counts = nullptr;  // tallies of a re-run go nowhere
};

static void stream_free(lmat_stream* st) {
    if (!st) return;
    hipSetDevice(st->c->device);
    hipDeviceSynchronize();
    for (auto& sl : st->slots) {
        void* pinned[] = {sl.h_bases, sl.h_off, sl.h_rec_off, sl.h_cls[0], sl.h_cls[1], sl.h_cls[2], sl.h_cls[3], sl.h_results, sl.h_cands, sl.h_cursor};
        for (void* p : pinned) if (p) hipHostFree(p);
        void* dev[] = {sl.d_bases, sl.d_off, sl.d_results, sl.d_cands, sl.reads.words, sl.reads.rec_off, sl.reads.cls_dev[0], sl.reads.cls_dev[1], sl.reads.cls_dev[2], sl.reads.cls_dev[3]};
        for (void* p : dev) if (p) hipFree(p);
        sl.reads.words = nullptr; sl.reads.rec_off = nullptr;
        for (int j = 0; j < kNCls; ++j) sl.reads.cls_dev[j] = nullptr;
        if (sl.ev_up) hipEventDestroy(sl.ev_up);
        if (sl.ev_done) hipEventDestroy(sl.ev_done);
        if (sl.ev_out) hipEventDestroy(sl.ev_out);
    }
    if (st->d_scratch_counts) hipFree(st->d_scratch_counts);
    if (st->s_h2d) hipStreamDestroy(st->s_h2d);
    delete st;
}

int lmat_stream_create(lmat_ctx* c, uint64_t max_reads, uint64_t max_bases, uint32_t cands_per_read, int n_slots, lmat_stream** out) {
    if (!c || !out || !max_reads || !max_bases || n_slots < 1 || n_slots > 8) return LMAT_E_ARG;
    *out = nullptr;
    if (!c->db_ready) return set_err(c, LMAT_E_ARG, "database not ready");
    if (max_reads > 0x7FFFFFFFull) return set_err(c, LMAT_E_ARG, "batch above 2^31 reads");
    hipSetDevice(c->device);
    lmat_stream* st = new lmat_stream();
    st->c = c;
    st->max_reads = max_reads;
    st->max_bases = max_bases;
    st->cands_per_read = cands_per_read;
    st->slots.resize(n_slots);
    if (!c->stream2) {  // also the side stream of the K4 kernels (run_classify)
        if (hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) { delete st; return set_err(c, LMAT_E_DEVICE, "cannot create a stream"); }
    }
    st->s_d2h = c->stream2;
    bool ok = hipStreamCreateWithFlags(&st->s_h2d, hipStreamNonBlocking) == hipSuccess &&
              hipMalloc(&st->d_scratch_counts, c->counts_bytes) == hipSuccess;
    const uint64_t max_words = max_bases / 16 + max_bases / 32 + 3 * max_reads + 16;  // rec_words summed, rounded up per read
    for (auto& sl : st->slots) {
        if (!ok) break;
        sl.cand_cap = cands_per_read ? (uint64_t)cands_per_read * max_reads + (uint64_t)kCandSubs * 2048 : 0;  // (+ what the sub-cursors may leave unused: kernels.hpp)
        ok = hipHostMalloc((void**)&sl.h_bases, max_bases + 16, hipHostMallocDefault) == hipSuccess &&
             hipHostMalloc((void**)&sl.h_off, (max_reads + 1) * 8, hipHostMallocDefault) == hipSuccess &&
             hipHostMalloc((void**)&sl.h_rec_off, (max_reads + 1) * 8, hipHostMallocDefault) == hipSuccess &&
             hipHostMalloc((void**)&sl.h_results, max_reads * sizeof(lmat_read_result), hipHostMallocDefault) == hipSuccess &&
             hipHostMalloc((void**)&sl.h_cursor, 64, hipHostMallocDefault) == hipSuccess &&
             hipMalloc((void**)&sl.d_bases, max_bases + 16) == hipSuccess && hipMalloc((void**)&sl.d_off, (max_reads + 1) * 8) == hipSuccess &&
             hipMalloc((void**)&sl.d_results, max_reads * sizeof(lmat_read_result)) == hipSuccess &&
             hipMalloc((void**)&sl.reads.words, max_words * 4 + 16384) == hipSuccess &&  // kernels read whole-record tiles: pad
             hipMalloc((void**)&sl.reads.rec_off, (max_reads + 1) * 8) == hipSuccess &&
             hipEventCreateWithFlags(&sl.ev_up, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&sl.ev_out, hipEventDisableTiming) == hipSuccess;
        for (int j = 0; j < kNCls && ok; ++j)
            ok = hipHostMalloc((void**)&sl.h_cls[j], max_reads * 4, hipHostMallocDefault) == hipSuccess &&
                 hipMalloc((void**)&sl.reads.cls_dev[j], max_reads * 4) == hipSuccess;
        if (ok && sl.cand_cap)
            ok = hipHostMalloc((void**)&sl.h_cands, sl.cand_cap * sizeof(lmat_cand), hipHostMallocDefault) == hipSuccess &&
                 hipMalloc((void**)&sl.d_cands, sl.cand_cap * sizeof(lmat_cand)) == hipSuccess;
        sl.reads.preset = true;
    }
    if (ok) ok = ensure_scratch(c, max_reads) == LMAT_OK;
    if (!ok) { stream_free(st); return set_err(c, LMAT_E_NOMEM, "out of (pinned or device) memory for the batch ring"); }
    *out = st;
    return LMAT_OK;
}

void lmat_stream_destroy(lmat_stream* st) { stream_free(st); }

int lmat_stream_acquire(lmat_stream* st, uint8_t** bases, uint64_t** off) {
    if (!st || !bases || !off) return LMAT_E_ARG;
    lmat_stream::Slot& sl = st->slots[st->head % st->slots.size()];
    if (sl.state != 0) return set_err(st->c, LMAT_E_ARG, "every slot is in use: take (lmat_stream_next) and release the oldest batch first");
    sl.state = 1;
    *bases = sl.h_bases;
    *off = sl.h_off;
    return LMAT_OK;
}

static int stream_launch(lmat_stream* st, lmat_stream::Slot& sl, bool to_scratch_counts) {
    lmat_ctx* c = st->c;
    c->out_results = sl.d_results;
    c->out_cands = sl.d_cands;
    c->out_counts = to_scratch_counts ? st->d_scratch_counts : nullptr;
    c->batch_err = true;
    const int rc = run_classify(c, &sl.reads, 0, sl.n, sl.cand_cap != 0, sl.cand_cap, false, pipeline_on());
    c->out_results = nullptr; c->out_cands = nullptr; c->out_counts = nullptr; c->batch_err = false;
    if (rc) return rc;
    // The counters are copied out on the stream the batch's kernels joined on, and the set's `done` event is recorded again
    // behind that copy: the next launch that takes this set of buffers clears the counters, and must not overtake it.
    // The results ride the context's second stream (idle once the decision kernels have joined; with LMAT_PIPELINE it is
    // the joining stream itself).  A stream of its own for them would be the fifth of the process: the runtime multiplexes
    // streams onto 4 hardware queues, and a copy that shares its queue with the compute stream holds up the next batch's
    // kernels behind it.
    hipStream_t js = c->join_stream;
    HIPCHK(c, hipMemcpyAsync(sl.h_cursor, c->d_cursor, 4, hipMemcpyDeviceToHost, js));
    HIPCHK(c, hipMemcpyAsync(sl.h_cursor + 1, c->d_cursor + 15, 4, hipMemcpyDeviceToHost, js));  // this batch's own error flags
    HIPCHK(c, hipEventRecord(c->ev_done, js));
    HIPCHK(c, hipEventRecord(sl.ev_done, js));
    if (st->s_d2h != js) HIPCHK(c, hipStreamWaitEvent(st->s_d2h, sl.ev_done, 0));
    HIPCHK(c, hipMemcpyAsync(sl.h_results, sl.d_results, sl.n * sizeof(lmat_read_result), hipMemcpyDeviceToHost, st->s_d2h));
    HIPCHK(c, hipEventRecord(sl.ev_out, st->s_d2h));
    return LMAT_OK;
}

// queues the acquired slot; ext_bases != nullptr: the reads are ext_off[0..n] (any base offset) inside the caller's own
// pinned buffer ext_bases instead of the slot's input buffers
static int stream_submit(lmat_stream* st, uint64_t n, uint64_t tag, const uint8_t* ext_bases, const uint64_t* ext_off) {
    if (!st) return LMAT_E_ARG;
    static const bool dbg = getenv("LMAT_DEBUG_STREAM") != nullptr;
    const auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (dbg) fprintf(stderr, "[stream] %s at %.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - T0).count()); };
    lmat_ctx* c = st->c;
    lmat_stream::Slot& sl = st->slots[st->head % st->slots.size()];
    if (sl.state != 1) return set_err(c, LMAT_E_ARG, "lmat_stream_acquire first");
    if (ext_bases) {
        if (n > st->max_reads) { sl.state = 0; return set_err(c, LMAT_E_ARG, "batch larger than the stream was created for"); }
        const uint64_t base = n ? ext_off[0] : 0;
        for (uint64_t i = 0; i <= n; ++i) sl.h_off[i] = ext_off[i] - base;
        ext_bases += base;
    }
    if (n > st->max_reads || (n && sl.h_off[n] > st->max_bases)) { sl.state = 0; return set_err(c, LMAT_E_ARG, "batch larger than the stream was created for"); }
    hipSetDevice(c->device);
    // record offsets and the length classes of the fast kernels: two passes over the n offsets, split over a few host
    // threads when the batch is large (2 M reads took 12 ms in one thread, longer than the GPU needs for them)
    const uint32_t k = (uint32_t)c->dev.k;
    const int nt = n >= (1u << 18) ? 4 : 1;
    struct Part { uint64_t words = 0, cn[kNCls] = {0, 0, 0, 0}; uint32_t max_len = 0, min_len = 0xFFFFFFFFu; bool bad = false; };
    std::vector<Part> part(nt);
    auto span = [&](int t, uint64_t& lo, uint64_t& hi) { lo = n * (uint64_t)t / nt; hi = n * (uint64_t)(t + 1) / nt; };
    auto pass1 = [&](int t) {
        uint64_t lo, hi;
        span(t, lo, hi);
        Part& p = part[t];
        for (uint64_t i = lo; i < hi; ++i) {
            const uint64_t len = sl.h_off[i + 1] - sl.h_off[i];
            if (sl.h_off[i + 1] < sl.h_off[i] || len > 0x7FFFFFFF) { p.bad = true; return; }
            p.max_len = std::max<uint32_t>(p.max_len, (uint32_t)len);
            p.min_len = std::min<uint32_t>(p.min_len, (uint32_t)len);
            p.words += rec_words((uint32_t)len);
            const uint32_t P = len >= k ? (uint32_t)len - k + 1 : 0;
            p.cn[len_class(P)]++;
        }
    };
    std::vector<uint64_t> wbase(nt + 1, 0), cbase(kNCls * (nt + 1), 0);
    auto pass2 = [&](int t) {
        uint64_t lo, hi;
        span(t, lo, hi);
        uint64_t w = wbase[t], cpos[kNCls] = {cbase[kNCls * t], cbase[kNCls * t + 1], cbase[kNCls * t + 2], cbase[kNCls * t + 3]};
        for (uint64_t i = lo; i < hi; ++i) {
            const uint32_t len = (uint32_t)(sl.h_off[i + 1] - sl.h_off[i]);
            sl.h_rec_off[i] = w;
            w += rec_words(len);
            const uint32_t P = len >= k ? len - k + 1 : 0;
            const int j = len_class(P);
            sl.h_cls[j][cpos[j]++] = (uint32_t)i;
        }
    };
    auto run_parts = [&](auto&& fn) {
        if (nt == 1) { fn(0); return; }
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(fn, t);
        fn(0);
        for (auto& x : th) x.join();
    };
    run_parts(pass1);
    uint32_t max_len = 0, min_len = 0xFFFFFFFFu;
    uint64_t cn[kNCls] = {0, 0, 0, 0};
    for (int t = 0; t < nt; ++t) {
        if (part[t].bad) { sl.state = 0; return set_err(c, LMAT_E_ARG, "offsets must ascend"); }
        max_len = std::max(max_len, part[t].max_len);
        min_len = std::min(min_len, part[t].min_len);
        wbase[t + 1] = wbase[t] + part[t].words;
        for (int j = 0; j < kNCls; ++j) { cbase[kNCls * (t + 1) + j] = cbase[kNCls * t + j] + part[t].cn[j]; cn[j] += part[t].cn[j]; }
    }
    // reads of one length (an untrimmed sequencer run): offsets and record offsets are i * length and i * words -- made on the
    // device, nothing computed here and 16 bytes per read less to copy in
    // (the slot's own offsets may start anywhere -- the general path copies them as given --, i * length holds from 0 only)
    const bool uniform = n > 0 && min_len == max_len && sl.h_off[0] == 0;
    if (!uniform) run_parts(pass2);
    sl.h_rec_off[n] = wbase[nt];
    lap("offsets done");
    sl.n = n;
    sl.tag = tag;
    sl.reads.n = n;
    sl.reads.n_words = sl.h_rec_off[n];
    sl.reads.max_len = max_len;
    sl.reads.class_len = max_len;
    const int used = (cn[0] != 0) + (cn[1] != 0) + (cn[2] != 0) + (cn[3] != 0);
    for (int j = 0; j < kNCls; ++j) sl.reads.cls_n[j] = used > 1 ? cn[j] : 0;
    if ((int)max_len > classify_max_read_len()) {  // refused before anything is queued: the slot goes back untouched
        sl.state = 0;
        return set_err(c, LMAT_E_CAPACITY, "read longer than " + std::to_string(classify_max_read_len()) + " bases");
    }
    if (n) {
        HIPCHK(c, hipMemcpyAsync(sl.d_bases, ext_bases ? ext_bases : sl.h_bases, sl.h_off[n], hipMemcpyHostToDevice, st->s_h2d));
        if (!uniform) {
            HIPCHK(c, hipMemcpyAsync(sl.d_off, sl.h_off, (n + 1) * 8, hipMemcpyHostToDevice, st->s_h2d));
            HIPCHK(c, hipMemcpyAsync(sl.reads.rec_off, sl.h_rec_off, (n + 1) * 8, hipMemcpyHostToDevice, st->s_h2d));
        }
        if (used > 1)
            for (int j = 0; j < kNCls; ++j)
                if (cn[j]) HIPCHK(c, hipMemcpyAsync(sl.reads.cls_dev[j], sl.h_cls[j], cn[j] * 4, hipMemcpyHostToDevice, st->s_h2d));
        HIPCHK(c, hipEventRecord(sl.ev_up, st->s_h2d));
        HIPCHK(c, hipStreamWaitEvent(c->stream, sl.ev_up, 0));
        lap("copies queued");
        if (uniform) launch_fill_offsets(sl.d_off, sl.reads.rec_off, n, max_len, c->stream);
        launch_pack_reads(sl.d_bases, sl.d_off, sl.reads.rec_off, sl.reads.words, n, c->stream);
        const int rc = stream_launch(st, sl, false);
        if (rc) {  // (an allocation failed: the copies and the pack kernel are already queued on this slot's buffers -- let them drain before the slot is handed out again)
            hipEventSynchronize(sl.ev_up);
            hipStreamSynchronize(c->stream);
            sl.state = 0;
            return rc;
        }
        lap("kernels queued");
    }
    sl.state = 2;
    ++st->head;
    return LMAT_OK;
}
int lmat_stream_submit(lmat_stream* st, uint64_t n, uint64_t tag) { return stream_submit(st, n, tag, nullptr, nullptr); }
int lmat_stream_submit_from(lmat_stream* st, const uint8_t* bases, const uint64_t* off, uint64_t n, uint64_t tag) {
    if (!st || (n && (!bases || !off))) return LMAT_E_ARG;
    return stream_submit(st, n, tag, bases ? bases : (const uint8_t*)"", off);
}
int lmat_host_alloc(uint64_t bytes, void** out) {
    if (!out) return LMAT_E_ARG;
    *out = nullptr;
    return hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? LMAT_OK : LMAT_E_NOMEM;
}
void lmat_host_free(void* p) { if (p) hipHostFree(p); }

int lmat_stream_next(lmat_stream* st, const lmat_read_result** results, const lmat_cand** cands, uint64_t* n_reads, uint64_t* n_cands,
                     uint64_t* tag) {
    if (!st) return LMAT_E_ARG;
    lmat_ctx* c = st->c;
    if (st->tail == st->head) return 1;  // nothing in flight
    lmat_stream::Slot& sl = st->slots[st->tail % st->slots.size()];
    if (sl.state != 2) return set_err(c, LMAT_E_ARG, "release the batch returned by the previous lmat_stream_next first");
    hipSetDevice(c->device);
    uint64_t nc = 0;
    if (sl.n) {
        for (;;) {
            HIPCHK(c, hipEventSynchronize(sl.ev_done));
            // the batch's own flags (every launch of the stream writes them into its per-batch counter block): an error
            // belongs to exactly this batch, comes back every time the caller asks for it again, and never leaks into the
            // batches queued behind it
            const uint32_t fresh = sl.h_cursor[1];
            nc = sl.h_cursor[0];
            if (!(fresh & kErrCandOverflow)) {
                if (fresh & kErrReadTooLong) return set_err(c, LMAT_E_CAPACITY, "read longer than the kernel's k-mer capacity");
                if (fresh & kErrLineageTrunc) return set_err(c, LMAT_E_CAPACITY, "taxonomy deeper than the lineage scratch (72 levels)");
                if (fresh & kErrNoNullModel) return set_err(c, LMAT_E_TAXONOMY, "ERROR, ALL TAXIDS MUST HAVE NULL MODELS");
                if (fresh & kErrTidOverflow) return set_err(c, LMAT_E_CAPACITY, "a read exceeds the largest tables (4096 taxids / 16384 list elements)");
                break;
            }
            // the batch printed more candidates than the slot holds: grow this slot fourfold and run it again (its packed reads
            // are still resident; its tallies were already counted, so the re-run's go to a scratch buffer).  The batches queued
            // behind it are left alone -- their flags are their own -- but they use the context's streams: wait for them first.
            HIPCHK(c, hipDeviceSynchronize());
            hipFree(sl.d_cands); hipHostFree(sl.h_cands);
            sl.d_cands = nullptr; sl.h_cands = nullptr;
            sl.cand_cap *= 4;
            if (hipMalloc((void**)&sl.d_cands, sl.cand_cap * sizeof(lmat_cand)) != hipSuccess ||
                hipHostMalloc((void**)&sl.h_cands, sl.cand_cap * sizeof(lmat_cand), hipHostMallocDefault) != hipSuccess)
                return set_err(c, LMAT_E_NOMEM, "out of memory growing a candidate buffer");
            const int rc = stream_launch(st, sl, true);
            if (rc) return rc;
        }
        // wait on this batch's own events, never on a whole stream: both streams already carry work of the batches behind it.
        // The candidates (their number is only known now) go over the copy-in stream, whose queued copies are short.
        if (sl.cand_cap && nc) {
            HIPCHK(c, hipMemcpyAsync(sl.h_cands, sl.d_cands, nc * sizeof(lmat_cand), hipMemcpyDeviceToHost, st->s_h2d));
            HIPCHK(c, hipEventRecord(sl.ev_up, st->s_h2d));  // ev_up of this slot is long past: reuse it as "candidates copied"
            HIPCHK(c, hipEventSynchronize(sl.ev_up));
        }
        HIPCHK(c, hipEventSynchronize(sl.ev_out));
    }
    sl.state = 3;
    if (results) *results = sl.h_results;
    if (cands) *cands = sl.h_cands;
    if (n_reads) *n_reads = sl.n;
    if (n_cands) *n_cands = sl.cand_cap ? nc : 0;
    if (tag) *tag = sl.tag;
    return LMAT_OK;
}

int lmat_stream_release(lmat_stream* st) {
    if (!st) return LMAT_E_ARG;
    lmat_stream::Slot& sl = st->slots[st->tail % st->slots.size()];
    if (sl.state != 3) return set_err(st->c, LMAT_E_ARG, "no batch is checked out");
    sl.state = 0;
    ++st->tail;
    return LMAT_OK;
}

int lmat_table_address(int k, uint64_t want_buckets, uint64_t kmer, uint64_t* n_buckets, uint32_t* bucket, uint32_t* tag) {
    const CptGeom g = cpt_geometry(k, want_buckets);
    if (n_buckets) *n_buckets = g.nb;
    if (!g.nb) return LMAT_E_ARG;
    if (k < 32 && (kmer >> (2 * k))) return LMAT_E_ARG;
    const uint64_t r = revcomp_fwd(kmer, k);
    uint32_t b = 0, t = 0;
    cpt_address(g, kmer < r ? kmer : r, kmer < r ? r : kmer, b, t);
    if (bucket) *bucket = b;
    if (tag) *tag = t;
    return LMAT_OK;
}

int64_t lmat_format_out(const lmat_ctx* c, const lmat_read_result* results, uint64_t n, const lmat_cand* cands,
                        const uint8_t* bases, const uint64_t* off, int prn_read, uint64_t first_index, char* buf,
                        uint64_t cap) {
    if (!c || (n && !results)) return LMAT_E_ARG;
    std::string s;
    for (uint64_t i = 0; i < n; ++i) {
        s += 'r';
        put_int(s, (long long)(first_index + i));
        s += '\t';
        if (prn_read && bases && off) s.append((const char*)bases + off[i], off[i + 1] - off[i]);
        else s += 'X';
        s += '\t';
        format_call(s, c->params, c->dev.k, results[i], cands);
    }
    if (buf && cap) {
        const uint64_t m = std::min<uint64_t>(s.size(), cap - 1);
        memcpy(buf, s.data(), m);
        buf[m] = 0;
    }
    return (int64_t)s.size();
}

}  // extern "C"
