// read_label_main.cpp -- `read_label`-compatible command line on top of liblmat_hip.so.
//
// Keeps the CLI contract bin/run_rl.sh relies on (argv built at bin/run_rl.sh:243; getopt string and
// flag meanings src/read_label.cpp:1351-1442; required-argument check :1443-1453; outputs
// <o><N>.out, <o>.<x>.<j>.fastsummary, <o>.<x>.<j>.nomatchsum :1642-1647,1836-1867; banner and
// progress lines :1460-1462,1570-1573,1633,1707,1758,1843,1860,1869).  The per-read work
// (proc_line, :1211-1279) runs on the GPU through the C ABI; this file only parses input, batches
// reads, formats records and merges tallies (:1760-1800).
//
// Differences that are forced by the environment, all loud:
//   * -d takes an image written by make_db_image, a tax_histo binary (make_db_table's input) or a text
//     file listing several; PERM heap images cannot be opened without perm-je.
//   * -n (null models), -s (permissive match) and -g/-m (run-time pruning) are supported; the blank line
//     upstream prints to stdout per pruned k-mer (TaxNodeStat.hpp:192) is not reproduced.
//   * -t N is the number of output shards (<o>0.out .. <o>N-1.out) and of formatter threads; reads
//     are dealt to shards in contiguous blocks (the reference deals them dynamically, so only the
//     multiset of lines across shards is defined there; -t 1 gives the reference's -t 1 file).
//   * GPUs: one context (and one driving thread) per GPU, min(-t, visible GPUs) of them unless LMAT_DEVICE (one
//     device) or LMAT_DEVICES (a comma list) says otherwise; every GPU holds the whole database, batches are dealt to
//     the GPUs as they come free, and the writer puts them back in input order, so the output files do not depend
//     on the number of GPUs.  Each GPU is fed through a ring of pinned batch slots (lmat_stream_*): parsing, copy
//     in, classification, copy out and formatting of consecutive batches overlap.
#include <fcntl.h>
#include <getopt.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <fstream>
#include <future>
#include <iostream>
#include <map>
#include <set>
#include <sstream>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <deque>
#include <queue>
#include <thread>
#include <vector>
#include "../../include/lmat_hip.h"
#include "fastx.hpp"
#include "outfmt.hpp"
#include "rollups.hpp"

#define LMAT_VERSION "1.2.4_2018a"

using namespace lmat;

static void usage(const char* exe) {
    std::cout << "==============================================" << std::endl;
    std::cout << "  Livermore  Metagenomics  Analysis  Toolkit  " << std::endl;
    std::cout << "  read_label -- MI355X engine (liblmat_hip)    " << std::endl;
    std::cout << "==============================================" << std::endl;
    std::cout << std::endl << "Taxonomic classification module usage:" << std::endl;
    std::cout << exe << " -d <tax_histo db file (list)> -i <query fasta file> -t <number of output shards>" << std::endl;
    std::cout << "-o <output path> [-l <human bias>] -c <tax tree file> -e <depth file> -f <32to16 map> [-w <rank map>]" << std::endl;
    std::cout << "[-x min score] [-j min kmers] [-z min found kmers] [-b sdiff] [-p] [-a] [-q] [-h:turn phiX screening off]" << std::endl;
    std::cout << "[-V:print version and exit] [-H:print this usage help and exit]" << std::endl;
}

static bool is_list_file(const std::string& fn) {
    FILE* f = fopen(fn.c_str(), "rb");
    if (!f) return false;
    unsigned char b[20];
    size_t n = fread(b, 1, 20, f);
    fclose(f);
    if (n >= 8 && (memcmp(b, "LMATIMG1", 8) == 0 || memcmp(b, "LMATIMG2", 8) == 0)) return false;  // database image
    if (n < 20) return true;
    for (int i = 12; i < 20; ++i)
        if (b[i] != 0xff) return true;
    return false;
}

// Base buffers of the batches: pinned host memory (lmat_host_alloc), so that the GPU feed copies straight out of
// what the parser wrote; allocated once, recycled through a pool.
class BasePool {
    std::mutex m;
    std::condition_variable cv;
    std::vector<uint8_t*> free_;
    std::vector<uint8_t*> all_;
public:
    size_t cap = 0;  // bytes per buffer
    bool init(size_t count, size_t bytes) {
        cap = bytes;
        for (size_t i = 0; i < count; ++i) {
            void* p = nullptr;
            if (lmat_host_alloc(bytes, &p) != LMAT_OK) return false;
            all_.push_back((uint8_t*)p);
            free_.push_back((uint8_t*)p);
        }
        return true;
    }
    uint8_t* get() {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return !free_.empty(); });
        uint8_t* p = free_.back();
        free_.pop_back();
        return p;
    }
    void put(uint8_t* p) { std::lock_guard<std::mutex> l(m); free_.push_back(p); cv.notify_one(); }
    ~BasePool() { for (uint8_t* p : all_) lmat_host_free(p); }
};
static BasePool g_pool;

struct Batch {
    uint8_t* bases = nullptr;         // pooled, pinned; nb bytes used
    size_t nb = 0;
    std::vector<uint64_t> off{0};
    std::string hdr_blob;             // headers back to back
    std::vector<uint64_t> hoff{0};    // [n+1] into hdr_blob
    Batch() { bases = g_pool.get(); }
    ~Batch() { if (bases) g_pool.put(bases); }
    Batch(const Batch&) = delete;
    Batch& operator=(const Batch&) = delete;
    size_t n() const { return hoff.size() - 1; }
    size_t room() const { return g_pool.cap - nb; }
    bool overflow = false;            // something did not fit the pinned buffer: the batch is unusable and the run fails
    bool add_bases(const void* p, size_t len) {
        if (len > room()) { overflow = true; return false; }
        memcpy(bases + nb, p, len);
        nb += len;
        return true;
    }
    void add_hdr(const char* p, size_t len) { hdr_blob.append(p, len); hoff.push_back(hdr_blob.size()); }
    void clear() { nb = 0; overflow = false; off.assign(1, 0); hdr_blob.clear(); hoff.assign(1, 0); }
};

// CPUs this process may actually use: the cgroup quota when there is one (a container with 16 CPUs on a 256-thread host
// must not start 256-thread pools), else the hardware count
static unsigned usable_cpus() {
    unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r");
    if (f) {
        char q[32];
        long long period = 0;
        if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            const long long quota = atoll(q);
            if (quota > 0) hw = std::min<unsigned>(hw, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
        }
        fclose(f);
    }
    return hw;
}

// A contiguous piece of a FASTA file that starts at a '>' line -> one batch, with the record rules of FastxReader
// (fastx.hpp; src/read_label.cpp:1651-1713): a record is its header line and the following lines up to the next '>'
// line, lines of length <= 1 are ignored, a record without sequence is not emitted.  Pieces are parsed in parallel.
static void parse_fasta_piece(const char* p, const char* end, Batch& b) {
    b.clear();
    const char* hdr = nullptr;
    size_t hdr_len = 0;
    bool open = false;  // sequence bytes of the current record already appended
    auto close = [&]() {
        if (open) { b.off.push_back(b.nb); b.add_hdr(hdr ? hdr : "", hdr ? hdr_len : 0); }
        open = false;
    };
    while (p < end) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        const char* le = nl ? nl : end;
        const size_t len = (size_t)(le - p);
        if (len && *p == '>') {
            close();
            hdr = p + 1;
            hdr_len = len - 1;
        } else if (len > 1) {
            if (!b.add_bases(p, len)) return;  // the cutter keeps pieces within a pool buffer; if one is not, the caller fails the run
            open = true;
        }
        p = nl ? nl + 1 : end;
    }
    close();
}

// A piece of a FASTQ file between two synchronised cut points (below) -> one batch: FastxReader's state machine (fastx.hpp;
// src/read_label.cpp:1651-1713) run over memory, with its quirks -- a record is pushed at its '+' / '-' line under the header of
// the PREVIOUS record, exactly one quality line is skipped, a '>' line is both a header and sequence.  prev_hdr: the header in
// force when the piece starts (the last '@' line before it); last: this piece ends the file (the machine's turn at end of input).
static void parse_fastq_piece(const char* p, const char* end, const std::string& prev_hdr, bool last, Batch& b) {
    b.clear();
    std::string hdr = prev_hdr, last_hdr;
    size_t rec0 = 0;   // where the open record's bases start
    bool finished = false;
    while (!finished) {
        const char* line = p;
        size_t len = 0;
        if (p < end) {
            const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
            len = (size_t)((nl ? nl : end) - p);
            p = nl ? nl + 1 : end;
        } else {
            if (!last) break;
            finished = true;
        }
        char c0 = len ? line[0] : '\0';
        if (c0 == '>' || c0 == '@') { last_hdr = hdr; hdr.assign(line + 1, len - 1); }
        if (c0 != '@' && c0 != '+' && c0 != '-') {
            if (len && !b.add_bases(line, len)) return;
            c0 = '\0';
        }
        if ((finished || c0 == '+' || c0 == '-') && b.nb > rec0) {
            b.off.push_back(b.nb);
            const std::string& h = finished ? hdr : last_hdr;
            b.add_hdr(h.data(), h.size());
            rec0 = b.nb;
            if (p < end) {  // the quality line
                const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
                p = nl ? nl + 1 : end;
            }
        }
    }
}

// One batch on its way through the three stages: parse (reader thread) -> classify (main thread, GPU) ->
// format + write + tally (writer thread).  The stages of consecutive batches overlap.
struct Work {
    Batch b;
    uint64_t seq = 0;                    // position of the batch in the input
    std::vector<lmat_read_result> res;
    std::unique_ptr<lmat_cand[]> cands;  // uninitialised on purpose: zero-filling 64 candidates per read cost more than the GPU work
    size_t ncand = 0;
    std::vector<std::string> text;       // per output shard, once formatted
};
class WorkQueue {
    std::mutex m;
    std::condition_variable cv;
    std::queue<std::unique_ptr<Work>> q;
    bool closed = false;
    size_t cap;
public:
    explicit WorkQueue(size_t c) : cap(c) {}
    void push(std::unique_ptr<Work> w) {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return q.size() < cap || closed; });
        if (closed) return;
        q.push(std::move(w));
        cv.notify_all();
    }
    std::unique_ptr<Work> pop() {  // null = the producer is done
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return !q.empty() || closed; });
        if (q.empty()) return nullptr;
        std::unique_ptr<Work> w = std::move(q.front());
        q.pop();
        cv.notify_all();
        return w;
    }
    std::unique_ptr<Work> try_pop(bool& done) {  // null + done = the producer is done; null + !done = nothing queued right now
        std::lock_guard<std::mutex> l(m);
        done = q.empty() && closed;
        if (q.empty()) return nullptr;
        std::unique_ptr<Work> w = std::move(q.front());
        q.pop();
        cv.notify_all();
        return w;
    }
    void close() { std::lock_guard<std::mutex> l(m); closed = true; cv.notify_all(); }
};

int main(int argc, char* argv[]) {
    int c;
    int n_threads = 0, k_size = -1;
    float min_score = 0.0f;
    int min_kmer = 35, min_fnd_kmer = 1;
    lmat_params prm = {1.0f, 3.0f, 0.0f, 35, 1, 0, 1};
    std::string rank_map_file, rank_ids, kmer_db_fn, query_fn, ofbase, tax_tree_fn, depth_file, id_bit_conv_fn, plasmid_file, rand_hits_file;
    bool fastq = false, prn_read = true;
    int permissive = 0, max_count = 0;
    std::string rank_table_file;
    while ((c = getopt(argc, argv, "u:ahn:j:b:ye:w:pk:c:v:k:i:d:l:t:r:sm:o:x:f:g:z:qVH")) != -1) {
        switch (c) {
            case 'h': prm.screen_phix = 0; break;
            case 'r': plasmid_file = optarg; break;
            case 'f': id_bit_conv_fn = optarg; break;
            case 'j': min_kmer = atoi(optarg); break;
            case 'z': min_fnd_kmer = atoi(optarg); break;
            case 'u': rank_ids = optarg; break;
            case 'x': min_score = atof(optarg); break;
            case 'a': prn_read = false; break;
            case 'w': rank_map_file = optarg; break;
            case 's': permissive = 1; break;
            case 'n': rand_hits_file = optarg; break;
            case 'b': prm.sdiff = atof(optarg); break;
            case 'l': prm.hbias = atof(optarg); break;
            case 'y': break;  // verbose dumps are not produced
            case 'e': depth_file = optarg; break;
            case 'q': fastq = true; break;
            case 'p': prm.prn_all = 1; break;
            case 'm': rank_table_file = optarg; break;
            case 't': n_threads = atoi(optarg); break;
            case 'v': break;  // threshold: parsed and unused upstream too (proc_line's `threshold`)
            case 'c': tax_tree_fn = optarg; break;
            case 'k': k_size = atoi(optarg); break;
            case 'g': max_count = atoi(optarg); break;
            case 'i': query_fn = optarg; break;
            case 'd': kmer_db_fn = optarg; break;
            case 'o': ofbase = optarg; break;
            case 'V': std::cout << "LMAT version " << LMAT_VERSION << std::endl; exit(0);
            case 'H': usage(argv[0]); exit(0);
            default: std::cerr << "WARNING! Unrecognized option " << (char)c << " to ignore." << std::endl;
        }
    }
    if (depth_file == "") std::cerr << "ERROR! Missing depth_file" << std::endl;
    if (ofbase == "") std::cerr << "ERROR! Missing ofbase" << std::endl;
    if (n_threads == 0) std::cerr << "ERROR! Missing n_threads" << std::endl;
    if (kmer_db_fn == "") std::cerr << "ERROR! Missing kmer_db_fn" << std::endl;
    if (query_fn == "") std::cerr << "ERROR! Missing query_fn" << std::endl;
    if (depth_file == "" || ofbase == "" || n_threads == 0 || kmer_db_fn == "" || query_fn == "") {
        std::cerr << "Params: " << ofbase << " " << n_threads << " " << kmer_db_fn << " " << query_fn << " " << depth_file << std::endl;
        usage(argv[0]);
        return -1;
    }
    // no -f: a database of 32-bit taxids (upstream's TID_SIZE=32 build); storage codes come from the tree
    prm.min_score = min_score;
    prm.min_kmer = min_kmer;
    prm.min_fnd_kmer = min_fnd_kmer;

    std::cout << "=== LMAT === read_label === ver. " << LMAT_VERSION << " ===" << std::endl;
    // one context per GPU
    std::vector<int> devices;
    if (const char* d = getenv("LMAT_DEVICE")) devices.push_back(atoi(d));
    else if (const char* dl = getenv("LMAT_DEVICES")) {
        std::stringstream ss(dl);
        std::string tok;
        while (std::getline(ss, tok, ',')) if (!tok.empty()) devices.push_back(atoi(tok.c_str()));
    } else {
        const int visible = lmat_device_count();
        for (int d = 0; d < std::max(1, std::min(n_threads, visible)); ++d) devices.push_back(d);
    }
    const int n_gpu = (int)devices.size();
    std::vector<lmat_ctx*> ctxs(n_gpu, nullptr);
    std::vector<lmat_stream*>* rings_p = nullptr;
    auto destroy_all = [&]() {
        if (rings_p) for (lmat_stream*& r : *rings_p) { if (r) lmat_stream_destroy(r); r = nullptr; }
        for (lmat_ctx*& x : ctxs) { if (x) lmat_ctx_destroy(x); x = nullptr; }
    };
    for (int g = 0; g < n_gpu; ++g)
        if (lmat_ctx_create(devices[g], &prm, &ctxs[g]) != LMAT_OK) {
            std::cerr << "ERROR! no usable HIP device " << devices[g] << " (this build has no CPU path)" << std::endl;
            destroy_all();
            return -1;
        }
    lmat_ctx* ctx = ctxs[0];
    if (id_bit_conv_fn.length() > 0) std::cout << "Loading map file " << id_bit_conv_fn << "... ";
    std::cout << "Reading taxonomy tree " << tax_tree_fn << std::endl;
    std::cout << "Reading taxonomy depth " << depth_file << std::endl;
    if (!rank_table_file.empty() && max_count <= 0) std::cout << "Need to set -h <tid-cutoff> to use rank file map!\n";  // :1544-1545
    std::cout << "Start kmer DB load..." << std::endl;
    std::vector<std::string> files;
    if (is_list_file(kmer_db_fn)) {
        std::ifstream l(kmer_db_fn.c_str());
        std::string f;
        while (l >> f) files.push_back(f);
    } else files.push_back(kmer_db_fn);
    if (files.empty()) { std::cerr << "Error: unable to open kmer db [" << kmer_db_fn << "]" << std::endl; destroy_all(); return -1; }
    bool is_image = false;
    {
        FILE* f = fopen(files[0].c_str(), "rb");
        char magic[8] = {0};
        if (f) { if (fread(magic, 1, 8, f) == 8 && (memcmp(magic, "LMATIMG1", 8) == 0 || memcmp(magic, "LMATIMG2", 8) == 0)) is_image = true; fclose(f); }
    }
    uint32_t klen = 0;
    uint64_t n_total = 0;  // k-mer counts of the headers size the table up front, so the files stream through
    if (!is_image) {
        // k-mer length comes from the first file's header (KmerFileMetaData.cpp:44-94)
        FILE* f = fopen(files[0].c_str(), "rb");
        if (!f) { std::cerr << "Error: unable to open kmer db [" << files[0] << "]" << std::endl; destroy_all(); return -1; }
        fseek(f, 25, SEEK_SET);
        if (fread(&klen, 4, 1, f) != 1) klen = 0;
        fclose(f);
        for (auto& fn : files) {
            FILE* h = fopen(fn.c_str(), "rb");
            uint64_t nk = 0;
            if (h) { fseek(h, 4, SEEK_SET); if (fread(&nk, 8, 1, h) != 1) nk = 0; fclose(h); }
            n_total += nk;
        }
    }
    // host pipeline sizes: parser and formatter pools follow the CPUs this process may use; a batch is one FASTA piece
    size_t kPiece = 8u << 20;
    if (const char* e = getenv("LMAT_FASTA_PIECE")) kPiece = std::max<size_t>(16, strtoull(e, nullptr, 10));  // tests: force many pieces
    const unsigned n_cpu = usable_cpus();
    unsigned n_parse = std::max(1u, std::min(8u, n_cpu / 4));
    if (const char* e = getenv("LMAT_PARSE_THREADS")) n_parse = std::max(1, atoi(e));
    const size_t kBatch = 1u << 20;                                   // reads per GPU batch ...
    const size_t kBatchBases = std::max<size_t>(kPiece, 1u << 20) + (1u << 20);  // ... and bases: one piece (+ the tail of its last record)
    const int kSlots = 3;
    const uint32_t cands_per_read = prm.prn_all ? 24 : 8;  // a slot grows itself when a batch prints more (a re-run and 0.5 s of re-pinning: the
                                                           // 64 GiB bench table prints 14.3 pairs a read, and 16 a read was one growth per slot)
    int n_fmt_plan = std::max<int>(1, (int)std::min<unsigned>(n_cpu > 6 ? n_cpu - n_parse - 2 : n_cpu, 32u));
    if (const char* e = getenv("LMAT_FORMAT_THREADS")) n_fmt_plan = std::max(1, atoi(e));
    std::vector<lmat_stream*> rings(n_gpu, nullptr);
    rings_p = &rings;
    // every GPU loads the taxonomy itself (small); the database is parsed and built ONCE, on the first GPU, and the
    // others receive a replica of its table, overflow table and list arena device to device (lmat_db_clone:
    // hipMemcpyPeer over xGMI) -- not one parse of a 64 GB database per GPU.  The pinned batch buffers are allocated meanwhile
    std::vector<std::string> setup_err(n_gpu);
    bool pool_ok = true;
    {
        std::promise<bool> first_ready;
        std::shared_future<bool> first_ok = first_ready.get_future().share();
        std::vector<std::thread> th;
        th.emplace_back([&]() { pool_ok = g_pool.init(n_parse + 4 + (size_t)(kSlots + 1) * n_gpu + 4 + n_fmt_plan + 8 + 2, kBatchBases); });
        for (int g = 0; g < n_gpu; ++g)
            th.emplace_back([&, g]() {
                lmat_ctx* x = ctxs[g];
                struct Signal {  // whatever way the first GPU's thread leaves, the others stop waiting
                    std::promise<bool>* p; bool ok = false;
                    ~Signal() { if (p) p->set_value(ok); }
                } sig{g == 0 ? &first_ready : nullptr};
                auto bad = [&](const char* what) { setup_err[g] = std::string(what) + ": " + lmat_last_error(x); };
                if (lmat_taxonomy_load_files(x, tax_tree_fn.c_str(), depth_file.c_str(), rank_map_file.empty() ? nullptr : rank_map_file.c_str(),
                                             id_bit_conv_fn.empty() ? nullptr : id_bit_conv_fn.c_str(), plasmid_file.empty() ? nullptr : plasmid_file.c_str()) != LMAT_OK)
                    return bad("taxonomy");
                if (!rand_hits_file.empty() && lmat_nullmodel_load(x, rand_hits_file.c_str()) != LMAT_OK) return bad("null models");
                if (lmat_set_label_modes(x, permissive, max_count, max_count > 0 && !rank_table_file.empty() ? rank_table_file.c_str() : nullptr) != LMAT_OK)
                    return bad("label modes");
                if (g > 0) {
                    if (!first_ok.get()) { setup_err[g] = "k-mer DB: the first GPU's build failed"; return; }
                    if (lmat_db_clone(x, ctxs[0]) != LMAT_OK) return bad("k-mer DB replica");
                } else if (is_image) {  // image written by make_db_image (the engine's counterpart of the PERM .db file)
                    if (lmat_db_load_image(x, files[0].c_str(), 0) != LMAT_OK) return bad("k-mer DB image");
                    if (lmat_db_finalize(x) != LMAT_OK) return bad("k-mer DB image");
                } else {
                    if (lmat_db_begin(x, (int)klen, n_total, 0) != LMAT_OK) return bad("k-mer DB");
                    for (auto& fn : files)
                        if (lmat_db_add_taxhisto(x, fn.c_str()) != LMAT_OK) return bad("k-mer DB");
                    if (lmat_db_finalize(x) != LMAT_OK) return bad("k-mer DB");
                }
                if (g == 0)   // LMAT_SAVE_DEVICE_IMAGE=<file>: leave the database as it now lies in HBM (LMATIMG2) -- the next run's -d, which then
                              // starts by streaming the file in instead of parsing and inserting (the reference maps its .db the same way)
                    if (const char* img = getenv("LMAT_SAVE_DEVICE_IMAGE"))
                        if (*img && lmat_db_save_image(x, img) != LMAT_OK) return bad("device image");
                if (g == 0) { sig.ok = true; sig.p->set_value(true); sig.p = nullptr; }
                if (lmat_stream_create(x, kBatch, kBatchBases, cands_per_read, kSlots, &rings[g]) != LMAT_OK) return bad("batch ring");
            });
        for (auto& x : th) x.join();
    }
    if (!pool_ok) { std::cerr << "ERROR! out of pinned host memory for the batch buffers" << std::endl; destroy_all(); return -1; }
    for (int g = 0; g < n_gpu; ++g)
        if (!setup_err[g].empty()) {
            std::cerr << "ERROR! " << setup_err[g] << std::endl;
            destroy_all();
            return -1;
        }
    std::cout << "OK!" << std::endl;
    if (k_size < 1) k_size = is_image ? lmat_db_kmer_length(ctx) : (int)klen;
    std::cout << "Loaded k-mer DB into a " << (lmat_db_table_bytes(ctx) >> 20) << " MiB GPU hash";
    if (n_gpu > 1) std::cout << " on each of " << n_gpu << " GPUs";
    std::cout << ". Num of k-mers: " << lmat_db_size(ctx) << " of size " << k_size << std::endl;
    if (k_size <= 0) { std::cerr << "ERROR! Unable to read database, k-mer size=" << k_size << std::endl; destroy_all(); return -1; }

    auto t_start = std::chrono::steady_clock::now();
    std::ifstream qf;
    std::istream* in = &std::cin;
    if (query_fn != "-") {
        qf.open(query_fn.c_str());
        if (!qf) { std::cerr << "ERROR! Did not open for reading: " << query_fn << std::endl; exit(-1); }
        in = &qf;
    }
    std::cout << "Classifing reads in parallel with " << n_threads << " processes in .out files..." << std::endl;
    // The <o><t>.out shards are independent files (read_label.cpp:1642-1647): each has a writer thread of its own, fed in input
    // order by the sequencer below -- one thread pushing every shard's text through one ofstream after the other was the
    // pipeline's narrowest stage (round 3: 22 M reads/s with -p).
    struct ShardOut {
        int fd = -1;
        std::mutex m;
        std::condition_variable cv;
        std::deque<std::string> q;
        bool closed = false, bad = false;
        double busy = 0;
    };
    std::vector<ShardOut> shard(n_threads);
    for (int t = 0; t < n_threads; ++t) {
        std::ostringstream nm;
        nm << ofbase << t << ".out";
        shard[t].fd = open(nm.str().c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    }
    std::vector<std::thread> shard_writers;
    for (int t = 0; t < n_threads; ++t)
        shard_writers.emplace_back([&shard, t]() {
            ShardOut& so = shard[t];
            for (;;) {
                std::string s;
                {
                    std::unique_lock<std::mutex> l(so.m);
                    so.cv.wait(l, [&] { return !so.q.empty() || so.closed; });
                    if (so.q.empty()) break;
                    s.swap(so.q.front());
                    so.q.pop_front();
                    so.cv.notify_all();
                }
                auto t0 = std::chrono::steady_clock::now();
                for (size_t w = 0; w < s.size() && so.fd >= 0;) {
                    const ssize_t r = write(so.fd, s.data() + w, s.size() - w);
                    if (r <= 0) { so.bad = true; break; }
                    w += (size_t)r;
                }
                so.busy += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            }
        });
    auto shard_push = [&shard](int t, std::string&& s) {
        if (s.empty()) return;
        ShardOut& so = shard[t];
        std::unique_lock<std::mutex> l(so.m);
        so.cv.wait(l, [&] { return so.q.size() < 6 || so.closed; });
        so.q.emplace_back(std::move(s));
        so.cv.notify_all();
    };

    std::map<uint32_t, int> merge_count;
    std::map<uint32_t, float> merge_score;
    std::map<int, int> nomatch_merge;
    struct Acc { uint32_t tid = 0; int cnt = 0; float score = 0; bool used = false; };
    std::vector<Acc> acc(1 << 12);   // the writer's running per-taxid tallies (filled into the maps above at the end)
    size_t acc_mask = acc.size() - 1, acc_n = 0;
    int nm_acc[3] = {0, 0, 0};
    std::atomic<size_t> read_count(0);
    double t_parse = 0, t_gpu = 0, t_fmt = 0, t_write = 0, t_write_seq = 0, t_tally = 0;  // LMAT_CLI_TIMING=1 prints the busy time of each stage
    auto now = []() { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b_) { return std::chrono::duration<double>(b_ - a).count(); };
    WorkQueue parsed(4), classified(8);
    std::atomic<bool> failed(false);  // any stage: first message wins, the input queue is closed, the run ends with -1
    std::mutex fail_m;
    std::string fail_msg;
    // A FASTA query in a regular file is mapped and cut at '>' lines into pieces that a pool of threads parses, one
    // batch per piece, numbered in file order (FASTQ and stdin keep the sequential reader: its header pairing is
    // line-order dependent).  Batches may reach the writer out of order; it puts them back.
    const char* map_base = nullptr;
    size_t map_size = 0;
    static const bool fastq_seq = getenv("LMAT_FASTQ_SEQUENTIAL") && atoi(getenv("LMAT_FASTQ_SEQUENTIAL")) != 0;  // (the one-thread reader, for A/B runs)
    if ((!fastq || !fastq_seq) && query_fn != "-") {
        int fd = open(query_fn.c_str(), O_RDONLY);
        struct stat sb;
        if (fd >= 0 && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) {
            void* m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) { map_base = (const char*)m; map_size = (size_t)sb.st_size; madvise(m, map_size, MADV_WILLNEED); }
        }
        if (fd >= 0) close(fd);
    }
    std::thread reader([&]() {  // stage 1: FASTA/FASTQ -> batches
        if (map_base) {
            auto tp0 = now();
            // Cut points: the first '>' line at or behind start + kPiece; a piece never exceeds a pool buffer (kBatchBases).
            // A record too long for that is cut between two of its lines (both halves are then reads the engine refuses as
            // too long -- the run fails loudly, never silently); a single LINE longer than a buffer cannot be cut at all.
            std::vector<size_t> cut(1, 0);
            bool cut_failed = false;
            while (cut.back() < map_size && !cut_failed) {
                const size_t start = cut.back();
                size_t pos = start + kPiece;
                size_t c = map_size;
                while (pos < map_size) {
                    const char* nl = (const char*)memchr(map_base + pos, '\n', map_size - pos);
                    const size_t at = nl ? (size_t)(nl - map_base) + 1 : map_size;  // start of the next line (or the end)
                    if (at - start > kBatchBases) {  // the line that crosses `pos` ends beyond the buffer: cut in front of it
                        const char* prev = (const char*)memrchr(map_base + start, '\n', pos - start);
                        const size_t ls = prev ? (size_t)(prev - map_base) + 1 : start;
                        if (ls <= start) {
                            std::lock_guard<std::mutex> l(fail_m);
                            if (!failed.exchange(true)) fail_msg = "input: a line of more than " + std::to_string(kBatchBases) + " bytes exceeds the batch buffer";
                            cut_failed = true;
                        }
                        c = ls;
                        break;
                    }
                    if (at >= map_size) break;
                    if (fastq) {
                        // A FASTQ piece starts where the sequential reader is provably between two records: at an '@' line that
                        // follows [sequence line]['+' or '-' line][one more line = the quality line it skips].  (A quality line may
                        // itself start with '@' or '+': then the lines in front of it do not fit, and the search moves on.)
                        auto prev_line = [&](size_t ls) -> size_t {  // start of the line in front of the one that starts at ls
                            if (ls <= start + 1) return (size_t)-1;
                            const char* q = (const char*)memrchr(map_base + start, '\n', ls - 1 - start);
                            return q ? (size_t)(q - map_base) + 1 : start;
                        };
                        if (map_base[at] == '@') {
                            const size_t l1 = prev_line(at), l2 = l1 == (size_t)-1 ? l1 : prev_line(l1), l3 = l2 == (size_t)-1 ? l2 : prev_line(l2);
                            if (l3 != (size_t)-1 && (map_base[l2] == '+' || map_base[l2] == '-') && l2 - l3 > 1 &&
                                map_base[l3] != '@' && map_base[l3] != '+' && map_base[l3] != '-' && map_base[l3] != '>') { c = at; break; }
                        }
                    } else if (map_base[at] == '>') { c = at; break; }
                    if (at - start + (64u << 10) >= kBatchBases) {  // a record longer than a piece: cut between its lines
                        if (fastq) {   // (a FASTQ reader cut anywhere else than between records would lose its place: refuse)
                            std::lock_guard<std::mutex> l(fail_m);
                            if (!failed.exchange(true)) fail_msg = "input: no FASTQ record boundary within " + std::to_string(kBatchBases) + " bytes (LMAT_FASTQ_SEQUENTIAL=1 reads such a file with one thread)";
                            cut_failed = true;
                        }
                        c = at;
                        break;
                    }
                    pos = at;
                }
                if (!cut_failed) cut.push_back(c);
            }
            if (cut_failed) { parsed.close(); return; }
            const size_t np = cut.size() - 1;
            // FASTQ: the header in force at each cut = the last '@' line in front of it (the previous record's: sequence lines do
            // not start with '@', and the line right before the cut is the skipped quality line, which may)
            std::vector<std::string> cut_hdr(np);
            if (fastq)
                for (size_t j = 1; j < np; ++j) {
                    size_t ls = cut[j];
                    auto back = [&](size_t x) -> size_t {
                        if (x == 0) return (size_t)-1;
                        const char* q = x >= 2 ? (const char*)memrchr(map_base, '\n', x - 1) : nullptr;
                        return q ? (size_t)(q - map_base) + 1 : 0;
                    };
                    ls = back(ls);                 // the quality line
                    if (ls != (size_t)-1) ls = back(ls);   // the '+' line
                    while (ls != (size_t)-1 && ls != 0 && map_base[ls] != '@') ls = back(ls);
                    if (ls != (size_t)-1 && map_base[ls] == '@') {
                        const char* nl = (const char*)memchr(map_base + ls, '\n', map_size - ls);
                        const size_t le = nl ? (size_t)(nl - map_base) : map_size;
                        cut_hdr[j].assign(map_base + ls + 1, le - ls - 1);
                    }
                }
            std::atomic<size_t> next_piece(0);
            std::vector<std::thread> th;
            for (unsigned t = 0; t < n_parse; ++t)
                th.emplace_back([&]() {
                    for (;;) {
                        std::unique_ptr<Work> w(new Work());  // takes its base buffer BEFORE its piece: whoever holds piece i can always finish it
                        const size_t j = next_piece.fetch_add(1);
                        if (j >= np) break;
                        if (fastq) parse_fastq_piece(map_base + cut[j], map_base + cut[j + 1], cut_hdr[j], j + 1 == np, w->b);
                        else parse_fasta_piece(map_base + cut[j], map_base + cut[j + 1], w->b);
                        if (w->b.overflow) {
                            std::lock_guard<std::mutex> l(fail_m);
                            if (!failed.exchange(true)) fail_msg = "input: a piece of the query file exceeds the batch buffer";
                            parsed.close();
                            break;
                        }
                        w->seq = j;
                        read_count += w->b.n();
                        parsed.push(std::move(w));
                    }
                });
            for (auto& x : th) x.join();
            t_parse += secs(tp0, now());
            std::cout << "Total reads loaded: " << read_count << std::endl;
            parsed.close();
            return;
        }
        FastxReader rd(*in, fastq);
        std::string read, hdr;
        bool more = true;
        uint64_t next_seq = 0;
        bool have_pending = false;
        while (more) {
            auto tp0 = now();
            std::unique_ptr<Work> w(new Work());
            Batch& b = w->b;
            while (b.n() < kBatch) {
                if (!have_pending && !rd.next(read, hdr)) { more = false; break; }
                have_pending = false;
                if (read.size() > b.room()) {
                    if (b.n() == 0) {
                        std::lock_guard<std::mutex> l(fail_m);
                        if (!failed.exchange(true)) fail_msg = "input: a read of " + std::to_string(read.size()) + " bases exceeds the batch buffer";
                        more = false;
                    }
                    else have_pending = true;  // goes into the next batch
                    break;
                }
                ++read_count;
                b.add_hdr(hdr.data(), hdr.size());   // an empty header is named by the writer, which knows the read's number
                b.add_bases(read.data(), read.size());
                b.off.push_back(b.nb);
            }
            if (!more) std::cout << "Total reads loaded: " << read_count << std::endl;
            t_parse += secs(tp0, now());
            if (b.n()) { w->seq = next_seq++; parsed.push(std::move(w)); }
        }
        parsed.close();
    });
    const int n_fmt = n_fmt_plan;
    size_t reads_before = 0;  // reads of the batches already written (writer thread only)
    // stage 3a: formatter pool.  Each worker turns whole batches into the text of the n_threads output shards (shard t
    // holds a contiguous block of every batch); batches are independent, so any number of them is formatted at once.
    WorkQueue formatted(8);
    std::mutex fmt_m;
    auto format_batch = [&](Work& w) {
        const Batch& b = w.b;
        const lmat_cand* cands = w.cands.get();
        const size_t n = b.n();
        const size_t per = (n + n_threads - 1) / n_threads;
        w.text.assign(n_threads, std::string());
        for (int t = 0; t < n_threads; ++t) {
            const size_t lo = std::min(n, t * per), hi = std::min(n, lo + per);
            std::string s;  // built locally: neighbouring std::string headers in a vector share cache lines
            s.reserve((hi - lo) * (prn_read ? 384 : 200));
            for (size_t i = lo; i < hi; ++i) {
                s.append(b.hdr_blob.data() + b.hoff[i], b.hoff[i + 1] - b.hoff[i]);
                s += '\t';
                if (prn_read) s.append((const char*)b.bases + b.off[i], b.off[i + 1] - b.off[i]);
                else s += 'X';
                s += '\t';
                format_call(s, prm, k_size, w.res[i], cands);
            }
            w.text[t] = std::move(s);
        }
    };
    std::vector<std::thread> formatters;
    std::atomic<int> fmt_live(n_fmt);
    for (int f = 0; f < n_fmt; ++f)
        formatters.emplace_back([&]() {
            double busy = 0;
            while (std::unique_ptr<Work> w = classified.pop()) {
                auto tp2 = now();
                format_batch(*w);
                busy += secs(tp2, now());
                formatted.push(std::move(w));
            }
            { std::lock_guard<std::mutex> l(fmt_m); t_fmt = std::max(t_fmt, busy); }
            if (--fmt_live == 0) formatted.close();
        });
    std::thread writer([&]() {  // stage 3b: files and tallies, in input order whatever GPU and formatter a batch went through
        std::map<uint64_t, std::unique_ptr<Work>> held;
        uint64_t want_seq = 0;
        for (;;) {
            std::unique_ptr<Work> w;
            auto it = held.find(want_seq);
            if (it != held.end()) { w = std::move(it->second); held.erase(it); }
            else {
                w = formatted.pop();
                if (!w) break;
                if (w->seq != want_seq) { const uint64_t sq = w->seq; held[sq] = std::move(w); continue; }
            }
            ++want_seq;
            auto tp3 = now();
            {   // "unknown_hdr:<running read number>" for records without a header (:1728-1732): only the writer knows the
                // read's number, so such a batch (rare) gets its headers now and is formatted again
                Batch& bb = w->b;
                const size_t nn = bb.n();
                bool unnamed = false;
                for (size_t i = 0; i < nn && !unnamed; ++i) unnamed = bb.hoff[i + 1] == bb.hoff[i] || bb.hdr_blob[bb.hoff[i]] == '\0';
                if (unnamed) {
                    std::string blob;
                    std::vector<uint64_t> ho(1, 0);
                    for (size_t i = 0; i < nn; ++i) {
                        if (bb.hoff[i + 1] == bb.hoff[i] || bb.hdr_blob[bb.hoff[i]] == '\0') blob += "unknown_hdr:" + std::to_string(reads_before + i + 1);
                        else blob.append(bb.hdr_blob, bb.hoff[i], bb.hoff[i + 1] - bb.hoff[i]);
                        ho.push_back(blob.size());
                    }
                    bb.hdr_blob.swap(blob);
                    bb.hoff.swap(ho);
                    format_batch(*w);
                }
                reads_before += nn;
            }
            const std::vector<lmat_read_result>& res = w->res;
            const size_t n = w->b.n();
            for (int t = 0; t < n_threads; ++t) shard_push(t, std::move(w->text[t]));
            auto tp4 = now();
            t_write_seq += secs(tp3, tp4);
            // tallies in read order (proc_line :1241-1276), float sums like a -t 1 run.  One thread sees every record, so the
            // per-taxid sums live in an open-addressing table (three std::map searches per read were the writer's largest
            // cost: 28 ns of its 37 per read) and go into the maps the summaries are made from when the run is over.
            for (size_t i = 0; i < n; ++i) {
                const lmat_read_result& r = res[i];
                if (r.status == LMAT_ST_SHORT_LEN || r.status == LMAT_ST_SHORT_VALID) nm_acc[0] += 1;
                else if (r.status == LMAT_ST_NODBHITS || r.status == LMAT_ST_SILENT) nm_acc[1] += 1;
                else if (r.status != LMAT_ST_PHIX && r.match_type == LMAT_MT_NOMATCH) nm_acc[1] += 1;
                else if (r.call_score >= min_score) {
                    size_t h = ((size_t)r.call_tid * 0x9E3779B1u >> 8) & acc_mask;
                    while (acc[h].used && acc[h].tid != r.call_tid) h = (h + 1) & acc_mask;
                    if (!acc[h].used) {
                        acc[h].used = true; acc[h].tid = r.call_tid; acc[h].cnt = 1; acc[h].score = r.call_score;
                        if (++acc_n * 2 > acc.size()) {  // keep it at most half full
                            std::vector<Acc> old(acc.size() * 2);
                            old.swap(acc);
                            acc_mask = acc.size() - 1;
                            for (const Acc& a : old)
                                if (a.used) {
                                    size_t g = ((size_t)a.tid * 0x9E3779B1u >> 8) & acc_mask;
                                    while (acc[g].used) g = (g + 1) & acc_mask;
                                    acc[g] = a;
                                }
                        }
                    } else { acc[h].cnt += 1; acc[h].score += r.call_score; }
                } else if (r.call_score < min_score) nm_acc[2] += 1;
            }
            t_tally += secs(tp4, now());
        }
    });
    // stage 2: the GPUs.  One thread per context keeps its ring of pinned slots full: a batch is copied into a slot and
    // queued (copy in, packing, classification and the copy of the results back are asynchronous), the oldest batch in
    // flight is collected when the ring is full or the input has nothing ready.
    std::vector<double> t_gpu_g(n_gpu, 0.0);
    std::vector<std::thread> gpu_threads;
    for (int g = 0; g < n_gpu; ++g)
        gpu_threads.emplace_back([&, g]() {
            lmat_ctx* x = ctxs[g];
            lmat_stream* st = rings[g];
            auto bail = [&](const char* what) {
                std::lock_guard<std::mutex> l(fail_m);
                if (!failed.exchange(true)) fail_msg = std::string(what) + ": " + lmat_last_error(x);
                parsed.close();
            };
            auto collect = [&](Work& w, size_t at) -> bool {  // oldest batch in flight -> w.res[at..], candidates appended
                const lmat_read_result* res = nullptr;
                const lmat_cand* cd = nullptr;
                uint64_t n = 0, nc = 0, tag = 0;
                if (lmat_stream_next(st, &res, &cd, &n, &nc, &tag) != LMAT_OK) return false;
                const size_t c0 = w.ncand;
                std::unique_ptr<lmat_cand[]> grown(new lmat_cand[std::max<uint64_t>(c0 + nc, 1)]);
                if (c0) memcpy(grown.get(), w.cands.get(), c0 * sizeof(lmat_cand));
                if (nc) memcpy(grown.get() + c0, cd, nc * sizeof(lmat_cand));
                w.cands = std::move(grown);
                w.ncand = c0 + nc;
                for (uint64_t i = 0; i < n; ++i) { w.res[at + i] = res[i]; w.res[at + i].cand_off += (uint32_t)c0; }
                lmat_stream_release(st);
                return true;
            };
            std::queue<std::unique_ptr<Work>> flying;
            bool input_done = false;
            while (!failed) {
                std::unique_ptr<Work> w;
                if ((int)flying.size() < kSlots && !input_done) {
                    if (flying.empty()) { w = parsed.pop(); if (!w) input_done = true; }
                    else w = parsed.try_pop(input_done);
                }
                auto tp1 = now();
                if (w) {
                    const size_t n = w->b.n();
                    w->res.resize(n);
                    if (n == 0) { classified.push(std::move(w)); continue; }  // an empty piece keeps its place in the order
                    uint8_t* hb = nullptr;
                    uint64_t* ho = nullptr;
                    if (n <= kBatch) {
                        if (lmat_stream_acquire(st, &hb, &ho) != LMAT_OK || lmat_stream_submit_from(st, w->b.bases, w->b.off.data(), n, w->seq) != LMAT_OK) { bail("classify"); break; }
                        flying.push(std::move(w));
                    } else {  // a piece of very short reads: more reads than a slot takes, in sub-batches one after the other
                        bool ok = true;
                        while (ok && !flying.empty()) { ok = collect(*flying.front(), 0); if (ok) { classified.push(std::move(flying.front())); flying.pop(); } }
                        for (size_t lo = 0; ok && lo < n; lo += kBatch) {
                            const size_t m = std::min(kBatch, n - lo);
                            ok = lmat_stream_acquire(st, &hb, &ho) == LMAT_OK && lmat_stream_submit_from(st, w->b.bases, w->b.off.data() + lo, m, w->seq) == LMAT_OK && collect(*w, lo);
                        }
                        if (!ok) { bail("classify"); break; }
                        classified.push(std::move(w));
                    }
                    t_gpu_g[g] += secs(tp1, now());
                    continue;
                }
                if (flying.empty()) break;  // input done, nothing in flight
                if (!collect(*flying.front(), 0)) { bail("classify"); break; }
                std::unique_ptr<Work> done = std::move(flying.front());
                flying.pop();
                t_gpu_g[g] += secs(tp1, now());
                classified.push(std::move(done));
            }
        });
    for (auto& x : gpu_threads) x.join();
    for (double v : t_gpu_g) t_gpu = std::max(t_gpu, v);
    if (failed) parsed.close();
    classified.close();
    reader.join();
    for (auto& x : formatters) x.join();
    writer.join();
    for (auto& so : shard) { std::lock_guard<std::mutex> l(so.m); so.closed = true; so.cv.notify_all(); }
    for (auto& x : shard_writers) x.join();
    for (auto& so : shard) {
        t_write = std::max(t_write, so.busy);
        if (so.fd < 0 || so.bad || close(so.fd) != 0) {
            std::lock_guard<std::mutex> l(fail_m);
            if (!failed.exchange(true)) fail_msg = "cannot write an output shard (" + ofbase + "<t>.out)";
        }
        so.fd = -1;
    }
    for (const Acc& a : acc)
        if (a.used) { merge_count[a.tid] = a.cnt; merge_score[a.tid] = a.score; }
    for (int j = 0; j < 3; ++j)
        if (nm_acc[j]) nomatch_merge[j] = nm_acc[j];
    if (failed) {
        std::cerr << "ERROR! " << fail_msg << std::endl;
        destroy_all();
        return -1;
    }
    // the device-side tallies of the contexts, merged (read_label.cpp:1760-1800): every context now holds the totals,
    // and their read counts must be the ones the writer summed on the host in input order
    if (lmat_counts_allreduce(ctxs.data(), n_gpu) != LMAT_OK) { std::cerr << "ERROR! merge: " << lmat_last_error(ctx) << std::endl; destroy_all(); return -1; }
    {
        std::vector<uint32_t> tids(70000);
        std::vector<uint64_t> cnts(70000);
        uint32_t nz = 0;
        uint64_t nm3[3] = {0, 0, 0};
        bool same = lmat_counts_get(ctx, tids.data(), cnts.data(), nullptr, 70000, &nz, nm3) == LMAT_OK && nz == merge_count.size();
        for (uint32_t i = 0; same && i < nz; ++i) same = merge_count.count(tids[i]) && (uint64_t)merge_count[tids[i]] == cnts[i];
        for (int j = 0; same && j < 3; ++j) same = (uint64_t)(nomatch_merge.count(j) ? nomatch_merge[j] : 0) == nm3[j];
        if (!same) { std::cerr << "ERROR! the merged device tallies differ from the per-record tallies" << std::endl; destroy_all(); return -1; }
    }
    if (getenv("LMAT_CLI_TIMING"))
        std::cerr << "[read_label] stage busy time: parse " << t_parse << " s, GPU feed (copy into the pinned slots + waiting for results, busiest of "
                  << n_gpu << " GPUs) " << t_gpu << " s, format (busiest of " << n_fmt << " workers) " << t_fmt << " s, sequencer " << t_write_seq << " s, write (busiest shard) " << t_write << " s, tally " << t_tally << " s" << std::endl;
    const double t_pipeline = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    std::cout << "Finished classifing reads, doing final steps sequentially..." << std::endl;

    // names for the called taxids from the -u file (:1812-1835)
    std::set<uint32_t> cand_tid;
    std::vector<std::pair<uint32_t, float>> sort_val(merge_score.begin(), merge_score.end());
    for (auto& p : sort_val) cand_tid.insert(p.first);
    std::map<uint32_t, std::string> save_id;
    if (!rank_ids.empty()) {
        std::ifstream ts(rank_ids.c_str());
        std::string proc;
        while (std::getline(ts, proc)) {
            std::vector<char> buff(proc.begin(), proc.end());
            buff.push_back('\0');
            char* val = strtok(buff.data(), "=,");
            while (val != NULL) {
                if (strcmp(val, "taxid") == 0) {
                    val = strtok(NULL, "=,");
                    if (!val) break;
                    uint32_t cid = (uint32_t)strtoul(val, nullptr, 10);
                    if (cand_tid.count(cid)) {
                        size_t pos = proc.rfind('\t');
                        save_id.insert(std::make_pair(cid, proc.substr(pos + 1)));
                    }
                    break;
                }
                val = strtok(NULL, "=,");
            }
        }
    }
    std::string sbase;
    { std::string t = ofbase; t += '.'; put_float(t, min_score); t += '.'; put_int(t, min_kmer); sbase = t; }
    {
        std::ofstream sum_ofs((sbase + ".fastsummary").c_str());
        if (!sum_ofs) { std::cerr << "ERROR! Could not open for writing " << sbase << ".fastsummary" << std::endl; return -1; }
        std::cout << "Writing FastSummary file in " << sbase << ".fastsummary" << std::endl;
        std::sort(sort_val.begin(), sort_val.end(),
                  [](const std::pair<uint32_t, float>& a, const std::pair<uint32_t, float>& b) { return a.second > b.second; });
        for (auto& p : sort_val) {
            std::string s;
            put_float(s, p.second); s += '\t'; put_int(s, merge_count[p.first]); s += '\t'; put_int(s, p.first); s += '\t';
            s += save_id[p.first];
            sum_ofs << s << std::endl;
        }
    }
    {
        std::ofstream nom_ofs((sbase + ".nomatchsum").c_str());
        if (!nom_ofs) { std::cerr << "ERROR! Could not open for writing " << sbase << ".nomatchsum" << std::endl; return -1; }
        std::cout << "Writing NoMatchSum file in " << sbase << ".nomatchsum" << std::endl;
        static const char* names[3] = {"ReadTooShort", "NoDbHits", "LowScore"};
        for (auto& p : nomatch_merge) nom_ofs << names[p.first] << "\t" << p.second << std::endl;
    }
    // LMAT_ROLLUPS=<ranks, e.g. plasmid,species,genus>: the roll-ups bin/run_rl.sh makes from the .fastsummary right after
    // read_label (tolineage.py + fsreport.py, :251-252) written here, from the inputs this run already has: -u names, -c tree,
    // -w rank table, -r plasmid list (+ $LMAT_DIR/plasmid.names.txt when it exists)
    if (const char* ru = getenv("LMAT_ROLLUPS")) {
        RollupInputs in;
        in.tree_fn = tax_tree_fn; in.rank_fn = rank_map_file; in.plasmid_fn = plasmid_file;
        if (const char* ld = getenv("LMAT_DIR")) in.plasmid_names_fn = std::string(ld) + "/plasmid.names.txt";
        std::string err, odir = ".";
        { const size_t sl = sbase.rfind('/'); if (sl != std::string::npos) odir = sbase.substr(0, sl); }
        bool ok = rank_ids.empty() || write_lineage(rank_ids, sbase + ".fastsummary", sbase + ".fastsummary.lineage", 10, 0.0, &err);
        if (ok && !rank_map_file.empty() && !tax_tree_fn.empty()) ok = write_rank_reports(sbase + ".fastsummary", ru, odir, in, &err);
        if (!ok) { std::cerr << "ERROR! roll-ups: " << err << std::endl; destroy_all(); return -1; }
    }
    // the query timer stops where upstream's does (read_label.cpp:1868): after the summaries, before teardown
    double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    if (getenv("LMAT_CLI_TIMING")) std::cerr << "[read_label] timeline: pipeline drained at " << t_pipeline << " s, summaries written at " << el << " s" << std::endl;
    std::cout << "DONE! Total query time: " << el << " sec = " << el / 60 << " min" << std::endl;
    destroy_all();
    return 0;
}
