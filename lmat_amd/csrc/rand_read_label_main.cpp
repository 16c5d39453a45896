// rand_read_label -- the null-model generator with the reference's command line (src/rand_read_label.cpp:410-513:
// getopt string "u:ah:n:j:b:ye:w:mpk:c:v:k:i:d:l:t:s:r:o:x:f:g:z:q:") on the MI355X engine: random reads of a fixed
// length in ten GC buckets (genRandRead, :83-103) are labelled with src/rkmer.hpp's retrieve_kmer_labels, and per
// (taxid, bucket) the largest k-mer fraction and the number of reads that hit the taxid go to <o>.rand_lst (:735-755).
//   -d <tax_histo | list | make_db_image output>   -c <tax tree>  -e <depth file>  -f <32->16 map>  -w <taxid rank names>
//   -t <threads>  -g <reads per thread>  -i <read length>  -o <output base>  [-h <tid cutoff> -r <numeric rank map>] [-k k]
//   extras: -S <seed> (upstream seeds from the clock), -O <fasta> dumps the generated reads (header: >r<n> gc=<bucket>)
// Upstream every OpenMP thread draws its own g reads, so t * g reads are evaluated in total; here t only multiplies.
#include <getopt.h>
#include <chrono>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>
#include "../../include/lmat_hip.h"
#include "outfmt.hpp"

using namespace lmat;

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static inline uint32_t rnd() {  // splitmix64, 31 bits like rand()
    uint64_t z = (g_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)((z ^ (z >> 31)) >> 33);
}

// genRandRead, rand_read_label.cpp:83-103: a GC fraction drawn from [beg, end] %, that many g/c, the rest a/t, shuffled
static void gen_rand_read(char* buf, unsigned rl, int beg, int end) {
    const int range = (end - beg) + 1;
    const int gc_draw = (int)(rnd() % (unsigned)range) + beg;
    const float gc_pcnt = gc_draw / 100.0;
    const unsigned num_gc = static_cast<unsigned>(gc_pcnt * rl);
    for (unsigned i = 0; i < num_gc; ++i) buf[i] = (rnd() % 100) < 50 ? 'g' : 'c';
    for (unsigned i = num_gc; i < rl; ++i) buf[i] = (rnd() % 100) < 50 ? 'a' : 't';
    for (unsigned i = rl; i > 1; --i) { const unsigned j = rnd() % i; std::swap(buf[i - 1], buf[j]); }  // random_shuffle
}

int main(int argc, char* argv[]) {
    std::string rank_map_file, kmer_db_fn, ofbase, tax_tree_fn, depth_file, rank_table_file, id_bit_conv_fn, dump_fn;
    int k_size = -1, max_count = 0;
    unsigned n_threads = 0, num_reads = 0, read_len = 0;
    uint64_t seed = (uint64_t)std::chrono::system_clock::now().time_since_epoch().count();
    int c;
    while ((c = getopt(argc, argv, "u:ah:n:j:b:ye:w:mpk:c:v:i:d:l:t:s:r:o:x:f:g:z:q:S:O:")) != -1) {
        switch (c) {
            case 'f': id_bit_conv_fn = optarg; break;
            case 'e': depth_file = optarg; break;
            case 'w': rank_map_file = optarg; break;
            case 'h': max_count = atoi(optarg); break;
            case 'r': rank_table_file = optarg; break;
            case 't': n_threads = (unsigned)atoi(optarg); break;
            case 'c': tax_tree_fn = optarg; break;
            case 'k': k_size = atoi(optarg); break;
            case 'g': num_reads = (unsigned)atoi(optarg); break;
            case 'i': read_len = (unsigned)atoi(optarg); break;
            case 'd': kmer_db_fn = optarg; break;
            case 'o': ofbase = optarg; break;
            case 'S': seed = strtoull(optarg, nullptr, 10); break;
            case 'O': dump_fn = optarg; break;
            case 'u': case 'a': case 'n': case 'j': case 'b': case 'y': case 'm': case 'p': case 'v': case 'l': case 's': case 'x':
            case 'z': case 'q': break;  // accepted and unused, as upstream
            default: std::cout << "Unrecognized option: " << (char)c << ", ignore." << std::endl;
        }
    }
    std::cout << "Total reads to evaluate: " << (uint64_t)num_reads * n_threads << std::endl;
    if (ofbase.empty() || n_threads == 0 || kmer_db_fn.empty() || tax_tree_fn.empty() || depth_file.empty() || !num_reads || !read_len) {
        std::cout << "Usage:\n" << argv[0] << " -d <db> -c <tax tree> -e <depth file> [-f <32to16 map>] -w <rank names> -t <threads> "
                  << "-g <reads per thread> -i <read length> -o <output base> [-h <tid-cutoff> -r <rank/tid-map-file>]\n";
        return -1;
    }
    g_state ^= seed * 0xD1342543DE82EF95ull;
    lmat_params prm = {1.0f, 3.0f, 0.0f, 1, 0, 0, 1};
    lmat_ctx* ctx = nullptr;
    const char* dv = getenv("LMAT_DEVICE");
    if (lmat_ctx_create(dv ? atoi(dv) : 0, &prm, &ctx) != LMAT_OK) { std::cerr << "cannot create a GPU context" << std::endl; return -1; }
    auto fail = [&](const char* what) { std::cerr << what << ": " << lmat_last_error(ctx) << std::endl; return -1; };
    if (lmat_taxonomy_load_files(ctx, tax_tree_fn.c_str(), depth_file.c_str(), rank_map_file.empty() ? nullptr : rank_map_file.c_str(),
                                 id_bit_conv_fn.empty() ? nullptr : id_bit_conv_fn.c_str(), nullptr) != LMAT_OK)
        return fail("taxonomy");
    if (lmat_rand_mode(ctx, 1) != LMAT_OK) return fail("rand mode");
    if (!rank_table_file.empty() && max_count <= 0) std::cout << "Need to set -h <tid-cutoff> to use rank file map!\n";  // :639
    if (lmat_set_label_modes(ctx, 0, max_count, max_count > 0 && !rank_table_file.empty() ? rank_table_file.c_str() : nullptr) != LMAT_OK)
        return fail("label modes");
    std::cout << "Start kmer DB load\n";
    {
        std::vector<std::string> files;
        char magic[8] = {0};
        unsigned char hdr[20] = {0};
        size_t got = 0;
        { FILE* f = fopen(kmer_db_fn.c_str(), "rb"); if (f) { got = fread(hdr, 1, 20, f); fclose(f); } memcpy(magic, hdr, 8); }
        if (got < 20) { std::cerr << "Error: unable to open kmer db [" << kmer_db_fn << "]" << std::endl; return -1; }
        if (memcmp(magic, "LMATIMG1", 8) == 0 || memcmp(magic, "LMATIMG2", 8) == 0) {
            if (lmat_db_load_image(ctx, kmer_db_fn.c_str(), 0) != LMAT_OK) return fail("k-mer DB image");
        } else {
            bool is_list = false;
            for (int i = 12; i < 20; ++i) if (hdr[i] != 0xff) is_list = true;  // a tax_histo binary has 64 one-bits here
            if (is_list) { std::ifstream l(kmer_db_fn.c_str()); std::string f; while (l >> f) files.push_back(f); }
            else files.push_back(kmer_db_fn);
            uint32_t klen = 0;
            { FILE* f = fopen(files[0].c_str(), "rb"); if (f) { fseek(f, 25, SEEK_SET); if (fread(&klen, 4, 1, f) != 1) klen = 0; fclose(f); } }
            uint64_t n_total = 0;
            for (auto& fn : files) {
                FILE* h = fopen(fn.c_str(), "rb");
                uint64_t nk = 0;
                if (h) { fseek(h, 4, SEEK_SET); if (fread(&nk, 8, 1, h) != 1) nk = 0; fclose(h); }
                n_total += nk;
            }
            if (lmat_db_begin(ctx, (int)klen, n_total, 0) != LMAT_OK) return fail("k-mer DB");
            for (auto& fn : files) if (lmat_db_add_taxhisto(ctx, fn.c_str()) != LMAT_OK) return fail("k-mer DB");
        }
        if (lmat_db_finalize(ctx) != LMAT_OK) return fail("k-mer DB");
    }
    if (k_size < 1) k_size = lmat_db_kmer_length(ctx);
    std::cout << "k size:  " << k_size << std::endl;
    std::cout << "num kmers: " << lmat_db_size(ctx) << " - " << k_size << std::endl;
    auto t0 = std::chrono::steady_clock::now();
    const int num_bins = 10;
    std::vector<std::pair<int, int>> gc_range(num_bins);  // :671-682
    {
        const float width = 100.0 / (float)num_bins;
        float lval = 0;
        for (int i = 0; i < num_bins; ++i) {
            gc_range[i] = std::make_pair((int)static_cast<float>(lval), (int)static_cast<float>(lval + width - 1));
            std::cout << "gc check " << i << " " << gc_range[i].first << " " << gc_range[i].second << std::endl;
            lval += width;
        }
    }
    if (lmat_rand_reset(ctx, num_bins) != LMAT_OK) return fail("tables");
    std::ofstream dump;
    if (!dump_fn.empty()) dump.open(dump_fn.c_str());
    const uint64_t total = (uint64_t)num_reads * n_threads, kBatch = 1u << 20;
    std::vector<uint8_t> bases;
    std::vector<uint64_t> off;
    std::vector<uint8_t> gcb;
    uint64_t serial = 0;
    for (unsigned th = 0; th < n_threads; ++th) {
        for (uint64_t done = 0; done < num_reads;) {
            const uint64_t n = std::min<uint64_t>(kBatch, num_reads - done);
            bases.assign(n * read_len + 1, 0);
            off.resize(n + 1);
            gcb.resize(n);
            for (uint64_t i = 0; i < n; ++i) {
                const int b = (int)((done + i) % num_bins);  // gc_bucket = i % num_gcbuckets, :695
                gen_rand_read((char*)bases.data() + i * read_len, read_len, gc_range[b].first, gc_range[b].second);
                off[i] = i * read_len;
                gcb[i] = (uint8_t)b;
                if (dump.is_open()) { dump << ">r" << serial << " gc=" << b << "\n"; dump.write((const char*)bases.data() + i * read_len, read_len); dump << "\n"; }
                ++serial;
            }
            off[n] = n * read_len;
            lmat_reads* dr = nullptr;
            if (lmat_reads_upload(ctx, bases.data(), off.data(), n, &dr) != LMAT_OK) return fail("read upload");
            if (lmat_rand_label(ctx, dr, 0, n, gcb.data()) != LMAT_OK) return fail("labelling");
            lmat_reads_free(ctx, dr);
            done += n;
            std::cout << "progress " << th << " " << done << std::endl;
        }
    }
    (void)total;
    std::cout << "Merge phase" << std::endl;
    uint32_t rows = 0;
    if (lmat_rand_get(ctx, nullptr, nullptr, nullptr, 0, &rows) != LMAT_OK) return fail("tables");
    std::vector<uint32_t> tid(std::max<uint32_t>(rows, 1)), cnt((size_t)std::max<uint32_t>(rows, 1) * num_bins);
    std::vector<float> mx((size_t)std::max<uint32_t>(rows, 1) * num_bins);
    if (lmat_rand_get(ctx, tid.data(), mx.data(), cnt.data(), rows, &rows) != LMAT_OK) return fail("tables");
    const std::string ofn = ofbase + ".rand_lst";
    std::ofstream sum_ofs(ofn.c_str());
    if (!sum_ofs) { std::cout << "Could not open for writing " << ofn << std::endl; return -1; }
    std::string line;
    for (uint32_t r = 0; r < rows; ++r) {  // :741-754: tid, then " max count" per bucket
        line.clear();
        put_int(line, tid[r]);
        for (int b = 0; b < num_bins; ++b) { line += ' '; put_float(line, mx[(size_t)r * num_bins + b]); line += ' '; put_int(line, cnt[(size_t)r * num_bins + b]); }
        sum_ofs << line << "\n";
    }
    lmat_ctx_destroy(ctx);
    std::cout << "query time: " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << std::endl;
    return 0;
}
