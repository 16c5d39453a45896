// nullmodel.cpp -- host loader of LMAT's null models for the -n scoring mode.
// Restates loadRandHits (src/read_label.cpp:512-678): list file lines `<kmer_count> <gz file relative to
// $LMAT_DIR>`; each gz file: first line num_bins, then per taxid `taxid <class>-<...> (num_obs max_val
// kmer_cnt) x num_bins`, with the zero-observation rules (:607-665), the "no_" -> genus rule (:594-601), the
// taxid 562 / 28384 special case (:622-629) and the rank tables gRank2num / gNum2rank (:519-547).
#include <hip/hip_runtime_api.h>
#include <zlib.h>
#include <algorithm>
#include <cstdlib>
#include <fstream>
#include <list>
#include <sstream>
#include "lmat_internal.hpp"

namespace lmat {

void free_null_models(lmat_ctx* c) {
    for (void* p : c->nm_allocs) hipFree(p);
    c->nm_allocs.clear();
    c->nm = NullModelDev();
}

template <class T>
static bool up(lmat_ctx* c, const std::vector<T>& v, const T** out) {
    void* d = nullptr;
    if (hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(T)) != hipSuccess) return false;
    if (!v.empty() && hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return false;
    c->nm_allocs.push_back(d);
    *out = (const T*)d;
    return true;
}

static bool gz_getline(gzFile f, std::string& out) {
    char buf[20004];  // the reference reads with a 20004-byte buffer (:577-585)
    if (!gzgets(f, buf, sizeof buf)) return false;
    out = buf;
    while (!out.empty() && out.back() == '\n') out.pop_back();
    return true;
}

int load_null_models(lmat_ctx* c, const char* list_fn) {
    free_null_models(c);
    if (!c->tax.loaded) return set_err(c, LMAT_E_ARG, "load the taxonomy before the null models");
    std::ifstream lst(list_fn);
    if (!lst) return set_err(c, LMAT_E_IO, std::string("Unexpected reading error (RandHits file list): ") + list_fn);
    // class strings -> ids; ids 0..10 are the strings the reference's rank tables know
    std::vector<std::string> cls_names = {"no_rank", "ethnic", "region", "species", "genus", "family", "order", "class",
                                          "phylum", "kingdom", "depth=0"};
    std::vector<uint8_t> cls_rank = {0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9};          // gRank2num
    std::vector<uint8_t> lower_cls = {0, 2, 3, 4, 5, 6, 7, 8, 9, 10};           // gNum2rank[0..9] ("no_rank" keeps key 0)
    auto cls_id = [&](const std::string& s) -> int {
        for (size_t i = 0; i < cls_names.size(); ++i)
            if (cls_names[i] == s) return (int)i;
        if (cls_names.size() >= 250) return -1;
        cls_names.push_back(s);
        cls_rank.push_back(0);  // gRank2num[unknown] default-inserts 0
        return (int)cls_names.size() - 1;
    };
    const uint32_t n_ids = c->tax.n + 1;
    std::vector<int> len_vec(1, 0);
    std::vector<std::pair<int, int>> len_to_table;  // (k-mer count, table) for files that were readable
    std::vector<std::vector<float>> vals;           // per table: [id * nb + bin]
    std::vector<std::vector<uint8_t>> clss;
    std::vector<int> nbins;
    int read_len;
    std::string file;
    while (lst >> read_len >> file) {
        const char* path = getenv("LMAT_DIR");
        if (path) file = std::string(path) + "/" + file;
        else fprintf(stderr, "WARNING! Missing LMAT_DIR environment variable!\n");
        printf("load: %d %s\n", read_len, file.c_str());
        len_vec.push_back(read_len);
        { std::ifstream pre(file.c_str()); if (!pre) { fprintf(stderr, "Unexpected reading error (RandHits file), skipping... %s\n", file.c_str()); continue; } }
        gzFile gz = gzopen(file.c_str(), "rb");
        if (!gz) continue;
        std::string line;
        gz_getline(gz, line);
        int nb = atoi(line.c_str());
        if (nb <= 0 || nb > 64) { gzclose(gz); return set_err(c, LMAT_E_IO, "null model with an unusable number of bins: " + file); }
        int t = -1;
        for (auto& lt : len_to_table) if (lt.first == read_len) t = lt.second;  // same class listed twice: same map upstream
        if (t < 0) {
            t = (int)vals.size();
            vals.push_back(std::vector<float>((size_t)n_ids * nb, 0.0f));
            clss.push_back(std::vector<uint8_t>(n_ids, 0xFF));
            nbins.push_back(nb);
            len_to_table.push_back(std::make_pair(read_len, t));
        } else if (nbins[t] != nb) { gzclose(gz); return set_err(c, LMAT_E_IO, "null models of one class disagree on the number of bins"); }
        // One row per taxid: `<taxid> <class>-<...>` then, per GC bin, (reads observed, largest score seen, k-mers of the
        // genome).  The row's value per bin (loadRandHits, src/read_label.cpp:585-665):
        //   observed            -> the largest score seen
        //   unobserved, k-mers >= 100000 -> 0.5
        //   unobserved, smaller genome   -> a "hole": the larger positive value among the two bins at the smallest
        //                                   distance that has one, 0.5 when the whole row is empty; holes are filled left
        //                                   to right and a filled hole serves the next ones
        //   taxid 28384 (the synthetic-construct bin) -> the row E. coli (562) has so far, class "genus"; its own holes are
        //                                   then filled over that copy
        std::vector<float> ecoli_row(nb, 0.5f);  // bins of taxid 562 seen so far in this file
        while (gz_getline(gz, line)) {
            std::istringstream is(line);
            uint32_t taxid = 0;
            std::string label;
            is >> taxid >> label;
            const size_t dash = label.find("-");
            if (dash == std::string::npos) { gzclose(gz); return set_err(c, LMAT_E_IO, "null-model class without '-': " + label); }
            std::string val = label.substr(0, dash);
            if (val.compare(0, 3, "no_") == 0) val = "genus";
            std::vector<float> cutoff(nb, 0.0f);
            std::vector<int> holes;
            float seen_max = 0;  // a field that fails to parse keeps the previous bin's value, as operator>> leaves it
            for (int bin = 0; bin < nb; ++bin) {
                int observed = 0, genome_kmers = 0;
                is >> observed >> seen_max >> genome_kmers;
                if (observed > 0) {
                    cutoff[bin] = seen_max;
                    if (taxid == 562) ecoli_row[bin] = seen_max;
                } else if (genome_kmers >= 100000) {
                    seen_max = 0.5f;
                    cutoff[bin] = 0.5f;
                } else {
                    holes.push_back(bin);
                }
            }
            if (taxid == 28384) { val = "genus"; cutoff = ecoli_row; }
            for (int h : holes) {
                for (int dist = 1; h - dist >= 0 || h + dist < nb; ++dist) {
                    const float left = h - dist >= 0 ? cutoff[h - dist] : 0.0f, right = h + dist < nb ? cutoff[h + dist] : 0.0f;
                    if (left > 0 || right > 0) cutoff[h] = std::max(left, right);
                    if (cutoff[h] > 0) break;
                }
                if (cutoff[h] <= 0) cutoff[h] = 0.5f;
            }
            auto it = c->tax.index_of.find(taxid);
            if (it == c->tax.index_of.end()) continue;  // taxid the database cannot produce
            const int ci = cls_id(val);
            if (ci < 0) { gzclose(gz); return set_err(c, LMAT_E_CAPACITY, "more than 250 distinct null-model classes"); }
            for (int bin = 0; bin < nb; ++bin) vals[t][(size_t)it->second * nb + bin] = cutoff[bin];
            clss[t][it->second] = (uint8_t)ci;
        }
        gzclose(gz);
    }
    std::sort(len_vec.begin(), len_vec.end());
    std::vector<int> len_avg;
    for (size_t i = 1; i < len_vec.size(); ++i) len_avg.push_back((len_vec[i - 1] + len_vec[i]) / 2);
    // table of each class, plus (last entry) the table of class 80, what getReadLen falls back to (:124-133)
    std::vector<int> len_table(len_vec.size() + 1, -1);
    auto table_of = [&](int len) { for (auto& lt : len_to_table) if (lt.first == len) return lt.second; return -1; };
    for (size_t i = 0; i < len_vec.size(); ++i) len_table[i] = table_of(len_vec[i]);
    len_table[len_vec.size()] = table_of(80);
    int nb_max = 1;
    for (int nb : nbins) nb_max = std::max(nb_max, nb);
    // flatten with a common bin stride
    std::vector<float> flat((size_t)vals.size() * n_ids * nb_max, 0.0f);
    std::vector<uint8_t> flat_cls((size_t)vals.size() * n_ids, 0xFF);
    for (size_t t = 0; t < vals.size(); ++t)
        for (uint32_t id = 0; id < n_ids; ++id) {
            flat_cls[t * n_ids + id] = clss[t][id];
            for (int b = 0; b < nbins[t]; ++b) flat[(t * n_ids + id) * nb_max + b] = vals[t][(size_t)id * nbins[t] + b];
        }
    NullModelDev d;
    hipSetDevice(c->device);
    bool ok = up(c, flat, &d.val) && up(c, flat_cls, &d.cls) && up(c, len_vec, &d.len_vec) && up(c, len_avg, &d.len_avg) &&
              up(c, len_table, &d.len_table) && up(c, nbins, &d.nbins) && up(c, cls_rank, &d.cls_rank) && up(c, lower_cls, &d.lower_cls);
    if (!ok) { free_null_models(c); return set_err(c, LMAT_E_NOMEM, "out of device memory for the null models"); }
    d.n_len = (int)len_vec.size();
    d.n_tables = (int)vals.size();
    d.nb_max = nb_max;
    d.n_cls = (int)cls_names.size();
    d.active = 1;
    c->nm = d;
    return LMAT_OK;
}

}  // namespace lmat
