// dbbuild.cpp -- see dbbuild.hpp.  Pure host code (g++); the same libstdc++ std::priority_queue the
// reference uses decides the order of equal-rank taxids during pruning.
#include "dbbuild.hpp"
#include <algorithm>
#include <new>
#include <cstring>
#include <queue>

namespace lmat {

namespace {
struct RankPair {  // MyPair, src/kmerdb/SortedDb.hpp:129-139: ordered by rank value only
    unsigned int first;
    uint32_t second;
    RankPair(unsigned int f, uint32_t s) : first(f), second(s) {}
    bool operator<(const RankPair& o) const { return first < o.first; }
};
int tokbits(char c) {  // kencode.hpp:27-40 (unknown characters encode as A, with a message upstream)
    switch (c) {
        case 'a': case 'A': return 0;
        case 'c': case 'C': return 1;
        case 'g': case 'G': return 2;
        case 't': case 'T': return 3;
    }
    return 0;
}
}  // namespace

bool Ingest::load_idmap(const char* fn) {
    FILE* f = fopen(fn, "r");
    if (!f) { err = std::string("cannot read 16-bit map file ") + fn; return false; }
    int src;
    short dest;
    while (fscanf(f, "%d%hd", &src, &dest) > 0) br[(uint32_t)src] = (uint16_t)dest;  // make_db_table.cpp:259-273
    fclose(f);
    return true;
}

bool tree_node_ids(const char* tree_fn, std::vector<uint32_t>& ids, std::string& err, bool any_size) {
    FILE* f = fopen(tree_fn, "r");
    if (!f) { err = std::string("failed to open ") + tree_fn + " for reading"; return false; }
    // two comment lines, one count line, then "id nchild child.. parent" / name line pairs (TaxTree.hpp:24-57)
    char* line = nullptr;
    size_t cap = 0;
    for (int i = 0; i < 3; ++i) if (getline(&line, &cap, f) < 0) break;
    ids.clear();
    while (getline(&line, &cap, f) >= 0) {
        char* e = nullptr;
        const unsigned long long v = strtoull(line, &e, 10);
        if (e != line) ids.push_back((uint32_t)v);
        if (getline(&line, &cap, f) < 0) break;  // name
    }
    free(line);
    fclose(f);
    std::sort(ids.begin(), ids.end());
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    if (ids.size() > 65534 && !any_size) {
        err = "the taxonomy has " + std::to_string(ids.size()) + " nodes; without a 32->16 map (-f) the engine takes at most 65534 (make_db_image -t <tree> -M <map> makes one from the database itself)";
        return false;
    }
    return true;
}

bool idmap_from_database(const std::vector<std::string>& files, const char* tree_fn, uint32_t adaptor_tid,
                         std::vector<std::pair<uint32_t, uint16_t>>& out, std::string& err) {
    std::unordered_map<uint32_t, uint32_t> parent;
    {   // "id nchild child.. parent" / name line pairs after three header lines (TaxTree.hpp:24-57)
        FILE* f = fopen(tree_fn, "r");
        if (!f) { err = std::string("failed to open ") + tree_fn + " for reading"; return false; }
        char* line = nullptr;
        size_t cap = 0;
        for (int i = 0; i < 3; ++i) if (getline(&line, &cap, f) < 0) break;
        while (getline(&line, &cap, f) >= 0) {
            uint32_t first = 0, last = 0;
            int n = 0;
            for (char* p = line;;) {
                char* e = nullptr;
                const unsigned long long v = strtoull(p, &e, 10);
                if (e == p) break;
                if (!n++) first = (uint32_t)v;
                last = (uint32_t)v;
                p = e;
            }
            if (n >= 3) parent[first] = last;
            if (getline(&line, &cap, f) < 0) break;  // name
        }
        free(line);
        fclose(f);
    }
    std::unordered_set<uint32_t> used;
    for (uint32_t special : {1u, 9606u, adaptor_tid ? adaptor_tid : 32630u})
        if (parent.count(special)) used.insert(special);
    std::vector<uint32_t> tids;
    for (auto& fn : files) {
        FILE* in = fopen(fn.c_str(), "rb");
        if (!in) { err = "Error: unable to open kmer db [" + fn + "]"; return false; }
        uint32_t data_start, version, klen;
        uint64_t kmer_count, test;
        char loc;
        bool ok = fread(&data_start, 4, 1, in) == 1 && fread(&kmer_count, 8, 1, in) == 1 && fread(&test, 8, 1, in) == 1 &&
                  fread(&version, 4, 1, in) == 1 && fread(&loc, 1, 1, in) == 1 && fread(&klen, 4, 1, in) == 1;
        if (!ok || test != ~0ull || version != 999 || loc != 'N') { fclose(in); err = "not a tax_histo file: " + fn; return false; }
        for (uint64_t i = 0; i < kmer_count; ++i) {
            uint64_t kmer;
            uint16_t n;
            if (fread(&kmer, 8, 1, in) != 1) break;  // a short file ends the scan as it ends the ingest
            if (fread(&n, 2, 1, in) != 1) { fclose(in); err = "truncated tax_histo record"; return false; }
            tids.resize(n);
            if (n && fread(tids.data(), 4, n, in) != n) { fclose(in); err = "truncated taxid list"; return false; }
            for (uint32_t t : tids) used.insert(t);
            if ((i + 1) % 1500 == 0 && (fread(&test, 8, 1, in) != 1 || test != ~0ull)) { fclose(in); err = "tax_histo sanity word missing"; return false; }
        }
        fclose(in);
    }
    std::vector<uint32_t> all(used.begin(), used.end());
    for (size_t i = 0; i < all.size(); ++i) {  // ancestors; a node that is its own parent ends the walk (TaxTree.hpp:60-91)
        auto it = parent.find(all[i]);
        if (it != parent.end() && it->second != all[i] && used.insert(it->second).second) all.push_back(it->second);
    }
    if (all.size() > 65534) {
        err = "the database's taxids and their ancestors are " + std::to_string(all.size()) + " ids; the engine's ids are 16 bits wide (at most 65534)";
        return false;
    }
    std::sort(all.begin(), all.end());
    out.clear();
    for (size_t i = 0; i < all.size(); ++i) out.push_back(std::make_pair(all[i], (uint16_t)(i + 1)));
    return true;
}

bool save_idmap(const std::vector<std::pair<uint32_t, uint16_t>>& map, const char* fn) {
    FILE* f = fopen(fn, "w");
    if (!f) return false;
    for (auto& m : map) fprintf(f, "%u %u\n", m.first, (unsigned)m.second);  // "src dest" pairs, make_db_table.cpp:259-273
    return fclose(f) == 0;
}

bool Ingest::idmap_from_tree(const char* tree_fn) {
    std::vector<uint32_t> ids;
    if (!tree_node_ids(tree_fn, ids, err)) return false;
    br.clear();
    for (size_t i = 0; i < ids.size(); ++i) br[ids[i]] = (uint16_t)(i + 1);
    return true;
}

bool Ingest::set_options(int cutoff, const char* species_map_fn, const char* human_fn, const char* adaptor_fn,
                         uint32_t adaptor) {
    tid_cutoff = cutoff;
    adaptor_tid = adaptor ? adaptor : 32630;
    if (cutoff > 0 && species_map_fn && *species_map_fn) {  // make_db_table.cpp:303-313
        FILE* f = fopen(species_map_fn, "r");
        if (!f) { err = std::string("cannot read rank map ") + species_map_fn; return false; }
        int s, d;
        while (fscanf(f, "%d%d", &s, &d) > 0) species_map[(uint32_t)s] = (uint32_t)d;
        fclose(f);
    }
    if (human_fn && *human_fn) {
        human_fp = fopen(human_fn, "r");
        if (!human_fp) { err = std::string("cannot read human k-mer file ") + human_fn; return false; }
    }
    if (adaptor_fn && *adaptor_fn) {  // get_kmer_set, SortedDb.cpp:61-80
        FILE* f = fopen(adaptor_fn, "r");
        if (!f) { err = std::string("cannot read adaptor k-mer file ") + adaptor_fn; return false; }
        uint64_t km;
        while ((km = read_encode(f)) != ~0ull) adaptor_set.insert(km);
        fclose(f);
        adaptor_loaded = true;
    }
    return true;
}

uint64_t Ingest::read_encode(FILE* f) {  // SortedDb.cpp:39-59: whitespace-separated k-mer strings, forward 2-bit code
    char buf[64];
    int rc = fscanf(f, "%63s", buf);
    if (rc == EOF || strlen(buf) == 0) return ~0ull;
    uint64_t v = 0;
    for (int i = 0; i < k; ++i) v = (v << 2) | (uint64_t)tokbits(buf[i]);
    return v;
}

bool Ingest::to16(uint32_t tid, uint16_t& out, const char* what) {
    auto b = br.find(tid);
    const uint16_t t16 = b == br.end() ? 0 : b->second;
    if (t16 == 0 || t16 > br.size() + 1) {  // SortedDb.cpp:503-511
        err = std::string(what) + ": " + std::to_string(tid) + " " + std::to_string(t16);
        return false;
    }
    out = t16;
    return true;
}

void Ingest::push(uint64_t kmer, const std::vector<uint16_t>& lst) {
    uint32_t p;
    if (lst.size() == 1) {
        p = lst[0];
    } else {
        auto li = list_index.find(lst);
        if (li == list_index.end()) {
            p = 65536u + (uint32_t)lists.size();
            list_index[lst] = (uint32_t)lists.size();
            lists.push_back(lst);
        } else p = 65536u + li->second;
    }
    kmers.push_back(kmer);
    payload.push_back(p);
    if (flush && flush_every && kmers.size() >= flush_every && !flush(*this)) stream_failed = true;
}

bool Ingest::add_taxhisto(const char* fn) {
    FILE* in = fopen(fn, "rb");
    if (!in) { err = std::string("Error: unable to open kmer db [") + fn + "]"; return false; }
    fseek(in, 0, SEEK_END);
    const long fsz = ftell(in);
    fseek(in, 0, SEEK_SET);
    uint32_t data_start, version, klen;
    uint64_t kmer_count, test;
    char loc;
    bool ok = fread(&data_start, 4, 1, in) == 1 && fread(&kmer_count, 8, 1, in) == 1 && fread(&test, 8, 1, in) == 1 &&
              fread(&version, 4, 1, in) == 1 && fread(&loc, 1, 1, in) == 1 && fread(&klen, 4, 1, in) == 1;
    if (!ok || test != ~0ull) { fclose(in); err = "kmer data file is invalid; should have read 64 1s, but didn't"; return false; }
    if (version != 999 || loc != 'N') { fclose(in); err = "not a tax_histo file (version/location flag)"; return false; }
    if (k == 0) k = (int)klen;
    if ((int)klen != k) { fclose(in); err = "k-mer length of file differs from the database's"; return false; }
    if (human_fp && !human_primed) { last_human = read_encode(human_fp); human_primed = true; }  // SortedDb.cpp:107-112
    uint16_t HUMAN_16 = 0, ADAPTOR_16 = 0;
    { auto h = br.find(9606); if (h != br.end()) HUMAN_16 = h->second; }
    { auto a = br.find(adaptor_tid); if (a != br.end()) ADAPTOR_16 = a->second; }
    const uint16_t human_store = HUMAN_16 ? HUMAN_16 : (uint16_t)9606;
    const uint16_t adaptor_store = ADAPTOR_16 ? ADAPTOR_16 : (uint16_t)adaptor_tid;
    bool good = true;
    std::vector<uint32_t> tids;
    std::vector<uint16_t> lst;
    for (uint64_t i = 0; i < kmer_count && good; ++i) {
        if (ftell(in) == fsz) break;
        uint64_t kmer;
        uint16_t tid_count;
        if (fread(&kmer, 8, 1, in) != 1) { err = "truncated tax_histo record"; good = false; break; }
        if (last_kmer > 0 && kmer <= last_kmer) { err = "Kmers arriving out of order."; good = false; break; }
        if (k < 32 && (kmer >> (2 * k))) { err = "k-mer wider than 2k bits"; good = false; break; }
        while (last_human < kmer) {  // human k-mers absent from the tax_histo stream, SortedDb.cpp:170-222
            lst.assign(1, (adaptor_loaded && adaptor_set.count(last_human)) ? adaptor_store : human_store);
            push(last_human, lst);
            new_human++;
            last_human = read_encode(human_fp);
        }
        bool add_human = false;
        if (last_human == kmer) { matched_in++; add_human = true; last_human = read_encode(human_fp); }
        if (fread(&tid_count, 2, 1, in) != 1) { err = "truncated tax_histo record"; good = false; break; }
        tids.resize(tid_count);
        if (tid_count && fread(tids.data(), 4, tid_count, in) != tid_count) { err = "truncated taxid list"; good = false; break; }
        lst.clear();
        if (raw32) {  // SortedDb<uint32_t>::add_data without a 32->16 map (SortedDb.cpp:503-515,678-690): ids stored as read
            for (uint16_t j = 0; j < tid_count; ++j) { lst.push_back((uint16_t)tids[j]); lst.push_back((uint16_t)(tids[j] >> 16)); }
            if (tid_count) push(kmer, lst);
            if ((i + 1) % 1500 == 0) {
                if (fread(&test, 8, 1, in) != 1 || test != ~0ull) { err = "tax_histo sanity word missing"; good = false; break; }
            }
            last_kmer = kmer;
            continue;
        }
        if (adaptor_loaded && adaptor_set.count(kmer)) {  // SortedDb.cpp:275-292
            lst.push_back(adaptor_store);
        } else {
            uint16_t tmp = tid_count;
            std::priority_queue<RankPair> q;
            if (tid_cutoff > 0 && (int)tid_count > tid_cutoff) {  // SortedDb.cpp:296-409
                if (species_map.empty()) {
                    tmp = 0;
                } else {
                    for (uint16_t j = 0; j < tid_count; ++j) {
                        if (add_human && tids[j] == 9606) add_human = false;
                        q.push(RankPair(species_map[tids[j]], tids[j]));
                    }
                    if (add_human) { new_isect++; q.push(RankPair(species_map[9606], 9606)); matched_in--; }
                    while (!q.empty()) {
                        const unsigned cur = q.top().first;
                        while (q.top().first == cur) { q.pop(); if (q.empty()) break; }
                        if ((int)q.size() <= tid_cutoff) { tmp = (uint16_t)q.size(); break; }
                    }
                    if (q.size() == 0) { tmp = 1; q.push(RankPair(1, 1)); cut_kmers++; }
                }
            }
            uint16_t t16;
            if (tmp > 1) {
                if (q.size() > 1) {  // pruned list, highest rank value first (SortedDb.cpp:589-637)
                    reduced_kmers++;
                    for (int j = 0; j < tmp; ++j) {
                        if (!(good = to16(q.top().second, t16, "bad set"))) break;
                        q.pop();
                        lst.push_back(t16);
                    }
                } else {  // stored as read (SortedDb.cpp:640-712)
                    for (uint16_t j = 0; j < tid_count; ++j) {
                        if (tids[j] == 9606) add_human = false;
                        if (!(good = to16(tids[j], t16, "bad read"))) break;
                        lst.push_back(t16);
                    }
                    if (good && add_human) {
                        if (!(good = to16(9606, t16, "bad read"))) break;
                        lst.push_back(t16);
                        new_isect++;
                    }
                }
            } else if (tid_count == 1) {  // SortedDb.cpp:426-515
                if (add_human && tids[0] != 9606) {
                    doubles++;
                    matched_in--;
                    if (!(good = to16(tids[0], t16, "bad read"))) break;
                    lst.push_back(t16);
                    lst.push_back(HUMAN_16);
                } else {
                    singletons++;
                    if (!(good = to16(tids[0], t16, "bad single"))) break;
                    lst.push_back(t16);
                }
            } else if (tmp == 1) {  // pruned down to one taxid (SortedDb.cpp:517-533)
                if (!(good = to16(q.top().second, t16, "bad single"))) break;
                lst.push_back(t16);
                reduced_kmers++;
            } else {  // cut to the root, stored unmapped (SortedDb.cpp:534-538)
                lst.push_back(1);
                cut_kmers++;
            }
        }
        if (!good) break;
        push(kmer, lst);
        if ((i + 1) % 1500 == 0) {
            if (fread(&test, 8, 1, in) != 1 || test != ~0ull) { err = "tax_histo sanity word missing"; good = false; break; }
        }
        last_kmer = kmer;
    }
    fclose(in);
    return good;
}

bool Ingest::lookup(uint64_t kmer, std::vector<uint16_t>& out) const {
    auto it = std::lower_bound(kmers.begin(), kmers.end(), kmer);
    if (it == kmers.end() || *it != kmer) return false;
    const uint32_t p = payload[it - kmers.begin()];
    if (p < 65536u) out.assign(1, (uint16_t)p);
    else out = lists[p - 65536u];
    return true;
}

// image: "LMATIMG1" | u32 k | u64 n_kmers | u64 n_lists | kmers u64[] | payload u32[] | per list: u32 n, u16[n]
bool Ingest::save_image(const char* fn) const {
    FILE* f = fopen(fn, "wb");
    if (!f) return false;
    const uint32_t kk = (uint32_t)k;
    const uint64_t n = kmers.size(), nl = lists.size();
    bool ok = fwrite("LMATIMG1", 8, 1, f) == 1 && fwrite(&kk, 4, 1, f) == 1 && fwrite(&n, 8, 1, f) == 1 && fwrite(&nl, 8, 1, f) == 1;
    if (n) ok = ok && fwrite(kmers.data(), 8, n, f) == n && fwrite(payload.data(), 4, n, f) == n;
    for (uint64_t i = 0; i < nl && ok; ++i) {
        const uint32_t m = (uint32_t)lists[i].size();
        ok = fwrite(&m, 4, 1, f) == 1 && (m == 0 || fwrite(lists[i].data(), 2, m, f) == m);
    }
    return fclose(f) == 0 && ok;
}

// An image comes from a file the user names: sizes are checked against the file before anything is allocated, and every
// k-mer / payload against what the engine assumes (keys of 2k bits, ascending; payloads that are a taxid or an existing list).
namespace {
bool image_header(FILE* f, uint32_t& kk, uint64_t& n, uint64_t& nl, std::string& err) {
    char magic[8];
    if (!(fread(magic, 8, 1, f) == 1 && memcmp(magic, "LMATIMG1", 8) == 0 && fread(&kk, 4, 1, f) == 1 && fread(&n, 8, 1, f) == 1 &&
          fread(&nl, 8, 1, f) == 1)) { err = "malformed database image"; return false; }
    fseek(f, 0, SEEK_END);
    const uint64_t fsz = (uint64_t)ftell(f);
    fseek(f, 28, SEEK_SET);
    if (kk < 1 || kk > 32 || n > fsz / 12 || nl > fsz / 4 || 28 + 12 * n + 4 * nl > fsz) { err = "malformed database image (sizes exceed the file)"; return false; }
    return true;
}
bool read_lists(FILE* f, uint64_t nl, std::vector<std::vector<uint16_t>>& lists, uint64_t fsz_left) {
    lists.resize(nl);
    for (uint64_t i = 0; i < nl; ++i) {
        uint32_t m;
        if (fread(&m, 4, 1, f) != 1 || (uint64_t)m * 2 > fsz_left) return false;
        lists[i].resize(m);
        if (m && fread(lists[i].data(), 2, m, f) != m) return false;
        fsz_left -= (uint64_t)m * 2;
    }
    return true;
}
bool check_chunk(const std::vector<uint64_t>& km, const std::vector<uint32_t>& pay, int k, uint64_t nl, uint64_t& last) {
    for (size_t i = 0; i < km.size(); ++i) {
        if ((k < 32 && (km[i] >> (2 * k))) || (last != ~0ull && km[i] <= last)) return false;
        if (pay[i] >= 65536u && pay[i] - 65536u >= nl) return false;
        last = km[i];
    }
    return true;
}
}  // namespace

bool Ingest::load_image(const char* fn) {
    FILE* f = fopen(fn, "rb");
    if (!f) { err = std::string("cannot open image ") + fn; return false; }
    uint32_t kk = 0;
    uint64_t n = 0, nl = 0;
    bool ok = image_header(f, kk, n, nl, err);
    try {
        if (ok) {
            k = (int)kk;
            kmers.resize(n);
            payload.resize(n);
            if (n) ok = fread(kmers.data(), 8, n, f) == n && fread(payload.data(), 4, n, f) == n;
            const long here = ftell(f);
            fseek(f, 0, SEEK_END);
            const uint64_t left = (uint64_t)(ftell(f) - here);
            fseek(f, here, SEEK_SET);
            ok = ok && read_lists(f, nl, lists, left);
            uint64_t last = ~0ull;
            ok = ok && check_chunk(kmers, payload, k, nl, last);
        }
    } catch (const std::bad_alloc&) { ok = false; }
    fclose(f);
    if (!ok && err.empty()) err = "malformed database image";
    else if (ok && n) last_kmer = kmers.back();
    return ok;
}

bool Ingest::load_image_streaming(const char* fn, uint64_t* n_kmers) {
    FILE* f = fopen(fn, "rb");
    if (!f) { err = std::string("cannot open image ") + fn; return false; }
    uint32_t kk = 0;
    uint64_t n = 0, nl = 0;
    bool ok = image_header(f, kk, n, nl, err);
    const long base = 28;
    try {
        if (ok) {
            k = (int)kk;
            ok = fseek(f, base + (long)(12 * n), SEEK_SET) == 0;  // the lists sit behind the two arrays
            const long here = ftell(f);
            fseek(f, 0, SEEK_END);
            const uint64_t left = (uint64_t)(ftell(f) - here);
            fseek(f, here, SEEK_SET);
            ok = ok && read_lists(f, nl, lists, left);
        }
        const uint64_t chunk = flush_every ? flush_every : (1ull << 24);
        uint64_t last = ~0ull;
        for (uint64_t s = 0; s < n && ok; s += chunk) {
            const uint64_t m = std::min(chunk, n - s);
            kmers.resize(m);
            payload.resize(m);
            ok = fseek(f, base + (long)(8 * s), SEEK_SET) == 0 && fread(kmers.data(), 8, m, f) == m &&
                 fseek(f, base + (long)(8 * n + 4 * s), SEEK_SET) == 0 && fread(payload.data(), 4, m, f) == m &&
                 check_chunk(kmers, payload, k, nl, last);
            if (ok && flush) ok = flush(*this);
        }
    } catch (const std::bad_alloc&) { ok = false; }
    fclose(f);
    if (!ok && err.empty()) err = "malformed database image";
    if (n_kmers) *n_kmers = n;
    return ok;
}

}  // namespace lmat
