// gene_oracle.hpp -- TEST INFRASTRUCTURE ONLY (like lmat_oracle.hpp: nothing under lmat_amd/ may include or link it).
// CPU restatement of the reference's gene_label (src/gene_label.cpp): the per-read vote over a k-mer -> gene-id-list
// database and the file driver around it.  Citations are file:line into the LMAT tree.
#pragma once
#include <zlib.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <list>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

namespace gene_oracle {

typedef uint64_t kmer_t;

// INDEXDB<uint32_t> filled by SortedDb<uint32_t>::add_data without a 32->16 map (SortedDb.cpp:84-751 with p_br_map == NULL:
// ids stored as read, :503-515, 678-690); the tax_histo record format of KmerFileMetaData.cpp:44-94 / tax_histo.cpp:257-281
struct GeneDb {
    int k = 0;
    std::unordered_map<kmer_t, std::vector<uint32_t>> table;
    bool add_taxhisto(const std::string& fn, std::string* err) {
        FILE* in = fopen(fn.c_str(), "rb");
        if (!in) { if (err) *err = "cannot open " + fn; return false; }
        uint32_t data_start, version, klen;
        uint64_t kmer_count, test;
        char loc;
        bool ok = fread(&data_start, 4, 1, in) == 1 && fread(&kmer_count, 8, 1, in) == 1 && fread(&test, 8, 1, in) == 1 &&
                  fread(&version, 4, 1, in) == 1 && fread(&loc, 1, 1, in) == 1 && fread(&klen, 4, 1, in) == 1;
        if (!ok || test != ~(uint64_t)0 || version != 999 || loc != 'N') { if (err) *err = "bad tax_histo header in " + fn; fclose(in); return false; }
        if (k == 0) k = (int)klen;
        std::vector<uint32_t> ids;
        for (uint64_t i = 0; i < kmer_count; ++i) {
            kmer_t kmer;
            uint16_t n;
            if (fread(&kmer, 8, 1, in) != 1 || fread(&n, 2, 1, in) != 1) break;
            ids.resize(n);
            if (n && fread(ids.data(), 4, n, in) != n) break;
            if (n) table[kmer] = ids;
            if ((i + 1) % 1500 == 0 && (fread(&test, 8, 1, in) != 1 || test != ~(uint64_t)0)) { if (err) *err = "sanity word missing"; fclose(in); return false; }
        }
        fclose(in);
        return true;
    }
};

struct GeneCall {
    bool any = false;      // geneid_lst non-empty: a line is written
    uint32_t gid = 0, top = 0, cnt = 0;
    float score = 0;
};

inline int encode_base(char c) {  // ENCODE, as in read_label.cpp:943-950 (gene_label.cpp uses the same macro)
    switch (c) { case 'a': case 'A': return 0; case 'c': case 'C': return 1; case 'g': case 'G': return 2; case 't': case 'T': return 3; }
    return -1;
}

// retrieve_kmer_labels (gene_label.cpp:218-267) + the vote of proc_line (:288-300)
inline GeneCall label_read(const GeneDb& db, const char* str, int slen, int klen) {
    GeneCall r;
    if (slen < klen) return r;  // :280-284: nothing printed
    std::list<uint32_t> geneid_lst;
    std::map<uint32_t, uint32_t> gene_track;
    unsigned valid_cnt = 0;
    int k = 0;
    const int highbits = (klen - 1) * 2;
    const kmer_t mask = ((kmer_t)1 << klen * 2) - 1;
    kmer_t forward = 0, reverse = 0;
    std::set<kmer_t> no_dups;
    for (int j = 0; j < slen; j++) {
        const int t = encode_base(str[j]);
        if (t < 0) { k = 0; continue; }
        forward = ((forward << 2) | (kmer_t)t) & mask;
        reverse = ((kmer_t)(t ^ 3) << highbits) | (reverse >> 2);
        if (++k >= klen) {
            const kmer_t kmer_id = forward < reverse ? forward : reverse;
            if (no_dups.find(kmer_id) != no_dups.end()) continue;
            no_dups.insert(kmer_id);
            ++valid_cnt;  // counts DISTINCT valid k-mers (:245-247)
            auto it = db.table.find(kmer_id);
            if (it == db.table.end()) continue;
            for (uint32_t gid : it->second) {
                if (gene_track.find(gid) == gene_track.end()) { gene_track.insert(std::make_pair(gid, 1u)); geneid_lst.push_back(gid); }
                else gene_track[gid] += 1;
            }
        }
    }
    if (geneid_lst.empty()) return r;
    std::vector<std::pair<uint32_t, uint32_t>> gsort;
    for (uint32_t g : geneid_lst) gsort.push_back(*gene_track.find(g));
    struct Cmp { bool operator()(const std::pair<uint32_t, uint32_t>& a, const std::pair<uint32_t, uint32_t> b) const { return a.second > b.second; } };
    std::sort(gsort.begin(), gsort.end(), Cmp());  // :296
    r.any = true;
    r.gid = gsort[0].first;
    r.top = gsort[0].second;
    r.cnt = valid_cnt;
    r.score = (float)gsort[0].second / (float)valid_cnt;
    return r;
}

// main() of gene_label.cpp (:540-706) for `-l <list of read_label .out files>`: one "thread" per file, <ofbase><i>.out per
// file, and the two summaries joined against the gene annotation table (-g, gzip).
// `label(file index, header, read)` stands for proc_line's lookup + vote (:269-300): the database for a real run, the votes
// printed in an earlier run's output files for replay_files below.
template <class Labeler>
inline bool run_files_with(Labeler label, const std::string& list_fn, const std::string& ofbase, const std::string& genefile,
                           float min_score, int min_kmer, float min_tax_score, std::string* err) {
    std::vector<std::string> files;
    { std::ifstream l(list_fn.c_str()); std::string f; while (l >> f) files.push_back(f); }
    const size_t nth = files.size();
    std::vector<std::map<uint32_t, std::map<uint32_t, uint32_t>>> track(nth), track_tax(nth);
    std::vector<std::map<uint32_t, std::map<uint32_t, float>>> score_track(nth), score_track_tax(nth);
    for (size_t th = 0; th < nth; ++th) {
        std::ifstream ifs(files[th].c_str());
        if (!ifs) { if (err) *err = "did not open for reading: " + files[th]; return false; }
        std::ostringstream nm;
        nm << ofbase << th << ".out";
        std::ofstream ofs(nm.str().c_str());
        bool finished = false;
        std::string line;
        while (!finished) {
            std::getline(ifs, line);
            if ((long)ifs.tellg() == -1) finished = true;  // :571-575: the last line is still processed
            const size_t p1 = line.find('\t');
            const std::string hdr = line.substr(0, p1);
            const size_t p2 = line.find('\t', p1 + 1);
            const std::string read_buff = line.substr(p1 + 1, p2 - p1 - 1);
            const size_t p3 = line.find('\t', p2 + 1);
            std::istringstream istrm2(line.substr(p2 + 1, p3 - p2 - 1));
            float score1 = 0, score2 = 0, score3 = 0;
            istrm2 >> score1 >> score2 >> score3;
            if (score3 == -1) continue;
            const size_t p4 = line.find('\t', p3 + 1);
            const size_t p5 = line.find('\t', p4 + 1);
            std::istringstream istrm(line.substr(p4 + 1, p5 - p4));
            uint32_t taxid = 0;
            float tax_score = 0.0;
            std::string match_type;
            istrm >> taxid >> tax_score >> match_type;
            if (match_type[0] == 'N' || match_type[0] == 'R') taxid = 0;
            // proc_line :269-313
            const GeneCall g = label(th, hdr, read_buff);
            // operator[] on the per-taxid maps happens before proc_line (:600-607): entries exist even when nothing is counted
            std::map<uint32_t, uint32_t>& gtrack = track[th][taxid];
            std::map<uint32_t, uint32_t>& gtrack_tax = track_tax[th][taxid];
            std::map<uint32_t, float>& sgtrack = score_track[th][taxid];
            std::map<uint32_t, float>& sgtrack_tax = score_track_tax[th][taxid];
            if (!g.any) continue;
            ofs << hdr << "\t" << read_buff << "\t" << taxid << " " << tax_score << "\t";
            ofs << "\t" << -1 << " " << g.top << " " << g.cnt << "\t" << g.gid << " " << g.score << " GL" << std::endl;
            if (g.score > min_score && (signed)g.cnt > min_kmer) { ++gtrack[g.gid]; sgtrack[g.gid] += g.score; }
            if (tax_score >= min_tax_score && g.score > min_score && (signed)g.cnt > min_kmer) { ++gtrack_tax[g.gid]; sgtrack_tax[g.gid] += g.score; }
        }
    }
    // doMerge / doMergeF (:135-186): gene -> label -> count / score, threads in order
    std::map<uint32_t, std::map<uint32_t, uint32_t>> merge_cnt, merge_cnt_tax;
    std::map<uint32_t, std::map<uint32_t, float>> score_merge, score_merge_tax;
    for (size_t th = 0; th < nth; ++th) {
        for (auto& a : track[th]) for (auto& b : a.second) merge_cnt[b.first][a.first] += b.second;
        for (auto& a : track_tax[th]) for (auto& b : a.second) merge_cnt_tax[b.first][a.first] += b.second;
        for (auto& a : score_track[th]) for (auto& b : a.second) {
            auto& m = score_merge[b.first];
            if (m.find(a.first) == m.end()) m[a.first] = b.second; else m[a.first] += b.second;
        }
        for (auto& a : score_track_tax[th]) for (auto& b : a.second) {
            auto& m = score_merge_tax[b.first];
            if (m.find(a.first) == m.end()) m[a.first] = b.second; else m[a.first] += b.second;
        }
    }
    gzFile gz = gzopen(genefile.c_str(), "rb");
    if (!gz) { if (err) *err = "Unable to unzip gene annotation table: " + genefile; return false; }
    std::ostringstream o1, o2;
    o1 << ofbase << "." << min_score << "." << min_kmer << ".genesummary";
    o2 << ofbase << "." << min_score << "." << min_kmer << ".genesummary.min_tax_score." << min_tax_score;
    std::ofstream sum_ofs(o1.str().c_str()), sum_ofs_tax(o2.str().c_str());
    char buff[20000];
    while (gzgets(gz, buff, sizeof buff)) {
        size_t n = strlen(buff);
        if (n && buff[n - 1] == '\n') buff[n - 1] = 0;  // istream::getline drops the newline (:677)
        std::istringstream istrm(buff);
        uint32_t tid = 0, gid = 0;
        istrm >> tid >> gid;
        if (merge_cnt.find(gid) != merge_cnt.end())
            for (auto& ti : merge_cnt[gid]) {
                const float avg = score_merge[gid][ti.first] / (float)ti.second;
                sum_ofs << avg << "\t" << ti.second << "\t" << ti.first << "\t" << buff << std::endl;
            }
        if (merge_cnt_tax.find(gid) != merge_cnt_tax.end())
            for (auto& ti : merge_cnt_tax[gid]) {
                const float avg = score_merge_tax[gid][ti.first] / (float)ti.second;
                sum_ofs_tax << avg << "\t" << ti.second << "\t" << ti.first << "\t" << buff << std::endl;
            }
    }
    gzclose(gz);
    return true;
}
inline bool run_files(const GeneDb& db, const std::string& list_fn, const std::string& ofbase, const std::string& genefile,
                      float min_score, int min_kmer, float min_tax_score, std::string* err) {
    return run_files_with([&](size_t, const std::string&, const std::string& read) { return label_read(db, read.c_str(), (int)read.length(), db.k); },
                          list_fn, ofbase, genefile, min_score, min_kmer, min_tax_score, err);
}
// The same run with every read's vote taken from the output files of an earlier gene_label run (`gl_list_fn`: one per input
// file, same order) instead of a database: the line is "hdr \t read \t tid score \t \t -1 top cnt \t gid gscore GL"
// (:299-300).  The reference's example run (example/example.tgz) is replayed this way by the tests.
inline bool replay_files(const std::string& list_fn, const std::string& gl_list_fn, const std::string& ofbase, const std::string& genefile,
                         float min_score, int min_kmer, float min_tax_score, std::string* err) {
    std::vector<std::map<std::string, GeneCall>> votes;
    std::ifstream l(gl_list_fn.c_str());
    std::string f;
    while (l >> f) {
        votes.emplace_back();
        std::ifstream in(f.c_str());
        if (!in) { if (err) *err = "did not open for reading: " + f; return false; }
        std::string line;
        while (std::getline(in, line)) {
            const size_t p1 = line.find('\t'), p2 = line.find('\t', p1 + 1), p3 = line.find('\t', p2 + 1), p4 = line.find('\t', p3 + 1),
                         p5 = line.find('\t', p4 + 1);
            if (p5 == std::string::npos) continue;
            GeneCall g;
            int minus1 = 0;
            std::istringstream a(line.substr(p4 + 1, p5 - p4 - 1)), b(line.substr(p5 + 1));
            a >> minus1 >> g.top >> g.cnt;
            b >> g.gid;
            g.any = true;
            g.score = (float)g.top / (float)g.cnt;
            votes.back()[line.substr(0, p1)] = g;
        }
    }
    return run_files_with([&](size_t th, const std::string& hdr, const std::string&) {
        GeneCall none;
        none.any = false;
        if (th >= votes.size()) return none;
        auto it = votes[th].find(hdr);
        return it == votes[th].end() ? none : it->second;
    }, list_fn, ofbase, genefile, min_score, min_kmer, min_tax_score, err);
}

}  // namespace gene_oracle
