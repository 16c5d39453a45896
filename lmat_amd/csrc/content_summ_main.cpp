// content_summ -- the consumer of read_label's .out files that bin/run_cs.sh runs (src/content_summ.cpp, argv at
// bin/run_cs.sh:148): the taxonomy tree of everything a run called, with read counts, and per rank the k-mer coverage
// of the reads called to a node -- how many distinct canonical k-mers (k = 8,10,12,14,17 by default of the script) those
// reads hold and how often each occurs.  No database lookups: a re-scan of the text records.  Same getopt letters, same
// report files (<o>, <o>.<rank>_kmer_cov) and stdout lines as upstream, byte for byte on the reference's example run
// (tests/test_content_summ.py).  Host code only; one worker per input file like upstream's OpenMP region (:361-413).
//
// Restated quirks that shape the files:
//   * the FIRST node of every rank the tree walk meets gets its <rank>_kmer_cov file created but no rows: upstream's new
//     stream lands in a shadowed variable (:522-531), so `kos` is still NULL for that node;
//   * rows are written only for nodes with more than one read (:534);
//   * a called taxid without a rank-table entry reads as rank "" (operator[] inserts): a file "<o>._kmer_cov" appears;
//   * strain -> species folding uses the FIRST species on the path (map::insert keeps the first, :342-351);
//   * a read's k-mers are counted once per read (set no_dups, :131-150).
#include <getopt.h>
#include <algorithm>
#include <chrono>
#include <cstring>
#include <fstream>
#include <iostream>
#include <list>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>
#include "outfmt.hpp"

#define LMAT_VERSION "1.2.4_2018a"

typedef uint32_t tid_t;
typedef std::map<uint64_t, uint64_t> kmer_cnt_t;

static std::unordered_set<int> gLowNumPlasmid;
static bool is_plasmid(tid_t t) { return (t >= 10000000 && t < 11000000) || gLowNumPlasmid.count((int)t); }  // :58
static bool is_human(tid_t t) { return t == 9606 || t == 63221 || t == 741158; }                              // tid_checks.hpp:15-28

// TaxTree<TID_T>(file) + getPathToRoot + getName (src/kmerdb/TaxTree.hpp:24-95, TaxNode.hpp:131-147)
struct Tree {
    std::unordered_map<tid_t, tid_t> parent;
    std::unordered_map<tid_t, std::string> name;
    bool load(const char* fn) {
        std::ifstream in(fn);
        if (!in.is_open()) { std::cerr << "failed to open " << fn << " for reading\n"; return false; }
        std::string line;
        std::getline(in, line);
        std::getline(in, line);
        int count;
        in >> count;
        std::getline(in, line);
        while (true) {
            std::streampos p = in.tellg();
            if (in.eof() || !in.good() || (int)p == -1) break;
            tid_t id = 0, ct = 0, c = 0, par = 0;
            in >> id >> ct;
            for (tid_t j = 0; j < ct; ++j) in >> c;
            in >> par;
            std::string nm;
            std::getline(in, nm);
            std::getline(in, nm);
            if (!in && nm.empty() && id == 0) break;
            parent[id] = par;
            name[id] = nm;
        }
        return true;
    }
    void path_to_root(tid_t t, std::vector<tid_t>& out) const {  // parent .. root
        out.clear();
        auto it = parent.find(t);
        size_t guard = 0;
        while (it != parent.end() && it->second != t && guard++ < 100000) {
            t = it->second;
            out.push_back(t);
            it = parent.find(t);
        }
    }
    std::string get_name(tid_t t) const { auto it = name.find(t); return it == name.end() ? "" : it->second; }
};

// the distinct canonical k-mers of one read, for every tracked k, counted once per read (:113-152)
static void store_kmers(const std::string& s, const std::vector<int>& klen, tid_t taxid, std::vector<std::map<tid_t, kmer_cnt_t>>& track) {
    const unsigned kls = (unsigned)klen.size();
    std::vector<int> k(kls, 0), high(kls, 0);
    std::vector<uint64_t> mask(kls, 0), fwd(kls, 0), rev(kls, 0);
    std::vector<std::set<uint64_t>> no_dups(kls);
    for (unsigned i = 0; i < kls; ++i) { high[i] = (klen[i] - 1) * 2; mask[i] = ((uint64_t)1 << klen[i] * 2) - 1; }
    for (size_t j = 0; j < s.size(); ++j) {
        int t;
        switch (s[j]) {
            case 'a': case 'A': t = 0; break;
            case 'c': case 'C': t = 1; break;
            case 'g': case 'G': t = 2; break;
            case 't': case 'T': t = 3; break;
            default: t = -1;
        }
        for (unsigned i = 0; i < kls; ++i) {
            if (t < 0) { k[i] = 0; continue; }
            fwd[i] = ((fwd[i] << 2) | (uint64_t)t) & mask[i];
            rev[i] = ((uint64_t)(t ^ 3) << high[i]) | (rev[i] >> 2);
            if (++k[i] >= klen[i]) {
                const uint64_t id = fwd[i] < rev[i] ? fwd[i] : rev[i];
                if (!no_dups[i].insert(id).second) continue;
                track[i][taxid][id] += 1;
            }
        }
    }
}

static void comp_kmer_cov(const std::vector<std::vector<std::map<tid_t, kmer_cnt_t>>>& track, tid_t tid, std::ofstream& ofs,
                          const std::vector<int>& kv) {  // :538-571
    for (unsigned ksi = 0; ksi < kv.size(); ++ksi) {
        kmer_cnt_t merge;
        uint64_t kmer_cnt = 0;
        int kcnt_sum = 0;
        for (auto& th : track) {
            auto it = th[ksi].find(tid);
            if (it == th[ksi].end()) continue;
            for (auto& kc : it->second) {
                kcnt_sum += (int)kc.second;
                auto m = merge.find(kc.first);
                if (m == merge.end()) { merge[kc.first] = kc.second; ++kmer_cnt; }
                else m->second += kc.second;
            }
        }
        std::vector<unsigned> ids;
        std::map<unsigned, unsigned> hist;
        for (auto& kc : merge) {
            const int cnt = (int)kc.second;
            if (!hist.count(cnt)) { hist[cnt] = 1; ids.push_back(cnt); }
            else hist[cnt] += 1;
        }
        std::sort(ids.begin(), ids.end());
        ofs << "taxid=" << tid << " distinct_kmer_cnt=" << kmer_cnt << " k_size=" << kv[ksi] << " tot_kmer_cnt=" << kcnt_sum << std::endl;
        for (unsigned id : ids) ofs << tid << " " << kv[ksi] << " " << id << " " << hist[id] << std::endl;
    }
}

int main(int argc, char* argv[]) {
    float threshold = 0.0f;
    std::string query_fn_lst, lmat_sum, ofbase, tax_tree_fn, rank_table_file, low_num_plasmid_file, k_size_str, rank_check_str;
    bool skip_human = false, do_human_reg = false;
    int c;
    while ((c = getopt(argc, argv, "m:f:a:h:njb:ye:wp:k:c:v:k:i:d:l:t:sr:o:x:f:q:V")) != -1) {
        switch (c) {
            case 'n': do_human_reg = true; break;
            case 'a': rank_check_str = optarg; break;
            case 'p': low_num_plasmid_file = optarg; break;
            case 's': skip_human = true; break;
            case 'r': rank_table_file = optarg; break;
            case 'y': break;  // verbose dumps are not produced
            case 'l': lmat_sum = optarg; break;
            case 'v': threshold = (float)atof(optarg); break;
            case 'c': tax_tree_fn = optarg; break;
            case 'k': k_size_str = optarg; break;
            case 'f': query_fn_lst = optarg; break;
            case 'i': break;
            case 'o': ofbase = optarg; break;
            case 'V': std::cout << "LMAT version " << LMAT_VERSION << "\n"; return 0;
            default: std::cout << "Unrecognized option: " << (char)c << ", ignore." << std::endl;
        }
    }
    std::vector<int> k_size;
    if (k_size_str.empty()) k_size = {8, 10, 14, 20};
    else {
        std::stringstream ss(k_size_str);
        std::string tok;
        while (std::getline(ss, tok, ',')) if (!tok.empty()) k_size.push_back(atoi(tok.c_str()));
    }
    for (int k : k_size) if (k < 1 || k > 31) { std::cerr << "k sizes must lie in 1..31" << std::endl; return -1; }
    std::set<std::string> rank_check;
    {
        std::stringstream ss(rank_check_str);
        std::string tok;
        while (std::getline(ss, tok, ',')) if (!tok.empty()) { std::cout << "rank store: [" << tok << "]" << std::endl; rank_check.insert(tok); }
    }
    for (int k : k_size) std::cout << "track k size=" << k << std::endl;
    if (!low_num_plasmid_file.empty()) {
        std::ifstream ifs(low_num_plasmid_file.c_str());
        if (!ifs) std::cerr << "Unexpected reading error: " << low_num_plasmid_file << std::endl;
        tid_t pid;
        while (ifs >> pid) gLowNumPlasmid.insert((int)pid);
    }
    std::unordered_map<tid_t, std::string> rank_table;
    if (!rank_table_file.empty()) {
        std::ifstream ifs(rank_table_file.c_str());
        tid_t t;
        std::string rank;
        while (ifs >> t >> rank) rank_table.insert(std::make_pair(t, rank));
    }
    std::vector<std::string> files;
    {
        std::ifstream ifs(query_fn_lst.c_str());
        std::string fn;
        while (ifs >> fn) files.push_back(fn);
    }
    const int n_threads = (int)files.size();
    std::cout << "set threads=" << n_threads << std::endl;
    if (n_threads < 1) { std::cerr << "no input files in [" << query_fn_lst << "]" << std::endl; return -1; }
    std::cout << "Read taxonomy tree: " << tax_tree_fn << std::endl;
    Tree tree;
    if (!tree.load(tax_tree_fn.c_str())) return -1;
    std::cout << "Done Read taxonomy tree: " << tax_tree_fn << std::endl;
    std::map<tid_t, float> weighted_readcnt;
    std::map<tid_t, int> read_cnts;
    std::ifstream call_ifs(lmat_sum.c_str());
    if (!call_ifs) { std::cerr << "Failed to open " << lmat_sum << " must exit now" << std::endl; return -1; }
    std::list<tid_t> clst;
    std::unordered_map<tid_t, tid_t> strain2spec;
    auto is_fold_rank = [&](tid_t t) {
        const std::string& r = rank_table[t];
        return (r == "species" && !do_human_reg) || (r == "region" && do_human_reg);
    };
    {
        std::string buff;
        while (std::getline(call_ifs, buff)) {
            if (buff.find("\tNULL\t") != std::string::npos) continue;
            std::istringstream istrm(buff);
            tid_t tid = 0;
            unsigned read_cnt = 0;
            std::string descrip;
            float wght_rc = 0;
            istrm >> wght_rc >> read_cnt >> tid >> descrip;
            weighted_readcnt.insert(std::make_pair(tid, wght_rc));
            read_cnts.insert(std::make_pair(tid, (int)read_cnt));
            if (is_fold_rank(tid)) strain2spec.insert(std::make_pair(tid, tid));
            if (!is_plasmid(tid)) {
                std::vector<tid_t> ptor;
                tree.path_to_root(tid, ptor);
                for (tid_t a : ptor) if (is_fold_rank(a)) strain2spec.insert(std::make_pair(tid, a));
            }
            clst.push_back(tid);
        }
    }
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::vector<std::map<tid_t, kmer_cnt_t>>> kmer_track(n_threads, std::vector<std::map<tid_t, kmer_cnt_t>>(k_size.size()));
    {
        std::vector<std::thread> th;
        std::vector<int> bad(n_threads, 0);
        for (int t = 0; t < n_threads; ++t)
            th.emplace_back([&, t]() {
                std::ifstream ifs(files[t].c_str());
                if (!ifs) { bad[t] = 1; return; }
                std::string line;
                while (std::getline(ifs, line)) {
                    const size_t p1 = line.find('\t'), p2 = line.find('\t', p1 + 1), p3 = line.find('\t', p2 + 1), p4 = line.find('\t', p3 + 1),
                                 p5 = line.find('\t', p4 + 1);
                    if (p1 == std::string::npos || p2 == std::string::npos) continue;
                    const std::string read_buff = line.substr(p1 + 1, p2 - p1 - 1);
                    const std::string call = line.substr(p4 + 1, p5 - p4 - 1);  // (npos + 1 == 0: a record without the candidate column reads from its start, as upstream)
                    if (call.empty() || call[0] == 'N' || call[0] == 'R') continue;
                    std::istringstream istrm(call);
                    float score = 0;
                    tid_t taxid = 0;
                    std::string match_type;
                    istrm >> taxid >> score >> match_type;
                    if (is_human(taxid) && skip_human) continue;
                    if (score < threshold) continue;
                    tid_t use_tid = taxid;
                    auto s2 = strain2spec.find(taxid);
                    if (s2 != strain2spec.end() && !is_plasmid(taxid)) use_tid = s2->second;
                    auto rk = rank_table.find(use_tid);
                    const std::string rnk = rk != rank_table.end() ? rk->second : "undef";
                    if (rank_check.count(rnk) || is_plasmid(taxid)) store_kmers(read_buff, k_size, use_tid, kmer_track[t]);
                }
            });
        for (auto& x : th) x.join();
        for (int t = 0; t < n_threads; ++t)
            if (bad[t]) { std::cerr << "did not open for reading: [" << files[t] << "] tid: [" << t << "]" << std::endl; return -1; }
    }
    // the tree of the called taxids (:415-440) and its walk, children in reverse order of arrival (:447-536)
    std::set<tid_t> seen;
    std::ofstream ofs(ofbase.c_str());
    std::map<tid_t, std::list<tid_t>> child;
    for (tid_t tid : clst) {
        std::vector<tid_t> ptor;
        tree.path_to_root(tid, ptor);
        tid_t child_node = tid;
        for (tid_t ptid : ptor) {
            if (!seen.count(child_node)) {
                seen.insert(child_node);
                child[ptid].push_back(child_node);
            }
            child_node = ptid;
        }
    }
    ofs << "Name\tTaxID\tReads\tWReads" << std::endl;
    std::map<tid_t, std::string> tab_lst;
    std::list<tid_t> open;
    open.push_back(1);  // "should always be root"
    std::map<std::string, std::ofstream*> rank_ofs;
    while (!open.empty()) {
        const tid_t tid = open.front();
        open.pop_front();
        const std::list<tid_t>& lst = child[tid];
        const std::string chk = tab_lst[tid] + "\t";
        for (tid_t ch : lst) { tab_lst[ch] = chk; open.push_front(ch); }
        const unsigned tot_read_cnt = (unsigned)read_cnts[tid];
        float wrdc = 0;
        if (tot_read_cnt > 0) {
            wrdc = weighted_readcnt[tid];
            std::string rank = rank_table[tid];
            if (rank != "no_rank") {
                if (is_plasmid(tid)) rank = "plasmid";
                std::ofstream* kos = nullptr;
                auto it = rank_ofs.find(rank);
                if (it != rank_ofs.end()) kos = it->second;
                else {
                    const std::string fn = ofbase + "." + rank + "_kmer_cov";
                    std::ofstream* made = new std::ofstream(fn.c_str());
                    if (!(*made)) std::cout << "Unable to write to " << fn << " will try to continue" << std::endl;
                    rank_ofs.insert(std::make_pair(rank, made));  // upstream's `kos` stays NULL here: the first node of a rank writes no rows
                }
                if (kos && tot_read_cnt > 1) comp_kmer_cov(kmer_track, tid, *kos, k_size);
            }
        }
        std::string s = tab_lst[tid];
        s += tree.get_name(tid);
        s += '\t'; lmat::put_int(s, (long long)tid);
        s += '\t'; lmat::put_int(s, (long long)tot_read_cnt);
        s += '\t'; lmat::put_float(s, wrdc);
        ofs << s << std::endl;
    }
    for (auto& p : rank_ofs) { p.second->close(); delete p.second; }
    std::cout << "query time: " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << std::endl;
    return 0;
}
