// rollups.hpp -- the two roll-ups bin/run_rl.sh makes from a run's .fastsummary right after read_label
// (bin/run_rl.sh:251-252), here part of the writer: read_label writes them itself when LMAT_ROLLUPS names the ranks,
// and `fs_rollup` makes them from an existing .fastsummary.
//   write_lineage       = bin/tolineage.py: per called taxid with more than `num` reads, the read count and the names of
//                         its ranked lineage (the -u names file), "no rank" levels dropped, tab-separated (Krona's input);
//   write_rank_reports  = bin/fsreport.py: per rank of interest (plasmid,species,genus) every call folded into its
//                         ancestor of that rank: average / total read score, read count, optionally the rRNA share and
//                         gene counts of a gene_label summary, and for a species its best strain.
// Numbers are printed as the scripts print them under the Python 2 the reference's example was made with:
// str(float) = "%.12g" (plus ".0" for integral values), "%.4f" for the averages.
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace lmat {

inline std::string py2_str_float(double v) {  // Python 2 str(float)
    char b[64];
    snprintf(b, sizeof b, "%.12g", v);
    std::string s(b);
    if (s.find_first_of(".en") == std::string::npos) s += ".0";   // "12" -> "12.0" (inf/nan have an 'n')
    return s;
}
inline std::vector<std::string> split_tabs(const std::string& s) {
    std::vector<std::string> out;
    size_t a = 0;
    for (;;) {
        const size_t b = s.find('\t', a);
        out.push_back(s.substr(a, b == std::string::npos ? std::string::npos : b - a));
        if (b == std::string::npos) break;
        a = b + 1;
    }
    return out;
}
inline std::vector<std::string> split_ws(const std::string& s) {
    std::vector<std::string> out;
    std::istringstream is(s);
    std::string t;
    while (is >> t) out.push_back(t);
    return out;
}

// bin/tolineage.py <names file> <fastsummary> <out> <num> <min_avg>
inline bool write_lineage(const std::string& names_fn, const std::string& fastsummary_fn, const std::string& out_fn, int num, double min_avg,
                          std::string* err) {
    std::ifstream a(names_fn.c_str());
    if (!a) { if (err) *err = "cannot read " + names_fn; return false; }
    std::unordered_map<std::string, std::string> tax;  // ktaxid -> whole line (with its newline, as the script keeps it)
    std::string line;
    while (std::getline(a, line)) {
        // t = line.split(',')[2].split('=')[1]
        size_t c1 = line.find(','), c2 = c1 == std::string::npos ? c1 : line.find(',', c1 + 1);
        if (c2 == std::string::npos) continue;
        const size_t c3 = line.find(',', c2 + 1);
        const std::string f = line.substr(c2 + 1, c3 == std::string::npos ? std::string::npos : c3 - c2 - 1);
        const size_t eq = f.find('=');
        if (eq == std::string::npos) continue;
        const size_t eq2 = f.find('=', eq + 1);
        tax[f.substr(eq + 1, eq2 == std::string::npos ? std::string::npos : eq2 - eq - 1)] = line + "\n";
    }
    std::ifstream fs(fastsummary_fn.c_str());
    if (!fs) { if (err) *err = "cannot read " + fastsummary_fn; return false; }
    std::ofstream out(out_fn.c_str());
    if (!out) { if (err) *err = "cannot write " + out_fn; return false; }
    while (std::getline(fs, line)) {
        const std::vector<std::string> t = split_ws(line);
        if (t.size() < 3) continue;
        const std::string& count = t[1];
        const double avg = atof(t[0].c_str()) / atof(t[1].c_str());
        const std::string& ktaxid = t[2];
        auto it = tax.find(ktaxid);
        if (it == tax.end()) continue;  // the script reports the entry on stdout and goes on
        std::vector<std::string> e2;
        if (atoi(ktaxid.c_str()) == 1) e2.push_back("Root,Root\n");
        else {
            const std::string& e = it->second;
            const size_t j = e.find('\t');
            if (j == std::string::npos) e2.push_back("Root,Root\n");
            else e2 = split_tabs(e.substr(j + 1));
        }
        if (atoi(count.c_str()) > num && avg >= min_avg) {
            out << count << '\t';
            for (size_t i = 0; i + 1 < e2.size(); ++i)
                if (e2[i].find("no rank") == std::string::npos) {
                    const size_t c = e2[i].find(',');
                    const size_t d = c == std::string::npos ? c : e2[i].find(',', c + 1);
                    if (c != std::string::npos) out << e2[i].substr(c + 1, d == std::string::npos ? std::string::npos : d - c - 1) << '\t';
                }
            const std::string& last = e2.back();
            const size_t j = last.find(',');
            out << (j == std::string::npos ? last : last.substr(j + 1));  // (find == -1: the script prints from index 0 as well)
        }
    }
    return true;
}

struct RollupInputs {
    std::string tree_fn, rank_fn, plasmid_fn, plasmid_names_fn;  // the four files fsreport.py takes from $LMAT_DIR (the last two may be absent)
    std::string gene_summary_fn;                                  // optional: gene_label's <..>.genesummary.min_tax_score.<x>
    int min_gene_cnt = 2;
};

// bin/fsreport.py <fastsummary> <ranks> <odir> [<gene summary> <min gene reads>]: one file <odir>/<basename(fastsummary)>.<rank> per rank
inline bool write_rank_reports(const std::string& fastsummary_fn, const std::string& rank_list, const std::string& odir, const RollupInputs& in,
                               std::string* err) {
    std::unordered_set<std::string> plasmids;
    {
        std::ifstream a(in.plasmid_fn.c_str());
        std::string l;
        while (std::getline(a, l)) { while (!l.empty() && isspace((unsigned char)l.back())) l.pop_back(); plasmids.insert(l); }
    }
    std::unordered_map<std::string, std::string> plasname;
    {
        std::ifstream a(in.plasmid_names_fn.c_str());
        std::string l;
        while (std::getline(a, l)) { const auto v = split_tabs(l); if (!v.empty()) plasname[v[0]] = v.back(); }
    }
    auto is_plasmid = [&](const std::string& tid) { const long v = atol(tid.c_str()); return plasmids.count(tid) || (v >= 10000000 && v < 20000000); };
    std::unordered_map<std::string, std::string> ranktable;
    {
        std::ifstream a(in.rank_fn.c_str());
        if (!a) { if (err) *err = "cannot read " + in.rank_fn; return false; }
        std::string l;
        while (std::getline(a, l)) { const auto v = split_ws(l); if (v.size() >= 2) ranktable[v[0]] = v[1]; }
    }
    std::unordered_map<std::string, std::string> parent, names;
    {
        std::ifstream a(in.tree_fn.c_str());
        if (!a) { if (err) *err = "cannot read " + in.tree_fn; return false; }
        std::string f, name;
        std::getline(a, f); std::getline(a, f); std::getline(a, f);
        while (std::getline(a, f)) {
            if (!std::getline(a, name)) name.clear();
            const auto t = split_ws(f);
            if (t.empty()) break;
            parent[t[0]] = t.back();
            names[t[0]] = name;
        }
    }
    std::vector<std::string> rank_lst;
    { std::stringstream ss(rank_list); std::string r; while (std::getline(ss, r, ',')) rank_lst.push_back(r); }
    auto rank_tid = [&](const std::string& rank, const std::string& tid) -> std::string {  // getRankTid; "" = none
        auto r0 = ranktable.find(tid);
        if ((r0 != ranktable.end() && r0->second == rank) || (rank == "plasmid" && is_plasmid(tid))) return tid;
        std::string s = tid;
        size_t guard = 0;
        for (;;) {
            auto p = parent.find(s);
            if (p == parent.end() || p->second == s || guard++ > 100000) return "";
            auto r = ranktable.find(s);
            if (r != ranktable.end() && r->second == rank) return s;
            s = p->second;
        }
    };
    struct Call { std::string taxid, wrc, count; };
    std::map<std::string, std::vector<std::pair<std::string, std::vector<Call>>>> store;  // rank -> [(tid, calls)] in order of first appearance
    std::unordered_map<std::string, std::string> orig;
    {
        std::ifstream a(fastsummary_fn.c_str());
        if (!a) { if (err) *err = "cannot read " + fastsummary_fn; return false; }
        std::string l;
        while (std::getline(a, l)) {
            while (!l.empty() && isspace((unsigned char)l.back())) l.pop_back();
            const auto t = split_tabs(l);
            if (t.size() < 4) continue;
            orig[t[2]] = t[3];
            if (!parent.count(t[2])) parent[t[2]] = "1";
            for (const std::string& rank : rank_lst) {
                const std::string tid = rank_tid(rank, t[2]);
                if (tid.empty()) continue;
                auto& v = store[rank];
                size_t i = 0;
                while (i < v.size() && v[i].first != tid) ++i;
                if (i == v.size()) v.push_back({tid, {}});
                v[i].second.push_back({t[2], t[0], t[1]});
            }
        }
    }
    std::map<std::string, std::map<std::string, long>> rrna_sum;                          // rank -> tid -> reads on rRNA genes
    std::map<std::string, std::map<std::string, std::map<std::string, long>>> gene_cnt;   // rank -> tid -> gene -> reads
    const bool genes = !in.gene_summary_fn.empty();
    if (genes) {
        std::ifstream a(in.gene_summary_fn.c_str());
        if (!a) { if (err) *err = "cannot read " + in.gene_summary_fn; return false; }
        std::string l;
        while (std::getline(a, l)) {
            while (!l.empty() && isspace((unsigned char)l.back())) l.pop_back();
            const auto t = split_tabs(l);
            if (t.size() < 8 || t[2] == "0") continue;
            if (!parent.count(t[2])) parent[t[2]] = "1";
            for (const std::string& rank : rank_lst) {
                const std::string tid = rank_tid(rank, t[2]);
                if (tid.empty()) continue;
                if (t[7] == "rRNA") rrna_sum[rank][tid] += atol(t[1].c_str());
                if (atol(t[1].c_str()) > in.min_gene_cnt) gene_cnt[rank][tid][t[4]] += atol(t[1].c_str());
            }
        }
    }
    std::string base = fastsummary_fn;
    { const size_t s = base.rfind('/'); if (s != std::string::npos) base = base.substr(s + 1); }
    for (auto& rs : store) {
        const std::string& rank = rs.first;
        std::ofstream fh((odir + "/" + base + "." + rank).c_str());
        if (!fh) { if (err) *err = "cannot write a report into " + odir; return false; }
        struct Row { double wrc_sum; long count_sum; std::string tid, name; long rrna; size_t n_genes; long gene_reads; std::string strain_info; };
        std::vector<Row> save;
        for (auto& tc : rs.second) {
            const std::string& tid = tc.first;
            std::string name_str;
            if (plasmids.count(tid) && plasname.count(tid) && rank == "plasmid") name_str = plasname[tid];
            else if (orig.count(tid)) name_str = orig[tid];
            else name_str = names[tid];
            const size_t idx = name_str.find(',');
            if (idx != std::string::npos) name_str = name_str.substr(idx + 1);
            const auto& lst = tc.second;
            if (lst.size() == 1 && is_plasmid(tid) && rank != "plasmid") continue;
            double best_wrc = -1, wrc_sum = 0;
            std::string best_count = "-1", top_strain;
            long count_sum = 0;
            for (const Call& cl : lst) {
                if (is_plasmid(cl.taxid)) ranktable[cl.taxid] = "plasmid";
                wrc_sum += atof(cl.wrc.c_str());
                count_sum += atol(cl.count.c_str());
                if (rank == "species" && ranktable[cl.taxid] == "strain" && best_wrc < atof(cl.wrc.c_str())) {
                    top_strain = cl.taxid;
                    best_wrc = atof(cl.wrc.c_str());
                    best_count = cl.count;
                }
            }
            std::string strain_info;
            if (!top_strain.empty()) strain_info = "\t" + py2_str_float(best_wrc) + "\t" + best_count + "\t" + top_strain + "\t" + orig[top_strain];
            Row r{wrc_sum, count_sum, tid, name_str, 0, 0, 0, strain_info};
            if (rrna_sum.count(rank) && rrna_sum[rank].count(tid)) r.rrna = rrna_sum[rank][tid];
            if (gene_cnt.count(rank) && gene_cnt[rank].count(tid)) {
                r.n_genes = gene_cnt[rank][tid].size();
                for (auto& g : gene_cnt[rank][tid]) r.gene_reads += g.second;
            }
            save.push_back(r);
        }
        std::stable_sort(save.begin(), save.end(), [](const Row& a, const Row& b) { return a.wrc_sum > b.wrc_sum; });
        fh << (genes ? "Average Read Score\tTotal Read Score\tRead Count\tPcnt. rRNA\tNo. Genes\tNo. Gene Reads\tTaxID\tName\tStrain Info"
                     : "Average Read Score\tTotal Read Score\tRead Count\tTaxID\tName\tStrain Info") << "\n";
        for (const Row& r : save) {
            char avg[64], pc[64];
            snprintf(avg, sizeof avg, "%.4f", r.wrc_sum / (double)r.count_sum);
            fh << avg << "\t" << py2_str_float(r.wrc_sum) << "\t" << r.count_sum << "\t";
            if (genes) {
                snprintf(pc, sizeof pc, "%.4f", (double)r.rrna / (double)r.count_sum);
                fh << pc << "\t" << r.n_genes << "\t" << r.gene_reads << "\t";
            }
            fh << r.tid << "\t" << r.name << r.strain_info << "\n";
        }
    }
    return true;
}

// Iteration order of a Python 2 dict whose keys are small non-negative ints inserted in the given order (CPython 2.7:
// open addressing, slot = hash & mask, probe i = 5 i + 1 + perturb, table of 8 growing fourfold at two thirds full).
// summary.py prints a node's k sizes in that order (8,17,10,12,14 for -k 8,10,12,14,17).
inline std::vector<long> py2_dict_order(const std::vector<long>& keys) {
    size_t size = 8, fill = 0;
    std::vector<long> tab(size, -1);
    auto insert = [&](std::vector<long>& t, size_t sz, long k) -> bool {
        const size_t mask = sz - 1;
        size_t i = (size_t)k & mask, perturb = (size_t)k;
        while (t[i] != -1 && t[i] != k) { i = (5 * i + 1 + perturb) & mask; perturb >>= 5; }
        const bool fresh = t[i] == -1;
        t[i] = k;
        return fresh;
    };
    for (long k : keys) {
        if (insert(tab, size, k)) ++fill;
        if (fill * 3 >= size * 2) {
            const size_t ns = size * (fill > 50000 ? 2 : 4);
            std::vector<long> nt(ns, -1);
            for (long v : tab) if (v != -1) insert(nt, ns, v);
            tab.swap(nt);
            size = ns;
        }
    }
    std::vector<long> out;
    for (long v : tab) if (v != -1) out.push_back(v);
    return out;
}

// bin/summary.py <summ file> <rank table> <fastsummary> <plasmid list> <out base> <ranks> (bin/run_cs.sh:150): per rank of
// interest the calls of that rank with everything below them summed (plasmids apart), a species shown as its best strain,
// shares of the reads, and per k the peak of the k-mer multiplicity histogram content_summ wrote.
inline bool write_ordered_reports(const std::string& summ_fn, const std::string& rank_fn, const std::string& fastsummary_fn,
                                  const std::string& plasmid_fn, const std::string& out_base, const std::string& rank_calls, std::string* err) {
    std::unordered_set<long> plasmids;
    { std::ifstream a(plasmid_fn.c_str()); std::string l; while (std::getline(a, l)) if (!split_ws(l).empty()) plasmids.insert(atol(l.c_str())); }
    auto is_plasmid = [&](long id) { return id >= 10000000 || plasmids.count(id) != 0; };
    std::unordered_map<long, std::string> rank_map, fsum;
    {
        std::ifstream a(rank_fn.c_str());
        if (!a) { if (err) *err = "cannot read " + rank_fn; return false; }
        std::string l;
        while (std::getline(a, l)) { const auto v = split_ws(l); if (v.size() >= 2) rank_map.insert({atol(v[0].c_str()), v[1]}); }
    }
    {
        std::ifstream a(fastsummary_fn.c_str());
        if (!a) { if (err) *err = "cannot read " + fastsummary_fn; return false; }
        std::string l;
        while (std::getline(a, l)) {
            while (!l.empty() && isspace((unsigned char)l.back())) l.pop_back();
            const auto v = split_ws(l);
            if (v.size() >= 3) fsum.insert({atol(v[2].c_str()), l});
        }
    }
    std::unordered_map<long, std::string> names;
    std::unordered_map<long, long> rdcnt;
    std::unordered_map<long, double> wrdcnt;
    std::unordered_map<long, std::vector<long>> child;
    {   // loadTree: the indentation of the .summ report gives the parent
        std::ifstream fh(summ_fn.c_str());
        if (!fh) { if (err) *err = "cannot read " + summ_fn; return false; }
        std::vector<std::pair<long, int>> lines{{1, 0}};  // front = most recent
        std::string l;
        while (std::getline(fh, l)) {
            while (!l.empty() && isspace((unsigned char)l.back())) l.pop_back();
            const auto v = split_tabs(l);
            if (v.empty() || v[0] == "Name") continue;
            size_t it = 0;
            while (it < v.size() && v[it].empty()) ++it;
            if (it + 3 >= v.size()) continue;
            const int tabs = (int)it;
            const long cnode = atol(v[it + 1].c_str());
            names[cnode] = v[it];
            rdcnt[cnode] = atol(v[it + 2].c_str());
            wrdcnt[cnode] = atof(v[it + 3].c_str());
            while (!lines.empty()) {
                if (tabs > lines.front().second) { child[lines.front().first].push_back(cnode); break; }
                lines.erase(lines.begin());
            }
            lines.insert(lines.begin(), {cnode, tabs});
        }
    }
    auto rank_is = [&](long n, const std::string& r) { auto i = rank_map.find(n); return i != rank_map.end() && i->second == r; };
    struct CallRow { long rep, call; double wrc; long rc; };
    std::stringstream rs(rank_calls);
    std::string ranktype;
    while (std::getline(rs, ranktype, ',')) {
        std::ofstream out((out_base + "." + ranktype).c_str());
        if (!out) { if (err) *err = "cannot write " + out_base + "." + ranktype; return false; }
        // loadKmerStats: tid -> k -> (peak multiplicity or -1, distinct k-mers, k-mer total); the first block of a (tid, k) wins
        std::unordered_map<long, std::map<long, std::vector<long>>> kcov;
        std::unordered_map<long, std::vector<long>> korder;
        {
            std::ifstream fh((summ_fn + "." + ranktype + "_kmer_cov").c_str());
            std::vector<std::pair<long, long>> distr;
            long tid = -1, kval = -1, kcnt = -1, tot = -1;
            bool save = false;
            auto flush = [&]() {
                if (distr.empty()) return;
                long peak = -1;
                bool fnd = false;
                for (size_t i = 1; i + 1 < distr.size(); ++i) {
                    if (!fnd && distr[i - 1].second >= distr[i].second && distr[i].second < distr[i + 1].second) fnd = true;
                    if (fnd && distr[i - 1].second <= distr[i].second && distr[i].second > distr[i + 1].second) { peak = distr[i].first; break; }
                }
                if (!kcov[tid].count(kval)) { kcov[tid][kval] = {peak, kcnt, tot}; korder[tid].push_back(kval); }
            };
            std::string ln;
            while (fh && std::getline(fh, ln)) {
                while (!ln.empty() && isspace((unsigned char)ln.back())) ln.pop_back();
                if (ln.empty()) break;
                if (ln.find("taxid=") != std::string::npos && ln.find("distinct_kmer_cnt=") != std::string::npos) {
                    flush();
                    save = false;
                    distr.clear();
                    std::vector<std::string> vals;
                    { std::stringstream es(ln); std::string p; while (std::getline(es, p, '=')) vals.push_back(p); }
                    if (vals.size() < 5) continue;
                    tid = atol(vals[1].c_str());
                    if (rank_is(tid, ranktype)) { kcnt = atol(vals[2].c_str()); kval = atol(vals[3].c_str()); tot = atol(vals[4].c_str()); save = true; }
                } else if (save) {
                    std::vector<std::string> vals;
                    { std::stringstream es(ln); std::string p; while (std::getline(es, p, ' ')) vals.push_back(p); }
                    if (vals.size() >= 4) distr.push_back({atol(vals[2].c_str()), atol(vals[3].c_str())});
                }
            }
            flush();
        }
        // bread_first_traverse + summNode
        std::vector<CallRow> calls;
        std::vector<long> lopen{1};
        while (!lopen.empty()) {
            const long cnode = lopen.front();
            lopen.erase(lopen.begin());
            if ((ranktype == "plasmid" && is_plasmid(cnode)) || (rank_is(cnode, ranktype) && !is_plasmid(cnode))) {
                double tw = wrdcnt[cnode];
                long tr = rdcnt[cnode], the_call = cnode;
                std::vector<long> strains, q = child.count(cnode) ? child[cnode] : std::vector<long>();
                while (!q.empty()) {
                    const long alt = q.front();
                    q.erase(q.begin());
                    if ((ranktype == "species" && !is_plasmid(alt)) || (ranktype != "species" && rdcnt[alt] > 0)) { tw += wrdcnt[alt]; tr += rdcnt[alt]; }
                    if (ranktype == "species" && rank_is(alt, "strain") && !is_plasmid(alt) && rdcnt[alt] > 0) strains.push_back(alt);
                    if (child.count(alt)) for (long nd : child[alt]) q.push_back(nd);
                }
                if (!strains.empty()) {
                    std::stable_sort(strains.begin(), strains.end(), [&](long a, long b) { return wrdcnt[a] > wrdcnt[b]; });
                    the_call = strains[0];
                }
                if (tr > 0) calls.push_back({cnode, the_call, tw, tr});
            } else if (child.count(cnode)) {
                for (long nd : child[cnode]) lopen.insert(lopen.begin(), nd);
            }
        }
        std::stable_sort(calls.begin(), calls.end(), [](const CallRow& a, const CallRow& b) { return a.wrc > b.wrc; });
        out << "% of Reads, Avg Read Score, Weighted Read Count (WRC), Read Count (RC), Original WRC, Original RC, Name, Taxid\n";
        long rc_sum = 0;
        for (auto& c : calls) rc_sum += c.rc;
        for (auto& c : calls) {
            std::string owrc = "-1", orc = "-1", name;
            auto f = fsum.find(c.call);
            if (f != fsum.end()) { const auto v1 = split_tabs(f->second); name = v1.size() > 3 ? v1[3] : ""; owrc = v1[0]; orc = v1.size() > 1 ? v1[1] : ""; }
            else name = names[c.call];
            out << py2_str_float((double)c.rc / (double)rc_sum) << "\t" << py2_str_float(c.wrc / (double)c.rc) << "\t" << py2_str_float(c.wrc) << "\t"
                << c.rc << "\t" << owrc << "\t" << orc << "\t" << name << "\t" << c.call << "\t" << c.rep;
            if (kcov.count(c.rep))
                for (long kv : py2_dict_order(korder[c.rep])) {
                    const auto& t = kcov[c.rep][kv];
                    out << "\t" << kv << "," << t[0] << "," << t[1] << "," << t[2];
                }
            out << "\n";
        }
    }
    return true;
}

}  // namespace lmat
