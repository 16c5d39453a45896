// ref_rkmer.cpp -- drives the REFERENCE's own retrieve_kmer_labels (src/rkmer.hpp:74-293, the function
// rand_read_label runs per read: k-mer extraction, per-read dedupe, SortedDb lookup, taxid filtering, depth sort,
// leaf-most filter, representative strain per species, lineage closure, registration order) on a FASTA file and
// prints what it produced, to pin oracle/lmat_oracle.hpp.  TEST INFRASTRUCTURE ONLY; compiled by oracle/Makefile
// against /root/reference in place, output in oracle/_ref/.  Nothing here is reference code: rkmer.hpp is a header of
// function definitions that expects its includer to own a handful of globals (src/rand_read_label.cpp:20-60 does);
// this file is such an includer, and the per-taxid position count below is the six-line loop of
// src/rand_read_label.cpp:383-396.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <list>
#include <map>
#include <set>
#include <sstream>
#include <vector>
#include "TaxNodeStat.hpp"
#include "TaxTree.hpp"

#define TID_T uint32_t
using namespace std;
using namespace metag;

#include "rkmer.hpp"

bool verbose = false;
bool gPERMISSIVE_MATCH = false;
map<TID_T, string> gRank_table;
my_map tid_rank_map;
id_convback_map_t conv_map;
bool tid_map_is_strain_species = false;

int main(int argc, char** argv) {
    // ref_rkmer <taxhisto.bin> <map32to16.txt> <tax.dat> <depth.dat> <rank.txt> <reads.fa> <k> [permissive [max_count [numeric_ranks]]]
    if (argc < 8) return 2;
    bitreduce_map_t br_map;
    {
        FILE* tfp = fopen(argv[2], "r");
        if (!tfp) return 3;
        uint32_t src;
        uint16_t dest;
        while (fscanf(tfp, "%d%hd", &src, &dest) > 0) { br_map[src] = dest; conv_map[dest] = src; }
        fclose(tfp);
    }
    const int k = atoi(argv[7]);
    gPERMISSIVE_MATCH = argc > 8 && atoi(argv[8]) != 0;
    const uint16_t max_count = argc > 9 ? (uint16_t)atoi(argv[9]) : (uint16_t)~0;  // -h of rand_read_label (:457)
    if (argc > 10) {  // -r rank table (:640-650)
        FILE* rmfp = fopen(argv[10], "r");
        if (!rmfp) return 4;
        uint32_t src, dest;
        while (fscanf(rmfp, "%d%d", &src, &dest) > 0) tid_rank_map[src] = dest;
        fclose(rmfp);
    }
    SortedDb<uint16_t>* db = new SortedDb<uint16_t>(4000000, (size_t)4000000 * 64 + (1 << 20));
    db->set_kmer_length(k);
    my_map species_map;
    db->add_data(argv[1], 0, true, &br_map, species_map, 0, false, NULL, NULL, 32630);
    TaxTree<TID_T> tax_tree(argv[3]);
    hmap_t dmap;
    {
        ifstream ifs(argv[4]);
        TID_T t, d;
        while (ifs >> t >> d) dmap[t] = d;  // src/rand_read_label.cpp:663-665
    }
    {
        ifstream ifs(argv[5]);
        TID_T t;
        string r;
        while (ifs >> t >> r) gRank_table.insert(make_pair(t, r));  // :519-525
    }
    ifstream fa(argv[6]);
    string line, read;
    unsigned idx = 0;
    auto run = [&](const string& rd) {
        // one line per read, written after the call: with pruning on the reference prints blank lines of its own
        // (TaxNodeStat.hpp:192) while it runs
        ostringstream o;
        const int ri_len = (int)rd.length();
        o << "R " << idx++ << " len=" << ri_len;
        if (ri_len < k) { cout << o.str() << " short" << endl; return; }
        vector<label_info_t> label_vec(ri_len - k + 1, make_pair(-1, tax_data_t()));
        list<TID_T> taxid_lst;
        hmap_t tax2idx, idx2tax;
        const pair<int, int> res = retrieve_kmer_labels(db, rd.c_str(), ri_len, k, label_vec, taxid_lst, tax2idx, idx2tax, dmap, tax_tree, max_count);
        map<TID_T, int> cnt_tids;  // src/rand_read_label.cpp:383-396
        int nonneg = 0;
        for (unsigned pos = 0; pos < label_vec.size(); ++pos) {
            if (label_vec[pos].first >= 0) ++nonneg;
            for (tax_data_t::const_iterator it = label_vec[pos].second.begin(); it != label_vec[pos].second.end(); ++it) cnt_tids[it->first] += 1;
        }
        o << " valid=" << res.first << " bin=" << res.second << " marked=" << nonneg << " reg=";
        for (list<TID_T>::const_iterator it = taxid_lst.begin(); it != taxid_lst.end(); ++it) o << (it == taxid_lst.begin() ? "" : ",") << *it << ":" << cnt_tids[*it];
        cout << o.str() << endl;
    };
    // LMAT_REF_TIME=<passes>: timing mode (scripts/ref_ratio.py) -- the reads are held in memory, the per-read loop runs that
    // many times with its text going nowhere, and the rate goes to stderr; database and taxonomy loading stay outside the window
    if (const char* tm = getenv("LMAT_REF_TIME")) {
        vector<string> reads;
        while (getline(fa, line)) {
            if (!line.empty() && line[0] == '>') { if (!read.empty()) reads.push_back(read); read.clear(); }
            else read += line;
        }
        if (!read.empty()) reads.push_back(read);
        const int passes = max(1, atoi(tm));
        ostringstream sink;
        streambuf* keep = cout.rdbuf(sink.rdbuf());
        const auto t0 = chrono::steady_clock::now();
        for (int p = 0; p < passes; ++p) { sink.str(""); for (const string& r : reads) run(r); }
        const double dt = chrono::duration<double>(chrono::steady_clock::now() - t0).count();
        cout.rdbuf(keep);
        cerr << "reads " << reads.size() << " passes " << passes << " seconds " << dt << " reads_per_s " << reads.size() * (double)passes / dt << endl;
        return 0;
    }
    while (getline(fa, line)) {
        if (!line.empty() && line[0] == '>') { if (!read.empty()) run(read); read.clear(); }
        else read += line;
    }
    if (!read.empty()) run(read);
    return 0;
}
