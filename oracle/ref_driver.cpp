// ref_driver.cpp -- drives the parts of the REFERENCE that build from its own
// sources (SortedDb, TaxNodeStat, TaxTree, kencode) to produce golden vectors
// that pin oracle/lmat_oracle.hpp.  TEST INFRASTRUCTURE ONLY; compiled by
// oracle/Makefile against /root/reference in place, output in oracle/_ref/.
// Nothing here is reference code: it only calls the reference's public
// interfaces (SortedDb.hpp:160,185,188,366,427; TaxNodeStat.hpp:60,208,258,262;
// TaxTree.hpp:24,60).  kencode.hpp defines non-inline functions, so its
// driver is a separate translation unit: ref_kencode.cpp.
#include <algorithm>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include "TaxNodeStat.hpp"
#include "TaxTree.hpp"
#include "tid_checks.hpp"

using namespace metag;

static int do_lookup(int argc, char** argv) {
    // lookup <taxhisto.bin> <map32to16.txt> <kmers.txt> [n_kmers_hint [tid_cutoff rank_map|- human_kmers|- adaptor_kmers|-]]
    if (argc < 5) return 2;
    bitreduce_map_t br_map;
    id_convback_map_t conv_map;
    {
        FILE* tfp = fopen(argv[3], "r");
        if (!tfp) return 3;
        uint32_t src;
        uint16_t dest;
        while (fscanf(tfp, "%d%hd", &src, &dest) > 0) {  // same parse as make_db_table.cpp:263 / read_label.cpp:1596
            br_map[src] = dest;
            conv_map[dest] = src;
        }
        fclose(tfp);
    }
    size_t n_kmers = argc > 5 ? strtoull(argv[5], 0, 10) : 4000000;
    size_t space = n_kmers * 64 + (1 << 20);
    SortedDb<uint16_t>* db = new SortedDb<uint16_t>(n_kmers, space);
    db->set_kmer_length(20);
    my_map species_map;
    int tid_cut = argc > 6 ? atoi(argv[6]) : 0;
    if (argc > 7 && std::string(argv[7]) != "-" && tid_cut > 0) {  // make_db_table.cpp:303-313
        FILE* smfp = fopen(argv[7], "r");
        uint32_t src, dest;
        while (fscanf(smfp, "%d%d", &src, &dest) > 0) species_map[src] = dest;
        fclose(smfp);
    }
    FILE* human_fp = (argc > 8 && std::string(argv[8]) != "-") ? fopen(argv[8], "r") : NULL;
    FILE* illu_fp = (argc > 9 && std::string(argv[9]) != "-") ? fopen(argv[9], "r") : NULL;
    db->add_data(argv[2], 0, true, &br_map, species_map, tid_cut, false, human_fp, illu_fp, 32630);
    std::ifstream kin(argv[4]);
    uint64_t kmer;
    my_map tid_rank_map;
    // run-time pruning as read_label passes it (read_label.cpp:1023,1543-1557): env REF_RT_CUT / REF_RT_RANKS
    int rt_cut = getenv("REF_RT_CUT") ? atoi(getenv("REF_RT_CUT")) : (uint16_t)~0;
    if (getenv("REF_RT_RANKS")) {
        FILE* rmfp = fopen(getenv("REF_RT_RANKS"), "r");
        uint32_t src, dest;
        while (fscanf(rmfp, "%d%d", &src, &dest) > 0) tid_rank_map[src] = dest;
        fclose(rmfp);
    }
    while (kin >> kmer) {
        TaxNodeStat<uint16_t> h(*db);
        h.begin(kmer, tid_rank_map, rt_cut, false, &conv_map);
        printf("%llu %u", (unsigned long long)kmer, (unsigned)h.taxidCount());
        while (h.next()) printf(" %u", h.taxid());
        printf("\n");
    }
    return 0;
}

static int do_paths(int argc, char** argv) {
    // paths <tax.dat> : for every node (ascending id) its getPathToRoot
    if (argc < 3) return 2;
    TaxTree<uint32_t> tree(argv[2]);
    std::vector<uint32_t> ids;
    for (auto it = tree.begin(); it != tree.end(); ++it) ids.push_back(it->first);
    std::sort(ids.begin(), ids.end());
    for (size_t i = 0; i < ids.size(); ++i) {
        std::vector<uint32_t> p;
        tree.getPathToRoot(ids[i], p);
        printf("%u :", ids[i]);
        for (size_t j = 0; j < p.size(); ++j) printf(" %u", p[j]);
        printf("\n");
    }
    return 0;
}

static int do_tidchecks() {
    // truth table of the hard-coded predicates (include/tid_checks.hpp:10-28)
    uint32_t ids[] = {1, 9606, 63221, 741158, 374840, 10847, 32630, 12721, 693660, 20999999, 2759, 10239, 2157, 2};
    for (uint32_t t : ids) printf("%u human=%d phix=%d\n", t, (int)isHuman(t), (int)(isPhiX(t)));
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    std::string m = argv[1];
    if (m == "lookup") return do_lookup(argc, argv);
    if (m == "paths") return do_paths(argc, argv);
    if (m == "tidchecks") return do_tidchecks();
    return 2;
}
