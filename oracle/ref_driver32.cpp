// ref_driver32.cpp -- drives the REFERENCE's SortedDb<uint32_t> + TaxNodeStat<uint32_t> (the TID_SIZE=32 build that
// gene_label uses, src/gene_label.cpp:15-22; CMakeLists.txt:92-105) to produce the golden lookups that pin the gene-database
// path.  TEST INFRASTRUCTURE ONLY; compiled by oracle/Makefile against /root/reference in place with -DTID_SIZE=32
// -DDBTID_T=uint32_t, output in oracle/_ref/.  Nothing here is reference code: it only calls the reference's public
// interfaces (SortedDb.hpp:160,185,427; TaxNodeStat.hpp:41,208,258,262).
#include <cstdio>
#include <fstream>
#include <iostream>
#include "TaxNodeStat.hpp"

using namespace metag;

int main(int argc, char** argv) {
    // <taxhisto-format gene db> <kmers.txt> [n_kmers_hint]
    if (argc < 3) return 2;
    size_t n_kmers = argc > 3 ? strtoull(argv[3], 0, 10) : 1000000;
    SortedDb<uint32_t>* db = new SortedDb<uint32_t>(n_kmers, n_kmers * 256 + (1 << 20));
    db->set_kmer_length(20);
    my_map species_map;
    db->add_data(argv[1], 0, true, NULL, species_map, 0, false, NULL, NULL, 32630);  // no 32->16 map: ids stored as read
    std::ifstream kin(argv[2]);
    uint64_t kmer;
    while (kin >> kmer) {
        TaxNodeStat<uint32_t> h(*db);
        h.begin(kmer, NULL);  // as gene_label.cpp:252
        printf("%llu %u", (unsigned long long)kmer, (unsigned)h.taxidCount());
        while (h.next()) printf(" %u", h.taxid());
        printf("\n");
    }
    return 0;
}
