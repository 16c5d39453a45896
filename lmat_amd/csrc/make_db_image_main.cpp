// make_db_image -- the engine's counterpart of make_db_table (src/make_db_table.cpp): tax_histo binaries in,
// database image out.  Same option letters where they exist upstream (getopt string
// "g:q:k:i:o:s: l h m:f:wj:c:u:V", make_db_table.cpp:150): -i input (-l: a list file), -o output image,
// -k k-mer length, -f 32->16 map, -g taxid cutoff with -m rank map, -j human k-mers, -u adaptor k-mers.
// A database of 32-bit taxids (the reference's TID_SIZE=32 build has no -f): -t <taxonomy tree> numbers the tree's nodes when it
// has at most 65534 of them; with -M <map file> as well the 32->16 map is made from the database's own taxids and their
// ancestors and written to that file -- pass it to read_label as -f (any size of tree; the closure must fit 16 bits).
// Needs no GPU.  -w (strain->species pruning) is "functionality disabled" upstream too (SortedDb.cpp:317-319).
#include <getopt.h>
#include <chrono>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>
#include "dbbuild.hpp"

int main(int argc, char* argv[]) {
    std::string inputfn, outputfn, species_map_fn, id_bit_conv_fn, human_kmer_fn, illu_kmer_fn, tree_fn, map_out_fn;
    bool list = false;
    int kmer_len = 0, tid_cut = 0, count = 0, c;
    std::cout << "invocation: ";
    for (int j = 0; j < argc; j++) std::cout << argv[j] << " ";
    std::cout << std::endl;
    while ((c = getopt(argc, argv, "g:q:k:i:o:s:lhm:f:wj:c:u:Vt:M:")) != -1) {
        switch (c) {
            case 'j': human_kmer_fn = optarg; break;
            case 'u': illu_kmer_fn = optarg; break;
            case 'w': std::cout << "functionality disabled!\n"; return 1;
            case 'f': id_bit_conv_fn = optarg; break;
            case 'M': map_out_fn = optarg; break;
            case 't': tree_fn = optarg; break;  // taxonomy tree: gives the id codes of a 32-bit-taxid database (no -f)
            case 'k': ++count; kmer_len = atoi(optarg); break;
            case 'l': list = true; break;
            case 'i': ++count; inputfn = optarg; break;
            case 'o': ++count; outputfn = optarg; break;
            case 'g': tid_cut = atoi(optarg); break;
            case 'm': species_map_fn = optarg; break;
            case 'q': case 's': case 'c': break;  // stopper / mmap size / extra k-mers: not needed for an image
            case 'h': std::cerr << "only the tax_histo input format is supported\n"; return 1;
            case 'V': std::cout << "LMAT version 1.2.4_2018a\n"; return 0;
            default: std::cerr << "usage: make_db_image -i <tax_histo|list> [-l] -o <image> -k <k> (-f <32to16 map> | -t <taxonomy tree> [-M <map out>]) [-g N -m rankmap] [-j human] [-u adaptors]\n"; return 1;
        }
    }
    if (count != 3 || (id_bit_conv_fn.empty() && tree_fn.empty())) {
        std::cerr << "usage: make_db_image -i <tax_histo|list> [-l] -o <image> -k <k> (-f <32to16 map> | -t <taxonomy tree> [-M <map out>]) [-g N -m rankmap] [-j human] [-u adaptors]\n";
        return 1;
    }
    std::vector<std::string> files;
    if (list) { std::ifstream ifs(inputfn.c_str()); std::string l; while (ifs >> l) files.push_back(l); }
    else files.push_back(inputfn);
    lmat::Ingest ing;
    ing.k = kmer_len;
    if (!map_out_fn.empty()) {
        if (tree_fn.empty() || !id_bit_conv_fn.empty()) { std::cerr << "-M <map file> goes with -t <taxonomy tree> and without -f\n"; return 1; }
        std::vector<std::pair<uint32_t, uint16_t>> m;
        std::string e;
        if (!lmat::idmap_from_database(files, tree_fn.c_str(), 32630, m, e) || !lmat::save_idmap(m, map_out_fn.c_str())) {
            std::cerr << (e.empty() ? "cannot write " + map_out_fn : e) << std::endl;
            return 1;
        }
        std::cout << "32->16 map of " << m.size() << " ids written to " << map_out_fn << std::endl;
        id_bit_conv_fn = map_out_fn;
    }
    if (!(id_bit_conv_fn.empty() ? ing.idmap_from_tree(tree_fn.c_str()) : ing.load_idmap(id_bit_conv_fn.c_str())) ||
        !ing.set_options(tid_cut, species_map_fn.c_str(), human_kmer_fn.c_str(), illu_kmer_fn.c_str(), 32630)) {
        std::cerr << ing.err << std::endl;
        return 1;
    }
    auto t0 = std::chrono::steady_clock::now();
    for (auto& f : files) {
        if (!ing.add_taxhisto(f.c_str())) { std::cerr << ing.err << std::endl; return 1; }
        std::cout << "kmer count: " << ing.kmers.size() << "\nsingletons: " << ing.singletons << "\ndoubles: " << ing.doubles
                  << "\nkmers reduced: " << ing.reduced_kmers << "\nkmers cut to 1: " << ing.cut_kmers << "\nnew human k-mers: "
                  << ing.new_human << "\nmatched human k-mers: " << ing.matched_in << "\nnew human + other k-mers: " << ing.new_isect << "\n";
    }
    if (!ing.save_image(outputfn.c_str())) { std::cerr << "cannot write " << outputfn << std::endl; return 1; }
    std::cout << "KmerDB load time: " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << std::endl;
    return 0;
}
