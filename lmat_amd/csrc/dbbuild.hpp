// dbbuild.hpp -- host-side ingest of make_db_table's input into the engine's canonical (k-mer, payload)
// form.  No GPU involved: this is what the `make_db_image` tool and lmat_db_add_taxhisto share.
//
// Restates SortedDb::add_data (src/kmerdb/SortedDb.cpp:84-751) for the tax_histo format, including the
// options of make_db_table that change database contents (src/make_db_table.cpp:150-213,259-313):
//   -f 32->16 map          every stored id goes through it (SortedDb.cpp:503-511,678-690)
//   -j human k-mer feed    sorted ASCII k-mers merged in: new singleton 9606 k-mers, and 9606 appended to
//                          the lists of matching k-mers (SortedDb.cpp:170-233,475-530,664-715)
//   -u adaptor k-mer feed  k-mers of this set become singleton adaptor-taxid entries (SortedDb.cpp:190-204,275-292)
//   -g N -m rank map       build-time pruning of lists longer than N by rank priority (SortedDb.cpp:296-409)
#pragma once
#include <stdint.h>
#include <cstdio>
#include <functional>
#include <map>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace lmat {

// canonical payload: 1..65535 = the 16-bit DB taxid of a one-element list; 65536 + i = lists[i]
bool tree_node_ids(const char* tree_fn, std::vector<uint32_t>& ids, std::string& err, bool any_size = false);  // sorted, unique, <= 65534

// The 32->16 map of a database made from its own content, for taxonomies of more than 65534 nodes (a TID_SIZE=32 build of the
// reference needs no map; this engine's ids are 16 bits wide): every taxid the tax_histo files hold, 1 / 9606 / the adaptor
// id where the tree has them (what the ingest options add), and all their ancestors in the tree, numbered from 1 in
// ascending order.  Fails when that closure has more than 65534 ids.  save_idmap writes the reference's -f file format.
bool idmap_from_database(const std::vector<std::string>& taxhisto_files, const char* tree_fn, uint32_t adaptor_tid,
                         std::vector<std::pair<uint32_t, uint16_t>>& out, std::string& err);
bool save_idmap(const std::vector<std::pair<uint32_t, uint16_t>>& map, const char* fn);

struct Ingest {
    int k = 0;
    bool raw32 = false;  // gene databases (gene_label): 32-bit ids stored as they come, two u16 (low, high) per id, no options
    std::unordered_map<uint32_t, uint16_t> br;  // 32 -> 16
    // options
    int tid_cutoff = 0;
    std::unordered_map<uint32_t, uint32_t> species_map;
    uint32_t adaptor_tid = 32630;
    FILE* human_fp = nullptr;
    bool adaptor_loaded = false;
    std::unordered_set<uint64_t> adaptor_set;
    // output
    std::vector<uint64_t> kmers;
    std::vector<uint32_t> payload;
    std::vector<std::vector<uint16_t>> lists;
    std::map<std::vector<uint16_t>, uint32_t> list_index;
    // add_data's function-static state
    uint64_t last_kmer = 0;
    uint64_t last_human = ~0ull;
    bool human_primed = false;
    // counters printed by add_data
    uint64_t singletons = 0, doubles = 0, reduced_kmers = 0, cut_kmers = 0, new_human = 0, matched_in = 0, new_isect = 0;
    std::string err;
    // streaming: when set, called whenever flush_every k-mers have accumulated (and by the owner at the end); it consumes
    // kmers/payload (the callee clears them), lists keep growing.  lookup()/save_image() then see only the unflushed tail.
    std::function<bool(Ingest&)> flush;
    size_t flush_every = 0;
    uint64_t flushed = 0;
    bool stream_failed = false;

    ~Ingest() { if (human_fp) fclose(human_fp); }
    bool load_idmap(const char* fn);
    // Databases built without a 32->16 map (the reference's TID_SIZE=32 builds): the storage code of a taxid is
    // its rank among the taxonomy tree's node ids, derived identically by the ingest and by the classifier.
    bool idmap_from_tree(const char* tree_fn);
    bool set_options(int cutoff, const char* species_map_fn, const char* human_fn, const char* adaptor_fn, uint32_t adaptor);
    bool add_taxhisto(const char* fn);
    bool save_image(const char* fn) const;
    bool load_image(const char* fn);
    // image -> flush callback in chunks of flush_every k-mers (lists are read first); nothing is kept in memory
    bool load_image_streaming(const char* fn, uint64_t* n_kmers);
    // stored list of a k-mer (16-bit ids, stored order); returns false when absent
    bool lookup(uint64_t kmer, std::vector<uint16_t>& out) const;

private:
    uint64_t read_encode(FILE* f);
    bool to16(uint32_t tid, uint16_t& out, const char* what);
    void push(uint64_t kmer, const std::vector<uint16_t>& lst);
};

}  // namespace lmat
