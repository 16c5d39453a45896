// lmat_oracle_main.cpp -- command-line front end of the CPU oracle.
// TEST INFRASTRUCTURE ONLY (see lmat_oracle.hpp header).  Mirrors the flag
// letters of the reference's read_label (src/read_label.cpp:1351-1442) so that
// tests read like reference invocations; -d takes a tax_histo binary (or a
// text file listing several), because PERM heap images cannot be opened here.
#include <getopt.h>
#include <unistd.h>
#include <chrono>
#include "lmat_oracle.hpp"

using namespace orc;

static bool is_list_file(const std::string& fn) {
    // a tax_histo binary starts with u32 data_start(29) u64 count u64 ~0
    FILE* f = fopen(fn.c_str(), "rb");
    if (!f) return false;
    unsigned char b[20];
    size_t n = fread(b, 1, 20, f);
    fclose(f);
    if (n < 20) return true;
    for (int i = 12; i < 20; ++i)
        if (b[i] != 0xff) return true;
    return false;
}

int main(int argc, char* argv[]) {
    Options opt;
    std::string rank_ids, kmer_db_fn, query_fn, ofbase, tax_tree_fn, depth_file, rank_map_file, id_map_fn, plasmid_fn,
        trace_fn, null_fn;
    int n_threads = 0, k_size = -1;
    int c;
    while ((c = getopt(argc, argv, "u:ahn:j:b:ye:w:pk:c:v:i:d:l:t:r:sm:o:x:f:g:z:qVHT:")) != -1) {
        switch (c) {
            case 'h': opt.screen_phix = false; break;
            case 'r': plasmid_fn = optarg; break;
            case 'f': id_map_fn = optarg; break;
            case 'j': opt.min_kmer = atoi(optarg); break;
            case 'z': opt.min_fnd_kmer = atoi(optarg); break;
            case 'u': rank_ids = optarg; break;
            case 'x': opt.min_score = atof(optarg); break;
            case 'a': opt.prn_read = false; break;
            case 'w': rank_map_file = optarg; break;
            case 'b': opt.diff_thresh = atof(optarg); break;
            case 'l': opt.diff_thresh2 = atof(optarg); break;
            case 'e': depth_file = optarg; break;
            case 'q': opt.fastq = true; break;
            case 'p': opt.prn_all = true; break;
            case 't': n_threads = atoi(optarg); break;
            case 'c': tax_tree_fn = optarg; break;
            case 'k': k_size = atoi(optarg); break;
            case 'i': query_fn = optarg; break;
            case 'd': kmer_db_fn = optarg; break;
            case 'o': ofbase = optarg; break;
            case 'T': trace_fn = optarg; break;  // oracle-only: per-read intermediate dump
            case 'n': null_fn = optarg; break;
            case 's': case 'm': case 'g':
                std::cerr << "oracle: option -" << (char)c << " (permissive / run-time pruning) is not restated\n";
                return -2;
            default: break;
        }
    }
    if (depth_file == "" || ofbase == "" || n_threads == 0 || kmer_db_fn == "" || query_fn == "") {
        std::cerr << "Params: " << ofbase << " " << n_threads << " " << kmer_db_fn << " " << query_fn << " " << depth_file << std::endl;
        return -1;
    }
    Taxonomy tax;
    if (!id_map_fn.empty() && !tax.load_idmap(id_map_fn)) { std::cerr << "cannot read " << id_map_fn << "\n"; return -1; }
    if (!rank_map_file.empty()) tax.load_rank(rank_map_file);
    if (!plasmid_fn.empty()) tax.load_plasmids(plasmid_fn);
    if (!tax.load_tree(tax_tree_fn)) return -1;
    if (id_map_fn.empty() && !tax.idmap_from_tree()) { std::cerr << "tree too large for a run without -f\n"; return -1; }
    if (!tax.load_depth(depth_file)) { std::cerr << "ERROR! Unable to open: " << depth_file << std::endl; return -1; }

    KmerDb db;
    std::vector<std::string> files;
    if (is_list_file(kmer_db_fn)) {
        std::ifstream l(kmer_db_fn.c_str());
        std::string f;
        while (l >> f) files.push_back(f);
    } else {
        files.push_back(kmer_db_fn);
    }
    for (auto& f : files) {
        std::string err;
        if (!db.add_taxhisto(f, tax, &err)) { std::cerr << "Error: " << err << "\n"; return -1; }
    }
    if (k_size < 1) k_size = db.k;

    std::ifstream qf;
    std::istream* in = &std::cin;
    if (query_fn != "-") {
        qf.open(query_fn.c_str());
        if (!qf) { std::cerr << "ERROR! Did not open for reading: " << query_fn << std::endl; return -1; }
        in = &qf;
    }
    NullModel nm;
    if (!null_fn.empty() && !nm.load(null_fn)) return -1;
    Classifier cls(tax, db, opt, nm.loaded ? &nm : nullptr);
    auto t0 = std::chrono::steady_clock::now();
    RunOutputs ro;
    if (trace_fn.empty()) {
        ro = run_reads(cls, *in, k_size, rank_ids);
    } else {
        // same as run_reads but also dumps intermediates per read
        std::ofstream tf(trace_fn.c_str());
        Tallies tl;
        std::ostringstream ofs;
        ReadStream rs(*in, opt.fastq);
        std::string read, hdr;
        size_t n = 0;
        while (rs.next(read, hdr)) {
            ++n;
            if (hdr.empty()) { std::ostringstream o; o << "unknown_hdr:" << n; hdr = o.str(); }
            ofs << hdr << "\t";
            if (opt.prn_read) ofs << read << "\t"; else ofs << "X\t";
            ReadTrace tr;
            cls.proc_line((int)read.length(), read, k_size, ofs, tl, &tr);
            tf << "R " << n - 1 << " valid=" << tr.valid_kmers << " bin=" << tr.bin_sel << " cand=" << tr.cand_kmer_cnt
               << " nuniq=" << tr.uniq_kmers.size() << "\n";
            tf << "K";
            for (size_t i = 0; i < tr.uniq_kmers.size(); ++i) tf << " " << tr.uniq_pos[i] << ":" << tr.uniq_kmers[i];
            tf << "\nT";
            for (size_t i = 0; i < tr.reg_order.size(); ++i) {
                tf << " " << tr.reg_order[i];
                if (i < tr.reg_count.size()) tf << ":" << tr.reg_count[i];
            }
            tf << "\n";
        }
        ro.n_reads = n;
        ro.out = ofs.str();
        write_summaries(tl, rank_ids, ro);
    }
    double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    {
        std::ofstream o((ofbase + "0.out").c_str());
        o << ro.out;
    }
    std::ostringstream b;
    b << ofbase << "." << opt.min_score << "." << opt.min_kmer;
    {
        std::ofstream o((b.str() + ".fastsummary").c_str());
        o << ro.fastsummary;
    }
    {
        std::ofstream o((b.str() + ".nomatchsum").c_str());
        o << ro.nomatchsum;
    }
    std::cout << "Total reads loaded: " << ro.n_reads << std::endl;
    std::cout << "DONE! Total query time: " << el << " sec = " << el / 60 << " min" << std::endl;
    return 0;
}
