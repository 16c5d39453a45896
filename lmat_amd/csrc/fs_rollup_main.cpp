// fs_rollup -- the roll-ups of a .fastsummary that bin/run_rl.sh makes with tolineage.py and fsreport.py (rollups.hpp), as one
// tool:  fs_rollup -s <fastsummary> -u <names file> -c <tree> -w <rank table> [-r <low-number plasmids>] [-P <plasmid names>]
//                  [-a plasmid,species,genus] [-n <min reads for .lineage: 10>] [-m <min average score: 0>] [-o <out dir>]
//                  [-g <gene_label summary> -q <min gene reads>] [-S <content_summ report> [-O <out base: <fastsummary>.ordered>]]
// writes <fastsummary>.lineage and <odir>/<basename>.<rank> for every rank that has calls; with -S also what bin/summary.py
// makes of content_summ's report (bin/run_cs.sh:150): <out base>.<rank> for each rank of -a.
#include <getopt.h>
#include <iostream>
#include "rollups.hpp"

int main(int argc, char* argv[]) {
    std::string fs, names, ranks = "plasmid,species,genus", odir, summ, obase;
    lmat::RollupInputs in;
    int num = 10, c;
    double min_avg = 0;
    while ((c = getopt(argc, argv, "s:u:c:w:r:P:a:n:m:o:g:q:S:O:")) != -1) {
        switch (c) {
            case 's': fs = optarg; break;
            case 'u': names = optarg; break;
            case 'c': in.tree_fn = optarg; break;
            case 'w': in.rank_fn = optarg; break;
            case 'r': in.plasmid_fn = optarg; break;
            case 'P': in.plasmid_names_fn = optarg; break;
            case 'a': ranks = optarg; break;
            case 'n': num = atoi(optarg); break;
            case 'm': min_avg = atof(optarg); break;
            case 'o': odir = optarg; break;
            case 'g': in.gene_summary_fn = optarg; break;
            case 'q': in.min_gene_cnt = atoi(optarg); break;
            case 'S': summ = optarg; break;
            case 'O': obase = optarg; break;
            default: std::cerr << "unknown option" << std::endl; return 1;
        }
    }
    if (fs.empty() || in.tree_fn.empty() || in.rank_fn.empty()) { std::cerr << "fs_rollup: -s, -c and -w are required" << std::endl; return 1; }
    if (odir.empty()) { const size_t s = fs.rfind('/'); odir = s == std::string::npos ? "." : fs.substr(0, s); }
    std::string err;
    if (!names.empty() && !lmat::write_lineage(names, fs, fs + ".lineage", num, min_avg, &err)) { std::cerr << "fs_rollup: " << err << std::endl; return 1; }
    if (!lmat::write_rank_reports(fs, ranks, odir, in, &err)) { std::cerr << "fs_rollup: " << err << std::endl; return 1; }
    if (!summ.empty() && !lmat::write_ordered_reports(summ, in.rank_fn, fs, in.plasmid_fn, obase.empty() ? fs + ".ordered" : obase, ranks, &err)) {
        std::cerr << "fs_rollup: " << err << std::endl;
        return 1;
    }
    return 0;
}
