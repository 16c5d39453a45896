// gene_label_main.cpp -- `gene_label`-compatible command line on top of liblmat_hip.so.
//
// Keeps the contract of src/gene_label.cpp: getopt string and flag meanings (:348-449; -d gene database, -l list of
// read_label output files -- one per upstream "thread" --, -g gzipped gene annotation table, -o output prefix, -x min gene
// score, -q min k-mers, -b min taxonomic score, -k k-mer length), the per-file outputs <o><i>.out with one line per read that
// hits a gene (:299-300), and the two summaries <o>.<x>.<q>.genesummary[.min_tax_score.<b>] joined against the annotation
// table (:658-703).  The per-read work (retrieve_kmer_labels + the vote, :218-300) runs on the GPU through the C ABI: a
// context opened with lmat_genedb_begin returns gene_label's vote from lmat_classify.
// -R <list of gene_label output files, one per input file> replays an earlier run instead: every read's vote is taken from
// those files (no database, no GPU) and the outputs and summaries are made again, e.g. under other thresholds.
// -d takes the database in the tax_histo record format with 32-bit gene ids (what make_db_table ingests for a TID_SIZE=32
// build), or a text file listing several; PERM heap images cannot be opened without perm-je.
#include <getopt.h>
#include <zlib.h>
#include <chrono>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>
#include "../../include/lmat_hip.h"
#include "outfmt.hpp"

#define LMAT_VERSION "1.2.4_2018a"
using namespace lmat;

static void usage(const char* exe) {
    std::cout << "Usage:\n";
    std::cout << exe << " -d <gene db file (list)> -l <list of read_label output files> -g <gene annotation table (gz)> -o <output path>\n";
    std::cout << "[-x <min gene score>] [-q <min k-mers>] [-b <min taxonomic score>] [-k <kmer size>]\n";
}

struct Rec { std::string hdr, read; uint32_t taxid; float tax_score; };

int main(int argc, char* argv[]) {
    signed char c;
    int n_threads = 0, k_size = -1, min_kmer = 0;
    float min_score = 0.0f, min_tax_score = 0.0f;
    std::string genefile, kmer_db_fn, query_fn, query_fn_lst, ofbase, replay_lst;
    while ((c = getopt(argc, argv, "b:h:n:jye:wmpk:c:v:k:i:d:l:t:s:r o:x:f:g:z:q:aVR:")) != -1) {
        switch (c) {
            case 'b': min_tax_score = atof(optarg); break;
            case 'g': genefile = optarg; break;
            case 'h': break;  // max_count: parsed and unused by the per-read path upstream too
            case 's': break;  // PERM heap size
            case 'j': case 'y': break;  // verbose dumps are not produced
            case 'l': query_fn_lst = optarg; break;
            case 'p': case 'a': break;
            case 't': n_threads = atoi(optarg); break;
            case 'x': min_score = atof(optarg); break;
            case 'q': min_kmer = atoi(optarg); break;
            case 'k': k_size = atoi(optarg); break;
            case 'i': query_fn = optarg; break;
            case 'd': kmer_db_fn = optarg; break;
            case 'R': replay_lst = optarg; break;
            case 'o': ofbase = optarg; break;
            case 'V': std::cout << "LMAT version " << LMAT_VERSION << "\n"; exit(0);
            default: std::cout << "Unrecognized option: " << c << ", ignore." << std::endl;
        }
    }
    const bool replay = !replay_lst.empty();
    if (ofbase == "" || (kmer_db_fn == "" && !replay)) {
        std::cout << "essential arguments missing: [" << ofbase << "] [" << n_threads << "] [" << kmer_db_fn << "] [" << query_fn << "] " << std::endl;
        usage(argv[0]);
        return -1;
    }
    lmat_ctx* ctx = nullptr;
    auto fail = [&](const char* what) { std::cerr << "ERROR! " << what << ": " << lmat_last_error(ctx) << std::endl; lmat_ctx_destroy(ctx); return -1; };
    // replay: per input file, header -> (gene, votes, valid k-mers) as an earlier run printed them
    struct Vote { uint32_t gid, top, cnt; };
    std::vector<std::map<std::string, Vote>> votes;
    if (replay) {
        std::ifstream l(replay_lst.c_str());
        std::string f, line;
        while (l >> f) {
            votes.emplace_back();
            std::ifstream in(f.c_str());
            if (!in) { std::cerr << "did not open for reading: [" << f << "]" << std::endl; return -1; }
            while (std::getline(in, line)) {
                const size_t p1 = line.find('\t'), p2 = line.find('\t', p1 + 1), p3 = line.find('\t', p2 + 1), p4 = line.find('\t', p3 + 1),
                             p5 = line.find('\t', p4 + 1);
                if (p5 == std::string::npos) continue;
                Vote v = {0, 0, 0};
                int minus1 = 0;
                std::istringstream a(line.substr(p4 + 1, p5 - p4 - 1)), b(line.substr(p5 + 1));
                a >> minus1 >> v.top >> v.cnt;
                b >> v.gid;
                votes.back()[line.substr(0, p1)] = v;
            }
        }
    } else {
    std::cout << "Start kmer DB load\n";
    int device = 0;
    if (const char* d = getenv("LMAT_DEVICE")) device = atoi(d);
    lmat_params prm = {1.0f, 0.0f, 0.0f, 0, 0, 0, 0};
    if (lmat_ctx_create(device, &prm, &ctx) != LMAT_OK) { std::cerr << "ERROR! no usable HIP device (this build has no CPU path)" << std::endl; return -1; }
    std::vector<std::string> files;
    {   // a tax_histo-format file starts with its 29-byte header (64 one-bits at offset 12); anything else is a list of such files
        FILE* f = fopen(kmer_db_fn.c_str(), "rb");
        unsigned char b[20];
        const size_t n = f ? fread(b, 1, 20, f) : 0;
        if (f) fclose(f);
        bool binary = n == 20;
        for (int i = 12; binary && i < 20; ++i) binary = b[i] == 0xff;
        if (binary) files.push_back(kmer_db_fn);
        else { std::ifstream l(kmer_db_fn.c_str()); std::string s; while (l >> s) files.push_back(s); }
    }
    if (files.empty()) { std::cout << "Error opening db file, must exit:" << kmer_db_fn << std::endl; return -1; }
    uint32_t klen = 0;
    uint64_t n_total = 0;
    for (auto& fn : files) {
        FILE* h = fopen(fn.c_str(), "rb");
        uint64_t nk = 0;
        if (h) { fseek(h, 4, SEEK_SET); if (fread(&nk, 8, 1, h) != 1) nk = 0; if (!klen) { fseek(h, 25, SEEK_SET); if (fread(&klen, 4, 1, h) != 1) klen = 0; } fclose(h); }
        n_total += nk;
    }
    if (lmat_genedb_begin(ctx, (int)klen, n_total, 0) != LMAT_OK) return fail("gene DB");
    for (auto& fn : files) if (lmat_db_add_taxhisto(ctx, fn.c_str()) != LMAT_OK) return fail("gene DB");
    if (lmat_db_finalize(ctx) != LMAT_OK) return fail("gene DB");
    if (k_size < 1) k_size = lmat_db_kmer_length(ctx);
    std::cout << "num kmers: " << lmat_db_size(ctx) << " - " << k_size << std::endl;
    }
    if (query_fn.length() > 0) { std::cout << "Sorry fasta input file not yet supported" << std::endl; exit(0); }
    std::vector<std::string> inputs;
    { std::ifstream ifs(query_fn_lst.c_str()); std::string s; while (ifs >> s) inputs.push_back(s); }
    if (n_threads != 0 && n_threads != (int)inputs.size())
        std::cout << "warning, thread count overwritten (for now assume when a list of LMAT taxonomy classification files are given, a thread is created for each file)" << std::endl;
    n_threads = (int)inputs.size();
    std::cout << "set threads=" << n_threads << std::endl;
    const auto t_start = std::chrono::steady_clock::now();

    // gene -> label (taxid) -> count / score sum, merged over the files in their order (doMerge / doMergeF, :135-186)
    std::map<uint32_t, std::map<uint32_t, uint32_t>> merge_cnt, merge_cnt_tax;
    std::map<uint32_t, std::map<uint32_t, float>> score_merge, score_merge_tax;
    const size_t kBatch = 1u << 19;
    for (size_t th = 0; th < inputs.size(); ++th) {
        std::ifstream ifs(inputs[th].c_str());
        if (!ifs) { std::cerr << "did not open for reading: [" << inputs[th] << "] tid: [" << th << "]" << std::endl; exit(-1); }
        std::ostringstream nm;
        nm << ofbase << th << ".out";
        std::ofstream ofs(nm.str().c_str());
        std::map<uint32_t, std::map<uint32_t, uint32_t>> track, track_tax;          // label -> gene -> count, this file
        std::map<uint32_t, std::map<uint32_t, float>> score_track, score_track_tax;
        std::vector<Rec> recs;
        std::vector<uint8_t> bases;
        std::vector<uint64_t> off(1, 0);
        auto flush = [&]() -> bool {
            if (recs.empty()) return true;
            std::vector<lmat_read_result> res(recs.size());
            if (replay) {
                for (size_t i = 0; i < recs.size(); ++i) {
                    res[i].status = LMAT_ST_NODBHITS;
                    if (th >= votes.size()) continue;
                    auto it = votes[th].find(recs[i].hdr);
                    if (it == votes[th].end()) continue;
                    res[i].status = LMAT_ST_CALL;
                    res[i].call_tid = it->second.gid;
                    res[i].n_cand = it->second.top;
                    res[i].cand_kmer_cnt = (uint16_t)it->second.cnt;
                    res[i].call_score = (float)it->second.top / (float)it->second.cnt;
                }
            } else {
                bases.push_back(0);
                lmat_reads* dr = nullptr;
                if (lmat_reads_upload(ctx, bases.data(), off.data(), recs.size(), &dr) != LMAT_OK) return false;
                const int rc = lmat_classify(ctx, dr, 0, recs.size(), res.data(), nullptr, 0, nullptr);
                lmat_reads_free(ctx, dr);
                if (rc != LMAT_OK) return false;
            }
            std::string s;
            for (size_t i = 0; i < recs.size(); ++i) {
                const lmat_read_result& r = res[i];
                if (r.status != LMAT_ST_CALL) continue;  // no gene: nothing is printed (:301-304, 280-284)
                const Rec& q = recs[i];
                const uint32_t gl = r.call_tid, cnt = r.cand_kmer_cnt;
                const float gscore = r.call_score;
                s.append(q.hdr); s += '\t'; s.append(q.read); s += '\t'; put_int(s, q.taxid); s += ' '; put_float(s, q.tax_score); s += '\t';
                s += "\t-1 "; put_int(s, r.n_cand); s += ' '; put_int(s, cnt); s += '\t'; put_int(s, gl); s += ' '; put_float(s, gscore); s += " GL\n";
                if (gscore > min_score && (int)cnt > min_kmer) { ++track[q.taxid][gl]; score_track[q.taxid][gl] += gscore; }
                if (q.tax_score >= min_tax_score && gscore > min_score && (int)cnt > min_kmer) { ++track_tax[q.taxid][gl]; score_track_tax[q.taxid][gl] += gscore; }
            }
            ofs << s;
            recs.clear(); bases.clear(); off.assign(1, 0);
            return true;
        };
        bool finished = false;
        std::string line;
        while (!finished) {  // the line loop of :565-610
            std::getline(ifs, line);
            if ((long)ifs.tellg() == -1) finished = true;
            const size_t p1 = line.find('\t');
            const std::string hdr = line.substr(0, p1);
            const size_t p2 = line.find('\t', p1 + 1);
            const std::string read_buff = line.substr(p1 + 1, p2 - p1 - 1);
            const size_t p3 = line.find('\t', p2 + 1);
            std::istringstream istrm2(line.substr(p2 + 1, p3 - p2 - 1));
            float score1 = 0, score2 = 0, score3 = 0;
            istrm2 >> score1 >> score2 >> score3;
            if (score3 == -1) continue;  // the read lacks valid k-mers
            const size_t p4 = line.find('\t', p3 + 1);
            const size_t p5 = line.find('\t', p4 + 1);
            std::istringstream istrm(line.substr(p4 + 1, p5 - p4));
            uint32_t taxid = 0;
            float tax_score = 0.0;
            std::string match_type;
            istrm >> taxid >> tax_score >> match_type;
            if (match_type[0] == 'N' || match_type[0] == 'R') taxid = 0;  // NoDbHits / NoMatch / ReadTooShort
            recs.push_back(Rec{hdr, read_buff, taxid, tax_score});
            bases.insert(bases.end(), read_buff.begin(), read_buff.end());
            off.push_back(bases.size());
            if (recs.size() >= kBatch && !flush()) return fail("classify");
        }
        if (!flush()) return fail("classify");
        for (auto& a : track) for (auto& b : a.second) merge_cnt[b.first][a.first] += b.second;
        for (auto& a : track_tax) for (auto& b : a.second) merge_cnt_tax[b.first][a.first] += b.second;
        for (auto& a : score_track) for (auto& b : a.second) {
            auto& m = score_merge[b.first];
            if (m.find(a.first) == m.end()) m[a.first] = b.second; else m[a.first] += b.second;
        }
        for (auto& a : score_track_tax) for (auto& b : a.second) {
            auto& m = score_merge_tax[b.first];
            if (m.find(a.first) == m.end()) m[a.first] = b.second; else m[a.first] += b.second;
        }
    }
    gzFile gz = gzopen(genefile.c_str(), "rb");
    if (!gz) { std::cout << "Unable to unzip gene annotation table: " << genefile << std::endl; lmat_ctx_destroy(ctx); return -1; }
    std::string o1 = ofbase, o2;
    o1 += '.'; put_float(o1, min_score); o1 += '.'; put_int(o1, min_kmer); o1 += ".genesummary";
    o2 = o1 + ".min_tax_score.";
    put_float(o2, min_tax_score);
    std::ofstream sum_ofs(o1.c_str()), sum_ofs_tax(o2.c_str());
    if (!sum_ofs) { std::cerr << "Can't write to " << o1 << std::endl; return -1; }
    if (!sum_ofs_tax) { std::cerr << "Can't write to " << o2 << std::endl; return -1; }
    static char buff[20000];
    while (gzgets(gz, buff, sizeof buff)) {
        const size_t n = strlen(buff);
        if (n && buff[n - 1] == '\n') buff[n - 1] = 0;
        std::istringstream istrm(buff);
        uint32_t tid = 0, gid = 0;
        istrm >> tid >> gid;
        auto emit = [&](std::ofstream& out, std::map<uint32_t, std::map<uint32_t, uint32_t>>& cnts, std::map<uint32_t, std::map<uint32_t, float>>& scores) {
            auto it = cnts.find(gid);
            if (it == cnts.end()) return;
            for (auto& ti : it->second) {
                std::string s;
                put_float(s, scores[gid][ti.first] / (float)ti.second); s += '\t'; put_int(s, ti.second); s += '\t'; put_int(s, ti.first); s += '\t';
                out << s << buff << std::endl;
            }
        };
        emit(sum_ofs, merge_cnt, score_merge);
        emit(sum_ofs_tax, merge_cnt_tax, score_merge_tax);
    }
    gzclose(gz);
    std::cout << "query time: " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() << std::endl;
    lmat_ctx_destroy(ctx);
    return 0;
}
