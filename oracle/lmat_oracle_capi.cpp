// lmat_oracle_capi.cpp -- C entry points of the CPU oracle for ctypes callers.
// TEST INFRASTRUCTURE ONLY (see lmat_oracle.hpp header): used by tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker.
#include <thread>
#include "lmat_oracle.hpp"
#include "gene_oracle.hpp"

using namespace orc;

struct orc_ctx {
    Taxonomy tax;
    KmerDb db;
    Options opt;
    NullModel nm;
    std::string err;
    std::string text;  // last classify output
};

extern "C" {

orc_ctx* orc_create(const char* tree, const char* depth, const char* rank, const char* idmap, const char* plasmids) {
    orc_ctx* c = new orc_ctx();
    if (idmap && *idmap && !c->tax.load_idmap(idmap)) { delete c; return nullptr; }
    if (rank && *rank) c->tax.load_rank(rank);
    if (plasmids && *plasmids) c->tax.load_plasmids(plasmids);
    if (!c->tax.load_tree(tree) || !c->tax.load_depth(depth)) { delete c; return nullptr; }
    if (!(idmap && *idmap) && !c->tax.idmap_from_tree()) { delete c; return nullptr; }
    return c;
}
void orc_destroy(orc_ctx* c) { delete c; }
const char* orc_error(orc_ctx* c) { return c->err.c_str(); }

int orc_set_build_options(orc_ctx* c, int tid_cutoff, const char* rank_map, const char* human, const char* adaptors) {
    return c->db.set_options(tid_cutoff, rank_map ? rank_map : "", human ? human : "", adaptors ? adaptors : "") ? 0 : -1;
}
// construct_labels' mean / stdev of a score list (Classifier::score_stats)
void orc_score_stats(const float* score, uint32_t n, float* log_avg, float* stdev) {
    const Classifier::ScoreStats s = Classifier::score_stats(score, n);
    *log_avg = s.log_avg;
    *stdev = s.stdev;
}

// rkmer.hpp's retrieve_kmer_labels per read, one text line each (see Classifier::rkmer_trace)
const char* orc_rkmer_trace(orc_ctx* c, const uint8_t* bases, const uint64_t* off, uint64_t n, int k, int permissive) {
    Options o = c->opt;  // run-time pruning (orc_set_label_modes) rides along
    o.rand_mode = true;
    o.permissive = permissive != 0;
    Classifier cls(c->tax, c->db, o, nullptr);
    std::ostringstream out;
    for (uint64_t i = 0; i < n; ++i) {
        std::string read((const char*)bases + off[i], (size_t)(off[i + 1] - off[i]));
        out << "R " << i << " " << cls.rkmer_trace(read, k) << "\n";
    }
    c->text = out.str();
    return c->text.c_str();
}

// rand_read_label over the given reads: rows (ascending taxid) of max label_prob and hit count per GC bucket
int orc_rand_label(orc_ctx* c, const uint8_t* bases, const uint64_t* off, uint64_t n, int k, const uint8_t* gc_bucket, uint32_t nb,
                   uint32_t* tid, float* mx, int32_t* ct, uint32_t cap) {
    Options o = c->opt;
    o.rand_mode = true;
    Classifier cls(c->tax, c->db, o, nullptr);
    Classifier::RandTable tab;
    for (uint64_t i = 0; i < n; ++i) {
        std::string read((const char*)bases + off[i], (size_t)(off[i + 1] - off[i]));
        cls.rand_proc_line(read, k, gc_bucket[i], nb, tab);
    }
    uint32_t r = 0;
    for (auto& kv : tab) {
        if (r < cap) {
            tid[r] = kv.first;
            for (uint32_t b = 0; b < nb; ++b) { mx[(size_t)r * nb + b] = kv.second.first[b]; ct[(size_t)r * nb + b] = kv.second.second[b]; }
        }
        ++r;
    }
    return (int)r;
}

int orc_set_label_modes(orc_ctx* c, int permissive, int tid_cutoff, const char* rank_map) {
    c->opt.permissive = permissive != 0;
    c->opt.max_count = tid_cutoff > 0 ? (uint16_t)tid_cutoff : 0xFFFF;
    c->opt.tid_rank_map.clear();
    if (tid_cutoff > 0 && rank_map && *rank_map) {
        FILE* f = fopen(rank_map, "r");
        if (!f) return -1;
        int a, b;
        while (fscanf(f, "%d%d", &a, &b) > 0) c->opt.tid_rank_map[(uint32_t)a] = (uint32_t)b;
        fclose(f);
    }
    return 0;
}
int orc_load_null_models(orc_ctx* c, const char* list_fn) { return c->nm.load(list_fn) ? 0 : -1; }
int orc_add_taxhisto(orc_ctx* c, const char* fn) { return c->db.add_taxhisto(fn, c->tax, &c->err) ? 0 : -1; }
int orc_db_k(orc_ctx* c) { return c->db.k; }
void orc_set_k(orc_ctx* c, int k) { c->db.k = k; }
uint64_t orc_db_size(orc_ctx* c) { return c->db.table.size(); }

// DB entries supplied directly as 32-bit ids in stored order (used when the lists
// come from another lookup source); ids are mapped 32->16 like add_data would.
int orc_add_list32(orc_ctx* c, uint64_t kmer, const uint32_t* tids, int n) {
    std::vector<KmerDb::code_t> l;
    for (int i = 0; i < n; ++i) {
        auto b = c->tax.br.find(tids[i]);
        if (b == c->tax.br.end() || b->second == 0) { c->err = "tid not in 32->16 map"; return -1; }
        l.push_back(b->second);
    }
    c->db.add_list(kmer, l);
    return 0;
}

// lookup: returns taxidCount, writes up to cap converted 32-bit ids in stored order; -1 = miss
int orc_lookup(orc_ctx* c, uint64_t kmer, uint32_t* out, int cap) {
    const std::vector<KmerDb::code_t>* l = c->db.lookup(kmer);
    if (!l) return -1;
    for (int i = 0; i < (int)l->size() && i < cap; ++i) {
        auto cv = c->tax.conv.find((*l)[i]);
        out[i] = cv == c->tax.conv.end() ? 0 : cv->second;
    }
    return (int)l->size();
}

// the sequence TaxNodeStat hands out under the current label modes (run-time pruning); -1 = miss
int orc_lookup_rt(orc_ctx* c, uint64_t kmer, uint32_t* out, int cap) {
    const std::vector<KmerDb::code_t>* l = c->db.lookup(kmer);
    if (!l) return -1;
    Classifier cls(c->tax, c->db, c->opt);
    std::vector<tid_t> seq;
    uint16_t count = 0;
    cls.taxnodestat_sequence(l, kmer, seq, count);
    for (int i = 0; i < (int)seq.size() && i < cap; ++i) out[i] = seq[i];
    return (int)count;
}

int orc_path_to_root(orc_ctx* c, uint32_t tid, uint32_t* out, int cap) {
    std::vector<tid_t> p;
    c->tax.path_to_root(tid, p);
    for (int i = 0; i < (int)p.size() && i < cap; ++i) out[i] = p[i];
    return (int)p.size();
}

// The decision step alone (std::sort with TCmp + findReadLabelVer2, src/read_label.cpp:892-896, 284-419) on a
// given candidate set: scores as construct_labels holds them after the human bias, stdev as it printed it.
// Lets the tests replay records of the reference's own example run.  match: 0 Direct, 1 Multi, 2 Partial, 3 NoMatch, 4 LCA error.
int orc_replay_decision(orc_ctx* c, const uint32_t* tids, const float* scores, int n, float stdev, uint32_t* call_tid,
                        float* call_score, int* match) {
    Classifier cls(c->tax, c->db, c->opt);
    std::vector<ufpair_t> rank_label;
    std::unordered_map<tid_t, float> all_cand_set;
    float top = 0;
    for (int i = 0; i < n; ++i) {
        rank_label.push_back(std::make_pair((tid_t)tids[i], scores[i]));
        all_cand_set[tids[i]] = scores[i];
        if (i == 0 || scores[i] > top) top = scores[i];
    }
    Classifier::TCmp tcmp{&c->tax};
    std::sort(rank_label.begin(), rank_label.end(), tcmp);
    std::list<ufpair_t> valid_cand;
    const auto res = cls.find_read_label(rank_label, stdev * c->opt.diff_thresh, valid_cand, all_cand_set, top);
    *call_tid = res.first.first;
    *call_score = res.first.second;
    *match = (int)res.second;
    return 0;
}

void orc_set_options(orc_ctx* c, float sdiff, float hbias, int prn_all, int screen_phix, float min_score, int min_kmer,
                     int min_fnd_kmer, int prn_read, int fastq) {
    c->opt.diff_thresh = sdiff;
    c->opt.diff_thresh2 = hbias;
    c->opt.prn_all = prn_all != 0;
    c->opt.screen_phix = screen_phix != 0;
    c->opt.min_score = min_score;
    c->opt.min_kmer = min_kmer;
    c->opt.min_fnd_kmer = min_fnd_kmer;
    c->opt.prn_read = prn_read != 0;
    c->opt.fastq = fastq != 0;
}

// unique canonical k-mers of one read in first-occurrence order (read_label.cpp:978-1017)
int orc_extract(const char* read, int len, int k, uint64_t* kmers, int* pos, int cap, int* valid_kmers, int* bin_sel) {
    Taxonomy tax;
    KmerDb db;
    Options opt;
    Classifier cls(tax, db, opt);
    std::vector<label_info_t> label_vec(len >= k ? len - k + 1 : 0, std::make_pair((int16_t)-1, tax_data_t()));
    std::list<tid_t> lst;
    hmap_t a, b;
    ReadTrace tr;
    if (len < k) { *valid_kmers = 0; *bin_sel = 0; return 0; }
    std::pair<int, int> r = cls.retrieve_kmer_labels(read, len, k, label_vec, lst, a, b, &tr);
    *valid_kmers = r.first;
    *bin_sel = r.second;
    int n = (int)tr.uniq_kmers.size();
    for (int i = 0; i < n && i < cap; ++i) { kmers[i] = tr.uniq_kmers[i]; pos[i] = tr.uniq_pos[i]; }
    return n;
}

// Classify n reads given as a blob + offsets (n+1 entries).  Headers are "r<first_index+i>".
// Produces exactly the bytes read_label -t 1 would write to <prefix>0.out for these reads.
// Returns length of the text; fetch it with orc_text().  Tallies are returned in the
// caller's arrays when non-null: counts/scores for up to cap distinct taxids.
long orc_classify(orc_ctx* c, const char* blob, const uint64_t* off, long n, long first_index, int k_size,
                  uint32_t* t_tid, int* t_cnt, float* t_score, int t_cap, int* n_tids, int* nomatch3) {
    Classifier cls(c->tax, c->db, c->opt, c->nm.loaded ? &c->nm : nullptr);
    Tallies tl;
    std::ostringstream ofs;
    for (long i = 0; i < n; ++i) {
        std::string read(blob + off[i], blob + off[i + 1]);
        ofs << "r" << (first_index + i) << "\t";
        if (c->opt.prn_read) ofs << read << "\t"; else ofs << "X" << "\t";
        cls.proc_line((int)read.length(), read, k_size, ofs, tl);
    }
    c->text = ofs.str();
    if (n_tids) {
        int j = 0;
        for (auto it = tl.count.begin(); it != tl.count.end() && j < t_cap; ++it, ++j) {
            t_tid[j] = it->first;
            t_cnt[j] = it->second;
            t_score[j] = tl.score[it->first];
        }
        *n_tids = (int)tl.count.size();
    }
    if (nomatch3) {
        for (int i = 0; i < 3; ++i) nomatch3[i] = tl.nomatch.count(i) ? tl.nomatch[i] : 0;
    }
    return (long)c->text.size();
}
const char* orc_text(orc_ctx* c) { return c->text.c_str(); }

// bulk form of orc_add_list32: counts[i] ids at tids[i*stride ..]
int orc_add_lists32(orc_ctx* c, const uint64_t* kmers, const uint32_t* counts, const uint32_t* tids, uint32_t stride,
                    uint64_t n) {
    for (uint64_t i = 0; i < n; ++i) {
        if (!counts[i]) continue;
        if (counts[i] > stride) { c->err = "list longer than stride"; return -1; }
        if (orc_add_list32(c, kmers[i], tids + i * stride, (int)counts[i]) != 0) return -1;
    }
    return 0;
}

// Throughput form for the CPU baseline: classifies the reads on nthreads host threads
// (reads are independent, read_label.cpp:1742-1748) and discards the text.  Returns the
// number of reads that produced a taxid call.
long orc_classify_mt(orc_ctx* c, const char* blob, const uint64_t* off, long n, int k_size, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    std::vector<long> calls(nthreads, 0);
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) {
        th.emplace_back([&, t]() {
            Classifier cls(c->tax, c->db, c->opt, c->nm.loaded ? &c->nm : nullptr);
            Tallies tl;
            std::ostringstream ofs;
            for (long i = t; i < n; i += nthreads) {
                std::string read(blob + off[i], blob + off[i + 1]);
                cls.proc_line((int)read.length(), read, k_size, ofs, tl);
                if (ofs.tellp() > (1 << 20)) { ofs.str(""); }
            }
            long s = 0;
            for (auto& kv : tl.count) s += kv.second;
            calls[t] = s;
        });
    }
    for (auto& x : th) x.join();
    long s = 0;
    for (long v : calls) s += v;
    return s;
}

// Summaries from a list of per-read calls (tid, score, is_nomatch), in read order: the tally rule of
// proc_line (read_label.cpp:1248-1268) followed by the .fastsummary/.nomatchsum writers (:1801-1867).
long orc_summaries_from_calls(orc_ctx* c, const uint32_t* tids, const float* scores, const int* nomatch, long n,
                              int extra_short, int extra_nodb, char* fastsummary, long fs_cap, char* nomatchsum,
                              long nm_cap) {
    Classifier cls(c->tax, c->db, c->opt, c->nm.loaded ? &c->nm : nullptr);
    Tallies tl;
    if (extra_short) tl.nomatch[kReadTooShort] = extra_short;
    if (extra_nodb) tl.nomatch[kNoDbHits] = extra_nodb;
    for (long i = 0; i < n; ++i)
        cls.tally_call(tl, std::make_pair(std::make_pair(tids[i], scores[i]), nomatch[i] ? kNoMatchT : kDirectMatch),
                       c->opt.min_kmer);
    RunOutputs ro;
    write_summaries(tl, "", ro);
    snprintf(fastsummary, fs_cap, "%s", ro.fastsummary.c_str());
    snprintf(nomatchsum, nm_cap, "%s", ro.nomatchsum.c_str());
    return (long)tl.count.size();
}

// Whole-file run (FASTA/FASTQ parsing included), as the CLI does.
long orc_run_file(orc_ctx* c, const char* query, int k_size, const char* rank_ids, char* fastsummary, long fs_cap,
                  char* nomatchsum, long nm_cap) {
    std::ifstream in(query);
    if (!in) { c->err = "cannot open query"; return -1; }
    Classifier cls(c->tax, c->db, c->opt, c->nm.loaded ? &c->nm : nullptr);
    RunOutputs ro = run_reads(cls, in, k_size, rank_ids ? rank_ids : "");
    c->text = ro.out;
    if (fastsummary) snprintf(fastsummary, fs_cap, "%s", ro.fastsummary.c_str());
    if (nomatchsum) snprintf(nomatchsum, nm_cap, "%s", ro.nomatchsum.c_str());
    return (long)c->text.size();
}

// ---- gene_label (gene_oracle.hpp) -----------------------------------------------------------------------------
void* orc_gene_create() { return new gene_oracle::GeneDb(); }
void orc_gene_destroy(void* g) { delete (gene_oracle::GeneDb*)g; }
int orc_gene_add(void* g, const char* fn) { std::string e; return ((gene_oracle::GeneDb*)g)->add_taxhisto(fn, &e) ? 0 : -1; }
int orc_gene_k(void* g) { return ((gene_oracle::GeneDb*)g)->k; }
uint64_t orc_gene_size(void* g) { return ((gene_oracle::GeneDb*)g)->table.size(); }
int orc_gene_lookup(void* g, uint64_t kmer, uint32_t* out, int cap) {
    auto& t = ((gene_oracle::GeneDb*)g)->table;
    auto it = t.find(kmer);
    if (it == t.end()) return 0;
    for (int i = 0; i < (int)it->second.size() && i < cap; ++i) out[i] = it->second[i];
    return (int)it->second.size();
}
// per read: any (0/1), gene id, top votes, distinct valid k-mers, score
void orc_gene_label(void* g, const char* blob, const uint64_t* off, long n, int k, uint8_t* any, uint32_t* gid, uint32_t* top,
                    uint32_t* cnt, float* score) {
    for (long i = 0; i < n; ++i) {
        const gene_oracle::GeneCall c = gene_oracle::label_read(*(gene_oracle::GeneDb*)g, blob + off[i], (int)(off[i + 1] - off[i]), k);
        any[i] = c.any; gid[i] = c.gid; top[i] = c.top; cnt[i] = c.cnt; score[i] = c.score;
    }
}
int orc_gene_run(void* g, const char* list_fn, const char* ofbase, const char* genefile, float min_score, int min_kmer, float min_tax_score) {
    std::string e;
    return gene_oracle::run_files(*(gene_oracle::GeneDb*)g, list_fn, ofbase, genefile, min_score, min_kmer, min_tax_score, &e) ? 0 : -1;
}
int orc_gene_replay(const char* list_fn, const char* gl_list_fn, const char* ofbase, const char* genefile, float min_score, int min_kmer,
                    float min_tax_score) {
    std::string e;
    return gene_oracle::replay_files(list_fn, gl_list_fn, ofbase, genefile, min_score, min_kmer, min_tax_score, &e) ? 0 : -1;
}

}  // extern "C"
