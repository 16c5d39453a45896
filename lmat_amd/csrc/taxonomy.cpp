// taxonomy.cpp -- loads the reference's taxonomy inputs and flattens them into the
// dense tables the kernels use.
//
// Replaces, for the classification path only:
//   TaxTree<uint32_t>(file) + TaxNode::read      src/kmerdb/TaxTree.hpp:24-57, TaxNode.hpp:131-147
//   TaxTree::getPathToRoot                        src/kmerdb/TaxTree.hpp:60-91
//   depth map / gRank_table / conv_map loading    src/read_label.cpp:1560-1602
//   isHuman / isPhiX / isPlasmid / badGenomes     include/tid_checks.hpp:13-28, read_label.cpp:69,82-104
#include <algorithm>
#include <cstdio>
#include <fstream>
#include <queue>
#include <sstream>
#include "lmat_internal.hpp"
#include "dbbuild.hpp"

namespace lmat {

int set_err(lmat_ctx* c, int code, const std::string& msg) {
    if (c) { c->err = msg; c->last_rc = code; }
    return code;
}

static bool is_human32(uint32_t t) { return t == 9606 || t == 63221 || t == 741158; }
static bool is_phix32(uint32_t t) { return t == 374840 || t == 10847 || t == 32630; }

int load_taxonomy_files(lmat_ctx* c, const char* tree_fn, const char* depth_fn, const char* rank_fn,
                        const char* idmap_fn, const char* plasmid_fn) {
    HostTaxonomy& T = c->tax;
    T = HostTaxonomy();
    if (!tree_fn || !depth_fn) return set_err(c, LMAT_E_ARG, "tree and depth files are required");

    // --- tree: two comment lines, one count line, then "id nchild child.. parent" / name pairs
    std::unordered_map<uint32_t, uint32_t> parent;
    {
        std::ifstream in(tree_fn);
        if (!in.is_open()) return set_err(c, LMAT_E_IO, std::string("failed to open ") + tree_fn + " for reading");
        std::string line;
        std::getline(in, line);
        std::getline(in, line);
        std::getline(in, line);  // count (ignored, as upstream)
        while (std::getline(in, line)) {
            std::istringstream is(line);
            std::vector<uint32_t> tok;
            uint64_t v;
            while (is >> v) tok.push_back((uint32_t)v);
            std::string name;
            std::getline(in, name);
            if (tok.size() < 3) {
                if (tok.empty()) continue;  // tolerate a trailing blank line
                return set_err(c, LMAT_E_IO, "malformed taxonomy node line: " + line);
            }
            parent[tok[0]] = tok.back();
        }
    }
    // --- depth file
    std::unordered_map<uint32_t, uint32_t> fdepth;
    {
        std::ifstream in(depth_fn);
        if (!in) return set_err(c, LMAT_E_IO, std::string("ERROR! Unable to open: ") + depth_fn);
        uint32_t t, d;
        while (in >> t >> d) fdepth[t] = d;
    }
    // --- rank file: first entry per taxid wins (std::map::insert)
    std::unordered_map<uint32_t, std::string> rank;
    if (rank_fn && *rank_fn) {
        std::ifstream in(rank_fn);
        if (!in) return set_err(c, LMAT_E_IO, std::string("cannot open rank file ") + rank_fn);
        uint32_t t;
        std::string r;
        while (in >> t >> r) rank.insert(std::make_pair(t, r));
    }
    // --- 32 -> 16 map; later lines overwrite (operator[])
    std::vector<uint32_t> wide_ids;   // no map and a tree beyond 65534 nodes: all its ids
    T.conv.assign(65536, 0);
    if (idmap_fn && *idmap_fn) {
        FILE* f = fopen(idmap_fn, "r");
        if (!f) return set_err(c, LMAT_E_IO, std::string("ERROR! Unable to read 16-bit map file:") + idmap_fn);
        int src;
        short dest;
        while (fscanf(f, "%d%hd", &src, &dest) > 0) {
            T.conv[(uint16_t)dest] = (uint32_t)src;
            T.br[(uint32_t)src] = (uint16_t)dest;
        }
        fclose(f);
    } else {
        // no map: a database of 32-bit taxids (TID_SIZE=32 builds, CMakeLists.txt:92-105).  Storage codes are the
        // ranks of the tree's node ids, the same rule the ingest uses (dbbuild.cpp:idmap_from_tree).
        std::vector<uint32_t> ids;
        std::string e;
        if (!tree_node_ids(tree_fn, ids, e, true)) return set_err(c, LMAT_E_IO, e);
        if (ids.size() > 65534) {
            // more nodes than 16-bit codes: the database stores its taxids as they are (32 bits), every node of the tree is an
            // id the engine may meet, and the classify kernels' WIDE classes take the reads (HostTaxonomy::wide)
            T.wide = true;
            wide_ids.swap(ids);
        } else
            for (size_t i = 0; i < ids.size(); ++i) {
                T.conv[i + 1] = ids[i];
                T.br[ids[i]] = (uint16_t)(i + 1);
            }
    }
    std::unordered_set<uint32_t> low_plasmid;
    if (plasmid_fn && *plasmid_fn) {
        std::ifstream in(plasmid_fn);
        if (!in) return set_err(c, LMAT_E_IO, std::string("Unexpected reading error (plasmids): ") + plasmid_fn);
        uint32_t p;
        while (in >> p) low_plasmid.insert(p);
    }

    // --- internal index space: map targets, 9606 when any human id is mapped, all ancestors
    std::unordered_set<uint32_t> S;
    bool any_human = false;
    for (uint32_t t16 = 0; t16 < 65536; ++t16) {
        uint32_t t = T.conv[t16];
        if (t) {
            S.insert(t);
            if (is_human32(t)) any_human = true;
        }
    }
    for (uint32_t t : wide_ids) { S.insert(t); if (is_human32(t)) any_human = true; }
    if (any_human) S.insert(9606);
    // the PhiX short-circuit reports ART_SEQ_TID 32630 (read_label.cpp:842): give it a tally slot
    for (uint32_t t16 = 0; t16 < 65536; ++t16)
        if (T.conv[t16] && is_phix32(T.conv[t16])) { S.insert(32630); break; }
    for (uint32_t t : wide_ids) if (is_phix32(t)) { S.insert(32630); break; }
    std::vector<uint32_t> work(S.begin(), S.end());
    for (size_t i = 0; i < work.size(); ++i) {
        uint32_t cur = work[i];
        int guard = 0;
        while (true) {
            auto it = parent.find(cur);
            if (it == parent.end()) {
                if (cur != work[i])
                    return set_err(c, LMAT_E_TAXONOMY, "failed to find parent TaxNode " + std::to_string(cur) +
                                                           " above taxid " + std::to_string(work[i]));
                break;  // taxid itself not in tree: empty path, as TaxTree::getPathToRoot
            }
            if (it->second == cur) break;
            cur = it->second;
            if (S.insert(cur).second) work.push_back(cur);
            if (++guard > 10000) return set_err(c, LMAT_E_TAXONOMY, "cycle in taxonomy above " + std::to_string(work[i]));
        }
    }
    if (S.size() > 65534) T.wide = true;   // (also a 16-bit map whose ids, with their ancestors, pass 65534: 32-bit internal ids)
    std::vector<uint32_t> ids(S.begin(), S.end());
    std::sort(ids.begin(), ids.end());
    T.n = (uint32_t)ids.size();
    T.tid32.assign(T.n + 1, 0);
    T.fdepth.assign(T.n + 1, 0);
    T.flags.assign(T.n + 1, 0);
    T.species_of.assign(T.n + 1, 0);
    T.path_off.assign(T.n + 1, 0);
    T.path_len.assign(T.n + 1, 0);
    for (uint32_t i = 0; i < T.n; ++i) {
        T.tid32[i + 1] = ids[i];
        T.index_of[ids[i]] = i + 1;
    }
    for (uint32_t i = 1; i <= T.n; ++i) {
        const uint32_t t = T.tid32[i];
        auto d = fdepth.find(t);
        if (d != fdepth.end() && d->second > 32767)
            return set_err(c, LMAT_E_TAXONOMY, "depth of taxid " + std::to_string(t) + " above 32767");  // bit 15 is the lineage flag
        T.fdepth[i] = d == fdepth.end() ? 0 : (uint16_t)d->second;
        uint8_t f = 0;
        auto r = rank.find(t);
        if (r != rank.end() && r->second == "strain") f |= kFlagStrain;
        if (is_human32(t)) f |= kFlagHuman;
        if (is_phix32(t)) f |= kFlagPhiX;
        if ((t >= 10000000 && t < 11000000) || low_plasmid.count(t)) f |= kFlagPlasmid;
        T.flags[i] = f;
        // path to root
        T.path_off[i] = (uint32_t)T.paths.size();
        uint32_t cur = t;
        auto it = parent.find(cur);
        if (it != parent.end()) {
            while (it->second != cur) {
                cur = it->second;
                T.paths.push_back(T.index_of[cur]);
                it = parent.find(cur);
            }
        }
        size_t len = T.paths.size() - T.path_off[i];
        if (len > 65535) return set_err(c, LMAT_E_TAXONOMY, "taxonomy too deep");
        T.path_len[i] = (uint16_t)len;
    }
    // species_of: first ancestor whose rank is "species" (read_label.cpp:1154-1163)
    for (uint32_t i = 1; i <= T.n; ++i) {
        if (!(T.flags[i] & kFlagStrain)) continue;
        for (uint32_t p = 0; p < T.path_len[i]; ++p) {
            const uint32_t a = T.paths[T.path_off[i] + p];
            auto r = rank.find(T.tid32[a]);
            if (r != rank.end() && r->second == "species") {
                T.species_of[i] = a;
                break;
            }
        }
    }
    auto h = T.index_of.find(9606);
    T.human_idx = h == T.index_of.end() ? 0 : h->second;
    T.loaded = true;
    build_euler_intervals(T);
    return upload_taxonomy(c);
}

// Euler-tour intervals over the forest defined by paths[]: isAncestor (read_label.cpp:138-150, "a is on
// b's getPathToRoot") becomes tin[a] < tin[b] && tout[b] <= tout[a] without touching memory per level.
// Ids that are not tree nodes (empty path, nobody's ancestor) are trees of one node: an interval of one tick of their own,
// numbered behind the real trees (every tin is unique; k4_wave's "related = the intervals intersect" relies on that).
void build_euler_intervals(HostTaxonomy& T) {
    const uint32_t n = T.n;
    T.tin.assign(n + 1, T.wide ? 0xFFFFFFFFu : 0xFFFFu);
    T.tout.assign(n + 1, T.wide ? 0xFFFFFFFFu : 0xFFFFu);
    std::vector<std::vector<uint32_t>> children(n + 1);
    std::vector<uint8_t> is_child(n + 1, 0), has_child(n + 1, 0);
    for (uint32_t i = 1; i <= n; ++i)
        if (T.path_len[i]) {
            const uint32_t par = T.paths[T.path_off[i]];
            children[par].push_back(i);
            is_child[i] = 1;
            has_child[par] = 1;
        }
    uint32_t clock = 0;
    std::vector<std::pair<uint32_t, size_t>> st;
    for (uint32_t r = 1; r <= n; ++r) {
        if (is_child[r]) continue;
        if (!has_child[r]) continue;  // roots that own a subtree (lone nodes are numbered behind them, below)
        st.push_back(std::make_pair(r, (size_t)0));
        T.tin[r] = clock++;
        while (!st.empty()) {
            auto& top = st.back();
            if (top.second < children[top.first].size()) {
                const uint32_t ch = children[top.first][top.second++];
                T.tin[ch] = clock++;
                st.push_back(std::make_pair(ch, (size_t)0));
            } else {
                T.tout[top.first] = clock - 1;
                st.pop_back();
            }
        }
    }
    // a node without parent and children is a tree of its own: an interval of one tick that nothing else touches (every node's
    // tin is then unique, and "related" is "the intervals intersect" without exceptions -- the decision step on the wave leans on it)
    for (uint32_t r = 1; r <= n; ++r)
        if (!is_child[r] && !has_child[r]) { T.tin[r] = T.tout[r] = clock; ++clock; }
}

// One raw DB list -> arena record.  Restates the per-k-mer part of retrieve_kmer_labels
// (src/read_label.cpp:1028-1134): 16->32 conversion, human folding, ignored ids, int16
// store of the raw count, std::sort by depth descending, leaf-most filter.  It is a pure
// function of the list, so it is evaluated once per distinct list at DB build time.
namespace {
struct RankPair {  // MyPair, src/kmerdb/SortedDb.hpp:129-139
    unsigned int first;
    uint32_t second;
    RankPair(unsigned int f, uint32_t s) : first(f), second(s) {}
    bool operator<(const RankPair& o) const { return first < o.first; }
};
}  // namespace

// seq32: the list's taxids as the reference's cursor hands them out (stored order, 32 bits); raw_units: how the stored list
// is kept behind the record for lookups (16-bit codes, or (low, high) pairs of a wide database); wide: ids in the record are pairs
static bool build_list_record_core(lmat_ctx* c, std::vector<uint32_t>& seq, const std::vector<uint16_t>& raw_units, size_t n_raw,
                                   bool wide, std::vector<uint16_t>& rec) {
    const HostTaxonomy& T = c->tax;
    // what TaxNodeStat::begin / next hand to the caller (TaxNodeStat.hpp:60-256): the stored ids converted
    // 16 -> 32, or, with run-time pruning (-g N [-m ranks]) on a list longer than N, the survivors of the
    // rank-priority queue in pop order (or just the first stored id when no rank map was given)
    uint32_t count = (uint32_t)seq.size();
    if (c->rt_tid_cut > 0 && (int)count > c->rt_tid_cut) {
        if (c->rt_rank_map.empty()) {
            seq.resize(1);
            count = 1;
        } else {
            std::priority_queue<RankPair> q;
            for (uint32_t t : seq) {
                auto r = c->rt_rank_map.find(t);
                q.push(RankPair(r == c->rt_rank_map.end() ? 0u : r->second, t));
            }
            while (!q.empty()) {
                const unsigned cur = q.top().first;
                while (q.top().first == cur) { q.pop(); if (q.empty()) break; }
                if ((int)q.size() <= c->rt_tid_cut) { count = (uint32_t)q.size(); break; }
            }
            if (q.size() == 0) { count = 1; q.push(RankPair(1, 1)); }
            seq.clear();
            for (uint32_t j = 0; j < count; ++j) { seq.push_back(q.top().second); q.pop(); }
        }
    }
    std::vector<uint32_t> obs;
    bool seen_human = false, neg_first = false;
    unsigned dcnt = 0;
    for (size_t i = 0; i < seq.size(); ++i) {  // read_label.cpp:1031-1066
        uint32_t tid = seq[i];
        if (is_human32(tid) && !c->rand_mode) {  // rkmer.hpp:121-123 has no human folding
            if (seen_human) continue;
            tid = 9606;
            seen_human = true;
        }
        if (tid == 20999999 || tid == 12721 || tid == 693660) continue;
        if (dcnt == 0) neg_first = ((int16_t)(uint16_t)count) < 0;
        auto it = T.index_of.find(tid);
        if (it == T.index_of.end()) {
            set_err(c, LMAT_E_TAXONOMY, "taxid " + std::to_string(tid) + " missing from internal index");
            return false;
        }
        obs.push_back(it->second);
        dcnt++;
    }
    std::vector<uint32_t> kept;
    auto add_unique = [&kept](uint32_t t) { if (std::find(kept.begin(), kept.end(), t) == kept.end()) kept.push_back(t); };
    if (c->permissive) {
        // -s (read_label.cpp:1050-1058,1075-1102): every accepted id joins the position set in list order, then,
        // walking the depth-sorted ids until one of depth 0, all their ancestors do; no leaf-most filter
        for (uint32_t t : obs) add_unique(t);
        std::vector<uint32_t> sorted(obs);
        std::sort(sorted.begin(), sorted.end(), [&T](uint32_t a, uint32_t b) { return (int)T.fdepth[a] > (int)T.fdepth[b]; });
        for (uint32_t t : sorted) {
            if (T.fdepth[t] == 0) break;
            for (uint32_t p = 0; p < T.path_len[t]; ++p) add_unique(T.paths[T.path_off[t] + p]);
        }
    } else {
        // CmpDepth1 (read_label.cpp:169-177) through the same std::sort the reference calls, then the leaf-most filter
        std::sort(obs.begin(), obs.end(), [&T](uint32_t a, uint32_t b) { return (int)T.fdepth[a] > (int)T.fdepth[b]; });
        std::unordered_set<uint32_t> non_leaf, have;
        for (size_t i = 0; i < obs.size(); ++i) {
            const uint32_t t = obs[i];
            if (non_leaf.count(t)) continue;
            if (!have.insert(t).second) {
                set_err(c, LMAT_E_IO, "taxid list holds taxid " + std::to_string(T.tid32[t]) + " twice");
                return false;
            }
            kept.push_back(t);
            for (uint32_t p = 0; p < T.path_len[t]; ++p) non_leaf.insert(T.paths[T.path_off[t] + p]);
        }
    }
    if (kept.size() > 65535 || n_raw > 65535) {
        set_err(c, LMAT_E_CAPACITY, "taxid list too long");
        return false;
    }
    std::vector<uint32_t> asc(kept);
    std::sort(asc.begin(), asc.end());  // internal index order == 32-bit taxid order
    rec.clear();
    rec.push_back(neg_first ? kListNegFirst : 0);
    rec.push_back((uint16_t)kept.size());
    rec.push_back((uint16_t)n_raw);
    auto put = [&](uint32_t t) { rec.push_back((uint16_t)t); if (wide) rec.push_back((uint16_t)(t >> 16)); };
    for (uint32_t t : kept) put(t);
    for (uint32_t t : asc) put(t);
    rec.insert(rec.end(), raw_units.begin(), raw_units.end());
    if (rec.size() & 1) rec.push_back(0);
    return true;
}

bool build_list_record(lmat_ctx* c, const std::vector<uint16_t>& raw, std::vector<uint16_t>& rec) {
    const HostTaxonomy& T = c->tax;
    std::vector<uint32_t> seq;
    for (size_t i = 0; i < raw.size(); ++i) {
        const uint32_t tid = T.conv[raw[i]];
        if (tid == 0) {  // TaxNodeStat.hpp:235-238: "bad taxid" assert
            set_err(c, LMAT_E_TAXONOMY, "bad taxid: 16-bit id " + std::to_string(raw[i]) + " has no 32-bit mapping");
            return false;
        }
        seq.push_back(tid);
    }
    if (!T.wide) return build_list_record_core(c, seq, raw, raw.size(), false, rec);
    // a 16-bit map under a taxonomy whose closure is wide: the record's ids are pairs, and so is the stored list kept behind it
    std::vector<uint16_t> pairs;
    for (uint32_t t : seq) { pairs.push_back((uint16_t)t); pairs.push_back((uint16_t)(t >> 16)); }
    return build_list_record_core(c, seq, pairs, seq.size(), true, rec);
}

bool build_list_record_wide(lmat_ctx* c, const std::vector<uint16_t>& raw_pairs, std::vector<uint16_t>& rec) {
    std::vector<uint32_t> seq;
    for (size_t i = 0; i + 1 < raw_pairs.size(); i += 2) seq.push_back((uint32_t)raw_pairs[i] | ((uint32_t)raw_pairs[i + 1] << 16));
    return build_list_record_core(c, seq, raw_pairs, seq.size(), true, rec);
}

}  // namespace lmat
