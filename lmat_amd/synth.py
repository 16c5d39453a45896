"""Deterministic synthetic LMAT inputs: taxonomy, genomes, tax_histo DB, reads.

Writes the file formats the reference's read_label consumes:
  * taxonomy .dat  (src/kmerdb/TaxTree.hpp:24-57, TaxNode.hpp:131-147)
  * depth file / rank file / 32->16 id map (src/read_label.cpp:1560-1602)
  * tax_histo binary (src/tax_histo.cpp:257-281; header KmerFileMetaData.cpp:16-31)
  * FASTA / FASTQ queries
The DB contents follow what tax_histo would emit: for every canonical k-mer of
every leaf genome, the owning taxids plus every node up to and including their
LCA (src/kmerdb/TaxTree.hpp:160-260).

Seeds follow SURVEY.md 8(d): taxonomy 1001, DB 2002, reads 3003.
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass, field

import numpy as np

RANKS = ["no_rank", "superkingdom", "phylum", "family", "genus", "species", "strain"]
SANITY = 0xFFFFFFFFFFFFFFFF


@dataclass
class Taxonomy:
    ids: list = field(default_factory=list)          # 32-bit ids
    parent: dict = field(default_factory=dict)
    depth: dict = field(default_factory=dict)
    rank: dict = field(default_factory=dict)
    name: dict = field(default_factory=dict)
    children: dict = field(default_factory=dict)
    leaves: list = field(default_factory=list)       # ids that own a genome
    species_of: dict = field(default_factory=dict)   # strain -> species
    genus_of: dict = field(default_factory=dict)     # species -> genus
    id16: dict = field(default_factory=dict)         # 32 -> 16

    def add(self, tid, par, rank, name):
        self.ids.append(tid)
        self.parent[tid] = par
        self.rank[tid] = rank
        self.name[tid] = name
        self.depth[tid] = 0 if tid == par else self.depth[par] + 1
        self.children.setdefault(tid, [])
        if tid != par:
            self.children[par].append(tid)

    def path(self, tid):
        out = []
        while self.parent[tid] != tid:
            tid = self.parent[tid]
            out.append(tid)
        return out


def make_taxonomy(branching=(3, 4, 4, 4, 4, 3), specials=True) -> Taxonomy:
    """Root(1) -> superkingdoms -> phyla -> families -> genera -> species -> strains.

    32-bit ids are sparse pseudo-NCBI (1000 + 7*dense).  With specials=True the
    hard-coded taxids of include/tid_checks.hpp and read_label.cpp:82-104 are added
    so those code paths fire: human 9606 + 63221, PhiX 374840, synthetic construct
    32630, a plasmid id in [1e7, 1.1e7), the ignored ids 12721 and 20999999.
    """
    t = Taxonomy()
    t.add(1, 1, "no_rank", "root")
    dense = [0]

    def new_id():
        dense[0] += 1
        return 1000 + 7 * dense[0]

    level = [1]
    for lv, b in enumerate(branching):
        nxt = []
        for p in level:
            for j in range(b):
                tid = new_id()
                t.add(tid, p, RANKS[lv + 1], f"{RANKS[lv + 1]}_{tid}")
                nxt.append(tid)
                if RANKS[lv + 1] == "strain":
                    t.species_of[tid] = p
                if RANKS[lv + 1] == "species":
                    t.genus_of[tid] = p
        level = nxt
    t.leaves = list(level)
    if specials:
        sk = t.children[1]
        euk, vir = sk[0], sk[-1]
        # human clade: a genus-like node holding 9606 and 63221 (both "species")
        hg = new_id()
        t.add(hg, t.children[euk][0], "genus", "Homo")
        t.add(9606, hg, "species", "Homo sapiens")
        t.add(63221, hg, "species", "Homo sapiens neanderthalensis")
        t.genus_of[9606] = hg
        t.genus_of[63221] = hg
        # PhiX + synthetic construct
        pg = new_id()
        t.add(pg, t.children[vir][0], "genus", "Microvirus-like")
        t.add(374840, pg, "species", "Enterobacteria phage phiX174 sensu lato")
        t.add(32630, t.children[vir][0], "species", "synthetic construct")
        t.genus_of[374840] = pg
        # a plasmid hanging off the first species, and the two ignored ids
        sp0 = t.parent[t.leaves[0]]
        t.add(10000001, sp0, "no_rank", "plasmid pSYN1")
        t.add(12721, t.children[vir][0], "species", "HIV-like ignored")
        t.add(20999999, 1, "no_rank", "ignored marker")
        t.leaves += [9606, 63221, 374840, 10000001, 12721]
    # 16-bit map: root -> 1, the rest dense from 2 in ascending 32-bit order
    # (as bin/Tid16_getMapping.py:83-91 does)
    t.id16[1] = 1
    nxt16 = 2
    for tid in sorted(t.ids):
        if tid == 1:
            continue
        t.id16[tid] = nxt16
        nxt16 += 1
    return t


def write_aux_files(outdir: str, tax: Taxonomy) -> dict:
    os.makedirs(outdir, exist_ok=True)
    p = {k: os.path.join(outdir, v) for k, v in dict(
        tree="tax.dat", depth="depth.dat", rank="rank.txt", idmap="map32to16.txt", names="rank_names.txt").items()}
    lines = ["# synthetic taxonomy", "# id nchild children... parent / name", str(len(tax.ids))]
    for tid in tax.ids:
        ch = tax.children[tid]
        lines.append(" ".join(str(x) for x in [tid, len(ch)] + ch + [tax.parent[tid]]))
        lines.append(tax.name[tid])
    with open(p["tree"], "w") as f:
        f.write("\n".join(lines))  # last name line must have NO trailing newline
    with open(p["depth"], "w") as f:
        for tid in tax.ids:
            f.write(f"{tid} {tax.depth[tid]}\n")
    with open(p["rank"], "w") as f:
        for tid in tax.ids:
            f.write(f"{tid} {tax.rank[tid]}\n")
    with open(p["idmap"], "w") as f:
        for tid in sorted(tax.ids):
            f.write(f"{tid} {tax.id16[tid]}\n")
    with open(p["names"], "w") as f:
        for tid in tax.ids:
            f.write(f"taxid={tid},rank={tax.rank[tid]}\t{tax.rank[tid]},{tax.name[tid]}\n")
    return p


# --------------------------------------------------------------------------- genomes
def make_genomes(tax: Taxonomy, G: int, seed: int = 2002, strain_sub=0.01, genus_block=0.10) -> dict:
    """leaf id -> uint8 array of 2-bit codes (A0 C1 G2 T3)."""
    rng = np.random.default_rng(seed)
    genus_anc, species_anc, genomes = {}, {}, {}
    blk = int(G * genus_block)

    def species_genome(sp):
        if sp not in species_anc:
            g = rng.integers(0, 4, size=G, dtype=np.uint8)
            ge = tax.genus_of.get(sp)
            if ge is not None and blk > 0:
                if ge not in genus_anc:
                    genus_anc[ge] = rng.integers(0, 4, size=blk, dtype=np.uint8)
                g[:blk] = genus_anc[ge]
            species_anc[sp] = g
        return species_anc[sp]

    def mutate(g, rate):
        g = g.copy()
        m = rng.random(g.size) < rate
        g[m] = (g[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) & 3
        return g

    for leaf in tax.leaves:
        if leaf in tax.species_of:
            genomes[leaf] = mutate(species_genome(tax.species_of[leaf]), strain_sub)
        elif tax.rank[leaf] == "species":
            genomes[leaf] = mutate(species_genome(leaf), 0.0)
        else:  # plasmid: a copy of part of the first strain of its parent species plus own sequence
            sib = [c for c in tax.children[tax.parent[leaf]] if c in genomes]
            own = rng.integers(0, 4, size=G // 2, dtype=np.uint8)
            genomes[leaf] = np.concatenate([genomes[sib[0]][: G // 4], own]) if sib else own
    # the second human id shares most of the first one's genome
    if 9606 in genomes and 63221 in genomes:
        genomes[63221] = mutate(genomes[9606], 0.02)
    return genomes


def kmers_of(codes: np.ndarray, k: int):
    """Canonical k-mers of a code array: min(forward, reverse complement) as in
    src/read_label.cpp:992-1009 (forward: first base in the high bits)."""
    n = codes.size - k + 1
    if n <= 0:
        return np.zeros(0, dtype=np.uint64)
    win = np.lib.stride_tricks.sliding_window_view(codes.astype(np.uint64), k)
    shifts_f = (2 * np.arange(k - 1, -1, -1)).astype(np.uint64)
    shifts_r = (2 * np.arange(0, k)).astype(np.uint64)
    fwd = np.bitwise_or.reduce(win << shifts_f, axis=1)
    rev = np.bitwise_or.reduce((np.uint64(3) - win) << shifts_r, axis=1)
    return np.minimum(fwd, rev)


def lca_closure(tax: Taxonomy, owners: tuple) -> list:
    """What tax_histo writes for a k-mer owned by `owners` (TaxTree.hpp:160-260):
    a single owner -> itself; otherwise owners + every node on the way up to the LCA."""
    if len(owners) == 1:
        return [owners[0]]
    paths = [[o] + tax.path(o) for o in owners]
    common = set(paths[0])
    for p in paths[1:]:
        common &= set(p)
    lca = max(common, key=lambda x: tax.depth[x])
    out = set()
    for p in paths:
        for x in p:
            out.add(x)
            if x == lca:
                break
    return sorted(out)


def build_kmer_table(tax: Taxonomy, genomes: dict, k: int = 20, extra_lists: bool = True):
    """-> (sorted uint64 k-mers, list of taxid lists in 'file order')."""
    ks, os_ = [], []
    leaf_index = {leaf: i for i, leaf in enumerate(tax.leaves)}
    for leaf, g in genomes.items():
        km = np.unique(kmers_of(g, k))
        ks.append(km)
        os_.append(np.full(km.size, leaf_index[leaf], dtype=np.int32))
    ks = np.concatenate(ks)
    os_ = np.concatenate(os_)
    order = np.lexsort((os_, ks))
    ks, os_ = ks[order], os_[order]
    uniq, start = np.unique(ks, return_index=True)
    end = np.append(start[1:], ks.size)
    cache, lists = {}, []
    for s, e in zip(start, end):
        key = tuple(os_[s:e].tolist())
        if key not in cache:
            owners = tuple(tax.leaves[i] for i in key)
            lst = lca_closure(tax, owners)
            # tax_histo writes in unordered_map iteration order: emulate "arbitrary but
            # deterministic" with a fixed scramble so nothing downstream relies on sorting
            lst = sorted(lst, key=lambda x: (x * 2654435761) & 0xFFFFFFFF)
            cache[key] = lst
        lists.append(cache[key])
    if extra_lists and uniq.size > 64:
        # salt a few lists with the ids the reference filters out (read_label.cpp:1033-1038)
        for j, extra in enumerate([20999999, 12721]):
            for idx in range(7 + j, uniq.size, max(uniq.size // 40, 1)):
                if extra in tax.parent and len(lists[idx]) >= 1 and extra not in lists[idx]:
                    lists[idx] = lists[idx] + [extra]
    return uniq, lists


def write_taxhisto(path: str, kmers: np.ndarray, lists: list, k: int = 20):
    with open(path, "wb") as f:
        f.write(struct.pack("<IQQIcI", 29, len(kmers), SANITY, 999, b"N", k))
        for i, (km, lst) in enumerate(zip(kmers.tolist(), lists)):
            f.write(struct.pack("<QH", km, len(lst)))
            f.write(struct.pack("<%dI" % len(lst), *lst))
            if (i + 1) % 1500 == 0:
                f.write(struct.pack("<Q", SANITY))


# --------------------------------------------------------------------------- reads
_ASCII = np.frombuffer(b"ACGT", dtype=np.uint8)


def _revcomp(codes):
    return (3 - codes)[::-1]


def make_reads(tax: Taxonomy, genomes: dict, n: int, L=150, seed: int = 3003, err=0.01,
               frac_random=0.10, frac_n=0.01, frac_lowc=0.01, frac_short=0.0, lower_frac=0.0):
    """-> list of (header, sequence).  L may be an int or a sequence of lengths."""
    rng = np.random.default_rng(seed)
    leaves = [l for l in tax.leaves if genomes[l].size >= 20]
    Ls = [L] if np.isscalar(L) else list(L)
    out = []
    for i in range(n):
        ln = int(Ls[rng.integers(0, len(Ls))])
        u = rng.random()
        if u < frac_random:
            codes = rng.integers(0, 4, size=ln, dtype=np.uint8)
        elif u < frac_random + frac_lowc:
            unit = rng.integers(0, 4, size=25, dtype=np.uint8)
            codes = np.tile(unit, ln // 25 + 1)[:ln]
        else:
            leaf = leaves[rng.integers(0, len(leaves))]
            g = genomes[leaf]
            ln_eff = min(ln, g.size)
            off = rng.integers(0, g.size - ln_eff + 1)
            codes = g[off:off + ln_eff].copy()
            if rng.random() < 0.5:
                codes = _revcomp(codes)
            m = rng.random(codes.size) < err
            codes[m] = (codes[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) & 3
        seq = _ASCII[codes].copy()
        v = rng.random()
        if v < frac_n:
            seq[rng.integers(0, seq.size)] = ord("N")
        elif v < frac_n * 1.5:
            for _ in range(4):
                seq[rng.integers(0, seq.size)] = ord("N")
        if rng.random() < frac_short:
            seq = seq[: int(rng.integers(1, 60))]
        s = seq.tobytes().decode()
        if rng.random() < lower_frac:
            s = s.lower()
        out.append((f"r{i}", s))
    return out


def write_fasta(path, reads, width=0):
    with open(path, "w") as f:
        for h, s in reads:
            f.write(f">{h}\n")
            if width and len(s) > width:
                for j in range(0, len(s), width):
                    f.write(s[j:j + width] + "\n")
            else:
                f.write(s + "\n")


def write_fastq(path, reads):
    with open(path, "w") as f:
        for h, s in reads:
            f.write(f"@{h}\n{s}\n+\n{'I' * len(s)}\n")


NULL_CLASS = {"no_rank": "depth=0", "superkingdom": "kingdom", "phylum": "phylum", "family": "family", "genus": "genus",
              "species": "species", "strain": "no_rank"}


def write_null_models(outdir: str, tax: Taxonomy, kmer_counts=(31, 56, 81, 131, 181, 281), num_bins=11, seed=4004) -> str:
    """Null-model list + gz tables in the grammar loadRandHits parses (src/read_label.cpp:553-671):
    list lines `<kmer_count> <file relative to $LMAT_DIR>`; each gz file: first line num_bins, then per taxid
    `taxid <class>-<anything> (num_obs max_val kmer_cnt) x num_bins`.  Exercises: zero observations with a large
    genome (-> 0.5), zero observations filled from neighbouring bins, class strings starting with no_ (-> genus),
    and one class the rank table does not know."""
    import gzip
    rng = np.random.default_rng(seed)
    os.makedirs(outdir, exist_ok=True)
    lst = os.path.join(outdir, "null_lst.txt")
    with open(lst, "w") as lf:
        for kc in kmer_counts:
            name = f"null.{kc}.rand_lst.gz"
            lf.write(f"{kc} {name}\n")
            with gzip.open(os.path.join(outdir, name), "wt") as g:
                g.write(f"{num_bins}\n")
                for i, tid in enumerate(tax.ids):
                    cls = NULL_CLASS.get(tax.rank[tid], "genus")
                    if i % 37 == 5:
                        cls = "tribe"  # not in gRank2num: rank 0
                    parts = [str(tid), f"{cls}-x{i}"]
                    big = i % 11 == 3
                    for b in range(num_bins):
                        u = rng.random()
                        if u < 0.15:
                            parts += ["0", "0", str(250000 if big else 4000)]
                        else:
                            parts += [str(int(rng.integers(1, 500))), f"{rng.random() * 0.3 * (1 + 0.1 * tax.depth[tid]):.6g}", "50000"]
                    g.write(" ".join(parts) + "\n")
    return lst


def generate_dataset(outdir: str, branching=(2, 2, 2, 2, 3, 3), G=600, n_reads=400, L=150, k=20,
                     specials=True, seeds=(1001, 2002, 3003), **read_kw) -> dict:
    tax = make_taxonomy(branching, specials)
    paths = write_aux_files(outdir, tax)
    genomes = make_genomes(tax, G, seeds[1])
    kmers, lists = build_kmer_table(tax, genomes, k)
    paths["db"] = os.path.join(outdir, "th.bin")
    write_taxhisto(paths["db"], kmers, lists, k)
    reads = make_reads(tax, genomes, n_reads, L, seeds[2], **read_kw)
    paths["fasta"] = os.path.join(outdir, "reads.fa")
    write_fasta(paths["fasta"], reads)
    paths["fastq"] = os.path.join(outdir, "reads.fq")
    write_fastq(paths["fastq"], reads)
    paths["n_kmers"] = int(kmers.size)
    paths["n_reads"] = len(reads)
    paths["k"] = k
    return paths


if __name__ == "__main__":
    import argparse, json

    ap = argparse.ArgumentParser()
    ap.add_argument("outdir")
    ap.add_argument("--G", type=int, default=600)
    ap.add_argument("--reads", type=int, default=400)
    ap.add_argument("--branching", default="2,2,2,2,3,3")
    a = ap.parse_args()
    info = generate_dataset(a.outdir, tuple(int(x) for x in a.branching.split(",")), a.G, a.reads,
                            frac_short=0.03, lower_frac=0.05)
    print(json.dumps(info))
