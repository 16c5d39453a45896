"""ctypes binding of the CPU oracle (oracle/liblmat_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg as the checker; never by the lmat_amd package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.environ.get("LMAT_ORACLE_LIB") or os.path.join(_HERE, "liblmat_oracle.so")  # (LMAT_ORACLE_LIB: the sanitizer build, scripts/sanitize_cpu.sh)


def build():
    subprocess.check_call(["make", "-C", _HERE, "lmat_oracle", "liblmat_oracle.so"], stdout=subprocess.DEVNULL)


def _load():
    if not os.path.exists(LIB):
        build()
    L = C.CDLL(LIB)
    vp, cp, u64, i32 = C.c_void_p, C.c_char_p, C.c_uint64, C.c_int
    L.orc_create.restype = vp
    L.orc_create.argtypes = [cp, cp, cp, cp, cp]
    L.orc_destroy.argtypes = [vp]
    L.orc_error.restype = cp
    L.orc_error.argtypes = [vp]
    L.orc_add_taxhisto.argtypes = [vp, cp]
    L.orc_load_null_models.argtypes = [vp, cp]
    L.orc_set_label_modes.argtypes = [vp, i32, i32, cp]
    L.orc_set_build_options.argtypes = [vp, i32, cp, cp, cp]
    L.orc_db_k.argtypes = [vp]
    L.orc_set_k.argtypes = [vp, i32]
    L.orc_db_size.restype = u64
    L.orc_db_size.argtypes = [vp]
    L.orc_add_list32.argtypes = [vp, u64, vp, i32]
    L.orc_lookup.argtypes = [vp, u64, vp, i32]
    L.orc_lookup_rt.argtypes = [vp, u64, vp, i32]
    L.orc_path_to_root.argtypes = [vp, C.c_uint32, vp, i32]
    L.orc_set_options.argtypes = [vp, C.c_float, C.c_float, i32, i32, C.c_float, i32, i32, i32, i32]
    L.orc_extract.argtypes = [cp, i32, i32, vp, vp, i32, C.POINTER(i32), C.POINTER(i32)]
    L.orc_classify.restype = C.c_long
    L.orc_classify.argtypes = [vp, vp, vp, C.c_long, C.c_long, i32, vp, vp, vp, i32, C.POINTER(i32), vp]
    L.orc_add_lists32.argtypes = [vp, vp, vp, vp, C.c_uint32, u64]
    L.orc_classify_mt.restype = C.c_long
    L.orc_classify_mt.argtypes = [vp, vp, vp, C.c_long, i32, i32]
    L.orc_text.restype = vp
    L.orc_text.argtypes = [vp]
    L.orc_summaries_from_calls.restype = C.c_long
    L.orc_summaries_from_calls.argtypes = [vp, vp, vp, vp, C.c_long, i32, i32, vp, C.c_long, vp, C.c_long]
    L.orc_score_stats.restype = None
    L.orc_score_stats.argtypes = [vp, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.orc_rkmer_trace.restype = cp
    L.orc_rkmer_trace.argtypes = [vp, vp, vp, u64, i32, i32]
    L.orc_rand_label.restype = i32
    L.orc_rand_label.argtypes = [vp, vp, vp, u64, i32, vp, C.c_uint32, vp, vp, vp, C.c_uint32]
    L.orc_run_file.restype = C.c_long
    L.orc_gene_create.restype = vp
    L.orc_gene_destroy.argtypes = [vp]
    L.orc_gene_add.argtypes = [vp, cp]
    L.orc_gene_k.argtypes = [vp]
    L.orc_gene_size.argtypes = [vp]
    L.orc_gene_size.restype = u64
    L.orc_gene_lookup.argtypes = [vp, u64, vp, i32]
    L.orc_gene_label.argtypes = [vp, vp, vp, C.c_long, i32, vp, vp, vp, vp, vp]
    L.orc_gene_run.argtypes = [vp, cp, cp, cp, C.c_float, i32, C.c_float]
    L.orc_gene_replay.argtypes = [cp, cp, cp, cp, C.c_float, i32, C.c_float]
    L.orc_replay_decision.argtypes = [vp, vp, vp, i32, C.c_float, vp, vp, vp]
    L.orc_run_file.argtypes = [vp, cp, i32, cp, vp, C.c_long, vp, C.c_long]
    return L


def gene_replay(list_fn, gl_list_fn, ofbase, genefile, min_score=0.0, min_kmer=0, min_tax_score=0.0):
    """gene_oracle::replay_files: gene_label's main() with every read's vote taken from an earlier run's output files."""
    if _load().orc_gene_replay(list_fn.encode(), gl_list_fn.encode(), ofbase.encode(), genefile.encode(), min_score, min_kmer, min_tax_score) != 0:
        raise RuntimeError("gene oracle: replay failed")


class GeneOracle:
    """gene_oracle.hpp: the CPU restatement of src/gene_label.cpp (test infrastructure)."""

    def __init__(self, files):
        self.L = _load()
        self.h = self.L.orc_gene_create()
        for f in ([files] if isinstance(files, str) else files):
            if self.L.orc_gene_add(self.h, f.encode()) != 0:
                raise RuntimeError("gene oracle: cannot read " + f)

    @property
    def k(self):
        return self.L.orc_gene_k(self.h)

    def __len__(self):
        return int(self.L.orc_gene_size(self.h))

    def lookup(self, kmer, cap=4096):
        out = np.zeros(cap, dtype=np.uint32)
        n = self.L.orc_gene_lookup(self.h, int(kmer), out.ctypes.data_as(C.c_void_p), cap)
        return out[:n].tolist()

    def label(self, blob, off, k):
        """-> structured array (any, gid, top, cnt, score) per read."""
        n = off.size - 1
        any_ = np.zeros(n, dtype=np.uint8)
        gid, top, cnt = (np.zeros(n, dtype=np.uint32) for _ in range(3))
        score = np.zeros(n, dtype=np.float32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self.L.orc_gene_label(self.h, p(blob), p(off), n, k, p(any_), p(gid), p(top), p(cnt), p(score))
        return any_, gid, top, cnt, score

    def run(self, list_fn, ofbase, genefile, min_score=0.0, min_kmer=0, min_tax_score=0.0):
        if self.L.orc_gene_run(self.h, list_fn.encode(), ofbase.encode(), genefile.encode(), min_score, min_kmer, min_tax_score) != 0:
            raise RuntimeError("gene oracle: run failed")

    def close(self):
        if self.h:
            self.L.orc_gene_destroy(self.h)
            self.h = None


class Oracle:
    def __init__(self, tree, depth, rank, idmap, plasmids=None):
        self.L = _load()
        e = lambda s: s.encode() if s else b""
        self.h = self.L.orc_create(e(tree), e(depth), e(rank), e(idmap), e(plasmids))
        if not self.h:
            raise RuntimeError("oracle: cannot load taxonomy files")

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def set_build_options(self, tid_cutoff=0, rank_map=None, human=None, adaptors=None):
        e = lambda s: s.encode() if s else b""
        if self.L.orc_set_build_options(self.h, tid_cutoff, e(rank_map), e(human), e(adaptors)) != 0:
            raise RuntimeError("oracle: cannot read build-option files")

    def set_label_modes(self, permissive=False, tid_cutoff=0, rank_map=None):
        if self.L.orc_set_label_modes(self.h, int(permissive), tid_cutoff, (rank_map or "").encode()) != 0:
            raise RuntimeError("oracle: cannot read rank map")

    def score_stats(self, scores):
        s = np.ascontiguousarray(scores, dtype=np.float32)
        a, b = C.c_float(0), C.c_float(0)
        self.L.orc_score_stats(s.ctypes.data, s.size, C.byref(a), C.byref(b))
        return a.value, b.value

    def rkmer_trace(self, blob, off, k, permissive=False):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        return self.L.orc_rkmer_trace(self.h, blob.ctypes.data, off.ctypes.data, off.size - 1, k, int(permissive)).decode()

    def rand_label(self, blob, off, k, gc_bucket, nb=10, cap=70000):
        """rand_read_label over these reads -> {taxid: ([max label_prob per bucket], [hit count per bucket])}"""
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        gb = np.ascontiguousarray(gc_bucket, dtype=np.uint8)
        tid = np.zeros(cap, dtype=np.uint32)
        mx = np.zeros((cap, nb), dtype=np.float32)
        ct = np.zeros((cap, nb), dtype=np.int32)
        n = self.L.orc_rand_label(self.h, blob.ctypes.data, off.ctypes.data, off.size - 1, k, gb.ctypes.data, nb, tid.ctypes.data,
                                  mx.ctypes.data, ct.ctypes.data, cap)
        assert 0 <= n <= cap
        return {int(tid[i]): (mx[i].copy(), ct[i].copy()) for i in range(n)}

    def load_null_models(self, list_fn):
        if self.L.orc_load_null_models(self.h, list_fn.encode()) != 0:
            raise RuntimeError("oracle: cannot read null-model list")

    def add_taxhisto(self, fn):
        if self.L.orc_add_taxhisto(self.h, fn.encode()) != 0:
            raise RuntimeError("oracle: " + self.L.orc_error(self.h).decode())

    def add_list32(self, kmer, tids):
        a = np.ascontiguousarray(tids, dtype=np.uint32)
        if self.L.orc_add_list32(self.h, int(kmer), a.ctypes.data, a.size) != 0:
            raise RuntimeError("oracle: " + self.L.orc_error(self.h).decode())

    def add_lists32(self, kmers, counts, tids):
        km = np.ascontiguousarray(kmers, dtype=np.uint64)
        ct = np.ascontiguousarray(counts, dtype=np.uint32)
        td = np.ascontiguousarray(tids, dtype=np.uint32)
        if self.L.orc_add_lists32(self.h, km.ctypes.data, ct.ctypes.data, td.ctypes.data, td.shape[1], km.size) != 0:
            raise RuntimeError("oracle: " + self.L.orc_error(self.h).decode())

    def classify_mt(self, blob, off, k, nthreads):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        return self.L.orc_classify_mt(self.h, blob.ctypes.data, off.ctypes.data, off.size - 1, k, nthreads)

    @property
    def k(self):
        return self.L.orc_db_k(self.h)

    def set_k(self, k):
        self.L.orc_set_k(self.h, k)

    def lookup(self, kmer, cap=4096):
        out = np.zeros(cap, dtype=np.uint32)
        n = self.L.orc_lookup(self.h, int(kmer), out.ctypes.data, cap)
        return n, out[:max(n, 0)].copy()

    def lookup_rt(self, kmer, cap=4096):
        out = np.zeros(cap, dtype=np.uint32)
        n = self.L.orc_lookup_rt(self.h, int(kmer), out.ctypes.data, cap)
        return n, out[:max(n, 0)].copy()

    def path_to_root(self, tid, cap=256):
        out = np.zeros(cap, dtype=np.uint32)
        n = self.L.orc_path_to_root(self.h, int(tid), out.ctypes.data, cap)
        return out[:n].tolist()

    def set_options(self, sdiff=1.0, hbias=0.0, prn_all=1, screen_phix=1, min_score=0.0, min_kmer=30, min_fnd_kmer=1,
                    prn_read=1, fastq=0):
        self.L.orc_set_options(self.h, sdiff, hbias, prn_all, screen_phix, min_score, min_kmer, min_fnd_kmer, prn_read,
                               fastq)

    def replay_decision(self, tids, scores, stdev):
        """TCmp sort + findReadLabelVer2 on a given candidate set -> (call taxid, call score, match code)."""
        t = np.ascontiguousarray(tids, dtype=np.uint32)
        s = np.ascontiguousarray(scores, dtype=np.float32)
        ct, cs, m = C.c_uint32(0), C.c_float(0), C.c_int(0)
        self.L.orc_replay_decision(self.h, t.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p), len(t), float(stdev),
                                   C.byref(ct), C.byref(cs), C.byref(m))
        return int(ct.value), float(cs.value), int(m.value)

    def extract(self, read: bytes, k=20):
        cap = max(len(read), 1)
        km = np.zeros(cap, dtype=np.uint64)
        ps = np.zeros(cap, dtype=np.int32)
        v, b = C.c_int(0), C.c_int(0)
        n = self.L.orc_extract(read, len(read), k, km.ctypes.data, ps.ctypes.data, cap, C.byref(v), C.byref(b))
        return km[:n].copy(), ps[:n].copy(), v.value, b.value

    def classify(self, blob, off, k, first_index=0, tally_cap=70000):
        """-> (.out text, {tid: (count, score)}, [ReadTooShort, NoDbHits, LowScore])"""
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        tid = np.zeros(tally_cap, dtype=np.uint32)
        cnt = np.zeros(tally_cap, dtype=np.int32)
        sc = np.zeros(tally_cap, dtype=np.float32)
        nt = C.c_int(0)
        nm = np.zeros(3, dtype=np.int32)
        ln = self.L.orc_classify(self.h, blob.ctypes.data, off.ctypes.data, off.size - 1, first_index, k, tid.ctypes.data,
                                 cnt.ctypes.data, sc.ctypes.data, tally_cap, C.byref(nt), nm.ctypes.data)
        text = C.string_at(self.L.orc_text(self.h), ln).decode()
        n = nt.value
        return text, {int(t): (int(c), float(s)) for t, c, s in zip(tid[:n], cnt[:n], sc[:n])}, nm.tolist()

    def summaries_from_calls(self, tids, scores, nomatch, extra_short=0, extra_nodb=0):
        t = np.ascontiguousarray(tids, dtype=np.uint32)
        s = np.ascontiguousarray(scores, dtype=np.float32)
        m = np.ascontiguousarray(nomatch, dtype=np.int32)
        fs = C.create_string_buffer(1 << 20)
        ns = C.create_string_buffer(1 << 12)
        self.L.orc_summaries_from_calls(self.h, t.ctypes.data, s.ctypes.data, m.ctypes.data, t.size, extra_short, extra_nodb,
                                        fs, len(fs), ns, len(ns))
        return fs.value.decode(), ns.value.decode()

    def run_file(self, query, k, rank_ids=None):
        fs = C.create_string_buffer(1 << 22)
        ns = C.create_string_buffer(1 << 12)
        ln = self.L.orc_run_file(self.h, query.encode(), k, (rank_ids or "").encode(), fs, len(fs), ns, len(ns))
        if ln < 0:
            raise RuntimeError("oracle: " + self.L.orc_error(self.h).decode())
        return C.string_at(self.L.orc_text(self.h), ln).decode(), fs.value.decode(), ns.value.decode()
