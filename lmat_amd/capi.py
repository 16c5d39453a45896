"""ctypes mirror of include/lmat_hip.h (same names, same argument order, same error codes)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LMAT_LIB") or os.path.join(_HERE, "liblmat_hip.so")  # LMAT_LIB: A/B builds when profiling


class LmatError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"lmat error {code}: {msg}")
        self.code = code


class Params(C.Structure):
    """lmat_params: ScoreOptions + proc_line thresholds (src/read_label.cpp:487-497,1336-1337)."""
    _fields_ = [("sdiff", C.c_float), ("hbias", C.c_float), ("min_score", C.c_float), ("min_kmer", C.c_int32),
                ("min_fnd_kmer", C.c_int32), ("prn_all", C.c_int32), ("screen_phix", C.c_int32)]

    @classmethod
    def run_rl(cls, prn_all=1):
        """The flags bin/run_rl.sh passes with --nullm=no: -x 0 -j 30 -l 0 -b 1.0 -p."""
        return cls(1.0, 0.0, 0.0, 30, 1, prn_all, 1)


READ_RESULT_DTYPE = np.dtype([("status", "u1"), ("match_type", "u1"), ("cand_kmer_cnt", "<u2"), ("valid_kmers", "<i4"),
                              ("read_len", "<i4"), ("log_avg", "<f4"), ("stdev", "<f4"), ("call_tid", "<u4"),
                              ("call_score", "<f4"), ("cand_off", "<u4"), ("n_cand", "<u4"), ("bin_sel", "<i4")])
CAND_DTYPE = np.dtype([("tid", "<u4"), ("score", "<f4")])

_lib = None


def load_library(path: str | None = None):
    """Loads liblmat_hip.so.  There is no fallback: a missing library is an error."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise ImportError(f"{p} not found: build it with `make -C lmat_amd/csrc` (hipcc, gfx950); "
                          "lmat_amd has no CPU implementation")
    lib = C.CDLL(p)
    vp, u64, u32, i32, cp = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_char_p
    P = C.POINTER
    sig = {
        "lmat_device_count": (i32, []),
        "lmat_ctx_create": (i32, [i32, P(Params), P(vp)]),
        "lmat_ctx_destroy": (None, [vp]),
        "lmat_last_error": (cp, [vp]),
        "lmat_set_params": (i32, [vp, P(Params)]),
        "lmat_taxonomy_load_files": (i32, [vp, cp, cp, cp, cp, cp]),
        "lmat_db_begin": (i32, [vp, i32, u64, u64]),
        "lmat_genedb_begin": (i32, [vp, i32, u64, u64]),
        "lmat_db_add_taxhisto": (i32, [vp, cp]),
        "lmat_db_finalize": (i32, [vp]),
        "lmat_db_set_build_options": (i32, [vp, i32, cp, cp, cp, u32]),
        "lmat_db_save_image": (i32, [vp, cp]),
        "lmat_db_load_image": (i32, [vp, cp, u64]),
        "lmat_ingest_create": (i32, [i32, cp, P(vp)]),
        "lmat_ingest_idmap_from_tree": (i32, [vp, cp]),
        "lmat_ingest_destroy": (None, [vp]),
        "lmat_ingest_error": (cp, [vp]),
        "lmat_ingest_set_options": (i32, [vp, i32, cp, cp, cp, u32]),
        "lmat_ingest_add_taxhisto": (i32, [vp, cp]),
        "lmat_ingest_save_image": (i32, [vp, cp]),
        "lmat_ingest_load_image": (i32, [cp, P(vp)]),
        "lmat_ingest_size": (u64, [vp]),
        "lmat_ingest_kmer_length": (i32, [vp]),
        "lmat_ingest_lookup": (i32, [vp, u64, vp, i32]),
        "lmat_db_from_ingest": (i32, [vp, vp, u64]),
        "lmat_set_label_modes": (i32, [vp, i32, i32, cp]),
        "lmat_rand_mode": (i32, [vp, i32]),
        "lmat_rand_reset": (i32, [vp, u32]),
        "lmat_rand_label": (i32, [vp, vp, u64, u64, vp]),
        "lmat_rand_get": (i32, [vp, vp, vp, vp, u32, P(u32)]),
        "lmat_nullmodel_load": (i32, [vp, cp]),
        "lmat_nullmodel_clear": (i32, [vp]),
        "lmat_db_kmer_length": (i32, [vp]),
        "lmat_db_size": (u64, [vp]),
        "lmat_db_list_count": (u64, [vp]),
        "lmat_db_arena_bytes": (u64, [vp]),
        "lmat_db_table_bytes": (u64, [vp]),
        "lmat_db_lookup": (i32, [vp, vp, u64, vp, vp, u32]),
        "lmat_synth_taxonomy": (i32, [vp, vp]),
        "lmat_synth_db_build": (i32, [vp, i32, u64, u64, u64]),
        "lmat_synth_db_build2": (i32, [vp, i32, u64, u64, u64, u32, u32]),
        "lmat_synth_db_build3": (i32, [vp, i32, u64, u64, u64, u32, u32, vp]),
        "lmat_reads_upload": (i32, [vp, vp, vp, u64, P(vp)]),
        "lmat_reads_synth": (i32, [vp, u64, vp, u32, u64, P(vp)]),
        "lmat_reads_download_ascii": (i32, [vp, vp, u64, u64, vp, vp]),
        "lmat_reads_count": (u64, [vp]),
        "lmat_reads_device_bytes": (u64, [vp]),
        "lmat_reads_free": (None, [vp, vp]),
        "lmat_classify": (i32, [vp, vp, u64, u64, vp, vp, u64, P(u64)]),
        "lmat_classify_async": (i32, [vp, vp, u64, u64]),
        "lmat_classify_async_cands": (i32, [vp, vp, u64, u64, u64]),
        "lmat_comm_available": (i32, []),
        "lmat_sync": (i32, [vp, P(C.c_float), P(u64)]),
        "lmat_last_timing": (i32, [vp, P(C.c_float), P(C.c_float), P(u64)]),
        "lmat_results_fetch": (i32, [vp, u64, u64, vp]),
        "lmat_counts_reset": (i32, [vp]),
        "lmat_counts_layout": (i32, [vp, P(u32), P(u64)]),
        "lmat_counts_device_ptr": (vp, [vp]),
        "lmat_counts_get": (i32, [vp, vp, vp, vp, u32, P(u32), vp]),
        "lmat_gather_bench": (i32, [vp, u64, u64, P(C.c_float), P(u64)]),
        "lmat_stream_create": (i32, [vp, u64, u64, u32, i32, P(vp)]),
        "lmat_stream_acquire": (i32, [vp, P(vp), P(vp)]),
        "lmat_stream_submit": (i32, [vp, u64, u64]),
        "lmat_stream_submit_from": (i32, [vp, vp, vp, u64, u64]),
        "lmat_host_alloc": (i32, [u64, P(vp)]),
        "lmat_host_free": (None, [vp]),
        "lmat_stream_next": (i32, [vp, P(vp), P(vp), P(u64), P(u64), P(u64)]),
        "lmat_stream_release": (i32, [vp]),
        "lmat_stream_destroy": (None, [vp]),
        "lmat_counts_allreduce": (i32, [P(vp), i32]),
        "lmat_comm_unique_id": (i32, [vp]),
        "lmat_comm_init": (i32, [vp, vp, i32, i32]),
        "lmat_comm_allreduce_counts": (i32, [vp]),
        "lmat_comm_size": (i32, [vp]),
        "lmat_comm_destroy": (None, [vp]),
        "lmat_db_clone": (i32, [vp, vp]),
        "lmat_debug_decide": (i32, [vp, vp, vp, vp, vp, u64, vp]),
        "lmat_debug_decide_counts": (i32, [vp, vp, vp, vp, vp, u64, i32, vp]),
        "lmat_synth_window": (i32, [vp, u32, u64, P(u64), vp, u32, P(u32)]),
        "lmat_debug_last_counters": (i32, [vp, vp]),
        "lmat_debug_probe_stats": (i32, [vp, vp, u64, vp]),
        "lmat_debug_div_check": (i32, [vp, vp]),
        "lmat_synth_read_windows": (i32, [vp, vp, u32, u64, u64, vp, vp, vp, u32, u32, P(u32), P(u32)]),
        "lmat_table_address": (i32, [i32, u64, u64, P(u64), P(u32), P(u32)]),
        "lmat_format_out": (C.c_int64, [vp, vp, u64, vp, vp, vp, i32, u64, vp, u64]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


EXPORTED = ["lmat_device_count", "lmat_ctx_create", "lmat_ctx_destroy", "lmat_last_error", "lmat_set_params", "lmat_taxonomy_load_files",
            "lmat_db_begin", "lmat_genedb_begin", "lmat_db_add_taxhisto", "lmat_db_finalize", "lmat_db_kmer_length", "lmat_db_size", "lmat_db_list_count", "lmat_db_arena_bytes",
            "lmat_db_set_build_options", "lmat_db_save_image", "lmat_db_load_image", "lmat_ingest_create", "lmat_ingest_idmap_from_tree", "lmat_rand_mode", "lmat_rand_reset", "lmat_rand_label", "lmat_rand_get",
            "lmat_ingest_destroy", "lmat_ingest_error", "lmat_ingest_set_options", "lmat_ingest_add_taxhisto",
            "lmat_ingest_save_image", "lmat_ingest_load_image", "lmat_ingest_size", "lmat_ingest_kmer_length",
            "lmat_ingest_lookup", "lmat_db_from_ingest", "lmat_nullmodel_load", "lmat_nullmodel_clear", "lmat_set_label_modes",
            "lmat_db_table_bytes", "lmat_db_lookup", "lmat_synth_taxonomy", "lmat_synth_db_build", "lmat_synth_db_build2", "lmat_synth_db_build3", "lmat_reads_upload",
            "lmat_reads_synth", "lmat_reads_download_ascii", "lmat_reads_count", "lmat_reads_device_bytes",
            "lmat_reads_free", "lmat_classify", "lmat_classify_async", "lmat_classify_async_cands", "lmat_comm_available", "lmat_sync", "lmat_last_timing", "lmat_results_fetch",
            "lmat_counts_reset", "lmat_counts_layout", "lmat_counts_device_ptr", "lmat_counts_get", "lmat_gather_bench",
            "lmat_table_address", "lmat_format_out", "lmat_stream_create", "lmat_stream_acquire", "lmat_stream_submit", "lmat_stream_submit_from", "lmat_host_alloc", "lmat_host_free",
            "lmat_stream_next", "lmat_stream_release", "lmat_stream_destroy", "lmat_counts_allreduce",
            "lmat_comm_unique_id", "lmat_comm_init", "lmat_comm_allreduce_counts", "lmat_comm_size", "lmat_comm_destroy", "lmat_db_clone", "lmat_debug_decide", "lmat_debug_decide_counts", "lmat_synth_window", "lmat_synth_read_windows", "lmat_debug_probe_stats", "lmat_debug_last_counters", "lmat_debug_div_check"]


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Reads:
    def __init__(self, eng, handle):
        self.eng, self.h = eng, handle

    def __len__(self):
        return int(self.eng.lib.lmat_reads_count(self.h))

    @property
    def device_bytes(self):
        return int(self.eng.lib.lmat_reads_device_bytes(self.h))

    def ascii(self, first=0, count=None):
        """-> (bases uint8 array, offsets uint64[count+1]) decoded back from the packed device records."""
        n = len(self) - first if count is None else count
        off = np.zeros(n + 1, dtype=np.uint64)
        self.eng._chk(self.eng.lib.lmat_reads_download_ascii(self.eng.ctx, self.h, first, n, None, _ptr(off)))
        bases = np.zeros(int(off[-1]) + 1, dtype=np.uint8)
        self.eng._chk(self.eng.lib.lmat_reads_download_ascii(self.eng.ctx, self.h, first, n, _ptr(bases), _ptr(off)))
        return bases[:-1], off

    def free(self):
        if self.h:
            self.eng.lib.lmat_reads_free(self.eng.ctx, self.h)
            self.h = None


class Ingest:
    """GPU-free ingest (lmat_ingest): make_db_table's parsing and options, canonical 16-bit lists."""

    def __init__(self, k=None, idmap=None, image=None, tree=None):
        """idmap: the 32->16 map file; or tree (no map): a database of 32-bit taxids, codes from the tree."""
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.lmat_ingest_load_image(image.encode(), C.byref(h)) if image else \
            self.lib.lmat_ingest_create(k, idmap.encode() if idmap else None, C.byref(h))
        if rc != 0:
            raise LmatError(rc, "cannot create ingest")
        self.h = h
        if not image and not idmap:
            self._chk(self.lib.lmat_ingest_idmap_from_tree(self.h, tree.encode()))

    def _chk(self, rc):
        if rc != 0:
            raise LmatError(rc, self.lib.lmat_ingest_error(self.h).decode(errors="replace"))

    def set_options(self, tid_cutoff=0, rank_map=None, human_kmers=None, adaptor_kmers=None, adaptor_tid=0):
        e = lambda s: s.encode() if s else None
        self._chk(self.lib.lmat_ingest_set_options(self.h, tid_cutoff, e(rank_map), e(human_kmers), e(adaptor_kmers), adaptor_tid))

    def add_taxhisto(self, fn):
        self._chk(self.lib.lmat_ingest_add_taxhisto(self.h, fn.encode()))

    def save_image(self, fn):
        self._chk(self.lib.lmat_ingest_save_image(self.h, fn.encode()))

    def __len__(self):
        return int(self.lib.lmat_ingest_size(self.h))

    @property
    def k(self):
        return int(self.lib.lmat_ingest_kmer_length(self.h))

    def lookup(self, kmer, cap=4096):
        out = np.zeros(cap, dtype=np.uint16)
        n = self.lib.lmat_ingest_lookup(self.h, int(kmer), _ptr(out), cap)
        return out[:max(n, 0)].tolist()

    def close(self):
        if self.h:
            self.lib.lmat_ingest_destroy(self.h)
            self.h = None


def host_alloc(nbytes):
    """Page-locked host memory (lmat_host_alloc) -> address; free with host_free."""
    lib = load_library()
    p = C.c_void_p()
    rc = lib.lmat_host_alloc(int(nbytes), C.byref(p))
    if rc != 0:
        raise LmatError(rc, "out of pinned host memory")
    return p.value


def host_free(addr):
    load_library().lmat_host_free(C.c_void_p(addr))


class Stream:
    """lmat_stream: a ring of pinned batch slots; copy in, classification and copy out of consecutive batches overlap."""

    def __init__(self, eng, max_reads, max_bases, cands_per_read=0, n_slots=3):
        self.eng, self.lib = eng, eng.lib
        self.max_reads, self.max_bases, self.n_slots = int(max_reads), int(max_bases), n_slots
        h = C.c_void_p()
        eng._chk(self.lib.lmat_stream_create(eng.ctx, self.max_reads, self.max_bases, cands_per_read, n_slots, C.byref(h)))
        self.h = h
        self.in_flight = 0

    def submit(self, blob, off, tag=0):
        """blob: uint8 ASCII bases, off: uint64[n + 1]; copied into the slot's pinned buffers, then queued."""
        pb, po = C.c_void_p(), C.c_void_p()
        self.eng._chk(self.lib.lmat_stream_acquire(self.h, C.byref(pb), C.byref(po)))
        n = off.size - 1
        nb = int(off[-1])
        C.memmove(pb, blob.ctypes.data, nb)
        C.memmove(po, off.ctypes.data, (n + 1) * 8)
        self.eng._chk(self.lib.lmat_stream_submit(self.h, n, tag))
        self.in_flight += 1

    def submit_pinned(self, bases_ptr, off, n, tag=0):
        """bases_ptr: address of the caller's own pinned buffer (host_alloc); off: uint64[n + 1] numpy array.  No copy."""
        pb, po = C.c_void_p(), C.c_void_p()
        self.eng._chk(self.lib.lmat_stream_acquire(self.h, C.byref(pb), C.byref(po)))
        self.eng._chk(self.lib.lmat_stream_submit_from(self.h, bases_ptr, off.ctypes.data, n, tag))
        self.in_flight += 1

    def next_nocopy(self):
        """Waits for the oldest batch and releases it at once -> (n_reads, n_cands, tag), or None."""
        pr, pc = C.c_void_p(), C.c_void_p()
        n, nc, tag = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        rc = self.lib.lmat_stream_next(self.h, C.byref(pr), C.byref(pc), C.byref(n), C.byref(nc), C.byref(tag))
        if rc == 1:
            return None
        self.eng._chk(rc)
        self.eng._chk(self.lib.lmat_stream_release(self.h))
        self.in_flight -= 1
        return int(n.value), int(nc.value), int(tag.value)

    def next(self):
        """-> (results copy, cands copy or None, tag) of the oldest batch, or None when nothing is in flight."""
        pr, pc = C.c_void_p(), C.c_void_p()
        n, nc, tag = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        rc = self.lib.lmat_stream_next(self.h, C.byref(pr), C.byref(pc), C.byref(n), C.byref(nc), C.byref(tag))
        if rc == 1:
            return None
        self.eng._chk(rc)
        res = np.ctypeslib.as_array(C.cast(pr, C.POINTER(C.c_uint8)), shape=(n.value * READ_RESULT_DTYPE.itemsize,)).view(READ_RESULT_DTYPE).copy() if n.value else np.zeros(0, READ_RESULT_DTYPE)
        cands = None
        if pc.value and nc.value:
            cands = np.ctypeslib.as_array(C.cast(pc, C.POINTER(C.c_uint8)), shape=(nc.value * CAND_DTYPE.itemsize,)).view(CAND_DTYPE).copy()
        self.eng._chk(self.lib.lmat_stream_release(self.h))
        self.in_flight -= 1
        return res, cands, int(tag.value)

    def close(self):
        if self.h:
            self.lib.lmat_stream_destroy(self.h)
            self.h = None


class Engine:
    """One context on one GPU (lmat_ctx)."""

    def __init__(self, device=0, params: Params | None = None):
        self.lib = load_library()
        self.params = params or Params.run_rl()
        ctx = C.c_void_p()
        rc = self.lib.lmat_ctx_create(device, C.byref(self.params), C.byref(ctx))
        if rc != 0:
            raise LmatError(rc, "lmat_ctx_create failed (no usable HIP device?)")
        self.ctx = ctx

    def _chk(self, rc):
        if rc != 0:
            raise LmatError(rc, self.lib.lmat_last_error(self.ctx).decode(errors="replace"))

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.lmat_ctx_destroy(self.ctx)
            self.ctx = None

    def set_params(self, p: Params):
        self.params = p
        self._chk(self.lib.lmat_set_params(self.ctx, C.byref(p)))

    # taxonomy / DB --------------------------------------------------------------
    def load_taxonomy(self, tree, depth, rank, idmap, plasmids=None):
        e = lambda s: s.encode() if s else None
        self._chk(self.lib.lmat_taxonomy_load_files(self.ctx, e(tree), e(depth), e(rank), e(idmap), e(plasmids)))

    def build_gene_db(self, files, k=20, table_bytes=0, n_kmers_hint=0):
        """A gene database (gene_label): tax_histo-format files whose lists are 32-bit gene ids; no taxonomy needed."""
        self._chk(self.lib.lmat_genedb_begin(self.ctx, k, n_kmers_hint, table_bytes))
        for f in ([files] if isinstance(files, str) else files):
            self._chk(self.lib.lmat_db_add_taxhisto(self.ctx, f.encode()))
        self._chk(self.lib.lmat_db_finalize(self.ctx))

    def build_db(self, files, k=20, table_bytes=0, tid_cutoff=0, rank_map=None, human_kmers=None, adaptor_kmers=None,
                 save_image=None, n_kmers_hint=0):
        """make_db_table's job: tax_histo files (+ its -g/-m, -j, -u options) -> device table.
        With n_kmers_hint or table_bytes the table is sized up front and the files stream through the GPU insert
        kernel in chunks (nothing but the distinct lists stays on the host; save_image is then unavailable)."""
        e = lambda s: s.encode() if s else None
        self._chk(self.lib.lmat_db_begin(self.ctx, k, n_kmers_hint, table_bytes))
        if tid_cutoff or human_kmers or adaptor_kmers:
            self._chk(self.lib.lmat_db_set_build_options(self.ctx, tid_cutoff, e(rank_map), e(human_kmers), e(adaptor_kmers), 0))
        for f in ([files] if isinstance(files, str) else files):
            self._chk(self.lib.lmat_db_add_taxhisto(self.ctx, f.encode()))
        if save_image:
            self._chk(self.lib.lmat_db_save_image(self.ctx, save_image.encode()))
        self._chk(self.lib.lmat_db_finalize(self.ctx))

    def load_image(self, path, table_bytes=0):
        self._chk(self.lib.lmat_db_load_image(self.ctx, path.encode(), table_bytes))
        self._chk(self.lib.lmat_db_finalize(self.ctx))

    def save_device_image(self, path):
        """The finalized database as it lies in HBM (LMATIMG2): load_image() of another context with the same taxonomy streams it back in."""
        self._chk(self.lib.lmat_db_save_image(self.ctx, path.encode()))

    def set_label_modes(self, permissive=False, tid_cutoff=0, rank_map=None):
        """-s / -g N -m ranks; call before build_db / load_image."""
        self._chk(self.lib.lmat_set_label_modes(self.ctx, int(permissive), tid_cutoff, rank_map.encode() if rank_map else None))

    # rand_read_label (null-model generation) -------------------------------------
    def rand_mode(self, on=True):
        self._chk(self.lib.lmat_rand_mode(self.ctx, int(on)))

    def rand_reset(self, n_buckets=10):
        self._chk(self.lib.lmat_rand_reset(self.ctx, n_buckets))
        self._rand_nb = n_buckets

    def rand_label(self, reads, gc_bucket, first=0, count=None):
        n = len(reads) - first if count is None else count
        gb = np.ascontiguousarray(gc_bucket, dtype=np.uint8)
        assert gb.size == n
        self._chk(self.lib.lmat_rand_label(self.ctx, reads.h, first, n, _ptr(gb)))

    def rand_table(self):
        """-> {taxid: ([max label_prob per bucket], [hit count per bucket])}"""
        n = C.c_uint32(0)
        self._chk(self.lib.lmat_rand_get(self.ctx, None, None, None, 0, C.byref(n)))
        rows, nb = int(n.value), self._rand_nb
        tid = np.zeros(max(rows, 1), dtype=np.uint32)
        mx = np.zeros((max(rows, 1), nb), dtype=np.float32)
        ct = np.zeros((max(rows, 1), nb), dtype=np.uint32)
        self._chk(self.lib.lmat_rand_get(self.ctx, _ptr(tid), _ptr(mx), _ptr(ct), rows, C.byref(n)))
        return {int(tid[i]): (mx[i].copy(), ct[i].copy()) for i in range(rows)}

    def load_null_models(self, list_fn):
        """-n: null-model list file (gz tables resolved against $LMAT_DIR like the reference)."""
        self._chk(self.lib.lmat_nullmodel_load(self.ctx, list_fn.encode()))

    def clear_null_models(self):
        self._chk(self.lib.lmat_nullmodel_clear(self.ctx))

    def synth_taxonomy(self, branching=(3, 4, 4, 4, 4, 3)):
        b = np.asarray(branching, dtype=np.uint32)
        self._chk(self.lib.lmat_synth_taxonomy(self.ctx, _ptr(b)))

    def synth_db(self, genome_len, k=20, seed=2002, table_bytes=0, genus_block_permille=100, list_replicas=1, conserved_permille=(0, 0, 0)):
        """Synthetic database on the device: a tenth of every genome is a block shared within its genus;
        conserved_permille: the heavy tail -- blocks shared by a whole family / phylum / superkingdom (lists of 69 / 277 / 1109 taxids)."""
        cp = np.asarray(conserved_permille, dtype=np.uint32)
        assert cp.size == 3
        self._chk(self.lib.lmat_synth_db_build3(self.ctx, k, int(genome_len), seed, int(table_bytes), genus_block_permille, list_replicas, _ptr(cp)))

    @property
    def k(self):
        return int(self.lib.lmat_db_kmer_length(self.ctx))

    @property
    def db_size(self):
        return int(self.lib.lmat_db_size(self.ctx))

    @property
    def n_lists(self):
        return int(self.lib.lmat_db_list_count(self.ctx))

    @property
    def arena_bytes(self):
        return int(self.lib.lmat_db_arena_bytes(self.ctx))

    @property
    def table_bytes(self):
        return int(self.lib.lmat_db_table_bytes(self.ctx))

    def lookup(self, kmers, stride=64):
        km = np.ascontiguousarray(kmers, dtype=np.uint64)
        counts = np.zeros(km.size, dtype=np.uint32)
        tids = np.zeros((km.size, stride), dtype=np.uint32)
        self._chk(self.lib.lmat_db_lookup(self.ctx, _ptr(km), km.size, _ptr(counts), _ptr(tids), stride))
        return counts, tids

    # reads ----------------------------------------------------------------------
    def upload_reads(self, seqs) -> Reads:
        """seqs: list of str/bytes, or (uint8 blob, uint64 offsets)."""
        if isinstance(seqs, tuple):
            blob, off = seqs
        else:
            bs = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
            off = np.zeros(len(bs) + 1, dtype=np.uint64)
            np.cumsum([len(b) for b in bs], out=off[1:])
            blob = np.frombuffer(b"".join(bs) + b"\0", dtype=np.uint8)
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        h = C.c_void_p()
        self._chk(self.lib.lmat_reads_upload(self.ctx, _ptr(blob), _ptr(off), off.size - 1, C.byref(h)))
        return Reads(self, h)

    def synth_reads(self, n, lengths=(150,), seed=3003) -> Reads:
        ln = np.asarray(lengths, dtype=np.uint32)
        h = C.c_void_p()
        self._chk(self.lib.lmat_reads_synth(self.ctx, int(n), _ptr(ln), ln.size, seed, C.byref(h)))
        return Reads(self, h)

    # classify -------------------------------------------------------------------
    def classify(self, reads: Reads, first=0, count=None, want_cands=True, cand_cap=None):
        n = len(reads) - first if count is None else count
        res = np.zeros(n, dtype=READ_RESULT_DTYPE)
        ncand = C.c_uint64(0)
        if want_cands:
            cap = cand_cap or max(64 * n, 1024)
            cands = np.zeros(cap, dtype=CAND_DTYPE)
            self._chk(self.lib.lmat_classify(self.ctx, reads.h, first, n, _ptr(res), _ptr(cands), cap, C.byref(ncand)))
            return res, cands
        self._chk(self.lib.lmat_classify(self.ctx, reads.h, first, n, _ptr(res), None, 0, C.byref(ncand)))
        return res, None

    def classify_async(self, reads: Reads, first=0, count=None):
        n = len(reads) - first if count is None else count
        self._chk(self.lib.lmat_classify_async(self.ctx, reads.h, first, n))

    def classify_async_cands(self, reads: Reads, first=0, count=None, cand_cap=0):
        """As classify_async with the -p candidate pairs written to a device-side buffer of cand_cap pairs."""
        n = len(reads) - first if count is None else count
        self._chk(self.lib.lmat_classify_async_cands(self.ctx, reads.h, first, n, int(cand_cap)))

    def sync(self):
        ms, nl = C.c_float(0), C.c_uint64(0)
        self._chk(self.lib.lmat_sync(self.ctx, C.byref(ms), C.byref(nl)))
        return float(ms.value), int(nl.value)

    def last_timing(self):
        """-> (classify_kernel ms, k4_kernel + re-run ms, launches) of the launches the last sync() waited for."""
        a, b, n = C.c_float(0), C.c_float(0), C.c_uint64(0)
        self._chk(self.lib.lmat_last_timing(self.ctx, C.byref(a), C.byref(b), C.byref(n)))
        return float(a.value), float(b.value), int(n.value)

    def fetch_results(self, first, count):
        res = np.zeros(count, dtype=READ_RESULT_DTYPE)
        self._chk(self.lib.lmat_results_fetch(self.ctx, first, count, _ptr(res)))
        return res

    def gather_bench(self, n_probes, seed=1):
        """-> (ms, bytes) of a random 64-B bucket gather over the table (probe access shape)."""
        ms, b = C.c_float(0), C.c_uint64(0)
        self._chk(self.lib.lmat_gather_bench(self.ctx, int(n_probes), seed, C.byref(ms), C.byref(b)))
        return float(ms.value), int(b.value)

    def format_out(self, res, cands, reads_ascii=None, first_index=0):
        """.out text for these results; reads_ascii = (blob, off) to echo the read, else 'X' (-a)."""
        blob, off = reads_ascii if reads_ascii is not None else (None, None)
        need = self.lib.lmat_format_out(self.ctx, _ptr(res), res.size, _ptr(cands), _ptr(blob), _ptr(off),
                                        1 if blob is not None else 0, first_index, None, 0)
        buf = C.create_string_buffer(int(need) + 1)
        self.lib.lmat_format_out(self.ctx, _ptr(res), res.size, _ptr(cands), _ptr(blob), _ptr(off),
                                 1 if blob is not None else 0, first_index, buf, need + 1)
        return buf.raw[:need].decode()

    # tallies ----------------------------------------------------------------------
    def counts_reset(self):
        self._chk(self.lib.lmat_counts_reset(self.ctx))

    def counts_layout(self):
        n, b = C.c_uint32(0), C.c_uint64(0)
        self._chk(self.lib.lmat_counts_layout(self.ctx, C.byref(n), C.byref(b)))
        return int(n.value), int(b.value)

    def counts_device_ptr(self):
        return int(self.lib.lmat_counts_device_ptr(self.ctx) or 0)

    # tallies across ranks: RCCL inside the engine (collective.cpp) ---------------------
    @staticmethod
    def comm_available() -> bool:
        """librccl and the entry points the engine uses load in this process (pre-flight before any rank enters comm_init)."""
        return bool(load_library().lmat_comm_available())

    @staticmethod
    def comm_unique_id() -> bytes:
        """ncclGetUniqueId: 128 bytes made by one rank and handed to the others by the launcher's own channel."""
        buf = C.create_string_buffer(128)
        rc = load_library().lmat_comm_unique_id(buf)
        if rc != 0:
            raise LmatError(rc, "cannot open librccl / make a communicator id")
        return buf.raw

    def comm_init(self, uid: bytes, n_ranks: int, rank: int):
        assert len(uid) == 128
        self._chk(self.lib.lmat_comm_init(self.ctx, C.c_char_p(uid), n_ranks, rank))

    def comm_allreduce_counts(self):
        """The merge of read_label.cpp:1760-1800 across ranks: every rank's tallies become the sum of all."""
        self._chk(self.lib.lmat_comm_allreduce_counts(self.ctx))

    def synth_window(self, species, pos):
        """-> (canonical k-mer, [taxid32...]) the synthetic database must hold for that ancestor window (host-derived)."""
        km, n = C.c_uint64(0), C.c_uint32(0)
        t = np.zeros(32, dtype=np.uint32)
        self._chk(self.lib.lmat_synth_window(self.ctx, species, int(pos), C.byref(km), _ptr(t), 32, C.byref(n)))
        return int(km.value), t[:n.value].tolist()

    def last_counters(self):
        """Reads the last launch passed from class to class (after sync()): fast -> E=512 -> middle tier -> large LDS -> global memory."""
        out = np.zeros(16, dtype=np.uint32)
        self._chk(self.lib.lmat_debug_last_counters(self.ctx, _ptr(out)))
        return {"past_fast": int(out[2]), "past_e512": int(out[3]), "past_middle": int(out[10]), "past_large": int(out[7])}

    def div_check(self):
        """-> (pairs whose fast quotient differs from the IEEE one, pairs tried, the same two for 2^26 drawn (float, 1..64) pairs): the
        divisions of the decision step on the wave."""
        out = np.zeros(4, dtype=np.uint64)
        self._chk(self.lib.lmat_debug_div_check(self.ctx, _ptr(out)))
        return tuple(int(x) for x in out)

    def probe_stats(self, kmers):
        """-> dict: where the lookups of these k-mers end (home bucket / absent at once / overflow hit / overflow miss, overflow buckets read)."""
        km = np.ascontiguousarray(kmers, dtype=np.uint64)
        out = np.zeros(5, dtype=np.uint64)
        self._chk(self.lib.lmat_debug_probe_stats(self.ctx, _ptr(km), km.size, _ptr(out)))
        return dict(zip(("home_hit", "absent_one_request", "overflow_hit", "overflow_miss", "overflow_buckets_read"), (int(x) for x in out)))

    def synth_read_windows(self, lengths, seed, r, cap=512, stride=32):
        """-> (kmers uint64[n], counts uint32[n], tids uint32[n, stride]) read r of synth_reads(lengths, seed) must find (host-derived)."""
        ln = np.asarray(lengths, dtype=np.uint32)
        km = np.zeros(cap, dtype=np.uint64)
        ct = np.zeros(cap, dtype=np.uint32)
        td = np.zeros((cap, stride), dtype=np.uint32)
        n, rl = C.c_uint32(0), C.c_uint32(0)
        self._chk(self.lib.lmat_synth_read_windows(self.ctx, _ptr(ln), ln.size, seed, int(r), _ptr(km), _ptr(td), _ptr(ct), cap, stride,
                                                   C.byref(n), C.byref(rl)))
        return km[:n.value], ct[:n.value], td[:n.value]

    def debug_decide(self, tables, stdevs):
        """tables: list of [(taxid32, score), ...]; stdevs: one per table -> results array (call_tid, call_score, match_type)."""
        off = np.zeros(len(tables) + 1, dtype=np.uint64)
        np.cumsum([len(t) for t in tables], out=off[1:])
        tids = np.array([t for tb in tables for t, _ in tb], dtype=np.uint32)
        sc = np.array([s for tb in tables for _, s in tb], dtype=np.float32)
        sd = np.ascontiguousarray(stdevs, dtype=np.float32)
        res = np.zeros(len(tables), dtype=READ_RESULT_DTYPE)
        self._chk(self.lib.lmat_debug_decide(self.ctx, _ptr(tids), _ptr(sc), _ptr(off), _ptr(sd), len(tables), _ptr(res)))
        return res

    def debug_decide_counts(self, tids, counts, off, cand, on_the_wave):
        """tids / counts: flat uint32 arrays, off: uint64[n + 1], cand: uint32[n] -> results array of the chosen decision path."""
        tids = np.ascontiguousarray(tids, dtype=np.uint32)
        counts = np.ascontiguousarray(counts, dtype=np.uint32)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        cand = np.ascontiguousarray(cand, dtype=np.uint32)
        res = np.zeros(cand.size, dtype=READ_RESULT_DTYPE)
        self._chk(self.lib.lmat_debug_decide_counts(self.ctx, _ptr(tids), _ptr(counts), _ptr(off), _ptr(cand), cand.size, int(on_the_wave), _ptr(res)))
        return res

    def clone_db_from(self, src: "Engine"):
        self._chk(self.lib.lmat_db_clone(self.ctx, src.ctx))

    def counts(self):
        n, _ = self.counts_layout()
        tid = np.zeros(n, dtype=np.uint32)
        cnt = np.zeros(n, dtype=np.uint64)
        sc = np.zeros(n, dtype=np.float64)
        nz = C.c_uint32(0)
        nm = np.zeros(3, dtype=np.uint64)
        self._chk(self.lib.lmat_counts_get(self.ctx, _ptr(tid), _ptr(cnt), _ptr(sc), n, C.byref(nz), _ptr(nm)))
        k = int(nz.value)
        return {int(t): (int(c), float(s)) for t, c, s in zip(tid[:k], cnt[:k], sc[:k])}, [int(x) for x in nm]
