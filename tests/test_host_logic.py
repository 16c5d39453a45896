"""CPU tests of host-side logic: C-ABI surface, std::sort replica, reader quirks, generator determinism."""
import os
import re
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import lmat_amd
    from lmat_amd import capi
    lib = lmat_amd.load_library()
    hdr = open(os.path.join(ROOT, "include", "lmat_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(lmat_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 28
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(capi.EXPORTED) == declared


def test_no_gpu_means_loud_failure():
    import lmat_amd
    import torch
    if torch.cuda.is_available():
        return
    try:
        lmat_amd.Engine()
    except lmat_amd.LmatError as e:
        assert e.code == -3
    else:
        raise AssertionError("Engine() must fail without a GPU (no CPU fallback)")


def test_std_sort_replica_matches_libstdcxx(tmp_path):
    """ss_sort (lmat_common.hpp) is the device's std::sort: same permutation as libstdc++ on random inputs,
    including the non-strict-weak TCmp comparator (read_label.cpp:475-485) and n > 16 (introsort path)."""
    src = tmp_path / "t.cpp"
    src.write_text(r'''
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#include "lmat_common.hpp"
struct E { float s; int d; int id; };
struct TCmp { bool operator()(const E& a, const E& b) const {
    if (fabs(a.s - b.s) < 0.001) return a.d < b.d; return a.s < b.s; } };
struct DCmp { bool operator()(const E& a, const E& b) const { return a.d > b.d; } };
template <class C> int run(int iters, unsigned seed, int maxn, C cmp) {
    std::mt19937 g(seed);
    for (int it = 0; it < iters; ++it) {
        int n = 1 + g() % maxn;
        std::vector<E> a(n);
        for (int i = 0; i < n; ++i) { a[i].s = (g() % 40) * 0.0004f + (g() % 3) * 0.01f; a[i].d = g() % 6; a[i].id = i; }
        std::vector<E> b = a;
        std::sort(a.begin(), a.end(), cmp);
        lmat::ss_sort(b.data(), n, cmp);
        for (int i = 0; i < n; ++i) if (a[i].id != b[i].id) { printf("MISMATCH n=%d i=%d\n", n, i); return 1; }
    }
    return 0;
}
int main() {
    if (run(20000, 1, 16, TCmp())) return 1;
    if (run(20000, 2, 200, TCmp())) return 1;
    if (run(5000, 3, 2000, DCmp())) return 1;
    if (run(20000, 4, 40, DCmp())) return 1;
    printf("OK\n");
    return 0;
}''')
    exe = tmp_path / "t"
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "lmat_amd", "csrc"), str(src), "-o", str(exe)])
    assert subprocess.run([str(exe)], capture_output=True, text=True).stdout.strip() == "OK"


def _oracle(ds):
    import oracle_py
    o = oracle_py.Oracle(ds["tree"], ds["depth"], ds["rank"], ds["idmap"])
    o.add_taxhisto(ds["db"])
    o.set_options()
    return o


def test_fastq_header_lag_and_fasta_rules(small_dataset, tmp_path):
    """Reader quirks restated from main() (read_label.cpp:1651-1713): FASTQ records carry the previous
    record's header and the first is unknown_hdr:1 (Q2); FASTA lines of length <= 1 are dropped (Q3);
    multi-line FASTA is concatenated."""
    o = _oracle(small_dataset)
    seqs = small_dataset["reads"][:6]
    fq = tmp_path / "a.fq"
    fq.write_text("".join(f"@h{i}\n{s}\n+\n{'I' * len(s)}\n" for i, s in enumerate(seqs)))
    o.set_options(fastq=1)
    text, _, _ = o.run_file(str(fq), 20)
    hdrs = [l.split("\t")[0] for l in text.splitlines()]
    assert hdrs == ["unknown_hdr:1"] + [f"h{i}" for i in range(5)]
    o.set_options(fastq=0)
    fa = tmp_path / "a.fa"
    s0, s1 = seqs[0], seqs[1]
    fa.write_text(f">x0\n{s0[:70]}\n{s0[70:]}\n>x1\nA\n>x2\n{s1}\n")
    text, _, nm = o.run_file(str(fa), 20)
    lines = text.splitlines()
    assert [l.split("\t")[0] for l in lines] == ["x0", "x2"]  # the 1-base record never becomes a read
    assert lines[0].split("\t")[1] == s0
    o.close()


def test_silent_record_quirk(small_dataset):
    """Q1: valid k-mers >= min but distinct k-mers < min -> 'hdr\\tread\\t' with no newline, tallied NoDbHits."""
    o = _oracle(small_dataset)
    unit = small_dataset["reads"][1][:25]
    read = (unit * 6)[:150]
    hit = small_dataset["reads"][1]
    blob = np.frombuffer((read + hit).encode() + b"\0", dtype=np.uint8)
    off = np.array([0, len(read), len(read) + len(hit)], dtype=np.uint64)
    text, tally, nm = o.classify(blob, off, 20)
    assert text.startswith(f"r0\t{read}\tr1\t{hit}\t")
    assert nm[1] == 1
    o.close()


def test_generator_is_deterministic(tmp_path):
    from lmat_amd import synth
    a = synth.generate_dataset(str(tmp_path / "a"), (2, 2, 2, 2, 2, 2), 200, 20)
    b = synth.generate_dataset(str(tmp_path / "b"), (2, 2, 2, 2, 2, 2), 200, 20)
    for k in ("db", "fasta", "tree", "idmap"):
        assert open(a[k], "rb").read() == open(b[k], "rb").read()


def test_committed_dataset_matches_generator(tmp_path):
    """tests/golden/ds is exactly what the generator emits today (fixtures stay reproducible)."""
    from lmat_amd import synth
    g = os.path.join(ROOT, "tests", "golden", "ds")
    a = synth.generate_dataset(str(tmp_path / "a"), (2, 2, 2, 2, 3, 3), 300, 300, frac_short=0.03, lower_frac=0.05)
    for f in ("th.bin", "reads.fa", "tax.dat", "map32to16.txt", "depth.dat", "rank.txt"):
        assert open(os.path.join(g, f), "rb").read() == open(os.path.join(str(tmp_path / "a"), f), "rb").read(), f


def test_device_logf_algorithm_matches_host_libm(tmp_path):
    """The null-model score is std::log(float) (read_label.cpp:680-690).  The device evaluates glibc's logf
    algorithm in double arithmetic; the same C code compiled for the host must give libm's bits on a dense
    sweep of the positive floats (the GPU parity tests then check the device build of it)."""
    src = tmp_path / "l.cpp"
    kern = open(os.path.join(ROOT, "lmat_amd", "csrc", "kernels.hip")).read()
    i0 = kern.index("__device__ __constant__ double kLogfInvC[16]")
    i1 = kern.index("// which null-model table a read")
    body = kern[i0:i1].replace("__device__ __constant__ ", "static const ").replace("__device__ __forceinline__ ", "static inline ")
    body = body.replace("__float_as_uint(", "f2u(").replace("__uint_as_float(", "u2f(")
    src.write_text(r'''
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
''' + body + r'''
int main() {
    unsigned long long bad = 0, n = 0;
    for (uint32_t u = 1; u < 0x7f800000u; u += 13) { float x = u2f(u); float a = logf(x), b = glibc_logf(x); ++n; if (f2u(a) != f2u(b)) ++bad; }
    for (int c = 1; c <= 300; ++c) for (int t = 1; t <= c; ++t) for (int d = 1; d < 400; d += 7) {
        float x = ((float)t / (float)c) / (0.0001f * d); float a = logf(x), b = glibc_logf(x); ++n; if (f2u(a) != f2u(b)) ++bad; }
    printf("%llu %llu\n", n, bad);
    return bad != 0;
}''')
    exe = tmp_path / "l"
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", str(src), "-o", str(exe)])
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    assert int(r.stdout.split()[0]) > 1e8


def test_float_text_matches_printf_g():
    """The .out writer formats floats with std::to_chars(general, 6); it must be printf's "%g" (what the
    reference's operator<<(float) prints) for every value: random bit patterns, score-like fractions, edge cases."""
    import struct
    import subprocess
    exe = os.path.join(ROOT, "lmat_amd", "csrc", "fmt_check")
    rng = np.random.default_rng(7)
    bits = list(rng.integers(0, 2**32, size=200000, dtype=np.uint64).astype(np.uint32))
    fr = [np.float32(a) / np.float32(b) for b in range(1, 140) for a in range(0, b + 1)]
    fr += [np.float32(x) for x in (0.0, -0.0, 1.0, 1e-5, 9.99999e-5, 1e-4, 123456.5, 999999.5, 999999.4, 1e6, 1.5e-45, 3.4e38, -10000.0,
                                   0.0009999995, 0.001, 99999.95, 0.5, 2.5, 1234565.0, 0.1234565)]
    bits += [np.frombuffer(np.float32(x).tobytes(), dtype=np.uint32)[0] for x in fr]
    inp = "".join("%08x\n" % int(b) for b in bits)
    got = subprocess.run([exe], input=inp, capture_output=True, text=True, check=True).stdout.split("\n")[:-1]
    assert len(got) == len(bits)
    for b, g in zip(bits, got):
        f = struct.unpack("<f", struct.pack("<I", int(b)))[0]
        want = "%g" % f
        if f != f:  # NaN: sign spelling is irrelevant to the path (scores are never NaN)
            assert "nan" in g
            continue
        assert g == want, (hex(int(b)), g, want)
