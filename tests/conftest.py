import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built artefacts: compile the engine (hipcc cross-compiles without a GPU) and the
    # oracle once per session; the tests themselves never fall back to anything if this fails
    import subprocess

    def make(*args):
        r = subprocess.run(["make", *args], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            pytest.exit("build failed: make %s\n%s" % (" ".join(args), r.stdout[-4000:]), returncode=2)

    if not os.path.exists(os.path.join(ROOT, "lmat_amd", "liblmat_hip.so")) or \
            not os.path.exists(os.path.join(ROOT, "lmat_amd", "csrc", "read_label")):
        make("-C", os.path.join(ROOT, "lmat_amd", "csrc"), "all")
    if not os.path.exists(os.path.join(ROOT, "oracle", "liblmat_oracle.so")):
        make("-C", os.path.join(ROOT, "oracle"), "lmat_oracle", "liblmat_oracle.so")


@pytest.fixture(scope="session")
def small_dataset(tmp_path_factory):
    """Synthetic taxonomy + tax_histo DB + reads, seeds 1001/2002/3003, small enough for seconds."""
    from lmat_amd import synth
    d = tmp_path_factory.mktemp("ds_small")
    info = synth.generate_dataset(str(d), (2, 2, 2, 2, 3, 3), 600, 400, frac_short=0.03, lower_frac=0.05)
    reads = []
    with open(info["fasta"]) as f:
        for line in f:
            if not line.startswith(">"):
                reads.append(line.rstrip("\n"))
    info["reads"] = reads
    return info


@pytest.fixture(scope="session")
def oracle_small(small_dataset):
    import oracle_py
    o = oracle_py.Oracle(small_dataset["tree"], small_dataset["depth"], small_dataset["rank"], small_dataset["idmap"])
    o.add_taxhisto(small_dataset["db"])
    o.set_options()
    yield o
    o.close()
