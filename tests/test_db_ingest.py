"""DB ingest tooling on CPU: the engine's GPU-free ingest and the oracle reproduce what the
REFERENCE's SortedDb::add_data stores under make_db_table's options (-g/-m pruning, -j human feed, -u adaptor
feed); fixtures come from the reference compiled in place (tests/golden/make_ref_goldens.py)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
DS = os.path.join(G, "ds")
OPTS = dict(tid_cutoff=2, rank_map=os.path.join(DS, "numeric_ranks.txt"), human_kmers=os.path.join(DS, "human_kmers.txt"),
            adaptor_kmers=os.path.join(DS, "adaptor_kmers.txt"))


def _golden(name):
    out = []
    for line in open(os.path.join(G, name)):
        f = line.split()
        out.append((int(f[0]), [int(x) for x in f[2:]]))
    return out


def _conv():
    m = {}
    for line in open(os.path.join(DS, "map32to16.txt")):
        a, b = line.split()
        m[int(b)] = int(a)
    return m


@pytest.mark.parametrize("with_opts", [False, True])
def test_ingest_matches_reference_add_data(with_opts, tmp_path):
    from lmat_amd import Ingest
    ing = Ingest(20, os.path.join(DS, "map32to16.txt"))
    if with_opts:
        ing.set_options(**OPTS)
    ing.add_taxhisto(os.path.join(DS, "th.bin"))
    conv = _conv()
    gold = _golden("ref_lookup_opts.txt" if with_opts else "ref_lookup.txt")
    changed = 0
    for km, want in gold:
        got = [conv[t] for t in ing.lookup(km)]
        assert got == want, km
    if with_opts:
        plain = dict(_golden("ref_lookup.txt"))
        changed = sum(1 for km, want in gold if km in plain and plain[km] != want)
        assert changed > 1000  # pruning / feeds really changed lists
    # image round trip through the GPU-free tool path
    img = str(tmp_path / "db.img")
    ing.save_image(img)
    back = Ingest(image=img)
    assert len(back) == len(ing) and back.k == 20
    for km, want in gold[::37]:
        assert [conv[t] for t in back.lookup(km)] == want
    ing.close()
    back.close()


def _tree_codes():
    ids = []
    with open(os.path.join(DS, "tax.dat")) as f:
        lines = f.read().split("\n")[3:]
    for i in range(0, len(lines) - 1, 2):
        if lines[i].strip():
            ids.append(int(lines[i].split()[0]))
    ids = sorted(set(ids))
    return {i + 1: t for i, t in enumerate(ids)}


@pytest.mark.parametrize("with_opts", [False, True])
def test_ingest_without_id_map_stores_the_same_lists(with_opts):
    """A database of 32-bit taxids (make_db_table without -f, TID_SIZE=32): storage codes come from the tree;
    decoded, the stored lists are the ones the reference's add_data stores."""
    from lmat_amd import Ingest
    ing = Ingest(20, tree=os.path.join(DS, "tax.dat"))
    if with_opts:
        ing.set_options(**OPTS)
    ing.add_taxhisto(os.path.join(DS, "th.bin"))
    code = _tree_codes()
    for km, want in _golden("ref_lookup_opts.txt" if with_opts else "ref_lookup.txt"):
        assert [code[t] for t in ing.lookup(km)] == want, km
    ing.close()


def test_oracle_matches_reference_add_data_with_options():
    import oracle_py
    o = oracle_py.Oracle(os.path.join(DS, "tax.dat"), os.path.join(DS, "depth.dat"), os.path.join(DS, "rank.txt"),
                         os.path.join(DS, "map32to16.txt"))
    o.set_build_options(OPTS["tid_cutoff"], OPTS["rank_map"], OPTS["human_kmers"], OPTS["adaptor_kmers"])
    o.add_taxhisto(os.path.join(DS, "th.bin"))
    for km, want in _golden("ref_lookup_opts.txt"):
        n, lst = o.lookup(km)
        assert (lst.tolist() if n > 0 else []) == want
    o.close()


def test_make_db_image_tool(tmp_path):
    exe = os.path.join(ROOT, "lmat_amd", "csrc", "make_db_image")
    img = str(tmp_path / "t.img")
    r = subprocess.run([exe, "-i", os.path.join(DS, "th.bin"), "-o", img, "-k", "20", "-f", os.path.join(DS, "map32to16.txt"),
                        "-g", "2", "-m", OPTS["rank_map"], "-j", OPTS["human_kmers"], "-u", OPTS["adaptor_kmers"]],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "new human k-mers:" in r.stdout
    from lmat_amd import Ingest
    back = Ingest(image=img)
    conv = _conv()
    for km, want in _golden("ref_lookup_opts.txt")[::11]:
        assert [conv[t] for t in back.lookup(km)] == want
    back.close()


@pytest.mark.parametrize("tag,ranks", [("ranks", True), ("noranks", False)])
def test_runtime_pruning_matches_reference_taxnodestat(tag, ranks):
    """read_label -g 2 [-m ranks]: the sequence TaxNodeStat::begin/next hands out (TaxNodeStat.hpp:60-256), taken from
    the REFERENCE's TaxNodeStat compiled in place; without a rank map only the first stored id survives."""
    import oracle_py
    o = oracle_py.Oracle(os.path.join(DS, "tax.dat"), os.path.join(DS, "depth.dat"), os.path.join(DS, "rank.txt"),
                         os.path.join(DS, "map32to16.txt"))
    o.add_taxhisto(os.path.join(DS, "th.bin"))
    o.set_label_modes(False, 2, os.path.join(DS, "numeric_ranks.txt") if ranks else None)
    pruned = 0
    plain = dict(_golden("ref_lookup.txt"))
    for km, want in _golden(f"ref_lookup_rt_{tag}.txt"):
        n, lst = o.lookup_rt(km)
        assert (lst.tolist() if n > 0 else []) == want
        pruned += want != plain[km]
    assert pruned > 3000
    o.close()


def _big_tree(tmp_path, extra=70000):
    """The fixture taxonomy with `extra` unrelated nodes under the root: more nodes than 16-bit ids can number."""
    src = open(os.path.join(DS, "tax.dat")).read().split("\n")
    head, body = src[:3], [l for l in src[3:]]
    big = str(tmp_path / "tax_big.dat")
    with open(big, "w") as f:
        f.write("\n".join(head) + "\n")
        f.write("\n".join(l for l in body if l != "") + "\n")
        for i in range(extra):
            f.write("%d 0 1\nfiller node %d\n" % (900000000 + i, i))
    return big


def test_map_from_database_for_a_taxonomy_above_16_bits(tmp_path):
    """A database of 32-bit taxids under a taxonomy of more than 65534 nodes (the reference's TID_SIZE=32 build): numbering the
    tree's nodes cannot work, so make_db_image -M makes the 32->16 map from the database's own taxids and their ancestors.
    Decoded through that map, the stored lists are the ones the reference's add_data stores."""
    exe = os.path.join(ROOT, "lmat_amd", "csrc", "make_db_image")
    big = _big_tree(tmp_path)
    img, mp = str(tmp_path / "t.img"), str(tmp_path / "derived_map.txt")
    base = [exe, "-i", os.path.join(DS, "th.bin"), "-o", img, "-k", "20", "-t", big]
    r = subprocess.run(base, capture_output=True, text=True)
    assert r.returncode != 0 and "65534" in r.stderr  # the tree alone is too large to number
    r = subprocess.run(base + ["-M", mp, "-g", "2", "-m", OPTS["rank_map"], "-j", OPTS["human_kmers"], "-u", OPTS["adaptor_kmers"]],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    conv = {}
    for line in open(mp):
        a, b = line.split()
        conv[int(b)] = int(a)
    assert 0 < len(conv) < 65535 and not any(t >= 900000000 for t in conv.values())  # only ids the database can reach
    from lmat_amd import Ingest
    back = Ingest(image=img)
    for km, want in _golden("ref_lookup_opts.txt")[::7]:
        assert [conv[t] for t in back.lookup(km)] == want
    back.close()


def test_corrupt_images_are_refused(tmp_path):
    """A database image is a file the user names: header sizes beyond the file, a payload that points past the lists, k-mers
    out of order or wider than 2k bits end in an error from the loader, not in an out-of-bounds access or an uncaught
    allocation failure."""
    import struct
    from lmat_amd import Ingest
    ing = Ingest(20, os.path.join(DS, "map32to16.txt"))
    ing.add_taxhisto(os.path.join(DS, "th.bin"))
    img = str(tmp_path / "ok.img")
    ing.save_image(img)
    n = len(ing)
    ing.close()
    raw = bytearray(open(img, "rb").read())
    magic, k, nk, nl = raw[:8], *struct.unpack_from("<IQQ", raw, 8)
    assert magic == b"LMATIMG1" and k == 20 and nk == n

    def refused(mut):
        bad = bytearray(raw)
        mut(bad)
        fn = str(tmp_path / "bad.img")
        open(fn, "wb").write(bad)
        with pytest.raises(Exception):
            Ingest(image=fn)

    refused(lambda b: struct.pack_into("<Q", b, 12, 1 << 60))                        # k-mer count beyond the file
    refused(lambda b: struct.pack_into("<Q", b, 20, 1 << 60))                        # list count beyond the file
    refused(lambda b: struct.pack_into("<I", b, 28 + 8 * nk, 65536 + nl + 5))        # payload past the lists
    refused(lambda b: struct.pack_into("<Q", b, 28 + 8, struct.unpack_from("<Q", b, 28)[0]))  # k-mers not ascending
    refused(lambda b: struct.pack_into("<Q", b, 28 + 8 * (nk - 1), 1 << 50))         # key wider than 2k bits
    refused(lambda b: b.__delitem__(slice(len(b) - 100, len(b))))                    # truncated
    back = Ingest(image=img)  # the untouched image still loads
    assert len(back) == n
    back.close()
