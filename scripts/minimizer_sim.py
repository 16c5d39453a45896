"""Design aid: bucket statistics of a minimizer-addressed k-mer table.
Random genomes with 1 % strain substitutions (the synthetic DB's shape); k-mers go to the bucket chosen by a hash of
their canonical minimizer.  Reports the share of full buckets / displaced k-mers and the number of distinct buckets a
150 bp read touches, for several minimizer lengths."""
import sys
import numpy as np

def canon(x, nb):  # canonical of a 2*nb-bit forward-encoded word array (first base in high bits)
    rc = np.zeros_like(x)
    t = ~x
    for i in range(nb):
        rc = (rc << np.uint64(2)) | ((t >> np.uint64(2 * i)) & np.uint64(3))
    rc &= np.uint64((1 << (2 * nb)) - 1)
    return np.minimum(x, rc)

def mix(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(33); x *= np.uint64(0xff51afd7ed558ccd)
    x ^= x >> np.uint64(33); x *= np.uint64(0xc4ceb9fe1a85ec53)
    x ^= x >> np.uint64(33)
    return x

def feistel(x, m, ks):
    mm = np.uint64((1 << m) - 1)
    L = (x >> np.uint64(m)); R = x & mm
    for i, kk in enumerate(ks):
        if i % 2 == 0: L = L ^ (((R * np.uint64(kk)) & np.uint64(0xFFFFFFFF)) >> np.uint64(7)) & mm
        else: R = R ^ (((L * np.uint64(kk)) & np.uint64(0xFFFFFFFF)) >> np.uint64(7)) & mm
    return (L << np.uint64(m)) | R

SCR = (0x6A09, 0x3B67, 0x5F1D, 0x7C15)   # cpt_scramble / cpt_mix of lmat_common.hpp
MIX = (0x52DB, 0x4F6D, 0x6E2B, 0x35A7)

def words(seq, n):
    out = np.zeros(len(seq) - n + 1, dtype=np.uint64)
    for i in range(n):
        out = (out << np.uint64(2)) | seq[i:len(seq) - n + 1 + i].astype(np.uint64)
    return out

def minimizers(seq, k, m):
    mm = feistel(canon(words(seq, m), m), m, SCR) if ENGINE else mix(canon(words(seq, m), m))  # ordering hash per m-mer position
    w = k - m + 1
    P = len(seq) - k + 1
    best = mm[0:P].copy()
    for j in range(1, w):
        best = np.minimum(best, mm[j:P + j])
    return best

ENGINE = len(sys.argv) > 3 and sys.argv[3] == "engine"   # the engine's own Feistel scramblers instead of murmur

def bucket_of(mn, m, nb):
    if ENGINE:
        return (feistel(mn, m, MIX).astype(np.float64) * nb / float(1 << (2 * m))).astype(np.int64)
    return (mix(mn ^ np.uint64(0x9e3779b97f4a7c15)) % np.uint64(nb)).astype(np.int64)

def main():
    rng = np.random.default_rng(7)
    G, k = 2_000_000, 20
    slots, avg = int(sys.argv[1]), float(sys.argv[2])
    anc = rng.integers(0, 4, G, dtype=np.uint8)
    genomes = [anc]
    for s in range(3):
        g = anc.copy()
        mut = rng.random(G) < 0.01
        g[mut] = (g[mut] + rng.integers(1, 4, mut.sum())) & 3
        genomes.append(g)
    for m in ((17, 16, 15) if ENGINE else (20, 18, 17, 16, 15, 14)):
        keys, mins = [], []
        for g in genomes:
            keys.append(canon(words(g, k), k)); mins.append(minimizers(g, k, m) if m < k else mix(canon(words(g, k), k)))
        keys = np.concatenate(keys); mins = np.concatenate(mins)
        uk, idx = np.unique(keys, return_index=True)
        um = mins[idx]
        nb = int(len(uk) / avg)
        b = bucket_of(um, m, nb)
        cnt = np.bincount(b, minlength=nb)
        full = (cnt >= slots).mean()
        displaced = np.maximum(cnt - slots, 0).sum() / len(uk)
        # distinct buckets per 150 bp read from strain 1
        g = genomes[1]
        touched = []
        for off in rng.integers(0, G - 150, 300):
            r = g[off:off + 150]
            mn = minimizers(r, k, m) if m < k else mix(canon(words(r, k), k))
            bb = bucket_of(mn, m, nb)
            runs = 1 + int((bb[1:] != bb[:-1]).sum())
            touched.append((len(np.unique(bb)), runs))
        t = np.array(touched)
        print("m=%2d  kmers %d  buckets %d  full-bucket share %.4f  displaced k-mers %.4f  max load %d  buckets/read %.1f (runs %.1f)"
              % (m, len(uk), nb, full, displaced, cnt.max(), t[:, 0].mean(), t[:, 1].mean()))

if __name__ == "__main__":
    main()
