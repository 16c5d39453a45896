// lmat_common.hpp -- layouts and small pure functions shared by host and device code.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define LM_HD __host__ __device__ __forceinline__
#else
#define LM_HD inline
#endif

namespace lmat {

// ---- device hash of the k-mer database -------------------------------------------------
// slot (u64) = canonical k-mer (40 bits for k<=20) << 24 | payload (24 bits); 0 = empty.
// payload 1..65535          : plain singleton, value = internal taxid index
// payload 65536 + o         : taxid-list record at arena[kListUnit*o] (arena in u16 units)
// bucket = 8 slots = 64 B = one HBM sector read by 4 lanes (16 B each); slots fill front to back; linear probing over buckets.
static const int kPayloadBits = 24;
static const uint32_t kPayloadMask = (1u << kPayloadBits) - 1;
static const uint32_t kListBase = 65536;
static const int kSlotsPerBucket = 8;

// list record in the arena (u16 units; a record starts on a 16-byte boundary and payload 65536 + o is the record at
// arena[kListUnit * o], so 2^24 payloads address 256 MB of records and a record's first 16 bytes are one aligned load):
//   [0] flags  bit0: raw count >= 32768 (label_vec.first goes negative, read_label.cpp:49,1045)
//   [1] n_kept [2] n_raw
//   [3 .. 3+n_kept)            kept ids, registration order (depth-sorted leaf-most set)
//   [3+n_kept .. 3+2*n_kept)   kept ids ascending by 32-bit taxid
//   [3+2*n_kept .. +n_raw)     raw list as stored in the DB (16-bit DB ids), for lookups
static const int kListHdr = 3;
static const int kListUnit = 8;   // u16 units per payload step at shift 0
// Larger arenas: a table may place its records on (16 << list_shift)-byte boundaries instead, so that the same 24-bit payloads
// address 256 MB << list_shift (up to 4 GB); payload 65536 + o is then the record at arena[(kListUnit << list_shift) * o].
static const int kListShiftMax = 4;
#define LMAT_LIST_OFF(pay, shift) ((size_t)kListUnit * ((size_t)((pay) - kListBase) << (shift)))
static const uint16_t kListNegFirst = 1;

// per-taxid flags (internal index space)
static const uint8_t kFlagStrain = 1;   // rank file says "strain"  (read_label.cpp:1150,1184)
static const uint8_t kFlagHuman = 2;    // isHuman (tid_checks.hpp:15-28)
static const uint8_t kFlagPhiX = 4;     // isPhiX (tid_checks.hpp:13)
static const uint8_t kFlagPlasmid = 8;  // isPlasmid (read_label.cpp:69)

LM_HD uint64_t mix64(uint64_t x) {  // murmur3 finaliser: k-mers of one genome overlap heavily
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}
LM_HD uint32_t bucket_of(uint64_t kmer, uint32_t nbuckets) {
    uint64_t h = mix64(kmer);
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__umul64hi(h, (uint64_t)nbuckets);
#else
    return (uint32_t)(((unsigned __int128)h * nbuckets) >> 64);
#endif
}

// ---- compact table: minimizer-addressed quotient buckets --------------------------------------
// The default layout of large tables (any table the tag width allows, see cpt_geometry).  A k-mer is filed under its
// MINIMIZER: the canonical m-mer (m = k - 3, so w = 4 m-mers per k-mer) that is smallest under a scrambled order.
// Consecutive k-mers of a read (and of a genome) mostly share their minimizer, so they land in the same 64-byte
// bucket: a 150 bp read touches ~53 buckets instead of 131.  The bucket index consumes most of the minimizer's
// bits (quotienting), so a slot needs only a 16-bit tag beside its 24-bit payload:
//   bucket (64 B) = tag u16[12] | payload low u16[12] | payload high u8[12] | header u32
//   header: bits 0..23 insert counter, bit 31 = a k-mer of this bucket lives in the overflow table
//   tag = 1 + ((((rho << lowbits | low) * 4 + j) * 2 + strand) * 64 + other), 0 = empty slot, where
//     S'     = mix(scramble(minimizer))  (bijective on 2m bits),  hi = S' >> lowbits, low = S' & (2^lowbits - 1)
//     bucket = hi / W,  rho = hi mod W   (W = width of a bucket in hi-space, a run-time integer)
//     j      = position of the minimizer inside the canonical k-mer, strand = the m-mer there is the larger of its pair
//     other  = the k-mer's 3 bases outside the minimizer
//   (bucket, tag) <-> canonical k-mer is a bijection, so a tag match is an exact key match.
// 12 slots at ~10 bytes per k-mer is a load of ~0.5: a k-mer whose bucket is full goes to a small overflow table in
// the wide (8 x 64-bit slot) layout above, keyed by the full k-mer; the header bit tells lookups to go there.
static const int kCptSlots = 12;
static const int kCptW = 4;                       // m-mers per k-mer
static const uint32_t kCptOvfFlag = 0x80000000u;
static const uint32_t kCptCountMask = 0x00FFFFFFu;

struct CptGeom {
    uint64_t nb = 0;       // buckets (up to 2^32: bucket indices are 32 bits wide); 0 = the table is in the wide layout
    uint32_t W = 0;        // bucket width in hi-space
    double invW = 0;       // 1.0 / W
    int k = 0, m = 0, lowbits = 0;
    int wshift = -1;       // log2(W) when W is a power of two (the classify kernel then shifts instead of multiplying in double), else -1;
                           // -2: FRACTIONAL width -- nb is any number between 2^30 and 2^32 and bucket = hi * nb >> 32 (below)
};

#if defined(__HIP_DEVICE_COMPILE__)
#define LM_MUL24(a, b) __umul24((a), (b))
#else
#define LM_MUL24(a, b) ((uint32_t)(a) * (uint32_t)(b))
#endif

// four Feistel rounds on the two m-bit halves of a 2m-bit value (m <= 17): a bijection whatever the round function
LM_HD uint64_t cpt_feistel(uint64_t x, int m, uint32_t ka, uint32_t kb, uint32_t kc, uint32_t kd) {
    const uint32_t mm = (1u << m) - 1;
    uint32_t L = (uint32_t)(x >> m), R = (uint32_t)x & mm;
    L ^= (LM_MUL24(R, ka) >> 7) & mm;  // R < 2^17, ka < 2^15: the product stays below 2^32
    R ^= (LM_MUL24(L, kb) >> 7) & mm;
    L ^= (LM_MUL24(R, kc) >> 7) & mm;
    R ^= (LM_MUL24(L, kd) >> 7) & mm;
    return ((uint64_t)L << m) | R;
}
LM_HD uint64_t cpt_scramble(uint64_t y, int m) { return cpt_feistel(y, m, 0x6A09u, 0x3B67u, 0x5F1Du, 0x7C15u); }  // minimizer order
LM_HD uint64_t cpt_mix(uint64_t s, int m) { return cpt_feistel(s, m, 0x52DBu, 0x4F6Du, 0x6E2Bu, 0x35A7u); }       // bucket hash

// reverse complement of a forward-encoded k-mer (first base in the high bits)
LM_HD uint64_t revcomp_fwd(uint64_t fwd, int k) {
    uint64_t x = ~fwd;
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    x = ((x >> 8) & 0x00FF00FF00FF00FFull) | ((x & 0x00FF00FF00FF00FFull) << 8);
    x = ((x >> 16) & 0x0000FFFF0000FFFFull) | ((x & 0x0000FFFF0000FFFFull) << 16);
    x = (x >> 32) | (x << 32);
    return x >> (64 - 2 * k);
}

// geometry for a table of about want_buckets buckets; nb == 0 when the compact layout cannot hold this k / size
// (then the caller uses the wide layout).  min_buckets: the smallest table the 16-bit tag allows for this k.
LM_HD CptGeom cpt_geometry(int k, uint64_t want_buckets, bool allow_fractional = true) {
    CptGeom g;
    if (k < 10 || k > 20) return g;
    const int m = k - (kCptW - 1), n = 2 * m;
    const int lowbits = n > 32 ? n - 32 : 0;
    const uint64_t space = 1ull << (n - lowbits);        // hi-space
    const uint64_t wmax = 127u >> lowbits;               // (W << lowbits) * 512 + 1 <= 65536
    if (want_buckets < 1) want_buckets = 1;
    uint64_t W = space / want_buckets;                   // floor: never fewer buckets than asked for
    if (W < 1) W = 1;
    if (W > wmax) W = wmax;
    const uint64_t nb = (space + W - 1) / W;
    g.k = k; g.m = m; g.lowbits = lowbits;
    // An integer width quantises the large tables coarsely: between 2^31 and 2^32 buckets (128 and 256 GiB) there is nothing,
    // and a 200 GB-class database landed on 2^31 buckets at a load of 8.7 of 12 slots with a quarter of its k-mers displaced.
    // Where the integer width overshoots the request by more than 2 % (and hi-space is the full 32 bits: k >= 19), the table
    // gets exactly the buckets asked for and a FRACTIONAL width: bucket = hi * nb >> 32, and rho = the place of hi among the
    // (at most 4) values that share the bucket = floor((hi * nb mod 2^32) / nb).  (bucket, rho) <-> hi stays a bijection.
    if (allow_fractional && n - lowbits == 32 && want_buckets > (1ull << 30) && want_buckets < (1ull << 32) && nb * 50 > want_buckets * 51) {
        g.nb = want_buckets;
        g.W = (uint32_t)((space + want_buckets - 1) / want_buckets);   // values of hi per bucket, at most (2 .. 4): bounds rho
        g.invW = 1.0 / (double)g.W;
        g.wshift = -2;
        return g;
    }
    g.nb = nb; g.W = (uint32_t)W; g.invW = 1.0 / (double)W;
    g.wshift = (W & (W - 1)) == 0 ? 63 - __builtin_clzll(W) : -1;
    return g;
}

// canonical k-mer c (c <= cr, cr = its reverse complement, both forward-encoded) -> bucket and tag
LM_HD void cpt_address(const CptGeom& g, uint64_t c, uint64_t cr, uint32_t& bucket, uint32_t& tag) {
    const int m = g.m;
    const uint64_t mmask = (1ull << (2 * m)) - 1;
    uint64_t best = ~0ull;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < kCptW; ++j) {
        const uint64_t x = (c >> (2 * (kCptW - 1 - j))) & mmask;  // m-mer j of the canonical k-mer
        const uint64_t xr = (cr >> (2 * j)) & mmask;              // its reverse complement
        const uint64_t y = x < xr ? x : xr;
        const uint64_t v = (cpt_scramble(y, m) << 3) | ((uint64_t)j << 1) | (xr < x ? 1u : 0u);
        best = v < best ? v : best;                                // equal minimizers: the smaller j
    }
    const int j = (int)(best >> 1) & 3;
    const uint64_t sp = cpt_mix(best >> 3, m);
    const uint32_t hi = (uint32_t)(sp >> g.lowbits), low = (uint32_t)sp & ((1u << g.lowbits) - 1);
    uint32_t b, rho;
    if (g.wshift == -2) {  // fractional width
        const uint64_t prod = (uint64_t)hi * (uint64_t)(uint32_t)g.nb;
        const uint64_t lowp = prod & 0xFFFFFFFFull, nb64 = g.nb;
        b = (uint32_t)(prod >> 32);
        rho = (lowp >= nb64 ? 1u : 0u) + (lowp >= 2 * nb64 ? 1u : 0u) + (lowp >= 3 * nb64 ? 1u : 0u);
    } else {
        // hi / W, exact: (hi + 0.5) / W is never within 2^-8 of an integer and the double product is good to 2^-20
        b = (uint32_t)(((double)hi + 0.5) * g.invW);
        rho = hi - b * g.W;
    }
    const int rs = 2 * (kCptW - 1 - j);                            // bits of the k-mer right of the minimizer
    const uint32_t other = (uint32_t)((c >> (2 * m + rs)) << rs) | ((uint32_t)c & ((1u << rs) - 1));
    bucket = b;
    tag = 1 + (((((rho << g.lowbits) | low) * 4 + (uint32_t)j) * 2 + ((uint32_t)best & 1u)) * 64 + other);
}

// ---- packed read record (4-byte words) ---------------------------------------------------
//   word 0: length in bases; then ceil(len/16) words of 2-bit codes (base j at bits 2*(j%16)),
//   then ceil(len/32) words of validity bits (1 = ACGT).
LM_HD uint32_t rec_words(uint32_t len) { return 1 + (len + 15) / 16 + (len + 31) / 32; }

// ---- counter-based PRNG for the synthetic genomes / reads --------------------------------
LM_HD uint64_t splitmix(uint64_t x) {
    x += 0x9e3779b97f4a7c15ULL;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
    return x ^ (x >> 31);
}

// ---- libstdc++ std::sort, restated ---------------------------------------------------------
// read_label sorts candidates with comparators that are not strict weak orders
// (TCmp, read_label.cpp:475-485) and relies on whatever libstdc++'s introsort does
// with ties, so the device needs the same algorithm, not just "a sort": insertion
// sort for n <= 16 (stable), median-of-3 quicksort + final insertion sort above,
// heapsort when the depth limit 2*floor(log2 n) is exhausted.  Works on any
// random-access array T with cmp(a,b).  Unguarded scans are clamped to the array:
// the reference would run out of bounds there (undefined behaviour).
// (LM_NOUNROLL: on the GPU these loops run in lanes that each follow their own path; unrolled and peeled they made the decision
// kernels 45-90 KB of code, more than the instruction cache holds.)
#if defined(__clang__)
#define LM_NOUNROLL _Pragma("nounroll")
#else
#define LM_NOUNROLL
#endif
template <class T, class Cmp>
LM_HD void ss_unguarded_linear_insert(T* first, int last, Cmp& cmp) {
    T val = first[last];
    int next = last - 1;
    LM_NOUNROLL
    while (next >= 0 && cmp(val, first[next])) {
        first[last] = first[next];
        last = next;
        --next;
    }
    first[last] = val;
}
// Written as ONE loop per element with the two cases of libstdc++'s __insertion_sort (new minimum: shift the whole prefix;
// otherwise: unguarded scan) folded into its condition: on the GPU every lane sorts its own read's table, and lanes that sit
// in different loops of a nest run one after the other.
template <class T, class Cmp>
LM_HD void ss_insertion_sort(T* a, int first, int last, Cmp& cmp) {
    if (first == last) return;
    LM_NOUNROLL
    for (int i = first + 1; i != last; ++i) {
        const T val = a[i];
        const bool to_front = cmp(val, a[first]);
        int j = i;
        LM_NOUNROLL
        while (j > first && (to_front || cmp(val, a[j - 1]))) {  // a[first] stops the unguarded scan in the reference
            a[j] = a[j - 1];
            --j;
        }
        a[j] = val;
    }
}
template <class T, class Cmp>
LM_HD void ss_adjust_heap(T* a, int first, int hole, int len, T value, Cmp& cmp) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (cmp(a[first + child], a[first + child - 1])) child--;
        a[first + hole] = a[first + child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        a[first + hole] = a[first + child - 1];
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && cmp(a[first + parent], value)) {
        a[first + hole] = a[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    a[first + hole] = value;
}
template <class T, class Cmp>
LM_HD void ss_heapsort(T* a, int first, int last, Cmp& cmp) {
    const int len = last - first;
    if (len >= 2) {
        int parent = (len - 2) / 2;
        while (true) {
            T v = a[first + parent];
            ss_adjust_heap(a, first, parent, len, v, cmp);
            if (parent == 0) break;
            parent--;
        }
    }
    for (int l = last; l - first > 1;) {
        --l;
        T v = a[l];
        a[l] = a[first];
        ss_adjust_heap(a, first, 0, l - first, v, cmp);
    }
}
// CAP = capacity of the stack of pending ranges.  Only ranges of more than 16 elements are kept (smaller ones need no
// partitioning), they are disjoint, and each carries a smaller depth budget than the one below it: at most
// min(n / 17, 2 * floor(log2 n) + 1) entries.  CAP <= 8 (n <= 152, the K4 kernels' tables) keeps the stack in registers --
// an indexed private array lives in scratch memory, which a GPU kernel pays for at every dispatch.
template <int CAP = 34, class T, class Cmp>
LM_HD void ss_sort(T* a, int n, Cmp cmp) {
    if (n <= 0) return;
    if (n <= 16) {  // introsort loop is a no-op below the threshold: straight to the final insertion sort
        ss_insertion_sort(a, 0, n, cmp);
        return;
    }
    // introsort loop, recursion on the right part made explicit; entry = first | last << 13 | depth << 26 (n < 8192).
    // One flat loop: each turn a lane either takes one partitioning step of its current range or fetches the next pending
    // range; the partition itself is one loop that makes one comparison per turn (scan up / scan down as a state).
    uint32_t st[CAP];
    int sp = 0;
    auto push = [&](uint32_t v) {
        if (CAP <= 8) {
#pragma unroll
            for (int q = 0; q < CAP; ++q) if (q == sp) st[q] = v;
        } else if (sp < CAP) st[sp] = v;
        ++sp;
    };
    auto pop = [&]() -> uint32_t {
        --sp;
        if (CAP <= 8) {
            uint32_t v = 0;
#pragma unroll
            for (int q = 0; q < CAP; ++q) if (q == sp) v = st[q];
            return v;
        }
        return st[sp];
    };
    int lg = 0;
    for (int t = n; t > 1; t >>= 1) ++lg;
    if (CAP <= 8) {
#pragma unroll
        for (int q = 0; q < CAP; ++q) st[q] = 0;
    }
    int first = 0, last = n, depth = 2 * lg;
    bool busy = true;
    LM_NOUNROLL
    while (busy) {
        if (last - first <= 16) {  // done with this range: the next pending one, if any
            if (sp > 0) {
                const uint32_t top = pop();
                first = (int)(top & 0x1FFFu); last = (int)((top >> 13) & 0x1FFFu); depth = (int)(top >> 26);
            } else busy = false;
        } else if (depth == 0) {
            ss_heapsort(a, first, last, cmp);
            last = first;
        } else {
            --depth;
            // median of (first+1, mid, last-1) moved to first (__move_median_to_first: all three comparisons made up front)
            const int ia = first + 1, ib = first + (last - first) / 2, ic = last - 1;
            const bool ab = cmp(a[ia], a[ib]), bc = cmp(a[ib], a[ic]), ac = cmp(a[ia], a[ic]);
            const int pick = ab ? (bc ? ib : (ac ? ic : ia)) : (ac ? ia : (bc ? ic : ib));
            { T t = a[first]; a[first] = a[pick]; a[pick] = t; }
            // unguarded partition of [first+1, last) around a[first]
            const T pivot = a[first];
            int lo = first + 1, hi = last - 1;
            bool up = true, more = true;  // up: scanning lo upwards; else hi downwards (hi was decremented on entering)
            LM_NOUNROLL
            while (more) {
                const T x = up ? a[lo] : a[hi];
                const bool in = up ? lo < last : hi > first;
                const bool step = in && (up ? cmp(x, pivot) : cmp(pivot, x));
                if (step) { if (up) ++lo; else --hi; }
                else if (up) up = false;
                else if (lo < hi) { const T t = a[lo]; a[lo] = x; a[hi] = t; ++lo; --hi; up = true; }
                else more = false;
            }
            const int cut = lo;
            // reference recurses on [cut,last) first, then loops on [first,cut)
            // (order of processing disjoint ranges does not change the result)
            if (last - cut > 16) push((uint32_t)cut | ((uint32_t)last << 13) | ((uint32_t)depth << 26));
            last = cut;
        }
    }
    ss_insertion_sort(a, 0, 16, cmp);
    LM_NOUNROLL
    for (int i = 16; i < n; ++i) ss_unguarded_linear_insert(a, i, cmp);
}

}  // namespace lmat
