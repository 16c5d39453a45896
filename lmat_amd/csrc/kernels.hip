// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the read-labeling engine.
//
// One wavefront classifies one read end to end (k-mer extraction -> dedupe -> hash probe
// -> taxid registration / lineage closure / counting -> score + LCA decision), keeping
// all per-read state in LDS.  The k-mer database is an open-addressed hash in HBM probed
// 4 lanes per 64-byte bucket (16 B each).  No MFMA: this is a hash/gather path.
//
// Reference semantics restated here (paths relative to the LMAT tree):
//   K1 extract/canonical/dedupe   src/read_label.cpp:943-950, 978-1017
//   K2 lookup                     src/kmerdb/SortedDb.hpp:279-385, TaxNodeStat.hpp:60-74,208-264
//   K3 registration/closure/count src/read_label.cpp:1104-1204, 692-764
//   K4 score + decision           src/read_label.cpp:803-941, 284-419, 225-282
#include <hip/hip_runtime.h>
#include <atomic>
#include <type_traits>
#include "kernels.hpp"

#ifndef LMAT_ABLATE
#define LMAT_ABLATE 0         // -DLMAT_ABLATE=1 (scripts/build_variant.sh): finer LMAT_STOP_AFTER points (40 ..) for the instruction-count ablation
#endif
#define STOP_AT(n, v) do { if (LMAT_ABLATE && A.prm.stop_after == (n)) { if (lane == 0) { emit(250, (v)); } return; } } while (0)
#ifndef LMAT_CLOSURE_FAST
#define LMAT_CLOSURE_FAST 1   // (A/B builds: -DLMAT_CLOSURE_FAST=0)
#endif
#ifndef LMAT_LDS_SHIFT
#define LMAT_LDS_SHIFT 0   // (-DLMAT_LDS_SHIFT=1: the lane shifts of the minimizer window and the repeat filter through LDS instead of DPP.
                           //  Measured in round 4, same box: 24.57 against 24.29 ms per 8 M reads -- 60 vector instructions fewer per read,
                           //  but two more LDS round trips with their waits in the front end of every read.  Off.)
#endif

namespace lmat {

#define WSYNC()                                              \
    do {                                                     \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                     \
    } while (0)

// Inside the classify kernels WSYNC is scope-aware: the class for very long reads keeps its per-read tables in
// global memory, where lanes of the wave order their accesses with a workgroup-scope fence.
template <bool GMEM>
__device__ __forceinline__ void wsync() {
    if (GMEM) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    else __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
static const int kGmemU = 32768;   // k-mer capacity of the global-memory class (reads up to 32787 bp)
static const int kGmemGrid = 2048; // its workgroups (one per-read table set of ~1.2 MB each: 2.4 GB of scratch).  Every step of a read in this class is a
                                   // round trip to memory, so what it needs is waves: 128 of them (rounds 1-3) were 8 ms per read of the heavy-tail
                                   // workload's superkingdom-wide lists (36 k such reads per 8 M: 2.2 s a launch)

// Pointers that arrive inside a by-value struct are "generic" to the compiler, which then emits flat_load
// (counted on lgkmcnt as well, so every LDS wait also waits for HBM).  The classify kernel therefore
// re-types every device pointer into address space 1 (global) up front.
#define GAS __attribute__((address_space(1)))
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));              // 16-byte table records
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));  // list records are only 4-byte aligned
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#define LAS __attribute__((address_space(3)))
#define G_ADD(p, v) __hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define G_OR(p, v) __hip_atomic_fetch_or((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
__device__ __forceinline__ void store_result(GAS uint64_t* dst, const lmat_read_result& r) {
    static_assert(sizeof(lmat_read_result) == 40, "result record is 5 x 8 bytes");
    uint64_t w[5];
    __builtin_memcpy(w, &r, 40);
    dst[0] = w[0]; dst[1] = w[1]; dst[2] = w[2]; dst[3] = w[3]; dst[4] = w[4];
}

__device__ __forceinline__ uint64_t lt_mask(int lane) { return lane == 0 ? 0ull : (~0ull >> (64 - lane)); }
__device__ __forceinline__ int popc64(uint64_t x) { return __popcll(x); }
// inclusive prefix sum over the 64 lanes in six DPP adds (shifts inside the rows of 16, then the row totals broadcast
// onward); all lanes must be active.  No LDS crossbar as with __shfl_up, no lane arithmetic.
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);  // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);  // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);  // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);  // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1 and 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2 and 3
    return x;
}
// inclusive running maximum over the 64 lanes, same six steps
__device__ __forceinline__ uint32_t wave_scan_max(uint32_t x) {
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false));
    return x;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t x) { return (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl(x), 63); }
// number of set bits of a wave-uniform mask below this lane (v_mbcnt: two instructions)
__device__ __forceinline__ uint32_t prefix_count(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
// bit `lane` of a wave-uniform mask as this lane's predicate: the mask itself, no instruction
__device__ __forceinline__ bool lane_bit(uint64_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
__device__ __forceinline__ uint32_t hash32(uint64_t x) {
    x *= 0x9E3779B97F4A7C15ull;
    return (uint32_t)(x >> 40);
}

// ------------------------------------------------------------------------------------------
// K0: ASCII -> packed records.  Four reads per wave, a lane per code word: 16 bases read as five aligned dwords, funnel-
// shifted to the read's byte offset and translated four at a time inside a register (letter tests as zero-byte tests, the
// 2-bit code from bits 1..2 of the letter).  A base per lane (round 2) was 3 steps of ~40 instructions for a 150 bp read;
// this is one step of ~150 for four reads.  The input buffer is padded by 32 bytes (the dword holding a read's last base).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t zero_bytes(uint32_t z) {  // 0x80 in every byte of z that is zero, 0 elsewhere (exact)
    const uint32_t t = (z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    return ~(t | z | 0x7F7F7F7Fu);
}
__global__ __launch_bounds__(256) void pack_reads_kernel(const uint8_t* __restrict__ bases,
                                                         const uint64_t* __restrict__ off,
                                                         const uint64_t* __restrict__ rec_off,
                                                         uint32_t* __restrict__ words, uint64_t n) {
    const int lane = threadIdx.x & 63;
    const uint32_t sub = (uint32_t)lane & 15u;   // this lane's code word within a stretch of 256 bases
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t nw = (uint64_t)gridDim.x * (blockDim.x >> 6);
    for (uint64_t r0 = wave * 4; r0 < n; r0 += nw * 4) {
        const uint64_t r = r0 + ((uint32_t)lane >> 4);
        const bool have = r < n;
        const uint64_t b0 = have ? off[r] : 0;
        const uint32_t len = have ? (uint32_t)(off[r + 1] - b0) : 0u;
        uint32_t* rec = words + (have ? rec_off[r] : 0);
        if (have && sub == 0) rec[0] = len;
        const uint32_t nb = (len + 15) / 16;
        for (uint32_t p0 = 0; __ballot(p0 < len) != 0; p0 += 256) {
            const uint32_t p = p0 + 16u * sub;
            uint32_t code = 0, valid16 = 0;
            if (p < len) {
                const uint64_t addr = b0 + p, a4 = addr & ~3ull, end = b0 + len;
                const uint32_t sh = (uint32_t)(addr & 3) * 8u;
                const uint32_t* src = (const uint32_t*)(bases + a4);
                uint32_t d[5];
#pragma unroll
                for (int q = 0; q < 5; ++q) d[q] = a4 + 4u * q < end ? src[q] : 0u;  // only dwords that hold a base of this read
                const uint32_t left = len - p;  // bases from p on (16 or more: the whole word)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint32_t w = __builtin_amdgcn_alignbit(d[q + 1], d[q], sh);
                    if (left < 4u * q + 4u) w &= left > 4u * q ? (1u << (8u * (left - 4u * q))) - 1u : 0u;  // bytes past the read's end
                    const uint32_t x = w & 0xDFDFDFDFu;  // ENCODE is case-insensitive (read_label.cpp:943-950)
                    const uint32_t vm = zero_bytes(x ^ 0x41414141u) | zero_bytes(x ^ 0x43434343u) | zero_bytes(x ^ 0x47474747u) | zero_bytes(x ^ 0x54545454u);
                    const uint32_t v1 = vm >> 7;  // 1 per valid byte
                    // A 0x41 -> 0, C 0x43 -> 1, G 0x47 -> 2, T 0x54 -> 3: bits 1..2 of the letter, the upper one folded onto the lower
                    const uint32_t c2 = (((x >> 1) & 0x03030303u) ^ ((x >> 2) & 0x01010101u)) & (v1 * 3u);
                    code |= ((c2 | (c2 >> 6) | (c2 >> 12) | (c2 >> 18)) & 0xFFu) << (8 * q);
                    valid16 |= ((v1 | (v1 >> 7) | (v1 >> 14) | (v1 >> 21)) & 0xFu) << (4 * q);
                }
                if ((p >> 4) < nb) rec[1 + (p >> 4)] = code;
            }
            // a validity word covers 32 bases: this lane's 16 bits and its odd neighbour's
            const uint32_t other = (uint32_t)__shfl_xor((int)valid16, 1);
            if (p < len && (sub & 1u) == 0) rec[1 + nb + (p >> 5)] = valid16 | (other << 16);
        }
    }
}

// ------------------------------------------------------------------------------------------
// DB build: insert (k-mer, payload) pairs.  Equal keys keep the smaller payload, which makes
// the synthetic generator deterministic; ingested files never repeat a key.
// ------------------------------------------------------------------------------------------
// wide layout: 8 x (k-mer << 24 | payload) per bucket, linear probing over buckets from b
__device__ __forceinline__ bool wide_insert(uint64_t* slots, uint32_t nbuckets, uint32_t b, uint64_t kmer, uint32_t payload) {
    const uint64_t val = (kmer << kPayloadBits) | payload;
    const uint32_t max_tries = nbuckets < (1u << 14) ? nbuckets : (1u << 14);  // a chain this long means the table is as good as full: fail, do not crawl
    for (uint32_t tries = 0; tries < max_tries; ++tries) {
        unsigned long long* s = (unsigned long long*)(slots + (uint64_t)b * kSlotsPerBucket);
        for (int j = 0; j < kSlotsPerBucket; ++j) {
            unsigned long long old = s[j];
            while (true) {
                if (old == 0) {
                    unsigned long long prev = atomicCAS(&s[j], 0ull, (unsigned long long)val);
                    if (prev == 0) return true;
                    old = prev;
                    continue;
                }
                if ((old >> kPayloadBits) == kmer) {
                    if (old <= val) return true;
                    unsigned long long prev = atomicCAS(&s[j], old, (unsigned long long)val);
                    if (prev == old) return true;
                    old = prev;
                    continue;
                }
                break;
            }
        }
        b = b + 1 == nbuckets ? 0 : b + 1;
    }
    return false;
}
__device__ __forceinline__ uint32_t wide_lookup(const uint64_t* __restrict__ slots, uint32_t nbuckets, uint32_t b, uint64_t kmer) {
    for (uint32_t tries = 0; tries < nbuckets; ++tries) {
        const uint64_t* s = slots + (uint64_t)b * kSlotsPerBucket;
        bool empty = false;
        for (int j = 0; j < kSlotsPerBucket; ++j) {
            const uint64_t v = s[j];
            if (v == 0) empty = true;
            else if ((v >> kPayloadBits) == kmer) return (uint32_t)(v & kPayloadMask);
        }
        if (empty) return 0;
        b = b + 1 == nbuckets ? 0 : b + 1;
    }
    return 0;
}
// where the overflow table of a compact table files a k-mer: a cheap function of its compact address
__device__ __forceinline__ uint32_t ovf_bucket_of(uint32_t bucket, uint32_t tag, uint32_t nbuckets) {
    uint32_t h = (bucket ^ (tag << 16) ^ (tag >> 3)) * 0x9E3779B1u;
    h ^= h >> 15;
    return __umulhi(h, nbuckets);
}

// compact layout (lmat_common.hpp).  Only canonical k-mers (kmer <= its reverse complement) live in the buckets: the
// address is a function of the strand pair, so a non-canonical key -- a database may hold one, SortedDb::begin_
// would find it only when asked for exactly that word -- goes to the overflow table, which stores full keys.
__device__ __forceinline__ bool table_insert(const DeviceTables& tb, uint64_t kmer, uint32_t payload) {
    if (!tb.cpt.nb) return wide_insert(tb.slots, tb.nbuckets, bucket_of(kmer, tb.nbuckets), kmer, payload);
    const uint64_t rc = revcomp_fwd(kmer, tb.cpt.k);
    uint32_t b, tag;
    cpt_address(tb.cpt, kmer < rc ? kmer : rc, kmer < rc ? rc : kmer, b, tag);
    uint32_t* bk = (uint32_t*)tb.slots + (uint64_t)b * 16;
    if (kmer <= rc) {
        const uint32_t i = atomicAdd(&bk[15], 1u) & kCptCountMask;
        if (i < (uint32_t)kCptSlots) {
            ((uint16_t*)bk)[12 + i] = (uint16_t)payload;
            ((uint8_t*)bk)[48 + i] = (uint8_t)(payload >> 16);
            ((uint16_t*)bk)[i] = (uint16_t)tag;
            return true;
        }
    }
    atomicOr(&bk[15], kCptOvfFlag);
    return wide_insert(tb.ovf_slots, tb.ovf_nbuckets, ovf_bucket_of(b, tag, tb.ovf_nbuckets), kmer, payload);
}
__device__ __forceinline__ uint32_t table_lookup(const DeviceTables& tb, uint64_t kmer) {
    if (!tb.cpt.nb) return wide_lookup(tb.slots, tb.nbuckets, bucket_of(kmer, tb.nbuckets), kmer);
    const uint64_t rc = revcomp_fwd(kmer, tb.cpt.k);
    uint32_t b, tag;
    cpt_address(tb.cpt, kmer < rc ? kmer : rc, kmer < rc ? rc : kmer, b, tag);
    const uint32_t* bk = (const uint32_t*)tb.slots + (uint64_t)b * 16;
    if (kmer <= rc) {
        for (int i = 0; i < kCptSlots; ++i)
            if (((const uint16_t*)bk)[i] == (uint16_t)tag)
                return (uint32_t)((const uint16_t*)bk)[12 + i] | ((uint32_t)((const uint8_t*)bk)[48 + i] << 16);
    }
    if (!(bk[15] & kCptOvfFlag)) return 0;
    return wide_lookup(tb.ovf_slots, tb.ovf_nbuckets, ovf_bucket_of(b, tag, tb.ovf_nbuckets), kmer);
}

__global__ void insert_pairs_kernel(DeviceTables tb, const uint64_t* __restrict__ kmers,
                                    const uint32_t* __restrict__ payload, uint64_t n, uint32_t* fail) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        if (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;  // a table without room fails once, not once per k-mer after a 16 384-bucket crawl each
        if (!table_insert(tb, kmers[i], payload[i])) atomicAdd(fail, 1u);
    }
}

// After a build of the compact layout, per bucket: two inserts of one key that raced into the same bucket are merged
// (smaller payload wins, as in the wide layout), the insert counter becomes the slot count, and the slots in use are
// summed into *total.  One thread per bucket.
__global__ void cpt_tidy_kernel(DeviceTables tb, unsigned long long* total) {
    uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long local = 0;
    for (; b < tb.cpt.nb; b += stride) {
        uint32_t* bk = (uint32_t*)tb.slots + b * 16;
        const uint32_t hdr = bk[15];
        uint32_t n = hdr & kCptCountMask;
        if (n > (uint32_t)kCptSlots) n = kCptSlots;
        uint16_t* tg = (uint16_t*)bk;
        uint8_t* ph = (uint8_t*)bk + 48;
        bool changed = false;
        for (uint32_t i = 0; i < n; ++i)
            for (uint32_t j = i + 1; j < n;) {
                if (tg[i] != tg[j]) { ++j; continue; }
                const uint32_t pi = (uint32_t)tg[12 + i] | ((uint32_t)ph[i] << 16), pj = (uint32_t)tg[12 + j] | ((uint32_t)ph[j] << 16);
                if (pj < pi) { tg[12 + i] = (uint16_t)pj; ph[i] = (uint8_t)(pj >> 16); }
                --n;  // the last slot moves into the hole
                tg[j] = tg[n]; tg[12 + j] = tg[12 + n]; ph[j] = ph[n];
                tg[n] = 0; tg[12 + n] = 0; ph[n] = 0;
                changed = true;
            }
        if (changed || (hdr & kCptCountMask) != n) bk[15] = (hdr & kCptOvfFlag) | n;
        local += n;
    }
    for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(total, local);
}
// ... and per overflow entry: a key that is also in its home bucket (a second insert of it arrived after the bucket
// filled up) is merged into the bucket copy -- the one lookups find -- and not counted; the others add to *total.
__global__ void cpt_merge_overflow_kernel(DeviceTables tb, unsigned long long* total) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x, nslots = (uint64_t)tb.ovf_nbuckets * kSlotsPerBucket;
    unsigned long long local = 0;
    for (; i < nslots; i += stride) {
        const uint64_t v = tb.ovf_slots[i];
        if (!v) continue;
        const uint64_t kmer = v >> kPayloadBits;
        const uint32_t pay = (uint32_t)(v & kPayloadMask);
        const uint64_t rc = revcomp_fwd(kmer, tb.cpt.k);
        bool merged = false;
        if (kmer <= rc) {
            uint32_t b, tag;
            cpt_address(tb.cpt, kmer, rc, b, tag);
            uint16_t* tg = (uint16_t*)((uint32_t*)tb.slots + (uint64_t)b * 16);
            uint8_t* ph = (uint8_t*)tg + 48;
            for (int s = 0; s < kCptSlots; ++s)
                if (tg[s] == (uint16_t)tag) {
                    const uint32_t ps = (uint32_t)tg[12 + s] | ((uint32_t)ph[s] << 16);
                    if (pay < ps) { tg[12 + s] = (uint16_t)pay; ph[s] = (uint8_t)(pay >> 16); }
                    merged = true;
                }
        }
        local += merged ? 0 : 1;
    }
    for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o);
    if ((threadIdx.x & 63) == 0 && local) { atomicAdd(total, local); atomicAdd(total + 1, local); }  // total[1]: overflow entries alone
}

// ------------------------------------------------------------------------------------------
// Synthetic genomes (bench configs): species ancestor = iid bases from a counter-based PRNG, except
// that the first `blk` bases are a block shared by all species of a genus (copied from a genus ancestor: the source of
// k-mers with many taxids); strain = ancestor with 1% substitutions.  All pure functions of (seed, species / strain,
// position) so host code can reproduce any window.
// ------------------------------------------------------------------------------------------
// which conserved block holds base pos: 0 family, 1 phylum, 2 superkingdom, -1 none (SynthGeo)
__host__ __device__ __forceinline__ int synth_cons_level(const SynthGeo& g, uint64_t pos) {
    if (pos < g.blk || pos >= g.cend[2]) return -1;
    return pos < g.cend[0] ? 0 : (pos < g.cend[1] ? 1 : 2);
}
__host__ __device__ __forceinline__ uint32_t synth_anc_base(const SynthGeo& g, uint32_t species, uint64_t pos) {
    if (pos < g.blk) return (uint32_t)(splitmix(g.seed ^ 0x47454E5553ull ^ ((uint64_t)(species / g.spg + 1) << 40) ^ pos) >> 13) & 3u;
    const int lv = synth_cons_level(g, pos);
    if (lv >= 0) return (uint32_t)(splitmix(g.seed ^ (0x434F4E5300ull + (uint64_t)lv) ^ ((uint64_t)(species / g.csz[lv] + 1) << 40) ^ pos) >> 13) & 3u;
    return (uint32_t)(splitmix(g.seed ^ ((uint64_t)(species + 1) << 40) ^ pos) >> 13) & 3u;
}
__host__ __device__ __forceinline__ bool synth_strain_mut(const SynthGeo& g, uint32_t strain_global, uint64_t pos) {
    if (synth_cons_level(g, pos) >= 0) return false;   // conserved: every strain carries the group's bases
    uint64_t h = splitmix((g.seed * 0x2545F4914F6CDD1Dull) ^ ((uint64_t)(strain_global + 1) << 40) ^ pos);
    return (h % 100) == 0;
}
__host__ __device__ __forceinline__ uint32_t synth_strain_base(const SynthGeo& g, uint32_t species, uint32_t strain_global, uint64_t pos) {
    uint32_t b = synth_anc_base(g, species, pos);
    if (synth_cons_level(g, pos) >= 0) return b;
    uint64_t h = splitmix((g.seed * 0x2545F4914F6CDD1Dull) ^ ((uint64_t)(strain_global + 1) << 40) ^ pos);
    if ((h % 100) == 0) b = (b + 1 + (uint32_t)((h >> 32) % 3)) & 3u;
    return b;
}
// a window that lies inside ONE conserved block: its level, else -1
__host__ __device__ __forceinline__ int synth_cons_window(const SynthGeo& g, uint64_t pos, int k) {
    const int lv = synth_cons_level(g, pos);
    return lv >= 0 && pos + k <= g.cend[lv] ? lv : -1;
}
__host__ __device__ __forceinline__ uint64_t canon_from_fwd(uint64_t fwd, int k) {
    const uint64_t x = revcomp_fwd(fwd, k);
    return fwd < x ? fwd : x;
}

// one thread per (species, window start).  A window inside the genus block belongs to all spg * S strains of the genus:
// the genus' first species handles it with a mask over those strains; any other window to the S strains of its species.
// list_payload: [species << S | mask] for species windows, then at g_off [genus << (spg * S) | mask] for block windows.
// rep > 1: a list payload points at one of rep copies of the arena (rep_stride payload units apart), picked by the k-mer.
__global__ void synth_db_kernel(DeviceTables tb, SynthGeo g, int k, const uint16_t* __restrict__ strain_idx,
                                const uint32_t* __restrict__ list_payload, uint64_t g_off, uint32_t rep, uint32_t rep_stride,
                                uint32_t* fail, unsigned long long* inserted) {
    const uint64_t npos = g.G - k + 1;
    const uint64_t total = (uint64_t)g.n_species * npos;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long local = 0;
    for (; i < total; i += stride) {
        if (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;  // no room left: stop, do not crawl (see cpt_displaced_share)
        const uint32_t sp = (uint32_t)(i / npos);
        const uint64_t pos = i % npos;
        const bool inblk = pos + k <= g.blk;
        if (inblk && sp % g.spg != 0) continue;
        uint64_t anc = 0;
        for (int j = 0; j < k; ++j) anc = (anc << 2) | synth_anc_base(g, sp, pos + j);
        const int clv = synth_cons_window(g, pos, k);
        if (clv >= 0) {  // conserved: one k-mer for the whole group, filed by its first species under the group's list
            if (sp % g.csz[clv] != 0) continue;
            if (!table_insert(tb, canon_from_fwd(anc, k), list_payload[g.coff[clv] + sp / g.csz[clv]])) atomicAdd(fail, 1u);
            ++local;
            continue;
        }
        const uint32_t ns = inblk ? g.spg * g.S : g.S;   // strains that carry this window; they are numbered from sp * S
        uint32_t mask = 0;
        for (uint32_t s = 0; s < ns; ++s) {
            const uint32_t sg = sp * g.S + s;
            bool mut = false;
            for (int j = 0; j < k; ++j) mut |= synth_strain_mut(g, sg, pos + j);
            if (!mut) {
                mask |= 1u << s;
            } else {
                uint64_t f = 0;
                for (int j = 0; j < k; ++j) f = (f << 2) | synth_strain_base(g, sg / g.S, sg, pos + j);
                if (!table_insert(tb, canon_from_fwd(f, k), strain_idx[sg])) atomicAdd(fail, 1u);
                ++local;
            }
        }
        if (mask) {
            const uint64_t km = canon_from_fwd(anc, k);
            uint32_t pay = inblk ? list_payload[g_off + ((uint64_t)(sp / g.spg) << (g.spg * g.S)) + mask]
                                 : list_payload[((uint64_t)sp << g.S) + mask];
            // replica of the list record: one per 512-base stretch of a genome, so neighbouring k-mers still share their list
            if (rep > 1 && pay >= kListBase) pay += (uint32_t)(mix64(((uint64_t)sp << 32) | (pos >> 9)) % rep) * rep_stride;
            if (!table_insert(tb, km, pay)) atomicAdd(fail, 1u);
            ++local;
        }
    }
    atomicAdd(inserted, local);
}

__global__ void count_slots_kernel(const uint64_t* __restrict__ slots, uint64_t nslots, unsigned long long* out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long c = 0;
    for (; i < nslots; i += stride) c += slots[i] != 0;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

// Synthetic reads written directly as packed records.  Fixed record stride so offsets are
// implicit in rec_off.  Mix per SURVEY 8d: 88% genome samples with 1% substitutions, 10%
// iid random, 1% with a single N, 1% low-complexity (25-base period).
__global__ void synth_reads_kernel(uint32_t* words, const uint64_t* __restrict__ rec_off,
                                   const uint32_t* __restrict__ lengths, uint32_t n_lengths, uint64_t n, uint64_t seed, SynthGeo g) {
    const uint32_t n_species = g.n_species, S = g.S;
    const uint64_t G = g.G;
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t nw = (uint64_t)gridDim.x * (blockDim.x >> 6);
    for (uint64_t r = wave; r < n; r += nw) {
        const uint64_t h0 = splitmix(seed ^ (r * 0x9E3779B97F4A7C15ull));
        const uint32_t len = lengths[(uint32_t)((h0 >> 48) % n_lengths)];
        uint32_t* rec = words + rec_off[r];
        if (lane == 0) rec[0] = len;
        const uint32_t kind = (uint32_t)(h0 % 100);  // 0..9 random, 10 low complexity, 11 single N, else genome
        const uint32_t sg = (uint32_t)((h0 >> 8) % ((uint64_t)n_species * S));
        const uint32_t sp = sg / S;
        const uint64_t maxoff = G > len ? G - len : 0;
        const uint64_t goff = (splitmix(h0) >> 4) % (maxoff + 1);
        const bool rc = (h0 >> 40) & 1;
        const uint32_t npos = (uint32_t)((splitmix(h0 ^ 77) >> 7) % len);
        const uint32_t nb = (len + 15) / 16;
        for (uint32_t w0 = 0; w0 < nb; w0 += 64) {
            const uint32_t w = w0 + lane;
            uint32_t code = 0, valid = 0;
            if (w < nb) {
                for (int j = 0; j < 16; ++j) {
                    const uint32_t p = w * 16 + j;
                    if (p >= len) break;
                    uint32_t b;
                    if (kind < 10) {
                        b = (uint32_t)(splitmix(h0 ^ ((uint64_t)p << 20) ^ 0x1234567) >> 9) & 3u;
                    } else if (kind == 10) {
                        b = (uint32_t)(splitmix(h0 ^ ((uint64_t)(p % 25) << 20) ^ 0x7654321) >> 9) & 3u;
                    } else {
                        const uint64_t gp = rc ? goff + (len - 1 - p) : goff + p;
                        b = synth_strain_base(g, sp, sg, gp < G ? gp : G - 1);
                        if (rc) b ^= 3u;
                        const uint64_t e = splitmix(h0 ^ ((uint64_t)p << 24) ^ 0xABCDEF);
                        if ((e % 100) == 0) b = (b + 1 + (uint32_t)((e >> 32) % 3)) & 3u;
                    }
                    bool ok = !(kind == 11 && p == npos);
                    code |= b << (2 * j);
                    valid |= (ok ? 1u : 0u) << j;
                }
                rec[1 + w] = code;
            }
            uint32_t other = __shfl_down(valid, 1);
            if (w < nb && (w & 1) == 0) rec[1 + nb + (w >> 1)] = valid | (((w + 1 < nb) ? other : 0u) << 16);
        }
    }
}

// ------------------------------------------------------------------------------------------
// TaxNodeStat-style lookup for tests and tooling: one thread per k-mer.
// ------------------------------------------------------------------------------------------
__global__ void lookup_kernel(DeviceTables tb, const uint64_t* __restrict__ kmers, uint64_t n, uint32_t* counts,
                              uint32_t* tids, uint32_t stride) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t pay = table_lookup(tb, kmers[i]);
    if (pay == 0) { counts[i] = 0; return; }
    if (pay < kListBase) {
        counts[i] = 1;
        if (stride) tids[i * stride] = tb.tid32[pay];
        return;
    }
    const size_t eoff = LMAT_LIST_OFF(pay, tb.list_shift);
    if (tb.arena[eoff] & 0x8000u) {  // gene database: [0x8000][n][0][id lo, id hi]...
        const uint32_t n = tb.arena[eoff + 1];
        counts[i] = n;
        for (uint32_t j = 0; j < n && j < stride; ++j)
            tids[i * stride + j] = (uint32_t)tb.arena[eoff + kListHdr + 2 * j] | ((uint32_t)tb.arena[eoff + kListHdr + 2 * j + 1] << 16);
        return;
    }
    const uint32_t nk = tb.arena[eoff + 1], nr = tb.arena[eoff + 2];
    counts[i] = nr;
    if (tb.wide) {  // ids and the stored list are (low, high) pairs of 32-bit taxids
        const size_t raw = eoff + kListHdr + 4 * (size_t)nk;
        for (uint32_t j = 0; j < nr && j < stride; ++j) tids[i * stride + j] = (uint32_t)tb.arena[raw + 2 * j] | ((uint32_t)tb.arena[raw + 2 * j + 1] << 16);
        return;
    }
    for (uint32_t j = 0; j < nr && j < stride; ++j) tids[i * stride + j] = tb.conv[tb.arena[eoff + kListHdr + 2 * nk + j]];
}

// Where a lookup ends (measurement only): out[0] found in the home bucket, out[1] absent and the bucket never spilled (one
// request), out[2] found in the overflow table, out[3] absent after the overflow table was asked as well; out[4] = overflow
// buckets read in all.  One thread per k-mer; compact layout.
__global__ void probe_stats_kernel(DeviceTables tb, const uint64_t* __restrict__ kmers, uint64_t n, unsigned long long* out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !tb.cpt.nb) return;
    const uint64_t kmer = kmers[i], rc = revcomp_fwd(kmer, tb.cpt.k);
    uint32_t b, tag;
    cpt_address(tb.cpt, kmer < rc ? kmer : rc, kmer < rc ? rc : kmer, b, tag);
    const uint32_t* bk = (const uint32_t*)tb.slots + (uint64_t)b * 16;
    if (kmer <= rc)
        for (int j = 0; j < kCptSlots; ++j)
            if (((const uint16_t*)bk)[j] == (uint16_t)tag) { atomicAdd(&out[0], 1ull); return; }
    if (!(bk[15] & kCptOvfFlag)) { atomicAdd(&out[1], 1ull); return; }
    uint32_t ob = ovf_bucket_of(b, tag, tb.ovf_nbuckets);
    unsigned long long reads = 0;
    for (uint32_t tries = 0; tries < tb.ovf_nbuckets; ++tries) {
        const uint64_t* s_ = tb.ovf_slots + (uint64_t)ob * kSlotsPerBucket;
        ++reads;
        bool empty = false;
        for (int j = 0; j < kSlotsPerBucket; ++j) {
            const uint64_t v = s_[j];
            if (v == 0) empty = true;
            else if ((v >> kPayloadBits) == kmer) { atomicAdd(&out[2], 1ull); atomicAdd(&out[4], reads); return; }
        }
        if (empty) break;
        ob = ob + 1 == tb.ovf_nbuckets ? 0 : ob + 1;
    }
    atomicAdd(&out[3], 1ull);
    atomicAdd(&out[4], reads);
}

// ------------------------------------------------------------------------------------------
// Per-wave LDS layout of the classify kernel.
//   U = capacity in distinct k-mers (a read of length L needs L-k+1 <= U), T = capacity in
//   registered taxids.  Regions are reused across phases (see classify_one).
// ------------------------------------------------------------------------------------------
struct LinEnt {  // candidate-lineage entry (read_label.cpp:225-262, 327-351), 12 bytes
    uint16_t tid;
    uint16_t dep;   // depth; bit 15 = member of no_good (set only after the depth sort; depths stay below 32768)
    uint16_t tin, tout;
    float score;
};
static const uint16_t kLinNoGood = 0x8000;
struct LinEntW {  // the same for a wide taxonomy (32-bit ids and Euler ticks), 20 bytes
    uint32_t tid;
    uint16_t dep, pad_;
    uint32_t tin, tout;
    float score;
};
// K4 in LDS, three table sizes (registered taxids of a read): per-lane block = 6 x u16[T], u8[T], f32[T], LinEnt[LIN], an
// odd number of dwords apart (conflict-free).  The lineage holds the candidates plus the appended ancestors: a read whose
// chain is longer is passed on to the scratch kernel.
static const int kFastE = 256;                       // kept-list elements a read of the fast classes may have (7 B of LDS each)
static const int kK4SmallT = 16;                     // 64 lanes x 500 B: 5 waves per CU
static const int kK4MidT = 32;                       // 64 lanes x 1028 B: 2 waves per CU
template <int TT> struct K4Lds {
    static constexpr int LIN = TT + (TT <= 16 ? 3 : 8);
    static constexpr int LANES = TT <= 32 ? 64 : 32;  // the 64-taxid tier runs half waves: 2 per CU as well
    static constexpr int STRIDE = ((17 * TT + 12 * LIN + 3) / 4) | 1;  // dwords per lane
    static constexpr int BYTES = STRIDE * 4 * LANES;
    static_assert(8 * TT <= 12 * LIN, "sort keys over the lineage entries");
};
static_assert(K4Lds<16>::STRIDE == 125, "small tier: 31.25 KB per wave");

// U = capacity in distinct k-mers (a read of length L needs L-k+1 <= U), T = capacity in registered
// taxids, E = capacity in kept-list elements summed over the read's distinct payloads.
// Regions are reused across phases (see classify_one).
constexpr int pow2_ceil(int x) { int p = 1; while (p < x) p <<= 1; return p; }

template <int U, int T, int E, bool INK4, bool CPT = false, bool WIDE = false>
struct WL {
    static constexpr int IDB = WIDE ? 4 : 2;   // bytes of a taxid / an Euler tick in the per-read tables (WIDE: the wide classes, INK4 only)
    static constexpr int HB = WIDE ? 8 : 4;    // ... and of a taxid-hash entry (id | slot)
    static_assert(!WIDE || INK4, "wide ids run in the classes that keep their tables in memory and decide in-kernel");
    static constexpr int H = pow2_ceil(U * 3 / 2);  // k-mer / payload hash slots
    static constexpr int D = INK4 ? U : 64;  // distinct payloads (taxid lists) per read; the fast classes hand a read with more to a larger class
    static constexpr int TH = 4 * T;   // taxid hash slots: <= T registered + <= T unregistered species keys
    static constexpr int LIN = T + 72; // lineage scratch entries (in-kernel K4 only)
    static constexpr int RD_WORDS = (U + 96) / 16 + (U + 96) / 32 + 4;
    static constexpr int R1_HASH = 8 * H;                              // u64 hv[H]
    // reg, stamp, cnt, leaf (u16) | hent, best (u32) | in-kernel K4: dep, ord, tin, tout (u16), score (f32), sflags (u8),
    // score0, nm_rp (f32), nm_cl (u8)
    static constexpr int R1_TID = (IDB + 6) * T + 2 * HB * TH + (INK4 ? (4 + 2 * IDB) * T + 4 * T + T + 4 * T + 4 * T + T : 0);
    // The k-mer hash of the compact path (reads whose repeat filter fires) is through before the probe writes anything: in the
    // classes that overlay the probe's block on their tables it may run on over R2 / R3 and the payload array behind R1.
    static constexpr bool HASH_OVER = CPT && U <= 512 && !INK4;
    static constexpr int R1 = HASH_OVER ? R1_TID : (R1_HASH > R1_TID ? R1_HASH : R1_TID);
    static constexpr int R2_K = CPT ? 0 : 8 * U + 4 * U;               // ukmer, ubucket (wide layout only)
    static constexpr int R2_D = 4 * D + 2 * D + 2 * D + 2 * D + D;     // dpay, dmult, dn, dstart, dfl
    static constexpr int R2_L = INK4 ? (WIDE ? 20 : 12) * LIN : 0;     // lineage (K4, after the d-arrays die)
    static constexpr int R2 = R2_K > R2_D ? (R2_K > R2_L ? R2_K : R2_L) : (R2_D > R2_L ? R2_D : R2_L);
    static constexpr int R3_P = 4 * U;                                 // upay
    static constexpr int R3_E = (INK4 ? (WIDE ? 24 : 16) : 7) * E;     // element staging (slots map to lanes: the per-id facts stay in registers)
    static constexpr int R3 = R3_P > R3_E ? R3_P : R3_E;
    static constexpr int OFF_RD = 0;
    // The packed record is read by K1 only.  Where a class keeps the k-mers of a read in registers (U <= 512) and addresses
    // the compact table, nothing else is written below OFF_R1 + 2048 before K1 is through, and the tables start at 0.
    static constexpr int OFF_R1 = CPT && U <= 512 ? 0 : ((RD_WORDS * 4 + 15) / 16) * 16;
    static constexpr int OFF_R2 = OFF_R1 + ((R1 + 15) / 16) * 16;
    static constexpr int OFF_R3 = OFF_R2 + ((R2 + 15) / 16) * 16;
    static constexpr int BYTES_BASE = OFF_R3 + ((R3 + 15) / 16) * 16;
    // compact-layout probe: a block of real LDS [bucket stage 2048: 32 buckets per round of loads | bucket of each group 256 |
    // repeat filter 512 | overflow-probe list 1024].  Classes that keep a read's k-mers in registers (U <= 512) overlay it
    // on R1..R3, which are idle until the probe is done; the others get it behind their tables (the global-memory class
    // keeps nothing else in LDS).
    static constexpr int XL_STAGE = 2048, XL_GBKT = XL_STAGE, XL_BLOOM = XL_GBKT + 256, XL_OLIST = XL_BLOOM + 512;
    static constexpr int XL_BYTES = XL_OLIST + 1024;
    static_assert(!(CPT && U <= 512) || RD_WORDS * 4 <= XL_STAGE, "the record lies under the stage, which K1 does not touch");
    static constexpr bool XL_OVERLAY = U <= 512;
    static constexpr int OFF_XL = XL_OVERLAY ? OFF_R1 : BYTES_BASE;
    // payload per k-mer position (compact) / per distinct k-mer (wide): at R3, or behind the block where that overlaps R3
    static constexpr int OFF_UPAY_C = XL_OVERLAY && OFF_R1 + XL_BYTES > OFF_R3 ? OFF_R1 + XL_BYTES : OFF_R3;
    static constexpr int BYTES_C = XL_OVERLAY ? (OFF_UPAY_C + 4 * U > BYTES_BASE ? ((OFF_UPAY_C + 4 * U + 15) / 16) * 16 : BYTES_BASE)
                                              : BYTES_BASE + XL_BYTES;
    static constexpr int BYTES_H = HASH_OVER && OFF_R1 + R1_HASH > BYTES_C ? OFF_R1 + R1_HASH : BYTES_C;
    static constexpr int BYTES = CPT ? BYTES_H : BYTES_BASE;
};

static const uint64_t kEmpty64 = ~0ull;

// min-insert (key << 16 | idx) into a u64 LDS hash; returns the key's slot (stable once claimed).
__device__ __forceinline__ uint32_t lds_min_insert(unsigned long long* hv, int hmask, uint64_t key, uint32_t idx) {
    const unsigned long long v = (key << 16) | idx;
    uint32_t h = hash32(key) & hmask;
    while (true) {
        unsigned long long old = hv[h];
        if (old == kEmpty64 || (old >> 16) == key) {
            if (old != kEmpty64 && old <= v) return h;
            unsigned long long prev = atomicCAS(&hv[h], old, v);
            if (prev == old) return h;
        } else {
            h = (h + 1) & hmask;
        }
    }
}
__device__ __forceinline__ uint32_t lds_find(const unsigned long long* hv, int hmask, uint64_t key) {
    uint32_t h = hash32(key) & hmask;
    while ((hv[h] >> 16) != key) h = (h + 1) & hmask;
    return h;
}

// taxid hash entry: low 16 bits = taxid index (0 = empty); high 16 bits = registration slot (< 0x8000),
// 0x8000|lane while a chunk decides who registers it, 0xFFFF = known key, not registered
// taxids are 16 bits wide here: a 24-bit multiply (full rate) mixes them as well as a 32-bit one (quarter rate)
__device__ __forceinline__ uint32_t tid_hash(uint32_t t, int thmask) { return (__umul24(t, 0x9E3779u) >> 10) & thmask; }
__device__ __forceinline__ uint32_t tid_find_or_claim(unsigned int* hent, int thmask, uint32_t t) {
    uint32_t h = tid_hash(t, thmask);
    while (true) {
        unsigned int cur = hent[h];
        if ((cur & 0xFFFFu) == t) return h;
        if (cur == 0) {
            unsigned int prev = atomicCAS(&hent[h], 0u, t | 0xFFFF0000u);
            if (prev == 0 || (prev & 0xFFFFu) == t) return h;
        }
        h = (h + 1) & thmask;
    }
}
__device__ __forceinline__ int tid_find(const unsigned int* hent, int thmask, uint32_t t) {
    uint32_t h = tid_hash(t, thmask);
    while (true) {
        unsigned int cur = hent[h];
        if ((cur & 0xFFFFu) == t) return (int)h;
        if (cur == 0) return -1;
        h = (h + 1) & thmask;
    }
}
__device__ __forceinline__ int tid_slot(const unsigned int* hent, int thmask, uint32_t t) {
    const int h = tid_find(hent, thmask, t);
    if (h < 0) return -1;
    const uint32_t s = hent[h] >> 16;
    return s >= 0x8000u ? -1 : (int)s;
}
// ... and the same hash for the wide classes: 64-bit entries, low 32 bits = taxid index (0 = empty), high 32 bits = registration slot
// (< 0x80000000), 0x80000000 | lane while a chunk decides who registers it, 0xFFFFFFFF = known key, not registered
__device__ __forceinline__ uint32_t tid_hash_w(uint32_t t, int thmask) { return ((t * 0x9E3779B1u) >> 12) & thmask; }
__device__ __forceinline__ uint32_t tid_find_or_claim(unsigned long long* hent, int thmask, uint32_t t) {
    uint32_t h = tid_hash_w(t, thmask);
    while (true) {
        unsigned long long cur = hent[h];
        if ((uint32_t)cur == t) return h;
        if (cur == 0) {
            unsigned long long prev = atomicCAS(&hent[h], 0ull, (unsigned long long)t | 0xFFFFFFFF00000000ull);
            if (prev == 0 || (uint32_t)prev == t) return h;
        }
        h = (h + 1) & thmask;
    }
}
__device__ __forceinline__ int tid_find(const unsigned long long* hent, int thmask, uint32_t t) {
    uint32_t h = tid_hash_w(t, thmask);
    while (true) {
        unsigned long long cur = hent[h];
        if ((uint32_t)cur == t) return (int)h;
        if (cur == 0) return -1;
        h = (h + 1) & thmask;
    }
}
__device__ __forceinline__ int tid_slot(const unsigned long long* hent, int thmask, uint32_t t) {
    const int h = tid_find(hent, thmask, t);
    if (h < 0) return -1;
    const uint32_t s = (uint32_t)(hent[h] >> 32);
    return s >= 0x80000000u ? -1 : (int)s;
}
// u16 counters packed two per dword so LDS atomics can add to them (sums stay below 65536)
__device__ __forceinline__ void add_u16(uint16_t* arr, uint32_t idx, uint32_t v) {
    atomicAdd((unsigned int*)arr + (idx >> 1), (idx & 1) ? (v << 16) : v);
}

// isAncestor (read_label.cpp:138-150) on Euler-tour intervals: a is a proper ancestor of b
__device__ __forceinline__ bool anc_iv(uint32_t tin_a, uint32_t tout_a, uint32_t tin_b, uint32_t tout_b) {
    return tin_a < tin_b && tout_b <= tout_a;
}

struct K4Key { uint32_t lo, hi; };  // sort key of a candidate: hi = score bits, lo = depth << 16 | slot
struct TCmpDev {  // TCmp, read_label.cpp:475-485
    const float* score;
    const uint16_t* dep;
    __device__ bool operator()(uint16_t a, uint16_t b) const {
        const float d = score[a] - score[b];
        if ((double)fabsf(d) < 0.001) return (int)dep[a] < (int)dep[b];
        return score[a] < score[b];
    }
};
struct CmpDepthDev {  // CmpDepth, read_label.cpp:159-167
    template <class LE> __device__ bool operator()(const LE& a, const LE& b) const { return (int)(a.dep & 0x7FFF) > (int)(b.dep & 0x7FFF); }
};

// glibc's logf (sysdeps/ieee754/flt-32/e_logf.c, the ARM optimized-routines algorithm): 16-entry table on the
// top mantissa bits, degree-3 polynomial in double, one final rounding.  read_label scores with
// std::log(float) when null models are on (read_label.cpp:680-690), so the device has to produce glibc's bits;
// checked against the host libm on 3e8 inputs (tests/test_host_logic.py) and through the GPU parity tests.
__device__ __constant__ double kLogfInvC[16] = {
    0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010bp+0, 0x1.3c995b0b80385p+0, 0x1.30d190c8864a5p+0,
    0x1.25e227b0b8eap+0, 0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0, 0x1.0953f419900a7p+0, 0x1p+0,
    0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aap-1, 0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1,
    0x1.767dcf5534862p-1};
__device__ __constant__ double kLogfLogC[16] = {
    -0x1.57bf7808caadep-2, -0x1.2bef0a7c06ddbp-2, -0x1.01eae7f513a67p-2, -0x1.b31d8a68224e9p-3, -0x1.6574f0ac07758p-3,
    -0x1.1aa2bc79c81p-3, -0x1.a4e76ce8c0e5ep-4, -0x1.1973c5a611cccp-4, -0x1.252f438e10c1ep-5, 0x0p+0,
    0x1.aa5aa5df25984p-5, 0x1.c5e53aa362eb4p-4, 0x1.526e57720db08p-3, 0x1.bc2860d22477p-3, 0x1.1058bc8a07ee1p-2,
    0x1.4043057b6ee09p-2};
__device__ __forceinline__ float glibc_logf(float x) {
    uint32_t ix = __float_as_uint(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (ix * 2 == 0) return -__builtin_inff();
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return __builtin_nanf("");
        ix = __float_as_uint(x * 0x1p23f);  // subnormal: normalise
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (tmp >> (23 - 4)) % 16;
    const int k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & 0xff800000u);
    const double invc = kLogfInvC[i], logc = kLogfLogC[i];
    const double z = (double)__uint_as_float(iz);
    const double r = z * invc - 1.0;
    const double y0 = logc + (double)k * 0x1.62e42fefa39efp-1;
    const double r2 = r * r;
    double y = 0x1.5575b0be00b6ap-2 * r + -0x1.ffffef20a4123p-2;
    y = -0x1.00ea348b88334p-2 * r2 + y;
    y = y * r2 + (y0 + r);
    return (float)y;
}

// which null-model table a read with `cand` distinct k-mers uses: closest / getReadLen (read_label.cpp:107-133),
// then _rand_hits.find (:738-739); -1 = no table, scores stay plain fractions
__device__ __forceinline__ int nm_table_of(const NullModelDev& nm, uint32_t cand) {
    if (!nm.active) return -1;
    const GAS int* len_vec = (const GAS int*)nm.len_vec;
    const GAS int* len_avg = (const GAS int*)nm.len_avg;
    const GAS int* len_table = (const GAS int*)nm.len_table;
    int i = 0;
    for (; i < nm.n_len - 1; ++i)
        if ((int)cand <= len_avg[i]) break;
    const int len = len_vec[i];
    return len > 0 ? len_table[i] : len_table[nm.n_len];
}

// K4 state that lives in lane 0's registers between the two sequential parts
struct K4State {
    float top_score, diff_thresh;
    int lidx, nlin, lowest, highest;
    unsigned highest_depth;
    int plasmid_slot;  // slot of the saved top-hit plasmid or -1
    bool done;         // PhiX short-circuit taken
};

// K4 part 1 (lane 0): scores, running sums, PhiX screen, mean/stdev, human bias, TCmp sort, lineage
// building loop of findReadLabelVer2.  read_label.cpp:748-764, 803-893, 295-325.
template <int LIN, class TID = uint16_t, class LE = LinEnt>   // (TID: taxids and Euler ticks -- u16, or u32 with LinEntW in the wide classes)
__device__ void k4_part1(const KernelParams& P, lmat_read_result& res, K4State& S, const uint16_t* cnt, float* score, float* score0,
                         const uint16_t* dep, const uint8_t* sflags, const TID* tin, const TID* tout,
                         const TID* reg, uint16_t* ord, LE* lin, int nT, uint32_t cand, bool use_nm,
                         const float* nm_rp, const uint8_t* nm_cl, const NullModelDev& ND, K4Key* keys = nullptr,
                         const float* preset_stdev = nullptr) {
    if (preset_stdev) {  // lmat_debug_decide: score[] and the standard deviation are given (a record of a reference run):
                         // straight to the sort and findReadLabelVer2, exactly the code every other read goes through below
        float top = 0.0f;
        for (int s = 0; s < nT; ++s) if (s == 0 || score[s] > top) top = score[s];
        res.cand_kmer_cnt = (uint16_t)cand;
        res.status = LMAT_ST_CALL;
        res.log_avg = 0;
        res.stdev = *preset_stdev;
        S.done = false;
        S.top_score = top;
    }
    bool fnd_phix = false, has_human = false;
    float log_sum = 0.0f, pos_log_sum = 0.0f, top_score = 0.0f, phix_score = 0.0f;
    unsigned sig_hits = 0, pos_sig_hits = 0;
    const float fcand = (float)cand;
    // per class string: running null-model probability (read_label.cpp:746,776-800).  An absent class reads as 0
    // (operator[] default-inserts) and every random_prob is > 0, so "first time: assign" == max with 0.
    float track[64];
    if (use_nm) {
        const GAS uint8_t* cls_rank = (const GAS uint8_t*)ND.cls_rank;
        const GAS uint8_t* lower_cls = (const GAS uint8_t*)ND.lower_cls;
        for (int c = 0; c < ND.n_cls; ++c) track[c] = 0.0f;
        for (int s = 0; s < nT; ++s) {
            const int cl = nm_cl[s];
            const float rp = nm_rp[s];
            track[cl] = track[cl] < rp ? rp : track[cl];              // std::max(random_prob, track[cval])
            for (int ti = (int)cls_rank[cl] - 1; ti >= 0; --ti) {
                const int lc = lower_cls[ti];
                track[cl] = track[cl] < track[lc] ? track[lc] : track[cl];
            }
        }
    }
    for (int s = 0; s < nT && !preset_stdev; ++s) {
        float sc = (float)cnt[s] / fcand;
        if (use_nm) {  // log_odds_score, read_label.cpp:680-690
            const float random_prob = track[nm_cl[s]];
            const float denom = random_prob <= 0 ? 0.00001f : random_prob;
            sc = glibc_logf(sc / denom);
        }
        score[s] = sc;
        if (score0) score0[s] = sc;  // all_cand_set keeps the pre-bias score (:821)
        const uint8_t fl = sflags[s];
        if (fl & kFlagHuman) has_human = true;
        log_sum += sc;
        sig_hits++;
        if (sc > 0) { pos_sig_hits++; pos_log_sum += sc; }
        if (P.screen_phix && (fl & kFlagPhiX)) { phix_score = sc; fnd_phix = true; }
        if (s == 0 || sc > top_score) top_score = sc;
    }
    float stdev1;
    if (!preset_stdev) {
    res.cand_kmer_cnt = (uint16_t)cand;
    S.done = false;
    S.top_score = top_score;
    if (P.screen_phix && phix_score >= top_score && fnd_phix) {  // :841-848
        res.status = LMAT_ST_PHIX;
        res.match_type = LMAT_MT_DIRECT;
        res.call_tid = 32630;
        res.call_score = phix_score;
        S.done = true;
        return;
    }
    unsigned use_sig_hits;
    float log_avg;
    if (pos_sig_hits > 3) { use_sig_hits = pos_sig_hits; log_avg = pos_log_sum / (float)pos_sig_hits; }
    else { use_sig_hits = sig_hits; log_avg = sig_hits > 0 ? log_sum / (float)sig_hits : 0; }
    float log_std = 0;
    for (int s = 0; s < nT; ++s) {
        const float sc = score[s];
        if (sc > 0 && pos_sig_hits > 3) { const float v = log_avg - sc; log_std += (v * v); }
        if (pos_sig_hits <= 3) { const float v = log_avg - sc; log_std += (v * v); }
    }
    stdev1 = use_sig_hits > 1 ? sqrtf(log_std / (float)(use_sig_hits - 1)) : 0;
    res.status = LMAT_ST_CALL;
    res.log_avg = log_avg;
    res.stdev = stdev1;
    if (has_human) {  // :883-891
        for (int s = 0; s < nT; ++s)
            if (sflags[s] & kFlagHuman) score[s] += (P.hbias * stdev1);
    }
    } else {
        stdev1 = *preset_stdev;
        top_score = S.top_score;
    }
    if (P.stop_after == 14) { S.done = true; return; }  // timing experiments: scores and statistics only
    if (keys) {
        // the same sort on packed (score | depth, slot) pairs: a comparison reads two of them instead of two slots, two
        // scores and two depths -- three dependent loads per step of the insertion loops become one
        for (int s = 0; s < nT; ++s) { K4Key kk; kk.hi = __float_as_uint(score[s]); kk.lo = ((uint32_t)dep[s] << 16) | (uint32_t)s; keys[s] = kk; }
        struct TCmpKey {  // TCmp, read_label.cpp:475-485
            __device__ bool operator()(const K4Key& a, const K4Key& b) const {
                const float sa = __uint_as_float(a.hi), sb = __uint_as_float(b.hi);
                const float d = sa - sb;
                if ((double)fabsf(d) < 0.001) return (int)(a.lo >> 16) < (int)(b.lo >> 16);
                return sa < sb;
            }
        };
        ss_sort<(LIN <= 152 ? 8 : 34)>(keys, nT, TCmpKey{});
        for (int s = 0; s < nT; ++s) ord[s] = (uint16_t)keys[s].lo;
    } else {
        for (int s = 0; s < nT; ++s) ord[s] = (uint16_t)s;
        ss_sort<(LIN <= 152 ? 8 : 34)>(ord, nT, TCmpDev{score, dep});  // :892-893
    }
    S.diff_thresh = stdev1 * P.sdiff;       // :895
    if (P.stop_after == 13) { S.done = true; return; }  // ... + the sort
    // findReadLabelVer2 :287-325
    S.plasmid_slot = -1;
    unsigned lowest_depth = 0, highest_depth = 0;
    int lowest = -1, highest = -1, lidx = -1, nlin = 0;
    bool lin_done = false;
    for (int i = nT - 1; i >= 0; --i) {
        const int s = ord[i];
        if (score[s] >= top_score && (sflags[s] & kFlagPlasmid)) S.plasmid_slot = s;
        if (!lin_done) {
            bool add = true;  // addToCandLineage :225-262
            const unsigned cd = dep[s];
            const uint32_t ti = tin[s], to = tout[s];
            for (int j = 0; j < nlin; ++j) {
                const unsigned chk = lin[j].dep;
                if (chk > cd && !anc_iv(ti, to, lin[j].tin, lin[j].tout)) { add = false; break; }
                else if (chk < cd && !anc_iv(lin[j].tin, lin[j].tout, ti, to)) { add = false; break; }
                else if (chk == cd) { add = false; break; }
            }
            if (!add) {
                lidx = i;
                lin_done = true;
            } else {
                LE e;
                e.tid = reg[s]; e.score = score[s]; e.dep = (uint16_t)cd; e.tin = (TID)ti; e.tout = (TID)to;
                lin[nlin++] = e;
                if (cd > lowest_depth || i == nT - 1) { lowest = s; lowest_depth = cd; }
                if (cd < highest_depth || i == nT - 1) { highest = s; highest_depth = cd; }
            }
        }
        if (lin_done && score[s] < top_score) break;
    }
    S.lidx = lidx; S.nlin = nlin; S.lowest = lowest; S.highest = highest; S.highest_depth = highest_depth;
}

// K4 part 2 (lane 0): lineage sort, competitor scan, call, candidate list.  read_label.cpp:344-419, 898-937.
template <int LIN, class TID = uint16_t, class LE = LinEnt>
__device__ void k4_part2(const KernelParams& P, const GAS uint32_t* tid32, lmat_read_result& res, const K4State& S,
                         const float* score, const TID* tin, const TID* tout, const TID* reg,
                         const uint16_t* ord, LE* lin, int nlin, int nT,
                         bool have_add, uint32_t high_tin, uint32_t high_tout, GAS lmat_cand* cand_out,
                         uint32_t* n_cand_out, uint32_t* call_idx_out) {
    const int nlin_total = nlin;
    if (!P.prn_all && cand_out) {  // without -p, MultiMatch prints cand_lin in list order (:917-927)
        for (int j = 0; j < nlin; ++j) { cand_out[j].tid = tid32[lin[j].tid]; cand_out[j].score = lin[j].score; }
    }
    ss_sort<(LIN <= 152 ? 8 : 34)>(lin, nlin, CmpDepthDev{});  // :344-351
    bool any_no_good = false;
    for (int i = S.lidx; i >= 0; --i) {  // :355-362, cmpCompLineage :264-282
        const int s = ord[i];
        const uint32_t ti = tin[s], to = tout[s];
        if (have_add && anc_iv(ti, to, high_tin, high_tout)) continue;  // member of add_set
        bool keep_going = true;
        const float cs = score[s];
        for (int j = 0; j < nlin; ++j) {
            if (anc_iv(lin[j].tin, lin[j].tout, ti, to)) break;
            const float ls = lin[j].score;
            if (ls != -10000.0f && (ls - cs) > S.diff_thresh) { keep_going = false; break; }
            if ((ls - cs) <= S.diff_thresh) { lin[j].dep |= kLinNoGood; any_no_good = true; }
        }
        if (!keep_going) break;
    }
    uint8_t match = LMAT_MT_NOMATCH;
    uint32_t call_tid = 0, call_tin = 0xFFFF, call_tout = 0xFFFF;
    float call_score = 0;
    if (nlin == 0 && !any_no_good) {
        match = LMAT_MT_NOMATCH;
    } else if (nlin > 0 && !any_no_good) {
        call_tid = reg[S.lowest]; call_tin = tin[S.lowest]; call_tout = tout[S.lowest];
        call_score = score[S.lowest];
        match = LMAT_MT_DIRECT;
    } else {
        float max_val = -10000.0f;
        int root_idx = -1;
        for (int j = 0; j < nlin; ++j) {
            max_val = max_val < lin[j].score ? lin[j].score : max_val;  // std::max(cand, max_val)
            if (!(lin[j].dep & kLinNoGood)) { root_idx = j; break; }
        }
        if (root_idx < 0) {
            match = LMAT_MT_LCA_ERROR;  // construct_labels leaves best_guess at (0,0), :931-936
        } else {
            match = LMAT_MT_MULTI;
            const uint32_t lca = lin[root_idx].tid;
            bool lca_is_cand = false;  // all_cand_set.find(lca_tid) :400
            for (int s = 0; s < nT; ++s) lca_is_cand |= reg[s] == lca;
            if (lca_is_cand) {
                if (max_val < lin[root_idx].score) { match = LMAT_MT_PARTIAL; max_val = lin[root_idx].score; }
            }
            call_tid = lca; call_tin = lin[root_idx].tin; call_tout = lin[root_idx].tout;
            call_score = max_val;
        }
    }
    if (S.plasmid_slot >= 0 && match != LMAT_MT_LCA_ERROR && match != LMAT_MT_NOMATCH &&
        anc_iv(call_tin, call_tout, tin[S.plasmid_slot], tout[S.plasmid_slot]))
        call_tid = reg[S.plasmid_slot];  // :410-416
    res.match_type = match;
    res.call_tid = call_tid ? tid32[call_tid] : 0;
    res.call_score = call_score;
    *call_idx_out = call_tid;
    uint32_t ncand = 0;
    if (P.prn_all) {  // :898-910
        if (cand_out) {
            for (int i = nT - 1; i >= 0; --i) {
                const int s = ord[i];
                if (score[s] >= 0) { cand_out[ncand].tid = tid32[reg[s]]; cand_out[ncand].score = score[s]; ++ncand; }
            }
        }
    } else if ((match == LMAT_MT_MULTI || match == LMAT_MT_PARTIAL) && cand_out) {
        ncand = nlin_total;
    }
    *n_cand_out = ncand;
}

__device__ __forceinline__ void spread_left(uint64_t& lo, uint64_t& hi, int s) {  // (hi:lo) |= (hi:lo) << s, 0 < s < 64
    const uint64_t nh = (hi << s) | (lo >> (64 - s)), nl = lo << s;
    hi |= nh;
    lo |= nl;
}

// ------------------------------------------------------------------------------------------
// K4 on the wave: score + sort + findReadLabelVer2 of one read by the 64 lanes of the wave that classified it, the
// read's taxid table never leaving the wave (no hand-off record, no separate decision kernels).  Lane s = registration
// slot s on entry.  Taken when the scores are plain k-mer fractions cnt / cand with cand <= 999 and no effective human
// bias: then two scores are either equal or more than 0.001 apart, TCmp (read_label.cpp:475-485) is the lexicographic
// order on (count, depth) -- a strict weak order -- and what libstdc++'s std::sort leaves is fixed by its partition
// steps alone:
//   * __unguarded_partition compares every element with the pivot only, so the pairs it swaps are "the i-th element
//     from the left that is not below the pivot with the i-th from the right that is not above it, while they have not
//     crossed": prefix counts over two ballots, one exchange through LDS per partition (ranges of more than 16 only);
//   * the final insertion sort of a strict weak order is the stable sort of what the partitions left: a rank count.
// The sequential float sums of the statistics (read_label.cpp:806-880) keep their order: one wave-uniform add per slot.
// Lineage building, competitor scan and call (:284-419) run with a candidate per lane (in sorted order) against a
// lineage entry per lane (in depth order).  Anything outside these preconditions returns false and the read takes the
// general path (hand-off record -> k4_*_kernel), which restates the reference statement by statement.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rl(uint32_t v, int i) { return (uint32_t)__builtin_amdgcn_readlane((int)v, i); }
// lane `from` (0 .. 63) of v, per lane: the bare ds_bpermute (__shfl adds the sub-group arithmetic of its width argument: two vector instructions)
__device__ __forceinline__ uint32_t lane_of(uint32_t v, uint32_t from) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(from << 2), (int)v); }
__device__ __forceinline__ uint32_t wave_or(uint32_t x) {
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
    return rl(x, 63);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x) {
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false));
    return rl(x, 63);
}
__device__ __forceinline__ uint64_t below_mask(int p) { return p <= 0 ? 0ull : (p >= 64 ? ~0ull : ((1ull << p) - 1)); }  // bits < p
// ... where the range of p is known (the clamps above are four or five scalar instructions a call, and the decision step is scalar-heavy)
__device__ __forceinline__ uint64_t below_mask_0_63(int p) { return (1ull << p) - 1; }    // p in 0 .. 63
__device__ __forceinline__ uint64_t below_mask_1_64(int p) { return ~0ull >> (64 - p); }  // p in 1 .. 64

// a / b for integers 0 <= a < 2^10, 1 <= b < 2^10 held as floats, correctly rounded (the reference divides in IEEE arithmetic,
// read_label.cpp:821): reciprocal to 1 ulp, a first quotient, its residual -- exact: a multiple of the quotient's ulp below 2^12 of
// them -- and one correction.  What is rounded at the end lies within 2^-45 (relative) of a / b, and a quotient of two such
// integers is never nearer than 2^-35 to a rounding boundary: the IEEE result, in 4 instructions instead of the 11 of the
// general sequence (v_div_scale .. v_div_fixup).  Checked over the whole domain by lmat_debug_div_check (tests/test_gpu_decide.py).
// The same holds for ANY normal float a >= 0 over an integer 1 <= b <= 64 (the two averages of the statistics, :806-880): the residual
// is a multiple of the quotient's ulp below 2^9 of them, and a 24-bit significand over such a b is never nearer than 2^-31 to a
// rounding boundary (a midpoint would need a 25-bit quotient times an odd factor of b to fit 24 bits).  That domain cannot be walked
// through; the check draws 2^26 pairs of it.
__device__ __forceinline__ float div_small_ints(float a, float b) {
    const float r = __builtin_amdgcn_rcpf(b);
    const float q0 = a * r;
    const float e = __builtin_fmaf(-b, q0, a);
    return __builtin_fmaf(e, r, q0);
}
__global__ void div_check_kernel(unsigned long long* out) {   // every pair of the domain against the IEEE division
    const uint32_t a = blockIdx.x, b = threadIdx.x + 1u + 256u * blockIdx.y;
    if (b < 1024u) {
        const float fa = (float)a, fb = (float)b;
        float ieee = fa / fb;
        if (__float_as_uint(div_small_ints(fa, fb)) != __float_as_uint(ieee)) atomicAdd(out, 1ull);
        atomicAdd(out + 1, 1ull);
    }
    // sums of scores and of squared deviations over 1 .. 64: 64 drawn floats in [2^-60, 2^7) per thread (zero among them)
    const uint32_t tid = (blockIdx.y * 1024u + blockIdx.x) * 256u + threadIdx.x;
    uint32_t bad = 0;
    for (uint32_t i = 0; i < 64u; ++i) {
        const uint32_t h = hash32(((uint64_t)tid << 8) | i), h2 = hash32(((uint64_t)h << 32) | tid);
        const uint32_t ex = 67u + (h2 >> 8) % 67u;                       // biased exponent 67 .. 133
        const float x = i == 0 ? 0.0f : __uint_as_float((ex << 23) | (h & 0x7FFFFFu));
        const float d = (float)(1u + (h2 & 63u));
        float q = x / d;
        bad += __float_as_uint(div_small_ints(x, d)) != __float_as_uint(q) ? 1u : 0u;
    }
    if (bad) atomicAdd(out + 2, (unsigned long long)bad);
    atomicAdd(out + 3, 64ull);
}
typedef const ClassifyArgs __attribute__((address_space(4))) CArgsK4;
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint64_t bal(bool b) { return __builtin_amdgcn_ballot_w64(b); }
// The vector pipe is what the classify kernel runs out of, so this step is written for few VECTOR instructions: per-slot
// loops read their operands as LDS broadcasts, four per load (a v_readlane is a vector instruction, a ds_read is not),
// wave-uniform integers and lane masks stay on the scalar unit, and nothing branches per lane.
template <int THM>
__device__ __forceinline__ bool k4_wave(CArgsK4* Ap, int lane, uint32_t nT, uint32_t cand, const u32x4 fz, uint32_t my_cnt,
                                        uint32_t my_id, const unsigned int* hent, uint32_t* xch, GAS uint64_t* out,
                                        int valid_kmers, uint32_t len, int bin_sel) {
    const float hbias = Ap->prm.hbias, sdiff = Ap->prm.sdiff;
    const int screen_phix = Ap->prm.screen_phix;
    // what does not depend on the read -- no null model, no debug stop outside 30 .. 34, depths that grow along every branch, candidates
    // only with -p (without it a multi match prints the lineage as built, :917-927: the general path's job; bin/run_rl.sh always
    // passes -p) -- is one bit the launcher worked out (k4_static_of)
    if (!(Ap->prm.k4_static & 1)) return false;
    const int stop = LMAT_ABLATE ? Ap->prm.stop_after : 0;   // 30..34: timing experiments (the step ends early with a placeholder record; ablation builds)
    auto placeholder = [&]() {
        if (lane == 0) {  // (on a zero the compiler cannot see through, as classify_one's emit: constant words would sit in registers across the read loop)
            uint32_t z = 0;
            asm volatile("" : "+v"(z));
            out[0] = (uint64_t)(z + ((uint32_t)LMAT_ST_SILENT | ((uint32_t)LMAT_MT_NOMATCH << 8) | (cand << 16))) | ((uint64_t)(uint32_t)valid_kmers << 32);
            out[1] = (uint64_t)len | ((uint64_t)z << 32);
            out[2] = (uint64_t)z | ((uint64_t)z << 32);
            out[3] = (uint64_t)z | ((uint64_t)z << 32);
            out[4] = (uint64_t)z | ((uint64_t)(uint32_t)bin_sel << 32);
        }
        return true;
    };
    if (stop == 30 || (stop == 34 && nT <= 16)) return placeholder();   // 34: only the reads with more than 16 taxids take the step
    if (cand - 1u > 998u || nT - 1u > 63u) return false;   // cand in 1..999, 1..64 taxids
    const bool act = (uint32_t)lane < nT;
    const uint32_t dep = act ? (fz.w & 0xFFFFu) : 0u, fl = act ? (fz.w >> 16) : 0u;
    if (bal(my_cnt > cand) | bal(dep > 0x7FFFu) | (hbias != 0.0f ? bal((fl & kFlagHuman) != 0) : 0ull)) return false;  // (a bias of 0 adds 0 * stdev: nothing)
    float* xs = (float*)xch;      // [64] scores by slot (stay: a registered ancestor's score is read from here)
    uint32_t* xk = xch + 64;      // [64] sort keys by array position; later the taxid of every sorted position
    uint32_t* xx = xch + 128;     // [128] exchange buffer of a partition step; later the Euler intervals by sorted position
    float* xv = (float*)(xch + 256);  // [64] squared deviations
    uint32_t* xp = xch + 320;     // [64] taxid by sorted position: parked here, fetched where needed (the kernel has 64 registers)
    const float fcand = (float)cand;
    const float sc = div_small_ints((float)my_cnt, fcand);   // cand in 1 .. 999 and counts up to it (checked above); lanes >= nT: 0
    xs[lane] = sc;
    // ---- std::sort(TCmp) (:892-893).  Element = key << 6 | slot, key = count << 15 | depth; lane p = array position p.
    uint32_t e = act ? (((my_cnt << 15) | dep) << 6) | (uint32_t)lane : 0xFFFFFFFFu;
    if (nT > 16) {
        int lg = 0;
        for (uint32_t t = nT; t > 1; t >>= 1) ++lg;
        uint32_t st0 = 0, st1 = 0, st2 = 0;  // pending ranges of more than 16: first | last << 8 | depth << 16 (at most 3 of them fit 64 elements)
        int sp = 0, first = 0, last = (int)nT, depth = 2 * lg;
        bool busy = true;
        while (busy) {
            if (last - first <= 16) {
                if (sp > 0) {
                    --sp;
                    const uint32_t top = sp == 0 ? st0 : (sp == 1 ? st1 : st2);
                    first = (int)(top & 0xFFu); last = (int)((top >> 8) & 0xFFu); depth = (int)(top >> 16);
                } else busy = false;
            } else if (depth == 0) {
                return false;  // heapsort turn of introsort: the general path replays it
            } else {
                --depth;
                const int ia = first + 1, ib = first + (last - first) / 2, ic = last - 1;
                const uint32_t ka = rl(e, ia) >> 6, kb = rl(e, ib) >> 6, kc = rl(e, ic) >> 6;
                const bool ab = ka < kb, bc = kb < kc, ac = ka < kc;
                const int pick = ab ? (bc ? ib : (ac ? ic : ia)) : (ac ? ia : (bc ? ic : ib));
                const uint32_t e_first = rl(e, first), e_pick = rl(e, pick);
                e = lane == first ? e_pick : (lane == pick ? e_first : e);
                const uint32_t K = e_pick >> 6;
                const uint64_t inr = below_mask(last) & ~below_mask(first + 1);
                const uint64_t GE = bal((e >> 6) >= K) & inr, LE = bal((e >> 6) <= K) & inr;  // where the scan up / the scan down stops
                const uint32_t pg = prefix_count(GE), pl = prefix_count(LE);
                const bool le = lane_bit(LE);
                const uint32_t above_le = (uint32_t)popc64(LE) - pl - (le ? 1u : 0u);
                const uint64_t SL = bal(above_le > pg) & GE;   // pair pg: its partner from the right lies above this lane
                const uint64_t SR = bal(pg > above_le) & LE;   // pair above_le: its partner from the left lies below
                int cut;
                if (SL) {
                    const bool swl = lane_bit(SL), swr = lane_bit(SR);
                    if (swl) xx[pg] = e;
                    if (swr) xx[64 + above_le] = e;
                    WSYNC();
                    if (swl) e = xx[64 + pg];
                    if (swr) e = xx[above_le];
                    WSYNC();
                    const int Lm = 63 - __builtin_clzll(SL), Rm = __builtin_ctzll(SR);
                    const uint64_t c = GE & ~below_mask(Lm + 1) & below_mask(Rm);
                    cut = c ? __builtin_ctzll(c) : Rm;
                } else {
                    cut = GE ? __builtin_ctzll(GE) : last;
                }
                if (last - cut > 16) {
                    const uint32_t v = (uint32_t)cut | ((uint32_t)last << 8) | ((uint32_t)depth << 16);
                    if (sp == 0) st0 = v; else if (sp == 1) st1 = v; else st2 = v;
                    ++sp;
                }
                last = cut;
            }
        }
    }
    // final insertion sort == stable sort by key of the array as it stands: rank = elements before this one
    const uint32_t keyr = act ? ((e & ~63u) | (uint32_t)lane) : 0xFFFFFFFFu;
    xk[lane] = keyr;
    WSYNC();
    // ---- statistics (read_label.cpp:803-880): sums in registration order; a zero score adds +0.0, so pos_log_sum == log_sum
    float log_sum = 0.0f;
    uint32_t rank = 0;
    for (uint32_t t = 0; t < nT; t += 4) {   // (four per step: slots past nT hold a zero score and the largest key -- neutral)
        const u32x4 q = *(const u32x4*)((const uint32_t*)xs + t);
        log_sum += __uint_as_float(q.x); log_sum += __uint_as_float(q.y); log_sum += __uint_as_float(q.z); log_sum += __uint_as_float(q.w);
        const u32x4 kq = *(const u32x4*)(xk + t);
        rank += (kq.x < keyr ? 1u : 0u) + (kq.y < keyr ? 1u : 0u) + (kq.z < keyr ? 1u : 0u) + (kq.w < keyr ? 1u : 0u);
    }
    const uint32_t pos_sig_hits = (uint32_t)popc64(bal(my_cnt > 0));
    const uint32_t use_sig_hits = pos_sig_hits > 3 ? pos_sig_hits : nT;
    const float log_avg = div_small_ints(log_sum, (float)use_sig_hits);   // (a sum of up to 64 scores over 1 .. 64: see div_small_ints)
    const float dv = log_avg - sc;
    xv[lane] = act && (pos_sig_hits > 3 ? my_cnt > 0 : true) ? dv * dv : 0.0f;
    // sorted-position space: lane i holds the candidate at position i of the sorted array (ascending; best = nT - 1)
    const uint32_t es = (uint32_t)__builtin_amdgcn_ds_permute((int)(rank << 2), (int)e);  // lanes >= nT rank nT and collide there: unused
    const uint32_t pslot4 = (es & 63u) << 2;
#define PDEP ((es >> 6) & 0x7FFFu)   /* depth of the candidate at this position (recomputed where used: one register fewer) */
    const uint32_t piv = (uint32_t)__builtin_amdgcn_ds_bpermute((int)pslot4, (int)fz.z);
    xp[lane] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)pslot4, (int)my_id);
#define PTID_AT(pos) (xp[pos])   /* uniform position: an LDS broadcast */
    const float pscore = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute((int)pslot4, (int)__float_as_uint(sc)));
    const float top_score = __uint_as_float(rl(__float_as_uint(pscore), (int)nT - 1));  // the largest count sorts last
    // top-scoring plasmids by position (:301-304), taken now: the flags are not needed again
    const uint64_t PLtop = bal(((uint32_t)__builtin_amdgcn_ds_bpermute((int)pslot4, (int)fz.w) >> 16) & kFlagPlasmid) & bal(pscore >= top_score) & below_mask_1_64((int)nT);
    WSYNC();
    float log_std = 0.0f;
    for (uint32_t t = 0; t < nT; t += 4) {
        const f32x4 q = *(const f32x4*)(xv + t);
        log_std += q.x; log_std += q.y; log_std += q.z; log_std += q.w;
    }
    const float stdev1 = use_sig_hits > 1 ? sqrtf(div_small_ints(log_std, (float)(use_sig_hits - 1))) : 0.0f;
    if (stop == 31) { if (stdev1 == -1.0f || top_score == -1.0f || piv == 0xFFFFFFFFu || PLtop == 5) G_OR((GAS uint32_t*)Ap->err, 0u); return placeholder(); }
    lmat_read_result res;
    res.status = LMAT_ST_CALL; res.match_type = LMAT_MT_NOMATCH; res.cand_kmer_cnt = (uint16_t)cand; res.valid_kmers = valid_kmers;
    res.read_len = (int)len; res.log_avg = log_avg; res.stdev = stdev1; res.call_tid = 0; res.call_score = 0; res.cand_off = 0; res.n_cand = 0;
    res.bin_sel = bin_sel;
    uint32_t call_idx = 0;
    const uint64_t PX = screen_phix ? bal((fl & kFlagPhiX) != 0) : 0ull;  // by slot
    bool phix = false;
    if (PX) {  // :841-848: the score of the LAST registered PhiX id against the top score
        const float phix_score = __uint_as_float(rl(__float_as_uint(sc), 63 - __builtin_clzll(PX)));
        if (phix_score >= top_score) {
            phix = true;
            res.status = LMAT_ST_PHIX; res.match_type = LMAT_MT_DIRECT; res.call_tid = 32630; res.call_score = phix_score;
            res.log_avg = 0; res.stdev = 0;
            call_idx = Ap->phix_call_idx;
        }
    }
    if (!phix) {
        const float diff_thresh = stdev1 * sdiff;
        const uint32_t ti = piv & 0xFFFFu, oi = piv >> 16;
        // ---- findReadLabelVer2, lineage building (:295-325): candidates from the best down; the first one that does not fit
        //      the lineage so far (addToCandLineage :225-262: equal depths, or the shallower one not an ancestor of the deeper) ends it.
        //      Every candidate above the first misfit was taken, so the lineage is "all positions above lidx" with
        //      lidx = the highest position that misfits ANY higher one: no sequential walk, one pass over all pairs.  Where every
        //      node lies deeper (in the -e file) than its parent -- checked once per taxonomy, else the general path takes the read --
        //      a candidate fits a member iff the two are related, i.e. their Euler intervals [tin, tout] (tout = the largest tin
        //      below; tins are unique) intersect: max(tin) <= min(tout).  Per member j: a max, a min, a compare, and the highest
        //      misfitting j kept per lane -- operands as LDS broadcasts, nothing on the scalar unit.
        uint32_t* xt = xx;
        uint32_t* xo = xx + 64;
        xt[lane] = act ? ti : 0u;          // spare positions relate to everything
        xo[lane] = act ? oi : 0xFFFFu;
        WSYNC();
        uint32_t mis = 0;  // highest position this candidate does not fit (0: none -- position 0 is above nobody)
        for (uint32_t j = 0; j < nT; j += 4) {   // (four members per step; the spare positions relate to everything)
            const u32x4 tq = *(const u32x4*)(xt + j), oq = *(const u32x4*)(xo + j);
            mis = max(mis, max(ti, tq.x) > min(oi, oq.x) ? j : 0u);
            mis = max(mis, max(ti, tq.y) > min(oi, oq.y) ? j + 1u : 0u);
            mis = max(mis, max(ti, tq.z) > min(oi, oq.z) ? j + 2u : 0u);
            mis = max(mis, max(ti, tq.w) > min(oi, oq.w) ? j + 3u : 0u);
        }
        const uint64_t am_ = below_mask_1_64((int)nT);   // 1 .. 64 taxids (checked at the top)
        const uint64_t F = bal(mis > (uint32_t)lane) & am_;
        const int lidx = F ? 63 - __builtin_clzll(F) : -1;
        const uint64_t ACC = am_ & ~below_mask_0_63(lidx + 1);   // lidx <= nT - 2
        // deepest and shallowest member (:316-321; depths within a lineage are all different)
        const bool accd_ = lane_bit(ACC);
        const uint32_t kmax = wave_max_u32(accd_ ? (PDEP << 6) | (uint32_t)lane : 0u);
        const uint32_t kmin = wave_max_u32(accd_ ? ((0x7FFFu - PDEP) << 6) | (uint32_t)lane : 0u);
        const int low_pos = (int)(kmax & 63u), high_pos = (int)(kmin & 63u);
        const uint32_t high_dep = 0x7FFFu - (kmin >> 6);
        if (stop == 32) { if (lidx == 77 || low_pos == 99 || high_dep == 0xFFFFFFu || high_pos == 99) G_OR((GAS uint32_t*)Ap->err, 0u); return placeholder(); }
        // the top-scoring plasmid the loop meets last (:301-304, 324): it runs on below lidx while the scores stay at the top
        int plasmid_pos = -1;
        if (PLtop) {
            const uint64_t Z = ~bal(pscore >= top_score) & below_mask(lidx + 1);
            const uint64_t seen = Z ? ~below_mask(64 - __builtin_clzll(Z)) : ~0ull;  // above the first position at or below lidx that is under the top score
            const uint64_t c = PLtop & seen;
            if (c) plasmid_pos = __builtin_ctzll(c);
        }
        uint8_t match = LMAT_MT_DIRECT;
        uint32_t call_tid = PTID_AT(low_pos), call_iv = rl(piv, low_pos);  // nothing poisoned: the deepest accepted candidate (:366-369)
        float call_score = __uint_as_float(rl(__float_as_uint(pscore), low_pos));
        uint32_t nlin = 0;
        if (lidx >= 0) {
            // ---- the lineage (:326-351): the accepted candidates and, when the shallowest of them is not at depth 0, its ancestors
            //      (with their own scores where they are registered, -10000 where not, :326-343), sorted by depth (CmpDepth
            //      :159-167).  Members sit in the lanes of their sorted positions, the ancestors in the spare lanes behind position
            //      nT - 1; entry r of the sorted lineage goes to lane r: interval, score, taxid | registered << 16 | present << 17.
            //      (The comparator alone defines the order only when no two depths are equal: then every rank is taken exactly
            //      once, and a rank nobody pushed to reads back 0 from ds_permute.)
            const uint32_t nacc = (uint32_t)popc64(ACC);
            const uint32_t high_iv = rl(piv, high_pos);
            const bool have_add = high_dep != 0;
            bool member = lane_bit(ACC);
            uint32_t m_iv = piv, m_sc = __float_as_uint(pscore), m_tid = xp[lane] | 0x30000u, dkey = PDEP + 1u, nmem = nT;
            nlin = nacc;
            if (have_add) {
                const u32x4 hf = ((const GAS u32x4*)Ap->tb.facts16)[PTID_AT(high_pos)];  // (fetched again rather than held in registers since the registration)
                const uint32_t aoff = hf.x, alen = hf.y & 0xFFFFu;
                if (nT + alen > 64u) return false;  // a lane per candidate and per ancestor: longer chains take the general path
                const bool mine = (uint32_t)lane >= nT && (uint32_t)lane < nT + alen;
                if (mine) {
                    const uint64_t pe = ((const GAS uint64_t*)Ap->tb.paths8)[aoff + ((uint32_t)lane - nT)];
                    const uint32_t a_ = (uint32_t)(pe & 0xFFFFu);
                    const int sl = tid_slot(hent, THM, a_);
                    m_tid = a_ | (sl >= 0 ? 0x30000u : 0x20000u);
                    dkey = ((uint32_t)(pe >> 16) & 0xFFFFu) + 1u;
                    m_iv = (uint32_t)(pe >> 32);
                    m_sc = sl >= 0 ? __float_as_uint(xs[sl]) : __float_as_uint(-10000.0f);  // a registered ancestor's score is its slot's, before any bias (:821)
                    member = true;
                }
                nmem = nT + alen;
                nlin = nacc + alen;
            }
            WSYNC();
            xk[lane] = member ? dkey : 0u;
            WSYNC();
            uint32_t drank = 0;
            for (uint32_t t = 0; t < nmem; t += 4) {   // (lanes that are no member wrote a zero key: neutral)
                const u32x4 q = *(const u32x4*)(xk + t);
                drank += (q.x > dkey ? 1u : 0u) + (q.y > dkey ? 1u : 0u) + (q.z > dkey ? 1u : 0u) + (q.w > dkey ? 1u : 0u);
            }
            const int to_lane = (int)((member ? drank : 63u) << 2);  // (ds_permute wraps modulo 64: the others push to lane 63, an entry only when all 64 are)
            const uint32_t d_iv = (uint32_t)__builtin_amdgcn_ds_permute(to_lane, (int)m_iv);
            const uint32_t d_sc = (uint32_t)__builtin_amdgcn_ds_permute(to_lane, (int)m_sc);
            const uint32_t d_tid = (uint32_t)__builtin_amdgcn_ds_permute(to_lane, (int)m_tid);
            const uint64_t lm = below_mask_1_64((int)nlin);   // at least the accepted candidate, at most 64 (checked above)
            if (bal(!(d_tid & 0x20000u)) & lm) return false;  // two entries of one depth: the general path replays std::sort on them
            // ---- competitors (:355-362, cmpCompLineage :264-282): candidates from lidx down that are not ancestors of the
            //      shallowest lineage member, each against the lineage from its deepest entry up
            uint64_t NG = 0;
            if (lidx >= 0) {
                const uint32_t htin = high_iv & 0xFFFFu, htout = high_iv >> 16;
                const uint64_t comp = below_mask_0_63(lidx + 1) & ~(have_add ? bal(ti < htin) & bal(htout <= oi) : 0ull);  // not the added ancestors themselves (:356)
                uint64_t active = comp, big = 0;
                uint32_t mlo = 0, mhi = 0;
                for (uint32_t j = 0; j < nlin && active; ++j) {
                    const uint32_t ivj = rl(d_iv, (int)j);
                    const float ls = __uint_as_float(rl(d_sc, (int)j));
                    const uint32_t tj = ivj & 0xFFFFu, oj = ivj >> 16;
                    const uint64_t anc = bal(tj < ti) & bal(oi <= oj);  // the lineage entry is an ancestor of the competitor: the walk ends
                    const float d = ls - pscore;
                    const uint64_t isbig = ls != -10000.0f ? bal(d > diff_thresh) : 0ull;
                    const uint64_t go = active & ~anc & ~isbig;
                    const uint64_t mark = go & bal(d <= diff_thresh);
                    const uint32_t bit = 1u << (j & 31u);
                    if (j < 32) mlo |= lane_bit(mark) ? bit : 0u; else mhi |= lane_bit(mark) ? bit : 0u;
                    big |= active & ~anc & isbig;
                    active = go;
                }
                // the first competitor (from lidx down) that meets a lineage score beyond the threshold ends the scan; its own
                // marks up to that entry stand
                const uint64_t counted = big ? comp & ~below_mask_0_63(63 - __builtin_clzll(big)) : comp;
                const bool cn = lane_bit(counted);
                NG = (uint64_t)wave_or(cn ? mlo : 0u) | (nlin > 32 ? (uint64_t)wave_or(cn ? mhi : 0u) << 32 : 0ull);
            }
            if (NG) {  // :370-408
                const uint64_t okm = ~NG & lm;
                if (!okm) {
                    match = LMAT_MT_LCA_ERROR;
                    call_tid = 0; call_iv = 0xFFFFFFFFu; call_score = 0;
                } else {
                    const int root = __builtin_ctzll(okm);
                    // std::max over the entries up to and including the first good one (scores are >= 0 or -10000: ordered like their bits as signed integers)
                    int mv = (int)__float_as_uint(-10000.0f);
                    for (int j = 0; j <= root; ++j) { const int x = (int)rl(d_sc, j); mv = x > mv ? x : mv; }
                    float max_val = __uint_as_float((uint32_t)mv);
                    const uint32_t rt = rl(d_tid, root);
                    const float rs = __uint_as_float(rl(d_sc, root));
                    match = LMAT_MT_MULTI;
                    if ((rt & 0x10000u) && max_val < rs) { match = LMAT_MT_PARTIAL; max_val = rs; }
                    call_tid = rt & 0xFFFFu;
                    call_iv = rl(d_iv, root);
                    call_score = max_val;
                }
            }
        }
        if (stop == 33) { if (call_tid == 0xFFFFFFu && call_score == -5.0f) G_OR((GAS uint32_t*)Ap->err, 0u); return placeholder(); }
        if (plasmid_pos >= 0 && match != LMAT_MT_LCA_ERROR) {  // :410-416
            const uint32_t piv_p = rl(piv, plasmid_pos);
            if ((call_iv & 0xFFFFu) < (piv_p & 0xFFFFu) && (piv_p >> 16) <= (call_iv >> 16)) call_tid = PTID_AT(plasmid_pos);
        }
        const GAS uint32_t* g_tid32 = (const GAS uint32_t*)Ap->tb.tid32;
        res.match_type = match;
        res.call_tid = call_tid ? g_tid32[call_tid] : 0u;
        res.call_score = call_score;
        call_idx = call_tid;
        // ---- candidates (:898-927): with -p all of them, best first; else, for a multi match, the lineage as built
        if (Ap->cands) {
            GAS uint32_t* g_cursor = (GAS uint32_t*)Ap->cursor;
            const uint32_t reserve = nT;   // (-p: every candidate, best first)
            const uint32_t chunk = Ap->cand_chunk;
            uint32_t coff = 0;
            if (lane == 0) {
                if (chunk) {  // from this workgroup's sub-cursor; an exhausted one takes the next chunk off the bump cursor (kernels.hpp)
                    GAS unsigned long long* sub = (GAS unsigned long long*)(g_cursor + kCursorWords) + 8u * (__builtin_amdgcn_workgroup_id_x() & Ap->cand_sub_mask);
                    const unsigned long long v = __hip_atomic_fetch_add(sub, (unsigned long long)reserve, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long nx = v & 0xFFFFFFFFull, en = v >> 32;
                    coff = (uint32_t)nx;
                    if (nx + reserve > en) {
                        // Exactly one read crosses the end of a chunk (next <= end < next + its pairs): that one brings the next chunk and
                        // installs it.  Those that come while it does (next already past the end) take their pairs straight off the bump
                        // cursor -- no second chunk is drawn and dropped (which, eight waves to a sub-cursor, cost a near-full buffer its room).
                        if (nx <= en) {
                            coff = G_ADD(&g_cursor[0], chunk);
                            __hip_atomic_exchange(sub, (unsigned long long)(coff + reserve) | ((unsigned long long)(coff + chunk) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        } else coff = G_ADD(&g_cursor[0], reserve);
                    }
                } else coff = G_ADD(&g_cursor[0], reserve);
            }
            coff = (uint32_t)__builtin_amdgcn_readfirstlane((int)coff);
            GAS uint64_t* co = nullptr;  // lmat_cand {tid, score}
            if ((uint64_t)coff + reserve <= Ap->cand_cap) co = (GAS uint64_t*)((GAS lmat_cand*)Ap->cands + coff);
            else if (lane == 0) G_OR((GAS uint32_t*)Ap->err, (uint32_t)kErrCandOverflow);
            res.cand_off = coff;
            if (co && act) co[nT - 1u - (uint32_t)lane] = (uint64_t)g_tid32[xp[lane]] | ((uint64_t)__float_as_uint(pscore) << 32);
            if (co) res.n_cand = nT;
        }
    }
    if (lane == 0) {
        GAS unsigned long long* tally_count = (GAS unsigned long long*)Ap->counts;
        GAS double* tally_score = (GAS double*)(tally_count + Ap->tb.n_ids);
        GAS unsigned long long* tally_nomatch = (GAS unsigned long long*)(tally_score + Ap->tb.n_ids);
        store_result(out, res);
        if (res.status != LMAT_ST_PHIX && res.match_type == LMAT_MT_NOMATCH) {  // tallies, proc_line :1241-1268
            G_ADD(&tally_nomatch[1], 1ull);
        } else if (res.call_score >= Ap->prm.min_score) {
            G_ADD(&tally_count[call_idx], 1ull);
            G_ADD(&tally_score[call_idx], (double)res.call_score);
        } else if (res.call_score < Ap->prm.min_score) {
            G_ADD(&tally_nomatch[2], 1ull);
        }
    }
    return true;
#undef PDEP
#undef PTID_AT
}

// ------------------------------------------------------------------------------------------
// K4 by rows: the same decision step as k4_wave for reads with at most 16 registered taxids (three in four), FOUR reads per
// wave -- a read per row of 16 lanes, a taxid per lane.  The step is a few hundred instructions whatever the number of lanes
// that take part, so on the classify wave (one read, 64 lanes) it cost a quarter of that kernel; here four reads share every
// instruction.  Reads come from the hand-off records of the classify kernel (k4buf, status 251); everything k4_wave keeps on
// the scalar unit (lane masks, positions, the loop limits of one read) is a per-row value in vector registers here, a "wave
// ballot" is the row's 16 bits of it, a broadcast within the row is a ds_bpermute or an LDS read at a per-row address, and the
// rows of a wave run their loops to the longest one's length with the shorter rows padded by neutral elements.  No partition
// steps: std::sort is a plain insertion sort up to 16 elements, i.e. the stable rank of (count, depth).
// A read outside k4_wave's preconditions goes on the bail list and is decided by the general path (k4_kernel, last launch).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t row_shr_max_i32(uint32_t x) {  // inclusive running maximum (signed) along a row of 16 lanes
    auto mx = [](uint32_t a, uint32_t b) { return (int)a > (int)b ? a : b; };
    x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x111, 0xf, 0xf, false));
    x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x112, 0xf, 0xf, false));
    x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x114, 0xf, 0xf, false));
    x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x118, 0xf, 0xf, false));
    return x;
}
__device__ __forceinline__ uint32_t row_shr_or(uint32_t x) {  // lane 15 of a row ends up with the OR of the row
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);
    return x;
}
__device__ __forceinline__ uint32_t row_shr_max_u32(uint32_t x) {  // lane 15 of a row ends up with the row's maximum (unsigned)
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true));
    return x;
}
__device__ __forceinline__ uint32_t low_bits(int p) { return p <= 0 ? 0u : (p >= 32 ? 0xFFFFFFFFu : ((1u << p) - 1u)); }  // bits < p

__global__ __launch_bounds__(64) void k4_row_kernel(ClassifyArgs A) {
    constexpr int G = 16;
    __shared__ __align__(16) uint32_t sh[6 * 64];
    float* xs = (float*)sh;        // scores by slot
    uint32_t* xk = sh + 64;        // sort keys by slot; later the depth keys of the lineage members
    uint32_t* xt = sh + 128;       // tin by sorted position; later the sorted lineage's intervals
    uint32_t* xo = sh + 192;       // tout by sorted position; later the sorted lineage's scores
    float* xv = (float*)(sh + 256);  // squared deviations
    const int lane = threadIdx.x & 63, gbase = lane & ~(G - 1), li = lane & (G - 1);
    auto rowbits = [&](bool p) -> uint32_t { return (uint32_t)(bal(p) >> gbase) & 0xFFFFu; };          // the row's 16 bits of a ballot
    auto bp = [&](uint32_t x, int idx) -> uint32_t { return (uint32_t)__builtin_amdgcn_ds_bpermute((gbase + idx) << 2, (int)x); };  // x of the row's lane idx
    const GAS uint32_t* g_tid32 = (const GAS uint32_t*)A.tb.tid32;
    const GAS u32x4* g_facts16 = (const GAS u32x4*)A.tb.facts16;
    GAS uint32_t* g_cursor = (GAS uint32_t*)A.cursor;
    GAS unsigned long long* tally_count = (GAS unsigned long long*)A.counts;
    GAS double* tally_score = (GAS double*)(tally_count + A.tb.n_ids);
    GAS unsigned long long* tally_nomatch = (GAS unsigned long long*)(tally_score + A.tb.n_ids);
    const uint64_t n = *(const GAS uint32_t*)(g_cursor + 9);
    const GAS uint32_t* list = (const GAS uint32_t*)A.k4_row;
    const float hbias = A.prm.hbias, sdiff = A.prm.sdiff;
    for (uint64_t base = (uint64_t)blockIdx.x * (64 / G); base < n; base += (uint64_t)gridDim.x * (64 / G)) {
        const uint64_t ri = base + (uint64_t)(lane / G);
        const bool vrow = ri < n;
        const uint64_t it = vrow ? (uint64_t)list[ri] : 0ull;
        const GAS uint32_t* krec = (const GAS uint32_t*)A.k4buf + it * kK4RecWords;
        const uint32_t hdr = vrow ? krec[0] : 0u;
        const uint32_t nT_rec = hdr & 0xFFFFu, cand = hdr >> 16;
        const uint32_t nT = nT_rec > (uint32_t)G ? 0u : nT_rec;  // (a longer table is not this kernel's: bailed out below, computed as empty)
        const bool act = (uint32_t)li < nT;
        const uint32_t w = act ? krec[2 + li] : 0u;
        const uint32_t reg = w & 0xFFFFu, cnt = w >> 16;
        const u32x4 f = g_facts16[reg];
        const uint32_t dep = act ? (f.w & 0xFFFFu) : 0u, fl = act ? (f.w >> 16) : 0u;
        bool bail = vrow && (cand - 1u > 998u || nT_rec - 1u > (uint32_t)(G - 1) ||
                             rowbits(act && (cnt > cand || dep > 0x7FFFu || (hbias != 0.0f && (fl & kFlagHuman)))) != 0u);
        const float fcand = (float)(cand ? cand : 1u);
        const float sc = act ? (float)cnt / fcand : 0.0f;
        const uint32_t e = act ? (((cnt << 15) | dep) << 6) | (uint32_t)li : 0xFFFFFFFFu;  // key << 6 | slot; the slot is the array position
        WSYNC();  // (the previous turn's reads of the arrays are done)
        xs[lane] = sc;
        xk[lane] = e;
        WSYNC();
        const uint32_t nTmax = max(max(rl(nT, 0), rl(nT, 16)), max(rl(nT, 32), rl(nT, 48)));
        // ---- statistics + the rank of the insertion sort, slots of shorter rows padded with +0.0 / the largest key
        float log_sum = 0.0f;
        uint32_t rank = 0;
        for (uint32_t t = 0; t < nTmax; t += 2) {
            const u32x2 q = *(const u32x2*)((const uint32_t*)xs + gbase + t), kq = *(const u32x2*)(xk + gbase + t);
            log_sum += __uint_as_float(q.x); log_sum += __uint_as_float(q.y);
            rank += (kq.x < e ? 1u : 0u) + (kq.y < e ? 1u : 0u);
        }
        const uint32_t pos_sig_hits = (uint32_t)__builtin_popcount(rowbits(cnt > 0));
        const uint32_t use_sig_hits = pos_sig_hits > 3 ? pos_sig_hits : nT;
        const float log_avg = log_sum / (float)(use_sig_hits ? use_sig_hits : 1u);
        const float dv = log_avg - sc;
        xv[lane] = act && (pos_sig_hits > 3 ? cnt > 0 : true) ? dv * dv : 0.0f;
        // sorted-position space: lane i of the row holds the candidate at position i (ascending; best = nT - 1)
        const uint32_t es = (uint32_t)__builtin_amdgcn_ds_permute((gbase + (int)rank) << 2, (int)e);  // spare lanes rank nT: a spare position of their own row
        const int pslot = (int)(es & 63u);
        const uint32_t pdep = (es >> 6) & 0x7FFFu;
        const uint32_t piv = bp(f.z, pslot), pfl = bp(f.w >> 16, pslot), ptid = bp(reg, pslot);
        const float pscore = __uint_as_float(bp(__float_as_uint(sc), pslot));
        const float top_score = __uint_as_float(bp(__float_as_uint(pscore), (int)nT - 1));  // the largest count sorts last
        const uint32_t am = low_bits((int)nT);
        const uint32_t GEm = rowbits(act && pscore >= top_score);
        const uint32_t PLtop = rowbits(act && (pfl & kFlagPlasmid)) & GEm;
        WSYNC();
        float log_std = 0.0f;
        for (uint32_t t = 0; t < nTmax; t += 4) {
            const f32x4 q = *(const f32x4*)(xv + gbase + t);
            log_std += q.x; log_std += q.y; log_std += q.z; log_std += q.w;
        }
        const float stdev1 = use_sig_hits > 1 ? sqrtf(log_std / (float)(use_sig_hits - 1)) : 0.0f;
        const float diff_thresh = stdev1 * sdiff;
        // ---- PhiX (:841-848): the score of the LAST registered PhiX id against the top score
        const uint32_t PX = A.prm.screen_phix ? rowbits(act && (fl & kFlagPhiX)) : 0u;
        const float phix_score = __uint_as_float(bp(__float_as_uint(sc), PX ? 31 - __builtin_clz(PX) : 0));
        const bool phix = PX != 0u && phix_score >= top_score;
        // ---- lineage building (:295-325) as in k4_wave: one pass over all pairs of the row
        const uint32_t ti = piv & 0xFFFFu, oi = piv >> 16;
        xt[lane] = act ? ti : 0u;          // spare positions relate to everything
        xo[lane] = act ? oi : 0xFFFFu;
        WSYNC();
        uint32_t mis = 0;
        for (uint32_t j = 0; j < nTmax; j += 2) {
            const u32x2 tq = *(const u32x2*)(xt + gbase + j), oq = *(const u32x2*)(xo + gbase + j);
            mis = max(mis, max(ti, tq.x) > min(oi, oq.x) ? j : 0u);
            mis = max(mis, max(ti, tq.y) > min(oi, oq.y) ? j + 1u : 0u);
        }
        const uint32_t F = rowbits(mis > (uint32_t)li) & am;
        const int lidx = F ? 31 - __builtin_clz(F) : -1;
        const uint32_t ACC = am & ~low_bits(lidx + 1);
        const bool accd = (ACC >> li) & 1u;
        const uint32_t kmax = bp(row_shr_max_u32(accd ? (pdep << 6) | (uint32_t)li : 0u), 15);
        const uint32_t kmin = bp(row_shr_max_u32(accd ? ((0x7FFFu - pdep) << 6) | (uint32_t)li : 0u), 15);
        const int low_pos = (int)(kmax & 63u);
        const uint32_t high_dep = 0x7FFFu - (kmin >> 6);
        int plasmid_pos = -1;  // the top-scoring plasmid the reference's loop meets last (:301-304, 324)
        {
            const uint32_t Z = ~GEm & low_bits(lidx + 1);
            const uint32_t seen = Z ? ~low_bits(32 - __builtin_clz(Z)) : 0xFFFFFFFFu;
            const uint32_t c = PLtop & seen;
            if (c) plasmid_pos = __builtin_ctz(c);
        }
        uint32_t match = LMAT_MT_DIRECT;
        uint32_t call_tid = bp(ptid, low_pos), call_iv = bp(piv, low_pos);
        float call_score = __uint_as_float(bp(__float_as_uint(pscore), low_pos));
        // ---- competitors (:355-362, :264-282) of the rows that have any
        const bool hascomp = vrow && !phix && lidx >= 0;
        if (bal(hascomp)) {
            // the lineage: accepted candidates + the ancestors of the shallowest one where it is not at depth 0 (:326-343), in the
            // row's spare lanes behind position nT - 1 (a row has 16: a longer lineage is the general path's)
            const uint32_t nacc = (uint32_t)__builtin_popcount(ACC);
            const bool have_add = hascomp && high_dep != 0u;
            const int high_pos = (int)(kmin & 63u);
            const uint32_t high_iv = bp(piv, high_pos);
            bool member = accd;
            uint32_t m_iv = piv, m_sc = __float_as_uint(pscore), m_tid = ptid | 0x30000u, dkey = pdep + 1u, nlin = nacc;
            if (bal(have_add)) {
                const u32x4 hf = g_facts16[have_add ? bp(ptid, high_pos) : 0u];
                const uint32_t aoff = hf.x, alen = have_add ? (hf.y & 0xFFFFu) : 0u;
                const bool fits = nT + alen <= (uint32_t)G;
                bail = bail || (have_add && !fits);
                const bool mine = have_add && fits && (uint32_t)li >= nT && (uint32_t)li < nT + alen;
                uint32_t a_ = 0;
                if (mine) {
                    const uint64_t pe = ((const GAS uint64_t*)A.tb.paths8)[aoff + ((uint32_t)li - nT)];
                    a_ = (uint32_t)(pe & 0xFFFFu);
                    dkey = ((uint32_t)(pe >> 16) & 0xFFFFu) + 1u;
                    m_iv = (uint32_t)(pe >> 32);
                    member = true;
                }
                // a registered ancestor scores as its candidate does (before any bias, :821): look the id up among the row's positions
                uint32_t hit = 0xFFFFFFFFu;
                for (uint32_t t = 0; t < nTmax; ++t) hit = (bp(ptid, (int)t) == a_ && t < nT) ? t : hit;
                const uint32_t hs = bp(__float_as_uint(pscore), hit == 0xFFFFFFFFu ? 0 : (int)hit);
                if (mine) { m_tid = a_ | (hit != 0xFFFFFFFFu ? 0x30000u : 0x20000u); m_sc = hit != 0xFFFFFFFFu ? hs : __float_as_uint(-10000.0f); }
                nlin = nacc + (have_add && fits ? alen : 0u);
            }
            WSYNC();
            xk[lane] = member ? dkey : 0u;
            WSYNC();
            uint32_t drank = 0;
            for (uint32_t t = 0; t < (uint32_t)G; t += 2) {
                const u32x2 q = *(const u32x2*)(xk + gbase + t);
                drank += (q.x > dkey ? 1u : 0u) + (q.y > dkey ? 1u : 0u);
            }
            const int to_lane = (gbase + (int)(member ? drank : 15u)) << 2;  // the others push to the row's last lane, an entry only when all 16 are
            const uint32_t d_iv = (uint32_t)__builtin_amdgcn_ds_permute(to_lane, (int)m_iv);
            const uint32_t d_sc = (uint32_t)__builtin_amdgcn_ds_permute(to_lane, (int)m_sc);
            const uint32_t d_tid = (uint32_t)__builtin_amdgcn_ds_permute(to_lane, (int)m_tid);
            bail = bail || (hascomp && rowbits((uint32_t)li < nlin && !(d_tid & 0x20000u)) != 0u);  // two members of one depth
            WSYNC();
            xt[lane] = d_iv;   // the sorted lineage, entry by entry, for the walks below
            xo[lane] = d_sc;
            WSYNC();
            const bool comp = hascomp && act && li <= lidx &&
                              !(have_add && ti < (high_iv & 0xFFFFu) && (high_iv >> 16) <= oi);  // not the added ancestors themselves (:356)
            bool active = comp, big = false;
            uint32_t marks = 0;
            const uint32_t nl = hascomp ? nlin : 0u;
            const uint32_t nlmax = max(max(rl(nl, 0), rl(nl, 16)), max(rl(nl, 32), rl(nl, 48)));
            for (uint32_t j = 0; j < nlmax && bal(active); ++j) {
                const uint32_t ivj = xt[gbase + j];
                const float ls = __uint_as_float(xo[gbase + j]);
                const bool inr = j < nlin;
                const bool anc = (ivj & 0xFFFFu) < ti && oi <= (ivj >> 16);  // the lineage entry is an ancestor of the competitor: the walk ends
                const float d = ls - pscore;
                const bool isbig = ls != -10000.0f && d > diff_thresh;
                const bool go = active && inr && !anc && !isbig;
                marks |= go && d <= diff_thresh ? 1u << j : 0u;
                big = big || (active && inr && !anc && isbig);
                active = go;
            }
            const uint32_t B = rowbits(big);
            const bool counted = comp && (B ? li >= 31 - __builtin_clz(B) : true);  // the first competitor (from lidx down) beyond the threshold ends the scan
            const uint32_t NG = hascomp ? bp(row_shr_or(counted ? marks : 0u), 15) : 0u;
            if (NG) {  // :370-408
                const uint32_t okm = ~NG & low_bits((int)nlin);
                if (!okm) {
                    match = LMAT_MT_LCA_ERROR;
                    call_tid = 0; call_iv = 0xFFFFFFFFu; call_score = 0;
                } else {
                    const int root = __builtin_ctz(okm);
                    float max_val = __uint_as_float(bp(row_shr_max_i32(d_sc), root));  // std::max over the entries up to the first good one
                    const uint32_t rt = bp(d_tid, root);
                    const float rs = __uint_as_float(bp(d_sc, root));
                    match = LMAT_MT_MULTI;
                    if ((rt & 0x10000u) && max_val < rs) { match = LMAT_MT_PARTIAL; max_val = rs; }
                    call_tid = rt & 0xFFFFu;
                    call_iv = bp(d_iv, root);
                    call_score = max_val;
                }
            }
        }
        if (plasmid_pos >= 0 && match != LMAT_MT_LCA_ERROR) {  // :410-416
            const uint32_t piv_p = bp(piv, plasmid_pos < 0 ? 0 : plasmid_pos);
            if ((call_iv & 0xFFFFu) < (piv_p & 0xFFFFu) && (piv_p >> 16) <= (call_iv >> 16)) call_tid = bp(ptid, plasmid_pos);
        }
        // ---- candidates (-p: all of them, best first), the record, the tallies
        const bool done = vrow && !bail;
        uint32_t coff = 0, ncand = 0;
        if (A.cands && bal(done && !phix)) {
            uint32_t c0 = 0;
            if (done && !phix && li == 0) c0 = G_ADD(&g_cursor[0], nT);
            coff = bp(c0, 0);
            const bool room = (uint64_t)coff + nT <= A.cand_cap;
            if (done && !phix && !room && li == 0) G_OR((GAS uint32_t*)A.err, (uint32_t)kErrCandOverflow);
            if (done && !phix && room && act)
                ((GAS uint64_t*)((GAS lmat_cand*)A.cands + coff))[nT - 1u - (uint32_t)li] = (uint64_t)g_tid32[ptid] | ((uint64_t)__float_as_uint(pscore) << 32);
            if (room) ncand = nT;
        }
        if (vrow && li == 0) {
            GAS uint64_t* out = (GAS uint64_t*)(A.results + it);
            if (bail) {  // the general path takes the read: its record says "pending, small table" again
                *(GAS uint8_t*)out = 254;
                ((GAS uint32_t*)A.k4_bail)[G_ADD(&g_cursor[6], 1u)] = (uint32_t)it;
            } else {
                lmat_read_result res;
                uint64_t wv[5];
                wv[0] = out[0]; wv[1] = out[1]; wv[2] = out[2]; wv[3] = out[3]; wv[4] = out[4];
                __builtin_memcpy(&res, wv, 40);   // valid_kmers, read_len, bin_sel as the classify kernel left them
                res.cand_kmer_cnt = (uint16_t)cand;
                uint32_t call_idx;
                if (phix) {
                    res.status = LMAT_ST_PHIX; res.match_type = LMAT_MT_DIRECT; res.call_tid = 32630; res.call_score = phix_score;
                    res.log_avg = 0; res.stdev = 0; res.cand_off = 0; res.n_cand = 0;
                    call_idx = A.phix_call_idx;
                } else {
                    res.status = LMAT_ST_CALL; res.match_type = (uint8_t)match; res.log_avg = log_avg; res.stdev = stdev1;
                    res.call_tid = call_tid ? g_tid32[call_tid] : 0u; res.call_score = call_score;
                    res.cand_off = coff; res.n_cand = ncand;
                    call_idx = call_tid;
                }
                store_result(out, res);
                if (res.status != LMAT_ST_PHIX && res.match_type == LMAT_MT_NOMATCH) {  // tallies, proc_line :1241-1268
                    G_ADD(&tally_nomatch[1], 1ull);
                } else if (res.call_score >= A.prm.min_score) {
                    G_ADD(&tally_count[call_idx], 1ull);
                    G_ADD(&tally_score[call_idx], (double)res.call_score);
                } else if (res.call_score < A.prm.min_score) {
                    G_ADD(&tally_nomatch[2], 1ull);
                }
            }
        }
    }
}

#define WSYNC_WAVE WSYNC
#pragma push_macro("WSYNC")
#undef WSYNC
#define WSYNC() wsync<(U > 2048)>()
// The launch arguments are read where they are used, from the kernel-argument segment (scalar loads from constant memory),
// through a pointer the compiler is made to forget at every phase boundary (REARGS): kept in scalar registers for the whole
// kernel they did not fit (85 spilled into vector-register lanes, and every spill and refill is a vector instruction).
// ------------------------------------------------------------------------------------------
// Tails.  A 150 bp read has 131 k-mer positions: two chunks of 64 lanes and three positions more, and a third chunk for
// those three costs the classify wave what a full one does (a vector instruction takes its issue slot whatever its lane
// mask; the kernel is bound by exactly that).  So the positions from 128 on are looked up here, LPR lanes per read with
// the tails of 64 / LPR reads side by side in a wave, and the classify wave picks the results up: canonical k-mer, bucket,
// payload, and the three m-mer values the k-mers at 125..127 take their minimizer from.  Reads with 128 < positions <=
// 128 + LPR only (compact layout); everything else runs its chunks as before.
// ------------------------------------------------------------------------------------------
template <int LPR>
__global__ __launch_bounds__(256) void tail_kernel(ClassifyArgs A) {
    const uint64_t gi = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t it = gi / LPR;
    const uint32_t j = (uint32_t)(gi % LPR);
    if (it >= A.count) return;
    const uint64_t r = A.index ? (uint64_t)A.index[it] : A.first + it;
    const DeviceTables& tb = A.tb;
    const int k = tb.k;
    const uint64_t off = A.rec_off[r];
    const uint32_t p = 128u + j;
    const uint32_t wb = (2 * p) >> 5, ws = (2 * p) & 31;
    // the code words are asked for together with the length (the buffer is padded: reading past a short record is harmless, and
    // what lies past this record's code words is replaced by zero below) -- one round trip fewer before the bucket's address
    const uint32_t* codes = A.words + off + 1;
    const uint32_t len = A.words[off], cr0 = codes[wb], cr1 = codes[wb + 1], cr2 = codes[wb + 2];
    if ((int)len < k) return;
    const uint32_t P = len - k + 1;
    if (P <= 128u || P > 128u + LPR) return;
    const uint32_t nb = (len + 15) / 16, nm = (len + 31) / 32;
    const uint32_t* vmask = codes + nb;
    auto cw = [&](uint32_t i) -> uint64_t { const uint32_t v = i == wb ? cr0 : (i == wb + 1 ? cr1 : cr2); return i < nb ? v : 0u; };   // the record's tail reads as zero, as in the classify wave's copy
    auto vw = [&](uint32_t i) -> uint64_t { return i < nm ? vmask[i] : 0u; };
    // the window at p (classify_one's `window`)
    const uint64_t kmask = (1ull << (2 * k)) - 1;
    const uint32_t wmask = (1u << k) - 1;
    const uint32_t mb = p >> 5, ms = p & 31;
    const uint64_t m2 = (vw(mb + 1) << 32) | vw(mb);
    const uint64_t lo = (cw(wb + 1) << 32) | cw(wb);
    uint64_t w = ws ? ((lo >> ws) | (cw(wb + 2) << (64 - ws))) : lo;
    w &= kmask;
    const uint64_t rev = (~w) & kmask;
    uint64_t f = __builtin_bitreverse64(w);
    f = ((f & 0x5555555555555555ull) << 1) | ((f >> 1) & 0x5555555555555555ull);
    f >>= (64 - 2 * k);
    const bool fc = f < rev;
    const uint64_t km = fc ? f : rev, kr = fc ? rev : f;
    const uint64_t ri = r - A.result_base;
    if (j < 3u) {  // positions 128..130 always hold an m-mer of the read (P > 128: the last one starts at P + 2)
        const int cm = tb.cpt.m;
        const uint64_t x = f >> (2 * (kCptW - 1)), xr = rev & ((1ull << (2 * cm)) - 1);
        ((uint64_t*)A.tail_u)[ri * 4 + j] = (cpt_scramble(x < xr ? x : xr, cm) << 4) | (xr < x ? 2u : 0u) | (x == xr ? 1u : 0u);
    }
    u32x4 e = {0u, 0u, 0u, 0u};
    if (p < P) {  // (the bucket is asked for whether or not the window is valid: its address needs the bases only, and the validity words are one more round trip away)
        uint32_t b, tag;
        cpt_address(*(const CptGeom*)&tb.cpt, km, kr, b, tag);
        const uint32_t jj = ((tag - 1u) >> 7) & 3u;  // which m-mer of the canonical k-mer is the minimizer
        const uint32_t* bk = (const uint32_t*)tb.slots + (uint64_t)b * 16;
        const u32x4 q0 = *(const u32x4*)bk, q1 = *(const u32x4*)(bk + 4), q2 = *(const u32x4*)(bk + 8), q3 = *(const u32x4*)(bk + 12);
        const uint32_t tg[6] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y};
        const uint32_t pw[6] = {q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};   // 16-bit payload halves, slots 0..11
        const uint32_t ph[3] = {q3.x, q3.y, q3.z};                      // payload bits 16..23, a byte per slot
        uint32_t pay = 0;
        bool hit = false;
#pragma unroll
        for (int i = kCptSlots - 1; i >= 0; --i) {  // the first matching slot wins, as in the classify wave
            const uint32_t t = (tg[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
            if (t == (tag & 0xFFFFu)) {
                hit = true;
                pay = ((pw[i >> 1] >> (16 * (i & 1))) & 0xFFFFu) | (((ph[i >> 2] >> (8 * (i & 3))) & 0xFFu) << 16);
            }
        }
        const bool valid = ((uint32_t)(m2 >> ms) & wmask) == wmask;
        if (valid) {
            if (!hit && (q3.w & kCptOvfFlag)) pay = wide_lookup(tb.ovf_slots, tb.ovf_nbuckets, ovf_bucket_of(b, tag & 0xFFFFu, tb.ovf_nbuckets), km);
            e = u32x4{(uint32_t)km, (uint32_t)(km >> 32), b, pay | (1u << 24) | ((fc ? jj : (uint32_t)(kCptW - 1) - jj) << 30)};
        }
    }
    ((u32x4*)A.tail16)[ri * LPR + j] = e;
}

typedef const ClassifyArgs __attribute__((address_space(4))) CArgs;
template <int U, int T, int E, bool INK4, bool PERM, bool CPT, bool WIDE = false>
__device__ __forceinline__ void classify_one(CArgs* Ap, uint64_t r, unsigned char* lds, LAS unsigned char* xl, int lane,
                                             const uint32_t* wcur, uint32_t (&nmacc)[2]) {
    using L = WL<U, T, E, INK4, CPT, WIDE>;
    constexpr int THM = L::TH - 1;
    // WIDE (a taxonomy of more than 65534 ids, HostTaxonomy::wide): taxids and Euler ticks are 32 bits wide in the per-read tables,
    // hash entries 64; the classes with in-kernel decision only.  Everything below that says tid_t / ent_t is the same code for both.
    using tid_t = typename std::conditional<WIDE, uint32_t, uint16_t>::type;
    using ent_t = typename std::conditional<WIDE, unsigned long long, unsigned int>::type;
    using lin_t = typename std::conditional<WIDE, LinEntW, LinEnt>::type;
    constexpr int ESH = WIDE ? 32 : 16;                       // where the slot sits in a hash entry
    constexpr ent_t EIDM = WIDE ? (ent_t)0xFFFFFFFFull : (ent_t)0xFFFFu;   // id part of an entry; also "slot = none"
    constexpr uint32_t EPEND = WIDE ? 0x80000000u : 0x8000u;  // slot field while a chunk decides
    // RELANE: values derived from the lane id (LDS addresses, masks) are cheap; re-deriving them per phase keeps the
    // register allocator from carrying (and spilling) them across the probe phase, where 34 VGPRs hold loads in flight.
    // (Without it the fast classes spill vector registers to scratch: 27.6 instead of 20.4 ms per 8 M reads.  A second barrier, on
    // the argument pointer, stood here until the end of round 4: it made every phase boundary reload the pointer from the vector
    // lanes it is spilled to -- 37 vector instructions a read, 1.3 % of the kernel's time.)
#define RELANE() do { asm volatile("" : "+v"(lane)); } while (0)
#pragma push_macro("A")
#pragma push_macro("tb")
#define A (*Ap)
#define tb (Ap->tb)
    RELANE();
    uint32_t* rd = (uint32_t*)(lds + L::OFF_RD);
    // R1, hash phases
    unsigned long long* hv = (unsigned long long*)(lds + L::OFF_R1);
    // R1, taxid phase
    tid_t* reg = (tid_t*)(lds + L::OFF_R1);
    uint16_t* stamp = (uint16_t*)(reg + T);
    uint16_t* cnt = stamp + T;
    uint16_t* leaf = cnt + T;
    ent_t* hent = (ent_t*)(leaf + T);
    ent_t* best = hent + L::TH;
    // in-kernel K4 only (large-capacity kernel)
    uint16_t* dep = (uint16_t*)(best + L::TH);
    uint16_t* ord = dep + T;
    tid_t* tin = (tid_t*)(ord + T);
    tid_t* tout = tin + T;
    float* score = (float*)(tout + T);
    uint8_t* sflags = (uint8_t*)(score + T);
    float* score0 = (float*)(sflags + T);
    float* nm_rp = score0 + T;
    uint8_t* nm_cl = (uint8_t*)(nm_rp + T);
    // R2: k-mers, then per-distinct-payload arrays, then the lineage scratch
    unsigned long long* ukmer = (unsigned long long*)(lds + L::OFF_R2);
    uint32_t* ubucket = (uint32_t*)(lds + L::OFF_R2 + 8 * U);
    uint32_t* dpay = (uint32_t*)(lds + L::OFF_R2);
    uint16_t* dmult = (uint16_t*)(lds + L::OFF_R2 + 4 * L::D);
    uint16_t* dn = dmult + L::D;
    uint16_t* dstart = dn + L::D;
    uint8_t* dfl = (uint8_t*)(dstart + L::D);
    lin_t* lin = (lin_t*)(lds + L::OFF_R2);
    // R3: payload per distinct k-mer, then the staged kept-list elements
    uint32_t* upay = (uint32_t*)(lds + (CPT ? L::OFF_UPAY_C : L::OFF_R3));
    // large classes (16 B per element): poff u32 | t | ta | d | sp | plen u16 | fl u8.
    // T <= 64 (7 B per element): t | ta (later: slot of ta) | item offset u16 | d u8 -- the facts of an id live in the
    // registers of the lane that is its registration slot.
    using eld_t = typename std::conditional<INK4, uint16_t, uint8_t>::type;
    uint32_t* el_poff = (uint32_t*)(lds + L::OFF_R3);
    tid_t* el_t = INK4 ? (tid_t*)(el_poff + E) : (tid_t*)(lds + L::OFF_R3);   // kept id, registration order
    tid_t* el_ta = el_t + E;                     // kept id, ascending order (closure order)
    tid_t* el_sp = el_ta + E;                    // large classes: species_of[ta] (wide layout: the three id arrays first, then the u16 ones)
    eld_t* el_d = WIDE ? (eld_t*)(el_sp + E) : (eld_t*)(el_ta + (INK4 ? E : 2 * E));  // owning distinct-payload index
    uint16_t* el_off = (uint16_t*)(el_ta + E);   // T <= 64: offset of the id's chain in the item list of the closure
    if constexpr (!WIDE) el_sp = (tid_t*)((uint16_t*)el_d + E);
    uint16_t* el_plen = WIDE ? (uint16_t*)el_d + E : (uint16_t*)el_sp + E;   // path_len[ta]
    uint8_t* el_fl = (uint8_t*)(el_plen + E);    // flags[ta]

    const int k = tb.k;
    const GAS uint64_t* g_slots = (const GAS uint64_t*)tb.slots;
    const GAS uint16_t* g_arena = (const GAS uint16_t*)tb.arena;
    const GAS uint32_t* g_tid32 = (const GAS uint32_t*)tb.tid32;
    const GAS uint16_t* g_fdepth = (const GAS uint16_t*)tb.fdepth;
    const GAS uint8_t* g_flags = (const GAS uint8_t*)tb.flags;
    const GAS uint32_t* g_path_off = (const GAS uint32_t*)tb.path_off;
    const GAS uint16_t* g_path_len = (const GAS uint16_t*)tb.path_len;
    const GAS tid_t* g_paths = (const GAS tid_t*)(WIDE ? (const void*)tb.paths32 : (const void*)tb.paths);
    const GAS tid_t* g_tin = (const GAS tid_t*)(WIDE ? (const void*)tb.tin32 : (const void*)tb.tin);
    const GAS tid_t* g_tout = (const GAS tid_t*)(WIDE ? (const void*)tb.tout32 : (const void*)tb.tout);
    const GAS tid_t* g_species_of = (const GAS tid_t*)(WIDE ? (const void*)tb.species_of32 : (const void*)tb.species_of);
    const GAS uint64_t* g_paths8 = (const GAS uint64_t*)tb.paths8;
    const GAS u32x4* g_facts16 = (const GAS u32x4*)tb.facts16;
    GAS uint32_t* g_cursor = (GAS uint32_t*)A.cursor;
    GAS uint32_t* g_err = (GAS uint32_t*)A.err;  // sticky error flags, shared by all launches
    // record words were prefetched by the caller: lane l holds word l + 64*j in wcur[j] (word 0 = length)
    const uint32_t len = (uint32_t)__builtin_amdgcn_readfirstlane((int)wcur[0]);
    // The result record is assembled from wave-uniform scalars at each exit: a struct kept live from here on
    // costs ten VGPRs across the probe phase (they were spilled to scratch, 4.5 KB of HBM writes per read).
    int valid_kmers = 0, bin_sel = 0;
    GAS uint64_t* out = (GAS uint64_t*)(A.results + (r - A.result_base));
    GAS unsigned long long* tally_count = (GAS unsigned long long*)A.counts;
    GAS double* tally_score = (GAS double*)(tally_count + tb.n_ids);
    GAS unsigned long long* tally_nomatch = (GAS unsigned long long*)(tally_score + tb.n_ids);
    auto emit = [&](uint32_t status, uint32_t cand_cnt) {
        // The record's words are built on a zero the compiler cannot see through: a word that is a compile-time constant at a call
        // site (status | match << 8 | 0 << 16, the zero fields) would otherwise be kept in a register of its own across the whole
        // read loop -- six call sites: ten registers of a kernel that has 64 -- and spilled once the decision step moved in.
        uint32_t z = 0;
        asm volatile("" : "+v"(z));
        const uint32_t w0 = z + (status | ((uint32_t)LMAT_MT_NOMATCH << 8) | (cand_cnt << 16));  // status, match_type, cand_kmer_cnt
        out[0] = (uint64_t)w0 | ((uint64_t)(uint32_t)valid_kmers << 32);     // valid_kmers
        out[1] = (uint64_t)len | ((uint64_t)z << 32);                        // read_len, log_avg
        out[2] = (uint64_t)z | ((uint64_t)z << 32);                          // stdev, call_tid
        out[3] = (uint64_t)z | ((uint64_t)z << 32);                          // call_score, cand_off
        out[4] = (uint64_t)z | ((uint64_t)(uint32_t)bin_sel << 32);          // n_cand, bin_sel
    };

    if ((int)len < k) {  // proc_line :1217-1223
        if (lane == 0) { emit(LMAT_ST_SHORT_LEN, 0); nmacc[0]++; }
        return;
    }
    const uint32_t P = len - k + 1;
    if (P < A.p_min || P > A.p_max) return;  // another launch over the same list takes this read
    if (P > (uint32_t)U) {  // the host sizes U from the batch's longest read
        if (lane == 0) {
            emit(255, 0);
            if (A.ovf_list) ((GAS uint32_t*)A.ovf_list)[G_ADD(&g_cursor[A.ovf_slot], 1u)] = (uint32_t)r;  // re-run by a larger class
            else G_OR(g_err, (uint32_t)kErrReadTooLong);
        }
        return;
    }
    // tail mode (tail_kernel): this read's positions from 128 on were looked up beforehand.  Worked out once and carried as one
    // 32-bit scalar (re-deriving it at each of the three places that ask was two scalar loads and a dozen scalar instructions a time).
    constexpr bool TAILOK = CPT && U == 160 && !INK4;
    uint32_t tail_flag = 0;
    if constexpr (TAILOK) {
        const uint32_t lpr = A.tail_lpr;
        tail_flag = (uint32_t)__builtin_amdgcn_readfirstlane((lpr != 0 && P > 128u && P <= 128u + lpr && !A.nm.active) ? 1 : 0);  // (a scalar, not a lane mask re-materialised from its conditions)
    }
    auto tail_mode = [&]() -> bool { return TAILOK && tail_flag != 0; };
    // ---- packed record -> LDS (coalesced), zero tail so windows past the end are invalid
    const uint32_t nb = (len + 15) / 16, nm = (len + 31) / 32;
#pragma unroll
    for (int j = 0; j < (L::RD_WORDS + 1 + 63) / 64; ++j) {
        const uint32_t w = (uint32_t)lane + 64u * j;  // record word index; rd[] starts at word 1
        if (w >= 1 && w - 1 < (uint32_t)L::RD_WORDS) rd[w - 1] = (w - 1 < nb + nm) ? wcur[j] : 0u;
    }
    if (!CPT) { for (int i = lane; i < L::H; i += 64) hv[i] = kEmpty64; }
    else { *(LAS u32x2*)(xl + L::XL_BLOOM + 8 * lane) = u32x2{0u, 0u}; }  // the repeat filter of the compact path
    WSYNC();
    const uint32_t* codes = rd;
    const uint32_t* vmask = rd + nb;
    const uint64_t kmask = (1ull << (2 * k)) - 1;  // k <= 20
    const uint32_t wmask = (1u << k) - 1;

    // canonical k-mer of the window starting at base p (read_label.cpp:992-1009).  Bases are
    // packed little-endian, so the 2k-bit window w has base p in its low bits: the reference's
    // "reverse" is ~w and its "forward" is the pair-reversal of w.  `other` = the strand that is not canonical.
    // windowfr: the two strands as they are (forward, reverse complement); window: ordered (canonical, other)
    auto windowfr = [&](uint32_t p, uint64_t& fwd, uint64_t& rc) -> bool {
        // funnel shifts (v_alignbit_b32: one instruction per output word, shift 0 included) instead of 64-bit shifts and a branch
        const uint32_t mb = p >> 5, ms = p & 31;
        const bool ok = (__builtin_amdgcn_alignbit(vmask[mb + 1], vmask[mb], ms) & wmask) == wmask;
        const uint32_t wb = (2 * p) >> 5, ws = (2 * p) & 31;
        const uint32_t c0 = codes[wb], c1 = codes[wb + 1], c2 = codes[wb + 2];
        uint64_t w = ((uint64_t)__builtin_amdgcn_alignbit(c2, c1, ws) << 32) | __builtin_amdgcn_alignbit(c1, c0, ws);
        w &= kmask;
        rc = (~w) & kmask;
        // pair-reversal: reverse the bits of each word (the words swap places), then swap the two bits of every base --
        // (m & a) | (~m & b) is one v_bfi_b32 -- and bring the 2k bits down from the top
        auto pairswap = [](uint32_t t) -> uint32_t { return (0xAAAAAAAAu & (t << 1)) | (~0xAAAAAAAAu & (t >> 1)); };
        const uint32_t fh = pairswap(__builtin_bitreverse32((uint32_t)w)), fl = pairswap(__builtin_bitreverse32((uint32_t)(w >> 32)));
        const uint32_t sh = 64u - 2u * (uint32_t)k;  // 24 at k = 20; k >= 16 here (the compact path's k is 10..20: see below)
        fwd = (((uint64_t)fh << 32) | fl) >> sh;
        return ok;
    };
    auto window = [&](uint32_t p, uint64_t& canon, uint64_t& other, bool& fwd_canon) -> bool {
        uint64_t f, rev;
        const bool ok = windowfr(p, f, rev);
        canon = f < rev ? f : rev;
        other = f < rev ? rev : f;
        fwd_canon = f < rev;
        return ok;
    };

    // ---- K1 pass 1: valid k-mers and GC accounting; wide layout: first-occurrence hash; compact layout: table
    //      address of every k-mer.  For U <= 512 the canonical k-mers (and hash slots / addresses) stay in registers
    //      for the later passes (chunk loop fully unrolled).
    constexpr int CH = (U + 32 + 63) / 64;   // 64-base chunks of the longest read of this class
    constexpr bool CACHE = U <= 512;
    constexpr int KC = CACHE ? CH : 1;
    uint64_t kreg[KC];
    uint32_t hreg[KC];   // wide: slot in the k-mer hash; compact: bucket
    uint32_t treg[KC];   // compact: tag
    uint64_t okm[KC];
    int gc = 0, tot = 0;
    uint64_t prevV = 0;
#pragma unroll
    for (int c = 0; c < (CACHE ? CH : 1); ++c) {
        if (CACHE) { kreg[c] = 0; hreg[c] = 0; treg[c] = 0; okm[c] = 0; }
    }
    const uint32_t want_gc = (uint32_t)__builtin_amdgcn_readfirstlane((int)A.nm.active);  // (a scalar, not a lane mask)
    // compact layout: the minimizer of a k-mer is the smallest of the 4 m-mers it covers, and neighbouring k-mers
    // share 3 of them, so every lane scrambles ONE m-mer -- the one starting at its own position -- and the window
    // minimum runs over the lanes (cpt_finish).  u = scrambled canonical m-mer << 4 | (reverse strand is the smaller)
    // << 1 | (palindrome); bits 2..3 are left for the m-mer's place in the k-mer.
    const int cm = tb.cpt.m;
    uint64_t ureg[KC];
    uint64_t fcm[KC];  // lanes whose forward strand is the canonical one
#pragma unroll
    for (int c = 0; c < KC; ++c) { ureg[c] = 0; fcm[c] = 0; }
    auto pass1_chunk = [&](uint32_t p0, uint64_t& km_out, uint32_t& h_out, uint32_t& t_out, uint64_t& V_out, uint64_t& u_out,
                           uint64_t& fc_out) {
        const uint32_t p = p0 + lane;
        uint64_t km = 0, kr = 0;
        bool fc = false;
        // compact path: every lane runs its window -- one that starts past the last k-mer runs into the record's zero tail and is
        // invalid by itself, and the m-mers of positions P .. P + 2 are wanted anyway (lanes beyond hold garbage nobody reads)
        bool ok;
        uint64_t f = 0, rv = 0;
        if constexpr (CPT) {
            ok = windowfr(p, f, rv);
            fc = f < rv;
            km = fc ? f : rv;
            if (!CACHE) kr = fc ? rv : f;
        } else ok = p < P && window(p, km, kr, fc);
        const uint64_t V = __ballot(ok);
        valid_kmers += popc64(V);
        uint32_t h = 0, t = 0;
        if (!CPT) { if (ok) h = lds_min_insert(hv, L::H - 1, km, p); }
        else if (CACHE) {
            const uint64_t x = f >> (2 * (kCptW - 1)), xr = rv & ((1ull << (2 * cm)) - 1);  // the m-mer at p and its reverse complement
            u_out = (cpt_scramble(x < xr ? x : xr, cm) << 4) | (xr < x ? 2u : 0u) | (x == xr ? 1u : 0u);
            fc_out = __ballot(fc);
        }
        // bases covered by at least one valid k-mer (read_label.cpp:987-1008): base b is covered
        // iff some window start in [b-k+1, b] is valid.  Only the null-model scores consume the GC decile.
        if (want_gc) {
            uint64_t lo = prevV, hi = V;
            int span = 1;
            while (span * 2 <= k) { spread_left(lo, hi, span); span *= 2; }
            if (k - span > 0) spread_left(lo, hi, k - span);
            bool isgc = false;
            if (p < len) {
                const uint32_t code = (codes[p >> 4] >> (2 * (p & 15))) & 3u;
                isgc = code == 1 || code == 2;
            }
            const bool covered = lane_bit(hi);
            gc += popc64(__ballot(covered && isgc));
            tot += popc64(hi);
            prevV = V;
        }
        km_out = km; h_out = h; t_out = t; V_out = V;
    };
    // bucket and tag of the k-mers of chunk c from the m-mer values of this chunk (u) and the next (un): cpt_address
    // with the minimum taken across lanes.  Among equal m-mers cpt_address takes the first in the CANONICAL k-mer's
    // direction, which is the last in the read's direction when the reverse strand is canonical: bits 2..3 of the
    // compared words count positions in that direction.
    // LSH (the 160-k-mer classes): the m-mer values of the next three positions come out of LDS -- every lane parks its own in an
    // array over the (dead) packed record, then reads at +1, +2, +3 -- instead of three wave_shl steps: a DPP shift with its
    // lane-63 patch is a v_readlane, a v_mov and a v_mov_dpp per half and step, 18 vector instructions a chunk, and the vector
    // pipe is what this kernel runs out of; the LDS pipe has room.  Same for the five wave_shr steps of the repeat filter below.
    constexpr bool LSH = LMAT_LDS_SHIFT && CACHE && CPT && 8 * (64 * CH + 4) <= WL<U, T, E, INK4, CPT>::XL_BLOOM;
    auto cpt_finish = [&](int c, uint64_t km, uint64_t u, uint64_t un, uint64_t fcmask, uint32_t& b_out, uint32_t& t_out) {
        const uint32_t ulo = (uint32_t)u, uhi = (uint32_t)(u >> 32);
        uint32_t u1l, u1h, u2l, u2h, u3l, u3h;
        if constexpr (LSH) {
            const LAS uint64_t* up = (const LAS uint64_t*)xl + c * 64 + lane;
            const uint64_t U1 = up[1], U2 = up[2], U3 = up[3];
            u1l = (uint32_t)U1; u1h = (uint32_t)(U1 >> 32); u2l = (uint32_t)U2; u2h = (uint32_t)(U2 >> 32); u3l = (uint32_t)U3; u3h = (uint32_t)(U3 >> 32);
        } else {
            const int n0l = __builtin_amdgcn_readlane((int)(uint32_t)un, 0), n0h = __builtin_amdgcn_readlane((int)(uint32_t)(un >> 32), 0);
            const int n1l = __builtin_amdgcn_readlane((int)(uint32_t)un, 1), n1h = __builtin_amdgcn_readlane((int)(uint32_t)(un >> 32), 1);
            const int n2l = __builtin_amdgcn_readlane((int)(uint32_t)un, 2), n2h = __builtin_amdgcn_readlane((int)(uint32_t)(un >> 32), 2);
            // wave_shl:1 -- lane i takes lane i + 1, lane 63 the next chunk's value
            u1l = (uint32_t)__builtin_amdgcn_update_dpp(n0l, (int)ulo, 0x130, 0xf, 0xf, false);
            u1h = (uint32_t)__builtin_amdgcn_update_dpp(n0h, (int)uhi, 0x130, 0xf, 0xf, false);
            u2l = (uint32_t)__builtin_amdgcn_update_dpp(n1l, (int)u1l, 0x130, 0xf, 0xf, false);
            u2h = (uint32_t)__builtin_amdgcn_update_dpp(n1h, (int)u1h, 0x130, 0xf, 0xf, false);
            u3l = (uint32_t)__builtin_amdgcn_update_dpp(n2l, (int)u2l, 0x130, 0xf, 0xf, false);
            u3h = (uint32_t)__builtin_amdgcn_update_dpp(n2h, (int)u2h, 0x130, 0xf, 0xf, false);
        }
        const bool fc = lane_bit(fcmask);
        // bits 2..3 (free in u): the m-mer's place in the canonical k-mer's direction, 4 i or 12 - 4 i = 4 i ^ 12 -- one xor-add each
        const uint32_t dirm = fc ? 0u : 12u;
        const uint64_t a0 = ((uint64_t)uhi << 32) | (ulo + dirm), a1 = ((uint64_t)u1h << 32) | (u1l + (dirm ^ 4u));
        const uint64_t a2 = ((uint64_t)u2h << 32) | (u2l + (dirm ^ 8u)), a3 = ((uint64_t)u3h << 32) | (u3l + (dirm ^ 12u));
        const uint64_t m01 = a0 < a1 ? a0 : a1, m23 = a2 < a3 ? a2 : a3;
        const uint64_t best = m01 < m23 ? m01 : m23;
        const uint32_t j = ((uint32_t)best >> 2) & 3u, fl = (uint32_t)best & 3u;
        const uint32_t strand = fc ? fl >> 1 : (fl == 0u ? 1u : 0u);  // the m-mer in the canonical k-mer is the larger of its pair
        const uint64_t sp = cpt_mix(best >> 4, cm);
        const int lb = tb.cpt.lowbits;
        const uint32_t hi = (uint32_t)(sp >> lb), low = (uint32_t)sp & ((1u << lb) - 1);
        // hi / W and hi mod W: a shift and a mask for the usual table sizes (W = 1, 2, 4, 8: 256 ... 32 GiB at k = 20), else the exact
        // double multiply of cpt_address (a 64-bit-float multiply and a 32-bit integer multiply are several issue slots each)
        const int wsh = tb.cpt.wshift;
        uint32_t b, rho;
        if (wsh >= 0) { b = hi >> wsh; rho = hi & ((1u << wsh) - 1u); }
        else if (wsh == -2) {  // fractional width (cpt_geometry): bucket = hi * nb >> 32, rho = floor((hi * nb mod 2^32) / nb) <= 3
            const uint32_t nb32 = (uint32_t)tb.cpt.nb, lowp = hi * nb32;
            const uint64_t nb64 = nb32;
            b = __umulhi(hi, nb32);
            rho = (lowp >= nb32 ? 1u : 0u) + ((uint64_t)lowp >= 2 * nb64 ? 1u : 0u) + ((uint64_t)lowp >= 3 * nb64 ? 1u : 0u);
        }
        else { b = (uint32_t)(((double)hi + 0.5) * tb.cpt.invW); rho = hi - b * tb.cpt.W; }
        const int rs = 2 * (kCptW - 1 - (int)j);
        const uint32_t other = (uint32_t)((km >> (2 * cm + rs)) << rs) | ((uint32_t)km & ((1u << rs) - 1));
        b_out = b;
        t_out = (1u + (((((rho << lb) | low) * 4 + j) * 2 + strand) * 64 + other)) | ((fc ? j : (uint32_t)(kCptW - 1) - j) << 30);
    };
    if (CACHE) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if ((uint32_t)c * 64 >= len) break;
            if (TAILOK && c == 2 && tail_mode()) {  // looked up beforehand: k-mer, bucket, minimizer offset; the payload goes straight to its place
                const uint32_t lpr = A.tail_lpr;
                const uint64_t ri = r - A.result_base;
                u32x4 te = {0u, 0u, 0u, 0u};
                uint64_t tu = 0;
                if ((uint32_t)lane < lpr) te = ((const GAS u32x4*)A.tail16)[ri * lpr + lane];
                if (lane < 3) tu = ((const GAS uint64_t*)A.tail_u)[ri * 4 + lane];
                kreg[c] = ((uint64_t)(te.y & 0xFFu) << 32) | te.x;
                hreg[c] = te.z;
                treg[c] = te.w & 0xC0000000u;
                okm[c] = __ballot((te.w >> 24) & 1u);
                valid_kmers += popc64(okm[c]);
                ureg[c] = tu;
                if (128u + (uint32_t)lane < P) upay[128 + lane] = te.w & 0xFFFFFFu;
                continue;
            }
            pass1_chunk((uint32_t)c * 64, kreg[c], hreg[c], treg[c], okm[c], ureg[c], fcm[c]);
        }
        STOP_AT(47, 0);
        if (CPT) {
            if constexpr (LSH) {  // every lane's m-mer value, parked for its three left neighbours (the packed record under it is through)
                WSYNC();
                LAS uint64_t* ush = (LAS uint64_t*)xl;
#pragma unroll
                for (int c = 0; c < CH; ++c) ush[c * 64 + lane] = ureg[c];
                if (lane < 4) ush[CH * 64 + lane] = 0ull;
                WSYNC();
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if ((uint32_t)c * 64 >= P) break;
                if (TAILOK && c == 2 && tail_mode()) continue;
                cpt_finish(c, kreg[c], ureg[c], c + 1 < CH ? ureg[c + 1] : 0ull, fcm[c], hreg[c], treg[c]);
            }
        }
    } else {
        for (uint32_t p0 = 0; p0 < len; p0 += 64) { uint64_t a; uint32_t b, t; uint64_t v, u, f; pass1_chunk(p0, a, b, t, v, u, f); }
    }
    WSYNC();
    if (tot > 0) {  // :1205-1206
        const float q = (float)gc / (float)tot;
        const float gc_pcnt = (float)((double)q * 100.0);
        bin_sel = __builtin_amdgcn_readfirstlane((int)(gc_pcnt / 10.0f));
    }
    if (valid_kmers < A.prm.min_kmer) {  // proc_line :1232-1238
        if (lane == 0) { emit(LMAT_ST_SHORT_VALID, 0); nmacc[0]++; }
        return;
    }
    STOP_AT(1, 0);
    RELANE();
    uint32_t nuniq = 0;   // distinct valid k-mers of the read
    uint32_t nscan = 0;   // entries of upay[]: one per distinct k-mer (wide) or per k-mer position (compact)
    if constexpr (!CPT) {
    // ---- K1 pass 2: compact first occurrences in position order
    auto pass2_chunk = [&](uint32_t p0, uint64_t km, uint32_t h, bool ok) {
        const uint32_t p = p0 + lane;
        const bool first = ok && (uint32_t)(hv[h] & 0xFFFF) == p;
        const uint64_t bm = __ballot(first);
        if (first) {
            const uint32_t rk = nuniq + prefix_count(bm);
            ukmer[rk] = (km << kPayloadBits) + 1;  // pre-shifted for the probe's slot compare
            ubucket[rk] = bucket_of(km, tb.nbuckets);
            upay[rk] = 0;
        }
        nuniq += popc64(bm);
    };
    if (CACHE) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if ((uint32_t)c * 64 >= P) break;
            pass2_chunk((uint32_t)c * 64, kreg[c], hreg[c], lane_bit(okm[c]));
        }
    } else {
        for (uint32_t p0 = 0; p0 < P; p0 += 64) {
            const uint32_t p = p0 + lane;
            uint64_t km = 0, kr = 0;
            uint32_t h = 0;
            bool fc;
            const bool ok = p < P && window(p, km, kr, fc);
            if (ok) h = lds_find(hv, L::H - 1, km);
            pass2_chunk(p0, km, h, ok);
        }
    }
    WSYNC();
    nscan = nuniq;
    if (A.prm.stop_after == 2) { if (lane == 0) { emit(250, nuniq); } return; }
    RELANE();
    // ---- K2: probe in rounds.  FOUR lanes read one 64-byte bucket, 16 B (two slots) each: one wave-instruction
    //      covers 16 buckets, still one request per 64-byte line, and a 150 bp read's nine wave-loads are all in
    //      flight before the first is consumed.  (Half the steps of an 8-lane layout: the kernel is bound by
    //      instruction issue.  One lane per bucket is slower: its four 16-byte loads are four requests per line.)
    //      Slots fill a bucket front to back (wide_insert), so "has a free slot" is "the last slot is empty".
    //      A k-mer whose bucket is full without the key (31 % at load 0.8) goes on a pending list; round 1 reads
    //      its next two buckets at once, later rounds four: nearly every read is done after three round trips.
    {
        const GAS u32x4* quarters = (const GAS u32x4*)g_slots;  // 4 per bucket
        uint16_t* plist = (uint16_t*)hv;                        // two lists of U entries; the k-mer hash is dead
        uint32_t npend = 0;                                     // wave-uniform length of the list being written
        auto probe = [&](auto nb_tag, auto nl_tag, auto ident_tag, uint32_t np, const uint16_t* pcur, uint16_t* pnext) {
            constexpr int NB = decltype(nb_tag)::value;     // buckets per k-mer: 1 (home), 2 or 4
            constexpr int NLX = decltype(nl_tag)::value;    // wave-loads in flight
            constexpr bool IDENT = decltype(ident_tag)::value;
            constexpr int LPK = 4 * NB;                     // lanes per k-mer
            constexpr int KPL = 64 / LPK;                   // k-mers per wave-load
            constexpr uint32_t GM = NB == 4 ? 0xFFFFu : (NB == 2 ? 0xFFu : 0xFu);
            const int gk = lane / LPK, bi = (lane >> 2) & (NB - 1), q4 = lane & 3;
            for (uint32_t base = 0; base < np; base += NLX * KPL) {
                u32x4 sl[NLX];
#pragma unroll
                for (int i = 0; i < NLX; ++i) {
                    const uint32_t li = base + i * KPL + gk;
                    u32x4 v = {0u, 0u, 0u, 0u};
                    if (li < np) {
                        uint32_t b = ubucket[IDENT ? li : (uint32_t)pcur[li]] + bi;
                        if (b >= tb.nbuckets) b -= tb.nbuckets;
                        v = quarters[(uint64_t)b * 4 + q4];
                    }
                    sl[i] = v;
                }
#pragma unroll
                for (int i = 0; i < NLX; ++i) {
                    if (base + i * KPL >= np) break;
                    const uint32_t li = base + i * KPL + gk;
                    const bool act = li < np;
                    const uint32_t idx = act ? (IDENT ? li : (uint32_t)pcur[li]) : 0u;
                    const uint64_t kmsh1 = act ? ukmer[idx] : 0ull;  // (k-mer << 24) + 1
                    // slot of key K with payload p is K << 24 | p: match <=> slot - ((K << 24) + 1) < 0xFFFFFF, p = that + 1
                    const uint64_t t0 = (((uint64_t)sl[i].y << 32) | sl[i].x) - kmsh1;
                    const uint64_t t1 = (((uint64_t)sl[i].w << 32) | sl[i].z) - kmsh1;
                    const bool m0 = act && t0 < 0xFFFFFFull, m1 = act && t1 < 0xFFFFFFull;
                    const bool free_slot = act && q4 == 3 && (sl[i].z | sl[i].w) == 0u;
                    const uint32_t se = (uint32_t)(__ballot(m0 || m1 || free_slot) >> (gk * LPK)) & GM;  // nibble j = bucket j
                    // the first bucket (in probe order) that holds the key or has a free slot settles the k-mer
                    const int settle = se ? (__builtin_ctz(se) >> 2) : NB;
                    if ((m0 || m1) && bi == settle) upay[idx] = (uint32_t)(m0 ? t0 : t1) + 1u;
                    const bool pend = act && (lane & (LPK - 1)) == 0 && settle == NB;
                    const uint64_t pm = __ballot(pend);
                    if (pend) {
                        uint32_t b = ubucket[idx] + NB;
                        if (b >= tb.nbuckets) b -= tb.nbuckets;
                        ubucket[idx] = b;
                        pnext[npend + prefix_count(pm)] = (uint16_t)idx;
                    }
                    npend += (uint32_t)popc64(pm);
                }
            }
        };
        constexpr int NL = (U + 15) / 16 < 9 ? (U + 15) / 16 : 9;
        probe(std::integral_constant<int, 1>{}, std::integral_constant<int, NL>{}, std::true_type{}, nuniq, plist, plist + U);
        WSYNC();
        int round = 1;
        while (npend > 0) {
            const uint16_t* pcur = plist + ((round & 1) ? U : 0);
            uint16_t* pnext = plist + ((round & 1) ? 0 : U);
            const uint32_t np = npend;
            npend = 0;
            if (round == 1) probe(std::integral_constant<int, 2>{}, std::integral_constant<int, 5>{}, std::false_type{}, np, pcur, pnext);
            else probe(std::integral_constant<int, 4>{}, std::integral_constant<int, 2>{}, std::false_type{}, np, pcur, pnext);
            WSYNC();
            ++round;
        }
    }
    WSYNC();
    } else {
    // ==== compact layout =====================================================================================
    // Per 64-position chunk: neighbouring k-mers that share a bucket form a group; the group's 64-byte bucket is
    // copied once from HBM straight into LDS (global_load_lds, 4 lanes x 16 B = one request per bucket, 16 buckets
    // per wave-instruction), then every k-mer's lane matches its 16-bit tag against the 12 tags of its group's
    // bucket.  A k-mer missing from a bucket that spilled (header bit) is looked up in the overflow table in one
    // batched pass at the end.  upay[] gets one entry per k-mer POSITION (0 for repeats, misses and invalid
    // windows): K3 only needs the payloads in first-occurrence order, which position order is.
    LAS uint32_t* stage = (LAS uint32_t*)xl;
    LAS uint32_t* gbkt = (LAS uint32_t*)(xl + L::XL_GBKT);  // bucket of group g of the chunk
    LAS unsigned int* bloomA = (LAS unsigned int*)(xl + L::XL_BLOOM);
    LAS unsigned int* bloomB = (LAS unsigned int*)(xl + L::XL_BLOOM + 256);
    LAS u32x4* olist = (LAS u32x4*)(xl + L::XL_OLIST);  // k-mers to look up in the overflow table: k-mer, bucket, tag | position << 16
    const GAS u32x4* quarters = (const GAS u32x4*)g_slots;
    uint64_t firstm[KC];
    // ---- repeats.  A k-mer seen twice in a read is looked up once (read_label.cpp:985,1010,1017).  Exact detection
    //      (the LDS hash of the wide path) runs only when a cheap filter cannot rule a repeat out.  Here a group is a
    //      run of k-mers around ONE occurrence of their minimizer (same bucket, same minimizer position in the read:
    //      at most 4 k-mers).  Two positions with the same k-mer are then either in one group, hence within 3 bases
    //      of each other (compared directly, on a 32-bit digest), or in two groups with the same bucket (two
    //      2048-bit filters over the buckets of the groups' first k-mers).
    bool suspect = !CACHE;
    if (CACHE) {
        // bucket / minimizer position / digests of the last lanes of the previous chunk.  An invalid window carries the digest and the
        // minimizer position kNone, which no valid one has (a digest by a 2^-32 chance: a false alarm costs the exact pass below, never
        // a miss): "lane - d is valid" is then part of the compare, not three shifted copies of the valid mask on the scalar side
        constexpr uint32_t kNone = 0xFFFFFFFFu;
        uint32_t pb = 0, pq = kNone, ps1 = kNone, ps2 = kNone, ps3 = kNone;
        uint64_t seen = 0;
        // LSH: bucket, digest and minimizer position of every k-mer parked in LDS (three zero entries in front: nothing precedes
        // the read), read back at -1, -2, -3
        LAS uint32_t* barr = (LAS uint32_t*)xl;
        LAS uint32_t* sarr = barr + (64 * CH + 4);
        LAS uint16_t* qarr = (LAS uint16_t*)(sarr + (64 * CH + 4));
        if constexpr (LSH) {
            static_assert(10 * (64 * CH + 4) <= WL<U, T, E, INK4, CPT>::XL_BLOOM, "the parked values end below the repeat filter's bits");
            WSYNC();  // (the m-mer values above are through)
            if (lane < 3) { barr[lane] = 0u; sarr[lane] = kNone; qarr[lane] = 0xFFFF; }
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if ((uint32_t)c * 64 >= P) break;
                barr[3 + c * 64 + lane] = hreg[c];
                sarr[3 + c * 64 + lane] = lane_bit(okm[c]) ? (uint32_t)(kreg[c] ^ (kreg[c] >> 17)) : kNone;
                qarr[3 + c * 64 + lane] = lane_bit(okm[c]) ? (uint16_t)((uint32_t)c * 64 + lane + (treg[c] >> 30)) : (uint16_t)0xFFFF;
            }
            WSYNC();
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if ((uint32_t)c * 64 >= P) break;
            const uint64_t V = okm[c];
            const uint32_t b = hreg[c];
            const uint32_t sig = lane_bit(V) ? (uint32_t)(kreg[c] ^ (kreg[c] >> 17)) : kNone;
            const uint32_t q = lane_bit(V) ? (uint32_t)c * 64 + lane + (treg[c] >> 30) : (LSH ? 0xFFFFu : kNone);  // where the k-mer's minimizer starts in the read
            uint32_t bprev, qprev, s1, s2, s3;
            if constexpr (LSH) {
                bprev = barr[c * 64 + lane + 2];
                qprev = qarr[c * 64 + lane + 2];
                s1 = sarr[c * 64 + lane + 2]; s2 = sarr[c * 64 + lane + 1]; s3 = sarr[c * 64 + lane];
            } else {
                bprev = (uint32_t)__builtin_amdgcn_update_dpp((int)pb, (int)b, 0x138, 0xf, 0xf, false);  // wave_shr:1
                qprev = (uint32_t)__builtin_amdgcn_update_dpp((int)pq, (int)q, 0x138, 0xf, 0xf, false);
                s1 = (uint32_t)__builtin_amdgcn_update_dpp((int)ps1, (int)sig, 0x138, 0xf, 0xf, false);
                s2 = (uint32_t)__builtin_amdgcn_update_dpp((int)ps2, (int)s1, 0x138, 0xf, 0xf, false);
                s3 = (uint32_t)__builtin_amdgcn_update_dpp((int)ps3, (int)s2, 0x138, 0xf, 0xf, false);
            }
            // lane masks, combined on the scalar side: a ballot of `a && b` goes through a 0/1 register and a compare, a ballot of
            // one compare is the compare
            uint64_t hitm = V & (__ballot(sig == s1) | __ballot(sig == s2) | __ballot(sig == s3));
            const uint64_t openm = V & ~(__ballot(b == bprev) & __ballot(q == qprev));  // lanes that open a group
            uint32_t both = 0;
            if (lane_bit(openm)) {
                const uint32_t h1 = b & 2047u, h2 = (b >> 11) & 2047u;
                const unsigned int oa = __hip_atomic_fetch_or(&bloomA[h1 >> 5], 1u << (h1 & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const unsigned int ob = __hip_atomic_fetch_or(&bloomB[h2 >> 5], 1u << (h2 & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                both = (oa >> (h1 & 31)) & (ob >> (h2 & 31)) & 1u;
            }
            hitm |= __ballot(both != 0u);
            seen |= hitm;
            if constexpr (!LSH) {
                pb = (uint32_t)__builtin_amdgcn_readlane((int)b, 63);
                pq = (uint32_t)__builtin_amdgcn_readlane((int)q, 63);
                ps1 = (uint32_t)__builtin_amdgcn_readlane((int)sig, 63);
                ps2 = (uint32_t)__builtin_amdgcn_readlane((int)sig, 62);
                ps3 = (uint32_t)__builtin_amdgcn_readlane((int)sig, 61);
            }
        }
        suspect = seen != 0;
    }
    if (suspect) {
        for (int i = lane; i < L::H; i += 64) hv[i] = kEmpty64;
        WSYNC();
        if (CACHE) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if ((uint32_t)c * 64 >= P) break;
                const bool ok = lane_bit(okm[c]);
                uint32_t h = 0;
                if (ok) h = lds_min_insert(hv, L::H - 1, kreg[c], (uint32_t)c * 64 + lane);
                treg[c] = (treg[c] & 0xC000FFFFu) | (h << 16);  // the tag is 16 bits wide; the hash slot rides above it
            }
            WSYNC();
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if ((uint32_t)c * 64 >= P) break;
                const bool ok = lane_bit(okm[c]);
                firstm[c] = __ballot(ok && (uint32_t)(hv[(treg[c] >> 16) & 0x3FFFu] & 0xFFFF) == (uint32_t)c * 64 + lane);
                nuniq += popc64(firstm[c]);
            }
        } else {
            for (uint32_t p0 = 0; p0 < P; p0 += 64) {
                const uint32_t p = p0 + lane;
                uint64_t km = 0, kr = 0;
                bool fc;
                if (p < P && window(p, km, kr, fc)) lds_min_insert(hv, L::H - 1, km, p);
            }
        }
        WSYNC();
    } else {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if ((uint32_t)c * 64 >= P) break;
            firstm[c] = okm[c];
            nuniq += popc64(okm[c]);
        }
    }
    STOP_AT(46, nuniq);
    RELANE();
    // ---- probe, chunk by chunk
    uint32_t nov = 0;  // entries of olist
    auto ovf_pass = [&]() {  // looks the listed k-mers up in the overflow table: wide layout, linear probing, 4 lanes per bucket
      // OVL rounds of 16 entries with their first buckets all in flight before any is looked at: a near-capacity table displaces a
      // quarter of its k-mers, a read then lists 30 .. 60 of them, and a round trip per 16 (the shape of this pass until round 4)
      // was three or four dependent trips to memory per read.  The rare chain beyond the first bucket is walked round by round.
      constexpr int OVL = (!INK4 && E <= kFastE && U <= 256) ? (U <= 160 ? 2 : 1) : 3;   // (the classes compiled for 8 waves per SIMD have 64 registers: two rounds fit, three spilled)
      const GAS u32x4* oq = (const GAS u32x4*)tb.ovf_slots;
      const int q4 = lane & 3;
      for (uint32_t e0 = 0; e0 < nov; e0 += 16 * OVL) {
        u32x4 v[OVL];
        uint32_t ob[OVL];
#pragma unroll
        for (int g = 0; g < OVL; ++g) {
            const uint32_t en = e0 + 16u * g + ((uint32_t)lane >> 2);
            v[g] = u32x4{1u, 0u, 1u, 0u};
            ob[g] = 0;
            if (en < nov) {
                const u32x4 e = olist[en];
                ob[g] = ovf_bucket_of(e.z, e.w & 0xFFFFu, tb.ovf_nbuckets);
                v[g] = oq[(uint64_t)ob[g] * 4 + q4];
            }
        }
#pragma unroll
        for (int g = 0; g < OVL; ++g) {
            if (e0 + 16u * g >= nov) break;
            const uint32_t en = e0 + 16u * g + ((uint32_t)lane >> 2);
            bool act = en < nov;
            u32x4 e = {0u, 0u, 0u, 0u};
            if (act) e = olist[en];
            const uint64_t key1 = ((((uint64_t)e.y << 32) | e.x) << kPayloadBits) + 1;
            u32x4 vv = v[g];
            uint32_t obg = ob[g];
            while (true) {
                const uint64_t t0 = (((uint64_t)vv.y << 32) | vv.x) - key1, t1 = (((uint64_t)vv.w << 32) | vv.z) - key1;
                const bool m0 = act && t0 < 0xFFFFFFull, m1 = act && t1 < 0xFFFFFFull;
                if (m0 || m1) upay[e.w >> 16] = (uint32_t)(m0 ? t0 : t1) + 1u;
                // found, or a free last slot (slots fill front to back) ends the chain for all four lanes of the entry
                const uint32_t done = (uint32_t)(__ballot(m0 || m1 || (q4 == 3 && (vv.z | vv.w) == 0u)) >> (lane & ~3)) & 0xFu;
                if (done) act = false;
                if (!__ballot(act)) break;
                obg = obg + 1 == tb.ovf_nbuckets ? 0 : obg + 1;
                vv = u32x4{1u, 0u, 1u, 0u};
                if (act) vv = oq[(uint64_t)obg * 4 + q4];
            }
        }
      }
      nov = 0;
    };
    auto probe_chunk = [&](uint32_t p0, uint64_t km, uint32_t b, uint32_t tag, uint64_t okmask, uint64_t firstmask) {
        const uint32_t p = p0 + lane;
        const uint64_t V1 = okmask << 1;  // lane - 1 holds a valid k-mer (lane 0 always opens a group)
        const uint32_t bprev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b, 0x138, 0xf, 0xf, false);  // wave_shr:1
        const uint64_t lm = okmask & ~(V1 & __ballot(b == bprev));  // leaders (masks combined on the scalar side, as in the filter)
        const bool leader = lane_bit(lm);
        const uint32_t ng = (uint32_t)popc64(lm);
        const uint32_t gidx = prefix_count(lm) + (leader ? 1u : 0u) - 1u;  // group of this lane (ok lanes): leaders at or below it, minus one
        if (leader) gbkt[gidx] = b;
        WSYNC();
        uint32_t pay = 0;
        uint32_t spillw = 0;  // the bucket's last word where the k-mer is not in it: bit 31 = the bucket spilled
        // the groups' buckets, 32 per round of loads (a chunk of a genome read has ~26 groups: one round)
        for (uint32_t g0 = 0; g0 < ng; g0 += 32) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (g0 + (uint32_t)s * 16 < ng) {
                    const uint32_t g = g0 + (uint32_t)s * 16 + ((uint32_t)lane >> 2);
                    if (g < ng) {
                        const uint32_t gb = gbkt[g];
                        __builtin_amdgcn_global_load_lds((const GAS void*)(quarters + (uint64_t)gb * 4 + (lane & 3)),
                                                         (LAS void*)(xl + s * 1024), 16, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_s_waitcnt(0);
            WSYNC();
            if (lane_bit(okmask & __ballot(gidx - g0 < 32u))) {
                const LAS uint32_t* bk = stage + (gidx - g0) * 16;
                const u32x4 t0 = *(const LAS u32x4*)bk;
                const uint32_t t4 = bk[4], t5 = bk[5];
                const uint32_t key2 = tag | (tag << 16);
                // z(x): each 16-bit half is 0 where the slot's tag equals the k-mer's, else 1 (tags are never 0, empty slots are)
                // (v_pk_min_u16 by name: the element-wise min builtin came out as two 16-bit compares, two selects and a byte permute)
                auto z = [&](uint32_t x) -> uint32_t {
                    uint32_t r;
                    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(x ^ key2), "v"(0x00010001u));
                    return r;
                };
                const uint32_t miss = z(t0.x) | (z(t0.y) << 1) | (z(t0.z) << 2) | (z(t0.w) << 3) | (z(t4) << 4) | (z(t5) << 5);
                const uint32_t acc = ~miss & 0x003F003Fu;  // bit i: slot 2i matches, bit 16 + i: slot 2i + 1
                if (acc) {
                    const uint32_t bit = (uint32_t)__builtin_ctz(acc);
                    const uint32_t slot = 2 * (bit & 15u) + (bit >> 4);
                    pay = (uint32_t)((const LAS uint16_t*)bk)[12 + slot] | ((uint32_t)((const LAS uint8_t*)bk)[48 + slot] << 16);
                } else {
                    spillw = bk[15];
                }
            }
            if (g0 + 32 < ng) WSYNC();  // the next round overwrites the stage
        }
        static_assert(kCptOvfFlag == 0x80000000u, "the spill flag is the sign bit");
        if (p < P) upay[p] = lane_bit(firstmask) ? pay : 0u;
        const uint64_t pm = firstmask & __ballot((int32_t)spillw < 0);
        if (pm) {
            if (nov + (uint32_t)popc64(pm) > 64u) { WSYNC(); ovf_pass(); WSYNC(); }  // the list holds 64 entries: one chunk's worth
            if (lane_bit(pm)) olist[nov + prefix_count(pm)] = u32x4{(uint32_t)km, (uint32_t)(km >> 32), b, tag | (p << 16)};
            nov += (uint32_t)popc64(pm);
        }
        WSYNC();
    };
    if (CACHE) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if ((uint32_t)c * 64 >= P) break;
            if (TAILOK && c == 2 && tail_mode()) {  // the payloads are in place; a repeat of an earlier k-mer of the read gives its own up
                if (suspect && 128u + (uint32_t)lane < P && !lane_bit(firstm[c])) upay[128 + lane] = 0u;
                WSYNC();
                continue;
            }
            probe_chunk((uint32_t)c * 64, kreg[c], hreg[c], treg[c] & 0xFFFFu, okm[c], firstm[c]);
        }
    } else {
        for (uint32_t p0 = 0; p0 < P; p0 += 64) {
            const uint32_t p = p0 + lane;
            uint64_t km = 0, kr = 0;
            uint32_t b = 0, t = 0;
            bool fc;
            const bool ok = p < P && window(p, km, kr, fc);
            bool first = false;
            if (ok) {
                cpt_address(*(const CptGeom*)&tb.cpt, km, kr, b, t);
                first = (uint32_t)(hv[lds_find(hv, L::H - 1, km)] & 0xFFFF) == p;
            }
            const uint64_t fm = __ballot(first);
            nuniq += (uint32_t)popc64(fm);
            probe_chunk(p0, km, b, t, __ballot(ok), fm);
        }
    }
    if (nov) { ovf_pass(); WSYNC(); }
    nscan = P;
    if (A.prm.stop_after == 2) { if (lane == 0) { emit(250, nuniq); } return; }
    }
    STOP_AT(3, upay[0]);
    RELANE();
    // ---- K3a: distinct payloads in first-occurrence order with multiplicities.  Identical payload
    //      means identical taxid list, hence identical contribution at every such position.
    //      A read has few distinct payloads, so they are peeled off one per step with ballots (lane d keeps
    //      payload d and its multiplicity in registers); the LDS hash is only for reads with more than 64.
    uint32_t ndist = 0;
    bool many = false;
    if constexpr (U <= 320) {  // (the 512 class would hold eight chunk masks in scalar registers it does not have)
        // all positions of the read at once, a chunk per register: a payload is then peeled exactly once, where it first occurs,
        // and counted in every later chunk in the same step -- no search for "seen in an earlier chunk" (a read's payloads span
        // chunks: the chunk-by-chunk form below met each one about twice, and this step is mostly scalar instructions)
        constexpr int KP = (U + 63) / 64;
        uint32_t pv[KP];
        uint64_t rm[KP];
#pragma unroll
        for (int c = 0; c < KP; ++c) {
            const uint32_t i = (uint32_t)c * 64 + lane;
            pv[c] = i < nscan ? upay[i] : 0u;
            rm[c] = __ballot(pv[c] != 0u);
        }
        uint32_t dp_reg = 0, dm_reg = 0;
        // Payload and multiplicity go into lane ndist of their registers with v_writelane (a compare, two moves and two selects
        // otherwise).  The lane number rides in m0 -- one scalar operand per vector instruction on this target --, which is the
        // compiler's own (the LDS address of the next global_load_lds may already be in it: the scheduler moves that move freely
        // across an asm that does not name m0): saved and put back inside the one asm statement, never across compiler code.
#pragma unroll
        for (int c0 = 0; c0 < KP; ++c0) {
            while (rm[c0]) {
                const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)pv[c0], __builtin_ctzll(rm[c0]));
                uint32_t cn = 0;
#pragma unroll
                for (int c = c0; c < KP; ++c) {
                    const uint64_t m = __ballot(pv[c] == v);
                    cn += (uint32_t)popc64(m);
                    rm[c] &= ~m;
                }
                uint32_t m0_keep;
                asm volatile("s_mov_b32 %2, m0\n\ts_mov_b32 m0, %4\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %5, m0\n\ts_mov_b32 m0, %2"
                             : "+v"(dp_reg), "+v"(dm_reg), "=&s"(m0_keep) : "s"(v), "s"(ndist), "s"(cn));
                ++ndist;   // (beyond 64 the lane number wraps and earlier entries are overwritten: such a read is handed to a larger class below)
            }
        }
        many = ndist > 64u;
        if (!many) {
            if ((uint32_t)lane < ndist) { dpay[lane] = dp_reg; dmult[lane] = (uint16_t)dm_reg; }  // overlay the (dead) k-mer arrays
            WSYNC();
        }
    } else {
        uint32_t dp_reg = 0, dm_reg = 0;
        for (uint32_t i0 = 0; i0 < nscan && !many; i0 += 64) {
            const uint32_t i = i0 + lane;
            const uint32_t pay = i < nscan ? upay[i] : 0;
            uint64_t rem = __ballot(pay != 0);
            while (rem) {
                const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)pay, __builtin_ctzll(rem));
                const uint64_t m = __ballot(pay == v);
                const uint32_t c = (uint32_t)popc64(m);
                const uint64_t ex = __ballot((uint32_t)lane < ndist && dp_reg == v);
                if (ex) {
                    if (lane == __builtin_ctzll(ex)) dm_reg += c;
                } else {
                    if (ndist >= 64) { many = true; break; }
                    if ((uint32_t)lane == ndist) { dp_reg = v; dm_reg = c; }
                    ++ndist;
                }
                rem &= ~m;
            }
        }
        if (!many) {
            if ((uint32_t)lane < ndist) { dpay[lane] = dp_reg; dmult[lane] = (uint16_t)dm_reg; }  // overlay the (dead) k-mer arrays
            WSYNC();
        }
    }
    if (many && !INK4) {  // more than 64 distinct taxid lists: a larger class takes the read
        if (lane == 0) {
            emit(255, 0);
            if (A.ovf_list) ((GAS uint32_t*)A.ovf_list)[G_ADD(&g_cursor[A.ovf_slot], 1u)] = (uint32_t)r;
            else G_OR(g_err, (uint32_t)kErrTidOverflow);
        }
        return;
    }
    if (INK4 && many) {
        for (int i = lane; i < L::H; i += 64) hv[i] = kEmpty64;
        WSYNC();
        for (uint32_t i0 = 0; i0 < nscan; i0 += 64) {
            const uint32_t i = i0 + lane;
            const uint32_t pay = i < nscan ? upay[i] : 0;
            if (pay) lds_min_insert(hv, L::H - 1, pay, i);
        }
        WSYNC();
        ndist = 0;
        for (uint32_t i0 = 0; i0 < nscan; i0 += 64) {  // dpay/dmult overlay the (dead) k-mer arrays
            const uint32_t i = i0 + lane;
            const uint32_t pay = i < nscan ? upay[i] : 0;
            bool owner = false;
            uint32_t h = 0;
            if (pay) {
                h = lds_find(hv, L::H - 1, pay);
                owner = (uint32_t)(hv[h] & 0xFFFF) == i;
            }
            const uint64_t bm = __ballot(owner);
            WSYNC();
            if (owner) {
                const uint32_t rk = ndist + prefix_count(bm);
                dpay[rk] = pay;
                hv[h] = ((unsigned long long)pay << 16) | (0x8000u | rk);  // first index -> rank (bit 15 marks it)
            }
            ndist += popc64(bm);
        }
        for (uint32_t d = lane; d < (ndist + 1) / 2; d += 64) ((unsigned int*)dmult)[d] = 0;
        WSYNC();
        for (uint32_t i0 = 0; i0 < nscan; i0 += 64) {  // multiplicity = number of distinct k-mers carrying the payload
            const uint32_t i = i0 + lane;
            const uint32_t pay = i < nscan ? upay[i] : 0;
            if (pay) add_u16(dmult, (uint32_t)(hv[lds_find(hv, L::H - 1, pay)] & 0x7FFFu), 1u);
        }
        WSYNC();
    }
    if (A.prm.stop_after == 4) { if (lane == 0) { emit(250, ndist); } return; }
    if (ndist == 0) {  // taxid_lst empty: NoDbHits record, proc_line :1270-1277
        if (lane == 0) { emit(LMAT_ST_NODBHITS, 0); nmacc[1]++; }
        return;
    }
    RELANE();
    if (A.gene_mode) {
        // ---- gene_label (src/gene_label.cpp:218-267, 288-313): the database maps a k-mer to a list of 32-bit gene ids;
        //      every distinct k-mer of the read votes for each gene of its list, genes are registered in first-seen order
        //      (k-mers in position order, then list order), and the call is the gene std::sort(Cmp: count descending) puts
        //      first, scored count / distinct valid k-mers.  List records here are [0x8000][n][0][id lo, id hi]...
        uint32_t* gkey = (uint32_t*)hent;     // gene id + 1 per hash slot (0 = empty)
        uint32_t* gval = (uint32_t*)best;     // registration slot of that gene; 0x80000000 | lane while a chunk decides; 0xFFFFFFFF = none yet
        uint32_t* greg = (uint32_t*)reg;      // [T] gene id per slot (reg + stamp)
        uint32_t* gcnt = (uint32_t*)cnt;      // [T] votes per slot (cnt + leaf)
        uint16_t* gord = (uint16_t*)(lds + L::OFF_R3);  // [T] sort order: the head of the element area, which this mode leaves unused
        const GAS uint16_t* arena = g_arena;
        uint32_t nel = 0;
        bool overflow = false;
        for (uint32_t d0 = 0; d0 < ndist; d0 += 64) {
            const uint32_t d = d0 + lane;
            uint32_t n = 0;
            if (d < ndist) n = arena[LMAT_LIST_OFF(dpay[d], tb.list_shift) + 1];
            uint32_t incl = n;
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t v = __shfl_up(incl, o);
                if (lane >= o) incl += v;
            }
            const uint32_t s0 = nel + incl - n;
            if (d < ndist) {
                dstart[d] = (uint16_t)s0;
                if (s0 + n <= (uint32_t)E)
                    for (uint32_t j = 0; j < n; ++j) el_d[s0 + j] = (eld_t)d;
            }
            nel += __shfl(incl, 63);
        }
        if (nel > (uint32_t)E) overflow = true;
        for (int i = lane; i < L::TH; i += 64) { gkey[i] = 0; gval[i] = 0xFFFFFFFFu; }
        WSYNC();
        uint32_t nT = 0;
        for (uint32_t e0 = 0; e0 < nel && !overflow; e0 += 64) {
            const uint32_t e = e0 + lane;
            const bool act = e < nel;
            uint32_t gid = 0, h = 0, m = 0;
            if (act) {
                const uint32_t d = el_d[e];
                const size_t at = LMAT_LIST_OFF(dpay[d], tb.list_shift) + kListHdr + 2 * (e - dstart[d]);
                gid = (uint32_t)arena[at] | ((uint32_t)arena[at + 1] << 16);
                m = dmult[d];
                const uint32_t key = gid + 1u;
                h = ((key * 0x9E3779B1u) >> 12) & THM;
                while (true) {
                    const uint32_t cur = gkey[h];
                    if (cur == key) break;
                    if (cur == 0) {
                        const uint32_t prev = atomicCAS(&gkey[h], 0u, key);
                        if (prev == 0 || prev == key) break;
                    }
                    h = (h + 1) & THM;
                }
                if (gval[h] >= 0x80000000u) atomicMin(&gval[h], 0x80000000u | (uint32_t)lane);
            }
            WSYNC();
            const bool isnew = act && gval[h] == (0x80000000u | (uint32_t)lane);
            const uint64_t nm_ = __ballot(isnew);
            const uint32_t newcnt = popc64(nm_);
            if (nT + newcnt > (uint32_t)T) { overflow = true; break; }
            if (isnew) {
                const uint32_t sl = nT + prefix_count(nm_);
                gval[h] = sl;
                greg[sl] = gid;
                gcnt[sl] = 0;
            }
            nT += newcnt;
            WSYNC();
            if (act) atomicAdd(&gcnt[gval[h]], m);
            WSYNC();
        }
        if (overflow) {
            if (lane == 0) {
                emit(255, 0);
                if (A.ovf_list) ((GAS uint32_t*)A.ovf_list)[G_ADD(&g_cursor[A.ovf_slot], 1u)] = (uint32_t)r;
                else G_OR(g_err, (uint32_t)kErrTidOverflow);
            }
            return;
        }
        if (lane == 0) {
            for (uint32_t i = 0; i < nT; ++i) gord[i] = (uint16_t)i;
            struct VoteCmp {  // Cmp, gene_label.cpp:102-106
                const uint32_t* c;
                __device__ bool operator()(uint16_t a, uint16_t b) const { return c[a] > c[b]; }
            };
            ss_sort<(T <= 128 ? 8 : 34)>(gord, (int)nT, VoteCmp{gcnt});
            lmat_read_result q;
            q.status = LMAT_ST_CALL; q.match_type = LMAT_MT_DIRECT; q.cand_kmer_cnt = (uint16_t)nuniq;
            q.valid_kmers = valid_kmers; q.read_len = (int)len; q.log_avg = 0; q.stdev = 0;
            q.call_tid = greg[gord[0]];
            q.call_score = (float)gcnt[gord[0]] / (float)nuniq;
            q.cand_off = 0; q.n_cand = gcnt[gord[0]]; q.bin_sel = 0;
            store_result(out, q);
        }
        return;
    }
    // ---- K3b stage 1: the first 16 bytes of every distinct payload's list record in one round of loads: header
    //      and, for lists that keep at most two ids (most do), both id orders.  Singletons are their own element.
    //      Each round trip to memory costs this kernel ~2 us whatever it carries, so the header and the ids come
    //      together and the per-id facts below are one 16-byte record, not six gathers.
    const GAS uint16_t* arena = g_arena;
    uint32_t nel = 0, cand = nuniq, fnd = 0;
    bool any_long = false;
    // The owner (distinct-list index) of every element: each list marks its first element, a running maximum over the elements
    // spreads the marks (list indices grow with the element positions).  A loop "for j < n: el_d[s0 + j] = d" in every list's
    // lane ran to the longest list's length.
    static_assert((E * sizeof(eld_t)) % 256 == 0, "the owner array is cleared a dword per lane and step");
    for (uint32_t i = lane; i < (uint32_t)(E * sizeof(eld_t) / 4); i += 64) ((uint32_t*)el_d)[i] = 0u;
    // The fast classes copy the first 64 bytes of every list record straight into LDS (global_load_lds, four lanes a record, 16
    // records a wave-instruction, as the probe stages its buckets): header and both id orders of a list of up to 14 ids in ONE
    // round trip -- the ids of the longer lists were a second one, asked for only once the header had said how long the list is.
    // The window is the 2560 bytes of the taxid tables (R1: idle until the registration below): the first 40 lists of a read;
    // what lies beyond -- lists 41.. of a read, ids 15.. of a list -- takes the loads from memory as before.
    constexpr bool LSTAGE = CPT && !INK4 && !WIDE && U <= 512;
    constexpr uint32_t kStageLists = 40;
    static_assert(!LSTAGE || (L::OFF_XL == L::OFF_R1 && L::OFF_R2 - L::OFF_R1 >= (int)kStageLists * 64), "the window lies under the taxid tables");
    if constexpr (LSTAGE) {
        const uint32_t nst = ndist < kStageLists ? ndist : kStageLists;
#pragma unroll
        for (int sgrp = 0; sgrp < 3; ++sgrp) {
            if ((uint32_t)sgrp * 16 < nst) {
                const uint32_t g = (uint32_t)sgrp * 16 + ((uint32_t)lane >> 2);
                if (g < nst) {
                    const uint32_t pay = dpay[g];
                    if (pay >= kListBase)
                        __builtin_amdgcn_global_load_lds((const GAS void*)((const GAS unsigned char*)(arena + LMAT_LIST_OFF(pay, tb.list_shift)) + 16 * (lane & 3)),
                                                         (LAS void*)(xl + sgrp * 1024), 16, 0, 0);
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0);
    }
    WSYNC();
    for (uint32_t d0 = 0; d0 < ndist; d0 += 64) {
        const uint32_t d = d0 + lane;
        uint32_t n = 0, fl = 0, m = 0, pay = 0;
        uint32_t w3 = 0, w4 = 0, w5 = 0, w6 = 0;
        if (d < ndist) {
            pay = dpay[d];
            m = dmult[d];
            n = 1;
            w3 = w4 = pay;
            if (pay >= kListBase) {
                u32x4 ch;  // [flags][n_kept][n_raw][ids...]
                if (LSTAGE && d < kStageLists) ch = *(const LAS u32x4*)(xl + d * 64);
                else ch = *(const GAS u32x4*)(arena + LMAT_LIST_OFF(pay, tb.list_shift));
                fl = ch.x & 0xFFFFu;
                n = ch.x >> 16;
                if constexpr (WIDE) {  // ids are (low, high) pairs: the first kept id and the first ascending one fit the 16 bytes
                    w3 = (ch.y >> 16) | (ch.z << 16);
                    w4 = (ch.z >> 16) | (ch.w << 16);
                } else { w3 = ch.y >> 16; w4 = ch.z & 0xFFFFu; w5 = ch.z >> 16; w6 = ch.w & 0xFFFFu; }
            }
        }
        // inclusive scan of n over the wave
        const uint32_t incl = wave_scan_incl(n);
        if (d < ndist) {
            const uint32_t s0 = nel + incl - n;
            dn[d] = (uint16_t)n;
            dfl[d] = (uint8_t)fl;
            dstart[d] = (uint16_t)s0;
            if (n != 0 && s0 < (uint32_t)E) el_d[s0] = (eld_t)d;
            if (s0 + n <= (uint32_t)E) {  // ids of the short lists
                if (n == 1) { el_t[s0] = (tid_t)w3; el_ta[s0] = (tid_t)w4; }
                else if (!WIDE && n == 2) {
                    el_t[s0] = (tid_t)w3; el_t[s0 + 1] = (tid_t)w4;
                    el_ta[s0] = (tid_t)w5; el_ta[s0 + 1] = (tid_t)w6;
                }
            }
        }
        any_long |= __ballot(n > (WIDE ? 1u : 2u)) != 0;
        // label_vec.first < 0 positions leave the candidate count (quirk Q4); positions with a
        // non-empty set count as found (construct_labels :722-725)
        const uint32_t both = wave_sum(((fl & kListNegFirst) ? m : 0u) | ((n ? m : 0u) << 16));  // each sum is at most the k-mer capacity
        cand -= both & 0xFFFFu;
        fnd += both >> 16;
        nel += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    if (A.prm.stop_after == 7) { if (lane == 0) { emit(250, nel > 65535u ? 65535u : nel); } return; }
    bool overflow = nel > (uint32_t)E;
    if (overflow) {
        if (lane == 0) {
            emit(255, 0);
            if (A.ovf_list) ((GAS uint32_t*)A.ovf_list)[G_ADD(&g_cursor[A.ovf_slot], 1u)] = (uint32_t)r;  // re-run by the large-capacity kernel
            else G_OR(g_err, (uint32_t)kErrTidOverflow);
        }
        return;
    }
    {
        uint32_t carry = 0;
        for (uint32_t e0 = 0; e0 < nel; e0 += 64) {
            const uint32_t e = e0 + lane;
            uint32_t v = e < nel ? (uint32_t)el_d[e] : 0u;
            v = max(wave_scan_max(v), carry);
            carry = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
            if (e < nel) el_d[e] = (eld_t)v;
        }
    }
    WSYNC();
    RELANE();
    // ---- K3b stage 2: ids of the longer lists (one more round, only when there are any), then one 16-byte
    //      fact record per id of the ascending-order copy
    if (any_long) {
        for (uint32_t e0 = 0; e0 < nel; e0 += 64) {
            const uint32_t e = e0 + lane;
            if (e < nel) {
                const uint32_t d = el_d[e], n = dn[d];
                if (n > (WIDE ? 1u : 2u)) {
                    const size_t eoff = LMAT_LIST_OFF(dpay[d], tb.list_shift) + kListHdr;
                    const uint32_t j = e - dstart[d];
                    if constexpr (WIDE) {
                        const uint32_t r0 = arena[eoff + 2 * j], r1 = arena[eoff + 2 * j + 1], a0 = arena[eoff + 2 * (n + j)], a1 = arena[eoff + 2 * (n + j) + 1];
                        el_t[e] = (tid_t)(r0 | (r1 << 16));
                        el_ta[e] = (tid_t)(a0 | (a1 << 16));
                    } else {
                        uint16_t t_reg, t_asc;
                        if (LSTAGE && d < kStageLists && n <= 14u) {  // both orders lie in the staged 64 bytes
                            const LAS uint16_t* rec = (const LAS uint16_t*)(xl + d * 64);
                            t_reg = rec[kListHdr + j]; t_asc = rec[kListHdr + n + j];
                        } else { t_reg = arena[eoff + j]; t_asc = arena[eoff + n + j]; }  // (both loads in flight before either is stored)
                        el_t[e] = t_reg;
                        el_ta[e] = t_asc;
                    }
                }
            }
        }
        WSYNC();
    }
    for (int i = lane; i < L::TH; i += 64) { hent[i] = 0; best[i] = 0; }  // (behind the list window, which lies under these tables)
    WSYNC();
    if constexpr (INK4) {
        for (uint32_t e0 = 0; e0 < nel; e0 += 64) {
            const uint32_t e = e0 + lane;
            if (e < nel) {
                if constexpr (WIDE) {
                    const uint32_t t = el_ta[e];
                    el_poff[e] = g_path_off[t];
                    el_plen[e] = g_path_len[t];
                    el_sp[e] = g_species_of[t];
                    el_fl[e] = g_flags[t];
                } else {
                    const u32x4 f = g_facts16[el_ta[e]];
                    el_poff[e] = f.x;
                    el_plen[e] = (uint16_t)(f.y & 0xFFFFu);
                    el_sp[e] = (tid_t)(f.y >> 16);
                    el_fl[e] = (uint8_t)(f.w >> 16);
                }
            }
        }
        WSYNC();
    }
    RELANE();
    // ---- phase 1 registration (read_label.cpp:1104-1122): first-lookup position order, then the depth-sorted
    //      order inside a k-mer's kept list == element order.  Within a chunk the lowest lane holding a new
    //      taxid registers it (atomicMin on the entry), so order is kept with several lists per chunk.
    //      The same pass makes leaf_track (:1112-1116) and the kept ids' own position counts (:701-721): the hash entry of
    //      an element's id is at hand.  cnt/leaf are packed u16 pairs updated with dword atomics, cleared for all slots.
    STOP_AT(41, nel);
    for (uint32_t i = lane; i < (uint32_t)T; i += 64) ((unsigned int*)cnt)[i] = 0;  // covers cnt[T] and leaf[T]
    uint32_t nT = 0;
    u32x4 fz = u32x4{0u, 0u, 0u, 0u};
    for (uint32_t e0 = 0; e0 < nel && !overflow; e0 += 64) {
        const uint32_t e = e0 + lane;
        const bool act = e < nel;
        const uint32_t t = act ? el_t[e] : 0;
        uint32_t h = 0;
        if (act) {
            h = tid_find_or_claim(hent, THM, t);
            atomicMin(&hent[h], (ent_t)t | ((ent_t)(EPEND | (uint32_t)lane) << ESH));
        }
        WSYNC();
        const uint64_t nm_ = __ballot(e < nel) & __ballot((uint32_t)(hent[h] >> ESH) == (EPEND | (uint32_t)lane));  // (idle lanes read entry 0: harmless)
        const uint32_t newcnt = popc64(nm_);
        if (nT + newcnt > (uint32_t)T) { overflow = true; break; }
        if (lane_bit(nm_)) {
            const uint32_t s = nT + prefix_count(nm_);
            hent[h] = (ent_t)t | ((ent_t)s << ESH);
            reg[s] = (tid_t)t;
            if constexpr (INK4) stamp[s] = 0xFFFF;  // (the lane-parallel closure of the fast classes keeps no stamps)
        }
        nT += newcnt;
        WSYNC();
        // T <= 64: the fact record of every kept id into the lane of its registration slot -- asked for chunk by chunk, as the slots are
        // taken, so that the records of the first chunks (most of a read's ids) are on their way while the later ones register
        if constexpr (!INK4) { if ((uint32_t)lane >= nT - newcnt && (uint32_t)lane < nT) fz = g_facts16[reg[lane]]; }
        if (act) {
            const uint32_t s = (uint32_t)(hent[h] >> ESH);
            const uint32_t m = dmult[el_d[e]];
            add_u16(leaf, s, m);
            add_u16(cnt, s, m);
        }
    }
    if (overflow) {
        if (lane == 0) {
            emit(255, 0);
            if (A.ovf_list) ((GAS uint32_t*)A.ovf_list)[G_ADD(&g_cursor[A.ovf_slot], 1u)] = (uint32_t)r;  // re-run by the large-capacity kernel
            else G_OR(g_err, (uint32_t)kErrTidOverflow);
        }
        return;
    }
    WSYNC();
    STOP_AT(5, nT);
    RELANE();
    // representative strain per species (read_label.cpp:1144-1177): max leaf count, ties -> smallest taxid
    // (-s: the whole post pass :1143-1204 is skipped; the list records already hold the lineages)
    if constexpr (INK4) {
        for (uint32_t e0 = 0; e0 < nel && !PERM; e0 += 64) {
            const uint32_t e = e0 + lane;
            if (e < nel && (el_fl[e] & kFlagStrain) && el_sp[e]) {
                const uint32_t u = el_ta[e];
                const uint32_t lf = leaf[(uint32_t)(hent[tid_find(hent, THM, u)] >> ESH)];
                const uint32_t h = tid_find_or_claim(hent, THM, el_sp[e]);
                atomicMax(&best[h], ((ent_t)lf << ESH) | (EIDM - (ent_t)u));
            }
        }
        WSYNC();
    }
    RELANE();
    // ---- phase 2 (read_label.cpp:1178-1203): per position, ancestors of the eligible kept ids, visited in
    //      ascending taxid order; positions ascending == distinct payloads in first-occurrence order.
    //      Everything but the path elements is already in LDS; the next chain is prefetched.
    auto eligible = [&](uint32_t e) -> bool {  // large classes
        if (PERM) return false;                          // gPERMISSIVE_MATCH: no closure pass
        if (dfl[el_d[e]] & kListNegFirst) return false;  // closure only where first >= 0 (:1179)
        if (!(el_fl[e] & kFlagStrain)) return true;      // rank != "strain" (:1184)
        const uint32_t sp = el_sp[e];
        if (!sp) return false;
        const int h = tid_find(hent, THM, sp);
        return h >= 0 && best[h] != 0 && (best[h] & EIDM) == (EIDM - (ent_t)el_ta[e]);
    };
    if constexpr (INK4) {
            // next eligible element at or after e (uniform scan)
            auto next_elig = [&](uint32_t e) -> uint32_t {
                while (e < nel) {
                    const uint32_t x = e + lane;
                    const uint64_t m = __ballot(x < nel && eligible(x));
                    if (m) return e + (uint32_t)__builtin_ctzll(m);
                    e += 64;
                }
                return nel;
            };
            uint32_t e = next_elig(0);
            uint32_t a_next = 0;
            if (e < nel) a_next = (uint32_t)lane < (uint32_t)el_plen[e] ? g_paths[el_poff[e] + lane] : 0u;
            int cur_d = -1;
            while (e < nel && !overflow) {
                const uint32_t d = el_d[e], m = dmult[d], plen = el_plen[e], poff = el_poff[e];
                if ((int)d != cur_d) {  // entering a new position set: its kept ids are members already
                    cur_d = (int)d;
                    const uint32_t s0 = dstart[d], n = dn[d];
                    for (uint32_t j = lane; j < n; j += 64) stamp[(uint32_t)(hent[tid_find(hent, THM, el_t[s0 + j])] >> ESH)] = (uint16_t)d;
                    WSYNC();
                }
                uint32_t a_cur = a_next;
                const uint32_t e2 = next_elig(e + 1);
                if (e2 < nel) a_next = (uint32_t)lane < (uint32_t)el_plen[e2] ? g_paths[el_poff[e2] + lane] : 0u;
                for (uint32_t c0 = 0; c0 < plen; c0 += 64) {
                    const uint32_t c = c0 + lane;
                    const bool act = c < plen;
                    const uint32_t a = c0 == 0 ? a_cur : (act ? g_paths[poff + c] : 0u);
                    uint32_t h = 0;
                    if (act) {
                        h = tid_find_or_claim(hent, THM, a);
                    }
                    const bool unreg = act && (uint32_t)(hent[h] >> ESH) == (uint32_t)EIDM;
                    const uint64_t nm_ = __ballot(unreg);
                    const uint32_t newcnt = popc64(nm_);
                    if (nT + newcnt > (uint32_t)T) { overflow = true; break; }
                    if (unreg) {
                        const uint32_t s = nT + prefix_count(nm_);
                        hent[h] = (ent_t)a | ((ent_t)s << ESH);
                        reg[s] = (tid_t)a; stamp[s] = 0xFFFF;
                    }
                    nT += newcnt;
                    WSYNC();
                    if (act) {
                        const uint32_t s = (uint32_t)(hent[h] >> ESH);
                        if (stamp[s] != (uint16_t)d) { stamp[s] = (uint16_t)d; cnt[s] += (uint16_t)m; }
                    }
                    WSYNC();
                }
                e = e2;
            }
    } else {
        // T <= 64: registration slots map to lanes, and the chain-by-chain walk above (one LDS-latency-bound step
        // per eligible id) becomes three wave-wide steps:
        //  (1) every ancestor of every eligible id as one flat item list in the walk's order (element, then path
        //      position), so new ids are registered in the same order with the lowest-lane rule of phase 1;
        //  (2) the Euler interval of every registered slot in its lane: word 2 of the fact record (a kept id's own; a new
        //      ancestor's comes with its path entry);
        //  (3) per slot a: count[a] += m_d for each position set d that has an eligible id below a and does not
        //      keep a itself -- what the stamps compute -- from per-set masks in LDS (steps 3a .. 3c below).
        constexpr int EC = E / 64;
        static_assert(E % 64 == 0 && T <= 64, "lane-parallel closure: one lane per registration slot");
        // facts of the id registered in this lane's slot
        const bool sl_p1 = (uint32_t)lane < nT;
        const uint32_t nT_p1 = nT;
        const uint32_t f_poff = fz.x, f_plen = fz.y & 0xFFFFu, f_sp = fz.y >> 16, f_fl = (fz.w >> 16) & 0xFFu;
        const uint32_t my_id = sl_p1 ? (uint32_t)reg[lane] : 0u;
        // representative strain per species, one vote per kept id
        uint32_t hs = 0;  // where the species of this lane's strain sits in the hash (kept for the check below: one search, not two)
        const uint64_t STR = PERM ? 0ull : (__ballot((uint32_t)lane < nT) & __ballot((f_fl & kFlagStrain) != 0) & __ballot(f_sp != 0));
        if (lane_bit(STR)) {
            hs = tid_find_or_claim(hent, THM, f_sp);
            atomicMax(&best[hs], ((uint32_t)leaf[lane] << 16) | (0xFFFFu - my_id));
        }
        WSYNC();
        // which kept ids take part in the closure (:1184-1190): everything but strains, and of the strains the representative
        const uint32_t bh = best[hs];  // (lanes without a strain read entry 0: unused)
        const uint64_t ES = PERM ? 0ull
                                 : ((__ballot((uint32_t)lane < nT) & ~__ballot((f_fl & kFlagStrain) != 0)) |
                                    (STR & __ballot(bh != 0u) & __ballot((bh & 0xFFFFu) == (0xFFFFu - my_id))));
        WSYNC();  // leaf and best are dead from here
        unsigned int* first = best;  // per slot: its first eligible element
        u32x2* anc_zw = (u32x2*)(best + T);  // per slot the closure registers: words 2 and 3 of its fact record (best has 4 T entries; the species votes are through)
        static_assert(L::TH >= 3 * T && ((L::IDB + 6) * T + L::HB * L::TH + 4 * T) % 8 == 0, "room and alignment behind the first-element array");
        first[lane] = 0xFFFFFFFFu;
        WSYNC();
        // Per element: the slot of its id and whether it is eligible (closure only where first >= 0, :1179).  Only the
        // FIRST eligible occurrence of a kept id has to be walked: a later walk of the same chain meets nothing but
        // registered ids.  (Lists of one genus repeat their ids: without this such a read walked thousands of items,
        // one round trip to memory per 64.)
        STOP_AT(42, nT);
        uint32_t W = 0;
        uint32_t sreg[EC];  // slot | eligible << 8
#pragma unroll
        for (int ch = 0; ch < EC; ++ch) {
            sreg[ch] = 0;
            if ((uint32_t)ch * 64 < nel) {
                const uint32_t e = (uint32_t)ch * 64 + lane;
                if (e < nel) {
                    const uint32_t sl = hent[tid_find(hent, THM, el_ta[e])] >> 16;
                    const bool el = !(dfl[el_d[e]] & kListNegFirst) && ((ES >> sl) & 1ull);
                    sreg[ch] = sl | (el ? 0x100u : 0u);
                    el_ta[e] = (uint16_t)sreg[ch];  // (kept for step (3b))
                    if (el) atomicMin(&first[sl], e);
                }
            }
        }
        WSYNC();
        STOP_AT(43, nT);
        // the slots with an eligible element.  One such slot in the whole read (the usual case: strains of one species, the
        // representative one eligible): its chain is the only one to walk -- no offsets to scan, no search per item.
        const uint64_t todo0 = __ballot(first[lane] != 0xFFFFFFFFu);
        const bool one = LMAT_CLOSURE_FAST && todo0 != 0 && (todo0 & (todo0 - 1)) == 0;
        const int sx0 = one ? __builtin_ctzll(todo0) : 0;
        if (one) {
            W = (uint32_t)__builtin_amdgcn_readlane((int)f_plen, sx0);
        } else {
#pragma unroll
            for (int ch = 0; ch < EC; ++ch) {
                if ((uint32_t)ch * 64 < nel) {
                    const uint32_t e = (uint32_t)ch * 64 + lane;
                    const uint32_t sl = sreg[ch] & 0xFFu;
                    const uint32_t plen = lane_of(f_plen, sl);
                    const bool walk = e < nel && first[sl] == e;
                    const uint32_t w = walk ? plen : 0u;
                    const uint32_t incl = wave_scan_incl(w);
                    if (e < nel) el_off[e] = (uint16_t)(W + incl - w);
                    W += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                }
            }
            WSYNC();
        }
        STOP_AT(44, W);
        if (W > 65535u) overflow = true;  // item offsets are u16; such a read goes to the large-capacity kernel
        for (uint32_t i0 = 0; i0 < W && !overflow; i0 += 64) {
            const uint32_t i = i0 + lane;
            const bool act = i < W;
            uint32_t a = 0, h = 0;
            uint64_t pe = 0;
            uint32_t sl_i = 0, rel = 0;
            if (one) {
                sl_i = (uint32_t)sx0;
                rel = i;
            } else if (act) {
                uint32_t lo = 0, hi = nel;  // last element whose offset is <= i (offsets are non-decreasing)
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if ((uint32_t)el_off[mid] <= i) lo = mid + 1; else hi = mid;
                }
                const uint32_t e = lo - 1;
                sl_i = (uint32_t)el_ta[e] & 0xFFu;
                rel = i - (uint32_t)el_off[e];
            }
            const uint32_t poff_i = lane_of(f_poff, sl_i);  // the chain's start, from the slot's lane
            uint32_t pf = 0;
            if (act) {
                pe = g_paths8[poff_i + rel];  // id | depth | tin | tout of that ancestor
                pf = ((const GAS uint8_t*)tb.paths_fl)[poff_i + rel];  // ... and its flags
                a = (uint32_t)(pe & 0xFFFFu);
                h = tid_find_or_claim(hent, THM, a);
                atomicMin(&hent[h], a | ((0x8000u | (uint32_t)lane) << 16));
            }
            WSYNC();
            const uint64_t nm_ = __ballot(i < W) & __ballot((hent[h] >> 16) == (0x8000u | (uint32_t)lane));
            const uint32_t newcnt = popc64(nm_);
            if (nT + newcnt > (uint32_t)T) { overflow = true; break; }
            if (lane_bit(nm_)) {
                const uint32_t s = nT + prefix_count(nm_);
                hent[h] = a | (s << 16);
                reg[s] = (uint16_t)a;
                anc_zw[s] = u32x2{(uint32_t)(pe >> 32), ((uint32_t)(pe >> 16) & 0xFFFFu) | (pf << 16)};  // interval, depth | flags << 16: words 2 and 3 of the id's fact record
            }
            nT += newcnt;
            WSYNC();
        }
        STOP_AT(45, nT);
        // What the decision step wants of the ids the closure registered -- interval, depth, flags -- came with their path entries:
        // into their slots' lanes from LDS (a gather of their fact records here was one more round trip to memory, and the
        // usual read had nothing to do while it was out).
        if (!overflow && (uint32_t)lane >= nT_p1 && (uint32_t)lane < nT) { const u32x2 zw = anc_zw[lane]; fz = u32x4{0u, 0u, zw.x, zw.y}; }
        // One eligible id in the whole read whose ancestors were ALL registered just now (nobody's kept list holds one of them)
        // and no position set with a negative first: every one of those ancestors is reached by exactly the sets that hold the
        // id -- its count is the id's own -- and steps (3a) .. (3c) have nothing else to find.  The usual read: strains of one
        // species, the representative one eligible.  (LMAT_CLOSURE_FAST, compile time.)
        const bool lone = one && !overflow && W > 0 && nT - nT_p1 == W && cand == nuniq;
        if (lone) {
            const uint32_t c0 = cnt[sx0];
            if ((uint32_t)lane >= nT_p1 && (uint32_t)lane < nT) cnt[lane] = (uint16_t)c0;
            WSYNC();
        } else if (!overflow && W > 0) {
            const bool sl_act = (uint32_t)lane < nT;
            uint32_t tin_s = 0xFFFF, tout_s = 0;
            if (sl_act) { tin_s = fz.z & 0xFFFFu; tout_s = fz.z >> 16; }  // (word 2 of the slot's fact record: its own for a kept id, off the path entry for a new ancestor)
            // (3a) for every kept id that is eligible somewhere: the set of slots that are proper ancestors of it, as a
            //      64-bit mask held by the id's own lane
            uint32_t anc_lo = 0, anc_hi = 0;
            uint64_t todo = todo0;
            while (todo) {
                const int sx = __builtin_ctzll(todo);
                todo &= todo - 1;
                const uint32_t tin_x = (uint32_t)__builtin_amdgcn_readlane((int)tin_s, sx);
                const uint32_t tout_x = (uint32_t)__builtin_amdgcn_readlane((int)tout_s, sx);
                const uint64_t am = __ballot(tin_s < tin_x) & __ballot(tout_x <= tout_s);  // (idle slots hold tin 0xFFFF: never below)
                // the mask into its id's own lane (one scalar operand per vector instruction on this target: the lane number rides in m0)
                // (m0 is the compiler's own -- the bucket loads keep the LDS address there --: saved and put back)
                uint32_t m0_keep;
                asm("s_mov_b32 %2, m0\n\ts_mov_b32 m0, %4\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %5, m0\n\ts_mov_b32 m0, %2"
                    : "+v"(anc_lo), "+v"(anc_hi), "=&s"(m0_keep) : "s"((uint32_t)am), "s"(sx), "s"((uint32_t)(am >> 32)));
            }
            // (3b) per position set d (at most 64 of them here): the slots it keeps, and the ancestors of its eligible ids -- two
            //      64-bit masks per set, OR-ed together in LDS by the set's elements, a lane per element (slot and eligibility
            //      are in el_ta).  The per-set loop this replaces ran to the longest list's length with ~18 vector
            //      instructions a step; the vector pipe is what the kernel runs out of.
            unsigned int* memw = (unsigned int*)el_t;    // [64] {low, high}: the registration-order ids and the item offsets are dead
            unsigned int* hitw = (unsigned int*)el_off;  // [64] {low, high}
            static_assert(2 * E >= 512 && sizeof(*el_t) == 2 && sizeof(*el_off) == 2, "64 mask pairs fit each of the two dead arrays");
            *(u32x2*)(memw + 2 * lane) = u32x2{0u, 0u};
            *(u32x2*)(hitw + 2 * lane) = u32x2{0u, 0u};
            const bool wide_t = nT > 32u;  // slots beyond 31 exist: the high halves of the masks are in play (seldom)
            WSYNC();
#pragma unroll
            for (int ch = 0; ch < EC; ++ch) {
                if ((uint32_t)ch * 64 < nel) {
                    const uint32_t e = (uint32_t)ch * 64 + lane;
                    const uint32_t se = e < nel ? (uint32_t)el_ta[e] : 0u;  // slot | eligible << 8 (stored above)
                    const uint32_t sl = se & 0xFFu;
                    const uint32_t alo = lane_of(anc_lo, sl), ahi = wide_t ? lane_of(anc_hi, sl) : 0u;
                    if (e < nel) {
                        const uint32_t d2 = 2u * (uint32_t)el_d[e];
                        __hip_atomic_fetch_or(&memw[d2 + (sl >> 5)], 1u << (sl & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (se & 0x100u) {
                            __hip_atomic_fetch_or(&hitw[d2], alo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            if (wide_t) __hip_atomic_fetch_or(&hitw[d2 + 1], ahi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                }
            }
            WSYNC();
            // (3c) count[a] += m_d for every position set d that reaches a without keeping it: the sets' masks as LDS
            //      broadcasts (a lane reads the half that holds its slot's bit), one bit test and one multiply-add per set
            {   // (every lane: a lane past the last set holds zeros, and the loop below runs over whole groups of four sets)
                const u32x2 me = *(const u32x2*)(memw + 2 * lane), hi_ = *(const u32x2*)(hitw + 2 * lane);
                *(u32x2*)(hitw + 2 * lane) = u32x2{hi_.x & ~me.x, hi_.y & ~me.y};
            }
            WSYNC();
            const unsigned int* aw = hitw + (lane >> 5);
            const uint32_t sh = (uint32_t)lane & 31u;
            uint32_t add = 0;
            // four sets a step: their mask words with constant offsets, their four multiplicities as one 8-byte broadcast; per set
            // a bit-field extract and a 16-bit multiply-add that takes its factor from either half of a register
            auto mad_lo = [](uint32_t bit, uint32_t mm, uint32_t acc) -> uint32_t { uint32_t r; asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(r) : "v"(bit), "v"(mm), "v"(acc)); return r; };
            auto mad_hi = [](uint32_t bit, uint32_t mm, uint32_t acc) -> uint32_t { uint32_t r; asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[0,1,0,0]" : "=v"(r) : "v"(bit), "v"(mm), "v"(acc)); return r; };
            static_assert(L::D == 64 && (L::OFF_R2 + 4 * L::D) % 8 == 0, "the multiplicities are read eight bytes at a time, up to set 63");
            for (uint32_t d = 0; d < ndist; d += 4) {
                const unsigned int* pw = aw + 2 * d;
                const uint32_t w0 = pw[0], w1 = pw[2], w2 = pw[4], w3 = pw[6];
                const u32x2 mm = *(const u32x2*)(dmult + d);
                add = mad_lo(__builtin_amdgcn_ubfe(w0, sh, 1u), mm.x, add);
                add = mad_hi(__builtin_amdgcn_ubfe(w1, sh, 1u), mm.x, add);
                add = mad_lo(__builtin_amdgcn_ubfe(w2, sh, 1u), mm.y, add);
                add = mad_hi(__builtin_amdgcn_ubfe(w3, sh, 1u), mm.y, add);
            }
            if (sl_act && add) cnt[lane] = (uint16_t)(cnt[lane] + add);
            WSYNC();
        }
    }
    if (overflow) {
        if (lane == 0) {
            emit(255, 0);
            if (A.ovf_list) ((GAS uint32_t*)A.ovf_list)[G_ADD(&g_cursor[A.ovf_slot], 1u)] = (uint32_t)r;  // re-run by the large-capacity kernel
            else G_OR(g_err, (uint32_t)kErrTidOverflow);
        }
        return;
    }
    STOP_AT(6, nT);
    if (A.rand_max) {  // rand_read_label: proc_line + construct_labels of src/rand_read_label.cpp:185-213,372-398
        const uint32_t gcb = ((const GAS uint8_t*)A.rand_gc)[r - A.result_base];
        GAS uint32_t* rmax = (GAS uint32_t*)A.rand_max;
        GAS uint32_t* rcnt = (GAS uint32_t*)A.rand_cnt;
        for (uint32_t s = lane; s < nT; s += 64) {
            const float label_prob = (float)cnt[s] / (float)valid_kmers;
            const size_t at = (size_t)reg[s] * A.rand_nb + gcb;
            __hip_atomic_fetch_max(&rmax[at], __float_as_uint(label_prob), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // non-negative floats order like their bits
            G_ADD(&rcnt[at], 1u);
        }
        if (lane == 0) emit(LMAT_ST_SILENT, nT);
        return;
    }
    RELANE();
    cand &= 0xFFFF;  // uint16_t cand_kmer_cnt (:699)
    // ---- construct_labels early exits (:727-733): nothing is written (quirk Q1), tallied NoDbHits
    if ((int)fnd < A.prm.min_fnd_kmer || (int)cand < A.prm.min_kmer) {
        if (lane == 0) {
            emit(nT ? LMAT_ST_SILENT : LMAT_ST_NODBHITS, cand);
            nmacc[1]++;
        }
        return;
    }
    if (nT == 0) {  // hits whose kept lists are all empty: taxid_lst empty, proc_line :1270-1277
        if (lane == 0) { emit(LMAT_ST_NODBHITS, 0); nmacc[1]++; }
        return;
    }
    if (!INK4) {
        // hand the read to the K4 kernels (one lane per read): registration-ordered (taxid, count) table + cand
        if (nT > (uint32_t)kK4T) {
            if (lane == 0) {
                emit(255, 0);
                ((GAS uint32_t*)A.ovf_list)[G_ADD(&g_cursor[A.ovf_slot], 1u)] = (uint32_t)r;
            }
            return;
        }
        // Where the decision is made: tables of up to 16 taxids go to k4_row_kernel, four reads to a wave (status 251); larger ones
        // are decided right here on the wave (k4_wave) when the read qualifies; everything else -- null models, an effective human
        // bias, -y style experiments, more than 999 candidate k-mers ... -- takes the general path, by table size.
        const bool rows = (A.prm.k4_static & 2) && nT <= 16u;
        if constexpr (!INK4) {
            if (!rows) {
                const uint32_t my_cnt = (uint32_t)lane < nT ? (uint32_t)cnt[lane] : 0u, my_id = (uint32_t)lane < nT ? (uint32_t)reg[lane] : 0u;
                if (k4_wave<THM>(Ap, lane, nT, cand, fz, my_cnt, my_id, hent, (uint32_t*)(lds + L::OFF_R3), out, valid_kmers, len, bin_sel)) return;
            }
        }
        GAS uint32_t* krec = (GAS uint32_t*)A.k4buf + (r - A.result_base) * kK4RecWords;
        if (lane < (int)nT) krec[2 + lane] = (uint32_t)reg[lane] | ((uint32_t)cnt[lane] << 16);
        if (lane == 0) {
            krec[0] = nT | (cand << 16);
            emit(rows ? 251u : (A.nm.active ? 253u : (nT <= (uint32_t)kK4SmallT ? 254u : (nT <= (uint32_t)kK4MidT ? 252u : 253u))), cand);  // pending K4
        }
        return;
    }
    // ---- K4 staging: per-slot taxonomy facts in one round of loads
    lmat_read_result res;
    res.status = LMAT_ST_NODBHITS; res.match_type = LMAT_MT_NOMATCH; res.cand_kmer_cnt = 0; res.valid_kmers = valid_kmers;
    res.read_len = (int)len; res.log_avg = 0; res.stdev = 0; res.call_tid = 0; res.call_score = 0;
    res.cand_off = 0; res.n_cand = 0; res.bin_sel = bin_sel;
    const NullModelDev& nm_g = *(const NullModelDev*)&A.nm;  // the large classes' own decision step takes generic references
    const KernelParams& prm_g = *(const KernelParams*)&A.prm;
    const int nmt = nm_table_of(nm_g, cand);
    for (uint32_t s = lane; s < nT; s += 64) {
        const uint32_t t = reg[s];
        dep[s] = g_fdepth[t];
        sflags[s] = g_flags[t];
        tin[s] = g_tin[t];
        tout[s] = g_tout[t];
        if (nmt >= 0) {  // null-model probability of this taxid at the read's GC bin (read_label.cpp:768-775)
            const size_t row = (size_t)nmt * tb.n_ids + t;
            const uint8_t cl = ((const GAS uint8_t*)A.nm.cls)[row];
            const int nb = ((const GAS int*)A.nm.nbins)[nmt];
            const int bin = res.bin_sel < nb ? res.bin_sel : nb - 1;
            if (cl == 0xFF) G_OR(g_err, (uint32_t)kErrNoNullModel);
            nm_cl[s] = cl == 0xFF ? 0 : cl;
            nm_rp[s] = (float)((double)((const GAS float*)A.nm.val)[row * A.nm.nb_max + bin] + 0.0001);
        }
    }
    WSYNC();
    // ---- K4 part 1 on lane 0 (LDS only)
    K4State S;
    S.done = false; S.nlin = 0; S.highest = -1; S.highest_depth = 0; S.lidx = -1; S.lowest = -1; S.plasmid_slot = -1;
    S.top_score = 0; S.diff_thresh = 0;
    if (lane == 0) k4_part1<L::LIN, tid_t, lin_t>(prm_g, res, S, cnt, score, score0, dep, sflags, tin, tout, reg, ord, lin, (int)nT, cand, nmt >= 0, nm_rp, nm_cl, nm_g);
    const int done = __builtin_amdgcn_readfirstlane((int)S.done);
    int nlin = __builtin_amdgcn_readfirstlane(S.nlin);
    const int highest = __builtin_amdgcn_readfirstlane(S.highest);
    const unsigned highest_depth = (unsigned)__builtin_amdgcn_readfirstlane((int)S.highest_depth);
    WSYNC();
    uint32_t call_idx = 0, ncand = 0, coff = 0;
    bool have_add = false;
    uint32_t high_tin = 0xFFFF, high_tout = 0xFFFF;
    if (!done) {
        // ancestors of the shallowest accepted node join the lineage with their pre-bias scores or -10000
        // (:326-343); one lane per ancestor
        const uint32_t high_tid = highest >= 0 ? reg[highest] : 0;
        have_add = highest_depth != 0 && high_tid != 0;
        if (have_add) {
            high_tin = tin[highest];
            high_tout = tout[highest];
            const uint32_t alen = g_path_len[high_tid], aoff = g_path_off[high_tid];
            uint32_t room = (uint32_t)L::LIN - (uint32_t)nlin;
            if (alen > room) { if (lane == 0) G_OR(g_err, (uint32_t)kErrLineageTrunc); }
            const uint32_t take = alen < room ? alen : room;
            for (uint32_t j0 = 0; j0 < take; j0 += 64) {
                const uint32_t j = j0 + lane;
                if (j < take) {
                    const uint32_t a = g_paths[aoff + j];
                    const int s = tid_slot(hent, THM, a);
                    lin_t en;
                    en.tid = (tid_t)a;
                    en.score = s >= 0 ? score0[s] : -10000.0f;
                    en.dep = g_fdepth[a]; en.tin = g_tin[a]; en.tout = g_tout[a];
                    lin[nlin + j] = en;
                }
            }
            nlin += (int)take;
        }
        WSYNC();
    }
    // ---- K4 part 2 on lane 0
    if (lane == 0) {
        GAS lmat_cand* cout_ = nullptr;
        if (!done) {
            if (A.cands) {
                const uint32_t reserve = A.prm.prn_all ? nT : (uint32_t)nlin;  // -p: the candidates; else the lineage as built (:917-927)
                coff = G_ADD(&g_cursor[0], reserve);
                if ((uint64_t)coff + reserve <= A.cand_cap) cout_ = (GAS lmat_cand*)A.cands + coff;
                else G_OR(g_err, (uint32_t)kErrCandOverflow);
            }
            k4_part2<L::LIN, tid_t, lin_t>(prm_g, g_tid32, res, S, score, tin, tout, reg, ord, lin, nlin, (int)nT, have_add, high_tin,
                     high_tout, cout_, &ncand, &call_idx);
        } else {
            call_idx = A.phix_call_idx;
        }
        res.cand_off = coff;
        res.n_cand = ncand;
        store_result(out, res);
        // tallies, proc_line :1241-1268
        if (res.status != LMAT_ST_PHIX && res.match_type == LMAT_MT_NOMATCH) {
            G_ADD(&tally_nomatch[1], 1ull);
        } else if (res.call_score >= A.prm.min_score) {
            G_ADD(&tally_count[call_idx], 1ull);
            G_ADD(&tally_score[call_idx], (double)res.call_score);
        } else if (res.call_score < A.prm.min_score) {
            G_ADD(&tally_nomatch[2], 1ull);
        }
    }
}

#pragma pop_macro("tb")
#pragma pop_macro("A")
#pragma pop_macro("WSYNC")

// ------------------------------------------------------------------------------------------
// K4 as its own step: one lane per read (64 reads per wave), running k4_part1 / k4_part2 as the in-kernel
// lane-0 path of the large-capacity kernel does.  The step is a chain of several hundred dependent accesses to
// the per-read tables, so where those tables live decides its speed:
//   k4_lds_kernel   reads with at most kK4SmallT registered taxids (three in four): tables in LDS, each lane's
//                   block an odd number of dwords apart (conflict-free), ~100 cycles per access;
//   k4_kernel       the rest, and everything when null models are loaded: tables in private (scratch)
//                   memory, which at 4 KB per lane lives in HBM, ~2 us per access.
// k4_compact_kernel splits the pending reads of a batch into the two lists (one atomic per wave).
// ------------------------------------------------------------------------------------------
template <int TT, int LIN, bool NM>
__device__ __forceinline__ bool k4_read(const ClassifyArgs& A, uint64_t it, uint16_t* reg, uint16_t* cnt, uint16_t* dep,
                                        uint16_t* tin, uint16_t* tout, uint16_t* ord, uint8_t* sflags, float* score,
                                        float* score0, uint8_t* nm_cl, float* nm_rp, LinEnt* lin, bool bail_on_long, K4Key* keys = nullptr) {
    const DeviceTables& tb = A.tb;
    const GAS uint32_t* g_tid32 = (const GAS uint32_t*)tb.tid32;
    const GAS uint64_t* g_paths8 = (const GAS uint64_t*)tb.paths8;
    const GAS u32x4* g_facts16 = (const GAS u32x4*)tb.facts16;
    GAS uint32_t* g_cursor = (GAS uint32_t*)A.cursor;
    GAS uint32_t* g_err = (GAS uint32_t*)A.err;  // sticky error flags, shared by all launches
    GAS unsigned long long* tally_count = (GAS unsigned long long*)A.counts;
    GAS double* tally_score = (GAS double*)(tally_count + tb.n_ids);
    GAS unsigned long long* tally_nomatch = (GAS unsigned long long*)(tally_score + tb.n_ids);
    const GAS uint32_t* krec = (const GAS uint32_t*)A.k4buf + it * kK4RecWords;
    GAS uint64_t* out = (GAS uint64_t*)(A.results + it);
    lmat_read_result res;
    {
        uint64_t w[5];
        w[0] = out[0]; w[1] = out[1]; w[2] = out[2]; w[3] = out[3]; w[4] = out[4];
        __builtin_memcpy(&res, w, 40);
    }
    if (res.status != 254 && res.status != 253 && res.status != 252) return true;
    const uint32_t hdr = krec[0];
    const int nT = (int)(hdr & 0xFFFFu);
    const uint32_t cand = hdr >> 16;
    if (nT > TT) return false;
    const int nmt = NM ? nm_table_of(A.nm, cand) : -1;  // NM: compile-time, keeps the plain path free of the track[] scratch
    // the (taxid, count) words and the per-taxid facts in batches of eight: 2 round trips to memory per batch, not per id
    for (int s0 = 0; s0 < nT; s0 += 8) {
        uint32_t w[8];
        u32x4 f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = s0 + j < nT ? krec[2 + s0 + j] : 0u;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = g_facts16[w[j] & 0xFFFFu];  // one record instead of four gathers
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int s = s0 + j;
            if (s < nT) {
                reg[s] = (uint16_t)(w[j] & 0xFFFFu);
                cnt[s] = (uint16_t)(w[j] >> 16);
                dep[s] = (uint16_t)(f[j].w & 0xFFFFu);
                sflags[s] = (uint8_t)(f[j].w >> 16);
                tin[s] = (uint16_t)(f[j].z & 0xFFFFu);
                tout[s] = (uint16_t)(f[j].z >> 16);
            }
        }
    }
    if (NM && nmt >= 0) {  // null-model probability of every taxid at the read's GC bin (read_label.cpp:768-775)
        for (int s = 0; s < nT; ++s) {
            const uint32_t t = reg[s];
            const size_t row = (size_t)nmt * tb.n_ids + t;
            const uint8_t cl = ((const GAS uint8_t*)A.nm.cls)[row];
            const int nb = ((const GAS int*)A.nm.nbins)[nmt];
            const int bin = res.bin_sel < nb ? res.bin_sel : nb - 1;
            if (cl == 0xFF) G_OR(g_err, (uint32_t)kErrNoNullModel);
            nm_cl[s] = cl == 0xFF ? 0 : cl;
            nm_rp[s] = (float)((double)((const GAS float*)A.nm.val)[row * A.nm.nb_max + bin] + 0.0001);
        }
    }
    if (A.prm.stop_after == 10) { res.status = LMAT_ST_SILENT; store_result(out, res); return true; }  // timing experiments: staging only
    K4State S;
    k4_part1<LIN>(A.prm, res, S, cnt, score, NM ? score0 : nullptr, dep, sflags, tin, tout, reg, ord, lin, nT, cand, NM && nmt >= 0,
                  nm_rp, nm_cl, A.nm, keys);
    if (A.prm.stop_after == 11) { res.status = LMAT_ST_SILENT; store_result(out, res); return true; }  // ... + scores, sort, lineage
    uint32_t call_idx = A.phix_call_idx, ncand = 0, coff = 0;
    if (!S.done) {
        int nlin = S.nlin;
        const uint32_t high_tid = S.highest >= 0 ? reg[S.highest] : 0;
        const bool have_add = S.highest_depth != 0 && high_tid != 0;
        uint32_t high_tin = 0xFFFF, high_tout = 0xFFFF;
        if (have_add) {  // ancestors of the shallowest accepted node (:326-343)
            high_tin = tin[S.highest];
            high_tout = tout[S.highest];
            const u32x4 hf = g_facts16[high_tid];
            const uint32_t alen = hf.y & 0xFFFFu, aoff = hf.x;
            if (bail_on_long && (uint32_t)nlin + alen > (uint32_t)LIN) return false;
            const float fcand = (float)cand;
            bool trunc = false;
            for (uint32_t j0 = 0; j0 < alen && !trunc; j0 += 8) {  // the chain in batches of eight loads
                uint64_t pes[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) pes[j] = j0 + j < alen ? g_paths8[aoff + j0 + j] : 0ull;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (j0 + j < alen && !trunc) {
                        if (nlin >= LIN) { G_OR(g_err, (uint32_t)kErrLineageTrunc); trunc = true; }
                        else {
                            const uint64_t pe = pes[j];
                            const uint32_t a = (uint32_t)(pe & 0xFFFFu);
                            int sl = -1;
                            for (int s = 0; s < nT; ++s) if (reg[s] == a) sl = s;
                            LinEnt en;
                            en.tid = (uint16_t)a;
                            // pre-bias score of a registered ancestor (all_cand_set, :821), -10000 otherwise
                            en.score = sl >= 0 ? (NM ? score0[sl] : (float)cnt[sl] / fcand) : -10000.0f;
                            en.dep = (uint16_t)(pe >> 16); en.tin = (uint16_t)(pe >> 32); en.tout = (uint16_t)(pe >> 48);
                            lin[nlin++] = en;
                        }
                    }
                }
            }
        }
        if (A.prm.stop_after == 12) { res.status = LMAT_ST_SILENT; store_result(out, res); return true; }  // ... + appended ancestors
        GAS lmat_cand* cout_ = nullptr;
        if (A.cands) {
            const uint32_t reserve = A.prm.prn_all ? (uint32_t)nT : (uint32_t)nlin;  // -p: the candidates; else the lineage as built
            coff = G_ADD(&g_cursor[0], reserve);
            if ((uint64_t)coff + reserve <= A.cand_cap) cout_ = (GAS lmat_cand*)A.cands + coff;
            else G_OR(g_err, (uint32_t)kErrCandOverflow);
        }
        k4_part2<LIN>(A.prm, g_tid32, res, S, score, tin, tout, reg, ord, lin, nlin, nT, have_add, high_tin, high_tout, cout_,
                 &ncand, &call_idx);
    }
    res.cand_off = coff;
    res.n_cand = ncand;
    store_result(out, res);
    if (res.status != LMAT_ST_PHIX && res.match_type == LMAT_MT_NOMATCH) {
        G_ADD(&tally_nomatch[1], 1ull);
    } else if (res.call_score >= A.prm.min_score) {
        G_ADD(&tally_count[call_idx], 1ull);
        G_ADD(&tally_score[call_idx], (double)res.call_score);
    } else if (res.call_score < A.prm.min_score) {
        G_ADD(&tally_nomatch[2], 1ull);
    }
    return true;
}

// pending reads of the batch -> index lists by table size and path (status 254: small, 252: mid, 253: large -- the general path;
// 251: up to 16 taxids, decided by k4_row_kernel); counts in cursor[4], cursor[8], cursor[5], cursor[9]
__global__ __launch_bounds__(256) void k4_compact_kernel(ClassifyArgs A) {
    GAS uint32_t* g_cursor = (GAS uint32_t*)A.cursor;
    const int lane = threadIdx.x & 63;
    constexpr int K = 16;  // 64 x K reads per wave and pass: one atomic per class for 1024 reads
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t base = wave * 64 * K; base < A.count; base += nwaves * 64 * K) {
        uint32_t st[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const uint64_t it = base + (uint64_t)j * 64 + lane;
            st[j] = it < A.count ? (uint32_t)*(const GAS uint8_t*)(A.results + it) : 0u;  // status is the record's first byte
        }
        for (int cls = 0; cls < 4; ++cls) {
            const uint32_t want = cls == 0 ? 254u : (cls == 1 ? 252u : (cls == 2 ? 253u : 251u));
            uint32_t total = 0;
#pragma unroll
            for (int j = 0; j < K; ++j) total += (uint32_t)popc64(__ballot(st[j] == want));
            if (!total) continue;
            uint32_t pos = 0;
            if (lane == 0) pos = G_ADD(&g_cursor[cls == 0 ? 4 : (cls == 1 ? 8 : (cls == 2 ? 5 : 9))], total);
            pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos);
            GAS uint32_t* list = (GAS uint32_t*)(cls == 0 ? A.k4_small : (cls == 1 ? A.k4_mid : (cls == 2 ? A.k4_large : A.k4_row)));
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const uint64_t m = __ballot(st[j] == want);
                if (st[j] == want) list[pos + prefix_count(m)] = (uint32_t)(base + (uint64_t)j * 64 + lane);
                pos += (uint32_t)popc64(m);
            }
        }
    }
}

template <int TT>
__global__ __launch_bounds__(64) void k4_lds_kernel(ClassifyArgs A) {
    using K = K4Lds<TT>;
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    if (lane >= K::LANES) return;
    unsigned char* blk = smem + (size_t)lane * K::STRIDE * 4;
    uint16_t* reg = (uint16_t*)blk;
    uint16_t* cnt = reg + TT;
    uint16_t* dep = cnt + TT;
    uint16_t* tin = dep + TT;
    uint16_t* tout = tin + TT;
    uint16_t* ord = tout + TT;
    float* score = (float*)(ord + TT);
    LinEnt* lin = (LinEnt*)(score + TT);
    uint8_t* sflags = (uint8_t*)(lin + K::LIN);
    GAS uint32_t* g_cursor = (GAS uint32_t*)A.cursor;
    const uint64_t n = *(const GAS uint32_t*)(g_cursor + (TT == kK4SmallT ? 4 : (TT == kK4MidT ? 8 : 5)));
    const GAS uint32_t* list = (const GAS uint32_t*)(TT == kK4SmallT ? A.k4_small : (TT == kK4MidT ? A.k4_mid : A.k4_large));
    const uint64_t stride = (uint64_t)gridDim.x * K::LANES;
    for (uint64_t i = (uint64_t)blockIdx.x * K::LANES + lane; i < n; i += stride) {
        const uint64_t it = list[i];
        // the sort keys lie over the lineage entries, which are written only after the sort
        if (!k4_read<TT, K::LIN, false>(A, it, reg, cnt, dep, tin, tout, ord, sflags, score, nullptr, nullptr, nullptr, lin, true, (K4Key*)lin))
            ((GAS uint32_t*)A.k4_bail)[G_ADD(&g_cursor[6], 1u)] = (uint32_t)it;  // lineage longer than the LDS block: scratch kernel, last launch
    }
}

// LANES < 64: fewer reads per wave.  A wave runs as long as its slowest lane at every step of the nested loops, so at larger
// tables half-filled waves finish sooner (and there are registers and scratch for all of them to be resident at once).
template <bool NM, int TT, int LANES>
__global__ __launch_bounds__(64, 8) void k4_kernel(ClassifyArgs A) {
    constexpr int LIN = TT == kK4T ? kK4T + 72 : TT + 8;
    if ((int)threadIdx.x >= LANES) return;
    GAS uint32_t* g_cursor = (GAS uint32_t*)A.cursor;
    const uint64_t n = *(const GAS uint32_t*)(g_cursor + A.k4_slot);  // 5: the large-table list, 8: up to 32 taxids, 6: reads an LDS/short kernel passed on
    const GAS uint32_t* list = (const GAS uint32_t*)(A.k4_slot == 5 ? A.k4_large : (A.k4_slot == 8 ? A.k4_mid : A.k4_bail));
    const uint64_t stride = (uint64_t)gridDim.x * LANES;
    for (uint64_t i = (uint64_t)blockIdx.x * LANES + threadIdx.x; i < n; i += stride) {
        const uint64_t it = list[i];
        uint16_t reg[TT], cnt[TT], dep[TT], tin[TT], tout[TT], ord[TT];
        uint8_t sflags[TT], nm_cl[NM ? TT : 1];
        float score[TT], score0[NM ? TT : 1], nm_rp[NM ? TT : 1];
        LinEnt lin[LIN];
        K4Key keys[TT];
        if (!k4_read<TT, LIN, NM>(A, it, reg, cnt, dep, tin, tout, ord, sflags, score, score0, nm_cl, nm_rp, lin, TT != kK4T, keys))
            ((GAS uint32_t*)A.k4_bail)[G_ADD(&g_cursor[6], 1u)] = (uint32_t)it;
    }
}

// lmat_debug_decide: the decision kernels' own code (k4_part1 from the sort on, the ancestors of the shallowest lineage
// member, k4_part2) on candidate tables given from outside -- taxid + score per candidate and the read's standard deviation,
// e.g. what a reference run printed.  One lane per record, tables in private memory like k4_kernel.
__global__ __launch_bounds__(64) void k4_debug_kernel(ClassifyArgs A, const uint32_t* __restrict__ idx, const float* __restrict__ scores,
                                                      const uint64_t* __restrict__ off, const float* __restrict__ stdevs, uint64_t n) {
    constexpr int TT = kK4T, LIN = kK4T + 72;
    const uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const GAS uint32_t* g_tid32 = (const GAS uint32_t*)A.tb.tid32;
    const GAS uint64_t* g_paths8 = (const GAS uint64_t*)A.tb.paths8;
    const GAS u32x4* g_facts16 = (const GAS u32x4*)A.tb.facts16;
    uint16_t reg[TT], cnt[TT], dep[TT], tin[TT], tout[TT], ord[TT];
    uint8_t sflags[TT];
    float score[TT], score0[TT];
    LinEnt lin[LIN];
    K4Key keys[TT];
    lmat_read_result res;
    res.status = 255; res.match_type = LMAT_MT_NOMATCH; res.cand_kmer_cnt = 0; res.valid_kmers = 0; res.read_len = 0; res.log_avg = 0;
    res.stdev = 0; res.call_tid = 0; res.call_score = 0; res.cand_off = 0; res.n_cand = 0; res.bin_sel = 0;
    const int nT = (int)(off[i + 1] - off[i]);
    if (nT >= 1 && nT <= TT) {
        for (int s = 0; s < nT; ++s) {
            const uint32_t t = idx[off[i] + s];
            const u32x4 f = g_facts16[t];
            reg[s] = (uint16_t)t; cnt[s] = 0; score[s] = score0[s] = scores[off[i] + s];
            dep[s] = (uint16_t)(f.w & 0xFFFFu); sflags[s] = (uint8_t)(f.w >> 16); tin[s] = (uint16_t)(f.z & 0xFFFFu); tout[s] = (uint16_t)(f.z >> 16);
        }
        const float sd = stdevs[i];
        K4State S;
        k4_part1<LIN>(A.prm, res, S, cnt, score, nullptr, dep, sflags, tin, tout, reg, ord, lin, nT, (uint32_t)nT, false, nullptr, nullptr, A.nm, keys, &sd);
        int nlin = S.nlin;
        const uint32_t high_tid = S.highest >= 0 ? reg[S.highest] : 0;
        const bool have_add = S.highest_depth != 0 && high_tid != 0;
        uint32_t high_tin = 0xFFFF, high_tout = 0xFFFF;
        bool trunc = false;
        if (have_add) {  // ancestors of the shallowest accepted node with their all_cand_set scores or -10000 (:326-343)
            high_tin = tin[S.highest];
            high_tout = tout[S.highest];
            const u32x4 hf = g_facts16[high_tid];
            const uint32_t alen = hf.y & 0xFFFFu, aoff = hf.x;
            for (uint32_t j = 0; j < alen && !trunc; ++j) {
                if (nlin >= LIN) { trunc = true; break; }
                const uint64_t pe = g_paths8[aoff + j];
                const uint32_t a = (uint32_t)(pe & 0xFFFFu);
                int sl = -1;
                for (int s = 0; s < nT; ++s) if (reg[s] == a) sl = s;
                LinEnt en;
                en.tid = (uint16_t)a;
                en.score = sl >= 0 ? score0[sl] : -10000.0f;
                en.dep = (uint16_t)(pe >> 16); en.tin = (uint16_t)(pe >> 32); en.tout = (uint16_t)(pe >> 48);
                lin[nlin++] = en;
            }
        }
        if (!trunc) {
            uint32_t ncand = 0, call_idx = 0;
            k4_part2<LIN>(A.prm, g_tid32, res, S, score, tin, tout, reg, ord, lin, nlin, nT, have_add, high_tin, high_tout, nullptr, &ncand, &call_idx);
        } else res.status = 255;
    }
    store_result((GAS uint64_t*)(A.results + i), res);
}

// lmat_debug_decide_counts, general path: k4_part1 / k4_part2 exactly as k4_kernel runs them for a read -- scores from counts,
// statistics, std::sort(TCmp) replayed, findReadLabelVer2 -- on (taxid, count) tables given from outside.  One lane per table.
__global__ __launch_bounds__(64) void k4_debug_counts_kernel(ClassifyArgs A, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ cnts,
                                                             const uint64_t* __restrict__ off, const uint32_t* __restrict__ cands, uint64_t n) {
    constexpr int TT = kK4T, LIN = kK4T + 72;
    const uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const GAS uint32_t* g_tid32 = (const GAS uint32_t*)A.tb.tid32;
    const GAS uint64_t* g_paths8 = (const GAS uint64_t*)A.tb.paths8;
    const GAS u32x4* g_facts16 = (const GAS u32x4*)A.tb.facts16;
    uint16_t reg[TT], cnt[TT], dep[TT], tin[TT], tout[TT], ord[TT];
    uint8_t sflags[TT];
    float score[TT], score0[TT];
    LinEnt lin[LIN];
    K4Key keys[TT];
    lmat_read_result res;
    res.status = 255; res.match_type = LMAT_MT_NOMATCH; res.cand_kmer_cnt = 0; res.valid_kmers = 0; res.read_len = 0; res.log_avg = 0;
    res.stdev = 0; res.call_tid = 0; res.call_score = 0; res.cand_off = 0; res.n_cand = 0; res.bin_sel = 0;
    const int nT = (int)(off[i + 1] - off[i]);
    const uint32_t cand = cands[i];
    if (nT >= 1 && nT <= TT) {
        for (int s = 0; s < nT; ++s) {
            const uint32_t t = idx[off[i] + s];
            const u32x4 f = g_facts16[t];
            reg[s] = (uint16_t)t; cnt[s] = (uint16_t)cnts[off[i] + s];
            dep[s] = (uint16_t)(f.w & 0xFFFFu); sflags[s] = (uint8_t)(f.w >> 16); tin[s] = (uint16_t)(f.z & 0xFFFFu); tout[s] = (uint16_t)(f.z >> 16);
        }
        K4State S;
        k4_part1<LIN>(A.prm, res, S, cnt, score, score0, dep, sflags, tin, tout, reg, ord, lin, nT, cand, false, nullptr, nullptr, A.nm, keys);
        if (!S.done) {
            int nlin = S.nlin;
            const uint32_t high_tid = S.highest >= 0 ? reg[S.highest] : 0;
            const bool have_add = S.highest_depth != 0 && high_tid != 0;
            uint32_t high_tin = 0xFFFF, high_tout = 0xFFFF;
            bool trunc = false;
            if (have_add) {  // ancestors of the shallowest accepted node with their all_cand_set scores or -10000 (:326-343)
                high_tin = tin[S.highest];
                high_tout = tout[S.highest];
                const u32x4 hf = g_facts16[high_tid];
                const uint32_t alen = hf.y & 0xFFFFu, aoff = hf.x;
                for (uint32_t j = 0; j < alen && !trunc; ++j) {
                    if (nlin >= LIN) { trunc = true; break; }
                    const uint64_t pe = g_paths8[aoff + j];
                    const uint32_t a = (uint32_t)(pe & 0xFFFFu);
                    int sl = -1;
                    for (int s = 0; s < nT; ++s) if (reg[s] == a) sl = s;
                    LinEnt en;
                    en.tid = (uint16_t)a;
                    en.score = sl >= 0 ? score0[sl] : -10000.0f;
                    en.dep = (uint16_t)(pe >> 16); en.tin = (uint16_t)(pe >> 32); en.tout = (uint16_t)(pe >> 48);
                    lin[nlin++] = en;
                }
            }
            if (!trunc) {
                uint32_t ncand = 0, call_idx = 0;
                k4_part2<LIN>(A.prm, g_tid32, res, S, score, tin, tout, reg, ord, lin, nlin, nT, have_add, high_tin, high_tout, nullptr, &ncand, &call_idx);
            } else res.status = 255;
        }
    }
    store_result((GAS uint64_t*)(A.results + i), res);
}

// ... and the path the benchmark runs: k4_wave, the decision step on the classify wave, on the same tables.  One wave per table,
// set up as classify_one leaves a read: lane s = registration slot s with the id's fact record in its registers, the ids in the
// wave's taxid hash.  A table k4_wave declines (its preconditions, the heapsort turn of introsort, a lineage beyond 64 lanes,
// two lineage entries of one depth) comes back with status 255: the general path's job.
__global__ __launch_bounds__(64) void k4_wave_debug_kernel(ClassifyArgs A, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ cnts,
                                                           const uint64_t* __restrict__ off, const uint32_t* __restrict__ cands, uint64_t n) {
    constexpr int THM = 255;
    __shared__ __align__(16) unsigned int hent[THM + 1];
    __shared__ __align__(16) uint32_t xch[384 + 64];
    const int lane = threadIdx.x & 63;
    for (uint64_t i = blockIdx.x; i < n; i += gridDim.x) {
        const uint32_t nT = (uint32_t)(off[i + 1] - off[i]);
        GAS uint64_t* out = (GAS uint64_t*)(A.results + i);
        bool done = false;
        if (nT >= 1 && nT <= 64) {
            for (int h = lane; h <= THM; h += 64) hent[h] = 0;
            WSYNC();
            const bool act = (uint32_t)lane < nT;
            const uint32_t my_id = act ? idx[off[i] + lane] : 0u, my_cnt = act ? cnts[off[i] + lane] : 0u;
            u32x4 fz = u32x4{0u, 0u, 0u, 0u};
            if (act) {
                fz = ((const GAS u32x4*)A.tb.facts16)[my_id];
                const uint32_t h = tid_find_or_claim(hent, THM, my_id);
                hent[h] = my_id | ((uint32_t)lane << 16);
            }
            WSYNC();
            done = k4_wave<THM>((CArgsK4*)__builtin_amdgcn_kernarg_segment_ptr(), lane, nT, cands[i], fz, my_cnt, my_id, hent, xch, out, 0, 0u, 0);
            WSYNC();
        }
        if (!done && lane == 0) {
            lmat_read_result res;
            res.status = 255; res.match_type = LMAT_MT_NOMATCH; res.cand_kmer_cnt = 0; res.valid_kmers = 0; res.read_len = 0; res.log_avg = 0;
            res.stdev = 0; res.call_tid = 0; res.call_score = 0; res.cand_off = 0; res.n_cand = 0; res.bin_sel = 0;
            store_result(out, res);
        }
    }
}

// Random 64-byte bucket gather with the probe's access shape (4 lanes x 16 B per bucket, 9 wave-loads = 144 buckets
// in flight per wave): the practical ceiling of K2 on this table, and a known byte count to calibrate the
// FETCH_SIZE counter against.
template <int LPB>  // lanes per probe: 4 = one 64-byte bucket (the probe's shape), 8 = an aligned 128-byte pair of buckets
__global__ __launch_bounds__(64) void gather_bench_kernel(const uint64_t* __restrict__ slots, uint32_t nbuckets,
                                                          uint64_t probes_per_wave, uint64_t seed,
                                                          unsigned long long* sink) {
    const int lane = threadIdx.x & 63, g = lane / LPB, q = lane % LPB;
    constexpr int PPL = 64 / LPB;  // probes per wave-load
    const GAS u32x4* quarters = (const GAS u32x4*)slots;
    unsigned long long acc = 0;
    uint64_t ctr = (uint64_t)blockIdx.x * probes_per_wave;
    for (uint64_t i = 0; i < probes_per_wave; i += 9 * PPL) {
        u32x4 sl[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            uint32_t b = bucket_of(seed + ctr + i + j * PPL + g, nbuckets);
            if (LPB == 8) b &= ~1u;
            sl[j] = quarters[(uint64_t)b * 4 + q];
        }
#pragma unroll
        for (int j = 0; j < 9; ++j) acc += (sl[j].x >> 13) + sl[j].w;
    }
    if (acc == 0x123456789ull) atomicAdd(sink, acc);
}

#pragma push_macro("WSYNC")
#undef WSYNC
#define WSYNC() wsync<(U > 2048)>()
// resident waves per SIMD each class is compiled for: what its LDS footprint allows (and no more registers than that needs)
#ifndef LMAT_FAST_WAVES
#define LMAT_FAST_WAVES 8   // (A/B builds: -DLMAT_FAST_WAVES=7 compiles the fast classes for 7 waves per SIMD, 72 registers)
#endif
constexpr int classify_waves(int U, int E, bool INK4, bool CPT) { return INK4 ? 1 : (E > kFastE ? (U <= 160 ? 5 : 2) : (U <= 160 ? (CPT ? LMAT_FAST_WAVES : 5) : (U <= 256 ? (CPT ? LMAT_FAST_WAVES : 5) : (U <= 320 ? (CPT ? 5 : 3) : (CPT ? 4 : 3))))); }
template <int U, int T, int E, bool INK4, bool PERM, bool CPT, bool WIDE = false>
__global__ __launch_bounds__(64, classify_waves(U, E, INK4, CPT)) void classify_kernel(ClassifyArgs A) {
    extern __shared__ __align__(16) unsigned char lds_smem[];
    // U > 2048: the per-read tables of this workgroup live in global memory
    unsigned char* smem = U > 2048 ? A.gscratch + (size_t)blockIdx.x * WL<U, T, E, INK4, false, WIDE>::BYTES : lds_smem;
    // scratch block of the compact-layout probe: always real LDS
    LAS unsigned char* xl = (LAS unsigned char*)(U > 2048 ? lds_smem : lds_smem + WL<U, T, E, INK4, CPT, WIDE>::OFF_XL);
    const int lane = threadIdx.x & 63;
    // Software pipeline over the reads of this wave: the record offset is fetched two reads ahead and the
    // record words one read ahead, so a read never starts with a chain of dependent HBM round trips.
    // What the loop needs from the kernel arguments (read count, list and record pointers) is read again in every trip -- scalar
    // loads from the argument segment -- instead of being carried across classify_one, where every carried scalar is one more
    // value spilled to a lane of a vector register around the decision step (the kernel has 78 scalar registers).
    constexpr int NW = (WL<U, T, E, INK4, CPT, WIDE>::RD_WORDS + 1 + 63) / 64;
    const uint32_t G = gridDim.x;
    uint32_t it = blockIdx.x;
    uint32_t wcur[NW], wnext[NW];
    uint32_t nmacc[2] = {0, 0};
    uint64_t off1 = 0;
    {
        const uint64_t count = A.count_ptr ? (uint64_t)*(const GAS uint32_t*)A.count_ptr : A.count;
        if (it >= count) return;
        const GAS uint32_t* index = (const GAS uint32_t*)A.index;
        const GAS uint64_t* rec_off = (const GAS uint64_t*)A.rec_off;
        const GAS uint32_t* words = (const GAS uint32_t*)A.words;
        auto r_of = [&](uint64_t i) -> uint64_t { return index ? (uint64_t)index[i] : A.first + i; };
        const uint64_t off0 = rec_off[r_of(it)];
        off1 = (uint64_t)it + G < count ? rec_off[r_of((uint64_t)it + G)] : 0;
#pragma unroll
        for (int j = 0; j < NW; ++j) wcur[j] = words[off0 + lane + 64 * j];  // may run past the record: the buffer is padded
    }
    while (true) {
        CArgs* Ap = (CArgs*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(Ap));  // (a fresh look at the arguments every trip: nothing derived from them is carried around the loop)
        const uint64_t count = Ap->count_ptr ? (uint64_t)*(const GAS uint32_t*)Ap->count_ptr : Ap->count;  // (it < count here: checked before the first trip and by `more` after that)
        const GAS uint32_t* index = (const GAS uint32_t*)Ap->index;
        const GAS uint64_t* rec_off = (const GAS uint64_t*)Ap->rec_off;
        const GAS uint32_t* words = (const GAS uint32_t*)Ap->words;
        auto r_of = [&](uint64_t i) -> uint64_t { return index ? (uint64_t)index[i] : Ap->first + i; };
        const uint64_t off2 = (uint64_t)it + 2 * (uint64_t)G < count ? rec_off[r_of((uint64_t)it + 2 * (uint64_t)G)] : 0;
        const bool more = (uint64_t)it + G < count;
#pragma unroll
        for (int j = 0; j < NW; ++j) wnext[j] = more ? words[off1 + lane + 64 * j] : 0u;
        classify_one<U, T, E, INK4, PERM, CPT, WIDE>(Ap, r_of(it), smem, xl, lane, wcur, nmacc);
        WSYNC();
#pragma unroll
        for (int j = 0; j < NW; ++j) wcur[j] = wnext[j];
        off1 = off2;
        if (!more) break;
        it += G;  // (below count <= 2^32 - 1: no wrap)
    }
    if (lane == 0) {  // ReadTooShort / NoDbHits counts of this wave's reads, one atomic each
        GAS unsigned long long* tally_nomatch = (GAS unsigned long long*)A.counts + 2 * (uint64_t)A.tb.n_ids;
        if (nmacc[0]) G_ADD(&tally_nomatch[0], (unsigned long long)nmacc[0]);
        if (nmacc[1]) G_ADD(&tally_nomatch[1], (unsigned long long)nmacc[1]);
    }
}

#pragma pop_macro("WSYNC")

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
// hipFuncSetAttribute applies to the CURRENT device's copy of a kernel: one process drives one context per GPU, each from its
// own thread, so "already raised" is kept per device (an atomic flag per device; setting it twice is harmless).
struct PerDeviceOnce {
    std::atomic<unsigned char> done[64];
    PerDeviceOnce() { for (auto& d : done) d.store(0); }
    template <class F> void operator()(F&& f) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { f(); return; }
        if (!done[dev].load(std::memory_order_acquire)) { f(); done[dev].store(1, std::memory_order_release); }
    }
};

static int grid_for(uint64_t n, int per_block, int cap) {
    uint64_t b = (n + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > (uint64_t)cap) b = cap;
    return (int)b;
}

// a batch of equal-length reads laid end to end: both offset arrays follow from the length (nothing to copy in)
__global__ __launch_bounds__(256) void fill_offsets_kernel(uint64_t* __restrict__ off, uint64_t* __restrict__ rec_off, uint64_t n,
                                                           uint32_t len, uint32_t words_per_rec) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i <= n) { off[i] = i * len; rec_off[i] = i * words_per_rec; }
}
void launch_fill_offsets(uint64_t* off, uint64_t* rec_off, uint64_t n, uint32_t len, hipStream_t stream) {
    fill_offsets_kernel<<<dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, stream>>>(off, rec_off, n, len, rec_words(len));
}
void launch_pack_reads(const uint8_t* bases, const uint64_t* off, const uint64_t* rec_off, uint32_t* words, uint64_t n,
                       hipStream_t stream) {
    hipLaunchKernelGGL(pack_reads_kernel, dim3(grid_for(n, 16, 16384)), dim3(256), 0, stream, bases, off, rec_off, words, n);  // 4 waves x 4 reads per block and pass
}
void launch_insert_pairs(const DeviceTables& tb, const uint64_t* kmers, const uint32_t* payload, uint64_t n,
                         uint32_t* fail, hipStream_t stream) {
    hipLaunchKernelGGL(insert_pairs_kernel, dim3(grid_for(n, 256, 16384)), dim3(256), 0, stream, tb, kmers, payload, n, fail);
}
void launch_synth_db(const DeviceTables& tb, const SynthGeo& g, int k, const uint16_t* strain_idx, const uint32_t* list_payload,
                     uint64_t g_off, uint32_t rep, uint32_t rep_stride, uint32_t* fail, unsigned long long* inserted, hipStream_t stream) {
    const uint64_t total = (uint64_t)g.n_species * (g.G - k + 1);
    hipLaunchKernelGGL(synth_db_kernel, dim3(grid_for(total, 256, 65536)), dim3(256), 0, stream, tb, g, k, strain_idx, list_payload,
                       g_off, rep, rep_stride, fail, inserted);
}
// number of k-mers in the table -> *out (after a build; the compact layout is tidied on the way)
void launch_table_count(const DeviceTables& tb, unsigned long long* out, hipStream_t stream) {
    if (!tb.cpt.nb) {
        const uint64_t nslots = (uint64_t)tb.nbuckets * kSlotsPerBucket;
        hipLaunchKernelGGL(count_slots_kernel, dim3(grid_for(nslots, 256, 16384)), dim3(256), 0, stream, tb.slots, nslots, out);
        return;
    }
    hipLaunchKernelGGL(cpt_tidy_kernel, dim3(grid_for(tb.cpt.nb, 256, 16384)), dim3(256), 0, stream, tb, out);
    hipLaunchKernelGGL(cpt_merge_overflow_kernel, dim3(grid_for((uint64_t)tb.ovf_nbuckets * kSlotsPerBucket, 256, 16384)), dim3(256), 0, stream, tb, out);
}
void launch_synth_reads(uint32_t* words, const uint64_t* rec_off, const uint32_t* lengths, uint32_t n_lengths, uint64_t n,
                        uint64_t seed, const SynthGeo& g, hipStream_t stream) {
    hipLaunchKernelGGL(synth_reads_kernel, dim3(grid_for(n, 4, 16384)), dim3(256), 0, stream, words, rec_off, lengths,
                       n_lengths, n, seed, g);
}
void launch_lookup(const DeviceTables& tb, const uint64_t* kmers, uint64_t n, uint32_t* counts, uint32_t* tids,
                   uint32_t stride, hipStream_t stream) {
    if (!n) return;
    hipLaunchKernelGGL(lookup_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, tb, kmers, n, counts, tids,
                       stride);
}

void launch_div_check(unsigned long long* out2, hipStream_t stream) { div_check_kernel<<<dim3(1024, 4), dim3(256), 0, stream>>>(out2); }

void launch_probe_stats(const DeviceTables& tb, const uint64_t* kmers, uint64_t n, unsigned long long* out, hipStream_t stream) {
    if (!n) return;
    probe_stats_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(tb, kmers, n, out);
}

void launch_gather_bench(const uint64_t* slots, uint32_t nbuckets, uint64_t n_probes, uint64_t seed,
                         unsigned long long* sink, hipStream_t stream, int bytes_per_probe) {
    const int grid = 256 * 16;
    if (bytes_per_probe == 128) {
        const uint64_t per_wave = (n_probes / grid + 71) / 72 * 72;
        gather_bench_kernel<8><<<dim3(grid), dim3(64), 0, stream>>>(slots, nbuckets, per_wave, seed, sink);
    } else {
        const uint64_t per_wave = (n_probes / grid + 143) / 144 * 144;
        gather_bench_kernel<4><<<dim3(grid), dim3(64), 0, stream>>>(slots, nbuckets, per_wave, seed, sink);
    }
}

// The K4 kernels of a batch, by table size, side by side: up to 16 taxids in LDS on `stream`, 17..32 in scratch memory on
// `stream2`, 33..64 in LDS (half waves) on `stream3`; the scratch kernel alone when null models are loaded (its extra tables do not fit the LDS
// blocks), and once more at the end for the few reads a tier passed on (lineage longer than its block).
template <int TT>
static void launch_k4_lds(const ClassifyArgs& a, uint64_t max_reads, hipStream_t stream) {
    using K = K4Lds<TT>;
    static PerDeviceOnce attr_once;
    attr_once([] { hipFuncSetAttribute((const void*)k4_lds_kernel<TT>, hipFuncAttributeMaxDynamicSharedMemorySize, K::BYTES); });
    const uint64_t per_cu = 160 * 1024 / K::BYTES;
    uint64_t g = (max_reads + K::LANES - 1) / K::LANES;
    if (g > 256 * per_cu) g = 256 * per_cu;
    static const int gcap = getenv("LMAT_K4_SMALL_GRID") ? atoi(getenv("LMAT_K4_SMALL_GRID")) : 0;  // experiments
    if (TT == kK4SmallT && gcap > 0 && g > (uint64_t)gcap) g = gcap;
    if (g < 1) g = 1;
    k4_lds_kernel<TT><<<dim3((unsigned)g), dim3(64), K::BYTES, stream>>>(a);
}
void launch_k4_begin(const ClassifyArgs& a, hipStream_t stream, hipStream_t stream2, hipStream_t stream3, hipStream_t small_stream,
                     hipEvent_t forked) {
    uint64_t blocks = (a.count + 4095) / 4096;  // a wave takes 1024 reads per pass
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    k4_compact_kernel<<<dim3((unsigned)blocks), dim3(256), 0, stream>>>(a);
    hipEventRecord(forked, stream);
    hipStreamWaitEvent(stream2, forked, 0);
    ClassifyArgs b = a;
    b.k4_slot = 5;
    const uint64_t waves = (a.count + 63) / 64;
    uint64_t g2 = waves < 256 * 32 ? waves : 256 * 32;
    if (g2 < 1) g2 = 1;
    static const int g2cap = getenv("LMAT_K4_MID_GRID") ? atoi(getenv("LMAT_K4_MID_GRID")) : 0;  // experiments
    if (g2cap > 0 && g2 > (uint64_t)g2cap) g2 = g2cap;
    else if (small_stream == stream3 && g2 > 1024) g2 = 1024;  // beside the next batch's classify kernel (LMAT_PIPELINE): take fewer wave slots
    // tables of 17..32 taxids: one lane per read with the tables in scratch memory; every wave is resident at once, which
    // the LDS tier of that size (2 waves per CU, a millisecond per pass) cannot offer.  LMAT_K4_MODE=1 runs it anyway.
    static const int mode = getenv("LMAT_K4_MODE") ? atoi(getenv("LMAT_K4_MODE")) : 0;
    if (a.nm.active) {
        k4_kernel<true, kK4T, 64><<<dim3((unsigned)g2), dim3(64), 0, stream2>>>(b);
    } else if (mode == 1) {
        launch_k4_lds<kK4MidT>(a, a.count, stream2);
    } else {
        b.k4_slot = 8;
        k4_kernel<false, kK4MidT, 64><<<dim3((unsigned)g2), dim3(64), 0, stream2>>>(b);
    }
    hipStreamWaitEvent(stream3, forked, 0);
    if (!a.nm.active && a.prm.k4_row) {  // LMAT_K4_ROW=1: tables of up to 16 taxids, four reads to a wave
        uint64_t gr = (a.count + 3) / 4;
        if (gr > 256 * 64) gr = 256 * 64;
        if (gr < 1) gr = 1;
        k4_row_kernel<<<dim3((unsigned)gr), dim3(64), 0, stream3>>>(a);
    }
    if (!a.nm.active) launch_k4_lds<kK4T>(a, a.count / 8 + 64, stream3);  // 33..64 taxids: few reads, but a pass over them is long
    if (small_stream != stream && small_stream != stream3) hipStreamWaitEvent(small_stream, forked, 0);
    launch_k4_lds<kK4SmallT>(a, a.count, small_stream);
}
// ... the caller may queue more work on stream3 (the re-run classes, which do their own K4) ...
// The tiers join on `join_stream` (one of the four), which then takes the few reads a tier passed on; `done` is recorded
// behind that.
void launch_k4_end(const ClassifyArgs& a, hipStream_t join_stream, hipStream_t stream2, hipStream_t stream3, hipStream_t small_stream,
                   hipEvent_t joined2, hipEvent_t joined3, hipEvent_t joined_small, hipEvent_t done) {
    if (stream2 != join_stream) { hipEventRecord(joined2, stream2); hipStreamWaitEvent(join_stream, joined2, 0); }
    if (stream3 != join_stream) { hipEventRecord(joined3, stream3); hipStreamWaitEvent(join_stream, joined3, 0); }
    if (small_stream != join_stream && small_stream != stream3 && small_stream != stream2) {
        hipEventRecord(joined_small, small_stream);
        hipStreamWaitEvent(join_stream, joined_small, 0);
    }
    ClassifyArgs b = a;
    b.k4_slot = 6;  // the few reads whose lineage outgrew a tier's block
    if (a.nm.active) k4_kernel<true, kK4T, 64><<<dim3(64), dim3(64), 0, join_stream>>>(b);
    else k4_kernel<false, kK4T, 64><<<dim3(64), dim3(64), 0, join_stream>>>(b);
    hipEventRecord(done, join_stream);
}

// Which decision paths the launch allows (bit 0: k4_wave, bit 1: k4_row_kernel), from what does not vary with the read.
static int k4_static_of(const ClassifyArgs& a) {
    const int stop = a.prm.stop_after;
    const bool base = !a.nm.active && a.tb.depth_consistent && !(a.cands && !a.prm.prn_all);
    const bool wave = base && (stop == 0 || (stop >= 30 && stop <= 34));
    const bool rows = base && a.prm.k4_row && stop == 0;
    return (wave ? 1 : 0) | (rows ? 2 : 0);
}
template <int U, int T, int E, bool INK4, bool PERM, bool CPT, bool WIDE = false>
static void launch_classify_t(const ClassifyArgs& a_in, hipStream_t stream) {
    using L = WL<U, T, E, INK4, CPT, WIDE>;
    ClassifyArgs a = a_in;
    a.prm.k4_static = k4_static_of(a);
    if (U > 2048) {  // tables in global memory: few workgroups; LDS only for the compact probe's scratch block
        int grid = kGmemGrid;
        if (!a.count_ptr && (uint64_t)grid > a.count) grid = (int)(a.count ? a.count : 1);
        classify_kernel<U, T, E, INK4, PERM, CPT, WIDE><<<dim3(grid), dim3(64), CPT ? L::XL_BYTES : 0, stream>>>(a);
        return;
    }
    static const int lds_pad = getenv("LMAT_LDS_PAD") ? atoi(getenv("LMAT_LDS_PAD")) : 0;  // experiments: fewer resident waves
    const int lds_bytes = L::BYTES + lds_pad;
    static PerDeviceOnce attr_once;
    attr_once([lds_bytes] { hipFuncSetAttribute((const void*)classify_kernel<U, T, E, INK4, PERM, CPT, WIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); });
    // one single-wave workgroup per read slot; enough groups to fill every CU's LDS several times over
    const int per_cu = 160 * 1024 / lds_bytes;
    // Reads differ in cost (one over genus-shared k-mers takes 3-4 times the usual), and a block keeps its share of the batch:
    // with two blocks per wave slot the last ones ran alone for a fifth of the kernel.  32 per slot (8 reads each at 2 M reads):
    // 6.66 -> 5.80 ms; beyond that the per-block start-up shows (LMAT_GRID_MULT to try).
    static const int gmult = getenv("LMAT_GRID_MULT") ? atoi(getenv("LMAT_GRID_MULT")) : 32;
    int grid = 256 * (per_cu < 1 ? 1 : (per_cu > 32 ? 32 : per_cu)) * (gmult > 0 ? gmult : 32);
    if (!a.count_ptr && (uint64_t)grid > a.count) grid = (int)a.count;
    if (a.count_ptr && INK4 && grid > 256 * (per_cu < 2 ? 2 : per_cu)) grid = 256 * (per_cu < 2 ? 2 : per_cu);  // the lists of the large classes are short (the E = 512 class may get a tenth of a batch): a block per wave the LDS holds
    if (grid < 1) grid = 1;
    classify_kernel<U, T, E, INK4, PERM, CPT, WIDE><<<dim3(grid), dim3(64), lds_bytes, stream>>>(a);
}

void launch_k4_debug(const ClassifyArgs& a, const uint32_t* idx, const float* scores, const uint64_t* off, const float* stdevs, uint64_t n,
                     hipStream_t stream) {
    if (!n) return;
    k4_debug_kernel<<<dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream>>>(a, idx, scores, off, stdevs, n);
}

void launch_k4_debug_counts(const ClassifyArgs& a, const uint32_t* idx, const uint32_t* cnts, const uint64_t* off, const uint32_t* cands, uint64_t n,
                            bool on_the_wave, hipStream_t stream) {
    if (!n) return;
    if (on_the_wave) {
        const uint64_t g = n < 256 * 32 ? n : 256 * 32;
        ClassifyArgs b = a;
        b.prm.k4_static = k4_static_of(b);
        k4_wave_debug_kernel<<<dim3((unsigned)g), dim3(64), 0, stream>>>(b, idx, cnts, off, cands, n);
    } else k4_debug_counts_kernel<<<dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream>>>(a, idx, cnts, off, cands, n);
}

void launch_tail(const ClassifyArgs& a, hipStream_t stream) {
    if (!a.tail_lpr || !a.count || a.count_ptr) return;
    const uint64_t lanes = a.count * a.tail_lpr;
    const dim3 grid((unsigned)((lanes + 255) / 256));
    if (a.tail_lpr == 4) tail_kernel<4><<<grid, dim3(256), 0, stream>>>(a);
    else if (a.tail_lpr == 8) tail_kernel<8><<<grid, dim3(256), 0, stream>>>(a);
    else tail_kernel<16><<<grid, dim3(256), 0, stream>>>(a);
}

int classify_max_read_len() { return kGmemU + 19; }
size_t classify_gmem_scratch_bytes() { return (size_t)kGmemGrid * WL<kGmemU, 4096, 16384, true, false, true>::BYTES; }   // (the wide layout: the larger of the two)

// permissive match (-s) is a compile-time variant: a run-time test of it inside the closure loops cost 22%
#define LC(U, T, E, K)                                                                                                   \
    (a.tb.cpt.nb ? (a.prm.permissive ? launch_classify_t<U, T, E, K, true, true>(a, stream) : launch_classify_t<U, T, E, K, false, true>(a, stream)) \
                 : (a.prm.permissive ? launch_classify_t<U, T, E, K, true, false>(a, stream) : launch_classify_t<U, T, E, K, false, false>(a, stream)))
// wide taxonomies (more than 65534 ids): 32-bit ids in the per-read tables, the classes that decide in-kernel only, compact layout
#define LCW(U, T, E) (a.prm.permissive ? launch_classify_t<U, T, E, true, true, true, true>(a, stream) : launch_classify_t<U, T, E, true, false, true, true>(a, stream))
bool launch_classify(const ClassifyArgs& a, uint32_t max_read_len, int tcap_class, hipStream_t stream) {
    const int k = a.tb.k;
    const uint32_t P = max_read_len >= (uint32_t)k ? max_read_len - k + 1 : 0;
    if (a.tb.wide) {
        // tcap_class 4: the first wide tier (128 taxids / 512 list elements: 30 KB of LDS, five waves per CU); 5: the second (512 /
        // 2048, one per CU); 1: tables in global memory (4096 / 16384).  Reads beyond 531 bp start in the second tier.
        if (!a.tb.cpt.nb) return false;
        if (tcap_class == 4) { if (P > 512) return false; LCW(512, 128, 512); }
        else if (tcap_class == 5) { if (P <= 512) LCW(512, 512, 2048); else if (P <= 2048) LCW(2048, 512, 2048); else return false; }
        else if (P <= (uint32_t)kGmemU && a.gscratch) LCW(kGmemU, 4096, 16384);
        else return false;
        return true;
    }
    if (tcap_class == 2) {  // reads whose kept lists add up to more than kFastE elements (many strains per k-mer): 512 of them
        if (P <= 160) LC(160, 64, 512, false); else LC(512, 64, 512, false);
    } else if (tcap_class == 3) {
        // the middle tier: up to 256 registered taxids / 1024 list elements, tables in LDS and the decision step on the wave's
        // first lane like the large class, but in 37 KB instead of 148: four waves per CU instead of one, and room left for the
        // fast classes beside it.  A near-capacity table returns chance hits for a read's erroneous k-mers (3.4 % of all 20-mers
        // are in a table of 18.6 G), each with a lineage of its own: 1 % of config 5's reads pass 64 taxids that way.
        if (P > 512) return false;
        LC(512, 256, 1024, true);
    } else if (P <= 160 && tcap_class == 0) {
        LC(160, 64, kFastE, false);
    } else if (P <= 256) {
        if (tcap_class == 0) LC(256, 64, kFastE, false); else LC(256, 1024, 4096, true);
    } else if (P <= 320 && tcap_class == 0) {
        LC(320, 64, kFastE, false);   // 257..320 k-mers (reads of up to 339 bp at k = 20): six chunks of registers instead of the nine of the 512 class
    } else if (P <= 512) {
        if (tcap_class == 0) LC(512, 64, kFastE, false); else LC(512, 1024, 4096, true);
    } else if (P <= 2048) {
        LC(2048, 1024, 4096, true);
    } else if (P <= (uint32_t)kGmemU && a.gscratch) {
        LC(kGmemU, 4096, 16384, true);
    } else {
        return false;
    }
    return true;
}

uint32_t synth_strain_base_host(const SynthGeo& g, uint32_t species, uint32_t strain_global, uint64_t pos) {
    return synth_strain_base(g, species, strain_global, pos);
}
// what synth_db_kernel files for the ancestor window at (species, pos): the canonical k-mer, the first strain that may carry
// it, and the mask of those strains (from first_strain on) whose copy of the window has no substitution
void synth_window_host(const SynthGeo& g, int k, uint32_t species, uint64_t pos, uint64_t* kmer, uint32_t* first_strain, uint32_t* mask, bool* inblk,
                       int* cons_level) {
    const int clv = synth_cons_window(g, pos, k);
    if (cons_level) *cons_level = clv;
    if (clv >= 0) {  // every strain of the group carries it: first_strain = the group's first, mask unused
        const uint32_t sp0 = species - species % g.csz[clv];
        uint64_t anc = 0;
        for (int j = 0; j < k; ++j) anc = (anc << 2) | synth_anc_base(g, sp0, pos + j);
        *kmer = canon_from_fwd(anc, k);
        *first_strain = sp0 * g.S;
        *mask = 0;
        *inblk = false;
        return;
    }
    const bool blk = pos + k <= g.blk;
    const uint32_t sp = blk ? species - species % g.spg : species;
    uint64_t anc = 0;
    for (int j = 0; j < k; ++j) anc = (anc << 2) | synth_anc_base(g, sp, pos + j);
    const uint32_t ns = blk ? g.spg * g.S : g.S;
    uint32_t m = 0;
    for (uint32_t s = 0; s < ns; ++s) {
        bool mut = false;
        for (int j = 0; j < k; ++j) mut |= synth_strain_mut(g, sp * g.S + s, pos + j);
        if (!mut) m |= 1u << s;
    }
    *kmer = canon_from_fwd(anc, k);
    *first_strain = sp * g.S;
    *mask = m;
    *inblk = blk;
}

// What synth_reads_kernel makes of read r, re-derived on the host: its length, and for every k-mer window that lies in a genome
// (a sampled read, kinds 11..99), holds no substituted base and not the N: where the window starts in the strain's genome (forward
// coordinates) -- the caller asks synth_window_host what the database must hold there.  Returns the number of such windows.
uint32_t synth_read_windows_host(const SynthGeo& g, const uint32_t* lengths, uint32_t n_lengths, uint64_t seed, uint64_t r, int k,
                                 uint32_t* len_out, uint32_t* strain_global, uint64_t* gpos, uint32_t* rpos, uint32_t cap) {
    const uint64_t h0 = splitmix(seed ^ (r * 0x9E3779B97F4A7C15ull));
    const uint32_t len = lengths[(uint32_t)((h0 >> 48) % n_lengths)];
    *len_out = len;
    const uint32_t kind = (uint32_t)(h0 % 100);
    const uint32_t sg = (uint32_t)((h0 >> 8) % ((uint64_t)g.n_species * g.S));
    *strain_global = sg;
    if (kind <= 10 || (int)len < k) return 0;  // random bases / low complexity: no genome behind the read
    const uint64_t maxoff = g.G > len ? g.G - len : 0;
    const uint64_t goff = (splitmix(h0) >> 4) % (maxoff + 1);
    const bool rc = (h0 >> 40) & 1;
    const uint32_t npos = (uint32_t)((splitmix(h0 ^ 77) >> 7) % len);
    std::vector<uint8_t> bad(len, 0);
    for (uint32_t p = 0; p < len; ++p) {
        const uint64_t e = splitmix(h0 ^ ((uint64_t)p << 24) ^ 0xABCDEF);
        const uint64_t gp = rc ? goff + (len - 1 - p) : goff + p;
        bad[p] = (e % 100) == 0 || (kind == 11 && p == npos) || gp >= g.G;
    }
    uint32_t n = 0, run = 0;
    for (uint32_t p = 0; p < len; ++p) {
        run = bad[p] ? 0 : run + 1;
        if (run >= (uint32_t)k && n < cap) {
            const uint32_t p0 = p + 1 - (uint32_t)k;   // window start in the read
            rpos[n] = p0;
            gpos[n] = rc ? goff + (len - 1 - p) : goff + p0;  // ... and in the genome (the window's lowest coordinate)
            ++n;
        }
    }
    return n;
}

}  // namespace lmat
