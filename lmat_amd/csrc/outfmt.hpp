// outfmt.hpp -- text of the per-read .out record and of the run summaries, byte for byte as
// the reference writes them:
//   record prefix  hdr \t (read | "X") \t            src/read_label.cpp:1733-1738
//   ReadTooShort / NoDbHits rows                     src/read_label.cpp:1218,1233,1271
//   PhiX row                                         src/read_label.cpp:844-848
//   stats, candidate list, call                      src/read_label.cpp:894-937
//   .fastsummary / .nomatchsum                       src/read_label.cpp:1836-1867
// Floats go through the same "%g" (precision 6) formatting as operator<<(float), applied to
// the exact float bits the device produced.
#pragma once
#include <charconv>
#include <cstdio>
#include <string>
#include "../../include/lmat_hip.h"

namespace lmat {

// operator<<(float) with the default precision is printf's "%g": 6 significant digits, correctly rounded, trailing zeros
// dropped, exponent form below 1e-4 and from 1e6.  libstdc++'s std::to_chars(general, 6) gives the same text but does
// not scale across threads (eight formatter threads ran no faster than one), and the CLI writes ~12 floats per read,
// so the common range is formatted here with exact integer arithmetic: the float is m * 2^e, and m * 2^e * 10^k is
// rounded half-to-even to an integer of 6 digits in 128-bit arithmetic.  Anything outside that range (and inf / nan)
// still goes through std::to_chars.  tests/test_host_logic.py checks the result against "%g" on a sweep of bit patterns.
inline int fmt_g6(char* out, float f) {
    uint32_t bits;
    __builtin_memcpy(&bits, &f, 4);
    const uint32_t ex = (bits >> 23) & 0xFF, man = bits & 0x7FFFFF;
    char* p = out;
    if (ex == 0xFF) return -1;
    if (bits >> 31) *p++ = '-';
    if (ex == 0 && man == 0) { *p++ = '0'; return (int)(p - out); }
    const uint64_t m = ex ? (man | 0x800000u) : man;
    const int e2 = (ex ? (int)ex : 1) - 150;               // value = m * 2^e2
    if (e2 < -90 || e2 > 40) return -1;                     // ~1e-20 .. ~1e19: everything a score or a statistic can be
    // decimal exponent d with 10^d <= value < 10^(d+1): estimate from the binary exponent, then correct
    const double v = (double)m * __builtin_ldexp(1.0, e2);
    int d = (int)__builtin_floor(__builtin_log10(v));
    static const uint64_t kPow10[20] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull, 100000000ull,
                                        1000000000ull, 10000000000ull, 100000000000ull, 1000000000000ull, 10000000000000ull,
                                        100000000000000ull, 1000000000000000ull, 10000000000000000ull, 100000000000000000ull,
                                        1000000000000000000ull, 10000000000000000000ull};
    unsigned __int128 q = 0;
    for (int attempt = 0; attempt < 3; ++attempt) {
        const int k = 5 - d;                                // digits = round(value * 10^k)
        if (k > 27 || k < -19) return -1;
        unsigned __int128 num = m, den = 1;
        // value * 10^k = m * 2^e2 * 2^k * 5^k
        if (k >= 0) { for (int i = 0; i < k; ++i) num *= 5; } else den = kPow10[-k] >> (-k), den = 1;
        int sh = e2 + k;                                    // remaining power of two
        if (k < 0) {                                        // divide by 10^-k = 5^-k * 2^-k
            den = 1;
            for (int i = 0; i < -k; ++i) den *= 5;
        }
        if (sh >= 0) num <<= sh; else den <<= -sh;
        q = num / den;
        const unsigned __int128 r = num - q * den;
        const unsigned __int128 twice = r << 1;
        if (twice > den || (twice == den && (q & 1))) ++q;
        if (q >= 1000000) { ++d; continue; }                // the estimate was one low, or rounding carried
        if (q < 100000) { --d; continue; }
        break;
    }
    if (q < 100000 || q >= 1000000) return -1;
    uint32_t dig = (uint32_t)q;
    char ds[6];
    for (int i = 5; i >= 0; --i) { ds[i] = (char)('0' + dig % 10); dig /= 10; }
    int nd = 6;
    while (nd > 1 && ds[nd - 1] == '0') --nd;               // %g drops trailing zeros
    if (d < -4 || d >= 6) {                                 // d.ddddde[+-]XX
        *p++ = ds[0];
        if (nd > 1) { *p++ = '.'; for (int i = 1; i < nd; ++i) *p++ = ds[i]; }
        *p++ = 'e';
        int a = d;
        if (a < 0) { *p++ = '-'; a = -a; } else *p++ = '+';
        if (a >= 100) { *p++ = (char)('0' + a / 100); a %= 100; }
        *p++ = (char)('0' + a / 10);
        *p++ = (char)('0' + a % 10);
    } else if (d >= 0) {                                    // ddd.ddd
        for (int i = 0; i <= d; ++i) *p++ = i < nd ? ds[i] : '0';
        if (nd > d + 1) { *p++ = '.'; for (int i = d + 1; i < nd; ++i) *p++ = ds[i]; }
    } else {                                                // 0.000ddd
        *p++ = '0'; *p++ = '.';
        for (int i = 0; i < -d - 1; ++i) *p++ = '0';
        for (int i = 0; i < nd; ++i) *p++ = ds[i];
    }
    return (int)(p - out);
}
inline int fmt_float(char* b, float f) {  // b: at least 48 bytes
    const int n = fmt_g6(b, f);
    if (n > 0) return n;
    return (int)(std::to_chars(b, b + 48, f, std::chars_format::general, 6).ptr - b);
}
// Scores are k-mer fractions: a few thousand distinct floats make up almost every score of a run, so each thread keeps
// the text of the floats it has formatted in a small direct-mapped table keyed by the float's bits.
// The record writers below work on a raw buffer (W): a record with its ~15 candidates is ~70 pieces of text, and a
// std::string append per piece -- size check, possible growth -- was most of the formatter's time.
struct W {
    char* p;
    inline void ch(char c) { *p++ = c; }
    inline void lit(const char* t, size_t n) { __builtin_memcpy(p, t, n); p += n; }
    template <size_t N> inline void lit(const char (&t)[N]) { __builtin_memcpy(p, t, N - 1); p += N - 1; }
    inline void u32(uint32_t v) {  // decimal, two digits a step
        static const char kPairs[201] =
            "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
        char b[10];
        int i = 10;
        while (v >= 100) { const uint32_t q = v / 100, r = v - q * 100; b[--i] = kPairs[2 * r + 1]; b[--i] = kPairs[2 * r]; v = q; }
        if (v >= 10) { b[--i] = kPairs[2 * v + 1]; b[--i] = kPairs[2 * v]; } else b[--i] = (char)('0' + v);
        __builtin_memcpy(p, b + i, (size_t)(10 - i));
        p += 10 - i;
    }
    inline void i64(long long v) {
        if (v >= 0 && v <= 0xFFFFFFFFll) { u32((uint32_t)v); return; }
        const auto r = std::to_chars(p, p + 24, v);
        p = r.ptr;
    }
    inline void flt(float f) {
        struct Ent { uint32_t bits; uint8_t len; char txt[19]; };
        static thread_local Ent cache[4096];
        static thread_local bool init = false;
        if (!init) { for (auto& e : cache) { e.bits = 0x7FC00001u; e.len = 0; } init = true; }  // a NaN payload no score has
        uint32_t bits;
        __builtin_memcpy(&bits, &f, 4);
        Ent& e = cache[(bits * 0x9E3779B1u) >> 20];
        if (e.bits == bits && e.len) { __builtin_memcpy(p, e.txt, sizeof e.txt); p += e.len; return; }  // (fixed-size copy: the buffer has the room)
        const int n = fmt_float(p, f);
        if (n <= (int)sizeof e.txt) { e.bits = bits; e.len = (uint8_t)n; __builtin_memcpy(e.txt, p, (size_t)n); }
        p += n;
    }
};
inline void put_float(std::string& s, float f) {
    char b[64];
    W w{b};
    w.flt(f);
    s.append(b, (size_t)(w.p - b));
}
inline void put_int(std::string& s, long long v) {
    char b[32];
    const auto r = std::to_chars(b, b + sizeof b, v);
    s.append(b, r.ptr);
}
inline const char* match_name(int m) {
    switch (m) {
        case LMAT_MT_DIRECT: return "DirectMatch";
        case LMAT_MT_MULTI: return "MultiMatch";
        case LMAT_MT_PARTIAL: return "PartialMultiMatch";
        case LMAT_MT_NOMATCH: return "NoMatch";
        case LMAT_MT_LCA_ERROR: return "LCA_ERROR";
    }
    return "error";
}

// Appends the part of the record proc_line/construct_labels write (everything after "hdr\tread\t").
// Room: 160 bytes + 30 per candidate (a taxid of up to 10 digits, a float of up to 16 characters, two blanks) + the float cache's fixed-size copy.
inline size_t format_call_room(const lmat_read_result& r) { return 192 + 30 * (size_t)r.n_cand; }
inline char* format_call_raw(char* out, const lmat_params& p, int k, const lmat_read_result& r, const lmat_cand* cands) {
    W w{out};
    switch (r.status) {
        case LMAT_ST_SHORT_LEN:
            w.lit("-1 -1 -1\t-1 -1\t"); w.i64(r.read_len); w.ch(' '); w.i64(k); w.lit(" ReadTooShort\n");
            return w.p;
        case LMAT_ST_SHORT_VALID:
            w.lit("-1 -1 -1\t-1 -1\t"); w.i64(r.valid_kmers); w.ch(' '); w.i64(p.min_kmer); w.lit(" ReadTooShort\n");
            return w.p;
        case LMAT_ST_NODBHITS:
            w.lit("-1 -1 "); w.i64(r.valid_kmers); w.lit("\t-1 -1\t"); w.i64(r.read_len); w.ch(' '); w.i64(k);
            w.lit(" NoDbHits\n");
            return w.p;
        case LMAT_ST_SILENT:
            return w.p;  // quirk Q1: nothing, not even a newline
        case LMAT_ST_PHIX:
            w.lit("-1 -1 "); w.i64(r.cand_kmer_cnt); w.ch('\t');
            w.i64(r.call_tid); w.ch(' '); w.flt(r.call_score); w.ch('\t');
            w.i64(r.call_tid); w.ch(' '); w.flt(r.call_score); w.lit(" DirectMatch\n");
            return w.p;
        default: break;
    }
    w.flt(r.log_avg); w.ch(' '); w.flt(r.stdev); w.ch(' '); w.i64(r.cand_kmer_cnt); w.ch('\t');
    const bool multi = r.match_type == LMAT_MT_MULTI || r.match_type == LMAT_MT_PARTIAL;
    if (p.prn_all || multi) {
        const lmat_cand* c = cands + r.cand_off;
        for (uint32_t i = 0; i < r.n_cand; ++i) { w.ch(' '); w.u32(c[i].tid); w.ch(' '); w.flt(c[i].score); }
        if (r.n_cand == 0) w.lit("-1 -1");
        w.ch('\t');
    }
    if (r.match_type == LMAT_MT_DIRECT || multi) {
        w.i64(r.call_tid); w.ch(' '); w.flt(r.call_score); w.ch(' ');
        const char* mn = match_name(r.match_type);
        w.lit(mn, __builtin_strlen(mn));
    } else if (r.match_type == LMAT_MT_NOMATCH) {
        w.lit("-1 -1 NoMatch");
    } else {
        w.lit("-1 -1 Unmatched");
    }
    w.ch('\n');
    return w.p;
}
inline void format_call(std::string& s, const lmat_params& p, int k, const lmat_read_result& r, const lmat_cand* cands) {
    const size_t at = s.size(), room = format_call_room(r);
    s.resize(at + room);
    char* b = &s[at];
    s.resize(at + (size_t)(format_call_raw(b, p, k, r, cands) - b));
}

}  // namespace lmat
