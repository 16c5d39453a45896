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

// std::to_chars(general, 6) is specified as printf's "%.6g" in the C locale, i.e. operator<<(float) with the default
// precision; it is several times faster than snprintf, which dominated the CLI's run time (tests/test_host_logic.py
// checks it against "%g" on a sweep of bit patterns).
inline void put_float(std::string& s, float f) {
    char b[48];
    const auto r = std::to_chars(b, b + sizeof b, f, std::chars_format::general, 6);
    s.append(b, r.ptr);
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
inline void format_call(std::string& s, const lmat_params& p, int k, const lmat_read_result& r, const lmat_cand* cands) {
    switch (r.status) {
        case LMAT_ST_SHORT_LEN:
            s += "-1 -1 -1\t-1 -1\t"; put_int(s, r.read_len); s += ' '; put_int(s, k); s += " ReadTooShort\n";
            return;
        case LMAT_ST_SHORT_VALID:
            s += "-1 -1 -1\t-1 -1\t"; put_int(s, r.valid_kmers); s += ' '; put_int(s, p.min_kmer); s += " ReadTooShort\n";
            return;
        case LMAT_ST_NODBHITS:
            s += "-1 -1 "; put_int(s, r.valid_kmers); s += "\t-1 -1\t"; put_int(s, r.read_len); s += ' '; put_int(s, k);
            s += " NoDbHits\n";
            return;
        case LMAT_ST_SILENT:
            return;  // quirk Q1: nothing, not even a newline
        case LMAT_ST_PHIX:
            s += "-1 -1 "; put_int(s, r.cand_kmer_cnt); s += '\t';
            put_int(s, r.call_tid); s += ' '; put_float(s, r.call_score); s += '\t';
            put_int(s, r.call_tid); s += ' '; put_float(s, r.call_score); s += " DirectMatch\n";
            return;
        default: break;
    }
    put_float(s, r.log_avg); s += ' '; put_float(s, r.stdev); s += ' '; put_int(s, r.cand_kmer_cnt); s += '\t';
    const bool multi = r.match_type == LMAT_MT_MULTI || r.match_type == LMAT_MT_PARTIAL;
    if (p.prn_all || multi) {
        for (uint32_t i = 0; i < r.n_cand; ++i) {
            s += ' '; put_int(s, cands[r.cand_off + i].tid); s += ' '; put_float(s, cands[r.cand_off + i].score);
        }
        if (r.n_cand == 0) s += "-1 -1";
        s += '\t';
    }
    if (r.match_type == LMAT_MT_DIRECT || multi) {
        put_int(s, r.call_tid); s += ' '; put_float(s, r.call_score); s += ' '; s += match_name(r.match_type);
    } else if (r.match_type == LMAT_MT_NOMATCH) {
        s += "-1 -1 NoMatch";
    } else {
        s += "-1 -1 Unmatched";
    }
    s += '\n';
}

}  // namespace lmat
