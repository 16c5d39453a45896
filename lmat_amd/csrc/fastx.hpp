// fastx.hpp -- FASTA/FASTQ record stream with the exact record/header pairing of the reference's
// producer loop (src/read_label.cpp:1651-1713):
//   * a FASTA record is pushed when the next '>' line (or EOF) arrives, with the header before it;
//     sequence lines of length <= 1 are ignored and multi-line records are concatenated (:1675);
//   * in FASTQ mode (-q) a record is pushed at its '+'/'-' line with the header of the PREVIOUS
//     record, the first one with an empty header (:1668-1692) -- downstream prints
//     "unknown_hdr:<n>" for an empty header (:1728-1732); exactly one quality line is skipped (:1701).
#pragma once
#include <istream>
#include <string>

namespace lmat {

class FastxReader {
public:
    FastxReader(std::istream& in, bool fastq) : in_(in), fastq_(fastq) {}

    // next (read, header); false at end of input
    bool next(std::string& read, std::string& hdr) {
        std::string line;
        while (!finished_) {
            if (!std::getline(in_, line)) {
                finished_ = true;
                line.clear();
            }
            char c0 = line.empty() ? '\0' : line[0];
            if (c0 == '>' || (fastq_ && c0 == '@')) {
                last_hdr_ = hdr_;
                hdr_.assign(line, 1, std::string::npos);
            }
            bool is_seq = false;
            if (!fastq_) is_seq = c0 != '>' && line.length() > 1;
            else is_seq = c0 != '@' && c0 != '+' && c0 != '-';
            if (is_seq) {
                buf_ += line;
                c0 = '\0';  // the reference clears `line` after appending it
            }
            const bool boundary = (c0 == '>' || finished_) || (fastq_ && (c0 == '+' || c0 == '-'));
            if (boundary && !buf_.empty()) {
                read.swap(buf_);
                buf_.clear();
                hdr = finished_ ? hdr_ : last_hdr_;
                if (fastq_) std::getline(in_, line);  // quality line
                return true;
            }
        }
        return false;
    }

private:
    std::istream& in_;
    bool fastq_;
    bool finished_ = false;
    std::string hdr_, last_hdr_, buf_;
};

}  // namespace lmat
