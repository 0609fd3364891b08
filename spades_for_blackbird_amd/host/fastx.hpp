// fastx.hpp -- FASTA/FASTQ(.gz) reader of the host side (C++ over zlib), the input end of the two
// CLIs.  Behaviour follows the reference's FastaFastqGzParser over the vendored kseq
// (common/io/reads/fasta_fastq_gz_parser.hpp:64-80,113-136; ext/include/kseq/kseq.h):
//   * '>' or '@' starts a record; the name ends at the first white space;
//   * sequence lines are concatenated until the next '>' / '@' / '+' (multi-line FASTA and FASTQ);
//   * bases are upper-cased at parse time (kseq.h:193-194);
//   * a '+' line starts the quality, read until it is as long as the sequence; a quality string of
//     another length makes the parser stop: the stream just ends there (kseq.h:209,212 returns -2,
//     fasta_fastq_gz_parser.hpp:130-136 treats any negative value as end of file).
// Names and qualities are dropped (the path only needs bases, read_converter.cpp:76).
#pragma once

#include <zlib.h>

#include <cstdint>
#include <string>
#include <vector>

namespace bbkhost {

struct ReadBatch {
    std::string bases;              // concatenated read sequences
    std::vector<uint64_t> offsets;  // n + 1
    ReadBatch() : offsets(1, 0) {}
    uint64_t size() const { return offsets.size() - 1; }
    void clear() {
        bases.clear();
        offsets.assign(1, 0);
    }
};

class FastxReader {
  public:
    explicit FastxReader(const std::string &path) : path_(path) { fp_ = gzopen(path.c_str(), "r"); if (fp_) gzbuffer(fp_, 1 << 20); }
    ~FastxReader() { if (fp_) gzclose(fp_); }
    FastxReader(const FastxReader &) = delete;
    FastxReader &operator=(const FastxReader &) = delete;
    bool is_open() const { return fp_ != nullptr; }

    // Appends up to max_reads records (and at most max_bases bases) to batch; returns the number read.
    uint64_t read(ReadBatch &batch, uint64_t max_reads, uint64_t max_bases) {
        uint64_t got = 0;
        while (!eof_ && got < max_reads && batch.bases.size() < max_bases) {
            if (!next(batch.bases)) break;
            batch.offsets.push_back(batch.bases.size());
            ++got;
        }
        return got;
    }

    // One record with its name and quality (what spades-read-filter passes through): name = the header line without
    // its first character (kseq reads the name up to the end of the line here, KS_SEP_LINE, kseq.h:182), seq = the
    // upper-cased sequence, qual = the quality string ("" for FASTA).  false at the end of the stream.
    bool next_record(std::string &name, std::string &seq, std::string &qual) {
        if (eof_) return false;
        keep_meta_ = true;
        seq.clear();
        name_.clear();
        qual_.clear();
        const bool ok = next(seq);
        keep_meta_ = false;
        if (!ok) return false;
        name.swap(name_);
        qual.swap(qual_);
        return true;
    }

  private:
    int getc_() {
        if (pos_ >= len_) {
            if (eof_in_) return -1;
            len_ = gzread(fp_, buf_, sizeof(buf_));
            pos_ = 0;
            if (len_ <= 0) {
                eof_in_ = true;
                len_ = 0;
                return -1;
            }
        }
        return (unsigned char)buf_[pos_++];
    }
    // reads the rest of the current line into dst (without the newline); false at end of input
    bool getline_(std::string *dst) {
        int c;
        bool any = false;
        hit_nl_ = false;
        while ((c = getc_()) != -1) {
            any = true;
            if (c == '\n') {
                hit_nl_ = true;
                return true;
            }
            if (dst) dst->push_back((char)c);
        }
        return any;
    }
    // kseq.h ks_getuntil2: after a line has been appended, ONE trailing '\r' is dropped when the string is longer than 1
    static void strip_cr_(std::string &s, size_t start) {
        if (s.size() - start > 1 && s.back() == '\r') s.pop_back();
    }
    bool next(std::string &out) {
        int c;
        if (last_char_ == 0) {  // jump to the next header line
            while ((c = getc_()) != -1 && c != '>' && c != '@') {}
            if (c == -1) { eof_ = true; return false; }
            last_char_ = c;
        }
        if (!getline_(keep_meta_ ? &name_ : nullptr)) {  // name + comment; a header character that is the last byte:
            eof_ = true;                                 // end of file (kseq.h:182)
            return false;
        }
        if (keep_meta_) strip_cr_(name_, 0);
        const size_t start = out.size();
        // sequence lines
        while ((c = getc_()) != -1 && c != '>' && c != '+' && c != '@') {
            if (c == '\n') continue;
            out.push_back((char)c);
            getline_(&out);
            strip_cr_(out, start);
        }
        // every character of a sequence line is kept and upper-cased (kseq.h:190-194)
        for (size_t r = start; r < out.size(); ++r) {
            const unsigned char ch = (unsigned char)out[r];
            if (ch >= 'a' && ch <= 'z') out[r] = (char)(ch - 32);
        }
        if (c == '>' || c == '@') last_char_ = c;
        else last_char_ = 0;
        if (c != '+') {
            if (c == -1) eof_ = eof_in_;
            return true;  // FASTA record
        }
        const size_t seq_len = out.size() - start;
        getline_(nullptr);  // rest of the '+' line
        if (!hit_nl_) {  // the '+' line is the last thing in the file: "no quality string" (kseq.h:205), the stream ends
            out.resize(start);
            eof_ = true;
            return false;
        }
        // at least one quality line is read, then more until it is as long as the sequence (kseq.h:208)
        std::string q;
        do {
            if (!getline_(&q)) break;
            strip_cr_(q, 0);
        } while (q.size() < seq_len);
        const size_t qlen = q.size();
        if (keep_meta_) qual_ = q;
        last_char_ = 0;
        if (qlen != seq_len) {  // truncated quality string: the reference stops here
            out.resize(start);
            eof_ = true;
            return false;
        }
        return true;
    }

    std::string path_;
    gzFile fp_ = nullptr;
    char buf_[1 << 16];
    int pos_ = 0, len_ = 0;
    bool eof_in_ = false, eof_ = false, hit_nl_ = false, keep_meta_ = false;
    std::string name_, qual_;
    int last_char_ = 0;
};

}  // namespace bbkhost
