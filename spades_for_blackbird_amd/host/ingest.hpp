// ingest.hpp -- the input end of the two CLIs: FASTA/FASTQ(.gz) text -> 2-bit packed reads, in parallel, in bounded
// blocks that are handed to the device while the next block is being parsed.
//
// Reference: the master thread of hammer::ReadProcessor parses with kseq while the workers consume
// (common/io/reads/read_processor.hpp:76-135, projects/kmercount/main.cpp:95-120); the parser is
// FastaFastqGzParser over the vendored kseq (common/io/reads/fasta_fastq_gz_parser.hpp:64-80,113-136;
// ext/include/kseq/kseq.h:170-212), followed by LongestValidWrap (io/reads/longest_valid_wrapper.hpp:15-52).
// Here the text of one block (an mmap view of a plain file, or a buffer filled by gzread) is cut into one chunk per
// thread at record starts; every thread runs the same record state machine as kseq over its chunk, applies the
// LongestValid rule and packs the run straight into 2-bit words (A=0 C=1 G=2 T=3, base i in bits 2(i%32) of word
// i/32, every read on a word boundary: the device layout).  A chunk boundary is a guess that is VERIFIED: the parser
// of chunk j must arrive exactly at the start of chunk j+1, otherwise the block is re-parsed by one thread (which is
// the serial algorithm and correct by construction).  A record that is not complete at the end of a block is carried
// into the next one, so a block never ends inside a record.
//
// kseq semantics kept (see fastx.hpp for the serial statement of the same rules):
//   * after a FASTQ record the parser skips to the next '>' or '@' at ANY position; after a FASTA record the header
//     character that ended it has already been consumed (here: the next record starts at that character);
//   * the header line is skipped; sequence lines are concatenated until a line starts with '>', '@' or '+'; empty
//     lines are skipped; one trailing '\r' of a line is dropped when the sequence so far is longer than 1;
//   * '+' starts the quality: the rest of that line is skipped, then quality lines are read until they are at least
//     as long as the sequence (at least one line is always read); a quality of another length, or a '+' line that
//     is the last thing in the file, ends the stream (kseq returns -2, the reference treats it as end of file);
//   * bases are upper-cased by kseq; the packing accepts both cases (common/sequence/nucl.hpp:45-62,120-130).
#pragma once

#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include <omp.h>

namespace bbkhost {

// one block of reads, packed
struct PackedReads {
    std::vector<uint64_t> words;
    std::vector<uint32_t> len;
    uint64_t bases = 0;
    void clear() {
        words.clear();
        len.clear();
        bases = 0;
    }
    uint64_t size() const { return len.size(); }
};

namespace detail {

struct NuclTable {
    unsigned char v[256];
    NuclTable() {
        memset(v, 0, sizeof(v));
        for (const char *p = "ACGTacgt"; *p; ++p) v[(unsigned char)*p] = 1;
    }
};
inline const unsigned char *nucl_table() {
    static const NuclTable t;
    return t.v;
}

#if defined(__x86_64__)
// AVX2 forms of the two per-base loops (chosen at run time; the scalar code below is the reference and the fallback):
// a line of a read file is almost always nothing but ACGT, so "is the whole string valid" decides LongestValid in one
// pass of 32 bytes per step, and 32 bases are packed into one 64-bit word with two multiply-adds and a byte shuffle.
__attribute__((target("avx2"))) inline bool all_valid_avx2(const char *s, size_t n) {
    const __m256i up = _mm256_set1_epi8((char)0xDF);
    const __m256i a = _mm256_set1_epi8('A'), c = _mm256_set1_epi8('C'), g = _mm256_set1_epi8('G'), t = _mm256_set1_epi8('T');
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        const __m256i v = _mm256_and_si256(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + i)), up);
        const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(v, a), _mm256_cmpeq_epi8(v, c)),
                                           _mm256_or_si256(_mm256_cmpeq_epi8(v, g), _mm256_cmpeq_epi8(v, t)));
        if ((uint32_t)_mm256_movemask_epi8(ok) != 0xFFFFFFFFu) return false;
    }
    if (i < n) {  // the last (n mod 32) bytes: one more block ending at the end of the string (n >= 32)
        const __m256i v = _mm256_and_si256(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + n - 32)), up);
        const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(v, a), _mm256_cmpeq_epi8(v, c)),
                                           _mm256_or_si256(_mm256_cmpeq_epi8(v, g), _mm256_cmpeq_epi8(v, t)));
        if ((uint32_t)_mm256_movemask_epi8(ok) != 0xFFFFFFFFu) return false;
    }
    return true;
}
// 32 ASCII bases (all ACGTacgt) -> 64 bits, base 0 in the low bits
__attribute__((target("avx2"))) inline uint64_t pack32_avx2(const char *s) {
    const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s));
    // code = ((c >> 1) ^ (c >> 2)) & 3 per byte (the 16-bit shifts only leak into bits that the mask drops)
    const __m256i code = _mm256_and_si256(_mm256_xor_si256(_mm256_srli_epi16(v, 1), _mm256_srli_epi16(v, 2)), _mm256_set1_epi8(3));
    const __m256i p4 = _mm256_maddubs_epi16(code, _mm256_set1_epi16(0x0401));  // words: b0 + 4 b1
    const __m256i p8 = _mm256_madd_epi16(p4, _mm256_set1_epi32(0x00100001));   // dwords: w0 + 16 w1 = 4 bases in 8 bits
    const __m256i sh = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
                                        0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    const __m256i q = _mm256_shuffle_epi8(p8, sh);  // low dword of each 128-bit lane = 16 bases
    return (uint64_t)(uint32_t)_mm256_extract_epi32(q, 0) | ((uint64_t)(uint32_t)_mm256_extract_epi32(q, 4) << 32);
}
inline bool have_avx2() {
    static const bool v = __builtin_cpu_supports("avx2") && getenv("BBK_NO_AVX2") == nullptr;
    return v;
}
#else
inline bool have_avx2() { return false; }
inline bool all_valid_avx2(const char *, size_t) { return false; }
inline uint64_t pack32_avx2(const char *) { return 0; }
#endif

// longest maximal run of ACGTacgt, first wins on ties (longest_valid_wrapper.hpp:15-41)
inline void longest_valid(const char *s, size_t n, size_t *from, size_t *to) {
    if (n >= 32 && have_avx2() && all_valid_avx2(s, n)) {
        *from = 0;
        *to = n;
        return;
    }
    const unsigned char *ok = nucl_table();
    size_t best = 0, best_pos = 0, i = 0;
    while (i < n) {
        while (i < n && !ok[(unsigned char)s[i]]) ++i;
        const size_t p = i;
        while (i < n && ok[(unsigned char)s[i]]) ++i;
        if (i - p > best) {
            best = i - p;
            best_pos = p;
        }
    }
    *from = best_pos;
    *to = best_pos + best;
}

// 8 ASCII bases (ACGTacgt) -> 16 bits, base 0 in the low bits: code = ((c >> 1) ^ (c >> 2)) & 3
inline uint64_t pack8(const char *s) {
    uint64_t x;
    memcpy(&x, s, 8);
    uint64_t y = ((x >> 1) ^ (x >> 2)) & 0x0303030303030303ull;
    y = (y | (y >> 6)) & 0x000F000F000F000Full;
    y = (y | (y >> 12)) & 0x000000FF000000FFull;
    y = (y | (y >> 24)) & 0xFFFFull;
    return y;
}

inline void pack_run(const char *s, size_t n, std::vector<uint64_t> &words) {
    const size_t nw = (n + 31) / 32, o = words.size();
    if (nw == 0) return;
    words.resize(o + nw);  // one growth check per run instead of one per word
    uint64_t *w = words.data() + o;
    size_t i = 0, j = 0;
    if (have_avx2()) {
        for (; i + 32 <= n; i += 32) w[j++] = pack32_avx2(s + i);
        if (i < n && n >= 32) {  // the last (n mod 32) bases: the 32 bases ending at the end of the run, shifted down
            w[j] = pack32_avx2(s + n - 32) >> (2 * (32 - (n - i)));
            return;
        }
    }
    for (; i + 32 <= n; i += 32)
        w[j++] = pack8(s + i) | (pack8(s + i + 8) << 16) | (pack8(s + i + 16) << 32) | (pack8(s + i + 24) << 48);
    if (i < n) {
        uint64_t x = 0;
        int sh = 0;
        for (; i + 8 <= n; i += 8, sh += 16) x |= pack8(s + i) << sh;
        for (; i < n; ++i, sh += 2) {
            const unsigned c = (unsigned char)s[i];
            x |= (uint64_t)(((c >> 1) ^ (c >> 2)) & 3u) << sh;
        }
        w[j] = x;
    }
}

struct ChunkOut {
    PackedReads reads;
    std::string scratch;   // multi-line records are concatenated here
    size_t reached = 0;    // parser state at the end: positioned at `reached` with no pending header character
    bool stopped = false;  // truncated quality: the stream of this file ends here
    bool incomplete = false;  // the record at `reached` is not complete in this block (non-final block)
    void add(const char *s, size_t n) {
        size_t f, t;
        longest_valid(s, n, &f, &t);
        reads.len.push_back((uint32_t)(t - f));
        reads.bases += t - f;
        pack_run(s + f, t - f, reads.words);
    }
};

// Parses the records whose header character lies in [begin, limit) of buf[0, end).  final: buf ends at the end of the
// file.  On return out.reached = the position the serial parser would continue from.
inline void parse_range(const char *buf, size_t begin, size_t limit, size_t end, bool final, ChunkOut &out) {
    size_t pos = begin;
    for (;;) {
        // skip to the next header character
        size_t h = pos;
        while (h < end && buf[h] != '>' && buf[h] != '@') ++h;
        if (h >= end) {  // nothing but skipped bytes are left
            out.reached = end;
            return;
        }
        if (h >= limit) {
            out.reached = h;
            return;
        }
        auto need_more = [&]() {
            out.reached = h;
            out.incomplete = true;
        };
        // header line
        size_t q = h + 1;
        if (q >= end) {  // the header character is the last byte
            if (!final) return need_more();
            out.reached = end;  // kseq: ks_getuntil finds nothing -> end of file, no record
            return;
        }
        {
            const char *nl = (const char *)memchr(buf + q, '\n', end - q);
            if (!nl) {
                if (!final) return need_more();
                out.add(buf, 0);  // a header without a newline at the end of the file: a record with no sequence
                out.reached = end;
                return;
            }
            q = (size_t)(nl - buf) + 1;
        }
        // sequence lines
        const char *seg = nullptr;  // the only line so far (no copy), or null when the scratch buffer is in use
        size_t seg_len = 0, seq_len = 0;
        bool multi = false;
        int c = -1;
        for (;;) {
            if (q >= end) {
                if (!final) return need_more();
                c = -1;
                break;
            }
            c = (unsigned char)buf[q];
            if (c == '>' || c == '@' || c == '+') break;
            if (c == '\n') {
                ++q;
                continue;
            }
            const char *nl = (const char *)memchr(buf + q, '\n', end - q);
            if (!nl && !final) return need_more();
            size_t e = nl ? (size_t)(nl - buf) : end;
            const size_t next = nl ? e + 1 : end;
            size_t n = e - q;
            if (seq_len + n > 1 && n > 0 && buf[e - 1] == '\r') --n;  // kseq.h: trailing '\r' dropped when str->l > 1
            if (seq_len == 0 && !multi) {
                seg = buf + q;
                seg_len = n;
            } else {
                if (!multi) {
                    out.scratch.assign(seg ? seg : "", seg_len);
                    multi = true;
                }
                out.scratch.append(buf + q, n);
            }
            seq_len += n;
            q = next;
        }
        if (c != '+') {  // FASTA record; the next record starts at the header character just seen (or at the end)
            if (multi) out.add(out.scratch.data(), out.scratch.size());
            else out.add(seg, seg_len);
            pos = q;
            continue;
        }
        // '+' line
        {
            const char *nl = (const char *)memchr(buf + q, '\n', end - q);
            if (!nl) {
                if (!final) return need_more();
                out.stopped = true;  // no quality string: the stream ends, the record is not emitted
                out.reached = end;
                return;
            }
            q = (size_t)(nl - buf) + 1;
        }
        // quality lines: until at least as long as the sequence, at least one line
        size_t qual_len = 0;
        do {
            if (q >= end) {
                if (!final) return need_more();
                break;  // nothing left: ks_getuntil2 returns -1
            }
            const char *nl = (const char *)memchr(buf + q, '\n', end - q);
            if (!nl && !final) return need_more();
            const size_t e = nl ? (size_t)(nl - buf) : end;
            size_t n = e - q;
            if (qual_len + n > 1 && n > 0 && buf[e - 1] == '\r') --n;
            qual_len += n;
            q = nl ? e + 1 : end;
        } while (qual_len < seq_len);
        if (qual_len != seq_len) {  // truncated / overlong quality: kseq returns -2 and the reference stops reading
            out.stopped = true;
            out.reached = q;
            return;
        }
        if (multi) out.add(out.scratch.data(), out.scratch.size());
        else out.add(seg, seg_len);
        pos = q;
    }
}

// first position >= from that looks like the start of a record (block-internal chunk boundaries; verified later)
inline size_t guess_record_start(const char *buf, size_t from, size_t end, bool fastq) {
    size_t p = from;
    // move to a line start
    if (p > 0 && buf[p - 1] != '\n') {
        const char *nl = (const char *)memchr(buf + p, '\n', end - p);
        if (!nl) return end;
        p = (size_t)(nl - buf) + 1;
    }
    while (p < end) {
        const char *nl1 = (const char *)memchr(buf + p, '\n', end - p);
        if (!fastq) {
            if (buf[p] == '>') return p;
        } else if (buf[p] == '@' && nl1) {
            // header of a 4-line record: the line after next starts with '+' (a quality line that starts with '@' is
            // followed by a header and a sequence line instead)
            const size_t l2 = (size_t)(nl1 - buf) + 1;
            const char *nl2 = l2 < end ? (const char *)memchr(buf + l2, '\n', end - l2) : nullptr;
            if (nl2) {
                const size_t l3 = (size_t)(nl2 - buf) + 1;
                if (l3 < end && buf[l3] == '+' && buf[l2] != '@' && buf[l2] != '+') return p;
            }
        }
        if (!nl1) return end;
        p = (size_t)(nl1 - buf) + 1;
    }
    return end;
}

}  // namespace detail

// Parses buf[0, n) (a block that starts where the serial parser stands) with `threads` threads into `out`
// (appended).  Returns the bytes consumed; *stopped: the stream of this file ended (truncated quality).
inline size_t parse_block(const char *buf, size_t n, bool final, int threads, PackedReads &out, bool *stopped,
                          bool *used_fallback = nullptr, size_t min_chunk = 1u << 16) {
    using namespace detail;
    *stopped = false;
    if (used_fallback) *used_fallback = false;
    if (n == 0) return 0;
    int T = std::max(1, threads);
    if (n < (size_t)T * min_chunk) T = 1;
    std::vector<ChunkOut> parts((size_t)T);
    std::vector<size_t> start((size_t)T + 1, n);
    start[0] = 0;
    if (T > 1) {
        size_t h = 0;
        while (h < n && buf[h] != '>' && buf[h] != '@') ++h;
        const bool fastq = h < n && buf[h] == '@';
#pragma omp parallel for num_threads(T) schedule(static)
        for (int j = 1; j < T; ++j) start[(size_t)j] = guess_record_start(buf, n / (size_t)T * (size_t)j, n, fastq);
        for (int j = 1; j <= T; ++j) start[(size_t)j] = std::max(start[(size_t)j], start[(size_t)j - 1]);
    }
#pragma omp parallel for num_threads(T) schedule(static)
    for (int j = 0; j < T; ++j) {
        if (start[(size_t)j] >= start[(size_t)j + 1] && j > 0) {
            parts[(size_t)j].reached = start[(size_t)j];
            continue;
        }
        parse_range(buf, start[(size_t)j], start[(size_t)j + 1], n, final, parts[(size_t)j]);
    }
    // verification: chunk j must hand over exactly at the start of chunk j + 1
    bool good = true;
    int last = T - 1;
    for (int j = 0; j < T; ++j) {
        const ChunkOut &p = parts[(size_t)j];
        if (p.stopped || p.incomplete) {
            // legitimate only if it is the end of the walk: a stop ends the stream; an incomplete record can only be
            // the last of the block
            last = j;
            if (p.incomplete)
                for (int i = j + 1; i < T; ++i)
                    if (start[(size_t)i] < start[(size_t)i + 1]) good = false;  // later chunks parsed something else
            break;
        }
        if (j + 1 < T && p.reached != start[(size_t)j + 1]) {
            good = false;
            break;
        }
    }
    if (!good) {  // a boundary guess was wrong (multi-line FASTQ, odd layout): one thread, the serial algorithm
        if (used_fallback) *used_fallback = true;
        ChunkOut one;
        parse_range(buf, 0, n, n, final, one);
        parts.clear();
        parts.push_back(std::move(one));
        last = 0;
    }
    // concatenate in chunk order (every read starts on a word boundary, so the word arrays just follow each other)
    size_t nw = out.words.size(), nr = out.len.size();
    std::vector<size_t> wo((size_t)last + 2), ro((size_t)last + 2);
    wo[0] = nw;
    ro[0] = nr;
    for (int j = 0; j <= last; ++j) {
        wo[(size_t)j + 1] = wo[(size_t)j] + parts[(size_t)j].reads.words.size();
        ro[(size_t)j + 1] = ro[(size_t)j] + parts[(size_t)j].reads.len.size();
        out.bases += parts[(size_t)j].reads.bases;
    }
    out.words.resize(wo[(size_t)last + 1]);
    out.len.resize(ro[(size_t)last + 1]);
#pragma omp parallel for num_threads(T) schedule(static)
    for (int j = 0; j <= last; ++j) {
        const PackedReads &r = parts[(size_t)j].reads;
        if (!r.words.empty()) memcpy(out.words.data() + wo[(size_t)j], r.words.data(), r.words.size() * 8);
        if (!r.len.empty()) memcpy(out.len.data() + ro[(size_t)j], r.len.data(), r.len.size() * 4);
    }
    *stopped = parts[(size_t)last].stopped;
    return parts[(size_t)last].reached;
}

// Streams the records of a list of files as packed blocks of about `block_bytes` of input text.
class Ingest {
  public:
    Ingest(std::vector<std::string> files, size_t block_bytes, int threads)
        : files_(std::move(files)), block_(std::max<size_t>(block_bytes, 1)), threads_(std::max(1, threads)) {}
    ~Ingest() { close_file(); }
    Ingest(const Ingest &) = delete;
    Ingest &operator=(const Ingest &) = delete;

    // Appends the reads of the next block to out (cleared first).  false: all files are done.
    // err receives a message when a file cannot be opened (the caller treats it as fatal).
    bool next(PackedReads &out, std::string &err) {
        out.clear();
        for (;;) {
            if (!open_) {
                if (file_idx_ >= files_.size()) return false;
                if (!open_file(files_[file_idx_], err)) return false;
            }
            bool stopped = false;
            size_t consumed = 0;
            if (map_) {
                const size_t left = size_ - off_;
                size_t take = std::min(left, cur_block_);
                const bool final = take == left;
                bool fb = false;
                consumed = parse_block(map_ + off_, take, final, threads_, out, &stopped, &fb, min_chunk());
                fallbacks_ += fb;
                if (!final && consumed == 0) {  // one record larger than the block
                    cur_block_ *= 2;
                    continue;
                }
                off_ += consumed;
                if (final || stopped) close_file();
            } else {
                // gz: the buffer holds the carry-over of the last block followed by fresh data
                if (buf_.size() < cur_block_) buf_.resize(cur_block_);
                while (!gz_eof_ && fill_ < cur_block_) {
                    const int want = (int)std::min<size_t>(cur_block_ - fill_, 1u << 30);
                    const int got = gzread(gz_, &buf_[fill_], (unsigned)want);
                    if (got <= 0) {
                        // got < 0: corrupt or truncated .gz.  kseq ends the stream there without a word (its reader
                        // returns -1 and kseq_read reports end of file, ext/include/kseq/kseq.h:95-110): the records read
                        // so far are kept, like the reference -- but not silently
                        if (got < 0) {
                            int en = 0;
                            const char *msg = gzerror(gz_, &en);
                            fprintf(stderr, "WARNING: %s: gzip stream error (%s): the input ends here, %llu bytes into the "
                                            "decompressed data\n",
                                    current_file().c_str(), msg ? msg : "?",
                                    (unsigned long long)gz_bytes_);
                        }
                        gz_eof_ = true;
                        break;
                    }
                    gz_bytes_ += (uint64_t)got;
                    fill_ += (size_t)got;
                }
                const bool final = gz_eof_;
                bool fb = false;
                consumed = parse_block(buf_.data(), fill_, final, threads_, out, &stopped, &fb, min_chunk());
                fallbacks_ += fb;
                if (!final && consumed == 0) {
                    cur_block_ *= 2;
                    continue;
                }
                memmove(&buf_[0], &buf_[consumed], fill_ - consumed);
                fill_ -= consumed;
                if (final || stopped) close_file();
            }
            total_reads_ += out.size();
            if (out.size()) return true;
            // nothing in this block (an empty tail, a file without records): go on
        }
    }
    uint64_t total_reads() const { return total_reads_; }
    uint64_t fallback_blocks() const { return fallbacks_; }
    const std::string &current_file() const { return files_[std::min(file_idx_, files_.size() - 1)]; }
    // called with the path whenever a file is opened (the "Processing <file>" log line of the CLIs)
    std::function<void(const std::string &)> on_file;
    // tests lower it to 1: chunks of a few bytes per thread (production: a thread is not worth less than 64 KiB)
    size_t min_block = 1u << 16;

  private:
    size_t min_chunk() const { return min_block; }
    bool open_file(const std::string &path, std::string &err) {
        if (on_file) on_file(path);
        const int fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) {
            err = "Cannot open " + path;
            return false;
        }
        unsigned char magic[2] = {0, 0};
        const ssize_t got = pread(fd, magic, 2, 0);
        struct stat st;
        const bool gz = got == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
        if (!gz && fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) {
            size_ = (size_t)st.st_size;
            if (size_ == 0) {
                close(fd);
                ++file_idx_;
                return true;  // empty file: nothing to do, open_ stays false
            }
            void *m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd, 0);
            close(fd);
            if (m == MAP_FAILED) {
                err = "Cannot map " + path;
                return false;
            }
            (void)madvise(m, size_, MADV_SEQUENTIAL);
            map_ = (const char *)m;
            off_ = 0;
        } else {  // gzip (or a pipe): through zlib, which also passes plain data through
            gz_ = gzdopen(fd, "r");
            if (!gz_) {
                close(fd);
                err = "Cannot open " + path;
                return false;
            }
            gzbuffer(gz_, 1u << 20);
            gz_eof_ = false;
            gz_bytes_ = 0;
            fill_ = 0;
        }
        open_ = true;
        cur_block_ = block_;
        return true;
    }
    void close_file() {
        if (map_) munmap((void *)map_, size_);
        if (gz_) gzclose(gz_);
        map_ = nullptr;
        gz_ = nullptr;
        if (open_) ++file_idx_;
        open_ = false;
    }

    std::vector<std::string> files_;
    size_t block_, cur_block_ = 0;
    int threads_;
    size_t file_idx_ = 0;
    bool open_ = false;
    const char *map_ = nullptr;
    size_t size_ = 0, off_ = 0;
    gzFile gz_ = nullptr;
    bool gz_eof_ = false;
    uint64_t gz_bytes_ = 0;  // decompressed bytes of the current file (for the message on a corrupt stream)
    std::string buf_;
    size_t fill_ = 0;
    uint64_t total_reads_ = 0, fallbacks_ = 0;
};

}  // namespace bbkhost
