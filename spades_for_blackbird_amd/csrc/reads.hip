// reads.hip -- the input side of the path: ASCII reads -> LongestValid run -> 2-bit packed words
// in HBM (replaces io::EasyStream + LongestValidWrap, reference common/io/reads/io_helper.cpp:19-32,
// longest_valid_wrapper.hpp:15-52; reverse complements are never materialised, kernels
// canonicalise instead of RCWrap, rc_reader_wrapper.hpp:33-42), plus the on-device synthetic
// read generator of SURVEY.md 8(d) used by bench.py.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "bbk_internal.h"
#include "kmer_ops.h"

namespace bbk {

static inline bool is_nucl(char c) {  // reference common/sequence/nucl.hpp:45-62
    switch (c) {
        case 'A': case 'C': case 'G': case 'T':
        case 'a': case 'c': case 'g': case 't':
            return true;
        default:
            return false;
    }
}

static inline uint64_t dignucl(char c) {  // nucl.hpp:120-130
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        default: return 3;
    }
}

// longest maximal run of valid bases, first wins on ties (longest_valid_wrapper.hpp:15-41)
static inline void longest_valid(const char *s, uint64_t len, uint64_t *from, uint64_t *to) {
    uint64_t best_len = 0, best_pos = 0, pos = 0;
    bool in = false;
    for (uint64_t i = 0; i <= len; ++i) {
        if (i < len && is_nucl(s[i])) {
            if (!in) {
                in = true;
                pos = i;
            }
        } else if (in) {
            if (i - pos > best_len) {
                best_len = i - pos;
                best_pos = pos;
            }
            in = false;
        }
    }
    *from = best_pos;
    *to = best_pos + best_len;
}

__device__ __host__ inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// genome base g: 2 bits of splitmix64(seed_genome + g/32)
__device__ inline uint32_t genome_base(uint64_t seed, uint64_t g) {
    return (uint32_t)((splitmix64(seed + (g >> 5)) >> ((g & 31) << 1)) & 3ull);
}

// one thread per (read, 32-base word)
__global__ void k_synth_reads(uint64_t n_reads, uint32_t read_len, uint32_t words_per_read, uint64_t genome_len,
                              uint32_t sub_thresh, uint64_t seed_genome, uint64_t seed_reads,
                              uint64_t *__restrict__ words, uint64_t *__restrict__ woff, uint32_t *__restrict__ len) {
    const uint64_t gid = BBK_GID();
    const uint64_t r = gid / words_per_read;
    const uint32_t wj = (uint32_t)(gid % words_per_read);
    if (r >= n_reads) return;
    const uint64_t h = splitmix64(seed_reads ^ (r * 0xD1B54A32D192ED03ull));
    const uint64_t start = (uint64_t)__umul64hi(splitmix64(h), genome_len - read_len + 1);
    const bool flip = (h >> 63) != 0;
    uint64_t w = 0;
    for (uint32_t j = 0; j < 32; ++j) {
        const uint32_t p = wj * 32 + j;  // position in the emitted read
        if (p >= read_len) break;
        const uint32_t q = flip ? (read_len - 1 - p) : p;  // position in the forward template
        uint32_t b = genome_base(seed_genome, start + q);
        const uint64_t e = splitmix64(h + 0x632BE59BD9B4E019ull * (q + 1));
        if ((uint32_t)(e >> 32) < sub_thresh) b = (b + 1 + (uint32_t)(e % 3)) & 3u;
        if (flip) b = 3 - b;
        w |= (uint64_t)b << (j << 1);
    }
    words[r * words_per_read + wj] = w;
    if (wj == 0) {
        woff[r] = r * words_per_read;
        len[r] = read_len;
        if (r == n_reads - 1) woff[n_reads] = n_reads * words_per_read;
    }
}

// word offset of every read when the reads are stored one after the other, each starting on a 64-bit word
// Metagenome-shaped reads (BASELINE configs[4]): read r picks genome g with probability ~ abundance x
// length (cdf: n_genomes + 1 thresholds scaled to 2^64), a uniform start inside it, strand and substitutions as
// k_synth_reads.  Genome g is the random sequence seeded with seed_genome ^ mix(g).
__global__ void k_synth_meta(uint64_t n_reads, uint32_t read_len, uint32_t words_per_read, uint32_t n_genomes,
                             const uint64_t *__restrict__ cdf, const uint64_t *__restrict__ glen, uint32_t sub_thresh,
                             uint64_t seed_genome, uint64_t seed_reads, uint64_t *__restrict__ words,
                             uint64_t *__restrict__ woff, uint32_t *__restrict__ len) {
    const uint64_t gid = BBK_GID();
    const uint64_t r = gid / words_per_read;
    const uint32_t wj = (uint32_t)(gid % words_per_read);
    if (r >= n_reads) return;
    const uint64_t h = splitmix64(seed_reads ^ (r * 0xD1B54A32D192ED03ull));
    const uint64_t u = splitmix64(h ^ 0x2545F4914F6CDD1Dull);
    uint32_t lo = 0, hi = n_genomes;  // largest g with cdf[g] <= u
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (cdf[mid] <= u) lo = mid;
        else hi = mid;
    }
    const uint32_t g = lo;
    const uint64_t gseed = seed_genome ^ splitmix64(0x9E3779B97F4A7C15ull * (g + 1));
    const uint64_t start = (uint64_t)__umul64hi(splitmix64(h), glen[g] - read_len + 1);
    const bool flip = (h >> 63) != 0;
    uint64_t w = 0;
    for (uint32_t j = 0; j < 32; ++j) {
        const uint32_t p = wj * 32 + j;
        if (p >= read_len) break;
        const uint32_t q = flip ? (read_len - 1 - p) : p;
        uint32_t b = genome_base(gseed, start + q);
        const uint64_t e = splitmix64(h + 0x632BE59BD9B4E019ull * (q + 1));
        if ((uint32_t)(e >> 32) < sub_thresh) b = (b + 1 + (uint32_t)(e % 3)) & 3u;
        if (flip) b = 3 - b;
        w |= (uint64_t)b << (j << 1);
    }
    words[r * words_per_read + wj] = w;
    if (wj == 0) {
        woff[r] = r * words_per_read;
        len[r] = read_len;
        if (r == n_reads - 1) woff[n_reads] = n_reads * words_per_read;
    }
}

__global__ void k_len_to_words(const uint32_t *__restrict__ len, uint64_t n, uint64_t *__restrict__ out) {
    const uint64_t i = BBK_GID();
    if (i < n) out[i] = ((uint64_t)len[i] + 31u) >> 5;
}

__global__ void k_len_to_u64(const uint32_t *__restrict__ len, uint64_t n, uint64_t *__restrict__ out) {
    const uint64_t i = BBK_GID();
    if (i < n) out[i] = len[i];
}

__global__ void k_unpack_ascii(const uint64_t *__restrict__ words, const uint64_t *__restrict__ woff,
                               const uint32_t *__restrict__ len, const uint64_t *__restrict__ boff, uint64_t n_reads,
                               char *__restrict__ out) {
    const uint64_t r = BBK_GID() >> 6;  // one wavefront per read
    if (r >= n_reads) return;
    const int lane = threadIdx.x & 63;
    const uint64_t *rw = words + woff[r];
    const uint32_t L = len[r];
    const uint64_t o = boff[r];
    for (uint32_t p = lane; p < L; p += 64) out[o + p] = "ACGT"[base_at(rw, p)];
}

}  // namespace bbk

extern "C" {

int bbk_reads_from_ascii(bbk_ctx *ctx, const char *h_bases, const uint64_t *h_offsets, uint64_t n_reads,
                         bbk_reads **out) {
    return bbk::guarded([&] {
        BBK_REQUIRE(ctx && out && (n_reads == 0 || (h_bases && h_offsets)), BBK_ERR_ARG,
                    "bbk_reads_from_ascii: NULL argument");
        BBK_HIP(hipSetDevice(ctx->device));
        std::vector<uint64_t> woff(n_reads + 1);
        std::vector<uint32_t> len(n_reads);
        std::vector<uint64_t> from(n_reads);
        uint64_t nw = 0, bases = 0;
        for (uint64_t r = 0; r < n_reads; ++r) {
            uint64_t f, t;
            bbk::longest_valid(h_bases + h_offsets[r], h_offsets[r + 1] - h_offsets[r], &f, &t);
            BBK_REQUIRE(t - f < (1ull << 32), BBK_ERR_ARG, "read %llu longer than 2^32 bases", (unsigned long long)r);
            from[r] = h_offsets[r] + f;
            len[r] = (uint32_t)(t - f);
            woff[r] = nw;
            nw += (len[r] + 31) / 32;
            bases += len[r];
        }
        woff[n_reads] = nw;
        std::vector<uint64_t> words(nw + 1, 0);
#pragma omp parallel for schedule(static) num_threads(32)
        for (uint64_t r = 0; r < n_reads; ++r) {
            const char *s = h_bases + from[r];
            uint64_t *w = words.data() + woff[r];
            for (uint32_t i = 0; i < len[r]; ++i) w[i >> 5] |= bbk::dignucl(s[i]) << ((i & 31) << 1);
        }
        auto rd = new bbk_reads();
        std::unique_ptr<bbk_reads> guard(rd);
        rd->ctx = ctx;
        rd->n = n_reads;
        rd->n_words = nw;
        rd->bases = bases;
        rd->own_words.alloc((nw + 1) * sizeof(uint64_t));
        rd->own_woff.alloc((n_reads + 1) * sizeof(uint64_t));
        rd->own_len.alloc((n_reads + 1) * sizeof(uint32_t));
        BBK_HIP(hipMemcpyAsync(rd->own_words.p, words.data(), (nw + 1) * sizeof(uint64_t), hipMemcpyHostToDevice,
                               ctx->stream));
        BBK_HIP(hipMemcpyAsync(rd->own_woff.p, woff.data(), (n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice,
                               ctx->stream));
        if (n_reads)
            BBK_HIP(hipMemcpyAsync(rd->own_len.p, len.data(), n_reads * sizeof(uint32_t), hipMemcpyHostToDevice,
                                   ctx->stream));
        BBK_HIP(hipStreamSynchronize(ctx->stream));
        rd->d_words = rd->own_words.as<uint64_t>();
        rd->d_woff = rd->own_woff.as<uint64_t>();
        rd->d_len = rd->own_len.as<uint32_t>();
        *out = guard.release();
    });
}

// ---- SPAdes binary read cache (.seq / .off), single reads ---------------------------------------
// Format written by io::BinaryWriter::ToBinary (common/io/reads/binary_converter.cpp:50-113):
//   .seq : ReadStreamStat {u64 read_count, u64 max_len, u64 total_len} (io/reads/read_stream.hpp:19-36), then
//          per read: u64 size, ceil(size/32) u64 words of 2-bit bases (Sequence::BinWrite,
//          common/sequence/sequence.hpp:410-442 -- the same LSB-first layout as the device arrays, so a
//          record is copied, not re-encoded), u16 left_offset, u16 right_offset (SingleReadSeq::BinWrite,
//          io/reads/single_read.hpp:286-299);
//   .off : the byte offset in .seq of every 100th read, starting with read 0 (CHUNK = 100,
//          binary_converter.hpp:35, binary_converter.cpp:70-78).
int bbk_reads_from_spades_binary(bbk_ctx *ctx, const char *seq_path, bbk_reads **out) {
    return bbk::guarded([&] {
        BBK_REQUIRE(ctx && seq_path && out, BBK_ERR_ARG, "bbk_reads_from_spades_binary: NULL argument");
        BBK_HIP(hipSetDevice(ctx->device));
        FILE *f = fopen(seq_path, "rb");
        BBK_REQUIRE(f != nullptr, BBK_ERR_IO, "cannot open %s", seq_path);
        std::unique_ptr<FILE, int (*)(FILE *)> fg(f, fclose);
        uint64_t hdr[3];
        BBK_REQUIRE(fread(hdr, 8, 3, f) == 3, BBK_ERR_IO, "%s: truncated header", seq_path);
        const uint64_t n_reads = hdr[0];
        std::vector<uint64_t> woff(n_reads + 1), words;
        std::vector<uint32_t> len(n_reads);
        words.reserve(hdr[2] / 32 + n_reads + 1);
        uint64_t bases = 0;
        for (uint64_t r = 0; r < n_reads; ++r) {
            uint64_t size;
            BBK_REQUIRE(fread(&size, 8, 1, f) == 1, BBK_ERR_IO, "%s: truncated at read %llu", seq_path,
                        (unsigned long long)r);
            BBK_REQUIRE(size < (1ull << 32), BBK_ERR_IO, "%s: read %llu has an implausible size", seq_path,
                        (unsigned long long)r);
            const uint64_t nw = (size + 31) / 32;
            woff[r] = words.size();
            words.resize(words.size() + nw);
            BBK_REQUIRE(nw == 0 || fread(words.data() + woff[r], 8, nw, f) == nw, BBK_ERR_IO,
                        "%s: truncated at read %llu", seq_path, (unsigned long long)r);
            // bits above the last base are not guaranteed to be zero in the cache: clear them
            if (size & 31) words[woff[r] + nw - 1] &= (1ull << ((size & 31) * 2)) - 1;
            uint16_t offs[2];
            BBK_REQUIRE(fread(offs, 2, 2, f) == 2, BBK_ERR_IO, "%s: truncated at read %llu", seq_path,
                        (unsigned long long)r);
            len[r] = (uint32_t)size;
            bases += size;
        }
        woff[n_reads] = words.size();
        words.push_back(0);
        auto rd = new bbk_reads();
        std::unique_ptr<bbk_reads> guard(rd);
        rd->ctx = ctx;
        rd->n = n_reads;
        rd->n_words = words.size() - 1;
        rd->bases = bases;
        rd->own_words.alloc(words.size() * sizeof(uint64_t));
        rd->own_woff.alloc((n_reads + 1) * sizeof(uint64_t));
        rd->own_len.alloc((n_reads + 1) * sizeof(uint32_t));
        BBK_HIP(hipMemcpyAsync(rd->own_words.p, words.data(), words.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        BBK_HIP(hipMemcpyAsync(rd->own_woff.p, woff.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        if (n_reads)
            BBK_HIP(hipMemcpyAsync(rd->own_len.p, len.data(), n_reads * 4, hipMemcpyHostToDevice, ctx->stream));
        BBK_HIP(hipStreamSynchronize(ctx->stream));
        rd->d_words = rd->own_words.as<uint64_t>();
        rd->d_woff = rd->own_woff.as<uint64_t>();
        rd->d_len = rd->own_len.as<uint32_t>();
        *out = guard.release();
    });
}

int bbk_reads_write_spades_binary(bbk_ctx *ctx, const bbk_reads *r, const char *prefix) {
    return bbk::guarded([&] {
        BBK_REQUIRE(ctx && r && prefix, BBK_ERR_ARG, "bbk_reads_write_spades_binary: NULL argument");
        BBK_HIP(hipSetDevice(ctx->device));
        std::vector<uint64_t> woff(r->n + 1), words(r->n_words + 1);
        std::vector<uint32_t> len(r->n + 1);
        BBK_HIP(hipMemcpyAsync(woff.data(), r->d_woff, (r->n + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (r->n) BBK_HIP(hipMemcpyAsync(len.data(), r->d_len, r->n * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (r->n_words)
            BBK_HIP(hipMemcpyAsync(words.data(), r->d_words, r->n_words * 8, hipMemcpyDeviceToHost, ctx->stream));
        BBK_HIP(hipStreamSynchronize(ctx->stream));
        const std::string sp = std::string(prefix) + ".seq", op = std::string(prefix) + ".off";
        FILE *fs = fopen(sp.c_str(), "wb");
        BBK_REQUIRE(fs != nullptr, BBK_ERR_IO, "cannot open %s for writing", sp.c_str());
        std::unique_ptr<FILE, int (*)(FILE *)> gs(fs, fclose);
        FILE *fo = fopen(op.c_str(), "wb");
        BBK_REQUIRE(fo != nullptr, BBK_ERR_IO, "cannot open %s for writing", op.c_str());
        std::unique_ptr<FILE, int (*)(FILE *)> go(fo, fclose);
        uint64_t hdr[3] = {r->n, 0, 0};
        for (uint64_t i = 0; i < r->n; ++i) {
            hdr[1] = std::max<uint64_t>(hdr[1], len[i]);
            hdr[2] += len[i];
        }
        bool ok = fwrite(hdr, 8, 3, fs) == 3;
        uint64_t pos = 24;
        const uint16_t zero[2] = {0, 0};
        for (uint64_t i = 0; i < r->n && ok; ++i) {
            if (i % 100 == 0) ok = fwrite(&pos, 8, 1, fo) == 1;
            const uint64_t size = len[i], nw = (size + 31) / 32;
            ok = ok && fwrite(&size, 8, 1, fs) == 1 && (nw == 0 || fwrite(words.data() + woff[i], 8, nw, fs) == nw) &&
                 fwrite(zero, 2, 2, fs) == 2;
            pos += 8 + nw * 8 + 4;
        }
        BBK_REQUIRE(ok, BBK_ERR_IO, "short write to %s", sp.c_str());
    });
}

int bbk_reads_from_packed(bbk_ctx *ctx, const uint64_t *h_words, uint64_t n_words, const uint32_t *h_len,
                          uint64_t n_reads, bbk_reads **out) {
    return bbk::guarded([&] {
        BBK_REQUIRE(ctx && out && (n_reads == 0 || h_len) && (n_words == 0 || h_words), BBK_ERR_ARG,
                    "bbk_reads_from_packed: NULL argument");
        BBK_HIP(hipSetDevice(ctx->device));
        auto rd = new bbk_reads();
        std::unique_ptr<bbk_reads> guard(rd);
        rd->ctx = ctx;
        rd->n = n_reads;
        rd->n_words = n_words;
        rd->own_words.alloc((n_words + 1) * sizeof(uint64_t));
        rd->own_woff.alloc((n_reads + 1) * sizeof(uint64_t));
        rd->own_len.alloc((n_reads + 1) * sizeof(uint32_t));
        if (n_words)
            BBK_HIP(hipMemcpyAsync(rd->own_words.p, h_words, n_words * sizeof(uint64_t), hipMemcpyHostToDevice,
                                   ctx->stream));
        if (n_reads)
            BBK_HIP(hipMemcpyAsync(rd->own_len.p, h_len, n_reads * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        uint64_t total_words = 0, bases = 0;
        if (n_reads) {
            // word offsets and the base count come from the lengths, on the device (two scans)
            hipLaunchKernelGGL(bbk::k_len_to_words, bbk::grid_blocks((n_reads + 255) / 256), dim3(256), 0, ctx->stream,
                               rd->own_len.as<uint32_t>(), n_reads, rd->own_woff.as<uint64_t>());
            bbk::check_launch("k_len_to_words");
            total_words = bbk::exclusive_scan_u64(ctx, rd->own_woff.as<uint64_t>(), rd->own_woff.as<uint64_t>(), n_reads);
            bbk::DevBuf bl((n_reads + 1) * sizeof(uint64_t));
            hipLaunchKernelGGL(bbk::k_len_to_u64, bbk::grid_blocks((n_reads + 255) / 256), dim3(256), 0, ctx->stream,
                               rd->own_len.as<uint32_t>(), n_reads, bl.as<uint64_t>());
            bbk::check_launch("k_len_to_u64");
            bases = bbk::exclusive_scan_u64(ctx, bl.as<uint64_t>(), bl.as<uint64_t>(), n_reads);
        }
        BBK_REQUIRE(total_words == n_words, BBK_ERR_ARG,
                    "bbk_reads_from_packed: the lengths need %llu words, %llu were passed", (unsigned long long)total_words,
                    (unsigned long long)n_words);
        BBK_HIP(hipMemcpyAsync(rd->own_woff.as<uint64_t>() + n_reads, &total_words, sizeof(uint64_t),
                               hipMemcpyHostToDevice, ctx->stream));
        BBK_HIP(hipStreamSynchronize(ctx->stream));
        rd->bases = bases;
        rd->d_words = rd->own_words.as<uint64_t>();
        rd->d_woff = rd->own_woff.as<uint64_t>();
        rd->d_len = rd->own_len.as<uint32_t>();
        *out = guard.release();
    });
}

int bbk_host_alloc(size_t bytes, void **out) {
    return bbk::guarded([&] {
        BBK_REQUIRE(out, BBK_ERR_ARG, "bbk_host_alloc: NULL argument");
        void *p = nullptr;
        BBK_HIP(hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault));
        *out = p;
    });
}

void bbk_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

int bbk_reads_from_device(bbk_ctx *ctx, const void *d_words, const void *d_word_off, const void *d_len,
                          uint64_t n_reads, uint64_t n_words, bbk_reads **out) {
    return bbk::guarded([&] {
        BBK_REQUIRE(ctx && out && d_words && d_word_off && d_len, BBK_ERR_ARG, "bbk_reads_from_device: NULL argument");
        auto rd = new bbk_reads();
        rd->ctx = ctx;
        rd->n = n_reads;
        rd->n_words = n_words;
        rd->d_words = (const uint64_t *)d_words;
        rd->d_woff = (const uint64_t *)d_word_off;
        rd->d_len = (const uint32_t *)d_len;
        rd->bases = 0;  // unknown without a reduction; only informational
        *out = rd;
    });
}

int bbk_reads_synth(bbk_ctx *ctx, uint64_t n_reads, uint32_t read_len, uint64_t genome_len, double sub_rate,
                    uint64_t seed_genome, uint64_t seed_reads, bbk_reads **out) {
    return bbk::guarded([&] {
        BBK_REQUIRE(ctx && out, BBK_ERR_ARG, "bbk_reads_synth: NULL argument");
        BBK_REQUIRE(read_len >= 1 && genome_len >= read_len && sub_rate >= 0 && sub_rate < 1, BBK_ERR_ARG,
                    "bbk_reads_synth: bad shape (read_len=%u genome_len=%llu sub_rate=%g)", read_len,
                    (unsigned long long)genome_len, sub_rate);
        BBK_HIP(hipSetDevice(ctx->device));
        const uint32_t wpr = (read_len + 31) / 32;
        auto rd = new bbk_reads();
        std::unique_ptr<bbk_reads> guard(rd);
        rd->ctx = ctx;
        rd->n = n_reads;
        rd->n_words = n_reads * wpr;
        rd->bases = n_reads * read_len;
        rd->own_words.alloc((rd->n_words + 1) * sizeof(uint64_t));
        rd->own_woff.alloc((n_reads + 1) * sizeof(uint64_t));
        rd->own_len.alloc((n_reads + 1) * sizeof(uint32_t));
        if (n_reads) {
            const uint64_t total = n_reads * wpr;
            const uint32_t thresh = (uint32_t)(sub_rate * 4294967296.0);
            hipLaunchKernelGGL(bbk::k_synth_reads, bbk::grid_blocks((total + 255) / 256), dim3(256), 0, ctx->stream,
                               n_reads, read_len, wpr, genome_len, thresh, seed_genome, seed_reads,
                               rd->own_words.as<uint64_t>(), rd->own_woff.as<uint64_t>(), rd->own_len.as<uint32_t>());
            bbk::check_launch("k_synth_reads");
        } else {
            BBK_HIP(hipMemsetAsync(rd->own_woff.p, 0, sizeof(uint64_t), ctx->stream));
        }
        BBK_HIP(hipStreamSynchronize(ctx->stream));
        rd->d_words = rd->own_words.as<uint64_t>();
        rd->d_woff = rd->own_woff.as<uint64_t>();
        rd->d_len = rd->own_len.as<uint32_t>();
        *out = guard.release();
    });
}

int bbk_reads_synth_meta(bbk_ctx *ctx, uint64_t n_reads, uint32_t read_len, uint32_t n_genomes, uint64_t min_len,
                         uint64_t max_len, double sigma, double sub_rate, uint64_t seed, bbk_reads **out,
                         uint64_t *h_genome_len, double *h_abundance) {
    return bbk::guarded([&] {
        BBK_REQUIRE(ctx && out, BBK_ERR_ARG, "bbk_reads_synth_meta: NULL argument");
        BBK_REQUIRE(read_len >= 1 && n_genomes >= 1 && n_genomes <= 65536 && min_len >= read_len && max_len >= min_len &&
                        sigma >= 0 && sub_rate >= 0 && sub_rate < 1,
                    BBK_ERR_ARG, "bbk_reads_synth_meta: bad shape");
        BBK_HIP(hipSetDevice(ctx->device));
        // genome lengths log-uniform in [min_len, max_len], abundances log-normal(sigma) (Box-Muller on splitmix64)
        std::vector<uint64_t> glen(n_genomes), cdf(n_genomes + 1);
        std::vector<double> ab(n_genomes), wgt(n_genomes);
        auto u01 = [](uint64_t x) { return ((double)(bbk::splitmix64(x) >> 11) + 0.5) / 9007199254740992.0; };
        double tot = 0;
        for (uint32_t g = 0; g < n_genomes; ++g) {
            const double a = u01(seed * 3 + 7919ull * g + 1), b1 = u01(seed * 5 + 104729ull * g + 2),
                         b2 = u01(seed * 7 + 1299709ull * g + 3);
            glen[g] = (uint64_t)((double)min_len * pow((double)max_len / (double)min_len, a));
            if (glen[g] < read_len) glen[g] = read_len;
            const double z = sqrt(-2.0 * log(b1)) * cos(6.283185307179586 * b2);
            ab[g] = exp(sigma * z);
            wgt[g] = ab[g] * (double)glen[g];
            tot += wgt[g];
        }
        double acc = 0;
        for (uint32_t g = 0; g < n_genomes; ++g) {
            cdf[g] = (uint64_t)(acc / tot * 18446744073709551615.0);
            acc += wgt[g];
        }
        cdf[0] = 0;
        cdf[n_genomes] = ~0ull;
        if (h_genome_len) memcpy(h_genome_len, glen.data(), n_genomes * sizeof(uint64_t));
        if (h_abundance) memcpy(h_abundance, ab.data(), n_genomes * sizeof(double));
        const uint32_t wpr = (read_len + 31) / 32;
        auto rd = new bbk_reads();
        std::unique_ptr<bbk_reads> guard(rd);
        rd->ctx = ctx;
        rd->n = n_reads;
        rd->n_words = n_reads * wpr;
        rd->bases = n_reads * read_len;
        rd->own_words.alloc((rd->n_words + 1) * sizeof(uint64_t));
        rd->own_woff.alloc((n_reads + 1) * sizeof(uint64_t));
        rd->own_len.alloc((n_reads + 1) * sizeof(uint32_t));
        bbk::DevBuf d_cdf((n_genomes + 1) * 8), d_glen(n_genomes * 8);
        BBK_HIP(hipMemcpyAsync(d_cdf.p, cdf.data(), (n_genomes + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        BBK_HIP(hipMemcpyAsync(d_glen.p, glen.data(), n_genomes * 8, hipMemcpyHostToDevice, ctx->stream));
        if (n_reads) {
            const uint64_t total = n_reads * wpr;
            const uint32_t thresh = (uint32_t)(sub_rate * 4294967296.0);
            hipLaunchKernelGGL(bbk::k_synth_meta, bbk::grid_blocks((total + 255) / 256), dim3(256), 0, ctx->stream, n_reads,
                               read_len, wpr, n_genomes, d_cdf.as<uint64_t>(), d_glen.as<uint64_t>(), thresh, seed,
                               seed ^ 0xA5A5A5A5A5A5A5A5ull, rd->own_words.as<uint64_t>(), rd->own_woff.as<uint64_t>(),
                               rd->own_len.as<uint32_t>());
            bbk::check_launch("k_synth_meta");
        } else {
            BBK_HIP(hipMemsetAsync(rd->own_woff.p, 0, sizeof(uint64_t), ctx->stream));
        }
        BBK_HIP(hipStreamSynchronize(ctx->stream));
        rd->d_words = rd->own_words.as<uint64_t>();
        rd->d_woff = rd->own_woff.as<uint64_t>();
        rd->d_len = rd->own_len.as<uint32_t>();
        *out = guard.release();
    });
}

uint64_t bbk_reads_count(const bbk_reads *r) { return r ? r->n : 0; }
uint64_t bbk_reads_bases(const bbk_reads *r) { return r ? r->bases : 0; }

int bbk_reads_get_ascii(bbk_ctx *ctx, const bbk_reads *r, uint64_t i, char *h_dst, uint32_t cap, uint32_t *len) {
    return bbk::guarded([&] {
        BBK_REQUIRE(ctx && r && h_dst && len && i < r->n, BBK_ERR_ARG, "bbk_reads_get_ascii: bad argument");
        uint64_t wo[2];
        uint32_t L;
        BBK_HIP(hipMemcpyAsync(wo, r->d_woff + i, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
        BBK_HIP(hipMemcpyAsync(&L, r->d_len + i, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        BBK_HIP(hipStreamSynchronize(ctx->stream));
        BBK_REQUIRE(L + 1 <= cap, BBK_ERR_ARG, "bbk_reads_get_ascii: buffer too small (%u needed)", L + 1);
        std::vector<uint64_t> w(wo[1] - wo[0] + 1);
        if (wo[1] > wo[0])
            BBK_HIP(hipMemcpyAsync(w.data(), r->d_words + wo[0], (wo[1] - wo[0]) * sizeof(uint64_t),
                                   hipMemcpyDeviceToHost, ctx->stream));
        BBK_HIP(hipStreamSynchronize(ctx->stream));
        for (uint32_t p = 0; p < L; ++p) h_dst[p] = "ACGT"[(w[p >> 5] >> ((p & 31) << 1)) & 3];
        h_dst[L] = 0;
        *len = L;
    });
}

int bbk_reads_export_ascii(bbk_ctx *ctx, const bbk_reads *r, char *h_bases, uint64_t *h_offsets, uint64_t cap_bases) {
    return bbk::guarded([&] {
        BBK_REQUIRE(ctx && r && h_offsets, BBK_ERR_ARG, "bbk_reads_export_ascii: NULL argument");
        BBK_HIP(hipSetDevice(ctx->device));
        const uint64_t n = r->n;
        h_offsets[0] = 0;
        if (n == 0) return;
        bbk::DevBuf boff((n + 1) * sizeof(uint64_t));
        hipLaunchKernelGGL(bbk::k_len_to_u64, bbk::grid_blocks((n + 255) / 256), dim3(256), 0, ctx->stream, r->d_len, n,
                           boff.as<uint64_t>());
        bbk::check_launch("k_len_to_u64");
        const uint64_t total = bbk::exclusive_scan_u64(ctx, boff.as<uint64_t>(), boff.as<uint64_t>(), n);
        BBK_HIP(hipMemcpyAsync(h_offsets, boff.p, n * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
        h_offsets[n] = total;
        if (h_bases) {
            BBK_REQUIRE(total <= cap_bases, BBK_ERR_ARG, "bbk_reads_export_ascii: buffer too small (%llu needed)",
                        (unsigned long long)total);
            bbk::DevBuf out(total + 1);
            hipLaunchKernelGGL(bbk::k_unpack_ascii, bbk::grid_blocks((n + 3) / 4), dim3(256), 0, ctx->stream, r->d_words,
                               r->d_woff, r->d_len, boff.as<uint64_t>(), n, out.as<char>());
            bbk::check_launch("k_unpack_ascii");
            BBK_HIP(hipMemcpyAsync(h_bases, out.p, total, hipMemcpyDeviceToHost, ctx->stream));
            BBK_HIP(hipStreamSynchronize(ctx->stream));
        }
        BBK_HIP(hipStreamSynchronize(ctx->stream));
    });
}

void bbk_reads_free(bbk_reads *r) { delete r; }

}  // extern "C"
