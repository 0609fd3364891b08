// count.hip -- k-mer extraction + counting on the device, and the bbk_kmerset part of the C ABI.
//
// Replaces, for one batch of reads resident in HBM:
//   BufferFiller::operator()            reference projects/kmercount/main.cpp:64-82
//   DeBruijnKMerSplitter::FillBufferFromSequence  common/utils/kmer_mph/kmer_splitters.hpp:25-41
//   KMerSortingSplitter::DumpBuffers    common/utils/kmer_mph/kmer_splitter.hpp:120-167 (sort + unique)
//   KMerDiskCounter::Count / MergeKMers common/utils/kmer_mph/kmer_index_builder.hpp:241-267,281-365
//   KMerDiskStorage::merge              kmer_index_builder.hpp:168-181 (the final_kmers order)
// The reference materialises both strands (RCWrap) and filters with IsMinimal; here every read
// position yields ONE canonical key (min(kmer, rc) in base order, rtseq.hpp:407-415) and the
// both-strand set of spades-kmercount is recovered from the distinct canonical keys as
// canon U rc(canon)  (SURVEY.md 7.4).
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <memory>
#include <vector>

#include "bbk_internal.h"
#include "kmer_ops.h"
#include "msd.h"
#include "accum.h"

namespace bbk {

// BBK_DISABLE_MSD=1 forces the LSD path (A/B timing, and tests that cover both paths)
static bool msd_enabled() {
    const char *e = getenv("BBK_DISABLE_MSD");
    return !(e && e[0] == '1');
}

__global__ void k_kmers_per_read(const uint32_t *__restrict__ len, uint64_t n, uint32_t k,
                                 uint64_t *__restrict__ nk) {
    const uint64_t i = BBK_GID();
    if (i < n) {
        const uint32_t L = len[i];
        nk[i] = L >= k ? (uint64_t)(L - k + 1) : 0ull;
    }
}

// One wavefront per read, lanes over k-mer positions: stores of one wave are 64 consecutive
// records.  MODE 0: canonical key.  MODE 1: canonical key + InOutMask bits of this occurrence
// (kmer_extension_index.hpp:67-69,92-106: a non-minimal k-mer stores position 7-pos).
template <int W, int MODE>
__global__ __launch_bounds__(256) void k_extract(const uint64_t *__restrict__ words,
                                                const uint64_t *__restrict__ woff,
                                                const uint32_t *__restrict__ len,
                                                const uint64_t *__restrict__ koff, uint64_t n_reads, int k,
                                                Key<W> *__restrict__ out, uint32_t *__restrict__ out_val) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (BBK_GID()) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave; r < n_reads; r += nwaves) {
        const uint32_t L = len[r];
        if (L < (uint32_t)k) continue;
        const uint32_t nk = L - (uint32_t)k + 1;
        const uint64_t *rw = words + woff[r];
        const uint64_t base = koff[r];
        for (uint32_t p = lane; p < nk; p += 64) {
            const Key<W> fwd = kmer_extract<W>(rw, p, k);
            const Key<W> rc = kmer_rc<W>(fwd, k);
            const bool minimal = !kmer_less_nucl<W>(rc, fwd);  // fwd <= rc (palindrome -> minimal)
            out[base + p] = key_select<W>(minimal, fwd, rc);
            if (MODE == 1) {
                uint32_t m = 0;
                if (p + (uint32_t)k < L) {  // (k+1)-mer starting at p: outgoing base of this k-mer
                    const uint32_t c = base_at(rw, p + (uint32_t)k);
                    m |= 1u << (minimal ? c : 7u - c);
                }
                if (p >= 1) {  // (k+1)-mer starting at p-1: incoming base
                    const uint32_t c = base_at(rw, p - 1);
                    m |= 1u << (minimal ? 4u + c : 3u - c);
                }
                out_val[base + p] = m;
            }
        }
    }
}

// out[2i] = key, out[2i+1] = rc(key): the both-strand set of spades-kmercount
// TAG (8-byte keys with >= 4 spare bits): the XXH3 bucket of 16 (KMerSegmentPolicy, kmer_buckets.hpp:28-33) is
// put right above the k-mer (bits 2k..2k+3), so that one ascending sort of the tagged keys yields the
// final_kmers order
template <int W, bool TAG>
__global__ __launch_bounds__(256) void k_expand_rc(const Key<W> *__restrict__ in, const uint32_t *__restrict__ cin,
                                                  uint64_t n, int k, Key<W> *__restrict__ out,
                                                  uint32_t *__restrict__ cout) {
    const uint64_t i = BBK_GID();
    if (i >= n) return;
    Key<W> x = in[i];
    Key<W> y = kmer_rc<W>(x, k);
    if (TAG) {
        x.w[0] |= __umul64hi(xxh3_64<W>(x), 16ull) << (2 * k);
        y.w[0] |= __umul64hi(xxh3_64<W>(y), 16ull) << (2 * k);
    }
    out[2 * i] = x;
    out[2 * i + 1] = y;
    if (cin) {
        const uint32_t c = cin[i];
        cout[2 * i] = c;
        cout[2 * i + 1] = c;
    }
}

// neighbours of the stored order: out[0] += strict descents (key[i+1] < key[i] in word order), out[1] += equal
// neighbours, out[2..] = the first positions i of a descent (out[2 + 32] of them at most)
template <int W>
__global__ __launch_bounds__(256) void k_order_check(const Key<W> *__restrict__ keys, uint64_t n,
                                                    unsigned long long *__restrict__ out) {
    const uint64_t i = BBK_GID();
    if (i + 1 >= n) return;
    const Key<W> a = key_load<W>(&keys[i]), b = key_load<W>(&keys[i + 1]);
    if (key_less_words<W>(b, a)) {
        const unsigned long long at = atomicAdd(&out[0], 1ull);
        if (at < 32) out[2 + at] = i;
    } else if (key_eq<W>(a, b)) {
        atomicAdd(&out[1], 1ull);
    }
}

// one thread per bucket boundary b = 0..16 of a set in the final_kmers order (XXH3 bucket ids never decrease):
// out[b] = first record whose bucket is >= b
template <int W>
__global__ void k_bucket_bounds(const Key<W> *__restrict__ keys, uint64_t n, unsigned long long *__restrict__ out) {
    const uint32_t b = threadIdx.x;
    if (b > 16) return;
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = lo + ((hi - lo) >> 1);
        const uint32_t bm = (uint32_t)__umul64hi(xxh3_64<W>(key_load<W>(&keys[mid])), 16ull);
        if (bm < b) lo = mid + 1;
        else hi = mid;
    }
    out[b] = lo;
}

template <int W>
static void launch_extract(bbk_ctx *ctx, const bbk_reads *rd, const uint64_t *koff, int k, void *out, uint32_t *vals,
                           uint64_t n_inst) {
    const unsigned blocks = (unsigned)std::min<uint64_t>((rd->n + 3) / 4, (uint64_t)ctx->num_cus * 32);
    KernelTimer t(ctx, "extract", (double)rd->n_words * 8 + (double)n_inst * (sizeof(Key<W>) + (vals ? 4 : 0)));
    if (vals)
        hipLaunchKernelGGL((k_extract<W, 1>), dim3(blocks ? blocks : 1), dim3(256), 0, ctx->stream, rd->d_words,
                           rd->d_woff, rd->d_len, koff, rd->n, k, (Key<W> *)out, vals);
    else
        hipLaunchKernelGGL((k_extract<W, 0>), dim3(blocks ? blocks : 1), dim3(256), 0, ctx->stream, rd->d_words,
                           rd->d_woff, rd->d_len, koff, rd->n, k, (Key<W> *)out, (uint32_t *)nullptr);
    check_launch("k_extract");
}

template <int W>
static void launch_expand(bbk_ctx *ctx, const void *in, const uint32_t *cin, uint64_t n, int k, void *out,
                          uint32_t *cout, bool tag) {
    if (n == 0) return;
    KernelTimer t(ctx, "expand", 3.0 * (double)n * sizeof(Key<W>));
    if (tag)
        hipLaunchKernelGGL((k_expand_rc<W, true>), bbk::grid_blocks((n + 255) / 256), dim3(256), 0, ctx->stream,
                           (const Key<W> *)in, cin, n, k, (Key<W> *)out, cout);
    else
        hipLaunchKernelGGL((k_expand_rc<W, false>), bbk::grid_blocks((n + 255) / 256), dim3(256), 0, ctx->stream,
                           (const Key<W> *)in, cin, n, k, (Key<W> *)out, cout);
    check_launch("k_expand_rc");
}

#define BBK_DISPATCH_W(W, ...)                                                   \
    switch (W) {                                                                 \
        case 1: { constexpr int W_ = 1; __VA_ARGS__; } break;                           \
        case 2: { constexpr int W_ = 2; __VA_ARGS__; } break;                           \
        case 3: { constexpr int W_ = 3; __VA_ARGS__; } break;                           \
        case 4: { constexpr int W_ = 4; __VA_ARGS__; } break;                           \
        default: BBK_REQUIRE(false, BBK_ERR_ARG, "unsupported key width %d", (int)(W)); \
    }

uint64_t drop_zero_vals(bbk_ctx *ctx, int W, const void *keys, const uint32_t *vals, uint64_t n, DevBuf &out_keys,
                        DevBuf &out_vals);

// ---- stage A: distinct canonical records (+ reduced payload) of one batch of reads, in ANY order ------------
// payload: with_mask -> OR of the InOutMask bits of every occurrence; want_vals -> multiplicity.
// MSD path (hash-partitioned dedup in LDS, hash-range passes above one device pass); LSD fallback = extract, sort,
// unique (its output happens to be ascending).
void dedup_reads(bbk_ctx *ctx, const bbk_reads *rd, unsigned k, bool with_mask, bool want_vals, DevBuf &out_keys,
                 DevBuf &out_vals, uint64_t &n_distinct, uint64_t &n_instances) {
    const int W = (int)words_of(k);
    n_distinct = 0;
    n_instances = 0;
    if (msd_enabled()) {
        const int op1 = with_mask ? MSD_OP_OR : (want_vals ? MSD_OP_COUNT : MSD_OP_NONE);
        if (superk_dedup_reads(ctx, rd, k, op1, out_keys, out_vals, n_distinct, n_instances)) return;
        n_distinct = 0;
        n_instances = 0;
        MsdOutput a;
        if (msd_sort_reduce(ctx, k, MSD_HASH, op1, rd, nullptr, nullptr, 0, with_mask, a)) {
            n_instances = a.instances;
            n_distinct = a.n;
            out_keys = std::move(a.keys);
            if (op1 != MSD_OP_NONE) out_vals = std::move(a.vals);
            return;
        }
    }
    DevBuf koff((rd->n + 1) * sizeof(uint64_t));
    if (rd->n) {
        hipLaunchKernelGGL(k_kmers_per_read, bbk::grid_blocks((rd->n + 255) / 256), dim3(256), 0, ctx->stream,
                           rd->d_len, rd->n, k, koff.as<uint64_t>());
        check_launch("k_kmers_per_read");
    }
    const uint64_t N = exclusive_scan_u64(ctx, koff.as<uint64_t>(), koff.as<uint64_t>(), rd->n);
    n_instances = N;
    if (N == 0) {
        out_keys.alloc(16);
        out_vals.alloc(16);
        return;
    }
    BBK_REQUIRE(N < (1ull << 32), BBK_ERR_ARG,
                "batch holds %llu k-mer instances; the LSD path is limited to 2^32-1 per batch (push the reads in "
                "smaller batches)",
                (unsigned long long)N);
    const size_t rec = (size_t)W * 8;
    DevBuf keys(N * rec), tmp(N * rec), vals, vtmp;
    if (with_mask) {
        vals.alloc(N * 4);
        vtmp.alloc(N * 4);
    }
    BBK_DISPATCH_W(W, launch_extract<W_>(ctx, rd, koff.as<uint64_t>(), (int)k, keys.p,
                                         with_mask ? vals.as<uint32_t>() : nullptr, N));
    sort_records(ctx, W, keys.p, tmp.p, with_mask ? vals.as<uint32_t>() : nullptr,
                 with_mask ? vtmp.as<uint32_t>() : nullptr, N, key_passes(k));
    // distinct keys land in tmp (sized for the worst case), then are copied to an exact buffer
    uint32_t *rv = nullptr;
    if (with_mask || want_vals) {
        if (!vtmp.p) vtmp.alloc(N * 4);
        rv = vtmp.as<uint32_t>();
    }
    const uint64_t D = unique_records(ctx, W, keys.p, with_mask ? vals.as<uint32_t>() : nullptr, N, tmp.p, rv,
                                      with_mask ? REDUCE_OR : REDUCE_COUNT, /*drop_zero=*/false);
    keys.release();
    vals.release();
    out_keys.alloc(D * rec + 16);
    BBK_HIP(bbk::copy_async(out_keys.p, tmp.p, D * rec, hipMemcpyDeviceToDevice, ctx->stream));
    if (rv) {
        out_vals.alloc(D * 4 + 16);
        BBK_HIP(bbk::copy_async(out_vals.p, rv, D * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    BBK_HIP(hipStreamSynchronize(ctx->stream));
    n_distinct = D;
}

// LSD sort + unique of a record array (the general path behind every MSD call that declines)
static uint64_t lsd_sort_unique(bbk_ctx *ctx, unsigned k, const void *d_keys, const uint32_t *d_vals, uint64_t n,
                                ReduceOp rop, DevBuf &out_keys, DevBuf &out_vals) {
    const int W = (int)words_of(k);
    const size_t rec = (size_t)W * 8;
    BBK_REQUIRE(n < (1ull << 32), BBK_ERR_ARG, "%llu records exceed the LSD path's 2^32-1", (unsigned long long)n);
    DevBuf a(n * rec), b(n * rec), ca, cb;
    BBK_HIP(bbk::copy_async(a.p, d_keys, n * rec, hipMemcpyDeviceToDevice, ctx->stream));
    if (d_vals) {
        ca.alloc(n * 4);
        cb.alloc(n * 4);
        BBK_HIP(bbk::copy_async(ca.p, d_vals, n * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    sort_records(ctx, W, a.p, b.p, d_vals ? ca.as<uint32_t>() : nullptr, d_vals ? cb.as<uint32_t>() : nullptr, n,
                 key_passes(k));
    const uint64_t D = unique_records(ctx, W, a.p, d_vals ? ca.as<uint32_t>() : nullptr, n, b.p,
                                      d_vals ? cb.as<uint32_t>() : nullptr, rop, false);
    out_keys.alloc(D * rec + 16);
    BBK_HIP(bbk::copy_async(out_keys.p, b.p, D * rec, hipMemcpyDeviceToDevice, ctx->stream));
    if (d_vals) {
        out_vals.alloc(D * 4 + 16);
        BBK_HIP(bbk::copy_async(out_vals.p, cb.p, D * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    BBK_HIP(hipStreamSynchronize(ctx->stream));
    return D;
}

// Distinct records of a record array (payloads summed / OR-ed), in ANY order: the merge step of the streaming
// count (MergeKMers analogue, kmer_index_builder.hpp:281-365) and of the multi-GPU exchange.
static uint64_t dedup_keys(bbk_ctx *ctx, unsigned k, const void *d_keys, const uint32_t *d_vals, uint64_t n, int op,
                           DevBuf &out_keys, DevBuf &out_vals) {
    if (n == 0) {
        out_keys.alloc(16);
        if (d_vals) out_vals.alloc(16);
        return 0;
    }
    if (msd_enabled()) {
        MsdOutput m;
        if (msd_sort_reduce(ctx, k, MSD_HASH, op, nullptr, d_keys, d_vals, n, false, m)) {
            out_keys = std::move(m.keys);
            if (op != MSD_OP_NONE) out_vals = std::move(m.vals);
            return m.n;
        }
    }
    return lsd_sort_unique(ctx, k, d_keys, d_vals, n, op == MSD_OP_OR ? REDUCE_OR : REDUCE_SUM, out_keys, out_vals);
}

// ---- stage B: a distinct set -> ascending (word 0 most significant, adt/array_vector.hpp:114-123) ---------------
static uint64_t sort_distinct(bbk_ctx *ctx, unsigned k, const void *d_keys, const uint32_t *d_vals, uint64_t n, int op,
                              DevBuf &out_keys, DevBuf &out_vals) {
    if (n == 0) {
        out_keys.alloc(16);
        if (d_vals) out_vals.alloc(16);
        return 0;
    }
    if (msd_enabled()) {
        MsdOutput b;
        // the input is distinct: stage B only orders it (assume_distinct: the sorted result is written directly)
        if (msd_sort_reduce(ctx, k, MSD_KEYS, d_vals ? op : MSD_OP_NONE, nullptr, d_keys, d_vals, n, false, b, 0u,
                            /*assume_distinct=*/true)) {
            out_keys = std::move(b.keys);
            if (d_vals) out_vals = std::move(b.vals);
            return b.n;
        }
    }
    return lsd_sort_unique(ctx, k, d_keys, d_vals, n, op == MSD_OP_OR ? REDUCE_OR : REDUCE_SUM, out_keys, out_vals);
}

// ---- accumulator of the streaming entry points -----------------------------------------------------------------
// What the reference gets from bounded per-thread cells, repeated DumpBuffers rounds (one sorted + uniqued run per
// bucket and round, kmer_splitter.hpp:73-167) and the final loser-tree run merge (MergeKMers,
// kmer_index_builder.hpp:281-365): the input never has to be resident as a whole.  Here every pushed batch is
// deduplicated on its own (stage A) and kept as a "run" of distinct canonical records; runs are merge-uniqued into
// the accumulated set whenever they outweigh half of it (so the total merge work stays linear in the input), and
// finish() orders the set once.
bool Accum::has_vals() const { return with_mask || want_vals; }
int Accum::merge_op() const { return with_mask ? MSD_OP_OR : (want_vals ? MSD_OP_SUM : MSD_OP_NONE); }

void Accum::push(const bbk_reads *rd) {
    auto one = [&](const bbk_reads *part) {
        Run r;
        uint64_t inst = 0;
        dedup_reads(ctx, part, k, with_mask, want_vals, r.keys, r.vals, r.n, inst);
        instances += inst;
        if (r.n == 0) return;
        runs_n += r.n;
        runs.push_back(std::move(r));
    };
    // 8-byte keys: one call above one pass of stage A (1.6 G instances with narrow records) would run in hash ranges,
    // and every range re-extracts every k-mer of the reads (100 M x 150 bp, k = 21: 7 ranges, 525 of 615 ms).  The
    // reads are cut into pieces of one pass each instead -- what a caller that pushes blocks gets anyway -- and the runs
    // are merged once.  (Wider keys go through super-k-mer records, which never re-extract.)
    const char *ec = getenv("BBK_READ_CHUNK");  // tests: bases per piece
    const uint64_t piece = ec ? strtoull(ec, nullptr, 10) : 1500000000ull;
    if (words_of(k) == 1 && rd->n >= 2 && rd->bases > piece && msd_enabled()) {
        const uint64_t np = (rd->bases + piece - 1) / piece, per = (rd->n + np - 1) / np;
        if (getenv("BBK_VERBOSE"))
            fprintf(stderr, "[bbk] count: %llu reads in %llu pieces of %llu\n", (unsigned long long)rd->n,
                    (unsigned long long)((rd->n + per - 1) / per), (unsigned long long)per);
        for (uint64_t r0 = 0; r0 < rd->n; r0 += per) {
            bbk_reads v;  // a view: owns nothing
            v.ctx = rd->ctx;
            v.n = std::min<uint64_t>(per, rd->n - r0);
            v.n_words = rd->n_words;
            v.bases = (uint64_t)((double)rd->bases * (double)v.n / (double)rd->n);
            v.d_words = rd->d_words;
            v.d_woff = rd->d_woff + r0;
            v.d_len = rd->d_len + r0;
            one(&v);
        }
    } else {
        one(rd);
    }
    ++batches;
    static const char *e = getenv("BBK_MERGE_MIN");  // tests force a merge after every push
    const uint64_t floor_n = e ? strtoull(e, nullptr, 10) : (32ull << 20);
    if ((runs.size() > 1 || n) && runs_n >= n / 2 + floor_n) merge();
}

void Accum::push_records(const void *d_keys, const uint32_t *d_vals, uint64_t n_rec) {
    if (n_rec == 0) return;
    Run r;
    r.n = dedup_keys(ctx, k, d_keys, has_vals() ? d_vals : nullptr, n_rec, merge_op(), r.keys, r.vals);
    instances += n_rec;
    ++batches;
    runs_n += r.n;
    runs.push_back(std::move(r));
}

// accumulated set + runs -> accumulated set
void Accum::merge() {
    if (runs.empty()) return;
    if (n == 0 && runs.size() == 1) {  // first batch: adopt
        keys = std::move(runs[0].keys);
        vals = std::move(runs[0].vals);
        n = runs[0].n;
        runs.clear();
        runs_n = 0;
        return;
    }
    const size_t rec = (size_t)words_of(k) * 8;
    const uint64_t total = n + runs_n;
    DevBuf ck(total * rec + 16), cv;
    if (has_vals()) cv.alloc(total * 4 + 16);
    uint64_t o = 0;
    auto put = [&](DevBuf &kb, DevBuf &vb, uint64_t cnt) {
        if (!cnt) return;
        BBK_HIP(bbk::copy_async(ck.as<char>() + o * rec, kb.p, cnt * rec, hipMemcpyDeviceToDevice, ctx->stream));
        if (has_vals())
            BBK_HIP(bbk::copy_async(cv.as<uint32_t>() + o, vb.p, cnt * 4, hipMemcpyDeviceToDevice, ctx->stream));
        o += cnt;
        BBK_HIP(hipStreamSynchronize(ctx->stream));
        kb.release();
        vb.release();
    };
    put(keys, vals, n);
    for (Run &r : runs) put(r.keys, r.vals, r.n);
    runs.clear();
    runs_n = 0;
    n = dedup_keys(ctx, k, ck.p, has_vals() ? cv.as<uint32_t>() : nullptr, total, merge_op(), keys, vals);
    ++merges;
}

// the accumulated set in ascending order (payloads alongside); the accumulator is left empty
uint64_t Accum::finish_sorted(DevBuf &out_keys, DevBuf &out_vals) {
    merge();
    const uint64_t D = sort_distinct(ctx, k, keys.p, has_vals() ? vals.as<uint32_t>() : nullptr, n, merge_op(), out_keys,
                                     out_vals);
    keys.release();
    vals.release();
    n = 0;
    return D;
}

// canon U rc(canon): expand, sort, unique.  A k-mer equal to its own RC (even k) appears twice
// and is merged by unique; its count doubles, as in the reference where both strands of such an
// occurrence are counted.
// want_ref: leave the set in the final_kmers order when that is free (tagged sort); s.ref_order tells.
static void expand_both_strands(bbk_ctx *ctx, unsigned k, const DevBuf &ck, const DevBuf *cv, uint64_t D,
                                bbk_kmerset &s, bool want_ref = false) {
    const bool wc = cv != nullptr;
    const int W = (int)words_of(k);
    const size_t rec = (size_t)W * 8;
    if (D == 0) {
        s.n = 0;
        s.keys.alloc(16);
        if (wc) s.counts.alloc(16);
        return;
    }
    // (sets above one device pass -- 2^32 records and far less for wide keys -- are sorted in key-range passes inside
    // msd_sort_reduce; only the LSD fallback below keeps the 32-bit limit)
    bool tag = want_ref && W == 1 && 2 * k + 4 <= 64 && msd_enabled();
    if (msd_enabled()) {
        // fused: the level-1 partition kernels generate key, reverse complement (and tag) from the canonical array
        // themselves -- the expanded array is never written
        MsdOutput m;
        // REF prefix (XXH3 bucket above the key bits: the final_kmers order straight from the sort) for every key
        // width; BBK_NO_WIDE_REF=1: keys above 16 bytes are sorted ascending and take one more stable pass on the bucket
        const bool ref_prefix = want_ref && !tag && (W <= 2 || getenv("BBK_NO_WIDE_REF") == nullptr);
        if (msd_sort_reduce(ctx, k, ref_prefix ? MSD_REF : MSD_KEYS, wc ? MSD_OP_SUM : MSD_OP_NONE, nullptr, ck.p,
                            wc ? cv->as<uint32_t>() : nullptr, D, false, m, tag ? 4u : 0u,
                            /*assume_distinct: odd k has no self-reverse-complementary k-mers*/ (k & 1) != 0,
                            /*expand_k=*/k)) {
            s.n = m.n;
            s.keys = std::move(m.keys);
            if (wc) s.counts = std::move(m.vals);
            s.ref_order = tag || ref_prefix;
            return;
        }
    }
    // general path: materialise the expanded array, sort, unique
    BBK_REQUIRE(2 * D < (1ull << 32), BBK_ERR_ARG, "too many distinct k-mers for the LSD path (%llu)", (unsigned long long)D);
    DevBuf e(2 * D * rec), et(2 * D * rec), ec, ect;
    if (wc) {
        ec.alloc(2 * D * 4);
        ect.alloc(2 * D * 4);
    }
    BBK_DISPATCH_W(W, launch_expand<W_>(ctx, ck.p, wc ? cv->as<uint32_t>() : nullptr, D, (int)k, e.p,
                                        wc ? ec.as<uint32_t>() : nullptr, false));
    sort_records(ctx, W, e.p, et.p, wc ? ec.as<uint32_t>() : nullptr, wc ? ect.as<uint32_t>() : nullptr, 2 * D,
                 key_passes(k));
    const uint64_t D2 = unique_records(ctx, W, e.p, wc ? ec.as<uint32_t>() : nullptr, 2 * D, et.p,
                                       wc ? ect.as<uint32_t>() : nullptr, wc ? REDUCE_SUM : REDUCE_COUNT, false);
    e.release();
    ec.release();
    s.n = D2;
    s.keys.alloc(D2 * rec);
    BBK_HIP(bbk::copy_async(s.keys.p, et.p, D2 * rec, hipMemcpyDeviceToDevice, ctx->stream));
    if (wc) {
        s.counts.alloc(D2 * 4);
        BBK_HIP(bbk::copy_async(s.counts.p, ect.p, D2 * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    BBK_HIP(hipStreamSynchronize(ctx->stream));
}

static void check_k(unsigned k) {
    // KMerCounter accepts any k in [1, MAX_K) (projects/kmercount/main.cpp has no parity check)
    BBK_REQUIRE(k >= 1 && k < BBK_MAX_K, BBK_ERR_ARG, "k-mer size %u out of range [1,%d)", k, BBK_MAX_K);
}

// finish of a count: the accumulated distinct canonical set -> the bbk_kmerset the flags ask for
static bbk_kmerset *finish_count(Accum &acc, unsigned flags) {
    bbk_ctx *ctx = acc.ctx;
    const unsigned k = acc.k;
    const bool both = (flags & BBK_BOTH_STRANDS) != 0;
    const bool wc = (flags & (BBK_WITH_COUNTS | BBK_WITH_MASKS)) != 0;  // a u32 payload travels with the keys
    const bool want_ref = (flags & BBK_REFERENCE_ORDER) != 0;
    auto s = std::make_unique<bbk_kmerset>();
    s->k = k;
    s->W = words_of(k);
    s->flags = flags;
    s->has_counts = wc;
    s->instances = both ? 2 * acc.instances : acc.instances;
    acc.merge();
    if (both) {
        expand_both_strands(ctx, k, acc.keys, wc ? &acc.vals : nullptr, acc.n, *s, want_ref);
        acc.keys.release();
        acc.vals.release();
    } else if (flags & BBK_UNSORTED) {
        s->n = acc.n;
        s->sorted = false;
        s->keys = std::move(acc.keys);
        if (wc) s->counts = std::move(acc.vals);
        if (!s->keys.p) s->keys.alloc(16);
    } else {
        s->n = acc.finish_sorted(s->keys, s->counts);
    }
    acc.n = 0;
    if (want_ref && !s->ref_order && s->sorted) {  // an ascending set -> the final_kmers order: one stable pass on the
        if (s->n) {                                // XXH3 bucket (key widths without room for the tag / REF prefix)
            const PassDesc pd{1, 0, 0, 8, 16};
            DevBuf nk(s->n * (size_t)s->W * 8), nc;
            if (wc) nc.alloc(s->n * 4);
            partition_records(ctx, (int)s->W, s->keys.p, nk.p, wc ? s->counts.as<uint32_t>() : nullptr,
                              wc ? nc.as<uint32_t>() : nullptr, s->n, pd);
            s->keys = std::move(nk);
            if (wc) s->counts = std::move(nc);
        }
        s->ref_order = true;
    }
    return s.release();
}

// The both-strand set of spades-kmercount from an accumulator that is NOT consumed (bbk_extindex_finish_with_set: the
// extension index is built from the same canonical records afterwards).  The payloads (mask bits) are not carried over.
bbk_kmerset *both_strands_of(Accum &acc, unsigned flags) {
    BBK_REQUIRE((flags & BBK_BOTH_STRANDS) && !(flags & (BBK_WITH_COUNTS | BBK_WITH_MASKS | BBK_UNSORTED | BBK_CANONICAL)),
                BBK_ERR_ARG, "the set built beside an extension index is BBK_BOTH_STRANDS [| BBK_REFERENCE_ORDER]");
    auto s = std::make_unique<bbk_kmerset>();
    s->k = acc.k;
    s->W = words_of(acc.k);
    s->flags = flags;
    s->has_counts = false;
    s->instances = 2 * acc.instances;
    acc.merge();
    const bool want_ref = (flags & BBK_REFERENCE_ORDER) != 0;
    expand_both_strands(acc.ctx, acc.k, acc.keys, nullptr, acc.n, *s, want_ref);
    if (want_ref && !s->ref_order && s->sorted) {
        if (s->n) {
            const PassDesc pd{1, 0, 0, 8, 16};
            DevBuf nk(s->n * (size_t)s->W * 8);
            partition_records(acc.ctx, (int)s->W, s->keys.p, nk.p, nullptr, nullptr, s->n, pd);
            s->keys = std::move(nk);
        }
        s->ref_order = true;
    }
    return s.release();
}

static void check_count_flags(unsigned flags) {
    const bool both = (flags & BBK_BOTH_STRANDS) != 0, canon = (flags & BBK_CANONICAL) != 0;
    BBK_REQUIRE(both != canon, BBK_ERR_ARG, "bbk_count: pass exactly one of BBK_BOTH_STRANDS / BBK_CANONICAL");
    BBK_REQUIRE(!((flags & BBK_REFERENCE_ORDER) && (flags & BBK_UNSORTED)), BBK_ERR_ARG,
                "bbk_count: BBK_REFERENCE_ORDER and BBK_UNSORTED exclude each other");
    BBK_REQUIRE(!(flags & BBK_WITH_MASKS) || (canon && !(flags & BBK_WITH_COUNTS)), BBK_ERR_ARG,
                "bbk_count: BBK_WITH_MASKS goes with BBK_CANONICAL and without BBK_WITH_COUNTS");
}

}  // namespace bbk

using namespace bbk;

struct bbk_counter {
    bbk::Accum acc;
    unsigned flags = 0;
};

extern "C" {

int bbk_count_begin(bbk_ctx *ctx, unsigned k, unsigned flags, bbk_counter **out) {
    return guarded([&] {
        BBK_REQUIRE(ctx && out, BBK_ERR_ARG, "bbk_count_begin: NULL argument");
        check_k(k);
        check_count_flags(flags);
        auto c = std::make_unique<bbk_counter>();
        c->acc.ctx = ctx;
        c->acc.k = k;
        c->acc.want_vals = (flags & BBK_WITH_COUNTS) != 0;
        c->acc.with_mask = (flags & BBK_WITH_MASKS) != 0;
        c->flags = flags;
        *out = c.release();
    });
}

int bbk_count_push_reads(bbk_counter *c, const bbk_reads *reads) {
    return guarded([&] {
        BBK_REQUIRE(c && reads, BBK_ERR_ARG, "bbk_count_push_reads: NULL argument");
        BBK_HIP(hipSetDevice(c->acc.ctx->device));
        c->acc.push(reads);
    });
}

int bbk_count_push_ascii(bbk_counter *c, const char *h_bases, const uint64_t *h_offsets, uint64_t n_reads) {
    if (!c) {
        set_error("bbk_count_push_ascii: NULL argument");
        return BBK_ERR_ARG;
    }
    bbk_reads *r = nullptr;
    const int rc = bbk_reads_from_ascii(c->acc.ctx, h_bases, h_offsets, n_reads, &r);
    if (rc != BBK_OK) return rc;
    const int rc2 = bbk_count_push_reads(c, r);
    bbk_reads_free(r);
    return rc2;
}

int bbk_count_finish(bbk_counter *c, bbk_kmerset **out) {
    const int rc = guarded([&] {
        BBK_REQUIRE(c && out, BBK_ERR_ARG, "bbk_count_finish: NULL argument");
        BBK_HIP(hipSetDevice(c->acc.ctx->device));
        *out = finish_count(c->acc, c->flags);
    });
    delete c;  // finished or failed: the counter is gone either way
    return rc;
}

void bbk_count_abort(bbk_counter *c) { delete c; }

uint64_t bbk_count_pushed_instances(const bbk_counter *c) { return c ? c->acc.instances : 0; }

int bbk_count(bbk_ctx *ctx, const bbk_reads *reads, unsigned k, unsigned flags, bbk_kmerset **out) {
    return guarded([&] {
        BBK_REQUIRE(ctx && reads && out, BBK_ERR_ARG, "bbk_count: NULL argument");
        check_k(k);
        check_count_flags(flags);
        BBK_HIP(hipSetDevice(ctx->device));
        // one batch through the streaming accumulator: hash-partitioned dedup of the canonical stream (stage A),
        // then expand + sort once (stage B)
        Accum acc;
        acc.ctx = ctx;
        acc.k = k;
        acc.want_vals = (flags & BBK_WITH_COUNTS) != 0;
        acc.with_mask = (flags & BBK_WITH_MASKS) != 0;
        acc.push(reads);
        *out = finish_count(acc, flags);
    });
}

const void *bbk_kmerset_keys(const bbk_kmerset *s, unsigned *order) {
    if (!s) return nullptr;
    if (order) *order = !s->sorted ? 0xFFFFFFFFu : (s->ref_order ? BBK_ORDER_REFERENCE_BUCKETS16 : BBK_ORDER_SORTED);
    return s->keys.p;
}

int bbk_kmerset_from_device(bbk_ctx *ctx, const void *d_keys, const void *d_counts, uint64_t n, unsigned k,
                            bbk_kmerset **out) {
    return bbk_kmerset_from_device_ex(ctx, d_keys, d_counts, n, k, 0, out);
}

int bbk_kmerset_from_device_ex(bbk_ctx *ctx, const void *d_keys, const void *d_counts, uint64_t n, unsigned k,
                               unsigned flags, bbk_kmerset **out) {
    return guarded([&] {
        BBK_REQUIRE(ctx && out && (n == 0 || d_keys), BBK_ERR_ARG, "bbk_kmerset_from_device: NULL argument");
        check_k(k);
        BBK_HIP(hipSetDevice(ctx->device));
        auto s = std::make_unique<bbk_kmerset>();
        s->k = k;
        s->W = words_of(k);
        s->has_counts = d_counts != nullptr;
        s->instances = n;
        const size_t rec = (size_t)s->W * 8;
        if (n == 0) {
            s->keys.alloc(16);
            if (d_counts) s->counts.alloc(16);
            *out = s.release();
            return;
        }
        if (msd_enabled()) {
            MsdOutput m;
            const bool unsorted = (flags & BBK_UNSORTED) != 0;
            if (msd_sort_reduce(ctx, k, unsorted ? MSD_HASH : MSD_KEYS, d_counts ? MSD_OP_SUM : MSD_OP_NONE, nullptr,
                                d_keys, (const uint32_t *)d_counts, n, false, m)) {
                s->n = m.n;
                s->sorted = !unsorted;
                s->keys = std::move(m.keys);
                if (d_counts) s->counts = std::move(m.vals);
                *out = s.release();
                return;
            }
        }
        DevBuf a(n * rec), b(n * rec), ca, cb;
        BBK_HIP(bbk::copy_async(a.p, d_keys, n * rec, hipMemcpyDeviceToDevice, ctx->stream));
        if (d_counts) {
            ca.alloc(n * 4);
            cb.alloc(n * 4);
            BBK_HIP(bbk::copy_async(ca.p, d_counts, n * 4, hipMemcpyDeviceToDevice, ctx->stream));
        }
        sort_records(ctx, (int)s->W, a.p, b.p, d_counts ? ca.as<uint32_t>() : nullptr,
                     d_counts ? cb.as<uint32_t>() : nullptr, n, key_passes(k));
        const uint64_t D = unique_records(ctx, (int)s->W, a.p, d_counts ? ca.as<uint32_t>() : nullptr, n, b.p,
                                          d_counts ? cb.as<uint32_t>() : nullptr,
                                          d_counts ? REDUCE_SUM : REDUCE_COUNT, false);
        s->n = D;
        s->keys.alloc(D * rec);
        BBK_HIP(bbk::copy_async(s->keys.p, b.p, D * rec, hipMemcpyDeviceToDevice, ctx->stream));
        if (d_counts) {
            s->counts.alloc(D * 4);
            BBK_HIP(bbk::copy_async(s->counts.p, cb.p, D * 4, hipMemcpyDeviceToDevice, ctx->stream));
        }
        BBK_HIP(hipStreamSynchronize(ctx->stream));
        *out = s.release();
    });
}

int bbk_kmerset_both_strands(bbk_ctx *ctx, const bbk_kmerset *canon, bbk_kmerset **out) {
    return bbk_kmerset_both_strands_ex(ctx, canon, 0, out);
}

int bbk_kmerset_both_strands_ex(bbk_ctx *ctx, const bbk_kmerset *canon, unsigned flags, bbk_kmerset **out) {
    return guarded([&] {
        BBK_REQUIRE(ctx && canon && out, BBK_ERR_ARG, "bbk_kmerset_both_strands: NULL argument");
        BBK_REQUIRE((flags & ~BBK_REFERENCE_ORDER) == 0, BBK_ERR_ARG,
                    "bbk_kmerset_both_strands_ex: only BBK_REFERENCE_ORDER is accepted");
        BBK_HIP(hipSetDevice(ctx->device));
        const bool want_ref = (flags & BBK_REFERENCE_ORDER) != 0;
        auto s = std::make_unique<bbk_kmerset>();
        s->k = canon->k;
        s->W = canon->W;
        s->flags = (canon->flags & ~(BBK_CANONICAL | BBK_UNSORTED)) | BBK_BOTH_STRANDS | flags;
        s->has_counts = canon->has_counts;
        s->instances = 2 * canon->instances;
        expand_both_strands(ctx, canon->k, canon->keys, canon->has_counts ? &canon->counts : nullptr, canon->n, *s,
                            want_ref);
        if (want_ref && !s->ref_order) {  // key widths without room for the tag: one stable pass on the bucket digit
            if (s->n) {
                const PassDesc pd{1, 0, 0, 8, 16};
                const bool wc = s->has_counts;
                DevBuf nk(s->n * (size_t)s->W * 8), nc;
                if (wc) nc.alloc(s->n * 4);
                partition_records(ctx, (int)s->W, s->keys.p, nk.p, wc ? s->counts.as<uint32_t>() : nullptr,
                                  wc ? nc.as<uint32_t>() : nullptr, s->n, pd);
                s->keys = std::move(nk);
                if (wc) s->counts = std::move(nc);
            }
            s->ref_order = true;
        }
        *out = s.release();
    });
}

uint64_t bbk_kmerset_size(const bbk_kmerset *s) { return s ? s->n : 0; }
unsigned bbk_kmerset_k(const bbk_kmerset *s) { return s ? s->k : 0; }
uint64_t bbk_kmerset_instances(const bbk_kmerset *s) { return s ? s->instances : 0; }

static bool is_device_ptr(const void *p) {
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();  // plain host memory is not an error
        return false;
    }
    return at.type == hipMemoryTypeDevice;
}

static void export_ordered(bbk_ctx *ctx, const bbk_kmerset *s, const PassDesc *pd, void *dst_keys, void *dst_counts,
                           uint64_t *h_counts) {
    BBK_HIP(hipSetDevice(ctx->device));
    // an unsorted set may only be partitioned by owner (order inside a segment is irrelevant there)
    BBK_REQUIRE(s->sorted || (pd && pd->kind == 2), BBK_ERR_ARG,
                "this k-mer set was built with BBK_UNSORTED: it can be owner-partitioned, expanded or re-merged, "
                "not exported in key order");
    const size_t rec = (size_t)s->W * 8;
    if (h_counts && pd) memset(h_counts, 0, pd->nb * sizeof(uint64_t));
    if (s->n == 0) return;
    const bool wc = s->has_counts && dst_counts;
    if (s->ref_order && !pd) {
        // ascending export of a set stored in the final_kmers order: sort a copy (not a hot path)
        DevBuf a(s->n * rec), b(s->n * rec), ca, cb;
        BBK_HIP(bbk::copy_async(a.p, s->keys.p, s->n * rec, hipMemcpyDeviceToDevice, ctx->stream));
        if (wc) {
            ca.alloc(s->n * 4);
            cb.alloc(s->n * 4);
            BBK_HIP(bbk::copy_async(ca.p, s->counts.p, s->n * 4, hipMemcpyDeviceToDevice, ctx->stream));
        }
        sort_records(ctx, (int)s->W, a.p, b.p, wc ? ca.as<uint32_t>() : nullptr, wc ? cb.as<uint32_t>() : nullptr, s->n,
                     key_passes(s->k));
        BBK_HIP(bbk::copy_async(dst_keys, a.p, s->n * rec, hipMemcpyDefault, ctx->stream));
        if (wc) BBK_HIP(bbk::copy_async(dst_counts, ca.p, s->n * 4, hipMemcpyDefault, ctx->stream));
        BBK_HIP(hipStreamSynchronize(ctx->stream));
        return;
    }
    if (!pd || (s->ref_order && pd->kind == 1 && pd->nb == 16)) {  // stored in the requested order: plain copy
        BBK_HIP(bbk::copy_async(dst_keys, s->keys.p, s->n * rec, hipMemcpyDefault, ctx->stream));
        if (wc) BBK_HIP(bbk::copy_async(dst_counts, s->counts.p, s->n * 4, hipMemcpyDefault, ctx->stream));
        BBK_HIP(hipStreamSynchronize(ctx->stream));
        return;
    }
    // One stable pass on the bucket digit keeps the ascending order inside each bucket.  It reads the set
    // and writes straight into the caller's buffer when that is device memory (no staging copies).
    const bool kdev = is_device_ptr(dst_keys), cdev = !wc || is_device_ptr(dst_counts);
    DevBuf tk, tc;
    if (!kdev) tk.alloc(s->n * rec);
    if (wc && !cdev) tc.alloc(s->n * 4);
    void *ok = kdev ? dst_keys : tk.p;
    uint32_t *oc = wc ? (cdev ? (uint32_t *)dst_counts : tc.as<uint32_t>()) : nullptr;
    uint64_t hc[256];  // the pass counts the digit values anyway: the per-owner counts come for free
    partition_records(ctx, (int)s->W, s->keys.p, ok, wc ? s->counts.as<uint32_t>() : nullptr, oc, s->n, *pd,
                      h_counts ? hc : nullptr);
    if (h_counts)
        for (unsigned i = 0; i < pd->nb; ++i) h_counts[i] = hc[i];
    if (!kdev) BBK_HIP(hipMemcpyAsync(dst_keys, tk.p, s->n * rec, hipMemcpyDeviceToHost, ctx->stream));
    if (wc && !cdev) BBK_HIP(hipMemcpyAsync(dst_counts, tc.p, s->n * 4, hipMemcpyDeviceToHost, ctx->stream));
    BBK_HIP(hipStreamSynchronize(ctx->stream));
}

int bbk_kmerset_export(bbk_ctx *ctx, const bbk_kmerset *s, unsigned order, void *dst_keys, void *dst_counts) {
    return guarded([&] {
        BBK_REQUIRE(ctx && s && (s->n == 0 || dst_keys), BBK_ERR_ARG, "bbk_kmerset_export: NULL argument");
        BBK_REQUIRE(order == BBK_ORDER_SORTED || order == BBK_ORDER_REFERENCE_BUCKETS16, BBK_ERR_ARG,
                    "bbk_kmerset_export: unknown order %u", order);
        if (order == BBK_ORDER_SORTED) {
            export_ordered(ctx, s, nullptr, dst_keys, dst_counts, nullptr);
        } else {
            const PassDesc pd{1, 0, 0, 8, 16};  // CountAll(16, ...) (projects/kmercount/main.cpp:215)
            export_ordered(ctx, s, &pd, dst_keys, dst_counts, nullptr);
        }
    });
}

int bbk_kmerset_export_by_owner(bbk_ctx *ctx, const bbk_kmerset *s, unsigned nranks, void *dst_keys,
                                void *dst_counts, uint64_t *h_counts) {
    return guarded([&] {
        BBK_REQUIRE(ctx && s && h_counts && (s->n == 0 || dst_keys), BBK_ERR_ARG,
                    "bbk_kmerset_export_by_owner: NULL argument");
        BBK_REQUIRE(nranks >= 1 && nranks <= 256, BBK_ERR_ARG, "bbk_kmerset_export_by_owner: nranks %u not in [1,256]",
                    nranks);
        const PassDesc pd{2, 0, 0, 8, nranks};
        export_ordered(ctx, s, &pd, dst_keys, dst_counts, h_counts);
    });
}

void bbk_kmerset_free(bbk_kmerset *s) { delete s; }

int bbk_kmerset_verify_order(bbk_ctx *ctx, const bbk_kmerset *s, uint64_t *n_runs, uint64_t *n_equal,
                             uint64_t *h_run_starts, unsigned cap) {
    return guarded([&] {
        BBK_REQUIRE(ctx && s && n_runs && n_equal, BBK_ERR_ARG, "bbk_kmerset_verify_order: NULL argument");
        BBK_HIP(hipSetDevice(ctx->device));
        *n_runs = s->n ? 1 : 0;
        *n_equal = 0;
        if (h_run_starts && cap) h_run_starts[0] = 0;
        if (s->n < 2) return;
        DevBuf d(8 * 40);
        BBK_HIP(hipMemsetAsync(d.p, 0, 8 * 40, ctx->stream));
        const uint64_t nblk = (s->n + 255) / 256;
        BBK_DISPATCH_W(s->W, hipLaunchKernelGGL((k_order_check<W_>), bbk::grid_blocks(nblk), dim3(256), 0, ctx->stream,
                                                (const Key<W_> *)s->keys.p, s->n,
                                                (unsigned long long *)d.p));
        check_launch("k_order_check");
        unsigned long long h[40];
        BBK_HIP(hipMemcpyAsync(h, d.p, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
        BBK_HIP(hipStreamSynchronize(ctx->stream));
        *n_runs = 1 + h[0];
        *n_equal = h[1];
        if (h_run_starts) {
            const unsigned got = (unsigned)std::min<unsigned long long>(h[0], 32ull);
            std::sort(h + 2, h + 2 + got);
            for (unsigned i = 0; i < got && i + 1 < cap; ++i) h_run_starts[i + 1] = h[2 + i] + 1;
        }
    });
}

int bbk_kmerset_bucket_offsets(bbk_ctx *ctx, const bbk_kmerset *s, uint64_t *h_offsets) {
    return guarded([&] {
        BBK_REQUIRE(ctx && s && h_offsets, BBK_ERR_ARG, "bbk_kmerset_bucket_offsets: NULL argument");
        BBK_REQUIRE(s->ref_order, BBK_ERR_ARG, "bbk_kmerset_bucket_offsets: the set is not in the final_kmers order");
        BBK_HIP(hipSetDevice(ctx->device));
        if (s->n == 0) {
            for (int b = 0; b <= 16; ++b) h_offsets[b] = 0;
            return;
        }
        DevBuf d(17 * 8);
        BBK_DISPATCH_W(s->W, hipLaunchKernelGGL((k_bucket_bounds<W_>), dim3(1), dim3(64), 0, ctx->stream,
                                                (const Key<W_> *)s->keys.p, s->n, (unsigned long long *)d.p));
        check_launch("k_bucket_bounds");
        unsigned long long h[17];
        BBK_HIP(hipMemcpyAsync(h, d.p, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
        BBK_HIP(hipStreamSynchronize(ctx->stream));
        for (int b = 0; b <= 16; ++b) h_offsets[b] = h[b];
    });
}

int bbk_kmerset_get(bbk_ctx *ctx, const bbk_kmerset *s, uint64_t first, uint64_t count, void *h_keys, void *h_counts) {
    return guarded([&] {
        BBK_REQUIRE(ctx && s && (count == 0 || h_keys), BBK_ERR_ARG, "bbk_kmerset_get: NULL argument");
        BBK_REQUIRE(first <= s->n && count <= s->n - first, BBK_ERR_ARG, "bbk_kmerset_get: range [%llu, +%llu) outside the set",
                    (unsigned long long)first, (unsigned long long)count);
        BBK_HIP(hipSetDevice(ctx->device));
        if (!count) return;
        const size_t rec = (size_t)s->W * 8;
        BBK_HIP(hipMemcpyAsync(h_keys, s->keys.as<char>() + first * rec, count * rec, hipMemcpyDeviceToHost, ctx->stream));
        if (h_counts && s->has_counts)
            BBK_HIP(hipMemcpyAsync(h_counts, s->counts.as<uint32_t>() + first, count * 4, hipMemcpyDeviceToHost,
                                   ctx->stream));
        BBK_HIP(hipStreamSynchronize(ctx->stream));
    });
}

int bbk_kmerset_write_final_kmers(bbk_ctx *ctx, const bbk_kmerset *s, const char *path) {
    return guarded([&] {
        BBK_REQUIRE(ctx && s && path, BBK_ERR_ARG, "bbk_kmerset_write_final_kmers: NULL argument");
        BBK_HIP(hipSetDevice(ctx->device));
        const size_t rec = (size_t)s->W * 8;
        // the records in the final_kmers order on the device (the set itself when it was built in that order),
        // then device -> file through the pinned staging buffers: no host copy of the whole set
        DevBuf tmp;
        const void *src = s->keys.p;
        if (!s->ref_order && s->n) {
            tmp.alloc(s->n * rec);
            const PassDesc pd{1, 0, 0, 8, 16};
            export_ordered(ctx, s, &pd, tmp.p, nullptr, nullptr);
            src = tmp.p;
        }
        const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
        BBK_REQUIRE(fd >= 0, BBK_ERR_IO, "cannot open %s for writing", path);
        bool ok = false;
        try {
            ok = d2f_big(ctx, fd, 0, src, s->n * rec);
        } catch (...) {  // a HIP error inside the stream: no descriptor and no full-size, half-written file is left behind
            (void)close(fd);
            (void)unlink(path);
            throw;
        }
        const int cl = close(fd);
        if (!(ok && cl == 0)) (void)unlink(path);
        BBK_REQUIRE(ok && cl == 0, BBK_ERR_IO, "short write to %s", path);
    });
}

}  // extern "C"
