// extindex.hip -- DeBruijnExtensionIndex on the device.
//
// Reference (common/utils/extension_index/kmer_extension_index_builder.hpp:62-106) builds it in
// four steps: count canonical (k+1)-mers, derive canonical k-mers from them, build a BooPHF over
// the k-mers, then for every (k+1)-mer do two MPHF lookups + two `omp atomic` byte ORs
// (FillExtensionsFromIndex :44-59, InOutMask::AddOutgoing/AddIncoming kmer_extension_index.hpp:92-106).
// Here it is ONE sort-reduce: every k-mer position of every read emits (canonical k-mer, the
// in/out bits its two neighbouring (k+1)-mers give it, already conjugated when the k-mer is not
// minimal -- position 7-pos, kmer_extension_index.hpp:67-69); a radix sort by key and a
// segmented OR give the sorted table (canonical k-mer, InOutMask).  The sorted table itself is
// the index (prefix table + binary search replaces the MPHF, kmer_index.hpp:85-90).
// K-mers that never get a bit (reads of length exactly k) are dropped, as in the reference where
// k-mers only come from (k+1)-mers (kmer_splitters.hpp:160-180).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <memory>

#include "accum.h"
#include "bbk_internal.h"
#include "kmer_ops.h"

namespace bbk {

// four masks per thread, one dword store (the array is padded to a multiple of 16 bytes)
__global__ void k_u32_to_u8(const uint32_t *__restrict__ in, uint64_t n, uint8_t *__restrict__ out) {
    const uint64_t i = (BBK_GID()) * 4;
    if (i >= n) return;
    uint32_t w = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (i + j < n) w |= (in[i + j] & 0xFFu) << (8 * j);
    *reinterpret_cast<uint32_t *>(out + i) = w;
}

// pref[t] = first index whose top `bits` bits of word 0 are >= t  (t in [0, 2^bits]).  IDX = uint32_t below 2^32 - 2
// records, uint64_t above (KMerIndex::seq_idx is a size_t, utils/kmer_mph/kmer_index.hpp:85-90)
template <class IDX>
__global__ void k_prefix_table(const uint64_t *__restrict__ keys, int W, uint64_t n, int shift, uint64_t nbins,
                               IDX *__restrict__ pref) {
    const uint64_t i = BBK_GID();
    if (i >= n) return;
    const uint64_t t = keys[i * W] >> shift;
    const int64_t tp = (i == 0) ? -1 : (int64_t)(keys[(i - 1) * W] >> shift);
    for (int64_t u = tp + 1; u <= (int64_t)t; ++u) pref[u] = (IDX)i;
    if (i == n - 1)
        for (uint64_t u = t + 1; u <= nbins; ++u) pref[u] = (IDX)n;
}

// 64-bit table entries are chosen by size; BBK_WIDE_INDEX=1 forces them (tests run the wide graph stage on small inputs)
bool index_is_wide(uint64_t n) {
    return n >= (1ull << 32) - 2 || getenv("BBK_WIDE_INDEX") != nullptr;  // read per call: tests switch it inside a process
}

// Prefix table over an ascending key array: ~8 keys per bin; word 0 holds min(2k, 64) populated bits.
// Returns the number of prefix bits; lookups shift word 0 right by (w0bits - bits).  *wide: the entries are u64.
unsigned build_prefix_index(bbk_ctx *ctx, const uint64_t *keys, unsigned W, unsigned k, uint64_t n, DevBuf &prefix,
                            bool *wide) {
    const int w0bits = (W == 1) ? (int)(2 * k) : 64;
    const int max_bits = n > (1ull << 27) ? 30 : 24;
    int bits = 4;
    while (bits < max_bits && (1ull << (bits + 3)) < n) ++bits;
    bits = std::min(bits, w0bits);
    const uint64_t nbins = 1ull << bits;
    *wide = index_is_wide(n);
    const size_t esz = *wide ? sizeof(uint64_t) : sizeof(uint32_t);
    prefix.alloc(((size_t)nbins + 1) * esz);
    if (n == 0) {
        BBK_HIP(hipMemsetAsync(prefix.p, 0, ((size_t)nbins + 1) * esz, ctx->stream));
    } else {
        const uint64_t nblk = (n + 255) / 256;
        if (*wide)
            hipLaunchKernelGGL(k_prefix_table<uint64_t>, bbk::grid_blocks(nblk), dim3(256), 0, ctx->stream, keys, (int)W, n,
                               w0bits - bits, nbins, prefix.as<uint64_t>());
        else
            hipLaunchKernelGGL(k_prefix_table<uint32_t>, bbk::grid_blocks(nblk), dim3(256), 0, ctx->stream, keys, (int)W, n,
                               w0bits - bits, nbins, prefix.as<uint32_t>());
        check_launch("k_prefix_table");
    }
    BBK_HIP(hipStreamSynchronize(ctx->stream));
    return (unsigned)bits;
}

void build_prefix_table(bbk_ctx *ctx, bbk_extindex *x) {
    x->prefix_bits = build_prefix_index(ctx, x->keys.as<uint64_t>(), x->W, x->k, x->n, x->prefix, &x->prefix_wide);
}

// the accumulated (canonical k-mer, OR of mask bits) records -> the index: ascending keys, one InOutMask byte each
static bbk_extindex *finish_extindex(Accum &acc) {
    bbk_ctx *ctx = acc.ctx;
    auto x = std::make_unique<bbk_extindex>();
    x->k = acc.k;
    x->W = words_of(acc.k);
    x->instances = acc.instances;
    DevBuf m32;
    x->n = acc.finish_sorted(x->keys, m32);
    if (x->n) {
        // k-mers that never received a bit (reads of length exactly k) are not part of the index
        DevBuf fk, fv;
        const uint64_t kept = drop_zero_vals(ctx, (int)x->W, x->keys.p, m32.as<uint32_t>(), x->n, fk, fv);
        if (kept != x->n) {
            x->keys = std::move(fk);
            m32 = std::move(fv);
            x->n = kept;
        }
    }
    if (!x->keys.p) x->keys.alloc(16);
    x->masks.alloc(x->n + 16);
    if (x->n) {
        const uint64_t nblk = ((x->n + 3) / 4 + 255) / 256;
        hipLaunchKernelGGL(k_u32_to_u8, bbk::grid_blocks(nblk), dim3(256), 0, ctx->stream, m32.as<uint32_t>(), x->n,
                           x->masks.as<uint8_t>());
        check_launch("k_u32_to_u8");
    }
    build_prefix_table(ctx, x.get());
    return x.release();
}

static void check_ext_k(unsigned k) {
    // the index is built from (k+1)-mers, so k+1 must be a legal k-mer size too
    BBK_REQUIRE(k >= 1 && k + 1 < BBK_MAX_K, BBK_ERR_ARG, "k-mer size %u out of range [1,%d)", k, BBK_MAX_K - 1);
}

}  // namespace bbk

using namespace bbk;

struct bbk_extbuilder {
    bbk::Accum acc;
};

extern "C" {

int bbk_extindex_build(bbk_ctx *ctx, const bbk_reads *reads, unsigned k, bbk_extindex **out) {
    return guarded([&] {
        BBK_REQUIRE(ctx && reads && out, BBK_ERR_ARG, "bbk_extindex_build: NULL argument");
        check_ext_k(k);
        BBK_HIP(hipSetDevice(ctx->device));
        Accum acc;
        acc.ctx = ctx;
        acc.k = k;
        acc.with_mask = true;
        acc.push(reads);
        *out = finish_extindex(acc);
    });
}

int bbk_extindex_from_device(bbk_ctx *ctx, const void *d_keys, const void *d_masks_u32, uint64_t n, unsigned k,
                             bbk_extindex **out) {
    return guarded([&] {
        BBK_REQUIRE(ctx && out && (n == 0 || (d_keys && d_masks_u32)), BBK_ERR_ARG, "bbk_extindex_from_device: NULL argument");
        check_ext_k(k);
        BBK_HIP(hipSetDevice(ctx->device));
        Accum acc;
        acc.ctx = ctx;
        acc.k = k;
        acc.with_mask = true;
        acc.push_records(d_keys, (const uint32_t *)d_masks_u32, n);
        *out = finish_extindex(acc);
    });
}

__global__ void k_u8_to_u32(const uint8_t *__restrict__ in, uint64_t n, uint32_t *__restrict__ out) {
    const uint64_t i = BBK_GID();
    if (i < n) out[i] = in[i];
}

int bbk_extindex_export_u32(bbk_ctx *ctx, const bbk_extindex *x, void *dst_keys, void *dst_masks_u32) {
    return guarded([&] {
        BBK_REQUIRE(ctx && x, BBK_ERR_ARG, "bbk_extindex_export_u32: NULL argument");
        BBK_HIP(hipSetDevice(ctx->device));
        if (x->n == 0) return;
        if (dst_keys) BBK_HIP(bbk::copy_async(dst_keys, x->keys.p, x->n * x->W * 8, hipMemcpyDefault, ctx->stream));
        if (dst_masks_u32) {
            DevBuf m(x->n * 4);
            hipLaunchKernelGGL(k_u8_to_u32, bbk::grid_blocks((x->n + 255) / 256), dim3(256), 0, ctx->stream,
                               x->masks.as<uint8_t>(), x->n, m.as<uint32_t>());
            check_launch("k_u8_to_u32");
            BBK_HIP(bbk::copy_async(dst_masks_u32, m.p, x->n * 4, hipMemcpyDefault, ctx->stream));
            BBK_HIP(hipStreamSynchronize(ctx->stream));
        }
        BBK_HIP(hipStreamSynchronize(ctx->stream));
    });
}

int bbk_extindex_begin(bbk_ctx *ctx, unsigned k, bbk_extbuilder **out) {
    return guarded([&] {
        BBK_REQUIRE(ctx && out, BBK_ERR_ARG, "bbk_extindex_begin: NULL argument");
        check_ext_k(k);
        auto b = std::make_unique<bbk_extbuilder>();
        b->acc.ctx = ctx;
        b->acc.k = k;
        b->acc.with_mask = true;
        *out = b.release();
    });
}

int bbk_extindex_push_reads(bbk_extbuilder *b, const bbk_reads *reads) {
    return guarded([&] {
        BBK_REQUIRE(b && reads, BBK_ERR_ARG, "bbk_extindex_push_reads: NULL argument");
        BBK_HIP(hipSetDevice(b->acc.ctx->device));
        b->acc.push(reads);
    });
}

int bbk_extindex_finish(bbk_extbuilder *b, bbk_extindex **out) {
    const int rc = guarded([&] {
        BBK_REQUIRE(b && out, BBK_ERR_ARG, "bbk_extindex_finish: NULL argument");
        BBK_HIP(hipSetDevice(b->acc.ctx->device));
        *out = finish_extindex(b->acc);
    });
    delete b;
    return rc;
}

// Count + extension index from ONE pass over the reads (BASELINE configs[2]): stage A runs once with the mask payload;
// its canonical records are (a) expanded to the both-strand set of spades-kmercount -- the k-mers of reads of length
// exactly k, which carry no extension bit, included -- and (b) ordered, stripped of the records without a bit and
// turned into the index.  Equal to bbk_count_finish(BBK_BOTH_STRANDS ...) + bbk_extindex_finish on the same pushes.
int bbk_extindex_finish_with_set(bbk_extbuilder *b, unsigned set_flags, bbk_kmerset **set, bbk_extindex **out) {
    const int rc = guarded([&] {
        BBK_REQUIRE(b && set && out, BBK_ERR_ARG, "bbk_extindex_finish_with_set: NULL argument");
        BBK_HIP(hipSetDevice(b->acc.ctx->device));
        std::unique_ptr<bbk_kmerset, void (*)(bbk_kmerset *)> s(both_strands_of(b->acc, set_flags), bbk_kmerset_free);
        *out = finish_extindex(b->acc);
        *set = s.release();
    });
    delete b;
    return rc;
}

int bbk_count_extindex(bbk_ctx *ctx, const bbk_reads *reads, unsigned k, unsigned set_flags, bbk_kmerset **set,
                       bbk_extindex **out) {
    bbk_extbuilder *b = nullptr;
    int rc = bbk_extindex_begin(ctx, k, &b);
    if (rc != BBK_OK) return rc;
    rc = bbk_extindex_push_reads(b, reads);
    if (rc != BBK_OK) {
        bbk_extindex_abort(b);
        return rc;
    }
    return bbk_extindex_finish_with_set(b, set_flags, set, out);
}

void bbk_extindex_abort(bbk_extbuilder *b) { delete b; }

uint64_t bbk_extindex_size(const bbk_extindex *x) { return x ? x->n : 0; }
unsigned bbk_extindex_k(const bbk_extindex *x) { return x ? x->k : 0; }

int bbk_extindex_export(bbk_ctx *ctx, const bbk_extindex *x, void *dst_keys, void *dst_masks) {
    return guarded([&] {
        BBK_REQUIRE(ctx && x, BBK_ERR_ARG, "bbk_extindex_export: NULL argument");
        BBK_HIP(hipSetDevice(ctx->device));
        if (x->n == 0) return;
        if (dst_keys)
            BBK_HIP(bbk::copy_async(dst_keys, x->keys.p, x->n * x->W * 8, hipMemcpyDefault, ctx->stream));
        if (dst_masks) BBK_HIP(bbk::copy_async(dst_masks, x->masks.p, x->n, hipMemcpyDefault, ctx->stream));
        BBK_HIP(hipStreamSynchronize(ctx->stream));
    });
}

void bbk_extindex_free(bbk_extindex *x) { delete x; }

}  // extern "C"
