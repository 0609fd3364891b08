// readfilter.hip -- per-read median k-mer multiplicity test (the device side of spades-read-filter).
//
// Replaces io::CoverageFilter / CountMedianMlt (common/io/reads/coverage_filtering_read_wrapper.hpp:22-76) over the
// counting quotient filter the reference fills with FillCoverageHistogram (common/utils/kmer_counting.hpp,
// projects/kmercount/read_filter.cpp:151-158): a read is kept when the (upper) median of the multiplicities of its
// k-mers, strands identified, is >= threshold.  The CQF holds 64-bit hashes with counts capped at the threshold, so
// its answers are the exact counts wherever it matters; here the exact counts come from the engine's own canonical
// k-mer set (bbk_count(BBK_CANONICAL | BBK_WITH_COUNTS) of the same reads).
//   median index n = nk / 2 of the ascending multiplicities (std::nth_element, :43-45):
//   m[n] >= T  <=>  #{k-mers with multiplicity >= T} >= nk - n      -- no sorting needed.
#include <hip/hip_runtime.h>

#include <vector>

#include "bbk_internal.h"
#include "kmer_ops.h"

namespace bbk {

unsigned build_prefix_index(bbk_ctx *ctx, const uint64_t *keys, unsigned W, unsigned k, uint64_t n, DevBuf &prefix,
                            bool *wide);

// one wavefront per read, lanes over its k-mer positions
template <int W>
__global__ __launch_bounds__(256) void k_median_filter(const uint64_t *__restrict__ words, const uint64_t *__restrict__ woff,
                                                      const uint32_t *__restrict__ len, uint64_t n_reads, int k,
                                                      const Key<W> *__restrict__ keys, const uint32_t *__restrict__ counts,
                                                      PrefixTable P, uint32_t threshold,
                                                      uint8_t *__restrict__ keep) {
    const uint64_t r = (BBK_GID()) >> 6;
    if (r >= n_reads) return;
    const int lane = threadIdx.x & 63;
    const uint32_t L = len[r];
    if (L < (uint32_t)k) {  // CountMedianMlt returns 0 (:35-36)
        if (lane == 0) keep[r] = threshold == 0 ? 1 : 0;
        return;
    }
    const uint32_t nk = L - (uint32_t)k + 1u;
    const uint64_t *rw = words + woff[r];
    uint32_t ge = 0;
    for (uint32_t p = lane; p < nk; p += 64) {
        const Key<W> f = kmer_extract<W>(rw, p, k);
        const Key<W> rc = kmer_rc<W>(f, k);
        const Key<W> c = key_select<W>(!kmer_less_nucl<W>(rc, f), f, rc);
        const uint64_t i = table_find<W>(keys, P, c);
        const uint32_t m = i == kNotFound ? 0u : counts[i];
        ge += m >= threshold ? 1u : 0u;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) ge += __shfl_xor(ge, d, 64);
    if (lane == 0) keep[r] = ge >= nk - nk / 2 ? 1 : 0;
}

template <int W>
static void median_filter_impl(bbk_ctx *ctx, const bbk_reads *rd, const bbk_kmerset *s, uint32_t threshold, uint8_t *d_keep) {
    DevBuf prefix;
    bool wide = false;
    const unsigned bits = build_prefix_index(ctx, s->keys.as<uint64_t>(), s->W, s->k, s->n, prefix, &wide);
    const int w0bits = (W == 1) ? (int)(2 * s->k) : 64;
    const uint64_t threads = rd->n * 64;
    hipLaunchKernelGGL(k_median_filter<W>, bbk::grid_blocks((threads + 255) / 256), dim3(256), 0, ctx->stream, rd->d_words,
                       rd->d_woff, rd->d_len, rd->n, (int)s->k, s->keys.as<Key<W>>(), s->counts.as<uint32_t>(),
                       PrefixTable{prefix.p, w0bits - (int)bits, wide ? 1 : 0}, threshold, d_keep);
    check_launch("k_median_filter");
    BBK_HIP(hipStreamSynchronize(ctx->stream));
}

}  // namespace bbk

using namespace bbk;

extern "C" int bbk_reads_median_filter(bbk_ctx *ctx, const bbk_reads *reads, const bbk_kmerset *counts, unsigned threshold,
                                       uint8_t *h_keep, uint64_t *n_kept) {
    return guarded([&] {
        BBK_REQUIRE(ctx && reads && counts && (h_keep || reads->n == 0), BBK_ERR_ARG, "bbk_reads_median_filter: NULL argument");
        BBK_REQUIRE((counts->flags & BBK_CANONICAL) && counts->has_counts && counts->sorted && !counts->ref_order,
                    BBK_ERR_ARG,
                    "bbk_reads_median_filter: needs an ascending canonical k-mer set with counts "
                    "(bbk_count(BBK_CANONICAL | BBK_WITH_COUNTS))");
        BBK_HIP(hipSetDevice(ctx->device));
        if (n_kept) *n_kept = 0;
        if (reads->n == 0) return;
        DevBuf keep(reads->n + 16);
        if (counts->n == 0) {
            BBK_HIP(hipMemsetAsync(keep.p, threshold == 0 ? 1 : 0, reads->n, ctx->stream));
        } else {
            switch (counts->W) {
                case 1: median_filter_impl<1>(ctx, reads, counts, threshold, keep.as<uint8_t>()); break;
                case 2: median_filter_impl<2>(ctx, reads, counts, threshold, keep.as<uint8_t>()); break;
                case 3: median_filter_impl<3>(ctx, reads, counts, threshold, keep.as<uint8_t>()); break;
                case 4: median_filter_impl<4>(ctx, reads, counts, threshold, keep.as<uint8_t>()); break;
                default: BBK_REQUIRE(false, BBK_ERR_ARG, "unsupported key width %u", counts->W);
            }
        }
        BBK_HIP(hipMemcpyAsync(h_keep, keep.p, reads->n, hipMemcpyDeviceToHost, ctx->stream));
        BBK_HIP(hipStreamSynchronize(ctx->stream));
        if (n_kept) {
            uint64_t c = 0;
            for (uint64_t i = 0; i < reads->n; ++i) c += h_keep[i];
            *n_kept = c;
        }
    });
}
