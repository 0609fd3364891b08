// tipclip.hip -- early tip clipping on the extension index.
//
// Replaces EarlyTipClipperProcessor::ClipTips + RemoveInconsistentForwardLinks
// (common/assembly_graph/construction/early_simplification.hpp:20-35,37-160), the step the main SPAdes
// pipeline runs between the extension index and the unitig compaction.  Mask-only work:
//   for every stored k-mer and its reverse complement with >= 2 outgoing edges: walk every outgoing branch
//   (FindForward :107-118: unique-in/unique-out k-mers up to length_bound, ending in a dead end = a tip); tips
//   shorter than the junction's longest outgoing branch (a branch that is not a tip counts as infinitely long) are
//   isolated (mask := 0); afterwards the junctions that lost a tip drop their links to the isolated k-mers.
// The reference does this with OpenMP threads modifying the masks in place.  The k-mers of a tip are reachable from
// exactly one junction in one orientation (unique incoming edge), so isolating them cannot change what any other walk
// sees; here the marks are therefore collected first (k_tips_find, masks read-only), applied (k_tips_apply), and the
// links fixed against that snapshot (k_tips_links).  Parity is checked against the sequential CPU oracle.
#include <hip/hip_runtime.h>

#include <memory>

#include "bbk_internal.h"
#include "kmer_ops.h"

namespace bbk {

template <int W>
struct Oriented {
    Key<W> key;    // the k-mer as oriented
    uint64_t idx;  // table index of its canonical form
    bool minimal;  // key is the canonical form
};

struct TipTable {
    const void *keys;
    const uint8_t *masks;
    PrefixTable P;
    int k;
    uint64_t n;
};

template <int W>
__device__ inline uint64_t tt_find(const TipTable &T, const Key<W> &q) {
    return table_find<W>(reinterpret_cast<const Key<W> *>(T.keys), T.P, q);
}

// InvertableKeyWithHash (utils/ph_map/key_with_hash.hpp:108-207)
template <int W>
__device__ inline bool tt_orient(const TipTable &T, const Key<W> &key, Oriented<W> &o) {
    const Key<W> rc = kmer_rc<W>(key, T.k);
    o.key = key;
    o.minimal = !kmer_less_nucl<W>(rc, key);  // IsMinimal (rtseq.hpp:407-415)
    o.idx = tt_find<W>(T, key_select<W>(o.minimal, key, rc));
    return o.idx != kNotFound;
}

// InvertableStoring::get_value (storing_traits.hpp:30-68)
template <int W>
__device__ inline uint32_t tt_mask(const uint8_t *masks, const Oriented<W> &o) {
    const uint32_t m = masks[o.idx];
    return o.minimal ? m : rev8(m);
}

__device__ inline bool unique4(uint32_t nib) { return __builtin_popcount(nib & 15u) == 1; }

// FindForward (:107-118): size of the tip that starts at `start`, 0 if the branch is not a tip.  MARK: set the flag
// of every k-mer of the tip (only called for branches already known to be tips).
template <int W, bool MARK>
__device__ inline uint32_t tip_walk(const TipTable &T, Key<W> cur, uint32_t length_bound, uint8_t *flag) {
    uint32_t n = 0;
    Oriented<W> o;
    uint32_t m = 0;
    for (;;) {
        if (!tt_orient<W>(T, cur, o)) return 0;
        m = tt_mask<W>(T.masks, o);
        if (!(n < length_bound && unique4(m >> 4) && unique4(m))) break;
        if (MARK) flag[o.idx] = 1;
        ++n;
        cur = kmer_shl<W>(cur, T.k, (uint32_t)__builtin_ctz(m & 15u));
    }
    if (MARK) flag[o.idx] = 1;
    ++n;
    if (!unique4(m >> 4) || (m & 15u) != 0) return 0;  // branching or too long
    return n;
}

// one thread per (stored k-mer, orientation)
template <int W>
__global__ __launch_bounds__(256) void k_tips_find(TipTable T, uint32_t length_bound, uint8_t *__restrict__ flag,
                                                  uint8_t *__restrict__ tipped, unsigned long long *__restrict__ removed) {
    const uint64_t t = BBK_GID();
    if (t >= 2 * T.n) return;
    const uint64_t i = t >> 1;
    const bool rc_side = t & 1;
    const Key<W> canon = key_load<W>(&reinterpret_cast<const Key<W> *>(T.keys)[i]);
    const Key<W> key = key_select<W>(rc_side, kmer_rc<W>(canon, T.k), canon);
    Oriented<W> kh;
    if (!tt_orient<W>(T, key, kh)) return;
    const uint32_t mask = tt_mask<W>(T.masks, kh);
    if (__builtin_popcount(mask & 15u) < 2) return;
    // RemoveForward (:136-149)
    uint32_t len[4] = {0, 0, 0, 0};
    uint32_t mx = 0;
#pragma unroll
    for (uint32_t c = 0; c < 4; ++c) {
        if (!(mask & (1u << c))) continue;
        len[c] = tip_walk<W, false>(T, kmer_shl<W>(key, T.k, c), length_bound, nullptr);
        const uint32_t l = len[c] ? len[c] : 0xFFFFFFFFu;
        mx = l > mx ? l : mx;
    }
    uint32_t rm = 0;
#pragma unroll
    for (uint32_t c = 0; c < 4; ++c) {
        if (len[c] && len[c] < mx) {
            (void)tip_walk<W, true>(T, kmer_shl<W>(key, T.k, c), length_bound, flag);
            rm += len[c];
        }
    }
    if (rm) {
        tipped[t] = 1;
        atomicAdd(removed, (unsigned long long)rm);
    }
}

__global__ void k_tips_apply(const uint8_t *__restrict__ masks, const uint8_t *__restrict__ flag, uint64_t n,
                             uint8_t *__restrict__ out) {
    const uint64_t i = BBK_GID();
    if (i < n) out[i] = flag[i] ? (uint8_t)0 : masks[i];  // IsolateVertex
}

// RemoveInconsistentForwardLinks (:20-35) for both orientations of stored k-mer i; T.masks = the masks after
// k_tips_apply (read-only here), `out` the final masks
template <int W>
__global__ __launch_bounds__(256) void k_tips_links(TipTable T, const uint8_t *__restrict__ tipped,
                                                   uint8_t *__restrict__ out, unsigned long long *__restrict__ links) {
    const uint64_t i = BBK_GID();
    if (i >= T.n) return;
    uint32_t stored = T.masks[i];
    if (tipped[2 * i] | tipped[2 * i + 1]) {
        const Key<W> canon = key_load<W>(&reinterpret_cast<const Key<W> *>(T.keys)[i]);
        uint32_t cnt = 0;
        for (int side = 0; side < 2; ++side) {
            if (!tipped[2 * i + side]) continue;
            const Key<W> key = key_select<W>(side == 1, kmer_rc<W>(canon, T.k), canon);
            Oriented<W> kh;
            if (!tt_orient<W>(T, key, kh)) continue;
            const uint32_t mask = kh.minimal ? stored : rev8(stored);
            const uint32_t first = kmer_base<W>(key, 0);
            for (uint32_t c = 0; c < 4; ++c) {
                if (!(mask & (1u << c))) continue;
                Oriented<W> nx;
                if (!tt_orient<W>(T, kmer_shl<W>(key, T.k, c), nx)) continue;
                if (!(tt_mask<W>(T.masks, nx) & (1u << (4 + first)))) {
                    // DeleteOutgoing: bit c of the oriented mask = bit (as_is ? c : 7 - c) of the stored byte
                    stored &= ~(1u << (kh.minimal ? c : 7u - c));
                    ++cnt;
                }
            }
        }
        if (cnt) atomicAdd(links, (unsigned long long)cnt);
    }
    out[i] = (uint8_t)stored;
}

template <int W>
static void clip_tips_impl(bbk_ctx *ctx, bbk_extindex *x, uint32_t length_bound, uint64_t *removed, uint64_t *links) {
    const int w0bits = (W == 1) ? (int)(2 * x->k) : 64;
    DevBuf flag(x->n + 16), tipped(2 * x->n + 16), m2(x->n + 16), m3(x->n + 16), ctr(16);
    BBK_HIP(hipMemsetAsync(flag.p, 0, x->n + 16, ctx->stream));
    BBK_HIP(hipMemsetAsync(tipped.p, 0, 2 * x->n + 16, ctx->stream));
    BBK_HIP(hipMemsetAsync(ctr.p, 0, 16, ctx->stream));
    TipTable T{x->keys.p, x->masks.as<uint8_t>(), PrefixTable{x->prefix.p, w0bits - (int)x->prefix_bits, x->prefix_wide ? 1 : 0},
               (int)x->k, x->n};
    {
        KernelTimer t(ctx, "tip_find", 0.0);
        hipLaunchKernelGGL(k_tips_find<W>, bbk::grid_blocks((2 * x->n + 255) / 256), dim3(256), 0, ctx->stream, T,
                           length_bound, flag.as<uint8_t>(), tipped.as<uint8_t>(), ctr.as<unsigned long long>());
        check_launch("k_tips_find");
    }
    hipLaunchKernelGGL(k_tips_apply, bbk::grid_blocks((x->n + 255) / 256), dim3(256), 0, ctx->stream,
                       x->masks.as<uint8_t>(), flag.as<uint8_t>(), x->n, m2.as<uint8_t>());
    check_launch("k_tips_apply");
    T.masks = m2.as<uint8_t>();
    hipLaunchKernelGGL(k_tips_links<W>, bbk::grid_blocks((x->n + 255) / 256), dim3(256), 0, ctx->stream, T,
                       tipped.as<uint8_t>(), m3.as<uint8_t>(), ctr.as<unsigned long long>() + 1);
    check_launch("k_tips_links");
    unsigned long long h[2] = {0, 0};
    BBK_HIP(hipMemcpyAsync(h, ctr.p, 16, hipMemcpyDeviceToHost, ctx->stream));
    BBK_HIP(hipStreamSynchronize(ctx->stream));
    x->masks = std::move(m3);
    if (removed) *removed = h[0];
    if (links) *links = h[1];
}

}  // namespace bbk

using namespace bbk;

extern "C" int bbk_extindex_clip_tips(bbk_ctx *ctx, bbk_extindex *x, uint32_t length_bound, uint64_t *removed_kmers,
                                      uint64_t *removed_links) {
    return guarded([&] {
        BBK_REQUIRE(ctx && x, BBK_ERR_ARG, "bbk_extindex_clip_tips: NULL argument");
        BBK_REQUIRE(x->n < (1ull << 37), BBK_ERR_ARG, "bbk_extindex_clip_tips: %llu k-mers exceed the launch grid",
                    (unsigned long long)x->n);
        BBK_HIP(hipSetDevice(ctx->device));
        if (removed_kmers) *removed_kmers = 0;
        if (removed_links) *removed_links = 0;
        if (x->n == 0) return;
        switch (x->W) {
            case 1: clip_tips_impl<1>(ctx, x, length_bound, removed_kmers, removed_links); break;
            case 2: clip_tips_impl<2>(ctx, x, length_bound, removed_kmers, removed_links); break;
            case 3: clip_tips_impl<3>(ctx, x, length_bound, removed_kmers, removed_links); break;
            case 4: clip_tips_impl<4>(ctx, x, length_bound, removed_kmers, removed_links); break;
            default: BBK_REQUIRE(false, BBK_ERR_ARG, "unsupported key width %u", x->W);
        }
    });
}
