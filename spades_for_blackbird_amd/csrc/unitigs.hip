// unitigs.hip -- unbranching-path extraction, graph linking and the GFA/FASTA writers.
//
// Replaces (reference common/assembly_graph/construction/debruijn_graph_constructor.hpp):
//   UnbranchingPathExtractor::AddStartDeEdges / StepRightIfPossible / ConstructSequenceWithEdge /
//     CalculateSequences (:201-245,267-286)  -> k_count_starts / k_fill_starts / k_walk (one thread
//     per start edge; the sorted (k-mer, mask) table + prefix table replaces the MPHF)
//   `if (s < !s) continue` (:279)              -> decided in-flight from the two end k-mers (see k_walk)
//   CleanCondensed (:288-304)                  -> a visited byte per canonical k-mer
//   CollectLoops / ConstructLoopFromVertex / SplitLoop (:248-265,308-344) -> the (rare) leftover
//     non-junction k-mers are compacted on the device and walked on the host, sequentially like
//     the reference
//   FastGraphFromSequencesConstructor (:390-518): LinkRecord (:400-430) keyed by the canonical end
//     k-mer's table index, device radix sort, vertices = distinct keys, links = incoming x outgoing
//   io/graph/gfa_writer.cpp:18-52 and projects/gbuilder/main.cpp:183-192 for the text formats.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <omp.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "bbk_internal.h"
#include "kmer_ops.h"

struct bbk_unitigs {
    unsigned k = 0;
    uint64_t n = 0, n_loops = 0, n_vertices = 0, n_links = 0;
    bbk::raw_vector<char> bases;        // concatenated ACGT
    bbk::raw_vector<uint64_t> offsets;  // n + 1
    bbk::raw_vector<uint64_t> links;    // 2 per link: (from << 1 | from_plus), (to << 1 | to_plus)
    bool has_cov = false;
    std::vector<uint64_t> kc;       // per unitig: sum of (k+1)-mer multiplicities (KC:i:)
    // Device-resident result (no perfect loops): the GFA text is formatted on the device and streamed
    // to the file; the host arrays above are filled lazily, only for the calls that need them.
    uint64_t total_bases = 0;
    bool host_valid = true;
    bbk::DevBuf d_bases, d_uoff, d_links;
};

namespace bbk {

__device__ __host__ inline bool mask_is_junction(uint32_t m) {
    // InOutMask::IsJunction (kmer_extension_index.hpp:46-58,144-162)
    return __builtin_popcount(m & 15u) != 1 || __builtin_popcount((m >> 4) & 15u) != 1;
}

__global__ void k_count_starts(const uint8_t *__restrict__ masks, uint64_t n, uint64_t *__restrict__ cnt) {
    const uint64_t i = BBK_GID();
    if (i >= n) return;
    const uint32_t m = masks[i];
    // outgoing edges of the k-mer and of its reverse complement (whose outgoing = this one's incoming)
    cnt[i] = mask_is_junction(m) ? (uint64_t)__builtin_popcount(m) : 0ull;
}

// start edge descriptor: idx << 3 | strand << 2 | base   (AddStartDeEdges :214-226: the k-mer's own
// outgoing edges for next = 0..3, then those of its reverse complement)
__global__ void k_fill_starts(const uint8_t *__restrict__ masks, uint64_t n, const uint64_t *__restrict__ off,
                              uint64_t *__restrict__ starts) {
    const uint64_t i = BBK_GID();
    if (i >= n) return;
    const uint32_t m = masks[i];
    if (!mask_is_junction(m)) return;
    uint64_t o = off[i];
    for (uint32_t c = 0; c < 4; ++c)
        if (m & (1u << c)) starts[o++] = (i << 3) | c;
    const uint32_t mr = rev8(m);
    for (uint32_t c = 0; c < 4; ++c)
        if (mr & (1u << c)) starts[o++] = (i << 3) | 4u | c;
}

struct WalkOut {
    // pass 0
    uint64_t *keep;      // [E] 0/1
    uint64_t *ulen;      // [E] k + appended bases if kept else 0
    // pass 1
    const uint64_t *uid;   // exclusive scan of keep
    const uint64_t *boff;  // exclusive scan of ulen
    char *bases;
    uint64_t *uoff;        // [U] base offset of unitig
    uint64_t *rec;         // [2U] link records: (idx<<2 | is_rc<<1 | is_start), ~0 = none
    uint8_t *selfconj;     // [U] the unitig equals its own reverse complement
    uint8_t *visited;      // [n]
    uint32_t *err;
};

// One thread per start edge.  PASS 0: length + keep decision (+ visited marks).  PASS 1: write the
// bases and the link records of kept unitigs.
//
// Keep rule (`if (s < !s) continue`, :279): s = x.c....z ; rc(s) starts with rc(z).  If
// x != rc(z) the first k bases decide.  If x == rc(z) the next base decides: s[k] = c against
// rc(s)[k] = comp(base preceding z in s); if these agree too, s and rc(s) leave the same junction
// by the same edge, the walk is deterministic, hence s == rc(s) (self-conjugate edge): keep.
template <int W, int PASS>
__global__ __launch_bounds__(256) void k_walk(const Key<W> *__restrict__ keys, const uint8_t *__restrict__ masks,
                                             PrefixTable P, uint64_t n, int k,
                                             const uint64_t *__restrict__ starts, uint64_t E, WalkOut o) {
    const uint64_t e = BBK_GID();
    if (e >= E) return;
    if (PASS == 1 && o.keep[e] == 0) return;
    const uint64_t d = starts[e];
    const uint64_t i = d >> 3;
    const bool s_rc = (d >> 2) & 1;
    const uint32_t c0 = (uint32_t)(d & 3);
    const Key<W> canon0 = key_load<W>(&keys[i]);
    const Key<W> x = key_select<W>(s_rc, kmer_rc<W>(canon0, k), canon0);
    // pass 1 writes the unitig's bases: 8 ASCII characters are collected in a register and leave as one 8-byte store
    // (global memory takes unaligned 8-byte stores; one store per base was 8x the store instructions of this
    // latency-bound walk)
    typedef uint64_t __attribute__((aligned(1))) u64_any;
    char *dst = nullptr;
    uint64_t pend = 0;   // characters not yet stored
    uint32_t npend = 0;  // how many (0..7)
    uint64_t wpos = 0;   // characters stored so far
    auto put = [&](uint32_t base) {
        pend |= (uint64_t)(unsigned char)"ACGT"[base] << (8 * npend);
        if (++npend == 8) {
            *reinterpret_cast<u64_any *>(dst + wpos) = pend;
            wpos += 8;
            pend = 0;
            npend = 0;
        }
    };
    if (PASS == 1) {
        dst = o.bases + o.boff[e];
        for (int j = 0; j < k; ++j) put(kmer_base<W>(x, j));
        put(c0);
    }
    Key<W> cur = kmer_shl<W>(x, k, c0);
    uint64_t len = 1;
    uint32_t prev_first = kmer_base<W>(x, 0);
    uint64_t j = kNotFound;
    bool cur_min = true;
    Key<W> rcur = cur;
    for (;;) {
        rcur = kmer_rc<W>(cur, k);
        cur_min = !kmer_less_nucl<W>(rcur, cur);
        const Key<W> q = key_select<W>(cur_min, cur, rcur);
        j = table_find<W>(keys, P, q);
        if (j == kNotFound) {
            atomicOr(o.err, 1u);
            return;
        }
        uint32_t m = masks[j];
        if (!cur_min) m = rev8(m);
        if (mask_is_junction(m)) break;
        if (PASS == 0) o.visited[j] = 1;
        const uint32_t c = (uint32_t)__builtin_ctz(m & 15u);
        prev_first = kmer_base<W>(cur, 0);
        cur = kmer_shl<W>(cur, k, c);
        if (PASS == 1) put(c);
        ++len;
        if (len > n + 2) {  // cannot happen on a consistent index: a start edge never re-enters itself
            atomicOr(o.err, 2u);
            return;
        }
    }
    if (PASS == 0) {
        bool keep;
        if (key_eq<W>(x, rcur)) {
            const uint32_t other = 3u - prev_first;
            keep = c0 >= other;
        } else {
            keep = kmer_less_nucl<W>(rcur, x);  // rc(s) < s
        }
        o.keep[e] = keep ? 1ull : 0ull;
        o.ulen[e] = keep ? (uint64_t)k + len : 0ull;
    } else {
        for (uint32_t j = 0; j < npend; ++j) dst[wpos + j] = (char)(pend >> (8 * j));  // the last 0..7 characters
        const uint64_t u = o.uid[e];
        o.uoff[u] = o.boff[e];
        const bool selfconj = key_eq<W>(x, rcur) && c0 == 3u - prev_first;
        // StartLink / EndLink (:432-448): canonical form of the end k-mers, is_rc = k-mer is not it
        o.rec[2 * u] = (i << 2) | ((uint64_t)(s_rc ? 1 : 0) << 1) | 1ull;
        o.rec[2 * u + 1] = selfconj ? ~0ull : ((j << 2) | ((uint64_t)(cur_min ? 0 : 1) << 1));
        o.selfconj[u] = selfconj ? 1 : 0;
    }
}

__global__ void k_loop_candidates(const uint8_t *__restrict__ masks, const uint8_t *__restrict__ visited, uint64_t n,
                                  uint64_t *__restrict__ flag) {
    const uint64_t i = BBK_GID();
    if (i >= n) return;
    flag[i] = (!mask_is_junction(masks[i]) && !visited[i]) ? 1ull : 0ull;
}

__global__ void k_compact_candidates(const uint64_t *__restrict__ flag_scan, const uint8_t *__restrict__ masks,
                                     const uint8_t *__restrict__ visited, uint64_t n, uint64_t *__restrict__ idx) {
    const uint64_t i = BBK_GID();
    if (i >= n) return;
    if (!mask_is_junction(masks[i]) && !visited[i]) idx[flag_scan[i]] = i;
}

// unitigs -> packed reads (bbk_unitigs_to_reads): words per unitig, then one wavefront per unitig packs 32 bases per lane
__global__ void k_unitig_words(const uint64_t *__restrict__ uoff, uint64_t nu, uint64_t *__restrict__ nw,
                               uint32_t *__restrict__ len, uint32_t *__restrict__ err) {
    const uint64_t i = BBK_GID();
    if (i >= nu) return;
    const uint64_t l = uoff[i + 1] - uoff[i];
    if (l > 0xFFFFFFFFull) atomicOr(err, 1u);
    len[i] = (uint32_t)l;
    nw[i] = (l + 31) >> 5;
}
__global__ __launch_bounds__(256) void k_pack_unitigs(const char *__restrict__ bases, const uint64_t *__restrict__ uoff,
                                                     const uint64_t *__restrict__ woff, uint64_t nu,
                                                     uint64_t *__restrict__ words) {
    const uint64_t i = (BBK_GID()) >> 6;
    if (i >= nu) return;
    const int lane = threadIdx.x & 63;
    const uint64_t b0 = uoff[i], len = uoff[i + 1] - b0, nw = (len + 31) >> 5;
    uint64_t *dst = words + woff[i];
    for (uint64_t w = lane; w < nw; w += 64) {
        const uint64_t lo = w << 5, hi = lo + 32 < len ? lo + 32 : len;
        uint64_t v = 0;
        for (uint64_t j = lo; j < hi; ++j) {
            const char c = bases[b0 + j];
            const uint64_t code = c == 'A' ? 0ull : c == 'C' ? 1ull : c == 'G' ? 2ull : 3ull;
            v |= code << ((j - lo) << 1);
        }
        dst[w] = v;
    }
}

__global__ void k_gather_candidates(const uint64_t *__restrict__ keys, const uint8_t *__restrict__ masks,
                                    const uint64_t *__restrict__ idx, uint64_t nc, int W, uint64_t *__restrict__ out_keys,
                                    uint8_t *__restrict__ out_masks) {
    const uint64_t c = BBK_GID();
    if (c >= nc) return;
    const uint64_t r = idx[c];
    for (int w = 0; w < W; ++w) out_keys[c * W + w] = keys[r * W + w];
    out_masks[c] = masks[r];
}

// XXH3 bucket (of nb) of every gathered candidate k-mer: the reference walks its k-mer file bucket by bucket
// (KMerSegmentPolicy, utils/kmer_mph/kmer_buckets.hpp:28-33; 10 x threads buckets, kmer_extension_index_builder.hpp:73)
template <int W>
__global__ void k_candidate_buckets(const uint64_t *__restrict__ keys, uint64_t nc, uint64_t nb, uint32_t *__restrict__ out) {
    const uint64_t c = BBK_GID();
    if (c >= nc) return;
    Key<W> q;
#pragma unroll
    for (int w = 0; w < W; ++w) q.w[w] = keys[c * W + w];
    out[c] = (uint32_t)__umul64hi(xxh3_64<W>(q), nb);
}

// Links from the sorted link records (vertices = groups of equal canonical k-mer index): for every
// canonical vertex every (incoming, outgoing) pair (GFAWriter::WriteLinks, io/graph/gfa_writer.cpp:43-52,
// over the edge lists ConstructionHelper::LinkIncomingEdge/LinkOutgoingEdge build,
// assembly_graph/core/construction_helper.hpp:80-90).  The group head does the work of its group.
__device__ inline bool rec_incoming(uint64_t key) {
    const bool st = key & 1, rc = (key >> 1) & 1;
    return (!st && !rc) || (st && rc);
}
__device__ inline bool rec_outgoing(uint64_t key) {
    const bool st = key & 1, rc = (key >> 1) & 1;
    return (st && !rc) || (!st && rc);
}

template <bool WRITE>
__global__ __launch_bounds__(256) void k_links(const uint64_t *__restrict__ key, const uint32_t *__restrict__ edge,
                                              uint64_t nrec, const uint8_t *__restrict__ selfconj,
                                              uint64_t *__restrict__ cnt, const uint64_t *__restrict__ off,
                                              uint64_t *__restrict__ links, unsigned long long *__restrict__ nvert) {
    const uint64_t r = BBK_GID();
    if (r >= nrec) return;
    const uint64_t kr = key[r];
    const bool head = kr != ~0ull && (r == 0 || (key[r - 1] >> 2) != (kr >> 2));
    if (!head) {
        if (!WRITE) cnt[r] = 0;
        return;
    }
    uint64_t e = r;
    uint32_t nin = 0, nout = 0;
    while (e < nrec && key[e] != ~0ull && (key[e] >> 2) == (kr >> 2)) {
        nin += rec_incoming(key[e]) ? 1u : 0u;
        nout += rec_outgoing(key[e]) ? 1u : 0u;
        ++e;
    }
    if (!WRITE) {
        cnt[r] = (uint64_t)nin * nout;
        atomicAdd(nvert, 1ull);
        return;
    }
    uint64_t o = off[r];
    for (uint64_t a = r; a < e; ++a) {
        const uint64_t ka = key[a];
        if (!rec_incoming(ka)) continue;
        const uint32_t ea = edge[a];
        const uint32_t oa = (!(ka & 1) || selfconj[ea]) ? 1u : 0u;
        for (uint64_t b = r; b < e; ++b) {
            const uint64_t kb = key[b];
            if (!rec_outgoing(kb)) continue;
            const uint32_t eb = edge[b];
            const uint32_t ob = ((kb & 1) || selfconj[eb]) ? 1u : 0u;
            links[2 * o] = ((uint64_t)ea << 1) | oa;
            links[2 * o + 1] = ((uint64_t)eb << 1) | ob;
            ++o;
        }
    }
}

__global__ void k_edge_ids(uint32_t *__restrict__ ids, uint64_t n2) {
    const uint64_t i = BBK_GID();
    if (i < n2) ids[i] = (uint32_t)(i >> 1);
}

// ---- host helpers for the loop path (rare; plain strings) -----------------------------------
static std::string str_rc(const std::string &s) {
    std::string r(s.rbegin(), s.rend());
    for (char &c : r) c = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A';
    return r;
}

static void pack_kmer(const char *s, int k, uint64_t *w, int W) {
    for (int i = 0; i < W; ++i) w[i] = 0;
    for (int i = 0; i < k; ++i) {
        const uint64_t c = s[i] == 'A' ? 0 : s[i] == 'C' ? 1 : s[i] == 'G' ? 2 : 3;
        w[i >> 5] |= c << ((i & 31) << 1);
    }
}

static std::string unpack_kmer(const uint64_t *w, int k) {
    std::string s((size_t)k, 'A');
    for (int i = 0; i < k; ++i) s[(size_t)i] = "ACGT"[(w[i >> 5] >> ((i & 31) << 1)) & 3];
    return s;
}

struct LoopTable {
    int k, W;
    std::vector<uint64_t> idx;     // index in the full table
    std::vector<uint64_t> keys;    // W words each, ascending
    std::vector<uint8_t> masks;
    std::vector<uint8_t> used;
    // position of an oriented k-mer's canonical form, -1 if absent
    long find(const std::string &kmer, bool *minimal) const {
        const std::string r = str_rc(kmer);
        *minimal = kmer <= r;
        const std::string &c = *minimal ? kmer : r;
        uint64_t q[4];
        pack_kmer(c.data(), k, q, W);
        size_t lo = 0, hi = idx.size();
        while (lo < hi) {
            const size_t mid = lo + (hi - lo) / 2;
            int cmp = 0;
            for (int i = 0; i < W && cmp == 0; ++i)
                cmp = keys[mid * W + i] < q[i] ? -1 : (keys[mid * W + i] > q[i] ? 1 : 0);
            if (cmp == 0) return (long)mid;
            if (cmp < 0) lo = mid + 1;
            else hi = mid;
        }
        return -1;
    }
};

template <int W>
static void run_walk(bbk_ctx *ctx, int pass, const bbk_extindex *x, const uint64_t *starts, uint64_t E, WalkOut o) {
    if (E == 0) return;
    const int w0bits = (x->W == 1) ? (int)(2 * x->k) : 64;
    const PrefixTable P{x->prefix.p, w0bits - (int)x->prefix_bits, x->prefix_wide ? 1 : 0};
    // bytes: every non-junction k-mer is stepped over once per orientation; a step is one lookup = 2 prefix-table
    // entries + ~3 key probes + 1 mask byte (latency-bound pointer chase: the figure is for reading the rate, not a
    // roofline claim); pass 1 also writes the bases
    KernelTimer t(ctx, pass == 0 ? "walk0" : "walk1", 2.0 * (double)x->n * (3.0 * x->W * 8 + 8 + 1));
    if (pass == 0)
        hipLaunchKernelGGL((k_walk<W, 0>), bbk::grid_blocks((E + 255) / 256), dim3(256), 0, ctx->stream,
                           x->keys.as<Key<W>>(), x->masks.as<uint8_t>(), P, x->n, (int)x->k, starts, E, o);
    else
        hipLaunchKernelGGL((k_walk<W, 1>), bbk::grid_blocks((E + 255) / 256), dim3(256), 0, ctx->stream,
                           x->keys.as<Key<W>>(), x->masks.as<uint8_t>(), P, x->n, (int)x->k, starts, E, o);
    check_launch("k_walk");
}

static void dispatch_walk(bbk_ctx *ctx, int pass, const bbk_extindex *x, const uint64_t *starts, uint64_t E,
                          WalkOut o) {
    switch (x->W) {
        case 1: run_walk<1>(ctx, pass, x, starts, E, o); break;
        case 2: run_walk<2>(ctx, pass, x, starts, E, o); break;
        case 3: run_walk<3>(ctx, pass, x, starts, E, o); break;
        case 4: run_walk<4>(ctx, pass, x, starts, E, o); break;
        default: BBK_REQUIRE(false, BBK_ERR_ARG, "unsupported key width %u", x->W);
    }
}

struct LinkRec {
    uint64_t key;
    uint32_t edge;
};

// device -> host on the context's stream (a plain hipMemcpy runs on the null stream and would not
// wait for kernels queued on a non-blocking stream)
static void d2h(bbk_ctx *ctx, void *dst, const void *src, size_t bytes) {
    BBK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    BBK_HIP(hipStreamSynchronize(ctx->stream));
}

// fills the host arrays of a device-resident result
static void ensure_host(bbk_ctx *ctx, const bbk_unitigs *cu) {
    bbk_unitigs *u = const_cast<bbk_unitigs *>(cu);
    if (u->host_valid) return;
    BBK_HIP(hipSetDevice(ctx->device));
    u->bases.resize(u->total_bases);
    u->offsets.resize(u->n + 1);
    u->links.resize(2 * u->n_links);
    if (u->total_bases) d2h_big(ctx, u->bases.data(), u->d_bases.p, u->total_bases);
    d2h_big(ctx, u->offsets.data(), u->d_uoff.p, (u->n + 1) * 8);
    if (u->n_links) d2h_big(ctx, u->links.data(), u->d_links.p, u->n_links * 16);
    u->host_valid = true;
}

// ---- GFA text on the device ------------------------------------------------------------------
__device__ inline uint32_t dev_dec_len(uint64_t v) {
    uint32_t n = 1;
    while (v >= 10) {
        v /= 10;
        ++n;
    }
    return n;
}
__device__ inline void dev_put_dec(char *dst, uint64_t v, uint32_t len) {
    for (uint32_t i = 0; i < len; ++i) {
        dst[len - 1 - i] = (char)('0' + v % 10);
        v /= 10;
    }
}
constexpr uint32_t kGfaTail = 15;  // "\tDP:f:0\tKC:i:0\n"

__global__ void k_gfa_s_len(const uint64_t *__restrict__ uoff, uint64_t nu, uint64_t *__restrict__ len) {
    const uint64_t i = BBK_GID();
    if (i < nu) len[i] = 2 + dev_dec_len(3 + 2 * i) + 1 + (uoff[i + 1] - uoff[i]) + kGfaTail;
}

// one wavefront per segment line: "S\t<3+2i>\t<bases>\tDP:f:0\tKC:i:0\n"
__global__ __launch_bounds__(256) void k_gfa_s_write(const char *__restrict__ bases, const uint64_t *__restrict__ uoff,
                                                    const uint64_t *__restrict__ pos, uint64_t nu,
                                                    char *__restrict__ out) {
    const uint64_t i = (BBK_GID()) >> 6;
    if (i >= nu) return;
    const int lane = threadIdx.x & 63;
    char *d = out + pos[i];
    const uint64_t id = 3 + 2 * i;
    const uint32_t idl = dev_dec_len(id);
    if (lane == 0) {
        d[0] = 'S';
        d[1] = '\t';
        dev_put_dec(d + 2, id, idl);
        d[2 + idl] = '\t';
    }
    const uint64_t b0 = uoff[i], len = uoff[i + 1] - b0;
    char *sq = d + 3 + idl;
    for (uint64_t j = lane; j < len; j += 64) sq[j] = bases[b0 + j];
    if (lane < (int)kGfaTail) sq[len + lane] = "\tDP:f:0\tKC:i:0\n"[lane];
}

__global__ void k_gfa_l_len(const uint64_t *__restrict__ links, uint64_t nl, uint32_t klen, uint64_t *__restrict__ len) {
    const uint64_t l = BBK_GID();
    if (l < nl)
        len[l] = 2 + dev_dec_len(3 + 2 * (links[2 * l] >> 1)) + 3 + dev_dec_len(3 + 2 * (links[2 * l + 1] >> 1)) + 3 + klen + 2;
}

// "L\t<e1>\t<+|->\t<e2>\t<+|->\t<k>M\n"
__global__ void k_gfa_l_write(const uint64_t *__restrict__ links, const uint64_t *__restrict__ pos, uint64_t nl,
                              uint32_t k, uint32_t klen, char *__restrict__ out) {
    const uint64_t l = BBK_GID();
    if (l >= nl) return;
    char *d = out + pos[l];
    const uint64_t a = links[2 * l], b = links[2 * l + 1];
    const uint64_t ia = 3 + 2 * (a >> 1), ib = 3 + 2 * (b >> 1);
    const uint32_t la = dev_dec_len(ia), lb = dev_dec_len(ib);
    *d++ = 'L';
    *d++ = '\t';
    dev_put_dec(d, ia, la);
    d += la;
    *d++ = '\t';
    *d++ = (a & 1ull) ? '+' : '-';
    *d++ = '\t';
    dev_put_dec(d, ib, lb);
    d += lb;
    *d++ = '\t';
    *d++ = (b & 1u) ? '+' : '-';
    *d++ = '\t';
    dev_put_dec(d, k, klen);
    d += klen;
    *d++ = 'M';
    *d++ = '\n';
}

static void build(bbk_ctx *ctx, bbk_extindex *x, bbk_unitigs &U, unsigned ref_threads) {
    const int k = (int)x->k;
    const uint64_t n = x->n;
    U.k = x->k;
    BBK_REQUIRE(k % 2 == 1, BBK_ERR_ARG, "k-mer size must be odd");  // projects/gbuilder/main.cpp:125-126
    // k-mer indices, start edges and link-record keys are 64-bit (KMerIndex::seq_idx is a size_t, kmer_index.hpp:85-90;
    // LinkRecord keys are 64-bit, debruijn_graph_constructor.hpp:400-430); only the launch grid bounds n
    BBK_REQUIRE(n < (1ull << 37), BBK_ERR_ARG, "extension index of %llu k-mers exceeds the launch grid", (unsigned long long)n);
    U.offsets.assign(1, 0);
    if (n == 0) return;
    U.host_valid = true;

    // ---- start edges
    DevBuf cnt((n + 1) * 8);
    hipLaunchKernelGGL(k_count_starts, bbk::grid_blocks((n + 255) / 256), dim3(256), 0, ctx->stream,
                       x->masks.as<uint8_t>(), n, cnt.as<uint64_t>());
    check_launch("k_count_starts");
    const uint64_t E = exclusive_scan_u64(ctx, cnt.as<uint64_t>(), cnt.as<uint64_t>(), n);
    BBK_REQUIRE(E <= 8 * n, BBK_ERR_INTERNAL, "unitigs: %llu start edges counted for %llu k-mers", (unsigned long long)E,
                (unsigned long long)n);
    DevBuf starts((E + 1) * 8);
    hipLaunchKernelGGL(k_fill_starts, bbk::grid_blocks((n + 255) / 256), dim3(256), 0, ctx->stream,
                       x->masks.as<uint8_t>(), n, cnt.as<uint64_t>(), starts.as<uint64_t>());
    check_launch("k_fill_starts");
    cnt.release();

    // ---- pass 0: lengths + keep
    DevBuf keep((E + 1) * 8), ulen((E + 1) * 8), visited(n + 16), err(16);
    BBK_HIP(hipMemsetAsync(visited.p, 0, n + 16, ctx->stream));
    BBK_HIP(hipMemsetAsync(err.p, 0, 16, ctx->stream));
    WalkOut o{};
    o.keep = keep.as<uint64_t>();
    o.ulen = ulen.as<uint64_t>();
    o.visited = visited.as<uint8_t>();
    o.err = err.as<uint32_t>();
    dispatch_walk(ctx, 0, x, starts.as<uint64_t>(), E, o);
    uint32_t herr = 0;
    d2h(ctx, &herr, err.p, 4);  // before the scans: a walk that gave up leaves its keep / length entries unwritten
    BBK_REQUIRE(herr == 0, BBK_ERR_INTERNAL, "unitig walk failed (code %u): extension index is inconsistent", herr);
    DevBuf uid((E + 1) * 8), boff((E + 1) * 8);
    const uint64_t NU = exclusive_scan_u64(ctx, keep.as<uint64_t>(), uid.as<uint64_t>(), E);
    const uint64_t NB = exclusive_scan_u64(ctx, ulen.as<uint64_t>(), boff.as<uint64_t>(), E);
    BBK_REQUIRE(NU <= E && NB <= 2 * n + (uint64_t)(k + 1) * NU, BBK_ERR_INTERNAL,
                "unitig walk: %llu unitigs / %llu bases from %llu start edges, %llu k-mers", (unsigned long long)NU,
                (unsigned long long)NB, (unsigned long long)E, (unsigned long long)n);
    // edge ids travel as the u32 payload of the link-record sort
    BBK_REQUIRE(NU < (1ull << 32) - 1, BBK_ERR_ARG, "%llu unitigs: edge ids are 32-bit", (unsigned long long)NU);

    // ---- pass 1: bases + link records
    DevBuf bases(NB + 16), uoff((NU + 1) * 8), rec((2 * NU + 2) * 8), selfc(NU + 16);
    o.selfconj = selfc.as<uint8_t>();
    o.uid = uid.as<uint64_t>();
    o.boff = boff.as<uint64_t>();
    o.bases = bases.as<char>();
    o.uoff = uoff.as<uint64_t>();
    o.rec = rec.as<uint64_t>();
    dispatch_walk(ctx, 1, x, starts.as<uint64_t>(), E, o);
    d2h(ctx, &herr, err.p, 4);
    BBK_REQUIRE(herr == 0, BBK_ERR_INTERNAL, "unitig walk (pass 1) failed (code %u)", herr);

    U.total_bases = NB;
    BBK_HIP(hipMemcpyAsync(uoff.as<uint64_t>() + NU, &NB, 8, hipMemcpyHostToDevice, ctx->stream));
    starts.release();
    keep.release();
    ulen.release();
    uid.release();
    boff.release();

    // ---- loop candidates first: without perfect loops (the common case) links are made on the device
    DevBuf flag((n + 1) * 8);
    hipLaunchKernelGGL(k_loop_candidates, bbk::grid_blocks((n + 255) / 256), dim3(256), 0, ctx->stream,
                       x->masks.as<uint8_t>(), visited.as<uint8_t>(), n, flag.as<uint64_t>());
    check_launch("k_loop_candidates");
    const uint64_t NC = exclusive_scan_u64(ctx, flag.as<uint64_t>(), flag.as<uint64_t>(), n);
    if (NC != 0) {  // perfect loops are appended on the host: bring the paths over now
        U.bases.resize(NB);
        U.offsets.resize(NU + 1);
        if (NB) d2h_big(ctx, U.bases.data(), bases.p, NB);
        d2h_big(ctx, U.offsets.data(), uoff.p, (NU + 1) * 8);
        U.host_valid = true;
    }

    // ---- link records: sort by (key, edge) on the device
    std::vector<LinkRec> recs;
    if (NU) {
        DevBuf ids(2 * NU * 4 + 16), rtmp((2 * NU + 2) * 8), itmp(2 * NU * 4 + 16);
        hipLaunchKernelGGL(k_edge_ids, bbk::grid_blocks((2 * NU + 255) / 256), dim3(256), 0, ctx->stream,
                           ids.as<uint32_t>(), 2 * NU);
        check_launch("k_edge_ids");
        int bits = 2;
        while ((1ull << (bits - 2)) < n + 1) ++bits;
        std::vector<PassDesc> passes;
        // keys are either < 2^bits or ~0 (no record): the low `bits` bits order the real records; one
        // final pass on the top byte (0x00 vs 0xFF) moves the "no record" entries to the end
        for (int sft = 0; sft < bits; sft += 8) passes.push_back({0, 0, sft, std::min(8, bits - sft), 0});
        passes.push_back({0, 0, 56, 8, 0});
        sort_records(ctx, 1, rec.p, rtmp.p, ids.as<uint32_t>(), itmp.as<uint32_t>(), 2 * NU, passes);
        if (NC == 0) {
            KernelTimer t(ctx, "links", 0);
            DevBuf lcnt((2 * NU + 1) * 8), nv(16);
            BBK_HIP(hipMemsetAsync(nv.p, 0, 16, ctx->stream));
            hipLaunchKernelGGL((k_links<false>), bbk::grid_blocks((2 * NU + 255) / 256), dim3(256), 0, ctx->stream,
                               rec.as<uint64_t>(), ids.as<uint32_t>(), 2 * NU, selfc.as<uint8_t>(), lcnt.as<uint64_t>(),
                               (const uint64_t *)nullptr, (uint64_t *)nullptr, nv.as<unsigned long long>());
            check_launch("k_links<count>");
            const uint64_t NL = exclusive_scan_u64(ctx, lcnt.as<uint64_t>(), lcnt.as<uint64_t>(), 2 * NU);
            DevBuf dl(NL * 16 + 16);
            hipLaunchKernelGGL((k_links<true>), bbk::grid_blocks((2 * NU + 255) / 256), dim3(256), 0, ctx->stream,
                               rec.as<uint64_t>(), ids.as<uint32_t>(), 2 * NU, selfc.as<uint8_t>(), (uint64_t *)nullptr,
                               lcnt.as<uint64_t>(), dl.as<uint64_t>(), (unsigned long long *)nullptr);
            check_launch("k_links<write>");
            unsigned long long hv = 0;
            d2h(ctx, &hv, nv.p, 8);
            U.n = NU;
            U.n_loops = 0;
            U.n_vertices = hv;
            U.n_links = NL;
            // the result stays on the device; host copies are made on demand (ensure_host)
            U.host_valid = false;
            U.d_bases = std::move(bases);
            U.d_uoff = std::move(uoff);
            U.d_links = std::move(dl);
            return;
        }
        raw_vector<uint64_t> hk(2 * NU);
        raw_vector<uint32_t> he(2 * NU);
        d2h_big(ctx, hk.data(), rec.p, 2 * NU * 8);
        d2h_big(ctx, he.data(), ids.p, 2 * NU * 4);
        recs.reserve(2 * NU);
        for (uint64_t r = 0; r < 2 * NU; ++r) {
            if (hk[r] == ~0ull) break;
            recs.push_back({hk[r], he[r]});
        }
    }
    rec.release();
    uoff.release();
    bases.release();
    U.total_bases = 0;  // host arrays are authoritative from here on (loops get appended)

    // ---- perfect loops: leftover non-junction k-mers (CollectLoops :308-344), walked on the host
    uint64_t n_paths = NU, n_loops = 0;
    if (NC) {
        LoopTable T;
        T.k = k;
        T.W = (int)x->W;
        T.idx.resize(NC);
        DevBuf cidx(NC * 8 + 16);
        hipLaunchKernelGGL(k_compact_candidates, bbk::grid_blocks((n + 255) / 256), dim3(256), 0, ctx->stream,
                           flag.as<uint64_t>(), x->masks.as<uint8_t>(), visited.as<uint8_t>(), n,
                           cidx.as<uint64_t>());
        check_launch("k_compact_candidates");
        d2h_big(ctx, T.idx.data(), cidx.p, NC * 8);
        T.keys.resize(NC * T.W);
        T.masks.resize(NC);
        T.used.assign(NC, 0);
        std::vector<uint32_t> cand_bucket;
        // gather the candidate rows on the device (they may lie anywhere in a table of billions of k-mers)
        {
            DevBuf gk(NC * T.W * 8 + 16), gm(NC + 16);
            hipLaunchKernelGGL(k_gather_candidates, bbk::grid_blocks((NC + 255) / 256), dim3(256), 0, ctx->stream,
                               x->keys.as<uint64_t>(), x->masks.as<uint8_t>(), cidx.as<uint64_t>(), NC, T.W,
                               gk.as<uint64_t>(), gm.as<uint8_t>());
            check_launch("k_gather_candidates");
            d2h_big(ctx, T.keys.data(), gk.p, NC * T.W * 8);
            d2h_big(ctx, T.masks.data(), gm.p, NC);
            // CollectLoops (:308-344) takes the first unvisited k-mer in K-MER FILE ORDER, and the file is the
            // concatenation of 10 x threads XXH3 buckets, ascending inside: both the rotation of a loop string and the
            // palindrome SplitLoop cuts a self-conjugate circle at follow from that order.  ref_threads = the -t of the
            // reference run to reproduce (0: plain ascending order)
            if (ref_threads) {
                DevBuf gb(NC * 4 + 16);
                const uint64_t nb = 10ull * ref_threads;
                switch (T.W) {
                    case 1: hipLaunchKernelGGL(k_candidate_buckets<1>, bbk::grid_blocks((NC + 255) / 256), dim3(256), 0, ctx->stream, gk.as<uint64_t>(), NC, nb, gb.as<uint32_t>()); break;
                    case 2: hipLaunchKernelGGL(k_candidate_buckets<2>, bbk::grid_blocks((NC + 255) / 256), dim3(256), 0, ctx->stream, gk.as<uint64_t>(), NC, nb, gb.as<uint32_t>()); break;
                    case 3: hipLaunchKernelGGL(k_candidate_buckets<3>, bbk::grid_blocks((NC + 255) / 256), dim3(256), 0, ctx->stream, gk.as<uint64_t>(), NC, nb, gb.as<uint32_t>()); break;
                    default: hipLaunchKernelGGL(k_candidate_buckets<4>, bbk::grid_blocks((NC + 255) / 256), dim3(256), 0, ctx->stream, gk.as<uint64_t>(), NC, nb, gb.as<uint32_t>()); break;
                }
                check_launch("k_candidate_buckets");
                cand_bucket.resize(NC);
                d2h_big(ctx, cand_bucket.data(), gb.p, NC * 4);
            }
        }
        // visiting order of the candidates: ascending, or (bucket, ascending) = the reference's file order
        std::vector<uint64_t> visit(NC);
        for (uint64_t c = 0; c < NC; ++c) visit[c] = c;
        if (!cand_bucket.empty())
            std::stable_sort(visit.begin(), visit.end(), [&](uint64_t a, uint64_t b) { return cand_bucket[a] < cand_bucket[b]; });
        auto oriented_mask = [&](long pos, bool minimal) -> uint32_t {
            return minimal ? T.masks[(size_t)pos] : rev8(T.masks[(size_t)pos]);
        };
        auto emit = [&](const std::string &s) {
            // records + storage of one loop piece; CleanCondensed(s) and CleanCondensed(rc s)
            const std::string r = str_rc(s);
            const std::string &best = (s < r) ? r : s;  // push max(s, rc s) (:330-334)
            const uint64_t id = U.offsets.size() - 1;
            U.bases.insert(U.bases.end(), best.begin(), best.end());
            U.offsets.push_back(U.bases.size());
            const bool selfconj = best == str_rc(best);
            for (int is_start = 1; is_start >= 0; --is_start) {
                if (!is_start && selfconj) continue;
                const std::string km = is_start ? best.substr(0, (size_t)k) : best.substr(best.size() - (size_t)k);
                bool minimal;
                const long pos = T.find(km, &minimal);
                BBK_REQUIRE(pos >= 0, BBK_ERR_INTERNAL, "loop end k-mer missing from the candidate table");
                recs.push_back({(T.idx[(size_t)pos] << 2) | ((uint64_t)(minimal ? 0 : 1) << 1) |
                                    (uint64_t)is_start,
                                (uint32_t)id});
            }
            for (const std::string *t : {&s, &r})
                for (size_t p = 0; p + (size_t)k <= t->size(); ++p) {
                    bool minimal;
                    const long pos = T.find(t->substr(p, (size_t)k), &minimal);
                    if (pos >= 0) T.used[(size_t)pos] = 1;
                }
            ++n_loops;
        };
        for (uint64_t vi = 0; vi < NC; ++vi) {
            const uint64_t c = visit[vi];
            if (T.used[c]) continue;
            // ConstructLoopFromVertex (:255-265) from the canonical k-mer
            const std::string x0 = unpack_kmer(&T.keys[c * T.W], k);
            std::string s = x0;
            std::string cur = x0;
            uint32_t m = T.masks[c];
            for (;;) {
                const int cb = __builtin_ctz(m & 15u);
                cur = cur.substr(1) + "ACGT"[cb];
                if (cur == x0) {  // edge (prev -> x0) closes the cycle: its base is appended, then stop
                    s.push_back("ACGT"[cb]);
                    break;
                }
                s.push_back("ACGT"[cb]);
                bool minimal;
                const long pos = T.find(cur, &minimal);
                BBK_REQUIRE(pos >= 0, BBK_ERR_INTERNAL, "loop walk left the candidate set");
                m = oriented_mask(pos, minimal);
                BBK_REQUIRE(!mask_is_junction(m), BBK_ERR_INTERNAL, "loop walk reached a junction");
                BBK_REQUIRE(s.size() <= 2 * NC + (size_t)k + 1, BBK_ERR_INTERNAL, "loop walk does not close");
            }
            // the reference string ends when the walk is back on its first EDGE: x0 . (cycle bases) with
            // the closing k-mer x0 spelled again minus ... -> length = cycle + k  (:232-241)
            // s currently = x0 + one base per cycle edge (cycle edges = n_cyc) -> length k + n_cyc: equal.
            // SplitLoop (:248-252) on the first (k+1)-mer equal to its own reverse complement
            size_t split = std::string::npos;
            for (size_t p = 0; p + (size_t)k + 1 <= s.size(); ++p) {
                const std::string e = s.substr(p, (size_t)k + 1);
                if (e == str_rc(e)) {
                    split = p;
                    break;
                }
            }
            if (split == std::string::npos) {
                emit(s);
            } else {
                emit(s.substr(split, (size_t)k + 1));
                emit(s.substr(split + 1, s.size() - (size_t)k - (split + 1)) + s.substr(0, split + (size_t)k));
            }
        }
        std::stable_sort(recs.begin(), recs.end(), [](const LinkRec &a, const LinkRec &b) {
            return a.key != b.key ? a.key < b.key : a.edge < b.edge;
        });
    }
    U.n = n_paths + n_loops;
    U.n_loops = n_loops;
    U.host_valid = true;

    // ---- vertices + links (gfa_writer.cpp:43-52 over construction_helper.hpp:80-90)
    std::vector<uint8_t> selfconj(U.n, 0);
    {
        // an edge with a start record but no end record is self-conjugate
        std::vector<uint8_t> has_end(U.n, 0);
        for (const LinkRec &r : recs)
            if (!(r.key & 1)) has_end[r.edge] = 1;
        for (uint64_t i = 0; i < U.n; ++i) selfconj[i] = !has_end[i];
    }
    uint64_t nv = 0;
    for (size_t p = 0; p < recs.size();) {
        size_t q = p;
        const uint64_t h = recs[p].key >> 2;
        while (q < recs.size() && (recs[q].key >> 2) == h) ++q;
        ++nv;
        for (size_t a = p; a < q; ++a) {
            const bool a_start = recs[a].key & 1, a_rc = (recs[a].key >> 1) & 1;
            if (!((!a_start && !a_rc) || (a_start && a_rc))) continue;  // incoming at the canonical vertex
            const uint32_t ea = recs[a].edge;
            const uint32_t oa = (!a_start || selfconj[ea]) ? 1u : 0u;
            for (size_t b = p; b < q; ++b) {
                const bool b_start = recs[b].key & 1, b_rc = (recs[b].key >> 1) & 1;
                if (!((b_start && !b_rc) || (!b_start && b_rc))) continue;  // outgoing
                const uint32_t eb = recs[b].edge;
                const uint32_t ob = (b_start || selfconj[eb]) ? 1u : 0u;
                U.links.push_back(((uint64_t)ea << 1) | oa);
                U.links.push_back(((uint64_t)eb << 1) | ob);
            }
        }
        p = q;
    }
    U.n_vertices = nv;
    U.n_links = U.links.size() / 2;
}

unsigned build_prefix_index(bbk_ctx *ctx, const uint64_t *keys, unsigned W, unsigned k, uint64_t n, DevBuf &prefix,
                            bool *wide);

// One thread per unitig: roll the (k+1)-mers of the sequence, look the canonical form up in the
// sorted (k+1)-mer count table, add the multiplicities (GraphCoverageFiller,
// assembly_graph/graph_support/coverage_filling.hpp:44-62).
template <int W>
__global__ __launch_bounds__(256) void k_unitig_kc(const char *__restrict__ bases, const uint64_t *__restrict__ off,
                                                  uint64_t n_unitigs, int k1, const Key<W> *__restrict__ keys,
                                                  const uint32_t *__restrict__ counts, PrefixTable P,
                                                  uint64_t *__restrict__ kc, uint32_t *__restrict__ err) {
    const uint64_t u = BBK_GID();
    if (u >= n_unitigs) return;
    const char *s = bases + off[u];
    const uint64_t len = off[u + 1] - off[u];
    uint64_t sum = 0;
    if (len >= (uint64_t)k1) {
        Key<W> cur;
#pragma unroll
        for (int j = 0; j < W; ++j) cur.w[j] = 0;
        auto code = [](char c) -> uint32_t { return c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : 3u; };
        for (int i = 0; i < k1 - 1; ++i) cur = kmer_shl<W>(cur, k1, code(s[i]));
        for (uint64_t p = (uint64_t)k1 - 1; p < len; ++p) {
            cur = kmer_shl<W>(cur, k1, code(s[p]));
            const Key<W> rc = kmer_rc<W>(cur, k1);
            const bool minimal = !kmer_less_nucl<W>(rc, cur);
            const Key<W> q = key_select<W>(minimal, cur, rc);
            const uint64_t j = table_find<W>(keys, P, q);
            if (j == kNotFound) atomicOr(err, 4u);
            else sum += counts[j];
        }
    }
    kc[u] = sum;
}

template <int W>
static void run_kc(bbk_ctx *ctx, const char *d_bases, const uint64_t *d_off, uint64_t nu, unsigned k1,
                   const bbk_kmerset *set, const DevBuf &pref, unsigned pbits, bool wide, uint64_t *d_kc, uint32_t *d_err) {
    const int w0bits = (W == 1) ? (int)(2 * k1) : 64;
    KernelTimer t(ctx, "coverage", 0);
    hipLaunchKernelGGL(k_unitig_kc<W>, bbk::grid_blocks((nu + 255) / 256), dim3(256), 0, ctx->stream, d_bases, d_off, nu,
                       (int)k1, set->keys.as<Key<W>>(), set->counts.as<uint32_t>(),
                       PrefixTable{pref.p, w0bits - (int)pbits, wide ? 1 : 0}, d_kc, d_err);
    check_launch("k_unitig_kc");
}

}  // namespace bbk

using namespace bbk;

extern "C" {

// KC of every unitig from a table of canonical (k+1)-mer multiplicities
static void coverage_from_counts(bbk_ctx *ctx, bbk_unitigs *u, const bbk_kmerset *set) {
    const unsigned k1 = u->k + 1;
    BBK_REQUIRE(set->k == k1 && set->has_counts && set->sorted && !set->ref_order && (set->flags & BBK_CANONICAL),
                BBK_ERR_ARG, "coverage needs the ascending canonical %u-mer set with counts "
                "(BBK_CANONICAL | BBK_WITH_COUNTS at k + 1)", k1);
    ensure_host(ctx, u);
    u->kc.assign(u->n, 0);
    u->has_cov = true;
    if (u->n == 0) return;
    DevBuf pref;
    bool wide = false;
    const unsigned pbits = build_prefix_index(ctx, set->keys.as<uint64_t>(), set->W, k1, set->n, pref, &wide);
    DevBuf d_bases(u->bases.size() + 16), d_off((u->n + 1) * 8), d_kc(u->n * 8), d_err(16);
    BBK_HIP(hipMemcpyAsync(d_bases.p, u->bases.data(), u->bases.size(), hipMemcpyHostToDevice, ctx->stream));
    BBK_HIP(hipMemcpyAsync(d_off.p, u->offsets.data(), (u->n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    BBK_HIP(hipMemsetAsync(d_err.p, 0, 16, ctx->stream));
    switch (set->W) {
        case 1: run_kc<1>(ctx, d_bases.as<char>(), d_off.as<uint64_t>(), u->n, k1, set, pref, pbits, wide, d_kc.as<uint64_t>(), d_err.as<uint32_t>()); break;
        case 2: run_kc<2>(ctx, d_bases.as<char>(), d_off.as<uint64_t>(), u->n, k1, set, pref, pbits, wide, d_kc.as<uint64_t>(), d_err.as<uint32_t>()); break;
        case 3: run_kc<3>(ctx, d_bases.as<char>(), d_off.as<uint64_t>(), u->n, k1, set, pref, pbits, wide, d_kc.as<uint64_t>(), d_err.as<uint32_t>()); break;
        case 4: run_kc<4>(ctx, d_bases.as<char>(), d_off.as<uint64_t>(), u->n, k1, set, pref, pbits, wide, d_kc.as<uint64_t>(), d_err.as<uint32_t>()); break;
        default: BBK_REQUIRE(false, BBK_ERR_ARG, "unsupported key width %u", set->W);
    }
    uint32_t herr = 0;
    d2h(ctx, &herr, d_err.p, 4);
    BBK_REQUIRE(herr == 0, BBK_ERR_INTERNAL, "coverage: a (k+1)-mer of a unitig is missing from the count table");
    d2h(ctx, u->kc.data(), d_kc.p, u->n * 8);
}

int bbk_unitigs_add_coverage(bbk_ctx *ctx, bbk_unitigs *u, const bbk_reads *reads) {
    return guarded([&] {
        BBK_REQUIRE(ctx && u && reads, BBK_ERR_ARG, "bbk_unitigs_add_coverage: NULL argument");
        BBK_HIP(hipSetDevice(ctx->device));
        // multiplicities of the canonical (k+1)-mers over reads + rc(reads)
        // (CoverageHashMapBuilder::FillCoverageFromStream, utils/ph_map/coverage_hash_map_builder.hpp:15-38)
        bbk_kmerset *set = nullptr;
        const int rc = bbk_count(ctx, reads, u->k + 1, BBK_CANONICAL | BBK_WITH_COUNTS, &set);
        if (rc != BBK_OK) throw Error{rc};
        std::unique_ptr<bbk_kmerset, void (*)(bbk_kmerset *)> guard(set, bbk_kmerset_free);
        coverage_from_counts(ctx, u, set);
    });
}

int bbk_unitigs_add_coverage_counts(bbk_ctx *ctx, bbk_unitigs *u, const bbk_kmerset *kp1_counts) {
    return guarded([&] {
        BBK_REQUIRE(ctx && u && kp1_counts, BBK_ERR_ARG, "bbk_unitigs_add_coverage_counts: NULL argument");
        BBK_HIP(hipSetDevice(ctx->device));
        coverage_from_counts(ctx, u, kp1_counts);
    });
}

int bbk_unitigs_export_kc(bbk_ctx *ctx, const bbk_unitigs *u, uint64_t *h_kc) {
    return guarded([&] {
        BBK_REQUIRE(ctx && u && h_kc && u->has_cov, BBK_ERR_ARG, "bbk_unitigs_export_kc: no coverage (call bbk_unitigs_add_coverage)");
        if (u->n) memcpy(h_kc, u->kc.data(), u->n * sizeof(uint64_t));
    });
}

int bbk_unitigs_to_reads(bbk_ctx *ctx, const bbk_unitigs *u, bbk_reads **out) {
    return guarded([&] {
        BBK_REQUIRE(ctx && u && out, BBK_ERR_ARG, "bbk_unitigs_to_reads: NULL argument");
        BBK_HIP(hipSetDevice(ctx->device));
        const uint64_t nu = u->n;
        // the device-resident result is used in place; a result that lives on the host (loops were appended) is uploaded
        DevBuf up_bases, up_off;
        const char *d_bases = u->d_bases.as<char>();
        const uint64_t *d_uoff = u->d_uoff.as<uint64_t>();
        uint64_t total = u->total_bases;
        if (u->host_valid) {
            total = u->bases.size();
            up_bases.alloc(total + 16);
            up_off.alloc((nu + 1) * 8);
            if (total) BBK_HIP(hipMemcpyAsync(up_bases.p, u->bases.data(), total, hipMemcpyHostToDevice, ctx->stream));
            BBK_HIP(hipMemcpyAsync(up_off.p, u->offsets.data(), (nu + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
            d_bases = up_bases.as<char>();
            d_uoff = up_off.as<uint64_t>();
        }
        auto rd = std::make_unique<bbk_reads>();
        rd->ctx = ctx;
        rd->n = nu;
        rd->bases = total;
        rd->own_woff.alloc((nu + 1) * sizeof(uint64_t));
        rd->own_len.alloc((nu + 1) * sizeof(uint32_t));
        uint64_t nwords = 0;
        if (nu) {
            DevBuf err(16);
            BBK_HIP(hipMemsetAsync(err.p, 0, 16, ctx->stream));
            hipLaunchKernelGGL(k_unitig_words, bbk::grid_blocks((nu + 255) / 256), dim3(256), 0, ctx->stream, d_uoff, nu,
                               rd->own_woff.as<uint64_t>(), rd->own_len.as<uint32_t>(), err.as<uint32_t>());
            check_launch("k_unitig_words");
            nwords = exclusive_scan_u64(ctx, rd->own_woff.as<uint64_t>(), rd->own_woff.as<uint64_t>(), nu);
            uint32_t herr = 0;
            d2h(ctx, &herr, err.p, 4);
            BBK_REQUIRE(herr == 0, BBK_ERR_ARG, "bbk_unitigs_to_reads: a unitig is longer than 2^32 - 1 bases");
        }
        BBK_HIP(hipMemcpyAsync(rd->own_woff.as<uint64_t>() + nu, &nwords, 8, hipMemcpyHostToDevice, ctx->stream));
        rd->n_words = nwords;
        rd->own_words.alloc((nwords + 1) * sizeof(uint64_t));
        if (nu) {
            hipLaunchKernelGGL(k_pack_unitigs, bbk::grid_blocks((nu * 64 + 255) / 256), dim3(256), 0, ctx->stream, d_bases,
                               d_uoff, rd->own_woff.as<uint64_t>(), nu, rd->own_words.as<uint64_t>());
            check_launch("k_pack_unitigs");
        }
        BBK_HIP(hipStreamSynchronize(ctx->stream));
        rd->d_words = rd->own_words.as<uint64_t>();
        rd->d_woff = rd->own_woff.as<uint64_t>();
        rd->d_len = rd->own_len.as<uint32_t>();
        *out = rd.release();
    });
}

int bbk_unitigs_build_ex(bbk_ctx *ctx, bbk_extindex *x, unsigned ref_threads, bbk_unitigs **out) {
    return guarded([&] {
        BBK_REQUIRE(ctx && x && out, BBK_ERR_ARG, "bbk_unitigs_build: NULL argument");
        BBK_REQUIRE(ref_threads <= (1u << 20), BBK_ERR_ARG, "bbk_unitigs_build_ex: ref_threads %u", ref_threads);
        BBK_HIP(hipSetDevice(ctx->device));
        auto u = std::make_unique<bbk_unitigs>();
        build(ctx, x, *u, ref_threads);
        *out = u.release();
    });
}

int bbk_unitigs_build(bbk_ctx *ctx, bbk_extindex *x, bbk_unitigs **out) { return bbk_unitigs_build_ex(ctx, x, 0, out); }

uint64_t bbk_unitigs_count(const bbk_unitigs *u) { return u ? u->n : 0; }
uint64_t bbk_unitigs_loops(const bbk_unitigs *u) { return u ? u->n_loops : 0; }
uint64_t bbk_unitigs_total_bases(const bbk_unitigs *u) {
    return u ? (u->host_valid ? u->bases.size() : u->total_bases) : 0;
}
uint64_t bbk_unitigs_vertices(const bbk_unitigs *u) { return u ? u->n_vertices : 0; }
uint64_t bbk_unitigs_links(const bbk_unitigs *u) { return u ? u->n_links : 0; }

int bbk_unitigs_export(bbk_ctx *ctx, const bbk_unitigs *u, char *h_bases, uint64_t *h_offsets) {
    return guarded([&] {
        BBK_REQUIRE(ctx && u, BBK_ERR_ARG, "bbk_unitigs_export: NULL argument");
        ensure_host(ctx, u);
        if (h_bases && !u->bases.empty()) memcpy(h_bases, u->bases.data(), u->bases.size());
        if (h_offsets) memcpy(h_offsets, u->offsets.data(), u->offsets.size() * sizeof(uint64_t));
    });
}

int bbk_unitigs_export_links(bbk_ctx *ctx, const bbk_unitigs *u, uint32_t *h_links) {
    return guarded([&] {
        BBK_REQUIRE(ctx && u && (u->n_links == 0 || h_links), BBK_ERR_ARG, "bbk_unitigs_export_links: NULL argument");
        ensure_host(ctx, u);
        for (uint64_t l = 0; l < u->n_links; ++l) {
            h_links[4 * l] = (uint32_t)(u->links[2 * l] >> 1);  // edge ids are below 2^32 - 1 (build)
            h_links[4 * l + 1] = (uint32_t)(u->links[2 * l] & 1u);
            h_links[4 * l + 2] = (uint32_t)(u->links[2 * l + 1] >> 1);
            h_links[4 * l + 3] = (uint32_t)(u->links[2 * l + 1] & 1u);
        }
    });
}

static size_t fmt_u64(char *dst, uint64_t v) {
    char tmp[24];
    size_t n = 0;
    do {
        tmp[n++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    for (size_t i = 0; i < n; ++i) dst[i] = tmp[n - 1 - i];
    return n;
}

// GFA text formatted on the device from the device-resident result, then streamed to the file in
// pinned chunks (copy of chunk i+1 overlaps the pwrite of chunk i).
static void write_gfa_device(bbk_ctx *ctx, const bbk_unitigs *u, const char *path) {
    BBK_HIP(hipSetDevice(ctx->device));
    const uint64_t nu = u->n, nl = u->n_links;
    DevBuf spos((nu + 1) * 8), lpos((nl + 1) * 8);
    uint64_t sbytes = 0, lbytes = 0;
    if (nu) {
        hipLaunchKernelGGL(k_gfa_s_len, bbk::grid_blocks((nu + 255) / 256), dim3(256), 0, ctx->stream,
                           u->d_uoff.as<uint64_t>(), nu, spos.as<uint64_t>());
        check_launch("k_gfa_s_len");
        sbytes = exclusive_scan_u64(ctx, spos.as<uint64_t>(), spos.as<uint64_t>(), nu);
    }
    uint32_t klen = 1;
    for (unsigned v = u->k; v >= 10; v /= 10) ++klen;
    if (nl) {
        hipLaunchKernelGGL(k_gfa_l_len, bbk::grid_blocks((nl + 255) / 256), dim3(256), 0, ctx->stream,
                           u->d_links.as<uint64_t>(), nl, klen, lpos.as<uint64_t>());
        check_launch("k_gfa_l_len");
        lbytes = exclusive_scan_u64(ctx, lpos.as<uint64_t>(), lpos.as<uint64_t>(), nl);
    }
    const uint64_t total = sbytes + lbytes;
    DevBuf text(total + 16);
    {
        KernelTimer t(ctx, "gfa_text", (double)total + (double)u->total_bases);
        if (nu) {
            hipLaunchKernelGGL(k_gfa_s_write, bbk::grid_blocks((nu * 64 + 255) / 256), dim3(256), 0, ctx->stream,
                               u->d_bases.as<char>(), u->d_uoff.as<uint64_t>(), spos.as<uint64_t>(), nu,
                               text.as<char>());
            check_launch("k_gfa_s_write");
        }
        if (nl) {
            hipLaunchKernelGGL(k_gfa_l_write, bbk::grid_blocks((nl + 255) / 256), dim3(256), 0, ctx->stream,
                               u->d_links.as<uint64_t>(), lpos.as<uint64_t>(), nl, (uint32_t)u->k, klen,
                               text.as<char>() + sbytes);
            check_launch("k_gfa_l_write");
        }
    }
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    BBK_REQUIRE(fd >= 0, BBK_ERR_IO, "cannot open %s for writing", path);
    const bool ok = d2f_big(ctx, fd, 0, text.p, (size_t)total);
    const int cl = close(fd);
    BBK_REQUIRE(ok && cl == 0, BBK_ERR_IO, "short write to %s", path);
}

static inline size_t dec_len(uint64_t v) {
    size_t n = 1;
    while (v >= 10) {
        v /= 10;
        ++n;
    }
    return n;
}

// parallel positional writes of one buffer (tmpfs / NVMe scale with writers; a single fwrite of
// 1.5 GB is a third of the whole GFA time otherwise)
// host worker threads for text formatting / file writes: the box may expose hundreds of logical CPUs
// of which only a share is ours
static int host_threads() { return std::max(1, std::min(omp_get_max_threads(), 32)); }

static bool pwrite_all(int fd, const char *buf, size_t bytes, off_t base) {
    const size_t chunk = 16ull << 20;
    const size_t nchunks = (bytes + chunk - 1) / chunk;
    bool ok = true;
#pragma omp parallel for schedule(dynamic, 1) num_threads(host_threads())
    for (size_t c = 0; c < nchunks; ++c) {
        size_t off = c * chunk;
        const size_t end = std::min(bytes, off + chunk);
        while (off < end) {
            const ssize_t w = pwrite(fd, buf + off, end - off, base + (off_t)off);
            if (w <= 0) {
#pragma omp atomic write
                ok = false;
                break;
            }
            off += (size_t)w;
        }
    }
    return ok;
}

int bbk_unitigs_write_gfa(bbk_ctx *ctx, const bbk_unitigs *u, const char *path) {
    return guarded([&] {
        BBK_REQUIRE(ctx && u && path, BBK_ERR_ARG, "bbk_unitigs_write_gfa: NULL argument");
        if (!u->host_valid && !u->has_cov) {
            write_gfa_device(ctx, u, path);
            return;
        }
        ensure_host(ctx, u);
        // S\t<id>\t<seq>\tDP:f:<cov>\tKC:i:<kc>\n with id = 3 + 2i (graph_core.hpp:228; edge i gets
        // min_id + 2i, debruijn_graph_constructor.hpp:457-458); coverage is 0 without -c.
        const uint64_t n = u->n;
        const bool verbose = getenv("BBK_VERBOSE") != nullptr;
        double t_prev = omp_get_wtime();
        auto lap = [&](const char *what) {
            if (verbose) {
                const double t = omp_get_wtime();
                fprintf(stderr, "[bbk] write_gfa %-10s %.3f s\n", what, t - t_prev);
                t_prev = t;
            }
        };
        raw_vector<uint64_t> pos(n + 1);
        // per-segment tail "\tDP:f:<float(KC/(len-k))>\tKC:i:<KC>\n": default ostream formatting of a float
        // is %g with 6 significant digits (gfa_writer.cpp:18-25; coverage = raw / length, coverage.hpp:58-64)
        std::vector<std::string> tails;
        static const char tail0[] = "\tDP:f:0\tKC:i:0\n";
        if (u->has_cov) {
            tails.resize(n);
#pragma omp parallel for schedule(static) num_threads(host_threads())
            for (uint64_t i = 0; i < n; ++i) {
                const uint64_t len = u->offsets[i + 1] - u->offsets[i];
                const double cov = (double)u->kc[i] / (double)(len - u->k);
                char b[96];
                snprintf(b, sizeof(b), "\tDP:f:%g\tKC:i:%llu\n", (double)(float)cov, (unsigned long long)u->kc[i]);
                tails[i] = b;
            }
        }
        // line lengths -> offsets (two-level parallel prefix sum)
        auto prefix_sum = [](raw_vector<uint64_t> &v, uint64_t cnt) {  // v[i+1] holds the length of item i; v[0] = 0
            const int T = host_threads();
            std::vector<uint64_t> part((size_t)T + 1, 0);
#pragma omp parallel num_threads(T)
            {
                const int t = omp_get_thread_num();
                const uint64_t lo = cnt * (uint64_t)t / T, hi = cnt * (uint64_t)(t + 1) / T;
                uint64_t sacc = 0;
                for (uint64_t i = lo; i < hi; ++i) sacc += v[i + 1];
                part[(size_t)t + 1] = sacc;
#pragma omp barrier
#pragma omp single
                for (int j = 0; j < T; ++j) part[(size_t)j + 1] += part[(size_t)j];
                uint64_t run = part[(size_t)t];
                for (uint64_t i = lo; i < hi; ++i) {
                    run += v[i + 1];
                    v[i + 1] = run;
                }
            }
        };
        pos[0] = 0;
#pragma omp parallel for schedule(static) num_threads(host_threads())
        for (uint64_t i = 0; i < n; ++i) {
            const size_t tl = u->has_cov ? tails[i].size() : sizeof(tail0) - 1;
            pos[i + 1] = 2 + dec_len(3 + 2 * i) + 1 + (u->offsets[i + 1] - u->offsets[i]) + tl;
        }
        prefix_sum(pos, n);
        lap("S-sizes");
        raw_vector<char> buf(pos[n]);
#pragma omp parallel for schedule(static) num_threads(host_threads())
        for (uint64_t i = 0; i < n; ++i) {
            char *d = buf.data() + pos[i];
            *d++ = 'S';
            *d++ = '\t';
            d += fmt_u64(d, 3 + 2 * i);
            *d++ = '\t';
            const uint64_t len = u->offsets[i + 1] - u->offsets[i];
            memcpy(d, u->bases.data() + u->offsets[i], len);
            d += len;
            if (u->has_cov) memcpy(d, tails[i].data(), tails[i].size());
            else memcpy(d, tail0, sizeof(tail0) - 1);
        }
        lap("S-format");
        // L\t<e1>\t<+|->\t<e2>\t<+|->\t<k>M\n
        const uint64_t nl = u->n_links;
        const size_t kl = dec_len(u->k);
        raw_vector<uint64_t> lpos(nl + 1);
        lpos[0] = 0;
#pragma omp parallel for schedule(static) num_threads(host_threads())
        for (uint64_t l = 0; l < nl; ++l)
            lpos[l + 1] = 2 + dec_len(3 + 2 * (u->links[2 * l] >> 1)) + 3 + dec_len(3 + 2 * (u->links[2 * l + 1] >> 1)) + 3 + kl + 2;
        prefix_sum(lpos, nl);
        raw_vector<char> lbuf(lpos[nl]);
#pragma omp parallel for schedule(static) num_threads(host_threads())
        for (uint64_t l = 0; l < nl; ++l) {
            char *d = lbuf.data() + lpos[l];
            const uint64_t a = u->links[2 * l], b2 = u->links[2 * l + 1];
            *d++ = 'L';
            *d++ = '\t';
            d += fmt_u64(d, 3 + 2 * (a >> 1));
            *d++ = '\t';
            *d++ = (a & 1u) ? '+' : '-';
            *d++ = '\t';
            d += fmt_u64(d, 3 + 2 * (b2 >> 1));
            *d++ = '\t';
            *d++ = (b2 & 1u) ? '+' : '-';
            *d++ = '\t';
            d += fmt_u64(d, u->k);
            *d++ = 'M';
            *d++ = '\n';
        }
        lap("L-format");
        const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
        BBK_REQUIRE(fd >= 0, BBK_ERR_IO, "cannot open %s for writing", path);
        bool ok = pwrite_all(fd, buf.data(), buf.size(), 0) && pwrite_all(fd, lbuf.data(), lbuf.size(), (off_t)buf.size());
        const int cl = close(fd);
        lap("pwrite");
        BBK_REQUIRE(ok && cl == 0, BBK_ERR_IO, "short write to %s", path);
    });
}

int bbk_unitigs_write_fastg(bbk_ctx *ctx, const bbk_unitigs *u, const char *path) {
    return guarded([&] {
        BBK_REQUIRE(ctx && u && path, BBK_ERR_ARG, "bbk_unitigs_write_fastg: NULL argument");
        ensure_host(ctx, u);
        // FastgWriter::WriteSegmentsAndLinks (common/io/graph/fastg_writer.cpp:20-47): one FASTA record per
        // edge AND per conjugate edge; header = name, ':' + comma-separated names of the edges leaving its end
        // vertex (a std::set, i.e. sorted as strings), ';'.  Names are BasicNamingF
        // (io/utils/edge_namer.hpp:33-38): EDGE_<id>_length_<len>_cov_<to_string(cov)>, a conjugate edge is
        // the canonical name + "'" (extended_namer_, fastg_writer.hpp:30).  Record order in the reference
        // follows its vertex numbering (BooPHF order); here: edge id order, the edge before its conjugate.
        const uint64_t n = u->n;
        std::vector<uint8_t> selfconj(n, 0);
#pragma omp parallel for schedule(static) num_threads(host_threads())
        for (uint64_t i = 0; i < n; ++i) {
            const char *sq = u->bases.data() + u->offsets[i];
            const uint64_t len = u->offsets[i + 1] - u->offsets[i];
            bool sc = true;
            for (uint64_t a = 0; a < len && sc; ++a) {
                const char c = sq[len - 1 - a];
                const char r = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A';
                sc = sq[a] == r;
            }
            selfconj[i] = sc;
        }
        auto flip = [&](uint64_t t) { return selfconj[t >> 1] ? t : (t ^ 1ull); };
        // adjacency of oriented edges: a stored link x -> y also means rc(y) -> rc(x)
        std::vector<std::pair<uint64_t, uint64_t>> adj;
        adj.reserve(2 * u->n_links);
        for (uint64_t l = 0; l < u->n_links; ++l) {
            const uint64_t x = u->links[2 * l], y = u->links[2 * l + 1];
            adj.emplace_back(x, y);
            adj.emplace_back(flip(y), flip(x));
        }
        std::sort(adj.begin(), adj.end());
        adj.erase(std::unique(adj.begin(), adj.end()), adj.end());
        auto name = [&](uint64_t t) {
            const uint64_t i = t >> 1;
            const uint64_t len = u->offsets[i + 1] - u->offsets[i];
            const double cov = u->has_cov ? (double)u->kc[i] / (double)(len - u->k) : 0.0;
            std::string s = "EDGE_" + std::to_string(3 + 2 * i) + "_length_" + std::to_string(len) + "_cov_" +
                            std::to_string(cov);
            if (!(t & 1ull)) s += "'";
            return s;
        };
        FILE *f = fopen(path, "wb");
        BBK_REQUIRE(f != nullptr, BBK_ERR_IO, "cannot open %s for writing", path);
        bool ok = true;
        std::string seq, hdr;
        size_t ai = 0;
        for (uint64_t i = 0; i < n && ok; ++i) {
            for (int o = 1; o >= 0 && ok; --o) {
                if (o == 0 && selfconj[i]) continue;
                const uint64_t t = (i << 1) | (uint64_t)o;
                // successors of t: adj is sorted by (from, to); orientation '-' (0) sorts before '+' (1)
                auto lo = std::lower_bound(adj.begin(), adj.end(), std::make_pair(t, (uint64_t)0));
                std::vector<std::string> next;
                for (auto it = lo; it != adj.end() && it->first == t; ++it) next.push_back(name(it->second));
                std::sort(next.begin(), next.end());
                hdr = ">" + name(t);
                const char *delim = ":";
                for (const std::string &nx : next) {
                    hdr += delim;
                    hdr += nx;
                    delim = ",";
                }
                hdr += ";\n";
                const char *sq = u->bases.data() + u->offsets[i];
                const uint64_t len = u->offsets[i + 1] - u->offsets[i];
                seq.assign(sq, sq + len);
                if (o == 0) {
                    std::reverse(seq.begin(), seq.end());
                    for (char &c : seq) c = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A';
                }
                ok = fwrite(hdr.data(), 1, hdr.size(), f) == hdr.size();
                for (uint64_t cur = 0; cur < len && ok; cur += 60) {
                    const uint64_t w = std::min<uint64_t>(60, len - cur);
                    ok = fwrite(seq.data() + cur, 1, w, f) == w && fputc('\n', f) != EOF;
                }
            }
        }
        (void)ai;
        const int cl = fclose(f);
        BBK_REQUIRE(ok && cl == 0, BBK_ERR_IO, "short write to %s", path);
    });
}

int bbk_unitigs_write_fasta(bbk_ctx *ctx, const bbk_unitigs *u, const char *path) {
    return guarded([&] {
        BBK_REQUIRE(ctx && u && path, BBK_ERR_ARG, "bbk_unitigs_write_fasta: NULL argument");
        ensure_host(ctx, u);
        // >EDGE_<i+1>_length_<len> + 60-column wrapped sequence (projects/gbuilder/main.cpp:183-192,
        // io/reads/header_naming.hpp:14-20, osequencestream.hpp:22-28)
        FILE *f = fopen(path, "wb");
        BBK_REQUIRE(f != nullptr, BBK_ERR_IO, "cannot open %s for writing", path);
        bool ok = true;
        for (uint64_t i = 0; i < u->n && ok; ++i) {
            const uint64_t len = u->offsets[i + 1] - u->offsets[i];
            ok = fprintf(f, ">EDGE_%llu_length_%llu\n", (unsigned long long)(i + 1), (unsigned long long)len) > 0;
            for (uint64_t cur = 0; cur < len && ok; cur += 60) {
                const uint64_t w = std::min<uint64_t>(60, len - cur);
                ok = fwrite(u->bases.data() + u->offsets[i] + cur, 1, w, f) == w && fputc('\n', f) != EOF;
            }
        }
        const int cl = fclose(f);
        BBK_REQUIRE(ok && cl == 0, BBK_ERR_IO, "short write to %s", path);
    });
}

// SPAdes binary graph: <basename>.grseq (io::binary::GraphIO::SaveImpl, common/io/binary/graph.hpp:27-46) +
// <basename>.cvr (BaseCoverageIO::SaveImpl, common/io/binary/coverage.hpp:24-29), what `spades-gbuilder --spades`
// writes through BasicGraphIO::Save (common/io/binary/basic.hpp:24-27, projects/gbuilder/main.cpp:221-222).
//   .grseq: u64 vreserved, u64 ereserved, u64 vertex_count; per vertex (id order): u64 id, u64 conjugate id, then per
//           outgoing edge e1 with conj(e1) >= e1: u64 e1, u64 e2 = conj(e1), u64 EdgeEnd(e1), u64 EdgeStart(e2),
//           Sequence (u64 length + ceil(length/32) u64 words, 2 bits per base, Sequence::BinWrite
//           common/sequence/sequence.hpp:431-442); u64 0 ends the vertex.
//   .cvr:   per canonical edge u64 id, u32 raw coverage; u64 0 at the end.
// Ids: edge i (GFA segment 3+2i) and its conjugate 3+2i+1 (a self-conjugate edge is its own), as
// FastGraphFromSequencesConstructor numbers them (debruijn_graph_constructor.hpp:450-465, graph_core.hpp:228,610-624).
// Vertices: one pair per distinct canonical end k-mer, numbered 3+2j / 3+2j+1 in ascending k-mer order -- the
// reference numbers them in BooPHF-index order (:494-515), which no other implementation can reproduce, and its
// loader (LoadImpl :48-96) accepts any consistent numbering; parity is therefore structural (tests rebuild the graph
// from the file and compare it with the GFA).
static void write_spades_graph(bbk_ctx *ctx, const bbk_unitigs *u, const char *basename) {
    ensure_host(ctx, u);
    const unsigned k = u->k;
    const uint64_t nu = u->n;
    const int W = (int)words_of(k);
    // end k-mers of every edge in canonical form
    struct End {
        uint64_t w[4];
        uint32_t edge;
        uint8_t is_end, is_rc;
    };
    std::vector<End> ends(2 * nu);
#pragma omp parallel for schedule(static) num_threads(host_threads())
    for (uint64_t i = 0; i < nu; ++i) {
        const char *s = u->bases.data() + u->offsets[i];
        const uint64_t len = u->offsets[i + 1] - u->offsets[i];
        for (int e = 0; e < 2; ++e) {
            const std::string km(s + (e ? len - k : 0), k);
            const std::string r = str_rc(km);
            const bool minimal = km <= r;  // IsMinimal: base-lexicographic, ties minimal (rtseq.hpp:407-415)
            End &d = ends[2 * i + e];
            memset(d.w, 0, sizeof(d.w));
            pack_kmer((minimal ? km : r).data(), (int)k, d.w, W);
            d.edge = (uint32_t)i;
            d.is_end = (uint8_t)e;
            d.is_rc = minimal ? 0 : 1;
        }
    }
    std::vector<uint32_t> order(2 * nu);
    for (uint64_t i = 0; i < 2 * nu; ++i) order[i] = (uint32_t)i;
    auto less = [&](uint32_t a, uint32_t b) {
        for (int w = 0; w < W; ++w)
            if (ends[a].w[w] != ends[b].w[w]) return ends[a].w[w] < ends[b].w[w];
        return a < b;
    };
    std::sort(order.begin(), order.end(), less);
    // vertex pair j for every end; vid(end) = 3 + 2j + (k-mer is the reverse complement of the canonical form)
    std::vector<uint64_t> vid(2 * nu);
    uint64_t nv = 0;
    for (uint64_t r = 0; r < 2 * nu; ++r) {
        const uint32_t a = order[r];
        if (r > 0) {
            const uint32_t p = order[r - 1];
            bool same = true;
            for (int w = 0; w < W; ++w) same = same && ends[a].w[w] == ends[p].w[w];
            if (!same) ++nv;
        }
        vid[a] = 3 + 2 * nv + ends[a].is_rc;
    }
    if (nu) ++nv;
    auto conj_v = [](uint64_t v) { return ((v - 3) ^ 1ull) + 3; };
    // outgoing lists: edge i (stored orientation) leaves vid(start of i)
    std::vector<std::vector<uint32_t>> out_of(2 * nv);
    for (uint64_t i = 0; i < nu; ++i) out_of[vid[2 * i] - 3].push_back((uint32_t)i);
    const std::string gpath = std::string(basename) + ".grseq", cpath = std::string(basename) + ".cvr";
    FILE *f = fopen(gpath.c_str(), "wb");
    BBK_REQUIRE(f != nullptr, BBK_ERR_IO, "cannot open %s for writing", gpath.c_str());
    bool ok = true;
    auto put64 = [&](uint64_t v) { ok = ok && fwrite(&v, 8, 1, f) == 1; };
    put64(3 + 2 * nv);  // reserved id ranges: every id handed out is below
    put64(3 + 2 * nu);
    put64(2 * nv);
    std::vector<uint64_t> words;
    for (uint64_t v = 0; v < 2 * nv && ok; ++v) {
        put64(3 + v);
        put64(conj_v(3 + v));
        for (uint32_t i : out_of[v]) {
            const char *s = u->bases.data() + u->offsets[i];
            const uint64_t len = u->offsets[i + 1] - u->offsets[i];
            // self-conjugate edge: s == rc(s), its conjugate is itself
            bool selfc = true;
            for (uint64_t a = 0; a < len && selfc; ++a) {
                const char c = s[len - 1 - a];
                selfc = s[a] == (c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A');
            }
            const uint64_t e1 = 3 + 2 * (uint64_t)i, e2 = selfc ? e1 : e1 + 1;
            put64(e1);
            put64(e2);
            put64(vid[2 * i + 1]);          // EdgeEnd(e1)
            put64(conj_v(vid[2 * i + 1]));  // EdgeStart(conj e1) = conjugate of EdgeEnd(e1)
            put64(len);
            words.assign((len + 31) / 32, 0);
            for (uint64_t a = 0; a < len; ++a) {
                const uint64_t c = s[a] == 'A' ? 0 : s[a] == 'C' ? 1 : s[a] == 'G' ? 2 : 3;
                words[a >> 5] |= c << ((a & 31) << 1);
            }
            ok = ok && (words.empty() || fwrite(words.data(), 8, words.size(), f) == words.size());
        }
        put64(0);
    }
    const int cl = fclose(f);
    BBK_REQUIRE(ok && cl == 0, BBK_ERR_IO, "short write to %s", gpath.c_str());
    f = fopen(cpath.c_str(), "wb");
    BBK_REQUIRE(f != nullptr, BBK_ERR_IO, "cannot open %s for writing", cpath.c_str());
    ok = true;
    for (uint64_t i = 0; i < nu && ok; ++i) {
        const uint64_t e1 = 3 + 2 * i;
        const uint64_t raw = u->has_cov ? u->kc[i] : 0;
        const uint32_t cov = raw > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)raw;
        ok = fwrite(&e1, 8, 1, f) == 1 && fwrite(&cov, 4, 1, f) == 1;
    }
    const uint64_t zero = 0;
    ok = ok && fwrite(&zero, 8, 1, f) == 1;
    const int cl2 = fclose(f);
    BBK_REQUIRE(ok && cl2 == 0, BBK_ERR_IO, "short write to %s", cpath.c_str());
}

int bbk_unitigs_write_spades(bbk_ctx *ctx, const bbk_unitigs *u, const char *basename) {
    return guarded([&] {
        BBK_REQUIRE(ctx && u && basename, BBK_ERR_ARG, "bbk_unitigs_write_spades: NULL argument");
        write_spades_graph(ctx, u, basename);
    });
}

void bbk_unitigs_free(bbk_unitigs *u) { delete u; }

}  // extern "C"
