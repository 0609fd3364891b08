// superk.hip -- stage A for wide k-mers (33 <= k <= 127: 16-, 24- and 32-byte keys): the duplicate-heavy canonical stream
// travels between the partition levels as SUPER-K-MER records instead of one key per k-mer.
//
// Why: at k = 55 a 150 bp read holds 96 k-mers = 1536 bytes of 16-byte keys, which the k-mer path (msd.hip) writes
// once and reads + writes once more before the in-LDS dedup reads them a third time -- 46 GB per 10 M reads, 64 % of
// the step.  Consecutive k-mers overlap in k-1 bases: a run of n consecutive k-mers is n+k-1 bases.  All k-mers of a
// run that share their MINIMIZER (the m-mer of the k-mer whose canonical form hashes lowest) go to the same bucket
// when buckets are chosen by the minimizer, and so do all other occurrences of those k-mers, on either strand: the
// canonical m-mers of a k-mer and of its reverse complement are the same set.  A record of RW = W+1 words holds up to
// 86 (118, 150) bases = a run of up to 32 k-mers at k = 55: ~9 records = 210 bytes per read instead of 1536.
//
//   k_sk_part1 : one lane per segment of C consecutive k-mers of one read.  Sliding-window minimum without divergence:
//                the window of w = k-m+1 m-mer positions of k-mer i is a suffix of block A = [s, s+w) and a prefix of
//                block B = [s+w, ...): one backward roll over A leaves the suffix minima in LDS, one forward roll over B
//                carries the prefix minimum in a register (two m-mer hashes per k-mer, no data-dependent loop).  Runs of
//                equal minimum become records (<= C k-mers, so they always fit), staged in LDS, then written to the
//                level-1 slot of their bin with one global atomic per (tile, bin).
//   k_sk_part2 : level 2 over the records of one level-1 slot: bin from further bits of the same hash, recomputed from
//                the m-mer the record points at; dense buckets from a histogram (<HIST> variant) + scan.
//   k_sk_dedup : one workgroup per bucket, streamed in chunks through LDS: one lane per k-mer INSTANCE extracts its
//                k-mer from the packed record, canonicalises it and looks it up in an open-addressing table whose
//                entries are pointers (record, offset, strand, 13-bit tag) into records kept in LDS -- 4 bytes per slot
//                whatever the key width; equal keys are recognised by comparing bit fields of the records.  Distinct
//                keys + reduced payload (count or OR of edge masks) are appended to the output through one global
//                atomic per bucket.
//
// The level-1 slots and the dedup tables are sized from estimates; what does not fit is not lost: records beyond a full
// level-1 slot go to a spill list and their buckets, like the buckets whose distinct keys exceed both table geometries
// (a low-complexity minimizer shared by thousands of k-mers), are expanded into one key per instance and deduplicated by
// the k-mer path (k_sk_expand + msd_sort_reduce), bucket by bucket.  Only when that share is large does the call return
// false and the caller (count.hip: dedup_reads) rerun the whole batch on the k-mer path.  Inputs above 3.8 M buckets run
// in passes over ranges of the minimizer hash; every pass re-reads the packed reads (0.4 GB per 10 M reads), never the
// k-mers.
// The output order is arbitrary (HASH semantics): stage B / the accumulator sort it.
// Environment: BBK_NO_SUPERK (A/B switch), BBK_SUPERK_MIN (instances below which the k-mer path is used; tests set 0),
// BBK_SUPERK_BUCKETS (tests: buckets per pass, forces several passes on small inputs), BBK_SUPERK_FILL (tests: planned
// instances of a bucket / table slots), BBK_SUPERK_SLOT_SCALE (tests: level-1 slots below their load),
// BBK_SUPERK_FALLBACK_MAX (share of a batch's instances the k-mer path may take over, default 0.25), BBK_VERBOSE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "bbk_internal.h"
#include "kmer_ops.h"
#include "msd.h"

namespace bbk {

namespace {

constexpr int kSkHdrBits = 19;   // n-1: 6 | minpos: 7 | has_prev: 1 | prev: 2 | has_next: 1 | next: 2
constexpr int kSkHdrShift = 64 - kSkHdrBits;
constexpr int kSk1NT = 256;               // level 1: lanes = segments of a tile
constexpr int kSk2NT = 256;               // level 2
constexpr int kSk2Items = 8;
constexpr int kSk2Tile = kSk2NT * kSk2Items;
// dedup: threads (= staged records of a chunk), table slots, records that introduced a key (the table's entries point
// into them).  First pass: four workgroups per CU (38 KB of LDS each at k = 55); buckets it gives up (table or KEEP
// area full: several times the planned distinct keys) go to the second-chance geometry, one workgroup per CU.
// Measured (10 M x 150 bp, k = 55, 11.8 ms): 2 ms of it without the lookups, 2 ms more with the k-mers prepared; the
// rest is the table walk (slot -> record behind the entry -> compare).  Issuing the four lookups of a work item together
// or walking them in lockstep was SLOWER (13.9 / 16.8 ms): the walk is bound by LDS accesses, not by their latency.
// Tried in round 3 and withdrawn: dropping a staged record that equals a kept one word for word before any of its
// k-mers is touched (a 1024-slot record table beside the k-mer table).  Slower (9.8 -> 11.1 ms at k = 55, 13.0 -> 13.9
// at k = 33): exact copies are rare, because records are cut at the boundaries of the read-relative segments of C
// k-mers a lane of k_sk_part1 owns, not only at minimizer changes -- two reads over the same stretch cut it differently.
struct SkdA {
    static constexpr int NT = 256, TS = 2048, KEEP = 384;
};
struct SkdB {
    static constexpr int NT = 512, TS = 8192, KEEP = 1024;  // 512 threads: 130 KB of LDS with 24-byte keys and 13 items per record
};
constexpr uint32_t kSkdFailCap = 1u << 16;  // buckets the second chance takes
constexpr uint32_t kSkdMaxProbes = 256;
// level 2 takes up to 4096 bins per slot: with more than ~1000 a tile of 2048 records leaves single records per bin
// (each its own 24-byte store), which costs far less than another pass over the reads (38 ms per 100 M reads)
constexpr uint32_t kSkMaxP1Bits = 10, kSkMaxP2 = 4096;

struct SkParams {
    uint32_t k, m, w, C;      // k-mer, minimizer, window (m-mers of a k-mer), k-mers of a segment
    uint32_t np, pass;        // passes over ranges of the bucket hash, this pass
    uint32_t b1bits, P1, P2;  // level-1 bins (power of two), level-2 bins of every level-1 slot
    uint32_t slot1;           // records of a level-1 slot
    uint32_t tps;             // level-2 tiles of a level-1 slot
    uint32_t spill_cap;       // records the level-1 spill list holds
    uint32_t xcd_tiles;       // level 2: tiles dealt to the XCDs by slot (k_sk_part2)
    // level 1: a slot is cut into 8 sub-slots of sub1 records, one per XCD, each with a cursor (cursor1[8 b + xcd]): the
    // (tile, bin) runs are one to four 24-byte records, and a 128-byte line that workgroups on different XCDs fill is
    // slow (msd.hip, level-1 call site).  0: one fill front per slot (slot1 = 8 * sub1 otherwise).
    uint32_t sub1;
    uint32_t tps_sub;         // level-2 tiles of a sub-slot (tps = 8 * tps_sub)
};

struct SkTile {
    uint32_t r0, nr;  // reads r0 .. r0+nr-1 own the segments of the tile
};

struct SkReads {
    const uint64_t *words;
    const uint64_t *woff;
    const uint32_t *len;
    const uint64_t *coff;  // exclusive scan of segments per read (n_reads + 1)
    const SkTile *tiles;
    uint64_t n_reads;
    uint64_t n_segs;
};

enum {
    SKF_SLOT1 = 0,   // too many bins of one tile ran over their level-1 slot, or the spill list is full: give up
    SKF_NFAIL = 1,   // buckets the first table gave up (listed)
    SKF_TABLE = 2,   // more buckets for the k-mer path than its list holds: give up
    SKF_STAGE = 3,
    SKF_OUT = 4,     // output buffer too small (regrown, pass repeated)
    SKF_SELECT = 5,
    SKF_MAXREC = 6,
    SKF_NSPILL = 7,  // records that did not fit their level-1 slot (spill list)
    SKF_NFAIL2 = 8   // buckets left to the k-mer path (hot: a spilled record belongs to them; or both tables gave up)
};

// hash of a canonical m-mer (F forward, R reverse complement, both right-aligned 2m bits)
__device__ __forceinline__ uint32_t sk_mhash(uint64_t F, uint64_t R) {
    const uint64_t c = F < R ? F : R;
    uint32_t h = ((uint32_t)c * 0x9E3779B1u) ^ ((uint32_t)(c >> 32) * 0x85EBCA6Bu);
    h ^= h >> 15;
    h *= 0xC2B2AE35u;
    h ^= h >> 13;
    return h;
}
// bucket hash of a minimizer: the MINIMUM of ~w hashes crowds near zero, so it is mixed again (bijective)
__device__ __forceinline__ uint32_t sk_g(uint32_t h) {
    uint32_t g = h * 0x9E3779B1u;
    g ^= g >> 16;
    g *= 0x85EBCA6Bu;
    g ^= g >> 13;
    g *= 0x27D4EB2Fu;
    g ^= g >> 15;
    return g;
}
__device__ __forceinline__ bool sk_select(uint32_t g, const SkParams &P, uint32_t &gp) {
    if (P.np <= 1) {
        gp = g;
        return true;
    }
    const uint64_t t = (uint64_t)g * P.np;
    gp = (uint32_t)t;
    return (uint32_t)(t >> 32) == P.pass;
}
__device__ __forceinline__ uint32_t sk_bin1(uint32_t gp, const SkParams &P) { return P.b1bits ? gp >> (32 - P.b1bits) : 0u; }
__device__ __forceinline__ uint32_t sk_bin2(uint32_t gp, const SkParams &P) {
    return __umulhi(P.b1bits ? gp << P.b1bits : gp, P.P2);
}

// 32 bases of the packed read from base p on (words past lastw are not touched; bits past the read are garbage
// or zero: callers mask)
__device__ __forceinline__ uint64_t sk_bases(const uint64_t *__restrict__ rw, uint32_t p, uint32_t lastw) {
    const uint32_t wi = p >> 5;
    const uint32_t sh = (p & 31u) << 1;
    const uint64_t lo = wi <= lastw ? rw[wi] : 0ull;
    const uint64_t hi = wi + 1u <= lastw ? rw[wi + 1u] : 0ull;
    return (lo >> sh) | ((hi << 1) << (63u - sh));
}

// one base at a time from a read, walking forwards or backwards: the current word stays in a register
struct SkBaseWalk {
    const uint64_t *rw;
    uint32_t wi;
    uint64_t cur;
    __device__ __forceinline__ void init(const uint64_t *r) {
        rw = r;
        wi = 0xFFFFFFFFu;
        cur = 0;
    }
    __device__ __forceinline__ uint32_t at(uint32_t p) {
        const uint32_t x = p >> 5;
        if (x != wi) {
            wi = x;
            cur = rw[x];
        }
        return (uint32_t)(cur >> ((p & 31u) << 1)) & 3u;
    }
};

__device__ inline uint64_t sk_read_at(const uint64_t *__restrict__ off, uint64_t lo, uint64_t hi, uint64_t j) {
    while (hi - lo > 1) {  // largest r in [lo, hi) with off[r] <= j
        const uint64_t mid = (lo + hi) >> 1;
        if (off[mid] <= j) lo = mid;
        else hi = mid;
    }
    return lo;
}

__global__ void k_sk_segments(const uint32_t *__restrict__ len, uint64_t n, uint32_t k, uint32_t C,
                              uint64_t *__restrict__ nk, uint64_t *__restrict__ nseg) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const uint32_t L = len[i];
        const uint64_t c = L >= k ? (uint64_t)(L - k + 1) : 0ull;
        nk[i] = c;
        nseg[i] = (c + C - 1) / C;
    }
}

__global__ void k_sk_tiles(const uint64_t *__restrict__ coff, uint64_t n_reads, uint64_t n_tiles, uint32_t tile,
                           uint64_t n_segs, SkTile *__restrict__ out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const uint64_t c0 = t * (uint64_t)tile;
    const uint64_t c1 = (c0 + tile < n_segs ? c0 + tile : n_segs) - 1;
    const uint64_t r0 = sk_read_at(coff, 0, n_reads, c0), r1 = sk_read_at(coff, 0, n_reads, c1);
    out[t] = SkTile{(uint32_t)r0, (uint32_t)(r1 - r0 + 1)};
}

// word wi of a record held in registers (no dynamic indexing: that would put the array in scratch)
template <int RW>
__device__ __forceinline__ uint64_t sk_word(const uint64_t (&rec)[RW], uint32_t wi) {
    uint64_t v = 0;
#pragma unroll
    for (int j = 0; j < RW; ++j) v = (wi == (uint32_t)j) ? rec[j] : v;
    return v;
}

// hash of the minimizer a record points at
template <int RW>
__device__ __forceinline__ uint32_t sk_rec_hash(const uint64_t (&rec)[RW], uint32_t m) {
    const uint32_t hdr = (uint32_t)(rec[RW - 1] >> kSkHdrShift);
    const uint32_t mp = (hdr >> 6) & 127u;
    const uint32_t wi = mp >> 5, sh = (mp & 31u) << 1;
    const uint64_t lo = sk_word<RW>(rec, wi), hi = sk_word<RW>(rec, wi + 1u);
    const uint64_t mmask = (1ull << (2u * m)) - 1ull;
    const uint64_t F = ((lo >> sh) | ((hi << 1) << (63u - sh))) & mmask;
    const uint64_t R = (~rev2(F)) >> (64u - 2u * m);
    return sk_mhash(F, R);
}

static size_t sk_part1_smem(uint32_t C, uint32_t P1, int RW) {
    (void)RW;
    return (size_t)C * kSk1NT * 5 + (size_t)P1 * 8;
}

// ---- level 1: reads -> super-k-mer records in level-1 slots -----------------------------------------------------
// The bucket comes from the minimum HASH alone (two m-mers with equal hashes may swap roles between the strands).  It
// has to be the full 32 bits: minima crowd in the lowest 1/w of the hash range, and with fewer distinct values than
// minimizer positions in the genome several positions share a value and the bucket sizes spread out.
template <int RW>
__global__ __launch_bounds__(kSk1NT) void k_sk_part1(SkReads S, SkParams P, uint32_t *__restrict__ cursor1,
                                                     uint64_t *__restrict__ out, uint64_t *__restrict__ spill,
                                                     uint32_t *__restrict__ flags) {
    constexpr int NT = kSk1NT;
    // bins of this tile whose run does not fit the rest of their slot (a hot minimizer): {base, records that fit,
    // position in the spill list}; their gbase entry is 0x80000000 | index
    __shared__ uint32_t ovf[64][3];
    __shared__ uint32_t ovf_n;
    if (threadIdx.x == 0) ovf_n = 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // layout: sufA[C][NT] u32 (suffix minima, then one word per record: bin | rank) | lhist[P1] | gbase[P1] |
    //         posA[C][NT] u8 (position of the minimum relative to the segment, then relative to the record)
    uint32_t *sufA = reinterpret_cast<uint32_t *>(smem);
    uint32_t *lhist = sufA + (size_t)P.C * NT;
    uint32_t *gbase = lhist + P.P1;
    uint8_t *posA = reinterpret_cast<uint8_t *>(gbase + P.P1);

    const uint32_t tid = threadIdx.x;
    const uint32_t k = P.k, m = P.m, w = P.w;
    for (uint32_t b = tid; b < P.P1; b += NT) lhist[b] = 0;
    __syncthreads();

    const uint64_t c = (uint64_t)blockIdx.x * NT + tid;
    const bool active = c < S.n_segs;
    uint64_t bmask = 0;  // bit i: k-mer s+i starts a record
    uint32_t s = 0, cnt = 0, len = 0, lastw = 0;
    const uint64_t *rw = S.words;
    if (active) {
        const SkTile T = S.tiles[blockIdx.x];
        const uint64_t r = sk_read_at(S.coff, T.r0, (uint64_t)T.r0 + T.nr, c);
        len = S.len[r];
        s = (uint32_t)(c - S.coff[r]) * P.C;  // first k-mer of the segment
        const uint32_t nk = len - k + 1u;     // the read owns a segment, so len >= k
        cnt = nk - s < P.C ? nk - s : P.C;
        rw = S.words + S.woff[r];
        lastw = (len - 1u) >> 5;
        const uint64_t mmask = (1ull << (2u * m)) - 1ull;  // m <= 31
        const uint32_t mtop = 2u * m - 2u;
        SkBaseWalk B;
        B.init(rw);

        // backward over block A = m-mer positions s+w-1 .. s: suffix minima
        uint64_t F = sk_bases(rw, s + w - 1u, lastw) & mmask;
        uint64_t R = (~rev2(F)) >> (64u - 2u * m);
        const uint64_t F_e = F, R_e = R;
        uint32_t sm = 0xFFFFFFFFu, sp = 0;
        for (int j = (int)w - 1; j >= 0; --j) {
            if (j < (int)w - 1) {
                const uint64_t b = B.at(s + (uint32_t)j);
                F = ((F << 2) | b) & mmask;
                R = (R >> 2) | ((3ull - b) << mtop);
            }
            const uint32_t h = sk_mhash(F, R);
            if (h <= sm) {
                sm = h;
                sp = (uint32_t)j;
            }
            if ((uint32_t)j < cnt) {
                sufA[(uint32_t)j * NT + tid] = sm;
                posA[(uint32_t)j * NT + tid] = (uint8_t)sp;
            }
        }
        // forward: k-mer s+i has the window [i, i+w-1] = suffix of A from i + prefix of B up to i+w-1
        F = F_e;
        R = R_e;
        uint32_t pre = 0xFFFFFFFFu, pp = 0, prevcur = 0;
        for (uint32_t i = 0; i < cnt; ++i) {
            if (i >= 1) {
                const uint64_t b = B.at(s + i + w - 1u + m - 1u);  // last base of the m-mer at s+i+w-1
                F = (F >> 2) | (b << mtop);
                R = ((R << 2) | (3ull - b)) & mmask;
                const uint32_t h = sk_mhash(F, R);
                if (h < pre) {
                    pre = h;
                    pp = i + w - 1u;
                }
            }
            const uint32_t a = sufA[i * NT + tid];
            const bool fromA = a <= pre;
            const uint32_t cur = fromA ? a : pre;
            const bool boundary = i == 0 || cur != prevcur;
            bmask |= (uint64_t)(boundary ? 1u : 0u) << i;
            if (boundary) {
                sufA[i * NT + tid] = cur;
                if (!fromA) posA[i * NT + tid] = (uint8_t)pp;
            }
            prevcur = cur;
        }
        // one record per run of equal minimum: its level-1 bin and its rank among the tile's records of that bin
        for (uint64_t bm = bmask; bm;) {
            const uint32_t i0 = (uint32_t)__ffsll((unsigned long long)bm) - 1u;
            bm &= bm - 1ull;
            const uint32_t cur = sufA[i0 * NT + tid];
            uint32_t gp, info = 0xFFFFFFFFu;
            if (sk_select(sk_g(cur), P, gp)) {
                const uint32_t b1 = sk_bin1(gp, P);
                const uint32_t rank = atomicAdd(&lhist[b1], 1u);  // < NT * C <= 16384
                info = (b1 << 16) | rank;
            }
            sufA[i0 * NT + tid] = info;
        }
    }
    __syncthreads();
    uint32_t xcc = 0;  // the XCD this workgroup runs on (placement only: any value gives a correct result)
    if (P.sub1) {
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
    }
    const uint32_t cap1 = P.sub1 ? P.sub1 : P.slot1;  // records of the (sub-)slot this workgroup fills
    const uint32_t sub_off = xcc * P.sub1;            // its place inside the slot
    for (uint32_t b = tid; b < P.P1; b += NT) {
        const uint32_t cb = lhist[b];
        uint32_t g = 0xFFFFFFFFu;
        if (cb) {
            const uint32_t base = atomicAdd(&cursor1[P.sub1 ? b * 8u + xcc : b], cb);
            if (base + cb <= cap1) {
                g = base;
            } else {  // what fits goes to the slot, the rest to the spill list
                const uint32_t fit = base >= cap1 ? 0u : cap1 - base;
                const uint32_t sb = atomicAdd(&flags[SKF_NSPILL], cb - fit);
                const uint32_t idx = atomicAdd(&ovf_n, 1u);
                if (idx < 64u && sb + (cb - fit) <= P.spill_cap) {
                    ovf[idx][0] = base;
                    ovf[idx][1] = fit;
                    ovf[idx][2] = sb;
                    g = 0x80000000u | idx;
                } else {
                    flags[SKF_SLOT1] = 1u;
                }
            }
        }
        gbase[b] = g;
    }
    __syncthreads();
    while (bmask) {
        const uint32_t i0 = (uint32_t)__ffsll((unsigned long long)bmask) - 1u;
        bmask &= bmask - 1ull;
        const uint32_t i1 = bmask ? (uint32_t)__ffsll((unsigned long long)bmask) - 1u : cnt;
        const uint32_t info = sufA[i0 * NT + tid];
        if (info == 0xFFFFFFFFu) continue;
        const uint32_t b1 = info >> 16, rank = info & 0xFFFFu;
        const uint32_t mp = (uint32_t)posA[i0 * NT + tid] - i0;  // the minimizer lies inside k-mer s+i0
        const uint32_t g = gbase[b1];
        if (g == 0xFFFFFFFFu) continue;
        const uint32_t n = i1 - i0;
        const uint32_t a = s + i0, nb = n + k - 1u;
        uint64_t rec[RW];
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            const int vb = 2 * (int)nb - 64 * j;  // populated bits of word j
            uint64_t v = vb > 0 ? sk_bases(rw, a + 32u * (uint32_t)j, lastw) : 0ull;
            if (vb > 0 && vb < 64) v &= (1ull << vb) - 1ull;
            rec[j] = v;
        }
        uint32_t hdr = (n - 1u) | (mp << 6);
        if (a > 0) hdr |= (1u << 13) | (base_at(rw, a - 1u) << 14);
        if (a + nb < len) hdr |= (1u << 16) | (base_at(rw, a + nb) << 17);
        rec[RW - 1] |= (uint64_t)hdr << kSkHdrShift;
        uint64_t *dst;
        if (g & 0x80000000u) {
            const uint32_t *o = ovf[g & 63u];
            dst = rank < o[1] ? out + ((uint64_t)b1 * P.slot1 + sub_off + o[0] + rank) * RW
                              : spill + ((uint64_t)o[2] + (rank - o[1])) * RW;
        } else {
            dst = out + ((uint64_t)b1 * P.slot1 + sub_off + g + rank) * RW;
        }
#pragma unroll
        for (int j = 0; j < RW; ++j) dst[j] = rec[j];
    }
}

// the buckets spilled records belong to: all their records must take the k-mer path together
template <int RW>
__global__ __launch_bounds__(256) void k_sk_mark_hot(const uint64_t *__restrict__ spill, uint32_t n, SkParams P,
                                                     uint8_t *__restrict__ hot, uint32_t *__restrict__ flags) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t rec[RW];
#pragma unroll
    for (int j = 0; j < RW; ++j) rec[j] = spill[(uint64_t)i * RW + j];
    uint32_t gp;
    if (!sk_select(sk_g(sk_rec_hash<RW>(rec, P.m)), P, gp)) {
        flags[SKF_SELECT] = 1u;
        return;
    }
    hot[(size_t)sk_bin1(gp, P) * P.P2 + sk_bin2(gp, P)] = 1;
}

// ---- level 2: records of a level-1 slot -> dense buckets (exact: histogram, scan, scatter) ------------------------
// The records of one minimizer cannot be split, and at 50x coverage one genomic position of a minimizer yields ~100
// records: bucket sizes vary far too much for fixed slots.  The buckets are therefore laid out densely from a
// histogram, and the dedup kernel streams a bucket of any size.
template <int RW, bool HIST>
__global__ __launch_bounds__(kSk2NT) void k_sk_part2(const uint64_t *__restrict__ in, SkParams P,
                                                     const uint32_t *__restrict__ cursor1,
                                                     unsigned long long *__restrict__ cursor2, uint64_t *__restrict__ out,
                                                     uint32_t *__restrict__ flags) {
    constexpr int NT = kSk2NT, ITEMS = kSk2Items;
    __shared__ uint32_t lhist[kSkMaxP2];
    __shared__ unsigned long long gbase[HIST ? 1 : kSkMaxP2];
    const uint32_t tid = threadIdx.x;
    // Workgroup w runs on XCD w % 8: the tiles of slot b1 all go to XCD b1 % 8, so that the buckets of a slot are filled
    // through ONE L2 (a line that several XCDs fill is slow: see the level-1 call site in msd.hip).  Grid: 8 *
    // ceil(P1 / 8) * tps; BBK_XCD_TILES=0 (P.xcd_tiles = 0): slot-major order.
    uint32_t b1, t;
    if (P.xcd_tiles) {
        const uint32_t j = blockIdx.x >> 3;
        b1 = (j / P.tps) * 8u + (blockIdx.x & 7u);
        t = j % P.tps;
        if (b1 >= P.P1) return;
    } else {
        b1 = blockIdx.x / P.tps;
        t = blockIdx.x % P.tps;
    }
    // tile t of the slot: tile t % tps_sub of sub-slot t / tps_sub when level 1 filled one sub-slot per XCD
    const uint32_t sx = P.sub1 ? t / P.tps_sub : 0u, tt = P.sub1 ? t % P.tps_sub : t;
    const uint32_t cap1 = P.sub1 ? P.sub1 : P.slot1;
    const uint32_t have = cursor1[P.sub1 ? b1 * 8u + sx : b1];
    const uint32_t cnt = have < cap1 ? have : cap1;
    if (tt * (uint32_t)kSk2Tile >= cnt) return;
    const uint32_t count = cnt - tt * (uint32_t)kSk2Tile < (uint32_t)kSk2Tile ? cnt - tt * (uint32_t)kSk2Tile : (uint32_t)kSk2Tile;
    const uint32_t begin = sx * P.sub1 + tt * (uint32_t)kSk2Tile;  // first record of the tile inside the slot
    for (uint32_t b = tid; b < P.P2; b += NT) lhist[b] = 0;
    __syncthreads();
    const uint64_t *src = in + ((uint64_t)b1 * P.slot1 + begin) * RW;
    uint64_t rec[ITEMS][RW];
    uint32_t binrank[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t p = (uint32_t)i * NT + tid;
        binrank[i] = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < RW; ++j) rec[i][j] = p < count ? src[(size_t)p * RW + j] : 0ull;
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t p = (uint32_t)i * NT + tid;
        if (p < count) {
            uint32_t gp;
            if (!sk_select(sk_g(sk_rec_hash<RW>(rec[i], P.m)), P, gp) || sk_bin1(gp, P) != b1) {
                flags[SKF_SELECT] = 1u;  // cannot happen: level 1 put the record here by the same hash
                continue;
            }
            const uint32_t b2 = sk_bin2(gp, P);
            binrank[i] = (b2 << 16) | atomicAdd(&lhist[b2], 1u);
        }
    }
    __syncthreads();
    for (uint32_t b = tid; b < P.P2; b += NT) {
        const uint32_t cb = lhist[b];
        if (cb) {
            const unsigned long long base = atomicAdd(&cursor2[(size_t)b1 * P.P2 + b], (unsigned long long)cb);
            if (!HIST) gbase[b] = base;
        }
    }
    if (HIST) return;
    __syncthreads();
    // (the records are awaited here by every lane: left to the compiler, the wait for the predicated loads -- vmcnt 0 --
    // lands in the conditional blocks of the store loop and makes every store wait for the one before)
#pragma unroll
    for (int i = 0; i < ITEMS; ++i)
#pragma unroll
        for (int j = 0; j < RW; ++j) asm volatile("" : "+v"(rec[i][j]));
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        if (binrank[i] == 0xFFFFFFFFu) continue;
        const uint32_t b2 = binrank[i] >> 16, rank = binrank[i] & 0xFFFFu;
        const uint64_t dst = ((uint64_t)gbase[b2] + rank) * RW;
#pragma unroll
        for (int j = 0; j < RW; ++j) out[dst + j] = rec[i][j];
    }
}

// ---- dedup: one bucket of records -> distinct canonical k-mers + reduced payload ---------------------------------
// k-mer starting at base j of a record in LDS (RW words at rp)
template <int W, int RW>
__device__ __forceinline__ Key<W> sk_kmer_at(const uint64_t *rp, uint32_t j, uint32_t k) {
    // the bit field starts at bit 2j: whole dwords by addressing, the rest by 32-bit funnel shifts (v_alignbit_b32);
    // 64-bit shifts by a register amount run at a quarter of the rate and there were four per extraction
    const uint32_t *rd = reinterpret_cast<const uint32_t *>(rp);
    const uint32_t d0 = j >> 4, sh = (j & 15u) << 1;  // dword offset, bit shift 0..30
    uint32_t x[2 * W + 1];
#pragma unroll
    for (int i = 0; i <= 2 * W; ++i) x[i] = (d0 + (uint32_t)i < (uint32_t)(2 * RW)) ? rd[d0 + (uint32_t)i] : 0u;
    Key<W> r;
#pragma unroll
    for (int i = 0; i < W; ++i) {
        const uint32_t lo = __builtin_amdgcn_alignbit(x[2 * i + 1], x[2 * i], sh);
        const uint32_t hi = __builtin_amdgcn_alignbit(x[2 * i + 2], x[2 * i + 1], sh);
        r.w[i] = ((uint64_t)hi << 32) | lo;
    }
    const uint32_t vb = 2u * k - 64u * (uint32_t)(W - 1);  // populated bits of the last word (2..64)
    if (vb < 64u) r.w[W - 1] &= (1ull << vb) - 1ull;
    return r;
}
__device__ __forceinline__ uint32_t sk_base_at(const uint64_t *rp, uint32_t p) {
    return (reinterpret_cast<const uint32_t *>(rp)[p >> 4] >> ((p & 15u) << 1)) & 3u;
}

// append base c (drop base 0) to F and prepend its complement to RC (drop RC's last base)
template <int W>
__device__ __forceinline__ void sk_roll(Key<W> &F, Key<W> &RC, uint32_t k, uint32_t c) {
    F = kmer_shl<W>(F, (int)k, c);
#pragma unroll
    for (int i = W - 1; i > 0; --i) RC.w[i] = (RC.w[i] << 2) | (RC.w[i - 1] >> 62);
    RC.w[0] = (RC.w[0] << 2) | (uint64_t)(3u - c);
    const uint32_t vb = 2u * k - 64u * (uint32_t)(W - 1);
    if (vb < 64u) RC.w[W - 1] &= (1ull << vb) - 1ull;
}

#ifndef BBK_SKD_SR
#define BBK_SKD_SR 4
#endif
constexpr int kSkdSR = BBK_SKD_SR;  // k-mers of a work item (consecutive k-mers of one record, rolled)

// F <= RC in base order (base 0 most significant, rtseq.hpp:732-741): decided by the first base in which they differ
template <int W>
__device__ __forceinline__ bool sk_minimal(const Key<W> &F, const Key<W> &RC) {
    uint64_t x = 0, f = 0, r = 0;
#pragma unroll
    for (int i = W - 1; i >= 0; --i) {
        const uint64_t d = F.w[i] ^ RC.w[i];
        x = d ? d : x;
        f = d ? F.w[i] : f;
        r = d ? RC.w[i] : r;
    }
    const uint64_t t = x & (~x + 1ull);                                // lowest differing bit
    const uint64_t lo = (t & 0x5555555555555555ull) ? t : (t >> 1);    // low bit of its base
    const uint64_t pm = lo | (lo << 1);
    return (f & pm) <= (r & pm);                                       // x == 0: equal, minimal
}

// table hash of a canonical k-mer: slot from the low bits, tag from the high bits
template <int W>
__device__ __forceinline__ uint32_t sk_khash(const Key<W> &x) {
    uint32_t h = 0x9E3779B9u;
#pragma unroll
    for (int i = 0; i < W; ++i) {
        h = (h ^ (uint32_t)x.w[i]) * 0x9E3779B1u;
        h = (h ^ (uint32_t)(x.w[i] >> 32)) * 0x85EBCA6Bu;
        h ^= h >> 15;
    }
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

// 32 bases of a record in LDS from base p on (bits past the record's last word read as zero)
template <int RW>
__device__ __forceinline__ uint64_t sk_rec_bases(const uint64_t *rp, uint32_t p) {
    const uint32_t *rd = reinterpret_cast<const uint32_t *>(rp);
    const uint32_t d0 = p >> 4, sh = (p & 15u) << 1;
    uint32_t x[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) x[i] = (d0 + (uint32_t)i < (uint32_t)(2 * RW)) ? rd[d0 + (uint32_t)i] : 0u;
    return ((uint64_t)__builtin_amdgcn_alignbit(x[2], x[1], sh) << 32) | __builtin_amdgcn_alignbit(x[1], x[0], sh);
}

// a bucket this geometry cannot finish is listed: first geometry -> counter SKF_NFAIL (second chance); second geometry
// -> counter SKF_NFAIL2 (the k-mer path)
__device__ __forceinline__ void sk_give_up(uint32_t *flags, uint32_t *fail_list, uint32_t fail_ctr, uint32_t b, uint32_t nrec) {
    const uint32_t at = atomicAdd(&flags[fail_ctr], 1u);
    if (at < kSkdFailCap) fail_list[at] = b;
    else if (fail_ctr == SKF_NFAIL2) flags[SKF_TABLE] = 1u;
    atomicMax(&flags[SKF_MAXREC], nrec);
}

// LDS: recs[KEEP + STG][RW] u64 | ioff[STG + 2] | newidx[STG] | table[TS] | tvals[TS] (OP) | tmp[64] | work[STG * ipr] u16
template <int W, int OP, class G>
static size_t sk_dedup_smem(uint32_t C) {
    const uint32_t ipr = (C + kSkdSR - 1) / kSkdSR;
    return (size_t)(G::KEEP + G::NT) * (W + 1) * 8 + (size_t)(G::NT + 2) * 4 + (size_t)G::NT * 4 + (size_t)G::TS * 4 +
           (OP ? (size_t)G::TS * 4 : 0) + 64 * 4 + (size_t)G::NT * ipr * 2 + 16;
}

// The bucket is streamed in chunks of STG records through a staging area.  A lane takes a work item = up to 4
// consecutive k-mers of one record: one bit-field extraction + reverse complement, then rolled base by base; every
// k-mer is canonicalised and looked up in an open-addressing table whose entries are pointers (record, offset,
// strand, 13-bit tag): equal keys are recognised by comparing bit fields of records.  A record that introduced a new
// key is moved to the KEEP area when its chunk is done (and the entries that point at it are redirected); at 10x
// coverage nine records in ten introduce nothing and are dropped.
template <int W, int OP, class GEO>  // OP 0: keys only, 1: multiplicity, 3: OR of the edge masks
__global__ __launch_bounds__(GEO::NT) void k_sk_dedup(const uint64_t *__restrict__ records,
                                                    const unsigned long long *__restrict__ boff, SkParams P,
                                                    Key<W> *__restrict__ out_keys, uint32_t *__restrict__ out_vals,
                                                    unsigned long long *__restrict__ out_cursor,
                                                    unsigned long long out_cap, uint32_t *__restrict__ flags,
                                                    const uint32_t *__restrict__ bucket_ids,
                                                    uint32_t *__restrict__ fail_list, uint32_t fail_ctr,
                                                    const uint8_t *__restrict__ hot) {
    constexpr int RW = W + 1, NT = GEO::NT, TS = GEO::TS, KEEP = GEO::KEEP, STG = GEO::NT, SR = kSkdSR;
    constexpr uint32_t EMPTY = 0xFFFFFFFFu;
    static_assert(KEEP + STG <= 2048 && STG <= 4096, "entry / work item bit fields");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t *recs = reinterpret_cast<uint64_t *>(smem);                     // [KEEP + STG][RW]
    uint32_t *ioff = reinterpret_cast<uint32_t *>(recs + (KEEP + STG) * RW);  // [STG + 2] exclusive scan of items
    uint32_t *newidx = ioff + STG + 2;                                        // [STG] 0: not referenced; else flag / new index
    uint32_t *table = newidx + STG;                                           // [TS]
    uint32_t *tvals = table + TS;                                             // [TS] (OP)
    uint32_t *tmp = tvals + (OP ? TS : 0);                                    // [64]
    uint16_t *work = reinterpret_cast<uint16_t *>(tmp + 64);                  // [STG * ipr]: record << 5 | item of the record
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t b = bucket_ids ? bucket_ids[blockIdx.x] : blockIdx.x;
    const unsigned long long r_begin = boff[b], r_end = boff[b + 1];
    if (hot && hot[b]) {  // a record of this bucket sits in the spill list (its slot may even be empty here)
        if (tid == 0) sk_give_up(flags, fail_list, fail_ctr, b, (uint32_t)(r_end - r_begin));
        return;
    }
    if (r_begin == r_end) return;
    const uint32_t k = P.k;
    for (uint32_t i = tid; i < (uint32_t)TS; i += NT) {
        table[i] = EMPTY;
        if (OP) tvals[i] = 0;
    }
    uint32_t keep = 0;  // records in the KEEP area (uniform)
    bool failed = false;
    for (unsigned long long c0 = r_begin; c0 < r_end; c0 += STG) {
        const uint32_t cnt = r_end - c0 < (unsigned long long)STG ? (uint32_t)(r_end - c0) : (uint32_t)STG;
        const uint64_t *src = records + c0 * RW;
        uint64_t *stg = recs + KEEP * RW;
        for (uint32_t i = tid; i < cnt * RW; i += NT) stg[i] = src[i];
        newidx[tid] = 0;
        __syncthreads();
        {  // work items before every staged record
            const uint32_t n = tid < cnt ? ((uint32_t)(stg[tid * RW + RW - 1] >> kSkHdrShift) & 63u) + 1u : 0u;
            const uint32_t items = (n + SR - 1) / SR;
            uint32_t incl = items;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t t = __shfl_up(incl, d, 64);
                if (lane >= (uint32_t)d) incl += t;
            }
            if (lane == 63) tmp[wave] = incl;
            __syncthreads();
            uint32_t before = 0;
            for (uint32_t j = 0; j < wave; ++j) before += tmp[j];
            const uint32_t first = before + incl - items;
            for (uint32_t q = 0; q < items; ++q) work[first + q] = (uint16_t)((tid << 5) | q);
            if (tid == NT - 1) ioff[STG] = before + incl;
        }
        __syncthreads();
        const uint32_t total = ioff[STG];
        for (uint32_t it = tid; it < total; it += NT) {
            const uint32_t wk = work[it];
            const uint32_t r = wk >> 5, j0 = (wk & 31u) * SR;
            const uint64_t *rp = stg + r * RW;
            const uint32_t hdr = (uint32_t)(rp[RW - 1] >> kSkHdrShift);
            const uint32_t n = (hdr & 63u) + 1u;
            const uint32_t cq = n - j0 < (uint32_t)SR ? n - j0 : (uint32_t)SR;
            Key<W> F = sk_kmer_at<W, RW>(rp, j0, k);
            Key<W> RC = kmer_rc<W>(F, (int)k);
            const uint32_t inb = (uint32_t)sk_rec_bases<RW>(rp, j0 + k);  // bases j0+k, j0+k+1, ...: the ones that enter
            uint32_t prevc = j0 > 0 ? sk_base_at(rp, j0 - 1u) : (hdr >> 14) & 3u;
            for (uint32_t u = 0; u < cq; ++u) {
                const uint32_t j = j0 + u;
                if (u) {
                    prevc = (uint32_t)F.w[0] & 3u;
                    sk_roll<W>(F, RC, k, (inb >> (2u * (u - 1u))) & 3u);
                }
                const bool minimal = sk_minimal<W>(F, RC);
                const Key<W> X = key_select<W>(minimal, F, RC);
                uint32_t val = 1;
                if (OP == 3) {
                    const bool hp = j > 0 || ((hdr >> 13) & 1u), hn = j + 1u < n || ((hdr >> 16) & 1u);
                    const uint32_t nextc = j + 1u < n ? (inb >> (2u * u)) & 3u : (hdr >> 17) & 3u;
                    val = 0;
                    if (hn) val |= 1u << (minimal ? nextc : 7u - nextc);
                    if (hp) val |= 1u << (minimal ? 4u + prevc : 3u - prevc);
                }
                const uint32_t h = sk_khash<W>(X);
                const uint32_t tag = (h >> 19) & 0x1FFFu;
                const uint32_t strand = minimal ? 0u : 1u;
                // entry: tag 13 | record index 11 (KEEP area, then staging) | offset 6 | strand 1; bit 31 clear
                const uint32_t entry = (tag << 18) | (((uint32_t)KEEP + r) << 7) | (j << 1) | strand;
                uint32_t slot = h & (uint32_t)(TS - 1);
                bool done = false;
                // Two nested loops: the inner one only walks to the first slot that is empty or carries my tag (a load
                // and two compares per step); cutting the entry's k-mer out of its record and comparing 2W words happens
                // in the outer loop, which nearly every lane leaves after one round (a 13-bit tag lets 1 in 8192 foreign
                // entries through).  Worth 3 % (10.1 -> 9.8 ms at 10 M reads, k = 55), not the 40 % round 2 expected from
                // "the divergent probe loop": the ISA of the loop body holds ~250 vector instructions per k-mer of
                // STRAIGHT-LINE code (128-bit roll, base-order comparison, hash, two bit-field extractions) -- the
                // instruction count measured with the SQ counters is the body, not the divergence.
                for (uint32_t p = 0; p < kSkdMaxProbes && !done;) {
                    uint32_t e = table[slot];
                    while (e != EMPTY && (e >> 18) != tag && ++p < kSkdMaxProbes) {
                        slot = (slot + 1u) & (uint32_t)(TS - 1);
                        e = table[slot];
                    }
                    if (p >= kSkdMaxProbes) break;
                    if (e == EMPTY) {
                        e = atomicCAS(&table[slot], EMPTY, entry);
                        if (e == EMPTY) {
                            newidx[r] = 1u;  // this record must outlive its chunk
                            done = true;
                            break;
                        }
                    }
                    if ((e >> 18) == tag) {
                        // same canonical k-mer <=> the entry's forward bits equal my forward bits (same strand) or my
                        // reverse complement (opposite strands)
                        const Key<W> G = sk_kmer_at<W, RW>(recs + ((e >> 7) & 2047u) * RW, (e >> 1) & 63u, k);
                        if (key_eq<W>(G, key_select<W>((e & 1u) == strand, F, RC))) {
                            done = true;
                            break;
                        }
                    }
                    slot = (slot + 1u) & (uint32_t)(TS - 1);
                    ++p;
                }
                if (done) {
                    if (OP == 1) atomicAdd(&tvals[slot], val);
                    if (OP == 3) atomicOr(&tvals[slot], val);
                } else {
                    failed = true;
                }
            }
        }
        __syncthreads();
        // promotion: referenced staged records move to the KEEP area
        {
            const uint32_t f = newidx[tid] ? 1u : 0u;
            uint32_t incl = f;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t t = __shfl_up(incl, d, 64);
                if (lane >= (uint32_t)d) incl += t;
            }
            if (lane == 63) tmp[16 + wave] = incl;
            __syncthreads();
            uint32_t before = 0, tot = 0;
            for (uint32_t j = 0; j < (uint32_t)(NT / 64); ++j) {
                const uint32_t x = tmp[16 + j];
                if (j < wave) before += x;
                tot += x;
            }
            if (keep + tot > (uint32_t)KEEP) {
                failed = true;  // uniform; nothing is moved, the bucket is given up below
                tot = 0;
            } else if (f) {
                const uint32_t ni = keep + before + incl - 1u;
                newidx[tid] = ni + 1u;  // + 1: 0 stays "not referenced"
#pragma unroll
                for (int j = 0; j < RW; ++j) recs[ni * RW + j] = stg[tid * RW + j];
            }
            __syncthreads();
            if (tot) {
                constexpr int SPT = TS / NT;
#pragma unroll
                for (int q = 0; q < SPT; ++q) {
                    const uint32_t e = table[tid * SPT + q];
                    const uint32_t ri = (e >> 7) & 2047u;
                    if (e != EMPTY && ri >= (uint32_t)KEEP)
                        table[tid * SPT + q] = (e & ~(2047u << 7)) | ((newidx[ri - KEEP] - 1u) << 7);
                }
            }
            keep += tot;
        }
        if (__syncthreads_or(failed)) {  // nothing of this bucket has been written
            if (tid == 0) sk_give_up(flags, fail_list, fail_ctr, b, (uint32_t)(r_end - r_begin));
            return;
        }
    }
    // distinct records of the bucket: thread t owns slots [t * SPT, (t + 1) * SPT)
    constexpr int SPT = TS / NT;
    uint32_t mine = 0;
#pragma unroll
    for (int q = 0; q < SPT; ++q) mine += table[tid * SPT + q] != EMPTY ? 1u : 0u;
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d, 64);
        if (lane >= (uint32_t)d) incl += t;
    }
    if (lane == 63) tmp[32 + wave] = incl;
    __syncthreads();
    uint32_t before = 0, D = 0;
    for (uint32_t j = 0; j < (uint32_t)(NT / 64); ++j) {
        const uint32_t x = tmp[32 + j];
        if (j < wave) before += x;
        D += x;
    }
    if (tid == 0) {
        const unsigned long long base = atomicAdd(out_cursor, (unsigned long long)D);
        reinterpret_cast<unsigned long long *>(tmp + 48)[0] = base;
    }
    __syncthreads();
    const unsigned long long base = reinterpret_cast<unsigned long long *>(tmp + 48)[0];
    if (base + D > out_cap) {
        if (tid == 0) flags[SKF_OUT] = 1u;
        return;
    }
    unsigned long long at = base + before + incl - mine;
#pragma unroll
    for (int q = 0; q < SPT; ++q) {
        const uint32_t e = table[tid * SPT + q];
        if (e == EMPTY) continue;
        const Key<W> G = sk_kmer_at<W, RW>(recs + ((e >> 7) & 2047u) * RW, (e >> 1) & 63u, k);
        const Key<W> X = (e & 1u) ? kmer_rc<W>(G, (int)k) : G;
        key_store<W>(&out_keys[at], X);
        if (OP) out_vals[at] = tvals[tid * SPT + q];
        ++at;
    }
}

// Records the tables could not take (hot buckets, spill list) -> one canonical k-mer (+ edge mask) per instance, for
// the k-mer path's dedup.  bucket_ids != null: one workgroup per listed bucket; else `n_flat` records of a flat list.
template <int W, int OP>
__global__ __launch_bounds__(256) void k_sk_expand(const uint64_t *__restrict__ records,
                                                   const unsigned long long *__restrict__ boff,
                                                   const uint32_t *__restrict__ bucket_ids, uint32_t n_flat, SkParams P,
                                                   Key<W> *__restrict__ keys, uint32_t *__restrict__ vals,
                                                   unsigned long long *__restrict__ cursor, unsigned long long cap,
                                                   uint32_t *__restrict__ flags) {
    constexpr int RW = W + 1;
    unsigned long long r0, r1, stride;
    if (bucket_ids) {
        const uint32_t b = bucket_ids[blockIdx.x];
        r0 = boff[b] + threadIdx.x;
        r1 = boff[b + 1];
        stride = blockDim.x;
    } else {
        r0 = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
        r1 = n_flat;
        stride = (unsigned long long)gridDim.x * blockDim.x;
    }
    const uint32_t k = P.k;
    for (unsigned long long r = r0; r < r1; r += stride) {
        const uint64_t *rp = records + r * RW;
        const uint32_t hdr = (uint32_t)(rp[RW - 1] >> kSkHdrShift);
        const uint32_t n = (hdr & 63u) + 1u;
        const unsigned long long base = atomicAdd(cursor, (unsigned long long)n);
        if (base + n > cap) {
            flags[SKF_OUT] = 1u;
            continue;
        }
        for (uint32_t j = 0; j < n; ++j) {
            const Key<W> F = sk_kmer_at<W, RW>(rp, j, k);
            const Key<W> RC = kmer_rc<W>(F, (int)k);
            const bool minimal = sk_minimal<W>(F, RC);
            key_store<W>(&keys[base + j], key_select<W>(minimal, F, RC));
            if (OP == 3) {
                const bool hp = j > 0 || ((hdr >> 13) & 1u), hn = j + 1u < n || ((hdr >> 16) & 1u);
                const uint32_t prevc = j > 0 ? sk_base_at(rp, j - 1u) : (hdr >> 14) & 3u;
                const uint32_t nextc = j + 1u < n ? sk_base_at(rp, j + k) : (hdr >> 17) & 3u;
                uint32_t val = 0;
                if (hn) val |= 1u << (minimal ? nextc : 7u - nextc);
                if (hp) val |= 1u << (minimal ? 4u + prevc : 3u - prevc);
                vals[base + j] = val;
            }
        }
    }
}

__global__ void k_sk_sum_buckets(const uint32_t *__restrict__ ids, uint32_t n, const unsigned long long *__restrict__ boff,
                                 unsigned long long *__restrict__ total) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(total, boff[ids[i] + 1] - boff[ids[i]]);
}

struct SkDedupArgs {
    const uint8_t *hot;
    uint32_t *fail2_list;
    const uint64_t *records;
    const unsigned long long *boff;
    void *out_keys;
    uint32_t *out_vals;
    unsigned long long *out_cursor;
    uint64_t out_cap;
    uint32_t *flags;
    uint32_t *fail_list;
};

template <int W, int OP, class G>
void launch_dedup_g(bbk_ctx *ctx, const char *fam, uint32_t nblocks, const SkParams &P, const SkDedupArgs &A,
                    const uint32_t *bucket_ids, uint32_t *fail_list, uint32_t fail_ctr, double bytes) {
    if (nblocks == 0) return;
    const size_t sm = sk_dedup_smem<W, OP, G>(P.C);
    auto fn = k_sk_dedup<W, OP, G>;
    BBK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
    KernelTimer t(ctx, fam, bytes);
    hipLaunchKernelGGL(fn, dim3(nblocks), dim3(G::NT), sm, ctx->stream, A.records, A.boff, P, (Key<W> *)A.out_keys, A.out_vals,
                       A.out_cursor, (unsigned long long)A.out_cap, A.flags, bucket_ids, fail_list, fail_ctr, A.hot);
    check_launch("k_sk_dedup");
}

// second == 0: all buckets, first geometry (failures are listed); else the `second` listed buckets, big geometry
template <int W>
void launch_dedup(bbk_ctx *ctx, int op, uint32_t nbuckets, uint32_t second, const SkParams &P, const SkDedupArgs &A, double bytes) {
    const char *fam = second ? "k_sk_dedup_B" : "k_sk_dedup";
#define BBK_SK_DEDUP(OPV)                                                                                    \
    if (second) launch_dedup_g<W, OPV, SkdB>(ctx, fam, second, P, A, A.fail_list, A.fail2_list, (uint32_t)SKF_NFAIL2, bytes); \
    else launch_dedup_g<W, OPV, SkdA>(ctx, fam, nbuckets, P, A, nullptr, A.fail_list, (uint32_t)SKF_NFAIL, bytes)
    switch (op) {
        case MSD_OP_NONE: BBK_SK_DEDUP(0); break;
        case MSD_OP_COUNT: BBK_SK_DEDUP(1); break;
        case MSD_OP_OR: BBK_SK_DEDUP(3); break;
        default: BBK_REQUIRE(false, BBK_ERR_INTERNAL, "superk: bad reduce op %d", op);
    }
#undef BBK_SK_DEDUP
}

template <int W>
bool superk_run(bbk_ctx *ctx, const bbk_reads *rd, unsigned k, int op, DevBuf &out_keys, DevBuf &out_vals,
                uint64_t &n_distinct, uint64_t &n_instances) {
    constexpr int RW = W + 1;
    const bool verbose = getenv("BBK_VERBOSE") != nullptr;
    if (rd->n == 0 || rd->n >= (1ull << 32)) return false;
    const auto t_start = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };
    double t_setup = 0, t_alloc = 0, t_passes = 0;
    // geometry: the record holds nbase_max bases -> runs of up to n_cap k-mers; the minimizer is as long as it can be
    // without natural runs (<= k-m+1 k-mers) exceeding the record, within [15, 31]
    const uint32_t nbase_max = (64u * RW - kSkHdrBits) / 2u;
    const uint32_t n_cap = std::min<uint32_t>(64u, nbase_max - k + 1u);
    const char *em = getenv("BBK_SUPERK_M");  // experiments: minimizer length (11..31, at least k - 63)
    const uint32_t m = em ? (uint32_t)std::min<int>(31, std::max<int>(std::max(11, (int)k - 63), atoi(em)))
                          : (uint32_t)std::min<int>(31, std::max<int>(15, (int)k - (int)n_cap + 1));
    const uint32_t w = k - m + 1u;
    const uint32_t C = std::min(w, n_cap);

    DevBuf nk((rd->n + 1) * sizeof(uint64_t)), coff((rd->n + 1) * sizeof(uint64_t));
    hipLaunchKernelGGL(k_sk_segments, dim3((unsigned)((rd->n + 255) / 256)), dim3(256), 0, ctx->stream, rd->d_len, rd->n, k,
                       C, nk.as<uint64_t>(), coff.as<uint64_t>());
    check_launch("k_sk_segments");
    const uint64_t N = exclusive_scan_u64(ctx, nk.as<uint64_t>(), nk.as<uint64_t>(), rd->n);
    const uint64_t n_segs = exclusive_scan_u64(ctx, coff.as<uint64_t>(), coff.as<uint64_t>(), rd->n);
    nk.release();
    const char *smin = getenv("BBK_SUPERK_MIN");
    const uint64_t min_inst = smin ? strtoull(smin, nullptr, 10) : (1ull << 22);
    if (N < min_inst || N == 0) return false;
    BBK_HIP(hipMemcpyAsync(coff.as<uint64_t>() + rd->n, &n_segs, sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    BBK_HIP(hipStreamSynchronize(ctx->stream));  // n_segs is a stack variable
    const uint64_t ntiles1 = (n_segs + kSk1NT - 1) / kSk1NT;
    if (ntiles1 >= (1ull << 31)) return false;

    // expected records: a segment of C k-mers starts one and the minimum changes with probability 2/(w+1) per step
    // (the last segment of a read is shorter, so this is an upper estimate); the level-1 slots carry 12 % slack.
    // Buckets are planned by INSTANCES: the dedup table holds distinct keys, at most the instances of its bucket
    const double per_seg = 1.0 + (double)(C - 1) * 2.0 / (double)(w + 1);
    const double est_total = (double)n_segs * per_seg * 1.05 + 65536.0;
    // The table holds DISTINCT keys.  Planned for the multiplicity of the previous batch of this context (x 0.5; none
    // yet: 1, every instance its own key): a bucket costs ~15 us of dependent latencies whatever it holds, so buckets
    // a quarter full would spend most of the kernel's time on them.  A batch that turns out less repetitive sends
    // its fuller buckets to the second-chance table (4x the slots) and, past 65536 of those, back to the k-mer path.
    const char *ef = getenv("BBK_SUPERK_FILL");  // tests: overfull buckets exercise the second-chance table
    // (x 0.5: measured at k = 55, multiplicity 3.7 -- planned instances per slot 0.55: 12.3 ms, 1.0: 10.1, 1.3: 10.4,
    // 1.63: 11.5, 2.0: 14.5, 2.4: 20.8; the table wants a load of ~0.3)
    const double dup_plan = std::min(4.0, std::max(1.0, 0.5 * ctx->superk_dup));
    const double fill = ef ? atof(ef) : 0.55 * dup_plan;
    const double nb_total = std::max(1.0, std::ceil((double)N / (fill * SkdA::TS)));
    const char *eb = getenv("BBK_SUPERK_BUCKETS");
    const double per_pass_max = eb ? (double)strtoull(eb, nullptr, 10) : (double)(1u << kSkMaxP1Bits) * kSkMaxP2 * 0.9;
    uint32_t np = (uint32_t)std::max(1.0, std::ceil(nb_total / per_pass_max));
    // one pass fewer when somewhat fuller buckets allow it: a pass re-reads the reads and repeats the minimizer work (38 ms
    // per 100 M reads), fuller tables cost less than that up to ~1.8x the planned load (configs[2]: 2 passes at 1.03
    // instances per slot = 390 ms of stage A, 1 pass at 1.24 = 330 ms)
    double nb_plan = nb_total;
    if (np > 1 && !ef) {
        const double fill_max = 0.55 * std::min(4.0, std::max(1.0, 0.9 * ctx->superk_dup));
        const double need = (double)N / (SkdA::TS * per_pass_max * (np - 1));
        if (need <= fill_max) {
            --np;
            nb_plan = std::ceil((double)N / (need * SkdA::TS));
        }
    }
    const double nbp = std::ceil(nb_plan / np);
    const double want1 = std::max(64.0, nbp / 768.0);
    uint32_t b1bits = 0;
    while (b1bits < kSkMaxP1Bits && (double)(1u << b1bits) < want1 && (double)(1u << b1bits) < nbp) ++b1bits;
    const uint32_t P1 = 1u << b1bits;
    const uint32_t P2 = (uint32_t)std::min<double>(kSkMaxP2, std::max(1.0, std::ceil(nbp / P1)));
    const double est_pass = est_total / np;
    const char *es1 = getenv("BBK_SUPERK_SLOT_SCALE");  // tests: slots below the load, so that records spill
    const uint64_t slot1_64 = (uint64_t)(est_pass / P1 * 1.12 * (es1 ? atof(es1) : 1.0)) + (es1 ? 16 : 4096);
    if (slot1_64 >= (1ull << 31)) return false;
    SkParams P{};
    P.k = k;
    P.m = m;
    P.w = w;
    P.C = C;
    P.np = np;
    P.b1bits = b1bits;
    P.P1 = P1;
    P.P2 = P2;
    static const bool xcd_slots = !(getenv("BBK_XCD_SLOTS") && atoi(getenv("BBK_XCD_SLOTS")) == 0);
    if (xcd_slots && ctx->num_xcds == 8) {  // eight sub-slots, each with the slack of a slot of its size
        P.sub1 = (uint32_t)((slot1_64 + 7) / 8) + (es1 ? 2u : 512u);
        P.slot1 = 8u * P.sub1;
        P.tps_sub = (P.sub1 + kSk2Tile - 1) / kSk2Tile;
        P.tps = 8u * P.tps_sub;
    } else {
        P.sub1 = 0;
        P.tps_sub = 0;
        P.slot1 = (uint32_t)slot1_64;
        P.tps = (P.slot1 + kSk2Tile - 1) / kSk2Tile;
    }
    P.spill_cap = (uint32_t)std::min<double>(2.0e9, es1 ? est_pass + 65536.0 : est_pass / 8 + 65536.0);
    const uint64_t nbuckets = (uint64_t)P1 * P2;
    if ((uint64_t)P1 * P.tps >= (1ull << 31)) return false;
    const size_t sm1 = sk_part1_smem(C, P1, RW);
    if (sm1 > 160 * 1024 || sk_dedup_smem<W, 3, SkdA>(C) > 160 * 1024 || sk_dedup_smem<W, 3, SkdB>(C) > 160 * 1024) return false;
    if (verbose)
        fprintf(stderr,
                "[bbk] superk: k=%u m=%u w=%u C=%u segs=%llu est_records=%.0f passes=%u P1=%u P2=%u slot1=%u lds1=%zu fill=%.2f\n",
                k, m, w, C, (unsigned long long)n_segs, est_total, np, P1, P2, P.slot1, sm1, (double)N / (nb_plan * SkdA::TS));

    DevBuf tiles((size_t)(ntiles1 + 1) * sizeof(SkTile));
    hipLaunchKernelGGL(k_sk_tiles, dim3((unsigned)((ntiles1 + 255) / 256)), dim3(256), 0, ctx->stream, coff.as<uint64_t>(),
                       rd->n, ntiles1, (uint32_t)kSk1NT, n_segs, tiles.as<SkTile>());
    check_launch("k_sk_tiles");
    SkReads S{rd->d_words, rd->d_woff, rd->d_len, coff.as<uint64_t>(), tiles.as<SkTile>(), rd->n, n_segs};
    if (verbose) {
        BBK_HIP(hipStreamSynchronize(ctx->stream));
        t_setup = since();
    }

    const size_t rec_bytes = (size_t)RW * 8;
    DevBuf buf1((size_t)P1 * P.slot1 * rec_bytes), buf2;
    DevBuf cur1((size_t)P1 * 8 * 4 + 16), boff((size_t)(nbuckets + 1) * 8 + 16), cur2((size_t)nbuckets * 8 + 16), dflags(64),
        dcursor(16), fail_list((size_t)kSkdFailCap * 4), fail2_list((size_t)kSkdFailCap * 4), hot, fb_total(16);
    DevBuf spill((size_t)P.spill_cap * rec_bytes + 16);
    // share of the instances the k-mer path may have to take over (hot buckets, spilled records) before the whole batch
    // is handed to it instead
    const char *efm = getenv("BBK_SUPERK_FALLBACK_MAX");
    const double fallback_max = efm ? atof(efm) : 0.25;
    BBK_HIP(hipMemsetAsync(dflags.p, 0, 64, ctx->stream));
    BBK_HIP(hipMemsetAsync(dcursor.p, 0, 16, ctx->stream));

    // output: distinct records are appended; sized from the multiplicity the caller is likely to see and regrown if
    // a pass runs over (its dedup kernel is then run again: the buckets are still there)
    const size_t key_bytes = (size_t)W * 8;
    uint64_t out_cap = std::min<uint64_t>(
        N, (uint64_t)(ctx->superk_dup >= 1.0 ? (double)N / ctx->superk_dup * 1.15 : (double)N / 3.0) + (1u << 20));
    DevBuf okeys(out_cap * key_bytes + 16), ovals;
    if (op != MSD_OP_NONE) ovals.alloc(out_cap * 4 + 16);

    {
        auto fn1 = k_sk_part1<RW>;
        BBK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fn1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm1));
    }
    if (verbose) {
        BBK_HIP(hipStreamSynchronize(ctx->stream));
        t_alloc = since();
    }
    uint32_t hflags[16];
    unsigned long long done_before = 0;
    auto declined = [&](uint32_t pass) {
        // (after a level-1 overflow the slots have holes: level 2 may then have met anything)
        BBK_REQUIRE(!hflags[SKF_SELECT] || hflags[SKF_SLOT1], BBK_ERR_INTERNAL,
                    "superk: a record arrived in a bin its hash does not name");
        if (verbose)
            fprintf(stderr, "[bbk] superk declines (pass %u): level-1 overflow=%u, buckets to the second chance=%u, to the k-mer path=%u (list full=%u), spilled records=%u, largest bucket %u records\n",
                    pass, hflags[SKF_SLOT1], hflags[SKF_NFAIL], hflags[SKF_NFAIL2], hflags[SKF_TABLE], hflags[SKF_NSPILL],
                    hflags[SKF_MAXREC]);
        ctx->add_stat("stat_superk_declined", 1);
        ctx->superk_dup = 0;
        return false;
    };
    static const bool xcd_tiles = !(getenv("BBK_XCD_TILES") && atoi(getenv("BBK_XCD_TILES")) == 0);
    P.xcd_tiles = xcd_tiles ? 1u : 0u;
    const uint32_t grid2 = xcd_tiles ? 8u * ((P1 + 7u) / 8u) * P.tps : P1 * P.tps;
    for (uint32_t pass = 0; pass < np; ++pass) {
        P.pass = pass;
        BBK_HIP(hipMemsetAsync(dflags.p, 0, 64, ctx->stream));
        BBK_HIP(hipMemsetAsync(cur1.p, 0, (size_t)P1 * 8 * 4, ctx->stream));
        BBK_HIP(hipMemsetAsync(boff.p, 0, (size_t)(nbuckets + 1) * 8, ctx->stream));
        {
            KernelTimer t(ctx, "k_sk_part1", (double)rd->n_words * 8 + est_pass * rec_bytes);
            hipLaunchKernelGGL(k_sk_part1<RW>, dim3((unsigned)ntiles1), dim3(kSk1NT), sm1, ctx->stream, S, P, cur1.as<uint32_t>(),
                               buf1.as<uint64_t>(), spill.as<uint64_t>(), dflags.as<uint32_t>());
            check_launch("k_sk_part1");
        }
        {
            KernelTimer t(ctx, "k_sk_part2_hist", est_pass * rec_bytes);
            hipLaunchKernelGGL((k_sk_part2<RW, true>), dim3(grid2), dim3(kSk2NT), 0, ctx->stream, buf1.as<uint64_t>(), P,
                               cur1.as<uint32_t>(), boff.as<unsigned long long>(), (uint64_t *)nullptr, dflags.as<uint32_t>());
            check_launch("k_sk_hist2");
        }
        const uint64_t n_rec = exclusive_scan_u64(ctx, boff.as<uint64_t>(), boff.as<uint64_t>(), nbuckets);
        BBK_HIP(hipMemcpyAsync(boff.as<uint64_t>() + nbuckets, &n_rec, 8, hipMemcpyHostToDevice, ctx->stream));
        BBK_HIP(bbk::copy_async(cur2.p, boff.p, (size_t)nbuckets * 8, hipMemcpyDeviceToDevice, ctx->stream));
        BBK_HIP(hipMemcpyAsync(hflags, dflags.p, 64, hipMemcpyDeviceToHost, ctx->stream));
        BBK_HIP(hipStreamSynchronize(ctx->stream));  // n_rec is a stack variable; level-1 flags
        if (hflags[SKF_SLOT1] || hflags[SKF_STAGE] || hflags[SKF_SELECT]) return declined(pass);
        const uint32_t n_spill = hflags[SKF_NSPILL];
        if (n_spill) {  // hot minimizers: the buckets of the spilled records go to the k-mer path as a whole
            if (!hot.p) hot.alloc((size_t)nbuckets + 16);
            BBK_HIP(hipMemsetAsync(hot.p, 0, (size_t)nbuckets, ctx->stream));
            hipLaunchKernelGGL(k_sk_mark_hot<RW>, dim3((n_spill + 255) / 256), dim3(256), 0, ctx->stream, spill.as<uint64_t>(),
                               n_spill, P, hot.as<uint8_t>(), dflags.as<uint32_t>());
            check_launch("k_sk_mark_hot");
            if (verbose) fprintf(stderr, "[bbk] superk: %u records spilled from full level-1 slots\n", n_spill);
            ctx->add_stat("stat_superk_spilled", (double)n_spill);
        }
        if (buf2.bytes < (n_rec + 1) * rec_bytes) {
            buf2.release();
            buf2.alloc((size_t)((double)(n_rec + 1) * (np > 1 ? 1.05 : 1.0)) * rec_bytes);
        }
        {
            KernelTimer t(ctx, "k_sk_part2", 2.0 * (double)n_rec * rec_bytes);
            hipLaunchKernelGGL((k_sk_part2<RW, false>), dim3(grid2), dim3(kSk2NT), 0, ctx->stream, buf1.as<uint64_t>(), P,
                               cur1.as<uint32_t>(), cur2.as<unsigned long long>(), buf2.as<uint64_t>(), dflags.as<uint32_t>());
            check_launch("k_sk_part2");
        }
        ctx->add_stat("stat_superk_records", (double)n_rec);
        for (int attempt = 0;; ++attempt) {
            // algorithmic bytes: the records read + the distinct keys written; their number is known once the cursor has
            // been read back (below) and is put into the timer's entry then
            const double db = (double)n_rec * rec_bytes;
            const size_t timer_entry = ctx->pending.size();
            SkDedupArgs A{n_spill ? hot.as<uint8_t>() : nullptr, fail2_list.as<uint32_t>(), buf2.as<uint64_t>(),
                          boff.as<unsigned long long>(), okeys.p, ovals.as<uint32_t>(), dcursor.as<unsigned long long>(), out_cap,
                          dflags.as<uint32_t>(), fail_list.as<uint32_t>()};
            BBK_HIP(hipMemsetAsync(dflags.as<uint32_t>() + SKF_NFAIL, 0, 4, ctx->stream));
            BBK_HIP(hipMemsetAsync(dflags.as<uint32_t>() + SKF_NFAIL2, 0, 4, ctx->stream));
            launch_dedup<W>(ctx, op, (uint32_t)nbuckets, 0, P, A, db);
            unsigned long long cursor_now = 0;
            BBK_HIP(hipMemcpyAsync(hflags, dflags.p, 64, hipMemcpyDeviceToHost, ctx->stream));
            BBK_HIP(hipMemcpyAsync(&cursor_now, dcursor.p, 8, hipMemcpyDeviceToHost, ctx->stream));
            BBK_HIP(hipStreamSynchronize(ctx->stream));
            if (ctx->profiling && ctx->pending.size() > timer_entry && cursor_now >= done_before)
                ctx->pending[timer_entry].bytes += (double)(cursor_now - done_before) * (key_bytes + (op != MSD_OP_NONE ? 4 : 0));
            if (hflags[SKF_NFAIL] && !hflags[SKF_OUT]) {  // second chance for the buckets the small table gave up
                if (hflags[SKF_NFAIL] > kSkdFailCap) return declined(pass);
                if (verbose)
                    fprintf(stderr, "[bbk] superk: %u buckets to the second-chance table (largest %u records)\n", hflags[SKF_NFAIL],
                            hflags[SKF_MAXREC]);
                ctx->add_stat("stat_superk_second_chance", (double)hflags[SKF_NFAIL]);
                launch_dedup<W>(ctx, op, (uint32_t)nbuckets, hflags[SKF_NFAIL], P, A, 0);
                BBK_HIP(hipMemcpyAsync(hflags, dflags.p, 64, hipMemcpyDeviceToHost, ctx->stream));
                BBK_HIP(hipMemcpyAsync(&cursor_now, dcursor.p, 8, hipMemcpyDeviceToHost, ctx->stream));
                BBK_HIP(hipStreamSynchronize(ctx->stream));
            }
            if (hflags[SKF_TABLE] || hflags[SKF_SELECT]) return declined(pass);
            const uint32_t n_fail2 = hflags[SKF_NFAIL2];
            if (!hflags[SKF_OUT] && (n_fail2 || n_spill)) {
                // What the tables could not take -- buckets with more distinct keys than the second-chance table holds
                // (one low-complexity minimizer shared by thousands of k-mers) and the buckets of spilled records -- is
                // expanded into one key per instance and deduplicated by the k-mer path; the keys of a bucket occur in
                // no other bucket, so the distinct records are simply appended.
                unsigned long long fb_rec = 0;
                BBK_HIP(hipMemsetAsync(fb_total.p, 0, 16, ctx->stream));
                if (n_fail2) {
                    hipLaunchKernelGGL(k_sk_sum_buckets, dim3((n_fail2 + 255) / 256), dim3(256), 0, ctx->stream,
                                       fail2_list.as<uint32_t>(), n_fail2, boff.as<unsigned long long>(),
                                       fb_total.as<unsigned long long>());
                    check_launch("k_sk_sum_buckets");
                }
                BBK_HIP(hipMemcpyAsync(&fb_rec, fb_total.p, 8, hipMemcpyDeviceToHost, ctx->stream));
                BBK_HIP(hipStreamSynchronize(ctx->stream));
                fb_rec += n_spill;
                const uint64_t inst_ub = fb_rec * (uint64_t)C;
                if (verbose)
                    fprintf(stderr, "[bbk] superk: %u buckets (largest %u records) + %u spilled records = %llu records to the k-mer path\n",
                            n_fail2, hflags[SKF_MAXREC], n_spill, fb_rec);
                if ((double)inst_ub > fallback_max * (double)N / np + 65536.0) return declined(pass);
                ctx->add_stat("stat_superk_kmer_path_records", (double)fb_rec);
                DevBuf fk(inst_ub * key_bytes + 16), fv;
                if (op == MSD_OP_OR) fv.alloc(inst_ub * 4 + 16);
                BBK_HIP(hipMemsetAsync(fb_total.p, 0, 16, ctx->stream));
                auto expand = [&](const uint64_t *recs, const uint32_t *ids, uint32_t nblocks, uint32_t n_flat) {
                    if (nblocks == 0) return;
                    KernelTimer t(ctx, "k_sk_expand", 0);
                    if (op == MSD_OP_OR)
                        hipLaunchKernelGGL((k_sk_expand<W, 3>), dim3(nblocks), dim3(256), 0, ctx->stream, recs, boff.as<unsigned long long>(),
                                           ids, n_flat, P, fk.as<Key<W>>(), fv.as<uint32_t>(), fb_total.as<unsigned long long>(),
                                           (unsigned long long)inst_ub, dflags.as<uint32_t>());
                    else
                        hipLaunchKernelGGL((k_sk_expand<W, 0>), dim3(nblocks), dim3(256), 0, ctx->stream, recs, boff.as<unsigned long long>(),
                                           ids, n_flat, P, fk.as<Key<W>>(), (uint32_t *)nullptr, fb_total.as<unsigned long long>(),
                                           (unsigned long long)inst_ub, dflags.as<uint32_t>());
                    check_launch("k_sk_expand");
                };
                expand(buf2.as<uint64_t>(), fail2_list.as<uint32_t>(), n_fail2, 0);
                expand(spill.as<uint64_t>(), nullptr, n_spill ? std::min<uint32_t>((n_spill + 255) / 256, 65536u) : 0u, n_spill);
                unsigned long long fb_inst = 0;
                BBK_HIP(hipMemcpyAsync(&fb_inst, fb_total.p, 8, hipMemcpyDeviceToHost, ctx->stream));
                BBK_HIP(hipMemcpyAsync(hflags, dflags.p, 64, hipMemcpyDeviceToHost, ctx->stream));
                BBK_HIP(hipStreamSynchronize(ctx->stream));
                BBK_REQUIRE(!hflags[SKF_OUT] && fb_inst <= inst_ub, BBK_ERR_INTERNAL, "superk: expansion ran over its bound");
                MsdOutput mo;
                if (!msd_sort_reduce(ctx, k, MSD_HASH, op, nullptr, fk.p, op == MSD_OP_OR ? fv.as<uint32_t>() : nullptr, fb_inst,
                                     false, mo))
                    return declined(pass);
                fk.release();
                fv.release();
                if (cursor_now + mo.n > out_cap) {  // room for the appended records
                    const uint64_t new_cap = cursor_now + mo.n + (N - std::min<uint64_t>(N, cursor_now + mo.n)) / 8;
                    DevBuf nkeys(new_cap * key_bytes + 16), nvals;
                    BBK_HIP(bbk::copy_async(nkeys.p, okeys.p, cursor_now * key_bytes, hipMemcpyDeviceToDevice, ctx->stream));
                    if (op != MSD_OP_NONE) {
                        nvals.alloc(new_cap * 4 + 16);
                        BBK_HIP(bbk::copy_async(nvals.p, ovals.p, cursor_now * 4, hipMemcpyDeviceToDevice, ctx->stream));
                    }
                    BBK_HIP(hipStreamSynchronize(ctx->stream));
                    okeys = std::move(nkeys);
                    if (op != MSD_OP_NONE) ovals = std::move(nvals);
                    out_cap = new_cap;
                }
                if (mo.n) {
                    BBK_HIP(bbk::copy_async(okeys.as<char>() + cursor_now * key_bytes, mo.keys.p, mo.n * key_bytes,
                                           hipMemcpyDeviceToDevice, ctx->stream));
                    if (op != MSD_OP_NONE)
                        BBK_HIP(bbk::copy_async(ovals.as<uint32_t>() + cursor_now, mo.vals.p, mo.n * 4, hipMemcpyDeviceToDevice,
                                               ctx->stream));
                }
                cursor_now += mo.n;
                BBK_HIP(hipMemcpyAsync(dcursor.p, &cursor_now, 8, hipMemcpyHostToDevice, ctx->stream));
                BBK_HIP(hipStreamSynchronize(ctx->stream));  // cursor_now is a stack variable; mo's buffers are about to go
            }
            if (!hflags[SKF_OUT]) {
                done_before = cursor_now;
                break;
            }
            // the output ran over: keep what earlier passes wrote, grow, run this pass's dedup again
            BBK_REQUIRE(attempt < 2, BBK_ERR_INTERNAL, "superk: output still too small after regrowing");
            const uint64_t new_cap =
                std::min<uint64_t>(N, attempt == 0 ? std::max<uint64_t>(2 * out_cap, done_before + N / np) : N);
            if (verbose)
                fprintf(stderr, "[bbk] superk: output regrown %llu -> %llu records (pass %u)\n", (unsigned long long)out_cap,
                        (unsigned long long)new_cap, pass);
            DevBuf nkeys(new_cap * key_bytes + 16), nvals;
            BBK_HIP(bbk::copy_async(nkeys.p, okeys.p, done_before * key_bytes, hipMemcpyDeviceToDevice, ctx->stream));
            if (op != MSD_OP_NONE) {
                nvals.alloc(new_cap * 4 + 16);
                BBK_HIP(bbk::copy_async(nvals.p, ovals.p, done_before * 4, hipMemcpyDeviceToDevice, ctx->stream));
            }
            BBK_HIP(hipStreamSynchronize(ctx->stream));
            okeys = std::move(nkeys);
            if (op != MSD_OP_NONE) ovals = std::move(nvals);
            out_cap = new_cap;
            BBK_HIP(hipMemcpyAsync(dcursor.p, &done_before, 8, hipMemcpyHostToDevice, ctx->stream));
            BBK_HIP(hipMemsetAsync(reinterpret_cast<uint32_t *>(dflags.p) + SKF_OUT, 0, 4, ctx->stream));
            BBK_HIP(hipStreamSynchronize(ctx->stream));  // done_before must stay put until the copy has run
        }
    }
    buf1.release();
    buf2.release();
    if (verbose) t_passes = since();
    const uint64_t D = done_before;
    n_instances = N;
    n_distinct = D;
    ctx->add_stat("stat_superk_batches", 1);
    if (N >= (1ull << 20)) ctx->superk_dup = D ? (double)N / (double)D : 0.0;  // a small batch says little about the next
    auto report = [&]() {
        if (verbose)
            fprintf(stderr, "[bbk] superk: %llu instances -> %llu distinct; wall %.3f s (setup %.3f, buffers %.3f, passes %.3f, result %.3f)\n",
                    (unsigned long long)N, (unsigned long long)D, since(), t_setup, t_alloc - t_setup, t_passes - t_alloc,
                    since() - t_passes);
    };
    // the caller keeps the result for the rest of the job: do not leave it in a buffer sized for the estimate
    if (out_cap > D + D / 8 + (1u << 20)) {
        DevBuf xk(D * key_bytes + 16), xv;
        BBK_HIP(bbk::copy_async(xk.p, okeys.p, D * key_bytes, hipMemcpyDeviceToDevice, ctx->stream));
        if (op != MSD_OP_NONE) {
            xv.alloc(D * 4 + 16);
            BBK_HIP(bbk::copy_async(xv.p, ovals.p, D * 4, hipMemcpyDeviceToDevice, ctx->stream));
        }
        BBK_HIP(hipStreamSynchronize(ctx->stream));
        out_keys = std::move(xk);
        if (op != MSD_OP_NONE) out_vals = std::move(xv);
    } else {
        out_keys = std::move(okeys);
        if (op != MSD_OP_NONE) out_vals = std::move(ovals);
    }
    report();
    return true;
}

}  // namespace

// Distinct canonical k-mers (+ count / OR of edge masks) of a batch of reads through super-k-mer records.
// false: not taken (key width, size, switch) or given up (a slot or table ran over): the caller uses the k-mer path.
bool superk_dedup_reads(bbk_ctx *ctx, const bbk_reads *rd, unsigned k, int op, DevBuf &out_keys, DevBuf &out_vals,
                        uint64_t &n_distinct, uint64_t &n_instances) {
    if (getenv("BBK_NO_SUPERK")) return false;
    const unsigned W = words_of(k);
    if (W == 2) return superk_run<2>(ctx, rd, k, op, out_keys, out_vals, n_distinct, n_instances);
    if (W == 3) return superk_run<3>(ctx, rd, k, op, out_keys, out_vals, n_distinct, n_instances);
    if (W == 4) return superk_run<4>(ctx, rd, k, op, out_keys, out_vals, n_distinct, n_instances);
    return false;
}

}  // namespace bbk
