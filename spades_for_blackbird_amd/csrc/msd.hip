// msd.hip -- the fast sort-and-count path: two MSD radix-partition levels in HBM, then one
// workgroup per bucket deduplicates / sorts + reduces its keys entirely in LDS.
//
// Why: the LSD path (primitives.hip) moves every record 3x per 8-bit pass (k=21: 6 passes, ~19 N*W
// bytes).  Here a record is written by the (fused) extraction+partition, read and written once
// more by the second partition level and read once by the bucket kernel, independent of the key
// length, and the duplicate-heavy k-mer stream (50x coverage) collapses inside LDS.
//
//   k_part_reads : fused k-mer extraction (chunks of one read per lane, rolled) + level-1 partition:
//                  LDS histogram with returning ds_add = local rank, one global atomicAdd per
//                  (tile, non-empty bin) reserves the output run, keys reordered in LDS so that a
//                  wave stores contiguous per-bin runs (unstable: the buckets get sorted later)
//   k_part       : the same over a key array (levels 1 and 2); <HIST> variants only count
//   k_bucket_hash / k_bucket_hashidx : one bucket in an LDS open-addressing table: dedup + reduce
//                  (HASH prefix: the order does not matter)
//   k_bucket_dist: one bucket sorted in LDS by one distribution pass + in-bin ranking, head flags +
//                  segmented reduce (count / sum / OR); k_bucket (ballot-ranked LSD radix) is the
//                  second chance for crowded bins and oversized buckets
//   k_compact    : buckets -> dense output
//
// Two modes (MsdRunner::run): the exact mode counts every level first (histogram kernels, dense
// layout); the slot mode (HASH prefix) gives segments and buckets fixed slots and sends what does
// not fit to a spill list that the exact mode finishes -- no histogram passes.
//
// The partition digit comes from a 32-bit "prefix" p(key): HASH (multiplicative mix: uniform whatever
// the sequence composition; used when only the distinct set matters), KEYS (the key's own top bits:
// output globally ascending) or REF (XXH3 bucket of 16, then key bits: the final_kmers order,
// reference kmer_buckets.hpp:28-33 + kmer_index_builder.hpp:168-181).  Level 2 maps the
// remaining prefix bits monotonically onto nb2 bins, so bucket order == prefix order.
// Buckets larger than CAP (a k-mer repeated thousands of times, skewed composition in KEYS mode)
// are finished by the LSD path, per bucket; if too much overflows the caller falls back entirely.
// Environment knobs (tests / diagnostics): BBK_DISABLE_MSD, BBK_NO_SLOTS, BBK_SLOTS_MIN, BBK_NO_DIST,
// BBK_PASS_LIMIT, BBK_VERBOSE; -DBBK_PHASE_PROF builds per-phase shader clocks into the kernels.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "bbk_internal.h"
#include "kmer_ops.h"
#include "msd.h"

namespace bbk {

// Partition tile of a key array: 8192 records of 8 B (4096 of 16 B) staged in LDS, 512 threads x 16
// (x 8) items so the loads stay wide.  Reads are partitioned by k_part_reads (own geometry below).
#ifndef BBK_KEYS_TILE
#define BBK_KEYS_TILE 8192
#endif
#ifndef BBK_KEYS_THREADS
#define BBK_KEYS_THREADS 512
#endif
template <int W>
struct PartCfg {
    static constexpr int TILE = (W == 1) ? BBK_KEYS_TILE : (W == 2 ? 4096 : 2048);  // 64 KB / 48 KB / 64 KB of LDS
    static constexpr int THREADS = BBK_KEYS_THREADS;
    static constexpr int ITEMS = TILE / THREADS;
};
// Fused extraction + level-1 partition: a lane owns one CHUNK of up to CH consecutive k-mer positions of
// ONE read (8-byte keys: 8, rolled base by base; wider keys: 4), a workgroup 1024 chunks.
constexpr int kRdThreads = 1024;      // scatter: big tiles, long per-bin runs
constexpr int kRdHistThreads = 512;   // histogram: nothing is staged, four workgroups per CU hide the prologues
constexpr int kRdSlots = 1024;  // reads of one tile whose cursor tables fit LDS
constexpr int kRdWords = 2048;  // packed read words of one tile staged in LDS (150 bp reads need ~330)
template <int W>
struct RdCfg {
    static constexpr int CH = (W == 1) ? 8 : (W == 2 ? 4 : 2);  // records of a tile: 64 KB (48 KB for 24-byte keys)
    static constexpr int TILE = kRdThreads * CH;
};
constexpr int kMaxBins = 1024;

// 32-bit partition prefix: bucket order == prefix order (only ~20 top bits are ever consumed)
template <int W>
__device__ inline uint32_t prefix_of(const Key<W> &key, int dmode, int w0bits) {
    if (dmode == MSD_HASH) return part_hash32<W>(key);
    const uint64_t top = (w0bits >= 64) ? key.w[0] : (key.w[0] << (64 - w0bits));
    if (dmode == MSD_KEYS) return (uint32_t)(top >> 32);
    const uint32_t b = (uint32_t)__umul64hi(xxh3_64<W>(key), 16ull);
    return (b << 28) | (uint32_t)(top >> 36);
}

struct PartLevel {
    int level;        // 1 or 2
    int b1;           // log2(nb1)
    uint32_t nb1;
    int dmode;
    int w0bits;
    // level 2: every level-1 segment gets its own bin count (sized from its record count, so a
    // skewed prefix distribution still gives buckets of the target size) and flat bin base
    const uint32_t *seg_nb2;
    const uint32_t *seg_bin_start;
    // range pass (inputs above one device batch): only records whose prefix lies in [sel_lo, sel_lo + sel_span)
    // take part (sel_span == 0: all of them); the prefix inside the range, (p - sel_lo) << sel_shl, drives the bins.
    // HASH prefix: 2^b equal hash ranges; KEYS / REF prefix: ranges of the key space sized from a histogram, so the
    // concatenated passes are in prefix order.
    uint32_t sel_lo;
    uint32_t sel_span;
    int sel_shl;
    uint32_t sel_mul;  // stretches (p - sel_lo) << sel_shl, which only reaches span << shl, over the whole 32 bits
    // slot mode (histogram-free HASH path): bin g of this level owns the fixed range [g*slot_cap, (g+1)*slot_cap) of
    // the output and `cursor[g]` starts at g*slot_cap; records that do not fit are appended to the spill list
    uint32_t slot_cap;     // 0: dense layout from an exact histogram
    uint32_t slot_stride;  // distance between slots (>= slot_cap; padded so that slots do not alias in HBM channels)
    void *spill_keys;      // Key<W>[spill_cap]
    uint32_t *spill_vals;  // payloads alongside (records with a payload)
    uint32_t *spill_count; // records appended (may run past spill_cap: the host checks)
    uint32_t spill_cap;
    // narrow stage A (8-byte keys, 2k - 32 = narrow_hb in [1, 10]): 4-byte records between the levels, see "narrow" below
    int narrow_hb;
    // narrow level 1: every segment slot is cut into 2^xcd_shift sub-slots of sub_cap records, one per XCD, with a
    // cursor each (cursor[(bin << xcd_shift) + xcc]).  A (tile, bin) run is ~60 bytes and starts wherever the last one
    // ended; with one fill front per bin a 128-byte line is filled by workgroups on different XCDs, i.e. through
    // different L2s, which is slow (see the call site).  With a fill front per (bin, XCD) every line is one XCD's.  Level 2
    // reads the sub-slots as segments of their own and sends them to the buckets of the parent segment.
    int xcd_shift;
    uint32_t sub_cap;
};

// ---- narrow records (stage A, 17 <= k <= 21) ---------------------------------------------------------------
// A k-mer of 2k <= 42 bits is (hi: 2k - 32 = hb bits, lo: 32 bits = its first 16 bases).  With t = mix(lo), level 1
// sends it to segment
//   bin1 = (t >> 22) ^ (hi << (10 - hb))      (the top ten hash bits of lo, hi folded into the upper hb of them)
// and stores ONLY lo: inside a segment lo determines hi (= (bin1 ^ t >> 22) >> (10 - hb)), so 4-byte records are exact --
// equal lo <=> equal k-mer.  Level 2 and the in-LDS dedup work on lo alone (their bins / slots are other bits of the
// same mix), the dedup kernel rebuilds the 8-byte key from (segment, lo) when it writes the distinct records.  The
// canonical stream -- 1.3 G records at BASELINE configs[1], 8.3x the distinct set -- travels as 4 bytes per record
// instead of 8 through its three passes (level-1 write, level-2 read + write, dedup read).
constexpr int kNwBins1 = 1024;
__device__ inline uint32_t nw_mix(uint32_t lo) {
    uint32_t t = lo * 0x9E3779B1u;
    t ^= t >> 15;
    t *= 0x85EBCA6Bu;
    t ^= t >> 13;
    return t;
}
__device__ inline uint32_t nw_slot(uint32_t t) { return (t * 0x27D4EB2Fu) >> 19; }  // 13 bits for the LDS table
__device__ inline uint32_t nw_bin1(uint32_t hi, uint32_t t, int hb) { return (t >> 22) ^ (hi << (10 - hb)); }
// prefix for level 2: the bits of the mix that level 1 has not used (top-aligned)
__device__ inline uint32_t nw_p2(uint32_t t) { return t << 10; }
__device__ inline uint64_t nw_key(uint32_t bin1, uint32_t lo, int hb) {
    const uint32_t hi = (bin1 ^ (nw_mix(lo) >> 22)) >> (10 - hb);
    return ((uint64_t)hi << 32) | lo;
}

// applies the range selection: false = the record belongs to another pass; p loses the selection bits
__device__ inline bool select_prefix(uint32_t &p, const PartLevel &L) {
    if (L.sel_span == 0) return true;
    const uint32_t d = p - L.sel_lo;
    if (d >= L.sel_span) return false;
    // a span that is not a power of two would leave the top of the prefix space (up to half of the bins) empty and
    // crowd the rest: scale by 2^32 / (span << shl) in (1, 2], monotone (bucket order = prefix order is kept)
    p = d << L.sel_shl;
    p += __umulhi(p, L.sel_mul);
    return true;
}

// bin of this level inside its segment (nb = bins of the segment at level 2)
__device__ inline uint32_t bin_of(uint32_t p, const PartLevel &L, uint32_t nb) {
    if (L.level == 1) return L.b1 == 0 ? 0u : (p >> (32 - L.b1));
    const uint32_t rest = L.b1 == 0 ? p : (p << L.b1);
    return __umulhi(rest, nb);
}

struct ReadSrc {
    const uint64_t *words;
    const uint64_t *woff;
    const uint32_t *len;
    const uint64_t *coff;       // exclusive scan of chunks per read (n_reads + 1)
    const struct RdTile *tiles;  // per tile: its reads and the window of packed words to stage (k_tile_reads)
    uint64_t n_reads;
    uint64_t n_chunks;
    int k;
};

// largest r in [0, nr) with s_rel[r] <= c
__device__ inline uint32_t read_of(const int32_t *s_rel, uint32_t nr, int32_t c) {
    uint32_t lo = 0, hi = nr;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (s_rel[mid] <= c) lo = mid;
        else hi = mid;
    }
    return lo;
}

// What a workgroup of k_part_reads needs to start on a tile, precomputed so that its prologue is ONE scalar load
// followed by the coalesced table/word copies instead of three dependent global round trips.
struct RdTile {
    uint64_t wbase;  // first packed word of the staged window
    uint32_t r0;     // read holding the tile's first chunk
    uint32_t nr;     // reads r0 .. r0+nr-1 own chunks of (or lie inside) the tile
    uint32_t wspan;  // words of the window; 0xFFFFFFFF: does not fit LDS / not in read order (global-memory path)
    uint32_t pad;
};

// largest r in [0, n_reads) with off[r] <= j
__device__ inline uint64_t read_at(const uint64_t *__restrict__ off, uint64_t n_reads, uint64_t j) {
    uint64_t lo = 0, hi = n_reads;
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (off[mid] <= j) lo = mid;
        else hi = mid;
    }
    return lo;
}

// one thread per tile of `tile` chunks (a chunk = ch k-mer positions of one read)
__global__ void k_tile_reads(const uint64_t *__restrict__ coff, const uint64_t *__restrict__ woff,
                             const uint32_t *__restrict__ len, uint64_t n_reads, uint64_t n_tiles, uint32_t tile,
                             uint32_t ch, uint32_t k, uint32_t max_reads, uint32_t max_words,
                             const uint32_t *__restrict__ unordered, RdTile *__restrict__ out) {
    const uint64_t t = BBK_GID();
    if (t >= n_tiles) return;
    const uint64_t c0 = t * (uint64_t)tile;
    const uint64_t r0 = read_at(coff, n_reads, c0), r1 = read_at(coff, n_reads, c0 + tile);
    // staged word window: from the word of the first base this tile touches in r0 (one base before the chunk,
    // for the incoming-edge bit) to the last word it can touch in r1
    const uint32_t p0 = (uint32_t)(c0 - coff[r0]) * ch;
    const uint64_t wbase = woff[r0] + ((p0 ? p0 - 1u : 0u) >> 5);
    const uint32_t len1 = len[r1];
    const uint64_t span1 = (c0 + tile - coff[r1]) * ch + k;  // base index the tile can reach in r1
    const uint32_t lastb1 = len1 ? (uint32_t)(span1 < (uint64_t)(len1 - 1u) ? span1 : (uint64_t)(len1 - 1u)) : 0u;
    const uint64_t wend = woff[r1] + (len1 ? (lastb1 >> 5) + 1u : 0u);
    const uint64_t nr = r1 - r0 + 1;
    // words in read order (checked once for all reads): every read of the tile then lies inside [wbase, wend)
    const bool fast = *unordered == 0 && nr <= (uint64_t)max_reads && wend >= wbase && wend - wbase <= (uint64_t)max_words;
    RdTile T;
    T.wbase = wbase;
    T.r0 = (uint32_t)r0;
    T.nr = (uint32_t)nr;
    T.wspan = fast ? (uint32_t)(wend - wbase) : 0xFFFFFFFFu;
    T.pad = 0;
    out[t] = T;
}

// Tile -> (segment, range).  Level 1: tile t covers records [t*TILE, ...).  Level 2: tiles never
// straddle a level-1 bin: seg_tile_start[b] = first tile of bin b (nb1 + 1 entries).
struct TileMap {
    const uint32_t *seg_tile_start;  // null for level 1
    const uint32_t *seg_off;         // record offset of every level-1 bin (nb1 + 1), level 2 only
    const uint32_t *seg_size;        // records of every level-1 bin; null: seg_off[s + 1] - seg_off[s] (dense)
    uint32_t nseg;
    uint64_t n;
    uint32_t ntiles;  // tiles of the level
    uint32_t group;   // histogram kernels: consecutive tiles one workgroup walks
    const uint4 *desc;  // level 2: per tile (first record, records, bins of its segment, flat index of bin 0),
                        // precomputed so that a workgroup starts with one load instead of a binary search
    // level 1 over a CANONICAL key array that is expanded on the fly: record 2c is key c, record 2c+1 its reverse
    // complement (the both-strand set of spades-kmercount; M.n counts records); expand_tag: the XXH3 bucket of 16
    // goes into bits 2k..2k+3 of either (final_kmers order by one ascending sort, see count.hip)
    int expand_k;  // 0: the array holds the records themselves
    int expand_tag;
};

// level 2: tile -> descriptor (one thread per tile)
__global__ void k_tile_desc(TileMap M, const uint32_t *__restrict__ seg_nb2, const uint32_t *__restrict__ seg_bin_start,
                            uint32_t tile_size, int sub_shift, const uint32_t *__restrict__ xstart,
                            uint4 *__restrict__ desc) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M.ntiles) return;
    uint32_t lo = 0, hi = M.nseg;  // largest s with seg_tile_start[s] <= t
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (M.seg_tile_start[mid] <= t) lo = mid;
        else hi = mid;
    }
    const uint32_t b = M.seg_off[lo] + (t - M.seg_tile_start[lo]) * tile_size;
    const uint32_t e = M.seg_size ? M.seg_off[lo] + M.seg_size[lo] : M.seg_off[lo + 1];
    // M's entries are the level-1 segments or their per-XCD sub-slots (sub_shift); xstart: the tiles of a segment go to
    // the workgroups of one XCD (see k_tile_desc_narrow)
    const uint32_t seg = lo >> sub_shift;
    const uint32_t at = xstart ? 8u * (xstart[lo] + (t - M.seg_tile_start[lo])) + (seg & 7u) : t;
    desc[at] = make_uint4(b, (e - b) < tile_size ? (e - b) : tile_size, seg_nb2[seg], seg_bin_start[seg]);
}

struct TileInfo {
    uint64_t begin;
    uint32_t count, nb;
    uint64_t gbin0;
};

__device__ inline TileInfo tile_info(const TileMap &M, const PartLevel &L, uint32_t tile, uint32_t tile_size) {
    TileInfo T;
    if (M.desc) {
        const uint4 d = M.desc[tile];
        T.begin = d.x;
        T.count = d.y;
        T.nb = d.z;
        T.gbin0 = d.w;
    } else {  // level 1: one segment, tile t covers records [t * tile_size, ...)
        T.begin = (uint64_t)tile * tile_size;
        const uint64_t rem = M.n - T.begin;
        T.count = rem < (uint64_t)tile_size ? (uint32_t)rem : tile_size;
        T.nb = L.nb1;
        T.gbin0 = 0;
    }
    return T;
}

#ifdef BBK_PHASE_PROF
// phase clocks of the scatter kernels (diagnostic build only): [kernel kind][phase] summed shader cycles of
// thread 0 of every workgroup, [..][7] = workgroups
__device__ unsigned long long g_phase[6][8];
#define BBK_PH(kind, ph, t_prev)                                                   \
    do {                                                                           \
        if (threadIdx.x == 0) {                                                    \
            const unsigned long long t_now = clock64();                            \
            atomicAdd(&g_phase[kind][ph], t_now - t_prev);                         \
            t_prev = t_now;                                                        \
        }                                                                          \
    } while (0)
#else
#define BBK_PH(kind, ph, t_prev) \
    do {                         \
    } while (0)
#endif

// Common tail of the scatter kernels.  On entry lhist[b] = records of bin b in this tile and binrank[i] =
// bin << 16 | rank-in-bin (0xFFFFFFFF: no record).  One global atomicAdd per non-empty bin reserves the
// tile's run in that bin; the records are reordered through LDS (stage) so that a wave stores contiguous
// per-bin runs.
// Inclusive prefix sum over the 64 lanes of a wave with DPP row shifts and row broadcasts: six v_add with a DPP operand.
// (__shfl_up goes through ds_bpermute: an address register per distance, an LDS-pipe operation and a select per step.)
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2, 3
    return v;
}

// Per-wave totals (nw <= 64 values in LDS, written before the last barrier) -> the sum of the waves before `wave` and
// the grand total.  Every wave scans the few values itself: log2(nw) shuffle steps instead of a loop of nw LDS reads
// per thread (which was ~80 vector instructions per thread in a workgroup of 16 waves).
template <int NW>
__device__ __forceinline__ void wave_totals(const uint32_t *tmp, int lane, int wave, uint32_t &before, uint32_t &total) {
    static_assert(NW <= 16, "one DPP row");
    const uint32_t v = lane < NW ? tmp[lane] : 0u;
    uint32_t inc = v;
    if (NW > 1) inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xf, 0xf, false);
    if (NW > 2) inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xf, 0xf, false);
    if (NW > 4) inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xf, 0xf, false);
    if (NW > 8) inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x118, 0xf, 0xf, false);
    total = (uint32_t)__builtin_amdgcn_readlane((int)inc, NW - 1);
    before = (uint32_t)__builtin_amdgcn_readlane((int)(inc - v), __builtin_amdgcn_readfirstlane(wave));
}

template <int W, int ITEMS, int THREADS, int MAXB, bool HAS_VAL>
__device__ __forceinline__ void part_tail(const Key<W> (&keys)[ITEMS], const uint32_t (&vals)[ITEMS],
                                          const uint32_t (&binrank)[ITEMS], uint32_t *lhist, uint32_t *lstart,
                                          uint32_t *goff, uint32_t *scan_tmp, Key<W> *stage, uint32_t *vstage,
                                          uint32_t nb, uint64_t gbin0, const PartLevel &L,
                                          uint32_t *__restrict__ cursor, Key<W> *__restrict__ out,
                                          uint32_t *__restrict__ vout, int prof_kind = 0,
                                          unsigned long long t_prev = 0) {
    const int tid = threadIdx.x;
    (void)prof_kind;
    (void)t_prev;
    uint32_t staged = 0;
    // level 1 in slot mode: one fill front (cursor and sub-slot) per (segment, XCD), see PartLevel::xcd_shift
    uint32_t xcc = 0;
    if (L.xcd_shift) {
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= (1u << L.xcd_shift) - 1u;
    }
    constexpr int BPT = (MAXB + THREADS - 1) / THREADS;
    uint32_t greserve[BPT], cq[BPT], ex0 = 0;
    {
        uint32_t c[BPT];
        uint32_t v = 0;
#pragma unroll
        for (int q = 0; q < BPT; ++q) {
            const uint32_t bq = BPT * tid + q;
            c[q] = bq < nb ? lhist[bq] : 0;
            cq[q] = c[q];
            v += c[q];
        }
        const int lane = tid & 63, wave = tid >> 6;
        uint32_t incl = v;
        incl = wave_scan_incl(incl);
        if (lane == 63) scan_tmp[wave] = incl;
        __syncthreads();
        uint32_t wbase, all;
        wave_totals<THREADS / 64>(scan_tmp, lane, wave, wbase, all);
        staged = all;  // records of this tile that take part
        ex0 = wbase + incl - v;
        uint32_t ex = ex0;
#pragma unroll
        for (int q = 0; q < BPT; ++q) {
            const uint32_t bq = BPT * tid + q;
            if (bq < nb) lstart[bq] = ex;
            // the reservation is issued now and consumed after the LDS reorder: its latency overlaps that phase
            greserve[q] = (bq < nb && c[q]) ? atomicAdd(&cursor[((gbin0 + bq) << L.xcd_shift) + xcc], c[q]) : 0u;
            ex += c[q];
        }
    }
    __syncthreads();
    BBK_PH(prof_kind, 2, t_prev);  // scan (+ reservation issue)

#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        if (binrank[i] != 0xFFFFFFFFu) {
            const uint32_t pos = lstart[binrank[i] >> 16] + (binrank[i] & 0xFFFFu);
            key_store<W>(&stage[pos], keys[i]);
            if (HAS_VAL) vstage[pos] = vals[i];
        }
    }
    // the reservations' results are awaited HERE, by every lane: the compiler otherwise puts the wait for them (vmcnt 0)
    // into the conditional blocks of the store loop below, where it makes every store wait for the one before
#pragma unroll
    for (int q = 0; q < BPT; ++q) asm volatile("" : "+v"(greserve[q]));
    {
        uint32_t ex = ex0;
#pragma unroll
        for (int q = 0; q < BPT; ++q) {
            const uint32_t bq = BPT * tid + q;
            if (bq < nb) {
                goff[bq] = greserve[q] - ex;
                if (L.slot_cap) {
                    // first staged position of this bin that no longer fits its slot (lhist is free by now)
                    const uint64_t slot_end = (gbin0 + bq) * (uint64_t)L.slot_stride +
                                              (L.xcd_shift ? (uint64_t)(xcc + 1u) * L.sub_cap : (uint64_t)L.slot_cap);
                    const int64_t room = (int64_t)slot_end - (int64_t)greserve[q];
                    lhist[bq] = (uint32_t)(int32_t)(room < -(int64_t)0x7FFF0000 ? -(int64_t)0x7FFF0000 : room) + ex;
                }
            }
            ex += cq[q];
        }
    }
    __syncthreads();
    BBK_PH(prof_kind, 3, t_prev);  // reorder into LDS

    // A bin whose slot is full (a k-mer repeated far beyond the coverage, a crowded bucket) spills.  Rare, and handled
    // after the stores: the spill counter's atomic returns a value, and a wait for it between the stores would make
    // every store wait for the one before.
    uint32_t full = 0;
    static_assert(ITEMS <= 32, "one bit per item");
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t pos = (uint32_t)(i * THREADS + tid);
        if (pos < staged) {
            const Key<W> key = key_load<W>(&stage[pos]);
            uint32_t pfx = prefix_of<W>(key, L.dmode, L.w0bits);
            (void)select_prefix(pfx, L);
            const uint32_t b = bin_of(pfx, L, nb);
            const uint32_t g = goff[b] + pos;
            if (L.slot_cap && (int32_t)pos >= (int32_t)lhist[b]) {
                full |= 1u << i;
            } else {
                key_store<W>(&out[g], key);
                if (HAS_VAL) vout[g] = vstage[pos];
            }
        }
    }
    if (full) {
#pragma unroll 1
        for (int i = 0; i < ITEMS; ++i) {
            if ((full >> i) & 1u) {
                const uint32_t pos = (uint32_t)(i * THREADS + tid);
                const uint32_t sp = atomicAdd(L.spill_count, 1u);
                if (sp < L.spill_cap) {
                    key_store<W>(&reinterpret_cast<Key<W> *>(L.spill_keys)[sp], key_load<W>(&stage[pos]));
                    if (HAS_VAL) L.spill_vals[sp] = vstage[pos];
                }
            }
        }
    }
    BBK_PH(prof_kind, 4, t_prev);  // store issue
#ifdef BBK_PHASE_PROF
    if (threadIdx.x == 0) atomicAdd(&g_phase[prof_kind][7], 1ull);
#endif
}

// Record of item i of a lane inside its tile.  8-byte keys: the items come in adjacent pairs, so that a full tile is
// read with 16-byte loads (global_load_dwordx4: half the load instructions of the 8-byte striping); the order of
// the records inside a tile is irrelevant (the scatter is unstable, the histogram a sum).
template <int W, int THREADS>
__device__ __forceinline__ uint32_t tile_local(int i, int tid) {
    if (W == 1) return (uint32_t)((((i >> 1) * THREADS + tid) << 1) | (i & 1));
    return (uint32_t)(i * THREADS + tid);
}

typedef uint64_t KeyPair __attribute__((ext_vector_type(2), aligned(8)));  // 16 bytes, 8-byte aligned

// all ITEMS records of a lane; every load is issued before the first use (a load inside a `local < count` branch
// would be waited for before the next one is issued: one memory latency per record)
template <int W, int ITEMS, int THREADS, bool HAS_VAL>
__device__ __forceinline__ void tile_load(const Key<W> *__restrict__ in, const uint32_t *__restrict__ vin, uint64_t begin,
                                          uint32_t count, int tid, Key<W> (&keys)[ITEMS], uint32_t (&vals)[ITEMS],
                                          int expand_k = 0, int expand_tag = 0) {
    if (expand_k) {  // uniform: record r of the tile = canonical key r/2 (even r) or its reverse complement (odd r)
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint32_t local = tile_local<W, THREADS>(i, tid);
            const uint64_t rec = begin + (local < count ? local : count - 1u);  // clamped into the tile
            const uint64_t at = rec >> 1;
            Key<W> x = key_load<W>(&in[at]);
            if (rec & 1) x = kmer_rc<W>(x, expand_k);
            if (W == 1 && expand_tag) x.w[0] |= __umul64hi(xxh3_64<W>(x), 16ull) << (2 * expand_k);
            keys[i] = x;
            vals[i] = HAS_VAL ? vin[at] : 0u;
        }
    } else if constexpr (W == 1) {
        if (count == (uint32_t)(ITEMS * THREADS)) {  // full tile (uniform): pairs
            const uint64_t *base = reinterpret_cast<const uint64_t *>(in) + begin;
#pragma unroll
            for (int i = 0; i < ITEMS; i += 2) {
                const uint32_t local = tile_local<W, THREADS>(i, tid);
                const KeyPair p = *reinterpret_cast<const KeyPair *>(base + local);
                keys[i].w[0] = p.x;
                keys[i + 1].w[0] = p.y;
                vals[i] = HAS_VAL ? vin[begin + local] : 0u;
                vals[i + 1] = HAS_VAL ? vin[begin + local + 1] : 0u;
            }
        } else {
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const uint32_t local = tile_local<W, THREADS>(i, tid);
                const uint64_t at = begin + (local < count ? local : count - 1u);  // clamped into the tile
                keys[i] = key_load<W>(&in[at]);
                vals[i] = HAS_VAL ? vin[at] : 0u;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint32_t local = tile_local<W, THREADS>(i, tid);
            const uint64_t at = begin + (local < count ? local : count - 1u);  // clamped into the tile
            keys[i] = key_load<W>(&in[at]);
            vals[i] = HAS_VAL ? vin[at] : 0u;
        }
    }
}

// One partition level over a key array.  HIST_ONLY: accumulate the level histogram; else scatter.
// LVL1: level-1 kernels have at most 512 bins (smaller LDS tables: two workgroups per CU)
template <int W, bool HAS_VAL, bool HIST_ONLY, bool LVL1>
__global__ __launch_bounds__(PartCfg<W>::THREADS) void k_part(const Key<W> *__restrict__ in,
                                                              const uint32_t *__restrict__ vin, TileMap M, PartLevel L,
                                                              uint32_t *__restrict__ ghist,   // HIST_ONLY: [nseg * nb]
                                                              uint32_t *__restrict__ cursor,  // scatter: running offsets
                                                              Key<W> *__restrict__ out, uint32_t *__restrict__ vout) {
    constexpr int kPartItems = PartCfg<W>::ITEMS, kPartTile = PartCfg<W>::TILE, kPartThreads = PartCfg<W>::THREADS;
    constexpr int MAXB = LVL1 ? 512 : kMaxBins;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // layout: lhist[MAXB] | lstart[MAXB] | goff[MAXB] | scan[32] | stage | vstage
    uint32_t *lhist = reinterpret_cast<uint32_t *>(smem);
    uint32_t *lstart = lhist + MAXB;
    uint32_t *goff = lstart + MAXB;
    uint32_t *scan_tmp = goff + MAXB;
    unsigned char *after = reinterpret_cast<unsigned char *>(scan_tmp + 32);
    Key<W> *stage = reinterpret_cast<Key<W> *>(after);
    uint32_t *vstage = reinterpret_cast<uint32_t *>(after + sizeof(Key<W>) * kPartTile);

    const int tid = threadIdx.x;
    if constexpr (HIST_ONLY) {
        // a workgroup walks `group` consecutive tiles and adds its LDS histogram to the global one when the
        // level-1 segment changes and at the end: one global atomic per (workgroup, bin), not per (tile, bin)
        const uint32_t t0 = blockIdx.x * M.group;
        const uint32_t t1 = t0 + M.group < M.ntiles ? t0 + M.group : M.ntiles;
        uint32_t nb = 0;
        uint64_t gbin0 = ~0ull;  // doubles as the identity of the current segment
        for (uint32_t t = t0; t < t1; ++t) {
            const TileInfo T = tile_info(M, L, t, (uint32_t)kPartTile);
            const uint64_t begin = T.begin;
            const uint32_t count = T.count;
            if (T.gbin0 != gbin0) {
                __syncthreads();
                for (uint32_t b = tid; b < nb; b += kPartThreads) {
                    const uint32_t c = lhist[b];
                    if (c) atomicAdd(&ghist[gbin0 + b], c);
                }
                __syncthreads();
                nb = T.nb;
                gbin0 = T.gbin0;
                for (uint32_t b = tid; b < nb; b += kPartThreads) lhist[b] = 0;
                __syncthreads();
            }
            Key<W> keys[kPartItems];
            uint32_t unused[kPartItems];
            tile_load<W, kPartItems, kPartThreads, false>(in, nullptr, begin, count, tid, keys, unused, M.expand_k,
                                                          M.expand_tag);
#pragma unroll
            for (int i = 0; i < kPartItems; ++i) {
                const uint32_t local = tile_local<W, kPartThreads>(i, tid);
                if (local < count) {
                    uint32_t pfx = prefix_of<W>(keys[i], L.dmode, L.w0bits);
                    if (select_prefix(pfx, L)) atomicAdd(&lhist[bin_of(pfx, L, nb)], 1u);
                }
            }
        }
        __syncthreads();
        for (uint32_t b = tid; b < nb; b += kPartThreads) {
            const uint32_t c = lhist[b];
            if (c) atomicAdd(&ghist[gbin0 + b], c);
        }
        return;
    }
#ifdef BBK_PHASE_PROF
    unsigned long long t_prev = clock64();
    const int prof_kind = LVL1 ? 1 : 2;
#else
    const unsigned long long t_prev = 0;
    const int prof_kind = 0;
#endif
    const TileInfo T = tile_info(M, L, blockIdx.x, (uint32_t)kPartTile);
    const uint64_t begin = T.begin, gbin0 = T.gbin0;  // gbin0: flat index of bin 0 in the cursor array
    const uint32_t count = T.count, nb = T.nb;
    if (count == 0) return;  // an unused place of the XCD-wise order of level-2 tiles (k_tile_desc)

    for (uint32_t b = tid; b < nb; b += kPartThreads) lhist[b] = 0;
    __syncthreads();
    BBK_PH(prof_kind, 0, t_prev);  // tile lookup

    Key<W> keys[kPartItems];
    uint32_t vals[kPartItems];
    uint32_t binrank[kPartItems];  // bin << 16 | rank (rank < 8192 fits 13 bits; bins < 1024)
    tile_load<W, kPartItems, kPartThreads, HAS_VAL>(in, vin, begin, count, tid, keys, vals, M.expand_k, M.expand_tag);
    // keep the records in registers: otherwise hipcc re-loads them from (restrict, read-only) memory for the LDS
    // reorder, which doubles the L2 traffic and, vmcnt being in-order, puts the reservation atomics issued in
    // between back on the critical path
#pragma unroll
    for (int i = 0; i < kPartItems; ++i) {
#pragma unroll
        for (int w = 0; w < W; ++w) asm volatile("" : "+v"(keys[i].w[w]));
        if (HAS_VAL) asm volatile("" : "+v"(vals[i]));
    }
#pragma unroll
    for (int i = 0; i < kPartItems; ++i) {
        const uint32_t local = tile_local<W, kPartThreads>(i, tid);
        binrank[i] = 0xFFFFFFFFu;
        if (local < count) {
            uint32_t pfx = prefix_of<W>(keys[i], L.dmode, L.w0bits);
            if (select_prefix(pfx, L)) {
                const uint32_t b = bin_of(pfx, L, nb);
                const uint32_t rank = atomicAdd(&lhist[b], 1u);
                binrank[i] = (b << 16) | rank;
            }
        }
    }
    __syncthreads();
    BBK_PH(prof_kind, 1, t_prev);  // load + LDS ranking
    part_tail<W, kPartItems, kPartThreads, MAXB, HAS_VAL>(keys, vals, binrank, lhist, lstart, goff, scan_tmp, stage, vstage,
                                                        nb, gbin0, L, cursor, out, vout, prof_kind, t_prev);
}

static size_t part_smem(int W, int tile, bool has_val, bool hist_only, bool lvl1) {
    size_t s = sizeof(uint32_t) * (3 * (lvl1 ? 512 : kMaxBins) + 32);
    if (!hist_only) s += (size_t)W * 8 * tile + (has_val ? 4 * (size_t)tile : 0);
    return s;
}

// ------------------------------------------------------------------------------------------
// fused k-mer extraction + level-1 partition over packed reads (HASH prefix)
// ------------------------------------------------------------------------------------------
// Instance space = chunks: read r contributes ceil(nk_r / CH) chunks of CH consecutive k-mer positions
// (the last one shorter); a lane owns one chunk, so it never crosses a read: one extraction, then (8-byte
// keys) every further k-mer is ROLLED from its predecessor -- with R = rev2(fwd) kept alongside one step is
// fwd = fwd>>2 | b<<2(k-1), R = R<<2 | b<<2(32-k), the reverse complement is (~R)>>pad and the canonical
// test is R <= (~fwd)<<pad: ~15 integer ops instead of a fresh extraction + bit reversal.
// The tile's reads (cursor tables + packed words, one coalesced copy) are staged in LDS first, so the lanes'
// dependent lookups (read of the chunk -> word offset -> words) cost LDS, not HBM, latency.  A tile whose
// reads do not fit (thousands of reads shorter than k in a row, words not laid out in read order) takes
// the same code over the global arrays.
struct ChunkWords {
    const uint64_t *rw;  // words of the chunk's read (LDS or global)
    uint32_t p;          // first k-mer position of the chunk
    uint32_t cnt;        // k-mers of the chunk (0: idle lane)
    uint32_t len;        // read length
};

// 64 bits of the packed read starting at base p (bases p .. p+31; words past `lastw` are not touched)
__device__ __forceinline__ uint64_t bases_from(const uint64_t *rw, uint32_t p, uint32_t lastw) {
    uint32_t wi = p >> 5;
    wi = wi <= lastw ? wi : lastw;
    const uint32_t sh = (p & 31u) << 1;
    const uint64_t lo = rw[wi];
    const uint64_t hi = rw[wi + 1 <= lastw ? wi + 1 : lastw];
    return (lo >> sh) | ((hi << 1) << (63u - sh));
}

template <int W, int CH, bool HAS_VAL>
__device__ __forceinline__ void chunk_records(const ChunkWords C, uint32_t k_, const PartLevel &L, uint32_t nb,
                                              uint32_t *lhist, Key<W> (&keys)[CH], uint32_t (&vals)[CH],
                                              uint32_t (&binrank)[CH]) {
    const uint64_t *rw = C.rw;
    // 8-byte keys, all state top-aligned so that every per-step shift is by a constant:
    //   Ft = fwd << pad (base 0 at bit pad, base k-1 at bits 62..63), Rv = rev2(fwd) (base 0 at the top)
    //   step: Ft = (Ft >> 2) & himask | b << 62,  Rv = Rv << 2 | b << pad
    //   canonical test (base-lexicographic fwd <= rc, rtseq.hpp:407-415): Rv <= ~Ft & himask
    // 16-byte keys (k = 33..64): the same with 128-bit state {hi, lo} -- Ft = F << pad (pad = 128 - 2k < 64, only
    // the low word has padding), Rv = {rev2(w0), rev2(w1)}; ~50 VALU per step against ~135 for a fresh extraction,
    // reverse complement and base-order comparison.
    const uint32_t pad = W == 1 ? 64u - 2u * k_ : (W == 2 ? 128u - 2u * k_ : 0u);
    const uint64_t himask = ~0ull << pad;
    uint64_t Ft = 0, Rv = 0;      // 8-byte keys; low words of the 128-bit state
    uint64_t FtH = 0, RvH = 0;    // high words (16-byte keys)
    uint32_t inb = 0;    // bases p+k, p+k+1, ...: the ones that enter (and the outgoing-edge bases)
    uint32_t prevb = 0;  // bases p-1, p, ...: the incoming-edge bases
    if (W == 1 && C.cnt) {
        const uint32_t lastw = (C.len - 1u) >> 5;
        const uint64_t f = bases_from(rw, C.p, lastw);
        Ft = f << pad;
        Rv = rev2(Ft >> pad);
        inb = (uint32_t)bases_from(rw, C.p + k_, lastw);
        if (HAS_VAL) prevb = ((uint32_t)f << 2) | (C.p ? base_at(rw, C.p - 1u) : 0u);
    }
    if (W == 2 && C.cnt) {
        const uint32_t lastw = (C.len - 1u) >> 5;
        const uint64_t w0 = bases_from(rw, C.p, lastw);                                   // bases p .. p+31
        const uint64_t w1 = (bases_from(rw, C.p + 32u, lastw) << pad) >> pad;              // bases p+32 .. p+k-1
        // F << pad as {hi, lo}
        FtH = pad ? (w1 << pad) | (w0 >> (64u - pad)) : w1;
        Ft = w0 << pad;
        RvH = rev2(w0);
        Rv = rev2(w1);
        inb = (uint32_t)bases_from(rw, C.p + k_, lastw);
        if (HAS_VAL) prevb = ((uint32_t)w0 << 2) | (C.p ? base_at(rw, C.p - 1u) : 0u);
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
#pragma unroll
        for (int w = 0; w < W; ++w) keys[i].w[w] = 0;
        vals[i] = 0;
        binrank[i] = 0xFFFFFFFFu;
        if ((uint32_t)i < C.cnt) {
            const uint32_t p = C.p + (uint32_t)i;
            bool minimal;
            uint32_t nextc, prevc;  // HAS_VAL: bases p+k and p-1
            if constexpr (W == 1) {
                if (i > 0) {
                    const uint64_t b = (inb >> (2 * (i - 1))) & 3u;
                    Ft = ((Ft >> 2) & himask) | (b << 62);
                    Rv = (Rv << 2) | (b << pad);
                }
                minimal = Rv <= (~Ft & himask);
                keys[i].w[0] = (minimal ? Ft : ~Rv) >> pad;
                nextc = (inb >> (2 * i)) & 3u;
                prevc = (prevb >> (2 * i)) & 3u;
            } else if constexpr (W == 2) {
                if (i > 0) {
                    const uint64_t b = (inb >> (2 * (i - 1))) & 3u;
                    Ft = ((Ft >> 2) | (FtH << 62)) & himask;
                    FtH = (FtH >> 2) | (b << 62);
                    RvH = (RvH << 2) | (Rv >> 62);
                    Rv = (Rv << 2) | (b << pad);
                }
                // canonical test: Rv <= ~Ft & himask128 as 128-bit numbers
                const uint64_t cH = ~FtH, cL = ~Ft & himask;
                minimal = RvH < cH || (RvH == cH && Rv <= cL);
                const uint64_t xH = minimal ? FtH : ~RvH, xL = minimal ? Ft : ~Rv;
                keys[i].w[0] = pad ? (xL >> pad) | (xH << (64u - pad)) : xL;
                keys[i].w[1] = xH >> pad;
                nextc = (inb >> (2 * i)) & 3u;
                prevc = (prevb >> (2 * i)) & 3u;
            } else {
                const Key<W> f = kmer_extract<W>(rw, p, (int)k_);
                const Key<W> rc = kmer_rc<W>(f, (int)k_);
                minimal = !kmer_less_nucl<W>(rc, f);
                keys[i] = key_select<W>(minimal, f, rc);
                if (HAS_VAL) {
                    nextc = p + k_ < C.len ? base_at(rw, p + k_) : 0u;
                    prevc = p >= 1 ? base_at(rw, p - 1) : 0u;
                }
            }
            if (HAS_VAL) {
                uint32_t m = 0;
                if (p + k_ < C.len) m |= 1u << (minimal ? nextc : 7u - nextc);
                if (p >= 1) m |= 1u << (minimal ? 4u + prevc : 3u - prevc);
                vals[i] = m;
            }
            uint32_t pfx = part_hash32<W>(keys[i]);
            if (select_prefix(pfx, L)) {
                const uint32_t b = L.b1 == 0 ? 0u : (pfx >> (32 - L.b1));
                const uint32_t rank = atomicAdd(&lhist[b], 1u);
                binrank[i] = (b << 16) | rank;
            }
        }
    }
}

template <int W, bool HAS_VAL, bool HIST_ONLY>
__global__ __launch_bounds__(HIST_ONLY ? kRdHistThreads : kRdThreads) void k_part_reads(ReadSrc S, PartLevel L, uint32_t *__restrict__ ghist,
                                                           uint32_t *__restrict__ cursor, Key<W> *__restrict__ out,
                                                           uint32_t *__restrict__ vout) {
    constexpr int NT = HIST_ONLY ? kRdHistThreads : kRdThreads;  // chunks per tile = threads
    constexpr int CH = RdCfg<W>::CH, TILE = RdCfg<W>::TILE, MAXB = 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // layout: lhist | lstart | goff | scan[32] | U, where U is the read tables while extracting
    // (rel[kRdSlots+2] | wrel[kRdSlots+2] | nk[kRdSlots+2] | words[kRdWords+W+2]) and the stage afterwards
    uint32_t *lhist = reinterpret_cast<uint32_t *>(smem);
    uint32_t *lstart = lhist + MAXB;
    uint32_t *goff = lstart + MAXB;
    uint32_t *scan_tmp = goff + MAXB;
    unsigned char *U = reinterpret_cast<unsigned char *>(scan_tmp + 32);
    int32_t *s_rel = reinterpret_cast<int32_t *>(U);  // first chunk of read r0+i, relative to the tile's first chunk
    int32_t *s_wrel = s_rel + (kRdSlots + 2);         // first word of read r0+i, relative to the staged window
    uint32_t *s_len = reinterpret_cast<uint32_t *>(s_wrel + (kRdSlots + 2));
    uint64_t *s_words = reinterpret_cast<uint64_t *>(s_len + (kRdSlots + 2));
    Key<W> *stage = reinterpret_cast<Key<W> *>(U);
    uint32_t *vstage = reinterpret_cast<uint32_t *>(U + sizeof(Key<W>) * TILE);

    const uint32_t tid = threadIdx.x;
    const uint32_t k_ = (uint32_t)S.k;
    const uint32_t nb = L.nb1;
    const uint32_t ntiles = (uint32_t)((S.n_chunks + NT - 1) / NT);
    for (uint32_t b = tid; b < nb; b += NT) lhist[b] = 0;
#ifdef BBK_PHASE_PROF
    unsigned long long t_prev = clock64();
#else
    const unsigned long long t_prev = 0;
#endif

    // scatter: one tile per workgroup (grid == tiles).  Histogram: a workgroup walks many tiles and adds
    // its LDS histogram to the global one once (512 atomics per workgroup instead of per tile).
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const uint64_t c0 = (uint64_t)tile * NT;  // first chunk of the tile
    const uint64_t left = S.n_chunks - c0;
    const uint32_t nch = left < (uint64_t)NT ? (uint32_t)left : (uint32_t)NT;
    const RdTile T = S.tiles[tile];
    const uint32_t r0 = T.r0, nr = T.nr;  // reads r0 .. r0+nr-1
    const uint64_t wbase = T.wbase;
    bool fast = T.wspan != 0xFFFFFFFFu;
    const uint32_t wspan = fast ? T.wspan : 0u;
    const uint64_t wend = wbase + wspan;
    if (fast) {
        bool bad = false;
        for (uint32_t i = tid; i <= nr; i += NT) {
            const uint64_t rr = (uint64_t)r0 + i;  // <= n_reads (coff holds n_reads + 1 entries)
            s_rel[i] = (int32_t)(int64_t)(S.coff[rr] - c0);
            if (i < nr) {
                const uint64_t wo = S.woff[rr];
                const uint32_t ln = S.len[rr];
                s_wrel[i] = (int32_t)(int64_t)(wo - wbase);
                s_len[i] = ln;
                // words must be laid out in read order: every read starts inside the window and all but
                // the last end inside it
                if (i > 0 && wo < wbase) bad = true;
                if (i + 1 < nr && wo + ((ln + 31u) >> 5) > wend) bad = true;
            }
        }
        for (uint32_t i = tid; i < wspan; i += NT) s_words[i] = S.words[wbase + i];
        fast = !__syncthreads_or(bad);
    } else {
        __syncthreads();
    }
    if (!HIST_ONLY) BBK_PH(0, 0, t_prev);  // read tables + words into LDS

    Key<W> keys[CH];
    uint32_t vals[CH];
    uint32_t binrank[CH];  // bin << 16 | rank (rank < 8192 fits 13 bits; bins < 512)
    if (fast) {
        ChunkWords C{s_words, 0, 0, 0};
        if (tid < nch) {
            const uint32_t ri = read_of(s_rel, nr, (int32_t)tid);
            C.p = (uint32_t)((int32_t)tid - s_rel[ri]) * CH;
            C.len = s_len[ri];
            const uint32_t nk = C.len - k_ + 1u;  // the read owns a chunk, so len >= k
            C.cnt = nk - C.p < (uint32_t)CH ? nk - C.p : (uint32_t)CH;
            C.rw = s_words + s_wrel[ri];
        }
        chunk_records<W, CH, HAS_VAL>(C, k_, L, nb, lhist, keys, vals, binrank);
    } else {
        ChunkWords C{S.words, 0, 0, 0};
        if (tid < nch) {
            const uint64_t c = c0 + tid;
            uint64_t lo = r0, hi = (uint64_t)r0 + nr;  // largest r with coff[r] <= c
            while (hi - lo > 1) {
                const uint64_t mid = (lo + hi) >> 1;
                if (S.coff[mid] <= c) lo = mid;
                else hi = mid;
            }
            C.p = (uint32_t)(c - S.coff[lo]) * CH;
            C.len = S.len[lo];
            const uint32_t nk = C.len - k_ + 1u;
            C.cnt = nk - C.p < (uint32_t)CH ? nk - C.p : (uint32_t)CH;
            C.rw = S.words + S.woff[lo];
        }
        chunk_records<W, CH, HAS_VAL>(C, k_, L, nb, lhist, keys, vals, binrank);
    }
    __syncthreads();  // histogram complete / the read tables may be overwritten

    if constexpr (!HIST_ONLY) {
        BBK_PH(0, 1, t_prev);  // extraction + LDS ranking
        part_tail<W, CH, NT, MAXB, HAS_VAL>(keys, vals, binrank, lhist, lstart, goff, scan_tmp, stage, vstage, nb,
                                                    0ull, L, cursor, out, vout, 0, t_prev);
        return;
    }
    }
    if (HIST_ONLY) {
        for (uint32_t b = tid; b < nb; b += NT) {
            const uint32_t c = lhist[b];
            if (c) atomicAdd(&ghist[b], c);
        }
    }
}

static size_t part_reads_smem(int W, bool has_val, bool hist_only) {
    const size_t tables = sizeof(uint32_t) * 3 * (kRdSlots + 2) + sizeof(uint64_t) * (kRdWords + W + 2);
    const size_t tile = (size_t)kRdThreads * (W == 1 ? 8 : (W == 2 ? 4 : 2));
    const size_t stage = hist_only ? 0 : (size_t)W * 8 * tile + (has_val ? 4 * tile : 0);
    return sizeof(uint32_t) * (3 * 512 + 32) + std::max(tables, stage);
}

// ------------------------------------------------------------------------------------------
// bucket kernel
// ------------------------------------------------------------------------------------------
__device__ inline uint64_t match8(uint32_t d, bool valid) {
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool bit = (d >> b) & 1u;
        const uint64_t m = __ballot(bit);
        peers &= bit ? m : ~m;
    }
    return peers;
}

struct BucketArgs {
    const uint32_t *boff;        // nbuckets + 1 record offsets
    uint32_t *dcount;            // distinct per bucket; 0xFFFFFFFF = overflow (left untouched)
    const uint32_t *bucket_ids;  // null: bucket = blockIdx.x; else the list of buckets to process
    int k;
    uint32_t *dbg;               // optional counters (BBK_VERBOSE): [0] buckets that took the all-words fallback
    // slot mode: bucket b lies at [b*slot_cap, b*slot_cap + min(reserved, slot_cap)), reserved = cursor[b] - b*slot_cap;
    // a bucket that reserved more than its slot is left alone (dcount = 0xFFFFFFFF): the host reprocesses it together
    // with the spill list
    uint32_t slot_cap, slot_stride;
    const uint32_t *cursor;
    // (the hash-dedup kernels write the distinct records back to the head of their bucket; until round 3 they could also
    // reserve a place in the dense result with an atomicAdd on one counter -- 2.6 ms for the 227 210 buckets of BASELINE
    // configs[1], tools/probes/single_counter_probe.hip)
    // sorting kernels, input known to hold (almost certainly) no duplicates: bucket b's records go straight to
    // sorted_keys[boff[b] ...] (the dense result: same offsets as the input when nothing is removed), word 0 masked
    // with strip_mask; a bucket that did remove a duplicate raises *dup_flag and the caller redoes the pass in place
    void *sorted_keys;
    uint32_t *sorted_vals;
    uint32_t *dup_flag;
    uint64_t strip_mask;
    // hash-dedup kernels: a probe sequence longer than this means the table is (nearly) full and the bucket is left to
    // the caller (kHashMaxProbes; tests lower it through BBK_HASH_MAX_PROBES to force that path on half-empty slots)
    uint32_t max_probes;
    // sorting kernels reading slots (stage B without histograms): bucket b's sorted records go to sorted_keys[out_off[b]
    // ...] (exclusive scan of the slot fills); null: the dense layout, output offset = input offset
    const uint32_t *out_off;
};

// first record and record count of bucket b (count 0xFFFFFFFF: the slot overflowed)
__device__ inline void bucket_range(const BucketArgs &A, uint32_t b, uint32_t *start, uint32_t *n) {
    if (A.slot_cap) {
        *start = b * A.slot_stride;
        const uint32_t reserved = A.cursor[b] - *start;
        *n = reserved > A.slot_cap ? 0xFFFFFFFFu : reserved;
    } else {
        *start = A.boff[b];
        *n = A.boff[b + 1] - *start;
    }
}

// OP: 0 unique only, 1 COUNT (run length), 2 SUM of vals, 3 OR of vals.  NT threads, CAP = NT * ITEMS.
// Heads + segmented reduce of a bucket that lies sorted in LDS (skeys[0, n), svals alongside when the records
// carry a payload); the distinct records are written back in place at buf[start ...], their reduced payloads
// to vals, the count to dcount[b].  Blocked ownership: thread t owns [t*ITEMS, (t+1)*ITEMS).
template <int W, int NT, int ITEMS, int OP>
__device__ __forceinline__ void bucket_reduce(Key<W> *skeys, uint32_t *svals, uint32_t *scan_tmp, uint32_t n, uint32_t start,
                                              uint32_t b, Key<W> *__restrict__ buf, uint32_t *__restrict__ vals,
                                              const BucketArgs &A) {
    constexpr int NWAVES = NT / 64;
    constexpr bool IN_VAL = OP >= 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t ostart = (A.sorted_keys && A.out_off) ? A.out_off[b] : start;  // where the sorted records go
    Key<W> mine[ITEMS];
    uint32_t mv[ITEMS];
    const uint32_t p0 = (uint32_t)tid * ITEMS;
    Key<W> prev;
#pragma unroll
    for (int j = 0; j < W; ++j) prev.w[j] = ~0ull;  // cannot equal a real key: unused high bits are 0
    if (p0 > 0 && p0 - 1 < n) prev = key_load<W>(&skeys[p0 - 1]);
    uint32_t nheads = 0;
    uint32_t headbits = 0;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
#pragma unroll
        for (int j = 0; j < W; ++j) mine[i].w[j] = 0;
        mv[i] = 0;
        if (p0 + i < n) {
            mine[i] = key_load<W>(&skeys[p0 + i]);
            if (IN_VAL) mv[i] = svals[p0 + i];
            const bool h = (i == 0) ? !key_eq<W>(mine[0], prev) : !key_eq<W>(mine[i], mine[i - 1]);
            if (h) {
                headbits |= 1u << i;
                ++nheads;
            }
        }
    }
    uint32_t excl, total;
    {
        uint32_t incl = nheads;
        incl = wave_scan_incl(incl);
        __syncthreads();  // everyone has its keys in registers: skeys may be reused below
        if (lane == 63) scan_tmp[wave] = incl;
        __syncthreads();
        uint32_t wbase, tot;
        wave_totals<NWAVES>(scan_tmp, lane, wave, wbase, tot);
        excl = wbase + incl - nheads;
        total = tot;
        // the loaded offset is awaited HERE by every lane: left to the compiler, the wait (vmcnt 0) lands in the
        // conditional blocks of the store loop below and makes every store wait for the one before
        asm volatile("" : "+v"(ostart));
    }
#ifndef BBK_AB_BLOCKED_REDUCE  // (A/B: -DBBK_AB_BLOCKED_REDUCE stores straight from the blocked ownership, as before round 3)
    if constexpr (OP == 0) {
        // No payload: the distinct keys go back into LDS at their place in the result (a place at or before the thread's
        // own records, all of which are in registers by now) and leave it with coalesced stores -- 512 contiguous bytes
        // per wave instruction.  Straight from the blocked ownership every lane stored its ITEMS keys 8 ITEMS bytes from
        // its neighbour's: 64 separate pieces per instruction.
        int seg = (int)excl - 1;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            if (p0 + i < n && (headbits & (1u << i))) {
                ++seg;
                Key<W> kx = mine[i];
                if (A.sorted_keys) kx.w[0] &= A.strip_mask;
                key_store<W>(&skeys[seg], kx);
            }
        }
        __syncthreads();
        Key<W> *dstk = A.sorted_keys ? reinterpret_cast<Key<W> *>(A.sorted_keys) + ostart : buf + start;
        for (uint32_t s = tid; s < total; s += NT) key_store<W>(&dstk[s], key_load<W>(&skeys[s]));
        if (tid == 0 && A.sorted_keys && total != n) atomicOr(A.dup_flag, 1u);
        if (tid == 0) A.dcount[b] = total;
        return;
    }
#endif
    uint32_t *acc = reinterpret_cast<uint32_t *>(skeys);  // CAP u32 fit in the key buffer
    if (OP != 0) {
        for (uint32_t s = tid; s < total; s += NT) acc[s] = 0;
        __syncthreads();
    }
    {
        int seg = (int)excl - 1;  // segment of the records before my first head
        uint32_t a = 0;
        bool any = false;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            if (p0 + i < n) {
                if (headbits & (1u << i)) {
                    if (OP != 0 && any) {
                        if (OP == 3) atomicOr(&acc[seg], a);
                        else atomicAdd(&acc[seg], a);
                    }
                    ++seg;
                    a = 0;
                    if (A.sorted_keys) {
                        Key<W> kx = mine[i];
                        kx.w[0] &= A.strip_mask;
                        key_store<W>(&reinterpret_cast<Key<W> *>(A.sorted_keys)[ostart + (uint32_t)seg], kx);
                    } else {
                        key_store<W>(&buf[start + (uint32_t)seg], mine[i]);  // distinct keys, in place
                    }
                }
                any = true;
                if (OP == 1) a += 1;
                else if (OP == 2) a += mv[i];
                else if (OP == 3) a |= mv[i];
            }
        }
        if (OP != 0 && any) {
            if (OP == 3) atomicOr(&acc[seg], a);
            else atomicAdd(&acc[seg], a);
        }
    }
    if (OP != 0) {
        __syncthreads();
        uint32_t *vdst = A.sorted_keys ? A.sorted_vals : vals;
        const uint32_t vstart = A.sorted_keys ? ostart : start;
        for (uint32_t s = tid; s < total; s += NT) vdst[vstart + s] = acc[s];
    }
    if (tid == 0 && A.sorted_keys && total != n) atomicOr(A.dup_flag, 1u);
    if (tid == 0) A.dcount[b] = total;
}

template <int W, int NT, int ITEMS, int OP>
__global__ __launch_bounds__(NT) void k_bucket(Key<W> *__restrict__ buf, uint32_t *__restrict__ vals, BucketArgs A) {
    constexpr int CAP = NT * ITEMS;
    constexpr int NWAVES = NT / 64;
    constexpr bool IN_VAL = OP >= 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // layout: wave_cnt[NWAVES][256] | dstart[256] | scan[32] | skeys[CAP] | svals[CAP] (IN_VAL)
    uint32_t(*wave_cnt)[256] = reinterpret_cast<uint32_t(*)[256]>(smem);
    uint32_t *dstart = reinterpret_cast<uint32_t *>(smem) + NWAVES * 256;
    uint32_t *scan_tmp = dstart + 256;
    Key<W> *skeys = reinterpret_cast<Key<W> *>(scan_tmp + 32);
    uint32_t *svals = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(skeys) + sizeof(Key<W>) * CAP);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t b = A.bucket_ids ? A.bucket_ids[blockIdx.x] : blockIdx.x;
    uint32_t start, n;
    bucket_range(A, b, &start, &n);
    if (n == 0) {
        if (tid == 0) A.dcount[b] = 0;
        return;
    }
    if (n > (uint32_t)CAP) {
        if (tid == 0) A.dcount[b] = 0xFFFFFFFFu;
        return;
    }
    {
        // all loads first (unconditional, index clamped into the bucket), then the LDS stores: a load inside the
        // `p < n` branch is waited for before the next one is issued -- one memory latency per record
        Key<W> rk[ITEMS];
        uint32_t rv[ITEMS];
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint32_t p = (uint32_t)(i * NT + tid);
            const uint32_t at = start + (p < n ? p : n - 1u);
            rk[i] = key_load<W>(&buf[at]);
            rv[i] = IN_VAL ? vals[at] : 0u;
        }
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint32_t p = (uint32_t)(i * NT + tid);
            if (p < n) {
                key_store<W>(&skeys[p], rk[i]);
                if (IN_VAL) svals[p] = rv[i];
            }
        }
    }
    __syncthreads();

    // ---- LSD radix sort inside LDS.  Only word 0 is radix-sorted, and only over the bits in which the
    // bucket's keys differ (keys of a KEYS-mode bucket share their top ~16 bits): subtract the bucket
    // minimum, sort the bits of (max - min).  Wider keys then order the (short) runs of equal word 0 by
    // their remaining words with an insertion sort; a bucket with a long run (> 48 keys sharing 32
    // bases) falls back to radix passes over every word.
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int lastbits = 2 * A.k - 64 * (W - 1);
    uint64_t kmin = 0;
    int sortbits = (W == 1) ? lastbits : 64;
    {
        uint64_t mn = ~0ull, mx = 0;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint32_t p = (uint32_t)(i * NT + tid);
            if (p < n) {
                const uint64_t x = skeys[p].w[0];
                mn = x < mn ? x : mn;
                mx = x > mx ? x : mx;
            }
        }
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) {
            const uint64_t a = __shfl_xor(mn, dd, 64), c = __shfl_xor(mx, dd, 64);
            mn = a < mn ? a : mn;
            mx = c > mx ? c : mx;
        }
        uint64_t *mm = reinterpret_cast<uint64_t *>(wave_cnt);  // counters are not live yet
        if (lane == 0) {
            mm[2 * wave] = mn;
            mm[2 * wave + 1] = mx;
        }
        __syncthreads();
        mn = ~0ull;
        mx = 0;
        for (int j = 0; j < NWAVES; ++j) {
            mn = mm[2 * j] < mn ? mm[2 * j] : mn;
            mx = mm[2 * j + 1] > mx ? mm[2 * j + 1] : mx;
        }
        __syncthreads();
        kmin = mn;
        sortbits = 64 - __builtin_clzll((mx - mn) | 1ull);
    }
    // stable radix passes over bits [0, nbits) of (word `wsel` - base)
    auto radix_passes = [&](int wsel, uint64_t base, int nbits) {
        for (int shift = 0; shift < nbits; shift += 8) {
            Key<W> keys[ITEMS];
            uint32_t v[ITEMS];
            uint32_t dr[ITEMS];  // digit << 16 | rank-in-wave
            if (tid < 256) {
#pragma unroll
                for (int j = 0; j < NWAVES; ++j) wave_cnt[j][tid] = 0;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const uint32_t p = (uint32_t)(wave * (ITEMS * 64) + i * 64 + lane);
                const bool valid = p < n;
                uint32_t d = 0;
#pragma unroll
                for (int j = 0; j < W; ++j) keys[i].w[j] = 0;
                v[i] = 0;
                if (valid) {
                    keys[i] = key_load<W>(&skeys[p]);
                    if (IN_VAL) v[i] = svals[p];
                    const uint64_t word = (W == 1) ? keys[i].w[0]
                                                   : reinterpret_cast<const uint64_t *>(&skeys[p])[wsel];
                    d = (uint32_t)((word - base) >> shift) & 0xFFu;
                }
                const uint64_t peers = match8(d, valid);
                const uint32_t pre = wave_cnt[wave][d];
                dr[i] = (d << 16) | (pre + (uint32_t)__popcll(peers & lt_mask));
                if (valid && (peers >> lane) == 1ull) wave_cnt[wave][d] = pre + (uint32_t)__popcll(peers);
            }
            __syncthreads();
            {
                uint32_t tot = 0, incl = 0;
                if (tid < 256) {
#pragma unroll
                    for (int j = 0; j < NWAVES; ++j) {
                        const uint32_t c = wave_cnt[j][tid];
                        wave_cnt[j][tid] = tot;
                        tot += c;
                    }
                    incl = tot;
                    incl = wave_scan_incl(incl);
                    if (lane == 63) scan_tmp[wave] = incl;
                }
                __syncthreads();
                if (tid < 256) {
                    uint32_t wbase = 0;
                    for (int j = 0; j < wave; ++j) wbase += scan_tmp[j];
                    dstart[tid] = wbase + incl - tot;
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const uint32_t p = (uint32_t)(wave * (ITEMS * 64) + i * 64 + lane);
                if (p < n) {
                    const uint32_t d = dr[i] >> 16;
                    const uint32_t pos = dstart[d] + wave_cnt[wave][d] + (dr[i] & 0xFFFFu);
                    key_store<W>(&skeys[pos], keys[i]);
                    if (IN_VAL) svals[pos] = v[i];
                }
            }
            __syncthreads();
        }
    };
    radix_passes(0, kmin, sortbits);
    if (W >= 2) {
        // runs of equal word 0: the thread that owns a run's first record orders the run by words 1..W-1
        bool bad = false;
        const uint32_t q0 = (uint32_t)tid * ITEMS;
        for (uint32_t p = q0; p < q0 + ITEMS && p < n; ++p) {
            const uint64_t w0 = skeys[p].w[0];
            if (p > 0 && skeys[p - 1].w[0] == w0) continue;  // not a run start
            uint32_t e = p + 1;
            while (e < n && skeys[e].w[0] == w0) ++e;
            if (e - p <= 1) continue;
            if (e - p > 48) {
                bad = true;
                continue;
            }
            for (uint32_t x = p + 1; x < e; ++x) {
                const Key<W> kx = key_load<W>(&skeys[x]);
                const uint32_t vx = IN_VAL ? svals[x] : 0u;
                uint32_t y = x;
                while (y > p) {
                    const Key<W> ky = key_load<W>(&skeys[y - 1]);
                    if (!key_less_words<W>(kx, ky)) break;
                    key_store<W>(&skeys[y], ky);
                    if (IN_VAL) svals[y] = svals[y - 1];
                    --y;
                }
                key_store<W>(&skeys[y], kx);
                if (IN_VAL) svals[y] = vx;
            }
        }
        if (__syncthreads_or(bad)) {
            if (A.dbg && tid == 0) atomicAdd(&A.dbg[0], 1u);
            for (int w = W - 1; w >= 0; --w) radix_passes(w, 0ull, (w == W - 1) ? lastbits : 64);
        }
    }

    bucket_reduce<W, NT, ITEMS, OP>(skeys, svals, scan_tmp, n, start, b, buf, vals, A);
}

// ---- first-choice bucket kernel: ONE distribution pass instead of ballot-ranked radix passes.
// The records of a bucket are spread evenly over its key range (KEYS/REF mode: a contiguous range of k-mers
// of a genome), so DistBins::N bins over the top bits of (word 0 - bucket minimum) hold about one record
// each: count with LDS atomics, scan, scatter with returning atomics (the order inside a bin is arbitrary),
// then the owner of a bin puts it in order by insertion on the whole key.  A bin above kDistMaxBin (skewed
// keys, a k-mer repeated hundreds of times in an unreduced stream) marks the bucket as overflowing and the
// host hands it to k_bucket, which takes any distribution.
// 4096 bins; wide keys WITH a payload: 2048, so that keys + payloads + bins stay below 80 KB and TWO workgroups fit a CU
// (16-byte keys: 57 + 14 + 8 KB).  With 4096 bins the sort of an extension index of 16-byte keys (k-mer + edge mask) ran
// one workgroup per CU: 8.1 ms against 5.0 ms for 257 M records.  (8-byte keys with a payload stay at 4096 bins and
// one workgroup per CU: with 5632 records per bucket the fuller bins cost more than the second workgroup gains.)
template <int W, int OP>
struct DistBins {
    static constexpr int N = (W >= 2 && OP >= 2) ? 2048 : 4096;
    static constexpr int LOG = (W >= 2 && OP >= 2) ? 11 : 12;
};
constexpr uint32_t kDistMaxBin = 96;  // equal keys insert in linear time; only distinct keys cost n^2

template <int W, int NT, int ITEMS, int OP>
__global__ __launch_bounds__(NT) void k_bucket_dist(Key<W> *__restrict__ buf, uint32_t *__restrict__ vals, BucketArgs A) {
    constexpr int CAP = NT * ITEMS;
    constexpr int NWAVES = NT / 64;
    constexpr int DB = DistBins<W, OP>::N;
    constexpr int BPT = DB / NT;
    constexpr bool IN_VAL = OP >= 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // layout: bins[DB] | scan[32] | mm[2 * NWAVES] (u64) | skeys[CAP] | svals[CAP] (IN_VAL)
    uint32_t *bins = reinterpret_cast<uint32_t *>(smem);
    uint32_t *scan_tmp = bins + DB;
    uint64_t *mm = reinterpret_cast<uint64_t *>(scan_tmp + 32);
    Key<W> *skeys = reinterpret_cast<Key<W> *>(mm + 2 * NWAVES);
    uint32_t *svals = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(skeys) + sizeof(Key<W>) * CAP);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t b = A.bucket_ids ? A.bucket_ids[blockIdx.x] : blockIdx.x;
    uint32_t start, n;
    bucket_range(A, b, &start, &n);
    if (n == 0) {
        if (tid == 0) A.dcount[b] = 0;
        return;
    }
    if (n > (uint32_t)CAP) {
        if (tid == 0) A.dcount[b] = 0xFFFFFFFFu;
        return;
    }
    for (uint32_t q = tid; q < (uint32_t)DB; q += NT) bins[q] = 0;
#ifdef BBK_PHASE_PROF
    unsigned long long t_prev = clock64();
#endif

    // records of this thread (striped over the bucket); all loads issued before the first use
    Key<W> keys[ITEMS];
    uint32_t v[IN_VAL ? ITEMS : 1];
    uint64_t mn = ~0ull, mx = 0;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t p = (uint32_t)(i * NT + tid);
        const uint32_t at = start + (p < n ? p : n - 1u);
        keys[i] = key_load<W>(&buf[at]);
        if (IN_VAL) v[i] = vals[at];
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {  // clamped duplicates do not change min / max
        const uint64_t x = keys[i].w[0];
        mn = x < mn ? x : mn;
        mx = x > mx ? x : mx;
    }
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) {
        const uint64_t a = __shfl_xor(mn, dd, 64), c = __shfl_xor(mx, dd, 64);
        mn = a < mn ? a : mn;
        mx = c > mx ? c : mx;
    }
    if (lane == 0) {
        mm[2 * wave] = mn;
        mm[2 * wave + 1] = mx;
    }
    __syncthreads();  // bins zeroed, min / max of every wave visible
    mn = ~0ull;
    mx = 0;
#pragma unroll
    for (int j = 0; j < NWAVES; ++j) {
        mn = mm[2 * j] < mn ? mm[2 * j] : mn;
        mx = mm[2 * j + 1] > mx ? mm[2 * j + 1] : mx;
    }
    BBK_PH(3, 0, t_prev);  // loads + min/max
    const uint64_t kmin = mn;
    const int rbits = 64 - __builtin_clzll((mx - mn) | 1ull);
    const int sh = rbits > DistBins<W, OP>::LOG ? rbits - DistBins<W, OP>::LOG : 0;  // digit = (word 0 - min) >> sh < bins

#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t p = (uint32_t)(i * NT + tid);
        if (p < n) atomicAdd(&bins[(uint32_t)((keys[i].w[0] - kmin) >> sh)], 1u);
    }
    __syncthreads();
    BBK_PH(3, 1, t_prev);  // count
    // exclusive scan of the bins; thread t owns bins [t*BPT, (t+1)*BPT)
    uint32_t c[BPT];
    uint32_t sum = 0;
    bool big = false;
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
        c[q] = bins[tid * BPT + q];
        sum += c[q];
        big = big || c[q] > kDistMaxBin;
    }
    uint32_t incl = sum;
    incl = wave_scan_incl(incl);
    if (lane == 63) scan_tmp[wave] = incl;
    if (__syncthreads_or(big)) {  // nothing has been written: the second-chance kernel takes the bucket
        if (tid == 0) A.dcount[b] = 0xFFFFFFFFu;
        return;
    }
    uint32_t first = incl - sum;
    for (int j = 0; j < wave; ++j) first += scan_tmp[j];
    {
        uint32_t ex = first;
#pragma unroll
        for (int q = 0; q < BPT; ++q) {
            bins[tid * BPT + q] = ex;
            ex += c[q];
        }
    }
    __syncthreads();
    BBK_PH(3, 2, t_prev);  // scan
    uint32_t pos_of[ITEMS];  // where the scatter put the record (breaks ties between equal keys)
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t p = (uint32_t)(i * NT + tid);
        pos_of[i] = 0;
        if (p < n) {
            const uint32_t pos = atomicAdd(&bins[(uint32_t)((keys[i].w[0] - kmin) >> sh)], 1u);
            key_store<W>(&skeys[pos], keys[i]);
            pos_of[i] = pos;
        }
    }
    __syncthreads();
    BBK_PH(3, 3, t_prev);  // scatter
    // order inside the bins, record-parallel: a record's final place is its bin's start plus the number of
    // records of the bin that go before it (smaller key; equal key: scattered to a lower position).  After the
    // scatter bins[d] is the END of bin d, so bin d = [bins[d-1], bins[d]).
    // Batched so that the LDS reads of several records are in flight together (one record at a time is three
    // dependent LDS round trips: bin bounds, candidates, compare): RB records per round, the first four candidates
    // of every bin read unconditionally; the rare fuller bins finish in a loop.
    uint32_t dest[ITEMS];
    if constexpr (W == 1) {
        constexpr int RB = 6;
    #pragma unroll
        for (int i0 = 0; i0 < ITEMS; i0 += RB) {
            uint32_t sb[RB], e[RB];
    #pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int i = i0 + u;
                sb[u] = e[u] = 0;
                if (i < ITEMS) {
                    const uint32_t p = (uint32_t)(i * NT + tid);
                    if (p < n) {
                        const uint32_t d = (uint32_t)((keys[i].w[0] - kmin) >> sh);
                        sb[u] = d ? bins[d - 1] : 0u;
                        e[u] = bins[d];
                    }
                }
            }
            Key<W> o[RB][4];
    #pragma unroll
            for (int u = 0; u < RB; ++u) {
    #pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint32_t y = sb[u] + c;
                    o[u][c] = key_load<W>(&skeys[y < e[u] ? y : (e[u] ? e[u] - 1u : 0u)]);
                }
            }
    #pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int i = i0 + u;
                if (i < ITEMS) {
                    dest[i] = 0xFFFFFFFFu;
                    const uint32_t p = (uint32_t)(i * NT + tid);
                    if (p < n) {
                        uint32_t before = 0;
    #pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const uint32_t y = sb[u] + c;
                            if (y < e[u]) {
                                const bool lt = key_less_words<W>(o[u][c], keys[i]);
                                const bool eq = key_eq<W>(o[u][c], keys[i]);
                                before += (lt || (eq && y < pos_of[i])) ? 1u : 0u;
                            }
                        }
                        for (uint32_t y = sb[u] + 4; y < e[u]; ++y) {  // bins above four records
                            const Key<W> ok = key_load<W>(&skeys[y]);
                            const bool lt = key_less_words<W>(ok, keys[i]);
                            const bool eq = key_eq<W>(ok, keys[i]);
                            before += (lt || (eq && y < pos_of[i])) ? 1u : 0u;
                        }
                        dest[i] = sb[u] + before;
                    }
                }
            }
        }
    } else {
        // wider keys: one record at a time, four candidates in flight (the batched form costs more registers than
        // it saves: measured 8 % slower for 16-byte keys)
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint32_t p = (uint32_t)(i * NT + tid);
            dest[i] = 0xFFFFFFFFu;
            if (p < n) {
                const uint32_t d = (uint32_t)((keys[i].w[0] - kmin) >> sh);
                const uint32_t sb = d ? bins[d - 1] : 0u, e = bins[d];
                uint32_t before = 0;
                if (e - sb > 1) {
                    for (uint32_t y = sb; y < e; y += 4) {
                        Key<W> o[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) o[u] = key_load<W>(&skeys[y + u < e ? y + u : e - 1]);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (y + u < e) {
                                const bool lt = key_less_words<W>(o[u], keys[i]);
                                const bool eq = key_eq<W>(o[u], keys[i]);
                                before += (lt || (eq && y + u < pos_of[i])) ? 1u : 0u;
                            }
                        }
                    }
                }
                dest[i] = sb + before;
            }
        }
    }
    __syncthreads();  // every rank is computed from the scattered order: only now overwrite it
    BBK_PH(3, 4, t_prev);  // rank
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        if (dest[i] != 0xFFFFFFFFu) {
            key_store<W>(&skeys[dest[i]], keys[i]);
            if (IN_VAL) svals[dest[i]] = v[i];
        }
    }
    __syncthreads();
    BBK_PH(3, 5, t_prev);  // write
    bucket_reduce<W, NT, ITEMS, OP>(skeys, svals, scan_tmp, n, start, b, buf, vals, A);
    BBK_PH(3, 6, t_prev);  // heads + reduce + output
#ifdef BBK_PHASE_PROF
    if (threadIdx.x == 0) atomicAdd(&g_phase[3][7], 1ull);
#endif
}

template <int W, int NT, int ITEMS, int OP>
static size_t bucket_dist_smem() {
    return sizeof(uint32_t) * (DistBins<W, OP>::N + 32) + sizeof(uint64_t) * 2 * (NT / 64) + (size_t)W * 8 * NT * ITEMS +
           (OP >= 2 ? 4 * NT * ITEMS : 0);
}

template <int W, int NT, int ITEMS, int OP>
static size_t bucket_smem() {
    return sizeof(uint32_t) * ((NT / 64) * 256 + 256 + 32) + (size_t)W * 8 * NT * ITEMS + (OP >= 2 ? 4 * NT * ITEMS : 0);
}

// ---- dedup by an LDS hash table (8-byte keys): when the caller only needs the distinct set (the
// hash-partitioned first stage: a second stage sorts the survivors anyway) the bucket does not have
// to be sorted.  Records are streamed from HBM straight into an open-addressing table with 64-bit
// ds_cmpst; with 50x coverage ~8 of 9 records find their key already there on the first probe.
// ~30 instructions per record instead of 6 radix passes.  The distinct keys (+ reduced payload)
// are written back in place in table order.
#ifdef BBK_AB_TABLE_WALK  // (A/B: the distinct keys always collected by a walk over the table's slots, as before round 3)
constexpr bool kHashDirectOut = false;
#else
constexpr bool kHashDirectOut = true;
#endif
constexpr int kHashThreads = 512;
#ifndef BBK_HASH_ITEMS
#define BBK_HASH_ITEMS 16
#endif
constexpr int kHashItems = BBK_HASH_ITEMS;          // 512 x 16 = 8192 records per bucket (x 12: 2 % slower, and the
                                                    // fullest bucket of a 10 M-read batch then overflows its slot)
constexpr uint32_t kHashSlots = 8192;               // distinct keys of a bucket: ~n / multiplicity, far below the slots
                                                    // for read data; all-distinct input fills ~0.7 of them
constexpr uint32_t kHashMaxProbes = 256;            // a probe sequence this long means the table is (nearly) full: the
                                                    // bucket holds more distinct keys than slots -> left to the caller

template <int OP>
__global__ __launch_bounds__(kHashThreads) void k_bucket_hash(Key<1> *__restrict__ buf, uint32_t *__restrict__ vals,
                                                             BucketArgs A) {
    constexpr bool IN_VAL = OP >= 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long *tab = reinterpret_cast<unsigned long long *>(smem);
    uint32_t *pay = reinterpret_cast<uint32_t *>(smem + sizeof(unsigned long long) * kHashSlots);
    uint32_t *scan_tmp = pay + (OP != 0 ? kHashSlots : 0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t b = A.bucket_ids ? A.bucket_ids[blockIdx.x] : blockIdx.x;
    uint32_t start, n;
    bucket_range(A, b, &start, &n);
    if (n == 0) {
        if (tid == 0) A.dcount[b] = 0;
        return;
    }
    if (n > (uint32_t)(kHashThreads * kHashItems)) {
        if (tid == 0) A.dcount[b] = 0xFFFFFFFFu;
        return;
    }
    constexpr unsigned long long EMPTY = ~0ull;
#ifdef BBK_PHASE_PROF
    unsigned long long t_prev = clock64();
#endif
    uint64_t kk[kHashItems];
    uint32_t vv[kHashItems];
#pragma unroll
    for (int i = 0; i < kHashItems; ++i) {  // all loads first: independent, in flight together -- and while the table
        const uint32_t p = (uint32_t)(i * kHashThreads + tid);  // is cleared below
        kk[i] = EMPTY;
        vv[i] = 0;
        if (p < n) {
            kk[i] = buf[start + p].w[0];
            if (IN_VAL) vv[i] = vals[start + p];
        }
    }
    for (uint32_t s = tid; s < kHashSlots; s += kHashThreads) {
        tab[s] = EMPTY;
        if (OP != 0) pay[s] = 0;
    }
    if (tid == 0) scan_tmp[14] = 0;
    __syncthreads();
    BBK_PH(4, 0, t_prev);  // table init
    uint32_t firsts = 0;  // bit i: record i of this lane was the first of its key in the table
    static_assert(kHashItems <= 32, "one bit per record of a lane");
#pragma unroll
    for (int i = 0; i < kHashItems; ++i) {
        if (kk[i] != EMPTY) {
            // 32-bit mix with multipliers of its own (the partition levels consumed the top bits of part_hash32)
            uint32_t h = ((uint32_t)kk[i] ^ 0x7F4A7C15u) * 0x2C1B3C6Du;
            h ^= h >> 15;
            h += (uint32_t)(kk[i] >> 32) * 0x297A2D39u;
            h ^= h >> 14;
            h *= 0x9E3779B1u;
            uint32_t slot = (h >> 19) & (kHashSlots - 1);
            // (probing all records of a lane in rounds, 12 ds_cmpst in flight, was measured 15 % slower)
            uint32_t probes = 0;
            for (;;) {
                const unsigned long long old = atomicCAS(&tab[slot], EMPTY, (unsigned long long)kk[i]);
                if (old == EMPTY) firsts |= 1u << i;
                if (old == EMPTY || old == kk[i]) break;
                slot = (slot + 1) & (kHashSlots - 1);
                if (kHashItems * kHashThreads > (int)(kHashSlots * 3 / 4) && ++probes > A.max_probes) {
                    scan_tmp[14] = 1;  // give up on this bucket (benign race: everyone writes 1)
                    break;
                }
            }
            if (OP == 1) atomicAdd(&pay[slot], 1u);
            else if (OP == 2) atomicAdd(&pay[slot], vv[i]);
            else if (OP == 3) atomicOr(&pay[slot], vv[i]);
        }
    }
    __syncthreads();
    if (scan_tmp[14]) {  // more distinct keys than the table takes: nothing has been written, the caller takes over
        if (tid == 0) A.dcount[b] = 0xFFFFFFFFu;
        return;
    }
    BBK_PH(4, 1, t_prev);  // loads + insert
    // compaction of the occupied slots: thread t owns slots t, t + 512, ... (consecutive lanes read
    // consecutive 8-byte slots: no LDS bank conflicts; the output order is free, the set is unsorted)
    constexpr int SPT = kHashSlots / kHashThreads;
    uint32_t cnt = 0;
    if constexpr (OP == 0 && kHashDirectOut) {
        cnt = (uint32_t)__popc(firsts);  // no payload to fetch: whoever put a key into the table writes it out
    } else {
#pragma unroll
        for (int j = 0; j < SPT; ++j) cnt += tab[j * kHashThreads + tid] != EMPTY ? 1u : 0u;
    }
    uint32_t incl = cnt;
    incl = wave_scan_incl(incl);
    if (lane == 63) scan_tmp[wave] = incl;
    __syncthreads();
    uint32_t wbase, total;
    wave_totals<kHashThreads / 64>(scan_tmp, lane, wave, wbase, total);
    Key<1> *obuf = buf;  // back to the head of the bucket (every record has been read before the barrier above)
    uint32_t *ovals = vals;
    const uint32_t obase = start;
    uint32_t o = obase + wbase + incl - cnt;
    if constexpr (OP == 0 && kHashDirectOut) {
#pragma unroll
        for (int i = 0; i < kHashItems; ++i) {
            if (firsts & (1u << i)) obuf[o++].w[0] = kk[i];
        }
    } else {
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            const unsigned long long key = tab[j * kHashThreads + tid];
            if (key != EMPTY) {
                obuf[o].w[0] = key;
                if (OP != 0) ovals[o] = pay[j * kHashThreads + tid];
                ++o;
            }
        }
    }
    BBK_PH(4, 2, t_prev);  // compaction + output
#ifdef BBK_PHASE_PROF
    if (threadIdx.x == 0) atomicAdd(&g_phase[4][7], 1ull);
#endif
    if (tid == 0) A.dcount[b] = total;
}

// Same idea for wider keys: the bucket's keys are staged in LDS and the table holds record INDICES
// (32-bit ds_cmpst); a probe that finds a different index compares the two keys.  All keys are in
// LDS before the first insertion, so there is no partially written slot to race with.
constexpr int kHashIdxThreads = 512;
// 16-byte keys: 512 x 8 = 4096 records per bucket, 8192 slots; 24/32-byte keys: 512 x 4 = 2048 records, 4096 slots
// (keys + table + payload table must fit the 160 KB of LDS)
template <int W>
struct HashIdxCfg {
    static constexpr int ITEMS = (W <= 2) ? 8 : 4;
    static constexpr uint32_t CAP = kHashIdxThreads * ITEMS;
    static constexpr uint32_t SLOTS = 2 * CAP;
};

template <int W, int OP>
__global__ __launch_bounds__(kHashIdxThreads) void k_bucket_hashidx(Key<W> *__restrict__ buf,
                                                                   uint32_t *__restrict__ vals, BucketArgs A) {
    constexpr int kHashIdxItems = HashIdxCfg<W>::ITEMS;
    constexpr uint32_t kHashIdxCap = HashIdxCfg<W>::CAP, kHashIdxSlots = HashIdxCfg<W>::SLOTS;
    constexpr bool IN_VAL = OP >= 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *tab = reinterpret_cast<uint32_t *>(smem);
    uint32_t *pay = tab + kHashIdxSlots;
    uint32_t *scan_tmp = pay + (OP != 0 ? kHashIdxSlots : 0);
    Key<W> *skeys = reinterpret_cast<Key<W> *>(scan_tmp + 32);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t b = A.bucket_ids ? A.bucket_ids[blockIdx.x] : blockIdx.x;
    uint32_t start, n;
    bucket_range(A, b, &start, &n);
    if (n == 0) {
        if (tid == 0) A.dcount[b] = 0;
        return;
    }
    if (n > kHashIdxCap) {
        if (tid == 0) A.dcount[b] = 0xFFFFFFFFu;
        return;
    }
    constexpr uint32_t EMPTY = 0xFFFFFFFFu;
    for (uint32_t s = tid; s < kHashIdxSlots; s += kHashIdxThreads) {
        tab[s] = EMPTY;
        if (OP != 0) pay[s] = 0;
    }
    uint32_t vv[kHashIdxItems];
    {
        // loads first, unconditional (see k_bucket)
        Key<W> rk[kHashIdxItems];
#pragma unroll
        for (int i = 0; i < kHashIdxItems; ++i) {
            const uint32_t p = (uint32_t)(i * kHashIdxThreads + tid);
            const uint32_t at = start + (p < n ? p : n - 1u);
            rk[i] = key_load<W>(&buf[at]);
            vv[i] = IN_VAL ? vals[at] : 0u;
        }
#pragma unroll
        for (int i = 0; i < kHashIdxItems; ++i) {
            const uint32_t p = (uint32_t)(i * kHashIdxThreads + tid);
            if (p < n) key_store<W>(&skeys[p], rk[i]);
            else vv[i] = 0;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kHashIdxItems; ++i) {
        const uint32_t p = (uint32_t)(i * kHashIdxThreads + tid);
        if (p < n) {
            const Key<W> key = key_load<W>(&skeys[p]);
            // bits of the hash other than the ones the partition consumed (its top ~20)
            uint32_t slot = (part_hash32<W>(key) * 0x9E3779B1u >> 7) & (kHashIdxSlots - 1);
            for (;;) {
                const uint32_t old = atomicCAS(&tab[slot], EMPTY, p);
                if (old == EMPTY) break;
                if (key_eq<W>(key_load<W>(&skeys[old]), key)) break;
                slot = (slot + 1) & (kHashIdxSlots - 1);
            }
            if (OP == 1) atomicAdd(&pay[slot], 1u);
            else if (OP == 2) atomicAdd(&pay[slot], vv[i]);
            else if (OP == 3) atomicOr(&pay[slot], vv[i]);
        }
    }
    __syncthreads();
    constexpr int SPT = kHashIdxSlots / kHashIdxThreads;
    uint32_t cnt = 0;
#pragma unroll
    for (int j = 0; j < SPT; ++j) cnt += tab[j * kHashIdxThreads + tid] != EMPTY ? 1u : 0u;  // no bank conflicts
    uint32_t incl = cnt;
    incl = wave_scan_incl(incl);
    if (lane == 63) scan_tmp[wave] = incl;
    __syncthreads();
    uint32_t wbase, total;
    wave_totals<kHashIdxThreads / 64>(scan_tmp, lane, wave, wbase, total);
    Key<W> *obuf = buf;  // back to the head of the bucket (every record has been read before the barrier above)
    uint32_t *ovals = vals;
    const uint32_t obase = start;
    uint32_t o = obase + wbase + incl - cnt;
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
        const uint32_t idx = tab[j * kHashIdxThreads + tid];
        if (idx != EMPTY) {
            key_store<W>(&obuf[o], key_load<W>(&skeys[idx]));
            if (OP != 0) ovals[o] = pay[j * kHashIdxThreads + tid];
            ++o;
        }
    }
    if (tid == 0) A.dcount[b] = total;
}

template <int W, int OP>
static size_t bucket_hashidx_smem() {
    return 4 * HashIdxCfg<W>::SLOTS + (OP != 0 ? 4 * HashIdxCfg<W>::SLOTS : 0) + 128 + (size_t)W * 8 * HashIdxCfg<W>::CAP;
}

template <int OP>
static size_t bucket_hash_smem() {
    return sizeof(unsigned long long) * kHashSlots + (OP != 0 ? 4 * kHashSlots : 0) + 64;
}

// one wave per bucket: dense output
template <int W, bool HAS_VAL>
__global__ __launch_bounds__(256) void k_compact(const Key<W> *__restrict__ buf, const uint32_t *__restrict__ vals,
                                                const uint32_t *__restrict__ boff, const uint32_t *__restrict__ dcount,
                                                const uint64_t *__restrict__ doff, uint32_t nbuckets,
                                                Key<W> *__restrict__ out, uint32_t *__restrict__ vout,
                                                uint64_t mask0,  // cleared from word 0 (sort tag), else ~0
                                                uint32_t slot_cap) {  // boff == null: bucket b starts at b*slot_cap
    const uint32_t b = (uint32_t)((BBK_GID()) >> 6);
    if (b >= nbuckets) return;
    const int lane = threadIdx.x & 63;
    uint32_t c = dcount[b];
    if (c == 0xFFFFFFFFu) c = 0;  // left to the caller (reprocessed with the spill list)
    const uint32_t s = boff ? boff[b] : b * slot_cap;
    const uint64_t d = doff[b];
    for (uint32_t i = lane; i < c; i += 64) {
        Key<W> key = key_load<W>(&buf[s + i]);
        key.w[0] &= mask0;
        key_store<W>(&out[d + i], key);
        if (HAS_VAL) vout[d + i] = vals[s + i];
    }
}

// ids of the buckets the first-pass kernel left alone (dcount == 0xFFFFFFFF); *count may run past cap
__global__ void k_flagged(const uint32_t *__restrict__ dcount, uint32_t n, uint32_t *__restrict__ ids, uint32_t cap,
                          uint32_t *__restrict__ count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && dcount[i] == 0xFFFFFFFFu) {
        const uint32_t at = atomicAdd(count, 1u);
        if (at < cap) ids[at] = i;
    }
}

// records every bucket slot holds (cursor - slot start, at most the slot's capacity)
__global__ void k_slot_counts(const uint32_t *__restrict__ cursor, uint32_t n, uint32_t stride, uint32_t cap,
                              uint64_t *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const uint32_t c = cursor[i] - i * stride;
        out[i] = c < cap ? c : cap;
    }
}

__global__ void k_iota_mul(uint32_t *__restrict__ out, uint32_t n, uint32_t mul) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = i * mul;
}

__global__ void k_u32_to_u64(const uint32_t *__restrict__ in, uint64_t n, uint64_t *__restrict__ out, uint32_t clampv) {
    const uint64_t i = BBK_GID();
    if (i < n) out[i] = in[i] == 0xFFFFFFFFu ? (uint64_t)clampv : (uint64_t)in[i];
}

__global__ void k_scan_to_u32(const uint64_t *__restrict__ in, uint64_t n, uint64_t total, uint32_t *__restrict__ out) {
    const uint64_t i = BBK_GID();
    if (i < n) out[i] = (uint32_t)in[i];
    if (i == n) out[n] = (uint32_t)total;
}

// k-mers and chunks (of ch k-mer positions) of every read; *unordered is set when the packed words of the reads do not
// lie one after the other in read order (then no tile stages its window of words in LDS: k_tile_reads)
__global__ void k_kmers_per_read2(const uint32_t *__restrict__ len, const uint64_t *__restrict__ woff, uint64_t n,
                                  uint32_t k, uint32_t ch, uint64_t *__restrict__ nk, uint64_t *__restrict__ nch,
                                  uint32_t *__restrict__ unordered) {
    const uint64_t i = BBK_GID();
    if (i < n) {
        const uint32_t L = len[i];
        const uint64_t c = L >= k ? (uint64_t)(L - k + 1) : 0ull;
        nk[i] = c;
        nch[i] = (c + ch - 1) / ch;
        if (i + 1 < n && woff[i + 1] < woff[i] + ((L + 31u) >> 5)) *unordered = 1u;
    }
}

// ------------------------------------------------------------------------------------------
// narrow stage A kernels (4-byte records, see "narrow records" above): level 1 from reads, level 2, dedup
// ------------------------------------------------------------------------------------------
constexpr int kNwThreads = 1024;
#ifndef BBK_NW_ROUNDS  // (experiments: -DBBK_NW_ROUNDS=4 -DBBK_NW_CHUNKS=3840 -DBBK_NW_WAVES=4 is one workgroup per CU with a 120 KB stage)
#define BBK_NW_ROUNDS 2
#define BBK_NW_CHUNKS 1920
#define BBK_NW_WAVES 8
#endif
constexpr int kNwRounds = BBK_NW_ROUNDS;  // consecutive chunks of 8 k-mer positions per lane (the last 64 lanes of a full tile idle)
constexpr int kNwChunks = BBK_NW_CHUNKS;  // chunks of a level-1 tile: 15360 records = 60 KB staged, runs of ~15 per bin;
                                  // (x 8 records) with the tables 77 KB of LDS: two workgroups per CU
// with a payload (one mask byte per record: the extension index) the same tile would take 94 KB = ONE workgroup per CU
// (measured 8.4 ms against 3.8 ms without payload); 1536 chunks = 12 288 records x 5 bytes + tables = 78 KB
template <bool HAS_VAL>
struct NwCfg {
    static constexpr int ROUNDS = HAS_VAL ? 1 : kNwRounds;       // with the mask extraction two rounds need 72 VGPRs:
    static constexpr int CHUNKS = HAS_VAL ? 1024 : kNwChunks;    // one workgroup of 1024 per CU.  One round: 8192 records
    static constexpr int TILE = CHUNKS * 8;
};

// Eight consecutive k-mer positions of one read from ONE 64-bit window (narrow k: 8 + k + 1 <= 30 bases fit).  With
// F = bases p .. p+31 (base p in the low bits) and NR = ~rev2(F) (the complement of base p at the top),
//   a_i = F  << (64 - 2k - 2i)   is k-mer i top-aligned (its last base in the top bits, other bases of the read below),
//   b_i = NR << 2i               is its reverse complement laid out the same way,
// and the canonical k-mer (base-lexicographic minimum of the two, rtseq.hpp:407-415) is min(a_i, b_i) >> (64 - 2k):
// comparing a k-mer x with rc(x) from the last base down decides like comparing them from the first base up (the first
// difference from the start, x[j] against ~x[k-1-j], is also the first one from the end, ~x[j] against x[k-1-j]).
// Two shifts, one compare, two selects per k-mer -- no carried state, against ~20 operations of the rolled form.
template <bool HAS_VAL>
__device__ __forceinline__ void nw_chunk(const uint64_t *rw, uint32_t p, uint32_t cnt, uint32_t len, uint32_t k_, int hb,
                                         uint32_t *lhist, uint32_t (&lo)[8], uint32_t (&bins)[3], uint32_t (&masks)[2]) {
    constexpr int CH = 8;
    const uint32_t pad = 64u - 2u * k_;  // 22 .. 30
    uint64_t F = 0, NR = 0;
    uint32_t pb = 0;
    if (cnt) {
        F = bases_from(rw, p, (len - 1u) >> 5);
        NR = ~rev2(F);
        if (HAS_VAL) pb = p ? base_at(rw, p - 1u) : 0u;
    }
    bins[0] = bins[1] = bins[2] = 0;
    masks[0] = masks[1] = 0;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        // (computed for idle positions too -- values, not branches: only the LDS atomic is conditional)
        const uint64_t a = F << (pad - 2u * (uint32_t)i);
        uint64_t b = NR << (2 * i);
        // a palindrome counts as minimal (only the mask bits can tell): the unused low bits of b are set
        if (HAS_VAL) b |= (1ull << pad) - 1ull;
        const bool minimal = a <= b;
        const uint64_t key = (minimal ? a : b) >> pad;
        const uint32_t klo = (uint32_t)key;
        const uint32_t bin = nw_bin1((uint32_t)(key >> 32), nw_mix(klo), hb);  // key < 4^k: bin < 1024
        if (HAS_VAL) {
            const uint32_t q = p + (uint32_t)i;
            const uint32_t nextc = (uint32_t)(F >> (2u * ((uint32_t)i + k_))) & 3u;  // base q + k (i + k <= 28)
            const uint32_t prevc = i == 0 ? pb : (uint32_t)(F >> (2 * (i > 0 ? i - 1 : 0))) & 3u;  // base q - 1
            uint32_t m = 0;
            if (q + k_ < len) m |= 1u << (minimal ? nextc : 7u - nextc);
            if (q >= 1) m |= 1u << (minimal ? 4u + prevc : 3u - prevc);
            masks[i >> 2] |= m << (8 * (i & 3));
        }
        if ((uint32_t)i < cnt) atomicAdd(&lhist[bin], 1u);  // count only: the place inside the bin is taken after the scan
        lo[i] = klo;
        bins[i / 3] |= bin << (10 * (i % 3));  // three 10-bit bins per register
        // with the mask arithmetic eight interleaved positions need more registers than two workgroups per CU leave
        if (HAS_VAL) __builtin_amdgcn_sched_barrier(0);
    }
}

// read (relative to the tile's first read) that owns chunk c of the tile: rel[r] <= c < rel[r + 1].  Reads are about
// equally long: the interpolated guess is right or off by one nearly always; otherwise a binary search.
__device__ __forceinline__ uint32_t nw_read_of(const int32_t *s_rel, uint32_t nr, int32_t c, int32_t rel0, float scale) {
    uint32_t g = (uint32_t)((float)(c - rel0) * scale);
    g = g < nr ? g : nr - 1u;
    if (s_rel[g] > c) --g;               // s_rel[0] <= 0 <= c: g stays >= 0
    else if (s_rel[g + 1] <= c) ++g;     // s_rel[nr] > c for every chunk of the tile: g stays < nr
    if (s_rel[g] > c || s_rel[g + 1] <= c) g = read_of(s_rel, nr, c);
    return g;
}

// A lane extracts ROUNDS consecutive chunks (usually of one read: the owner of the first is looked up, the next ones
// follow from it).
template <bool FAST, bool HAS_VAL>
__device__ __forceinline__ void nw_extract(const ReadSrc &S, const PartLevel &L, uint32_t k_, uint32_t tid, uint32_t nch,
                                           uint64_t c0, uint32_t r0, uint32_t nr, const int32_t *s_rel,
                                           const int32_t *s_wrel, const uint32_t *s_len, const uint64_t *s_words,
                                           uint32_t *lhist, uint32_t (&lo)[8 * NwCfg<HAS_VAL>::ROUNDS],
                                           uint32_t (&bins)[3 * NwCfg<HAS_VAL>::ROUNDS],
                                           uint32_t (&masks)[2 * NwCfg<HAS_VAL>::ROUNDS], uint32_t &cnts) {
    constexpr int CH = 8, R = NwCfg<HAS_VAL>::ROUNDS;
    cnts = 0;  // records of round r in bits 4r .. 4r+3
    const int hb = L.narrow_hb;
    const uint32_t ci0 = tid * (uint32_t)R;
    uint32_t ri = 0, p = 0, len = 0, nk = 0;
    const uint64_t *rw = FAST ? s_words : S.words;
    auto rel = [&](uint32_t i) -> int64_t {  // first chunk of read r0 + i, relative to the tile
        return FAST ? (int64_t)s_rel[i] : (int64_t)(S.coff[(uint64_t)r0 + i] - c0);
    };
    auto enter = [&](uint32_t i) {  // per-read values
        if (FAST) {
            len = s_len[i];
            rw = s_words + s_wrel[i];
        } else {
            len = S.len[(uint64_t)r0 + i];
            rw = S.words + S.woff[(uint64_t)r0 + i];
        }
        nk = len - k_ + 1u;
    };
    if (ci0 < nch) {
        if (FAST) {
            const int32_t rel0 = s_rel[0];
            const float scale = (float)nr / (float)(s_rel[nr] - rel0);
            ri = nw_read_of(s_rel, nr, (int32_t)ci0, rel0, scale);
        } else {
            uint32_t a = 0, b = nr;  // largest i with rel(i) <= ci0
            while (b - a > 1) {
                const uint32_t mid = (a + b) >> 1;
                if (rel(mid) <= (int64_t)ci0) a = mid;
                else b = mid;
            }
            ri = a;
        }
        enter(ri);
        p = (uint32_t)((int64_t)ci0 - rel(ri)) * CH;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t ci = ci0 + (uint32_t)r;
        uint32_t cnt = 0;
        if (ci < nch) {
            if (r > 0) {
                p += CH;
                if (p >= nk) {  // the next read that has chunks (rel(nr) lies beyond the tile: the walk ends)
                    do ++ri;
                    while (rel(ri + 1u) <= (int64_t)ci);
                    enter(ri);
                    p = 0;
                }
            }
            cnt = nk - p < (uint32_t)CH ? nk - p : (uint32_t)CH;
        }
        uint32_t l8[CH], b3[3], m2[2];
        nw_chunk<HAS_VAL>(rw, p, cnt, len, k_, hb, lhist, l8, b3, m2);
#pragma unroll
        for (int i = 0; i < CH; ++i) lo[r * CH + i] = l8[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) bins[r * 3 + i] = b3[i];
        masks[r * 2] = m2[0];
        masks[r * 2 + 1] = m2[1];
        cnts |= cnt << (4 * r);
    }
}

// Level 1: fused extraction + partition into 1024 segments.  The bin of a staged record cannot be recomputed from lo
// alone, and a per-record side array would cost as much LDS as the stage itself: the staged order is bin-major, so
// one bit per position marks where a non-empty bin starts and the r-th non-empty bin owns position pos when
// r = #marks at or before pos - 1 (a 64-position word of marks is exactly what a wave handles per step).  What a
// store needs of its bin -- global offset and room left in the slot -- sits in one 8-byte entry indexed by r.
template <bool HAS_VAL>
__global__ __launch_bounds__(kNwThreads) __attribute__((amdgpu_waves_per_eu(BBK_NW_WAVES, BBK_NW_WAVES))) void k_part_reads_narrow(ReadSrc S, PartLevel L, uint32_t ntiles,
                                                                 const RdTile *__restrict__ tiles_arg,  // = S.tiles: as an
                                                                 // argument of its own the descriptor is a scalar load
                                                                 uint32_t *__restrict__ cursor,
                                                                 uint32_t *__restrict__ out, uint32_t *__restrict__ vout) {
    constexpr int NT = kNwThreads, CH = 8, MAXB = kNwBins1, ITEMS = CH * NwCfg<HAS_VAL>::ROUNDS;
    constexpr int MW = NwCfg<HAS_VAL>::TILE / 64;  // 64-bit mark words
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *lhist = reinterpret_cast<uint32_t *>(smem);                   // counts; later, with the next array:
    uint2 *tab = reinterpret_cast<uint2 *>(smem);                           // r -> (global offset - staged start, limit)
    uint32_t *lstart = lhist + 2 * MAXB;
    uint32_t *scan_tmp = lstart + MAXB;                                     // 64 entries
    unsigned long long *mark = reinterpret_cast<unsigned long long *>(scan_tmp + 64);  // MW words
    uint16_t *mbase = reinterpret_cast<uint16_t *>(mark + MW);              // marks before every word
    uint16_t *nz = mbase + MW;                                              // r-th non-empty bin
    unsigned char *U = reinterpret_cast<unsigned char *>(nz + MAXB);
    int32_t *s_rel = reinterpret_cast<int32_t *>(U);
    int32_t *s_wrel = s_rel + (kRdSlots + 2);
    uint32_t *s_len = reinterpret_cast<uint32_t *>(s_wrel + (kRdSlots + 2));
    uint64_t *s_words = reinterpret_cast<uint64_t *>(s_len + (kRdSlots + 2));
    uint32_t *stage = reinterpret_cast<uint32_t *>(U);
    uint8_t *vstage = reinterpret_cast<uint8_t *>(stage + NwCfg<HAS_VAL>::TILE);  // payloads of this path are 8 mask bits

    uint32_t tid = threadIdx.x;
    const uint32_t k_ = (uint32_t)S.k;
    const int hb = L.narrow_hb;
    uint32_t xcc = 0;  // the XCD this workgroup runs on (placement is for speed only: any value gives a correct result)
    if (L.xcd_shift) {
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= (1u << L.xcd_shift) - 1u;
    }
#ifdef BBK_PHASE_PROF
    unsigned long long t_prev = clock64();
#else
    const unsigned long long t_prev = 0;
    (void)t_prev;
#endif

    // A workgroup walks tiles blockIdx.x, + gridDim.x, ... (normally one: grid = tiles) and loads what the NEXT tile
    // needs (its read tables and packed words: two dependent round trips to memory after the descriptor) into registers
    // while it stores the current one.  One table entry and two words per thread: a tile with more reads or words than
    // that takes the global-memory path (as does one not laid out in read order).
    struct Pre {
        uint64_t coff, woff, w0, w1;
        uint32_t len;
    };
    auto staged_ok = [&](const RdTile &T) { return T.wspan != 0xFFFFFFFFu && T.nr < (uint32_t)NT && T.wspan <= 2u * NT; };
    auto prefetch = [&](const RdTile &T, Pre &Q) {
        Q = Pre{0, 0, 0, 0, 0};  // (the previous tile's values end here: they must not stay alive through the loop)
        if (!staged_ok(T)) return;  // (uniform)
        // unconditional loads, indices clamped into the tile's tables (a staged tile has >= 1 read and >= 1 word)
        const uint32_t i0 = tid < T.nr ? tid : T.nr, i1 = tid < T.nr ? tid : T.nr - 1u;
        const uint32_t j0 = tid < T.wspan ? tid : T.wspan - 1u, j1 = tid + NT < T.wspan ? tid + NT : T.wspan - 1u;
        Q.coff = S.coff[(uint64_t)T.r0 + i0];
        Q.woff = S.woff[(uint64_t)T.r0 + i1];
        Q.len = S.len[(uint64_t)T.r0 + i1];
        Q.w0 = S.words[T.wbase + j0];
        Q.w1 = S.words[T.wbase + j1];
    };

#ifdef BBK_NW_TILES_VIA_STRUCT  // (A/B: the descriptor through the pointer inside S -- a vector load + readfirstlane)
    const RdTile *tiles = S.tiles;
    (void)tiles_arg;
#else
    const RdTile *__restrict__ tiles = tiles_arg;
#endif
    uint32_t tile = blockIdx.x;
    if (tile >= ntiles) return;
    RdTile T = tiles[tile];
    Pre Q{0, 0, 0, 0, 0};
    prefetch(T, Q);
    for (;;) {
        // (the thread index is made opaque per iteration: the compiler otherwise computes every address that depends on
        // it -- 16 stage positions, table slots ... -- once before the loop and keeps ~45 registers alive through it,
        // which is one workgroup per CU instead of two)
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = tid >> 6;
        const uint32_t tile_next = tile + gridDim.x;
        const bool more = tile_next < ntiles;  // uniform: every wave of the workgroup leaves the loop together
        RdTile Tn = T;
        if (more) Tn = tiles[tile_next];

        const uint64_t c0 = (uint64_t)tile * NwCfg<HAS_VAL>::CHUNKS;
        const uint64_t left = S.n_chunks - c0;
        const uint32_t nch = left < (uint64_t)NwCfg<HAS_VAL>::CHUNKS ? (uint32_t)left : (uint32_t)NwCfg<HAS_VAL>::CHUNKS;
        const uint32_t r0 = T.r0, nr = T.nr;
        const uint64_t wbase = T.wbase;
        const bool fast = staged_ok(T);
        lhist[tid] = 0;  // NT == MAXB
        if (tid < 2 * MW) reinterpret_cast<uint32_t *>(mark)[tid] = 0u;
        if (fast) {  // (a staged tile's reads lie inside its window of words: k_tile_reads)
            if (tid <= nr) s_rel[tid] = (int32_t)(int64_t)(Q.coff - c0);
            if (tid < nr) {
                s_wrel[tid] = (int32_t)(int64_t)(Q.woff - wbase);
                s_len[tid] = Q.len;
            }
            if (tid < T.wspan) s_words[tid] = Q.w0;
            if (tid + NT < T.wspan) s_words[tid + NT] = Q.w1;
        }
        __syncthreads();
        BBK_PH(5, 0, t_prev);  // read tables + words into LDS

        uint32_t lo[ITEMS], bins[3 * NwCfg<HAS_VAL>::ROUNDS], masks[2 * NwCfg<HAS_VAL>::ROUNDS], cnts;
        // (two instantiations: the address space of the packed words -- LDS or global -- must be static, a pointer that
        // may be either compiles to flat loads)
        if (fast) nw_extract<true, HAS_VAL>(S, L, k_, tid, nch, c0, r0, nr, s_rel, s_wrel, s_len, s_words, lhist, lo, bins, masks, cnts);
        else nw_extract<false, HAS_VAL>(S, L, k_, tid, nch, c0, r0, nr, s_rel, s_wrel, s_len, s_words, lhist, lo, bins, masks, cnts);
        __syncthreads();  // histogram complete; the read tables may be overwritten by the stage
        BBK_PH(5, 1, t_prev);  // extraction + LDS ranking

        // scan of the 1024 bin counts (one bin per thread) and of the non-empty flags; reservation of the tile's run
        const uint32_t c = lhist[tid];
        uint32_t incl = c;
        incl = wave_scan_incl(incl);
        const unsigned long long nzb = __ballot(c != 0);
        if (lane == 63) scan_tmp[wave] = incl | ((uint32_t)__popcll(nzb) << 16);  // records < 2^16, non-empty bins <= 1024
        __syncthreads();
        uint32_t before, total;
        wave_totals<NT / 64>(scan_tmp, lane, wave, before, total);
        const uint32_t staged = total & 0xFFFFu;
        const uint32_t ex = (before & 0xFFFFu) + incl - c;
        const uint32_t myr = (before >> 16) + (uint32_t)__popcll(nzb & ((1ull << lane) - 1ull));
        lstart[tid] = ex;
        uint32_t greserve = 0;
        if (c) {
            greserve = atomicAdd(&cursor[(tid << L.xcd_shift) + xcc], c);
            nz[myr] = (uint16_t)tid;
            atomicOr(&mark[ex >> 6], 1ull << (ex & 63u));
        }
        __syncthreads();  // (every thread has read its count: lhist may become the table)
        BBK_PH(5, 2, t_prev);  // scans + reservation issue + marks
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int r = i / CH, j = i % CH;
            if ((uint32_t)j < ((cnts >> (4 * r)) & 15u)) {
                const uint32_t bin = (bins[r * 3 + j / 3] >> (10 * (j % 3))) & 1023u;
                const uint32_t pos = atomicAdd(&lstart[bin], 1u);  // (lstart ends as the bins' end offsets; nothing reads it again)
                stage[pos] = lo[i];
                if (HAS_VAL) vstage[pos] = (uint8_t)(masks[r * 2 + (j >> 2)] >> (8 * (j & 3)));
            }
        }
        asm volatile("" : "+v"(greserve));  // awaited by every lane here, not inside the store loop's conditional blocks
        if (c) {
            // first staged position of this bin that no longer fits its slot
            const uint64_t slot_end = L.xcd_shift ? (uint64_t)tid * L.slot_stride + (uint64_t)(xcc + 1u) * L.sub_cap
                                                  : (uint64_t)tid * L.slot_stride + L.slot_cap;
            const int64_t room = (int64_t)slot_end - (int64_t)greserve;
            tab[myr] = make_uint2(greserve - ex,
                                  (uint32_t)(int32_t)(room < -(int64_t)0x7FFF0000 ? -(int64_t)0x7FFF0000 : room) + ex);
        }
        if (wave == 0) {  // marks before every 64-position word: lane l owns words WPL*l .. WPL*l + WPL-1
            constexpr int WPL = (MW + 63) / 64;
            uint32_t pw[WPL], tot = 0;
#pragma unroll
            for (int j = 0; j < WPL; ++j) {
                const int idx = lane * WPL + j;
                pw[j] = tot;
                tot += idx < MW ? (uint32_t)__popcll(mark[idx]) : 0u;
            }
            uint32_t inc2 = tot;
            inc2 = wave_scan_incl(inc2);
            const uint32_t lb = inc2 - tot;
#pragma unroll
            for (int j = 0; j < WPL; ++j) {
                const int idx = lane * WPL + j;
                if (idx < MW) mbase[idx] = (uint16_t)(lb + pw[j]);
            }
        }
        __syncthreads();
        BBK_PH(5, 3, t_prev);  // reorder into LDS + mark prefix
        if (more) prefetch(Tn, Q);  // in flight during the stores below
        else Q = Pre{0, 0, 0, 0, 0};   // (the old values end here either way: they must not stay alive through the loop)
        const unsigned long long upto = (2ull << lane) - 1ull;  // this lane and the ones below
        // pos = i * NT + tid: the 64 lanes of a wave cover mark word i * (NT / 64) + wave
        const unsigned long long *wmark = mark + wave;
        const uint16_t *wmbase = mbase + wave;
        uint32_t full = 0;  // items whose slot is full (rare; handled after the stores so that no atomic sits between them)
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint32_t pos = (uint32_t)i * NT + tid;
            if (pos < staged) {
                const uint32_t r = (uint32_t)wmbase[i * (NT / 64)] + (uint32_t)__popcll(wmark[i * (NT / 64)] & upto) - 1u;
                unsigned long long e = reinterpret_cast<const unsigned long long *>(tab)[r];
                const uint32_t rec = stage[pos];
                asm volatile("" : "+v"(e));  // one 8-byte LDS read (otherwise: the limit, a branch, then the offset)
                if ((int32_t)pos >= (int32_t)(uint32_t)(e >> 32)) {
                    full |= 1u << i;
                } else {
                    const uint32_t g = (uint32_t)e + pos;
                    out[g] = rec;
                    if (HAS_VAL) vout[g] = vstage[pos];
                }
            }
        }
        if (full) {
#pragma unroll 1
            for (int i = 0; i < ITEMS; ++i) {
                if ((full >> i) & 1u) {
                    const uint32_t pos = (uint32_t)i * NT + tid;
                    const uint32_t r = (uint32_t)wmbase[i * (NT / 64)] + (uint32_t)__popcll(wmark[i * (NT / 64)] & upto) - 1u;
                    const uint32_t sp = atomicAdd(L.spill_count, 1u);
                    if (sp < L.spill_cap) {
                        reinterpret_cast<uint64_t *>(L.spill_keys)[sp] = nw_key(nz[r], stage[pos], hb);
                        if (HAS_VAL) L.spill_vals[sp] = vstage[pos];
                    }
                }
            }
        }
        BBK_PH(5, 4, t_prev);  // store issue
#ifdef BBK_PHASE_PROF
        if (threadIdx.x == 0) atomicAdd(&g_phase[5][7], 1ull);
#endif
        if (!more) break;
        __syncthreads();  // the stage and the tables have been read: the next tile may overwrite them
        tile = tile_next;
        T = Tn;
    }
}

static size_t part_reads_narrow_smem(bool has_val) {
    const size_t tables = sizeof(uint32_t) * 3 * (kRdSlots + 2) + sizeof(uint64_t) * (kRdWords + 1 + 2);
    const size_t tile = has_val ? NwCfg<true>::TILE : NwCfg<false>::TILE;
    const size_t stage = tile * (has_val ? 5 : 4);
    const size_t fixed = sizeof(uint32_t) * (3 * kNwBins1 + 64) + (tile / 64) * (8 + 2) + (size_t)kNwBins1 * 2;
    return fixed + std::max(tables, stage);
}

// Level 2: one tile (<= 16384 records) of one segment -> the bucket slots of that segment.  Everything derives from lo.
constexpr int kNw2Threads = 1024;
// records per lane: 12 without payload (60 VGPRs: two workgroups of 1024 per CU; with 16 the kernel needed 72 and ran
// one: 3.2 -> 2.6 ms at BASELINE configs[1]), 8 with a payload (the mask array costs the registers of four records)
template <bool HAS_VAL>
struct Nw2Cfg {
    static constexpr int ITEMS = HAS_VAL ? 8 : 12;
    static constexpr int TILE = kNw2Threads * ITEMS;
};

template <bool HAS_VAL>
__global__ __launch_bounds__(kNw2Threads) void k_part_narrow2(const uint32_t *__restrict__ in, const uint32_t *__restrict__ vin,
                                                             const uint4 *__restrict__ desc, PartLevel L,
                                                             uint32_t *__restrict__ cursor, uint32_t *__restrict__ out,
                                                             uint32_t *__restrict__ vout) {
    constexpr int NT = kNw2Threads, ITEMS = Nw2Cfg<HAS_VAL>::ITEMS, MAXB = kMaxBins;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *lhist = reinterpret_cast<uint32_t *>(smem);
    uint32_t *lstart = lhist + MAXB;
    uint32_t *goff = lstart + MAXB;
    uint32_t *scan_tmp = goff + MAXB;
    uint32_t *stage = scan_tmp + 32;
    uint32_t *vstage = stage + Nw2Cfg<HAS_VAL>::TILE;
    const uint32_t tid = threadIdx.x;
    const int hb = L.narrow_hb;
    const uint4 d = desc[blockIdx.x];  // first record, records, bins of the segment | segment << 16, flat index of bin 0
    const uint32_t begin = d.x, count = d.y, nb = d.z & 0xFFFFu, seg = d.z >> 16, gbin0 = d.w;
    if (count == 0) return;  // an unused place of the XCD-wise order (k_tile_desc_narrow)
    lhist[tid] = 0;  // NT == MAXB
    __syncthreads();
    uint32_t lo[ITEMS], vals[ITEMS], binrank[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {  // all loads first (index clamped into the tile)
        const uint32_t local = (uint32_t)i * NT + tid;
        const uint32_t at = begin + (local < count ? local : count - 1u);
        lo[i] = in[at];
        vals[i] = HAS_VAL ? vin[at] : 0u;
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        asm volatile("" : "+v"(lo[i]));
        if (HAS_VAL) asm volatile("" : "+v"(vals[i]));
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t local = (uint32_t)i * NT + tid;
        binrank[i] = 0xFFFFFFFFu;
        if (local < count) {
            const uint32_t b = __umulhi(nw_p2(nw_mix(lo[i])), nb);
            const uint32_t rank = atomicAdd(&lhist[b], 1u);
            binrank[i] = (b << 16) | rank;
        }
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t c = tid < nb ? lhist[tid] : 0u;
    uint32_t incl = c;
    incl = wave_scan_incl(incl);
    if (lane == 63) scan_tmp[wave] = incl;
    __syncthreads();
    uint32_t wb, staged;
    wave_totals<NT / 64>(scan_tmp, lane, wave, wb, staged);
    const uint32_t ex = wb + incl - c;
    if (tid < nb) lstart[tid] = ex;
    uint32_t greserve = c ? atomicAdd(&cursor[gbin0 + tid], c) : 0u;
    __syncthreads();
    // the reservation's result is awaited HERE, by every lane: the compiler otherwise puts the wait for it (vmcnt 0) into
    // the conditional blocks of the store loop below, where it makes every store wait for the one before
    asm volatile("" : "+v"(greserve));
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        if (binrank[i] != 0xFFFFFFFFu) {
            const uint32_t pos = lstart[binrank[i] >> 16] + (binrank[i] & 0xFFFFu);
            stage[pos] = lo[i];
            if (HAS_VAL) vstage[pos] = vals[i];
        }
    }
    if (tid < nb) {
        goff[tid] = greserve - ex;
        const int64_t room = (int64_t)((uint64_t)(gbin0 + tid) * L.slot_stride + L.slot_cap) - (int64_t)greserve;
        lhist[tid] = (uint32_t)(int32_t)(room < -(int64_t)0x7FFF0000 ? -(int64_t)0x7FFF0000 : room) + ex;
    }
    __syncthreads();
    // (records whose slot is full are rare and handled after the stores: the spill counter's atomic returns a value, and
    // a wait for it between the stores would make every store wait for the one before)
    uint32_t full = 0;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t pos = (uint32_t)i * NT + tid;
        if (pos < staged) {
            const uint32_t rec = stage[pos];
            const uint32_t b = __umulhi(nw_p2(nw_mix(rec)), nb);
            if ((int32_t)pos >= (int32_t)lhist[b]) {
                full |= 1u << i;
            } else {
                const uint32_t g = goff[b] + pos;
                out[g] = rec;
                if (HAS_VAL) vout[g] = vstage[pos];
            }
        }
    }
    if (full) {
#pragma unroll 1
        for (int i = 0; i < ITEMS; ++i) {
            if ((full >> i) & 1u) {
                const uint32_t pos = (uint32_t)i * NT + tid;
                const uint32_t sp = atomicAdd(L.spill_count, 1u);
                if (sp < L.spill_cap) {
                    reinterpret_cast<uint64_t *>(L.spill_keys)[sp] = nw_key(seg, stage[pos], hb);
                    if (HAS_VAL) L.spill_vals[sp] = vstage[pos];
                }
            }
        }
    }
}

static size_t part_narrow2_smem(bool has_val) {
    return sizeof(uint32_t) * (3 * kMaxBins + 32) + (size_t)(has_val ? Nw2Cfg<true>::TILE : Nw2Cfg<false>::TILE) * 4 * (has_val ? 2 : 1);
}

// level-2 tile descriptors of the narrow path: like k_tile_desc, with the segment id beside the bin count
// xstart (optional): the tiles of level-1 segment s are dealt to the workgroups that run on XCD s % 8 (workgroup b runs on
// XCD b % 8): tile i of M-entry e becomes workgroup 8 * (xstart[e] + i) + s % 8.  All tiles that fill the buckets of one
// segment then write through ONE L2 (lines filled from several XCDs are what makes a scatter slow, see the level-1 call
// site), and few segments are in flight per XCD at a time.  Unused places keep a zero descriptor (no records).
__global__ void k_tile_desc_narrow(TileMap M, const uint32_t *__restrict__ seg_nb2, const uint32_t *__restrict__ seg_bin_start,
                                   uint32_t tile_size, int sub_shift, const uint32_t *__restrict__ xstart,
                                   uint4 *__restrict__ desc) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M.ntiles) return;
    uint32_t lo = 0, hi = M.nseg;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (M.seg_tile_start[mid] <= t) lo = mid;
        else hi = mid;
    }
    const uint32_t b = M.seg_off[lo] + (t - M.seg_tile_start[lo]) * tile_size;
    const uint32_t e = M.seg_off[lo] + M.seg_size[lo];
    const uint32_t seg = lo >> sub_shift;  // M's entries are the per-XCD sub-slots of the level-1 segments
    const uint32_t at = xstart ? 8u * (xstart[lo] + (t - M.seg_tile_start[lo])) + (seg & 7u) : t;
    desc[at] = make_uint4(b, (e - b) < tile_size ? (e - b) : tile_size, seg_nb2[seg] | (seg << 16), seg_bin_start[seg]);
}

// Dedup of one bucket of 4-byte records in an LDS table (32-bit ds_cmpst); the distinct records leave as 8-byte keys
// rebuilt from (segment of the bucket, lo).  The all-ones record (16 x T) is the table's empty marker and is counted
// on the side.
constexpr int kNwHashThreads = 512;
constexpr int kNwHashItems = 16;  // 8192 records per bucket
constexpr uint32_t kNwHashSlots = 8192;

template <int OP>
__global__ __launch_bounds__(kNwHashThreads) void k_bucket_hash32(uint32_t *__restrict__ buf,
                                                                 uint32_t *__restrict__ vals, BucketArgs A,
                                                                 const uint16_t *__restrict__ bucket_seg, int hb) {
    constexpr bool IN_VAL = OP >= 2;
    constexpr uint32_t EMPTY = 0xFFFFFFFFu;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *tab = reinterpret_cast<uint32_t *>(smem);
    uint32_t *pay = tab + kNwHashSlots;
    uint32_t *scan_tmp = pay + (OP != 0 ? kNwHashSlots : 0);  // [0..7] wave totals, [12] payload of the all-ones record,
                                                              // [13] its presence, [14] give-up flag, [15] output base
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t b = blockIdx.x;
#ifdef BBK_PHASE_PROF
    unsigned long long t_prev = clock64();
#else
    const unsigned long long t_prev = 0;
    (void)t_prev;
#endif
    uint32_t start, n;
    bucket_range(A, b, &start, &n);
    if (n == 0) {
        if (tid == 0) A.dcount[b] = 0;
        return;
    }
    if (n > (uint32_t)(kNwHashThreads * kNwHashItems)) {
        if (tid == 0) A.dcount[b] = 0xFFFFFFFFu;
        return;
    }
    uint32_t kk[kNwHashItems], vv[kNwHashItems];
#pragma unroll
    for (int i = 0; i < kNwHashItems; ++i) {  // (the loads are in flight while the table is cleared)
        const uint32_t p = (uint32_t)(i * kNwHashThreads + tid);
        const uint32_t at = start + (p < n ? p : n - 1u);
        kk[i] = buf[at];
        vv[i] = IN_VAL ? vals[at] : 0u;
    }
    for (uint32_t s = tid; s < kNwHashSlots; s += kNwHashThreads) {
        tab[s] = EMPTY;
        if (OP != 0) pay[s] = 0;
    }
    if (tid < 4) scan_tmp[12 + tid] = 0;
    __syncthreads();
    BBK_PH(4, 0, t_prev);  // table cleared
#ifdef BBK_PHASE_PROF
#pragma unroll
    for (int i = 0; i < kNwHashItems; ++i) asm volatile("" : "+v"(kk[i]));
    BBK_PH(4, 1, t_prev);  // records loaded
#endif
    uint32_t firsts = 0;  // bit i: record i of this lane was the first of its key in the table
#pragma unroll
    for (int i = 0; i < kNwHashItems; ++i) {
        const uint32_t p = (uint32_t)(i * kNwHashThreads + tid);
        if (p < n) {
            if (kk[i] == EMPTY) {
                scan_tmp[13] = 1;
                if (OP == 1) atomicAdd(&scan_tmp[12], 1u);
                else if (OP == 2) atomicAdd(&scan_tmp[12], vv[i]);
                else if (OP == 3) atomicOr(&scan_tmp[12], vv[i]);
                continue;
            }
            uint32_t slot = nw_slot(nw_mix(kk[i])) & (kNwHashSlots - 1);
            uint32_t probes = 0;
            for (;;) {
                const uint32_t old = atomicCAS(&tab[slot], EMPTY, kk[i]);
                if (old == EMPTY) firsts |= 1u << i;
                if (old == EMPTY || old == kk[i]) break;
                slot = (slot + 1) & (kNwHashSlots - 1);
                if (++probes > A.max_probes) {
                    scan_tmp[14] = 1;
                    break;
                }
            }
            if (OP == 1) atomicAdd(&pay[slot], 1u);
            else if (OP == 2) atomicAdd(&pay[slot], vv[i]);
            else if (OP == 3) atomicOr(&pay[slot], vv[i]);
        }
    }
    __syncthreads();
    BBK_PH(4, 2, t_prev);  // inserted
    if (scan_tmp[14]) {  // the table is (nearly) full: nothing has been written, the caller takes over
        if (tid == 0) A.dcount[b] = 0xFFFFFFFFu;
        return;
    }
    constexpr int SPT = kNwHashSlots / kNwHashThreads;
    uint32_t cnt = 0;
    if constexpr (OP == 0 && kHashDirectOut) {
        cnt = (uint32_t)__popc(firsts);  // no payload to fetch: whoever put a key into the table writes it out -- no walk
    } else {                             // over the 8192 slots
#pragma unroll
        for (int j = 0; j < SPT; ++j) cnt += tab[j * kNwHashThreads + tid] != EMPTY ? 1u : 0u;
    }
    uint32_t incl = cnt;
    incl = wave_scan_incl(incl);
    if (lane == 63) scan_tmp[wave] = incl;
    __syncthreads();
    uint32_t wbase, total;
    wave_totals<kNwHashThreads / 64>(scan_tmp, lane, wave, wbase, total);
    const uint32_t extra = scan_tmp[13] ? 1u : 0u;
    // The distinct records (4 bytes, still without their segment) go back to the head of the bucket's own slot; a
    // pass over the bucket counts gives the offsets of the dense result and k_compact_narrow widens them into it.
    // (Until round 3 every bucket reserved its place in the result with an atomicAdd on ONE counter: 227 210 buckets at
    // BASELINE configs[1], served one after the other at ~11 ns each -- 2.6 ms of the kernel's 2.8,
    // tools/probes/single_counter_probe.hip.)  Every record of the bucket has been loaded AND used before the barrier
    // that follows the insertions: nothing is overwritten before it has been read.
    uint32_t o = start + wbase + incl - cnt;
    if constexpr (OP == 0 && kHashDirectOut) {
        // (the loaded records are awaited here by every lane: the insertion loop used them under `p < n` only, and the
        // compiler would otherwise wait for them -- vmcnt 0 -- in front of every store below)
#pragma unroll
        for (int i = 0; i < kNwHashItems; ++i) asm volatile("" : "+v"(kk[i]));
#pragma unroll
        for (int i = 0; i < kNwHashItems; ++i) {
            if (firsts & (1u << i)) buf[o++] = kk[i];
        }
    } else {
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            const uint32_t rec = tab[j * kNwHashThreads + tid];
            if (rec != EMPTY) {
                buf[o] = rec;
                if (OP != 0) vals[o] = pay[j * kNwHashThreads + tid];
                ++o;
            }
        }
    }
    BBK_PH(4, 3, t_prev);  // compaction + output
#ifdef BBK_PHASE_PROF
    if (threadIdx.x == 0) atomicAdd(&g_phase[4][7], 1ull);
#endif
    if (tid == 0) {
        if (extra) {
            buf[start + total] = EMPTY;
            if (OP != 0) vals[start + total] = scan_tmp[12];
        }
        A.dcount[b] = total + extra;
    }
}

// one wave per bucket of the narrow path: the distinct 4-byte records at the head of every bucket slot -> 8-byte keys
// (nw_key: the segment gives the high bits) at their place in the dense result
template <bool HAS_VAL>
__global__ __launch_bounds__(256) void k_compact_narrow(const uint32_t *__restrict__ buf, const uint32_t *__restrict__ vals,
                                                       const uint32_t *__restrict__ dcount, const uint64_t *__restrict__ doff,
                                                       uint32_t nbuckets, uint32_t slot_stride,
                                                       const uint16_t *__restrict__ bucket_seg, int hb,
                                                       uint64_t *__restrict__ out, uint32_t *__restrict__ vout) {
    const uint32_t b = (uint32_t)((BBK_GID()) >> 6);
    if (b >= nbuckets) return;
    const int lane = threadIdx.x & 63;
    uint32_t c = dcount[b];
    if (c == 0xFFFFFFFFu) c = 0;  // left to the caller (reprocessed with the spill list)
    const uint32_t s = b * slot_stride, seg = bucket_seg[b];
    const uint64_t d = doff[b];
    for (uint32_t i = lane; i < c; i += 64) {
        out[d + i] = nw_key(seg, buf[s + i], hb);
        if (HAS_VAL) vout[d + i] = vals[s + i];
    }
}

template <int OP>
static size_t bucket_hash32_smem() {
    return sizeof(uint32_t) * kNwHashSlots * (OP != 0 ? 2 : 1) + sizeof(uint32_t) * 16;
}

// 4-byte records of one segment -> 8-byte keys (overflowing slots are reprocessed by the exact path on a key array)
__global__ void k_nw_widen(const uint32_t *__restrict__ in, uint32_t n, uint32_t seg, int hb, uint64_t *__restrict__ out) {
    const uint32_t i = (uint32_t)BBK_GID();  // cnt is a 32-bit count
    if (i < n) out[i] = nw_key(seg, in[i], hb);
}

// ------------------------------------------------------------------------------------------
// host orchestration
// ------------------------------------------------------------------------------------------
// ODD on purpose: in the blocked phases thread t reads records t*ITEMS + i, i.e. lanes are ITEMS*W*2
// dwords apart; with an even ITEMS that stride is a multiple of 16 dwords and a wave hits 2-4 LDS banks
// (16- to 32-way conflicts); with an odd ITEMS the ds_read_b64/b128 of a lane group are conflict-free.
// First pass (k_bucket_dist): 512 threads x 11 records of 8 bytes (CAP 5632), x 7 of 16 bytes (CAP 3584) -- two
// workgroups per CU and few records per lane (the kernel is issue-bound: 256 x 23 was 20 % slower, 512 x 11
// records of 16 bytes, one workgroup per CU, 75 % slower).  Second chance (k_bucket, radix): 512 x 23 / 512 x 11.
template <int W>
struct BktCfg {
#ifndef BBK_BKT_NT
#define BBK_BKT_NT 512
#define BBK_BKT_ITEMS 11
#endif
#ifndef BBK_BKT2_NT
#define BBK_BKT2_NT 512
#define BBK_BKT2_ITEMS 7
#endif
    static constexpr int NT = (W == 1) ? BBK_BKT_NT : BBK_BKT2_NT;
    static constexpr int ITEMS = (W == 1) ? BBK_BKT_ITEMS : (W == 2 ? BBK_BKT2_ITEMS : (W == 3 ? 5 : 3));
    static constexpr uint32_t CAP = NT * ITEMS;
    static constexpr int NT2 = 512;                               // second-chance kernel
    static constexpr int ITEMS2 = (W == 1) ? 23 : (W == 2 ? 11 : (W == 3 ? 7 : 5));
    static constexpr uint32_t CAP2 = NT2 * ITEMS2;
};
// mean bucket = 0.70 CAP: a bucket holds ~100 distinct genomic k-mers x their multiplicity (~40 at 50x
// coverage), so its size varies far more than Poisson on the record count would suggest
constexpr double kBucketFill = 0.70;

template <int W>
struct MsdRunner {
    bbk_ctx *ctx;
    unsigned k;
    int dmode;
    int op;        // MSD_OP_*
    bool in_vals;  // records carry a payload from the start (mask extraction or input counts)
    uint64_t strip_mask = ~0ull;  // tagged sort: bits of word 0 that survive in the output
    bool slots_ok = getenv("BBK_NO_SLOTS") == nullptr;  // histogram-free slot mode allowed (HASH prefix)
    bool kslots_ok = getenv("BBK_NO_KSLOTS") == nullptr;  // ... and for the ordering pass of a distinct key array
    bool never_decline = false;  // finish whatever overflows bucket by bucket on the LSD path instead of declining
    bool even_part = false;      // the call sorts a materialised range of an expanded input (run_level0): key slots apply
    bool assume_distinct = false;  // caller's hint (key arrays, KEYS / REF prefix): duplicates are not expected
    unsigned expand_k = 0;         // key-array input holds CANONICAL k-mers of this length: both strands are generated
    bool expand_tag = false;       // ... with the XXH3 bucket tag above the k-mer (the runner's k is then k + 2)

    template <bool HAS_VAL, bool HIST>
    void launch_part(const char *fam, double bytes, uint32_t ntiles, const Key<W> *in, const uint32_t *vin, TileMap M,
                     PartLevel L, uint32_t *ghist, uint32_t *cursor, Key<W> *out, uint32_t *vout) {
        if (L.level == 1) launch_part_l<HAS_VAL, HIST, true>(fam, bytes, ntiles, in, vin, M, L, ghist, cursor, out, vout);
        else launch_part_l<HAS_VAL, HIST, false>(fam, bytes, ntiles, in, vin, M, L, ghist, cursor, out, vout);
    }

    template <bool HAS_VAL, bool HIST, bool LVL1>
    void launch_part_l(const char *fam, double bytes, uint32_t ntiles, const Key<W> *in, const uint32_t *vin, TileMap M,
                       PartLevel L, uint32_t *ghist, uint32_t *cursor, Key<W> *out, uint32_t *vout) {
        if (ntiles == 0) return;
        const size_t sm = part_smem(W, PartCfg<W>::TILE, HAS_VAL, HIST, LVL1);
        auto fn = k_part<W, HAS_VAL, HIST, LVL1>;
        BBK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)sm));
        M.ntiles = ntiles;
        M.group = HIST ? 16u : 1u;
        const uint32_t grid = (ntiles + M.group - 1) / M.group;
        KernelTimer t(ctx, fam, bytes);
        hipLaunchKernelGGL(fn, dim3(grid), dim3(PartCfg<W>::THREADS), sm, ctx->stream, in, vin, M, L, ghist, cursor, out,
                           vout);
        check_launch(fam);
    }

    template <bool HAS_VAL, bool HIST>
    void launch_part_reads(const char *fam, double bytes, uint32_t ntiles, ReadSrc S, PartLevel L, uint32_t *ghist,
                           uint32_t *cursor, Key<W> *out, uint32_t *vout) {
        if (ntiles == 0) return;
        const size_t sm = part_reads_smem(W, HAS_VAL, HIST);
        auto fn = k_part_reads<W, HAS_VAL, HIST>;
        BBK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)sm));
        // histogram: ~8 resident-workgroup rounds, each workgroup walks its tiles and flushes once
        const uint32_t grid = HIST ? std::min<uint32_t>(ntiles, 8192u) : ntiles;
        KernelTimer t(ctx, fam, bytes);
        hipLaunchKernelGGL(fn, dim3(grid), dim3(HIST ? kRdHistThreads : kRdThreads), sm, ctx->stream, S, L, ghist, cursor,
                           out, vout);
        check_launch(fam);
    }

    // first pass: the one-pass distribution sort; second chance (SECOND): ballot-ranked radix passes, which
    // take any key distribution and twice the records
    template <bool SECOND, int OP>
    void launch_bucket(uint32_t nblocks, Key<W> *buf, uint32_t *vals, BucketArgs A, double bytes) {
        if (nblocks == 0) return;
        constexpr int NT = SECOND ? BktCfg<W>::NT2 : BktCfg<W>::NT;
        constexpr int IT = SECOND ? BktCfg<W>::ITEMS2 : BktCfg<W>::ITEMS;
        static const bool no_dist = getenv("BBK_NO_DIST") != nullptr;  // A/B switch
        const bool dist = !SECOND && !no_dist;
        const size_t sm = dist ? bucket_dist_smem<W, NT, IT, OP>() : bucket_smem<W, NT, IT, OP>();
        auto fn = dist ? k_bucket_dist<W, NT, IT, OP> : k_bucket<W, NT, IT, OP>;
        BBK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)sm));
        KernelTimer t(ctx, dist ? "k_bucket_dist" : "k_bucket", bytes);
        hipLaunchKernelGGL(fn, dim3(nblocks), dim3(NT), sm, ctx->stream, buf, vals, A);
        check_launch("k_bucket");
    }

    template <int OP>
    void launch_bucket_hash(uint32_t nblocks, Key<W> *buf, uint32_t *vals, BucketArgs A, double bytes) {
        if constexpr (W == 1) {
            if (nblocks == 0) return;
            const size_t sm = bucket_hash_smem<OP>();
            auto fn = k_bucket_hash<OP>;
            BBK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)sm));
            KernelTimer t(ctx, "k_bucket_hash", bytes);
            hipLaunchKernelGGL(fn, dim3(nblocks), dim3(kHashThreads), sm, ctx->stream, buf, vals, A);
            check_launch("k_bucket_hash");
        }
    }

    static uint32_t hash_max_probes() {
        static const char *e = getenv("BBK_HASH_MAX_PROBES");  // tests: force the table give-up on half-empty slots
        return e ? (uint32_t)strtoul(e, nullptr, 10) : kHashMaxProbes;
    }
    // unsorted dedup is enough when a later stage sorts the distinct records (HASH mode)
    bool use_hash_dedup() const { return W == 1 && dmode == MSD_HASH && 2 * k < 64; }
    bool use_hashidx_dedup() const { return W >= 2 && dmode == MSD_HASH; }
    // records a first-pass bucket kernel can hold
    uint32_t bucket_cap() const {
        if (use_hashidx_dedup()) return HashIdxCfg<W>::CAP;
        if (use_hash_dedup()) return (uint32_t)(kHashThreads * kHashItems);
        return BktCfg<W>::CAP;
    }

    template <int OP>
    void launch_bucket_hashidx(uint32_t nblocks, Key<W> *buf, uint32_t *vals, BucketArgs A, double bytes) {
        if constexpr (W >= 2) {
            if (nblocks == 0) return;
            const size_t sm = bucket_hashidx_smem<W, OP>();
            auto fn = k_bucket_hashidx<W, OP>;
            BBK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)sm));
            KernelTimer t(ctx, "k_bucket_hashidx", bytes);
            hipLaunchKernelGGL(fn, dim3(nblocks), dim3(kHashIdxThreads), sm, ctx->stream, buf, vals, A);
            check_launch("k_bucket_hashidx");
        }
    }

    template <bool SECOND>
    void bucket_dispatch(uint32_t nblocks, Key<W> *buf, uint32_t *vals, BucketArgs A, double bytes,
                         bool allow_hash = true) {
        if (allow_hash && use_hashidx_dedup()) {
            switch (op) {
                case MSD_OP_NONE: launch_bucket_hashidx<0>(nblocks, buf, vals, A, bytes); return;
                case MSD_OP_COUNT: launch_bucket_hashidx<1>(nblocks, buf, vals, A, bytes); return;
                case MSD_OP_SUM: launch_bucket_hashidx<2>(nblocks, buf, vals, A, bytes); return;
                case MSD_OP_OR: launch_bucket_hashidx<3>(nblocks, buf, vals, A, bytes); return;
                default: BBK_REQUIRE(false, BBK_ERR_ARG, "bad reduce op");
            }
        }
        if (allow_hash && use_hash_dedup()) {
            switch (op) {
                case MSD_OP_NONE: launch_bucket_hash<0>(nblocks, buf, vals, A, bytes); return;
                case MSD_OP_COUNT: launch_bucket_hash<1>(nblocks, buf, vals, A, bytes); return;
                case MSD_OP_SUM: launch_bucket_hash<2>(nblocks, buf, vals, A, bytes); return;
                case MSD_OP_OR: launch_bucket_hash<3>(nblocks, buf, vals, A, bytes); return;
                default: BBK_REQUIRE(false, BBK_ERR_ARG, "bad reduce op");
            }
        }
        switch (op) {
            case MSD_OP_NONE: launch_bucket<SECOND, 0>(nblocks, buf, vals, A, bytes); break;
            case MSD_OP_COUNT: launch_bucket<SECOND, 1>(nblocks, buf, vals, A, bytes); break;
            case MSD_OP_SUM: launch_bucket<SECOND, 2>(nblocks, buf, vals, A, bytes); break;
            case MSD_OP_OR: launch_bucket<SECOND, 3>(nblocks, buf, vals, A, bytes); break;
            default: BBK_REQUIRE(false, BBK_ERR_ARG, "bad reduce op");
        }
    }

    // returns false if the caller should use the LSD path instead (too much overflow)
    // records two partition levels can take in one pass (bins <= 512 x ~768 of 0.7 CAP records)
    // mean bucket fill the bin plan aims at.  Key arrays deduplicated through the LDS hash table (merge of received
    // shards / pushed batches: multiplicity 1..few) are nearly all distinct: keep the table's load around 0.5 there
    double plan_fill(bool from_reads) const { return (!from_reads && use_hash_dedup()) ? 0.52 : kBucketFill; }
    uint64_t pass_limit(bool from_reads) const {
        static const char *e = getenv("BBK_PASS_LIMIT");  // tests force range passes on small inputs
        if (e) return strtoull(e, nullptr, 10);
        return (uint64_t)(plan_fill(from_reads) * 512 * 0.75 * kMaxBins * bucket_cap() * 0.98);
    }

    // One range of a range pass: prefixes in [lo, lo + span) (span == 0: everything), est = planning estimate of
    // the records it holds (exact for KEYS / REF ranges, which come from a histogram).
    struct Sel {
        uint32_t lo = 0, span = 0;
        int shl = 0;
        uint32_t mul = 0;
        uint64_t est = 0;
        // shl and mul from span: (d << shl) + mulhi(d << shl, mul) maps [0, span) monotonically onto [0, 2^32)
        void finish() {
            shl = __builtin_clz(span - 1u);
            const uint64_t S = (uint64_t)span << shl;  // in (2^31, 2^32]
            mul = S >= (1ull << 32) ? 0u : (uint32_t)((((1ull << 32) - S) << 32) / S);
        }
    };
    // Where the dense result of an exact-mode pass goes when the caller has already allocated it (range passes
    // write one after the other into the final array: no concatenation copy of a 100 GB result).
    struct Dst {
        void *keys = nullptr;
        uint32_t *vals = nullptr;
    };

    // Returns 1 = done, 0 = declined (caller uses the LSD path), 2 = the input holds more records than one
    // pass takes (*too_big set): run_all splits it into ranges of the prefix space.
    int run(const bbk_reads *rd, const void *d_keys, const uint32_t *d_vals, uint64_t n_in, bool with_mask,
            MsdOutput &out, Sel sel = Sel(), bool *too_big = nullptr, Dst dst = Dst()) {
        constexpr uint32_t kPartTileK = PartCfg<W>::TILE;
        const bool from_reads = rd != nullptr;
        const bool has_val = with_mask || d_vals != nullptr;
        const size_t rec = (size_t)W * 8;
        const int w0bits = (W == 1) ? (int)(2 * k) : 64;

        // ---- instance space (reads: k-mers for the sizes, chunks for the level-1 tiles)
        DevBuf coff, tile_read, unordered;
        uint64_t N = (expand_k && rd == nullptr) ? 2 * n_in : n_in, n_chunks = 0;
        if (from_reads) {
            BBK_REQUIRE(dmode == MSD_HASH, BBK_ERR_INTERNAL, "reads are partitioned by hash prefix only");
            DevBuf nk((rd->n + 1) * sizeof(uint64_t));
            coff.alloc((rd->n + 1) * sizeof(uint64_t));
            unordered.alloc(16);
            BBK_HIP(hipMemsetAsync(unordered.p, 0, 16, ctx->stream));
            if (rd->n) {
                hipLaunchKernelGGL(k_kmers_per_read2, bbk::grid_blocks((rd->n + 255) / 256), dim3(256), 0, ctx->stream,
                                   rd->d_len, rd->d_woff, rd->n, k, (uint32_t)RdCfg<W>::CH, nk.as<uint64_t>(),
                                   coff.as<uint64_t>(), unordered.as<uint32_t>());
                check_launch("k_kmers_per_read2");
            }
            N = exclusive_scan_u64(ctx, nk.as<uint64_t>(), nk.as<uint64_t>(), rd->n);
            n_chunks = exclusive_scan_u64(ctx, coff.as<uint64_t>(), coff.as<uint64_t>(), rd->n);
            BBK_HIP(hipMemcpyAsync(coff.as<uint64_t>() + rd->n, &n_chunks, sizeof(uint64_t), hipMemcpyHostToDevice,
                                   ctx->stream));
            BBK_HIP(hipStreamSynchronize(ctx->stream));  // n_chunks is a stack variable
        }
        out.instances = N;
        out.n = 0;
        const bool has_dst = dst.keys != nullptr;
        auto empty_out = [&]() {
            if (!has_dst) {
                out.keys.alloc(16);
                out.vals.alloc(16);
            }
        };
        if (N == 0) {
            empty_out();
            return 1;
        }
        const bool ranged = sel.span != 0;
        if (!ranged && N > pass_limit(from_reads) && too_big) {
            *too_big = true;
            return 2;
        }
        const uint64_t Ntot = N;        // instance space of the level-1 tiles
        if (ranged) N = sel.est;        // planning estimate of this range's share (exact for KEYS / REF ranges)
        // record offsets inside one pass are 32-bit; the input of a call may hold more (range passes walk the
        // 64-bit instance space once per range)
        BBK_REQUIRE(N < (1ull << 32) - kPartTileK, BBK_ERR_ARG,
                    "batch holds %llu records%s; one pass is limited to 2^32-1 (split the input)",
                    (unsigned long long)N, ranged ? " in one range of the prefix space" : "");

        // ---- bin plan: nb1 (power of two) level-1 bins; level-2 bin counts are chosen per segment below
        const double fill = plan_fill(from_reads);
        const double target = fill * bucket_cap();
        const double want = std::max(1.0, std::ceil((double)N / target));
        uint32_t nb1 = 1;
        int b1 = 0;
        while (nb1 < 512 && (double)nb1 * nb1 < want) {
            nb1 <<= 1;
            ++b1;
        }
        // REF prefix: the 4 XXH3 bucket bits must be consumed before the buckets (a bucket is sorted by key alone
        // and may not hold records of two XXH3 buckets) -- by the range selection (a range inside one XXH3 bucket
        // shifts them all out: shl >= 4; run_all only makes wider ranges as aligned groups of 2^j whole buckets,
        // shl = 4 - j) and, for what is left, by level 1
        const int ref_bits = dmode == MSD_REF ? std::max(0, 4 - (ranged ? sel.shl : 0)) : 0;
        if (b1 < ref_bits) {
            b1 = ref_bits;
            nb1 = 1u << b1;
        }
        // narrow stage A: reads, 8-byte keys of 33..42 bits, slot mode, one pass -> 4-byte records between the levels
        // (1024 level-1 segments whatever the size: the segment carries the key bits the record drops)
        const char *smin_nw = getenv("BBK_SLOTS_MIN");
        const uint64_t slots_min_nw = smin_nw ? strtoull(smin_nw, nullptr, 10) : (1ull << 22);
        const int nw_hb = (int)(2 * k) - 32;
        static const bool no_narrow = getenv("BBK_NO_NARROW") != nullptr;  // A/B switch
        const bool narrow = W == 1 && from_reads && !ranged && !has_dst && nw_hb >= 1 && nw_hb <= 10 && slots_ok &&
                            dmode == MSD_HASH && use_hash_dedup() && N >= slots_min_nw && !no_narrow &&
                            (double)N / fill * 1.1 + (double)N < 4.2e9 && (double)N / kNwBins1 / target < 0.75 * kMaxBins;
        if (narrow) {
            b1 = 10;
            nb1 = kNwBins1;
        }
        const bool verbose = getenv("BBK_VERBOSE") != nullptr;
        if (want / nb1 > 0.75 * kMaxBins) {  // would need a third level: leave to the LSD path
            if (verbose) fprintf(stderr, "[bbk] msd declines: N=%llu needs more than two levels\n", (unsigned long long)N);
            return 0;
        }
        PartLevel L1{1, b1, nb1, dmode, w0bits, nullptr, nullptr, sel.lo, sel.span, sel.shl, sel.mul};

        // level-1 tiles cover the whole instance space; a range pass keeps its share of every tile.  Reads:
        // a tile is `threads` chunks, so the histogram (512 threads) and the scatter (1024) have their own tables
        const uint32_t rd_tile = narrow ? (uint32_t)(has_val ? NwCfg<true>::CHUNKS : NwCfg<false>::CHUNKS)
                                        : (uint32_t)kRdThreads;  // chunks of a level-1 tile
        const uint32_t ntiles1 = from_reads ? (uint32_t)((n_chunks + rd_tile - 1) / rd_tile)
                                            : (uint32_t)((Ntot + kPartTileK - 1) / kPartTileK);
        const uint32_t ntiles1h = (uint32_t)((n_chunks + kRdHistThreads - 1) / kRdHistThreads);
        DevBuf tiles_h;
        ReadSrc S{}, Sh{};
        if (from_reads) {
            tile_read.alloc(((size_t)ntiles1 + 1) * sizeof(RdTile));
            tiles_h.alloc(((size_t)ntiles1h + 1) * sizeof(RdTile));
            if (ntiles1) {
                hipLaunchKernelGGL(k_tile_reads, dim3((ntiles1 + 255) / 256), dim3(256), 0, ctx->stream, coff.as<uint64_t>(),
                                   rd->d_woff, rd->d_len, rd->n, (uint64_t)ntiles1, rd_tile,
                                   (uint32_t)RdCfg<W>::CH, k, (uint32_t)kRdSlots, (uint32_t)kRdWords,
                                   unordered.as<uint32_t>(), tile_read.as<RdTile>());
                hipLaunchKernelGGL(k_tile_reads, dim3((ntiles1h + 255) / 256), dim3(256), 0, ctx->stream,
                                   coff.as<uint64_t>(), rd->d_woff, rd->d_len, rd->n, (uint64_t)ntiles1h,
                                   (uint32_t)kRdHistThreads, (uint32_t)RdCfg<W>::CH, k, (uint32_t)kRdSlots,
                                   (uint32_t)kRdWords, unordered.as<uint32_t>(), tiles_h.as<RdTile>());
                check_launch("k_tile_reads");
            }
            S = ReadSrc{rd->d_words, rd->d_woff, rd->d_len, coff.as<uint64_t>(), tile_read.as<RdTile>(), rd->n, n_chunks,
                        (int)k};
            Sh = S;
            Sh.tiles = tiles_h.as<RdTile>();
        }
        TileMap M1{nullptr, nullptr, nullptr, 1, Ntot, 0, 1, nullptr, (int)expand_k, expand_tag ? 1 : 0};

        // ---- slot mode (HASH prefix + LDS hash dedup): no histogram passes.  The hash spreads the records evenly,
        // so every level-1 segment gets a fixed slot of the mean size + 1 % and every bucket a slot of the dedup
        // kernel's capacity; the scatter kernels reserve space with the same per-(tile, bin) atomics, records
        // that do not fit their slot go to a spill list, and whatever overflowed (spill list + the contents of the
        // overflowing segments / buckets, i.e. every record of the affected keys) is reprocessed by the exact path
        // below on a key array.  Heavy repeats therefore cost a second pass over a small part of the data.
        const char *smin = getenv("BBK_SLOTS_MIN");  // tests lower it to run the slot mode on small inputs
        const uint64_t slots_min = smin ? strtoull(smin, nullptr, 10) : (1ull << 22);
        const bool hslots = slots_ok && !has_dst && dmode == MSD_HASH && (use_hash_dedup() || use_hashidx_dedup()) && nb1 > 1 &&
                            N >= slots_min && (double)N / fill * 1.1 + (double)N < 4.2e9;  // u32 slot offsets
        // Stage B (ordering a distinct key array, KEYS / REF prefix, sorted result written directly): the same slots
        // instead of the two histogram passes.  The prefix is the key itself, so the spread is only as even as the
        // data: ANY record that misses its slot, any bucket the sort kernel turns down, any duplicate sends the call
        // back to the exact path (the slots cannot be patched up in key order the way the hash slots can).
        // (only for EXPANDED input: both strands of a set spread evenly over the key space; a canonical set does not --
        // its last base is A four times as often as T -- and 119 of 512 segments overflowed their slots in every
        // extension-index build: 3.8 ms of a 30 ms build spent on an attempt that never holds)
        // (even_part: a materialised range of an expanded input -- run_level0 -- is as evenly spread, holds exactly
        // sel.est records and writes into its place of the final array)
        const bool kslots = kslots_ok && slots_ok && (even_part || (!has_dst && !ranged && expand_k != 0)) && !from_reads &&
                            assume_distinct && (dmode == MSD_KEYS || dmode == MSD_REF) && nb1 > 1 && N >= slots_min &&
                            getenv("BBK_NO_DIRECT") == nullptr && (double)N / fill * 1.1 + (double)N < 4.2e9;
        const bool slots = hslots || kslots;
        BBK_REQUIRE(!narrow || slots, BBK_ERR_INTERNAL, "narrow records need the slot mode");
        // narrow level 1: one sub-slot (and cursor) per XCD inside every segment slot (PartLevel::xcd_shift); the XCDs do
        // not take exactly equal shares of the tiles, so the sub-slots get 6 % + 2048 records of slack.
        // A 128-byte line of a segment that workgroups on DIFFERENT XCDs fill (their ~60-byte runs are adjacent) is
        // what makes this kernel's store pattern slow: tools/probes/reserve_scatter_probe.hip replays the pattern
        // without any arithmetic -- 3.8 ms with one fill front per segment, 2.1 ms with one per (segment, XCD), 6.4 ms
        // when adjacent runs ALWAYS come from different XCDs.  The kernel itself: 3.64 -> 2.94 ms (same call, round 3;
        // in round 2 its arithmetic took as long as the stores and hid the gain: 3.96 -> 3.79).  BBK_XCD_SLOTS=0: A/B.
        static const bool use_xcd = !(getenv("BBK_XCD_SLOTS") && atoi(getenv("BBK_XCD_SLOTS")) == 0);
        const int xs = (slots && use_xcd && ctx->num_xcds == 8) ? 3 : 0;
        const uint32_t nsub = nb1 << xs;  // level-1 cursors = level-2 input segments
        const uint32_t sub_cap = xs ? ((uint32_t)((double)N / nsub * 1.06) + 2048u) | 1u : 0u;
        const uint32_t seg_cap = !slots ? 0u
                                 : xs ? sub_cap << xs
                                      : ((uint32_t)((double)N / nb1 * (kslots ? 1.06 : 1.01)) + 8192u) | 1u;
        const uint32_t cap2 = narrow ? (uint32_t)(kNwHashThreads * kNwHashItems) : bucket_cap();
        // bucket slots 256 B further apart than their capacity: with a power-of-two-ish stride every bucket's
        // fill front sits in the same HBM channel (level-2 scatter measured 10 % slower)
        const size_t rec_ab = narrow ? 4 : rec;  // record width between the levels
        const uint32_t stride2 = cap2 + (uint32_t)(256 / rec_ab);
        DevBuf spill_k, spill_v, spill_n;  // spill_n: u32 counters [0] spilled records [1] unused [2] buckets left to the caller
        const uint32_t spill_cap = slots ? (uint32_t)(N / 8 + 65536) : 0u;
        if (slots) {
            spill_k.alloc((size_t)spill_cap * rec);
            if (has_val) spill_v.alloc((size_t)spill_cap * 4);
            spill_n.alloc(16);
            BBK_HIP(hipMemsetAsync(spill_n.p, 0, 16, ctx->stream));
            L1.slot_cap = seg_cap;
            L1.slot_stride = seg_cap;
            L1.spill_keys = spill_k.p;
            L1.spill_vals = spill_v.as<uint32_t>();
            L1.spill_count = spill_n.as<uint32_t>();
            L1.spill_cap = spill_cap;
            L1.narrow_hb = narrow ? nw_hb : 0;
            L1.xcd_shift = xs;
            L1.sub_cap = sub_cap;
        }

        // ---- level 1: histogram (exact mode), offsets, scatter
        // (off1 / tstart / hsub are per level-1 CURSOR: per segment, or per (segment, XCD) sub-slot on the narrow path)
        DevBuf hist1((size_t)nb1 * 4 + 16), cur1((size_t)nsub * 4 + 16);
        const Key<W> *kin = (const Key<W> *)d_keys;
        std::vector<uint32_t> h1(nb1), off1(nsub + 1), tstart(nsub + 1), snb2(nb1), sbin(nb1 + 1), hsub(nsub), fill1(nsub);
        std::vector<uint32_t> over_seg;  // slot mode: segments that ran over (reprocessed as a whole)
        if (!slots) {
            BBK_HIP(hipMemsetAsync(hist1.p, 0, (size_t)nb1 * 4 + 16, ctx->stream));
            if (nb1 > 1 || ranged) {
                const double hb = from_reads ? (double)rd->n_words * 8 : (double)N * rec;
                if (from_reads) {
                    launch_part_reads<false, true>("k_part_reads_hist", hb, ntiles1h, Sh, L1, hist1.as<uint32_t>(), nullptr, nullptr, nullptr);
                } else {
                    launch_part<false, true>("k_part_hist1", hb, ntiles1, kin, nullptr, M1, L1, hist1.as<uint32_t>(), nullptr, nullptr, nullptr);
                }
            } else {
                const uint32_t n32 = (uint32_t)N;
                BBK_HIP(hipMemcpyAsync(hist1.p, &n32, 4, hipMemcpyHostToDevice, ctx->stream));
            }
            BBK_HIP(hipMemcpyAsync(h1.data(), hist1.p, (size_t)nb1 * 4, hipMemcpyDeviceToHost, ctx->stream));
            BBK_HIP(hipStreamSynchronize(ctx->stream));
            off1[0] = 0;
            for (uint32_t b = 0; b < nb1; ++b) off1[b + 1] = off1[b] + h1[b];
            if (ranged) {
                N = off1[nb1];  // the records of this range
                out.instances = N;
            }
            BBK_REQUIRE(off1[nb1] == (uint32_t)N, BBK_ERR_INTERNAL, "level-1 histogram does not add up (%u vs %llu)",
                        off1[nb1], (unsigned long long)N);
            if (N == 0) {
                empty_out();
                return 1;
            }
        } else {
            BBK_REQUIRE((uint64_t)nb1 * seg_cap + N < (1ull << 32), BBK_ERR_INTERNAL, "slot layout exceeds 32-bit offsets");
            for (uint32_t s2 = 0; s2 <= nsub; ++s2)
                off1[s2] = xs ? (s2 >> xs) * seg_cap + (s2 & ((1u << xs) - 1u)) * sub_cap : s2 * seg_cap;
        }
        BBK_HIP(hipMemcpyAsync(cur1.p, off1.data(), (size_t)nsub * 4, hipMemcpyHostToDevice, ctx->stream));

        const uint64_t nA = slots ? (uint64_t)nb1 * seg_cap : N;  // records bufA holds (slot layout has gaps)
        DevBuf bufA(nA * rec_ab), bufB, valA, valB;
        const bool need_vbuf = has_val || op != MSD_OP_NONE;
        if (has_val) valA.alloc(nA * 4);
        {
            const double pb = (from_reads ? (double)rd->n_words * 8 : (double)N * (rec + (has_val ? 4 : 0))) +
                              (double)N * (rec + (has_val ? 4 : 0));
            if (narrow) {
                if constexpr (W == 1) {
                    const double pbn = (double)rd->n_words * 8 + (double)N * (4 + (has_val ? 4 : 0));
                    const size_t sm = part_reads_narrow_smem(has_val);
                    // One tile per workgroup.  The kernel can also run as persistent workgroups that walk every grid-th
                    // tile and load the next tile's tables during the stores of the current one
                    // (BBK_NW_WGS_PER_CU=2): measured slower in the same call, 3.38-3.46 ms against 2.94 -- the wait
                    // for the loaded tables at the top of the loop is also a wait for the tile's stores, and the
                    // hardware dispatcher balances the tiles better than a static stride.
                    static const uint32_t per_cu = getenv("BBK_NW_WGS_PER_CU") ? (uint32_t)atoi(getenv("BBK_NW_WGS_PER_CU")) : 0u;
                    const uint32_t grid = per_cu ? std::min<uint32_t>(ntiles1, (uint32_t)ctx->num_cus * per_cu) : ntiles1;
                    KernelTimer t(ctx, "k_part_reads_narrow", pbn);
                    if (has_val) {
                        auto fn = k_part_reads_narrow<true>;
                        BBK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
                        hipLaunchKernelGGL(fn, dim3(grid), dim3(kNwThreads), sm, ctx->stream, S, L1, ntiles1, S.tiles, cur1.as<uint32_t>(), bufA.as<uint32_t>(), valA.as<uint32_t>());
                    } else {
                        auto fn = k_part_reads_narrow<false>;
                        BBK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
                        hipLaunchKernelGGL(fn, dim3(grid), dim3(kNwThreads), sm, ctx->stream, S, L1, ntiles1, S.tiles, cur1.as<uint32_t>(), bufA.as<uint32_t>(), (uint32_t *)nullptr);
                    }
                    check_launch("k_part_reads_narrow");
                }
            } else if (from_reads) {
                if (has_val) launch_part_reads<true, false>("k_part_reads", pb, ntiles1, S, L1, nullptr, cur1.as<uint32_t>(), bufA.as<Key<W>>(), valA.as<uint32_t>());
                else launch_part_reads<false, false>("k_part_reads", pb, ntiles1, S, L1, nullptr, cur1.as<uint32_t>(), bufA.as<Key<W>>(), nullptr);
            } else {
                if (has_val) launch_part<true, false>("k_part_l1", pb, ntiles1, kin, d_vals, M1, L1, nullptr, cur1.as<uint32_t>(), bufA.as<Key<W>>(), valA.as<uint32_t>());
                else launch_part<false, false>("k_part_l1", pb, ntiles1, kin, nullptr, M1, L1, nullptr, cur1.as<uint32_t>(), bufA.as<Key<W>>(), nullptr);
            }
        }
        if (slots) {
            // the cursors tell what every segment received
            std::vector<uint32_t> c1(nsub);
            BBK_HIP(hipMemcpyAsync(c1.data(), cur1.p, (size_t)nsub * 4, hipMemcpyDeviceToHost, ctx->stream));
            BBK_HIP(hipStreamSynchronize(ctx->stream));
            uint64_t got = 0;
            const uint32_t cap1 = xs ? sub_cap : seg_cap;
            for (uint32_t b = 0; b < nb1; ++b) {
                bool over = false;
                uint32_t tot = 0;
                for (uint32_t s2 = b << xs; s2 < (b + 1) << xs; ++s2) {
                    const uint32_t reserved = c1[s2] - off1[s2];
                    got += reserved;
                    over = over || reserved > cap1;
                    fill1[s2] = std::min(reserved, cap1);  // what the slot really holds
                    tot += fill1[s2];
                }
                if (over) over_seg.push_back(b);
                // an overflowing segment is kept out of level 2 as a whole: every record of a key must meet in one bucket
                h1[b] = over ? 0u : tot;
                for (uint32_t s2 = b << xs; s2 < (b + 1) << xs; ++s2) hsub[s2] = over ? 0u : fill1[s2];
            }
            if (got > Ntot) {  // cannot be: every instance reserves one place.  Say what was read before failing
                std::vector<uint32_t> c2(nsub);
                BBK_HIP(hipMemcpyAsync(c2.data(), cur1.p, (size_t)nsub * 4, hipMemcpyDeviceToHost, ctx->stream));
                BBK_HIP(hipStreamSynchronize(ctx->stream));
                uint32_t shown = 0, differ = 0;
                for (uint32_t b = 0; b < nsub; ++b) differ += c1[b] != c2[b];
                for (uint32_t b = 0; b < nsub && shown < 8; ++b)
                    if (c1[b] - off1[b] > 2 * seg_cap) {
                        fprintf(stderr, "[bbk] level-1 cursor %u: start %u now %u (second read %u), slot capacity %u\n", b,
                                off1[b], c1[b], c2[b], seg_cap);
                        ++shown;
                    }
                fprintf(stderr, "[bbk] level-1 cursors: %u of %u differ between two reads; cur1 at %p\n", differ, nsub, cur1.p);
                BBK_REQUIRE(false, BBK_ERR_INTERNAL, "level-1 reservations exceed the instance space (%llu vs %llu)",
                            (unsigned long long)got, (unsigned long long)Ntot);
            }
            if (ranged) {
                N = got;
                out.instances = N;
            }
            BBK_REQUIRE(got == N, BBK_ERR_INTERNAL, "level-1 reservations do not add up (%llu vs %llu)",
                        (unsigned long long)got, (unsigned long long)N);
            if (N == 0) {
                empty_out();
                return 1;
            }
            if (kslots && !over_seg.empty()) {
                if (verbose) fprintf(stderr, "[bbk] msd key slots: %zu segments overflow, exact mode\n", over_seg.size());
                return 4;
            }
        }
        if (!slots)
            for (uint32_t b = 0; b < nb1; ++b) hsub[b] = h1[b];  // exact mode: one dense run per segment
        tstart[0] = 0;
        sbin[0] = 0;
        const uint32_t tile2 = narrow ? (uint32_t)(has_val ? Nw2Cfg<true>::TILE : Nw2Cfg<false>::TILE) : kPartTileK;
        for (uint32_t s2 = 0; s2 < nsub; ++s2) tstart[s2 + 1] = tstart[s2] + (hsub[s2] + tile2 - 1) / tile2;
        for (uint32_t b = 0; b < nb1; ++b) {
            snb2[b] = (uint32_t)std::min<double>(kMaxBins, std::max(1.0, std::ceil((double)h1[b] / target)));
            sbin[b + 1] = sbin[b] + snb2[b];
        }
        const uint32_t nbuckets = sbin[nb1];

        // ---- level 2
        DevBuf seg_tile(((size_t)nsub + 1) * 4), seg_off(((size_t)nsub + 1) * 4), seg_nb2((size_t)nb1 * 4 + 16),
            seg_bin(((size_t)nb1 + 1) * 4), seg_size((size_t)nsub * 4 + 16);
        BBK_HIP(hipMemcpyAsync(seg_tile.p, tstart.data(), ((size_t)nsub + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
        BBK_HIP(hipMemcpyAsync(seg_off.p, off1.data(), ((size_t)nsub + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
        BBK_HIP(hipMemcpyAsync(seg_nb2.p, snb2.data(), (size_t)nb1 * 4, hipMemcpyHostToDevice, ctx->stream));
        BBK_HIP(hipMemcpyAsync(seg_bin.p, sbin.data(), ((size_t)nb1 + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
        BBK_HIP(hipMemcpyAsync(seg_size.p, hsub.data(), (size_t)nsub * 4, hipMemcpyHostToDevice, ctx->stream));
        PartLevel L2{2, b1, nb1, dmode, w0bits, seg_nb2.as<uint32_t>(), seg_bin.as<uint32_t>(), sel.lo, sel.span, sel.shl, sel.mul};
        const uint32_t ntiles2 = tstart[nsub];
        TileMap M2{seg_tile.as<uint32_t>(), seg_off.as<uint32_t>(), seg_size.as<uint32_t>(), nsub, N, ntiles2, 1, nullptr, 0, 0};
        // narrow level 2: workgroups dealt to the XCDs by segment (k_tile_desc_narrow); BBK_XCD_TILES=0: A/B
        static const bool xcd_tiles = !(getenv("BBK_XCD_TILES") && atoi(getenv("BBK_XCD_TILES")) == 0);
        uint32_t nwg2 = ntiles2;  // workgroups of the level-2 kernel
        DevBuf xstart_d;
        std::vector<uint32_t> xstart;  // (lives to the end of the call: the copy below is asynchronous)
        if (xcd_tiles && ntiles2) {  // (exact mode too: its histogram pass keeps the plain order, see desc2h)
            xstart.resize(nsub);
            uint32_t per_xcd[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (uint32_t s2 = 0; s2 < nsub; ++s2) {
                uint32_t &c = per_xcd[(s2 >> xs) & 7u];
                xstart[s2] = c;
                c += tstart[s2 + 1] - tstart[s2];
            }
            nwg2 = 8u * *std::max_element(per_xcd, per_xcd + 8);
            xstart_d.alloc((size_t)nsub * 4);
            BBK_HIP(hipMemcpyAsync(xstart_d.p, xstart.data(), (size_t)nsub * 4, hipMemcpyHostToDevice, ctx->stream));
        }
        DevBuf desc2((size_t)nwg2 * sizeof(uint4) + 16);
        if (ntiles2) {
            if (narrow) {
                if (xstart_d.p) BBK_HIP(hipMemsetAsync(desc2.p, 0, (size_t)nwg2 * sizeof(uint4), ctx->stream));
                hipLaunchKernelGGL(k_tile_desc_narrow, dim3((ntiles2 + 255) / 256), dim3(256), 0, ctx->stream, M2,
                                   seg_nb2.as<uint32_t>(), seg_bin.as<uint32_t>(), tile2, xs,
                                   (const uint32_t *)xstart_d.p, desc2.as<uint4>());
            }
            else {
                if (xstart_d.p) BBK_HIP(hipMemsetAsync(desc2.p, 0, (size_t)nwg2 * sizeof(uint4), ctx->stream));
                hipLaunchKernelGGL(k_tile_desc, dim3((ntiles2 + 255) / 256), dim3(256), 0, ctx->stream, M2, seg_nb2.as<uint32_t>(),
                                   seg_bin.as<uint32_t>(), kPartTileK, xs, (const uint32_t *)xstart_d.p, desc2.as<uint4>());
            }
            check_launch("k_tile_desc");
        }
        M2.desc = desc2.as<uint4>();
        // exact mode: the histogram kernel walks 16 consecutive tiles per workgroup and wants them in plain order
        DevBuf desc2h;
        if (!slots && xstart_d.p && !narrow) {
            desc2h.alloc((size_t)ntiles2 * sizeof(uint4) + 16);
            hipLaunchKernelGGL(k_tile_desc, dim3((ntiles2 + 255) / 256), dim3(256), 0, ctx->stream, M2, seg_nb2.as<uint32_t>(),
                               seg_bin.as<uint32_t>(), kPartTileK, xs, (const uint32_t *)nullptr, desc2h.as<uint4>());
            check_launch("k_tile_desc");
        }
        DevBuf hist2((size_t)nbuckets * 4 + 16), boff(((size_t)nbuckets + 1) * 4 + 16);
        const uint64_t nB = slots ? (uint64_t)nbuckets * stride2 : N;
        if (slots) BBK_REQUIRE(nB + N < (1ull << 32), BBK_ERR_INTERNAL, "slot layout exceeds 32-bit offsets");
        bufB.alloc(nB * rec_ab);
        if (need_vbuf) valB.alloc(nB * 4);
        if (!slots) {
            BBK_HIP(hipMemsetAsync(hist2.p, 0, (size_t)nbuckets * 4 + 16, ctx->stream));
            {
                TileMap M2h = M2;
                if (desc2h.p) M2h.desc = desc2h.as<uint4>();
                launch_part<false, true>("k_part_hist2", (double)N * rec, ntiles2, bufA.as<Key<W>>(), nullptr, M2h, L2,
                                         hist2.as<uint32_t>(), nullptr, nullptr, nullptr);
            }
            DevBuf h64(((size_t)nbuckets + 1) * 8);
            hipLaunchKernelGGL(k_u32_to_u64, dim3((nbuckets + 255) / 256), dim3(256), 0, ctx->stream,
                               hist2.as<uint32_t>(), (uint64_t)nbuckets, h64.as<uint64_t>(), 0u);
            check_launch("k_u32_to_u64");
            const uint64_t tot = exclusive_scan_u64(ctx, h64.as<uint64_t>(), h64.as<uint64_t>(), nbuckets);
            BBK_REQUIRE(tot == N, BBK_ERR_INTERNAL, "level-2 histogram does not add up");
            hipLaunchKernelGGL(k_scan_to_u32, dim3((nbuckets + 1 + 255) / 256), dim3(256), 0, ctx->stream,
                               h64.as<uint64_t>(), (uint64_t)nbuckets, tot, boff.as<uint32_t>());
            check_launch("k_scan_to_u32");
            BBK_HIP(bbk::copy_async(hist2.p, boff.p, (size_t)nbuckets * 4, hipMemcpyDeviceToDevice, ctx->stream));
            BBK_HIP(hipStreamSynchronize(ctx->stream));
        } else {
            // cursor of bucket g starts at its slot
            hipLaunchKernelGGL(k_iota_mul, dim3((nbuckets + 255) / 256), dim3(256), 0, ctx->stream, hist2.as<uint32_t>(),
                               nbuckets, stride2);
            check_launch("k_iota_mul");
            L2.slot_cap = cap2;
            L2.slot_stride = stride2;
            L2.spill_keys = spill_k.p;
            L2.spill_vals = spill_v.as<uint32_t>();
            L2.spill_count = spill_n.as<uint32_t>();
            L2.spill_cap = spill_cap;
            L2.narrow_hb = narrow ? nw_hb : 0;
        }
        if (narrow) {
            if (ntiles2) {
                const double pb = 2.0 * (double)N * (4 + (has_val ? 4 : 0));
                const size_t sm = part_narrow2_smem(has_val);
                KernelTimer t(ctx, "k_part_narrow2", pb);
                if (has_val) {
                    auto fn = k_part_narrow2<true>;
                    BBK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
                    hipLaunchKernelGGL(fn, dim3(nwg2), dim3(kNw2Threads), sm, ctx->stream, bufA.as<uint32_t>(), valA.as<uint32_t>(),
                                       desc2.as<uint4>(), L2, hist2.as<uint32_t>(), bufB.as<uint32_t>(), valB.as<uint32_t>());
                } else {
                    auto fn = k_part_narrow2<false>;
                    BBK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
                    hipLaunchKernelGGL(fn, dim3(nwg2), dim3(kNw2Threads), sm, ctx->stream, bufA.as<uint32_t>(), (const uint32_t *)nullptr,
                                       desc2.as<uint4>(), L2, hist2.as<uint32_t>(), bufB.as<uint32_t>(), (uint32_t *)nullptr);
                }
                check_launch("k_part_narrow2");
            }
        } else {
            const double pb = 2.0 * (double)N * (rec + (has_val ? 4 : 0));
            if (has_val) launch_part<true, false>("k_part_l2", pb, nwg2, bufA.as<Key<W>>(), valA.as<uint32_t>(), M2, L2, nullptr, hist2.as<uint32_t>(), bufB.as<Key<W>>(), valB.as<uint32_t>());
            else launch_part<false, false>("k_part_l2", pb, nwg2, bufA.as<Key<W>>(), nullptr, M2, L2, nullptr, hist2.as<uint32_t>(), bufB.as<Key<W>>(), nullptr);
        }

        // ---- buckets in LDS
        DevBuf dcount((size_t)nbuckets * 4 + 16);
        DevBuf dbg(64);
        BBK_HIP(hipMemsetAsync(dbg.p, 0, 64, ctx->stream));
        // slot mode: the hash-dedup kernels write the distinct records straight into the (unordered) result
        const bool out_vals = op != MSD_OP_NONE;
        // (slot mode: the dedup kernels leave the distinct records at the head of every bucket slot, like the exact mode
        // in its dense buckets; the compaction below makes the result.  BucketArgs::out_keys / out_total -- every bucket
        // reserving its place in the result with an atomicAdd on one counter -- is no longer used: the counter served
        // the 227 210 buckets of BASELINE configs[1] one after the other, 2.6 ms of 2.8.)
        BucketArgs A{boff.as<uint32_t>(), dcount.as<uint32_t>(), nullptr, (int)k, verbose ? dbg.as<uint32_t>() : nullptr,
                     slots ? cap2 : 0u, slots ? stride2 : 0u, hist2.as<uint32_t>(),
                     nullptr, nullptr, nullptr, ~0ull, hash_max_probes(), nullptr};
        const double bb = (double)N * (rec + (has_val ? 4 : 0));
        // Sorted output of a key array that should hold no duplicates (both strands of a distinct canonical set, odd
        // k): the dense result has the offsets of the input, so the sorting kernels write it directly -- no
        // compaction pass.  Should a bucket remove a duplicate after all, or be left to the second chance, the pass
        // is redone in place (the direct pass does not touch the buckets).
        const bool direct = (!slots || kslots) && assume_distinct && (dmode == MSD_KEYS || dmode == MSD_REF) &&
                            getenv("BBK_NO_DIRECT") == nullptr;
        DevBuf dupf, slot_off;
        if (kslots) {
            // dense output offsets = exclusive scan of the slot fills; a total below N means a record missed its slot
            DevBuf c64(((size_t)nbuckets + 1) * 8);
            hipLaunchKernelGGL(k_slot_counts, dim3((nbuckets + 255) / 256), dim3(256), 0, ctx->stream, hist2.as<uint32_t>(),
                               nbuckets, stride2, cap2, c64.as<uint64_t>());
            check_launch("k_slot_counts");
            const uint64_t tot = exclusive_scan_u64(ctx, c64.as<uint64_t>(), c64.as<uint64_t>(), nbuckets);
            if (tot != N) {
                if (verbose) fprintf(stderr, "[bbk] msd key slots: %llu of %llu records placed, exact mode\n",
                                     (unsigned long long)tot, (unsigned long long)N);
                return 4;
            }
            slot_off.alloc(((size_t)nbuckets + 1) * 4 + 16);
            hipLaunchKernelGGL(k_scan_to_u32, dim3((nbuckets + 1 + 255) / 256), dim3(256), 0, ctx->stream, c64.as<uint64_t>(),
                               (uint64_t)nbuckets, tot, slot_off.as<uint32_t>());
            check_launch("k_scan_to_u32");
            BBK_HIP(hipStreamSynchronize(ctx->stream));  // c64 goes out of scope
            A.out_off = slot_off.as<uint32_t>();
        }
        if (direct) {
            if (!has_dst) {
                out.keys.alloc(N * rec + 16);
                if (out_vals) out.vals.alloc(N * 4 + 16);
            }
            dupf.alloc(16);
            BBK_HIP(hipMemsetAsync(dupf.p, 0, 16, ctx->stream));
            A.sorted_keys = has_dst ? dst.keys : out.keys.p;
            A.sorted_vals = has_dst ? dst.vals : out.vals.as<uint32_t>();
            A.dup_flag = dupf.as<uint32_t>();
            A.strip_mask = strip_mask;
        }
        DevBuf bseg;
        std::vector<uint16_t> h_bseg;
        if (narrow) {
            if constexpr (W == 1) {
                h_bseg.resize((size_t)nbuckets + 1);
                for (uint32_t b = 0; b < nb1; ++b)
                    for (uint32_t g = sbin[b]; g < sbin[b + 1]; ++g) h_bseg[g] = (uint16_t)b;
                bseg.alloc(((size_t)nbuckets + 1) * 2);
                BBK_HIP(hipMemcpyAsync(bseg.p, h_bseg.data(), (size_t)nbuckets * 2, hipMemcpyHostToDevice, ctx->stream));
                const double bbn = (double)N * (4 + (has_val ? 4 : 0));
                auto launch32 = [&](auto fn, size_t sm) {
                    BBK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
                    KernelTimer t(ctx, "k_bucket_hash32", bbn);
                    hipLaunchKernelGGL(fn, dim3(nbuckets), dim3(kNwHashThreads), sm, ctx->stream, bufB.as<uint32_t>(),
                                       valB.as<uint32_t>(), A, bseg.as<uint16_t>(), nw_hb);
                    check_launch("k_bucket_hash32");
                };
                if (nbuckets) switch (op) {
                    case MSD_OP_NONE: launch32(k_bucket_hash32<0>, bucket_hash32_smem<0>()); break;
                    case MSD_OP_COUNT: launch32(k_bucket_hash32<1>, bucket_hash32_smem<1>()); break;
                    case MSD_OP_SUM: launch32(k_bucket_hash32<2>, bucket_hash32_smem<2>()); break;
                    case MSD_OP_OR: launch32(k_bucket_hash32<3>, bucket_hash32_smem<3>()); break;
                    default: BBK_REQUIRE(false, BBK_ERR_ARG, "bad reduce op");
                }
            }
        } else {
            bucket_dispatch<false>(nbuckets, bufB.as<Key<W>>(), valB.as<uint32_t>(), A, bb);
        }

        // buckets the first pass left alone: listed on the device, only the (short) list comes to the host
        constexpr uint32_t kFlagCap = 65536;
        DevBuf flag_ids((size_t)kFlagCap * 4), flag_n;
        uint32_t *d_flag_n = nullptr;
        if (slots) {
            d_flag_n = spill_n.as<uint32_t>() + 2;
        } else {
            flag_n.alloc(16);
            d_flag_n = flag_n.as<uint32_t>();
        }
        uint32_t ctr[4] = {0, 0, 0, 0};  // spilled, direct, flagged, duplicates seen by the direct pass
        auto fetch_flags = [&]() {
            if (!slots) BBK_HIP(hipMemsetAsync(flag_n.p, 0, 16, ctx->stream));
            hipLaunchKernelGGL(k_flagged, dim3((nbuckets + 255) / 256), dim3(256), 0, ctx->stream, dcount.as<uint32_t>(),
                               nbuckets, flag_ids.as<uint32_t>(), kFlagCap, d_flag_n);
            check_launch("k_flagged");
            if (slots) BBK_HIP(hipMemcpyAsync(ctr, spill_n.p, 12, hipMemcpyDeviceToHost, ctx->stream));
            else BBK_HIP(hipMemcpyAsync(ctr + 2, flag_n.p, 4, hipMemcpyDeviceToHost, ctx->stream));
            if (direct) BBK_HIP(hipMemcpyAsync(ctr + 3, dupf.p, 4, hipMemcpyDeviceToHost, ctx->stream));
            BBK_HIP(hipStreamSynchronize(ctx->stream));
        };
        fetch_flags();
        if (direct) {
            if (ctr[2] == 0 && ctr[3] == 0 && (!kslots || ctr[0] == 0)) {  // every bucket sorted, nothing removed or
                out.n = N;                                                  // spilled: the result is complete
                out.nbuckets = 0;
                out.overflow_buckets = 0;
                if (kslots && verbose) fprintf(stderr, "[bbk] msd key slots N=%llu buckets=%u: ordered without histograms\n",
                                               (unsigned long long)N, nbuckets);
                return 1;
            }
            if (kslots) {
                if (verbose) fprintf(stderr, "[bbk] msd key slots given up (spill=%u flagged=%u dup=%u), exact mode\n", ctr[0],
                                     ctr[2], ctr[3]);
                out.keys.release();
                out.vals.release();
                return 4;
            }
            if (verbose) fprintf(stderr, "[bbk] msd direct output withdrawn (flagged=%u dup=%u): in-place pass\n", ctr[2], ctr[3]);
            if (!has_dst) {
                out.keys.release();
                out.vals.release();
            }
            A.sorted_keys = nullptr;
            A.sorted_vals = nullptr;
            A.dup_flag = nullptr;
            A.strip_mask = ~0ull;
            bucket_dispatch<false>(nbuckets, bufB.as<Key<W>>(), valB.as<uint32_t>(), A, bb);
            fetch_flags();
        }
        const uint32_t n_spill = ctr[0], n_flag = ctr[2];
        if (n_flag > kFlagCap) {  // tens of thousands of overflowing buckets: not an input for this path
            if (verbose) fprintf(stderr, "[bbk] msd: %u buckets overflow\n", n_flag);
            return slots ? 3 : 0;
        }
        std::vector<uint32_t> flagged(n_flag);
        if (n_flag) {
            BBK_HIP(hipMemcpyAsync(flagged.data(), flag_ids.p, (size_t)n_flag * 4, hipMemcpyDeviceToHost, ctx->stream));
            BBK_HIP(hipStreamSynchronize(ctx->stream));
            std::sort(flagged.begin(), flagged.end());
        }
        std::vector<uint32_t> hd, hb;  // exact mode with flagged buckets (or verbose): per-bucket counts / offsets
        if (!slots && (n_flag || verbose)) {
            hd.resize(nbuckets);
            hb.resize((size_t)nbuckets + 1);
            BBK_HIP(hipMemcpyAsync(hd.data(), dcount.p, (size_t)nbuckets * 4, hipMemcpyDeviceToHost, ctx->stream));
            BBK_HIP(hipMemcpyAsync(hb.data(), boff.p, ((size_t)nbuckets + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
            BBK_HIP(hipStreamSynchronize(ctx->stream));
        }

        MsdOutput extra;  // slot mode: distinct records of everything that overflowed
        uint64_t novf = 0, ovf_rec = 0;
        if (slots) {
            if (n_spill > spill_cap) {  // more than an eighth of the input overflowed: not an input for this mode
                if (verbose) fprintf(stderr, "[bbk] msd slots: spill list overflow (%u), exact mode\n", n_spill);
                return 3;
            }
            const std::vector<uint32_t> &over_bkt = flagged;
            // records a flagged bucket's slot really holds: a bucket is also flagged when its LDS table gives up on a
            // slot that is NOT full (more distinct keys than the table takes), and the rest of such a slot is stale
            // pool memory -- never copy more than the level-2 cursor says was written
            std::vector<uint32_t> bkt_fill(over_bkt.size(), cap2);
            if (!over_bkt.empty()) {
                std::vector<uint32_t> cur2(nbuckets);
                BBK_HIP(hipMemcpyAsync(cur2.data(), hist2.p, (size_t)nbuckets * 4, hipMemcpyDeviceToHost, ctx->stream));
                BBK_HIP(hipStreamSynchronize(ctx->stream));
                for (size_t i = 0; i < over_bkt.size(); ++i) {
                    const uint32_t g = over_bkt[i];
                    bkt_fill[i] = std::min<uint32_t>(cap2, cur2[g] - g * stride2);
                }
            }
            uint64_t n_extra = (uint64_t)n_spill;
            for (uint32_t b : over_seg)
                for (uint32_t s2 = b << xs; s2 < (b + 1) << xs; ++s2) n_extra += fill1[s2];
            for (uint32_t f : bkt_fill) n_extra += f;
            if (verbose)
                fprintf(stderr, "[bbk] msd slots%s N=%llu nb1=%u seg_cap=%u buckets=%u spill=%u over_seg=%zu over_bkt=%zu\n",
                        narrow ? " (narrow records)" : "", (unsigned long long)N, nb1, seg_cap, nbuckets, n_spill,
                        over_seg.size(), over_bkt.size());
            ctx->add_stat("stat_slot_records", (double)N);
            ctx->add_stat("stat_slot_spilled", (double)n_spill);
            ctx->add_stat("stat_slot_overflow_segments", (double)over_seg.size());
            ctx->add_stat("stat_slot_overflow_buckets", (double)over_bkt.size());
            ctx->add_stat("stat_slot_reprocessed", (double)n_extra);
            if (n_extra > N / 2) {  // most of the input overflowed (a handful of distinct k-mers): not for this mode
                if (verbose) fprintf(stderr, "[bbk] msd slots: %llu of %llu records overflowed, exact mode\n",
                                     (unsigned long long)n_extra, (unsigned long long)N);
                return 3;
            }
            if (n_extra) {
                // every record of the affected keys: the spill list, the overflowing segments' and buckets' slots
                const bool tiny = n_extra <= (uint64_t)BktCfg<W>::CAP2;  // fits one workgroup of the radix kernel
                DevBuf ek(n_extra * rec), ev;
                if (has_val || (tiny && op != MSD_OP_NONE)) ev.alloc(n_extra * 4 + 16);
                uint64_t o = 0;
                // narrow path: slots hold 4-byte records, widened with the segment they belong to (seg < 0: 8-byte keys)
                auto put = [&](const void *ksrc, const uint32_t *vsrc, uint64_t first, uint64_t cnt, int seg = -1) {
                    if (!cnt) return;
                    if (seg >= 0) {
                        hipLaunchKernelGGL(k_nw_widen, bbk::grid_blocks((cnt + 255) / 256), dim3(256), 0, ctx->stream,
                                           (const uint32_t *)ksrc + first, (uint32_t)cnt, (uint32_t)seg, nw_hb,
                                           ek.as<uint64_t>() + o);
                        check_launch("k_nw_widen");
                    } else
                    BBK_HIP(bbk::copy_async(ek.as<char>() + o * rec, (const char *)ksrc + first * rec, cnt * rec,
                                           hipMemcpyDeviceToDevice, ctx->stream));
                    if (has_val)
                        BBK_HIP(bbk::copy_async(ev.as<uint32_t>() + o, vsrc + first, cnt * 4, hipMemcpyDeviceToDevice,
                                               ctx->stream));
                    o += cnt;
                };
                put(spill_k.p, spill_v.as<uint32_t>(), 0, n_spill);
                for (uint32_t b : over_seg)  // the written part of its slot (of every per-XCD sub-slot on the narrow path)
                    for (uint32_t s2 = b << xs; s2 < (b + 1) << xs; ++s2)
                        put(bufA.p, valA.as<uint32_t>(), (uint64_t)off1[s2], fill1[s2], narrow ? (int)b : -1);
                for (size_t i = 0; i < over_bkt.size(); ++i)
                    put(bufB.p, valB.as<uint32_t>(), (uint64_t)over_bkt[i] * stride2, bkt_fill[i],
                        narrow ? (int)h_bseg[over_bkt[i]] : -1);
                if (tiny) {
                    // the usual case (one or two crowded buckets): ONE workgroup sorts + reduces all of it in LDS,
                    // instead of a whole partition pipeline for a few thousand records
                    DevBuf tb(16), tc(16);
                    const uint32_t hbo[2] = {0u, (uint32_t)n_extra};
                    BBK_HIP(hipMemcpyAsync(tb.p, hbo, 8, hipMemcpyHostToDevice, ctx->stream));
                    BucketArgs At{tb.as<uint32_t>(), tc.as<uint32_t>(), nullptr, (int)k, nullptr, 0u, 0u,
                                  nullptr, nullptr, nullptr, nullptr, ~0ull, hash_max_probes(), nullptr};
                    MsdRunner<W> sorter = *this;
                    sorter.dmode = MSD_KEYS;  // picks the sorting kernels in bucket_dispatch
                    sorter.expand_k = 0;
                    sorter.template bucket_dispatch<true>(1u, ek.as<Key<W>>(), ev.as<uint32_t>(), At,
                                                          (double)n_extra * (rec + (has_val ? 4 : 0)),
                                                          /*allow_hash=*/false);
                    uint32_t d = 0;
                    BBK_HIP(hipMemcpyAsync(&d, tc.p, 4, hipMemcpyDeviceToHost, ctx->stream));
                    BBK_HIP(hipStreamSynchronize(ctx->stream));
                    BBK_REQUIRE(d != 0xFFFFFFFFu && d <= n_extra, BBK_ERR_INTERNAL, "overflow pass: bad count");
                    extra.n = d;
                    extra.keys = std::move(ek);
                    if (op != MSD_OP_NONE) extra.vals = std::move(ev);
                } else {
                BBK_HIP(hipStreamSynchronize(ctx->stream));
                // The records were selected BY their hash bucket, so the same hash would pile them up again:
                // partition them by key instead (the order of the result does not matter), and never decline --
                // a k-mer with a million instances is finished by the per-bucket LSD fallback.
                MsdRunner<W> exact = *this;
                exact.slots_ok = false;
                exact.dmode = MSD_KEYS;
                exact.never_decline = true;
                exact.expand_k = 0;
                // declined (e.g. one k-mer makes up most of it): so does this call, the caller takes the LSD path
                if (!exact.run_all(nullptr, ek.p, has_val ? ev.as<uint32_t>() : nullptr, n_extra, false, extra)) return 0;
                extra.bucket_off.release();
                }
            }
            novf = over_bkt.size() + over_seg.size();
        }
        bufA.release();
        valA.release();

        if (!slots && (n_flag || verbose)) {
            // ---- buckets above CAP: a second pass with 512-thread workgroups (2 x CAP); what still does not
            // fit (a k-mer repeated > 12 k times in one bucket) is finished by the LSD path, one by one
            std::vector<uint32_t> big;
            uint64_t big_rec = 0;
            // second chance: the ballot-ranked radix kernel with 512 threads -- buckets above the first pass's
            // capacity (8-byte keys: 2 x CAP; wider keys: more than the 4096-record hash kernel) and buckets the
            // distribution sort turned down for a crowded bin
            const uint32_t cap2nd = BktCfg<W>::CAP2;
            for (uint32_t b = 0; b < nbuckets; ++b)
                if (hd[b] == 0xFFFFFFFFu && hb[b + 1] - hb[b] <= cap2nd) {
                    big.push_back(b);
                    big_rec += hb[b + 1] - hb[b];
                }
            if (!big.empty()) {
                DevBuf ids(big.size() * 4);
                BBK_HIP(hipMemcpyAsync(ids.p, big.data(), big.size() * 4, hipMemcpyHostToDevice, ctx->stream));
                BucketArgs A2{boff.as<uint32_t>(), dcount.as<uint32_t>(), ids.as<uint32_t>(), (int)k, nullptr, 0u, 0u,
                              nullptr, nullptr, nullptr, nullptr, ~0ull, hash_max_probes(), nullptr};
                const double b2 = (double)big_rec * (rec + (has_val ? 4 : 0));
                bucket_dispatch<true>((uint32_t)big.size(), bufB.as<Key<W>>(), valB.as<uint32_t>(), A2, b2,
                                      /*allow_hash=*/false);
                BBK_HIP(hipMemcpyAsync(hd.data(), dcount.p, (size_t)nbuckets * 4, hipMemcpyDeviceToHost, ctx->stream));
                BBK_HIP(hipStreamSynchronize(ctx->stream));
            }
            for (uint32_t b = 0; b < nbuckets; ++b)
                if (hd[b] == 0xFFFFFFFFu) {
                    ++novf;
                    ovf_rec += hb[b + 1] - hb[b];
                }
            if (verbose) {
                uint32_t mx = 0;
                for (uint32_t b = 0; b < nbuckets; ++b) mx = std::max(mx, hb[b + 1] - hb[b]);
                uint32_t hdbg[2] = {0, 0};
                BBK_HIP(hipMemcpyAsync(hdbg, dbg.p, 8, hipMemcpyDeviceToHost, ctx->stream));
                BBK_HIP(hipStreamSynchronize(ctx->stream));
                fprintf(stderr, "[bbk] msd all-words-fallback buckets=%u\n", hdbg[0]);
                fprintf(stderr, "[bbk] msd mode=%d N=%llu nb1=%u buckets=%u max_bucket=%u cap=%u big=%zu lsd=%llu (%llu rec)\n",
                        dmode, (unsigned long long)N, nb1, nbuckets, mx, bucket_cap(), big.size(), (unsigned long long)novf,
                        (unsigned long long)ovf_rec);
            }
            if (!never_decline && (novf > 256 || ovf_rec > N / 4)) return 0;
            if (novf) {
                const ReduceOp rop = op == MSD_OP_OR ? REDUCE_OR : (op == MSD_OP_SUM ? REDUCE_SUM : REDUCE_COUNT);
                for (uint32_t b = 0; b < nbuckets; ++b) {
                    if (hd[b] != 0xFFFFFFFFu) continue;
                    const uint64_t cnt = hb[b + 1] - hb[b];
                    Key<W> *kb = bufB.as<Key<W>>() + hb[b];
                    uint32_t *vb = need_vbuf ? valB.as<uint32_t>() + hb[b] : nullptr;
                    DevBuf tk(cnt * rec), tv(cnt * 4), ok(cnt * rec), ov(cnt * 4);
                    sort_records(ctx, W, kb, tk.p, has_val ? vb : nullptr, has_val ? tv.as<uint32_t>() : nullptr, cnt,
                                 key_passes(k));
                    const uint64_t d = unique_records(ctx, W, kb, has_val ? vb : nullptr, cnt, ok.p,
                                                      op != MSD_OP_NONE ? ov.as<uint32_t>() : nullptr, rop, false);
                    BBK_HIP(bbk::copy_async(kb, ok.p, d * rec, hipMemcpyDeviceToDevice, ctx->stream));
                    if (op != MSD_OP_NONE)
                        BBK_HIP(bbk::copy_async(vb, ov.p, d * 4, hipMemcpyDeviceToDevice, ctx->stream));
                    BBK_HIP(hipStreamSynchronize(ctx->stream));
                    hd[b] = (uint32_t)d;
                }
                BBK_HIP(hipMemcpyAsync(dcount.p, hd.data(), (size_t)nbuckets * 4, hipMemcpyHostToDevice, ctx->stream));
            }
        }
        out.overflow_buckets = novf;

        // ---- dense output: scan of the bucket counts + compaction.  (Slot mode: overflowing buckets wrote nothing and
        // count 0 here; their records are in `extra`, appended below.)
        uint64_t D = 0;
        DevBuf d64;
        {
            d64.alloc(((size_t)nbuckets + 1) * 8);
            hipLaunchKernelGGL(k_u32_to_u64, dim3((nbuckets + 255) / 256), dim3(256), 0, ctx->stream,
                               dcount.as<uint32_t>(), (uint64_t)nbuckets, d64.as<uint64_t>(), 0u);
            check_launch("k_u32_to_u64");
            D = exclusive_scan_u64(ctx, d64.as<uint64_t>(), d64.as<uint64_t>(), nbuckets);
            if (slots) BBK_REQUIRE(D + extra.n <= N, BBK_ERR_INTERNAL, "more distinct records than records");
            out.n = D + (slots ? extra.n : 0);
            if (!has_dst) {
                out.keys.alloc(out.n * rec + 16);
                if (out_vals) out.vals.alloc(out.n * 4 + 16);
            }
            Key<W> *ck = has_dst ? (Key<W> *)dst.keys : out.keys.as<Key<W>>();
            uint32_t *cv = has_dst ? dst.vals : out.vals.as<uint32_t>();
            const uint32_t *cboff = slots ? nullptr : boff.as<uint32_t>();  // slot mode: bucket b starts at b * stride2
            const unsigned blocks = (unsigned)(((uint64_t)nbuckets * 64 + 255) / 256);
            if (narrow) {
                if constexpr (W == 1) {
                    KernelTimer t(ctx, "compact", (double)D * (4 + 8 + (out_vals ? 8 : 0)));
                    if (out_vals)
                        hipLaunchKernelGGL(k_compact_narrow<true>, dim3(blocks), dim3(256), 0, ctx->stream, bufB.as<uint32_t>(),
                                           valB.as<uint32_t>(), dcount.as<uint32_t>(), d64.as<uint64_t>(), nbuckets, stride2,
                                           bseg.as<uint16_t>(), nw_hb, reinterpret_cast<uint64_t *>(ck), cv);
                    else
                        hipLaunchKernelGGL(k_compact_narrow<false>, dim3(blocks), dim3(256), 0, ctx->stream, bufB.as<uint32_t>(),
                                           (const uint32_t *)nullptr, dcount.as<uint32_t>(), d64.as<uint64_t>(), nbuckets,
                                           stride2, bseg.as<uint16_t>(), nw_hb, reinterpret_cast<uint64_t *>(ck),
                                           (uint32_t *)nullptr);
                }
            } else {
                KernelTimer t(ctx, "compact", 2.0 * (double)D * (rec + (out_vals ? 4 : 0)));
                if (out_vals)
                    hipLaunchKernelGGL((k_compact<W, true>), dim3(blocks), dim3(256), 0, ctx->stream, bufB.as<Key<W>>(),
                                       valB.as<uint32_t>(), cboff, dcount.as<uint32_t>(), d64.as<uint64_t>(), nbuckets, ck, cv,
                                       strip_mask, stride2);
                else
                    hipLaunchKernelGGL((k_compact<W, false>), dim3(blocks), dim3(256), 0, ctx->stream, bufB.as<Key<W>>(),
                                       (const uint32_t *)nullptr, cboff, dcount.as<uint32_t>(), d64.as<uint64_t>(), nbuckets, ck,
                                       (uint32_t *)nullptr, strip_mask, stride2);
            }
            check_launch("k_compact");
        }
        if (extra.n) {
            BBK_HIP(bbk::copy_async(out.keys.as<char>() + D * rec, extra.keys.p, extra.n * rec, hipMemcpyDeviceToDevice,
                                   ctx->stream));
            if (out_vals)
                BBK_HIP(bbk::copy_async(out.vals.as<uint32_t>() + D, extra.vals.p, extra.n * 4, hipMemcpyDeviceToDevice,
                                       ctx->stream));
        }
        // bucket table (exact HASH mode only): offsets of every bucket in the dense output
        out.nbuckets = slots ? 0 : nbuckets;
        if (!slots) {
            out.bucket_off.alloc(((size_t)nbuckets + 1) * 4);
            hipLaunchKernelGGL(k_scan_to_u32, dim3((nbuckets + 1 + 255) / 256), dim3(256), 0, ctx->stream,
                               d64.as<uint64_t>(), (uint64_t)nbuckets, D, out.bucket_off.as<uint32_t>());
            check_launch("k_scan_to_u32");
        }
        BBK_HIP(hipStreamSynchronize(ctx->stream));
        return 1;
    }

    // Histogram of the whole input over the top `bits` bits of the prefix (key arrays, KEYS / REF prefix): the
    // range passes of run_all are sized from it, so a skewed key space still gives passes that fit.
    std::vector<uint64_t> prefix_histogram(const void *d_keys, uint64_t n_records, int bits) {
        constexpr uint32_t kPartTileK = PartCfg<W>::TILE;
        const uint32_t nb = 1u << bits;
        const int w0bits = (W == 1) ? (int)(2 * k) : 64;
        DevBuf h((size_t)nb * 4 + 16);
        BBK_HIP(hipMemsetAsync(h.p, 0, (size_t)nb * 4 + 16, ctx->stream));
        PartLevel L{1, bits, nb, dmode, w0bits, nullptr, nullptr, 0u, 0u, 0, 0u};
        TileMap M{nullptr, nullptr, nullptr, 1, n_records, 0, 1, nullptr, (int)expand_k, expand_tag ? 1 : 0};
        const uint64_t nt = (n_records + kPartTileK - 1) / kPartTileK;
        BBK_REQUIRE(nt < (1ull << 32), BBK_ERR_ARG, "input of %llu records exceeds the tile space", (unsigned long long)n_records);
        const size_t rec = (size_t)W * 8;
        launch_part<false, true>("k_part_hist0", (double)(expand_k ? n_records / 2 : n_records) * rec, (uint32_t)nt,
                                 (const Key<W> *)d_keys, nullptr, M, L, h.as<uint32_t>(), nullptr, nullptr, nullptr);
        std::vector<uint32_t> h32(nb);
        BBK_HIP(hipMemcpyAsync(h32.data(), h.p, (size_t)nb * 4, hipMemcpyDeviceToHost, ctx->stream));
        BBK_HIP(hipStreamSynchronize(ctx->stream));
        std::vector<uint64_t> out(nb);
        uint64_t tot = 0;
        for (uint32_t i = 0; i < nb; ++i) tot += (out[i] = h32[i]);
        // a 32-bit counter that wrapped (one fine range above 2^32 records) shows up here
        BBK_REQUIRE(tot == n_records, BBK_ERR_ARG,
                    "key space too skewed for range passes (%llu of %llu records counted)", (unsigned long long)tot,
                    (unsigned long long)n_records);
        return out;
    }

    // Ranges of the prefix space for an input above one pass.  HASH: 2^b equal hash ranges (uniform whatever the
    // input).  KEYS / REF: consecutive fine ranges (1/512 of the prefix space) grouped up to the pass limit from
    // an exact histogram; REF additionally never lets a range straddle XXH3 buckets except as an aligned group of
    // 2^j whole buckets (see ref_bits in run()).
    bool plan_ranges(const bbk_reads *rd, const void *d_keys, uint64_t N, std::vector<Sel> &ranges) {
        const uint64_t limit = (uint64_t)((double)pass_limit(rd != nullptr) * 0.92);
        const bool verbose = getenv("BBK_VERBOSE") != nullptr;
        ranges.clear();
        if (dmode == MSD_HASH) {
            // equal spans of the 32-bit hash space, as few as fit (not a power of two: 9.6 G records of 16-byte keys take
            // 10 passes, not 16 -- every pass re-extracts all k-mers of the reads)
            const uint64_t nr = (N + limit - 1) / limit;
            if (nr > 4096) return false;
            const uint64_t span = ((1ull << 32) + nr - 1) / nr;
            for (uint64_t v = 0; v < nr; ++v) {
                Sel s;
                const uint64_t lo = v * span, hi = std::min<uint64_t>(1ull << 32, lo + span);
                if (lo >= hi) break;
                s.lo = (uint32_t)lo;
                s.span = (uint32_t)(hi - lo);
                s.finish();
                s.est = (uint64_t)((double)N * (double)(hi - lo) / 4294967296.0 * 1.04) + 1;
                ranges.push_back(s);
            }
            if (verbose) fprintf(stderr, "[bbk] msd: %llu records in %zu hash ranges\n", (unsigned long long)N, ranges.size());
            return true;
        }
        BBK_REQUIRE(rd == nullptr, BBK_ERR_INTERNAL, "reads are partitioned by hash prefix only");
        constexpr int FB = 9;  // fine ranges: 512 (the level-1 histogram kernel's bin limit)
        const std::vector<uint64_t> h = prefix_histogram(d_keys, N, FB);
        auto emit = [&](uint32_t f0, uint32_t f1, uint64_t cnt) {  // fine ranges [f0, f1)
            Sel s;
            s.lo = f0 << (32 - FB);
            const uint64_t span = (uint64_t)(f1 - f0) << (32 - FB);
            if (span >= (1ull << 32)) {  // everything (cannot happen for an input above the limit, kept for safety)
                s.span = 0;
                s.shl = 0;
            } else {
                s.span = (uint32_t)span;
                s.finish();
            }
            s.est = cnt;
            if (cnt) ranges.push_back(s);
        };
        auto greedy = [&](uint32_t f0, uint32_t f1) -> bool {  // groups the fine ranges [f0, f1)
            uint32_t g0 = f0;
            uint64_t acc = 0;
            for (uint32_t f = f0; f < f1; ++f) {
                if (h[f] > limit) return false;
                if (acc + h[f] > limit) {
                    emit(g0, f, acc);
                    g0 = f;
                    acc = 0;
                }
                acc += h[f];
            }
            emit(g0, f1, acc);
            return true;
        };
        bool ok = true;
        if (dmode == MSD_KEYS) {
            // equal aligned pieces of the key space when they fit a pass (b0 bits: 2, 4, ... 64 pieces): run_all can then
            // split the input into them with ONE partition pass instead of selecting a piece from the whole input per pass
            int b0 = 0;
            for (int b = 1; b <= 6 && !b0; ++b) {
                const uint32_t per = (1u << FB) >> b;
                bool fits = true;
                for (uint32_t f = 0; f < (1u << FB) && fits; f += per) {
                    uint64_t t = 0;
                    for (uint32_t j = 0; j < per; ++j) t += h[f + j];
                    fits = t <= limit;
                }
                if (fits) b0 = b;
            }
            if (b0) {
                const uint32_t per = (1u << FB) >> b0;
                for (uint32_t f = 0; f < (1u << FB); f += per) {
                    uint64_t t = 0;
                    for (uint32_t j = 0; j < per; ++j) t += h[f + j];
                    Sel sp;  // kept even when empty: the pieces stay equal and aligned
                    sp.lo = f << (32 - FB);
                    sp.span = per << (32 - FB);
                    sp.finish();
                    sp.est = t;
                    ranges.push_back(sp);
                }
            } else {
                ok = greedy(0, 1u << FB);
            }
        } else {  // MSD_REF: fine ranges 32 x b .. 32 x b + 31 make up XXH3 bucket b
            constexpr uint32_t per = (1u << FB) / 16;
            uint64_t bt[16];
            for (int b = 0; b < 16; ++b) {
                bt[b] = 0;
                for (uint32_t f = 0; f < per; ++f) bt[b] += h[b * per + f];
            }
            int g = 8;  // largest aligned group of whole buckets that fits a pass
            for (; g >= 1; g >>= 1) {
                bool fits = true;
                for (int b = 0; b < 16 && fits; b += g) {
                    uint64_t t = 0;
                    for (int j = 0; j < g; ++j) t += bt[b + j];
                    fits = t <= limit;
                }
                if (fits) break;
            }
            if (g >= 1) {
                for (int b = 0; b < 16; b += g) {
                    uint64_t t = 0;
                    for (int j = 0; j < g; ++j) t += bt[b + j];
                    emit((uint32_t)b * per, (uint32_t)(b + g) * per, t);
                }
            } else {
                for (int b = 0; b < 16 && ok; ++b) ok = greedy((uint32_t)b * per, (uint32_t)(b + 1) * per);
            }
        }
        if (verbose) {
            fprintf(stderr, "[bbk] msd: %llu records, prefix mode %d, %zu key ranges%s:", (unsigned long long)N, dmode,
                    ranges.size(), ok ? "" : " (a fine range exceeds one pass)");
            for (const Sel &r : ranges) fprintf(stderr, " %llu", (unsigned long long)r.est);
            fprintf(stderr, "\n");
        }
        return ok;
    }

    // Key-array input in R >= 3 equal, aligned ranges of the prefix space (KEYS: pieces of the key space; REF: groups of
    // XXH3 buckets): selecting one range from the WHOLE input per pass reads -- and for an expanded canonical array
    // regenerates -- everything R times, twice (histogram + scatter), and keeps 1/R of every tile.  Instead a "level 0"
    // partition pass writes the records grouped by range once (the counts are already known from the planning
    // histogram), and every range is then an ordinary key array.  Level 0 runs in chunks of 2^j ranges that stay below
    // the 32-bit record offsets of one pass.  false: not applicable (the caller selects per pass as before).
    bool level0_ranges(const void *d_keys, const uint32_t *d_vals, uint64_t n_in, uint64_t Nrec, const std::vector<Sel> &ranges,
                       MsdOutput &out, uint64_t &D, uint64_t &inst) {
        constexpr uint32_t kPartTileK = PartCfg<W>::TILE;
        const size_t R = ranges.size();
        if (R < 3 || (R & (R - 1)) || getenv("BBK_NO_LEVEL0")) return false;
        const uint32_t span = ranges[0].span;
        if (span == 0 || (span & (span - 1))) return false;
        for (size_t i = 0; i < R; ++i)
            if (ranges[i].span != span || ranges[i].lo != (uint32_t)(i * (uint64_t)span)) return false;
        if ((uint64_t)span * R != (1ull << 32)) return false;
        const bool verbose = getenv("BBK_VERBOSE") != nullptr;
        const bool has_val = d_vals != nullptr;
        const size_t rec = (size_t)W * 8;
        const int w0bits = (W == 1) ? (int)(2 * k) : 64;
        // ranges per chunk: the largest power of two whose chunks all stay below the pass's 32-bit offsets
        size_t m = R;
        for (; m > 1; m >>= 1) {
            bool fits = true;
            for (size_t c = 0; c < R && fits; c += m) {
                uint64_t t = 0;
                for (size_t j = 0; j < m; ++j) t += ranges[c + j].est;
                fits = t < (3500ull << 20);
            }
            if (fits) break;
        }
        if (m < 2) return false;
        int jbits = 0;
        while ((1u << jbits) < m) ++jbits;
        const uint64_t Ntot = expand_k ? 2 * n_in : n_in;  // instance space of the level-0 tiles
        const uint64_t nt = (Ntot + kPartTileK - 1) / kPartTileK;
        if (nt >= (1ull << 32)) return false;
        const unsigned ek = expand_k;
        const bool et = expand_tag;
        auto wall = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        for (size_t c = 0; c < R; c += m) {
            std::vector<uint32_t> off(m + 1, 0);
            for (size_t j = 0; j < m; ++j) off[j + 1] = off[j] + (uint32_t)ranges[c + j].est;
            const uint64_t nc = off[m];
            if (nc == 0) continue;
            const double t0 = wall();
            DevBuf buf0(nc * rec + 16), val0, cur(m * 4 + 16);
            if (has_val) val0.alloc(nc * 4 + 16);
            BBK_HIP(hipMemcpyAsync(cur.p, off.data(), m * 4, hipMemcpyHostToDevice, ctx->stream));
            Sel cs;
            cs.lo = ranges[c].lo;
            const uint64_t cspan = (uint64_t)span * m;
            if (cspan < (1ull << 32)) {
                cs.span = (uint32_t)cspan;
                cs.finish();
            }
            PartLevel L0{1, jbits, (uint32_t)m, dmode, w0bits, nullptr, nullptr, cs.lo, cs.span, cs.shl, cs.mul};
            TileMap M0{nullptr, nullptr, nullptr, 1, Ntot, 0, 1, nullptr, (int)ek, et ? 1 : 0};
            const double pb = (double)(ek ? n_in : Ntot) * (rec + (has_val ? 4 : 0)) + (double)nc * (rec + (has_val ? 4 : 0));
            if (has_val)
                launch_part<true, false>("k_part_l0", pb, (uint32_t)nt, (const Key<W> *)d_keys, d_vals, M0, L0, nullptr,
                                         cur.as<uint32_t>(), buf0.as<Key<W>>(), val0.as<uint32_t>());
            else
                launch_part<false, false>("k_part_l0", pb, (uint32_t)nt, (const Key<W> *)d_keys, nullptr, M0, L0, nullptr,
                                          cur.as<uint32_t>(), buf0.as<Key<W>>(), nullptr);
            std::vector<uint32_t> end(m);
            BBK_HIP(hipMemcpyAsync(end.data(), cur.p, m * 4, hipMemcpyDeviceToHost, ctx->stream));
            BBK_HIP(hipStreamSynchronize(ctx->stream));
            for (size_t j = 0; j < m; ++j)
                BBK_REQUIRE(end[j] == off[j + 1], BBK_ERR_INTERNAL, "level 0: range %zu received %u records, planned %u", c + j,
                            end[j] - off[j], off[j + 1] - off[j]);
            if (verbose) fprintf(stderr, "[bbk] level 0: %llu records into %zu ranges: %.3f s\n", (unsigned long long)nc, m, wall() - t0);
            // every range of the chunk: an ordinary key array (its records are what the expansion generated)
            expand_k = 0;
            expand_tag = false;
            struct Restore {
                MsdRunner *r;
                unsigned ek;
                bool et;
                ~Restore() {
                    r->expand_k = ek;
                    r->expand_tag = et;
                }
            } restore{this, ek, et};
            for (size_t j = 0; j < m; ++j) {
                const Sel &sr = ranges[c + j];
                if (sr.est == 0) continue;
                MsdOutput part;
                Dst dst{out.keys.as<char>() + D * rec, out.vals.p ? out.vals.as<uint32_t>() + D : nullptr};
                const double t1 = wall();
                // ranges of an expanded (both-strand) input spread as evenly as the whole: the key slots of the ordering
                // pass apply (no histogram passes, XCD-local fill fronts); should they not hold, the exact mode redoes it
                even_part = ek != 0 && getenv("BBK_NO_PART_KSLOTS") == nullptr;
                int rv = run(nullptr, buf0.as<char>() + (size_t)off[j] * rec, has_val ? val0.as<uint32_t>() + off[j] : nullptr,
                             sr.est, false, part, sr, nullptr, dst);
                if (rv == 4) {
                    kslots_ok = false;
                    rv = run(nullptr, buf0.as<char>() + (size_t)off[j] * rec, has_val ? val0.as<uint32_t>() + off[j] : nullptr,
                             sr.est, false, part, sr, nullptr, dst);
                }
                even_part = false;
                if (verbose) fprintf(stderr, "[bbk] key range (materialised): %.3f s\n", wall() - t1);
                BBK_REQUIRE(rv == 1, BBK_ERR_INTERNAL, "a materialised range did not sort (%d)", rv);
                D += part.n;
                inst += part.instances;
            }
        }
        (void)Nrec;
        return true;
    }

    // run() plus the split into ranges of the prefix space when the input holds more records than one pass takes
    // (what the reference does with bounded buffers, repeated DumpBuffers rounds and the run merge,
    // kmer_splitter.hpp:73-167, kmer_index_builder.hpp:281-365).  HASH prefix: every hash range is deduplicated on
    // its own (disjoint key sets) and the distinct records are concatenated.  KEYS / REF prefix: the ranges are
    // consecutive in prefix order and every pass writes straight into the final array.
    bool run_all(const bbk_reads *rd, const void *d_keys, const uint32_t *d_vals, uint64_t n_in, bool with_mask,
                 MsdOutput &out) {
        bool too_big = false;
        int r = run(rd, d_keys, d_vals, n_in, with_mask, out, Sel(), &too_big);
        if (r == 3) {  // the slot mode gave up (too much of the input overflowed its slots): exact histograms
            slots_ok = false;
            r = run(rd, d_keys, d_vals, n_in, with_mask, out, Sel(), &too_big);
        }
        if (r == 4) {  // the key slots of the ordering pass did not hold (skewed key space): exact histograms
            kslots_ok = false;
            r = run(rd, d_keys, d_vals, n_in, with_mask, out, Sel(), &too_big);
        }
        if (r != 2) return r == 1;
        const size_t rec = (size_t)W * 8;
        const bool out_vals = op != MSD_OP_NONE;
        const uint64_t Nrec = out.instances;  // records of the whole input (run() counted them before it returned 2)
        std::vector<Sel> ranges;
        if (!plan_ranges(rd, d_keys, Nrec, ranges)) return false;
        uint64_t D = 0, inst = 0;
        if (dmode != MSD_HASH) {
            // ordered output: one array for all passes (upper bound: every record distinct, which is the usual case --
            // stage B sorts a distinct set)
            out.keys.alloc(Nrec * rec + 16);
            if (out_vals) out.vals.alloc(Nrec * 4 + 16);
            if (level0_ranges(d_keys, d_vals, n_in, Nrec, ranges, out, D, inst)) {
                BBK_REQUIRE(inst == Nrec, BBK_ERR_INTERNAL, "range passes saw %llu of %llu records", (unsigned long long)inst,
                            (unsigned long long)Nrec);
                out.n = D;
                out.instances = Nrec;
                out.nbuckets = 0;
                return true;
            }
            D = 0;
            inst = 0;
            for (const Sel &sr : ranges) {
                if (sr.est == 0) continue;
                MsdOutput part;
                Dst dst{out.keys.as<char>() + D * rec, out_vals ? out.vals.as<uint32_t>() + D : nullptr};
                const double t0k = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
                const int rv = run(rd, d_keys, d_vals, n_in, with_mask, part, sr, nullptr, dst);
                if (getenv("BBK_VERBOSE"))
                    fprintf(stderr, "[bbk] key range: %.3f s\n",
                            std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0k);
                if (rv != 1) return false;
                D += part.n;
                inst += part.instances;
            }
            BBK_REQUIRE(inst == Nrec, BBK_ERR_INTERNAL, "range passes saw %llu of %llu records", (unsigned long long)inst,
                        (unsigned long long)Nrec);
            out.n = D;
            out.instances = Nrec;
            out.nbuckets = 0;
            return true;
        }
        std::vector<MsdOutput> parts(ranges.size());
        const bool verbose = getenv("BBK_VERBOSE") != nullptr;
        auto wall = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        for (size_t v = 0; v < ranges.size(); ++v) {
            MsdOutput &pt = parts[v];
            const double t0 = wall();
            int rv = run(rd, d_keys, d_vals, n_in, with_mask, pt, ranges[v], nullptr);
            if (verbose) fprintf(stderr, "[bbk] hash range %zu/%zu: %.3f s\n", v + 1, ranges.size(), wall() - t0);
            if (rv == 3) {
                slots_ok = false;
                rv = run(rd, d_keys, d_vals, n_in, with_mask, pt, ranges[v], nullptr);
            }
            if (rv != 1) return false;
            D += pt.n;
            inst += pt.instances;
            pt.bucket_off.release();
            // the slot mode sizes a part for the worst case (every record distinct): keep what is used
            if (pt.keys.bytes > pt.n * rec + (64u << 20)) {
                DevBuf ek(pt.n * rec + 16), ev;
                BBK_HIP(bbk::copy_async(ek.p, pt.keys.p, pt.n * rec, hipMemcpyDeviceToDevice, ctx->stream));
                if (out_vals) {
                    ev.alloc(pt.n * 4 + 16);
                    BBK_HIP(bbk::copy_async(ev.p, pt.vals.p, pt.n * 4, hipMemcpyDeviceToDevice, ctx->stream));
                }
                BBK_HIP(hipStreamSynchronize(ctx->stream));
                pt.keys = std::move(ek);
                if (out_vals) pt.vals = std::move(ev);
            }
        }
        out.n = D;
        out.instances = inst;
        out.keys.alloc(D * rec + 16);
        if (out_vals) out.vals.alloc(D * 4 + 16);
        uint64_t o = 0;
        for (auto &p : parts) {
            if (p.n) {
                BBK_HIP(bbk::copy_async(out.keys.as<char>() + o * rec, p.keys.p, p.n * rec, hipMemcpyDeviceToDevice,
                                       ctx->stream));
                if (out_vals)
                    BBK_HIP(bbk::copy_async(out.vals.as<uint32_t>() + o, p.vals.p, p.n * 4, hipMemcpyDeviceToDevice,
                                           ctx->stream));
            }
            o += p.n;
            BBK_HIP(hipStreamSynchronize(ctx->stream));
            p.keys.release();  // hand the part back before the next copy: peak = result + one part
            p.vals.release();
        }
        out.nbuckets = 0;
        return true;
    }
};

#ifdef BBK_PHASE_PROF
static void dump_phases() {
    unsigned long long h[6][8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase), sizeof(h)) != hipSuccess) return;
    static const char *kinds[6] = {"scatter1_reads", "scatter1_keys", "scatter2", "bucket_dist", "bucket_hash",
                                   "scatter1_narrow"};
    for (int q = 0; q < 6; ++q) {
        if (!h[q][7]) continue;
        fprintf(stderr, "[bbk phase] %-15s wgs=%llu cycles/wg:", kinds[q], h[q][7]);
        for (int p = 0; p < 7; ++p) fprintf(stderr, " p%d=%.0f", p, (double)h[q][p] / (double)h[q][7]);
        fprintf(stderr, "\n");
    }
    memset(h, 0, sizeof(h));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase), h, sizeof(h));
}
#endif

bool msd_sort_reduce(bbk_ctx *ctx, unsigned k, int dmode, int op, const bbk_reads *rd, const void *d_keys,
                     const uint32_t *d_vals, uint64_t n, bool with_mask, MsdOutput &out, unsigned tag_bits,
                     bool assume_distinct, unsigned expand_k) {
    const int W = (int)words_of(k);
    if (tag_bits) {
        // the tag sits right above the k-mer (bits [2k, 2k + tag_bits)): sort as a (k + tag_bits/2)-mer, clear the
        // tag on the way out
        BBK_REQUIRE(W == 1 && rd == nullptr && dmode == MSD_KEYS && tag_bits % 2 == 0 && 2 * k + tag_bits <= 64 &&
                        (expand_k == 0 || expand_k == k),
                    BBK_ERR_INTERNAL, "tagged sort needs 8-byte keys with %u spare bits", tag_bits);
        MsdRunner<1> r{ctx, k + tag_bits / 2, dmode, op, d_vals != nullptr};
        r.strip_mask = (2 * k >= 64) ? ~0ull : ((1ull << (2 * k)) - 1ull);
        r.assume_distinct = assume_distinct;
        r.expand_k = expand_k;
        r.expand_tag = expand_k != 0;
        return r.run_all(rd, d_keys, d_vals, n, with_mask, out);
    }
#ifdef BBK_PHASE_PROF
    struct Dump {
        ~Dump() { dump_phases(); }
    } dump_on_exit;
#endif
    if (W == 1) {
        MsdRunner<1> r{ctx, k, dmode, op, with_mask || d_vals != nullptr};
        r.assume_distinct = assume_distinct;
        r.expand_k = expand_k;
        return r.run_all(rd, d_keys, d_vals, n, with_mask, out);
    }
    if (W == 2) {
        MsdRunner<2> r{ctx, k, dmode, op, with_mask || d_vals != nullptr};
        r.assume_distinct = assume_distinct;
        r.expand_k = expand_k;
        return r.run_all(rd, d_keys, d_vals, n, with_mask, out);
    }
    if (W == 3) {
        MsdRunner<3> r{ctx, k, dmode, op, with_mask || d_vals != nullptr};
        r.assume_distinct = assume_distinct;
        r.expand_k = expand_k;
        return r.run_all(rd, d_keys, d_vals, n, with_mask, out);
    }
    if (W == 4) {
        MsdRunner<4> r{ctx, k, dmode, op, with_mask || d_vals != nullptr};
        r.assume_distinct = assume_distinct;
        r.expand_k = expand_k;
        return r.run_all(rd, d_keys, d_vals, n, with_mask, out);
    }
    return false;
}

}  // namespace bbk
