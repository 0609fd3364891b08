// kmer_ops.h -- device-side 2-bit k-mer primitives (gfx950).
//
// Layout follows RtSeq (reference common/sequence/rtseq.hpp:120-131): base i lives in bits
// [2(i%32), 2(i%32)+1] of word i/32, A=0 C=1 G=2 T=3, unused high bits zero.  Unlike the
// reference, k-mers are not rolled base by base: a k-mer at read position p is a bit-field
// of the packed read, extracted with two 64-bit funnel shifts per word, and its reverse
// complement is a bit reversal (v_bfrev) + pair swap + shift -- O(words), no loop over bases.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bbk {

template <int W>
struct Key {
    uint64_t w[W];
};

template <int W>
__host__ __device__ inline bool key_eq(const Key<W> &a, const Key<W> &b) {
    bool e = true;
#pragma unroll
    for (int i = 0; i < W; ++i) e = e && (a.w[i] == b.w[i]);
    return e;
}

// word order, word 0 most significant (reference adt/array_vector.hpp:114-123)
template <int W>
__host__ __device__ inline bool key_less_words(const Key<W> &a, const Key<W> &b) {
#pragma unroll
    for (int i = 0; i < W; ++i) {
        if (a.w[i] != b.w[i]) return a.w[i] < b.w[i];
    }
    return false;
}

// c ? a : b, word by word (a struct-level ?: makes hipcc spill both operands to scratch)
template <int W>
__host__ __device__ inline Key<W> key_select(bool c, const Key<W> &a, const Key<W> &b) {
    Key<W> r;
#pragma unroll
    for (int i = 0; i < W; ++i) r.w[i] = c ? a.w[i] : b.w[i];
    return r;
}

template <int W>
__device__ inline Key<W> key_load(const Key<W> *__restrict__ p) {
    Key<W> r;
    const uint64_t *q = reinterpret_cast<const uint64_t *>(p);
#pragma unroll
    for (int i = 0; i < W; ++i) r.w[i] = q[i];
    return r;
}

template <int W>
__device__ inline void key_store(Key<W> *__restrict__ p, const Key<W> &v) {
    uint64_t *q = reinterpret_cast<uint64_t *>(p);
#pragma unroll
    for (int i = 0; i < W; ++i) q[i] = v.w[i];
}

__device__ inline uint64_t rev2(uint64_t v) {  // reverse the order of the 32 2-bit groups
    v = __brevll(v);
    return ((v & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((v & 0x5555555555555555ull) << 1);
}

// k-mer starting at base p of a read whose packed words start at rw.
template <int W>
__device__ inline Key<W> kmer_extract(const uint64_t *__restrict__ rw, uint32_t p, int k) {
    Key<W> r;
    const uint32_t wi = p >> 5;
    const uint32_t sh = (p & 31u) << 1;
    const uint32_t endw = (2u * (p + (uint32_t)k) - 1u) >> 6;  // last word holding a base of the k-mer
#pragma unroll
    for (int i = 0; i < W; ++i) {
        uint64_t lo = (wi + i <= endw) ? rw[wi + i] : 0ull;
        uint64_t v = lo >> sh;
        if (sh != 0 && wi + i + 1 <= endw) v |= rw[wi + i + 1] << (64u - sh);
        r.w[i] = v;
    }
    const int vb = 2 * k - 64 * (W - 1);  // populated bits of the last word (1..64)
    if (vb < 64) r.w[W - 1] &= (1ull << vb) - 1ull;
    return r;
}

// 8-byte keys (k <= 32): branch-free extraction (two unconditional word loads, the second clamped to
// the read's last word) and one-reversal canonicalisation.  With R = rev2(fwd) (the k-mer read
// backwards, left aligned) the reverse complement is (~R) >> (64-2k), and comparing fwd with it in
// base order (rtseq.hpp:407-415) is comparing R with (~fwd) << (64-2k) numerically.
__device__ inline Key<1> kmer_extract_canon1(const uint64_t *__restrict__ rw, uint32_t p, int k, uint32_t last_word,
                                             bool *minimal) {
    const uint32_t wi = p >> 5;
    const uint32_t sh = (p & 31u) << 1;
    const uint64_t lo = rw[wi];
    const uint64_t hi = rw[wi + 1 <= last_word ? wi + 1 : last_word];
    uint64_t fwd = (lo >> sh) | ((hi << 1) << (63u - sh));
    const uint32_t pad = 64u - 2u * (uint32_t)k;
    fwd = (fwd << pad) >> pad;
    const uint64_t R = rev2(fwd);
    const uint64_t C = (~fwd) << pad;
    *minimal = R <= C;
    Key<1> o;
    o.w[0] = *minimal ? fwd : ((~R) >> pad);
    return o;
}

// cheap 32-bit mixing hash for the partition levels (three 32-bit multiplies; the 64-bit
// splitmix of owner_mix costs ~4x more on this hardware and only ~20 top bits are consumed)
template <int W>
__device__ inline uint32_t part_hash32(const Key<W> &key) {
    uint32_t h = 0x9E3779B9u;
#pragma unroll
    for (int i = 0; i < W; ++i) {
        h ^= (uint32_t)key.w[i];
        h *= 0x85EBCA6Bu;
        h ^= h >> 15;
        h += (uint32_t)(key.w[i] >> 32) * 0xC2B2AE35u;
        h ^= h >> 13;
    }
    h *= 0x27D4EB2Fu;
    h ^= h >> 16;
    return h;
}

__device__ inline uint32_t base_at(const uint64_t *__restrict__ rw, uint32_t p) {
    return (uint32_t)((rw[p >> 5] >> ((p & 31u) << 1)) & 3ull);
}

// reverse complement (RtSeq::operator!, rtseq.hpp:79-115,387-400)
template <int W>
__device__ inline Key<W> kmer_rc(const Key<W> &x, int k) {
    uint64_t r[W];
#pragma unroll
    for (int i = 0; i < W; ++i) r[i] = rev2(~x.w[W - 1 - i]);
    const uint32_t sh = 2u * (32u * W - (uint32_t)k);  // < 64 because k > 32(W-1)
    Key<W> o;
#pragma unroll
    for (int i = 0; i < W; ++i) {
        uint64_t v = r[i] >> sh;
        if (sh != 0 && i + 1 < W) v |= r[i + 1] << (64u - sh);
        o.w[i] = v;
    }
    return o;
}

// base-lexicographic a < b (RtSeq operator<, rtseq.hpp:732-741): base 0 most significant
template <int W>
__device__ inline bool kmer_less_nucl(const Key<W> &a, const Key<W> &b) {
#pragma unroll
    for (int i = 0; i < W; ++i) {
        if (a.w[i] != b.w[i]) return rev2(a.w[i]) < rev2(b.w[i]);
    }
    return false;
}

// drop base 0, append c as base k-1 (RtSeq::operator<<=, rtseq.hpp:450-467)
template <int W>
__device__ inline Key<W> kmer_shl(const Key<W> &x, int k, uint32_t c) {
    Key<W> o;
#pragma unroll
    for (int i = 0; i < W - 1; ++i) o.w[i] = (x.w[i] >> 2) | ((x.w[i + 1] & 3ull) << 62);
    const uint32_t lastshift = (uint32_t)(((k + 31) & 31) << 1);
    o.w[W - 1] = (x.w[W - 1] >> 2) | ((uint64_t)(c & 3u) << lastshift);
    return o;
}

template <int W>
__device__ inline uint32_t kmer_base(const Key<W> &x, int i) {
    return (uint32_t)((x.w[i >> 5] >> ((i & 31) << 1)) & 3ull);
}

// ---- XXH3-64 (xxHash 0.8.0, seed 0, default secret) for 8/16/24/32-byte inputs -----------------
// Restated from the published algorithm; the reference calls it through RtSeq::GetHash
// (rtseq.hpp:681-687) -> KMerSegmentPolicy (utils/kmer_mph/kmer_buckets.hpp:28-33).
__host__ __device__ inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

__device__ inline uint64_t mul128_fold64(uint64_t a, uint64_t b) { return (a * b) ^ __umul64hi(a, b); }

__device__ inline uint64_t xxh3_avalanche(uint64_t h) {
    h ^= h >> 37;
    h *= 0x165667919E3779F9ull;
    h ^= h >> 32;
    return h;
}

template <int W>
__device__ inline uint64_t xxh3_64(const Key<W> &key) {
    constexpr uint64_t s0 = 0xbe4ba423396cfeb8ull, s8 = 0x1cad21f72c81017cull, s16 = 0xdb979083e96dd4deull,
                       s24 = 0x1f67b3b7a4a44072ull, s32 = 0x78e5c0cc4ee679cbull, s40 = 0x2172ffcc7dd05a82ull,
                       s48 = 0x8e2443f7744608b8ull;
    if constexpr (W == 1) {
        uint64_t x = key.w[0];
        uint64_t h = ((x >> 32) + (x << 32)) ^ (s8 ^ s16);
        h ^= rotl64(h, 49) ^ rotl64(h, 24);
        h *= 0x9FB21C651E98DF25ull;
        h ^= (h >> 35) + 8;
        h *= 0x9FB21C651E98DF25ull;
        return h ^ (h >> 28);
    } else if constexpr (W == 2) {
        uint64_t lo = key.w[0] ^ (s24 ^ s32);
        uint64_t hi = key.w[1] ^ (s40 ^ s48);
        uint64_t sw = __builtin_bswap64(lo);
        uint64_t acc = 16ull + sw + hi + mul128_fold64(lo, hi);
        return xxh3_avalanche(acc);
    } else {
        uint64_t acc = (uint64_t)(8 * W) * 0x9E3779B185EBCA87ull;
        acc += mul128_fold64(key.w[0] ^ s0, key.w[1] ^ s8);
        acc += mul128_fold64(key.w[W - 2] ^ s16, key.w[W - 1] ^ s24);
        return xxh3_avalanche(acc);
    }
}

// multi-GPU owner hash (hash, not raw prefix, so skewed inputs stay balanced)
template <int W>
__device__ inline uint64_t owner_mix(const Key<W> &key) {
    uint64_t x = 0x9E3779B97F4A7C15ull;
#pragma unroll
    for (int i = 0; i < W; ++i) {
        x ^= key.w[i];
        x ^= x >> 30;
        x *= 0xBF58476D1CE4E5B9ull;
        x ^= x >> 27;
        x *= 0x94D049BB133111EBull;
        x ^= x >> 31;
    }
    return x;
}

// 8-bit reversal: InOutMask::conjugate (kmer_extension_index.hpp:19-40,87-90)
__host__ __device__ inline uint32_t rev8(uint32_t m) {
    m = ((m & 0xF0u) >> 4) | ((m & 0x0Fu) << 4);
    m = ((m & 0xCCu) >> 2) | ((m & 0x33u) << 2);
    m = ((m & 0xAAu) >> 1) | ((m & 0x55u) << 1);
    return m;
}

// The sorted table of distinct k-mers IS the index (it replaces the BooPHF of utils/kmer_mph/kmer_index.hpp:85-90):
// a prefix table over the top bits of word 0 bounds a short binary search.  Record indices are 64-bit like the
// reference's size_t; the prefix table holds u32 entries below 2^32 - 2 records and u64 entries above (`wide`,
// uniform over a launch).
constexpr uint64_t kNotFound = ~0ull;
struct PrefixTable {
    const void *pref;
    int pshift;
    int wide;
};
template <int W>
__device__ inline uint64_t table_find(const Key<W> *__restrict__ keys, const PrefixTable &P, const Key<W> &q) {
    const uint64_t t = q.w[0] >> P.pshift;
    uint64_t lo, hi;
    if (P.wide) {
        const uint64_t *p = reinterpret_cast<const uint64_t *>(P.pref);
        lo = p[t];
        hi = p[t + 1];
    } else {
        const uint32_t *p = reinterpret_cast<const uint32_t *>(P.pref);
        lo = p[t];
        hi = p[t + 1];
    }
    while (lo < hi) {
        const uint64_t mid = lo + ((hi - lo) >> 1);
        const Key<W> km = key_load<W>(&keys[mid]);
        if (key_eq<W>(km, q)) return mid;
        if (key_less_words<W>(km, q)) lo = mid + 1;
        else hi = mid;
    }
    return kNotFound;
}

}  // namespace bbk
