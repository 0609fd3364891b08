/*
 * bbk_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 * See bbk_oracle.h for the scope, the parity pins and the citation convention.
 * Paths in comments are relative to /root/reference/assembler/src.
 */
#include "bbk_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ */
/* small growable array of fixed-width uint64 records                  */
/* ------------------------------------------------------------------ */
typedef struct {
    uint64_t *d;
    size_t n, cap;
} u64vec;

static int u64vec_reserve(u64vec *v, size_t want) {
    if (want <= v->cap) return 0;
    size_t nc = v->cap ? v->cap : 1024;
    while (nc < want) nc *= 2;
    uint64_t *p = (uint64_t *)realloc(v->d, nc * sizeof(uint64_t));
    if (!p) return -1;
    v->d = p;
    v->cap = nc;
    return 0;
}

static inline int u64vec_push(u64vec *v, const uint64_t *rec, int nw) {
    if (v->n + (size_t)nw > v->cap && u64vec_reserve(v, v->n + (size_t)nw)) return -1;
    for (int i = 0; i < nw; ++i) v->d[v->n + i] = rec[i];
    v->n += (size_t)nw;
    return 0;
}

/* ------------------------------------------------------------------ */
/* sequence primitives                                                 */
/* ------------------------------------------------------------------ */
int orc_words(int k) { return (k + 31) >> 5; } /* rtseq.hpp:129-131 */

int orc_is_nucl(char c) { /* nucl.hpp:45-62: ACGT, acgt */
    switch (c) {
        case 'A': case 'C': case 'G': case 'T':
        case 'a': case 'c': case 'g': case 't':
            return 1;
        default:
            return 0;
    }
}

int orc_dignucl(char c) { /* nucl.hpp:120-130 */
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return (int)c & 3; /* already 0..3 */
    }
}

int orc_kmer_get(const orc_kmer *x, int i) { /* base i = bits 2(i%32) of word i/32 */
    return (int)((x->w[i >> 5] >> ((i & 31) << 1)) & 3);
}

static inline void kmer_set(orc_kmer *x, int i, int c) {
    x->w[i >> 5] |= ((uint64_t)(c & 3)) << ((i & 31) << 1);
}

void orc_kmer_from_ascii(const char *s, int k, orc_kmer *out) { /* rtseq.hpp:289-330 */
    memset(out, 0, sizeof(*out));
    for (int i = 0; i < k; ++i) kmer_set(out, i, orc_dignucl(s[i]));
}

void orc_kmer_to_ascii(const orc_kmer *x, int k, char *out) {
    static const char nt[4] = {'A', 'C', 'G', 'T'};
    for (int i = 0; i < k; ++i) out[i] = nt[orc_kmer_get(x, i)];
    out[k] = 0;
}

/* rtseq.hpp:450-467 operator<<= : drop base 0, append c as base k-1 */
void orc_kmer_shl(orc_kmer *x, int k, int c) {
    int nw = orc_words(k);
    if (nw == 0) return;
    for (int i = 0; i < nw - 1; ++i)
        x->w[i] = (x->w[i] >> 2) | ((x->w[i + 1] & 3) << 62);
    unsigned lastshift = (unsigned)(((k + 31) & 31) << 1);
    x->w[nw - 1] = (x->w[nw - 1] >> 2) | ((uint64_t)(c & 3) << lastshift);
}

/* rtseq.hpp:387-400 operator! (FastRC, :79-115): base i of result = 3 - base(k-1-i) */
void orc_kmer_rc(const orc_kmer *x, int k, orc_kmer *out) {
    orc_kmer r;
    memset(&r, 0, sizeof(r));
    for (int i = 0; i < k; ++i) kmer_set(&r, i, 3 - orc_kmer_get(x, k - 1 - i));
    *out = r;
}

/* rtseq.hpp:407-415 */
int orc_kmer_is_minimal(const orc_kmer *x, int k) {
    for (int i = 0; (i << 1) + 1 <= k; ++i) {
        int front = orc_kmer_get(x, i);
        int end = 3 - orc_kmer_get(x, k - 1 - i);
        if (front != end) return front < end;
    }
    return 1;
}

/* rtseq.hpp:732-741 free operator< : base-lexicographic */
int orc_kmer_less_nucl(const orc_kmer *a, const orc_kmer *b, int k) {
    for (int i = 0; i < k; ++i) {
        int x = orc_kmer_get(a, i), y = orc_kmer_get(b, i);
        if (x != y) return x < y;
    }
    return 0;
}

/* adt/array_vector.hpp:114-123: word 0 first, numeric */
int orc_kmer_cmp_words(const uint64_t *a, const uint64_t *b, int nw) {
    for (int i = 0; i < nw; ++i) {
        if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* XXH3-64, seed 0, default secret; inputs of 8/16/24/32 bytes          */
/* (xxHash 0.8.0, ext/include/xxh/xxhash.h)                             */
/* ------------------------------------------------------------------ */
/* little-endian 64-bit words of the default secret (xxhash.h:2513-2526), offsets 0..64 */
static const uint64_t kSec0 = 0xbe4ba423396cfeb8ULL;   /* secret+0  */
static const uint64_t kSec8 = 0x1cad21f72c81017cULL;   /* secret+8  */
static const uint64_t kSec16 = 0xdb979083e96dd4deULL;  /* secret+16 */
static const uint64_t kSec24 = 0x1f67b3b7a4a44072ULL;  /* secret+24 */
static const uint64_t kSec32 = 0x78e5c0cc4ee679cbULL;  /* secret+32 */
static const uint64_t kSec40 = 0x2172ffcc7dd05a82ULL;  /* secret+40 */
static const uint64_t kSec48 = 0x8e2443f7744608b8ULL;  /* secret+48 */

static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t bswap64(uint64_t x) { return __builtin_bswap64(x); }

/* ---- Sequence views (common/sequence/sequence.hpp) ------------------------------------------------------- */
int orc_complement(int code) { return 3 - code; }            /* nucl.hpp:27-34 */
char orc_nucl(int code) { return "ACGT"[code & 3]; }         /* nucl.hpp:105-112 */
orc_seq orc_seq_make(const char *acgt) {
    orc_seq s = {acgt, 0, strlen(acgt), 0};
    return s;
}
int orc_seq_at(const orc_seq *s, size_t i) { /* sequence.hpp:311-319 */
    if (s->rtl) return orc_complement(orc_dignucl(s->data[s->from + s->size - 1 - i]));
    return orc_dignucl(s->data[s->from + i]);
}
orc_seq orc_seq_rc(const orc_seq *s) { /* operator! :232-234 */
    orc_seq r = *s;
    r.rtl = !s->rtl;
    return r;
}
orc_seq orc_seq_subseq(const orc_seq *s, size_t from, size_t to) { /* :324-333 */
    orc_seq r = *s;
    r.size = to - from;
    r.from = s->rtl ? s->from + s->size - to : s->from + from;
    return r;
}
int orc_seq_less(const orc_seq *a, const orc_seq *b) { /* :222-230 */
    const size_t n = a->size < b->size ? a->size : b->size;
    for (size_t i = 0; i < n; ++i) {
        const int x = orc_seq_at(a, i), y = orc_seq_at(b, i);
        if (x != y) return x < y;
    }
    return a->size < b->size;
}
int orc_seq_eq(const orc_seq *a, const orc_seq *b) {
    if (a->size != b->size) return 0;
    for (size_t i = 0; i < a->size; ++i)
        if (orc_seq_at(a, i) != orc_seq_at(b, i)) return 0;
    return 1;
}
void orc_seq_str(const orc_seq *s, char *out) { /* :377-383 */
    for (size_t i = 0; i < s->size; ++i) out[i] = orc_nucl(orc_seq_at(s, i));
    out[s->size] = 0;
}
void orc_seq_concat(const orc_seq *a, const orc_seq *b, char *out) { /* :361-362 */
    orc_seq_str(a, out);
    orc_seq_str(b, out + a->size);
}

uint64_t orc_mulhi64(uint64_t x, uint64_t y) { /* lemiere_mod_reduce.hpp:17-35 */
    return (uint64_t)(((__uint128_t)x * (__uint128_t)y) >> 64);
}

static inline uint64_t mul128_fold64(uint64_t a, uint64_t b) { /* xxhash.h:2683-2688 */
    __uint128_t p = (__uint128_t)a * (__uint128_t)b;
    return (uint64_t)p ^ (uint64_t)(p >> 64);
}

static inline uint64_t xxh3_avalanche(uint64_t h) { /* xxhash.h:2701-2707 */
    h ^= h >> 37;
    h *= 0x165667919E3779F9ULL;
    h ^= h >> 32;
    return h;
}

uint64_t orc_xxh3_64(const uint64_t *w, int nw) {
    if (nw == 1) { /* XXH3_len_4to8_64b with len = 8 (xxhash.h:2781-2794) + rrmxmx (:2714-2722) */
        uint64_t x = w[0];
        uint64_t input64 = (x >> 32) + ((x & 0xffffffffULL) << 32);
        uint64_t h = input64 ^ (kSec8 ^ kSec16);
        h ^= rotl64(h, 49) ^ rotl64(h, 24);
        h *= 0x9FB21C651E98DF25ULL;
        h ^= (h >> 35) + 8;
        h *= 0x9FB21C651E98DF25ULL;
        return h ^ (h >> 28);
    }
    if (nw == 2) { /* XXH3_len_9to16_64b, len = 16 (xxhash.h:2797-2811) */
        uint64_t lo = w[0] ^ (kSec24 ^ kSec32);
        uint64_t hi = w[1] ^ (kSec40 ^ kSec48);
        uint64_t acc = 16 + bswap64(lo) + hi + mul128_fold64(lo, hi);
        return xxh3_avalanche(acc);
    }
    /* XXH3_len_17to128_64b with 16 < len <= 32 (xxhash.h:2884-2908), mix16B (:2850-2881) */
    uint64_t len = (uint64_t)nw * 8;
    uint64_t acc = len * 0x9E3779B185EBCA87ULL;
    acc += mul128_fold64(w[0] ^ kSec0, w[1] ^ kSec8);
    acc += mul128_fold64(w[nw - 2] ^ kSec16, w[nw - 1] ^ kSec24);
    return xxh3_avalanche(acc);
}

uint64_t orc_bucket(const uint64_t *w, int nw, uint64_t nb) { /* kmer_buckets.hpp:28-33 */
    if (nb == 1) return 0;
    return orc_mulhi64(orc_xxh3_64(w, nw), nb);
}

/* ------------------------------------------------------------------ */
/* read normalisation                                                  */
/* ------------------------------------------------------------------ */
void orc_longest_valid(const char *s, size_t len, size_t *from, size_t *to) {
    /* longest_valid_wrapper.hpp:15-41: longest maximal run of is_nucl, first wins on ties */
    const size_t none = (size_t)-1;
    size_t best_len = 0, best_pos = none, pos = none;
    for (size_t i = 0; i <= len; ++i) {
        if (i < len && orc_is_nucl(s[i])) {
            if (pos == none) pos = i;
        } else {
            if (pos != none) {
                size_t l = i - pos;
                if (l > best_len) {
                    best_len = l;
                    best_pos = pos;
                }
            }
            pos = none;
        }
    }
    if (best_len == 0) {
        *from = 0;
        *to = 0;
    } else {
        *from = best_pos;
        *to = best_pos + best_len;
    }
}

/* ------------------------------------------------------------------ */
/* sorting of fixed-width records (word order)                          */
/* ------------------------------------------------------------------ */
static int g_cmp_nw_tls(void);
#if defined(_OPENMP)
static int cmp_nw_storage;
#pragma omp threadprivate(cmp_nw_storage)
#else
static int cmp_nw_storage;
#endif
static int g_cmp_nw_tls(void) { return cmp_nw_storage; }

static int cmp_rec(const void *a, const void *b) {
    return orc_kmer_cmp_words((const uint64_t *)a, (const uint64_t *)b, g_cmp_nw_tls());
}

static void sort_u64(uint64_t *a, size_t n) { /* introsort-lite for the W=1 case */
    while (n > 24) {
        uint64_t x = a[0], y = a[n / 2], z = a[n - 1];
        uint64_t p = x < y ? (y < z ? y : (x < z ? z : x)) : (x < z ? x : (y < z ? z : y));
        size_t i = 0, j = n - 1;
        for (;;) {
            while (a[i] < p) ++i;
            while (a[j] > p) --j;
            if (i >= j) break;
            uint64_t t = a[i];
            a[i] = a[j];
            a[j] = t;
            ++i;
            --j;
        }
        size_t ln = j + 1, rn = n - ln;
        if (ln < rn) {
            sort_u64(a, ln);
            a += ln;
            n = rn;
        } else {
            sort_u64(a + ln, rn);
            n = ln;
        }
    }
    for (size_t i = 1; i < n; ++i) {
        uint64_t v = a[i];
        size_t j = i;
        while (j > 0 && a[j - 1] > v) {
            a[j] = a[j - 1];
            --j;
        }
        a[j] = v;
    }
}

/* sort + unique (kmer_splitter.hpp:135-141: libcxx::sort + std::unique); returns new count.
 * If cnt != NULL it receives the run lengths (cnt must hold n entries). */
static size_t sort_unique_records(uint64_t *d, size_t n, int nw, uint32_t *cnt) {
    if (n == 0) return 0;
    if (nw == 1) {
        sort_u64(d, n);
    } else {
        cmp_nw_storage = nw;
        qsort(d, n, (size_t)nw * sizeof(uint64_t), cmp_rec);
    }
    size_t m = 0;
    uint32_t run = 1;
    for (size_t i = 1; i < n; ++i) {
        if (orc_kmer_cmp_words(d + i * nw, d + m * nw, nw) != 0) {
            if (cnt) cnt[m] = run;
            run = 1;
            ++m;
            if (m != i) memcpy(d + m * nw, d + i * nw, (size_t)nw * sizeof(uint64_t));
        } else {
            if (run != UINT32_MAX) ++run;
        }
    }
    if (cnt) cnt[m] = run;
    return m + 1;
}

/* ------------------------------------------------------------------ */
/* splitter: reads (+RC) -> per-bucket raw k-mers                       */
/* ------------------------------------------------------------------ */
/*
 * Follows BufferFiller::operator() (projects/kmercount/main.cpp:64-82) when
 * only_minimal == 0 and DeBruijnKMerSplitter::FillBufferFromSequence
 * (utils/kmer_mph/kmer_splitters.hpp:25-41) with StoringTypeFilter<Invertable>
 * (utils/ph_map/storing_traits.hpp:90-101) when only_minimal != 0.
 * The read is first cut to its longest valid run (io_helper.cpp:19-32,
 * longest_valid_wrapper.hpp) and then emitted twice: as is and reverse
 * complemented (rc_reader_wrapper.hpp:33-42).
 */
static int split_one_strand(const char *s, size_t len, int rc, int K, int only_minimal,
                            unsigned nb, u64vec *buckets) {
    if (len < (size_t)K) return 0;
    int nw = orc_words(K);
    orc_kmer kmer;
    memset(&kmer, 0, sizeof(kmer));
    /* seq.start<RtSeq>(K) >> 'A' : first K-1 bases in positions 1..K-1 (rtseq.hpp:560-579) */
    for (int i = 0; i < K - 1; ++i) {
        int c = rc ? 3 - orc_dignucl(s[len - 1 - (size_t)i]) : orc_dignucl(s[i]);
        kmer_set(&kmer, i + 1, c);
    }
    for (size_t j = (size_t)K - 1; j < len; ++j) {
        int c = rc ? 3 - orc_dignucl(s[len - 1 - j]) : orc_dignucl(s[j]);
        orc_kmer_shl(&kmer, K, c);
        if (only_minimal && !orc_kmer_is_minimal(&kmer, K)) continue;
        uint64_t b = orc_bucket(kmer.w, nw, nb);
        if (u64vec_push(&buckets[b], kmer.w, nw)) return -1;
    }
    return 0;
}

typedef struct {
    unsigned nb;
    int nw;
    u64vec *b;       /* nb buckets, sorted unique after count */
    uint32_t **cnt;  /* per bucket multiplicities (optional) */
} bucket_set;

static void bucket_set_free(bucket_set *s) {
    if (s->b) {
        for (unsigned i = 0; i < s->nb; ++i) free(s->b[i].d);
        free(s->b);
    }
    if (s->cnt) {
        for (unsigned i = 0; i < s->nb; ++i) free(s->cnt[i]);
        free(s->cnt);
    }
    memset(s, 0, sizeof(*s));
}

/*
 * KMerDiskCounter::Count (utils/kmer_mph/kmer_index_builder.hpp:241-267):
 * Split into nb buckets, then per bucket sort + unique (the `.idx`-less branch of
 * MergeKMers, :281-365; with several runs the loser-tree merge yields the same set).
 */
static int count_from_reads(const orc_reads *reads, int K, int only_minimal, unsigned nb,
                            int nthreads, int want_counts, bucket_set *out) {
    int nw = orc_words(K);
    memset(out, 0, sizeof(*out));
    out->nb = nb;
    out->nw = nw;
    if (nthreads < 1) nthreads = 1;
    int err = 0;
    /* per-thread x per-bucket cells (kmer_splitter.hpp:73-118) */
    u64vec *cells = (u64vec *)calloc((size_t)nthreads * nb, sizeof(u64vec));
    if (!cells) return -1;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 256)
    for (size_t r = 0; r < reads->n; ++r) {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        const char *s = reads->bases + reads->offsets[r];
        size_t len = (size_t)(reads->offsets[r + 1] - reads->offsets[r]);
        size_t from, to;
        orc_longest_valid(s, len, &from, &to);
        u64vec *mine = cells + (size_t)tid * nb;
        if (split_one_strand(s + from, to - from, 0, K, only_minimal, nb, mine) ||
            split_one_strand(s + from, to - from, 1, K, only_minimal, nb, mine)) {
#pragma omp atomic write
            err = 1;
        }
    }
    out->b = (u64vec *)calloc(nb, sizeof(u64vec));
    if (want_counts) out->cnt = (uint32_t **)calloc(nb, sizeof(uint32_t *));
    if (!out->b || (want_counts && !out->cnt)) err = 1;
    if (!err) {
        /* DumpBuffers gathers one bucket across threads (kmer_splitter.hpp:120-167) */
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
        for (unsigned b = 0; b < nb; ++b) {
            size_t tot = 0;
            for (int t = 0; t < nthreads; ++t) tot += cells[(size_t)t * nb + b].n;
            u64vec v = {0, 0, 0};
            if (tot && u64vec_reserve(&v, tot)) {
#pragma omp atomic write
                err = 1;
                continue;
            }
            for (int t = 0; t < nthreads; ++t) {
                u64vec *c = &cells[(size_t)t * nb + b];
                if (c->n) memcpy(v.d + v.n, c->d, c->n * sizeof(uint64_t));
                v.n += c->n;
                free(c->d);
                c->d = NULL;
                c->n = c->cap = 0;
            }
            size_t nrec = v.n / (size_t)nw;
            uint32_t *cnt = NULL;
            if (want_counts && nrec) cnt = (uint32_t *)malloc(nrec * sizeof(uint32_t));
            size_t m = sort_unique_records(v.d, nrec, nw, cnt);
            v.n = m * (size_t)nw;
            out->b[b] = v;
            if (want_counts) out->cnt[b] = cnt;
        }
    }
    for (size_t i = 0; i < (size_t)nthreads * nb; ++i) free(cells[i].d);
    free(cells);
    if (err) {
        bucket_set_free(out);
        return -1;
    }
    return 0;
}

/* KMerDiskStorage::merge (kmer_index_builder.hpp:168-181): plain concatenation in bucket order */
static int merge_buckets(const bucket_set *s, uint64_t **out, size_t *n_out, uint32_t **counts,
                         size_t *bucket_start) {
    size_t tot = 0;
    for (unsigned b = 0; b < s->nb; ++b) {
        if (bucket_start) bucket_start[b] = tot / (size_t)s->nw;
        tot += s->b[b].n;
    }
    if (bucket_start) bucket_start[s->nb] = tot / (size_t)s->nw;
    size_t nrec = tot / (size_t)s->nw;
    uint64_t *d = (uint64_t *)malloc((tot ? tot : 1) * sizeof(uint64_t));
    uint32_t *c = NULL;
    if (counts) c = (uint32_t *)malloc((nrec ? nrec : 1) * sizeof(uint32_t));
    if (!d || (counts && !c)) {
        free(d);
        free(c);
        return -1;
    }
    size_t off = 0;
    for (unsigned b = 0; b < s->nb; ++b) {
        if (s->b[b].n) memcpy(d + off, s->b[b].d, s->b[b].n * sizeof(uint64_t));
        if (c && s->b[b].n) memcpy(c + off / (size_t)s->nw, s->cnt[b], (s->b[b].n / (size_t)s->nw) * sizeof(uint32_t));
        off += s->b[b].n;
    }
    *out = d;
    *n_out = nrec;
    if (counts) *counts = c;
    return 0;
}

int orc_kmercount(const orc_reads *reads, int k, unsigned nbuckets, int nthreads,
                  uint64_t **out, size_t *n_out, uint32_t **counts) {
    /* projects/kmercount/main.cpp:214-219: CountAll(16, T, merge=true) */
    if (k < 1 || k >= 128 || nbuckets == 0) return -1;
    bucket_set s;
    if (count_from_reads(reads, k, /*only_minimal=*/0, nbuckets, nthreads, counts != NULL, &s)) return -1;
    int rc = merge_buckets(&s, out, n_out, counts, NULL);
    bucket_set_free(&s);
    return rc;
}

/* ------------------------------------------------------------------ */
/* extension index                                                     */
/* ------------------------------------------------------------------ */
void orc_extindex_free(orc_extindex *x) {
    free(x->kp1);
    free(x->kp1_count);
    free(x->kmers);
    free(x->masks);
    free(x->bucket_start);
    memset(x, 0, sizeof(*x));
}

/* position of a canonical k-mer inside the merged k-mer file: bucket (kmer_index.hpp:85-90
 * plays this role via the per-bucket MPHF; any bijection onto [0,n_k) is equivalent) */
size_t orc_extindex_find(const orc_extindex *x, const orc_kmer *canon) {
    int nw = orc_words(x->k);
    uint64_t b = orc_bucket(canon->w, nw, x->nbuckets);
    size_t lo = x->bucket_start[b], hi = x->bucket_start[b + 1];
    while (lo < hi) {
        size_t mid = lo + (hi - lo) / 2;
        int c = orc_kmer_cmp_words(x->kmers + mid * nw, canon->w, nw);
        if (c == 0) return mid;
        if (c < 0) lo = mid + 1;
        else hi = mid;
    }
    return (size_t)-1;
}

/* InvertableKeyWithHash (utils/ph_map/key_with_hash.hpp:108-207): canonical form + is_minimal */
typedef struct {
    orc_kmer key;   /* oriented k-mer */
    size_t idx;     /* index of its canonical form */
    int minimal;    /* key is the canonical form */
} kwh_t;

static int kwh_make(const orc_extindex *x, const orc_kmer *key, kwh_t *out) {
    out->key = *key;
    out->minimal = orc_kmer_is_minimal(key, x->k);
    if (out->minimal) {
        out->idx = orc_extindex_find(x, key);
    } else {
        orc_kmer r;
        orc_kmer_rc(key, x->k, &r);
        out->idx = orc_extindex_find(x, &r);
    }
    return out->idx == (size_t)-1 ? -1 : 0;
}

static inline uint8_t invert_byte(uint8_t a) { /* kmer_extension_index.hpp:19-40 */
    uint8_t r = 0;
    for (int i = 0; i < 8; ++i) {
        r = (uint8_t)((r << 1) | (a & 1));
        a >>= 1;
    }
    return r;
}

/* InvertableStoring::get_value (storing_traits.hpp:30-68) */
static inline uint8_t kwh_mask(const orc_extindex *x, const kwh_t *k) {
    uint8_t m = x->masks[k->idx];
    return k->minimal ? m : invert_byte(m);
}

static const int8_t kUniqueNext[16] = {-1, 0, 1, -1, 2, -1, -1, -1, 3, -1, -1, -1, -1, -1, -1, -1};
static inline int mask_unique_out(uint8_t m) { return kUniqueNext[m & 0xF] >= 0; }  /* :46-51,148-150 */
static inline int mask_unique_in(uint8_t m) { return kUniqueNext[(m >> 4) & 0xF] >= 0; }
static inline int mask_is_junction(uint8_t m) { return !mask_unique_out(m) || !mask_unique_in(m); } /* :144-146 */

int orc_extindex_build(const orc_reads *reads, int k, unsigned T, orc_extindex *out) {
    memset(out, 0, sizeof(*out));
    if (k < 1 || k + 1 >= 128 || T == 0) return -1;
    int nthreads = (int)T;
    out->k = k;
    out->nbuckets = 10 * T; /* kmer_extension_index_builder.hpp:73 */
    int nw1 = orc_words(k + 1), nw = orc_words(k);

    /* step 1: canonical (k+1)-mers of reads + rc(reads) (:62-80) */
    bucket_set kp1;
    if (count_from_reads(reads, k + 1, /*only_minimal=*/1, out->nbuckets, nthreads, 1, &kp1)) return -1;
    if (merge_buckets(&kp1, &out->kp1, &out->n_kp1, &out->kp1_count, NULL)) {
        bucket_set_free(&kp1);
        return -1;
    }
    bucket_set_free(&kp1);

    /* step 2: canonical k-mers of every (k+1)-mer and of its RC
     * (DeBruijnKMerKMerSplitter::FillBufferFromKMers, kmer_splitters.hpp:160-180, add_rc = true) */
    u64vec *kb = (u64vec *)calloc(out->nbuckets, sizeof(u64vec));
    if (!kb) return -1;
    int err = 0;
    for (size_t i = 0; i < out->n_kp1 && !err; ++i) {
        orc_kmer e, erc;
        memset(&e, 0, sizeof(e));
        memcpy(e.w, out->kp1 + i * nw1, (size_t)nw1 * sizeof(uint64_t));
        orc_kmer_rc(&e, k + 1, &erc);
        const orc_kmer *src[2] = {&e, &erc};
        for (int s = 0; s < 2; ++s) {
            for (int p = 0; p < 2; ++p) { /* the two k-mers of a (k+1)-mer */
                orc_kmer km;
                memset(&km, 0, sizeof(km));
                for (int j = 0; j < k; ++j) kmer_set(&km, j, orc_kmer_get(src[s], j + p));
                if (!orc_kmer_is_minimal(&km, k)) continue;
                uint64_t b = orc_bucket(km.w, nw, out->nbuckets);
                if (u64vec_push(&kb[b], km.w, nw)) err = 1;
            }
        }
    }
    out->bucket_start = (size_t *)calloc((size_t)out->nbuckets + 1, sizeof(size_t));
    if (!out->bucket_start) err = 1;
    if (!err) {
        bucket_set ks;
        ks.nb = out->nbuckets;
        ks.nw = nw;
        ks.b = kb;
        ks.cnt = NULL;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
        for (unsigned b = 0; b < out->nbuckets; ++b) {
            size_t m = sort_unique_records(kb[b].d, kb[b].n / (size_t)nw, nw, NULL);
            kb[b].n = m * (size_t)nw;
        }
        if (merge_buckets(&ks, &out->kmers, &out->n_k, NULL, out->bucket_start)) err = 1;
    }
    for (unsigned b = 0; b < out->nbuckets; ++b) free(kb[b].d);
    free(kb);
    if (err) {
        orc_extindex_free(out);
        return -1;
    }
    out->masks = (uint8_t *)calloc(out->n_k ? out->n_k : 1, 1);
    if (!out->masks) {
        orc_extindex_free(out);
        return -1;
    }

    /* step 3: FillExtensionsFromIndex (:44-59) + InOutMask::AddOutgoing/AddIncoming
     * (kmer_extension_index.hpp:67-69,92-106) */
    for (size_t i = 0; i < out->n_kp1; ++i) {
        orc_kmer e;
        memset(&e, 0, sizeof(e));
        memcpy(e.w, out->kp1 + i * nw1, (size_t)nw1 * sizeof(uint64_t));
        int pnucl = orc_kmer_get(&e, 0), nnucl = orc_kmer_get(&e, k);
        orc_kmer pre, suf;
        memset(&pre, 0, sizeof(pre));
        memset(&suf, 0, sizeof(suf));
        for (int j = 0; j < k; ++j) {
            kmer_set(&pre, j, orc_kmer_get(&e, j));
            kmer_set(&suf, j, orc_kmer_get(&e, j + 1));
        }
        kwh_t a, b;
        if (kwh_make(out, &pre, &a) || kwh_make(out, &suf, &b)) {
            orc_extindex_free(out);
            return -2;
        }
        out->masks[a.idx] |= (uint8_t)(1u << (a.minimal ? nnucl : 7 - nnucl));
        out->masks[b.idx] |= (uint8_t)(1u << (b.minimal ? pnucl + 4 : 7 - (pnucl + 4)));
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* unitigs                                                             */
/* ------------------------------------------------------------------ */
typedef struct {
    char *d;
    size_t n, cap;
} strbuf;

static int strbuf_push(strbuf *b, char c) {
    if (b->n + 1 > b->cap) {
        size_t nc = b->cap ? b->cap * 2 : 256;
        char *p = (char *)realloc(b->d, nc);
        if (!p) return -1;
        b->d = p;
        b->cap = nc;
    }
    b->d[b->n++] = c;
    return 0;
}

static const char kNt[4] = {'A', 'C', 'G', 'T'};

typedef struct {
    kwh_t start, end;
} deedge_t;

static int kwh_shl(const orc_extindex *x, const kwh_t *from, int c, kwh_t *to) {
    orc_kmer nk = from->key;
    orc_kmer_shl(&nk, x->k, c);
    return kwh_make(x, &nk, to);
}

static int deedge_eq(const deedge_t *a, const deedge_t *b, int nw) {
    return orc_kmer_cmp_words(a->start.key.w, b->start.key.w, nw) == 0 &&
           orc_kmer_cmp_words(a->end.key.w, b->end.key.w, nw) == 0;
}

/* ConstructSequenceWithEdge (debruijn_graph_constructor.hpp:232-241) */
static int construct_sequence(const orc_extindex *x, deedge_t edge, strbuf *sb) {
    int k = x->k, nw = orc_words(k);
    sb->n = 0;
    for (int i = 0; i < k; ++i)
        if (strbuf_push(sb, kNt[orc_kmer_get(&edge.start.key, i)])) return -1;
    if (strbuf_push(sb, kNt[orc_kmer_get(&edge.end.key, k - 1)])) return -1;
    deedge_t initial = edge;
    for (;;) {
        /* StepRightIfPossible (:222-230) */
        uint8_t m = kwh_mask(x, &edge.end);
        if (!(mask_unique_out(m) && mask_unique_in(m))) break;
        deedge_t nxt;
        nxt.start = edge.end;
        if (kwh_shl(x, &edge.end, kUniqueNext[m & 0xF], &nxt.end)) return -2;
        edge = nxt;
        if (deedge_eq(&edge, &initial, nw)) break;
        if (strbuf_push(sb, kNt[orc_kmer_get(&edge.end.key, k - 1)])) return -1;
    }
    return 0;
}

static void str_rc(const char *s, size_t n, char *out) {
    for (size_t i = 0; i < n; ++i) {
        char c = s[n - 1 - i];
        out[i] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A';
    }
    out[n] = 0;
}

/* Sequence::operator< (common/sequence/sequence.hpp:222-230), equal lengths */
static int str_less(const char *a, const char *b, size_t n) { return memcmp(a, b, n) < 0; }

static int unitigs_push(orc_unitigs *u, size_t *cap, const char *s, size_t n) {
    if (u->n == *cap) {
        size_t nc = *cap ? *cap * 2 : 1024;
        char **ps = (char **)realloc(u->seq, nc * sizeof(char *));
        if (!ps) return -1;
        u->seq = ps;
        size_t *pl = (size_t *)realloc(u->len, nc * sizeof(size_t));
        if (!pl) return -1;
        u->len = pl;
        *cap = nc;
    }
    char *c = (char *)malloc(n + 1);
    if (!c) return -1;
    memcpy(c, s, n);
    c[n] = 0;
    u->seq[u->n] = c;
    u->len[u->n] = n;
    u->n++;
    return 0;
}

/* CleanCondensed(sequence) (:288-296): IsolateVertex on every k-mer of the string */
static int clean_condensed(orc_extindex *x, const char *s, size_t n) {
    int k = x->k;
    if (n < (size_t)k) return 0;
    orc_kmer km;
    orc_kmer_from_ascii(s, k, &km);
    kwh_t h;
    if (kwh_make(x, &km, &h)) return -2;
    x->masks[h.idx] = 0;
    for (size_t pos = (size_t)k; pos < n; ++pos) {
        orc_kmer_shl(&km, k, orc_dignucl(s[pos]));
        if (kwh_make(x, &km, &h)) return -2;
        x->masks[h.idx] = 0;
    }
    return 0;
}

void orc_unitigs_free(orc_unitigs *u) {
    if (u->seq)
        for (size_t i = 0; i < u->n; ++i) free(u->seq[i]);
    free(u->seq);
    free(u->len);
    free(u->kc);
    memset(u, 0, sizeof(*u));
}

/* ---- early tip clipping on the extension index (assembly_graph/construction/early_simplification.hpp) ---- */

/* RemoveInconsistentForwardLinks (:20-35) */
static size_t remove_inconsistent_forward_links(orc_extindex *x, const kwh_t *kh) {
    size_t count = 0;
    uint8_t mask = kwh_mask(x, kh);
    for (int c = 0; c < 4; ++c) {
        if (!(mask & (1u << c))) continue;
        kwh_t next;
        if (kwh_shl(x, kh, c, &next)) continue;
        int first = orc_kmer_get(&kh->key, 0);
        if (!(kwh_mask(x, &next) & (1u << (4 + first)))) {
            /* DeleteOutgoing(kh, c): bit c of the oriented mask = bit (as_is ? c : 7 - c) of the stored byte
             * (kmer_extension_index.hpp:67-69,108-114) */
            x->masks[kh->idx] &= (uint8_t) ~(1u << (kh->minimal ? c : 7 - c));
            ++count;
        }
    }
    return count;
}

/* FindForward (:107-118): fills tip[] with the indices of the canonical forms; returns its size (0 = not a tip) */
static size_t find_forward(const orc_extindex *x, kwh_t kh, size_t length_bound, size_t *tip) {
    size_t n = 0;
    for (;;) {
        uint8_t m = kwh_mask(x, &kh);
        if (!(n < length_bound && mask_unique_in(m) && mask_unique_out(m))) break;
        tip[n++] = kh.idx;
        kwh_t nx;
        if (kwh_shl(x, &kh, kUniqueNext[m & 0xF], &nx)) return 0;
        kh = nx;
    }
    tip[n++] = kh.idx;
    uint8_t m = kwh_mask(x, &kh);
    if (!mask_unique_in(m) || (m & 0xF) != 0) return 0; /* branching or too long */
    return n;
}

/* EarlyTipClipperProcessor::ClipTips (:52-96) run by one thread over the k-mer file in order: every stored k-mer and
 * its reverse complement; a start with >= 2 outgoing edges loses every outgoing tip that is shorter than its longest
 * outgoing branch (non-tips count as infinitely long), then the phantom links of the tipped junctions are removed.
 * Returns the number of isolated k-mers; *clipped_links = links removed in the second phase. */
size_t orc_extindex_clip_tips(orc_extindex *x, size_t length_bound, size_t *clipped_links) {
    int k = x->k, nw = orc_words(k);
    size_t removed = 0, n_tipped = 0, cap_tipped = 1024;
    kwh_t *tipped = (kwh_t *)malloc(cap_tipped * sizeof(kwh_t));
    size_t *tips[4];
    for (int c = 0; c < 4; ++c) tips[c] = (size_t *)malloc((length_bound + 2) * sizeof(size_t));
    for (size_t i = 0; i < x->n_k; ++i) {
        for (int o = 0; o < 2; ++o) {
            kwh_t kh;
            memset(&kh, 0, sizeof(kh));
            memcpy(kh.key.w, x->kmers + i * nw, (size_t)nw * sizeof(uint64_t));
            if (o) {
                orc_kmer r;
                orc_kmer_rc(&kh.key, k, &r);
                kh.key = r;
            }
            if (kwh_make(x, &kh.key, &kh)) continue;
            uint8_t mask = kwh_mask(x, &kh);
            if (__builtin_popcount(mask & 0xF) < 2) continue;
            /* RemoveForward (:136-149) */
            size_t len[4] = {0, 0, 0, 0}, max = 0;
            for (int c = 0; c < 4; ++c) {
                if (!(mask & (1u << c))) continue;
                kwh_t khc;
                if (kwh_shl(x, &kh, c, &khc)) continue;
                len[c] = find_forward(x, khc, length_bound, tips[c]);
                size_t l = len[c] ? len[c] : (size_t)-1;
                if (l > max) max = l;
            }
            size_t rm = 0;
            for (int c = 0; c < 4; ++c) {
                if (len[c] && len[c] < max) {
                    for (size_t j = 0; j < len[c]; ++j) x->masks[tips[c][j]] = 0; /* IsolateVertex */
                    rm += len[c];
                }
            }
            removed += rm;
            if (rm) {
                if (n_tipped == cap_tipped) {
                    cap_tipped *= 2;
                    tipped = (kwh_t *)realloc(tipped, cap_tipped * sizeof(kwh_t));
                }
                tipped[n_tipped++] = kh;
            }
        }
    }
    size_t links = 0;
    for (size_t i = 0; i < n_tipped; ++i) links += remove_inconsistent_forward_links(x, &tipped[i]);
    if (clipped_links) *clipped_links = links;
    for (int c = 0; c < 4; ++c) free(tips[c]);
    free(tipped);
    return removed;
}

/* CalculateSequences (:267-286) over the canonical k-mers [lo, hi) of the merged file: what ONE chunk of
 * ExtractUnbranchingPaths' `#pragma omp parallel for` (:351-375) computes; the masks are only read */
static int extract_chunk(const orc_extindex *x, size_t lo, size_t hi, orc_unitigs *out) {
    int k = x->k, nw = orc_words(k);
    size_t cap = 0;
    strbuf sb = {0, 0, 0};
    char *rcbuf = NULL;
    size_t rccap = 0;
    int rc = 0;
    for (size_t i = lo; i < hi && !rc; ++i) {
        kwh_t kh;
        memset(&kh, 0, sizeof(kh));
        memcpy(kh.key.w, x->kmers + i * nw, (size_t)nw * sizeof(uint64_t));
        kh.idx = i;
        kh.minimal = 1;
        uint8_t ext = x->masks[i];
        if (!mask_is_junction(ext)) continue;
        /* AddStartDeEdges (:214-226) */
        deedge_t starts[8];
        int ns = 0;
        for (int next = 0; next < 4; ++next) {
            if (!(ext & (1u << next))) continue;
            starts[ns].start = kh;
            if (kwh_shl(x, &kh, next, &starts[ns].end)) { rc = -2; break; }
            ++ns;
        }
        if (rc) break;
        kwh_t inv;
        orc_kmer rk;
        orc_kmer_rc(&kh.key, k, &rk);
        if (kwh_make(x, &rk, &inv)) { rc = -2; break; }
        if (!inv.minimal) {
            uint8_t m2 = kwh_mask(x, &inv);
            for (int next = 0; next < 4; ++next) {
                if (!(m2 & (1u << next))) continue;
                starts[ns].start = inv;
                if (kwh_shl(x, &inv, next, &starts[ns].end)) { rc = -2; break; }
                ++ns;
            }
        }
        for (int e = 0; e < ns && !rc; ++e) {
            if ((rc = construct_sequence(x, starts[e], &sb))) break;
            if (sb.n + 1 > rccap) {
                rccap = sb.n * 2 + 16;
                char *p = (char *)realloc(rcbuf, rccap);
                if (!p) { rc = -1; break; }
                rcbuf = p;
            }
            str_rc(sb.d, sb.n, rcbuf);
            if (str_less(sb.d, rcbuf, sb.n)) continue; /* if (s < !s) continue; (:279-280) */
            if (unitigs_push(out, &cap, sb.d, sb.n)) rc = -1;
        }
    }
    free(sb.d);
    free(rcbuf);
    return rc;
}

int orc_unitigs_extract(orc_extindex *x, orc_unitigs *out) { return orc_unitigs_extract_mt(x, out, 1); }

/* nthreads > 1: the path phase runs over 16 * nthreads chunks of the k-mer file in parallel and the per-chunk results
 * are concatenated in chunk order, as ExtractUnbranchingPaths does (:351-375); CleanCondensed (:298-304) is a parallel
 * loop in the reference too.  The result does not depend on nthreads (the reference's does, through its chunk
 * boundaries being page-aligned file offsets: only the order of the paths, never the set). */
int orc_unitigs_extract_mt(orc_extindex *x, orc_unitigs *out, int nthreads) {
    memset(out, 0, sizeof(*out));
    int k = x->k, nw = orc_words(k);
    size_t cap = 0;
    strbuf sb = {0, 0, 0};
    char *rcbuf = NULL;
    size_t rccap = 0;
    int rc = 0;
    if (nthreads < 1) nthreads = 1;

    {
        const size_t nchunks = nthreads == 1 ? 1 : (size_t)16 * (size_t)nthreads;
        orc_unitigs *parts = (orc_unitigs *)calloc(nchunks, sizeof(orc_unitigs));
        if (!parts) return -1;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
        for (size_t c = 0; c < nchunks; ++c) {
            const size_t lo = x->n_k * c / nchunks, hi = x->n_k * (c + 1) / nchunks;
            const int r = extract_chunk(x, lo, hi, &parts[c]);
            if (r) {
#pragma omp atomic write
                rc = r;
            }
        }
        size_t total = 0;
        for (size_t c = 0; c < nchunks; ++c) total += parts[c].n;
        if (!rc && total) {
            out->seq = (char **)malloc(total * sizeof(char *));
            out->len = (size_t *)malloc(total * sizeof(size_t));
            if (!out->seq || !out->len) rc = -1;
        }
        for (size_t c = 0; c < nchunks; ++c) {
            if (!rc) {
                for (size_t j = 0; j < parts[c].n; ++j) {
                    out->seq[out->n] = parts[c].seq[j];
                    out->len[out->n] = parts[c].len[j];
                    out->n++;
                }
            } else {
                for (size_t j = 0; j < parts[c].n; ++j) free(parts[c].seq[j]);
            }
            free(parts[c].seq);
            free(parts[c].len);
        }
        free(parts);
        cap = out->n;
        if (rc) return rc;
    }
    size_t n_paths = out->n;

    /* CleanCondensed(result) (:298-304): a parallel loop in the reference as well; every write is "mask := 0" */
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 64)
    for (size_t i = 0; i < n_paths; ++i) {
        size_t n = out->len[i];
        char *rb = (char *)malloc(n + 1);
        int r = rb ? 0 : -1;
        if (!r) {
            str_rc(out->seq[i], n, rb);
            r = clean_condensed(x, out->seq[i], n);
            if (!r) r = clean_condensed(x, rb, n);
        }
        free(rb);
        if (r) {
#pragma omp atomic write
            rc = r;
        }
    }

    /* CollectLoops (:308-344) */
    size_t *starts_idx = NULL;
    size_t n_starts = 0;
    if (!rc) {
        starts_idx = (size_t *)malloc((x->n_k ? x->n_k : 1) * sizeof(size_t));
        if (!starts_idx) rc = -1;
    }
    if (!rc)
        for (size_t i = 0; i < x->n_k; ++i)
            if (!mask_is_junction(x->masks[i])) starts_idx[n_starts++] = i;
    for (size_t si = 0; si < n_starts && !rc; ++si) {
        size_t i = starts_idx[si];
        if (mask_is_junction(x->masks[i])) continue;
        /* ConstructLoopFromVertex (:255-265) */
        kwh_t kh;
        memset(&kh, 0, sizeof(kh));
        memcpy(kh.key.w, x->kmers + i * nw, (size_t)nw * sizeof(uint64_t));
        kh.idx = i;
        kh.minimal = 1;
        deedge_t bp;
        bp.start = kh;
        if (kwh_shl(x, &kh, kUniqueNext[x->masks[i] & 0xF], &bp.end)) { rc = -2; break; }
        if ((rc = construct_sequence(x, bp, &sb))) break;
        size_t n = sb.n;
        char *s = (char *)malloc(n + 1);
        if (!s) { rc = -1; break; }
        memcpy(s, sb.d, n);
        s[n] = 0;
        /* look for a (k+1)-mer equal to its own RC */
        size_t split_pos = (size_t)-1;
        {
            orc_kmer km, kr;
            for (size_t pos = 0; pos + (size_t)k + 1 <= n; ++pos) {
                orc_kmer_from_ascii(s + pos, k + 1, &km);
                orc_kmer_rc(&km, k + 1, &kr);
                if (orc_kmer_cmp_words(km.w, kr.w, orc_words(k + 1)) == 0) {
                    split_pos = pos;
                    break;
                }
            }
        }
        char *parts[2] = {NULL, NULL};
        size_t plen[2] = {0, 0};
        int np = 0;
        if (split_pos == (size_t)-1) {
            parts[0] = s;
            plen[0] = n;
            np = 1;
        } else {
            /* SplitLoop (:248-252) */
            size_t pos = split_pos;
            plen[0] = (size_t)k + 1;
            parts[0] = (char *)malloc(plen[0] + 1);
            plen[1] = (n - (size_t)k - (pos + 1)) + (pos + (size_t)k);
            parts[1] = (char *)malloc(plen[1] + 1);
            if (!parts[0] || !parts[1]) { rc = -1; free(parts[0]); free(parts[1]); free(s); break; }
            memcpy(parts[0], s + pos, plen[0]);
            parts[0][plen[0]] = 0;
            memcpy(parts[1], s + pos + 1, n - (size_t)k - (pos + 1));
            memcpy(parts[1] + (n - (size_t)k - (pos + 1)), s, pos + (size_t)k);
            parts[1][plen[1]] = 0;
            np = 2;
            free(s);
        }
        for (int p = 0; p < np && !rc; ++p) {
            if (plen[p] + 1 > rccap) {
                rccap = plen[p] * 2 + 16;
                char *q = (char *)realloc(rcbuf, rccap);
                if (!q) { rc = -1; break; }
                rcbuf = q;
            }
            str_rc(parts[p], plen[p], rcbuf);
            if (str_less(parts[p], rcbuf, plen[p])) {
                if (unitigs_push(out, &cap, rcbuf, plen[p])) rc = -1;
            } else {
                if (unitigs_push(out, &cap, parts[p], plen[p])) rc = -1;
            }
            if (!rc) rc = clean_condensed(x, parts[p], plen[p]);
            if (!rc) rc = clean_condensed(x, rcbuf, plen[p]);
        }
        for (int p = 0; p < np; ++p) free(parts[p]);
    }
    out->n_loops = out->n - n_paths;
    free(starts_idx);

    /* KC: sum of canonical (k+1)-mer multiplicities along the edge
     * (graph_support/coverage_filling.hpp:44-62) */
    if (!rc) {
        out->kc = (uint64_t *)calloc(out->n ? out->n : 1, sizeof(uint64_t));
        if (!out->kc) rc = -1;
    }
    if (!rc && x->kp1_count) {
        int nw1 = orc_words(k + 1);
        /* bucket starts of the (k+1)-mer file are recomputed by a linear scan */
        size_t *bs = (size_t *)calloc((size_t)x->nbuckets + 1, sizeof(size_t));
        if (!bs) rc = -1;
        if (!rc) {
            for (size_t i = 0; i < x->n_kp1; ++i) bs[orc_bucket(x->kp1 + i * nw1, nw1, x->nbuckets) + 1]++;
            for (unsigned b = 0; b < x->nbuckets; ++b) bs[b + 1] += bs[b];
            for (size_t u = 0; u < out->n; ++u) {
                const char *s = out->seq[u];
                for (size_t pos = 0; pos + (size_t)k + 1 <= out->len[u]; ++pos) {
                    orc_kmer km, kr;
                    orc_kmer_from_ascii(s + pos, k + 1, &km);
                    if (!orc_kmer_is_minimal(&km, k + 1)) {
                        orc_kmer_rc(&km, k + 1, &kr);
                        km = kr;
                    }
                    uint64_t b = orc_bucket(km.w, nw1, x->nbuckets);
                    size_t lo = bs[b], hi = bs[b + 1];
                    while (lo < hi) {
                        size_t mid = lo + (hi - lo) / 2;
                        int c = orc_kmer_cmp_words(x->kp1 + mid * nw1, km.w, nw1);
                        if (c == 0) { out->kc[u] += x->kp1_count[mid]; break; }
                        if (c < 0) lo = mid + 1;
                        else hi = mid;
                    }
                }
            }
        }
        free(bs);
    }
    free(sb.d);
    free(rcbuf);
    if (rc) orc_unitigs_free(out);
    return rc;
}

/* ------------------------------------------------------------------ */
/* graph ids + GFA                                                     */
/* ------------------------------------------------------------------ */
typedef struct {
    uint64_t hash_and_mask; /* (idx << 2) | is_rc << 1 | is_start  (LinkRecord, :400-430) */
    uint64_t edge;          /* edge id */
} link_rec;

static int link_cmp(const void *a, const void *b) {
    const link_rec *x = (const link_rec *)a, *y = (const link_rec *)b;
    if (x->hash_and_mask != y->hash_and_mask) return x->hash_and_mask < y->hash_and_mask ? -1 : 1;
    if (x->edge != y->edge) return x->edge < y->edge ? -1 : 1;
    return 0;
}

static void fmt_float(double v, char *buf, size_t n) {
    /* default ostream formatting of float(cov): %g with 6 significant digits (gfa_writer.cpp:23) */
    snprintf(buf, n, "%g", (double)(float)v);
}

int orc_gfa_write(const orc_extindex *x, const orc_unitigs *u, int with_cov, FILE *f,
                  size_t *n_vertices, size_t *n_links) {
    int k = x->k;
    const uint64_t ID_BIAS = 3; /* assembly_graph/core/graph_core.hpp:228 */
    size_t nrec = 0;
    link_rec *recs = (link_rec *)malloc((u->n ? 2 * u->n : 1) * sizeof(link_rec));
    char *selfconj = (char *)calloc(u->n ? u->n : 1, 1);
    char *rcbuf = NULL;
    size_t rccap = 0;
    if (!recs || !selfconj) {
        free(recs);
        free(selfconj);
        return -1;
    }
    /* CollectLinkRecords (:450-465): edge i -> id 3+2i, conjugate +1 unless self-conjugate
     * (graph_core.hpp:610-624) */
    for (size_t i = 0; i < u->n; ++i) {
        size_t n = u->len[i];
        if (n + 1 > rccap) {
            rccap = 2 * n + 16;
            rcbuf = (char *)realloc(rcbuf, rccap);
        }
        str_rc(u->seq[i], n, rcbuf);
        selfconj[i] = memcmp(u->seq[i], rcbuf, n) == 0;
        uint64_t eid = ID_BIAS + 2 * i;
        for (int is_start = 1; is_start >= 0; --is_start) {
            if (!is_start && selfconj[i]) continue; /* (:460-463) */
            orc_kmer km, kr;
            orc_kmer_from_ascii(is_start ? u->seq[i] : u->seq[i] + n - (size_t)k, k, &km);
            orc_kmer_rc(&km, k, &kr);
            int is_rc = !orc_kmer_less_nucl(&km, &kr, k); /* StartLink/EndLink (:432-448) */
            size_t idx = orc_extindex_find(x, is_rc ? &kr : &km);
            if (idx == (size_t)-1) {
                free(recs);
                free(selfconj);
                free(rcbuf);
                return -2;
            }
            recs[nrec].hash_and_mask = ((uint64_t)idx << 2) | ((uint64_t)is_rc << 1) | (uint64_t)is_start;
            recs[nrec].edge = eid;
            ++nrec;
        }
    }
    qsort(recs, nrec, sizeof(link_rec), link_cmp);

    /* WriteSegments (gfa_writer.cpp:18-25,35-41): canonical edges in id order */
    for (size_t i = 0; i < u->n; ++i) {
        uint64_t kc = with_cov && u->kc ? u->kc[i] : 0;
        double cov = with_cov ? (double)kc / (double)(u->len[i] - (size_t)k) : 0.0; /* coverage.hpp:58-64 */
        char fb[64];
        fmt_float(cov, fb, sizeof(fb));
        fprintf(f, "S\t%llu\t%s\tDP:f:%s\tKC:i:%llu\n", (unsigned long long)(ID_BIAS + 2 * i), u->seq[i], fb,
                (unsigned long long)kc);
    }
    /* vertices = distinct canonical end k-mers (:483-517); links = incoming x outgoing at each
     * canonical vertex (gfa_writer.cpp:43-52, construction_helper.hpp:80-90) */
    size_t nv = 0, nl = 0;
    for (size_t p = 0; p < nrec;) {
        size_t q = p;
        uint64_t h = recs[p].hash_and_mask >> 2;
        while (q < nrec && (recs[q].hash_and_mask >> 2) == h) ++q;
        ++nv;
        for (size_t a = p; a < q; ++a) {
            int a_start = (int)(recs[a].hash_and_mask & 1), a_rc = (int)((recs[a].hash_and_mask >> 1) & 1);
            /* incoming at v: end && !rc  -> (e,+) ; start && rc -> (conj e) */
            int a_in = (!a_start && !a_rc) || (a_start && a_rc);
            if (!a_in) continue;
            size_t ea = (size_t)((recs[a].edge - ID_BIAS) / 2);
            char oa = (!a_start || selfconj[ea]) ? '+' : '-';
            for (size_t b = p; b < q; ++b) {
                int b_start = (int)(recs[b].hash_and_mask & 1), b_rc = (int)((recs[b].hash_and_mask >> 1) & 1);
                /* outgoing at v: start && !rc -> (e,+) ; end && rc -> (conj e) */
                int b_out = (b_start && !b_rc) || (!b_start && b_rc);
                if (!b_out) continue;
                size_t eb = (size_t)((recs[b].edge - ID_BIAS) / 2);
                char ob = (b_start || selfconj[eb]) ? '+' : '-';
                fprintf(f, "L\t%llu\t%c\t%llu\t%c\t%dM\n", (unsigned long long)recs[a].edge, oa,
                        (unsigned long long)recs[b].edge, ob, k);
                ++nl;
            }
        }
        p = q;
    }
    if (n_vertices) *n_vertices = nv;
    if (n_links) *n_links = nl;
    free(recs);
    free(selfconj);
    free(rcbuf);
    return 0;
}

int orc_unitigs_fasta_write(const orc_unitigs *u, FILE *f) {
    /* projects/gbuilder/main.cpp:183-192; header_naming.hpp:14-20; osequencestream.hpp:22-28 */
    for (size_t i = 0; i < u->n; ++i) {
        fprintf(f, ">EDGE_%zu_length_%zu\n", i + 1, u->len[i]);
        for (size_t cur = 0; cur < u->len[i]; cur += 60) {
            size_t w = u->len[i] - cur < 60 ? u->len[i] - cur : 60;
            fwrite(u->seq[i] + cur, 1, w, f);
            fputc('\n', f);
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* FASTA/FASTQ(.gz) reader                                             */
/* ------------------------------------------------------------------ */
/* Serial restatement of the reference's parser: FastaFastqGzParser::operator>> over the vendored kseq on gzread
 * (common/io/reads/fasta_fastq_gz_parser.hpp:64-80,113-136; ext/include/kseq/kseq.h:66-76 ks_getc, :88-137
 * ks_getuntil2, :170-212 kseq_read).  One master thread parses in the reference too (read_processor.hpp:95-111).
 * Names, comments and qualities are dropped; bases are upper-cased (kseq.h:193-194). */
#include <zlib.h>

typedef struct {
    gzFile f;
    unsigned char buf[16384]; /* KSEQ_INIT bufsize, kseq.h:223 */
    int begin, end, is_eof;
} orc_ks;

static int orc_ks_getc(orc_ks *ks) { /* kseq.h:66-76 */
    if (ks->is_eof && ks->begin >= ks->end) return -1;
    if (ks->begin >= ks->end) {
        ks->begin = 0;
        ks->end = gzread(ks->f, ks->buf, sizeof(ks->buf));
        if (ks->end <= 0) {
            ks->end = 0;
            ks->is_eof = 1;
            return -1;
        }
    }
    return (int)ks->buf[ks->begin++];
}

typedef struct {
    char *s;
    size_t l, m;
} orc_str;

static void orc_str_reserve(orc_str *st, size_t need) {
    if (st->m < need) {
        st->m = need < 256 ? 256 : need * 2;
        st->s = (char *)realloc(st->s, st->m);
    }
}

/* one line appended to st (st == NULL: skipped); *dret = the delimiter seen ('\n') or 0; returns -1 when nothing
 * was read and the input is at its end (kseq.h:88-137, delimiter KS_SEP_LINE).  The trailing '\r' rule of
 * kseq.h:131 looks at the string being appended to, which for a sequence line is the record's sequence so far:
 * st->s + base. */
static long orc_ks_getline(orc_ks *ks, orc_str *st, int *dret, size_t base) {
    int gotany = 0;
    if (dret) *dret = 0;
    for (;;) {
        if (ks->begin >= ks->end) {
            if (ks->is_eof) break;
            ks->begin = 0;
            ks->end = gzread(ks->f, ks->buf, sizeof(ks->buf));
            if (ks->end <= 0) {
                ks->end = 0;
                ks->is_eof = 1;
                break;
            }
        }
        int i;
        for (i = ks->begin; i < ks->end; ++i)
            if (ks->buf[i] == '\n') break;
        gotany = 1;
        if (st) {
            orc_str_reserve(st, st->l + (size_t)(i - ks->begin) + 2);
            memcpy(st->s + st->l, ks->buf + ks->begin, (size_t)(i - ks->begin));
            st->l += (size_t)(i - ks->begin);
        }
        ks->begin = i + 1;
        if (i < ks->end) {
            if (dret) *dret = '\n';
            break;
        }
    }
    if (!gotany && ks->is_eof && ks->begin >= ks->end) return -1;
    if (st && st->l - base > 1 && st->s[st->l - 1] == '\r') --st->l;
    return st ? (long)(st->l - base) : 0;
}

int orc_fastx_read(const char *path, char **bases, uint64_t **offsets, size_t *n_out) {
    orc_ks *ks = (orc_ks *)calloc(1, sizeof(orc_ks));
    ks->f = gzopen(path, "r");
    if (!ks->f) {
        free(ks);
        return -1;
    }
    orc_str all = {NULL, 0, 0}, qual = {NULL, 0, 0};
    size_t n = 0, cap = 1024;
    uint64_t *off = (uint64_t *)malloc((cap + 1) * sizeof(uint64_t));
    off[0] = 0;
    int last_char = 0, c;
    for (;;) { /* kseq_read, kseq.h:170-212 */
        if (last_char == 0) { /* jump to the next header line */
            while ((c = orc_ks_getc(ks)) != -1 && c != '>' && c != '@') {}
            if (c == -1) break;
            last_char = c;
        }
        if (orc_ks_getline(ks, NULL, &c, 0) < 0) break; /* name: the whole header line (KS_SEP_LINE, kseq.h:182) */
        const size_t start = all.l;                     /* the record's sequence is the tail of `all` */
        while ((c = orc_ks_getc(ks)) != -1 && c != '>' && c != '+' && c != '@') {
            if (c == '\n') continue; /* skip empty lines */
            orc_str_reserve(&all, all.l + 2);
            all.s[all.l++] = (char)c;
            orc_ks_getline(ks, &all, NULL, start); /* rest of the line */
        }
        if (c == '>' || c == '@') last_char = c; /* the first header char has been read */
        else last_char = 0;
        for (size_t j = start; j < all.l; ++j) /* toupper, kseq.h:193-194 */
            if (all.s[j] >= 'a' && all.s[j] <= 'z') all.s[j] = (char)(all.s[j] - 32);
        if (c == '+') {
            while ((c = orc_ks_getc(ks)) != -1 && c != '\n') {} /* skip the rest of the '+' line */
            if (c == -1) {                                       /* -2: no quality string */
                all.l = start;
                break;
            }
            const size_t seq_l = all.l - start;
            qual.l = 0;
            while (orc_ks_getline(ks, &qual, NULL, 0) >= 0 && qual.l < seq_l) {}
            last_char = 0;
            if (qual.l != seq_l) { /* -2: fasta_fastq_gz_parser.hpp:130-136 treats it as the end of the stream */
                all.l = start;
                break;
            }
        }
        if (n == cap) {
            cap *= 2;
            off = (uint64_t *)realloc(off, (cap + 1) * sizeof(uint64_t));
        }
        off[++n] = all.l;
    }
    gzclose(ks->f);
    free(ks);
    free(qual.s);
    if (!all.s) all.s = (char *)calloc(1, 1);
    *bases = all.s;
    *offsets = off;
    *n_out = n;
    return 0;
}
