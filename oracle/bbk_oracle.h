/*
 * bbk_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C restatement of the reference's k-mer counting / de Bruijn graph
 * construction path (SPAdes 3.15.4, `spades-kmercount` and `spades-gbuilder`).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this; the product path (spades_for_blackbird_amd/csrc) never does.
 *
 * Parity pin: this restatement is checked against
 *   - the reference's own known-answer tests
 *     (assembler/src/test/debruijn/construction_test.cpp:32-66,99-107 and
 *      assembler/src/test/include_test/rtseq_test.cpp), restated as data in
 *     tests/golden (JSON files),
 *   - outputs of the reference binaries recorded in SURVEY.md section 8(c)
 *     (md5 of final_kmers for assembler/test_dataset, unitig/vertex/link
 *     counts, KC values, loop / self-RC goldens),
 *   - the vendored xxHash 0.8.0 header compiled where it lies
 *     (oracle/Makefile -> oracle/_ref/libxxh3_ref.so) and python-xxhash.
 * The reference's C++ path itself is unbuildable here under the round rules
 * (k_range.hpp/config.hpp are cmake-generated, bamtools needs bzlib.h); see
 * DESIGN.md.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/assembler/src unless noted).
 */
#ifndef BBK_ORACLE_H_
#define BBK_ORACLE_H_

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_WORDS 4 /* MAX_K = 128 -> 4 x uint64 (common/sequence/seq_common.hpp:34) */

typedef struct {
    uint64_t w[ORC_MAX_WORDS];
} orc_kmer;

/* ---- sequence primitives (common/sequence/rtseq.hpp, nucl.hpp) ---- */
int orc_words(int k);                                   /* rtseq.hpp:129-131 GetDataSize */
int orc_is_nucl(char c);                                /* nucl.hpp:45-62 */
int orc_dignucl(char c);                                /* nucl.hpp:120-130 */
int orc_kmer_get(const orc_kmer *x, int i);             /* rtseq.hpp operator[] */
void orc_kmer_from_ascii(const char *s, int k, orc_kmer *out); /* rtseq.hpp:289-330 */
void orc_kmer_to_ascii(const orc_kmer *x, int k, char *out);   /* rtseq.hpp:633-639 */
void orc_kmer_shl(orc_kmer *x, int k, int c);           /* rtseq.hpp:450-467 operator<<= */
void orc_kmer_rc(const orc_kmer *x, int k, orc_kmer *out); /* rtseq.hpp:79-115,387-400 */
int orc_kmer_is_minimal(const orc_kmer *x, int k);      /* rtseq.hpp:407-415 */
int orc_kmer_less_nucl(const orc_kmer *a, const orc_kmer *b, int k); /* rtseq.hpp:732-741 operator< */
int orc_kmer_cmp_words(const uint64_t *a, const uint64_t *b, int nw); /* adt/array_vector.hpp:114-123 */

/* ---- Sequence (common/sequence/sequence.hpp): a view (from, size, rtl) over shared 2-bit storage; the unitig
 *      orientation rule `if (s < !s) continue` (debruijn_graph_constructor.hpp:279) is stated in its operators.
 *      `data` stands for the storage: ACGT text of the underlying forward sequence. ---- */
typedef struct {
    const char *data;
    size_t from, size;
    int rtl; /* read right-to-left, complemented */
} orc_seq;
orc_seq orc_seq_make(const char *acgt);                         /* sequence.hpp:71-122 */
int orc_seq_at(const orc_seq *s, size_t i);                     /* operator[] :311-319: rtl -> complement(data[from+size-1-i]) */
orc_seq orc_seq_rc(const orc_seq *s);                           /* operator! :232-234: same storage, rtl flipped */
orc_seq orc_seq_subseq(const orc_seq *s, size_t from, size_t to); /* Subseq :324-333 */
int orc_seq_less(const orc_seq *a, const orc_seq *b);           /* operator< :222-230: base-lexicographic, shorter prefix first */
int orc_seq_eq(const orc_seq *a, const orc_seq *b);             /* operator== */
void orc_seq_str(const orc_seq *s, char *out);                  /* str() :377-383; out holds size + 1 bytes */
void orc_seq_concat(const orc_seq *a, const orc_seq *b, char *out); /* operator+ :361-362: Sequence(str() + s.str()) */
int orc_complement(int code);                                   /* nucl.hpp:27-34 */
char orc_nucl(int code);                                        /* nucl.hpp:105-112 */

/* ---- bucket policy ---- */
uint64_t orc_xxh3_64(const uint64_t *words, int nwords); /* ext/include/xxh/xxhash.h:2781-2822,2850-2908, seed 0 */
uint64_t orc_mulhi64(uint64_t x, uint64_t y);            /* adt/lemiere_mod_reduce.hpp:17-35 */
uint64_t orc_bucket(const uint64_t *words, int nwords, uint64_t nbuckets); /* utils/kmer_mph/kmer_buckets.hpp:28-33 */

/* ---- read normalisation ---- */
/* io/reads/longest_valid_wrapper.hpp:15-41 ; returns [from,to) */
void orc_longest_valid(const char *s, size_t len, size_t *from, size_t *to);

/* A batch of reads as one ASCII blob + offsets (n+1 entries). */
typedef struct {
    const char *bases;
    const uint64_t *offsets;
    size_t n;
} orc_reads;

/* FASTA/FASTQ(.gz) file -> reads (serial, kseq semantics: io/reads/fasta_fastq_gz_parser.hpp:64-80,113-136,
 * ext/include/kseq/kseq.h:170-212).  *bases and *offsets (n+1 entries) are malloc'ed. */
int orc_fastx_read(const char *path, char **bases, uint64_t **offsets, size_t *n_out);

/* ---- spades-kmercount (projects/kmercount/main.cpp:64-82,95-120,214-219) ---- */
/*
 * All k-mers of the normalised reads and of their reverse complements, split
 * into `nbuckets` XXH3 buckets, each sorted (word order) + uniqued, buckets
 * concatenated: the exact content of <workdir>/final_kmers.
 * *out is malloc'ed: (*n_out) records of orc_words(k) uint64 each.
 * If counts != NULL, *counts receives the multiplicity of each record.
 */
int orc_kmercount(const orc_reads *reads, int k, unsigned nbuckets, int nthreads,
                  uint64_t **out, size_t *n_out, uint32_t **counts);

/* ---- extension index (utils/extension_index/) ---- */
typedef struct {
    int k;
    unsigned nbuckets;      /* 10 * T (kmer_extension_index_builder.hpp:73) */
    /* distinct canonical (k+1)-mers in merged-file order */
    size_t n_kp1;
    uint64_t *kp1;          /* orc_words(k+1) words each */
    uint32_t *kp1_count;    /* occurrences over reads + rc(reads) (coverage_hash_map_builder.hpp:15-38) */
    /* distinct canonical k-mers in merged-file order, + InOutMask byte each */
    size_t n_k;
    uint64_t *kmers;        /* orc_words(k) words each */
    uint8_t *masks;         /* kmer_extension_index.hpp:42-196 */
    size_t *bucket_start;   /* nbuckets+1 */
} orc_extindex;

int orc_extindex_build(const orc_reads *reads, int k, unsigned T, orc_extindex *out);
void orc_extindex_free(orc_extindex *x);
/* index of a canonical k-mer in merged-file order, or (size_t)-1 */
size_t orc_extindex_find(const orc_extindex *x, const orc_kmer *canon);

/* ---- early tip clipping (assembly_graph/construction/early_simplification.hpp:37-160) ---- */
/* EarlyTipClipperProcessor(index, length_bound).ClipTips(): modifies the masks in place; returns the number of
 * isolated k-mers ("<n> (k+1)-mers were removed by early tip clipper"), *clipped_links the phantom links removed */
size_t orc_extindex_clip_tips(orc_extindex *x, size_t length_bound, size_t *clipped_links);

/* ---- unitigs (assembly_graph/construction/debruijn_graph_constructor.hpp:182-388) ---- */
typedef struct {
    size_t n;          /* paths + loops, reference order (loops last) */
    size_t n_loops;
    char **seq;        /* ACGT strings */
    size_t *len;
    uint64_t *kc;      /* sum of (k+1)-mer multiplicities (coverage_filling.hpp:44-62) */
} orc_unitigs;

/* Destroys the masks of x (CleanCondensed), as the reference does. */
int orc_unitigs_extract(orc_extindex *x, orc_unitigs *out);
int orc_unitigs_extract_mt(orc_extindex *x, orc_unitigs *out, int nthreads); /* :351-375: chunks in parallel */
void orc_unitigs_free(orc_unitigs *u);

/* ---- graph ids + GFA text (debruijn_graph_constructor.hpp:390-518, io/graph/gfa_writer.cpp:18-52) ---- */
/* with_cov != 0 mirrors `-c`. L lines are emitted in (vertex k-mer file index) order. */
int orc_gfa_write(const orc_extindex *x, const orc_unitigs *u, int with_cov, FILE *f,
                  size_t *n_vertices, size_t *n_links);
/* `--unitigs` FASTA (projects/gbuilder/main.cpp:183-192) */
int orc_unitigs_fasta_write(const orc_unitigs *u, FILE *f);

#ifdef __cplusplus
}
#endif
#endif
