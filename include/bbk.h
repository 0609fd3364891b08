/*
 * bbk.h -- C ABI of the MI355X k-mer counting / de Bruijn graph construction engine.
 *
 * The reference (SPAdes 3.15.4 fork, /root/reference/assembler/src) has no FFI layer; its
 * boundaries for this path are two argv contracts, a C++ operator API and two file formats
 * (SURVEY.md 8b).  Each entry point below names the reference interface it replaces
 * (paths relative to /root/reference/assembler/src).  All functions return 0 on success or a
 * negative bbk_status; the message is available from bbk_last_error() (thread local).
 * No exceptions cross the ABI.  Handles are opaque.  One context per GPU per host thread;
 * a context is not re-entrant (same contract as KMerSortingSplitter, whose per-thread
 * buffers make Split() one-call-at-a-time: common/utils/kmer_mph/kmer_splitter.hpp:111-118).
 *
 * Pointers named d_* must be device (HBM) pointers, h_* host pointers; `dst` pointers of the
 * export calls may be either (hipMemcpyDefault).
 */
#ifndef BBK_H_
#define BBK_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bbk_ctx bbk_ctx;
typedef struct bbk_reads bbk_reads;       /* 2-bit packed reads resident in HBM                     */
typedef struct bbk_kmerset bbk_kmerset;   /* sorted distinct k-mers (+ multiplicities) in HBM       */
typedef struct bbk_extindex bbk_extindex; /* sorted canonical k-mers + InOutMask byte each, in HBM  */
typedef struct bbk_unitigs bbk_unitigs;   /* condensed edges + link records (host + device)         */

enum bbk_status {
    BBK_OK = 0,
    BBK_ERR_ARG = -1,      /* bad argument (k out of [1,128), even k for the graph, ...)            */
    BBK_ERR_HIP = -2,      /* a HIP runtime call failed (FATAL_ERROR analogue, utils/logger/logger.hpp:177-190) */
    BBK_ERR_NOMEM = -3,
    BBK_ERR_INTERNAL = -4, /* a device-side invariant failed (VERIFY analogue, utils/verify.hpp)      */
    BBK_ERR_IO = -5
};

#define BBK_MAX_K 128 /* cmake/options.cmake:55-56 SPADES_MAX_K; k must be < BBK_MAX_K */

/* ---- context ---------------------------------------------------------------------------- */
const char *bbk_last_error(void);
const char *bbk_version(void);
int bbk_ctx_create(int device, bbk_ctx **out);
int bbk_ctx_destroy(bbk_ctx *ctx);
/* Run all kernels of this context on an existing hipStream_t (e.g. torch's current stream). */
int bbk_ctx_set_stream(bbk_ctx *ctx, void *hip_stream);
int bbk_ctx_synchronize(bbk_ctx *ctx);
/* Returns device memory the allocator holds but does not use to the driver -- where that is safe: the block-cache
 * allocator (BBK_NO_VMM=1) frees its cached blocks; the default arena allocator keeps its mapped high-water mark for the
 * life of the process (unmapping and re-mapping chunks faults on this platform, see primitives.hip). */
int bbk_ctx_trim(bbk_ctx *ctx);
/* Accumulated HIP-event time (ms) and launch count of one named kernel family since the last
 * reset ("extract", "hist", "scan", "scatter", "unique", "expand", "mask", "walk", ...).
 * Timing is only recorded while profiling is enabled (it serialises nothing: events are
 * recorded on the context's stream around each launch). */
int bbk_ctx_profile_enable(bbk_ctx *ctx, int on);
int bbk_ctx_profile_reset(bbk_ctx *ctx);
int bbk_ctx_profile_get(bbk_ctx *ctx, const char *family, double *ms_total, uint64_t *launches,
                        double *bytes_total);
/* Event counters are read the same way (launches = events, bytes_total = summed value): "stat_slot_records",
 * "stat_slot_spilled", "stat_slot_overflow_segments", "stat_slot_overflow_buckets", "stat_slot_reprocessed" -- what the
 * histogram-free slot mode of stage A placed, spilled and had to reprocess (skewed inputs). */

/* ---- reads: replaces io::EasyStream(file, followed_by_rc=true, handle_Ns=true)
 *      (common/io/reads/io_helper.cpp:19-32) and the binary read cache
 *      (common/io/reads/binary_converter.cpp:50-113) --------------------------------------- */
/* ASCII reads (concatenated; offsets has n+1 entries).  Applies the LongestValid rule
 * (common/io/reads/longest_valid_wrapper.hpp:15-52), accepts ACGTacgt (common/sequence/nucl.hpp:45-62),
 * packs 2 bits/base (A=0 C=1 G=2 T=3, base i in bits 2(i%32) of word i/32, every read starts on
 * a 64-bit word) and uploads.  Reverse complements are NOT materialised: kernels canonicalise. */
int bbk_reads_from_ascii(bbk_ctx *ctx, const char *h_bases, const uint64_t *h_offsets, uint64_t n_reads,
                         bbk_reads **out);
/* Packed reads in HOST memory (what a parser thread of the host produces): read i occupies ceil(len[i]/32) u64
 * words, the reads follow each other in order, each starting on a 64-bit word (the layout above); bits above a
 * read's last base must be zero.  n_words must equal the sum of the per-read word counts.  Uploads (fastest from
 * bbk_host_alloc memory); the word offsets are computed on the device.  The LongestValid rule has already been
 * applied by the caller (one run of ACGTacgt per record, io/reads/longest_valid_wrapper.hpp:15-52). */
int bbk_reads_from_packed(bbk_ctx *ctx, const uint64_t *h_words, uint64_t n_words, const uint32_t *h_len,
                          uint64_t n_reads, bbk_reads **out);
/* Page-locked host memory for the buffers above (host -> device copies from pageable memory run at a fraction of
 * the link rate). */
int bbk_host_alloc(size_t bytes, void **out);
void bbk_host_free(void *p);
/* Adopt packed reads already in HBM (not copied, not freed): d_words[u64], d_word_off[u64, n+1]
 * (first word of each read), d_len[u32, n] (bases). */
int bbk_reads_from_device(bbk_ctx *ctx, const void *d_words, const void *d_word_off, const void *d_len,
                          uint64_t n_reads, uint64_t n_words, bbk_reads **out);
/* Synthetic reads generated on the device: uniform genome of genome_len bases
 * (seed_genome), uniform start, strand flip p=0.5, substitution rate sub_rate (seed_reads). */
int bbk_reads_synth(bbk_ctx *ctx, uint64_t n_reads, uint32_t read_len, uint64_t genome_len, double sub_rate,
                    uint64_t seed_genome, uint64_t seed_reads, bbk_reads **out);
/* Metagenome-shaped synthetic reads (BASELINE configs[4]): n_genomes random genomes with lengths
 * log-uniform in [min_len, max_len] and abundances log-normal(sigma) (sigma = 2: three decades of coverage skew);
 * a read picks its genome with probability ~ abundance x length; read model as bbk_reads_synth.  h_genome_len /
 * h_abundance (optional, n_genomes entries) receive the drawn community. */
int bbk_reads_synth_meta(bbk_ctx *ctx, uint64_t n_reads, uint32_t read_len, uint32_t n_genomes, uint64_t min_len,
                         uint64_t max_len, double sigma, double sub_rate, uint64_t seed, bbk_reads **out,
                         uint64_t *h_genome_len, double *h_abundance);
/* SPAdes binary read cache of single reads (<prefix>.seq / <prefix>.off, io::BinaryWriter::ToBinary,
 * common/io/reads/binary_converter.cpp:50-113; record layout Sequence::BinWrite, common/sequence/sequence.hpp:410-442).
 * Its 2-bit words are exactly the device layout: records are copied, not re-encoded. */
int bbk_reads_from_spades_binary(bbk_ctx *ctx, const char *seq_path, bbk_reads **out);
int bbk_reads_write_spades_binary(bbk_ctx *ctx, const bbk_reads *r, const char *prefix);
uint64_t bbk_reads_count(const bbk_reads *r);
uint64_t bbk_reads_bases(const bbk_reads *r);
/* Copy read i back as ASCII (tests); dst must hold len+1 bytes; returns the length via *len. */
int bbk_reads_get_ascii(bbk_ctx *ctx, const bbk_reads *r, uint64_t i, char *h_dst, uint32_t cap, uint32_t *len);
/* All reads back as ASCII (h_bases may be NULL to query sizes: h_offsets[n] = total bases). */
int bbk_reads_export_ascii(bbk_ctx *ctx, const bbk_reads *r, char *h_bases, uint64_t *h_offsets, uint64_t cap_bases);
void bbk_reads_free(bbk_reads *r);

/* ---- k-mer counting: replaces kmers::KMerDiskCounter<RtSeq>::Count / CountAll
 *      (common/utils/kmer_mph/kmer_index_builder.hpp:195-217,241-279) over a
 *      KMerSortingSplitter (kmer_splitter.hpp:24-52,73-167) ------------------------------- */
#define BBK_BOTH_STRANDS 1u /* spades-kmercount semantics: k-mers of reads and of rc(reads)
                               (projects/kmercount/main.cpp:64-82,106) */
#define BBK_CANONICAL 2u    /* only IsMinimal k-mers (utils/ph_map/storing_traits.hpp:90-101), as the
                               gbuilder splitters use (kmer_splitters.hpp:25-41) */
#define BBK_WITH_COUNTS 4u  /* keep multiplicities (occurrences over reads + rc(reads)) */
#define BBK_UNSORTED 8u     /* distinct set only, internal (hash-bucket) order: enough for the owner partition,
                               bbk_kmerset_both_strands and a later bbk_kmerset_from_device; skips the sort */
#define BBK_REFERENCE_ORDER 16u /* store the set in the final_kmers order (BBK_ORDER_REFERENCE_BUCKETS16) instead of
                                  ascending: what CountAll(16, ..., merge=true) leaves on disk
                                  (projects/kmercount/main.cpp:214-219).  Exporting in that order is then a plain copy
                                  and bbk_kmerset_keys() is the result itself */
#define BBK_WITH_MASKS 32u /* with BBK_CANONICAL: the payload of every k-mer is the OR of the InOutMask bits of its
                              occurrences (the records of the extension index before they are ordered) instead of a
                              multiplicity; bbk_kmerset_export* hand it out where they hand out counts.  The multi-GPU
                              path exchanges these records by owner and builds each shard of the index from them
                              (bbk_extindex_from_device). */
int bbk_count(bbk_ctx *ctx, const bbk_reads *reads, unsigned k, unsigned flags, bbk_kmerset **out);
/* Streaming count: the input never has to be resident as a whole.  Replaces the bounded-memory contract of
 * KMerSortingSplitter -- per-thread cells of `-b` bytes (PrepareBuffers, common/utils/kmer_mph/kmer_splitter.hpp:73-109),
 * one sorted + uniqued run per bucket every time the cells fill up (DumpBuffers, :120-167), and the loser-tree merge of
 * the runs at the end (KMerDiskCounter::MergeKMers, kmer_index_builder.hpp:281-365).  Every pushed batch is
 * deduplicated on the device and kept as a run of distinct canonical k-mers; runs are merge-uniqued into the
 * accumulated set as they pile up; bbk_count_finish orders the set as `flags` ask (same flags and same result as one
 * bbk_count over the concatenated batches).  Device memory is bounded by the batch and the DISTINCT set, host memory
 * by the batch.  bbk_count_finish releases the counter (also on failure); bbk_count_abort drops it without a result. */
typedef struct bbk_counter bbk_counter;
int bbk_count_begin(bbk_ctx *ctx, unsigned k, unsigned flags, bbk_counter **out);
int bbk_count_push_reads(bbk_counter *c, const bbk_reads *reads);
/* ASCII batch (same arguments and LongestValid rule as bbk_reads_from_ascii) */
int bbk_count_push_ascii(bbk_counter *c, const char *h_bases, const uint64_t *h_offsets, uint64_t n_reads);
int bbk_count_finish(bbk_counter *c, bbk_kmerset **out);
void bbk_count_abort(bbk_counter *c);
uint64_t bbk_count_pushed_instances(const bbk_counter *c); /* k-mer positions seen so far */
/* Device pointer to the records of the set (size * words u64) in the order it is stored in; valid until
 * bbk_kmerset_free.  *order receives BBK_ORDER_SORTED / BBK_ORDER_REFERENCE_BUCKETS16, or 0xFFFFFFFF for a
 * BBK_UNSORTED set. */
const void *bbk_kmerset_keys(const bbk_kmerset *s, unsigned *order);
/* Sort + unique an array of k-mer records already in HBM (n records of bbk_words(k) u64 each,
 * optional u32 multiplicities that are summed).  Used after the multi-GPU exchange. */
int bbk_kmerset_from_device(bbk_ctx *ctx, const void *d_keys, const void *d_counts, uint64_t n, unsigned k,
                            bbk_kmerset **out);
/* same with flags: BBK_UNSORTED deduplicates only (hash-bucket order) */
int bbk_kmerset_from_device_ex(bbk_ctx *ctx, const void *d_keys, const void *d_counts, uint64_t n, unsigned k,
                               unsigned flags, bbk_kmerset **out);
/* canon U rc(canon): the both-strand set of spades-kmercount from a BBK_CANONICAL set (each rank
 * applies it to its own shard after the multi-GPU exchange). */
int bbk_kmerset_both_strands(bbk_ctx *ctx, const bbk_kmerset *canon, bbk_kmerset **out);
/* same; flags = BBK_REFERENCE_ORDER stores the result in the final_kmers order */
int bbk_kmerset_both_strands_ex(bbk_ctx *ctx, const bbk_kmerset *canon, unsigned flags, bbk_kmerset **out);
unsigned bbk_words(unsigned k); /* RtSeq::GetDataSize (common/sequence/rtseq.hpp:129-131) */
uint64_t bbk_kmerset_size(const bbk_kmerset *s);
unsigned bbk_kmerset_k(const bbk_kmerset *s);
uint64_t bbk_kmerset_instances(const bbk_kmerset *s); /* k-mer instances that entered the sort */
#define BBK_ORDER_SORTED 0u              /* ascending, word 0 most significant (adt/array_vector.hpp:114-123) */
#define BBK_ORDER_REFERENCE_BUCKETS16 1u /* the final_kmers order: XXH3 bucket (16) major, ascending inside
                                            (kmer_buckets.hpp:28-33, kmer_index_builder.hpp:168-181) */
/* dst_keys: size * words u64 (host or device); dst_counts (u32, may be NULL). */
int bbk_kmerset_export(bbk_ctx *ctx, const bbk_kmerset *s, unsigned order, void *dst_keys, void *dst_counts);
/* Multi-GPU owner partition: owner(key) = mulhi(mix(key), nranks).  Writes the
 * records grouped by owner to dst (host or device) and the per-owner record counts to h_counts. */
int bbk_kmerset_export_by_owner(bbk_ctx *ctx, const bbk_kmerset *s, unsigned nranks, void *dst_keys,
                                void *dst_counts, uint64_t *h_counts);
/* The device side of spades-read-filter: io::CoverageFilter / CountMedianMlt
 * (common/io/reads/coverage_filtering_read_wrapper.hpp:22-76, projects/kmercount/read_filter.cpp:76-121): h_keep[i] = 1
 * when the upper median of the multiplicities of read i's k-mers (strands identified) is >= threshold (the tool passes
 * its -c value + 1).  `counts` must be the ascending canonical set with counts of the whole dataset
 * (bbk_count(BBK_CANONICAL | BBK_WITH_COUNTS)).  Reads shorter than k have median 0.  For a pair the tool keeps both
 * mates when either passes (coverage_filtering_read_wrapper.hpp:78-94). */
int bbk_reads_median_filter(bbk_ctx *ctx, const bbk_reads *reads, const bbk_kmerset *counts, unsigned threshold,
                            uint8_t *h_keep, uint64_t *n_kept);
void bbk_kmerset_free(bbk_kmerset *s);
/* VERIFY(std::is_sorted(run)) analogue (KMerDiskCounter::MergeKMers, common/utils/kmer_mph/kmer_index_builder.hpp:297),
 * on the device, for sets too large to download: *n_runs = maximal ascending runs of the stored order (1 for an
 * ascending set, <= 16 for a set in the final_kmers order), *n_equal = equal neighbours (0 for a distinct set);
 * h_run_starts (optional, cap entries): record index where each of the first runs begins. */
int bbk_kmerset_verify_order(bbk_ctx *ctx, const bbk_kmerset *s, uint64_t *n_runs, uint64_t *n_equal,
                             uint64_t *h_run_starts, unsigned cap);
/* records [first, first + count) of the stored order (and their multiplicities, if kept) to host memory */
int bbk_kmerset_get(bbk_ctx *ctx, const bbk_kmerset *s, uint64_t first, uint64_t count, void *h_keys, void *h_counts);
/* Writes <path> in the final_kmers format (raw little-endian records, no header). */
int bbk_kmerset_write_final_kmers(bbk_ctx *ctx, const bbk_kmerset *s, const char *path);

/* ---- extension index: replaces DeBruijnExtensionIndexBuilder::BuildExtensionIndexFromStream
 *      (common/utils/extension_index/kmer_extension_index_builder.hpp:62-106) -------------- */
int bbk_extindex_build(bbk_ctx *ctx, const bbk_reads *reads, unsigned k, bbk_extindex **out);
/* The index of a record array already in HBM: n canonical k-mers (bbk_words(k) u64 each) with u32 mask payloads, in
 * any order, duplicates allowed (their masks are OR-ed).  What a rank builds its shard from after the owner exchange
 * of BBK_WITH_MASKS records, and what the gathered shards are merged with before the unitig stage. */
int bbk_extindex_from_device(bbk_ctx *ctx, const void *d_keys, const void *d_masks_u32, uint64_t n, unsigned k,
                             bbk_extindex **out);
/* masks as u32 (the payload layout of the exchange), dst may be host or device */
int bbk_extindex_export_u32(bbk_ctx *ctx, const bbk_extindex *x, void *dst_keys, void *dst_masks_u32);
/* Streaming build (same contract as bbk_count_begin / push / finish): (canonical k-mer, mask bits) records of every
 * pushed batch are OR-reduced on the device, merged as they pile up, ordered once by bbk_extindex_finish. */
typedef struct bbk_extbuilder bbk_extbuilder;
int bbk_extindex_begin(bbk_ctx *ctx, unsigned k, bbk_extbuilder **out);
int bbk_extindex_push_reads(bbk_extbuilder *b, const bbk_reads *reads);
int bbk_extindex_finish(bbk_extbuilder *b, bbk_extindex **out);
void bbk_extindex_abort(bbk_extbuilder *b);
/* Count + extension index from ONE pass over the reads (BASELINE configs[2]: both are wanted of the same reads): the
 * canonical records the index is built from are also the canonical set of the count -- including the k-mers of reads of
 * length exactly k, which the index drops (they never get an extension bit) and spades-kmercount keeps.
 * set_flags = BBK_BOTH_STRANDS [| BBK_REFERENCE_ORDER].  Results equal bbk_count / bbk_extindex_build run separately. */
int bbk_count_extindex(bbk_ctx *ctx, const bbk_reads *reads, unsigned k, unsigned set_flags, bbk_kmerset **set,
                       bbk_extindex **out);
int bbk_extindex_finish_with_set(bbk_extbuilder *b, unsigned set_flags, bbk_kmerset **set, bbk_extindex **out);
uint64_t bbk_extindex_size(const bbk_extindex *x);
unsigned bbk_extindex_k(const bbk_extindex *x);
/* sorted canonical k-mers (size*words u64) and their InOutMask bytes
 * (kmer_extension_index.hpp:42-196: bits 0-3 outgoing A,C,G,T; bits 4-7 incoming) */
int bbk_extindex_export(bbk_ctx *ctx, const bbk_extindex *x, void *dst_keys, void *dst_masks);
/* Early tip clipping on the index, in place: EarlyTipClipperProcessor(index, length_bound).ClipTips()
 * (common/assembly_graph/construction/early_simplification.hpp:37-160; the main pipeline calls it between the
 * extension index and the unitig extraction, stages/construction.cpp:218-275, with length_bound = read length - k
 * unless configured).  Tips (dead-end branches of at most length_bound k-mers) shorter than the longest outgoing
 * branch of their junction are isolated (mask 0) and the junction's links to them removed.
 * *removed_kmers = isolated k-mers (the count the reference logs), *removed_links = links dropped afterwards. */
int bbk_extindex_clip_tips(bbk_ctx *ctx, bbk_extindex *x, uint32_t length_bound, uint64_t *removed_kmers,
                           uint64_t *removed_links);
void bbk_extindex_free(bbk_extindex *x);

/* ---- unitigs + graph links: replaces UnbranchingPathExtractor::ExtractUnbranchingPathsAndLoops
 *      and FastGraphFromSequencesConstructor::ConstructGraph
 *      (common/assembly_graph/construction/debruijn_graph_constructor.hpp:182-388,390-518) ----- */
int bbk_unitigs_build(bbk_ctx *ctx, bbk_extindex *x, bbk_unitigs **out);
/* ref_threads > 0: perfect loops are collected in the k-mer FILE order of a reference run with -t ref_threads (10 x t
 * XXH3 buckets, ascending inside: kmer_extension_index_builder.hpp:73, CollectLoops debruijn_graph_constructor.hpp:308-344),
 * which fixes where a loop string starts and which palindromic (k+1)-mer SplitLoop (:248-252) cuts a self-conjugate
 * circle at -- the two things that depend on the thread count in the reference itself.  0 = ascending k-mer order. */
int bbk_unitigs_build_ex(bbk_ctx *ctx, bbk_extindex *x, unsigned ref_threads, bbk_unitigs **out);
/* `-c`: coverage of every condensed edge = sum over its (k+1)-mers of their multiplicity in
 * reads + rc(reads) (CoverageHashMapBuilder, common/utils/ph_map/coverage_hash_map_builder.hpp:15-54;
 * FillCoverageAndFlankingFromPHM, assembly_graph/graph_support/coverage_filling.hpp:44-62).  After this
 * call the GFA carries DP:f:<KC/(len-k)> and KC:i:<KC> (projects/gbuilder/main.cpp:200-211). */
int bbk_unitigs_add_coverage(bbk_ctx *ctx, bbk_unitigs *u, const bbk_reads *reads);
/* same from a table the caller has already counted (streaming input: the reads are gone by now): the ascending
 * canonical (k+1)-mer set with multiplicities, bbk_count*(k + 1, BBK_CANONICAL | BBK_WITH_COUNTS) */
int bbk_unitigs_add_coverage_counts(bbk_ctx *ctx, bbk_unitigs *u, const bbk_kmerset *kp1_counts);
int bbk_unitigs_export_kc(bbk_ctx *ctx, const bbk_unitigs *u, uint64_t *h_kc);
uint64_t bbk_unitigs_count(const bbk_unitigs *u);
uint64_t bbk_unitigs_loops(const bbk_unitigs *u);
uint64_t bbk_unitigs_total_bases(const bbk_unitigs *u);
uint64_t bbk_unitigs_vertices(const bbk_unitigs *u);
uint64_t bbk_unitigs_links(const bbk_unitigs *u);
/* The condensed edges as a read set in HBM (packed on the device, nothing crosses PCIe): what the main pipeline does
 * when it feeds the contigs of the previous K into the construction of the next
 * (common/stages/construction.cpp:117-119,228-236: contigs_streams merged into the read streams), and what the
 * full-size tests use to recount the (k+1)-mers of the graph. */
int bbk_unitigs_to_reads(bbk_ctx *ctx, const bbk_unitigs *u, bbk_reads **out);
/* h_bases: total_bases ASCII bytes (no separators); h_offsets: count+1 entries. */
int bbk_unitigs_export(bbk_ctx *ctx, const bbk_unitigs *u, char *h_bases, uint64_t *h_offsets);
/* links: 4 x u32 per link (from_unitig, from_orient(1='+'), to_unitig, to_orient) */
int bbk_unitigs_export_links(bbk_ctx *ctx, const bbk_unitigs *u, uint32_t *h_links);
/* GFA1 text as GFAWriter::WriteSegmentsAndLinks (common/io/graph/gfa_writer.cpp:18-52), segment
 * ids 3+2i (assembly_graph/core/graph_core.hpp:228,610-624); `--unitigs` FASTA as
 * projects/gbuilder/main.cpp:183-192. */
int bbk_unitigs_write_gfa(bbk_ctx *ctx, const bbk_unitigs *u, const char *path);
int bbk_unitigs_write_fasta(bbk_ctx *ctx, const bbk_unitigs *u, const char *path);
/* FASTG as FastgWriter::WriteSegmentsAndLinks (common/io/graph/fastg_writer.cpp:20-47): every edge and its
 * conjugate, header = EDGE_<id>_length_<len>_cov_<cov>['] : successors ; */
int bbk_unitigs_write_fastg(bbk_ctx *ctx, const bbk_unitigs *u, const char *path);
/* SPAdes binary graph <basename>.grseq + <basename>.cvr as io::binary::BasicGraphIO<Graph>().Save
 * (common/io/binary/basic.hpp:24-27, graph.hpp:27-46, coverage.hpp:24-29; gbuilder --spades,
 * projects/gbuilder/main.cpp:221-222).  Edge ids 3+2i as in the GFA; vertex ids in ascending end-k-mer order (the
 * reference's follow its BooPHF indices; its loader takes any consistent numbering). */
int bbk_unitigs_write_spades(bbk_ctx *ctx, const bbk_unitigs *u, const char *basename);
void bbk_unitigs_free(bbk_unitigs *u);


/* ---- several GPUs of one node in one process (bbk_ctx_create(devices, ndev); 8e: the exchange) --------
 * The reference tools are one process for the whole job with hash buckets owned by worker threads
 * (projects/kmercount/main.cpp:186-228, utils/kmer_mph/kmer_buckets.hpp:28-33); here the owner of a canonical k-mer is a
 * GPU, owner(key) = mulhi(mix(key), ndev).  A group names the devices and holds what they need to talk to each other;
 * every rank is driven by ITS OWN host thread, which creates its context with bbk_ctx_create(bbk_group_device(g, rank))
 * and makes all calls for that rank.  The bbk_group_* calls below that take a rank are COLLECTIVE: every rank's thread
 * calls them once, in the same order (like the grouped ncclSend/ncclRecv they issue).  If a rank fails inside one, the
 * others return BBK_ERR_INTERNAL instead of waiting for ever. */
typedef struct bbk_group bbk_group;
#define BBK_EXCHANGE_RCCL 0u /* one grouped ncclSend/ncclRecv all-to-all over xGMI, messages <= 256 MiB (librccl is loaded
                                at this point, not before; devices must be distinct) */
#define BBK_EXCHANGE_COPY 1u /* the same segments moved by peer copies (hipMemcpyPeerAsync); ranks may share a device:
                                the way to run the N-rank path on one GPU */
int bbk_group_create(const int *devices, int ndev, unsigned exchange, bbk_group **out);
int bbk_group_size(const bbk_group *g);
int bbk_group_device(const bbk_group *g, int rank);
void bbk_group_destroy(bbk_group *g);
/* a rank that cannot reach its next collective call (input error, ...) tells the others */
void bbk_group_abort(bbk_group *g);
/* local: this rank's distinct canonical k-mers in any order (bbk_count*(BBK_CANONICAL | BBK_UNSORTED [| BBK_WITH_COUNTS]))
 * -> *shard: the canonical k-mers this rank owns, merged over all ranks (multiplicities summed);
 * flags: BBK_UNSORTED keeps hash-bucket order (enough for bbk_kmerset_both_strands_ex), 0 sorts ascending. */
int bbk_group_exchange_kmers(bbk_group *g, int rank, bbk_ctx *ctx, const bbk_kmerset *local, unsigned flags,
                             bbk_kmerset **shard);
/* local: BBK_CANONICAL | BBK_UNSORTED | BBK_WITH_MASKS records -> this rank's shard of the extension index (ascending
 * canonical k-mers it owns with their complete InOutMask): "ext records follow their k-mer's owner, no second exchange" */
int bbk_group_exchange_extindex(bbk_group *g, int rank, bbk_ctx *ctx, const bbk_kmerset *local_masks,
                                bbk_extindex **shard);
/* all shards -> one index on rank dst (*full; NULL on the other ranks): the unitig walk crosses owners */
int bbk_group_gather_extindex(bbk_group *g, int rank, bbk_ctx *ctx, const bbk_extindex *shard, int dst,
                              bbk_extindex **full);
/* the same for a sharded set with multiplicities -> one ascending set on rank dst (gbuilder -c: the (k+1)-mer counts) */
int bbk_group_gather_kmers(bbk_group *g, int rank, bbk_ctx *ctx, const bbk_kmerset *shard, int dst, bbk_kmerset **full);

/* Device memory of the calling thread's allocator on the context's device: bytes mapped now, bytes mapped since the
 * process started (growth is what a first call pays: ~30 ms/GiB when the driver has to clear the memory first), seconds
 * spent mapping, and the device's free / total bytes as the driver reports them.  Any pointer may be NULL. */
int bbk_ctx_memory_stats(bbk_ctx *ctx, uint64_t *mapped_now, uint64_t *mapped_total, double *map_seconds,
                         uint64_t *device_free, uint64_t *device_total);
/* What the engine found on the device at bbk_ctx_create: compute units, and the XCDs a probe launch's workgroups were seen
 * on (HW_REG_XCC_ID).  The partition kernels keep one fill front per (segment, XCD) and deal level-2 tiles to the XCDs by
 * segment when that is eight (an MI355X in SPX mode); any other value -- a partitioned device -- switches both off.
 * Placement only: results never depend on it.  (Diagnostics; no reference counterpart.) */
int bbk_ctx_device_info(bbk_ctx *ctx, int *num_cus, int *num_xcds);
/* XXH3 bucket boundaries of a set stored in the final_kmers order: h_offsets[b] = first record of bucket b (b = 0..16,
 * h_offsets[16] = size); what a writer that merges several shards into one final_kmers file needs
 * (KMerDiskStorage::merge, kmer_index_builder.hpp:168-181) */
int bbk_kmerset_bucket_offsets(bbk_ctx *ctx, const bbk_kmerset *s, uint64_t *h_offsets);

#ifdef __cplusplus
}
#endif
#endif /* BBK_H_ */
