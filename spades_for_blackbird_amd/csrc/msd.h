// msd.h -- interface of the MSD-partition + in-LDS sort path (msd.hip).
#pragma once

#include "bbk_internal.h"

namespace bbk {

enum { MSD_HASH = 0, MSD_KEYS = 1, MSD_REF = 2 };                      // partition prefix
enum { MSD_OP_NONE = 0, MSD_OP_COUNT = 1, MSD_OP_SUM = 2, MSD_OP_OR = 3 };  // per-key reduction

struct MsdOutput {
    DevBuf keys;        // distinct records: bucket-major in prefix order, ascending inside a bucket
    DevBuf vals;        // reduced payload (op != NONE)
    DevBuf bucket_off;  // nbuckets + 1 offsets into keys
    uint64_t n = 0;
    uint64_t instances = 0;
    uint64_t overflow_buckets = 0;
    uint32_t nbuckets = 0;
};

// Sort + reduce records that come either from reads (rd != null: canonical k-mers are extracted on
// the fly, with_mask adds the InOutMask bits of every occurrence as payload) or from a key array
// (d_keys[, d_vals], n).  Returns false when this path declines (key width, size, too much
// overflow): the caller then uses the LSD path.  MSD_KEYS output is globally ascending.
// tag_bits > 0 (8-byte keys, key array input, MSD_KEYS): bits [2k, 2k + tag_bits) of every key hold a tag
// that is more significant than the k-mer (the caller put it there); the records are ordered by
// (tag, k-mer) and the tag is cleared in the output.
bool msd_sort_reduce(bbk_ctx *ctx, unsigned k, int dmode, int op, const bbk_reads *rd, const void *d_keys,
                     const uint32_t *d_vals, uint64_t n, bool with_mask, MsdOutput &out, unsigned tag_bits = 0,
                     bool assume_distinct = false, unsigned expand_k = 0);
// expand_k = k (key array input): the array holds n CANONICAL k-mers and the records are generated on the fly -- 2n of
// them: every key and its reverse complement, with the tag of tag_bits (the XXH3 bucket of 16) written by the
// level-1 kernels themselves.
// assume_distinct (key array, KEYS / REF prefix): the caller expects no duplicates, so the sorted result is written
// directly at the offsets of the input (no compaction pass); verified on the fly, redone in place otherwise.

// superk.hip: stage A of a batch of reads for 16- and 24-byte keys through super-k-mer records (distinct canonical
// k-mers + count / OR of edge masks, in any order).  false: not taken or given up -- use msd_sort_reduce.
bool superk_dedup_reads(bbk_ctx *ctx, const bbk_reads *rd, unsigned k, int op, DevBuf &out_keys, DevBuf &out_vals,
                        uint64_t &n_distinct, uint64_t &n_instances);

}  // namespace bbk
