// accum.h -- accumulator behind the streaming entry points (bbk_count_begin / push / finish,
// bbk_extindex_begin / push / finish); defined in count.hip.
#pragma once

#include <vector>

#include "bbk_internal.h"

namespace bbk {

// What the reference gets from bounded per-thread cells, repeated DumpBuffers rounds (one sorted + uniqued run per
// bucket and round, common/utils/kmer_mph/kmer_splitter.hpp:73-167) and the final loser-tree run merge (MergeKMers,
// kmer_index_builder.hpp:281-365): the input never has to be resident as a whole.  Every pushed batch is
// deduplicated on its own (stage A) and kept as a "run" of distinct canonical records; runs are merge-uniqued into
// the accumulated set whenever they outweigh half of it (the total merge work stays linear in the input), and
// finish orders the set once (stage B).
struct Accum {
    bbk_ctx *ctx = nullptr;
    unsigned k = 0;
    bool with_mask = false;  // payload = InOutMask bits (OR) instead of multiplicities (SUM)
    bool want_vals = false;
    DevBuf keys, vals;       // the accumulated distinct canonical set (any order)
    uint64_t n = 0;
    struct Run {
        DevBuf keys, vals;
        uint64_t n = 0;
    };
    std::vector<Run> runs;
    uint64_t runs_n = 0;
    uint64_t instances = 0;  // k-mer positions seen
    uint64_t batches = 0, merges = 0;

    bool has_vals() const;
    int merge_op() const;
    void push(const bbk_reads *rd);
    // a record array already in HBM (keys + payloads, duplicates allowed) becomes one more run
    void push_records(const void *d_keys, const uint32_t *d_vals, uint64_t n_rec);
    void merge();
    uint64_t finish_sorted(DevBuf &out_keys, DevBuf &out_vals);
};

// count.hip: canon U rc(canon) of the accumulated canonical records (payloads dropped); the accumulator keeps them
bbk_kmerset *both_strands_of(Accum &acc, unsigned flags);

uint64_t drop_zero_vals(bbk_ctx *ctx, int W, const void *keys, const uint32_t *vals, uint64_t n, DevBuf &out_keys,
                        DevBuf &out_vals);

}  // namespace bbk
