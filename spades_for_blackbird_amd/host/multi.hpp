// multi.hpp -- the CLIs on several GPUs of one node (`--devices a,b,...`): ONE process like the reference tools
// (projects/kmercount/main.cpp:186-228, projects/gbuilder/main.cpp:89-233), one host thread + one context per device.
// The parser thread (ingest.hpp) fills a pool of blocks; whichever rank is free takes the next block, uploads it to ITS
// device and pushes it into its local accumulator (the reference's worker threads pull reads from the one parsing
// thread the same way, common/io/reads/read_processor.hpp:76-135).  When the input is exhausted every rank calls its
// `finish`, which holds the collective steps (bbk_group_exchange_*: the one all-to-all of SURVEY 8e).
#pragma once

#include <fcntl.h>

#include <atomic>
#include <deque>
#include <future>

#include "common.hpp"

namespace bbkhost {

inline bool parse_devices(const std::string &arg, std::vector<int> &out) {
    out.clear();
    size_t i = 0;
    while (i < arg.size()) {
        size_t j = arg.find(',', i);
        if (j == std::string::npos) j = arg.size();
        unsigned long long v = 0;
        if (j == i || !parse_uint(arg.substr(i, j - i).c_str(), &v) || v > 1023) return false;
        out.push_back((int)v);
        i = j + 1;
    }
    return !out.empty();
}

// The group (RCCL: loading librccl and ncclCommInitAll take seconds) is set up on a thread of its own while the ranks
// parse and count; the first collective call waits for it.
inline std::shared_future<bbk_group *> create_group_async(const std::vector<int> &devices, unsigned exchange) {
    return std::async(std::launch::async, [devices, exchange] {
               bbk_group *g = nullptr;
               const double t0 = now_s();
               check(bbk_group_create(devices.data(), (int)devices.size(), exchange, &g), "bbk_group_create");
               info("Device group ready (%d device(s), %s exchange, %.2f s)", (int)devices.size(),
                    exchange == BBK_EXCHANGE_RCCL ? "RCCL" : "peer-copy", now_s() - t0);
               return g;
           })
        .share();
}

struct RankHooks {
    std::function<void(int rank, bbk_ctx *ctx)> init;                      // accumulators of the rank
    std::function<void(int rank, bbk_ctx *ctx, bbk_reads *r)> push;        // one uploaded block
    std::function<void(int rank, bbk_ctx *ctx)> finish;                    // collectives + per-rank results
};

// Runs the ranks to completion; returns the number of reads.  ctxs[rank] receives the rank's context (left alive: the
// caller's writer may still need the results that live on it).  Any failure ends the process (fatal), like the single-
// device path.
inline uint64_t run_ranks(const std::vector<int> &devices, const std::vector<std::string> &files, size_t block_bytes, int threads,
                          Phases &ph, const RankHooks &hooks, std::vector<bbk_ctx *> &ctxs) {
    const int n = (int)devices.size();
    ctxs.assign((size_t)n, nullptr);
    Ingest ing(files, block_bytes, threads);
    ing.on_file = [](const std::string &f) { info("Processing %s", f.c_str()); };
    const int nslots = n + 1;
    std::vector<PackedReads> slot((size_t)nslots);
    std::mutex mu;
    std::condition_variable cv;
    std::deque<int> free_slots, ready;
    for (int i = 0; i < nslots; ++i) free_slots.push_back(i);
    bool done = false;
    std::string err;
    double parse_s = 0;
    std::thread producer([&] {
        for (;;) {
            int s;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !free_slots.empty(); });
                s = free_slots.front();
                free_slots.pop_front();
            }
            const double t0 = now_s();
            const bool more = ing.next(slot[(size_t)s], err);
            parse_s += now_s() - t0;
            {
                std::lock_guard<std::mutex> lk(mu);
                if (more) ready.push_back(s);
                else done = true;
            }
            cv.notify_all();
            if (!more) return;
        }
    });
    std::atomic<uint64_t> total{0}, blocks{0};
    std::vector<double> up((size_t)n, 0), dev((size_t)n, 0), ctx_s((size_t)n, 0);
    std::vector<std::thread> ranks;
    for (int r = 0; r < n; ++r)
        ranks.emplace_back([&, r] {
            const double t0c = now_s();
            check(bbk_ctx_create(devices[(size_t)r], &ctxs[(size_t)r]), "bbk_ctx_create");
            ctx_s[(size_t)r] = now_s() - t0c;
            bbk_ctx *ctx = ctxs[(size_t)r];
            if (hooks.init) hooks.init(r, ctx);
            for (;;) {
                int s = -1;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return !ready.empty() || done; });
                    if (ready.empty()) break;
                    s = ready.front();
                    ready.pop_front();
                }
                PackedReads &p = slot[(size_t)s];
                bbk_reads *rd = nullptr;
                double t0 = now_s();
                check(bbk_reads_from_packed(ctx, p.words.data(), p.words.size(), p.len.data(), p.len.size(), &rd),
                      "bbk_reads_from_packed");
                up[(size_t)r] += now_s() - t0;
                total += p.size();
                {
                    std::lock_guard<std::mutex> lk(mu);
                    free_slots.push_back(s);  // the upload has completed: the parser may refill the slot
                }
                cv.notify_all();
                t0 = now_s();
                hooks.push(r, ctx, rd);
                dev[(size_t)r] += now_s() - t0;
                bbk_reads_free(rd);
                ++blocks;
            }
            const double t0 = now_s();
            if (hooks.finish) hooks.finish(r, ctx);
            dev[(size_t)r] += now_s() - t0;
        });
    for (auto &t : ranks) t.join();
    producer.join();
    if (!err.empty()) fatal("%s", err.c_str());
    ph.parse = parse_s;
    ph.blocks = blocks;
    ph.fallback_blocks = ing.fallback_blocks();
    for (int r = 0; r < n; ++r) {  // the slowest rank is what the wall clock saw
        ph.ctx = std::max(ph.ctx, ctx_s[(size_t)r]);
        ph.upload = std::max(ph.upload, up[(size_t)r]);
        ph.device = std::max(ph.device, dev[(size_t)r]);
    }
    info("Total %llu reads processed", (unsigned long long)total.load());
    return total;
}

// ---- final_kmers from N shards ------------------------------------------------------------------------------------
// KMerDiskStorage::merge (common/utils/kmer_mph/kmer_index_builder.hpp:168-181) concatenates the 16 bucket files; a
// bucket here is spread over the ranks' shards (each ascending inside the bucket, disjoint k-mers): bucket b of the file
// is the N-way merge of the ranks' runs of bucket b.  Buckets are merged in parallel and written at their final offsets.
struct ShardOnHost {
    std::vector<uint64_t> keys;   // records * W words, final_kmers order
    uint64_t off[17] = {0};       // bucket boundaries in records
};

inline bool write_final_kmers_merged(const std::vector<ShardOnHost> &sh, unsigned W, const std::string &path,
                                     uint64_t *n_total) {
    const size_t N = sh.size(), rec = (size_t)W * 8;
    uint64_t bucket_start[17];
    bucket_start[0] = 0;
    for (int b = 0; b < 16; ++b) {
        uint64_t c = 0;
        for (size_t r = 0; r < N; ++r) c += sh[r].off[b + 1] - sh[r].off[b];
        bucket_start[b + 1] = bucket_start[b] + c;
    }
    *n_total = bucket_start[16];
    const int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return false;
    if (bucket_start[16]) (void)fallocate(fd, 0, 0, (off_t)(bucket_start[16] * rec));
    bool ok = true;
#pragma omp parallel for schedule(dynamic, 1) num_threads(16)
    for (int b = 0; b < 16; ++b) {
        std::vector<const uint64_t *> cur(N), end(N);
        for (size_t r = 0; r < N; ++r) {
            cur[r] = sh[r].keys.data() + sh[r].off[b] * W;
            end[r] = sh[r].keys.data() + sh[r].off[b + 1] * W;
        }
        const size_t cap = (8u << 20) / rec;  // records per write
        std::vector<uint64_t> out(cap * W);
        size_t fill = 0;
        uint64_t at = bucket_start[b];
        auto flush = [&] {
            size_t done = 0;
            const char *p = (const char *)out.data();
            while (done < fill * rec) {
                const ssize_t w = pwrite(fd, p + done, fill * rec - done, (off_t)(at * rec + done));
                if (w <= 0) {
#pragma omp atomic write
                    ok = false;
                    return;
                }
                done += (size_t)w;
            }
            at += fill;
            fill = 0;
        };
        for (;;) {
            // smallest head in record order (word 0 most significant, adt/array_vector.hpp:114-123)
            int best = -1;
            for (size_t r = 0; r < N; ++r) {
                if (cur[r] == end[r]) continue;
                if (best < 0) {
                    best = (int)r;
                    continue;
                }
                const uint64_t *a = cur[r], *c = cur[(size_t)best];
                for (unsigned w = 0; w < W; ++w)
                    if (a[w] != c[w]) {
                        if (a[w] < c[w]) best = (int)r;
                        break;
                    }
            }
            if (best < 0) break;
            for (unsigned w = 0; w < W; ++w) out[fill * W + w] = cur[(size_t)best][w];
            cur[(size_t)best] += W;
            if (++fill == cap) flush();
        }
        if (fill) flush();
    }
    return close(fd) == 0 && ok;
}

}  // namespace bbkhost
