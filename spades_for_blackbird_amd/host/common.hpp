// common.hpp -- shared host plumbing of the two CLIs: logging in the reference's style
// (common/utils/logger: elapsed time prefix), C-ABI error handling, batched upload of reads.
#pragma once

#include <unistd.h>

#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <omp.h>

#include "../../include/bbk.h"
#include "dataset.hpp"
#include "fastx.hpp"
#include "ingest.hpp"

namespace bbkhost {

inline double &t0_ref() {
    static double t0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    return t0;
}

inline void info(const char *fmt, ...) {
    const double now = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    const double el = now - t0_ref();
    const int h = (int)(el / 3600), m = (int)(el / 60) % 60, s = (int)el % 60, ms = (int)((el - (long)el) * 1000);
    fprintf(stdout, "%3d:%02d:%02d.%03d  INFO  ", h, m, s, ms);
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stdout, fmt, ap);
    va_end(ap);
    fputc('\n', stdout);
    fflush(stdout);
}

// FATAL_ERROR analogue (common/utils/logger/logger.hpp:177-190): message, then exit(-1)
[[noreturn]] inline void fatal(const char *fmt, ...) {
    fprintf(stderr, "=== Error ===\n");
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
    exit(-1);
}

inline void check(int rc, const char *what) {
    if (rc != BBK_OK) fatal("%s failed (%d): %s", what, rc, bbk_last_error());
}

inline double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// wall seconds of the phases of a run; printed as one machine-readable line when BBK_PHASES is set (bench.py reads it)
struct Phases {
    double ctx = 0, parse = 0, upload = 0, device = 0, finish = 0, write = 0, total = 0, parse_wait = 0;
    uint64_t blocks = 0, fallback_blocks = 0;
    // device memory the calling thread's allocator mapped and what mapping it cost (part of device_s / finish_s: ~30 ms
    // per GiB when the driver has to clear the memory first, ~0.1 ms on memory nobody has used since boot)
    double map_s = 0, mapped_gb = 0;
    void memory(bbk_ctx *c) {
        uint64_t total = 0;
        if (c && bbk_ctx_memory_stats(c, nullptr, &total, &map_s, nullptr, nullptr) == BBK_OK) mapped_gb = (double)total / 1e9;
    }
    void report(const char *tool) const {
        if (!getenv("BBK_PHASES")) return;
        printf("BBK_PHASES {\"tool\": \"%s\", \"ctx_s\": %.4f, \"parse_s\": %.4f, \"parse_wait_s\": %.4f, \"upload_s\": %.4f, "
               "\"device_s\": %.4f, \"finish_s\": %.4f, \"write_s\": %.4f, \"total_s\": %.4f, \"blocks\": %llu, "
               "\"fallback_blocks\": %llu, \"memory_mapped_gb\": %.2f, \"memory_map_s\": %.4f}\n",
               tool, ctx, parse, parse_wait, upload, device, finish, write, total, (unsigned long long)blocks,
               (unsigned long long)fallback_blocks, mapped_gb, map_s);
        fflush(stdout);
    }
};

// End of a tool whose outputs are written and closed: leave without tearing down the context and the HIP runtime
// (unmapping tens of GB of device and pinned memory chunk by chunk cost 0.1-0.15 s of a 1 s run; the process image goes
// away as a whole anyway).  BBK_FULL_TEARDOWN=1 (leak checks, sanitizer builds) takes the long way.
[[noreturn]] inline void finish_process(bbk_ctx *ctx, int code) {
    fflush(stdout);
    fflush(stderr);
    if (getenv("BBK_FULL_TEARDOWN")) {
        bbk_ctx_destroy(ctx);
        exit(code);
    }
    _exit(code);
}

inline int default_threads() {
    int t = omp_get_max_threads();
    if (t > 16) t = 16;  // the parser saturates memory bandwidth long before that
    return t < 1 ? 1 : t;
}

// Streams every file through the parallel block parser (ingest.hpp) and hands each block, uploaded as a bbk_reads, to
// `push`.  One producer thread parses block i+1 while this thread uploads and pushes block i (the reference overlaps
// its parsing master thread with the consuming workers the same way, common/io/reads/read_processor.hpp:76-135).
// block_bytes bounds the host memory: two blocks of text-equivalent packed reads are alive at any time (the -b
// contract of the reference: bounded buffers, kmer_splitter.hpp:73-109).
// `init` (optional) runs on this thread right after the parser thread has been started and before the first upload:
// the tools create the HIP context (0.1-0.2 s) and their accumulators there, under the parse of the first block; `ctx` is
// read after it.
template <class Push>
inline uint64_t stream_reads(bbk_ctx *&ctx, const std::vector<std::string> &files, size_t block_bytes, int threads,
                             Phases &ph, Push push, const std::function<void()> &init = nullptr) {
    // a single SPAdes binary read cache (<prefix>.seq) is taken as is
    if (files.size() == 1 && ends_with(files[0], ".seq")) {
        if (init) init();
        info("Processing %s (binary read cache)", files[0].c_str());
        bbk_reads *r = nullptr;
        check(bbk_reads_from_spades_binary(ctx, files[0].c_str(), &r), "bbk_reads_from_spades_binary");
        const uint64_t n = bbk_reads_count(r);
        const double t0 = now_s();
        push(r);
        ph.device += now_s() - t0;
        bbk_reads_free(r);
        info("Total %llu reads processed", (unsigned long long)n);
        return n;
    }
    Ingest ing(files, block_bytes, threads);
    ing.on_file = [](const std::string &f) { info("Processing %s", f.c_str()); };
    PackedReads slot[2];
    std::mutex mu;
    std::condition_variable cv;
    int ready[2] = {0, 0};  // 0 = free, 1 = filled, 2 = end of input
    std::string err;
    double parse_s = 0;
    std::thread producer([&] {
        for (int i = 0;; i ^= 1) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return ready[i] == 0; });
            }
            const double t0 = now_s();
            const bool more = ing.next(slot[i], err);
            parse_s += now_s() - t0;
            {
                std::lock_guard<std::mutex> lk(mu);
                ready[i] = more ? 1 : 2;
            }
            cv.notify_all();
            if (!more) return;
        }
    });
    if (init) init();
    uint64_t total = 0;
    std::string fail;
    for (int i = 0;; i ^= 1) {
        const double tw = now_s();
        int st;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return ready[i] != 0; });
            st = ready[i];
        }
        ph.parse_wait += now_s() - tw;
        if (st == 2) break;
        bbk_reads *r = nullptr;
        if (fail.empty()) {
            const double t0 = now_s();
            const int rc = bbk_reads_from_packed(ctx, slot[i].words.data(), slot[i].words.size(), slot[i].len.data(),
                                                 slot[i].len.size(), &r);
            ph.upload += now_s() - t0;
            if (rc != BBK_OK) fail = std::string("bbk_reads_from_packed: ") + bbk_last_error();
        }
        total += slot[i].size();
        {
            std::lock_guard<std::mutex> lk(mu);
            ready[i] = 0;  // the upload has completed: the producer may refill this slot
        }
        cv.notify_all();
        if (r) {
            const double t0 = now_s();
            push(r);
            ph.device += now_s() - t0;
            bbk_reads_free(r);
            ++ph.blocks;
        }
    }
    producer.join();
    ph.parse = parse_s;
    ph.fallback_blocks = ing.fallback_blocks();
    if (!err.empty()) fatal("%s", err.c_str());
    if (!fail.empty()) fatal("%s", fail.c_str());
    info("Total %llu reads processed", (unsigned long long)total);
    return total;
}

inline bool parse_uint(const char *s, unsigned long long *v) {
    if (!s || !*s) return false;
    char *end = nullptr;
    *v = strtoull(s, &end, 10);
    return end && *end == 0;
}

}  // namespace bbkhost
