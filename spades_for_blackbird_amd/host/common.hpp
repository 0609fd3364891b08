// common.hpp -- shared host plumbing of the two CLIs: logging in the reference's style
// (common/utils/logger: elapsed time prefix), C-ABI error handling, batched upload of reads.
#pragma once

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/bbk.h"
#include "dataset.hpp"
#include "fastx.hpp"

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

// Reads every file into one host batch (names and qualities dropped) and uploads it.
inline bbk_reads *load_reads(bbk_ctx *ctx, const std::vector<std::string> &files, uint64_t *n_reads) {
    // a single SPAdes binary read cache (<prefix>.seq) is taken as is
    if (files.size() == 1 && ends_with(files[0], ".seq")) {
        info("Processing %s (binary read cache)", files[0].c_str());
        bbk_reads *r = nullptr;
        check(bbk_reads_from_spades_binary(ctx, files[0].c_str(), &r), "bbk_reads_from_spades_binary");
        if (n_reads) *n_reads = bbk_reads_count(r);
        info("Total %llu reads processed", (unsigned long long)bbk_reads_count(r));
        return r;
    }
    ReadBatch batch;
    for (const std::string &f : files) {
        info("Processing %s", f.c_str());
        FastxReader rd(f);
        if (!rd.is_open()) fatal("Cannot open %s", f.c_str());
        while (rd.read(batch, ~0ull, ~0ull) > 0) {}
    }
    info("Total %llu reads processed", (unsigned long long)batch.size());
    bbk_reads *r = nullptr;
    check(bbk_reads_from_ascii(ctx, batch.bases.data(), batch.offsets.data(), batch.size(), &r), "bbk_reads_from_ascii");
    if (n_reads) *n_reads = batch.size();
    return r;
}

inline bool parse_uint(const char *s, unsigned long long *v) {
    if (!s || !*s) return false;
    char *end = nullptr;
    *v = strtoull(s, &end, 10);
    return end && *end == 0;
}

}  // namespace bbkhost
