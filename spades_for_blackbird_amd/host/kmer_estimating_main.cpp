// spades-kmer-estimating drop-in: same argv contract and the same result line as the reference tool
// (projects/kmercount/kmer_estimating.cpp:39-58 for the flags, :60-104 for the flow).  The reference feeds a
// symmetric (strand-independent) rolling hash of every k-mer of the reads into a HyperLogLog
// (common/utils/kmer_counting.hpp:18-43,182-263) and prints the ESTIMATE of the number of distinct k-mers, reverse
// complements identified.  The engine counts them exactly (bbk_count, canonical set, dedup only), so the number
// printed here is the exact cardinality the estimate approximates (HLL error is ~1 %).
//   -k/--kmer <int=21>  -d/--dataset <yaml> (required)  -t/--threads <int>  -h/--help     (+ --device <int>, ours)
#include <string>
#include <vector>

#include "common.hpp"

using namespace bbkhost;

static void usage(const char *argv0) {
    printf("SYNOPSIS\n        %s [-k <value>] -d <dir> [-t <value>] [-h]\n\n"
           "OPTIONS\n"
           "        -k, --kmer <value>      K-mer length\n"
           "        -d, --dataset <dir>     Dataset description (in YAML)\n"
           "        -t, --threads <value>   # of threads to use\n"
           "        -h, --help              Show help\n"
           "        --device <value>        GPU to use (default 0)\n\n"
           "DESCRIPTION\n         Kmer number estimating.  Kmers from reverse-complementary reads aren't taken into account.\n",
           argv0);
}

int main(int argc, char **argv) {
    unsigned K = 21, device = 0;
    std::string dataset;
    bool help = false, bad = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto need = [&](unsigned long long *v) { return i + 1 < argc && parse_uint(argv[++i], v); };
        unsigned long long v = 0;
        if (a == "-k" || a == "--kmer") { if (need(&v)) K = (unsigned)v; else bad = true; }
        else if (a == "-t" || a == "--threads") { if (!need(&v)) bad = true; }
        else if (a == "--device") { if (need(&v)) device = (unsigned)v; else bad = true; }
        else if (a == "-d" || a == "--dataset") { if (i + 1 < argc) dataset = argv[++i]; else bad = true; }
        else if (a == "-h" || a == "--help") help = true;
        else bad = true;
    }
    if (bad || help || dataset.empty()) {  // kmer_estimating.cpp:50-57: -d is required
        usage(argv[0]);
        return help ? 0 : 1;
    }
    if (K < 1 || K >= BBK_MAX_K) fatal("k-mer size %u is out of range [1, %d)", K, BBK_MAX_K);

    info("Starting kmer spectra cardinality (MI355X, %s)", bbk_version());
    info("K-mer length set to %u", K);
    std::vector<std::string> files;
    std::string err;
    if (!load_dataset_yaml(dataset, files, err)) fatal("%s", err.c_str());
    bbk_ctx *ctx = nullptr;
    check(bbk_ctx_create((int)device, &ctx), "bbk_ctx_create");
    info("Estimating kmer cardinality");
    // strand-independent distinct k-mers: the canonical set; hash-bucket order is enough (no sort); streamed block by block
    bbk_counter *counter = nullptr;
    check(bbk_count_begin(ctx, K, BBK_CANONICAL | BBK_UNSORTED, &counter), "bbk_count_begin");
    Phases ph;
    stream_reads(ctx, files, 512u << 20, default_threads(), ph,
                 [&](bbk_reads *r) { check(bbk_count_push_reads(counter, r), "bbk_count_push_reads"); });
    bbk_kmerset *set = nullptr;
    check(bbk_count_finish(counter, &set), "bbk_count_finish");
    info("Kmer number estimation: %llu", (unsigned long long)bbk_kmerset_size(set));  // :99, exact here
    bbk_kmerset_free(set);
    finish_process(ctx, 0);
}
