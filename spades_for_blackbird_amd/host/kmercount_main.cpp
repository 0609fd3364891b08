// spades-kmercount drop-in: same argv contract and the same <workdir>/final_kmers file as the
// reference tool (projects/kmercount/main.cpp:124-184 for the flags, :186-228 for the flow), with
// the splitter + KMerDiskCounter replaced by the MI355X engine behind the C ABI (include/bbk.h).
//   -k/--kmer <int=21>  -d/--dataset <yaml>  -t/--threads <int>  -w/--workdir <dir>
//   -b/--bufsize <bytes>  -h/--help  [input files...]        (+ --device <int>, --devices a,b,.. [--exchange rccl|copy], ours)
// -t = parser threads (the reference: OpenMP threads of the splitter); -b = bytes of input text per block (the
// reference: sorting buffer per thread): the input is streamed through bbk_count_begin / push / finish block by block,
// so host memory is bounded by two blocks and device memory by one block + the distinct set -- the reference's
// bounded-memory contract (kmer_splitter.hpp:73-109).
#include <cerrno>
#include <cstring>
#include <string>
#include <vector>

#include <sys/stat.h>

#include "common.hpp"
#include "multi.hpp"

using namespace bbkhost;

static void usage(const char *argv0) {
    printf("SYNOPSIS\n        %s [-k <value>] [-d <dir>] [-t <value>] [-w <dir>] [-b <value>] [-h] [<input files>]...\n\n"
           "OPTIONS\n"
           "        -k, --kmer <value>      K-mer length\n"
           "        -d, --dataset <dir>     Dataset description (in YAML), input files ignored\n"
           "        -t, --threads <value>   # of threads to use\n"
           "        -w, --workdir <dir>     Working directory to use\n"
           "        -b, --bufsize <value>   Sorting buffer size, per thread\n"
           "        -h, --help              Show help\n"
           "        --device <value>        GPU to use (default 0)\n"
           "        --devices <a,b,...>     GPUs to use: one host thread + context per device, k-mers sharded by owner hash,\n"
           "                                one RCCL all-to-all after the local count (--exchange copy: peer copies)\n\n"
           "DESCRIPTION\n        SPAdes k-mer counting engine (MI355X)\n\n"
           "        Output: <output_dir>/final_kmers - unordered set of kmers in binary format. Kmers from both forward and\n"
           "        reverse-complementary reads are taken into account.\n\n"
           "        Output format: All kmers are written sequentially without any separators. Each kmer takes the same\n"
           "        number of bits. One kmer of length K takes 2*K bits. Kmers are aligned by 64 bits. Each nucleotide is\n"
           "        coded with 2 bits: 00 - A, 01 - C, 10 - G, 11 - T.\n",
           argv0);
}

int main(int argc, char **argv) {
    unsigned K = 21, device = 0;
    unsigned long long threads = 0, bufsize = 536870912ull;  // projects/kmercount/main.cpp:124-130
    std::string workdir, dataset, devices_arg, exchange_arg = "rccl";
    std::vector<std::string> input;
    bool help = false, bad = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto need = [&](unsigned long long *v) { return i + 1 < argc && parse_uint(argv[++i], v); };
        unsigned long long v = 0;
        if (a == "-k" || a == "--kmer") { if (need(&v)) K = (unsigned)v; else bad = true; }
        else if (a == "-t" || a == "--threads") { if (need(&v)) threads = v; else bad = true; }
        else if (a == "-b" || a == "--bufsize") { if (need(&v)) bufsize = v; else bad = true; }
        else if (a == "--device") { if (need(&v)) device = (unsigned)v; else bad = true; }
        else if (a == "--devices") { if (i + 1 < argc) devices_arg = argv[++i]; else bad = true; }
        else if (a == "--exchange") { if (i + 1 < argc) exchange_arg = argv[++i]; else bad = true; }
        else if (a == "-d" || a == "--dataset") { if (i + 1 < argc) dataset = argv[++i]; else bad = true; }
        else if (a == "-w" || a == "--workdir") { if (i + 1 < argc) workdir = argv[++i]; else bad = true; }
        else if (a == "-h" || a == "--help") help = true;
        else if (!a.empty() && a[0] == '-' && a.size() > 1) bad = true;
        else input.push_back(a);
    }
    if (bad || help) {  // projects/kmercount/main.cpp:169-176
        usage(argv[0]);
        return help ? 0 : 1;
    }
    if (input.empty() && dataset.empty()) {  // :178-182
        fprintf(stderr, "ERROR: No input files were specified\n\n");
        usage(argv[0]);
        return 255;  // exit(-1)
    }
    if (K < 1 || K >= BBK_MAX_K) fatal("k-mer size %u is out of range [1, %d)", K, BBK_MAX_K);

    info("Starting SPAdes k-mer counting engine (MI355X, %s)", bbk_version());
    info("K-mer length set to %u", K);
    std::vector<std::string> files = input;
    if (!dataset.empty()) {
        files.clear();
        std::string err;
        if (!load_dataset_yaml(dataset, files, err)) fatal("%s", err.c_str());
    }
    std::vector<int> devices;
    if (!devices_arg.empty() && !parse_devices(devices_arg, devices)) fatal("--devices: expected a comma-separated list of GPU indices");
    if (exchange_arg != "rccl" && exchange_arg != "copy") fatal("--exchange: rccl or copy");
    Phases ph;
    const double t_start = now_s();
    if (!devices.empty()) {
        // ---- several devices (or one, through the same code): local count -> owner exchange -> per-rank final_kmers order
        //      -> one file.  The group (RCCL communicators) is set up while the parser reads the first block.
        auto gf = create_group_async(devices, exchange_arg == "rccl" ? BBK_EXCHANGE_RCCL : BBK_EXCHANGE_COPY);
        const int n = (int)devices.size();
        info("Using %d device(s), %s exchange", n, exchange_arg.c_str());
        const unsigned W = bbk_words(K);
        std::vector<bbk_counter *> counters((size_t)n, nullptr);
        std::vector<ShardOnHost> shards((size_t)n);
        std::vector<bbk_ctx *> ctxs;
        RankHooks hooks;
        hooks.init = [&](int r, bbk_ctx *c) {
            check(bbk_count_begin(c, K, BBK_CANONICAL | BBK_UNSORTED, &counters[(size_t)r]), "bbk_count_begin");
        };
        hooks.push = [&](int r, bbk_ctx *, bbk_reads *rd) { check(bbk_count_push_reads(counters[(size_t)r], rd), "bbk_count_push_reads"); };
        hooks.finish = [&](int r, bbk_ctx *c) {
            bbk_kmerset *local = nullptr, *shard = nullptr, *both = nullptr;
            check(bbk_count_finish(counters[(size_t)r], &local), "bbk_count_finish");
            check(bbk_group_exchange_kmers(gf.get(), r, c, local, BBK_UNSORTED, &shard), "bbk_group_exchange_kmers");
            bbk_kmerset_free(local);
            // a k-mer and its reverse complement share the canonical owner: the shard expands on its own
            check(bbk_kmerset_both_strands_ex(c, shard, BBK_REFERENCE_ORDER, &both), "bbk_kmerset_both_strands_ex");
            bbk_kmerset_free(shard);
            ShardOnHost &sh = shards[(size_t)r];
            check(bbk_kmerset_bucket_offsets(c, both, sh.off), "bbk_kmerset_bucket_offsets");
            sh.keys.resize((size_t)bbk_kmerset_size(both) * W);
            check(bbk_kmerset_export(c, both, BBK_ORDER_REFERENCE_BUCKETS16, sh.keys.data(), nullptr), "bbk_kmerset_export");
            bbk_kmerset_free(both);
        };
        run_ranks(devices, files, (size_t)bufsize, threads ? (int)threads : default_threads(), ph, hooks, ctxs);
        if (!workdir.empty()) mkdir(workdir.c_str(), 0755);
        const std::string out = (workdir.empty() ? std::string("") : workdir + "/") + "final_kmers";
        const double t0w = now_s();
        uint64_t total = 0;
        if (!write_final_kmers_merged(shards, W, out, &total)) fatal("cannot write %s", out.c_str());
        ph.write = now_s() - t0w;
        info("K-mer counting done. There are %llu kmers in total.", (unsigned long long)total);
        info("K-mer counting done, kmers saved to %s", out.c_str());
        ph.total = now_s() - t_start;
        ph.report("spades-kmercount");
        finish_process(ctxs.empty() ? nullptr : ctxs[0], 0);
    }
    bbk_ctx *ctx = nullptr;
    bbk_counter *counter = nullptr;
    // the context (HIP initialisation: 0.1-0.2 s) and the counter are created while the first block is being parsed;
    // the set is built in the final_kmers order (what CountAll(16, ..., merge=true) leaves on disk, :214-219)
    auto init = [&] {
        const double t0c = now_s();
        check(bbk_ctx_create((int)device, &ctx), "bbk_ctx_create");
        ph.ctx = now_s() - t0c;
        check(bbk_count_begin(ctx, K, BBK_BOTH_STRANDS | BBK_REFERENCE_ORDER, &counter), "bbk_count_begin");
    };
    stream_reads(ctx, files, (size_t)bufsize, threads ? (int)threads : default_threads(), ph, [&](bbk_reads *r) {
        check(bbk_count_push_reads(counter, r), "bbk_count_push_reads");
    }, init);
    double t0 = now_s();
    bbk_kmerset *set = nullptr;
    check(bbk_count_finish(counter, &set), "bbk_count_finish");
    ph.finish = now_s() - t0;
    // same line as KMerDiskCounter::Count (common/utils/kmer_mph/kmer_index_builder.hpp:260)
    info("K-mer counting done. There are %llu kmers in total.", (unsigned long long)bbk_kmerset_size(set));
    if (!workdir.empty()) mkdir(workdir.c_str(), 0755);
    const std::string out = (workdir.empty() ? std::string("") : workdir + "/") + "final_kmers";
    t0 = now_s();
    check(bbk_kmerset_write_final_kmers(ctx, set, out.c_str()), "bbk_kmerset_write_final_kmers");
    ph.write = now_s() - t0;
    info("K-mer counting done, kmers saved to %s", out.c_str());
    bbk_kmerset_free(set);
    ph.total = now_s() - t_start;
    ph.memory(ctx);
    ph.report("spades-kmercount");
    finish_process(ctx, 0);
}
