// spades-gbuilder drop-in: same argv contract as the reference tool (projects/gbuilder/main.cpp:47-87)
//   <dataset yaml | fasta/fastq> <output> [-k <int=21>] [-c] [-t <int>] [-tmp-dir <dir>] [-b <bytes>]
//   one of --unitigs (default) | --fastg | --gfa | --spades   (+ --device <int>, --devices a,b,.. [--exchange rccl|copy], ours)
// and the same flow (:103-237): reads -> extension index -> unbranching paths + loops ->
// unitig FASTA or graph -> GFA, with every step behind the C ABI (include/bbk.h).
// -t = parser threads, -b = bytes of input text per block: the reads are streamed block by block through
// bbk_extindex_begin / push / finish (and, with -c, a (k+1)-mer counter), so host and device memory are bounded as in
// the reference (binary read chunks + bounded sort buffers); -tmp-dir is accepted (there are no temp files).
// --spades writes <output>.grseq + <output>.cvr (io::binary::BasicGraphIO::Save, :221-222).
#include <cstring>
#include <string>
#include <vector>

#include "common.hpp"
#include "multi.hpp"

using namespace bbkhost;

static void usage(const char *argv0) {
    printf("SYNOPSIS\n        %s <dataset description (in YAML) or input FASTA file> <output filename> [-k <value>] [-c]\n"
           "           [-t <value>] [-tmp-dir <dir>] [-b <value>] [--unitigs|--fastg|--gfa|--spades]\n\n"
           "OPTIONS\n"
           "        -k <value>  k-mer length to use\n"
           "        -c          infer coverage\n"
           "        -t <value>  # of threads to use\n"
           "        -tmp-dir <dir>\n                    scratch directory to use\n"
           "        -b <value>  sorting buffer size, per thread\n"
           "        --unitigs   produce unitigs (default)\n"
           "        --fastg     produce graph in FASTG format\n"
           "        --gfa       produce graph in GFA1 format\n"
           "        --spades    produce graph in SPAdes internal format\n"
           "        --device <value>  GPU to use (default 0)\n"
           "        --devices <a,b,...>  GPUs to use: extension-index records sharded by k-mer owner over one RCCL all-to-all,\n"
           "                    the shards gathered on the first device for the unitig stage (--exchange copy: peer copies)\n",
           argv0);
}

int main(int argc, char **argv) {
    unsigned k = 21, device = 0;
    unsigned long long threads = 0, bufsize = 536870912ull;  // projects/gbuilder/main.cpp:47-52
    bool coverage = false, bad = false;
    enum { UNITIGS, FASTG, GFA, SPADES } mode = UNITIGS;
    int modes_given = 0;
    std::vector<std::string> pos;
    std::string devices_arg, exchange_arg = "rccl";
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        unsigned long long v = 0;
        auto need = [&](unsigned long long *x) { return i + 1 < argc && parse_uint(argv[++i], x); };
        if (a == "-k") { if (need(&v)) k = (unsigned)v; else bad = true; }
        else if (a == "-c") coverage = true;
        else if (a == "-t") { if (need(&v)) threads = v; else bad = true; }
        else if (a == "-b") { if (need(&v)) bufsize = v; else bad = true; }
        else if (a == "--device") { if (need(&v)) device = (unsigned)v; else bad = true; }
        else if (a == "--devices") { if (i + 1 < argc) devices_arg = argv[++i]; else bad = true; }
        else if (a == "--exchange") { if (i + 1 < argc) exchange_arg = argv[++i]; else bad = true; }
        else if (a == "-tmp-dir") { if (i + 1 < argc) ++i; else bad = true; }
        else if (a == "--unitigs") { mode = UNITIGS; ++modes_given; }
        else if (a == "--fastg") { mode = FASTG; ++modes_given; }
        else if (a == "--gfa") { mode = GFA; ++modes_given; }
        else if (a == "--spades") { mode = SPADES; ++modes_given; }
        else if (!a.empty() && a[0] == '-' && a.size() > 1) bad = true;
        else pos.push_back(a);
    }
    if (bad || pos.size() != 2 || modes_given > 1) {  // projects/gbuilder/main.cpp:82-86
        usage(argv[0]);
        return 1;
    }
    const std::string file = pos[0], outfile = pos[1];

    info("Starting SPAdes standalone graph builder (MI355X, %s)", bbk_version());
    // projects/gbuilder/main.cpp:121-126
    if (k < 1) fatal("k-mer size %u is too low", k);
    if (k >= BBK_MAX_K) fatal("k-mer size %u is too high, recompile with larger SPADES_MAX_K option", k);
    if (k % 2 == 0) fatal("k-mer size must be odd");
    info("K-mer length set to %u", k);
    switch (mode) {
        case UNITIGS: info("Producing unitigs only"); break;
        case FASTG: info("Producing graph in FASTG format"); break;
        case GFA: info("Producing graph in GFA1 format"); break;
        case SPADES: info("Producing graph in SPAdes internal format"); break;
    }
    if (coverage && mode == UNITIGS) info("Note: -c has no effect on --unitigs output");

    // LoadDataset (:89-101)
    std::vector<std::string> files;
    if (ends_with(file, ".yaml")) {
        std::string err;
        if (!load_dataset_yaml(file, files, err)) fatal("%s", err.c_str());
    } else {
        FILE *f = fopen(file.c_str(), "rb");
        if (!f) fatal("Dataset description file: %s does not exist or is not a valid YAML file", file.c_str());
        fclose(f);
        files.push_back(file);
    }

    std::vector<int> devices;
    if (!devices_arg.empty() && !parse_devices(devices_arg, devices)) fatal("--devices: expected a comma-separated list of GPU indices");
    if (exchange_arg != "rccl" && exchange_arg != "copy") fatal("--exchange: rccl or copy");
    Phases ph;
    const double t_start = now_s();
    bbk_ctx *ctx = nullptr;
    const bool want_cov = coverage && mode != UNITIGS;
    bbk_extindex *ext = nullptr;
    bbk_kmerset *kp1 = nullptr;  // -c: ascending canonical (k+1)-mers with multiplicities
    double t0 = 0;
    if (!devices.empty()) {
        // ---- several devices: every rank reduces its blocks to (canonical k-mer, OR of mask bits) records, the records go
        //      to the k-mer's owner in one all-to-all, the owners' shards are gathered on rank 0 for the walk
        auto gf = create_group_async(devices, exchange_arg == "rccl" ? BBK_EXCHANGE_RCCL : BBK_EXCHANGE_COPY);
        const int n = (int)devices.size();
        info("Using %d device(s), %s exchange", n, exchange_arg.c_str());
        std::vector<bbk_counter *> xc((size_t)n, nullptr), cc((size_t)n, nullptr);
        std::vector<bbk_ctx *> ctxs;
        RankHooks hooks;
        hooks.init = [&](int r, bbk_ctx *c) {
            check(bbk_count_begin(c, k, BBK_CANONICAL | BBK_UNSORTED | BBK_WITH_MASKS, &xc[(size_t)r]), "bbk_count_begin");
            if (want_cov) check(bbk_count_begin(c, k + 1, BBK_CANONICAL | BBK_UNSORTED | BBK_WITH_COUNTS, &cc[(size_t)r]), "bbk_count_begin");
        };
        hooks.push = [&](int r, bbk_ctx *, bbk_reads *rd) {
            check(bbk_count_push_reads(xc[(size_t)r], rd), "bbk_count_push_reads");
            if (want_cov) check(bbk_count_push_reads(cc[(size_t)r], rd), "bbk_count_push_reads");
        };
        hooks.finish = [&](int r, bbk_ctx *c) {
            bbk_kmerset *local = nullptr;
            bbk_extindex *shard = nullptr, *full = nullptr;
            check(bbk_count_finish(xc[(size_t)r], &local), "bbk_count_finish");
            check(bbk_group_exchange_extindex(gf.get(), r, c, local, &shard), "bbk_group_exchange_extindex");
            bbk_kmerset_free(local);
            check(bbk_group_gather_extindex(gf.get(), r, c, shard, 0, &full), "bbk_group_gather_extindex");
            bbk_extindex_free(shard);
            if (r == 0) ext = full;
            if (want_cov) {
                bbk_kmerset *lc = nullptr, *sc = nullptr, *fc = nullptr;
                check(bbk_count_finish(cc[(size_t)r], &lc), "bbk_count_finish");
                check(bbk_group_exchange_kmers(gf.get(), r, c, lc, BBK_UNSORTED, &sc), "bbk_group_exchange_kmers");
                bbk_kmerset_free(lc);
                check(bbk_group_gather_kmers(gf.get(), r, c, sc, 0, &fc), "bbk_group_gather_kmers");
                bbk_kmerset_free(sc);
                if (r == 0) kp1 = fc;
            }
        };
        const uint64_t n_reads = run_ranks(devices, files, (size_t)bufsize, threads ? (int)threads : default_threads(), ph, hooks, ctxs);
        info("Used %llu reads", (unsigned long long)n_reads);
        ctx = ctxs[0];
        t0 = now_s();
    } else {
    // Step 1: build extension index (:169-172), block by block; with -c the canonical (k+1)-mer multiplicities
    // (CoverageHashMapBuilder, :200-211) are counted from the same blocks.  The context (HIP initialisation:
    // 0.1-0.2 s) and the two accumulators are created while the first block is being parsed.
    bbk_extbuilder *xb = nullptr;
    bbk_counter *covc = nullptr;
    auto init = [&] {
        const double t0c = now_s();
        check(bbk_ctx_create((int)device, &ctx), "bbk_ctx_create");
        ph.ctx = now_s() - t0c;
        check(bbk_extindex_begin(ctx, k, &xb), "bbk_extindex_begin");
        if (want_cov) check(bbk_count_begin(ctx, k + 1, BBK_CANONICAL | BBK_WITH_COUNTS, &covc), "bbk_count_begin");
    };
    const uint64_t n_reads =
        stream_reads(ctx, files, (size_t)bufsize, threads ? (int)threads : default_threads(), ph, [&](bbk_reads *r) {
            check(bbk_extindex_push_reads(xb, r), "bbk_extindex_push_reads");
            if (covc) check(bbk_count_push_reads(covc, r), "bbk_count_push_reads");
        }, init);
    info("Used %llu reads", (unsigned long long)n_reads);
    t0 = now_s();
    check(bbk_extindex_finish(xb, &ext), "bbk_extindex_finish");
    if (covc) check(bbk_count_finish(covc, &kp1), "bbk_count_finish");
    }
    info("K-mer counting done. There are %llu kmers in total.", (unsigned long long)bbk_extindex_size(ext));
    info("Building k-mer extensions from k+1-mers finished.");

    // Step 2: extract unbranching paths (:175-181)
    bbk_unitigs *u = nullptr;
    check(bbk_unitigs_build(ctx, ext, &u), "bbk_unitigs_build");
    info("Extracting unbranching paths finished. %llu sequences extracted",
         (unsigned long long)(bbk_unitigs_count(u) - bbk_unitigs_loops(u)));
    info("Collecting perfect loops finished. %llu loops collected", (unsigned long long)bbk_unitigs_loops(u));

    if (mode == UNITIGS) {
        ph.finish = now_s() - t0;
        info("Saving unitigs to %s", outfile.c_str());
        t0 = now_s();
        check(bbk_unitigs_write_fasta(ctx, u, outfile.c_str()), "bbk_unitigs_write_fasta");
        ph.write = now_s() - t0;
    } else {
        info("Total %llu edges to create", (unsigned long long)(2 * bbk_unitigs_count(u)));
        info("Total %llu vertices to create", (unsigned long long)bbk_unitigs_vertices(u));
        if (kp1) {  // Step 4: infer coverage (projects/gbuilder/main.cpp:200-211)
            info("Filling coverage index");
            check(bbk_unitigs_add_coverage_counts(ctx, u, kp1), "bbk_unitigs_add_coverage_counts");
            bbk_kmerset_free(kp1);
            info("Filling coverage and flanking coverage from PHM");
        }
        ph.finish = now_s() - t0;
        info("Saving graph to %s", outfile.c_str());
        t0 = now_s();
        if (mode == GFA) check(bbk_unitigs_write_gfa(ctx, u, outfile.c_str()), "bbk_unitigs_write_gfa");
        else if (mode == FASTG) check(bbk_unitigs_write_fastg(ctx, u, outfile.c_str()), "bbk_unitigs_write_fastg");
        else check(bbk_unitigs_write_spades(ctx, u, outfile.c_str()), "bbk_unitigs_write_spades");  // <out>.grseq + .cvr
        ph.write = now_s() - t0;
    }
    bbk_unitigs_free(u);
    bbk_extindex_free(ext);
    ph.total = now_s() - t_start;
    ph.memory(ctx);
    ph.report("spades-gbuilder");
    info("SPAdes standalone graph builder finished");
    finish_process(ctx, 0);
}
