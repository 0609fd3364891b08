// spades-read-filter drop-in: same argv contract and the same output layout as the reference tool
// (projects/kmercount/read_filter.cpp:43-71 for the flags, :120-251 for the flow):
//   [-k <int=21>] [-c <int=2>] -d <yaml> [-t <int>] [-o <dir=.>] [--drop-names] [--drop-quality] [-h]   (+ --device)
// Reads whose median k-mer multiplicity is <= -c are dropped (a pair is kept when either mate passes).  The reference
// estimates multiplicities with a counting quotient filter over a strand-symmetric rolling hash (utils/kmer_counting.hpp,
// adt/cqf.hpp: approximate, capped at -c + 1); here they are the EXACT multiplicities of the canonical k-mers, counted
// on the device by the streaming counter, and the per-read median test runs on the device too
// (bbk_reads_median_filter = io::CoverageFilter / CountMedianMlt, io/reads/coverage_filtering_read_wrapper.hpp:22-94).
// Output: per library i (1-based) <i>.1.fastq / <i>.2.fastq (paired), <i>.s.fastq (single), <i>.m.fastq (merged) --
// .fasta with --drop-names / --drop-quality -- and dataset.yaml describing them, as the reference writes
// (read_filter.cpp:165-244; records as FastqWriter / FastaWriter, io/reads/osequencestream.hpp:118-133).
#include <sys/stat.h>

#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "common.hpp"

using namespace bbkhost;

static void usage(const char *argv0) {
    printf("SYNOPSIS\n        %s [-k <value>] [-c <value>] -d <yaml> [-t <value>] [-o <dir>] [--drop-names] [--drop-quality] [-h]\n\n"
           "OPTIONS\n"
           "        -k, --kmer <value>      K-mer length\n"
           "        -c, --cov <value>       Median kmer count threshold (read pairs, s.t. kmer count median for BOTH reads LESS\n"
           "                                OR EQUAL to this value will be ignored)\n"
           "        -d, --dataset <yaml>    Dataset description (in YAML)\n"
           "        -t, --threads <value>   # of threads to use\n"
           "        -o, --outdir <dir>      Output directory to use\n"
           "        --drop-names            Drop read names and quality (makes everything faster)\n"
           "        --drop-quality          Drop read quality (makes everything faster)\n"
           "        -h, --help              Show help\n"
           "        --device <value>        GPU to use (default 0)\n\n"
           "DESCRIPTION\n         Kmer count read filter (MI355X)\n",
           argv0);
}

struct Rec {
    std::string name, seq, qual;
};

// FastqWriter / FastaWriter (io/reads/osequencestream.hpp:118-133)
static void write_rec(FILE *f, const Rec &r, bool fasta, bool drop_names) {
    if (fasta) {
        fprintf(f, ">%s\n", drop_names ? "" : r.name.c_str());
        for (size_t cur = 0; cur < r.seq.size(); cur += 60) {
            fwrite(r.seq.data() + cur, 1, std::min<size_t>(60, r.seq.size() - cur), f);
            fputc('\n', f);
        }
    } else {
        fprintf(f, "@%s\n%s\n+\n%s\n", r.name.c_str(), r.seq.c_str(), r.qual.c_str());
    }
}

struct FilterCtx {
    bbk_ctx *ctx;
    const bbk_kmerset *counts;
    unsigned threshold;
    uint64_t processed = 0, retained = 0;
    // keep flags of a batch of sequences (LongestValid is applied by the upload, as filter_reads does per read, :103-106)
    void flags(const std::vector<Rec> &recs, std::vector<uint8_t> &keep) {
        std::string bases;
        std::vector<uint64_t> off(1, 0);
        for (const Rec &r : recs) {
            bases += r.seq;
            off.push_back(bases.size());
        }
        keep.assign(recs.size(), 0);
        if (recs.empty()) return;
        bbk_reads *rd = nullptr;
        check(bbk_reads_from_ascii(ctx, bases.data(), off.data(), recs.size(), &rd), "bbk_reads_from_ascii");
        uint64_t kept = 0;
        check(bbk_reads_median_filter(ctx, rd, counts, threshold, keep.data(), &kept), "bbk_reads_median_filter");
        bbk_reads_free(rd);
    }
};

static const size_t kBatch = 1u << 20;  // FILTER_READS_BUFF_SIZE (read_filter.cpp:163)

// single-ended stream: files one after the other
static void filter_single(FilterCtx &F, const std::vector<std::string> &files, const std::string &out, bool fasta,
                          bool drop_names) {
    FILE *fo = fopen(out.c_str(), "wb");
    if (!fo) fatal("Cannot open %s for writing", out.c_str());
    std::vector<Rec> recs;
    std::vector<uint8_t> keep;
    for (const std::string &f : files) {
        FastxReader rd(f);
        if (!rd.is_open()) fatal("Cannot open %s", f.c_str());
        bool more = true;
        while (more) {
            recs.clear();
            Rec r;
            while (recs.size() < kBatch && (more = rd.next_record(r.name, r.seq, r.qual))) recs.push_back(r);
            F.flags(recs, keep);
            for (size_t i = 0; i < recs.size(); ++i)
                if (keep[i]) {
                    write_rec(fo, recs[i], fasta, drop_names);
                    ++F.retained;
                }
            F.processed += recs.size();
        }
    }
    fclose(fo);
}

// paired stream: left[i] / right[i] in lockstep, or interlaced files (mates alternate); a pair stays when either mate
// passes (CoverageFilter<PairedRead>, coverage_filtering_read_wrapper.hpp:78-94)
static void filter_paired(FilterCtx &F, const std::vector<std::string> &left, const std::vector<std::string> &right,
                          const std::vector<std::string> &interlaced, const std::string &out1, const std::string &out2,
                          bool fasta, bool drop_names) {
    FILE *f1 = fopen(out1.c_str(), "wb"), *f2 = fopen(out2.c_str(), "wb");
    if (!f1 || !f2) fatal("Cannot open %s / %s for writing", out1.c_str(), out2.c_str());
    std::vector<Rec> a, b;
    std::vector<uint8_t> ka, kb;
    auto flush = [&]() {
        F.flags(a, ka);
        F.flags(b, kb);
        for (size_t i = 0; i < a.size(); ++i)
            if (ka[i] || kb[i]) {
                write_rec(f1, a[i], fasta, drop_names);
                write_rec(f2, b[i], fasta, drop_names);
                ++F.retained;
            }
        F.processed += a.size();
        a.clear();
        b.clear();
    };
    if (left.size() != right.size()) fatal("Dataset: %zu left but %zu right read files", left.size(), right.size());
    for (size_t i = 0; i < left.size(); ++i) {
        FastxReader r1(left[i]), r2(right[i]);
        if (!r1.is_open()) fatal("Cannot open %s", left[i].c_str());
        if (!r2.is_open()) fatal("Cannot open %s", right[i].c_str());
        Rec x, y;
        // a pair needs both mates: the stream ends with the shorter file
        while (r1.next_record(x.name, x.seq, x.qual) && r2.next_record(y.name, y.seq, y.qual)) {
            a.push_back(x);
            b.push_back(y);
            if (a.size() >= kBatch) flush();
        }
        flush();
    }
    for (const std::string &f : interlaced) {
        FastxReader r(f);
        if (!r.is_open()) fatal("Cannot open %s", f.c_str());
        Rec x, y;
        while (r.next_record(x.name, x.seq, x.qual) && r.next_record(y.name, y.seq, y.qual)) {
            a.push_back(x);
            b.push_back(y);
            if (a.size() >= kBatch) flush();
        }
        flush();
    }
    fclose(f1);
    fclose(f2);
}

int main(int argc, char **argv) {
    unsigned k = 21, thr = 2, device = 0;
    unsigned long long threads = 0;
    std::string dataset, outdir = ".";
    bool drop_names = false, drop_quality = false, help = false, bad = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        unsigned long long v = 0;
        auto need = [&](unsigned long long *x) { return i + 1 < argc && parse_uint(argv[++i], x); };
        if (a == "-k" || a == "--kmer") { if (need(&v)) k = (unsigned)v; else bad = true; }
        else if (a == "-c" || a == "--cov") { if (need(&v)) thr = (unsigned)v; else bad = true; }
        else if (a == "-t" || a == "--threads") { if (need(&v)) threads = v; else bad = true; }
        else if (a == "--device") { if (need(&v)) device = (unsigned)v; else bad = true; }
        else if (a == "-d" || a == "--dataset") { if (i + 1 < argc) dataset = argv[++i]; else bad = true; }
        else if (a == "-o" || a == "--outdir") { if (i + 1 < argc) outdir = argv[++i]; else bad = true; }
        else if (a == "--drop-names") drop_names = true;
        else if (a == "--drop-quality") drop_quality = true;
        else if (a == "-h" || a == "--help") help = true;
        else bad = true;
    }
    if (bad || help || dataset.empty()) {  // read_filter.cpp:62-70: -d is required
        usage(argv[0]);
        return help ? 0 : 1;
    }
    if (k < 1 || k >= BBK_MAX_K) fatal("k-mer size %u is out of range [1, %d)", k, BBK_MAX_K);
    info("Starting kmer count based read filtering (MI355X, %s)", bbk_version());
    info("K-mer length set to %u", k);
    const int nthreads = threads ? (int)threads : default_threads();
    info("# of threads to use: %d", nthreads);

    std::vector<DatasetLib> libs;
    std::string err;
    if (!load_dataset_libs(dataset, libs, err)) fatal("%s", err.c_str());
    mkdir(outdir.c_str(), 0755);

    bbk_ctx *ctx = nullptr;
    check(bbk_ctx_create((int)device, &ctx), "bbk_ctx_create");

    // exact multiplicities of the canonical k-mers of every read of every library (the reference: cardinality estimate
    // + CQF fill over single_binary_readers_for_libs(..., followed_by_rc=false, including_paired=true), :139-152)
    info("Estimating kmer cardinality");
    std::vector<std::string> all;
    for (const DatasetLib &l : libs)
        for (int j = 0; j < 5; ++j)
            for (const std::string &f : l.v[j]) all.push_back(f);
    bbk_counter *counter = nullptr;
    check(bbk_count_begin(ctx, k, BBK_CANONICAL | BBK_WITH_COUNTS, &counter), "bbk_count_begin");
    info("Filling kmer coverage");
    Phases ph;
    stream_reads(ctx, all, 512u << 20, nthreads, ph,
                 [&](bbk_reads *r) { check(bbk_count_push_reads(counter, r), "bbk_count_push_reads"); });
    bbk_kmerset *counts = nullptr;
    check(bbk_count_finish(counter, &counts), "bbk_count_finish");
    info("Kmer coverage filled");

    const bool fasta = drop_names || drop_quality;  // read_filter.cpp:181,205,224
    const char *ext = fasta ? "fasta" : "fastq";
    std::vector<DatasetLib> outlibs;
    for (size_t i = 0; i < libs.size(); ++i) {
        info("Filtering library %zu", i);
        FilterCtx F{ctx, counts, thr + 1};
        DatasetLib outl;
        outl.type = libs[i].type;
        outl.orientation = libs[i].orientation;
        const std::string id = std::to_string(i + 1);
        const DatasetLib &L = libs[i];
        if (!L.v[LIB_LEFT].empty() || !L.v[LIB_INTERLACED].empty()) {
            const std::string l = id + ".1." + ext, r = id + ".2." + ext;
            filter_paired(F, L.v[LIB_LEFT], L.v[LIB_RIGHT], L.v[LIB_INTERLACED], outdir + "/" + l, outdir + "/" + r, fasta,
                          drop_names);
            outl.v[LIB_LEFT].push_back(l);
            outl.v[LIB_RIGHT].push_back(r);
        }
        if (!L.v[LIB_SINGLE].empty()) {
            const std::string s = id + ".s." + ext;
            filter_single(F, L.v[LIB_SINGLE], outdir + "/" + s, fasta, drop_names);
            outl.v[LIB_SINGLE].push_back(s);
        }
        if (!L.v[LIB_MERGED].empty()) {
            const std::string m = id + ".m." + ext;
            filter_single(F, L.v[LIB_MERGED], outdir + "/" + m, fasta, drop_names);
            outl.v[LIB_MERGED].push_back(m);
        }
        info("Total %llu reads processed, %llu reads left after filtering", (unsigned long long)F.processed,
             (unsigned long long)F.retained);
        outlibs.push_back(outl);
    }
    info("Filtering finished");
    const std::string fname = outdir + "/dataset.yaml";
    info("Saving filtered dataset description to %s", fname.c_str());
    if (!save_dataset_yaml(fname, outlibs)) fatal("Cannot write %s", fname.c_str());
    bbk_kmerset_free(counts);
    finish_process(ctx, 0);
}
