// bbk-fastx-dump [--fast <threads> <block_bytes> | --serial-lv] <file>...: test helper for the host-side
// FASTA/FASTQ(.gz) readers (no GPU needed).
//   default      every parsed read sequence of the serial reader (fastx.hpp) on its own line
//   --serial-lv  the same reads after the LongestValid rule (longest run of ACGTacgt, upper-cased)
//   --fast       the parallel block parser (ingest.hpp): its packed reads decoded back to ACGT -- must print
//                exactly what --serial-lv prints, for any thread count and block size
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "fastx.hpp"
#include "ingest.hpp"

int main(int argc, char **argv) {
    int first = 1;
    bool fast = false, lv = false;
    int threads = 1;
    size_t block = 1u << 20;
    if (argc > 1 && !strcmp(argv[1], "--fast")) {
        if (argc < 4) return 2;
        fast = true;
        threads = atoi(argv[2]);
        block = (size_t)strtoull(argv[3], nullptr, 10);
        first = 4;
    } else if (argc > 1 && !strcmp(argv[1], "--serial-lv")) {
        lv = true;
        first = 2;
    }
    if (fast) {
        std::vector<std::string> files(argv + first, argv + argc);
        bbkhost::Ingest ing(files, block, threads);
        ing.min_block = 1;  // tests use blocks of a few bytes to exercise the carry-over
        bbkhost::PackedReads b;
        std::string err;
        while (ing.next(b, err)) {
            size_t w = 0;
            for (uint64_t r = 0; r < b.size(); ++r) {
                const uint32_t L = b.len[r];
                for (uint32_t p = 0; p < L; ++p) fputc("ACGT"[(b.words[w + (p >> 5)] >> ((p & 31) << 1)) & 3], stdout);
                fputc('\n', stdout);
                w += (L + 31) / 32;
            }
            if (w != b.words.size()) {
                fprintf(stderr, "word count mismatch\n");
                return 3;
            }
        }
        if (!err.empty()) {
            fprintf(stderr, "%s\n", err.c_str());
            return 2;
        }
        fprintf(stderr, "fallback_blocks=%llu\n", (unsigned long long)ing.fallback_blocks());
        return 0;
    }
    for (int i = first; i < argc; ++i) {
        bbkhost::FastxReader rd(argv[i]);
        if (!rd.is_open()) {
            fprintf(stderr, "cannot open %s\n", argv[i]);
            return 2;
        }
        bbkhost::ReadBatch b;
        while (rd.read(b, 1000, ~0ull) > 0) {
            for (uint64_t r = 0; r < b.size(); ++r) {
                const char *s = b.bases.data() + b.offsets[r];
                size_t n = b.offsets[r + 1] - b.offsets[r], f = 0, t = n;
                if (lv) bbkhost::detail::longest_valid(s, n, &f, &t);
                fwrite(s + f, 1, t - f, stdout);
                fputc('\n', stdout);
            }
            b.clear();
        }
    }
    return 0;
}
