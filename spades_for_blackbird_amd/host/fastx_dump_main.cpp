// bbk-fastx-dump <file>...: prints every parsed read sequence on its own line (test helper for the
// host-side FASTA/FASTQ(.gz) reader; no GPU needed).
#include <cstdio>

#include "fastx.hpp"

int main(int argc, char **argv) {
    for (int i = 1; i < argc; ++i) {
        bbkhost::FastxReader rd(argv[i]);
        if (!rd.is_open()) {
            fprintf(stderr, "cannot open %s\n", argv[i]);
            return 2;
        }
        bbkhost::ReadBatch b;
        while (rd.read(b, 1000, ~0ull) > 0) {
            for (uint64_t r = 0; r < b.size(); ++r) {
                fwrite(b.bases.data() + b.offsets[r], 1, b.offsets[r + 1] - b.offsets[r], stdout);
                fputc('\n', stdout);
            }
            b.clear();
        }
    }
    return 0;
}
