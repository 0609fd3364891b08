// host-only randomized test of the writer that merges the ranks' shards into one final_kmers file
// (spades_for_blackbird_amd/host/multi.hpp: write_final_kmers_merged; KMerDiskStorage::merge analogue,
// common/utils/kmer_mph/kmer_index_builder.hpp:168-181): bucket b of the file = the N-way merge of the shards' runs
// of bucket b, records in word order (adt/array_vector.hpp:114-123)
#include <algorithm>
#include <cassert>
#include <cstdio>
#include <random>
#include <set>
#include <vector>

#include "../spades_for_blackbird_amd/host/multi.hpp"

using namespace bbkhost;

int main(int argc, char **argv) {
    const unsigned seed = argc > 1 ? (unsigned)atoi(argv[1]) : 1u;
    const char *path = argc > 2 ? argv[2] : "/tmp/bbk_merge_fuzz.bin";
    std::mt19937_64 rng(seed);
    for (int round = 0; round < 40; ++round) {
        const unsigned W = 1 + (unsigned)(rng() % 4);
        const size_t N = 1 + (size_t)(rng() % 8);
        const size_t total = (round % 5 == 0) ? (size_t)(rng() % 5) : (size_t)(rng() % 20000);
        typedef std::vector<uint64_t> Rec;
        std::set<Rec> uniq;
        while (uniq.size() < total) {
            Rec r(W);
            for (unsigned w = 0; w < W; ++w) r[w] = (rng() % 3 == 0) ? (rng() % 4) : rng();  // many equal leading words
            uniq.insert(r);
        }
        // bucket and owner of every record; some buckets and some shards stay empty
        std::vector<std::vector<std::vector<Rec>>> part(N, std::vector<std::vector<Rec>>(16));
        std::vector<std::vector<Rec>> want(16);
        const unsigned nb_used = 1 + (unsigned)(rng() % 16);
        const size_t n_used = 1 + (size_t)(rng() % N);
        for (const Rec &r : uniq) {
            const unsigned b = (unsigned)(rng() % nb_used);
            part[(size_t)(rng() % n_used)][b].push_back(r);
            want[b].push_back(r);
        }
        std::vector<ShardOnHost> sh(N);
        for (size_t r = 0; r < N; ++r) {
            uint64_t o = 0;
            for (int b = 0; b < 16; ++b) {
                std::sort(part[r][(size_t)b].begin(), part[r][(size_t)b].end());
                sh[r].off[b] = o;
                for (const Rec &x : part[r][(size_t)b]) sh[r].keys.insert(sh[r].keys.end(), x.begin(), x.end());
                o += part[r][(size_t)b].size();
            }
            sh[r].off[16] = o;
        }
        uint64_t n_total = 0;
        const bool ok = write_final_kmers_merged(sh, W, path, &n_total);
        assert(ok && n_total == total);
        std::vector<uint64_t> expect;
        for (int b = 0; b < 16; ++b) {
            std::sort(want[(size_t)b].begin(), want[(size_t)b].end());
            for (const Rec &x : want[(size_t)b]) expect.insert(expect.end(), x.begin(), x.end());
        }
        std::vector<uint64_t> got(expect.size());
        FILE *f = fopen(path, "rb");
        assert(f);
        const size_t rd = got.empty() ? 0 : fread(got.data(), 8, got.size(), f);
        assert(rd == got.size());
        assert(fgetc(f) == EOF);
        fclose(f);
        assert(got == expect);
    }
    remove(path);
    printf("MERGE-FUZZ-OK seed=%u\n", seed);
    return 0;
}
