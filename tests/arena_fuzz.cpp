// host-only randomized test of the device allocator's arena bookkeeping (spades_for_blackbird_amd/csrc/arena.h)
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <vector>

#include "../spades_for_blackbird_amd/csrc/arena.h"

int main(int argc, char **argv) {
    const unsigned seed = argc > 1 ? (unsigned)atoi(argv[1]) : 1u;
    std::mt19937_64 rng(seed);
    const size_t G = 2, CH = 64;  // granule and chunk in test units
    bbk::ArenaIndex A;
    std::map<size_t, size_t> live;  // offset -> size
    size_t live_bytes = 0, max_mapped = 0;
    std::vector<size_t> retired;  // offsets of chunks that were unmapped: never handed out again
    auto check = [&]() {
        // free blocks: inside backed chunks, coalesced, both indices agree; live and free blocks tile the backed chunks
        // exactly; nothing lies in a retired chunk; chunk offsets ascend and stay below top
        size_t prev_end = (size_t)-1, fb = 0;
        for (auto &kv : A.free_off) {
            assert(kv.second > 0 && kv.first + kv.second <= A.top);
            assert(prev_end == (size_t)-1 || kv.first > prev_end);  // not adjacent: would have been merged
            prev_end = kv.first + kv.second;
            fb += kv.second;
        }
        assert(A.free_size.size() == A.free_off.size());
        for (auto &kv : A.free_size) assert(A.free_off.count(kv.second) && A.free_off[kv.second] == kv.first);
        assert(A.mapped == A.chunk_off.size() * CH);
        assert(fb + live_bytes == A.mapped);
        auto f = A.free_off.begin();
        auto l = live.begin();
        size_t prev_chunk = (size_t)-1;
        for (size_t c : A.chunk_off) {
            assert(prev_chunk == (size_t)-1 || c >= prev_chunk + CH);
            assert(c + CH <= A.top);
            prev_chunk = c;
        }
        // walk the backed space: consecutive chunks form runs; blocks may span chunks of a run, never a hole
        for (size_t i = 0; i < A.chunk_off.size();) {
            size_t j = i;
            while (j + 1 < A.chunk_off.size() && A.chunk_off[j + 1] == A.chunk_off[j] + CH) ++j;
            size_t cur = A.chunk_off[i];
            const size_t end = A.chunk_off[j] + CH;
            while (cur < end) {
                if (f != A.free_off.end() && f->first == cur) { cur += f->second; ++f; }
                else { assert(l != live.end() && l->first == cur); cur += l->second; ++l; }
            }
            assert(cur == end);
            i = j + 1;
        }
        assert(f == A.free_off.end() && l == live.end());
        for (size_t r : retired)
            for (size_t c : A.chunk_off) assert(c != r);
    };
    for (int step = 0; step < 200000; ++step) {
        const int op = (int)(rng() % 100);
        if (op < 50) {  // allocate
            size_t want = G * (1 + rng() % (rng() % 8 == 0 ? 300 : 20));
            size_t off;
            if (!A.take(want, &off)) {
                const size_t tail = A.free_tail();
                const size_t need = (want - tail + CH - 1) / CH;
                for (size_t i = 0; i < need; ++i) A.grown(CH);
                const bool ok = A.take(want, &off);
                assert(ok);
            }
            assert(off % G == 0 && off + want <= A.top);
            live[off] = want;
            live_bytes += want;
        } else if (op < 95) {  // free a random live block
            if (live.empty()) continue;
            auto it = live.lower_bound(rng() % (A.top + 1));
            if (it == live.end()) it = live.begin();
            A.add_free(it->first, it->second);
            live_bytes -= it->second;
            live.erase(it);
        } else {  // trim: unmap every chunk at the end that is free
            size_t at = 0;
            while (A.shrink_one(CH, &at)) retired.push_back(at);
            // what is left of the last chunk is in use somewhere, or nothing is backed
            assert(A.chunk_off.empty() || A.free_tail() < CH);
            if (retired.size() > 4096) retired.erase(retired.begin(), retired.begin() + 2048);
        }
        if (A.mapped > max_mapped) max_mapped = A.mapped;
        if (step % 5000 == 4999 && live.empty()) { /* an empty arena keeps its retired offsets retired */ assert(A.mapped == A.free_bytes()); }
        if (step % 64 == 0) check();
    }
    check();
    printf("ARENA-FUZZ-OK seed=%u max_mapped=%zu live=%zu\n", seed, max_mapped, live.size());
    return 0;
}
