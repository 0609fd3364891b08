// arena.h -- bookkeeping of the device allocator's arena (primitives.hip): physical chunks are mapped one after the
// other into a reserved virtual range; inside the backed part a best-fit free list with coalescing.  Pure host data
// structure (no HIP calls: mapping and unmapping are the caller's), so that it can be exercised by a host-only
// randomized test (tests/test_arena.py).
//
// An offset that was unmapped is NEVER mapped again: on this platform accesses to a virtual address that is unmapped
// and then given new physical memory can still reach the old memory (tools/probes/vmm_remap_copy_probe.hip: a kernel
// that fills and re-reads a re-mapped range finds millions of wrong words; the same range at fresh addresses is
// clean), so `top` -- where the next chunk goes -- only ever grows and a retired chunk leaves a hole behind.
#pragma once

#include <cstddef>
#include <iterator>
#include <map>
#include <vector>

namespace bbk {

struct ArenaIndex {
    size_t top = 0;                           // next chunk is mapped at this offset; never decreases
    size_t mapped = 0;                        // bytes backed now (sum of the chunks in chunk_off)
    std::vector<size_t> chunk_off;            // offsets of the backed chunks, ascending
    size_t chunk_size = 0;                    // all chunks of an arena have one size (set by the first grown())
    std::map<size_t, size_t> free_off;        // offset -> size, coalesced, inside backed chunks
    std::multimap<size_t, size_t> free_size;  // size -> offset

    void erase_size(size_t size, size_t off) {
        auto r = free_size.equal_range(size);
        for (auto it = r.first; it != r.second; ++it)
            if (it->second == off) {
                free_size.erase(it);
                return;
            }
    }
    // blocks on both sides of a hole are never adjacent in offset terms (the hole is at least one chunk wide), so
    // coalescing by offsets cannot join memory across a hole
    void add_free(size_t off, size_t size) {
        auto nx = free_off.lower_bound(off);
        if (nx != free_off.end() && off + size == nx->first) {  // merge with the block after
            size += nx->second;
            erase_size(nx->second, nx->first);
            nx = free_off.erase(nx);
        }
        if (nx != free_off.begin()) {
            auto pv = std::prev(nx);
            if (pv->first + pv->second == off) {  // merge with the block before
                off = pv->first;
                size += pv->second;
                erase_size(pv->second, pv->first);
                free_off.erase(pv);
            }
        }
        free_off[off] = size;
        free_size.emplace(size, off);
    }
    // best fit; false when nothing backed is large enough
    bool take(size_t want, size_t *off) {
        auto it = free_size.lower_bound(want);
        if (it == free_size.end()) return false;
        const size_t size = it->first, o = it->second;
        free_size.erase(it);
        free_off.erase(o);
        if (size > want) add_free(o + want, size - want);
        *off = o;
        return true;
    }
    // bytes of the free block that ends exactly at `top` (0 if the last backed byte is in use, or if the chunk below
    // `top` has been retired): a request may start there and continue into chunks mapped at `top`
    size_t free_tail() const {
        if (free_off.empty() || chunk_off.empty() || chunk_off.back() + chunk_size != top) return 0;
        auto last = std::prev(free_off.end());
        return last->first + last->second == top ? last->second : 0;
    }
    // a chunk of `chunk` bytes has been mapped at [top, top + chunk)
    void grown(size_t chunk) {
        chunk_size = chunk;
        chunk_off.push_back(top);
        add_free(top, chunk);
        top += chunk;
        mapped += chunk;
    }
    // true (and the index updated) if the LAST backed chunk is entirely free and can be unmapped; *off = where it was.
    // Its offsets are retired: `top` stays.
    bool shrink_one(size_t chunk, size_t *off = nullptr) {
        if (chunk_off.empty()) return false;
        const size_t lo = chunk_off.back();
        auto it = free_off.upper_bound(lo);
        if (it == free_off.begin()) return false;
        --it;  // the free block that starts at or before lo
        if (it->first + it->second != lo + chunk) return false;  // nothing is backed above the last chunk
        const size_t boff = it->first, bsize = it->second;
        erase_size(bsize, boff);
        free_off.erase(it);
        if (lo > boff) {
            free_off[boff] = lo - boff;
            free_size.emplace(lo - boff, boff);
        }
        chunk_off.pop_back();
        mapped -= chunk;
        if (off) *off = lo;
        return true;
    }
    size_t free_bytes() const {
        size_t t = 0;
        for (auto &kv : free_off) t += kv.second;
        return t;
    }
};

}  // namespace bbk
