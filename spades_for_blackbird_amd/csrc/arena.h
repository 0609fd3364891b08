// arena.h -- bookkeeping of the device allocator's arena (primitives.hip): a range [0, mapped) of a reserved virtual
// range is backed by physical chunks; inside it a best-fit free list with coalescing.  Pure host data structure (no HIP
// calls: mapping and unmapping are the caller's), so that it can be exercised by a host-only randomized test
// (tests/test_arena.py).
#pragma once

#include <cstddef>
#include <iterator>
#include <map>

namespace bbk {

struct ArenaIndex {
    size_t mapped = 0;                        // [0, mapped) is backed
    std::map<size_t, size_t> free_off;        // offset -> size, coalesced, inside [0, mapped)
    std::multimap<size_t, size_t> free_size;  // size -> offset

    void erase_size(size_t size, size_t off) {
        auto r = free_size.equal_range(size);
        for (auto it = r.first; it != r.second; ++it)
            if (it->second == off) {
                free_size.erase(it);
                return;
            }
    }
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
    // best fit; false when nothing mapped is large enough
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
    // bytes of the free block that ends exactly at `mapped` (0 if the last mapped byte is in use)
    size_t free_tail() const {
        if (free_off.empty()) return 0;
        auto last = std::prev(free_off.end());
        return last->first + last->second == mapped ? last->second : 0;
    }
    // a chunk of `chunk` bytes has been mapped at [mapped, mapped + chunk)
    void grown(size_t chunk) {
        add_free(mapped, chunk);
        mapped += chunk;
    }
    // true (and the index updated) if the LAST chunk [mapped - chunk, mapped) is entirely free and can be unmapped
    bool shrink_one(size_t chunk) {
        if (mapped < chunk) return false;
        const size_t lo = mapped - chunk;
        auto it = free_off.upper_bound(lo);
        if (it == free_off.begin()) return false;
        --it;  // the free block that starts at or before lo
        if (it->first + it->second != mapped) return false;  // (it->first <= lo holds by construction)
        const size_t off = it->first, size = it->second;
        erase_size(size, off);
        free_off.erase(it);
        if (lo > off) {
            free_off[off] = lo - off;
            free_size.emplace(lo - off, off);
        }
        mapped = lo;
        return true;
    }
    size_t free_bytes() const {
        size_t t = 0;
        for (auto &kv : free_off) t += kv.second;
        return t;
    }
};

}  // namespace bbk
