// primitives.hip -- device-wide building blocks for the k-mer engine on gfx950:
//   * exclusive scan (u64),
//   * stable LSD radix sort of fixed-width k-mer records (+ optional u32 payload) with
//     LDS-staged digit histograms, wavefront ballot match-any ranking (64-wide) and an LDS
//     reorder so that global stores leave the CU as contiguous per-digit runs,
//   * unique / reduce-by-key over a sorted record array.
// They replace, on the device, what the reference does with libcxx::sort + std::unique per
// bucket (common/utils/kmer_mph/kmer_splitter.hpp:120-167) and the loser-tree merge
// (kmer_index_builder.hpp:281-365).  Integer/HBM-bound work: no MFMA.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "arena.h"
#include "bbk_internal.h"
#include "kmer_ops.h"

namespace bbk {

static thread_local char g_err[1024] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char *get_error() { return g_err; }

// ---- device allocator --------------------------------------------------------------------------
// hipMalloc of fresh memory costs ~30 ms/GiB on this platform (page tables, not bandwidth): a configs[2]-size call
// allocates hundreds of GB of working buffers whose sizes never repeat exactly, and a block cache that matches sizes
// either misses or fills the device and has to be trimmed -- allocation was 70 % of the wall time there.  The allocator
// is therefore an ARENA per (device, host thread): one reserved virtual range (hipMemAddressReserve), physical memory
// mapped into it in 1 GiB chunks as the high-water mark grows (hipMemCreate + hipMemMap), and a best-fit free list
// with coalescing inside the range.  Physical memory is paid for once; every later request of any size is carved out
// of what is already mapped.  (Fallback when the virtual-memory API is unavailable: the size-matching block cache.)
//
// bbk_ctx_trim / context destruction unmap and release the chunks at the end of the arena that are entirely free.
// A virtual address that was unmapped is never mapped again (arena.h: ArenaIndex::top only grows): on this platform a
// range that is unmapped and then given NEW physical memory is not coherent afterwards -- a kernel that fills and
// re-reads it finds words of the wrong chunk, a small host->device copy does not arrive where the next kernel reads
// (tools/probes/vmm_remap_copy_probe.hip, profiles/r03/vmm_remap_probe.log; the round-2 "abort after trim").  New
// chunks at fresh addresses are clean.  When the reserved range is used up the arena gets a new GENERATION (another
// reservation); an old generation only takes its blocks back and disappears when the last one has returned.
//
// Keys are (device, host thread).  The ABI's contract is one context per GPU per host thread, all of a context's work
// is issued on its one stream, and every entry point returns with that stream idle or with the released blocks' last
// use already queued on it -- so inside one key, handing released memory to the next request is stream-ordered.
// Memory never crosses devices nor threads.  bbk_ctx_set_stream drains the old stream first.
namespace {
struct PoolKey {
    int device;
    std::thread::id thread;
    bool operator<(const PoolKey &o) const { return device != o.device ? device < o.device : thread < o.thread; }
};
constexpr size_t kPoolGranule = 2ull << 20;   // request sizes are rounded to 2 MiB
constexpr size_t kArenaChunk = 1ull << 30;    // physical memory is mapped in 1 GiB chunks

struct Arena : ArenaIndex {  // bookkeeping in arena.h (host-only, fuzzed by tests/test_arena.py)
    char *base = nullptr;
    size_t va_bytes = 0;  // size of the reservation
    std::vector<hipMemGenericAllocationHandle_t> chunks;  // parallel to chunk_off
    bool contains(const void *p) const { return base && (const char *)p >= base && (const char *)p < base + va_bytes; }
};
struct ArenaGens {
    std::vector<std::unique_ptr<Arena>> gens;  // the last one serves requests
};

struct Pool {
    std::mutex mu;
    int vmm = -1;  // -1 undecided, 0 block cache, 1 arenas
    std::map<PoolKey, ArenaGens> arenas;
    std::map<PoolKey, std::multimap<size_t, void *>> free_blocks;  // block-cache fallback: key -> size -> block
    // statistics (BBK_VERBOSE prints them when a context is destroyed)
    double malloc_s = 0, free_s = 0;
    uint64_t mallocs = 0, frees = 0, hits = 0, generations = 0;
    double malloc_bytes = 0;
};
inline double wall_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
Pool &pool() {
    static Pool p;
    return p;
}
int current_device() {
    int d = 0;
    (void)hipGetDevice(&d);
    return d;
}

// reservation of a new generation: as large as the platform grants (BBK_ARENA_VA_GB: tests make it small)
std::unique_ptr<Arena> arena_reserve() {
    auto A = std::make_unique<Arena>();
    const char *e = getenv("BBK_ARENA_VA_GB");
    // 4 TiB: hundreds of trim cycles of a 288 GB device before a new generation is needed, and eight rank threads of one
    // process (spades-kmercount --devices) together stay far below the 128 TiB of user address space
    const size_t first = e ? (size_t)strtoull(e, nullptr, 10) << 30 : (4ull << 40);
    for (size_t sz = first; sz >= (e ? first : (1ull << 38)); sz >>= 1) {
        void *p = nullptr;
        if (hipMemAddressReserve(&p, sz, 0, nullptr, 0) == hipSuccess && p) {
            A->base = (char *)p;
            A->va_bytes = sz;
            ++pool().generations;
            return A;
        }
        (void)hipGetLastError();
        if (e) break;
    }
    return nullptr;
}

// unmaps and releases the chunks at the END of the backed range that are completely free (all of them when nothing
// is allocated); caller holds the lock and has made sure the device is idle.  Their addresses are retired.
void arena_shrink(Arena &A) {
    const double t0 = wall_s();
    size_t at = 0;
    bool unmapped = false;
    while (!A.chunks.empty() && A.shrink_one(kArenaChunk, &at)) {
        const hipError_t eu = hipMemUnmap(A.base + at, kArenaChunk);
        if (eu != hipSuccess) {  // still mapped, but no longer in the index: the chunk is lost to this arena, not reused
            fprintf(stderr, "[bbk] arena: hipMemUnmap(+%zu GiB) failed: %s\n", at >> 30, hipGetErrorString(eu));
            (void)hipGetLastError();
        } else {
            const hipError_t er = hipMemRelease(A.chunks.back());
            if (er != hipSuccess) {
                fprintf(stderr, "[bbk] arena: hipMemRelease failed: %s\n", hipGetErrorString(er));
                (void)hipGetLastError();
            }
        }
        A.chunks.pop_back();
        ++pool().frees;
        unmapped = true;
    }
    if (unmapped) {
        // an ordinary allocation freed after the unmaps makes the driver invalidate the device's address translations
        // (probe modes flush_before / flush_after: a re-mapped range is coherent again after this).  The arena does not
        // depend on it -- it never maps at a retired address -- but whoever gets the freed memory next should not meet
        // stale translations of ours either.
        void *o = nullptr;
        if (hipMalloc(&o, 64ull << 20) == hipSuccess) {
            (void)hipMemset(o, 0, 64ull << 20);
            (void)hipDeviceSynchronize();
            (void)hipFree(o);
        }
        (void)hipGetLastError();
    }
    pool().free_s += wall_s() - t0;
}

// maps `n` more chunks at the top of the arena; false on out-of-memory or when the reservation is used up (nothing is
// left half-mapped: the chunks that were mapped stay and are usable)
bool arena_grow(Arena &A, int dev, size_t n) {
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    const double t0 = wall_s();
    size_t done = 0;
    for (; done < n; ++done) {
        if (A.top + kArenaChunk > A.va_bytes) break;
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, kArenaChunk, &prop, 0) != hipSuccess) break;
        if (hipMemMap(A.base + A.top, kArenaChunk, 0, h, 0) != hipSuccess ||
            hipMemSetAccess(A.base + A.top, kArenaChunk, &acc, 1) != hipSuccess) {
            (void)hipMemRelease(h);
            break;
        }
        A.chunks.push_back(h);
        A.grown(kArenaChunk);
    }
    (void)hipGetLastError();
    pool().malloc_s += wall_s() - t0;
    pool().mallocs += done;
    pool().malloc_bytes += (double)done * kArenaChunk;
    return done == n;
}

// gives every cached block of `device` (all threads) back to the driver (block-cache fallback)
void trim_device_blocks(int device) {
    for (auto &kv : pool().free_blocks) {
        if (kv.first.device != device) continue;
        const double t0 = wall_s();
        for (auto &b : kv.second) (void)hipFree(b.second);
        pool().free_s += wall_s() - t0;
        pool().frees += kv.second.size();
        kv.second.clear();
    }
}

// trims every generation of a key; generations other than the serving one go away once nothing is backed in them
// (their reservation stays reserved: a freed range could be handed out again by the driver -- the reuse to avoid)
void gens_shrink(ArenaGens &G) {
    for (size_t i = 0; i < G.gens.size();) {
        arena_shrink(*G.gens[i]);
        if (i + 1 < G.gens.size() && G.gens[i]->mapped == 0) G.gens.erase(G.gens.begin() + (long)i);
        else ++i;
    }
}
}  // namespace

void pool_report() {
    std::lock_guard<std::mutex> g(pool().mu);
    size_t mapped = 0, retired = 0;
    for (auto &kv : pool().arenas)
        for (auto &A : kv.second.gens) {
            mapped += A->mapped;
            retired += A->top - A->mapped;
        }
    fprintf(stderr, "[bbk] allocator (%s): %llu %s (%.1f GB, %.3f s), %llu requests served from mapped memory, %llu releases "
                    "(%.3f s), %.1f GB mapped now, %.1f GB of addresses retired, %llu reservation(s)\n",
            pool().vmm == 1 ? "arena" : "block cache", (unsigned long long)pool().mallocs,
            pool().vmm == 1 ? "chunks mapped" : "hipMalloc", pool().malloc_bytes / 1e9, pool().malloc_s,
            (unsigned long long)pool().hits, (unsigned long long)pool().frees, pool().free_s, (double)mapped / 1e9,
            (double)retired / 1e9, (unsigned long long)pool().generations);
}

bool pool_owns(const void *p) {
    std::lock_guard<std::mutex> g(pool().mu);
    for (auto &kv : pool().arenas)
        for (auto &A : kv.second.gens)
            if (A->contains(p)) return true;
    return false;
}

void pool_stats(int device, uint64_t *mapped_now, uint64_t *mapped_total, double *map_seconds) {
    std::lock_guard<std::mutex> g(pool().mu);
    const PoolKey key{device, std::this_thread::get_id()};
    uint64_t now = 0;
    auto it = pool().arenas.find(key);
    if (it != pool().arenas.end())
        for (auto &A : it->second.gens) now += A->mapped;
    if (mapped_now) *mapped_now = now;
    if (mapped_total) *mapped_total = (uint64_t)pool().malloc_bytes;
    if (map_seconds) *map_seconds = pool().malloc_s;
}

size_t pool_mapped_bytes(int device) {
    std::lock_guard<std::mutex> g(pool().mu);
    size_t mapped = 0;
    for (auto &kv : pool().arenas)
        if (kv.first.device == device)
            for (auto &A : kv.second.gens) mapped += A->mapped;
    return mapped;
}

static void *pool_alloc_impl(size_t bytes, size_t *granted, int *device);

// BBK_POOL_POISON=1 (tests): every block handed out is filled with 0xCD first, so that code which reads device memory
// it never wrote -- and got away with it on the driver's zero-filled fresh allocations -- fails deterministically
void *pool_alloc(size_t bytes, size_t *granted, int *device) {
    void *p = pool_alloc_impl(bytes, granted, device);
    static const bool poison = getenv("BBK_POOL_POISON") != nullptr;
    if (poison) {
        (void)hipDeviceSynchronize();
        (void)hipMemset(p, 0xCD, *granted);
        (void)hipDeviceSynchronize();
    }
    return p;
}

static void *pool_alloc_impl(size_t bytes, size_t *granted, int *device) {
    const size_t want = bytes <= kPoolGranule ? kPoolGranule : ((bytes + kPoolGranule - 1) / kPoolGranule) * kPoolGranule;
    const int dev = current_device();
    *device = dev;
    const PoolKey key{dev, std::this_thread::get_id()};
    std::unique_lock<std::mutex> g(pool().mu);
    if (pool().vmm == -1) pool().vmm = getenv("BBK_NO_VMM") ? 0 : 1;
    if (pool().vmm == 1) {
        ArenaGens &G = pool().arenas[key];
        if (G.gens.empty()) {
            auto A = arena_reserve();
            if (!A) {
                bool others = false;
                for (auto &kv : pool().arenas) others = others || !kv.second.gens.empty();
                pool().arenas.erase(key);
                if (others) {
                    set_error("hipMemAddressReserve failed");
                    throw Error{BBK_ERR_HIP};
                }
                pool().vmm = 0;  // no virtual-memory API on this platform: block cache
            } else {
                G.gens.push_back(std::move(A));
            }
        }
    }
    if (pool().vmm == 1) {
        ArenaGens &G = pool().arenas[key];
        for (int attempt = 0;; ++attempt) {
            Arena &A = *G.gens.back();
            size_t off = 0;
            if (A.take(want, &off)) {
                ++pool().hits;
                *granted = want;
                return A.base + off;
            }
            // grow: the request may start in the free tail of the backed range
            const size_t tail = A.free_tail();
            const size_t need = (want - tail + kArenaChunk - 1) / kArenaChunk;
            if (A.top + need * kArenaChunk > A.va_bytes) {
                // the reservation is used up (retired addresses are not reused): a new generation serves from here on
                BBK_REQUIRE(attempt == 0 && want + kArenaChunk <= A.va_bytes, BBK_ERR_NOMEM,
                            "request of %zu bytes does not fit a reservation of %zu", want, A.va_bytes);
                auto N = arena_reserve();
                BBK_REQUIRE(N != nullptr, BBK_ERR_HIP, "hipMemAddressReserve failed for a new arena generation");
                G.gens.push_back(std::move(N));
                continue;
            }
            if (!arena_grow(A, dev, need)) {
                // out of device memory: give back what the arenas of this device hold unused -- other threads' and our
                // own free chunks (their owners' streams may still have the last use of that memory queued: wait for
                // the device first)
                (void)hipDeviceSynchronize();
                for (auto &kv : pool().arenas)
                    if (kv.first.device == dev) gens_shrink(kv.second);
                Arena &B = *G.gens.back();
                const size_t tail2 = B.free_tail();
                const size_t need2 = want > tail2 ? (want - tail2 + kArenaChunk - 1) / kArenaChunk : 0;
                if (B.top + need2 * kArenaChunk > B.va_bytes || !arena_grow(B, dev, need2)) {
                    set_error("device memory exhausted: %zu bytes requested, %.1f GB mapped, %.1f GB of it free but fragmented",
                              want, (double)B.mapped / 1e9, (double)B.free_bytes() / 1e9);
                    throw Error{BBK_ERR_NOMEM};
                }
            }
            if (!G.gens.back()->take(want, &off)) {
                set_error("arena: internal error after growing");
                throw Error{BBK_ERR_INTERNAL};
            }
            *granted = want;
            return G.gens.back()->base + off;
        }
    }
    // ---- block-cache fallback
    {
        auto &fl = pool().free_blocks[key];
        auto it = fl.lower_bound(want);
        if (it != fl.end() && it->first <= want + want / 8) {  // accept a cached block up to 12.5 % larger than asked
            void *p = it->second;
            *granted = it->first;
            fl.erase(it);
            ++pool().hits;
            return p;
        }
    }
    void *p = nullptr;
    const double t0 = wall_s();
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        trim_device_blocks(dev);  // give cached blocks back and retry once
        e = hipMalloc(&p, want);
    }
    pool().malloc_s += wall_s() - t0;
    ++pool().mallocs;
    pool().malloc_bytes += (double)want;
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
        throw Error{e == hipErrorOutOfMemory ? BBK_ERR_NOMEM : BBK_ERR_HIP};
    }
    *granted = want;
    return p;
}

void pool_free(void *p, size_t bytes, int device) {
    std::lock_guard<std::mutex> g(pool().mu);
    const PoolKey key{device, std::this_thread::get_id()};
    if (pool().vmm == 1) {
        // the block returns to the arena it came from: a generation of this (device, thread), or -- a handle released
        // by another thread -- whichever arena of the device contains the address
        for (auto &kv : pool().arenas) {
            if (kv.first.device != device) continue;
            for (auto &A : kv.second.gens)
                if (A->contains(p)) {
                    A->add_free((size_t)((char *)p - A->base), bytes);
                    return;
                }
        }
        return;  // not ours (cannot happen)
    }
    pool().free_blocks[key].emplace(bytes, p);
}

// gives the unused memory of this (device, calling thread) back to the driver: what a context being destroyed (or
// bbk_ctx_trim) leaves behind.  The caller has drained the context's stream; the device is synchronised here because
// unmapping under a running kernel of another stream would fault it.
void pool_trim(int device) {
    std::lock_guard<std::mutex> g(pool().mu);
    const PoolKey key{device, std::this_thread::get_id()};
    if (pool().vmm == 1) {
        auto it = pool().arenas.find(key);
        if (it == pool().arenas.end() || getenv("BBK_ARENA_KEEP")) return;  // BBK_ARENA_KEEP=1: A/B switch, never unmap
        (void)hipDeviceSynchronize();
        gens_shrink(it->second);
        return;
    }
    auto it = pool().free_blocks.find(key);
    if (it == pool().free_blocks.end()) return;
    const double t0 = wall_s();
    for (auto &b : it->second) (void)hipFree(b.second);
    pool().free_s += wall_s() - t0;
    pool().frees += it->second.size();
    pool().free_blocks.erase(it);
}

hipError_t copy_async(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t stream) {
    if (bytes == 0) return hipSuccess;
    return hipMemcpyAsync(dst, src, bytes, kind, stream);
}

void d2h_big(bbk_ctx *ctx, void *dst, const void *src, size_t bytes) {
    constexpr size_t kChunk = 32ull << 20;
    if (bytes < (4ull << 20)) {
        BBK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        BBK_HIP(hipStreamSynchronize(ctx->stream));
        return;
    }
    if (!ctx->pinned[0]) {
        BBK_HIP(hipHostMalloc(&ctx->pinned[0], kChunk, hipHostMallocDefault));
        BBK_HIP(hipHostMalloc(&ctx->pinned[1], kChunk, hipHostMallocDefault));
        ctx->pinned_bytes = kChunk;
    }
    hipEvent_t ev[2];
    BBK_HIP(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
    BBK_HIP(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
    const size_t nchunks = (bytes + kChunk - 1) / kChunk;
    auto issue = [&](size_t c) {
        const size_t off = c * kChunk, sz = std::min(kChunk, bytes - off);
        (void)hipMemcpyAsync(ctx->pinned[c & 1], (const char *)src + off, sz, hipMemcpyDeviceToHost, ctx->stream);
        (void)hipEventRecord(ev[c & 1], ctx->stream);
    };
    issue(0);
    for (size_t c = 0; c < nchunks; ++c) {
        if (c + 1 < nchunks) issue(c + 1);
        BBK_HIP(hipEventSynchronize(ev[c & 1]));
        const size_t off = c * kChunk, sz = std::min(kChunk, bytes - off);
        memcpy((char *)dst + off, ctx->pinned[c & 1], sz);
    }
    (void)hipEventDestroy(ev[0]);
    (void)hipEventDestroy(ev[1]);
}

bool d2f_big(bbk_ctx *ctx, int fd, uint64_t file_off, const void *src, size_t bytes) {
    constexpr size_t kChunk = 32ull << 20;
    static const int kWriters = [] {
        const char *e = getenv("BBK_WRITE_THREADS");
        const int t = e ? atoi(e) : 4;  // measured on tmpfs: 2 -> 3.8, 4 -> 6.4, 8 -> 3.5, 16 -> 4.6 GB/s (one box; the writers share 16 cores)
        return t < 1 ? 1 : (t > 64 ? 64 : t);
    }();
    if (bytes == 0) return true;
    if (!ctx->pinned[0]) {
        BBK_HIP(hipHostMalloc(&ctx->pinned[0], kChunk, hipHostMallocDefault));
        BBK_HIP(hipHostMalloc(&ctx->pinned[1], kChunk, hipHostMallocDefault));
        ctx->pinned_bytes = kChunk;
    }
    // reserve the file's pages in one call first: concurrent writers that each allocate page-cache pages contend (tmpfs:
    // 3-6 GB/s, noisy; pre-allocated: 6.1-6.6 GB/s whatever the writer count).  The raw syscall, not posix_fallocate: a
    // file system without support just says so (the glibc emulation would write zeros).  What remains is the kernel's
    // own page allocation for ONE tmpfs file (~7.5 GB/s, serialised on the inode): writing through a mapping of the
    // file, or letting the device copy straight into the registered mapping (no CPU copy at all), measured the same.
    if (!getenv("BBK_NO_FALLOCATE")) (void)fallocate(fd, 0, (off_t)file_off, (off_t)bytes);
    hipEvent_t ev[2];
    BBK_HIP(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
    BBK_HIP(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
    const size_t nchunks = (bytes + kChunk - 1) / kChunk;
    auto issue = [&](size_t c) {
        const size_t off = c * kChunk, sz = std::min(kChunk, bytes - off);
        (void)hipMemcpyAsync(ctx->pinned[c & 1], (const char *)src + off, sz, hipMemcpyDeviceToHost, ctx->stream);
        (void)hipEventRecord(ev[c & 1], ctx->stream);
    };
    bool ok = true;
    issue(0);
    for (size_t c = 0; c < nchunks && ok; ++c) {
        if (hipEventSynchronize(ev[c & 1]) != hipSuccess) {
            ok = false;
            break;
        }
        // chunk c + 1 goes to the other buffer, whose writes (chunk c - 1) ended with the last iteration
        if (c + 1 < nchunks) issue(c + 1);
        const size_t off = c * kChunk, sz = std::min(kChunk, bytes - off);
        // several writers per chunk: a single pwrite stream into tmpfs runs at ~1/3 of what the box can do
        const int T = kWriters;
        const size_t part = (((sz + T - 1) / T) + 4095) & ~(size_t)4095;  // page-aligned shares
#pragma omp parallel for num_threads(T) schedule(static)
        for (int t = 0; t < T; ++t) {
            size_t o = std::min(sz, (size_t)t * part);
            const size_t e = std::min(sz, o + part);
            while (o < e) {
                const ssize_t w = pwrite(fd, (const char *)ctx->pinned[c & 1] + o, e - o, (off_t)(file_off + off + o));
                if (w <= 0) {
#pragma omp atomic write
                    ok = false;
                    break;
                }
                o += (size_t)w;
            }
        }
    }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipEventDestroy(ev[0]);
    (void)hipEventDestroy(ev[1]);
    return ok;
}

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kRadix = 256;

template <int W>
struct SortCfg {
    // tile sized so that static + staged LDS stays below 64 KiB for every key width
    static constexpr int ITEMS = (W == 1) ? 16 : (W <= 3 ? 8 : 4);
    static constexpr int TILE = kThreads * ITEMS;
};

// ------------------------------------------------------------------------------------------
// block-wide exclusive scan of one u32 per thread (256 threads), wave shuffles + LDS
// ------------------------------------------------------------------------------------------
__device__ inline uint32_t wave_incl_scan(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// returns the exclusive prefix of v over the block; *total receives the block sum.
__device__ inline uint32_t block_excl_scan(uint32_t v, uint32_t *smem_waves /*[kWaves+1]*/, uint32_t *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = wave_incl_scan(v);
    if (lane == 63) smem_waves[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
        uint32_t s = smem_waves[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

// ------------------------------------------------------------------------------------------
// exclusive scan of u64
// ------------------------------------------------------------------------------------------
constexpr int kScanItems = 8;
constexpr int kScanTile = kThreads * kScanItems;

__global__ __launch_bounds__(kThreads) void k_scan_reduce(const uint64_t *__restrict__ in, uint64_t n,
                                                         uint64_t *__restrict__ bsum) {
    __shared__ uint64_t sm[kWaves];
    const uint64_t base = (uint64_t)blockIdx.x * kScanTile;
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        uint64_t idx = base + (uint64_t)i * kThreads + threadIdx.x;
        if (idx < n) s += in[idx];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_down(s, d, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t t = 0;
        for (int w = 0; w < kWaves; ++w) t += sm[w];
        bsum[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(kThreads) void k_scan_apply(const uint64_t *__restrict__ in, uint64_t *__restrict__ out,
                                                        uint64_t n, const uint64_t *__restrict__ boff) {
    __shared__ uint64_t sm[kWaves];
    const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanItems;
    uint64_t v[kScanItems];
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        v[i] = (base + i < n) ? in[base + i] : 0;
        s += v[i];
    }
    // block exclusive scan of s
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t incl = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    if (lane == 63) sm[wave] = incl;
    __syncthreads();
    uint64_t wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += sm[w];
    uint64_t run = boff[blockIdx.x] + wbase + incl - s;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        if (base + i < n) out[base + i] = run;
        run += v[i];
    }
}

// One level of the scan: block sums, (recursively) their exclusive scan in place, then the blocks.  Nothing waits on the
// host in between: the block-sum buffers of every level live in `keep` until the caller has synchronised once, and the
// grand total is parked in *d_total on the way down.
static void scan_level(bbk_ctx *ctx, const uint64_t *in, uint64_t *out, uint64_t n, std::vector<DevBuf> &keep,
                       uint64_t *d_total) {
    const uint64_t nb = (n + kScanTile - 1) / kScanTile;
    keep.emplace_back((nb + 1) * sizeof(uint64_t));
    uint64_t *bsum = keep.back().as<uint64_t>();
    hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)nb), dim3(kThreads), 0, ctx->stream, in, n, bsum);
    check_launch("k_scan_reduce");
    if (nb == 1) {
        BBK_HIP(hipMemcpyAsync(d_total, bsum, sizeof(uint64_t), hipMemcpyDeviceToDevice, ctx->stream));
        BBK_HIP(hipMemsetAsync(bsum, 0, sizeof(uint64_t), ctx->stream));
    } else {
        scan_level(ctx, bsum, bsum, nb, keep, d_total);
    }
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(kThreads), 0, ctx->stream, in, out, n, bsum);
    check_launch("k_scan_apply");
}

uint64_t exclusive_scan_u64(bbk_ctx *ctx, const uint64_t *in, uint64_t *out, uint64_t n) {
    if (n == 0) return 0;
    std::vector<DevBuf> keep;
    keep.reserve(8);  // levels: log_{tile}(n)
    DevBuf dt(16);
    scan_level(ctx, in, out, n, keep, dt.as<uint64_t>());
    uint64_t total = 0;
    BBK_HIP(hipMemcpyAsync(&total, dt.p, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    BBK_HIP(hipStreamSynchronize(ctx->stream));  // one wait per scan (it was one per level): the buffers go back now
    return total;
}

// ------------------------------------------------------------------------------------------
// radix sort
// ------------------------------------------------------------------------------------------
// Digit of a record that sits in memory (global or LDS) at p.  For a key-bit pass the sort word
// is re-read from memory with a dynamic MEMORY index: indexing the register copy with the
// runtime pd.word would push the whole tile to scratch.
template <int W>
__device__ inline uint32_t digit_of(const Key<W> &key, const Key<W> *p, const PassDesc &pd) {
    if (pd.kind == 0) {
        const uint64_t w = (W == 1) ? key.w[0] : reinterpret_cast<const uint64_t *>(p)[pd.word];
        return (uint32_t)(w >> pd.shift) & ((1u << pd.bits) - 1u);
    }
    if (pd.kind == 1) return (uint32_t)__umul64hi(xxh3_64<W>(key), (uint64_t)pd.nb);
    return (uint32_t)__umul64hi(owner_mix<W>(key), (uint64_t)pd.nb);
}

// per-tile digit histogram -> hist[tile][256]
template <int W>
__global__ __launch_bounds__(kThreads) void k_hist(const Key<W> *__restrict__ in, uint64_t n, PassDesc pd,
                                                  uint32_t *__restrict__ hist) {
    constexpr int ITEMS = SortCfg<W>::ITEMS, TILE = SortCfg<W>::TILE;
    __shared__ uint32_t h[kRadix];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * TILE;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        uint64_t idx = base + (uint64_t)i * kThreads + threadIdx.x;
        if (idx < n) {
            const Key<W> key = key_load<W>(&in[idx]);
            atomicAdd(&h[digit_of<W>(key, &in[idx], pd)], 1u);
        }
    }
    __syncthreads();
    hist[(uint64_t)blockIdx.x * kRadix + threadIdx.x] = h[threadIdx.x];
}

// column sums over chunks of kChunk tiles
constexpr int kChunk = 256;
__global__ __launch_bounds__(kRadix) void k_colsum(const uint32_t *__restrict__ hist, uint64_t ntiles,
                                                  uint64_t *__restrict__ chunk_sum) {
    const uint64_t t0 = (uint64_t)blockIdx.x * kChunk;
    const uint64_t t1 = (t0 + kChunk < ntiles) ? t0 + kChunk : ntiles;
    uint64_t s = 0;
    for (uint64_t t = t0; t < t1; ++t) s += hist[t * kRadix + threadIdx.x];
    chunk_sum[(uint64_t)blockIdx.x * kRadix + threadIdx.x] = s;
}

// single block: chunk_sum[c][d] -> exclusive global base of (chunk c, digit d) in digit-major order
__global__ __launch_bounds__(kRadix) void k_scan_chunks(uint64_t *__restrict__ chunk_sum, uint64_t nchunks,
                                                       uint64_t *__restrict__ digit_total) {
    __shared__ uint64_t tot[kRadix];
    const int d = threadIdx.x;
    uint64_t run = 0;
    for (uint64_t c = 0; c < nchunks; ++c) {
        uint64_t v = chunk_sum[c * kRadix + d];
        chunk_sum[c * kRadix + d] = run;
        run += v;
    }
    tot[d] = run;
    if (digit_total) digit_total[d] = run;
    __syncthreads();
    uint64_t base = 0;
    for (int j = 0; j < d; ++j) base += tot[j];
    for (uint64_t c = 0; c < nchunks; ++c) chunk_sum[c * kRadix + d] += base;
}

// hist[t][d] (counts) -> global start offset of (tile t, digit d)
__global__ __launch_bounds__(kRadix) void k_tile_offsets(uint32_t *__restrict__ hist, uint64_t ntiles,
                                                        const uint64_t *__restrict__ chunk_base) {
    const uint64_t t0 = (uint64_t)blockIdx.x * kChunk;
    const uint64_t t1 = (t0 + kChunk < ntiles) ? t0 + kChunk : ntiles;
    // offsets RELATIVE to the chunk's base (a chunk of 256 tiles holds < 2^32 records): the scatter kernel adds the
    // 64-bit base, so that an array of 2^32 records or more sorts like any other
    uint32_t run = 0;
    for (uint64_t t = t0; t < t1; ++t) {
        uint32_t v = hist[t * kRadix + threadIdx.x];
        hist[t * kRadix + threadIdx.x] = run;
        run += v;
    }
    (void)chunk_base;
}

// wavefront match-any on an 8-bit digit: mask of lanes (among `valid` lanes) holding the same digit
__device__ inline uint64_t match_digit(uint32_t d, bool valid) {
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool bit = (d >> b) & 1u;
        const uint64_t m = __ballot(bit);
        peers &= bit ? m : ~m;
    }
    return peers;
}

template <int W, bool HAS_VAL>
__global__ __launch_bounds__(kThreads) void k_scatter(const Key<W> *__restrict__ in, Key<W> *__restrict__ out,
                                                     const uint32_t *__restrict__ vin, uint32_t *__restrict__ vout,
                                                     uint64_t n, PassDesc pd,
                                                     const uint32_t *__restrict__ tile_off,
                                                     const uint64_t *__restrict__ chunk_base) {
    constexpr int ITEMS = SortCfg<W>::ITEMS, TILE = SortCfg<W>::TILE;
    __shared__ uint32_t wave_cnt[kWaves][kRadix];
    __shared__ uint32_t digit_start[kRadix];
    __shared__ uint64_t goff[kRadix];
    __shared__ uint32_t scan_tmp[kWaves + 1];
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
    Key<W> *stage = reinterpret_cast<Key<W> *>(dyn_smem);
    uint32_t *vstage = reinterpret_cast<uint32_t *>(dyn_smem + sizeof(Key<W>) * TILE);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t tile = blockIdx.x;
    const uint64_t base = tile * TILE;
    const uint32_t tile_n = (uint32_t)((n - base < (uint64_t)TILE) ? (n - base) : (uint64_t)TILE);

#pragma unroll
    for (int w = 0; w < kWaves; ++w) wave_cnt[w][tid] = 0;
    __syncthreads();

    Key<W> keys[ITEMS];
    uint32_t vals[ITEMS];
    uint32_t dig[ITEMS];
    uint32_t rank[ITEMS];
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t local = (uint32_t)(wave * (ITEMS * 64) + i * 64 + lane);
        const bool valid = local < tile_n;
        uint32_t d = 0;
#pragma unroll
        for (int j = 0; j < W; ++j) keys[i].w[j] = 0;
        vals[i] = 0;
        if (valid) {
            keys[i] = key_load<W>(&in[base + local]);
            if (HAS_VAL) vals[i] = vin[base + local];
            d = digit_of<W>(keys[i], &in[base + local], pd);
        }
        dig[i] = d;
        const uint64_t peers = match_digit(d, valid);
        const uint32_t pre = wave_cnt[wave][d];
        rank[i] = pre + (uint32_t)__popcll(peers & lt_mask);
        // the highest lane of each peer group publishes the new running count
        if (valid && (peers >> lane) == 1ull) wave_cnt[wave][d] = pre + (uint32_t)__popcll(peers);
    }
    __syncthreads();

    // digit tid: exclusive prefix over waves, then exclusive scan over digits
    uint32_t tot = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
        uint32_t c = wave_cnt[w][tid];
        wave_cnt[w][tid] = tot;
        tot += c;
    }
    uint32_t tile_total;
    const uint32_t dstart = block_excl_scan(tot, scan_tmp, &tile_total);
    digit_start[tid] = dstart;
    goff[tid] = chunk_base[(tile / kChunk) * kRadix + tid] + tile_off[tile * kRadix + tid] - dstart;  // + pos below
    __syncthreads();

#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t local = (uint32_t)(wave * (ITEMS * 64) + i * 64 + lane);
        if (local < tile_n) {
            const uint32_t pos = digit_start[dig[i]] + wave_cnt[wave][dig[i]] + rank[i];
            key_store<W>(&stage[pos], keys[i]);
            if (HAS_VAL) vstage[pos] = vals[i];
        }
    }
    __syncthreads();

#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t pos = (uint32_t)(i * kThreads + tid);
        if (pos < tile_n) {
            const Key<W> key = key_load<W>(&stage[pos]);
            const uint32_t d = digit_of<W>(key, &stage[pos], pd);
            const uint64_t g = goff[d] + pos;
            key_store<W>(&out[g], key);
            if (HAS_VAL) vout[g] = vstage[pos];
        }
    }
}

std::vector<PassDesc> key_passes(unsigned k) {
    // least significant first: last word (populated low bits only) ... word 0
    std::vector<PassDesc> p;
    const int W = (int)words_of(k);
    for (int w = W - 1; w >= 0; --w) {
        const int bits = (w == W - 1) ? (int)(2 * k - 64 * (W - 1)) : 64;
        for (int s = 0; s < bits; s += 8) {
            const int b = (bits - s < 8) ? bits - s : 8;
            p.push_back({0, w, s, b, 0});
        }
    }
    return p;
}

// one stable counting pass src -> dst on the digit `pd` (hist/chunk: scratch of the caller)
template <int W>
static void sort_pass(bbk_ctx *ctx, const Key<W> *src, Key<W> *dst, const uint32_t *vsrc, uint32_t *vdst, uint64_t n,
                      const PassDesc &pd, DevBuf &hist, DevBuf &chunk, uint64_t *d_digit_totals = nullptr) {
    constexpr int TILE = SortCfg<W>::TILE;
    const uint64_t ntiles = (n + TILE - 1) / TILE;
    const uint64_t nchunks = (ntiles + kChunk - 1) / kChunk;
    const bool hv = vsrc != nullptr;
    const size_t dyn = sizeof(Key<W>) * TILE + (hv ? sizeof(uint32_t) * TILE : 0);
    const double rec_bytes = (double)(sizeof(Key<W>) + (hv ? 4 : 0));
    {
        KernelTimer t(ctx, "hist", (double)n * sizeof(Key<W>));
        hipLaunchKernelGGL(k_hist<W>, dim3((unsigned)ntiles), dim3(kThreads), 0, ctx->stream, src, n, pd,
                           hist.as<uint32_t>());
        check_launch("k_hist");
    }
    {
        KernelTimer t(ctx, "scan", (double)ntiles * kRadix * 4 * 3);
        hipLaunchKernelGGL(k_colsum, dim3((unsigned)nchunks), dim3(kRadix), 0, ctx->stream, hist.as<uint32_t>(), ntiles,
                           chunk.as<uint64_t>());
        hipLaunchKernelGGL(k_scan_chunks, dim3(1), dim3(kRadix), 0, ctx->stream, chunk.as<uint64_t>(), nchunks,
                           d_digit_totals);
        hipLaunchKernelGGL(k_tile_offsets, dim3((unsigned)nchunks), dim3(kRadix), 0, ctx->stream, hist.as<uint32_t>(),
                           ntiles, chunk.as<uint64_t>());
        check_launch("scan kernels");
    }
    {
        KernelTimer t(ctx, "scatter", 2.0 * (double)n * rec_bytes);
        if (hv) {
            hipLaunchKernelGGL((k_scatter<W, true>), dim3((unsigned)ntiles), dim3(kThreads), dyn, ctx->stream, src, dst,
                               vsrc, vdst, n, pd, hist.as<uint32_t>(), chunk.as<uint64_t>());
        } else {
            hipLaunchKernelGGL((k_scatter<W, false>), dim3((unsigned)ntiles), dim3(kThreads), dyn, ctx->stream, src, dst,
                               (const uint32_t *)nullptr, (uint32_t *)nullptr, n, pd, hist.as<uint32_t>(), chunk.as<uint64_t>());
        }
        check_launch("k_scatter");
    }
}

template <int W>
static void sort_scratch(uint64_t n, bool with_vals, DevBuf &hist, DevBuf &chunk) {
    BBK_REQUIRE(n < (1ull << 40), BBK_ERR_ARG, "sort_records: n=%llu", (unsigned long long)n);
    constexpr int TILE = SortCfg<W>::TILE;
    const uint64_t ntiles = (n + TILE - 1) / TILE;
    const uint64_t nchunks = (ntiles + kChunk - 1) / kChunk;
    hist.alloc(ntiles * kRadix * sizeof(uint32_t));
    chunk.alloc(nchunks * kRadix * sizeof(uint64_t));
    const size_t dyn = sizeof(Key<W>) * TILE + (with_vals ? sizeof(uint32_t) * TILE : 0);
    if (with_vals) {
        BBK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_scatter<W, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    } else {
        BBK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_scatter<W, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    }
}

template <int W>
static void sort_impl(bbk_ctx *ctx, Key<W> *keys, Key<W> *tmp, uint32_t *vals, uint32_t *vtmp, uint64_t n,
                      const std::vector<PassDesc> &passes) {
    if (n <= 1 || passes.empty()) return;
    DevBuf hist, chunk;
    sort_scratch<W>(n, vals != nullptr, hist, chunk);
    Key<W> *src = keys, *dst = tmp;
    uint32_t *vsrc = vals, *vdst = vtmp;
    for (const PassDesc &pd : passes) {
        sort_pass<W>(ctx, src, dst, vsrc, vdst, n, pd, hist, chunk);
        std::swap(src, dst);
        std::swap(vsrc, vdst);
    }
    if (src != keys) {
        BBK_HIP(bbk::copy_async(keys, src, n * sizeof(Key<W>), hipMemcpyDeviceToDevice, ctx->stream));
        if (vals) BBK_HIP(bbk::copy_async(vals, vsrc, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
    }
    BBK_HIP(hipStreamSynchronize(ctx->stream));  // hist/chunk are freed on return
}

template <int W>
static void partition_impl(bbk_ctx *ctx, const Key<W> *src, Key<W> *dst, const uint32_t *vsrc, uint32_t *vdst, uint64_t n,
                           const PassDesc &pd, uint64_t *h_digit_totals) {
    if (h_digit_totals) memset(h_digit_totals, 0, kRadix * sizeof(uint64_t));
    if (n == 0) return;
    DevBuf hist, chunk, tot;
    sort_scratch<W>(n, vsrc != nullptr, hist, chunk);
    if (h_digit_totals) tot.alloc(kRadix * sizeof(uint64_t));
    sort_pass<W>(ctx, src, dst, vsrc, vdst, n, pd, hist, chunk, h_digit_totals ? tot.as<uint64_t>() : nullptr);
    if (h_digit_totals)
        BBK_HIP(hipMemcpyAsync(h_digit_totals, tot.p, kRadix * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    BBK_HIP(hipStreamSynchronize(ctx->stream));
}

// one stable pass src -> dst (both on the device, not aliased): records ordered by the digit, input order kept inside
void partition_records(bbk_ctx *ctx, int W, const void *src, void *dst, const uint32_t *vsrc, uint32_t *vdst, uint64_t n,
                       const PassDesc &pd, uint64_t *h_digit_totals) {
    switch (W) {
        case 1: partition_impl<1>(ctx, (const Key<1> *)src, (Key<1> *)dst, vsrc, vdst, n, pd, h_digit_totals); break;
        case 2: partition_impl<2>(ctx, (const Key<2> *)src, (Key<2> *)dst, vsrc, vdst, n, pd, h_digit_totals); break;
        case 3: partition_impl<3>(ctx, (const Key<3> *)src, (Key<3> *)dst, vsrc, vdst, n, pd, h_digit_totals); break;
        case 4: partition_impl<4>(ctx, (const Key<4> *)src, (Key<4> *)dst, vsrc, vdst, n, pd, h_digit_totals); break;
        default: BBK_REQUIRE(false, BBK_ERR_ARG, "unsupported key width %d", W);
    }
}

void sort_records(bbk_ctx *ctx, int W, void *keys, void *keys_tmp, uint32_t *vals, uint32_t *vals_tmp, uint64_t n,
                  const std::vector<PassDesc> &passes) {
    switch (W) {
        case 1: sort_impl<1>(ctx, (Key<1> *)keys, (Key<1> *)keys_tmp, vals, vals_tmp, n, passes); break;
        case 2: sort_impl<2>(ctx, (Key<2> *)keys, (Key<2> *)keys_tmp, vals, vals_tmp, n, passes); break;
        case 3: sort_impl<3>(ctx, (Key<3> *)keys, (Key<3> *)keys_tmp, vals, vals_tmp, n, passes); break;
        case 4: sort_impl<4>(ctx, (Key<4> *)keys, (Key<4> *)keys_tmp, vals, vals_tmp, n, passes); break;
        default: BBK_REQUIRE(false, BBK_ERR_ARG, "unsupported key width %d", W);
    }
}

// ------------------------------------------------------------------------------------------
// unique / reduce-by-key on a sorted array
// ------------------------------------------------------------------------------------------
constexpr int kUniqItems = 8;
constexpr int kUniqTile = kThreads * kUniqItems;

template <int W>
__device__ inline bool is_head(const Key<W> *__restrict__ keys, uint64_t idx) {
    if (idx == 0) return true;
    return !key_eq<W>(keys[idx], keys[idx - 1]);
}

template <int W>
__global__ __launch_bounds__(kThreads) void k_head_count(const Key<W> *__restrict__ keys, uint64_t n,
                                                        uint64_t *__restrict__ bcount) {
    __shared__ uint32_t sm[kWaves];
    const uint64_t base = (uint64_t)blockIdx.x * kUniqTile;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < kUniqItems; ++i) {
        uint64_t idx = base + (uint64_t)i * kThreads + threadIdx.x;
        if (idx < n && is_head<W>(keys, idx)) ++c;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_down(c, d, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < kWaves; ++w) t += sm[w];
        bcount[blockIdx.x] = t;
    }
}

// writes distinct keys and the index of each run head (head_idx has ndistinct+1 entries; the
// last one, = n, is written by the host wrapper)
template <int W>
__global__ __launch_bounds__(kThreads) void k_head_compact(const Key<W> *__restrict__ keys, uint64_t n,
                                                          const uint64_t *__restrict__ boff,
                                                          Key<W> *__restrict__ out_keys,
                                                          uint32_t *__restrict__ head_idx) {
    __shared__ uint32_t scan_tmp[kWaves + 1];
    const uint64_t base = (uint64_t)blockIdx.x * kUniqTile + (uint64_t)threadIdx.x * kUniqItems;
    bool h[kUniqItems];
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < kUniqItems; ++i) {
        h[i] = (base + i < n) && is_head<W>(keys, base + i);
        c += h[i] ? 1u : 0u;
    }
    uint32_t tot;
    uint32_t pre = block_excl_scan(c, scan_tmp, &tot);
    uint64_t pos = boff[blockIdx.x] + pre;
#pragma unroll
    for (int i = 0; i < kUniqItems; ++i) {
        if (h[i]) {
            if (out_keys) out_keys[pos] = keys[base + i];
            head_idx[pos] = (uint32_t)(base + i);
            ++pos;
        }
    }
}

__global__ __launch_bounds__(kThreads) void k_seg_reduce(const uint32_t *__restrict__ head_idx, uint64_t nseg,
                                                        const uint32_t *__restrict__ vals, int op,
                                                        uint32_t *__restrict__ out_vals) {
    const uint64_t s = BBK_GID();
    if (s >= nseg) return;
    const uint32_t a = head_idx[s], b = head_idx[s + 1];
    uint32_t r;
    if (op == REDUCE_COUNT) {
        r = b - a;
    } else if (op == REDUCE_SUM) {
        uint64_t t = 0;
        for (uint32_t i = a; i < b; ++i) t += vals[i];
        r = t > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)t;
    } else {
        r = 0;
        for (uint32_t i = a; i < b; ++i) r |= vals[i];
    }
    out_vals[s] = r;
}

// compaction of (key, val) pairs with val != 0
template <int W>
__global__ __launch_bounds__(kThreads) void k_nonzero_count(const uint32_t *__restrict__ vals, uint64_t n,
                                                           uint64_t *__restrict__ bcount) {
    __shared__ uint32_t sm[kWaves];
    const uint64_t base = (uint64_t)blockIdx.x * kUniqTile;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < kUniqItems; ++i) {
        uint64_t idx = base + (uint64_t)i * kThreads + threadIdx.x;
        if (idx < n && vals[idx] != 0) ++c;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_down(c, d, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < kWaves; ++w) t += sm[w];
        bcount[blockIdx.x] = t;
    }
}

template <int W>
__global__ __launch_bounds__(kThreads) void k_nonzero_compact(const Key<W> *__restrict__ keys,
                                                             const uint32_t *__restrict__ vals, uint64_t n,
                                                             const uint64_t *__restrict__ boff,
                                                             Key<W> *__restrict__ out_keys,
                                                             uint32_t *__restrict__ out_vals) {
    __shared__ uint32_t scan_tmp[kWaves + 1];
    const uint64_t base = (uint64_t)blockIdx.x * kUniqTile + (uint64_t)threadIdx.x * kUniqItems;
    bool h[kUniqItems];
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < kUniqItems; ++i) {
        h[i] = (base + i < n) && vals[base + i] != 0;
        c += h[i] ? 1u : 0u;
    }
    uint32_t tot;
    uint32_t pre = block_excl_scan(c, scan_tmp, &tot);
    uint64_t pos = boff[blockIdx.x] + pre;
#pragma unroll
    for (int i = 0; i < kUniqItems; ++i) {
        if (h[i]) {
            out_keys[pos] = keys[base + i];
            out_vals[pos] = vals[base + i];
            ++pos;
        }
    }
}

template <int W>
static uint64_t unique_impl(bbk_ctx *ctx, const Key<W> *keys, const uint32_t *vals, uint64_t n, Key<W> *out_keys,
                            uint32_t *out_vals, ReduceOp op, bool drop_zero) {
    if (n == 0) return 0;
    BBK_REQUIRE(n < (1ull << 32), BBK_ERR_ARG, "unique_records: n=%llu does not fit 32-bit offsets",
                (unsigned long long)n);
    const uint64_t nb = (n + kUniqTile - 1) / kUniqTile;
    DevBuf bcount(nb * sizeof(uint64_t));
    {
        KernelTimer t(ctx, "unique", (double)n * sizeof(Key<W>));
        hipLaunchKernelGGL(k_head_count<W>, dim3((unsigned)nb), dim3(kThreads), 0, ctx->stream, keys, n,
                           bcount.as<uint64_t>());
        check_launch("k_head_count");
    }
    const uint64_t nd = exclusive_scan_u64(ctx, bcount.as<uint64_t>(), bcount.as<uint64_t>(), nb);
    DevBuf head((nd + 1) * sizeof(uint32_t));
    const bool need_tmp = drop_zero && op == REDUCE_OR;
    DevBuf tkeys, tvals;
    Key<W> *dk = out_keys;
    uint32_t *dv = out_vals;
    if (need_tmp) {
        tkeys.alloc(nd * sizeof(Key<W>));
        tvals.alloc(nd * sizeof(uint32_t));
        dk = tkeys.as<Key<W>>();
        dv = tvals.as<uint32_t>();
    }
    {
        KernelTimer t(ctx, "unique", (double)n * sizeof(Key<W>) + (double)nd * sizeof(Key<W>));
        hipLaunchKernelGGL(k_head_compact<W>, dim3((unsigned)nb), dim3(kThreads), 0, ctx->stream, keys, n,
                           bcount.as<uint64_t>(), dk, head.as<uint32_t>());
        check_launch("k_head_compact");
    }
    const uint32_t n32 = (uint32_t)n;
    BBK_HIP(hipMemcpyAsync(head.as<uint32_t>() + nd, &n32, sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    if (dv) {
        KernelTimer t(ctx, "reduce", (double)nd * 8 + (vals ? (double)n * 4 : 0));
        hipLaunchKernelGGL(k_seg_reduce, bbk::grid_blocks((nd + kThreads - 1) / kThreads), dim3(kThreads), 0,
                           ctx->stream, head.as<uint32_t>(), nd, vals, (int)op, dv);
        check_launch("k_seg_reduce");
    }
    uint64_t result = nd;
    if (need_tmp) {
        const uint64_t nb2 = (nd + kUniqTile - 1) / kUniqTile;
        DevBuf bc2(nb2 * sizeof(uint64_t));
        hipLaunchKernelGGL(k_nonzero_count<W>, dim3((unsigned)nb2), dim3(kThreads), 0, ctx->stream, dv, nd,
                           bc2.as<uint64_t>());
        check_launch("k_nonzero_count");
        result = exclusive_scan_u64(ctx, bc2.as<uint64_t>(), bc2.as<uint64_t>(), nb2);
        hipLaunchKernelGGL(k_nonzero_compact<W>, dim3((unsigned)nb2), dim3(kThreads), 0, ctx->stream, dk, dv, nd,
                           bc2.as<uint64_t>(), out_keys, out_vals);
        check_launch("k_nonzero_compact");
        BBK_HIP(hipStreamSynchronize(ctx->stream));
    }
    BBK_HIP(hipStreamSynchronize(ctx->stream));
    return result;
}

template <int W>
static uint64_t drop_zero_impl(bbk_ctx *ctx, const Key<W> *keys, const uint32_t *vals, uint64_t n, DevBuf &ok,
                               DevBuf &ov) {
    if (n == 0) return 0;
    const uint64_t nb = (n + kUniqTile - 1) / kUniqTile;
    DevBuf bc(nb * sizeof(uint64_t));
    hipLaunchKernelGGL(k_nonzero_count<W>, dim3((unsigned)nb), dim3(kThreads), 0, ctx->stream, vals, n,
                       bc.as<uint64_t>());
    check_launch("k_nonzero_count");
    const uint64_t kept = exclusive_scan_u64(ctx, bc.as<uint64_t>(), bc.as<uint64_t>(), nb);
    if (kept == n) return n;  // nothing to drop: caller keeps its buffers (the usual case: nothing is allocated)
    ok.alloc(kept * sizeof(Key<W>) + 16);
    ov.alloc(kept * 4 + 16);
    hipLaunchKernelGGL(k_nonzero_compact<W>, dim3((unsigned)nb), dim3(kThreads), 0, ctx->stream, keys, vals, n,
                       bc.as<uint64_t>(), ok.as<Key<W>>(), ov.as<uint32_t>());
    check_launch("k_nonzero_compact");
    BBK_HIP(hipStreamSynchronize(ctx->stream));
    return kept;
}

// Compacts the (key, val) pairs with val != 0 into out_* (allocated here, only when something is dropped); returns
// how many were kept (if all are kept nothing is allocated or written).
uint64_t drop_zero_vals(bbk_ctx *ctx, int W, const void *keys, const uint32_t *vals, uint64_t n, DevBuf &out_keys,
                        DevBuf &out_vals) {
    switch (W) {
        case 1: return drop_zero_impl<1>(ctx, (const Key<1> *)keys, vals, n, out_keys, out_vals);
        case 2: return drop_zero_impl<2>(ctx, (const Key<2> *)keys, vals, n, out_keys, out_vals);
        case 3: return drop_zero_impl<3>(ctx, (const Key<3> *)keys, vals, n, out_keys, out_vals);
        case 4: return drop_zero_impl<4>(ctx, (const Key<4> *)keys, vals, n, out_keys, out_vals);
        default: BBK_REQUIRE(false, BBK_ERR_ARG, "unsupported key width %d", W);
    }
    return 0;
}

uint64_t unique_records(bbk_ctx *ctx, int W, const void *keys, const uint32_t *vals, uint64_t n, void *out_keys,
                        uint32_t *out_vals, ReduceOp op, bool drop_zero) {
    switch (W) {
        case 1: return unique_impl<1>(ctx, (const Key<1> *)keys, vals, n, (Key<1> *)out_keys, out_vals, op, drop_zero);
        case 2: return unique_impl<2>(ctx, (const Key<2> *)keys, vals, n, (Key<2> *)out_keys, out_vals, op, drop_zero);
        case 3: return unique_impl<3>(ctx, (const Key<3> *)keys, vals, n, (Key<3> *)out_keys, out_vals, op, drop_zero);
        case 4: return unique_impl<4>(ctx, (const Key<4> *)keys, vals, n, (Key<4> *)out_keys, out_vals, op, drop_zero);
        default: BBK_REQUIRE(false, BBK_ERR_ARG, "unsupported key width %d", W);
    }
    return 0;
}

}  // namespace bbk

// ---- context (C ABI) -------------------------------------------------------------------------
void bbk_ctx::resolve_pending() {
    for (auto &p : pending) {
        (void)hipEventSynchronize(p.b);
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            auto &s = stats[p.family];
            s.ms += ms;
            s.launches += 1;
            s.bytes += p.bytes;
        }
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    pending.clear();
}

namespace bbk {
const char *get_error();
}

extern "C" {

const char *bbk_last_error(void) { return bbk::get_error(); }
const char *bbk_version(void) { return "bbk 0.1 (gfx950)"; }

// which XCDs the workgroups of a launch land on (one bit per HW_REG_XCC_ID value seen)
__global__ void k_xcd_probe(uint32_t *mask) {
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) atomicOr(mask, 1u << (xcc & 15u));
}

int bbk_ctx_create(int device, bbk_ctx **out) {
    return bbk::guarded([&] {
        BBK_REQUIRE(out != nullptr, BBK_ERR_ARG, "bbk_ctx_create: out is NULL");
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        BBK_REQUIRE(e == hipSuccess && ndev > 0, BBK_ERR_HIP,
                    "bbk_ctx_create: no HIP device available (%s); this engine has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        BBK_REQUIRE(device >= 0 && device < ndev, BBK_ERR_ARG, "bbk_ctx_create: device %d out of range [0,%d)", device,
                    ndev);
        BBK_HIP(hipSetDevice(device));
        bbk_ctx *c = new bbk_ctx();
        c->device = device;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cus = prop.multiProcessorCount;
        BBK_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
        {
            uint32_t *d_mask = nullptr, h_mask = 0;
            if (hipMalloc(&d_mask, 4) == hipSuccess) {
                (void)hipMemsetAsync(d_mask, 0, 4, c->stream);
                hipLaunchKernelGGL(k_xcd_probe, dim3(4096), dim3(64), 0, c->stream, d_mask);
                if (hipMemcpyAsync(&h_mask, d_mask, 4, hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
                    hipStreamSynchronize(c->stream) == hipSuccess && h_mask)
                    c->num_xcds = __builtin_popcount(h_mask);
                (void)hipFree(d_mask);
            }
            (void)hipGetLastError();
        }
        *out = c;
    });
}

int bbk_ctx_destroy(bbk_ctx *ctx) {
    if (!ctx) return BBK_OK;
    (void)hipSetDevice(ctx->device);
    ctx->resolve_pending();
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    bbk::pool_trim(ctx->device);
    if (getenv("BBK_VERBOSE")) bbk::pool_report();
    if (ctx->pinned[0]) (void)hipHostFree(ctx->pinned[0]);
    if (ctx->pinned[1]) (void)hipHostFree(ctx->pinned[1]);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return BBK_OK;
}

int bbk_ctx_set_stream(bbk_ctx *ctx, void *hip_stream) {
    return bbk::guarded([&] {
        BBK_REQUIRE(ctx != nullptr, BBK_ERR_ARG, "bbk_ctx_set_stream: ctx is NULL");
        // drain the old stream: blocks this context has released (their last use queued on it) may be handed out
        // again for work on the new one
        BBK_HIP(hipSetDevice(ctx->device));
        BBK_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->own_stream && ctx->stream) BBK_HIP(hipStreamDestroy(ctx->stream));
        ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
        ctx->own_stream = false;
    });
}

int bbk_ctx_trim(bbk_ctx *ctx) {
    return bbk::guarded([&] {
        BBK_REQUIRE(ctx != nullptr, BBK_ERR_ARG, "bbk_ctx_trim: ctx is NULL");
        BBK_HIP(hipSetDevice(ctx->device));
        BBK_HIP(hipStreamSynchronize(ctx->stream));
        bbk::pool_trim(ctx->device);
    });
}

int bbk_ctx_memory_stats(bbk_ctx *ctx, uint64_t *mapped_now, uint64_t *mapped_total, double *map_seconds,
                         uint64_t *device_free, uint64_t *device_total) {
    return bbk::guarded([&] {
        BBK_REQUIRE(ctx != nullptr, BBK_ERR_ARG, "bbk_ctx_memory_stats: ctx is NULL");
        BBK_HIP(hipSetDevice(ctx->device));
        bbk::pool_stats(ctx->device, mapped_now, mapped_total, map_seconds);
        if (device_free || device_total) {
            size_t f = 0, t = 0;
            BBK_HIP(hipMemGetInfo(&f, &t));
            if (device_free) *device_free = f;
            if (device_total) *device_total = t;
        }
    });
}

int bbk_ctx_device_info(bbk_ctx *ctx, int *num_cus, int *num_xcds) {
    return bbk::guarded([&] {
        BBK_REQUIRE(ctx != nullptr, BBK_ERR_ARG, "bbk_ctx_device_info: ctx is NULL");
        if (num_cus) *num_cus = ctx->num_cus;
        if (num_xcds) *num_xcds = ctx->num_xcds;
    });
}

int bbk_ctx_synchronize(bbk_ctx *ctx) {
    return bbk::guarded([&] {
        BBK_REQUIRE(ctx != nullptr, BBK_ERR_ARG, "bbk_ctx_synchronize: ctx is NULL");
        BBK_HIP(hipStreamSynchronize(ctx->stream));
    });
}

int bbk_ctx_profile_enable(bbk_ctx *ctx, int on) {
    if (!ctx) return BBK_ERR_ARG;
    ctx->profiling = on != 0;
    return BBK_OK;
}

int bbk_ctx_profile_reset(bbk_ctx *ctx) {
    if (!ctx) return BBK_ERR_ARG;
    ctx->resolve_pending();
    ctx->stats.clear();
    return BBK_OK;
}

int bbk_ctx_profile_get(bbk_ctx *ctx, const char *family, double *ms_total, uint64_t *launches, double *bytes_total) {
    if (!ctx || !family) return BBK_ERR_ARG;
    ctx->resolve_pending();
    auto it = ctx->stats.find(family);
    bbk::FamilyStat s;
    if (it != ctx->stats.end()) s = it->second;
    if (ms_total) *ms_total = s.ms;
    if (launches) *launches = s.launches;
    if (bytes_total) *bytes_total = s.bytes;
    return BBK_OK;
}

unsigned bbk_words(unsigned k) { return bbk::words_of(k); }

}  // extern "C"
