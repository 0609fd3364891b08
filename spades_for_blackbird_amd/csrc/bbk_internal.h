// bbk_internal.h -- host-side plumbing shared by the HIP translation units (context, errors,
// device buffers, per-kernel-family HIP-event timing).  Not part of the C ABI.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/bbk.h"

namespace bbk {

void set_error(const char *fmt, ...);

struct Error {
    int code;
};

#define BBK_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess) {                                                               \
            ::bbk::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, \
                             __LINE__);                                                        \
            throw ::bbk::Error{BBK_ERR_HIP};                                                   \
        }                                                                                      \
    } while (0)

#define BBK_REQUIRE(cond, code, ...)         \
    do {                                     \
        if (!(cond)) {                       \
            ::bbk::set_error(__VA_ARGS__);   \
            throw ::bbk::Error{code};        \
        }                                    \
    } while (0)

// Runs f(), maps exceptions to a status code: no exception crosses the C ABI.
template <class F>
int guarded(F &&f) {
    try {
        f();
        return BBK_OK;
    } catch (const Error &e) {
        return e.code;
    } catch (const std::bad_alloc &) {
        set_error("host allocation failed");
        return BBK_ERR_NOMEM;
    } catch (const std::exception &e) {
        set_error("internal error: %s", e.what());
        return BBK_ERR_INTERNAL;
    }
}

void *pool_alloc(size_t bytes, size_t *granted, int *device);  // on the current device
void pool_free(void *p, size_t bytes, int device);
void pool_trim(int device);  // gives the free memory this (device, host thread) holds back to the driver
void pool_report();          // allocator statistics to stderr
size_t pool_mapped_bytes(int device);
void pool_stats(int device, uint64_t *mapped_now, uint64_t *mapped_total, double *map_seconds);  // calling thread's arenas  // device memory the arenas of all host threads hold mapped on `device`

// Owning device buffer.
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int dev = 0;  // device the block lives on (it returns to that device's free list)
    DevBuf() = default;
    explicit DevBuf(size_t n) { alloc(n); }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), bytes(o.bytes), dev(o.dev) {
        o.p = nullptr;
        o.bytes = 0;
    }
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) {
            release();
            p = o.p;
            bytes = o.bytes;
            dev = o.dev;
            o.p = nullptr;
            o.bytes = 0;
        }
        return *this;
    }
    ~DevBuf() { release(); }
    // Device memory comes from a caching pool (primitives.hip) keyed by (device, host thread): hipMalloc/hipFree of
    // multi-GB buffers cost far more than the kernels that use them, and every step of the path
    // asks for the same sizes again.  All work of a context is issued on one stream, so handing a
    // released block to the next request of the same context is stream-ordered and needs no synchronisation.
    void alloc(size_t n) {
        release();
        if (n == 0) n = 16;
        size_t got = 0;
        p = pool_alloc(n, &got, &dev);
        bytes = got;
    }
    void release() {
        if (p) pool_free(p, bytes, dev);
        p = nullptr;
        bytes = 0;
    }
    template <class T>
    T *as() const {
        return reinterpret_cast<T *>(p);
    }
};

// Device-to-device / default-kind copies of the library: one place to change how they are done.
hipError_t copy_async(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t stream);
bool pool_owns(const void *p);  // the address lies in an arena of the allocator

// std::vector whose resize() does not zero-fill (multi-hundred-MB host buffers that are about to
// be overwritten by a device-to-host copy)
template <class T>
struct default_init_alloc : std::allocator<T> {
    template <class U>
    struct rebind {
        using other = default_init_alloc<U>;
    };
    template <class U, class... A>
    void construct(U *p, A &&...a) {
        if constexpr (sizeof...(A) == 0) ::new ((void *)p) U;
        else ::new ((void *)p) U(std::forward<A>(a)...);
    }
};
template <class T>
using raw_vector = std::vector<T, default_init_alloc<T>>;

struct FamilyStat {
    double ms = 0;
    uint64_t launches = 0;
    double bytes = 0;
};

}  // namespace bbk

struct bbk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool profiling = false;
    int num_cus = 256;
    // XCDs the workgroups of a launch are seen to run on (HW_REG_XCC_ID of a probe launch at context creation): the
    // per-XCD fill fronts of the partition kernels assume eight; anything else (a partitioned device) turns them off
    int num_xcds = 8;
    // pending (start, stop, family, bytes) event pairs; resolved lazily
    struct Pending {
        hipEvent_t a, b;
        std::string family;
        double bytes;
    };
    std::vector<Pending> pending;
    std::map<std::string, bbk::FamilyStat> stats;
    void resolve_pending();
    // event counters of the path (read back like a kernel family: launches = events, bytes_total = summed value)
    void add_stat(const char *name, double value) {
        if (!profiling) return;
        bbk::FamilyStat &f = stats[name];
        f.launches += 1;
        f.bytes += value;
    }
    // pinned staging for large device-to-host copies (pageable copies run at a fraction of PCIe)
    void *pinned[2] = {nullptr, nullptr};
    size_t pinned_bytes = 0;
    // super-k-mer stage A (superk.hip): instances / distinct keys of the last batch it finished (0: none yet).  The
    // buckets of the next batch are planned for that multiplicity instead of the worst case 1
    double superk_dup = 0;
};

namespace bbk {

// RAII timer around one kernel launch (or a short run of launches) of a named family.
struct KernelTimer {
    bbk_ctx *ctx;
    hipEvent_t a = nullptr, b = nullptr;
    std::string family;
    double bytes;
    KernelTimer(bbk_ctx *c, const char *fam, double algorithmic_bytes = 0) : ctx(c), family(fam), bytes(algorithmic_bytes) {
        if (ctx->profiling) {
            BBK_HIP(hipEventCreate(&a));
            BBK_HIP(hipEventCreate(&b));
            BBK_HIP(hipEventRecord(a, ctx->stream));
        }
    }
    ~KernelTimer() {
        if (a) {
            (void)hipEventRecord(b, ctx->stream);
            ctx->pending.push_back({a, b, family, bytes});
        }
    }
};

inline void check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("kernel launch %s failed: %s", what, hipGetErrorString(e));
        throw Error{BBK_ERR_HIP};
    }
}

inline unsigned words_of(unsigned k) { return (k + 31) >> 5; }

// One thread per item over billions of items.  A launch whose x dimension holds 2^32 threads or more is neither refused
// nor run in full: the thread count is taken modulo 2^32 and the rest of the items is silently skipped (an index of
// 4 968 413 780 k-mers had its last 4.29 G masks never converted, its start edges never counted:
// profiles/r03/d2d_copy_above_4GiB.log -- the first untouched item is n - 2^32).  Per-item kernels therefore get their
// blocks in two dimensions (grid_blocks) and read their item through BBK_GID(); every one of them checks it against n.
#define BBK_GID() ((((uint64_t)blockIdx.y * gridDim.x) + blockIdx.x) * blockDim.x + threadIdx.x)
inline dim3 grid_blocks(uint64_t blocks) {
    constexpr uint64_t kMaxX = 1ull << 21;  // x 1024 threads at most: 2^31 threads in x
    if (blocks <= kMaxX) return dim3((unsigned)(blocks ? blocks : 1));
    return dim3((unsigned)kMaxX, (unsigned)((blocks + kMaxX - 1) / kMaxX));
}

// Large device -> host copy through the context's pinned staging buffers (chunked, two buffers:
// the copy of chunk i+1 overlaps the host memcpy of chunk i).
void d2h_big(bbk_ctx *ctx, void *dst, const void *src, size_t bytes);

// Device -> file: chunks through the pinned staging buffers (copy of chunk i+1 overlaps the positional writes of
// chunk i, several writers per chunk).  Returns false on a short write.
bool d2f_big(bbk_ctx *ctx, int fd, uint64_t file_off, const void *src, size_t bytes);

// ---- primitives.hip -------------------------------------------------------------------------
// One LSD pass selector: kind 0 = bits [shift, shift+bits) of key word `word`;
// kind 1 = XXH3 bucket (kmer_buckets.hpp:28-33) with nb <= 256 buckets;
// kind 2 = owner(mix(key), nb) for the multi-GPU partition.
struct PassDesc {
    int kind;
    int word;
    int shift;
    int bits;
    unsigned nb;
};

// Key-bit passes that sort records of W words into the reference order (word 0 most
// significant, adt/array_vector.hpp:114-123) given that only the low 2k bits are populated.
std::vector<PassDesc> key_passes(unsigned k);

// Stable LSD radix sort of n records (W u64 words each, optional u32 payload). keys/vals are
// sorted in place; tmp buffers must hold n records.  n < 2^32.
void sort_records(bbk_ctx *ctx, int W, void *keys, void *keys_tmp, uint32_t *vals, uint32_t *vals_tmp, uint64_t n,
                  const std::vector<PassDesc> &passes);
// Per-bin record counts of one pass (256 bins, written to h_counts), e.g. bucket sizes.
// one stable counting pass src -> dst (device buffers, not aliased) on the digit `pd`; h_digit_totals (256 entries,
// optional) receives the number of records of every digit value
void partition_records(bbk_ctx *ctx, int W, const void *src, void *dst, const uint32_t *vsrc, uint32_t *vdst, uint64_t n,
                       const PassDesc &pd, uint64_t *h_digit_totals = nullptr);

enum ReduceOp { REDUCE_COUNT = 0, REDUCE_SUM = 1, REDUCE_OR = 2 };
// Unique of a sorted record array: distinct keys to out_keys (may alias nothing), per-key reduction of
// vals (COUNT: run length, SUM: sum of vals, OR: or of vals) to out_vals (may be null for COUNT-less).
// Returns the number of distinct keys.  drop_zero: omit keys whose reduced value is 0 (OR mode).
uint64_t unique_records(bbk_ctx *ctx, int W, const void *keys, const uint32_t *vals, uint64_t n, void *out_keys,
                        uint32_t *out_vals, ReduceOp op, bool drop_zero);
// Exclusive scan of n u64 values (in place allowed); returns the total.
uint64_t exclusive_scan_u64(bbk_ctx *ctx, const uint64_t *in, uint64_t *out, uint64_t n);

}  // namespace bbk

struct bbk_reads {
    bbk_ctx *ctx = nullptr;
    uint64_t n = 0;        // reads
    uint64_t n_words = 0;  // packed words
    uint64_t bases = 0;    // total bases (sum of len)
    const uint64_t *d_words = nullptr;
    const uint64_t *d_woff = nullptr;  // n+1
    const uint32_t *d_len = nullptr;   // n
    bbk::DevBuf own_words, own_woff, own_len;
};

struct bbk_kmerset {
    unsigned k = 0, W = 0;
    unsigned flags = 0;
    uint64_t n = 0;          // distinct records
    uint64_t instances = 0;  // k-mer instances that entered the sort
    bbk::DevBuf keys;        // n * W u64, ascending
    bbk::DevBuf counts;      // n u32 (optional)
    bool has_counts = false;
    bool sorted = true;      // false: distinct but in hash-bucket order (BBK_UNSORTED)
    bool ref_order = false;  // true: final_kmers order (16 XXH3 buckets, ascending inside) instead of ascending
};

struct bbk_extindex {
    unsigned k = 0, W = 0;
    uint64_t n = 0;
    uint64_t instances = 0;
    bbk::DevBuf keys;   // n * W u64 ascending (canonical k-mers)
    bbk::DevBuf masks;  // n u8
    bbk::DevBuf prefix; // lookup accelerator: (1<<prefix_bits)+1 entries, u32 (u64 when prefix_wide: 2^32-2 k-mers or more)
    unsigned prefix_bits = 0;
    bool prefix_wide = false;
};
