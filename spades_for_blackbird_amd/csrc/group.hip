// group.hip -- several GPUs of one node in ONE process: a context per device (each driven by its own host thread),
// and the one exchange step of the path between them.
//
// The reference tools are one process for the whole job (projects/kmercount/main.cpp:186-228,
// projects/gbuilder/main.cpp:89-233) with the k-mers spread over hash buckets (KMerSegmentPolicy,
// utils/kmer_mph/kmer_buckets.hpp:28-33) that worker threads own.  Here the owner of a canonical k-mer is a GPU:
// owner(key) = mulhi(mix(key), ndev).  Every rank deduplicates its share of the reads locally, groups the distinct
// canonical records by owner (one stable counting pass) and sends each group to its owner -- the single all-to-all of
// ncclGroupStart; (ndev-1) x ncclSend + ncclRecv per rank; ncclGroupEnd over xGMI, no message above
// 256 MiB -- which merge-uniques what it receives.  Payloads (multiplicities, InOutMask bits) travel with their k-mer.
//
// RCCL is loaded at run time (dlopen), only when a group with the RCCL exchange is created: a single-GPU run never
// pays for loading it.  BBK_EXCHANGE_COPY moves the same segments with peer copies instead (hipMemcpyPeerAsync: also
// xGMI between the GPUs of a node); it exists so that the whole multi-rank path -- partition, message rounds, merge,
// output -- can be run with several ranks on ONE device, where RCCL refuses to form a communicator.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <vector>

#include <rccl/rccl.h>  // types only: the library is dlopen'ed

#include "bbk_internal.h"

namespace bbk {
namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool load() {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) return false;
        auto sym = [&](const char *n) { return dlsym(lib, n); };
        CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        Send = (decltype(Send))sym("ncclSend");
        Recv = (decltype(Recv))sym("ncclRecv");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        return CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv && GetErrorString;
    }
};

#define BBK_NCCL(g, expr)                                                                                       \
    do {                                                                                                        \
        ncclResult_t r__ = (expr);                                                                              \
        if (r__ != ncclSuccess) {                                                                               \
            set_error("%s failed: %s", #expr, (g)->rccl.GetErrorString ? (g)->rccl.GetErrorString(r__) : "?"); \
            throw Error{BBK_ERR_HIP};                                                                           \
        }                                                                                                       \
    } while (0)

}  // namespace
}  // namespace bbk

struct bbk_group {
    int n = 0;
    unsigned exchange = BBK_EXCHANGE_RCCL;
    std::vector<int> devices;
    bbk::Rccl rccl;
    std::vector<ncclComm_t> comms;
    size_t max_msg = 256ull << 20;  // bytes per message (RCCL 2.26 drops the tail of self-copies above 1 GiB: DESIGN 6)
    // rendezvous of the rank threads
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    int failed = 0;  // a rank gave up inside a collective: the others must not wait for it for ever
    // what the ranks publish for each other during an exchange
    std::vector<std::vector<uint64_t>> counts;  // [src][dst] records
    std::vector<const void *> send_keys, send_vals;
    std::vector<hipStream_t> streams;
    std::vector<uint64_t> gather_n;
    std::vector<const void *> gather_keys, gather_vals;

    // all ranks arrive, the last one releases them; false if some rank has failed
    bool barrier() {
        std::unique_lock<std::mutex> lk(mu);
        const uint64_t gen = generation;
        if (++arrived == n) {
            arrived = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return generation != gen || failed; });
        }
        return !failed;
    }
    void fail() {
        std::lock_guard<std::mutex> lk(mu);
        failed = 1;
        cv.notify_all();
    }
};

using namespace bbk;

namespace {

// The exchange proper: rank `rank` holds `send` grouped by owner (counts[rank][*] published); returns the records
// it owns in recv (peers' segments, then its own).  vbytes = 0: no payload.
void exchange_segments(bbk_group *g, int rank, bbk_ctx *ctx, size_t rec, const DevBuf &send_k, const DevBuf &send_v,
                       bool with_vals, DevBuf &recv_k, DevBuf &recv_v, uint64_t &n_recv) {
    const int n = g->n;
    std::vector<uint64_t> sstart(n + 1, 0), rstart(n + 1, 0);
    for (int j = 0; j < n; ++j) sstart[j + 1] = sstart[j] + g->counts[rank][j];
    // receive layout: peers in rank order, own segment last
    uint64_t o = 0;
    for (int j = 0; j < n; ++j) {
        if (j == rank) continue;
        rstart[j] = o;
        o += g->counts[j][rank];
    }
    rstart[rank] = o;
    const uint64_t own = g->counts[rank][rank];
    n_recv = o + own;
    recv_k.alloc(n_recv * rec + 16);
    if (with_vals) recv_v.alloc(n_recv * 4 + 16);
    if (own) {
        BBK_HIP(bbk::copy_async(recv_k.as<char>() + rstart[rank] * rec, send_k.as<char>() + sstart[rank] * rec, own * rec,
                               hipMemcpyDeviceToDevice, ctx->stream));
        if (with_vals)
            BBK_HIP(bbk::copy_async(recv_v.as<uint32_t>() + rstart[rank], send_v.as<uint32_t>() + sstart[rank], own * 4,
                                   hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (n > 1 && g->exchange == BBK_EXCHANGE_RCCL) {
        // message rounds: every (src, dst) pair cuts its segment into pieces of at most max_msg bytes; round t carries
        // piece t of every pair that has one.  All ranks read the same count matrix, so they agree on the rounds.
        const uint64_t cap = std::max<uint64_t>(1, g->max_msg / rec);
        uint64_t max_seg = 0;
        for (int a = 0; a < n; ++a)
            for (int b = 0; b < n; ++b)
                if (a != b) max_seg = std::max(max_seg, g->counts[a][b]);
        const uint64_t rounds = (max_seg + cap - 1) / cap;
        for (uint64_t t = 0; t < rounds; ++t) {
            const uint64_t lo = t * cap;
            BBK_NCCL(g, g->rccl.GroupStart());
            for (int p = 0; p < n; ++p) {
                if (p == rank) continue;
                const uint64_t sc = g->counts[rank][p], rc = g->counts[p][rank];
                if (sc > lo) {
                    const uint64_t c = std::min(cap, sc - lo);
                    BBK_NCCL(g, g->rccl.Send(send_k.as<char>() + (sstart[p] + lo) * rec, c * rec, ncclUint8, p, g->comms[rank],
                                             ctx->stream));
                    if (with_vals)
                        BBK_NCCL(g, g->rccl.Send(send_v.as<uint32_t>() + sstart[p] + lo, c, ncclUint32, p, g->comms[rank],
                                                 ctx->stream));
                }
                if (rc > lo) {
                    const uint64_t c = std::min(cap, rc - lo);
                    BBK_NCCL(g, g->rccl.Recv(recv_k.as<char>() + (rstart[p] + lo) * rec, c * rec, ncclUint8, p, g->comms[rank],
                                             ctx->stream));
                    if (with_vals)
                        BBK_NCCL(g, g->rccl.Recv(recv_v.as<uint32_t>() + rstart[p] + lo, c, ncclUint32, p, g->comms[rank],
                                                 ctx->stream));
                }
            }
            BBK_NCCL(g, g->rccl.GroupEnd());
        }
        BBK_HIP(hipStreamSynchronize(ctx->stream));
    } else if (n > 1) {
        // peer copies: every rank PULLS its segments out of the peers' send buffers (published below)
        for (int p = 0; p < n; ++p) {
            if (p == rank) continue;
            const uint64_t c = g->counts[p][rank];
            if (!c) continue;
            uint64_t pstart = 0;
            for (int j = 0; j < rank; ++j) pstart += g->counts[p][j];
            const uint64_t cap = std::max<uint64_t>(1, g->max_msg / rec);
            for (uint64_t q = 0; q < c; q += cap) {  // pieces: see bbk::copy_async
                const uint64_t m = std::min(cap, c - q);
                BBK_HIP(hipMemcpyPeerAsync(recv_k.as<char>() + (rstart[p] + q) * rec, g->devices[rank],
                                           (const char *)g->send_keys[p] + (pstart + q) * rec, g->devices[p], m * rec, ctx->stream));
                if (with_vals)
                    BBK_HIP(hipMemcpyPeerAsync(recv_v.as<uint32_t>() + rstart[p] + q, g->devices[rank],
                                               (const uint32_t *)g->send_vals[p] + pstart + q, g->devices[p], m * 4, ctx->stream));
            }
        }
        BBK_HIP(hipStreamSynchronize(ctx->stream));
    } else {
        BBK_HIP(hipStreamSynchronize(ctx->stream));
    }
}

// local distinct canonical set (optional payload) -> the records this rank owns, as raw device arrays
void exchange_set(bbk_group *g, int rank, bbk_ctx *ctx, const bbk_kmerset *local, DevBuf &recv_k, DevBuf &recv_v,
                  uint64_t &n_recv, bool &with_vals) {
    BBK_REQUIRE(g && ctx && local && rank >= 0 && rank < g->n, BBK_ERR_ARG, "bbk_group exchange: bad argument");
    BBK_REQUIRE(local->flags & BBK_CANONICAL, BBK_ERR_ARG,
                "bbk_group exchange: the owner of a k-mer is defined on canonical k-mers (BBK_CANONICAL set expected)");
    BBK_HIP(hipSetDevice(ctx->device));
    const size_t rec = (size_t)local->W * 8;
    with_vals = local->has_counts;
    DevBuf send_k(local->n * rec + 16), send_v;
    if (with_vals) send_v.alloc(local->n * 4 + 16);
    std::vector<uint64_t> cnt((size_t)g->n, 0);
    bool ok = true;
    try {
        const int rc = bbk_kmerset_export_by_owner(ctx, local, (unsigned)g->n, send_k.p, with_vals ? send_v.p : nullptr, cnt.data());
        if (rc != BBK_OK) throw Error{rc};
        BBK_HIP(hipStreamSynchronize(ctx->stream));
    } catch (...) {
        ok = false;
    }
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->counts[rank] = cnt;
        g->send_keys[rank] = send_k.p;
        g->send_vals[rank] = send_v.p;
    }
    if (!ok) g->fail();
    if (!g->barrier() || !ok) {  // counts and send buffers of every rank are published
        if (ok) set_error("bbk_group exchange: another rank failed");
        throw Error{BBK_ERR_INTERNAL};
    }
    try {
        exchange_segments(g, rank, ctx, rec, send_k, send_v, with_vals, recv_k, recv_v, n_recv);
    } catch (...) {
        g->fail();
        throw;
    }
    // nobody frees its send buffer while a peer may still be reading it
    if (!g->barrier()) {
        set_error("bbk_group exchange: another rank failed");
        throw Error{BBK_ERR_INTERNAL};
    }
}

// (keys, u32 payload) arrays of every rank -> concatenated on rank dst (rk, rv, total); collective.  my_rc: the status of
// this rank's preparation (a rank that failed still takes part in the rendezvous so that nobody waits for ever)
void gather_arrays(bbk_group *g, int rank, bbk_ctx *ctx, size_t rec, const DevBuf &sk, const DevBuf &sv, uint64_t mine, int my_rc,
                   int dst, DevBuf &rk, DevBuf &rv, uint64_t &total) {
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->gather_n[rank] = mine;
        g->gather_keys[rank] = sk.p;
        g->gather_vals[rank] = sv.p;
    }
    if (my_rc == BBK_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) my_rc = BBK_ERR_HIP;
    if (my_rc != BBK_OK) g->fail();
    if (!g->barrier() || my_rc != BBK_OK) {
        if (my_rc == BBK_OK) set_error("bbk_group gather: another rank failed");
        throw Error{my_rc != BBK_OK ? my_rc : BBK_ERR_INTERNAL};
    }
    total = 0;
    try {
        if (rank == dst) {
            for (int p = 0; p < g->n; ++p) total += g->gather_n[p];
            rk.alloc(total * rec + 16);
            rv.alloc(total * 4 + 16);
        }
        const uint64_t cap = std::max<uint64_t>(1, g->max_msg / rec);
        if (g->n > 1 && g->exchange == BBK_EXCHANGE_RCCL) {
            uint64_t max_seg = 0;
            for (int p = 0; p < g->n; ++p)
                if (p != dst) max_seg = std::max(max_seg, g->gather_n[p]);
            const uint64_t rounds = (max_seg + cap - 1) / cap;
            for (uint64_t t = 0; t < rounds; ++t) {
                const uint64_t lo = t * cap;
                BBK_NCCL(g, g->rccl.GroupStart());
                if (rank == dst) {
                    uint64_t o = 0;
                    for (int p = 0; p < g->n; ++p) {
                        if (p != dst && g->gather_n[p] > lo) {
                            const uint64_t c = std::min(cap, g->gather_n[p] - lo);
                            BBK_NCCL(g, g->rccl.Recv(rk.as<char>() + (o + lo) * rec, c * rec, ncclUint8, p, g->comms[rank], ctx->stream));
                            BBK_NCCL(g, g->rccl.Recv(rv.as<uint32_t>() + o + lo, c, ncclUint32, p, g->comms[rank], ctx->stream));
                        }
                        o += g->gather_n[p];
                    }
                } else if (mine > lo) {
                    const uint64_t c = std::min(cap, mine - lo);
                    BBK_NCCL(g, g->rccl.Send(sk.as<char>() + lo * rec, c * rec, ncclUint8, dst, g->comms[rank], ctx->stream));
                    BBK_NCCL(g, g->rccl.Send(sv.as<uint32_t>() + lo, c, ncclUint32, dst, g->comms[rank], ctx->stream));
                }
                BBK_NCCL(g, g->rccl.GroupEnd());
            }
        }
        if (rank == dst) {
            uint64_t o = 0;
            for (int p = 0; p < g->n; ++p) {
                const uint64_t c = g->gather_n[p];
                if (c && (p == dst || g->exchange == BBK_EXCHANGE_COPY)) {
                    for (uint64_t q = 0; q < c; q += cap) {  // pieces: see bbk::copy_async
                        const uint64_t m = std::min(cap, c - q);
                        BBK_HIP(hipMemcpyPeerAsync(rk.as<char>() + (o + q) * rec, g->devices[rank],
                                                   (const char *)g->gather_keys[p] + q * rec, g->devices[p], m * rec, ctx->stream));
                        BBK_HIP(hipMemcpyPeerAsync(rv.as<uint32_t>() + o + q, g->devices[rank],
                                                   (const uint32_t *)g->gather_vals[p] + q, g->devices[p], m * 4, ctx->stream));
                    }
                }
                o += c;
            }
        }
        BBK_HIP(hipStreamSynchronize(ctx->stream));
    } catch (...) {
        g->fail();
        throw;
    }
    if (!g->barrier()) {  // the sources keep their buffers until dst has read them
        set_error("bbk_group gather: another rank failed");
        throw Error{BBK_ERR_INTERNAL};
    }
}

}  // namespace

extern "C" {

int bbk_group_create(const int *devices, int ndev, unsigned exchange, bbk_group **out) {
    return guarded([&] {
        BBK_REQUIRE(devices && out && ndev >= 1 && ndev <= 64, BBK_ERR_ARG, "bbk_group_create: bad argument (ndev=%d)", ndev);
        BBK_REQUIRE(exchange == BBK_EXCHANGE_RCCL || exchange == BBK_EXCHANGE_COPY, BBK_ERR_ARG,
                    "bbk_group_create: unknown exchange %u", exchange);
        int have = 0;
        BBK_HIP(hipGetDeviceCount(&have));
        for (int i = 0; i < ndev; ++i)
            BBK_REQUIRE(devices[i] >= 0 && devices[i] < have, BBK_ERR_ARG, "bbk_group_create: device %d out of range [0,%d)",
                        devices[i], have);
        auto g = std::make_unique<bbk_group>();
        g->n = ndev;
        g->exchange = exchange;
        g->devices.assign(devices, devices + ndev);
        g->counts.assign((size_t)ndev, std::vector<uint64_t>((size_t)ndev, 0));
        g->send_keys.assign((size_t)ndev, nullptr);
        g->send_vals.assign((size_t)ndev, nullptr);
        g->gather_n.assign((size_t)ndev, 0);
        g->gather_keys.assign((size_t)ndev, nullptr);
        g->gather_vals.assign((size_t)ndev, nullptr);
        if (const char *e = getenv("BBK_GROUP_MAX_MSG")) g->max_msg = std::max<size_t>(64, strtoull(e, nullptr, 10));  // tests
        if (exchange == BBK_EXCHANGE_RCCL) {
            for (int i = 0; i < ndev; ++i)
                for (int j = 0; j < i; ++j)
                    BBK_REQUIRE(devices[i] != devices[j], BBK_ERR_ARG,
                                "bbk_group_create: RCCL needs distinct devices (device %d listed twice); use "
                                "BBK_EXCHANGE_COPY to run several ranks on one device", devices[i]);
            BBK_REQUIRE(g->rccl.load(), BBK_ERR_HIP, "bbk_group_create: cannot load librccl (%s)", dlerror());
            g->comms.assign((size_t)ndev, nullptr);
            BBK_NCCL(g.get(), g->rccl.CommInitAll(g->comms.data(), ndev, devices));
        } else {
            // peer access for the copies between distinct devices (already enabled / same device: fine)
            for (int i = 0; i < ndev; ++i)
                for (int j = 0; j < ndev; ++j)
                    if (devices[i] != devices[j]) {
                        (void)hipSetDevice(devices[i]);
                        (void)hipDeviceEnablePeerAccess(devices[j], 0);
                        (void)hipGetLastError();
                    }
        }
        *out = g.release();
    });
}

int bbk_group_size(const bbk_group *g) { return g ? g->n : 0; }
int bbk_group_device(const bbk_group *g, int rank) { return g && rank >= 0 && rank < g->n ? g->devices[rank] : -1; }

void bbk_group_destroy(bbk_group *g) {
    if (!g) return;
    for (ncclComm_t c : g->comms)
        if (c && g->rccl.CommDestroy) (void)g->rccl.CommDestroy(c);
    delete g;
}

void bbk_group_abort(bbk_group *g) {
    if (g) g->fail();
}

int bbk_group_exchange_kmers(bbk_group *g, int rank, bbk_ctx *ctx, const bbk_kmerset *local, unsigned flags,
                             bbk_kmerset **shard) {
    return guarded([&] {
        BBK_REQUIRE(shard != nullptr, BBK_ERR_ARG, "bbk_group_exchange_kmers: shard is NULL");
        DevBuf rk, rv;
        uint64_t n = 0;
        bool with_vals = false;
        exchange_set(g, rank, ctx, local, rk, rv, n, with_vals);
        // merge-unique of what the ranks sent (multiplicities are summed)
        const int rc = bbk_kmerset_from_device_ex(ctx, rk.p, with_vals ? rv.p : nullptr, n, local->k, flags, shard);
        if (rc != BBK_OK) throw Error{rc};
        (*shard)->flags |= BBK_CANONICAL | (with_vals ? BBK_WITH_COUNTS : 0u) | (flags & BBK_UNSORTED);
    });
}

int bbk_group_exchange_extindex(bbk_group *g, int rank, bbk_ctx *ctx, const bbk_kmerset *local_masks, bbk_extindex **shard) {
    return guarded([&] {
        BBK_REQUIRE(shard != nullptr && local_masks && (local_masks->flags & BBK_WITH_MASKS), BBK_ERR_ARG,
                    "bbk_group_exchange_extindex: a BBK_CANONICAL | BBK_WITH_MASKS set is expected");
        DevBuf rk, rv;
        uint64_t n = 0;
        bool with_vals = false;
        exchange_set(g, rank, ctx, local_masks, rk, rv, n, with_vals);
        const int rc = bbk_extindex_from_device(ctx, rk.p, rv.p, n, local_masks->k, shard);  // ORs the masks of equal k-mers
        if (rc != BBK_OK) throw Error{rc};
    });
}

// every rank's shard of the index -> one index on rank `dst` (the unitig stage walks across owners); the other ranks
// get *full = NULL.  The shards hold disjoint k-mers: concatenation + one ordering pass.
int bbk_group_gather_extindex(bbk_group *g, int rank, bbk_ctx *ctx, const bbk_extindex *shard, int dst, bbk_extindex **full) {
    return guarded([&] {
        BBK_REQUIRE(g && ctx && shard && full && rank >= 0 && rank < g->n && dst >= 0 && dst < g->n, BBK_ERR_ARG,
                    "bbk_group_gather_extindex: bad argument");
        BBK_HIP(hipSetDevice(ctx->device));
        *full = nullptr;
        const size_t rec = (size_t)shard->W * 8;
        // masks travel as u32 (the payload layout bbk_extindex_from_device takes)
        DevBuf sk(shard->n * rec + 16), sv(shard->n * 4 + 16);
        int erc = bbk_extindex_export_u32(ctx, shard, sk.p, sv.p);
        DevBuf rk, rv;
        uint64_t total = 0;
        gather_arrays(g, rank, ctx, rec, sk, sv, shard->n, erc, dst, rk, rv, total);
        if (rank == dst) {
            const int rc = bbk_extindex_from_device(ctx, rk.p, rv.p, total, shard->k, full);
            if (rc != BBK_OK) throw Error{rc};
        }
    });
}

// the same for a sharded k-mer set with multiplicities (gbuilder -c: the (k+1)-mer counts the coverage is read from)
int bbk_group_gather_kmers(bbk_group *g, int rank, bbk_ctx *ctx, const bbk_kmerset *shard, int dst, bbk_kmerset **full) {
    return guarded([&] {
        BBK_REQUIRE(g && ctx && shard && full && rank >= 0 && rank < g->n && dst >= 0 && dst < g->n, BBK_ERR_ARG,
                    "bbk_group_gather_kmers: bad argument");
        BBK_REQUIRE(shard->has_counts, BBK_ERR_ARG, "bbk_group_gather_kmers: a set with multiplicities is expected");
        BBK_HIP(hipSetDevice(ctx->device));
        *full = nullptr;
        const size_t rec = (size_t)shard->W * 8;
        DevBuf rk, rv;
        uint64_t total = 0;
        gather_arrays(g, rank, ctx, rec, shard->keys, shard->counts, shard->n, BBK_OK, dst, rk, rv, total);
        if (rank == dst) {
            bbk_kmerset *out = nullptr;
            const int rc = bbk_kmerset_from_device_ex(ctx, rk.p, rv.p, total, shard->k, 0, &out);  // ascending, counts summed
            if (rc != BBK_OK) throw Error{rc};
            out->flags |= BBK_CANONICAL | BBK_WITH_COUNTS;
            *full = out;
        }
    });
}

}  // extern "C"
