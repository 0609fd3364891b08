// What bounds the level-1 partition kernel when its arithmetic is taken away: the per-tile reservations (one returning
// global atomic per bin and tile, all tiles on the same 1024 counters) or the scattered stores (runs of ~15 4-byte
// records into 1024 segments)?  Same launch shape as k_part_reads_narrow at BASELINE configs[1]:
// 84 640 tiles of 15 360 records, workgroups of 1024 threads (one thread per bin for the reservation).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NT = 1024, ITEMS = 15, NB = 1024;

struct P {
    int reserve;       // 0: offsets computed from the tile number, 1: returning atomic per bin and tile, 2: atomics only
    uint32_t gstride;  // counters: group of 32 (one 128-byte line) every gstride bytes
    int xcd;           // 1: counters and sub-slots per XCD
    int tiles_per_wg;  // consecutive tiles one workgroup handles (one reservation for all of them when reserve = 1)
    int nt;            // 1: nontemporal stores
    int order;         // computed offsets: 0 tile order, 1 each XCD fills its own eighth of every segment
    uint32_t tiles;
    uint64_t slot;     // records of a segment
};

__global__ __launch_bounds__(NT) void k(uint32_t *cur, uint32_t *out, P p) {
    __shared__ uint32_t goff[NB];
    const uint32_t tid = threadIdx.x;
    uint32_t xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    const uint32_t T = (uint32_t)p.tiles_per_wg;
    const uint32_t tile0 = blockIdx.x * T;
    const uint32_t sub = (uint32_t)(p.slot / 8);
    uint32_t g;
    if (p.reserve) {
        const uint32_t ci = p.xcd ? tid * 8 + xcc : tid;
        uint32_t *c = (uint32_t *)((char *)cur + (size_t)(ci >> 5) * p.gstride + (ci & 31u) * 4);
        g = atomicAdd(c, (uint32_t)ITEMS * T);
        if (p.xcd) g += xcc * sub;
        if (p.reserve == 2) {
            if (g == 0xFFFFFFFFu) out[0] = g;
            return;
        }
    } else if (p.order == 1) {
        // workgroup w runs on XCD w % 8 (round-robin dispatch): the j-th workgroup of an XCD takes the j-th place of its eighth
        g = (blockIdx.x & 7u) * sub + (blockIdx.x >> 3) * ITEMS * T;
    } else {
        g = tile0 * ITEMS;
    }
    goff[tid] = g;
    __syncthreads();
    for (uint32_t t = 0; t < T; ++t) {
        if (tile0 + t >= p.tiles) break;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint32_t pos = (uint32_t)i * NT + tid;
            const uint32_t b = pos / ITEMS, r = pos - b * ITEMS;
            const uint64_t at = (uint64_t)b * p.slot + (goff[b] + t * ITEMS + r);
            if (p.nt) __builtin_nontemporal_store(pos, &out[at]);
            else out[at] = pos;
        }
    }
}

int main(int argc, char **argv) {
    const uint32_t tiles = 84640;
    const uint64_t slot = (((uint64_t)tiles * ITEMS + 64 + 7) / 8) * 8 + 8 * 1024;  // records of one segment (+ slack per eighth)
    uint32_t *cur, *out;
    const size_t cur_bytes = (size_t)NB * 8 / 32 * 8192 + 4096;
    CK(hipMalloc(&cur, cur_bytes));
    CK(hipMalloc(&out, slot * NB * 4));
    CK(hipMemset(out, 0, slot * NB * 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    struct Case { P p; const char *what; };
    auto mk = [&](int reserve, uint32_t gstride, int xcd, int tpw, int nt, int order) {
        P p{reserve, gstride, xcd, tpw, nt, order, tiles, slot};
        return p;
    };
    const Case cases[] = {
        {mk(2, 128, 0, 1, 0, 0), "atomics only, counters dense (32 per 128 B line, lines adjacent)"},
        {mk(2, 256, 0, 1, 0, 0), "atomics only, one line of counters every 256 B"},
        {mk(2, 1024, 0, 1, 0, 0), "atomics only, one line of counters every 1 KB"},
        {mk(2, 4096, 0, 1, 0, 0), "atomics only, one line of counters every 4 KB"},
        {mk(2, 4224, 0, 1, 0, 0), "atomics only, one line of counters every 4 KB + 128 B"},
        {mk(2, 8192, 0, 1, 0, 0), "atomics only, one line of counters every 8 KB"},
        {mk(2, 128, 0, 8, 0, 0), "atomics only, one reservation per 8 tiles"},
        {mk(0, 128, 0, 1, 0, 0), "stores only, tile order"},
        {mk(0, 128, 0, 1, 1, 0), "stores only, tile order, nontemporal"},
        {mk(0, 128, 0, 1, 0, 1), "stores only, each XCD fills its own eighth of a segment"},
        {mk(0, 128, 0, 8, 0, 0), "stores only, 8 consecutive tiles per workgroup"},
        {mk(0, 128, 0, 8, 0, 1), "stores only, 8 tiles per workgroup, XCD eighths"},
        {mk(0, 128, 0, 32, 0, 1), "stores only, 32 tiles per workgroup, XCD eighths"},
        {mk(1, 128, 0, 1, 0, 0), "reservations + stores (the kernel's pattern)"},
        {mk(1, 128, 0, 1, 1, 0), "reservations + stores, nontemporal"},
        {mk(1, 128, 1, 1, 0, 0), "reservations + stores, counters and sub-slots per XCD"},
        {mk(1, 4224, 0, 1, 0, 0), "reservations (lines 4 KB + 128 B apart) + stores"},
        {mk(1, 128, 0, 8, 0, 0), "one reservation per 8 tiles + stores"},
        {mk(1, 128, 1, 8, 0, 0), "one reservation per 8 tiles + stores, per XCD"},
        {mk(1, 128, 1, 32, 0, 0), "one reservation per 32 tiles + stores, per XCD"},
    };
    for (const Case &c : cases) {
        float best = 1e9f;
        const uint32_t grid = (tiles + c.p.tiles_per_wg - 1) / c.p.tiles_per_wg;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(cur, 0, cur_bytes));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(k, dim3(grid), dim3(NT), 0, 0, cur, out, c.p);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            best = ms < best ? ms : best;
        }
        printf("%-70s %7.3f ms\n", c.what, best);
        fflush(stdout);
    }
    return 0;
}
