// Which accesses reach the NEW physical memory after hipMemUnmap + hipMemRelease + hipMemCreate + hipMemMap at a virtual
// address that was mapped before?  Round 3 found the engine's level-1 cursors (a 2 KB pageable hipMemcpyAsync
// host->device on a non-blocking stream, then atomics from a kernel, then a device->host copy) holding "counts from
// zero": the kernel and the device->host copy saw the new memory, the host->device copy did not arrive.  This probe
// repeats that on a bare reserved range, per access kind, at re-used and at fresh virtual addresses.
//   hipcc --offload-arch=gfx950 -O2 -o vmm_remap_copy_probe vmm_remap_copy_probe.hip ; ./vmm_remap_copy_probe [chunks]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void fill(unsigned *p, size_t n, unsigned tag) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (unsigned)i ^ tag;
}
__global__ void check(const unsigned *p, size_t n, unsigned tag, unsigned long long *bad) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (p[i] != ((unsigned)i ^ tag)) atomicAdd(bad, 1ull);
}
__global__ void check_const(const unsigned *p, size_t n, unsigned v, unsigned long long *bad) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (p[i] != v) atomicAdd(bad, 1ull);
}

static hipMemAllocationProp prop;
static hipMemAccessDesc acc;
static const size_t chunk = 1ull << 30;
static hipStream_t st;
static unsigned long long *bad;

static unsigned long long take_bad() {
    unsigned long long h = 0;
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
    CK(hipMemset(bad, 0, 8));
    return h;
}

// the access kinds on `n` words at device address d (inside re-mapped memory); returns number of bad words per kind
static void probe_at(const char *what, char *d, char *stable /* never unmapped */, unsigned *pinned) {
    const size_t n = 512;  // 2 KB like the engine's cursor tables
    std::vector<unsigned> h(n), back(n);
    unsigned long long r[7] = {0};
    // A: async pageable H2D, first touch of this memory
    for (size_t i = 0; i < n; ++i) h[i] = (unsigned)i ^ 0xA0A0u;
    CK(hipMemcpyAsync(d, h.data(), n * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(check, dim3(1), dim3(256), 0, st, (const unsigned *)d, n, 0xA0A0u, bad);
    r[0] = take_bad();
    // B: async pinned H2D
    for (size_t i = 0; i < n; ++i) pinned[i] = (unsigned)i ^ 0xB0B0u;
    CK(hipMemcpyAsync(d + 4096, pinned, n * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(check, dim3(1), dim3(256), 0, st, (const unsigned *)(d + 4096), n, 0xB0B0u, bad);
    r[1] = take_bad();
    // C: synchronous hipMemcpy H2D (null stream)
    for (size_t i = 0; i < n; ++i) h[i] = (unsigned)i ^ 0xC0C0u;
    CK(hipMemcpy(d + 8192, h.data(), n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(check, dim3(1), dim3(256), 0, st, (const unsigned *)(d + 8192), n, 0xC0C0u, bad);
    r[2] = take_bad();
    // D: hipMemsetAsync
    CK(hipMemsetAsync(d + 12288, 0x5A, n * 4, st));
    hipLaunchKernelGGL(check_const, dim3(1), dim3(256), 0, st, (const unsigned *)(d + 12288), n, 0x5A5A5A5Au, bad);
    r[3] = take_bad();
    // E: async D2D from memory that was never unmapped
    hipLaunchKernelGGL(fill, dim3(1), dim3(256), 0, st, (unsigned *)stable, n, 0xE0E0u);
    CK(hipMemcpyAsync(d + 16384, stable, n * 4, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(check, dim3(1), dim3(256), 0, st, (const unsigned *)(d + 16384), n, 0xE0E0u, bad);
    r[4] = take_bad();
    // F: kernel write, async pageable D2H
    hipLaunchKernelGGL(fill, dim3(1), dim3(256), 0, st, (unsigned *)(d + 20480), n, 0xF0F0u);
    CK(hipMemcpyAsync(back.data(), d + 20480, n * 4, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    for (size_t i = 0; i < n; ++i) r[5] += back[i] != ((unsigned)i ^ 0xF0F0u);
    // G: async pageable H2D again, now that kernels have touched the neighbourhood
    for (size_t i = 0; i < n; ++i) h[i] = (unsigned)i ^ 0x1111u;
    CK(hipMemcpyAsync(d + 24576, h.data(), n * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(check, dim3(1), dim3(256), 0, st, (const unsigned *)(d + 24576), n, 0x1111u, bad);
    r[6] = take_bad();
    printf("  %-34s H2D-pageable %llu  H2D-pinned %llu  H2D-sync %llu  memset %llu  D2D %llu  D2H %llu  H2D-pageable-again %llu\n",
           what, r[0], r[1], r[2], r[3], r[4], r[5], r[6]);
    fflush(stdout);
}

// full-range fill + check by kernels over the chunks listed
static unsigned long long sweep(char *base, const std::vector<size_t> &chunks, unsigned tag) {
    for (size_t c : chunks) hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, st, (unsigned *)(base + c * chunk), chunk / 4, tag + (unsigned)c);
    for (size_t c : chunks) hipLaunchKernelGGL(check, dim3(2048), dim3(256), 0, st, (const unsigned *)(base + c * chunk), chunk / 4, tag + (unsigned)c, bad);
    return take_bad();
}

// modes: reuse        unmap the upper half, map new memory at the SAME addresses             (what arena_shrink/grow did)
//        flush_before the same, with hipMalloc + hipMemset + hipFree between unmap and map
//        flush_after  the same, with hipMalloc + hipMemset + hipFree after the new mapping
//        fresh        unmap the upper half, map new memory at addresses never used before
//        rereserve    unmap everything, hipMemAddressFree, reserve again, map
//        reserve      only reports which reservation sizes the platform grants
int main(int argc, char **argv) {
    const size_t N = argc > 1 ? (size_t)atoi(argv[1]) : 8;
    const std::string mode = argc > 2 ? argv[2] : "reuse";
    const bool small = argc > 3;  // also run the small-copy probes
    CK(hipSetDevice(0));
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    if (mode == "reserve") {
        for (int tib : {256, 128, 64, 32, 16, 8, 4, 2, 1}) {
            void *p = nullptr;
            const hipError_t e = hipMemAddressReserve(&p, (size_t)tib << 40, 0, nullptr, 0);
            printf("reserve %3d TiB: %s (%p)\n", tib, hipGetErrorString(e), p);
            if (e == hipSuccess) (void)hipMemAddressFree(p, (size_t)tib << 40);
            (void)hipGetLastError();
        }
        return 0;
    }
    char *base; CK(hipMemAddressReserve((void **)&base, 1ull << 40, 0, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> h(4 * N + 8);
    auto map = [&](size_t i) { CK(hipMemCreate(&h[i], chunk, &prop, 0)); CK(hipMemMap(base + i * chunk, chunk, 0, h[i], 0)); CK(hipMemSetAccess(base + i * chunk, chunk, &acc, 1)); };
    auto unmap = [&](size_t i) { CK(hipMemUnmap(base + i * chunk, chunk)); CK(hipMemRelease(h[i])); };
    auto flush_trick = [&]() { void *o; CK(hipMalloc(&o, 64 << 20)); CK(hipMemset(o, 0xAB, 64 << 20)); CK(hipDeviceSynchronize()); CK(hipFree(o)); };
    CK(hipMalloc(&bad, 8)); CK(hipMemset(bad, 0, 8));
    unsigned *pinned; CK(hipHostMalloc((void **)&pinned, 4096, hipHostMallocDefault));
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    std::vector<size_t> live;
    for (size_t i = 0; i < N; ++i) { map(i); live.push_back(i); }
    printf("mode %s, %zu chunks; first sweep: %llu bad\n", mode.c_str(), N, sweep(base, live, 0x7000u));
    size_t fresh_next = 2 * N;
    for (int round = 0; round < 3; ++round) {
        CK(hipDeviceSynchronize());
        if (mode == "rereserve") {
            for (size_t i : live) unmap(i);
            CK(hipMemAddressFree(base, 1ull << 40));
            char *nb; CK(hipMemAddressReserve((void **)&nb, 1ull << 40, 0, nullptr, 0));
            printf("round %d: reservation %p -> %p\n", round, (void *)base, (void *)nb);
            base = nb;
            for (size_t i : live) map(i);
        } else {
            std::vector<size_t> keep(live.begin(), live.begin() + N / 2), tail(live.begin() + N / 2, live.end());
            for (size_t j = tail.size(); j-- > 0;) unmap(tail[j]);
            if (mode == "flush_before") flush_trick();
            live = keep;
            for (size_t j = 0; j < tail.size(); ++j) {
                const size_t at = mode == "fresh" ? fresh_next++ : tail[j];
                map(at);
                live.push_back(at);
            }
            if (mode == "flush_after") flush_trick();
        }
        if (small) {
            probe_at("re-mapped chunk, offset 2 MiB", base + live[N / 2] * chunk + (2 << 20), base, pinned);
            probe_at("last re-mapped chunk, 6 MiB", base + live[N - 1] * chunk + (6 << 20), base, pinned);
        }
        printf("round %d: sweep over %zu chunks (last at chunk slot %zu): %llu bad\n", round, live.size(), live.back(),
               sweep(base, live, 0x3300u + 16 * round));
        fflush(stdout);
    }
    printf("VMM-REMAP-COPY-PROBE %s done\n", mode.c_str());
    return 0;
}
