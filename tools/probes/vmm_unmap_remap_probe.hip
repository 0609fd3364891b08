// unmap chunks at the end of a reserved range, map NEW physical chunks at the same addresses, use them: safe?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void fill(unsigned long long *p, size_t n, unsigned long long tag) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = i ^ tag;
}
__global__ void check(const unsigned long long *p, size_t n, unsigned long long tag, unsigned long long *bad) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (p[i] != (i ^ tag)) atomicAdd(bad, 1ull);
}
int main(int argc, char **argv) {
    const bool same_stream_sync = argc > 1;
    CK(hipSetDevice(0));
    hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    const size_t chunk = 1ull << 30, N = 16;
    char *base; CK(hipMemAddressReserve((void **)&base, 1ull << 40, 0, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> h(N);
    auto map = [&](size_t i) -> int { CK(hipMemCreate(&h[i], chunk, &prop, 0)); CK(hipMemMap(base + i * chunk, chunk, 0, h[i], 0)); CK(hipMemSetAccess(base + i * chunk, chunk, &acc, 1)); return 0; };
    for (size_t i = 0; i < N; ++i) if (map(i)) return 1;
    unsigned long long *bad; CK(hipMalloc(&bad, 8)); CK(hipMemset(bad, 0, 8));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    size_t n = N * chunk / 8;
    for (int round = 0; round < 6; ++round) {
        hipLaunchKernelGGL(fill, dim3(65536), dim3(256), 0, st, (unsigned long long *)base, n, 0x1000ull + round);
        hipLaunchKernelGGL(check, dim3(65536), dim3(256), 0, st, (const unsigned long long *)base, n, 0x1000ull + round, bad);
        CK(hipStreamSynchronize(st));
        // release the upper half, map new physical memory there
        for (size_t i = N / 2; i < N; ++i) { CK(hipMemUnmap(base + i * chunk, chunk)); CK(hipMemRelease(h[i])); }
        // something else grabs memory in between (like torch would)
        void *other; CK(hipMalloc(&other, 3 * chunk)); CK(hipMemset(other, 0xAB, 3 * chunk));
        for (size_t i = N / 2; i < N; ++i) if (map(i)) return 1;
        CK(hipFree(other));
        // runtime copy / fill paths on the re-mapped half (the engine uses them all the time)
        CK(hipMemsetAsync(base + (N / 2) * chunk, 0x5A, chunk, st));
        CK(hipMemcpyAsync(base + (N / 2 + 1) * chunk, base, chunk, hipMemcpyDeviceToDevice, st));
        std::vector<unsigned long long> hostbuf(1 << 20);
        CK(hipMemcpyAsync(hostbuf.data(), base + (N / 2) * chunk + 4096, hostbuf.size() * 8, hipMemcpyDeviceToHost, st));
        CK(hipMemcpyAsync(base + (N - 1) * chunk, hostbuf.data(), hostbuf.size() * 8, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        if (hostbuf[5] != 0x5A5A5A5A5A5A5A5Aull) { printf("memset on re-mapped memory not visible: %llx\n", hostbuf[5]); return 2; }
        unsigned long long hb; CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
        printf("round %d ok so far, mismatches %llu\n", round, hb); fflush(stdout);
    }
    hipLaunchKernelGGL(fill, dim3(65536), dim3(256), 0, st, (unsigned long long *)base, n, 7ull);
    hipLaunchKernelGGL(check, dim3(65536), dim3(256), 0, st, (const unsigned long long *)base, n, 7ull, bad);
    CK(hipStreamSynchronize(st));
    unsigned long long hb; CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
    printf("VMM-REMAP-PROBE: %s (mismatches %llu)\n", hb ? "CORRUPTION" : "clean", hb);
    return 0;
}
