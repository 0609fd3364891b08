// Does a kernel that uses scratch (private segment) compute correctly on this pool with many waves in flight?
// Each thread keeps a 2-word-key array in a dynamically indexed private array (forces scratch), permutes it and
// writes a checksum; the host recomputes the same on the CPU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
struct K2 { uint64_t w[2]; };
__host__ __device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; return x ^ (x >> 33); }
__host__ __device__ inline uint64_t work(uint64_t tid, int rounds) {
    K2 a[24];
    for (int i = 0; i < 24; ++i) { a[i].w[0] = mix(tid * 31 + i); a[i].w[1] = mix(tid * 17 + i * 3); }
    uint64_t acc = 0;
    for (int r = 0; r < rounds; ++r) {
        int j = (int)((mix(tid + r) >> 7) % 24), q = (int)((mix(tid * 3 + r) >> 9) % 24);
        K2 t = (a[j].w[0] < a[q].w[0]) ? a[j] : a[q];   // struct-level select + dynamic index
        a[j] = a[q]; a[q] = t;
        acc += a[(j + q) % 24].w[1] ^ t.w[0];
    }
    return acc;
}
__global__ void k(uint64_t *out, uint64_t n, int rounds) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = work(i, rounds);
}
int main() {
    const uint64_t n = 1ull << 22; const int rounds = 64;
    uint64_t *d; hipMalloc(&d, n * 8);
    hipFuncAttributes at; hipFuncGetAttributes(&at, (const void *)k);
    printf("kernel localSizeBytes (scratch) = %zu, numRegs = %d\n", at.localSizeBytes, at.numRegs);
    int bad_total = 0;
    for (int rep = 0; rep < 5; ++rep) {
        hipLaunchKernelGGL(k, dim3((unsigned)(n / 256)), dim3(256), 0, 0, d, n, rounds);
        std::vector<uint64_t> h(n);
        hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
        int bad = 0;
        for (uint64_t i = 0; i < n; i += 97) if (h[i] != work(i, rounds)) ++bad;
        printf("rep %d: mismatches %d of %llu sampled\n", rep, bad, (unsigned long long)(n / 97));
        bad_total += bad;
    }
    printf(bad_total ? "SCRATCH-PROBE: WRONG RESULTS\n" : "SCRATCH-PROBE: all correct\n");
    return 0;
}
