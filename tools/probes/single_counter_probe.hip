// One returning atomicAdd per workgroup on ONE address (how the dedup kernels reserve their place in the dense output):
// what does the serialisation of same-address atomics cost at 227 210 workgroups (the buckets of BASELINE configs[1])?
// Variants: 1 counter, 8 (one per XCD), 64, 1024 (one per level-1 segment), each on its own 128-byte line.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(512) void k(uint32_t *cur, uint32_t ncounters, int by_xcd, uint32_t *out, int work) {
    __shared__ uint32_t base;
    uint32_t xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const uint32_t c = by_xcd ? (xcc & 7u) : blockIdx.x % ncounters;
    // some work before the reservation, as in the kernel (table insert): a dependent LDS/ALU chain
    uint32_t v = threadIdx.x;
    for (int i = 0; i < work; ++i) v = v * 0x9E3779B1u + (v >> 15);
    if (threadIdx.x == 0) base = atomicAdd(&cur[c * 32u], 600u + (v & 1u));
    __syncthreads();
    if (base == 0xFFFFFFFFu) out[threadIdx.x] = v;
}

int main() {
    uint32_t *cur, *out;
    CK(hipMalloc(&cur, 1024 * 128));
    CK(hipMalloc(&out, 4096));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const uint32_t wgs = 227210;
    struct Case { uint32_t n; int by_xcd; int work; const char *what; };
    const Case cases[] = {
        {1, 0, 0, "1 counter"}, {1, 1, 0, "8 counters (per XCD)"}, {64, 0, 0, "64 counters"}, {1024, 0, 0, "1024 counters"},
        {1, 0, 2000, "1 counter, 2000 dependent multiplies before"}, {1, 1, 2000, "per XCD, 2000 multiplies before"},
        {1024, 0, 2000, "1024 counters, 2000 multiplies before"},
    };
    for (const Case &c : cases) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(cur, 0, 1024 * 128));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(k, dim3(wgs), dim3(512), 0, 0, cur, c.n, c.by_xcd, out, c.work);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            best = ms < best ? ms : best;
        }
        printf("%-48s %7.3f ms\n", c.what, best);
    }
    return 0;
}
