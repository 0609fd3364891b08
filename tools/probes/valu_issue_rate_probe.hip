// issue rate of a few gfx950 vector instructions: 4 independent dependency chains per lane, 8 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHAIN4(INSN)                                                            \
    asm volatile(INSN " %0, %0, %4\n\t" INSN " %1, %1, %4\n\t" INSN " %2, %2, %4\n\t" INSN " %3, %3, %4" \
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m))
template <int OP>
__global__ void k(uint32_t *out, uint32_t seed, int iters) {
    uint32_t a = threadIdx.x + seed, b = a * 3 + 1, c = a ^ 0x55u, d = b + 7, m = seed * 0x9E3779B1u | 1u;
    unsigned long long p = ((unsigned long long)a << 32) | b, q = ((unsigned long long)c << 32) | d;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (OP == 0) CHAIN4("v_mul_lo_u32");
            if (OP == 1) CHAIN4("v_mul_u32_u24");
            if (OP == 2) CHAIN4("v_add_u32");
            if (OP == 3) CHAIN4("v_mul_hi_u32");
            if (OP == 4) CHAIN4("v_xor_b32");
            if (OP == 5) asm volatile("v_lshlrev_b64 %0, %2, %0\n\tv_lshrrev_b64 %1, %2, %1\n\tv_lshlrev_b64 %0, %2, %0\n\tv_lshrrev_b64 %1, %2, %1" : "+v"(p), "+v"(q) : "v"(m));
            if (OP == 6) asm volatile("v_alignbit_b32 %0, %0, %1, %4\n\tv_alignbit_b32 %1, %1, %2, %4\n\tv_alignbit_b32 %2, %2, %3, %4\n\tv_alignbit_b32 %3, %3, %0, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (OP == 7) asm volatile("v_bfrev_b32 %0, %0\n\tv_bfrev_b32 %1, %1\n\tv_bfrev_b32 %2, %2\n\tv_bfrev_b32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ (uint32_t)p ^ (uint32_t)q;
}
int main() {
    uint32_t *d; (void)hipMalloc(&d, 512 * 2048 * 4);
    const char *names[8] = {"v_mul_lo_u32", "v_mul_u32_u24", "v_add_u32", "v_mul_hi_u32", "v_xor_b32", "v_lsh*_b64 (variable)", "v_alignbit_b32", "v_bfrev_b32"};
    for (int op = 0; op < 8; ++op) {
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(a);
            switch (op) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(2048), dim3(512), 0, 0, d, 3u, 1024); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(2048), dim3(512), 0, 0, d, 3u, 1024); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(2048), dim3(512), 0, 0, d, 3u, 1024); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(2048), dim3(512), 0, 0, d, 3u, 1024); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(2048), dim3(512), 0, 0, d, 3u, 1024); break;
                case 5: hipLaunchKernelGGL(k<5>, dim3(2048), dim3(512), 0, 0, d, 3u, 1024); break;
                case 6: hipLaunchKernelGGL(k<6>, dim3(2048), dim3(512), 0, 0, d, 3u, 1024); break;
                case 7: hipLaunchKernelGGL(k<7>, dim3(2048), dim3(512), 0, 0, d, 3u, 1024); break;
            }
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            (void)hipEventElapsedTime(&ms, a, b);
        }
        const double insns = 2048.0 * 512 * 1024 * 64;  // lane-instructions
        printf("%-24s %8.3f ms  %6.2f T lane-instructions/s\n", names[op], ms, insns / ms / 1e9);
    }
    return 0;
}
