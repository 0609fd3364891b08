// probe: does a kernel with >64 KiB of LDS (static + dynamic) get a private allocation per block?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int *out, int nwords) {
    extern __shared__ int sm[];
    __shared__ int stat[1541];
    for (int i = threadIdx.x; i < nwords; i += blockDim.x) sm[i] = blockIdx.x;
    stat[threadIdx.x] = blockIdx.x;
    __syncthreads();
    // spin a little so blocks overlap in time
    long long t0 = clock64();
    while (clock64() - t0 < 200000) {}
    __syncthreads();
    int bad = 0;
    for (int i = threadIdx.x; i < nwords; i += blockDim.x) bad += sm[i] != (int)blockIdx.x;
    bad += stat[threadIdx.x] != (int)blockIdx.x;
    if (bad) atomicAdd(out, bad);
}
int main() {
    int *d; hipMalloc(&d, 4);
    for (int kb : {32, 64, 70, 100, 150}) {
        for (int setattr = 0; setattr < 2; ++setattr) {
            hipMemset(d, 0, 4);
            size_t dyn = (size_t)kb * 1024;
            hipError_t e1 = hipSuccess;
            if (setattr) e1 = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
            hipLaunchKernelGGL(k, dim3(4096), dim3(256), dyn, 0, d, (int)(dyn / 4));
            hipError_t e2 = hipGetLastError();
            hipError_t e3 = hipDeviceSynchronize();
            int h = -1; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
            printf("dyn=%dKB setattr=%d: attr=%s launch=%s sync=%s bad=%d\n", kb, setattr, hipGetErrorString(e1), hipGetErrorString(e2), hipGetErrorString(e3), h);
        }
    }
    return 0;
}
