// Which SIMD does wave w of a 512-thread workgroup run on?  (HW_REG_HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13)
// build: hipcc --offload-arch=gfx950 -O2 -o wave_simd_probe tools/probe/wave_simd_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(512) void probe(unsigned* out) {
    extern __shared__ unsigned char lds[];
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
    lds[threadIdx.x] = 1;
}

int main() {
    const int blocks = 6;
    unsigned* d;
    hipMalloc(&d, blocks * 8 * 4);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 98304, 0, d);
    unsigned h[blocks * 8];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < blocks; ++b) {
        printf("block %d:", b);
        for (int w = 0; w < 8; ++w) printf("  w%d simd%u slot%u cu%u", w, (h[b * 8 + w] >> 4) & 3, h[b * 8 + w] & 15, (h[b * 8 + w] >> 8) & 15);
        printf("\n");
    }
    return 0;
}
