// exp_hwid.hip -- round 5: which SIMD of its CU does wave w of a 1024-thread workgroup run on?  (s_getreg_b32 HW_ID: SIMD_ID = bits 5:4,
// WAVE_ID = bits 3:0, CU_ID = bits 11:8)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(1024) void k(unsigned* out) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
}
int main() {
    unsigned* d;
    hipMalloc(&d, 4 * 16 * 4);
    for (int nt : {1024, 512, 256}) {
        hipMemset(d, 0xff, 4 * 16 * 4);
        hipLaunchKernelGGL(k, dim3(4), dim3(nt), 0, 0, d);
        unsigned h[64];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        for (int b = 0; b < 2; ++b) {
            std::printf("threads %4d block %d: wave -> simd.slot (cu):", nt, b);
            for (int w = 0; w < nt / 64; ++w) std::printf(" %d:%u.%u(%u)", w, (h[b * 16 + w] >> 4) & 3, h[b * 16 + w] & 15, (h[b * 16 + w] >> 8) & 15);
            std::printf("\n");
        }
    }
    return 0;
}
