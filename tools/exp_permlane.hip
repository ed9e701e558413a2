// Issue cost of v_permlane16_swap / v_permlane32_swap / DPP moves / v_pk_fma_f32 on gfx950 (cycles per instruction, one wave per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(long long* out, float* sink, int iters) {
    float a = threadIdx.x, b = a + 1.f, c = a + 2.f, d = a + 3.f;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MODE == 0) asm volatile("v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (MODE == 1) asm volatile("v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (MODE == 2) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (MODE == 3) asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %2, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (MODE == 4) asm volatile("v_sqrt_f32 %0, %1\n\tv_sqrt_f32 %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        }
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (a + b + c + d == 123.456f) sink[0] = a;
}
template <int MODE>
double run(int waves_per_simd) {
    long long* d_out; float* d_sink;
    hipMalloc(&d_out, 8); hipMalloc(&d_sink, 4);
    const int iters = 4000;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(256 * waves_per_simd), 0, 0, d_out, d_sink, iters);
    long long c; hipMemcpy(&c, d_out, 8, hipMemcpyDeviceToHost);
    hipFree(d_out); hipFree(d_sink);
    return (double)c / (iters * 32.0 * waves_per_simd);
}
int main() {
    for (int w : {1, 2, 4}) {
        printf("%d wave(s) per SIMD, clock64 ticks per instruction and wave: permlane16_swap %.2f  permlane32_swap %.2f  dpp mov %.2f  v_add_f32 %.2f  v_sqrt_f32 %.2f\n", w,
               run<0>(w), run<1>(w), run<2>(w), run<3>(w), run<4>(w));
    }
    return 0;
}
