// exp_lds2.hip -- round 5: what does the compiler's merging of two 8-byte LDS accesses into ds_read2_b64 / ds_write2_b64 cost?
// 16 waves of one workgroup, each `iters` x 16 accesses of 8 bytes per lane (conflict free: lane-contiguous), issued as
//   0: 16 ds_read_b64      1: 8 ds_read2_b64 (offsets 64 apart)      2: 8 ds_read2st64_b64      3: 8 ds_read_b128 (16 bytes per lane)
//   4: 16 ds_write_b64     5: 8 ds_write2_b64                        6: 8 ds_write_b128
// Cycles of the whole workgroup per 8 bytes x 64 lanes moved.
#include <hip/hip_runtime.h>
#include <cstdio>
#define R4(x) x x x x
#define R8(x) R4(x) R4(x)
#define R16(x) R8(x) R8(x)
template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, float* sink, int iters) {
    extern __shared__ float2 lds[];
    for (int i = threadIdx.x; i < 16384; i += 1024) lds[i] = make_float2(i, 1.f);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int a8 = (wave * 1024 + lane) * 8, a16 = (wave * 1024 + lane * 2) * 8;   // byte addresses: a wave's own 8 KB
    typedef float v2 __attribute__((ext_vector_type(2)));
    typedef float v4 __attribute__((ext_vector_type(4)));
    v2 r0 = {0, 0}; v4 q0 = {0, 0, 0, 0}, q1 = q0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) asm volatile("ds_read_b64 %0, %1\n\tds_read_b64 %0, %1 offset:512\n\tds_read_b64 %0, %1 offset:1024\n\tds_read_b64 %0, %1 offset:1536\n\t"
                                    "ds_read_b64 %0, %1 offset:2048\n\tds_read_b64 %0, %1 offset:2560\n\tds_read_b64 %0, %1 offset:3072\n\tds_read_b64 %0, %1 offset:3584\n\t"
                                    "ds_read_b64 %0, %1 offset:4096\n\tds_read_b64 %0, %1 offset:4608\n\tds_read_b64 %0, %1 offset:5120\n\tds_read_b64 %0, %1 offset:5632\n\t"
                                    "ds_read_b64 %0, %1 offset:6144\n\tds_read_b64 %0, %1 offset:6656\n\tds_read_b64 %0, %1 offset:7168\n\tds_read_b64 %0, %1 offset:7680\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r0) : "v"(a8) : "memory");
        if (MODE == 1) asm volatile("ds_read2_b64 %0, %1 offset0:0 offset1:64\n\tds_read2_b64 %0, %1 offset0:128 offset1:192\n\tds_read2_b64 %0, %1 offset0:1 offset1:65\n\tds_read2_b64 %0, %1 offset0:129 offset1:193\n\t"
                                    "ds_read2_b64 %0, %1 offset0:2 offset1:66\n\tds_read2_b64 %0, %1 offset0:130 offset1:194\n\tds_read2_b64 %0, %1 offset0:3 offset1:67\n\tds_read2_b64 %0, %1 offset0:131 offset1:195\n\ts_waitcnt lgkmcnt(0)" : "=&v"(q0) : "v"(a8) : "memory");
        if (MODE == 2) asm volatile(R8("ds_read2st64_b64 %0, %1 offset0:0 offset1:1\n\t") "s_waitcnt lgkmcnt(0)" : "=&v"(q0) : "v"(a8) : "memory");
        if (MODE == 3) asm volatile("ds_read_b128 %0, %1\n\tds_read_b128 %0, %1 offset:1024\n\tds_read_b128 %0, %1 offset:2048\n\tds_read_b128 %0, %1 offset:3072\n\t"
                                    "ds_read_b128 %0, %1 offset:4096\n\tds_read_b128 %0, %1 offset:5120\n\tds_read_b128 %0, %1 offset:6144\n\tds_read_b128 %0, %1 offset:7168\n\ts_waitcnt lgkmcnt(0)" : "=&v"(q0) : "v"(a16) : "memory");
        if (MODE == 4) asm volatile("ds_write_b64 %1, %0\n\tds_write_b64 %1, %0 offset:512\n\tds_write_b64 %1, %0 offset:1024\n\tds_write_b64 %1, %0 offset:1536\n\t"
                                    "ds_write_b64 %1, %0 offset:2048\n\tds_write_b64 %1, %0 offset:2560\n\tds_write_b64 %1, %0 offset:3072\n\tds_write_b64 %1, %0 offset:3584\n\t"
                                    "ds_write_b64 %1, %0 offset:4096\n\tds_write_b64 %1, %0 offset:4608\n\tds_write_b64 %1, %0 offset:5120\n\tds_write_b64 %1, %0 offset:5632\n\t"
                                    "ds_write_b64 %1, %0 offset:6144\n\tds_write_b64 %1, %0 offset:6656\n\tds_write_b64 %1, %0 offset:7168\n\tds_write_b64 %1, %0 offset:7680\n\ts_waitcnt lgkmcnt(0)" :: "v"(r0), "v"(a8) : "memory");
        if (MODE == 5) asm volatile("ds_write2_b64 %2, %0, %1 offset0:0 offset1:64\n\tds_write2_b64 %2, %0, %1 offset0:128 offset1:192\n\tds_write2_b64 %2, %0, %1 offset0:1 offset1:65\n\tds_write2_b64 %2, %0, %1 offset0:129 offset1:193\n\t"
                                    "ds_write2_b64 %2, %0, %1 offset0:2 offset1:66\n\tds_write2_b64 %2, %0, %1 offset0:130 offset1:194\n\tds_write2_b64 %2, %0, %1 offset0:3 offset1:67\n\tds_write2_b64 %2, %0, %1 offset0:131 offset1:195\n\ts_waitcnt lgkmcnt(0)" :: "v"(r0), "v"(r0), "v"(a8) : "memory");
        if (MODE == 6) asm volatile("ds_write_b128 %1, %0\n\tds_write_b128 %1, %0 offset:1024\n\tds_write_b128 %1, %0 offset:2048\n\tds_write_b128 %1, %0 offset:3072\n\t"
                                    "ds_write_b128 %1, %0 offset:4096\n\tds_write_b128 %1, %0 offset:5120\n\tds_write_b128 %1, %0 offset:6144\n\tds_write_b128 %1, %0 offset:7168\n\ts_waitcnt lgkmcnt(0)" :: "v"(q1), "v"(a16) : "memory");
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    sink[threadIdx.x] = r0.x + q0.x + q0.w;
}
template <int MODE>
void run(const char* what) {
    unsigned long long* d; float* s;
    hipMalloc(&d, 8); hipMalloc(&s, 4096);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    const int iters = 2000;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(1024), 131072, 0, d, s, iters);
    unsigned long long c; hipMemcpy(&c, d, 8, hipMemcpyDeviceToHost);
    std::printf("%-40s %6.2f cycles per 512 bytes (one 8-byte access of a wave), 16 waves -> %5.1f B/clk/CU\n", what, (double)c / (iters * 16.0 * 16.0), 512.0 * iters * 256.0 / c);
    hipFree(d); hipFree(s);
}
int main() {
    run<0>("16 x ds_read_b64");
    run<1>("8 x ds_read2_b64");
    run<2>("8 x ds_read2st64_b64");
    run<3>("8 x ds_read_b128");
    run<4>("16 x ds_write_b64");
    run<5>("8 x ds_write2_b64");
    run<6>("8 x ds_write_b128");
    return 0;
}
