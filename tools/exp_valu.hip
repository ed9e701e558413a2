// exp_valu.hip -- round 5: what a gfx950 SIMD does with the packed-f32 instructions the FFT core is made of.
// One workgroup on one CU, w waves on ONE SIMD (w = 1, 2, 4: waves 0, 4, 8, 12 of a 16-wave workgroup share SIMD 0; the other
// waves exit at once), each running a chain of N instructions, dependent or independent; cycles per instruction and wave from
// s_memtime around the chain.
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/exp_valu tools/exp_valu.hip && tools/bin/exp_valu
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

// MODE 0: dependent v_pk_fma_f32 chain; 1: four independent chains interleaved; 2: cmul pairs (mul -> fma) back to back, each pair
// feeding the next; 3: four cmul pairs interleaved; 4: dependent v_pk_add_f32; 5: v_sqrt_f32 dependent; 6: v_sqrt_f32 independent x4;
// 7: v_permlane32_swap pairs; 8: ds_read_b64 dependent address chain (LDS latency); 9: v_mov_b32 dependent; 10: plain v_fma_f32 dependent
template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, float* sink, int waves_on_simd) {
    __shared__ float2 lds[1024];
    const int wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 1024; i += 1024) lds[i] = make_float2(0.f, 0.f);
    __syncthreads();
    if ((wave & 3) != 0 || (wave >> 2) >= waves_on_simd) return;
    v2f a = {1.0f + threadIdx.x * 1e-9f, 0.5f}, b = {0.999f, 1e-3f}, c = {0.f, 0.f}, d = {1.f, 1.f}, e = {2.f, 2.f}, f = {3.f, 3.f};
    v2f g = c, h = c, p = c, q = c;
    float s0 = 1.5f + threadIdx.x, s1 = 2.5f, s2 = 3.5f, s3 = 4.5f;
    int addr = (threadIdx.x & 63) * 8;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 16; ++it) {
        if constexpr (MODE == 0) {
            asm volatile(REP64("v_pk_fma_f32 %0, %0, %1, %2\n\t") : "+v"(a) : "v"(b), "v"(c));
        } else if constexpr (MODE == 1) {
            asm volatile(REP16("v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5\n\t")
                         : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c));
        } else if constexpr (MODE == 2) {
            asm volatile(REP16(REP4("v_pk_mul_f32 %1, %0, %2 op_sel_hi:[0,1]\n\tv_pk_fma_f32 %0, %0, %2, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t") "s_nop 0\n\t")
                         : "+v"(a), "+v"(d) : "v"(b));
        } else if constexpr (MODE == 3) {
            asm volatile(REP16("v_pk_mul_f32 %4, %0, %8 op_sel_hi:[0,1]\n\tv_pk_mul_f32 %5, %1, %8 op_sel_hi:[0,1]\n\tv_pk_mul_f32 %6, %2, %8 op_sel_hi:[0,1]\n\tv_pk_mul_f32 %7, %3, %8 op_sel_hi:[0,1]\n\t"
                               "v_pk_fma_f32 %0, %0, %8, %4 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\tv_pk_fma_f32 %1, %1, %8, %5 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
                               "v_pk_fma_f32 %2, %2, %8, %6 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\tv_pk_fma_f32 %3, %3, %8, %7 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t")
                         : "+v"(a), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(p), "+v"(q) : "v"(b));
        } else if constexpr (MODE == 4) {
            asm volatile(REP64("v_pk_add_f32 %0, %0, %1\n\t") : "+v"(a) : "v"(c));
        } else if constexpr (MODE == 5) {
            asm volatile(REP64("v_sqrt_f32 %0, %0\n\t") : "+v"(s0));
        } else if constexpr (MODE == 6) {
            asm volatile(REP16("v_sqrt_f32 %0, %0\n\tv_sqrt_f32 %1, %1\n\tv_sqrt_f32 %2, %2\n\tv_sqrt_f32 %3, %3\n\t") : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));
        } else if constexpr (MODE == 7) {
            asm volatile(REP16("v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane16_swap_b32 %0, %2\n\tv_permlane16_swap_b32 %1, %3\n\t") : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));
        } else if constexpr (MODE == 8) {
            asm volatile(REP64("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_add_u32 %1, %1, %0\n\t") : "+v"(s1), "+v"(addr)::"memory");
        } else if constexpr (MODE == 9) {
            asm volatile(REP64("v_mov_b32 %0, %0\n\t") : "+v"(s0));
        } else if constexpr (MODE == 10) {
            asm volatile(REP64("v_fma_f32 %0, %0, %1, %2\n\t") : "+v"(s0) : "v"(s1), "v"(s2));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) {
        out[2 * (wave >> 2)] = t0;
        out[2 * (wave >> 2) + 1] = t1;
    }
    sink[threadIdx.x] = a.x + a.y + d.x + e.x + f.x + c.x + g.x + h.x + p.x + q.x + s0 + s1 + s2 + s3 + addr;
}

template <int MODE>
void run(const char* what, int per_iter) {
    unsigned long long* d_out;
    float* d_sink;
    hipMalloc(&d_out, 64);
    hipMalloc(&d_sink, 4096);
    for (int w : {1, 2, 4}) {
        hipMemset(d_out, 0, 64);
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(1024), 0, 0, d_out, d_sink, w);
        hipDeviceSynchronize();
        unsigned long long h[8];
        hipMemcpy(h, d_out, 64, hipMemcpyDeviceToHost);
        const double n = 16.0 * per_iter;
        unsigned long long first = ~0ULL, last = 0;
        for (int i = 0; i < w; ++i) first = h[2 * i] < first ? h[2 * i] : first, last = h[2 * i + 1] > last ? h[2 * i + 1] : last;
        std::printf("%-52s waves/SIMD %d: oldest wave %6.2f, youngest %6.2f cycles per instruction; SIMD: %5.2f cycles per instruction\n", what, w,
                    (h[1] - h[0]) / n, (h[2 * w - 1] - h[2 * w - 2]) / n, (last - first) / n / w);
    }
    hipFree(d_out);
    hipFree(d_sink);
}

int main() {
    run<0>("v_pk_fma_f32 dependent", 64);
    run<1>("v_pk_fma_f32 four independent chains", 64);
    run<2>("cmul (pk_mul -> pk_fma) pairs back to back", 128);
    run<3>("cmul pairs, four interleaved", 128);
    run<4>("v_pk_add_f32 dependent", 64);
    run<10>("v_fma_f32 dependent", 64);
    run<9>("v_mov_b32 dependent", 64);
    run<5>("v_sqrt_f32 dependent", 64);
    run<6>("v_sqrt_f32 four independent", 64);
    run<7>("v_permlane32/16_swap", 64);
    run<8>("ds_read_b64 -> wait -> address (LDS round trip)", 64);
    return 0;
}
