// LDS bank behaviour on gfx950: cycles per ds_read_b32 / b64 / b128 of one wavefront as a function of the lane stride.
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/exp_ldsbank tools/exp_ldsbank.hip && tools/bin/exp_ldsbank
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <class T>
__global__ void k(const int* __restrict__ idx, long long* out, float* sink, int iters) {
    extern __shared__ unsigned char smem[];
    T* buf = reinterpret_cast<T*>(smem);
    for (int i = threadIdx.x; i < 65536 / (int)sizeof(T); i += blockDim.x) buf[i] = T{};
    __syncthreads();
    const int my = idx[threadIdx.x];
    float acc = 0.f;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        // 8 independent reads per iteration at constant offsets (same bank pattern)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            T v = buf[my + j * (4096 / (int)sizeof(T))];
            acc += reinterpret_cast<float*>(&v)[0];
        }
        asm volatile("" : "+v"(acc));
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc == 123.456f) sink[0] = acc;
}

template <class T>
__global__ void kw(const int* __restrict__ idx, long long* out, int iters) {
    extern __shared__ unsigned char smem[];
    T* buf = reinterpret_cast<T*>(smem);
    const int my = idx[threadIdx.x];
    T v{};
    reinterpret_cast<float*>(&v)[0] = (float)threadIdx.x;
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) buf[my + j * (4096 / (int)sizeof(T))] = v;
        asm volatile("" ::: "memory");
    }
    __syncthreads();
    const long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <class T>
double runw(const std::vector<int>& lane_index, int waves) {
    int* d_idx; long long* d_out;
    std::vector<int> idx(64 * waves);
    for (int w = 0; w < waves; ++w) for (int l = 0; l < 64; ++l) idx[w * 64 + l] = lane_index[l];
    hipMalloc(&d_idx, idx.size() * 4); hipMalloc(&d_out, 8);
    hipMemcpy(d_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice);
    const int iters = 2000;
    hipLaunchKernelGGL(kw<T>, dim3(1), dim3(64 * waves), 65536, 0, d_idx, d_out, iters);
    hipLaunchKernelGGL(kw<T>, dim3(1), dim3(64 * waves), 65536, 0, d_idx, d_out, iters);
    long long c; hipMemcpy(&c, d_out, 8, hipMemcpyDeviceToHost);
    hipFree(d_idx); hipFree(d_out);
    return (double)c / (iters * 8.0 * waves);
}

template <class T>
double run(const std::vector<int>& lane_index, int waves) {
    int* d_idx; long long* d_out; float* d_sink;
    std::vector<int> idx(64 * waves);
    for (int w = 0; w < waves; ++w) for (int l = 0; l < 64; ++l) idx[w * 64 + l] = lane_index[l];
    hipMalloc(&d_idx, idx.size() * 4); hipMalloc(&d_out, 8); hipMalloc(&d_sink, 4);
    hipMemcpy(d_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice);
    const int iters = 2000;
    hipLaunchKernelGGL(k<T>, dim3(1), dim3(64 * waves), 65536, 0, d_idx, d_out, d_sink, iters);
    hipLaunchKernelGGL(k<T>, dim3(1), dim3(64 * waves), 65536, 0, d_idx, d_out, d_sink, iters);
    long long c; hipMemcpy(&c, d_out, 8, hipMemcpyDeviceToHost);
    hipFree(d_idx); hipFree(d_out); hipFree(d_sink);
    return (double)c / (iters * 8.0 * waves);   // clock64 ticks per wave-level read (ticks at 100 MHz: relative numbers only)
}

int main() {
    const int waves = 8;
    printf("ticks per wave-level read (relative), %d waves\n", waves);
    for (int stride : {1, 2, 3, 4, 5, 8, 9, 16, 17, 32, 33, 34}) {
        std::vector<int> li(64);
        for (int l = 0; l < 64; ++l) li[l] = l * stride;
        printf("lane stride %2d elements:  b32 %.4f   b64 %.4f   b128 %.4f\n", stride, run<float>(li, waves), run<float2>(li, waves), run<float4>(li, waves));
    }
    // row-pair order on the usual frame layout (slot 17 p) vs the padded one (16 p + p / 2), 8-byte elements
    std::vector<int> a(64), b(64), c(64);
    for (int l = 0; l < 64; ++l) {
        const int p = 2 * (l & 15) + ((l >> 4) & 1) + (l & 32);
        a[l] = 17 * p; b[l] = 16 * p + (p >> 1); c[l] = 17 * l;
    }
    printf("b64 READ  slot 17 p (row-pair p) %.4f | slot 16 p + p/2 (row-pair p) %.4f | slot 17 lane %.4f\n", run<float2>(a, waves), run<float2>(b, waves), run<float2>(c, waves));
    printf("b64 WRITE slot 17 p (row-pair p) %.4f | slot 16 p + p/2 (row-pair p) %.4f | slot 17 lane %.4f\n", runw<float2>(a, waves), runw<float2>(b, waves), runw<float2>(c, waves));
    for (int stride : {1, 2, 4}) {
        std::vector<int> li(64);
        for (int l = 0; l < 64; ++l) li[l] = l * stride;
        printf("WRITE lane stride %d: b32 %.4f  b64 %.4f  b128 %.4f\n", stride, runw<float>(li, waves), runw<float2>(li, waves), runw<float4>(li, waves));
    }
    // which lanes are served together: all lanes on their own banks except lane B, which sits on lane 0's bank (another address)
    for (int B : {1, 4, 8, 15, 16, 24, 31, 32, 40, 48, 63}) {
        std::vector<int> r64(64), r32(64);
        for (int l = 0; l < 64; ++l) { r64[l] = l; r32[l] = l; }
        r64[B] = 64;    // 8-byte slots: 64 = bank pair 0 again
        r32[B] = 64;    // 4-byte slots: 64 = bank 0 again
        printf("lane %2d on lane 0's bank: b64 read %.4f write %.4f | b32 read %.4f write %.4f\n", B, run<float2>(r64, waves), runw<float2>(r64, waves),
               run<float>(r32, waves), runw<float>(r32, waves));
    }
    return 0;
}
