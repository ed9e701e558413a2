// LDS behaviour of 16-byte accesses on gfx950 (the double2 exchanges of zafx_f64.hip): which paddings are conflict free.
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/exp_lds128 tools/exp_lds128.hip && tools/bin/exp_lds128
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
    // 16-byte elements (double2 of the float64 transforms, zafx_f64.hip): reads and writes of one wavefront under the padding i + (i >> 4)
    const int waves = 8;
    auto pat = [&](const char* name, int (*f)(int)) {
        std::vector<int> li(64);
        for (int l = 0; l < 64; ++l) li[l] = f(l);
        std::vector<int> l2(64);
        for (int l = 0; l < 64; ++l) l2[l] = 2 * f(l);   // the same addresses in 8-byte units
        printf("%-44s b128 read %.4f write %.4f | as b64 (one half) read %.4f write %.4f\n", name, run<float4>(li, waves), runw<float4>(li, waves), run<float2>(l2, waves), runw<float2>(l2, waves));
    };
    pat("consecutive (lane)", [](int l) { return l; });
    pat("physd(lane)           = lane + lane/16", [](int l) { return l + (l >> 4); });
    pat("physd(16 lane)        = 17 lane", [](int l) { return 17 * l; });
    pat("physd(256 (l/16) + l%16) = 272 (l/16) + l%16", [](int l) { return 272 * (l >> 4) + (l & 15); });
    pat("16 lane (no padding)", [](int l) { return 16 * l; });
    pat("17 lane + lane/4", [](int l) { return 17 * l + (l >> 2); });
    pat("9 lane", [](int l) { return 9 * l; });
    pat("33 lane", [](int l) { return 33 * l; });
    pat("lane + 2 (lane/16)", [](int l) { return l + 2 * (l >> 4); });
    pat("lane + 4 (lane/16)", [](int l) { return l + 4 * (l >> 4); });
    for (int B : {1, 2, 4, 7, 8, 15, 16, 24, 31, 32, 40, 48, 63}) {   // which lanes of a b128 access are served together: lane B on lane 0's banks
        std::vector<int> r(64);
        for (int l = 0; l < 64; ++l) r[l] = l;
        r[B] = 64;
        printf("lane %2d on lane 0's banks (another address): b128 read %.4f write %.4f\n", B, run<float4>(r, waves), runw<float4>(r, waves));
    }
    return 0;
}
