// Microbenchmark: what does the memory system give for the ISTFT input pattern?
// spec[clip][row][t] complex64, rows = 2048, T = 432; a workgroup reads a [2048 rows][RUN frames] tile.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(1024) void k_linear(const float4* __restrict__ in, long long n4, float* sink) {
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * 1024 + threadIdx.x; i < n4; i += (long long)gridDim.x * 1024) {
        const float4 v = in[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) sink[0] = acc;
}
// one workgroup per (clip, tile); lanes: tt = tid % RUN (frames), kq = tid / RUN; loop over rows
template <int RUN, int NT, int UNROLL>
__global__ __launch_bounds__(NT) void k_tile(const float2* __restrict__ in, int T, int tiles, int rows, float* sink) {
    const int clip = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const int tt = threadIdx.x % RUN, kq = threadIdx.x / RUN;
    const int t = tile * RUN + tt;
    if (t >= T) return;
    const float2* p = in + (long long)clip * rows * T + t;
    float acc = 0.f;
#pragma unroll UNROLL
    for (int k = kq; k < rows; k += NT / RUN) {
        const float2 v = p[(long long)k * T];
        acc += v.x + v.y;
    }
    if (acc == 12345.678f) sink[0] = acc;
}
// persistent: a workgroup walks the tiles of whole clips in order (the carry kernel's traversal)
template <int RUN, int NT, int UNROLL>
__global__ __launch_bounds__(NT) void k_walk(const float2* __restrict__ in, int T, int tiles, int rows, int clips, float* sink) {
    const int tt = threadIdx.x % RUN, kq = threadIdx.x / RUN;
    float acc = 0.f;
    for (int clip = blockIdx.x; clip < clips; clip += gridDim.x)
        for (int tile = 0; tile < tiles; ++tile) {
            const int t = tile * RUN + tt;
            if (t >= T) continue;
            const float2* p = in + (long long)clip * rows * T + t;
#pragma unroll UNROLL
            for (int k = kq; k < rows; k += NT / RUN) {
                const float2 v = p[(long long)k * T];
                acc += v.x + v.y;
            }
        }
    if (acc == 12345.678f) sink[0] = acc;
}
// same walk, 16-B loads: a lane reads 2 consecutive frames of a row (RUN/2 lanes per run)
template <int RUN, int NT, int UNROLL>
__global__ __launch_bounds__(NT) void k_walk16(const float2* __restrict__ in, int T, int tiles, int rows, int clips, float* sink) {
    constexpr int LPR = RUN / 2;
    const int tt = threadIdx.x % LPR, kq = threadIdx.x / LPR;
    float acc = 0.f;
    for (int clip = blockIdx.x; clip < clips; clip += gridDim.x)
        for (int tile = 0; tile < tiles; ++tile) {
            const int t = tile * RUN + 2 * tt;
            if (t >= T) continue;
            const float2* p = in + (long long)clip * rows * T + t;
#pragma unroll UNROLL
            for (int k = kq; k < rows; k += NT / LPR) {
                const float4 v = *reinterpret_cast<const float4*>(p + (long long)k * T);
                acc += v.x + v.y + v.z + v.w;
            }
        }
    if (acc == 12345.678f) sink[0] = acc;
}

int main() {
    const int clips = 1024, rows = 2048, T = 432;
    const long long n = (long long)clips * rows * T;
    float2* d;
    float* sink;
    CK(hipMalloc(&d, n * 8));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(d, 0, n * 8));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    auto run = [&](const char* name, auto launch) {
        launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        for (int r = 0; r < 5; ++r) launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        ms /= 5;
        printf("%-44s %7.3f ms  %7.1f GB/s\n", name, ms, n * 8 / ms / 1e6);
    };
    run("linear float4, 2048 blocks", [&] { k_linear<<<2048, 1024>>>(reinterpret_cast<const float4*>(d), n / 2, sink); });
    run("tile RUN=16 (128 B) 1024 thr, unroll 8", [&] { k_tile<16, 1024, 8><<<clips * 27, 1024>>>(d, T, 27, rows, sink); });
    run("tile RUN=8 (64 B) 1024 thr, unroll 8", [&] { k_tile<8, 1024, 8><<<clips * 54, 1024>>>(d, T, 54, rows, sink); });
    run("walk RUN=16 1024 thr x256, unroll 1", [&] { k_walk<16, 1024, 1><<<256, 1024>>>(d, T, 27, rows, clips, sink); });
    run("walk RUN=16 1024 thr x256, unroll 2", [&] { k_walk<16, 1024, 2><<<256, 1024>>>(d, T, 27, rows, clips, sink); });
    run("walk RUN=16 1024 thr x256, unroll 4", [&] { k_walk<16, 1024, 4><<<256, 1024>>>(d, T, 27, rows, clips, sink); });
    run("walk RUN=16 1024 thr x256, unroll 8", [&] { k_walk<16, 1024, 8><<<256, 1024>>>(d, T, 27, rows, clips, sink); });
    run("walk RUN=16 1024 thr x256, unroll 16", [&] { k_walk<16, 1024, 16><<<256, 1024>>>(d, T, 27, rows, clips, sink); });
    run("walk RUN=16 1024 thr x256, unroll 32", [&] { k_walk<16, 1024, 32><<<256, 1024>>>(d, T, 27, rows, clips, sink); });
    run("walk RUN=16 512 thr x256, unroll 2", [&] { k_walk<16, 512, 2><<<256, 512>>>(d, T, 27, rows, clips, sink); });
    run("walk RUN=16 512 thr x256, unroll 4", [&] { k_walk<16, 512, 4><<<256, 512>>>(d, T, 27, rows, clips, sink); });
    run("walk RUN=16 512 thr x256, unroll 8", [&] { k_walk<16, 512, 8><<<256, 512>>>(d, T, 27, rows, clips, sink); });
    run("walk RUN=16 512 thr x256, unroll 16", [&] { k_walk<16, 512, 16><<<256, 512>>>(d, T, 27, rows, clips, sink); });
    run("walk RUN=16 512 thr x256, unroll 32", [&] { k_walk<16, 512, 32><<<256, 512>>>(d, T, 27, rows, clips, sink); });
    run("walk RUN=16 512 thr x512, unroll 8", [&] { k_walk<16, 512, 8><<<512, 512>>>(d, T, 27, rows, clips, sink); });
    run("walk RUN=16 256 thr x256, unroll 8", [&] { k_walk<16, 256, 8><<<256, 256>>>(d, T, 27, rows, clips, sink); });
    run("walk RUN=16 256 thr x256, unroll 16", [&] { k_walk<16, 256, 16><<<256, 256>>>(d, T, 27, rows, clips, sink); });
    return 0;
}
