// Microbenchmark: what does the memory system give for the STFT output pattern?
// out[clip][row][t] complex64, rows = 2048, T = 432; a workgroup writes a [2048 rows][RUN frames] tile.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// MODE 0: tile pattern, lanes: tt = tid % RUN (frames), kq = tid / RUN ; loop over rows
template <int RUN, int NT>
__global__ __launch_bounds__(NT) void k_tile(float2* __restrict__ out, int T, int tiles, int rows, int xcdmap) {
    int b = blockIdx.x;
    if (xcdmap) {   // all tiles of a clip on one XCD, consecutive
        const int xcd = b & 7, idx = b >> 3;
        const int cl = idx / tiles, tile = idx % tiles;
        b = (cl * 8 + xcd) * tiles + tile;
    }
    const int clip = b / tiles, tile = b % tiles;
    const int tt = threadIdx.x % RUN, kq = threadIdx.x / RUN;
    const int t = tile * RUN + tt;
    if (t >= T) return;
    float2* o = out + (long long)clip * rows * T + t;
    const float2 val = make_float2((float)threadIdx.x, (float)b);
    for (int k = kq; k < rows; k += NT / RUN) o[(long long)k * T] = val;
}
// MODE 1: linear (copy-like) stores of the same total bytes
__global__ __launch_bounds__(1024) void k_linear(float2* __restrict__ out, long long n) {
    const long long i0 = (long long)blockIdx.x * 1024 * 32 + threadIdx.x;
    const float2 val = make_float2((float)threadIdx.x, 1.f);
#pragma unroll
    for (int j = 0; j < 32; ++j) { long long i = i0 + j * 1024; if (i < n) out[i] = val; }
}
__global__ __launch_bounds__(1024) void k_linear4(float4* __restrict__ out, long long n4) {
    const long long i0 = (long long)blockIdx.x * 1024 * 16 + threadIdx.x;
    const float4 val = make_float4((float)threadIdx.x, 1.f, 2.f, 3.f);
#pragma unroll
    for (int j = 0; j < 16; ++j) { long long i = i0 + j * 1024; if (i < n4) out[i] = val; }
}
// FT tile with 16-B stores: a lane writes 2 consecutive frames of one row (8 lanes per 128-B run)
template <int NT>
__global__ __launch_bounds__(NT) void k_tile16B(float4* __restrict__ out, int T, int tiles, int rows) {
    const int clip = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const int tp = threadIdx.x % 8, kq = threadIdx.x / 8;
    const int t = tile * 16 + 2 * tp;
    if (t >= T) return;
    float4* o = reinterpret_cast<float4*>(reinterpret_cast<float2*>(out) + (long long)clip * rows * T + t);
    const float4 val = make_float4((float)threadIdx.x, 1.f, 2.f, 3.f);
    for (int k = kq; k < rows; k += NT / 8) o[(long long)k * T / 2] = val;
}
// MODE 2: frame-major: each frame's 2048 bins contiguous (TF layout)
__global__ __launch_bounds__(1024) void k_tf(float2* __restrict__ out, int T, int tiles) {
    const int clip = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const int slot = threadIdx.x / 64, p = threadIdx.x % 64;
    const int t = tile * 16 + slot;
    if (t >= T) return;
    float2* o = out + ((long long)clip * T + t) * 2048;
    const float2 val = make_float2((float)threadIdx.x, 1.f);
#pragma unroll
    for (int i = 0; i < 32; ++i) o[p + 64 * i] = val;
}


// MODE 3: read the tile's input samples (16 frames x 1024 new samples + halo) and write the FT tile; no compute
template <int NT>
__global__ __launch_bounds__(NT) void k_rw(const float* __restrict__ x, float2* __restrict__ out, long long n_samples, int T, int tiles, int rows) {
    const int clip = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const float4* xc = reinterpret_cast<const float4*>(x + (long long)clip * n_samples + (long long)tile * 16 * 1024);
    // 17 * 1024 floats = 4352 float4
    float acc = 0.f;
    for (int i = threadIdx.x; i < 4352; i += NT) {
        long long idx = (long long)tile * 16 * 1024 + 4LL * i;
        if (idx + 3 < n_samples) { float4 v = xc[i]; acc += v.x + v.y + v.z + v.w; }
    }
    const int tt = threadIdx.x % 16, kq = threadIdx.x / 16;
    const int t = tile * 16 + tt;
    if (t >= T) return;
    float2* o = out + (long long)clip * rows * T + t;
    const float2 val = make_float2(acc, (float)blockIdx.x);
    for (int k = kq; k < rows; k += NT / 16) o[(long long)k * T] = val;
}
// MODE 4: persistent version of MODE 3 (256*WGPC blocks loop over tiles)
template <int NT>
__global__ __launch_bounds__(NT) void k_rw_p(const float* __restrict__ x, float2* __restrict__ out, long long n_samples, int T, int tiles, int rows, int total) {
    for (int b = blockIdx.x; b < total; b += gridDim.x) {
        const int clip = b / tiles, tile = b % tiles;
        const float4* xc = reinterpret_cast<const float4*>(x + (long long)clip * n_samples + (long long)tile * 16 * 1024);
        float acc = 0.f;
        for (int i = threadIdx.x; i < 4352; i += NT) {
            long long idx = (long long)tile * 16 * 1024 + 4LL * i;
            if (idx + 3 < n_samples) { float4 v = xc[i]; acc += v.x + v.y + v.z + v.w; }
        }
        const int tt = threadIdx.x % 16, kq = threadIdx.x / 16;
        const int t = tile * 16 + tt;
        if (t < T) {
            float2* o = out + (long long)clip * rows * T + t;
            const float2 val = make_float2(acc, (float)b);
            for (int k = kq; k < rows; k += NT / 16) o[(long long)k * T] = val;
        }
    }
}


// MODE 5: read-only FT tile pattern (ISTFT input): lanes along t, loop over rows, UNR loads in flight
template <int RUN, int NT, int UNR>
__global__ __launch_bounds__(NT) void k_tile_read(const float2* __restrict__ in, float* __restrict__ sink, int T, int tiles, int rows) {
    const int clip = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const int tt = threadIdx.x % RUN, kq = threadIdx.x / RUN;
    const int t = tile * RUN + tt;
    if (t >= T) return;
    const float2* o = in + (long long)clip * rows * T + t;
    float acc = 0.f;
    constexpr int STEP = NT / RUN;
    for (int k = kq; k < rows; k += STEP * UNR) {
        float2 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) v[u] = o[(long long)(k + u * STEP) * T];
#pragma unroll
        for (int u = 0; u < UNR; ++u) acc += v[u].x + v[u].y;
    }
    if (acc == 12345.f) sink[0] = acc;
}


// MODE 6: store-only, persistent on a SUBSET of the CUs: is the per-CU store rate capped?
template <int NT>
__global__ __launch_bounds__(NT) void k_tile_p(float2* __restrict__ out, int T, int tiles, int rows, int total) {
    for (int b = blockIdx.x; b < total; b += gridDim.x) {
        const int clip = b / tiles, tile = b % tiles;
        const int tt = threadIdx.x % 16, kq = threadIdx.x / 16;
        const int t = tile * 16 + tt;
        if (t >= T) continue;
        float2* o = out + (long long)clip * rows * T + t;
        const float2 val = make_float2((float)threadIdx.x, (float)b);
        for (int k = kq; k < rows; k += NT / 16) o[(long long)k * T] = val;
    }
}

template <class F> void timeit(const char* name, double bytes, F launch) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    printf("%-44s %.3f ms  %.0f GB/s\n", name, ms, bytes / ms / 1e6);
}

int main() {
    const int B = 1024, T = 432, rows = 2048;
    const long long n = (long long)B * rows * T;
    float2* out; CK(hipMalloc(&out, n * 8)); CK(hipMemset(out, 0, n * 8));
    const double bytes = (double)n * 8;
    timeit("linear float2 stores", bytes, [&] { hipLaunchKernelGGL(k_linear, dim3((unsigned)((n + 32767) / 32768)), dim3(1024), 0, 0, out, n); });
    timeit("linear float4 stores", bytes, [&] { hipLaunchKernelGGL(k_linear4, dim3((unsigned)((n / 2 + 16383) / 16384)), dim3(1024), 0, 0, (float4*)out, n / 2); });
    timeit("FT tile, 16-B stores (2 frames per lane)", bytes, [&] { hipLaunchKernelGGL((k_tile16B<1024>), dim3(27 * B), dim3(1024), 0, 0, (float4*)out, T, 27, rows); });
    timeit("TF layout (frame-major 16 KB runs)", bytes, [&] { hipLaunchKernelGGL(k_tf, dim3(27 * B), dim3(1024), 0, 0, out, T, 27); });
    timeit("FT tile RUN=16 (128 B runs) 1024 thr", bytes, [&] { hipLaunchKernelGGL((k_tile<16, 1024>), dim3(27 * B), dim3(1024), 0, 0, out, T, 27, rows, 0); });
    timeit("FT tile RUN=16 (128 B) xcd-clip map", bytes, [&] { hipLaunchKernelGGL((k_tile<16, 1024>), dim3(27 * B), dim3(1024), 0, 0, out, T, 27, rows, 1); });
    timeit("FT tile RUN=16 (128 B) 256 thr", bytes, [&] { hipLaunchKernelGGL((k_tile<16, 256>), dim3(27 * B), dim3(256), 0, 0, out, T, 27, rows, 0); });
    timeit("FT tile RUN=32 (256 B runs)", bytes, [&] { hipLaunchKernelGGL((k_tile<32, 1024>), dim3(14 * B), dim3(1024), 0, 0, out, T, 14, rows, 0); });
    timeit("FT tile RUN=64 (512 B runs)", bytes, [&] { hipLaunchKernelGGL((k_tile<64, 1024>), dim3(7 * B), dim3(1024), 0, 0, out, T, 7, rows, 0); });
    timeit("FT tile RUN=8 (64 B runs)", bytes, [&] { hipLaunchKernelGGL((k_tile<8, 1024>), dim3(54 * B), dim3(1024), 0, 0, out, T, 54, rows, 0); });
    {
        const long long ns = 441000; float* x; CK(hipMalloc(&x, (size_t)B * ns * 4)); CK(hipMemset(x, 0, (size_t)B * ns * 4));
        const double rw = bytes + (double)B * ns * 4;
        timeit("read input + FT tile write, 1024 thr", rw, [&] { hipLaunchKernelGGL((k_rw<1024>), dim3(27 * B), dim3(1024), 0, 0, x, out, ns, T, 27, rows); });
        timeit("read input + FT tile write, 256 thr", rw, [&] { hipLaunchKernelGGL((k_rw<256>), dim3(27 * B), dim3(256), 0, 0, x, out, ns, T, 27, rows); });
        timeit("read+write persistent 1024 thr x 256", rw, [&] { hipLaunchKernelGGL((k_rw_p<1024>), dim3(256), dim3(1024), 0, 0, x, out, ns, T, 27, rows, 27 * B); });
        timeit("read+write persistent 1024 thr x 512", rw, [&] { hipLaunchKernelGGL((k_rw_p<1024>), dim3(512), dim3(1024), 0, 0, x, out, ns, T, 27, rows, 27 * B); });
        timeit("read+write persistent 512 thr x 1024", rw, [&] { hipLaunchKernelGGL((k_rw_p<512>), dim3(1024), dim3(512), 0, 0, x, out, ns, T, 27, rows, 27 * B); });
        CK(hipFree(x));
    }
    {
        float* sink; CK(hipMalloc(&sink, 64));
        timeit("FT tile READ RUN=16 1024 thr unr 4", bytes, [&] { hipLaunchKernelGGL((k_tile_read<16, 1024, 4>), dim3(27 * B), dim3(1024), 0, 0, out, sink, T, 27, rows); });
        timeit("FT tile READ RUN=16 1024 thr unr 16", bytes, [&] { hipLaunchKernelGGL((k_tile_read<16, 1024, 16>), dim3(27 * B), dim3(1024), 0, 0, out, sink, T, 27, rows); });
        timeit("FT tile READ RUN=16 256 thr unr 16", bytes, [&] { hipLaunchKernelGGL((k_tile_read<16, 256, 16>), dim3(27 * B), dim3(256), 0, 0, out, sink, T, 27, rows); });
        timeit("FT tile READ RUN=32 1024 thr unr 8", bytes, [&] { hipLaunchKernelGGL((k_tile_read<32, 1024, 8>), dim3(14 * B), dim3(1024), 0, 0, out, sink, T, 14, rows); });
        timeit("FT tile READ RUN=8 1024 thr unr 8", bytes, [&] { hipLaunchKernelGGL((k_tile_read<8, 1024, 8>), dim3(54 * B), dim3(1024), 0, 0, out, sink, T, 54, rows); });
    }
    for (int g : {32, 64, 128, 256, 512}) {
        char nm[96]; snprintf(nm, 96, "store-only persistent, %d WGs x 512 thr (1/8 of the data)", g);
        timeit(nm, bytes / 8, [&] { hipLaunchKernelGGL((k_tile_p<512>), dim3(g), dim3(512), 0, 0, out, T, 27, rows, 27 * B / 8); });
    }
    // padded T = 512 (4 KB row pitch): does the 3456-B pitch matter?
    {
        const int T2 = 512; const long long n2 = (long long)B * rows * T2; float2* o2; CK(hipMalloc(&o2, n2 * 8));
        const double bytes2 = (double)B * rows * 432 * 8;
        timeit("FT tile RUN=16, row pitch 4096 B (T pad 512)", bytes2, [&] { hipLaunchKernelGGL((k_tile<16, 1024>), dim3(27 * B), dim3(1024), 0, 0, o2, T2, 27, rows, 0); });
        CK(hipFree(o2));
    }
    return 0;
}
