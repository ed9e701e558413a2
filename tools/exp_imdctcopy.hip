// What does the memory system give for the IMDCT's traffic with no arithmetic at all?
// in: coefs[clip][1024 rows][TP] float32 (time-minor), a workgroup walks 32-frame tiles of a clip in order: per tile
// 1024 rows x 128 B gathered (16-byte lanes, 8 lanes per row, as k_imdct), 32 x 1024 floats written linearly.
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/exp_imdctcopy tools/exp_imdctcopy.hip && tools/bin/exp_imdctcopy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <bool PREFETCH, bool SHIFT = false>
__global__ __launch_bounds__(1024) void k_copy(const float* __restrict__ in, float* __restrict__ out, int T, int TP, int tiles, int clips, long long out_len) {
    constexpr int M = 1024, FPB = 32;
    const int tid = threadIdx.x, fs4 = (tid % 8) * 4, mq = tid / 8;
    float4 r[8];
    auto gather = [&](int clip, int tile) {
        const int t = tile * FPB + fs4;
        const float* cp = in + (long long)clip * M * TP + t;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // SHIFT: the odd rows (64 bytes off the 128-byte grid when the pitch is 432 floats) are read 16 columns further on, i.e. as
            // whole lines -- what a kernel that keeps the second half of such a line for the next tile would fetch
            const int sh = (SHIFT && ((mq + i * 128) & 1) && t + 16 + 3 < TP) ? 16 : 0;
            r[i] = (t < TP) ? *reinterpret_cast<const float4*>(cp + (long long)(mq + i * 128) * TP + sh) : make_float4(0, 0, 0, 0);
        }
    };
    for (int clip = blockIdx.x; clip < clips; clip += gridDim.x) {
        if (PREFETCH) gather(clip, 0);
        for (int tile = 0; tile < tiles; ++tile) {
            if (!PREFETCH) gather(clip, tile);
            float4 s[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) s[i] = r[i];
            if (PREFETCH && tile + 1 < tiles) gather(clip, tile + 1);
            float2* o = reinterpret_cast<float2*>(out + (long long)clip * out_len + (long long)tile * FPB * M) + tid;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                o[(2 * i) * 1024] = make_float2(s[i].x, s[i].y);
                o[(2 * i + 1) * 1024] = make_float2(s[i].z, s[i].w);
            }
        }
    }
}

// the forward direction: 32 x 1024 samples read linearly, 1024 rows x 128 B written (8-byte lanes, 16 lanes per row, as k_mdct_ft32)
__global__ __launch_bounds__(1024) void k_copy_fwd(const float* __restrict__ in, float* __restrict__ out, int T, int TP, int tiles, int clips, long long in_len) {
    constexpr int M = 1024, FPB = 32;
    const int tid = threadIdx.x, tp = tid % 16, fq = tid / 16;
    for (long long tl = blockIdx.x; tl < (long long)clips * tiles; tl += gridDim.x) {
        const int clip = (int)(tl / tiles), tile = (int)(tl % tiles);
        const float4* src = reinterpret_cast<const float4*>(in + (long long)clip * in_len + (long long)tile * FPB * M) + tid;
        float4 r[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = src[i * 1024];
        const int t = tile * FPB + 2 * tp;
        if (t + 1 < TP) {
            float* o = out + (long long)clip * M * TP + t;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                *reinterpret_cast<float2*>(o + (long long)(fq + 64 * (2 * i)) * TP) = make_float2(r[i].x, r[i].y);
                *reinterpret_cast<float2*>(o + (long long)(fq + 64 * (2 * i + 1)) * TP) = make_float2(r[i].z, r[i].w);
            }
        }
    }
}

int main() {
    const int clips = 1024, T = 431, M = 1024;
    for (int TP : {432, 448}) {
        const int tiles = (T + 31) / 32;
        const long long out_len = (long long)tiles * 32 * M;
        float *in, *out;
        CK(hipMalloc(&in, (size_t)clips * M * TP * 4 + 4096));
        CK(hipMalloc(&out, (size_t)clips * out_len * 4));
        CK(hipMemset(in, 0, (size_t)clips * M * TP * 4));
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int pf = 0; pf < 4; ++pf) {   // 2, 3: odd rows shifted onto the line grid
            float best = 1e9f;
            for (int it = 0; it < 6; ++it) {
                hipEventRecord(e0);
                if (pf == 1) hipLaunchKernelGGL(k_copy<true>, dim3(256), dim3(1024), 0, 0, in, out, T, TP, tiles, clips, out_len);
                else if (pf == 0) hipLaunchKernelGGL(k_copy<false>, dim3(256), dim3(1024), 0, 0, in, out, T, TP, tiles, clips, out_len);
                else if (pf == 3) hipLaunchKernelGGL((k_copy<true, true>), dim3(256), dim3(1024), 0, 0, in, out, T, TP, tiles, clips, out_len);
                else hipLaunchKernelGGL((k_copy<false, true>), dim3(256), dim3(1024), 0, 0, in, out, T, TP, tiles, clips, out_len);
                hipEventRecord(e1);
                CK(hipEventSynchronize(e1));
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (it && ms < best) best = ms;
            }
            const double gb = ((double)clips * M * T * 4 + (double)clips * tiles * 32 * M * 4) / 1e9;
            printf("row pitch %d floats, prefetch %d%s: %.3f ms  (%.2f GB moved, %.2f TB/s)\n", TP, pf & 1, pf >= 2 ? ", odd rows shifted by 16 columns" : "", best, gb, gb / best);
        }
        {
            float best = 1e9f;
            for (int it = 0; it < 6; ++it) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(k_copy_fwd, dim3(256), dim3(1024), 0, 0, out, in, T, TP, tiles, clips, out_len);
                hipEventRecord(e1);
                CK(hipEventSynchronize(e1));
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (it && ms < best) best = ms;
            }
            const double gb = ((double)clips * M * T * 4 + (double)clips * tiles * 32 * M * 4) / 1e9;
            printf("row pitch %d floats, forward (linear reads, row stores): %.3f ms  (%.2f GB moved, %.2f TB/s)\n", TP, best, gb, gb / best);
        }
        hipFree(in); hipFree(out);
    }
    return 0;
}
