// What does the memory system give for the ISTFT's traffic (config 2 shapes) with no arithmetic at all?
// in: spec[clip][2048 rows][TP] complex64 (time-minor); a workgroup takes tiles of 16 frames: 2048 rows x 128 B gathered,
// 16 x 1024 floats written linearly.  Variants: 8-byte lanes (16 per row) / 16-byte lanes (8 per row), with and without the next
// tile requested ahead of the stores, tiles dealt round-robin or in the XCD-aware order of the library.
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/exp_istftcopy tools/exp_istftcopy.hip && tools/bin/exp_istftcopy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ long long xcd_order(long long v, long long total) {   // consecutive tiles to the workgroups of one XCD
    const long long per = total / 8;
    return (v % 8) * per + v / 8;
}

template <int LANE_BYTES, bool PREFETCH, bool XCD>
__global__ __launch_bounds__(1024) void k_copy(const float2* __restrict__ in, float* __restrict__ out, int T, int TP, int tiles, long long total, long long out_len) {
    constexpr int W = 2048, FPB = 16, LPR = FPB * 8 / LANE_BYTES, RPP = 1024 / LPR, NL = W / RPP, V = LANE_BYTES / 8;
    const int tid = threadIdx.x, fl = (tid % LPR) * V, rq = tid / LPR;
    float2 r[NL][V];
    auto gather = [&](long long tlv) {
        const long long tl = (XCD && total % 8 == 0) ? xcd_order(tlv, total) : tlv;
        const int clip = (int)(tl / tiles), tile = (int)(tl % tiles);
        const int t = tile * FPB + fl;
        const float2* cp = in + (long long)clip * W * TP + t;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            if (V == 1) r[i][0] = (t < TP) ? cp[(long long)(rq + i * RPP) * TP] : make_float2(0, 0);
            else {
                const float4 q = (t < TP) ? *reinterpret_cast<const float4*>(cp + (long long)(rq + i * RPP) * TP) : make_float4(0, 0, 0, 0);
                r[i][0] = make_float2(q.x, q.y);
                r[i][V - 1] = make_float2(q.z, q.w);
            }
        }
    };
    if (PREFETCH && blockIdx.x < total) gather(blockIdx.x);
    for (long long tlv = blockIdx.x; tlv < total; tlv += gridDim.x) {
        const long long tl = (XCD && total % 8 == 0) ? xcd_order(tlv, total) : tlv;
        const int clip = (int)(tl / tiles), tile = (int)(tl % tiles);
        if (!PREFETCH) gather(tlv);
        float s[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) s[j] = 0.f;
#pragma unroll
        for (int i = 0; i < NL; ++i)
#pragma unroll
            for (int v = 0; v < V; ++v) {
                s[(i * V + v) % 16] += r[i][v].x;
                s[(i * V + v + 7) % 16] += r[i][v].y;
            }
        if (PREFETCH && tlv + gridDim.x < total) gather(tlv + gridDim.x);
        float2* o = reinterpret_cast<float2*>(out + (long long)clip * out_len + (long long)tile * FPB * 1024) + tid;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i * 1024] = make_float2(s[2 * i], s[2 * i + 1]);
    }
}

int main() {
    const int clips = 1024, T = 432, W = 2048;
    for (int TP : {432, 448}) {
        const int tiles = (T + 15) / 16;
        const long long out_len = (long long)tiles * 16 * 1024, total = (long long)clips * tiles;
        float2* in;
        float* out;
        CK(hipMalloc(&in, (size_t)clips * W * TP * 8 + 4096));
        CK(hipMalloc(&out, (size_t)clips * out_len * 4));
        CK(hipMemset(in, 0, (size_t)clips * W * TP * 8));
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        auto run = [&](const char* name, void (*k)(const float2*, float*, int, int, int, long long, long long), int grid) {
            float best = 1e9f;
            for (int it = 0; it < 6; ++it) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3(grid), dim3(1024), 0, 0, in, out, T, TP, tiles, total, out_len);
                hipEventRecord(e1);
                CK(hipEventSynchronize(e1));
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (it && ms < best) best = ms;
            }
            const double gb = ((double)clips * W * T * 8 + (double)clips * tiles * 16 * 1024 * 4) / 1e9;
            printf("pitch %d grid %d %-28s %.3f ms  (%.2f GB, %.2f TB/s)\n", TP, grid, name, best, gb, gb / best);
        };
        for (int grid : {256, 512}) {
            run("8B", k_copy<8, false, false>, grid);
            run("8B xcd", k_copy<8, false, true>, grid);
            run("16B", k_copy<16, false, false>, grid);
            run("16B xcd", k_copy<16, false, true>, grid);
            run("8B prefetch", k_copy<8, true, false>, grid);
            run("8B prefetch xcd", k_copy<8, true, true>, grid);
            run("16B prefetch", k_copy<16, true, false>, grid);
            run("16B prefetch xcd", k_copy<16, true, true>, grid);
        }
        hipFree(in); hipFree(out);
    }
    return 0;
}
