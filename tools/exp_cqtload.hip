// Experiment: how fast can one CU pull the 128-KB CQT frames (32768 samples, hop 1764) into registers?
// 1024 threads per workgroup, one workgroup per CU, frames dealt either as tiles of 16 consecutive frames per workgroup
// (mode 0) or round-robin over the workgroups of an XCD (mode 1); lanes load 8 bytes (16 loads) or 16 bytes (8 loads).
//   hipcc -O3 --offload-arch=gfx950 tools/exp_cqtload.hip -o tools/bin/exp_cqtload && tools/bin/exp_cqtload
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2 __attribute__((ext_vector_type(2)));
typedef float v4 __attribute__((ext_vector_type(4)));

template <int MODE, int BYTES>
__global__ __launch_bounds__(1024) void k_load(const float* __restrict__ x, float* __restrict__ sink, long long n_samples, int T, int n_clips,
                                                int frames_per_wg, int inflight_waves) {
    const int p = threadIdx.x;
    float acc = 0.f;
    const int nb = gridDim.x;
    for (int f = 0; f < frames_per_wg; ++f) {
        long long g;   // global frame index
        if (MODE == 0) {   // tiles of 16 consecutive frames, block-cyclic over all workgroups
            const int tile = (f / 16) * nb + blockIdx.x;
            g = (long long)tile * 16 + f % 16;
        } else {   // XCD group = blockIdx % 8 owns clips group, group + 8, ...; frames round-robin inside the group
            const int group = blockIdx.x % 8, slot = blockIdx.x / 8, slots = nb / 8;
            const long long gi = (long long)f * slots + slot;   // index in the group's list
            g = ((gi / T) * 8 + group) * T + gi % T;
        }
        const int clip = (int)(g / T) % n_clips, t = (int)(g % T);
        const float* xc = x + (long long)clip * n_samples + (long long)t * 1764;
        if (BYTES == 8) {
            v2 v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = *reinterpret_cast<const v2*>(xc + 2 * (p + i * 1024));
#pragma unroll
            for (int i = 0; i < 16; ++i) acc += v[i].x + v[i].y;
        } else {
            v4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const v4*>(xc + 4 * (p + i * 1024));
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += v[i].x + v[i].y + v[i].z + v[i].w;
        }
        __syncthreads();
    }
    if (acc == 123.456f) sink[0] = acc;
}

int main() {
    const int n_clips = 1024, T = 730;   // frames that stay inside the clip
    const long long n_samples = 1323000;
    float* x;
    float* sink;
    hipMalloc(&x, (size_t)n_clips * n_samples * 4);
    hipMalloc(&sink, 4);
    hipMemset(x, 0, (size_t)n_clips * n_samples * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int frames = 480;
    auto run = [&](auto kern, const char* name) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 0, 0, x, sink, n_samples, T, n_clips, frames, 0);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double bytes = 256.0 * frames * 131072;
        printf("%-44s %8.3f ms  %7.1f GB/s total  %6.1f B/clk/CU (2.4 GHz)  %7.0f cycles/frame\n", name, ms, bytes / ms / 1e6,
               bytes / 256 / (ms * 1e-3 * 2.4e9), ms * 1e-3 * 2.4e9 / frames);
    };
    run(k_load<0, 8>, "tiles of 16 frames, 8-byte lanes");
    run(k_load<0, 16>, "tiles of 16 frames, 16-byte lanes");
    run(k_load<1, 8>, "XCD round-robin frames, 8-byte lanes");
    run(k_load<1, 16>, "XCD round-robin frames, 16-byte lanes");
    return 0;
}
