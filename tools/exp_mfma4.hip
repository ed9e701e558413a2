// exp_mfma4.hip -- operand layout of v_mfma_f32_4x4x1_16b_f32 on gfx950 (the instruction behind k_cqt's contraction).
//   hipcc -O2 --offload-arch=gfx950 tools/exp_mfma4.hip -o tools/bin/exp_mfma4 && tools/bin/exp_mfma4
// Every lane supplies a = 100 + lane, b = 1000 + lane; the four results of every lane tell which (a-lane, b-lane) pair each
// accumulator register holds.
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k(float* out) {
    const int lane = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(100.f + lane, 1000.f + lane, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = acc[r];
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 4; ++r) {
            // expected: D[i = r][j = lane % 4] of block lane / 4 = a(lane 4 blk + r) * b(lane)
            const float want = (100.f + 4 * (lane / 4) + r) * (1000.f + lane);
            if (h[lane * 4 + r] != want) {
                if (bad < 8) printf("lane %d reg %d: got %.0f want %.0f\n", lane, r, h[lane * 4 + r], want);
                ++bad;
            }
        }
    printf(bad ? "layout differs (%d)\n" : "layout as assumed: reg r of lane (blk, j) = a(blk, r) * b(blk, j)\n", bad);
    return bad != 0;
}
