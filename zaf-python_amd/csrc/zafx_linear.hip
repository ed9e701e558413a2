// zafx_linear.hip -- batched dense linear map y = M x per clip, on the matrix cores.
//
// Carries the single-vector transforms of the reference that are NOT windowed:
// zaf.dct / zaf.dst, types I-IV (zaf.py:703-981).  The reference evaluates each by an FFT of a
// 2N-2 / 4N / 8N / 2N+2 point symmetric extension; every one of them is an orthonormal N x N real
// matrix, so a batch of B vectors is one (N x N) . (N x B) GEMM.  At the sizes the reference uses
// (N ~ 1024) that is 2 MFLOP per vector -- 36 us per 1024 vectors at f32 MFMA rate -- works for ANY
// length (the FFT route would need N-1, N or N+1 to be a power of two per type), and is exact f32.
//
// Plain LDS-tiled MFMA GEMM: 64 x 64 output tile per 256-thread workgroup, K tile 32, each wave a
// 32 x 32 sub-tile as 2 x 2 v_mfma_f32_16x16x4_f32 accumulators; LDS rows padded to 33 floats so the
// fragment reads (16 rows x 4 consecutive k) hit distinct banks.
#include "zafx_internal.hpp"

namespace zafx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kLinTile = 64, kLinK = 32, kLinPitch = kLinK + 1;

__global__ __launch_bounds__(256) void k_linear(const float* __restrict__ mat, const float* __restrict__ x, float* __restrict__ y,
                                                int n_rows, int n_cols, long long n_clips) {
    __shared__ float As[kLinTile * kLinPitch];   // M tile: [row][k]
    __shared__ float Bs[kLinTile * kLinPitch];   // X tile: [clip][k]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.x * kLinTile;
    const long long b0 = (long long)blockIdx.y * kLinTile;
    const int wr = (wave >> 1) * 32, wb = (wave & 1) * 32;   // this wave's 32 x 32 corner of the tile
    const int fi = lane & 15, fk = lane >> 4;
    f32x4 acc[2][2] = {};
    for (int k0 = 0; k0 < n_cols; k0 += kLinK) {
        // stage both tiles: 64 rows x 32 floats each, 8 threads per row (4 consecutive floats per thread)
        for (int idx = tid; idx < kLinTile * (kLinK / 4); idx += 256) {
            const int row = idx / (kLinK / 4), kq = (idx % (kLinK / 4)) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + kq + j;
                As[row * kLinPitch + kq + j] = (r0 + row < n_rows && k < n_cols) ? mat[(long long)(r0 + row) * n_cols + k] : 0.f;
                Bs[row * kLinPitch + kq + j] = (b0 + row < n_clips && k < n_cols) ? x[(b0 + row) * n_cols + k] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < kLinK; ks += 4) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = As[(wr + 16 * i + fi) * kLinPitch + ks + fk];
                b[i] = Bs[(wb + 16 * i + fi) * kLinPitch + ks + fk];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // D[row = 4 (lane >> 4) + reg][col = lane & 15]: rows are output coefficients, columns are clips
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long long clip = b0 + wb + 16 * j + fi;
            if (clip >= n_clips) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = r0 + wr + 16 * i + 4 * fk + r;
                if (row < n_rows) y[clip * n_rows + row] = acc[i][j][r];
            }
        }
}

const char* linear_kernel_name() { return "k_linear"; }

hipError_t launch_linear(const zafx_plan& pl, const float* x, float* y, int64_t n_clips) {
    if (n_clips <= 0) return hipSuccess;
    const int n_rows = pl.prm.n_filters, n_cols = pl.W;
    const dim3 grid((unsigned)((n_rows + kLinTile - 1) / kLinTile), (unsigned)((n_clips + kLinTile - 1) / kLinTile));
    hipLaunchKernelGGL(k_linear, grid, dim3(256), 0, pl.stream, pl.d_matrix, x, y, n_rows, n_cols, (long long)n_clips);
    return hipGetLastError();
}

}  // namespace zafx
