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

// The same map for the aligned case (n_cols a multiple of 16, 16-byte aligned operands): 128 x 128 output tile per
// 256-thread workgroup, each wave a 64 x 64 corner as 4 x 4 accumulators (64 VGPRs), K tile 32, operands fetched as
// float4 and double-buffered (registers -> LDS) so that the global loads of tile k+1 fly under the MFMAs of tile k;
// one barrier per K tile.  LDS rows padded to 34 floats: a 4-byte LDS read is served 32 lanes at a time over 32 banks, and
// 34 fi + fk = 2 fi + fk (mod 32) is distinct for the 16 x 2 fragment lanes of each half wave (33, the first choice, put them
// two by two: SQ_LDS_BANK_CONFLICT 0.33 of the active cycles; 0.438 -> 0.433 ms).  Results leave as float4 (the 4 accumulator registers of a lane are 4 consecutive output rows).
constexpr int kBigTile = 128, kBigK = 32, kBigPitch = kBigK + 2;

__global__ __launch_bounds__(256) void k_linear128(const float* __restrict__ mat, const float* __restrict__ x, float* __restrict__ y,
                                                   int n_rows, int n_cols, long long n_clips) {
    __shared__ float As[2][kBigTile * kBigPitch];
    __shared__ float Bs[2][kBigTile * kBigPitch];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.x * kBigTile;
    const long long b0 = (long long)blockIdx.y * kBigTile;
    const int wr = (wave >> 1) * 64, wb = (wave & 1) * 64;
    const int fi = lane & 15, fk = lane >> 4;
    // staging: 128 rows x 32 floats = 1024 float4 per operand, 4 per thread: rows (tid / 8) + 32 j, k piece (tid % 8) * 4
    const int srow = tid >> 3, skq = (tid & 7) * 4;
    // Rows / clips past the end are read from the last valid one: their products land in output rows / clips that are not
    // stored.  (A select between the loaded value and a zero constant made the compiler select between the ADDRESSES of the
    // row and of a zero on the stack: 32 flat 4-byte loads per thread and K tile instead of 8 global 16-byte loads.)
    const float* ap[4];
    const float* bp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ap[j] = mat + (long long)min(r0 + srow + 32 * j, n_rows - 1) * n_cols + skq;
        bp[j] = x + min(b0 + srow + 32 * j, n_clips - 1) * n_cols + skq;
    }
    float4 ra[4], rb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ra[j] = *reinterpret_cast<const float4*>(ap[j] + k0);
            rb[j] = *reinterpret_cast<const float4*>(bp[j] + k0);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float* a = As[buf] + (srow + 32 * j) * kBigPitch + skq;
            float* b = Bs[buf] + (srow + 32 * j) * kBigPitch + skq;
            a[0] = ra[j].x; a[1] = ra[j].y; a[2] = ra[j].z; a[3] = ra[j].w;
            b[0] = rb[j].x; b[1] = rb[j].y; b[2] = rb[j].z; b[3] = rb[j].w;
        }
    };
    f32x4 acc[4][4] = {};
    fetch(0);
    stash(0);
    __syncthreads();
    const int n_tiles = n_cols / kBigK;
    for (int kt = 0; kt < n_tiles; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < n_tiles) fetch((kt + 1) * kBigK);
        const float* as = As[buf] + (wr + fi) * kBigPitch + fk;
        const float* bs = Bs[buf] + (wb + fi) * kBigPitch + fk;
#pragma unroll
        for (int ks = 0; ks < kBigK; ks += 4) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = as[16 * i * kBigPitch + ks];
                b[i] = bs[16 * i * kBigPitch + ks];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < n_tiles) stash(buf ^ 1);   // the other buffer: its last readers passed the previous barrier
        __syncthreads();
    }
    const bool vec_out = n_rows % 4 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long clip = b0 + wb + 16 * j + fi;
            const int row = r0 + wr + 16 * i + 4 * fk;
            if (clip >= n_clips) continue;
            if (vec_out && row + 3 < n_rows) {
                *reinterpret_cast<float4*>(y + clip * n_rows + row) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (row + r < n_rows) y[clip * n_rows + row + r] = acc[i][j][r];
            }
        }
}

const char* linear_kernel_name() { return "k_linear"; }   // prefix of both kernels (rocprofv3 rows are matched by substring)

hipError_t launch_linear(const zafx_plan& pl, const float* x, float* y, int64_t n_clips) {
    if (n_clips <= 0) return hipSuccess;
    const int n_rows = pl.prm.n_filters, n_cols = pl.W;
    if (n_cols % kBigK == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(pl.d_matrix) % 16 == 0 &&
        (long long)n_rows * n_clips >= 128LL * 128 * 64) {   // enough tiles to fill the chip; small problems keep the 64-tile kernel
        const dim3 big((unsigned)((n_rows + kBigTile - 1) / kBigTile), (unsigned)((n_clips + kBigTile - 1) / kBigTile));
        pl.ran = "k_linear128";
        hipLaunchKernelGGL(k_linear128, big, dim3(256), 0, pl.stream, pl.d_matrix, x, y, n_rows, n_cols, (long long)n_clips);
        return hipGetLastError();
    }
    const dim3 grid((unsigned)((n_rows + kLinTile - 1) / kLinTile), (unsigned)((n_clips + kLinTile - 1) / kLinTile));
    pl.ran = "k_linear";
    hipLaunchKernelGGL(k_linear, grid, dim3(256), 0, pl.stream, pl.d_matrix, x, y, n_rows, n_cols, (long long)n_clips);
    return hipGetLastError();
}

}  // namespace zafx
