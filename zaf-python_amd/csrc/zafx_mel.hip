// zafx_mel.hip -- fused melspectrogram / MFCC kernel for gfx950 (MI355X).
//
// One workgroup = 16 consecutive frames of one clip.  The STFT never reaches HBM:
//   framing + window + real FFT (as k_stft)            zaf.py:369 / :436  (stft)
//   |X[k]| or |X[k]|^2 for k = 1..W/2, in place in LDS  zaf.py:370 / :437-439
//   mel = FB . S           -- v_mfma_f32_16x16x4_f32    zaf.py:373 / :445  (np.matmul)
//   log(mel + eps), DCT-II rows 1..ncoef as a second MFMA GEMM   zaf.py:443-452
//
// The filterbank is banded (zaf.py:305-316: each row is one triangle), so only the
// K-steps that hold non-zeros of a 16-row block are multiplied (pack_band in
// zafx_capi.cpp).  A operand: packed FB fragment from global/L2 (one coalesced 256-B
// load per MFMA); B operand: S[t][c] from LDS, frame pitch = 2 (mod 32) dwords so the
// 16 frames x 4 columns of a fragment hit 64 distinct banks.
#include "zafx_fft.hpp"
#include "zafx_internal.hpp"

namespace zafx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int mel_threads(int log2n, int log2e) {
    const int p = fft_threads(log2n, log2e);
    return ((1024 / p) < 16 ? (1024 / p) : 16) * p;
}

template <int LOG2N, int LOG2E>
struct MelCfg {
    using C = FftCfg<LOG2N, LOG2E>;
    static constexpr int FPB = 16;   // MFMA N dimension
    static constexpr int NSLOT = (1024 / C::P) < FPB ? (1024 / C::P) : FPB;   // frames transformed concurrently
    static constexpr int NT = NSLOT * C::P;
    static constexpr size_t SMEM_BASE = (size_t)(FPB * C::PITCH + C::TW) * 8;
};

template <int LOG2N, int LOG2E>
__global__ __launch_bounds__(mel_threads(LOG2N, LOG2E)) void k_mel(
    const float* __restrict__ x, const float* __restrict__ win, const float2* __restrict__ twp,
    const float2* __restrict__ tws, const float* __restrict__ fb_pack, const int* __restrict__ fb_meta, int fb_blocks,
    const float* __restrict__ dct_pack, const int* __restrict__ dct_meta, int dct_blocks, float* __restrict__ out,
    long long n_samples, int hop, int T, int tiles, int n_filters, int n_coefs, int mfcc, int layout) {
    using C = FftCfg<LOG2N, LOG2E>;
    using G = MelCfg<LOG2N, LOG2E>;
    constexpr int N = C::N, P = C::P, E = C::E, NT = G::NT, FPB = G::FPB, NSLOT = G::NSLOT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* frames = reinterpret_cast<float2*>(smem_raw);
    float2* tw_l = frames + FPB * C::PITCH;
    float* ltile = reinterpret_cast<float*>(tw_l + C::TW);   // [fb_blocks*16][16] log-mel (mfcc only)
    const int tid = threadIdx.x;
    for (int i = tid; i < C::TW; i += NT) tw_l[i] = twp[i];
    __syncthreads();

    const int slot = tid / P, p = tid % P;
    const int clip = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const int t0 = tile * FPB;
    const float* xc = x + (long long)clip * n_samples;
    const float2* w2 = reinterpret_cast<const float2*>(win);

    // ---- STFT of the tile's frames, NSLOT at a time; spectrum -> magnitude/power in place
#pragma unroll 1
    for (int f0 = 0; f0 < FPB; f0 += NSLOT) {
        const int fr = f0 + slot;
        const int t = t0 + fr;
        float2* buf = frames + fr * C::PITCH;
        float2 v[E];
        const long long s0 = (long long)t * hop - N;
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const int n = p + i * P;
            const long long s = s0 + 2 * n;
            const float2 wv = w2[n];
            const float a = (t < T && s >= 0 && s < n_samples) ? xc[s] : 0.f;
            const float b = (t < T && s + 1 >= 0 && s + 1 < n_samples) ? xc[s + 1] : 0.f;
            v[i] = make_float2(a * wv.x, b * wv.y);
        }
        fft_frame<LOG2N, LOG2E>(v, buf, p, tw_l);
        // real split of the (k, N-k) pairs this thread owns, kept in registers
        float mk[E / 2], mn[E / 2];
#pragma unroll
        for (int i = 0; i < E / 2; ++i) {
            const int k = p + i * P;
            float2 xk, xn;
            if (k == 0) {
                const float2 z0 = buf[0], zc = buf[phys(N / 2)];
                xk = zc;                                  // |X[N/2]| = |Z[N/2]|
                xn = make_float2(z0.x - z0.y, 0.f);       // X[N] (Nyquist, kept: zaf.py:370)
            } else {
                const float2 zk = buf[phys(k)], zn = buf[phys(N - k)];
                const float2 t_k = tws[k];
                const float2 e = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
                const float2 d = make_float2(0.5f * (zk.x - zn.x), 0.5f * (zk.y + zn.y));
                const float2 to = cmul(t_k, make_float2(d.y, -d.x));
                xk = cadd(e, to);
                xn = csub(e, to);
            }
            const float pk = xk.x * xk.x + xk.y * xk.y, pn = xn.x * xn.x + xn.y * xn.y;
            mk[i] = mfcc ? pk : sqrtf(pk);
            mn[i] = mfcc ? pn : sqrtf(pn);
        }
        frame_sync<P>();   // every Z read of this frame is done before S overwrites it
        float* sf = reinterpret_cast<float*>(buf);   // S[c], c = bin - 1, c = 0..N-1
#pragma unroll
        for (int i = 0; i < E / 2; ++i) {
            const int k = p + i * P;
            if (k == 0) {
                sf[N / 2 - 1] = mk[i];
                sf[N - 1] = mn[i];
            } else {
                sf[k - 1] = mk[i];
                sf[N - k - 1] = mn[i];
            }
        }
    }
    __syncthreads();

    // ---- mel = FB . S on the matrix cores; 16 filters x 16 frames per accumulator
    const int lane = tid & 63, wave = tid >> 6, nwaves = NT >> 6;
    const float* sall = reinterpret_cast<const float*>(frames);
    const int bt = lane & 15, bk = lane >> 4;
    const float* sb = sall + (size_t)bt * (2 * C::PITCH) + bk;
    const float eps = 2.220446049250313e-16f;   // np.finfo(float).eps (zaf.py:445)
    for (int blk = wave; blk < fb_blocks; blk += nwaves) {
        const int first = fb_meta[blk * 4 + 0], steps = fb_meta[blk * 4 + 1], off = fb_meta[blk * 4 + 2];
        const float* ap = fb_pack + (size_t)off * 64 + lane;
        const float* bp = sb + first;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        int s = 0;
        for (; s + 1 < steps; s += 2) {   // two independent accumulators hide the 40-cycle MFMA latency
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[(size_t)s * 64], bp[4 * s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[(size_t)(s + 1) * 64], bp[4 * s + 4], acc1, 0, 0, 0);
        }
        if (s < steps) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[(size_t)s * 64], bp[4 * s], acc0, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float val = acc0[r] + acc1[r];
            const int m = 16 * blk + 4 * bk + r;
            if (mfcc) {
                ltile[m * 16 + bt] = m < n_filters ? logf(val + eps) : 0.f;
            } else if (m < n_filters && t0 + bt < T) {
                if (layout == ZAFX_LAYOUT_FT) out[((long long)clip * n_filters + m) * T + t0 + bt] = val;
                else out[((long long)clip * T + t0 + bt) * n_filters + m] = val;
            }
        }
    }
    if (!mfcc) return;
    __syncthreads();

    // ---- MFCC: rows 1..ncoef of the orthonormal DCT-II over the mel axis, as a second MFMA GEMM
    for (int blk = wave; blk < dct_blocks; blk += nwaves) {
        const int first = dct_meta[blk * 4 + 0], steps = dct_meta[blk * 4 + 1], off = dct_meta[blk * 4 + 2];
        const float* ap = dct_pack + (size_t)off * 64 + lane;
        const float* bp = ltile + (size_t)(first + bk) * 16 + bt;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        int s = 0;
        for (; s + 1 < steps; s += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[(size_t)s * 64], bp[(size_t)s * 64], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[(size_t)(s + 1) * 64], bp[(size_t)(s + 1) * 64], acc1, 0, 0, 0);
        }
        if (s < steps) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[(size_t)s * 64], bp[(size_t)s * 64], acc0, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = 16 * blk + 4 * bk + r;
            if (q < n_coefs && t0 + bt < T) {
                const float val = acc0[r] + acc1[r];
                if (layout == ZAFX_LAYOUT_FT) out[((long long)clip * n_coefs + q) * T + t0 + bt] = val;
                else out[((long long)clip * T + t0 + bt) * n_coefs + q] = val;
            }
        }
    }
}

template <int LOG2N>
static hipError_t run_mel(const zafx_plan& pl, const float* x, float* out, int64_t n_clips, int64_t n_samples, int T) {
    constexpr int LOG2E = default_log2e(LOG2N);
    using G = MelCfg<LOG2N, LOG2E>;
    auto kern = k_mel<LOG2N, LOG2E>;
    const int mfcc = pl.kind == ZAFX_MFCC;
    const size_t smem = G::SMEM_BASE + (mfcc ? (size_t)pl.fb.n_blocks * 16 * 16 * sizeof(float) : 0);
    if (smem > (size_t)kMaxLdsBytes) {
        set_error("mfcc: n_filters too large for the LDS log-mel tile at this window_length");
        return hipErrorInvalidValue;
    }
    static size_t attr_set[64] = {};
    if (attr_set[pl.device] < smem) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        attr_set[pl.device] = smem;
    }
    const int tiles = (T + G::FPB - 1) / G::FPB;
    const long long blocks = (long long)tiles * n_clips;
    if (blocks <= 0) return hipSuccess;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(G::NT), smem, pl.stream, x, pl.d_window, pl.d_tw_pass, pl.d_tw_aux, pl.fb.d_pack,
                       pl.fb.d_meta, pl.fb.n_blocks, pl.dct.d_pack, pl.dct.d_meta, pl.dct.n_blocks, out, (long long)n_samples, pl.H, T, tiles,
                       pl.prm.n_filters, pl.prm.n_coefs, mfcc, pl.layout);
    return hipGetLastError();
}

const char* mel_kernel_name() { return "k_mel"; }

hipError_t launch_mel(const zafx_plan& pl, const float* x, float* out, int64_t n_clips, int64_t n_samples, int T) {
    switch (pl.log2nf) {
        case 5: return run_mel<5>(pl, x, out, n_clips, n_samples, T);
        case 9: return run_mel<9>(pl, x, out, n_clips, n_samples, T);
        case 10: return run_mel<10>(pl, x, out, n_clips, n_samples, T);
    }
    set_error("mel/mfcc: unsupported window_length");
    return hipErrorInvalidValue;
}

}  // namespace zafx
