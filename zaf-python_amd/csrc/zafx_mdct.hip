// zafx_mdct.hip -- batched MDCT / IMDCT kernels for gfx950 (MI355X).
//
// The reference computes each MDCT frame with a W-point complex FFT between two
// twiddle passes (zaf.py:1047-1073) and each IMDCT frame with a zero-padded 2F-point
// FFT (zaf.py:1138-1169).  Both equal the direct (unnormalised) MDCT
//     X[k] = sum_n x[n] w[n] cos(2 pi / W (n + 1/2 + W/4)(k + 1/2)),  k < W/2
// and its transpose scaled by 2/F (checked against the reference to 1e-13 in
// tests/test_oracle_golden.py via the oracle).  The kernels use the W/4-point
// algorithm instead -- 4x fewer butterflies, so the path stays HBM-bound:
//     fold   v = (-c_r - d, a - b_r)                (W -> W/2 reals, TDAC symmetries)
//     pack   c[m] = (v[2m] + i v[M-1-2m]) g_m,      g_m = exp(-i pi (8m+1) / (8M)),  M = W/2
//     FFT    Y = FFT_{M/2}(c)
//     post   y_k = Y[k] g_k ;  X[2k] = Re y_k ;  X[M-1-2k] = -Im y_k        (DCT-IV)
// The IMDCT runs the same DCT-IV on the coefficients and unfolds
//     frame = (2/M) w * (u2, -u2_r, -u1_r, -u1),   u = DCT-IV(X) = (u1, u2)
// followed by the 2-frame TDAC overlap-add (zaf.py:1172-1179) and trim (:1182).
#include <algorithm>

#include "zafx_fft.hpp"
#include "zafx_internal.hpp"

namespace zafx {

template <int LOG2NF, int LOG2E, int FPB>
struct MdctCfg {
    using C = FftCfg<LOG2NF, LOG2E>;
    static constexpr int NF = C::N;       // complex FFT length = W/4
    static constexpr int M = 2 * NF;      // coefficients per frame = W/2 = hop
    static constexpr int W = 4 * NF;
    static constexpr int NT = FPB * C::P;
    static constexpr int STAGE = W + 4;   // floats of windowed-frame staging per slot
    static constexpr size_t SMEM_FWD = (size_t)FPB * C::PITCH * 8 + (size_t)FPB * STAGE * 4 + (size_t)C::TW * 8;
    static constexpr size_t SMEM_INV = (size_t)FPB * C::PITCH * 8 + (size_t)C::TW * 8;
};

__device__ __forceinline__ int fidx(int f) { return 2 * phys(f >> 1) + (f & 1); }   // float f of a padded complex frame

// ---------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------
template <int LOG2NF, int LOG2E, int FPB, int LAYOUT>
__global__ __launch_bounds__(FPB * fft_threads(LOG2NF, LOG2E)) void k_mdct(
    const float* __restrict__ x, const float* __restrict__ win, const float2* __restrict__ twp,
    const float2* __restrict__ tw8, float* __restrict__ out, long long n_samples, int T, int tiles) {
    using C = FftCfg<LOG2NF, LOG2E>;
    using G = MdctCfg<LOG2NF, LOG2E, FPB>;
    constexpr int NF = G::NF, M = G::M, W = G::W, P = C::P, E = C::E, NT = G::NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* frames = reinterpret_cast<float2*>(smem_raw);
    float2* tw_l = frames + FPB * C::PITCH;
    float* stage_all = reinterpret_cast<float*>(tw_l + C::TW);
    const int tid = threadIdx.x;
    for (int i = tid; i < C::TW; i += NT) tw_l[i] = twp[i];

    const int slot = tid / P, p = tid % P;
    const int clip = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const int t0 = tile * FPB;
    const int t = t0 + slot;
    float2* buf = frames + slot * C::PITCH;
    float* u = stage_all + slot * G::STAGE;

    // ---- stage the windowed frame (coalesced): u[n] = xpad[t M + n] w[n], left pad = M (zaf.py:1036-1064)
    {
        const float* xc = x + (long long)clip * n_samples;
        const long long s0 = (long long)t * M - M;
        for (int n = p; n < W; n += P) {
            const long long s = s0 + n;
            u[n] = (t < T && s >= 0 && s < n_samples) ? xc[s] * win[n] : 0.f;
        }
    }
    __syncthreads();   // staging + twiddle table visible

    // ---- fold + pack + pre-twiddle straight into registers
    float2 v[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const int m = p + i * P;
        float re, im;
        if (2 * m < NF) {   // 2m < M/2
            re = -u[3 * NF - 1 - 2 * m] - u[3 * NF + 2 * m];
            im = u[NF - 1 - 2 * m] - u[NF + 2 * m];
        } else {
            re = u[2 * m - NF] - u[3 * NF - 1 - 2 * m];
            im = -u[NF + 2 * m] - u[5 * NF - 1 - 2 * m];
        }
        v[i] = cmul(make_float2(re, im), tw8[m]);
    }
    fft_frame<LOG2NF, LOG2E>(v, buf, p, tw_l);

    // ---- post-twiddle, de-interleave, store
    if constexpr (LAYOUT == ZAFX_LAYOUT_TF) {
        if (t >= T) return;
        float* o = out + ((long long)clip * T + t) * M;
        for (int f = p; f < M; f += P) {
            const int k = (f & 1) ? (M - 1 - f) >> 1 : f >> 1;
            const float2 yk = cmul(buf[phys(k)], tw8[k]);
            o[f] = (f & 1) ? -yk.y : yk.x;
        }
    } else {
        if constexpr (NT > 64) __syncthreads();
        const int tt = tid % FPB, fq = tid / FPB;
        if (t0 + tt >= T) return;
        const float2* fb = frames + tt * C::PITCH;
        float* o = out + (long long)clip * M * T + (t0 + tt);
        for (int f = fq; f < M; f += P) {
            const int k = (f & 1) ? (M - 1 - f) >> 1 : f >> 1;
            const float2 yk = cmul(fb[phys(k)], tw8[k]);
            o[(long long)f * T] = (f & 1) ? -yk.y : yk.x;
        }
    }
}

// ---------------------------------------------------------------------------------
// forward, reference (time-minor) layout, persistent form
// ---------------------------------------------------------------------------------
// The (W/2, T) float32 output wants 32 frames per tile (32 x 4 B = one 128-B line per row; 64-B
// runs cost 40 % of the write bandwidth on MI355X).  One persistent 16-wave workgroup per CU:
// tables (pass twiddles, g_m, the sign-folded window) staged in LDS once; each wave folds and
// transforms 2 frames per tile straight from global memory (no staging copy); the store phase
// writes frame PAIRS as 8-byte stores so that one instruction covers 4 rows x 128 B.
constexpr int kMdctTile = 32;

template <int LOG2NF, int LOG2E>
struct MdctPCfg {
    using C = FftCfg<LOG2NF, LOG2E>;
    static_assert(C::P == 64, "one wavefront per frame");
    static constexpr int NF = C::N;
    static constexpr int NSLOT = 8;   // 8 fat waves (up to 256 VGPRs): the persistent loop does not spill
    static constexpr size_t SMEM = (size_t)(kMdctTile * C::PITCH + C::TW + NF) * 8 + (size_t)NF * 16;
};

template <int LOG2NF, int LOG2E>
__global__ __launch_bounds__(512) void k_mdct_ft32(
    const float* __restrict__ x, const float4* __restrict__ wfold, const float2* __restrict__ twp,
    const float2* __restrict__ tw8, float* __restrict__ out, long long n_samples, int T, int tiles, int total_tiles) {
    using C = FftCfg<LOG2NF, LOG2E>;
    constexpr int NF = C::N, M = 2 * NF, W = 4 * NF, P = C::P, E = C::E, FPB = kMdctTile, NSLOT = 8, NT = NSLOT * P;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* frames = reinterpret_cast<float2*>(smem_raw);
    float2* tw_l = frames + FPB * C::PITCH;
    float2* g_l = tw_l + C::TW;
    float4* wf_l = reinterpret_cast<float4*>(g_l + NF);
    const int tid = threadIdx.x;
    for (int i = tid; i < C::TW; i += NT) tw_l[i] = twp[i];
    for (int i = tid; i < NF; i += NT) { g_l[i] = tw8[i]; wf_l[i] = wfold[i]; }
    __syncthreads();
    const int slot = tid / P, p = tid % P;
    const int tp = tid % 16, fq = tid / 16;
    const bool pair_ok = (T % 2 == 0) && (reinterpret_cast<uintptr_t>(out) % 8 == 0);

    for (int tl = blockIdx.x; tl < total_tiles; tl += gridDim.x) {
        const int clip = tl / tiles, tile = tl % tiles;
        const int t0 = tile * FPB;
        const float* xc = x + (long long)clip * n_samples;
#pragma unroll 1
        for (int f = 0; f < FPB / NSLOT; ++f) {
            const int j = slot * (FPB / NSLOT) + f;
            const int t = t0 + j;
            const long long s0 = (long long)t * M - M;   // left pad = M (zaf.py:1036-1041)
            const bool live = t < T;
            const bool interior = live && s0 >= 0 && s0 + W <= n_samples;
            float2 v[E];
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const int m = p + i * P;
                int a, b, c, d;
                if (i < E / 2) { a = 3 * NF - 1 - 2 * m; b = 3 * NF + 2 * m; c = NF - 1 - 2 * m; d = NF + 2 * m; }
                else { a = 2 * m - NF; b = 3 * NF - 1 - 2 * m; c = NF + 2 * m; d = 5 * NF - 1 - 2 * m; }
                float xa, xb, xcv, xd;
                if (interior) {
                    xa = xc[s0 + a]; xb = xc[s0 + b]; xcv = xc[s0 + c]; xd = xc[s0 + d];
                } else {
                    const long long sa = s0 + a, sb = s0 + b, sc = s0 + c, sd = s0 + d;
                    xa = (live && sa >= 0 && sa < n_samples) ? xc[sa] : 0.f;
                    xb = (live && sb >= 0 && sb < n_samples) ? xc[sb] : 0.f;
                    xcv = (live && sc >= 0 && sc < n_samples) ? xc[sc] : 0.f;
                    xd = (live && sd >= 0 && sd < n_samples) ? xc[sd] : 0.f;
                }
                const float4 wf = wf_l[m];
                v[i] = cmul(make_float2(xa * wf.x + xb * wf.y, xcv * wf.z + xd * wf.w), g_l[m]);
            }
            fft_frame<LOG2NF, LOG2E>(v, frames + j * C::PITCH, p, tw_l);
        }
        __syncthreads();
        const int ta = t0 + 2 * tp;
        if (ta < T) {
            const float2* ba = frames + (2 * tp) * C::PITCH;
            const float2* bb = ba + C::PITCH;
            float* o = out + (long long)clip * M * T + ta;
            const bool two = ta + 1 < T;
            for (int f = fq; f < M; f += NT / 16) {
                const int k = (f & 1) ? (M - 1 - f) >> 1 : f >> 1;
                const float2 g = g_l[k];
                const float2 ya = cmul(ba[phys(k)], g), yb = cmul(bb[phys(k)], g);
                const float va = (f & 1) ? -ya.y : ya.x, vb = (f & 1) ? -yb.y : yb.x;
                float* dst = o + (long long)f * T;
                if (pair_ok && two) *reinterpret_cast<float2*>(dst) = make_float2(va, vb);
                else {
                    dst[0] = va;
                    if (two) dst[1] = vb;
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------
// inverse
// ---------------------------------------------------------------------------------
template <int LOG2NF, int LOG2E, int FPB, int NSLOT, int LAYOUT>
__global__ __launch_bounds__(NSLOT * fft_threads(LOG2NF, LOG2E)) void k_imdct(
    const float* __restrict__ coefs, const float* __restrict__ win, const float2* __restrict__ twp,
    const float2* __restrict__ tw8g, float* __restrict__ y, int T, long long out_len, int tiles, int total_tiles) {
    using C = FftCfg<LOG2NF, LOG2E>;
    constexpr int NF = C::N, M = 2 * NF, P = C::P, E = C::E, NT = NSLOT * P;
    constexpr int OWNED = FPB - 1;   // one halo frame: every output sample sums exactly 2 frames
    static_assert(NT % FPB == 0 || LAYOUT == ZAFX_LAYOUT_TF, "time-minor gather needs NT to be a multiple of FPB");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* frames = reinterpret_cast<float2*>(smem_raw);
    float2* tw_l = frames + FPB * C::PITCH;
    float2* tw8 = tw_l + C::TW;                         // g_m, NF entries
    float* win_l = reinterpret_cast<float*>(tw8 + NF);  // window, 4 NF floats
    const int tid = threadIdx.x;
    for (int i = tid; i < C::TW; i += NT) tw_l[i] = twp[i];
    for (int i = tid; i < NF; i += NT) tw8[i] = tw8g[i];
    for (int i = tid; i < 4 * NF; i += NT) win_l[i] = win[i];
    __syncthreads();

    // persistent: one workgroup per CU loops over the tiles, the tables above are staged once
    for (int tl = blockIdx.x; tl < total_tiles; tl += gridDim.x) {
    const int clip = tl / tiles, tile = tl % tiles;
    const int t_first = tile * OWNED - 1;

    // ---- phase A: c[m] = (X[2m] + i X[M-1-2m]) g_m  -> LDS (natural order)
    if constexpr (LAYOUT == ZAFX_LAYOUT_TF) {
        const int mq = tid % P;
        for (int fs = tid / P; fs < FPB; fs += NSLOT) {
            const int t = t_first + fs;
            if (t < 0 || t >= T) continue;
            float2* fb = frames + fs * C::PITCH;
            const float* cp = coefs + ((long long)clip * T + t) * M;
            for (int m = mq; m < NF; m += P) fb[phys(m)] = cmul(make_float2(cp[2 * m], cp[M - 1 - 2 * m]), tw8[m]);
        }
    } else {
        const int fs = tid % FPB, mq = tid / FPB;   // lanes run along t: FPB * 4 B contiguous per row
        const int t = t_first + fs;
        if (t >= 0 && t < T) {
            float2* fb = frames + fs * C::PITCH;
            const float* cp = coefs + (long long)clip * M * T + t;
            for (int m = mq; m < NF; m += NT / FPB) {
                const float re = cp[(long long)(2 * m) * T];
                const float im = cp[(long long)(M - 1 - 2 * m) * T];
                fb[phys(m)] = cmul(make_float2(re, im), tw8[m]);
            }
        }
    }
    __syncthreads();

    // ---- phase B: FFT, then DCT-IV post-twiddle written in place as M reals per frame
    {
        const int p = tid % P;
#pragma unroll 1
        for (int slot = tid / P; slot < FPB; slot += NSLOT) {
            float2* buf = frames + slot * C::PITCH;
            float2 v[E];
            regs_read<LOG2NF, LOG2E>(v, buf, p);
            frame_sync<P>();
            fft_frame<LOG2NF, LOG2E>(v, buf, p, tw_l);
            // pair (k, NF-1-k): u[2k] = Re y_k, u[2k+1] = -Im y_kk, u[2kk] = Re y_kk, u[2kk+1] = -Im y_k
            for (int k = p; k < NF / 2; k += P) {
                const int kk = NF - 1 - k;
                const float2 a = cmul(buf[phys(k)], tw8[k]);
                const float2 b = cmul(buf[phys(kk)], tw8[kk]);
                buf[phys(k)] = make_float2(a.x, -b.y);
                buf[phys(kk)] = make_float2(b.x, -a.y);
            }
        }
    }
    __syncthreads();

    // ---- phase C: unfold + window + TDAC overlap-add of the 2 covering frames, trim (zaf.py:1166-1182)
    {
        const float* fl = reinterpret_cast<const float*>(frames);
        const float gain = 2.f / (float)M;
        const int t_end = min((tile + 1) * OWNED, T);
        const long long s_begin = (long long)tile * OWNED * M;
        const long long s_end = (tile == tiles - 1) ? (long long)(T + 1) * M : (long long)t_end * M;
        float* yc = y + (long long)clip * out_len;
        for (long long s = s_begin + tid; s < s_end; s += NT) {
            const long long o = s - M;
            if (o < 0 || o >= out_len) continue;
            const int j1 = (int)(s / M);
            const int n1 = (int)(s - (long long)j1 * M);   // in [0, M)
            float acc = 0.f;
            if (j1 >= 1) {   // older frame first (ascending j, as the reference's loop)
                const int n0 = n1 + M;   // in [M, 2M)
                const float* fr = fl + (size_t)(j1 - 1 - t_first) * (2 * C::PITCH);
                const float uu = (n0 < 3 * NF) ? -fr[fidx(3 * NF - 1 - n0)] : -fr[fidx(n0 - 3 * NF)];
                acc += uu * win_l[n0];
            }
            if (j1 < T) {
                const float* fr = fl + (size_t)(j1 - t_first) * (2 * C::PITCH);
                const float uu = (n1 < NF) ? fr[fidx(NF + n1)] : -fr[fidx(3 * NF - 1 - n1)];
                acc += uu * win_l[n1];
            }
            yc[o] = acc * gain;
        }
    }
    __syncthreads();   // the next tile overwrites the frame buffers
    }
}

// ---------------------------------------------------------------------------------
// launch plumbing
// ---------------------------------------------------------------------------------
constexpr int mdct_fpb(int log2nf, int layout) {
    const int p = (1 << log2nf) >> default_log2e(log2nf);
    int cap = 1024 / p;
    const int pitch = (1 << log2nf) + ((1 << log2nf) >> 4) + 1;
    const int per_frame = pitch * 8 + ((4 << log2nf) + 4) * 4;
    int lds_cap = (kMaxLdsBytes - twiddle_total(log2nf, default_log2e(log2nf)) * 8) / per_frame;
    int f = layout == ZAFX_LAYOUT_FT ? 16 : 4;
    if (f > cap) f = cap;
    if (f > lds_cap) f = lds_cap;
    int r = 1;
    while (r * 2 <= f) r *= 2;
    return r;
}

constexpr bool mdct_use_persistent(int log2nf, int layout) {
    return layout == ZAFX_LAYOUT_FT && log2nf >= 7 && log2nf <= 9;
}

template <int LOG2NF>
static hipError_t run_mdct_p(const zafx_plan& pl, const float* x, float* out, int64_t n_clips, int64_t n_samples, int T) {
    constexpr int LOG2E = default_log2e(LOG2NF);
    using G = MdctPCfg<LOG2NF, LOG2E>;
    auto kern = k_mdct_ft32<LOG2NF, LOG2E>;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, G::SMEM); e != hipSuccess) return e;
    const int tiles = (T + kMdctTile - 1) / kMdctTile;
    const long long total = (long long)tiles * n_clips;
    if (total <= 0) return hipSuccess;
    const int per_cu = (int)std::min<size_t>(2, (size_t)kMaxLdsBytes / G::SMEM);
    const long long grid = std::min<long long>(total, (long long)pl.n_cus * std::max(per_cu, 1));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(G::NSLOT * 64), G::SMEM, pl.stream, x, pl.d_wfold, pl.d_tw_pass, pl.d_tw_aux, out,
                       (long long)n_samples, T, tiles, (int)total);
    return hipGetLastError();
}

template <int LOG2NF, int LAYOUT>
static hipError_t run_mdct(const zafx_plan& pl, const float* x, float* out, int64_t n_clips, int64_t n_samples, int T) {
    if constexpr (mdct_use_persistent(LOG2NF, LAYOUT)) {
        return run_mdct_p<LOG2NF>(pl, x, out, n_clips, n_samples, T);
    } else {
        constexpr int LOG2E = default_log2e(LOG2NF);
        constexpr int FPB = mdct_fpb(LOG2NF, LAYOUT);
        using G = MdctCfg<LOG2NF, LOG2E, FPB>;
        auto kern = k_mdct<LOG2NF, LOG2E, FPB, LAYOUT>;
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, G::SMEM_FWD); e != hipSuccess) return e;
        const int tiles = (T + FPB - 1) / FPB;
        const long long blocks = (long long)tiles * n_clips;
        if (blocks <= 0) return hipSuccess;
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(G::NT), G::SMEM_FWD, pl.stream, x, pl.d_window, pl.d_tw_pass, pl.d_tw_aux, out,
                           (long long)n_samples, T, tiles);
        return hipGetLastError();
    }
}

// IMDCT tile geometry: FPB frames resident (FPB - 1 owned + 1 halo), NSLOT of them transformed at a time.
// The time-minor layout wants FPB = 32 (32 frames x 4 B = one 128-B line per gathered row).
constexpr int imdct_fpb(int log2nf, int layout) {
    const int pitch = (1 << log2nf) + ((1 << log2nf) >> 4) + 1;
    const int lds_cap = (kMaxLdsBytes - twiddle_total(log2nf, default_log2e(log2nf)) * 8 - (24 << log2nf)) / (pitch * 8);
    int f = layout == ZAFX_LAYOUT_FT ? 32 : 8;
    while (f > lds_cap) f /= 2;
    return f < 2 ? 2 : f;
}
constexpr int imdct_nslot(int log2nf, int fpb) {
    const int p = (1 << log2nf) >> default_log2e(log2nf);
    int s = 1024 / p;   // (8 fat waves measured slower here: 2.02 vs 1.53 ms)
    if (s > fpb) s = fpb;
    // NT = s * p must be a multiple of fpb for the time-minor gather
    while ((s * p) % fpb != 0 && s < 1024 / p) ++s;
    return s;
}

template <int LOG2NF, int LAYOUT>
static hipError_t run_imdct(const zafx_plan& pl, const float* coefs, float* y, int64_t n_clips, int T, int64_t out_len) {
    constexpr int LOG2E = default_log2e(LOG2NF);
    constexpr int FPB = imdct_fpb(LOG2NF, LAYOUT);
    constexpr int NSLOT = imdct_nslot(LOG2NF, FPB);
    using C = FftCfg<LOG2NF, LOG2E>;
    constexpr size_t SMEM = (size_t)(FPB * C::PITCH + C::TW + C::N) * 8 + (size_t)C::N * 16;   // frames + twiddles + g_m + window
    static_assert(SMEM <= (size_t)kMaxLdsBytes, "IMDCT tile does not fit LDS");
    auto kern = k_imdct<LOG2NF, LOG2E, FPB, NSLOT, LAYOUT>;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, SMEM); e != hipSuccess) return e;
    constexpr int OWNED = FPB - 1;
    const int tiles = (T + OWNED - 1) / OWNED;
    const long long total = (long long)tiles * n_clips;
    if (total <= 0 || out_len <= 0) return hipSuccess;
    const int per_cu = (int)std::min<size_t>(2, (size_t)kMaxLdsBytes / SMEM);
    const long long grid = std::min<long long>(total, (long long)pl.n_cus * std::max(per_cu, 1));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NSLOT * C::P), SMEM, pl.stream, coefs, pl.d_window, pl.d_tw_pass, pl.d_tw_aux, y, T,
                       (long long)out_len, tiles, (int)total);
    return hipGetLastError();
}

#define ZAFX_MDCT_SIZES(X) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11)

bool mdct_supported(int log2nf) { return log2nf >= 4 && log2nf <= 11; }
int mdct_frames_per_block(int log2nf, int layout) { return mdct_fpb(log2nf, layout); }
const char* mdct_kernel_name(int log2nf, int layout) { return mdct_use_persistent(log2nf, layout) ? "k_mdct_ft32" : "k_mdct"; }
const char* imdct_kernel_name() { return "k_imdct"; }

hipError_t launch_mdct(const zafx_plan& pl, const float* x, float* out, int64_t n_clips, int64_t n_samples, int T) {
    switch (pl.log2nf) {
#define X(L)                                                                                        \
    case L:                                                                                         \
        return pl.layout == ZAFX_LAYOUT_FT ? run_mdct<L, ZAFX_LAYOUT_FT>(pl, x, out, n_clips, n_samples, T) \
                                           : run_mdct<L, ZAFX_LAYOUT_TF>(pl, x, out, n_clips, n_samples, T);
        ZAFX_MDCT_SIZES(X)
#undef X
    }
    return hipErrorInvalidValue;
}

hipError_t launch_imdct(const zafx_plan& pl, const float* coefs, float* y, int64_t n_clips, int T, int64_t out_len) {
    switch (pl.log2nf) {
#define X(L)                                                                                           \
    case L:                                                                                            \
        return pl.layout == ZAFX_LAYOUT_FT ? run_imdct<L, ZAFX_LAYOUT_FT>(pl, coefs, y, n_clips, T, out_len) \
                                           : run_imdct<L, ZAFX_LAYOUT_TF>(pl, coefs, y, n_clips, T, out_len);
        ZAFX_MDCT_SIZES(X)
#undef X
    }
    return hipErrorInvalidValue;
}

}  // namespace zafx
