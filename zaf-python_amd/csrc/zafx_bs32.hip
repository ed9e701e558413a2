// zafx_bs32.hip -- float32 kernels for windows that are NOT a power of two (np.fft takes any length, so does the
// reference: zaf.py:139, :223, :1068, :1159): STFT, the ISTFT's frames, MDCT and the IMDCT's frames as Bluestein
// convolutions on the tuned FFT core (zafx_fft.hpp).
//
//   n k = (n^2 + k^2 - (k - n)^2) / 2,   c[n] = exp(-i pi n^2 / W)   =>   X[k] = c[k] sum_n (x[n] c[n]) conj(c)[k - n]:
// a W-point DFT is a convolution of length M = 2^ceil(log2(2 W - 1)) <= 16384 for W <= 8192 (bs32_supported).  The host provides c (exact
// angle reduction in integers) and Bhat = FFT_M(conj(c) wrapped), both evaluated in long double; one workgroup of M / 16
// threads owns a frame: a = x w c -> FFT_M (radix-16 passes, frame in LDS) -> conj(. Bhat) -> FFT_M -> c conj(.) / M.
// The float64 forms of zafx_f64.hip (one radix-2 pass per barrier) ran these windows at 3.4 Gsamples/s (W = 1764, hop 441);
// they remain the ZAFX_PRECISION_F64 path and the route of windows below 33 samples.
#include <algorithm>

#include "zafx_fft.hpp"
#include "zafx_internal.hpp"

namespace zafx {

namespace {

template <int LOG2M>
struct BsCfg {
    static constexpr int LOG2E = default_log2e(LOG2M);
    using C = FftCfg<LOG2M, LOG2E>;
    static constexpr int M = C::N, P = C::P, E = C::E;
    static constexpr size_t SMEM = (size_t)C::PITCH * sizeof(float2);
};

// `buf` holds a[n] = x[n] c[n] (n < W) and zeros (W <= n < M) in the padded natural order of the FFT core.  On return it
// holds conj(y) M with y the convolution: X[k] = c[k] conj(buf[k]) / M for k < W (bs_out).
template <int LOG2M>
__device__ __forceinline__ void bluestein32(float2* buf, int p, const float2* __restrict__ tw, const float2* __restrict__ bhat) {
    using B = BsCfg<LOG2M>;
    using C = typename B::C;
    float2 v[B::E];
    regs_read<LOG2M, B::LOG2E>(v, buf, p);
    frame_sync<B::P>();
    fft_frame<LOG2M, B::LOG2E>(v, buf, p, tw);
    regs_read<LOG2M, B::LOG2E>(v, buf, p);
#pragma unroll
    for (int i = 0; i < B::E; ++i) v[i] = cconj(cmul(v[i], bhat[p + i * B::P]));   // conj: the next forward transform inverts
    frame_sync<B::P>();
    fft_frame<LOG2M, B::LOG2E>(v, buf, p, tw);
    (void)sizeof(C);
}
template <int LOG2M>
__device__ __forceinline__ float2 bs_out(const float2* buf, int k, const float2* __restrict__ chirp) {
    using C = typename BsCfg<LOG2M>::C;
    const float inv = 1.f / (float)C::N;
    const float2 y = cconj(buf[phys_t<C::PS>(k)]);
    return cmul(chirp[k], make_float2(y.x * inv, y.y * inv));
}
template <int LOG2M>
__device__ __forceinline__ void bs_zero_tail(float2* buf, int p, int W) {
    using B = BsCfg<LOG2M>;
    for (int n = W + p; n < B::M; n += B::P) buf[phys_t<B::C::PS>(n)] = make_float2(0.f, 0.f);
}

// frame id of a workgroup: consecutive frames of a clip to the workgroups of one XCD (xcd_order), so that the 8-byte
// stores of neighbouring frames to a row of the reference layout meet in one L2
__device__ __forceinline__ long long frame_of_block(long long total) { return xcd_order((int)blockIdx.x, (int)total); }

// zaf.py:112-139 for one frame per workgroup
template <int LOG2M>
__global__ __launch_bounds__(BsCfg<LOG2M>::P) void k_stft_bs32(
    const float* __restrict__ x, const float* __restrict__ win, const float2* __restrict__ tw, const float2* __restrict__ chirp,
    const float2* __restrict__ bhat, float2* __restrict__ out, long long n_samples, int hop, int T, int TP, int W, int layout, int spec,
    long long total) {
    using B = BsCfg<LOG2M>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* buf = reinterpret_cast<float2*>(smem_raw);
    const int p = threadIdx.x;
    const long long g = frame_of_block(total);
    const long long clip = g / T;
    const int t = (int)(g - clip * T);
    const float* xc = x + clip * n_samples;
    const long long s0 = (long long)t * hop - W / 2;   // floor(W/2) samples of left padding (zaf.py:99)
    for (int n = p; n < W; n += B::P) {
        const long long s = s0 + n;
        const float v = (s >= 0 && s < n_samples) ? xc[s] * win[n] : 0.f;
        const float2 c = chirp[n];
        buf[phys_t<B::C::PS>(n)] = make_float2(v * c.x, v * c.y);
    }
    bs_zero_tail<LOG2M>(buf, p, W);
    frame_sync<B::P>();
    bluestein32<LOG2M>(buf, p, tw, bhat);
    const int rows = spec ? W / 2 + 1 : W;
    const long long stride = layout == ZAFX_LAYOUT_FT ? TP : 1;
    const long long base = layout == ZAFX_LAYOUT_FT ? clip * rows * TP + t : (clip * T + t) * rows;
    for (int k = p; k < rows; k += B::P) {
        float2 v = bs_out<LOG2M>(buf, k, chirp);
        if (k == 0 || 2 * k == W) v.y = 0.f;   // real input: DC and Nyquist are real (np.fft returns exact zeros there)
        if (spec >= ZAFX_SPECTRUM_MAGNITUDE) {
            const float pw = v.x * v.x + v.y * v.y;
            reinterpret_cast<float*>(out)[base + k * stride] = spec == ZAFX_SPECTRUM_MAGNITUDE ? __builtin_amdgcn_sqrtf(pw) : pw;
        } else {
            out[base + k * stride] = v;
        }
    }
}

// real(ifft(X)) of one frame per workgroup for any W: ifft(X) = conj(fft(conj(X))) / W  (zaf.py:223)
template <int LOG2M>
__global__ __launch_bounds__(BsCfg<LOG2M>::P) void k_ifft_frames_bs32(
    const float2* __restrict__ spec, const float2* __restrict__ tw, const float2* __restrict__ chirp, const float2* __restrict__ bhat,
    float* __restrict__ frames, int T, int TP, int W, int layout, int one, long long total) {
    using B = BsCfg<LOG2M>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* buf = reinterpret_cast<float2*>(smem_raw);
    const int p = threadIdx.x, rows = one ? W / 2 + 1 : W;
    const long long g = frame_of_block(total);
    const long long clip = g / T;
    const int t = (int)(g - clip * T);
    const long long stride = layout == ZAFX_LAYOUT_FT ? TP : 1;
    const float2* sp = layout == ZAFX_LAYOUT_FT ? spec + clip * rows * TP + t : spec + (clip * T + t) * rows;
    for (int k = p; k < W; k += B::P) {
        // one-sided input: X[W-k] = conj X[k]; conj(X) goes in
        const float2 xin = (one && k > W / 2) ? sp[(long long)(W - k) * stride] : cconj(sp[(long long)k * stride]);
        buf[phys_t<B::C::PS>(k)] = cmul(xin, chirp[k]);
    }
    bs_zero_tail<LOG2M>(buf, p, W);
    frame_sync<B::P>();
    bluestein32<LOG2M>(buf, p, tw, bhat);
    float* fr = frames + g * W;
    const float inv = 1.f / (float)W;
    for (int n = p; n < W; n += B::P) fr[n] = bs_out<LOG2M>(buf, n, chirp).x * inv;   // Re(conj(.)) = Re(.)
}

// zaf.dct / zaf.dst, types I-IV, of ANY length (zaf.py:760-839, :900-981: the reference's np.fft.fft takes any): the transform as a chirp-z
// sum y[k] = Re(Q[k] sum_n (x[n] P[n]) conj(c)[k - n]) (tables P | Q and the chirp's transform from zafx_plan_create, which derives them for
// the eight transforms at once).  One vector per frame of the FFT core, rows walked by a persistent grid; 8 bytes per sample.
template <int LOG2M>
__global__ __launch_bounds__(BsCfg<LOG2M>::P) void k_dct_bs32(const float* __restrict__ x, float* __restrict__ y, const float2* __restrict__ tw,
                                                              const float2* __restrict__ pq, const float2* __restrict__ bhat, int N, long long n_rows) {
    using B = BsCfg<LOG2M>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* buf = reinterpret_cast<float2*>(smem_raw);
    for (long long row = blockIdx.x; row < n_rows; row += gridDim.x) {
        int p = threadIdx.x;
        asm volatile("" : "+v"(p));   // (opaque per row: the thread's twiddles and chirp-transform values are re-read, not hoisted out of the loop and spilled)
        const float* xr = x + row * N;
        for (int n = p; n < N; n += B::P) {
            const float v = xr[n];
            const float2 c = pq[n];
            buf[phys_t<B::C::PS>(n)] = make_float2(v * c.x, v * c.y);
        }
        bs_zero_tail<LOG2M>(buf, p, N);
        frame_sync<B::P>();
        bluestein32<LOG2M>(buf, p, tw, bhat);   // conj(convolution) M
        float* yr = y + row * N;
        for (int k = p; k < N; k += B::P) {
            const float2 z = buf[phys_t<B::C::PS>(k)], q = pq[N + k];   // (Q carries 1 / M and the orthonormal scale)
            yr[k] = q.x * z.x + q.y * z.y;                              // Re(Q conj(z))
        }
        frame_sync<B::P>();   // the frame is free for the next row
    }
}

// overlap-add in ascending frame order (zaf.py:226-233 / :1172-1179), trim (:236-238 / :1182), gain (:241)
__global__ __launch_bounds__(256) void k_ola_f32(const float* __restrict__ frames, float* __restrict__ y, int T, int W, int hop, long long out_len,
                                                 long long total, float scale) {
    for (long long i = (long long)blockIdx.x * (int)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * (int)blockDim.x) {
        const long long clip = i / out_len, o = i - clip * out_len;
        const long long s = o + (W - hop);
        const long long j_hi = std::min<long long>(T - 1, s / hop);
        const long long j_lo = s >= W ? (s - W) / hop + 1 : 0;
        float acc = 0.f;
        for (long long j = j_lo; j <= j_hi; ++j) acc += frames[(clip * T + j) * W + (s - j * hop)];
        y[i] = acc * scale;
    }
}

// MDCT of any even window length, the reference's own formulation (zaf.py:1047-1073): W-point FFT of x w pre, first W/2
// outputs times post, real part;  pre[n] = exp(-i pi n / W),  post[k] = exp(-i pi (W/2 + 1)(k + 1/2) / W)  (host tables `aux`:
// pre[W] | post[W/2])
template <int LOG2M>
__global__ __launch_bounds__(BsCfg<LOG2M>::P) void k_mdct_bs32(
    const float* __restrict__ x, const float* __restrict__ win, const float2* __restrict__ tw, const float2* __restrict__ chirp,
    const float2* __restrict__ bhat, const float2* __restrict__ aux, float* __restrict__ out, long long n_samples, int T, int TP, int W,
    int layout, long long total) {
    using B = BsCfg<LOG2M>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* buf = reinterpret_cast<float2*>(smem_raw);
    const int p = threadIdx.x, F = W / 2;
    const long long g = frame_of_block(total);
    const long long clip = g / T;
    const int t = (int)(g - clip * T);
    const float* xc = x + clip * n_samples;
    const long long s0 = (long long)t * F - F;   // left pad = W/2 (zaf.py:1036-1041)
    for (int n = p; n < W; n += B::P) {
        const long long s = s0 + n;
        const float v = (s >= 0 && s < n_samples) ? xc[s] * win[n] : 0.f;
        const float2 a = cmul(aux[n], chirp[n]);
        buf[phys_t<B::C::PS>(n)] = make_float2(v * a.x, v * a.y);
    }
    bs_zero_tail<LOG2M>(buf, p, W);
    frame_sync<B::P>();
    bluestein32<LOG2M>(buf, p, tw, bhat);
    const long long stride = layout == ZAFX_LAYOUT_FT ? TP : 1;
    const long long base = layout == ZAFX_LAYOUT_FT ? clip * F * TP + t : (clip * T + t) * F;
    for (int k = p; k < F; k += B::P) {
        const float2 X = bs_out<LOG2M>(buf, k, chirp), post = aux[W + k];
        out[base + k * stride] = X.x * post.x - X.y * post.y;
    }
}

// IMDCT frames of any even window length (zaf.py:1138-1169): W-point FFT of X pre zero-padded from F to W, times post, real
// part, times 2 w;  pre[k] = exp(-i pi (F + 1) k / W),  post[n] = exp(-i pi (n + 1/2 + F/2) / W) / F  (host tables `aux`:
// pre[F] | post[W], the 1/F left to the kernel)
template <int LOG2M>
__global__ __launch_bounds__(BsCfg<LOG2M>::P) void k_imdct_frames_bs32(
    const float* __restrict__ coefs, const float* __restrict__ win, const float2* __restrict__ tw, const float2* __restrict__ chirp,
    const float2* __restrict__ bhat, const float2* __restrict__ aux, float* __restrict__ frames, int T, int TP, int W, int layout,
    long long total) {
    using B = BsCfg<LOG2M>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* buf = reinterpret_cast<float2*>(smem_raw);
    const int p = threadIdx.x, F = W / 2;
    const long long g = frame_of_block(total);
    const long long clip = g / T;
    const int t = (int)(g - clip * T);
    const long long stride = layout == ZAFX_LAYOUT_FT ? TP : 1;
    const float* cp = layout == ZAFX_LAYOUT_FT ? coefs + clip * F * TP + t : coefs + (clip * T + t) * F;
    for (int k = p; k < F; k += B::P) {
        const float v = cp[(long long)k * stride];
        const float2 a = cmul(aux[k], chirp[k]);
        buf[phys_t<B::C::PS>(k)] = make_float2(v * a.x, v * a.y);
    }
    bs_zero_tail<LOG2M>(buf, p, F);
    frame_sync<B::P>();
    bluestein32<LOG2M>(buf, p, tw, bhat);
    float* fr = frames + g * W;
    const float scale = 2.f / (float)F;
    for (int n = p; n < W; n += B::P) {
        const float2 Y = bs_out<LOG2M>(buf, n, chirp), post = aux[F + n];
        fr[n] = scale * (Y.x * post.x - Y.y * post.y) * win[n];
    }
}

template <class F>
hipError_t by_log2m(int log2m, F&& f) {
    switch (log2m) {
        case 7: return f(std::integral_constant<int, 7>{});
        case 8: return f(std::integral_constant<int, 8>{});
        case 9: return f(std::integral_constant<int, 9>{});
        case 10: return f(std::integral_constant<int, 10>{});
        case 11: return f(std::integral_constant<int, 11>{});
        case 12: return f(std::integral_constant<int, 12>{});
        case 13: return f(std::integral_constant<int, 13>{});
        case 14: return f(std::integral_constant<int, 14>{});
    }
    set_error("bluestein: unsupported convolution length");
    return hipErrorInvalidValue;
}

hipError_t launch_ola(zafx_plan& pl, float* y, int64_t n_clips, int T, int hop, int64_t out_len, float scale) {
    const long long total = (long long)n_clips * out_len;
    const long long grid = std::min<long long>((total + 255) / 256, (long long)pl.n_cus * 32);
    hipLaunchKernelGGL(k_ola_f32, dim3((unsigned)grid), dim3(256), 0, pl.stream, reinterpret_cast<const float*>(pl.d_scratch64), y, T, pl.W, hop,
                       (long long)out_len, total, scale);
    return hipGetLastError();
}

}  // namespace

bool bs32_supported(int W) { return W >= 33 && W <= 8192; }   // M = 128 ... 16384 (the frame of 2^14 points is 139 KB of LDS)

hipError_t launch_stft_bs32(const zafx_plan& pl, const float* x, float2* out, int64_t n_clips, int64_t n_samples, int T) {
    const long long total = (long long)n_clips * T;
    if (total <= 0) return hipSuccess;
    return by_log2m(pl.bs_log2m, [&](auto tag) {
        constexpr int L = decltype(tag)::value;
        auto kern = k_stft_bs32<L>;
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, BsCfg<L>::SMEM); e != hipSuccess) return e;
        pl.ran = "k_stft_bs32";
        hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(BsCfg<L>::P), BsCfg<L>::SMEM, pl.stream, x, pl.d_window, pl.d_tw_pass, pl.d_bs_chirp,
                           pl.d_bs_bhat, out, (long long)n_samples, pl.H, T, (int)row_pitch(pl, T), pl.W, pl.layout, pl.prm.spectrum, total);
        return hipGetLastError();
    });
}

// The inverse transforms park the time-domain frames of a chunk of clips in the plan's scratch (scratch_clips_per_chunk), then
// gather-overlap-add them: no limit on ceil(W / hop), unlike the tiled float32 kernels.
hipError_t launch_istft_bs32(zafx_plan& pl, const float2* spec_all, float* y_all, int64_t n_clips_all, int T, int64_t out_len) {
    if ((long long)n_clips_all * T <= 0 || out_len <= 0) return hipSuccess;
    const int64_t chunk = scratch_clips_per_chunk(n_clips_all, T, pl.W, sizeof(float));
    if (hipError_t e = grow_scratch(pl, (size_t)chunk * T * pl.W * sizeof(float)); e != hipSuccess) return e;
    const int one = pl.prm.spectrum == ZAFX_SPECTRUM_ONE_SIDED ? 1 : 0;
    const int64_t rows = one ? pl.W / 2 + 1 : pl.W;
    const int64_t in_per_clip = pl.layout == ZAFX_LAYOUT_FT ? rows * row_pitch(pl, T) : (int64_t)T * rows;
    for (int64_t c0 = 0; c0 < n_clips_all; c0 += chunk) {
        const int64_t n_clips = std::min(chunk, n_clips_all - c0);
        const long long total = (long long)n_clips * T;
        const float2* spec = spec_all + c0 * in_per_clip;
        hipError_t e = by_log2m(pl.bs_log2m, [&](auto tag) {
            constexpr int L = decltype(tag)::value;
            auto kern = k_ifft_frames_bs32<L>;
            if (hipError_t e2 = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, BsCfg<L>::SMEM); e2 != hipSuccess) return e2;
            pl.ran = "k_ifft_frames_bs32";
            hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(BsCfg<L>::P), BsCfg<L>::SMEM, pl.stream, spec, pl.d_tw_pass, pl.d_bs_chirp, pl.d_bs_bhat,
                               reinterpret_cast<float*>(pl.d_scratch64), T, (int)row_pitch(pl, T), pl.W, pl.layout, one, total);
            return hipGetLastError();
        });
        if (e != hipSuccess) return e;
        if (hipError_t e2 = launch_ola(pl, y_all + c0 * out_len, n_clips, T, pl.H, out_len, 1.f / pl.cola_gain); e2 != hipSuccess) return e2;
    }
    return hipSuccess;
}

hipError_t launch_dct_bs32(const zafx_plan& pl, const float* x, float* y, int64_t n_rows) {
    if (n_rows <= 0) return hipSuccess;
    return by_log2m(pl.bs_log2m, [&](auto tag) {
        constexpr int L = decltype(tag)::value;
        auto kern = k_dct_bs32<L>;
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, BsCfg<L>::SMEM); e != hipSuccess) return e;
        size_t per_cu = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(2048 / BsCfg<L>::P, 16), (size_t)kMaxLdsBytes / BsCfg<L>::SMEM));
        int resident = 0;   // (registers may admit fewer workgroups than LDS: a persistent grid beyond what is resident runs a second, part-empty round)
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, kern, BsCfg<L>::P, BsCfg<L>::SMEM) == hipSuccess && resident > 0)
            per_cu = std::min<size_t>(per_cu, (size_t)resident);
        const long long grid = std::min<long long>(n_rows, (long long)pl.n_cus * (long long)per_cu);
        pl.ran = "k_dct_bs32";
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(BsCfg<L>::P), BsCfg<L>::SMEM, pl.stream, x, y, pl.d_tw_pass, pl.d_tw_aux, pl.d_bs_bhat, pl.W,
                           (long long)n_rows);
        return hipGetLastError();
    });
}

hipError_t launch_mdct_bs32(const zafx_plan& pl, const float* x, float* out, int64_t n_clips, int64_t n_samples, int T) {
    const long long total = (long long)n_clips * T;
    if (total <= 0) return hipSuccess;
    return by_log2m(pl.bs_log2m, [&](auto tag) {
        constexpr int L = decltype(tag)::value;
        auto kern = k_mdct_bs32<L>;
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, BsCfg<L>::SMEM); e != hipSuccess) return e;
        pl.ran = "k_mdct_bs32";
        hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(BsCfg<L>::P), BsCfg<L>::SMEM, pl.stream, x, pl.d_window, pl.d_tw_pass, pl.d_bs_chirp,
                           pl.d_bs_bhat, pl.d_tw_aux, out, (long long)n_samples, T, (int)row_pitch(pl, T), pl.W, pl.layout, total);
        return hipGetLastError();
    });
}

hipError_t launch_imdct_bs32(zafx_plan& pl, const float* coefs_all, float* y_all, int64_t n_clips_all, int T, int64_t out_len) {
    if ((long long)n_clips_all * T <= 0 || out_len <= 0) return hipSuccess;
    const int64_t chunk = scratch_clips_per_chunk(n_clips_all, T, pl.W, sizeof(float));
    if (hipError_t e = grow_scratch(pl, (size_t)chunk * T * pl.W * sizeof(float)); e != hipSuccess) return e;
    const int64_t in_per_clip = pl.layout == ZAFX_LAYOUT_FT ? (int64_t)(pl.W / 2) * row_pitch(pl, T) : (int64_t)T * (pl.W / 2);
    for (int64_t c0 = 0; c0 < n_clips_all; c0 += chunk) {
        const int64_t n_clips = std::min(chunk, n_clips_all - c0);
        const long long total = (long long)n_clips * T;
        const float* coefs = coefs_all + c0 * in_per_clip;
        hipError_t e = by_log2m(pl.bs_log2m, [&](auto tag) {
            constexpr int L = decltype(tag)::value;
            auto kern = k_imdct_frames_bs32<L>;
            if (hipError_t e2 = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, BsCfg<L>::SMEM); e2 != hipSuccess) return e2;
            pl.ran = "k_imdct_frames_bs32";
            hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(BsCfg<L>::P), BsCfg<L>::SMEM, pl.stream, coefs, pl.d_window, pl.d_tw_pass, pl.d_bs_chirp,
                               pl.d_bs_bhat, pl.d_tw_aux, reinterpret_cast<float*>(pl.d_scratch64), T, (int)row_pitch(pl, T), pl.W, pl.layout, total);
            return hipGetLastError();
        });
        if (e != hipSuccess) return e;
        // two-frame TDAC overlap-add in ascending frame order (zaf.py:1172-1179) and the trim [H : -H-1] (:1182): the ISTFT's
        // gather with hop = W/2
        if (hipError_t e2 = launch_ola(pl, y_all + c0 * out_len, n_clips, T, pl.H, out_len, 1.f); e2 != hipSuccess) return e2;
    }
    return hipSuccess;
}

}  // namespace zafx
