// zafx_dct.hip -- zaf.dct / zaf.dst, types I-IV (zaf.py:703-839, :842-981), on the FFT core.
//
// The reference evaluates each of the eight transforms as ONE np.fft.fft of a symmetric extension of the vector -- 2N-2
// points (DCT-I), 2N+2 (DST-I), 4N (types II / III), 8N (type IV) -- and keeps the real or imaginary parts of N of its bins.
// Three quarters and more of those extensions are zeros and mirror images; the same numbers come out of ONE M-point COMPLEX
// transform per vector, M = N/2 (types II-IV), N-1 (DCT-I), N+1 (DST-I), with a pre-map in front and a post-map behind:
//
//   type I    the even (DCT) / odd (DST) extension v of 2M real points is transformed as z[m] = v[2m] + i v[2m+1] and split
//             into V[0..M] (split_pair); DCT: Re V[k] / 2, DST: -Im V[k+1] / 2        (zaf.py:769-776, :906-910)
//   type II   v = even-indexed samples followed by the odd-indexed ones reversed (the 4N extension's non-zero quarter, folded),
//             real transform as above, out[k] = Re(e^{-i pi k / 2N} V[k]), out[N-k] = -Im(...)        (zaf.py:788-792)
//   type III  the transpose: H[k] = e^{i pi k / 2N} (x[k] - i x[N-k]) / 2 is the Hermitian half of v's spectrum; the packed
//             inverse real transform runs on the forward core through conj(FFT(conj .))          (zaf.py:811-817)
//   type IV   t[n] = (x[2n] + i x[N-1-2n]) e^{-i pi (4n+1) / 4N}, u = FFT_M(t)[k] e^{-i pi k / N}, out[2k] = Re u,
//             out[N-1-2k] = -Im u -- the W/4-point algorithm of k_mdct_ft32 without the window       (zaf.py:828-835)
//   DST II / III / IV are the DCTs of the sign-alternated / reversed vector (DST-II[k] = DCT-II((-1)^n x)[N-1-k],
//             DST-III[n] = (-1)^n DCT-III(reversed x)[n], DST-IV[k] = (-1)^k DCT-IV(reversed x)[k]): index and sign
//             changes inside the same maps                                                     (zaf.py:922-981)
//
// and the orthonormal scalings of zaf.py:764-766, :779-780, :795-796, :806, :820, :838 applied on the way out.  One vector is a
// "frame" of the FFT core (P = M / E threads); a workgroup of 256 threads (or P, if larger) transforms 256 / P vectors at a
// time: rows are loaded and stored as whole contiguous runs (16-byte pieces when N % 4 == 0), the maps work on LDS.
// HBM-bound: 8 bytes per sample.
//
// Round 6, BS = true: lengths N = 4 j whose N/2 is NOT a power of two (types II-IV; N = 1000 ...): the SAME maps around an Mh = N/2-point
// transform of any length, evaluated as a Bluestein convolution of 2^LOG2M >= 2 Mh - 1 points (z c -> FFT -> x Bhat, conj -> FFT -> c conj(.)):
// two transforms of half the length k_dct_bs32's chirp-z sum takes (N = 1000: 1024 points instead of 2048).  Types I and lengths that are not
// a multiple of four stay on k_dct_bs32.
#include "zafx_internal.hpp"

namespace zafx {

constexpr int dct_threads(int log2m) { return fft_threads(log2m, default_log2e(log2m)) > 256 ? fft_threads(log2m, default_log2e(log2m)) : 256; }

template <int LOG2M, bool BS = false>
struct DctCfg {
    static constexpr int LOG2E = default_log2e(LOG2M);
    using C = FftCfg<LOG2M, LOG2E>;
    static constexpr int M = C::N, E = C::E, P = C::P;
    static constexpr int NT = dct_threads(LOG2M);
    static constexpr int R = NT / P;              // vectors per workgroup pass
    static constexpr int XP = (BS ? M : 2 * M) + 4;   // floats per staged row (N <= M + 1 for type I, 2 M otherwise; BS: N = 2 Mh <= M + 1; 16-byte multiple)
    static constexpr size_t SMEM = (size_t)R * XP * 4 + (size_t)R * C::PITCH * 8;
};

// FAM = 1 .. 4 (the type); sine: the DST of that type.  tab = [A: M + 1 entries | B: M + 1 entries] (zafx_capi.cpp):
//   A[k] = exp(-2 pi i k / 2M) (types I-III: split roots)   or exp(-i pi (4k+1) / 4N) (type IV: pre-twiddle)
//   B[k] = exp(-i pi k / 2N)   (types II, III)              or exp(-i pi k / N)       (type IV: post-twiddle)
// BS: the transform has mh = N / 2 points (any number), LOG2M is the convolution's length; chirp[mh] = exp(-i pi m^2 / mh), bhat[2^LOG2M] = the
// transform of its wrapped conjugate (zafx_capi.cpp: bluestein_tables); scale carries the 1 / 2^LOG2M of the second transform.
template <int LOG2M, int FAM, bool BS = false>
__global__ __launch_bounds__(dct_threads(LOG2M)) void k_dct(const float* __restrict__ x, float* __restrict__ y, const float2* __restrict__ tw,
                                                            const float2* __restrict__ tab, int N, int sine, float scale, long long n_rows,
                                                            int n_tiles, int mh, const float2* __restrict__ chirp, const float2* __restrict__ bhat) {
    using G = DctCfg<LOG2M, BS>;
    using C = typename G::C;
    constexpr int E = G::E, P = G::P, NT = G::NT, R = G::R, XP = G::XP;
    static_assert(!BS || FAM != 1, "type I stays on the chirp-z sum");
    const int M = BS ? mh : G::M;   // points of the transform the maps are built around (a constant without BS)
    extern __shared__ __align__(16) unsigned char smem[];
    float* const xs = reinterpret_cast<float*>(smem);
    float2* const bufs = reinterpret_cast<float2*>(smem + (size_t)R * XP * 4);
    const int tid = threadIdx.x, r = tid / P;
    int p = tid % P;
    float* const xr = xs + r * XP;
    float2* const buf = bufs + r * C::PITCH;
    const float2* const tabA = tab;
    const float2* const tabB = tab + (M + 1);
    const float rs2 = 0.70710678118654752440f;
    const bool vec = (N & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long row0 = (long long)tile * R;
        const int rows = (int)(n_rows - row0 < R ? n_rows - row0 : R);
        // BS: the thread's table values (chirp, Bhat, A, B at its own indices) are the same in every pass; carried across the passes they take 244
        // registers at 1024 points -- two workgroups per CU where LDS admits three -- and spill when capped (N = 1000: 0.108 ms carried, 0.203
        // capped to three waves, 0.070 re-read per pass from L1: 110-122 registers)
        if constexpr (BS) asm volatile("" : "+v"(p));
        // ---- rows of this pass: one contiguous run of rows * N floats
        if (vec) {
            const int q = N >> 2, total = rows * q;
            const float4* src = reinterpret_cast<const float4*>(x + row0 * N);
            for (int i = tid; i < total; i += NT) {
                const int rr = i / q, c = i - rr * q;
                *reinterpret_cast<float4*>(xs + rr * XP + 4 * c) = src[i];
            }
        } else {
            const int total = rows * N;
            const float* src = x + row0 * N;
            for (int i = tid; i < total; i += NT) {
                const int rr = i / N, c = i - rr * N;
                xs[rr * XP + c] = src[i];
            }
        }
        __syncthreads();
        // ---- pre-map: the frame's points v[i] = z[p + i P]
        float2 v[E];
        if (r < rows) {
#pragma unroll
            for (int i = 0; i < E; ++i) {
                // BS: the points m >= M are the convolution's zero padding.  M <= (2^LOG2M + 1) / 2, so i > E / 2 is padding whatever p; the rest is
                // computed at a clamped index and zeroed -- no data-dependent branch around the table loads (behind one, each is waited for alone)
                if (BS && i > E / 2) {
                    v[i] = make_float2(0.f, 0.f);
                    continue;
                }
                const int m0 = p + i * P;
                const int m = BS ? min(m0, M - 1) : m0;
                if constexpr (FAM == 1) {
                    auto ext = [&](int j) -> float {   // the 2M-point even / odd extension
                        if (!sine) {
                            const float a = xr[j <= M ? j : 2 * M - j];
                            return (j == 0 || j == M) ? a * 1.41421356237309504880f : a;   // zaf.py:765-766
                        }
                        if (j == 0 || j == M) return 0.f;
                        return j < M ? xr[j - 1] : -xr[2 * M - 1 - j];                     // zaf.py:906-908
                    };
                    v[i] = make_float2(ext(2 * m), ext(2 * m + 1));
                } else if constexpr (FAM == 2) {
                    const bool lo = 2 * m < M;
                    const float sg = (sine && !lo) ? -1.f : 1.f;   // (-1)^n on the odd-indexed samples
                    v[i] = make_float2(sg * xr[lo ? 4 * m : 2 * N - 1 - 4 * m], sg * xr[lo ? 4 * m + 2 : 2 * N - 3 - 4 * m]);
                } else if constexpr (FAM == 3) {
                    auto xp = [&](int j) -> float {   // x'[j]: scaled first coefficient (zaf.py:806 / :941), reversed for the DST
                        const int jj = min(j, N - 1);
                        const float a = xr[sine ? N - 1 - jj : jj];
                        return j >= N ? 0.f : j == 0 ? a * rs2 : a;
                    };
                    auto H = [&](int k) -> float2 {   // Hermitian half of the spectrum of v
                        const float2 h = cmulc(make_float2(xp(k), -xp(N - k)), tabB[k]);
                        return k == 0 ? make_float2(xp(0), 0.f) : make_float2(0.5f * h.x, 0.5f * h.y);
                    };
                    const float2 hk = H(m), hn = cconj(H(M - m));
                    const float2 d = cmulc(csub(hk, hn), tabA[m]);        // e^{+2 pi i m / N} (H[m] - H[m + M])
                    const float2 zp = make_float2(hk.x + hn.x - d.y, hk.y + hn.y + d.x);
                    v[i] = cconj(zp);
                } else {
                    const float a = xr[2 * m], b = xr[N - 1 - 2 * m];
                    v[i] = cmul(sine ? make_float2(b, a) : make_float2(a, b), tabA[m]);
                }
                if constexpr (BS) {
                    const float2 c = chirp[m];
                    v[i] = m0 < M ? cmul(v[i], c) : make_float2(0.f, 0.f);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < E; ++i) v[i] = make_float2(0.f, 0.f);
        }
        fft_frame<LOG2M, G::LOG2E>(v, buf, p, tw);
        if constexpr (BS) {   // x Bhat, conj (the next forward transform inverts), transform: buf[k] = conj(Z[k]) 2^LOG2M / c[k]
            regs_read<LOG2M, G::LOG2E>(v, buf, p);
#pragma unroll
            for (int i = 0; i < E; ++i) v[i] = cconj(cmul(v[i], bhat[p + i * P]));
            frame_sync<P>();
            fft_frame<LOG2M, G::LOG2E>(v, buf, p, tw);
        }
        auto Z = [&](int k) -> float2 {   // bin k of the M-point transform
            const float2 b = buf[phys_t<C::PS>(k)];
            if constexpr (BS) return cmul(chirp[k], cconj(b));
            else return b;
        };
        // ---- post-map: results into the row's staging floats (every read of the row happened before the transform's exchanges)
        if (r < rows) {
            if constexpr (FAM == 1 || FAM == 2) {
#pragma unroll
                for (int i = 0; i <= (BS ? E / 4 : E / 2); ++i) {   // (BS: M / 2 <= 2^LOG2M / 4, and as above: clamp, compute, keep the stores conditional)
                    const int k0 = p + i * P;
                    if constexpr (!BS) {
                        if (k0 > M / 2) continue;
                    }
                    const int k = BS ? min(k0, M / 2) : k0;
                    const bool live = !BS || k0 <= M / 2;
                    const float2 zk = Z(k), zn = Z(k == 0 ? 0 : M - k);
                    float2 vk, vn;
                    split_pair(zk, zn, tabA[k], vk, vn);   // V[k], V[M - k]
                    if constexpr (FAM == 1) {
                        const float s = 0.5f * scale;
                        if (!live) {
                        } else if (!sine) {
                            xr[k] = vk.x * (k == 0 ? s * rs2 : s);
                            xr[M - k] = vn.x * (k == 0 ? s * rs2 : s);
                        } else {
                            if (k >= 1) {
                                xr[k - 1] = -vk.y * s;
                                xr[M - k - 1] = -vn.y * s;
                            }
                        }
                    } else {
                        auto put = [&](int idx, float val) {
                            if (live) xr[sine ? N - 1 - idx : idx] = val;
                        };
                        const float2 a = cmul(vk, tabB[k]), b = cmul(vn, tabB[M - k]);
                        put(k, a.x * (k == 0 ? scale * rs2 : scale));
                        if (k > 0) put(N - k, -a.y * scale);
                        put(M - k, b.x * scale);
                        put(M + k, -b.y * scale);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < (BS ? E / 2 + 1 : E); ++i) {
                    const int m0 = p + i * P;
                    const int m = BS ? min(m0, M - 1) : m0;
                    const bool live = !BS || m0 < M;
                    const float2 z = Z(m);
                    if constexpr (FAM == 3) {
                        // v[2m] = Re, v[2m+1] = -Im of the forward transform of the conjugate; v[j] is y[2j] (j < M) or y[2N-1-2j]
                        const int j0 = 2 * m, j1 = 2 * m + 1;
                        const int o0 = j0 < M ? 2 * j0 : 2 * N - 1 - 2 * j0, o1 = j1 < M ? 2 * j1 : 2 * N - 1 - 2 * j1;
                        const float sg = (sine && j0 >= M) ? -scale : scale;   // (both of a pair are on the same side; odd outputs of the DST flip)
                        if (live) {
                            xr[o0] = z.x * sg;
                            xr[o1] = -z.y * sg;
                        }
                    } else {
                        const float2 u = cmul(z, tabB[m]);
                        if (live) {
                            xr[2 * m] = u.x * scale;
                            xr[N - 1 - 2 * m] = sine ? u.y * scale : -u.y * scale;
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (vec) {
            const int q = N >> 2, total = rows * q;
            float4* dst = reinterpret_cast<float4*>(y + row0 * N);
            for (int i = tid; i < total; i += NT) {
                const int rr = i / q, c = i - rr * q;
                dst[i] = *reinterpret_cast<const float4*>(xs + rr * XP + 4 * c);
            }
        } else {
            const int total = rows * N;
            float* dst = y + row0 * N;
            for (int i = tid; i < total; i += NT) {
                const int rr = i / N, c = i - rr * N;
                dst[i] = xs[rr * XP + c];
            }
        }
        __syncthreads();
    }
}

bool dct_supported(int log2m) { return log2m >= 5 && log2m <= 13; }
const char* dct_kernel_name() { return "k_dct"; }

template <int LOG2M, int FAM, bool BS = false>
static hipError_t run_dct(const zafx_plan& pl, const float* x, float* y, int64_t n_rows) {
    using G = DctCfg<LOG2M, BS>;
    static_assert(G::SMEM <= (size_t)kMaxLdsBytes, "rows + frames exceed LDS");
    auto kern = k_dct<LOG2M, FAM, BS>;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, G::SMEM); e != hipSuccess) return e;
    const long long tiles = (n_rows + G::R - 1) / G::R;
    if (tiles <= 0) return hipSuccess;
    int per_cu = (int)std::max<size_t>(1, std::min<size_t>(std::min<size_t>(2048 / G::NT, 8), (size_t)kMaxLdsBytes / G::SMEM));
    if constexpr (BS) {   // the persistent grid = what is resident (registers hold fewer workgroups than LDS here; a grid beyond that runs a second, half-empty round)
        int resident = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, kern, G::NT, G::SMEM) == hipSuccess && resident > 0) per_cu = std::min(per_cu, resident);
    }
    const long long grid = std::min<long long>(tiles, (long long)pl.n_cus * per_cu);
    const int N = pl.W;
    const int M = 1 << LOG2M;
    const float scale = FAM == 1 ? std::sqrt(2.0f / (float)M) : BS ? std::sqrt(2.0f / (float)N) / (float)M : std::sqrt(2.0f / (float)N);
    pl.ran = BS ? "k_dct_bsh" : "k_dct";
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(G::NT), G::SMEM, pl.stream, x, y, pl.d_tw_pass, pl.d_tw_aux, N, pl.prm.transform_sine, scale,
                       (long long)n_rows, (int)tiles, pl.dct_half, pl.d_bs_chirp, pl.d_bs_bhat);
    return hipGetLastError();
}

template <int LOG2M>
static hipError_t dispatch_dct_half(const zafx_plan& pl, const float* x, float* y, int64_t n_rows) {
    switch (pl.prm.transform_type) {
        case 2: return run_dct<LOG2M, 2, true>(pl, x, y, n_rows);
        case 3: return run_dct<LOG2M, 3, true>(pl, x, y, n_rows);
        case 4: return run_dct<LOG2M, 4, true>(pl, x, y, n_rows);
    }
    return hipErrorInvalidValue;
}

template <int LOG2M>
static hipError_t dispatch_dct(const zafx_plan& pl, const float* x, float* y, int64_t n_rows) {
    switch (pl.prm.transform_type) {
        case 1: return run_dct<LOG2M, 1>(pl, x, y, n_rows);
        case 2: return run_dct<LOG2M, 2>(pl, x, y, n_rows);
        case 3: return run_dct<LOG2M, 3>(pl, x, y, n_rows);
        case 4: return run_dct<LOG2M, 4>(pl, x, y, n_rows);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_dct(const zafx_plan& pl, const float* x, float* y, int64_t n_rows) {
    if (n_rows >= (1LL << 31)) return hipErrorInvalidValue;
    if (pl.dct_half > 0) {   // N / 2 points, not a power of two: the maps around a Bluestein convolution of 2^bs_log2m points
        switch (pl.bs_log2m) {
#define X(L) \
    case L: return dispatch_dct_half<L>(pl, x, y, n_rows);
            X(7) X(8) X(9) X(10) X(11) X(12) X(13)
#undef X
        }
        return hipErrorInvalidValue;
    }
    switch (pl.log2nf) {
#define X(L) \
    case L: return dispatch_dct<L>(pl, x, y, n_rows);
        X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13)
#undef X
    }
    return hipErrorInvalidValue;
}

}  // namespace zafx
