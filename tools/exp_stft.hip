// Experimental playground for STFT kernel structure (not part of the product).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I zaf-python_amd/csrc tools/exp_stft.hip -o gpurun_out/exp_stft
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstring>
#include "zafx_fft.hpp"
#include "zafx_twiddle.hpp"
using namespace zafx;
typedef float f32x2_t0 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 nt_load2a(const float* p) {
    const f32x2_t0 v = __builtin_nontemporal_load(reinterpret_cast<const f32x2_t0*>(p));
    return make_float2(v.x, v.y);
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__device__ __forceinline__ void split_pair(float2 zk, float2 zn, float2 t, float2& xk, float2& xn) {
    const float2 e = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
    const float2 d = make_float2(0.5f * (zk.x - zn.x), 0.5f * (zk.y + zn.y));
    const float2 o = make_float2(d.y, -d.x);
    const float2 to = cmul(t, o);
    xk = cadd(e, to);
    xn = cconj(csub(e, to));
}

// DBG bit0: skip stores; bit1: skip loads; bit2: skip fft ; REMAP: pair tiles on one XCD
template <int FPB, int DBG, int REMAP>
__global__ __launch_bounds__(FPB * 64) void k_stft_v(
    const float* __restrict__ x, const float* __restrict__ win, const float2* __restrict__ twp,
    const float2* __restrict__ tws, float2* __restrict__ out, long long n_samples, int hop, int T, int tiles, float magic) {
    using C = FftCfg<10, 4>;
    constexpr int N = C::N, P = C::P, E = C::E, W = 2 * N, NT = FPB * P;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* frames = reinterpret_cast<float2*>(smem_raw);
    float2* tw_l = frames + FPB * C::PITCH;
    const int tid = threadIdx.x;
    for (int i = tid; i < C::TW; i += NT) tw_l[i] = twp[i];
    __syncthreads();
    const int slot = tid / P, p = tid % P;
    int b = blockIdx.x;
    if (REMAP) {   // blocks b and b+8 (same XCD) take adjacent tiles
        const int g = b >> 4, r = b & 15;
        b = g * 16 + ((r & 7) << 1) + (r >> 3);
    }
    const int clip = b / tiles, tile = b % tiles;
    if (clip * 1 >= gridDim.x / tiles + 1) return;
    const int t0 = tile * FPB;
    const int t = t0 + slot;
    float2* buf = frames + slot * C::PITCH;
    float2 v[E];
    {
        const float* xc = x + (long long)clip * n_samples;
        const long long s0 = (long long)t * hop - N;
        const float2* w2 = reinterpret_cast<const float2*>(win);
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const int n = p + i * P;
            const long long s = s0 + 2 * n;
            const float2 wv = w2[n];
            float a, bb;
            if (DBG & 2) { a = magic * n; bb = magic; }
            else {
                a = (t < T && s >= 0 && s < n_samples) ? xc[s] : 0.f;
                bb = (t < T && s + 1 >= 0 && s + 1 < n_samples) ? xc[s + 1] : 0.f;
            }
            v[i] = make_float2(a * wv.x, bb * wv.y);
        }
    }
    if (DBG & 4) {
#pragma unroll
        for (int i = 0; i < E; ++i) buf[phys(p + i * P)] = v[i];
        frame_sync<P>();
    } else {
        fft_frame<10, 4>(v, buf, p, tw_l);
    }
    if constexpr (NT > 64) __syncthreads();
    const int tt = tid % FPB, kq = tid / FPB;
    if (t0 + tt >= T) return;
    const float2* fb = frames + tt * C::PITCH;
    float2* o = out + (long long)clip * W * T + (t0 + tt);
    for (int k = kq; k < N / 2; k += P) {
        float2 xk, xn, xa, xb;
        if (k == 0) {
            const float2 z0 = fb[0], zc = fb[phys(N / 2)];
            xk = make_float2(z0.x + z0.y, 0.f); xn = make_float2(z0.x - z0.y, 0.f); xa = cconj(zc); xb = zc;
            if (!(DBG & 1) || xk.x == magic) {
                o[0] = xk; o[(long long)N * T] = xn; o[(long long)(N / 2) * T] = xa; o[(long long)(N + N / 2) * T] = xb;
            }
        } else {
            split_pair(fb[phys(k)], fb[phys(N - k)], tws[k], xk, xn);
            if (!(DBG & 1) || xk.x == magic) {
                o[(long long)k * T] = xk;
                o[(long long)(W - k) * T] = cconj(xk);
                o[(long long)(N - k) * T] = xn;
                o[(long long)(N + k) * T] = cconj(xn);
            }
        }
    }
}


// ---- persistent, prefetching variant: one WG per CU loops over tiles; the next tile's raw
// samples are fetched into registers before the current tile's stores are issued.
template <int FPB, int PITCH, int DBG>
__global__ __launch_bounds__(FPB * 64) void k_stft_p(
    const float* __restrict__ x, const float* __restrict__ win, const float2* __restrict__ twp,
    const float2* __restrict__ tws, float2* __restrict__ out, long long n_samples, int hop, int T, int tiles, int total_tiles) {
    using C = FftCfg<10, 4>;
    constexpr int N = C::N, P = C::P, E = C::E, W = 2 * N, NT = FPB * P;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* frames = reinterpret_cast<float2*>(smem_raw);
    float2* tw_l = frames + FPB * PITCH;
    float2* win_l = tw_l + C::TW;              // N float2
    float2* tws_l = win_l + N;                 // N/2+1
    const int tid = threadIdx.x;
    for (int i = tid; i < C::TW; i += NT) tw_l[i] = twp[i];
    for (int i = tid; i < N; i += NT) win_l[i] = reinterpret_cast<const float2*>(win)[i];
    for (int i = tid; i <= N / 2; i += NT) tws_l[i] = tws[i];
    __syncthreads();
    const int slot = tid / P, p = tid % P;
    float2* buf = frames + slot * PITCH;
    const int tt = tid % FPB, kq = tid / FPB;
    const float2* fb = frames + tt * PITCH;

    float2 xr[E];
    auto prefetch = [&](int tl) {
        const int clip = tl / tiles, tile = tl % tiles;
        const int t = tile * FPB + slot;
        const float* xc = x + (long long)clip * n_samples;
        const long long s0 = (long long)t * hop - N;
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const long long s = s0 + 2 * (p + i * P);
            if (tl < total_tiles && t < T && s >= 0 && s + 1 < n_samples) xr[i] = *reinterpret_cast<const float2*>(xc + s);
            else {
                xr[i].x = (tl < total_tiles && t < T && s >= 0 && s < n_samples) ? xc[s] : 0.f;
                xr[i].y = (tl < total_tiles && t < T && s + 1 >= 0 && s + 1 < n_samples) ? xc[s + 1] : 0.f;
            }
        }
    };
    int tl = blockIdx.x;
    prefetch(tl);
    for (; tl < total_tiles; tl += gridDim.x) {
        const int clip = tl / tiles, tile = tl % tiles;
        const int t0 = tile * FPB;
        float2 v[E];
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const float2 wv = win_l[p + i * P];
            v[i] = make_float2(xr[i].x * wv.x, xr[i].y * wv.y);
        }
        fft_frame<10, 4>(v, buf, p, tw_l);   // NOTE: uses C::PITCH-independent phys(); buf pitch is ours
        __syncthreads();
        prefetch(tl + gridDim.x);
        if (t0 + tt < T) {
            float2* o = out + (long long)clip * W * T + (t0 + tt);
            for (int k = kq; k < N / 2; k += P) {
                float2 xk, xn;
                if (k == 0) {
                    const float2 z0 = fb[0], zc = fb[phys(N / 2)];
                    if (!(DBG & 1)) {
                    o[0] = make_float2(z0.x + z0.y, 0.f); o[(long long)N * T] = make_float2(z0.x - z0.y, 0.f);
                    o[(long long)(N / 2) * T] = cconj(zc); o[(long long)(N + N / 2) * T] = zc; }
                } else {
                    split_pair(fb[phys(k)], fb[phys(N - k)], tws_l[k], xk, xn);
                    if (!(DBG & 1) || xk.x == 12345.f) {
                    o[(long long)k * T] = xk;
                    o[(long long)(W - k) * T] = cconj(xk);
                    o[(long long)(N - k) * T] = xn;
                    o[(long long)(N + k) * T] = cconj(xn); }
                }
            }
        }
        __syncthreads();
    }
}


// ---- multi-frame FFT: FPW frames per wave advance pass by pass together (half the fences, 2x ILP)
template <int LOG2N, int LOG2E, int FPW, int LOG2NS = 0>
__device__ __forceinline__ void fft_frames(float2 (&v)[FPW][1 << LOG2E], float2* buf0, int pitch, int p, const float2* tw) {
    using C = FftCfg<LOG2N, LOG2E>;
    if constexpr (LOG2NS < LOG2N) {
        constexpr int LR = pass_log2r(LOG2N - LOG2NS, LOG2E);
#pragma unroll
        for (int f = 0; f < FPW; ++f)
            pass_write<LOG2N, LOG2E, LOG2NS, LR>(v[f], buf0 + f * pitch, p, tw + twiddle_offset(LOG2N, LOG2E, LOG2NS));
        frame_sync<C::P>();
        if constexpr (LOG2NS + LR < LOG2N) {
#pragma unroll
            for (int f = 0; f < FPW; ++f) regs_read<LOG2N, LOG2E>(v[f], buf0 + f * pitch, p);
            frame_sync<C::P>();
            fft_frames<LOG2N, LOG2E, FPW, LOG2NS + LR>(v, buf0, pitch, p, tw);
        }
    }
}

// ---- persistent variant 2: 16-frame tile, WAVES waves, each wave transforms 16/WAVES frames
// (fewer, fatter waves: up to 256 VGPRs at 8 waves/CU, so the prefetch never spills)
template <int WAVES, int PITCH, int DBG, int MF = 0>
__global__ __launch_bounds__(WAVES * 64) void k_stft_q(
    const float* __restrict__ x, const float* __restrict__ win, const float2* __restrict__ twp,
    const float2* __restrict__ tws, float2* __restrict__ out, long long n_samples, int hop, int T, int tiles, int total_tiles, unsigned long long* __restrict__ prof = nullptr) {
    unsigned long long acc_fft = 0, acc_bar1 = 0, acc_pre = 0, acc_store = 0, acc_bar2 = 0, ntile = 0;
    using C = FftCfg<10, 4>;
    constexpr int N = C::N, P = C::P, E = C::E, W = 2 * N, NT = WAVES * 64, FPB = 16, FPW = FPB / WAVES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* frames = reinterpret_cast<float2*>(smem_raw);
    float2* tw_l = frames + FPB * PITCH;
    float2* win_l = tw_l + C::TW;              // N float2
    float2* tws_l = win_l + N;                 // N/2+1
    const int tid = threadIdx.x;
    for (int i = tid; i < C::TW; i += NT) tw_l[i] = twp[i];
    for (int i = tid; i < N; i += NT) win_l[i] = reinterpret_cast<const float2*>(win)[i];
    for (int i = tid; i <= N / 2; i += NT) tws_l[i] = tws[i];
    __syncthreads();
    const int wave = tid / P, p = tid % P;
    const int tt = tid % FPB, kq = tid / FPB;   // kq in [0, NT/16)
    const float2* fb = frames + tt * PITCH;

    float2 xr[FPW][E];
    auto prefetch = [&](int tl) {
        const int clip = tl / tiles, tile = tl % tiles;
        const float* xc = x + (long long)clip * n_samples;
#pragma unroll
        for (int f = 0; f < FPW; ++f) {
            const int t = tile * FPB + wave * FPW + f;
            const long long s0 = (long long)t * hop - N;
            const bool inside = tl < total_tiles && t < T && s0 >= 0 && s0 + W <= n_samples;
            if (inside) {
#pragma unroll
                for (int i = 0; i < E; ++i) xr[f][i] = *reinterpret_cast<const float2*>(xc + s0 + 2 * (p + i * P));
            } else {
#pragma unroll
                for (int i = 0; i < E; ++i) {
                    const long long s = s0 + 2 * (p + i * P);
                    const bool ok = tl < total_tiles && t < T;
                    xr[f][i].x = (ok && s >= 0 && s < n_samples) ? xc[s] : 0.f;
                    xr[f][i].y = (ok && s + 1 >= 0 && s + 1 < n_samples) ? xc[s + 1] : 0.f;
                }
            }
        }
    };
    int tl = blockIdx.x;
    prefetch(tl);
    if (DBG & 32) {   // stagger the CUs' load/FFT/store phases: (block % 4) quarter periods
        const unsigned long long t_start = __builtin_readcyclecounter();
        const unsigned long long wait = (unsigned long long)(blockIdx.x & 3) * 9500ull;
        while (__builtin_readcyclecounter() - t_start < wait) __builtin_amdgcn_s_sleep(8);
    }
    if (DBG & 64) {   // 2-phase stagger, half a period
        const unsigned long long t_start = __builtin_readcyclecounter();
        const unsigned long long wait = (unsigned long long)(blockIdx.x & 1) * 19000ull;
        while (__builtin_readcyclecounter() - t_start < wait) __builtin_amdgcn_s_sleep(8);
    }
    for (; tl < total_tiles; tl += gridDim.x) {
        const int clip = tl / tiles, tile = tl % tiles;
        const int t0 = tile * FPB;
        const float2* tw_i = tw_l; const float2* win_i = win_l; const float2* tws_i = tws_l;
        if (DBG & 128) { int zero = 0; asm volatile("" : "+s"(zero)); tw_i += zero; win_i += zero; tws_i += zero; }
        unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0;
        if (DBG & 8) c0 = __builtin_readcyclecounter();
        if constexpr (MF) {
            float2 v[FPW][E];
#pragma unroll
            for (int f = 0; f < FPW; ++f)
#pragma unroll
                for (int i = 0; i < E; ++i) {
                    const float2 wv = win_i[p + i * P];
                    v[f][i] = make_float2(xr[f][i].x * wv.x, xr[f][i].y * wv.y);
                }
            fft_frames<10, 4, FPW>(v, frames + (wave * FPW) * PITCH, PITCH, p, tw_i);
        } else {
#pragma unroll
        for (int f = 0; f < FPW; ++f) {
            float2 v[E];
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const float2 wv = win_i[p + i * P];
                v[i] = make_float2(xr[f][i].x * wv.x, xr[f][i].y * wv.y);
            }
            fft_frame<10, 4>(v, frames + (wave * FPW + f) * PITCH, p, tw_i);
        }
        }
        if (DBG & 8) c1 = __builtin_readcyclecounter();
        __syncthreads();
        if (DBG & 8) c2 = __builtin_readcyclecounter();
        prefetch(tl + gridDim.x);
        if (DBG & 8) c3 = __builtin_readcyclecounter();
        if (t0 + tt < T) {
            float2* o = out + (long long)clip * W * T + (t0 + tt);
            for (int k = kq; k < N / 2; k += NT / FPB) {
                float2 xk, xn;
                if (k == 0) {
                    const float2 z0 = fb[0], zc = fb[phys(N / 2)];
                    if (!(DBG & 1)) {
                    o[0] = make_float2(z0.x + z0.y, 0.f); o[(long long)N * T] = make_float2(z0.x - z0.y, 0.f);
                    o[(long long)(N / 2) * T] = cconj(zc); o[(long long)(N + N / 2) * T] = zc; }
                } else {
                    split_pair(fb[phys(k)], fb[phys(N - k)], tws_i[k], xk, xn);
                    if (!(DBG & 1) || xk.x == 12345.f) {
                    o[(long long)k * T] = xk;
                    o[(long long)(W - k) * T] = cconj(xk);
                    o[(long long)(N - k) * T] = xn;
                    o[(long long)(N + k) * T] = cconj(xn); }
                }
            }
        }
        if (DBG & 8) c4 = __builtin_readcyclecounter();
        __syncthreads();
        if (DBG & 8) { c5 = __builtin_readcyclecounter(); acc_fft += c1 - c0; acc_bar1 += c2 - c1; acc_pre += c3 - c2; acc_store += c4 - c3; acc_bar2 += c5 - c4; ++ntile; }
    }
    if ((DBG & 8) && prof && (threadIdx.x & 63) == 0) {
        atomicAdd(prof + 0, acc_fft); atomicAdd(prof + 1, acc_bar1); atomicAdd(prof + 2, acc_pre); atomicAdd(prof + 3, acc_store); atomicAdd(prof + 4, acc_bar2); atomicAdd(prof + 5, ntile);
    }
}


// ---- persistent variant 3: branch-free prefetch (clamped float2 loads, masked at use), multi-frame FFT
template <int WAVES, int PITCH, int DBG>
__global__ __launch_bounds__(WAVES * 64) void k_stft_r(
    const float* __restrict__ x, const float* __restrict__ win, const float2* __restrict__ twp,
    const float2* __restrict__ tws, float2* __restrict__ out, long long n_samples, int hop, int T, int tiles, int total_tiles, unsigned long long* __restrict__ prof = nullptr) {
    using C = FftCfg<10, 4>;
    constexpr int N = C::N, P = C::P, E = C::E, W = 2 * N, NT = WAVES * 64, FPB = 16, FPW = FPB / WAVES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* frames = reinterpret_cast<float2*>(smem_raw);
    float2* tw_l = frames + FPB * PITCH;
    float2* win_l = tw_l + C::TW;              // N float2
    float2* tws_l = win_l + N;                 // N/2+1
    const int tid = threadIdx.x;
    for (int i = tid; i < C::TW; i += NT) tw_l[i] = twp[i];
    for (int i = tid; i < N; i += NT) win_l[i] = reinterpret_cast<const float2*>(win)[i];
    for (int i = tid; i <= N / 2; i += NT) tws_l[i] = tws[i];
    __syncthreads();
    const int wave = tid / P, p = tid % P;
    const int tt = tid % FPB, kq = tid / FPB;
    const float2* fb = frames + tt * PITCH;

    float2 xr[FPW][E];
    long long s_base[FPW];     // first sample of each of my frames in the prefetched tile (or a huge value = invalid)
    auto prefetch = [&](int tl) {
        const int clip = tl / tiles, tile = tl % tiles;
        const float* xc = x + (long long)clip * n_samples;
#pragma unroll
        for (int f = 0; f < FPW; ++f) {
            const int t = tile * FPB + wave * FPW + f;
            const bool ok = tl < total_tiles && t < T;
            const long long s0 = ok ? (long long)t * hop - N : (long long)1 << 40;
            s_base[f] = s0;
            const float* src = ok ? xc : x;
#pragma unroll
            for (int i = 0; i < E; ++i) {
                long long s = s0 + 2 * (p + i * P);
                s = s < 0 ? 0 : (s > n_samples - 2 ? n_samples - 2 : s);
                xr[f][i] = *reinterpret_cast<const float2*>(src + s);
            }
        }
    };
    int tl = blockIdx.x;
    prefetch(tl);
    for (; tl < total_tiles; tl += gridDim.x) {
        const int clip = tl / tiles, tile = tl % tiles;
        const int t0 = tile * FPB;
        float2 v[FPW][E];
#pragma unroll
        for (int f = 0; f < FPW; ++f) {
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const float2 wv = win_l[p + i * P];
                const long long s = s_base[f] + 2 * (p + i * P);
                const bool in = s >= 0 && s < n_samples;   // n_samples even, s even: covers s+1 too
                v[f][i] = in ? make_float2(xr[f][i].x * wv.x, xr[f][i].y * wv.y) : make_float2(0.f, 0.f);
            }
        }
        fft_frames<10, 4, FPW>(v, frames + (wave * FPW) * PITCH, PITCH, p, tw_l);
        __syncthreads();
        prefetch(tl + gridDim.x);
        if (t0 + tt < T) {
            float2* o = out + (long long)clip * W * T + (t0 + tt);
            for (int k = kq; k < N / 2; k += NT / FPB) {
                float2 xk, xn;
                if (k == 0) {
                    const float2 z0 = fb[0], zc = fb[phys(N / 2)];
                    if (!(DBG & 1)) {
                    o[0] = make_float2(z0.x + z0.y, 0.f); o[(long long)N * T] = make_float2(z0.x - z0.y, 0.f);
                    o[(long long)(N / 2) * T] = cconj(zc); o[(long long)(N + N / 2) * T] = zc; }
                } else {
                    split_pair(fb[phys(k)], fb[phys(N - k)], tws_l[k], xk, xn);
                    if (!(DBG & 1) || xk.x == 12345.f) {
                    o[(long long)k * T] = xk;
                    o[(long long)(W - k) * T] = cconj(xk);
                    o[(long long)(N - k) * T] = xn;
                    o[(long long)(N + k) * T] = cconj(xn); }
                }
            }
        }
        __syncthreads();
    }
}


// ---- parity-split variant: blockIdx&1 = parity q.  A workgroup transforms 16 frames but only the
// bins k = q (mod 2): one radix-2 DIF stage in registers, then a 512-point FFT (E=8, one wave per frame,
// ~64 VGPRs) -> LDS 70 KB per WG, 2 WGs (32 waves) per CU, 128-B runs kept.
template <int DBG>
__global__ __launch_bounds__(1024) void k_stft_ps(
    const float* __restrict__ x, const float* __restrict__ win, const float2* __restrict__ twp,
    const float2* __restrict__ tws, float2* __restrict__ out, long long n_samples, int hop, int T, int tiles, unsigned long long* __restrict__ prof) {
    using C = FftCfg<9, 3>;
    unsigned long long tA = 0, tB = 0, tC = 0, tD = 0, tE = 0;
    if (DBG & 8) tA = __builtin_readcyclecounter();
    constexpr int NH = C::N, N = 2 * NH, P = C::P, E = C::E, W = 2 * N, FPB = 16, NT = FPB * P, PITCH = NH + NH / 16 + 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* frames = reinterpret_cast<float2*>(smem_raw);
    float2* tw_l = frames + FPB * PITCH;
    float2* tws_l = tw_l + C::TW;               // N/2 + 1 roots of W
    const int tid = threadIdx.x;
    for (int i = tid; i < C::TW; i += NT) tw_l[i] = twp[i];
    for (int i = tid; i <= N / 2; i += NT) tws_l[i] = tws[i];
    __syncthreads();
    if (DBG & 8) tB = __builtin_readcyclecounter();
    const int slot = tid / P, p = tid % P;
    const int q = blockIdx.x & 1;
    const int b = blockIdx.x >> 1;
    const int clip = b / tiles, tile = b % tiles;
    const int t0 = tile * FPB;
    const int t = t0 + slot;
    float2* buf = frames + slot * PITCH;
    float2 v[E];
    {
        const float* xc = x + (long long)clip * n_samples;
        const long long s0 = (long long)t * hop - N;
        const float2* w2 = reinterpret_cast<const float2*>(win);
        const bool inside = t < T && s0 >= 0 && s0 + W <= n_samples;
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const int n = p + i * P;
            float2 a, c2;
            if (inside) {
                a = *reinterpret_cast<const float2*>(xc + s0 + 2 * n);
                c2 = *reinterpret_cast<const float2*>(xc + s0 + 2 * (n + NH));
            } else {
                const long long sa = s0 + 2 * n, sc = s0 + 2 * (n + NH);
                a.x = (t < T && sa >= 0 && sa < n_samples) ? xc[sa] : 0.f;
                a.y = (t < T && sa + 1 >= 0 && sa + 1 < n_samples) ? xc[sa + 1] : 0.f;
                c2.x = (t < T && sc >= 0 && sc < n_samples) ? xc[sc] : 0.f;
                c2.y = (t < T && sc + 1 >= 0 && sc + 1 < n_samples) ? xc[sc + 1] : 0.f;
            }
            const float2 wa = (DBG & 16) ? make_float2(0.5f, 0.25f) : w2[n], wc = (DBG & 16) ? make_float2(0.75f, 0.5f) : w2[n + NH];
            const float2 za = make_float2(a.x * wa.x, a.y * wa.y), zc = make_float2(c2.x * wc.x, c2.y * wc.y);
            if (q == 0) v[i] = cadd(za, zc);
            else {
                // (za - zc) * exp(-2 pi i n / N) ;  exp(-2 pi i n/N) = root_W[2n]
                const int idx = 2 * n;
                float2 r = idx <= N / 2 ? tws_l[idx] : tws_l[N - idx];
                if (idx > N / 2) r = make_float2(-r.x, r.y);   // -conj
                v[i] = cmul(csub(za, zc), r);
            }
        }
    }
    if (DBG & 8) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tC = __builtin_readcyclecounter(); }
    fft_frame<9, 3>(v, buf, p, tw_l);
    __syncthreads();
    if (DBG & 8) tD = __builtin_readcyclecounter();
    const int tt = tid % FPB, mq = tid / FPB;   // mq in [0, 64)
    if (t0 + tt >= T) return;
    const float2* fb = frames + tt * PITCH;
    float2* o = out + (long long)clip * W * T + (t0 + tt);
    for (int m = mq; m < NH / 2; m += NT / FPB) {
        float2 xk, xn;
        if (q == 0 && m == 0) {
            const float2 z0 = fb[0], zc = fb[phys(NH / 2)];   // Z[0], Z[N/2] = Y0[256]
            if (!(DBG & 1)) {
            o[0] = make_float2(z0.x + z0.y, 0.f); o[(long long)N * T] = make_float2(z0.x - z0.y, 0.f);
            o[(long long)(N / 2) * T] = cconj(zc); o[(long long)(N + N / 2) * T] = zc; }
        } else {
            const int k = 2 * m + q;
            const int mp = NH - m - q;           // index of Z[N-k] in this parity's FFT
            split_pair(fb[phys(m)], fb[phys(mp)], tws_l[k], xk, xn);
            if (!(DBG & 1) || xk.x == 12345.f) {
            o[(long long)k * T] = xk;
            o[(long long)(W - k) * T] = cconj(xk);
            o[(long long)(N - k) * T] = xn;
            o[(long long)(N + k) * T] = cconj(xn); }
        }
    }
    if (DBG & 8) {
        tE = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long tF = __builtin_readcyclecounter();
        if (tid == 0) {
            atomicAdd(prof + 0, tB - tA); atomicAdd(prof + 1, tC - tB); atomicAdd(prof + 2, tD - tC);
            atomicAdd(prof + 3, tE - tD); atomicAdd(prof + 4, tF - tE); atomicAdd(prof + 5, 1ull);
        }
    }
}


// ---- persistent parity-split: grid = 2 WGs per CU; tables staged once; WG (xcd, slot) loops over tiles,
// the two parities of a tile run on the same XCD at about the same time (second one hits L2).
template <int DBG>
__global__ __launch_bounds__(1024, 8) void k_stft_pp(
    const float* __restrict__ x, const float* __restrict__ win, const float2* __restrict__ twp,
    const float2* __restrict__ tws, float2* __restrict__ out, long long n_samples, int hop, int T, int tiles, int total_tiles) {
    using C = FftCfg<9, 3>;
    constexpr int NH = C::N, N = 2 * NH, P = C::P, E = C::E, W = 2 * N, FPB = 16, NT = FPB * P, PITCH = NH + NH / 16 + 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* frames = reinterpret_cast<float2*>(smem_raw);
    float2* tw_l = frames + FPB * PITCH;
    float2* tws_l = tw_l + C::TW;
    const int tid = threadIdx.x;
    for (int i = tid; i < C::TW; i += NT) tw_l[i] = twp[i];
    for (int i = tid; i <= N / 2; i += NT) tws_l[i] = tws[i];
    __syncthreads();
    const int slot = tid / P, p = tid % P;
    const int tt = tid % FPB, mq = tid / FPB;
    float2* buf = frames + slot * PITCH;
    const float2* fb = frames + tt * PITCH;
    const float2* w2 = reinterpret_cast<const float2*>(win);
    // block -> (xcd, pair, parity)
    const int xcd = blockIdx.x & 7, sl = blockIdx.x >> 3;
    const int q = sl & 1, pair = sl >> 1, npairs = gridDim.x >> 4;
    for (int it = pair; ; it += npairs) {
        const int tl = it * 8 + xcd;
        if (tl >= total_tiles) break;
        const int clip = tl / tiles, tile = tl % tiles;
        const int t0 = tile * FPB;
        const int t = t0 + slot;
        float2 v[E];
        const float2* w2i = w2;
        asm volatile("" : "+s"(w2i));   // keep the window loads inside the loop (no LICM -> no 32 live VGPRs)
        {
            const float* xc = x + (long long)clip * n_samples;
            const long long s0 = (long long)t * hop - N;
            const bool inside = t < T && s0 >= 0 && s0 + W <= n_samples;
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const int n = p + i * P;
                float2 a, c2;
                if (inside) {
                    a = nt_load2a(xc + s0 + 2 * n);
                    c2 = nt_load2a(xc + s0 + 2 * (n + NH));
                } else {
                    const long long sa = s0 + 2 * n, sc = s0 + 2 * (n + NH);
                    a.x = (t < T && sa >= 0 && sa < n_samples) ? xc[sa] : 0.f;
                    a.y = (t < T && sa + 1 >= 0 && sa + 1 < n_samples) ? xc[sa + 1] : 0.f;
                    c2.x = (t < T && sc >= 0 && sc < n_samples) ? xc[sc] : 0.f;
                    c2.y = (t < T && sc + 1 >= 0 && sc + 1 < n_samples) ? xc[sc + 1] : 0.f;
                }
                const float2 wa = w2i[n], wc = w2i[n + NH];
                const float2 za = make_float2(a.x * wa.x, a.y * wa.y), zc = make_float2(c2.x * wc.x, c2.y * wc.y);
                if (q == 0) v[i] = cadd(za, zc);
                else {
                    const int idx = 2 * n;
                    float2 r = idx <= N / 2 ? tws_l[idx] : tws_l[N - idx];
                    if (idx > N / 2) r = make_float2(-r.x, r.y);
                    v[i] = cmul(csub(za, zc), r);
                }
            }
        }
        fft_frame<9, 3>(v, buf, p, tw_l);
        __syncthreads();
        if (t0 + tt < T) {
            float2* o = out + (long long)clip * W * T + (t0 + tt);
            for (int m = mq; m < NH / 2; m += NT / FPB) {
                float2 xk, xn;
                if (q == 0 && m == 0) {
                    const float2 z0 = fb[0], zc = fb[phys(NH / 2)];
                    if (!(DBG & 1)) {
                    o[0] = make_float2(z0.x + z0.y, 0.f); o[(long long)N * T] = make_float2(z0.x - z0.y, 0.f);
                    o[(long long)(N / 2) * T] = cconj(zc); o[(long long)(N + N / 2) * T] = zc; }
                } else {
                    const int k = 2 * m + q;
                    const int mp = NH - m - q;
                    split_pair(fb[phys(m)], fb[phys(mp)], tws_l[k], xk, xn);
                    if (!(DBG & 1) || xk.x == 12345.f) {
                    o[(long long)k * T] = xk;
                    o[(long long)(W - k) * T] = cconj(xk);
                    o[(long long)(N - k) * T] = xn;
                    o[(long long)(N + k) * T] = cconj(xn); }
                }
            }
        }
        __syncthreads();
    }
}

typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 nt_load2(const float* p) {
    const f32x2_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x2_t*>(p));
    return make_float2(v.x, v.y);
}

// ---- wave-specialised persistent variant: 8 FFT waves + 8 store waves per workgroup.
// Store waves hold the previous tile's packed spectrum in registers (64 VGPRs) and issue its
// 256 KB of stores while the FFT waves transform the next tile into LDS.
template <int DBG>
__global__ __launch_bounds__(1024) void k_stft_w(
    const float* __restrict__ x, const float* __restrict__ win, const float2* __restrict__ twp,
    const float2* __restrict__ tws, float2* __restrict__ out, long long n_samples, int hop, int T, int tiles, int total_tiles) {
    using C = FftCfg<10, 4>;
    constexpr int N = C::N, P = C::P, E = C::E, W = 2 * N, NT = 1024, FPB = 16, PITCH = 1090;
    constexpr int KI = (N / 2) / 32;   // 16 pairs per store thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* frames = reinterpret_cast<float2*>(smem_raw);
    float2* tw_l = frames + FPB * PITCH;
    float2* win_l = tw_l + C::TW;
    float2* tws_l = win_l + N;
    const int tid = threadIdx.x;
    for (int i = tid; i < C::TW; i += NT) tw_l[i] = twp[i];
    for (int i = tid; i < N; i += NT) win_l[i] = reinterpret_cast<const float2*>(win)[i];
    for (int i = tid; i <= N / 2; i += NT) tws_l[i] = tws[i];
    __syncthreads();
    const bool fft_role = tid < 512;
    const int wave = (tid & 511) / P, p = tid % P;
    const int tt = tid % FPB, kq = (tid & 511) / FPB;     // store role: frame tt, bins kq + 32 i
    const float2* fb = frames + tt * PITCH;
    float2 zk[KI], zn[KI];
    int prev_clip = -1, prev_t0 = 0;

    for (int tl = blockIdx.x;; tl += gridDim.x) {
        const bool have = tl < total_tiles;
        const int clip = have ? tl / tiles : 0, tile = have ? tl % tiles : 0;
        const int t0 = tile * FPB;
        if (fft_role) {
            if (have) {
                const float* xc = x + (long long)clip * n_samples;
#pragma unroll 1
                for (int f = 0; f < 2; ++f) {
                    const int t = t0 + wave * 2 + f;
                    const long long s0 = (long long)t * hop - N;
                    float2 v[E];
                    if (t < T && s0 >= 0 && s0 + W <= n_samples) {
#pragma unroll
                        for (int i = 0; i < E; ++i) {
                            const float2 xv = *reinterpret_cast<const float2*>(xc + s0 + 2 * (p + i * P));
                            const float2 wv = win_l[p + i * P];
                            v[i] = make_float2(xv.x * wv.x, xv.y * wv.y);
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < E; ++i) {
                            const long long s = s0 + 2 * (p + i * P);
                            const float2 wv = win_l[p + i * P];
                            const float a = (t < T && s >= 0 && s < n_samples) ? xc[s] : 0.f;
                            const float b = (t < T && s + 1 >= 0 && s + 1 < n_samples) ? xc[s + 1] : 0.f;
                            v[i] = make_float2(a * wv.x, b * wv.y);
                        }
                    }
                    fft_frame<10, 4>(v, frames + (wave * 2 + f) * PITCH, p, tw_l);
                }
            }
        } else if (prev_clip >= 0 && prev_t0 + tt < T) {
            // stores of the PREVIOUS tile, from registers, while the FFT waves work
            float2* o = out + (long long)prev_clip * W * T + (prev_t0 + tt);
#pragma unroll
            for (int i = 0; i < KI; ++i) {
                const int k = kq + 32 * i;
                if (k == 0) {
                    const float2 z0 = zk[i], zc = zn[i];
                    if (!(DBG & 1)) {
                    o[0] = make_float2(z0.x + z0.y, 0.f); o[(long long)N * T] = make_float2(z0.x - z0.y, 0.f);
                    o[(long long)(N / 2) * T] = cconj(zc); o[(long long)(N + N / 2) * T] = zc; }
                } else {
                    float2 xk, xn;
                    split_pair(zk[i], zn[i], tws_l[k], xk, xn);
                    if (!(DBG & 1) || xk.x == 12345.f) {
                    o[(long long)k * T] = xk;
                    o[(long long)(W - k) * T] = cconj(xk);
                    o[(long long)(N - k) * T] = xn;
                    o[(long long)(N + k) * T] = cconj(xn); }
                }
            }
        }
        __syncthreads();   // A: LDS holds tile tl (if any); the store waves have issued tile tl - grid
        if (!have) break;
        if (!fft_role) {
#pragma unroll
            for (int i = 0; i < KI; ++i) {
                const int k = kq + 32 * i;
                zk[i] = k == 0 ? fb[0] : fb[phys(k)];
                zn[i] = k == 0 ? fb[phys(N / 2)] : fb[phys(N - k)];
            }
            prev_clip = clip; prev_t0 = t0;
        }
        __syncthreads();   // B: LDS may be overwritten
    }
}


// ======================================================================================
// Frame-pair packed FFT: a lane holds the SAME element of TWO frames in the two halves of a
// 64-bit register pair (SoA: re pair, im pair).  Every butterfly op is one v_pk_*_f32 on whole
// pairs -- no half swaps, so none of the v_mov_b32 that the (re,im)-pair form needs (500 of them
// per 2 frames in k_stft_ft16); twiddles are shared by the two frames (splat).
// ======================================================================================
typedef float v2f __attribute__((ext_vector_type(2)));
struct C2 { v2f re, im; };
__device__ __forceinline__ C2 c2add(C2 a, C2 b) { return C2{a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ C2 c2sub(C2 a, C2 b) { return C2{a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ C2 c2mi(C2 a) { return C2{a.im, -a.re}; }                       // * (-i)
__device__ __forceinline__ C2 c2mul(C2 a, float wr, float wi) {                          // * (wr + i wi), splat
    return C2{a.re * wr - a.im * wi, a.re * wi + a.im * wr};
}
__device__ __forceinline__ void c2dft4(C2& v0, C2& v1, C2& v2, C2& v3) {
    C2 t0 = c2add(v0, v2), t1 = c2sub(v0, v2), t2 = c2add(v1, v3), t3 = c2mi(c2sub(v1, v3));
    v0 = c2add(t0, t2); v1 = c2add(t1, t3); v2 = c2sub(t0, t2); v3 = c2sub(t1, t3);
}
template <int R> struct Dft2;
template <> struct Dft2<4> { static __device__ __forceinline__ void run(C2* a) { c2dft4(a[0], a[1], a[2], a[3]); } };
template <> struct Dft2<16> {
    static __device__ __forceinline__ void run(C2* a) {
        const float h = 0.70710678118654752440f, c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f;
        C2 m[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { m[r][0] = a[r]; m[r][1] = a[r + 4]; m[r][2] = a[r + 8]; m[r][3] = a[r + 12]; c2dft4(m[r][0], m[r][1], m[r][2], m[r][3]); }
        m[1][1] = c2mul(m[1][1], c1, -s1);
        m[1][2] = c2mul(m[1][2], h, -h);
        m[1][3] = c2mul(m[1][3], s1, -c1);
        m[2][1] = c2mul(m[2][1], h, -h);
        m[2][2] = c2mi(m[2][2]);
        m[2][3] = c2mul(m[2][3], -h, -h);
        m[3][1] = c2mul(m[3][1], s1, -c1);
        m[3][2] = c2mul(m[3][2], -h, -h);
        m[3][3] = c2mul(m[3][3], -c1, s1);
#pragma unroll
        for (int q = 0; q < 4; ++q) { c2dft4(m[0][q], m[1][q], m[2][q], m[3][q]); a[q] = m[0][q]; a[q + 4] = m[1][q]; a[q + 8] = m[2][q]; a[q + 12] = m[3][q]; }
    }
};
template <int LOG2N, int LOG2E, int LOG2NS, int LR>
__device__ __forceinline__ void pass_write2(const C2* v, v2f* rp, v2f* ip, int p, const float2* tw) {
    using C = FftCfg<LOG2N, LOG2E>;
    constexpr int R = 1 << LR, NS = 1 << LOG2NS, NB = C::E / R;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int j = p + b * C::P;
        const int k = j & (NS - 1);
        C2 a[R];
#pragma unroll
        for (int r = 0; r < R; ++r) a[r] = v[b + r * NB];
        if (LOG2NS > 0) {
#pragma unroll
            for (int r = 1; r < R; ++r) { const float2 w = tw[(r - 1) * NS + k]; a[r] = c2mul(a[r], w.x, w.y); }
        }
        Dft2<R>::run(a);
        const int base = ((j >> LOG2NS) << (LOG2NS + LR)) + k;
        const int pb = phys(base);
#pragma unroll
        for (int r = 0; r < R; ++r) { const int o = phys_off<NS * R>(pb, base, r * NS); rp[o] = a[r].re; ip[o] = a[r].im; }
    }
}
template <int LOG2N, int LOG2E>
__device__ __forceinline__ void regs_read2(C2* v, const v2f* rp, const v2f* ip, int p) {
    using C = FftCfg<LOG2N, LOG2E>;
    const int pp = phys(p);
#pragma unroll
    for (int i = 0; i < C::E; ++i) { const int o = phys_off<C::P>(pp, p, i * C::P); v[i].re = rp[o]; v[i].im = ip[o]; }
}
template <int LOG2N, int LOG2E, int LOG2NS = 0>
__device__ __forceinline__ void fft2_frame(C2* v, v2f* rp, v2f* ip, int p, const float2* tw) {
    using C = FftCfg<LOG2N, LOG2E>;
    if constexpr (LOG2NS < LOG2N) {
        constexpr int LR = pass_log2r(LOG2N - LOG2NS, LOG2E);
        pass_write2<LOG2N, LOG2E, LOG2NS, LR>(v, rp, ip, p, tw + twiddle_offset(LOG2N, LOG2E, LOG2NS));
        frame_sync<C::P>();
        if constexpr (LOG2NS + LR < LOG2N) {
            regs_read2<LOG2N, LOG2E>(v, rp, ip, p);
            frame_sync<C::P>();
            fft2_frame<LOG2N, LOG2E, LOG2NS + LR>(v, rp, ip, p, tw);
        }
    }
}

// persistent fat kernel on the frame-pair FFT: 8 waves, wave w owns frames (2w, 2w+1) of the tile
template <int PITCH, int DBG>
__global__ __launch_bounds__(512) void k_stft_q2(
    const float* __restrict__ x, const float* __restrict__ win, const float2* __restrict__ twp,
    const float2* __restrict__ tws, float2* __restrict__ out, long long n_samples, int hop, int T, int tiles, int total_tiles) {
    using C = FftCfg<10, 4>;
    constexpr int N = C::N, P = C::P, E = C::E, W = 2 * N, NT = 512, FPB = 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f* planes = reinterpret_cast<v2f*>(smem_raw);                 // 8 pair buffers x (re plane | im plane) x PITCH
    float2* tw_l = reinterpret_cast<float2*>(planes + 8 * 2 * PITCH);
    float2* win_l = tw_l + C::TW;
    float2* tws_l = win_l + N;
    const int tid = threadIdx.x;
    for (int i = tid; i < C::TW; i += NT) tw_l[i] = twp[i];
    for (int i = tid; i < N; i += NT) win_l[i] = reinterpret_cast<const float2*>(win)[i];
    for (int i = tid; i <= N / 2; i += NT) tws_l[i] = tws[i];
    __syncthreads();
    const int wave = tid / P, p = tid % P;
    const int tt = tid % FPB, kq = tid / FPB;
    v2f* my_re = planes + (size_t)wave * 2 * PITCH;
    v2f* my_im = my_re + PITCH;
    const float* st_re = reinterpret_cast<const float*>(planes + (size_t)(tt >> 1) * 2 * PITCH) + (tt & 1);
    const float* st_im = st_re + 2 * PITCH;

    float2 xr[2][E];
    auto prefetch = [&](int tl) {
        if (tl >= total_tiles) return;
        const int clip = tl / tiles, tile = tl % tiles;
        const float* xc = x + (long long)clip * n_samples;
        const long long first = (long long)tile * FPB * hop - N;
        const long long last = first + (long long)(FPB - 1) * hop + W;
        if (first >= 0 && last <= n_samples && tile * FPB + FPB <= T) {
            const float* src = xc + first + (long long)(wave * 2) * hop + 2 * p;
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int i = 0; i < E; ++i) xr[f][i] = *reinterpret_cast<const float2*>(src + (long long)f * hop + 2 * i * P);
        } else {
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const int t = tile * FPB + wave * 2 + f;
                const long long s0 = (long long)t * hop - N;
#pragma unroll
                for (int i = 0; i < E; ++i) {
                    const long long sidx = s0 + 2 * (p + i * P);
                    xr[f][i].x = (t < T && sidx >= 0 && sidx < n_samples) ? xc[sidx] : 0.f;
                    xr[f][i].y = (t < T && sidx + 1 >= 0 && sidx + 1 < n_samples) ? xc[sidx + 1] : 0.f;
                }
            }
        }
    };
    int tl = blockIdx.x;
    prefetch(tl);
    for (; tl < total_tiles; tl += gridDim.x) {
        const int clip = tl / tiles, tile = tl % tiles;
        const int t0 = tile * FPB;
        {
            C2 v[E];
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const float2 wv = win_l[p + i * P];
                v[i].re = v2f{xr[0][i].x * wv.x, xr[1][i].x * wv.x};
                v[i].im = v2f{xr[0][i].y * wv.y, xr[1][i].y * wv.y};
            }
            fft2_frame<10, 4>(v, my_re, my_im, p, tw_l);
        }
        __syncthreads();
        prefetch(tl + gridDim.x);
        if (t0 + tt < T) {
            float2* o = out + (long long)clip * W * T + (t0 + tt);
            for (int k = kq; k < N / 2; k += NT / FPB) {
                float2 xk, xn;
                if (k == 0) {
                    const float2 z0 = make_float2(st_re[0], st_im[0]);
                    const float2 zc = make_float2(st_re[2 * phys(N / 2)], st_im[2 * phys(N / 2)]);
                    if (!(DBG & 1)) {
                    o[0] = make_float2(z0.x + z0.y, 0.f); o[(long long)N * T] = make_float2(z0.x - z0.y, 0.f);
                    o[(long long)(N / 2) * T] = cconj(zc); o[(long long)(N + N / 2) * T] = zc; }
                } else {
                    const int a = 2 * phys(k), b = 2 * phys(N - k);
                    split_pair(make_float2(st_re[a], st_im[a]), make_float2(st_re[b], st_im[b]), tws_l[k], xk, xn);
                    if (!(DBG & 1) || xk.x == 12345.f) {
                    o[(long long)k * T] = xk;
                    o[(long long)(W - k) * T] = cconj(xk);
                    o[(long long)(N - k) * T] = xn;
                    o[(long long)(N + k) * T] = cconj(xn); }
                }
            }
        }
        __syncthreads();
    }
}

static bool selected(const char* name) {
    const char* sel = getenv("SEL");
    return !sel || strstr(name, sel) != nullptr;
}
struct Ctx { float *x, *win; float2 *twp, *tws, *out; long long n; int hop, T, B; };

template <int FPB, int DBG, int REMAP>
float run(const Ctx& c, const char* name, int reps = 10) {
    if (!selected(name)) return 0;
    using C = FftCfg<10, 4>;
    auto kern = k_stft_v<FPB, DBG, REMAP>;
    size_t smem = (size_t)(FPB * C::PITCH + C::TW) * 8;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    int tiles = (c.T + FPB - 1) / FPB;
    int blocks = tiles * c.B;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(FPB * 64), smem, 0, c.x, c.win, c.twp, c.tws, c.out, c.n, c.hop, c.T, tiles, 12345.678f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(FPB * 64), smem, 0, c.x, c.win, c.twp, c.tws, c.out, c.n, c.hop, c.T, tiles, 12345.678f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    double bytes = (double)c.B * (4.0 * c.n + 8.0 * 2048 * c.T);
    int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, FPB * 64, smem));
    printf("%-28s FPB=%2d blocks=%6d occ=%d WG/CU  %.3f ms  %.0f GB/s (alg)\n", name, FPB, blocks, occ, ms, bytes / ms / 1e6);
    return ms;
}

template <int FPB, int PITCH, int DBG>
float runp(const Ctx& c, const char* name, int wg_per_cu = 1, int reps = 10) {
    if (!selected(name)) return 0;
    using C = FftCfg<10, 4>;
    auto kern = k_stft_p<FPB, PITCH, DBG>;
    size_t smem = (size_t)(FPB * PITCH + C::TW + 1024 + 513) * 8;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    int tiles = (c.T + FPB - 1) / FPB;
    int total = tiles * c.B;
    int blocks = 256 * wg_per_cu; if (blocks > total) blocks = total;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(FPB * 64), smem, 0, c.x, c.win, c.twp, c.tws, c.out, c.n, c.hop, c.T, tiles, total);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(FPB * 64), smem, 0, c.x, c.win, c.twp, c.tws, c.out, c.n, c.hop, c.T, tiles, total);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    double bytes = (double)c.B * (4.0 * c.n + 8.0 * 2048 * c.T);
    int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, FPB * 64, smem));
    printf("%-28s FPB=%2d smem=%zu blocks=%6d occ=%d  %.3f ms  %.0f GB/s (alg)\n", name, FPB, smem, blocks, occ, ms, bytes / ms / 1e6);
    return ms;
}

template <int WAVES, int PITCH, int DBG, int VAR = 0>
float runq(const Ctx& c, const char* name, int reps = 10) {
    if (!selected(name)) return 0;
    using C = FftCfg<10, 4>;
    auto kern = VAR == 0 ? k_stft_q<WAVES, PITCH, DBG> : (VAR == 2 ? k_stft_q<WAVES, PITCH, DBG, 1> : k_stft_r<WAVES, PITCH, DBG>);
    size_t smem = (size_t)(16 * PITCH + C::TW + 1024 + 513) * 8;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    int tiles = (c.T + 15) / 16;
    int total = tiles * c.B;
    int blocks = 256; if (blocks > total) blocks = total;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned long long* prof; CK(hipMalloc(&prof, 64));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(WAVES * 64), smem, 0, c.x, c.win, c.twp, c.tws, c.out, c.n, c.hop, c.T, tiles, total, prof);
    CK(hipDeviceSynchronize());
    CK(hipMemset(prof, 0, 64));
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(WAVES * 64), smem, 0, c.x, c.win, c.twp, c.tws, c.out, c.n, c.hop, c.T, tiles, total, prof);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    double bytes = (double)c.B * (4.0 * c.n + 8.0 * 2048 * c.T);
    printf("%-28s WAVES=%2d smem=%zu blocks=%6d  %.3f ms  %.0f GB/s (alg)\n", name, WAVES, smem, blocks, ms, bytes / ms / 1e6);
    if (DBG & 8) {
        unsigned long long h[8]; CK(hipMemcpy(h, prof, 64, hipMemcpyDeviceToHost));
        double nt = (double)h[5];
        printf("   per-wave per-tile cycles: fft(+win) %.0f | barrier-wait %.0f | prefetch-issue %.0f | store-issue %.0f | end-barrier %.0f\n",
               h[0] / nt, h[1] / nt, h[2] / nt, h[3] / nt, h[4] / nt);
    }
    return ms;
}

template <int DBG>
float runps(const Ctx& c, const float2* twp9, const char* name, int reps = 10) {
    if (!selected(name)) return 0;
    using C = FftCfg<9, 3>;
    auto kern = k_stft_ps<DBG>;
    size_t smem = (size_t)(16 * (512 + 32 + 2) + C::TW + 513) * 8;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    int tiles = (c.T + 15) / 16;
    int blocks = tiles * c.B * 2;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned long long* prof; CK(hipMalloc(&prof, 64)); CK(hipMemset(prof, 0, 64));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), smem, 0, c.x, c.win, twp9, c.tws, c.out, c.n, c.hop, c.T, tiles, prof);
    CK(hipDeviceSynchronize());
    CK(hipMemset(prof, 0, 64));
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), smem, 0, c.x, c.win, twp9, c.tws, c.out, c.n, c.hop, c.T, tiles, prof);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    double bytes = (double)c.B * (4.0 * c.n + 8.0 * 2048 * c.T);
    int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 1024, smem));
    printf("%-28s smem=%zu blocks=%6d occ=%d  %.3f ms  %.0f GB/s (alg)\n", name, smem, blocks, occ, ms, bytes / ms / 1e6);
    if (DBG & 8) {
        unsigned long long h[8]; CK(hipMemcpy(h, prof, 64, hipMemcpyDeviceToHost));
        double nb = (double)h[5];
        printf("   per-WG cycles (wave 0): tables %.0f | load-wait %.0f | prestage+fft+sync %.0f | store-issue %.0f | store-drain %.0f   (n=%.0f)\n",
               h[0] / nb, h[1] / nb, h[2] / nb, h[3] / nb, h[4] / nb, nb);
    }
    return ms;
}

template <int DBG>
float runpp(const Ctx& c, const float2* twp9, const char* name, int reps = 10) {
    if (!selected(name)) return 0;
    using C = FftCfg<9, 3>;
    auto kern = k_stft_pp<DBG>;
    size_t smem = (size_t)(16 * (512 + 32 + 2) + C::TW + 513) * 8;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    int tiles = (c.T + 15) / 16;
    int total = tiles * c.B;
    int blocks = 512;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), smem, 0, c.x, c.win, twp9, c.tws, c.out, c.n, c.hop, c.T, tiles, total);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), smem, 0, c.x, c.win, twp9, c.tws, c.out, c.n, c.hop, c.T, tiles, total);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    double bytes = (double)c.B * (4.0 * c.n + 8.0 * 2048 * c.T);
    int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 1024, smem));
    printf("%-28s smem=%zu blocks=%6d occ=%d  %.3f ms  %.0f GB/s (alg)\n", name, smem, blocks, occ, ms, bytes / ms / 1e6);
    return ms;
}

template <int DBG>
float runw(const Ctx& c, const char* name, int reps = 10) {
    if (!selected(name)) return 0;
    using C = FftCfg<10, 4>;
    auto kern = k_stft_w<DBG>;
    size_t smem = (size_t)(16 * 1090 + C::TW + 1024 + 513) * 8;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    int tiles = (c.T + 15) / 16;
    int total = tiles * c.B;
    int blocks = 256;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), smem, 0, c.x, c.win, c.twp, c.tws, c.out, c.n, c.hop, c.T, tiles, total);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), smem, 0, c.x, c.win, c.twp, c.tws, c.out, c.n, c.hop, c.T, tiles, total);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    double bytes = (double)c.B * (4.0 * c.n + 8.0 * 2048 * c.T);
    printf("%-28s smem=%zu blocks=%6d  %.3f ms  %.0f GB/s (alg)\n", name, smem, blocks, ms, bytes / ms / 1e6);
    return ms;
}

template <int PITCH, int DBG>
float runq2(const Ctx& c, const char* name, int reps = 10) {
    if (!selected(name)) return 0;
    using C = FftCfg<10, 4>;
    auto kern = k_stft_q2<PITCH, DBG>;
    size_t smem = (size_t)(16 * PITCH + C::TW + 1024 + 513) * 8;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    int tiles = (c.T + 15) / 16;
    int total = tiles * c.B;
    int blocks = 256;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), smem, 0, c.x, c.win, c.twp, c.tws, c.out, c.n, c.hop, c.T, tiles, total);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), smem, 0, c.x, c.win, c.twp, c.tws, c.out, c.n, c.hop, c.T, tiles, total);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    double bytes = (double)c.B * (4.0 * c.n + 8.0 * 2048 * c.T);
    printf("%-28s smem=%zu blocks=%6d  %.3f ms  %.0f GB/s (alg)\n", name, smem, blocks, ms, bytes / ms / 1e6);
    return ms;
}

double checksum(const Ctx& c) {
    size_t n = (size_t)2048 * c.T * 4;   // first 4 clips
    std::vector<float2> h(n);
    CK(hipMemcpy(h.data(), c.out, n * sizeof(float2), hipMemcpyDeviceToHost));
    double s = 0; for (size_t i = 0; i < n; ++i) s += (double)h[i].x * ((i % 97) + 1) + (double)h[i].y * ((i % 89) + 1);
    return s;
}

std::vector<float2> snapshot(const Ctx& c) {
    size_t n = (size_t)2048 * c.T * 4;
    std::vector<float2> h(n);
    CK(hipMemcpy(h.data(), c.out, n * sizeof(float2), hipMemcpyDeviceToHost));
    return h;
}
double maxdiff(const std::vector<float2>& a, const std::vector<float2>& b) {
    double d = 0, m = 0;
    for (size_t i = 0; i < a.size(); ++i) { d = std::max(d, (double)std::hypot(a[i].x - b[i].x, a[i].y - b[i].y)); m = std::max(m, (double)std::hypot(a[i].x, a[i].y)); }
    return d / m;
}
int main() {
    Ctx c; c.B = 1024; c.n = 441000; c.hop = 1024; c.T = 432;
    const int W = 2048, N = 1024;
    std::vector<float> hx((size_t)c.B * c.n), hw(W);
    srand(1);
    for (size_t i = 0; i < (size_t)8 * c.n; ++i) hx[i] = (rand() / (float)RAND_MAX - 0.5f) * 3.4f;
    for (size_t i = (size_t)8 * c.n; i < hx.size(); ++i) hx[i] = hx[i - (size_t)8 * c.n];
    for (int i = 0; i < W; ++i) hw[i] = 0.54f - 0.46f * cosf(2 * M_PI * i / W);
    auto twp = build_pass_twiddles(10, 4);
    std::vector<cf32> tws(N / 2 + 1);
    for (int k = 0; k <= N / 2; ++k) tws[k] = unit_root(k, W);
    CK(hipMalloc(&c.x, hx.size() * 4)); CK(hipMemcpy(c.x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&c.win, W * 4)); CK(hipMemcpy(c.win, hw.data(), W * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&c.twp, twp.size() * 8)); CK(hipMemcpy(c.twp, twp.data(), twp.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&c.tws, tws.size() * 8)); CK(hipMemcpy(c.tws, tws.data(), tws.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&c.out, (size_t)c.B * W * c.T * 8));
    CK(hipMemset(c.out, 0, (size_t)c.B * W * c.T * 8));
    run<16, 0, 0>(c, "base FPB16"); double cs0 = checksum(c); auto ref = snapshot(c);
    CK(hipMemset(c.out, 0, (size_t)c.B * W * c.T * 8));
    runq2<1090, 0>(c, "pairfft"); if (selected("pairfft")) printf("  checksum match: %d  rel diff %.3e\n", checksum(c) == cs0, maxdiff(snapshot(c), ref));
    runq2<1090, 1>(c, "pairfft no-store");
    CK(hipMemset(c.out, 0, (size_t)c.B * W * c.T * 8));
    runw<0>(c, "wavespec"); if (selected("wavespec")) printf("  checksum match: %d  rel diff %.3e\n", checksum(c) == cs0, maxdiff(snapshot(c), ref));
    runw<1>(c, "wavespec no-store");
    {
        auto twp9 = build_pass_twiddles(9, 3);
        float2* d9; CK(hipMalloc(&d9, twp9.size() * 8)); CK(hipMemcpy(d9, twp9.data(), twp9.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemset(c.out, 0, (size_t)c.B * W * c.T * 8));
        runps<0>(c, d9, "parity-split"); if (selected("parity-split")) printf("  rel diff vs base: %.3e\n", maxdiff(snapshot(c), ref));
        runps<1>(c, d9, "parity-split no-store");
        CK(hipMemset(c.out, 0, (size_t)c.B * W * c.T * 8));
        runpp<0>(c, d9, "pparity persistent"); if (selected("pparity persistent")) printf("  rel diff vs base: %.3e\n", maxdiff(snapshot(c), ref));
        runpp<1>(c, d9, "pparity persistent no-store");
        runps<8>(c, d9, "parity-split timed");
        runps<9>(c, d9, "parity-split timed no-store");
        runps<16>(c, d9, "parity-split nowin");
        runps<17>(c, d9, "parity-split nowin no-store");
        runps<24>(c, d9, "parity-split nowin timed");
    }
    run<8, 0, 0>(c, "FPB8"); printf("  checksum match: %d\n", checksum(c) == cs0);
    run<8, 0, 1>(c, "FPB8 xcd-pair remap"); printf("  checksum match: %d\n", checksum(c) == cs0);
    run<4, 0, 0>(c, "FPB4"); printf("  checksum match: %d\n", checksum(c) == cs0);
    CK(hipMemset(c.out, 0, (size_t)c.B * W * c.T * 8));
    runp<16, 1089, 0>(c, "persist pitch1089"); printf("  checksum match: %d\n", checksum(c) == cs0);
    CK(hipMemset(c.out, 0, (size_t)c.B * W * c.T * 8));
    runp<16, 1090, 0>(c, "persist pitch1090"); printf("  checksum match: %d\n", checksum(c) == cs0);
    runp<16, 1090, 1>(c, "persist pitch1090 no-store");
    CK(hipMemset(c.out, 0, (size_t)c.B * W * c.T * 8));
    runq<8, 1090, 0, 2>(c, "persist-qmf 8 waves"); if (selected("persist-qmf 8 waves")) printf("  checksum match: %d\n", checksum(c) == cs0);
    runq<8, 1090, 1, 2>(c, "persist-qmf 8 waves no-store");
    runq<4, 1090, 0, 2>(c, "persist-qmf 4 waves");
    runq<16, 1090, 128>(c, "persist-q 16 waves nolicm");
    runq<8, 1090, 128>(c, "persist-q 8 waves nolicm");
    runq<8, 1090, 8>(c, "persist-q 8 waves timed");
    runq<8, 1090, 0>(c, "persist-q 8 waves"); printf("  checksum match: %d\n", checksum(c) == cs0);
    CK(hipMemset(c.out, 0, (size_t)c.B * W * c.T * 8));
    runq<4, 1090, 0>(c, "persist-q 4 waves"); printf("  checksum match: %d\n", checksum(c) == cs0);
    CK(hipMemset(c.out, 0, (size_t)c.B * W * c.T * 8));
    runq<8, 1090, 0, 1>(c, "persist-r 8 waves"); printf("  checksum match: %d\n", checksum(c) == cs0);
    CK(hipMemset(c.out, 0, (size_t)c.B * W * c.T * 8));
    runq<4, 1090, 0, 1>(c, "persist-r 4 waves"); printf("  checksum match: %d\n", checksum(c) == cs0);
    CK(hipMemset(c.out, 0, (size_t)c.B * W * c.T * 8));
    runq<16, 1090, 0, 1>(c, "persist-r 16 waves"); printf("  checksum match: %d\n", checksum(c) == cs0);
    runq<8, 1090, 1, 1>(c, "persist-r 8 waves no-store");
    runq<16, 1090, 0>(c, "persist-q 16 waves");
    runq<8, 1090, 1>(c, "persist-q 8 waves no-store");
    run<16, 1, 0>(c, "FPB16 no-store");
    run<16, 2, 0>(c, "FPB16 no-load");
    run<16, 4, 0>(c, "FPB16 no-fft");
    run<16, 5, 0>(c, "FPB16 no-fft no-store");
    run<16, 3, 0>(c, "FPB16 no-load no-store (fft only)");
    run<8, 1, 0>(c, "FPB8 no-store");
    run<8, 4, 0>(c, "FPB8 no-fft");
    return 0;
}
