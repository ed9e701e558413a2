// zafx_stft.hip -- batched STFT / ISTFT kernels for gfx950 (MI355X).
//
//   k_stft  : framing + window + real-input FFT + Hermitian mirror, fused
//             (replaces np.pad zaf.py:112-125, the frame loop :132-136 and
//             np.fft.fft(axis=0) :139)
//   k_istft : Hermitian symmetrisation + inverse real FFT + gather overlap-add +
//             trim + COLA gain, fused (replaces zaf.py:223, :226-233, :236-241)
//
// A frame of W real samples is transformed as ONE complex FFT of N = W/2 points
// (z[n] = x[2n] + i x[2n+1]) followed by the real-split butterfly
//   X[k]   = E + t_k O,  X[N-k] = conj(E - t_k O),  t_k = exp(-2 pi i k / W)
//   E = (Z[k] + conj Z[N-k]) / 2,  O = -i (Z[k] - conj Z[N-k]) / 2,
// and the upper half is written as the conjugate mirror (the reference API returns
// the two-sided spectrum).  A workgroup owns FPB consecutive frames of one clip;
// in the reference (frequency-major, time-minor) layout the FPB frames supply the
// contiguous run along t for every stored row (16 frames x 8 B = one 128-B line).
#include <algorithm>
#include <type_traits>

#include "zafx_fft.hpp"
#include "zafx_internal.hpp"

namespace zafx {

// Output rows of the forward kernels by spectrum kind SPEC (enum zafx_spectrum): 0 two-sided complex,
// 1 one-sided complex, 2 one-sided |X| and 3 one-sided |X|^2 as float32 (what the examples of the reference
// compute from the result, zaf.py:83).  `spec_base` offsets the output in ELEMENTS of the kind; `put_bin`
// stores element idx relative to it.
template <int SPEC>
__device__ __forceinline__ float2* spec_base(float2* out, long long off) {
    if constexpr (SPEC >= 2) return reinterpret_cast<float2*>(reinterpret_cast<float*>(out) + off);
    else return out + off;
}
// STREAM: non-temporal store of the complex kinds (the reference-layout kernel writes each 128-B line exactly once, from
// one instruction); the 64-B runs of the real kinds do not gain from it.
template <int SPEC, bool STREAM = false>
__device__ __forceinline__ void put_bin(float2* o, long long idx, float2 v) {
    if constexpr (SPEC >= 2) {
        const float pw = v.x * v.x + v.y * v.y;
        reinterpret_cast<float*>(o)[idx] = SPEC == 2 ? __builtin_amdgcn_sqrtf(pw) : pw;
    } else if constexpr (STREAM) {
        store_stream(o + idx, v);
    } else {
        o[idx] = v;
    }
}

template <int LOG2N, int LOG2E, int FPB>
struct StftCfg {
    using C = FftCfg<LOG2N, LOG2E>;
    static constexpr int NT = FPB * C::P;
    static constexpr bool TW_LDS = (size_t)(FPB * C::PITCH + C::TW) * 8 <= (size_t)kMaxLdsBytes;
    static constexpr size_t SMEM = (size_t)(FPB * C::PITCH + (TW_LDS ? C::TW : 0)) * 8;
};

// ---------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------
// SPEC = enum zafx_spectrum: 0 two-sided; 1..3 one-sided (rows 0..W/2 only, the mirror is not written).
template <int LOG2N, int LOG2E, int FPB, int LAYOUT, int SPEC>
__global__ __launch_bounds__(FPB * fft_threads(LOG2N, LOG2E)) void k_stft(
    const float* __restrict__ x, const float* __restrict__ win, const float2* __restrict__ twp,
    const float2* __restrict__ tws, float2* __restrict__ out, long long n_samples, int hop, int T, int TP, int tiles) {
    using C = FftCfg<LOG2N, LOG2E>;
    using S = StftCfg<LOG2N, LOG2E, FPB>;
    constexpr int N = C::N, P = C::P, E = C::E, W = 2 * N, NT = S::NT, ROWS = SPEC ? N + 1 : W;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* frames = reinterpret_cast<float2*>(smem_raw);
    const float2* tw = twp;
    const int tid = threadIdx.x;
    if constexpr (S::TW_LDS) {
        float2* tw_l = frames + FPB * C::PITCH;
        for (int i = tid; i < C::TW; i += NT) tw_l[i] = twp[i];
        tw = tw_l;
        __syncthreads();
    }
    const int slot = tid / P, p = tid % P;
    const int bid = ZAFX_XCD_ORDER ? xcd_order((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;   // neighbouring tiles to one XCD: their partial lines merge in its L2
    const int clip = bid / tiles, tile = bid % tiles;
    const int t0 = tile * FPB;
    const int t = t0 + slot;
    float2* buf = frames + slot * C::PITCH;

    // ---- framing + window: v[i] = (x[2n] w[2n], x[2n+1] w[2n+1]), n = p + i P (zaf.py:112-136)
    float2 v[E];
    {
        const float* xc = x + (long long)clip * n_samples;
        const long long s0 = (long long)t * hop - N;   // floor(W/2) = N samples of left padding
        const float2* w2 = reinterpret_cast<const float2*>(win);
        if (t < T && s0 >= 0 && s0 + W <= n_samples && reinterpret_cast<uintptr_t>(xc + s0) % 8 == 0) {
            // interior frame: unconditional 8-byte loads (uniform per frame)
            const float2* x2 = reinterpret_cast<const float2*>(xc + s0);
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const int n = p + i * P;
                const float2 xv = x2[n], wv = w2[n];
                v[i] = make_float2(xv.x * wv.x, xv.y * wv.y);
            }
        } else {
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const int n = p + i * P;
                const long long s = s0 + 2 * n;
                const float2 wv = w2[n];
                const float a = (t < T && s >= 0 && s < n_samples) ? xc[s] : 0.f;
                const float b = (t < T && s + 1 >= 0 && s + 1 < n_samples) ? xc[s + 1] : 0.f;
                v[i] = make_float2(a * wv.x, b * wv.y);
            }
        }
    }
    fft_frame<LOG2N, LOG2E>(v, buf, p, tw);

    if constexpr (LAYOUT == ZAFX_LAYOUT_TF) {
        if (t >= T) return;
        float2* o = spec_base<SPEC>(out, ((long long)clip * T + t) * ROWS);
#pragma unroll
        for (int i = 0; i < E / 2; ++i) {
            const int k = p + i * P;
            if (k == 0) {
                const float2 z0 = buf[0], zc = buf[phys(N / 2)];
                put_bin<SPEC>(o, 0, make_float2(z0.x + z0.y, 0.f));
                put_bin<SPEC>(o, N, make_float2(z0.x - z0.y, 0.f));
                put_bin<SPEC>(o, N / 2, cconj(zc));
                if (SPEC == 0) put_bin<SPEC>(o, N + N / 2, zc);
            } else {
                float2 xk, xn;
                split_pair(buf[phys(k)], buf[phys(N - k)], tws[k], xk, xn);
                put_bin<SPEC>(o, k, xk);
                put_bin<SPEC>(o, N - k, xn);
                if (SPEC == 0) {
                    put_bin<SPEC>(o, W - k, cconj(xk));
                    put_bin<SPEC>(o, N + k, cconj(xn));
                }
            }
        }
    } else {
        if constexpr (NT > 64) __syncthreads();
        const int tt = tid % FPB, kq = tid / FPB;
        if (t0 + tt >= T) return;
        const float2* fb = frames + tt * C::PITCH;
        float2* o = spec_base<SPEC>(out, (long long)clip * ROWS * TP + (t0 + tt));
        for (int k = kq; k < N / 2; k += P) {
            if (k == 0) {
                const float2 z0 = fb[0], zc = fb[phys(N / 2)];
                put_bin<SPEC>(o, 0, make_float2(z0.x + z0.y, 0.f));
                put_bin<SPEC>(o, (long long)N * TP, make_float2(z0.x - z0.y, 0.f));
                put_bin<SPEC>(o, (long long)(N / 2) * TP, cconj(zc));
                if (SPEC == 0) put_bin<SPEC>(o, (long long)(N + N / 2) * TP, zc);
            } else {
                float2 xk, xn;
                split_pair(fb[phys(k)], fb[phys(N - k)], tws[k], xk, xn);
                put_bin<SPEC>(o, (long long)k * TP, xk);
                put_bin<SPEC>(o, (long long)(N - k) * TP, xn);
                if (SPEC == 0) {
                    put_bin<SPEC>(o, (long long)(W - k) * TP, cconj(xk));
                    put_bin<SPEC>(o, (long long)(N + k) * TP, cconj(xn));
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// forward, reference (frequency-major) layout, persistent "fat wave" form
// ---------------------------------------------------------------------------------
// Measured on MI355X (profiles/r01_notes.md): with one 16-wave workgroup per CU the
// load, FFT and store phases of a tile run back to back and every tile pays the table
// staging again.  This kernel keeps the 16-frame tile (128-B runs along t) but
//   * is persistent: one workgroup per CU loops over tiles, so window, pass twiddles and
//     split roots are staged in LDS once;
//   * uses 8 wavefronts that each transform 2 frames (up to 256 VGPRs per lane), so the
//     NEXT tile's samples are prefetched into registers before the current tile's stores
//     are issued -- the load latency hides under the store phase without spilling;
//   * loads interior frames with unconditional 8-byte loads (no per-sample bounds
//     branches); only the clip-edge frames take the predicated path;
//   * pads the frame pitch to 2 (mod 32) complex slots: the transposed read of the store
//     phase (16 frames x 2 bins per 32-lane group) is then LDS-bank-conflict free.
ZAFX_PROF_ARRAY(g_prof_stft)
constexpr int kFatWaves = 8;
constexpr int kFatFrames = 16;

// FPB_ = frames per tile: 16 (128-byte runs along t), or 8 for W = 4096, where 16 frames do not fit LDS -- 64-byte runs, but
// still the persistent schedule (tables staged once, the next tile requested ahead of the barrier) instead of one workgroup per
// tile; the window (16 KB) is then read from global memory.
template <int LOG2N, int LOG2E, int FPB_ = kFatFrames>
struct FatCfg {
    using C = FftCfg<LOG2N, LOG2E>;
    static_assert(C::P == 64 || C::P == 32, "a frame is owned by one wavefront, or by half of one (32 points per thread)");
    static constexpr int N = C::N;
    static constexpr int FPB = FPB_;
    static constexpr bool WIN_LDS = FPB_ == kFatFrames;
    // frame pitch = 2 (mod 32) complex for 16 frames, 4 (mod 32) for 8: the transposed read of the store phase (FPB frames x 32 / FPB
    // bins per 32-lane group) then meets 64 distinct banks
    static constexpr int PITCH = ((N + (N >> C::PS) + 31) / 32) * 32 + 32 / FPB_;
    static constexpr int NT = kFatWaves * 64;
    static constexpr int FPW = FPB_ * C::P / NT;   // frames a group of P lanes transforms per tile: 2, or 1 (P = 32; 8-frame tiles)
    static constexpr size_t SMEM = (size_t)(FPB_ * PITCH + C::TW + (WIN_LDS ? N : 0) + N / 2 + 1) * 8;
};

// int16 PCM in the loads (zafx_execute_pcm): a point of the packed frame -- two samples -- as the kernel keeps it until the window multiply.
// PCM 0: float2.  1: int16 mono, the two samples in one dword (held as a float's bits).  2: int16 stereo, two (left, right) frames in two
// dwords: the channels are added on the way out; the 2^-15 / 2^-16 of zaf.py:1202 and :65 rides in the window the kernel stages.
template <int PCM>
struct PcmPoint {
    using T = std::conditional_t<PCM == 1, float, float2>;
    static __device__ __forceinline__ float2 get(T v) {
        if constexpr (PCM == 1) {
            const int d = __builtin_bit_cast(int, v);
            return make_float2((float)(short)(d & 0xffff), (float)(d >> 16));
        } else if constexpr (PCM == 2) {
            const int a = __builtin_bit_cast(int, v.x), b = __builtin_bit_cast(int, v.y);
            return make_float2((float)((int)(short)(a & 0xffff) + (a >> 16)), (float)((int)(short)(b & 0xffff) + (b >> 16)));
        } else {
            return v;
        }
    }
    static constexpr float scale() { return PCM == 1 ? 1.f / 32768.f : PCM == 2 ? 1.f / 65536.f : 1.f; }
};

// Opaque per-tile lane indices also below 32 points per thread: without them the compiler carries the thread's window and twiddle values across the tiles
// (W = 1024: 224 registers, one workgroup per CU where LDS admits two; with them 71-81).  Measured, 1024 clips x 10 s at hop W / 2 on padded rows: W = 512
// 2.01 -> 1.87 ms, W = 1024 one-sided 1.21 -> 1.16, |X| 1.08 -> 1.07 (compact rows: 1.15 -> 1.28), W = 256 2.13 -> 2.16: on for W = 512 and the complex kinds of W = 1024.
#ifndef ZAFX_STFT_FAT_OPAQUE
#define ZAFX_STFT_FAT_OPAQUE(log2n, spec) ((log2n) == 8 || ((log2n) == 9 && (spec) <= 1))
#endif
template <int LOG2N, int LOG2E, bool ALIGNED, int SPEC, int FPB_ = kFatFrames, int PCM = 0>
__global__ __launch_bounds__(kFatWaves * 64) void k_stft_ft16(
    const float* __restrict__ x, const float* __restrict__ win, const float2* __restrict__ twp,
    const float2* __restrict__ tws, float2* __restrict__ out, long long n_samples, int hop, int T, int TP, int tiles,
    int total_tiles) {
    using C = FftCfg<LOG2N, LOG2E>;
    using F = FatCfg<LOG2N, LOG2E, FPB_>;
    constexpr int N = C::N, P = C::P, E = C::E, W = 2 * N, NT = F::NT, FPB = F::FPB, FPW = F::FPW, PITCH = F::PITCH;
    static_assert(FPW >= 1, "every group of P lanes transforms at least one frame per tile");
    constexpr int ROWS = SPEC ? N + 1 : W;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* frames = reinterpret_cast<float2*>(smem_raw);
    float2* tw_l = frames + FPB * PITCH;
    float2* win_s = tw_l + C::TW;   // N float2 = W window samples (16-frame tiles)
    float2* tws_l = win_s + (F::WIN_LDS ? N : 0);      // N/2 + 1 roots of W
    const float2* win_l = F::WIN_LDS ? win_s : reinterpret_cast<const float2*>(win);
    const int tid = threadIdx.x;
    for (int i = tid; i < C::TW; i += NT) tw_l[i] = twp[i];
    static_assert(PCM == 0 || F::WIN_LDS, "int16 input: the staged window carries the scale");
    using PP = PcmPoint<PCM>;
    if constexpr (F::WIN_LDS)
        for (int i = tid; i < N; i += NT) {
            const float2 wv = reinterpret_cast<const float2*>(win)[i];
            win_s[i] = make_float2(wv.x * PP::scale(), wv.y * PP::scale());
        }
    for (int i = tid; i <= N / 2; i += NT) tws_l[i] = tws[i];
    __syncthreads();
    const int wave = tid / P, p_lane = tid % P, p = p_lane;
    const int tt = tid % FPB, kq = tid / FPB;
    const float2* fb = frames + tt * PITCH;
    const bool lines_whole = TP % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 128 == 0;

    typename PP::T xr[FPW][E];
    // The fast / edge decision is taken once per TILE (block-uniform): a tile whose 16 frames all lie
    // inside the clip issues 2 x 16 unconditional 8-byte loads with no control flow in between, so the
    // loads stay in flight across the store phase; only the first and last tiles of a clip take the
    // predicated path (zero padding of zaf.py:112-125).
    const bool xcd = ZAFX_XCD_ORDER && gridDim.x % 8 == 0;   // (see xcd_order: units of a clip to the workgroups of one XCD)
    auto prefetch = [&](int tlv) {
        if (tlv >= total_tiles) return;
        const int tl = xcd ? xcd_order(tlv, total_tiles) : tlv;
        int p = p_lane;   // (opaque at 32 points per thread: the 64-bit sample offsets of the edge path are recomputed, not hoisted)
        if constexpr (E >= 32 || ZAFX_STFT_FAT_OPAQUE(LOG2N, SPEC)) asm volatile("" : "+v"(p));
        const int clip = tl / tiles, tile = tl % tiles;
        const float* xc = x + (long long)clip * n_samples;   // (PCM 2: a "sample" is one 4-byte frame of two int16)
        const long long first = (long long)tile * FPB * hop - N;               // first sample of the tile
        const long long last = first + (long long)(FPB - 1) * hop + W;          // one past its last sample
        if constexpr (PCM == 1) {   // int16 mono: a point is 4 bytes
            const short* xs = reinterpret_cast<const short*>(x) + (long long)clip * n_samples;
            if (ALIGNED && first >= 0 && last <= n_samples && tile * FPB + FPB <= T) {
                const short* src = xs + first + (long long)(wave * FPW) * hop + 2 * p;
#pragma unroll
                for (int f = 0; f < FPW; ++f) {
#pragma unroll
                    for (int i = 0; i < E; ++i) xr[f][i] = *reinterpret_cast<const float*>(src + (long long)f * hop + 2 * i * P);
                }
            } else {
#pragma unroll
                for (int f = 0; f < FPW; ++f) {
                    const int t = tile * FPB + wave * FPW + f;
                    const long long s0 = (long long)t * hop - N;
#pragma unroll
                    for (int i = 0; i < E; ++i) {
                        const long long s = s0 + 2 * (p + i * P);
                        const unsigned lo = (t < T && s >= 0 && s < n_samples) ? (unsigned short)xs[s] : 0u;
                        const unsigned hi = (t < T && s + 1 >= 0 && s + 1 < n_samples) ? (unsigned short)xs[s + 1] : 0u;
                        xr[f][i] = __builtin_bit_cast(float, lo | hi << 16);
                    }
                }
            }
        } else if (ALIGNED && first >= 0 && last <= n_samples && tile * FPB + FPB <= T) {
            const float* src = xc + first + (long long)(wave * FPW) * hop + 2 * p;
#pragma unroll
            for (int f = 0; f < FPW; ++f) {
#pragma unroll
                for (int i = 0; i < E; ++i) xr[f][i] = *reinterpret_cast<const float2*>(src + (long long)f * hop + 2 * i * P);
            }
        } else {
#pragma unroll
            for (int f = 0; f < FPW; ++f) {
                const int t = tile * FPB + wave * FPW + f;
                const long long s0 = (long long)t * hop - N;
#pragma unroll
                for (int i = 0; i < E; ++i) {
                    const long long s = s0 + 2 * (p + i * P);
                    xr[f][i].x = (t < T && s >= 0 && s < n_samples) ? xc[s] : 0.f;
                    xr[f][i].y = (t < T && s + 1 >= 0 && s + 1 < n_samples) ? xc[s + 1] : 0.f;
                }
            }
        }
    };
    int tlv = blockIdx.x;
    prefetch(tlv);
    PROF_INIT(g_prof_stft);
    for (; tlv < total_tiles; tlv += gridDim.x) {
        const int tl = xcd ? xcd_order(tlv, total_tiles) : tlv;
        const int clip = tl / tiles, tile = tl % tiles;
        const int t0 = tile * FPB;
        PROF_MARK(0);
        int po = p;   // opaque copy: window and twiddle reads are not hoisted out of the tile loop (at 32 points per
        if constexpr (E >= 32 || ZAFX_STFT_FAT_OPAQUE(LOG2N, SPEC)) asm volatile("" : "+v"(po));   // thread the hoisted values no longer fit beside the prefetch)
#pragma unroll
        for (int f = 0; f < FPW; ++f) {
            float2 v[E];
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const float2 wv = win_l[po + i * P], xv = PP::get(xr[f][i]);
                v[i] = make_float2(xv.x * wv.x, xv.y * wv.y);
            }
            fft_frame<LOG2N, LOG2E>(v, frames + (wave * FPW + f) * PITCH, po, tw_l);
        }
#ifndef ZAFX_STFT_EARLY
#define ZAFX_STFT_EARLY 1
#endif
        // The next tile is requested by each wave as soon as ITS frames are transformed, ahead of the barrier (EARLY): the waves
        // finish a few thousand cycles apart, so their bursts of requests arrive spread out while the early waves would only wait,
        // and the store phase starts right behind the barrier -- requested behind the barrier, the 32 loads of a lane blocked at
        // the CU's vector-memory queue for 5-11 k cycles before the wave's first store.
        // (The two frames of a wave are adjacent and overlap by W - hop samples: requested together, the shared half is served by
        // the vector cache; requested half a store phase apart it was fetched from HBM twice.)
        if (ZAFX_STFT_EARLY) prefetch(tlv + gridDim.x);
        PROF_MARK(1);
        lds_barrier();
        PROF_MARK(2);
        if (!ZAFX_STFT_EARLY) prefetch(tlv + gridDim.x);   // in flight while this tile is stored
        PROF_MARK(3);
        if (t0 + tt < T) {
            float2* o = spec_base<SPEC>(out, (long long)clip * ROWS * TP + (t0 + tt));
            int kqo = kq;   // (opaque at 32 points per thread: the split roots of the 16 iterations are not carried across tiles)
            if constexpr (E >= 32 || ZAFX_STFT_FAT_OPAQUE(LOG2N, SPEC)) asm volatile("" : "+v"(kqo));
            auto store_tile = [&](auto stream) {
                constexpr bool ST = decltype(stream)::value;
                // Every workgroup starts its sweep over the rows somewhere else (its XCD and its place in the XCD decide): the
                // persistent workgroups run in step, and with all of them on the same rows at the same time the chip wrote a few
                // narrow bands of every clip at once; staggered, the stores of a moment spread over the clips' whole extent
                // (tools/placement.py, 13 allocations: 1.539-1.573 -> 1.495-1.538 ms; ZAFX_STFT_ROWROT_EXPR=0 restores the old order).
                constexpr int ITER = (N / 2) / (NT / FPB);
#ifndef ZAFX_STFT_ROWROT_EXPR
#define ZAFX_STFT_ROWROT_EXPR (((int)blockIdx.x & 7) * 2 + ((int)blockIdx.x >> 3))
#endif
                static_assert((N / 2) % (NT / FPB) == 0, "the row sweep is a whole number of iterations");
                // (only where every store is a whole line: rows that straddle lines rely on the neighbouring tile -- the neighbouring
                // workgroup of the XCD -- writing the other part of the line at about the same time, which L2 then merges)
                const int rot = ST ? ZAFX_STFT_ROWROT_EXPR : 0;
                for (int it = 0; it < ITER; ++it) {
                    const int k = kqo + ((it + rot) % ITER) * (NT / FPB);
                    if (k == 0) {
                        const float2 z0 = fb[0], zc = fb[phys_t<C::PS>(N / 2)];
                        put_bin<SPEC, ST>(o, 0, make_float2(z0.x + z0.y, 0.f));
                        put_bin<SPEC, ST>(o, (long long)N * TP, make_float2(z0.x - z0.y, 0.f));
                        put_bin<SPEC, ST>(o, (long long)(N / 2) * TP, cconj(zc));
                        if (SPEC == 0) put_bin<SPEC, ST>(o, (long long)(N + N / 2) * TP, zc);
                    } else {
                        float2 xk, xn;
                        split_pair(fb[phys_t<C::PS>(k)], fb[phys_t<C::PS>(N - k)], tws_l[k], xk, xn);
                        put_bin<SPEC, ST>(o, (long long)k * TP, xk);
                        if (SPEC == 0) put_bin<SPEC, ST>(o, (long long)(W - k) * TP, cconj(xk));
                        put_bin<SPEC, ST>(o, (long long)(N - k) * TP, xn);
                        if (SPEC == 0) put_bin<SPEC, ST>(o, (long long)(N + k) * TP, cconj(xn));
                    }
                }
            };
            // whole-line rows (pitch and base multiples of 128 B) stream past L2; rows that straddle lines keep the
            // write-combining of ordinary stores (non-temporal partial lines: T = 433, 1.11 -> 1.51 ms per 256 clips)
#ifndef ZAFX_STFT_NO_NT
            if (SPEC < 2 && lines_whole && FPB == 16) store_tile(std::true_type{});   // (8-frame tiles write half lines: ordinary stores)
            else
#endif
                store_tile(std::false_type{});
        }
        PROF_MARK(4);
        lds_barrier();   // LDS reads of the tile are done; its global stores are NOT waited for
    }
}

// ---------------------------------------------------------------------------------
// forward, reference layout, rows that are NOT whole 128-byte lines (T % 16 != 0): k_stft_ft16 with a register carry
// ---------------------------------------------------------------------------------
// The reference's (W, T) array is compact (zaf.py:128), so unless T is a multiple of 16 a tile's 128-byte run of a row
// straddles two lines and every line is written in two parts by two workgroups at different times: HBM3E has no byte masks
// (partial sectors are read-modify-written), and the 8 MB a XCD's workgroups have in flight do not let L2 merge the parts --
// T = 433 ran at 1.96 TB/s where T = 432 runs at 5.4 (round 3 bench line, stft_offgrid).  This form walks the tiles of a clip
// segment IN ORDER and carries every row's previous 16 values in registers: thread (frame column tt, rows k = kq + 32 it)
// keeps its own X[k], X[N-k] of the previous tile (64 VGPRs; the mirror rows are their conjugates).  For a row whose run
// starts a frames into a line, lanes tt < 16 - a store the current tile's value at frame t0 + tt and lanes tt >= 16 - a store
// the CARRIED value at frame t0 - 16 + tt: the sixteen lanes of an instruction again cover exactly one line.  Partial lines
// remain only at the two ends of a segment (head: no carry yet; tail: flushed with the last tile).  Radix-16 schedule
// (32 + 64 registers of transform and prefetch beside the carry; the radix-32 form of k_stft_ft16 stands at 206).
template <int LOG2N, int LOG2E, bool ALIGNED, int SPEC, int PCM = 0>
__global__ __launch_bounds__(kFatWaves * 64) void k_stft_ft16c(
    const float* __restrict__ x, const float* __restrict__ win, const float2* __restrict__ twp,
    const float2* __restrict__ tws, float2* __restrict__ out, long long n_samples, int hop, int T, int TP, int tiles,
    int segs, int seg_tiles, int units) {
    static_assert(SPEC < 2, "complex spectra only (the real kinds write 64-byte runs)");
    using C = FftCfg<LOG2N, LOG2E>;
    using F = FatCfg<LOG2N, LOG2E>;
    constexpr int N = C::N, P = C::P, E = C::E, W = 2 * N, NT = F::NT, FPB = kFatFrames, FPW = F::FPW, PITCH = F::PITCH;
    constexpr int ROWS = SPEC ? N + 1 : W;
    constexpr int ITER = (N / 2) / (NT / FPB);
    static_assert((N / 2) % (NT / FPB) == 0 && ITER >= 1, "the row sweep is a whole number of iterations");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* frames = reinterpret_cast<float2*>(smem_raw);
    float2* tw_l = frames + FPB * PITCH;
    float2* win_l = tw_l + C::TW;
    float2* tws_l = win_l + N;
    const int tid = threadIdx.x;
    for (int i = tid; i < C::TW; i += NT) tw_l[i] = twp[i];
    using PP = PcmPoint<PCM>;
    static_assert(PCM == 0 || ALIGNED, "int16 input: the buffer-load form");
    for (int i = tid; i < N; i += NT) {
        const float2 wv = reinterpret_cast<const float2*>(win)[i];
        win_l[i] = make_float2(wv.x * PP::scale(), wv.y * PP::scale());
    }
    for (int i = tid; i <= N / 2; i += NT) tws_l[i] = tws[i];
    __syncthreads();
    const int wave = tid / P, p_lane = tid % P;
    const int tt = tid % FPB, kq = tid / FPB;
    const float2* fb = frames + tt * PITCH;
    const bool xcd = ZAFX_XCD_ORDER && gridDim.x % 8 == 0;
    const int b0 = (int)((reinterpret_cast<uintptr_t>(out) >> 3) & 15);   // phase of the array's first element in its line

    typename PP::T xr[FPW][E];
    auto prefetch = [&](int clip, int tile) {
        const float* xc = x + (long long)clip * n_samples;
        if constexpr (ALIGNED) {
            // buffer loads with the CLIP as descriptor: a pair of samples before the clip's first or behind its last sample is out
            // of the descriptor's range and reads as zero -- which is the reference's zero padding (zaf.py:112-125; n_samples, hop
            // and every pair's first sample are even, so a pair is inside or outside as a whole).  No edge path, no 64-bit addresses.
            if constexpr (PCM == 1) {   // int16 mono: a point is 4 bytes
                const __amdgpu_buffer_rsrc_t rs = make_rsrc(reinterpret_cast<const short*>(x) + (long long)clip * n_samples, (unsigned)(n_samples * 2));
#pragma unroll
                for (int f = 0; f < FPW; ++f) {
                    const int s0 = (tile * FPB + wave * FPW + f) * hop - N;
                    const int vo = (s0 + 2 * p_lane) * 2;
#pragma unroll
                    for (int i = 0; i < E; ++i) xr[f][i] = buf_load_f32(rs, vo + i * P * 4);
                }
            } else {
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(xc, (unsigned)(n_samples * 4));
#pragma unroll
            for (int f = 0; f < FPW; ++f) {
                const int s0 = (tile * FPB + wave * FPW + f) * hop - N;   // (clips below 2^29 samples: run_stft_fat_carry)
                const int vo = (s0 + 2 * p_lane) * 4;
#pragma unroll
                for (int i = 0; i < E; ++i) xr[f][i] = buf_load_f32x2(rs, vo + i * P * 8);
            }
            }
        } else if constexpr (PCM != 1) {
#pragma unroll
            for (int f = 0; f < FPW; ++f) {
                const int t = tile * FPB + wave * FPW + f;
                const long long s0 = (long long)t * hop - N;
#pragma unroll
                for (int i = 0; i < E; ++i) {
                    const long long s = s0 + 2 * (p_lane + i * P);
                    xr[f][i].x = (t < T && s >= 0 && s < n_samples) ? xc[s] : 0.f;
                    xr[f][i].y = (t < T && s + 1 >= 0 && s + 1 < n_samples) ? xc[s + 1] : 0.f;
                }
            }
        }
    };
    // unit v of this workgroup's walk -> clip and tile range [j, j1) of its segment
    auto unit_of = [&](int v, int& clip, int& j, int& j1) {
        const int u = xcd ? xcd_order(v, units) : v;
        clip = u / segs;
        j = (u % segs) * seg_tiles;
        j1 = min(j + seg_tiles, tiles);
    };
    int v = blockIdx.x;
    if (v >= units) return;
    int clip, j, j1;
    unit_of(v, clip, j, j1);
    prefetch(clip, j);
    float2 ck[ITER], cn[ITER];   // the thread's X[k], X[N-k] of the previous tile (k = kq + 32 it)
#pragma unroll
    for (int it = 0; it < ITER; ++it) ck[it] = cn[it] = make_float2(0.f, 0.f);
    bool have_prev = false;
    for (;;) {
        int po = p_lane;
        asm volatile("" : "+v"(po));
        // The windowed samples of the wave's later frames are parked in their own (still unused) frame buffers while the first
        // frame is transformed: the prefetch registers are then free through the butterflies, whose ~110 registers beside the
        // 64 of the carry and 32 of a second frame's samples spilled (100 bytes of scratch per lane, reloads draining vmcnt).
#pragma unroll
        for (int f = 1; f < FPW; ++f) {
            float2* park = frames + (wave * FPW + f) * PITCH;
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const float2 wv = win_l[po + i * P], xv = PP::get(xr[f][i]);
                park[po + i * P] = make_float2(xv.x * wv.x, xv.y * wv.y);
            }
        }
#pragma unroll
        for (int f = 0; f < FPW; ++f) {
            float2 vv[E];
            float2* buf = frames + (wave * FPW + f) * PITCH;
            if (f == 0) {
#pragma unroll
                for (int i = 0; i < E; ++i) {
                    const float2 wv = win_l[po + i * P], xv = PP::get(xr[0][i]);
                    vv[i] = make_float2(xv.x * wv.x, xv.y * wv.y);
                }
            } else {
                frame_sync<P>();
#pragma unroll
                for (int i = 0; i < E; ++i) vv[i] = buf[po + i * P];
                frame_sync<P>();   // every lane has its samples before the first pass overwrites the buffer
            }
            fft_frame<LOG2N, LOG2E>(vv, buf, po, tw_l);
            __builtin_amdgcn_sched_barrier(0);
        }
        // the next tile of the walk, requested ahead of the barrier (as k_stft_ft16)
        int nclip = clip, nj = j + 1, nj1 = j1, nv = v;
        bool more = true;
        if (nj >= j1) {
            nv = v + gridDim.x;
            if (nv < units) unit_of(nv, nclip, nj, nj1);
            else more = false;
        }
        if (more) prefetch(nclip, nj);
        lds_barrier();
        {
            const int t0 = j * FPB;
            const bool last = j + 1 >= j1, cur_ok = t0 + tt < T;
            float2* o = out + (long long)clip * ROWS * TP + (t0 + tt);
            const int c0 = (int)(((long long)clip * ROWS) & 15), tp = TP & 15;
            int kqo = kq;
            asm volatile("" : "+v"(kqo));
            auto sweep = [&](auto stream) {
                constexpr bool ST = decltype(stream)::value;
                // one row of the sweep: `cur` = this tile's value of the thread's frame, `prev` = the carried one
                auto emit = [&](int r, float2 cur, float2 prev) {
                    const int a = (b0 + (c0 + r) * tp) & 15;   // frames of this tile's run that lie before the first line boundary ... the run starts a frames into a line
                    const bool from_prev = tt >= 16 - a;        // (a = 0: never -- the run is a whole line)
                    const float2 val = from_prev ? prev : cur;
                    float2* dst = o + (long long)r * TP;
                    if (from_prev ? have_prev : cur_ok) {
                        if constexpr (ST) store_stream(dst + (from_prev ? -16 : 0), val);
                        else dst[from_prev ? -16 : 0] = val;
                    }
                    if (last && from_prev && cur_ok) *dst = cur;   // tail of the segment's last run (a partial line)
                };
#pragma unroll
                for (int it = 0; it < ITER; ++it) {
                    const int k = kqo + it * (NT / FPB);
                    float2 xk, xn;
                    if (it == 0 && k == 0) {
                        const float2 z0 = fb[0], zc = fb[phys_t<C::PS>(N / 2)];
                        xk = cconj(zc);                                   // row N/2 (its mirror, row 3N/2: zc)
                        xn = make_float2(z0.x + z0.y, z0.x - z0.y);       // rows 0 and N: both real, carried as one pair
                        emit(N / 2, xk, ck[0]);
                        if (SPEC == 0) emit(N + N / 2, cconj(xk), cconj(ck[0]));
                        emit(0, make_float2(xn.x, 0.f), make_float2(cn[0].x, 0.f));
                        emit(N, make_float2(xn.y, 0.f), make_float2(cn[0].y, 0.f));
                    } else {
                        split_pair(fb[phys_t<C::PS>(k)], fb[phys_t<C::PS>(N - k)], tws_l[k], xk, xn);
                        emit(k, xk, ck[it]);
                        if (SPEC == 0) emit(W - k, cconj(xk), cconj(ck[it]));
                        emit(N - k, xn, cn[it]);
                        if (SPEC == 0) emit(N + k, cconj(xn), cconj(cn[it]));
                    }
                    ck[it] = xk;
                    cn[it] = xn;
                    __builtin_amdgcn_sched_barrier(0);   // (iterations stay apart: their LDS reads hoisted ahead cost the registers the carry needs)
                }
            };
            // a segment's first tile writes partial lines (no carry yet): ordinary stores, which L2 may still merge with the
            // neighbouring segment's tail; every later tile writes whole lines and streams them
            if (have_prev) sweep(std::true_type{});
            else sweep(std::false_type{});
        }
        lds_barrier();
        if (!more) break;
        have_prev = nj != 0 && nv == v;   // the walk continues inside the same segment
        clip = nclip, j = nj, j1 = nj1, v = nv;
    }
}

// ---------------------------------------------------------------------------------
// forward, reference layout, W = 4096: the persistent 16-frame kernel over two BANDS of bins (k_stft_ft16b)
// ---------------------------------------------------------------------------------
// Sixteen packed frames of 2048 points are 256 KB: they do not fit LDS, so W = 4096 ran on the one-workgroup-per-tile kernel
// (2.8 TB/s), and eight-frame tiles write 64-byte runs (measured: 6.5 ms against 3.1).  One decimation-in-frequency step on
// the way in splits the packed frame z (M = 2048 points) into two 1024-point transforms that are complete on their own:
//     Z[2q]     = FFT_1024( z[n] + z[n + 1024] )[q]                         (band 0)
//     Z[2q + 1] = FFT_1024( (z[n] - z[n + 1024]) exp(-2 pi i n / 2048) )[q]  (band 1)
// and the real split pairs Z[m] with Z[M - m], i.e. band 0 with band 0 (q, 1024 - q) and band 1 with band 1 (q, 1023 - q):
// a band is transformed, split and stored without the other.  So a tile is two rounds of the W = 2048 kernel's phases on the
// SAME sixteen frame buffers -- transforms of band 0, barrier, stores of the even rows (128-byte runs), barrier, transforms
// of band 1 (from registers: both bands are formed from the prefetched samples at once), request of the next tile, barrier,
// stores of the odd rows.  The window (16 KB) does not fit beside the frames and is read from global memory once per tile
// and lane (both frames of a wave use the same 32 pairs); exp(-2 pi i n / 2048), n = lane + 64 i, is one per-lane root
// (loop invariant, a register pair) times a compile-time constant.
struct BandCfg {
    using C = FftCfg<10, 4>;
    static constexpr int N = 1024, M = 2048, W = 4096, FPB = kFatFrames, NT = kFatWaves * 64, FPW = 2;
    static constexpr int PITCH = FatCfg<10, 4>::PITCH;
    static constexpr size_t SMEM = (size_t)(FPB * PITCH + C::TW + M / 2 + 1) * 8;
};
static_assert(BandCfg::SMEM <= (size_t)kMaxLdsBytes, "k_stft_ft16b: tile + tables exceed LDS");

template <bool ALIGNED, int SPEC>
__global__ __launch_bounds__(kFatWaves * 64) void k_stft_ft16b(
    const float* __restrict__ x, const float* __restrict__ win, const float2* __restrict__ twp,
    const float2* __restrict__ tws, float2* __restrict__ out, long long n_samples, int hop, int T, int TP, int tiles,
    int total_tiles) {
    using B = BandCfg;
    using C = B::C;
    constexpr int N = B::N, M = B::M, W = B::W, P = 64, E = 16, NT = B::NT, FPB = B::FPB, FPW = B::FPW, PITCH = B::PITCH;
    constexpr int ROWS = SPEC ? M + 1 : W;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* frames = reinterpret_cast<float2*>(smem_raw);
    float2* tw_l = frames + FPB * PITCH;   // pass tables of the 1024-point transform
    float2* tws_l = tw_l + C::TW;          // exp(-2 pi i k / W), k <= M / 2
    const int tid = threadIdx.x;
    for (int i = tid; i < C::TW; i += NT) tw_l[i] = twp[i];
    for (int i = tid; i <= M / 2; i += NT) tws_l[i] = tws[i];
    __syncthreads();
    const int wave = tid / P, p_lane = tid % P;
    const int tt = tid % FPB, kq = tid / FPB;
    const float2* fb = frames + tt * PITCH;
    const bool lines_whole = TP % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 128 == 0;
    const bool xcd = ZAFX_XCD_ORDER && gridDim.x % 8 == 0;
    const float2 wp = tws_l[2 * p_lane];   // exp(-2 pi i lane / M)

    float2 xr[FPW][2 * E];   // z[lane + 64 j], j < 32, of the wave's two frames
    auto prefetch = [&](int tlv) {
        if (tlv >= total_tiles) return;
        const int tl = xcd ? xcd_order(tlv, total_tiles) : tlv;
        const int clip = tl / tiles, tile = tl % tiles;
        const float* xc = x + (long long)clip * n_samples;
        // the clip as buffer descriptor: samples outside it read as zero = the reference's padding (zaf.py:112-125); no edge path,
        // no 64-bit addresses (clips below 2^29 samples: run_stft).  ALIGNED (even clip length, hop and base): a pair is inside
        // or outside as a whole and comes as one 8-byte load; otherwise two 4-byte loads.
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(xc, (unsigned)(n_samples * 4));
#pragma unroll
        for (int f = 0; f < FPW; ++f) {
            const int s0 = (tile * FPB + wave * FPW + f) * hop - M;
            const int vo = (s0 + 2 * p_lane) * 4;
            int vo1 = vo + 4;
            // (opaque: were the two 4-byte loads of a pair provably adjacent the compiler would merge them into one 8-byte load, and
            // a pair that straddles an end of the clip would then be out of range as a whole -- its inside sample read as zero)
            if constexpr (!ALIGNED) asm volatile("" : "+v"(vo1));
#pragma unroll
            for (int j = 0; j < 2 * E; ++j) {
                if constexpr (ALIGNED) {
                    xr[f][j] = buf_load_f32x2(rs, vo + j * P * 8);
                } else {
                    xr[f][j].x = buf_load_f32(rs, vo + j * P * 8);
                    xr[f][j].y = buf_load_f32(rs, vo1 + j * P * 8);
                }
            }
        }
    };
    // rows of one band: X[k] and X[M - k] (+ their mirrors W - k, M + k) from the pair (Z[k], Z[M - k]), k = 2 q + S
    auto store_band = [&](auto band, auto stream, float2* o) {
        constexpr int S = decltype(band)::value;
        constexpr bool ST = decltype(stream)::value;
        constexpr int ITER = (N / 2) / (NT / FPB);
        static_assert((N / 2) % (NT / FPB) == 0, "the row sweep is a whole number of iterations");
        int kqo = kq;
        asm volatile("" : "+v"(kqo));   // (the split roots of the 16 iterations are not carried across tiles)
        const int rot = ST ? ZAFX_STFT_ROWROT_EXPR : 0;
        for (int it = 0; it < ITER; ++it) {
            const int q = kqo + ((it + rot) % ITER) * (NT / FPB);
            if (S == 0 && q == 0) {
                const float2 z0 = fb[0], zc = fb[phys_t<C::PS>(N / 2)];   // Z[0], Z[M / 2]
                put_bin<SPEC, ST>(o, 0, make_float2(z0.x + z0.y, 0.f));
                put_bin<SPEC, ST>(o, (long long)M * TP, make_float2(z0.x - z0.y, 0.f));
                put_bin<SPEC, ST>(o, (long long)(M / 2) * TP, cconj(zc));
                if (SPEC == 0) put_bin<SPEC, ST>(o, (long long)(M + M / 2) * TP, zc);
            } else {
                const int k = 2 * q + S;
                float2 xk, xn;
                split_pair(fb[phys_t<C::PS>(q)], fb[phys_t<C::PS>(N - S - q)], tws_l[k], xk, xn);
                put_bin<SPEC, ST>(o, (long long)k * TP, xk);
                if (SPEC == 0) put_bin<SPEC, ST>(o, (long long)(W - k) * TP, cconj(xk));
                put_bin<SPEC, ST>(o, (long long)(M - k) * TP, xn);
                if (SPEC == 0) put_bin<SPEC, ST>(o, (long long)(M + k) * TP, cconj(xn));
            }
        }
    };
    int tlv = blockIdx.x;
    prefetch(tlv);
    for (; tlv < total_tiles; tlv += gridDim.x) {
        const int tl = xcd ? xcd_order(tlv, total_tiles) : tlv;
        const int clip = tl / tiles, tile = tl % tiles;
        const int t0 = tile * FPB;
        int po = p_lane;
        asm volatile("" : "+v"(po));   // (window offsets and table addresses are recomputed per tile, not hoisted)
        {
            // window pairs of this lane (the same for both frames), then both bands of both frames in place:
            // xr[f][i] <- a + b, xr[f][i + 16] <- (a - b) exp(-2 pi i (lane + 64 i) / M), a = w z[n], b = w z[n + 1024]
            const float2* w2 = reinterpret_cast<const float2*>(win) + po;
            // exp(-2 pi i k / 32) = (c32[k], -s32[k])
            const float c32[16] = {1.f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f, 0.70710678118654752440f,
                                   0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f, 0.f,
                                   -0.19509032201612826785f, -0.38268343236508977173f, -0.55557023301960222474f, -0.70710678118654752440f,
                                   -0.83146961230254523708f, -0.92387953251128675613f, -0.98078528040323044913f};
            const float s32[16] = {0.f, 0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f, 0.70710678118654752440f,
                                   0.83146961230254523708f, 0.92387953251128675613f, 0.98078528040323044913f, 1.f,
                                   0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f, 0.70710678118654752440f,
                                   0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f};
            float2 wv[2 * E];
#pragma unroll
            for (int j = 0; j < 2 * E; ++j) wv[j] = w2[j * P];
#pragma unroll
            for (int f = 0; f < FPW; ++f) {
#pragma unroll
                for (int i = 0; i < E; ++i) {
                    const float2 a = make_float2(xr[f][i].x * wv[i].x, xr[f][i].y * wv[i].y);
                    const float2 b = make_float2(xr[f][i + E].x * wv[i + E].x, xr[f][i + E].y * wv[i + E].y);
                    xr[f][i] = cadd(a, b);
                    const float2 d = csub(a, b);
                    xr[f][i + E] = cmul(i == 0 ? d : (i == 8 ? mul_mi(d) : cmulk(d, c32[i], -s32[i])), wp);
                }
            }
        }
#pragma unroll
        for (int f = 0; f < FPW; ++f) fft_frame<10, 4>(&xr[f][0], frames + (wave * FPW + f) * PITCH, po, tw_l);
        lds_barrier();
        float2* o = spec_base<SPEC>(out, (long long)clip * ROWS * TP + (t0 + tt));
        const bool stream = SPEC < 2 && lines_whole;
        if (t0 + tt < T) {
            if (stream) store_band(std::integral_constant<int, 0>{}, std::true_type{}, o);
            else store_band(std::integral_constant<int, 0>{}, std::false_type{}, o);
        }
        lds_barrier();
#pragma unroll
        for (int f = 0; f < FPW; ++f) fft_frame<10, 4>(&xr[f][E], frames + (wave * FPW + f) * PITCH, po, tw_l);
        prefetch(tlv + gridDim.x);   // each wave as soon as ITS frames are done, ahead of the barrier (as k_stft_ft16)
        lds_barrier();
        if (t0 + tt < T) {
            if (stream) store_band(std::integral_constant<int, 1>{}, std::true_type{}, o);
            else store_band(std::integral_constant<int, 1>{}, std::false_type{}, o);
        }
        lds_barrier();
    }
}

// ---------------------------------------------------------------------------------
// forward, reference layout, W = 8192: the persistent 16-frame kernel over FOUR classes of bins (k_stft_ft16q)
// ---------------------------------------------------------------------------------
// Two decimation steps of the kind k_stft_ft16b takes once would pair band 1 with band 3 in the real split (Z[m] with Z[M - m]), and two bands of a
// 16-frame tile do not fit LDS together.  Decimating the REAL frame v[n] = w[n] x[n] (x_m[n] = v[n + 2048 m]) instead gives four 1024-point complex
// transforms that are each complete on their own, with no pairing across them and no split arithmetic but for the first:
//     X[4q]      = RFFT_2048(s)[q],  s = x0 + x1 + x2 + x3             packed (s[2n], s[2n + 1]) + the real split, as W = 2048   (class A)
//     X[8p + 2]  = FFT_1024(c)[p],   c[n] = (r[n] - i r[n + 1024]) w_4096^n,  r = x0 - x1 + x2 - x3                                (class C)
//     X[8p + 1]  = FFT_1024(h[n] + h[n + 1024])[p],  h[n] = ((x0 - x2) - i (x1 - x3))[n] w_8192^n,  n < 2048                      (class D1)
//     X[8p + 5]  = FFT_1024((h[n] - h[n + 1024]) w_2048^n)[p]                                                                      (class D5)
// and every other row is a mirror, X[W - k] = conj X[k] (rows 6, 7, 3 mod 8).  A tile is four rounds of the W = 2048 kernel's phases on the same
// sixteen frame buffers; the frame is read twice (classes A + C from the sums x0 + x2, x1 + x3; D1 + D5 from the differences), as 4-byte buffer
// loads of n = lane + 64 i (any clip alignment; samples outside the clip read 0 = the reference's padding, zaf.py:112-125), the second reader
// from L2.  Class A goes through the frame buffer once (its packed pairs lie in neighbouring lanes); the others are formed in the transform's
// own register arrangement, the waiting class (C, D5) kept in 64 registers.
#ifndef ZAFX_QCH
#define ZAFX_QCH 1
#endif
constexpr int QCH = ZAFX_QCH;   // pairs (n, n + 1024) of a frame requested at once
struct QuadCfg {
    using C = FftCfg<10, 4>;
    static constexpr int N = 1024, Q = 2048, W = 8192, FPB = kFatFrames, NT = kFatWaves * 64, FPW = 2;
    static constexpr int PITCH = FatCfg<10, 4>::PITCH;
    static constexpr size_t SMEM = (size_t)(FPB * PITCH + C::TW + N / 2 + 1) * 8;
};
static_assert(QuadCfg::SMEM <= (size_t)kMaxLdsBytes, "k_stft_ft16q: tile + tables exceed LDS");

template <int SPEC>
__global__ __launch_bounds__(kFatWaves * 64) void k_stft_ft16q(
    const float* __restrict__ x, const float* __restrict__ win, const float2* __restrict__ twp, const float2* __restrict__ twq,
    float2* __restrict__ out, long long n_samples, int hop, int T, int TP, int tiles, int total_tiles) {
    using B = QuadCfg;
    using C = B::C;
    constexpr int N = B::N, Q = B::Q, W = B::W, P = 64, E = 16, NT = B::NT, FPB = B::FPB, FPW = B::FPW, PITCH = B::PITCH;
    constexpr int ROWS = SPEC ? W / 2 + 1 : W;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* frames = reinterpret_cast<float2*>(smem_raw);
    float2* tw_l = frames + FPB * PITCH;   // pass tables of the 1024-point transform
    float2* tws_l = tw_l + C::TW;          // exp(-2 pi i q / 2048), q <= 512: the split roots of class A
    const int tid = threadIdx.x;
    for (int i = tid; i < C::TW; i += NT) tw_l[i] = twp[i];
    for (int i = tid; i <= N / 2; i += NT) tws_l[i] = twq[4 * i];
    __syncthreads();
    const int wave = tid / P, p_lane = tid % P;
    const int tt = tid % FPB, kq = tid / FPB;
    const float2* fb = frames + tt * PITCH;
    const bool lines_whole = TP % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 128 == 0;
    const bool xcd = ZAFX_XCD_ORDER && gridDim.x % 8 == 0;
    // QCH pairs (n, n + 1024), n = lane + 64 (i0 + j), of the wave's two frames (samples s0, s0 + hop on): the four quarters' samples and window
    // values, requested one chunk ahead of their use (walk)
    struct Chunk {
        float xs[FPW][QCH][2][4], ws[QCH][2][4];
    };
    auto request = [&](const __amdgpu_buffer_rsrc_t& rs, int s0, int i0, Chunk& c) {
        int pc = p_lane;
        asm volatile("" : "+v"(pc));   // (the window values of a chunk are read again in the other pass, not kept)
#pragma unroll
        for (int j = 0; j < QCH; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int n = pc + 64 * (i0 + j) + N * h + Q * m;
#pragma unroll
                    for (int f = 0; f < FPW; ++f) c.xs[f][j][h][m] = buf_load_f32(rs, (s0 + f * hop + n) * 4);
                    c.ws[j][h][m] = win[n];
                }
    };
    // every pair of the two frames in turn: use(i, v) with v[f][h][m] = w x of frame f, sample lane + 64 i + 1024 h + 2048 m
    auto walk = [&](const __amdgpu_buffer_rsrc_t& rs, int s0, auto use) {
        Chunk cb[2];
        int pu = p_lane;
        asm volatile("" : "+v"(pu));   // (the roots of a pass are read again per tile, not hoisted out of the tile loop and spilled)
        request(rs, s0, 0, cb[0]);
#pragma unroll
        for (int i0 = 0; i0 < E; i0 += QCH) {
            const int cur = (i0 / QCH) & 1;
            if (i0 + QCH < E) request(rs, s0, i0 + QCH, cb[cur ^ 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < QCH; ++j) {
                float v[FPW][2][4];
#pragma unroll
                for (int f = 0; f < FPW; ++f)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int m = 0; m < 4; ++m) v[f][h][m] = cb[cur].xs[f][j][h][m] * cb[cur].ws[j][h][m];
                use(i0 + j, pu, v);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // rows 4q and their mirrors from the packed transform of s: the W = 2048 split, rows four apart
    auto store_a = [&](auto stream, float2* o) {
        constexpr bool ST = decltype(stream)::value;
        constexpr int ITER = (N / 2) / (NT / FPB);
        int kqo = kq;
        asm volatile("" : "+v"(kqo));
#pragma unroll 2
        for (int it = 0; it < ITER; ++it) {
            const int q = kqo + it * (NT / FPB);
            if (q == 0) {
                const float2 z0 = fb[0], zc = fb[phys_t<C::PS>(N / 2)];
                put_bin<SPEC, ST>(o, 0, make_float2(z0.x + z0.y, 0.f));
                put_bin<SPEC, ST>(o, (long long)(W / 2) * TP, make_float2(z0.x - z0.y, 0.f));
                put_bin<SPEC, ST>(o, (long long)(W / 4) * TP, cconj(zc));
                if (SPEC == 0) put_bin<SPEC, ST>(o, (long long)(3 * W / 4) * TP, zc);
            } else {
                float2 xk, xn;
                split_pair(fb[phys_t<C::PS>(q)], fb[phys_t<C::PS>(N - q)], tws_l[q], xk, xn);
                put_bin<SPEC, ST>(o, (long long)(4 * q) * TP, xk);
                if (SPEC == 0) put_bin<SPEC, ST>(o, (long long)(W - 4 * q) * TP, cconj(xk));
                put_bin<SPEC, ST>(o, (long long)(W / 2 - 4 * q) * TP, xn);
                if (SPEC == 0) put_bin<SPEC, ST>(o, (long long)(W / 2 + 4 * q) * TP, cconj(xn));
            }
        }
    };
    // rows 8p + A and their mirrors W - (8p + A): the transform's values as they are
    auto store_c = [&](auto alpha, auto stream, float2* o) {
        constexpr int A = decltype(alpha)::value;
        constexpr bool ST = decltype(stream)::value;
        constexpr int ITER = N / (NT / FPB);
        int kqo = kq;
        asm volatile("" : "+v"(kqo));
#pragma unroll 4
        for (int it = 0; it < ITER; ++it) {
            const int pq = kqo + it * (NT / FPB);
            const float2 z = fb[phys_t<C::PS>(pq)];
            const int k = 8 * pq + A;
            if (SPEC == 0) {
                put_bin<SPEC, ST>(o, (long long)k * TP, z);
                put_bin<SPEC, ST>(o, (long long)(W - k) * TP, cconj(z));
            } else if (k <= W / 2) {
                put_bin<SPEC, ST>(o, (long long)k * TP, z);
            } else {
                put_bin<SPEC, ST>(o, (long long)(W - k) * TP, cconj(z));
            }
        }
    };
    // A memory instruction of a wave completes in issue order (one counter for loads and stores): a load issued behind a store sweep waits for
    // the sweep to drain.  So both reads of the frame are issued BEFORE a sweep, with the transform's results already in the frame buffers --
    // the differences (classes D1, D5) ahead of the stores of class C, the sums of the NEXT tile (A, C) ahead of the stores of class D5 -- and
    // wait in registers (128) while the sweep is issued; the stores then drain under the transforms that follow.
    float sv[FPW][2 * E];   // s[lane + 64 i], i < 32, of the wave's two frames (class A: through the frame buffer once the last sweep is out)
    float2 hold[FPW][E];    // the waiting class (C, D5)
    auto tile_of = [&](int tlv, int& clip, int& t0) {
        const int tl = xcd ? xcd_order(tlv, total_tiles) : tlv;
        clip = tl / tiles;
        t0 = (tl % tiles) * FPB;
    };
    auto gather_sums = [&](int tlv) {
        if (tlv >= total_tiles) {   // (behind the last tile: the values are dead -- said here, or they would be carried through the whole loop body)
#pragma unroll
            for (int f = 0; f < FPW; ++f)
#pragma unroll
                for (int i = 0; i < 2 * E; ++i) sv[f][i] = 0.f;
            return;
        }
        int clip, t0;
        tile_of(tlv, clip, t0);
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(x + (long long)clip * n_samples, (unsigned)(n_samples * 4));
        __builtin_amdgcn_sched_barrier(0);
        walk(rs, (t0 + wave * FPW) * hop - W / 2, [&](int i, int lane, float (&v)[FPW][2][4]) {
            const float2 wc = twq[2 * (lane + 64 * i)];
#pragma unroll
            for (int f = 0; f < FPW; ++f) {
                float r[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float a = v[f][h][0] + v[f][h][2], b = v[f][h][1] + v[f][h][3];
                    sv[f][i + E * h] = a + b;
                    r[h] = a - b;
                }
                hold[f][i] = cmul(make_float2(r[0], -r[1]), wc);
            }
        });
    };
    int tlv = blockIdx.x;
    gather_sums(tlv);
    for (; tlv < total_tiles; tlv += gridDim.x) {
        int clip, t0;
        tile_of(tlv, clip, t0);
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(x + (long long)clip * n_samples, (unsigned)(n_samples * 4));
        int po = p_lane;
        asm volatile("" : "+v"(po));   // (table offsets are recomputed per tile, not hoisted and spilled)
        float2* o = spec_base<SPEC>(out, (long long)clip * ROWS * TP + (t0 + tt));
        const bool stream = SPEC < 2 && lines_whole;
        const bool mine = t0 + tt < T;
        // ---- class A: s through the frame buffer as packed pairs (s[2n], s[2n + 1])
#pragma unroll
        for (int f = 0; f < FPW; ++f) {
            float2* buf = frames + (wave * FPW + f) * PITCH;
            float* bufs = reinterpret_cast<float*>(buf);
#pragma unroll
            for (int i = 0; i < 2 * E; ++i) {
                const int n = po + 64 * i;
                bufs[2 * phys_t<C::PS>(n >> 1) + (n & 1)] = sv[f][i];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < FPW; ++f) {
            float2* buf = frames + (wave * FPW + f) * PITCH;
            float2 v[E];
            frame_sync<P>();
            regs_read<10, 4>(v, buf, po);
            frame_sync<P>();
            fft_frame<10, 4>(v, buf, po, tw_l);
        }
        lds_barrier();
        if (mine) {
            if (stream) store_a(std::true_type{}, o);
            else store_a(std::false_type{}, o);
        }
        lds_barrier();
        // ---- class C from the registers
#pragma unroll
        for (int f = 0; f < FPW; ++f) fft_frame<10, 4>(&hold[f][0], frames + (wave * FPW + f) * PITCH, po, tw_l);
        // ---- the differences: h = ((x0 - x2) - i (x1 - x3)) w_8192^n; h[n + 1024] carries w_8192^1024 = exp(-i pi / 4)
        float2 d1[FPW][E];
        __builtin_amdgcn_sched_barrier(0);
        walk(rs, (t0 + wave * FPW) * hop - W / 2, [&](int i, int lane, float (&v)[FPW][2][4]) {
            const int n = lane + 64 * i;
            const float2 wn = twq[n], w4 = twq[4 * n];
            const float hs = 0.70710678118654752440f;
#pragma unroll
            for (int f = 0; f < FPW; ++f) {
                const float2 g0 = make_float2(v[f][0][0] - v[f][0][2], v[f][0][3] - v[f][0][1]);
                const float2 g1 = make_float2(v[f][1][0] - v[f][1][2], v[f][1][3] - v[f][1][1]);
                const float2 h0 = cmul(g0, wn), h1 = cmul(cmulk(g1, hs, -hs), wn);
                d1[f][i] = cadd(h0, h1);
                hold[f][i] = cmul(csub(h0, h1), w4);
            }
        });
        lds_barrier();
        if (mine) {
            if (stream) store_c(std::integral_constant<int, 2>{}, std::true_type{}, o);
            else store_c(std::integral_constant<int, 2>{}, std::false_type{}, o);
        }
        lds_barrier();
#pragma unroll
        for (int f = 0; f < FPW; ++f) fft_frame<10, 4>(&d1[f][0], frames + (wave * FPW + f) * PITCH, po, tw_l);
        lds_barrier();
        if (mine) {
            if (stream) store_c(std::integral_constant<int, 1>{}, std::true_type{}, o);
            else store_c(std::integral_constant<int, 1>{}, std::false_type{}, o);
        }
        lds_barrier();
#pragma unroll
        for (int f = 0; f < FPW; ++f) fft_frame<10, 4>(&hold[f][0], frames + (wave * FPW + f) * PITCH, po, tw_l);
        gather_sums(tlv + gridDim.x);
        lds_barrier();
        if (mine) {
            if (stream) store_c(std::integral_constant<int, 5>{}, std::true_type{}, o);
            else store_c(std::integral_constant<int, 5>{}, std::false_type{}, o);
        }
        lds_barrier();
    }
}

// ---------------------------------------------------------------------------------
// melspectrogram / mfcc at W = 4096, fused on the two-band kernel (k_mel_ft16b)
// ---------------------------------------------------------------------------------
// k_stft_ft16b with the stores of a band replaced by its share of the filterbank product: after a band's transforms every thread turns ITS
// pairs (Z[q], Z[N - S - q]) of its frame into |X| (mel, zaf.py:370) or |X|^2 (mfcc, zaf.py:437-439) IN PLACE (each element of the frame
// belongs to exactly one pair, so no other thread reads it in that phase; slot q of band S then holds bin k = 2 q + S, bin M sits in the
// padding slot N), and after a barrier thread (frame t = tid % 16, group g = tid / 16) adds the band's non-zeros of the filters g, g + 32, ...
// to accumulators that live across both bands.  A filter is its band of non-zeros [first, first + count) (zaf.py:305-316: one triangle; the
// float32 band arrays of k_melfb), of which a band of bins takes every other one: about 2 000 products per frame and band against the
// transform's 50 000 flops -- vector FMAs, no matrix cores, and the spectrum never reaches HBM (the k_melfb route writes and re-reads it:
// 12 B/sample against 4.5).  mfcc: log(mel + eps) through the (then dead) frame buffers, DCT-II rows from there (zaf.py:443-452).
template <bool ALIGNED, bool MFCC>
__global__ __launch_bounds__(kFatWaves * 64) void k_mel_ft16b(
    const float* __restrict__ x, const float* __restrict__ win, const float2* __restrict__ twp, const float2* __restrict__ tws,
    const float* __restrict__ fb_vals, const int* __restrict__ fb_meta, const float* __restrict__ dct, float* __restrict__ out,
    long long n_samples, int hop, int T, int TP, int tiles, int total_tiles, int n_filters, int n_coefs, int layout) {
    using B = BandCfg;
    using C = B::C;
    constexpr int N = B::N, M = B::M, P = 64, E = 16, NT = B::NT, FPB = B::FPB, FPW = B::FPW, PITCH = B::PITCH;
    constexpr int NFI = 8;   // filters per thread (n_filters <= 256 = 32 groups x 8)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* frames = reinterpret_cast<float2*>(smem_raw);
    float2* tw_l = frames + FPB * PITCH;
    float2* tws_l = tw_l + C::TW;
    const int tid = threadIdx.x;
    for (int i = tid; i < C::TW; i += NT) tw_l[i] = twp[i];
    for (int i = tid; i <= M / 2; i += NT) tws_l[i] = tws[i];
    __syncthreads();
    const int wave = tid / P, p_lane = tid % P;
    const int tt = tid % FPB, kq = tid / FPB;
    float2* fbw = frames + tt * PITCH;
    const bool xcd = ZAFX_XCD_ORDER && gridDim.x % 8 == 0;
    const float2 wp = tws_l[2 * p_lane];
    const float eps = 2.220446049250313e-16f;   // np.finfo(float).eps (zaf.py:445)

    float2 xr[FPW][2 * E];
    auto prefetch = [&](int tlv) {
        if (tlv >= total_tiles) return;
        const int tl = xcd ? xcd_order(tlv, total_tiles) : tlv;
        const int clip = tl / tiles, tile = tl % tiles;
        const float* xc = x + (long long)clip * n_samples;
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(xc, (unsigned)(n_samples * 4));   // (outside the clip: zero = the reference's padding)
#pragma unroll
        for (int f = 0; f < FPW; ++f) {
            const int s0 = (tile * FPB + wave * FPW + f) * hop - M;
            const int vo = (s0 + 2 * p_lane) * 4;
            int vo1 = vo + 4;
            if constexpr (!ALIGNED) asm volatile("" : "+v"(vo1));   // (see k_stft_ft16b: the two 4-byte loads of a pair must not be merged)
#pragma unroll
            for (int j = 0; j < 2 * E; ++j) {
                if constexpr (ALIGNED) {
                    xr[f][j] = buf_load_f32x2(rs, vo + j * P * 8);
                } else {
                    xr[f][j].x = buf_load_f32(rs, vo + j * P * 8);
                    xr[f][j].y = buf_load_f32(rs, vo1 + j * P * 8);
                }
            }
        }
    };
    auto level = [&](float2 v) {   // |X| (mel) or |X|^2 (mfcc)
        const float pw = v.x * v.x + v.y * v.y;
        return MFCC ? pw : __builtin_amdgcn_sqrtf(pw);
    };
    // a band's spectrum -> levels, in place: slot q <- bin 2 q + S (band 0: slot N <- bin M; slot 0, bin 0, is never read).  Two pairs at
    // a time, all their reads issued before the first write (the writes go to the array the next reads come from, so the compiler cannot move
    // those up: one pair at a time every iteration waited for its own LDS round trip), and both powers of a pair from split_pair_pow4.
    auto levels = [&](auto band) {
        constexpr int S = decltype(band)::value;
        constexpr int ITER = (N / 2) / (NT / FPB), BT = 2;
        int kqo = kq;
        asm volatile("" : "+v"(kqo));
        if (S == 0 && kqo == 0) {   // the thread that holds q = 0 of band 0: bins M and M / 2 (its pair slots 0 / N below are bin 0 and scratch)
            const float2 z0 = fbw[0], zc = fbw[phys_t<C::PS>(N / 2)];
            fbw[phys_t<C::PS>(N)].x = MFCC ? (z0.x - z0.y) * (z0.x - z0.y) : fabsf(z0.x - z0.y);   // X[M] = Re Z[0] - Im Z[0]
            fbw[phys_t<C::PS>(N / 2)].x = level(zc);                                                   // X[M/2] = conj Z[M/2]
        }
#pragma unroll
        for (int it0 = 0; it0 < ITER; it0 += BT) {
            float2 zk[BT], zn[BT], tw[BT];
#pragma unroll
            for (int u = 0; u < BT; ++u) {
                const int q = kqo + (it0 + u) * (NT / FPB);
                zk[u] = fbw[phys_t<C::PS>(q)];
                zn[u] = fbw[phys_t<C::PS>(N - S - q)];
                tw[u] = tws_l[2 * q + S];
            }
#pragma unroll
            for (int u = 0; u < BT; ++u) {
                const int q = kqo + (it0 + u) * (NT / FPB);
                if (S == 0 && q == 0) continue;
                const float2 p4 = split_pair_pow4(zk[u], zn[u], tw[u]);   // (4 |X[k]|^2, 4 |X[M - k]|^2): eight packed instructions for the pair
                fbw[phys_t<C::PS>(q)].x = MFCC ? 0.25f * p4.x : 0.5f * __builtin_amdgcn_sqrtf(p4.x);
                fbw[phys_t<C::PS>(N - S - q)].x = MFCC ? 0.25f * p4.y : 0.5f * __builtin_amdgcn_sqrtf(p4.y);
            }
        }
    };
    // the band's share of FB . S for this thread's frame and filters
    float acc[NFI];
    const int g = tid / FPB;
    const float* sf = reinterpret_cast<const float*>(frames + tt * PITCH);
    auto product = [&](int S) {
#pragma unroll
        for (int i = 0; i < NFI; ++i) {
            const int f = g + 32 * i;
            if (f < n_filters) {
                const int first = fb_meta[3 * f], count = fb_meta[3 * f + 1];
                const float* v = fb_vals + fb_meta[3 * f + 2];
                // bin k = first + 1 + j (zaf.py:370: column c <-> bin c + 1), slot q = (k - S) / 2; four products in flight (the values come
                // from global memory / L1: one at a time the loop is a chain of load latencies)
                float a0 = acc[i], a1 = 0.f, a2 = 0.f, a3 = 0.f;
                const int j0 = (S - first - 1) & 1, q0 = (first + 1 + j0 - S) >> 1;
                const int n = (count - j0 + 1) >> 1;   // entries of this band
                int m = 0;
                for (; m + 4 <= n; m += 4) {
                    const float v0 = v[j0 + 2 * m], v1 = v[j0 + 2 * m + 2], v2 = v[j0 + 2 * m + 4], v3 = v[j0 + 2 * m + 6];
                    const float s0 = sf[2 * phys_t<C::PS>(q0 + m)], s1 = sf[2 * phys_t<C::PS>(q0 + m + 1)];
                    const float s2 = sf[2 * phys_t<C::PS>(q0 + m + 2)], s3 = sf[2 * phys_t<C::PS>(q0 + m + 3)];
                    a0 = fmaf(v0, s0, a0);
                    a1 = fmaf(v1, s1, a1);
                    a2 = fmaf(v2, s2, a2);
                    a3 = fmaf(v3, s3, a3);
                }
                for (; m < n; ++m) a0 = fmaf(v[j0 + 2 * m], sf[2 * phys_t<C::PS>(q0 + m)], a0);
                acc[i] = (a0 + a1) + (a2 + a3);
            }
        }
    };
    int tlv = blockIdx.x;
    prefetch(tlv);
    for (; tlv < total_tiles; tlv += gridDim.x) {
        const int tl = xcd ? xcd_order(tlv, total_tiles) : tlv;
        const int clip = tl / tiles, tile = tl % tiles;
        const int t0 = tile * FPB;
        int po = p_lane;
        asm volatile("" : "+v"(po));
        {
            const float2* w2 = reinterpret_cast<const float2*>(win) + po;
            const float c32[16] = {1.f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f, 0.70710678118654752440f,
                                   0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f, 0.f,
                                   -0.19509032201612826785f, -0.38268343236508977173f, -0.55557023301960222474f, -0.70710678118654752440f,
                                   -0.83146961230254523708f, -0.92387953251128675613f, -0.98078528040323044913f};
            const float s32[16] = {0.f, 0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f, 0.70710678118654752440f,
                                   0.83146961230254523708f, 0.92387953251128675613f, 0.98078528040323044913f, 1.f,
                                   0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f, 0.70710678118654752440f,
                                   0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f};
            float2 wv[2 * E];
#pragma unroll
            for (int j = 0; j < 2 * E; ++j) wv[j] = w2[j * P];
#pragma unroll
            for (int f = 0; f < FPW; ++f) {
#pragma unroll
                for (int i = 0; i < E; ++i) {
                    const float2 a = make_float2(xr[f][i].x * wv[i].x, xr[f][i].y * wv[i].y);
                    const float2 b = make_float2(xr[f][i + E].x * wv[i + E].x, xr[f][i + E].y * wv[i + E].y);
                    xr[f][i] = cadd(a, b);
                    const float2 d = csub(a, b);
                    xr[f][i + E] = cmul(i == 0 ? d : (i == 8 ? mul_mi(d) : cmulk(d, c32[i], -s32[i])), wp);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NFI; ++i) acc[i] = 0.f;
#pragma unroll
        for (int f = 0; f < FPW; ++f) fft_frame<10, 4>(&xr[f][0], frames + (wave * FPW + f) * PITCH, po, tw_l);
        lds_barrier();
        levels(std::integral_constant<int, 0>{});
        lds_barrier();
        product(0);
        lds_barrier();
#pragma unroll
        for (int f = 0; f < FPW; ++f) fft_frame<10, 4>(&xr[f][E], frames + (wave * FPW + f) * PITCH, po, tw_l);
        prefetch(tlv + gridDim.x);
        lds_barrier();
        levels(std::integral_constant<int, 1>{});
        lds_barrier();
        product(1);
        const int t = t0 + tt;
        if constexpr (MFCC) {
            lds_barrier();   // every thread is done with the levels: the frame buffers become the log-mel tile [filter][frame]
            float* lm = reinterpret_cast<float*>(frames);
#pragma unroll
            for (int i = 0; i < NFI; ++i)
                if (g + 32 * i < n_filters) lm[(g + 32 * i) * FPB + tt] = logf(acc[i] + eps);
            lds_barrier();
            if (t < T) {
                for (int c = g; c < n_coefs; c += 32) {
                    const float* d = dct + (long long)c * n_filters;
                    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;   // (eight products in flight: the DCT rows come from global memory / L1)
                    int f = 0;
                    for (; f + 8 <= n_filters; f += 8) {
                        const float d0 = d[f], d1 = d[f + 1], d2 = d[f + 2], d3 = d[f + 3], d4 = d[f + 4], d5 = d[f + 5], d6 = d[f + 6], d7 = d[f + 7];
                        a0 = fmaf(d0, lm[f * FPB + tt], a0);
                        a1 = fmaf(d1, lm[(f + 1) * FPB + tt], a1);
                        a2 = fmaf(d2, lm[(f + 2) * FPB + tt], a2);
                        a3 = fmaf(d3, lm[(f + 3) * FPB + tt], a3);
                        a0 = fmaf(d4, lm[(f + 4) * FPB + tt], a0);
                        a1 = fmaf(d5, lm[(f + 5) * FPB + tt], a1);
                        a2 = fmaf(d6, lm[(f + 6) * FPB + tt], a2);
                        a3 = fmaf(d7, lm[(f + 7) * FPB + tt], a3);
                    }
                    for (; f < n_filters; ++f) a0 = fmaf(d[f], lm[f * FPB + tt], a0);
                    const float r = (a0 + a1) + (a2 + a3);
                    if (layout == ZAFX_LAYOUT_FT) out[((long long)clip * n_coefs + c) * TP + t] = r;
                    else out[((long long)clip * T + t) * n_coefs + c] = r;
                }
            }
        } else if (t < T) {
#pragma unroll
            for (int i = 0; i < NFI; ++i) {
                const int f = g + 32 * i;
                if (f < n_filters) {
                    if (layout == ZAFX_LAYOUT_FT) out[((long long)clip * n_filters + f) * TP + t] = acc[i];
                    else out[((long long)clip * T + t) * n_filters + f] = acc[i];
                }
            }
        }
        lds_barrier();
    }
}

// ---------------------------------------------------------------------------------
// forward, reference layout, W = 4096, complex rows that are NOT whole 128-byte lines: one band per workgroup + register carry (k_stft_ft16bc)
// ---------------------------------------------------------------------------------
// k_stft_ft16b writes a line of such a row in two parts far apart (T = 217: 4.33 ms against 3.24 on the generic kernel), and the carry
// that cures it in k_stft_ft16c -- every thread's X[k], X[M - k] of the previous tile -- would be 128 VGPRs for both bands.  The two bands
// never meet (k_stft_ft16b), so here a workgroup owns ONE band of a clip segment: it walks the segment's tiles in order, forms only its
// band from the samples (the other band's workgroup reads the same samples at about the same time on the same XCD: units (segment, band 0)
// and (segment, band 1) are neighbours in the XCD order), and carries its own 1024 rows of the previous tile in 64 VGPRs exactly as
// k_stft_ft16c does: lanes tt < 16 - a store the current value at frame t0 + tt, lanes tt >= 16 - a the carried one at t0 - 16 + tt --
// sixteen lanes, one whole line, streamed.  The raw samples of a frame are 64 VGPRs per lane, so they are requested one FRAME ahead
// (frame 1 of the tile under the transform of frame 0, frame 0 of the next tile under the transform of frame 1 and the stores), not one
// tile ahead.
template <bool ALIGNED, int SPEC>
__global__ __launch_bounds__(kFatWaves * 64) void k_stft_ft16bc(
    const float* __restrict__ x, const float* __restrict__ win, const float2* __restrict__ twp,
    const float2* __restrict__ tws, float2* __restrict__ out, long long n_samples, int hop, int T, int TP, int tiles,
    int segs, int seg_tiles, int units) {
    static_assert(SPEC < 2, "complex spectra only");
    using B = BandCfg;
    using C = B::C;
    constexpr int N = B::N, M = B::M, W = B::W, P = 64, E = 16, NT = B::NT, FPB = B::FPB, FPW = B::FPW, PITCH = B::PITCH;
    constexpr int ROWS = SPEC ? M + 1 : W;
    constexpr int ITER = (N / 2) / (NT / FPB);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* frames = reinterpret_cast<float2*>(smem_raw);
    float2* tw_l = frames + FPB * PITCH;
    float2* tws_l = tw_l + C::TW;
    const int tid = threadIdx.x;
    for (int i = tid; i < C::TW; i += NT) tw_l[i] = twp[i];
    for (int i = tid; i <= M / 2; i += NT) tws_l[i] = tws[i];
    __syncthreads();
    const int wave = tid / P, p_lane = tid % P;
    const int tt = tid % FPB, kq = tid / FPB;
    const float2* fb = frames + tt * PITCH;
    const bool xcd = ZAFX_XCD_ORDER && gridDim.x % 8 == 0;
    const int b0 = (int)((reinterpret_cast<uintptr_t>(out) >> 3) & 15);   // phase of the array's first element in its line
    const float2 wp = tws_l[2 * p_lane];   // exp(-2 pi i lane / M)

    // raw samples z[lane + 64 j], j < 32, of frame `fr` of tile `tile` of `clip` -> dst
    auto request = [&](float2 (&dst)[2 * E], int clip, int tile, int fr) {
        const float* xc = x + (long long)clip * n_samples;
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(xc, (unsigned)(n_samples * 4));   // (outside the clip: zero = the reference's padding)
        const int s0 = (tile * FPB + fr) * hop - M;
        const int vo = (s0 + 2 * p_lane) * 4;
        int vo1 = vo + 4;
        if constexpr (!ALIGNED) asm volatile("" : "+v"(vo1));   // (see k_stft_ft16b: the two 4-byte loads of a pair must not be merged)
#pragma unroll
        for (int j = 0; j < 2 * E; ++j) {
            if constexpr (ALIGNED) {
                dst[j] = buf_load_f32x2(rs, vo + j * P * 8);
            } else {
                dst[j].x = buf_load_f32(rs, vo + j * P * 8);
                dst[j].y = buf_load_f32(rs, vo1 + j * P * 8);
            }
        }
    };
    // unit v of this workgroup's walk -> clip, band and tile range [j, j1) of its segment
    auto unit_of = [&](int v, int& clip, int& band, int& j, int& j1) {
        const int u = xcd ? xcd_order(v, units) : v;
        band = u & 1;
        const int sg = u >> 1;
        clip = sg / segs;
        j = (sg % segs) * seg_tiles;
        j1 = min(j + seg_tiles, tiles);
    };
    int v = blockIdx.x;
    if (v >= units) return;
    int clip, band, j, j1;
    unit_of(v, clip, band, j, j1);
    float2 xa[2 * E], xb[2 * E];
    request(xa, clip, j, wave * FPW);
    float2 ck[ITER], cn[ITER];   // the thread's X[k], X[M - k] of the previous tile (k = 2 (kq + 32 it) + band)
#pragma unroll
    for (int it = 0; it < ITER; ++it) ck[it] = cn[it] = make_float2(0.f, 0.f);
    bool have_prev = false;
    for (;;) {
        int po = p_lane;
        asm volatile("" : "+v"(po));
        // the walk's next tile
        int nclip = clip, nband = band, nj = j + 1, nj1 = j1, nv = v;
        bool more = true;
        if (nj >= j1) {
            nv = v + gridDim.x;
            if (nv < units) unit_of(nv, nclip, nband, nj, nj1);
            else more = false;
        }
        // this band of one frame, in place: src[i] <- a + b (band 0) or (a - b) exp(-2 pi i (lane + 64 i) / M) (band 1)
        auto form = [&](float2 (&src)[2 * E]) {
            const float2* w2 = reinterpret_cast<const float2*>(win) + po;
            const float c32[16] = {1.f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f, 0.70710678118654752440f,
                                   0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f, 0.f,
                                   -0.19509032201612826785f, -0.38268343236508977173f, -0.55557023301960222474f, -0.70710678118654752440f,
                                   -0.83146961230254523708f, -0.92387953251128675613f, -0.98078528040323044913f};
            const float s32[16] = {0.f, 0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f, 0.70710678118654752440f,
                                   0.83146961230254523708f, 0.92387953251128675613f, 0.98078528040323044913f, 1.f,
                                   0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f, 0.70710678118654752440f,
                                   0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f};
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const float2 wa = w2[i * P], wb = w2[(i + E) * P];
                const float2 a = make_float2(src[i].x * wa.x, src[i].y * wa.y);
                const float2 b = make_float2(src[i + E].x * wb.x, src[i + E].y * wb.y);
                if (band == 0) {   // (uniform)
                    src[i] = cadd(a, b);
                } else {
                    const float2 d = csub(a, b);
                    src[i] = cmul(i == 0 ? d : (i == 8 ? mul_mi(d) : cmulk(d, c32[i], -s32[i])), wp);
                }
            }
        };
        form(xa);
        // frame 1 of this tile, under the transform of frame 0 (requested AHEAD of form(xa) its 32 loads sit in front of frame 0's window
        // loads in the in-order return queue: 2.50 against 2.37 ms)
        request(xb, clip, j, wave * FPW + 1);
        fft_frame<10, 4>(&xa[0], frames + (wave * FPW) * PITCH, po, tw_l);
        __builtin_amdgcn_sched_barrier(0);
        form(xb);
        if (more) request(xa, nclip, nj, wave * FPW);             // frame 0 of the next tile, under the transform of frame 1 and the stores
        fft_frame<10, 4>(&xb[0], frames + (wave * FPW + 1) * PITCH, po, tw_l);
        lds_barrier();
        {
            const int t0 = j * FPB;
            const bool last = j + 1 >= j1, cur_ok = t0 + tt < T;
            float2* o = out + (long long)clip * ROWS * TP + (t0 + tt);
            const int c0 = (int)(((long long)clip * ROWS) & 15), tp = TP & 15;
            int kqo = kq;
            asm volatile("" : "+v"(kqo));
            auto sweep = [&](auto stream) {
                constexpr bool ST = decltype(stream)::value;
                auto emit = [&](int r, float2 cur, float2 prev) {   // one row: `cur` = this tile's value of the thread's frame, `prev` = the carried one
                    const int a = (b0 + (c0 + r) * tp) & 15;        // the row's run starts a frames into a line
                    const bool from_prev = tt >= 16 - a;            // (a = 0: never -- the run is a whole line)
                    const float2 val = from_prev ? prev : cur;
                    float2* dst = o + (long long)r * TP;
                    if (from_prev ? have_prev : cur_ok) {
                        if constexpr (ST) store_stream(dst + (from_prev ? -16 : 0), val);
                        else dst[from_prev ? -16 : 0] = val;
                    }
                    if (last && from_prev && cur_ok) *dst = cur;    // tail of the segment's last run (a partial line)
                };
#pragma unroll
                for (int it = 0; it < ITER; ++it) {
                    const int q = kqo + it * (NT / FPB);
                    float2 xk, xn;
                    if (it == 0 && q == 0 && band == 0) {
                        const float2 z0 = fb[0], zc = fb[phys_t<C::PS>(N / 2)];   // Z[0], Z[M / 2]
                        xk = cconj(zc);                                 // row M/2 (its mirror, row 3M/2: zc)
                        xn = make_float2(z0.x + z0.y, z0.x - z0.y);     // rows 0 and M: both real, carried as one pair
                        emit(M / 2, xk, ck[0]);
                        if (SPEC == 0) emit(M + M / 2, cconj(xk), cconj(ck[0]));
                        emit(0, make_float2(xn.x, 0.f), make_float2(cn[0].x, 0.f));
                        emit(M, make_float2(xn.y, 0.f), make_float2(cn[0].y, 0.f));
                    } else {
                        const int k = 2 * q + band;
                        split_pair(fb[phys_t<C::PS>(q)], fb[phys_t<C::PS>(N - band - q)], tws_l[k], xk, xn);
                        emit(k, xk, ck[it]);
                        if (SPEC == 0) emit(W - k, cconj(xk), cconj(ck[it]));
                        emit(M - k, xn, cn[it]);
                        if (SPEC == 0) emit(M + k, cconj(xn), cconj(cn[it]));
                    }
                    ck[it] = xk;
                    cn[it] = xn;
                    __builtin_amdgcn_sched_barrier(0);   // (iterations stay apart: their LDS reads hoisted ahead cost the registers the carry needs)
                }
            };
            if (have_prev) sweep(std::true_type{});
            else sweep(std::false_type{});
        }
        lds_barrier();
        if (!more) break;
        have_prev = nj != 0 && nv == v;   // the walk continues inside the same segment (and band)
        clip = nclip, band = nband, j = nj, j1 = nj1, v = nv;
    }
}

// ---------------------------------------------------------------------------------
// forward, frame-major layout (ZAFX_LAYOUT_TF), persistent and barrier free
// ---------------------------------------------------------------------------------
// Every frame's 2 W bins are contiguous in this layout, so a frame never has to meet its
// neighbours: one wavefront (half of one for 1024 points, two radix-32 passes) owns a frame from load to store (private LDS
// exchange buffer, coalesced stores) and the wavefronts of the persistent workgroup drift apart -- the store
// issue of one overlaps the butterflies of another.  No s_barrier after the table staging.
// (16 waves at 100 VGPRs: 1.87 ms; 8 waves: 1.95 ms on the same box.)
#ifndef ZAFX_TF_WAVES
#define ZAFX_TF_WAVES 16
#endif
constexpr int kTfWaves = ZAFX_TF_WAVES;

#ifndef ZAFX_TF_R32
#define ZAFX_TF_R32 1   // 1024 x 10 s: 2.01 ms against 2.11 ms for the one-wavefront radix-16 schedule on the same box
#endif
// points per thread / threads per workgroup of k_stft_tf: radix-32 schedule (8 waves, a frame per half wavefront) for
// 1024 points when enabled, else one frame per wavefront and kTfWaves wavefronts
constexpr int tf_log2e(int log2n) { return (ZAFX_TF_R32 && log2n == 10) ? 5 : default_log2e(log2n); }
constexpr int tf_threads(int log2n) { return (ZAFX_TF_R32 && log2n == 10) ? 512 : kTfWaves * 64; }

template <int LOG2N, int LOG2E, bool ALIGNED, int SPEC>
__global__ __launch_bounds__(tf_threads(LOG2N)) void k_stft_tf(
    const float* __restrict__ x, const float* __restrict__ win, const float2* __restrict__ twp,
    const float2* __restrict__ tws, float2* __restrict__ out, long long n_samples, int hop, int T, long long total_frames) {
    using C = FftCfg<LOG2N, LOG2E>;
    static_assert(C::P == 64 || C::P == 32, "a frame is owned by one wavefront or by half of one");
    constexpr int N = C::N, P = C::P, E = C::E, W = 2 * N, NT = tf_threads(LOG2N), WAVES = NT / P;   // WAVES = frame slots
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* frames = reinterpret_cast<float2*>(smem_raw);
    float2* tw_l = frames + WAVES * C::PITCH;
    float2* win_l = tw_l + C::TW;
    float2* tws_l = win_l + N;
    const int tid = threadIdx.x;
    for (int i = tid; i < C::TW; i += NT) tw_l[i] = twp[i];
    for (int i = tid; i < N; i += NT) win_l[i] = reinterpret_cast<const float2*>(win)[i];
    for (int i = tid; i <= N / 2; i += NT) tws_l[i] = tws[i];
    __syncthreads();
    const int wave = tid / P, p_lane = tid % P;
    float2* buf = frames + wave * C::PITCH;
    const long long stride = (long long)gridDim.x * WAVES;
    for (long long g = (long long)blockIdx.x * WAVES + wave; g < total_frames; g += stride) {
        const long long clip = g / T;
        const int t = (int)(g - clip * T);
        const float* xc = x + clip * n_samples;
        const long long s0 = (long long)t * hop - N;
        int p = p_lane;   // opaque copy: per-lane offsets and table reads are recomputed per frame, not hoisted and spilled
        asm volatile("" : "+v"(p));
        float2 v[E];
        if (ALIGNED && s0 >= 0 && s0 + W <= n_samples) {
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const int n = p + i * P;
                const float2 xv = *reinterpret_cast<const float2*>(xc + s0 + 2 * n);
                const float2 wv = win_l[n];
                v[i] = make_float2(xv.x * wv.x, xv.y * wv.y);
            }
        } else {
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const int n = p + i * P;
                const long long s = s0 + 2 * n;
                const float2 wv = win_l[n];
                const float a = (s >= 0 && s < n_samples) ? xc[s] : 0.f;
                const float b = (s + 1 >= 0 && s + 1 < n_samples) ? xc[s + 1] : 0.f;
                v[i] = make_float2(a * wv.x, b * wv.y);
            }
        }
        fft_frame<LOG2N, LOG2E>(v, buf, p, tw_l);
        float2* o = spec_base<SPEC>(out, g * (SPEC ? N + 1 : W));
#pragma unroll
        for (int i = 0; i < E / 2; ++i) {
            const int k = p + i * P;
            if (k == 0) {
                const float2 z0 = buf[0], zc = buf[phys_t<C::PS>(N / 2)];
                put_bin<SPEC>(o, 0, make_float2(z0.x + z0.y, 0.f));
                put_bin<SPEC>(o, N, make_float2(z0.x - z0.y, 0.f));
                put_bin<SPEC>(o, N / 2, cconj(zc));
                if (SPEC == 0) put_bin<SPEC>(o, N + N / 2, zc);
            } else {
                float2 xk, xn;
                split_pair(buf[phys_t<C::PS>(k)], buf[phys_t<C::PS>(N - k)], tws_l[k], xk, xn);
                put_bin<SPEC>(o, k, xk);
                put_bin<SPEC>(o, N - k, xn);
                if (SPEC == 0) {
                    put_bin<SPEC>(o, W - k, cconj(xk));
                    put_bin<SPEC>(o, N + k, cconj(xn));
                }
            }
        }
        frame_sync<P>();   // the split reads of this frame precede the next frame's first pass writes
    }
}

// ---------------------------------------------------------------------------------
// inverse
// ---------------------------------------------------------------------------------
// Build the packed half-length spectrum of one (k, N-k) pair from the four two-sided
// bins, for an ARBITRARY input (the reference takes real(ifft(.)) of anything,
// zaf.py:223).  Unscaled by 4; stored re<->im swapped so that a FORWARD transform
// yields the inverse one.
__device__ __forceinline__ void unsplit_pair(float2 xk, float2 xwk, float2 xnk, float2 xnpk, float2 t,
                                             float2& zk, float2& zn) {
    const float2 ak = make_float2(xk.x + xwk.x, xk.y - xwk.y);       // X[k] + conj X[W-k]
    const float2 an = make_float2(xnk.x + xnpk.x, xnk.y - xnpk.y);   // X[N-k] + conj X[N+k]
    const float2 e = make_float2(ak.x + an.x, ak.y - an.y);          // a_k + conj a_{N-k}
    const float2 d = make_float2(ak.x - an.x, ak.y + an.y);          // a_k - conj a_{N-k}
    const float2 o = cmulc(d, t);                                    // d * conj(t_k)
    // Z[k] = e + i o ; Z[N-k] = conj(e) + i conj(o)
    const float2 Zk = make_float2(e.x - o.y, e.y + o.x);
    const float2 Zn = make_float2(e.x + o.y, -e.y + o.x);
    zk = make_float2(Zk.y, Zk.x);
    zn = make_float2(Zn.y, Zn.x);
}

// ONE = one-sided input (ZAFX_SPECTRUM_ONE_SIDED): rows 0..W/2, completed as X[W-k] = conj X[k].
template <int LOG2N, int LOG2E, int FPB, int LAYOUT, bool ONE>
__global__ __launch_bounds__(FPB * fft_threads(LOG2N, LOG2E)) void k_istft(
    const float2* __restrict__ spec, const float2* __restrict__ twp, const float2* __restrict__ tws,
    float* __restrict__ y, int T, int TP, int hop, long long out_len, float scale, int tiles, int owned, int halo) {
    using C = FftCfg<LOG2N, LOG2E>;
    using S = StftCfg<LOG2N, LOG2E, FPB>;
    constexpr int N = C::N, P = C::P, E = C::E, W = 2 * N, NT = S::NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* frames = reinterpret_cast<float2*>(smem_raw);
    const float2* tw = twp;
    const int tid = threadIdx.x;
    if constexpr (S::TW_LDS) {
        float2* tw_l = frames + FPB * C::PITCH;
        for (int i = tid; i < C::TW; i += NT) tw_l[i] = twp[i];
        tw = tw_l;
    }
    const int bid = ZAFX_XCD_ORDER ? xcd_order((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;   // neighbouring tiles to one XCD: their partial lines merge in its L2
    const int clip = bid / tiles, tile = bid % tiles;
    const int t_first = tile * owned - halo;   // frame held by slot 0 (may be < 0)

    // ---- phase A: gather the four two-sided bins of every pair, write packed Z to LDS
    {
        int fs, kq, kstep;
        if constexpr (LAYOUT == ZAFX_LAYOUT_TF) { fs = tid / P; kq = tid % P; kstep = P; }
        else { fs = tid % FPB; kq = tid / FPB; kstep = P; }
        const int t = t_first + fs;
        if (fs < owned + halo && t >= 0 && t < T) {
            float2* fb = frames + fs * C::PITCH;
            constexpr int ROWS = ONE ? N + 1 : W;
            long long base, kstride;
            if constexpr (LAYOUT == ZAFX_LAYOUT_TF) { base = ((long long)clip * T + t) * ROWS; kstride = 1; }
            else { base = (long long)clip * ROWS * TP + t; kstride = TP; }
            const float2* sp = spec + base;
            for (int k = kq; k < N / 2; k += kstep) {
                if (k == 0) {
                    const float a0 = 2.f * sp[0].x, an = 2.f * sp[(long long)N * kstride].x;
                    // Z[0] = (a0 + aN) + i (a0 - aN), stored swapped
                    fb[0] = make_float2(a0 - an, a0 + an);
                    const float2 xc = sp[(long long)(N / 2) * kstride];
                    const float2 xd = ONE ? cconj(xc) : sp[(long long)(N + N / 2) * kstride];
                    // A = X[N/2] + conj X[3N/2]; Z[N/2] = 2 conj(A), stored swapped
                    const float2 a = make_float2(xc.x + xd.x, xc.y - xd.y);
                    fb[phys(N / 2)] = make_float2(-2.f * a.y, 2.f * a.x);
                } else {
                    const float2 xk = sp[(long long)k * kstride], xnk = sp[(long long)(N - k) * kstride];
                    const float2 xwk = ONE ? cconj(xk) : sp[(long long)(W - k) * kstride];
                    const float2 xnpk = ONE ? cconj(xnk) : sp[(long long)(N + k) * kstride];
                    float2 zk, zn;
                    unsplit_pair(xk, xwk, xnk, xnpk, tws[k], zk, zn);
                    fb[phys(k)] = zk;
                    fb[phys(N - k)] = zn;
                }
            }
        }
    }
    __syncthreads();

    // ---- phase B: forward FFT of the swapped spectrum == swapped inverse FFT
    {
        const int slot = tid / P, p = tid % P;
        float2* buf = frames + slot * C::PITCH;
        float2 v[E];
        regs_read<LOG2N, LOG2E>(v, buf, p);
        frame_sync<P>();
        fft_frame<LOG2N, LOG2E>(v, buf, p, tw);
    }
    __syncthreads();

    // ---- phase C: gather overlap-add in ascending frame order (zaf.py:226-233), trim (:236-238),
    //      COLA gain (:241); no atomics, every output sample is written exactly once
    {
        const float* fl = reinterpret_cast<const float*>(frames);
        const int t_end = min((tile + 1) * owned, T);
        const long long s_begin = (long long)tile * owned * hop;
        const long long s_end = (tile == tiles - 1) ? (long long)T * hop + (W - hop) : (long long)t_end * hop;
        float* yc = y + (long long)clip * out_len;
        for (long long s = s_begin + tid; s < s_end; s += NT) {
            const long long o = s - (W - hop);
            if (o < 0 || o >= out_len) continue;
            const int j_hi = (int)min((long long)(T - 1), s / hop);
            const int j_lo = s >= W ? (int)((s - W) / hop) + 1 : 0;
            float acc = 0.f;
            for (int j = j_lo; j <= j_hi; ++j) {
                const int n = (int)(s - (long long)j * hop);
                const int f = (2 * phys(n >> 1) + (n & 1)) ^ 1;   // swapped components
                acc += fl[(size_t)(j - t_first) * (2 * C::PITCH) + f];
            }
            yc[o] = acc * scale;
        }
    }
}

// ---------------------------------------------------------------------------------
// inverse, reference (frequency-major) layout, persistent carry form
// ---------------------------------------------------------------------------------
// One persistent 8-wave workgroup per CU walks the 16-frame tiles of a clip segment IN ORDER and
// keeps the overlap of the last frames with the next tile (W - hop samples, the "carry") in LDS.
// Tiles therefore start at t = 16 * tile: every gathered row piece is a 128-B aligned run and no
// frame is read twice (the halo form re-read ceil(W/H)-1 frames per tile and straddled two lines
// per run).  A segment that does not start a clip first runs the tile before it in carry-only mode
// (only its last `halo` frames are loaded and transformed, nothing is written).  The carry slot c
// is read and rewritten by the same thread (c = tid mod NT), and the adds keep the reference's
// ascending frame order (zaf.py:226-233), so the result is deterministic.
//
// The NEXT tile's two-sided bins are prefetched into registers in two halves: the first half is
// issued before the FFT phase, the second before the overlap-add phase (when the FFT registers are
// dead), and both are folded into LDS after the stores -- the gather latency hides under the FFT
// and the store phase.  (All 64 x 8 B per lane at once, at 16 waves, spilled: 7.4 ms vs 3.4 ms.)
// ---------------------------------------------------------------------------------
// gather overlap-add of a tile held in LDS
// ---------------------------------------------------------------------------------
// Sample offset c = q * hop + r from the tile's first frame.  The frames of the tile that cover it
// are jl = q - d with r + d * hop < W; they are added in ascending jl (the reference's order,
// zaf.py:226-233), clipped to [0, n_valid).  `fl` is the tile (pitch2 floats per frame, re/im
// swapped packed real output of the inverse transform).
template <int W>
__device__ __forceinline__ float ola_sum(const float* fl, int pitch2, int q, int r, int hop, int n_valid, float acc) {
    int d = 0;
    while (r + (d + 1) * hop < W) ++d;
    int jl = q - d, n = r + d * hop;
    if (jl < 0) {
        n += jl * hop;
        jl = 0;
    }
    const int j_hi = min(q, n_valid - 1);
    for (; jl <= j_hi; ++jl, n -= hop) acc += fl[jl * pitch2 + ((2 * phys(n >> 1) + (n & 1)) ^ 1)];
    return acc;
}

// Phase C of the carry kernel: outputs of the tile (carry first), then the carry for the next tile.
// Thread `tid` owns offsets c = tid + m NT in both loops, so a carry slot is read and rewritten by
// one thread.  q/r advance incrementally (no integer division per sample).
// OS (template parameter of the phases below) = 2: the tile holds one BAND of a W = 4096 frame's samples (k_istft_ft16b): local
// sample pair m is the real pair (4 m + boff, 4 m + boff + 1), i.e. local offset o maps to (o >> 1) * 4 + (o & 1) + boff; out_len stays real.
struct OlaArgs {
    int boff = 0;
    const float* fl;      // tile
    float* carry;
    int pitch2, ncarry, hop, n_valid, c_end;
    bool write_out, make_carry;
    float* yc;            // this clip's output
    long long o_first;    // output index of offset 0 (may be negative: trimmed head)
    long long out_len;
    float scale;
};

template <int W, int NT, int FPB, int OS = 1>
__device__ __forceinline__ void ola_phase(const OlaArgs& a, int tid) {
    const int q0 = tid / a.hop, r0 = tid % a.hop, qstep = NT / a.hop, rstep = NT % a.hop;
    if (a.write_out) {
        int q = q0, r = r0;
        for (int c = tid; c < a.c_end; c += NT) {
            float acc = c < a.ncarry ? a.carry[c] : 0.f;
            acc = ola_sum<W>(a.fl, a.pitch2, q, r, a.hop, a.n_valid, acc);
            const long long ol = a.o_first + c;
            const long long o = OS == 1 ? ol : (ol >> 1) * 4 + (ol & 1) + a.boff;
            if (ol >= 0 && o < a.out_len) a.yc[o] = acc * a.scale;
            q += qstep;
            r += rstep;
            if (r >= a.hop) r -= a.hop, ++q;
        }
    }
    if (a.make_carry) {
        int q = q0 + FPB, r = r0;
        for (int c = tid; c < a.ncarry; c += NT) {
            a.carry[c] = ola_sum<W>(a.fl, a.pitch2, q, r, a.hop, a.n_valid, 0.f);
            q += qstep;
            r += rstep;
            if (r >= a.hop) r -= a.hop, ++q;
        }
    } else {
        for (int c = tid; c < a.ncarry; c += NT) a.carry[c] = 0.f;   // next tile starts another segment
    }
}

// The same for an even hop >= W/2: at most two frames cover a sample (q - 1 and q), and the samples
// (2m, 2m+1) of a frame sit in one LDS float2 -- branch-free 8-byte LDS reads, 8-byte stores.
template <int W, int NT, int FPB, int OS = 1>
__device__ __forceinline__ void ola_phase_pairs(const OlaArgs& a, int tid, bool y_aligned) {
    const float2* fr = reinterpret_cast<const float2*>(a.fl);
    float2* carry2 = reinterpret_cast<float2*>(a.carry);
    const int pitch = a.pitch2 / 2, hop2 = a.hop / 2, ncarry2 = a.ncarry / 2, c_end2 = a.c_end / 2;
    const int q0 = tid / hop2, r0 = tid % hop2, qstep = NT / hop2, rstep = NT % hop2;
    auto sum2 = [&](int q, int r, float2 acc) {   // r in pairs
        const bool prev_ok = q >= 1 && q - 1 < a.n_valid && r + hop2 < W / 2, cur_ok = q < a.n_valid;
        const float2 u = prev_ok ? fr[(q - 1) * pitch + phys(r + hop2)] : make_float2(0.f, 0.f);
        const float2 v = cur_ok ? fr[q * pitch + phys(r)] : make_float2(0.f, 0.f);
        return make_float2((acc.x + u.y) + v.y, (acc.y + u.x) + v.x);   // components are stored swapped
    };
    if (a.write_out) {
        int q = q0, r = r0;
#pragma unroll 4
        for (int c = tid; c < c_end2; c += NT) {
            float2 acc = c < ncarry2 ? carry2[c] : make_float2(0.f, 0.f);
            acc = sum2(q, r, acc);
            const long long ol = a.o_first + 2 * c;
            const long long o = OS == 1 ? ol : ol * 2 + a.boff;
            if (ol >= 0) {
                if (y_aligned && o + 1 < a.out_len) {
                    *reinterpret_cast<float2*>(a.yc + o) = make_float2(acc.x * a.scale, acc.y * a.scale);
                } else {
                    if (o < a.out_len) a.yc[o] = acc.x * a.scale;
                    if (o + 1 < a.out_len) a.yc[o + 1] = acc.y * a.scale;
                }
            }
            q += qstep;
            r += rstep;
            if (r >= hop2) r -= hop2, ++q;
        }
    }
    if (a.make_carry) {
        int q = q0 + FPB, r = r0;
        for (int c = tid; c < ncarry2; c += NT) {
            carry2[c] = sum2(q, r, make_float2(0.f, 0.f));
            q += qstep;
            r += rstep;
            if (r >= hop2) r -= hop2, ++q;
        }
    } else {
        for (int c = tid; c < ncarry2; c += NT) carry2[c] = make_float2(0.f, 0.f);
    }
}

// The same for hop = W/2 with NT a multiple of hop/2, full tile inside the clip: thread tid owns the sample pair r = tid mod hop/2
// of the frames q0, q0 + QS, ... (QS = 2 NT / hop frames per pass of the workgroup), so its two LDS slots, the frame it starts
// with and its output pointer are fixed: two 8-byte LDS reads, three adds, one scale and one 8-byte store per pass, unrolled
// (ola_phase_pairs re-derives frame, offset and four predicates per pair: 5-7 k of the tile's 35 k cycles at W = 2048).  The
// same additions in the same order: bit-identical.  Returns false when the tile is not of that kind.
template <int W, int NT, int FPB, int OS = 1>
__device__ __forceinline__ bool ola_phase_sweep(const OlaArgs& a, int tid, bool y_aligned) {
    constexpr int HOP2 = W / 4, QS = NT / HOP2, NIT = (NT % HOP2 == 0 && QS >= 1 && FPB % (QS > 0 ? QS : 1) == 0) ? FPB / (QS > 0 ? QS : 1) : 0;
    if constexpr (NIT == 0 || HOP2 % 64 != 0) {
        return false;
    } else {
        const bool full = a.write_out && a.make_carry && 2 * a.hop == W && a.n_valid == FPB && a.c_end == FPB * a.hop &&
                          (a.o_first + (long long)FPB * a.hop) * OS <= a.out_len;
        if (!full) return false;
        const float2* fr = reinterpret_cast<const float2*>(a.fl);
        float2* carry2 = reinterpret_cast<float2*>(a.carry);
        const int pitch = a.pitch2 / 2;
        int to = tid;   // opaque: slots and pointers are recomputed per tile (carried through the transforms they would spill)
        asm volatile("" : "+v"(to));
        const int r = to % HOP2;
        const int q0 = __builtin_amdgcn_readfirstlane(to / HOP2);   // (whole waves: HOP2 is a multiple of 64)
        const float2* pv = fr + q0 * pitch + phys(r);                    // frame q, first half
        const float2* pu = fr + (q0 - 1) * pitch + phys(r + HOP2);       // frame q - 1, second half
        float* dst = a.yc + (a.o_first + 2 * to) * OS + (OS == 1 ? 0 : a.boff);
        const bool skip0 = a.o_first < 0 && q0 == 0;   // (the clip's first tile: its first W - hop samples are trimmed)
        // (compiled per alignment of the clip's output: with the test at every store the compiler folded both forms into 4-byte stores)
        auto passes = [&](auto ALIGNED8) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                float2 acc = make_float2(0.f, 0.f), u = make_float2(0.f, 0.f);
                if (it == 0 && q0 == 0) acc = carry2[r];
                else u = pu[(size_t)it * QS * pitch];
                const float2 v = pv[(size_t)it * QS * pitch];
                const float2 s = make_float2((acc.x + u.y) + v.y, (acc.y + u.x) + v.x);   // components are stored swapped
                if (!(it == 0 && skip0)) {
                    float* d = dst + (size_t)it * 2 * NT * OS;
                    if constexpr (decltype(ALIGNED8)::value) {
                        *reinterpret_cast<float2*>(d) = make_float2(s.x * a.scale, s.y * a.scale);
                    } else {
                        d[0] = s.x * a.scale;
                        d[1] = s.y * a.scale;
                    }
                }
            }
        };
        if (y_aligned) passes(std::true_type{});
        else passes(std::false_type{});
        // carry: the second half of the tile's last frame (ola_phase_pairs: (0 + u.y) + 0, (0 + u.x) + 0)
        if (q0 == 0) {
            const float2 ul = fr[(FPB - 1) * pitch + phys(r + HOP2)];
            carry2[r] = make_float2((0.f + ul.y) + 0.f, (0.f + ul.x) + 0.f);
        }
        return true;
    }
}

ZAFX_PROF_ARRAY(g_prof)

// ---------------------------------------------------------------------------------
// inverse, reference (frequency-major) layout, persistent carry form
// ---------------------------------------------------------------------------------
// One persistent 16-wave workgroup per CU walks the 16-frame tiles of a clip segment IN ORDER and
// keeps the overlap of the last frames with the next tile (W - hop samples, the "carry") in LDS.
// Tiles therefore start at t = 16 * tile: every gathered row piece is a 128-B aligned run and no
// frame is read twice (the halo form re-read ceil(W/H)-1 frames per tile and straddled two lines
// per run).  A segment that does not start a clip first runs the tile before it in carry-only mode
// (only its last `halo` frames are loaded and transformed, nothing is written).  The carry slot c
// is read and rewritten by the same thread, and the adds keep the reference's ascending frame
// order (zaf.py:226-233), so the result is deterministic.
//
// Measured on MI355X (profiles/r01_notes.md): the gather runs at HBM speed only with SHALLOW
// per-wave queues (16 waves x <= 8 loads; 8 waves x 64 loads of register prefetch ran the same bytes
// 2x slower), so the tile's sweeps are streamed DEPTH at a time straight into the Hermitian fold
// (round 1: prefetching 1-3 sweeps of the next tile across the FFT phase did not pay, the FFT needed
// 116 of the 128 VGPRs; round 2, with the second exchange in registers: ONE sweep rides across the
// transforms and the overlap-add, 1.945 -> 1.92 ms; two sweeps spill, 1.99 ms; a second sweep requested
// when the transforms are done spills as well, 1.82 -> 1.91 ms).  Barriers order LDS only (lds_barrier): the output stores of a tile are not
// waited for.
template <int LOG2N, int LOG2E, int DEPTH, bool ONE, int FV, bool TF = false>
__global__ __launch_bounds__(1024) void k_istft_ft16(
    const float2* __restrict__ spec, const float2* __restrict__ twp, const float2* __restrict__ tws,
    float* __restrict__ y, int T, int TP, int hop, long long out_len, float scale, int tiles, int segs, int seg_tiles,
    int total_units, int halo) {
    using C = FftCfg<LOG2N, LOG2E>;
    using F = FatCfg<LOG2N, LOG2E>;
    constexpr int N = C::N, P = C::P, E = C::E, W = 2 * N, NT = 1024, FPB = kFatFrames, PITCH = F::PITCH;
    constexpr int ROWS = ONE ? N + 1 : W;      // one-sided input: rows 0..N, X[W-k] = conj X[k]
    // FV frames per lane and load: 2 = 16-byte loads of two adjacent frames (needs an even row pitch).  The CU's
    // vector-memory queue holds ~64 wave-level loads whatever their width, so 16-byte lanes double the
    // bytes in flight (64 KB) and with them the gather rate of a CU that is alone in its load phase.
    constexpr int LPR = FPB / FV;              // lanes per row run
    constexpr int KSTEP = NT / LPR;            // bins handled per sweep
    constexpr int KI = (N / 2) / KSTEP;        // sweeps per thread
    static_assert((FV == 1 || FV == 2) && KI >= 1 && (N / 2) % KSTEP == 0, "pair sweep must divide N/2");
    static_assert(!TF || (FV == 1 && P == 64), "frame-major input: one wavefront loads, folds and transforms a whole frame");
    using RV = std::conditional_t<FV == 2, float4, float2>;   // one row piece of my FV frames
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* frames = reinterpret_cast<float2*>(smem_raw);
    float2* tw_l = frames + FPB * PITCH;
    float2* tws_l = tw_l + C::TW;   // N/2 + 1 roots of W
    float* carry = reinterpret_cast<float*>(tws_l + N / 2 + 1);   // W - hop floats (<= N float2)
    const int tid = threadIdx.x;
    for (int i = tid; i < C::TW; i += NT) tw_l[i] = twp[i];
    for (int i = tid; i <= N / 2; i += NT) tws_l[i] = tws[i];
    const int wave = tid / P, p = tid % P;
    const int fs = (tid % LPR) * FV, kq = tid / LPR;   // my first frame of the tile, my bin within a sweep
    float2* fbuf = frames + fs * PITCH;
    const int ncarry = W - hop;
    const bool pairs = hop % 2 == 0 && 2 * hop >= W;
    const bool y_base_aligned = reinterpret_cast<uintptr_t>(y) % 8 == 0;

    struct Tile {
        int unit, tile, tile_a, tile_b;
    };
    auto enter = [&](Tile& it) {   // first tile of it.unit (one before the segment when it needs a carry)
        const int seg = it.unit % segs;
        it.tile_a = seg * seg_tiles;
        it.tile_b = min(it.tile_a + seg_tiles, tiles);
        it.tile = it.tile_a > 0 ? it.tile_a - 1 : 0;
    };
    auto my_frame_needed = [&](const Tile& it) {   // (FV = 2: the pitch is even, so a pair's second frame is in the row; past T it is unused)
        return it.tile * FPB + fs < T && fs + FV - 1 >= (it.tile < it.tile_a ? FPB - halo : 0);
    };
    // Rows k, W-k, N-k, N+k of sweep s (k = 0: rows 0, N/2, N, 3N/2) for my frame(s).  Buffer loads: the
    // clip's descriptor and the sweep's row offsets are wave-uniform (SGPRs), the per-lane part is two
    // 32-bit offsets for the whole tile -- 64-bit flat addresses would cost 8 VGPRs per sweep in flight.
    const int row_bytes = TP * 8;   // TP = row pitch in frames (>= T)
    const int v_up = kq * row_bytes + fs * 8, v_down = (KSTEP - kq) * row_bytes + fs * 8;
    struct Src {
        __amdgpu_buffer_rsrc_t rsrc;
        int t_bytes;   // byte offset of the tile's first frame within a row
    };
    auto source = [&](const Tile& it) {
        Src src;
        src.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(spec) + (long long)(it.unit / segs) * ROWS * TP, 0,
                                                     ROWS * row_bytes, 0x00020000);
        src.t_bytes = it.tile * FPB * 8;
        return src;
    };
    auto ld = [&](const Src& src, int voff, int soff) {
        RV f;
        if constexpr (FV == 2) {
            const auto raw = __builtin_amdgcn_raw_buffer_load_b128(src.rsrc, voff, soff + src.t_bytes, 0);
            __builtin_memcpy(&f, &raw, 16);
        } else {
            const auto raw = __builtin_amdgcn_raw_buffer_load_b64(src.rsrc, voff, soff + src.t_bytes, 0);
            __builtin_memcpy(&f, &raw, 8);
        }
        return f;
    };
    // (one-sided input: only rows k and N-k are loaded -- plus row N/2 for the lane that holds k = 0 --
    // and fold4 completes the other two as conjugates)
    auto load4 = [&](const Src& src, int s, RV (&r)[4]) {
        if (s == 0) {   // the sweep that holds k = 0: per-lane row select
            const int k = kq;
            r[0] = ld(src, v_up, 0);
            r[2] = ld(src, (N - k) * row_bytes + fs * 8, 0);
            if (ONE) {
                r[1] = ld(src, (k == 0 ? N / 2 : k) * row_bytes + fs * 8, 0);
                r[3] = r[2];   // (defined, unused: fold4 completes it as a conjugate -- an element left undefined on one side of a branch sends the array to scratch)
            } else {
                r[1] = ld(src, (k == 0 ? N / 2 : W - k) * row_bytes + fs * 8, 0);
                r[3] = ld(src, (k == 0 ? N + N / 2 : N + k) * row_bytes + fs * 8, 0);
            }
        } else {
            r[0] = ld(src, v_up, s * KSTEP * row_bytes);
            r[2] = ld(src, v_down, (N - (s + 1) * KSTEP) * row_bytes);
            if (!ONE) {
                r[1] = ld(src, v_down, (W - (s + 1) * KSTEP) * row_bytes);
                r[3] = ld(src, v_up, (N + s * KSTEP) * row_bytes);
            } else {
                r[1] = r[0];
                r[3] = r[2];
            }
        }
    };
    // Hermitian fold of one sweep into the packed half-length spectrum of one frame
    auto fold_one = [&](int k, float2 r0, float2 r1, float2 r2, float2 r3, float2* fb) {
        if (k == 0) {   // r = X[0], X[N/2], X[N], X[3N/2]
            const float a0 = 2.f * r0.x, an = 2.f * r2.x;
            fb[0] = make_float2(a0 - an, a0 + an);
            if (ONE) r3 = cconj(r1);
            const float2 a = make_float2(r1.x + r3.x, r1.y - r3.y);
            fb[phys(N / 2)] = make_float2(-2.f * a.y, 2.f * a.x);
        } else {
            float2 zk, zn;
            unsplit_pair(r0, ONE ? cconj(r0) : r1, r2, ONE ? cconj(r2) : r3, tws_l[k], zk, zn);
            fb[phys(k)] = zk;
            fb[phys(N - k)] = zn;
        }
    };
    auto fold4 = [&](int s, const RV (&r)[4]) {
        const int k = kq + s * KSTEP;
        if constexpr (FV == 2) {
            fold_one(k, make_float2(r[0].x, r[0].y), make_float2(r[1].x, r[1].y), make_float2(r[2].x, r[2].y),
                     make_float2(r[3].x, r[3].y), fbuf);
            fold_one(k, make_float2(r[0].z, r[0].w), make_float2(r[1].z, r[1].w), make_float2(r[2].z, r[2].w),
                     make_float2(r[3].z, r[3].w), fbuf + PITCH);
        } else {
            fold_one(k, r[0], r[1], r[2], r[3], fbuf);
        }
    };
#ifndef ZAFX_ISTFT_PF
#define ZAFX_ISTFT_PF 1
#endif
    constexpr int PF = (!TF && FV == 2 && KI >= ZAFX_ISTFT_PF) ? ZAFX_ISTFT_PF : 0;   // sweeps of the next tile requested ahead
    RV pre[PF > 0 ? PF : 1][4];
    bool pre_ok = false;
    Tile cur;
    cur.unit = blockIdx.x;
    if (cur.unit >= total_units) return;
    enter(cur);
    for (int c = tid; c < ncarry; c += NT) carry[c] = 0.f;
    lds_barrier();   // tables staged
    PROF_INIT(g_prof);

    while (true) {
        PROF_MARK(0);
        const bool carry_only = cur.tile < cur.tile_a;
        const int t_first = cur.tile * FPB;
        Tile nxt = cur;
        if (++nxt.tile >= nxt.tile_b) {
            nxt.unit += gridDim.x;
            if (nxt.unit < total_units) enter(nxt);
        }
        const bool has_next = nxt.unit < total_units;
        if constexpr (TF) {
            // ---- phases A + B, frame-major input: a frame's rows are contiguous, so wave w streams frame t_first + w
            //      (512-B coalesced runs of rows k, W-k, N-k, N+k), folds it into its own LDS buffer and transforms it
            //      without meeting the other waves.
            const int t = t_first + wave;
            if (wave >= (carry_only ? FPB - halo : 0) && t < T) {   // wave-uniform
                const float2* sp = spec + ((long long)(cur.unit / segs) * T + t) * ROWS;
                float2* buf = frames + wave * PITCH;
                int po = p;   // (opaque: the split roots of the sweeps are not carried across tiles)
                asm volatile("" : "+v"(po));
#pragma unroll DEPTH
                for (int s = 0; s < (N / 2) / P; ++s) {
                    const int k = po + s * P;
                    const bool dc = s == 0 && k == 0;   // lane 0 of the first sweep holds rows 0, N/2, N, 3N/2
                    float2 r0 = sp[k], r2 = sp[N - k], r1, r3 = make_float2(0.f, 0.f);
                    if (ONE) {
                        r1 = sp[dc ? N / 2 : k];
                    } else {
                        r1 = sp[dc ? N / 2 : W - k];
                        r3 = sp[dc ? N + N / 2 : N + k];
                    }
                    fold_one(k, r0, r1, r2, r3, buf);
                }
                float2 v[E];
                frame_sync<P>();
                regs_read<LOG2N, LOG2E>(v, buf, po);
                frame_sync<P>();
                fft_frame<LOG2N, LOG2E>(v, buf, po, tw_l);
            }
            PROF_MARK(1);
            PROF_MARK(2);
        } else {
        // ---- phase A: stream the tile's sweeps, DEPTH at a time, through the Hermitian fold into LDS.  The first PF sweeps
        //      were requested when the PREVIOUS tile had been folded and rode in registers across its transforms and its
        //      overlap-add (round 2: the kernel is at 96 VGPRs, the transforms no longer need 116): the gather, half of the tile
        //      time and the only phase with loads in flight, starts half done.
        if (my_frame_needed(cur)) {
            const Src sp = source(cur);
            if constexpr (PF > 0) {
                if (pre_ok) {
                    RV r[4];
                    if constexpr (KI > PF) load4(sp, PF, r);   // the first streamed sweep flies under the folds of the prefetched ones
#pragma unroll
                    for (int s = 0; s < PF; ++s) fold4(s, pre[s]);
                    if constexpr (KI > PF) fold4(PF, r);
#pragma unroll DEPTH
                    for (int s = PF + 1; s < KI; ++s) {
                        load4(sp, s, r);
                        fold4(s, r);
                    }
                } else {
#pragma unroll DEPTH
                    for (int s = 0; s < KI; ++s) {
                        RV r[4];
                        load4(sp, s, r);
                        fold4(s, r);
                    }
                }
            } else {
#pragma unroll DEPTH
            for (int s = 0; s < KI; ++s) {
                RV r[4];
                load4(sp, s, r);
                fold4(s, r);
            }
            }
        }
        if constexpr (PF > 0) {
            pre_ok = has_next && my_frame_needed(nxt);
            if (pre_ok) {
                const Src spn = source(nxt);
#pragma unroll
                for (int s = 0; s < PF; ++s) load4(spn, s, pre[s]);
            }
        }
        PROF_MARK(1);
        lds_barrier();
        PROF_MARK(2);
        // ---- phase B: forward FFT of the swapped spectrum == swapped inverse FFT
        if (wave >= (carry_only ? FPB - halo : 0)) {   // wave-uniform
            float2* buf = frames + wave * PITCH;
            float2 v[E];
            regs_read<LOG2N, LOG2E>(v, buf, p);
            frame_sync<P>();
            fft_frame<LOG2N, LOG2E>(v, buf, p, tw_l);
        }
        }
        PROF_MARK(3);
        lds_barrier();
        PROF_MARK(4);
        // ---- phase C: overlap-add (carry first), trim (:236-238), COLA gain (:241); next carry
        {
            OlaArgs a;
            a.fl = reinterpret_cast<const float*>(frames);
            a.carry = carry;
            a.pitch2 = 2 * PITCH;
            a.ncarry = ncarry;
            a.hop = hop;
            a.n_valid = min(FPB, T - t_first);
            a.c_end = cur.tile == tiles - 1 ? a.n_valid * hop + ncarry : FPB * hop;
            a.write_out = !carry_only;
            a.make_carry = cur.tile + 1 < cur.tile_b;
            const long long clip = cur.unit / segs;
            a.yc = y + clip * out_len;
            a.o_first = (long long)t_first * hop - ncarry;
            a.out_len = out_len;
            a.scale = scale;
            const bool ya = y_base_aligned && (clip * out_len) % 2 == 0;
            if (pairs) {
                if (!ola_phase_sweep<W, NT, FPB>(a, tid, ya)) ola_phase_pairs<W, NT, FPB>(a, tid, ya);
            } else {
                ola_phase<W, NT, FPB>(a, tid);
            }
        }
        PROF_MARK(5);
        lds_barrier();
        PROF_MARK(6);
        if (!has_next) break;
        cur = nxt;
    }
}

// ---------------------------------------------------------------------------------
// inverse, reference layout, W = 4096: the carry kernel over two BANDS of samples, one band per workgroup (k_istft_ft16b)
// ---------------------------------------------------------------------------------
// Sixteen packed frames of M = 2048 points do not fit LDS, so the inverse ran on the one-workgroup-per-tile kernel with 8-frame tiles
// (64-byte gather runs, a halo frame re-read per tile: 2.3 TB/s).  Decimation in TIME of the inverse transform:
//     z[2m + b] = (1/M) sum_{k<1024} (Z[k] + (-1)^b Z[k + 1024]) e^{2 pi i k b / M} e^{2 pi i k m / 1024},   b = 0, 1
// -- the even and the odd packed samples of a frame are each ONE 1024-point inverse transform of a combination of the two halves
// of Z.  The reference's overlap-add has no synthesis window (zaf.py:226-241: a plain sum, then the COLA gain), and a hop that
// is a multiple of 4 keeps a sample's residue mod 4: band b of every frame adds up to exactly the output samples 4m + 2b,
// 4m + 2b + 1.  So band b is an ISTFT of its own -- frames of 2048 samples, hop H / 2, writing every other sample pair -- and
// k_istft_ft16's machinery (16-frame tiles, carry in LDS, the three overlap-add forms) runs it unchanged behind a different
// front end and an output index map (OS = 2).  A workgroup owns one band of a clip segment; the two bands' workgroups are
// neighbours in the XCD order and gather the same rows at about the same time (k_stft_ft16bc measured 1.1 x, not 2 x, of
// fetch for this arrangement).
// Front end, in the swapped representation S = (Im Z, Re Z) that the forward transform inverts: the fold of rows k, W - k, M - k,
// M + k gives S[k], S[M - k], that of rows 1024 - k, 3072 + k, 1024 + k, 3072 - k gives S[1024 - k], S[1024 + k]; then
//     U_b[k] = (S[k] +- S[1024 + k]) c_b(k),  U_b[1024 - k] = (S[1024 - k] +- S[M - k]) c_b(1024 - k),
// c_1(k) = e^{-2 pi i k / 2048} = root[2k], c_1(1024 - k) = -conj root[2k], c_0 = 1; the lane with k = 0 forms U_b[0] from the
// special rows 0, M/2, M, 3M/2 and U_b[512] from the pair of k = 512.
template <int DEPTH, bool ONE, int FV>
__global__ __launch_bounds__(1024) void k_istft_ft16b(
    const float2* __restrict__ spec, const float2* __restrict__ twp, const float2* __restrict__ tws,
    float* __restrict__ y, int T, int TP, int hop /* H / 2: the band's hop */, long long out_len, float scale, int tiles, int segs, int seg_tiles,
    int total_units, int halo) {
    using C = FftCfg<10, 4>;
    using F = FatCfg<10, 4>;
    constexpr int N = C::N, P = C::P, E = C::E, W = 2 * N /* samples of a band's frame */, M = 2048, WR = 4096, NT = 1024, FPB = kFatFrames, PITCH = F::PITCH;
    constexpr int ROWS = ONE ? M + 1 : WR;
    // FV frames per lane and load: 2 = 16-byte loads of two adjacent frames (even row pitch, 16-byte aligned array), else 1
    constexpr int LPR = FPB / FV, KSTEP = NT / LPR, KI = (N / 2) / KSTEP;
    static_assert(FV == 1 || FV == 2, "one or two frames per lane");
    using RV = std::conditional_t<FV == 2, float4, float2>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* frames = reinterpret_cast<float2*>(smem_raw);
    float2* tw_l = frames + FPB * PITCH;
    float2* tws_l = tw_l + C::TW;                                 // M/2 + 1 roots of 4096
    float* carry = reinterpret_cast<float*>(tws_l + M / 2 + 1);   // W - hop floats of the band
    const int tid = threadIdx.x;
    for (int i = tid; i < C::TW; i += NT) tw_l[i] = twp[i];
    for (int i = tid; i <= M / 2; i += NT) tws_l[i] = tws[i];
    const int wave = tid / P, p = tid % P;
    const int fs = (tid % LPR) * FV, kq = tid / LPR;
    float2* fbuf = frames + fs * PITCH;
    const int ncarry = W - hop;
    const bool pairs = hop % 2 == 0 && 2 * hop >= W;
    const bool y_base_aligned = reinterpret_cast<uintptr_t>(y) % 8 == 0;
    const bool xcd = ZAFX_XCD_ORDER && gridDim.x % 8 == 0;

    struct Tile {
        int v, band, clip, tile, tile_a, tile_b;
    };
    auto enter = [&](Tile& it) {   // first tile of walk position it.v (one before the segment when it needs a carry)
        const int u = xcd ? xcd_order(it.v, total_units) : it.v;
        it.band = u & 1;
        const int sg = u >> 1, seg = sg % segs;
        it.clip = sg / segs;
        it.tile_a = seg * seg_tiles;
        it.tile_b = min(it.tile_a + seg_tiles, tiles);
        it.tile = it.tile_a > 0 ? it.tile_a - 1 : 0;
    };
    auto my_frame_needed = [&](const Tile& it) { return it.tile * FPB + fs < T && fs + FV - 1 >= (it.tile < it.tile_a ? FPB - halo : 0); };
    const int row_bytes = TP * 8;
    const int v_up = kq * row_bytes + fs * 8, v_down = (KSTEP - kq) * row_bytes + fs * 8;
    struct Src {
        __amdgpu_buffer_rsrc_t rsrc;
        int t_bytes;
    };
    auto source = [&](const Tile& it) {
        Src src;
        src.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(spec) + (long long)it.clip * ROWS * TP, 0, ROWS * row_bytes, 0x00020000);
        src.t_bytes = it.tile * FPB * 8;
        return src;
    };
    auto ld = [&](const Src& src, int voff, int soff) {
        RV f;
        if constexpr (FV == 2) {
            const auto raw = __builtin_amdgcn_raw_buffer_load_b128(src.rsrc, voff, soff + src.t_bytes, 0);
            __builtin_memcpy(&f, &raw, 16);
        } else {
            const auto raw = __builtin_amdgcn_raw_buffer_load_b64(src.rsrc, voff, soff + src.t_bytes, 0);
            __builtin_memcpy(&f, &raw, 8);
        }
        return f;
    };
    // rows of sweep s for my frame(s): r[0..3] = X[k], X[WR-k], X[M-k], X[M+k]; r[4..7] = X[1024-k], X[3072+k], X[1024+k], X[3072-k]
    // (k = 0: X[0], X[M/2], X[M], X[3M/2] and the rows of k = 512: X[512], X[3584], X[1536], X[2560]).  One-sided input: the even
    // slots only (+ slot 1 = row M/2 for k = 0); the fold completes the others as conjugates.
    auto row_at = [&](const Src& src, int row) { return ld(src, row * row_bytes + fs * 8, 0); };
    // (one-sided: rows k, M - k, 1024 - k, 1024 + k and -- for the lane with k = 0 -- M / 2 only: an array of their own, RN = 5, every
    // element always written: with half-defined elements the array went to scratch)
    constexpr int RN = ONE ? 5 : 8;
    auto load8 = [&](const Src& src, int s, RV (&q)[RN]) {
        if constexpr (ONE) {
            if (s == 0) {
                const int k = kq;
                const bool z = k == 0;
                q[0] = row_at(src, k);
                q[1] = row_at(src, M - k);
                q[2] = row_at(src, z ? 512 : 1024 - k);
                q[3] = row_at(src, z ? 1536 : 1024 + k);
                q[4] = row_at(src, z ? M / 2 : k);
            } else {
                q[0] = ld(src, v_up, s * KSTEP * row_bytes);
                q[1] = ld(src, v_down, (M - (s + 1) * KSTEP) * row_bytes);
                q[2] = ld(src, v_down, (1024 - (s + 1) * KSTEP) * row_bytes);
                q[3] = ld(src, v_up, (1024 + s * KSTEP) * row_bytes);
                q[4] = q[0];
            }
        } else {
            if (s == 0) {   // the sweep that holds k = 0: per-lane row select
                const int k = kq;
                const bool z = k == 0;
                q[0] = row_at(src, k);
                q[1] = row_at(src, z ? M / 2 : WR - k);
                q[2] = row_at(src, M - k);
                q[3] = row_at(src, z ? M + M / 2 : M + k);
                q[4] = row_at(src, z ? 512 : 1024 - k);
                q[5] = row_at(src, z ? 3584 : 3072 + k);
                q[6] = row_at(src, z ? 1536 : 1024 + k);
                q[7] = row_at(src, z ? 2560 : 3072 - k);
            } else {
                q[0] = ld(src, v_up, s * KSTEP * row_bytes);
                q[1] = ld(src, v_down, (WR - (s + 1) * KSTEP) * row_bytes);
                q[2] = ld(src, v_down, (M - (s + 1) * KSTEP) * row_bytes);
                q[3] = ld(src, v_up, (M + s * KSTEP) * row_bytes);
                q[4] = ld(src, v_down, (1024 - (s + 1) * KSTEP) * row_bytes);
                q[5] = ld(src, v_up, (3072 + s * KSTEP) * row_bytes);
                q[6] = ld(src, v_up, (1024 + s * KSTEP) * row_bytes);
                q[7] = ld(src, v_down, (3072 - (s + 1) * KSTEP) * row_bytes);
            }
        }
    };
    // one frame: the eight rows of bin k -> U_b[k], U_b[1024 - k] (k = 0: U_b[0], U_b[512]) in the frame buffer fb
    auto fold_one = [&](int k, int band, float2 r0, float2 r1, float2 r2, float2 r3, float2 r4, float2 r5, float2 r6, float2 r7, float2* fb) {
        if (ONE) {
            r3 = cconj(r2);
            r5 = cconj(r4);
            r7 = cconj(r6);
        }
        if (k == 0) {
            const float a0 = 2.f * r0.x, an = 2.f * r2.x;
            const float2 s0 = make_float2(a0 - an, a0 + an);                                  // S[0]
            if (ONE) r3 = cconj(r1);
            const float2 a = make_float2(r1.x + r3.x, r1.y - r3.y);
            const float2 sh = make_float2(-2.f * a.y, 2.f * a.x);                             // S[M/2] = S[1024]
            fb[0] = band ? csub(s0, sh) : cadd(s0, sh);
            float2 s5, s15;                                                                     // S[512], S[1536]
            unsplit_pair(r4, r5, r6, r7, tws_l[512], s5, s15);
            fb[phys(512)] = band ? mul_mi(csub(s5, s15)) : cadd(s5, s15);                      // c_1(512) = -i
        } else {
            if (ONE) r1 = cconj(r0);
            float2 sk, smk, s1mk, s1pk;   // S[k], S[M-k], S[1024-k], S[1024+k]
            unsplit_pair(r0, r1, r2, r3, tws_l[k], sk, smk);
            unsplit_pair(r4, r5, r6, r7, tws_l[1024 - k], s1mk, s1pk);
            if (band) {
                const float2 c = tws_l[2 * k];
                fb[phys(k)] = cmul(csub(sk, s1pk), c);
                const float2 d = csub(smk, s1mk);          // -(S[1024-k] - S[M-k])
                fb[phys(1024 - k)] = cmulc(d, c);            // (S[1024-k] - S[M-k]) * (-conj c)
            } else {
                fb[phys(k)] = cadd(sk, s1pk);
                fb[phys(1024 - k)] = cadd(s1mk, smk);
            }
        }
    };
    auto fold8 = [&](int s, int band, const RV (&q)[RN]) {
        const int k = kq + s * KSTEP;
        RV r[8];   // slots as in the comment above load8 (one-sided: 3, 5, 7 are replaced by conjugates in fold_one)
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = q[ONE ? ((i & 1) ? (i == 1 ? 4 : 0) : i / 2) : i];
        if constexpr (FV == 2) {
            auto lo = [](float4 v) { return make_float2(v.x, v.y); };
            auto hi = [](float4 v) { return make_float2(v.z, v.w); };
            fold_one(k, band, lo(r[0]), lo(r[1]), lo(r[2]), lo(r[3]), lo(r[4]), lo(r[5]), lo(r[6]), lo(r[7]), fbuf);
            fold_one(k, band, hi(r[0]), hi(r[1]), hi(r[2]), hi(r[3]), hi(r[4]), hi(r[5]), hi(r[6]), hi(r[7]), fbuf + PITCH);
        } else {
            fold_one(k, band, r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], fbuf);
        }
    };
    Tile cur;
    cur.v = blockIdx.x;
    if (cur.v >= total_units) return;
    enter(cur);
    for (int c = tid; c < ncarry; c += NT) carry[c] = 0.f;
    lds_barrier();   // tables staged

    while (true) {
        const bool carry_only = cur.tile < cur.tile_a;
        const int t_first = cur.tile * FPB;
        Tile nxt = cur;
        if (++nxt.tile >= nxt.tile_b) {
            nxt.v += gridDim.x;
            if (nxt.v < total_units) enter(nxt);
        }
        const bool has_next = nxt.v < total_units;
        // ---- phase A: stream the tile's sweeps, DEPTH at a time, through the fold into LDS
        if (my_frame_needed(cur)) {
            const Src sp = source(cur);
#pragma unroll DEPTH
            for (int s = 0; s < KI; ++s) {
                RV r[RN];
                load8(sp, s, r);
                fold8(s, cur.band, r);
            }
        }
        lds_barrier();
        // ---- phase B: forward FFT of the swapped spectrum == swapped inverse FFT
        if (wave >= (carry_only ? FPB - halo : 0)) {   // wave-uniform
            float2* buf = frames + wave * PITCH;
            float2 v[E];
            regs_read<10, 4>(v, buf, p);
            frame_sync<P>();
            fft_frame<10, 4>(v, buf, p, tw_l);
        }
        lds_barrier();
        // ---- phase C: overlap-add of the band (carry first), trim, COLA gain; next carry
        {
            OlaArgs a;
            a.boff = 2 * cur.band;
            a.fl = reinterpret_cast<const float*>(frames);
            a.carry = carry;
            a.pitch2 = 2 * PITCH;
            a.ncarry = ncarry;
            a.hop = hop;
            a.n_valid = min(FPB, T - t_first);
            a.c_end = cur.tile == tiles - 1 ? a.n_valid * hop + ncarry : FPB * hop;
            a.write_out = !carry_only;
            a.make_carry = cur.tile + 1 < cur.tile_b;
            a.yc = y + (long long)cur.clip * out_len;
            a.o_first = (long long)t_first * hop - ncarry;
            a.out_len = out_len;
            a.scale = scale;
            const bool ya = y_base_aligned && ((long long)cur.clip * out_len) % 2 == 0;
            if (pairs) {
                if (!ola_phase_sweep<W, NT, FPB, 2>(a, tid, ya)) ola_phase_pairs<W, NT, FPB, 2>(a, tid, ya);
            } else {
                ola_phase<W, NT, FPB, 2>(a, tid);
            }
        }
        lds_barrier();
        if (!has_next) break;
        cur = nxt;
    }
}

// ---------------------------------------------------------------------------------
// launch plumbing
// ---------------------------------------------------------------------------------
constexpr int stft_fpb(int log2n, int layout) {
    const int p = (1 << log2n) >> default_log2e(log2n);
    int cap = 1024 / p;
    // LDS cap: FPB * PITCH * 8 <= 160 KiB
    const int pitch = (1 << log2n) + ((1 << log2n) >> 4) + 1;
    int lds_cap = kMaxLdsBytes / (pitch * 8);
    int f = layout == ZAFX_LAYOUT_FT ? 16 : 4;
    if (f > cap) f = cap;
    if (f > lds_cap) f = lds_cap;
    // round down to a power of two
    int r = 1;
    while (r * 2 <= f) r *= 2;
    return r;
}

constexpr bool stft_use_fat(int log2n, int layout) {
    return layout == ZAFX_LAYOUT_FT && log2n >= 7 && log2n <= 10;   // one wavefront per frame
}

template <int LOG2N, bool ALIGNED, int SPEC>
static hipError_t run_stft_fat(const zafx_plan& pl, const float* x, float2* out, int64_t n_clips, int64_t n_samples, int T) {
#ifndef ZAFX_STFT_R32
#define ZAFX_STFT_R32 1
#endif
    // 1024 points as two radix-32 passes (32 points per thread, a frame per half wavefront) instead of 16 x 16 x 4:
    // 495 instead of 604 instructions per frame, one LDS exchange instead of two; FFT phase 9.6 k -> 7.4 k cycles per
    // tile.  The two-sided kernel is bound by its store drain and does not move (1.95 ms either way); the one-sided /
    // magnitude / power outputs gain 4-5 %.  (The fused mel kernel is better off with 16 waves x radix 16: 1.48 vs 1.67 ms.)
    // W = 4096 (LOG2N = 11): 2048 points as 32 x 32 x 2, a frame per wavefront, 8-frame tiles (16 do not fit LDS)
    constexpr int LOG2E = ((ZAFX_STFT_R32 && LOG2N == 10) || LOG2N == 11) ? 5 : default_log2e(LOG2N);
    constexpr int FPB = LOG2N == 11 ? 8 : kFatFrames;
    using F = FatCfg<LOG2N, LOG2E, FPB>;
    static_assert(F::SMEM <= (size_t)kMaxLdsBytes, "tile + tables exceed LDS");
    auto kern = k_stft_ft16<LOG2N, LOG2E, ALIGNED, SPEC, FPB>;
    if constexpr (LOG2N == 10 && SPEC < 2 && FPB == kFatFrames) {   // (zafx_execute_pcm: int16 in the loads; stft_pcm_direct_ok vouches for the geometry)
        const int pcm = take_pcm_mode();
        if (pcm == 1) kern = k_stft_ft16<LOG2N, LOG2E, ALIGNED, SPEC, FPB, 1>;
        if (pcm == 2) kern = k_stft_ft16<LOG2N, LOG2E, ALIGNED, SPEC, FPB, 2>;
    }
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, F::SMEM); e != hipSuccess) return e;
    const int tiles = (T + FPB - 1) / FPB;
    const long long total = (long long)tiles * n_clips;
    if (total <= 0) return hipSuccess;
    const int per_cu = (int)std::min<size_t>(2, (size_t)kMaxLdsBytes / F::SMEM);
    const long long grid = std::min<long long>(total, (long long)pl.n_cus * std::max(per_cu, 1));
    pl.ran = "k_stft_ft16";
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(F::NT), F::SMEM, pl.stream, x, pl.d_window, LOG2E == 5 ? pl.d_tw_r32 : pl.d_tw_pass,
                       pl.d_tw_aux, out, (long long)n_samples, pl.H, T, (int)row_pitch(pl, T), tiles, (int)total);
    return hipGetLastError();
}

// k_stft_ft16c: complex spectra whose rows are not whole 128-byte lines (see the kernel)
template <int LOG2N, bool ALIGNED, int SPEC>
static hipError_t run_stft_fat_carry(const zafx_plan& pl, const float* x, float2* out, int64_t n_clips, int64_t n_samples, int T) {
    if constexpr (SPEC < 2) {
        constexpr int LOG2E = default_log2e(LOG2N);
        using F = FatCfg<LOG2N, LOG2E>;
        auto kern = k_stft_ft16c<LOG2N, LOG2E, ALIGNED, SPEC>;
        if constexpr (LOG2N == 10 && ALIGNED) {
            const int pcm = take_pcm_mode();
            if (pcm == 1) kern = k_stft_ft16c<LOG2N, LOG2E, true, SPEC, 1>;
            if (pcm == 2) kern = k_stft_ft16c<LOG2N, LOG2E, true, SPEC, 2>;
        }
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, F::SMEM); e != hipSuccess) return e;
        const int tiles = (T + kFatFrames - 1) / kFatFrames;
        if ((long long)tiles * n_clips <= 0) return hipSuccess;
        const long long max_grid = pl.n_cus;
        const int segs = carry_segments(n_clips, tiles, max_grid);
        const int seg_tiles = (tiles + segs - 1) / segs;
        const long long units = (long long)n_clips * segs;
        const long long grid = std::min<long long>(units, max_grid);
        pl.ran = "k_stft_ft16c";
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(F::NT), F::SMEM, pl.stream, x, pl.d_window, pl.d_tw_pass, pl.d_tw_aux, out,
                           (long long)n_samples, pl.H, T, (int)row_pitch(pl, T), tiles, segs, seg_tiles, (int)units);
        return hipGetLastError();
    } else {
        return hipErrorInvalidValue;
    }
}

// k_stft_ft16b: W = 4096 in the reference layout, two bands of bins per tile (see the kernel)
template <bool ALIGNED, int SPEC>
static hipError_t run_stft_band(const zafx_plan& pl, const float* x, float2* out, int64_t n_clips, int64_t n_samples, int T) {
    using B = BandCfg;
    auto kern = k_stft_ft16b<ALIGNED, SPEC>;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, B::SMEM); e != hipSuccess) return e;
    const int tiles = (T + B::FPB - 1) / B::FPB;
    const long long total = (long long)tiles * n_clips;
    if (total <= 0) return hipSuccess;
    const long long grid = std::min<long long>(total, (long long)pl.n_cus);
    pl.ran = "k_stft_ft16b";
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(B::NT), B::SMEM, pl.stream, x, pl.d_window, pl.d_tw_sub, pl.d_tw_aux, out,
                       (long long)n_samples, pl.H, T, (int)row_pitch(pl, T), tiles, (int)total);
    return hipGetLastError();
}

// k_stft_ft16q: W = 8192 in the reference layout, four classes of bins per tile (see the kernel)
template <int SPEC>
static hipError_t run_stft_quad(const zafx_plan& pl, const float* x, float2* out, int64_t n_clips, int64_t n_samples, int T) {
    using B = QuadCfg;
    auto kern = k_stft_ft16q<SPEC>;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, B::SMEM); e != hipSuccess) return e;
    const int tiles = (T + B::FPB - 1) / B::FPB;
    const long long total = (long long)tiles * n_clips;
    if (total <= 0) return hipSuccess;
    const long long grid = std::min<long long>(total, (long long)pl.n_cus);
    pl.ran = "k_stft_ft16q";
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(B::NT), B::SMEM, pl.stream, x, pl.d_window, pl.d_tw_sub, pl.d_tw_quad, out,
                       (long long)n_samples, pl.H, T, (int)row_pitch(pl, T), tiles, (int)total);
    return hipGetLastError();
}

// k_stft_ft16bc: W = 4096, complex rows off the 128-byte grid (one band per workgroup, register carry)
template <bool ALIGNED, int SPEC>
static hipError_t run_stft_band_carry(const zafx_plan& pl, const float* x, float2* out, int64_t n_clips, int64_t n_samples, int T) {
    if constexpr (SPEC < 2) {
        using B = BandCfg;
        auto kern = k_stft_ft16bc<ALIGNED, SPEC>;
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, B::SMEM); e != hipSuccess) return e;
        const int tiles = (T + B::FPB - 1) / B::FPB;
        if ((long long)tiles * n_clips <= 0) return hipSuccess;
        const long long max_grid = pl.n_cus;
        const int segs = carry_segments(2 * n_clips, tiles, max_grid);   // (two units -- one per band -- for every segment)
        const int seg_tiles = (tiles + segs - 1) / segs;
        const long long units = 2LL * n_clips * segs;
        const long long grid = std::min<long long>(units, max_grid);
        pl.ran = "k_stft_ft16bc";
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(B::NT), B::SMEM, pl.stream, x, pl.d_window, pl.d_tw_sub, pl.d_tw_aux, out,
                           (long long)n_samples, pl.H, T, (int)row_pitch(pl, T), tiles, segs, seg_tiles, (int)units);
        return hipGetLastError();
    } else {
        return hipErrorInvalidValue;
    }
}

// k_mel_ft16b: mel / mfcc at W = 4096, fused on the two-band kernel.  false: the caller takes the spectrum-kernel + k_melfb route (clips too
// long for 32-bit byte offsets, unaligned base)
bool mel_band_usable(const zafx_plan& pl, const float* x, int64_t n_clips, int64_t n_samples, int T) {
    return pl.log2nf == 11 && pl.d_tw_sub && pl.d_fbw && pl.prm.n_filters <= 256 && n_samples < (1LL << 29) && (long long)(T + 16) * pl.H < (1LL << 29) &&
           (long long)n_clips * ((T + 15) / 16) < (1LL << 31) && reinterpret_cast<uintptr_t>(x) % 4 == 0;
}
hipError_t launch_mel_band(const zafx_plan& pl, const float* x, float* out, int64_t n_clips, int64_t n_samples, int T) {
    using B = BandCfg;
    const bool mfcc = pl.kind == ZAFX_MFCC;
    const bool aligned = (n_samples % 2 == 0) && (pl.H % 2 == 0) && (reinterpret_cast<uintptr_t>(x) % 8 == 0);
    auto kern = mfcc ? (aligned ? k_mel_ft16b<true, true> : k_mel_ft16b<false, true>) : (aligned ? k_mel_ft16b<true, false> : k_mel_ft16b<false, false>);
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, B::SMEM); e != hipSuccess) return e;
    const int tiles = (T + B::FPB - 1) / B::FPB;
    const long long total = (long long)tiles * n_clips;
    if (total <= 0) return hipSuccess;
    const long long grid = std::min<long long>(total, (long long)pl.n_cus);
    pl.ran = "k_mel_ft16b";
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(B::NT), B::SMEM, pl.stream, x, pl.d_window, pl.d_tw_sub, pl.d_tw_aux, pl.d_fbw, pl.d_fbw_meta, pl.d_dctw, out,
                       (long long)n_samples, pl.H, T, (int)row_pitch(pl, T), tiles, (int)total, pl.prm.n_filters, mfcc ? pl.prm.n_coefs : 0, pl.layout);
    return hipGetLastError();
}

constexpr bool stft_use_tf(int log2n, int layout) {
    return layout == ZAFX_LAYOUT_TF && log2n >= 7 && log2n <= 10;   // one wavefront (or half of one) per frame
}

template <int LOG2N, bool ALIGNED, int SPEC>
static hipError_t run_stft_tf(const zafx_plan& pl, const float* x, float2* out, int64_t n_clips, int64_t n_samples, int T) {
    constexpr int LOG2E = tf_log2e(LOG2N);
    using C = FftCfg<LOG2N, LOG2E>;
    constexpr int NT = tf_threads(LOG2N), SLOTS = NT / C::P;
    constexpr size_t SMEM = (size_t)(SLOTS * C::PITCH + C::TW + C::N + C::N / 2 + 1) * 8;
    static_assert(SMEM <= (size_t)kMaxLdsBytes, "frame-major STFT tables + buffers exceed LDS");
    auto kern = k_stft_tf<LOG2N, LOG2E, ALIGNED, SPEC>;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, SMEM); e != hipSuccess) return e;
    const long long total = (long long)T * n_clips;
    if (total <= 0) return hipSuccess;
    const int per_cu = (int)std::min<size_t>(2, (size_t)kMaxLdsBytes / SMEM);
    const long long grid = std::min<long long>((total + SLOTS - 1) / SLOTS, (long long)pl.n_cus * std::max(per_cu, 1));
    pl.ran = "k_stft_tf";
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), SMEM, pl.stream, x, pl.d_window, LOG2E == 5 ? pl.d_tw_r32 : pl.d_tw_pass,
                       pl.d_tw_aux, out, (long long)n_samples, pl.H, T, total);
    return hipGetLastError();
}

template <int LOG2N, int LAYOUT, int SPEC>
static hipError_t run_stft(const zafx_plan& pl, const float* x, float2* out, int64_t n_clips, int64_t n_samples, int T) {
    const bool aligned = (n_samples % 2 == 0) && (pl.H % 2 == 0) && (reinterpret_cast<uintptr_t>(x) % 8 == 0);
#ifndef ZAFX_STFT_FAT8
#define ZAFX_STFT_FAT8 0   // (measured: 6.54 ms against 3.13 ms for the one-workgroup-per-tile kernel at W = 4096 -- eight waves, one 2048-point
#endif                     //  frame each as 32 x 32 x 2 with the window from global memory, are latency bound; profiles/r03_notes.md)
    if constexpr (ZAFX_STFT_FAT8 && LOG2N == 11 && LAYOUT == ZAFX_LAYOUT_FT) {
        // W = 4096, reference layout: the persistent kernel with 8-frame tiles (experiment, off)
        if (pl.d_tw_r32)
            return aligned ? run_stft_fat<LOG2N, true, SPEC>(pl, x, out, n_clips, n_samples, T)
                           : run_stft_fat<LOG2N, false, SPEC>(pl, x, out, n_clips, n_samples, T);
    }
#ifndef ZAFX_STFT_BAND
#define ZAFX_STFT_BAND 1
#endif
    if constexpr (ZAFX_STFT_BAND && LOG2N == 11 && LAYOUT == ZAFX_LAYOUT_FT) {
        // W = 4096, reference layout: 16-frame tiles in two bands of bins (buffer loads: 32-bit byte offsets inside a clip)
        // Complex rows off the 128-byte grid: k_stft_ft16b writes every line in two parts far apart (1024 clips, two-sided, T = 217: 4.33 ms
        // against 3.24 on the one-workgroup-per-tile kernel); the carry form k_stft_ft16bc (one band per workgroup) completes the lines:
        // T = 217: 2.33 ms, T = 434 (hop 1024): 4.44 against 5.20.  One-sided output has half the stores to gain from and the form reads
        // every sample twice: it wins at hop >= W / 2 only (T = 217: 1.89 against 1.96; hop 1024: 3.72 against 3.36 -> generic kernel).
        // The float32 kinds write half lines either way and gain from k_stft_ft16b at every T (magnitude, T = 217: 1.33 against 1.67 ms).
        const bool whole = row_pitch(pl, T) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 128 == 0;
#ifndef ZAFX_STFT_BAND_CARRY
#define ZAFX_STFT_BAND_CARRY 1
#endif
        const bool fits = pl.d_tw_sub && (long long)n_clips * ((T + 15) / 16) < (1LL << 30) && n_samples < (1LL << 29) && (long long)(T + 16) * pl.H < (1LL << 29) && reinterpret_cast<uintptr_t>(x) % 4 == 0;
        if constexpr (SPEC < 2 && ZAFX_STFT_BAND_CARRY) {
            if (!whole && fits && (SPEC == 0 || 2 * pl.H >= pl.W) && reinterpret_cast<uintptr_t>(out) % 8 == 0 && n_clips * (long long)T < (1LL << 31))
                return aligned ? run_stft_band_carry<true, SPEC>(pl, x, out, n_clips, n_samples, T) : run_stft_band_carry<false, SPEC>(pl, x, out, n_clips, n_samples, T);
        }
        if ((SPEC >= 2 || whole) && pl.d_tw_sub && (long long)n_clips * ((T + 15) / 16) < (1LL << 31) && n_samples < (1LL << 29) && (long long)(T + 16) * pl.H < (1LL << 29) && reinterpret_cast<uintptr_t>(x) % 4 == 0)
            return aligned ? run_stft_band<true, SPEC>(pl, x, out, n_clips, n_samples, T) : run_stft_band<false, SPEC>(pl, x, out, n_clips, n_samples, T);
    }
#ifndef ZAFX_STFT_QUAD
#define ZAFX_STFT_QUAD 1
#endif
    if constexpr (ZAFX_STFT_QUAD && LOG2N == 12 && LAYOUT == ZAFX_LAYOUT_FT) {
        // W = 8192, reference layout: 16-frame tiles in four classes of bins (4-byte buffer loads: 32-bit byte offsets inside a clip)
        // 1024 clips x 10 s, hop 4096, T = 112 (rows are whole lines): two-sided 2.19-2.29 ms against 2.77 on the one-workgroup-per-tile kernel, one-sided
        // 1.68 against 1.99, |X| 1.60 against 1.84; T = 109: 3.3-4.3 against 3.04 two-sided (every line written in two parts by two workgroups: the
        // generic kernel keeps those), 2.19 against 2.17 one-sided, 1.74 against 1.93 |X|
        const bool whole = row_pitch(pl, T) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 128 == 0;
        if ((SPEC != 0 || whole) && pl.d_tw_sub && pl.d_tw_quad && (long long)n_clips * ((T + 15) / 16) < (1LL << 31) && n_samples < (1LL << 29) &&
            (long long)(T + 16) * pl.H < (1LL << 28) && reinterpret_cast<uintptr_t>(x) % 4 == 0)
            return run_stft_quad<SPEC>(pl, x, out, n_clips, n_samples, T);
    }
    if constexpr (stft_use_fat(LOG2N, LAYOUT)) {
#ifndef ZAFX_STFT_CARRY
#define ZAFX_STFT_CARRY 1
#endif
#ifndef ZAFX_STFT_CARRY_ALWAYS
#define ZAFX_STFT_CARRY_ALWAYS(log2n) ((log2n) == 9)   // W = 1024, two-sided: the carry form also on the line grid (863 / 864 frames: 1.90 ms against k_stft_ft16's 2.22; one-sided 1.31 against 1.21: not; W = 512, 256: 2.15 / 2.54 against 2.05 / 2.12: not)
#endif
        if constexpr (SPEC < 2 && ZAFX_STFT_CARRY) {
            // rows off the 128-byte grid (the array itself only needs its 8-byte alignment): the carry form writes whole lines anyway
            if (((SPEC == 0 && ZAFX_STFT_CARRY_ALWAYS(LOG2N)) || row_pitch(pl, T) % 16 != 0 || reinterpret_cast<uintptr_t>(out) % 128 != 0) && reinterpret_cast<uintptr_t>(out) % 8 == 0 && n_clips * (long long)T < (1LL << 31))
                return aligned && n_samples < (1LL << 29) && (long long)T * pl.H < (1LL << 29) ? run_stft_fat_carry<LOG2N, true, SPEC>(pl, x, out, n_clips, n_samples, T)
                               : run_stft_fat_carry<LOG2N, false, SPEC>(pl, x, out, n_clips, n_samples, T);
        }
        return aligned ? run_stft_fat<LOG2N, true, SPEC>(pl, x, out, n_clips, n_samples, T)
                       : run_stft_fat<LOG2N, false, SPEC>(pl, x, out, n_clips, n_samples, T);
    } else if constexpr (stft_use_tf(LOG2N, LAYOUT)) {
        return aligned ? run_stft_tf<LOG2N, true, SPEC>(pl, x, out, n_clips, n_samples, T)
                       : run_stft_tf<LOG2N, false, SPEC>(pl, x, out, n_clips, n_samples, T);
    } else {
        constexpr int LOG2E = default_log2e(LOG2N);
        constexpr int FPB = stft_fpb(LOG2N, LAYOUT);
        using S = StftCfg<LOG2N, LOG2E, FPB>;
        auto kern = k_stft<LOG2N, LOG2E, FPB, LAYOUT, SPEC>;
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, S::SMEM); e != hipSuccess) return e;
        const int tiles = (T + FPB - 1) / FPB;
        const long long blocks = (long long)tiles * n_clips;
        if (blocks <= 0) return hipSuccess;
        pl.ran = "k_stft";
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(S::NT), S::SMEM, pl.stream, x, pl.d_window, pl.d_tw_pass, pl.d_tw_aux,
                           out, (long long)n_samples, pl.H, T, (int)row_pitch(pl, T), tiles);
        return hipGetLastError();
    }
}

// Cut every clip's tiles into `segs` segments so that the persistent grid is evenly loaded: whole
// clips when there are enough of them, otherwise shorter segments (each pays one carry-only tile).
int carry_segments(long long n_clips, int tiles, long long grid) {
    int best = 1;
    double best_cost = 1e300;
    for (int segs = 1; segs <= tiles; ++segs) {
        const int seg_tiles = (tiles + segs - 1) / segs;
        if (segs > 1 && (segs - 1) * seg_tiles >= tiles) continue;   // would leave an empty segment
        const long long rounds = (n_clips * segs + grid - 1) / grid;
        const double cost = (double)rounds * (seg_tiles + (segs > 1 ? 0.5 : 0.0));
        if (cost < best_cost - 1e-9) best_cost = cost, best = segs;
    }
    return best;
}

#ifndef ZAFX_ISTFT_TF_DEPTH
#define ZAFX_ISTFT_TF_DEPTH 2
#endif
template <int LOG2N, bool ONE, bool TF = false>
static hipError_t run_istft_fat(const zafx_plan& pl, const float2* spec, float* y, int64_t n_clips, int T, int64_t out_len) {
    constexpr int LOG2E = default_log2e(LOG2N);
    using F = FatCfg<LOG2N, LOG2E>;
    // 16-byte gathers (two adjacent frames per lane) when the rows allow it; see the kernel
    constexpr bool can_vec = LOG2N >= 8 && !TF;
    const int TP = (int)row_pitch(pl, T);
    // (buffer loads take 16 bytes at any 8-byte offset: odd row pitches -- T = 433: 2.46 ms on the 8-byte form against 1.80 at T = 432 -- ride the
    // 16-byte form too; a pair whose second frame lies past T reads the next row's first element, or 0 past the clip, and that frame is never used)
    const bool vec = can_vec && reinterpret_cast<uintptr_t>(spec) % 8 == 0;
    // sweeps streamed together: 8 loads per wave in flight is the measured optimum (profiles/r01_notes.md);
    // a one-sided sweep has 2 loads instead of 4
    constexpr int D1 = TF ? ZAFX_ISTFT_TF_DEPTH : ONE ? 4 : 2, D2 = ONE ? 4 : 1;
    auto kern = vec ? k_istft_ft16<LOG2N, LOG2E, can_vec ? D2 : D1, ONE, can_vec ? 2 : 1, TF> : k_istft_ft16<LOG2N, LOG2E, D1, ONE, 1, TF>;
    const int nt = 1024;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, F::SMEM); e != hipSuccess) return e;
    const int W = 2 << LOG2N;
    const int halo = (W + pl.H - 1) / pl.H - 1;
    if (halo >= kFatFrames) {
        set_error("istft: step_length too small for this window_length (ceil(W/H) exceeds frames per workgroup)");
        return hipErrorInvalidValue;
    }
    const int tiles = (T + kFatFrames - 1) / kFatFrames;
    if ((long long)tiles * n_clips <= 0 || out_len <= 0) return hipSuccess;
    const int per_cu = (int)std::min<size_t>(2, (size_t)kMaxLdsBytes / F::SMEM);
    const long long max_grid = (long long)pl.n_cus * std::max(per_cu, 1);
    const int segs = carry_segments(n_clips, tiles, max_grid);
    const int seg_tiles = (tiles + segs - 1) / segs;
    const long long units = (long long)n_clips * segs;
    const float scale = 1.f / (4.f * (float)(1 << LOG2N) * pl.cola_gain);
    const long long grid = std::min<long long>(units, max_grid);
    pl.ran = "k_istft_ft16";
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(nt), F::SMEM, pl.stream, spec, pl.d_tw_pass, pl.d_tw_aux, y, T, TP, pl.H,
                       (long long)out_len, scale, tiles, segs, seg_tiles, (int)units, halo);
    return hipGetLastError();
}

#ifndef ZAFX_ISTFT_BAND
#define ZAFX_ISTFT_BAND 1
#endif
// k_istft_ft16b: W = 4096 in the reference layout, one band of samples per workgroup (see the kernel)
template <bool ONE>
static hipError_t run_istft_band(const zafx_plan& pl, const float2* spec, float* y, int64_t n_clips, int T, int64_t out_len) {
    using F = FatCfg<10, 4>;
    constexpr int M = 2048;
    const size_t smem = (size_t)(kFatFrames * F::PITCH + FftCfg<10, 4>::TW + M / 2 + 1) * 8 + (size_t)(2048 - pl.H / 2) * 4;
    const bool vec = row_pitch(pl, T) % 2 == 0 && reinterpret_cast<uintptr_t>(spec) % 16 == 0;   // 16-byte row pieces of two adjacent frames
    auto kern = vec ? k_istft_ft16b<1, ONE, 2> : k_istft_ft16b<ONE ? 2 : 1, ONE, 1>;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, smem); e != hipSuccess) return e;
    const int halo = (4096 + pl.H - 1) / pl.H - 1;
    const int tiles = (T + kFatFrames - 1) / kFatFrames;
    if ((long long)tiles * n_clips <= 0 || out_len <= 0) return hipSuccess;
    const long long max_grid = pl.n_cus;
    const int segs = carry_segments(2 * n_clips, tiles, max_grid);   // (two units -- one per band -- for every segment)
    const int seg_tiles = (tiles + segs - 1) / segs;
    const long long units = 2LL * n_clips * segs;
    const float scale = 1.f / (4.f * (float)M * pl.cola_gain);
    const long long grid = std::min<long long>(units, max_grid);
    pl.ran = "k_istft_ft16b";
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(1024), smem, pl.stream, spec, pl.d_tw_sub, pl.d_tw_aux, y, T, (int)row_pitch(pl, T), pl.H / 2,
                       (long long)out_len, scale, tiles, segs, seg_tiles, (int)units, halo);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// inverse, reference layout, W = 8192, hop = W / 2: four CLASSES of rows per 8-frame tile (k_istft_ft8q; round 6)
// ---------------------------------------------------------------------------------
// The transpose of k_stft_ft16q.  The spectrum is the large side (16 bytes of it per output sample), so every row is read ONCE, and the
// rows are taken class by class -- with x_m[n] = v[n + 2048 m] the quarter frames of the real frame v = real(ifft(X)) (zaf.py:223) and Xh the
// Hermitian part (X[k] + conj X[W-k]) / 2 of an arbitrary input:
//   rows 4q            = DFT_2048(s)[q],  s = x0 + x1 + x2 + x3 (real): the W = 2048 inverse (unsplit_pair + one 1024-point transform)
//   rows 8p + 2 (+ 6)  = DFT_1024(g)[p],  g[n] = (r[n] - i r[n + 1024]) w_4096^n,  r = x0 - x1 + x2 - x3
//   rows 8p + 1 (+ 7)  = DFT_1024(u)[p],  u[n] = h[n] + h[n + 1024],  h[n] = ((x0 - x2) - i (x1 - x3))[n] w_8192^n
//   rows 8p + 5 (+ 3)  = DFT_1024(d)[p],  d[n] = (h[n] - h[n + 1024]) w_2048^n
// (the rows in parentheses are the mirrors that complete the Hermitian part).  Each class is ONE 1024-point inverse transform per frame: a
// round is gather + fold by the whole workgroup, then a wavefront transforms its frame in its own buffer and reads ITS share of the result
// into registers -- lane l keeps the indices n = 4 l + 256 i + c (+ 1024) of s, r, u: 96 registers through the four rounds -- and the quarter
// frames come out of four-point butterflies in registers: x0, x2 = s/4 + r/4 +- e/2, x1, x3 = s/4 - r/4 +- o/2.  Neither the four rounds of partial
// results nor the tile's output fit LDS, and at 1024 threads not the registers either: hence 8-frame tiles (64-byte row pieces), 8 waves, 256
// registers.  Overlap-add in the reference's ascending frame order (zaf.py:226-233): a frame's second half goes through LDS to the wave of
// the next frame (the tile's last one to the next tile), which adds its first half and stores 16-byte pieces; the trim of zaf.py:236-238
// drops frame 0's first half and the last frame's second.
#ifndef ZAFX_ISTFT_QUAD
#define ZAFX_ISTFT_QUAD 1
#endif
#ifndef ZAFX_ISTFT_QUAD_FV
#define ZAFX_ISTFT_QUAD_FV 2   // frames per lane and load where the row pitch is even
#endif
#ifndef ZAFX_ISTFT_QUAD_DEPTH
#define ZAFX_ISTFT_QUAD_DEPTH 2   // sweeps of a class's gather in flight per thread (two loads each; class A: four loads, half as many sweeps): 1 / 2 / 3 / 4 / 8 / 16 measured 3.32 / 3.00 / 3.14 / 3.35 / 3.15 / 5.01 ms -- shallow queues, as k_istft_ft16 found
#endif
#ifndef ZAFX_ISTFT_QUAD_PF
#define ZAFX_ISTFT_QUAD_PF 0   // sweeps (of 16) of a class requested ahead of the transform before it: 2 / 4 / 8 measured 3.59 / 3.50 / 3.79 ms against 3.35 (they spill)
#endif
struct IstftQCfg {
    using C = FftCfg<10, 4>;
    static constexpr int N = C::N, FPB = 8, NT = 512, PITCH = C::PITCH, HALF = 4096;
    static constexpr size_t REGION = (size_t)(FPB - 1) * HALF * 4 > (size_t)FPB * PITCH * 8 ? (size_t)(FPB - 1) * HALF * 4 : (size_t)FPB * PITCH * 8;
    static constexpr size_t SMEM = REGION + 2 * HALF * 4 + (size_t)(C::TW + N / 2 + 1) * 8;
};
static_assert(IstftQCfg::SMEM <= (size_t)kMaxLdsBytes, "k_istft_ft8q: tile + carries + tables exceed LDS");

// FV: frames per lane and load (2: 16-byte loads of two adjacent frames; needs an even row pitch)
template <bool ONE, int FV>
__global__ __launch_bounds__(IstftQCfg::NT) void k_istft_ft8q(const float2* __restrict__ spec, const float2* __restrict__ twp, const float2* __restrict__ twq,
                                                               float* __restrict__ y, int T, int TP, long long out_len, float scale, int tiles, int segs,
                                                               int seg_tiles, int total_units) {
    using Q = IstftQCfg;
    using C = Q::C;
    constexpr int N = C::N, P = 64, E = C::E, W = 8192, FPB = Q::FPB, NT = Q::NT, PITCH = Q::PITCH, HALF = Q::HALF, ROWS = ONE ? W / 2 + 1 : W;
    static_assert(C::P == 64, "a frame's 1024-point transform is one wavefront's");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* frames = reinterpret_cast<float2*>(smem_raw);        // FPB transform buffers ...
    float* xch = reinterpret_cast<float*>(smem_raw);              // ... and, behind the fourth round, FPB - 1 second halves of frames
    float* carry = reinterpret_cast<float*>(smem_raw + Q::REGION);   // the tile's last second half, by tile parity
    float2* tw_l = reinterpret_cast<float2*>(carry + 2 * HALF);
    float2* tws_l = tw_l + C::TW;                                 // exp(-2 pi i k / 2048), k <= 512: the split roots of class A
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    constexpr int LPR = FPB / FV, KS = NT / LPR;                  // lanes per row piece, rows per sweep (64 or 128)
    const int fr = (tid % LPR) * FV, kq = tid / LPR;              // gather: my first frame of the tile, my row within a sweep
    for (int i = tid; i < C::TW; i += NT) tw_l[i] = twp[i];
    for (int i = tid; i <= N / 2; i += NT) tws_l[i] = twq[4 * i];
    for (int i = tid; i < 2 * HALF; i += NT) carry[i] = 0.f;
    lds_barrier();
    const int row_bytes = TP * 8;
    float2* const fb = frames + fr * PITCH;      // the buffer my gathers fold into
    float2* const buf = frames + wave * PITCH;   // the buffer my wave transforms

    PROF_INIT(g_prof);
    for (int unit = blockIdx.x; unit < total_units; unit += gridDim.x) {
        const int seg = unit % segs;
        const long long clip = unit / segs;
        const int tile_a = seg * seg_tiles, tile_b = min(tile_a + seg_tiles, tiles);
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(spec) + clip * ROWS * TP, 0, ROWS * row_bytes, 0x00020000);
        float* const yc = y + clip * out_len;
        for (int tile = tile_a > 0 ? tile_a - 1 : 0; tile < tile_b; ++tile) {
            const bool write_out = tile >= tile_a;   // (the tile in front of a segment only leaves its last second half behind)
            const int t0 = tile * FPB, par = tile & 1;
            int lo = lane;
            asm volatile("" : "+v"(lo));   // (opaque per tile: addresses are recomputed, not carried -- spilled -- across the rounds)
            const int tb = (t0 + fr) * 8;  // byte offset of my frame within a row
            using RV = std::conditional_t<FV == 2, float4, float2>;   // X[row] of my FV frames
            auto raw = [&](int row) {
                RV f;
                if constexpr (FV == 2) {
                    const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, row * row_bytes + tb, 0, 0);
                    __builtin_memcpy(&f, &v, 16);
                } else {
                    const auto v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, row * row_bytes + tb, 0, 0);
                    __builtin_memcpy(&f, &v, 8);
                }
                return f;
            };
            auto conj_rv = [](RV v) {
                if constexpr (FV == 2) return make_float4(v.x, -v.y, v.z, -v.w);
                else return cconj(v);
            };
            auto part = [](RV v, int f) {   // frame f of a piece
                if constexpr (FV == 2) return f ? make_float2(v.z, v.w) : make_float2(v.x, v.y);
                else return v;
            };
            auto xat = [&](int k) {        // X[k], k < W: a one-sided input completes X[W - k] = conj X[k]
                if (ONE && k > W / 2) return conj_rv(raw(W - k));
                return raw(k);
            };
            // lane l of a wave keeps the indices j = 4 l + 256 i + c (c < 4, i < 4) and j + 1024 of its frame's sequences
            float sA[4][2][4], rB[4][2][4];
            float2 uC[4][4];
            float own[4][2][2][4];   // [i][m = 0, 1][h][c]: first half of the frame (x0, x1)
            // ---- classes B, C, D: rows 8p + c0 and their mirrors W - (8p + c0): 2 Xh[8p + c0], swapped.  The first PF sweeps of a class are
            //      requested in front of the transform of the class before it and ride in registers across it (issue_class / fold_class).
            constexpr int PF = ZAFX_ISTFT_QUAD_PF;
            RV pa[PF > 0 ? PF : 1], pb[PF > 0 ? PF : 1];
            auto issue_class = [&](int c0) {
#pragma unroll
                for (int sw = 0; sw < PF; ++sw) {
                    const int a = 8 * (kq + KS * sw) + c0;
                    if (ONE) pa[sw] = xat(a);
                    else pa[sw] = raw(a), pb[sw] = raw(W - a);
                }
            };
            auto fold_row = [&](int p, RV va, RV vb) {
#pragma unroll
                for (int f = 0; f < FV; ++f) {
                    const float2 qa = part(va, f), qb = part(vb, f);
                    const float2 g = ONE ? make_float2(2.f * qa.x, 2.f * qa.y) : make_float2(qa.x + qb.x, qa.y - qb.y);
                    fb[f * PITCH + phys(p)] = make_float2(g.y, g.x);
                }
            };
            auto fold_class = [&](int c0) {
                lds_barrier();   // every wave has read the class before
#pragma unroll
                for (int sw = 0; sw < PF; ++sw) fold_row(kq + KS * sw, pa[sw], pb[sw]);
#pragma unroll ZAFX_ISTFT_QUAD_DEPTH
                for (int sw = PF; sw < 1024 / KS; ++sw) {
                    const int p = kq + KS * sw, a = 8 * p + c0;
                    if (ONE) fold_row(p, xat(a), RV{});
                    else fold_row(p, raw(a), raw(W - a));
                }
            };
            PROF_MARK(0);
            // ---- class A: rows 4q, the W = 2048 inverse
            constexpr int kDepthA = ZAFX_ISTFT_QUAD_DEPTH / 2 > 0 ? ZAFX_ISTFT_QUAD_DEPTH / 2 : 1;   // (class A issues four loads per sweep: half as many sweeps in flight)
#pragma unroll kDepthA
            for (int sw = 0; sw < 512 / KS; ++sw) {
                const int k = kq + KS * sw;
                if (k == 0) {
                    const RV r0 = xat(0), r1 = xat(2048), r2 = xat(4096), r3 = ONE ? conj_rv(r1) : xat(6144);
#pragma unroll
                    for (int f = 0; f < FV; ++f) {
                        const float2 q0 = part(r0, f), q1 = part(r1, f), q2 = part(r2, f), q3 = part(r3, f);
                        const float a0 = 2.f * q0.x, an = 2.f * q2.x;
                        fb[f * PITCH] = make_float2(a0 - an, a0 + an);
                        const float2 a = make_float2(q1.x + q3.x, q1.y - q3.y);
                        fb[f * PITCH + phys(N / 2)] = make_float2(-2.f * a.y, 2.f * a.x);
                    }
                } else {
                    const RV r0 = xat(4 * k), r2 = xat(4096 - 4 * k);
                    const RV r1 = ONE ? conj_rv(r0) : xat(W - 4 * k), r3 = ONE ? conj_rv(r2) : xat(4096 + 4 * k);
#pragma unroll
                    for (int f = 0; f < FV; ++f) {
                        float2 zk, zn;
                        unsplit_pair(part(r0, f), part(r1, f), part(r2, f), part(r3, f), tws_l[k], zk, zn);
                        fb[f * PITCH + phys(k)] = zk;
                        fb[f * PITCH + phys(N - k)] = zn;
                    }
                }
            }
            auto transform = [&]() {   // my wave's frame: forward transform of the swapped spectrum = swapped inverse transform
                lds_barrier();
                float2 v[E];
                regs_read<10, 4>(v, buf, lo);
                frame_sync<P>();
                fft_frame<10, 4>(v, buf, lo, tw_l);
                frame_sync<P>();
            };
            PROF_MARK(1);
            issue_class(2);
            transform();
            PROF_MARK(2);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float2 q0 = buf[phys(2 * lo + 128 * i + 512 * h)], q1 = buf[phys(2 * lo + 1 + 128 * i + 512 * h)];   // (s[2n + 1], s[2n])
                    sA[i][h][0] = q0.y, sA[i][h][1] = q0.x, sA[i][h][2] = q1.y, sA[i][h][3] = q1.x;
                }
            float2 w8[4][4];   // exp(-2 pi i j / 8192) of my indices
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) w8[i][c] = twq[4 * lo + 256 * i + c];
            PROF_MARK(3);
            fold_class(2);
            PROF_MARK(4);
            issue_class(1);
            transform();
            PROF_MARK(5);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float2 q = buf[phys(4 * lo + 256 * i + c)];
                    const float2 w4 = cmul(w8[i][c], w8[i][c]);                  // w_4096^j
                    const float2 cb = cmulc(make_float2(q.y, q.x), w4);          // g conj(w_4096^j) = 2 (r[j] - i r[j + 1024])
                    rB[i][0][c] = cb.x, rB[i][1][c] = -cb.y;
                }
            fold_class(1);
            PROF_MARK(6);
            issue_class(5);
            transform();
            PROF_MARK(7);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float2 q = buf[phys(4 * lo + 256 * i + c)];
                    uC[i][c] = make_float2(q.y, q.x);
                }
            fold_class(5);
            PROF_MARK(8);
            transform();
            PROF_MARK(9);
            float sec[4][2][2][4];   // [i][m - 2][h][c]: second half of the frame (x2, x3)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float2 q = buf[phys(4 * lo + 256 * i + c)];
                    const float2 w1 = w8[i][c], w4 = cmul(w1, w1), w2 = cmul(w4, w4);          // w_8192^j, w_4096^j, w_2048^j
                    const float2 dd = cmulc(make_float2(q.y, q.x), w2);                        // d conj(w_2048^j)
                    const float2 h0 = make_float2(uC[i][c].x + dd.x, uC[i][c].y + dd.y), h1 = make_float2(uC[i][c].x - dd.x, uC[i][c].y - dd.y);   // 2 h[j], 2 h[j + 1024]
                    const float2 w1b = cmul(w1, make_float2(0.70710678118654752440f, -0.70710678118654752440f));   // w_8192^(j + 1024)
                    const float2 c0 = cmulc(h0, w1), c1 = cmulc(h1, w1b);                      // ((x0 - x2) - i (x1 - x3)) x 4, at j and j + 1024
                    const float e[2] = {c0.x, c1.x}, o[2] = {-c0.y, -c1.y};
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float b = 0.5f * sA[i][h][c], r = rB[i][h][c];
                        own[i][0][h][c] = (b + r) + e[h];   // x0
                        sec[i][0][h][c] = (b + r) - e[h];   // x2
                        own[i][1][h][c] = (b - r) + o[h];   // x1
                        sec[i][1][h][c] = (b - r) - o[h];   // x3
                    }
                }
            PROF_MARK(10);
            lds_barrier();   // every wave has read its class D: the buffers become the exchange area
            {
                float* dst = wave == FPB - 1 ? carry + par * HALF : xch + wave * HALF;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int h = 0; h < 2; ++h)
                            *reinterpret_cast<float4*>(dst + 2048 * m + 1024 * h + 256 * i + 4 * lo) =
                                make_float4(sec[i][m][h][0], sec[i][m][h][1], sec[i][m][h][2], sec[i][m][h][3]);
            }
            lds_barrier();
            {
                const int t = t0 + wave;
                const float* src = wave == 0 ? carry + (par ^ 1) * HALF : xch + (wave - 1) * HALF;
                const bool store = write_out && t >= 1 && t < T;   // frame t's first half + frame t - 1's second = samples (t - 1) hop ... t hop of the trimmed clip
                float* o = yc + (long long)(t - 1) * HALF + 4 * lo;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const float4 pv = *reinterpret_cast<const float4*>(src + 2048 * m + 1024 * h + 256 * i + 4 * lo);
                            const float4 v = make_float4((pv.x + own[i][m][h][0]) * scale, (pv.y + own[i][m][h][1]) * scale, (pv.z + own[i][m][h][2]) * scale,
                                                         (pv.w + own[i][m][h][3]) * scale);
                            if (store) *reinterpret_cast<float4*>(o + 2048 * m + 1024 * h + 256 * i) = v;
                        }
            }
            lds_barrier();   // the exchange area is read: the next tile folds into it
            PROF_MARK(11);
        }
        // (a unit's first tile reads the carry of the tile before it: a clip starts with zeros there)
        lds_barrier();
        for (int i = tid; i < 2 * HALF; i += NT) carry[i] = 0.f;
        lds_barrier();
    }
}

template <bool ONE>
static hipError_t run_istft_quad(const zafx_plan& pl, const float2* spec, float* y, int64_t n_clips, int T, int64_t out_len) {
    using Q = IstftQCfg;
    const int tiles = (T + Q::FPB - 1) / Q::FPB;
    if ((long long)tiles * n_clips <= 0 || out_len <= 0) return hipSuccess;
    const bool wide = ZAFX_ISTFT_QUAD_FV == 2 && row_pitch(pl, T) % 2 == 0 && reinterpret_cast<uintptr_t>(spec) % 16 == 0;   // 16-byte pieces of two frames
    auto kern = wide ? k_istft_ft8q<ONE, 2> : k_istft_ft8q<ONE, 1>;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, Q::SMEM); e != hipSuccess) return e;
    const int segs = carry_segments(n_clips, tiles, pl.n_cus);
    const int seg_tiles = (tiles + segs - 1) / segs;
    const long long units = n_clips * segs;
    const float scale = 1.f / (8192.f * pl.cola_gain);   // 1 / W of the inverse transform (the classes carry 2 Xh); zaf.py:241
    pl.ran = "k_istft_ft8q";
    hipLaunchKernelGGL(kern, dim3((unsigned)std::min<long long>(units, pl.n_cus)), dim3(Q::NT), Q::SMEM, pl.stream, spec, pl.d_tw_sub, pl.d_tw_quad, y, T,
                       (int)row_pitch(pl, T), (long long)out_len, scale, tiles, segs, seg_tiles, (int)units);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// inverse, reference layout, W = 4096, hop = W / 2: two CLASSES of rows per 16-frame tile (k_istft_ft16d; round 6)
// ---------------------------------------------------------------------------------
// k_istft_ft8q's first two classes are the whole of W = 4096: with x0, x1 the halves of the real frame v = real(ifft(X)) (zaf.py:223) and Xh the
// Hermitian part of an arbitrary input,
//   rows 2q           = DFT_2048(s)[q],  s = x0 + x1 (real): the W = 2048 inverse (unsplit_pair + one 1024-point transform)
//   rows 4p + 1 (+ 3) = DFT_1024(u)[p],  u[n] = (d[n] - i d[n + 1024]) w_4096^n,  d = x0 - x1
// and x0 = (s + d) / 2, x1 = (s - d) / 2 in registers (a lane keeps its 32 values of s across the second round).  Every row is read ONCE --
// k_istft_ft16b runs one band of SAMPLES per workgroup and each band needs every row: 1.79 x the algorithmic bytes -- as 128-byte pieces (16
// frames), a wavefront transforms its frame in its own buffer, and the overlap-add (zaf.py:226-233, hop = W / 2: a frame's second half + the next
// frame's first) goes through LDS to the wave of the next frame (the tile's last one: to the next tile) with 16-byte stores.
#ifndef ZAFX_ISTFT_DUO
#define ZAFX_ISTFT_DUO 1
#endif
#ifndef ZAFX_ISTFT_DUO_DEPTH
#define ZAFX_ISTFT_DUO_DEPTH 1   // sweeps of a class's gather in flight per thread (1 / 2 / 4: 1.87 / 1.92 / 1.98 ms on padded rows, 2.95 / 3.06 / 3.53 compact at T = 217)
#endif
struct IstftDCfg {
    using C = FftCfg<10, 4>;
    static constexpr int N = C::N, FPB = 16, NT = 1024, PITCH = C::PITCH, HALF = 2048;
    static constexpr size_t REGION = (size_t)FPB * PITCH * 8;   // FPB transform buffers; behind the second round FPB - 1 second halves (8 KB each)
    static constexpr size_t SMEM = REGION + (size_t)HALF * 4 + (size_t)(C::TW + N) * 8;   // buffers | carry | pass twiddles | exp(-2 pi i k / 4096), k < 1024
};
static_assert(IstftDCfg::SMEM <= (size_t)kMaxLdsBytes && IstftDCfg::REGION >= (size_t)(IstftDCfg::FPB - 1) * IstftDCfg::HALF * 4, "k_istft_ft16d: tile + carry + tables exceed LDS");

// FV: frames per lane and load (2: 16-byte loads of two adjacent frames; needs an even row pitch)
template <bool ONE, int FV>
__global__ __launch_bounds__(IstftDCfg::NT) void k_istft_ft16d(const float2* __restrict__ spec, const float2* __restrict__ twp, const float2* __restrict__ tww,
                                                                float* __restrict__ y, int T, int TP, long long out_len, float scale, int tiles, int segs,
                                                                int seg_tiles, int total_units) {
    using Q = IstftDCfg;
    using C = Q::C;
    constexpr int N = C::N, P = 64, E = C::E, W = 4096, FPB = Q::FPB, NT = Q::NT, PITCH = Q::PITCH, HALF = Q::HALF, ROWS = ONE ? W / 2 + 1 : W;
    static_assert(C::P == 64, "a frame's 1024-point transform is one wavefront's");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* frames = reinterpret_cast<float2*>(smem_raw);        // FPB transform buffers ...
    float* xch = reinterpret_cast<float*>(smem_raw);              // ... and, behind the second round, FPB - 1 second halves of frames
    float* carry = reinterpret_cast<float*>(smem_raw + Q::REGION);   // the second half of the frame before the tile
    float2* tw_l = reinterpret_cast<float2*>(carry + HALF);
    float2* tww_l = tw_l + C::TW;                                 // exp(-2 pi i k / 4096), k < 1024: the second class's roots; [2k] = the first class's split roots
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    constexpr int LPR = FPB / FV, KS = NT / LPR;                  // lanes per row piece, rows per sweep (128 or 64)
    const int fr = (tid % LPR) * FV, kq = tid / LPR;              // gather: my first frame of the tile, my row within a sweep
    for (int i = tid; i < C::TW; i += NT) tw_l[i] = twp[i];
    for (int i = tid; i < N; i += NT) tww_l[i] = tww[i];   // (the plan's real-split roots)
    for (int i = tid; i < HALF; i += NT) carry[i] = 0.f;
    lds_barrier();
    const int row_bytes = TP * 8;
    float2* const fb = frames + fr * PITCH;      // the buffer my gathers fold into
    float2* const buf = frames + wave * PITCH;   // the buffer my wave transforms

    for (int unit = blockIdx.x; unit < total_units; unit += gridDim.x) {
        const int seg = unit % segs;
        const long long clip = unit / segs;
        const int tile_a = seg * seg_tiles, tile_b = min(tile_a + seg_tiles, tiles);
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(spec) + clip * ROWS * TP, 0, ROWS * row_bytes, 0x00020000);
        float* const yc = y + clip * out_len;
        for (int tile = tile_a > 0 ? tile_a - 1 : 0; tile < tile_b; ++tile) {
            const bool write_out = tile >= tile_a;   // (the tile in front of a segment only leaves its last second half behind)
            const int t0 = tile * FPB;
            int lo = lane;
            asm volatile("" : "+v"(lo));   // (opaque per tile: addresses are recomputed, not carried -- spilled -- across the rounds)
            const int tb = (t0 + fr) * 8;  // byte offset of my frame within a row
            using RV = std::conditional_t<FV == 2, float4, float2>;   // X[row] of my FV frames
            auto raw = [&](int row) {
                RV f;
                if constexpr (FV == 2) {
                    const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, row * row_bytes + tb, 0, 0);
                    __builtin_memcpy(&f, &v, 16);
                } else {
                    const auto v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, row * row_bytes + tb, 0, 0);
                    __builtin_memcpy(&f, &v, 8);
                }
                return f;
            };
            auto conj_rv = [](RV v) {
                if constexpr (FV == 2) return make_float4(v.x, -v.y, v.z, -v.w);
                else return cconj(v);
            };
            auto part = [](RV v, int f) {   // frame f of a piece
                if constexpr (FV == 2) return f ? make_float2(v.z, v.w) : make_float2(v.x, v.y);
                else return v;
            };
            auto xat = [&](int k) {        // X[k], k < W: a one-sided input completes X[W - k] = conj X[k]
                if (ONE && k > W / 2) return conj_rv(raw(W - k));
                return raw(k);
            };
            // ---- first class: rows 2q, the W = 2048 inverse
#pragma unroll ZAFX_ISTFT_DUO_DEPTH
            for (int sw = 0; sw < 512 / KS; ++sw) {
                const int k = kq + KS * sw;
                if (k == 0) {
                    const RV r0 = xat(0), r1 = xat(1024), r2 = xat(2048), r3 = ONE ? conj_rv(r1) : xat(3072);
#pragma unroll
                    for (int f = 0; f < FV; ++f) {
                        const float2 q0 = part(r0, f), q1 = part(r1, f), q2 = part(r2, f), q3 = part(r3, f);
                        const float a0 = 2.f * q0.x, an = 2.f * q2.x;
                        fb[f * PITCH] = make_float2(a0 - an, a0 + an);
                        const float2 a = make_float2(q1.x + q3.x, q1.y - q3.y);
                        fb[f * PITCH + phys(N / 2)] = make_float2(-2.f * a.y, 2.f * a.x);
                    }
                } else {
                    const RV r0 = xat(2 * k), r2 = xat(2048 - 2 * k);
                    const RV r1 = ONE ? conj_rv(r0) : xat(W - 2 * k), r3 = ONE ? conj_rv(r2) : xat(2048 + 2 * k);
#pragma unroll
                    for (int f = 0; f < FV; ++f) {
                        float2 zk, zn;
                        unsplit_pair(part(r0, f), part(r1, f), part(r2, f), part(r3, f), tww_l[2 * k], zk, zn);
                        fb[f * PITCH + phys(k)] = zk;
                        fb[f * PITCH + phys(N - k)] = zn;
                    }
                }
            }
            auto transform = [&]() {   // my wave's frame: forward transform of the swapped spectrum = swapped inverse transform
                lds_barrier();
                float2 v[E];
                regs_read<10, 4>(v, buf, lo);
                frame_sync<P>();
                fft_frame<10, 4>(v, buf, lo, tw_l);
                frame_sync<P>();
            };
            transform();
            // lane l of a wave keeps the indices j = 4 l + 256 i + c (c < 4, i < 4) and j + 1024 of its frame's sequences
            float sA[4][2][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float2 q0 = buf[phys(2 * lo + 128 * i + 512 * h)], q1 = buf[phys(2 * lo + 1 + 128 * i + 512 * h)];   // (s[2n + 1], s[2n])
                    sA[i][h][0] = q0.y, sA[i][h][1] = q0.x, sA[i][h][2] = q1.y, sA[i][h][3] = q1.x;
                }
            // ---- second class: rows 4p + 1 and their mirrors W - (4p + 1): 2 Xh[4p + 1], swapped
            lds_barrier();   // every wave has read the first class
#pragma unroll ZAFX_ISTFT_DUO_DEPTH
            for (int sw = 0; sw < 1024 / KS; ++sw) {
                const int p = kq + KS * sw, a = 4 * p + 1;
                const RV va = ONE ? xat(a) : raw(a), vb = ONE ? RV{} : raw(W - a);
#pragma unroll
                for (int f = 0; f < FV; ++f) {
                    const float2 qa = part(va, f), qb = part(vb, f);
                    const float2 g = ONE ? make_float2(2.f * qa.x, 2.f * qa.y) : make_float2(qa.x + qb.x, qa.y - qb.y);
                    fb[f * PITCH + phys(p)] = make_float2(g.y, g.x);
                }
            }
            transform();
            float own[4][2][4], sec[4][2][4];   // [i][h][c]: the frame's first half x0 and second half x1 at j + 1024 h
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int j = 4 * lo + 256 * i + c;
                    const float2 q = buf[phys(j)];
                    const float2 cb = cmulc(make_float2(q.y, q.x), tww_l[j]);   // u conj(w_4096^j) = 2 (d[j] - i d[j + 1024])
                    const float r[2] = {cb.x, -cb.y};
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float b = 0.5f * sA[i][h][c];
                        own[i][h][c] = b + r[h];
                        sec[i][h][c] = b - r[h];
                    }
                }
            lds_barrier();   // every wave has read its second class: the buffers become the exchange area
            if (wave < FPB - 1) {
                float* dst = xch + wave * HALF;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        *reinterpret_cast<float4*>(dst + 1024 * h + 256 * i + 4 * lo) = make_float4(sec[i][h][0], sec[i][h][1], sec[i][h][2], sec[i][h][3]);
            }
            lds_barrier();
            {
                const int t = t0 + wave;
                const float* src = wave == 0 ? carry : xch + (wave - 1) * HALF;
                const bool store = write_out && t >= 1 && t < T;   // frame t's first half + frame t - 1's second = samples (t - 1) hop ... t hop of the trimmed clip
                float* o = yc + (long long)(t - 1) * HALF + 4 * lo;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float4 pv = *reinterpret_cast<const float4*>(src + 1024 * h + 256 * i + 4 * lo);
                        const float4 v = make_float4((pv.x + own[i][h][0]) * scale, (pv.y + own[i][h][1]) * scale, (pv.z + own[i][h][2]) * scale,
                                                     (pv.w + own[i][h][3]) * scale);
                        if (store) *reinterpret_cast<float4*>(o + 1024 * h + 256 * i) = v;
                    }
            }
            lds_barrier();   // the exchange area and the carry are read: the next tile folds into the one, the tile's last frame leaves its second half in the other
            if (wave == FPB - 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        *reinterpret_cast<float4*>(carry + 1024 * h + 256 * i + 4 * lo) = make_float4(sec[i][h][0], sec[i][h][1], sec[i][h][2], sec[i][h][3]);
            }
        }
        // (a unit's first tile reads the carry of the tile before it: a clip starts with zeros there)
        lds_barrier();
        for (int i = tid; i < HALF; i += NT) carry[i] = 0.f;
        lds_barrier();
    }
}

template <bool ONE>
static hipError_t run_istft_duo(const zafx_plan& pl, const float2* spec, float* y, int64_t n_clips, int T, int64_t out_len) {
    using Q = IstftDCfg;
    const int tiles = (T + Q::FPB - 1) / Q::FPB;
    if ((long long)tiles * n_clips <= 0 || out_len <= 0) return hipSuccess;
    const bool wide = row_pitch(pl, T) % 2 == 0 && reinterpret_cast<uintptr_t>(spec) % 16 == 0;   // 16-byte pieces of two frames
    auto kern = wide ? k_istft_ft16d<ONE, 2> : k_istft_ft16d<ONE, 1>;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, Q::SMEM); e != hipSuccess) return e;
    const int segs = carry_segments(n_clips, tiles, pl.n_cus);
    const int seg_tiles = (tiles + segs - 1) / segs;
    const long long units = n_clips * segs;
    const float scale = 1.f / (4096.f * pl.cola_gain);   // 1 / W of the inverse transform (the classes carry 2 Xh); zaf.py:241
    pl.ran = "k_istft_ft16d";
    hipLaunchKernelGGL(kern, dim3((unsigned)std::min<long long>(units, pl.n_cus)), dim3(Q::NT), Q::SMEM, pl.stream, spec, pl.d_tw_sub, pl.d_tw_aux, y, T,
                       (int)row_pitch(pl, T), (long long)out_len, scale, tiles, segs, seg_tiles, (int)units);
    return hipGetLastError();
}

template <int LOG2N, int LAYOUT, bool ONE>
static hipError_t run_istft(const zafx_plan& pl, const float2* spec, float* y, int64_t n_clips, int T, int64_t out_len) {
    if constexpr (ZAFX_ISTFT_DUO && LOG2N == 11 && LAYOUT == ZAFX_LAYOUT_FT) {
        // W = 4096, hop = W / 2: the two-class kernel (32-bit byte offsets inside a clip's spectrum, 16-byte stores into the clip's samples)
        const int64_t TP = row_pitch(pl, T);
        if (pl.d_tw_sub && pl.d_tw_aux && pl.H == 2048 && reinterpret_cast<uintptr_t>(spec) % 8 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0 &&
            (long long)4096 * TP * 8 < (1LL << 31) && n_clips * (((long long)T + 15) / 16 + 1) < (1LL << 30))
            return run_istft_duo<ONE>(pl, spec, y, n_clips, T, out_len);
    }
    if constexpr (ZAFX_ISTFT_BAND && LOG2N == 11 && LAYOUT == ZAFX_LAYOUT_FT) {
        // W = 4096: the band form wants a hop that is a multiple of 4 (a sample keeps its residue mod 4 from frame to frame), at least 512
        // (ceil(W / H) - 1 <= 7 halo frames; the carry fits LDS), 32-bit byte offsets inside a clip's spectrum
        const int64_t TP = row_pitch(pl, T);
        if (pl.d_tw_sub && pl.H % 4 == 0 && pl.H >= 512 && pl.H <= 4096 && reinterpret_cast<uintptr_t>(spec) % 8 == 0 &&
            (long long)4096 * TP * 8 < (1LL << 31) && 2LL * n_clips * ((T + 15) / 16) < (1LL << 30))
            return run_istft_band<ONE>(pl, spec, y, n_clips, T, out_len);
    }
    if constexpr (ZAFX_ISTFT_QUAD && LOG2N == 12 && LAYOUT == ZAFX_LAYOUT_FT) {
        // W = 8192, hop = W / 2: the four-class kernel (32-bit byte offsets inside a clip's spectrum, 16-byte stores into the clip's samples)
        const int64_t TP = row_pitch(pl, T);
        if (pl.d_tw_sub && pl.d_tw_quad && pl.H == 4096 && reinterpret_cast<uintptr_t>(spec) % 8 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0 &&
            (long long)8192 * TP * 8 < (1LL << 31) && n_clips * (((long long)T + 7) / 8 + 1) < (1LL << 30))
            return run_istft_quad<ONE>(pl, spec, y, n_clips, T, out_len);
    }
    if constexpr (stft_use_fat(LOG2N, LAYOUT)) {
        // the carry kernel addresses a clip through one buffer descriptor (32-bit byte offsets)
        if ((long long)(2 << LOG2N) * row_pitch(pl, T) * 8 < (1LL << 31)) return run_istft_fat<LOG2N, ONE>(pl, spec, y, n_clips, T, out_len);
    } else if constexpr (stft_use_tf(LOG2N, LAYOUT)) {
        return run_istft_fat<LOG2N, ONE, true>(pl, spec, y, n_clips, T, out_len);
    }
    constexpr int LOG2E = default_log2e(LOG2N);
    constexpr int FPB = stft_fpb(LOG2N, ZAFX_LAYOUT_FT);
    using S = StftCfg<LOG2N, LOG2E, FPB>;
    auto kern = k_istft<LOG2N, LOG2E, FPB, LAYOUT, ONE>;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, S::SMEM); e != hipSuccess) return e;
    const int W = 2 << LOG2N;
    const int halo = (W + pl.H - 1) / pl.H - 1;
    const int owned = FPB - halo;
    if (owned < 1) {
        set_error("istft: step_length too small for this window_length (ceil(W/H) exceeds frames per workgroup)");
        return hipErrorInvalidValue;
    }
    const int tiles = (T + owned - 1) / owned;
    const long long blocks = (long long)tiles * n_clips;
    if (blocks <= 0 || out_len <= 0) return hipSuccess;
    const float scale = 1.f / (4.f * (float)(1 << LOG2N) * pl.cola_gain);
    pl.ran = "k_istft";
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(S::NT), S::SMEM, pl.stream, spec, pl.d_tw_pass, pl.d_tw_aux, y, T,
                       (int)row_pitch(pl, T), pl.H, (long long)out_len, scale, tiles, owned, halo);
    return hipGetLastError();
}

#define ZAFX_STFT_SIZES(X) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12)

bool stft_supported(int log2n) { return log2n >= 5 && log2n <= 12; }
int stft_frames_per_block(int log2n, int layout) { return stft_fpb(log2n, layout); }
// int16 PCM (one or two channels) straight into the complex STFT at W = 2048 in the reference layout (k_stft_ft16 / k_stft_ft16c): even clip length and
// hop, everything inside the 32-bit offsets of one descriptor (zafx_execute_pcm)
bool stft_pcm_direct_ok(const zafx_plan& pl, int64_t n_frames, int n_channels, int sample_bytes, const void* d_pcm, int T) {
    return sample_bytes == 2 && (n_channels == 1 || n_channels == 2) && pl.kind == ZAFX_STFT && pl.prm.precision == ZAFX_PRECISION_F32 && pl.bs_log2m == 0 &&
           pl.log2nf == 10 && pl.layout == ZAFX_LAYOUT_FT && pl.prm.spectrum <= ZAFX_SPECTRUM_ONE_SIDED && n_frames % 2 == 0 && pl.H % 2 == 0 &&
           reinterpret_cast<uintptr_t>(d_pcm) % 8 == 0 && n_frames < (1LL << 29) && (long long)T * pl.H < (1LL << 29);
}

const char* stft_kernel_name(int log2n, int layout) {
    if (ZAFX_STFT_FAT8 && log2n == 11 && layout == ZAFX_LAYOUT_FT) return "k_stft_ft16";
    if (ZAFX_STFT_BAND && log2n == 11 && layout == ZAFX_LAYOUT_FT) return "k_stft_ft16b";
    if (ZAFX_STFT_QUAD && log2n == 12 && layout == ZAFX_LAYOUT_FT) return "k_stft_ft16q";   // (planned: the four-class family; what RAN is last_kernel)
    return stft_use_fat(log2n, layout) ? "k_stft_ft16" : stft_use_tf(log2n, layout) ? "k_stft_tf" : "k_stft";
}
const char* istft_kernel_name(int log2n, int layout) {
    if (ZAFX_ISTFT_BAND && log2n == 11 && layout == ZAFX_LAYOUT_FT) return "k_istft_ft16b";
    return stft_use_fat(log2n, layout) || stft_use_tf(log2n, layout) ? "k_istft_ft16" : "k_istft";
}

template <int L, int LAYOUT>
static hipError_t dispatch_stft_spec(const zafx_plan& pl, const float* x, float2* out, int64_t n_clips, int64_t n_samples, int T) {
    switch (pl.prm.spectrum) {
        case ZAFX_SPECTRUM_ONE_SIDED: return run_stft<L, LAYOUT, 1>(pl, x, out, n_clips, n_samples, T);
        case ZAFX_SPECTRUM_MAGNITUDE: return run_stft<L, LAYOUT, 2>(pl, x, out, n_clips, n_samples, T);
        case ZAFX_SPECTRUM_POWER: return run_stft<L, LAYOUT, 3>(pl, x, out, n_clips, n_samples, T);
        default: return run_stft<L, LAYOUT, 0>(pl, x, out, n_clips, n_samples, T);
    }
}

template <int L>
static hipError_t dispatch_stft(const zafx_plan& pl, const float* x, float2* out, int64_t n_clips, int64_t n_samples, int T) {
    return pl.layout == ZAFX_LAYOUT_FT ? dispatch_stft_spec<L, ZAFX_LAYOUT_FT>(pl, x, out, n_clips, n_samples, T)
                                       : dispatch_stft_spec<L, ZAFX_LAYOUT_TF>(pl, x, out, n_clips, n_samples, T);
}

template <int L>
static hipError_t dispatch_istft(const zafx_plan& pl, const float2* spec, float* y, int64_t n_clips, int T, int64_t out_len) {
    const bool one = pl.prm.spectrum == ZAFX_SPECTRUM_ONE_SIDED;
    if (pl.layout == ZAFX_LAYOUT_FT)
        return one ? run_istft<L, ZAFX_LAYOUT_FT, true>(pl, spec, y, n_clips, T, out_len) : run_istft<L, ZAFX_LAYOUT_FT, false>(pl, spec, y, n_clips, T, out_len);
    return one ? run_istft<L, ZAFX_LAYOUT_TF, true>(pl, spec, y, n_clips, T, out_len) : run_istft<L, ZAFX_LAYOUT_TF, false>(pl, spec, y, n_clips, T, out_len);
}

hipError_t launch_stft(const zafx_plan& pl, const float* x, float2* out, int64_t n_clips, int64_t n_samples, int T) {
    if (hipError_t e = hipSuccess; launch_spec2(pl, x, reinterpret_cast<float*>(out), n_clips, n_samples, T, e)) return e;   // W = 2048, |X| / |X|^2, reference layout
    switch (pl.log2nf) {
#define X(L) \
    case L: return dispatch_stft<L>(pl, x, out, n_clips, n_samples, T);
        ZAFX_STFT_SIZES(X)
#undef X
    }
    return hipErrorInvalidValue;
}

hipError_t launch_istft(const zafx_plan& pl, const float2* spec, float* y, int64_t n_clips, int T, int64_t out_len) {
    switch (pl.log2nf) {
#define X(L) \
    case L: return dispatch_istft<L>(pl, spec, y, n_clips, T, out_len);
        ZAFX_STFT_SIZES(X)
#undef X
    }
    return hipErrorInvalidValue;
}

}  // namespace zafx

ZAFX_PROF_EXPORT(zafx_debug_prof_istft, g_prof)
ZAFX_PROF_EXPORT(zafx_debug_prof_stft, g_prof_stft)
