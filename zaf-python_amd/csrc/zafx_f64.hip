// zafx_f64.hip -- float64 compute mode of every transform on the path ("bit-closer parity").
//
// The reference computes in float64 / complex128 (zaf.py:128, :139, :223).  The tuned kernels of
// zafx_stft.hip are float32; a plan created with zafx_params.precision = ZAFX_PRECISION_F64 runs the
// kernels below instead: same framing, padding, spectrum kinds and layouts, double arithmetic
// throughout, results within 1e-12 of the reference (tests/test_gpu_parity.py).  Two families:
//   * tiled kernels for the benchmark geometries (W = 2048 in the reference layout; CQT: fft_length 32768) -- k_stft_ft8_f64,
//     k_mdct_ft16_f64, k_imdct_ft16_f64, k_istft_ft8_f64 (round 5), k_mel_ft8_f64, k_cqt_ft_f64 (round 6): a frame per wavefront,
//     1024-point transforms in registers + the wave's own LDS, tiles that make every row piece a whole 128-byte line;
//   * for every other geometry one workgroup per frame, a radix-2 Stockham FFT of the packed half-length transform in LDS
//     (twiddles from a float64 table the host builds in long double), the ISTFT through a per-call scratch of time-domain
//     frames and a gather overlap-add in the reference's ascending frame order (zaf.py:226-233); windows that are not a
//     power of two as Bluestein convolutions.
#include <algorithm>
#include <cstdlib>

#include "zafx_fft.hpp"
#include "zafx_internal.hpp"
#include "zafx_mel64.hpp"
#include "zafx_cqt64.hpp"

namespace zafx {

namespace {

__device__ __forceinline__ double2 dadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 dsub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double2 dmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ double2 dmulc(double2 a, double2 b) { return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }
__device__ __forceinline__ double2 dconj(double2 a) { return make_double2(a.x, -a.y); }

// A 16-byte value to LDS.  ZAFX_F64_ST64 = 1 writes it as TWO 8-byte stores (volatile: never merged).  tools/exp_lds128.hip (round 6) times a
// ds_write_b128 at four times a ds_write_b64 where ds_read_b128 costs the same as ds_read_b64 -- but in the kernels the pairs are SLOWER
// (128 clips x 10 s: mdct 0.300 against 0.283 ms, mel 0.386 / 0.370, cqt 64 x 30 s 4.63 / 4.39): twice the DS instructions to issue.  Kept off.
#ifndef ZAFX_F64_ST64
#define ZAFX_F64_ST64 0
#endif
__device__ __forceinline__ void lds_st(double2* p, double2 v) {
#if ZAFX_F64_ST64
    typedef __attribute__((address_space(3))) volatile double lds_f64;
    lds_f64* q = (lds_f64*)p;
    q[0] = v.x;
    q[1] = v.y;
#else
    *p = v;
#endif
}

#ifndef ZAFX_F64_TILED
#define ZAFX_F64_TILED 1   // W = 2048, reference layout: k_stft_ft8_f64 / k_mdct_ft16_f64 instead of the frame-per-workgroup kernels
#endif
constexpr int kThreads = 256;       // workgroup size of the small frames
constexpr int kThreadsBig = 1024;   // ... of frames whose LDS image leaves room for one workgroup per CU only (see threads_for)

// In-place (ping-pong) forward FFT of N = 2^log2n points held in LDS; returns the buffer with the result.
// tw[m] = exp(-2 pi i m / N), m < N/2.
__device__ double2* fft_lds(double2* a, double2* b, int log2n, const double2* __restrict__ tw) {
    const int n = 1 << log2n;
    for (int s = 0; s < log2n; ++s) {
        const int ns = 1 << s;
        for (int j = threadIdx.x; j < n / 2; j += (int)blockDim.x) {
            const int k = j & (ns - 1);
            const double2 u = a[j], v = dmul(a[j + n / 2], tw[k << (log2n - 1 - s)]);
            const int i0 = ((j - k) << 1) + k;
            b[i0] = dadd(u, v);
            b[i0 + ns] = dsub(u, v);
        }
        __syncthreads();
        double2* t = a;
        a = b;
        b = t;
    }
    return a;
}

// zaf.py:112-139 for one frame per workgroup
__global__ __launch_bounds__(kThreadsBig) void k_stft_f64(
    const double* __restrict__ x, const double* __restrict__ win, const double2* __restrict__ tw, const double2* __restrict__ tws,
    double2* __restrict__ out, long long n_samples, int hop, int T, int TP, int log2n, int layout, int spec) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int N = 1 << log2n, W = 2 * N, rows = spec ? N + 1 : W;
    const bool one = spec != 0;
    double2* a = reinterpret_cast<double2*>(smem_raw);
    double2* b = a + N;
    const long long g = xcd_order((int)blockIdx.x, (int)gridDim.x);   // neighbouring frames to one XCD: the rows they share lines of meet in its L2
    const long long clip = g / T;
    const int t = (int)(g - clip * T);
    const double* xc = x + clip * n_samples;
    const long long s0 = (long long)t * hop - N;   // floor(W/2) = N samples of left padding
    for (int n = threadIdx.x; n < N; n += (int)blockDim.x) {
        const long long s = s0 + 2 * n;
        const double u = (s >= 0 && s < n_samples) ? xc[s] * win[2 * n] : 0.0;
        const double v = (s + 1 >= 0 && s + 1 < n_samples) ? xc[s + 1] * win[2 * n + 1] : 0.0;
        a[n] = make_double2(u, v);
    }
    __syncthreads();
    const double2* z = fft_lds(a, b, log2n, tw);
    // real split: X[k] = E + t_k O, X[N-k] = conj(E - t_k O)
    const long long stride = layout == ZAFX_LAYOUT_FT ? TP : 1;   // TP = row pitch (>= T)
    const long long base = layout == ZAFX_LAYOUT_FT ? clip * rows * TP + t : (clip * T + t) * rows;
    auto put = [&](long long row, double2 v) {   // complex, |X| or |X|^2 by spectrum kind
        if (spec >= ZAFX_SPECTRUM_MAGNITUDE) {
            const double pw = v.x * v.x + v.y * v.y;
            reinterpret_cast<double*>(out)[base + row * stride] = spec == ZAFX_SPECTRUM_MAGNITUDE ? sqrt(pw) : pw;
        } else {
            out[base + row * stride] = v;
        }
    };
    for (int k = threadIdx.x; k < N / 2; k += (int)blockDim.x) {
        if (k == 0) {
            const double2 z0 = z[0], zc = z[N / 2];
            put(0, make_double2(z0.x + z0.y, 0.0));
            put(N, make_double2(z0.x - z0.y, 0.0));
            put(N / 2, dconj(zc));
            if (!one) put(N + N / 2, zc);
        } else {
            const double2 zk = z[k], zn = z[N - k];
            const double2 e = make_double2(0.5 * (zk.x + zn.x), 0.5 * (zk.y - zn.y));
            const double2 d = make_double2(0.5 * (zk.x - zn.x), 0.5 * (zk.y + zn.y));
            const double2 to = dmul(tws[k], make_double2(d.y, -d.x));
            const double2 xk = dadd(e, to), xn = dconj(dsub(e, to));
            put(k, xk);
            put(N - k, xn);
            if (!one) {
                put(W - k, dconj(xk));
                put(N + k, dconj(xn));
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// k_stft_ft8_f64: the reference's own dtype (float64 in, complex128 out: zaf.py:128, :139) on the tiled structure of the float32
// headline kernel.  W = 2048, reference layout, complex kinds.  A tile is 8 frames of one clip -- 8 x 16 B = one 128-byte line of
// every output row --, a workgroup 8 waves, a wave one frame: 1024 packed points as 16 x 16 x 4 in registers (64 lanes x 16 double2)
// with two exchanges through the wave's own 17 KB of LDS, real split in place, then the whole workgroup writes the tile row by row,
// eight lanes to a line.  At 40 B per sample the kernel is bound by HBM, not by the 78 TF of float64 vector arithmetic: the
// one-frame-per-workgroup form (k_stft_f64: a barrier per radix-2 stage, 16-byte stores 6.9 KB apart) ran at 0.17 of it.
// Twiddles: exp(-2 pi i m / 1024) from the plan's half-circle table (m < 512; the other half by sign), no tables in LDS.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ double2 dmul_mi(double2 a) { return make_double2(a.y, -a.x); }   // a * (-i)
__device__ __forceinline__ void dft4d(double2& v0, double2& v1, double2& v2, double2& v3) {
    const double2 t0 = dadd(v0, v2), t1 = dsub(v0, v2), t2 = dadd(v1, v3), t3 = dmul_mi(dsub(v1, v3));
    v0 = dadd(t0, t2);
    v1 = dadd(t1, t3);
    v2 = dsub(t0, t2);
    v3 = dsub(t1, t3);
}
__device__ __forceinline__ void dft16d(double2* a) {   // natural order in and out, forward sign
    const double h = 0.70710678118654752440, c1 = 0.92387953251128675613, s1 = 0.38268343236508977173;
    double2 m[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        m[r][0] = a[r]; m[r][1] = a[r + 4]; m[r][2] = a[r + 8]; m[r][3] = a[r + 12];
        dft4d(m[r][0], m[r][1], m[r][2], m[r][3]);
    }
    m[1][1] = dmul(m[1][1], make_double2(c1, -s1));
    m[1][2] = dmul(m[1][2], make_double2(h, -h));
    m[1][3] = dmul(m[1][3], make_double2(s1, -c1));
    m[2][1] = dmul(m[2][1], make_double2(h, -h));
    m[2][2] = dmul_mi(m[2][2]);
    m[2][3] = dmul(m[2][3], make_double2(-h, -h));
    m[3][1] = dmul(m[3][1], make_double2(s1, -c1));
    m[3][2] = dmul(m[3][2], make_double2(-h, -h));
    m[3][3] = dmul(m[3][3], make_double2(-c1, s1));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        dft4d(m[0][q], m[1][q], m[2][q], m[3][q]);
        a[q] = m[0][q]; a[q + 4] = m[1][q]; a[q + 8] = m[2][q]; a[q + 12] = m[3][q];
    }
}
#ifndef ZAFX_F64_FPB
#define ZAFX_F64_FPB 8   // (4: 64-byte pieces and two workgroups per CU -- measured 1.30-1.32 ms against 1.23-1.29 for 256 clips x 10 s)
#endif
constexpr int kF64Frames = ZAFX_F64_FPB, kF64N = 1024, kF64Pitch = kF64N + kF64N / 16 + 1;
__device__ __forceinline__ int physd(int i) { return i + (i >> 4); }
__device__ __forceinline__ double2 root1024(const double2* __restrict__ tw, int m) {   // exp(-2 pi i m / 1024), m < 1024
    const double2 w = tw[m & 511];
    return m & 512 ? make_double2(-w.x, -w.y) : w;
}
// 1024-point forward transform of one wavefront: v[i] = z[lane + 64 i] in, natural order in `buf` out.  SPLIT_ROOTS: the real split's
// roots tk[i] = tws[lane + 64 i] are requested together with the last pass's (one L2 round trip instead of two).
template <bool SPLIT_ROOTS = false>
__device__ __forceinline__ void fft1024_f64(double2* v, double2* buf, int lane, const double2 (&w2)[16], const double2* __restrict__ tw, double2* tk = nullptr,
                                            const double2* __restrict__ tws = nullptr) {
    dft16d(v);
#pragma unroll
    for (int r = 0; r < 16; ++r) lds_st(&buf[physd(16 * lane + r)], v[r]);
    frame_sync<64>();
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = buf[physd(lane + 64 * i)];
    frame_sync<64>();
    {
        const int k = lane & 15;
#pragma unroll
        for (int r = 1; r < 16; ++r) v[r] = dmul(v[r], w2[r]);   // exp(-2 pi i r k / 256)
        dft16d(v);
        const int base = ((lane >> 4) << 8) + k;
#pragma unroll
        for (int r = 0; r < 16; ++r) lds_st(&buf[physd(base + 16 * r)], v[r]);
    }
    frame_sync<64>();
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = buf[physd(lane + 64 * i)];
    frame_sync<64>();
    if constexpr (SPLIT_ROOTS) {
#pragma unroll
        for (int i = 0; i < 8; ++i) tk[i] = tws[lane + 64 * i];
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int k = lane + 64 * b;
        double2 a0 = v[b], a1 = dmul(v[b + 4], root1024(tw, k)), a2 = dmul(v[b + 8], root1024(tw, 2 * k)), a3 = dmul(v[b + 12], root1024(tw, 3 * k));
        dft4d(a0, a1, a2, a3);
        lds_st(&buf[physd(k)], a0);
        lds_st(&buf[physd(k + 256)], a1);
        lds_st(&buf[physd(k + 512)], a2);
        lds_st(&buf[physd(k + 768)], a3);
    }
    frame_sync<64>();
}

template <bool ONE>
__global__ __launch_bounds__(kF64Frames * 64) void k_stft_ft8_f64(const double* __restrict__ x, const double* __restrict__ win, const double2* __restrict__ tw,
                                                                   const double2* __restrict__ tws, double2* __restrict__ out, long long n_samples, int hop, int T,
                                                                   int TP, int tiles, int total_tiles) {
    constexpr int N = kF64N, W = 2 * N, FPB = kF64Frames, PITCH = kF64Pitch, ROWS = ONE ? N + 1 : W;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double2* frames = reinterpret_cast<double2*>(smem_raw);
    double* nyq = reinterpret_cast<double*>(frames + FPB * PITCH);   // X[N] of every frame (real)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double2* buf = frames + wave * PITCH;
    const bool xcd = gridDim.x % 8 == 0;
    // samples of the wave's frame of tile `tlv` (raw: the window is applied where they are consumed), requested one tile ahead -- before the
    // store phase of the tile in front, whose LDS reads and 16-byte stores they fly under
    double2 v[16];
    auto request = [&](int tlv) {
        if (tlv >= total_tiles) return;
        const int tl = xcd ? xcd_order(tlv, total_tiles) : tlv;
        const int clip = tl / tiles, t = (tl % tiles) * FPB + wave;
        const double* xc = x + (long long)clip * n_samples;
        const long long s0 = (long long)t * hop - N;   // floor(W / 2) samples of left padding (zaf.py:99, :112)
        if (t < T && s0 >= 0 && s0 + W <= n_samples && ((s0 | n_samples) & 1) == 0) {   // (uniform) interior frame, 16-byte loads
            const double2* xp = reinterpret_cast<const double2*>(xc + s0);
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = xp[lane + 64 * i];
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const long long s = s0 + 2 * (lane + 64 * i);
                v[i].x = (t < T && s >= 0 && s < n_samples) ? xc[s] : 0.0;
                v[i].y = (t < T && s + 1 >= 0 && s + 1 < n_samples) ? xc[s + 1] : 0.0;
            }
        }
    };
    request(blockIdx.x);
    for (int tlv = blockIdx.x; tlv < total_tiles; tlv += gridDim.x) {
        const int tl = xcd ? xcd_order(tlv, total_tiles) : tlv;
        const int clip = tl / tiles, t0 = (tl % tiles) * FPB;
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));   // (opaque per tile: window and twiddle values are re-read from L1, not hoisted out of the loop and spilled)
        double2 w2[16];
#pragma unroll
        for (int r = 1; r < 16; ++r) w2[r] = root1024(tw, 4 * r * (lane_o & 15));   // exp(-2 pi i r k / 256)
        {
            const double2* wp = reinterpret_cast<const double2*>(win);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const double2 w = wp[lane_o + 64 * i];
                v[i] = make_double2(v[i].x * w.x, v[i].y * w.y);
            }
        }
        fft1024_f64(v, buf, lane_o, w2, tw);
        // real split in place: X[k] = E + t_k O, X[N-k] = conj(E - t_k O) (as k_stft_f64)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = lane + 64 * i;
            if (k == 0) {
                const double2 z0 = buf[0], zc = buf[physd(N / 2)];
                buf[0] = make_double2(z0.x + z0.y, 0.0);
                nyq[wave] = z0.x - z0.y;
                lds_st(&buf[physd(N / 2)], dconj(zc));
            } else {
                const double2 zk = buf[physd(k)], zn = buf[physd(N - k)];
                const double2 e = make_double2(0.5 * (zk.x + zn.x), 0.5 * (zk.y - zn.y));
                const double2 d = make_double2(0.5 * (zk.x - zn.x), 0.5 * (zk.y + zn.y));
                const double2 to = dmul(tws[k], make_double2(d.y, -d.x));
                lds_st(&buf[physd(k)], dadd(e, to));
                lds_st(&buf[physd(N - k)], dconj(dsub(e, to)));
            }
        }
        lds_barrier();
        request(tlv + gridDim.x);
        // the tile's rows: eight lanes (frames) to a 128-byte line, 64 rows per instruction of the workgroup
        {
            const int f = tid & (FPB - 1), g = tid / FPB;
            const double2* fb = frames + f * PITCH;
            double2* o = out + (long long)clip * ROWS * TP + t0 + f;
            if (t0 + f < T) {
#pragma unroll 4
                for (int r = g; r < ROWS; r += 64) {   // (64 = threads / frames: rows per instruction of the workgroup)
                    double2 val;
                    if (r < N) val = fb[physd(r)];
                    else if (r == N) val = make_double2(nyq[f], 0.0);
                    else val = dconj(fb[physd(W - r)]);
                    typedef double f64x2 __attribute__((ext_vector_type(2)));
                    f64x2 q;
                    q.x = val.x;
                    q.y = val.y;
                    __builtin_nontemporal_store(q, reinterpret_cast<f64x2*>(o + (long long)r * TP));   // one 16-byte streaming store
                }
            }
        }
        lds_barrier();
    }
}

// ---------------------------------------------------------------------------------
// k_mdct_ft16_f64: the forward MDCT in float64 (zaf.py:1029-1073) on the tiled structure, W = 2048, reference layout.  A tile is 16 frames of
// one clip -- 16 x 8 B = one 128-byte line of every coefficient row --, a workgroup 16 waves, a wave one frame: window + TDAC fold + pre-twiddle
// straight out of global memory into registers (c[m] = (v[2m] + i v[M-1-2m]) g_m), 512 points as 8 x 8 x 8 with two exchanges through the wave's
// 9 KB of LDS, post-twiddle in place (the pair k, 511 - k trades its imaginary parts: out[2k] = Re y_k, out[2k+1] = -Im y_{511-k}), then the
// workgroup writes the tile row by row, sixteen lanes to a line.  16 bytes per sample: bound by HBM (k_mdct_f64, a frame per workgroup, 0.11 of it).
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void dft8d(double2* a) {
    const double h = 0.70710678118654752440;
    double2 e0 = a[0], e1 = a[2], e2 = a[4], e3 = a[6], o0 = a[1], o1 = a[3], o2 = a[5], o3 = a[7];
    dft4d(e0, e1, e2, e3);
    dft4d(o0, o1, o2, o3);
    o1 = dmul(o1, make_double2(h, -h));
    o2 = dmul_mi(o2);
    o3 = dmul(o3, make_double2(-h, -h));
    a[0] = dadd(e0, o0); a[4] = dsub(e0, o0);
    a[1] = dadd(e1, o1); a[5] = dsub(e1, o1);
    a[2] = dadd(e2, o2); a[6] = dsub(e2, o2);
    a[3] = dadd(e3, o3); a[7] = dsub(e3, o3);
}
constexpr int kMd64Frames = 16, kMd64NF = 512, kMd64Pitch = kMd64NF + kMd64NF / 8 + 1;
__device__ __forceinline__ int phys8(int i) { return i + (i >> 3); }
__device__ __forceinline__ double2 root512(const double2* __restrict__ tw, int m) {   // exp(-2 pi i m / 512), m < 512
    const double2 w = tw[m & 255];
    return m & 256 ? make_double2(-w.x, -w.y) : w;
}

__global__ __launch_bounds__(kMd64Frames * 64) void k_mdct_ft16_f64(const double* __restrict__ x, const double* __restrict__ win, const double2* __restrict__ tw,
                                                                     const double2* __restrict__ g, double* __restrict__ out, long long n_samples, int T, int TP,
                                                                     int tiles, int total_tiles) {
    constexpr int NF = kMd64NF, M = 2 * NF, FPB = kMd64Frames, PITCH = kMd64Pitch;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double2* frames = reinterpret_cast<double2*>(smem_raw);
    const int tid = threadIdx.x, wave = tid >> 6;
    double2* buf = frames + wave * PITCH;
    const bool xcd = gridDim.x % 8 == 0;
    for (int tlv = blockIdx.x; tlv < total_tiles; tlv += gridDim.x) {
        int lane = tid & 63;
        asm volatile("" : "+v"(lane));   // (opaque per tile: the lane's window, twiddle and table values are re-read, not hoisted out of the loop and spilled)
        const int tl = xcd ? xcd_order(tlv, total_tiles) : tlv;
        const int clip = tl / tiles, t0 = (tl % tiles) * FPB, t = t0 + wave;
        const double* xc = x + (long long)clip * n_samples;
        const long long s0 = (long long)t * M - M;   // left pad = M (zaf.py:1036-1041)
        const bool inside = t < T && s0 >= 0 && s0 + 2 * M <= n_samples && ((s0 | n_samples) & 1) == 0;   // (uniform) 16-byte loads
        // The frame's two halves go through the wave's buffer one after the other, windowed, as coalesced 16-byte pieces; the fold reads its
        // partners (a sample and its mirror image) from there.  (Folded straight out of global memory -- 32 + 32 strided 8-byte loads per lane in
        // flight -- the kernel needed more than its 128 registers: 676 bytes of scratch, 1.63 ms for 256 clips.)
        double* ub = reinterpret_cast<double*>(buf);
        auto stage = [&](int half) {   // windowed samples [1024 half, 1024 half + 1024) of the frame -> ub[0 .. 1023]
            const double2* wp = reinterpret_cast<const double2*>(win) + NF * half;
            if (inside) {
                const double2* xp = reinterpret_cast<const double2*>(xc + s0) + NF * half;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int q = lane + 64 * j;
                    const double2 a = xp[q], w = wp[q];
                    reinterpret_cast<double2*>(ub)[q] = make_double2(a.x * w.x, a.y * w.y);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int q = lane + 64 * j;
                    const long long sa = s0 + M * half + 2 * q;
                    const double2 w = wp[q];
                    const double a = (t < T && sa >= 0 && sa < n_samples) ? xc[sa] : 0.0, b = (t < T && sa + 1 >= 0 && sa + 1 < n_samples) ? xc[sa + 1] : 0.0;
                    reinterpret_cast<double2*>(ub)[q] = make_double2(a * w.x, b * w.y);
                }
            }
        };
        double va[8], vb[8];
        stage(1);   // u[2 NF + n]: the parts of the fold below NF, -u[3 NF - 1 - i] - u[3 NF + i]
        frame_sync<64>();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = lane + 64 * i, a = 2 * m, b = M - 1 - 2 * m;
            if (i < 4) va[i] = -ub[NF - 1 - a] - ub[NF + a];
            else vb[i] = -ub[NF - 1 - b] - ub[NF + b];
        }
        frame_sync<64>();
        stage(0);   // u[n]: the parts from NF up, u[i - NF] - u[3 NF - 1 - i]
        frame_sync<64>();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = lane + 64 * i, a = 2 * m, b = M - 1 - 2 * m;
            if (i < 4) vb[i] = ub[b - NF] - ub[3 * NF - 1 - b];
            else va[i] = ub[a - NF] - ub[3 * NF - 1 - a];
        }
        frame_sync<64>();
        double2 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = dmul(make_double2(va[i], vb[i]), g[lane + 64 * i]);   // c[m] = (v[2m] + i v[M-1-2m]) g_m (zaf.py:1047-1056 as k_mdct_f64 writes it)
        // 512 points: radix 8 three times
        dft8d(v);
#pragma unroll
        for (int r = 0; r < 8; ++r) lds_st(&buf[phys8(8 * lane + r)], v[r]);
        frame_sync<64>();
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = buf[phys8(lane + 64 * i)];
        frame_sync<64>();
        {
            const int k = lane & 7;
#pragma unroll
            for (int r = 1; r < 8; ++r) v[r] = dmul(v[r], root512(tw, 8 * r * k));   // exp(-2 pi i r k / 64)
            dft8d(v);
            const int base = ((lane >> 3) << 6) + k;
#pragma unroll
            for (int r = 0; r < 8; ++r) lds_st(&buf[phys8(base + 8 * r)], v[r]);
        }
        frame_sync<64>();
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = buf[phys8(lane + 64 * i)];
        frame_sync<64>();
#pragma unroll
        for (int r = 1; r < 8; ++r) v[r] = dmul(v[r], root512(tw, r * lane));           // exp(-2 pi i r k / 512), k = lane
        dft8d(v);
        // Z[lane + 64 r] = v[r]; post-twiddle y_k = Z[k] g_k and the pair's trade, in place: lane (k < 256: r < 4) needs Z[511 - k], held by lane 63 - lane
        // in register 7 - r: through LDS
#pragma unroll
        for (int r = 0; r < 8; ++r) lds_st(&buf[phys8(lane + 64 * r)], dmul(v[r], g[lane + 64 * r]));   // y_k
        frame_sync<64>();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = lane + 64 * r, kp = NF - 1 - k;
            const double2 yk = buf[phys8(k)], yp = buf[phys8(kp)];
            lds_st(&buf[phys8(k)], make_double2(yk.x, -yp.y));    // out[2k], out[2k+1] = out[M-1-2k']
            lds_st(&buf[phys8(kp)], make_double2(yp.x, -yk.y));   // out[2k'], out[2k'+1] = out[M-1-2k]
        }
        lds_barrier();
        {
            const int f = tid & 15, gq = tid >> 4;
            const double* fb = reinterpret_cast<const double*>(frames + f * PITCH);
            double* o = out + (long long)clip * M * TP + t0 + f;
            if (t0 + f < T) {
#pragma unroll 4
                for (int r = gq; r < M; r += 64) __builtin_nontemporal_store(fb[2 * phys8(r >> 1) + (r & 1)], o + (long long)r * TP);
            }
        }
        lds_barrier();
    }
}

// ---------------------------------------------------------------------------------
// k_imdct_ft16_f64: the inverse MDCT in float64 (zaf.py:1125-1182) on the tiled structure, W = 2048, reference layout.  A workgroup walks the
// 16-frame tiles of a clip segment in order: coefficient rows gathered as 128-byte lines (sixteen frames x 8 B) straight into the pre-twiddled
// DCT-IV input c[m] = (X[2m] + i X[M-1-2m]) g_m, a frame per wavefront (512 points as 8 x 8 x 8, post-twiddle and the pair's trade in place as in
// k_mdct_ft16_f64), and the time-domain frames are never formed: an output sample is the windowed unfold of two neighbouring frames' DCT-IV
// outputs, read from LDS where it is stored (coalesced 512-byte runs per wave).  Seventeen frame buffers in rotation: the last frame of a tile
// stays where it is as the next tile's left neighbour.  (k_imdct_frames_f64 + k_ola_f64: frames through a scratch array, a frame per workgroup.)
// ---------------------------------------------------------------------------------
constexpr int kImd64Slots = kMd64Frames + 1;
__global__ __launch_bounds__(kMd64Frames * 64) void k_imdct_ft16_f64(const double* __restrict__ coefs, const double* __restrict__ win, const double2* __restrict__ tw,
                                                                      const double2* __restrict__ g, double* __restrict__ y, int T, int TP, long long out_len,
                                                                      int tiles, int segs, int seg_tiles, int units) {
    constexpr int NF = kMd64NF, M = 2 * NF, FPB = kMd64Frames, PITCH = kMd64Pitch, NS = kImd64Slots;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double2* frames = reinterpret_cast<double2*>(smem_raw);
    const int tid = threadIdx.x, wave = tid >> 6;
    const double gain = 2.0 / (double)M;
    // DCT-IV of the frame whose pre-twiddled input is in `buf` (natural order); leaves u as elements (u[2k], u[2k+1])
    auto dct4_wave = [&](double2* buf, int lane) {
        double2 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = buf[phys8(lane + 64 * i)];
        frame_sync<64>();
        dft8d(v);
#pragma unroll
        for (int r = 0; r < 8; ++r) lds_st(&buf[phys8(8 * lane + r)], v[r]);
        frame_sync<64>();
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = buf[phys8(lane + 64 * i)];
        frame_sync<64>();
        {
            const int k = lane & 7;
#pragma unroll
            for (int r = 1; r < 8; ++r) v[r] = dmul(v[r], root512(tw, 8 * r * k));
            dft8d(v);
            const int base = ((lane >> 3) << 6) + k;
#pragma unroll
            for (int r = 0; r < 8; ++r) lds_st(&buf[phys8(base + 8 * r)], v[r]);
        }
        frame_sync<64>();
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = buf[phys8(lane + 64 * i)];
        frame_sync<64>();
#pragma unroll
        for (int r = 1; r < 8; ++r) v[r] = dmul(v[r], root512(tw, r * lane));
        dft8d(v);
#pragma unroll
        for (int r = 0; r < 8; ++r) lds_st(&buf[phys8(lane + 64 * r)], dmul(v[r], g[lane + 64 * r]));   // y_k = Z[k] g_k
        frame_sync<64>();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = lane + 64 * r, kp = NF - 1 - k;
            const double2 yk = buf[phys8(k)], yp = buf[phys8(kp)];
            lds_st(&buf[phys8(k)], make_double2(yk.x, -yp.y));
            lds_st(&buf[phys8(kp)], make_double2(yp.x, -yk.y));
        }
    };
    for (int unit = blockIdx.x; unit < units; unit += gridDim.x) {
        const int clip = unit / segs, seg = unit % segs;
        const int tile_a = seg * seg_tiles, tile_b = min(tile_a + seg_tiles, tiles);
        const double* cp = coefs + (long long)clip * M * TP;
        double* yc = y + (long long)clip * out_len;
        int rot = 0;
        if (tile_a > 0) {   // the segment's left neighbour: frame 16 tile_a - 1, by the last wave alone
            if (wave == FPB - 1) {
                int lane = tid & 63;
                asm volatile("" : "+v"(lane));
                double2* buf = frames + ((rot + FPB) % NS) * PITCH;
                const int t = tile_a * FPB - 1;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int m = lane + 64 * i;
                    lds_st(&buf[phys8(m)], dmul(make_double2(cp[(long long)(2 * m) * TP + t], cp[(long long)(M - 1 - 2 * m) * TP + t]), g[m]));
                }
                frame_sync<64>();
                dct4_wave(buf, lane);
            }
            lds_barrier();
        }
        for (int tile = tile_a; tile < tile_b; ++tile) {
            const int t0 = tile * FPB;
            int to = tid;
            asm volatile("" : "+v"(to));   // (opaque per tile: table values are re-read, not hoisted and spilled)
            {   // rows of the tile: thread = frame (to & 15) x 64 values of m, the two rows of each as 8-byte pieces of a 128-byte line
                const int f = to & 15, mq = to >> 4, t = t0 + f;
                double2* buf = frames + ((rot + f) % NS) * PITCH;
                if (t < T) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int m = mq + 64 * i;
                        lds_st(&buf[phys8(m)], dmul(make_double2(cp[(long long)(2 * m) * TP + t], cp[(long long)(M - 1 - 2 * m) * TP + t]), g[m]));
                    }
                }
            }
            lds_barrier();
            if (t0 + wave < T) dct4_wave(frames + ((rot + wave) % NS) * PITCH, to & 63);
            lds_barrier();
            // output blocks b = t0 .. t0 + 15 (and b = T behind the clip's last frame): sample s = b M + i of the padded signal is the second half of
            // frame b - 1 plus the first half of frame b (zaf.py:1172-1179), output index s - M (the trim of :1182)
            const int b_hi = tile + 1 >= tiles ? T : t0 + FPB - 1;
            for (int b = max(t0, 1); b <= b_hi; ++b) {
                const int i = to;   // (M = 1024 = the workgroup)
                const long long o = (long long)(b - 1) * M + i;
                double acc = 0.0;
                {   // frame b - 1, n = M + i
                    const int fprev = b - 1 - t0;   // -1: the tile before (slot rot + 16)
                    const double* u = reinterpret_cast<const double*>(frames + ((rot + (fprev < 0 ? FPB : fprev)) % NS) * PITCH);
                    const int j = i < NF ? NF - 1 - i : i - NF;
                    acc = -u[2 * phys8(j >> 1) + (j & 1)] * win[M + i] * gain;
                }
                if (b < T && b <= t0 + FPB - 1) {   // frame b, n = i
                    const double* u = reinterpret_cast<const double*>(frames + ((rot + (b - t0)) % NS) * PITCH);
                    const int j = i < NF ? NF + i : 3 * NF - 1 - i;
                    const double uv = u[2 * phys8(j >> 1) + (j & 1)];
                    acc += (i < NF ? uv : -uv) * win[i] * gain;
                }
                if (o < out_len) __builtin_nontemporal_store(acc, yc + o);
            }
            rot = (rot + FPB) % NS;
            lds_barrier();
        }
    }
}

// ---------------------------------------------------------------------------------
// k_istft_ft8_f64: the inverse STFT in float64 (zaf.py:214-241) on the tiled structure, W = 2048, reference layout, hop >= W / 2 (a sample under at
// most two frames).  A workgroup of 8 waves walks the 8-frame tiles of a clip segment in order: spectrum rows k, W-k, N-k, N+k gathered as 128-byte
// lines (eight frames x 16 B) straight into the Hermitian fold (the packed half-length spectrum with re / im swapped, so that the FORWARD transform
// inverts: k_ifft_frames_f64), a frame per wavefront (fft1024_f64), and the overlap-add reads the frames where they lie -- nine buffers in rotation,
// the last frame of a tile stays as the next tile's left neighbour --, adds in the reference's ascending frame order and stores coalesced runs.
// ---------------------------------------------------------------------------------
constexpr int kIst64Slots = kF64Frames + 1;
template <bool ONE>
__global__ __launch_bounds__(kF64Frames * 64) void k_istft_ft8_f64(const double2* __restrict__ spec, const double2* __restrict__ tw, const double2* __restrict__ tws,
                                                                    double* __restrict__ y, int T, int TP, int hop, long long out_len, double scale, int tiles,
                                                                    int segs, int seg_tiles, int units) {
    constexpr int N = kF64N, W = 2 * N, FPB = kF64Frames, PITCH = kF64Pitch, NS = kIst64Slots, ROWS = ONE ? N + 1 : W;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double2* frames = reinterpret_cast<double2*>(smem_raw);
    const int tid = threadIdx.x, wave = tid >> 6;
    // Hermitian fold of the pair (k, N - k) of frame t (column t of `sp`) into buf (k_ifft_frames_f64's formulas)
    auto fold = [&](const double2* sp, int t, int k, double2* buf) {
        auto row = [&](int r) { return sp[(long long)r * TP + t]; };
        if (k == 0) {
            const double a0 = 2.0 * row(0).x, an = 2.0 * row(N).x;
            buf[0] = make_double2(a0 - an, a0 + an);
            const double2 xc = row(N / 2);
            const double2 xd = ONE ? dconj(xc) : row(N + N / 2);
            const double2 h = make_double2(xc.x + xd.x, xc.y - xd.y);
            lds_st(&buf[physd(N / 2)], make_double2(-2.0 * h.y, 2.0 * h.x));
        } else {
            const double2 xk = row(k), xnk = row(N - k);
            const double2 xwk = ONE ? dconj(xk) : row(W - k);
            const double2 xnpk = ONE ? dconj(xnk) : row(N + k);
            const double2 ak = make_double2(xk.x + xwk.x, xk.y - xwk.y);
            const double2 an = make_double2(xnk.x + xnpk.x, xnk.y - xnpk.y);
            const double2 e = make_double2(ak.x + an.x, ak.y - an.y);
            const double2 d = make_double2(ak.x - an.x, ak.y + an.y);
            const double2 o = dmulc(d, tws[k]);
            const double2 zk = make_double2(e.x - o.y, e.y + o.x), zn = make_double2(e.x + o.y, -e.y + o.x);
            lds_st(&buf[physd(k)], make_double2(zk.y, zk.x));
            lds_st(&buf[physd(N - k)], make_double2(zn.y, zn.x));
        }
    };
    auto transform = [&](double2* buf, int lane) {
        double2 v[16], w2[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = buf[physd(lane + 64 * i)];
        frame_sync<64>();
#pragma unroll
        for (int r = 1; r < 16; ++r) w2[r] = root1024(tw, 4 * r * (lane & 15));
        fft1024_f64(v, buf, lane, w2, tw);
    };
    for (int unit = blockIdx.x; unit < units; unit += gridDim.x) {
        const int clip = unit / segs, seg = unit % segs;
        const int tile_a = seg * seg_tiles, tile_b = min(tile_a + seg_tiles, tiles);
        const double2* sp = spec + (long long)clip * ROWS * TP;
        double* yc = y + (long long)clip * out_len;
        int rot = 0;
        if (tile_a > 0) {   // the segment's left neighbour: frame 8 tile_a - 1, by the last wave alone
            if (wave == FPB - 1) {
                int lane = tid & 63;
                asm volatile("" : "+v"(lane));
                double2* buf = frames + ((rot + FPB) % NS) * PITCH;
#pragma unroll
                for (int i = 0; i < 8; ++i) fold(sp, tile_a * FPB - 1, lane + 64 * i, buf);
                frame_sync<64>();
                transform(buf, lane);
            }
            lds_barrier();
        }
        for (int tile = tile_a; tile < tile_b; ++tile) {
            const int t0 = tile * FPB;
            int to = tid;
            asm volatile("" : "+v"(to));   // (opaque per tile: table values are re-read, not hoisted and spilled)
            {
                const int f = to & 7, kq = to >> 3, t = t0 + f;
                double2* buf = frames + ((rot + f) % NS) * PITCH;
                if (t < T) {
#pragma unroll 2   // (4 or 8 pairs in flight: the same 0.65 ms)
                    for (int i = 0; i < 8; ++i) fold(sp, t, kq + 64 * i, buf);
                }
            }
            lds_barrier();
            if (t0 + wave < T) transform(frames + ((rot + wave) % NS) * PITCH, to & 63);
            lds_barrier();
            // samples s of the padded signal that no later frame reaches: [t0 hop, (t0 + 8) hop), and everything up to the end behind the clip's last frame
            const long long s_lo = (long long)t0 * hop;
            const long long s_hi = tile + 1 >= tiles ? (long long)(T - 1) * hop + W : (long long)(t0 + FPB) * hop;
            for (long long s = s_lo + to; s < s_hi; s += FPB * 64) {
                const long long o = s - (W - hop);
                if (o < 0 || o >= out_len) continue;
                const int j = (int)min((long long)(T - 1), s / hop);   // the last frame that reaches s; the one before it when s - (j - 1) hop < W
                double acc = 0.0;
#pragma unroll
                for (int d = 1; d >= 0; --d) {   // ascending frame order (zaf.py:226-233)
                    const int jj = j - d;
                    const long long n = s - (long long)jj * hop;
                    if (jj < 0 || n < 0 || n >= W) continue;
                    const int fl = jj - t0;   // -1: the tile before
                    const double* fr = reinterpret_cast<const double*>(frames + ((rot + (fl < 0 ? FPB : fl)) % NS) * PITCH);
                    acc += fr[2 * physd((int)n >> 1) + (((int)n & 1) ^ 1)];   // components come out swapped
                }
                __builtin_nontemporal_store(acc * scale, yc + o);
            }
            rot = (rot + FPB) % NS;
            lds_barrier();
        }
    }
}

// real(ifft(X)) of one frame per workgroup (zaf.py:223), W samples into the scratch, unscaled by 2 W
__global__ __launch_bounds__(kThreadsBig) void k_ifft_frames_f64(
    const double2* __restrict__ spec, const double2* __restrict__ tw, const double2* __restrict__ tws, double* __restrict__ frames,
    int T, int TP, int log2n, int layout, int one) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int N = 1 << log2n, W = 2 * N, rows = one ? N + 1 : W;
    double2* a = reinterpret_cast<double2*>(smem_raw);
    double2* b = a + N;
    const long long g = xcd_order((int)blockIdx.x, (int)gridDim.x);   // neighbouring frames to one XCD: the rows they share lines of meet in its L2
    const long long clip = g / T;
    const int t = (int)(g - clip * T);
    const long long stride = layout == ZAFX_LAYOUT_FT ? TP : 1;
    const double2* sp = layout == ZAFX_LAYOUT_FT ? spec + clip * rows * TP + t : spec + (clip * T + t) * rows;
    // packed half-length spectrum of the Hermitian part of X, re/im swapped so that a FORWARD transform inverts
    for (int k = threadIdx.x; k < N / 2; k += (int)blockDim.x) {
        if (k == 0) {
            const double a0 = 2.0 * sp[0].x, an = 2.0 * sp[(long long)N * stride].x;
            a[0] = make_double2(a0 - an, a0 + an);
            const double2 xc = sp[(long long)(N / 2) * stride];
            const double2 xd = one ? dconj(xc) : sp[(long long)(N + N / 2) * stride];
            const double2 h = make_double2(xc.x + xd.x, xc.y - xd.y);
            a[N / 2] = make_double2(-2.0 * h.y, 2.0 * h.x);
        } else {
            const double2 xk = sp[(long long)k * stride], xnk = sp[(long long)(N - k) * stride];
            const double2 xwk = one ? dconj(xk) : sp[(long long)(W - k) * stride];
            const double2 xnpk = one ? dconj(xnk) : sp[(long long)(N + k) * stride];
            const double2 ak = make_double2(xk.x + xwk.x, xk.y - xwk.y);       // X[k] + conj X[W-k]
            const double2 an = make_double2(xnk.x + xnpk.x, xnk.y - xnpk.y);   // X[N-k] + conj X[N+k]
            const double2 e = make_double2(ak.x + an.x, ak.y - an.y);
            const double2 d = make_double2(ak.x - an.x, ak.y + an.y);
            const double2 o = dmulc(d, tws[k]);
            const double2 zk = make_double2(e.x - o.y, e.y + o.x), zn = make_double2(e.x + o.y, -e.y + o.x);
            a[k] = make_double2(zk.y, zk.x);
            a[N - k] = make_double2(zn.y, zn.x);
        }
    }
    __syncthreads();
    const double2* z = fft_lds(a, b, log2n, tw);
    double* fr = frames + g * W;
    for (int n = threadIdx.x; n < N; n += (int)blockDim.x) {   // components come out swapped
        fr[2 * n] = z[n].y;
        fr[2 * n + 1] = z[n].x;
    }
}

// overlap-add in ascending frame order (zaf.py:226-233), trim (:236-238), gain (:241)
__global__ __launch_bounds__(kThreadsBig) void k_ola_f64(const double* __restrict__ frames, double* __restrict__ y, int T, int W, int hop,
                                                       long long out_len, long long total, double scale) {
    for (long long i = (long long)blockIdx.x * (int)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * (int)blockDim.x) {
        const long long clip = i / out_len, o = i - clip * out_len;
        const long long s = o + (W - hop);
        const long long j_hi = std::min<long long>(T - 1, s / hop);
        const long long j_lo = s >= W ? (s - W) / hop + 1 : 0;
        double acc = 0.0;
        for (long long j = j_lo; j <= j_hi; ++j) acc += frames[(clip * T + j) * W + (s - j * hop)];
        y[i] = acc * scale;
    }
}

// ---- MDCT / IMDCT (zaf.py:984-1184) through the W/4-point DCT-IV of zafx_mdct.hip, in float64 -------------------
// c[m] = (v[2m] + i v[M-1-2m]) g_m, Y = FFT_{M/2}(c), y_k = Y[k] g_k, out[2k] = Re y_k, out[M-1-2k] = -Im y_k,
// with g_m = exp(-i pi (8m+1) / (8M)); `v` (M doubles) is consumed, `u` (M doubles) receives the result.
__device__ void dct4_lds(const double* v, double* u, double2* a, double2* b, int log2nf, const double2* __restrict__ tw,
                         const double2* __restrict__ g) {
    const int NF = 1 << log2nf, M = 2 * NF;
    for (int m = threadIdx.x; m < NF; m += (int)blockDim.x) a[m] = dmul(make_double2(v[2 * m], v[M - 1 - 2 * m]), g[m]);
    __syncthreads();
    const double2* z = fft_lds(a, b, log2nf, tw);
    for (int k = threadIdx.x; k < NF; k += (int)blockDim.x) {
        const double2 y = dmul(z[k], g[k]);
        u[2 * k] = y.x;
        u[M - 1 - 2 * k] = -y.y;
    }
    __syncthreads();
}

// one frame per workgroup: window, TDAC fold (W -> M reals), DCT-IV
__global__ __launch_bounds__(kThreadsBig) void k_mdct_f64(
    const double* __restrict__ x, const double* __restrict__ win, const double2* __restrict__ tw, const double2* __restrict__ g,
    double* __restrict__ out, long long n_samples, int T, int TP, int log2nf, int layout) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int NF = 1 << log2nf, M = 2 * NF, W = 4 * NF;
    double* u = reinterpret_cast<double*>(smem_raw);        // W: the windowed frame, then (first M) the coefficients
    double* v = u + W;                                      // M: the folded frame
    double2* a = reinterpret_cast<double2*>(v + M);
    double2* b = a + NF;
    const long long gi = blockIdx.x;
    const long long clip = gi / T;
    const int t = (int)(gi - clip * T);
    const double* xc = x + clip * n_samples;
    const long long s0 = (long long)t * M - M;              // left pad = M (zaf.py:1036-1064)
    for (int n = threadIdx.x; n < W; n += (int)blockDim.x) {
        const long long s = s0 + n;
        u[n] = (s >= 0 && s < n_samples) ? xc[s] * win[n] : 0.0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < M; i += (int)blockDim.x) {        // v = (-c_r - d, a - b_r), frame = (a, b, c, d) quarters of NF
        v[i] = i < NF ? -u[3 * NF - 1 - i] - u[3 * NF + i] : u[i - NF] - u[3 * NF - 1 - i];
    }
    __syncthreads();
    dct4_lds(v, u, a, b, log2nf, tw, g);
    const long long stride = layout == ZAFX_LAYOUT_FT ? TP : 1;
    const long long base = layout == ZAFX_LAYOUT_FT ? clip * M * TP + t : (clip * T + t) * M;
    for (int f = threadIdx.x; f < M; f += (int)blockDim.x) out[base + f * stride] = u[f];
}

// one frame per workgroup: DCT-IV of the coefficients, unfold (u2, -u2_r, -u1_r, -u1), window, 2/M (zaf.py:1138-1169)
__global__ __launch_bounds__(kThreadsBig) void k_imdct_frames_f64(
    const double* __restrict__ coefs, const double* __restrict__ win, const double2* __restrict__ tw, const double2* __restrict__ g,
    double* __restrict__ frames, int T, int TP, int log2nf, int layout) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int NF = 1 << log2nf, M = 2 * NF, W = 4 * NF;
    double* u = reinterpret_cast<double*>(smem_raw);        // M
    double* v = u + M;                                      // M
    double2* a = reinterpret_cast<double2*>(v + M);
    double2* b = a + NF;
    const long long gi = blockIdx.x;
    const long long clip = gi / T;
    const int t = (int)(gi - clip * T);
    const long long stride = layout == ZAFX_LAYOUT_FT ? TP : 1;
    const double* cp = layout == ZAFX_LAYOUT_FT ? coefs + clip * M * TP + t : coefs + (clip * T + t) * M;
    for (int f = threadIdx.x; f < M; f += (int)blockDim.x) v[f] = cp[f * stride];
    __syncthreads();
    dct4_lds(v, u, a, b, log2nf, tw, g);
    double* fr = frames + gi * W;
    const double gain = 2.0 / (double)M;
    for (int n = threadIdx.x; n < W; n += (int)blockDim.x) {
        const double val = n < NF ? u[NF + n] : n < 3 * NF ? -u[3 * NF - 1 - n] : -u[n - 3 * NF];
        fr[n] = val * win[n] * gain;
    }
}

// ---- melspectrogram / mfcc (zaf.py:324-454), one frame per workgroup ----------------------------------------------
// STFT of the frame as k_stft_f64, |X| (mel) or |X|^2 (mfcc) of bins 1..W/2 (zaf.py:370, :437-439: DC dropped, Nyquist
// kept), the filterbank rows as bands (zaf.py:373 / :445; the matrix is 1-2 % dense), and for mfcc log(. + eps) and rows
// 1..n_coefs of the orthonormal DCT-II (zaf.py:443-452).
// filterbank rows as bands over `mag` (bins 1..W/2 at slots 0..), log + DCT rows for mfcc, store; `mel` = n_filters scratch doubles
__device__ void mel_tail_f64(const double* mag, double* mel, const double* __restrict__ fb, const int* __restrict__ fb_meta,
                             const double* __restrict__ dct, double* __restrict__ out, long long clip, int t, int T, int TP, int layout,
                             int n_filters, int n_coefs) {
    const bool mfcc = n_coefs > 0;
    for (int m = threadIdx.x; m < n_filters; m += (int)blockDim.x) {
        const int lo = fb_meta[3 * m], cnt = fb_meta[3 * m + 1];
        const double* row = fb + fb_meta[3 * m + 2];
        double acc = 0.0;
        for (int j = 0; j < cnt; ++j) acc += row[j] * mag[lo + j];
        mel[m] = mfcc ? log(acc + 2.220446049250313e-16) : acc;   // np.finfo(float).eps (zaf.py:446)
    }
    __syncthreads();
    const int rows = mfcc ? n_coefs : n_filters;
    const long long stride = layout == ZAFX_LAYOUT_FT ? TP : 1;
    const long long base = layout == ZAFX_LAYOUT_FT ? clip * rows * TP + t : (clip * T + t) * rows;
    for (int r = threadIdx.x; r < rows; r += (int)blockDim.x) {
        double v;
        if (mfcc) {
            const double* d = dct + (long long)r * n_filters;
            v = 0.0;
            for (int m = 0; m < n_filters; ++m) v += d[m] * mel[m];
        } else {
            v = mel[r];
        }
        out[base + r * stride] = v;
    }
}

__global__ __launch_bounds__(kThreadsBig) void k_mel_f64(
    const double* __restrict__ x, const double* __restrict__ win, const double2* __restrict__ tw, const double2* __restrict__ tws,
    const double* __restrict__ fb, const int* __restrict__ fb_meta, const double* __restrict__ dct, double* __restrict__ out,
    long long n_samples, int hop, int T, int TP, int log2n, int layout, int n_filters, int n_coefs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int N = 1 << log2n;
    const bool mfcc = n_coefs > 0;
    double2* a = reinterpret_cast<double2*>(smem_raw);
    double2* b = a + N;
    const long long g = xcd_order((int)blockIdx.x, (int)gridDim.x);   // neighbouring frames to one XCD: the rows they share lines of meet in its L2
    const long long clip = g / T;
    const int t = (int)(g - clip * T);
    const double* xc = x + clip * n_samples;
    const long long s0 = (long long)t * hop - N;
    for (int n = threadIdx.x; n < N; n += (int)blockDim.x) {
        const long long s = s0 + 2 * n;
        const double u = (s >= 0 && s < n_samples) ? xc[s] * win[2 * n] : 0.0;
        const double v = (s + 1 >= 0 && s + 1 < n_samples) ? xc[s + 1] * win[2 * n + 1] : 0.0;
        a[n] = make_double2(u, v);
    }
    __syncthreads();
    const double2* z = fft_lds(a, b, log2n, tw);
    double* mag = reinterpret_cast<double*>(z == a ? b : a);   // the idle FFT buffer: N bins, then n_filters band sums
    double* mel = mag + N;
    auto put = [&](int k, double2 v) {   // bin k >= 1 -> slot k - 1
        const double h = hypot(v.x, v.y);   // np.abs of a complex (zaf.py:370); the power is its square (:437-439)
        mag[k - 1] = mfcc ? h * h : h;
    };
    for (int k = threadIdx.x; k < N / 2; k += (int)blockDim.x) {
        if (k == 0) {
            const double2 z0 = z[0], zc = z[N / 2];
            put(N, make_double2(z0.x - z0.y, 0.0));
            put(N / 2, dconj(zc));
        } else {
            const double2 zk = z[k], zn = z[N - k];
            const double2 e = make_double2(0.5 * (zk.x + zn.x), 0.5 * (zk.y - zn.y));
            const double2 d = make_double2(0.5 * (zk.x - zn.x), 0.5 * (zk.y + zn.y));
            const double2 to = dmul(tws[k], make_double2(d.y, -d.x));
            put(k, dadd(e, to));
            put(N - k, dconj(dsub(e, to)));
        }
    }
    __syncthreads();
    mel_tail_f64(mag, mel, fb, fb_meta, dct, out, clip, t, T, TP, layout, n_filters, n_coefs);
}

// ---------------------------------------------------------------------------------
// k_mel_ft8_f64: melspectrogram / mfcc in the reference's own dtype (zaf.py:369-373, :436-452) on the tiled structure of k_stft_ft8_f64.
// W = 2048, reference layout, up to 128 filters.  A workgroup is 8 waves, a wave one frame at a time: the 1024-point packed transform in
// registers + the wave's own 17 KB of LDS (fft1024_f64), the real split straight into |X| (mel) or |X|^2 (mfcc) of bins 1 .. 1024, which
// replace the frame in that buffer as S[c].  The filterbank is 98.6 % zeros (1 889 of 131 072 entries at 128 filters) and the float64
// matrix rate of gfx950 equals its vector rate, so the product runs on the vector pipe over the non-zeros only, wave-local (no barrier):
// the host deals the non-zeros, in row-major order, to the 64 lanes in equal consecutive shares (zafx_mel64.hpp: 30 entries per lane at
// 128 filters, whatever the filters' lengths); an entry is 16 bytes {value, column, slot}, a step one coalesced 16-byte load, one
// ds_read_b64 of the column and one fma; where a filter ends inside a lane's share the running sum goes to a partial-sum slot in LDS, and a
// lane then adds the slots of its (at most two) filters in ascending
// column order (deterministic).  mfcc: log(. + eps) of the 128 band sums (zaf.py:446), then the DCT-II rows as 2 x 32 lanes (coefficient x
// half of the filters) from a transposed table, halves added across the wave.
// A TILE is 16 frames of one clip = two rounds of the 8 waves, so that every output row is written as one 128-byte line of float64: the
// rounds' results wait in a [row][16] staging array.  Samples of the wave's next frame are requested as soon as the transform has left the
// registers for LDS (a second set of 64 registers for them under the transform spilled).
// ---------------------------------------------------------------------------------
constexpr int kMel64Rows = 128;                      // filters (mel) / coefficients (mfcc) the staging array holds
constexpr int kMel64OutPitch = 17;                   // doubles per staged row of 16 frames (+ 1: rows on distinct banks)
constexpr int kMel64Spare = 2 * kF64Pitch - kF64N;   // doubles of a wave's frame buffer behind S: partial sums, then the log-mel column

ZAFX_PROF_ARRAY(g_prof_mel64)
template <bool MFCC>
__global__ __launch_bounds__(kF64Frames * 64) void k_mel_ft8_f64(const double* __restrict__ x, const double* __restrict__ win, const double2* __restrict__ tw,
                                                                  const double2* __restrict__ tws, const int4* __restrict__ stream, const int2* __restrict__ fin,
                                                                  const double* __restrict__ dctT, double* __restrict__ out, long long n_samples, int hop, int T, int TP,
                                                                  int tiles, int total_tiles, int n_filters, int n_coefs, int n_steps, int n_slots, int max_parts,
                                                                  int cpitch, int dct_half) {
    constexpr int N = kF64N, W = 2 * N, FPB = kF64Frames, PITCH = kF64Pitch, OP = kMel64OutPitch;
    static_assert(FPB == 8, "two rounds of eight frames make a 16-frame tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double2* frames = reinterpret_cast<double2*>(smem_raw);
    double* stage = reinterpret_cast<double*>(frames + FPB * PITCH);   // [rows][OP]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double2* buf = frames + wave * PITCH;
    double* S = reinterpret_cast<double*>(buf);   // S[c] = |X[c + 1]| or |X[c + 1]|^2 (zaf.py:370, :437-439: bins 1 .. W/2)
    double* parts = S + N;
    double* logmel = parts + n_slots;
    const int rows = MFCC ? n_coefs : n_filters;
    const bool xcd = gridDim.x % 8 == 0;
    const bool has0 = lane < n_filters, has1 = lane + 64 < n_filters;   // the lane's (at most two) filters and where their partial sums are
    const int2 f0 = has0 ? fin[lane] : make_int2(0, 0), f1 = has1 ? fin[lane + 64] : make_int2(0, 0);
    auto load_frame = [&](double2 (&d)[16], int tlv, int round) {   // raw samples of this wave's frame of round `round` of tile `tlv`
        if (tlv >= total_tiles) return;
        const int tl = xcd ? xcd_order(tlv, total_tiles) : tlv;
        const int clip = tl / tiles, t = (tl % tiles) * 16 + round * FPB + wave;
        const double* xc = x + (long long)clip * n_samples;
        const long long s0 = (long long)t * hop - N;   // floor(W / 2) samples of left padding (zaf.py:99, :112)
        if (t < T && s0 >= 0 && s0 + W <= n_samples && ((s0 | n_samples) & 1) == 0) {   // (uniform) interior frame, 16-byte loads
            const double2* xp = reinterpret_cast<const double2*>(xc + s0);
#pragma unroll
            for (int i = 0; i < 16; ++i) d[i] = xp[lane + 64 * i];
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const long long s = s0 + 2 * (lane + 64 * i);
                d[i].x = (t < T && s >= 0 && s < n_samples) ? xc[s] : 0.0;
                d[i].y = (t < T && s + 1 >= 0 && s + 1 < n_samples) ? xc[s + 1] : 0.0;
            }
        }
    };
    double2 v[16];
    load_frame(v, blockIdx.x, 0);
    PROF_INIT(g_prof_mel64);
    for (int tlv = blockIdx.x; tlv < total_tiles; tlv += gridDim.x) {
        const int tl = xcd ? xcd_order(tlv, total_tiles) : tlv;
        const int clip = tl / tiles, t0 = (tl % tiles) * 16;
#pragma unroll 1
        for (int round = 0; round < 2; ++round) {
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));   // (opaque per frame: window and twiddle values are re-read from L1, not hoisted out of the loop and spilled)
            {
                const double2* wp = reinterpret_cast<const double2*>(win);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const double2 w = wp[lane_o + 64 * i];
                    v[i] = make_double2(v[i].x * w.x, v[i].y * w.y);
                }
            }
            PROF_MARK(0);
            double2 w2[16];
#pragma unroll
            for (int r = 1; r < 16; ++r) w2[r] = root1024(tw, 4 * r * (lane_o & 15));   // exp(-2 pi i r k / 256)
            const int4* st = stream + lane_o;
            double2 tk[8];
            fft1024_f64<true>(v, buf, lane_o, w2, tw, tk, tws);   // (the real split's roots are requested with the last pass's)
            PROF_MARK(1);
            load_frame(v, round == 0 ? tlv : tlv + (int)gridDim.x, round ^ 1);   // the transform is in LDS: the wave's next frame flies under split, product and stores
            // real split X[k] = E + t_k O, X[N-k] = conj(E - t_k O) (as k_stft_f64), straight into levels
            double ma[8], mb[8];
#pragma unroll
            for (int i0 = 0; i0 < 8; i0 += 4) {   // (four pairs at a time: eight kept the registers above the budget)
                double2 zk[4], zn[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k = lane_o + 64 * (i0 + i);
                    zk[i] = buf[physd(k)];
                    zn[i] = buf[physd(k == 0 ? N / 2 : N - k)];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double2 ev = make_double2(0.5 * (zk[i].x + zn[i].x), 0.5 * (zk[i].y - zn[i].y));
                    const double2 d = make_double2(0.5 * (zk[i].x - zn[i].x), 0.5 * (zk[i].y + zn[i].y));
                    const double2 to = dmul(tk[i0 + i], make_double2(d.y, -d.x));
                    double2 xk = dadd(ev, to), xn = dsub(ev, to);
                    if (i0 + i == 0 && lane_o == 0) {   // k = 0: the pair is (X[N/2] = conj z[N/2], X[N] = Re z[0] - Im z[0]); X[0] is not a mel column
                        xk = zn[0];
                        xn = make_double2(zk[0].x - zk[0].y, 0.0);
                    }
                    const double pa = xk.x * xk.x + xk.y * xk.y, pb = xn.x * xn.x + xn.y * xn.y;
                    ma[i0 + i] = MFCC ? pa : sqrt(pa);
                    mb[i0 + i] = MFCC ? pb : sqrt(pb);
                }
            }
            int4 e[8];   // first eight entries of the lane_o's share of the filterbank (requested here: under the split they spilled)
#pragma unroll
            for (int q = 0; q < 8; ++q) e[q] = st[q * 64];
            frame_sync<64>();   // every lane_o has read its pairs: the levels replace the frame
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = lane_o + 64 * i;
                S[k == 0 ? N / 2 - 1 : k - 1] = ma[i];
                S[N - 1 - k] = mb[i];
            }
            frame_sync<64>();
            PROF_MARK(2);
            // filterbank over the non-zeros: the lane_o's stream, eight entries requested while the eight before are used
            {
                double acc = 0.0;
                for (int b = 0; b < n_steps; b += 8) {
                    int4 nx[8];
                    if (b + 8 < n_steps) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) nx[q] = st[(b + 8 + q) * 64];
                    }
                    double c[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) c[q] = S[e[q].z];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        acc = fma(__hiloint2double(e[q].y, e[q].x), c[q], acc);
                        if (e[q].w >= 0) {
                            parts[e[q].w] = acc;
                            acc = 0.0;
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) e[q] = nx[q];
                }
            }
            double dn[16];   // mfcc: the first sixteen terms' DCT values, in flight under the partial sums and the logarithms
            const int dh = lane_o >> 5, dcl = lane_o & 31;
            const double* dp = dctT + (long long)dh * dct_half * cpitch + dcl;
            if (MFCC) {
#pragma unroll
                for (int q = 0; q < 16; ++q) dn[q] = dp[q * cpitch];
            }
            frame_sync<64>();
            PROF_MARK(3);
            const int fcol = round * FPB + wave;
            {   // a lane_o adds the partial sums of its filters in ascending column order
                double s0 = 0.0, s1 = 0.0;
                for (int p = 0; p < max_parts; ++p) {
                    if (p < f0.y) s0 += parts[f0.x + p];
                    if (p < f1.y) s1 += parts[f1.x + p];
                }
                if (MFCC) {   // np.finfo(float).eps (zaf.py:446)
                    if (has0) logmel[lane_o] = log(s0 + 2.220446049250313e-16);
                    if (has1) logmel[lane_o + 64] = log(s1 + 2.220446049250313e-16);
                    if (n_filters + lane_o < 2 * dct_half) logmel[n_filters + lane_o] = 0.0;   // (the DCT's padded terms: zero rows of the table times these)
                } else {
                    if (has0) stage[lane_o * OP + fcol] = s0;
                    if (has1) stage[(lane_o + 64) * OP + fcol] = s1;
                }
            }
            PROF_MARK(4);
            if (MFCC) {
                frame_sync<64>();
                // scipy.fftpack.dct(., norm="ortho") rows 1 .. n_coefs (zaf.py:449-452) as a matrix: lane_o = (coefficient, half of the filters)
                // (the table holds 2 dct_half rows, zero behind n_filters; sixteen terms requested while the sixteen before are used, four chains)
                const double* lm = logmel + dh * dct_half;
                for (int c0 = 0; c0 < n_coefs; c0 += 32) {
                    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
                    for (int n = 0; n < dct_half; n += 16) {
                        double d[16], y[16];
#pragma unroll
                        for (int q = 0; q < 16; ++q) d[q] = dn[q];
                        const int nn = n + 16 < dct_half ? n + 16 : 0, cn = n + 16 < dct_half ? c0 : c0 + 32;   // next: this group's next terms, or the next group's first
                        if (n + 16 < dct_half || c0 + 32 < n_coefs) {
#pragma unroll
                            for (int q = 0; q < 16; ++q) dn[q] = dp[(nn + q) * cpitch + cn];
                        }
#pragma unroll
                        for (int q = 0; q < 16; ++q) y[q] = lm[n + q];
#pragma unroll
                        for (int q = 0; q < 16; q += 4) {
                            s0 = fma(d[q], y[q], s0);
                            s1 = fma(d[q + 1], y[q + 1], s1);
                            s2 = fma(d[q + 2], y[q + 2], s2);
                            s3 = fma(d[q + 3], y[q + 3], s3);
                        }
                    }
                    double sum = (s0 + s1) + (s2 + s3);
                    sum += __shfl_xor(sum, 32, 64);
                    if (dh == 0 && c0 + dcl < n_coefs) stage[(c0 + dcl) * OP + fcol] = sum;
                }
            }
            frame_sync<64>();   // S / partial sums / log-mel column are read: the next frame may take the buffer
            PROF_MARK(5);
        }
        lds_barrier();
        PROF_MARK(6);
        {   // the tile's rows: sixteen lanes (frames) to a 128-byte line, 32 rows per instruction of the workgroup
            const int f = tid & 15;
            double* o = out + (long long)clip * rows * TP + t0 + f;
            if (t0 + f < T)
                for (int r = tid >> 4; r < rows; r += 32) o[(long long)r * TP] = stage[r * OP + f];
        }
        PROF_MARK(7);
        lds_barrier();
        PROF_MARK(8);
    }
}

// ---- windows that are not a power of two (the reference's np.fft takes any length): Bluestein ---------------------------
// W-point DFT as a convolution of length M = 2^ceil(log2(2W-1)):  n k = (n^2 + k^2 - (k-n)^2) / 2, so with
// c[n] = exp(-i pi n^2 / W):   X[k] = c[k] * sum_n (x[n] c[n]) conj(c)[k-n].  The host provides c (exact: n^2 mod 2W in
// integers, long double) and Bhat = FFT_M(conj(c) wrapped); the two transforms of length M run in LDS as above.
// In: a[n] = x[n] for n < W.  Out: X[k], k < W, in the returned buffer (the other one is free).
__device__ double2* bluestein_lds(double2* a, double2* b, int W, int log2m, const double2* __restrict__ twm,
                                  const double2* __restrict__ chirp, const double2* __restrict__ bhat) {
    const int M = 1 << log2m;
    for (int n = threadIdx.x; n < M; n += (int)blockDim.x) a[n] = n < W ? dmul(a[n], chirp[n]) : make_double2(0.0, 0.0);
    __syncthreads();
    double2* z = fft_lds(a, b, log2m, twm);
    for (int j = threadIdx.x; j < M; j += (int)blockDim.x) z[j] = dconj(dmul(z[j], bhat[j]));   // conj: the next forward transform inverts
    __syncthreads();
    double2* o = z == a ? b : a;
    double2* y = fft_lds(z, o, log2m, twm);
    const double inv = 1.0 / (double)M;
    for (int k = threadIdx.x; k < W; k += (int)blockDim.x) {
        const double2 c = dconj(y[k]);
        y[k] = dmul(chirp[k], make_double2(c.x * inv, c.y * inv));
    }
    __syncthreads();
    return y;
}

__global__ __launch_bounds__(kThreadsBig) void k_stft_bs_f64(
    const double* __restrict__ x, const double* __restrict__ win, const double2* __restrict__ twm, const double2* __restrict__ chirp,
    const double2* __restrict__ bhat, const double* __restrict__ fb, const int* __restrict__ fb_meta, const double* __restrict__ dct,
    double2* __restrict__ out, long long n_samples, int hop, int T, int TP, int W, int log2m, int layout, int spec, int n_filters,
    int n_coefs, int mel_mode) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int M = 1 << log2m;
    double2* a = reinterpret_cast<double2*>(smem_raw);
    double2* b = a + M;
    const long long g = xcd_order((int)blockIdx.x, (int)gridDim.x);   // neighbouring frames to one XCD: the rows they share lines of meet in its L2
    const long long clip = g / T;
    const int t = (int)(g - clip * T);
    const double* xc = x + clip * n_samples;
    const long long s0 = (long long)t * hop - W / 2;   // floor(W/2) samples of left padding (zaf.py:99)
    for (int n = threadIdx.x; n < W; n += (int)blockDim.x) {
        const long long s = s0 + n;
        a[n] = make_double2((s >= 0 && s < n_samples) ? xc[s] * win[n] : 0.0, 0.0);
    }
    __syncthreads();
    const double2* X = bluestein_lds(a, b, W, log2m, twm, chirp, bhat);
    if (mel_mode) {   // zaf.py:370 / :437-439: bins 1 .. int(W/2)
        double* mag = reinterpret_cast<double*>(X == a ? b : a);
        double* mel = mag + W / 2;
        for (int k = 1 + threadIdx.x; k <= W / 2; k += (int)blockDim.x) {
            const double h = hypot(X[k].x, X[k].y);
            mag[k - 1] = n_coefs > 0 ? h * h : h;
        }
        __syncthreads();
        mel_tail_f64(mag, mel, fb, fb_meta, dct, reinterpret_cast<double*>(out), clip, t, T, TP, layout, n_filters, n_coefs);
        return;
    }
    const int rows = spec ? W / 2 + 1 : W;
    const long long stride = layout == ZAFX_LAYOUT_FT ? TP : 1;
    const long long base = layout == ZAFX_LAYOUT_FT ? clip * rows * TP + t : (clip * T + t) * rows;
    for (int k = threadIdx.x; k < rows; k += (int)blockDim.x) {
        double2 v = X[k];
        if (k == 0 || 2 * k == W) v.y = 0.0;   // real input: DC and Nyquist are real (np.fft returns exact zeros there)
        if (spec >= ZAFX_SPECTRUM_MAGNITUDE) {
            const double pw = v.x * v.x + v.y * v.y;
            reinterpret_cast<double*>(out)[base + k * stride] = spec == ZAFX_SPECTRUM_MAGNITUDE ? sqrt(pw) : pw;
        } else {
            out[base + k * stride] = v;
        }
    }
}

// real(ifft(X)) of one frame per workgroup for any W: ifft(X) = conj(fft(conj(X))) / W
__global__ __launch_bounds__(kThreadsBig) void k_ifft_frames_bs_f64(
    const double2* __restrict__ spec, const double2* __restrict__ twm, const double2* __restrict__ chirp, const double2* __restrict__ bhat,
    double* __restrict__ frames, int T, int TP, int W, int log2m, int layout, int one) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int M = 1 << log2m, rows = one ? W / 2 + 1 : W;
    double2* a = reinterpret_cast<double2*>(smem_raw);
    double2* b = a + M;
    const long long g = xcd_order((int)blockIdx.x, (int)gridDim.x);   // neighbouring frames to one XCD: the rows they share lines of meet in its L2
    const long long clip = g / T;
    const int t = (int)(g - clip * T);
    const long long stride = layout == ZAFX_LAYOUT_FT ? TP : 1;
    const double2* sp = layout == ZAFX_LAYOUT_FT ? spec + clip * rows * TP + t : spec + (clip * T + t) * rows;
    for (int k = threadIdx.x; k < W; k += (int)blockDim.x) {
        // one-sided input: X[W-k] = conj X[k]; conj(X) goes in
        a[k] = (one && k > W / 2) ? sp[(long long)(W - k) * stride] : dconj(sp[(long long)k * stride]);
    }
    __syncthreads();
    const double2* y = bluestein_lds(a, b, W, log2m, twm, chirp, bhat);
    double* fr = frames + g * W;
    const double inv = 1.0 / (double)W;
    for (int n = threadIdx.x; n < W; n += (int)blockDim.x) fr[n] = y[n].x * inv;   // Re(conj(.)) = Re(.)
}

// exp(-i pi num / den) with the angle reduced in integers first (num, den > 0)
__device__ __forceinline__ double2 unit_mpi(long long num, long long den) {
    const long long r = num % (2 * den);
    double sn, cs;
    sincospi((double)r / (double)den, &sn, &cs);
    return make_double2(cs, -sn);
}

// MDCT of any even window length, the reference's own formulation (zaf.py:1047-1073): W-point FFT of x w pre, first W/2
// outputs times post, real part;  pre[n] = exp(-i pi n / W),  post[k] = exp(-i pi (W/2 + 1)(k + 1/2) / W)
__global__ __launch_bounds__(kThreadsBig) void k_mdct_bs_f64(
    const double* __restrict__ x, const double* __restrict__ win, const double2* __restrict__ twm, const double2* __restrict__ chirp,
    const double2* __restrict__ bhat, double* __restrict__ out, long long n_samples, int T, int TP, int W, int log2m, int layout) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int M = 1 << log2m, F = W / 2;
    double2* a = reinterpret_cast<double2*>(smem_raw);
    double2* b = a + M;
    const long long g = xcd_order((int)blockIdx.x, (int)gridDim.x);   // neighbouring frames to one XCD: the rows they share lines of meet in its L2
    const long long clip = g / T;
    const int t = (int)(g - clip * T);
    const double* xc = x + clip * n_samples;
    const long long s0 = (long long)t * F - F;   // left pad = W/2 (zaf.py:1036-1041)
    for (int n = threadIdx.x; n < W; n += (int)blockDim.x) {
        const long long s = s0 + n;
        const double v = (s >= 0 && s < n_samples) ? xc[s] * win[n] : 0.0;
        const double2 pre = unit_mpi(n, W);
        a[n] = make_double2(v * pre.x, v * pre.y);
    }
    __syncthreads();
    const double2* X = bluestein_lds(a, b, W, log2m, twm, chirp, bhat);
    const long long stride = layout == ZAFX_LAYOUT_FT ? TP : 1;
    const long long base = layout == ZAFX_LAYOUT_FT ? clip * F * TP + t : (clip * T + t) * F;
    for (int k = threadIdx.x; k < F; k += (int)blockDim.x) {
        const double2 post = unit_mpi((long long)(F + 1) * (2 * k + 1), 2LL * W);
        out[base + k * stride] = X[k].x * post.x - X[k].y * post.y;
    }
}

// IMDCT frames of any even window length (zaf.py:1138-1169): W-point FFT of X pre zero-padded from F to W, times post, real
// part, times 2 w;  pre[k] = exp(-i pi (F + 1) k / W),  post[n] = exp(-i pi (n + 1/2 + F/2) / W) / F
__global__ __launch_bounds__(kThreadsBig) void k_imdct_frames_bs_f64(
    const double* __restrict__ coefs, const double* __restrict__ win, const double2* __restrict__ twm, const double2* __restrict__ chirp,
    const double2* __restrict__ bhat, double* __restrict__ frames, int T, int TP, int W, int log2m, int layout) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int M = 1 << log2m, F = W / 2;
    double2* a = reinterpret_cast<double2*>(smem_raw);
    double2* b = a + M;
    const long long g = xcd_order((int)blockIdx.x, (int)gridDim.x);   // neighbouring frames to one XCD: the rows they share lines of meet in its L2
    const long long clip = g / T;
    const int t = (int)(g - clip * T);
    const long long stride = layout == ZAFX_LAYOUT_FT ? TP : 1;
    const double* cp = layout == ZAFX_LAYOUT_FT ? coefs + clip * F * TP + t : coefs + (clip * T + t) * F;
    for (int k = threadIdx.x; k < W; k += (int)blockDim.x) {
        if (k < F) {
            const double v = cp[(long long)k * stride];
            const double2 pre = unit_mpi((long long)(F + 1) * k, W);
            a[k] = make_double2(v * pre.x, v * pre.y);
        } else {
            a[k] = make_double2(0.0, 0.0);
        }
    }
    __syncthreads();
    const double2* Y = bluestein_lds(a, b, W, log2m, twm, chirp, bhat);
    double* fr = frames + g * W;
    for (int n = threadIdx.x; n < W; n += (int)blockDim.x) {
        const double2 post = unit_mpi(2LL * n + 1 + F, 2LL * W);   // (n + 1/2 + F/2) / W = (2n + 1 + F) / (2W)
        fr[n] = 2.0 * (Y[n].x * post.x - Y[n].y * post.y) / (double)F * win[n];
    }
}

// ---- cqtspectrogram / cqtchromagram (zaf.py:562-700) ---------------------------------------------------------------
// A frame is fft_length real samples (up to 32768: 512 KB as complex128, more than LDS).  Decimation in time by
// N1 = fft_length / N2, N2 = min(fft_length, 4096): the N1 sub-sequences x[n1 + N1 m] are transformed in LDS one after the
// other (F_n1, N2 points each, parked in a per-workgroup global scratch when N1 > 1), and a spectrum bin is recombined
// only where the sparse kernel has a column:  X[c] = sum_n1 exp(-2 pi i n1 c / W) F_n1[c mod N2].  Then the CSR
// mat-vec and np.absolute (zaf.py:630-632): one wavefront per row, lanes over its entries.  Persistent workgroups.
__global__ __launch_bounds__(kThreadsBig) void k_cqt_f64(
    const double* __restrict__ x, const double2* __restrict__ tw, const double2* __restrict__ roots, const int* __restrict__ indptr,
    const int* __restrict__ indices, const double2* __restrict__ values, double2* __restrict__ scratch, double* __restrict__ out,
    long long n_samples, int step, int left, int T, int TP, long long total_frames, int log2w, int n_bins, int chroma_res, int layout) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int W = 1 << log2w;
    const int log2n2 = log2w < 12 ? log2w : 12, N2 = 1 << log2n2, N1 = W >> log2n2;
    double2* a = reinterpret_cast<double2*>(smem_raw);
    double2* b = a + N2;
    double* spec = reinterpret_cast<double*>(b + N2);   // n_bins magnitudes of the frame
    double2* F = scratch + (long long)blockIdx.x * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = (int)blockDim.x / 64;
    for (long long g = blockIdx.x; g < total_frames; g += gridDim.x) {
        const long long clip = g / T;
        const int t = (int)(g - clip * T);
        const double* xc = x + clip * n_samples;
        const long long s0 = (long long)t * step - left;   // zaf.py:612-620: `left` zeros in front of the clip
        const double2* f_lds = nullptr;
        for (int n1 = 0; n1 < N1; ++n1) {
            for (int m = threadIdx.x; m < N2; m += (int)blockDim.x) {
                const long long s = s0 + n1 + (long long)N1 * m;
                a[m] = make_double2((s >= 0 && s < n_samples) ? xc[s] : 0.0, 0.0);
            }
            __syncthreads();
            const double2* z = fft_lds(a, b, log2n2, tw);
            if (N1 == 1) {
                f_lds = z;
            } else {
                for (int k = threadIdx.x; k < N2; k += (int)blockDim.x) F[(long long)n1 * N2 + k] = z[k];
                __syncthreads();   // the buffers are refilled by the next sub-sequence
            }
        }
        if (N1 > 1) {
            __threadfence_block();
            __syncthreads();
        }
        for (int r = wave; r < n_bins; r += n_waves) {
            double2 acc = make_double2(0.0, 0.0);
            for (int e = indptr[r] + lane; e < indptr[r + 1]; e += 64) {
                const int c = indices[e];
                double2 xcol;
                if (N1 == 1) {
                    xcol = f_lds[c];
                } else {
                    xcol = make_double2(0.0, 0.0);
                    const int k2 = c & (N2 - 1);
                    for (int n1 = 0; n1 < N1; ++n1)
                        xcol = dadd(xcol, dmul(roots[(int)(((long long)n1 * c) & (W - 1))], F[(long long)n1 * N2 + k2]));
                }
                acc = dadd(acc, dmul(values[e], xcol));
            }
            for (int off = 32; off > 0; off >>= 1) {
                acc.x += __shfl_down(acc.x, off, 64);
                acc.y += __shfl_down(acc.y, off, 64);
            }
            if (lane == 0) spec[r] = hypot(acc.x, acc.y);
        }
        __syncthreads();
        const int rows = chroma_res > 0 ? chroma_res : n_bins;
        const long long stride = layout == ZAFX_LAYOUT_FT ? TP : 1;
        const long long base = layout == ZAFX_LAYOUT_FT ? clip * rows * TP + t : (clip * T + t) * rows;
        for (int r = threadIdx.x; r < rows; r += (int)blockDim.x) {
            double v;
            if (chroma_res > 0) {   // zaf.py:693-698: rows i, i + r, i + 2r, ... summed in ascending order
                v = 0.0;
                for (int q = r; q < n_bins; q += chroma_res) v += spec[q];
            } else {
                v = spec[r];
            }
            out[base + r * stride] = v;
        }
        __syncthreads();   // spec and the FFT buffers are reused by the next frame
    }
}


// ---------------------------------------------------------------------------------
// k_cqt_ft_f64: cqtspectrogram / cqtchromagram in the reference's own dtype (zaf.py:562-700) on the structure of the float32 k_cqt.
// fft_length W = 32768: the frame is one N = 16384-point packed transform, N = 16 x 1024 by decimation in frequency --
//     Z[16 k + q] = FFT_1024{ y_q }[k],   y_q[m] = w_N^(m q) sum_r z[m + 1024 r] w_16^(r q)
// -- a radix-16 pass across the workgroup's registers (a thread owns m and m + 512: coalesced 16-byte loads of z[m + 1024 r]), then sixteen
// wave-local 1024-point transforms (fft1024 below: registers + the wave's own 17 KB of LDS).  In float64 only eight of the sixteen
// sub-sequences fit LDS, so a frame takes two ROUNDS of the 8 waves: the even q first, the odd q wait in registers (64 per thread).  A pair
// (c, N - c) of the real split lies in sub-transforms q and 16 - q: the same round.  Only the one-sided bins the kernel matrix reads are
// split (zafx_cqt64.hpp: 2 634 of 16 384 for the reference's kernel), by fixed threads into registers; after the second round they go to a
// compact array in the (then free) transform buffers, and the matrix's non-zeros run over it as one stream per thread -- equal shares in
// CSR order, a partial sum per (thread, row) piece, rows finished in ascending column order (deterministic) --, then np.absolute and, for
// the chromagram, the sums over octaves (zaf.py:693-698).  Workgroups share clips per XCD as k_cqt's do, so that the 94.6 % overlap of
// neighbouring frames is served by one L2.
// Twiddles: w_N^(m q) as products of the table values w^m, w^2m, w^4m, w^8m (at most three factors: 3e-16); the sub-transforms' first
// exchange takes exp(-2 pi i r k / 256) from a 4-KB table in LDS, the last pass its roots from a 12-KB one (from L2 they cost three round
// trips per sub-transform; in registers, 48 of them, they spilled).
// ---------------------------------------------------------------------------------
__device__ __forceinline__ double2 root1024s(const double2* __restrict__ tws, int m) {   // exp(-2 pi i m / 1024), m < 1024, from the plan's exp(-2 pi i j / 32768)
    return tws[m << 5];
}
// 1024-point forward transform of one wavefront, input and output in `buf` (natural order)
// FULL = false: no index k = bin >> 4 the split reads lies in 256 .. 767 (bins c and N - c of every kernel column c: k <= klo and k >= 1023 - klo with
// klo = (highest column) >> 4 < 256 -- the reference's kernel: columns up to bin 2634, klo = 164), and the last pass neither forms nor writes the two
// middle quarters of its output: 8 of 16 stores (256 clips x 30 s: 16.61 -> 15.94 ms).  Skipping, wave by wave, also the quarter-waves of the outer
// quarters that klo leaves out (6 stores) is SLOWER (16.2: a scalar branch per store).
template <int RS, bool FULL>   // RS: stride of the root table: 1 (the LDS copy) or 32 (the plan's roots of 32768); FULL: klo >= 256
__device__ __forceinline__ void fft1024_cq(double2* buf, int lane, const double2* w2tab, const double2* rtab) {   // rtab[m] = exp(-2 pi i m / 1024), m < 768 (LDS)
    double2 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = buf[physd(lane + 64 * i)];
    frame_sync<64>();
    dft16d(v);
#pragma unroll
    for (int r = 0; r < 16; ++r) lds_st(&buf[physd(16 * lane + r)], v[r]);
    frame_sync<64>();
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = buf[physd(lane + 64 * i)];
    frame_sync<64>();
    {
        const int k = lane & 15;
#pragma unroll
        for (int r = 1; r < 16; ++r) v[r] = dmul(v[r], w2tab[r * 16 + k]);   // exp(-2 pi i r k / 256)
        dft16d(v);
        const int base = ((lane >> 4) << 8) + k;
#pragma unroll
        for (int r = 0; r < 16; ++r) lds_st(&buf[physd(base + 16 * r)], v[r]);
    }
    frame_sync<64>();
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = buf[physd(lane + 64 * i)];
    frame_sync<64>();
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int k = lane + 64 * b;
        double2 a0 = v[b], a1 = dmul(v[b + 4], rtab[k * RS]), a2 = dmul(v[b + 8], rtab[2 * k * RS]), a3 = dmul(v[b + 12], rtab[3 * k * RS]);
        if constexpr (FULL) {
            dft4d(a0, a1, a2, a3);
            lds_st(&buf[physd(k)], a0);
            lds_st(&buf[physd(k + 256)], a1);
            lds_st(&buf[physd(k + 512)], a2);
            lds_st(&buf[physd(k + 768)], a3);
        } else {   // outputs k and k + 768 only
            const double2 t0 = dadd(a0, a2), t1 = dsub(a0, a2), t2 = dadd(a1, a3), t3 = dmul_mi(dsub(a1, a3));
            lds_st(&buf[physd(k)], dadd(t0, t2));
            lds_st(&buf[physd(k + 768)], dsub(t1, t3));
        }
    }
    frame_sync<64>();
}

#ifndef ZAFX_CQ64_SHALLOW
#define ZAFX_CQ64_SHALLOW 1   // the first pass's loads: 1 = the second half requested when the first has arrived (sixteen 16-byte loads in flight per thread instead of
                              // thirty-two: 256 clips x 30 s 16.84 -> 16.69 ms), 2 / 3 = eight / four in flight (16.82 / 16.83), 0 = all thirty-two at once
#endif
#ifndef ZAFX_CQ64_PREFETCH
#define ZAFX_CQ64_PREFETCH 0   // 1, 2: half / all of the next first pass's samples requested behind the transforms of the phase before -- measured 6.8 / 8.9 ms against 4.85 (64 clips x 30 s): the 64 / 128 registers they hold spill
#endif
#ifndef ZAFX_CQ64_BATCH
#define ZAFX_CQ64_BATCH 4   // entries of the contraction's stream in flight per thread (8: 5.15 ms against 4.74 for 64 clips x 30 s with 24-byte entries)
#endif
#ifndef ZAFX_CQ64_RTAB_LDS
#define ZAFX_CQ64_RTAB_LDS 1   // the last pass's roots from LDS (0: from the plan's table in global memory)
#endif
ZAFX_PROF_ARRAY(g_prof_cqt64)
template <int KC2, bool REALK, bool FULL>
__global__ __launch_bounds__(kCq64Threads) void k_cqt_ft_f64(const double* __restrict__ x, const double2* __restrict__ tws, const double2* __restrict__ tw1,
                                                              const int* __restrict__ split_tab, const void* __restrict__ cvals, const int* __restrict__ cmeta,
                                                              const int2* __restrict__ fin, double* __restrict__ out, long long n_samples, int step, int left, int T,
                                                              int TP, int n_clips, int n_groups, int n_bins, int chroma_res, int layout, int n_cols, int n_steps,
                                                              int n_slots, int max_parts) {
    constexpr int N = 16384, W = 2 * N, PITCH = kF64Pitch, NW = kCq64Threads / 64, NT = kCq64Threads;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double2* frames = reinterpret_cast<double2*>(smem_raw);   // NW sub-transform buffers; behind the second round: compact spectrum, partial sums, levels
    double2* w2tab = frames + NW * PITCH;                     // [16][16] exp(-2 pi i r k / 256)
    double2* rtab = w2tab + 256;                              // [768] exp(-2 pi i m / 1024): the last pass's roots of the wave-local transforms
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double2* buf = frames + wave * PITCH;
    if (tid < 256) w2tab[tid] = root1024s(tws, 4 * (tid >> 4) * (tid & 15));
    int col[2][KC2];   // the bins this thread splits: bin | compact index << 14, -1: none
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < KC2; ++j) col[r][j] = split_tab[(r * KC2 + j) * NT + tid];
    for (int i = tid; i < 768; i += NT) rtab[i] = root1024s(tws, i);
    lds_barrier();
    // work list (as k_cqt): group = blockIdx % n_groups (the XCD when n_groups = 8) owns clips group, group + n_groups, ...; its frames, clip
    // after clip, are dealt round-robin to the group's workgroups
    const int group = blockIdx.x % n_groups, slot = blockIdx.x / n_groups;
    const int n_slots_g = (gridDim.x - group + n_groups - 1) / n_groups;
    const long long n_work = (long long)((n_clips - group + n_groups - 1) / n_groups) * T;
    PROF_INIT(g_prof_cqt64);
    // samples z[m + 1024 r] of frame `g` for m = tid + 512 j: buffer loads with the clip as descriptor, so that the zeros around the clip
    // (zaf.py:612-620) are its out-of-range reads -- one path for every frame, one 32-bit offset per lane
    auto load_z = [&](double2 (&z)[16], long long g, int j, int tid_l) {   // (tid_l: the thread index, opaque per frame -- see tid_o)
        if (g >= n_work) return;
        const int clip = group + (int)(g / T) * n_groups, t = (int)(g % T);
        const auto rx = make_rsrc(x + (long long)clip * n_samples, (unsigned)(n_samples * 8));
        const long long s0 = (long long)t * step - left;
        const int voff = (int)(s0 * 8) + (tid_l + NT * j) * 16;   // (negative in front of the clip: out of range as unsigned)
        if (s0 >= 0 && s0 + W <= n_samples) {   // (uniform) the frame lies inside the clip: 16-byte loads
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#if ZAFX_CQ64_SHALLOW >= 2
                if (r % (ZAFX_CQ64_SHALLOW == 2 ? 8 : 4) == 0 && r) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                const auto raw = __builtin_amdgcn_raw_buffer_load_b128(rx, voff + r * 16384, 0, 0);   // (the whole offset per lane: the range check does not see a scalar offset)
                __builtin_memcpy(&z[r], &raw, 16);
            }
        } else {   // a frame over an end of the clip: a 16-byte load that straddles the end comes back as zeros whole, so the two samples are loaded apart
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const auto re = __builtin_amdgcn_raw_buffer_load_b64(rx, voff + r * 16384, 0, 0), im = __builtin_amdgcn_raw_buffer_load_b64(rx, voff + r * 16384 + 8, 0, 0);
                __builtin_memcpy(&z[r].x, &re, 8);
                __builtin_memcpy(&z[r].y, &im, 8);
            }
        }
    };
    // requested ahead -- behind the transforms of the phase before, whose registers they would not fit beside: in flight under the split (and, for
    // the next frame, under the contraction): both halves (m = tid, tid + 512) of the next first pass's samples
    double2 zq[2][16];
    if (ZAFX_CQ64_PREFETCH) load_z(zq[0], slot, 0, tid);
    if (ZAFX_CQ64_PREFETCH == 2) load_z(zq[1], slot, 1, tid);
    for (long long g = slot; g < n_work; g += n_slots_g) {
        const int clip = group + (int)(g / T) * n_groups, t = (int)(g % T);
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));   // (opaque per frame: LDS addresses and table pointers are recomputed, not hoisted out of the loop and spilled)
        double2 xs[2][KC2];   // the thread's bins X[c] (real split of the packed transform, as k_stft_f64)
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            // first pass for this round's eight q (even: the 8-point transform of z[r] + z[r + 8]; odd: of (z[r] - z[r + 8]) w_16^r): the samples
            // are read once per round (the second time from L2) -- sixteen outputs at once held 64 registers through the first round and spilled
            if (!ZAFX_CQ64_PREFETCH) load_z(zq[0], g, 0, tid_o);
#if ZAFX_CQ64_SHALLOW
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            if (ZAFX_CQ64_PREFETCH != 2) load_z(zq[1], g, 1, tid_o);   // (1: only the first half rides ahead; both kept 128 registers through the split and spilled)
            // both halves folded first (16 -> 8 values each: the 128 registers of samples are down to 64 before the transforms need theirs)
            double2 af[2][8];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 8; ++r) af[j][r] = round == 0 ? dadd(zq[j][r], zq[j][r + 8]) : dsub(zq[j][r], zq[j][r + 8]);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int m = tid_o + NT * j;
                double2 (&a)[8] = af[j];
                const double2 w2 = tw1[1024 + m], w4 = tw1[2048 + m], w8 = tw1[3072 + m];   // tw1[i][m] = exp(-2 pi i m 2^i / N): coalesced (the same values out of the
                                                                                             // plan's table of the roots of W lie a cache line apart: 350 KB through the L1 per round)
                double2* fm = frames + physd(m);
                if (round == 0) {
                    dft8d(a);   // a[p] = sum_r z[m + 1024 r] w_16^(2 p r)
                    const double2 w6 = dmul(w2, w4);
                    lds_st(&fm[0 * PITCH], a[0]);
                    lds_st(&fm[1 * PITCH], dmul(a[1], w2));
                    lds_st(&fm[2 * PITCH], dmul(a[2], w4));
                    lds_st(&fm[3 * PITCH], dmul(a[3], w6));
                    lds_st(&fm[4 * PITCH], dmul(a[4], w8));
                    lds_st(&fm[5 * PITCH], dmul(a[5], dmul(w2, w8)));
                    lds_st(&fm[6 * PITCH], dmul(a[6], dmul(w4, w8)));
                    lds_st(&fm[7 * PITCH], dmul(a[7], dmul(w6, w8)));
                } else {
                    const double h = 0.70710678118654752440, c1 = 0.92387953251128675613, s1 = 0.38268343236508977173;
                    a[1] = dmul(a[1], make_double2(c1, -s1));
                    a[2] = dmul(a[2], make_double2(h, -h));
                    a[3] = dmul(a[3], make_double2(s1, -c1));
                    a[4] = dmul_mi(a[4]);
                    a[5] = dmul(a[5], make_double2(-s1, -c1));
                    a[6] = dmul(a[6], make_double2(-h, -h));
                    a[7] = dmul(a[7], make_double2(-c1, -s1));
                    dft8d(a);   // a[p] = sum_r z[m + 1024 r] w_16^((2 p + 1) r)
                    const double2 w1 = tw1[m];
                    const double2 w3 = dmul(w1, w2), w5 = dmul(w1, w4), w9 = dmul(w1, w8);
                    lds_st(&fm[0 * PITCH], dmul(a[0], w1));
                    lds_st(&fm[1 * PITCH], dmul(a[1], w3));
                    lds_st(&fm[2 * PITCH], dmul(a[2], w5));
                    lds_st(&fm[3 * PITCH], dmul(a[3], dmul(w3, w4)));
                    lds_st(&fm[4 * PITCH], dmul(a[4], w9));
                    lds_st(&fm[5 * PITCH], dmul(a[5], dmul(w3, w8)));
                    lds_st(&fm[6 * PITCH], dmul(a[6], dmul(w5, w8)));
                    lds_st(&fm[7 * PITCH], dmul(a[7], dmul(dmul(w3, w4), w8)));
                }
            }
            PROF_MARK(0);
            lds_barrier();
            PROF_MARK(1);
            {
                int tid_f = tid;
                asm volatile("" : "+v"(tid_f));   // (opaque per sub-transform: its forty LDS addresses are recomputed, not kept -- spilled -- across the loop)
                double2* buf_f = frames + (tid_f >> 6) * PITCH;
                if (ZAFX_CQ64_RTAB_LDS) fft1024_cq<1, FULL>(buf_f, tid_f & 63, w2tab, rtab);   // wave p: sub-transform q = 2 p + round
                else fft1024_cq<32, FULL>(buf_f, tid_f & 63, w2tab, tws);
            }
            PROF_MARK(2);
            if (ZAFX_CQ64_PREFETCH) {
                const long long gn = round == 0 ? g : g + n_slots_g;
                load_z(zq[0], gn, 0, tid_o);
                if (ZAFX_CQ64_PREFETCH == 2) load_z(zq[1], gn, 1, tid_o);
            }
            lds_barrier();
            PROF_MARK(3);
#pragma unroll
            for (int j = 0; j < KC2; ++j) {
                int cb = col[round][j];
                asm volatile("" : "+v"(cb));
                const int bin = cb < 0 ? 16 + round : cb & 0x3fff;   // (idle slot: any bin of this round, result unused)
                const int q = bin & 15, k = bin >> 4;
                const int qn = (16 - q) & 15, kn = 1024 - k - (q != 0);   // N - bin = 16 kn + qn
                const double2 zk = frames[(q >> 1) * PITCH + physd(k)], zn = frames[(qn >> 1) * PITCH + physd(kn)];
                const double2 e = make_double2(0.5 * (zk.x + zn.x), 0.5 * (zk.y - zn.y));
                const double2 d = make_double2(0.5 * (zk.x - zn.x), 0.5 * (zk.y + zn.y));
                xs[round][j] = dadd(e, dmul(tws[bin], make_double2(d.y, -d.x)));
            }
            lds_barrier();
            PROF_MARK(4);
        }
        constexpr int CB = ZAFX_CQ64_BATCH;
        using KV = std::conditional_t<REALK, double, double2>;
        const KV* vp = reinterpret_cast<const KV*>(cvals) + tid_o;
        const int* mp = cmeta + tid_o;
        KV kv0[CB], kv1[CB];   // the first two batches of the thread's share of the kernel matrix: requested here, used behind two barriers
        int km0[CB], km1[CB];
#pragma unroll
        for (int q = 0; q < CB; ++q) kv0[q] = vp[q * NT], km0[q] = mp[q * NT];
        if (CB < n_steps) {
#pragma unroll
            for (int q = 0; q < CB; ++q) kv1[q] = vp[(CB + q) * NT], km1[q] = mp[(CB + q) * NT];
        }
        double2* Xc = frames;
        double2* parts = Xc + n_cols;
        double* spec = reinterpret_cast<double*>(parts + n_slots);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < KC2; ++j) {
                int cb = col[r][j];
                asm volatile("" : "+v"(cb));
                if (cb >= 0) Xc[cb >> 14] = xs[r][j];
            }
        lds_barrier();
        PROF_MARK(5);
        {   // cqt_kernel * fft(frame) (zaf.py:631) over the non-zeros: the thread's share, CB entries at a time, two batches in flight (an L2 round
            // trip per batch otherwise: six per frame); the first two were requested in front of the compact spectrum's barrier (kv0 / kv1 above).
            // REALK: the matrix is real up to round-off (the reference's kernel: |imag| / |real| = 1.2e-16): 12 bytes per entry instead of 24
            double2 acc = make_double2(0.0, 0.0);
            for (int b = 0; b < n_steps; b += CB) {
                KV kv2[CB];
                int km2[CB];
                if (b + 2 * CB < n_steps) {
#pragma unroll
                    for (int q = 0; q < CB; ++q) kv2[q] = vp[(b + 2 * CB + q) * NT], km2[q] = mp[(b + 2 * CB + q) * NT];
                }
                double2 xb[CB];   // (all of the batch's bins first: behind the conditional stores below every read would wait on its own)
#pragma unroll
                for (int q = 0; q < CB; ++q) xb[q] = Xc[km0[q] & 0x1fff];
#pragma unroll
                for (int q = 0; q < CB; ++q) {
                    double2 xv = xb[q];
                    if (km0[q] & 0x2000) xv.y = -xv.y;   // a column above W/2: X[c] = conj X[W - c]
                    if constexpr (REALK) acc = make_double2(fma(kv0[q], xv.x, acc.x), fma(kv0[q], xv.y, acc.y));
                    else acc = dadd(acc, dmul(kv0[q], xv));
                    if (km0[q] >> 14) {   // slot + 1 of the partial sum this entry ends
                        parts[(km0[q] >> 14) - 1] = acc;
                        acc = make_double2(0.0, 0.0);
                    }
                }
#pragma unroll
                for (int q = 0; q < CB; ++q) kv0[q] = kv1[q], km0[q] = km1[q], kv1[q] = kv2[q], km1[q] = km2[q];
            }
        }
        lds_barrier();
        PROF_MARK(6);
        for (int r = tid; r < n_bins; r += NT) {   // a row's partial sums in ascending column order, then np.absolute (zaf.py:631)
            const int2 f = fin[r];
            double2 sum = make_double2(0.0, 0.0);
            for (int p = 0; p < f.y; ++p) sum = dadd(sum, parts[f.x + p]);
            spec[r] = hypot(sum.x, sum.y);
        }
        lds_barrier();
        {
            const int rows = chroma_res > 0 ? chroma_res : n_bins;
            const long long stride = layout == ZAFX_LAYOUT_FT ? TP : 1;
            const long long base = layout == ZAFX_LAYOUT_FT ? (long long)clip * rows * TP + t : ((long long)clip * T + t) * rows;
            for (int r = tid; r < rows; r += NT) {
                double v;
                if (chroma_res > 0) {   // zaf.py:693-698: rows i, i + r, i + 2r, ... summed in ascending order
                    v = 0.0;
                    for (int q = r; q < n_bins; q += chroma_res) v += spec[q];
                } else {
                    v = spec[r];
                }
                out[base + r * stride] = v;
            }
        }
        lds_barrier();   // the compact spectrum lies in the transform buffers the next frame fills
        PROF_MARK(7);
    }
}
}  // namespace

// Host side of k_mel_ft8_f64: the filterbank's bands as rows of 64 segments (zafx_mel64.hpp; see the kernel's header).
hipError_t build_mel64_fb(zafx_plan& pl) {
    pl.mel64_ok = false;
    const int nf = pl.prm.n_filters, cols = pl.W / 2;
    if (!ZAFX_F64_TILED || pl.W != 2 * kF64N || pl.bs_log2m > 0 || nf < 1 || nf > kMel64Rows || pl.h_fb64.size() != (size_t)nf * cols) return hipSuccess;
    const Mel64Tables t = mel64_tables(pl.h_fb64.data(), nf, cols, kMel64Spare);
    if (!t.ok) return hipSuccess;   // (the frame-per-workgroup kernel takes such a matrix)
    auto up = [](auto** d, const void* h, size_t bytes) -> hipError_t {
        if (*d) (void)hipFree(*d);
        *d = nullptr;
        if (hipError_t e = hipMalloc((void**)d, bytes); e != hipSuccess) return e;
        return hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice);
    };
    static_assert(sizeof(Mel64Entry) == sizeof(int4), "an entry is one 16-byte load");
    if (hipError_t e = up(&pl.d_mel64_stream, t.stream.data(), t.stream.size() * sizeof(Mel64Entry)); e != hipSuccess) return e;
    if (hipError_t e = up(&pl.d_mel64_fin, t.fin.data(), t.fin.size() * sizeof(int)); e != hipSuccess) return e;
    pl.mel64_steps = t.steps;
    pl.mel64_slots = t.slots;
    pl.mel64_max_parts = t.max_parts;
    pl.mel64_ok = true;
    return hipSuccess;
}

hipError_t build_mel64_dct(zafx_plan& pl) {   // DCT-II rows transposed and padded: [2 dct_half filters][coefficients up to a multiple of 32], zeros outside
    const int nf = pl.prm.n_filters, nc = pl.prm.n_coefs;
    pl.mel64_cpitch = 0;
    if (nc < 1 || nc > kMel64Rows || nf < 1 || pl.h_dct64.size() != (size_t)nc * nf) return hipSuccess;
    const int cp = (nc + 31) / 32 * 32, half = ((nf + 1) / 2 + 15) / 16 * 16;
    std::vector<double> t((size_t)2 * half * cp, 0.0);
    for (int c = 0; c < nc; ++c)
        for (int n = 0; n < nf; ++n) t[(size_t)n * cp + c] = pl.h_dct64[(size_t)c * nf + n];
    if (pl.d_mel64_dctT) (void)hipFree(pl.d_mel64_dctT);
    pl.d_mel64_dctT = nullptr;
    if (hipError_t e = hipMalloc((void**)&pl.d_mel64_dctT, t.size() * sizeof(double)); e != hipSuccess) return e;
    if (hipError_t e = hipMemcpy(pl.d_mel64_dctT, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice); e != hipSuccess) return e;
    pl.mel64_cpitch = cp;
    pl.mel64_dct_half = half;
    return hipSuccess;
}

// Host side of k_cqt_ft_f64 (zafx_cqt64.hpp): rebuilt by launch_cqt_f64 whenever the kernel matrix changed
constexpr int kCq64MaxKc2 = 4;
static hipError_t build_cqt64(zafx_plan& pl) {
    pl.cqt64_ok = false;
    pl.cqt64_dirty = false;
    const int rows = (int)pl.h_indptr.size() - 1;
    if (!ZAFX_F64_TILED || pl.W != 32768 || rows < 1 || (int)pl.h_values64.size() != pl.nnz || pl.h_indptr.back() != pl.nnz) return hipSuccess;
    const Cqt64Tables t = cqt64_tables(pl.h_indptr.data(), pl.h_indices.data(), reinterpret_cast<const double*>(pl.h_values64.data()), rows, pl.W, kCq64MaxKc2);
    if (!t.ok || (size_t)t.n_cols * 16 + (size_t)t.slots * 16 + (size_t)rows * 8 > (size_t)kCq64Threads / 64 * kF64Pitch * sizeof(double2)) return hipSuccess;
    // real up to round-off?  (then the imaginary parts move the result by less than 1e-13 of it: dropped, 12 bytes per entry instead of 24)
    double vmax = 0.0, imax = 0.0;
    for (const Cqt64Entry& en : t.stream) vmax = std::max(vmax, std::fabs(en.re)), imax = std::max(imax, std::fabs(en.im));
    pl.cqt64_real = imax <= vmax * 0x1p-44;
    std::vector<double> vals;
    std::vector<int> meta(t.stream.size());
    if (t.slots + 1 >= (1 << 17) || t.n_cols > 0x1fff) return hipSuccess;
    for (size_t i = 0; i < t.stream.size(); ++i) {
        const Cqt64Entry& en = t.stream[i];
        vals.push_back(en.re);
        if (!pl.cqt64_real) vals.push_back(en.im);
        meta[i] = (en.index & 0x1fff) | (en.index < 0 ? 0x2000 : 0) | ((en.slot + 1) << 14);   // compact index | conjugate << 13 | (slot + 1) << 14
    }
    auto up = [](auto** d, const void* h, size_t bytes) -> hipError_t {
        if (*d) (void)hipFree(*d);
        *d = nullptr;
        if (hipError_t e = hipMalloc((void**)d, bytes); e != hipSuccess) return e;
        return hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice);
    };
    std::vector<double2> tw1((size_t)4 * 1024);   // tw1[i][m] = exp(-2 pi i m 2^i / 16384), in long double
    for (int i = 0; i < 4; ++i)
        for (int m = 0; m < 1024; ++m) {
            const long long num = ((long long)m << i) % 16384;
            const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)num / 16384.0L;
            tw1[(size_t)i * 1024 + m] = num == 0 ? make_double2(1.0, 0.0) : num == 4096 ? make_double2(0.0, -1.0) : num == 8192 ? make_double2(-1.0, 0.0)
                                        : num == 12288 ? make_double2(0.0, 1.0) : make_double2((double)cosl(a), (double)sinl(a));
        }
    if (hipError_t e = up(&pl.d_cqt64_tw1, tw1.data(), tw1.size() * sizeof(double2)); e != hipSuccess) return e;
    if (hipError_t e = up(&pl.d_cqt64_split, t.split.data(), t.split.size() * sizeof(int)); e != hipSuccess) return e;
    if (hipError_t e = up(&pl.d_cqt64_vals, vals.data(), vals.size() * sizeof(double)); e != hipSuccess) return e;
    if (hipError_t e = up(&pl.d_cqt64_meta, meta.data(), meta.size() * sizeof(int)); e != hipSuccess) return e;
    if (hipError_t e = up(&pl.d_cqt64_fin, t.fin.data(), t.fin.size() * sizeof(int)); e != hipSuccess) return e;
    pl.cqt64_kc2 = t.kc2;
    pl.cqt64_cols = t.n_cols;
    pl.cqt64_steps = t.steps;
    pl.cqt64_slots = t.slots;
    pl.cqt64_max_parts = t.max_parts;
    int max_col = 0;
    for (int i = 0; i < pl.nnz; ++i) max_col = std::max(max_col, pl.h_indices[(size_t)i]);
    pl.cqt64_klo = max_col >> 4;   // (columns 1 .. 8191: klo <= 511; the mirrors N - c lie at k >= 1023 - klo)
    pl.cqt64_ok = true;
    return hipSuccess;
}

const char* cqt_f64_kernel_name() { return "k_cqt_f64"; }
const char* mel_f64_kernel_name() { return "k_mel_f64"; }
const char* mdct_f64_kernel_name() { return "k_mdct_f64"; }
const char* imdct_f64_kernel_name() { return "k_imdct_frames_f64"; }
const char* stft_f64_kernel_name() { return "k_stft_f64"; }
const char* istft_f64_kernel_name() { return "k_ifft_frames_f64"; }

// Workgroup size of a frame kernel by its LDS image: the occupancy of a CU comes from several small workgroups, or --
// when one frame fills LDS -- from one big one
static int threads_for(size_t smem) { return smem > 80 * 1024 ? kThreadsBig : smem > 40 * 1024 ? 512 : kThreads; }

// grow-only scratch of time-domain frames, owned by the plan
hipError_t grow_scratch(zafx_plan& pl, size_t need) {
    if (need <= pl.scratch_bytes) return hipSuccess;
    if (hipError_t e = hipStreamSynchronize(pl.stream); e != hipSuccess) return e;
    if (pl.d_scratch64) (void)hipFree(pl.d_scratch64);
    pl.d_scratch64 = nullptr;
    pl.scratch_bytes = 0;
    if (hipError_t e = hipMalloc((void**)&pl.d_scratch64, need); e != hipSuccess) return e;
    pl.scratch_bytes = need;
    return hipSuccess;
}

// STFT / mel / mfcc of a window that is not a power of two
static hipError_t launch_bs_f64(const zafx_plan& pl, const double* x, void* out, int64_t n_clips, int64_t n_samples, int T, bool mel_mode) {
    const long long blocks = (long long)n_clips * T;
    if (blocks <= 0) return hipSuccess;
    const size_t smem = ((size_t)2 << pl.bs_log2m) * sizeof(double2);
    auto kern = k_stft_bs_f64;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, smem); e != hipSuccess) return e;
    pl.ran = "k_stft_bs_f64";
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(threads_for(smem)), smem, pl.stream, x, pl.d_window64, pl.d_tw64, pl.d_tws64, pl.d_bhat64,
                       pl.d_fb64, pl.d_fb64_meta, pl.d_dct64, (double2*)out, (long long)n_samples, pl.H, T, (int)row_pitch(pl, T), pl.W,
                       pl.bs_log2m, pl.layout, pl.prm.spectrum, pl.prm.n_filters, pl.kind == ZAFX_MFCC ? pl.prm.n_coefs : 0, mel_mode ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_stft_f64(const zafx_plan& pl, const double* x, double2* out, int64_t n_clips, int64_t n_samples, int T) {
    if (pl.bs_log2m > 0) return launch_bs_f64(pl, x, out, n_clips, n_samples, T, false);
    const long long blocks = (long long)n_clips * T;
    if (blocks <= 0) return hipSuccess;
    if (ZAFX_F64_TILED && pl.W == 2048 && pl.layout == ZAFX_LAYOUT_FT && pl.prm.spectrum <= ZAFX_SPECTRUM_ONE_SIDED && reinterpret_cast<uintptr_t>(x) % 16 == 0 &&
        reinterpret_cast<uintptr_t>(out) % 16 == 0) {
        const int tiles = (T + kF64Frames - 1) / kF64Frames;
        const long long total = (long long)tiles * n_clips;
        if (total < (1LL << 31)) {
            const size_t smem8 = (size_t)kF64Frames * kF64Pitch * sizeof(double2) + kF64Frames * sizeof(double);
            const bool one = pl.prm.spectrum == ZAFX_SPECTRUM_ONE_SIDED;
            auto k8 = one ? k_stft_ft8_f64<true> : k_stft_ft8_f64<false>;
            if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(k8), pl.device, smem8); e != hipSuccess) return e;
            pl.ran = "k_stft_ft8_f64";
            hipLaunchKernelGGL(k8, dim3((unsigned)std::min<long long>(total, (long long)pl.n_cus * (kF64Frames <= 4 ? 2 : 1))), dim3(kF64Frames * 64), smem8, pl.stream, x, pl.d_window64, pl.d_tw64,
                               pl.d_tws64, out, (long long)n_samples, pl.H, T, (int)row_pitch(pl, T), tiles, (int)total);
            return hipGetLastError();
        }
    }
    const size_t smem = (size_t)pl.W * sizeof(double2);   // two buffers of W/2 points
    auto kern = k_stft_f64;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, smem); e != hipSuccess) return e;
    pl.ran = "k_stft_f64";
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(threads_for(smem)), smem, pl.stream, x, pl.d_window64, pl.d_tw64, pl.d_tws64, out,
                       (long long)n_samples, pl.H, T, (int)row_pitch(pl, T), pl.log2nf, pl.layout, pl.prm.spectrum);
    return hipGetLastError();
}

// The inverse transforms park every time-domain frame of the batch in a plan-owned scratch (n_clips x T x W doubles) before
// the gather overlap-add; large batches go through it in chunks of clips so that the scratch stays below scratch_budget()
// (1024 clips x 432 frames x 2048 samples would be 7 GB, pinned by the cached plan until it is destroyed).
static size_t scratch_budget() {   // 1 GiB; ZAFX_SCRATCH_BUDGET_MB overrides it (tests force the chunked path with it)
    static const size_t budget = [] {
        const char* env = std::getenv("ZAFX_SCRATCH_BUDGET_MB");
        const long mb = env ? std::atol(env) : 0;
        return mb > 0 ? (size_t)mb << 20 : (size_t)1 << 30;
    }();
    return budget;
}
int64_t scratch_clips_per_chunk(int64_t n_clips, int T, int W, size_t elem_bytes) {
    const size_t per_clip = (size_t)T * (size_t)W * elem_bytes;
    return std::max<int64_t>(1, std::min<int64_t>(n_clips, (int64_t)(scratch_budget() / std::max<size_t>(per_clip, 1))));
}
static int64_t clips_per_chunk(int64_t n_clips, int T, int W) { return scratch_clips_per_chunk(n_clips, T, W, sizeof(double)); }

hipError_t launch_istft_f64(zafx_plan& pl, const double2* spec_all, double* y_all, int64_t n_clips_all, int T, int64_t out_len) {
    if ((long long)n_clips_all * T <= 0 || out_len <= 0) return hipSuccess;
    if (ZAFX_F64_TILED && pl.W == 2048 && pl.layout == ZAFX_LAYOUT_FT && pl.bs_log2m == 0 && 2 * pl.H >= pl.W && pl.H <= pl.W &&
        (long long)T * pl.H < (1LL << 40)) {
        const int tiles = (T + kF64Frames - 1) / kF64Frames;
        const int segs = carry_segments(n_clips_all, tiles, pl.n_cus);
        const int seg_tiles = (tiles + segs - 1) / segs;
        const long long units = (long long)n_clips_all * segs;
        if (units < (1LL << 31)) {
            const size_t smem = (size_t)kIst64Slots * kF64Pitch * sizeof(double2);
            const bool one = pl.prm.spectrum == ZAFX_SPECTRUM_ONE_SIDED;
            auto kern = one ? k_istft_ft8_f64<true> : k_istft_ft8_f64<false>;
            if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, smem); e != hipSuccess) return e;
            const double scale = 1.0 / (2.0 * (double)pl.W * pl.cola_gain64);   // 1/W of the inverse DFT x the factor 2 left by the fold
            pl.ran = "k_istft_ft8_f64";
            hipLaunchKernelGGL(kern, dim3((unsigned)std::min<long long>(units, pl.n_cus)), dim3(kF64Frames * 64), smem, pl.stream, spec_all, pl.d_tw64, pl.d_tws64, y_all,
                               T, (int)row_pitch(pl, T), pl.H, (long long)out_len, scale, tiles, segs, seg_tiles, (int)units);
            return hipGetLastError();
        }
    }
    const int64_t chunk = clips_per_chunk(n_clips_all, T, pl.W);
    if (hipError_t e = grow_scratch(pl, (size_t)chunk * T * pl.W * sizeof(double)); e != hipSuccess) return e;
    const int64_t rows = pl.prm.spectrum == ZAFX_SPECTRUM_ONE_SIDED ? pl.W / 2 + 1 : pl.W;
    const int64_t in_per_clip = pl.layout == ZAFX_LAYOUT_FT ? rows * row_pitch(pl, T) : (int64_t)T * rows;
    for (int64_t c0 = 0; c0 < n_clips_all; c0 += chunk) {
        const int64_t n_clips = std::min(chunk, n_clips_all - c0);
        const double2* spec = spec_all + c0 * in_per_clip;
        double* y = y_all + c0 * out_len;
        const long long blocks = (long long)n_clips * T;
        const long long total = (long long)n_clips * out_len;
        const long long grid = std::min<long long>((total + kThreads - 1) / kThreads, (long long)pl.n_cus * 32);
        if (pl.bs_log2m > 0) {   // window that is not a power of two
            const size_t smem = ((size_t)2 << pl.bs_log2m) * sizeof(double2);
            auto kern = k_ifft_frames_bs_f64;
            if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, smem); e != hipSuccess) return e;
            pl.ran = "k_ifft_frames_bs_f64";
            hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(threads_for(smem)), smem, pl.stream, spec, pl.d_tw64, pl.d_tws64, pl.d_bhat64,
                               pl.d_scratch64, T, (int)row_pitch(pl, T), pl.W, pl.bs_log2m, pl.layout,
                               pl.prm.spectrum == ZAFX_SPECTRUM_ONE_SIDED ? 1 : 0);
            if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
            hipLaunchKernelGGL(k_ola_f64, dim3((unsigned)grid), dim3(kThreads), 0, pl.stream, pl.d_scratch64, y, T, pl.W, pl.H,
                               (long long)out_len, total, 1.0 / pl.cola_gain64);
        } else {
            const size_t smem = (size_t)pl.W * sizeof(double2);
            auto kern = k_ifft_frames_f64;
            if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, smem); e != hipSuccess) return e;
            pl.ran = "k_ifft_frames_f64";
            hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(threads_for(smem)), smem, pl.stream, spec, pl.d_tw64, pl.d_tws64, pl.d_scratch64, T,
                               (int)row_pitch(pl, T), pl.log2nf, pl.layout, pl.prm.spectrum == ZAFX_SPECTRUM_ONE_SIDED ? 1 : 0);
            if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
            const double scale = 1.0 / (2.0 * (double)pl.W * pl.cola_gain64);   // 1/W of the inverse DFT x the factor 2 left by the fold
            hipLaunchKernelGGL(k_ola_f64, dim3((unsigned)grid), dim3(kThreads), 0, pl.stream, pl.d_scratch64, y, T, pl.W, pl.H, (long long)out_len,
                               total, scale);
        }
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_cqt_f64(zafx_plan& pl, const double* x, double* out, int64_t n_clips, int64_t n_samples, int T) {
    const long long total = (long long)n_clips * T;
    if (total <= 0) return hipSuccess;
    if (pl.cqt64_dirty)
        if (hipError_t e = build_cqt64(pl); e != hipSuccess) return e;
    if (pl.cqt64_ok && reinterpret_cast<uintptr_t>(x) % 16 == 0 && total < (1LL << 31)) {
        const int n_groups = (int)std::min<int64_t>(8, n_clips);
        const long long per_group = ((n_clips + n_groups - 1) / n_groups) * (long long)T;   // frames of the longest list
        const int grid = (int)std::min<long long>(std::max(pl.n_cus / n_groups, 1) * (long long)n_groups, per_group * n_groups);
        const size_t smem = (size_t)(kCq64Threads / 64) * kF64Pitch * sizeof(double2) + (256 + 768) * sizeof(double2);
        const int diff = pl.W - pl.H;
        const int left = diff >= 0 ? (diff + 1) / 2 : -((-diff) / 2);   // ceil((fft_length - step) / 2), zaf.py:615
        auto pick = [&](auto full) {
            constexpr bool F = decltype(full)::value;
            return pl.cqt64_real ? (pl.cqt64_kc2 <= 1 ? k_cqt_ft_f64<1, true, F> : pl.cqt64_kc2 == 2 ? k_cqt_ft_f64<2, true, F> : pl.cqt64_kc2 == 3 ? k_cqt_ft_f64<3, true, F> : k_cqt_ft_f64<4, true, F>)
                                 : (pl.cqt64_kc2 <= 1 ? k_cqt_ft_f64<1, false, F> : pl.cqt64_kc2 == 2 ? k_cqt_ft_f64<2, false, F> : pl.cqt64_kc2 == 3 ? k_cqt_ft_f64<3, false, F> : k_cqt_ft_f64<4, false, F>);
        };
        auto kern = pl.cqt64_klo >= 256 ? pick(std::true_type{}) : pick(std::false_type{});   // (FULL: the last pass of the sub-transforms writes every quarter)
        const int kc2 = pl.cqt64_kc2 <= 3 ? std::max(pl.cqt64_kc2, 1) : 4;
        if (kc2 != pl.cqt64_kc2) return hipErrorInvalidValue;   // (the split table is laid out for the plan's own count)
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, smem); e != hipSuccess) return e;
        pl.ran = "k_cqt_ft_f64";
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kCq64Threads), smem, pl.stream, x, pl.d_tws64, pl.d_cqt64_tw1, pl.d_cqt64_split, pl.d_cqt64_vals, pl.d_cqt64_meta,
                           pl.d_cqt64_fin, out, (long long)n_samples, pl.H, left, T, (int)row_pitch(pl, T), (int)n_clips, n_groups, pl.prm.n_bins,
                           pl.kind == ZAFX_CHROMA ? pl.prm.octave_resolution : 0, pl.layout, pl.cqt64_cols, pl.cqt64_steps, pl.cqt64_slots, pl.cqt64_max_parts);
        return hipGetLastError();
    }
    const int log2w = pl.log2nf + 1;
    const int n2 = std::min(pl.W, kCqt64Sub);
    const long long grid = std::min<long long>(total, (long long)pl.n_cus * 2);
    if (pl.W > n2) {
        if (hipError_t e = grow_scratch(pl, (size_t)grid * pl.W * sizeof(double2)); e != hipSuccess) return e;
    }
    const size_t smem = (size_t)n2 * 2 * sizeof(double2) + (size_t)pl.prm.n_bins * sizeof(double);
    auto kern = k_cqt_f64;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, smem); e != hipSuccess) return e;
    const int diff = pl.W - pl.H;
    const int left = diff >= 0 ? (diff + 1) / 2 : -((-diff) / 2);   // ceil((fft_length - step) / 2), zaf.py:615
    pl.ran = "k_cqt_f64";
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(threads_for(smem)), smem, pl.stream, x, pl.d_tw64, pl.d_tws64, pl.d_indptr, pl.d_indices,
                       pl.d_values64, reinterpret_cast<double2*>(pl.d_scratch64), out, (long long)n_samples, pl.H, left, T,
                       (int)row_pitch(pl, T), total, log2w, pl.prm.n_bins, pl.kind == ZAFX_CHROMA ? pl.prm.octave_resolution : 0, pl.layout);
    return hipGetLastError();
}

hipError_t launch_mel_f64(const zafx_plan& pl, const double* x, double* out, int64_t n_clips, int64_t n_samples, int T) {
    if (pl.bs_log2m > 0) return launch_bs_f64(pl, x, out, n_clips, n_samples, T, true);
    const long long blocks = (long long)n_clips * T;
    if (blocks <= 0) return hipSuccess;
    const bool mfcc = pl.kind == ZAFX_MFCC;
    if (ZAFX_F64_TILED && pl.mel64_ok && (!mfcc || (pl.mel64_cpitch > 0 && pl.mel64_slots + 2 * pl.mel64_dct_half <= kMel64Spare)) && pl.layout == ZAFX_LAYOUT_FT && reinterpret_cast<uintptr_t>(x) % 16 == 0) {
        const int tiles = (T + 15) / 16;
        const long long total = (long long)tiles * n_clips;
        if (total < (1LL << 31)) {
            const size_t smem8 = (size_t)kF64Frames * kF64Pitch * sizeof(double2) + (size_t)kMel64Rows * kMel64OutPitch * sizeof(double);
            auto k8 = mfcc ? k_mel_ft8_f64<true> : k_mel_ft8_f64<false>;
            if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(k8), pl.device, smem8); e != hipSuccess) return e;
            pl.ran = "k_mel_ft8_f64";
            hipLaunchKernelGGL(k8, dim3((unsigned)std::min<long long>(total, pl.n_cus)), dim3(kF64Frames * 64), smem8, pl.stream, x, pl.d_window64, pl.d_tw64, pl.d_tws64,
                               pl.d_mel64_stream, pl.d_mel64_fin, pl.d_mel64_dctT, out, (long long)n_samples, pl.H, T,
                               (int)row_pitch(pl, T), tiles, (int)total, pl.prm.n_filters, mfcc ? pl.prm.n_coefs : 0, pl.mel64_steps, pl.mel64_slots,
                               pl.mel64_max_parts, pl.mel64_cpitch, pl.mel64_dct_half);
            return hipGetLastError();
        }
    }
    const size_t smem = (size_t)pl.W * sizeof(double2);   // two buffers of W/2 points; the idle one later holds bins + band sums
    auto kern = k_mel_f64;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, smem); e != hipSuccess) return e;
    pl.ran = "k_mel_f64";
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(threads_for(smem)), smem, pl.stream, x, pl.d_window64, pl.d_tw64, pl.d_tws64, pl.d_fb64,
                       pl.d_fb64_meta, pl.d_dct64, out, (long long)n_samples, pl.H, T, (int)row_pitch(pl, T), pl.log2nf, pl.layout,
                       pl.prm.n_filters, pl.kind == ZAFX_MFCC ? pl.prm.n_coefs : 0);
    return hipGetLastError();
}

hipError_t launch_mdct_f64(const zafx_plan& pl, const double* x, double* out, int64_t n_clips, int64_t n_samples, int T) {
    const long long blocks = (long long)n_clips * T;
    if (blocks <= 0) return hipSuccess;
    if (pl.bs_log2m > 0) {   // window that is not a power of two
        const size_t smem = ((size_t)2 << pl.bs_log2m) * sizeof(double2);
        auto kern = k_mdct_bs_f64;
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, smem); e != hipSuccess) return e;
        pl.ran = "k_mdct_bs_f64";
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(threads_for(smem)), smem, pl.stream, x, pl.d_window64, pl.d_tw64, pl.d_tws64, pl.d_bhat64,
                           out, (long long)n_samples, T, (int)row_pitch(pl, T), pl.W, pl.bs_log2m, pl.layout);
        return hipGetLastError();
    }
    if (ZAFX_F64_TILED && pl.W == 2048 && pl.layout == ZAFX_LAYOUT_FT) {
        const int tiles = (T + kMd64Frames - 1) / kMd64Frames;
        const long long total = (long long)tiles * n_clips;
        if (total < (1LL << 31)) {
            const size_t smem16 = (size_t)kMd64Frames * kMd64Pitch * sizeof(double2);
            auto k16 = k_mdct_ft16_f64;
            if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(k16), pl.device, smem16); e != hipSuccess) return e;
            pl.ran = "k_mdct_ft16_f64";
            hipLaunchKernelGGL(k16, dim3((unsigned)std::min<long long>(total, pl.n_cus)), dim3(kMd64Frames * 64), smem16, pl.stream, x, pl.d_window64, pl.d_tw64,
                               pl.d_tws64, out, (long long)n_samples, T, (int)row_pitch(pl, T), tiles, (int)total);
            return hipGetLastError();
        }
    }
    const size_t smem = (size_t)pl.W * 8 + (size_t)(pl.W / 2) * 8 + (size_t)(pl.W / 4) * 32;   // u, v, two FFT buffers
    auto kern = k_mdct_f64;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, smem); e != hipSuccess) return e;
    pl.ran = "k_mdct_f64";
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(threads_for(smem)), smem, pl.stream, x, pl.d_window64, pl.d_tw64, pl.d_tws64, out,
                       (long long)n_samples, T, (int)row_pitch(pl, T), pl.log2nf, pl.layout);
    return hipGetLastError();
}

hipError_t launch_imdct_f64(zafx_plan& pl, const double* coefs_all, double* y_all, int64_t n_clips_all, int T, int64_t out_len) {
    if ((long long)n_clips_all * T <= 0 || out_len <= 0) return hipSuccess;
    if (ZAFX_F64_TILED && pl.W == 2048 && pl.layout == ZAFX_LAYOUT_FT && pl.bs_log2m == 0) {
        const int tiles = (T + kMd64Frames - 1) / kMd64Frames;
        const int segs = carry_segments(n_clips_all, tiles, pl.n_cus);
        const int seg_tiles = (tiles + segs - 1) / segs;
        const long long units = (long long)n_clips_all * segs;
        if (units < (1LL << 31)) {
            const size_t smem = (size_t)kImd64Slots * kMd64Pitch * sizeof(double2);
            auto kern = k_imdct_ft16_f64;
            if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, smem); e != hipSuccess) return e;
            pl.ran = "k_imdct_ft16_f64";
            hipLaunchKernelGGL(kern, dim3((unsigned)std::min<long long>(units, pl.n_cus)), dim3(kMd64Frames * 64), smem, pl.stream, coefs_all, pl.d_window64, pl.d_tw64,
                               pl.d_tws64, y_all, T, (int)row_pitch(pl, T), (long long)out_len, tiles, segs, seg_tiles, (int)units);
            return hipGetLastError();
        }
    }
    const int64_t chunk = clips_per_chunk(n_clips_all, T, pl.W);   // (scratch budget: see launch_istft_f64)
    if (hipError_t e = grow_scratch(pl, (size_t)chunk * T * pl.W * sizeof(double)); e != hipSuccess) return e;
    const int64_t in_per_clip = pl.layout == ZAFX_LAYOUT_FT ? (int64_t)(pl.W / 2) * row_pitch(pl, T) : (int64_t)T * (pl.W / 2);
    for (int64_t c0 = 0; c0 < n_clips_all; c0 += chunk) {
        const int64_t n_clips = std::min(chunk, n_clips_all - c0);
        const double* coefs = coefs_all + c0 * in_per_clip;
        double* y = y_all + c0 * out_len;
        const long long blocks = (long long)n_clips * T;
        if (pl.bs_log2m > 0) {   // window that is not a power of two
            const size_t smem = ((size_t)2 << pl.bs_log2m) * sizeof(double2);
            auto kern = k_imdct_frames_bs_f64;
            if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, smem); e != hipSuccess) return e;
            pl.ran = "k_imdct_frames_bs_f64";
            hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(threads_for(smem)), smem, pl.stream, coefs, pl.d_window64, pl.d_tw64, pl.d_tws64,
                               pl.d_bhat64, pl.d_scratch64, T, (int)row_pitch(pl, T), pl.W, pl.bs_log2m, pl.layout);
        } else {
            const size_t smem = (size_t)(pl.W / 2) * 16 + (size_t)(pl.W / 4) * 32;
            auto kern = k_imdct_frames_f64;
            if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, smem); e != hipSuccess) return e;
            pl.ran = "k_imdct_frames_f64";
            hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(threads_for(smem)), smem, pl.stream, coefs, pl.d_window64, pl.d_tw64, pl.d_tws64,
                               pl.d_scratch64, T, (int)row_pitch(pl, T), pl.log2nf, pl.layout);
        }
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
        // two-frame TDAC overlap-add in ascending frame order (zaf.py:1172-1179) and the trim [H : -H-1] (:1182): the same
        // gather as the ISTFT's with hop = W/2
        const long long total = (long long)n_clips * out_len;
        const long long grid = std::min<long long>((total + kThreads - 1) / kThreads, (long long)pl.n_cus * 32);
        hipLaunchKernelGGL(k_ola_f64, dim3((unsigned)grid), dim3(kThreads), 0, pl.stream, pl.d_scratch64, y, T, pl.W, pl.H, (long long)out_len,
                           total, 1.0);
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace zafx
ZAFX_PROF_EXPORT(zafx_debug_prof_mel64, g_prof_mel64)
ZAFX_PROF_EXPORT(zafx_debug_prof_cqt64, g_prof_cqt64)
