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
#include <type_traits>

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
    const float2* __restrict__ tw8, float* __restrict__ out, long long n_samples, int T, int TP, int tiles) {
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
    const int bid = ZAFX_XCD_ORDER ? xcd_order((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;   // neighbouring tiles to one XCD: their partial lines merge in its L2
    const int clip = bid / tiles, tile = bid % tiles;
    const int t0 = tile * FPB;
    const int t = t0 + slot;
    float2* buf = frames + slot * C::PITCH;
    float* u = stage_all + slot * G::STAGE;

    // ---- stage the windowed frame (coalesced): u[n] = xpad[t M + n] w[n], left pad = M (zaf.py:1036-1064)
    {
        const float* xc = x + (long long)clip * n_samples;
        const long long s0 = (long long)t * M - M;
        if (W % 4 == 0 && t < T && s0 >= 0 && s0 + W <= n_samples && reinterpret_cast<uintptr_t>(xc + s0) % 16 == 0 &&
            reinterpret_cast<uintptr_t>(win) % 16 == 0) {   // interior frame: 16-byte loads, no per-sample bounds tests
            const float4* x4 = reinterpret_cast<const float4*>(xc + s0);
            const float4* w4 = reinterpret_cast<const float4*>(win);
            float4* u4 = reinterpret_cast<float4*>(u);
            for (int n = p; n < W / 4; n += P) {
                const float4 a = x4[n], b = w4[n];
                u4[n] = make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
            }
        } else {
            for (int n = p; n < W; n += P) {
                const long long s = s0 + n;
                u[n] = (t < T && s >= 0 && s < n_samples) ? xc[s] * win[n] : 0.f;
            }
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
        float* o = out + (long long)clip * M * TP + (t0 + tt);   // TP = row pitch (>= T)
        for (int f = fq; f < M; f += P) {
            const int k = (f & 1) ? (M - 1 - f) >> 1 : f >> 1;
            const float2 yk = cmul(fb[phys(k)], tw8[k]);
            o[(long long)f * TP] = (f & 1) ? -yk.y : yk.x;
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
ZAFX_PROF_ARRAY(g_prof_mdct)
ZAFX_PROF_ARRAY(g_prof_imdct)

template <int LOG2NF, int LOG2E>
struct MdctPCfg {
    using C = FftCfg<LOG2NF, LOG2E>;
    static_assert(C::P == 64, "one wavefront per frame");
    static constexpr int NF = C::N;
    static constexpr int NSLOT = 16;  // 16 waves x 2 frames per tile: the LDS round trips of the short passes need the occupancy
    static constexpr size_t SMEM = (size_t)(kMdctTile * C::PITCH + C::TW + NF) * 8 + (size_t)NF * 16;
};

// TFOUT = frame-major output (ZAFX_LAYOUT_TF): a frame's M coefficients are contiguous, so the wave that transformed a
// frame also stores it (512-B coalesced runs) and the workgroup never meets after the table staging.
// CARRY (reference layout, even T): rows that are not whole lines or half lines (T % 16 != 0) are completed from a register carry
// of the previous tile, as in k_stft_ft16c: the
// workgroup walks the tiles of a clip segment in order, thread (frame pair tp, rows fq + 64 i) keeps its own 16 pairs of the
// previous tile (32 VGPRs, instead of the resident window quadruples), and for a row whose run starts `a` floats into a line the
// lanes tp < 16 - a / 2 store the current pair at frame t0 + 2 tp, the others the carried pair at frame t0 - 32 + 2 tp: sixteen
// lanes, one whole line.  segs / seg_tiles / units: the segment walk (carry_segments).
// PCM (ALIGNED forms): `x` holds int16 -- 1: mono, n_samples 2-byte samples per clip, a 4-sample piece is 8 bytes; 2: stereo, n_samples 4-byte frames,
// a piece is the float32 form's 16 bytes and both channels are added -- normalised (zaf.py:1202) and averaged (zaf.py:65) on the way into the fold:
// the power of two rides in the window quadruples (zafx_execute_pcm).
template <int LOG2NF, int LOG2E, bool ALIGNED, int NSLOT, bool TFOUT = false, bool CARRY = false, int PCM = 0>
__global__ __launch_bounds__(NSLOT * 64) void k_mdct_ft32(
    const float* __restrict__ x, const float4* __restrict__ wfold, const float2* __restrict__ twp,
    const float2* __restrict__ tw8, float* __restrict__ out, long long n_samples, int T, int TP, int tiles, int total_tiles,
    int segs = 1, int seg_tiles = 0, int units = 0) {
    static_assert(!(CARRY && TFOUT), "the carry form is for the reference layout");
    using C = FftCfg<LOG2NF, LOG2E>;
    constexpr int NF = C::N, M = 2 * NF, P = C::P, E = C::E, FPB = kMdctTile, NT = NSLOT * P;
    constexpr int FPW = FPB / NSLOT;                    // frames per wave and tile
    constexpr int NU = NF / 4;                          // 16-sample groups per frame (see fold below)
    constexpr int UPL = NU >= P ? NU / P : 1;           // groups per lane
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* frames = reinterpret_cast<float2*>(smem_raw);
    float2* tw_l = frames + FPB * C::PITCH;
    float2* g_l = tw_l + C::TW;
    float4* wf_l = reinterpret_cast<float4*>(g_l + NF);
    const int tid = threadIdx.x;
    for (int i = tid; i < C::TW; i += NT) tw_l[i] = twp[i];
    static_assert(PCM == 0 || ALIGNED, "int16 input: the buffer-load form");
    const float pcm_scale = PCM == 1 ? 1.f / 32768.f : PCM == 2 ? 1.f / 65536.f : 1.f;
    for (int i = tid; i < NF; i += NT) {
        g_l[i] = tw8[i];
        const float4 wv = wfold[i];
        wf_l[i] = make_float4(wv.x * pcm_scale, wv.y * pcm_scale, wv.z * pcm_scale, wv.w * pcm_scale);
    }
    lds_barrier();
    const int slot = tid / P, p = tid % P;
    const bool pair_ok = (TFOUT || TP % 2 == 0) && (reinterpret_cast<uintptr_t>(out) % 8 == 0);   // TP = row pitch (>= T)
    const bool lines_whole = TP % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 64 == 0;   // rows of whole 64-B half lines: stream them
    const bool lane_loads = p < NU;

    // A frame's W = 4 NF samples are fetched as 16-byte pieces.  Group u takes four of them,
    //   A3 = x[3NF+4u ..+3]   R2 = x[3NF-4-4u ..+3]   A1 = x[NF+4u ..+3]   R0 = x[NF-4-4u ..+3],
    // exactly the 16 samples that the TDAC fold needs for the four outputs m = 2u, 2u+1, NF-1-2u,
    // NF-2-2u (every sample of the frame is used once), so the loads are coalesced 1-KB rows per wave
    // instead of 4-byte gathers.  The next frame of the wave is requested as soon as the current one
    // is folded: its latency hides under this frame's FFT (or, across tiles, under the store phase).
    // Wave w owns frames w and w + NSLOT of the tile (rounds): the frames the waves fetch at the same time are
    // neighbours, so the half frames they share meet in L1 / L2.  (With adjacent pairs (2w, 2w+1) per wave the PMC pass
    // showed every sample fetched twice from the fabric, 3.7 GB for 1.8 GB of input; reusing the shared half from the
    // wave's registers recovered part of it, 1.075 ms -- rounds need no such code and run 1.05 ms.)
    auto frame_of = [&](int f) { return f * NSLOT + slot; };
    float4 q[UPL][4];
    auto fetch = [&](int tl, int f) {
        if (tl >= total_tiles || !lane_loads) return;
        const int clip = tl / tiles, tile = tl % tiles;
        const int t = tile * FPB + frame_of(f);
        const float* xc = x + (long long)clip * n_samples;
        const long long s0 = (long long)t * M - M;   // left pad = M (zaf.py:1036-1041)
        if constexpr (ALIGNED) {
            // Buffer loads through a descriptor of the clip: a 16-byte piece that lies before the clip's first or after its last
            // sample is out of the descriptor's range and reads as zero -- the zero padding of zaf.py:1036-1041, with no edge
            // path (n_samples and every piece's first sample are multiples of 4: a piece is inside or outside as a whole;
            // offsets are 32-bit and wrap, a negative one is a huge unsigned one).  Four address registers per frame.
            if constexpr (PCM == 1) {   // four int16 samples = 8 bytes per piece
                const __amdgpu_buffer_rsrc_t rs = make_rsrc(reinterpret_cast<const short*>(x) + (long long)clip * n_samples, (unsigned)(n_samples * 2));
                const int b = (int)s0 * 2 + 8 * p, rb = (int)s0 * 2 - 8 * p - 8 * (UPL - 1) * P;
                auto piece = [&](int off) {
                    const float2 d = buf_load_f32x2(rs, off);
                    const int d0 = __builtin_bit_cast(int, d.x), d1 = __builtin_bit_cast(int, d.y);
                    return make_float4((float)(short)(d0 & 0xffff), (float)(d0 >> 16), (float)(short)(d1 & 0xffff), (float)(d1 >> 16));
                };
#pragma unroll
                for (int r = 0; r < UPL; ++r) {
                    q[r][0] = piece(b + 6 * NF + 8 * r * P);
                    q[r][1] = piece(rb + 6 * NF - 8 + 8 * (UPL - 1 - r) * P);
                    q[r][2] = piece(b + 2 * NF + 8 * r * P);
                    q[r][3] = piece(rb + 2 * NF - 8 + 8 * (UPL - 1 - r) * P);
                }
                return;
            }
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(xc, (unsigned)(n_samples * 4));
            const int b = (int)s0 * 4 + 16 * p;                  // + 16 u forward pieces
            const int rb = (int)s0 * 4 - 16 * p - 16 * (UPL - 1) * P;   // - 16 u reversed pieces, lowest address of the UPL
            auto piece = [&](int off) {
                const float4 d = buf_load_f32x4(rs, off);
                if constexpr (PCM == 2) {   // four (left, right) frames: the sum of each (the 1/2 of the mean is in the window)
                    auto sum16 = [](float f) { const int v = __builtin_bit_cast(int, f); return (float)((int)(short)(v & 0xffff) + (v >> 16)); };
                    return make_float4(sum16(d.x), sum16(d.y), sum16(d.z), sum16(d.w));
                }
                return d;
            };
#pragma unroll
            for (int r = 0; r < UPL; ++r) {
                q[r][0] = piece(b + 12 * NF + 16 * r * P);
                q[r][1] = piece(rb + 12 * NF - 16 + 16 * (UPL - 1 - r) * P);
                q[r][2] = piece(b + 4 * NF + 16 * r * P);
                q[r][3] = piece(rb + 4 * NF - 16 + 16 * (UPL - 1 - r) * P);
            }
        } else {   // clip edges (zero padding), frames past T, unaligned clips
            auto at = [&](long long s) { return (t < T && s >= 0 && s < n_samples) ? xc[s] : 0.f; };
#pragma unroll
            for (int r = 0; r < UPL; ++r) {
                const int u = p + r * P;
                const long long b3 = s0 + 3 * NF + 4 * u, b2 = s0 + 3 * NF - 4 - 4 * u, b1 = s0 + NF + 4 * u, b0 = s0 + NF - 4 - 4 * u;
                q[r][0] = make_float4(at(b3), at(b3 + 1), at(b3 + 2), at(b3 + 3));
                q[r][1] = make_float4(at(b2), at(b2 + 1), at(b2 + 2), at(b2 + 3));
                q[r][2] = make_float4(at(b1), at(b1 + 1), at(b1 + 2), at(b1 + 3));
                q[r][3] = make_float4(at(b0), at(b0 + 1), at(b0 + 2), at(b0 + 3));
            }
        }
    };
    // fold + window + pre-twiddle: c[m] = (xa wa + xb wb, xc wc + xd wd) g_m -> LDS, natural order
#ifndef ZAFX_MDCT_TABLES_IN_REGS
#define ZAFX_MDCT_TABLES_IN_REGS 1
#endif
    // The lane's window quadruples are the same for every frame.  REGS: held in registers (32 at W = 2048; the buffer-load form
    // of fetch() left the room -- with the pre-twiddles as well, 48, the kernel spills): 8 of the frame's 16-byte LDS reads
    // fewer on a kernel whose transforms are bound by LDS.
    constexpr bool REGS = ZAFX_MDCT_TABLES_IN_REGS && ALIGNED && UPL <= 2 && !CARRY;   // (the carry takes the window's registers)
    float4 wq[REGS ? UPL : 1][4];
    if constexpr (REGS) {
#pragma unroll
        for (int r = 0; r < UPL; ++r) {
            const int u = (p + r * P) % NU;   // (lanes beyond the frame's groups never fold)
            const int m[4] = {2 * u, 2 * u + 1, NF - 1 - 2 * u, NF - 2 - 2 * u};
#pragma unroll
            for (int j = 0; j < 4; ++j) wq[r][j] = wf_l[m[j]];
        }
    }
    auto fold = [&](float2* buf) {
        if (!lane_loads) return;
        int opaque = 0;   // keeps the per-lane table addresses inside the loop: hoisted, they spill
        asm volatile("" : "+v"(opaque));
#pragma unroll
        for (int r = 0; r < UPL; ++r) {
            const int u = p + r * P + opaque;
            const float4 A3 = q[r][0], R2 = q[r][1], A1 = q[r][2], R0 = q[r][3];
            const int m0 = 2 * u, m1 = 2 * u + 1, m2 = NF - 1 - 2 * u, m3 = NF - 2 - 2 * u;
            const float4 w0 = REGS ? wq[REGS ? r : 0][0] : wf_l[m0], w1 = REGS ? wq[REGS ? r : 0][1] : wf_l[m1];
            const float4 w2 = REGS ? wq[REGS ? r : 0][2] : wf_l[m2], w3 = REGS ? wq[REGS ? r : 0][3] : wf_l[m3];
            const float2 g0 = g_l[m0], g1 = g_l[m1], g2 = g_l[m2], g3 = g_l[m3];
            buf[phys(m0)] = cmul(make_float2(R2.w * w0.x + A3.x * w0.y, R0.w * w0.z + A1.x * w0.w), g0);
            buf[phys(m1)] = cmul(make_float2(R2.y * w1.x + A3.z * w1.y, R0.y * w1.z + A1.z * w1.w), g1);
            buf[phys(m2)] = cmul(make_float2(R0.z * w2.x + A1.y * w2.y, R2.z * w2.z + A3.y * w2.w), g2);
            buf[phys(m3)] = cmul(make_float2(R0.x * w3.x + A1.w * w3.y, R2.x * w3.z + A3.w * w3.w), g3);
        }
    };

    // walk of this workgroup: tile tl, tl + gridDim.x, ... -- or (CARRY) the tiles j .. j1 - 1 of unit v, then of unit v + gridDim.x, ...
    int tl = blockIdx.x;
    int wv = blockIdx.x, wj = 0, wj1 = 0;
    constexpr int ITER = M / (NT / 16);
    float cva[CARRY ? ITER : 1], cvb[CARRY ? ITER : 1];   // the thread's pairs of the previous tile
    bool have_prev = false;
    auto unit_tile = [&](int v, int& j, int& j1) -> int {   // first tile (linear index) of unit v
        const int u = (ZAFX_XCD_ORDER && gridDim.x % 8 == 0) ? xcd_order(v, units) : v;
        const int c = u / segs;
        j = (u % segs) * seg_tiles;
        j1 = min(j + seg_tiles, tiles);
        return c * tiles + j;
    };
    if constexpr (CARRY) {
        if (wv >= units) return;
        tl = unit_tile(wv, wj, wj1);
#pragma unroll
        for (int i = 0; i < ITER; ++i) cva[i] = cvb[i] = 0.f;
    }
    fetch(tl, 0);
    PROF_INIT(g_prof_mdct);
    for (; tl < total_tiles;) {
        const int clip = tl / tiles, tile = tl % tiles;
        const int t0 = tile * FPB;
        int tl_next = tl + gridDim.x, nj = wj + 1, nj1 = wj1, nv = wv;
        if constexpr (CARRY) {
            if (nj < wj1) tl_next = tl + 1;
            else {
                nv = wv + gridDim.x;
                tl_next = nv < units ? unit_tile(nv, nj, nj1) : total_tiles;
            }
        }
        PROF_MARK(0);
#pragma unroll 1
        for (int f = 0; f < FPW; ++f) {
            float2* buf = frames + frame_of(f) * C::PITCH;
            fold(buf);
            PROF_MARK(1);
            if (f + 1 < FPW) fetch(tl, f + 1);
            else fetch(tl_next, 0);
            frame_sync<P>();
            int po = p;   // opaque copy: the pass-twiddle reads stay in the loop (hoisted, they spill at 128 VGPRs)
            asm volatile("" : "+v"(po));
            float2 v[E];
            regs_read<LOG2NF, LOG2E>(v, buf, po);
            frame_sync<P>();
            fft_frame_post<LOG2NF, LOG2E>(v, buf, po, tw_l, g_l);   // the frame now holds conj(y[k]), y[k] = Z[k] g[k]
            PROF_MARK(2);
            if constexpr (TFOUT) {
                // coefficients (2j, 2j+1) = (Re y[j], -Im y[NF-1-j]) with y[k] = Z[k] g[k] (zaf.py:1087-1091 after the
                // N/4-point reduction): one 8-byte store per lane
                frame_sync<P>();
                const int t = t0 + frame_of(f);
                if (t < T) {   // wave-uniform
                    float* o = out + ((long long)clip * T + t) * M;
#pragma unroll 4
                    for (int i = 0; i < NF / P; ++i) {
                        const int j = po + i * P;
                        const float re = buf[phys(j)].x, mim = buf[phys(NF - 1 - j)].y;   // Re y[j], -Im y[NF-1-j]
                        if (pair_ok) {
                            *reinterpret_cast<float2*>(o + 2 * j) = make_float2(re, mim);
                        } else {
                            o[2 * j] = re;
                            o[2 * j + 1] = mim;
                        }
                    }
                }
                frame_sync<P>();   // these reads precede the buffer's next fold
            }
        }
        if constexpr (TFOUT) {   // wave-private buffers: no workgroup barrier in the tile loop
            tl = tl_next;
            continue;
        }
        lds_barrier();
        PROF_MARK(3);
        int tido = tid;   // opaque: the store-phase indices are recomputed per tile (kept live across the FFT loop they are
        asm volatile("" : "+v"(tido));   // spilled at 128 VGPRs, and a scratch reload here waits for vmcnt(0): the next tile's prefetch)
        const int tp = tido % 16, fq = tido / 16;
        const int ta = t0 + 2 * tp;
        if constexpr (CARRY) {
            int fqo = (fq & ~3) | ((fq & 1) << 1) | ((fq >> 1) & 1);   // (row order of the lane groups: see below)
            asm volatile("" : "+v"(fqo));
            const float2* ba = frames + (2 * tp) * C::PITCH;
            const float2* bb = ba + C::PITCH;
            float* o = out + (long long)clip * M * TP + ta;
            const bool cur_ok = ta < T, last = wj + 1 >= wj1;
            const int odd = fqo & 1;
            const int k = odd ? (M - 1 - fqo) >> 1 : fqo >> 1;
            constexpr int DK = NT / 32, DPH = DK + DK / 16;
            const float* pa = reinterpret_cast<const float*>(ba + phys(k)) + odd;
            const float* pb2 = reinterpret_cast<const float*>(bb + phys(k)) + odd;
            const int dslot = odd ? -2 * DPH : 2 * DPH;
            // phase of a row's run in its line, in floats: (array + (clip M + f) TP) mod 32 -- even, since TP and the array's offset are
            const int b0 = (int)((reinterpret_cast<uintptr_t>(out) >> 2) & 31), c0 = (int)(((long long)clip * M) & 31), tpm = TP & 31;
            const bool pairs = TP % 2 == 0 && reinterpret_cast<uintptr_t>(out) % 8 == 0;
            auto sweep = [&](auto stream) {
                constexpr bool ST = decltype(stream)::value;
#pragma unroll
                for (int i = 0; i < ITER; ++i) {
                    const int f = fqo + i * (NT / 16);
                    const float va = pa[i * dslot], vb = pb2[i * dslot];
                    const int a = (b0 + (c0 + f) * tpm) & 31;
                    float* dst = o + (long long)f * TP;
                    if (pairs) {   // (uniform) even T: a is even, a lane's pair lies in one line
                        const bool from_prev = tp >= 16 - (a >> 1);   // (a = 0: never)
                        const float2 val = from_prev ? make_float2(cva[i], cvb[i]) : make_float2(va, vb);
                        if (from_prev ? have_prev : cur_ok) {
                            if constexpr (ST) store_stream(reinterpret_cast<float2*>(dst + (from_prev ? -32 : 0)), val);
                            else *reinterpret_cast<float2*>(dst + (from_prev ? -32 : 0)) = val;
                        }
                        if (last && from_prev && cur_ok) *reinterpret_cast<float2*>(dst) = make_float2(va, vb);   // tail of the segment's last run
                    } else {
                        // odd T: rows start at any float, a pair may straddle the line boundary: the two floats of a lane choose for
                        // themselves, as two 4-byte stores -- two instructions of the same wave, back to back, that together cover the
                        // line (L2 merges them)
                        const bool pa_ = 2 * tp >= 32 - a, pb_ = 2 * tp + 1 >= 32 - a;
                        if (pa_ ? have_prev : cur_ok) dst[pa_ ? -32 : 0] = pa_ ? cva[i] : va;
                        if (pb_ ? have_prev : (ta + 1 < T)) dst[pb_ ? -31 : 1] = pb_ ? cvb[i] : vb;
                        if (last && pa_ && cur_ok) dst[0] = va;
                        if (last && pb_ && ta + 1 < T) dst[1] = vb;
                    }
                    cva[i] = va;
                    cvb[i] = vb;
                }
            };
            if (have_prev) sweep(std::true_type{});
            else sweep(std::false_type{});
        } else
        if (ta < T) {
            // Row of lane group g = (tid / 16) % 4 within the wave's 4 rows: 0, 2, 1, 3.  ds_read_b64 serves 32 lanes
            // per cycle; rows f and f + 2 of a half-wave read bins k and k + 1 (bank offset 2 dwords: conflict free with
            // the 4-dword frame-pair stride), rows f and f + 1 read bins k and M/2 - 1 - k (same bank class: 2-way).
            int fqo = (fq & ~3) | ((fq & 1) << 1) | ((fq >> 1) & 1);   // (opaque: g_l[k] is not carried across tiles)
            asm volatile("" : "+v"(fqo));
            const float2* ba = frames + (2 * tp) * C::PITCH;
            const float2* bb = ba + C::PITCH;
            float* o = out + (long long)clip * M * TP + ta;
            const bool two = ta + 1 < T;
            // coefficient f of a frame is Re y[f / 2] (f even) or -Im y[(M - 1 - f) / 2] (f odd): one float of the frame's
            // conj(y) image.  A thread's f advances by NT / 16 (even), so its parity, the component it reads and the stride
            // of its bin (+- NT / 32, i.e. +- (NT / 32 + NT / 512) padded slots) are fixed: two ds_read_b32, one 8-byte
            // store and three additions per coefficient pair.
            const int odd = fqo & 1;
            int k = odd ? (M - 1 - fqo) >> 1 : fqo >> 1;
            constexpr int DK = NT / 32, DPH = DK + DK / 16;   // bins / padded slots per step of f
            const float* pa = reinterpret_cast<const float*>(ba + phys(k)) + odd;
            const float* pb2 = reinterpret_cast<const float*>(bb + phys(k)) + odd;
            const int dslot = odd ? -2 * DPH : 2 * DPH;     // floats
            float* dst = o + (long long)fqo * TP;
            const long long dstep = (long long)(NT / 16) * TP;
            // (A sweep that starts somewhere else in every workgroup, as in k_stft_ft16, gains nothing here: 0.791 against 0.787 ms,
            // A/B on one box, round 3 -- the switch stays for the experiment.)
            constexpr int ITER = M / (NT / 16);
#ifndef ZAFX_MDCT_ROWROT_EXPR
#define ZAFX_MDCT_ROWROT_EXPR 0
#endif
            const int rot = (lines_whole && pair_ok && M % (NT / 16) == 0) ? (ZAFX_MDCT_ROWROT_EXPR) % ITER : 0;
            auto sweep = [&](int i0, int i1) {
                const float* qa = pa + (long long)i0 * dslot;
                const float* qb = pb2 + (long long)i0 * dslot;
                float* d = dst + (long long)i0 * dstep;
#pragma unroll 4
                for (int f = fqo + i0 * (NT / 16); f < M && i0 < i1; f += NT / 16, ++i0) {
                    const float va = *qa, vb = *qb;
                    if (pair_ok && two) {
                        if (lines_whole) store_stream(reinterpret_cast<float2*>(d), make_float2(va, vb));   // a 128-B line per 16 lanes, written once
                        else *reinterpret_cast<float2*>(d) = make_float2(va, vb);
                    }
                    else {
                        d[0] = va;
                        if (two) d[1] = vb;
                    }
                    qa += dslot;
                    qb += dslot;
                    d += dstep;
                }
            };
            sweep(rot, ITER + 1);
            if (rot) sweep(0, rot);
        }
        PROF_MARK(4);
        lds_barrier();   // LDS reads of the tile are done; its global stores are not waited for
        if constexpr (CARRY) {
            have_prev = nv == wv;   // the walk continues inside the same segment
            wv = nv, wj = nj, wj1 = nj1;
        }
        tl = tl_next;
    }
}

// ---------------------------------------------------------------------------------
// forward, reference layout, W = 4096: 32-frame tiles over two BANDS of bins (k_mdct_ft32b)
// ---------------------------------------------------------------------------------
// Thirty-two packed frames of NF = 1024 points are 256 KB: W = 4096 ran on the one-workgroup-per-tile kernel with 6-frame tiles
// (24-byte runs, 1.5 TB/s).  As in k_stft_ft16b one decimation-in-frequency step on the way in splits the packed frame c into
// two 512-point transforms, Y[2q] = FFT_512(c[n] + c[n + 512])[q] and Y[2q + 1] = FFT_512((c[n] - c[n + 512]) w^n)[q],
// w = exp(-2 pi i / 1024); every bin yields its two coefficients on its own (X[2k] = Re y_k, X[M - 1 - 2k] = -Im y_k), so a band is
// transformed, post-twiddled and stored without the other: a tile is two rounds of k_mdct_ft32's phases on the same 32 frame
// buffers, and every stored row is again a 128-byte run of 32 frames.
//   * c[n] = F[n] g[n] and g[n + 512] = g[n] exp(-i pi / 4), so the two bands of a pair are g[n] (F[n] + k F[n + 512]) and
//     (g[n] w^n) (F[n] - k F[n + 512]) with the constant k = exp(-i pi / 4): the tables are gb (g in band-major order -- it also
//     serves the post-twiddle of each band) and bt[n] = g[n] w^n, both in LDS.
//   * lane p folds the 16-sample groups u = p + 64 r AND 255 - u (r = 0, 1): the group 255 - u holds exactly the partners
//     c[n + 512] of the group u's c[n] (and vice versa), so both bands of eight points per half frame come out of one lane's
//     loads; the reversed groups are still coalesced 1-KB rows.
//   * band 1 waits in registers (16 pairs per lane for the wave's two frames) while band 0 is transformed and stored.
//   * the sign-folded window (16 KB) does not fit LDS beside the frames and is read from global memory (L2) per half frame.
struct MdctBandCfg {
    using C = FftCfg<9, 3>;   // 512-point band transforms: 64 lanes x 8 points, radix 8 x 8 x 8
    static constexpr int NF = 1024, HB = 512, M = 2048, FPB = kMdctTile, NSLOT = 16, NT = NSLOT * 64;
    static constexpr size_t SMEM = (size_t)(FPB * C::PITCH + C::TW + NF + HB) * 8;
};
static_assert(MdctBandCfg::SMEM <= (size_t)kMaxLdsBytes, "k_mdct_ft32b: tile + tables exceed LDS");

__global__ __launch_bounds__(MdctBandCfg::NT) void k_mdct_ft32b(
    const float* __restrict__ x, const float4* __restrict__ wfold, const float2* __restrict__ twp, const float2* __restrict__ band_tw,
    float* __restrict__ out, long long n_samples, int T, int TP, int tiles, int total_tiles) {
    using G = MdctBandCfg;
    using C = G::C;
    constexpr int NF = G::NF, HB = G::HB, M = G::M, P = 64, E = 8, FPB = G::FPB, NSLOT = G::NSLOT, NT = G::NT, FPW = FPB / NSLOT;
    [[maybe_unused]] constexpr int NU = NF / 4;   // 16-sample groups per frame
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* frames = reinterpret_cast<float2*>(smem_raw);
    float2* tw_l = frames + FPB * C::PITCH;
    float2* gb_l = tw_l + C::TW;   // gb[s][q] = g[2 q + s]
    float2* bt_l = gb_l + NF;      // bt[n] = g[n] exp(-2 pi i n / 1024), n < 512
    const int tid = threadIdx.x;
    for (int i = tid; i < C::TW; i += NT) tw_l[i] = twp[i];
    for (int i = tid; i < NF + HB; i += NT) gb_l[i] = band_tw[i];
    lds_barrier();
    const int slot = tid / P, p = tid % P;
    const bool pair_ok = TP % 2 == 0 && reinterpret_cast<uintptr_t>(out) % 8 == 0;
    const bool lines_whole = TP % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 64 == 0;
    auto frame_of = [&](int f) { return f * NSLOT + slot; };

    // half frame h = 2 f + r of the tile: the groups u = p + 64 r and NU - 1 - u, eight 16-byte pieces
    //   q[0..3] = A3, R2, A1, R0 of u (as k_mdct_ft32), q[4..7] = the same of NU - 1 - u
    float4 q[8];
    auto fetch = [&](int tl, int h) {
        if (tl >= total_tiles) return;
        const int clip = tl / tiles, tile = tl % tiles;
        const int t = tile * FPB + frame_of(h >> 1);
        const float* xc = x + (long long)clip * n_samples;
        // the clip as buffer descriptor: pieces outside it read as zero = the zero padding of zaf.py:1036-1041 (n_samples and every
        // piece's first sample are multiples of 4; 32-bit offsets wrap, a negative one is a huge unsigned one)
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(xc, (unsigned)(n_samples * 4));
        const int u = p + (h & 1) * P;
        const int s0 = (t - 1) * M * 4;   // byte offset of the frame's first sample (left pad = M)
        const int fw = s0 + 16 * u, bw = s0 - 16 - 16 * u;
        q[0] = buf_load_f32x4(rs, fw + 12 * NF);
        q[1] = buf_load_f32x4(rs, bw + 12 * NF);
        q[2] = buf_load_f32x4(rs, fw + 4 * NF);
        q[3] = buf_load_f32x4(rs, bw + 4 * NF);
        q[4] = buf_load_f32x4(rs, bw + 16 * NF);
        q[5] = buf_load_f32x4(rs, fw + 8 * NF);
        q[6] = buf_load_f32x4(rs, bw + 8 * NF);
        q[7] = buf_load_f32x4(rs, fw);
    };
    float2 hold[FPW][8];   // band 1 of the wave's frames: b[n] for n = 2u, 2u+1, 510-2u, 511-2u of both half frames
    // fold one half frame: band 0 -> the frame buffer (natural order), band 1 -> hold[f][4 r ..]
    auto fold = [&](float2* buf, int r, float2* hb) {
        int opaque = 0;   // keeps the per-lane table addresses inside the loop
        asm volatile("" : "+v"(opaque));
        const int u = p + r * P + opaque;
        const int m0 = 2 * u, m1 = 2 * u + 1, m2 = NF - 1 - 2 * u, m3 = NF - 2 - 2 * u;   // group u
        const int n0 = HB - 2 - 2 * u, n1 = HB - 1 - 2 * u;                                 // group NU-1-u: m0', m1' (its m2' = m1 + HB, m3' = m0 + HB)
        const float4 A3 = q[0], R2 = q[1], A1 = q[2], R0 = q[3], B3 = q[4], S2 = q[5], B1 = q[6], S0 = q[7];
        // F[m] of the two groups (k_mdct_ft32's fold), one group after the other: the eight window quadruples of both at once
        // are 32 registers the kernel does not have (44 bytes of scratch)
        const float4 w0 = wfold[m0], w1 = wfold[m1], w2 = wfold[m2], w3 = wfold[m3];
        const float2 F0 = make_float2(R2.w * w0.x + A3.x * w0.y, R0.w * w0.z + A1.x * w0.w);   // F[2u]
        const float2 F1 = make_float2(R2.y * w1.x + A3.z * w1.y, R0.y * w1.z + A1.z * w1.w);   // F[2u+1]
        const float2 F2 = make_float2(R0.z * w2.x + A1.y * w2.y, R2.z * w2.z + A3.y * w2.w);   // F[NF-1-2u]
        const float2 F3 = make_float2(R0.x * w3.x + A1.w * w3.y, R2.x * w3.z + A3.w * w3.w);   // F[NF-2-2u]
        __builtin_amdgcn_sched_barrier(0);
        const float4 v0 = wfold[n0], v1 = wfold[n1], v2 = wfold[m1 + HB], v3 = wfold[m0 + HB];
        const float2 G0 = make_float2(S2.w * v0.x + B3.x * v0.y, S0.w * v0.z + B1.x * v0.w);   // F[510-2u]
        const float2 G1 = make_float2(S2.y * v1.x + B3.z * v1.y, S0.y * v1.z + B1.z * v1.w);   // F[511-2u]
        const float2 G2 = make_float2(S0.z * v2.x + B1.y * v2.y, S2.z * v2.z + B3.y * v2.w);   // F[513+2u]
        const float2 G3 = make_float2(S0.x * v3.x + B1.w * v3.y, S2.x * v3.z + B3.w * v3.w);   // F[512+2u]
        const float h = 0.70710678118654752440f;
        auto bands = [&](int n, float2 lo, float2 hi, float2& b) {
            const float2 d = cmulk(hi, h, -h);   // F[n + 512] exp(-i pi / 4)
            buf[phys(n)] = cmul(cadd(lo, d), gb_l[(n & 1) * HB + (n >> 1)]);
            b = cmul(csub(lo, d), bt_l[n]);
        };
        bands(m0, F0, G3, hb[0]);
        bands(m1, F1, G2, hb[1]);
        bands(n0, G0, F3, hb[2]);
        bands(n1, G1, F2, hb[3]);
    };
    auto store_band = [&](int s, int clip, int t0) {
        int tido = tid;   // opaque: the store-phase indices are recomputed per tile
        asm volatile("" : "+v"(tido));
        const int tp = tido % 16, fq = tido / 16;
        const int ta = t0 + 2 * tp;
        if (ta >= T) return;
        // thread: frame pair (2 tp, 2 tp + 1), parity e of its rows (0: X[2k] = Re y_k, 1: X[M - 1 - 2k] = -Im y_k; the frame holds
        // conj(y)), bins q = qi + 32 i of the band, k = 2 q + s
        const int e = fq & 1, qi = fq >> 1;
        const float* pa = reinterpret_cast<const float*>(frames + (2 * tp) * C::PITCH + phys(qi)) + e;
        const float* pb = pa + 2 * C::PITCH;
        constexpr int DPH = 2 * (32 + 32 / 16);   // floats per step of 32 bins (padded slots)
        const int f0 = e ? M - 1 - 4 * qi - 2 * s : 4 * qi + 2 * s;
        float* d = out + (long long)clip * M * TP + ta + (long long)f0 * TP;
        const long long dstep = (long long)(e ? -128 : 128) * TP;
        const bool two = ta + 1 < T;
#pragma unroll 4
        for (int i = 0; i < HB / 32; ++i) {
            const float va = pa[i * DPH], vb = pb[i * DPH];
            if (pair_ok && two) {
                if (lines_whole) store_stream(reinterpret_cast<float2*>(d), make_float2(va, vb));
                else *reinterpret_cast<float2*>(d) = make_float2(va, vb);
            } else {
                d[0] = va;
                if (two) d[1] = vb;
            }
            d += dstep;
        }
    };

    int tl = blockIdx.x;
    fetch(tl, 0);
    for (; tl < total_tiles; tl += gridDim.x) {
        const int clip = tl / tiles, tile = tl % tiles;
        const int t0 = tile * FPB;
        const int tl_next = tl + gridDim.x;
        int po = p;   // opaque copy: the pass-twiddle reads stay in the loop
        asm volatile("" : "+v"(po));
#pragma unroll
        for (int f = 0; f < FPW; ++f) {
            float2* buf = frames + frame_of(f) * C::PITCH;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                fold(buf, r, &hold[f][4 * r]);
                const int h = 2 * f + r + 1;
                if (h < 2 * FPW) fetch(tl, h);
                else fetch(tl_next, 0);
            }
            frame_sync<P>();
            float2 v[E];
            regs_read<9, 3>(v, buf, po);
            frame_sync<P>();
            fft_frame_post<9, 3>(v, buf, po, tw_l, gb_l);   // the frame now holds conj(y[2q]), y[k] = Y[k] g[k]
        }
        lds_barrier();
        store_band(0, clip, t0);
        lds_barrier();
#pragma unroll
        for (int f = 0; f < FPW; ++f) {
            float2* buf = frames + frame_of(f) * C::PITCH;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int u = po + r * P;
                buf[phys(2 * u)] = hold[f][4 * r];
                buf[phys(2 * u + 1)] = hold[f][4 * r + 1];
                buf[phys(HB - 2 - 2 * u)] = hold[f][4 * r + 2];
                buf[phys(HB - 1 - 2 * u)] = hold[f][4 * r + 3];
            }
            frame_sync<P>();
            float2 v[E];
            regs_read<9, 3>(v, buf, po);
            frame_sync<P>();
            fft_frame_post<9, 3>(v, buf, po, tw_l, gb_l + HB);   // conj(y[2q + 1])
        }
        lds_barrier();
        store_band(1, clip, t0);
        lds_barrier();
    }
}

// ---------------------------------------------------------------------------------
// forward, reference layout, W = 8192: 32-frame tiles over FOUR bands of bins (k_mdct_ft32q)
// ---------------------------------------------------------------------------------
// The packed frame c (NF = 2048 points) in two decimation steps: Y[4q + j] = FFT_512(b_j)[q] with
//     b_0 = e[n] + e[n + 512],  b_2 = (e[n] - e[n + 512]) w^2n,        e[n] = c[n] + c[n + 1024]
//     b_1 = (d[n] - i d[n + 512]) w^n,  b_3 = (d[n] + i d[n + 512]) w^3n,  d[n] = c[n] - c[n + 1024],   w = exp(-2 pi i / 2048), n < 512
// and every bin yields its two coefficients on its own (X[2k] = Re y_k, X[M - 1 - 2k] = -Im y_k, y_k = Y[k] g_k): a tile is four rounds of
// k_mdct_ft32's phases on the same 32 frame buffers, every stored row a 128-byte run of 32 frames.  The frame is folded twice (bands 0 + 2 from the
// sums e, 1 + 3 from the differences d).  A lane folds the points n + 512 k and their mirrors (511 - n) + 512 k together: c[m] and c[NF - 1 - m] use
// the odd and the even neighbours of the same four sample pairs, so every sample comes once per pass, in an 8-byte load of a coalesced run (two 4-byte
// loads for clips off the 8-byte grid; samples outside the clip read 0 = the reference's padding, zaf.py:1036-1041), and the mirrored points reach
// their lane 63 - p by a lane permute.  A memory instruction of a wave completes in issue order, so both
// folds are issued AHEAD of a store sweep (k_stft_ft16q) and their bands wait in 64 registers.
struct MdctQuadCfg {
    using C = FftCfg<9, 3>;   // 512-point band transforms: 64 lanes x 8 points
    static constexpr int NF = 2048, HB = 512, M = 4096, W = 8192, FPB = kMdctTile, NSLOT = 16, NT = NSLOT * 64;
    static constexpr size_t SMEM = (size_t)(FPB * C::PITCH + C::TW + NF) * 8;
};
static_assert(MdctQuadCfg::SMEM <= (size_t)kMaxLdsBytes, "k_mdct_ft32q: tile + tables exceed LDS");

template <bool ALIGNED>
__global__ __launch_bounds__(MdctQuadCfg::NT) void k_mdct_ft32q(
    const float* __restrict__ x, const float* __restrict__ win, const float2* __restrict__ twp, const float2* __restrict__ g, const float2* __restrict__ wq,
    float* __restrict__ out, long long n_samples, int T, int TP, int tiles, int total_tiles) {
    using G = MdctQuadCfg;
    using C = G::C;
    constexpr int NF = G::NF, HB = G::HB, M = G::M, P = 64, E = 8, FPB = G::FPB, NSLOT = G::NSLOT, NT = G::NT, FPW = FPB / NSLOT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* frames = reinterpret_cast<float2*>(smem_raw);
    float2* tw_l = frames + FPB * C::PITCH;
    const int tid = threadIdx.x;
    float2* g_l = tw_l + C::TW;   // g[m], m < NF: the pre-twiddle of the fold and the post-twiddle of the sweeps
    for (int i = tid; i < C::TW; i += NT) tw_l[i] = twp[i];
    for (int i = tid; i < NF; i += NT) g_l[i] = g[i];
    lds_barrier();
    const int slot = __builtin_amdgcn_readfirstlane(tid / P), p = tid % P;
    const bool xcd = ZAFX_XCD_ORDER && gridDim.x % 8 == 0;
    auto tile_of = [&](int tlv, int& clip, int& t0) {
        const int tl = xcd ? xcd_order(tlv, total_tiles) : tlv;
        clip = tl / tiles;
        t0 = (tl % tiles) * FPB;
    };
    float2 now[FPW][E], wait[FPW][E];   // the band transformed next and the one behind it, of the wave's two frames
    // the fold of one pass (PASS 0: bands 0 and 2, PASS 1: bands 1 and 3) of the tile tlv
    auto fold = [&](int tlv, auto pass) {
        constexpr int PASS = decltype(pass)::value;
        if (tlv >= total_tiles) {   // (behind the last tile: dead values, said so that they are not carried through the loop)
#pragma unroll
            for (int f = 0; f < FPW; ++f)
#pragma unroll
                for (int i = 0; i < E; ++i) now[f][i] = wait[f][i] = make_float2(0.f, 0.f);
            return;
        }
        int clip, t0;
        tile_of(tlv, clip, t0);
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(x + (long long)clip * n_samples, (unsigned)(n_samples * 4));
        __builtin_amdgcn_sched_barrier(0);
        const int s0 = (t0 + slot - 1) * M;   // the wave's frames f = 0, 1 start at s0 + f NSLOT M
#pragma unroll
        for (int i = 0; i < E / 2; ++i) {
            int po = p;
            asm volatile("" : "+v"(po));   // (window values and roots are read again per step, not kept across the tile loop)
            const int n = po + 64 * i, nb = HB - 1 - n;   // the lane's points n + 512 k and their mirrors (511 - n) + 512 k: NF - 1 - m of each other
            float2 cn[FPW][4], cb[FPW][4];
            // c[m] and c[NF - 1 - m], m < NF / 2, from four sample pairs (k_mdct's fold: c[m] = (-u[3NF-1-2m] - u[3NF+2m], u[NF-1-2m] - u[NF+2m]) g[m];
            // c[m'] = (u[2m'-NF] - u[3NF-1-2m'], -u[NF+2m'] - u[5NF-1-2m']) g[m'], m' = NF - 1 - m -- the odd / even neighbours of the same samples);
            // window values and roots once for the wave's two frames
            auto both = [&](int m, int k, int kb, float2 (&cm)[FPW][4], float2 (&cmb)[FPW][4]) {
                const int e[4] = {NF - 2 - 2 * m, NF + 2 * m, 3 * NF - 2 - 2 * m, 3 * NF + 2 * m};
                float2 v[FPW][4], w[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int f = 0; f < FPW; ++f) {
                        const int off = (s0 + f * NSLOT * M + e[j]) * 4;
                        if constexpr (ALIGNED) {
                            v[f][j] = buf_load_f32x2(rs, off);
                        } else {
                            int off1 = off + 4;
                            asm volatile("" : "+v"(off1));   // (two 4-byte loads that must not be merged: a pair may straddle an end of the clip)
                            v[f][j].x = buf_load_f32(rs, off);
                            v[f][j].y = buf_load_f32(rs, off1);
                        }
                    }
                    w[j] = *reinterpret_cast<const float2*>(win + e[j]);
                }
                const float2 gm = g_l[m], gb = g_l[NF - 1 - m];
#pragma unroll
                for (int f = 0; f < FPW; ++f) {
                    float2 q[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) q[j] = make_float2(v[f][j].x * w[j].x, v[f][j].y * w[j].y);
                    cm[f][k] = cmul(make_float2(-q[2].y - q[3].x, q[0].y - q[1].x), gm);
                    cmb[f][kb] = cmul(make_float2(q[0].x - q[1].y, -q[2].x - q[3].y), gb);
                }
            };
            both(n, 0, 3, cn, cb);
            both(nb, 0, 3, cb, cn);
            both(n + HB, 1, 2, cn, cb);
            both(nb + HB, 1, 2, cb, cn);
            const float2 wa = PASS == 0 ? wq[2 * n] : wq[n], wb = PASS == 0 ? wq[2 * nb] : wq[nb];
            float2 wa3 = wa, wb3 = wb;
            if constexpr (PASS == 1) wa3 = wq[3 * n], wb3 = wq[3 * nb];
            const int opp = (63 - po) * 4;   // point nb = (63 - lane) + 64 (7 - i) belongs to the opposite lane's register 7 - i
            auto flip = [&](float2 v) {
                return make_float2(__builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(opp, __builtin_bit_cast(int, v.x))),
                                   __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(opp, __builtin_bit_cast(int, v.y))));
            };
#pragma unroll
            for (int f = 0; f < FPW; ++f) {
                float2 rn, rw, qn, qw;   // the two bands at n and at nb
                if constexpr (PASS == 0) {
                    const float2 e0 = cadd(cn[f][0], cn[f][2]), e1 = cadd(cn[f][1], cn[f][3]), f0 = cadd(cb[f][0], cb[f][2]), f1 = cadd(cb[f][1], cb[f][3]);
                    rn = cadd(e0, e1);
                    rw = cmul(csub(e0, e1), wa);
                    qn = cadd(f0, f1);
                    qw = cmul(csub(f0, f1), wb);
                } else {
                    const float2 d0 = csub(cn[f][0], cn[f][2]), d1 = csub(cn[f][1], cn[f][3]), f0 = csub(cb[f][0], cb[f][2]), f1 = csub(cb[f][1], cb[f][3]);
                    rn = cmul(make_float2(d0.x + d1.y, d0.y - d1.x), wa);     // d0 - i d1
                    rw = cmul(make_float2(d0.x - d1.y, d0.y + d1.x), wa3);    // d0 + i d1
                    qn = cmul(make_float2(f0.x + f1.y, f0.y - f1.x), wb);
                    qw = cmul(make_float2(f0.x - f1.y, f0.y + f1.x), wb3);
                }
                now[f][i] = rn;
                wait[f][i] = rw;
                now[f][E - 1 - i] = flip(qn);
                wait[f][E - 1 - i] = flip(qw);
            }
        }
    };
    auto transform = [&](float2 (&v)[FPW][E]) {
        int po = p;
        asm volatile("" : "+v"(po));
#pragma unroll
        for (int f = 0; f < FPW; ++f) fft_frame<9, 3>(&v[f][0], frames + (f * NSLOT + slot) * C::PITCH, po, tw_l);
    };
    // rows 8q + 2j and M - 1 - 8q - 2j of band j, frame pairs as 8-byte stores (one instruction: 4 rows x 128 B)
    auto store_band = [&](int j, int clip, int t0) {
        int tido = tid;
        asm volatile("" : "+v"(tido));
        const int tp = tido % 16, fq = tido / 16;
        const int ta = t0 + 2 * tp;
        if (ta >= T) return;
        const float2* fa = frames + (2 * tp) * C::PITCH;
        const float2* fb = fa + C::PITCH;
        float* o = out + (long long)clip * M * TP + ta;
        const bool pairs = TP % 2 == 0 && reinterpret_cast<uintptr_t>(out) % 8 == 0 && ta + 1 < T;
#pragma unroll 2
        for (int it = 0; it < HB / 64; ++it) {
            const int q = fq + 64 * it;
            const float2 gk = g_l[4 * q + j];
            const float2 ya = cmul(fa[phys(q)], gk), yb = cmul(fb[phys(q)], gk);
            float* r1 = o + (long long)(8 * q + 2 * j) * TP;
            float* r2 = o + (long long)(M - 1 - 8 * q - 2 * j) * TP;
            if (pairs) {
                store_stream(reinterpret_cast<float2*>(r1), make_float2(ya.x, yb.x));
                store_stream(reinterpret_cast<float2*>(r2), make_float2(-ya.y, -yb.y));
            } else {
                r1[0] = ya.x;
                r2[0] = -ya.y;
                if (ta + 1 < T) r1[1] = yb.x, r2[1] = -yb.y;
            }
        }
    };
    int tlv = blockIdx.x;
    fold(tlv, std::integral_constant<int, 0>{});
    for (; tlv < total_tiles; tlv += gridDim.x) {
        int clip, t0;
        tile_of(tlv, clip, t0);
        transform(now);   // band 0
        lds_barrier();
        store_band(0, clip, t0);
        lds_barrier();
        transform(wait);   // band 2
        fold(tlv, std::integral_constant<int, 1>{});
        lds_barrier();
        store_band(2, clip, t0);
        lds_barrier();
        transform(now);   // band 1
        lds_barrier();
        store_band(1, clip, t0);
        lds_barrier();
        transform(wait);   // band 3
        fold(tlv + gridDim.x, std::integral_constant<int, 0>{});
        lds_barrier();
        store_band(3, clip, t0);
        lds_barrier();
    }
}

// ---------------------------------------------------------------------------------
// forward, reference layout, W = 4096, rows that are not whole (half) lines: one band per workgroup + register carry (k_mdct_ft32bc)
// ---------------------------------------------------------------------------------
// As k_stft_ft16bc: the two bands of k_mdct_ft32b never meet, so a workgroup owns ONE band of a clip segment (units (segment, band 0 / 1)
// are neighbours in the XCD order: both fold every sample, L2 serves the second reader), walks the segment's 32-frame tiles in order and
// carries its 16 row pairs of the previous tile in 32 VGPRs -- the registers k_mdct_ft32b spends on the waiting band -- exactly as
// k_mdct_ft32<CARRY> does: for a row whose run starts a floats into a line the lanes tp < 16 - a / 2 store the current pair at frame
// t0 + 2 tp, the others the carried pair at t0 - 32 + 2 tp (odd T: the two floats of a lane choose for themselves, 4-byte stores).
__global__ __launch_bounds__(MdctBandCfg::NT) void k_mdct_ft32bc(
    const float* __restrict__ x, const float4* __restrict__ wfold, const float2* __restrict__ twp, const float2* __restrict__ band_tw,
    float* __restrict__ out, long long n_samples, int T, int TP, int tiles, int segs, int seg_tiles, int units) {
    using G = MdctBandCfg;
    using C = G::C;
    constexpr int NF = G::NF, HB = G::HB, M = G::M, P = 64, E = 8, FPB = G::FPB, NSLOT = G::NSLOT, NT = G::NT, FPW = FPB / NSLOT;
    constexpr int ITER = HB / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* frames = reinterpret_cast<float2*>(smem_raw);
    float2* tw_l = frames + FPB * C::PITCH;
    float2* gb_l = tw_l + C::TW;   // gb[s][q] = g[2 q + s]
    float2* bt_l = gb_l + NF;      // bt[n] = g[n] exp(-2 pi i n / 1024), n < 512
    const int tid = threadIdx.x;
    for (int i = tid; i < C::TW; i += NT) tw_l[i] = twp[i];
    for (int i = tid; i < NF + HB; i += NT) gb_l[i] = band_tw[i];
    lds_barrier();
    const int slot = __builtin_amdgcn_readfirstlane(tid / P), p = tid % P;   // (the wave's frame slot is a scalar: its buffer addresses and frame indices stay in SGPRs)
    const bool xcd = ZAFX_XCD_ORDER && gridDim.x % 8 == 0;
    auto frame_of = [&](int f) { return f * NSLOT + slot; };
    float4 q[8];
    auto fetch = [&](int clip, int tile, int h) {   // half frame h = 2 f + r of a tile: the groups u = p + 64 r and 255 - u (k_mdct_ft32b)
        const int t = tile * FPB + frame_of(h >> 1);
        const float* xc = x + (long long)clip * n_samples;
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(xc, (unsigned)(n_samples * 4));
        int pp = p;   // (opaque: the per-lane byte offsets are recomputed per request, not carried across the tile)
        asm volatile("" : "+v"(pp));
        const int u = pp + (h & 1) * P;
        const int s0 = (t - 1) * M * 4;
        const int fw = s0 + 16 * u, bw = s0 - 16 - 16 * u;
        q[0] = buf_load_f32x4(rs, fw + 12 * NF);
        q[1] = buf_load_f32x4(rs, bw + 12 * NF);
        q[2] = buf_load_f32x4(rs, fw + 4 * NF);
        q[3] = buf_load_f32x4(rs, bw + 4 * NF);
        q[4] = buf_load_f32x4(rs, bw + 16 * NF);
        q[5] = buf_load_f32x4(rs, fw + 8 * NF);
        q[6] = buf_load_f32x4(rs, bw + 8 * NF);
        q[7] = buf_load_f32x4(rs, fw);
    };
    auto fold = [&](float2* buf, int r, int band) {   // one half frame: this band only -> the frame buffer (natural order)
        int opaque = 0;
        asm volatile("" : "+v"(opaque));
        const int u = p + r * P + opaque;
        const int m0 = 2 * u, m1 = 2 * u + 1, m2 = NF - 1 - 2 * u, m3 = NF - 2 - 2 * u;
        const int n0 = HB - 2 - 2 * u, n1 = HB - 1 - 2 * u;
        const float4 A3 = q[0], R2 = q[1], A1 = q[2], R0 = q[3], B3 = q[4], S2 = q[5], B1 = q[6], S0 = q[7];
        const float4 w0 = wfold[m0], w1 = wfold[m1], w2 = wfold[m2], w3 = wfold[m3];
        const float2 F0 = make_float2(R2.w * w0.x + A3.x * w0.y, R0.w * w0.z + A1.x * w0.w);
        const float2 F1 = make_float2(R2.y * w1.x + A3.z * w1.y, R0.y * w1.z + A1.z * w1.w);
        const float2 F2 = make_float2(R0.z * w2.x + A1.y * w2.y, R2.z * w2.z + A3.y * w2.w);
        const float2 F3 = make_float2(R0.x * w3.x + A1.w * w3.y, R2.x * w3.z + A3.w * w3.w);
        __builtin_amdgcn_sched_barrier(0);
        const float4 v0 = wfold[n0], v1 = wfold[n1], v2 = wfold[m1 + HB], v3 = wfold[m0 + HB];
        const float2 G0 = make_float2(S2.w * v0.x + B3.x * v0.y, S0.w * v0.z + B1.x * v0.w);
        const float2 G1 = make_float2(S2.y * v1.x + B3.z * v1.y, S0.y * v1.z + B1.z * v1.w);
        const float2 G2 = make_float2(S0.z * v2.x + B1.y * v2.y, S2.z * v2.z + B3.y * v2.w);
        const float2 G3 = make_float2(S0.x * v3.x + B1.w * v3.y, S2.x * v3.z + B3.w * v3.w);
        const float h = 0.70710678118654752440f;
        auto one = [&](int n, float2 lo, float2 hi) {
            const float2 d = cmulk(hi, h, -h);   // F[n + 512] exp(-i pi / 4)
            buf[phys(n)] = band ? cmul(csub(lo, d), bt_l[n]) : cmul(cadd(lo, d), gb_l[(n & 1) * HB + (n >> 1)]);
        };
        one(m0, F0, G3);
        one(m1, F1, G2);
        one(n0, G0, F3);
        one(n1, G1, F2);
    };
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
    float cva[ITER], cvb[ITER];   // the thread's pairs of the previous tile
#pragma unroll
    for (int i = 0; i < ITER; ++i) cva[i] = cvb[i] = 0.f;
    bool have_prev = false;
    fetch(clip, j, 0);
    // the walk's next tile (recomputed where it is needed -- for the request of its first half frame and at the end of the tile -- instead of
    // kept across the folds and transforms: at 128 VGPRs the kernel spilled six registers, and a scratch reload waits for every load in flight)
    auto next_of = [&](int& nclip, int& nband, int& nj, int& nj1, int& nv) -> bool {
        nclip = clip, nband = band, nj = j + 1, nj1 = j1, nv = v;
        if (nj >= j1) {
            nv = v + gridDim.x;
            if (nv >= units) return false;
            unit_of(nv, nclip, nband, nj, nj1);
        }
        return true;
    };
    for (;;) {
        int po = p;
        asm volatile("" : "+v"(po));
#pragma unroll
        for (int f = 0; f < FPW; ++f) {
            float2* buf = frames + frame_of(f) * C::PITCH;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                fold(buf, r, band);
                const int h = 2 * f + r + 1;
                if (h < 2 * FPW) {
                    fetch(clip, j, h);
                } else {
                    int nclip, nband, nj, nj1, nv;
                    if (next_of(nclip, nband, nj, nj1, nv)) fetch(nclip, nj, 0);
                }
            }
            frame_sync<P>();
            float2 vv[E];
            regs_read<9, 3>(vv, buf, po);
            frame_sync<P>();
            fft_frame_post<9, 3>(vv, buf, po, tw_l, gb_l + band * HB);   // the frame now holds conj(y[2q + band])
        }
        lds_barrier();
        {
            int tido = tid;
            asm volatile("" : "+v"(tido));
            const int tp = tido % 16, fq = tido / 16;
            const int t0 = j * FPB, ta = t0 + 2 * tp;
            const int e = fq & 1, qi = fq >> 1;
            const float* pa = reinterpret_cast<const float*>(frames + (2 * tp) * C::PITCH + phys(qi)) + e;
            const float* pb = pa + 2 * C::PITCH;
            constexpr int DPH = 2 * (32 + 32 / 16);
            const int f0 = e ? M - 1 - 4 * qi - 2 * band : 4 * qi + 2 * band;
            float* o = out + (long long)clip * M * TP + ta;
            const bool cur_ok = ta < T, last = j + 1 >= j1;
            // phase of a row's run in its line, in floats: (array + (clip M + f) TP) mod 32
            const int b0 = (int)((reinterpret_cast<uintptr_t>(out) >> 2) & 31), c0 = (int)(((long long)clip * M) & 31), tpm = TP & 31;
            const bool pairs = TP % 2 == 0 && reinterpret_cast<uintptr_t>(out) % 8 == 0;
            // (the thread's rows are 128 apart: the phase of their runs, (b0 + (c0 + f) tpm) mod 32, is the same for all sixteen)
            const int a = (b0 + (c0 + f0) * tpm) & 31;
            const long long dstep = (long long)(e ? -128 : 128) * TP;
            auto sweep = [&](auto stream) {
                constexpr bool ST = decltype(stream)::value;
                float* dst = o + (long long)f0 * TP;
#pragma unroll
                for (int i = 0; i < ITER; ++i, dst += dstep) {
                    const float va = pa[i * DPH], vb = pb[i * DPH];
                    if (pairs) {   // (uniform) even T: a is even, a lane's pair lies in one line
                        const bool from_prev = tp >= 16 - (a >> 1);   // (a = 0: never)
                        const float2 val = from_prev ? make_float2(cva[i], cvb[i]) : make_float2(va, vb);
                        if (from_prev ? have_prev : cur_ok) {
                            if constexpr (ST) store_stream(reinterpret_cast<float2*>(dst + (from_prev ? -32 : 0)), val);
                            else *reinterpret_cast<float2*>(dst + (from_prev ? -32 : 0)) = val;
                        }
                        if (last && from_prev && cur_ok) *reinterpret_cast<float2*>(dst) = make_float2(va, vb);   // tail of the segment's last run
                    } else {
                        // odd T: rows start at any float, a pair may straddle the line boundary: the two floats of a lane choose for themselves,
                        // as two 4-byte stores -- two instructions of the same wave, back to back, that together cover the line (one 8-byte
                        // store per lane at a 4-byte aligned address, with two floats only for the straddling lane: 1.38 against 1.33 ms)
                        const bool pa_ = 2 * tp >= 32 - a, pb_ = 2 * tp + 1 >= 32 - a;
                        if (pa_ ? have_prev : cur_ok) dst[pa_ ? -32 : 0] = pa_ ? cva[i] : va;
                        if (pb_ ? have_prev : (ta + 1 < T)) dst[pb_ ? -31 : 1] = pb_ ? cvb[i] : vb;
                        if (last && pa_ && cur_ok) dst[0] = va;
                        if (last && pb_ && ta + 1 < T) dst[1] = vb;
                    }
                    cva[i] = va;
                    cvb[i] = vb;
                }
            };
            if (have_prev) sweep(std::true_type{});
            else sweep(std::false_type{});
        }
        lds_barrier();
        int nclip, nband, nj, nj1, nv;
        if (!next_of(nclip, nband, nj, nj1, nv)) break;
        have_prev = nj != 0 && nv == v;   // the walk continues inside the same segment (and band)
        clip = nclip, band = nband, j = nj, j1 = nj1, v = nv;
    }
}

// ---------------------------------------------------------------------------------
// inverse
// ---------------------------------------------------------------------------------
// Persistent carry form (as k_istft_ft16): one workgroup per CU walks the FPB-frame tiles of a clip
// segment in order; the second-half contribution of a tile's last frame (M samples, "carry") stays
// in LDS for the next tile.  Tiles start at t = FPB * tile, so the time-minor gather reads aligned
// 128-B runs and no frame is read twice (the halo form re-read one frame in 32 and straddled two
// lines per run).  A segment that does not start a clip first runs the tile before it in carry-only
// mode (its last frame alone).  Barriers order LDS only; the output stores are not waited for.
// WINL: the window (4 NF floats) is staged in LDS; false (W = 4096): it is read from global memory -- the sweep form of the
// overlap-add reads a thread's two window pairs once per tile -- which makes room for 16-frame tiles (64-byte gather runs) instead of 8.
template <int LOG2NF, int LOG2E, int FPB, int NSLOT, int LAYOUT, bool WINL = true>
__global__ __launch_bounds__(NSLOT * fft_threads(LOG2NF, LOG2E)) void k_imdct(
    const float* __restrict__ coefs, const float* __restrict__ win, const float2* __restrict__ twp,
    const float2* __restrict__ tw8g, float* __restrict__ y, int T, int TP, long long out_len, int tiles, int segs, int seg_tiles,
    int total_units) {
    using C = FftCfg<LOG2NF, LOG2E>;
    constexpr int NF = C::N, M = 2 * NF, P = C::P, E = C::E, NT = NSLOT * P;
    static_assert(NT % FPB == 0 || LAYOUT == ZAFX_LAYOUT_TF, "time-minor gather needs NT to be a multiple of FPB");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* frames = reinterpret_cast<float2*>(smem_raw);
    float2* tw_l = frames + FPB * C::PITCH;
    float2* tw8 = tw_l + C::TW;                         // g_m, NF entries
    float* win_s = reinterpret_cast<float*>(tw8 + NF);  // window, 4 NF floats (WINL)
    float* carry = win_s + (WINL ? 4 * NF : 0);         // M floats
    const float* win_l = win;
    if constexpr (WINL) win_l = win_s;
    const int tid = threadIdx.x;
    for (int i = tid; i < C::TW; i += NT) tw_l[i] = twp[i];
    for (int i = tid; i < NF; i += NT) tw8[i] = tw8g[i];
    if constexpr (WINL)
        for (int i = tid; i < 4 * NF; i += NT) win_s[i] = win[i];
    lds_barrier();
    const float gain = 2.f / (float)M;
    // (buffer loads: 16 bytes per lane at ANY 4-byte alignment -- rows off the line grid, T % 4 != 0, took the 4-byte path before, four times
    // the load instructions and no prefetch: T = 433 1.31 ms against 0.76 at T = 432 -- and a piece that runs past the clip's last row reads 0)
    const bool vec4 = LAYOUT == ZAFX_LAYOUT_FT && FPB % 4 == 0 && NT % (FPB / 4) == 0 && NF >= NT / (FPB / 4) && (long long)M * TP * 4 < (1LL << 32) &&
                      reinterpret_cast<uintptr_t>(coefs) % 4 == 0;

    // 16-byte gathers (vec4): a lane reads 4 adjacent frames of a row (8 lanes per 128-B run; the CU's vector-memory queue
    // holds ~64 wave-level loads whatever their width, 4-byte lanes leave it carrying 256 B per entry).  The rows of the
    // NEXT tile are requested as soon as this tile's are folded into LDS (KI x 2 x 16 B per lane in registers), so the
    // gather -- a third of the tile time when it ran in phase A -- flies under the FFT and the overlap-add.
#ifndef ZAFX_IMDCT_PREFETCH
#define ZAFX_IMDCT_PREFETCH 1
#endif
    // Overlap-add of a full tile as a sweep (phase C): FR = 2 NT / M frames per pass of the workgroup, NIT passes, one 8-byte store
    // per thread and pass.
    constexpr bool SWEEP = NF >= 128 && (2 * NT) % M == 0 && FPB % ((2 * NT) / M > 0 ? (2 * NT) / M : 1) == 0;
    constexpr int FR = SWEEP ? (2 * NT) / M : 1, NIT = FPB / FR;
    constexpr int LPR = FPB >= 4 ? FPB / 4 : 1, MSTEP = NT / LPR;
    constexpr int KI = (NF % MSTEP == 0 && NF >= MSTEP) ? NF / MSTEP : 1;
    constexpr bool PRE = ZAFX_IMDCT_PREFETCH && LAYOUT == ZAFX_LAYOUT_FT && KI <= 4;
    const int fs4 = (tid % LPR) * 4, mq4 = tid / LPR;
    float4 pre_re[KI], pre_im[KI];
    bool pre_ok = false;
    bool converted = false;   // pre_re / pre_im hold the pre-twiddled FFT inputs (convert_rows), not the raw rows
    auto gather4 = [&](int unit_n, int tile_n, int part = 2) {   // rows of my 4 frames of tile_n of unit_n -> registers (part 0: the rows 2m, 1: the rows M-1-2m, 2: both)
        pre_ok = false;
        converted = false;
        if (!vec4 || unit_n >= total_units) return;
        const int tile_a_n = (unit_n % segs) * seg_tiles;
        const int first_needed_n = tile_n < tile_a_n ? FPB - 1 : 0;
        const int t = tile_n * FPB + fs4;
        if (t < T && fs4 + 3 >= first_needed_n) {   // pitch % 4 == 0: the four frames lie in the row (those past T are not used)
            pre_ok = true;
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(coefs + (long long)(unit_n / segs) * M * TP, (unsigned)((long long)M * TP * 4));
#pragma unroll
            for (int i = 0; i < KI; ++i) {
                const int m = mq4 + i * MSTEP;
                if (part != 1) pre_re[i] = buf_load_f32x4(rs, (int)(((unsigned)(2 * m) * (unsigned)TP + (unsigned)t) * 4u));
                if (part != 0) pre_im[i] = buf_load_f32x4(rs, (int)(((unsigned)(M - 1 - 2 * m) * (unsigned)TP + (unsigned)t) * 4u));
            }
        }
    };
#ifndef ZAFX_IMDCT_CONVERT_EARLY
#define ZAFX_IMDCT_CONVERT_EARLY 1
#endif
    // The requested rows are turned into the packed, pre-twiddled FFT inputs (same registers: pre_re[i] = (c0, c1), pre_im[i] =
    // (c2, c3) of the lane's four frames) at the START of the overlap-add, i.e. before this tile's stores are issued: the wait for
    // the rows is then a wait for loads only.  Consumed in the next phase A -- behind the stores -- it was a wait for every store
    // to be acknowledged (the compiler cannot count the stores in between, profiles/r02_notes.md): 0.4-4.5 k cycles per tile.
    auto convert_rows = [&]() {
        if (!pre_ok || converted) return;
#pragma unroll
        for (int i = 0; i < KI; ++i) {
            const int m = mq4 + i * MSTEP;
            const float4 re = pre_re[i], im = pre_im[i];
            const float2 g = tw8[m];
            const float2 c0 = cmul(make_float2(re.x, im.x), g), c1 = cmul(make_float2(re.y, im.y), g);
            const float2 c2 = cmul(make_float2(re.z, im.z), g), c3 = cmul(make_float2(re.w, im.w), g);
            pre_re[i] = make_float4(c0.x, c0.y, c1.x, c1.y);
            pre_im[i] = make_float4(c2.x, c2.y, c3.x, c3.y);
        }
        converted = true;
    };
    auto first_tile = [&](int unit_n) {
        const int ta = (unit_n % segs) * seg_tiles;
        return ta > 0 ? ta - 1 : 0;
    };
    if constexpr (PRE) gather4(blockIdx.x, blockIdx.x < total_units ? first_tile(blockIdx.x) : 0);
    // Frame-major input: a frame's M coefficients are contiguous; lane p of the wave that owns a frame takes the pairs
    // (m, NF-1-m): the two 8-byte reads X[2m..2m+1] and X[M-2-2m..M-1-2m] hold both packed inputs
    // c[m] = X[2m] + i X[M-1-2m] and c[NF-1-m] = X[M-2-2m] + i X[2m+1] -- coalesced 512-B runs instead of 4-byte
    // reads of every second float -- and the next tile's ride in registers as above.
    constexpr bool PRE_TF = ZAFX_IMDCT_PREFETCH && LAYOUT == ZAFX_LAYOUT_TF && P == 64 && FPB == 2 * NSLOT && NF >= 2 * P && NF / 2 / P <= 4;
    constexpr int KP = PRE_TF ? NF / 2 / P : 1;
    float2 pa[2][KP], pb[2][KP];
    bool tf_ok[2] = {false, false};
    const bool vec_tf = reinterpret_cast<uintptr_t>(coefs) % 8 == 0;
    auto gather_tf = [&](int unit_n, int tile_n) {
        tf_ok[0] = tf_ok[1] = false;
        if (!vec_tf || unit_n >= total_units) return;
        const int tile_a_n = (unit_n % segs) * seg_tiles;
        const int first_needed_n = tile_n < tile_a_n ? FPB - 1 : 0;
        const int pl = tid % P;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int fsn = tid / P + j * NSLOT, t = tile_n * FPB + fsn;
            if (t < T && fsn >= first_needed_n) {
                tf_ok[j] = true;
                const float* cp = coefs + ((long long)(unit_n / segs) * T + t) * M;
#pragma unroll
                for (int i = 0; i < KP; ++i) {
                    const int m = pl + i * P;
                    pa[j][i] = *reinterpret_cast<const float2*>(cp + 2 * m);
                    pb[j][i] = *reinterpret_cast<const float2*>(cp + M - 2 - 2 * m);
                }
            }
        }
    };
    if constexpr (PRE_TF) gather_tf(blockIdx.x, blockIdx.x < total_units ? first_tile(blockIdx.x) : 0);

    PROF_INIT(g_prof_imdct);
    for (int unit = blockIdx.x; unit < total_units; unit += gridDim.x) {
    const int clip = unit / segs, seg = unit % segs;
    const int tile_a = seg * seg_tiles, tile_b = min(tile_a + seg_tiles, tiles);
    for (int c = tid; c < M; c += NT) carry[c] = 0.f;
    for (int tile = tile_a > 0 ? tile_a - 1 : 0; tile < tile_b; ++tile) {
    const bool carry_only = tile < tile_a;
    const int t_first = tile * FPB;
    const int first_needed = carry_only ? FPB - 1 : 0;
    PROF_MARK(0);

    // ---- phase A: c[m] = (X[2m] + i X[M-1-2m]) g_m  -> LDS (natural order)
    if (PRE_TF && vec_tf) {
        if constexpr (PRE_TF) {
            const int pl = tid % P;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (!tf_ok[j]) continue;
                float2* fb = frames + (tid / P + j * NSLOT) * C::PITCH;
#pragma unroll
                for (int i = 0; i < KP; ++i) {
                    const int m = pl + i * P, m2 = NF - 1 - m;
                    fb[phys(m)] = cmul(make_float2(pa[j][i].x, pb[j][i].y), tw8[m]);
                    fb[phys(m2)] = cmul(make_float2(pb[j][i].x, pa[j][i].y), tw8[m2]);
                }
            }
        }
    } else if constexpr (LAYOUT == ZAFX_LAYOUT_TF) {
        const int mq = tid % P;
        for (int fs = tid / P; fs < FPB; fs += NSLOT) {
            const int t = t_first + fs;
            if (t >= T || fs < first_needed) continue;
            float2* fb = frames + fs * C::PITCH;
            const float* cp = coefs + ((long long)clip * T + t) * M;
            for (int m = mq; m < NF; m += P) fb[phys(m)] = cmul(make_float2(cp[2 * m], cp[M - 1 - 2 * m]), tw8[m]);
        }
    } else if (vec4) {
        if constexpr (!PRE) gather4(unit, tile);
        if (pre_ok) {
            convert_rows();   // (already done at the start of the previous overlap-add, except for the first tile of the launch)
            float2* fb = frames + fs4 * C::PITCH;
#pragma unroll
            for (int i = 0; i < KI; ++i) {
                const int m = mq4 + i * MSTEP;
                fb[phys(m)] = make_float2(pre_re[i].x, pre_re[i].y);
                fb[C::PITCH + phys(m)] = make_float2(pre_re[i].z, pre_re[i].w);
                fb[2 * C::PITCH + phys(m)] = make_float2(pre_im[i].x, pre_im[i].y);
                fb[3 * C::PITCH + phys(m)] = make_float2(pre_im[i].z, pre_im[i].w);
            }
        }
    } else {
        const int fs = tid % FPB, mq = tid / FPB;   // lanes run along t: FPB * 4 B contiguous per row
        const int t = t_first + fs;
        if (t < T && fs >= first_needed) {
            float2* fb = frames + fs * C::PITCH;
            const float* cp = coefs + (long long)clip * M * TP + t;
#pragma unroll 4
            for (int m = mq; m < NF; m += NT / FPB) {
                const float re = cp[(long long)(2 * m) * TP];
                const float im = cp[(long long)(M - 1 - 2 * m) * TP];
                fb[phys(m)] = cmul(make_float2(re, im), tw8[m]);
            }
        }
    }
    PROF_MARK(1);
    lds_barrier();
    PROF_MARK(2);
#ifndef ZAFX_IMDCT_STAGGER
#define ZAFX_IMDCT_STAGGER 2
#endif
    // The next tile of this workgroup (same unit, or the first tile of its next unit) is requested in two halves: every wave asks for
    // its rows 2m now and for its rows M-1-2m when the first of its two frames is transformed -- 16 waves x 8 loads at once overrun
    // the CU's vector-memory queue and the youngest waves start their transforms behind it (1.083 ms); half of the waves now and
    // the other half after their transforms: 1.058; this form: 1.042.
    constexpr int STAGGER = ZAFX_IMDCT_STAGGER;
    const bool gather_now = STAGGER != 1 || tid < NT / 2;
    auto gather_next = [&](int part = 2) {
        int unit_n = unit, tile_n = tile + 1;
        if (tile_n >= tile_b) {
            unit_n = unit + gridDim.x;
            tile_n = unit_n < total_units ? first_tile(unit_n) : 0;
        }
        gather4(unit_n, tile_n, part);
    };
    if constexpr (PRE) {
        if (gather_now) gather_next(STAGGER == 2 ? 0 : 2);
    }
    bool second_half = false;
    if constexpr (PRE_TF) {
        int unit_n = unit, tile_n = tile + 1;
        if (tile_n >= tile_b) {
            unit_n = unit + gridDim.x;
            tile_n = unit_n < total_units ? first_tile(unit_n) : 0;
        }
        gather_tf(unit_n, tile_n);
    }

    // ---- phase B: FFT with the DCT-IV post-twiddle in its last pass
    {
        const int p = tid % P;
#pragma unroll 1
        for (int slot = tid / P; slot < FPB; slot += NSLOT) {
            if (P <= 64 && slot < first_needed) continue;   // (frames wider than a wave synchronise with s_barrier: no skipping)
            float2* buf = frames + slot * C::PITCH;
            float2 v[E];
            regs_read<LOG2NF, LOG2E>(v, buf, p);
            frame_sync<P>();
            // DCT-IV post-twiddle folded into the last pass (as k_mdct_ft32): slot k holds conj(y_k g_k), and the 2 NF reals of the
            // frame are read from it where they are needed, u[2k] = slot[k].x, u[2k+1] = -Im(y g)[NF-1-k] = slot[NF-1-k].y
            // (upair).  The separate sweep over the pairs (k, NF-1-k) -- four 8-byte LDS reads and two writes per pair,
            // a quarter of the phase's LDS time -- is gone.
            fft_frame_post<LOG2NF, LOG2E>(v, buf, p, tw_l, tw8);
            if constexpr (PRE && ZAFX_IMDCT_STAGGER == 2) {
                if (!second_half) gather_next(1);   // (the wave's first frame is done: the other half of its requests)
                second_half = true;
            }
        }
    }
    if constexpr (PRE) {
        if (STAGGER == 1 && !gather_now) gather_next();
        if (STAGGER == 2 && !second_half) gather_next(1);   // (a wave without a frame to transform in this tile)
    }
    PROF_MARK(3);
    lds_barrier();
    PROF_MARK(4);
    if constexpr (PRE && ZAFX_IMDCT_CONVERT_EARLY) convert_rows();

    // ---- phase C: unfold + window + TDAC overlap-add of the 2 covering frames (older first, as the
    //      reference's loop), trim (zaf.py:1166-1182); then the carry for the next tile
    {
        const int n_valid = min(FPB, T - t_first);
        // Two samples (n1, n1 + 1) at a time: each of the four operands is one aligned 8-byte LDS read (the
        // reversed halves of the unfold come out swapped), the result one 8-byte store.
        const float2* frames2 = frames;
        const float2* win2 = reinterpret_cast<const float2*>(win_l);
        float2* carry2 = reinterpret_cast<float2*>(carry);
        auto upair = [&](const float2* fr, int k) {   // (u[2k], u[2k+1]) of a transformed frame
            return make_float2(fr[phys(k)].x, fr[phys(NF - 1 - k)].y);
        };
        auto older2 = [&](int j, int n1) {   // frame j's contribution to samples n1, n1 + 1 of the next frame's span
            const int n0 = n1 + M;
            const float2* fr = frames2 + (size_t)j * C::PITCH;
            float2 u;
            if (n0 < 3 * NF) {
                const float2 v = upair(fr, (3 * NF - 2 - n0) >> 1);   // floats (3NF-2-n0, 3NF-1-n0)
                u = make_float2(-v.y, -v.x);
            } else {
                const float2 v = upair(fr, (n0 - 3 * NF) >> 1);
                u = make_float2(-v.x, -v.y);
            }
            const float2 w = win2[n0 >> 1];
            return make_float2(u.x * w.x, u.y * w.y);
        };
        if (!carry_only) {
            const int c_end2 = (tile == tiles - 1 ? (n_valid + 1) * M : FPB * M) / 2;
            float* yc = y + (long long)clip * out_len;
            const long long o_first = (long long)t_first * M - M;
            const bool y_aligned = (reinterpret_cast<uintptr_t>(yc) % 8) == 0;
            // A full tile inside the clip: thread tid owns the sample pair n1 = 2 tid mod M of the frames h, h + FR, ... (FR = 2 NT / M
            // frames per sweep of the workgroup), so which half of the unfold it reads, its two window pairs and its slots in a
            // frame are fixed and a sweep is two 8-byte LDS reads, three packed operations and one 8-byte store at constant strides
            // (the general loop below re-derives all of that per pair and waits for each LDS read in turn: 13 k of the tile's
            // 38 k cycles at W = 2048).  Same operations in the same order: bit-identical to the general loop.
            const bool last_tile = tile == tiles - 1;   // (partial: frames up to n_valid, one more span with the older term only, trimmed at out_len)
            if (SWEEP && (last_tile || (n_valid == FPB && o_first + (long long)FPB * M <= out_len))) {
                int to = tid;   // opaque: the slots and window pairs are recomputed per tile (carried through the transforms they spill)
                asm volatile("" : "+v"(to));
                const int n1 = (2 * to) & (M - 1);
                const int h = __builtin_amdgcn_readfirstlane((2 * to) / M);           // (whole waves: NF >= 128)
                const bool lo = __builtin_amdgcn_readfirstlane(n1 < NF ? 1 : 0) != 0;
                const int kc = lo ? (NF + n1) >> 1 : (3 * NF - 2 - n1) >> 1, ko = lo ? (NF - 2 - n1) >> 1 : (n1 - NF) >> 1;   // the pairs' k
                const float* pcx = reinterpret_cast<const float*>(frames2 + (size_t)h * C::PITCH + phys(kc));
                const float* pcy = reinterpret_cast<const float*>(frames2 + (size_t)h * C::PITCH + phys(NF - 1 - kc)) + 1;
                const float* pox = reinterpret_cast<const float*>(frames2 + (size_t)h * C::PITCH - C::PITCH + phys(ko));
                const float* poy = reinterpret_cast<const float*>(frames2 + (size_t)h * C::PITCH - C::PITCH + phys(NF - 1 - ko)) + 1;
                const float2 wc = win2[n1 >> 1], wo = win2[(n1 + M) >> 1];
                float2* dst = reinterpret_cast<float2*>(yc + o_first) + to;
                const bool skip0 = t_first == 0 && h == 0;   // the first M samples of the clip's first tile are trimmed
                auto sweep = [&](auto LO, auto LAST, auto ALIGNED8) {
                    constexpr bool L = decltype(LAST)::value, A8 = decltype(ALIGNED8)::value;
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        const int j = FR * it + h;   // (uniform) the frame of this pass
                        if (L && j > n_valid) continue;
                        const bool cur = !L || j < n_valid;
                        const float2 c = cur ? make_float2(pcx[(size_t)it * FR * C::PITCH * 2], pcy[(size_t)it * FR * C::PITCH * 2]) : make_float2(0.f, 0.f);
                        float2 t;
                        if (it == 0 && h == 0) {
                            t = carry2[n1 >> 1];
                        } else {
                            const float2 o = make_float2(pox[(size_t)it * FR * C::PITCH * 2], poy[(size_t)it * FR * C::PITCH * 2]);
                            t = decltype(LO)::value ? make_float2(-o.y * wo.x, -o.x * wo.y) : make_float2(-o.x * wo.x, -o.y * wo.y);
                        }
                        float2 a = t;
                        if (cur)
                            a = decltype(LO)::value ? make_float2(t.x + c.x * wc.x, t.y + c.y * wc.y) : make_float2(t.x + -c.y * wc.x, t.y + -c.x * wc.y);
                        if (!(it == 0 && skip0)) {
                            // (the reference's output length (T-1) M - 1 is odd: every second clip of a batch starts on a 4-byte boundary)
                            const long long o = o_first + 2 * to + (long long)it * 2 * NT;
                            // (compiled per alignment: with the test at every store the compiler folded both forms into 4-byte stores)
                            if (A8 && (!L || o + 1 < out_len)) {
                                dst[(size_t)it * NT] = make_float2(a.x * gain, a.y * gain);
                            } else {
                                float* d1 = reinterpret_cast<float*>(dst + (size_t)it * NT);
                                if (!L || o < out_len) d1[0] = a.x * gain;
                                if (!L || o + 1 < out_len) d1[1] = a.y * gain;
                            }
                        }
                    }
                };
                auto sweep_lo = [&](auto LAST, auto ALIGNED8) {
                    if (lo) sweep(std::true_type{}, LAST, ALIGNED8);
                    else sweep(std::false_type{}, LAST, ALIGNED8);
                };
                if (last_tile) {
                    if (y_aligned) sweep_lo(std::true_type{}, std::true_type{});
                    else sweep_lo(std::true_type{}, std::false_type{});
                } else {
                    if (y_aligned) sweep_lo(std::false_type{}, std::true_type{});
                    else sweep_lo(std::false_type{}, std::false_type{});
                }
            } else {
            for (int c2 = tid; c2 < c_end2; c2 += NT) {
                const int c = 2 * c2, j1 = c / M, n1 = c % M;
                float2 acc = j1 >= 1 ? older2(j1 - 1, n1) : carry2[n1 >> 1];
                if (j1 < n_valid) {
                    const float2* fr = frames2 + (size_t)j1 * C::PITCH;
                    float2 u;
                    if (n1 < NF) {
                        u = upair(fr, (NF + n1) >> 1);
                    } else {
                        const float2 v = upair(fr, (3 * NF - 2 - n1) >> 1);
                        u = make_float2(-v.y, -v.x);
                    }
                    const float2 w = win2[n1 >> 1];
                    acc = make_float2(acc.x + u.x * w.x, acc.y + u.y * w.y);
                }
                const long long o = o_first + c;
                if (o >= 0) {
                    if (y_aligned && o + 1 < out_len) {
                        *reinterpret_cast<float2*>(yc + o) = make_float2(acc.x * gain, acc.y * gain);
                    } else {
                        if (o < out_len) yc[o] = acc.x * gain;
                        if (o + 1 < out_len) yc[o + 1] = acc.y * gain;
                    }
                }
            }
            }
        }
        if (tile + 1 < tile_b) {   // then n_valid == FPB
            for (int c2 = tid; c2 < M / 2; c2 += NT) carry2[c2] = older2(FPB - 1, 2 * c2);
        }
    }
    PROF_MARK(5);
    lds_barrier();   // the next tile overwrites the frame buffers
    }
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

// int16 PCM (one or two channels) straight into k_mdct_ft32: W = 2048, reference layout, clips of a multiple of four frames (zafx_execute_pcm)
bool mdct_pcm_direct_ok(const zafx_plan& pl, int64_t n_frames, int n_channels, int sample_bytes, const void* d_pcm) {
    return sample_bytes == 2 && (n_channels == 1 || n_channels == 2) && pl.kind == ZAFX_MDCT && pl.prm.precision == ZAFX_PRECISION_F32 && pl.bs_log2m == 0 &&
           pl.log2nf == 9 && pl.layout == ZAFX_LAYOUT_FT && n_frames % 4 == 0 && n_frames < (1LL << 28) && reinterpret_cast<uintptr_t>(d_pcm) % 16 == 0;
}

constexpr bool mdct_use_persistent(int log2nf, int layout) {
    return log2nf >= 7 && log2nf <= 9;   // either layout
}

template <int LOG2NF, bool TFOUT>
static hipError_t run_mdct_p(const zafx_plan& pl, const float* x, float* out, int64_t n_clips, int64_t n_samples, int T) {
    constexpr int LOG2E = default_log2e(LOG2NF);
    using G = MdctPCfg<LOG2NF, LOG2E>;
    const bool aligned = n_samples % 4 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 && n_samples < (1LL << 28);   // (32-bit byte offsets inside a clip)
    auto kern = aligned ? k_mdct_ft32<LOG2NF, LOG2E, true, G::NSLOT, TFOUT> : k_mdct_ft32<LOG2NF, LOG2E, false, G::NSLOT, TFOUT>;
    const int pcm = take_pcm_mode();   // (zafx_execute_pcm: int16 in the loads; mdct_pcm_direct_ok vouches for `aligned`, W = 2048 and the reference layout)
    if constexpr (LOG2NF == 9 && !TFOUT) {
        if (pcm == 1) kern = k_mdct_ft32<LOG2NF, LOG2E, true, G::NSLOT, false, false, 1>;
        if (pcm == 2) kern = k_mdct_ft32<LOG2NF, LOG2E, true, G::NSLOT, false, false, 2>;
    }
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, G::SMEM); e != hipSuccess) return e;
    const int tiles = (T + kMdctTile - 1) / kMdctTile;
    const long long total = (long long)tiles * n_clips;
    if (total <= 0) return hipSuccess;
    const int per_cu = (int)std::min<size_t>(2, (size_t)kMaxLdsBytes / G::SMEM);
    const long long max_grid = (long long)pl.n_cus * std::max(per_cu, 1);
#ifndef ZAFX_MDCT_CARRY
#define ZAFX_MDCT_CARRY 1
#endif
    if constexpr (!TFOUT && ZAFX_MDCT_CARRY) {
        // Rows that are not even whole 64-byte half lines (T % 16 != 0; odd T with 4-byte stores): the carry form -- 1024 clips: T = 434 / 436 / 440 run
        // 0.84 ms (4.3 TB/s) with it against 1.37 / 1.29 / 1.09 ms without.  At T % 16 == 0 (the benchmark's 432: odd rows start 64 bytes
        // into a line and leave as two streamed half lines) the plain form is faster, 0.771 against 0.826 ms: the carry takes the
        // registers of the resident window quadruples, worth 10 % on this LDS-bound kernel.
        const int TP = (int)row_pitch(pl, T);
        if (aligned && (TP % 16 != 0 || reinterpret_cast<uintptr_t>(out) % 64 != 0) && total < (1LL << 31)) {
            auto kc = k_mdct_ft32<LOG2NF, LOG2E, true, G::NSLOT, false, true>;
            if constexpr (LOG2NF == 9) {
                if (pcm == 1) kc = k_mdct_ft32<LOG2NF, LOG2E, true, G::NSLOT, false, true, 1>;
                if (pcm == 2) kc = k_mdct_ft32<LOG2NF, LOG2E, true, G::NSLOT, false, true, 2>;
            }
            if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kc), pl.device, G::SMEM); e != hipSuccess) return e;
            const int segs = carry_segments(n_clips, tiles, max_grid);
            const int seg_tiles = (tiles + segs - 1) / segs;
            const long long units = (long long)n_clips * segs;
            pl.ran = "k_mdct_ft32";
            hipLaunchKernelGGL(kc, dim3((unsigned)std::min<long long>(units, max_grid)), dim3(G::NSLOT * 64), G::SMEM, pl.stream, x, pl.d_wfold, pl.d_tw_pass,
                               pl.d_tw_aux, out, (long long)n_samples, T, TP, tiles, (int)total, segs, seg_tiles, (int)units);
            return hipGetLastError();
        }
    }
    const long long grid = std::min<long long>(total, max_grid);
    pl.ran = "k_mdct_ft32";
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(G::NSLOT * 64), G::SMEM, pl.stream, x, pl.d_wfold, pl.d_tw_pass, pl.d_tw_aux, out,
                       (long long)n_samples, T, (int)row_pitch(pl, T), tiles, (int)total, 1, 0, 0);
    return hipGetLastError();
}

#ifndef ZAFX_MDCT_BAND
#define ZAFX_MDCT_BAND 1
#endif
static hipError_t run_mdct_band(const zafx_plan& pl, const float* x, float* out, int64_t n_clips, int64_t n_samples, int T) {
    using G = MdctBandCfg;
    auto kern = k_mdct_ft32b;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, G::SMEM); e != hipSuccess) return e;
    const int tiles = (T + G::FPB - 1) / G::FPB;
    const long long total = (long long)tiles * n_clips;
    if (total <= 0) return hipSuccess;
    const long long grid = std::min<long long>(total, (long long)pl.n_cus);
    pl.ran = "k_mdct_ft32b";
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(G::NT), G::SMEM, pl.stream, x, pl.d_wfold, pl.d_tw_sub, pl.d_tw_band, out, (long long)n_samples, T,
                       (int)row_pitch(pl, T), tiles, (int)total);
    return hipGetLastError();
}

// k_mdct_ft32q: W = 8192 in the reference layout, four bands of bins per 32-frame tile (see the kernel)
#ifndef ZAFX_MDCT_QUAD
#define ZAFX_MDCT_QUAD 1
#endif
static hipError_t run_mdct_quad(const zafx_plan& pl, const float* x, float* out, int64_t n_clips, int64_t n_samples, int T) {
    using G = MdctQuadCfg;
    const bool aligned = n_samples % 2 == 0 && reinterpret_cast<uintptr_t>(x) % 8 == 0;   // sample pairs as 8-byte loads
    auto kern = aligned ? k_mdct_ft32q<true> : k_mdct_ft32q<false>;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, G::SMEM); e != hipSuccess) return e;
    const int tiles = (T + G::FPB - 1) / G::FPB;
    const long long total = (long long)tiles * n_clips;
    if (total <= 0) return hipSuccess;
    const long long grid = std::min<long long>(total, (long long)pl.n_cus);
    pl.ran = "k_mdct_ft32q";
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(G::NT), G::SMEM, pl.stream, x, pl.d_window, pl.d_tw_sub, pl.d_tw_aux, pl.d_tw_quad, out, (long long)n_samples, T,
                       (int)row_pitch(pl, T), tiles, (int)total);
    return hipGetLastError();
}

#ifndef ZAFX_MDCT_BAND_CARRY
#define ZAFX_MDCT_BAND_CARRY 1
#endif
static hipError_t run_mdct_band_carry(const zafx_plan& pl, const float* x, float* out, int64_t n_clips, int64_t n_samples, int T) {
    using G = MdctBandCfg;
    auto kern = k_mdct_ft32bc;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, G::SMEM); e != hipSuccess) return e;
    const int tiles = (T + G::FPB - 1) / G::FPB;
    if ((long long)tiles * n_clips <= 0) return hipSuccess;
    const long long max_grid = pl.n_cus;
    const int segs = carry_segments(2 * n_clips, tiles, max_grid);   // (two units -- one per band -- for every segment)
    const int seg_tiles = (tiles + segs - 1) / segs;
    const long long units = 2LL * n_clips * segs;
    pl.ran = "k_mdct_ft32bc";
    hipLaunchKernelGGL(kern, dim3((unsigned)std::min<long long>(units, max_grid)), dim3(G::NT), G::SMEM, pl.stream, x, pl.d_wfold, pl.d_tw_sub, pl.d_tw_band, out,
                       (long long)n_samples, T, (int)row_pitch(pl, T), tiles, segs, seg_tiles, (int)units);
    return hipGetLastError();
}

template <int LOG2NF, int LAYOUT>
static hipError_t run_mdct(const zafx_plan& pl, const float* x, float* out, int64_t n_clips, int64_t n_samples, int T) {
    if constexpr (ZAFX_MDCT_BAND && ZAFX_MDCT_BAND_CARRY && LOG2NF == 10 && LAYOUT == ZAFX_LAYOUT_FT) {
        // rows that are not whole 64-byte half lines: the carry form (one band per workgroup)
        const int64_t TP = row_pitch(pl, T);
        if (pl.d_tw_sub && pl.d_tw_band && n_samples % 4 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 && n_samples < (1LL << 28) &&
            (TP % 16 != 0 || reinterpret_cast<uintptr_t>(out) % 64 != 0) && reinterpret_cast<uintptr_t>(out) % 4 == 0 &&
            2LL * n_clips * ((T + 31) / 32) < (1LL << 30) && (long long)n_clips * 2048 * TP < (1LL << 40))
            return run_mdct_band_carry(pl, x, out, n_clips, n_samples, T);
    }
    if constexpr (ZAFX_MDCT_BAND && LOG2NF == 10 && LAYOUT == ZAFX_LAYOUT_FT) {
        // W = 4096, reference layout: 32-frame tiles in two bands of bins (16-byte buffer loads: clips of a multiple of four samples)
        if (pl.d_tw_sub && pl.d_tw_band && n_samples % 4 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 && n_samples < (1LL << 28) &&
            (long long)n_clips * ((T + 31) / 32) < (1LL << 31))
            return run_mdct_band(pl, x, out, n_clips, n_samples, T);
    }
    if constexpr (ZAFX_MDCT_QUAD && LOG2NF == 11 && LAYOUT == ZAFX_LAYOUT_FT) {
        // W = 8192, reference layout: 32-frame tiles in four bands of bins (4-byte buffer loads: 32-bit byte offsets inside a clip)
        if (pl.d_tw_sub && pl.d_tw_quad && n_samples < (1LL << 28) && (long long)(T + 33) * 4096 < (1LL << 28) && reinterpret_cast<uintptr_t>(x) % 4 == 0 &&
            reinterpret_cast<uintptr_t>(out) % 4 == 0 && (long long)n_clips * ((T + 31) / 32) < (1LL << 31))
            return run_mdct_quad(pl, x, out, n_clips, n_samples, T);
    }
    if constexpr (mdct_use_persistent(LOG2NF, LAYOUT)) {
        return run_mdct_p<LOG2NF, LAYOUT == ZAFX_LAYOUT_TF>(pl, x, out, n_clips, n_samples, T);
    } else {
        constexpr int LOG2E = default_log2e(LOG2NF);
        constexpr int FPB = mdct_fpb(LOG2NF, LAYOUT);
        using G = MdctCfg<LOG2NF, LOG2E, FPB>;
        auto kern = k_mdct<LOG2NF, LOG2E, FPB, LAYOUT>;
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, G::SMEM_FWD); e != hipSuccess) return e;
        const int tiles = (T + FPB - 1) / FPB;
        const long long blocks = (long long)tiles * n_clips;
        if (blocks <= 0) return hipSuccess;
        pl.ran = "k_mdct";
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(G::NT), G::SMEM_FWD, pl.stream, x, pl.d_window, pl.d_tw_pass, pl.d_tw_aux, out,
                           (long long)n_samples, T, (int)row_pitch(pl, T), tiles);
        return hipGetLastError();
    }
}

// IMDCT tile geometry: FPB frames resident, NSLOT of them transformed at a time.
// The time-minor layout wants FPB = 32 (32 frames x 4 B = one 128-B line per gathered row).
#ifndef ZAFX_IMDCT_WIN_GLOBAL
#define ZAFX_IMDCT_WIN_GLOBAL 1
#endif
// tables beside the frames: pass twiddles, g_m (8 NF), carry (8 NF) and -- unless it is read from global memory -- the window (16 NF)
constexpr int imdct_fpb_with(int log2nf, bool win_lds) {
    const int pitch = (1 << log2nf) + ((1 << log2nf) >> 4) + 1;
    const int lds_cap = (kMaxLdsBytes - twiddle_total(log2nf, default_log2e(log2nf)) * 8 - ((win_lds ? 32 : 16) << log2nf)) / (pitch * 8);
    int f = 32;   // (both layouts: the carry form walks 32-frame tiles; the time-minor layout also NEEDS 32 for whole-line rows)
    while (f > lds_cap) f /= 2;
    return f < 2 ? 2 : f;
}
// the window leaves LDS where that doubles the tile (W = 4096: 16 frames instead of 8; 1024 clips x 10 s: 2.43 -> 1.50 ms at T = 217, 1.72 -> 1.16 ms at T = 224)
constexpr bool imdct_win_lds(int log2nf) { return !(ZAFX_IMDCT_WIN_GLOBAL && imdct_fpb_with(log2nf, false) > imdct_fpb_with(log2nf, true)); }
constexpr int imdct_fpb(int log2nf, int layout) {
    (void)layout;
    return imdct_fpb_with(log2nf, imdct_win_lds(log2nf));
}
constexpr int imdct_nslot(int log2nf, int fpb) {
    const int p = (1 << log2nf) >> default_log2e(log2nf);
    int s = 1024 / p;   // (8 fat waves measured slower here: 2.02 vs 1.53 ms)
    if (s > fpb) s = fpb;
    // NT = s * p must be a multiple of fpb for the time-minor gather
    while ((s * p) % fpb != 0 && s < 1024 / p) ++s;
    return s;
}

// ---------------------------------------------------------------------------------
// inverse, reference layout, W = 8192: two CLASSES of rows per 16-frame tile (k_imdct_q; round 6)
// ---------------------------------------------------------------------------------
// The frame's NF = 2048-point transform y = FFT(c), c[m] = (X[2m] + i X[M-1-2m]) g_m (zaf.py:1138-1163 as the W/4-point algorithm of k_imdct),
// does not fit LDS sixteen frames at a time, and the coefficient rows are the large side: they are read ONCE, as two classes by the parity of
// m -- rows 4m', M-1-4m' (c[2m']) and rows 4m'+2, M-3-4m' (c[2m'+1]) --, each ONE 1024-point transform per frame (a wavefront per frame,
// in its own buffer): y[k], y[k + 1024] = Ye[k] +- w_2048^k Yo[k].  A lane keeps Ye of its sixteen k across the second round (32
// registers), forms z = y g and with it four of the frame's reals u (zaf.py:1166-1169 before the window): u[2k] = Re z_k,
// u[2k + 2048] = Re z_(k+1024), u[4095 - 2k] = -Im z_k, u[2047 - 2k] = -Im z_(k+1024).  The unfold + window + TDAC overlap-add
// (zaf.py:1166-1182: older frame first) then runs on u in LDS, eight frames at a time (16 frames x 16 KB do not fit): a wave stores the
// samples between its frame and the one before (the tile's first frame: the carry of the tile before) as 16-byte pieces.
#ifndef ZAFX_IMDCT_QUAD
#define ZAFX_IMDCT_QUAD 1
#endif
#ifndef ZAFX_IMDCT_QUAD_DEPTH
#define ZAFX_IMDCT_QUAD_DEPTH 1   // sweeps of a class's gather in flight per thread (two 16-byte loads each; measured 1 / 2 / 4: 1.46 / 1.51 / 1.67 ms;
                                  // sweeps requested ahead -- the next tile's first class through the exchange, the second class through the first
                                  // transform, 1 to 4 sweeps -- 1.62 to 1.81 ms: profiles/r06_notes.md section 7)
#endif
#ifndef ZAFX_IMDCT_QUAD_OUT
#define ZAFX_IMDCT_QUAD_OUT 4     // output pieces (two 16-byte stores each) in flight per lane
#endif
struct ImdctQCfg {
    using C = FftCfg<10, 4>;
    static constexpr int N = C::N, FPB = 16, NT = 1024, PITCH = C::PITCH, NF = 2048, M = 4096;
    static constexpr size_t REGION = (size_t)FPB * PITCH * 8;   // one transform buffer per frame; behind the second round u of eight frames (128 KB) + a carry
    static constexpr size_t SMEM = REGION + (size_t)NF * 4 + (size_t)C::TW * 8 + (size_t)N * 8;   // buffers | carry (the lower half of a frame's u) | pass twiddles | g[0 .. 1024)
};
static_assert(ImdctQCfg::SMEM <= (size_t)kMaxLdsBytes, "k_imdct_q: tile + carry + tables exceed LDS");
static_assert(ImdctQCfg::REGION >= (size_t)(ImdctQCfg::FPB / 2) * ImdctQCfg::M * 4 + (size_t)ImdctQCfg::NF * 4, "k_imdct_q: no room for u of eight frames and the second carry in the transform buffers");
typedef float zafx_f4u __attribute__((ext_vector_type(4), aligned(4)));   // a 16-byte piece at any 4-byte alignment (the reference's output length (T-1) M - 1 is odd)

__global__ __launch_bounds__(ImdctQCfg::NT) void k_imdct_q(const float* __restrict__ coefs, const float* __restrict__ win, const float2* __restrict__ twp,
                                                           const float2* __restrict__ g, const float2* __restrict__ w2048, float* __restrict__ y, int T, int TP,
                                                           long long out_len, int tiles, int segs, int seg_tiles, int total_units) {
    using Q = ImdctQCfg;
    using C = Q::C;
    constexpr int P = 64, E = C::E, FPB = Q::FPB, NT = Q::NT, PITCH = Q::PITCH, NF = Q::NF, M = Q::M;
    static_assert(C::P == 64, "a frame's 1024-point transform is one wavefront's");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* frames = reinterpret_cast<float2*>(smem_raw);   // FPB transform buffers ...
    float* us = reinterpret_cast<float*>(smem_raw);          // ... and, behind the second round, u of eight frames at a time
    float* carry = reinterpret_cast<float*>(smem_raw + Q::REGION);   // u[0 .. NF) of the frame before the tile
    float* carry_b = us + (FPB / 2) * M;   // ... of the tile's eighth frame: behind u of eight frames, in the transform buffers' last 8 KB
    float2* tw_l = reinterpret_cast<float2*>(carry + NF);
    float2* gl = tw_l + C::TW;   // g_m, m < 1024 (g_(m + 1024) = g_m e^(-i pi / 4): unit_root(8 m + 1, 8 W))
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int f4 = (tid & 3) * 4, rq = tid >> 2;             // gather: my four frames of the tile, my row within a sweep of 256
    for (int i = tid; i < C::TW; i += NT) tw_l[i] = twp[i];
    for (int i = tid; i < NF; i += NT) carry[i] = 0.f;
    for (int i = tid; i < C::N; i += NT) gl[i] = g[i];
    lds_barrier();
    const float2 wl = w2048[lane];   // w_2048^k of my k = lane + 64 i: this times exp(-2 pi i i / 32) (uniform: scalar registers)
    float2 wi[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) wi[i] = w2048[64 * i];
    constexpr float kR8 = 0.70710678118654752440f;
    float2* const buf = frames + wave * PITCH;
    PROF_INIT(g_prof_imdct);
    const float gain = 2.f / (float)M;

    for (int unit = blockIdx.x; unit < total_units; unit += gridDim.x) {
        const int seg = unit % segs;
        const long long clip = unit / segs;
        const int tile_a = seg * seg_tiles, tile_b = min(tile_a + seg_tiles, tiles);
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(coefs + clip * M * TP, (unsigned)((long long)M * TP * 4));
        float* const yc = y + clip * out_len;
        const int tile_0 = tile_a > 0 ? tile_a - 1 : 0;
        for (int tile = tile_0; tile < tile_b; ++tile) {
            const bool write_out = tile >= tile_a;   // (the tile in front of a segment only leaves its last frame's lower half behind)
            const int t0 = tile * FPB;
            int lo = lane;
            asm volatile("" : "+v"(lo));   // (opaque per tile: addresses are recomputed, not carried across the rounds)
            auto fold_class = [&](int par) {   // c[2m' + par] of my four frames -> their buffers, slot m'
#pragma unroll ZAFX_IMDCT_QUAD_DEPTH
                for (int sw = 0; sw < 4; ++sw) {
                    const int mp = rq + 256 * sw, m = 2 * mp + par;   // rows 2m and M-1-2m, four frames each
                    const float4 re = buf_load_f32x4(rs, (int)(((unsigned)(2 * m) * (unsigned)TP + (unsigned)(t0 + f4)) * 4u));
                    const float4 im = buf_load_f32x4(rs, (int)(((unsigned)(M - 1 - 2 * m) * (unsigned)TP + (unsigned)(t0 + f4)) * 4u));
                    float2 gm = gl[2 * (rq + 256 * (sw & 1)) + par];
                    if (sw >= 2) gm = make_float2(kR8 * (gm.x + gm.y), kR8 * (gm.y - gm.x));
                    float2* fb = frames + f4 * PITCH + phys(mp);
                    fb[0] = cmul(make_float2(re.x, im.x), gm);
                    fb[PITCH] = cmul(make_float2(re.y, im.y), gm);
                    fb[2 * PITCH] = cmul(make_float2(re.z, im.z), gm);
                    fb[3 * PITCH] = cmul(make_float2(re.w, im.w), gm);
                }
            };
            auto transform = [&]() {
                lds_barrier();
                float2 v[E];
                regs_read<10, 4>(v, buf, lo);
                frame_sync<P>();
                fft_frame<10, 4>(v, buf, lo, tw_l);
                frame_sync<P>();
            };
            PROF_MARK(0);
            fold_class(0);
            PROF_MARK(1);
            transform();
            PROF_MARK(2);
            float2 ye[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) ye[i] = buf[phys(lo + 64 * i)];
            lds_barrier();   // every wave has read its first class
            PROF_MARK(3);
            fold_class(1);
            PROF_MARK(4);
            transform();
            PROF_MARK(5);
            float ua[16], ub[16], uc[16], ud[16];   // u[2k], u[2k + 2048], u[4095 - 2k], u[2047 - 2k] of my k = lane + 64 i
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int k = lo + 64 * i;
                const float2 t = cmul(cmul(wl, wi[i]), buf[phys(k)]);
                const float2 y0 = make_float2(ye[i].x + t.x, ye[i].y + t.y), y1 = make_float2(ye[i].x - t.x, ye[i].y - t.y);
                const float2 gk = gl[k];
                const float2 z0 = cmul(y0, gk), z1 = cmul(y1, gk);   // (z1 still lacks e^(-i pi / 4))
                ua[i] = z0.x, ub[i] = kR8 * (z1.x + z1.y), uc[i] = -z0.y, ud[i] = kR8 * (z1.x - z1.y);
            }
            PROF_MARK(6);
            lds_barrier();   // every wave has read its second class: the buffers become u of eight frames
            PROF_MARK(7);
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {   // (16 frames x 16 KB of u do not fit: frames 0-7, then 8-15; the other eight waves wait)
                const bool mine = (wave >> 3) == half;
                float* um = us + (wave & 7) * M;
                if (mine) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int k = lo + 64 * i;
                        um[2 * k] = ua[i];
                        um[2 * k + 2048] = ub[i];
                        um[4095 - 2 * k] = uc[i];
                        um[2047 - 2 * k] = ud[i];
                    }
                    if ((wave & 7) == 7) {   // the group's last frame: its lower half is the next group's (the next tile's) older frame
                        float* cw = half == 0 ? carry_b : carry;
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int k = lo + 64 * i;
                            cw[2 * k] = ua[i];
                            cw[2047 - 2 * k] = ud[i];
                        }
                    }
                }
                PROF_MARK(8);
                lds_barrier();
                PROF_MARK(9);
                if (mine) {
                    // the M samples between my frame t and frame t - 1 (zaf.py:1172-1182): out[n1] = gain (old(n1) w[n1 + M] + cur(n1) w[n1]),
                    //   cur = u_t[NF + n1] (n1 < NF), -u_t[3 NF - 1 - n1];   old = -u_(t-1)[NF - 1 - n1] (n1 < NF), -u_(t-1)[n1 - NF]
                    const int t = t0 + wave;
                    const float* up = (wave & 7) != 0 ? um - M : half == 0 ? carry : carry_b;   // u[0 .. NF) of the frame before
                    const bool store = write_out && t >= 1 && t < T;
                    const long long o0 = (long long)(t - 1) * M;
#pragma unroll ZAFX_IMDCT_QUAD_OUT
                    for (int q = 0; q < 8; ++q) {
                        const int n1 = 4 * lo + 256 * q;   // and n1 + NF
                        const float4 c0 = *reinterpret_cast<const float4*>(um + NF + n1);             // u[NF + n1 ..]
                        const float4 c1 = *reinterpret_cast<const float4*>(um + 2 * NF - 4 - n1);      // u[3 NF - 1 - (n1 + NF) - 3 ..] reversed
                        const float4 p0 = *reinterpret_cast<const float4*>(up + NF - 4 - n1);          // u'[NF - 1 - n1 - 3 ..] reversed
                        const float4 p1 = *reinterpret_cast<const float4*>(up + n1);                   // u'[(n1 + NF) - NF ..]
                        const float4 wc0 = *reinterpret_cast<const float4*>(win + n1), wo0 = *reinterpret_cast<const float4*>(win + M + n1);
                        const float4 wc1 = *reinterpret_cast<const float4*>(win + NF + n1), wo1 = *reinterpret_cast<const float4*>(win + M + NF + n1);
                        zafx_f4u r0, r1;
                        r0.x = (-p0.w * wo0.x + c0.x * wc0.x) * gain, r0.y = (-p0.z * wo0.y + c0.y * wc0.y) * gain;
                        r0.z = (-p0.y * wo0.z + c0.z * wc0.z) * gain, r0.w = (-p0.x * wo0.w + c0.w * wc0.w) * gain;
                        r1.x = (-p1.x * wo1.x + -c1.w * wc1.x) * gain, r1.y = (-p1.y * wo1.y + -c1.z * wc1.y) * gain;
                        r1.z = (-p1.z * wo1.z + -c1.y * wc1.z) * gain, r1.w = (-p1.w * wo1.w + -c1.x * wc1.w) * gain;
                        if (store) {
                            *reinterpret_cast<zafx_f4u*>(yc + o0 + n1) = r0;
                            float* d = yc + o0 + NF + n1;
                            if (o0 + NF + n1 + 3 < out_len) {
                                *reinterpret_cast<zafx_f4u*>(d) = r1;
                            } else {   // the clip's last piece: the trim [H : -H - 1] drops one more sample (zaf.py:1182)
                                if (o0 + NF + n1 < out_len) d[0] = r1.x;
                                if (o0 + NF + n1 + 1 < out_len) d[1] = r1.y;
                                if (o0 + NF + n1 + 2 < out_len) d[2] = r1.z;
                            }
                        }
                    }
                }
                PROF_MARK(10);
                lds_barrier();   // the frames' u are read
                PROF_MARK(11);
            }
        }
        for (int i = tid; i < NF; i += NT) carry[i] = 0.f;   // (a clip starts with no frame before it; read behind the next tile's barriers)
    }
}

static hipError_t run_imdct_quad(const zafx_plan& pl, const float* coefs, float* y, int64_t n_clips, int T, int64_t out_len) {
    using Q = ImdctQCfg;
    const int tiles = (T + Q::FPB - 1) / Q::FPB;
    if ((long long)tiles * n_clips <= 0 || out_len <= 0) return hipSuccess;
    auto kern = k_imdct_q;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, Q::SMEM); e != hipSuccess) return e;
    const int segs = carry_segments(n_clips, tiles, pl.n_cus);
    const int seg_tiles = (tiles + segs - 1) / segs;
    const long long units = n_clips * segs;
    pl.ran = "k_imdct_q";
    hipLaunchKernelGGL(kern, dim3((unsigned)std::min<long long>(units, pl.n_cus)), dim3(Q::NT), Q::SMEM, pl.stream, coefs, pl.d_window, pl.d_tw_sub, pl.d_tw_aux,
                       pl.d_tw_quad, y, T, (int)row_pitch(pl, T), (long long)out_len, tiles, segs, seg_tiles, (int)units);
    return hipGetLastError();
}

template <int LOG2NF, int LAYOUT>
static hipError_t run_imdct(const zafx_plan& pl, const float* coefs, float* y, int64_t n_clips, int T, int64_t out_len) {
    if constexpr (ZAFX_IMDCT_QUAD && LOG2NF == 11 && LAYOUT == ZAFX_LAYOUT_FT) {
        // W = 8192: the two-class kernel (four frames per 16-byte piece: a row pitch that is a multiple of 4; 32-bit byte offsets inside a clip)
        const int64_t TP = row_pitch(pl, T);
        if (pl.d_tw_sub && pl.d_tw_quad && TP % 4 == 0 && reinterpret_cast<uintptr_t>(coefs) % 16 == 0 && reinterpret_cast<uintptr_t>(pl.d_window) % 16 == 0 &&
            (long long)4096 * TP * 4 < (1LL << 31) && n_clips * (((long long)T + 15) / 16 + 1) < (1LL << 30))
            return run_imdct_quad(pl, coefs, y, n_clips, T, out_len);
    }
    constexpr int LOG2E = default_log2e(LOG2NF);
    constexpr int FPB = imdct_fpb(LOG2NF, LAYOUT);
    constexpr int NSLOT = imdct_nslot(LOG2NF, FPB);
    using C = FftCfg<LOG2NF, LOG2E>;
    constexpr bool WINL = imdct_win_lds(LOG2NF);
    constexpr size_t SMEM = (size_t)(FPB * C::PITCH + C::TW + C::N) * 8 + (WINL ? (size_t)C::N * 16 : 0) + (size_t)C::N * 8;   // frames + twiddles + g_m + window + carry
    static_assert(SMEM <= (size_t)kMaxLdsBytes, "IMDCT tile does not fit LDS");
    auto kern = k_imdct<LOG2NF, LOG2E, FPB, NSLOT, LAYOUT, WINL>;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, SMEM); e != hipSuccess) return e;
    const int tiles = (T + FPB - 1) / FPB;
    if ((long long)tiles * n_clips <= 0 || out_len <= 0) return hipSuccess;
    const int per_cu = (int)std::min<size_t>(2, (size_t)kMaxLdsBytes / SMEM);
    const long long max_grid = (long long)pl.n_cus * std::max(per_cu, 1);
    const int segs = carry_segments(n_clips, tiles, max_grid);
    const int seg_tiles = (tiles + segs - 1) / segs;
    const long long units = (long long)n_clips * segs;
    const long long grid = std::min<long long>(units, max_grid);
    pl.ran = "k_imdct";
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NSLOT * C::P), SMEM, pl.stream, coefs, pl.d_window, pl.d_tw_pass, pl.d_tw_aux, y, T,
                       (int)row_pitch(pl, T), (long long)out_len, tiles, segs, seg_tiles, (int)units);
    return hipGetLastError();
}

#define ZAFX_MDCT_SIZES(X) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11)

bool mdct_supported(int log2nf) { return log2nf >= 4 && log2nf <= 11; }
int mdct_frames_per_block(int log2nf, int layout) { return mdct_fpb(log2nf, layout); }
const char* mdct_kernel_name(int log2nf, int layout) {
    if (ZAFX_MDCT_BAND && log2nf == 10 && layout == ZAFX_LAYOUT_FT) return "k_mdct_ft32b";
    if (ZAFX_MDCT_QUAD && log2nf == 11 && layout == ZAFX_LAYOUT_FT) return "k_mdct_ft32q";
    return mdct_use_persistent(log2nf, layout) ? "k_mdct_ft32" : "k_mdct";
}
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

ZAFX_PROF_EXPORT(zafx_debug_prof_mdct, g_prof_mdct)
ZAFX_PROF_EXPORT(zafx_debug_prof_imdct, g_prof_imdct)
