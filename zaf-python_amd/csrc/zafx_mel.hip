// zafx_mel.hip -- fused melspectrogram / MFCC kernel for gfx950 (MI355X).
//
// One persistent workgroup per CU; a tile = 16 consecutive frames of one clip.  The STFT
// never reaches HBM:
//   framing + window + real FFT (as k_stft)             zaf.py:369 / :436  (stft)
//   |X[k]| or |X[k]|^2 for k = 1..W/2, in place in LDS   zaf.py:370 / :437-439
//   mel = FB . S            -- v_mfma_f32_16x16x4_f32    zaf.py:373 / :445  (np.matmul)
//   log(mel + eps), DCT-II rows 1..ncoef as a second MFMA GEMM   zaf.py:443-452
//
// The filterbank is banded (zaf.py:305-316: each row is one triangle), so only the K-steps
// that hold non-zeros of a 16-row block are multiplied.  The band of a block grows with
// frequency (5 ... 97 K-steps for 128 filters at W = 2048), so the host cuts the blocks into
// work items of bounded length and deals them to the wavefronts (longest first); an item
// leaves its 16 x 16 partial tile in an LDS slot and a fixed-order reduction adds the parts
// of a block -- balanced AND deterministic (no atomics).  A operand: packed FB fragment from
// global/L2 (one coalesced 256-B load per MFMA); B operand: S[t][c] from LDS, frame pitch
// = 2 (mod 32) dwords so the 16 frames x 4 columns of a fragment hit 64 distinct banks.
// Slots (256 floats each) live in the upper halves of the frame buffers, which are dead
// once the magnitudes have been written.
#include <algorithm>
#include <cstdlib>

#include "zafx_fft.hpp"
#include "zafx_internal.hpp"

namespace zafx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
ZAFX_PROF_ARRAY(g_prof_mel)

#ifndef ZAFX_MEL_THREADS
#define ZAFX_MEL_THREADS 1024
#endif
#ifndef ZAFX_MEL_FPB
#define ZAFX_MEL_FPB 16
#endif
constexpr int kMelThreads = ZAFX_MEL_THREADS;
// frames per tile.  8 (with ZAFX_MEL_THREADS = 512): two workgroups per CU, each with 8 wavefronts and 8-frame tiles -- their
// phases (transforms | filterbank GEMM | reduction) drift apart, so one workgroup's butterflies fill the vector pipe while the
// other sits in its matrix / reduction / barrier third; half of every MFMA's 16 columns is then unused (the matrix pipe is
// 90 % idle), and the window is read from global memory (two workgroups' tables do not fit LDS beside 2 x 8 frame buffers)
constexpr int kMelFpb = ZAFX_MEL_FPB;
constexpr bool kMelWinLds = kMelFpb == 16;   // 512: 8 fat waves (2 frames each, register prefetch); 1024: 16 waves, one frame each
#ifndef ZAFX_MEL_R32
#define ZAFX_MEL_R32 0
#endif
// window + first radix-16 butterflies of the NEXT tile's frame beside the matrix instructions of the filterbank product: the vector
// pipe is idle there -- four waves' dependent MFMA chains keep a SIMD for 2.3 k cycles.  (Round 3 ran these 500 cycles per wave after
// the reduction, ahead of the tile's last barrier.  A first form of round 4 issued them piecewise between the K-steps of one wave: the
// register allocator then spilled the resident A fragments and reloaded them inside the dependent chain, 2.0 ms.)
#ifndef ZAFX_MEL_GEMM_PRE
#define ZAFX_MEL_GEMM_PRE 1
#endif
// (experiment switch) A fragments of the filterbank re-read from L2 every tile, ahead of the barrier, instead of riding in registers
#ifndef ZAFX_MEL_RELOAD_A
#define ZAFX_MEL_RELOAD_A 1
#endif

// points per thread of the FFT: 32 (two radix-32 passes, a frame per half wavefront, 8 waves) for W = 2048 when enabled
constexpr int mel_log2e(int log2n) { return (ZAFX_MEL_R32 && log2n == 10) ? 5 : default_log2e(log2n); }
constexpr int mel_threads(int log2n, int log2e) {
    const int p = fft_threads(log2n, log2e);
    return ((kMelThreads / p) < 16 ? (kMelThreads / p) : 16) * p;
}

template <int LOG2N, int LOG2E>
struct MelCfg {
    using C = FftCfg<LOG2N, LOG2E>;
    static constexpr int N = C::N;
    static constexpr int FPB = kMelFpb;   // frames per tile (<= 16 = the MFMA N dimension)
    static constexpr int NSLOT = (kMelThreads / C::P) < FPB ? (kMelThreads / C::P) : FPB;   // frames transformed concurrently
    static constexpr int NT = NSLOT * C::P;
    // 256-float slots: in the dead upper half of every frame buffer when it is large enough,
    // and (for the smaller windows) in a separate region after the tables
    static constexpr int UPPER = 2 * C::PITCH - N;           // free floats per frame buffer after S
    static constexpr int SLOTS_PER_BUF = UPPER / 256;
    static constexpr int IN_FRAMES = FPB * SLOTS_PER_BUF;         // slots in the dead upper halves
    static constexpr int EXTRA_SLOTS = N >= 1024 ? 0 : 64;         // + a region of their own where LDS is plentiful (W <= 1024)
    static constexpr int CAPACITY = IN_FRAMES + EXTRA_SLOTS;
    static constexpr size_t TABLES = (size_t)(C::TW + (kMelWinLds ? N + N / 2 + 1 : 0)) * 8;     // pass twiddles, window, split roots (8-frame form: the latter two from global memory)
    static constexpr size_t SMEM = (size_t)FPB * C::PITCH * 8 + TABLES + (size_t)EXTRA_SLOTS * 256 * 4;
};

// One banded GEMM stage: every wave runs its items (16 rows x 16 frames x `steps` K-steps) and
// leaves the partial tile in slot `slot0 + item`.
template <class SlotFn, class BFn>
__device__ __forceinline__ void gemm_items(const float* __restrict__ pack, const int4* __restrict__ items,
                                           const int* __restrict__ wave_ptr, int wave, int lane, int slot0, SlotFn slot_ptr, BFn b_at) {
    for (int it = wave_ptr[wave]; it < wave_ptr[wave + 1]; ++it) {
        const int4 w = items[it];   // x = item id (slot), y = first column, z = steps, w = offset into pack (in steps)
        const float* ap = pack + (size_t)w.w * 64 + lane;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        int s = 0;
        for (; s + 1 < w.z; s += 2) {   // two independent accumulators cover the 40-cycle dependent latency
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[(size_t)s * 64], b_at(w.y + 4 * s), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[(size_t)(s + 1) * 64], b_at(w.y + 4 * s + 4), acc1, 0, 0, 0);
        }
        if (s < w.z) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[(size_t)s * 64], b_at(w.y + 4 * s), acc0, 0, 0, 0);
        float* dst = slot_ptr(slot0 + w.x);
        const int bt = lane & 15, bk = lane >> 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(4 * bk + r) * 16 + bt] = acc0[r] + acc1[r];
    }
}

// One banded GEMM stage with the wave's A fragments resident in registers (RA K-steps in execution order): no vector-memory
// traffic of its own, so the samples of the next tile can be requested underneath, one load after every K-step (`hook`) -- a
// burst of 16 loads per lane overruns the CU's vector-memory queue and blocks the wave at issue (profiles/r02_notes.md).
// Step i of the wave: 16 bits of desc[i / 2] (SGPRs, tile-invariant; PackedBand::d_desc): first column / 4 | slot << 8 | item ends << 15.
template <int RA, int CHUNK = 0, class SlotFn, class BFn, class Hook>
__device__ __forceinline__ void gemm_resident(const float (&a)[RA], const int (&desc)[(RA + 1) / 2], int n, int lane, int slot0, SlotFn slot_ptr, BFn b_at, Hook hook) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int bt = lane & 15, bk = lane >> 4;
    // opaque copies: decoded ahead of the persistent loop, the steps' columns, flags and slot addresses would take some
    // sixty scalar registers (spilled to VGPR lanes and read back step by step)
    int dw[(RA + 1) / 2];
#pragma unroll
    for (int j = 0; j < (RA + 1) / 2; ++j) {
        dw[j] = desc[j];
        asm volatile("" : "+s"(dw[j]));
    }
    asm volatile("" : "+s"(n));
    // the B fragments of a chunk of steps are requested up front: the MFMA chain then waits for LDS once per chunk, not once per
    // step (a chunk = all steps up to 18; the 8-frame form's 36 steps go in two chunks, for the registers' sake)
    constexpr int CH = CHUNK > 0 ? CHUNK : RA > 18 ? (RA + 1) / 2 : RA;   // (CHUNK: with vector work between the steps the chain can afford to wait for LDS more often)
#pragma unroll
    for (int c0 = 0; c0 < RA; c0 += CH) {
        float b[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if (c0 + i < RA) b[i] = b_at(4 * ((dw[(c0 + i) >> 1] >> (16 * ((c0 + i) & 1))) & 255));   // (no step: descriptor 0, column 0)
#pragma unroll
        for (int ii = 0; ii < CH; ++ii) {
            const int i = c0 + ii;
            if (i < RA) {
                if (i < n) {   // (uniform)
                    const int d = dw[i >> 1] >> (16 * (i & 1));
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[ii], acc, 0, 0, 0);
                    if (d & 0x8000) {   // the item ends: leave its partial tile in its slot
                        float* dst = slot_ptr(slot0 + ((d >> 8) & 127)) + (4 * bk) * 16 + bt;
#pragma unroll
                        for (int q = 0; q < 4; ++q) dst[q * 16] = acc[q];
                        acc = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
                hook(i);
            }
        }
    }
}

// RES: filterbank (and DCT) fragments resident in registers + prefetch of the next tile under the filterbank phases.
// MF: 1 = mel, 2 = mfcc, 3 = mfcc with the register-fed DCT compiled in (resident form), 0 = the kernel argument decides.
template <int LOG2N, int LOG2E, bool ALIGNED, bool RES, int MF>
__global__ __launch_bounds__(mel_threads(LOG2N, LOG2E), kMelFpb == 8 ? 2 * mel_threads(LOG2N, LOG2E) / 256 : 1) void k_mel(
    const float* __restrict__ x, const float* __restrict__ win, const float2* __restrict__ twp,
    const float2* __restrict__ tws, const float* __restrict__ fb_pack, const int4* __restrict__ fb_items,
    const int* __restrict__ fb_wave_ptr, const int* __restrict__ fb_blk_ptr, const unsigned short* __restrict__ fb_desc, int fb_blocks, int fb_nitems, int fb_steps, int dct_steps,
    const float* __restrict__ dct_pack, const int4* __restrict__ dct_items, const int* __restrict__ dct_wave_ptr,
    const int* __restrict__ dct_blk_ptr, const unsigned short* __restrict__ dct_desc, const float* __restrict__ dct_direct, int dct_j, int dct_blocks, float* __restrict__ out, long long n_samples, int hop, int T, int TP,
    int tiles, int total_tiles, int n_filters, int n_coefs, int mfcc_arg, int layout) {
    const bool mfcc = MF == 0 ? mfcc_arg != 0 : MF >= 2;
    constexpr bool DIRECT = MF == 3;   // DCT fed from the registers of the filterbank's reduction (below)
    using C = FftCfg<LOG2N, LOG2E>;
    using G = MelCfg<LOG2N, LOG2E>;
    constexpr int N = C::N, P = C::P, E = C::E, W = 2 * N, NT = G::NT, FPB = G::FPB, NSLOT = G::NSLOT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* frames = reinterpret_cast<float2*>(smem_raw);
    float2* tw_l = frames + FPB * C::PITCH;
    float2* win_s = tw_l + C::TW;
    float2* tws_s = win_s + (kMelWinLds ? N : 0);
    const float2* win_l = kMelWinLds ? win_s : reinterpret_cast<const float2*>(win);   // (8-frame form: from global memory / L1)
    const float2* tws_l = kMelWinLds ? tws_s : tws;
    float* extra = reinterpret_cast<float*>(tws_s + (kMelWinLds ? N / 2 + 1 : 0));
    float* fall = reinterpret_cast<float*>(frames);
    const int tid = threadIdx.x;
    // The raw samples of a round are requested one phase ahead.  Fat waves (NT <= 512, two rounds per tile): before the FFT of
    // the previous round.  16 thin waves (one round per tile): when the tile's spectra are done, so that the 128 KB of the next
    // tile fly under the filterbank phases -- all 16 waves otherwise request, wait and transform in lockstep, and the three
    // resources (vector memory 5.8 k cycles per tile, LDS 6.5 k, VALU 7 k) are used one after the other.
    constexpr bool TEAM8 = FPB == 8 && RES;   // two 8-wave workgroups per CU, 8-frame tiles: the resident form's schedule, A fragments re-read every tile
    constexpr bool PREFETCH = NT <= 512 && !TEAM8;
    constexpr bool LATE = !PREFETCH && RES;
    // 16-byte lane loads for the prefetched frame (W = 2048, resident form): the lanes l and l + 16 (l in an even 16-lane
    // row) share their loads -- the first fetches the points (n, n + 1) + 64 i for i = 0 .. 7, the second for i = 8 .. 15 --
    // and one v_permlane16_swap per register hands each lane its own points (row_pair_index / row_pair_unpack).  Eight loads
    // per lane instead of sixteen: the request of the next tile blocks half as long at the CU's vector-memory queue.
    constexpr bool PAIR16 = LATE && LOG2N == 10 && LOG2E == 4;   // (without resident fragments the filterbank's own loads would queue behind the prefetch)
    // window at the points the lane holds: from the LDS table, or (8-frame form) from global memory
    auto win_at = [&](int po, int i) -> float2 {
        if constexpr (kMelWinLds) return win_l[po + i * P];
        else return win_l[(PAIR16 ? row_pair_index(po) : po) + i * P];
    };
    for (int i = tid; i < C::TW; i += NT) tw_l[i] = twp[i];
    // (pair form, below: the window in lane order, win_l[64 i + lane] = the window at the lane's points p + 64 i)
    if constexpr (kMelWinLds)
        for (int i = tid; i < N; i += NT) win_s[i] = reinterpret_cast<const float2*>(win)[PAIR16 ? (i & ~63) + row_pair_index(i & 63) : i];
    if constexpr (kMelWinLds)
        for (int i = tid; i <= N / 2; i += NT) tws_s[i] = tws[i];
    __syncthreads();

    auto slot_ptr = [&](int s) -> float* {
        if constexpr (G::IN_FRAMES > 0) {
            if (G::EXTRA_SLOTS == 0 || s < G::IN_FRAMES)
                return fall + (size_t)(s / G::SLOTS_PER_BUF) * (2 * C::PITCH) + N + (s % G::SLOTS_PER_BUF) * 256;
        }
        return extra + (size_t)(s - G::IN_FRAMES) * 256;
    };

    const int slot = tid / P, p = tid % P;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: item ranges and descriptors load into SGPRs
    const float eps = 2.220446049250313e-16f;   // np.finfo(float).eps (zaf.py:445)
    const int lt0 = fb_nitems;                  // first log-mel slot
    const int dslot0 = fb_nitems + fb_blocks;   // first DCT partial slot

    // raw samples of one frame of this wave's slot: xr[i] = (x[2n], x[2n+1]), n = p + i P (zero padding of zaf.py:112-125).
    // fetch_begin() sets up the frame (an interior frame is then requested load by load with fetch_one(), or all at once
    // with fetch()); a frame that touches the clip's edges is loaded in full by fetch_begin() itself.
    float2 xr[E];
    __amdgpu_buffer_rsrc_t frx = make_rsrc(x, 0);
    int fvoff = 0;
    bool interior = false;   // the frame fetch_begin set up is still to be requested load by load (fetch_one)
    const bool xcd = ZAFX_XCD_ORDER && gridDim.x % 8 == 0;   // (xcd_order: tiles of a clip to the workgroups of one XCD)
    auto fetch_begin = [&](int tlv, int f0, int p) -> bool {   // p: the lane's points are p + i P (an opaque copy inside the tile loop)
        interior = false;
        if (tlv >= total_tiles) return false;
        const int tl = xcd ? xcd_order(tlv, total_tiles) : tlv;
        const int clip = tl / tiles, tile = tl % tiles;
        const int t = tile * FPB + f0 + (P == 64 ? wave : slot);   // (a frame per wave: uniform, the descriptor below stays in SGPRs)
        const float* xc = x + (long long)clip * n_samples;
        const long long s0 = (long long)t * hop - N;
        if (ALIGNED && t < T && s0 >= 0 && s0 + W <= n_samples) {   // interior frame (uniform per frame)
            // buffer loads: one descriptor per clip in SGPRs + one 32-bit offset per lane (16 flat loads in flight would hold
            // 16 64-bit addresses: the prefetch across the filterbank phases then spills)
            frx = make_rsrc(xc, (unsigned)std::min<long long>(n_samples * 4, 0xfffffffcLL));
            fvoff = PAIR16 ? ((int)s0 + 2 * (p & ~1)) * 4 + (p & 1) * (E / 2 * P * 8) : ((int)s0 + 2 * p) * 4;
            interior = true;
            return true;
        }
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const long long s = s0 + 2 * (p + i * P);
            xr[i].x = (t < T && s >= 0 && s < n_samples) ? xc[s] : 0.f;
            xr[i].y = (t < T && s + 1 >= 0 && s + 1 < n_samples) ? xc[s + 1] : 0.f;
        }
        if constexpr (PAIR16 && ZAFX_MEL_GEMM_PRE) {
            // into the layout the 16-byte loads leave (row_pair_unpack is its own inverse): every frame then takes the same way
            // through the staged pre-transform under the filterbank's matrix instructions
            float2 lo[E / 2], hi[E / 2];
#pragma unroll
            for (int i = 0; i < E / 2; ++i) {
                lo[i] = xr[i];
                hi[i] = xr[i + E / 2];
                row_pair_unpack(lo[i], hi[i]);
            }
#pragma unroll
            for (int i = 0; i < E / 2; ++i) {
                xr[2 * i] = lo[i];
                xr[2 * i + 1] = hi[i];
            }
            return true;
        }
        return false;
    };
    auto fetch_one = [&](int i) {
        if constexpr (PAIR16) {
            if (i < E / 2) {   // raw: (xr[2i], xr[2i+1]) = the lane's two points of load i; unpack_pairs() sorts them out
                const float4 q = buf_load_f32x4(frx, fvoff, i * P * 8);
                xr[2 * i] = make_float2(q.x, q.y);
                xr[2 * i + 1] = make_float2(q.z, q.w);
            }
        } else {
            xr[i] = buf_load_f32x2(frx, fvoff, i * P * 8);
        }
    };
    auto unpack_pairs = [&]() {   // raw (a_i, b_i) = (xr[2 i], xr[2 i + 1]) -> xr[i], xr[i + E/2]
        float2 lo[E / 2], hi[E / 2];
#pragma unroll
        for (int i = 0; i < E / 2; ++i) {
            lo[i] = xr[2 * i];
            hi[i] = xr[2 * i + 1];
            row_pair_unpack(lo[i], hi[i]);
        }
#pragma unroll
        for (int i = 0; i < E / 2; ++i) {
            xr[i] = lo[i];
            xr[i + E / 2] = hi[i];
        }
    };
    auto fetch = [&](int tl, int f0, int p) {
        if (fetch_begin(tl, f0, p)) {
#pragma unroll
            for (int i = 0; i < E; ++i) fetch_one(i);
        }
    };
#ifndef ZAFX_MEL_PREPASS
#define ZAFX_MEL_PREPASS 1
#endif
    bool raw = false;   // PAIR16: xr holds raw 16-byte loads (an interior frame) that unpack_pairs() must sort out
    bool pre_done = false;   // PAIR16: xr already holds the windowed frame after the butterflies of pass 1
    auto pre_transform = [&](int po) {   // po: lane index (an opaque copy)
        if constexpr (PAIR16) {
            if (raw) unpack_pairs();
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const float2 wv = win_at(po, i);   // (pair form: the table is in lane order)
                xr[i] = mul_elem(xr[i], wv);
            }
            Dft<16>::run(xr);
        }
    };
    if constexpr (PREFETCH || LATE) {
        raw = fetch_begin(blockIdx.x, 0, PAIR16 ? row_pair_index(p) : p);
        if (interior) {
#pragma unroll
            for (int i = 0; i < E; ++i) fetch_one(i);
        }
    }
    // resident A fragments of this wave: lane l holds A[l & 15][4 step + (l >> 4)] of its K-steps
    constexpr int RAFB = TEAM8 ? 2 * kMelResidentFb : kMelResidentFb, RADCT = TEAM8 ? 2 * kMelResidentDct : kMelResidentDct;
    float afb[RES ? RAFB : 1], adct[RES ? RADCT : 1];
    int sfb[RES ? (RAFB + 1) / 2 : 1], sdct[RES ? (RADCT + 1) / 2 : 1];   // the steps' descriptors, two per scalar register
    int nfb = 0, ndct = 0;
    constexpr int NW = NT / 64;
    // the wave's share of the K-steps is a contiguous range of the packed fragments (pack_band).  16 waves: loaded once, resident for
    // the whole launch; 8-frame form: re-read from L2 every tile (twice the steps per wave: they do not fit beside the transform)
    const int g0 = RES ? (int)((long long)fb_steps * wave / NW) : 0, h0 = RES ? (int)((long long)dct_steps * wave / NW) : 0;
    auto load_fragments = [&](int lane_o) {   // (vector loads only: the descriptors and counts below are read once)
        // one base address per table (opaque: the 44 step addresses of the 8-frame form are immediates off it, not scalars kept across tiles);
        // the tables carry spare steps behind the last one, so a wave with fewer steps than registers reads in bounds
        const float* fp = fb_pack + (size_t)g0 * 64 + lane_o;
        const float* dp = DIRECT ? dct_direct + (size_t)wave * 4 * 64 + lane_o : dct_pack + (size_t)h0 * 64 + lane_o;
        asm volatile("" : "+v"(fp), "+v"(dp));
#pragma unroll
        for (int i = 0; i < RAFB; ++i) afb[i] = fp[i * 64];
        if (mfcc) {   // (uniform; a mel plan has no DCT table)
#pragma unroll
            for (int i = 0; i < RADCT; ++i) adct[i] = dp[i * 64];
        }
    };
    if constexpr (RES) {
        nfb = __builtin_amdgcn_readfirstlane((int)((long long)fb_steps * (wave + 1) / NW) - g0);
#pragma unroll
        for (int i = 0; i < (RAFB + 1) / 2; ++i) sfb[i] = __builtin_amdgcn_readfirstlane(2 * i < nfb ? (fb_desc[g0 + 2 * i] | fb_desc[g0 + 2 * i + 1] << 16) : 0);   // (two spare entries behind the table)
        ndct = __builtin_amdgcn_readfirstlane(mfcc ? (int)((long long)dct_steps * (wave + 1) / NW) - h0 : 0);
#pragma unroll
        for (int i = 0; i < (RADCT + 1) / 2; ++i) sdct[i] = __builtin_amdgcn_readfirstlane(2 * i < ndct ? (dct_desc[h0 + 2 * i] | dct_desc[h0 + 2 * i + 1] << 16) : 0);
        if constexpr (!TEAM8 && !(PAIR16 && ZAFX_MEL_GEMM_PRE && ZAFX_MEL_RELOAD_A)) load_fragments(tid & 63);
    }
    PROF_INIT(g_prof_mel);
    for (int tlv = blockIdx.x; tlv < total_tiles; tlv += gridDim.x) {
        const int tl = xcd ? xcd_order(tlv, total_tiles) : tlv;
        const int clip = tl / tiles, tile = tl % tiles;
        const int t0 = tile * FPB;
        PROF_MARK(0);

        // ---- STFT of the tile's frames, NSLOT at a time; spectrum -> magnitude/power in place.  The raw
        // samples of the NEXT round (or of the next tile's first round) are requested before this
        // round's FFT, so their latency hides under the butterflies.
#pragma unroll 1
        for (int f0 = 0; f0 < FPB; f0 += NSLOT) {
            const int fr = f0 + slot;
            float2* buf = frames + fr * C::PITCH;
            // opaque copy of the lane's index: the window, twiddle and split-root reads stay inside the loop
            // (hoisted out of the persistent loop they cost > 100 VGPRs and spill)
            int po = p;
            asm volatile("" : "+v"(po));
            if constexpr (!PREFETCH && !LATE) fetch(tlv, f0, po);   // 16 thin waves, streamed filterbank: request, wait, transform
            float2 v[E];
            if constexpr (PAIR16) {
                if (!pre_done) pre_transform(po);   // (the first tile: later ones did this ahead of the previous tile's last barrier)
#pragma unroll
                for (int i = 0; i < E; ++i) v[i] = xr[i];
            } else {
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const float2 wv = win_at(po, i);
                v[i] = make_float2(xr[i].x * wv.x, xr[i].y * wv.y);
            }
            }
            if constexpr (PREFETCH) {
                if (f0 + NSLOT < FPB) fetch(tlv, f0 + NSLOT, po);
                else fetch(tlv + gridDim.x, 0, po);
            }
            if constexpr (PAIR16) fft1024_wave<false, true, true>(v, buf, po, (const float2*)tw_l, row_pair_index(po));
            else fft_frame<LOG2N, LOG2E>(v, buf, po, tw_l);
            // real split of the (k, N-k) pairs this thread owns (k = po + i P), kept in registers.  Only lane 0 holds a pair
            // without a partner (i = 0: bins N/2 and N); P is a whole number of padding periods, so the slots of
            // k + i P and N - k - i P are constant offsets from those of po and N - po.
            float mk[E / 2], mn[E / 2];
            constexpr bool LINEAR = P % (1 << C::PS) == 0;
            constexpr int STEP = P + (P >> C::PS);
            const float2* zk = buf + phys_t<C::PS>(po);
            const float2* zn = buf + phys_t<C::PS>(N - po);
#pragma unroll
            for (int i = 0; i < E / 2; ++i) {
                const int k = po + i * P;
                const float2 za = LINEAR ? zk[i * STEP] : buf[phys_t<C::PS>(k)], zb = LINEAR ? zn[-i * STEP] : buf[phys_t<C::PS>(N - k)];
                float2 pw = split_pair_pow4(za, zb, tws_l[k]);   // (4 |X[k]|^2, 4 |X[N-k]|^2): the packed filterbank carries the 1/4 (1/2 for |X|)
                if (i == 0 && k == 0) {
                    const float2 zc = buf[phys_t<C::PS>(N / 2)];   // |X[N/2]| = |Z[N/2]|
                    const float ny = za.x - za.y;                  // X[N] (Nyquist, kept: zaf.py:370)
                    pw = make_float2(4.f * (zc.x * zc.x + zc.y * zc.y), 4.f * (ny * ny + 0.f * 0.f));
                }
                mk[i] = mfcc ? pw.x : __builtin_amdgcn_sqrtf(pw.x);   // v_sqrt_f32, 1 ulp
                mn[i] = mfcc ? pw.y : __builtin_amdgcn_sqrtf(pw.y);
            }
            frame_sync<P>();   // every Z read of this frame is done before S overwrites it
            float* sf = reinterpret_cast<float*>(buf);   // S[c], c = bin - 1, c = 0..N-1
            float* sk = sf + (po == 0 ? N / 2 : po) - 1;   // bin k of i = 0 (lane 0: bin N/2)
            float* sn = sf + N - po - 1;                   // bin N - k of i = 0
            sk[0] = mk[0];
            sn[0] = mn[0];
#pragma unroll
            for (int i = 1; i < E / 2; ++i) {
                sf[po + i * P - 1] = mk[i];
                sn[-i * P] = mn[i];
            }
        }
#ifndef ZAFX_MEL_EARLY
#define ZAFX_MEL_EARLY 1
#endif
        // opaque copy of the thread index for the filterbank phases: their per-lane addresses are tile-invariant, and hoisted
        // out of the persistent loop they ride through the FFT phase (with the prefetched samples: 128 VGPRs + 212 B of scratch)
        int to = tid;
        asm volatile("" : "+v"(to));
        const int lane = to & 63, bt = lane & 15, bk = lane >> 4;
        // The next tile's frame.  EARLY: requested by each wave as soon as ITS spectrum is written, ahead of the barrier -- the waves
        // finish their transforms 6 k cycles apart (oldest first), so the sixteen bursts of eight loads arrive spread out and
        // vector memory, idle through the transforms, works while the early waves wait at the barrier; the filterbank GEMM then
        // runs without loads in between.  Otherwise: one load per K-step of the GEMM.
        bool fast = false;
        if constexpr (TEAM8 || (PAIR16 && ZAFX_MEL_GEMM_PRE && ZAFX_MEL_RELOAD_A)) load_fragments(lane);   // from L2, ahead of the request of the next frame (they are needed first)
        if constexpr (LATE) {
            fast = fetch_begin(tlv + gridDim.x, 0, PAIR16 ? row_pair_index(to % P) : to % P);
            if (ZAFX_MEL_EARLY && interior) {
#pragma unroll
                for (int i = 0; i < E; ++i) fetch_one(i);
            }
        }
        raw = fast;
        PROF_MARK(1);
        lds_barrier();
        PROF_MARK(2);

        // ---- mel = FB . S on the matrix cores
        constexpr bool GEMM_PRE = PAIR16 && ZAFX_MEL_GEMM_PRE && ZAFX_MEL_PREPASS && ZAFX_MEL_EARLY;
        {
            const float* sb = fall + (size_t)(bt % FPB) * (2 * C::PITCH) + bk;   // (8-frame tiles: columns 8..15 repeat 0..7 and are not used)
            if constexpr (RES && GEMM_PRE) {
                // The next frame's window + first radix-16 butterflies (500 cycles of vector issue per wave) beside the filterbank product
                // (four dependent MFMA chains keep a SIMD's matrix pipe for 2.3 k cycles, its vector pipe idle): of the four waves of a
                // SIMD (w, w + 4, w + 8, w + 12) two run the butterflies first and the product second, two the other way round.
                pre_done = tlv + gridDim.x < total_tiles;   // (uniform)
                const bool pre_first = (wave & 8) == 0;   // (the waves that finished their transforms first: their samples were requested thousands of cycles ago)
                auto pre = [&]() {
                    if (pre_done) {
                        int pq = p;
                        asm volatile("" : "+v"(pq));
                        pre_transform(pq);
                    }
                };
                if (pre_first) pre();
                gemm_resident(afb, sfb, nfb, lane, 0, slot_ptr, [&](int col) { return sb[col]; }, [](int) {});
                if (!pre_first) pre();
            } else if constexpr (RES)
                gemm_resident(afb, sfb, nfb, lane, 0, slot_ptr, [&](int col) { return sb[col]; }, [&](int i) { if (LATE && !ZAFX_MEL_EARLY && i < E && fast) fetch_one(i); });
            else gemm_items(fb_pack, fb_items, fb_wave_ptr, wave, lane, 0, slot_ptr, [&](int col) { return sb[col]; });
        }
        PROF_MARK(3);
        lds_barrier();
        PROF_MARK(4);
        // ---- fixed-order reduction of the parts of every 16-filter block
        // (a block = four whole waves; with NT a multiple of 256 a wave's blocks are wave / 4 + (NT / 256) j: scalar and the same
        // for every tile, so the ranges of their parts are loaded into SGPRs once, ahead of the tile loop)
        float lm[2] = {0.f, 0.f};   // (DIRECT) log-mel values of this thread: element of the B fragment of the DCT's K-steps wave, wave + 16
        for (int j = 0; (NT % 256 == 0) ? j * (NT / 256) + (wave >> 2) < fb_blocks : j * NT + to < fb_blocks * 256; ++j) {
            const int idx = to + j * NT;
            const int blk = (NT % 256 == 0) ? j * (NT / 256) + (wave >> 2) : __builtin_amdgcn_readfirstlane(idx >> 8), e = idx & 255;
            float val = 0.f;
            for (int it = fb_blk_ptr[blk]; it < fb_blk_ptr[blk + 1]; ++it) val += slot_ptr(it)[e];
            const int m = 16 * blk + (e >> 4), tq = e & 15;
            if (FPB < 16 && tq >= FPB) continue;
            if (mfcc) {
                const float l = m < n_filters ? logf(val + eps) : 0.f;
                if constexpr (DIRECT) {
                    if (j == 0) lm[0] = l;
                    else lm[1] = l;
                } else {
                    slot_ptr(lt0 + blk)[e] = l;
                }
            } else if (m < n_filters && t0 + tq < T) {
                if (layout == ZAFX_LAYOUT_FT) out[((long long)clip * n_filters + m) * TP + t0 + tq] = val;   // TP = row pitch (>= T)
                else out[((long long)clip * T + t0 + tq) * n_filters + m] = val;
            }
        }
        if (mfcc) {
            if constexpr (DIRECT) {
                // ---- rows 1..ncoef of the orthonormal DCT-II over the mel axis, register fed: the thread's log-mel value of pass j is
                // element (k = 4 s + (lane >> 4), n = lane & 15) of the B fragment of K-step s = wave + 16 j -- no trip through LDS, no
                // barrier between the logarithm and the product.  Every wave leaves one partial tile per 16-row block (slots behind
                // the filterbank's: those are still being read by the other waves).
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    if (b < dct_blocks) {
                        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(adct[b], lm[0], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(adct[2 + b], lm[1], acc, 0, 0, 0);
                        float* dst = slot_ptr(lt0 + b * NW + wave) + (4 * bk) * 16 + bt;
#pragma unroll
                        for (int q = 0; q < 4; ++q) dst[q * 16] = acc[q];
                    }
                }
                lds_barrier();
                for (int idx = to; idx < dct_blocks * 256; idx += NT) {
                    const int blk = idx >> 8, e = idx & 255;
                    float val = 0.f;
#pragma unroll
                    for (int w = 0; w < NW; ++w) val += slot_ptr(lt0 + blk * NW + w)[e];
                    const int q = 16 * blk + (e >> 4), tq = e & 15;
                    if (q < n_coefs && tq < FPB && t0 + tq < T) {
                        if (layout == ZAFX_LAYOUT_FT) out[((long long)clip * n_coefs + q) * TP + t0 + tq] = val;
                        else out[((long long)clip * T + t0 + tq) * n_coefs + q] = val;
                    }
                }
            } else {
            lds_barrier();
            // ---- rows 1..ncoef of the orthonormal DCT-II over the mel axis: second MFMA GEMM
            auto logmel = [&](int row) { return slot_ptr(lt0 + ((row + bk) >> 4))[((row + bk) & 15) * 16 + bt]; };
            if constexpr (RES) gemm_resident(adct, sdct, ndct, lane, dslot0, slot_ptr, logmel, [](int) {});
            else gemm_items(dct_pack, dct_items, dct_wave_ptr, wave, lane, dslot0, slot_ptr, logmel);
            lds_barrier();
            for (int j = 0; (NT % 256 == 0) ? j * (NT / 256) + (wave >> 2) < dct_blocks : j * NT + to < dct_blocks * 256; ++j) {
                const int idx = to + j * NT;
                const int blk = (NT % 256 == 0) ? j * (NT / 256) + (wave >> 2) : __builtin_amdgcn_readfirstlane(idx >> 8), e = idx & 255;
                float val = 0.f;
                for (int it = dct_blk_ptr[blk]; it < dct_blk_ptr[blk + 1]; ++it) val += slot_ptr(dslot0 + it)[e];
                const int q = 16 * blk + (e >> 4), tq = e & 15;
                if (q < n_coefs && tq < FPB && t0 + tq < T) {
                    if (layout == ZAFX_LAYOUT_FT) out[((long long)clip * n_coefs + q) * TP + t0 + tq] = val;
                    else out[((long long)clip * T + t0 + tq) * n_coefs + q] = val;
                }
            }
            }
        }
        PROF_MARK(5);
        if constexpr (PAIR16 && !GEMM_PRE) {
            // window and first radix-16 butterflies of the next tile's frame, in registers, AHEAD of the barrier that frees the frame
            // buffers: only the writes of pass 1 have to wait for the other waves' reductions
            pre_done = ZAFX_MEL_PREPASS && tlv + gridDim.x < total_tiles;
            if (pre_done) {
                int pq = p;
                asm volatile("" : "+v"(pq));
                pre_transform(pq);
            }
        }
        lds_barrier();   // slots and S are dead: the next tile may overwrite the frame buffers
    }
}

// ---------------------------------------------------------------------------------
// k_mel2: the filterbank product of tile i UNDER the transforms of tile i + 1 (W = 2048; mel up to 256 filters, mfcc up to 128 filters x 32 rows)
// ---------------------------------------------------------------------------------
// k_mel's tile is transforms | barrier | product | barrier | reduction | barrier: sixteen waves in step, the vector pipe idle through the
// last two thirds (54 % issue over the kernel), the matrix pipe idle through the first.  Here a frame buffer is two halves: the lower 4 KB
// hold S[t][c] of the CURRENT tile, the upper 4.6 KB are the exchange area of the wave's transform of the NEXT tile's frame -- the first
// exchange in two rounds of eight outputs per lane (readers ll < 8, then ll >= 8), the spectrum's upper half staged once for the split
// (a lane holds Z[lane + 64 b + 256 r] after pass 3: its own k < 512 are in registers, only the partners N - k come from LDS).  The
// magnitudes wait in registers for the barrier behind which nobody reads S any more.  The product runs on WHOLE 16-filter blocks
// (PackedBand::d_whole: the widest cut in K, dealt so that every SIMD's matrix pipe carries a quarter): a wave's accumulators are
// the block's tile -- no partial-tile slots, no reduction phase (a cut block's helper hands one partial tile over through 1 KB of LDS).
// Per tile: [product of tile i (waves that own items) ; transform of the frame of tile i + 1] | barrier | S <- magnitudes, request of tile
// i + 2 | barrier | stores of tile i.  Two barriers instead of four, matrix and vector pipes busy in the same phase.
#ifndef ZAFX_MEL2
#define ZAFX_MEL2 1
#endif
constexpr int kMel2DualPitch = 1057;   // float2 slots per frame buffer of the one-pass mel + mfcc form (MODE 4)
// MFCC: the levels are |X|^2 (zaf.py:437-439); between the two barriers the owners take log(mel + eps) of their tiles, turn them into B
// fragments (a 4 x 4 transpose of register index against 16-lane row: two v_permlane swaps) and multiply them with their block's
// columns of the DCT-II rows (zaf.py:443-452; A fragments resident: 8 registers per owned block); the partial coefficient tiles go through
// the exchange areas -- free between the barriers -- and 16 x n_coefs threads add them in block order.  Two more barriers per tile.
// MODE 4 (round 6, BASELINE config 3 in ONE pass: zaf.melspectrogram and zaf.mfcc start with the same stft call, zaf.py:369 and :436): mfcc WITH the
// melspectrogram of the same transforms.  The levels in LDS are the powers; a K-step of the product issues two matrix instructions on the same
// A fragment (the filterbank scaled for the powers) -- B = 4 |X|^2 and B = 2 sqrt(4 |X|^2) = 2 (2 |X|): the very v_sqrt_f32 MODE 0 takes, and the
// factor 2 against MODE 0's fragments is a power of two, so both tiles come out bit for bit as the single-output kernels' -- and an owner stores
// its mel tile (rows 0 .. n_filters - 1 of the clip's output) before it goes on with log and DCT (rows n_filters ...).
// MODE 0 mel, 1 mfcc; 2 / 3: the one-sided |X| / |X|^2 spectrogram of the STFT (zaf.py:83; ZAFX_SPECTRUM_MAGNITUDE / _POWER at W = 2048 in
// the reference layout): the same transforms and levels, and in the product's place the tile's rows 0 .. N leave from the levels in LDS
// (thread = frame tid & 15 x rows (tid >> 4) + 64 j: 64-byte runs) -- on k_stft_ft16's 8 fat waves these kinds ran at 3.2 TB/s (1.12 ms
// per 1024 x 10 s), bound by the transforms of two frames per wave, not by their 3.6 GB.
// PCM: what `x` holds (wavread's normalisation x / 2^15 and the channel mean, zaf.py:1202 and :65, inside the loads) --
// 0 float32 samples; 1 int16 mono (n_samples 2-byte samples per clip: 8 bytes per lane and load instead of 16); 2 int16 stereo
// (n_samples 4-byte frames per clip: the float32 form's loads, both channels added on the way into the window multiply).  The factor
// 2^-15 (2^-16: the mean of two channels) is a power of two and rides in the window.
template <bool ALIGNED, int MODE, int PCM = 0>
__global__ __launch_bounds__(1024, 1) void k_mel2(const float* __restrict__ x, const float* __restrict__ win, const float2* __restrict__ twp,
                                                   const float2* __restrict__ tws, const float* __restrict__ fb_pack, const int4* __restrict__ fb_whole,
                                                   const float* __restrict__ dct2, const int* __restrict__ owner2, float* __restrict__ out, long long n_samples, int hop,
                                                   int T, int TP, int tiles, int total_tiles, int n_filters, int n_coefs, int layout) {
    using C = FftCfg<10, 4>;
    constexpr bool DUAL = MODE == 4, MFCC = MODE == 1 || DUAL, SPECM = MODE == 2 || MODE == 3, SQUARES = MODE == 1 || MODE == 3 || DUAL;
    // (DUAL: frame buffers 32 slots shorter -- still 2 mod 64 floats apart, the exchange area still holds what the asserts below ask for -- : the 4 KB
    // make room for the second set of partial tiles; the other modes keep the pitch they were tuned on, and bin 0's slot behind it)
    constexpr int N = C::N, P = 64, E = 16, W = 2 * N, NT = 1024, FPB = 16, PITCH = DUAL ? kMel2DualPitch : C::PITCH, EXOFF = N / 2;
    constexpr int DCSLOT = 2 * (EXOFF + N / 2 + N / 32) + 8;   // (SPECM) float slot of bin 0's level in a frame buffer: behind the exchange area
    static_assert(!SPECM || DCSLOT < 2 * PITCH, "bin 0's level fits the frame buffer");   // exchange area: float2 slots EXOFF .. PITCH - 1 of a frame buffer
    static_assert(PITCH - EXOFF >= 8 * 63 + 31 + 8 && PITCH - EXOFF >= N / 2 + N / 32, "exchange area holds a round of the first exchange and the staged half spectrum");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* frames = reinterpret_cast<float2*>(smem_raw);
    float2* tw_l = frames + FPB * PITCH;
    float2* win_s = tw_l + C::TW;
    float2* tws_s = win_s + N;
    float* xpart = reinterpret_cast<float*>(tws_s + N / 2 + 1);   // kMel2Slots partial tiles of cut blocks (DUAL: kMel2Slots more behind them, the mel tiles)
    float* fall = reinterpret_cast<float*>(frames);
    const int tid = threadIdx.x;
    for (int i = tid; i < C::TW; i += NT) tw_l[i] = twp[i];
    {
        const float pcm_scale = PCM == 1 ? 1.f / 32768.f : PCM == 2 ? 1.f / 65536.f : 1.f;
        for (int i = tid; i < N; i += NT) {   // lane order: win_s[64 i + lane]
            const float2 w = reinterpret_cast<const float2*>(win)[(i & ~63) + row_pair_index(i & 63)];
            win_s[i] = make_float2(w.x * pcm_scale, w.y * pcm_scale);
        }
    }
    for (int i = tid; i <= N / 2; i += NT) tws_s[i] = tws[i];
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool xcd = ZAFX_XCD_ORDER && gridDim.x % 8 == 0;
    // this wave's items of the product (scalar)
    // this wave's items of the product (scalar)
    int it_first[2], it_steps[2], it_off[2], it_code[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int4 m = SPECM ? make_int4(0, -1, 0, 0) : fb_whole[wave * 2 + j];   // (the spectrogram kinds have no product)
        it_first[j] = __builtin_amdgcn_readfirstlane(m.x);
        it_steps[j] = __builtin_amdgcn_readfirstlane(m.y);   // < 0: no item
        it_off[j] = __builtin_amdgcn_readfirstlane(m.z);
        it_code[j] = __builtin_amdgcn_readfirstlane(m.w);
    }
    __syncthreads();

    // ---- samples of the wave's frame of tile `tlv` (k_mel's pair form: 16-byte loads shared by the lanes l and l + 16 of a row pair)
    PROF_INIT(g_prof_mel);
    // xr[2 i], xr[2 i + 1]: the two points (n, n + 1), n = (p & ~1) + 512 (p & 1) + 64 i, of one 16-byte load -- also where the frame touches
    // the clip's ends and the samples come one by one: ONE register arrangement into the transform (a second one, chosen at run time,
    // cost ~50 v_mov per frame where the two paths meet; round 5)
    float2 xr[E];
    __amdgpu_buffer_rsrc_t frx = make_rsrc(x, 0);
    // (a request past the last tile repeats the last one: xr is defined on EVERY path of every iteration -- left conditional, the old values
    // count as live through the transform and its registers spill)
    auto request = [&](int tlv, int lane) {   // lane: an opaque copy
        tlv = min(tlv, total_tiles - 1);
        const int p = row_pair_index(lane);
        const int tl = xcd ? xcd_order(tlv, total_tiles) : tlv;
        const int clip = tl / tiles, tile = tl % tiles;
        const int t = tile * FPB + wave;
        const long long s0 = (long long)t * hop - N;
        if constexpr (PCM == 1) {
            // int16 mono: a point (two samples) is one dword; xr[2 i] = the dwords of the points n, n + 1 of load i (xr[2 i + 1] is not used)
            const short* xc = reinterpret_cast<const short*>(x) + (long long)clip * n_samples;
            if (ALIGNED && t < T && s0 >= 0 && s0 + W <= n_samples) {   // interior frame (uniform)
                frx = make_rsrc(xc, (unsigned)std::min<long long>(n_samples * 2, 0xfffffffcLL));
                const int fvoff = ((int)s0 + 2 * (p & ~1)) * 2 + (p & 1) * (E / 2 * P * 4);
#pragma unroll
                for (int i = 0; i < E / 2; ++i) xr[2 * i] = buf_load_f32x2(frx, fvoff, i * P * 4);
                return;
            }
            const int nb = (p & ~1) + (p & 1) * (E / 2 * P);
#pragma unroll
            for (int i = 0; i < E; ++i) {   // a frame that touches the clip's ends: sample by sample, packed the way the loads deliver them
                const long long s = s0 + 2 * (nb + (i >> 1) * P + (i & 1));
                const unsigned lo = (t < T && s >= 0 && s < n_samples) ? (unsigned short)xc[s] : 0u;
                const unsigned hi = (t < T && s + 1 >= 0 && s + 1 < n_samples) ? (unsigned short)xc[s + 1] : 0u;
                const float bits = __builtin_bit_cast(float, lo | hi << 16);
                if (i & 1) xr[i - 1].y = bits;
                else xr[i].x = bits;
            }
            return;
        }
        const float* xc = x + (long long)clip * n_samples;   // (PCM == 2: a "sample" is one frame of two int16, moved as the 4 bytes it is)
        if (ALIGNED && t < T && s0 >= 0 && s0 + W <= n_samples) {   // interior frame (uniform)
            frx = make_rsrc(xc, (unsigned)std::min<long long>(n_samples * 4, 0xfffffffcLL));
            const int fvoff = ((int)s0 + 2 * (p & ~1)) * 4 + (p & 1) * (E / 2 * P * 8);
#pragma unroll
            for (int i = 0; i < E / 2; ++i) {
                const float4 q = buf_load_f32x4(frx, fvoff, i * P * 8);
                xr[2 * i] = make_float2(q.x, q.y);
                xr[2 * i + 1] = make_float2(q.z, q.w);
            }
            return;
        }
        const int nb = (p & ~1) + (p & 1) * (E / 2 * P);
#pragma unroll
        for (int i = 0; i < E; ++i) {   // a frame that touches the clip's ends or lies past its last frame: zero padding (zaf.py:112-125)
            const long long s = s0 + 2 * (nb + (i >> 1) * P + (i & 1));
            xr[i].x = (t < T && s >= 0 && s < n_samples) ? xc[s] : 0.f;
            xr[i].y = (t < T && s + 1 >= 0 && s + 1 < n_samples) ? xc[s + 1] : 0.f;
        }
    };
    // ---- transform of the requested frame; returns 4 |X|^2 (the filterbank carries the 1/2 of |X|) of the bins k = lane + 64 i (mk) and N - k (mn)
    auto transform = [&](int lane, float (&mk)[E / 2], float (&mn)[E / 2], float& dc) {   // dc (SPECM, lane 0): 4 |X[0]|^2 or its root
        float2* ex = frames + wave * PITCH + EXOFF;
        const int p1 = row_pair_index(lane);
        float2 v[E];
#pragma unroll
        for (int i = 0; i < E / 2; ++i) {   // the lane's own points i and i + 8 out of the pair's two loads
            float2 qa = xr[2 * i], qb = xr[2 * i + 1];
            if constexpr (PCM != 0) {   // integers -> floats (the scale is in the window)
                auto lo16 = [](float f) { return (float)(short)(__builtin_bit_cast(int, f) & 0xffff); };
                auto hi16 = [](float f) { return (float)(__builtin_bit_cast(int, f) >> 16); };
                auto sum16 = [](float f) { const int d = __builtin_bit_cast(int, f); return (float)((int)(short)(d & 0xffff) + (d >> 16)); };
                if constexpr (PCM == 1) {
                    qb = make_float2(lo16(qa.y), hi16(qa.y));
                    qa = make_float2(lo16(qa.x), hi16(qa.x));
                } else {
                    qa = make_float2(sum16(qa.x), sum16(qa.y));
                    qb = make_float2(sum16(qb.x), sum16(qb.y));
                }
            }
            row_pair_unpack(qa, qb);
            v[i] = mul_elem(qa, lds_ld(win_s + lane + i * P));
            v[i + E / 2] = mul_elem(qb, lds_ld(win_s + lane + (i + E / 2) * P));
        }
#ifdef ZAFX_PROF
        asm volatile("" :: "v"(v[0].x), "v"(v[15].y));
        PROF_MARK(10);
#endif
        Dft<16>::run(v);
        // first exchange, two rounds: position 16 p1 + r, r < 8 then r >= 8, at slot 8 p1 + (p1 >> 1) + (r & 7); lane (lh, ll) reads
        // position lane + 64 i = 16 (lh + 4 i) + ll in the round that holds r = ll
        const int wb = 8 * p1 + (p1 >> 1);
        const int lh = lane >> 4, ll = lane & 15;
        const int rb = 8 * lh + (lh >> 1) + (ll & 7);
        float2 u[E];
#pragma unroll
        for (int r = 0; r < 8; ++r) ex[wb + r] = v[r];
        frame_sync<64>();
        if (ll < 8) {
#pragma unroll
            for (int i = 0; i < E; ++i) u[i] = lds_ld(ex + rb + 34 * i);
        }
        frame_sync<64>();
#pragma unroll
        for (int r = 0; r < 8; ++r) ex[wb + r] = v[r + 8];
        frame_sync<64>();
        if (ll >= 8) {
#pragma unroll
            for (int i = 0; i < E; ++i) u[i] = lds_ld(ex + rb + 34 * i);
        }
        frame_sync<64>();
        PROF_MARK(6);
        float2 a[E];
        {
            float2 w[E];
            {
                const float2* t = (const float2*)tw_l + twiddle_offset(10, 4, 4) + (lane & 15);
#pragma unroll
                for (int r = 1; r < E; ++r) w[r] = lds_ld(t + (r - 1) * 16);
            }
            a[0] = u[0];
#pragma unroll
            for (int r = 1; r < E; ++r) a[r] = cmul(u[r], w[r]);
        }
        Dft<16>::run(a);
#pragma unroll
        for (int rh = 0; rh < 4; ++rh) {
            lane_row_transpose4(a[4 * rh], a[4 * rh + 1], a[4 * rh + 2], a[4 * rh + 3]);
        }
        PROF_MARK(7);
        // pass 3 in registers: z[b][r] = Z[lane + 64 b + 256 r]  (register 4 b + r' holds position lane + 64 b + 256 r')
        float2 z[4][4];
        const float2* t3 = (const float2*)tw_l + twiddle_offset(10, 4, 8);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int k = lane + 64 * b;
            z[b][0] = a[4 * b];
#pragma unroll
            for (int r = 1; r < 4; ++r) z[b][r] = cmul(a[4 * b + r], lds_ld(t3 + (r - 1) * 256 + k));
            dft4(z[b][0], z[b][1], z[b][2], z[b][3]);
        }
        // upper half of the spectrum to LDS: Z[512 + q] at phys(q), q = lane + 64 b + 256 (r - 2)
        const int pq = phys(lane);
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 2; r < 4; ++r) ex[phys_off<64>(pq, lane, 64 * b + 256 * (r - 2))] = z[b][r];
        frame_sync<64>();
        PROF_MARK(8);
#pragma unroll
        for (int i = 0; i < E / 2; ++i) {
            const int k = lane + i * P;
            const float2 za = z[i & 3][i >> 2];
            const float2 zb = lds_ld(ex + phys(N / 2 - (k == 0 ? N / 2 : k)));   // Z[N - k] (k = 0: Z[N / 2], staged at q = 0)
            float2 pw = split_pair_pow4(za, zb, lds_ld(tws_s + k));   // (4 |X[k]|^2, 4 |X[N-k]|^2)
            if (i == 0 && k == 0) {
                const float ny = za.x - za.y;   // X[N] (Nyquist, kept: zaf.py:370)
                pw = make_float2(4.f * (zb.x * zb.x + zb.y * zb.y), 4.f * (ny * ny + 0.f * 0.f));   // |X[N/2]| = |Z[N/2]|
            }
            if constexpr (SPECM) {
                if (i == 0) {   // X[0] = Re Z[0] + Im Z[0] (lane 0's za)
                    const float d0 = za.x + za.y;
                    dc = SQUARES ? 4.f * (d0 * d0) : 2.f * fabsf(d0);
                }
            }
            mk[i] = SQUARES ? pw.x : __builtin_amdgcn_sqrtf(pw.x);   // v_sqrt_f32, 1 ulp
            mn[i] = SQUARES ? pw.y : __builtin_amdgcn_sqrtf(pw.y);
        }
        frame_sync<64>();
        PROF_MARK(9);
    };
    auto put_levels = [&](int lane, const float (&mk)[E / 2], const float (&mn)[E / 2], float dc) {   // S[c], c = bin - 1, into the lower half of the wave's buffer
        float* sf = reinterpret_cast<float*>(frames + wave * PITCH);
        float* sk = sf + (lane == 0 ? N / 2 : lane) - 1;   // bin k of i = 0 (lane 0: bin N / 2)
        float* sn = sf + N - lane - 1;                     // bin N - k of i = 0 (lane 0: bin N)
        sk[0] = mk[0];
        sn[0] = mn[0];
#pragma unroll
        for (int i = 1; i < E / 2; ++i) {
            sf[lane + i * P - 1] = mk[i];
            sn[-i * P] = mn[i];
        }
        if constexpr (SPECM) {
            if (lane == 0) sf[DCSLOT] = dc;
        }
    };
    // (SPECM) rows 0 .. N of the tile whose levels are in LDS (2 |X| / 4 |X|^2).  Rows of whole 16-byte pieces (pitch % 4 = 0, base aligned; 8-byte
    // pieces for the other even pitches): thread = four frames (tid & 3) x rows (tid >> 2) + 256 j, one 16-byte store per row -- 5 vector-memory instructions per thread and
    // tile instead of 17 four-byte ones (the youngest waves waited 5.5 k cycles at the store queue).  Else thread = frame x rows, 4 bytes.
    const int rowv = !SPECM ? 1 : (TP % 4 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0) ? 4 : (TP % 2 == 0 && reinterpret_cast<uintptr_t>(out) % 8 == 0) ? 2 : 1;   // frames per lane and store
    auto store_rows = [&](int tlv) {
        int to = tid;
        asm volatile("" : "+v"(to));
        const int tl = xcd ? xcd_order(tlv, total_tiles) : tlv;
        const int clip = tl / tiles, t0 = (tl % tiles) * FPB;
        const float sc = SQUARES ? 0.25f : 0.5f;
        if (rowv == 4 && t0 + FPB <= T) {   // (uniform) a whole tile, 16-byte pieces
            const int tq = to & 3, kq = to >> 2;
            const float* sf = fall + (size_t)(4 * tq) * (2 * PITCH);
            float* o = out + (long long)clip * (N + 1) * TP + t0 + 4 * tq;
            float4 v[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int k = j < 4 ? kq + 256 * j : N, at = k == 0 ? DCSLOT : k - 1;
                v[j] = make_float4(sf[at], sf[at + 2 * PITCH], sf[at + 4 * PITCH], sf[at + 6 * PITCH]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<float4*>(o + (long long)(kq + 256 * j) * TP) = make_float4(sc * v[j].x, sc * v[j].y, sc * v[j].z, sc * v[j].w);
            if (kq == 0) *reinterpret_cast<float4*>(o + (long long)N * TP) = make_float4(sc * v[4].x, sc * v[4].y, sc * v[4].z, sc * v[4].w);
            return;
        }
        if (rowv == 2 && t0 + FPB <= T) {   // (uniform) a whole tile, 8-byte pieces (an even pitch that is not a multiple of 4)
            const int tq = to & 7, kq = to >> 3;
            const float* sf = fall + (size_t)(2 * tq) * (2 * PITCH);
            float* o = out + (long long)clip * (N + 1) * TP + t0 + 2 * tq;
            float2 v[9];
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const int k = j < 8 ? kq + 128 * j : N, at = k == 0 ? DCSLOT : k - 1;
                v[j] = make_float2(sf[at], sf[at + 2 * PITCH]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) *reinterpret_cast<float2*>(o + (long long)(kq + 128 * j) * TP) = make_float2(sc * v[j].x, sc * v[j].y);
            if (kq == 0) *reinterpret_cast<float2*>(o + (long long)N * TP) = make_float2(sc * v[8].x, sc * v[8].y);
            return;
        }
        const int t = t0 + (to & 15), kq = to >> 4;
        if (t >= T) return;
        const float* sf = fall + (size_t)(to & 15) * (2 * PITCH);
        float* o = out + (long long)clip * (N + 1) * TP + t;
        float v[E + 1];
#pragma unroll
        for (int j = 0; j < E; ++j) v[j] = sf[kq + 64 * j == 0 ? DCSLOT : kq + 64 * j - 1];
        v[E] = sf[N - 1];
#pragma unroll
        for (int j = 0; j < E; ++j) o[(long long)(kq + 64 * j) * TP] = sc * v[j];
        if (kq == 0) o[(long long)N * TP] = sc * v[E];
    };
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    f32x4v acc[2][2];
    f32x4v accm[DUAL ? 2 : 1];   // (DUAL) the mel tiles of the wave's items, their two chains already added
    // ---- the wave's items of FB . S for the tile whose levels are in LDS: A fragments from L2, eight steps' operands requested together.
    // (Measured and dropped, profiles/r04_notes.md: fragments requested a chunk or a whole tile ahead, items of <= 24 steps on every
    // wave with the partial tiles through L2, items on eight waves only -- each lost more to registers or to sixteen matrix-instruction
    // chains starting together than it gained on the 9 k cycles the longest item waits for L2 here.)
    const __amdgpu_buffer_rsrc_t fbr = make_rsrc(fb_pack, 0xfffffffcu);
    auto product = [&](int lane) {
        const float* sb = fall + (size_t)(lane & 15) * (2 * PITCH) + (lane >> 4);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            acc[j][0] = acc[j][1] = f32x4v{0.f, 0.f, 0.f, 0.f};
            f32x4v am[2] = {f32x4v{0.f, 0.f, 0.f, 0.f}, f32x4v{0.f, 0.f, 0.f, 0.f}};
            if constexpr (DUAL) accm[j] = am[0];
            const int steps = it_steps[j];
            if (steps <= 0) continue;
            // A fragments: buffer loads, the step in the scalar offset (no 64-bit address per lane and load); B: the levels of LDS at immediate
            // offsets from one address per chunk.  Past the item's last step both read on (the next item's fragments, the frame buffer behind S):
            // never multiplied.
            const float* bp = sb + it_first[j];
            // (chunks of 16 or 32 steps -- fewer round trips to L2 -- are slower: 0.92 against 0.91 ms, and 4.3 ms with 32 loads per wave in the
            // vector-memory queue; profiles/r05_notes.md)
            for (int s0 = 0; s0 < steps; s0 += 8) {
                float a[8], b[8];
                const float* bq = bp + 4 * s0;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    a[u] = buf_load_f32(fbr, lane * 4, (it_off[j] + min(s0 + u, steps - 1)) * 256);
                    b[u] = bq[4 * u];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (s0 + u < steps) {
                        acc[j][u & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc[j][u & 1], 0, 0, 0);
                        if constexpr (DUAL) {
                            const float m2 = __builtin_amdgcn_sqrtf(b[u]);   // 2 |X| (MODE 0's level); twice that against fragments scaled for the powers
                            am[u & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], m2 + m2, am[u & 1], 0, 0, 0);
                        }
                    }
            }
            if constexpr (DUAL) accm[j] = am[0] + am[1];   // (MODE 0 adds its two chains in the epilogue: the same sum)
        }
    };

    // (MFCC) the DCT rows over the filters of the blocks this wave owns: [item][coefficient block][step]
    float dfr[MFCC ? 2 : 1][8];
    if constexpr (MFCC) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool own = it_steps[j] >= 0 && ((it_code[j] >> 8) & 3) == 1;
            const float* dp = dct2 + (size_t)(own ? it_code[j] & 255 : 0) * 8 * 64 + (tid & 63);
#pragma unroll
            for (int u = 0; u < 8; ++u) dfr[j][u] = dp[u * 64];
        }
    }
    const int n_blocks = (n_filters + 15) >> 4;
    int owners[8];   // (MFCC, scalar) wave | place << 8 of the item that owns block b (at most 8 blocks: build_mel2_dct)
#pragma unroll
    for (int bb = 0; bb < 8; ++bb) owners[bb] = MFCC && bb < n_blocks ? __builtin_amdgcn_readfirstlane(owner2[bb]) : 0;
    // a cut block's helper hands its partial tile over.  mel: behind the first barrier (the owner adds it behind the second).  mfcc: right
    // behind the product -- the owner of the tile before read its parts two barriers ago --, so that the owners' stage can start at the
    // first barrier: three barriers per tile instead of four.
    auto hand_over = [&](int lane) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (it_steps[j] >= 0 && ((it_code[j] >> 8) & 3) == 2) {
                float* dst = xpart + ((it_code[j] >> 10) & 3) * 256 + (4 * (lane >> 4)) * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[r * 16] = acc[j][0][r] + acc[j][1][r];
                if constexpr (DUAL) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[kMel2Slots * 256 + r * 16] = accm[j][r];
                }
            }
        }
    };
    int tlv = blockIdx.x;
    if (tlv >= total_tiles) return;   // (uniform; the launcher's grid never exceeds the tiles)
    {
        float mk[E / 2], mn[E / 2];
        int lane = tid & 63;
        asm volatile("" : "+v"(lane));
        float dc = 0.f;
        request(tlv, lane);
        transform(lane, mk, mn, dc);
        put_levels(lane, mk, mn, dc);
        request(tlv + gridDim.x, lane);
        lds_barrier();
    }
    // One tile.  HAS_NEXT is a compile-time flag: the workgroup's last tile is peeled off below, so the transform of the loop body is
    // unconditional -- no run-time test around it, no levels zero-initialised for the path that skips it (16 v_mov per frame, round 5).
    auto tile_body = [&](auto has_next_c) {
        constexpr bool has_next = decltype(has_next_c)::value;
        PROF_MARK(0);
        int lane = tid & 63;
        asm volatile("" : "+v"(lane));   // (per-lane addresses are recomputed per tile, not carried through the transform)
#ifndef ZAFX_MEL2_EARLYOUT
#define ZAFX_MEL2_EARLYOUT 1
#endif
        float mk[E / 2], mn[E / 2], dc = 0.f;
        // (the transform first and the product behind it: 0.98 against 0.91 ms; profiles/r05_notes.md)
        if constexpr (SPECM) store_rows(tlv);   // (behind the wave's own transform instead: the same on the line grid, 7 % slower off it)
        else product(lane);
        if constexpr (!SPECM && (MFCC || ZAFX_MEL2_EARLYOUT)) hand_over(lane);
        PROF_MARK(1);
        if constexpr (has_next) transform(lane, mk, mn, dc);
        // the frame after that: requested by each wave as soon as ITS transform is done -- the waves finish thousands of cycles apart, so
        // the sixteen bursts of eight loads arrive spread out (all at once they overrun the CU's vector-memory queue: 2.5-5 k cycles blocked)
        if constexpr (has_next) request(tlv + 2 * gridDim.x, lane);
        PROF_MARK(2);
        lds_barrier();   // nobody reads the current levels any more; every transform is done with its exchange area
        PROF_MARK(3);
        if constexpr (has_next) put_levels(lane, mk, mn, dc);
        if constexpr (SPECM) {
            lds_barrier();   // the next tile's levels are in LDS
            return;
        }
        if constexpr (!MFCC && !ZAFX_MEL2_EARLYOUT) hand_over(lane);
        PROF_MARK(4);
        if constexpr (!MFCC && !ZAFX_MEL2_EARLYOUT) lds_barrier();   // the next tile's levels and the partial tiles are in LDS
        PROF_MARK(5);
        const int tl = xcd ? xcd_order(tlv, total_tiles) : tlv;
        const int clip = tl / tiles, t0 = (tl % tiles) * FPB;
        float* exf = reinterpret_cast<float*>(frames + wave * PITCH + EXOFF);   // (MFCC) this wave's exchange area: free until the barrier that ends the tile
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (it_steps[j] < 0 || ((it_code[j] >> 8) & 3) != 1) continue;   // (the item loop's own continue)
            const int blk = it_code[j] & 255, mask = (it_code[j] >> 12) & 7;
            float val[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                val[r] = acc[j][0][r] + acc[j][1][r];
                const int e = (4 * (lane >> 4) + r) * 16 + (lane & 15);
#pragma unroll
                for (int h = 0; h < kMel2Slots; ++h)
                    if (mask & (1 << h)) val[r] += xpart[h * 256 + e];
            }
            if constexpr (DUAL) {   // the melspectrogram's tile: rows 0 .. n_filters - 1 of the clip's n_filters + n_coefs rows
                const int t = t0 + (lane & 15), rows = n_filters + n_coefs;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float vm = accm[j][r];
                    const int e = (4 * (lane >> 4) + r) * 16 + (lane & 15);
#pragma unroll
                    for (int h = 0; h < kMel2Slots; ++h)
                        if (mask & (1 << h)) vm += xpart[(kMel2Slots + h) * 256 + e];
                    const int m = 16 * blk + 4 * (lane >> 4) + r;
                    if (m < n_filters && t < T) {
                        if (layout == ZAFX_LAYOUT_FT) out[((long long)clip * rows + m) * TP + t] = vm;
                        else out[((long long)clip * T + t) * rows + m] = vm;
                    }
                }
            }
            if constexpr (MFCC) {
                const float eps = 2.220446049250313e-16f;   // np.finfo(float).eps (zaf.py:445)
#pragma unroll
                for (int r = 0; r < 4; ++r) val[r] = 16 * blk + 4 * (lane >> 4) + r < n_filters ? logf(val[r] + eps) : 0.f;
                // register r of 16-lane row q holds filter 4 q + r of the block; the B fragment of step s wants filter 4 s + q there
                lane_row_transpose4(val[0], val[1], val[2], val[3]);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    if (16 * c >= n_coefs) continue;   // (uniform)
                    f32x4v d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int st = 0; st < 4; ++st) d = __builtin_amdgcn_mfma_f32_16x16x4f32(dfr[j][4 * c + st], val[st], d, 0, 0, 0);
                    float* dst = exf + (2 * j + c) * 256 + (4 * (lane >> 4)) * 16 + (lane & 15);
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[r * 16] = d[r];
                }
            } else {
                const int t = t0 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 16 * blk + 4 * (lane >> 4) + r;
                    if (m < n_filters && t < T) {
                        if (layout == ZAFX_LAYOUT_FT) out[((long long)clip * n_filters + m) * TP + t] = val[r];
                        else out[((long long)clip * T + t) * n_filters + m] = val[r];
                    }
                }
            }
        }
        if constexpr (!MFCC && ZAFX_MEL2_EARLYOUT) lds_barrier();   // the next tile's levels are in LDS (the owners' stores are on their way)
        if constexpr (MFCC) {
            lds_barrier();   // every block's partial coefficient tile is in LDS
            if (tid < 16 * n_coefs) {
                const int q = tid >> 4, t = t0 + (tid & 15);
                float part[8];
#pragma unroll
                for (int bb = 0; bb < 8; ++bb) {
                    const int ow = owners[bb];
                    part[bb] = bb < n_blocks ? reinterpret_cast<const float*>(frames + (ow & 255) * PITCH + EXOFF)[(2 * (ow >> 8) + (q >> 4)) * 256 + (q & 15) * 16 + (tid & 15)] : 0.f;
                }
                float sum = 0.f;
#pragma unroll
                for (int bb = 0; bb < 8; ++bb) sum += part[bb];   // (block order: deterministic)
                if (t < T) {
                    const int rows = DUAL ? n_filters + n_coefs : n_coefs, row = DUAL ? n_filters + q : q;
                    if (layout == ZAFX_LAYOUT_FT) out[((long long)clip * rows + row) * TP + t] = sum;
                    else out[((long long)clip * T + t) * rows + row] = sum;
                }
            }
            lds_barrier();   // the exchange areas are free for the next transforms
        }
    };
    for (; tlv + (int)gridDim.x < total_tiles; tlv += gridDim.x) tile_body(std::true_type{});
    tile_body(std::false_type{});
}

template <int LOG2N, bool ALIGNED>
static hipError_t run_mel(const zafx_plan& pl, const float* x, float* out, int64_t n_clips, int64_t n_samples, int T) {
    constexpr int LOG2E = mel_log2e(LOG2N);
    using G = MelCfg<LOG2N, LOG2E>;
    const int mfcc = pl.kind == ZAFX_MFCC;
    if constexpr (ZAFX_MEL2 && LOG2N == 10 && LOG2E == 4 && kMelFpb == 16 && kMelThreads == 1024) {
        if (pl.fb.whole_ok && pl.fb.n_waves == 16 && (!mfcc || pl.dct.dct2_ok)) {   // k_mel2: the product of a tile under the transforms of the next
            using C = FftCfg<10, 4>;
            const bool dual = mfcc && pl.prm.with_mel;
            const size_t smem = (size_t)(16 * (dual ? kMel2DualPitch : C::PITCH) + C::TW + C::N + C::N / 2 + 1) * 8 + (dual ? 2 : 1) * kMel2Slots * 1024;
            static_assert((size_t)(16 * C::PITCH + C::TW + C::N + C::N / 2 + 1) * 8 + kMel2Slots * 1024 <= (size_t)kMaxLdsBytes, "k_mel2: LDS");
            static_assert((size_t)(16 * kMel2DualPitch + C::TW + C::N + C::N / 2 + 1) * 8 + 2 * kMel2Slots * 1024 <= (size_t)kMaxLdsBytes, "k_mel2, one-pass mel + mfcc: LDS");
            const int pcm = take_pcm_mode();   // (zafx_execute_pcm: the input is int16, mono or stereo; pcm_direct_ok vouches for the alignment ALIGNED stands for)
            auto k2 = pcm == 1 ? (mfcc ? k_mel2<ALIGNED, 1, 1> : k_mel2<ALIGNED, 0, 1>) : pcm == 2 ? (mfcc ? k_mel2<ALIGNED, 1, 2> : k_mel2<ALIGNED, 0, 2>)
                                                                                                   : (mfcc ? k_mel2<ALIGNED, 1> : k_mel2<ALIGNED, 0>);
            if (dual) k2 = pcm == 1 ? k_mel2<ALIGNED, 4, 1> : pcm == 2 ? k_mel2<ALIGNED, 4, 2> : k_mel2<ALIGNED, 4>;
            if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(k2), pl.device, smem); e != hipSuccess) return e;
            const int tiles2 = (T + 15) / 16;
            const long long total2 = (long long)tiles2 * n_clips;
            if (total2 <= 0) return hipSuccess;
            const long long grid2 = std::min<long long>(total2, (long long)pl.n_cus);
            pl.ran = "k_mel2";
            hipLaunchKernelGGL(k2, dim3((unsigned)grid2), dim3(1024), smem, pl.stream, x, pl.d_window, pl.d_tw_pass, pl.d_tw_aux, pl.fb.d_pack, pl.fb.d_whole, pl.dct.d_dct2,
                               pl.dct.d_owner2, out, (long long)n_samples, pl.H, T, (int)row_pitch(pl, T), tiles2, (int)total2, pl.prm.n_filters, mfcc ? pl.prm.n_coefs : 0, pl.layout);
            return hipGetLastError();
        }
    }
    if (mfcc && pl.prm.with_mel) {
        set_error("ZAFX_MFCC with_mel: the one-pass mel + mfcc kernel takes window_length 2048, up to 128 filters and up to 32 coefficients (run the two plans instead)");
        return hipErrorInvalidValue;
    }
    // filterbank and DCT fragments resident in registers when the busiest wave's K-steps fit (128 filters at W = 2048: 17 + 4)
    // (a block without non-zeros has no K-step to carry its zero tile: streamed form)
    constexpr bool team8 = kMelFpb == 8 && G::NT == 512;   // (8-frame form: twice the steps per wave, re-read every tile)
    const bool res = (G::NT == 1024 || team8) && pl.fb.max_wave_steps <= (team8 ? 2 : 1) * kMelResidentFb && pl.fb.n_empty == 0 && pl.fb.desc_ok &&
                     (!mfcc || (pl.dct.max_wave_steps <= (team8 ? 2 : 1) * kMelResidentDct && pl.dct.n_empty == 0 && pl.dct.desc_ok));
    auto kern = k_mel<LOG2N, LOG2E, ALIGNED, false, 0>;
    if constexpr (G::NT == 1024 || team8) {
        if (res) kern = mfcc ? (pl.dct.direct_j > 0 ? k_mel<LOG2N, LOG2E, ALIGNED, true, 3> : k_mel<LOG2N, LOG2E, ALIGNED, true, 2>) : k_mel<LOG2N, LOG2E, ALIGNED, true, 1>;
    }
    const int direct_j = (res && mfcc) ? pl.dct.direct_j : 0;   // (register-fed DCT: one partial tile per wave and 16-row block)
    const int slots = pl.fb.n_items + (mfcc ? (direct_j > 0 ? pl.dct.n_blocks * (G::NT / 64) : pl.fb.n_blocks + pl.dct.n_items) : 0);
    if (slots > G::CAPACITY) {
        set_error("mel/mfcc: too many filterbank work items for the LDS slots at this window_length");
        return hipErrorInvalidValue;
    }
    if (pl.fb.n_waves != G::NT / 64 || (mfcc && pl.dct.n_waves != G::NT / 64)) {
        set_error("mel/mfcc: filterbank was packed for a different workgroup size");
        return hipErrorInvalidValue;
    }
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, G::SMEM); e != hipSuccess) return e;
    const int tiles = (T + G::FPB - 1) / G::FPB;
    const long long total = (long long)tiles * n_clips;
    if (total <= 0) return hipSuccess;
    const int per_cu = (int)std::min<size_t>(2, (size_t)kMaxLdsBytes / G::SMEM);
    const long long grid = std::min<long long>(total, (long long)pl.n_cus * std::max(per_cu, 1));
    pl.ran = "k_mel";
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(G::NT), G::SMEM, pl.stream, x, pl.d_window, LOG2E == 5 ? pl.d_tw_r32 : pl.d_tw_pass, pl.d_tw_aux, pl.fb.d_pack,
                       pl.fb.d_items, pl.fb.d_wave_ptr, pl.fb.d_blk_ptr, pl.fb.d_desc, pl.fb.n_blocks, pl.fb.n_items, pl.fb.total_steps, mfcc ? pl.dct.total_steps : 0, pl.dct.d_pack, pl.dct.d_items,
                       pl.dct.d_wave_ptr, pl.dct.d_blk_ptr, pl.dct.d_desc, pl.dct.d_direct, direct_j, pl.dct.n_blocks, out, (long long)n_samples, pl.H, T, (int)row_pitch(pl, T), tiles, (int)total,
                       pl.prm.n_filters, pl.prm.n_coefs, mfcc, pl.layout);
    return hipGetLastError();
}

// ZAFX_SPECTRUM_MAGNITUDE / _POWER of an STFT plan at W = 2048 in the reference layout on k_mel2 (MODE 2 / 3); false: not this geometry
#ifndef ZAFX_SPEC2
#define ZAFX_SPEC2 1
#endif
bool launch_spec2(const zafx_plan& pl, const float* x, float* out, int64_t n_clips, int64_t n_samples, int T, hipError_t& err) {
    using C = FftCfg<10, 4>;
    if (!ZAFX_SPEC2 || pl.log2nf != 10 || pl.log2e != 4 || pl.layout != ZAFX_LAYOUT_FT || kMelFpb != 16 || kMelThreads != 1024 || !pl.d_tw_pass || !pl.d_tw_aux) return false;
    if (pl.prm.spectrum != ZAFX_SPECTRUM_MAGNITUDE && pl.prm.spectrum != ZAFX_SPECTRUM_POWER) return false;
    if (n_samples >= (1LL << 29)) return false;
    const int pcm = take_pcm_mode();
    const bool aligned = (n_samples % 2 == 0) && (pl.H % 2 == 0) && (reinterpret_cast<uintptr_t>(x) % 8 == 0);
    const bool power = pl.prm.spectrum == ZAFX_SPECTRUM_POWER;
    auto k2 = aligned ? (power ? k_mel2<true, 3> : k_mel2<true, 2>) : (power ? k_mel2<false, 3> : k_mel2<false, 2>);
    if (pcm == 1) k2 = aligned ? (power ? k_mel2<true, 3, 1> : k_mel2<true, 2, 1>) : (power ? k_mel2<false, 3, 1> : k_mel2<false, 2, 1>);
    if (pcm == 2) k2 = aligned ? (power ? k_mel2<true, 3, 2> : k_mel2<true, 2, 2>) : (power ? k_mel2<false, 3, 2> : k_mel2<false, 2, 2>);
    const size_t smem = (size_t)(16 * C::PITCH + C::TW + C::N + C::N / 2 + 1) * 8 + kMel2Slots * 1024;
    err = ensure_dynamic_lds(reinterpret_cast<const void*>(k2), pl.device, smem);
    if (err != hipSuccess) return true;
    const int tiles = (T + 15) / 16;
    const long long total = (long long)tiles * n_clips;
    if (total <= 0 || total >= (1LL << 31)) return total <= 0;
    const long long grid = std::min<long long>(total, (long long)pl.n_cus);
    pl.ran = "k_mel2";
    hipLaunchKernelGGL(k2, dim3((unsigned)grid), dim3(1024), smem, pl.stream, x, pl.d_window, pl.d_tw_pass, pl.d_tw_aux, (const float*)nullptr, (const int4*)nullptr,
                       (const float*)nullptr, (const int*)nullptr, out, (long long)n_samples, pl.H, T, (int)row_pitch(pl, T), tiles, (int)total, 0, 0, pl.layout);
    err = hipGetLastError();
    return true;
}

// Does a call with int16 PCM (n_channels = 1 or 2) of this plan run on k_mel2, which takes the integers in its loads?  (Everything else converts
// into a float32 staging array first: zafx_execute_pcm.)
bool pcm_direct_ok(const zafx_plan& pl, int64_t n_frames, int n_channels, int sample_bytes) {
    if (sample_bytes != 2 || (n_channels != 1 && n_channels != 2) || pl.prm.precision != ZAFX_PRECISION_F32 || pl.bs_log2m > 0) return false;
    if (pl.log2nf != 10 || pl.log2e != 4 || kMelFpb != 16 || kMelThreads != 1024 || n_frames >= (1LL << 29)) return false;
    if (pl.kind == ZAFX_MEL || pl.kind == ZAFX_MFCC)
        return ZAFX_MEL2 && !mel_takes_wide_route(pl) && pl.fb.whole_ok && pl.fb.n_waves == 16 && (pl.kind == ZAFX_MEL || pl.dct.dct2_ok);
    if (pl.kind == ZAFX_STFT)
        return ZAFX_SPEC2 && pl.layout == ZAFX_LAYOUT_FT && (pl.prm.spectrum == ZAFX_SPECTRUM_MAGNITUDE || pl.prm.spectrum == ZAFX_SPECTRUM_POWER);
    return false;
}

const char* mel_kernel_name() { return "k_mel"; }
int mel_waves(int log2n) { return mel_threads(log2n, mel_log2e(log2n)) / 64; }

template <int LOG2N>
static hipError_t run_mel_any(const zafx_plan& pl, const float* x, float* out, int64_t n_clips, int64_t n_samples, int T) {
    // (the aligned form addresses a clip through one buffer descriptor with 32-bit byte offsets: clips below 2^29 samples)
    const bool aligned = (n_samples % 2 == 0) && (pl.H % 2 == 0) && (reinterpret_cast<uintptr_t>(x) % 8 == 0) && n_samples < (1LL << 29);
    return aligned ? run_mel<LOG2N, true>(pl, x, out, n_clips, n_samples, T) : run_mel<LOG2N, false>(pl, x, out, n_clips, n_samples, T);
}

// ---------------------------------------------------------------------------------
// windows of 4096 / 8192 samples: spectrum kernel + banded filterbank kernel (k_melfb)
// ---------------------------------------------------------------------------------
// Sixteen frames of such a window do not fit LDS, so the fused kernel stops at W = 2048 and these sizes ran on the float64
// kernel (one workgroup per frame, float64 arrays in and out).  Here the |X| (mel) or |X|^2 (mfcc) rows 0..W/2 of a CHUNK of
// clips are written by the STFT kernels' magnitude / power kinds (W = 4096: k_stft_ft16b) into a plan-owned scratch with rows
// padded to whole 128-byte lines, and k_melfb multiplies them with the filterbank: a 16-wave workgroup per (clip, 64 frames),
// a lane per frame, wave w the filters w, w + 16, ...; a filter's row is its band of non-zeros [first, first + count)
// (zaf.py:305-316: one triangle), so a wave reads count coalesced 256-byte rows of the spectrum per filter and each spectrum
// row is read by the two filters that overlap it -- the second time from the vector cache or L2.  mfcc: the workgroup's mel
// values go through LDS as log(mel + eps) and every wave forms its DCT-II rows from there (zaf.py:443-452).  Chunk: 256 MB of
// scratch (ZAFX_MELW_CHUNK_MB), rounded to whole rounds of the persistent spectrum kernel -- measured, 1024 clips x 10 s at
// W = 4096: 2.44 / 2.01 / 1.93 / 1.93 ms with 64 / 128 / 256 / 1024 MB (keeping the scratch inside the memory-side cache buys
// nothing; whole rounds and fewer launches do).
__global__ __launch_bounds__(1024) void k_melfb(const float* __restrict__ spec, const float* __restrict__ fb_vals, const int* __restrict__ fb_meta,
                                                const float* __restrict__ dct, float* __restrict__ out, int n_filters, int n_coefs, int rows,
                                                int SP, int T, int TP, int layout) {
    extern __shared__ float logmel[];   // [n_filters][64] (mfcc only)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + lane;
    const long long clip = blockIdx.y;
    const bool live = t < T;
    const float* sp = spec + (clip * rows + 1) * SP + (live ? t : 0);   // bin c + 1 <-> filterbank column c (zaf.py:370)
    const float eps = 2.220446049250313e-16f;   // np.finfo(float).eps (zaf.py:445)
    for (int f = wave; f < n_filters; f += 16) {
        const int first = __builtin_amdgcn_readfirstlane(fb_meta[3 * f]), count = __builtin_amdgcn_readfirstlane(fb_meta[3 * f + 1]);
        const float* v = fb_vals + __builtin_amdgcn_readfirstlane(fb_meta[3 * f + 2]);
        const float* s = sp + (long long)first * SP;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int j = 0;
        for (; j + 4 <= count; j += 4) {
            a0 += v[j] * s[(long long)j * SP];
            a1 += v[j + 1] * s[(long long)(j + 1) * SP];
            a2 += v[j + 2] * s[(long long)(j + 2) * SP];
            a3 += v[j + 3] * s[(long long)(j + 3) * SP];
        }
        for (; j < count; ++j) a0 += v[j] * s[(long long)j * SP];
        const float mel = (a0 + a1) + (a2 + a3);
        if (n_coefs > 0) {
            logmel[f * 64 + lane] = logf(mel + eps);
        } else if (live) {
            if (layout == ZAFX_LAYOUT_FT) out[(clip * n_filters + f) * TP + t] = mel;
            else out[(clip * T + t) * n_filters + f] = mel;
        }
    }
    if (n_coefs > 0) {
        __syncthreads();
        for (int c = wave; c < n_coefs; c += 16) {
            const float* d = dct + (long long)c * n_filters;
            float a0 = 0.f, a1 = 0.f;
            int f = 0;
            for (; f + 2 <= n_filters; f += 2) {
                a0 += d[f] * logmel[f * 64 + lane];
                a1 += d[f + 1] * logmel[(f + 1) * 64 + lane];
            }
            if (f < n_filters) a0 += d[f] * logmel[f * 64 + lane];
            if (live) {
                if (layout == ZAFX_LAYOUT_FT) out[(clip * n_coefs + c) * TP + t] = a0 + a1;
                else out[(clip * T + t) * n_coefs + c] = a0 + a1;
            }
        }
    }
}

static size_t melw_chunk_bytes() {
    static const size_t bytes = [] {
        const char* env = std::getenv("ZAFX_MELW_CHUNK_MB");
        const long mb = env ? std::atol(env) : 0;
        return (size_t)(mb > 0 ? mb : 256) << 20;
    }();
    return bytes;
}

#ifndef ZAFX_MEL_BAND
#define ZAFX_MEL_BAND 1
#endif
static hipError_t run_mel_wide(zafx_plan& pl, const float* x, float* out, int64_t n_clips, int64_t n_samples, int T) {
    if ((long long)n_clips * T <= 0) return hipSuccess;
    if (ZAFX_MEL_BAND && mel_band_usable(pl, x, n_clips, n_samples, T)) return launch_mel_band(pl, x, out, n_clips, n_samples, T);   // W = 4096: fused (k_mel_ft16b)
    const int mfcc = pl.kind == ZAFX_MFCC;
    const int rows = pl.W / 2 + 1;
    const int SP = (T + 31) / 32 * 32;   // spectrum rows as whole 128-byte lines
    const size_t per_clip = (size_t)rows * SP * sizeof(float);
    int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(n_clips, (int64_t)(melw_chunk_bytes() / per_clip)));
    chunk = std::min<int64_t>(chunk, 65535);   // k_melfb puts the clips of a chunk in grid.y
    {   // whole rounds of the persistent spectrum kernel: (16-frame tiles per clip) x chunk a multiple of the workgroups
        const int tiles = (T + 15) / 16;
        int g = tiles, h = pl.n_cus;
        while (h) { const int r = g % h; g = h; h = r; }
        const int64_t m = pl.n_cus / g;
        if (chunk >= m) chunk = chunk / m * m;
    }
    if (hipError_t e = grow_scratch(pl, (size_t)chunk * per_clip); e != hipSuccess) return e;
    float* spec = reinterpret_cast<float*>(pl.d_scratch64);
    // the spectrum kernels see an STFT plan of this window: one-sided |X| (mel, zaf.py:370) or |X|^2 (mfcc, zaf.py:437-439),
    // reference layout, rows padded to 32 floats
    zafx_plan st;
    st.device = pl.device; st.n_cus = pl.n_cus; st.kind = ZAFX_STFT; st.prm = pl.prm; st.stream = pl.stream;
    st.prm.spectrum = mfcc ? ZAFX_SPECTRUM_POWER : ZAFX_SPECTRUM_MAGNITUDE;
    st.prm.row_align = 32;
    st.W = pl.W; st.H = pl.H; st.layout = ZAFX_LAYOUT_FT; st.log2nf = pl.log2nf; st.log2e = pl.log2e;
    st.d_window = pl.d_window; st.d_tw_pass = pl.d_tw_pass; st.d_tw_aux = pl.d_tw_aux; st.d_tw_sub = pl.d_tw_sub; st.d_tw_r32 = pl.d_tw_r32; st.d_tw_quad = pl.d_tw_quad;
    st.bs_log2m = pl.bs_log2m; st.d_bs_chirp = pl.d_bs_chirp; st.d_bs_bhat = pl.d_bs_bhat;   // (a window that is not a power of two: the Bluestein STFT)
    const size_t smem = mfcc ? (size_t)pl.prm.n_filters * 64 * sizeof(float) : 0;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(k_melfb), pl.device, std::max<size_t>(smem, 1)); e != hipSuccess) return e;
    const int64_t out_per_clip = pl.layout == ZAFX_LAYOUT_FT ? (int64_t)(mfcc ? pl.prm.n_coefs : pl.prm.n_filters) * row_pitch(pl, T)
                                                             : (int64_t)T * (mfcc ? pl.prm.n_coefs : pl.prm.n_filters);
    for (int64_t c0 = 0; c0 < n_clips; c0 += chunk) {
        const int64_t n = std::min(chunk, n_clips - c0);
        if (hipError_t e = st.bs_log2m > 0 ? launch_stft_bs32(st, x + c0 * n_samples, reinterpret_cast<float2*>(spec), n, n_samples, T)
                                           : launch_stft(st, x + c0 * n_samples, reinterpret_cast<float2*>(spec), n, n_samples, T);
            e != hipSuccess)
            return e;
        pl.ran = "k_melfb";
        hipLaunchKernelGGL(k_melfb, dim3((unsigned)((T + 63) / 64), (unsigned)n), dim3(1024), smem, pl.stream, spec, pl.d_fbw, pl.d_fbw_meta, pl.d_dctw,
                           out + c0 * out_per_clip, pl.prm.n_filters, mfcc ? pl.prm.n_coefs : 0, rows, SP, T, (int)row_pitch(pl, T), pl.layout);
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    }
    return hipSuccess;
}

const char* mel_wide_kernel_name() { return "k_melfb"; }

hipError_t launch_mel(zafx_plan& pl, const float* x, float* out, int64_t n_clips, int64_t n_samples, int T) {
    if (mel_takes_wide_route(pl)) return run_mel_wide(pl, x, out, n_clips, n_samples, T);
    switch (pl.log2nf) {
        case 5: return run_mel_any<5>(pl, x, out, n_clips, n_samples, T);
        case 6: return run_mel_any<6>(pl, x, out, n_clips, n_samples, T);
        case 7: return run_mel_any<7>(pl, x, out, n_clips, n_samples, T);
        case 8: return run_mel_any<8>(pl, x, out, n_clips, n_samples, T);
        case 9: return run_mel_any<9>(pl, x, out, n_clips, n_samples, T);
        case 10: return run_mel_any<10>(pl, x, out, n_clips, n_samples, T);
    }
    set_error("mel/mfcc: unsupported window_length");
    return hipErrorInvalidValue;
}

}  // namespace zafx

ZAFX_PROF_EXPORT(zafx_debug_prof_mel, g_prof_mel)
