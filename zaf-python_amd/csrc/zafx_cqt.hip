// zafx_cqt.hip -- constant-Q spectrogram / chromagram kernel for gfx950 (MI355X).
//
// Replaces the per-frame loop of zaf.py:627-633:
//     cqt[:, j] = abs(cqt_kernel * np.fft.fft(xpad[j*step : j*step + fft_len]))
// with cqt_kernel a sparse (n_bins x fft_len) complex CSR matrix (zaf.py:554-557).
//
// Persistent workgroups, one per CU, each transforming one frame at a time: the whole frame (fft_len real = N =
// fft_len/2 complex points, up to 128 KiB) lives in LDS, owned by N/16 threads (1024 for fft_len 32768).
// Frames overlap by (fft_len - step)/fft_len (94.6 % at config Q), and the work is dealt out so that the overlap is
// served by the L2 of ONE chiplet: the workgroups of an XCD (block b runs on XCD b % 8) share a list of clips and
// take its frames round-robin -- at any time the 32 CUs of an XCD read ~32 neighbouring frames (0.35 MB of samples
// against 4 MB of L2), so a sample crosses the fabric once and the other reads hit L2.  (Tiles of 16 consecutive
// frames per workgroup put 32 x 237 KB of live samples on each XCD: every re-read missed L2, 8.3 x the input over the
// fabric and 4 ... 6 k cycles per frame of blocked load issue; profiles/r02_notes.md.)
// After the real split the CSR rows are contracted against the one-sided spectrum in LDS.  Real matrices whose columns lie in the
// lower half (the reference's kernels) take the MATRIX-CORE form (MM, build_cqt_mm in zafx_capi.cpp): rows in pairs, a pair's
// columns in segments of S entries, two segments per 4-lane block of v_mfma_f32_4x4x1_16b_f32, whose accumulation over S
// instructions IS the sum over a row's columns -- no lane reductions; a finishing pass adds a row's segments.  Everything else
// (complex matrices, conjugated bins, S above the registers, the 65536 double form) takes the lane-reduction form:
//   * the host sorts the rows by length and deals them out in "steps": a wavefront works on four short rows at once
//     (one per 16-lane DPP row, lane j of a row taking its entries j, j + 16, j + 32, ...), on two medium rows (32 lanes
//     each) or on one long row (64 lanes); a step ends with ONE pair of DPP reductions (row sums, + row_bcast for the
//     wider shapes) for all its rows, and the last lane of a row's lane group writes |.|^2 straight into the LDS
//     column -- no partial sums in LDS, no finishing pass;
//   * a wave's share of the matrix (value + LDS byte address of the spectrum bin per entry) stays in REGISTERS for the
//     whole launch when it fits (<= 12 entries per lane: config Q has 9 450 non-zeros, 12 iterations on
//     the busiest wave); larger matrices (the reference's own cqtkernel example, 60 879 non-zeros) stream it from L2
//     every frame;
//   * a numerically real matrix (the reference's kernels are: max |imag| / max |real| = 1e-16) is contracted as
//     real x complex.
// A frame's |.|^2 column is staged in LDS and stored by the first n_bins threads: 4-byte stores along the row pitch in
// the reference layout, which the XCD's L2 merges with the neighbouring frames written by its other CUs.
// The chromagram (zaf.py:693-698) is a strided row sum over that LDS column.
#include <algorithm>

#include "zafx_fft.hpp"
#include "zafx_internal.hpp"

namespace zafx {

ZAFX_PROF_ARRAY(g_prof_cqt)

// LDS carve shared by the kernel and the launcher (bytes before the wave / step tables, 16-B aligned:
// a misaligned ds_read_b128 is replayed at 64 cycles)
// DOUBLE: fft_length 65536 as the even and the odd bins of two 16384-point transforms (zafx_internal.hpp, cqt_double): the
// kernel is the LOG2N = 14 one, its root tables are those of twice the size.
template <int LOG2N, int LOG2E, bool DOUBLE = false>
struct CqtCfg {
    using C = FftCfg<LOG2N, LOG2E>;
    static constexpr int LT = LOG2N + (DOUBLE ? 1 : 0);            // log2 of the packed transform the tables are for
    static constexpr int NHI = LT > 7 ? 1 << (LT - 7) : 1;         // two-level roots of N (zafx_fft.hpp)
    static constexpr int NH2 = LT > 8 ? 1 << (LT - 8) : 1;         // two-level roots of 2N for k < N/2 (real split)
    static constexpr bool SPLIT = cqt_split(LOG2N);                // 16 x 1024 decomposition (zafx_internal.hpp)
    static constexpr int SLOTS = cqt_slots(LOG2N);                 // complex slots of the spectrum image
    static constexpr int NSUB = SPLIT ? twiddle_total(10, 4) : 0;  // pass tables of the 1024-point sub-transforms (8 KB: round 3; a two-level root table + product trees before)
    static constexpr size_t HEAD = (((size_t)(SLOTS + NHI + 128 + NH2 + 128 + NSUB) * 8 + 15) / 16) * 16;
};

// DPP row_bcast adds (gfx9 wave64 reductions): lane 15 of rows 0, 2 into every lane of rows 1, 3; lane 31 into rows 2, 3
__device__ __forceinline__ float bcast15_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));
}
__device__ __forceinline__ float bcast31_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false));
}

// REALK: kernel values are float (real matrix), else float2.  RES > 0: a wave's entries (RES iterations) ride in registers.
// MM > 0: the contraction runs on the matrix cores (below), MM = steps a wave keeps in registers (REALK, RES = 0).
template <int LOG2N, int LOG2E, bool ALIGNED, bool REALK, int RES, bool DOUBLE = false, int MM = 0>
__global__ __launch_bounds__(fft_threads(LOG2N, LOG2E)) void k_cqt(
    const float* __restrict__ x, const float2* __restrict__ twp, const float2* __restrict__ tws,
    const int4* __restrict__ wave_tab, const int* __restrict__ addrs, const float* __restrict__ values, const int* __restrict__ mm_fin, int mm_steps, int mm_segs,
    float* __restrict__ out, long long n_samples, int step, int left_pad, int T, int TP, int n_clips, int n_groups, int n_bins, int chroma_res,
    int layout, int k_lo, int k_hi, int k_special, int n_entries, int prune3) {
    using C = FftCfg<LOG2N, LOG2E>;
    using G = CqtCfg<LOG2N, LOG2E, DOUBLE>;
    using KV = std::conditional_t<REALK, float, float2>;
    constexpr int N = C::N, P = C::P, E = C::E, W = (DOUBLE ? 4 : 2) * N, NHI = G::NHI, NH2 = G::NH2;   // W = samples of a frame
    static_assert(!DOUBLE || (G::SPLIT && LOG2N == 14), "the double form stands on the 16 x 1024 transform");
    constexpr bool RESIDENT = RES > 0;
    static_assert(P >= 64, "CQT frames are owned by whole wavefronts");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* buf = reinterpret_cast<float2*>(smem_raw);            // G::SLOTS slots: bin k at slot_of(k), X[N] at NYQ
    float2* tw_hi = buf + G::SLOTS;                               // two-level root table of N
    float2* tw_lo = tw_hi + NHI;
    float2* sp_hi = tw_lo + 128;                                  // two-level root table of 2N (split twiddles)
    float2* sp_lo = sp_hi + NH2;
    float2* sub_hi = sp_lo + 128;                                 // (SPLIT) pass tables of the 1024-point sub-transforms
    auto slot_of = [](int k) { return cqt_slot(LOG2N, k); };   // (DOUBLE: of a POSITION of the 16384-point transform)
    constexpr int NYQ = cqt_nyquist_slot(LOG2N);
    int4* wave_l = reinterpret_cast<int4*>(smem_raw + G::HEAD);   // [P / 64] {first iteration, iterations, step-end mask of the resident form, 0}
    float* mags = reinterpret_cast<float*>(wave_l + P / 64);      // [n_bins] |.|^2 of the current frame
    float2* part = reinterpret_cast<float2*>(mags + ((n_bins + 1) & ~1));   // (MM) [P]: every lane's two partial sums
    int* fin_l = reinterpret_cast<int*>(part + P + 1);                        // (MM) [pairs]: first stream slot | segments << 16  (part[P] = 0: what the finishing pass reads past a row's last segment)
    static_assert(MM == 0 || (REALK && RES == 0 && !DOUBLE), "matrix-core contraction: real matrix, no other resident form");
    const int p = threadIdx.x;
    for (int i = p; i < NHI + 128; i += P) tw_hi[i] = twp[i];
    if constexpr (G::SPLIT)
        for (int i = p; i < G::NSUB; i += P) sub_hi[i] = twp[NHI + 128 + i];
    for (int i = p; i < NH2 + 128; i += P) sp_hi[i] = tws[i];
    for (int i = p; i < P / 64; i += P) wave_l[i] = wave_tab[i];
    if constexpr (MM > 0) {
        for (int i = p; i < (n_bins + 1) / 2; i += P) fin_l[i] = mm_fin[i];
        if (p == 0) part[P] = make_float2(0.f, 0.f);
    }
    lds_barrier();
    const TwoLevelTw tw2l{tw_hi, tw_lo};
    const int wave = p >> 6;
    // work list of this workgroup: group = blockIdx % n_groups (the XCD when n_groups = 8) owns clips group, group +
    // n_groups, ...; its frames, clip after clip, are dealt round-robin to the group's workgroups
    const int group = blockIdx.x % n_groups, slot = blockIdx.x / n_groups;
    const int n_slots = (gridDim.x - group + n_groups - 1) / n_groups;
    const long long n_work = (long long)((n_clips - group + n_groups - 1) / n_groups) * T;
    // (Round 4 also dealt the list out in units of four consecutive frames, so that a finished row left as one 16-byte piece per
    // unit: the XCD's workgroups then spread over 128 frames instead of 32, the L2 held neither the samples nor the partial lines --
    // 10.5 GB fetched and 1.37 GB written per launch against 5.42 and 0.446, 1.4 % slower.  Not kept.)
    // wave-uniform values read from LDS land in VGPRs; readfirstlane tells the compiler they are scalars
    // (scalar branches and SGPR operands instead of exec-mask juggling)
    const int4 wt = wave_l[wave];
    const int it0 = __builtin_amdgcn_readfirstlane(wt.x), n_it = __builtin_amdgcn_readfirstlane(wt.y);
    const int endmask = __builtin_amdgcn_readfirstlane(wt.z);
    const auto raddr = make_rsrc(addrs, (unsigned)n_entries * 4u);
    const auto rvals = make_rsrc(values, (unsigned)n_entries * (unsigned)sizeof(KV));
    auto load_kv = [&](int voff_entries) -> KV {
        if constexpr (REALK) return buf_load_f32(rvals, voff_entries * 4);
        else return buf_load_f32x2(rvals, voff_entries * 8);
    };

    // ---- a wave's share of the kernel matrix, resident in registers for the whole launch
    KV kv[RESIDENT ? RES : MM > 0 ? MM : 1];
    int ad[RESIDENT ? RES : MM > 0 ? MM : 1];
    if constexpr (MM > 0) {   // (the tables hold mm_steps <= MM steps per wave)
#pragma unroll
        for (int i = 0; i < MM; ++i) {
            const int e = i < mm_steps ? ((p >> 6) * mm_steps + i) * 64 + (p & 63) : -1;   // out of range: reads 0
            ad[i] = buf_load_i32(raddr, e * 4);
            kv[i] = load_kv(e);
        }
    }
    if constexpr (RESIDENT) {
#pragma unroll
        for (int i = 0; i < RES; ++i) {
            const int e = i < n_it ? (it0 + i) * 64 + (p & 63) : -1;   // out of range: reads 0
            ad[i] = buf_load_i32(raddr, e * 4);
            kv[i] = load_kv(e);
        }
    }

    // ---- framing, no window (it lives in the kernel): zaf.py:612-620, :631.  Frames inside the clip take
    // unconditional 8-byte loads; the zero-padded edge frames take the predicated path.
    float2 v[E];
    const unsigned clip_bytes = (unsigned)std::min<long long>(n_samples * 4, 0xfffffffcLL);
    // load_frame() returns true when v holds the raw 16-byte loads of the split form (unpack_pairs() before the first pass).
    // With `deferred` the eight loads of that form are not issued: the caller requests them one at a time (load_one) between
    // the iterations of its contraction.
    __amdgpu_buffer_rsrc_t frx = make_rsrc(x, 0);
    int fvoff = 0;
    auto load_one = [&](int i) {   // raw: (v[2i], v[2i+1]) = the lane's two points of load i; unpack_pairs() sorts them out
        const float4 q = buf_load_f32x4(frx, fvoff, i * P * 8);
        v[2 * i] = make_float2(q.x, q.y);
        v[2 * i + 1] = make_float2(q.z, q.w);
    };
    // DOUBLE: the first half of the frame only (points p + 1024 i as 8-byte loads); load_second() fetches the second half where it
    // is needed and forms z[n] + z[n + 16384] (even bins) or z[n] - z[n + 16384] (odd bins)
    bool d_fast = false;
    auto load_frame = [&](long long g, int p, bool deferred = false) -> bool {   // g: index into the group's frame list; p: thread id (an opaque copy inside the frame loop)
        const int clip = group + (int)((unsigned)g / (unsigned)T) * n_groups, t = (int)((unsigned)g % (unsigned)T);   // (g < 2^31: zafx_execute)
        const auto rx = make_rsrc(x + (long long)clip * n_samples, clip_bytes);
        const long long s0 = (long long)t * step - left_pad;
        if constexpr (DOUBLE) {
            frx = rx;
            d_fast = ALIGNED && s0 >= 0 && s0 + W <= n_samples;
            if (d_fast) {
                fvoff = ((int)s0 + 2 * p) * 4;
#pragma unroll
                for (int i = 0; i < E; ++i) v[i] = buf_load_f32x2(frx, fvoff, i * P * 8);
            } else {
                fvoff = (int)s0;   // (sample index; the predicated loads of load_second() do the whole frame)
            }
            return false;
        }
        if (ALIGNED && s0 >= 0 && s0 + W <= n_samples) {
            if constexpr (G::SPLIT) {
                // 16-byte lanes: a CU pulls a 128-KB frame out of L2 in 3.5 k cycles with them, 5.8 k with 8-byte lanes
                // (tools/exp_cqtload.hip).  The lane pair (2q, 2q + 1) shares its loads: the even lane fetches the points
                // (2q, 2q + 1) + 1024 r for r = 0 .. 7, the odd lane for r = 8 .. 15; each keeps its own point, sends the
                // other's across (DPP swap inside the pair).  The odd lane thus holds its 16 points rotated by 8, i.e. its
                // radix-16 outputs carry (-1)^k -- absorbed by negating its base twiddle in first_pass.
                const int odd = p & 1;
                frx = rx;
                fvoff = ((int)s0 + 2 * (p - odd)) * 4 + odd * (8 * P * 8);
                if (!deferred) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) load_one(i);
                }
                return true;
            } else {
                const int voff = ((int)s0 + 2 * p) * 4;
#pragma unroll
                for (int i = 0; i < E; ++i) v[i] = buf_load_f32x2(rx, voff, i * P * 8);
            }
        } else {
            const int rot = G::SPLIT ? 8 * (p & 1) : 0;   // (split form: odd lanes hold their points rotated by 8, as above)
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const long long s = s0 + 2 * (p + ((i + rot) & (E - 1)) * P);
                v[i].x = (s >= 0 && s < n_samples) ? buf_load_f32(rx, (int)s * 4) : 0.f;
                v[i].y = (s + 1 >= 0 && s + 1 < n_samples) ? buf_load_f32(rx, (int)(s + 1) * 4) : 0.f;
            }
        }
        return false;
    };
    // the pair exchange of the 16-byte loads (see load_frame), done when the data is needed -- not where it was requested
    auto unpack_pairs = [&](int p) {
        if constexpr (G::SPLIT) {
            const int odd = p & 1;
            float2 lo[8], hi[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                lo[i] = v[2 * i];
                hi[i] = v[2 * i + 1];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float2 keep = odd ? hi[i] : lo[i], send = odd ? lo[i] : hi[i];
                v[i] = keep;
                v[i + 8].x = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send.x), 0xB1, 0xf, 0xf, true));
                v[i + 8].y = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send.y), 0xB1, 0xf, 0xf, true));
            }
        }
    };
    // First pass of the 16 x 1024 form, in registers: radix-16 across the workgroup on the samples 1024 apart (thread p
    // holds n2 = p), times w^(p k1).  It runs BEFORE the barrier that frees the LDS frame, i.e. under the tail of the
    // previous frame's contraction.
    // DOUBLE: second half of the frame and the sum (odd = 0: even bins) or difference (odd = 1) of the halves; clip_n = samples of the clip
    auto load_second = [&](int p, int odd) {
        if (d_fast) {
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const float2 q = buf_load_f32x2(frx, fvoff + 2 * N * 4, i * P * 8);   // point p + 1024 i of the second half: 2 N samples on
                v[i] = odd ? make_float2(v[i].x - q.x, v[i].y - q.y) : make_float2(v[i].x + q.x, v[i].y + q.y);
            }
        } else {   // a frame that touches the clip's edges (zero padding, zaf.py:612-620): both halves, sample by sample
            const long long s0 = fvoff;
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const long long s = s0 + 2 * (p + i * P), s2 = s + 2 * N;
                auto at = [&](long long q) { return (q >= 0 && q < n_samples) ? buf_load_f32(frx, (int)q * 4) : 0.f; };
                const float2 a = make_float2(at(s), at(s + 1)), b = make_float2(at(s2), at(s2 + 1));
                v[i] = odd ? make_float2(a.x - b.x, a.y - b.y) : make_float2(a.x + b.x, a.y + b.y);
            }
        }
    };
    auto first_pass = [&](int p, int odd = 0) {
        if constexpr (DOUBLE) {
            // even bins: the 16384-point transform of the sums; odd bins: of the differences times w^n, w = exp(-2 pi i / 32768),
            // n = p + 1024 r -- the factor exp(-2 pi i r / 32) goes on the inputs (constants), w^p on the outputs with the pass's own
            // twiddles: output k1 of the radix-16 pass gets w^(p (2 k1 + odd)).
            const float2 b1 = tw2(tw2l, p);          // w^p (tw2l: the roots of 32768 here)
            float2 w[16];
            w[1] = cmul(b1, b1);                      // the root of 16384 to the p
#pragma unroll
            for (int r = 2; r < 16; ++r) w[r] = cmul(w[r >> 1], w[r - (r >> 1)]);
            if (odd) {
                static constexpr float kC[16][2] = {{1.000000000f, -0.000000000f}, {0.980785280f, -0.195090322f}, {0.923879533f, -0.382683432f}, {0.831469612f, -0.555570233f}, {0.707106781f, -0.707106781f}, {0.555570233f, -0.831469612f}, {0.382683432f, -0.923879533f}, {0.195090322f, -0.980785280f}, {0.000000000f, -1.000000000f}, {-0.195090322f, -0.980785280f}, {-0.382683432f, -0.923879533f}, {-0.555570233f, -0.831469612f}, {-0.707106781f, -0.707106781f}, {-0.831469612f, -0.555570233f}, {-0.923879533f, -0.382683432f}, {-0.980785280f, -0.195090322f}};   // exp(-2 pi i r / 32)
#pragma unroll
                for (int r = 1; r < 16; ++r) v[r] = cmul(v[r], make_float2(kC[r][0], kC[r][1]));
            }
            Dft<16>::run(v);
            if (odd) {
                v[0] = cmul(v[0], b1);
#pragma unroll
                for (int r = 1; r < 16; ++r) v[r] = cmul(v[r], cmul(w[r], b1));
            } else {
#pragma unroll
                for (int r = 1; r < 16; ++r) v[r] = cmul(v[r], w[r]);
            }
        } else if constexpr (G::SPLIT) {
            static_assert(!G::SPLIT || (E == 16 && P == 1024), "split form: 16 points per thread, 16 wavefronts");
            Dft<16>::run(v);
            float2 w[16];
            w[1] = tw2(tw2l, p);
            if (p & 1) w[1] = make_float2(-w[1].x, -w[1].y);   // odd lanes hold their points rotated by 8: (-1)^k on output k (load_frame)
#pragma unroll
            for (int r = 2; r < 16; ++r) w[r] = cmul(w[r >> 1], w[r - (r >> 1)]);
#pragma unroll
            for (int r = 1; r < 16; ++r) v[r] = cmul(v[r], w[r]);
        }
    };
#ifndef ZAFX_CQT_EARLY_SCATTER
#define ZAFX_CQT_EARLY_SCATTER 0   // (measured with the matrix-core form: 24.94 against 24.81 ms -- the finishing pass's reads then queue behind sixteen waves' writes)
#endif
    // 16384 = 16 x 1024: output k1 of the first pass goes to sub-sequence k1 at position p (the frame image must be free: behind
    // the barrier that ends the previous frame's contraction)
    auto scatter_first_pass = [&](int p) {
        if constexpr (G::SPLIT) {
            const int pp = phys(p);
#pragma unroll
            for (int r = 0; r < 16; ++r) buf[r * kCqtRegion + pp] = v[r];
        }
    };
    if (slot < n_work) {
        if (load_frame(slot, threadIdx.x)) unpack_pairs(threadIdx.x);
        if constexpr (DOUBLE) load_second(threadIdx.x, 0);
        first_pass(threadIdx.x);
        if constexpr (ZAFX_CQT_EARLY_SCATTER) scatter_first_pass(threadIdx.x);
    }
    PROF_INIT(g_prof_cqt);
    int odd = 0;              // DOUBLE: which half of the bins the transform in flight yields (every frame runs the loop body twice)
    float2 ev[DOUBLE ? 4 : 1];   // DOUBLE: the thread's even bins, waiting for the odd ones

#pragma unroll 1
    for (long long g = slot; g < n_work; g += (DOUBLE && odd == 1) ? 0 : n_slots) {   // (DOUBLE: `odd` has been flipped at the end of the body)
        PROF_MARK(0);
        // Opaque copy of the thread id: everything below recomputes its (cheap) per-lane LDS and buffer
        // offsets every frame.  Left to itself the compiler hoists ~50 of them out of the loop and
        // spills them; every scratch reload then drains vmcnt and with it the prefetches in flight.
        int p = threadIdx.x;
        asm volatile("" : "+v"(p));
        const int lane = p & 63;
        if constexpr (G::SPLIT) {
            // 16384 = 16 x 1024: output k1 of the first pass goes to sub-sequence k1 at position p.  Then wave w
            // transforms sub-sequence w on its own (three wave-local passes, no workgroup barrier):
            // X[k1 + 16 k2] = FFT_1024(sub-sequence k1)[k2].
            // (the outputs of the first pass went to LDS behind the previous frame's last barrier -- scatter_first_pass --, under the
            // latency of its finishing pass and column store)
            if constexpr (!ZAFX_CQT_EARLY_SCATTER) scatter_first_pass(p);
            lds_barrier();
            float2* sub = buf + (p >> 6) * kCqtRegion;
            regs_read<10, 4>(v, sub, lane);
            frame_sync<64>();
            // (the last pass leaves only the positions the split and the contraction read: run_cqt's mask)
#ifndef ZAFX_CQT_LDS_EXCH
#define ZAFX_CQT_LDS_EXCH 0
#endif
            if constexpr (ZAFX_CQT_LDS_EXCH) {   // experiment: both exchanges of the sub-transform through LDS (80 VALU instructions fewer, 32 LDS ones more)
                const float2* t = (const float2*)sub_hi;
                pass_write<10, 4, 0, 4>(v, sub, lane, t);
                frame_sync<64>();
                regs_read<10, 4>(v, sub, lane);
                frame_sync<64>();
                pass_write<10, 4, 4, 4>(v, sub, lane, t + twiddle_offset(10, 4, 4));
                frame_sync<64>();
                regs_read<10, 4>(v, sub, lane);
                frame_sync<64>();
                switch (prune3) {
                    case 0: pass3_write_pruned<0>(v, sub, lane, t); break;
                    case 1: pass3_write_pruned<1>(v, sub, lane, t); break;
                    case 2: pass3_write_pruned<2>(v, sub, lane, t); break;
                    default: pass3_write(v, sub, lane, t);
                }
                frame_sync<64>();
            } else
            fft1024_wave(v, sub, lane, (const float2*)sub_hi, 0, prune3);   // (twiddles from the pass tables: 54 VALU instructions per frame and wave fewer than with the product trees)
        } else {
            fft_frame_chain<LOG2N, LOG2E>(v, buf, p, tw2l);
        }
        const long long g_next = (DOUBLE && odd == 0) ? g : g + n_slots;   // frame of the transform after this one
        const bool more = g_next < n_work;
        bool raw = false;
#ifndef ZAFX_CQT_EARLY
#define ZAFX_CQT_EARLY 1
#endif
        // EARLY (split form): every wave requests the next frame as soon as ITS sub-transform is done (v is dead from here to the
        // next first pass), ahead of the barrier: the waves finish a few thousand cycles apart, the requests arrive spread out and
        // fly under the split and the contraction of all sixteen waves.
        constexpr bool EARLY = ZAFX_CQT_EARLY && G::SPLIT;
        if constexpr (EARLY) {
            if (more) raw = load_frame(g_next, p);
        }
        if constexpr (G::SPLIT) lds_barrier();
        PROF_MARK(1);
        // ---- real split in place, only for the pairs (k, N-k) that the kernel's columns touch:
        // slots 0..N-1 <- X[0..N-1], slot NYQ <- X[N];  t_k = exp(-2 pi i k / 2N) = sp_hi[k >> 7] sp_lo[k & 127]
        if constexpr (DOUBLE) {
            // position q of this transform is bin k = 2 q + odd of the 32768-point spectrum Z; its partner N - k sits at position
            // 16384 - q (even) / 16383 - q (odd) of the SAME transform.  Even bins wait in registers; with the odd ones they are
            // laid out as cqt_slot(15, k): odd bin at the slot of position q, even bin at the slot of position 8192 + q.
            const int q_lo = k_lo >> 1, q_hi = k_hi >> 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = q_lo + p + i * P;
                if (q <= q_hi && (odd || q >= 1)) {
                    const int k = 2 * q + odd;
                    float2 xk, xn;
                    split_pair(buf[cqt_slot14(q)], buf[cqt_slot14(16384 - odd - q)], cmul(sp_hi[k >> 7], sp_lo[k & 127]), xk, xn);
                    if (odd) {
                        buf[cqt_slot14(q)] = xk;
                        buf[cqt_slot14(8192 + q)] = ev[i];
                    } else {
                        ev[i] = xk;
                    }
                }
            }
        } else {
        if (k_special && p == 0) {
            const float2 z0 = buf[0], zc = buf[slot_of(N / 2)];
            buf[0] = make_float2(z0.x + z0.y, 0.f);
            buf[NYQ] = make_float2(z0.x - z0.y, 0.f);
            buf[slot_of(N / 2)] = cconj(zc);
        }
        for (int k = k_lo + p; k <= k_hi; k += P) {
            float2 xk, xn;
            split_pair(buf[slot_of(k)], buf[slot_of(N - k)], cmul(sp_hi[k >> 7], sp_lo[k & 127]), xk, xn);
            buf[slot_of(k)] = xk;
            buf[slot_of(N - k)] = xn;
        }
        }
        PROF_MARK(2);
        lds_barrier();
        PROF_MARK(3);
        // The next frame's samples: v is dead until the next first pass.  Requested AFTER the split: 16 wavefronts x 16
        // loads overrun the CU's vector-memory queue, and a wave that blocks there ahead of its share of the split holds
        // the whole workgroup at the barrier (profiles/r02_notes.md; requested ahead of the split: 30.8 instead of 29.6 ms).
        // Some of the waves request first and contract second, the others the other way round: 16 waves x 8 loads at once
        // overrun the CU's vector-memory queue, and the waves served last (the youngest) started their contraction 3 k cycles late.
#ifndef ZAFX_CQT_LOADS_FIRST
#define ZAFX_CQT_LOADS_FIRST 8
#endif
        const bool loads_first = wave < ZAFX_CQT_LOADS_FIRST * (P / 64) / 16;
        if (!EARLY && more && loads_first) raw = load_frame(g_next, p);
        PROF_MARK(7);
        // ---- CSR mat-vec against the spectrum + |.|^2 (zaf.py:630-632).  Entry (iteration, lane) of the wave's share is a
        // value and a word: bits 0-17 the LDS byte address of its spectrum bin, bit 31 "conjugate" (a column of the upper
        // half), bits 29-30 (the same in all lanes) 0 or the shape 1 / 2 / 3 = 16 / 32 / 64 lanes per row of a step that ENDS
        // with this iteration, bits 18-28 the row this lane then writes (0x7ff: none).  Padding entries are 0 * bin 0.
        if constexpr (MM > 0) {
            // ---- the same product on the matrix cores (build_cqt_mm, zafx_capi.cpp): step i multiplies, in every 4-lane block, the
            // lanes' A values (rows 2 pa + {0, 1} of stream a, 2 pb + {0, 1} of stream b at the step's column) with their B values
            // (re, im of stream a's bin, re, im of stream b's) and accumulates: register r of lane (blk, j) = sum over the steps of
            // A(blk, r) B(blk, j).  Lanes 0, 1 of a block keep registers 0, 1 (stream a: re / im of its two rows), lanes 2, 3
            // registers 2, 3; the cross terms are dropped.  No lane reductions, no selects: ~5 vector instructions per frame and wave
            // against ~175 of the lane-reduction form below.
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            int ni = mm_steps;
            asm volatile("" : "+s"(ni));
            float bq[MM];
#pragma unroll
            for (int i = 0; i < MM; ++i) {
                int a = ad[i];
                asm volatile("" : "+v"(a));
                bq[i] = i < ni ? *reinterpret_cast<const float*>(smem_raw + a) : 0.f;
            }
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < MM; ++i)
                if (i < ni) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(kv[i], bq[i], acc, 0, 0, 0);
            part[p] = (lane & 2) ? make_float2(acc[2], acc[3]) : make_float2(acc[0], acc[1]);
        } else
        if (!DOUBLE || odd) {   // (DOUBLE: the even half has no spectrum to contract yet)
            float ar = 0.f, ai = 0.f;
            auto mac = [&](KV k, int a) {
                float2 xv = *reinterpret_cast<const float2*>(smem_raw + (a & 0x3ffff));
                xv.y = __builtin_bit_cast(float, __builtin_bit_cast(int, xv.y) ^ (a & (int)0x80000000));
                if constexpr (REALK) {
                    ar = fmaf(k, xv.x, ar);
                    ai = fmaf(k, xv.y, ai);
                } else {
                    ar += k.x * xv.x - k.y * xv.y;
                    ai += k.x * xv.y + k.y * xv.x;
                }
            };
            auto finish = [&](int a, int shape) {   // end of a step: one reduction pair serves all its rows
                ar = row16_sum(ar);
                ai = row16_sum(ai);
                if (shape >= 2) {
                    ar = bcast15_add(ar);
                    ai = bcast15_add(ai);
                }
                if (shape == 3) {
                    ar = bcast31_add(ar);
                    ai = bcast31_add(ai);
                }
                const int row = (a >> 18) & 0x7ff;
                if (row != 0x7ff) mags[row] = ar * ar + ai * ai;   // (the square root is taken when the column is stored)
                ar = 0.f;
                ai = 0.f;
            };
            if constexpr (RESIDENT) {
                // Scalar conditions on values that are re-read every frame: hoisted out of the frame loop the RES
                // comparisons cost 2 SGPRs each and spill, and the address masks two more VGPRs per entry.
                int em = endmask, ni = n_it;
                asm volatile("" : "+s"(em), "+s"(ni));
#pragma unroll
                for (int i = 0; i < RES; ++i) {
                    if (i < ni) {
                        int a = ad[i];
                        asm volatile("" : "+v"(a));
                        mac(kv[i], a);
                        if (em & (1 << i)) finish(a, (__builtin_amdgcn_readfirstlane(a) >> 29) & 3);
                    }
                }
            } else {
                constexpr int GR = 8;   // iterations requested together
                for (int b = 0; b < n_it; b += GR) {
                    KV kq[GR];
                    int aq[GR];
#pragma unroll
                    for (int g = 0; g < GR; ++g) {
                        const int e = b + g < n_it ? (it0 + b + g) * 64 + lane : -1;
                        aq[g] = buf_load_i32(raddr, e * 4);
                        kq[g] = load_kv(e);
                    }
#pragma unroll
                    for (int g = 0; g < GR; ++g) {
                        if (b + g < n_it) {
                            mac(kq[g], aq[g]);
                            const int shape = (__builtin_amdgcn_readfirstlane(aq[g]) >> 29) & 3;
                            if (shape) finish(aq[g], shape);
                        }
                    }
                }
            }
        }
        if (!EARLY && more && !loads_first) raw = load_frame(g_next, p);
        PROF_MARK(4);
        if (more) {   // (waits for the prefetched samples; the other waves are still contracting)
            if (raw) unpack_pairs(p);
            if constexpr (DOUBLE) load_second(p, odd ^ 1);
            first_pass(p, DOUBLE ? odd ^ 1 : 0);
        }
        PROF_MARK(5);
        lds_barrier();
        PROF_MARK(6);
        if (ZAFX_CQT_EARLY_SCATTER && more) scatter_first_pass(p);
        if constexpr (DOUBLE) {
            odd ^= 1;
            if (odd == 1) continue;   // the even half is done: no column yet
        }
        // ---- store the frame's column (the next write of `mags` is three barriers away)
        {
            const int clip = group + (int)((unsigned)g / (unsigned)T) * n_groups, t = (int)((unsigned)g % (unsigned)T);   // (g < 2^31: zafx_execute)
            if constexpr (MM > 0) {
                // finishing pass of the matrix-core form: L lanes per row (8, or fewer when the rows then do not fit ONE pass of the
                // workgroup: two passes kept waves 0 and 1, alone with a second one, between everybody and the next barrier) add up the
                // row's segments (float 4 s + 2 c + m of `part`: stream slot s, c = re / im, m = the row's place in its pair) -- eight
                // reads per lane requested together, lanes past the row's last segment read the zero slot behind `part` --, DPP adds, |.|, store
                const float* pf = reinterpret_cast<const float*>(part);
                const int lg = n_bins * 8 <= P ? 3 : n_bins * 4 <= P ? 2 : n_bins * 2 <= P ? 1 : 0;   // (uniform)
                const int rows_per_pass = P >> lg;
                for (int r0 = 0; r0 + ((p & ~63) >> lg) < n_bins; r0 += rows_per_pass) {   // (wave-uniform trip count)
                    const int r = r0 + (p >> lg), q = p & ((1 << lg) - 1);
                    const int w = r < n_bins ? fin_l[r >> 1] : 0, s0 = w & 0xffff, ns = w >> 16;
                    const int base = 4 * (s0 + q) + (r & 1);
                    float re = 0.f, im = 0.f;
                    for (int i0 = 0; __builtin_amdgcn_ballot_w64(q + i0 < ns) != 0; i0 += 8 << lg) {
                        float a[8], b[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int i = i0 + (u << lg);
                            const bool in = q + i < ns;
                            const int at = in ? base + 4 * i : 2 * P;
                            a[u] = pf[at];
                            b[u] = pf[in ? at + 2 : at];
                        }
                        re += ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
                        im += ((b[0] + b[1]) + (b[2] + b[3])) + ((b[4] + b[5]) + (b[6] + b[7]));
                    }
                    auto dpp_add = [](float v, auto ctrl) {
                        return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xf, 0xf, true));
                    };
                    if (lg >= 1) {   // (uniform)
                        re = dpp_add(re, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
                        im = dpp_add(im, std::integral_constant<int, 0xB1>{});
                    }
                    if (lg >= 2) {
                        re = dpp_add(re, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
                        im = dpp_add(im, std::integral_constant<int, 0x4E>{});
                    }
                    if (lg >= 3) {
                        re = dpp_add(re, std::integral_constant<int, 0x141>{});   // row_half_mirror
                        im = dpp_add(im, std::integral_constant<int, 0x141>{});
                    }
                    if (q == 0 && r < n_bins) {
                        if (chroma_res > 0) {
                            mags[r] = re * re + im * im;
                        } else {
                            const float val = __builtin_amdgcn_sqrtf(re * re + im * im);
                            if (layout == ZAFX_LAYOUT_FT) out[((long long)clip * n_bins + r) * TP + t] = val;
                            else out[((long long)clip * T + t) * n_bins + r] = val;
                        }
                    }
                }
                if (chroma_res > 0) lds_barrier();   // (uniform) the chroma sums below read every row
            }
            if (chroma_res > 0) {
                for (int ch = p; ch < chroma_res; ch += P) {
                    float acc = 0.f;
                    for (int r = ch; r < n_bins; r += chroma_res) acc += __builtin_amdgcn_sqrtf(mags[r]);   // zaf.py:696-698
                    if (layout == ZAFX_LAYOUT_FT) out[((long long)clip * chroma_res + ch) * TP + t] = acc;   // TP = row pitch (>= T)
                    else out[((long long)clip * T + t) * chroma_res + ch] = acc;
                }
            } else if constexpr (MM == 0) {
                for (int r = p; r < n_bins; r += P) {
                    const float val = __builtin_amdgcn_sqrtf(mags[r]);
                    if (layout == ZAFX_LAYOUT_FT) out[((long long)clip * n_bins + r) * TP + t] = val;
                    else out[((long long)clip * T + t) * n_bins + r] = val;
                }
            }
        }
    }
}

// LDS bytes of k_cqt
// (LOG2NP = log2 of the packed frame: 15 = fft_length 65536, which runs the LOG2N = 14 kernel in its double form)
template <int LOG2NP>
static size_t cqt_lds(int n_bins) {
    constexpr bool DOUBLE = cqt_double(LOG2NP);
    constexpr int LOG2N = DOUBLE ? 14 : LOG2NP;
    constexpr int LOG2E = default_log2e(LOG2N);
    using C = FftCfg<LOG2N, LOG2E>;
    using G = CqtCfg<LOG2N, LOG2E, DOUBLE>;
    return G::HEAD + (size_t)(C::P / 64) * 16 + (size_t)n_bins * sizeof(float);
}

template <int LOG2NP>
static hipError_t run_cqt(const zafx_plan& pl, const float* x, float* out, int64_t n_clips, int64_t n_samples, int T) {
    constexpr bool DOUBLE = cqt_double(LOG2NP);
    constexpr int LOG2N = DOUBLE ? 14 : LOG2NP;
    constexpr int LOG2E = default_log2e(LOG2N);
    using C = FftCfg<LOG2N, LOG2E>;
    const int diff = pl.W - pl.H;                              // may be negative if step > fft_len
    const int left = diff >= 0 ? (diff + 1) / 2 : -((-diff) / 2);   // ceil(diff / 2)  (zaf.py:615)
    const bool aligned = n_samples % 2 == 0 && pl.H % 2 == 0 && left % 2 == 0 && reinterpret_cast<uintptr_t>(x) % 8 == 0;
    // a real matrix whose busiest wave has <= kCqtResident iterations keeps its entries in registers (16 would spill at 1024 threads)
    const bool realk = pl.cqt_real;
    const bool res = realk && pl.cqt_resident == kCqtResident && !DOUBLE;   // (the double form has no registers left for them: 156 bytes of scratch)
#ifndef ZAFX_CQT_MM
#define ZAFX_CQT_MM 1
#endif
    // matrix-core contraction: its partial sums (8 bytes per thread) and the pairs' segment table sit behind the column in LDS
    const size_t smem_mm = cqt_lds<LOG2NP>((pl.prm.n_bins + 1) & ~1) + (size_t)(C::P + 1) * 8 + (size_t)((pl.prm.n_bins + 1) / 2) * 4;
    const bool mm = ZAFX_CQT_MM && !DOUBLE && realk && pl.cqt_mm_steps > 0 && pl.cqt_mm_steps <= kCqtMmSteps && smem_mm <= (size_t)kMaxLdsBytes;
    auto pick = [&](auto al) {
        constexpr bool AL = decltype(al)::value;
        if constexpr (!DOUBLE) {
            if (mm) return k_cqt<LOG2N, LOG2E, AL, true, 0, false, kCqtMmSteps>;
        }
        return !realk ? k_cqt<LOG2N, LOG2E, AL, false, 0, DOUBLE> : res ? k_cqt<LOG2N, LOG2E, AL, true, kCqtResident, DOUBLE> : k_cqt<LOG2N, LOG2E, AL, true, 0, DOUBLE>;
    };
    auto kern = aligned ? pick(std::true_type{}) : pick(std::false_type{});
    const size_t smem = mm ? smem_mm : cqt_lds<LOG2NP>(pl.prm.n_bins);
    if (smem > (size_t)kMaxLdsBytes) {
        set_error("cqt: kernel matrix has too many rows for LDS at this fft_length");
        return hipErrorInvalidValue;
    }
    if (n_samples >= (1LL << 29)) {   // a clip is addressed through one buffer descriptor with 32-bit byte offsets
        set_error("cqt: clips of 2^29 samples or more are not supported (3.4 h at 44.1 kHz); cut the signal");
        return hipErrorInvalidValue;
    }
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, smem); e != hipSuccess) return e;
    if (n_clips * (long long)T <= 0) return hipSuccess;
    // one persistent workgroup per CU (it needs nearly all of the CU's LDS), in n_groups groups that each share a list of clips;
    // with >= 8 clips a group is an XCD (block b runs on XCD b % 8), fewer clips are shared by several XCDs
    const int n_groups = (int)std::min<int64_t>(8, n_clips);
    const long long per_group = ((n_clips + n_groups - 1) / n_groups) * (long long)T;   // frames of the longest list
    const int grid = (int)std::min<long long>(std::max(pl.n_cus / n_groups, 1) * (long long)n_groups, per_group * n_groups);
    // Bin k lives in sub-transform k & 15 at position k >> 4.  The split reads the pairs (k, N - k) of k_lo .. k_hi: positions up to
    // k_hi >> 4 and their mirrors from 1024 - (k_hi >> 4); when they all lie in the lowest / highest 64 (HB + 1) positions, HB <= 2,
    // the last pass of the sub-transforms forms only those (pass3_write_pruned).  Columns 0, N / 2, N need position 512: full pass.
    int prune3 = -1;
#ifndef ZAFX_CQT_PRUNE
#define ZAFX_CQT_PRUNE 1
#endif
    const int pos_hi = (DOUBLE ? pl.cqt_k_hi >> 1 : pl.cqt_k_hi) >> 4;   // (double form: bin k of the spectrum is position k >> 1 of its transform)
    if (ZAFX_CQT_PRUNE && cqt_split(LOG2N) && !pl.cqt_k_special && pl.cqt_k_hi >= pl.cqt_k_lo && (pos_hi >> 6) <= 2)
        prune3 = pos_hi >> 6;
    pl.ran = "k_cqt";
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(C::P), smem, pl.stream, x, pl.d_tw_pass, pl.d_tw_aux, pl.d_cqt_waves,
                       mm ? pl.d_cqt_mm_addr : pl.d_cqt_addrs, mm ? pl.d_cqt_mm_vals : pl.d_cqt_vals, pl.d_cqt_mm_fin, mm ? pl.cqt_mm_steps : 0, pl.cqt_mm_segs, out, (long long)n_samples, pl.H, left, T, (int)row_pitch(pl, T), (int)n_clips, n_groups,
                       pl.prm.n_bins, pl.kind == ZAFX_CHROMA ? pl.prm.octave_resolution : 0, pl.layout, pl.cqt_k_lo, pl.cqt_k_hi,
                       pl.cqt_k_special, mm ? (C::P / 64) * pl.cqt_mm_steps * 64 : std::max(pl.cqt_n_entries, 1), prune3);
    return hipGetLastError();
}

bool cqt_supported(int log2n) { return log2n >= 8 && log2n <= 15; }
int cqt_waves(int log2n) {
    if (cqt_double(log2n)) log2n = 14;   // (the double form runs the 16384-point kernel)
    return fft_threads(log2n, default_log2e(log2n)) / 64;
}
const char* cqt_kernel_name() { return "k_cqt"; }

// Largest number of rows a float32 plan of this fft_length can hold, capped by the 11-bit row field of the entry words
int cqt_max_bins(int log2n) {
    auto fit = [](auto tag) {
        constexpr int L = decltype(tag)::value;
        int lo = 0, hi = 1 << 20;
        while (lo < hi) {
            const int mid = (lo + hi + 1) / 2;
            if (cqt_lds<L>(mid) <= (size_t)kMaxLdsBytes) lo = mid;
            else hi = mid - 1;
        }
        return std::min(lo, 0x7fe);
    };
    switch (log2n) {
        case 8: return fit(std::integral_constant<int, 8>{});
        case 9: return fit(std::integral_constant<int, 9>{});
        case 10: return fit(std::integral_constant<int, 10>{});
        case 11: return fit(std::integral_constant<int, 11>{});
        case 12: return fit(std::integral_constant<int, 12>{});
        case 13: return fit(std::integral_constant<int, 13>{});
        case 14: return fit(std::integral_constant<int, 14>{});
        case 15: return fit(std::integral_constant<int, 15>{});
    }
    return 0;
}

hipError_t launch_cqt(const zafx_plan& pl, const float* x, float* out, int64_t n_clips, int64_t n_samples, int T) {
    switch (pl.log2nf) {
        case 8: return run_cqt<8>(pl, x, out, n_clips, n_samples, T);
        case 9: return run_cqt<9>(pl, x, out, n_clips, n_samples, T);
        case 10: return run_cqt<10>(pl, x, out, n_clips, n_samples, T);
        case 11: return run_cqt<11>(pl, x, out, n_clips, n_samples, T);
        case 12: return run_cqt<12>(pl, x, out, n_clips, n_samples, T);
        case 13: return run_cqt<13>(pl, x, out, n_clips, n_samples, T);
        case 14: return run_cqt<14>(pl, x, out, n_clips, n_samples, T);
        case 15: return run_cqt<15>(pl, x, out, n_clips, n_samples, T);
    }
    set_error("cqt: unsupported fft_length");
    return hipErrorInvalidValue;
}

}  // namespace zafx

ZAFX_PROF_EXPORT(zafx_debug_prof_cqt, g_prof_cqt)
