// zafx_cqt.hip -- constant-Q spectrogram / chromagram kernel for gfx950 (MI355X).
//
// Replaces the per-frame loop of zaf.py:627-633:
//     cqt[:, j] = abs(cqt_kernel * np.fft.fft(xpad[j*step : j*step + fft_len]))
// with cqt_kernel a sparse (n_bins x fft_len) complex CSR matrix (zaf.py:554-557).
//
// One workgroup owns FW (16, fewer when the kernel matrix has many rows) consecutive frames of one clip and
// transforms them one after the other: the whole frame (fft_len real = N = fft_len/2 complex points, up to
// 128 KiB) lives in LDS, owned by N/16 threads (1024 for fft_len 32768).  Frames of a tile overlap by
// (fft_len - step)/fft_len (94.6 % at config Q), so the re-reads of the input hit L2; HBM sees each sample about
// once per tile.  After the real split the CSR rows are contracted against the one-sided spectrum in LDS:
//   * the host sorts the rows by length and deals them out in "steps": a wavefront works on four short rows at once
//     (one per 16-lane DPP row, lane j of a row taking its entries j, j + 16, j + 32, ...), on two medium rows (32 lanes
//     each) or on one long row (64 lanes); a step ends with ONE pair of DPP reductions (row sums, + row_bcast for the
//     wider shapes) for all its rows, and the last lane of a row's lane group writes |.|^2 straight into the LDS
//     tile -- no partial sums in LDS, no finishing pass;
//   * a wave's share of the matrix (value + LDS byte address of the spectrum bin per entry) stays in REGISTERS across
//     the frames of the tile when it fits (<= 12 entries per lane: config Q has 9 450 non-zeros, 12 iterations on
//     the busiest wave); larger matrices (the reference's own cqtkernel example, 60 879 non-zeros) stream it from L2
//     every frame;
//   * a numerically real matrix (the reference's kernels are: max |imag| / max |real| = 1e-16) is contracted as
//     real x complex.
// The magnitudes of the tile's frames are staged in LDS so that the (n_bins, T) store writes 64-B runs along t.
// The chromagram (zaf.py:693-698) is a strided row sum over that LDS tile.
#include <algorithm>

#include "zafx_fft.hpp"
#include "zafx_internal.hpp"

namespace zafx {

ZAFX_PROF_ARRAY(g_prof_cqt)

// LDS carve shared by the kernel and the launcher (bytes before the wave / step tables, 16-B aligned:
// a misaligned ds_read_b128 is replayed at 64 cycles)
template <int LOG2N, int LOG2E>
struct CqtCfg {
    using C = FftCfg<LOG2N, LOG2E>;
    static constexpr int NHI = LOG2N > 7 ? 1 << (LOG2N - 7) : 1;   // two-level roots of N (zafx_fft.hpp)
    static constexpr int NH2 = LOG2N > 8 ? 1 << (LOG2N - 8) : 1;   // two-level roots of 2N for k < N/2 (real split)
    static constexpr bool SPLIT = cqt_split(LOG2N);                // 16 x 1024 decomposition (zafx_internal.hpp)
    static constexpr int SLOTS = cqt_slots(LOG2N);                 // complex slots of the spectrum image
    static constexpr int NSUB = SPLIT ? 8 + 128 : 0;               // two-level roots of the 1024-point sub-transforms
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
template <int LOG2N, int LOG2E, bool ALIGNED, bool REALK, int RES>
__global__ __launch_bounds__(fft_threads(LOG2N, LOG2E)) void k_cqt(
    const float* __restrict__ x, const float2* __restrict__ twp, const float2* __restrict__ tws,
    const int4* __restrict__ wave_tab, const int4* __restrict__ step_tab, const int* __restrict__ addrs, const float* __restrict__ values,
    float* __restrict__ out, long long n_samples, int step, int left_pad, int T, int TP, int tiles, int n_bins, int chroma_res, int layout,
    int n_steps, int FW, int k_lo, int k_hi, int k_special, int n_entries) {
    using C = FftCfg<LOG2N, LOG2E>;
    using G = CqtCfg<LOG2N, LOG2E>;
    using KV = std::conditional_t<REALK, float, float2>;
    constexpr int N = C::N, P = C::P, E = C::E, W = 2 * N, NHI = G::NHI, NH2 = G::NH2;
    constexpr bool RESIDENT = RES > 0;
    static_assert(P >= 64, "CQT frames are owned by whole wavefronts");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* buf = reinterpret_cast<float2*>(smem_raw);            // G::SLOTS slots: bin k at slot_of(k), X[N] at NYQ
    float2* tw_hi = buf + G::SLOTS;                               // two-level root table of N
    float2* tw_lo = tw_hi + NHI;
    float2* sp_hi = tw_lo + 128;                                  // two-level root table of 2N (split twiddles)
    float2* sp_lo = sp_hi + NH2;
    float2* sub_hi = sp_lo + 128;                                 // (SPLIT) two-level root table of 1024
    auto slot_of = [](int k) { return cqt_slot(LOG2N, k); };
    constexpr int NYQ = cqt_nyquist_slot(LOG2N);
    int4* wave_l = reinterpret_cast<int4*>(smem_raw + G::HEAD);   // [P / 64] {first iteration, iterations, first step, steps}
    int* mask_l = reinterpret_cast<int*>(wave_l + P / 64);        // [P / 64] bit i: iteration i of the wave ends a step (resident form)
    int4* step_l = reinterpret_cast<int4*>(mask_l + ((P / 64 + 3) & ~3));   // [n_steps] rows of the step's four DPP rows (-1: none) ...
    int* iters_l = reinterpret_cast<int*>(step_l + n_steps);      // [n_steps] ... and its iteration count | lanes per row (16, 32, 64) << 16
    float* tile = reinterpret_cast<float*>(iters_l + ((n_steps + 3) & ~3));   // [n_bins][FW]
    const int p = threadIdx.x;
    for (int i = p; i < NHI + 128; i += P) tw_hi[i] = twp[i];
    if constexpr (G::SPLIT)
        for (int i = p; i < G::NSUB; i += P) sub_hi[i] = twp[NHI + 128 + i];
    for (int i = p; i < NH2 + 128; i += P) sp_hi[i] = tws[i];
    for (int i = p; i < P / 64; i += P) {
        wave_l[i] = wave_tab[2 * i];
        mask_l[i] = wave_tab[2 * i + 1].x;
    }
    for (int i = p; i < n_steps; i += P) {
        const int4 s = step_tab[2 * i], m = step_tab[2 * i + 1];
        step_l[i] = s;
        iters_l[i] = m.x;
    }
    lds_barrier();
    const TwoLevelTw tw2l{tw_hi, tw_lo};
    const int wave = p >> 6;
    const int clip = blockIdx.x / tiles, tl = blockIdx.x % tiles;
    const int t0 = tl * FW;
    const float* xc = x + (long long)clip * n_samples;
    // wave-uniform values read from LDS land in VGPRs; readfirstlane tells the compiler they are scalars
    // (scalar branches and SGPR operands instead of exec-mask juggling)
    const int4 wt = wave_l[wave];
    const int it0 = __builtin_amdgcn_readfirstlane(wt.x), n_it = __builtin_amdgcn_readfirstlane(wt.y);
    const int s0 = __builtin_amdgcn_readfirstlane(wt.z), s1 = s0 + __builtin_amdgcn_readfirstlane(wt.w);
    const int endmask = __builtin_amdgcn_readfirstlane(mask_l[wave]);
    const auto raddr = make_rsrc(addrs, (unsigned)n_entries * 4u);
    const auto rvals = make_rsrc(values, (unsigned)n_entries * (unsigned)sizeof(KV));
    auto load_kv = [&](int voff_entries) -> KV {
        if constexpr (REALK) return buf_load_f32(rvals, voff_entries * 4);
        else return buf_load_f32x2(rvals, voff_entries * 8);
    };

    // ---- a wave's share of the kernel matrix, resident in registers for the whole tile
    KV kv[RESIDENT ? RES : 1];
    int ad[RESIDENT ? RES : 1];
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
    const auto rx = make_rsrc(xc, clip_bytes);
    auto load_frame = [&](int t, int p) {   // p: thread id (an opaque copy inside the frame loop)
        const long long s0 = (long long)t * step - left_pad;
        if (ALIGNED && s0 >= 0 && s0 + W <= n_samples) {
            const int voff = ((int)s0 + 2 * p) * 4;
#pragma unroll
            for (int i = 0; i < E; ++i) v[i] = buf_load_f32x2(rx, voff, i * P * 8);
        } else {
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const long long s = s0 + 2 * (p + i * P);
                v[i].x = (s >= 0 && s < n_samples) ? buf_load_f32(rx, (int)s * 4) : 0.f;
                v[i].y = (s + 1 >= 0 && s + 1 < n_samples) ? buf_load_f32(rx, (int)(s + 1) * 4) : 0.f;
            }
        }
    };
    // First pass of the 16 x 1024 form, in registers: radix-16 across the workgroup on the samples 1024 apart (thread p
    // holds n2 = p), times w^(p k1).  It runs BEFORE the barrier that frees the LDS frame, i.e. under the tail of the
    // previous frame's contraction.
    auto first_pass = [&](int p) {
        if constexpr (G::SPLIT) {
            static_assert(!G::SPLIT || (E == 16 && P == 1024), "split form: 16 points per thread, 16 wavefronts");
            Dft<16>::run(v);
            float2 w[16];
            w[1] = tw2(tw2l, p);
#pragma unroll
            for (int r = 2; r < 16; ++r) w[r] = cmul(w[r >> 1], w[r - (r >> 1)]);
#pragma unroll
            for (int r = 1; r < 16; ++r) v[r] = cmul(v[r], w[r]);
        }
    };
    if (t0 < T) {
        load_frame(t0, threadIdx.x);
        first_pass(threadIdx.x);
    }
    PROF_INIT(g_prof_cqt);

#pragma unroll 1
    for (int jj = 0; jj < FW; ++jj) {
        const int t = t0 + jj;
        if (t >= T) break;   // uniform across the block
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
            const int pp = phys(p);
#pragma unroll
            for (int r = 0; r < 16; ++r) buf[r * kCqtRegion + pp] = v[r];
            lds_barrier();
            float2* sub = buf + (p >> 6) * kCqtRegion;
            regs_read<10, 4>(v, sub, lane);
            frame_sync<64>();
            fft_frame_chain<10, 4>(v, sub, lane, TwoLevelTw{sub_hi, sub_hi + 8});
            lds_barrier();
        } else {
            fft_frame_chain<LOG2N, LOG2E>(v, buf, p, tw2l);
        }
        PROF_MARK(1);
        // the next frame's samples: v is dead until the next first pass, the loads fly under the split + contraction
        const bool more = jj + 1 < FW && t + 1 < T;
        if (more) load_frame(t + 1, p);
        // ---- real split in place, only for the pairs (k, N-k) that the kernel's columns touch:
        // slots 0..N-1 <- X[0..N-1], slot NYQ <- X[N];  t_k = exp(-2 pi i k / 2N) = sp_hi[k >> 7] sp_lo[k & 127]
        if (k_special && p == 0) {
            const float2 z0 = buf[0], zc = buf[slot_of(N / 2)];
            buf[0] = make_float2(z0.x + z0.y, 0.f);
            buf[NYQ] = make_float2(z0.x - z0.y, 0.f);
            buf[slot_of(N / 2)] = cconj(zc);
        }
        for (int k = k_lo + p; k <= k_hi; k += P) {
            const float2 zk = buf[slot_of(k)], zn = buf[slot_of(N - k)];
            const float2 tk = cmul(sp_hi[k >> 7], sp_lo[k & 127]);
            const float2 e = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
            const float2 d = make_float2(0.5f * (zk.x - zn.x), 0.5f * (zk.y + zn.y));
            const float2 to = cmul(tk, make_float2(d.y, -d.x));
            buf[slot_of(k)] = cadd(e, to);
            buf[slot_of(N - k)] = cconj(csub(e, to));
        }
        PROF_MARK(2);
        lds_barrier();
        PROF_MARK(3);
        // ---- CSR mat-vec against the spectrum + |.|^2 (zaf.py:630-632).  Lane (g, j) = (lane >> 4, lane & 15) of the
        // wave's current step works on row step_l[s][g]; entry (iteration it, lane) carries the value and the LDS byte
        // address of its spectrum bin (bit 31: conjugate, a column of the upper half); padding entries are 0 * bin 0.
        {
            float ar = 0.f, ai = 0.f;
            auto mac = [&](KV k, int a) {
                asm volatile("" : "+v"(a));   // (the masks below are frame-invariant: hoisted they would cost two more registers per resident entry)
                float2 xv = *reinterpret_cast<const float2*>(smem_raw + (a & 0x7fffffff));
                xv.y = __builtin_bit_cast(float, __builtin_bit_cast(int, xv.y) ^ (a & (int)0x80000000));
                if constexpr (REALK) {
                    ar = fmaf(k, xv.x, ar);
                    ai = fmaf(k, xv.y, ai);
                } else {
                    ar += k.x * xv.x - k.y * xv.y;
                    ai += k.x * xv.y + k.y * xv.x;
                }
            };
            auto finish = [&](int s) {   // end of a step: one reduction pair serves all its rows
                const int4 rows = step_l[s];
                const int shape = __builtin_amdgcn_readfirstlane(iters_l[s]) >> 16;
                ar = row16_sum(ar);
                ai = row16_sum(ai);
                if (shape >= 32) {
                    ar = bcast15_add(ar);
                    ai = bcast15_add(ai);
                }
                if (shape == 64) {
                    ar = bcast31_add(ar);
                    ai = bcast31_add(ai);
                }
                const int g = lane >> 4;
                const int row = g == 0 ? rows.x : g == 1 ? rows.y : g == 2 ? rows.z : rows.w;   // (-1 for lane groups that end no row)
                if ((lane & 15) == 15 && row >= 0) tile[row * FW + jj] = ar * ar + ai * ai;   // (the square root is taken when the tile is stored)
                ar = 0.f;
                ai = 0.f;
            };
            if constexpr (RESIDENT) {
                // every wave runs RES iterations (padding entries are 0 * bin 0); the step ends are scalar bit tests on a mask
                // that is re-read every frame -- hoisted out of the frame loop the RES conditions cost 2 SGPRs each and spill
                int s = s0, em = endmask;
                asm volatile("" : "+s"(em));
#pragma unroll
                for (int i = 0; i < RES; ++i) {
                    mac(kv[i], ad[i]);
                    if (em & (1 << i)) finish(s++);
                }
            } else {
                constexpr int GR = 4;   // iterations requested together
                int it = it0;
                for (int s = s0; s < s1; ++s) {
                    const int ni = __builtin_amdgcn_readfirstlane(iters_l[s]) & 0xffff;
                    for (int b = 0; b < ni; b += GR) {
                        KV kq[GR];
                        int aq[GR];
#pragma unroll
                        for (int g = 0; g < GR; ++g) {
                            const int e = b + g < ni ? (it + b + g) * 64 + lane : -1;
                            aq[g] = buf_load_i32(raddr, e * 4);
                            kq[g] = load_kv(e);
                        }
#pragma unroll
                        for (int g = 0; g < GR; ++g)
                            if (b + g < ni) mac(kq[g], aq[g]);
                    }
                    it += ni;
                    finish(s);
                }
            }
        }
        PROF_MARK(4);
        if (more) first_pass(p);   // (waits for the prefetched samples; the other waves are still contracting)
        PROF_MARK(5);
        lds_barrier();
        PROF_MARK(6);
    }
    lds_barrier();

    // ---- store the tile (FW * 4-B runs along t in the reference layout)
    const int nvalid = min(FW, T - t0);
    if (chroma_res > 0) {
        for (int idx = p; idx < chroma_res * FW; idx += P) {
            const int ch = idx / FW, jj = idx % FW;
            if (jj >= nvalid) continue;
            float acc = 0.f;
            for (int r = ch; r < n_bins; r += chroma_res) acc += __builtin_amdgcn_sqrtf(tile[r * FW + jj]);   // zaf.py:696-698
            if (layout == ZAFX_LAYOUT_FT) out[((long long)clip * chroma_res + ch) * TP + t0 + jj] = acc;   // TP = row pitch (>= T)
            else out[((long long)clip * T + t0 + jj) * chroma_res + ch] = acc;
        }
    } else {
        for (int idx = p; idx < n_bins * FW; idx += P) {
            const int r = idx / FW, jj = idx % FW;
            if (jj >= nvalid) continue;
            const float val = __builtin_amdgcn_sqrtf(tile[idx]);
            if (layout == ZAFX_LAYOUT_FT) out[((long long)clip * n_bins + r) * TP + t0 + jj] = val;
            else out[((long long)clip * T + t0 + jj) * n_bins + r] = val;
        }
    }
}

// LDS bytes of k_cqt for `frames` frames per tile; 0 when LOG2N is not built
template <int LOG2N>
static size_t cqt_lds(int n_bins, int n_steps, int frames) {
    constexpr int LOG2E = default_log2e(LOG2N);
    using C = FftCfg<LOG2N, LOG2E>;
    using G = CqtCfg<LOG2N, LOG2E>;
    return G::HEAD + (size_t)(C::P / 64) * 16 + (size_t)((C::P / 64 + 3) & ~3) * 4 + (size_t)n_steps * 16 + (size_t)((n_steps + 3) & ~3) * 4 +
           (size_t)n_bins * frames * sizeof(float);
}

template <int LOG2N>
static hipError_t run_cqt(const zafx_plan& pl, const float* x, float* out, int64_t n_clips, int64_t n_samples, int T) {
    constexpr int LOG2E = default_log2e(LOG2N);
    using C = FftCfg<LOG2N, LOG2E>;
    const int diff = pl.W - pl.H;                              // may be negative if step > fft_len
    const int left = diff >= 0 ? (diff + 1) / 2 : -((-diff) / 2);   // ceil(diff / 2)  (zaf.py:615)
    const bool aligned = n_samples % 2 == 0 && pl.H % 2 == 0 && left % 2 == 0 && reinterpret_cast<uintptr_t>(x) % 8 == 0;
    // a real matrix whose busiest wave has <= kCqtResident iterations keeps its entries in registers (16 would spill at 1024 threads)
    const bool realk = pl.cqt_real;
    const bool res = realk && pl.cqt_resident == kCqtResident;
    auto pick = [&](auto al) {
        constexpr bool AL = decltype(al)::value;
        return !realk ? k_cqt<LOG2N, LOG2E, AL, false, 0> : res ? k_cqt<LOG2N, LOG2E, AL, true, kCqtResident> : k_cqt<LOG2N, LOG2E, AL, true, 0>;
    };
    auto kern = aligned ? pick(std::true_type{}) : pick(std::false_type{});
    // frames per tile: 16 (64-B output runs) when the rows fit beside the frame, else 8, 4, 2, 1
    int fw = 16;
    while (fw > 1 && cqt_lds<LOG2N>(pl.prm.n_bins, pl.cqt_n_steps, fw) > (size_t)kMaxLdsBytes) fw >>= 1;
    const size_t smem = cqt_lds<LOG2N>(pl.prm.n_bins, pl.cqt_n_steps, fw);
    if (smem > (size_t)kMaxLdsBytes) {
        set_error("cqt: kernel matrix has too many rows for LDS at this fft_length");
        return hipErrorInvalidValue;
    }
    if (n_samples >= (1LL << 29)) {   // a clip is addressed through one buffer descriptor with 32-bit byte offsets
        set_error("cqt: clips of 2^29 samples or more are not supported (3.4 h at 44.1 kHz); cut the signal");
        return hipErrorInvalidValue;
    }
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, smem); e != hipSuccess) return e;
    const int tiles = (T + fw - 1) / fw;
    const long long blocks = (long long)tiles * n_clips;
    if (blocks <= 0) return hipSuccess;
    if (blocks > 0x7fffffffLL) {
        set_error("cqt: batch too large for one launch");
        return hipErrorInvalidValue;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(C::P), smem, pl.stream, x, pl.d_tw_pass, pl.d_tw_aux, pl.d_cqt_waves, pl.d_cqt_steps,
                       pl.d_cqt_addrs, pl.d_cqt_vals, out, (long long)n_samples, pl.H, left, T, (int)row_pitch(pl, T), tiles, pl.prm.n_bins,
                       pl.kind == ZAFX_CHROMA ? pl.prm.octave_resolution : 0, pl.layout, pl.cqt_n_steps, fw, pl.cqt_k_lo, pl.cqt_k_hi,
                       pl.cqt_k_special, std::max(pl.cqt_n_entries, 1));
    return hipGetLastError();
}

bool cqt_supported(int log2n) { return log2n >= 8 && log2n <= 14; }
int cqt_waves(int log2n) { return fft_threads(log2n, default_log2e(log2n)) / 64; }
const char* cqt_kernel_name() { return "k_cqt"; }

// Largest number of rows a float32 plan of this fft_length can hold (one frame per tile, ceil(rows / 4) steps)
int cqt_max_bins(int log2n) {
    auto fit = [](auto tag) {
        constexpr int L = decltype(tag)::value;
        int lo = 0, hi = 1 << 20;
        while (lo < hi) {
            const int mid = (lo + hi + 1) / 2;
            if (cqt_lds<L>(mid, (mid + 3) / 4, 1) <= (size_t)kMaxLdsBytes) lo = mid;
            else hi = mid - 1;
        }
        return lo;
    };
    switch (log2n) {
        case 8: return fit(std::integral_constant<int, 8>{});
        case 9: return fit(std::integral_constant<int, 9>{});
        case 10: return fit(std::integral_constant<int, 10>{});
        case 11: return fit(std::integral_constant<int, 11>{});
        case 12: return fit(std::integral_constant<int, 12>{});
        case 13: return fit(std::integral_constant<int, 13>{});
        case 14: return fit(std::integral_constant<int, 14>{});
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
    }
    set_error("cqt: unsupported fft_length");
    return hipErrorInvalidValue;
}

}  // namespace zafx

ZAFX_PROF_EXPORT(zafx_debug_prof_cqt, g_prof_cqt)
