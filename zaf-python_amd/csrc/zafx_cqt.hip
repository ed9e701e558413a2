// zafx_cqt.hip -- constant-Q spectrogram / chromagram kernel for gfx950 (MI355X).
//
// Replaces the per-frame loop of zaf.py:627-633:
//     cqt[:, j] = abs(cqt_kernel * np.fft.fft(xpad[j*step : j*step + fft_len]))
// with cqt_kernel a sparse (n_bins x fft_len) complex CSR matrix (zaf.py:554-557).
//
// One workgroup owns FW = 16 consecutive frames of one clip and transforms them one
// after the other: the whole frame (fft_len real = N = fft_len/2 complex points, up
// to 128 KiB) lives in LDS, owned by N/16 threads (1024 for fft_len 32768).  Frames of
// a tile overlap by (fft_len - step)/fft_len (94.6 % at config Q), so the re-reads of
// the input hit L2; HBM sees each sample about once per tile.  After the real-split the
// CSR rows are contracted against the one-sided spectrum in LDS (one wave per row,
// lanes across the row's contiguous non-zeros, shuffle reduction) and the magnitudes of
// the 16 frames are staged in LDS so that the (n_bins, T) store writes 64-B runs along t.
// The chromagram (zaf.py:693-698) is a strided row sum over that LDS tile.
#include <algorithm>

#include "zafx_fft.hpp"
#include "zafx_internal.hpp"

namespace zafx {

constexpr int kCqtFramesPerBlock = 16;
ZAFX_PROF_ARRAY(g_prof_cqt)

// LDS carve shared by the kernel and the launcher (bytes before the chunk descriptors, 16-B aligned:
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

template <int LOG2N, int LOG2E, bool ALIGNED>
__global__ __launch_bounds__(fft_threads(LOG2N, LOG2E)) void k_cqt(
    const float* __restrict__ x, const float2* __restrict__ twp, const float2* __restrict__ tws,
    const int4* __restrict__ chunks, const int* __restrict__ chunk_ptr, const int* __restrict__ slots, const float2* __restrict__ values, float* __restrict__ out,
    long long n_samples, int step, int left_pad, int T, int TP, int tiles, int n_bins, int chroma_res, int layout, int n_chunks,
    int k_lo, int k_hi, int k_special, int nnz) {
    using C = FftCfg<LOG2N, LOG2E>;
    using G = CqtCfg<LOG2N, LOG2E>;
    constexpr int N = C::N, P = C::P, E = C::E, W = 2 * N, FW = kCqtFramesPerBlock, NHI = G::NHI, NH2 = G::NH2;
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
    int4* chunk_l = reinterpret_cast<int4*>(smem_raw + G::HEAD);  // [n_chunks]
    float* tile = reinterpret_cast<float*>(chunk_l + n_chunks);   // [n_bins][FW]
    float2* part = reinterpret_cast<float2*>(tile + n_bins * FW);   // [n_bins][4] row sums of a frame's products
    int* chunk_ptr_l = reinterpret_cast<int*>(part + n_bins * 4);   // [waves + 1]
    const int p = threadIdx.x;
    for (int i = p; i < NHI + 128; i += P) tw_hi[i] = twp[i];
    if constexpr (G::SPLIT)
        for (int i = p; i < G::NSUB; i += P) sub_hi[i] = twp[NHI + 128 + i];
    for (int i = p; i < NH2 + 128; i += P) sp_hi[i] = tws[i];
    for (int i = p; i < n_chunks; i += P) chunk_l[i] = chunks[i];
    for (int i = p; i <= P / 64; i += P) chunk_ptr_l[i] = chunk_ptr[i];
    lds_barrier();
    const TwoLevelTw tw2l{tw_hi, tw_lo};
    const int wave = p >> 6;
    const int clip = blockIdx.x / tiles, tl = blockIdx.x % tiles;
    const int t0 = tl * FW;
    const float* xc = x + (long long)clip * n_samples;
    // wave-uniform values read from LDS land in VGPRs; readfirstlane tells the compiler they are scalars
    // (scalar branches and SGPR operands instead of exec-mask juggling around every chunk)
    const int c0 = __builtin_amdgcn_readfirstlane(chunk_ptr_l[wave]), c1 = __builtin_amdgcn_readfirstlane(chunk_ptr_l[wave + 1]);
    auto chunk_at = [&](int c) {
        const int4 d = chunk_l[c];
        return make_int4(__builtin_amdgcn_readfirstlane(d.x), __builtin_amdgcn_readfirstlane(d.y), __builtin_amdgcn_readfirstlane(d.z),
                         __builtin_amdgcn_readfirstlane(d.w));
    };

    // ---- framing, no window (it lives in the kernel): zaf.py:612-620, :631.  Frames inside the clip take
    // unconditional 8-byte loads; the zero-padded edge frames take the predicated path.
    float2 v[E];
    const unsigned clip_bytes = (unsigned)std::min<long long>(n_samples * 4, 0xfffffffcLL);
    const auto rx = make_rsrc(xc, clip_bytes);
    const auto rslots = make_rsrc(slots, (unsigned)nnz * 4u), rvals = make_rsrc(values, (unsigned)nnz * 8u);
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
    if (t0 < T) load_frame(t0, threadIdx.x);
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
            // 16384 = 16 x 1024.  Radix-16 across the workgroup on the samples 1024 apart (thread p holds n2 = p), times
            // w^(p k1); output k1 goes to sub-sequence k1 at position p.  Then wave w transforms sub-sequence w on its own
            // (three wave-local passes, no workgroup barrier): X[k1 + 16 k2] = FFT_1024(sub-sequence k1)[k2].
            static_assert(E == 16 && P == 1024, "split form: 16 points per thread, 16 wavefronts");
            Dft<16>::run(v);
            {
                float2 w[16];
                w[1] = tw2(tw2l, p);
#pragma unroll
                for (int r = 2; r < 16; ++r) w[r] = cmul(w[r >> 1], w[r - (r >> 1)]);
#pragma unroll
                for (int r = 1; r < 16; ++r) v[r] = cmul(v[r], w[r]);
            }
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
        // ---- CSR mat-vec, part 1: request the slots and values of my first G chunks now; the L2 round
        // trip hides under the real split
        constexpr int G1 = 8;   // two rounds cover a wave's share at config Q (238 chunks over 16 waves)
        int slot[G1];
        float2 kv[G1];
        auto request = [&](int cb) {
#pragma unroll
            for (int g = 0; g < G1; ++g) {
                const int4 ch = cb + g < c1 ? chunk_at(cb + g) : make_int4(0, 0, 0, 0);
                // branch-free: lanes past the chunk read out of range, i.e. 0 (a select on the loaded value
                // would make the wave wait for the load right here)
                const bool on = lane < ch.z;
                slot[g] = buf_load_i32(rslots, on ? (ch.y + lane) * 4 : -4);
                kv[g] = buf_load_f32x2(rvals, on ? (ch.y + lane) * 8 : -8);
            }
        };
        request(c0);
        // ---- real split in place, only for the pairs (k, N-k) that the kernel's columns touch:
        // slots 0..N-1 <- X[0..N-1], slot PITCH-1 <- X[N];  t_k = exp(-2 pi i k / 2N) = sp_hi[k >> 7] sp_lo[k & 127]
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
        // ---- CSR mat-vec against the spectrum + magnitude (zaf.py:630-632).  The host cut the rows into
        // chunks of <= 64 non-zeros, dealt whole rows to the wavefronts (balanced) and translated every
        // column into its LDS slot (bit 31: use the conjugate, i.e. a column of the upper half).  Chunk
        // descriptors sit in LDS; the slots and values of G chunks are requested together, so a frame
        // pays ceil(chunks / G) L2 round trips instead of two per chunk.  Order of the requests: group 1
        // before the split (above), group 2 after group 1 is consumed, THEN the next frame's samples
        // (v is free until the next FFT) -- loads return in order, so group 2 must not queue behind
        // the 128-KB frame prefetch.
        {
            float ar = 0.f, ai = 0.f;
            auto contract = [&](int cb) {
#pragma unroll
                for (int g = 0; g < G1; ++g) {
                    if (cb + g >= c1) break;
                    const int4 ch = chunk_at(cb + g);   // {row, first entry, count, last-of-row}
                    if (lane < ch.z) {
                        float2 xv = buf[slot[g] & 0x7fffffff];
                        if (slot[g] < 0) xv.y = -xv.y;
                        ar += kv[g].x * xv.x - kv[g].y * xv.y;
                        ai += kv[g].x * xv.y + kv[g].y * xv.x;
                    }
                    if (ch.w) {   // end of row ch.x: leave the four 16-lane partial sums in LDS (finished below)
                        ar = row16_sum(ar);
                        ai = row16_sum(ai);
                        if ((lane & 15) == 0) part[ch.x * 4 + (lane >> 4)] = make_float2(ar, ai);
                        ar = 0.f;
                        ai = 0.f;
                    }
                }
            };
            contract(c0);
            if (c0 + G1 < c1) request(c0 + G1);
            if (jj + 1 < FW && t + 1 < T) load_frame(t + 1, p);
            PROF_MARK(4);
            if (c0 + G1 < c1) contract(c0 + G1);
            for (int cb = c0 + 2 * G1; cb < c1; cb += G1) {
                request(cb);
                contract(cb);
            }
        }
        PROF_MARK(5);
        lds_barrier();
        PROF_MARK(6);
        // |.|^2 of every row of this frame (the square root is taken once, when the tile is stored)
        for (int r = p; r < n_bins; r += P) {
            const float4 a = *reinterpret_cast<const float4*>(part + r * 4), b = *reinterpret_cast<const float4*>(part + r * 4 + 2);
            const float sr = (a.x + a.z) + (b.x + b.z), si = (a.y + a.w) + (b.y + b.w);
            tile[r * FW + jj] = sr * sr + si * si;
        }
    }
    lds_barrier();

    // ---- store the tile (64-B runs along t in the reference layout)
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

template <int LOG2N>
static hipError_t run_cqt(const zafx_plan& pl, const float* x, float* out, int64_t n_clips, int64_t n_samples, int T) {
    constexpr int LOG2E = default_log2e(LOG2N);
    using C = FftCfg<LOG2N, LOG2E>;
    using G = CqtCfg<LOG2N, LOG2E>;
    const int diff = pl.W - pl.H;                              // may be negative if step > fft_len
    const int left = diff >= 0 ? (diff + 1) / 2 : -((-diff) / 2);   // ceil(diff / 2)  (zaf.py:615)
    const bool aligned = n_samples % 2 == 0 && pl.H % 2 == 0 && left % 2 == 0 && reinterpret_cast<uintptr_t>(x) % 8 == 0;
    auto kern = aligned ? k_cqt<LOG2N, LOG2E, true> : k_cqt<LOG2N, LOG2E, false>;
    const size_t smem = G::HEAD + (size_t)pl.n_chunks * 16 + (size_t)pl.prm.n_bins * (kCqtFramesPerBlock * sizeof(float) + 4 * sizeof(float2)) + (size_t)(C::P / 64 + 1) * 4;
    if (smem > (size_t)kMaxLdsBytes) {
        set_error("cqt: kernel matrix (bins / non-zeros) too large for LDS at this fft_length");
        return hipErrorInvalidValue;
    }
    if (n_samples >= (1LL << 29)) {   // a clip is addressed through one buffer descriptor with 32-bit byte offsets
        set_error("cqt: clips of 2^29 samples or more are not supported (3.4 h at 44.1 kHz); cut the signal");
        return hipErrorInvalidValue;
    }
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.device, smem); e != hipSuccess) return e;
    const int tiles = (T + kCqtFramesPerBlock - 1) / kCqtFramesPerBlock;
    const long long blocks = (long long)tiles * n_clips;
    if (blocks <= 0) return hipSuccess;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(C::P), smem, pl.stream, x, pl.d_tw_pass, pl.d_tw_aux, pl.d_chunks, pl.d_chunk_ptr, pl.d_slots,
                       pl.d_values, out, (long long)n_samples, pl.H, left, T, (int)row_pitch(pl, T), tiles, pl.prm.n_bins,
                       pl.kind == ZAFX_CHROMA ? pl.prm.octave_resolution : 0, pl.layout, pl.n_chunks, pl.cqt_k_lo, pl.cqt_k_hi,
                       pl.cqt_k_special, std::max(pl.nnz, 1));
    return hipGetLastError();
}

bool cqt_supported(int log2n) { return log2n >= 8 && log2n <= 14; }
int cqt_waves(int log2n) { return fft_threads(log2n, default_log2e(log2n)) / 64; }
const char* cqt_kernel_name() { return "k_cqt"; }

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
