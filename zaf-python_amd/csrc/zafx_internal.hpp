// zafx_internal.hpp -- plan object and launcher prototypes shared by the C-ABI
// translation unit (zafx_capi.cpp) and the kernel translation units (*.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <atomic>
#include <string>
#include <vector>

#include "../../include/zafx.h"
#include "zafx_twiddle.hpp"

namespace zafx {

constexpr int kMaxLdsBytes = 160 * 1024;   // LDS per CU on gfx950
// k_cqt at fft_length 32768: the 16384-point packed transform is split as 16 x 1024 -- one radix-16 pass across the
// workgroup, then every wavefront transforms its own 1024-point sub-sequence without meeting the others.  Bin k of
// the result then lives in sub-transform k & 15 at position k >> 4: LDS slot of bin k (kCqtRegion = 2 mod 32 complex,
// so 64 consecutive bins hit distinct banks), and the slot that holds X[N].
#ifndef ZAFX_CQT_SPLIT
#define ZAFX_CQT_SPLIT 1
#endif
// fft_length 65536 (log2n = 15; minimum frequencies down to 23 Hz at 44.1 kHz and 24 bins per octave): the 32768-point packed
// transform does not fit LDS, so a frame is transformed as the EVEN and the ODD bins of two 16384-point transforms (decimation in
// frequency: Z[2q] = FFT(z[n] + z[n + 16384])[q], Z[2q + 1] = FFT((z[n] - z[n + 16384]) w^n)[q]) that use the 16 x 1024 machinery
// one after the other.  A pair (k, N - k) of the real split lies in ONE of the two, so each is split on its own; the even bins
// wait in registers and then sit beside the odd ones in the LDS image: odd bin k at the slot of position k >> 1, even bin k at
// the slot of position 8192 + (k >> 1).  That needs the kernel matrix to touch one-sided bins 1 .. kCqtDoubleMaxBin only (5.5 kHz
// at 44.1 kHz -- the low-frequency kernels that need such a long frame do); others run on the float64 kernel.
constexpr bool cqt_double(int log2n) { return log2n == 15; }
constexpr int kCqtDoubleMaxBin = 8191;
constexpr bool cqt_split(int log2n) { return ZAFX_CQT_SPLIT && (log2n == 14 || log2n == 15); }
constexpr int kCqtRegion = 1090;
constexpr int cqt_slots(int log2n) { return cqt_split(log2n) ? 16 * kCqtRegion : (1 << log2n) + ((1 << log2n) >> 4) + 1; }
constexpr int cqt_slot14(int k) { return (k & 15) * kCqtRegion + (k >> 4) + (k >> 8); }
constexpr int cqt_slot(int log2n, int k) {
    return cqt_double(log2n) ? ((k & 1) ? cqt_slot14(k >> 1) : cqt_slot14(8192 + (k >> 1))) : cqt_split(log2n) ? cqt_slot14(k) : k + (k >> 4);
}
constexpr int cqt_nyquist_slot(int log2n) { return cqt_split(log2n) ? kCqtRegion - 1 : (1 << log2n) + ((1 << log2n) >> 4); }
constexpr int kMel2Slots = 3;   // k_mel2: partial tiles of cut filterbank blocks (1 KB of LDS each)   // k_mel2: longest item of the filterbank product, partial tiles per 16-frame tile
constexpr int kCqtMmSteps = 14;            // k_cqt, matrix-core contraction: steps (entries per lane) a wave keeps in registers
constexpr int kCqtResident = 12;           // k_cqt: iterations (entries per lane) of a wave's share of the kernel matrix that ride in registers
constexpr int kCqt64Sub = 4096;            // float64 CQT: length of the sub-transforms that fit LDS (2 x 4096 x 16 B)

// Banded, MFMA-fragment-packed matrix (mel filterbank or DCT-II rows) cut into balanced work
// items: see zafx_mel.hip and pack_band in zafx_capi.cpp.
struct PackedBand {
    int n_rows = 0;        // logical rows (filters / coefficients)
    int n_blocks = 0;      // ceil(n_rows / 16)
    int n_cols = 0;        // logical K extent
    int n_items = 0;       // work items (a block's band cut into parts of bounded length)
    int n_waves = 0;       // wavefronts the items were dealt to
    int total_steps = 0;
    float* d_pack = nullptr;     // [total_steps][64] : lane l -> A[l & 15][4*step + (l >> 4)]
    int4* d_items = nullptr;     // per wave, longest first: {slot id, first column, steps, offset into d_pack (steps)}
    int* d_wave_ptr = nullptr;   // [n_waves + 1] ranges into d_items
    int* d_blk_ptr = nullptr;    // [n_blocks + 1] ranges of slot ids belonging to a block
    float* d_direct = nullptr;   // DCT rows as MFMA A fragments for the register-fed form of k_mel: [wave][j][row block][64 lanes], K-step w + 16 j
    int direct_j = 0;            // K-steps per wave of that form (0: not available)
    unsigned short* d_desc = nullptr;   // [total_steps] K-step descriptors of the resident form: first column / 4 | slot id << 8 | item ends << 15
    bool desc_ok = false;        // every step fits the 16-bit descriptor
    int n_empty = 0;             // blocks without non-zeros (their zero tiles are written by the streamed form only)
    // k_mel2: whole 16-row blocks (one longer than a SIMD's quarter of the steps is cut in K; at most kMel2Slots cuts) dealt to the four
    // SIMDs longest first, then to a SIMD's waves w, w + 4, w + 8, w + 12, at most two items per wave: d_whole [wave][2] = {first column,
    // steps (-1: no item), offset into d_pack (steps), block | role << 8 (1 owner: stores the tile, 2 helper: hands its partial tile
    // over through LDS) | helper's slot << 10 | owner's mask of slots to add << 12}
    int4* d_whole = nullptr;
    bool whole_ok = false;
    std::vector<int> h_owner;    // (filterbank) wave | place << 8 of the item that owns block b's tile
    // k_mel2, mfcc (on the DCT band): d_dct2 [filterbank block][2 coefficient blocks][4 steps][64] = the DCT rows as A fragments over the
    // block's sixteen filters (lane l -> D[16 c + (l & 15)][16 b + 4 s + (l >> 4)]); d_owner2 [filterbank blocks] = the owner waves
    float* d_dct2 = nullptr;
    int* d_owner2 = nullptr;
    bool dct2_ok = false;
    int max_wave_steps = 0;      // steps of the busiest wave: wave w owns steps [total w / n_waves, total (w + 1) / n_waves)
};
constexpr int kMelResidentFb = 18;  // K-steps of the filterbank / of the DCT rows a wave of k_mel keeps in registers
constexpr int kMelResidentDct = 4;

}  // namespace zafx

struct zafx_plan {
    int device = 0;
    int n_cus = 256;   // compute units of the device (persistent-kernel grid size)
    int kind = 0;
    zafx_params prm{};
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // zafx_run_host: upload / download streams, the events between the pipeline's stages ([0..1] upload done, [2..3] kernel
    // done, [4..5] download done, per buffer set) and two sets of device staging buffers (grow-only, freed with the plan)
    hipStream_t stream_up = nullptr, stream_down = nullptr;
    hipEvent_t pipe_ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    void* lane_in[2] = {nullptr, nullptr};
    void* lane_out[2] = {nullptr, nullptr};
    size_t lane_in_bytes[2] = {0, 0}, lane_out_bytes[2] = {0, 0};
    void* lane_pcm[2] = {nullptr, nullptr};   // zafx_run_host_pcm: the uploaded integer PCM of a chunk
    size_t lane_pcm_bytes[2] = {0, 0};

    int W = 0;        // window length (or CQT fft_length)
    int H = 0;        // hop / step
    int layout = 0;
    int log2nf = 0;   // log2 of the complex FFT length the kernels run
    int log2e = 0;

    // device constants
    float* d_window = nullptr;
    float* d_matrix = nullptr;     // ZAFX_LINEAR: dense transform matrix [n_filters][window_length]
    float4* d_wfold = nullptr;     // MDCT: sign-folded window, 4 taps per packed input (zafx_mdct.hip)
    float2* d_tw_pass = nullptr;   // per-pass twiddles for fft_frame<log2nf, log2e>
    float2* d_tw_aux = nullptr;    // real-split roots (STFT family / CQT) or tw8 (MDCT family)
    float cola_gain = 0.f;         // sum(w[0:W:H])  (zaf.py:241)
    zafx::PackedBand fb, dct;
    int* d_indptr = nullptr;
    int* d_indices = nullptr;
    float2* d_values = nullptr;
    int nnz = 0;
    // k_cqt's view of the kernel matrix (build_cqt_chunks in zafx_capi.cpp): rows sorted by length and dealt out in "steps"
    // (four rows of 16 lanes, two of 32 or one of 64), steps dealt to the wavefronts; entry (iteration, lane) = value + LDS
    // byte address of its spectrum bin
    int4* d_cqt_waves = nullptr;   // [waves]: {first iteration, iterations, step-end mask (bit i: iteration i ends a step), 0}
    int* d_cqt_addrs = nullptr;    // [iterations][64]: bits 0-17 byte address of the bin in the LDS spectrum image, 31 conjugate, 29-30 shape of
                                   // a step that ends here (0: none), 18-28 row this lane then writes (0x7ff: none)  (zafx_cqt.hip)
    float* d_cqt_vals = nullptr;   // [iterations][64] float (real matrix) or float2
    int cqt_n_entries = 0;
    bool cqt_real = false;
    int cqt_resident = 0;          // kCqtResident: the busiest wave's iterations fit the registers; 0: entries streamed from L2 every frame
    // the matrix-core form of the contraction (zafx_cqt.hip, "MM"): rows in pairs, a pair's columns cut into segments of cqt_mm_steps
    // consecutive entries, two segments per 4-lane block of v_mfma_f32_4x4x1_16b_f32; 0: not available for this matrix
    int cqt_mm_steps = 0;
    int cqt_mm_segs = 0;              // segments of the longest pair
    float* d_cqt_mm_vals = nullptr;   // [waves][steps][64]: lane (blk, i): K[row 2 pair + (i & 1)][column of the step] of the block's stream i >> 1
    int* d_cqt_mm_addr = nullptr;     // [waves][steps][64]: lane (blk, j): LDS byte address of the re (j even) / im (j odd) part of that column's bin, stream j >> 1
    int* d_cqt_mm_fin = nullptr;      // [pairs]: first stream slot | segments << 16 of the pair (its segments are consecutive stream slots)
    // mel / mfcc plans of W = 4096 / 8192 (k_melfb): filterbank rows as bands of float32 (values of [first, first + count) of every
    // row back to back; meta [n_filters][3] = first column, count, offset) and the DCT rows dense [n_coefs][n_filters]
    float* d_fbw = nullptr;
    int* d_fbw_meta = nullptr;
    float* d_dctw = nullptr;
    float2* d_tw_band = nullptr;   // k_mdct_ft32b (MDCT plans of W = 4096): gb[s][q] = g[2 q + s] (1024 entries), then bt[n] = g[n] exp(-2 pi i n / 1024) (512)
    float2* d_tw_sub = nullptr;    // pass twiddles of the band transforms: 1024 points (k_stft_ft16b; STFT / mel plans of W = 4096), 512 points (k_mdct_ft32b; MDCT plans of W = 4096)
    float2* d_tw_quad = nullptr;   // exp(-2 pi i n / 8192), n < 4096 (k_stft_ft16q; STFT / mel plans of W = 8192)
    float2* d_tw_r32 = nullptr;    // pass twiddles of the radix-32 schedule (1024 points as 32 x 32), STFT plans of W = 2048
    // float64 mode (ZAFX_PRECISION_F64, zafx_f64.hip)
    double* d_window64 = nullptr;
    double2* d_tw64 = nullptr;     // exp(-2 pi i m / (W/2)), m < W/4
    double2* d_tws64 = nullptr;    // exp(-2 pi i k / W), k <= W/4
    double* d_scratch64 = nullptr; // ISTFT: time-domain frames of the current call (grow-only)
    size_t scratch_bytes = 0;
    double cola_gain64 = 0.0;
    std::vector<double> h_window64, h_fb64, h_dct64;
    double* d_fb64 = nullptr;      // mel filterbank rows as bands: values of [first, first + count) of every row, back to back
    int* d_fb64_meta = nullptr;    // [n_filters][3]: first column, count, offset into d_fb64
    double* d_dct64 = nullptr;     // [n_coefs][n_filters]
    // k_mel_ft8_f64 (W = 2048, reference layout; build_mel64_fb in zafx_f64.hip, zafx_mel64.hpp): the filterbank's non-zeros as one equally
    // long stream per lane
    int4* d_mel64_stream = nullptr;   // [steps][64] entries {value (2 words), column, slot of the partial sum or -1}
    int2* d_mel64_fin = nullptr;      // [n_filters]: {first slot, slots} of a filter's partial sums (consecutive, ascending columns)
    double* d_mel64_dctT = nullptr;   // mfcc: DCT-II rows transposed, [2 mel64_dct_half filters][coefficient pitch (a multiple of 32)], zeros outside
    int mel64_steps = 0, mel64_slots = 0, mel64_max_parts = 0, mel64_cpitch = 0, mel64_dct_half = 0;
    bool mel64_ok = false;
    double2* d_values64 = nullptr; // CQT kernel values of a float64 plan (complex128)
    // k_cqt_ft_f64 (fft_length 32768; build_cqt64 in zafx_f64.hip, zafx_cqt64.hpp)
    double2* d_cqt64_tw1 = nullptr;     // [4][1024]: exp(-2 pi i m 2^i / 16384), the first pass's twiddles (products of these)
    int* d_cqt64_split = nullptr;       // [2 rounds][kc2][512]: the one-sided bins a thread splits (bin | compact index << 14, -1: none)
    double* d_cqt64_vals = nullptr;     // [steps][512] float64 (cqt64_real) or complex128: the matrix's non-zeros as one stream per thread ...
    int* d_cqt64_meta = nullptr;        // ... compact index | conjugate << 13 | (slot of the partial sum + 1) << 14
    int2* d_cqt64_fin = nullptr;        // [rows]: {first slot, slots}
    int cqt64_kc2 = 0, cqt64_cols = 0, cqt64_steps = 0, cqt64_slots = 0, cqt64_max_parts = 0, cqt64_klo = 511;   // klo: (highest kernel column) >> 4, what the sub-transforms' last pass may skip (fft1024_cq)
    bool cqt64_ok = false, cqt64_dirty = true, cqt64_real = false;
    int bs_log2m = 0;              // > 0: window that is not a power of two -- Bluestein convolution length 2^bs_log2m (zafx_f64.hip, zafx_bs32.hip)
    void* d_pcm_float = nullptr;   // zafx_execute_pcm's float32 staging for the kinds that do not (grow-only)
    size_t pcm_float_bytes = 0;
    int dct_half = 0;              // ZAFX_DCT, types II-IV, N = 4 j with N / 2 not a power of two: N / 2, the points of the transform inside k_dct<.., BS> (zafx_dct.hip)
    long long dct_den2 = 0;        // ZAFX_DCT on the chirp-z form: 2 D, the denominator of its chirp exp(-i pi j^2 / (2 D)) (k_dct_bs32)
    float2* d_bs_chirp = nullptr;  // float32 Bluestein plans: c[n] = exp(-i pi n^2 / W), n < W
    float2* d_bs_bhat = nullptr;   // ... and FFT_M of the wrapped conjugate chirp
    double2* d_bhat64 = nullptr;   // FFT of the wrapped conjugate chirp, 2^bs_log2m entries
    std::vector<double2> h_values64;
    int cqt_k_lo = 0, cqt_k_hi = -1, cqt_k_special = 0;   // real-split pairs the kernel's columns need
    bool cqt_dirty = true;

    // host shadows (needed to re-pack after an RCCL broadcast)
    std::vector<float> h_window, h_fb, h_dct, h_matrix;
    std::vector<int32_t> h_indptr, h_indices;
    std::vector<zafx::cf32> h_values;

    std::string kernel_name;            // the kernel this plan is expected to run (set at creation)
    mutable std::atomic<const char*> ran{nullptr};  // the kernel the last execute really launched (routes depend on T, alignment and hop); zafx_plan_last_kernel_name
};

namespace zafx {

// Every launcher enqueues on plan.stream and returns hipGetLastError().
// Frames between the starts of consecutive rows of a plan's (F, T) array: T rounded up to prm.row_align elements
// (reference-layout plans only; 0 / 1 = compact, the reference's own memory order).
inline int64_t row_pitch(const zafx_plan& pl, int64_t T) {
    const int64_t a = pl.prm.row_align;
    return (pl.layout == ZAFX_LAYOUT_FT && a > 1) ? (T + a - 1) / a * a : T;
}

hipError_t launch_stft(const zafx_plan& pl, const float* x, float2* out, int64_t n_clips, int64_t n_samples, int T);
hipError_t launch_istft(const zafx_plan& pl, const float2* spec, float* y, int64_t n_clips, int T, int64_t out_len);
hipError_t launch_stft_f64(const zafx_plan& pl, const double* x, double2* out, int64_t n_clips, int64_t n_samples, int T);
hipError_t launch_istft_f64(zafx_plan& pl, const double2* spec, double* y, int64_t n_clips, int T, int64_t out_len);
hipError_t launch_cqt_f64(zafx_plan& pl, const double* x, double* out, int64_t n_clips, int64_t n_samples, int T);
hipError_t launch_mel_f64(const zafx_plan& pl, const double* x, double* out, int64_t n_clips, int64_t n_samples, int T);
hipError_t launch_mdct_f64(const zafx_plan& pl, const double* x, double* out, int64_t n_clips, int64_t n_samples, int T);
hipError_t build_mel64_fb(zafx_plan& pl);    // host tables of k_mel_ft8_f64 (zafx_f64.hip), rebuilt whenever the filterbank / DCT constant is set
hipError_t build_mel64_dct(zafx_plan& pl);
hipError_t launch_imdct_f64(zafx_plan& pl, const double* coefs, double* y, int64_t n_clips, int T, int64_t out_len);
const char* stft_f64_kernel_name();
const char* istft_f64_kernel_name();
const char* mdct_f64_kernel_name();
const char* mel_f64_kernel_name();
const char* cqt_f64_kernel_name();
const char* imdct_f64_kernel_name();
// float32 Bluestein forms (zafx_bs32.hip): windows of 33 ... 8192 samples that are not a power of two (convolution length
// M = 2^ceil(log2(2 W - 1)) <= 16384, about 139 KB of LDS per frame); the float64 Bluestein forms stop at 2048 samples
bool bs32_supported(int W);
hipError_t launch_stft_bs32(const zafx_plan& pl, const float* x, float2* out, int64_t n_clips, int64_t n_samples, int T);
hipError_t launch_istft_bs32(zafx_plan& pl, const float2* spec, float* y, int64_t n_clips, int T, int64_t out_len);
hipError_t launch_mdct_bs32(const zafx_plan& pl, const float* x, float* out, int64_t n_clips, int64_t n_samples, int T);
hipError_t launch_imdct_bs32(zafx_plan& pl, const float* coefs, float* y, int64_t n_clips, int T, int64_t out_len);
hipError_t launch_dct_bs32(const zafx_plan& pl, const float* x, float* y, int64_t n_rows);   // zaf.dct / zaf.dst of any length <= 8192 as chirp-z sums
// plan-owned scratch of the inverse transforms that park their time-domain frames (zafx_f64.hip): grow-only, and the number
// of clips per pass that keeps it under the budget (1 GiB; ZAFX_SCRATCH_BUDGET_MB)
hipError_t grow_scratch(zafx_plan& pl, size_t need);
int64_t scratch_clips_per_chunk(int64_t n_clips, int T, int W, size_t elem_bytes);
hipError_t launch_mdct(const zafx_plan& pl, const float* x, float* out, int64_t n_clips, int64_t n_samples, int T);
hipError_t launch_imdct(const zafx_plan& pl, const float* coefs, float* y, int64_t n_clips, int T, int64_t out_len);
hipError_t launch_mel(zafx_plan& pl, const float* x, float* out, int64_t n_clips, int64_t n_samples, int T);   // (windows above 2048 samples grow the plan's scratch)
const char* mel_wide_kernel_name();
// mel / mfcc plans that run as a spectrum kernel + k_melfb (or, W = 4096 with up to 256 filters, the fused two-band kernel) instead of k_mel
inline bool mel_takes_wide_route(const zafx_plan& pl) { return pl.log2nf >= 11 || pl.bs_log2m > 0 || pl.prm.n_filters > 256; }
bool mel_band_usable(const zafx_plan& pl, const float* x, int64_t n_clips, int64_t n_samples, int T);   // W = 4096: the fused two-band kernel k_mel_ft16b (zafx_stft.hip)
hipError_t launch_mel_band(const zafx_plan& pl, const float* x, float* out, int64_t n_clips, int64_t n_samples, int T);
hipError_t launch_cqt(const zafx_plan& pl, const float* x, float* out, int64_t n_clips, int64_t n_samples, int T);
bool pcm_direct_ok(const zafx_plan& pl, int64_t n_frames, int n_channels, int sample_bytes);   // zafx_mel.hip
bool stft_pcm_direct_ok(const zafx_plan& pl, int64_t n_frames, int n_channels, int sample_bytes, const void* d_pcm, int T);   // zafx_stft.hip
bool mdct_pcm_direct_ok(const zafx_plan& pl, int64_t n_frames, int n_channels, int sample_bytes, const void* d_pcm);   // zafx_mdct.hip
bool launch_spec2(const zafx_plan& pl, const float* x, float* out, int64_t n_clips, int64_t n_samples, int T, hipError_t& err);   // zafx_mel.hip: |X| / |X|^2 rows of an STFT plan on k_mel2
hipError_t launch_linear(const zafx_plan& pl, const float* x, float* y, int64_t n_clips);
hipError_t launch_dct(const zafx_plan& pl, const float* x, float* y, int64_t n_rows);   // zafx_dct.hip: dct / dst I-IV on the FFT core
bool dct_supported(int log2m);   // log2 of the complex FFT length M
const char* dct_kernel_name();
const char* linear_kernel_name();
hipError_t launch_pcm_to_float(hipStream_t stream, const void* pcm, float* out, int64_t n_total, int n_channels, int sample_bytes);

// names of the dominant kernels (what rocprofv3 --kernel-trace prints, prefix match)
const char* stft_kernel_name(int log2n, int layout);
const char* istft_kernel_name(int log2n, int layout);
const char* mdct_kernel_name(int log2nf, int layout);
const char* imdct_kernel_name();
const char* mel_kernel_name();
int mel_waves(int log2n);   // wavefronts per workgroup of k_mel for this FFT size
const char* cqt_kernel_name();

bool stft_supported(int log2n);   // log2 of complex FFT length = log2(W) - 1
bool mdct_supported(int log2nf);  // log2(W) - 2
bool cqt_supported(int log2n);    // log2(fft_length) - 1
int cqt_waves(int log2n);         // wavefronts per workgroup of k_cqt
int cqt_max_bins(int log2n);      // rows of a kernel matrix the float32 k_cqt can hold in LDS beside the frame
int stft_frames_per_block(int log2n, int layout);
int mdct_frames_per_block(int log2nf, int layout);

void set_error(const std::string& msg);

// zafx_execute_pcm -> launcher hand-off of "the input is int16, this many channels" (1 mono, 2 stereo; 0: float samples).  Per THREAD, not per
// plan (plans are cached and shared between threads: a field of the plan let a concurrent float execute pick the int16 kernel), set
// around the one zafx_execute the caller makes.  A launcher that reads int16 itself TAKES the mode (take_pcm_mode); execute_pcm fails hard when
// nobody took it -- a route that forgot it would feed int16 to a float kernel.
int take_pcm_mode();           // the calling thread's mode; marks it taken when non-zero
void set_pcm_mode(int mode);   // (execute_pcm only) also clears the taken mark
bool pcm_mode_taken();

// Carry kernels (k_istft_ft16, k_imdct): number of segments to cut every clip's tile sequence into so that
// `grid` persistent workgroups are evenly loaded (1 = whole clips; each extra segment pays one carry-only tile).
int carry_segments(long long n_clips, int tiles, long long grid);

// Raise a kernel's dynamic-LDS limit to `bytes` on `device` (once per (kernel, device); thread safe).
hipError_t ensure_dynamic_lds(const void* kernel, int device, size_t bytes);

}  // namespace zafx

// ---------------------------------------------------------------------------------
// Optional per-phase cycle counters (build with -DZAFX_PROF; tools/prof_phases.py reads them).
// One wave of one workgroup accumulates the cycles between consecutive PROF_MARKs.
// ---------------------------------------------------------------------------------
#ifdef ZAFX_PROF
#define ZAFX_PROF_ARRAY(name) __device__ unsigned long long name[16]; __device__ int name##_thread = 64;   // thread whose wave is timed
#define PROF_INIT(name)                            \
    unsigned long long* const prof_ = name;        \
    const int prof_thread_ = name##_thread;        \
    unsigned long long tprev_ = __builtin_readcyclecounter()
#define PROF_MARK(i)                                                                       \
    do {                                                                                   \
        const unsigned long long now_ = __builtin_readcyclecounter();                     \
        if (blockIdx.x == 7 && (int)threadIdx.x == prof_thread_) atomicAdd(&prof_[i], now_ - tprev_);   \
        tprev_ = now_;                                                                     \
    } while (0)
#define ZAFX_PROF_EXPORT(fn, name)                                                                  \
    extern "C" int fn##_thread(int t) { return hipMemcpyToSymbol(HIP_SYMBOL(zafx::name##_thread), &t, sizeof(t)) != hipSuccess; } \
    extern "C" int fn(unsigned long long* out) {                                                    \
        unsigned long long zero[16] = {};                                                           \
        if (hipMemcpyFromSymbol(out, HIP_SYMBOL(zafx::name), sizeof(zero)) != hipSuccess) return 1; \
        return hipMemcpyToSymbol(HIP_SYMBOL(zafx::name), zero, sizeof(zero)) != hipSuccess;         \
    }
#else
#define ZAFX_PROF_ARRAY(name)
#define PROF_INIT(name)
#define PROF_MARK(i)
#define ZAFX_PROF_EXPORT(fn, name)
#endif
