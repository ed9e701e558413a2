/* zafx.h -- C-ABI of libzafx.so: MI355X (gfx950) kernels for the windowed-transform
 * hot path of zafarrafii/Zaf-Python (zaf.py).
 *
 * The reference has no FFI: its boundary is the Python signatures of zaf.py.  This
 * header is the boundary the reference-side binding (a ctypes stub)
 * binds instead of NumPy/SciPy for each call site below.  Plain C types only; no
 * exceptions cross; every function returns 0 on success or a non-zero code whose
 * text is available from zafx_last_error() (thread-local).
 *
 *   entry point / plan kind        replaces (zaf.py file:line)
 *   -----------------------------  -------------------------------------------------
 *   ZAFX_STFT                      stft            zaf.py:45-141  (np.pad :112, frame
 *                                                  loop :132-136, np.fft.fft :139)
 *   ZAFX_ISTFT                     istft           zaf.py:144-243 (np.fft.ifft :223,
 *                                                  overlap-add :226-233, trim :236, gain :241)
 *   ZAFX_MEL                       melspectrogram  zaf.py:324-375 (stft :369, abs :370,
 *                                                  np.matmul :373)
 *   ZAFX_MFCC                      mfcc            zaf.py:378-454 (power :437, matmul+log
 *                                                  :443-446, scipy.fftpack.dct :443-449)
 *   ZAFX_CQT                       cqtspectrogram  zaf.py:562-635 (np.pad :612, np.fft.fft
 *                                                  + CSR mat-vec + abs :630-632)
 *   ZAFX_CHROMA                    cqtchromagram   zaf.py:638-700 (strided row sums :693-698)
 *   ZAFX_MDCT                      mdct            zaf.py:984-1075 (per-frame FFT :1061-1073)
 *   ZAFX_IMDCT                     imdct           zaf.py:1078-1184 (FFT :1159, TDAC
 *                                                  overlap-add :1172-1179, trim :1182)
 *   ZAFX_DCT                       dct / dst       zaf.py:703-839, :842-981 (np.fft.fft of the 2N-2 /
 *                                                  2N+2 / 4N / 8N point extension :771, :791, :816, :834,
 *                                                  :909, :926, :948, :974; orthonormal scalings)
 *   zafx_plan_set_constant         operands built by melfilterbank zaf.py:246-321 and
 *                                  cqtkernel zaf.py:457-559 (scipy.sparse CSR, consumed
 *                                  at zaf.py:373, :445, :631)
 *
 * Ownership: host buffers belong to the caller; device buffers belong to whoever
 * called zafx_alloc; a plan owns its HIP stream, events, twiddle tables and constants.
 * Threading: a plan is bound to one device and one stream; calls on distinct plans
 * are thread-safe; calls on one plan must be serialised by the caller.
 * All device arrays are float32 / complex64 (interleaved re,im), C-contiguous -- float64 /
 * complex128 for plans created with ZAFX_PRECISION_F64.
 */
#ifndef ZAFX_H
#define ZAFX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZAFX_VERSION 101

typedef struct zafx_plan zafx_plan;
typedef struct zafx_comm zafx_comm;

enum zafx_kind {
    ZAFX_STFT = 1,   /* in (B, N) f32            -> out (B, W, T) c64 [FT] or (B, T, W) [TF]       */
    ZAFX_ISTFT = 2,  /* in (B, W, T)/(B, T, W)   -> out (B, T*H - (W-H)) f32    (one-sided: W/2+1 rows) */
    ZAFX_MDCT = 3,   /* in (B, N) f32            -> out (B, W/2, T) f32 [FT] or (B, T, W/2) [TF]   */
    ZAFX_IMDCT = 4,  /* in (B, W/2, T)/(B,T,W/2) -> out (B, (W/2)*(T-1) - 1) f32                   */
    ZAFX_MEL = 5,    /* in (B, N) f32            -> out (B, n_filters, T) or (B, T, n_filters)     */
    ZAFX_MFCC = 6,   /* in (B, N) f32            -> out (B, n_coefs, T)   or (B, T, n_coefs)  (with_mel: n_filters + n_coefs rows) */
    ZAFX_CQT = 7,    /* in (B, N) f32            -> out (B, n_bins, T)    or (B, T, n_bins)        */
    ZAFX_CHROMA = 8, /* in (B, N) f32            -> out (B, octave_resolution, T) or transposed    */
    ZAFX_LINEAR = 9, /* in (B, window_length) f32 -> out (B, n_filters) f32: y = M x per clip (a caller's own dense
                        map; dct / dst lengths the FFT form below does not take)                          */
    ZAFX_DCT = 10    /* in (B, N) f32 -> out (B, N) f32: the orthonormal dct / dst of zaf.py:703-839 / :842-981, type
                        params.transform_type = 1..4, params.transform_sine = 0 (dct) / 1 (dst), N = window_length;
                        one M-point complex FFT per vector where M = N/2 (N-1 / N+1 for type I) is a power of two 32..8192,
                        every other N from 2 to 8192 on Bluestein convolutions: types 2-4 of N = 4 j two transforms of
                        2^ceil(log2(N-1)) points around the N/2-point transform, the rest two of 2^ceil(log2(2N-1)) (chirp-z sum) */
};

enum zafx_layout {
    ZAFX_LAYOUT_FT = 0, /* reference memory order: frequency-major, time minor (zaf.py:128) */
    ZAFX_LAYOUT_TF = 1  /* frame-major: each frame's bins contiguous                        */
};

enum zafx_spectrum {      /* STFT output / ISTFT input rows                             */
    ZAFX_SPECTRUM_TWO_SIDED = 0, /* W rows, as np.fft.fft returns them (zaf.py:139) -- the reference contract  */
    ZAFX_SPECTRUM_ONE_SIDED = 1, /* rows 0..W/2 only (what every example keeps, zaf.py:83); the ISTFT completes
                                    X[W-k] = conj X[k], i.e. equals istft of the two-sided spectrum of a real signal */
    ZAFX_SPECTRUM_MAGNITUDE = 2, /* STFT only: |X[k]|, k = 0..W/2, real (float32 / float64) -- the spectrogram the
                                    examples compute, np.absolute(audio_stft[0:W/2+1]) (zaf.py:83)                   */
    ZAFX_SPECTRUM_POWER = 3      /* STFT only: |X[k]|^2, k = 0..W/2, real                                          */
};

enum zafx_precision {     /* device arithmetic and array types                          */
    ZAFX_PRECISION_F32 = 0, /* float32 / complex64 arrays and arithmetic (every kind)                        */
    ZAFX_PRECISION_F64 = 1  /* every kind but ZAFX_LINEAR: float64 / complex128 arrays AND constants (window float64[W],
                               mel filterbank / DCT rows float64, CQT values complex128); the reference's own dtype
                               (zaf.py:128, :139), results within 1e-12 of it (mfcc 1e-10); any power-of-two window.
                               window_length 2048 in the reference layout (CQT: fft_length 32768, kernel columns in the one-sided
                               bins 1..8191) runs on tiled kernels of its own -- k_stft_ft8_f64, k_istft_ft8_f64,
                               k_mdct_ft16_f64, k_imdct_ft16_f64 (round 5), k_mel_ft8_f64, k_cqt_ft_f64 (round 6): 2-3 x
                               behind the float32 kernels --, every other geometry one workgroup per frame (5-10 x)   */
};

enum zafx_constant {
    ZAFX_CONST_WINDOW = 1,      /* float32[W]                                   (window_function)  */
    ZAFX_CONST_MEL_FB = 2,      /* float32[n_filters * W/2], dense row-major    (FB.toarray())     */
    ZAFX_CONST_DCT = 3,         /* float32[n_coefs * n_filters], row-major      (DCT-II rows 1..)  */
    ZAFX_CONST_CQT_INDPTR = 4,  /* int32[n_bins + 1]                            (CSR of cqt_kernel) */
    ZAFX_CONST_CQT_INDICES = 5, /* int32[nnz], 0 <= col < fft_length                                */
    ZAFX_CONST_CQT_VALUES = 6,  /* complex64[nnz]                                                   */
    ZAFX_CONST_MATRIX = 7       /* float32[n_filters * window_length], row-major   (ZAFX_LINEAR)      */
};

typedef struct zafx_params {
    int32_t struct_size;       /* = sizeof(zafx_params)                                         */
    int32_t window_length;     /* W, power of two, 64..8192 (STFT family, MDCT family, float32 MEL / MFCC), or --
                                  ZAFX_STFT / ZAFX_ISTFT / ZAFX_MEL / ZAFX_MFCC / ZAFX_MDCT / ZAFX_IMDCT -- any other length 33..8192 (MDCT family: even):
                                  those run as float32 Bluestein convolutions (np.fft takes any length, so does the reference).
                                  With ZAFX_PRECISION_F64 any length 2..2048 (MDCT family: even, 4..2048), every kind            */
    int32_t step_length;       /* hop H >= 1 (STFT/MEL/MFCC: any, also above W as zaf.stft allows; ISTFT: H <= W and
                                  any: above ceil(W/H) = 16 frames (8 at W = 4096, 4 at 8192) the float32 frames +
                                  gather overlap-add form runs instead of the tiled kernel); CQT: frame step        */
    int32_t layout;            /* enum zafx_layout of the 2-D (frequency x time) side           */
    int32_t n_filters;         /* MEL / MFCC: 1..576 (above 256: spectrum kernel + k_melfb; ZAFX_PRECISION_F64: 1..W/2) */
    int32_t n_coefs;           /* MFCC                                                          */
    int32_t fft_length;        /* CQT / CHROMA: power of two, 512..32768; 65536 when the kernel matrix touches the one-sided
                                  bins 1..8191 only (low-frequency kernels); ..131072 with ZAFX_PRECISION_F64 */
    int32_t n_bins;            /* CQT / CHROMA                                                  */
    int32_t octave_resolution; /* CHROMA                                                        */
    int32_t spectrum;          /* enum zafx_spectrum (STFT / ISTFT); 0 = reference contract     */
    int32_t precision;         /* enum zafx_precision (STFT / ISTFT); 0 = float32               */
    int32_t row_align;         /* ZAFX_LAYOUT_FT only: every row of the 2-D (F, T) array starts at a multiple of this many
                                  elements, i.e. the row pitch is T rounded up (zafx_plan_row_pitch).  0 / 1 = compact,
                                  the reference's own memory order.  16 (complex64) / 32 (float32) put every row on a
                                  128-byte line whatever T is -- the reference-layout STFT store runs at full rate only
                                  then (T = 433 compact: 2.1x slower than T = 432; padded: the same).  Power of two
                                  <= 1024.  The padding elements are never written (forward) nor used (inverse).     */
    int32_t transform_type;    /* ZAFX_DCT: 1, 2, 3 or 4 (dct_type / dst_type of zaf.py:703, :842)                    */
    int32_t transform_sine;    /* ZAFX_DCT: 0 = zaf.dct, 1 = zaf.dst                                                */
    int32_t with_mel;          /* ZAFX_MFCC, float32, window_length 2048, <= 128 filters, <= 32 coefficients: 1 = the melspectrogram of the SAME
                                  transforms rides along (zaf.melspectrogram and zaf.mfcc both start with zaf.stft, zaf.py:369 / :436; BASELINE
                                  config 3 in one pass: k_mel2 MODE 4).  Output = n_filters + n_coefs rows per clip: rows 0 .. n_filters - 1 the
                                  melspectrogram, the rest the MFCCs -- both bit-identical to the single-output plans' results           */
    int32_t reserved[1];
} zafx_params;

/* ---- library / device ------------------------------------------------------------ */
int zafx_version(void);
const char* zafx_last_error(void);
int zafx_device_count(int* count);
int zafx_device_name(int device, char* buf, size_t buflen);

/* ---- device memory (synchronous helpers; caller owns the allocations) -------------- */
#define ZAFX_ERROR_OUT_OF_MEMORY 2 /* zafx_alloc: the device has no room (= hipErrorOutOfMemory); other codes are other faults */
/* Arrays of 1 GiB and more are assembled from separate physical allocations of ZAFX_ALLOC_CHUNK_MB (default 64) MiB mapped back to back into one range
 * (HIP's virtual-memory API; 0 = plain hipMalloc): where a multi-GB array lies in physical memory moves the kernels that write it by up to 12 %, and on most
 * boxes the chunked form lands where hipMalloc's first allocation does not.  Pointers from zafx_alloc go back to zafx_free, not to hipFree. */
int zafx_alloc(int device, void** dptr, size_t bytes);
int zafx_free(int device, void* dptr);
/* The fastest of `n_candidates` allocations of `bytes` for the OUTPUT of `plan` (no reference counterpart).  Where a
 * multi-GB array lands in physical memory changes the rate of the kernels that write it with a row stride -- the reference layout's
 * (W, T) spectra of zaf.py:139: the same STFT runs in 1.51 ms into one 7.25 GB allocation and in 1.70 ms into another of the same
 * process -- and the address is not the caller's to choose, so a long-lived output buffer is picked by trial: all candidates are
 * allocated (held at once: a freed one would come straight back), zafx_execute(plan, d_in, candidate, n_clips, n_in) is timed on each
 * (one untimed launch, then `reps`; HIP events on the plan's stream), the best stays in *dptr, the others are freed.
 * probe_ms: NULL or n_candidates floats (the time per launch of each candidate).  bytes must hold the plan's output for (n_clips,
 * n_in).  Fewer candidates than asked are tried when the device runs out of memory (at least one).  The C twin of the Python
 * layer's DeviceBuffer.placed. */
int zafx_alloc_placed(zafx_plan* plan, void** dptr, size_t bytes, const void* d_in, int64_t n_clips, int64_t n_in, int n_candidates, int reps,
                      float* probe_ms);
int zafx_memset(int device, void* dptr, int value, size_t bytes);
int zafx_h2d(int device, void* dst, const void* src, size_t bytes);
int zafx_d2h(int device, void* dst, const void* src, size_t bytes);
int zafx_d2d(int device, void* dst, const void* src, size_t bytes);

/* Page-locked host memory for the transfers above: pageable NumPy buffers move at ~25 GB/s (and fault on first
 * touch), pinned ones at the PCIe rate (~56 GB/s measured).  Free with zafx_host_free. */
int zafx_host_alloc(void** hptr, size_t bytes);
int zafx_host_free(void* hptr);

/* ---- plans -------------------------------------------------------------------------- */
int zafx_plan_create(zafx_plan** plan, int device, int kind, const zafx_params* params);
int zafx_plan_destroy(zafx_plan* plan);
/* Upload one constant (copied; the host buffer may be released on return). */
int zafx_plan_set_constant(zafx_plan* plan, int which, const void* host, size_t bytes);
/* Output geometry for `n_in` (samples per clip for forward kinds, frames T for inverse
 * kinds): dims[0] = rows F (or samples L), dims[1] = frames T (or 1). */
int zafx_plan_out_dims(const zafx_plan* plan, int64_t n_in, int64_t dims[2]);
/* Elements between the starts of consecutive rows of the plan's 2-D array for `n_in` (as above): T for compact
 * plans, T rounded up to params.row_align otherwise; the array then holds clips x F x pitch elements.  For
 * ZAFX_LAYOUT_TF and ZAFX_LINEAR plans this is the (contiguous) row length itself. */
int zafx_plan_row_pitch(const zafx_plan* plan, int64_t n_in, int64_t* pitch);
/* Enqueue the transform of n_clips clips on the plan's stream (asynchronous). */
int zafx_execute(zafx_plan* plan, const void* d_in, void* d_out, int64_t n_clips, int64_t n_in);
int zafx_sync(zafx_plan* plan);
/* Bytes of ONE clip on the input and on the output side of the plan for `n_in` (as zafx_plan_out_dims; rows at the
 * plan's pitch): what a host array of n_clips clips must hold for zafx_run_host. */
int zafx_plan_clip_bytes(const zafx_plan* plan, int64_t n_in, int64_t* in_bytes, int64_t* out_bytes);
/* The host-array boundary of the reference (zaf.py:45: NumPy array in, NumPy array out) in one call: h_in -> HBM ->
 * transform -> h_out, in chunks of `chunk_clips` clips (0: chosen by the library, about 128 MB per chunk) through a
 * three-stage pipeline (upload stream, the plan's stream, download stream; two sets of plan-owned device staging buffers),
 * so that upload, kernel and download of neighbouring chunks overlap and the call runs at the rate of the slower PCIe
 * direction instead of the sum of the three.  Synchronous: returns when h_out
 * is complete.  Page-locked host arrays (zafx_host_alloc) transfer asynchronously at the PCIe rate; pageable ones work
 * and are staged by the runtime.  Serialise with other calls on the same plan. */
int zafx_run_host(zafx_plan* plan, const void* h_in, void* h_out, int64_t n_clips, int64_t n_in, int64_t chunk_clips);
/* HIP-event stopwatch on the plan's stream (the stream the kernels run on). */
int zafx_timer_start(zafx_plan* plan);
int zafx_timer_stop(zafx_plan* plan, float* elapsed_ms);
/* Name of the kernel family the plan was built for (fixed at zafx_plan_create / the last constant upload: the route a
 * geometry is expected to take). */
int zafx_plan_kernel_name(const zafx_plan* plan, char* buf, size_t buflen);
/* Name of the dominant kernel the LAST zafx_execute / zafx_run_host of this plan really launched (for matching rocprofv3
 * rows; the carry / band / generic forms are chosen per call from T, hop and the buffers' alignment); "" before the
 * first execute. */
int zafx_plan_last_kernel_name(const zafx_plan* plan, char* buf, size_t buflen);

/* Largest number of rows (n_bins) of a CQT kernel matrix that a float32 ZAFX_CQT / ZAFX_CHROMA plan of this fft_length
 * holds (k_cqt keeps the frame and the rows' bookkeeping in the 160 KB of LDS); 0 when fft_length itself is outside the
 * float32 kernels.  Larger kernels (and fft_length up to 131072) run as ZAFX_PRECISION_F64 plans. */
int zafx_cqt_max_bins(int fft_length, int* n_bins);

/* ---- PCM ingest: the step in front of the path ------------------------------ */
/* wavread's normalisation (zaf.py:1202: x / 2^(8*itemsize - 1)) and the channel mean every example
 * applies before the transforms (zaf.py:65: np.mean(audio_signal, 1)), on device:
 *   out[c][i] = mean_ch( in[c][i][ch] ) / 2^(8*sample_bytes - 1)
 * in: (n_clips, n_frames, n_channels) interleaved int16 (sample_bytes 2) or int32 (4); out: float32.
 * Enqueued on the plan's stream, so it is ordered before a following zafx_execute on that plan. */
int zafx_pcm_to_float(zafx_plan* plan, const void* d_pcm, void* d_out, int64_t n_clips, int64_t n_frames,
                      int n_channels, int sample_bytes);
/* The transform of integer PCM that is already on the device, in one call: d_pcm (n_clips, n_frames, n_channels) interleaved int16 / int32 ->
 * what zafx_execute writes for the normalised mono signal (zaf.py:1202 x / 2^(bits-1), zaf.py:65 mean over the channels, then the plan's
 * transform).  Plans whose kernel takes the integers in its own loads -- int16, one or two channels, into ZAFX_MEL / ZAFX_MFCC, ZAFX_STFT
 * (every spectrum kind) and ZAFX_MDCT at window_length 2048 in the reference layout -- read 2 bytes per sample and channel of HBM instead of 6 + 4; every other
 * plan converts into a float32 staging array it owns (zafx_pcm_to_float) and runs zafx_execute on that.  Kinds as zafx_run_host_pcm. */
int zafx_execute_pcm(zafx_plan* plan, const void* d_pcm, void* d_out, int64_t n_clips, int64_t n_frames, int n_channels, int sample_bytes);

/* zafx_run_host for integer PCM: h_pcm = (n_clips, n_frames, n_channels) interleaved int16 / int32 as wavread's source
 * holds them (zaf.py:1187-1204); every chunk crosses PCIe as integers (2 or 4 bytes per sample and channel instead of 4 per
 * float32 sample), is normalised and mixed down on the device in front of the transform, and the transform's result comes
 * back as from zafx_run_host.  Plans that take samples: STFT, MDCT, MEL, MFCC, CQT, CHROMA, DCT (float32). */
int zafx_run_host_pcm(zafx_plan* plan, const void* h_pcm, void* h_out, int64_t n_clips, int64_t n_frames, int n_channels,
                      int sample_bytes, int64_t chunk_clips);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI --------------------------------- */
/* The only collective on this path: broadcast of the shared constants (window,
 * filterbank, DCT matrix, CQT kernel) from `root`.  No reduction exists. */
int zafx_comm_unique_id(void* id128 /* 128 bytes out */);
int zafx_comm_create(zafx_comm** comm, int device, int rank, int n_ranks, const void* id128);
int zafx_comm_destroy(zafx_comm* comm);
/* Size of the communicator and this process's rank in it as RCCL reports them (ncclCommCount, ncclCommUserRank). */
int zafx_comm_count(zafx_comm* comm, int* n_ranks);
int zafx_comm_user_rank(zafx_comm* comm, int* rank);
int zafx_comm_broadcast_constants(zafx_comm* comm, zafx_plan* plan, int root);

#ifdef __cplusplus
}
#endif
#endif /* ZAFX_H */
