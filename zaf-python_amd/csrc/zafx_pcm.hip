// zafx_pcm.hip -- PCM ingest: integer samples -> normalised float32 mono.
//
// The step in front of the hot path in every example of the reference:
//   audio_signal, fs = zaf.wavread(file)        zaf.py:1199-1204  (x / 2^(8*itemsize - 1))
//   audio_signal = np.mean(audio_signal, 1)     zaf.py:65
// HBM-bound elementwise kernel: 2-4 B per channel sample in, 4 B per frame out; grid-stride, one frame per thread and pass (scalar
// reads of its channels, a 4-byte store).  The kinds whose kernel takes int16 in its own loads (k_mel2, zafx_execute_pcm) do not come here.
#include "zafx_internal.hpp"

namespace zafx {

template <class S>
__global__ __launch_bounds__(256) void k_pcm_to_float(const S* __restrict__ pcm, float* __restrict__ out, long long n_total,
                                                      int n_channels, float scale) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_total; i += stride) {
        const S* src = pcm + i * n_channels;
        float acc = 0.f;
        for (int c = 0; c < n_channels; ++c) acc += (float)src[c] * scale;   // exact: |sample| * 2^-k is a float
        out[i] = acc / (float)n_channels;
    }
}

hipError_t launch_pcm_to_float(hipStream_t stream, const void* pcm, float* out, int64_t n_total, int n_channels, int sample_bytes) {
    if (n_total <= 0) return hipSuccess;
    const int blocks = (int)std::min<int64_t>((n_total + 255) / 256, 256 * 8);
    if (sample_bytes == 2)
        hipLaunchKernelGGL(k_pcm_to_float<int16_t>, dim3(blocks), dim3(256), 0, stream, (const int16_t*)pcm, out, (long long)n_total,
                           n_channels, 1.f / 32768.f);
    else
        hipLaunchKernelGGL(k_pcm_to_float<int32_t>, dim3(blocks), dim3(256), 0, stream, (const int32_t*)pcm, out, (long long)n_total,
                           n_channels, 1.f / 2147483648.f);
    return hipGetLastError();
}

}  // namespace zafx
