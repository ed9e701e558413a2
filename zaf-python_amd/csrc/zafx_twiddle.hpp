// zafx_twiddle.hpp -- host-side (float64 -> float32) builders for every constant
// table the kernels read.  No trigonometry runs on the device.
#pragma once
#include <cmath>
#include <vector>

#include "zafx_fft.hpp"

namespace zafx {

struct cf32 { float re, im; };

// exp(sign * 2*pi*i * num/den), evaluated in float64 with exact octant reduction.
inline cf32 unit_root(long long num, long long den, int sign = -1) {
    num %= den;
    if (num < 0) num += den;
    const double a = 2.0 * M_PI * (double)num / (double)den;
    cf32 r;
    r.re = (float)std::cos(a);
    r.im = (float)(sign * std::sin(a));
    // exact values on the axes
    if (4 * num == den) { r.re = 0.f; r.im = (float)sign; }
    if (2 * num == den) { r.re = -1.f; r.im = 0.f; }
    if (4 * num == 3 * den) { r.re = 0.f; r.im = (float)-sign; }
    if (num == 0) { r.re = 1.f; r.im = 0.f; }
    return r;
}

// Per-pass [r-1][k] tables for fft_frame<LOG2N, LOG2E> (layout: zafx_fft.hpp).
inline std::vector<cf32> build_pass_twiddles(int log2n, int log2e) {
    std::vector<cf32> t((size_t)twiddle_total(log2n, log2e));
    int ns = 0;
    while (ns < log2n) {
        const int lr = pass_log2r(log2n - ns, log2e);
        if (ns > 0) {
            const int off = twiddle_offset(log2n, log2e, ns);
            const int NS = 1 << ns, R = 1 << lr;
            for (int r = 1; r < R; ++r)
                for (int k = 0; k < NS; ++k) t[(size_t)off + (size_t)(r - 1) * NS + k] = unit_root((long long)r * k, (long long)NS * R);
        }
        ns += lr;
    }
    return t;
}

// Two-level root table for fft_frame_chain: [hi: max(N/128, 1) entries | lo: 128 entries].
inline int two_level_hi_count(int log2n) { return log2n > 7 ? 1 << (log2n - 7) : 1; }
inline std::vector<cf32> build_two_level_twiddles(int log2n) {
    const long long n = 1LL << log2n;
    const int nh = two_level_hi_count(log2n);
    std::vector<cf32> t((size_t)nh + 128);
    for (int j = 0; j < nh; ++j) t[(size_t)j] = unit_root((long long)j << 7, n);
    for (int j = 0; j < 128; ++j) t[(size_t)nh + j] = unit_root(j, n);
    return t;
}

// Two-level table of the real-split twiddles of a length-2N real transform packed as N = 2^log2n complex
// points: exp(-2 pi i k / 2N) for k < N/2 as hi[k >> 7] * lo[k & 127]  ([hi: max(N/256, 1) | lo: 128]).
inline int split_hi_count(int log2n) { return log2n > 8 ? 1 << (log2n - 8) : 1; }
inline std::vector<cf32> build_split_two_level_twiddles(int log2n) {
    const long long w = 2LL << log2n;
    const int nh = split_hi_count(log2n);
    std::vector<cf32> t((size_t)nh + 128);
    for (int j = 0; j < nh; ++j) t[(size_t)j] = unit_root((long long)j << 7, w);
    for (int j = 0; j < 128; ++j) t[(size_t)nh + j] = unit_root(j, w);
    return t;
}

}  // namespace zafx
