// CPU emulation of the LDS Stockham FFT core (zafx_fft.hpp) -- "threads" are loops.
// Build: g++ -O2 -std=c++17 -DZAFX_HOST_EMU -I zaf-python_amd/csrc tests/host_emu/fft_emu.cpp -o fft_emu
// Prints max normwise error vs a float64 naive DFT for every (log2n, log2e) the kernels use.
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "zafx_twiddle.hpp"

using namespace zafx;

template <int LOG2N, int LOG2E, int LOG2NS>
struct Runner {
    static void run(std::vector<float2>& regs, std::vector<float2>& buf, const float2* tw) {
        using C = FftCfg<LOG2N, LOG2E>;
        if constexpr (LOG2NS < LOG2N) {
            constexpr int LR = pass_log2r(LOG2N - LOG2NS, LOG2E);
            for (int p = 0; p < C::P; ++p)
                pass_write<LOG2N, LOG2E, LOG2NS, LR>(&regs[(size_t)p * C::E], buf.data(), p,
                                                      tw + twiddle_offset(LOG2N, LOG2E, LOG2NS));
            if constexpr (LOG2NS + LR < LOG2N) {
                for (int p = 0; p < C::P; ++p) regs_read<LOG2N, LOG2E>(&regs[(size_t)p * C::E], buf.data(), p);
                Runner<LOG2N, LOG2E, LOG2NS + LR>::run(regs, buf, tw);
            }
        }
    }
};

template <int LOG2N, int LOG2E>
double check() {
    using C = FftCfg<LOG2N, LOG2E>;
    std::vector<std::complex<double>> x(C::N), ref(C::N);
    srand(LOG2N * 131 + LOG2E);
    for (auto& v : x) v = {rand() / (double)RAND_MAX - 0.5, rand() / (double)RAND_MAX - 0.5};
    for (int k = 0; k < C::N; ++k) {
        std::complex<double> s = 0;
        for (int n = 0; n < C::N; ++n) s += x[n] * std::polar(1.0, -2.0 * M_PI * (double)((long long)n * k % C::N) / C::N);
        ref[k] = s;
    }
    auto twv = build_pass_twiddles(LOG2N, LOG2E);
    std::vector<float2> tw(twv.size() + 1);
    for (size_t i = 0; i < twv.size(); ++i) tw[i] = make_float2(twv[i].re, twv[i].im);
    std::vector<float2> regs((size_t)C::N), buf((size_t)C::PITCH);
    for (int p = 0; p < C::P; ++p)
        for (int i = 0; i < C::E; ++i) regs[(size_t)p * C::E + i] = make_float2((float)x[p + i * C::P].real(), (float)x[p + i * C::P].imag());
    Runner<LOG2N, LOG2E, 0>::run(regs, buf, tw.data());
    double err = 0, mx = 0;
    for (int k = 0; k < C::N; ++k) {
        std::complex<double> got(buf[phys_t<C::PS>(k)].x, buf[phys_t<C::PS>(k)].y);
        err = std::max(err, std::abs(got - ref[k]));
        mx = std::max(mx, std::abs(ref[k]));
    }
    printf("log2n=%d log2e=%d P=%d tw=%d relerr=%.3e\n", LOG2N, LOG2E, C::P, C::TW, err / mx);
    return err / mx;
}

int main() {
    double worst = 0;
    worst = std::max(worst, check<4, 1>());
    worst = std::max(worst, check<5, 1>());
    worst = std::max(worst, check<6, 1>());
    worst = std::max(worst, check<7, 1>());
    worst = std::max(worst, check<8, 2>());
    worst = std::max(worst, check<9, 3>());
    worst = std::max(worst, check<10, 4>());
    worst = std::max(worst, check<10, 5>());   // two radix-32 passes
    worst = std::max(worst, check<9, 5>());    // 32 x 16
    worst = std::max(worst, check<11, 4>());
    worst = std::max(worst, check<12, 4>());
    worst = std::max(worst, check<14, 4>());
    printf("worst=%.3e\n", worst);
    return worst < 2e-6 ? 0 : 1;
}
