// CPU run of k_mel_ft8_f64's filterbank product over the host table of zafx_mel64.hpp -- "lanes" are loops.
// Build: g++ -O2 -std=c++17 -I zaf-python_amd/csrc tests/host_emu/mel64_emu.cpp -o mel64_emu
// stdin (binary): int32 n_filters, int32 cols, int32 spare, float64 fb[n_filters][cols], float64 S[cols]
// stdout (binary): int32 ok, steps, slots, max_parts; then (ok) float64 mel[n_filters]
#include <cstdio>
#include <vector>

#include "zafx_mel64.hpp"

int main() {
    int hdr[3];
    if (fread(hdr, sizeof(int), 3, stdin) != 3) return 2;
    const int nf = hdr[0], cols = hdr[1], spare = hdr[2];
    std::vector<double> fb((size_t)nf * cols), S(cols);
    if (fread(fb.data(), sizeof(double), fb.size(), stdin) != fb.size() || fread(S.data(), sizeof(double), S.size(), stdin) != S.size()) return 2;
    const zafx::Mel64Tables t = zafx::mel64_tables(fb.data(), nf, cols, spare);
    int out[4] = {t.ok, t.steps, t.slots, t.max_parts};
    fwrite(out, sizeof(int), 4, stdout);
    if (!t.ok) return 0;
    if (t.steps % 8) return 5;   // the kernel requests eight entries at a time
    std::vector<double> parts(t.slots, -1e300), mel(nf);
    for (int lane = 0; lane < 64; ++lane) {   // the kernel's loop, lane by lane
        double acc = 0.0;
        for (int s = 0; s < t.steps; ++s) {
            const zafx::Mel64Entry& e = t.stream[(size_t)s * 64 + lane];
            if (e.column < 0 || e.column >= cols) return 3;   // the kernel would read outside the spectrum
            acc += e.value * S[e.column];
            if (e.slot >= 0) {
                if (e.slot >= t.slots) return 4;
                parts[e.slot] = acc;
                acc = 0.0;
            }
        }
    }
    for (int m = 0; m < nf; ++m) {
        double s = 0.0;
        for (int p = 0; p < t.max_parts; ++p)
            if (p < t.fin[2 * m + 1]) s += parts[t.fin[2 * m] + p];
        mel[m] = s;
    }
    fwrite(mel.data(), sizeof(double), mel.size(), stdout);
    return 0;
}
