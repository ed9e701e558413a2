// zafx_mel64.hpp -- host-side work table of k_mel_ft8_f64 (zafx_f64.hip): the mel filterbank's non-zeros as one stream per lane.
// Plain C++ (no HIP): tests/host_emu/mel64_emu.cpp runs the kernel's product loop over this table on the CPU.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

namespace zafx {

struct Mel64Entry {   // 16 bytes: one non-zero
    double value;
    int32_t column;   // of the spectrum S (bin column + 1)
    int32_t slot;     // >= 0: the running sum goes to this partial-sum slot after the entry, and restarts
};

struct Mel64Tables {
    std::vector<Mel64Entry> stream;   // [steps][64]: entry `step` of lane l
    std::vector<int> fin;             // [n_filters][2]: {first slot, slots} of a filter's partial sums (consecutive slots = ascending columns)
    int steps = 0, slots = 0, max_parts = 0;
    bool ok = false;
};

// fb: dense [n_filters][cols] float64.  The non-zeros in row-major order (a filter's band [first non-zero, last non-zero], zeros inside a
// band included) are dealt to the 64 lanes of a wavefront in equal consecutive shares of ceil(nnz / 64) entries -- every lane runs the
// same number of steps (rounded up to 8: the kernel requests them eight at a time), whatever the filters' lengths.  A lane's running sum
// is handed to a partial-sum slot where a filter ends inside its share and at the end of the share; a filter's slots are consecutive.
// ok = false when slots + n_filters exceed `spare` (the doubles of LDS behind the spectrum) or a lane would run more than `max_steps`.
inline Mel64Tables mel64_tables(const double* fb, int nf, int cols, int spare, int max_steps = 256) {
    Mel64Tables t;
    if (nf < 1 || cols < 1) return t;
    struct Nz { int row, col; };
    std::vector<Nz> nz;
    for (int r = 0; r < nf; ++r) {
        const double* row = fb + (size_t)r * cols;
        int a = 0, b = cols;
        while (a < cols && row[a] == 0.0) ++a;
        while (b > a && row[b - 1] == 0.0) --b;
        for (int c = a; c < b; ++c) nz.push_back({r, c});
    }
    const int share = std::max(1, ((int)nz.size() + 63) / 64);
    t.steps = (share + 7) / 8 * 8;
    if (t.steps > max_steps) return t;
    t.stream.assign((size_t)t.steps * 64, Mel64Entry{0.0, 0, -1});
    t.fin.assign((size_t)nf * 2, 0);
    int slot = 0;
    for (size_t q = 0; q < nz.size(); ++q) {
        const int lane = (int)(q / share), step = (int)(q % share);
        Mel64Entry& e = t.stream[(size_t)step * 64 + lane];
        e.value = fb[(size_t)nz[q].row * cols + nz[q].col];
        e.column = nz[q].col;
        const bool ends = q + 1 == nz.size() || nz[q + 1].row != nz[q].row || (q + 1) % share == 0;
        if (ends) {
            e.slot = slot;
            int* f = &t.fin[(size_t)nz[q].row * 2];
            if (f[1] == 0) f[0] = slot;
            ++f[1];
            t.max_parts = std::max(t.max_parts, f[1]);
            ++slot;
        }
    }
    t.slots = slot;
    if (t.slots + nf > spare) return t;
    t.ok = true;
    return t;
}

}  // namespace zafx
