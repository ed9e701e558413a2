// zafx_cqt64.hpp -- host-side work tables of k_cqt_ft_f64 (zafx_f64.hip): which one-sided bins of a frame's spectrum the CQT kernel matrix
// reads, which thread computes them, and the matrix's non-zeros as one stream per thread.
// Plain C++ (no HIP): tests/host_emu/cqt64_emu.cpp runs the kernel's contraction over these tables on the CPU.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace zafx {

constexpr int kCq64Threads = 512;    // 8 wavefronts
constexpr int kCq64MaxBin = 8191;    // one-sided bins 1 .. 8191 (and their mirrors): a pair (c, N - c) of the real split then never has both ends in the set

struct Cqt64Entry {   // 24 bytes on the device as a double2 + an int2
    double re, im;    // K[row][column]
    int32_t index;    // compact index of the one-sided bin | conjugate << 31 (a column above W/2: X[c] = conj X[W - c])
    int32_t slot;     // >= 0: the running sum goes to this partial-sum slot after the entry, and restarts
};

struct Cqt64Tables {
    int n_cols = 0;                    // distinct one-sided bins, ascending; compact index = position
    std::vector<int> col_bin;          // [n_cols]
    int kc2 = 0;                       // bins per thread and round
    std::vector<int> split;            // [2 rounds][kc2][512]: bin | compact index << 14, or -1 (round 0: even bins, round 1: odd bins)
    int steps = 0, slots = 0, max_parts = 0;
    std::vector<Cqt64Entry> stream;    // [steps][512]
    std::vector<int> fin;              // [rows][2]: {first slot, slots}
    bool ok = false;
};

// CSR matrix (rows x W, W = 2 N) -> tables.  ok = false (the decimating kernel takes the plan) when a column's one-sided bin min(c, W - c)
// lies outside 1 .. kCq64MaxBin, when a thread would hold more than max_kc2 bins per round or when the matrix is empty.
inline Cqt64Tables cqt64_tables(const int32_t* indptr, const int32_t* indices, const double* values /* re, im pairs */, int rows, int W, int max_kc2,
                                int max_steps = 4096) {
    Cqt64Tables t;
    const int nnz = rows > 0 ? indptr[rows] : 0;
    if (rows < 1 || nnz < 1) return t;
    std::vector<int> compact((size_t)kCq64MaxBin + 1, -1);
    for (int e = 0; e < nnz; ++e) {
        const int c = indices[e];
        if (c < 0 || c >= W) return t;
        const int cc = std::min(c, W - c);
        if (cc < 1 || cc > kCq64MaxBin) return t;
        compact[(size_t)cc] = 0;
    }
    for (int cc = 1; cc <= kCq64MaxBin; ++cc)
        if (compact[(size_t)cc] == 0) {
            compact[(size_t)cc] = t.n_cols++;
            t.col_bin.push_back(cc);
        }
    // split: even bins in round 0, odd bins in round 1 (a bin's sub-transform is bin & 15: the even ones are transformed first), dealt to the threads in turn
    std::vector<int> lists[2];
    for (int cc : t.col_bin) lists[cc & 1].push_back(cc);
    t.kc2 = (int)std::max<size_t>(1, (std::max(lists[0].size(), lists[1].size()) + kCq64Threads - 1) / kCq64Threads);
    if (t.kc2 > max_kc2) return t;
    t.split.assign((size_t)2 * t.kc2 * kCq64Threads, -1);
    for (int r = 0; r < 2; ++r)
        for (size_t i = 0; i < lists[r].size(); ++i)
            t.split[((size_t)r * t.kc2 + i / kCq64Threads) * kCq64Threads + i % kCq64Threads] = lists[r][i] | (compact[(size_t)lists[r][i]] << 14);
    // contraction: the non-zeros in CSR order in equal consecutive shares
    const int share = (nnz + kCq64Threads - 1) / kCq64Threads;
    t.steps = (share + 7) / 8 * 8;   // (the kernel requests eight entries at a time)
    if (t.steps > max_steps) return t;
    t.stream.assign((size_t)t.steps * kCq64Threads, Cqt64Entry{0.0, 0.0, 0, -1});
    t.fin.assign((size_t)rows * 2, 0);
    int row = 0, slot = 0;
    for (int e = 0; e < nnz; ++e) {
        while (e >= indptr[row + 1]) ++row;
        const int thread = e / share, step = e % share;
        Cqt64Entry& en = t.stream[(size_t)step * kCq64Threads + thread];
        const int c = indices[e];
        en.re = values[2 * (size_t)e];
        en.im = values[2 * (size_t)e + 1];
        en.index = compact[(size_t)std::min(c, W - c)] | (c > W / 2 ? (int32_t)0x80000000u : 0);
        const bool ends = e + 1 == nnz || e + 1 >= indptr[row + 1] || (e + 1) % share == 0;
        if (ends) {
            en.slot = slot;
            int* f = &t.fin[(size_t)row * 2];
            if (f[1] == 0) f[0] = slot;
            ++f[1];
            t.max_parts = std::max(t.max_parts, f[1]);
            ++slot;
        }
    }
    t.slots = slot;
    t.ok = true;
    return t;
}

}  // namespace zafx
