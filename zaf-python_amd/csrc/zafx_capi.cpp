// zafx_capi.cpp -- the C-ABI of libzafx.so (include/zafx.h): plans, constants,
// device memory helpers, HIP-event timing and the RCCL broadcast of constants.
// Host C++ only; every kernel lives in the *.hip translation units.
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <utility>

#include "zafx_internal.hpp"

#ifndef ZAFX_STFT_FAT8_TABLES
#define ZAFX_STFT_FAT8_TABLES 0   // pass tables of the W = 4096 persistent STFT experiment (zafx_stft.hip, ZAFX_STFT_FAT8)
#endif

namespace zafx {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }

hipError_t ensure_dynamic_lds(const void* kernel, int device, size_t bytes) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, size_t> granted;
    std::lock_guard<std::mutex> lk(mu);
    size_t& have = granted[std::make_pair(kernel, device)];
    if (have >= bytes) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) have = bytes;
    return e;
}
namespace { thread_local int t_pcm_mode = 0; thread_local bool t_pcm_taken = false; }
int take_pcm_mode() {
    if (t_pcm_mode) t_pcm_taken = true;
    return t_pcm_mode;
}
void set_pcm_mode(int mode) { t_pcm_mode = mode, t_pcm_taken = false; }
bool pcm_mode_taken() { return t_pcm_taken; }


static int fail(const std::string& where, hipError_t e) {
    set_error(where + ": " + hipGetErrorString(e));
    return (int)e ? (int)e : -1;
}
static int fail_msg(const std::string& msg, int code = -1) {
    set_error(msg);
    return code;
}

#define ZAFX_HIP(call)                                    \
    do {                                                  \
        hipError_t e__ = (call);                          \
        if (e__ != hipSuccess) return fail(#call, e__);   \
    } while (0)

static int ilog2_exact(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

template <class T>
static hipError_t upload(T** dptr, const void* host, size_t bytes) {
    if (*dptr) {
        hipError_t e = hipFree(*dptr);
        *dptr = nullptr;
        if (e != hipSuccess) return e;
    }
    if (bytes == 0) return hipSuccess;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(dptr), bytes);
    if (e != hipSuccess) return e;
    return hipMemcpy(*dptr, host, bytes, hipMemcpyHostToDevice);
}

// ---------------------------------------------------------------------------------
// MFMA-fragment packing of a banded (rows x cols) matrix: per 16-row block keep only
// the column range that holds non-zeros (rounded to multiples of 4 = K of
// v_mfma_f32_16x16x4_f32) and lay each 16x4 step out in the A-operand lane order
// (lane l holds A[l & 15][l >> 4]).
// ---------------------------------------------------------------------------------
static hipError_t pack_band(PackedBand& pb, const float* dense, int n_rows, int n_cols, int n_waves) {
    pb.n_rows = n_rows;
    pb.n_cols = n_cols;
    pb.n_blocks = (n_rows + 15) / 16;
    pb.n_waves = n_waves;
    struct Blk { int first, steps, off; };
    std::vector<Blk> blks((size_t)pb.n_blocks);
    std::vector<float> pack;
    pb.total_steps = 0;
    for (int b = 0; b < pb.n_blocks; ++b) {
        int lo = n_cols, hi = -1;
        for (int r = 16 * b; r < std::min(16 * b + 16, n_rows); ++r)
            for (int c = 0; c < n_cols; ++c)
                if (dense[(size_t)r * n_cols + c] != 0.f) {
                    lo = std::min(lo, c);
                    hi = std::max(hi, c);
                }
        int first = 0, steps = 0;
        if (hi >= lo) {
            first = lo & ~3;
            steps = (hi - first) / 4 + 1;
        }
        blks[(size_t)b] = Blk{first, steps, pb.total_steps};
        for (int s = 0; s < steps; ++s)
            for (int l = 0; l < 64; ++l) {
                const int r = 16 * b + (l & 15), c = first + 4 * s + (l >> 4);
                pack.push_back((r < n_rows && c < n_cols) ? dense[(size_t)r * n_cols + c] : 0.f);
            }
        pb.total_steps += steps;
    }
    // Deal the K-steps to the wavefronts in equal contiguous shares of the (block-major) step sequence; a share that crosses
    // a block boundary becomes one work item per block (slot ids are (block, part)-ordered).  Every wave gets
    // ceil(total / n_waves) steps or one less: 128 filters at W = 2048 are 271 steps = 17 per wave (cutting the blocks into
    // parts dealt longest-first left the busiest wave with 22).
    struct Item { int slot, first, steps, off; };
    std::vector<int> blk_ptr((size_t)pb.n_blocks + 1, 0);
    std::vector<std::vector<Item>> per_wave((size_t)n_waves);
    {
        int b = 0;
        for (int w = 0; w < n_waves; ++w) {
            int g0 = (int)((long long)pb.total_steps * w / n_waves);
            const int g1 = (int)((long long)pb.total_steps * (w + 1) / n_waves);
            while (g0 < g1) {
                while (blks[(size_t)b].off + blks[(size_t)b].steps <= g0) ++b;   // the block that holds step g0 (empty blocks: below)
                const Blk& k = blks[(size_t)b];
                const int s0 = g0 - k.off, s1 = std::min(g1 - k.off, k.steps);
                per_wave[(size_t)w].push_back(Item{-1, k.first + 4 * s0, s1 - s0, k.off + s0});
                per_wave[(size_t)w].back().slot = -1 - b;   // block, resolved to a slot id below
                g0 = k.off + s1;
            }
        }
        // slot ids in (block, part) order; a block without non-zeros gets one empty item so that its rows are written (as zeros)
        std::vector<std::vector<Item*>> by_block((size_t)pb.n_blocks);
        for (auto& v : per_wave)
            for (Item& it : v) by_block[(size_t)(-1 - it.slot)].push_back(&it);
        int next = 0;
        std::vector<std::pair<int, Item>> empties;   // (appended afterwards: a push_back would move the items pointed to)
        for (int bb = 0; bb < pb.n_blocks; ++bb) {
            blk_ptr[(size_t)bb] = next;
            if (by_block[(size_t)bb].empty()) {
                empties.emplace_back(bb % n_waves, Item{next++, 0, 0, 0});
                continue;
            }
            for (Item* it : by_block[(size_t)bb]) it->slot = next++;
        }
        for (const auto& e : empties) per_wave[(size_t)e.first].push_back(e.second);
        blk_ptr[(size_t)pb.n_blocks] = next;
        pb.n_items = next;
        pb.n_empty = (int)empties.size();
    }
    // k_mel2's items (see PackedBand::d_whole)
    pb.whole_ok = false;
    if (n_waves == 16) {
        struct Part { int blk, first, steps, off, role, slot; };
        std::vector<Part> parts;
        for (int b = 0; b < pb.n_blocks; ++b) parts.push_back(Part{b, blks[(size_t)b].first, blks[(size_t)b].steps, blks[(size_t)b].off, 1, 0});
        int helpers = 0;
        std::vector<int> addmask((size_t)pb.n_blocks, 0);
        while (helpers < kMel2Slots) {   // the longest part in two while it is more than a SIMD's quarter
            auto big = std::max_element(parts.begin(), parts.end(), [](const Part& a, const Part& b) { return a.steps < b.steps; });
            if (big == parts.end() || big->steps * 4 <= pb.total_steps || big->steps < 8) break;
            const int h = big->steps / 2;
            Part rest{big->blk, big->first + 4 * (big->steps - h), h, big->off + (big->steps - h), 2, helpers};
            big->steps -= h;
            addmask[(size_t)rest.blk] |= 1 << helpers;
            ++helpers;
            parts.push_back(rest);
        }
        std::stable_sort(parts.begin(), parts.end(), [](const Part& a, const Part& b) { return a.steps > b.steps; });
        std::vector<std::vector<Part>> simd(4);
        int load[4] = {0, 0, 0, 0};
        bool fits = true;
        for (const Part& q : parts) {
            int best = -1;
            for (int i = 0; i < 4; ++i)
                if (simd[(size_t)i].size() < 8 && (best < 0 || load[i] < load[best])) best = i;
            if (best < 0) { fits = false; break; }
            simd[(size_t)best].push_back(q);
            load[best] += q.steps + 2;   // (+ an item's fixed cost)
        }
        if (fits) {
            std::vector<int> whole((size_t)16 * 2 * 4, 0);
            for (size_t i = 0; i < whole.size(); i += 4) whole[i + 1] = -1;
            for (int sd = 0; sd < 4; ++sd)
                for (size_t j = 0; j < simd[(size_t)sd].size(); ++j) {   // first round: one item per wave (the longest to the oldest), second round: the short ones
                    const Part& q = simd[(size_t)sd][j];
                    const int wave = sd + 4 * (int)(j % 4), place = (int)(j / 4);
                    int* m = &whole[(size_t)((wave * 2 + place) * 4)];
                    m[0] = q.first;
                    m[1] = q.steps;
                    m[2] = q.off;
                    m[3] = q.blk | q.role << 8 | q.slot << 10 | (q.role == 1 ? addmask[(size_t)q.blk] : 0) << 12;
                }
            if (hipError_t e = upload(&pb.d_whole, whole.data(), whole.size() * sizeof(int)); e != hipSuccess) return e;
            pb.h_owner.assign((size_t)pb.n_blocks, 0);
            for (int w = 0; w < 16; ++w)
                for (int place = 0; place < 2; ++place) {
                    const int* m = &whole[(size_t)((w * 2 + place) * 4)];
                    if (m[1] >= 0 && ((m[3] >> 8) & 3) == 1) pb.h_owner[(size_t)(m[3] & 255)] = w | place << 8;
                }
            pb.whole_ok = true;
        }
    }
    // per K-step descriptors (resident form of k_mel: 16 bits per step in scalar registers instead of the item walk):
    // first column / 4 | slot id << 8 | item ends << 15
    std::vector<unsigned short> desc((size_t)pb.total_steps + 2, 0);
    pb.desc_ok = true;
    for (const auto& v : per_wave)
        for (const Item& it : v)
            for (int s = 0; s < it.steps; ++s) {
                const int col4 = it.first / 4 + s;
                if (col4 > 255 || it.slot > 127) pb.desc_ok = false;
                desc[(size_t)(it.off + s)] = (unsigned short)((col4 & 255) | ((it.slot & 127) << 8) | (s == it.steps - 1 ? 1 << 15 : 0));
            }
    std::vector<int> flat, wave_ptr((size_t)n_waves + 1, 0);
    for (int w = 0; w < n_waves; ++w) {
        wave_ptr[(size_t)w] = (int)flat.size() / 4;
        for (const Item& it : per_wave[(size_t)w]) {
            flat.push_back(it.slot); flat.push_back(it.first); flat.push_back(it.steps); flat.push_back(it.off);
        }
    }
    wave_ptr[(size_t)n_waves] = (int)flat.size() / 4;
    pb.max_wave_steps = (pb.total_steps + n_waves - 1) / n_waves;
    if (pack.empty()) pack.push_back(0.f);
    pack.resize(pack.size() + 64 * 2 * (size_t)std::max(kMelResidentFb, kMelResidentDct), 0.f);   // spare steps: k_mel reads a fixed number of fragments per wave
    if (flat.empty()) flat.assign(4, 0);
    hipError_t e = upload(&pb.d_pack, pack.data(), pack.size() * sizeof(float));
    if (e == hipSuccess) e = upload(&pb.d_items, flat.data(), flat.size() * sizeof(int));
    if (e == hipSuccess) e = upload(&pb.d_wave_ptr, wave_ptr.data(), wave_ptr.size() * sizeof(int));
    if (e == hipSuccess) e = upload(&pb.d_blk_ptr, blk_ptr.data(), blk_ptr.size() * sizeof(int));
    if (e == hipSuccess) e = upload(&pb.d_desc, desc.data(), desc.size() * sizeof(unsigned short));
    return e;
}

static void free_band(PackedBand& pb) {
    if (pb.d_pack) (void)hipFree(pb.d_pack);
    if (pb.d_items) (void)hipFree(pb.d_items);
    if (pb.d_wave_ptr) (void)hipFree(pb.d_wave_ptr);
    if (pb.d_blk_ptr) (void)hipFree(pb.d_blk_ptr);
    if (pb.d_desc) (void)hipFree(pb.d_desc);
    if (pb.d_direct) (void)hipFree(pb.d_direct);
    if (pb.d_whole) (void)hipFree(pb.d_whole);
    if (pb.d_dct2) (void)hipFree(pb.d_dct2);
    if (pb.d_owner2) (void)hipFree(pb.d_owner2);
    pb = PackedBand{};
}

// k_cqt's view of the CQT kernel matrix: rows sorted by length, four per step, steps dealt to the wavefronts.
static int build_cqt_chunks(zafx_plan* pl);
static int build_cqt_mm(zafx_plan* pl, int n_waves);

// Bluestein tables of a W-point DFT as a convolution of length M = 2^log2m, in long double: the chirp
// c[n] = exp(-i pi n^2 / W) (angle reduced in integers: n^2 mod 2W) and Bhat = FFT_M of conj(c) wrapped to length M.
// `den` = W for a DFT; the chirp-z sums of zafx_bs32.hip's k_dct_bs32 (a step of pi / D per n k instead of 2 pi / W) pass den = 2 D
// and count = the vector length.
typedef std::complex<long double> cld;
static void bluestein_tables(int count, long long den, int log2m, std::vector<cld>& chirp, std::vector<cld>& bhat) {
    const int M = 1 << log2m;
    const long double pi = 3.14159265358979323846264338327950288L;
    const long long W = den;
    chirp.assign((size_t)count, cld(0, 0));
    bhat.assign((size_t)M, cld(0, 0));
    for (long long k = 0; k < count; ++k) {
        const long long r = (k * k) % (2LL * W);
        const long double ang = pi * (long double)r / (long double)W;
        cld v(cosl(ang), sinl(ang));   // conj(c[k]) = exp(+i pi k^2 / W)
        if (r == 0) v = cld(1, 0);
        if (r == W) v = cld(-1, 0);
        if (2 * r == W) v = cld(0, 1);
        if (2 * r == 3LL * W) v = cld(0, -1);
        chirp[(size_t)k] = std::conj(v);
        bhat[(size_t)k] = v;
        if (k) bhat[(size_t)(M - k)] = v;
    }
    for (int i = 1, j = 0; i < M; ++i) {   // bit reversal, then radix-2 passes
        int bit = M >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(bhat[(size_t)i], bhat[(size_t)j]);
    }
    for (int len = 2; len <= M; len <<= 1)
        for (int i = 0; i < M; i += len)
            for (int k = 0; k < len / 2; ++k) {
                const long double ang = -2.0L * pi * (long double)k / (long double)len;
                const cld w(cosl(ang), sinl(ang));
                const cld u = bhat[(size_t)(i + k)], v = bhat[(size_t)(i + k + len / 2)] * w;
                bhat[(size_t)(i + k)] = u + v;
                bhat[(size_t)(i + k + len / 2)] = u - v;
            }
}

static bool is_stft_family(int kind) { return kind == ZAFX_STFT || kind == ZAFX_ISTFT || kind == ZAFX_MEL || kind == ZAFX_MFCC; }
static bool is_mdct_family(int kind) { return kind == ZAFX_MDCT || kind == ZAFX_IMDCT; }
static bool is_cqt_family(int kind) { return kind == ZAFX_CQT || kind == ZAFX_CHROMA; }

// (re)build everything that is derived from the host shadows of the constants
// k_mel2's mfcc stage (PackedBand::d_dct2 of pl->dct): needs the filterbank's whole-block items and the DCT rows, whichever constant comes last
static hipError_t build_mel2_dct(zafx_plan* pl) {
    zafx::PackedBand& d = pl->dct;
    d.dct2_ok = false;
    const int nf = pl->prm.n_filters, nc = pl->prm.n_coefs, nb = pl->fb.n_blocks;
    if (pl->kind != ZAFX_MFCC || !pl->fb.whole_ok || pl->h_dct.size() != (size_t)nc * nf || nc > 32 || nb > 8 || (int)pl->fb.h_owner.size() != nb) return hipSuccess;
    std::vector<float> frag((size_t)nb * 2 * 4 * 64, 0.f);
    for (int b = 0; b < nb; ++b)
        for (int c = 0; c < 2; ++c)
            for (int st = 0; st < 4; ++st)
                for (int l = 0; l < 64; ++l) {
                    const int row = 16 * c + (l & 15), col = 16 * b + 4 * st + (l >> 4);
                    if (row < nc && col < nf) frag[(((size_t)b * 2 + c) * 4 + st) * 64 + l] = pl->h_dct[(size_t)row * nf + col];
                }
    if (hipError_t e = upload(&d.d_dct2, frag.data(), frag.size() * sizeof(float)); e != hipSuccess) return e;
    if (hipError_t e = upload(&d.d_owner2, pl->fb.h_owner.data(), pl->fb.h_owner.size() * sizeof(int)); e != hipSuccess) return e;
    d.dct2_ok = true;
    return hipSuccess;
}

static int finalize_constant(zafx_plan* pl, int which) {
    switch (which) {
        case ZAFX_CONST_WINDOW: {
            if (pl->prm.precision == ZAFX_PRECISION_F64) {
                ZAFX_HIP(upload(&pl->d_window64, pl->h_window64.data(), pl->h_window64.size() * sizeof(double)));
                long double g = 0;   // zaf.py:241  sum(window_function[0:W:H])
                for (int i = 0; i < pl->W; i += pl->H) g += (long double)pl->h_window64[(size_t)i];
                pl->cola_gain64 = (double)g;
                pl->cola_gain = (float)g;
                return 0;
            }
            ZAFX_HIP(upload(&pl->d_window, pl->h_window.data(), pl->h_window.size() * sizeof(float)));
            double g = 0;   // zaf.py:241  sum(window_function[0:W:H])
            if (pl->H > 0)
                for (int i = 0; i < pl->W; i += pl->H) g += (double)pl->h_window[(size_t)i];
            pl->cola_gain = (float)g;
            if (is_mdct_family(pl->kind)) {
                // sign-folded window for the fold + pack step of k_mdct_ft32: packed input m reads taps
                // (a, b | c, d); re = x[a] w0 + x[b] w1, im = x[c] w2 + x[d] w3
                const int nf = pl->W / 4;
                std::vector<float> wf((size_t)nf * 4);
                const float* w = pl->h_window.data();
                for (int m = 0; m < nf; ++m) {
                    float* o = &wf[(size_t)m * 4];
                    if (2 * m < nf) {
                        o[0] = -w[3 * nf - 1 - 2 * m]; o[1] = -w[3 * nf + 2 * m];
                        o[2] = w[nf - 1 - 2 * m];      o[3] = -w[nf + 2 * m];
                    } else {
                        o[0] = w[2 * m - nf];          o[1] = -w[3 * nf - 1 - 2 * m];
                        o[2] = -w[nf + 2 * m];         o[3] = -w[5 * nf - 1 - 2 * m];
                    }
                }
                ZAFX_HIP(upload(&pl->d_wfold, wf.data(), wf.size() * sizeof(float)));
            }
            return 0;
        }
        case ZAFX_CONST_MEL_FB:
            if (pl->prm.precision == ZAFX_PRECISION_F64) {   // rows as bands [first non-zero, last non-zero]
                const int rows = pl->prm.n_filters, cols = pl->W / 2;
                std::vector<int> meta((size_t)rows * 3, 0);
                std::vector<double> vals;
                for (int r = 0; r < rows; ++r) {
                    const double* row = &pl->h_fb64[(size_t)r * cols];
                    int lo = 0, hi = cols;
                    while (lo < cols && row[lo] == 0.0) ++lo;
                    while (hi > lo && row[hi - 1] == 0.0) --hi;
                    meta[(size_t)r * 3] = lo;
                    meta[(size_t)r * 3 + 1] = hi - lo;
                    meta[(size_t)r * 3 + 2] = (int)vals.size();
                    vals.insert(vals.end(), row + lo, row + hi);
                }
                if (vals.empty()) vals.push_back(0.0);
                ZAFX_HIP(upload(&pl->d_fb64, vals.data(), vals.size() * sizeof(double)));
                ZAFX_HIP(upload(&pl->d_fb64_meta, meta.data(), meta.size() * sizeof(int)));
                ZAFX_HIP(zafx::build_mel64_fb(*pl));
                return 0;
            }
            if (zafx::mel_takes_wide_route(*pl)) {   // W = 4096 / 8192, windows that are not a power of two, more than 256 filters (k_melfb): rows as float32 bands
                const int rows = pl->prm.n_filters, cols = pl->W / 2;
                std::vector<int> meta((size_t)rows * 3, 0);
                std::vector<float> vals;
                for (int r = 0; r < rows; ++r) {
                    const float* row = &pl->h_fb[(size_t)r * cols];
                    int lo = 0, hi = cols;
                    while (lo < cols && row[lo] == 0.f) ++lo;
                    while (hi > lo && row[hi - 1] == 0.f) --hi;
                    meta[(size_t)r * 3] = lo;
                    meta[(size_t)r * 3 + 1] = hi - lo;
                    meta[(size_t)r * 3 + 2] = (int)vals.size();
                    vals.insert(vals.end(), row + lo, row + hi);
                }
                if (vals.empty()) vals.push_back(0.f);
                ZAFX_HIP(upload(&pl->d_fbw, vals.data(), vals.size() * sizeof(float)));
                ZAFX_HIP(upload(&pl->d_fbw_meta, meta.data(), meta.size() * sizeof(int)));
                return 0;
            }
            {
                // k_mel hands the filterbank 2 |X| (mel) or 4 |X|^2 (mfcc) -- the real split without its two halvings -- so the packed
                // fragments carry the factor 1/2 or 1/4: powers of two, the products and sums come out bit for bit as before.
                const float scale = pl->kind == ZAFX_MFCC ? 0.25f : 0.5f;
                std::vector<float> scaled(pl->h_fb.size());
                for (size_t i = 0; i < scaled.size(); ++i) scaled[i] = pl->h_fb[i] * scale;
                ZAFX_HIP(pack_band(pl->fb, scaled.data(), pl->prm.n_filters, pl->W / 2, mel_waves(pl->log2nf)));
                ZAFX_HIP(build_mel2_dct(pl));
            }
            return 0;
        case ZAFX_CONST_DCT:
            if (pl->prm.precision == ZAFX_PRECISION_F64) {
                ZAFX_HIP(upload(&pl->d_dct64, pl->h_dct64.data(), pl->h_dct64.size() * sizeof(double)));
                ZAFX_HIP(zafx::build_mel64_dct(*pl));
                return 0;
            }
            if (zafx::mel_takes_wide_route(*pl)) {   // (k_melfb): dense rows
                ZAFX_HIP(upload(&pl->d_dctw, pl->h_dct.data(), pl->h_dct.size() * sizeof(float)));
                return 0;
            }
            ZAFX_HIP(pack_band(pl->dct, pl->h_dct.data(), pl->prm.n_coefs, pl->prm.n_filters, mel_waves(pl->log2nf)));
            {
                // Register-fed form (k_mel, 16 waves): after the filterbank's reduction wave w holds, lane for lane, the B fragment of
                // the DCT's K-steps w, w + 16, ... (log-mel rows 4 s + (lane >> 4), frame lane & 15), so its A fragments are those
                // steps of every 16-row block of DCT rows: [w][j][block][lane] = D[16 block + (lane & 15)][4 (w + 16 j) + (lane >> 4)].
                zafx::PackedBand& d = pl->dct;
                const int nw = mel_waves(pl->log2nf), nf = pl->prm.n_filters, nc = pl->prm.n_coefs;
                const int ksteps = (nf + 3) / 4, jn = (ksteps + nw - 1) / nw;
                d.direct_j = 0;
                if (nw == 16 && jn <= 2 && d.n_blocks <= 2) {   // a fixed 2 x 2 arrangement [w][j][block] (zero fragments where there is no step / block)
                    std::vector<float> frag((size_t)nw * 4 * 64, 0.f);
                    for (int w = 0; w < nw; ++w)
                        for (int j = 0; j < jn; ++j)
                            for (int b = 0; b < d.n_blocks; ++b)
                                for (int l = 0; l < 64; ++l) {
                                    const int row = 16 * b + (l & 15), col = 4 * (w + nw * j) + (l >> 4);
                                    if (row < nc && col < nf) frag[(((size_t)w * 2 + j) * 2 + b) * 64 + l] = pl->h_dct[(size_t)row * nf + col];
                                }
                    ZAFX_HIP(upload(&d.d_direct, frag.data(), frag.size() * sizeof(float)));
                    d.direct_j = 2;
                }
            }
            ZAFX_HIP(build_mel2_dct(pl));
            return 0;
        case ZAFX_CONST_MATRIX:
            ZAFX_HIP(upload(&pl->d_matrix, pl->h_matrix.data(), pl->h_matrix.size() * sizeof(float)));
            return 0;
        case ZAFX_CONST_CQT_INDPTR:
            ZAFX_HIP(upload(&pl->d_indptr, pl->h_indptr.data(), pl->h_indptr.size() * sizeof(int32_t)));
            pl->cqt_dirty = true;
            pl->cqt64_dirty = true;
            return 0;
        case ZAFX_CONST_CQT_INDICES:
            ZAFX_HIP(upload(&pl->d_indices, pl->h_indices.data(), pl->h_indices.size() * sizeof(int32_t)));
            pl->nnz = (int)pl->h_indices.size();
            pl->cqt_dirty = true;
            pl->cqt64_dirty = true;
            return 0;
        case ZAFX_CONST_CQT_VALUES:
            if (pl->prm.precision == ZAFX_PRECISION_F64) {
                ZAFX_HIP(upload(&pl->d_values64, pl->h_values64.data(), pl->h_values64.size() * sizeof(double2)));
                pl->cqt64_dirty = true;
                return 0;
            }
            ZAFX_HIP(upload(&pl->d_values, pl->h_values.data(), pl->h_values.size() * sizeof(cf32)));
            return 0;
    }
    return fail_msg("unknown constant id");
}

static int build_cqt_chunks(zafx_plan* pl) {
    // k_cqt's view of the CSR matrix (zafx_cqt.hip): rows sorted by length, four rows per step (one per 16-lane DPP
    // row; lane j of a row takes its entries j, j + 16, ...), steps dealt to the wavefronts, longest first, always
    // to the least loaded wave.
    const int n_waves = std::max(1, cqt_waves(pl->log2nf));
    const int n_rows = (int)pl->h_indptr.size() - 1;
    const int n = pl->W / 2, w = pl->W;
    if (n_rows > cqt_max_bins(pl->log2nf)) return fail_msg("cqt: kernel matrix has too many rows for LDS at this fft_length");
    std::vector<int> order((size_t)n_rows);
    for (int r = 0; r < n_rows; ++r) {
        order[(size_t)r] = r;
        if (pl->h_indptr[(size_t)r + 1] < pl->h_indptr[(size_t)r]) return fail_msg("CQT kernel indptr is not monotone");
    }
    auto nnz_of = [&](int r) { return pl->h_indptr[(size_t)r + 1] - pl->h_indptr[(size_t)r]; };
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return nnz_of(a) > nnz_of(b); });
    // shapes: a long row takes the whole wave (64 lanes), medium rows half of it, short rows a DPP row each -- steps then
    // stay a few iterations long and balance over the waves (config Q: 18 iterations on the busiest wave with four-row
    // steps only, 12 with the wide shapes)
    struct Step { int lanes; int row_of_group[4]; int iters; };   // row_of_group[g]: row whose entries DPP row g works on (-1: none)
    std::vector<Step> steps;
    for (int i = 0; i < n_rows;) {
        const int m = nnz_of(order[(size_t)i]);
        const int lanes = m > 128 ? 64 : m > 64 ? 32 : 16, per = 64 / lanes;
        Step st{lanes, {-1, -1, -1, -1}, 1};
        for (int q = 0; q < per && i < n_rows; ++q, ++i) {
            const int r = order[(size_t)i];
            for (int g = q * (lanes / 16); g < (q + 1) * (lanes / 16); ++g) st.row_of_group[g] = r;
            st.iters = std::max(st.iters, (nnz_of(r) + lanes - 1) / lanes);
        }
        steps.push_back(st);
    }
    // numerically real matrix (the reference's kernels: the temporal kernels are centred, zaf.py:540-544)
    float vmax = 0.f, imax = 0.f;
    for (const cf32& v : pl->h_values) {
        vmax = std::max(vmax, std::max(std::fabs(v.re), std::fabs(v.im)));
        imax = std::max(imax, std::fabs(v.im));
    }
    pl->cqt_real = imax <= vmax * 9.3e-10f;   // 2^-30: far below the float32 rounding of the products
    // deal the steps to the waves, longest first, always to the least loaded wave
    std::vector<int> by_len((size_t)steps.size());
    for (size_t i = 0; i < steps.size(); ++i) by_len[i] = (int)i;
    std::stable_sort(by_len.begin(), by_len.end(), [&](int a, int b) { return steps[(size_t)a].iters > steps[(size_t)b].iters; });
    std::vector<std::vector<int>> per_wave((size_t)n_waves);
    std::vector<int> load((size_t)n_waves, 0);
    for (int s : by_len) {
        const int wv = (int)(std::min_element(load.begin(), load.end()) - load.begin());
        per_wave[(size_t)wv].push_back(s);
        load[(size_t)wv] += steps[(size_t)s].iters + 4;   // (a step end costs about four iterations: reductions + the tile write)
    }
    // flatten: steps in wave order, entries as [iteration][lane]
    std::vector<int> wave_tab;
    std::vector<int32_t> addrs;
    std::vector<float> vals;
    int k_lo = n, k_hi = -1, special = 0, max_iters = 0;
    for (int wv = 0; wv < n_waves; ++wv) {
        const int it0 = (int)addrs.size() / 64;
        unsigned mask = 0;
        int it = 0;
        for (int s : per_wave[(size_t)wv]) {
            const Step& st = steps[(size_t)s];
            const int gl = st.lanes / 16;   // DPP rows per matrix row
            const int shape = st.lanes == 16 ? 1 : st.lanes == 32 ? 2 : 3;
            for (int i = 0; i < st.iters; ++i)
                for (int l = 0; l < 64; ++l) {
                    const int g = l >> 4, row = st.row_of_group[g], j = (l & (st.lanes - 1)) + st.lanes * i;
                    int32_t word = 0x7ff << 18;
                    cf32 v{0.f, 0.f};
                    if (row >= 0 && j < nnz_of(row)) {
                        const size_t e = (size_t)pl->h_indptr[(size_t)row] + (size_t)j;
                        const int c = pl->h_indices[e];
                        if (c < 0 || c >= w) return fail_msg("CQT kernel column index out of range");
                        const int m = c <= n ? c : w - c;   // one-sided bin holding X[c] (conjugated when c > n)
                        if (cqt_double(pl->log2nf) && (m < 1 || m > kCqtDoubleMaxBin))
                            return fail_msg("cqt: at fft_length 65536 the float32 kernel takes matrices whose columns lie in bins 1 .. 8191 (and their mirrors); use ZAFX_PRECISION_F64");
                        const int slot = m == n ? cqt_nyquist_slot(pl->log2nf) : cqt_slot(pl->log2nf, m);
                        word |= (int32_t)(slot * 8) | (c > n ? (int32_t)0x80000000 : 0);
                        v = pl->h_values[e];
                        const int k = std::min(m, n - m);   // pair index of the real split
                        if (k == 0 || 2 * m == n) special = 1;
                        else k_lo = std::min(k_lo, k), k_hi = std::max(k_hi, k);
                    }
                    if (i == st.iters - 1) {   // the step ends here: shape in every lane, the row in the last lane of its lane group
                        word |= shape << 29;
                        if ((l & 15) == 15 && g % gl == gl - 1 && row >= 0) word = (word & ~(0x7ff << 18)) | (row << 18);
                    }
                    addrs.push_back(word);
                    vals.push_back(v.re);
                    if (!pl->cqt_real) vals.push_back(v.im);
                }
            it += st.iters;
            if (it <= 32) mask |= 1u << (it - 1);
        }
        max_iters = std::max(max_iters, it);
        wave_tab.insert(wave_tab.end(), {it0, it, (int)mask, 0});
    }
    pl->cqt_resident = max_iters <= kCqtResident ? kCqtResident : 0;
    pl->cqt_n_entries = (int)addrs.size();
    pl->cqt_k_lo = k_lo;
    pl->cqt_k_hi = k_hi;
    pl->cqt_k_special = special;
    if (addrs.empty()) addrs.assign(64, 0), vals.assign(pl->cqt_real ? 64 : 128, 0.f);
    ZAFX_HIP(upload(&pl->d_cqt_addrs, addrs.data(), addrs.size() * sizeof(int32_t)));
    ZAFX_HIP(upload(&pl->d_cqt_vals, vals.data(), vals.size() * sizeof(float)));
    ZAFX_HIP(upload(&pl->d_cqt_waves, wave_tab.data(), wave_tab.size() * sizeof(int)));
    if (int rc = build_cqt_mm(pl, n_waves)) return rc;
    pl->cqt_dirty = false;
    return 0;
}

// The matrix-core form of the contraction (zafx_cqt.hip): v_mfma_f32_4x4x1_16b_f32 multiplies, in each of its sixteen 4-lane
// blocks, four A values with four B values and ACCUMULATES over successive instructions -- the sum over a row's columns needs
// no cross-lane reduction.  Rows go in pairs (2p, 2p + 1: neighbours of a constant-Q kernel overlap by three quarters); the
// sorted union of a pair's columns is a stream, cut into segments of S entries; a block works on two segments at once (A lanes
// 0, 1: the two rows of stream a, lanes 2, 3: of stream b; B lanes 0, 1: re, im of stream a's bin, lanes 2, 3: of stream b's), so
// a wavefront's instruction advances 32 segments by one column.  Segments of a pair are consecutive stream slots
// (slot = 2 (16 wave + block) + stream): the kernel's finishing pass adds them up from there.  Real matrices whose columns all
// lie in the lower half (no conjugated bins) and whose S fits the registers; anything else stays on the lane-reduction form.
#ifndef ZAFX_CQT_ROTATE
#define ZAFX_CQT_ROTATE 1
#endif
static int build_cqt_mm(zafx_plan* pl, int n_waves) {
    pl->cqt_mm_steps = 0;
    if (!pl->cqt_real || cqt_double(pl->log2nf)) return 0;
    const int n_rows = (int)pl->h_indptr.size() - 1, n = pl->W / 2, n_pairs = (n_rows + 1) / 2;
    if (n_rows < 1) return 0;
    std::vector<std::vector<int>> cols((size_t)n_pairs);
    for (int r = 0; r < n_rows; ++r)
        for (int e = pl->h_indptr[(size_t)r]; e < pl->h_indptr[(size_t)r + 1]; ++e) {
            const int c = pl->h_indices[(size_t)e];
            if (c < 0 || c > n) return 0;   // (a conjugated bin: the lane-reduction form handles it)
            cols[(size_t)(r >> 1)].push_back(c);
        }
    size_t total = 0;
    for (auto& c : cols) {
        std::sort(c.begin(), c.end());
        c.erase(std::unique(c.begin(), c.end()), c.end());
        total += c.size();
    }
    const int slots = 32 * n_waves;
    int S = (int)std::max<size_t>(1, (total + slots - 1) / slots);
    auto segments = [&](int s) {
        long long k = 0;
        for (auto& c : cols) k += ((long long)c.size() + s - 1) / s;
        return k;
    };
    while (S <= kCqtMmSteps && segments(S) > slots) ++S;
    if (S > kCqtMmSteps) return 0;
    std::vector<float> vals((size_t)n_waves * S * 64, 0.f);
    std::vector<int32_t> addr((size_t)n_waves * S * 64, 0);   // (padding steps: 0 x bin 0)
    std::vector<int32_t> fin((size_t)n_pairs, 0);
    // Segments in slot order (a pair's segments are consecutive slots: the finishing pass adds them up from fin[pair]).
    struct Seg { int pair, first, len; };
    std::vector<Seg> segs;
    pl->cqt_mm_segs = 0;
    for (int p = 0; p < n_pairs; ++p) {
        const int size = (int)cols[(size_t)p].size(), nseg = (size + S - 1) / S;
        pl->cqt_mm_segs = std::max(pl->cqt_mm_segs, nseg);
        fin[(size_t)p] = (int)segs.size() | nseg << 16;
        for (int g = 0; g < nseg; ++g) segs.push_back({p, g * S, std::min(S, size - g * S)});
    }
    auto slot_of_col = [&](int col) { return col == n ? cqt_nyquist_slot(pl->log2nf) : cqt_slot(pl->log2nf, col); };
    // The 32 segments of a wavefront read one bin each per step (re, im: two neighbouring banks = one of 16 bank pairs), and the sum over a
    // segment's columns does not care for their order or for the steps they take.  Which column goes to which step is an edge colouring of the
    // bipartite graph segments x bank pairs (an edge per column, a colour per step): with a bank pair split into ceil(degree / S) copies
    // every node has at most S edges, so S colours suffice (Koenig) and no step has more segments on a bank pair than its copies -- 2 where
    // the wave's columns spread evenly.  Sorted columns as they come: 4.1 segments on the worst bank pair of a step; coloured: 2.4
    // (1024 clips x 30 s: 23.77 -> 22.9 ms; the contraction's gathers were the LDS's busiest stretch).
    for (int w0 = 0; w0 < (int)segs.size(); w0 += 32) {
        const int nl = std::min<int>(32, (int)segs.size() - w0), wave = w0 >> 5;
        struct Edge { int u, v, col; };
        std::vector<Edge> edges;
        int deg[16] = {}, seen[16] = {};
        for (int j = 0; j < nl; ++j)
            for (int q = 0; q < segs[(size_t)(w0 + j)].len; ++q) ++deg[slot_of_col(cols[(size_t)segs[(size_t)(w0 + j)].pair][(size_t)(segs[(size_t)(w0 + j)].first + q)]) & 15];
        int copy0[17] = {};   // first right-hand node of each bank pair
        for (int b = 0; b < 16; ++b) copy0[b + 1] = copy0[b] + std::max(1, (deg[b] + S - 1) / S);
        for (int j = 0; j < nl; ++j) {
            const Seg& sg = segs[(size_t)(w0 + j)];
            for (int q = 0; q < sg.len; ++q) {
                const int col = cols[(size_t)sg.pair][(size_t)(sg.first + q)], b = slot_of_col(col) & 15;
                edges.push_back({j, copy0[b] + seen[b]++ / S, col});
            }
        }
        // (the overflow copies first: their few edges take the lowest colours together, so the steps that carry a third segment coincide)
        if (ZAFX_CQT_ROTATE) std::stable_sort(edges.begin(), edges.end(), [&](const Edge& a, const Edge& b) {
            auto rank = [&](const Edge& e) { int b2 = 0; while (copy0[b2 + 1] <= e.v) ++b2; return e.v - copy0[b2]; };
            return rank(a) > rank(b);
        });
        const int nr = copy0[16];
        std::vector<int> L((size_t)nl * S, -1), R((size_t)nr * S, -1), colour(edges.size(), -1);
        for (int e = 0; e < (int)edges.size(); ++e) {
            const int u = edges[(size_t)e].u, v = edges[(size_t)e].v;
            int a = 0, b = 0;
            while (ZAFX_CQT_ROTATE && L[(size_t)u * S + a] >= 0) ++a;
            while (ZAFX_CQT_ROTATE && R[(size_t)v * S + b] >= 0) ++b;
            if (!ZAFX_CQT_ROTATE) {   // (experiment switch: sorted columns at consecutive steps, as before round 5)
                while (L[(size_t)u * S + a] >= 0) ++a;
                b = a;
            }
            if (a != b) {   // the a / b alternating path from v: swap its colours, then a is free at both ends
                std::vector<int> path;
                int side = 1, node = v, cc = a;
                for (;;) {
                    const int e2 = side ? R[(size_t)node * S + cc] : L[(size_t)node * S + cc];
                    if (e2 < 0) break;
                    path.push_back(e2);
                    node = side ? edges[(size_t)e2].u : edges[(size_t)e2].v;
                    side ^= 1;
                    cc = cc == a ? b : a;
                }
                for (int e2 : path) L[(size_t)edges[(size_t)e2].u * S + colour[(size_t)e2]] = R[(size_t)edges[(size_t)e2].v * S + colour[(size_t)e2]] = -1;
                for (int e2 : path) {
                    const int c1 = colour[(size_t)e2] == a ? b : a;
                    colour[(size_t)e2] = c1;
                    L[(size_t)edges[(size_t)e2].u * S + c1] = R[(size_t)edges[(size_t)e2].v * S + c1] = e2;
                }
            }
            colour[(size_t)e] = a;
            L[(size_t)u * S + a] = e;
            if (ZAFX_CQT_ROTATE) R[(size_t)v * S + a] = e;
        }
        for (int e = 0; e < (int)edges.size(); ++e) {
            const int slot = w0 + edges[(size_t)e].u, blk = (slot >> 1) & 15, stream = slot & 1, i = colour[(size_t)e], col = edges[(size_t)e].col;
            const int p = segs[(size_t)slot].pair, lds = slot_of_col(col) * 8;
            const size_t base = ((size_t)wave * S + i) * 64 + blk * 4 + stream * 2;
            for (int m = 0; m < 2; ++m) {
                const int r = 2 * p + m;
                addr[base + m] = lds + 4 * m;   // B lane: re / im
                if (r >= n_rows) continue;
                const auto b = pl->h_indices.begin() + pl->h_indptr[(size_t)r], en = pl->h_indices.begin() + pl->h_indptr[(size_t)r + 1];
                for (auto it = b; it != en; ++it)   // (duplicate column entries of a row add up, as in a CSR product)
                    if (*it == col) vals[base + m] += pl->h_values[(size_t)(it - pl->h_indices.begin())].re;
            }
        }
    }
    ZAFX_HIP(upload(&pl->d_cqt_mm_vals, vals.data(), vals.size() * sizeof(float)));
    ZAFX_HIP(upload(&pl->d_cqt_mm_addr, addr.data(), addr.size() * sizeof(int32_t)));
    ZAFX_HIP(upload(&pl->d_cqt_mm_fin, fin.data(), fin.size() * sizeof(int32_t)));
    pl->cqt_mm_steps = S;
    return 0;
}

}  // namespace zafx

using namespace zafx;

extern "C" {

int zafx_version(void) { return ZAFX_VERSION; }
const char* zafx_last_error(void) { return g_err.c_str(); }

int zafx_device_count(int* count) {
    if (!count) return fail_msg("null argument");
    ZAFX_HIP(hipGetDeviceCount(count));
    return 0;
}

int zafx_device_name(int device, char* buf, size_t buflen) {
    if (!buf || !buflen) return fail_msg("null argument");
    hipDeviceProp_t prop;
    ZAFX_HIP(hipGetDeviceProperties(&prop, device));
    std::snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return 0;
}

// ---------------------------------------------------------------------------------
// Large device arrays: physical memory in CHUNKS, mapped back to back into one reserved range (HIP's virtual-memory API).
// ---------------------------------------------------------------------------------
// Where hipMalloc puts the 7.25 GB spectrum of BASELINE config 2 moves k_stft_ft16 between 1.50 and 1.70 ms (six rounds of "placement").
// Round 6 (tools/placement_vmm.py; bench.py --kind stft in fresh processes on a dozen boxes): the same array built from separate physical allocations of
// 32 ... 512 MiB lands at 1.49-1.55 ms on most boxes where hipMalloc's lands at 1.69-1.70 (four boxes, chunks of 64 MiB | 512 MiB | hipMalloc:
// 1.52 | 1.55 | 1.70, 1.49 | 1.53 | 1.69, 1.52 | 1.52 | 1.52, 1.69 | 1.69 | 1.69; ONE 8-GiB physical allocation: as hipMalloc) -- never worse, not always
// better: what decides is where in physical memory the array lies, and separate allocations are drawn from elsewhere.  zafx_alloc therefore
// assembles every array of kVmmMin bytes or more from pieces of ZAFX_ALLOC_CHUNK_MB (64) MiB; smaller ones, and everything when ZAFX_ALLOC_CHUNK_MB=0 or the API is not there,
// come from hipMalloc.  zafx_free unmaps and releases; copies, memsets and kernels see one contiguous range either way.
namespace {
constexpr size_t kVmmMin = 1ull << 30;
std::mutex g_vmm_mu;
std::map<void*, size_t> g_vmm;   // base -> bytes mapped

size_t vmm_chunk_bytes() {
    static const size_t chunk = [] {
        const char* s = std::getenv("ZAFX_ALLOC_CHUNK_MB");
        const long mb = s && *s ? std::atol(s) : 64;
        return mb > 0 ? (size_t)mb << 20 : (size_t)0;
    }();
    return chunk;
}

void vmm_release(void* base, size_t mapped, size_t reserved) {
    if (mapped) (void)hipMemUnmap(base, mapped);
    (void)hipMemAddressFree(base, reserved);
}

hipError_t vmm_alloc(int device, void** out, size_t bytes) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t g = 0;
    if (hipError_t e = hipMemGetAllocationGranularity(&g, &prop, hipMemAllocationGranularityRecommended); e != hipSuccess || g == 0) return e != hipSuccess ? e : hipErrorNotSupported;
    const size_t chunk = (vmm_chunk_bytes() + g - 1) / g * g, total = (bytes + g - 1) / g * g;
    void* base = nullptr;
    if (hipError_t e = hipMemAddressReserve(&base, total, 0, nullptr, 0); e != hipSuccess) return e;
    size_t off = 0;
    while (off < total) {
        const size_t n = std::min(chunk, total - off);
        hipMemGenericAllocationHandle_t h;
        hipError_t e = hipMemCreate(&h, n, &prop, 0);
        if (e == hipSuccess) {
            e = hipMemMap(static_cast<char*>(base) + off, n, 0, h, 0);
            (void)hipMemRelease(h);   // (the mapping keeps the memory; it goes with the unmap)
        }
        if (e != hipSuccess) {
            vmm_release(base, off, total);
            return e;
        }
        off += n;
    }
    hipMemAccessDesc access = {};
    access.location.type = hipMemLocationTypeDevice;
    access.location.id = device;
    access.flags = hipMemAccessFlagsProtReadWrite;
    if (hipError_t e = hipMemSetAccess(base, total, &access, 1); e != hipSuccess) {
        vmm_release(base, total, total);
        return e;
    }
    {
        std::lock_guard<std::mutex> lock(g_vmm_mu);
        g_vmm[base] = total;
    }
    *out = base;
    return hipSuccess;
}

// the library's device allocator (zafx_alloc, zafx_alloc_placed)
hipError_t device_alloc(int device, void** out, size_t bytes) {
    if (bytes >= kVmmMin && vmm_chunk_bytes() > 0) {
        const hipError_t e = vmm_alloc(device, out, bytes);
        if (e == hipSuccess || e == hipErrorOutOfMemory) return e;
        (void)hipGetLastError();   // (no virtual-memory API on this stack: the plain allocator)
    }
    return hipMalloc(out, bytes ? bytes : 1);
}

hipError_t device_free(void* p) {
    size_t mapped = 0;
    {
        std::lock_guard<std::mutex> lock(g_vmm_mu);
        const auto it = g_vmm.find(p);
        if (it != g_vmm.end()) {
            mapped = it->second;
            g_vmm.erase(it);
        }
    }
    if (!mapped) return hipFree(p);
    if (hipError_t e = hipDeviceSynchronize(); e != hipSuccess) return e;   // (as hipFree does: no kernel may still use the range)
    if (hipError_t e = hipMemUnmap(p, mapped); e != hipSuccess) return e;
    return hipMemAddressFree(p, mapped);
}
}  // namespace

int zafx_alloc(int device, void** dptr, size_t bytes) {
    if (!dptr) return fail_msg("null argument");
    ZAFX_HIP(hipSetDevice(device));
    ZAFX_HIP(device_alloc(device, dptr, bytes));
    return 0;
}
int zafx_free(int device, void* dptr) {
    ZAFX_HIP(hipSetDevice(device));
    ZAFX_HIP(device_free(dptr));
    return 0;
}
int zafx_memset(int device, void* dptr, int value, size_t bytes) {
    ZAFX_HIP(hipSetDevice(device));
    ZAFX_HIP(hipMemset(dptr, value, bytes));
    return 0;
}
int zafx_h2d(int device, void* dst, const void* src, size_t bytes) {
    ZAFX_HIP(hipSetDevice(device));
    ZAFX_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return 0;
}
int zafx_d2h(int device, void* dst, const void* src, size_t bytes) {
    ZAFX_HIP(hipSetDevice(device));
    ZAFX_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return 0;
}
int zafx_d2d(int device, void* dst, const void* src, size_t bytes) {
    ZAFX_HIP(hipSetDevice(device));
    ZAFX_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice));
    return 0;
}

int zafx_host_alloc(void** hptr, size_t bytes) {
    if (!hptr) return fail_msg("null argument");
    *hptr = nullptr;
    if (bytes == 0) return 0;
    ZAFX_HIP(hipHostMalloc(hptr, bytes, hipHostMallocDefault));
    return 0;
}
int zafx_host_free(void* hptr) {
    if (hptr) ZAFX_HIP(hipHostFree(hptr));
    return 0;
}

int zafx_plan_create(zafx_plan** out, int device, int kind, const zafx_params* params) {
    if (!out || !params) return fail_msg("null argument");
    if (params->struct_size != (int32_t)sizeof(zafx_params)) return fail_msg("zafx_params.struct_size mismatch");
    if (params->layout != ZAFX_LAYOUT_FT && params->layout != ZAFX_LAYOUT_TF) return fail_msg("bad layout");
    if (params->spectrum < ZAFX_SPECTRUM_TWO_SIDED || params->spectrum > ZAFX_SPECTRUM_POWER) return fail_msg("bad spectrum");
    if (params->spectrum != ZAFX_SPECTRUM_TWO_SIDED && kind != ZAFX_STFT && kind != ZAFX_ISTFT)
        return fail_msg("spectrum applies to ZAFX_STFT / ZAFX_ISTFT only");
    if (params->spectrum >= ZAFX_SPECTRUM_MAGNITUDE && kind != ZAFX_STFT)
        return fail_msg("magnitude / power spectra are outputs of ZAFX_STFT only");
    if (params->precision != ZAFX_PRECISION_F32 && params->precision != ZAFX_PRECISION_F64) return fail_msg("bad precision");
    if (params->precision == ZAFX_PRECISION_F64 && (kind == ZAFX_LINEAR || kind == ZAFX_DCT))
        return fail_msg("ZAFX_PRECISION_F64 is not available for ZAFX_LINEAR / ZAFX_DCT");
    if (params->row_align < 0 || params->row_align > 1024 || (params->row_align & (params->row_align - 1)))
        return fail_msg("row_align must be 0 or a power of two <= 1024");
    if (params->row_align > 1 && (params->layout != ZAFX_LAYOUT_FT || kind == ZAFX_LINEAR || kind == ZAFX_DCT))
        return fail_msg("row_align applies to the 2-D arrays of ZAFX_LAYOUT_FT plans only");
    {
        int n_dev = 0;
        ZAFX_HIP(hipGetDeviceCount(&n_dev));
        if (device < 0 || device >= n_dev) return fail_msg("device index out of range");
    }
    zafx_plan* pl = new zafx_plan();
    pl->device = device;
    pl->kind = kind;
    pl->prm = *params;
    pl->layout = params->layout;
    auto bail = [&](const std::string& m) {
        delete pl;
        return fail_msg(m);
    };
    std::vector<cf32> aux;
    if (is_stft_family(kind)) {
        pl->W = params->window_length;
        pl->H = params->step_length;
        int lw = ilog2_exact(pl->W);
        if ((lw < 0 || lw < 6) && params->precision == ZAFX_PRECISION_F64 && pl->W >= 2 && pl->W <= 2048) {
            // any length up to 2048 in the float64 mode: Bluestein convolution of length 2^bs_log2m >= 2 W - 1
            while ((1 << pl->bs_log2m) < 2 * pl->W - 1) ++pl->bs_log2m;
            lw = 6;   // (only sizes the unused float32 tables below)
        } else if (lw < 0 && params->precision == ZAFX_PRECISION_F32 && bs32_supported(pl->W)) {
            // float32 Bluestein forms (zafx_bs32.hip); MEL / MFCC: the Bluestein STFT's magnitude / power kind + k_melfb (run_mel_wide)
            while ((1 << pl->bs_log2m) < 2 * pl->W - 1) ++pl->bs_log2m;
            lw = 6;
        } else if (lw < 0 || !stft_supported(lw - 1)) {
            return bail("window_length must be a power of two in [64, 8192] or any other length in [33, 8192] "
                        "(any length in [2, 2048] with ZAFX_PRECISION_F64)");
        }
        if (pl->H < 1) return bail("step_length must be >= 1");
        if (kind == ZAFX_ISTFT && pl->H > pl->W) return bail("istft: step_length must not exceed window_length");
        if (pl->H > (1 << 20)) return bail("step_length must not exceed 2^20");
        if (kind == ZAFX_ISTFT && params->precision == ZAFX_PRECISION_F32 && pl->bs_log2m == 0 && bs32_supported(pl->W)) {
            // A hop so small that more frames cover a sample than the tiled overlap-add keeps in LDS (16): the frames + gather
            // overlap-add form of zafx_bs32.hip has no such limit and takes any length, powers of two included.
            const int tile = pl->W <= 2048 ? 16 : pl->W == 4096 ? 8 : 4;   // frames of the tiled kernel's LDS tile
            if ((pl->W + pl->H - 1) / pl->H > tile)
                while ((1 << pl->bs_log2m) < 2 * pl->W - 1) ++pl->bs_log2m;
        }
        pl->log2nf = lw - 1;
        if (kind == ZAFX_MEL || kind == ZAFX_MFCC) {
            // (float32: up to 256 rows on the fused kernels; above that the spectrum kernel + k_melfb route, whose mfcc form holds a
            // [n_filters][64] log-mel tile in LDS: 576 rows)
            if (params->n_filters < 1 || (params->precision != ZAFX_PRECISION_F64 && params->n_filters > 576))
                return bail("n_filters must be in [1, 576] (up to window_length / 2 with ZAFX_PRECISION_F64)");
            if (params->with_mel != 0 && (params->with_mel != 1 || kind != ZAFX_MFCC || params->precision != ZAFX_PRECISION_F32 || params->window_length != 2048 ||
                                          params->n_filters > 128 || params->n_coefs > 32))
                return bail("with_mel (melspectrogram + mfcc in one pass) takes a float32 ZAFX_MFCC plan of window_length 2048, up to 128 filters and up to 32 coefficients");
            if (kind == ZAFX_MFCC && (params->n_coefs < 1 || params->n_coefs > params->n_filters))
                return bail("n_coefs must be in [1, n_filters]");
            if (params->precision == ZAFX_PRECISION_F64) {
                if (params->n_filters > std::max(pl->W / 2, 1)) return bail("n_filters must not exceed window_length / 2");
            } else if (lw - 1 < 5 || lw - 1 > 12) {
                return bail("float32 mel/mfcc kernels are built for window_length 64 ... 8192 (any power of two with ZAFX_PRECISION_F64)");
            }
        }
        const int n = pl->W / 2;
        aux.resize((size_t)n / 2 + 1);
        for (int k = 0; k <= n / 2; ++k) aux[(size_t)k] = unit_root(k, pl->W);
        pl->kernel_name = kind == ZAFX_STFT ? stft_kernel_name(lw - 1, pl->layout) : kind == ZAFX_ISTFT ? istft_kernel_name(lw - 1, pl->layout) : params->n_filters > 256 ? mel_wide_kernel_name() : lw - 1 == 11 ? "k_mel_ft16b" : lw - 1 >= 12 ? mel_wide_kernel_name() : mel_kernel_name();
    } else if (is_mdct_family(kind)) {
        pl->W = params->window_length;
        pl->H = pl->W / 2;   // zaf.py:1029
        int lw = ilog2_exact(pl->W);
        if ((lw < 0 || lw < 6) && params->precision == ZAFX_PRECISION_F64 && pl->W >= 4 && pl->W <= 2048 && pl->W % 2 == 0) {
            while ((1 << pl->bs_log2m) < 2 * pl->W - 1) ++pl->bs_log2m;   // any even length in the float64 mode (Bluestein)
            lw = 6;
        } else if (lw < 0 && params->precision == ZAFX_PRECISION_F32 && pl->W % 2 == 0 && bs32_supported(pl->W)) {
            while ((1 << pl->bs_log2m) < 2 * pl->W - 1) ++pl->bs_log2m;   // float32 Bluestein forms (zafx_bs32.hip)
            lw = 6;
        } else if (lw < 0 || !mdct_supported(lw - 2)) {
            return bail("window_length must be a power of two in [64, 8192] or any even length in [34, 8192] "
                        "(any even length in [4, 2048] with ZAFX_PRECISION_F64)");
        }
        pl->log2nf = lw - 2;
        const int nf = pl->W / 4, m = pl->W / 2;
        aux.resize((size_t)nf);
        for (int i = 0; i < nf; ++i) aux[(size_t)i] = unit_root(8LL * i + 1, 16LL * m);
        pl->kernel_name = kind == ZAFX_MDCT ? mdct_kernel_name(lw - 2, pl->layout) : imdct_kernel_name();
    } else if (is_cqt_family(kind)) {
        pl->W = params->fft_length;
        pl->H = params->step_length;
        const int lw = ilog2_exact(pl->W);
        if (params->precision == ZAFX_PRECISION_F64) {   // k_cqt_f64 decimates the frame: no LDS limit on its length
            if (lw < 9 || lw > 17) return bail("fft_length must be a power of two in [512, 131072] (ZAFX_PRECISION_F64 plan)");
        } else if (lw < 0 || !cqt_supported(lw - 1)) {
            return bail("fft_length must be a power of two in [512, 65536] (up to 131072 with ZAFX_PRECISION_F64)");
        }
        if (pl->H < 1) return bail("step_length must be >= 1");
        if (params->n_bins < 1 || params->n_bins > 1024) return bail("n_bins must be in [1, 1024]");
        if (kind == ZAFX_CHROMA && (params->octave_resolution < 1 || params->octave_resolution > params->n_bins))
            return bail("octave_resolution must be in [1, n_bins]");
        pl->log2nf = lw - 1;
        aux = build_split_two_level_twiddles(pl->log2nf);   // exp(-2 pi i k / W), k < W/4, as hi[k >> 7] * lo[k & 127]
        pl->kernel_name = cqt_kernel_name();
    } else if (kind == ZAFX_LINEAR) {
        pl->W = params->window_length;
        if (pl->W < 1 || pl->W > 16384 || params->n_filters < 1 || params->n_filters > 16384)
            return bail("linear map: window_length (columns) and n_filters (rows) must be in [1, 16384]");
        pl->log2nf = 4;   // only so that the (unused) FFT tables below are small
        aux.assign(1, cf32{1.f, 0.f});
        pl->kernel_name = linear_kernel_name();
    } else if (kind == ZAFX_DCT) {
        // zaf.dct / zaf.dst on the FFT core (zafx_dct.hip): M complex points per vector, tables A | B of M + 1 entries each
        const int N = params->window_length, type = params->transform_type;
        const bool sine = params->transform_sine != 0;
        if (type < 1 || type > 4) return bail("transform_type must be 1, 2, 3 or 4");
        if (params->transform_sine != 0 && params->transform_sine != 1) return bail("transform_sine must be 0 (dct) or 1 (dst)");
        if (N < 2) return bail("dct / dst: window_length must be at least 2");
        const int M = type == 1 ? (sine ? N + 1 : N - 1) : (N % 2 == 0 ? N / 2 : -1);
        const int lm = ilog2_exact(M);
        pl->W = N;
        if (lm >= 0 && dct_supported(lm)) {
            pl->log2nf = lm;
            aux.resize(2 * (size_t)(M + 1));
            for (int k = 0; k <= M; ++k) {
                aux[(size_t)k] = type == 4 ? unit_root(4LL * k + 1, 8LL * N) : unit_root(k, 2LL * M);
                aux[(size_t)(M + 1 + k)] = type == 4 ? unit_root(k, 2LL * N) : unit_root(k, 4LL * N);
            }
            pl->kernel_name = dct_kernel_name();
        } else if (type >= 2 && N % 4 == 0 && N <= 8192) {
            // N / 2 points, not a power of two (round 6): k_dct's maps around a Bluestein convolution of 2^bs_log2m >= 2 (N / 2) - 1 points --
            // half the transform length of the chirp-z sum below (zafx_dct.hip, BS = true); tables A | B as above, chirp and its transform below
            const int Mh = N / 2;
            pl->dct_half = Mh;
            pl->dct_den2 = Mh;   // (the chirp of an Mh-point DFT: exp(-i pi m^2 / Mh))
            pl->log2nf = 4;
            pl->bs_log2m = 7;
            while ((1 << pl->bs_log2m) < 2 * Mh - 1) ++pl->bs_log2m;
            aux.resize(2 * (size_t)(Mh + 1));
            for (int k = 0; k <= Mh; ++k) {
                aux[(size_t)k] = type == 4 ? unit_root(4LL * k + 1, 8LL * N) : unit_root(k, 2LL * Mh);
                aux[(size_t)(Mh + 1 + k)] = type == 4 ? unit_root(k, 2LL * N) : unit_root(k, 4LL * N);
            }
            pl->kernel_name = "k_dct_bsh";
        } else if (N <= 8192) {
            // Every other length (the reference's np.fft.fft takes any, zaf.py:760-839, :900-981): all eight transforms are
            //     y[k] = s_out[k] sum_n s_in[n] x[n] cos | sin(pi (n + a)(k + b) / D),   a, b in {0, 1/2, 1},  D = N - 1 | N | N + 1,
            // and with n k = (n^2 + k^2 - (k - n)^2) / 2 the sum is a convolution with a chirp of step pi / D (a chirp-z transform):
            //     y[k] = Re(Q[k] sum_n (x[n] P[n]) conj(c)[k - n]),   c[j] = exp(-i pi j^2 / (2 D)),
            //     P[n] = s_in[n] exp(-i pi n b / D) c[n],   Q[k] = s_out[k] exp(-i pi (a k + a b) / D) c[k]  (x i for the sines),
            // on the Bluestein machinery of zafx_bs32.hip (k_dct_bs32: two transforms of 2^ceil(log2(2N - 1)) points per vector instead of
            // the dense N x N product of round 4).  Angles in units of pi / (4 D), reduced in integers.
            const int a2 = sine ? (type == 1 ? 2 : type == 3 ? 2 : 1) : (type == 2 || type == 4 ? 1 : 0);   // 2 a
            const int b2 = sine ? (type == 1 ? 2 : type == 2 ? 2 : 1) : (type == 3 || type == 4 ? 1 : 0);   // 2 b
            const long long D = type == 1 ? (sine ? N + 1 : N - 1) : N;
            pl->dct_den2 = 2 * D;
            pl->log2nf = 4;   // (the FFT tables are those of the convolution length, below)
            pl->bs_log2m = 7;
            while ((1 << pl->bs_log2m) < 2 * N - 1) ++pl->bs_log2m;
            const double L = (double)(1 << pl->bs_log2m), s = std::sqrt(2.0 / (double)D), r2 = std::sqrt(0.5);
            aux.resize(2 * (size_t)N);
            for (long long n = 0; n < N; ++n) {
                double sin_ = 1.0, sout = s;
                if (type == 1 && !sine && (n == 0 || n == N - 1)) sin_ = r2, sout = s * r2;
                if (type == 2 && (sine ? n == N - 1 : n == 0)) sout = s * r2;
                if (type == 3 && (sine ? n == N - 1 : n == 0)) sin_ = r2;
                auto root = [&](long long m, double scale) {   // scale exp(-i pi m / (4 D)), in float64, exact on the axes
                    m %= 8 * D;
                    const double ang = M_PI * (double)m / (double)(4 * D);
                    double c = std::cos(ang), sn = -std::sin(ang);
                    if (m % (2 * D) == 0) {
                        const int quarter = (int)(m / (2 * D));
                        c = quarter == 0 ? 1.0 : quarter == 2 ? -1.0 : 0.0;
                        sn = quarter == 1 ? -1.0 : quarter == 3 ? 1.0 : 0.0;
                    }
                    return cf32{(float)(scale * c), (float)(scale * sn)};
                };
                aux[(size_t)n] = root(2 * n * b2 + 2 * n * n, sin_);
                aux[(size_t)(N + n)] = root(2 * n * a2 + a2 * b2 + 2 * n * n + (sine ? 6 * D : 0), sout / L);   // (x i = exp(-i pi 6 D / (4 D)))
            }
            pl->kernel_name = "k_dct_bs32";
        } else {
            return bail("dct / dst: lengths above 8192 need N / 2 (types 2-4), N - 1 (dct type 1) or N + 1 (dst type 1) to be a power of two (at most 8192)");
        }
    } else {
        return bail("unknown plan kind");
    }
    pl->log2e = default_log2e(pl->log2nf);

    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&pl->n_cus, hipDeviceAttributeMultiprocessorCount, device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&pl->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&pl->ev0);
    if (e == hipSuccess) e = hipEventCreate(&pl->ev1);
    if (e == hipSuccess) {
        auto tw = is_cqt_family(kind) ? build_two_level_twiddles(pl->log2nf) : build_pass_twiddles(pl->log2nf, pl->log2e);
        if (is_cqt_family(kind) && cqt_split(pl->log2nf)) {   // + the pass tables of the 1024-point sub-transforms
            const auto sub = build_pass_twiddles(10, 4);
            tw.insert(tw.end(), sub.begin(), sub.end());
        }
        if (tw.empty()) tw.push_back(cf32{1.f, 0.f});
        e = upload(&pl->d_tw_pass, tw.data(), tw.size() * sizeof(cf32));
        if (e == hipSuccess && (kind == ZAFX_STFT || kind == ZAFX_MEL || kind == ZAFX_MFCC) && (pl->log2nf == 10 || (ZAFX_STFT_FAT8_TABLES && pl->log2nf == 11 && kind == ZAFX_STFT))) {
            auto tw5 = build_pass_twiddles(pl->log2nf, 5);   // (W = 4096: 2048 points as 32 x 32 x 2 in the persistent 8-frame form)
            e = upload(&pl->d_tw_r32, tw5.data(), tw5.size() * sizeof(cf32));
        }
    }
    if (e == hipSuccess && (kind == ZAFX_STFT || kind == ZAFX_ISTFT || kind == ZAFX_MEL || kind == ZAFX_MFCC) && pl->log2nf == 11 && pl->prm.precision == ZAFX_PRECISION_F32) {
        const auto sub = build_pass_twiddles(10, 4);   // the two 1024-point band transforms of k_stft_ft16b
        e = upload(&pl->d_tw_sub, sub.data(), sub.size() * sizeof(cf32));
    }
    if (e == hipSuccess && (kind == ZAFX_STFT || kind == ZAFX_ISTFT || kind == ZAFX_MEL || kind == ZAFX_MFCC) && pl->log2nf == 12 && pl->prm.precision == ZAFX_PRECISION_F32 && pl->bs_log2m == 0) {
        const auto sub = build_pass_twiddles(10, 4);   // the four 1024-point class transforms of k_stft_ft16q, and the roots that form their inputs
        e = upload(&pl->d_tw_sub, sub.data(), sub.size() * sizeof(cf32));
        std::vector<cf32> q(4096);
        for (int n = 0; n < 4096; ++n) q[(size_t)n] = unit_root(n, 8192);
        if (e == hipSuccess) e = upload(&pl->d_tw_quad, q.data(), q.size() * sizeof(cf32));
    }
    if (e == hipSuccess && kind == ZAFX_MDCT && pl->log2nf == 11 && pl->prm.precision == ZAFX_PRECISION_F32 && pl->bs_log2m == 0) {
        const auto sub = build_pass_twiddles(9, 3);   // k_mdct_ft32q: the four 512-point band transforms, and exp(-2 pi i n / 2048)
        e = upload(&pl->d_tw_sub, sub.data(), sub.size() * sizeof(cf32));
        std::vector<cf32> q(2048);
        for (int n = 0; n < 2048; ++n) q[(size_t)n] = unit_root(n, 2048);
        if (e == hipSuccess) e = upload(&pl->d_tw_quad, q.data(), q.size() * sizeof(cf32));
    }
    if (e == hipSuccess && kind == ZAFX_IMDCT && pl->log2nf == 11 && pl->prm.precision == ZAFX_PRECISION_F32 && pl->bs_log2m == 0) {
        const auto sub = build_pass_twiddles(10, 4);   // k_imdct_q: the two 1024-point class transforms, and exp(-2 pi i n / 2048) that joins them
        e = upload(&pl->d_tw_sub, sub.data(), sub.size() * sizeof(cf32));
        std::vector<cf32> q(2048);
        for (int n = 0; n < 2048; ++n) q[(size_t)n] = unit_root(n, 2048);
        if (e == hipSuccess) e = upload(&pl->d_tw_quad, q.data(), q.size() * sizeof(cf32));
    }
    if (e == hipSuccess && kind == ZAFX_MDCT && pl->log2nf == 10 && pl->prm.precision == ZAFX_PRECISION_F32 && pl->bs_log2m == 0) {
        // k_mdct_ft32b: pass tables of the two 512-point band transforms; g in band-major order, and g[n] exp(-2 pi i n / 1024)
        const auto sub = build_pass_twiddles(9, 3);
        e = upload(&pl->d_tw_sub, sub.data(), sub.size() * sizeof(cf32));
        const int nf = pl->W / 4;
        std::vector<cf32> bt((size_t)nf + nf / 2);
        for (int s = 0; s < 2; ++s)
            for (int q = 0; q < nf / 2; ++q) bt[(size_t)s * (nf / 2) + q] = unit_root(8LL * (2 * q + s) + 1, 8LL * pl->W);
        for (int n = 0; n < nf / 2; ++n) bt[(size_t)nf + n] = unit_root(40LL * n + 1, 32LL * nf);
        if (e == hipSuccess) e = upload(&pl->d_tw_band, bt.data(), bt.size() * sizeof(cf32));
    }
    if (e == hipSuccess) e = upload(&pl->d_tw_aux, aux.data(), aux.size() * sizeof(cf32));
    if (e == hipSuccess && pl->prm.precision == ZAFX_PRECISION_F32 && pl->bs_log2m > 0) {   // float32 Bluestein plan (zafx_bs32.hip)
        const int M = 1 << pl->bs_log2m, W = pl->dct_half > 0 ? pl->dct_half : pl->W, F = W / 2;   // (W: the points of the transform that is convolved)
        auto twm = build_pass_twiddles(pl->bs_log2m, default_log2e(pl->bs_log2m));
        if (twm.empty()) twm.push_back(cf32{1.f, 0.f});
        e = upload(&pl->d_tw_pass, twm.data(), twm.size() * sizeof(cf32));
        std::vector<cld> c, b;
        bluestein_tables(W, kind == ZAFX_DCT ? pl->dct_den2 : (long long)W, pl->bs_log2m, c, b);
        std::vector<cf32> cf((size_t)W), bf((size_t)M);
        for (int i = 0; i < W; ++i) cf[(size_t)i] = cf32{(float)c[(size_t)i].real(), (float)c[(size_t)i].imag()};
        for (int i = 0; i < M; ++i) bf[(size_t)i] = cf32{(float)b[(size_t)i].real(), (float)b[(size_t)i].imag()};
        if (e == hipSuccess) e = upload(&pl->d_bs_chirp, cf.data(), cf.size() * sizeof(cf32));
        if (e == hipSuccess) e = upload(&pl->d_bs_bhat, bf.data(), bf.size() * sizeof(cf32));
        if (is_mdct_family(kind)) {   // pre / post twiddles of the reference's own W-point formulation (zaf.py:1047-1056, :1138-1156)
            std::vector<cf32> pp;
            if (kind == ZAFX_MDCT) {
                for (int n = 0; n < W; ++n) pp.push_back(unit_root(n, 2LL * W));                                  // exp(-i pi n / W)
                for (int k = 0; k < F; ++k) pp.push_back(unit_root((long long)(F + 1) * (2 * k + 1), 4LL * W));   // exp(-i pi (F+1)(2k+1) / (2W))
            } else {
                for (int k = 0; k < F; ++k) pp.push_back(unit_root((long long)(F + 1) * k, 2LL * W));             // exp(-i pi (F+1) k / W)
                for (int n = 0; n < W; ++n) pp.push_back(unit_root(2LL * n + 1 + F, 4LL * W));                    // exp(-i pi (2n+1+F) / (2W))
            }
            if (e == hipSuccess) e = upload(&pl->d_tw_aux, pp.data(), pp.size() * sizeof(cf32));
        }
        pl->kernel_name = kind == ZAFX_DCT ? (pl->dct_half > 0 ? "k_dct_bsh" : "k_dct_bs32") : kind == ZAFX_STFT ? "k_stft_bs32" : kind == ZAFX_ISTFT ? "k_ifft_frames_bs32" : kind == ZAFX_MDCT ? "k_mdct_bs32"
                          : (kind == ZAFX_MEL || kind == ZAFX_MFCC) ? mel_wide_kernel_name() : "k_imdct_frames_bs32";
    }
    if (e == hipSuccess && pl->prm.precision == ZAFX_PRECISION_F64) {   // float64 tables, evaluated in long double
        const bool mdct = is_mdct_family(kind), cqt = is_cqt_family(kind);
        const int n = mdct ? pl->W / 4 : cqt ? std::min(pl->W, kCqt64Sub) : pl->W / 2;   // FFT length (CQT: of one decimated sub-sequence)
        auto root = [](long long num, long long den) {
            const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)(num % den) / (long double)den;
            double2 r = make_double2((double)cosl(a), (double)sinl(a));
            if (num % den == 0) r = make_double2(1.0, 0.0);
            if (4 * (num % den) == den) r = make_double2(0.0, -1.0);
            if (2 * (num % den) == den) r = make_double2(-1.0, 0.0);
            return r;
        };
        std::vector<double2> tw((size_t)std::max(n / 2, 1)), tws((size_t)(mdct ? n : cqt ? pl->W : n / 2 + 1));
        for (int m = 0; m < n / 2; ++m) tw[(size_t)m] = root(m, n);
        if (pl->bs_log2m > 0) {   // Bluestein: roots of M, the chirp c[n] = exp(-i pi n^2 / W), and FFT_M of its wrapped conjugate
            const int M = 1 << pl->bs_log2m, W = pl->W;
            tw.assign((size_t)M / 2, make_double2(0, 0));
            for (int m = 0; m < M / 2; ++m) tw[(size_t)m] = root(m, M);
            tws.assign((size_t)W, make_double2(0, 0));
            for (long long k = 0; k < W; ++k) tws[(size_t)k] = root((k * k) % (2LL * W), 2LL * W);
            std::vector<cld> c, b;
            bluestein_tables(W, (long long)W, pl->bs_log2m, c, b);
            std::vector<double2> bhat((size_t)M);
            for (int i = 0; i < M; ++i) bhat[(size_t)i] = make_double2((double)b[(size_t)i].real(), (double)b[(size_t)i].imag());
            e = upload(&pl->d_bhat64, bhat.data(), bhat.size() * sizeof(double2));
        } else if (mdct) {   // g_m = exp(-i pi (8m+1) / (8M)), M = W/2: pre- and post-twiddle of the DCT-IV
            for (int m = 0; m < n; ++m) tws[(size_t)m] = root(8LL * m + 1, 8LL * pl->W);
        } else if (cqt) {   // every root of W: the decimation-in-time recombination reads exp(-2 pi i (n1 k mod W) / W)
            for (int k = 0; k < pl->W; ++k) tws[(size_t)k] = root(k, pl->W);
        } else {
            for (int k = 0; k <= n / 2; ++k) tws[(size_t)k] = root(k, pl->W);
        }
        if (e == hipSuccess) e = upload(&pl->d_tw64, tw.data(), tw.size() * sizeof(double2));
        if (e == hipSuccess) e = upload(&pl->d_tws64, tws.data(), tws.size() * sizeof(double2));
        pl->kernel_name = kind == ZAFX_STFT ? stft_f64_kernel_name() : kind == ZAFX_ISTFT ? istft_f64_kernel_name()
                          : kind == ZAFX_MDCT ? mdct_f64_kernel_name() : kind == ZAFX_IMDCT ? imdct_f64_kernel_name()
                          : cqt ? cqt_f64_kernel_name() : mel_f64_kernel_name();
        if (pl->bs_log2m > 0)
            pl->kernel_name = kind == ZAFX_ISTFT ? "k_ifft_frames_bs_f64" : kind == ZAFX_MDCT ? "k_mdct_bs_f64"
                              : kind == ZAFX_IMDCT ? "k_imdct_frames_bs_f64" : "k_stft_bs_f64";
    }
    if (e != hipSuccess) {
        zafx_plan_destroy(pl);
        return fail("zafx_plan_create", e);
    }
    *out = pl;
    return 0;
}

int zafx_plan_destroy(zafx_plan* pl) {
    if (!pl) return 0;
    (void)hipSetDevice(pl->device);
    if (pl->stream) (void)hipStreamSynchronize(pl->stream);
    for (int l = 0; l < 2; ++l)
        if (pl->lane_pcm[l]) (void)hipFree(pl->lane_pcm[l]);
    if (pl->d_window) (void)hipFree(pl->d_window);
    if (pl->d_matrix) (void)hipFree(pl->d_matrix);
    if (pl->d_wfold) (void)hipFree(pl->d_wfold);
    if (pl->d_tw_pass) (void)hipFree(pl->d_tw_pass);
    if (pl->d_tw_r32) (void)hipFree(pl->d_tw_r32);
    if (pl->d_tw_quad) (void)hipFree(pl->d_tw_quad);
    if (pl->d_tw_sub) (void)hipFree(pl->d_tw_sub);
    if (pl->d_tw_band) (void)hipFree(pl->d_tw_band);
    if (pl->d_fbw) (void)hipFree(pl->d_fbw);
    if (pl->d_fbw_meta) (void)hipFree(pl->d_fbw_meta);
    if (pl->d_dctw) (void)hipFree(pl->d_dctw);
    if (pl->d_tw_aux) (void)hipFree(pl->d_tw_aux);
    if (pl->d_indptr) (void)hipFree(pl->d_indptr);
    if (pl->d_indices) (void)hipFree(pl->d_indices);
    if (pl->d_values) (void)hipFree(pl->d_values);
    if (pl->d_cqt_waves) (void)hipFree(pl->d_cqt_waves);
    if (pl->d_cqt_addrs) (void)hipFree(pl->d_cqt_addrs);
    if (pl->d_cqt_vals) (void)hipFree(pl->d_cqt_vals);
    if (pl->d_cqt_mm_vals) (void)hipFree(pl->d_cqt_mm_vals);
    if (pl->d_cqt_mm_addr) (void)hipFree(pl->d_cqt_mm_addr);
    if (pl->d_cqt_mm_fin) (void)hipFree(pl->d_cqt_mm_fin);
    if (pl->d_window64) (void)hipFree(pl->d_window64);
    if (pl->d_tw64) (void)hipFree(pl->d_tw64);
    if (pl->d_tws64) (void)hipFree(pl->d_tws64);
    if (pl->d_scratch64) (void)hipFree(pl->d_scratch64);
    if (pl->d_fb64) (void)hipFree(pl->d_fb64);
    if (pl->d_fb64_meta) (void)hipFree(pl->d_fb64_meta);
    if (pl->d_dct64) (void)hipFree(pl->d_dct64);
    if (pl->d_cqt64_tw1) (void)hipFree(pl->d_cqt64_tw1);
    if (pl->d_cqt64_split) (void)hipFree(pl->d_cqt64_split);
    if (pl->d_cqt64_vals) (void)hipFree(pl->d_cqt64_vals);
    if (pl->d_cqt64_meta) (void)hipFree(pl->d_cqt64_meta);
    if (pl->d_cqt64_fin) (void)hipFree(pl->d_cqt64_fin);
    if (pl->d_mel64_stream) (void)hipFree(pl->d_mel64_stream);
    if (pl->d_mel64_fin) (void)hipFree(pl->d_mel64_fin);
    if (pl->d_mel64_dctT) (void)hipFree(pl->d_mel64_dctT);
    if (pl->d_values64) (void)hipFree(pl->d_values64);
    if (pl->d_bhat64) (void)hipFree(pl->d_bhat64);
    if (pl->d_pcm_float) (void)hipFree(pl->d_pcm_float);
    if (pl->d_bs_chirp) (void)hipFree(pl->d_bs_chirp);
    if (pl->d_bs_bhat) (void)hipFree(pl->d_bs_bhat);
    free_band(pl->fb);
    free_band(pl->dct);
    for (hipStream_t st : {pl->stream_up, pl->stream_down})
        if (st) (void)hipStreamSynchronize(st);
    for (int l = 0; l < 2; ++l) {
        if (pl->lane_in[l]) (void)hipFree(pl->lane_in[l]);
        if (pl->lane_out[l]) (void)hipFree(pl->lane_out[l]);
    }
    for (hipEvent_t ev : pl->pipe_ev)
        if (ev) (void)hipEventDestroy(ev);
    for (hipStream_t st : {pl->stream_up, pl->stream_down})
        if (st) (void)hipStreamDestroy(st);
    if (pl->ev0) (void)hipEventDestroy(pl->ev0);
    if (pl->ev1) (void)hipEventDestroy(pl->ev1);
    if (pl->stream) (void)hipStreamDestroy(pl->stream);
    delete pl;
    return 0;
}

static int expected_constant_bytes(const zafx_plan* pl, int which, size_t bytes, size_t* elem) {
    switch (which) {
        case ZAFX_CONST_WINDOW:
            if (is_cqt_family(pl->kind) || pl->kind == ZAFX_LINEAR || pl->kind == ZAFX_DCT) return fail_msg("this plan kind takes no window");
            if (pl->prm.precision == ZAFX_PRECISION_F64) {
                *elem = sizeof(double);
                return bytes == (size_t)pl->W * sizeof(double) ? 0 : fail_msg("window must hold window_length float64 (ZAFX_PRECISION_F64 plan)");
            }
            *elem = sizeof(float);
            return bytes == (size_t)pl->W * sizeof(float) ? 0 : fail_msg("window must hold window_length float32");
        case ZAFX_CONST_MEL_FB:
            *elem = pl->prm.precision == ZAFX_PRECISION_F64 ? sizeof(double) : sizeof(float);
            if (pl->kind != ZAFX_MEL && pl->kind != ZAFX_MFCC) return fail_msg("plan takes no mel filterbank");
            return bytes == (size_t)pl->prm.n_filters * (pl->W / 2) * *elem ? 0
                       : fail_msg("mel filterbank must be dense float32 (float64 for a ZAFX_PRECISION_F64 plan) [n_filters][window_length/2]");
        case ZAFX_CONST_DCT:
            *elem = pl->prm.precision == ZAFX_PRECISION_F64 ? sizeof(double) : sizeof(float);
            if (pl->kind != ZAFX_MFCC) return fail_msg("plan takes no DCT matrix");
            return bytes == (size_t)pl->prm.n_coefs * pl->prm.n_filters * *elem ? 0
                       : fail_msg("DCT matrix must be float32 (float64 for a ZAFX_PRECISION_F64 plan) [n_coefs][n_filters]");
        case ZAFX_CONST_MATRIX:
            *elem = sizeof(float);
            if (pl->kind != ZAFX_LINEAR) return fail_msg("plan takes no transform matrix");
            return bytes == (size_t)pl->prm.n_filters * pl->W * sizeof(float) ? 0 : fail_msg("matrix must be float32 [n_filters][window_length]");
        case ZAFX_CONST_CQT_INDPTR:
            *elem = sizeof(int32_t);
            if (!is_cqt_family(pl->kind)) return fail_msg("plan takes no CQT kernel");
            return bytes == (size_t)(pl->prm.n_bins + 1) * sizeof(int32_t) ? 0 : fail_msg("indptr must hold n_bins + 1 int32");
        case ZAFX_CONST_CQT_INDICES:
            *elem = sizeof(int32_t);
            if (!is_cqt_family(pl->kind)) return fail_msg("plan takes no CQT kernel");
            return bytes % sizeof(int32_t) == 0 ? 0 : fail_msg("indices must be int32");
        case ZAFX_CONST_CQT_VALUES:
            *elem = pl->prm.precision == ZAFX_PRECISION_F64 ? sizeof(double2) : sizeof(cf32);
            if (!is_cqt_family(pl->kind)) return fail_msg("plan takes no CQT kernel");
            return bytes % *elem == 0 ? 0 : fail_msg("values must be complex64 (complex128 for a ZAFX_PRECISION_F64 plan)");
    }
    return fail_msg("unknown constant id");
}

static int store_shadow(zafx_plan* pl, int which, const void* host, size_t bytes) {
    switch (which) {
        case ZAFX_CONST_WINDOW:
            if (pl->prm.precision == ZAFX_PRECISION_F64) pl->h_window64.assign((const double*)host, (const double*)host + bytes / sizeof(double));
            else pl->h_window.assign((const float*)host, (const float*)host + bytes / sizeof(float));
            break;
        case ZAFX_CONST_MEL_FB:
            if (pl->prm.precision == ZAFX_PRECISION_F64) pl->h_fb64.assign((const double*)host, (const double*)host + bytes / sizeof(double));
            else pl->h_fb.assign((const float*)host, (const float*)host + bytes / sizeof(float));
            break;
        case ZAFX_CONST_DCT:
            if (pl->prm.precision == ZAFX_PRECISION_F64) pl->h_dct64.assign((const double*)host, (const double*)host + bytes / sizeof(double));
            else pl->h_dct.assign((const float*)host, (const float*)host + bytes / sizeof(float));
            break;
        case ZAFX_CONST_MATRIX:
            pl->h_matrix.assign((const float*)host, (const float*)host + bytes / sizeof(float));
            break;
        case ZAFX_CONST_CQT_INDPTR:
            pl->h_indptr.assign((const int32_t*)host, (const int32_t*)host + bytes / sizeof(int32_t));
            break;
        case ZAFX_CONST_CQT_INDICES: {
            pl->h_indices.assign((const int32_t*)host, (const int32_t*)host + bytes / sizeof(int32_t));
            for (int32_t c : pl->h_indices)
                if (c < 0 || c >= pl->W) return fail_msg("CQT kernel column index out of range");
            break;
        }
        case ZAFX_CONST_CQT_VALUES:
            if (pl->prm.precision == ZAFX_PRECISION_F64) pl->h_values64.assign((const double2*)host, (const double2*)host + bytes / sizeof(double2));
            else pl->h_values.assign((const cf32*)host, (const cf32*)host + bytes / sizeof(cf32));
            break;
    }
    return 0;
}

int zafx_plan_set_constant(zafx_plan* pl, int which, const void* host, size_t bytes) {
    if (!pl || (!host && bytes)) return fail_msg("null argument");
    size_t elem = 1;
    if (int rc = expected_constant_bytes(pl, which, bytes, &elem)) return rc;
    ZAFX_HIP(hipSetDevice(pl->device));
    ZAFX_HIP(hipStreamSynchronize(pl->stream));
    if (int rc = store_shadow(pl, which, host, bytes)) return rc;
    return finalize_constant(pl, which);
}

static int stft_frames(int64_t n, int w, int h) {   // zaf.py:99-109
    const int64_t pad = w / 2;
    const int64_t num = n + 2 * pad - w;
    const int64_t q = num >= 0 ? (num + h - 1) / h : -((-num) / h);
    return (int)(q + 1);
}

int zafx_plan_out_dims(const zafx_plan* pl, int64_t n_in, int64_t dims[2]) {
    if (!pl || !dims) return fail_msg("null argument");
    if (n_in < 0) return fail_msg("negative size");
    const int64_t w = pl->W, h = pl->H;
    switch (pl->kind) {
        case ZAFX_STFT:
            dims[0] = pl->prm.spectrum != ZAFX_SPECTRUM_TWO_SIDED ? w / 2 + 1 : w;
            dims[1] = stft_frames(n_in, pl->W, pl->H);
            return 0;
        case ZAFX_MEL: dims[0] = pl->prm.n_filters; dims[1] = stft_frames(n_in, pl->W, pl->H); return 0;
        case ZAFX_MFCC: dims[0] = pl->prm.n_coefs + (pl->prm.with_mel ? pl->prm.n_filters : 0); dims[1] = stft_frames(n_in, pl->W, pl->H); return 0;
        case ZAFX_ISTFT: dims[0] = std::max<int64_t>(0, n_in * h - (w - h)); dims[1] = 1; return 0;   // zaf.py:217, :236-238
        case ZAFX_MDCT: dims[0] = w / 2; dims[1] = (n_in + h - 1) / h + 1; return 0;                  // zaf.py:1033
        case ZAFX_IMDCT: dims[0] = std::max<int64_t>(0, h * (n_in - 1) - 1); dims[1] = 1; return 0;    // zaf.py:1132, :1182
        case ZAFX_CQT: dims[0] = pl->prm.n_bins; dims[1] = n_in / h; return 0;                         // zaf.py:606
        case ZAFX_CHROMA: dims[0] = pl->prm.octave_resolution; dims[1] = n_in / h; return 0;
        case ZAFX_LINEAR:
            if (n_in != pl->W) return fail_msg("linear map: input length must equal window_length");
            dims[0] = pl->prm.n_filters; dims[1] = 1; return 0;
        case ZAFX_DCT:
            if (n_in != pl->W) return fail_msg("dct / dst: input length must equal window_length");
            dims[0] = pl->W; dims[1] = 1; return 0;
    }
    return fail_msg("unknown plan kind");
}

int zafx_plan_row_pitch(const zafx_plan* pl, int64_t n_in, int64_t* pitch) {
    if (!pl || !pitch) return fail_msg("null argument");
    int64_t dims[2];
    if (int rc = zafx_plan_out_dims(pl, n_in, dims)) return rc;
    switch (pl->kind) {
        case ZAFX_ISTFT:   // the 2-D side is the input: n_in = T
            *pitch = pl->layout == ZAFX_LAYOUT_FT ? row_pitch(*pl, n_in)
                                                  : (pl->prm.spectrum != ZAFX_SPECTRUM_TWO_SIDED ? pl->W / 2 + 1 : pl->W);
            return 0;
        case ZAFX_IMDCT: *pitch = pl->layout == ZAFX_LAYOUT_FT ? row_pitch(*pl, n_in) : pl->W / 2; return 0;
        case ZAFX_LINEAR: case ZAFX_DCT: *pitch = dims[0]; return 0;
        default: *pitch = pl->layout == ZAFX_LAYOUT_FT ? row_pitch(*pl, dims[1]) : dims[0]; return 0;
    }
}

int zafx_execute(zafx_plan* pl, const void* d_in, void* d_out, int64_t n_clips, int64_t n_in) {
    if (!pl) return fail_msg("null plan");
    if (n_clips < 0 || n_in < 0) return fail_msg("negative size");
    if (n_clips == 0) return 0;
    if (!d_in || !d_out) return fail_msg("null device pointer");
    int64_t dims[2];
    if (int rc = zafx_plan_out_dims(pl, n_in, dims)) return rc;
    if (pl->kind == ZAFX_LINEAR && !pl->d_matrix) return fail_msg("matrix constant not set");
    if (!is_cqt_family(pl->kind) && pl->kind != ZAFX_LINEAR && pl->kind != ZAFX_DCT && !pl->d_window && !pl->d_window64) return fail_msg("window constant not set");
    if ((pl->kind == ZAFX_MEL || pl->kind == ZAFX_MFCC) && !pl->fb.d_pack && !pl->d_fb64 && !pl->d_fbw) return fail_msg("mel filterbank constant not set");
    if (pl->kind == ZAFX_MFCC && !pl->dct.d_pack && !pl->d_dct64 && !pl->d_dctw) return fail_msg("DCT constant not set");
    if (is_cqt_family(pl->kind)) {
        const bool f64 = pl->prm.precision == ZAFX_PRECISION_F64;
        if (!pl->d_indptr || !pl->d_indices || !(f64 ? (void*)pl->d_values64 : (void*)pl->d_values)) return fail_msg("CQT kernel constants not set");
        if ((int)(f64 ? pl->h_values64.size() : pl->h_values.size()) != pl->nnz || pl->h_indptr.back() != pl->nnz)
            return fail_msg("CQT kernel CSR arrays are inconsistent");
        for (size_t r = 0; f64 && r + 1 < pl->h_indptr.size(); ++r)
            if (pl->h_indptr[r + 1] < pl->h_indptr[r]) return fail_msg("CQT kernel indptr is not monotone");
        if (!f64 && pl->cqt_dirty) {
            ZAFX_HIP(hipSetDevice(pl->device));
            if (int rc = build_cqt_chunks(pl)) return rc;
        }
    }
    if (n_clips * std::max<int64_t>(dims[1], 1) > 0x7fffffffLL) return fail_msg("batch too large for one launch (clips x frames >= 2^31)");
    ZAFX_HIP(hipSetDevice(pl->device));
    hipError_t e = hipSuccess;
    switch (pl->kind) {
        case ZAFX_STFT:
            if (pl->prm.precision == ZAFX_PRECISION_F64) e = launch_stft_f64(*pl, (const double*)d_in, (double2*)d_out, n_clips, n_in, (int)dims[1]);
            else if (pl->bs_log2m > 0) e = launch_stft_bs32(*pl, (const float*)d_in, (float2*)d_out, n_clips, n_in, (int)dims[1]);
            else e = launch_stft(*pl, (const float*)d_in, (float2*)d_out, n_clips, n_in, (int)dims[1]);
            break;
        case ZAFX_ISTFT:
            if (pl->cola_gain == 0.f) return fail_msg("istft: sum(window[0:W:H]) is zero (zaf.py:241 would divide by zero)");
            if (pl->prm.precision == ZAFX_PRECISION_F64) e = launch_istft_f64(*pl, (const double2*)d_in, (double*)d_out, n_clips, (int)n_in, dims[0]);
            else if (pl->bs_log2m > 0) e = launch_istft_bs32(*pl, (const float2*)d_in, (float*)d_out, n_clips, (int)n_in, dims[0]);
            else e = launch_istft(*pl, (const float2*)d_in, (float*)d_out, n_clips, (int)n_in, dims[0]);
            break;
        case ZAFX_MDCT:
            if (pl->prm.precision == ZAFX_PRECISION_F64) e = launch_mdct_f64(*pl, (const double*)d_in, (double*)d_out, n_clips, n_in, (int)dims[1]);
            else if (pl->bs_log2m > 0) e = launch_mdct_bs32(*pl, (const float*)d_in, (float*)d_out, n_clips, n_in, (int)dims[1]);
            else e = launch_mdct(*pl, (const float*)d_in, (float*)d_out, n_clips, n_in, (int)dims[1]);
            break;
        case ZAFX_IMDCT:
            if (pl->prm.precision == ZAFX_PRECISION_F64) e = launch_imdct_f64(*pl, (const double*)d_in, (double*)d_out, n_clips, (int)n_in, dims[0]);
            else if (pl->bs_log2m > 0) e = launch_imdct_bs32(*pl, (const float*)d_in, (float*)d_out, n_clips, (int)n_in, dims[0]);
            else e = launch_imdct(*pl, (const float*)d_in, (float*)d_out, n_clips, (int)n_in, dims[0]);
            break;
        case ZAFX_MEL:
        case ZAFX_MFCC:
            if (pl->prm.precision == ZAFX_PRECISION_F64) e = launch_mel_f64(*pl, (const double*)d_in, (double*)d_out, n_clips, n_in, (int)dims[1]);
            else e = launch_mel(*pl, (const float*)d_in, (float*)d_out, n_clips, n_in, (int)dims[1]);
            break;
        case ZAFX_LINEAR:
            e = launch_linear(*pl, (const float*)d_in, (float*)d_out, n_clips);
            break;
        case ZAFX_DCT:
            e = pl->bs_log2m > 0 && pl->dct_half == 0 ? launch_dct_bs32(*pl, (const float*)d_in, (float*)d_out, n_clips) : launch_dct(*pl, (const float*)d_in, (float*)d_out, n_clips);
            break;
        case ZAFX_CQT:
        case ZAFX_CHROMA:
            if (pl->prm.precision == ZAFX_PRECISION_F64) e = launch_cqt_f64(*pl, (const double*)d_in, (double*)d_out, n_clips, n_in, (int)dims[1]);
            else e = launch_cqt(*pl, (const float*)d_in, (float*)d_out, n_clips, n_in, (int)dims[1]);
            break;
        default:
            return fail_msg("unknown plan kind");
    }
    if (e != hipSuccess) {
        if (g_err.empty() || e != hipErrorInvalidValue) return fail("zafx_execute", e);
        return (int)e;
    }
    return 0;
}

int zafx_sync(zafx_plan* pl) {
    if (!pl) return fail_msg("null plan");
    ZAFX_HIP(hipSetDevice(pl->device));
    ZAFX_HIP(hipStreamSynchronize(pl->stream));
    return 0;
}

// Bytes of ONE clip on either side of the plan for `n_in` (as zafx_plan_out_dims): the caller's host arrays hold
// n_clips x these, C-contiguous, rows at the plan's pitch.
static int clip_bytes(const zafx_plan* pl, int64_t n_in, int64_t* in_b, int64_t* out_b) {
    int64_t dims[2], pitch = 0;
    if (int rc = zafx_plan_out_dims(pl, n_in, dims)) return rc;
    if (int rc = zafx_plan_row_pitch(pl, n_in, &pitch)) return rc;
    const bool f64 = pl->prm.precision == ZAFX_PRECISION_F64;
    const int64_t real = f64 ? 8 : 4, cplx = 2 * real;
    const bool ft = pl->layout == ZAFX_LAYOUT_FT;
    switch (pl->kind) {
        case ZAFX_STFT: {
            const int64_t e = pl->prm.spectrum >= ZAFX_SPECTRUM_MAGNITUDE ? real : cplx;
            *in_b = n_in * real;
            *out_b = (ft ? dims[0] * pitch : dims[1] * dims[0]) * e;
            return 0;
        }
        case ZAFX_MDCT: case ZAFX_MEL: case ZAFX_MFCC: case ZAFX_CQT: case ZAFX_CHROMA:
            *in_b = n_in * real;
            *out_b = (ft ? dims[0] * pitch : dims[1] * dims[0]) * real;
            return 0;
        case ZAFX_ISTFT: {
            const int64_t rows = pl->prm.spectrum != ZAFX_SPECTRUM_TWO_SIDED ? pl->W / 2 + 1 : pl->W;
            *in_b = (ft ? rows * pitch : n_in * rows) * cplx;
            *out_b = dims[0] * real;
            return 0;
        }
        case ZAFX_IMDCT:
            *in_b = (ft ? (int64_t)(pl->W / 2) * pitch : n_in * (pl->W / 2)) * real;
            *out_b = dims[0] * real;
            return 0;
        case ZAFX_LINEAR: case ZAFX_DCT:
            *in_b = (int64_t)pl->W * 4;
            *out_b = dims[0] * 4;
            return 0;
    }
    return fail_msg("unknown plan kind");
}

int zafx_plan_clip_bytes(const zafx_plan* pl, int64_t n_in, int64_t* in_bytes, int64_t* out_bytes) {
    if (!pl || !in_bytes || !out_bytes) return fail_msg("null argument");
    return clip_bytes(pl, n_in, in_bytes, out_bytes);
}

// Host array in -> transform -> host array out, in chunks of clips through a three-stage pipeline: uploads on their own
// stream, kernels on the plan's stream, downloads on a third, two sets of device staging buffers, events between the stages
//     host:      waits for down(c - 2) (both buffers of set c & 1 are free again), then enqueues
//     up(c)      on the upload stream                                  -> ev_up[c & 1]
//     kernel(c)  on the plan's stream, waits for up(c)                  -> ev_k[c & 1]
//     down(c)    on the download stream, waits for kernel(c)           -> ev_down[c & 1]
// so the download of chunk c runs while chunk c + 1 is uploaded and transformed, and the call approaches the rate of the
// slower PCIe direction alone instead of the sum of the three phases.  One stream per DIRECTION matters: with upload and
// download of a chunk on the same stream (round 3's first version: two lanes of upload -> kernel -> download) the two lanes fell
// into step, uploaded together and downloaded together, and nothing overlapped (2.92 against 2.81 Gsamples/s serial;
// tools/exp_pcie.hip: one stream per direction overlaps fully, 19.6 ms for 4.7 + 18.8).  Page-locked host arrays
// (zafx_host_alloc) make the copies asynchronous; pageable ones are staged by the runtime (correct, slower).  Plans whose
// kernels share a plan-owned scratch (float64 and Bluestein inverse forms) are safe too: their kernels stay on one stream.
// pcm_bytes > 0: h_in holds interleaved integer PCM (n_clips, n_in, pcm_channels) of pcm_bytes per sample; every chunk is uploaded as it
// is (2 or 4 bytes per sample and channel cross PCIe) and normalised + mixed down on the device (k_pcm_to_float on the plan's stream, in front
// of the transform) -- zaf.py:1202 and :65 -- into the float32 staging buffer the transform reads.
// One transform of integer PCM that is already on the device: kinds whose kernel reads int16 itself (k_mel2: mel, mfcc, |X| / |X|^2 at W = 2048, mono or
// stereo) get the integers; everything else goes through wavread's normalisation and the channel mean (k_pcm_to_float) into `staging` first.
static int execute_pcm(zafx_plan* pl, const void* d_pcm, void* d_out, int64_t n_clips, int64_t n_frames, int n_channels, int sample_bytes, float* staging) {
    int64_t pdims[2] = {0, 0};
    if (pl->kind == ZAFX_STFT) {
        if (int rc = zafx_plan_out_dims(pl, n_frames, pdims)) return rc;
    }
    if ((pcm_direct_ok(*pl, n_frames, n_channels, sample_bytes) && reinterpret_cast<uintptr_t>(d_pcm) % 8 == 0) ||
        mdct_pcm_direct_ok(*pl, n_frames, n_channels, sample_bytes, d_pcm) || stft_pcm_direct_ok(*pl, n_frames, n_channels, sample_bytes, d_pcm, (int)pdims[1])) {
        zafx::set_pcm_mode(n_channels);
        const int rc = zafx_execute(pl, d_pcm, d_out, n_clips, n_frames);
        const bool taken = zafx::pcm_mode_taken();
        zafx::set_pcm_mode(0);
        if (rc == 0 && !taken && n_clips * n_frames > 0)   // (a route that reads float samples was handed int16: the output is garbage -- say so)
            return fail_msg("internal: the plan's route did not take the int16 input it was promised to (zafx_execute_pcm)");
        return rc;
    }
    if (staging) {   // (zafx_run_host_pcm: a chunk-sized lane of the pipeline)
        if (n_clips * n_frames > 0) ZAFX_HIP(launch_pcm_to_float(pl->stream, d_pcm, staging, n_clips * n_frames, n_channels, sample_bytes));
        return zafx_execute(pl, staging, d_out, n_clips, n_frames);
    }
    // the plan's own float32 staging array, bounded: clips are converted and transformed in chunks under the scratch budget (1 GiB;
    // ZAFX_SCRATCH_BUDGET_MB) -- a 1024 x 10 s call used to leave 1.8 GB pinned to a cached plan
    int64_t in_b = 0, out_b = 0;
    if (int rc = clip_bytes(pl, n_frames, &in_b, &out_b)) return rc;
    const int64_t chunk = zafx::scratch_clips_per_chunk(n_clips, 1, (int)std::min<int64_t>(n_frames, INT32_MAX), sizeof(float));
    const size_t need = (size_t)std::max<int64_t>(chunk * n_frames * 4, 1);
    if (pl->pcm_float_bytes < need) {
        ZAFX_HIP(hipStreamSynchronize(pl->stream));
        if (pl->d_pcm_float) ZAFX_HIP(hipFree(pl->d_pcm_float));
        pl->d_pcm_float = nullptr, pl->pcm_float_bytes = 0;
        ZAFX_HIP(hipMalloc(&pl->d_pcm_float, need));
        pl->pcm_float_bytes = need;
    }
    const int64_t pcm_clip_b = n_frames * n_channels * sample_bytes;
    for (int64_t c0 = 0; c0 < n_clips; c0 += chunk) {   // (one stream: a chunk's conversion waits for the transform of the chunk before it)
        const int64_t n = std::min(chunk, n_clips - c0);
        if (n * n_frames > 0)
            ZAFX_HIP(launch_pcm_to_float(pl->stream, (const char*)d_pcm + c0 * pcm_clip_b, (float*)pl->d_pcm_float, n * n_frames, n_channels, sample_bytes));
        if (int rc = zafx_execute(pl, pl->d_pcm_float, (char*)d_out + c0 * out_b, n, n_frames)) return rc;
    }
    return 0;
}

int zafx_execute_pcm(zafx_plan* pl, const void* d_pcm, void* d_out, int64_t n_clips, int64_t n_frames, int n_channels, int sample_bytes) {
    if (!pl) return fail_msg("null plan");
    if (n_clips < 0 || n_frames < 0) return fail_msg("negative size");
    if (n_channels < 1 || n_channels > 64) return fail_msg("n_channels must be in [1, 64]");
    if (sample_bytes != 2 && sample_bytes != 4) return fail_msg("sample_bytes must be 2 (int16) or 4 (int32)");
    if (pl->prm.precision != ZAFX_PRECISION_F32) return fail_msg("PCM ingest feeds the float32 plans");
    switch (pl->kind) {
        case ZAFX_STFT: case ZAFX_MDCT: case ZAFX_MEL: case ZAFX_MFCC: case ZAFX_CQT: case ZAFX_CHROMA: case ZAFX_DCT: break;
        default: return fail_msg("PCM ingest feeds the plans that take samples (stft, mdct, mel, mfcc, cqt, chroma, dct)");
    }
    if (n_clips == 0) return 0;
    if (!d_pcm || !d_out) return fail_msg("null device pointer");
    ZAFX_HIP(hipSetDevice(pl->device));
    return execute_pcm(pl, d_pcm, d_out, n_clips, n_frames, n_channels, sample_bytes, nullptr);
}

static int run_host_impl(zafx_plan* pl, const void* h_in, void* h_out, int64_t n_clips, int64_t n_in, int64_t chunk_clips, int pcm_channels, int pcm_bytes) {
    if (!pl) return fail_msg("null plan");
    if (n_clips < 0 || n_in < 0) return fail_msg("negative size");
    if (n_clips == 0) return 0;
    if (!h_in || !h_out) return fail_msg("null host pointer");
    int64_t in_b = 0, out_b = 0;
    if (int rc = clip_bytes(pl, n_in, &in_b, &out_b)) return rc;
    const int64_t host_in_b = pcm_bytes > 0 ? n_in * pcm_channels * pcm_bytes : in_b;   // bytes of one clip in h_in
    ZAFX_HIP(hipSetDevice(pl->device));
    if (chunk_clips <= 0) {
        // default: chunks of about 128 MB (both sides together): about two milliseconds of PCIe each, so the pipeline fills
        // quickly (its first upload and last download are not overlapped) and the fixed costs per chunk (launch, copy set-up,
        // events: tens of microseconds) stay at a few percent (tools/e2e_pcie.py: 64 ... 256 MB are within 1 % of each other)
        chunk_clips = std::max<int64_t>(1, (int64_t)(128 << 20) / std::max<int64_t>(host_in_b + out_b, 1));
    }
    chunk_clips = std::min(chunk_clips, n_clips);
    const int64_t n_chunks = (n_clips + chunk_clips - 1) / chunk_clips;
    const int sets = n_chunks > 1 ? 2 : 1;
    if (sets == 2 && !pl->stream_up) {
        // created into locals and handed to the plan only when ALL of them exist: a partial set (a failed event) would leave a later
        // call with a null stream (the legacy default stream) or null events and a pipeline without its ordering
        hipStream_t up = nullptr, down = nullptr;
        hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        hipError_t e = hipStreamCreateWithFlags(&up, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&down, hipStreamNonBlocking);
        for (int i = 0; i < 6 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
        if (e != hipSuccess) {
            for (int i = 0; i < 6; ++i)
                if (ev[i]) (void)hipEventDestroy(ev[i]);
            if (down) (void)hipStreamDestroy(down);
            if (up) (void)hipStreamDestroy(up);
            ZAFX_HIP(e);
        }
        for (int i = 0; i < 6; ++i) pl->pipe_ev[i] = ev[i];
        pl->stream_down = down;
        pl->stream_up = up;   // (last: the guard above reads it)
    }
    // integer PCM that the plan's kernel reads in its own loads (every chunk has the same geometry and the lanes are 256-byte aligned)?
    bool pcm_direct = false;
    if (pcm_bytes > 0) {
        int64_t pd[2] = {0, 0};
        if (pl->kind == ZAFX_STFT && zafx_plan_out_dims(pl, n_in, pd)) return 1;
        pcm_direct = pcm_direct_ok(*pl, n_in, pcm_channels, pcm_bytes) || mdct_pcm_direct_ok(*pl, n_in, pcm_channels, pcm_bytes, nullptr) ||
                     stft_pcm_direct_ok(*pl, n_in, pcm_channels, pcm_bytes, nullptr, (int)pd[1]);
    }
    for (int l = 0; l < sets; ++l) {   // grow-only staging buffers
        const size_t need_in = (size_t)std::max<int64_t>(chunk_clips * in_b, 1), need_out = (size_t)std::max<int64_t>(chunk_clips * out_b, 1);
        if (pl->lane_in_bytes[l] < need_in && !pcm_direct) {   // (integer PCM read by the kernel itself: no float32 lane)
            if (pl->lane_in[l]) ZAFX_HIP(hipFree(pl->lane_in[l]));
            pl->lane_in[l] = nullptr, pl->lane_in_bytes[l] = 0;
            ZAFX_HIP(hipMalloc(&pl->lane_in[l], need_in));
            pl->lane_in_bytes[l] = need_in;
        }
        const size_t need_pcm = pcm_bytes > 0 ? (size_t)std::max<int64_t>(chunk_clips * host_in_b, 1) : 0;
        if (pl->lane_pcm_bytes[l] < need_pcm) {
            if (pl->lane_pcm[l]) ZAFX_HIP(hipFree(pl->lane_pcm[l]));
            pl->lane_pcm[l] = nullptr, pl->lane_pcm_bytes[l] = 0;
            ZAFX_HIP(hipMalloc(&pl->lane_pcm[l], need_pcm));
            pl->lane_pcm_bytes[l] = need_pcm;
        }
        if (pl->lane_out_bytes[l] < need_out) {
            if (pl->lane_out[l]) ZAFX_HIP(hipFree(pl->lane_out[l]));
            pl->lane_out[l] = nullptr, pl->lane_out_bytes[l] = 0;
            ZAFX_HIP(hipMalloc(&pl->lane_out[l], need_out));
            pl->lane_out_bytes[l] = need_out;
        }
    }
    hipStream_t const s_k = pl->stream;
    hipStream_t const s_up = sets == 2 ? pl->stream_up : s_k, s_down = sets == 2 ? pl->stream_down : s_k;
    hipEvent_t* const ev_up = pl->pipe_ev, * const ev_k = pl->pipe_ev + 2, * const ev_down = pl->pipe_ev + 4;
    int ret = 0;
    hipError_t e = hipSuccess;
    for (int64_t c = 0; c < n_chunks && !ret; ++c) {
        const int l = (int)(c % sets);
        const int64_t first = c * chunk_clips, count = std::min(chunk_clips, n_clips - first);
        // The host runs at most two chunks ahead: before buffer set l is used again it waits for the download that emptied it
        // (which implies that the kernel before it has read the input buffer).  With the whole batch enqueued at once, more
        // than ~70 chunks in flight made the runtime fall off a cliff (1024 clips in 147 chunks: 444 ms instead of 135).
        if (sets == 2 && c >= 2) e = hipEventSynchronize(ev_down[l]);
        if (e == hipSuccess && count * host_in_b > 0)
            e = hipMemcpyAsync(pcm_bytes > 0 ? pl->lane_pcm[l] : pl->lane_in[l], (const char*)h_in + first * host_in_b, (size_t)(count * host_in_b), hipMemcpyHostToDevice, s_up);
        if (e == hipSuccess && sets == 2) e = hipEventRecord(ev_up[l], s_up);
        if (e == hipSuccess && sets == 2) e = hipStreamWaitEvent(s_k, ev_up[l], 0);
        if (e != hipSuccess) { ret = fail("zafx_run_host: upload", e); break; }
        if (pcm_bytes > 0) ret = execute_pcm(pl, pl->lane_pcm[l], pl->lane_out[l], count, n_in, pcm_channels, pcm_bytes, pcm_direct ? nullptr : (float*)pl->lane_in[l]);
        else ret = zafx_execute(pl, pl->lane_in[l], pl->lane_out[l], count, n_in);   // (on pl->stream = s_k)
        if (ret) break;
        if (sets == 2) e = hipEventRecord(ev_k[l], s_k);
        if (e == hipSuccess && sets == 2) e = hipStreamWaitEvent(s_down, ev_k[l], 0);
        if (e == hipSuccess && count * out_b > 0)
            e = hipMemcpyAsync((char*)h_out + first * out_b, pl->lane_out[l], (size_t)(count * out_b), hipMemcpyDeviceToHost, s_down);
        if (e == hipSuccess && sets == 2) e = hipEventRecord(ev_down[l], s_down);
        if (e != hipSuccess) { ret = fail("zafx_run_host: download", e); break; }
    }
    for (hipStream_t st : {s_up, s_k, s_down}) {   // (also after an error: nothing of this call is left in flight)
        e = hipStreamSynchronize(st);
        if (e != hipSuccess && !ret) ret = fail("zafx_run_host: sync", e);
    }
    return ret;
}

int zafx_run_host(zafx_plan* pl, const void* h_in, void* h_out, int64_t n_clips, int64_t n_in, int64_t chunk_clips) {
    return run_host_impl(pl, h_in, h_out, n_clips, n_in, chunk_clips, 0, 0);
}

int zafx_run_host_pcm(zafx_plan* pl, const void* h_pcm, void* h_out, int64_t n_clips, int64_t n_frames, int n_channels, int sample_bytes,
                      int64_t chunk_clips) {
    if (!pl) return fail_msg("null plan");
    if (n_channels < 1 || n_channels > 64) return fail_msg("n_channels must be in [1, 64]");
    if (sample_bytes != 2 && sample_bytes != 4) return fail_msg("sample_bytes must be 2 (int16) or 4 (int32)");
    if (pl->prm.precision != ZAFX_PRECISION_F32) return fail_msg("PCM ingest feeds float32 plans");
    switch (pl->kind) {
        case ZAFX_STFT: case ZAFX_MDCT: case ZAFX_MEL: case ZAFX_MFCC: case ZAFX_CQT: case ZAFX_CHROMA: case ZAFX_DCT: break;
        default: return fail_msg("PCM ingest feeds the plans that take samples (stft, mdct, mel, mfcc, cqt, chroma, dct)");
    }
    return run_host_impl(pl, h_pcm, h_out, n_clips, n_frames, chunk_clips, n_channels, sample_bytes);
}

int zafx_alloc_placed(zafx_plan* pl, void** dptr, size_t bytes, const void* d_in, int64_t n_clips, int64_t n_in, int n_candidates, int reps, float* probe_ms) {
    if (!pl || !dptr) return fail_msg("null argument");
    if (n_candidates < 1 || n_candidates > 64 || reps < 1) return fail_msg("n_candidates must be in [1, 64], reps >= 1");
    int64_t in_b = 0, out_b = 0;
    if (int rc = clip_bytes(pl, n_in, &in_b, &out_b)) return rc;
    if (n_clips < 0 || (uint64_t)(n_clips * out_b) > (uint64_t)bytes) return fail_msg("bytes is smaller than the plan's output for (n_clips, n_in)");
    if (n_clips > 0 && !d_in) return fail_msg("null device pointer");
    ZAFX_HIP(hipSetDevice(pl->device));
    std::vector<void*> cand;
    for (int i = 0; i < n_candidates; ++i) {
        void* p = nullptr;
        const hipError_t e = device_alloc(pl->device, &p, bytes);   // (the library's allocator: arrays of 1 GiB and more in chunks, see zafx_alloc)
        if (e != hipSuccess) {
            (void)hipGetLastError();
            if (cand.empty()) return fail("zafx_alloc_placed", e);
            break;   // out of room: the candidates so far
        }
        cand.push_back(p);
    }
    int best = 0, ret = 0;
    float best_ms = 0.f;
    for (size_t i = 0; i < cand.size() && !ret; ++i) {
        float ms = 0.f;
        for (int w = 0; w < (i == 0 ? 2 : 1) && !ret; ++w) ret = zafx_execute(pl, d_in, cand[i], n_clips, n_in);   // untimed (the first candidate also carries the clock ramp)
        if (!ret) ret = zafx_timer_start(pl);
        for (int r = 0; r < reps && !ret; ++r) ret = zafx_execute(pl, d_in, cand[i], n_clips, n_in);
        if (!ret) ret = zafx_timer_stop(pl, &ms);
        ms /= (float)reps;
        if (probe_ms) probe_ms[i] = ms;
        if (i == 0 || ms < best_ms) best = (int)i, best_ms = ms;
    }
    if (probe_ms)
        for (int i = (int)cand.size(); i < n_candidates; ++i) probe_ms[i] = -1.f;   // (not tried)
    for (size_t i = 0; i < cand.size(); ++i)
        if (ret || (int)i != best) (void)device_free(cand[i]);
    if (ret) return ret;
    *dptr = cand[(size_t)best];
    return 0;
}

int zafx_timer_start(zafx_plan* pl) {
    if (!pl) return fail_msg("null plan");
    ZAFX_HIP(hipSetDevice(pl->device));
    ZAFX_HIP(hipEventRecord(pl->ev0, pl->stream));
    return 0;
}

int zafx_timer_stop(zafx_plan* pl, float* ms) {
    if (!pl || !ms) return fail_msg("null argument");
    ZAFX_HIP(hipSetDevice(pl->device));
    ZAFX_HIP(hipEventRecord(pl->ev1, pl->stream));
    ZAFX_HIP(hipEventSynchronize(pl->ev1));
    ZAFX_HIP(hipEventElapsedTime(ms, pl->ev0, pl->ev1));
    return 0;
}

int zafx_pcm_to_float(zafx_plan* pl, const void* d_pcm, void* d_out, int64_t n_clips, int64_t n_frames, int n_channels,
                      int sample_bytes) {
    if (!pl) return fail_msg("null plan");
    if (n_clips < 0 || n_frames < 0) return fail_msg("negative size");
    if (n_channels < 1 || n_channels > 64) return fail_msg("n_channels must be in [1, 64]");
    if (sample_bytes != 2 && sample_bytes != 4) return fail_msg("sample_bytes must be 2 (int16) or 4 (int32)");
    if (n_clips * n_frames == 0) return 0;
    if (!d_pcm || !d_out) return fail_msg("null device pointer");
    ZAFX_HIP(hipSetDevice(pl->device));
    ZAFX_HIP(launch_pcm_to_float(pl->stream, d_pcm, (float*)d_out, n_clips * n_frames, n_channels, sample_bytes));
    return 0;
}

int zafx_cqt_max_bins(int fft_length, int* n_bins) {
    if (!n_bins) return fail_msg("null argument");
    const int lw = ilog2_exact(fft_length);
    *n_bins = (lw >= 1 && cqt_supported(lw - 1)) ? cqt_max_bins(lw - 1) : 0;
    return 0;
}

int zafx_plan_kernel_name(const zafx_plan* pl, char* buf, size_t buflen) {
    if (!pl || !buf || !buflen) return fail_msg("null argument");
    std::snprintf(buf, buflen, "%s", pl->kernel_name.c_str());
    return 0;
}

int zafx_plan_last_kernel_name(const zafx_plan* pl, char* buf, size_t buflen) {
    if (!pl || !buf || !buflen) return fail_msg("null argument");
    const char* ran = pl->ran.load(std::memory_order_acquire);
    std::snprintf(buf, buflen, "%s", ran ? ran : "");
    return 0;
}

// ---------------------------------------------------------------------------------
// RCCL (resolved lazily with dlopen so that single-GPU use needs no librccl)
// ---------------------------------------------------------------------------------
struct uid128 { char b[128]; };
struct rccl_api {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, /* ncclUniqueId by value: 128 bytes */ uid128, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
    int (*CommUserRank)(void*, int*) = nullptr;
};

static rccl_api g_rccl;
static std::mutex g_rccl_mu;

static int rccl_load() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.lib) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) return fail_msg(std::string("cannot load librccl: ") + dlerror());
    g_rccl.GetUniqueId = (int (*)(void*))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(void**, int, uid128, int))dlsym(h, "ncclCommInitRank");
    g_rccl.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
    g_rccl.Broadcast = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclBroadcast");
    g_rccl.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    g_rccl.CommCount = (int (*)(void*, int*))dlsym(h, "ncclCommCount");
    g_rccl.CommUserRank = (int (*)(void*, int*))dlsym(h, "ncclCommUserRank");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.Broadcast)
        return fail_msg("librccl lacks a required symbol");
    g_rccl.lib = h;
    return 0;
}

static int rccl_fail(const char* where, int rc) {
    std::string m = std::string(where) + ": rccl error " + std::to_string(rc);
    if (g_rccl.GetErrorString) m += std::string(" (") + g_rccl.GetErrorString(rc) + ")";
    return fail_msg(m, 1000 + rc);
}

}  // extern "C"

struct zafx_comm {
    int device = 0, rank = 0, n_ranks = 1;
    void* comm = nullptr;
};

extern "C" {

int zafx_comm_unique_id(void* id128) {
    if (!id128) return fail_msg("null argument");
    if (int rc = rccl_load()) return rc;
    int rc = g_rccl.GetUniqueId(id128);
    return rc ? rccl_fail("ncclGetUniqueId", rc) : 0;
}

int zafx_comm_create(zafx_comm** out, int device, int rank, int n_ranks, const void* id128) {
    if (!out || !id128) return fail_msg("null argument");
    if (int rc = rccl_load()) return rc;
    ZAFX_HIP(hipSetDevice(device));
    zafx_comm* c = new zafx_comm();
    c->device = device;
    c->rank = rank;
    c->n_ranks = n_ranks;
    uid128 id;
    std::memcpy(&id, id128, sizeof(id));
    int rc = g_rccl.CommInitRank(&c->comm, n_ranks, id, rank);
    if (rc) {
        delete c;
        return rccl_fail("ncclCommInitRank", rc);
    }
    *out = c;
    return 0;
}

int zafx_comm_destroy(zafx_comm* c) {
    if (!c) return 0;
    if (c->comm && g_rccl.CommDestroy) {
        (void)hipSetDevice(c->device);
        (void)g_rccl.CommDestroy(c->comm);
    }
    delete c;
    return 0;
}

// What the communicator itself says about its size and this process's place in it (ncclCommCount / ncclCommUserRank):
// the record of a multi-GPU run quotes these, not the launcher's environment.
int zafx_comm_count(zafx_comm* c, int* n_ranks) {
    if (!c || !n_ranks) return fail_msg("null argument");
    if (!c->comm || !g_rccl.CommCount) return fail_msg("librccl lacks ncclCommCount");
    int rc = g_rccl.CommCount(c->comm, n_ranks);
    return rc ? rccl_fail("ncclCommCount", rc) : 0;
}
int zafx_comm_user_rank(zafx_comm* c, int* rank) {
    if (!c || !rank) return fail_msg("null argument");
    if (!c->comm || !g_rccl.CommUserRank) return fail_msg("librccl lacks ncclCommUserRank");
    int rc = g_rccl.CommUserRank(c->comm, rank);
    return rc ? rccl_fail("ncclCommUserRank", rc) : 0;
}

// Broadcast every constant the plan kind uses from `root`: an 8-byte length header,
// then the raw bytes (ncclInt8), through a device staging buffer on the plan's stream.
int zafx_comm_broadcast_constants(zafx_comm* c, zafx_plan* pl, int root) {
    if (!c || !pl) return fail_msg("null argument");
    if (c->device != pl->device) return fail_msg("communicator and plan are bound to different devices");
    ZAFX_HIP(hipSetDevice(pl->device));
    std::vector<int> ids;
    if (pl->kind == ZAFX_LINEAR) ids.push_back(ZAFX_CONST_MATRIX);
    else if (pl->kind == ZAFX_DCT) {}   // (no caller-supplied constants: the tables follow from the parameters on every rank)
    else if (!is_cqt_family(pl->kind)) ids.push_back(ZAFX_CONST_WINDOW);
    if (pl->kind == ZAFX_MEL || pl->kind == ZAFX_MFCC) ids.push_back(ZAFX_CONST_MEL_FB);
    if (pl->kind == ZAFX_MFCC) ids.push_back(ZAFX_CONST_DCT);
    if (is_cqt_family(pl->kind)) {
        ids.push_back(ZAFX_CONST_CQT_INDPTR);
        ids.push_back(ZAFX_CONST_CQT_INDICES);
        ids.push_back(ZAFX_CONST_CQT_VALUES);
    }
    unsigned long long* d_len = nullptr;
    ZAFX_HIP(hipMalloc((void**)&d_len, sizeof(unsigned long long)));
    int ret = 0;
    for (int which : ids) {
        const void* hsrc = nullptr;
        unsigned long long len = 0;
        auto span = [&](auto& vec) {
            hsrc = vec.data();
            len = (unsigned long long)vec.size() * sizeof(vec[0]);
        };
        if (c->rank == root) {
            switch (which) {
                case ZAFX_CONST_WINDOW:
                    if (pl->prm.precision == ZAFX_PRECISION_F64) span(pl->h_window64);
                    else span(pl->h_window);
                    break;
                case ZAFX_CONST_MEL_FB:
                    if (pl->prm.precision == ZAFX_PRECISION_F64) span(pl->h_fb64);
                    else span(pl->h_fb);
                    break;
                case ZAFX_CONST_DCT:
                    if (pl->prm.precision == ZAFX_PRECISION_F64) span(pl->h_dct64);
                    else span(pl->h_dct);
                    break;
                case ZAFX_CONST_MATRIX: span(pl->h_matrix); break;
                case ZAFX_CONST_CQT_INDPTR: span(pl->h_indptr); break;
                case ZAFX_CONST_CQT_INDICES: span(pl->h_indices); break;
                case ZAFX_CONST_CQT_VALUES:
                    if (pl->prm.precision == ZAFX_PRECISION_F64) span(pl->h_values64);
                    else span(pl->h_values);
                    break;
            }
            if (len == 0) { ret = fail_msg("root rank has not set every constant before the broadcast"); break; }
        }
        hipError_t e = hipMemcpy(d_len, &len, sizeof(len), hipMemcpyHostToDevice);
        if (e != hipSuccess) { ret = fail("hipMemcpy", e); break; }
        int rc = g_rccl.Broadcast(d_len, d_len, sizeof(len), /*ncclInt8*/ 0, root, c->comm, pl->stream);
        if (rc) { ret = rccl_fail("ncclBroadcast(len)", rc); break; }
        e = hipStreamSynchronize(pl->stream);
        if (e == hipSuccess) e = hipMemcpy(&len, d_len, sizeof(len), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { ret = fail("broadcast header", e); break; }
        void* d_buf = nullptr;
        e = hipMalloc(&d_buf, (size_t)len);
        if (e != hipSuccess) { ret = fail("hipMalloc", e); break; }
        if (c->rank == root) e = hipMemcpy(d_buf, hsrc, (size_t)len, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            rc = g_rccl.Broadcast(d_buf, d_buf, (size_t)len, 0, root, c->comm, pl->stream);
            if (rc) ret = rccl_fail("ncclBroadcast(data)", rc);
            else e = hipStreamSynchronize(pl->stream);
        }
        if (!ret && e == hipSuccess && c->rank != root) {
            std::vector<uint8_t> host((size_t)len);
            e = hipMemcpy(host.data(), d_buf, (size_t)len, hipMemcpyDeviceToHost);
            if (e == hipSuccess) {
                size_t elem = 1;
                ret = expected_constant_bytes(pl, which, (size_t)len, &elem);
                if (!ret) ret = store_shadow(pl, which, host.data(), (size_t)len);
                if (!ret) ret = finalize_constant(pl, which);
            }
        }
        (void)hipFree(d_buf);
        if (!ret && e != hipSuccess) ret = fail("broadcast payload", e);
        if (ret) break;
    }
    (void)hipFree(d_len);
    return ret;
}

}  // extern "C"
