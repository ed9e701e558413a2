// zafx_fft.hpp -- per-frame complex FFT core shared by every zafx kernel (gfx950).
//
// Replaces the np.fft.fft / np.fft.ifft call sites of the reference hot path
// (zaf.py:139, :223, :631, :1068, :1159).  Design (DESIGN.md "FFT core"):
//
//   * A frame of N = 2^LOG2N complex points is owned by P = N/E threads, each
//     holding E = 2^LOG2E points in registers (E = 16 for N >= 1024, so a
//     1024-point frame is exactly ONE 64-lane wavefront and needs no barrier).
//   * Stockham autosort, decimation in time.  log2(N) radix-2 stages are fused
//     into passes of radix 16/8/4/2 done entirely in registers; between passes
//     the frame is exchanged through LDS (write scattered, read p + i*P).
//   * The LDS image is padded by one complex every 16 (phys(i) = i + i/16) so the
//     stride-R scatter of a pass is bank-conflict free for ds_write_b64.
//   * Twiddles are never computed on device (no __sinf): the host builds, in
//     float64, one table per pass laid out [r][k] so that lanes read consecutive
//     k (conflict free); see zafx_twiddle_layout below, used by host and device.
//
// The same header is compiled by g++ with -DZAFX_HOST_EMU for the CPU-side
// algorithm test (tests/host_emu), where "threads" are emulated by loops.
#pragma once

#if defined(ZAFX_HOST_EMU)
#include <cmath>
struct float2 { float x, y; };
static inline float2 make_float2(float x, float y) { float2 r; r.x = x; r.y = y; return r; }
#define ZAFX_HD inline
#else
#include <hip/hip_runtime.h>

#include <type_traits>
#define ZAFX_HD __host__ __device__ __forceinline__
#endif

namespace zafx {

// ---------------------------------------------------------------- complex helpers
// On the device the complex products are written as the two packed-f32 instructions they need
// (v_pk_mul_f32 + v_pk_fma_f32 with op_sel / neg modifiers): from the scalar formula the compiler emits four to
// five packed instructions and a repacking v_mov per product (profiles/r02_notes.md).
#if defined(__HIP_DEVICE_COMPILE__)
typedef float zafx_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ zafx_v2f to_v2(float2 a) { return __builtin_bit_cast(zafx_v2f, a); }
__device__ __forceinline__ float2 to_f2(zafx_v2f a) { return __builtin_bit_cast(float2, a); }
#define ZAFX_PK 1
#endif
// Sums and differences as <2 x float> values: v_pk_add_f32 by construction.  Written component-wise, the compiler's SLP pass pairs
// the x parts of two DIFFERENT points now and then -- four v_mov to gather them and two scalar additions behind, where one packed
// addition does (round 5: 127 scalar f32 operations and ~40 such v_mov per frame in k_mel2).
#if defined(ZAFX_PK)
ZAFX_HD float2 cadd(float2 a, float2 b) { return to_f2(to_v2(a) + to_v2(b)); }
ZAFX_HD float2 csub(float2 a, float2 b) { return to_f2(to_v2(a) - to_v2(b)); }
#else
ZAFX_HD float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
ZAFX_HD float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
#endif
ZAFX_HD float2 cmul(float2 a, float2 b) {
#if defined(ZAFX_PK)
    zafx_v2f r;   // (a.x b.x, a.x b.y), then + (-a.y b.y, a.y b.x)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\tv_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=&v"(r) : "v"(to_v2(a)), "v"(to_v2(b)));
    return to_f2(r);
#else
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
#endif
}
ZAFX_HD float2 cmulc(float2 a, float2 b) {   // a * conj(b)
#if defined(ZAFX_PK)
    zafx_v2f r;   // (a.x b.x, -a.x b.y), then + (a.y b.y, a.y b.x)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]\n\tv_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]"
        : "=&v"(r) : "v"(to_v2(a)), "v"(to_v2(b)));
    return to_f2(r);
#else
    return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
#endif
}
// a * k for a compile-time constant k: the constant rides in a scalar register pair
ZAFX_HD float2 cmulk(float2 a, float kx, float ky) {
#if defined(ZAFX_PK)
    zafx_v2f r, k;
    k.x = kx;
    k.y = ky;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\tv_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=&v"(r) : "v"(to_v2(a)), "s"(k));
    return to_f2(r);
#else
    return make_float2(a.x * kx - a.y * ky, a.x * ky + a.y * kx);
#endif
}
// (a.x b.x, a.y b.y) as ONE packed multiply the compiler cannot contract into a neighbouring addition: code inlined at several
// sites (k_mel's pre-transform) then rounds the same way at each of them
ZAFX_HD float2 mul_elem(float2 a, float2 b) {
#if defined(ZAFX_PK)
    zafx_v2f r;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(to_v2(a)), "v"(to_v2(b)));
    return to_f2(r);
#else
    return make_float2(a.x * b.x, a.y * b.y);
#endif
}
ZAFX_HD float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
ZAFX_HD float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }   // a * (-i)
ZAFX_HD float2 add_mi(float2 a, float2 b) {   // a + (-i) b = (a.x + b.y, a.y - b.x)
#if defined(ZAFX_PK)
    zafx_v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(to_v2(a)), "v"(to_v2(b)));
    return to_f2(r);
#else
    return make_float2(a.x + b.y, a.y - b.x);
#endif
}
ZAFX_HD float2 sub_mi(float2 a, float2 b) {   // a - (-i) b = (a.x - b.y, a.y + b.x)
#if defined(ZAFX_PK)
    zafx_v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(to_v2(a)), "v"(to_v2(b)));
    return to_f2(r);
#else
    return make_float2(a.x - b.y, a.y + b.x);
#endif
}
#if defined(ZAFX_PK)
ZAFX_HD float2 cscale(float2 a, float s) { return to_f2(to_v2(a) * s); }
#else
ZAFX_HD float2 cscale(float2 a, float s) { return make_float2(a.x * s, a.y * s); }
#endif
ZAFX_HD float2 cadd_conj(float2 a, float2 b) {   // a + conj(b)
#if defined(ZAFX_PK)
    zafx_v2f r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(to_v2(a)), "v"(to_v2(b)));
    return to_f2(r);
#else
    return make_float2(a.x + b.x, a.y - b.y);
#endif
}
ZAFX_HD float2 csub_conj(float2 a, float2 b) {   // a - conj(b)
#if defined(ZAFX_PK)
    zafx_v2f r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(to_v2(a)), "v"(to_v2(b)));
    return to_f2(r);
#else
    return make_float2(a.x - b.x, a.y + b.y);
#endif
}
// Real-input split of one (k, N-k) pair of a packed half-length transform: zk = Z[k], zn = Z[N-k], tk = exp(-2 pi i k / 2N).
// X[k] = (E + t_k O), X[N-k] = conj(E - t_k O) with E = (zk + conj zn) / 2, O = -i (zk - conj zn) / 2: eight packed instructions.
ZAFX_HD void split_pair(float2 zk, float2 zn, float2 tk, float2& xk, float2& xn) {
    const float2 e = cadd_conj(zk, zn), u = cmul(csub_conj(zk, zn), tk);
    const float2 a = add_mi(e, u), b = sub_mi(e, u);
#if defined(ZAFX_PK)
    const zafx_v2f hk = {0.5f, 0.5f}, hn = {0.5f, -0.5f};
    xk = to_f2(to_v2(a) * hk);
    xn = to_f2(to_v2(b) * hn);
#else
    xk = make_float2(0.5f * a.x, 0.5f * a.y);
    xn = make_float2(0.5f * b.x, -0.5f * b.y);
#endif
}

// The same pair when only |X[k]|^2 and |X[N-k]|^2 are wanted (k_mel): returns (4 |X[k]|^2, 4 |X[N-k]|^2).  a = E' - i U, b = E' + i U
// with E' = 2 E, U = 2 t_k O are formed component-wise -- (a.x, b.x) and (a.y, b.y) -- so that both squares cost one packed
// multiply and one packed fma: eight packed instructions for the pair instead of twelve (split_pair + two mul / fma pairs),
// and the factor 4 is a power of two that the caller's filterbank absorbs exactly.
ZAFX_HD float2 split_pair_pow4(float2 zk, float2 zn, float2 tk) {
    const float2 e = cadd_conj(zk, zn), u = cmul(csub_conj(zk, zn), tk);
#if defined(ZAFX_PK)
    zafx_v2f px, py, m;
    asm("v_pk_add_f32 %0, %3, %4 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"     // (e.x + u.y, e.x - u.y) = (a.x, b.x)
        "v_pk_add_f32 %1, %3, %4 op_sel:[1,0] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"     // (e.y - u.x, e.y + u.x) = (a.y, b.y)
        "v_pk_mul_f32 %2, %0, %0\n\t"
        "v_pk_fma_f32 %2, %1, %1, %2"
        : "=&v"(px), "=&v"(py), "=&v"(m) : "v"(to_v2(e)), "v"(to_v2(u)));
    return to_f2(m);
#else
    const float ax = e.x + u.y, ay = e.y - u.x, bx = e.x - u.y, by = e.y + u.x;
    return make_float2(ax * ax + ay * ay, bx * bx + by * by);
#endif
}

// ---------------------------------------------------------------- pass schedule
// log2 of the radix of the pass that starts with `rem` radix-2 stages left, when
// a thread holds 2^log2e points.  Shared by host (table builder) and device.
constexpr int pass_log2r(int rem, int log2e) {
    int m = log2e < 5 ? log2e : 5;   // radix <= 32 (32 only for threads that hold 32 points)
    if (rem <= m) return rem;
    if (rem - m == 1 && m >= 3) return m - 1;   // avoid a trailing radix-2 pass
    return m;
}
// offset (in float2 entries) of the twiddle table of the pass that starts at Ns = 2^log2ns
constexpr int twiddle_offset(int log2n, int log2e, int log2ns) {
    int off = 0, ns = 0;
    while (ns < log2ns) {
        int lr = pass_log2r(log2n - ns, log2e);
        if (ns > 0) off += ((1 << lr) - 1) << ns;
        ns += lr;
    }
    return off;
}
constexpr int twiddle_total(int log2n, int log2e) { return twiddle_offset(log2n, log2e, log2n); }

constexpr int default_log2e(int log2n) {
    return log2n >= 10 ? 4 : (log2n >= 7 ? log2n - 6 : 1);
}

// threads that own one frame (usable inside __launch_bounds__, which is a macro)
constexpr int fft_threads(int log2n, int log2e) { return (1 << log2n) >> log2e; }

template <int LOG2N, int LOG2E>
struct FftCfg {
    static constexpr int N = 1 << LOG2N;
    static constexpr int E = 1 << LOG2E;
    static constexpr int P = N / E;                 // threads per frame
    static constexpr int PS = LOG2E >= 5 ? 5 : 4;   // one padding slot every 2^PS points: the widest radix of the config
    static constexpr int PITCH = N + (N >> PS) + 1; // padded complex slots per frame (odd-ish pitch)
    static constexpr int TW = twiddle_total(LOG2N, LOG2E);
};

template <int PS>
ZAFX_HD int phys_t(int i) { return i + (i >> PS); }
ZAFX_HD int phys(int i) { return phys_t<4>(i); }
// phys(base + off) for a compile-time `off`, given pb = phys(base).  When the low part of `base`
// (below the power-of-two SPAN >= 16 that `off` is a multiple-of-stride within) cannot carry into
// the padding term, phys(base + off) = phys(base) + off + off/16: a constant that folds into the
// DS instruction's immediate offset instead of costing address VALU per access.
template <int SPAN, int PS = 4>
ZAFX_HD int phys_off(int pb, int base, int off) {
    if (SPAN >= (1 << PS)) return pb + off + (off >> PS);
    return phys_t<PS>(base + off);
}

// ---------------------------------------------------------------- register DFTs
// Natural-order in, natural-order out, forward sign (e^{-2 pi i nk/R}).
ZAFX_HD void dft2(float2& a, float2& b) {
    float2 t = csub(a, b);
    a = cadd(a, b);
    b = t;
}
ZAFX_HD void dft4(float2& v0, float2& v1, float2& v2, float2& v3) {
    float2 t0 = cadd(v0, v2), t1 = csub(v0, v2);
    float2 t2 = cadd(v1, v3), t3 = csub(v1, v3);
    v0 = cadd(t0, t2);
    v1 = add_mi(t1, t3);
    v2 = csub(t0, t2);
    v3 = sub_mi(t1, t3);
}

// dft4 of (v0, v1, -i v2, v3): the rotation of v2 rides in the first two butterflies
ZAFX_HD void dft4_v2mi(float2& v0, float2& v1, float2& v2, float2& v3) {
    float2 t0 = add_mi(v0, v2), t1 = sub_mi(v0, v2);
    float2 t2 = cadd(v1, v3), t3 = csub(v1, v3);
    v0 = cadd(t0, t2);
    v1 = add_mi(t1, t3);
    v2 = csub(t0, t2);
    v3 = sub_mi(t1, t3);
}

template <int R>
struct Dft;
template <>
struct Dft<1> {
    static ZAFX_HD void run(float2*) {}
};
template <>
struct Dft<2> {
    static ZAFX_HD void run(float2* a) { dft2(a[0], a[1]); }
};
template <>
struct Dft<4> {
    static ZAFX_HD void run(float2* a) { dft4(a[0], a[1], a[2], a[3]); }
};
template <>
struct Dft<8> {
    static ZAFX_HD void run(float2* a) {
        const float h = 0.70710678118654752440f;
        float2 e0 = a[0], e1 = a[2], e2 = a[4], e3 = a[6];
        float2 o0 = a[1], o1 = a[3], o2 = a[5], o3 = a[7];
        dft4(e0, e1, e2, e3);
        dft4(o0, o1, o2, o3);
        o1 = cmulk(o1, h, -h);     // * w8^1
        o3 = cmulk(o3, -h, -h);    // * w8^3
        a[0] = cadd(e0, o0); a[4] = csub(e0, o0);
        a[1] = cadd(e1, o1); a[5] = csub(e1, o1);
        a[2] = add_mi(e2, o2); a[6] = sub_mi(e2, o2);   // * w8^2 = -i folded into the butterfly
        a[3] = cadd(e3, o3); a[7] = csub(e3, o3);
    }
};
template <>
struct Dft<16> {
    static ZAFX_HD void run(float2* a) {
        const float h = 0.70710678118654752440f;
        const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f;   // cos/sin(pi/8)
        float2 m[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {   // 4-point DFTs over n = r + 4 q'
            m[r][0] = a[r]; m[r][1] = a[r + 4]; m[r][2] = a[r + 8]; m[r][3] = a[r + 12];
            dft4(m[r][0], m[r][1], m[r][2], m[r][3]);
        }
        // twiddle m[r][q] *= w16^(r q)
        m[1][1] = cmulk(m[1][1], c1, -s1);    // w^1
        m[1][2] = cmulk(m[1][2], h, -h);      // w^2
        m[1][3] = cmulk(m[1][3], s1, -c1);    // w^3
        m[2][1] = cmulk(m[2][1], h, -h);      // w^2
        // m[2][2] * w^4 = -i m[2][2]: folded into the butterflies of q = 2 below (no swap of its parts in registers)
        m[2][3] = cmulk(m[2][3], -h, -h);     // w^6
        m[3][1] = cmulk(m[3][1], s1, -c1);    // w^3
        m[3][2] = cmulk(m[3][2], -h, -h);     // w^6
        m[3][3] = cmulk(m[3][3], -c1, s1);    // w^9
#pragma unroll
        for (int q = 0; q < 4; ++q) {   // 4-point DFTs over r; output k = q + 4 s
            if (q == 2) dft4_v2mi(m[0][q], m[1][q], m[2][q], m[3][q]);
            else dft4(m[0][q], m[1][q], m[2][q], m[3][q]);
            a[q] = m[0][q]; a[q + 4] = m[1][q]; a[q + 8] = m[2][q]; a[q + 12] = m[3][q];
        }
    }
};

template <>
struct Dft<32> {
    // 32 = 2 x 16: even and odd inputs through Dft<16>, odd outputs times w32^k, radix-2 combine
    static ZAFX_HD void run(float2* a) {
        float2 e[16], o[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            e[i] = a[2 * i];
            o[i] = a[2 * i + 1];
        }
        Dft<16>::run(e);
        Dft<16>::run(o);
        // w32^k = exp(-2 pi i k / 32), k = 1..15
        const float c[16] = {1.f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f, 0.70710678118654752440f,
                             0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f, 0.f,
                             -0.19509032201612826785f, -0.38268343236508977173f, -0.55557023301960222474f, -0.70710678118654752440f,
                             -0.83146961230254523708f, -0.92387953251128675613f, -0.98078528040323044913f};
        const float sn[16] = {0.f, 0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f, 0.70710678118654752440f,
                              0.83146961230254523708f, 0.92387953251128675613f, 0.98078528040323044913f, 1.f,
                              0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f, 0.70710678118654752440f,
                              0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f};
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (k == 8) {   // w32^8 = -i: in the butterfly
                a[k] = add_mi(e[k], o[k]);
                a[k + 16] = sub_mi(e[k], o[k]);
                continue;
            }
            const float2 t = k == 0 ? o[0] : cmulk(o[k], c[k], -sn[k]);
            a[k] = cadd(e[k], t);
            a[k + 16] = csub(e[k], t);
        }
    }
};

// ---------------------------------------------------------------- one Stockham pass
// v[i] holds x[p + i*P].  Twiddle, radix-R butterflies, scatter to the padded
// LDS frame `buf`.  tw points at this pass's [r-1][k] table (unused when Ns = 1).
template <int LOG2N, int LOG2E, int LOG2NS, int LR>
ZAFX_HD void pass_write(const float2* v, float2* buf, int p, const float2* tw) {
    using C = FftCfg<LOG2N, LOG2E>;
    constexpr int R = 1 << LR, NS = 1 << LOG2NS, NB = C::E / R;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int j = p + b * C::P;
        const int k = j & (NS - 1);
        float2 a[R];
#pragma unroll
        for (int r = 0; r < R; ++r) a[r] = v[b + r * NB];
        if (LOG2NS > 0) {
#pragma unroll
            for (int r = 1; r < R; ++r) a[r] = cmul(a[r], tw[(r - 1) * NS + k]);
        }
        Dft<R>::run(a);
        const int base = ((j >> LOG2NS) << (LOG2NS + LR)) + k;
        const int pb = phys_t<C::PS>(base);
#pragma unroll
        for (int r = 0; r < R; ++r) buf[phys_off<NS * R, C::PS>(pb, base, r * NS)] = a[r];
    }
}

// pass_write whose outputs are multiplied by conj-free table entries on their way out: buf[k] = conj(a * post[k]), k the
// natural index of the output.  Used as the LAST pass of transforms that end with a post-twiddle (MDCT): the multiply costs
// one product per point here, against a product and a select per stored coefficient in the store phase.
template <int LOG2N, int LOG2E, int LOG2NS, int LR>
ZAFX_HD void pass_write_post(const float2* v, float2* buf, int p, const float2* tw, const float2* post) {
    using C = FftCfg<LOG2N, LOG2E>;
    constexpr int R = 1 << LR, NS = 1 << LOG2NS, NB = C::E / R;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int j = p + b * C::P;
        const int k = j & (NS - 1);
        float2 a[R];
#pragma unroll
        for (int r = 0; r < R; ++r) a[r] = v[b + r * NB];
        if (LOG2NS > 0) {
#pragma unroll
            for (int r = 1; r < R; ++r) a[r] = cmul(a[r], tw[(r - 1) * NS + k]);
        }
        Dft<R>::run(a);
        const int base = ((j >> LOG2NS) << (LOG2NS + LR)) + k;
        const int pb = phys_t<C::PS>(base);
#pragma unroll
        for (int r = 0; r < R; ++r) buf[phys_off<NS * R, C::PS>(pb, base, r * NS)] = cconj(cmul(a[r], post[base + r * NS]));
    }
}

// ---------------------------------------------------------------- chained-twiddle pass
// For frames whose per-pass tables do not fit LDS (CQT: 16384 points -> 131 KB) the base
// twiddle w = exp(-2 pi i k / (Ns R)) comes from a two-level table (root(idx) = hi[idx >> 7] *
// lo[idx & 127], 2 KB for N = 16384) and w^2 .. w^(R-1) from a product tree of depth <= 4
// (a few ulp; the CQT tolerance is 1e-4).  No global-memory twiddle traffic, few live registers.
struct TwoLevelTw {
    const float2* hi;   // hi[j] = exp(-2 pi i (j << 7) / N)
    const float2* lo;   // lo[j] = exp(-2 pi i j / N), j < 128
};
ZAFX_HD float2 tw2(const TwoLevelTw& t, int idx) { return cmul(t.hi[idx >> 7], t.lo[idx & 127]); }

template <int LOG2N, int LOG2E, int LOG2NS, int LR>
ZAFX_HD void pass_write_chain(const float2* v, float2* buf, int p, const TwoLevelTw& t) {
    using C = FftCfg<LOG2N, LOG2E>;
    constexpr int R = 1 << LR, NS = 1 << LOG2NS, NB = C::E / R;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int j = p + b * C::P;
        const int k = j & (NS - 1);
        float2 a[R];
#pragma unroll
        for (int r = 0; r < R; ++r) a[r] = v[b + r * NB];
        if (LOG2NS > 0) {
            float2 w[R];
            w[1] = tw2(t, k << (LOG2N - LOG2NS - LR));
#pragma unroll
            for (int r = 2; r < R; ++r) w[r] = cmul(w[r >> 1], w[r - (r >> 1)]);   // w^r = w^(r/2) * w^(r - r/2)
#pragma unroll
            for (int r = 1; r < R; ++r) a[r] = cmul(a[r], w[r]);
        }
        Dft<R>::run(a);
        const int base = ((j >> LOG2NS) << (LOG2NS + LR)) + k;
        const int pb = phys_t<C::PS>(base);
#pragma unroll
        for (int r = 0; r < R; ++r) buf[phys_off<NS * R, C::PS>(pb, base, r * NS)] = a[r];
    }
}

// One 8-byte LDS read that STAYS one.  Left alone, the compiler pairs neighbouring ds_read_b64 of one base register into ds_read2_b64 /
// ds_read2st64_b64, which the LDS serves at 128 B/clk where two single reads get 256 (tools/exp_lds2.hip, round 5: 4.0 against 2.04 cycles
// per 512 bytes with sixteen waves reading; the guide's LDS table says the same).  A volatile access is never paired; the waits stay the compiler's.
// Used where it was measured to pay (k_mel2's transform: mfcc 0.99 -> 0.94 ms, mel unchanged); in regs_read and the wave-local 1024-point core
// it is neutral for the HBM-bound kernels and costs k_cqt 1.6 % (24.02 -> 24.40 ms: twice the DS instructions to issue), so they keep the pairs.
#ifndef ZAFX_LDS_SINGLE
#define ZAFX_LDS_SINGLE 1
#endif
ZAFX_HD float2 lds_ld(const float2* p) {   // p: an address in LDS
#if ZAFX_LDS_SINGLE && defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) const volatile zafx_v2f lds_v2f;
    return to_f2(*(lds_v2f*)(p));
#else
    return *p;
#endif
}

template <int LOG2N, int LOG2E>
ZAFX_HD void regs_read(float2* v, const float2* buf, int p) {
    using C = FftCfg<LOG2N, LOG2E>;
    const int pp = phys_t<C::PS>(p);
#pragma unroll
    for (int i = 0; i < C::E; ++i) v[i] = buf[phys_off<C::P, C::PS>(pp, p, i * C::P)];
}

}  // namespace zafx

#if !defined(ZAFX_HOST_EMU) && defined(__HIPCC__)   // (device code: not for the host-only g++ build of the C-ABI layer, `make asan`)
namespace zafx {

// Buffer (SRSRC) loads: a wave-uniform 128-bit descriptor in SGPRs + one 32-bit byte offset per lane,
// instead of a 64-bit flat address per lane and load -- the persistent kernels keep several load
// streams in flight and cannot afford 2 VGPRs of address for each.  Offsets past `bytes` read 0.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load_f32(__amdgpu_buffer_rsrc_t r, int voff, int soff = 0) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ int buf_load_i32(__amdgpu_buffer_rsrc_t r, int voff, int soff = 0) {
    return __builtin_bit_cast(int, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ float2 buf_load_f32x2(__amdgpu_buffer_rsrc_t r, int voff, int soff = 0) {
    const auto raw = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    float2 f;
    __builtin_memcpy(&f, &raw, 8);
    return f;
}

__device__ __forceinline__ float4 buf_load_f32x4(__amdgpu_buffer_rsrc_t r, int voff, int soff = 0) {
    const auto raw = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    float4 f;
    __builtin_memcpy(&f, &raw, 16);
    return f;
}

// Non-temporal ("streaming") 8-byte store, for an output line that one instruction writes whole and nobody re-reads.
// Measured: k_stft_ft16 1.97 -> 1.87 ms, k_mdct_ft32 1.06 -> 1.01 ms.  It does not generalise: the frame-major STFT
// (lines completed by four instructions at different times) loses 2.0 -> 2.6 ms, the ISTFT's sample stores 1.96 -> 2.13,
// and non-temporal LOADS of the once-read spectra cost the ISTFT / IMDCT 3 % (profiles/r01_notes.md).
typedef float zafx_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_stream(float2* p, float2 v) {
    zafx_f32x2 t;
    t.x = v.x;
    t.y = v.y;
    __builtin_nontemporal_store(t, reinterpret_cast<zafx_f32x2*>(p));   // global_store_dwordx2 ... nt (sc0 / sc1 variants: no difference)
}

// XCD-aware order of a persistent kernel's work list.  Block b runs on XCD b % 8 (observed, not promised: a wrong guess is slower,
// never wrong), so the virtual index v = blockIdx + i gridDim of a grid that is a multiple of 8 belongs to XCD v % 8; each XCD
// gets one contiguous range of the `total` units and its workgroups walk it side by side.  Neighbouring units -- tiles of the
// same clip, whose output rows share cache lines when a row is not a whole number of lines -- then meet in ONE L2 at about the
// same time, and their partial lines merge there instead of leaving for HBM twice.  Bijective for any `total`.
#ifndef ZAFX_XCD_ORDER
#define ZAFX_XCD_ORDER 1
#endif
__device__ __forceinline__ int xcd_order(int v, int total) {
    const int x = v & 7, q = total >> 3, r = total & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (v >> 3);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it waits
// for every global store of the wave to be acknowledged (measured: 8 000 cycles per tile after the
// ISTFT store phase); a kernel whose waves exchange data through LDS alone does not need that.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Synchronise the P threads that own one frame.  P <= 64 -> the frame lives in a
// single wavefront whose LDS operations execute in program order: no barrier.
// The compiler, however, may legally reorder one THREAD's LDS store past its later
// LDS load of a provably different address -- which breaks the cross-lane exchange --
// so the single-wave case still needs a release/acquire fence pair (s_waitcnt
// lgkmcnt(0), no s_barrier).
template <int P>
__device__ __forceinline__ void frame_sync() {
    if constexpr (P > 64) {
        lds_barrier();   // the passes exchange data through LDS only
    } else {
#if defined(ZAFX_WAVE_SYNC_FENCE)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#else
        // LDS services one wavefront's DS instructions in issue order, so a later ds_read of the
        // same wave observes an earlier ds_write without waiting for lgkmcnt(0); what must be
        // prevented is COMPILER reordering across the exchange point.
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
#endif
    }
}

// Sum of `v` over each 16-lane row of the wavefront (DPP, 4 adds, no LDS crossbar): afterwards every
// lane holds its row's sum.
__device__ __forceinline__ float row16_sum(float v) {
    auto dpp_add = [](float x, auto ctrl) {
        constexpr int C = decltype(ctrl)::value;
        const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), C, 0xf, 0xf, true);
        return x + __builtin_bit_cast(float, moved);
    };
    v = dpp_add(v, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
    v = dpp_add(v, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
    v = dpp_add(v, std::integral_constant<int, 0x141>{});   // row_half_mirror
    v = dpp_add(v, std::integral_constant<int, 0x140>{});   // row_mirror
    return v;
}

// Sum of `v` over the 64 lanes of the wavefront, returned in every lane: row sums + four v_readlane
// instead of six dependent ds_bpermute round trips (__shfl_xor).
__device__ __forceinline__ float wave_sum(float v) {
    const int bits = __builtin_bit_cast(int, row16_sum(v));
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bits, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bits, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bits, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bits, 48));
    return (r0 + r1) + (r2 + r3);
}

// ---------------------------------------------------------------- 1024 points in one wavefront, second exchange in registers
// fft_frame<10, 4> is three passes (16 . 16 . 4) with two exchanges through LDS.  The second one is special: after pass 2
// lane (lh, ll) = (lane >> 4, lane & 15) holds in register (rh, rl) = (r >> 2, r & 3) the point at position
//     lh * 256 + ll + 16 * (4 rh + rl),
// and pass 3 (radix 4 over positions 256 apart, butterfly b of lane l' on positions l' + 64 b + 256 r') wants it in lane
// (rl, ll), register (rh, lh): the low four lane bits stay, the DPP-row index of the lane trades places with the low two
// register bits.  That 4 x 4 transpose is two v_permlane32_swap + two v_permlane16_swap per four registers (gfx950) --
// 32 VALU instructions per frame instead of 16 ds_write_b64 + 16 ds_read_b64, on kernels whose FFT phase is bound by
// LDS bandwidth (profiles/r02_notes.md).
#ifndef ZAFX_FFT_PERMLANE
#define ZAFX_FFT_PERMLANE 1
#endif
__device__ __forceinline__ void lane_row_transpose4(float& x0, float& x1, float& x2, float& x3) {
    // x_c[row R] <-> x_R[row c] over the four 16-lane rows: lane bit 5 against register bit 1 (v_permlane32_swap: lanes 32..63 of
    // the first operand trade places with lanes 0..31 of the second), then lane bit 4 against register bit 0
    // (v_permlane16_swap: odd 16-lane rows of the first with even rows of the second).  Inline asm: through
    // __builtin_amdgcn_permlane16_swap the compiler (ROCm 7.2) dropped the second result of half of the swaps in this
    // context (tools/exp: eight of the sixteen v_permlane16_swap gone, wrong spectra).  s_nop 1: a v_permlane*_swap must
    // not read a VGPR in the two wait states after a VALU wrote it, and the hazard recogniser does not look inside asm.
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\ts_nop 1\n\t"
        "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3"
        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
}

// The same transpose on both parts of four complex registers in one block: the four v_permlane32_swap stand between every register's
// VALU write and its v_permlane16_swap, so one leading s_nop serves all eight swaps (two blocks of lane_row_transpose4 carry four).
__device__ __forceinline__ void lane_row_transpose4(float2& c0, float2& c1, float2& c2, float2& c3) {
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\tv_permlane32_swap_b32 %4, %6\n\tv_permlane32_swap_b32 %5, %7\n\t"
        "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\tv_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7"
        : "+v"(c0.x), "+v"(c1.x), "+v"(c2.x), "+v"(c3.x), "+v"(c0.y), "+v"(c1.y), "+v"(c2.y), "+v"(c3.y));
}

// Lane pairs that share 16-byte loads, paired across 16-lane rows: lane l of an even row and lane l + 16 take the points
// n and n + 1 (mod 64), so that one v_permlane16_swap per register hands each its own (row_pair_unpack).  Every 32 lanes
// still hold 32 consecutive residues: the radix-16 outputs of pass 1 (slot 17 p1 + r) spread over all banks.
__device__ __forceinline__ int row_pair_index(int lane) { return 2 * (lane & 15) + ((lane >> 4) & 1) + (lane & 32); }
// a, b: the two points (n, n + 1) of one 16-byte load; lanes of even rows loaded them for i, lanes of odd rows for i + 8.
// On return a = the lane's own point of i, b = its own point of i + 8.
__device__ __forceinline__ void row_pair_unpack(float2& a, float2& b) {
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %2\n\tv_permlane16_swap_b32 %1, %3" : "+v"(a.x), "+v"(a.y), "+v"(b.x), "+v"(b.y));
}

// twiddles of pass 2 (w[r] = exp(-2 pi i r k / 256), k = lane & 15) and of butterfly b of pass 3 come from the pass tables
// of fft_frame<10, 4> or from the two-level table + product tree of fft_frame_chain<10, 4>
__device__ __forceinline__ void pass2_twiddles(float2* w, int k, const float2* tw) {
    const float2* t = tw + twiddle_offset(10, 4, 4);
#pragma unroll
    for (int r = 1; r < 16; ++r) w[r] = t[(r - 1) * 16 + k];
}
__device__ __forceinline__ void pass2_twiddles(float2* w, int k, const TwoLevelTw& t) {
    w[1] = tw2(t, k << 2);
#pragma unroll
    for (int r = 2; r < 16; ++r) w[r] = cmul(w[r >> 1], w[r - (r >> 1)]);
}
__device__ __forceinline__ void pass3_write(const float2* v, float2* buf, int lane, const float2* tw) {
    pass_write<10, 4, 8, 2>(v, buf, lane, tw + twiddle_offset(10, 4, 8));
}
__device__ __forceinline__ void pass3_write(const float2* v, float2* buf, int lane, const TwoLevelTw& t) {
    pass_write_chain<10, 4, 8, 2>(v, buf, lane, t);
}
// Pass 3 of the 1024-point frame (radix 4: butterfly b of a lane yields the positions lane + 64 b + 256 r, r = 0 .. 3) with only
// the outputs a caller reads.  k_cqt's kernel matrix touches the low bins and, through the real split, their mirrors (config Q:
// positions 2 .. 167 and 857 .. 1021 of every sub-transform): with the wanted positions below 64 (HB + 1) and above
// 1024 - 64 (HB + 1) only output 0 of the butterflies b <= HB and output 3 of the butterflies b >= 3 - HB are formed and stored
// (compile-time: the other outputs' additions and stores are not there; HB = 2: 6 of 16 LDS stores per lane).
template <int HB>
__device__ __forceinline__ void pass3_write_pruned(const float2* v, float2* buf, int lane, const TwoLevelTw& t) {
    using C = FftCfg<10, 4>;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const bool lo = b <= HB, hi = b >= 3 - HB;
        if (!lo && !hi) continue;
        const int k = lane + b * 64;
        float2 a[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = v[b + 4 * r];
        float2 w[4];
        w[1] = tw2(t, k);
        w[2] = cmul(w[1], w[1]);
        w[3] = cmul(w[1], w[2]);
#pragma unroll
        for (int r = 1; r < 4; ++r) a[r] = cmul(a[r], w[r]);
        Dft<4>::run(a);
        const int pb = phys_t<C::PS>(k);
        if (lo) buf[pb] = a[0];
        if (hi) buf[phys_off<1024, C::PS>(pb, k, 768)] = a[3];
    }
}
// (the same with the pass's [r - 1][k] table: three LDS reads per butterfly instead of a two-level product and two more products)
template <int HB>
__device__ __forceinline__ void pass3_write_pruned(const float2* v, float2* buf, int lane, const float2* tw) {
    using C = FftCfg<10, 4>;
    const float2* t = tw + twiddle_offset(10, 4, 8);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const bool lo = b <= HB, hi = b >= 3 - HB;
        if (!lo && !hi) continue;
        const int k = lane + b * 64;
        float2 a[4];
        a[0] = v[b];
#pragma unroll
        for (int r = 1; r < 4; ++r) a[r] = cmul(v[b + 4 * r], t[(r - 1) * 256 + k]);
        Dft<4>::run(a);
        const int pb = phys_t<C::PS>(k);
        if (lo) buf[pb] = a[0];
        if (hi) buf[phys_off<1024, C::PS>(pb, k, 768)] = a[3];
    }
}
__device__ __forceinline__ void pass1_write(const float2* v, float2* buf, int lane, const float2* tw) { pass_write<10, 4, 0, 4>(v, buf, lane, tw); }
__device__ __forceinline__ void pass1_write(const float2* v, float2* buf, int lane, const TwoLevelTw& t) { pass_write_chain<10, 4, 0, 4>(v, buf, lane, t); }

// Input: v[i] = x[lane + 64 i].  Output: natural-order spectrum in the padded LDS frame `buf` (as fft_frame<10, 4>).
// ODDROT: the odd lanes hold their 16 points rotated by 8, v[i] = x[lane + 64 ((i + 8) & 15)] (lane pairs that share 16-byte
// loads, see k_mel / k_cqt): their radix-16 outputs of pass 1 then carry (-1)^k, undone here.
// PAIR (row-pair form): p1 = the index n mod 64 of the points the lane holds on entry, v[i] = x[p1 + 64 i], p1 = row_pair_index(lane)
// (below).  Sixteen consecutive lanes then hold sixteen p1 of one parity, and in the usual layout (slot 17 p1 + r) their 8-byte
// writes of pass 1 would meet two by two in the banks (SQ_LDS_BANK_CONFLICT 0.07 -> 0.17 of the active cycles in k_mel); the first
// exchange of this form therefore pads one slot every 32 points instead of every 16: slot 16 p1 + (p1 >> 1) + r, read back by
// lane (lh, ll) at 16 lh + (lh >> 1) + ll + 66 i (position lane + 64 i = 16 (lh + 4 i) + ll) -- conflict free both ways.
// PREDFT (pair form): the radix-16 butterflies of pass 1 have been run on v already (k_mel does them ahead of the barrier that frees the frame).
// prune3 = 0 .. 2 (wave-uniform): pass 3 forms only the low and mirrored positions (pass3_write_pruned<prune3>).
template <bool ODDROT = false, bool PAIR = false, bool PREDFT = false, class TW>
__device__ __forceinline__ void fft1024_wave(float2* v, float2* buf, int lane, const TW& tw, int p1 = 0, int prune3 = -1) {
    if constexpr (ODDROT) {
        Dft<16>::run(v);
        const float sg = (lane & 1) ? -1.f : 1.f;
        const int pb = phys(lane << 4);
#pragma unroll
        for (int r = 0; r < 16; ++r) buf[pb + r] = (r & 1) ? make_float2(v[r].x * sg, v[r].y * sg) : v[r];   // position 16 lane + r
        frame_sync<64>();
        regs_read<10, 4>(v, buf, lane);
    } else if constexpr (PAIR) {
        if constexpr (!PREDFT) Dft<16>::run(v);
        const int pb = 16 * p1 + (p1 >> 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) buf[pb + r] = v[r];
        frame_sync<64>();
        const int rb = 16 * (lane >> 4) + (lane >> 5) + (lane & 15);
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = buf[rb + 66 * i];
    } else {
        pass1_write(v, buf, lane, tw);   // radix 16, no twiddles: position 16 lane + r
        frame_sync<64>();
        regs_read<10, 4>(v, buf, lane);
    }
    frame_sync<64>();                // every lane has its points of pass 1 before pass 3 overwrites the frame
    float2 a[16];
    {
        float2 w[16];
        pass2_twiddles(w, lane & 15, tw);
        a[0] = v[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) a[r] = cmul(v[r], w[r]);
    }
    Dft<16>::run(a);
#pragma unroll
    for (int rh = 0; rh < 4; ++rh) lane_row_transpose4(a[4 * rh], a[4 * rh + 1], a[4 * rh + 2], a[4 * rh + 3]);
#pragma unroll
    for (int b = 0; b < 4; ++b)   // register 4 b + r' now holds position lane + 64 b + 256 r': pass 3 reads v[b + 4 r']
#pragma unroll
        for (int r = 0; r < 4; ++r) v[b + 4 * r] = a[4 * b + r];
    switch (prune3) {   // (uniform; -1 = all positions)
        case 0: pass3_write_pruned<0>(v, buf, lane, tw); break;
        case 1: pass3_write_pruned<1>(v, buf, lane, tw); break;
        case 2: pass3_write_pruned<2>(v, buf, lane, tw); break;
        default: pass3_write(v, buf, lane, tw);
    }
    frame_sync<64>();
}

// All passes.  Input: v[i] = x[p + i*P].  Output: natural-order spectrum in the
// padded LDS frame `buf` (visible to the frame's threads after the final sync).
// `tw` = table built by the host with zafx_twiddle_layout (LDS or global).
template <int LOG2N, int LOG2E, int LOG2NS = 0>
__device__ __forceinline__ void fft_frame(float2* v, float2* buf, int p, const float2* tw) {
    using C = FftCfg<LOG2N, LOG2E>;
    if constexpr (ZAFX_FFT_PERMLANE && LOG2N == 10 && LOG2E == 4 && LOG2NS == 0) {
        fft1024_wave(v, buf, p, tw);
    } else if constexpr (LOG2NS < LOG2N) {
        constexpr int LR = pass_log2r(LOG2N - LOG2NS, LOG2E);
        pass_write<LOG2N, LOG2E, LOG2NS, LR>(v, buf, p, tw + twiddle_offset(LOG2N, LOG2E, LOG2NS));
        frame_sync<C::P>();
        if constexpr (LOG2NS + LR < LOG2N) {
            regs_read<LOG2N, LOG2E>(v, buf, p);
            frame_sync<C::P>();
            fft_frame<LOG2N, LOG2E, LOG2NS + LR>(v, buf, p, tw);
        }
    }
}

// fft_frame whose last pass leaves conj(X[k] post[k]) in the frame (see pass_write_post)
template <int LOG2N, int LOG2E, int LOG2NS = 0>
__device__ __forceinline__ void fft_frame_post(float2* v, float2* buf, int p, const float2* tw, const float2* post) {
    using C = FftCfg<LOG2N, LOG2E>;
    constexpr int LR = pass_log2r(LOG2N - LOG2NS, LOG2E);
    if constexpr (LOG2NS + LR < LOG2N) {
        pass_write<LOG2N, LOG2E, LOG2NS, LR>(v, buf, p, tw + twiddle_offset(LOG2N, LOG2E, LOG2NS));
        frame_sync<C::P>();
        regs_read<LOG2N, LOG2E>(v, buf, p);
        frame_sync<C::P>();
        fft_frame_post<LOG2N, LOG2E, LOG2NS + LR>(v, buf, p, tw, post);
    } else {
        pass_write_post<LOG2N, LOG2E, LOG2NS, LR>(v, buf, p, tw + twiddle_offset(LOG2N, LOG2E, LOG2NS), post);
        frame_sync<C::P>();
    }
}

// fft_frame with chained twiddles (see pass_write_chain)
template <int LOG2N, int LOG2E, int LOG2NS = 0>
__device__ __forceinline__ void fft_frame_chain(float2* v, float2* buf, int p, const TwoLevelTw& t) {
    using C = FftCfg<LOG2N, LOG2E>;
    if constexpr (ZAFX_FFT_PERMLANE && LOG2N == 10 && LOG2E == 4 && LOG2NS == 0) {
        fft1024_wave(v, buf, p, t);
    } else if constexpr (LOG2NS < LOG2N) {
        constexpr int LR = pass_log2r(LOG2N - LOG2NS, LOG2E);
        pass_write_chain<LOG2N, LOG2E, LOG2NS, LR>(v, buf, p, t);
        frame_sync<C::P>();
        if constexpr (LOG2NS + LR < LOG2N) {
            regs_read<LOG2N, LOG2E>(v, buf, p);
            frame_sync<C::P>();
            fft_frame_chain<LOG2N, LOG2E, LOG2NS + LR>(v, buf, p, t);
        }
    }
}

}  // namespace zafx
#endif
