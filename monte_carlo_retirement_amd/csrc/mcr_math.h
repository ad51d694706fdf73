// mcr_math.h — fp64 math specialised for the path kernel's operand domains (gfx950).
//
// The generic device libm (OCML) pays for ranges and special cases this kernel never sees
// (denormals, overflow, NaN, huge trig arguments).  Measured on MI355X one fp64 add/mul/fma
// costs one VALU issue slot, v_rcp/rsq_f64 ~3.5 slots, an IEEE fp64 division ~13.5 slots, and
// the RNG + growth factors alone were ~45 % of the 725 VALU instructions per path-month.
//
//   fdiv / recip_nr   the IEEE division sequence without v_div_scale/v_div_fixup: bit-identical to
//                     `a / b` whenever no exponent scaling is needed (balances live in 1e-6..1e15);
//                     the Newton reciprocal is shared by quotients with the same divisor.  Inside the path ONE Newton
//                     step (FULL = false): still `a / b` on every sampled pair (a sampled claim, tests/test_gpu_math.py).
//   fexp              exp(x), |x| < 700: 2^(k/512) table (LDS, 4 KB), ONE-constant reduction, degree-4 polynomial:
//                     <= ~1.5 ulp + 0.5 ulp per unit of |x|.
//   neg2_log_u32      -2 ln((x+0.5) 2^-32) straight from the Philox integer: 128-entry table of (1/c, -2 ln c) + degree-6
//                     series, one-constant exponent term: <= 2 ulp of the result (Box-Muller radius^2).
//   fsqrt             sqrt(w) for normal positive w: v_rsq_f64 + ONE coupled Goldschmidt step, no correction: relative error
//                     <= 1e-14 (measured 7.9e-15, ~36 ulp) — the random-number side is sized to the 1e-9 path tolerance.
//   sincos_u32        sin/cos(2 pi (x+0.5) 2^-32): top 8 bits index a 256-entry (sin,cos) table of bin centres, the low 24
//                     bits give |delta| <= pi/256, rotated with short series; absolute error < 3e-16.
// PATH = true (template parameter of fexp / neg2_log_u32 / sincos_u32; used by growth_rows2, i.e. inside the month loop
// only): the series are cut where their truncation error is still far inside the 1e-9 path tolerance — exp with r^2/24
// replaced by its zero-mean fit a^2/40 (|error| <= 3.5e-15 relative, mean 0), the logarithm's series at r^4 (r^5 and r^6 terms, <= 3.6e-13 absolute, dropped), cos
// without dl^6/720 (<= 4.7e-15 absolute): 7.5 instructions per path-month less (7.00 -> 6.81 ms per 1e6 paths), worst
// path-level error against the oracle 1.6e-11 of the 1e-9 allowed, relative to the path's money scale (4 x 2e6 paths, 0 Success
// flags flipped; profiles/r03/k1_accuracy_2e6_paths.txt, tools/k1_accuracy.py).  The unit-function entry points and the
// shock-row API keep the full forms.
// (All bounds measured on the device: tests/test_gpu_math.py.)
// Tables are correctly rounded (tools/gen_tables.py) and staged in LDS once per workgroup.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mcr_tables.h"

namespace mcr {

constexpr int kMathTabBytes = kTabDoubles * (int)sizeof(double);  // 10 240 B of LDS

// every thread of a kBlockThreads-wide workgroup calls this once; caller syncs afterwards
__device__ __forceinline__ void load_math_tables(double* lds_tab, int tid, int nthreads) {
    for (int i = tid; i < kTabDoubles; i += nthreads) lds_tab[i] = kMathTab[i];
}

// 1/b: v_rcp_f64 seed + Newton steps.  FULL = two steps (the core of LLVM's IEEE fdiv lowering; with div_by below the
// quotient is bit-identical to `a / b`).  FULL = false = ONE step, used inside the path: y is then good to ~2^-50 and
// the residual correction of div_by squares that, so the quotient is still the correctly rounded one except when
// a / b lies within ~2^-100 (relative) of a rounding boundary — never observed (tests/test_gpu_math.py).
template <bool FULL = true>
__device__ __forceinline__ double recip_nr(double b) {
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    if (FULL) {
        e = __builtin_fma(-b, y, 1.0);
        y = __builtin_fma(y, e, y);
    }
    return y;
}
// a / b given y = recip_nr(b): quotient estimate + one residual correction (== v_div_fmas unscaled)
__device__ __forceinline__ double div_by(double a, double b, double y) {
    const double q = a * y;
    const double r = __builtin_fma(-b, q, a);
    return __builtin_fma(r, y, q);
}
template <bool FULL = true>
__device__ __forceinline__ double fdiv(double a, double b) { return div_by(a, b, recip_nr<FULL>(b)); }

// The addends of the two-constant FMAs below (p = r c1 + c0).  A VOP3 instruction reads at most ONE scalar operand on
// gfx9, so with both constants in SGPRs the compiler copies one into a VGPR pair in front of every such FMA (a
// v_mov_b64, or two v_mov_b32, per evaluation).  The path kernel keeps these few in VGPRs for the whole launch
// (pinned(): an empty asm the register allocator cannot see through); everyone else passes literals().
struct MathRegs {
    double exp_c6;     // 1/6     (fexp); the PATH forms hold 1/2 + a^2/40 here instead (kExpPathHalf)
    double log_c0;     // -3.2    (neg2_log_u32)
    double sin_c6;     // -1/6    (sincos_u32)
    double cos_c24;    // 1/24
    double ang_bias;   // kAngleBias
    static __device__ __forceinline__ MathRegs literals();
    static __device__ __forceinline__ MathRegs pinned();
    static __device__ __forceinline__ MathRegs literals_path();   // the register set the PATH = true forms expect
    static __device__ __forceinline__ MathRegs pinned_path();
};

constexpr double kExpStep = 0x1.62e42fefa39efp-10;   // ln 2 / 512, correctly rounded
constexpr double kExpPathHalf = 0.5 + (0x1.62e42fefa39efp-11 * 0x1.62e42fefa39efp-11) / 40.0;   // 1/2 + a^2/40, a = ln 2 / 1024
constexpr double kM2Ln2 = -0x1.62e42fefa39efp+0;     // -2 ln 2, correctly rounded

template <bool PATH = false>
__device__ __forceinline__ double fexp(double x, const double* tab, const MathRegs& R) {
    // k = rint(x 512/ln2) by the shifter trick: the sum lands on the unit grid of [2^52, 2^53), so its low word IS k
    // (two's complement) and subtracting the shifter gives k as a double: no v_rndne / v_cvt_i32.
    const double shifted = __builtin_fma(x, kExpScale, 6755399441055744.0);   // 1.5 * 2^52
    const int k = (int)(uint32_t)(uint64_t)__double_as_longlong(shifted);
    const double kf = shifted - 6755399441055744.0;
    // one-constant reduction: the step's representation error (1.1e-16 relative) times |k| step ~ |x| adds 1.1e-16 |x| to
    // r, i.e. half an ulp of the result per unit of |x| — monthly log-returns are |x| < 2 (a two-constant Cody-Waite
    // reduction only pays for arguments this kernel never sees)
    const double r = __builtin_fma(-kf, kExpStep, x);
    const double t = tab[kTabExp2 + (k & ((1 << kExp2Bits) - 1))];
    // e^r - 1 = r + r^2 (1/2 + r/6 + r^2/24), |r| <= ln2/1024: the next term r^5/120 < 1.2e-18
    double p;
    if (PATH) {
        // path form: r^2/24 is replaced by its mean-square fit a^2/40 (a = ln 2 / 1024, the half-width of r): the error of
        // e^r - 1, r^4/24 - a^2 r^2/40, has ZERO mean over the reduction interval and |.| <= a^4/60 = 3.5e-15.  (Simply
        // dropping r^4/24 is one-signed: every growth factor low by 1.7e-15 on average, and a 2040-month path with a long
        // drawdown amplified that past the 1e-9 path tolerance — tests/test_gpu_edge_cases.py::test_very_long_horizon.)
        p = __builtin_fma(r, 1.0 / 6.0, R.exp_c6);           // exp_c6 = 1/2 + a^2/40 here (MathRegs::pinned_path / literals_path)
    } else {
        p = __builtin_fma(r, 1.0 / 24.0, R.exp_c6);
        p = __builtin_fma(r, p, 0.5);
    }
    p = __builtin_fma(r * r, p, r);
    // t (1 + p) 2^(k >> 9): the scale goes straight into the exponent field of the high word (v_ashr + v_lshl_add_u32,
    // both 32-bit) instead of v_ashr + v_ldexp_f64.  Exact while the result is a normal number (|x| < 700).
    const double v = __builtin_fma(t, p, t);
    const uint64_t vb = (uint64_t)__double_as_longlong(v);
    uint32_t hi;
    asm("v_lshl_add_u32 %0, %1, 20, %2" : "=v"(hi) : "v"(k >> kExp2Bits), "v"((uint32_t)(vb >> 32)));  // (the compiler splits it in three)
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | (vb & 0xFFFFFFFFull)));
}

template <bool PATH = false>
__device__ __forceinline__ double neg2_log_u32(uint32_t x, const double* tab, const MathRegs& R) {
    const double d = __builtin_fma((double)x, 2.0, 1.0);  // 2x+1, exact, in [1, 2^33);  u = d 2^-33
    const uint32_t hi = (uint32_t)((uint64_t)__double_as_longlong(d) >> 32);
    const int i = (int)((hi >> 13) & 127u);               // top 7 fraction bits
    const double m = __builtin_amdgcn_frexp_mant(d);      // d = m 2^ex, m in [1/2, 1): v_frexp_mant_f64 / v_frexp_exp_i32_f64
    const int ex = __builtin_amdgcn_frexp_exp(d);
    const double inv_c = tab[kTabLog + 2 * i], w_i = tab[kTabLog + 2 * i + 1];   // 1/c_i and -2 ln c_i, c_i in [1, 2)
    // r = 2m / c_i - 1, |r| <= 2^-8; the code works with s = r/2 = m / c_i - 1/2 (no mantissa rebuilt in integer ops):
    // -2 ln(1+r) = -2r + r^2 (1 - 2r/3 + r^2/2 - 2r^3/5 + r^4/3) = 4 (s^2 p(2s) - s); every coefficient below is a
    // power-of-two multiple of the series', so the Horner values are the same bits as in terms of r
    const double sh = __builtin_fma(m, inv_c, -0.5);
    double p;
    if (PATH) {                                   // path form: the series stops at r^4/2 (r^5 and r^6 terms <= 3.6e-13 absolute dropped)
        p = __builtin_fma(sh, 2.0, -4.0 / 3.0);
    } else {
        p = __builtin_fma(sh, 16.0 / 3.0, R.log_c0);
        p = __builtin_fma(sh, p, 2.0);
        p = __builtin_fma(sh, p, -4.0 / 3.0);
    }
    p = __builtin_fma(sh, p, 1.0);
    const double q4 = __builtin_fma(sh * sh, p, -sh);      // (-2 ln(1+r)) / 4
    const double e = (double)(ex - 34);                   // u = (2m) 2^(ex - 34)
    // e (-2 ln 2) + w_i in one FMA: the constant's representation error is 8e-17 of the product, and the product is the
    // bulk of the result whenever |e| > 1 (for e = -1 it is 1.1e-16 absolute)
    return __builtin_fma(4.0, q4, __builtin_fma(e, kM2Ln2, w_i));
}

__device__ __forceinline__ double fsqrt(double w) {  // w normal, > 0
    // v_rsq_f64 seed + one coupled Goldschmidt step: the seed's relative error e becomes 1.5 e^2
    const double y = __builtin_amdgcn_rsq(w);
    const double g = w * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    return __builtin_fma(g, r, g);
}

template <bool WANT_SIN, bool PATH = false>
__device__ __forceinline__ void sincos_u32(uint32_t x, const double* tab, const MathRegs& R, double& s, double& c) {
    const int k = (int)(x >> 24);
    const double S = tab[kTabSinCos + 2 * k], C = tab[kTabSinCos + 2 * k + 1];
    const double dl = __builtin_fma((double)(x & 0x00FFFFFFu), kAngleScale, R.ang_bias);  // |dl| <= pi/256
    const double d2 = dl * dl;
    // (dropping sin's dl^5/120 <= 2.3e-12 as well was measured in round 3: 3.3 instead of 2.8 % faster, path error 2.4e-10 of
    //  the 1e-9 allowed on 2e5 paths — too little margin for longer soaks: not taken, LABNOTES.md)
    const double sd = __builtin_fma(dl * d2, __builtin_fma(d2, 1.0 / 120.0, R.sin_c6), dl);  // sin(dl)
    double cp;
    if (PATH) {                                   // path form: dl^6/720 <= 4.7e-15 dropped
        cp = __builtin_fma(d2, R.cos_c24, -0.5);
    } else {
        cp = __builtin_fma(d2, -1.0 / 720.0, R.cos_c24);
        cp = __builtin_fma(d2, cp, -0.5);
    }
    const double cd = d2 * cp;                                                               // cos(dl) - 1
    const double cosd = 1.0 + cd;                                                            // (shared by both outputs)
    c = __builtin_fma(C, cosd, -(S * sd));
    if (WANT_SIN) s = __builtin_fma(S, cosd, C * sd);
}

__device__ __forceinline__ MathRegs MathRegs::literals() { return MathRegs{1.0 / 6.0, -3.2, -1.0 / 6.0, 1.0 / 24.0, kAngleBias}; }
__device__ __forceinline__ MathRegs MathRegs::pinned() {
    MathRegs R = literals();
    asm volatile("" : "+v"(R.exp_c6), "+v"(R.log_c0), "+v"(R.sin_c6), "+v"(R.cos_c24), "+v"(R.ang_bias));
    return R;
}
__device__ __forceinline__ MathRegs MathRegs::literals_path() {
    MathRegs R = literals();
    R.exp_c6 = kExpPathHalf;
    return R;
}
__device__ __forceinline__ MathRegs MathRegs::pinned_path() {    // (log_c0 is not read by the PATH forms: left to the compiler)
    MathRegs R = literals_path();
    asm volatile("" : "+v"(R.exp_c6), "+v"(R.sin_c6), "+v"(R.cos_c24), "+v"(R.ang_bias));
    return R;
}

}  // namespace mcr
