// mcr_device.h — device-side building blocks of the Monte Carlo retirement path kernel (gfx950).
//
// One simulated path lives in ONE LANE: the monthly state (two balances, two cost bases, two
// gain accumulators, the price level, the contribution) stays in VGPRs for the whole horizon;
// scenario parameters are wave-uniform kernel arguments (SGPRs).  All arithmetic is IEEE fp64
// and the translation unit is compiled with -ffp-contract=off: the STATE MACHINE's a*b+c rounds
// twice, as in the reference (CPython floats); FMAs appear only where they are spelled out
// (__builtin_fma / inline v_fma_f64): inside exp / log / sincos, the Newton steps of the
// divisions, and the log-returns a + b z of monthly_gross / growth_rows2 (one rounding there
// instead of the reference's two, 1e-16 |x| on the argument of exp: part of the 1e-9 budget).
// Each function cites the reference lines it replaces (rflamino/monte_carlo_retirement,
// backend/simulation.py).
//
// Data-dependent branches of the reference (sell inv1 / sell inv2, early-outs) are written
// branch-free with selected operands: the lanes of a wavefront diverge on them almost every
// month, so both sides would otherwise be executed under exec masks.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mcr.h"
#include "mcr_math.h"

// `if (c) { MCR_MASKED_MOVE; x = y; }`: the empty asm keeps the assignment a BRANCH under the exec mask (v_cmp, s_and_saveexec,
// moves, s_or) where the compiler would if-convert it into v_cndmask pairs — a 64-bit select costs two fp64 issue slots, the
// masked move one and often none (LABNOTES.md, rounds 1-3 section 5).  MM (a template parameter in scope at every use) = false leaves the
// choice to the compiler: the producer / consumer SPLIT launches are bound by each wave's dependency chain, not by issue
// slots, and there the compare -> scalar mask -> branch -> move round trip is the slower form (lone 50 000-path probe
// 0.84 -> 0.79 ms, measured).  Same values either way.
#define MCR_MASKED_MOVE do { if (MM) asm volatile(""); } while (0)
namespace mcr {

constexpr double kEps = MCR_SMALL_EPSILON;
constexpr int kMPY = MCR_MONTHS_PER_YEAR;
constexpr int kBlock = 256;  // threads per workgroup = 4 wavefronts, one per SIMD (128 measured: no better)

// Per-stream data prepared on the host (mcr_abi: derive_params) from mcr_stream.
struct DevStream {
    double amount;        // monthly_amount_today
    double keep;          // 1.0 - tax_rate                      (simulation.py:675-677)
    double amount_keep;   // amount * keep: the tolerance form of the month nets the amount first ((a k) price instead of (a price) k)
    int32_t start_month;  // stream_payment_start_month_index    (:47-63, :603-608)
    int32_t end_month;    // start + duration_years*12, INT32_MAX for None (:609-613, :653-656)
    int32_t indexed;      // inflation_indexed
    int32_t lock_slot;    // column of the frozen nominal amount (non-indexed streams), else -1: slots below
                          // DevParams::n_lock_slots are LDS columns, the others rows of DevParams::lock_overflow
};
// Records past the by-value block live in a device table that the kernel reads through the CONSTANT address space:
// scalar loads (s_load_dwordx8, the record lands in SGPRs like a kernel argument) whatever the alias analysis makes of the
// kernel's own stores.
typedef const __attribute__((address_space(4))) DevStream* DevStreamTable;

// Wave-uniform scenario block (kernel argument -> SGPRs).
struct DevParams {
    double initial_balance, monthly_contribution, contrib_growth_factor /* 1 + g */, monthly_expenses;
    double alloc1, alloc2;             // allocation_inv1_pct, 1.0 - allocation_inv1_pct (config.py:124-126)
    double real_rate1, real_rate2;     // realized-gains rate if the asset uses that system, else 0.0
    double annual_rate1, annual_rate2; // annual-gains rate if the asset does NOT use the realized system, else 0.0
    double a1, b1;                     // inv1:      mu_log/12, sigma_log/sqrt(12)   (:473)
    double ainf, binf;                 // inflation: idem
    double aprem, bprem;               // inv2 premium over inflation: idem
    double rho, rho_c;                 // rho, sqrt(max(0, 1 - rho^2))               (:460-464)
    double binf_rho, binf_rho_c;       // binf * rho, binf * rho_c: the inflation log-return straight from two normals (growth_rows2)
    int32_t working_months, retirement_years, total_months, shock_rows;
    int32_t num_working_years, trajectory_len;
    int32_t n_streams;                 // records in `streams` below: min(len(other_income_streams), MCR_INLINE_STREAMS)
    int32_t n_lock_slots;              // lock slots kept in LDS ([n_lock_slots][kBlock] doubles); the launcher decides (LDS budget)
    int32_t contrib_grows;             // contribution_growth_rate_annual > 0        (:516)
    int32_t any_annual_tax;            // annual_rate1 > 0 || annual_rate2 > 0
    int32_t any_real_rate;             // real_rate1 > 0 || real_rate2 > 0
    int32_t tax_mask;                  // bit 0: real_rate1 > 0, bit 1: real_rate2 > 0  -> selects the TAXED kernel variant (0 .. 3)
    int32_t exact_month;               // a realized-gains rate above 1 - 1e-6: the reference's denominator clamps can bind -> exact forms, generic variants
    int32_t reserved_;
    DevStream streams[MCR_INLINE_STREAMS];
    // other_income_streams beyond the by-value block (the list has any length, config.py:99)
    int32_t n_extra_streams;           // records in `extra_streams`
    int32_t n_lock_slots_total;        // non-indexed streams of the whole list (derive_params numbers their slots in list order)
    const DevStream* extra_streams;    // DEVICE table [n_extra_streams] (read through DevStreamTable), nullptr if none
    double* lock_overflow;             // DEVICE [n_lock_slots_total - n_lock_slots][lock_stride]: the slots that did not fit in LDS
    int64_t lock_stride;               // = grid.x * kBlock: every lane of the launch has its own column
};

// The four parameters that feed per-lane SELECTS (seller's weight / seller's rate).  They are
// wave-uniform, but a select between two SGPR pairs costs two v_mov per dword (one constant-bus
// operand per VALU instruction) and the kernel is already SGPR-bound (57 spills): the path kernel keeps
// a VGPR-resident copy.
struct LaneParams {
    double alloc1, alloc2, real_rate1, real_rate2;
};
__device__ __forceinline__ LaneParams lane_params(const DevParams& P) {
    LaneParams L{P.alloc1, P.alloc2, P.real_rate1, P.real_rate2};
    asm volatile("" : "+v"(L.alloc1), "+v"(L.alloc2), "+v"(L.real_rate1), "+v"(L.real_rate2));
    return L;
}
// The tolerance form of the month (below) selects the seller's rate and the seller's weight x rate; it needs neither weight
// on its own (the drift of asset 2 is minus the drift of asset 1).  In the SAME struct, so that the helpers keep one
// signature: alloc1 / alloc2 then hold the products.
__device__ __forceinline__ LaneParams lane_params_tol(const DevParams& P) {
    LaneParams L{P.alloc1 * P.real_rate1, P.alloc2 * P.real_rate2, P.real_rate1, P.real_rate2};
    asm volatile("" : "+v"(L.alloc1), "+v"(L.alloc2), "+v"(L.real_rate1), "+v"(L.real_rate2));
    return L;
}

// ---------------------------------------------------------------------------------------------
// RNG: Philox4x32-10 (Salmon et al. SC'11, Random123 constants).  Counter = (path_lo, path_hi,
// month, stream_id), key = (seed_lo, seed_hi).  The key schedule is wave-uniform (scalar ALU).
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
    // The key is wave-uniform.  Left alone, the compiler hoists all ten round keys out of the month loop and then
    // spills them (18 SGPRs -> v_readlane + s_nop per use); behind this barrier it bumps the key with two scalar adds
    // per round instead, which are free next to the VALU work.
    asm volatile("" : "+s"(k0), "+s"(k1));
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        // hi ^ counter ^ key as ONE v_bitop3_b32 (truth table 0x96 = three-way xor; gfx950) once all three operands are
        // per-lane values — in the first rounds parts are still wave-uniform and fold into scalar xors
        const uint32_t n0 = r >= 2 ? __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96) : (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = r >= 1 ? __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96) : (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// The path's standard-normal sequence n[0], n[1], ... : pair j = Box-Muller of Philox words
// (2(j&1), 2(j&1)+1) of block j>>1 (counter = (path_lo, path_hi, block, stream_id)):
//   n[2j] = sqrt(-2 ln u_r) cos(2 pi u_a), n[2j+1] = ... sin(...),  u = (x + 0.5) 2^-32.
// Shock row k = (n[3k], rho n[3k] + rho_c n[3k+1], n[3k+2])  (replaces _draw_shock_path, :452-466),
// a pure function of (seed, stream, path, k): common random numbers across working-month candidates
// and across any sharding.  Rows are consumed in order, so a lane carries the half-used pair / block:
// 4 rows cost 3 Philox blocks and 6 (log, sqrt, sincos) pairs instead of 4 and 8.
__device__ __forceinline__ void bm_pair(uint32_t xr, uint32_t xa, const double* tab, const MathRegs& R, double& zc, double& zs) {
    const double r = fsqrt(neg2_log_u32(xr, tab, R));  // radius and angle straight from the Philox integers
    double s, c;
    sincos_u32<true>(xa, tab, R, s, c);
    zc = r * c;
    zs = r * s;
}

// The same pair as its PARTS: radius r and the angle's (cos, sin), the two normals being r cos and r sin.  growth_rows2
// folds the products into the log-returns instead of forming the normals first.
__device__ __forceinline__ void bm_parts(uint32_t xr, uint32_t xa, const double* tab, const MathRegs& R, double& r, double& c, double& s) {
    r = fsqrt(neg2_log_u32<true>(xr, tab, R));      // path forms of the series (mcr_math.h)
    sincos_u32<true, true>(xa, tab, R, s, c);
}

struct ShockGen {      // per-lane carry between consecutive rows
    double carry_z;    // second normal of a pair whose first was used by the previous row
    uint32_t cw0, cw1; // words 2,3 of the last Philox block (the next pair)
};

// Row `k` for k % 4 == PHASE, given that rows 0..k-1 of this path were generated by this ShockGen in order.
// PHASE is a template parameter: which words feed which pair, whether a Philox block is drawn and what is carried
// are then fixed at compile time — no per-row selects or carry shuffles between the four cases (the caller
// dispatches on the wave-uniform k & 3 with scalar branches).
//   PHASE 0: block 3t   -> A = words 0,1, B = words 2,3          row = (A.cos, A.sin, B.cos), carry B.sin
//   PHASE 1: block 3t+1 -> A = words 0,1; words 2,3 carried      row = (carried, A.cos, A.sin)
//   PHASE 2: block 3t+2 -> A = carried words, B = words 0,1      row = (A.cos, A.sin, B.cos), carry B.sin, words 2,3
//   PHASE 3: no block   -> A = carried words                     row = (carried, A.cos, A.sin)
template <int PHASE>
__device__ __forceinline__ void shock_row_phase(ShockGen& G, uint64_t seed, uint32_t stream_id, uint64_t path,
                                                uint32_t k, double rho, double rho_c, const double* tab,
                                                double& z_eq, double& z_inf, double& z_prem) {
    uint32_t x[4] = {0u, 0u, 0u, 0u};
    if (PHASE != 3)
        philox4x32_10((uint32_t)path, (uint32_t)(path >> 32), 3u * (k >> 2) + (uint32_t)PHASE, stream_id, (uint32_t)seed,
                      (uint32_t)(seed >> 32), x);
    const uint32_t ar = PHASE < 2 ? x[0] : G.cw0, aa = PHASE < 2 ? x[1] : G.cw1;
    double ac, as;
    bm_pair(ar, aa, tab, MathRegs::literals(), ac, as);
    double n0, n1, n2;
    if ((PHASE & 1) == 0) {
        const uint32_t br = PHASE == 0 ? x[2] : x[0], ba = PHASE == 0 ? x[3] : x[1];
        double bc, bs;
        bm_pair(br, ba, tab, MathRegs::literals(), bc, bs);
        n0 = ac; n1 = as; n2 = bc;
        G.carry_z = bs;
    } else {
        n0 = G.carry_z; n1 = ac; n2 = as;
    }
    if (PHASE == 1 || PHASE == 2) { G.cw0 = x[2]; G.cw1 = x[3]; }
    z_eq = n0;
    z_inf = __builtin_fma(rho, n0, rho_c * n1);
    z_prem = n2;
}

// Row `k`, given that rows 0..k-1 of this path were generated by this ShockGen in order (k is wave-uniform).
__device__ __forceinline__ void shock_row_seq(ShockGen& G, uint64_t seed, uint32_t stream_id, uint64_t path,
                                              uint32_t k, double rho, double rho_c, const double* tab,
                                              double& z_eq, double& z_inf, double& z_prem) {
    switch (__builtin_amdgcn_readfirstlane((int)(k & 3u))) {
        case 0: shock_row_phase<0>(G, seed, stream_id, path, k, rho, rho_c, tab, z_eq, z_inf, z_prem); break;
        case 1: shock_row_phase<1>(G, seed, stream_id, path, k, rho, rho_c, tab, z_eq, z_inf, z_prem); break;
        case 2: shock_row_phase<2>(G, seed, stream_id, path, k, rho, rho_c, tab, z_eq, z_inf, z_prem); break;
        default: shock_row_phase<3>(G, seed, stream_id, path, k, rho, rho_c, tab, z_eq, z_inf, z_prem); break;
    }
}

// _monthly_gross_from_shock (:468-474) with a = mu_log/12 and b = sigma_log/sqrt(12) precomputed.
__device__ __forceinline__ double monthly_gross(double a, double b, double z, const double* tab, const MathRegs& R = MathRegs::literals()) {
    return fexp(__builtin_fma(b, z, a), tab, R);   // (one rounding instead of the reference's two: 1e-16 |x| on the argument)
}

// TWO consecutive months at once.  Rows 4t .. 4t+3 of a path use exactly Philox blocks 3t .. 3t+2 -> Box-Muller
// pairs P0 .. P5 (normals n[12t + 2i], n[12t + 2i + 1] = Pi.cos, Pi.sin), so with a PAIR of rows as the unit which
// word feeds which pair is fixed at compile time by the half (no per-row selects, nothing shuffled between rows):
//   HALF 0 (rows 4t, 4t+1):   blocks 3t and 3t+1;  P0 = (w0,w1), P1 = (w2,w3) of 3t, P2 = (w0,w1) of 3t+1,
//                             words 2,3 of 3t+1 are carried;   rows = (P0.c, P0.s, P1.c), (P1.s, P2.c, P2.s)
//   HALF 1 (rows 4t+2, 4t+3): block 3t+2;  P3 = carried words, P4 = (w0,w1), P5 = (w2,w3);
//                             rows = (P3.c, P3.s, P4.c), (P4.s, P5.c, P5.s)
// The path kernel calls this at every even row (wave-uniform) and stages the two months' gross factors
// (g1, g_inflation, g2 = g_inflation * g_premium; :522-532) in the lane's own LDS column:
// `stage[j * kBlock]`, j = 3 * (row & 1) + {0, 1, 2} (12 KB per workgroup).  The six pairs of a HALF are
// independent straight-line work (lots of instruction-level parallelism next to the serial month bodies of
// the other waves).  The normals are shock_row_seq's; the log-returns are associated on their parts (below).
struct PairCarry { uint32_t w2, w3; };
template <int HALF>
__device__ __forceinline__ void growth_rows2(const DevParams& P, const MathRegs& M, uint64_t seed, uint32_t stream_id, uint64_t path,
                                             uint32_t t, const double* tab, double* stage, PairCarry& C) {
    // pair i: radius rad[i], trig[2 i] = cos, trig[2 i + 1] = sin; normal j of the half = rad[j >> 1] * trig[j]
    double rad[3], trig[6];
    uint32_t x[4];
    if (HALF == 0) {
        philox4x32_10((uint32_t)path, (uint32_t)(path >> 32), 3u * t, stream_id, (uint32_t)seed, (uint32_t)(seed >> 32), x);
        bm_parts(x[0], x[1], tab, M, rad[0], trig[0], trig[1]);
        bm_parts(x[2], x[3], tab, M, rad[1], trig[2], trig[3]);
        philox4x32_10((uint32_t)path, (uint32_t)(path >> 32), 3u * t + 1u, stream_id, (uint32_t)seed, (uint32_t)(seed >> 32), x);
        bm_parts(x[0], x[1], tab, M, rad[2], trig[4], trig[5]);
        C.w2 = x[2]; C.w3 = x[3];
    } else {
        bm_parts(C.w2, C.w3, tab, M, rad[0], trig[0], trig[1]);
        philox4x32_10((uint32_t)path, (uint32_t)(path >> 32), 3u * t + 2u, stream_id, (uint32_t)seed, (uint32_t)(seed >> 32), x);
        bm_parts(x[0], x[1], tab, M, rad[1], trig[2], trig[3]);
        bm_parts(x[2], x[3], tab, M, rad[2], trig[4], trig[5]);
    }
    // The three log-returns of a month, x = a + b z (:473) with z_inf = rho n0 + rho_c n1 (:461-464), written on the
    // PARTS of the normals: x_eq = (b1 r) t + a1, x_inf = (binf rho r0) t0 + ((binf rho_c r1) t1 + a_inf), x_prem
    // likewise — the same eight fp64 operations as "normals first", but every FMA now has ONE scalar operand (a VOP3
    // instruction reads at most one: b z + a with a and b both in SGPRs cost two v_mov_b32 per evaluation, six a month).
    // The product is associated differently from shock_row_seq's (b (r t) vs (b r) t): the arguments of exp agree to
    // ~1e-17, a tenth of an ulp of the growth factor.
    // (v_fma_f64 with the SCALAR addend spelled out: left to itself the compiler picks the two-address v_fmac_f64 and
    //  copies the scalar into its accumulator first — the very two v_mov_b32 this arrangement is there to avoid)
    auto fma_vvs = [](double a, double b, double c_scalar) {
        double d;
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c_scalar));
        return d;
    };
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int je = 3 * r, ji = 3 * r + 1, jp = 3 * r + 2;
        const double x_eq = fma_vvs(P.b1 * rad[je >> 1], trig[je], P.a1);
        const double x_inf = __builtin_fma(P.binf_rho * rad[je >> 1], trig[je], fma_vvs(P.binf_rho_c * rad[ji >> 1], trig[ji], P.ainf));
        const double x_prem = fma_vvs(P.bprem * rad[jp >> 1], trig[jp], P.aprem);
        const double g1 = fexp<true>(x_eq, tab, M);
        const double ginf = fexp<true>(x_inf, tab, M);
        const double gprem = fexp<true>(x_prem, tab, M);
        stage[(3 * r + 0) * kBlock] = g1;
        stage[(3 * r + 1) * kBlock] = ginf;
        stage[(3 * r + 2) * kBlock] = ginf * gprem;                             // :532
    }
}
constexpr int kStageDoubles = 6 * kBlock;   // 12 KB of LDS per workgroup

// STRICT vs path form of the helpers.  Inside the path kernel the state obeys 0 <= balance,
// 0 <= cost basis, 0 <= rate <= 1 and "amount sold <= amount held" by construction, which makes several
// clamps of the reference exact no-ops in IEEE arithmetic (x <= y => fl(y - x) >= 0; f <= 1, c >= 0 =>
// fl(c f) <= c; fl(x / y) <= 1 for x <= y).  STRICT = true keeps every clamp for arbitrary inputs (the
// unit-function entry point, bit-exact vs the reference on its test vectors); STRICT = false drops the
// provable no-ops — identical results on every reachable state, ~3 % fewer VALU instructions.

// _net_liquidation_value (:256-272); rate = realized rate if that system applies else 0.0.
// TAXED = false is the compile-time variant for scenarios in which NEITHER asset has an effective
// realized-gains rate (the reference's DEFAULT configuration, config.py:74-75,80-81).  With rate == 0 every
// product with it is exactly +0, t / 1 = t, so the untaxed forms are bit-identical to the general ones and
// skip the dead arithmetic (two of the three divisions in withdraw / rebalance): +13 % paths/s there.
// (A run-time branch instead cost the taxed kernel 4 %, hence a template parameter.)
template <bool STRICT = true, bool TAXED = true, bool MM = true>
__device__ __forceinline__ double net_liquidation_value(double bal, double cb, double rate) {
    double v = bal;  // rate == 0: tax = 0, max(0, bal - 0) = bal for bal > eps
    if (TAXED) {
        const double tax = fmax(0.0, bal - cb) * rate;
        v = STRICT ? fmax(0.0, bal - tax) : bal - tax;  // tax <= bal when cb >= 0, rate <= 1
    }
    if (STRICT) return bal <= kEps ? 0.0 : v;
    if (bal <= kEps) { MCR_MASKED_MOVE; v = 0.0; }   // path form: exec-masked move instead of a select (see withdraw)
    return v;
}

// _calculate_withdrawal_and_update (:201-254): every lane evaluates the arithmetic; only the final selections differ by form.
template <bool STRICT = true, bool TAXED = true, bool MM = true>
__device__ __forceinline__ void withdraw(double& bal, double& cb, double net_target, double rate,
                                         double& gross_out, double& net_out) {
    const bool skip = (bal <= kEps) || (net_target <= 0.0);              // :218
    const double inv_bal = recip_nr<STRICT>(bal);                        // shared by the two divisions by bal
    double gross;
    if (TAXED) {
        const double gain_fraction = div_by(fmax(0.0, bal - cb), bal, inv_bal);  // :221
        const double net_fraction = fmax(kEps, 1.0 - gain_fraction * rate);  // :222-227
        gross = fmin(fdiv<STRICT>(net_target, net_fraction), bal);       // :228-231
    } else {
        gross = fmin(net_target, bal);                                   // net_fraction = max(eps, 1 - g*0) = 1; t / 1 = t
    }
    const double sold = div_by(gross, bal, inv_bal);
    const double fraction_sold = STRICT ? fmin(1.0, sold) : sold;        // :233  (gross <= bal)
    const double basis_part = cb * fraction_sold;
    const double basis_removed = STRICT ? fmin(cb, basis_part) : basis_part;  // :234  (fraction <= 1, cb >= 0)
    double net_cash = gross;                                             // rate == 0: tax_paid = 0, max(0, gross) = gross
    if (TAXED) {
        const double taxable_gain = fmax(0.0, gross - basis_removed);    // :235
        const double tax_paid = taxable_gain * rate;                     // :236-240
        net_cash = STRICT ? fmax(0.0, gross - tax_paid) : gross - tax_paid;  // :241 (tax <= gross)
    }
    double nb = STRICT ? fmax(0.0, bal - gross) : bal - gross;           // :243
    double ncb = STRICT ? fmax(0.0, cb - basis_removed) : cb - basis_removed;  // :244
    const bool dust = nb <= kEps;                                        // :245-247
    if (STRICT) {
        nb = dust ? 0.0 : nb;
        ncb = dust ? 0.0 : ncb;
        bal = skip ? fmax(0.0, bal) : nb;                                // :219
        cb = skip ? fmax(0.0, cb) : ncb;
        gross_out = skip ? 0.0 : gross;
        net_out = skip ? 0.0 : net_cash;
    } else {
        // Path form: the same selections as exec-masked moves.  A 64-bit select is two v_cndmask at a full issue slot
        // each; a move under an exec mask costs about half of that, and the scalar mask bookkeeping is free next to
        // the VALU work.  (The empty asm keeps the compiler from converting the branches back into selects.)
        double rb = nb, rc = ncb, rg = gross, rn = net_cash;
        if (dust) { MCR_MASKED_MOVE; rb = 0.0; rc = 0.0; }
        if (skip) { MCR_MASKED_MOVE; rb = bal; rc = cb; rg = 0.0; rn = 0.0; }   // bal, cb >= 0 already
        bal = rb; cb = rc; gross_out = rg; net_out = rn;
    }
}

// Path form of TWO independent evaluations of the same helper (the two assets of one month): all the arithmetic of
// both first — two independent dependency chains the scheduler can interleave —, then the exec-masked fix-ups
// (each branch ends a basic block; placed between the two chains they would serialise them).
// (T1 / T2: does asset 1 / 2 carry an effective realized-gains rate?  Per asset since round 3: the reference's DEFAULT
//  configuration taxes realized gains on inv2 only, config.py:74-75,80-81 — with a rate of exactly 0 every product with it is
//  +0 and t / 1 = t, so the untaxed form of ONE asset is bit-identical too and skips its dead divisions.)
template <bool T1, bool T2 = T1, bool MM = true>
__device__ __forceinline__ void net_liquidation_values2(double b1, double c1, double r1, double b2, double c2, double r2,
                                                        double& v1, double& v2) {
    v1 = b1; v2 = b2;
    if (T1) v1 = b1 - fmax(0.0, b1 - c1) * r1;                   // tax <= bal when cb >= 0, rate <= 1
    if (T2) v2 = b2 - fmax(0.0, b2 - c2) * r2;
    if (b1 <= kEps) { MCR_MASKED_MOVE; v1 = 0.0; }
    if (b2 <= kEps) { MCR_MASKED_MOVE; v2 = 0.0; }
}

struct WithdrawCand { double nb, ncb, gross, net; };   // (the dust / skip tests are evaluated at the fix-up: a compare there
                                                       //  is one instruction, a flag carried across the other chain's blocks is two)
template <bool TAXED>
__device__ __forceinline__ WithdrawCand withdraw_arith(double bal, double cb, double net_target, double rate) {
    WithdrawCand w;
    const double inv_bal = recip_nr<false>(bal);
    if (TAXED) {
        const double gain_fraction = div_by(fmax(0.0, bal - cb), bal, inv_bal);  // :221
        const double net_fraction = fmax(kEps, 1.0 - gain_fraction * rate);  // :222-227
        w.gross = fmin(fdiv<false>(net_target, net_fraction), bal);      // :228-231
    } else {
        w.gross = fmin(net_target, bal);
    }
    const double fraction_sold = div_by(w.gross, bal, inv_bal);         // :233  (gross <= bal)
    const double basis_removed = cb * fraction_sold;                    // :234  (fraction <= 1, cb >= 0)
    w.net = w.gross;
    if (TAXED) {
        const double taxable_gain = fmax(0.0, w.gross - basis_removed); // :235
        w.net = w.gross - taxable_gain * rate;                          // :236-241 (tax <= gross)
    }
    w.nb = bal - w.gross;                                               // :243
    w.ncb = cb - basis_removed;                                         // :244
    return w;
}
template <bool MM = true>
__device__ __forceinline__ void withdraw_fixup(WithdrawCand& w, double& bal, double& cb, double net_target, double& gross_out,
                                               double& net_out) {
    if (w.nb <= kEps) { MCR_MASKED_MOVE; w.nb = 0.0; w.ncb = 0.0; }                        // :245-247
    if ((bal <= kEps) || (net_target <= 0.0)) {                                             // :218
        MCR_MASKED_MOVE;
        w.nb = bal; w.ncb = cb; w.gross = 0.0; w.net = 0.0;                                 // bal, cb >= 0 already (:219)
    }
    bal = w.nb; cb = w.ncb; gross_out = w.gross; net_out = w.net;
}
template <bool T1, bool T2 = T1, bool MM = true>
__device__ __forceinline__ void withdraw2(double& b1, double& c1, double t1, double r1, double& g1, double& n1,
                                          double& b2, double& c2, double t2, double r2, double& g2, double& n2) {
    WithdrawCand w1 = withdraw_arith<T1>(b1, c1, t1, r1);
    WithdrawCand w2 = withdraw_arith<T2>(b2, c2, t2, r2);
    withdraw_fixup<MM>(w1, b1, c1, t1, g1, n1);
    withdraw_fixup<MM>(w2, b2, c2, t2, g2, n2);
}

// _rebalance_portfolio (:274-359): every lane evaluates the arithmetic with the over-weight asset as the seller.
template <bool STRICT = true, bool TAXED = true, bool MM = true>
__device__ __forceinline__ void rebalance(const LaneParams& P, double& b1, double& c1, double& b2,
                                          double& c2) {
    const double total = b1 + b2;                                  // :288
    const double drift1 = b1 - total * P.alloc1;                   // :293-294
    const bool act = (total > kEps) && (fabs(drift1) > kEps);      // :290-296
    // Wave-uniform early-out: the rebalance that closes the yearly tax step runs right after the monthly
    // one, when every lane is already within eps of its target (nothing to do for the whole wave).
    // (the two compares are balloted separately: each folds into its v_cmp, whereas a ballot of their AND can come
    //  out as a 0/1 VGPR that is compared again)
    if ((__builtin_amdgcn_ballot_w64(total > kEps) & __builtin_amdgcn_ballot_w64(fabs(drift1) > kEps)) == 0ull) return;
    const bool sell1 = drift1 > 0.0;                               // :298
    const double drift2 = b2 - total * P.alloc2;                   // :328
    const double bs = sell1 ? b1 : b2, cs = sell1 ? c1 : c2;       // seller
    const double bb = sell1 ? b2 : b1, cbuy = sell1 ? c2 : c1;     // buyer
    const double drift = sell1 ? drift1 : drift2;
    const double alloc_s = sell1 ? P.alloc1 : P.alloc2;            // the SOLD asset's own weight (:309,:337)
    const double rate_s = sell1 ? P.real_rate1 : P.real_rate2;
    const double inv_bs = recip_nr<STRICT>(bs);                    // shared by the two divisions by bs
    double gross_sale;
    if (TAXED) {
        const double gain_fraction = div_by(fmax(0.0, bs - cs), bs, inv_bs);  // :301 / :329
        const double tax_per_dollar = gain_fraction * rate_s;      // :302-306
        const double denom = fmax(kEps, 1.0 - alloc_s * tax_per_dollar);  // :307-310
        gross_sale = fmin(bs, fdiv<STRICT>(drift, denom));         // :311
    } else {
        gross_sale = fmin(bs, drift);                              // tax_per_dollar = 0, denom = 1, drift / 1 = drift
    }
    const double fraction_sold = div_by(gross_sale, bs, inv_bs);   // :312
    const double basis_part = cs * fraction_sold;
    const double basis_removed = STRICT ? fmin(cs, basis_part) : basis_part;  // :313 (gross_sale <= bs, cs >= 0)
    double net_purchase = gross_sale;                              // rates 0: tax_paid = 0
    if (TAXED) {
        const double taxable_gain = fmax(0.0, gross_sale - basis_removed);  // :314
        const double tax_paid = taxable_gain * rate_s;             // :315-319
        net_purchase = gross_sale - tax_paid;                      // :320
    }
    double nbs = STRICT ? fmax(0.0, bs - gross_sale) : bs - gross_sale;        // :322
    double ncs = STRICT ? fmax(0.0, cs - basis_removed) : cs - basis_removed;  // :323
    double nbb = bb + net_purchase;                                // :324
    double ncb = cbuy + net_purchase;                              // :325
    const bool dust_s = nbs <= kEps, dust_b = nbb <= kEps;         // :355-358
    if (STRICT) {
        nbs = dust_s ? 0.0 : nbs;
        ncs = dust_s ? 0.0 : ncs;
        nbb = dust_b ? 0.0 : nbb;
        ncb = dust_b ? 0.0 : ncb;
        const double r1 = sell1 ? nbs : nbb, rc1 = sell1 ? ncs : ncb;
        const double r2 = sell1 ? nbb : nbs, rc2 = sell1 ? ncb : ncs;
        b1 = act ? r1 : b1;
        c1 = act ? rc1 : c1;
        b2 = act ? r2 : b2;
        c2 = act ? rc2 : c2;
    } else {   // path form: the same selections as exec-masked moves (see withdraw)
        if (dust_s) { MCR_MASKED_MOVE; nbs = 0.0; ncs = 0.0; }
        if (dust_b) { MCR_MASKED_MOVE; nbb = 0.0; ncb = 0.0; }
        if (act) {
            MCR_MASKED_MOVE;
            if (sell1) { MCR_MASKED_MOVE; b1 = nbs; c1 = ncs; b2 = nbb; c2 = ncb; }
            else { MCR_MASKED_MOVE; b1 = nbb; c1 = ncb; b2 = nbs; c2 = ncs; }
        }
    }
}

// Path form of _rebalance_portfolio (:274-359).  Same arithmetic as rebalance<false, TAXED>; what differs is how the
// results land.  The whole update runs under the exec mask of the lanes that act, so nothing has to be selected
// back for the others; the buyer is not selected on the way in: the net purchase is added to BOTH assets in place
// (:324-325 / :352-353 for whichever is the buyer) and the seller's pair is then overwritten under its own mask
// (:322-323 / :350-351).  10 v_cndmask + 4 masked moves, against 14 + 8 for the select-in / select-out form.
template <bool TAXED, bool MM = true>
__device__ __forceinline__ void rebalance_path(const LaneParams& P, double& b1, double& c1, double& b2, double& c2) {
    const double total = b1 + b2;                                  // :288
    const double drift1 = b1 - total * P.alloc1;                   // :293-294
    // (A wave in which no lane acts — e.g. the rebalance that closes the yearly tax step, right after the monthly
    //  one — skips the block through the s_cbranch_execz of this branch: no separate ballot is needed.)
    if ((total > kEps) && (fabs(drift1) > kEps)) {                 // :290-296
        MCR_MASKED_MOVE;
        const bool sell1 = drift1 > 0.0;                           // :298
        const double drift2 = b2 - total * P.alloc2;               // :328
        const double bs = sell1 ? b1 : b2, cs = sell1 ? c1 : c2;   // seller
        const double drift = sell1 ? drift1 : drift2;
        const double alloc_s = sell1 ? P.alloc1 : P.alloc2;        // the SOLD asset's own weight (:309,:337)
        const double rate_s = sell1 ? P.real_rate1 : P.real_rate2;
        const double inv_bs = recip_nr<false>(bs);                 // shared by the two divisions by bs
        double gross_sale;
        if (TAXED) {
            const double gain_fraction = div_by(fmax(0.0, bs - cs), bs, inv_bs);  // :301 / :329
            const double tax_per_dollar = gain_fraction * rate_s;  // :302-306
            const double denom = fmax(kEps, 1.0 - alloc_s * tax_per_dollar);  // :307-310
            gross_sale = fmin(bs, fdiv<false>(drift, denom));      // :311
        } else {
            gross_sale = fmin(bs, drift);                          // tax_per_dollar = 0, denom = 1, drift / 1 = drift
        }
        const double fraction_sold = div_by(gross_sale, bs, inv_bs);   // :312
        const double basis_removed = cs * fraction_sold;           // :313 (gross_sale <= bs, cs >= 0: the min is a no-op)
        double net_purchase = gross_sale;                          // rates 0: tax_paid = 0
        if (TAXED) {
            const double taxable_gain = fmax(0.0, gross_sale - basis_removed);  // :314
            net_purchase = gross_sale - taxable_gain * rate_s;     // :315-320
        }
        const double nbs = bs - gross_sale;                        // :322 (>= 0)
        const double ncs = cs - basis_removed;                     // :323 (>= 0)
        // (locals, not the references: a store of one value to either of two addresses would be merged into a store
        //  through a selected pointer, which pins the whole state in scratch memory)
        double r1b = b1 + net_purchase, r1c = c1 + net_purchase;   // :324-325 for the buyer; the seller's pair is replaced below
        double r2b = b2 + net_purchase, r2c = c2 + net_purchase;
        if (sell1) { MCR_MASKED_MOVE; r1b = nbs; r1c = ncs; }
        else { MCR_MASKED_MOVE; r2b = nbs; r2c = ncs; }
        if (r1b <= kEps) { MCR_MASKED_MOVE; r1b = 0.0; r1c = 0.0; }  // :355-358
        if (r2b <= kEps) { MCR_MASKED_MOVE; r2b = 0.0; r2c = 0.0; }
        b1 = r1b; c1 = r1c; b2 = r2b; c2 = r2c;
    }
}

// ---------------------------------------------------------------------------------------------
// TOLERANCE FORM of the month (what the path kernel runs since round 4; the forms above stay the unit API, the corner
// configurations below and the A/B build -DMCR_K1_EXACT_MONTH).  The forms above mirror the reference's roundings operation
// by operation; the stated tolerance of the path is 1e-9, six orders above one rounding, so inside the month loop the same
// quantities are computed from the CLOSED FORM of the reference's formulas, fused multiply-adds and uncorrected reciprocals
// (v_rcp_f64 + one Newton step, 2^-50):
//
//  withdrawal (:726-790).  With G_i = max(0, b_i - c_i) the liquidation value is cap_i = b_i - r_i G_i (:256-272), the asset's
//    net fraction 1 - gf_i r_i = cap_i / b_i (:221-227), and its net target t_i = target cap_i / cap (:750-765).  Hence
//        gross_i = t_i / (cap_i / b_i) = b_i (target / cap):
//    BOTH assets sell the same fraction phi = target / cap of their balance, basis and gain (:233-244):
//        b_i <- b_i (1 - phi),  c_i <- c_i (1 - phi),  net cash = phi cap = target,  gross = phi (b1 + b2).
//    One reciprocal (of cap) instead of five, no per-asset quotient; the net-cash test (:784-790) IS the capacity test
//    (:743-748).  min(gross, bal) (:228-231) is phi <= 1.
//  rebalance (:274-359).  With G = max(0, bs - cs) of the seller, gross_sale = drift / (1 - a_s r_s G / bs) and
//    fraction_sold = gross_sale / bs (:301-312) are ONE quotient: fraction_sold = drift / (bs - a_s r_s G); the tax is
//    fraction_sold G r_s (:314-319: max(0, gross - basis_removed) IS fraction_sold G); the drift of asset 2 is minus the
//    drift of asset 1 (:328).
//
// Validity: the reference clamps its denominators at 1e-6 (:227, :307-310).  1 - gf r >= 1 - r and 1 - a gf r >= 1 - r, so
// with both effective realized-gains rates <= 1 - 1e-6 the clamps never bind and the closed form is the reference's
// arithmetic up to roundings (DevParams::exact_month = 0, derive_params).  Configurations with a rate above that (a 100 % tax
// on realized gains) run the exact forms in the generic kernel variants.  One sub-case is resolved differently: alive, total
// balance > 1e-6 but total liquidation value <= 1e-6 (needs a balance below 1e-6 / (1 - r) dollars in the very month the path
// fails): the reference then splits the target by the allocation weights instead of the capacity shares (:750-755), the closed
// form keeps the capacity shares — the two differ by at most 1e-6 / (1 - r) dollars in the failing year's residual sample.
// Every other difference is a relative perturbation of ~1e-16 per operation — what a rounding is; measured against the oracle
// (profiles/r04/k1_accuracy_*.txt): worst path-level error relative to the path's money scale, flips of Success flags.

// liquidation values (:256-272, :726-737); returns cap1 + cap2
template <bool T1, bool T2 = T1, bool MM = true>
__device__ __forceinline__ double capacity_tol(double b1, double c1, double r1, double b2, double c2, double r2) {
    double v1 = b1, v2 = b2;
    if (T1) v1 = __builtin_fma(-fmax(0.0, b1 - c1), r1, b1);
    if (T2) v2 = __builtin_fma(-fmax(0.0, b2 - c2), r2, b2);
    if (b1 <= kEps) { MCR_MASKED_MOVE; v1 = 0.0; }
    if (b2 <= kEps) { MCR_MASKED_MOVE; v2 = 0.0; }
    return v1 + v2;
}
// both assets sell the fraction phi (:757-776 as derived above)
template <bool MM = true>
__device__ __forceinline__ void sell_fraction_tol(double phi, double& b1, double& c1, double& b2, double& c2) {
    double nb1 = __builtin_fma(-b1, phi, b1), nc1 = __builtin_fma(-c1, phi, c1);     // :243-244
    double nb2 = __builtin_fma(-b2, phi, b2), nc2 = __builtin_fma(-c2, phi, c2);
    if (nb1 <= kEps) { MCR_MASKED_MOVE; nb1 = 0.0; nc1 = 0.0; }                       // :245-247
    if (nb2 <= kEps) { MCR_MASKED_MOVE; nb2 = 0.0; nc2 = 0.0; }
    if (b1 <= kEps) { MCR_MASKED_MOVE; nb1 = b1; nc1 = c1; }                         // :218-219: an asset without a balance is left alone
    if (b2 <= kEps) { MCR_MASKED_MOVE; nb2 = b2; nc2 = c2; }
    b1 = nb1; c1 = nc1; b2 = nb2; c2 = nc2;
}
// (:274-359)  L = lane_params_tol(P): L.alloc1 / L.alloc2 hold weight x rate of asset 1 / 2.
template <bool TAXED, bool MM = true>
__device__ __forceinline__ void rebalance_tol(const DevParams& P, const LaneParams& L, double& b1, double& c1, double& b2, double& c2) {
    const double total = b1 + b2;                                  // :288
    const double drift1 = __builtin_fma(-total, P.alloc1, b1);     // :293-294
    if ((total > kEps) && (fabs(drift1) > kEps)) {                 // :290-296
        MCR_MASKED_MOVE;
        const bool sell1 = drift1 > 0.0;                           // :298
        const double bs = sell1 ? b1 : b2, cs = sell1 ? c1 : c2;   // seller
        const double drift = fabs(drift1);                         // :328: b2 - total (1 - alloc1) = -(b1 - total alloc1)
        double fraction_sold, net_purchase;
        if (TAXED) {
            const double rate_s = sell1 ? L.real_rate1 : L.real_rate2;
            const double ar_s = sell1 ? L.alloc1 : L.alloc2;       // the SOLD asset's own weight (:309,:337) x its rate
            const double G = fmax(0.0, bs - cs);                   // :301 / :329 (x bs)
            fraction_sold = fmin(1.0, drift * recip_nr<false>(__builtin_fma(-ar_s, G, bs)));   // :302-312
            net_purchase = fraction_sold * __builtin_fma(-G, rate_s, bs);                       // :314-320: gross - (fraction G) rate
        } else {
            fraction_sold = fmin(1.0, drift * recip_nr<false>(bs));
            net_purchase = fraction_sold * bs;
        }
        const double nbs = __builtin_fma(-bs, fraction_sold, bs);  // :322
        const double ncs = __builtin_fma(-cs, fraction_sold, cs);  // :313, :323
        double r1b = b1 + net_purchase, r1c = c1 + net_purchase;   // :324-325 for the buyer; the seller's pair is replaced below
        double r2b = b2 + net_purchase, r2c = c2 + net_purchase;
        if (sell1) { MCR_MASKED_MOVE; r1b = nbs; r1c = ncs; }
        else { MCR_MASKED_MOVE; r2b = nbs; r2c = ncs; }
        if (r1b <= kEps) { MCR_MASKED_MOVE; r1b = 0.0; r1c = 0.0; }  // :355-358
        if (r2b <= kEps) { MCR_MASKED_MOVE; r2b = 0.0; r2c = 0.0; }
        b1 = r1b; c1 = r1c; b2 = r2b; c2 = r2c;
    }
}

// TOL: the closing rebalance in its tolerance form (then L = lane_params_tol(P)).
template <bool STRICT = true, bool TAXED = true, bool ANNUAL = true, bool T1 = TAXED, bool T2 = TAXED, bool MM = true, bool TOL = false>
__device__ __forceinline__ bool annual_gain_taxes(const DevParams& P, const LaneParams& L, double& b1, double& c1,
                                                  double& b2, double& c2, double gain1, double gain2) {
    bool tax_failed = false;
    if (ANNUAL && P.any_annual_tax) {  // wave-uniform: with no annual-tax asset the bill is 0 (:380-390)
        const double due1 = fmax(0.0, gain1) * P.annual_rate1;        // :380-384
        const double due2 = fmax(0.0, gain2) * P.annual_rate2;        // :385-389
        const double total_due = due1 + due2;                         // :390
        const double cap1 = net_liquidation_value<STRICT, T1, MM>(b1, c1, L.real_rate1);  // :392-397
        const double cap2 = net_liquidation_value<STRICT, T2, MM>(b2, c2, L.real_rate2);  // :398-403
        const double cap = cap1 + cap2;                               // :404
        const double pay = fmin(total_due, cap);                      // :405
        tax_failed = pay < total_due - kEps;                          // :406
        if (cap > kEps && pay > 0.0) {                                // :408
            const double share1 = fdiv<STRICT>(cap1, cap);            // :409
            const double share2 = 1.0 - share1;                       // :410
            double g, net1, net2;
            withdraw<STRICT, T1, MM>(b1, c1, pay * share1, L.real_rate1, g, net1);  // :411-419
            withdraw<STRICT, T2, MM>(b2, c2, pay * share2, L.real_rate2, g, net2);  // :420-428
            tax_failed = tax_failed || (net1 + net2 < total_due - kEps);  // :429-430
        }
    }
    if (STRICT) rebalance<true, TAXED, MM>(L, b1, c1, b2, c2);  // :432-442 (always)
    else if (TOL) rebalance_tol<TAXED, MM>(P, L, b1, c1, b2, c2);
    else rebalance_path<TAXED, MM>(L, b1, c1, b2, c2);
    return tax_failed;
}

// Market step shared by both phases (:522-538 and :695-714), given the month's gross factors.
template <bool ANNUAL = true, bool TOL = false>
__device__ __forceinline__ void market_step(double g1, double ginf, double g2, double& b1, double& b2,
                                            double& gacc1, double& gacc2, double& infl) {
    if (ANNUAL) {                     // the accumulators only feed the annual tax bill
        if (TOL) {
            gacc1 = __builtin_fma(b1, g1 - 1.0, gacc1);
            gacc2 = __builtin_fma(b2, g2 - 1.0, gacc2);
        } else {
            gacc1 += b1 * (g1 - 1.0);     // :534
            gacc2 += b2 * (g2 - 1.0);     // :535
        }
    }
    b1 *= g1;                         // :536
    b2 *= g2;                         // :537
    infl *= ginf;                     // :538
}

}  // namespace mcr
