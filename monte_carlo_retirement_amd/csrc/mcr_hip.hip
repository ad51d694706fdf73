// mcr_hip.hip — kernels + C ABI (include/mcr.h) of the MI355X Monte Carlo retirement engine.
//
// K1  path_kernel<MODE, RNG>   one path per lane; the whole horizon in registers
//       MODE 0: success count only          (no per-path HBM traffic; BASELINE config 2)
//       MODE 1: + per-path summary fields   (SoA, 49 B/path)
//       MODE 2: + yearly trajectories       (time-major [T][N]: 512 contiguous bytes per wave store)
//       RNG 0: Philox4x32-10 + Box-Muller (mcr_device.h);  RNG 1: NumPy's SeedSequence -> PCG64 ->
//              ziggurat stream (mcr_numpy_rng.h), for literal seed parity with the reference
//     reductions: wave ballot+popcount -> LDS -> one global atomic per workgroup.
// Helpers: helper_kernel (device unit functions + the specialised math of mcr_math.h),
//          shocks_kernel / np_shocks_kernel (_draw_shock_path).
// Aggregation kernels (quantiles / histogram) live in mcr_aggregate.hip.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see csrc/build.py).  gfx950 only.

#include <hip/hip_runtime.h>

#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

#include "mcr_device.h"
#include "mcr_numpy_rng.h"
#include "mcr_host.h"

namespace mcr {

// ---------------------------------------------------------------------------------------------
// K1: the per-path state machine (_run_single_simulation_path, simulation.py:476-950)
// ---------------------------------------------------------------------------------------------
constexpr int kMaxSegments = 8;
struct KernelIO {
    uint64_t seed;           // Philox key
    uint64_t path_begin;
    uint64_t n_paths;
    const double* injected;  // [n_paths][shock_rows][3] or nullptr
    mcr_outputs out;
    uint32_t stream_id;
    // MCR_RNG_NUMPY (mcr_numpy_rng.h)
    uint32_t n_entropy;
    uint32_t entropy[MCR_MAX_ENTROPY_WORDS];
    uint64_t child_offset;
    const uint32_t* path_seeds;  // [n_paths] explicit seeds or nullptr
    // search probes that share their accumulation phase (PHASE 1 / 2 of path_kernel; mcr_probe_months_rng)
    double* snap;                    // [n_snap][kSnapFields][snap_stride] state at the end of month snap_months[c]
    int64_t snap_stride;
    int32_t n_snap;
    int32_t snap_months[MCR_MAX_PROBE_CANDIDATES];   // ascending
    int32_t cand_out[MCR_MAX_PROBE_CANDIDATES];      // PHASE 2: counter block of candidate c = counters + cand_out[c] * MCR_N_COUNTERS
    // PHASE 3 (time-sliced path blocks, see path_kernel): the first seg_n_split path blocks of the launch are cut into seg_q
    // segments at retirement-year boundaries; `snap` holds their hand-over state, seg_flags says which segments are done
    int32_t seg_n_split, seg_n_full, seg_q;
    int32_t seg_blocks_per_cand;                     // PHASE 4: path blocks per search candidate (the launch lists candidate-major)
    double* seg_state;                               // [seg_n_split][fields][kBlock] hand-over state
    int32_t seg_year[kMaxSegments + 1];              // segment k covers retirement years [seg_year[k], seg_year[k + 1]); segment 0 also the accumulation
    unsigned int* seg_flags;                         // [seg_n_split][seg_q], zeroed before the launch
    int32_t seg_max_polls;                           // x ~1 us: how long a successor looks for its predecessor's flag before it recomputes the block itself
};
constexpr int kSplitVotePairs = 16;   // SPLIT: pairs of months between two stop votes of a workgroup (a power of two)
constexpr int kSnapFields = 10;   // b1 b2 c1 c2 gacc1 gacc2 infl contrib | pre_fail | Philox carry words

// NaN OUTPUT values travel as integer bit patterns (robust against any no-NaN math assumption: the
// state machine itself never produces a NaN for valid scenarios).
// (YearsToRuin of successful paths, withdrawal rates after failure.)
constexpr unsigned long long kNanBits = 0x7ff8000000000000ull;
__device__ __forceinline__ unsigned long long f64_bits(double x) { return (unsigned long long)__double_as_longlong(x); }
__device__ __forceinline__ void store_bits(double* p, unsigned long long bits) { *reinterpret_cast<unsigned long long*>(p) = bits; }

// Resident waves per SIMD are worth more than a few spilled registers: the trimmed count-only kernel loses 4.4 % on
// large batches at 4 instead of 6 workgroups per CU (measured with a larger LDS footprint, LABNOTES.md rounds 1-3 section 9), and the Philox
// variants with per-path outputs gain 4.5 % (summary) / 3 % (trajectories) at 5 waves (<= 96 VGPRs, 4-10 of them
// spilled) over 4 (up to 128); at 6 the summary variant spills too much and gives that back.  The NumPy-stream and
// injection variants keep 4 (they need 107-128).  The count-only Philox
// variants are held to 6 waves per SIMD (<= 80 VGPRs): the BASELINE workload of 1e6 paths is 15.26 waves per SIMD, and
// with 5 resident waves that is 5 + 5 + 5 + a lone fourth round (+9 %, LABNOTES.md rounds 1-3 section 5); the annual-tax variant sat at 81.
// INJ = true: shocks come from io.injected (the parity hook) instead of the RNG; only instantiated with MODE 2
// (every output is null-checked), so the hot variants carry neither the injection branches nor their registers.
// PHASE 0: the whole path.  PHASE 1 / 2 split it at retirement for the search (simulation.py:1180-1194 re-simulates the
// accumulation months for every probed candidate although, under common random numbers, they do not depend on it,
// :513-579): PHASE 1 runs the accumulation once to the largest candidate and stores the state at the end of every
// candidate month; PHASE 2 (grid.y = candidate) resumes each candidate's decumulation from its snapshot.
// SPLIT = true (count-only Philox variants; launches that leave SIMDs idle — a lone 50 000-path search probe is 782
// wavefronts on 1 024 SIMDs and runs at ~12 cycles per instruction, latency-bound): the workgroup has 2 x kBlock threads
// for its kBlock paths.  Threads kBlock .. 2 kBlock - 1 are PRODUCERS: they run growth_rows2 (Philox, Box-Muller, exp) one
// pair of months ahead into a double-buffered stage; threads 0 .. kBlock - 1 are CONSUMERS: the state machine.  The two
// halves of a month's dependency chain then run on different SIMDs of the CU.  One workgroup barrier per pair of months
// hands a buffer over; every wave executes the same number of them (no early exit when all lanes have failed) or has
// terminated.  The arithmetic of every path is unchanged: counts are bit-identical to SPLIT = false.
// PHASE 3 (Philox launches of a few rounds of workgroups, any output mode; launch_paths): TIME-SLICED path blocks.  A launch of B
// equal workgroups on W resident slots ends when the busiest slot has run ceil(B / W) of them: 10^6 paths are 15.26 workgroups
// per CU-slot, so the chip idles through most of a sixteenth round (measured: 5.15 ms where 4.77 would do, LABNOTES R4.6).
// Small work items at the END of the dispatch order fix that (longest-processing-time-first): the first S path blocks of the
// launch are cut into Q segments at retirement-year boundaries, and the grid is ordered [segment 0 of the S blocks] [the
// other blocks, whole] [segment 1 x S] ... [segment Q - 1 x S].  A segment ends by storing its lanes' state (balances, bases,
// gain accumulators, price level, flags, Philox carry, lock columns) and raising a flag; its successor — dispatched at least
// S workgroups later, i.e. after it has long finished — loads it.  A successor that does not see the flag within a bounded
// number of polls recomputes the block from month 0 itself: no workgroup ever waits on another one to make progress.
// The arithmetic of every path is unchanged: counts and bins are bit-identical to PHASE 0.
// XS = true ("extended streams"): the variants that can read income-stream records beyond the by-value block from the device
// table and keep lock slots beyond the LDS budget in the global overflow block (DevParams::extra_streams / lock_overflow:
// other_income_streams has no length limit in the reference, config.py:99).  A compile-time variant because the headline
// kernels have no SGPR to spare for the two extra tests a month; instantiated for the generic tax form only (TAXED = 3,
// ANNUAL = true: exact zeros for a zero rate, like the NumPy-stream variants).
// EXACT = true: the month in its exact-rounding forms (mcr_device.h) instead of the tolerance form — for configurations whose
// realized-gains rate lets the reference's denominator clamps bind (DevParams::exact_month), instantiated for the generic XS
// variants only; -DMCR_K1_EXACT_MONTH builds a library that runs every variant that way (A/B).
#ifdef MCR_K1_EXACT_MONTH
constexpr bool kExactMonthDefault = true;
#else
constexpr bool kExactMonthDefault = false;
#endif
template <int MODE, int RNG, int TAXED, bool ANNUAL, bool INJ = false, int PHASE = 0, bool SPLIT = false, bool XS = false, bool EXACT = kExactMonthDefault>
__global__ __launch_bounds__(SPLIT ? 2 * kBlock : kBlock, SPLIT ? 4 : (MODE == 0 && RNG == 0 && (PHASE == 0 || PHASE == 3 || PHASE == 4)) ? 6 : ((MODE == 1 || MODE == 2) && RNG == 0 && !INJ) ? 5 : 4) void path_kernel(const DevParams P_arg, const KernelIO io,
                                                         const DevParams* __restrict__ cand_params) {
    static_assert(!SPLIT || (MODE == 0 && RNG == 0 && !INJ), "the producer / consumer split exists for the count-only Philox variants");
    static_assert(!XS || (PHASE == 0 && !SPLIT && TAXED == 3 && ANNUAL), "extended stream lists run the generic whole-path form");
    // PHASE 4 = PHASE 2 (a candidate's decumulation resumed from its accumulation snapshot) time-sliced like PHASE 3: the
    // 17-month verification window of the search is 17 x 196 workgroups = 2.17 rounds of the resident slots.
    constexpr bool kCand = PHASE == 2 || PHASE == 4;      // resumes a search candidate from its snapshot; per-candidate parameter block
    constexpr bool kSliced = PHASE == 3 || PHASE == 4;    // time-sliced path blocks
    static_assert(!kSliced || (RNG == 0 && !INJ && !SPLIT && !XS), "time-sliced blocks exist for the plain Philox variants");
    static_assert(PHASE != 4 || MODE == 0, "the search probes count only");
    // TAXED: which assets carry an effective realized-gains rate (bit 0: inv1, bit 1: inv2; DevParams::tax_mask)
    static_assert(TAXED >= 0 && TAXED <= 3, "TAXED is a two-bit mask");
    constexpr bool T1 = (TAXED & 1) != 0, T2 = (TAXED & 2) != 0, TANY = TAXED != 0;
    static_assert(!EXACT || XS || kExactMonthDefault, "the exact month is instantiated for the generic variants only");
    constexpr bool TOL = !EXACT;         // the month in its tolerance form (mcr_device.h: "TOLERANCE FORM of the month")
    constexpr bool MM = !SPLIT;          // exec-masked moves (issue-bound launches) vs the compiler's selects (latency-bound SPLIT launches): MCR_MASKED_MOVE, mcr_device.h
    constexpr int kThreads = SPLIT ? 2 * kBlock : kBlock;
    const int tid = SPLIT ? (int)(threadIdx.x & (kBlock - 1)) : (int)threadIdx.x;    // the path's lane column in every per-path LDS array
    const bool producer = SPLIT && threadIdx.x >= (unsigned)kBlock;                  // wave-uniform (kBlock = 4 wavefronts)
    // PHASE 2: the parameter block of candidate blockIdx.y, in device memory (a separate const __restrict__ kernel
    // argument so that its loads are provably invariant and uniform: scalar loads, like the by-value block)
    // time-sliced launches (1-D grid): which path block (of which candidate) and which segment of it this workgroup runs (seg < 0: a whole block)
    int seg = -1, seg_block = 0;
    unsigned int path_block = blockIdx.x, cand = PHASE == 2 ? blockIdx.y : 0u;
    if (kSliced) {
        const int S = io.seg_n_split, F = io.seg_n_full, bid = (int)blockIdx.x;
        unsigned int lb = (unsigned)bid;                  // the block's position in the launch's list of (candidate, path block) pairs
        if (bid < S) { seg = 0; seg_block = bid; }
        else if (bid >= S + F) { const int k = bid - S - F; seg = 1 + k / S; seg_block = k % S; lb = (unsigned)seg_block; }
        if (PHASE == 4) { cand = lb / (unsigned)io.seg_blocks_per_cand; path_block = lb % (unsigned)io.seg_blocks_per_cand; }
        else path_block = lb;
    }
    const DevParams& P = kCand ? cand_params[cand] : P_arg;
    // LDS.  STATIC: the math tables (mcr_math.h) and, for the Philox stream, the [6][kBlock] stage of two months' gross
    // factors — static because the compiler then knows their addresses (offset 0 ...) and a table lookup is index << 3 +
    // ds_read with an immediate offset; against the dynamic region every address is `base + ...` with a base it only learns
    // to be 0 after instruction selection: v_lshl_add_u32 x, 3, 0 (a 3-operand op, 4.6 cycles instead of 2.7) or a literal
    // v_add_u32 0.  DYNAMIC: the NumPy ziggurat tables, [n_lock_slots][kBlock] doubles (frozen nominal stream amounts),
    // the block counters.
    constexpr bool kStaged = RNG == (int)MCR_RNG_PHILOX && !INJ;
    __shared__ __align__(16) double tab_s[kTabDoubles];
    __shared__ __align__(16) double stage_s[kStaged ? (SPLIT ? 2 : 1) * kStageDoubles : 1];
    // Per-path values that are written once or twice in a lifetime and read at the very end (first-year withdrawals,
    // YearsToRuin) live in the lane's own LDS column in the variants with per-path outputs: held to 5 waves per SIMD those
    // variants had no registers for them (round 2: 12 / 36 bytes of scratch per lane, 4-12 VGPRs spilled).
    // A/B on one box (tools/k1_modes_ab.py): summary output, 2e7 S60 paths 126.44 -> 126.23 ms; trajectories, 1e7 jorge paths
    // 50.28 -> 50.07 ms, 4e6 config.json paths 29.59 -> 29.40 ms.  A fourth column (the start-of-retirement balance) frees the
    // trajectory variant of its last 12 bytes of scratch but its 2 KB cost a resident workgroup as soon as the scenario has a
    // non-indexed income stream (one more LDS column): 50.07 -> 52.2 ms on jorge.json.  Three it is.
    constexpr bool kSumLds = MODE >= 1 && kStaged;
    __shared__ __align__(16) double sum_s[kSumLds ? 3 * kBlock : 1];
    extern __shared__ __align__(16) unsigned char smem_raw[];
#ifdef MCR_K1_TIMELINE   // diagnostic build only (tools/k1_timeline.py): per-wave start / end stamps and placement
    const unsigned long long tl_t0 = wall_clock64();
#endif
    double* tab = tab_s;
    load_math_tables(tab, threadIdx.x, kThreads);
    ZigTables zig{nullptr, nullptr, nullptr, nullptr};
    if (RNG == (int)MCR_RNG_NUMPY) { zig = load_zig_tables(smem_raw, threadIdx.x, kThreads); zig.math_tab = tab; }
    // Philox stream: the gross factors of two months at a time, staged per lane (growth_rows2)
    double* stage = stage_s + (kStaged ? tid : 0);
    double* sum_col = sum_s + (kSumLds ? tid : 0);     // [0] first-year gross, [kBlock] first-year real gross, [2 kBlock] YearsToRuin bits
    double* lock_lds = reinterpret_cast<double*>(smem_raw + (RNG == (int)MCR_RNG_NUMPY ? kZigLdsBytes : 0));
    unsigned int* blk = reinterpret_cast<unsigned int*>(lock_lds + (size_t)P.n_lock_slots * kBlock);
    // blk[0] = success count; blk[1 .. 1+ry+2) = ruin bins; then [ry+1] done-years histogram; then the
    // [hist_n_bins] final-balance histogram of the workgroup (mcr_outputs.hist_bins), when requested
    const int ry = P.retirement_years;
    const int n_blk = 1 + (ry + 2) + (ry + 1);
    const int n_hist = ((PHASE == 0 || PHASE == 3) && io.out.hist_bins != nullptr) ? io.out.hist_n_bins : 0;
    for (int k = threadIdx.x; k < n_blk + n_hist; k += kThreads) blk[k] = 0u;
    __syncthreads();

    const uint64_t local = (uint64_t)path_block * kBlock + (unsigned)tid;
    const bool valid = local < io.n_paths;
    const uint64_t li = valid ? local : (io.n_paths - 1);  // tail lanes shadow the last path, write nothing
    const uint64_t path = io.path_begin + li;
    const int64_t stride = io.out.path_stride;
    const double* inj = INJ ? io.injected + (size_t)li * 3u * (size_t)P.shock_rows : nullptr;

    constexpr bool kSummary = MODE >= 1;
    constexpr bool kTraj = MODE >= 2;
    double* traj = kTraj ? io.out.trajectory : nullptr;
    double* rtraj = kTraj ? io.out.real_trajectory : nullptr;
    double* wrt = kTraj ? io.out.withdrawal_rate_trajectory : nullptr;

    auto put_sample = [&](int t, double nominal, double px) {  // :574-576, :928-931
        if (kTraj && valid) {
            if (traj) traj[(int64_t)t * stride + (int64_t)li] = nominal;
            if (rtraj) rtraj[(int64_t)t * stride + (int64_t)li] = px > kEps ? nominal / px : 0.0;
        }
    };
    Pcg64 gen;  // NumPy stream: one generator per path, rows are consumed strictly in order
    if (RNG == (int)MCR_RNG_NUMPY && !INJ) {
        const uint32_t s32 = io.path_seeds ? io.path_seeds[li]
                                           : np_path_seed(io.entropy, (int)io.n_entropy, io.stream_id, io.child_offset + path);
        pcg64_seed_u32(gen, s32);
    }
    // Top of every month (wave-uniform, outside any divergent region): rows are visited in order 0, 1, 2, ... across
    // both phases, so each pair of rows is generated exactly when its first row comes up.
    PairCarry carry{0u, 0u};
    const MathRegs GR = kStaged ? MathRegs::pinned_path() : MathRegs::literals();
    // Wave priority falls as the path advances (s_setprio takes an immediate: four levels).  The SIMD arbitrates VALU
    // issue by priority, then age; left alone, the oldest wave of a SIMD runs far ahead and the youngest is left to
    // finish ALONE at the end of the launch, at a fraction of the SIMD's issue rate (measured with per-wave
    // s_memrealtime stamps: waves of one 10^6-path launch took 1.2 to 3.6 ms and the drain was 3 of its 8.4 ms).
    // With laggards served first the waves of a SIMD finish together, the bands narrowing towards the end of the
    // path: 8.40 -> 7.98 ms at exactly 10^6 paths (profiles/r02/k1_timeline_*.txt).
    const int prio_t1 = P.total_months / 2, prio_t2 = (P.total_months * 3) / 4, prio_t3 = (P.total_months * 9) / 10;
    __builtin_amdgcn_s_setprio(3);
    bool wg_dead = false;      // SPLIT: no consumer lane of the workgroup is alive any more (wave-uniform, agreed at a barrier)
    bool lane_alive = true;    // SPLIT: this consumer lane still has months to simulate
    auto begin_month = [&](int row) {
        if (row == prio_t1) __builtin_amdgcn_s_setprio(2);
        else if (row == prio_t2) __builtin_amdgcn_s_setprio(1);
        else if (row == prio_t3) __builtin_amdgcn_s_setprio(0);
        if (kStaged && (row & 1) == 0) {
            if (SPLIT) {
                // the producers have staged the pair (row, row + 1) in buffer (row >> 1) & 1.  Every kSplitVotePairs-th pair the barrier
                // also votes (a voting barrier costs about two plain ones: every 6th pair 0.93 ms per lone probe, every 16th 0.88): once no consumer lane of the workgroup is alive, producers and consumers stop together (the unsplit
                // kernel lets a wave leave as soon as all of ITS lanes have failed)
                if (wg_dead) return;
                if (((row >> 1) & (kSplitVotePairs - 1)) == 0) { if (__syncthreads_or(__builtin_amdgcn_ballot_w64(lane_alive) != 0ull ? 1 : 0) == 0) wg_dead = true; }
                else __syncthreads();
            } else {
                if ((row & 2) == 0) growth_rows2<0>(P, GR, io.seed, io.stream_id, path, (uint32_t)row >> 2, tab, stage, carry);
                else growth_rows2<1>(P, GR, io.seed, io.stream_id, path, (uint32_t)row >> 2, tab, stage, carry);
            }
        }
    };
    // gross factors of month `row` (:522-532)
    auto growth = [&](int row, double& g1, double& ginf, double& g2) {
        if (kStaged) {
            const double* c = stage + (size_t)(3 * (row & 1)) * kBlock + (SPLIT ? (size_t)((row >> 1) & 1) * kStageDoubles : 0);
            g1 = c[0]; ginf = c[kBlock]; g2 = c[2 * kBlock];
            return;
        }
        double ze, zi, zp;
        if (INJ) {
            const int r = row < P.shock_rows - 1 ? row : P.shock_rows - 1;  // :692
            ze = inj[3 * r + 0]; zi = inj[3 * r + 1]; zp = inj[3 * r + 2];
        } else {
            const double z0 = np_standard_normal(gen, zig);  // standard_normal((n, 3)) fills row-major (:458)
            const double z1 = np_standard_normal(gen, zig);
            const double z2 = np_standard_normal(gen, zig);
            ze = z0;
            zi = P.rho * z0 + P.rho_c * z1;                  // :461-464
            zp = z2;
        }
        g1 = monthly_gross(P.a1, P.b1, ze, tab);
        ginf = monthly_gross(P.ainf, P.binf, zi, tab);
        g2 = ginf * monthly_gross(P.aprem, P.bprem, zp, tab);  // :532
    };

    const LaneParams L = TOL ? lane_params_tol(P) : lane_params(P);

    // ---- initial state (:490-510) ----
    double b1 = P.initial_balance * P.alloc1;  // :499
    double b2 = P.initial_balance - b1;        // :500
    double c1 = b1, c2 = b2;                   // :501-502
    double contrib = P.monthly_contribution;   // :504
    double gacc1 = 0.0, gacc2 = 0.0;           // :505-506
    double infl = 1.0;                         // :508
    bool pre_fail = false;                     // :510
    int t_idx = 0;
    put_sample(t_idx++, P.initial_balance, 1.0);  // :490-492
    const int wm = P.working_months;
    // snapshot c, field f of local path li: snap[(c * kSnapFields + f) * snap_stride + li]
    auto snap_at = [&](int c, int f) { return io.snap + ((size_t)c * kSnapFields + (size_t)f) * (size_t)io.snap_stride + (size_t)li; };
    int snap_i = 0;
    auto save_snapshot = [&]() {
        if (valid) {
            *snap_at(snap_i, 0) = b1; *snap_at(snap_i, 1) = b2; *snap_at(snap_i, 2) = c1; *snap_at(snap_i, 3) = c2;
            *snap_at(snap_i, 4) = gacc1; *snap_at(snap_i, 5) = gacc2; *snap_at(snap_i, 6) = infl; *snap_at(snap_i, 7) = contrib;
            *snap_at(snap_i, 8) = pre_fail ? 1.0 : 0.0;
            if (!SPLIT) store_bits(snap_at(snap_i, 9), ((unsigned long long)carry.w3 << 32) | (unsigned long long)carry.w2);   // (SPLIT: the producer's)
        }
        ++snap_i;
    };
    if (SPLIT && producer) {
        // Rows [first_row, last_row), a pair per iteration, one barrier per pair: exactly the barriers the consumers execute
        // in begin_month at every even row they visit (they visit every row of this range, in order, and never leave early).
        const int first_row = PHASE == 2 ? (wm & ~1) : 0;                 // PHASE 2 resumes with the pair that holds row wm
        const int last_row = PHASE == 1 ? wm : P.total_months;
        if (PHASE == 2) {
            const unsigned long long cw = f64_bits(*snap_at((int)blockIdx.y, 9));
            carry.w2 = (uint32_t)cw; carry.w3 = (uint32_t)(cw >> 32);
        }
        // PHASE 1: the Philox words carried past the end of candidate month m are those in hand once the pair that holds row
        // m - 1 has been generated (m = 0: none yet) — what the unsplit kernel stores from its single `carry`
        auto put_carry = [&]() {
            if (valid) store_bits(snap_at(snap_i, 9), ((unsigned long long)carry.w3 << 32) | (unsigned long long)carry.w2);
            ++snap_i;
        };
        if (PHASE == 1) while (snap_i < io.n_snap && io.snap_months[snap_i] == 0) put_carry();
        for (int row = first_row; row < last_row; row += 2) {
            double* st = stage + (size_t)((row >> 1) & 1) * kStageDoubles;
            if ((row & 2) == 0) growth_rows2<0>(P, GR, io.seed, io.stream_id, path, (uint32_t)row >> 2, tab, st, carry);
            else growth_rows2<1>(P, GR, io.seed, io.stream_id, path, (uint32_t)row >> 2, tab, st, carry);
            if (PHASE == 1) while (snap_i < io.n_snap && ((io.snap_months[snap_i] - 1) >> 1) == (row >> 1)) put_carry();
            if (((row >> 1) & (kSplitVotePairs - 1)) == 0) { if (__syncthreads_or(0) == 0) return; }   // (the consumers' vote, begin_month)
            else __syncthreads();
        }
        return;
    }
    if (PHASE == 1) while (snap_i < io.n_snap && io.snap_months[snap_i] == 0) save_snapshot();
    // PHASE 3: a later segment of a time-sliced block takes its lanes' state over from its predecessor
    // b1 b2 c1 c2 gacc1 gacc2 infl | flags | Philox carry | (per-path outputs: balance and price level at retirement, the
    // three write-once columns) ; then the lock columns
    constexpr int kSegFixedFields = MODE >= 1 ? 14 : 9;
    double seg_start_balance = 0.0, seg_infl_ret = 0.0;
    auto seg_at = [&](int f) { return io.seg_state + ((size_t)seg_block * (size_t)(kSegFixedFields + P.n_lock_slots) + (size_t)f) * kBlock + (size_t)tid; };
    __shared__ int seg_ok_s;
    bool seg_resumed = false;
    int y_begin = 0, y_end = ry;
    unsigned long long seg_state_flags = 0ull;
    if (kSliced && seg >= 0) {
        y_end = io.seg_year[seg + 1];
        if (seg > 0) {
            if (threadIdx.x == 0) {
                const unsigned int* f = io.seg_flags + (size_t)seg_block * (size_t)io.seg_q + (size_t)(seg - 1);
                int ok = 0;
                for (int spin = 0; spin < io.seg_max_polls; ++spin) {
                    if (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 1; break; }
                    __builtin_amdgcn_s_sleep(32);
                }
                seg_ok_s = ok;
            }
            __syncthreads();
            seg_resumed = seg_ok_s != 0;     // (otherwise: this workgroup runs the block from month 0 itself, up to its own end)
            if (seg_resumed) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                b1 = *seg_at(0); b2 = *seg_at(1); c1 = *seg_at(2); c2 = *seg_at(3);
                if (ANNUAL) { gacc1 = *seg_at(4); gacc2 = *seg_at(5); }    // (otherwise identically 0 and never read)
                infl = *seg_at(6);
                seg_state_flags = f64_bits(*seg_at(7));
                const unsigned long long cw = f64_bits(*seg_at(8));
                carry.w2 = (uint32_t)cw; carry.w3 = (uint32_t)(cw >> 32);
                for (int k = 0; k < P.n_lock_slots; ++k) lock_lds[(size_t)k * kBlock + tid] = *seg_at(kSegFixedFields + k);
                if (MODE >= 1) {
                    seg_start_balance = *seg_at(9); seg_infl_ret = *seg_at(10);
                    sum_col[0] = *seg_at(11); sum_col[kBlock] = *seg_at(12); sum_col[2 * kBlock] = *seg_at(13);
                }
                y_begin = io.seg_year[seg];
                const int r0 = wm + kMPY * y_begin;
                if (r0 >= prio_t3) __builtin_amdgcn_s_setprio(0);
                else if (r0 >= prio_t2) __builtin_amdgcn_s_setprio(1);
                else if (r0 >= prio_t1) __builtin_amdgcn_s_setprio(2);
                if (r0 & 1) begin_month(r0 - 1);   // the pair of rows (r0 - 1, r0) was staged by the predecessor: stage it again
            }
        }
    }

    if (kCand && !seg_resumed) {   // (a segment that took its predecessor's state over needs neither the snapshot nor the re-staged pair)
        const int c = (int)cand;
        b1 = *snap_at(c, 0); b2 = *snap_at(c, 1); c1 = *snap_at(c, 2); c2 = *snap_at(c, 3);
        gacc1 = *snap_at(c, 4); gacc2 = *snap_at(c, 5); infl = *snap_at(c, 6); contrib = *snap_at(c, 7);
        pre_fail = *snap_at(c, 8) != 0.0;
        const unsigned long long cw = f64_bits(*snap_at(c, 9));
        carry.w2 = (uint32_t)cw; carry.w3 = (uint32_t)(cw >> 32);
        if (wm & 1) begin_month(wm - 1);   // the pair of rows (wm - 1, wm) was staged during the accumulation: stage it again
    }
    // ---- accumulation (:513-579): no lane leaves this loop early ----
    for (int m = 1; m <= ((kCand || (kSliced && seg_resumed)) ? 0 : wm); ++m) {
        if (P.contrib_grows && (m - 1) % kMPY == 0 && m > 1) {  // :514-517 (wave-uniform: a scalar branch, not a select)
            asm volatile("");
            contrib *= P.contrib_growth_factor;
        }
        begin_month(m - 1);
        double g1, ginf, g2;
        growth(m - 1, g1, ginf, g2);                                   // :519-532
        market_step<ANNUAL, TOL>(g1, ginf, g2, b1, b2, gacc1, gacc2, infl); // :534-538
        const double k1 = contrib * P.alloc1;                          // :540-542
        const double k2 = contrib - k1;                                // :543
        b1 += k1; c1 += k1; b2 += k2; c2 += k2;                        // :544-547
        if (TOL) rebalance_tol<TANY, MM>(P, L, b1, c1, b2, c2);        // :549-553
        else rebalance_path<TANY, MM>(L, b1, c1, b2, c2);
        if (m % kMPY == 0) {                                           // :557
            pre_fail |= annual_gain_taxes<false, TANY, ANNUAL, T1, T2, MM, TOL>(P, L, b1, c1, b2, c2, gacc1, gacc2);  // :558-573
            put_sample(t_idx++, b1 + b2, infl);                        // :574-576
            gacc1 = 0.0; gacc2 = 0.0;                                  // :578-579
        }
        if (PHASE == 1 && snap_i < io.n_snap && m == io.snap_months[snap_i]) save_snapshot();
    }
    if (PHASE == 1) return;   // (every thread of the workgroup: nothing below is needed)
    double start_balance = b1 + b2;        // :581
    double infl_ret = infl;                // :582
    if (kSliced && seg_resumed) { start_balance = seg_start_balance; infl_ret = seg_infl_ret; t_idx = P.trajectory_len - ry + y_begin; }
    else if (wm > 0 && wm % kMPY != 0) put_sample(t_idx++, start_balance, infl_ret);  // :590-594

    // ---- decumulation (:632-868) ----
    double fy_gross = 0.0, fy_real = 0.0;            // :623-624
    bool alive = !pre_fail;                          // :627, :633
    bool succeeded = !pre_fail;
    unsigned long long ytr_bits = pre_fail ? f64_bits(0.0) : kNanBits;  // YearsToRuin (:497, :628-629)
    if (kSumLds && !(kSliced && seg_resumed)) { sum_col[0] = 0.0; sum_col[kBlock] = 0.0; store_bits(&sum_col[2 * kBlock], ytr_bits); }
    // A launch that cannot fill the chip (SPLIT) is bound by each wave's dependency chain, and re-reading a stream's record
    // from the kernel arguments every month is three dependent scalar loads on it (the compiler loads start, then end, then
    // the rest): the first two records stay in SGPRs there (82 + 16 of them; the unsplit kernel has none to spare).
    DevStream S0 = {}, S1 = {};
    if (SPLIT) {
        if (P.n_streams > 0) S0 = P.streams[0];
        if (P.n_streams > 1) S1 = P.streams[1];
        // (opaque to the compiler from here on: kernel-argument loads are otherwise rematerialised in the loop)
        if (TOL) {   // (the tolerance form of the month reads the netted amount only)
            asm volatile("" : "+s"(S0.amount_keep), "+s"(S0.start_month), "+s"(S0.end_month), "+s"(S0.indexed), "+s"(S0.lock_slot));
            asm volatile("" : "+s"(S1.amount_keep), "+s"(S1.start_month), "+s"(S1.end_month), "+s"(S1.indexed), "+s"(S1.lock_slot));
        } else {
            asm volatile("" : "+s"(S0.amount), "+s"(S0.keep), "+s"(S0.start_month), "+s"(S0.end_month), "+s"(S0.indexed), "+s"(S0.lock_slot));
            asm volatile("" : "+s"(S1.amount), "+s"(S1.keep), "+s"(S1.start_month), "+s"(S1.end_month), "+s"(S1.indexed), "+s"(S1.lock_slot));
        }
    }
    int ruin_bin = pre_fail ? 0 : -1;
    int done_years = 0;  // completed (observed) retirement years = non-NaN WR entries
    int year = 0;
    if (kSliced && seg_resumed) {   // flags of the hand-over state: alive | succeeded << 1 | (ruin_bin + 1) << 8 | done_years << 24
        alive = (seg_state_flags & 1ull) != 0ull;
        succeeded = (seg_state_flags & 2ull) != 0ull;
        ruin_bin = (int)((seg_state_flags >> 8) & 0xFFFFull) - 1;
        done_years = (int)(seg_state_flags >> 24);
        year = y_begin;
    }
    for (; year < (kSliced ? y_end : ry); ++year) {
        if (!SPLIT && __builtin_amdgcn_ballot_w64(alive) == 0ull) break;  // every lane of this wave has failed: nothing left to simulate
        if (SPLIT) { lane_alive = alive; if (wg_dead) break; }           // (SPLIT: the wave keeps pace with its producers' barriers until the workgroup votes to stop)
        double tg1 = 0.0, tg2 = 0.0, treal = 0.0;  // :635-637
        bool yfail = false;                        // :638
        int fail_rmi = 0;
        for (int mi = 0; mi < kMPY; ++mi) {
            const int rmi = year * kMPY + mi;  // :641-643
            begin_month(wm + rmi);
            if (alive && !yfail) {
                double g1, ginf, g2;
                if (kStaged) growth(wm + rmi, g1, ginf, g2);           // staged factors: the LDS reads are issued early
                const double price = infl;                             // :644
                const double expenses = P.monthly_expenses * price;    // :645-647
                // exact form: income accumulates (:649-677) and need = max(0, expenses - income); tolerance form: `income` runs
                // DOWN from the expenses, one FMA per indexed stream ((amount keep) price), one subtraction per frozen stream
                // (its slot holds the netted amount): need = max(0, what is left)
                double income = TOL ? expenses : 0.0;                  // :649
                auto stream_income = [&](const DevStream& S) {
                    if (rmi < S.start_month || rmi >= S.end_month) return;    // :653-658
                    double nominal = 0.0;
                    if (S.indexed) {
                        if (TOL) { income = __builtin_fma(-S.amount_keep, price, income); return; }
                        nominal = S.amount * price;                    // :661-665
                    } else if (!XS || S.lock_slot < P.n_lock_slots) {      // (wave-uniform; without XS every slot is an LDS column)
                        double* slot = lock_lds + (size_t)S.lock_slot * kBlock + tid;
                        if (rmi == S.start_month) *slot = (TOL ? S.amount_keep : S.amount) * price;  // :667-671 (first active month)
                        nominal = *slot;                               // :672-674
                    } else {                                           // a slot beyond the LDS budget: the lane's column of the overflow block
                        double* slot = P.lock_overflow + (size_t)(S.lock_slot - P.n_lock_slots) * (size_t)P.lock_stride + (size_t)local;
                        if (rmi == S.start_month) *slot = (TOL ? S.amount_keep : S.amount) * price;
                        nominal = *slot;
                    }
                    if (TOL) income -= nominal;
                    else income += nominal * S.keep;                   // :675-677
                };
                int s = 0;
                if (SPLIT) {            // the first two streams sit in SGPRs for the whole launch (see S0, S1 above)
                    if (P.n_streams > 0) stream_income(S0);
                    if (P.n_streams > 1) stream_income(S1);
                    s = 2;
                }
                for (; s < P.n_streams; ++s) stream_income(P.streams[s]);    // :650 (wave-uniform; the record is re-read from the kernel arguments)
                if (XS && P.n_extra_streams > 0) {                                 // the rest of the list (config.py:99 has no length limit): scalar loads from the device table
                    const DevStreamTable xs = (DevStreamTable)P.extra_streams;
                    for (int x = 0; x < P.n_extra_streams; ++x) {
                        DevStream S;
                        S.amount = xs[x].amount; S.keep = xs[x].keep; S.amount_keep = xs[x].amount_keep; S.start_month = xs[x].start_month;
                        S.end_month = xs[x].end_month; S.indexed = xs[x].indexed; S.lock_slot = xs[x].lock_slot;
                        stream_income(S);
                    }
                }
                const double need = fmax(0.0, TOL ? income : expenses - income);      // :679-682
                bool stop = false;
                if (b1 + b2 <= kEps && need > kEps) {                  // :684-690 (FAIL-1, no shock consumed)
                    yfail = true; stop = true;
                }
                if (!stop) {
                    if (!kStaged) growth(wm + rmi, g1, ginf, g2);      // :692-705 (sequential generators draw here)
                    market_step<ANNUAL, TOL>(g1, ginf, g2, b1, b2, gacc1, gacc2, infl);  // :706-714
                    if (b1 + b2 <= kEps && need > kEps) {              // :717-724 (FAIL-2)
                        MCR_MASKED_MOVE;                              // keep it a branch: no lane takes it in most months
                        b1 = fmax(0.0, b1); b2 = fmax(0.0, b2);
                        yfail = true; stop = true;
                    }
                }
                if (!stop && TOL) {
                    // the withdrawal in closed form (mcr_device.h): both assets sell the fraction target / capacity
                    const double cap = capacity_tol<T1, T2, MM>(b1, c1, L.real_rate1, b2, c2, L.real_rate2);   // :726-738
                    const double target = fmin(need, cap);                            // :739-742
                    if (need > kEps && target < need - kEps) yfail = true;            // :743-748 (FAIL-3) = :784-790 (FAIL-4): the net cash is the target
                    double phi = target * recip_nr<false>(cap);                       // :750-765
                    if (!(cap > 0.0)) { MCR_MASKED_MOVE; phi = 0.0; }                 // (0 < cap <= 1e-6: still the capacity shares, see mcr_device.h)
                    if (kSummary) {
                        const double gross = phi * (b1 + b2);                         // :766, :777: gross withdrawals of the month
                        tg1 += gross;
                        treal = __builtin_fma(gross * infl_ret, recip_nr<false>(fmax(price, kEps)), treal);  // :778-782
                    }
                    sell_fraction_tol<MM>(phi, b1, c1, b2, c2);                       // :757-776
                    rebalance_tol<TANY, MM>(P, L, b1, c1, b2, c2);                    // :792-796
                    if (!yfail && (wm + rmi + 1) % kMPY == 0) {                       // :798-804
                        const bool tf = annual_gain_taxes<false, TANY, ANNUAL, T1, T2, MM, true>(P, L, b1, c1, b2, c2, gacc1, gacc2);  // :805-818
                        gacc1 = 0.0; gacc2 = 0.0;                                     // :819-820
                        yfail = yfail || tf;                                          // :821-822
                    }
                }
                if (!stop && !TOL) {
                    double cap1, cap2;
                    net_liquidation_values2<T1, T2, MM>(b1, c1, L.real_rate1, b2, c2, L.real_rate2, cap1, cap2);  // :726-737
                    const double cap = cap1 + cap2;                                   // :738
                    const double target = fmin(need, cap);                            // :739-742 (need, cap >= 0: the max(0, .) is a no-op)
                    if (need > kEps && target < need - kEps) yfail = true;            // :743-748 (FAIL-3)
                    double prop1 = fdiv<false>(cap1, cap);                            // :750-754
                    if (!(cap > kEps)) { MCR_MASKED_MOVE; prop1 = P.alloc1; }        // (exec-masked move, not a select)
                    const double prop2 = 1.0 - prop1;                                 // :755
                    double gw1, nw1, gw2, nw2;
                    withdraw2<T1, T2, MM>(b1, c1, target * prop1, L.real_rate1, gw1, nw1,  // :757-765
                                     b2, c2, target * prop2, L.real_rate2, gw2, nw2);  // :768-776
                    tg1 += gw1;                                                       // :766
                    tg2 += gw2;                                                       // :777
                    if (kSummary) treal += fdiv<false>((gw1 + gw2) * infl_ret, fmax(price, kEps));  // :778-782
                    if (need > kEps && nw1 + nw2 < need - kEps) yfail = true;         // :784-790 (FAIL-4)
                    rebalance_path<TANY, MM>(L, b1, c1, b2, c2);                      // :792-796
                    if (!yfail && (wm + rmi + 1) % kMPY == 0) {                       // :798-804
                        const bool tf = annual_gain_taxes<false, TANY, ANNUAL, T1, T2, MM>(P, L, b1, c1, b2, c2, gacc1, gacc2);  // :805-818
                        gacc1 = 0.0; gacc2 = 0.0;                                     // :819-820
                        yfail = yfail || tf;                                          // :821-822
                    }
                }
                if (yfail) fail_rmi = rmi;  // :825-828, :844-847
            }
        }
        // ---- year end (:830-868); lanes that were already dead pad with 0 / NaN (:902-916,:934-935) ----
        double sample = 0.0;
        unsigned long long wr_bits = kNanBits;
        if (alive) {
            const double ygw = tg1 + tg2;                                              // :830-832
            const double wr_pct = start_balance > kEps ? (treal / start_balance) * 100.0 : 0.0;  // :834-840
            if (year == 0) {                                                           // :852-856, :861-865
                if (kSumLds) { sum_col[0] = ygw; sum_col[kBlock] = treal; }
                else { fy_gross = ygw; fy_real = treal; }
            }
            if (yfail) {
                succeeded = false;                                                     // :843
                if (kSumLds) sum_col[2 * kBlock] = (double)(fail_rmi + 1) / (double)kMPY;  // :825-827, :844-847
                else ytr_bits = f64_bits((double)(fail_rmi + 1) / (double)kMPY);
                ruin_bin = 1 + year;
                sample = fmax(0.0, b1 + b2);                                           // :848
                alive = false;                                                         // :857
            } else {
                wr_bits = f64_bits(wr_pct);                                            // :859
                sample = b1 + b2;                                                      // :867
                done_years = year + 1;
            }
        }
        put_sample(t_idx, sample, infl);  // dead lanes: 0 / px = 0 (:906-916, :928-931)
        ++t_idx;
        if (kTraj && valid && wrt) store_bits(&wrt[(int64_t)year * stride + (int64_t)li], wr_bits);  // :851, :859, :934-935
    }
    for (; year < (kSliced ? y_end : ry); ++year) {  // the whole wave failed early: pad (:902-916, :934-935)
        put_sample(t_idx++, 0.0, infl);
        if (kTraj && valid && wrt) store_bits(&wrt[(int64_t)year * stride + (int64_t)li], kNanBits);
    }
    if (kSliced && seg >= 0 && seg < io.seg_q - 1) {
        // not the block's last segment: hand the lanes' state over and raise the flag (every wave gets here: no early return above)
        *seg_at(0) = b1; *seg_at(1) = b2; *seg_at(2) = c1; *seg_at(3) = c2;
        if (ANNUAL) { *seg_at(4) = gacc1; *seg_at(5) = gacc2; }
        *seg_at(6) = infl;
        store_bits(seg_at(7), (alive ? 1ull : 0ull) | (succeeded ? 2ull : 0ull) | ((unsigned long long)(ruin_bin + 1) << 8) | ((unsigned long long)done_years << 24));
        store_bits(seg_at(8), ((unsigned long long)carry.w3 << 32) | (unsigned long long)carry.w2);
        for (int k = 0; k < P.n_lock_slots; ++k) *seg_at(kSegFixedFields + k) = lock_lds[(size_t)k * kBlock + tid];
        if (MODE >= 1) {
            *seg_at(9) = start_balance; *seg_at(10) = infl_ret;
            *seg_at(11) = sum_col[0]; *seg_at(12) = sum_col[kBlock]; *seg_at(13) = sum_col[2 * kBlock];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_store(io.seg_flags + (size_t)seg_block * (size_t)io.seg_q + (size_t)seg, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }

    // ---- terminal partial tax period (:873-898) ----
    if (P.total_months % kMPY != 0) {  // wave-uniform
        if (succeeded) {
            const bool tf = annual_gain_taxes<false, TANY, ANNUAL, T1, T2, MM, TOL>(P, L, b1, c1, b2, c2, gacc1, gacc2);  // :880-893
            if (tf) {                                                            // :894-896
                succeeded = false; ruin_bin = ry + 1;
                if (kSumLds) sum_col[2 * kBlock] = (double)ry; else ytr_bits = f64_bits((double)ry);
            }
            put_sample(P.trajectory_len - 1, b1 + b2, infl);                     // :897-898
        }
    }
    const double final_balance = fmax(0.0, b1 + b2);  // :900, :941

    // ---- outputs ----
    if (kSumLds) { fy_gross = sum_col[0]; fy_real = sum_col[kBlock]; ytr_bits = f64_bits(sum_col[2 * kBlock]); }
    if (kSummary && valid) {
        const mcr_outputs& o = io.out;
        if (o.start_balance) o.start_balance[li] = start_balance;
        if (o.final_balance) o.final_balance[li] = final_balance;
        if (o.years_to_ruin) store_bits(&o.years_to_ruin[li], ytr_bits);
        if (o.first_year_gross_withdrawal) o.first_year_gross_withdrawal[li] = fy_gross;
        if (o.first_year_real_gross_withdrawal) o.first_year_real_gross_withdrawal[li] = fy_real;
        if (o.inflation_at_retirement) o.inflation_at_retirement[li] = infl_ret;
        if (o.success) o.success[li] = succeeded ? 1 : 0;
    }
#ifdef MCR_K1_TIMELINE   // the stamps go to a buffer of their own (the otherwise unused path_seeds pointer of a Philox launch)
    if (MODE == 0 && RNG == 0 && io.path_seeds && (threadIdx.x & 63) == 0) {
        unsigned long long* tl = (unsigned long long*)io.path_seeds + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 3;
        tl[0] = tl_t0; tl[1] = wall_clock64();
        tl[2] = ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32) | __builtin_amdgcn_s_getreg(4 | (31 << 11));  // XCC_ID | HW_ID
    }
#endif
    // success count: wave ballot + popcount -> LDS -> one atomic per workgroup
    const unsigned long long ok = __builtin_amdgcn_ballot_w64(valid && succeeded);
    if ((threadIdx.x & 63) == 0) atomicAdd(&blk[0], (unsigned int)__popcll(ok));
    const bool want_bins = io.out.ruin_year_bins != nullptr || io.out.wr_obs_counts != nullptr;
    if (want_bins && valid) {
        if (ruin_bin >= 0) atomicAdd(&blk[1 + ruin_bin], 1u);
        atomicAdd(&blk[1 + (ry + 2) + done_years], 1u);
    }
    // "Final Balance" of the successful cohort on the caller's bin edges (plotting.py:44-59; np.histogram(x, bins=edges):
    // bin k = [e_k, e_k+1), the last one closed, values outside the edges dropped).  Once per path, after 10^2..10^3
    // months of arithmetic: a per-lane binary search straight over the (L2-resident) edge array costs nothing measurable.
    if ((PHASE == 0 || PHASE == 3) && n_hist > 0 && valid && succeeded) {
        const double* __restrict__ e = io.out.hist_edges;
        if (final_balance >= e[0] && final_balance <= e[n_hist]) {
            int lo = 0, hi = n_hist;             // e[lo] <= x and (hi == n_hist or x < e[hi])
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (e[mid] <= final_balance) lo = mid; else hi = mid;
            }
            atomicAdd(&blk[n_blk + lo], 1u);
        }
    }
    __syncthreads();
    uint64_t* ctr = io.out.counters ? io.out.counters + (kCand ? (size_t)io.cand_out[cand] * MCR_N_COUNTERS : 0) : nullptr;
    if (threadIdx.x == 0 && ctr) {
        atomicAdd((unsigned long long*)&ctr[MCR_CTR_SUCCESS], (unsigned long long)blk[0]);
        const uint64_t first = (uint64_t)path_block * kBlock;
        const uint64_t cnt = io.n_paths - first < (uint64_t)kBlock ? io.n_paths - first : (uint64_t)kBlock;
        atomicAdd((unsigned long long*)&ctr[MCR_CTR_PATHS], (unsigned long long)cnt);
    }
    if (want_bins) {
        for (int k = threadIdx.x; k < ry + 2; k += kBlock) {
            if (io.out.ruin_year_bins && blk[1 + k])
                atomicAdd((unsigned long long*)&io.out.ruin_year_bins[k], (unsigned long long)blk[1 + k]);
        }
        // wr_obs_counts[y] = #paths with done_years > y   (wr_df.count(axis=1), :1111-1113)
        for (int y = threadIdx.x; y < ry; y += kBlock) {
            unsigned int c = 0;
            for (int d = y + 1; d <= ry; ++d) c += blk[1 + (ry + 2) + d];
            if (io.out.wr_obs_counts && c)
                atomicAdd((unsigned long long*)&io.out.wr_obs_counts[y], (unsigned long long)c);
        }
    }
    for (int k = threadIdx.x; k < n_hist; k += kBlock)   // one global atomic per non-empty bin per workgroup
        if (blk[n_blk + k]) atomicAdd((unsigned long long*)&io.out.hist_bins[k], (unsigned long long)blk[n_blk + k]);
}

// ---------------------------------------------------------------------------------------------
// Device unit functions exposed for the reference's helper-level tests
// ---------------------------------------------------------------------------------------------
__global__ void helper_kernel(int which, const DevParams P, const double* in, double* out, int64_t n) {
    __shared__ double tab[kTabDoubles];
    load_math_tables(tab, threadIdx.x, blockDim.x);
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    switch (which) {
        case MCR_HELPER_WITHDRAW: {
            const double* x = in + 5 * i;
            double bal = x[0], cb = x[1], g, nt;
            const double rate = (x[3] != 0.0 && x[4] > 0.0) ? x[4] : 0.0;  // use_real_tax and rate > 0 (:224)
            withdraw(bal, cb, x[2], rate, g, nt);
            out[4 * i + 0] = bal; out[4 * i + 1] = cb; out[4 * i + 2] = g; out[4 * i + 3] = nt;
            break;
        }
        case MCR_HELPER_NLV: {
            const double* x = in + 4 * i;
            const double rate = (x[2] != 0.0 && x[3] > 0.0) ? x[3] : 0.0;  // :269
            out[i] = net_liquidation_value(x[0], x[1], rate);
            break;
        }
        case MCR_HELPER_REBALANCE: {
            const double* x = in + 4 * i;
            double b1 = x[0], c1 = x[1], b2 = x[2], c2 = x[3];
            rebalance(lane_params(P), b1, c1, b2, c2);
            out[4 * i + 0] = b1; out[4 * i + 1] = c1; out[4 * i + 2] = b2; out[4 * i + 3] = c2;
            break;
        }
        case MCR_HELPER_ANNUAL_TAX: {
            const double* x = in + 6 * i;
            double b1 = x[0], c1 = x[1], b2 = x[2], c2 = x[3];
            const bool tf = annual_gain_taxes(P, lane_params(P), b1, c1, b2, c2, x[4], x[5]);
            out[5 * i + 0] = b1; out[5 * i + 1] = c1; out[5 * i + 2] = b2; out[5 * i + 3] = c2;
            out[5 * i + 4] = tf ? 1.0 : 0.0;
            break;
        }
        case MCR_HELPER_MONTHLY_GROSS: {
            const double* x = in + 3 * i;
            out[i] = monthly_gross(x[0] / (double)kMPY, x[1] / sqrt((double)kMPY), x[2], tab);  // :473
            break;
        }
        case MCR_HELPER_MATH_EXP: out[i] = fexp(in[i], tab, MathRegs::literals()); break;
        case MCR_HELPER_MATH_DIV: out[i] = fdiv(in[2 * i], in[2 * i + 1]); break;
        case MCR_HELPER_MATH_DIV_PATH: out[i] = fdiv<false>(in[2 * i], in[2 * i + 1]); break;
        case MCR_HELPER_MATH_SQRT: out[i] = fsqrt(in[i]); break;
        case MCR_HELPER_WITHDRAW2_PATH: {
            const double* x = in + 6 * i;
            const LaneParams L = lane_params(P);
            double b1 = x[0], c1 = x[1], b2 = x[3], c2 = x[4], g1, n1, g2, n2;
            switch (P.tax_mask) {   // the per-asset forms the path kernel runs for this parameter block
                case 0: withdraw2<false, false>(b1, c1, x[2], L.real_rate1, g1, n1, b2, c2, x[5], L.real_rate2, g2, n2); break;
                case 1: withdraw2<true, false>(b1, c1, x[2], L.real_rate1, g1, n1, b2, c2, x[5], L.real_rate2, g2, n2); break;
                case 2: withdraw2<false, true>(b1, c1, x[2], L.real_rate1, g1, n1, b2, c2, x[5], L.real_rate2, g2, n2); break;
                default: withdraw2<true, true>(b1, c1, x[2], L.real_rate1, g1, n1, b2, c2, x[5], L.real_rate2, g2, n2); break;
            }
            double* o = out + 8 * i;
            o[0] = b1; o[1] = c1; o[2] = g1; o[3] = n1; o[4] = b2; o[5] = c2; o[6] = g2; o[7] = n2;
            break;
        }
        case MCR_HELPER_NLV2_PATH: {
            const double* x = in + 4 * i;
            const LaneParams L = lane_params(P);
            double v1, v2;
            switch (P.tax_mask) {
                case 0: net_liquidation_values2<false, false>(x[0], x[1], L.real_rate1, x[2], x[3], L.real_rate2, v1, v2); break;
                case 1: net_liquidation_values2<true, false>(x[0], x[1], L.real_rate1, x[2], x[3], L.real_rate2, v1, v2); break;
                case 2: net_liquidation_values2<false, true>(x[0], x[1], L.real_rate1, x[2], x[3], L.real_rate2, v1, v2); break;
                default: net_liquidation_values2<true, true>(x[0], x[1], L.real_rate1, x[2], x[3], L.real_rate2, v1, v2); break;
            }
            out[2 * i] = v1; out[2 * i + 1] = v2;
            break;
        }
        case MCR_HELPER_REBALANCE_PATH: {
            const double* x = in + 4 * i;
            double b1 = x[0], c1 = x[1], b2 = x[2], c2 = x[3];
            if (P.any_real_rate) rebalance_path<true>(lane_params(P), b1, c1, b2, c2);
            else rebalance_path<false>(lane_params(P), b1, c1, b2, c2);
            out[4 * i + 0] = b1; out[4 * i + 1] = c1; out[4 * i + 2] = b2; out[4 * i + 3] = c2;
            break;
        }
        case MCR_HELPER_ANNUAL_TAX_PATH: {
            const double* x = in + 6 * i;
            double b1 = x[0], c1 = x[1], b2 = x[2], c2 = x[3];
            const LaneParams L = lane_params(P);
            bool tf;
#define MCR_ATAX(T1_, T2_) (P.any_annual_tax ? annual_gain_taxes<false, (T1_) || (T2_), true, T1_, T2_>(P, L, b1, c1, b2, c2, x[4], x[5]) \
                                             : annual_gain_taxes<false, (T1_) || (T2_), false, T1_, T2_>(P, L, b1, c1, b2, c2, x[4], x[5]))
            switch (P.tax_mask) {
                case 0: tf = MCR_ATAX(false, false); break;
                case 1: tf = MCR_ATAX(true, false); break;
                case 2: tf = MCR_ATAX(false, true); break;
                default: tf = MCR_ATAX(true, true); break;
            }
#undef MCR_ATAX
            out[5 * i + 0] = b1; out[5 * i + 1] = c1; out[5 * i + 2] = b2; out[5 * i + 3] = c2;
            out[5 * i + 4] = tf ? 1.0 : 0.0;
            break;
        }
        case MCR_HELPER_WITHDRAW_MONTH: {   // the month's withdrawal as the path kernel runs it for this parameter block
            const double* x = in + 5 * i;
            double b1 = x[0], c1 = x[1], b2 = x[2], c2 = x[3];
            const double need = x[4];
            double gross, net;
            if (P.exact_month) {            // (:726-790 in the exact path forms, as in path_kernel<..., EXACT = true>)
                const LaneParams L = lane_params(P);
                double cap1, cap2, g1, n1, g2, n2;
                net_liquidation_values2<true, true>(b1, c1, L.real_rate1, b2, c2, L.real_rate2, cap1, cap2);
                const double cap = cap1 + cap2, target = fmin(need, cap);
                double prop1 = fdiv<false>(cap1, cap);
                if (!(cap > kEps)) prop1 = P.alloc1;
                withdraw2<true, true>(b1, c1, target * prop1, L.real_rate1, g1, n1, b2, c2, target * (1.0 - prop1), L.real_rate2, g2, n2);
                gross = g1 + g2; net = n1 + n2;
            } else {                        // the closed form (mcr_device.h: TOLERANCE FORM of the month)
                const LaneParams L = lane_params_tol(P);
                const double cap = P.tax_mask == 3 ? capacity_tol<true, true>(b1, c1, L.real_rate1, b2, c2, L.real_rate2)
                                 : P.tax_mask == 2 ? capacity_tol<false, true>(b1, c1, L.real_rate1, b2, c2, L.real_rate2)
                                 : P.tax_mask == 1 ? capacity_tol<true, false>(b1, c1, L.real_rate1, b2, c2, L.real_rate2)
                                                   : capacity_tol<false, false>(b1, c1, L.real_rate1, b2, c2, L.real_rate2);
                const double target = fmin(need, cap);
                double phi = target * recip_nr<false>(cap);
                if (!(cap > 0.0)) phi = 0.0;
                gross = phi * (b1 + b2); net = target;
                sell_fraction_tol<true>(phi, b1, c1, b2, c2);
            }
            double* o = out + 6 * i;
            o[0] = b1; o[1] = c1; o[2] = b2; o[3] = c2; o[4] = gross; o[5] = net;
            break;
        }
        case MCR_HELPER_REBALANCE_MONTH: {  // the month's rebalance as the path kernel runs it for this parameter block
            const double* x = in + 4 * i;
            double b1 = x[0], c1 = x[1], b2 = x[2], c2 = x[3];
            if (P.exact_month) rebalance_path<true>(lane_params(P), b1, c1, b2, c2);
            else if (P.any_real_rate) rebalance_tol<true>(P, lane_params_tol(P), b1, c1, b2, c2);
            else rebalance_tol<false>(P, lane_params_tol(P), b1, c1, b2, c2);
            out[4 * i + 0] = b1; out[4 * i + 1] = c1; out[4 * i + 2] = b2; out[4 * i + 3] = c2;
            break;
        }
        case MCR_HELPER_MATH_NEG2LOG: out[i] = neg2_log_u32((uint32_t)in[i], tab, MathRegs::literals()); break;
        case MCR_HELPER_MATH_EXP_PATH: out[i] = fexp<true>(in[i], tab, MathRegs::literals_path()); break;
        case MCR_HELPER_MATH_NEG2LOG_PATH: out[i] = neg2_log_u32<true>((uint32_t)in[i], tab, MathRegs::literals_path()); break;
        case MCR_HELPER_MATH_SINCOS_PATH: {
            double sn, cs;
            sincos_u32<true, true>((uint32_t)in[i], tab, MathRegs::literals_path(), sn, cs);
            out[2 * i] = sn; out[2 * i + 1] = cs;
            break;
        }
        case MCR_HELPER_MATH_SINCOS: {
            double sn, cs;
            sincos_u32<true>((uint32_t)in[i], tab, MathRegs::literals(), sn, cs);
            out[2 * i] = sn; out[2 * i + 1] = cs;
            break;
        }
        default: break;
    }
}

// _draw_shock_path (:452-466): out[n_paths][n_months][3]; one thread per path, rows in order
__global__ void shocks_kernel(uint64_t seed, uint32_t stream_id, uint64_t path_begin, uint64_t n_paths,
                              int32_t n_months, double rho, double rho_c, double* out) {
    __shared__ double tab[kTabDoubles];
    load_math_tables(tab, threadIdx.x, blockDim.x);
    __syncthreads();
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_paths) return;
    ShockGen G{0.0, 0u, 0u};
    double* o = out + (size_t)p * 3u * (size_t)n_months;
    for (int32_t m = 0; m < n_months; ++m)
        shock_row_seq(G, seed, stream_id, path_begin + p, (uint32_t)m, rho, rho_c, tab, o[3 * m], o[3 * m + 1], o[3 * m + 2]);
}

// _draw_shock_path with the reference's NumPy stream: one thread per path, rows in order
__global__ void np_shocks_kernel(const KernelIO io, int32_t n_months, double rho, double rho_c, double* out) {
    __shared__ __align__(16) unsigned char zraw[kZigLdsBytes];
    __shared__ double mtab[kTabDoubles];
    load_math_tables(mtab, threadIdx.x, blockDim.x);
    ZigTables zig = load_zig_tables(zraw, threadIdx.x, blockDim.x);
    zig.math_tab = mtab;
    __syncthreads();
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= io.n_paths) return;
    Pcg64 g;
    const uint32_t s32 = io.path_seeds ? io.path_seeds[i]
                                       : np_path_seed(io.entropy, (int)io.n_entropy, io.stream_id, io.child_offset + io.path_begin + i);
    pcg64_seed_u32(g, s32);
    double* o = out + (size_t)i * 3u * (size_t)n_months;
    for (int32_t m = 0; m < n_months; ++m) {
        const double z0 = np_standard_normal(g, zig), z1 = np_standard_normal(g, zig), z2 = np_standard_normal(g, zig);
        o[3 * m + 0] = z0; o[3 * m + 1] = rho * z0 + rho_c * z1; o[3 * m + 2] = z2;
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return MCR_ERR_HIP;
}

int use_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        set_error("no usable HIP device (the engine has no CPU fallback)");
        return MCR_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) {
        set_error("device %d out of range (%d devices)", device, n);
        return MCR_ERR_INVALID_ARG;
    }
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    return MCR_OK;
}

// ---- leased contexts of the host-buffer entry points (mcr_host.h): a process-wide pool keyed by device ----
static std::mutex g_pool_mu;
static std::vector<HostCtx*> g_idle_ctx;   // idle contexts, most recently used last

static void host_ctx_destroy(HostCtx* c) {   // (the caller has set the device)
    if (c->block) (void)hipFree(c->block);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

HostCtx* host_ctx_acquire(int device) {
    {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        for (size_t i = g_idle_ctx.size(); i-- > 0;)
            if (g_idle_ctx[i]->device == device) {
                HostCtx* c = g_idle_ctx[i];
                g_idle_ctx.erase(g_idle_ctx.begin() + (long)i);
                return c;
            }
    }
    HostCtx* c = new HostCtx{device, nullptr, nullptr, 0};
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { (void)hip_fail(e, "hipStreamCreate (host context)"); delete c; return nullptr; }
    return c;
}

void host_ctx_release(HostCtx* c) {
    if (c->capacity > kHostCtxKeepBytes) { (void)hipFree(c->block); c->block = nullptr; c->capacity = 0; }
    HostCtx* surplus = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        g_idle_ctx.push_back(c);
        int same = 0;
        for (HostCtx* o : g_idle_ctx) same += o->device == c->device;
        if (same > kHostCtxIdlePerDevice)   // drop the least recently used idle context of this device
            for (size_t i = 0; i < g_idle_ctx.size(); ++i)
                if (g_idle_ctx[i]->device == c->device) { surplus = g_idle_ctx[i]; g_idle_ctx.erase(g_idle_ctx.begin() + (long)i); break; }
    }
    if (surplus) host_ctx_destroy(surplus);   // same device as the call that is ending: it is still current
}

hipError_t host_ctx_reserve(HostCtx* c, size_t bytes) {
    if (bytes <= c->capacity) return hipSuccess;
    if (c->block) { (void)hipFree(c->block); c->block = nullptr; c->capacity = 0; }
    hipError_t e = hipMalloc(&c->block, bytes);
    if (e == hipSuccess) c->capacity = bytes; else c->block = nullptr;
    return e;
}

// Fork/join streams (mcr_host.h: StreamFork): a few non-blocking side streams + events, leased from a process-wide pool
// keyed by device for the duration of ONE call's enqueue (like HostCtx).  A lease only has to cover the host-side
// enqueue: the side streams are in-order, and a stream that waits on an event waits for the record that preceded the
// wait call, so the next lessee's records cannot disturb work that is still running.  User: mcr_probe_months_rng
// (candidates forked onto side streams; mcr_row_quantiles' row-group pipelining was measured and dropped in round 3).
static std::vector<StreamFork*> g_idle_fork;   // guarded by g_pool_mu
constexpr int kStreamForkIdlePerDevice = 2;

static void stream_fork_destroy(StreamFork* f) {   // pending work on a destroyed stream still completes (stream-ordered release)
    for (int i = 0; i < kForkStreams; ++i) {
        if (f->done[i]) (void)hipEventDestroy(f->done[i]);
        if (f->side[i]) (void)hipStreamDestroy(f->side[i]);
    }
    if (f->fork) (void)hipEventDestroy(f->fork);
    delete f;
}
StreamFork* stream_fork_acquire(int device) {
    {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        for (size_t i = g_idle_fork.size(); i-- > 0;)
            if (g_idle_fork[i]->device == device) {
                StreamFork* f = g_idle_fork[i];
                g_idle_fork.erase(g_idle_fork.begin() + (long)i);
                return f;
            }
    }
    StreamFork* f = new StreamFork();
    std::memset(f, 0, sizeof(*f));
    f->device = device;
    bool ok = hipEventCreateWithFlags(&f->fork, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; ok && i < kForkStreams; ++i)
        ok = hipStreamCreateWithFlags(&f->side[i], hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&f->done[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); stream_fork_destroy(f); return nullptr; }
    return f;
}
void stream_fork_release(StreamFork* f) {
    StreamFork* surplus = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        g_idle_fork.push_back(f);
        int same = 0;
        for (StreamFork* o : g_idle_fork) same += o->device == f->device;
        if (same > kStreamForkIdlePerDevice)
            for (size_t i = 0; i < g_idle_fork.size(); ++i)
                if (g_idle_fork[i]->device == f->device) { surplus = g_idle_fork[i]; g_idle_fork.erase(g_idle_fork.begin() + (long)i); break; }
    }
    if (surplus) stream_fork_destroy(surplus);
}

// Preconditions of the path kernel (its STRICT = false forms drop clamps that are no-ops only for rates and
// weights in [0, 1] and non-negative amounts; fexp needs |x| < 700).  The reference enforces the same ranges
// in its pydantic Config (backend/config.py:56-99); a ctypes caller that bypasses Config gets an error here,
// not silently different arithmetic.
// other_income_streams has any length (backend/config.py:99): the first MCR_INLINE_STREAMS records sit in the block, the
// rest behind mcr_params.extra_streams (a host pointer)
static inline const mcr_stream& stream_at(const mcr_params* p, int s) {
    return s < MCR_INLINE_STREAMS ? p->streams[s] : p->extra_streams[s - MCR_INLINE_STREAMS];
}
static int check_stream_list(const mcr_params* p) {
    if (p->n_streams < 0) { set_error("n_streams %d must be >= 0", p->n_streams); return MCR_ERR_INVALID_ARG; }
    if (p->n_streams > MCR_INLINE_STREAMS && !p->extra_streams) {
        set_error("n_streams = %d but extra_streams is NULL (entries %d.. of the list go there)", p->n_streams, MCR_INLINE_STREAMS);
        return MCR_ERR_INVALID_ARG;
    }
    return MCR_OK;
}

static int validate_params(const mcr_params* p) {
    if (!p) { set_error("null params"); return MCR_ERR_INVALID_ARG; }
    auto bad = [](const char* name, double v, const char* want) {
        set_error("params.%s = %g: must be %s", name, v, want);
        return MCR_ERR_INVALID_ARG;
    };
    auto unit = [](double v) { return v >= 0.0 && v <= 1.0; };            // false for NaN
    auto nonneg = [](double v) { return v >= 0.0 && std::isfinite(v); };
    if (!nonneg(p->initial_balance)) return bad("initial_balance", p->initial_balance, "finite and >= 0 (config.py:56)");
    if (!nonneg(p->monthly_contribution)) return bad("monthly_contribution", p->monthly_contribution, "finite and >= 0 (config.py:57)");
    if (!nonneg(p->contribution_growth_rate_annual)) return bad("contribution_growth_rate_annual", p->contribution_growth_rate_annual, "finite and >= 0 (config.py:58)");
    if (!nonneg(p->monthly_expenses)) return bad("monthly_expenses", p->monthly_expenses, "finite and >= 0 (config.py:59)");
    if (!std::isfinite(p->current_age)) return bad("current_age", p->current_age, "finite (config.py:62)");
    if (!unit(p->allocation_inv1_pct)) return bad("allocation_inv1_pct", p->allocation_inv1_pct, "in [0, 1] (config.py:70)");
    if (!unit(p->inv1_annual_tax_on_gains_rate)) return bad("inv1_annual_tax_on_gains_rate", p->inv1_annual_tax_on_gains_rate, "in [0, 1] (config.py:73)");
    if (!unit(p->inv1_realized_gains_tax_rate)) return bad("inv1_realized_gains_tax_rate", p->inv1_realized_gains_tax_rate, "in [0, 1] (config.py:74)");
    if (!unit(p->inv2_annual_tax_on_gains_rate)) return bad("inv2_annual_tax_on_gains_rate", p->inv2_annual_tax_on_gains_rate, "in [0, 1] (config.py:79)");
    if (!unit(p->inv2_realized_gains_tax_rate)) return bad("inv2_realized_gains_tax_rate", p->inv2_realized_gains_tax_rate, "in [0, 1] (config.py:80)");
    if (!(p->equity_inflation_rho >= -1.0 && p->equity_inflation_rho <= 1.0)) return bad("equity_inflation_rho", p->equity_inflation_rho, "in [-1, 1] (config.py:85)");
    // monthly log-growth a + b z with |z| < 40 (Box-Muller of 32-bit uniforms: |z| < 6.8, rho-mix < 9.6; ziggurat tail far
    // below 40) must stay inside fexp's domain
    const double sq12 = std::sqrt((double)kMPY);
    const double mu[3] = {p->inv1_mu_log, p->inf_mu_log, p->prem_mu_log}, sg[3] = {p->inv1_sigma_log, p->inf_sigma_log, p->prem_sigma_log};
    const char* nm[3] = {"inv1", "inf", "prem"};
    for (int i = 0; i < 3; ++i) {
        if (!std::isfinite(mu[i]) || !std::isfinite(sg[i]) || sg[i] < 0.0 || std::fabs(mu[i]) / kMPY + 40.0 * sg[i] / sq12 >= 700.0) {
            set_error("params.%s_mu_log / %s_sigma_log = %g / %g: need finite values, sigma >= 0 and |mu|/12 + 40 sigma/sqrt(12) < 700",
                      nm[i], nm[i], mu[i], sg[i]);
            return MCR_ERR_INVALID_ARG;
        }
    }
    if (int rc = check_stream_list(p)) return rc;
    for (int s = 0; s < p->n_streams; ++s) {
        const mcr_stream& st = stream_at(p, s);
        if (!nonneg(st.monthly_amount_today) || !unit(st.tax_rate) || !std::isfinite(st.start_at_age)) {
            set_error("params.streams[%d]: monthly_amount_today %g must be finite and >= 0, tax_rate %g in [0, 1], start_at_age %g finite "
                      "(config.py:18,23,45)", s, st.monthly_amount_today, st.tax_rate, st.start_at_age);
            return MCR_ERR_INVALID_ARG;
        }
    }
    return MCR_OK;
}

static int query_sizes(const mcr_params* p, int32_t wm, mcr_sizes* s) {
    if (!p || !s) { set_error("null argument"); return MCR_ERR_INVALID_ARG; }
    if (wm < 0) { set_error("working_months must be >= 0 (got %d)", wm); return MCR_ERR_INVALID_ARG; }
    if (p->retirement_years <= 0) { set_error("retirement_years must be > 0"); return MCR_ERR_INVALID_ARG; }
    if (int rc = check_stream_list(p)) return rc;
    if ((int64_t)wm + (int64_t)p->retirement_years * kMPY > (int64_t)INT32_MAX / 4) {
        set_error("horizon too long");
        return MCR_ERR_INVALID_ARG;
    }
    s->total_months = wm + p->retirement_years * kMPY;                   // simulation.py:487
    s->shock_rows = s->total_months > 1 ? s->total_months : 1;           // :488
    s->num_working_years = wm > 0 ? (wm + kMPY - 1) / kMPY : 0;          // :585-589
    s->trajectory_len = 1 + s->num_working_years + p->retirement_years;  // :902
    s->retirement_years = p->retirement_years;
    s->ruin_bins = p->retirement_years + 2;
    return MCR_OK;
}

// stream_payment_start_month_index (simulation.py:47-63)
static int32_t start_month_index(double current_age, int32_t wm, double start_at_age) {
    const double retirement_start = current_age + (double)wm / (double)kMPY;  // :34
    const double eligible = start_at_age > retirement_start ? start_at_age : retirement_start;  // max(ret, start) :44
    const double c = std::ceil((eligible - retirement_start) * (double)kMPY - kEps);            // :58-61
    if (!(c > 0.0)) return 0;
    if (c > (double)(INT32_MAX / 2)) return INT32_MAX / 2;
    return (int32_t)c;
}

// Host-side derivation of the wave-uniform parameter block (same fp64 expressions as the reference).
// `extra` receives the records of the streams beyond the by-value block (device-table layout); callers that cannot carry
// such a table pass nullptr and get MCR_ERR_UNSUPPORTED for longer lists.
static int derive_params(const mcr_params* p, int32_t wm, DevParams* d, std::vector<DevStream>* extra = nullptr) {
    mcr_sizes sz;
    int rc = query_sizes(p, wm, &sz);
    if (rc != MCR_OK) return rc;
    rc = validate_params(p);
    if (rc != MCR_OK) return rc;
    std::memset(d, 0, sizeof(*d));
    d->initial_balance = p->initial_balance;
    d->monthly_contribution = p->monthly_contribution;
    d->contrib_growth_factor = 1 + p->contribution_growth_rate_annual;  // :517
    d->contrib_grows = p->contribution_growth_rate_annual > 0;          // :516
    d->monthly_expenses = p->monthly_expenses;
    d->alloc1 = p->allocation_inv1_pct;
    d->alloc2 = 1.0 - p->allocation_inv1_pct;  // config.py:124-126
    // "use_real_tax and rate > 0" (:224,:238,:269) and "if use_realized" (:304,:317): a zero rate
    // multiplies to exactly 0.0, so one effective rate covers both spellings.
    d->real_rate1 = (p->inv1_use_realized_gains_tax_system && p->inv1_realized_gains_tax_rate > 0) ? p->inv1_realized_gains_tax_rate : 0.0;
    d->real_rate2 = (p->inv2_use_realized_gains_tax_system && p->inv2_realized_gains_tax_rate > 0) ? p->inv2_realized_gains_tax_rate : 0.0;
    d->annual_rate1 = !p->inv1_use_realized_gains_tax_system ? p->inv1_annual_tax_on_gains_rate : 0.0;  // :380-384
    d->annual_rate2 = !p->inv2_use_realized_gains_tax_system ? p->inv2_annual_tax_on_gains_rate : 0.0;  // :385-389
    d->any_annual_tax = (d->annual_rate1 > 0.0) || (d->annual_rate2 > 0.0);
    d->any_real_rate = (d->real_rate1 > 0.0) || (d->real_rate2 > 0.0);
    d->tax_mask = (d->real_rate1 > 0.0 ? 1 : 0) | (d->real_rate2 > 0.0 ? 2 : 0);
    // the tolerance form of the month is the reference's arithmetic while its denominator clamps (max(1e-6, 1 - gf r), :227,
    // :307-310) cannot bind: both effective rates <= 1 - 1e-6 (mcr_device.h).  Otherwise: exact forms, generic variants.
    d->exact_month = (d->real_rate1 > 1.0 - kEps || d->real_rate2 > 1.0 - kEps) ? 1 : 0;
    const double sqrt12 = std::sqrt((double)kMPY);
    d->a1 = p->inv1_mu_log / (double)kMPY;   d->b1 = p->inv1_sigma_log / sqrt12;     // :473
    d->ainf = p->inf_mu_log / (double)kMPY;  d->binf = p->inf_sigma_log / sqrt12;
    d->aprem = p->prem_mu_log / (double)kMPY; d->bprem = p->prem_sigma_log / sqrt12;
    d->rho = p->equity_inflation_rho;
    const double om = 1.0 - d->rho * d->rho;
    d->rho_c = std::sqrt(om > 0.0 ? om : 0.0);  // :463
    d->binf_rho = d->binf * d->rho;
    d->binf_rho_c = d->binf * d->rho_c;
    d->working_months = wm;
    d->retirement_years = p->retirement_years;
    d->total_months = sz.total_months;
    d->shock_rows = sz.shock_rows;
    d->num_working_years = sz.num_working_years;
    d->trajectory_len = sz.trajectory_len;
    d->n_streams = p->n_streams < MCR_INLINE_STREAMS ? p->n_streams : MCR_INLINE_STREAMS;
    d->n_extra_streams = p->n_streams - d->n_streams;
    if (extra) extra->assign((size_t)d->n_extra_streams, DevStream{});
    int slots = 0;
    for (int s = 0; s < p->n_streams; ++s) {
        const mcr_stream& in = stream_at(p, s);
        DevStream scratch;
        DevStream& o = s < MCR_INLINE_STREAMS ? d->streams[s] : (extra ? (*extra)[(size_t)(s - MCR_INLINE_STREAMS)] : scratch);
        o.amount = in.monthly_amount_today;
        o.keep = 1.0 - in.tax_rate;  // :676
        o.amount_keep = o.amount * o.keep;
        o.start_month = start_month_index(p->current_age, wm, in.start_at_age);  // :603-608
        if (in.duration_years < 0) {
            o.end_month = INT32_MAX;  // None: forever (:654)
        } else {
            const int64_t e = (int64_t)o.start_month + (int64_t)in.duration_years * kMPY;  // :609-613,:655
            o.end_month = e > INT32_MAX ? INT32_MAX : (int32_t)e;
        }
        o.indexed = in.inflation_indexed ? 1 : 0;
        o.lock_slot = o.indexed ? -1 : slots++;
    }
    d->n_lock_slots_total = slots;
    d->n_lock_slots = slots;     // (the launcher lowers it to what its kernel variant's LDS budget holds: plan_lock_slots)
    return MCR_OK;
}

// LDS of one path_kernel launch.  STATIC part of the variant — it mirrors the __shared__ declarations at the top of the
// kernel: the math tables, the stage of growth factors (Philox stream without injection; twice for the producer / consumer
// form) and the three per-path summary columns (those variants with per-path outputs) — plus the launch's DYNAMIC part: the
// ziggurat tables (NumPy stream), the block counters / year bins / histogram bins, and as many [kBlock] lock columns of
// non-indexed income streams as keep FOUR workgroups resident on a CU (160 KB of LDS: 40 KB each), never more than the 64 KB
// a workgroup may use without opting in; the remaining slots go to a global overflow block (DevParams::lock_overflow).
// Measured at 10^6 config.json paths (tools/streams_timing.py, profiles/r04): 16 frozen streams with every column in LDS
// (2 resident workgroups) 18.4 ms count-only, all in the overflow block 12.5 ms; 8 frozen streams (4 resident workgroups
// with all columns in LDS) 8.7 vs 8.9 ms.  MCR_K1_LDS_LOCK_SLOTS=n in the environment overrides the occupancy rule: up to n
// slots in LDS (A/B, tests).
constexpr size_t kLdsPerWorkgroup = 64 * 1024;
constexpr size_t kLdsForFourResident = 40 * 1024;
static size_t path_kernel_static_lds(int mode, bool numpy_rng, bool injected, bool split) {
    const bool staged = !numpy_rng && !injected;
    return (size_t)kMathTabBytes + (staged ? (size_t)(split ? 2 : 1) * kStageDoubles * sizeof(double) : 16) +
           ((mode >= 1 && staged) ? (size_t)3 * kBlock * sizeof(double) : 16);
}
static int plan_path_kernel_lds(DevParams& d, int mode, bool numpy_rng, bool injected, bool split, int n_hist_bins, size_t* dynamic_bytes) {
    const size_t fixed = path_kernel_static_lds(mode, numpy_rng, injected, split) + (numpy_rng ? (size_t)kZigLdsBytes : (size_t)0) +
                         (size_t)(1 + (d.retirement_years + 2) + (d.retirement_years + 1) + n_hist_bins) * sizeof(unsigned int);
    if (fixed > kLdsPerWorkgroup) { set_error("too many retirement years / histogram bins for the LDS of a workgroup"); return MCR_ERR_UNSUPPORTED; }
    constexpr size_t kSlotBytes = (size_t)kBlock * sizeof(double);
    long slots = (long)((kLdsPerWorkgroup - fixed) / kSlotBytes);
    const char* e = std::getenv("MCR_K1_LDS_LOCK_SLOTS");
    if (e && *e) slots = std::min(slots, std::max(0l, std::strtol(e, nullptr, 10)));
    else slots = std::min(slots, fixed < kLdsForFourResident ? (long)((kLdsForFourResident - fixed) / kSlotBytes) : 0l);
    d.n_lock_slots = (int32_t)std::min<long>(slots, d.n_lock_slots_total);
    *dynamic_bytes = fixed - path_kernel_static_lds(mode, numpy_rng, injected, split) + (size_t)d.n_lock_slots * kBlock * sizeof(double);
    return MCR_OK;
}

// The device side of a launch's stream list: the table of the records beyond the by-value block and the overflow block of
// lock slots, ONE stream-ordered allocation (freed behind the kernel).  The host copy of the table is owned by the stream
// until the upload has run (hipLaunchHostFunc): nothing here waits for the device.
struct StreamSideBlock {
    void* mem = nullptr;
    int attach(DevParams& d, const std::vector<DevStream>& extra, unsigned grid_x, hipStream_t stream) {
        const int n_over = d.n_lock_slots_total - d.n_lock_slots;
        if (extra.empty() && n_over <= 0) return MCR_OK;
        d.lock_stride = (int64_t)grid_x * kBlock;
        const size_t table_bytes = (extra.size() * sizeof(DevStream) + 255) & ~(size_t)255;
        const size_t over_bytes = (size_t)(n_over > 0 ? n_over : 0) * (size_t)d.lock_stride * sizeof(double);
        hipError_t e = hipMallocAsync(&mem, table_bytes + over_bytes, stream);
        if (e != hipSuccess) { mem = nullptr; return hip_fail(e, "income-stream table / lock-slot overflow allocation"); }
        if (!extra.empty()) {
            void* host = std::malloc(extra.size() * sizeof(DevStream));
            if (!host) { set_error("out of host memory"); return MCR_ERR_HIP; }
            std::memcpy(host, extra.data(), extra.size() * sizeof(DevStream));
            e = hipMemcpyAsync(mem, host, extra.size() * sizeof(DevStream), hipMemcpyHostToDevice, stream);
            const hipError_t ef = hipLaunchHostFunc(stream, [](void* h) { std::free(h); }, host);
            if (ef != hipSuccess) { (void)hipStreamSynchronize(stream); std::free(host); }
            if (e != hipSuccess) return hip_fail(e, "income-stream table upload");
            d.extra_streams = (const DevStream*)mem;
        }
        if (n_over > 0) d.lock_overflow = (double*)((char*)mem + table_bytes);
        return MCR_OK;
    }
    hipError_t release(hipStream_t stream) {
        if (!mem) return hipSuccess;
        const hipError_t e = hipFreeAsync(mem, stream);
        mem = nullptr;
        return e;
    }
};

// Launches of at most this many path-wavefronts take the producer / consumer split (SPLIT = true): up to 3 per SIMD a
// wavefront is latency-bound and the second wave per path hides half of its chain; above it the chip is busy either way
// and the split only adds barriers.  MCR_K1_SPLIT_MAX_WAVES overrides it (0 = never).
static unsigned split_max_waves() {   // (read at every launch: tests compare both forms in one process)
    const char* e = std::getenv("MCR_K1_SPLIT_MAX_WAVES");
    return (e && *e) ? (unsigned)std::strtoul(e, nullptr, 10) : 3072u;
}
// Plan of a time-sliced launch (PHASE 3 of path_kernel): how many path blocks are sliced, into how many segments, at which
// retirement years.  Resident slots = CUs x 6 workgroups (count-only) or x 5 (variants with per-path outputs); the slices are equal in COST
// (an accumulation month is ~0.83 of a retirement month: no withdrawal).
struct SegmentPlan { int n_split, n_full, q, max_polls, year[kMaxSegments + 1]; };
static bool plan_segments(const DevParams& d, unsigned n_blocks, int mode, SegmentPlan* plan, bool retirement_only = false) {
    int q = -1;     // (chosen below from the shape of the launch unless the environment says otherwise)
    if (const char* e = std::getenv("MCR_K1_SEGMENTS")) q = std::atoi(e);
    if (q >= 0 && q < 2) return false;
    static int cus_cached = 0;         // (one device model per process in practice; a wrong figure costs time, not results)
    if (cus_cached == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) { (void)hipGetLastError(); cus = 256; }
        cus_cached = cus;
    }
    const unsigned slots = (unsigned)cus_cached * (mode == 0 ? 6u : 5u);     // resident workgroups: the variants' launch bounds
    if (n_blocks <= slots || d.retirement_years < 4 || d.n_extra_streams > 0 || d.n_lock_slots < d.n_lock_slots_total) return false;
    if (q < 0) q = n_blocks < 2 * slots ? 8 : 6;     // (measured: 500 000 paths 3.12 ms plain, 2.77 with 4 segments, 2.58 with 6; 10^6 and 2 10^6: 6 = 4 - 0.4 %)
    q = std::min(std::min(q, kMaxSegments), d.retirement_years / 2);
    // Worth it where the last round of a plain launch is mostly empty: rounds r = blocks / slots, loss of the plain launch up to
    // ceil(r) / r.  Measured (10^6-path neighbourhood, tools/k1_segments_ab.py): r = 2.54 -3.8 %, 2.29 -7 %, 5.09 -5.6 %, 1.27
    // -11 %; r = 2.0, 3.0 +2 % (nothing to gain, the extra workgroups cost), 2.8 +1 %, 3.81 0.  MCR_K1_SEGMENTS_ALWAYS=1 (tests).
    const double r = (double)n_blocks / (double)slots;
    const char* always = std::getenv("MCR_K1_SEGMENTS_ALWAYS");
    if (!(always && always[0] == '1') && std::ceil(r) / r < 1.08) return false;
    plan->max_polls = 20000;            // x ~1 us; a successor is dispatched long after its predecessor has finished
    if (const char* e = std::getenv("MCR_K1_SEGMENT_POLLS")) plan->max_polls = std::max(0, std::atoi(e));   // (0: every successor recomputes — tests)
    plan->q = q;
    plan->n_split = (int)slots;
    plan->n_full = (int)(n_blocks - slots);
    const double acc = retirement_only ? 0.0 : 0.83 * d.working_months, total = acc + (double)kMPY * d.retirement_years;   // (PHASE 4: the candidates resume at retirement)
    plan->year[0] = 0;
    for (int k = 1; k < q; ++k) {
        int y = (int)std::lround((total * k / q - acc) / kMPY);
        y = std::max(y, plan->year[k - 1] + (k == 1 ? 0 : 1));
        plan->year[k] = std::min(y, d.retirement_years - (q - k));
        if (plan->year[k] < plan->year[k - 1]) return false;
    }
    plan->year[q] = d.retirement_years;
    return true;
}

static int check_rng(const mcr_rng* rng) {
    if (!rng) { set_error("null rng"); return MCR_ERR_INVALID_ARG; }
    if (rng->kind != MCR_RNG_PHILOX && rng->kind != MCR_RNG_NUMPY) { set_error("unknown rng kind %u", rng->kind); return MCR_ERR_INVALID_ARG; }
    if (rng->kind == MCR_RNG_NUMPY && !rng->path_seeds &&
        (rng->n_entropy_words < 1 || rng->n_entropy_words > MCR_MAX_ENTROPY_WORDS)) {
        set_error("numpy rng: main seed must have 1..%d uint32 words (got %u)", MCR_MAX_ENTROPY_WORDS, rng->n_entropy_words);
        return MCR_ERR_INVALID_ARG;
    }
    return MCR_OK;
}

static void fill_io_rng(KernelIO& io, const mcr_rng* rng, const uint32_t* device_path_seeds) {
    io.seed = rng->philox_seed;
    io.n_entropy = rng->n_entropy_words;
    for (int i = 0; i < MCR_MAX_ENTROPY_WORDS; ++i) io.entropy[i] = rng->entropy[i];
    io.child_offset = rng->child_offset;
    io.path_seeds = device_path_seeds;
}

static int launch_paths(const mcr_params* p, const mcr_rng* rng, uint32_t stream_id, uint64_t path_begin,
                        uint64_t n_paths, int32_t wm, const double* injected, const mcr_outputs* out,
                        hipStream_t stream) {
    DevParams d;
    std::vector<DevStream> extra;
    int rc = derive_params(p, wm, &d, &extra);
    if (rc != MCR_OK) return rc;
    rc = check_rng(rng);
    if (rc != MCR_OK) return rc;
    if (!out) { set_error("null outputs"); return MCR_ERR_INVALID_ARG; }
    if (n_paths == 0) return MCR_OK;
    if (n_paths > (uint64_t)INT32_MAX * kBlock) { set_error("n_paths too large for one launch"); return MCR_ERR_INVALID_ARG; }
    if (path_begin + n_paths < path_begin) { set_error("path range overflows 64 bits"); return MCR_ERR_INVALID_ARG; }
    const bool want_traj = out->trajectory || out->real_trajectory || out->withdrawal_rate_trajectory;
    const bool want_summary = out->start_balance || out->final_balance || out->years_to_ruin ||
                              out->first_year_gross_withdrawal || out->first_year_real_gross_withdrawal ||
                              out->inflation_at_retirement || out->success;
    KernelIO io;
    std::memset(&io, 0, sizeof(io));
    fill_io_rng(io, rng, rng->path_seeds);
    io.stream_id = stream_id; io.path_begin = path_begin; io.n_paths = n_paths;
    io.injected = injected; io.out = *out;
    if (io.out.path_stride <= 0) io.out.path_stride = (int64_t)n_paths;
    if (want_traj && (uint64_t)io.out.path_stride < n_paths) {
        set_error("path_stride %lld < n_paths %llu", (long long)io.out.path_stride, (unsigned long long)n_paths);
        return MCR_ERR_INVALID_ARG;
    }
    const bool np_rng = rng->kind == MCR_RNG_NUMPY;
    if (io.out.hist_bins == nullptr || io.out.hist_n_bins == 0) { io.out.hist_bins = nullptr; io.out.hist_edges = nullptr; io.out.hist_n_bins = 0; }
    if (io.out.hist_bins && (io.out.hist_n_bins < 0 || io.out.hist_n_bins > MCR_MAX_HIST_BINS || !io.out.hist_edges)) {
        set_error("hist_bins requested with hist_n_bins = %d (1..%d) / hist_edges = %p", io.out.hist_n_bins, MCR_MAX_HIST_BINS, (const void*)io.out.hist_edges);
        return MCR_ERR_INVALID_ARG;
    }
    const dim3 grid((unsigned)((n_paths + kBlock - 1) / kBlock)), block(kBlock);
    const int mode = injected ? 2 : (want_traj ? 2 : (want_summary ? 1 : 0));
    // kernel variant: output mode x RNG x (any effective realized-gains rate?) x (any annual-gains tax?); injected
    // shocks (parity hook) always take the full-output variant, whose every store is null-checked
    bool split = !injected && !np_rng && mode == 0 && (uint64_t)grid.x * (kBlock / 64) <= split_max_waves();
    // The producer / consumer form counts on both halves of a workgroup executing the same number of barriers: producers run
    // rows [0, total_months), consumers wm accumulation months + 12 months per retirement year.  (It also counts on gfx9's
    // s_barrier not waiting for waves that have ended — producers return while consumers still reduce their counts.)
    if (split && d.total_months != d.working_months + kMPY * d.retirement_years) { set_error("internal: total_months != working_months + 12 retirement_years"); return MCR_ERR_INVALID_ARG; }
    size_t lds = 0;
    rc = plan_path_kernel_lds(d, mode, np_rng, injected != nullptr, split, io.out.hist_n_bins, &lds);
    if (rc != MCR_OK) return rc;
    if (split && (d.n_lock_slots < d.n_lock_slots_total || d.n_extra_streams > 0 || (d.exact_month && !kExactMonthDefault))) {
        // the producer / consumer form doubles the stage: where only IT cannot hold every lock column, the unsplit kernel runs
        split = false;
        rc = plan_path_kernel_lds(d, mode, np_rng, false, false, io.out.hist_n_bins, &lds);
        if (rc != MCR_OK) return rc;
    }
    // XS: records beyond the by-value block and / or lock slots beyond the LDS budget -> the extended-stream variants
    //     (and configurations that need the exact month: the generic variants carry both forms of it)
    const bool exact = d.exact_month && !kExactMonthDefault;
    const bool xs = d.n_lock_slots < d.n_lock_slots_total || d.n_extra_streams > 0 || exact;
    StreamSideBlock side;
    rc = side.attach(d, extra, grid.x, stream);
    if (rc != MCR_OK) { (void)side.release(stream); return rc; }
    if (xs) {
#define MCR_LAUNCH_X(M, R, I) do { if (exact) hipLaunchKernelGGL((path_kernel<M, R, 3, true, I, 0, false, true, true>), grid, block, lds, stream, d, io, (const DevParams*)nullptr); \
                                   else hipLaunchKernelGGL((path_kernel<M, R, 3, true, I, 0, false, true>), grid, block, lds, stream, d, io, (const DevParams*)nullptr); } while (0)
        if (injected) MCR_LAUNCH_X(2, 0, true);
        else if (!np_rng) { if (mode == 2) MCR_LAUNCH_X(2, 0, false); else if (mode == 1) MCR_LAUNCH_X(1, 0, false); else MCR_LAUNCH_X(0, 0, false); }
        else { if (mode == 2) MCR_LAUNCH_X(2, 1, false); else if (mode == 1) MCR_LAUNCH_X(1, 1, false); else MCR_LAUNCH_X(0, 1, false); }
#undef MCR_LAUNCH_X
        hipError_t ex = hipGetLastError();
        const hipError_t ef = side.release(stream);
        if (ex != hipSuccess) return hip_fail(ex, "path_kernel launch (extended streams)");
        if (ef != hipSuccess) return hip_fail(ef, "path_kernel launch (extended streams): side block release");
        return MCR_OK;
    }
    // Time-sliced path blocks (PHASE 3 of path_kernel): Philox launches of a few rounds of resident workgroups whose last round
    // would be mostly empty (plan_segments).  MCR_K1_SEGMENTS = segments per sliced block (default 6 or 8; 0 or 1 = never).
    if (!np_rng && !injected && !xs && !split) {
        SegmentPlan plan;
        if (plan_segments(d, grid.x, mode, &plan)) {
            const size_t state_bytes = (size_t)plan.n_split * (size_t)((mode >= 1 ? 14 : 9) + d.n_lock_slots) * kBlock * sizeof(double);
            const size_t flag_bytes = (size_t)plan.n_split * (size_t)plan.q * sizeof(unsigned int);
            void* mem = nullptr;
            hipError_t e = hipMallocAsync(&mem, state_bytes + flag_bytes, stream);
            if (e == hipSuccess) {
                io.seg_state = (double*)mem;
                io.seg_flags = (unsigned int*)((char*)mem + state_bytes);
                io.seg_n_split = plan.n_split; io.seg_n_full = plan.n_full; io.seg_q = plan.q; io.seg_max_polls = plan.max_polls;
                for (int k = 0; k <= plan.q; ++k) io.seg_year[k] = plan.year[k];
                e = hipMemsetAsync(io.seg_flags, 0, flag_bytes, stream);
                const dim3 gseg((unsigned)(plan.n_full + plan.q * plan.n_split));
#define MCR_LAUNCH_GM(M, T, A) hipLaunchKernelGGL((path_kernel<M, 0, T, A, false, 3>), gseg, block, lds, stream, d, io, (const DevParams*)nullptr)
#define MCR_LAUNCH_G(T, A) do { if (mode == 2) MCR_LAUNCH_GM(2, T, A); else if (mode == 1) MCR_LAUNCH_GM(1, T, A); else MCR_LAUNCH_GM(0, T, A); } while (0)
#define MCR_LAUNCH_GA(T) do { if (d.any_annual_tax) MCR_LAUNCH_G(T, true); else MCR_LAUNCH_G(T, false); } while (0)
                if (e == hipSuccess) {
                    switch (d.tax_mask) { case 0: MCR_LAUNCH_GA(0); break; case 1: MCR_LAUNCH_GA(1); break; case 2: MCR_LAUNCH_GA(2); break; default: MCR_LAUNCH_GA(3); break; }
                    e = hipGetLastError();
                }
#undef MCR_LAUNCH_GA
#undef MCR_LAUNCH_G
#undef MCR_LAUNCH_GM
                const hipError_t ef = hipFreeAsync(mem, stream);
                if (e != hipSuccess) return hip_fail(e, "path_kernel launch (time-sliced blocks)");
                if (ef != hipSuccess) return hip_fail(ef, "path_kernel launch (time-sliced blocks): state release");
                return MCR_OK;
            }
            (void)hipGetLastError();   // (allocation refused: the plain launch below)
        }
    }
    if (split) {
        const dim3 block2(2 * kBlock);
#define MCR_LAUNCH_S(T, A) hipLaunchKernelGGL((path_kernel<0, 0, T, A, false, 0, true>), grid, block2, lds, stream, d, io, (const DevParams*)nullptr)
#define MCR_LAUNCH_SA(T) do { if (d.any_annual_tax) MCR_LAUNCH_S(T, true); else MCR_LAUNCH_S(T, false); } while (0)
        switch (d.tax_mask) { case 0: MCR_LAUNCH_SA(0); break; case 1: MCR_LAUNCH_SA(1); break; case 2: MCR_LAUNCH_SA(2); break; default: MCR_LAUNCH_SA(3); break; }
#undef MCR_LAUNCH_SA
#undef MCR_LAUNCH_S
        hipError_t es = hipGetLastError();
        if (es != hipSuccess) return hip_fail(es, "path_kernel launch (split)");
        return MCR_OK;
    }
#define MCR_LAUNCH(M, R, T, A, I) hipLaunchKernelGGL((path_kernel<M, R, T, A, I>), grid, block, lds, stream, d, io, (const DevParams*)nullptr)
    // the engine's own stream: one variant per tax mask (which of the two assets is taxed on realized gains); the parity
    // hook and the NumPy stream: taxed / untaxed only (mask 3 computes a zero-rate asset's tax arithmetic as exact zeros)
#define MCR_LAUNCH_A(M, R, T, I) do { if (d.any_annual_tax) MCR_LAUNCH(M, R, T, true, I); else MCR_LAUNCH(M, R, T, false, I); } while (0)
#define MCR_LAUNCH_T(M, R, I)                                                                      \
    do {                                                                                           \
        if (R == 0 && !I) {                                                                        \
            switch (d.tax_mask) { case 0: MCR_LAUNCH_A(M, 0, 0, false); break; case 1: MCR_LAUNCH_A(M, 0, 1, false); break; \
                                  case 2: MCR_LAUNCH_A(M, 0, 2, false); break; default: MCR_LAUNCH_A(M, 0, 3, false); break; } \
        } else if (d.any_real_rate) MCR_LAUNCH_A(M, R, 3, I);                                       \
        else MCR_LAUNCH_A(M, R, 0, I);                                                              \
    } while (0)
    if (injected) {
        MCR_LAUNCH_T(2, 0, true);
    } else if (!np_rng) {
        if (mode == 2) MCR_LAUNCH_T(2, 0, false);
        else if (mode == 1) MCR_LAUNCH_T(1, 0, false);
        else MCR_LAUNCH_T(0, 0, false);
    } else {
        if (mode == 2) MCR_LAUNCH_T(2, 1, false);
        else if (mode == 1) MCR_LAUNCH_T(1, 1, false);
        else MCR_LAUNCH_T(0, 1, false);
    }
#undef MCR_LAUNCH_T
#undef MCR_LAUNCH_A
#undef MCR_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "path_kernel launch");
    return MCR_OK;
}

}  // namespace mcr

// =============================================================================================
// C ABI
// =============================================================================================
using namespace mcr;

extern "C" {

int mcr_abi_version(void) { return MCR_ABI_VERSION; }

int mcr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

const char* mcr_last_error(void) { return g_err; }

int mcr_query_sizes(const mcr_params* p, int32_t working_months, mcr_sizes* out) {
    return query_sizes(p, working_months, out);
}

int32_t mcr_stream_start_month_index(double current_age, int32_t working_months, double start_at_age) {
    return start_month_index(current_age, working_months, start_at_age);
}

static mcr_rng philox_rng(uint64_t seed) {
    mcr_rng r;
    std::memset(&r, 0, sizeof(r));
    r.kind = MCR_RNG_PHILOX;
    r.philox_seed = seed;
    return r;
}

int mcr_run_batch_rng(const mcr_params* p, const mcr_rng* rng, uint32_t stream_id, uint64_t path_begin,
                      uint64_t n_paths, int32_t working_months, const double* injected_shocks,
                      const mcr_outputs* out, int device, void* hip_stream) {
    MCR_ENTER_DEVICE(device);
    return launch_paths(p, rng, stream_id, path_begin, n_paths, working_months, injected_shocks, out,
                        (hipStream_t)hip_stream);
}

// Several candidates over the same paths, Philox stream: ONE accumulation sweep to the largest candidate that stores the
// state at the end of every candidate month (PHASE 1), then ONE launch whose grid.y is the candidate and which resumes
// every decumulation from its snapshot (PHASE 2).  The candidates' parameter blocks (stream start months, horizon) go to
// device memory; snapshots and blocks are stream-ordered allocations.  Counts are identical to one launch per candidate.
static int probe_shared_prefix(const mcr_params* p, const mcr_rng* rng, uint32_t stream_id, uint64_t path_begin, uint64_t n_paths,
                               const int32_t* working_months, int32_t n_cand, uint64_t* counts, hipStream_t stream) {
    if (n_cand < 2 || n_paths == 0 || n_paths > ((uint64_t)1 << 31)) return MCR_ERR_UNSUPPORTED;
    int order[MCR_MAX_PROBE_CANDIDATES];
    for (int i = 0; i < n_cand; ++i) {
        int k = i;
        while (k > 0 && working_months[order[k - 1]] > working_months[i]) { order[k] = order[k - 1]; --k; }
        order[k] = i;
    }
    for (int i = 1; i < n_cand; ++i)
        if (working_months[order[i]] == working_months[order[i - 1]]) return MCR_ERR_UNSUPPORTED;   // duplicates: plain route
    std::vector<DevParams> blocks((size_t)n_cand);
    for (int i = 0; i < n_cand; ++i) {
        const int rc = derive_params(p, working_months[order[i]], &blocks[(size_t)i]);
        if (rc != MCR_OK) return rc;
    }
    DevParams& top = blocks[(size_t)n_cand - 1];
    // (stream lists beyond the by-value block, or lock slots beyond the LDS of the split form, take one launch per candidate:
    //  each then carries its own device table / overflow block)
    if (top.n_extra_streams > 0 || (top.exact_month && !kExactMonthDefault)) return MCR_ERR_UNSUPPORTED;
    size_t lds = 0;
    if (plan_path_kernel_lds(top, 0, false, false, true, 0, &lds) != MCR_OK || top.n_lock_slots < top.n_lock_slots_total) return MCR_ERR_UNSUPPORTED;
    for (DevParams& b : blocks) b.n_lock_slots = top.n_lock_slots;
    KernelIO io;
    std::memset(&io, 0, sizeof(io));
    fill_io_rng(io, rng, nullptr);
    io.stream_id = stream_id; io.path_begin = path_begin; io.n_paths = n_paths;
    io.out.counters = counts;
    io.n_snap = n_cand;
    io.snap_stride = (int64_t)((n_paths + 63) / 64 * 64);
    for (int i = 0; i < n_cand; ++i) { io.snap_months[i] = working_months[order[i]]; io.cand_out[i] = order[i]; }
    const size_t snap_bytes = (size_t)n_cand * kSnapFields * (size_t)io.snap_stride * sizeof(double);
    const size_t blocks_bytes = (size_t)n_cand * sizeof(DevParams);
    if (snap_bytes > ((size_t)4 << 30)) return MCR_ERR_UNSUPPORTED;   // huge probes are throughput-bound anyway: plain route
    void* mem = nullptr;
    if (hipMallocAsync(&mem, snap_bytes + blocks_bytes, stream) != hipSuccess) { (void)hipGetLastError(); return MCR_ERR_UNSUPPORTED; }
    io.snap = (double*)mem;
    const DevParams* d_blocks = (const DevParams*)((char*)mem + snap_bytes);
    hipError_t e = hipMemcpyAsync((char*)mem + snap_bytes, blocks.data(), blocks_bytes, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) {
        const dim3 block(kBlock), g1((unsigned)((n_paths + kBlock - 1) / kBlock)), g2(g1.x, (unsigned)n_cand);
        // (either phase takes the producer / consumer split on its own while its launch leaves SIMDs idle)
        if (top.total_months != top.working_months + kMPY * top.retirement_years) { (void)hipFreeAsync(mem, stream); set_error("internal: total_months != working_months + 12 retirement_years"); return MCR_ERR_INVALID_ARG; }   // (the split forms' barrier counts, see launch_paths)
        const bool split1 = (uint64_t)g1.x * (kBlock / 64) <= split_max_waves();
        const bool split2 = (uint64_t)g2.x * g2.y * (kBlock / 64) <= split_max_waves();
        const dim3 block2(2 * kBlock);
        // the candidates' decumulations time-sliced (PHASE 4) where their workgroups are a few rounds of the resident slots
        // with a mostly empty last one: the search's 17-month verification window at 50 000 paths is 3 332 workgroups = 2.17 rounds
        SegmentPlan plan;
        void* seg_mem = nullptr;
        dim3 g4(0);
        if (!split2 && plan_segments(top, g2.x * g2.y, 0, &plan, true)) {
            const size_t state_bytes = (size_t)plan.n_split * (size_t)(9 + top.n_lock_slots) * kBlock * sizeof(double);
            const size_t flag_bytes = (size_t)plan.n_split * (size_t)plan.q * sizeof(unsigned int);
            if (hipMallocAsync(&seg_mem, state_bytes + flag_bytes, stream) == hipSuccess &&
                hipMemsetAsync((char*)seg_mem + state_bytes, 0, flag_bytes, stream) == hipSuccess) {
                io.seg_state = (double*)seg_mem;
                io.seg_flags = (unsigned int*)((char*)seg_mem + state_bytes);
                io.seg_n_split = plan.n_split; io.seg_n_full = plan.n_full; io.seg_q = plan.q; io.seg_max_polls = plan.max_polls;
                io.seg_blocks_per_cand = (int32_t)g2.x;
                for (int k = 0; k <= plan.q; ++k) io.seg_year[k] = plan.year[k];
                g4 = dim3((unsigned)(plan.n_full + plan.q * plan.n_split));
            } else {
                (void)hipGetLastError();
                if (seg_mem) { (void)hipFreeAsync(seg_mem, stream); seg_mem = nullptr; }
            }
        }
#define MCR_PHASES(T, A)                                                                                          \
        do {                                                                                                       \
            if (split1) hipLaunchKernelGGL((path_kernel<0, 0, T, A, false, 1, true>), g1, block2, lds, stream, top, io, d_blocks); \
            else hipLaunchKernelGGL((path_kernel<0, 0, T, A, false, 1>), g1, block, lds, stream, top, io, d_blocks);    \
            if (split2) hipLaunchKernelGGL((path_kernel<0, 0, T, A, false, 2, true>), g2, block2, lds, stream, top, io, d_blocks); \
            else if (g4.x) hipLaunchKernelGGL((path_kernel<0, 0, T, A, false, 4>), g4, block, lds, stream, top, io, d_blocks);    \
            else hipLaunchKernelGGL((path_kernel<0, 0, T, A, false, 2>), g2, block, lds, stream, top, io, d_blocks);    \
        } while (0)
#define MCR_PHASES_A(T) do { if (top.any_annual_tax) MCR_PHASES(T, true); else MCR_PHASES(T, false); } while (0)
        switch (top.tax_mask) { case 0: MCR_PHASES_A(0); break; case 1: MCR_PHASES_A(1); break; case 2: MCR_PHASES_A(2); break; default: MCR_PHASES_A(3); break; }
#undef MCR_PHASES_A
#undef MCR_PHASES
        e = hipGetLastError();
        if (seg_mem) (void)hipFreeAsync(seg_mem, stream);
    }
    const hipError_t ef = hipFreeAsync(mem, stream);
    if (e != hipSuccess) return hip_fail(e, "shared-prefix probe");
    if (ef != hipSuccess) return hip_fail(ef, "shared-prefix probe (free)");
    return MCR_OK;
}

int mcr_probe_months_rng(const mcr_params* p, const mcr_rng* rng, uint32_t stream_id, uint64_t path_begin,
                         uint64_t n_paths, const int32_t* working_months, int32_t n_candidates,
                         uint64_t* counts, int device, void* hip_stream) {
    MCR_ENTER_DEVICE(device);
    int rc = MCR_OK;
    if (!working_months || !counts || n_candidates < 0) { set_error("null candidates / counts"); return MCR_ERR_INVALID_ARG; }
    if (n_candidates == 0) return MCR_OK;
    // validate every candidate BEFORE enqueueing anything, so a bad one leaves no half-forked work behind
    for (int32_t c = 0; c < n_candidates; ++c) {
        DevParams d;
        rc = derive_params(p, working_months[c], &d);
        if (rc != MCR_OK) return rc;
    }
    rc = check_rng(rng);
    if (rc != MCR_OK) return rc;
    hipStream_t main = (hipStream_t)hip_stream;
    hipError_t e = hipMemsetAsync(counts, 0, sizeof(uint64_t) * MCR_N_COUNTERS * (size_t)n_candidates, main);
    if (e != hipSuccess) return hip_fail(e, "probe counters memset");
    if (n_candidates == 1) {
        mcr_outputs o;
        std::memset(&o, 0, sizeof(o));
        o.counters = counts;
        return launch_paths(p, rng, stream_id, path_begin, n_paths, working_months[0], nullptr, &o, main);
    }
    if (rng->kind == MCR_RNG_PHILOX && n_candidates <= MCR_MAX_PROBE_CANDIDATES) {
        const int prc = probe_shared_prefix(p, rng, stream_id, path_begin, n_paths, working_months, n_candidates, counts, main);
        if (prc != MCR_ERR_UNSUPPORTED) return prc;     // (unsupported shape / allocation refused: one launch per candidate below)
    }
    StreamForkLease fork_lease(device);
    StreamFork* f = fork_lease.f;
    if (!f) { set_error("could not create probe streams"); return MCR_ERR_HIP; }
    const int used = n_candidates < kForkStreams ? n_candidates : kForkStreams;
    if ((e = hipEventRecord(f->fork, main)) != hipSuccess) return hip_fail(e, "probe fork");
    for (int i = 0; i < used; ++i)
        if ((e = hipStreamWaitEvent(f->side[i], f->fork, 0)) != hipSuccess) return hip_fail(e, "probe fork wait");
    int first_rc = MCR_OK;
    for (int32_t c = 0; c < n_candidates && first_rc == MCR_OK; ++c) {
        mcr_outputs o;
        std::memset(&o, 0, sizeof(o));
        o.counters = counts + (size_t)c * MCR_N_COUNTERS;
        first_rc = launch_paths(p, rng, stream_id, path_begin, n_paths, working_months[c], nullptr, &o, f->side[c % used]);
    }
    // always join, also after a failed launch: `main` must not run ahead of work already forked
    for (int i = 0; i < used; ++i) {
        if ((e = hipEventRecord(f->done[i], f->side[i])) != hipSuccess) return hip_fail(e, "probe join record");
        if ((e = hipStreamWaitEvent(main, f->done[i], 0)) != hipSuccess) return hip_fail(e, "probe join wait");
    }
    return first_rc;
}

int mcr_run_batch(const mcr_params* p, uint64_t seed, uint32_t stream_id, uint64_t path_begin,
                  uint64_t n_paths, int32_t working_months, const double* injected_shocks,
                  const mcr_outputs* out, int device, void* hip_stream) {
    const mcr_rng r = philox_rng(seed);
    return mcr_run_batch_rng(p, &r, stream_id, path_begin, n_paths, working_months, injected_shocks, out, device, hip_stream);
}

int mcr_run_batch_host(const mcr_params* p, uint64_t seed, uint32_t stream_id, uint64_t path_begin,
                       uint64_t n_paths, int32_t working_months, const double* injected_shocks,
                       const mcr_outputs* out, int device) {
    const mcr_rng r = philox_rng(seed);
    return mcr_run_batch_host_rng(p, &r, stream_id, path_begin, n_paths, working_months, injected_shocks, out, device);
}

// One host-buffer batch on `device`: device buffers carved from the thread's cached block, uploads / kernel /
// downloads on the thread's private stream, ONE stream synchronisation at the end.
static int run_batch_host_on(int device, const mcr_params* p, const mcr_rng* rng_in, uint32_t stream_id, uint64_t path_begin,
                             uint64_t n_paths, int32_t working_months, const double* injected_shocks, const mcr_outputs* out) {
    MCR_ENTER_DEVICE(device);
    int rc = check_rng(rng_in);
    if (rc != MCR_OK) return rc;
    mcr_sizes sz;
    rc = query_sizes(p, working_months, &sz);
    if (rc != MCR_OK) return rc;
    if (!out) { set_error("null outputs"); return MCR_ERR_INVALID_ARG; }
    if (n_paths == 0) return MCR_OK;
    const int64_t hstride = out->path_stride > 0 ? out->path_stride : (int64_t)n_paths;
    if ((uint64_t)hstride < n_paths) { set_error("path_stride < n_paths"); return MCR_ERR_INVALID_ARG; }
    const size_t n = (size_t)n_paths;
    mcr_outputs d = {};
    d.path_stride = (int64_t)((n + 63) / 64 * 64);  // device rows padded to whole wavefronts
    // plan: every device buffer is a slice of one block (256-byte aligned)
    struct Buf { void** dev; void* host; size_t rows, row_bytes, dev_pitch, host_pitch, offset; bool upload, download; };
    std::vector<Buf> bufs;
    size_t total = 0;
    auto plan = [&](void** dev, const void* host, size_t rows, size_t row_bytes, size_t dev_pitch, size_t host_pitch, bool up, bool down) {
        if (!host) return;
        bufs.push_back({dev, const_cast<void*>(host), rows, row_bytes, dev_pitch, host_pitch, total, up, down});
        total += ((rows == 1 ? row_bytes : rows * dev_pitch) + 255) & ~(size_t)255;
    };
    auto vec = [&](double* host, double** dev) { plan((void**)dev, host, 1, n * sizeof(double), 0, 0, false, true); };
    auto mat = [&](double* host, double** dev, int rows) {
        plan((void**)dev, host, (size_t)rows, n * sizeof(double), (size_t)d.path_stride * sizeof(double), (size_t)hstride * sizeof(double), false, true);
    };
    vec(out->start_balance, &d.start_balance);
    vec(out->final_balance, &d.final_balance);
    vec(out->years_to_ruin, &d.years_to_ruin);
    vec(out->first_year_gross_withdrawal, &d.first_year_gross_withdrawal);
    vec(out->first_year_real_gross_withdrawal, &d.first_year_real_gross_withdrawal);
    vec(out->inflation_at_retirement, &d.inflation_at_retirement);
    plan((void**)&d.success, out->success, 1, n, 0, 0, false, true);
    mat(out->trajectory, &d.trajectory, sz.trajectory_len);
    mat(out->real_trajectory, &d.real_trajectory, sz.trajectory_len);
    mat(out->withdrawal_rate_trajectory, &d.withdrawal_rate_trajectory, sz.retirement_years);
    // accumulated counters: the device copy starts from the caller's current values
    plan((void**)&d.counters, out->counters, 1, MCR_N_COUNTERS * sizeof(uint64_t), 0, 0, true, true);
    plan((void**)&d.wr_obs_counts, out->wr_obs_counts, 1, (size_t)sz.retirement_years * sizeof(uint64_t), 0, 0, true, true);
    plan((void**)&d.ruin_year_bins, out->ruin_year_bins, 1, (size_t)sz.ruin_bins * sizeof(uint64_t), 0, 0, true, true);
    if (out->hist_bins && out->hist_n_bins != 0) {   // in-kernel final-balance histogram on the caller's edges
        const int nb = out->hist_n_bins;
        if (nb < 0 || nb > MCR_MAX_HIST_BINS || !out->hist_edges) { set_error("hist_bins requested with hist_n_bins = %d (1..%d) / null hist_edges", nb, MCR_MAX_HIST_BINS); return MCR_ERR_INVALID_ARG; }
        for (int k = 0; k <= nb; ++k)   // host pointer: the edges can be checked here (np.histogram: "bins must increase monotonically")
            if (!std::isfinite(out->hist_edges[k]) || (k > 0 && out->hist_edges[k] < out->hist_edges[k - 1])) {
                set_error("hist_edges[%d] = %g: edges must be finite and ascending", k, out->hist_edges[k]);
                return MCR_ERR_INVALID_ARG;
            }
        d.hist_n_bins = nb;
        plan((void**)&d.hist_edges, out->hist_edges, 1, (size_t)(nb + 1) * sizeof(double), 0, 0, true, false);
        plan((void**)&d.hist_bins, out->hist_bins, 1, (size_t)nb * sizeof(uint64_t), 0, 0, true, true);
    }
    double* d_inj = nullptr;
    plan((void**)&d_inj, injected_shocks, 1, n * (size_t)sz.shock_rows * 3 * sizeof(double), 0, 0, true, false);
    mcr_rng rng = *rng_in;
    uint32_t* d_seeds = nullptr;   // explicit per-path seeds: upload
    plan((void**)&d_seeds, rng.path_seeds, 1, n * sizeof(uint32_t), 0, 0, true, false);

    HostCtxLease lease(device);
    HostCtx* ctx = lease.ctx;
    if (!ctx) return MCR_ERR_HIP;
    hipError_t e = host_ctx_reserve(ctx, total);
    if (e != hipSuccess) return hip_fail(e, "device allocation (host-buffer batch)");
    for (Buf& b : bufs) {
        *b.dev = (char*)ctx->block + b.offset;
        if (b.upload && e == hipSuccess) e = hipMemcpyAsync(*b.dev, b.host, b.row_bytes, hipMemcpyHostToDevice, ctx->stream);
    }
    if (e != hipSuccess) { (void)hipStreamSynchronize(ctx->stream); return hip_fail(e, "upload"); }
    if (rng.path_seeds) rng.path_seeds = d_seeds;
    rc = launch_paths(p, &rng, stream_id, path_begin, n_paths, working_months, d_inj, &d, ctx->stream);
    if (rc == MCR_OK) {
        for (const Buf& b : bufs) {
            if (!b.download || e != hipSuccess) continue;
            if (b.rows == 1) e = hipMemcpyAsync(b.host, *b.dev, b.row_bytes, hipMemcpyDeviceToHost, ctx->stream);
            else e = hipMemcpy2DAsync(b.host, b.host_pitch, *b.dev, b.dev_pitch, b.row_bytes, b.rows, hipMemcpyDeviceToHost, ctx->stream);
        }
    }
    const hipError_t es = hipStreamSynchronize(ctx->stream);   // this call's stream only
    if (rc != MCR_OK) return rc;
    if (e != hipSuccess) return hip_fail(e, "download");
    if (es != hipSuccess) return hip_fail(es, "path_kernel execution");
    return MCR_OK;
}

// The same batch sharded over several devices: contiguous global path ranges, one host thread per device
// (each with its own stream and scratch), counter / bin vectors summed on the host.  No collective is needed:
// the exchange step of the path is < 2 KB and lands in host memory anyway.
static int run_batch_host_multi(const int32_t* devices, int32_t n_devices, const mcr_params* p, const mcr_rng* rng_in,
                                uint32_t stream_id, uint64_t path_begin, uint64_t n_paths, int32_t working_months,
                                const double* injected_shocks, const mcr_outputs* out) {
    if (!out || !rng_in) { set_error("null outputs / rng"); return MCR_ERR_INVALID_ARG; }
    mcr_sizes sz;
    int rc = query_sizes(p, working_months, &sz);
    if (rc != MCR_OK) return rc;
    std::vector<int> devs;
    if (!devices || n_devices <= 0) {
        const int n = mcr_device_count();
        if (n <= 0) { set_error("no usable HIP device (the engine has no CPU fallback)"); return MCR_ERR_NO_DEVICE; }
        for (int i = 0; i < n; ++i) devs.push_back(i);
    } else {
        devs.assign(devices, devices + n_devices);
    }
    const size_t W = devs.size();
    if (W == 1) return run_batch_host_on(devs[0], p, rng_in, stream_id, path_begin, n_paths, working_months, injected_shocks, out);
    const int64_t hstride = out->path_stride > 0 ? out->path_stride : (int64_t)n_paths;
    const uint64_t per = (n_paths + W - 1) / W;
    struct Shard {
        int rc = MCR_OK;
        char err[512] = "";
        std::vector<uint64_t> counters, wr, ruin, hist;
    };
    std::vector<Shard> shards(W);
    std::vector<std::thread> threads;
    for (size_t w = 0; w < W; ++w) {
        const uint64_t begin = std::min<uint64_t>(w * per, n_paths);
        const uint64_t count = std::min<uint64_t>(per, n_paths - begin);
        if (count == 0) continue;
        Shard& S = shards[w];
        S.counters.assign(MCR_N_COUNTERS, 0);
        S.wr.assign((size_t)sz.retirement_years, 0);
        S.ruin.assign((size_t)sz.ruin_bins, 0);
        if (out->hist_bins && out->hist_n_bins > 0) S.hist.assign((size_t)out->hist_n_bins, 0);
        threads.emplace_back([&, w, begin, count]() {
            Shard& T = shards[w];
            mcr_outputs o = *out;    // host pointers of this shard's columns
            auto shift = [&](double*& ptr) { if (ptr) ptr += begin; };
            shift(o.start_balance); shift(o.final_balance); shift(o.years_to_ruin);
            shift(o.first_year_gross_withdrawal); shift(o.first_year_real_gross_withdrawal); shift(o.inflation_at_retirement);
            if (o.success) o.success += begin;
            shift(o.trajectory); shift(o.real_trajectory); shift(o.withdrawal_rate_trajectory);
            o.path_stride = hstride;
            o.counters = out->counters ? T.counters.data() : nullptr;
            o.wr_obs_counts = out->wr_obs_counts ? T.wr.data() : nullptr;
            o.ruin_year_bins = out->ruin_year_bins ? T.ruin.data() : nullptr;
            o.hist_bins = T.hist.empty() ? nullptr : T.hist.data();
            mcr_rng r = *rng_in;
            if (r.path_seeds) r.path_seeds += begin;
            const double* inj = injected_shocks ? injected_shocks + (size_t)begin * (size_t)sz.shock_rows * 3u : nullptr;
            T.rc = run_batch_host_on(devs[w], p, &r, stream_id, path_begin + begin, count, working_months, inj, &o);
            if (T.rc != MCR_OK) std::snprintf(T.err, sizeof(T.err), "device %d: %s", devs[w], mcr_last_error());
        });
    }
    for (std::thread& t : threads) t.join();
    for (const Shard& S : shards)
        if (S.rc != MCR_OK) { set_error("%s", S.err); return S.rc; }
    for (const Shard& S : shards) {
        if (S.counters.empty()) continue;
        if (out->counters) for (int k = 0; k < MCR_N_COUNTERS; ++k) out->counters[k] += S.counters[k];
        if (out->wr_obs_counts) for (int k = 0; k < sz.retirement_years; ++k) out->wr_obs_counts[k] += S.wr[k];
        if (out->ruin_year_bins) for (int k = 0; k < sz.ruin_bins; ++k) out->ruin_year_bins[k] += S.ruin[k];
        for (size_t k = 0; k < S.hist.size(); ++k) out->hist_bins[k] += S.hist[k];
    }
    return MCR_OK;
}

int mcr_run_batch_host_rng(const mcr_params* p, const mcr_rng* rng_in, uint32_t stream_id, uint64_t path_begin,
                           uint64_t n_paths, int32_t working_months, const double* injected_shocks,
                           const mcr_outputs* out, int device) {
    if (device == MCR_DEVICE_ALL)
        return run_batch_host_multi(nullptr, 0, p, rng_in, stream_id, path_begin, n_paths, working_months, injected_shocks, out);
    return run_batch_host_on(device, p, rng_in, stream_id, path_begin, n_paths, working_months, injected_shocks, out);
}

int mcr_run_batch_multi_host_rng(const mcr_params* p, const mcr_rng* rng, uint32_t stream_id, uint64_t path_begin,
                                 uint64_t n_paths, int32_t working_months, const double* injected_shocks,
                                 const mcr_outputs* out, const int32_t* devices, int32_t n_devices) {
    return run_batch_host_multi(devices, n_devices, p, rng, stream_id, path_begin, n_paths, working_months, injected_shocks, out);
}

int mcr_validate_params(const mcr_params* p) { return validate_params(p); }

// numpy.random.RandomState(seed).choice(n, k, replace=False) without shuffling an n-element array (include/mcr.h).
// MT19937 as NumPy's legacy generator runs it: init_genrand seeding (numpy/random/src/mt19937/mt19937.c: mt19937_seed),
// tempered 32-bit outputs, bounded draws by masked rejection (legacy-distributions.c: legacy_random_interval ->
// random_interval), Fisher-Yates from i = n - 1 down to 1 (_mt19937 / mtrand.pyx: _shuffle_raw).
// Host-only and on the critical path of the class API at large n (10^7 paths: 81 ms in round 3's form, twice the path
// kernel): the generator refills 624 tempered words at a time (plain loops the compiler vectorises), the rejection loop is
// branch-free (a draw is written to J[i] either way, i moves on only when it is accepted: the ~30 % of rejected draws were
// mispredicted branches), and the trace-back keeps its <= 8 tracked positions in one vector register (one compare pair and
// a test per step, AVX2 when the CPU has it).  10^7 paths: 135 -> 44 ms on the build container's Xeon.
extern "C++" {
namespace {
struct Mt19937Block {
    uint32_t mt[624];
    uint32_t out[624];   // the tempered outputs of the current block
    void seed(uint32_t s) {
        mt[0] = s;
        for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    }
    __attribute__((always_inline)) void refill() {      // (inlined into the dispatched cores below: AVX2 code where the CPU has it)
        constexpr uint32_t kUpper = 0x80000000u, kLower = 0x7fffffffu, kMatrix = 0x9908b0dfu;
        uint32_t* m = mt;
        for (int i = 0; i < 227; ++i) { const uint32_t y = (m[i] & kUpper) | (m[i + 1] & kLower); m[i] = m[i + 397] ^ (y >> 1) ^ ((0u - (y & 1u)) & kMatrix); }
        for (int i = 227; i < 623; ++i) { const uint32_t y = (m[i] & kUpper) | (m[i + 1] & kLower); m[i] = m[i - 227] ^ (y >> 1) ^ ((0u - (y & 1u)) & kMatrix); }
        { const uint32_t y = (m[623] & kUpper) | (m[0] & kLower); m[623] = m[396] ^ (y >> 1) ^ ((0u - (y & 1u)) & kMatrix); }
        for (int i = 0; i < 624; ++i) { uint32_t y = m[i]; y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18; out[i] = y; }
    }
};
// the value that ends at position p started at the position found by undoing the swaps, last one first: J[i] = the partner
// drawn at step i.  Eight tracked positions in one vector (clang vector extensions: AVX2 under the target attribute below,
// two SSE2 halves otherwise).
typedef uint32_t u32x8 __attribute__((vector_size(32)));
typedef int32_t i32x8 __attribute__((vector_size(32)));
template <int>
__attribute__((always_inline)) inline void trace_back8(const uint32_t* J, uint64_t n, uint32_t* where8) {
    u32x8 w;
    std::memcpy(&w, where8, sizeof(w));
    for (uint64_t t = 1; t < n; ++t) {
        const uint32_t j = J[t], ii = (uint32_t)t;
        const u32x8 vi = {ii, ii, ii, ii, ii, ii, ii, ii}, vj = {j, j, j, j, j, j, j, j};
        const i32x8 ei = (i32x8)(w == vi), ej = (i32x8)(w == vj);
        const i32x8 any = ei | ej;
        if (__builtin_reduce_or(any)) w = (u32x8)((ei & (i32x8)vj) | (ej & (i32x8)vi) | (~any & (i32x8)w));
    }
    std::memcpy(where8, &w, sizeof(w));
}
// the n - 1 bounded draws of the shuffle into J, then the trace-back of positions 0 .. k - 1 (where[64], identity on entry)
template <int>
__attribute__((always_inline)) inline void sample_core(Mt19937Block* g, uint32_t* J, uint64_t n, int k, uint32_t* where) {
    int pos = 624;
    uint64_t i = n - 1;                                  // (i <= 2^32 - 1: the 32-bit branch of random_interval)
    while (i >= 1) {
        uint64_t mask = i;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        const uint32_t lo = (uint32_t)((mask >> 1) + 1);    // the smallest i with this mask
        const uint32_t m32 = (uint32_t)mask;
        uint32_t ii = (uint32_t)i;
        while (ii >= lo) {
            if (pos == 624) { g->refill(); pos = 0; }
            int q = pos;
            for (; q < 624 && ii >= lo; ++q) { const uint32_t v = g->out[q] & m32; J[ii] = v; ii -= (v <= ii) ? 1u : 0u; }   // masked rejection, branch-free
            pos = q;
        }
        i = ii;
    }
    if (k <= 8) {
        trace_back8<0>(J, n, where);
    } else {
        for (uint64_t t = 1; t < n; ++t) {
            const uint32_t j = J[t], ii = (uint32_t)t;
            for (int p = 0; p < k; ++p) {
                const uint32_t w = where[p];
                where[p] = w == ii ? j : (w == j ? ii : w);
            }
        }
    }
}
__attribute__((target("avx2"))) void sample_core_avx2(Mt19937Block* g, uint32_t* J, uint64_t n, int k, uint32_t* w) { sample_core<1>(g, J, n, k, w); }
void sample_core_generic(Mt19937Block* g, uint32_t* J, uint64_t n, int k, uint32_t* w) { sample_core<0>(g, J, n, k, w); }
}  // namespace
}  // extern "C++"

int mcr_sample_columns(uint32_t seed, uint64_t n, int32_t k, int64_t* out) {
    if (!out || k < 1 || k > 64 || (uint64_t)k > n || n > ((uint64_t)1 << 32)) { set_error("mcr_sample_columns: need 1 <= k <= min(n, 64), n <= 2^32"); return MCR_ERR_INVALID_ARG; }
    if (n == 1) { out[0] = 0; return MCR_OK; }
    if (n > 50000000ull) {   // once per process: the draw is sequential host work, O(n) time and 4 n bytes of host scratch
        static std::once_flag once;
        std::call_once(once, [n] { std::fprintf(stderr, "mcr_sample_columns: %llu paths: the sampled-column draw (NumPy's RandomState.choice, MT19937, "
                                                        "sequential) takes ~5 ns and 4 bytes of host scratch per path\n", (unsigned long long)n); });
    }
    // J[i] = the partner position drawn at step i (i = n - 1 ... 1); J[0] unused
    uint32_t* J = (uint32_t*)std::malloc((size_t)n * sizeof(uint32_t));
    if (!J) { set_error("mcr_sample_columns: out of host memory (%llu bytes)", (unsigned long long)(n * 4)); return MCR_ERR_INVALID_ARG; }
    Mt19937Block* g = new Mt19937Block;
    g->seed(seed);
    uint32_t where[64];
    for (int p = 0; p < 64; ++p) where[p] = (uint32_t)p;
    if (__builtin_cpu_supports("avx2")) sample_core_avx2(g, J, n, k, where); else sample_core_generic(g, J, n, k, where);
    delete g;
    std::free(J);
    for (int p = 0; p < k; ++p) out[p] = (int64_t)where[p];
    return MCR_OK;
}

int mcr_release_cached(int device) {
    std::vector<HostCtx*> ctxs;
    std::vector<StreamFork*> forks;
    {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        for (size_t i = g_idle_ctx.size(); i-- > 0;)
            if (device < 0 || g_idle_ctx[i]->device == device) { ctxs.push_back(g_idle_ctx[i]); g_idle_ctx.erase(g_idle_ctx.begin() + (long)i); }
        for (size_t i = g_idle_fork.size(); i-- > 0;)
            if (device < 0 || g_idle_fork[i]->device == device) { forks.push_back(g_idle_fork[i]); g_idle_fork.erase(g_idle_fork.begin() + (long)i); }
    }
    int rc = MCR_OK;
    for (HostCtx* c : ctxs) {
        DeviceScope scope(c->device);
        if (scope.rc != MCR_OK) rc = scope.rc;   // (device gone: drop the bookkeeping anyway)
        host_ctx_destroy(c);
    }
    for (StreamFork* f : forks) {
        DeviceScope scope(f->device);
        if (scope.rc != MCR_OK) rc = scope.rc;
        stream_fork_destroy(f);
    }
    return rc;
}

int mcr_draw_shocks_host(uint64_t seed, uint32_t stream_id, uint64_t path_begin, uint64_t n_paths,
                         int32_t n_months, double rho, double* out, int device) {
    const mcr_rng r = philox_rng(seed);
    return mcr_draw_shocks_host_rng(&r, stream_id, path_begin, n_paths, n_months, rho, out, device);
}

int mcr_draw_shocks_host_rng(const mcr_rng* rng, uint32_t stream_id, uint64_t path_begin, uint64_t n_paths,
                             int32_t n_months, double rho, double* out, int device) {
    MCR_ENTER_DEVICE(device);
    int rc = MCR_OK;
    rc = check_rng(rng);
    if (rc != MCR_OK) return rc;
    const uint64_t seed = rng->philox_seed;
    if (!out || n_months < 0) { set_error("bad arguments"); return MCR_ERR_INVALID_ARG; }
    const uint64_t total = n_paths * (uint64_t)n_months;
    if (total == 0) return MCR_OK;
    if (total > ((uint64_t)1 << 31)) { set_error("too many shock rows for one call"); return MCR_ERR_INVALID_ARG; }
    DeviceArena arena;
    double* d = nullptr;
    hipError_t e = arena.alloc((void**)&d, (size_t)total * 3 * sizeof(double));
    if (e != hipSuccess) return hip_fail(e, "hipMalloc");
    const double om = 1.0 - rho * rho;
    const double rho_c = std::sqrt(om > 0.0 ? om : 0.0);
    if (rng->kind == MCR_RNG_NUMPY) {
        KernelIO io;
        std::memset(&io, 0, sizeof(io));
        uint32_t* d_seeds = nullptr;
        if (rng->path_seeds) {
            e = arena.alloc((void**)&d_seeds, (size_t)n_paths * sizeof(uint32_t));
            if (e == hipSuccess) e = hipMemcpy(d_seeds, rng->path_seeds, (size_t)n_paths * sizeof(uint32_t), hipMemcpyHostToDevice);
            if (e != hipSuccess) return hip_fail(e, "seed upload");
        }
        fill_io_rng(io, rng, d_seeds);
        io.stream_id = stream_id; io.path_begin = path_begin; io.n_paths = n_paths;
        hipLaunchKernelGGL(np_shocks_kernel, dim3((unsigned)((n_paths + 63) / 64)), dim3(64), 0, nullptr, io, n_months, rho, rho_c, d);
    } else {
        hipLaunchKernelGGL(shocks_kernel, dim3((unsigned)((n_paths + 63) / 64)), dim3(64), 0, nullptr, seed,
                           stream_id, path_begin, n_paths, n_months, rho, rho_c, d);
    }
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(out, d, (size_t)total * 3 * sizeof(double), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return hip_fail(e, "shocks_kernel");
    return MCR_OK;
}

int mcr_eval_helper_host(int which, const mcr_params* p, const double* in, double* out, int64_t n, int device) {
    MCR_ENTER_DEVICE(device);
    int rc = MCR_OK;
    int n_in, n_out;
    switch (which) {
        case MCR_HELPER_WITHDRAW: n_in = 5; n_out = 4; break;
        case MCR_HELPER_NLV: n_in = 4; n_out = 1; break;
        case MCR_HELPER_REBALANCE: n_in = 4; n_out = 4; break;
        case MCR_HELPER_ANNUAL_TAX: n_in = 6; n_out = 5; break;
        case MCR_HELPER_MONTHLY_GROSS: n_in = 3; n_out = 1; break;
        case MCR_HELPER_MATH_EXP: n_in = 1; n_out = 1; break;
        case MCR_HELPER_MATH_DIV: n_in = 2; n_out = 1; break;
        case MCR_HELPER_MATH_DIV_PATH: n_in = 2; n_out = 1; break;
        case MCR_HELPER_WITHDRAW2_PATH: n_in = 6; n_out = 8; break;
        case MCR_HELPER_NLV2_PATH: n_in = 4; n_out = 2; break;
        case MCR_HELPER_REBALANCE_PATH: n_in = 4; n_out = 4; break;
        case MCR_HELPER_ANNUAL_TAX_PATH: n_in = 6; n_out = 5; break;
        case MCR_HELPER_MATH_SQRT: n_in = 1; n_out = 1; break;
        case MCR_HELPER_MATH_NEG2LOG: n_in = 1; n_out = 1; break;
        case MCR_HELPER_MATH_SINCOS: n_in = 1; n_out = 2; break;
        case MCR_HELPER_MATH_EXP_PATH: n_in = 1; n_out = 1; break;
        case MCR_HELPER_MATH_NEG2LOG_PATH: n_in = 1; n_out = 1; break;
        case MCR_HELPER_MATH_SINCOS_PATH: n_in = 1; n_out = 2; break;
        case MCR_HELPER_WITHDRAW_MONTH: n_in = 5; n_out = 6; break;
        case MCR_HELPER_REBALANCE_MONTH: n_in = 4; n_out = 4; break;
        default: set_error("unknown helper %d", which); return MCR_ERR_INVALID_ARG;
    }
    if (!in || !out || n < 0) { set_error("bad arguments"); return MCR_ERR_INVALID_ARG; }
    if (n == 0) return MCR_OK;
    DevParams d;
    std::memset(&d, 0, sizeof(d));
    if (which == MCR_HELPER_REBALANCE || which == MCR_HELPER_ANNUAL_TAX || which == MCR_HELPER_WITHDRAW2_PATH ||
        which == MCR_HELPER_NLV2_PATH || which == MCR_HELPER_REBALANCE_PATH || which == MCR_HELPER_ANNUAL_TAX_PATH ||
        which == MCR_HELPER_WITHDRAW_MONTH || which == MCR_HELPER_REBALANCE_MONTH) {
        if (!p) { set_error("helper %d needs params", which); return MCR_ERR_INVALID_ARG; }
        rc = derive_params(p, 0, &d);
        if (rc != MCR_OK) return rc;
    }
    DeviceArena arena;
    double *din = nullptr, *dout = nullptr;
    hipError_t e = arena.alloc((void**)&din, (size_t)n * n_in * sizeof(double));
    if (e == hipSuccess) e = arena.alloc((void**)&dout, (size_t)n * n_out * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(din, in, (size_t)n * n_in * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) return hip_fail(e, "helper upload");
    hipLaunchKernelGGL(helper_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, which, d, din, dout, n);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(out, dout, (size_t)n * n_out * sizeof(double), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return hip_fail(e, "helper_kernel");
    return MCR_OK;
}

}  // extern "C"
