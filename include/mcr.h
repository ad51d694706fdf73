/*
 * mcr.h — C ABI of the MI355X-native Monte Carlo retirement path engine.
 *
 * This is the drop-in boundary for ONE hot path of rflamino/monte_carlo_retirement:
 * the per-path loop `RetirementMonteCarloSimulator._run_single_simulation_path`
 * (reference backend/simulation.py:476-950) and the batch driver above it
 * (`run_monte_carlo_simulations`, backend/simulation.py:952-1128).  The reference
 * has no FFI layer of its own (it is pure Python); these entry points are what a
 * ctypes binding inside the reference's `run_monte_carlo_simulations` would call
 * (see INTEGRATION.md).  Plain C types only: no torch, no C++ in the signatures.
 *
 * Every compute entry point runs hand-written HIP kernels on a gfx950 device.
 * There is NO CPU fallback in this library: without a usable HIP device every
 * compute call returns MCR_ERR_NO_DEVICE and sets mcr_last_error().
 *
 * All arithmetic on the path is IEEE fp64, as in the reference (Python float).
 */
#ifndef MCR_H_
#define MCR_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCR_ABI_VERSION 7
#define MCR_INLINE_STREAMS 16   /* other_income_streams entries carried INSIDE mcr_params; the rest of the list (any length,
                                   backend/config.py:99) follows through mcr_params.extra_streams */
#define MCR_MAX_PROBE_CANDIDATES 32 /* candidates of one mcr_probe_months_rng call that can share their accumulation sweep */
#define MCR_MAX_HIST_BINS 4096  /* bins of the in-kernel final-balance histogram (mcr_outputs.hist_bins) */
#define MCR_MONTHS_PER_YEAR 12  /* backend/constants.py:1 */
#define MCR_SMALL_EPSILON 1e-6  /* backend/constants.py:3 (absolute dollar threshold) */

/* error codes (0 = success); message text via mcr_last_error() */
#define MCR_OK 0
#define MCR_ERR_INVALID_ARG (-1)
#define MCR_ERR_NO_DEVICE (-2)
#define MCR_ERR_HIP (-3)
#define MCR_ERR_UNSUPPORTED (-4)

/* Seed streams: replaces SeedSequence(main).spawn(2) (backend/simulation.py:147-151). */
#define MCR_STREAM_SEARCH 0u
#define MCR_STREAM_FINAL 1u

/*
 * Random stream of a batch.
 *  MCR_RNG_PHILOX  the engine's native stream: Philox4x32-10, key = philox_seed, counter =
 *                  (path_lo, path_hi, month, stream_id) -> Box-Muller (3 normals / month).
 *  MCR_RNG_NUMPY   the REFERENCE's own stream, reproduced on the device: path i of the batch is
 *                  child (stream_id, child_offset + i) of SeedSequence(main_seed) -> generate_state(1)
 *                  -> default_rng(seed32) = PCG64 -> Generator.standard_normal (ziggurat), exactly
 *                  what backend/simulation.py:148-149,195-197,457-458 executes through NumPy.
 *                  `entropy` = main_seed as little-endian uint32 words; `child_offset` = the stream's
 *                  SeedSequence.n_children_spawned when the reference would spawn this batch's seeds;
 *                  `path_seeds` (optional) = explicit uint32 seed per path instead of the derivation
 *                  (what _run_single_simulation_path(working_months, path_seed) receives).
 */
#define MCR_RNG_PHILOX 0u
#define MCR_RNG_NUMPY 1u
#define MCR_MAX_ENTROPY_WORDS 8
typedef struct mcr_rng {
    uint32_t kind;
    uint32_t n_entropy_words;                /* 1..MCR_MAX_ENTROPY_WORDS (numpy) */
    uint32_t entropy[MCR_MAX_ENTROPY_WORDS]; /* numpy: main_seed words, least significant first */
    uint64_t philox_seed;                    /* philox key */
    uint64_t child_offset;                   /* numpy */
    const uint32_t* path_seeds;              /* numpy, optional; DEVICE ptr for mcr_run_batch_rng, HOST ptr for *_host_rng */
} mcr_rng;

/* One `OtherIncomeStreamConfig` (backend/config.py:12-47), raw field values. */
typedef struct mcr_stream {
    double monthly_amount_today; /* config.py:18 */
    double start_at_age;         /* config.py:23 */
    double tax_rate;             /* config.py:45 */
    int32_t duration_years;      /* config.py:33; -1 encodes None (paid indefinitely) */
    int32_t inflation_indexed;   /* config.py:41 (0/1) */
} mcr_stream;

/*
 * The scalars of `Config` (backend/config.py:48-126) that the path reads, plus the
 * lognormal parameters `RetirementMonteCarloSimulator.__init__` derives from them
 * (backend/simulation.py:156-170, via arithmetic_to_log_params :14-29).
 */
typedef struct mcr_params {
    double initial_balance;                 /* config.py:56 */
    double monthly_contribution;            /* config.py:57 */
    double contribution_growth_rate_annual; /* config.py:58 */
    double monthly_expenses;                /* config.py:59 */
    double current_age;                     /* config.py:62 */
    double allocation_inv1_pct;             /* config.py:70 */
    double inv1_annual_tax_on_gains_rate;   /* config.py:73 */
    double inv1_realized_gains_tax_rate;    /* config.py:74 */
    double inv2_annual_tax_on_gains_rate;   /* config.py:79 */
    double inv2_realized_gains_tax_rate;    /* config.py:80 */
    double inv1_mu_log, inv1_sigma_log;     /* simulation.py:157-159 */
    double inf_mu_log, inf_sigma_log;       /* simulation.py:160-162 */
    double prem_mu_log, prem_sigma_log;     /* simulation.py:163-166 */
    double equity_inflation_rho;            /* simulation.py:170 */
    int32_t retirement_years;               /* config.py:68 (> 0) */
    int32_t inv1_use_realized_gains_tax_system; /* config.py:75 (0/1) */
    int32_t inv2_use_realized_gains_tax_system; /* config.py:81 (0/1) */
    int32_t n_streams;                      /* len(other_income_streams): ANY length >= 0, as in the reference (config.py:99) */
    mcr_stream streams[MCR_INLINE_STREAMS]; /* config.py:99, list order preserved: entries 0 .. min(n_streams, 16) - 1 */
    /* Entries MCR_INLINE_STREAMS .. n_streams - 1 of the list, in order (a HOST pointer on every entry point: the
     * parameter block is always host memory; the library copies the records into a device table next to the launch).
     * Must be non-NULL when n_streams > MCR_INLINE_STREAMS; ignored otherwise.  The loop over the streams is the
     * reference's (backend/simulation.py:602-621, 649-677): O(n_streams) per retirement month, whatever the length. */
    const mcr_stream* extra_streams;
} mcr_params;

/* Shapes implied by (params, working_months); simulation.py:487,585-589,902. */
typedef struct mcr_sizes {
    int32_t total_months;      /* working_months + retirement_years*12 */
    int32_t shock_rows;        /* max(total_months, 1)  (simulation.py:488) */
    int32_t num_working_years; /* ceil(working_months/12) */
    int32_t trajectory_len;    /* T = 1 + num_working_years + retirement_years */
    int32_t retirement_years;  /* rows of the withdrawal-rate trajectory */
    int32_t ruin_bins;         /* retirement_years + 2, see mcr_outputs.ruin_year_bins */
} mcr_sizes;

/* indices into mcr_outputs.counters */
#define MCR_CTR_SUCCESS 0 /* number of paths with Success == True */
#define MCR_CTR_PATHS 1   /* number of paths simulated */
#define MCR_N_COUNTERS 2

/*
 * Output buffers of one batch.  Any pointer may be NULL = "not requested"; the kernel
 * variant is chosen from what is requested (success-count only / + per-path summary /
 * + yearly trajectories).  Layout is struct-of-arrays; trajectories are TIME-MAJOR
 * (`row[t*path_stride + i]` for local path i) so that one wavefront store is 512
 * contiguous bytes.  Field meanings follow the dict returned by
 * _run_single_simulation_path (simulation.py:939-950).
 *
 * For mcr_run_batch the pointers are DEVICE pointers; for mcr_run_batch_host they are
 * HOST pointers.  Counters / bins are ACCUMULATED into (the caller zeroes them), so
 * several launches (or several GPUs before an all-reduce) can share one vector.
 */
typedef struct mcr_outputs {
    double* start_balance;                    /* [n] "Start Balance" */
    double* final_balance;                    /* [n] "Final Balance" = max(0, .) */
    double* years_to_ruin;                    /* [n] "YearsToRuin" (NaN if success) */
    double* first_year_gross_withdrawal;      /* [n] */
    double* first_year_real_gross_withdrawal; /* [n] */
    double* inflation_at_retirement;          /* [n] */
    uint8_t* success;                         /* [n] "Success" (0/1) */
    double* trajectory;                       /* [T][path_stride]  "Trajectory" */
    double* real_trajectory;                  /* [T][path_stride]  "RealTrajectory" */
    double* withdrawal_rate_trajectory;       /* [ry][path_stride] NaN-padded */
    int64_t path_stride;                      /* >= n_paths (elements) */
    uint64_t* counters;                       /* [MCR_N_COUNTERS] */
    uint64_t* wr_obs_counts;                  /* [ry]   wr_df.count(axis=1), simulation.py:1111-1113 */
    uint64_t* ruin_year_bins;                 /* [ry+2] [0]=pre-retirement tax failure (YearsToRuin 0.0,
                                                 simulation.py:628-629); [1+y]=failed in retirement year y;
                                                 [ry+1]=failed at the terminal tax settlement (:894-896) */
    /* In-kernel histogram of "Final Balance" over the successful cohort (the CLI's chart, backend/plotting.py:44-59) on
     * CALLER-SUPPLIED bin edges: hist_bins[k] += #{successful paths: hist_edges[k] <= Final Balance < hist_edges[k+1]}, the
     * last bin closed on the right, values outside [hist_edges[0], hist_edges[hist_n_bins]] dropped — exactly
     * np.histogram(final_balance[success], bins=hist_edges); with hist_edges = np.linspace(lo, hi, n + 1) that is also
     * np.histogram(..., bins=n, range=(lo, hi)), and any other monotone spacing (np.geomspace for log-spaced bins) works
     * the same way.  Every lane bins its own path at the end of the horizon (binary search over the edges, LDS-privatised
     * bins, one global atomic per non-empty bin per workgroup): no per-path output, no second kernel, and the bins sit in
     * the same accumulated integer block as the counters — a multi-GPU caller sums everything with ONE all-reduce.
     * hist_edges: [hist_n_bins + 1] ascending finite doubles (DEVICE pointer for mcr_run_batch*, HOST for *_host*);
     * hist_n_bins in 1..MCR_MAX_HIST_BINS (above 512 bins the LDS footprint costs one resident workgroup per CU);
     * hist_bins == NULL or hist_n_bins == 0: not requested. */
    const double* hist_edges;
    uint64_t* hist_bins;                      /* [hist_n_bins], accumulated into like the counters */
    int32_t hist_n_bins;
    int32_t hist_reserved;                    /* 0 */
} mcr_outputs;

/* ---- library / device ------------------------------------------------------------ */
int mcr_abi_version(void);
/* Number of usable HIP devices (0 if none; never fails). */
int mcr_device_count(void);
/* Thread-local message of the last failing call on this thread ("" if none). */
const char* mcr_last_error(void);

/* Frees the library's idle cached device resources on `device` (< 0: every device): the leased-context pool of the
 * *_host entry points (streams + scratch blocks, see mcr_run_batch_host) and the side streams of mcr_probe_months_rng.
 * Contexts that are in use by concurrent calls are untouched.  Safe to call at any time from any thread. */
int mcr_release_cached(int device);

/* ---- host-side derivations (no device needed) -------------------------------------- */
/* Shapes for (params, working_months).  Returns MCR_ERR_INVALID_ARG for working_months<0,
 * retirement_years<=0, n_streams<0. */
int mcr_query_sizes(const mcr_params* p, int32_t working_months, mcr_sizes* out);
/* Range check of a parameter block — what the reference's pydantic Config enforces (backend/config.py:56-99)
 * and the kernel relies on: amounts finite and >= 0, rates / allocation / stream tax rates in [0, 1], rho in
 * [-1, 1], finite log-parameters with sigma >= 0 and |mu|/12 + 40 sigma/sqrt(12) < 700 (domain of the kernel's
 * exp), n_streams >= 0 (with extra_streams set beyond MCR_INLINE_STREAMS).  Every compute entry point applies it and returns
 * MCR_ERR_INVALID_ARG (message via mcr_last_error) instead of computing with out-of-range inputs.  Not
 * checkable up front: balances are assumed to stay within 1e-6 .. 1e15 (unscaled fp64 division). */
int mcr_validate_params(const mcr_params* p);
/* The five sampled paths of run_monte_carlo_simulations: trajectory_df.sample(n=5, axis=1, random_state=main_seed)
 * (simulation.py:1063-1078) = numpy.random.RandomState(seed).choice(n, k, replace=False), i.e. the first k entries of
 * RandomState(seed).permutation(n): MT19937 seeded with init_genrand(seed), a Fisher-Yates shuffle from the top with
 * masked-rejection bounded draws.  NumPy shuffles an n-element array to get them (7 ms at n = 1e6, 0.7 s at 1e8 — as
 * long as the path kernel takes for n paths); this restatement only generates the n - 1 draws and traces the k wanted
 * positions back through the swaps (branch-free rejection, the tracked positions in one vector register): the same
 * indices, bit for bit (tests/test_abi_cpu.py), in a fifth of NumPy's time.  Host-only (no device needed).  out: HOST int64[k].
 * Needs 1 <= k <= 64, k <= n, n <= 2^32; seed is the 32-bit seed (RandomState rejects larger ones).  Allocates 4 n bytes
 * of scratch on the host for the call and takes ~4 ns per path of sequential host time (MT19937 cannot be split): 40 ms and
 * 40 MB at 10^7 paths, 0.4 s and 400 MB at 10^8 — above 5 x 10^7 paths the call says so once on stderr.  Callers hide it on
 * a host thread next to the asynchronous kernel launch (the drop-in class does). */
int mcr_sample_columns(uint32_t seed, uint64_t n, int32_t k, int64_t* out);
/* stream_payment_start_month_index (simulation.py:47-63). */
int32_t mcr_stream_start_month_index(double current_age, int32_t working_months, double start_at_age);

/* ---- the hot path ------------------------------------------------------------------ */
/*
 * Simulate global paths [path_begin, path_begin+n_paths) for `working_months`, replacing
 * the loop over _path_seeds in run_monte_carlo_simulations (simulation.py:987-990) with one
 * kernel launch (one path per lane).  Shocks: Philox4x32-10 keyed by (seed), counter
 * (path_lo, path_hi, month, stream_id) -> Box-Muller -> rho-mix (replaces _draw_shock_path,
 * simulation.py:452-466).  Row k depends only on (seed, stream, path, k): common random
 * numbers across working-month candidates and across any sharding of the path range.
 *
 * injected_shocks (optional, device pointer, may be NULL): [n_paths][shock_rows][3] doubles
 * (equity, inflation, premium) used INSTEAD of the RNG — the parity hook that mirrors
 * assigning `sim._draw_shock_path` in the reference.
 *
 * out: device pointers.  hip_stream: a hipStream_t (NULL = default stream).  The call is
 * asynchronous: it enqueues on hip_stream and returns.  device: HIP device ordinal.
 * (Count-only Philox launches of at most 3 072 path-wavefronts — 196 608 paths — run a latency-oriented form of the kernel,
 * two waves per 64 paths: same results, bit for bit.  MCR_K1_SPLIT_MAX_WAVES in the environment moves the limit, 0 = never.
 * Philox launches of a few rounds of resident workgroups whose last round would be mostly empty — 10^6 paths are 2.54 rounds —
 * run TIME-SLICED: some path blocks are cut into segments at retirement-year boundaries and dispatched so that the small
 * work items come last; the hand-over state lives in a stream-ordered allocation of the call.  Same results, bit for bit;
 * MCR_K1_SEGMENTS = segments per sliced block, 0 = never.)
 */
int mcr_run_batch(const mcr_params* p, uint64_t seed, uint32_t stream_id,
                  uint64_t path_begin, uint64_t n_paths, int32_t working_months,
                  const double* injected_shocks, const mcr_outputs* out,
                  int device, void* hip_stream);

/* Same, with HOST buffers in `out` / `injected_shocks` (the entry point for non-torch callers, e.g. a ctypes
 * binding inside the reference's run_monte_carlo_simulations, simulation.py:952-1010): the call LEASES a context
 * (a private non-blocking stream + a scratch block its device buffers are carved from) out of a process-wide pool
 * keyed by device, runs uploads / kernel / downloads on that stream, returns after ONE synchronisation of it and
 * hands the context back — concurrent calls from several host threads (the reference's server runs requests on
 * executor threads, server.py:309,405) overlap on the GPU, and short-lived caller threads leave nothing behind.
 * Retention: at most 4 idle contexts per device stay cached, each with a block of at most 256 MiB (larger blocks are
 * freed on return); mcr_release_cached() frees them on demand.
 * device: a HIP device ordinal, or MCR_DEVICE_ALL = shard the path range over every visible device (below). */
#define MCR_DEVICE_ALL (-2)
int mcr_run_batch_host(const mcr_params* p, uint64_t seed, uint32_t stream_id,
                       uint64_t path_begin, uint64_t n_paths, int32_t working_months,
                       const double* injected_shocks, const mcr_outputs* out, int device);

/* The same three entry points with an explicit random-stream descriptor (mcr_rng); the functions
 * above are the MCR_RNG_PHILOX special case. */
int mcr_run_batch_rng(const mcr_params* p, const mcr_rng* rng, uint32_t stream_id, uint64_t path_begin,
                      uint64_t n_paths, int32_t working_months, const double* injected_shocks,
                      const mcr_outputs* out, int device, void* hip_stream);
int mcr_run_batch_host_rng(const mcr_params* p, const mcr_rng* rng, uint32_t stream_id, uint64_t path_begin,
                           uint64_t n_paths, int32_t working_months, const double* injected_shocks,
                           const mcr_outputs* out, int device);
int mcr_draw_shocks_host_rng(const mcr_rng* rng, uint32_t stream_id, uint64_t path_begin, uint64_t n_paths,
                             int32_t n_months, double rho, double* out, int device);

/*
 * Multi-GPU form of mcr_run_batch_host_rng for callers that bind the C ABI without torch.distributed — what
 * replaces the reference's only parallelism, Pool.starmap over independent paths (simulation.py:996-1001):
 * the global path range is cut into contiguous shards, one per listed device (devices == NULL or
 * n_devices <= 0: every visible device), each shard runs on its own host thread / stream / scratch, per-path
 * outputs land in the caller's HOST arrays at the shard's columns, and the counter / bin vectors (< 2 KB) are
 * summed on the host.  The Philox counter carries the GLOBAL path index, so the result is bit-identical to
 * a single-device call whatever the device list.  A device may be listed more than once.
 */
int mcr_run_batch_multi_host_rng(const mcr_params* p, const mcr_rng* rng, uint32_t stream_id, uint64_t path_begin,
                                 uint64_t n_paths, int32_t working_months, const double* injected_shocks,
                                 const mcr_outputs* out, const int32_t* devices, int32_t n_devices);

/*
 * Search driver support (find_minimum_working_months, simulation.py:1138-1342): success counts of
 * SEVERAL candidate working-month counts over the same path range — what the reference obtains by
 * calling run_monte_carlo_simulations(candidate, num_simulations_search) once per candidate and taking
 * _success_probability of each summary (simulation.py:1186-1199).  Philox stream, 2..32 distinct candidates: the
 * accumulation months do not depend on the candidate under common random numbers (simulation.py:513-579), so ONE sweep
 * runs them to the largest candidate and stores the state at the end of every candidate month, and ONE launch
 * (grid.y = candidate) resumes every decumulation from its snapshot; the snapshots are a stream-ordered allocation
 * (80 B per path and candidate; requests that would need more than 4 GiB of them, stream lists beyond the by-value block and
 * parameter blocks that need the exact month take the route below).  Otherwise: one count-only launch per candidate, forked onto internal HIP streams
 * so the candidates share the GPU concurrently, joined back onto `hip_stream`.  Asynchronous like mcr_run_batch.
 * counts: DEVICE uint64 [n_candidates][MCR_N_COUNTERS] = {successes, paths} per candidate (zeroed by
 * the call).  Common random numbers across candidates hold as in the reference (same path range, same
 * stream).
 */
int mcr_probe_months_rng(const mcr_params* p, const mcr_rng* rng, uint32_t stream_id, uint64_t path_begin,
                         uint64_t n_paths, const int32_t* working_months, int32_t n_candidates,
                         uint64_t* counts, int device, void* hip_stream);

/* _draw_shock_path (simulation.py:452-466) for n_paths paths: host out [n_paths][n_months][3]. */
int mcr_draw_shocks_host(uint64_t seed, uint32_t stream_id, uint64_t path_begin,
                         uint64_t n_paths, int32_t n_months, double rho, double* out,
                         int device);

/* ---- device unit functions (the scalar helpers the reference's tests call directly) ---- */
#define MCR_HELPER_WITHDRAW 0      /* _calculate_withdrawal_and_update :201-254; in[5]=(bal,cb,net_target,use_real,rate) out[4]=(bal,cb,gross,net) */
#define MCR_HELPER_NLV 1           /* _net_liquidation_value :256-272;           in[4]=(bal,cb,use_real,rate) out[1] */
#define MCR_HELPER_REBALANCE 2     /* _rebalance_portfolio :274-359;             in[4]=(b1,cb1,b2,cb2) out[4] */
#define MCR_HELPER_ANNUAL_TAX 3    /* _apply_annual_gain_taxes :361-450;         in[6]=(b1,cb1,b2,cb2,g1,g2) out[5]=(b1,cb1,b2,cb2,tax_failed) */
#define MCR_HELPER_MONTHLY_GROSS 4 /* _monthly_gross_from_shock :468-474;        in[3]=(mu_log,sigma_log,z) out[1] */
/* the kernel's specialised fp64 math (csrc/mcr_math.h), exposed so its accuracy can be measured */
#define MCR_HELPER_MATH_EXP 5      /* in[1]=(x)            out[1]=exp(x)                              */
#define MCR_HELPER_MATH_DIV 6      /* in[2]=(a,b)          out[1]=a/b (Newton sequence, no scaling)   */
#define MCR_HELPER_MATH_SQRT 7     /* in[1]=(w)            out[1]=sqrt(w)                             */
#define MCR_HELPER_MATH_NEG2LOG 8  /* in[1]=(x as uint32)  out[1]=-2 ln((x+0.5) 2^-32)                */
#define MCR_HELPER_MATH_SINCOS 9   /* in[1]=(x as uint32)  out[2]=(sin, cos)(2 pi (x+0.5) 2^-32)      */
#define MCR_HELPER_MATH_DIV_PATH 10 /* in[2]=(a,b)         out[1]=a/b as the path kernel divides (1 Newton step) */
/* The PATH FORMS of the state-machine helpers — the code the path kernel actually runs (clamps that are provable no-ops
 * on reachable states dropped, selections as masked moves, compile-time tax variants chosen from the parameter block as
 * the launcher does).  Inputs must be reachable states: balances and cost bases >= 0, targets >= 0.  Rates, allocation
 * and the tax systems come from the parameter block. */
#define MCR_HELPER_WITHDRAW2_PATH 11  /* in[6]=(b1,cb1,target1,b2,cb2,target2) out[8]=(b1,cb1,gross1,net1,b2,cb2,gross2,net2) */
#define MCR_HELPER_NLV2_PATH 12       /* in[4]=(b1,cb1,b2,cb2) out[2] */
#define MCR_HELPER_REBALANCE_PATH 13  /* in[4]=(b1,cb1,b2,cb2) out[4] */
#define MCR_HELPER_ANNUAL_TAX_PATH 14 /* in[6]=(b1,cb1,b2,cb2,g1,g2) out[5]=(b1,cb1,b2,cb2,tax_failed) */
/* the PATH FORMS of the math (shorter series inside the month loop: their truncation errors are sized to the 1e-9 path
 * tolerance instead of the last ulp — csrc/mcr_math.h; bounds measured in tests/test_gpu_math.py) */
#define MCR_HELPER_MATH_EXP_PATH 15     /* in[1]=(x)            out[1]=exp(x), r^2/24 -> a^2/40 (|err| <= 3.5e-15, mean 0) */
#define MCR_HELPER_MATH_NEG2LOG_PATH 16 /* in[1]=(x as uint32)  out[1]=-2 ln((x+0.5) 2^-32), series to r^4 (<= 3.6e-13 absolute) */
#define MCR_HELPER_MATH_SINCOS_PATH 17  /* in[1]=(x as uint32)  out[2]=(sin, cos), cos series to dl^4 (<= 4.7e-15 absolute)  */
/* The month as the path kernel RUNS it since ABI v7 (csrc/mcr_device.h, "TOLERANCE FORM of the month"): the reference's
 * formulas in closed form — both assets sell the fraction target / capacity of their balance, the rebalance is one quotient —
 * with fused multiply-adds and uncorrected reciprocals: equal to the reference's arithmetic up to roundings (~1e-16 relative per
 * operation) while both effective realized-gains rates are <= 1 - 1e-6; parameter blocks with a higher rate run the exact path
 * forms above (the reference's denominator clamps, simulation.py:227,:307-310, can bind there), and so do these two helpers. */
#define MCR_HELPER_WITHDRAW_MONTH 18   /* :726-790  in[5]=(b1,cb1,b2,cb2,need) out[6]=(b1,cb1,b2,cb2,gross withdrawn,net cash) */
#define MCR_HELPER_REBALANCE_MONTH 19  /* :274-359  in[4]=(b1,cb1,b2,cb2) out[4] */
/* Evaluates helper `which` ON THE DEVICE for n rows (host buffers, row-major). */
int mcr_eval_helper_host(int which, const mcr_params* p, const double* in, double* out,
                         int64_t n, int device);

/* ---- device-side aggregation (replaces the pandas block, simulation.py:1045-1118) ---- */
/*
 * Row-wise quantiles with pandas' semantics (linear interpolation between order statistics,
 * NaNs skipped): for each of n_rows rows of `n` doubles at rows[r*row_stride + i], writes
 * out[r*n_q + j] = quantile(q[j]) (NaN for an all-NaN row) and, if counts != NULL,
 * counts[r] = number of non-NaN entries.  rows/out/counts are DEVICE pointers (out / counts may also be pinned,
 * device-visible host memory — a few KB the kernels write directly: no download after the call); q is HOST.
 * scratch: device buffer of mcr_row_quantiles_scratch_bytes(n_rows, n_q, n) bytes (selection state, digit and
 * sub-bin histograms, per-row candidate buffers of n/64 + 4096 keys and n/8 + 4096 values, cell lists); 0 =
 * unsupported shape.
 *
 * mcr_row_quantiles picks the route by row length.  Short rows: the exact radix select (8 digit passes,
 * 4 of them over the slab), fully asynchronous.  Rows of >= 2^14 entries, SEVEN launches: the first 4096 entries of
 * every row are sorted in LDS (coarse brackets); a counting pass over a sample (the first n/32 entries) and a small
 * per-row kernel turn them into fine brackets; ONE pass over the slab counts the entries around the brackets (with
 * sub-histograms inside them) and compacts the few % inside — two kernels side by side: rows whose bracket bounds fit a
 * 2048-bucket value table take the table-driven one (-0.0 is counted and returned as +0.0 there), the others a generic
 * search; two per-row kernels locate every target in one
 * sub-bin, collect that bin's keys and select among them.  Rows whose counts do not prove their brackets right
 * take the radix passes afterwards.  The result is exact on either route.  The second route reads one word back
 * from the device (how many rows need the radix passes): it synchronises hip_stream ONCE per call.
 * mcr_row_quantiles_last_fallback_rows reports, for the calling thread's last call, -1 (radix route) or the
 * number of rows that needed the radix passes (diagnostics / tests).
 */
int64_t mcr_row_quantiles_scratch_bytes(int32_t n_rows, int32_t n_q, int64_t n);
/*
 * The same selection as separate steps, for rows SHARDED across GPUs (each rank holds n_local of the
 * n_total entries of every row): begin once, then for pass = 0..7: hist (rank-local digit histograms;
 * pass 3 also compacts candidates) -> the caller sums the dense 32-bit counter block of the scratch
 * buffer across ranks (byte offset / word count from mcr_row_quantiles_reduce_block; one small
 * all-reduce per pass, the trajectories themselves never move) -> scan (identical on every rank).
 * After pass 7, `out`/`counts` hold the exact global quantiles on every rank.
 */
int64_t mcr_row_quantiles_reduce_block(int32_t n_rows, int64_t* n_words);
int mcr_row_quantiles_begin(void* scratch, int32_t n_rows, int device, void* hip_stream);
int mcr_row_quantiles_hist(const double* rows, int64_t row_stride, int32_t n_rows, int64_t n_local, int32_t n_q,
                           int32_t pass, void* scratch, int device, void* hip_stream);
int mcr_row_quantiles_scan(int32_t n_rows, int64_t n_total, const double* q, int32_t n_q, int32_t pass, double* out,
                           uint64_t* counts, void* scratch, int device, void* hip_stream);
int mcr_row_quantiles(const double* rows, int64_t row_stride, int32_t n_rows, int64_t n,
                      const double* q, int32_t n_q, double* out, uint64_t* counts,
                      void* scratch, int device, void* hip_stream);
int mcr_row_quantiles_last_fallback_rows(void);
/*
 * The bracketed route for rows SHARDED across GPUs (rank r holds n_local of the n_total entries of every row; the
 * trajectories never move).  Between its stages the library calls `reduce(ctx, device_buf, count, dtype)`, which
 * must SUM the `count` elements of `device_buf` (MCR_DT_I32: int32, MCR_DT_I64: int64; always inside `scratch`)
 * across all ranks, in place, ordered after the work already enqueued on hip_stream (an RCCL all-reduce on that stream,
 * or a synchronising host implementation), and return 0.  About ten such calls per invocation, the largest the two
 * sub-histogram blocks (n_rows * 64 KiB) and the cell lists (n_rows * 96 KiB).  Every rank must call with the same
 * n_rows, n_total, q; every rank returns the same exact quantiles.  Needs rank 0 to hold >= 4096 entries, world <= 64,
 * fewer than 16 quantiles: otherwise MCR_ERR_UNSUPPORTED (use the stepwise radix select above) — on EVERY rank: the one
 * condition that depends on a single rank's shard (rank 0's first sample) is agreed through the first `reduce` call, so no
 * rank leaves while its peers wait in a collective.  A `reduce` callback that fails on one rank only is the caller's to
 * avoid (the peers would wait in the collective it skipped).  Synchronises hip_stream a few times.
 * scratch: mcr_row_quantiles_scratch_bytes(n_rows, n_q, n_local).
 */
#define MCR_DT_I32 0
#define MCR_DT_I64 1
typedef int (*mcr_reduce_fn)(void* ctx, void* device_buf, int64_t count, int32_t dtype);
int mcr_row_quantiles_sharded(const double* rows, int64_t row_stride, int32_t n_rows, int64_t n_local, int64_t n_total,
                              const double* q, int32_t n_q, double* out, uint64_t* counts, void* scratch, int32_t rank,
                              int32_t world, mcr_reduce_fn reduce, void* reduce_ctx, int device, void* hip_stream);

/*
 * Histogram of final balances over the successful cohort (the CLI's 100-bin chart,
 * backend/plotting.py:53-59, np.histogram semantics: n_bins equal-width bins over
 * [lo, hi], last bin closed).  values/success: DEVICE [n].  minmax: DEVICE [2] doubles;
 * if use_given_range == 0 the kernel first reduces min/max of the cohort into it.
 * bins: DEVICE [n_bins] uint64, accumulated into (caller zeroes).
 */
int mcr_minmax_success(const double* values, const uint8_t* success, int64_t n,
                       double* minmax, int device, void* hip_stream);
int mcr_histogram_success(const double* values, const uint8_t* success, int64_t n,
                          const double* minmax, int32_t n_bins, uint64_t* bins,
                          int device, void* hip_stream);

/*
 * Rows for the summary statistics of the response document (backend/server.py:446-458 and
 * median_first_year_withdrawal_rate, backend/simulation.py:78-96), laid out for mcr_row_quantiles
 * (which skips NaN exactly as pandas' median/quantile do):
 *   row 0 = Start Balance                      row 1 = Final Balance
 *   row 2 = Final Balance where Success, else NaN
 *   row 3 = First Year Real Gross Withdrawal / Start Balance * 100 where Start Balance > 1e-6, else NaN
 * Inputs: DEVICE [n].  rows: DEVICE [4][row_stride] (row_stride >= n).
 */
#define MCR_N_STAT_ROWS 4
int mcr_summary_stat_rows(const double* start_balance, const double* final_balance,
                          const double* first_year_real_gross, const uint8_t* success, int64_t n,
                          double* rows, int64_t row_stride, int device, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* MCR_H_ */
