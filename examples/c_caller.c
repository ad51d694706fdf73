/* A caller of the C ABI in plain C99: no Python, no torch.  The config.json scenario, success-count only, the path
 * range sharded over every visible GPU by the library (MCR_DEVICE_ALL).
 *
 *   gcc -std=c99 -O2 -Iinclude examples/c_caller.c -o c_caller \
 *       -Lmonte_carlo_retirement_amd/csrc -lmcr_hip -Wl,-rpath,$PWD/monte_carlo_retirement_amd/csrc -lm
 *   ./c_caller [n_paths] [working_months] [seed]
 *
 * prints `paths=<n> success=<count> probability=<p> devices=<d>` and `hist=<10 comma-separated bin counts>`.
 * The lognormal parameters follow arithmetic_to_log_params (backend/simulation.py:14-29), as __init__ derives them
 * (:156-170). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mcr.h"

static void log_params(double mean, double vol, double* mu, double* sigma) { /* simulation.py:14-29 */
    const double g = 1.0 + mean;
    const double s2 = log(1.0 + (vol * vol) / (g * g));
    *sigma = sqrt(s2);
    *mu = log(g) - 0.5 * s2;
}

int main(int argc, char** argv) {
    const uint64_t n_paths = argc > 1 ? strtoull(argv[1], NULL, 10) : 100000ull;
    const int32_t working_months = argc > 2 ? atoi(argv[2]) : 233;
    const uint64_t seed = argc > 3 ? strtoull(argv[3], NULL, 10) : 12345ull;

    mcr_params p;
    memset(&p, 0, sizeof p);
    p.initial_balance = 240000.0;
    p.monthly_contribution = 5000.0;
    p.contribution_growth_rate_annual = 0.04;
    p.monthly_expenses = 10000.0;
    p.current_age = 40.0;
    p.retirement_years = 50;
    p.allocation_inv1_pct = 0.6;
    p.inv1_annual_tax_on_gains_rate = 0.0;
    p.inv1_realized_gains_tax_rate = 0.1;
    p.inv1_use_realized_gains_tax_system = 1;
    p.inv2_annual_tax_on_gains_rate = 0.0;
    p.inv2_realized_gains_tax_rate = 0.1;
    p.inv2_use_realized_gains_tax_system = 1;
    log_params(0.12, 0.02, &p.inv1_mu_log, &p.inv1_sigma_log);
    log_params(0.062, 0.0235, &p.inf_mu_log, &p.inf_sigma_log);
    log_params(0.05, 0.02, &p.prem_mu_log, &p.prem_sigma_log);
    p.equity_inflation_rho = 0.0;
    p.n_streams = 2;
    p.streams[0].monthly_amount_today = 4000.0; /* State Pension */
    p.streams[0].start_at_age = 65.0;
    p.streams[0].duration_years = -1; /* None: paid indefinitely */
    p.streams[0].inflation_indexed = 1;
    p.streams[0].tax_rate = 0.275;
    p.streams[1].monthly_amount_today = 0.0; /* Rental Income (Apt) */
    p.streams[1].start_at_age = 40.0;
    p.streams[1].duration_years = 35;
    p.streams[1].inflation_indexed = 0;
    p.streams[1].tax_rate = 0.2;

    if (mcr_abi_version() != MCR_ABI_VERSION) {
        fprintf(stderr, "ABI mismatch: header %d, library %d\n", MCR_ABI_VERSION, mcr_abi_version());
        return 2;
    }
    if (mcr_validate_params(&p) != MCR_OK) {
        fprintf(stderr, "invalid scenario: %s\n", mcr_last_error());
        return 2;
    }
    const int devices = mcr_device_count();
    if (devices < 1) {
        fprintf(stderr, "no HIP device (this engine has no CPU path)\n");
        return 3;
    }

    mcr_rng rng;
    memset(&rng, 0, sizeof rng);
    rng.kind = MCR_RNG_PHILOX;
    rng.philox_seed = seed;

    uint64_t counters[MCR_N_COUNTERS] = {0, 0};
    mcr_outputs out;
    memset(&out, 0, sizeof out); /* every per-path pointer NULL: the success-count-only kernel */
    out.path_stride = (int64_t)n_paths;
    out.counters = counters;
    /* final balances of the successful paths on 10 log-spaced bins from 1e5 to 1e10 dollars, binned inside the path
     * kernel (np.histogram(final[success], bins=edges); backend/plotting.py:44-59 charts this cohort) */
    enum { N_BINS = 10 };
    double edges[N_BINS + 1];
    uint64_t bins[N_BINS];
    for (int k = 0; k <= N_BINS; ++k) edges[k] = 1e5 * pow(10.0, 0.5 * k);
    memset(bins, 0, sizeof bins);
    out.hist_edges = edges;
    out.hist_bins = bins;
    out.hist_n_bins = N_BINS;

    const int rc = mcr_run_batch_multi_host_rng(&p, &rng, /*stream_id=*/1u, /*path_begin=*/0u, n_paths, working_months,
                                                /*injected_shocks=*/NULL, &out, /*devices=*/NULL, /*n_devices=*/0);
    if (rc != MCR_OK) {
        fprintf(stderr, "mcr_run_batch_multi_host_rng: %d (%s)\n", rc, mcr_last_error());
        return 1;
    }
    printf("paths=%llu success=%llu probability=%.6f devices=%d\n", (unsigned long long)counters[MCR_CTR_PATHS],
           (unsigned long long)counters[MCR_CTR_SUCCESS], (double)counters[MCR_CTR_SUCCESS] / (double)counters[MCR_CTR_PATHS], devices);
    printf("hist=");
    for (int k = 0; k < N_BINS; ++k) printf("%llu%s", (unsigned long long)bins[k], k + 1 < N_BINS ? "," : "\n");
    return 0;
}
