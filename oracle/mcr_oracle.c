/*
 * mcr_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, scalar, fp64 CPU restatement of the reference's per-path algorithm
 * (rflamino/monte_carlo_retirement, backend/simulation.py), used ONLY as the checker
 * in tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under
 * monte_carlo_retirement_amd/ may import, link or call it.
 *
 * Parity status: PINNED.  In shock-injection mode this file reproduces the reference's
 * own `_run_single_simulation_path` BIT-FOR-BIT (all 10 result keys) on the golden vectors
 * under tests/golden/, which were produced by importing /root/reference/backend in the
 * build container (tests/golden/generate_golden.py) — same glibc `exp`, same IEEE
 * double arithmetic, no FMA contraction (-ffp-contract=off).  The reference's own
 * closed-form unit pins (tests/test_simulation_correctness.py:605-662, :335-361) are
 * replayed in tests/test_oracle_golden.py (helper KATs of tests/golden/helpers.json).
 *
 * RNG: the reference draws shocks with NumPy (SeedSequence -> PCG64 -> ziggurat,
 * simulation.py:457-458).  The engine's stream is Philox4x32-10 + Box-Muller, restated
 * here (orc_draw_shocks); the reference's tests pin no RNG-dependent number, so at the RNG
 * boundary parity is by construction (identical shocks injected into the reference loop).
 *
 * Each function cites the reference lines it follows (file backend/simulation.py unless
 * another file is named).  Python's max(a,b)/min(a,b) are restated as pymax/pymin so that
 * argument order (which decides ties, signed zeros and NaNs) is preserved.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/mcr.h"

#define MPY MCR_MONTHS_PER_YEAR
#define EPS MCR_SMALL_EPSILON

/* Python builtins: max(a, b) returns a unless b > a; min(a, b) returns a unless b < a. */
static inline double pymax(double a, double b) { return (b > a) ? b : a; }
static inline double pymin(double a, double b) { return (b < a) ? b : a; }

/* ------------------------------------------------------------------------------------ */
/* a1: arithmetic_to_log_params, simulation.py:14-29.  Returns 0, or -1 for the ValueErrors. */
int orc_arithmetic_to_log_params(double mean, double vol, double* mu_log, double* sigma_log) {
    if (mean <= -1.0) return -1; /* :19-20 */
    if (vol < 0) return -1;      /* :21-22 */
    if (vol == 0) {              /* :23-25 */
        *mu_log = log(1.0 + mean);
        *sigma_log = 0.0;
        return 0;
    }
    double one_plus_mean = 1.0 + mean;                                         /* :26 */
    double s = sqrt(log(1.0 + (vol * vol) / (one_plus_mean * one_plus_mean))); /* :27 */
    *sigma_log = s;
    *mu_log = log(one_plus_mean) - 0.5 * (s * s); /* :28 */
    return 0;
}

/* a2: retirement_age :32-34, stream_payment_start_age :37-44, ..._start_month_index :47-63 */
int32_t orc_stream_start_month_index(double current_age, int32_t working_months, double start_at_age) {
    double retirement_start = current_age + (double)working_months / (double)MPY; /* :34 */
    double eligible_age = pymax(retirement_start, start_at_age);                  /* :44 */
    double c = ceil((eligible_age - retirement_start) * (double)MPY - EPS);       /* :58-61 */
    int32_t ic = (int32_t)c;
    return ic > 0 ? ic : 0; /* max(0, int(...)) :55-63 */
}

/* a14: trajectory_time_points :99-123.  out must hold 1 + ceil(wm/12) + ry doubles; returns count. */
int32_t orc_trajectory_time_points(int32_t working_months, int32_t retirement_years, double* out) {
    int32_t full = working_months / MPY, rem = working_months % MPY; /* :109-111 */
    int32_t n = 0;
    out[n++] = 0.0;                                         /* :112 */
    for (int32_t y = 1; y <= full; ++y) out[n++] = (double)y; /* :113 */
    double retirement_time = (double)working_months / (double)MPY; /* :115 */
    if (rem) out[n++] = retirement_time;                    /* :116-117 */
    for (int32_t y = 1; y <= retirement_years; ++y) out[n++] = retirement_time + (double)y; /* :119-122 */
    return n;
}

/* a8: _calculate_withdrawal_and_update :201-254 */
void orc_withdraw(double bal_inv, double cb_inv, double net_target, int use_real_tax,
                  double real_tax_rate, double* new_bal, double* new_cb, double* gross,
                  double* net) {
    if (bal_inv <= EPS || net_target <= 0) { /* :218-219 */
        *new_bal = pymax(0.0, bal_inv);
        *new_cb = pymax(0.0, cb_inv);
        *gross = 0.0;
        *net = 0.0;
        return;
    }
    double gain_fraction = pymax(0.0, bal_inv - cb_inv) / bal_inv; /* :221 */
    double effective_tax_fraction =
        (use_real_tax && real_tax_rate > 0) ? gain_fraction * real_tax_rate : 0.0; /* :222-226 */
    double net_fraction = pymax(EPS, 1.0 - effective_tax_fraction);                /* :227 */
    double gross_withdrawal = pymin(net_target / net_fraction, bal_inv);           /* :228-231 */
    double fraction_sold = pymin(1.0, gross_withdrawal / bal_inv);                 /* :233 */
    double basis_removed = pymin(cb_inv, cb_inv * fraction_sold);                  /* :234 */
    double taxable_gain = pymax(0.0, gross_withdrawal - basis_removed);            /* :235 */
    double tax_paid = (use_real_tax && real_tax_rate > 0) ? taxable_gain * real_tax_rate : 0.0; /* :236-240 */
    double net_cash = pymax(0.0, gross_withdrawal - tax_paid);                     /* :241 */
    double nb = pymax(0.0, bal_inv - gross_withdrawal);                            /* :243 */
    double ncb = pymax(0.0, cb_inv - basis_removed);                               /* :244 */
    if (nb <= EPS) { nb = 0.0; ncb = 0.0; }                                        /* :245-247 */
    *new_bal = nb; *new_cb = ncb; *gross = gross_withdrawal; *net = net_cash;
}

/* a9: _net_liquidation_value :256-272 */
double orc_nlv(double balance, double cost_basis, int use_real, double rate) {
    if (balance <= EPS) return 0.0;                                   /* :264-265 */
    double taxable_gain = pymax(0.0, balance - cost_basis);           /* :266 */
    double tax = (use_real && rate > 0) ? taxable_gain * rate : 0.0;  /* :267-271 */
    return pymax(0.0, balance - tax);                                 /* :272 */
}

/* a7: _rebalance_portfolio :274-359 */
void orc_rebalance(const mcr_params* p, double* b1, double* cb1, double* b2, double* cb2) {
    double bal1 = *b1, c1 = *cb1, bal2 = *b2, c2 = *cb2;
    double alloc1 = p->allocation_inv1_pct;
    double alloc2 = 1.0 - p->allocation_inv1_pct; /* config.py:124-126 */
    double total = bal1 + bal2;                   /* :288 */
    if (total <= EPS) return;                     /* :290-291 */
    double target1 = total * alloc1;              /* :293 */
    double drift1 = bal1 - target1;               /* :294 */
    if (fabs(drift1) <= EPS) return;              /* :295-296 */
    double nb1, nc1, nb2, nc2;
    if (drift1 > 0) { /* sell inv1 :298-325 */
        double gain_fraction = pymax(0.0, bal1 - c1) / bal1; /* :301 */
        double tax_per_dollar = p->inv1_use_realized_gains_tax_system
                                    ? gain_fraction * p->inv1_realized_gains_tax_rate : 0.0; /* :302-306 */
        double denom = pymax(EPS, 1.0 - alloc1 * tax_per_dollar); /* :307-310 */
        double gross_sale = pymin(bal1, drift1 / denom);          /* :311 */
        double fraction_sold = gross_sale / bal1;                 /* :312 */
        double basis_removed = pymin(c1, c1 * fraction_sold);     /* :313 */
        double taxable_gain = pymax(0.0, gross_sale - basis_removed); /* :314 */
        double tax_paid = p->inv1_use_realized_gains_tax_system
                              ? taxable_gain * p->inv1_realized_gains_tax_rate : 0.0; /* :315-319 */
        double net_purchase = gross_sale - tax_paid; /* :320 */
        nb1 = pymax(0.0, bal1 - gross_sale);         /* :322 */
        nc1 = pymax(0.0, c1 - basis_removed);        /* :323 */
        nb2 = bal2 + net_purchase;                   /* :324 */
        nc2 = c2 + net_purchase;                     /* :325 */
    } else { /* sell inv2 :326-353 */
        double drift2 = bal2 - total * alloc2;                /* :328 */
        double gain_fraction = pymax(0.0, bal2 - c2) / bal2;  /* :329 */
        double tax_per_dollar = p->inv2_use_realized_gains_tax_system
                                    ? gain_fraction * p->inv2_realized_gains_tax_rate : 0.0; /* :330-334 */
        double denom = pymax(EPS, 1.0 - alloc2 * tax_per_dollar); /* :335-338 */
        double gross_sale = pymin(bal2, drift2 / denom);          /* :339 */
        double fraction_sold = gross_sale / bal2;                 /* :340 */
        double basis_removed = pymin(c2, c2 * fraction_sold);     /* :341 */
        double taxable_gain = pymax(0.0, gross_sale - basis_removed); /* :342 */
        double tax_paid = p->inv2_use_realized_gains_tax_system
                              ? taxable_gain * p->inv2_realized_gains_tax_rate : 0.0; /* :343-347 */
        double net_purchase = gross_sale - tax_paid; /* :348 */
        nb2 = pymax(0.0, bal2 - gross_sale);         /* :350 */
        nc2 = pymax(0.0, c2 - basis_removed);        /* :351 */
        nb1 = bal1 + net_purchase;                   /* :352 */
        nc1 = c1 + net_purchase;                     /* :353 */
    }
    if (nb1 <= EPS) { nb1 = 0.0; nc1 = 0.0; } /* :355-356 */
    if (nb2 <= EPS) { nb2 = 0.0; nc2 = 0.0; } /* :357-358 */
    *b1 = nb1; *cb1 = nc1; *b2 = nb2; *cb2 = nc2;
}

/* a10: _apply_annual_gain_taxes :361-450.  Returns tax_failed. */
int orc_annual_tax(const mcr_params* p, double* b1, double* cb1, double* b2, double* cb2,
                   double gain1, double gain2) {
    int use1 = p->inv1_use_realized_gains_tax_system, use2 = p->inv2_use_realized_gains_tax_system;
    double tax_due1 = !use1 ? pymax(0.0, gain1) * p->inv1_annual_tax_on_gains_rate : 0.0; /* :380-384 */
    double tax_due2 = !use2 ? pymax(0.0, gain2) * p->inv2_annual_tax_on_gains_rate : 0.0; /* :385-389 */
    double total_tax_due = tax_due1 + tax_due2;                                           /* :390 */
    double cap1 = orc_nlv(*b1, *cb1, use1, p->inv1_realized_gains_tax_rate);              /* :392-397 */
    double cap2 = orc_nlv(*b2, *cb2, use2, p->inv2_realized_gains_tax_rate);              /* :398-403 */
    double total_capacity = cap1 + cap2;                                                  /* :404 */
    double net_tax_payment = pymin(total_tax_due, total_capacity);                        /* :405 */
    int tax_failed = net_tax_payment < total_tax_due - EPS;                               /* :406 */
    if (total_capacity > EPS && net_tax_payment > 0) {                                    /* :408 */
        double share1 = cap1 / total_capacity; /* :409 */
        double share2 = 1.0 - share1;          /* :410 */
        double g, net1, net2;
        orc_withdraw(*b1, *cb1, net_tax_payment * share1, use1, p->inv1_realized_gains_tax_rate,
                     b1, cb1, &g, &net1); /* :411-419 */
        orc_withdraw(*b2, *cb2, net_tax_payment * share2, use2, p->inv2_realized_gains_tax_rate,
                     b2, cb2, &g, &net2); /* :420-428 */
        if (net1 + net2 < total_tax_due - EPS) tax_failed = 1; /* :429-430 */
    }
    orc_rebalance(p, b1, cb1, b2, cb2); /* :432-442 */
    return tax_failed;
}

/* a6: _monthly_gross_from_shock :468-474 */
double orc_monthly_gross(double mu_log, double sigma_log, double z) {
    return exp(mu_log / (double)MPY + sigma_log / sqrt((double)MPY) * z); /* :473 */
}

/* ------------------------------------------------------------------------------------ */
/* Engine RNG (replaces numpy default_rng/standard_normal, simulation.py:457-458).
 * Philox4x32-10 (Salmon et al., SC'11; Random123 constants), counter
 * (path_lo, path_hi, month, stream_id), key (seed_lo, seed_hi). */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

#define ORC_TWO_PI 6.283185307179586476925286766559
/*
 * The path's standard-normal sequence n[0], n[1], ... (engine definition, replaces the NumPy draw):
 *   pair j = Box-Muller of two Philox words (xr, xa): u = (x + 0.5) 2^-32,
 *            n[2j] = sqrt(-2 ln u_r) cos(2 pi u_a),  n[2j+1] = sqrt(-2 ln u_r) sin(2 pi u_a);
 *   pair j takes words 2(j&1), 2(j&1)+1 of Philox block j>>1, counter (path_lo, path_hi, block, stream_id),
 *   key (seed_lo, seed_hi).
 * Shock row k (absolute month k+1) = (equity, inflation, premium) =
 *   (n[3k], rho n[3k] + sqrt(max(0,1-rho^2)) n[3k+1], n[3k+2])          (simulation.py:459-466)
 * so 4 rows consume exactly 3 Philox blocks = 6 pairs, and row k is a pure function of
 * (seed, stream, path, k): common random numbers across working-month candidates.
 */
static void orc_bm_pair(uint32_t xr, uint32_t xa, double* zc, double* zs) {
    const double S = 2.3283064365386962890625e-10; /* 2^-32 */
    double ur = ((double)xr + 0.5) * S, ua = ((double)xa + 0.5) * S;
    double r = sqrt(-2.0 * log(ur));
    *zc = r * cos(ORC_TWO_PI * ua);
    *zs = r * sin(ORC_TWO_PI * ua);
}

/* n[i] for one path, from scratch */
double orc_path_normal(uint64_t seed, uint32_t stream_id, uint64_t path, uint64_t i) {
    uint64_t pair = i >> 1, block = pair >> 1;
    uint32_t ctr[4] = {(uint32_t)path, (uint32_t)(path >> 32), (uint32_t)block, stream_id};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t x[4];
    double zc, zs;
    orc_philox4x32_10(ctr, key, x);
    orc_bm_pair(x[2 * (pair & 1)], x[2 * (pair & 1) + 1], &zc, &zs);
    return (i & 1) ? zs : zc;
}

void orc_shock_row(uint64_t seed, uint32_t stream_id, uint64_t path, uint32_t month, double rho,
                   double out[3]) {
    double z0 = orc_path_normal(seed, stream_id, path, 3ull * month);
    double z1 = orc_path_normal(seed, stream_id, path, 3ull * month + 1);
    double z2 = orc_path_normal(seed, stream_id, path, 3ull * month + 2);
    out[0] = z0;
    out[1] = rho * z0 + sqrt(pymax(0.0, 1.0 - rho * rho)) * z1; /* :461-464 */
    out[2] = z2;
}

/* _draw_shock_path equivalent: out[n_months][3] for one path. */
void orc_draw_shocks(uint64_t seed, uint32_t stream_id, uint64_t path, int32_t n_months, double rho,
                     double* out) {
    for (int32_t m = 0; m < n_months; ++m)
        orc_shock_row(seed, stream_id, path, (uint32_t)m, rho, out + 3 * (size_t)m);
}

/* ------------------------------------------------------------------------------------ */
int orc_query_sizes(const mcr_params* p, int32_t wm, mcr_sizes* s) {
    if (!p || !s || wm < 0 || p->retirement_years <= 0 || p->n_streams < 0 ||
        (p->n_streams > MCR_INLINE_STREAMS && !p->extra_streams))
        return MCR_ERR_INVALID_ARG;
    s->total_months = wm + p->retirement_years * MPY;                   /* :487 */
    s->shock_rows = s->total_months > 1 ? s->total_months : 1;          /* :488 */
    s->num_working_years = wm > 0 ? (wm + MPY - 1) / MPY : 0;           /* :585-589 */
    s->trajectory_len = 1 + s->num_working_years + p->retirement_years; /* :902 */
    s->retirement_years = p->retirement_years;
    s->ruin_bins = p->retirement_years + 2;
    return MCR_OK;
}

/* entry s of other_income_streams: the first MCR_INLINE_STREAMS records sit in the block, the rest behind extra_streams */
static inline const mcr_stream* orc_stream(const mcr_params* p, int32_t s) {
    return s < MCR_INLINE_STREAMS ? &p->streams[s] : &p->extra_streams[s - MCR_INLINE_STREAMS];
}

typedef struct orc_path_result {
    double start_balance, final_balance, years_to_ruin;
    double first_year_gross, first_year_real_gross, inflation_at_retirement;
    int success;
    int ruin_bin; /* index into mcr_outputs.ruin_year_bins, -1 if success */
} orc_path_result;

/*
 * a11: _run_single_simulation_path :476-950.  `shocks` = [shock_rows][3] for this path.
 * traj/real_traj: [T] (stride `ts`), wr: [ry] (stride `ts`); any may be NULL.
 */
static void orc_single_path(const mcr_params* p, int32_t working_months, const double* shocks,
                            int32_t shock_rows, orc_path_result* res, double* traj_out,
                            double* real_out, double* wr_out, int64_t ts, uint64_t* wr_obs_counts) {
    const int32_t ry = p->retirement_years;
    const int use1 = p->inv1_use_realized_gains_tax_system, use2 = p->inv2_use_realized_gains_tax_system;
    const double rate1 = p->inv1_realized_gains_tax_rate, rate2 = p->inv2_realized_gains_tax_rate;
    const double alloc1 = p->allocation_inv1_pct;
    const int32_t num_working_years = working_months > 0 ? (working_months + MPY - 1) / MPY : 0; /* :585-589 */
    const int32_t expected_len = 1 + num_working_years + ry;                                     /* :902 */
    /* the reference's growing lists; one spare slot so an over-long list could be seen (:917) */
    double* traj = (double*)malloc(sizeof(double) * (size_t)(expected_len + 2));
    double* px = (double*)malloc(sizeof(double) * (size_t)(expected_len + 2));
    double* wr = (double*)malloc(sizeof(double) * (size_t)(ry + 1));
    int32_t n_traj = 0, n_wr = 0;
    traj[n_traj] = p->initial_balance; px[n_traj] = 1.0; n_traj++; /* :490-492 */
    double years_to_ruin = NAN;                                    /* :497 */
    int ruin_bin = -1;

    double bal1 = p->initial_balance * alloc1; /* :499 */
    double bal2 = p->initial_balance - bal1;   /* :500 */
    double cb1 = bal1, cb2 = bal2;             /* :501-502 */
    double contrib = p->monthly_contribution;  /* :504 */
    double gacc1 = 0.0, gacc2 = 0.0;           /* :505-506 */
    double infl = 1.0;                         /* :508 */
    int32_t shock_idx = 0;                     /* :509 */
    int pre_fail = 0;                          /* :510 */

    for (int32_t m_idx = 1; m_idx <= working_months; ++m_idx) { /* :513 */
        if ((m_idx - 1) % MPY == 0 && m_idx > 1) {              /* :514 */
            if (p->contribution_growth_rate_annual > 0)         /* :516 */
                contrib *= 1 + p->contribution_growth_rate_annual; /* :517 */
        }
        const double* z = shocks + 3 * (size_t)shock_idx; /* :519 */
        shock_idx++;
        double g1 = orc_monthly_gross(p->inv1_mu_log, p->inv1_sigma_log, z[0]);     /* :522-524 */
        double ginf = orc_monthly_gross(p->inf_mu_log, p->inf_sigma_log, z[1]);     /* :525-527 */
        double gprem = orc_monthly_gross(p->prem_mu_log, p->prem_sigma_log, z[2]);  /* :528-530 */
        double g2 = ginf * gprem;                                                   /* :532 */
        gacc1 += bal1 * (g1 - 1.0); /* :534 */
        gacc2 += bal2 * (g2 - 1.0); /* :535 */
        bal1 *= g1;                 /* :536 */
        bal2 *= g2;                 /* :537 */
        infl *= ginf;               /* :538 */
        double c1 = contrib * alloc1; /* :540-542 */
        double c2 = contrib - c1;     /* :543 */
        bal1 += c1; cb1 += c1; bal2 += c2; cb2 += c2; /* :544-547 */
        orc_rebalance(p, &bal1, &cb1, &bal2, &cb2);   /* :549-553 */
        if (m_idx % MPY == 0) {                       /* :557 */
            int tf = orc_annual_tax(p, &bal1, &cb1, &bal2, &cb2, gacc1, gacc2); /* :558-571 */
            if (tf) pre_fail = 1;                                               /* :572-573 */
            traj[n_traj] = bal1 + bal2; px[n_traj] = infl; n_traj++;            /* :574-576 */
            gacc1 = 0.0; gacc2 = 0.0;                                           /* :578-579 */
        }
    }
    double start_balance = bal1 + bal2; /* :581 */
    double infl_ret = infl;             /* :582 */
    if (working_months > 0 && working_months % MPY != 0) { /* :590-594 */
        traj[n_traj] = start_balance; px[n_traj] = infl_ret; n_traj++;
    }
    /* :602-621 per-stream start month / duration; nominal_fixed "None" tracked by a flag */
    const size_t ns = (size_t)(p->n_streams > 0 ? p->n_streams : 1);   /* the list has any length (config.py:99) */
    int32_t* s_start = (int32_t*)malloc(sizeof(int32_t) * ns);
    int32_t* s_dur = (int32_t*)malloc(sizeof(int32_t) * ns);
    int* s_fixed_set = (int*)malloc(sizeof(int) * ns);
    double* s_fixed = (double*)malloc(sizeof(double) * ns);
    for (int32_t s = 0; s < p->n_streams; ++s) {
        s_start[s] = orc_stream_start_month_index(p->current_age, working_months, orc_stream(p, s)->start_at_age);
        s_dur[s] = orc_stream(p, s)->duration_years < 0 ? -1 : orc_stream(p, s)->duration_years * MPY;
        s_fixed_set[s] = 0; s_fixed[s] = 0.0;
    }
    double fy_gross = 0.0, fy_real = 0.0; /* :623-624 */
    int succeeded = !pre_fail;            /* :627 */
    if (pre_fail) { years_to_ruin = 0.0; ruin_bin = 0; } /* :628-629 */

    for (int32_t year_num = 0; year_num < ry; ++year_num) { /* :632 */
        if (pre_fail) break;                                /* :633-634 */
        double tg1 = 0.0, tg2 = 0.0, treal = 0.0;           /* :635-637 */
        int yfail = 0;                                      /* :638 */
        int32_t rmi = 0;
        for (int32_t mi = 0; mi < MPY; ++mi) {              /* :640 */
            rmi = year_num * MPY + mi;                      /* :641-643 */
            double price = infl;                            /* :644 */
            double expenses = p->monthly_expenses * price;  /* :645-647 */
            double income = 0.0;                            /* :649 */
            for (int32_t s = 0; s < p->n_streams; ++s) {    /* :650 */
                int active = rmi >= s_start[s] && (s_dur[s] < 0 || rmi < s_start[s] + s_dur[s]); /* :653-656 */
                if (!active) continue;
                const mcr_stream* st = orc_stream(p, s);
                double nominal;
                if (st->inflation_indexed) {
                    nominal = st->monthly_amount_today * price; /* :661-665 */
                } else {
                    if (!s_fixed_set[s]) { s_fixed[s] = st->monthly_amount_today * price; s_fixed_set[s] = 1; } /* :667-671 */
                    nominal = s_fixed[s];
                }
                income += nominal * (1.0 - st->tax_rate); /* :675-677 */
            }
            double need = pymax(0.0, expenses - income); /* :679-682 */
            double tot_before = bal1 + bal2;             /* :684 */
            if (tot_before <= EPS && need > EPS) { yfail = 1; break; } /* :685-690 */
            int32_t si = shock_idx < shock_rows - 1 ? shock_idx : shock_rows - 1; /* :692 */
            const double* z = shocks + 3 * (size_t)si;
            shock_idx++;
            double g1 = orc_monthly_gross(p->inv1_mu_log, p->inv1_sigma_log, z[0]);    /* :695-697 */
            double ginf = orc_monthly_gross(p->inf_mu_log, p->inf_sigma_log, z[1]);    /* :698-700 */
            double gprem = orc_monthly_gross(p->prem_mu_log, p->prem_sigma_log, z[2]); /* :701-703 */
            double g2 = ginf * gprem;                                                  /* :704 */
            gacc1 += bal1 * (g1 - 1.0); /* :706-708 */
            gacc2 += bal2 * (g2 - 1.0); /* :709-711 */
            bal1 *= g1; bal2 *= g2; infl *= ginf; /* :712-714 */
            double tot_after = bal1 + bal2;       /* :715 */
            if (tot_after <= EPS && need > EPS) { /* :717-724 */
                bal1 = pymax(0, bal1); bal2 = pymax(0, bal2);
                yfail = 1; break;
            }
            double cap1 = orc_nlv(bal1, cb1, use1, rate1); /* :726-731 */
            double cap2 = orc_nlv(bal2, cb2, use2, rate2); /* :732-737 */
            double cap = cap1 + cap2;                      /* :738 */
            double target = pymax(0.0, pymin(need, cap));  /* :739-742 */
            if (need > EPS && target < need - EPS) yfail = 1; /* :743-748 */
            double prop1 = cap > EPS ? cap1 / cap : alloc1;   /* :750-754 */
            double prop2 = 1.0 - prop1;                       /* :755 */
            double gw1, nw1, gw2, nw2;
            orc_withdraw(bal1, cb1, target * prop1, use1, rate1, &bal1, &cb1, &gw1, &nw1); /* :757-765 */
            tg1 += gw1;                                                                    /* :766 */
            orc_withdraw(bal2, cb2, target * prop2, use2, rate2, &bal2, &cb2, &gw2, &nw2); /* :768-776 */
            tg2 += gw2;                                                                    /* :777 */
            treal += (gw1 + gw2) * infl_ret / pymax(price, EPS);                           /* :778-782 */
            double net_cash = nw1 + nw2;                                                   /* :784 */
            if (need > EPS && net_cash < need - EPS) yfail = 1;                            /* :785-790 */
            orc_rebalance(p, &bal1, &cb1, &bal2, &cb2);                                    /* :792-796 */
            int32_t absm = working_months + rmi + 1;                                       /* :798-800 */
            if (!yfail && absm % MPY == 0) {                                               /* :801-804 */
                int tf = orc_annual_tax(p, &bal1, &cb1, &bal2, &cb2, gacc1, gacc2);        /* :805-818 */
                gacc1 = 0.0; gacc2 = 0.0;                                                  /* :819-820 */
                if (tf) yfail = 1;                                                         /* :821-822 */
            }
            if (yfail) { years_to_ruin = (double)(rmi + 1) / (double)MPY; break; }         /* :824-828 */
        }
        double ygw = tg1 + tg2; /* :830-832 */
        double wr_pct = start_balance > EPS ? (treal / start_balance) * 100.0 : 0.0; /* :834-840 */
        if (yfail) {                                                                 /* :842 */
            succeeded = 0;                                                           /* :843 */
            if (isnan(years_to_ruin)) years_to_ruin = (double)(rmi + 1) / (double)MPY; /* :844-847 */
            ruin_bin = 1 + year_num;
            traj[n_traj] = pymax(0.0, bal1 + bal2); px[n_traj] = infl; n_traj++;     /* :848-849 */
            wr[n_wr++] = NAN;                                                        /* :851 */
            if (year_num == 0) { fy_gross = ygw; fy_real = treal; }                  /* :852-856 */
            break;                                                                   /* :857 */
        }
        wr[n_wr++] = wr_pct;                                    /* :859 */
        if (year_num == 0) { fy_gross = ygw; fy_real = treal; } /* :861-865 */
        traj[n_traj] = bal1 + bal2; px[n_traj] = infl; n_traj++; /* :867-868 */
    }
    int32_t total_sim_months = working_months + ry * MPY; /* :873-875 */
    if (succeeded && total_sim_months % MPY != 0) {       /* :876-879 */
        int tf = orc_annual_tax(p, &bal1, &cb1, &bal2, &cb2, gacc1, gacc2); /* :880-893 */
        if (tf) { succeeded = 0; years_to_ruin = (double)ry; ruin_bin = ry + 1; } /* :894-896 */
        if (n_traj > 0) traj[n_traj - 1] = bal1 + bal2;                           /* :897-898 */
    }
    double final_total = bal1 + bal2; /* :900 */
    if (n_traj < expected_len) {      /* :905-916 */
        double pad = !succeeded ? 0.0 : (n_traj > 0 ? traj[n_traj - 1] : 0.0);
        double last_px = n_traj > 0 ? px[n_traj - 1] : 1.0;
        while (n_traj < expected_len) { traj[n_traj] = pad; px[n_traj] = last_px; n_traj++; }
    } else if (n_traj > expected_len) { /* :917-919 */
        n_traj = expected_len;
    }
    for (int32_t t = 0; t < n_traj; ++t) { /* :928-931 */
        if (traj_out) traj_out[(int64_t)t * ts] = traj[t];
        if (real_out) real_out[(int64_t)t * ts] = px[t] > EPS ? traj[t] / px[t] : 0.0;
    }
    while (n_wr < ry) wr[n_wr++] = NAN; /* :934-935 */
    if (wr_out) for (int32_t y = 0; y < ry; ++y) wr_out[(int64_t)y * ts] = wr[y];
    /* wr_df.count(axis=1) (:1111-1113): non-NaN observations per retirement year */
    if (wr_obs_counts) for (int32_t y = 0; y < ry; ++y) if (!isnan(wr[y])) wr_obs_counts[y] += 1;

    res->start_balance = start_balance;          /* :940 */
    res->final_balance = pymax(0, final_total);  /* :941 */
    res->success = succeeded;                    /* :942 */
    res->years_to_ruin = years_to_ruin;          /* :943 */
    res->first_year_gross = fy_gross;            /* :944 */
    res->first_year_real_gross = fy_real;        /* :945 */
    res->inflation_at_retirement = infl_ret;     /* :949 */
    res->ruin_bin = ruin_bin;
    free(traj); free(px); free(wr);
    free(s_start); free(s_dur); free(s_fixed_set); free(s_fixed);
}

/*
 * Batch driver (host buffers): same contract as mcr_run_batch_host in include/mcr.h, on the CPU.
 * Follows the sequential branch of run_monte_carlo_simulations (:987-990).
 */
int orc_run_batch(const mcr_params* p, uint64_t seed, uint32_t stream_id, uint64_t path_begin,
                  uint64_t n_paths, int32_t working_months, const double* injected_shocks,
                  const mcr_outputs* out) {
    mcr_sizes sz;
    int rc = orc_query_sizes(p, working_months, &sz);
    if (rc != MCR_OK || !out) return MCR_ERR_INVALID_ARG;
    const int64_t ts = out->path_stride > 0 ? out->path_stride : (int64_t)n_paths;
    double* shocks = injected_shocks ? NULL : (double*)malloc(sizeof(double) * 3 * (size_t)sz.shock_rows);
    for (uint64_t i = 0; i < n_paths; ++i) {
        const double* sh;
        if (injected_shocks) {
            sh = injected_shocks + (size_t)i * 3 * (size_t)sz.shock_rows;
        } else {
            orc_draw_shocks(seed, stream_id, path_begin + i, sz.shock_rows, p->equity_inflation_rho, shocks);
            sh = shocks;
        }
        orc_path_result r;
        orc_single_path(p, working_months, sh, sz.shock_rows, &r,
                        out->trajectory ? out->trajectory + i : NULL,
                        out->real_trajectory ? out->real_trajectory + i : NULL,
                        out->withdrawal_rate_trajectory ? out->withdrawal_rate_trajectory + i : NULL, ts,
                        out->wr_obs_counts);
        if (out->start_balance) out->start_balance[i] = r.start_balance;
        if (out->final_balance) out->final_balance[i] = r.final_balance;
        if (out->years_to_ruin) out->years_to_ruin[i] = r.years_to_ruin;
        if (out->first_year_gross_withdrawal) out->first_year_gross_withdrawal[i] = r.first_year_gross;
        if (out->first_year_real_gross_withdrawal) out->first_year_real_gross_withdrawal[i] = r.first_year_real_gross;
        if (out->inflation_at_retirement) out->inflation_at_retirement[i] = r.inflation_at_retirement;
        if (out->success) out->success[i] = (uint8_t)r.success;
        if (out->counters) { out->counters[MCR_CTR_SUCCESS] += (uint64_t)r.success; out->counters[MCR_CTR_PATHS] += 1; }
        if (out->ruin_year_bins && r.ruin_bin >= 0) out->ruin_year_bins[r.ruin_bin] += 1;
    }
    free(shocks);
    return MCR_OK;
}
