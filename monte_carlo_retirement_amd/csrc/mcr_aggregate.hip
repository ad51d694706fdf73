// mcr_aggregate.hip — device-side aggregation over the outputs of the path kernel (gfx950).
//
// Replaces the pandas block of run_monte_carlo_simulations (reference backend/simulation.py:
// 1045-1118): per-time-point quantile bands of the yearly trajectories and the withdrawal-rate
// rows, observation counts, and the histogram of successful final balances (plotting.py:53-59).
//
// K3  row quantiles: EXACT order statistics by most-significant-digit radix select (8 bits x 8
//     passes over order-preserving 64-bit keys), all target ranks of a row found together, then
//     NumPy's `linear` interpolation arithmetic (numpy/lib/_function_base_impl.py:
//     _QuantileMethods['linear'] / _get_indexes / _lerp), which is what DataFrame.quantile(q, axis=1)
//     evaluates.  NaNs are skipped (the WR rows carry NaN by design, simulation.py:851,934-935).
//     HBM-bound: each pass streams the [rows][n] slab once, 8 B/element, coalesced.
// K2  cohort min/max + np.histogram-style equal-width bins over the successful cohort.
#include <hip/hip_runtime.h>

#include <cstring>

#include "../../include/mcr.h"
#include "mcr_host.h"

namespace mcr {

constexpr int kRqMaxQ = 16;            // quantiles per call
constexpr int kRqMaxT = 2 * kRqMaxQ;   // order statistics per row (lower/upper neighbour of each)
constexpr int kRqBlock = 256;

struct RqRow {  // per-row selection state, lives in the caller's scratch buffer
    unsigned long long n_valid;          // non-NaN entries (= wr_df.count(axis=1) for WR rows)
    unsigned long long nan_count;
    unsigned long long rank[kRqMaxT];    // remaining rank of target t inside its group's prefix
    unsigned long long prefix[kRqMaxT];  // key prefix of group g (high 8*pass bits)
    double value[kRqMaxT];               // selected order statistic of target t (after the last pass)
    double gamma[kRqMaxQ];               // interpolation weight of quantile j
    int group_of[kRqMaxT];
    int n_targets, n_groups;
};

struct RqArgs {
    double q[kRqMaxQ];
    int n_q;
};

__device__ __forceinline__ unsigned long long key_of(double x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return b ^ ((b >> 63) ? ~0ull : 0x8000000000000000ull);  // order-preserving for non-NaN doubles
}
__device__ __forceinline__ double value_of(unsigned long long k) {
    const unsigned long long b = k ^ ((k >> 63) ? 0x8000000000000000ull : ~0ull);
    return __longlong_as_double((long long)b);
}

// One histogram pass: hist[row][group][digit] += #elements whose key has the group's prefix.
template <bool FIRST>
__global__ __launch_bounds__(kRqBlock) void rq_hist_kernel(const double* __restrict__ rows, int64_t row_stride,
                                                          int64_t n, int pass, RqRow* st, unsigned int* hist) {
    __shared__ unsigned int lh[kRqMaxT * 256];
    __shared__ unsigned long long lpref[kRqMaxT];
    __shared__ unsigned int lnan;
    const int row = blockIdx.y;
    const int G = FIRST ? 1 : st[row].n_groups;
    for (int k = threadIdx.x; k < G * 256; k += kRqBlock) lh[k] = 0u;
    if (threadIdx.x < G) lpref[threadIdx.x] = FIRST ? 0ull : st[row].prefix[threadIdx.x];
    if (threadIdx.x == 0) lnan = 0u;
    __syncthreads();
    const double* r = rows + (int64_t)row * row_stride;
    const int shift_digit = 56 - 8 * pass;
    unsigned int my_nan = 0;
    for (int64_t i = (int64_t)blockIdx.x * kRqBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kRqBlock) {
        const double x = r[i];
        if (x != x) { my_nan += FIRST ? 1u : 0u; continue; }
        const unsigned long long k = key_of(x);
        const unsigned int digit = (unsigned int)(k >> shift_digit) & 0xFFu;
        if (FIRST) {
            atomicAdd(&lh[digit], 1u);
        } else {
            const unsigned long long hk = k >> (shift_digit + 8);
            for (int g = 0; g < G; ++g)
                if (hk == lpref[g]) atomicAdd(&lh[g * 256 + digit], 1u);
        }
    }
    if (FIRST && my_nan) atomicAdd(&lnan, my_nan);
    __syncthreads();
    unsigned int* gh = hist + (size_t)row * kRqMaxT * 256;
    for (int k = threadIdx.x; k < G * 256; k += kRqBlock)
        if (lh[k]) atomicAdd(&gh[k], lh[k]);
    if (FIRST && threadIdx.x == 0 && lnan) atomicAdd(&st[row].nan_count, (unsigned long long)lnan);
}

// Per row: advance every target by one digit, regroup, clear the histograms; after the last pass
// interpolate (NumPy `linear`) and write the quantiles.
__global__ __launch_bounds__(64) void rq_scan_kernel(int64_t n, int pass, RqRow* st, unsigned int* hist,
                                                    const RqArgs args, double* out, unsigned long long* counts) {
    __shared__ unsigned long long new_prefix[kRqMaxT];
    const int row = blockIdx.x;
    RqRow& S = st[row];
    unsigned int* gh = hist + (size_t)row * kRqMaxT * 256;
    const int t = threadIdx.x;
    if (pass == 0 && t == 0) {
        const unsigned long long m = (unsigned long long)n - S.nan_count;
        S.n_valid = m;
        if (counts) counts[row] = m;
        int nt = 0;
        if (m > 0) {
            for (int j = 0; j < args.n_q; ++j) {
                // _QuantileMethods['linear'].get_virtual_index = (n - 1) * quantiles
                const double q = args.q[j];
                const double vi = (double)(m - 1) * q;
                double prev = floor(vi), next = prev + 1.0;          // _get_indexes
                if (vi >= (double)(m - 1)) { prev = (double)(m - 1); next = prev; }
                if (vi < 0.0) { prev = 0.0; next = 0.0; }
                S.gamma[j] = vi - floor(vi);                          // _get_gamma (linear: unchanged)
                S.rank[nt] = (unsigned long long)prev; S.group_of[nt] = 0; ++nt;
                S.rank[nt] = (unsigned long long)next; S.group_of[nt] = 0; ++nt;
            }
        }
        S.n_targets = nt;
        S.n_groups = 1;
        S.prefix[0] = 0ull;
    }
    __syncthreads();
    const int nt = S.n_targets;
    if (t < nt) {
        const int g = S.group_of[t];
        const unsigned int* h = gh + g * 256;
        unsigned long long rank = S.rank[t], cum = 0;
        int d = 0;
        for (; d < 255; ++d) {
            const unsigned long long c = h[d];
            if (rank < cum + c) break;
            cum += c;
        }
        S.rank[t] = rank - cum;
        new_prefix[t] = (S.prefix[g] << 8) | (unsigned long long)d;
    }
    __syncthreads();
    const int old_groups = S.n_groups;
    __syncthreads();
    if (t == 0) {  // regroup: targets that still share a prefix share a histogram
        int ng = 0;
        for (int a = 0; a < nt; ++a) {
            int g = -1;
            for (int b = 0; b < ng; ++b)
                if (S.prefix[b] == new_prefix[a]) { g = b; break; }
            if (g < 0) { g = ng++; S.prefix[g] = new_prefix[a]; }
            S.group_of[a] = g;
        }
        S.n_groups = ng > 0 ? ng : 1;
    }
    for (int k = t; k < old_groups * 256; k += 64) gh[k] = 0u;  // ready for the next pass / call
    __syncthreads();
    if (pass == 7) {
        if (t < nt) S.value[t] = value_of(new_prefix[t]);
        __syncthreads();
        if (t < args.n_q) {
            double r;
            if (S.n_valid == 0) {
                r = __longlong_as_double(0x7ff8000000000000LL);  // all-NaN row -> NaN (pandas na_value)
            } else {
                const double a = S.value[2 * t], b = S.value[2 * t + 1], g = S.gamma[t];
                const double diff = b - a;                       // _lerp
                r = a + diff * g;
                if (g >= 0.5) r = b - diff * (1.0 - g);
            }
            out[(size_t)row * args.n_q + t] = r;
        }
    }
}

__global__ void rq_init_kernel(RqRow* st, unsigned int* hist, int n_rows) {
    const size_t total = (size_t)n_rows * kRqMaxT * 256;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (size_t)gridDim.x * blockDim.x)
        hist[k] = 0u;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_rows) { st[r].nan_count = 0ull; st[r].n_valid = 0ull; st[r].n_targets = 0; st[r].n_groups = 1; st[r].prefix[0] = 0ull; }
}

// ---- K2: successful-cohort min/max and equal-width histogram ------------------------------------
__device__ __forceinline__ void atomic_min_f64(double* addr, double v) {  // v is non-NaN
    unsigned long long* a = (unsigned long long*)addr;
    unsigned long long old = *a;
    while (__longlong_as_double((long long)old) > v) {
        const unsigned long long assumed = old;
        old = atomicCAS(a, assumed, (unsigned long long)__double_as_longlong(v));
        if (old == assumed) break;
    }
}
__device__ __forceinline__ void atomic_max_f64(double* addr, double v) {
    unsigned long long* a = (unsigned long long*)addr;
    unsigned long long old = *a;
    while (__longlong_as_double((long long)old) < v) {
        const unsigned long long assumed = old;
        old = atomicCAS(a, assumed, (unsigned long long)__double_as_longlong(v));
        if (old == assumed) break;
    }
}

__global__ __launch_bounds__(256) void minmax_init_kernel(double* minmax) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        minmax[0] = __longlong_as_double(0x7ff0000000000000LL);   // +inf
        minmax[1] = __longlong_as_double((long long)0xfff0000000000000ULL);  // -inf
    }
}

__global__ __launch_bounds__(256) void minmax_kernel(const double* __restrict__ v, const uint8_t* __restrict__ ok,
                                                    int64_t n, double* minmax) {
    __shared__ double smin[4], smax[4];
    double lo = __longlong_as_double(0x7ff0000000000000LL), hi = -lo;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        if (ok[i]) { const double x = v[i]; lo = fmin(lo, x); hi = fmax(hi, x); }
    }
    for (int off = 32; off > 0; off >>= 1) {
        lo = fmin(lo, __shfl_down(lo, off, 64));
        hi = fmax(hi, __shfl_down(hi, off, 64));
    }
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        lo = fmin(fmin(smin[0], smin[1]), fmin(smin[2], smin[3]));
        hi = fmax(fmax(smax[0], smax[1]), fmax(smax[2], smax[3]));
        if (lo <= hi) { atomic_min_f64(&minmax[0], lo); atomic_max_f64(&minmax[1], hi); }
    }
}

// np.histogram(x, bins=n_bins) over [lo, hi] (numpy/lib/_histograms_impl.py, uniform-bin fast path):
// index = int((x - lo) / (hi - lo) * n_bins), the right edge belongs to the last bin, then a
// +-1 correction against the linspace edges.
__global__ __launch_bounds__(256) void hist_kernel(const double* __restrict__ v, const uint8_t* __restrict__ ok,
                                                  int64_t n, const double* __restrict__ minmax, int n_bins,
                                                  unsigned long long* bins) {
    extern __shared__ unsigned int lbins[];
    for (int k = threadIdx.x; k < n_bins; k += 256) lbins[k] = 0u;
    __syncthreads();
    double lo = minmax[0], hi = minmax[1];
    if (lo == hi) { lo = lo - 0.5; hi = hi + 0.5; }  // _get_outer_edges: degenerate range widened
    const double denom = hi - lo;
    const double step = denom / (double)n_bins;      // np.linspace step
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        if (!ok[i]) continue;
        const double x = v[i];
        if (!(x >= lo && x <= hi)) continue;
        int idx = (int)(((x - lo) / denom) * (double)n_bins);
        if (idx == n_bins) idx -= 1;
        const double e_lo = idx == n_bins ? hi : lo + (double)idx * step;
        if (x < e_lo) idx -= 1;
        else {
            const double e_hi = (idx + 1) == n_bins ? hi : lo + (double)(idx + 1) * step;
            if (x >= e_hi && idx != n_bins - 1) idx += 1;
        }
        atomicAdd(&lbins[idx], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < n_bins; k += 256)
        if (lbins[k]) atomicAdd(&bins[k], (unsigned long long)lbins[k]);
}

static int grid_for(int64_t n, int per_block, int cap) {
    int64_t b = (n + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

}  // namespace mcr

using namespace mcr;

extern "C" {

int64_t mcr_row_quantiles_scratch_bytes(int32_t n_rows, int32_t n_q) {
    if (n_rows <= 0 || n_q <= 0 || n_q > kRqMaxQ) return 0;
    return (int64_t)n_rows * (int64_t)(sizeof(RqRow) + (size_t)kRqMaxT * 256 * sizeof(unsigned int));
}

int mcr_row_quantiles(const double* rows, int64_t row_stride, int32_t n_rows, int64_t n, const double* q,
                      int32_t n_q, double* out, uint64_t* counts, void* scratch, int device, void* hip_stream) {
    int rc = use_device(device);
    if (rc != MCR_OK) return rc;
    if (!rows || !q || !out || !scratch || n_rows <= 0 || n <= 0 || n_q <= 0) { set_error("bad arguments"); return MCR_ERR_INVALID_ARG; }
    if (n_q > kRqMaxQ) { set_error("at most %d quantiles per call", kRqMaxQ); return MCR_ERR_INVALID_ARG; }
    if (row_stride < n) { set_error("row_stride < n"); return MCR_ERR_INVALID_ARG; }
    if (n >= ((int64_t)1 << 32)) { set_error("n must be < 2^32 per row"); return MCR_ERR_INVALID_ARG; }
    if (n_rows > 65535) { set_error("too many rows"); return MCR_ERR_INVALID_ARG; }
    for (int j = 0; j < n_q; ++j)
        if (!(q[j] >= 0.0 && q[j] <= 1.0)) { set_error("quantile %d out of [0,1]", j); return MCR_ERR_INVALID_ARG; }
    hipStream_t s = (hipStream_t)hip_stream;
    RqRow* st = (RqRow*)scratch;
    unsigned int* hist = (unsigned int*)((char*)scratch + (size_t)n_rows * sizeof(RqRow));
    RqArgs a;
    std::memset(&a, 0, sizeof(a));
    a.n_q = n_q;
    for (int j = 0; j < n_q; ++j) a.q[j] = q[j];
    hipLaunchKernelGGL(rq_init_kernel, dim3(grid_for((int64_t)n_rows * kRqMaxT * 256, 256, 1024)), dim3(256), 0, s, st, hist, n_rows);
    int bx = grid_for(n, kRqBlock * 8, 4096 / n_rows > 0 ? 4096 / n_rows : 1);
    if (bx < 1) bx = 1;
    const dim3 grid(bx, n_rows), block(kRqBlock);
    for (int pass = 0; pass < 8; ++pass) {
        if (pass == 0) hipLaunchKernelGGL(rq_hist_kernel<true>, grid, block, 0, s, rows, row_stride, n, pass, st, hist);
        else hipLaunchKernelGGL(rq_hist_kernel<false>, grid, block, 0, s, rows, row_stride, n, pass, st, hist);
        hipLaunchKernelGGL(rq_scan_kernel, dim3(n_rows), dim3(64), 0, s, n, pass, st, hist, a, out, (unsigned long long*)counts);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "row quantile kernels");
    return MCR_OK;
}

int mcr_minmax_success(const double* values, const uint8_t* success, int64_t n, double* minmax, int device, void* hip_stream) {
    int rc = use_device(device);
    if (rc != MCR_OK) return rc;
    if (!values || !success || !minmax || n < 0) { set_error("bad arguments"); return MCR_ERR_INVALID_ARG; }
    hipStream_t s = (hipStream_t)hip_stream;
    hipLaunchKernelGGL(minmax_init_kernel, dim3(1), dim3(256), 0, s, minmax);
    if (n > 0) hipLaunchKernelGGL(minmax_kernel, dim3(grid_for(n, 256 * 8, 2048)), dim3(256), 0, s, values, success, n, minmax);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "minmax kernels");
    return MCR_OK;
}

int mcr_histogram_success(const double* values, const uint8_t* success, int64_t n, const double* minmax,
                          int32_t n_bins, uint64_t* bins, int device, void* hip_stream) {
    int rc = use_device(device);
    if (rc != MCR_OK) return rc;
    if (!values || !success || !minmax || !bins || n < 0 || n_bins <= 0 || n_bins > 8192) { set_error("bad arguments"); return MCR_ERR_INVALID_ARG; }
    if (n == 0) return MCR_OK;
    hipLaunchKernelGGL(hist_kernel, dim3(grid_for(n, 256 * 8, 2048)), dim3(256), (size_t)n_bins * sizeof(unsigned int),
                       (hipStream_t)hip_stream, values, success, n, minmax, (int)n_bins, (unsigned long long*)bins);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "hist_kernel");
    return MCR_OK;
}

}  // extern "C"
