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
//     Large rows avoid most of those passes (mcr_row_quantiles, single GPU), in FIVE launches: (1) the first 4096
//     entries of every row are sorted in LDS -> coarse key brackets around every quantile; (2) a counting pass over
//     a SAMPLE (the first n/32 entries) tallies the keys below / inside every coarse bracket, with a 1024-bin
//     sub-histogram inside each; (3) from those counts, a fine bracket per quantile that contains the wanted
//     order statistics of the WHOLE row with overwhelming probability; (4) ONE pass over the slab counts the keys
//     below / inside every fine bracket (again with sub-histograms) and compacts the few percent inside;
//     (5) per row, the counts locate every target in one sub-bin ("cell", a few hundred keys), the candidates of
//     the wanted cells are collected in LDS and selected there, and the quantiles are interpolated.  Exactness
//     never depends on the samples: a row whose counts show a target outside its bracket (or whose candidates
//     overflow: giant ties) simply takes the full radix passes.  Traffic: ~1.1 reads of the slab instead of 4.
// K2  cohort min/max + np.histogram-style equal-width bins over the successful cohort.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "../../include/mcr.h"
#include "mcr_host.h"

namespace mcr {

constexpr int kRqMaxQ = 16;            // quantiles per call
constexpr int kRqMaxT = 2 * kRqMaxQ;   // order statistics per row (lower/upper neighbour of each)
constexpr int kRqBlock = 256;
constexpr int kRqScanBlock = 256;    // rq_scan_kernel: one workgroup per row (targets use the first <= 32 threads)

struct RqRow {  // per-row selection state, lives in the caller's scratch buffer
    unsigned long long n_valid;          // non-NaN entries (= wr_df.count(axis=1) for WR rows)
    unsigned long long rank[kRqMaxT];    // remaining rank of target t inside its group's prefix
    unsigned long long prefix[kRqMaxT];  // key prefix of group g (high 8*pass bits)
    double value[kRqMaxT];               // selected order statistic of target t (after the last pass)
    double gamma[kRqMaxQ];               // interpolation weight of quantile j
    int group_of[kRqMaxT];
    int n_targets, n_groups;
    unsigned int cand_count;             // keys appended to the (rank-local) candidate buffer in pass 3
    unsigned int const_row;              // 1: every non-NaN entry of the row is the same value (done after pass 0)
    unsigned long long kmin, kmax;       // min / max key of the row (pass 0; only when the shortcut is enabled)
};

constexpr unsigned long long kRqKeyPosInf = 0xFFF0000000000000ull;   // key_of(+inf): the largest non-NaN key
constexpr double kRqBracketSigmas = 4.5;   // half-width of a bracket in binomial standard deviations of the sample rank: a side misses with
                                           // probability 3.4e-6 (one row in ~150 calls of 136 rows x 7 quantiles takes the radix route); 5.0 cost 10 % more candidates

// Disjoint, ascending key intervals of one row (from its sample) and what the bracket pass counted for them.
struct RqBracket {
    unsigned long long lo[kRqMaxQ], hi[kRqMaxQ];   // closed key intervals [lo, hi]
    // bracket pass: # non-NaN keys by position among the sorted bounds lo0 <= hi0 < lo1 <= hi1 < ...:
    // position 2b+1 = inside interval b, position 2b = in the gap below it (2*n_intervals = above the last)
    unsigned long long pos_count[2 * kRqMaxQ + 1];
    unsigned long long n_nan;
    int interval_of_q[kRqMaxQ];
    int shift[kRqMaxQ];                            // sub-bin of a key inside interval b: (key - lo[b]) >> shift[b], or, if < 0,
    double xlo[kRqMaxQ], inv_w[kRqMaxQ];           //   min(bins - 1, (int)((x - xlo[b]) * inv_w[b])): equal-width bins in VALUE
    int n_intervals;
    unsigned int cand_count;                       // values appended to the row's candidate buffer
    unsigned int fallback;                         // 1: the row takes the full radix passes
    unsigned int open_lo, open_hi;                 // coarse level: bit j = quantile j's window ran off the first sample
    // bucket table of the slab pass (rq_slab_kernel): bucket(x) = clamp((int)fma(x, lut_s, lut_c), 0, kRqLutSize - 1), an equal-width
    // grid in VALUE over the span of the bounds; lut_ok = 1 when no bucket holds more than two of the row's bounds
    double lut_s, lut_c;
    unsigned int lut_ok;
    unsigned int lut_pad;
};
constexpr int kRqLutSize = 2048;
constexpr int kRqTiny = 4096;              // first sample: sorted in LDS by one workgroup per row
constexpr double kRqCoarseSigmas = 7.0;    // its brackets leave room for the second sample's own 5-sigma window
constexpr int kRqMaxSubBins = 1024;        // sub-histogram bins per interval (stride of the global histograms)
constexpr int kRqCoarseSubBits = 10;       // sample pass: 1024 bins per coarse interval
constexpr int kRqWaves = 16;               // waves of the per-row kernels (1024 threads)
constexpr int kRqWaveStage = 128;          // candidates a wave of the slab pass stages in LDS between two appends
constexpr unsigned int kRqListCap = 12288; // keys of all wanted cells of one row (LDS of rq_select_kernel: 96 KB)
// Scratch layout: RqRow[n_rows] | hist u32[n_rows][kRqMaxT][256] | aux u32[n_rows][2] | cand u64[n_rows][cap].
// hist|aux is ONE dense block of 32-bit counters: a multi-GPU caller sums it across ranks after every
// histogram step (aux[r][0] = NaN count, aux[r][1] = #ranks whose candidate buffer overflowed).

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

// Wave-aggregated LDS histogram update.  In the first digit passes nearly every lane of a wave hits
// the same bin (balances of one year share sign/exponent), and 64 same-address LDS atomics serialise;
// so the two most common (group, digit) keys of the wave are peeled with a ballot + one atomic each,
// and only the remaining lanes issue individual atomics.
__device__ __forceinline__ void hist_add_aggregated(unsigned int* lh, bool active, unsigned int key) {
    unsigned long long todo = __ballot(active);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        if (todo == 0ull) return;
        const int lead = __ffsll((long long)todo) - 1;
        const unsigned int k0 = (unsigned int)__shfl((int)key, lead, 64);
        const unsigned long long same = __ballot(active && key == k0) & todo;
        if ((int)(threadIdx.x & 63) == lead) atomicAdd(&lh[k0], (unsigned int)__popcll(same));
        todo &= ~same;
    }
    if (active && ((todo >> (threadIdx.x & 63)) & 1ull)) atomicAdd(&lh[key], 1u);
}

// One histogram pass: hist[row][group][digit] += #elements whose key has the group's prefix.
// PASS3 additionally compacts the matching keys of the row into cand[row][..] (the later passes
// then stream only those).  16-byte loads (two doubles per lane) when the row allows it.
template <bool FIRST, bool COMPACT>
__global__ __launch_bounds__(kRqBlock) void rq_hist_kernel(const double* __restrict__ rows, int64_t row_stride,
                                                          int64_t n, int pass, RqRow* st, unsigned int* hist,
                                                          unsigned int* aux, unsigned long long* cand,
                                                          unsigned int cand_cap, int only_overflowed, int track_minmax,
                                                          const unsigned int* __restrict__ row_list) {
    extern __shared__ __align__(16) unsigned int lh[];  // [max groups of this call][256], sized by the host
    __shared__ unsigned long long lpref[kRqMaxT];
    __shared__ unsigned int lnan;
    constexpr int kStage = 1024;  // candidate keys staged per workgroup before ONE global slot reservation
    __shared__ unsigned long long stage[COMPACT ? kStage : 1];
    __shared__ unsigned int stage_n, stage_base;
    const int row = row_list ? (int)row_list[blockIdx.y] : (int)blockIdx.y;   // a call may concern a list of rows only
    if (only_overflowed && !aux[2 * row + 1]) return;  // slow path only for rows whose candidates overflowed
    if (!FIRST && st[row].const_row) return;           // all-equal row: finished after pass 0
    const int G = FIRST ? 1 : st[row].n_groups;
    for (int k = threadIdx.x; k < G * 256; k += kRqBlock) lh[k] = 0u;
    if ((int)threadIdx.x < G) lpref[threadIdx.x] = FIRST ? 0ull : st[row].prefix[threadIdx.x];
    if (threadIdx.x == 0) { lnan = 0u; stage_n = 0u; }
    __syncthreads();
    const double* r = rows + (int64_t)row * row_stride;
    const int shift_digit = 56 - 8 * pass;
    unsigned int my_nan = 0;
    unsigned long long my_min = ~0ull, my_max = 0ull;  // FIRST + track_minmax: this lane's key range
    const bool vec2 = ((row_stride & 1) == 0) && ((reinterpret_cast<uintptr_t>(rows) & 15) == 0);
    const int64_t n_pairs = vec2 ? n / 2 : 0;
    unsigned long long* crow = COMPACT ? cand + (size_t)row * cand_cap : nullptr;

    auto consume = [&](double x, bool in_range) {
        const bool isnan_x = x != x;
        if (FIRST && in_range && isnan_x) ++my_nan;
        const unsigned long long k = key_of(x);
        const unsigned int digit = (unsigned int)(k >> shift_digit) & 0xFFu;
        bool active = in_range && !isnan_x;
        if (FIRST && active) { my_min = k < my_min ? k : my_min; my_max = k > my_max ? k : my_max; }
        unsigned int key = digit;
        if (!FIRST) {
            const unsigned long long hk = k >> (shift_digit + 8);
            int g = -1;
            for (int j = 0; j < G; ++j)
                if (hk == lpref[j]) g = j;   // groups have distinct prefixes: at most one matches
            active = active && g >= 0;
            key = (unsigned int)(g < 0 ? 0 : g) * 256u + digit;
        }
        hist_add_aggregated(lh, active, key);
        if (COMPACT) {  // wave-aggregated append of the matching keys into the workgroup's LDS stage
            const unsigned long long m = __ballot(active);
            if (m) {
                const int lane = threadIdx.x & 63;
                const int lead = __ffsll((long long)m) - 1;
                unsigned int base = 0;
                if (lane == lead) base = atomicAdd(&stage_n, (unsigned int)__popcll(m));
                base = (unsigned int)__shfl((int)base, lead, 64);
                if (active) {
                    const unsigned int slot = base + (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
                    if (slot < (unsigned int)kStage) {
                        stage[slot] = k;
                    } else {  // stage full (heavy ties): reserve a global slot directly
                        const unsigned int gslot = atomicAdd(&st[row].cand_count, 1u);
                        if (gslot < cand_cap) crow[gslot] = k;
                    }
                }
            }
        }
    };

    // main loop: every lane of a wave takes the same number of trips (the aggregation uses ballots)
    const int64_t step = (int64_t)gridDim.x * kRqBlock;
    if (vec2) {
        typedef double d2_t __attribute__((ext_vector_type(2)));
        const d2_t* r2 = reinterpret_cast<const d2_t*>(r);
        const int64_t trips = (n_pairs + step - 1) / step;
        constexpr int kUnroll = 4;  // four 16-byte loads in flight per lane: HBM latency needs the bytes
        for (int64_t t = 0; t < trips; t += kUnroll) {
            d2_t v[kUnroll];
            bool ok[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const int64_t i = (t + u) * step + (int64_t)blockIdx.x * kRqBlock + threadIdx.x;
                ok[u] = i < n_pairs;
                v[u] = ok[u] ? __builtin_nontemporal_load(&r2[i]) : d2_t{0.0, 0.0};
            }
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                consume(v[u].x, ok[u]);
                consume(v[u].y, ok[u]);
            }
        }
        if ((n & 1) && blockIdx.x == 0) consume(r[n - 1], threadIdx.x == 0);  // odd tail element
    } else {
        const int64_t trips = (n + step - 1) / step;
        for (int64_t t = 0; t < trips; ++t) {
            const int64_t i = t * step + (int64_t)blockIdx.x * kRqBlock + threadIdx.x;
            const bool ok = i < n;
            consume(ok ? r[i] : 0.0, ok);
        }
    }
    if (FIRST && my_nan) atomicAdd(&lnan, my_nan);
    __syncthreads();
    unsigned int* gh = hist + (size_t)row * kRqMaxT * 256;
    for (int k = threadIdx.x; k < G * 256; k += kRqBlock)
        if (lh[k]) atomicAdd(&gh[k], lh[k]);
    if (FIRST && threadIdx.x == 0 && lnan) atomicAdd(&aux[2 * row], lnan);
    if (FIRST && track_minmax) {  // wave reduce, then one pair of global atomics per wave
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long omin = __shfl_down(my_min, off, 64), omax = __shfl_down(my_max, off, 64);
            my_min = omin < my_min ? omin : my_min;
            my_max = omax > my_max ? omax : my_max;
        }
        if ((threadIdx.x & 63) == 0 && my_min <= my_max) { atomicMin(&st[row].kmin, my_min); atomicMax(&st[row].kmax, my_max); }
    }
    if (COMPACT) {  // flush the stage: one global reservation per workgroup, coalesced copy
        const unsigned int cnt = stage_n < (unsigned int)kStage ? stage_n : (unsigned int)kStage;
        if (threadIdx.x == 0) stage_base = cnt ? atomicAdd(&st[row].cand_count, cnt) : 0u;
        __syncthreads();
        for (unsigned int k = threadIdx.x; k < cnt; k += kRqBlock)
            if (stage_base + k < cand_cap) crow[stage_base + k] = stage[k];
    }
}

// Passes 4..7 over the compacted candidate keys of one row (one workgroup per row; the candidates
// are a few thousand keys, L2-resident).  Rows whose candidates overflowed are left to the slow path.
__global__ __launch_bounds__(kRqBlock) void rq_cand_hist_kernel(int pass, RqRow* st, unsigned int* hist,
                                                               const unsigned int* aux,
                                                               const unsigned long long* cand, unsigned int cand_cap,
                                                               const unsigned int* __restrict__ row_list) {
    extern __shared__ __align__(16) unsigned int lh[];
    __shared__ unsigned long long lpref[kRqMaxT];
    const int row = row_list ? (int)row_list[blockIdx.y] : (int)blockIdx.y;
    RqRow& S = st[row];
    if (aux[2 * row + 1] || S.const_row || S.n_targets == 0) return;
    const int G = S.n_groups;
    for (int k = threadIdx.x; k < G * 256; k += kRqBlock) lh[k] = 0u;
    if ((int)threadIdx.x < G) lpref[threadIdx.x] = S.prefix[threadIdx.x];
    __syncthreads();
    const unsigned int cnt = S.cand_count;
    const unsigned long long* crow = cand + (size_t)row * cand_cap;
    const int shift_digit = 56 - 8 * pass;
    const unsigned int stride = gridDim.x * kRqBlock;
    const unsigned int trips = (cnt + stride - 1) / stride;
    for (unsigned int t = 0; t < trips; ++t) {
        const unsigned int i = t * stride + blockIdx.x * kRqBlock + threadIdx.x;
        const bool ok = i < cnt;
        const unsigned long long k = ok ? crow[i] : 0ull;
        const unsigned long long hk = k >> (shift_digit + 8);
        int g = -1;
        for (int j = 0; j < G; ++j)
            if (hk == lpref[j]) g = j;
        hist_add_aggregated(lh, ok && g >= 0, (unsigned int)(g < 0 ? 0 : g) * 256u + ((unsigned int)(k >> shift_digit) & 0xFFu));
    }
    __syncthreads();
    unsigned int* gh = hist + (size_t)row * kRqMaxT * 256;
    for (int k = threadIdx.x; k < G * 256; k += kRqBlock)
        if (lh[k]) atomicAdd(&gh[k], lh[k]);
}

// Per row: advance every target by one digit, regroup, clear the histograms; after the last pass
// interpolate (NumPy `linear`) and write the quantiles.
__global__ __launch_bounds__(kRqScanBlock) void rq_scan_kernel(int64_t n, int pass, RqRow* st, unsigned int* hist,
                                                    const RqArgs args, double* out, unsigned long long* counts,
                                                    unsigned int* aux,
                                                    const unsigned int* __restrict__ row_list) {
    __shared__ unsigned long long new_prefix[kRqMaxT], grp_prefix[kRqMaxT];
    __shared__ int grp_of[kRqMaxT], n_grp;
    extern __shared__ __align__(16) unsigned int sh[];  // this row's histograms, [n_groups][256]
    const int row = row_list ? (int)row_list[blockIdx.x] : (int)blockIdx.x;
    // the row's state is worked on in LDS (one coalesced load, one store): the serial parts below would otherwise
    // be chains of dependent global accesses
    __shared__ RqRow S;
    static_assert(sizeof(RqRow) % sizeof(unsigned int) == 0, "RqRow is copied word by word");
    constexpr int kRowWords = (int)(sizeof(RqRow) / sizeof(unsigned int));
    unsigned int* g_state = reinterpret_cast<unsigned int*>(&st[row]);
    unsigned int* l_state = reinterpret_cast<unsigned int*>(&S);
    unsigned int* gh = hist + (size_t)row * kRqMaxT * 256;
    const int t = threadIdx.x;
    for (int k = t; k < kRowWords; k += kRqScanBlock) l_state[k] = g_state[k];
    {
        // (groups beyond the row's current count hold zeros: reading up to the call's maximum needs no dependent load)
        const int ng = pass == 0 ? 1 : 2 * args.n_q;
        const uint4* gh4 = reinterpret_cast<const uint4*>(gh);   // 16-byte loads: the block is 32 KB-aligned per row
        uint4* sh4 = reinterpret_cast<uint4*>(sh);
        for (int k = t; k < ng * 64; k += kRqScanBlock) sh4[k] = gh4[k];
    }
    __syncthreads();
    auto store_state = [&]() {
        __syncthreads();
        for (int k = t; k < kRowWords; k += kRqScanBlock) g_state[k] = l_state[k];
    };
    if (pass == 0 && t == 0) {
        const unsigned long long m = (unsigned long long)n - (unsigned long long)aux[2 * row];
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
        // all non-NaN entries equal (e.g. the t = 0 rows: every path starts from the same balance): every order
        // statistic is that value — finish now; later passes skip the row (only when min/max were tracked)
        S.const_row = (m > 0 && S.kmin == S.kmax) ? 1u : 0u;
    }
    __syncthreads();
    if (S.const_row) {
        if (pass == 0) {
            for (int k = t; k < 256; k += kRqScanBlock) gh[k] = 0u;
            if (t < args.n_q) out[(size_t)row * args.n_q + t] = value_of(S.kmin);  // lerp(a, a, g) = a
            store_state();
        }
        return;
    }
    const int nt = S.n_targets;
    // each target: the digit d whose bin holds its rank, i.e. the first d with rank < h[0] + ... + h[d] (255 if none).
    // One wave per target at a time: a lane sums 4 bins, a wave scan finds the lane, that lane walks its 4 bins.
    for (int tg = t >> 6; tg < nt; tg += kRqScanBlock / 64) {
        const int lane = t & 63;
        const int g = S.group_of[tg];
        const unsigned int* h = sh + g * 256 + 4 * lane;
        const unsigned long long c0 = h[0], c1 = h[1], c2 = h[2], c3 = h[3];
        const unsigned long long rank = S.rank[tg];
        unsigned long long incl = c0 + c1 + c2 + c3;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long up = (unsigned long long)__shfl_up((long long)incl, off, 64);
            incl += lane >= off ? up : 0ull;
        }
        const unsigned long long excl = incl - (c0 + c1 + c2 + c3);
        const unsigned long long hit = __ballot(rank >= excl && rank < incl);   // at most one lane
        int d = 255;
        unsigned long long cum = 0ull;
        if (hit) {
            const int L = __ffsll((long long)hit) - 1;
            if (lane == L) {
                cum = excl;
                d = 4 * lane;
                if (rank >= cum + c0) { cum += c0; ++d; if (rank >= cum + c1) { cum += c1; ++d; if (rank >= cum + c2) { cum += c2; ++d; } } }
            }
            d = __shfl(d, L, 64);
            cum = (unsigned long long)__shfl((long long)cum, L, 64);
        } else {   // rank beyond the histogram's total: the scan ends in the last bin, below which lie all the others
            const unsigned long long total = (unsigned long long)__shfl((long long)incl, 63, 64);
            cum = total - (unsigned long long)sh[g * 256 + 255];
        }
        if (lane == 0) {
            S.rank[tg] = rank - cum;
            new_prefix[tg] = (S.prefix[g] << 8) | (unsigned long long)d;
        }
    }
    __syncthreads();
    const int old_groups = S.n_groups;
    __syncthreads();
    if (t == 0) {  // regroup: targets that still share a prefix share a histogram (worked out in LDS, stored below)
        int ng = 0;
        for (int a = 0; a < nt; ++a) {
            int g = -1;
            for (int b = 0; b < ng; ++b)
                if (grp_prefix[b] == new_prefix[a]) { g = b; break; }
            if (g < 0) { g = ng++; grp_prefix[g] = new_prefix[a]; }
            grp_of[a] = g;
        }
        n_grp = ng > 0 ? ng : 1;
    }
    __syncthreads();
    if (t < nt) S.group_of[t] = grp_of[t];
    if (t < n_grp && nt > 0) S.prefix[t] = grp_prefix[t];
    if (t == 0) S.n_groups = n_grp;
    {   // ready for the next pass / call
        uint4* gz = reinterpret_cast<uint4*>(gh);
        for (int k = t; k < old_groups * 64; k += kRqScanBlock) gz[k] = uint4{0u, 0u, 0u, 0u};
    }
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
    store_state();
}

// after the compaction pass: did this rank's candidate buffer overflow for the row?
__global__ void rq_flag_kernel(const RqRow* st, unsigned int* aux, int n_rows, unsigned int cand_cap) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_rows) aux[2 * r + 1] = st[r].cand_count > cand_cap ? 1u : 0u;
}

__global__ void rq_init_kernel(RqRow* st, unsigned int* hist, int n_rows) {
    const size_t total = (size_t)n_rows * kRqMaxT * 256 + (size_t)n_rows * 2;  // hist | aux
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (size_t)gridDim.x * blockDim.x)
        hist[k] = 0u;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_rows) {
        st[r].n_valid = 0ull; st[r].n_targets = 0; st[r].n_groups = 1; st[r].prefix[0] = 0ull;
        st[r].cand_count = 0u; st[r].const_row = 0u;
        st[r].kmin = ~0ull; st[r].kmax = 0ull;  // kmin > kmax: "not tracked" (never equal)
    }
}

// ---- K3, bracketed single pass -------------------------------------------------------------------
// Sub-bin of a key inside an interval.  Equal steps of the order-preserving KEY are (piecewise) equal steps of
// log |x|: the right resolution for one-signed data of any dynamic range, useless for an interval that reaches or
// straddles zero (almost all of its key range is magnitudes nobody has).  Those intervals use equal-width bins in
// VALUE instead.  Either map is monotone in x, which is all the counting needs.
__device__ __forceinline__ int rq_sub_bin(unsigned long long key, double x, unsigned long long lo, int shift, double xlo, double inv_w,
                                          int bins) {
    if (shift >= 0) return (int)((key - lo) >> shift);
    const int b = (int)((x - xlo) * inv_w);          // x >= xlo; the conversion saturates
    return b < bins - 1 ? b : bins - 1;
}

// The slab pass compares in the DOUBLE domain (no key conversion per element): "key(x) >= kb" for a key bound kb is
// "x >= rq_bound_value(kb)" for every non-NaN x — a bound below key(-inf) is "always" (-inf), one above key(+inf) is
// "never" (a NaN: every comparison with it is false) — except that the double order has ONE zero where the key order
// has two (key(-0.0) = key(+0.0) - 1).  The slab pass therefore treats -0.0 AS +0.0 throughout: positions come from
// double comparisons, and a candidate is canonicalised (x + 0.0) before it is binned and stored, so everything
// downstream sees the key of +0.0.  The quantiles are unchanged as numbers (-0.0 == +0.0).
constexpr unsigned long long kRqKeyNegInf = 0x000FFFFFFFFFFFFFull;   // key_of(-inf): the smallest non-NaN key
__device__ __forceinline__ double rq_bound_value(unsigned long long kb) {
    if (kb <= kRqKeyNegInf) return __longlong_as_double((long long)0xfff0000000000000ull);
    if (kb > kRqKeyPosInf) return __longlong_as_double(0x7ff8000000000000LL);
    return value_of(kb);
}
__device__ __forceinline__ int rq_lut_bucket(double x, double s, double c) {
    const int t = (int)__builtin_fma(x, s, c);          // v_cvt_i32_f64: saturates, NaN -> 0; monotone in x for s >= 0
    return t < 0 ? 0 : (t > kRqLutSize - 1 ? kRqLutSize - 1 : t);
}
__device__ __forceinline__ int rq_lut_bucket_of_bound(double bv, double s, double c) {
    return bv != bv ? kRqLutSize : rq_lut_bucket(bv, s, c);   // a "never" bound lies beyond every bucket
}

// Merge the brackets [lo_j, hi_j] of n_q quantiles (closed key intervals) into disjoint ascending intervals of one
// row and set every interval's sub-bin shift for `sub_bits`-bit sub-histograms; counters zeroed.  (One thread.)
__device__ void rq_make_intervals(RqBracket& B, const unsigned long long* lo, const unsigned long long* hi, int n_q, int sub_bits) {
    int order[kRqMaxQ];
    for (int j = 0; j < n_q; ++j) {
        int k = j;
        while (k > 0 && lo[order[k - 1]] > lo[j]) { order[k] = order[k - 1]; --k; }   // insertion sort by lo
        order[k] = j;
    }
    for (int b = 0; b < kRqMaxQ; ++b) { B.lo[b] = ~0ull; B.hi[b] = 0ull; B.interval_of_q[b] = 0; B.shift[b] = 0; }
    int nb = 0;
    for (int i = 0; i < n_q; ++i) {
        const int j = order[i];
        if (nb > 0 && lo[j] <= B.hi[nb - 1]) {
            if (hi[j] > B.hi[nb - 1]) B.hi[nb - 1] = hi[j];      // overlapping brackets share an interval
        } else {
            B.lo[nb] = lo[j]; B.hi[nb] = hi[j]; ++nb;
        }
        B.interval_of_q[j] = nb - 1;
    }
    B.n_intervals = nb;
    for (int b = 0; b < nb; ++b) {
        const unsigned long long span = B.hi[b] - B.lo[b];
        const int bits = span ? 64 - __clzll((long long)span) : 0;   // (span >> shift) < 2^sub_bits
        B.shift[b] = bits > sub_bits ? bits - sub_bits : 0;
        B.xlo[b] = 0.0; B.inv_w[b] = 0.0;
        if (B.lo[b] != 0ull && B.hi[b] < kRqKeyPosInf && span) {     // closed, finite at the top: may take value bins
            const double xl = value_of(B.lo[b]), xh = value_of(B.hi[b]);
            const double w = xh - xl;
            if (!(xl > 0.0) && !(xh < 0.0) && w > 0.0 && w < 1.0e300 && xl > -1.0e300) {   // reaches or straddles zero
                B.shift[b] = -1;
                B.xlo[b] = xl;
                B.inv_w[b] = (double)(1 << sub_bits) / w;
            }
        }
    }
    for (int k = 0; k <= 2 * kRqMaxQ; ++k) B.pos_count[k] = 0ull;
    B.n_nan = 0ull; B.cand_count = 0u; B.fallback = 0u; B.open_lo = 0u; B.open_hi = 0u;
    // bucket grid of the slab pass: equal steps of VALUE from the lowest to the highest finite bound
    {
        const double inf = __longlong_as_double(0x7ff0000000000000LL);
        double xa = inf, xb = -inf;
        for (int i = 0; i < 2 * nb; ++i) {
            const double v = rq_bound_value((i & 1) ? B.hi[i >> 1] + 1ull : B.lo[i >> 1]);
            if (v == v && v > -inf && v < inf) { xa = v < xa ? v : xa; xb = v > xb ? v : xb; }
        }
        double sc = 0.0, cc = 0.0;
        if (xb > xa) {
            sc = (double)(kRqLutSize - 2) / (xb - xa);
            cc = -xa * sc;
            if (!(sc > 0.0 && sc < 1.0e300 && cc == cc && cc > -1.0e300 && cc < 1.0e300)) { sc = 0.0; cc = 0.0; }
        }
        B.lut_s = sc; B.lut_c = cc; B.lut_pad = 0u;
        int prev2 = -1, prev1 = -1, ok = 1;       // the bounds ascend: three in one bucket <=> bucket[i] == bucket[i - 2]
        for (int i = 0; i < 2 * nb; ++i) {
            const int k = rq_lut_bucket_of_bound(rq_bound_value((i & 1) ? B.hi[i >> 1] + 1ull : B.lo[i >> 1]), sc, cc);
            if (i >= 2 && k == prev2) ok = 0;
            prev2 = prev1; prev1 = k;
        }
        B.lut_ok = (unsigned int)ok;
    }
}

// (1) Coarse brackets.  One workgroup per row sorts the first kRqTiny entries in LDS (bitonic, NaNs last) and takes,
// per quantile, the sample's order statistics kRqCoarseSigmas binomial standard deviations either side of the
// quantile's rank.  A one-key interval (lo == hi: the bracket sits inside one giant tie) stays a one-key interval
// all the way down and needs no candidates.
__global__ __launch_bounds__(1024) void rq_tiny_kernel(const double* __restrict__ rows, int64_t row_stride, int64_t n_avail,
                                                      const RqArgs args, RqBracket* br1, unsigned int* hist1, unsigned int* fb_count) {
    __shared__ unsigned long long keys[kRqTiny];
    __shared__ unsigned long long qlo[kRqMaxQ], qhi[kRqMaxQ];
    __shared__ unsigned int nan_n, open_lo, open_hi;
    const int row = blockIdx.x, t = threadIdx.x;
    if (row == 0 && t == 0 && fb_count) *fb_count = 0u;   // (nullptr: the caller zeroes it — row groups on several streams)
    if (t == 0) { nan_n = 0u; open_lo = 0u; open_hi = 0u; }
    __syncthreads();
    const double* r = rows + (int64_t)row * row_stride;
    unsigned int my_nan = 0u;
    // element i = e * 1024 + t lives in key[e] of thread t
    unsigned long long key[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const double x = (int64_t)(e * 1024 + t) < n_avail ? r[e * 1024 + t] : __longlong_as_double(0x7ff8000000000000LL);
        const bool isnan_x = x != x;
        key[e] = isnan_x ? ~0ull : key_of(x);        // (~0 is not the key of any non-NaN double: NaNs sort last)
        my_nan += isnan_x ? 1u : 0u;
    }
    if (my_nan) atomicAdd(&nan_n, my_nan);
    unsigned int* hrow = hist1 + (size_t)row * kRqMaxQ * kRqMaxSubBins;
    for (int k = t; k < kRqMaxQ * kRqMaxSubBins; k += 1024) hrow[k] = 0u;
    // Bitonic sort, ascending.  Partner of i at distance j: another register of the same thread (j >= 1024), another
    // lane of the same wave (j < 64: a shuffle, no barrier) or a thread of another wave (through LDS, two barriers).
    for (int k = 2; k <= kRqTiny; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 1024) {   // (constant register indices: a run-time index would push key[] into scratch)
                auto cx = [&](unsigned long long& a, unsigned long long& b, bool up) {
                    if ((a > b) == up) { const unsigned long long x = a; a = b; b = x; }
                };
                if (j == 2048) { cx(key[0], key[2], true); cx(key[1], key[3], true); }          // k = 4096: ascending
                else { cx(key[0], key[1], true); cx(key[2], key[3], k == 4096); }               // j = 1024: k = 2048 or 4096
            } else {
                unsigned long long other[4];
                if (j >= 64) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) keys[e * 1024 + t] = key[e];
                    __syncthreads();
#pragma unroll
                    for (int e = 0; e < 4; ++e) other[e] = keys[e * 1024 + (t ^ j)];
                    __syncthreads();
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) other[e] = (unsigned long long)__shfl_xor((long long)key[e], j, 64);
                }
                const bool lower = (t & j) == 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool up = (((e * 1024 + t) & k) == 0);
                    const unsigned long long mn = key[e] < other[e] ? key[e] : other[e], mx = key[e] < other[e] ? other[e] : key[e];
                    key[e] = (lower == up) ? mn : mx;
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) keys[e * 1024 + t] = key[e];
    __syncthreads();
    RqBracket& B = br1[row];
    const int v0 = kRqTiny - (int)nan_n;
    if (v0 < 256) {   // hardly any non-NaN entry in the sample: no basis for brackets
        if (t == 0) { rq_make_intervals(B, qlo, qhi, 0, kRqCoarseSubBits); B.fallback = 1u; }
        return;
    }
    if (t < args.n_q) {
        const double q = args.q[t];
        const double vi = (double)(v0 - 1) * q;
        const double d = ceil(kRqCoarseSigmas * sqrt((double)v0 * q * (1.0 - q))) + 2.0;
        const double lo = floor(vi) - d, hi = floor(vi) + 1.0 + d;
        // An end that runs off the sample stops at the sample's own minimum / maximum, NOT at the end of the key range:
        // an interval open to key 0 or +inf would spread its sub-bins over the whole exponent range and resolve nothing.
        // Such an end is flagged: if the second sample's window reaches beyond it (extreme quantiles), the fine bracket
        // is opened to the end of the key range there.
        if (lo < 0.0) atomicOr(&open_lo, 1u << t);
        if (hi > (double)(v0 - 1)) atomicOr(&open_hi, 1u << t);
        qlo[t] = keys[lo < 0.0 ? 0 : (int)lo];
        qhi[t] = keys[hi > (double)(v0 - 1) ? v0 - 1 : (int)hi];
    }
    __syncthreads();
    if (t == 0) { rq_make_intervals(B, qlo, qhi, args.n_q, kRqCoarseSubBits); B.open_lo = open_lo; B.open_hi = open_hi; }
}

struct RqRefineShared {
    unsigned long long below[kRqMaxQ], upto[kRqMaxQ], qlo[kRqMaxQ], qhi[kRqMaxQ];
    unsigned int miss;
};
__device__ __forceinline__ unsigned long long rq_ld_u64(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned int rq_ld_u32(const unsigned int* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ void rq_refine_row(int row, int64_t m, const RqArgs& args, const RqBracket* br1, const unsigned int* hist1,
                              RqBracket* br2, unsigned int* hist2, int sub_bits2, unsigned int* gfill, unsigned int* pre, RqRefineShared& S);
// (2) and (4) THE counting pass, over the first `n` entries of every row (the sample, then the whole slab).  Every
// non-NaN key is located among the row's sorted interval bounds by a branch-free binary search (P-entry table in LDS:
// bound 2b = lo[b], bound 2b+1 = hi[b]+1, padded with ~0; position = number of bounds <= key), counted in a per-thread
// LDS histogram laid out [position][thread] (conflict-free, no contended atomics), and — when its position is odd,
// i.e. it lies inside an interval that is a real range — tallied in that interval's sub-histogram (LDS atomics on
// 2^sub_bits bins: the keys of one wave scatter) and, if COMPACT, appended to the row's candidate buffer (staged per
// wave in LDS: one global reservation and a coalesced copy per ~256 candidates, no workgroup barrier in the loop).
// 8 B/element read, a few % written.
template <int P, bool COMPACT>
__global__ __launch_bounds__(kRqBlock) __attribute__((amdgpu_waves_per_eu(5, 8))) void rq_count_kernel(const double* __restrict__ rows, int64_t row_stride, int64_t n,
                                                           RqBracket* br, unsigned int* hist, int sub_bits, int max_q, double* cand,
                                                           unsigned int cand_cap, int skip_lut_rows) {
    constexpr int kWaveStage = kRqWaveStage;             // candidates a wave collects in LDS before it appends them
    __shared__ unsigned long long bound[P];
    __shared__ int shl[P / 2];
    __shared__ double xlo_s[P / 2], invw_s[P / 2];
    __shared__ unsigned int nan_n;
    // dynamic LDS, sized by the host for the call's quantile count (every KB counts: 6 workgroups per CU at 7 quantiles):
    // stage f64 [4][kWaveStage] (COMPACT) | poshist u32 [2 n_q + 1][256] | subhist u32 [n_q][2^sub_bits]
    extern __shared__ __align__(16) unsigned char dyn_lds[];
    double* stage = reinterpret_cast<double*>(dyn_lds);
    unsigned int* poshist = reinterpret_cast<unsigned int*>(dyn_lds + (COMPACT ? (kRqBlock / 64) * kWaveStage * sizeof(double) : 0));
    unsigned int* subhist = poshist + (size_t)(2 * max_q + 1) * kRqBlock;
    const int row = blockIdx.y;
    RqBracket& B = br[row];
    const int nb = B.n_intervals;
    if (B.fallback || 2 * nb >= P || nb > max_q) return;   // (the host picks P > 2 * n_q >= 2 * nb)
    if (skip_lut_rows && B.lut_ok) return;                 // rq_slab_kernel takes this row
    const int bins = 1 << sub_bits;
    if (threadIdx.x < P) {
        const int b = threadIdx.x >> 1;
        bound[threadIdx.x] = b < nb ? ((threadIdx.x & 1) ? B.hi[b] + 1ull : B.lo[b]) : ~0ull;   // hi <= key(+inf): no wrap
        if ((threadIdx.x & 1) == 0) { shl[b] = b < nb ? B.shift[b] : 0; xlo_s[b] = b < nb ? B.xlo[b] : 0.0; invw_s[b] = b < nb ? B.inv_w[b] : 0.0; }
    }
    for (int k = threadIdx.x; k < (2 * nb + 1) * kRqBlock; k += kRqBlock) poshist[k] = 0u;
    for (int k = threadIdx.x; k < nb * bins; k += kRqBlock) subhist[k] = 0u;
    if (threadIdx.x == 0) nan_n = 0u;
    unsigned int keep = 0u;                            // bit b: interval b is a real range (lo < hi): members are candidates
    for (int b = 0; b < nb; ++b) keep |= (B.lo[b] < B.hi[b]) ? (1u << b) : 0u;
    keep = (unsigned int)__builtin_amdgcn_readfirstlane((int)keep);
    __syncthreads();
    const double* r = rows + (int64_t)row * row_stride;
    double* crow = COMPACT ? cand + (size_t)row * cand_cap : nullptr;
    const int lane = threadIdx.x & 63;
    unsigned int my_nan = 0u;
    double* wstage = stage + (COMPACT ? (threadIdx.x >> 6) * kWaveStage : 0);   // this wave's stage; `filled` is wave-uniform
    unsigned int filled = 0u;
    // append the wave's staged candidates to the row's buffer: one global reservation, coalesced copy
    auto flush = [&]() {
        if (!COMPACT || filled == 0u) return;
        unsigned int base = 0u;
        if (lane == 0) base = atomicAdd(&B.cand_count, filled);
        base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
        for (unsigned int i = lane; i < filled; i += 64u)
            if (base + i < cand_cap) crow[base + i] = wstage[i];
        filled = 0u;
    };

    // K elements per lane at a time: first all K binary searches (independent LDS chains the scheduler can overlap),
    // then ONE wave scan + ONE stage reservation for the candidates among the wave's 64 * K elements.
    auto consume = [&](auto kc, const double* xs, const bool* oks) {
        constexpr int K = decltype(kc)::value;
        bool ins[K], valid[K];
        unsigned long long key[K];
        unsigned int pos[K];
        unsigned int mine = 0u;
#pragma unroll
        for (int u = 0; u < K; ++u) {
            const double x = xs[u];
            const bool isnan_x = x != x;
            my_nan += (oks[u] && isnan_x) ? 1u : 0u;
            valid[u] = oks[u] && !isnan_x;
            key[u] = key_of(x);
            pos[u] = 0u;
        }
        // level by level across the K elements: K independent LDS reads in flight per level, no branches
#pragma unroll
        for (int half = P / 2; half >= 1; half >>= 1) {
#pragma unroll
            for (int u = 0; u < K; ++u) pos[u] += (bound[pos[u] + half - 1] <= key[u]) ? (unsigned int)half : 0u;
        }
#pragma unroll
        for (int u = 0; u < K; ++u) {
            atomicAdd(&poshist[pos[u] * kRqBlock + threadIdx.x], valid[u] ? 1u : 0u);   // own slot: ds_add_u32, no conflicts
            ins[u] = valid[u] && (pos[u] & 1u) && ((keep >> (pos[u] >> 1)) & 1u);
            if (ins[u]) {
                const unsigned int b = pos[u] >> 1;
                atomicAdd(&subhist[b * bins + (unsigned int)rq_sub_bin(key[u], xs[u], bound[2 * b], shl[b], xlo_s[b], invw_s[b], bins)], 1u);
            }
            mine += ins[u] ? 1u : 0u;
        }
        if (COMPACT && __ballot(mine != 0u)) {           // wave-uniform
            unsigned int scan = mine;         // inclusive scan of the per-lane candidate counts
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned int up = (unsigned int)__shfl_up((int)scan, off, 64);
                scan += lane >= off ? up : 0u;
            }
            const unsigned int total = (unsigned int)__builtin_amdgcn_readlane((int)scan, 63);
            if (total > (unsigned int)kWaveStage) {
                // more than a stage of candidates in one batch (a wide interval): straight to the row's buffer
                flush();
                unsigned int base = 0u;
                if (lane == 0) base = atomicAdd(&B.cand_count, total);
                unsigned int slot = (unsigned int)__builtin_amdgcn_readfirstlane((int)base) + scan - mine;
#pragma unroll
                for (int u = 0; u < K; ++u)
                    if (ins[u]) { if (slot < cand_cap) crow[slot] = xs[u]; ++slot; }
            } else {
                if (filled + total > (unsigned int)kWaveStage) flush();
                unsigned int slot = filled + scan - mine;
#pragma unroll
                for (int u = 0; u < K; ++u)
                    if (ins[u]) wstage[slot++] = xs[u];
                filled += total;
            }
        }
    };

    const int64_t step = (int64_t)gridDim.x * kRqBlock;
    const bool vec2 = ((row_stride & 1) == 0) && ((reinterpret_cast<uintptr_t>(rows) & 15) == 0);
    if (vec2) {
        typedef double d2_t __attribute__((ext_vector_type(2)));
        const d2_t* r2 = reinterpret_cast<const d2_t*>(r);
        const int64_t n_pairs = n / 2;
        const int64_t trips = (n_pairs + step - 1) / step;
        constexpr int kUnroll = 4;     // four 16-byte loads per lane and batch; the next batch is in flight while this one is consumed
        d2_t v[kUnroll];
        bool okv[kUnroll];
        auto fetch = [&](int64_t t) {
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const int64_t i = (t + u) * step + (int64_t)blockIdx.x * kRqBlock + threadIdx.x;
                okv[u] = i < n_pairs;
                v[u] = okv[u] ? __builtin_nontemporal_load(&r2[i]) : d2_t{0.0, 0.0};
            }
        };
        if (trips > 0) fetch(0);
        for (int64_t t = 0; t < trips; t += kUnroll) {
            double xs[2 * kUnroll];
            bool oks[2 * kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) { xs[2 * u] = v[u].x; xs[2 * u + 1] = v[u].y; oks[2 * u] = oks[2 * u + 1] = okv[u]; }
            if (t + kUnroll < trips) fetch(t + kUnroll);
            consume(std::integral_constant<int, 2 * kUnroll>{}, xs, oks);
        }
        if ((n & 1) && blockIdx.x == 0) {
            const double x1 = r[n - 1];
            const bool ok1 = threadIdx.x == 0;
            consume(std::integral_constant<int, 1>{}, &x1, &ok1);
        }
    } else {
        const int64_t trips = (n + step - 1) / step;
        for (int64_t t = 0; t < trips; ++t) {
            const int64_t i = t * step + (int64_t)blockIdx.x * kRqBlock + threadIdx.x;
            const bool ok1 = i < n;
            const double x1 = ok1 ? r[i] : 0.0;
            consume(std::integral_constant<int, 1>{}, &x1, &ok1);
        }
    }
    if (my_nan) atomicAdd(&nan_n, my_nan);
    __syncthreads();
    // fold the per-thread histograms: one wave per position at a time, one global atomic per position and workgroup
    for (int p = threadIdx.x >> 6; p <= 2 * nb; p += kRqBlock / 64) {
        unsigned int c = poshist[p * kRqBlock + lane] + poshist[p * kRqBlock + 64 + lane] +
                         poshist[p * kRqBlock + 128 + lane] + poshist[p * kRqBlock + 192 + lane];
        for (int off = 32; off > 0; off >>= 1) c += (unsigned int)__shfl_down((int)c, off, 64);
        if (lane == 0 && c) atomicAdd(&B.pos_count[p], (unsigned long long)c);
    }
    if (threadIdx.x == 0 && nan_n) atomicAdd(&B.n_nan, (unsigned long long)nan_n);
    unsigned int* hrow = hist + (size_t)row * kRqMaxQ * kRqMaxSubBins;
    for (int k = threadIdx.x; k < nb * bins; k += kRqBlock)
        if (subhist[k]) atomicAdd(&hrow[(k >> sub_bits) * kRqMaxSubBins + (k & (bins - 1))], subhist[k]);
    flush();
}

// (4') THE pass over the slab for rows whose bounds fit the bucket table (B.lut_ok; the others take rq_count_kernel<P, true>).
// Same counts, sub-histograms and candidate buffer as rq_count_kernel<P, true>, at about half its instructions per
// element — the slab pass was VALU-bound (81 % busy at 5.0 TB/s, profiles/r02), not HBM-bound:
//  * no key conversion and no binary search: an equal-width bucket grid in VALUE over the span of the row's bounds (2048
//    buckets: one fma, one conversion, one clamp) yields, through a byte table in LDS, the number of bounds below the
//    bucket; the at most two bounds INSIDE the bucket are compared directly (one 16-byte LDS read of the pair
//    (bound[base], bound[base + 1]), two v_cmp_ge_f64 + add-with-carry).  The 4-level search cost 4 dependent LDS reads,
//    4 64-bit compares and 4 selects per element, plus the key conversion;
//  * NaNs are routed to a trash position (one compare + select) instead of a separate per-lane counter;
//  * members of a bracket (~4 % of the entries) are only STAGED in the element loop (ballot + mbcnt slot, one LDS write
//    under the exec mask); their sub-histogram tally runs when the wave's stage is flushed, on full lanes — in the
//    element loop that code ran for the whole wave whenever ANY lane held a member, i.e. nearly always.
#ifndef MCR_RQ_UNROLL
#define MCR_RQ_UNROLL 4
#endif
template <int P>
__global__ __launch_bounds__(kRqBlock) __attribute__((amdgpu_waves_per_eu(5, 8))) void rq_slab_kernel(const double* __restrict__ rows, int64_t row_stride, int64_t n,
                                                          RqBracket* br, unsigned int* hist, int sub_bits, int max_q, double* cand,
                                                          unsigned int cand_cap) {
    constexpr int kStage = 192;                          // candidates a wave stages in LDS between two flushes
    typedef double d2_t __attribute__((ext_vector_type(2)));
    __shared__ unsigned long long bound[P];              // key bounds (sub-bin arithmetic of the staged candidates)
    __shared__ __align__(16) d2_t pairs[P + 1];          // (value of bound i, value of bound i + 1), NaN-padded
    __shared__ int bbucket[P];
    __shared__ int shl[P / 2];
    __shared__ double xlo_s[P / 2], invw_s[P / 2];
    __shared__ unsigned char lut[kRqLutSize + 4];
    // dynamic LDS: stage f64 [4][kStage] | poshist u32 [2 max_q + 2][256] | subhist u32 [max_q][2^sub_bits]
    extern __shared__ __align__(16) unsigned char dyn_lds[];
    double* stage = reinterpret_cast<double*>(dyn_lds);
    unsigned int* poshist = reinterpret_cast<unsigned int*>(dyn_lds + (kRqBlock / 64) * kStage * sizeof(double));
    unsigned int* subhist = poshist + (size_t)(2 * max_q + 2) * kRqBlock;
    const int row = blockIdx.y;
    RqBracket& B = br[row];
    const int nb = B.n_intervals;
    if (B.fallback || !B.lut_ok || 2 * nb >= P || nb > max_q) return;
    const int bins = 1 << sub_bits;
    const double lut_s = B.lut_s;
    double lut_c = B.lut_c;
    asm volatile("" : "+v"(lut_c));                      // fma(x, s, c): one scalar operand at most — keep the addend in a VGPR
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    if (threadIdx.x < P) {
        const int b = threadIdx.x >> 1;
        const unsigned long long kb = b < nb ? ((threadIdx.x & 1) ? B.hi[b] + 1ull : B.lo[b]) : ~0ull;
        bound[threadIdx.x] = kb;
        const double bv = b < nb ? rq_bound_value(kb) : nan;
        reinterpret_cast<double*>(pairs)[2 * threadIdx.x] = bv;
        if (threadIdx.x > 0) reinterpret_cast<double*>(pairs)[2 * threadIdx.x - 1] = bv;
        bbucket[threadIdx.x] = rq_lut_bucket_of_bound(bv, lut_s, lut_c);
        if ((threadIdx.x & 1) == 0) { shl[b] = b < nb ? B.shift[b] : 0; xlo_s[b] = b < nb ? B.xlo[b] : 0.0; invw_s[b] = b < nb ? B.inv_w[b] : 0.0; }
    }
    if (threadIdx.x == 0) { reinterpret_cast<double*>(pairs)[2 * P - 1] = nan; reinterpret_cast<double*>(pairs)[2 * P] = nan; reinterpret_cast<double*>(pairs)[2 * P + 1] = nan; }
    const int n_pos = 2 * nb + 2;                        // positions 0 .. 2 nb, then the trash position of the NaNs
    const unsigned int trash = (unsigned int)(2 * nb + 1);
    for (int k = threadIdx.x; k < n_pos * kRqBlock; k += kRqBlock) poshist[k] = 0u;
    for (int k = threadIdx.x; k < nb * bins; k += kRqBlock) subhist[k] = 0u;
    unsigned int keep = 0u;                              // bit p (odd p = 2b + 1): interval b is a real range: members are candidates
    for (int b = 0; b < nb; ++b) keep |= (B.lo[b] < B.hi[b]) ? (1u << (2 * b + 1)) : 0u;
    keep = (unsigned int)__builtin_amdgcn_readfirstlane((int)keep);
    __syncthreads();
    for (int k = threadIdx.x; k < kRqLutSize; k += kRqBlock) {      // bounds in buckets below k
        int c = 0;
        for (int i = 0; i < 2 * nb; ++i) c += bbucket[i] < k ? 1 : 0;
        lut[k] = (unsigned char)c;
    }
    __syncthreads();
    const double* r = rows + (int64_t)row * row_stride;
    double* crow = cand + (size_t)row * cand_cap;
    const int lane = threadIdx.x & 63;
    double* wstage = stage + (threadIdx.x >> 6) * kStage;      // this wave's stage; `filled` is wave-uniform
    unsigned int filled = 0u;
    unsigned int* myhist = poshist + threadIdx.x;

    // position of x among the bounds (number of bounds <= x); NaN -> anything (the caller routes NaNs to the trash slot)
    auto position = [&](double x) -> unsigned int {
        const int k = rq_lut_bucket(x, lut_s, lut_c);
        const unsigned int base = lut[k];
        const d2_t bp = pairs[base];
        return base + (x >= bp.x ? 1u : 0u) + (x >= bp.y ? 1u : 0u);
    };
    // sub-histogram tally of ONE member of interval b (full lanes at a flush; under the exec mask in the rare direct path)
    auto tally_in = [&](double x, unsigned int b) {
        atomicAdd(&subhist[b * bins + (unsigned int)rq_sub_bin(key_of(x), x, bound[2 * b], shl[b], xlo_s[b], invw_s[b], bins)], 1u);
    };
    auto tally = [&](double x) { tally_in(x, position(x) >> 1); };
    auto flush = [&]() {
        if (filled == 0u) return;
        unsigned int base = 0u;
        if (lane == 0) base = atomicAdd(&B.cand_count, filled);
        base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
        for (unsigned int i = lane; i < filled; i += 64u) {
            const double x = wstage[i] + 0.0;                   // -0.0 -> +0.0 (see rq_bound_value)
            tally(x);
            // (the few writes of the pass — 4 % of its bytes — cost it a fifth of its read rate: a property of mixed streams on
            //  this part, tools/ubench/hbm_read.hip "S" lines; nontemporal stores help the bare sweep by 6 % and this call not at
            //  all.  The part-by-part timing of this kernel — staging / atomic / stores / tally switched off one at a time — was
            //  an experiment build of round 3: profiles/r03/k3_slab_parts.txt, LABNOTES.md)
            if (base + i < cand_cap) crow[base + i] = x;
        }
        filled = 0u;
    };

    auto consume = [&](auto kc, const double* xs, const bool* oks) {
        constexpr int K = decltype(kc)::value;
        unsigned long long mask[K];
        unsigned int total = 0u;
#pragma unroll
        for (int u = 0; u < K; ++u) {
            const double x = xs[u];
            unsigned int pos = position(x);
            pos = x != x ? trash : pos;
            if (oks[u]) atomicAdd(&myhist[pos * kRqBlock], 1u);            // own slot: ds_add_u32, no conflicts
            mask[u] = __ballot(oks[u] && ((keep >> pos) & 1u));            // (trash is even or beyond the keep bits: never a member)
            total += (unsigned int)__popcll(mask[u]);
        }
        if (total == 0u) return;
        if (total > (unsigned int)kStage) {
            // more than a stage of members in one batch (a wide interval / a giant tie straddling a bound): straight to the row's buffer
            flush();
#pragma unroll
            for (int u = 0; u < K; ++u) {
                if (mask[u] == 0ull) continue;
                unsigned int base = 0u;
                if (lane == 0) base = atomicAdd(&B.cand_count, (unsigned int)__popcll(mask[u]));
                base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
                if ((mask[u] >> lane) & 1ull) {
                    const double x = xs[u] + 0.0;
                    const unsigned int slot = base + __builtin_amdgcn_mbcnt_hi((unsigned int)(mask[u] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mask[u], 0u));
                    tally(x);
                    if (slot < cand_cap) crow[slot] = x;
                }
            }
            return;
        }
        if (filled + total > (unsigned int)kStage) flush();
#pragma unroll
        for (int u = 0; u < K; ++u) {
            if (mask[u] == 0ull) continue;                                  // wave-uniform
            if ((mask[u] >> lane) & 1ull)
                wstage[filled + __builtin_amdgcn_mbcnt_hi((unsigned int)(mask[u] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mask[u], 0u))] = xs[u];
            filled += (unsigned int)__popcll(mask[u]);
        }
    };

    const int64_t step = (int64_t)gridDim.x * kRqBlock;
    const bool vec2 = ((row_stride & 1) == 0) && ((reinterpret_cast<uintptr_t>(rows) & 15) == 0);
    if (vec2) {
        const d2_t* r2 = reinterpret_cast<const d2_t*>(r);
        const int64_t n_pairs = n / 2;
        const int64_t trips = (n_pairs + step - 1) / step;
        constexpr int kUnroll = MCR_RQ_UNROLL;     // 16-byte loads per lane and batch; the next batch is in flight while this one is consumed
        d2_t v[kUnroll];
        bool okv[kUnroll];
        auto fetch = [&](int64_t t) {
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                // a batch = ONE contiguous 16 KB tile of the row per workgroup (kUnroll chunks a grid stride apart measured 1 % slower)
                const int64_t i = t * step + ((int64_t)blockIdx.x * kUnroll + u) * kRqBlock + threadIdx.x;
                okv[u] = i < n_pairs;
                v[u] = okv[u] ? __builtin_nontemporal_load(&r2[i]) : d2_t{0.0, 0.0};   // (plain loads: 2.21 -> 2.37 ms)
            }
        };
        if (trips > 0) fetch(0);
        for (int64_t t = 0; t < trips; t += kUnroll) {
            double xs[2 * kUnroll];
            bool oks[2 * kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) { xs[2 * u] = v[u].x; xs[2 * u + 1] = v[u].y; oks[2 * u] = oks[2 * u + 1] = okv[u]; }
            if (t + kUnroll < trips) fetch(t + kUnroll);
            consume(std::integral_constant<int, 2 * kUnroll>{}, xs, oks);
        }
        if ((n & 1) && blockIdx.x == 0) {
            const double x1 = r[n - 1];
            const bool ok1 = threadIdx.x == 0;
            consume(std::integral_constant<int, 1>{}, &x1, &ok1);
        }
    } else {
        const int64_t trips = (n + step - 1) / step;
        for (int64_t t = 0; t < trips; ++t) {
            const int64_t i = t * step + (int64_t)blockIdx.x * kRqBlock + threadIdx.x;
            const bool ok1 = i < n;
            const double x1 = ok1 ? r[i] : 0.0;
            consume(std::integral_constant<int, 1>{}, &x1, &ok1);
        }
    }
    flush();
    __syncthreads();
    // fold the per-thread histograms: one wave per position at a time, one global atomic per position and workgroup
    for (int p = threadIdx.x >> 6; p < n_pos; p += kRqBlock / 64) {
        unsigned int c = poshist[p * kRqBlock + lane] + poshist[p * kRqBlock + 64 + lane] +
                         poshist[p * kRqBlock + 128 + lane] + poshist[p * kRqBlock + 192 + lane];
        for (int off = 32; off > 0; off >>= 1) c += (unsigned int)__shfl_down((int)c, off, 64);
        if (lane == 0 && c) atomicAdd(p == (int)trash ? &B.n_nan : &B.pos_count[p], (unsigned long long)c);
    }
    unsigned int* hrow = hist + (size_t)row * kRqMaxQ * kRqMaxSubBins;
    for (int k = threadIdx.x; k < nb * bins; k += kRqBlock)
        if (subhist[k]) atomicAdd(&hrow[(k >> sub_bits) * kRqMaxSubBins + (k & (bins - 1))], subhist[k]);
}

// One wave turns h[0 .. len) (LDS) into its inclusive prefix sums, in place.
__device__ __forceinline__ void rq_wave_inclusive_scan(unsigned int* h, int len, int lane) {
    const int per = (len + 63) / 64;
    const int first = lane * per;
    unsigned int sum = 0u;
    for (int i = first; i < first + per && i < len; ++i) sum += h[i];
    unsigned int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned int up = (unsigned int)__shfl_up((int)incl, off, 64);
        incl += lane >= off ? up : 0u;
    }
    unsigned int run = incl - sum;
    for (int i = first; i < first + per && i < len; ++i) { run += h[i]; h[i] = run; }
}
// first s in [0, len) with x < pre[s] (pre = inclusive prefix sums, x < pre[len - 1])
__device__ __forceinline__ int rq_upper_bound(const unsigned int* pre, int len, unsigned long long x) {
    int lo = 0, hi = len - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (x < (unsigned long long)pre[mid]) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// (3) Fine brackets.  Per row: the counts of the sample pass say where the sample's order statistics 5 sigma either
// side of every quantile lie — in which sub-bin of which coarse interval; the edges of those sub-bins are key bounds
// that contain the wanted order statistics of the WHOLE row with overwhelming probability (and exactness is checked
// against the slab's own counts afterwards).  A row whose window leaves its coarse interval takes the radix route.
// (All 256 threads of one workgroup, for one row.  The counters of the sample pass are read with device-scope atomic loads:
//  the caller may be the LAST workgroup of that very pass — the others' counts arrived through global atomics.)
__device__ void rq_refine_row(int row, int64_t m, const RqArgs& args, const RqBracket* br1, const unsigned int* hist1,
                              RqBracket* br2, unsigned int* hist2, int sub_bits2, unsigned int* gfill, unsigned int* pre, RqRefineShared& S) {
    unsigned long long* below = S.below; unsigned long long* upto = S.upto; unsigned long long* qlo = S.qlo; unsigned long long* qhi = S.qhi;
    unsigned int& miss = S.miss;
    const int t = threadIdx.x, lane = t & 63;
    const RqBracket& B1 = br1[row];
    RqBracket& B2 = br2[row];
    unsigned int* h2 = hist2 + (size_t)row * kRqMaxQ * kRqMaxSubBins;
    for (int k = t; k < kRqMaxQ * kRqMaxSubBins; k += 256) h2[k] = 0u;
    if (t < kRqMaxT) gfill[(size_t)row * kRqMaxT + t] = 0u;
    const int nb = B1.n_intervals;
    constexpr int bins = 1 << kRqCoarseSubBits;
    if (t == 0) miss = B1.fallback;
    const unsigned int* h1 = hist1 + (size_t)row * kRqMaxQ * kRqMaxSubBins;
    for (int k = t; k < nb * bins; k += 256) pre[k] = rq_ld_u32(&h1[(k >> kRqCoarseSubBits) * kRqMaxSubBins + (k & (bins - 1))]);
    __syncthreads();
    for (int b = t >> 6; b < nb; b += 4) rq_wave_inclusive_scan(pre + b * bins, bins, lane);
    if (t == 0) {
        unsigned long long run = 0ull;
        for (int b = 0; b < nb; ++b) {
            run += rq_ld_u64(&B1.pos_count[2 * b]); below[b] = run;
            run += rq_ld_u64(&B1.pos_count[2 * b + 1]); upto[b] = run;
        }
    }
    __syncthreads();
    unsigned long long mv;   // non-NaN entries of the sample (m <= 0: a sample pooled over ranks, its size is the sum of the counts)
    if (m > 0) mv = (unsigned long long)m - rq_ld_u64(&B1.n_nan);
    else { mv = 0ull; for (int k = 0; k <= 2 * nb; ++k) mv += rq_ld_u64(&B1.pos_count[k]); }
    if (t < args.n_q && !miss) {
        if (mv < 1024ull) {
            miss = 1u;
        } else {
            const double q = args.q[t];
            const double vi = (double)(mv - 1) * q;
            const double d = ceil(kRqBracketSigmas * sqrt((double)mv * q * (1.0 - q))) + 2.0;
            const double rl = floor(vi) - d, rh = floor(vi) + 1.0 + d;
            const int b = B1.interval_of_q[t];
            const unsigned long long lo_b = B1.lo[b], hi_b = B1.hi[b];
            const bool tie = lo_b == hi_b;
            unsigned long long klo, khi;
            const bool olo = (B1.open_lo >> t) & 1u, ohi = (B1.open_hi >> t) & 1u;
            if (rl < 0.0) klo = 0ull;                                        // window open below: whole key range
            else {
                const unsigned long long r = (unsigned long long)rl;
                if (r < below[b] && olo) klo = 0ull;                         // beyond a flagged coarse end: open there too
                else if (r >= upto[b] && ohi) klo = hi_b;                    // whole window above the flagged top end
                else if (r < below[b] || r >= upto[b]) { miss = 1u; klo = lo_b; }   // the coarse bracket missed
                else if (tie) klo = lo_b;
                else {
                    const int sb = rq_upper_bound(pre + b * bins, bins, r - below[b]);
                    if (B1.shift[b] >= 0) klo = lo_b + ((unsigned long long)sb << B1.shift[b]);
                    else {   // value bins: an edge 1/1000 of a bin on the safe side of the exact one (rounding is far smaller)
                        klo = B1.inv_w[b] > 0.0 ? key_of(B1.xlo[b] + ((double)sb - 1.0e-3) / B1.inv_w[b]) : lo_b;
                        if (klo < lo_b) klo = lo_b;
                    }
                }
            }
            if (rh > (double)(mv - 1)) khi = kRqKeyPosInf;                   // window open above
            else {
                const unsigned long long r = (unsigned long long)rh;
                if (r >= upto[b] && ohi) khi = kRqKeyPosInf;
                else if (r < below[b] && olo) khi = lo_b;                    // whole window below the flagged bottom end
                else if (r < below[b] || r >= upto[b]) { miss = 1u; khi = hi_b; }
                else if (tie) khi = hi_b;
                else {
                    const int sb = rq_upper_bound(pre + b * bins, bins, r - below[b]);
                    if (B1.shift[b] >= 0) {
                        unsigned long long off = ((unsigned long long)(sb + 1) << B1.shift[b]) - 1ull;
                        if (off > hi_b - lo_b) off = hi_b - lo_b;           // (also catches the shift wrapping past 2^64)
                        khi = lo_b + off;
                    } else {
                        khi = (B1.inv_w[b] > 0.0 && sb < bins - 1) ? key_of(B1.xlo[b] + ((double)sb + 1.0 + 1.0e-3) / B1.inv_w[b]) : hi_b;
                        if (khi > hi_b) khi = hi_b;
                    }
                }
            }
            qlo[t] = klo; qhi[t] = khi;
        }
    }
    __syncthreads();
    if (t == 0) {
        if (miss) { rq_make_intervals(B2, qlo, qhi, 0, sub_bits2); B2.fallback = 1u; }
        else rq_make_intervals(B2, qlo, qhi, args.n_q, sub_bits2);
    }
}
// stand-alone form (rows sharded over ranks: the sample's counts are summed across ranks before the brackets are refined)
__global__ __launch_bounds__(256) void rq_refine_kernel(int64_t m, const RqArgs args, const RqBracket* br1, const unsigned int* hist1,
                                                       RqBracket* br2, unsigned int* hist2, int sub_bits2, unsigned int* gfill) {
    extern __shared__ __align__(16) unsigned int pre[];   // [n_intervals][1024]: inclusive prefix sums of the sub-histograms
    __shared__ RqRefineShared S;
    rq_refine_row(blockIdx.x, m, args, br1, hist1, br2, hist2, sub_bits2, gfill, pre, S);
}

// (5) and (6) Per row, after the slab pass: the true ranks (NumPy `linear`, as rq_scan_kernel computes them) are
// checked against the counts and located in ONE sub-bin of their interval (a "cell": a few hundred keys, its size
// known exactly from the sub-histogram).  rq_collect_kernel streams the row's candidates (a few workgroups per row)
// and appends those of the wanted cells to the row's cell lists; rq_select_kernel then selects each target inside its
// cell by a per-wave MSD radix select in LDS and interpolates the quantiles.  Both kernels derive the cells with the
// same deterministic routine.  A row any of whose targets lies outside its interval, whose candidates overflowed,
// or whose cells do not fit the list is appended to the list of rows for the full radix passes.
struct RqResolved {   // lives in LDS
    unsigned long long below[kRqMaxQ], upto[kRqMaxQ], ivlo[kRqMaxQ], ivhi[kRqMaxQ];
    unsigned long long tgt_res[kRqMaxT], tgt_key[kRqMaxT];
    unsigned long long m_valid;
    double gamma_q[kRqMaxQ];
    double ivxlo[kRqMaxQ], ivinvw[kRqMaxQ];
    int ivshift[kRqMaxQ];
    int tgt_cell[kRqMaxT], tgt_b[kRqMaxT], tgt_s[kRqMaxT];
    int cell_b[kRqMaxT], cell_s[kRqMaxT];
    unsigned int cell_off[kRqMaxT], cell_size[kRqMaxT];
    unsigned int fb;
    int n_cells;
};
// All threads of a 1024-thread workgroup.  pre: [kRqMaxQ][bins] u32 (LDS), want: [kRqMaxQ][bins] u8 (LDS) or nullptr.
__device__ void rq_resolve_row(RqResolved& R, unsigned int* pre, unsigned char* want, int64_t n, const RqArgs& args,
                               const RqBracket& B, const unsigned int* h2, int sub_bits, unsigned int cand_cap, unsigned int list_cap,
                               const unsigned int* cand_total /* rows sharded over ranks: the candidates of ALL ranks, else nullptr */) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int nb = B.n_intervals;
    const int bins = 1 << sub_bits;
    const int nt = 2 * args.n_q;
    if (t == 0) { R.fb = B.fallback; R.n_cells = 0; R.m_valid = 0ull; }
    for (int k = t; k < nb * bins; k += 1024) {
        pre[k] = h2[(k >> sub_bits) * kRqMaxSubBins + (k & (bins - 1))];
        if (want) want[k] = 255;
    }
    __syncthreads();
    for (int b = wave; b < nb; b += kRqWaves) rq_wave_inclusive_scan(pre + b * bins, bins, lane);
    if (t == 0 && !R.fb) {
        unsigned long long run = 0ull, inside = 0ull;
        for (int b = 0; b < nb; ++b) {
            run += B.pos_count[2 * b]; R.below[b] = run;
            run += B.pos_count[2 * b + 1]; R.upto[b] = run;
            R.ivlo[b] = B.lo[b]; R.ivhi[b] = B.hi[b]; R.ivshift[b] = B.shift[b]; R.ivxlo[b] = B.xlo[b]; R.ivinvw[b] = B.inv_w[b];
            if (B.lo[b] != B.hi[b]) inside += R.upto[b] - R.below[b];           // one-key intervals stored nothing
        }
        R.m_valid = (unsigned long long)n - B.n_nan;
        const unsigned long long stored = cand_total ? (unsigned long long)*cand_total : (unsigned long long)B.cand_count;
        if (B.cand_count > cand_cap || inside != stored) R.fb = 1u;
    }
    __syncthreads();
    const unsigned long long m = R.m_valid;
    if (!R.fb && m > 0 && t < nt) {
        const int j = t >> 1;
        const double q = args.q[j];
        const double vi = (double)(m - 1) * q;                    // _QuantileMethods['linear'].get_virtual_index
        double prev = floor(vi), next = prev + 1.0;               // _get_indexes
        if (vi >= (double)(m - 1)) { prev = (double)(m - 1); next = prev; }
        if (vi < 0.0) { prev = 0.0; next = 0.0; }
        if ((t & 1) == 0) R.gamma_q[j] = vi - floor(vi);          // _get_gamma (linear: unchanged)
        const unsigned long long r = (unsigned long long)((t & 1) ? next : prev);
        const int b = B.interval_of_q[j];
        R.tgt_b[t] = b; R.tgt_cell[t] = -1; R.tgt_s[t] = 0; R.tgt_res[t] = 0ull; R.tgt_key[t] = 0ull;
        if (r < R.below[b] || r >= R.upto[b]) R.fb = 1u;          // outside the bracket: not provable here
        else if (R.ivlo[b] == R.ivhi[b]) R.tgt_key[t] = R.ivlo[b];   // one key: every member IS that value
        else {
            const unsigned long long rr = r - R.below[b];
            if ((unsigned long long)pre[b * bins + bins - 1] != R.upto[b] - R.below[b]) R.fb = 1u;   // (sub-histogram / position counts disagree: never)
            else {
                const int s = rq_upper_bound(pre + b * bins, bins, rr);
                R.tgt_s[t] = s;
                R.tgt_res[t] = rr - (s ? (unsigned long long)pre[b * bins + s - 1] : 0ull);
                R.tgt_cell[t] = -2;                               // needs a cell
            }
        }
    }
    __syncthreads();
    if (t == 0 && !R.fb && m > 0) {   // distinct (interval, sub-bin) cells; sizes are exact
        unsigned int total = 0u;
        int nc = 0;
        for (int a = 0; a < nt; ++a) {
            if (R.tgt_cell[a] != -2) continue;
            int c = -1;
            for (int k = 0; k < nc; ++k)
                if (R.cell_b[k] == R.tgt_b[a] && R.cell_s[k] == R.tgt_s[a]) { c = k; break; }
            if (c < 0) {
                c = nc++;
                R.cell_b[c] = R.tgt_b[a]; R.cell_s[c] = R.tgt_s[a];
                const unsigned int* p = pre + R.tgt_b[a] * bins;
                R.cell_size[c] = p[R.tgt_s[a]] - (R.tgt_s[a] ? p[R.tgt_s[a] - 1] : 0u);
                R.cell_off[c] = total;
                total += R.cell_size[c];
                if (want) want[R.tgt_b[a] * bins + R.tgt_s[a]] = (unsigned char)c;
            }
            R.tgt_cell[a] = c;
        }
        R.n_cells = nc;
        if (total > list_cap) R.fb = 1u;
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void rq_collect_kernel(int64_t n, const RqArgs args, const RqBracket* br2, const unsigned int* hist2,
                                                         int sub_bits, const double* __restrict__ cand, unsigned int cand_cap,
                                                         unsigned int list_cap, unsigned long long* glist, unsigned int* gfill,
                                                         const unsigned int* cand_total, const unsigned int* lsz, int rank, int n_rows) {
    extern __shared__ __align__(16) unsigned char dyn[];     // pre u32 [kRqMaxQ][bins] | want u8 [kRqMaxQ][bins]
    __shared__ RqResolved R;
    __shared__ unsigned int cell_base[kRqMaxT];               // rows sharded over ranks: where this rank's keys start in each cell
    const int row = blockIdx.y, t = threadIdx.x;
    const RqBracket& B = br2[row];
    const int bins = 1 << sub_bits;
    unsigned int* pre = reinterpret_cast<unsigned int*>(dyn);
    unsigned char* want = dyn + (size_t)kRqMaxQ * bins * sizeof(unsigned int);
    rq_resolve_row(R, pre, want, n, args, B, hist2 + (size_t)row * kRqMaxQ * kRqMaxSubBins, sub_bits, cand_cap, list_cap,
                   cand_total ? cand_total + row : nullptr);
    if (R.fb || R.m_valid == 0ull || R.n_cells == 0) return;
    if (t < kRqMaxT) {   // (lsz[r][row][c] = keys of cell c that rank r holds: the ranks before this one come first)
        unsigned int base = 0u;
        if (lsz) for (int r = 0; r < rank; ++r) base += lsz[((size_t)r * n_rows + row) * kRqMaxT + t];
        cell_base[t] = base;
    }
    __syncthreads();
    const int nb = B.n_intervals;
    const double* crow = cand + (size_t)row * cand_cap;        // 16-byte aligned: cand_cap is even
    unsigned long long* lrow = glist + (size_t)row * list_cap;
    unsigned int* frow = gfill + (size_t)row * kRqMaxT;
    const unsigned int cnt = B.cand_count;
    // the interval ends live in registers (up to 8 intervals: the 7-quantile band set); the search is then pure VALU
    // and the only LDS traffic per candidate is its interval's (lo, shift) and the `want` byte
    unsigned long long hi_r[8];
#pragma unroll
    for (int v = 0; v < 8; ++v) hi_r[v] = v < nb ? R.ivhi[v] : ~0ull;
    const bool small = nb <= 8;
    auto take = [&](double x, bool ok) {
        if (!ok) return;
        const unsigned long long k = key_of(x);
        int b = 0;
        if (small) {
#pragma unroll
            for (int v = 0; v < 8; ++v) b += (k > hi_r[v]) ? 1 : 0;   // intervals are disjoint and ascending: k lies in one
        } else {
            for (int v = 0; v < nb; ++v) b += (k > R.ivhi[v]) ? 1 : 0;
        }
        const unsigned int c = want[b * bins + rq_sub_bin(k, x, R.ivlo[b], R.ivshift[b], R.ivxlo[b], R.ivinvw[b], bins)];
        if (c != 255u) {
            const unsigned int slot = cell_base[c] + atomicAdd(&frow[c], 1u);
            if (slot < R.cell_size[c]) lrow[R.cell_off[c] + slot] = k;
        }
    };
    // this workgroup's share of the row's candidates: four 16-byte loads in flight per lane
    typedef double d2_t __attribute__((ext_vector_type(2)));
    const d2_t* c2 = reinterpret_cast<const d2_t*>(crow);
    const unsigned int n_pairs = cnt / 2u;
    const unsigned int share = (n_pairs + gridDim.x - 1) / gridDim.x;
    const unsigned int first = blockIdx.x * share;
    const unsigned int last = first + share < n_pairs ? first + share : n_pairs;
    constexpr int kUnroll = 4;
    for (unsigned int base = first; base < last; base += 1024u * kUnroll) {
        d2_t v[kUnroll];
        bool ok[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const unsigned int i = base + (unsigned int)u * 1024u + (unsigned int)t;
            ok[u] = i < last;
            v[u] = ok[u] ? __builtin_nontemporal_load(&c2[i]) : d2_t{0.0, 0.0};
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) { take(v[u].x, ok[u]); take(v[u].y, ok[u]); }
    }
    if ((cnt & 1u) && blockIdx.x == 0 && t == 0) take(crow[cnt - 1u], true);
}

__global__ __launch_bounds__(1024) void rq_select_kernel(int64_t n, const RqArgs args, RqBracket* br2, const unsigned int* hist2,
                                                        int sub_bits, unsigned int cand_cap, unsigned int list_cap,
                                                        const unsigned long long* glist, const unsigned int* gfill, double* out,
                                                        unsigned long long* counts, unsigned int* row_fallback, unsigned int* fb_list,
                                                        unsigned int* fb_count, const unsigned int* cand_total, int row0) {
    extern __shared__ __align__(16) unsigned char dyn[];     // pre u32 [kRqMaxQ][bins] | list u64 [list_cap]
    __shared__ RqResolved R;
    __shared__ unsigned int whist[kRqWaves * 256];
    const int row = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    RqBracket& B = br2[row];
    const int bins = 1 << sub_bits;
    unsigned int* pre = reinterpret_cast<unsigned int*>(dyn);
    unsigned long long* list = reinterpret_cast<unsigned long long*>(dyn + (size_t)kRqMaxQ * bins * sizeof(unsigned int));
    rq_resolve_row(R, pre, nullptr, n, args, B, hist2 + (size_t)row * kRqMaxQ * kRqMaxSubBins, sub_bits, cand_cap, list_cap,
                   cand_total ? cand_total + row : nullptr);
    const int nt = 2 * args.n_q;
    const unsigned long long m = R.m_valid;
    if (!R.fb && m > 0) {   // every cell must have received exactly its keys
        if (t < R.n_cells && gfill[(size_t)row * kRqMaxT + t] != R.cell_size[t]) R.fb = 1u;
    }
    __syncthreads();
    if (R.fb) {
        // (row0: the kernel may run on a GROUP of rows whose per-row arrays arrive shifted; the fallback list holds row ids of the whole call)
        if (t == 0) { B.fallback = 1u; row_fallback[row] = 1u; fb_list[atomicAdd(fb_count, 1u)] = (unsigned int)(row0 + row); }
        return;
    }
    if (t == 0) { row_fallback[row] = 0u; if (counts) counts[row] = m; }
    if (m == 0) {   // all-NaN row -> NaN (pandas na_value)
        if (t < args.n_q) out[(size_t)row * args.n_q + t] = __longlong_as_double(0x7ff8000000000000LL);
        return;
    }
    {
        unsigned int total = 0u;
        if (R.n_cells > 0) total = R.cell_off[R.n_cells - 1] + R.cell_size[R.n_cells - 1];
        const unsigned long long* lrow = glist + (size_t)row * list_cap;
        for (unsigned int i = t; i < total; i += 1024u) list[i] = lrow[i];
    }
    __syncthreads();
    // one wave per target: MSD radix select (8-bit digits) of rank tgt_res inside its cell, starting at the first
    // digit in which the cell's bounds differ
    for (int tg = wave; tg < nt; tg += kRqWaves) {
        const int c = R.tgt_cell[tg];
        if (c < 0) continue;
        // key bins: the cell's bounds share their leading digits; value bins: start from the top digit
        const int csh = R.ivshift[R.cell_b[c]];
        const unsigned long long clo = csh >= 0 ? R.ivlo[R.cell_b[c]] + ((unsigned long long)R.cell_s[c] << csh) : 0ull;
        const unsigned long long chi = csh >= 0 ? clo + ((1ull << csh) - 1ull) : ~0ull;
        const unsigned long long diff = clo ^ chi;
        int pass = diff ? (__clzll((long long)diff) >> 3) : 8;
        unsigned long long prefix = pass ? (clo >> (64 - 8 * pass)) : 0ull;
        unsigned long long rank = R.tgt_res[tg];
        const unsigned long long* L = list + R.cell_off[c];
        const unsigned int len = R.cell_size[c];
        unsigned int* h = whist + wave * 256;
        for (; pass < 8; ++pass) {
            const int shift_digit = 56 - 8 * pass;
            h[lane] = 0u; h[lane + 64] = 0u; h[lane + 128] = 0u; h[lane + 192] = 0u;
            for (unsigned int i = lane; i < len; i += 64u) {
                const unsigned long long k = L[i];
                if (pass == 0 || (k >> (shift_digit + 8)) == prefix) atomicAdd(&h[(unsigned int)(k >> shift_digit) & 0xFFu], 1u);
            }
            const unsigned long long c0 = h[4 * lane], c1 = h[4 * lane + 1], c2 = h[4 * lane + 2], c3 = h[4 * lane + 3];
            unsigned long long incl = c0 + c1 + c2 + c3;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned long long up = (unsigned long long)__shfl_up((long long)incl, off, 64);
                incl += lane >= off ? up : 0ull;
            }
            const unsigned long long excl = incl - (c0 + c1 + c2 + c3);
            const unsigned long long hit = __ballot(rank >= excl && rank < incl);   // exactly one lane (rank < cell size)
            const int Lh = hit ? __ffsll((long long)hit) - 1 : 63;
            int d = 4 * lane;
            unsigned long long cum = excl;
            if (rank >= cum + c0) { cum += c0; ++d; if (rank >= cum + c1) { cum += c1; ++d; if (rank >= cum + c2) { cum += c2; ++d; } } }
            d = __shfl(d, Lh, 64);
            cum = (unsigned long long)__shfl((long long)cum, Lh, 64);
            rank -= cum;
            prefix = (prefix << 8) | (unsigned long long)d;
        }
        if (lane == 0) R.tgt_key[tg] = prefix;
    }
    __syncthreads();
    if (t < args.n_q) {
        const double a = value_of(R.tgt_key[2 * t]), b = value_of(R.tgt_key[2 * t + 1]), g = R.gamma_q[t];
        const double diff = b - a;                       // _lerp
        double r = a + diff * g;
        if (g >= 0.5) r = b - diff * (1.0 - g);
        out[(size_t)row * args.n_q + t] = r;
    }
}

// ---- rows sharded over ranks: what crosses ranks between the stages of the bracketed route --------------------
// Dense counter block of one level: per row pos_count[2 * kRqMaxQ + 1] | n_nan | candidates stored | "my candidate
// buffer overflowed".  Every rank packs its own, the caller sums the block (and the level's sub-histograms) across
// ranks, every rank unpacks the sums.
constexpr int kRqCntWords = 2 * kRqMaxQ + 4;
constexpr int kRqMaxWorld = 64;
__global__ void rq_pack_kernel(const RqBracket* br, unsigned long long* cnt, int n_rows, unsigned int cand_cap) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    const RqBracket& B = br[row];
    unsigned long long* c = cnt + (size_t)row * kRqCntWords;
    for (int k = 0; k <= 2 * kRqMaxQ; ++k) c[k] = B.pos_count[k];
    c[2 * kRqMaxQ + 1] = B.n_nan;
    c[2 * kRqMaxQ + 2] = B.cand_count;
    c[2 * kRqMaxQ + 3] = B.cand_count > cand_cap ? 1ull : 0ull;
}
__global__ void rq_unpack_kernel(RqBracket* br, const unsigned long long* cnt, unsigned int* cand_total, int n_rows) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    RqBracket& B = br[row];
    const unsigned long long* c = cnt + (size_t)row * kRqCntWords;
    for (int k = 0; k <= 2 * kRqMaxQ; ++k) B.pos_count[k] = c[k];
    B.n_nan = c[2 * kRqMaxQ + 1];
    cand_total[row] = c[2 * kRqMaxQ + 2] > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned int)c[2 * kRqMaxQ + 2];
    if (c[2 * kRqMaxQ + 3]) B.fallback = 1u;           // some rank's candidates overflowed: every rank drops the row together
}
// After the global counts are in: the cells of every row (identical on all ranks) and how many keys of each THIS rank
// holds (from its own sub-histograms, kept aside before they were summed).
__global__ __launch_bounds__(1024) void rq_cells_kernel(int64_t n, const RqArgs args, const RqBracket* br2, const unsigned int* hist2,
                                                       const unsigned int* hist_local, int sub_bits, unsigned int cand_cap,
                                                       unsigned int list_cap, const unsigned int* cand_total, unsigned int* lsz,
                                                       int rank, int n_rows) {
    extern __shared__ __align__(16) unsigned char dyn[];     // pre u32 [kRqMaxQ][bins]
    __shared__ RqResolved R;
    const int row = blockIdx.x, t = threadIdx.x;
    rq_resolve_row(R, reinterpret_cast<unsigned int*>(dyn), nullptr, n, args, br2[row], hist2 + (size_t)row * kRqMaxQ * kRqMaxSubBins,
                   sub_bits, cand_cap, list_cap, cand_total + row);
    if (R.fb || R.m_valid == 0ull) return;
    if (t < R.n_cells)
        lsz[((size_t)rank * n_rows + row) * kRqMaxT + t] =
            hist_local[(size_t)row * kRqMaxQ * kRqMaxSubBins + (size_t)R.cell_b[t] * kRqMaxSubBins + R.cell_s[t]];
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

// Streaming helper of the two K2 passes: every lane takes PAIRS of paths — one 16-byte value load and one 2-byte flag
// load per pair, four pairs in flight — and hands each successful value to `f`.  (values 16-byte, flags 2-byte aligned:
// the caller checks; a stray last element is taken by one lane.)
template <typename F>
__device__ __forceinline__ void k2_stream(const double* __restrict__ v, const uint8_t* __restrict__ ok, int64_t n, bool vec2, F&& f) {
    const int64_t step = (int64_t)gridDim.x * 256;
    if (vec2) {
        typedef double d2_t __attribute__((ext_vector_type(2)));
        const d2_t* v2 = reinterpret_cast<const d2_t*>(v);
        const unsigned short* ok2 = reinterpret_cast<const unsigned short*>(ok);
        const int64_t n_pairs = n / 2;
        constexpr int kUnroll = 4;
        for (int64_t base = (int64_t)blockIdx.x * 256 + threadIdx.x; base < n_pairs; base += step * kUnroll) {
            d2_t x[kUnroll];
            unsigned int fl[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const int64_t i = base + u * step;
                const bool in = i < n_pairs;
                x[u] = in ? __builtin_nontemporal_load(&v2[i]) : d2_t{0.0, 0.0};
                fl[u] = in ? (unsigned int)ok2[i] : 0u;
            }
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                if (fl[u] & 0xFFu) f(x[u].x);
                if (fl[u] >> 8) f(x[u].y);
            }
        }
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0 && ok[n - 1]) f(v[n - 1]);
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += step)
            if (ok[i]) f(v[i]);
    }
}

__global__ __launch_bounds__(256) void minmax_kernel(const double* __restrict__ v, const uint8_t* __restrict__ ok,
                                                    int64_t n, double* minmax, int vec2) {
    __shared__ double smin[4], smax[4];
    double lo = __longlong_as_double(0x7ff0000000000000LL), hi = -lo;
    k2_stream(v, ok, n, vec2 != 0, [&](double x) { lo = fmin(lo, x); hi = fmax(hi, x); });
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

// Summary-statistic rows of the response document (server.py:446-458; simulation.py:78-96).  HBM-bound:
// 25 B read + 32 B written per path.  Plain IEEE division and multiply: the rate is (w / start) * 100.0,
// rounded twice like the pandas expression.
__global__ __launch_bounds__(256) void stat_rows_kernel(const double* __restrict__ start, const double* __restrict__ fin,
                                                       const double* __restrict__ fy_real, const uint8_t* __restrict__ ok,
                                                       int64_t n, double* __restrict__ rows, int64_t stride) {
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double s = start[i], f = fin[i];
        rows[i] = s;
        rows[stride + i] = f;
        rows[2 * stride + i] = ok[i] ? f : nan;
        rows[3 * stride + i] = (s > 1e-6) ? (fy_real[i] / s) * 100.0 : nan;
    }
}

// np.histogram(x, bins=n_bins) over [lo, hi] (numpy/lib/_histograms_impl.py, uniform-bin fast path):
// index = int((x - lo) / (hi - lo) * n_bins), the right edge belongs to the last bin, then a
// +-1 correction against the linspace edges.
__global__ __launch_bounds__(256) void hist_kernel(const double* __restrict__ v, const uint8_t* __restrict__ ok,
                                                  int64_t n, const double* __restrict__ minmax, int n_bins,
                                                  unsigned long long* bins, int vec2) {
    extern __shared__ unsigned int lbins[];
    for (int k = threadIdx.x; k < n_bins; k += 256) lbins[k] = 0u;
    __syncthreads();
    double lo = minmax[0], hi = minmax[1];
    if (lo == hi) { lo = lo - 0.5; hi = hi + 0.5; }  // _get_outer_edges: degenerate range widened
    const double denom = hi - lo;
    const double step = denom / (double)n_bins;      // np.linspace step
    k2_stream(v, ok, n, vec2 != 0, [&](double x) {
        if (!(x >= lo && x <= hi)) return;
        int idx = (int)(((x - lo) / denom) * (double)n_bins);
        if (idx == n_bins) idx -= 1;
        const double e_lo = idx == n_bins ? hi : lo + (double)idx * step;
        if (x < e_lo) idx -= 1;
        else {
            const double e_hi = (idx + 1) == n_bins ? hi : lo + (double)(idx + 1) * step;
            if (x >= e_hi && idx != n_bins - 1) idx += 1;
        }
        atomicAdd(&lbins[idx], 1u);
    });
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

static unsigned int rq_cand_cap(int64_t n) { return (unsigned int)(n / 64 + 4096); }

static unsigned int rq_bracket_cand_cap(int64_t n) { return (unsigned int)((n / 8 + 4096) & ~(int64_t)1); }  // even: 16-byte rows

struct RqLayout {
    RqRow* st;
    unsigned int* hist;
    unsigned int* aux;
    unsigned long long* cand;
    size_t reduce_offset, reduce_words;
    // bracketed single pass (mcr_row_quantiles only): coarse / fine brackets and their sub-histograms
    RqBracket* br1;
    RqBracket* br2;
    unsigned int* hist1;        // [n_rows][kRqMaxQ][kRqMaxSubBins]
    unsigned int* hist2;
    unsigned long long* glist;  // [n_rows][kRqListCap]: keys of the wanted cells
    unsigned int* gfill;        // [n_rows][kRqMaxT]: keys appended per cell
    unsigned long long* cnt;    // rows sharded over ranks: [n_rows][kRqCntWords] counter block that is summed across ranks
    unsigned int* cand_total;   //   [n_rows] candidates of all ranks
    unsigned int* lsz;          //   [kRqMaxWorld][n_rows][kRqMaxT] keys of every cell held by every rank
    size_t glist_bytes, gfill_words, hist_words, br_bytes, cnt_words, lsz_words;
    unsigned int* row_fallback;
    unsigned int* fb_list;      // rows that take the full radix passes, and how many
    unsigned int* fb_count;
    double* bcand;
    size_t total_bytes;
};
static size_t rq_align16(size_t x) { return (x + 15) & ~(size_t)15; }
// Offsets first (also for a null base: mcr_row_quantiles_scratch_bytes), pointers after.  The stepwise
// (multi-GPU) entry points pass n = 0: they only use st | hist | aux | cand, whose offsets do not depend on n.
static RqLayout rq_layout(void* scratch, int32_t n_rows, int64_t n = 0) {
    RqLayout L;
    L.reduce_offset = (size_t)n_rows * sizeof(RqRow);
    L.reduce_words = (size_t)n_rows * kRqMaxT * 256 + (size_t)n_rows * 2;
    const size_t off_aux = L.reduce_offset + (size_t)n_rows * kRqMaxT * 256 * sizeof(unsigned int);
    const size_t off_cand = L.reduce_offset + ((L.reduce_words + 1) & ~(size_t)1) * sizeof(unsigned int);  // 8-byte aligned
    const size_t off_br = rq_align16(off_cand + (size_t)n_rows * (size_t)rq_cand_cap(n) * sizeof(unsigned long long));
    const size_t off_rn = rq_align16(off_br + (size_t)n_rows * 2 * sizeof(RqBracket));
    const size_t off_h = rq_align16(off_rn + ((size_t)n_rows * 2 + 1) * sizeof(unsigned int));
    const size_t hist_bytes = (size_t)n_rows * kRqMaxQ * kRqMaxSubBins * sizeof(unsigned int);
    const size_t off_gl = rq_align16(off_h + 2 * hist_bytes);
    const size_t off_gf = rq_align16(off_gl + (size_t)n_rows * kRqListCap * sizeof(unsigned long long));
    const size_t off_cn = rq_align16(off_gf + (size_t)n_rows * kRqMaxT * sizeof(unsigned int));
    const size_t off_ct = rq_align16(off_cn + (size_t)n_rows * kRqCntWords * sizeof(unsigned long long));
    const size_t off_ls = rq_align16(off_ct + (size_t)n_rows * sizeof(unsigned int));
    const size_t off_bc = rq_align16(off_ls + (size_t)kRqMaxWorld * n_rows * kRqMaxT * sizeof(unsigned int));
    L.total_bytes = off_bc + (size_t)n_rows * (size_t)rq_bracket_cand_cap(n) * sizeof(double);
    L.glist_bytes = (size_t)n_rows * kRqListCap * sizeof(unsigned long long);
    L.gfill_words = (size_t)n_rows * kRqMaxT;
    L.hist_words = (size_t)n_rows * kRqMaxQ * kRqMaxSubBins;
    L.br_bytes = (size_t)n_rows * sizeof(RqBracket);
    L.cnt_words = (size_t)n_rows * kRqCntWords;
    L.lsz_words = (size_t)kRqMaxWorld * n_rows * kRqMaxT;
    char* base = (char*)scratch;
    L.st = (RqRow*)base;
    L.hist = (unsigned int*)(base + L.reduce_offset);
    L.aux = (unsigned int*)(base + off_aux);
    L.cand = (unsigned long long*)(base + off_cand);
    L.br1 = (RqBracket*)(base + off_br);
    L.br2 = L.br1 + n_rows;
    L.row_fallback = (unsigned int*)(base + off_rn);
    L.fb_list = L.row_fallback + n_rows;
    L.fb_count = L.fb_list + n_rows;
    L.hist1 = (unsigned int*)(base + off_h);
    L.hist2 = (unsigned int*)(base + off_h + hist_bytes);
    L.glist = (unsigned long long*)(base + off_gl);
    L.gfill = (unsigned int*)(base + off_gf);
    L.cnt = (unsigned long long*)(base + off_cn);
    L.cand_total = (unsigned int*)(base + off_ct);
    L.lsz = (unsigned int*)(base + off_ls);
    L.bcand = (double*)(base + off_bc);
    return L;
}

static int rq_check(const void* scratch, int32_t n_rows, int64_t n_local, int32_t n_q) {
    if (!scratch || n_rows <= 0 || n_rows > 65535 || n_q <= 0 || n_q > kRqMaxQ) { set_error("bad row-quantile arguments"); return MCR_ERR_INVALID_ARG; }
    if (n_local < 0 || n_local >= ((int64_t)1 << 32)) { set_error("n must be < 2^32 per row"); return MCR_ERR_INVALID_ARG; }
    return MCR_OK;
}

// kernels whose dynamic LDS request can exceed the 64 KB default
static void rq_opt_in_lds(int device) {
    static thread_local int lds_opt_in_device = -1;
    if (lds_opt_in_device == device) return;
    const int big = 160 * 1024 - 24 * 1024;
    (void)hipFuncSetAttribute((const void*)rq_select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void*)rq_collect_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute((const void*)rq_cells_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute((const void*)rq_refine_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute((const void*)(rq_count_kernel<32, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    (void)hipFuncSetAttribute((const void*)(rq_count_kernel<32, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    (void)hipFuncSetAttribute((const void*)(rq_count_kernel<16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute((const void*)(rq_count_kernel<16, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute((const void*)(rq_slab_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute((const void*)(rq_slab_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    (void)hipGetLastError();
    lds_opt_in_device = device;
}

// One counting pass over the first `len` entries of every row.
static void rq_count_pass(hipStream_t s, const double* rows, int64_t row_stride, int32_t n_rows, int64_t len, RqBracket* br,
                          unsigned int* hist, int sub_bits, bool compact, int n_q, double* bcand, unsigned int bcap) {
    const bool small_p = 2 * n_q < 16;                        // bound table: the next power of two above 2 * (intervals <= quantiles)
    // slab pass: ~4096 workgroups (1280 .. 16384 measured flat within 2 % at 136 rows x 1e7); sample pass: few per row — every workgroup flushes its sub-histograms
    // (up to 1024 bins per interval) with global atomics, and the sample is only 1/32 of the slab
    int per_row = compact ? (4096 / n_rows > 0 ? 4096 / n_rows : 1) : (1024 / n_rows > 1 ? 1024 / n_rows : 2);
    if (const char* e = std::getenv(compact ? "MCR_RQ_SLAB_WGS" : "MCR_RQ_SAMPLE_WGS")) { const int t = std::atoi(e); if (t > 0) per_row = t / n_rows > 0 ? t / n_rows : 1; }
    const int bx = grid_for(len > 0 ? len : 1, kRqBlock * 16, per_row);
    const dim3 grid(bx, n_rows), block(kRqBlock);
    const int max_q = n_q;
    const size_t lds = (compact ? (size_t)(kRqBlock / 64) * kRqWaveStage * sizeof(double) : 0) +
                       (size_t)(2 * max_q + 1) * kRqBlock * sizeof(unsigned int) + (size_t)max_q * ((size_t)1 << sub_bits) * sizeof(unsigned int);
    // the slab pass: rows whose bounds fit the bucket table take rq_slab_kernel, the others the generic kernel (whose
    // workgroups return at once for the rows they do not own, and vice versa).  MCR_RQ_SLAB_LUT=0 keeps everything generic.
    static const bool use_lut = [] { const char* e = std::getenv("MCR_RQ_SLAB_LUT"); return !(e && e[0] == '0'); }();
    const int skip = (compact && use_lut) ? 1 : 0;     // (the sample pass stays generic: most of ITS entries are bracket members, tallied in place either way; measured 103 vs 125 us)
#define MCR_COUNT(P, C) hipLaunchKernelGGL((rq_count_kernel<P, C>), grid, block, lds, s, rows, row_stride, len, br, hist, sub_bits, max_q, bcand, bcap, skip)
    if (small_p) { if (compact) MCR_COUNT(16, true); else MCR_COUNT(16, false); }
    else { if (compact) MCR_COUNT(32, true); else MCR_COUNT(32, false); }
#undef MCR_COUNT
    if (skip) {
        const size_t lds2 = (size_t)(kRqBlock / 64) * 192 * sizeof(double) + (size_t)(2 * max_q + 2) * kRqBlock * sizeof(unsigned int) +
                            (size_t)max_q * ((size_t)1 << sub_bits) * sizeof(unsigned int);
        if (small_p) hipLaunchKernelGGL((rq_slab_kernel<16>), grid, block, lds2, s, rows, row_stride, len, br, hist, sub_bits, max_q, bcand, bcap);
        else hipLaunchKernelGGL((rq_slab_kernel<32>), grid, block, lds2, s, rows, row_stride, len, br, hist, sub_bits, max_q, bcand, bcap);
    }
}

}  // namespace mcr
using namespace mcr;
extern "C" {

int64_t mcr_row_quantiles_scratch_bytes(int32_t n_rows, int32_t n_q, int64_t n) {
    if (n_rows <= 0 || n_q <= 0 || n_q > kRqMaxQ || n < 0 || n >= ((int64_t)1 << 32)) return 0;
    return (int64_t)rq_layout(nullptr, n_rows, n).total_bytes;
}

int64_t mcr_row_quantiles_reduce_block(int32_t n_rows, int64_t* n_words) {
    if (n_rows <= 0) return -1;
    if (n_words) *n_words = (int64_t)n_rows * kRqMaxT * 256 + (int64_t)n_rows * 2;
    return (int64_t)((size_t)n_rows * sizeof(RqRow));
}

int mcr_row_quantiles_begin(void* scratch, int32_t n_rows, int device, void* hip_stream) {
    MCR_ENTER_DEVICE(device);
    int rc = MCR_OK;
    rc = rq_check(scratch, n_rows, 0, 1);
    if (rc != MCR_OK) return rc;
    const RqLayout L = rq_layout(scratch, n_rows);
    hipLaunchKernelGGL(rq_init_kernel, dim3(grid_for((int64_t)L.reduce_words, 256, 1024)), dim3(256), 0, (hipStream_t)hip_stream,
                       L.st, L.hist, n_rows);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "rq_init_kernel");
    return MCR_OK;
}

static int rq_hist_step(const double* rows, int64_t row_stride, int32_t n_rows, int64_t n_local, int32_t n_q,
                        int32_t pass, void* scratch, int device, void* hip_stream, int track_minmax,
                        const unsigned int* row_list = nullptr, int n_list = 0);

int mcr_row_quantiles_hist(const double* rows, int64_t row_stride, int32_t n_rows, int64_t n_local, int32_t n_q,
                           int32_t pass, void* scratch, int device, void* hip_stream) {
    // rows sharded across ranks: no all-equal-row shortcut (a row's min/max is not summable across ranks)
    return rq_hist_step(rows, row_stride, n_rows, n_local, n_q, pass, scratch, device, hip_stream, 0);
}

static int rq_hist_step(const double* rows, int64_t row_stride, int32_t n_rows, int64_t n_local, int32_t n_q,
                        int32_t pass, void* scratch, int device, void* hip_stream, int track_minmax,
                        const unsigned int* row_list, int n_list) {
    MCR_ENTER_DEVICE(device);
    int rc = MCR_OK;
    rc = rq_check(scratch, n_rows, n_local, n_q);
    if (rc != MCR_OK) return rc;
    if (pass < 0 || pass > 7) { set_error("pass out of range"); return MCR_ERR_INVALID_ARG; }
    if (n_local > 0 && (!rows || row_stride < n_local)) { set_error("bad rows / row_stride"); return MCR_ERR_INVALID_ARG; }
    hipStream_t s = (hipStream_t)hip_stream;
    const RqLayout L = rq_layout(scratch, n_rows);
    const unsigned int cap = rq_cand_cap(n_local);
    const int64_t n = n_local;
    if (n > 0) {
        // each workgroup streams >= 16 elements per lane; ~4096 workgroups in flight fill the 256 CUs
        const int gy = row_list ? n_list : n_rows;       // a call on a row list launches exactly those rows
        int bx = grid_for(n, kRqBlock * 16, 4096 / gy > 0 ? 4096 / gy : 1);
        const dim3 grid(bx, gy), block(kRqBlock);
        // rows whose candidates overflowed (big ties) re-stream alone; the workgroups of all other rows exit at
        // once, and since that is the common case the grid stays modest (idle workgroups cost ~10 ns each)
        const dim3 grid_slow(grid_for(n, kRqBlock * 16, 512), gy);
        const size_t lds_groups = (size_t)(2 * n_q) * 256 * sizeof(unsigned int);  // <= 2 targets per quantile
        const size_t lds_first = 256 * sizeof(unsigned int);
        if (pass == 0) {
            hipLaunchKernelGGL((rq_hist_kernel<true, false>), grid, block, lds_first, s, rows, row_stride, n, pass, L.st, L.hist, L.aux, L.cand, cap, 0, track_minmax, row_list);
        } else if (pass < 3) {
            hipLaunchKernelGGL((rq_hist_kernel<false, false>), grid, block, lds_groups, s, rows, row_stride, n, pass, L.st, L.hist, L.aux, L.cand, cap, 0, 0, row_list);
        } else if (pass == 3) {
            hipLaunchKernelGGL((rq_hist_kernel<false, true>), grid, block, lds_groups, s, rows, row_stride, n, pass, L.st, L.hist, L.aux, L.cand, cap, 0, 0, row_list);
        } else {
            hipLaunchKernelGGL(rq_cand_hist_kernel, dim3(16, gy), block, lds_groups, s, pass, L.st, L.hist, L.aux, L.cand, cap, row_list);
            hipLaunchKernelGGL((rq_hist_kernel<false, false>), grid_slow, block, lds_groups, s, rows, row_stride, n, pass, L.st, L.hist, L.aux, L.cand, cap, 1, 0, row_list);
        }
    }
    if (pass == 3) hipLaunchKernelGGL(rq_flag_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, s, L.st, L.aux, n_rows, cap);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "row quantile histogram step");
    return MCR_OK;
}

static int rq_scan_step(int32_t n_rows, int64_t n_total, const double* q, int32_t n_q, int32_t pass, double* out,
                        uint64_t* counts, void* scratch, int device, void* hip_stream,
                        const unsigned int* row_list, int n_list) {
    MCR_ENTER_DEVICE(device);
    int rc = MCR_OK;
    rc = rq_check(scratch, n_rows, 0, n_q);
    if (rc != MCR_OK) return rc;
    if (!q || !out || pass < 0 || pass > 7 || n_total <= 0 || n_total >= ((int64_t)1 << 32)) { set_error("bad arguments"); return MCR_ERR_INVALID_ARG; }
    for (int j = 0; j < n_q; ++j)
        if (!(q[j] >= 0.0 && q[j] <= 1.0)) { set_error("quantile %d out of [0,1]", j); return MCR_ERR_INVALID_ARG; }
    const RqLayout L = rq_layout(scratch, n_rows);
    RqArgs a;
    std::memset(&a, 0, sizeof(a));
    a.n_q = n_q;
    for (int j = 0; j < n_q; ++j) a.q[j] = q[j];
    const size_t lds_groups = (size_t)(2 * n_q) * 256 * sizeof(unsigned int);
    hipLaunchKernelGGL(rq_scan_kernel, dim3(row_list ? n_list : n_rows), dim3(kRqScanBlock), lds_groups, (hipStream_t)hip_stream, n_total, pass,
                       L.st, L.hist, a, out, (unsigned long long*)counts, L.aux, row_list);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "rq_scan_kernel");
    return MCR_OK;
}

int mcr_row_quantiles_scan(int32_t n_rows, int64_t n_total, const double* q, int32_t n_q, int32_t pass, double* out,
                           uint64_t* counts, void* scratch, int device, void* hip_stream) {
    return rq_scan_step(n_rows, n_total, q, n_q, pass, out, counts, scratch, device, hip_stream, nullptr, 0);
}

// The full radix select (8 digit passes) over rows of length n: all rows, or the n_list rows listed in row_list.
static int rq_radix_select(const double* rows, int64_t row_stride, int32_t n_rows, int64_t n, const double* q, int32_t n_q,
                           double* out, uint64_t* counts, void* scratch, int device, void* hip_stream,
                           const unsigned int* row_list, int n_list) {
    int rc = MCR_OK;
    for (int pass = 0; pass < 8 && rc == MCR_OK; ++pass) {
        rc = rq_hist_step(rows, row_stride, n_rows, n, n_q, pass, scratch, device, hip_stream, /*track_minmax=*/1, row_list, n_list);
        if (rc == MCR_OK) rc = rq_scan_step(n_rows, n, q, n_q, pass, out, counts, scratch, device, hip_stream, row_list, n_list);
    }
    return rc;
}

// Rows at least this long take the bracketed single pass.  Measured on 136 rows (tools/k3_time.py, either route forced):
// the six-launch route wins from the smallest row it accepts — 8 192 entries 0.21 vs 0.24 ms, 131 072: 0.28 vs 0.36,
// 10^6: 0.48 vs 1.11, 2 10^6: 0.71 vs 2.01 — (round 1's 45-launch version needed 2^21 entries to pay).  16 384 keeps a
// margin for rows whose brackets fail and take both routes.  MCR_RQ_BRACKET_MIN_N overrides it (tests force either route).
static int64_t rq_bracket_min_n() {
    const char* e = std::getenv("MCR_RQ_BRACKET_MIN_N");
    if (e && *e) return (int64_t)std::strtoll(e, nullptr, 10);
    return (int64_t)1 << 14;
}

static thread_local int g_last_fallback_rows = -1;
int mcr_row_quantiles_last_fallback_rows(void) { return g_last_fallback_rows; }

int mcr_row_quantiles(const double* rows, int64_t row_stride, int32_t n_rows, int64_t n, const double* q,
                      int32_t n_q, double* out, uint64_t* counts, void* scratch, int device, void* hip_stream) {
    g_last_fallback_rows = -1;
    if (!rows || n <= 0) { set_error("bad arguments"); return MCR_ERR_INVALID_ARG; }
    MCR_ENTER_DEVICE(device);
    int rc = MCR_OK;
    if (n < rq_bracket_min_n() || n < kRqTiny || 2 * n_q >= 32) {
        rc = mcr_row_quantiles_begin(scratch, n_rows, device, hip_stream);
        if (rc != MCR_OK) return rc;
        return rq_radix_select(rows, row_stride, n_rows, n, q, n_q, out, counts, scratch, device, hip_stream, nullptr, 0);
    }
    rc = rq_check(scratch, n_rows, n, n_q);
    if (rc != MCR_OK) return rc;
    if (!q || !out || row_stride < n) { set_error("bad arguments"); return MCR_ERR_INVALID_ARG; }
    for (int j = 0; j < n_q; ++j)
        if (!(q[j] >= 0.0 && q[j] <= 1.0)) { set_error("quantile %d out of [0,1]", j); return MCR_ERR_INVALID_ARG; }
    hipStream_t s = (hipStream_t)hip_stream;
    const RqLayout L = rq_layout(scratch, n_rows, n);
    RqArgs a;
    std::memset(&a, 0, sizeof(a));
    a.n_q = n_q;
    for (int j = 0; j < n_q; ++j) a.q[j] = q[j];
    rq_opt_in_lds(device);
    int64_t m = n / 32;           // the second sample: the first n/32 entries of every row
    if (const char* e = std::getenv("MCR_RQ_SAMPLE_DIV")) { const int t = std::atoi(e); if (t >= 2) m = n / t; }
    if (m < 65536) m = 65536;
    if (m > n) m = n;
    m &= ~(int64_t)1;
    const int sub_bits2 = n <= ((int64_t)1 << 24) ? 8 : 10;   // cells of a few hundred keys either way
    const unsigned int bcap = rq_bracket_cand_cap(n);
    const size_t bins = (size_t)1 << sub_bits2;
    const unsigned int list_cap = sub_bits2 == 8 ? kRqListCap : kRqListCap / 2;
    int collect_per_row = n >= ((int64_t)1 << 23) ? 4 : (n >= ((int64_t)1 << 21) ? 4 : 2);   // a few MB of candidates per row
    if (const char* e = std::getenv("MCR_RQ_COLLECT_PER_ROW")) { const int t = std::atoi(e); if (t > 0) collect_per_row = t; }
    const size_t hrow = (size_t)kRqMaxQ * kRqMaxSubBins;
    // The six stages for rows [r0, r0 + nr) on stream `st` (every per-row array shifted to the group's first row).  Every caller
    // passes the whole call (r0 = 0, nr = n_rows) since round 3 measured the row-group pipelining and dropped it (below); the
    // row-range form of the stages is kept because it costs nothing and documents what a stage depends on.
    auto head = [&](hipStream_t st, int r0, int nr, unsigned int* fb_reset) {
        const double* rg = rows + (int64_t)r0 * row_stride;
        // (1) coarse brackets from the first kRqTiny entries of every row, sorted in LDS (paths are exchangeable; an
        //     unlucky or adversarial prefix only costs the affected rows the fallback below)
        hipLaunchKernelGGL(rq_tiny_kernel, dim3(nr), dim3(1024), 0, st, rg, row_stride, n, a, L.br1 + r0, L.hist1 + r0 * hrow, fb_reset);
        // (2) the counting pass over the sample, (3) fine brackets from its counts
        //     (measured and dropped, round 3: letting the LAST workgroup of each row's sample pass refine that row on the spot
        //      — one launch less — made the pass 120 us slower than pass + refine kernel: the fused kernel needs scratch memory)
        rq_count_pass(st, rg, row_stride, nr, m, L.br1 + r0, L.hist1 + r0 * hrow, kRqCoarseSubBits, false, n_q, L.bcand + (size_t)r0 * bcap, bcap);
        hipLaunchKernelGGL(rq_refine_kernel, dim3(nr), dim3(256), (size_t)kRqMaxQ * kRqMaxSubBins * sizeof(unsigned int), st, m, a, L.br1 + r0,
                           L.hist1 + r0 * hrow, L.br2 + r0, L.hist2 + r0 * hrow, sub_bits2, L.gfill + (size_t)r0 * kRqMaxT);
    };
    auto slab = [&](hipStream_t st, int r0, int nr) {   // (4) the one pass over the slab
        rq_count_pass(st, rows + (int64_t)r0 * row_stride, row_stride, nr, n, L.br2 + r0, L.hist2 + r0 * hrow, sub_bits2, true, n_q,
                      L.bcand + (size_t)r0 * bcap, bcap);
    };
    auto tail = [&](hipStream_t st, int r0, int nr) {   // (5) ranks -> cells; the candidates of the wanted cells; (6) selection in LDS -> interpolation
        hipLaunchKernelGGL(rq_collect_kernel, dim3(collect_per_row, nr), dim3(1024), (size_t)kRqMaxQ * bins * 5, st, n, a, L.br2 + r0, L.hist2 + r0 * hrow,
                           sub_bits2, L.bcand + (size_t)r0 * bcap, bcap, list_cap, L.glist + (size_t)r0 * list_cap, L.gfill + (size_t)r0 * kRqMaxT,
                           (const unsigned int*)nullptr, (const unsigned int*)nullptr, 0, nr);
        const size_t lds = (size_t)kRqMaxQ * bins * sizeof(unsigned int) + (size_t)list_cap * sizeof(unsigned long long);
        hipLaunchKernelGGL(rq_select_kernel, dim3(nr), dim3(1024), lds, st, n, a, L.br2 + r0, L.hist2 + r0 * hrow, sub_bits2, bcap, list_cap,
                           L.glist + (size_t)r0 * list_cap, L.gfill + (size_t)r0 * kRqMaxT, out + (size_t)r0 * n_q,
                           counts ? (unsigned long long*)counts + r0 : nullptr, L.row_fallback + r0, L.fb_list, L.fb_count, (const unsigned int*)nullptr, r0);
    };
    // (Measured and dropped, round 3: cutting the rows into G groups on G streams, chained so that group g's slab pass starts
    //  when group g - 1's ends and the latency-bound kernels of the neighbouring groups run under it.  136 rows x 1e7, A/B in
    //  one process: G = 1 2.262 ms, G = 2 2.345, G = 3 2.478, G = 4 2.717 — cross-stream event waits cost more than the
    //  0.35 ms of small kernels they were meant to hide.)
    hipError_t e = hipSuccess;
    head(s, 0, n_rows, L.fb_count);
    slab(s, 0, n_rows);
    tail(s, 0, n_rows);
    // (6) rows the brackets could not decide (a target outside its bracket, candidates overflowing on a wide tie
    //     that straddles a bracket end, a sample without data) take the full passes.  How many is only known on the
    //     device: this route reads one word back (ONE stream synchronisation per call) rather than launch eight
    //     passes of idle workgroups.
    unsigned int n_fb = 0;
    e = hipMemcpyAsync(&n_fb, L.fb_count, sizeof(n_fb), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return hip_fail(e, "bracketed row quantiles");
    g_last_fallback_rows = (int)n_fb;
    if (n_fb > 0) {
        rc = mcr_row_quantiles_begin(scratch, n_rows, device, hip_stream);
        if (rc != MCR_OK) return rc;
        rc = rq_radix_select(rows, row_stride, n_rows, n, q, n_q, out, counts, scratch, device, hip_stream, L.fb_list, (int)n_fb);
        if (rc != MCR_OK) return rc;
    }
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "bracketed row quantiles");
    return MCR_OK;
}

// The bracketed route for rows SHARDED over ranks (every rank holds n_local of the n_total entries of each row).  Same
// stages as mcr_row_quantiles; between them the caller's `reduce` sums small blocks across ranks in place: the coarse
// brackets of rank 0 (a sum in which the other ranks contribute zeros), the counter block and sub-histograms of each
// counting pass, the per-rank cell sizes, the cell lists (disjoint slots: a sum is a gather) and their fill counts.
// The slab never moves; every rank ends with the same exact quantiles.
int mcr_row_quantiles_sharded(const double* rows, int64_t row_stride, int32_t n_rows, int64_t n_local, int64_t n_total,
                              const double* q, int32_t n_q, double* out, uint64_t* counts, void* scratch, int32_t rank,
                              int32_t world, mcr_reduce_fn reduce, void* reduce_ctx, int device, void* hip_stream) {
    g_last_fallback_rows = -1;
    MCR_ENTER_DEVICE(device);
    int rc = rq_check(scratch, n_rows, n_local, n_q);
    if (rc != MCR_OK) return rc;
    if (!reduce || world < 1 || world > kRqMaxWorld || rank < 0 || rank >= world) { set_error("bad rank / world / reduce callback"); return MCR_ERR_INVALID_ARG; }
    if (!q || !out || n_total <= 0 || n_total >= ((int64_t)1 << 32) || n_local > n_total || (n_local > 0 && (!rows || row_stride < n_local))) {
        set_error("bad arguments");
        return MCR_ERR_INVALID_ARG;
    }
    if (2 * n_q >= 32 || n_total < kRqTiny) {   // (identical arguments on every rank: every rank returns here together)
        set_error("shape not supported by the sharded bracketed route (use the stepwise radix select)");
        return MCR_ERR_UNSUPPORTED;
    }
    for (int j = 0; j < n_q; ++j)
        if (!(q[j] >= 0.0 && q[j] <= 1.0)) { set_error("quantile %d out of [0,1]", j); return MCR_ERR_INVALID_ARG; }
    hipStream_t s = (hipStream_t)hip_stream;
    const RqLayout L = rq_layout(scratch, n_rows, n_local);
    // Eligibility that depends on ONE rank's shard (rank 0 must hold the first sample) is decided COLLECTIVELY: every rank
    // contributes a flag to the first reduction and all of them leave together — a rank that returned on its own would
    // leave its peers waiting inside the caller's all-reduce.
    {
        const unsigned int mine = (rank == 0 && n_local < kRqTiny) ? 1u : 0u;
        unsigned int total = 0u;
        hipError_t e0 = hipMemcpyAsync(L.fb_count, &mine, sizeof(mine), hipMemcpyHostToDevice, s);
        if (e0 == hipSuccess) e0 = hipStreamSynchronize(s);       // (`mine` is a stack variable)
        if (e0 != hipSuccess) return hip_fail(e0, "sharded quantiles (eligibility)");
        if (reduce(reduce_ctx, L.fb_count, 1, MCR_DT_I32) != 0) { set_error("reduce callback failed (eligibility)"); return MCR_ERR_HIP; }
        e0 = hipMemcpyAsync(&total, L.fb_count, sizeof(total), hipMemcpyDeviceToHost, s);
        if (e0 == hipSuccess) e0 = hipStreamSynchronize(s);
        if (e0 != hipSuccess) return hip_fail(e0, "sharded quantiles (eligibility)");
        if (total != 0u) {
            set_error("rank 0 holds fewer than %d entries of each row: the sharded bracketed route needs its first sample there "
                      "(every rank returns this together; use the stepwise radix select)", kRqTiny);
            return MCR_ERR_UNSUPPORTED;
        }
    }
    RqArgs a;
    std::memset(&a, 0, sizeof(a));
    a.n_q = n_q;
    for (int j = 0; j < n_q; ++j) a.q[j] = q[j];
    rq_opt_in_lds(device);
    auto sum = [&](void* buf, size_t count, int32_t dtype, const char* what) -> int {
        if (reduce(reduce_ctx, buf, (int64_t)count, dtype) != 0) { set_error("reduce callback failed (%s)", what); return MCR_ERR_HIP; }
        return MCR_OK;
    };
    const int blocks = (n_rows + 255) / 256;
    const unsigned int bcap = rq_bracket_cand_cap(n_local);
    const int sub_bits2 = n_total <= ((int64_t)1 << 24) ? 8 : 10;
    hipError_t e = hipSuccess;
    // (1) coarse brackets: rank 0's first sample decides for everybody
    hipLaunchKernelGGL(rq_tiny_kernel, dim3(n_rows), dim3(1024), 0, s, rows, row_stride, n_local, a, L.br1, L.hist1, L.fb_count);
    if (rank != 0) e = hipMemsetAsync(L.br1, 0, L.br_bytes, s);
    if (e != hipSuccess) return hip_fail(e, "sharded quantiles");
    if ((rc = sum(L.br1, L.br_bytes / 4, MCR_DT_I32, "coarse brackets")) != MCR_OK) return rc;
    // (2) the sample is pooled: every rank counts a prefix of its own shard; (3) fine brackets, identical everywhere
    int64_t m_total = n_total / 32;
    if (m_total < 65536) m_total = 65536;
    if (m_total > n_total) m_total = n_total;
    int64_t m_local = (m_total + world - 1) / world;
    if (m_local > n_local) m_local = n_local;
    m_local &= ~(int64_t)1;
    rq_count_pass(s, rows, row_stride, n_rows, m_local, L.br1, L.hist1, kRqCoarseSubBits, false, n_q, L.bcand, bcap);
    hipLaunchKernelGGL(rq_pack_kernel, dim3(blocks), dim3(256), 0, s, L.br1, L.cnt, (int)n_rows, bcap);
    if ((rc = sum(L.cnt, L.cnt_words, MCR_DT_I64, "sample counts")) != MCR_OK) return rc;
    if ((rc = sum(L.hist1, L.hist_words, MCR_DT_I32, "sample sub-histograms")) != MCR_OK) return rc;
    hipLaunchKernelGGL(rq_unpack_kernel, dim3(blocks), dim3(256), 0, s, L.br1, L.cnt, L.cand_total, (int)n_rows);
    hipLaunchKernelGGL(rq_refine_kernel, dim3(n_rows), dim3(256), (size_t)kRqMaxQ * kRqMaxSubBins * sizeof(unsigned int), s, (int64_t)0, a,
                       L.br1, L.hist1, L.br2, L.hist2, sub_bits2, L.gfill);
    // (4) the one pass over the local slab; this rank's own sub-histograms are kept aside (hist1 is free again)
    rq_count_pass(s, rows, row_stride, n_rows, n_local, L.br2, L.hist2, sub_bits2, true, n_q, L.bcand, bcap);
    e = hipMemcpyAsync(L.hist1, L.hist2, L.hist_words * sizeof(unsigned int), hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return hip_fail(e, "sharded quantiles");
    hipLaunchKernelGGL(rq_pack_kernel, dim3(blocks), dim3(256), 0, s, L.br2, L.cnt, (int)n_rows, bcap);
    if ((rc = sum(L.cnt, L.cnt_words, MCR_DT_I64, "slab counts")) != MCR_OK) return rc;
    if ((rc = sum(L.hist2, L.hist_words, MCR_DT_I32, "slab sub-histograms")) != MCR_OK) return rc;
    hipLaunchKernelGGL(rq_unpack_kernel, dim3(blocks), dim3(256), 0, s, L.br2, L.cnt, L.cand_total, (int)n_rows);
    // (5) cells (identical everywhere) and how many of each cell's keys every rank holds -> where each rank writes
    const size_t bins = (size_t)1 << sub_bits2;
    const unsigned int list_cap = sub_bits2 == 8 ? kRqListCap : kRqListCap / 2;
    const size_t lsz_used = (size_t)world * n_rows * kRqMaxT;
    e = hipMemsetAsync(L.lsz, 0, lsz_used * sizeof(unsigned int), s);
    if (e == hipSuccess) e = hipMemsetAsync(L.glist, 0, L.glist_bytes, s);
    if (e != hipSuccess) return hip_fail(e, "sharded quantiles");
    hipLaunchKernelGGL(rq_cells_kernel, dim3(n_rows), dim3(1024), (size_t)kRqMaxQ * bins * sizeof(unsigned int), s, n_total, a, L.br2, L.hist2,
                       L.hist1, sub_bits2, bcap, list_cap, L.cand_total, L.lsz, (int)rank, (int)n_rows);
    if ((rc = sum(L.lsz, lsz_used, MCR_DT_I32, "cell sizes")) != MCR_OK) return rc;
    const int per_row = n_local >= ((int64_t)1 << 23) ? 8 : (n_local >= ((int64_t)1 << 21) ? 4 : 2);
    hipLaunchKernelGGL(rq_collect_kernel, dim3(per_row, n_rows), dim3(1024), (size_t)kRqMaxQ * bins * 5, s, n_total, a, L.br2, L.hist2, sub_bits2,
                       L.bcand, bcap, list_cap, L.glist, L.gfill, L.cand_total, L.lsz, (int)rank, (int)n_rows);
    if ((rc = sum(L.glist, L.glist_bytes / 8, MCR_DT_I64, "cell lists")) != MCR_OK) return rc;
    if ((rc = sum(L.gfill, L.gfill_words, MCR_DT_I32, "cell fills")) != MCR_OK) return rc;
    // (6) selection in LDS -> interpolation (every rank computes the same)
    const size_t lds = (size_t)kRqMaxQ * bins * sizeof(unsigned int) + (size_t)list_cap * sizeof(unsigned long long);
    hipLaunchKernelGGL(rq_select_kernel, dim3(n_rows), dim3(1024), lds, s, n_total, a, L.br2, L.hist2, sub_bits2, bcap, list_cap, L.glist, L.gfill,
                       out, (unsigned long long*)counts, L.row_fallback, L.fb_list, L.fb_count, L.cand_total, 0);
    unsigned int n_fb = 0;
    e = hipMemcpyAsync(&n_fb, L.fb_count, sizeof(n_fb), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return hip_fail(e, "sharded bracketed row quantiles");
    g_last_fallback_rows = (int)n_fb;
    if (n_fb > 0) {   // the same rows on every rank: the distributed radix select, digit histograms summed after every pass
        rc = mcr_row_quantiles_begin(scratch, n_rows, device, hip_stream);
        for (int pass = 0; pass < 8 && rc == MCR_OK; ++pass) {
            rc = rq_hist_step(rows, row_stride, n_rows, n_local, n_q, pass, scratch, device, hip_stream, /*track_minmax=*/0, L.fb_list, (int)n_fb);
            if (rc == MCR_OK) rc = sum(L.hist, L.reduce_words, MCR_DT_I32, "digit histograms");
            if (rc == MCR_OK) rc = rq_scan_step(n_rows, n_total, q, n_q, pass, out, counts, scratch, device, hip_stream, L.fb_list, (int)n_fb);
        }
        if (rc != MCR_OK) return rc;
    }
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "sharded bracketed row quantiles");
    return MCR_OK;
}

int mcr_minmax_success(const double* values, const uint8_t* success, int64_t n, double* minmax, int device, void* hip_stream) {
    MCR_ENTER_DEVICE(device);
    if (!values || !success || !minmax || n < 0) { set_error("bad arguments"); return MCR_ERR_INVALID_ARG; }
    hipStream_t s = (hipStream_t)hip_stream;
    hipLaunchKernelGGL(minmax_init_kernel, dim3(1), dim3(256), 0, s, minmax);
    const int vec2 = ((reinterpret_cast<uintptr_t>(values) & 15) == 0 && (reinterpret_cast<uintptr_t>(success) & 1) == 0) ? 1 : 0;
    if (n > 0) hipLaunchKernelGGL(minmax_kernel, dim3(grid_for(n, 256 * 16, 2048)), dim3(256), 0, s, values, success, n, minmax, vec2);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "minmax kernels");
    return MCR_OK;
}

int mcr_histogram_success(const double* values, const uint8_t* success, int64_t n, const double* minmax,
                          int32_t n_bins, uint64_t* bins, int device, void* hip_stream) {
    MCR_ENTER_DEVICE(device);
    if (!values || !success || !minmax || !bins || n < 0 || n_bins <= 0 || n_bins > 8192) { set_error("bad arguments"); return MCR_ERR_INVALID_ARG; }
    if (n == 0) return MCR_OK;
    const int vec2 = ((reinterpret_cast<uintptr_t>(values) & 15) == 0 && (reinterpret_cast<uintptr_t>(success) & 1) == 0) ? 1 : 0;
    hipLaunchKernelGGL(hist_kernel, dim3(grid_for(n, 256 * 16, 2048)), dim3(256), (size_t)n_bins * sizeof(unsigned int),
                       (hipStream_t)hip_stream, values, success, n, minmax, (int)n_bins, (unsigned long long*)bins, vec2);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "hist_kernel");
    return MCR_OK;
}

int mcr_summary_stat_rows(const double* start_balance, const double* final_balance, const double* first_year_real_gross,
                          const uint8_t* success, int64_t n, double* rows, int64_t row_stride, int device, void* hip_stream) {
    MCR_ENTER_DEVICE(device);
    if (!start_balance || !final_balance || !first_year_real_gross || !success || !rows || n < 0 || row_stride < n) {
        set_error("bad arguments");
        return MCR_ERR_INVALID_ARG;
    }
    if (n == 0) return MCR_OK;
    hipLaunchKernelGGL(stat_rows_kernel, dim3(grid_for(n, 256 * 4, 4096)), dim3(256), 0, (hipStream_t)hip_stream,
                       start_balance, final_balance, first_year_real_gross, success, n, rows, row_stride);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "stat_rows_kernel");
    return MCR_OK;
}

}  // extern "C"
