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
//     Large rows avoid most of those passes (mcr_row_quantiles, single GPU): the same select runs first
//     on a SAMPLE (the first n/32 entries) to get, per quantile, a bracket of keys that contains the
//     wanted order statistics with overwhelming probability; ONE pass over the slab then counts the keys
//     below / inside every bracket and compacts the few percent inside; the select finishes on those
//     candidates with ranks shifted by the counts.  Exactness never depends on the sample: a row whose
//     counts show a target outside its bracket (or whose candidates overflow: giant ties) simply takes
//     the full radix passes.  Traffic: ~1.5 reads of the slab instead of 4.
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
    unsigned int clamp_lo, clamp_hi;     // sample mode: bit j = bracket end of quantile j runs off the sample's range
};

// What the targets of a select are (rq_scan_kernel):
constexpr int kRqQuantiles = 0;  // order statistics around (n-1)*q, interpolated and written to `out` (the plain select)
constexpr int kRqSample = 1;     // bracket ranks around (m-1)*q in a sample of m entries; selected values stay in RqRow
constexpr int kRqExplicit = 2;   // ranks preset by rq_resolve_kernel (candidate select); selected values stay in RqRow
constexpr unsigned long long kRqKeyPosInf = 0xFFF0000000000000ull;   // key_of(+inf): the largest non-NaN key
constexpr double kRqBracketSigmas = 5.0;   // half-width of a bracket in binomial standard deviations of the sample rank

// Disjoint, ascending key intervals of one row (from its sample) and what the bracket pass counted for them.
struct RqBracket {
    unsigned long long lo[kRqMaxQ], hi[kRqMaxQ];   // closed key intervals [lo, hi]
    // bracket pass: # non-NaN keys by position among the sorted bounds lo0 <= hi0 < lo1 <= hi1 < ...:
    // position 2b+1 = inside interval b, position 2b = in the gap below it (2*n_intervals = above the last)
    unsigned long long pos_count[2 * kRqMaxQ + 1];
    unsigned long long n_nan;
    int interval_of_q[kRqMaxQ];
    int n_intervals;
    unsigned int cand_count;                       // values appended to the row's candidate buffer
    unsigned int fallback;                         // 1: the row takes the full radix passes
};
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
                                                          const unsigned int* __restrict__ row_n,
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
    if (row_n) {                                       // ragged rows (candidate buffers): this row's own length
        const int64_t mine = (int64_t)row_n[row];
        n = mine < n ? mine : n;
        if (n == 0) return;
    }
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
                                                    unsigned int* aux, int mode,
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
    if (pass == 0 && t == 0 && mode != kRqExplicit) {
        const unsigned long long m = (unsigned long long)n - (unsigned long long)aux[2 * row];
        S.n_valid = m;
        if (counts) counts[row] = m;
        int nt = 0;
        unsigned int clamp_lo = 0u, clamp_hi = 0u;
        if (m > 0) {
            for (int j = 0; j < args.n_q; ++j) {
                // _QuantileMethods['linear'].get_virtual_index = (n - 1) * quantiles
                const double q = args.q[j];
                const double vi = (double)(m - 1) * q;
                if (mode == kRqQuantiles) {
                    double prev = floor(vi), next = prev + 1.0;          // _get_indexes
                    if (vi >= (double)(m - 1)) { prev = (double)(m - 1); next = prev; }
                    if (vi < 0.0) { prev = 0.0; next = 0.0; }
                    S.gamma[j] = vi - floor(vi);                          // _get_gamma (linear: unchanged)
                    S.rank[nt] = (unsigned long long)prev; S.group_of[nt] = 0; ++nt;
                    S.rank[nt] = (unsigned long long)next; S.group_of[nt] = 0; ++nt;
                } else {
                    // sample of m entries: the order statistics of the whole row around quantile q lie, with
                    // overwhelming probability, between the sample's order statistics this far from (m-1)*q
                    const double d = ceil(kRqBracketSigmas * sqrt((double)m * q * (1.0 - q))) + 2.0;
                    double lo = floor(vi) - d, hi = floor(vi) + 1.0 + d;
                    if (lo < 0.0) { lo = 0.0; clamp_lo |= 1u << j; }                      // bracket open below
                    if (hi > (double)(m - 1)) { hi = (double)(m - 1); clamp_hi |= 1u << j; }  // bracket open above
                    S.rank[nt] = (unsigned long long)lo; S.group_of[nt] = 0; ++nt;
                    S.rank[nt] = (unsigned long long)hi; S.group_of[nt] = 0; ++nt;
                }
            }
        }
        S.n_targets = nt;
        S.n_groups = 1;
        S.prefix[0] = 0ull;
        S.clamp_lo = clamp_lo;
        S.clamp_hi = clamp_hi;
        // all non-NaN entries equal (e.g. the t = 0 rows: every path starts from the same balance): every order
        // statistic is that value — finish now; later passes skip the row (only when min/max were tracked)
        S.const_row = (mode == kRqQuantiles && m > 0 && S.kmin == S.kmax) ? 1u : 0u;
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
        if (mode == kRqQuantiles && t < args.n_q) {
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
// After the sample select (kRqSample): turn the 2 bracket keys per quantile into disjoint ascending intervals.
__global__ void rq_bracket_prep_kernel(const RqRow* st, RqBracket* br, int n_q, int n_rows, unsigned int* fb_count) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row == 0) *fb_count = 0u;
    if (row >= n_rows) return;
    const RqRow& S = st[row];
    RqBracket& B = br[row];
    B.n_nan = 0ull; B.cand_count = 0u; B.fallback = 0u; B.n_intervals = 0;
    for (int b = 0; b < kRqMaxQ; ++b) { B.lo[b] = ~0ull; B.hi[b] = 0ull; B.interval_of_q[b] = 0; }
    for (int k = 0; k <= 2 * kRqMaxQ; ++k) B.pos_count[k] = 0ull;
    if (S.n_targets != 2 * n_q) { B.fallback = 1u; return; }   // the sample had no non-NaN entry
    unsigned long long lo[kRqMaxQ], hi[kRqMaxQ];
    int order[kRqMaxQ];
    for (int j = 0; j < n_q; ++j) {
        // open ends: the whole range of non-NaN keys.  The ends are exact order statistics of the sample, so a
        // bracket that sits inside one giant tie comes out as a ONE-KEY interval (lo == hi), which needs no candidates.
        lo[j] = ((S.clamp_lo >> j) & 1u) ? 0ull : key_of(S.value[2 * j]);
        hi[j] = ((S.clamp_hi >> j) & 1u) ? kRqKeyPosInf : key_of(S.value[2 * j + 1]);
        int k = j;
        while (k > 0 && lo[order[k - 1]] > lo[j]) { order[k] = order[k - 1]; --k; }   // insertion sort by lo
        order[k] = j;
    }
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
}

// THE pass over the slab.  Every non-NaN key is located among the row's sorted interval bounds by a branch-free
// binary search (P-entry table in LDS: bound 2b = lo[b], bound 2b+1 = hi[b]+1, padded with ~0; position = number of
// bounds <= key), counted in a per-thread LDS histogram laid out [position][thread] (conflict-free, plain
// read-modify-write: no atomics), and — when its position is odd, i.e. it lies inside an interval that is a real
// range — appended to the row's candidate buffer (staged per wave in LDS: one global reservation and a coalesced
// copy per ~256 candidates, no workgroup barrier in the loop).
// 8 B/element read, a few % written.
template <int P>
__global__ __launch_bounds__(kRqBlock) void rq_bracket_kernel(const double* __restrict__ rows, int64_t row_stride, int64_t n,
                                                             RqBracket* br, double* cand, unsigned int cand_cap) {
    constexpr int kWaveStage = 256;                      // candidates a wave collects in LDS before it appends them
    __shared__ double stage[(kRqBlock / 64) * kWaveStage];
    __shared__ unsigned long long bound[P];
    __shared__ unsigned int poshist[P * kRqBlock];
    __shared__ unsigned int nan_n;
    const int row = blockIdx.y;
    RqBracket& B = br[row];
    const int nb = B.n_intervals;
    if (B.fallback || 2 * nb >= P) return;             // (the host picks P > 2 * n_q >= 2 * nb)
    if (threadIdx.x < P) {
        const int b = threadIdx.x >> 1;
        bound[threadIdx.x] = b < nb ? ((threadIdx.x & 1) ? B.hi[b] + 1ull : B.lo[b]) : ~0ull;   // hi <= key(+inf): no wrap
    }
    for (int k = threadIdx.x; k < P * kRqBlock; k += kRqBlock) poshist[k] = 0u;
    if (threadIdx.x == 0) nan_n = 0u;
    unsigned int keep = 0u;                            // bit b: interval b is a real range (lo < hi): members are candidates
    for (int b = 0; b < nb; ++b) keep |= (B.lo[b] < B.hi[b]) ? (1u << b) : 0u;
    keep = (unsigned int)__builtin_amdgcn_readfirstlane((int)keep);
    __syncthreads();
    const double* r = rows + (int64_t)row * row_stride;
    double* crow = cand + (size_t)row * cand_cap;
    const int lane = threadIdx.x & 63;
    unsigned int my_nan = 0u;
    double* wstage = stage + (threadIdx.x >> 6) * kWaveStage;   // this wave's stage; `filled` is wave-uniform
    unsigned int filled = 0u;
    // append the wave's staged candidates to the row's buffer: one global reservation, coalesced copy
    auto flush = [&]() {
        if (filled == 0u) return;
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
            mine += ins[u] ? 1u : 0u;
        }
        if (__ballot(mine != 0u)) {           // wave-uniform
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
        constexpr int kUnroll = 4;     // four 16-byte loads in flight per lane
        for (int64_t t = 0; t < trips; t += kUnroll) {
            d2_t v[kUnroll];
            double xs[2 * kUnroll];
            bool oks[2 * kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const int64_t i = (t + u) * step + (int64_t)blockIdx.x * kRqBlock + threadIdx.x;
                oks[2 * u] = oks[2 * u + 1] = i < n_pairs;
                v[u] = oks[2 * u] ? __builtin_nontemporal_load(&r2[i]) : d2_t{0.0, 0.0};
            }
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) { xs[2 * u] = v[u].x; xs[2 * u + 1] = v[u].y; }
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
    flush();
}

// Per row, after the bracket pass: the true ranks (NumPy `linear`, as rq_scan_kernel computes them), checked
// against the counts and shifted into ranks inside the candidate buffer.  A row any of whose targets lies outside
// its interval, or whose candidates overflowed, is handed to the full radix passes.
__global__ void rq_resolve_kernel(int64_t n, const RqArgs args, RqBracket* br, RqRow* st, unsigned int* row_n,
                                  unsigned int* row_fallback, unsigned int* fb_list, unsigned int* fb_count,
                                  unsigned long long* counts, unsigned int cand_cap, int n_rows) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    RqBracket& B = br[row];
    RqRow& S = st[row];   // freshly initialised by rq_init_kernel
    bool fb = B.fallback != 0u;
    if (!fb) {
        const unsigned long long m = (unsigned long long)n - B.n_nan;
        S.n_valid = m;
        if (counts) counts[row] = m;
        unsigned long long below[kRqMaxQ], upto[kRqMaxQ], run = 0ull;   // # keys < lo[b], # keys <= hi[b]
        for (int b = 0; b < B.n_intervals; ++b) {
            run += B.pos_count[2 * b]; below[b] = run;
            run += B.pos_count[2 * b + 1]; upto[b] = run;
        }
        unsigned long long inside_before[kRqMaxQ], total_inside = 0ull;
        unsigned int known = 0u;   // bit j: quantile j sits in a one-key interval
        for (int b = 0; b < B.n_intervals; ++b) {
            inside_before[b] = total_inside;
            if (B.lo[b] != B.hi[b]) total_inside += upto[b] - below[b];   // one-key intervals stored nothing
        }
        if (B.cand_count > cand_cap || total_inside != (unsigned long long)B.cand_count) fb = true;
        int nt = 0;
        if (!fb && m > 0) {
            for (int j = 0; j < args.n_q; ++j) {
                const double q = args.q[j];
                const double vi = (double)(m - 1) * q;
                double prev = floor(vi), next = prev + 1.0;
                if (vi >= (double)(m - 1)) { prev = (double)(m - 1); next = prev; }
                if (vi < 0.0) { prev = 0.0; next = 0.0; }
                S.gamma[j] = vi - floor(vi);
                const int b = B.interval_of_q[j];
                const bool tie = B.lo[b] == B.hi[b];   // one key: every member IS that value, none was stored
                const unsigned long long r2[2] = {(unsigned long long)prev, (unsigned long long)next};
                for (int e = 0; e < 2; ++e) {
                    if (r2[e] < below[b] || r2[e] >= upto[b]) fb = true;           // outside the bracket: not provable here
                    S.rank[nt] = tie ? 0ull : r2[e] - below[b] + inside_before[b];   // rank among the candidates
                    S.group_of[nt] = 0;
                    ++nt;
                }
                if (tie) known |= 1u << j;
            }
        }
        S.n_targets = fb ? 0 : nt;
        S.n_groups = 1;
        S.prefix[0] = 0ull;
        S.clamp_lo = known;        // (field reused: read by rq_finalize_kernel)
    }
    if (fb) { B.fallback = 1u; S.n_targets = 0; }
    row_n[row] = fb ? 0u : B.cand_count;
    row_fallback[row] = fb ? 1u : 0u;
    if (fb) fb_list[atomicAdd(fb_count, 1u)] = (unsigned int)row;
}

// Interpolate the rows that were decided on their candidates (the fallback rows are written by the radix passes).
__global__ void rq_finalize_kernel(const RqRow* st, const RqBracket* br, const unsigned int* row_fallback, int n_q, int n_rows,
                                   double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * n_q) return;
    const int row = i / n_q, t = i % n_q;
    if (row_fallback[row]) return;
    const RqRow& S = st[row];
    double r;
    if (S.n_valid == 0) {
        r = __longlong_as_double(0x7ff8000000000000LL);
    } else {
        double a = S.value[2 * t], b = S.value[2 * t + 1];
        const double g = S.gamma[t];
        if ((S.clamp_lo >> t) & 1u) a = b = value_of(br[row].lo[br[row].interval_of_q[t]]);   // one-key interval
        const double diff = b - a;                       // _lerp
        r = a + diff * g;
        if (g >= 0.5) r = b - diff * (1.0 - g);
    }
    out[(size_t)row * n_q + t] = r;
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

static unsigned int rq_cand_cap(int64_t n) { return (unsigned int)(n / 64 + 4096); }

static unsigned int rq_bracket_cand_cap(int64_t n) { return (unsigned int)((n / 8 + 4096) & ~(int64_t)1); }  // even: 16-byte rows

struct RqLayout {
    RqRow* st;
    unsigned int* hist;
    unsigned int* aux;
    unsigned long long* cand;
    size_t reduce_offset, reduce_words;
    // bracketed single pass (mcr_row_quantiles only)
    RqBracket* br;
    unsigned int* row_n;
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
    const size_t off_rn = rq_align16(off_br + (size_t)n_rows * sizeof(RqBracket));
    const size_t off_bc = rq_align16(off_rn + ((size_t)n_rows * 3 + 1) * sizeof(unsigned int));
    L.total_bytes = off_bc + (size_t)n_rows * (size_t)rq_bracket_cand_cap(n) * sizeof(double);
    char* base = (char*)scratch;
    L.st = (RqRow*)base;
    L.hist = (unsigned int*)(base + L.reduce_offset);
    L.aux = (unsigned int*)(base + off_aux);
    L.cand = (unsigned long long*)(base + off_cand);
    L.br = (RqBracket*)(base + off_br);
    L.row_n = (unsigned int*)(base + off_rn);
    L.row_fallback = L.row_n + n_rows;
    L.fb_list = L.row_fallback + n_rows;
    L.fb_count = L.fb_list + n_rows;
    L.bcand = (double*)(base + off_bc);
    return L;
}

static int rq_check(const void* scratch, int32_t n_rows, int64_t n_local, int32_t n_q) {
    if (!scratch || n_rows <= 0 || n_rows > 65535 || n_q <= 0 || n_q > kRqMaxQ) { set_error("bad row-quantile arguments"); return MCR_ERR_INVALID_ARG; }
    if (n_local < 0 || n_local >= ((int64_t)1 << 32)) { set_error("n must be < 2^32 per row"); return MCR_ERR_INVALID_ARG; }
    return MCR_OK;
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
                        const unsigned int* row_n = nullptr, const unsigned int* row_list = nullptr, int n_list = 0,
                        int slow_cap = 512);

int mcr_row_quantiles_hist(const double* rows, int64_t row_stride, int32_t n_rows, int64_t n_local, int32_t n_q,
                           int32_t pass, void* scratch, int device, void* hip_stream) {
    // rows sharded across ranks: no all-equal-row shortcut (a row's min/max is not summable across ranks)
    return rq_hist_step(rows, row_stride, n_rows, n_local, n_q, pass, scratch, device, hip_stream, 0);
}

static int rq_hist_step(const double* rows, int64_t row_stride, int32_t n_rows, int64_t n_local, int32_t n_q,
                        int32_t pass, void* scratch, int device, void* hip_stream, int track_minmax,
                        const unsigned int* row_n, const unsigned int* row_list, int n_list, int slow_cap) {
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
        const dim3 grid_slow(grid_for(n, kRqBlock * 16, slow_cap), gy);
        const size_t lds_groups = (size_t)(2 * n_q) * 256 * sizeof(unsigned int);  // <= 2 targets per quantile
        const size_t lds_first = 256 * sizeof(unsigned int);
        if (pass == 0) {
            hipLaunchKernelGGL((rq_hist_kernel<true, false>), grid, block, lds_first, s, rows, row_stride, n, pass, L.st, L.hist, L.aux, L.cand, cap, 0, track_minmax, row_n, row_list);
        } else if (pass < 3) {
            hipLaunchKernelGGL((rq_hist_kernel<false, false>), grid, block, lds_groups, s, rows, row_stride, n, pass, L.st, L.hist, L.aux, L.cand, cap, 0, 0, row_n, row_list);
        } else if (pass == 3) {
            hipLaunchKernelGGL((rq_hist_kernel<false, true>), grid, block, lds_groups, s, rows, row_stride, n, pass, L.st, L.hist, L.aux, L.cand, cap, 0, 0, row_n, row_list);
        } else {
            hipLaunchKernelGGL(rq_cand_hist_kernel, dim3(16, gy), block, lds_groups, s, pass, L.st, L.hist, L.aux, L.cand, cap, row_list);
            hipLaunchKernelGGL((rq_hist_kernel<false, false>), grid_slow, block, lds_groups, s, rows, row_stride, n, pass, L.st, L.hist, L.aux, L.cand, cap, 1, 0, row_n, row_list);
        }
    }
    if (pass == 3) hipLaunchKernelGGL(rq_flag_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, s, L.st, L.aux, n_rows, cap);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "row quantile histogram step");
    return MCR_OK;
}

static int rq_scan_step(int32_t n_rows, int64_t n_total, const double* q, int32_t n_q, int32_t pass, double* out,
                        uint64_t* counts, void* scratch, int device, void* hip_stream, int mode,
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
                       L.st, L.hist, a, out, (unsigned long long*)counts, L.aux, mode, row_list);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "rq_scan_kernel");
    return MCR_OK;
}

int mcr_row_quantiles_scan(int32_t n_rows, int64_t n_total, const double* q, int32_t n_q, int32_t pass, double* out,
                           uint64_t* counts, void* scratch, int device, void* hip_stream) {
    return rq_scan_step(n_rows, n_total, q, n_q, pass, out, counts, scratch, device, hip_stream, kRqQuantiles, nullptr, 0);
}

// The full radix select (8 digit passes) over rows of length n: all rows, or the n_list rows listed in row_list.
static int rq_radix_select(const double* rows, int64_t row_stride, int32_t n_rows, int64_t n, const double* q, int32_t n_q,
                           double* out, uint64_t* counts, void* scratch, int device, void* hip_stream, int mode,
                           const unsigned int* row_n, const unsigned int* row_list, int n_list) {
    int rc = MCR_OK;
    for (int pass = 0; pass < 8 && rc == MCR_OK; ++pass) {
        rc = rq_hist_step(rows, row_stride, n_rows, n, n_q, pass, scratch, device, hip_stream,
                          /*track_minmax=*/mode == kRqQuantiles ? 1 : 0, row_n, row_list, n_list,
                          /*slow_cap=*/mode == kRqQuantiles ? 512 : 32);   // the sample / candidate selects are short rows
        if (rc == MCR_OK) rc = rq_scan_step(n_rows, n, q, n_q, pass, out, counts, scratch, device, hip_stream, mode, row_list, n_list);
    }
    return rc;
}

// Rows at least this long take the bracketed single pass (below it the extra launches cost more than the
// three streaming passes they save).  MCR_RQ_BRACKET_MIN_N overrides it (tests force either route).
static int64_t rq_bracket_min_n() {
    const char* e = std::getenv("MCR_RQ_BRACKET_MIN_N");
    if (e && *e) return (int64_t)std::strtoll(e, nullptr, 10);
    return (int64_t)1 << 21;
}

static thread_local int g_last_fallback_rows = -1;
int mcr_row_quantiles_last_fallback_rows(void) { return g_last_fallback_rows; }

int mcr_row_quantiles(const double* rows, int64_t row_stride, int32_t n_rows, int64_t n, const double* q,
                      int32_t n_q, double* out, uint64_t* counts, void* scratch, int device, void* hip_stream) {
    g_last_fallback_rows = -1;
    if (!rows || n <= 0) { set_error("bad arguments"); return MCR_ERR_INVALID_ARG; }
    int rc = mcr_row_quantiles_begin(scratch, n_rows, device, hip_stream);
    if (rc != MCR_OK) return rc;
    if (n < rq_bracket_min_n() || n < 4096 || 2 * n_q >= 32)
        return rq_radix_select(rows, row_stride, n_rows, n, q, n_q, out, counts, scratch, device, hip_stream, kRqQuantiles, nullptr, nullptr, 0);

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
    // (1) brackets from a sample: the first n/32 entries of every row (paths are exchangeable; an unlucky or
    //     adversarial prefix only costs the affected rows the fallback below)
    int64_t m = n / 32;
    if (m < 65536) m = 65536;
    if (m > n) m = n;
    m &= ~(int64_t)1;
    rc = rq_radix_select(rows, row_stride, n_rows, m, q, n_q, out, nullptr, scratch, device, hip_stream, kRqSample, nullptr, nullptr, 0);
    if (rc != MCR_OK) return rc;
    hipLaunchKernelGGL(rq_bracket_prep_kernel, dim3((n_rows + 63) / 64), dim3(64), 0, s, L.st, L.br, (int)n_q, (int)n_rows, L.fb_count);
    // (2) the one pass over the slab
    const unsigned int bcap = rq_bracket_cand_cap(n);
    {
        const int bx = grid_for(n, kRqBlock * 16, 4096 / n_rows > 0 ? 4096 / n_rows : 1);
        const dim3 grid(bx, n_rows), block(kRqBlock);
#define MCR_BRACKET(P) hipLaunchKernelGGL((rq_bracket_kernel<P>), grid, block, 0, s, rows, row_stride, n, L.br, L.bcand, bcap)
        if (2 * n_q < 16) MCR_BRACKET(16);        // bound table: the next power of two above 2 * (intervals <= quantiles)
        else MCR_BRACKET(32);
#undef MCR_BRACKET
    }
    // (3) ranks among the candidates, then the select on the candidate buffers (ragged rows)
    rc = mcr_row_quantiles_begin(scratch, n_rows, device, hip_stream);
    if (rc != MCR_OK) return rc;
    hipLaunchKernelGGL(rq_resolve_kernel, dim3((n_rows + 63) / 64), dim3(64), 0, s, n, a, L.br, L.st, L.row_n, L.row_fallback,
                       L.fb_list, L.fb_count, (unsigned long long*)counts, bcap, (int)n_rows);
    rc = rq_radix_select(L.bcand, (int64_t)bcap, n_rows, (int64_t)bcap, q, n_q, out, nullptr, scratch, device, hip_stream,
                         kRqExplicit, L.row_n, nullptr, 0);
    if (rc != MCR_OK) return rc;
    hipLaunchKernelGGL(rq_finalize_kernel, dim3((n_rows * n_q + 255) / 256), dim3(256), 0, s, L.st, L.br, L.row_fallback, (int)n_q,
                       (int)n_rows, out);
    // (4) rows the brackets could not decide (a target outside its bracket, candidates overflowing on a wide tie
    //     that straddles a bracket end, an all-NaN sample) take the full passes.  How many is only known on the
    //     device: this route reads one word back (ONE stream synchronisation per call) rather than launch eight
    //     passes of idle workgroups.
    unsigned int n_fb = 0;
    hipError_t e = hipMemcpyAsync(&n_fb, L.fb_count, sizeof(n_fb), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return hip_fail(e, "bracketed row quantiles");
    g_last_fallback_rows = (int)n_fb;
    if (n_fb > 0) {
        rc = mcr_row_quantiles_begin(scratch, n_rows, device, hip_stream);
        if (rc != MCR_OK) return rc;
        rc = rq_radix_select(rows, row_stride, n_rows, n, q, n_q, out, counts, scratch, device, hip_stream, kRqQuantiles, nullptr,
                             L.fb_list, (int)n_fb);
        if (rc != MCR_OK) return rc;
    }
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "bracketed row quantiles");
    return MCR_OK;
}

int mcr_minmax_success(const double* values, const uint8_t* success, int64_t n, double* minmax, int device, void* hip_stream) {
    MCR_ENTER_DEVICE(device);
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
    MCR_ENTER_DEVICE(device);
    if (!values || !success || !minmax || !bins || n < 0 || n_bins <= 0 || n_bins > 8192) { set_error("bad arguments"); return MCR_ERR_INVALID_ARG; }
    if (n == 0) return MCR_OK;
    hipLaunchKernelGGL(hist_kernel, dim3(grid_for(n, 256 * 8, 2048)), dim3(256), (size_t)n_bins * sizeof(unsigned int),
                       (hipStream_t)hip_stream, values, success, n, minmax, (int)n_bins, (unsigned long long*)bins);
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
