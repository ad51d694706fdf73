// What the MI355X's memory system gives a kernel shaped like the K3 slab pass (136 rows x 10^7 doubles = 10.88 GB, read
// once), measured on fresh allocations of the slab.  Three tables per allocation, all in TB/s of bytes READ:
//   R  a kernel that does nothing but read: access shape (stride / chunk / the slab pass's rows x parts grid), plain vs
//      nontemporal 16-byte loads, 4 or 8 loads in flight per lane
//   L  the slab pass rebuilt on that sweep one ingredient at a time (its batch loop with the next batch prefetched,
//      resident workgroups per CU capped through dynamic LDS like its own 32 KB do, dependent FMAs per element, each of
//      its three LDS operations alone, and a search-free variant that counts against bounds held in SGPRs)
//   S  the sweep plus candidate-like WRITES: 192 doubles per wave every 9th batch = 4 % of the bytes it reads, by store flavour
// Findings on the round-3 boxes (profiles/r03/hbm_read_gfx950.txt): reading alone 6.5-7.0 TB/s (nt) / 6.0-6.3 (plain), at
// 3 workgroups per CU still 6.6; 20 FMAs per element free, every LDS operation of the pass free; but 4 % of writes mixed
// into the stream take it from 6.9 to 5.6 (plain) / 5.9 (nt, sc1) — the 0.3-0.5 ms the slab pass sits above its read time.
//   hipcc --offload-arch=gfx950 -O3 -o hbm_read hbm_read.hip && ./hbm_read [allocations=6] [reps=7]
// Shapes: "stride"  tile t of 256 x U x 16 B goes to workgroup t mod G (all workgroups read one moving window),
//         "chunk"   workgroup g owns the contiguous g-th G-th of the slab (workgroups far apart),
//         "rows"    the K3 slab pass's own shape: 136 x 30 workgroups, each reads the 16 KB tiles of its 30th of one row
//                   (k3: consecutive workgroups share a row, as the slab pass's (30, 136) grid does; alt: they sit in different rows).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(2); } } while (0)

typedef double v2d __attribute__((ext_vector_type(2)));

template <bool NT> __device__ __forceinline__ v2d ld(const v2d* p) {
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}

// shape 0 = stride, 1 = chunk
template <int U, bool NT, int SHAPE>
__global__ __launch_bounds__(256) void sweep(const v2d* __restrict__ p, uint64_t n_tiles, double* __restrict__ out) {
    const uint64_t tile = 256ull * U;
    uint64_t t0, t1, step;
    if (SHAPE == 0) { t0 = blockIdx.x; t1 = n_tiles; step = gridDim.x; }
    else {
        const uint64_t per = (n_tiles + gridDim.x - 1) / gridDim.x;
        t0 = per * blockIdx.x; t1 = std::min<uint64_t>(t0 + per, n_tiles); step = 1;
    }
    double acc = 0.0;
    for (uint64_t t = t0; t < t1; t += step) {
        const v2d* q = p + t * tile + threadIdx.x;
        v2d v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld<NT>(q + u * 256);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y;
    }
    if (acc == 1.2345e300) out[0] = acc;
}

// the K3 slab pass's shape: `rows` rows of `n` doubles; workgroup g serves row g % rows, part g / rows of `parts`
template <int U, bool NT, bool K3ORDER>
__global__ __launch_bounds__(256) void sweep_rows(const v2d* __restrict__ p, uint64_t n16_per_row, int rows, int parts, double* __restrict__ out) {
    // K3ORDER: consecutive workgroups are the parts of ONE row (grid (parts, rows) of the slab pass); else consecutive workgroups sit in different rows
    const int row = K3ORDER ? blockIdx.x / parts : blockIdx.x % rows, part = K3ORDER ? blockIdx.x % parts : blockIdx.x / rows;
    const uint64_t tile = 256ull * U;
    const uint64_t tiles = n16_per_row / tile;
    const uint64_t per = (tiles + parts - 1) / parts;
    const uint64_t t0 = per * part, t1 = std::min<uint64_t>(t0 + per, tiles);
    const v2d* base = p + uint64_t(row) * n16_per_row;
    double acc = 0.0;
    for (uint64_t t = t0; t < t1; ++t) {
        const v2d* q = base + t * tile + threadIdx.x;
        v2d v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld<NT>(q + u * 256);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y;
    }
    if (acc == 1.2345e300) out[0] = acc;
}


// The slab pass rebuilt feature by feature on top of the bare row sweep, to see which of its ingredients costs bandwidth:
//   PREFETCH  the batch loop of rq_slab_kernel (next batch of U 16-byte loads issued before the current one is consumed)
//   WORK      W dependent fp64 FMAs per element (the pass spends ~23 VALU instructions per element)
//   LDSOPS    per element: bucket = clamp(int(fma)), one byte of a 2 KB table, one 16-byte pair, two compares, one ds_add
//             on the lane's own counter, a ballot; members (~4 %) staged in LDS and written out 192 at a time
// dynamic LDS bytes cap the resident workgroups per CU like the pass's own 32 KB do.
template <int U, int WORK, int LDSOPS, int STAGE = 2>
__global__ __launch_bounds__(256) void slab_like(const v2d* __restrict__ p, uint64_t n16_per_row, int rows, int parts, double* __restrict__ out,
                                                 double* __restrict__ cand, unsigned int* __restrict__ cand_count, double lut_s, double lut_c) {
    extern __shared__ __align__(16) unsigned char dyn[];
    __shared__ unsigned char lut[2048 + 4];
    __shared__ __align__(16) v2d pairs[17];
    unsigned int* poshist = reinterpret_cast<unsigned int*>(dyn);          // [12][256]
    double* stage = reinterpret_cast<double*>(dyn + 12 * 256 * 4);         // [4][192]
    const int row = blockIdx.x / parts, part = blockIdx.x % parts;
    if (LDSOPS) {
        for (int k = threadIdx.x; k < 2048; k += 256) lut[k] = (unsigned char)(k * 10 / 2048);
        // bound i sits at (i + 0.5) / 10 +- 0.002: 10 intervals of width 0.004 -> 4 % members
        if (threadIdx.x < 16) { const int i = threadIdx.x; pairs[i] = v2d{(i + 0.5) / 10.0 - 0.002, (i + 0.5) / 10.0 + 0.002}; }
        for (int k = threadIdx.x; k < 12 * 256; k += 256) poshist[k] = 0u;
        __syncthreads();
    }
    const uint64_t tile = 256ull * U;
    const uint64_t tiles = n16_per_row / tile;
    const uint64_t per = (tiles + parts - 1) / parts;
    const uint64_t t0 = per * part, t1 = std::min<uint64_t>(t0 + per, tiles);
    const v2d* base = p + uint64_t(row) * n16_per_row + threadIdx.x;
    const int lane = threadIdx.x & 63;
    double* wstage = stage + (threadIdx.x >> 6) * 192;
    unsigned int filled = 0u;
    unsigned int cnt[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long nanmask = 0ull;
    double acc = 0.0;
    v2d v[U];
    auto fetch = [&](uint64_t t) {
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(base + t * tile + u * 256);
    };
    auto element = [&](double x) {
        if (WORK > 0) {
            double a = x;
#pragma unroll
            for (int w = 0; w < WORK; ++w) a = __builtin_fma(a, x, 0.25);
            acc += a;
        }
        if (LDSOPS == 2) {
            // compare-count: 10 bounds in SGPRs, per bound one v_cmp_ge_f64 into an SGPR mask, s_bcnt1 + s_add; membership = parity of the nested masks
            unsigned long long par = 0ull;
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const double bnd = (i >> 1) * 0.1 + 0.05 + ((i & 1) ? 0.002 : -0.002);
                const unsigned long long m = __ballot(x >= bnd);
                cnt[i] += (unsigned int)__popcll(m);
                par ^= m;
            }
            nanmask |= __ballot(x != x);
            if (STAGE > 0 && par) {
                const unsigned int c2 = (unsigned int)__popcll(par);
                if (filled + c2 > 192u) {
                    if (STAGE == 2) {
                        unsigned int b0 = 0u;
                        if (lane == 0) b0 = atomicAdd(cand_count + row, filled);
                        b0 = (unsigned int)__builtin_amdgcn_readfirstlane((int)b0);
                        for (unsigned int i = lane; i < filled; i += 64u) cand[(uint64_t)row * (1u << 20) + ((b0 + i) & ((1u << 20) - 1))] = wstage[i];
                    } else {
                        for (unsigned int i = lane; i < filled; i += 64u) acc += wstage[i];
                    }
                    filled = 0u;
                }
                if ((par >> lane) & 1ull)
                    wstage[filled + __builtin_amdgcn_mbcnt_hi((unsigned int)(par >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)par, 0u))] = x;
                filled += c2;
            }
            if (STAGE == 0) acc += (double)par;
        }
        if (LDSOPS >= 3) {
            int k = (int)__builtin_fma(x, lut_s, lut_c);
            k = k < 0 ? 0 : (k > 2047 ? 2047 : k);
            if (LDSOPS == 3) acc += lut[k];                                              // the byte table alone
            if (LDSOPS == 4) { const v2d bp = pairs[k & 15]; acc += bp.x + bp.y; }       // the 16-byte pair alone (few distinct addresses, as in the pass)
            if (LDSOPS == 5) atomicAdd(&poshist[((unsigned int)k % 12u) * 256 + threadIdx.x], 1u);   // the counter alone
        }
        if (LDSOPS == 1) {
            int k = (int)__builtin_fma(x, lut_s, lut_c);
            k = k < 0 ? 0 : (k > 2047 ? 2047 : k);
            const unsigned int b = lut[k];
            const v2d bp = pairs[b];
            const unsigned int pos = 2u * b + (x >= bp.x ? 1u : 0u) + (x >= bp.y ? 1u : 0u);
            atomicAdd(&poshist[(pos % 12u) * 256 + threadIdx.x], 1u);
            const unsigned long long m = __ballot(pos & 1u);
            if (m) {
                const unsigned int cnt = (unsigned int)__popcll(m);
                if (filled + cnt > 192u) {
                    unsigned int b0 = 0u;
                    if (lane == 0) b0 = atomicAdd(cand_count + row, filled);
                    b0 = (unsigned int)__builtin_amdgcn_readfirstlane((int)b0);
                    for (unsigned int i = lane; i < filled; i += 64u) cand[(uint64_t)row * (1u << 20) + ((b0 + i) & ((1u << 20) - 1))] = wstage[i];
                    filled = 0u;
                }
                if ((m >> lane) & 1ull)
                    wstage[filled + __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u))] = x;
                filled += cnt;
            }
        }
        if (WORK == 0 && LDSOPS == 0) acc += x;
    };
    if (t0 < t1) fetch(t0);
    for (uint64_t t = t0; t < t1; ++t) {
        double xs[2 * U];
#pragma unroll
        for (int u = 0; u < U; ++u) { xs[2 * u] = v[u].x; xs[2 * u + 1] = v[u].y; }
        if (t + 1 < t1) fetch(t + 1);
#pragma unroll
        for (int u = 0; u < 2 * U; ++u) element(xs[u]);
    }
    if (LDSOPS == 2) { for (int i = 0; i < 10; ++i) acc += cnt[i]; acc += (double)nanmask + filled; }
    if (LDSOPS == 1 || LDSOPS == 5) {
        __syncthreads();
        unsigned int c = 0;
        for (int k = threadIdx.x; k < 12 * 256; k += 256) c += poshist[k];
        acc += c + filled;
    }
    if (acc == 1.2345e300) out[0] = acc;
}

// The bare prefetching row sweep plus candidate-like WRITES: every `every`-th batch the wave stores 192 doubles (three
// 512-byte wave stores) — 4 % of the bytes it reads when every = 9 —, MODE 0 into a region of its own that it keeps
// rewriting (stays in the L2), 1 streaming forward through a buffer of the size the pass fills (0.44 GB), 2 like 1 with
// nontemporal stores, 3 like 1 with 16-byte stores (96 lanes' worth in two instructions).
template <int U, int MODE>
__global__ __launch_bounds__(256) void sweep_store(const v2d* __restrict__ p, uint64_t n16_per_row, int rows, int parts, double* __restrict__ out,
                                                   double* __restrict__ cand, int every) {
    extern __shared__ __align__(16) unsigned char dyn[];
    const int row = blockIdx.x / parts, part = blockIdx.x % parts;
    const uint64_t tile = 256ull * U;
    const uint64_t tiles = n16_per_row / tile;
    const uint64_t per = (tiles + parts - 1) / parts;
    const uint64_t t0 = per * part, t1 = std::min<uint64_t>(t0 + per, tiles);
    const v2d* base = p + uint64_t(row) * n16_per_row + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const uint64_t wave = uint64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    const uint64_t flushes = (per + every - 1) / every;                        // per wave
    double* mine = cand + wave * (MODE == 0 ? 192ull : 192ull * flushes);
    double acc = 0.0;
    v2d v[U];
    auto fetch = [&](uint64_t t) {
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(base + t * tile + u * 256);
    };
    if (t0 < t1) fetch(t0);
    int k = 0;
    uint64_t f = 0;
    for (uint64_t t = t0; t < t1; ++t) {
        double xs[2 * U];
#pragma unroll
        for (int u = 0; u < U; ++u) { xs[2 * u] = v[u].x; xs[2 * u + 1] = v[u].y; }
        if (t + 1 < t1) fetch(t + 1);
#pragma unroll
        for (int u = 0; u < 2 * U; ++u) acc += xs[u];
        if (++k == every) {
            k = 0;
            double* q = mine + (MODE == 0 ? 0ull : 192ull * f);
            ++f;
            if (MODE == 3) {
                if (lane < 48) { v2d w = {xs[0], xs[1]}; *reinterpret_cast<v2d*>(q + 2 * lane) = w; *reinterpret_cast<v2d*>(q + 96 + 2 * lane) = w; }
            } else {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    double* a = q + 64 * j + lane;
                    if (MODE == 2) __builtin_nontemporal_store(xs[j], a);
                    else if (MODE == 4) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(a), "v"(xs[j]) : "memory");
                    else if (MODE == 5) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(a), "v"(xs[j]) : "memory");
                    else if (MODE == 6) asm volatile("global_store_dwordx2 %0, %1, off nt sc1" :: "v"(a), "v"(xs[j]) : "memory");
                    else if (MODE == 7) asm volatile("global_store_dwordx2 %0, %1, off nt sc0 sc1" :: "v"(a), "v"(xs[j]) : "memory");
                    else if (MODE == 8) asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(a), "v"(xs[j]) : "memory");
                    else a[0] = xs[j];
                }
                if (MODE >= 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (inline-asm stores are invisible to the compiler's wait counting)
            }
        }
    }
    if (acc == 1.2345e300) out[0] = acc;
}

__global__ void fill(double* p, uint64_t n) {
    for (uint64_t i = blockIdx.x * uint64_t(blockDim.x) + threadIdx.x; i < n; i += uint64_t(gridDim.x) * blockDim.x)
        p[i] = double((i * 0x9E3779B97F4A7C15ull) >> 11) * (1.0 / 9007199254740992.0);   // [0, 1), scrambled
}

struct Timer {
    hipEvent_t a, b;
    Timer() { CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b)); }
    template <class F> double ms(F&& f, int reps) {
        std::vector<float> t;
        f();
        for (int r = 0; r < reps; ++r) {
            CHECK(hipEventRecord(a, 0)); f(); CHECK(hipEventRecord(b, 0)); CHECK(hipEventSynchronize(b));
            float x; CHECK(hipEventElapsedTime(&x, a, b)); t.push_back(x);
        }
        std::sort(t.begin(), t.end());
        return t[t.size() / 2];
    }
};

int main(int argc, char** argv) {
    const int allocations = argc > 1 ? std::atoi(argv[1]) : 6;
    const int reps = argc > 2 ? std::atoi(argv[2]) : 7;
    const int rows = 136;
    const uint64_t n = 10000000ull, total = rows * n;
    const double gb = double(total) * 8.0 / 1e9;
    const bool second_table = argc > 3 ? std::atoi(argv[3]) != 0 : true;
    double* out; CHECK(hipMalloc(&out, 64));
    double* cand; CHECK(hipMalloc(&cand, size_t(rows) * (1u << 21) * 8));
    unsigned int* cand_count; CHECK(hipMalloc(&cand_count, rows * 4));
    Timer tm;
    std::printf("# slab %d x %llu doubles = %.2f GB; median of %d launches; TB/s\n", rows, (unsigned long long)n, gb, reps);
    std::printf("# R alloc  spacer_GB  stride4nt  stride4  stride8nt  chunk4nt  chunk8nt  rows_k3_nt  rows_k3  rows_alt_nt\n");
    for (int a = 0; a < allocations; ++a) {
        void* spacer = nullptr;
        const size_t spacer_bytes = size_t(a) * 1370000000ull + (a ? 4096 * 37 : 0);
        if (spacer_bytes) CHECK(hipMalloc(&spacer, spacer_bytes));
        double* slab; CHECK(hipMalloc(&slab, total * 8));
        fill<<<4096, 256>>>(slab, total);
        CHECK(hipDeviceSynchronize());
        const v2d* p = reinterpret_cast<const v2d*>(slab);
        const uint64_t n16 = total / 2;
        auto rate = [&](double ms, double bytes) { return bytes / (ms * 1e-3) / 1e12; };
        auto run = [&](auto kern, int U, int grid) {
            const uint64_t tiles = n16 / (256ull * U);
            const double bytes = double(tiles) * 256.0 * U * 16.0;
            return rate(tm.ms([&] { kern<<<grid, 256>>>(p, tiles, out); }, reps), bytes);
        };
        const double s4nt = run(sweep<4, true, 0>, 4, 2048);
        const double s4 = run(sweep<4, false, 0>, 4, 2048);
        const double s8nt = run(sweep<8, true, 0>, 8, 2048);
        const double c4nt = run(sweep<4, true, 1>, 4, 2048);
        const double c8nt = run(sweep<8, true, 1>, 8, 2048);
        auto run_rows = [&](auto kern, int U, int parts) {
            const uint64_t n16r = n / 2;
            const uint64_t tiles = n16r / (256ull * U);
            const double bytes = double(tiles) * 256.0 * U * 16.0 * rows;
            return rate(tm.ms([&] { kern<<<rows * parts, 256>>>(p, n16r, rows, parts, out); }, reps), bytes);
        };
        const double r4nt = run_rows(sweep_rows<4, true, true>, 4, 30);
        const double r4 = run_rows(sweep_rows<4, false, true>, 4, 30);
        const double r4nt8 = run_rows(sweep_rows<4, true, false>, 4, 30);
        std::printf("R %d      %6.2f     %.3f     %.3f    %.3f     %.3f     %.3f     %.3f        %.3f      %.3f\n", a, spacer_bytes / 1e9,
                    s4nt, s4, s8nt, c4nt, c8nt, r4nt, r4, r4nt8);
        std::fflush(stdout);
        if (second_table) {
            auto run_like = [&](auto kern, int U, size_t lds) {
                const uint64_t n16r = n / 2;
                const uint64_t tiles = n16r / (256ull * U);
                const double bytes = double(tiles) * 256.0 * U * 16.0 * rows;
                CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096));
                return rate(tm.ms([&] { CHECK(hipMemsetAsync(cand_count, 0, rows * 4, 0)); kern<<<rows * 30, 256, lds>>>(p, n16r, rows, 30, out, cand, cand_count, 2048.0, 0.0); }, reps), bytes);
            };
            const size_t kb = 1024;
            std::printf("L %d  prefetching batch loop: 8 wg/CU %.3f  5 wg/CU %.3f  3 wg/CU %.3f | 5 wg/CU + 20 FMAs/element %.3f  + 40 %.3f | + byte table %.3f  + 16-byte pair %.3f  + ds_add %.3f | all three + per-element staging %.3f | bounds in SGPRs, counting only %.3f\n", a,
                        run_like(slab_like<4, 0, 0>, 4, 18 * kb), run_like(slab_like<4, 0, 0>, 4, 30 * kb), run_like(slab_like<4, 0, 0>, 4, 50 * kb),
                        run_like(slab_like<4, 20, 0>, 4, 30 * kb), run_like(slab_like<4, 40, 0>, 4, 30 * kb),
                        run_like(slab_like<4, 0, 3>, 4, 28 * kb), run_like(slab_like<4, 0, 4>, 4, 28 * kb), run_like(slab_like<4, 0, 5>, 4, 28 * kb),
                        run_like(slab_like<4, 0, 1>, 4, 28 * kb), run_like(slab_like<4, 0, 2, 0>, 4, 28 * kb));
            auto run_store = [&](auto kern, int U, size_t lds, int every) {
                const uint64_t n16r = n / 2;
                const uint64_t tiles = n16r / (256ull * U);
                const double bytes = double(tiles) * 256.0 * U * 16.0 * rows;
                CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096));
                return rate(tm.ms([&] { kern<<<rows * 30, 256, lds>>>(p, n16r, rows, 30, out, cand, every); }, reps), bytes);
            };
            std::printf("S %d  5 wg/CU, reads + 192 doubles stored every 9th batch (4 %% of the bytes): none %.3f | own region %.3f | streaming %.3f | nt %.3f | sc1 %.3f | sc0 sc1 %.3f | nt sc1 %.3f | nt sc0 sc1 %.3f | sc0 %.3f | plain+wait %.3f\n", a,
                        run_like(slab_like<4, 0, 0>, 4, 30 * kb),
                        run_store(sweep_store<4, 0>, 4, 30 * kb, 9), run_store(sweep_store<4, 1>, 4, 30 * kb, 9), run_store(sweep_store<4, 2>, 4, 30 * kb, 9),
                        run_store(sweep_store<4, 4>, 4, 30 * kb, 9), run_store(sweep_store<4, 5>, 4, 30 * kb, 9), run_store(sweep_store<4, 6>, 4, 30 * kb, 9),
                        run_store(sweep_store<4, 7>, 4, 30 * kb, 9), run_store(sweep_store<4, 8>, 4, 30 * kb, 9), run_store(sweep_store<4, 9>, 4, 30 * kb, 9));
            std::fflush(stdout);
        }
        CHECK(hipFree(slab));
        if (spacer) CHECK(hipFree(spacer));
    }
    return 0;
}
