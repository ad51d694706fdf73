// mcr_numpy_rng.h — NumPy's own random stream on the device (literal seed parity).
//
// Restates, from the published algorithms, what the reference executes through NumPy
// (backend/simulation.py:148-149 SeedSequence(main).spawn(2); :195-197 spawn(n) + generate_state(1);
// :457-458 default_rng(path_seed).standard_normal((n, 3))):
//   SeedSequence  (M.E. O'Neill's seed_seq_fe128 variant; numpy/random/bit_generator.pyx)
//   PCG64         (128-bit LCG, XSL-RR 128/64 output; numpy/random/src/pcg64)
//   standard_normal = 256-layer ziggurat (Marsaglia & Tsang; numpy/random/src/distributions)
// numpy >= 2.4.2 is pinned by the reference (uv.lock:420-421); the stream is stable across versions.
// The executable model tools/numpy_rng_model.py reproduces NumPy bit-for-bit and is the blueprint.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mcr_math.h"
#include "mcr_numpy_tables.h"

namespace mcr {

constexpr int kZigLdsBytes = kZigN * (8 + 8 + 8);  // ki, wi, fi staged in LDS

struct ZigTables {  // LDS views
    const uint64_t* ki;
    const double* wi;
    const double* fi;
    const double* math_tab;   // the workgroup's math tables (mcr_math.h): the wedge test's exp
};

__device__ __forceinline__ ZigTables load_zig_tables(unsigned char* lds, int tid, int nthreads) {
    uint64_t* ki = reinterpret_cast<uint64_t*>(lds);
    double* wi = reinterpret_cast<double*>(lds + kZigN * 8);
    double* fi = reinterpret_cast<double*>(lds + kZigN * 16);
    for (int i = tid; i < kZigN; i += nthreads) { ki[i] = kZigKi[i]; wi[i] = kZigWi[i]; fi[i] = kZigFi[i]; }
    return ZigTables{ki, wi, fi, nullptr};
}

// ---- SeedSequence -----------------------------------------------------------------------------
struct SeedSeqHasher {
    uint32_t hash_const;
    __device__ __forceinline__ uint32_t hashmix(uint32_t v) {
        v ^= hash_const;
        hash_const *= 0x931e8875u;  // MULT_A
        v *= hash_const;
        return v ^ (v >> 16);
    }
};
__device__ __forceinline__ uint32_t seedseq_mix(uint32_t x, uint32_t y) {
    uint32_t r = 0xca01f9ddu * x - 0x4973f715u * y;  // MIX_MULT_L, MIX_MULT_R
    return r ^ (r >> 16);
}

// pool of SeedSequence(entropy words, spawn_key); n_ent >= 1.  With a spawn key the entropy is
// zero-padded to the pool size before the key words are appended (get_assembled_entropy).
__device__ __forceinline__ void seedseq_pool(const uint32_t* ent, int n_ent, const uint32_t* key, int n_key,
                                             uint32_t pool[4]) {
    const int n_run = (n_key > 0 && n_ent < 4) ? 4 : n_ent;
    const int n_all = n_run + n_key;
    auto word = [&](int i) -> uint32_t {
        if (i < n_run) return i < n_ent ? ent[i] : 0u;
        return key[i - n_run];
    };
    SeedSeqHasher h{0x43b0d7e5u};  // INIT_A
    for (int i = 0; i < 4; ++i) pool[i] = h.hashmix(i < n_all ? word(i) : 0u);
    for (int s = 0; s < 4; ++s)
        for (int d = 0; d < 4; ++d)
            if (s != d) pool[d] = seedseq_mix(pool[d], h.hashmix(pool[s]));
    for (int s = 4; s < n_all; ++s)
        for (int d = 0; d < 4; ++d) pool[d] = seedseq_mix(pool[d], h.hashmix(word(s)));
}

// generate_state(n_words, uint32): word i of the output stream
__device__ __forceinline__ void seedseq_generate(const uint32_t pool[4], int n_words, uint32_t* out) {
    uint32_t hc = 0x8b51f9ddu;  // INIT_B
    for (int i = 0; i < n_words; ++i) {
        uint32_t v = pool[i & 3] ^ hc;
        hc *= 0x58f38dedu;  // MULT_B
        v *= hc;
        out[i] = v ^ (v >> 16);
    }
}

// ---- PCG64 ------------------------------------------------------------------------------------
struct Pcg64 {
    unsigned __int128 state, inc;
};
__device__ __forceinline__ void pcg64_step(Pcg64& g) {
    const unsigned __int128 mult = ((unsigned __int128)2549297995355413924ull << 64) | 4865540595714422341ull;
    g.state = g.state * mult + g.inc;
}
__device__ __forceinline__ uint64_t pcg64_next(Pcg64& g) {
    pcg64_step(g);
    const uint64_t hi = (uint64_t)(g.state >> 64), lo = (uint64_t)g.state;
    const uint64_t x = hi ^ lo;
    const unsigned rot = (unsigned)(hi >> 58);
    return (x >> rot) | (x << ((64u - rot) & 63u));
}
// default_rng(seed32): PCG64(SeedSequence(seed32)) -> generate_state(4, uint64) = (initstate, initseq)
__device__ __forceinline__ void pcg64_seed_u32(Pcg64& g, uint32_t seed) {
    uint32_t pool[4], w[8];
    seedseq_pool(&seed, 1, nullptr, 0, pool);
    seedseq_generate(pool, 8, w);
    const uint64_t u0 = w[0] | ((uint64_t)w[1] << 32), u1 = w[2] | ((uint64_t)w[3] << 32);
    const uint64_t u2 = w[4] | ((uint64_t)w[5] << 32), u3 = w[6] | ((uint64_t)w[7] << 32);
    const unsigned __int128 initstate = ((unsigned __int128)u0 << 64) | u1;
    const unsigned __int128 initseq = ((unsigned __int128)u2 << 64) | u3;
    g.state = 0;
    g.inc = (initseq << 1) | 1;
    pcg64_step(g);
    g.state += initstate;
    pcg64_step(g);
}
__device__ __forceinline__ double pcg64_next_double(Pcg64& g) {
    return (double)(pcg64_next(g) >> 11) * (1.0 / 9007199254740992.0);
}

// ---- Generator.standard_normal (ziggurat) -----------------------------------------------------
__device__ __forceinline__ double np_standard_normal(Pcg64& g, const ZigTables& T) {
    constexpr double kR = 3.6541528853610087963519472518;
    constexpr double kInvR = 0.27366123732975827203338247596;
    for (;;) {
        uint64_t r = pcg64_next(g);
        const int idx = (int)(r & 0xff);
        r >>= 8;
        const int sign = (int)(r & 1);
        const uint64_t rabs = (r >> 1) & 0x000fffffffffffffull;
        double x = (double)rabs * T.wi[idx];
        if (sign) x = -x;
        if (rabs < T.ki[idx]) return x;  // ~99.3 % of the draws
        if (idx == 0) {
            for (;;) {  // tail
                const double xx = -kInvR * log1p(-pcg64_next_double(g));
                const double yy = -log1p(-pcg64_next_double(g));
                if (yy + yy > xx * xx) return ((rabs >> 8) & 1) ? -(kR + xx) : kR + xx;
            }
        } else {  // wedge.  exp(-x^2 / 2), |x| < 3.66, only decides accept / reject: the kernel's table exp (1.5 ulp) serves as
                  // well as the device libm's (neither is glibc's bit for bit: a decision can differ where the two sides of the
                  // test agree to ~1e-16, once in ~1e15 wedge draws) at a third of the instructions — and a wave takes this
                  // branch whenever ANY of its 64 lanes does, i.e. for a third of all normals
            if ((T.fi[idx - 1] - T.fi[idx]) * pcg64_next_double(g) + T.fi[idx] < fexp(-0.5 * x * x, T.math_tab, MathRegs::literals())) return x;
        }
    }
}

// Path seed of child `child` of stream `stream_index` of SeedSequence(main_seed):
// SeedSequence(main).spawn(2)[stream].spawn(..)[child].generate_state(1)[0]   (simulation.py:148-149,195-197)
__device__ __forceinline__ uint32_t np_path_seed(const uint32_t* ent, int n_ent, uint32_t stream_index, uint64_t child) {
    uint32_t pool[4], out[1];
    if (child <= 0xffffffffull) {
        const uint32_t key[2] = {stream_index, (uint32_t)child};
        seedseq_pool(ent, n_ent, key, 2, pool);
    } else {  // a spawn-key element >= 2^32 is coerced to two uint32 words
        const uint32_t key[3] = {stream_index, (uint32_t)child, (uint32_t)(child >> 32)};
        seedseq_pool(ent, n_ent, key, 3, pool);
    }
    seedseq_generate(pool, 1, out);
    return out[0];
}

}  // namespace mcr
