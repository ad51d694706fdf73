#!/usr/bin/env python3
"""Executable model of NumPy's default_rng(seed).standard_normal stream
(SeedSequence -> PCG64 XSL-RR 128/64 -> 256-layer ziggurat), written from the published
algorithms, and validated bit-for-bit against the installed NumPy.  It is the blueprint of the
device implementation (csrc/mcr_numpy_rng.h) and generates its tables."""

from __future__ import annotations

import math
import sys
from decimal import Decimal, getcontext

import numpy as np

M32 = 0xFFFFFFFF
M64 = (1 << 64) - 1
M128 = (1 << 128) - 1

# ---- SeedSequence (O'Neill's seed_seq_fe128 variant used by numpy.random.bit_generator) ----
INIT_A, MULT_A = 0x43B0D7E5, 0x931E8875
INIT_B, MULT_B = 0x8B51F9DD, 0x58F38DED
MIX_L, MIX_R = 0xCA01F9DD, 0x4973F715
XSHIFT = 16
POOL = 4


def _u32_words(n: int):
    if n == 0:
        return [0]
    out = []
    while n:
        out.append(n & M32)
        n >>= 32
    return out


def seedseq_pool(entropy: int, spawn_key=()):
    run = _u32_words(entropy)
    spawn = list(spawn_key)
    if spawn and len(run) < POOL:
        run = run + [0] * (POOL - len(run))
    ent = run + spawn
    hc = [INIT_A]

    def hashmix(v):
        v = (v ^ hc[0]) & M32
        hc[0] = (hc[0] * MULT_A) & M32
        v = (v * hc[0]) & M32
        return (v ^ (v >> XSHIFT)) & M32

    def mix(x, y):
        r = (MIX_L * x - MIX_R * y) & M32
        return (r ^ (r >> XSHIFT)) & M32

    pool = [hashmix(ent[i] if i < len(ent) else 0) for i in range(POOL)]
    for i_src in range(POOL):
        for i_dst in range(POOL):
            if i_src != i_dst:
                pool[i_dst] = mix(pool[i_dst], hashmix(pool[i_src]))
    for i_src in range(POOL, len(ent)):
        for i_dst in range(POOL):
            pool[i_dst] = mix(pool[i_dst], hashmix(ent[i_src]))
    return pool


def generate_state_u32(pool, n_words: int):
    hc = INIT_B
    out = []
    for i in range(n_words):
        v = pool[i % POOL] ^ hc
        hc = (hc * MULT_B) & M32
        v = (v * hc) & M32
        out.append((v ^ (v >> XSHIFT)) & M32)
    return out


# ---- PCG64 ----
PCG_MULT = (2549297995355413924 << 64) | 4865540595714422341


class PCG64:
    def __init__(self, seed_u32: int):
        w = generate_state_u32(seedseq_pool(seed_u32), 8)
        u64 = [w[2 * i] | (w[2 * i + 1] << 32) for i in range(4)]
        initstate = (u64[0] << 64) | u64[1]
        initseq = (u64[2] << 64) | u64[3]
        self.inc = ((initseq << 1) | 1) & M128
        self.state = 0
        self._step()
        self.state = (self.state + initstate) & M128
        self._step()

    def _step(self):
        self.state = (self.state * PCG_MULT + self.inc) & M128

    def next64(self) -> int:
        self._step()
        s = self.state
        x = ((s >> 64) ^ s) & M64
        rot = s >> 122
        return ((x >> rot) | (x << ((-rot) & 63))) & M64


# ---- ziggurat tables (Marsaglia & Tsang 2000 / Doornik 2005, 256 layers) ----
ZIG_R = 3.6541528853610087963519472518
ZIG_INV_R = 0.27366123732975827203338247596


def make_tables():
    getcontext().prec = 60
    r = Decimal("3.6541528853610087963519472518")

    def f(x):
        return (-(x * x) / 2).exp()

    # strip area: v = r f(r) + integral_r^inf f = r f(r) + sqrt(pi/2) erfc(r / sqrt 2)   (published
    # to 12 digits as 0.00492867323399; the tables need it to full precision)
    pi = Decimal("3.14159265358979323846264338327950288419716939937510582097494459")
    t = r / Decimal(2).sqrt()
    term, erf_sum, n = t, t, 0
    while abs(term) > Decimal(10) ** -55:  # erf(t) = 2/sqrt(pi) sum (-1)^n t^(2n+1) / (n! (2n+1))
        n += 1
        term = -term * t * t / n
        erf_sum += term / (2 * n + 1)
    erfc = 1 - 2 / pi.sqrt() * erf_sum
    v = r * f(r) + (pi / 2).sqrt() * erfc

    # Marsaglia & Tsang's zigset layout: index 0 is the base strip (q = v / f(r)), x grows with the
    # index up to x[255] = r; ki[0] = r/q, ki[1] = 0, ki[i+1] = x[i]/x[i+1]; fi[0] = 1.
    x = [Decimal(0)] * 256
    q = v / f(r)
    x[255] = r
    for i in range(254, 0, -1):
        x[i] = (-2 * (v / x[i + 1] + f(x[i + 1])).ln()).sqrt()
    two52 = Decimal(2) ** 52
    ki = [0] * 256
    wi = [0.0] * 256
    fi = [0.0] * 256
    ki[0] = int((r / q) * two52)
    ki[1] = 0
    wi[0] = float(q / two52)
    fi[0] = 1.0
    for i in range(1, 256):
        wi[i] = float(x[i] / two52)
        fi[i] = float(f(x[i]))
        if i + 1 < 256:
            ki[i + 1] = int((x[i] / x[i + 1]) * two52)
    return ki, wi, fi


class Ziggurat:
    def __init__(self, ki, wi, fi):
        self.ki, self.wi, self.fi = ki, wi, fi

    def normal(self, g: PCG64) -> float:
        while True:
            r = g.next64()
            idx = r & 0xFF
            r >>= 8
            sign = r & 1
            rabs = (r >> 1) & 0x000FFFFFFFFFFFFF
            x = rabs * self.wi[idx]
            if sign:
                x = -x
            if rabs < self.ki[idx]:
                return x
            if idx == 0:
                while True:
                    xx = -ZIG_INV_R * math.log1p(-((g.next64() >> 11) * (1.0 / 9007199254740992.0)))
                    yy = -math.log1p(-((g.next64() >> 11) * (1.0 / 9007199254740992.0)))
                    if yy + yy > xx * xx:
                        return -(ZIG_R + xx) if ((rabs >> 8) & 1) else ZIG_R + xx
            else:
                u = (g.next64() >> 11) * (1.0 / 9007199254740992.0)
                if (self.fi[idx - 1] - self.fi[idx]) * u + self.fi[idx] < math.exp(-0.5 * x * x):
                    return x


def path_seed_u32(main_seed: int, stream_index: int, child: int) -> int:
    """SeedSequence(main).spawn(2)[stream].spawn(..)[child].generate_state(1)[0]  (simulation.py:148-149,195-197)."""
    return generate_state_u32(seedseq_pool(main_seed, (stream_index, child)), 1)[0]


def validate(n_draws=300_000):
    ki, wi, fi = make_tables()
    z = Ziggurat(ki, wi, fi)
    # SeedSequence pools / states
    for ent, key in [(0, ()), (12345, ()), (12345, (1,)), (12345, (1, 7)), (2**40 + 3, (0, 99)), (2**130 + 5, (1, 2))]:
        ss = np.random.SeedSequence(ent, spawn_key=key)
        assert [int(v) for v in ss.pool] == seedseq_pool(ent, key), (ent, key)
        assert [int(v) for v in ss.generate_state(6)] == generate_state_u32(seedseq_pool(ent, key), 6)
    root = np.random.SeedSequence(12345)
    s, f_ = root.spawn(2)
    kids = f_.spawn(5)
    assert [int(k.generate_state(1)[0]) for k in kids] == [path_seed_u32(12345, 1, j) for j in range(5)]
    # PCG64 raw stream
    for seed in (0, 1, 3735928559, 42):
        g = PCG64(seed)
        ref = np.random.PCG64(seed).random_raw(16)
        assert [g.next64() for _ in range(16)] == [int(v) for v in ref], seed
    # normals, bit for bit
    bad = 0
    for seed in (5, 99, 2024):
        g = PCG64(seed)
        ref = np.random.default_rng(seed).standard_normal(n_draws // 3)
        mine = np.array([z.normal(g) for _ in range(n_draws // 3)])
        bad += int((mine.view(np.uint64) != ref.view(np.uint64)).sum())
    print(f"validated: SeedSequence, PCG64 and {n_draws} normals; mismatching normals: {bad}")
    return bad


if __name__ == "__main__":
    sys.exit(1 if validate() else 0)
