// Per-instruction issue cost on gfx950: each kernel runs a long unrolled stream of ONE VALU
// instruction kind (8 independent chains per lane), 4 waves per SIMD on every CU.
//   hipcc --offload-arch=gfx950 -O3 -o valu_cost valu_cost.hip && ./valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define ITERS 2048
#define CHAINS 8

#define KERNEL(NAME, TYPE, INIT, BODY)                                                  \
    __global__ __launch_bounds__(256) void k_##NAME(TYPE* out, TYPE seed) {             \
        TYPE v[CHAINS];                                                                 \
        _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) v[c] = INIT;                 \
        for (int i = 0; i < ITERS; ++i) {                                               \
            _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) { TYPE& x = v[c]; BODY; } \
        }                                                                               \
        TYPE acc = v[0];                                                                \
        _Pragma("unroll") for (int c = 1; c < CHAINS; ++c) acc = acc + v[c];            \
        if (acc == (TYPE)123456789) out[0] = acc;                                       \
    }

#define ASM1(op) asm volatile(op " %0, %0" : "+v"(x))
#define ASM2(op) asm volatile(op " %0, %0, %1" : "+v"(x) : "v"(seed))
#define ASM3(op) asm volatile(op " %0, %0, %1, %0" : "+v"(x) : "v"(seed))

KERNEL(fma_f64, double, seed + threadIdx.x + c, ASM3("v_fma_f64"))
KERNEL(mul_f64, double, seed + threadIdx.x + c, ASM2("v_mul_f64"))
KERNEL(add_f64, double, seed + threadIdx.x + c, ASM2("v_add_f64"))
KERNEL(max_f64, double, seed + threadIdx.x + c, ASM2("v_max_f64"))
KERNEL(rcp_f64, double, seed + threadIdx.x + c, ASM1("v_rcp_f64"))
KERNEL(rsq_f64, double, seed + threadIdx.x + c, ASM1("v_rsq_f64"))
KERNEL(sqrt_f64, double, seed + threadIdx.x + c, ASM1("v_sqrt_f64"))
KERNEL(ldexp_f64, double, seed + threadIdx.x + c, asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(x)))
KERNEL(rndne_f64, double, seed + threadIdx.x + c, ASM1("v_rndne_f64"))
KERNEL(frexp_mant_f64, double, seed + threadIdx.x + c, ASM1("v_frexp_mant_f64"))
KERNEL(div_scale_f64, double, seed + threadIdx.x + c, asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(x) : "v"(seed) : "vcc"))
KERNEL(div_fixup_f64, double, seed + threadIdx.x + c, asm volatile("v_div_fixup_f64 %0, %0, %1, %0" : "+v"(x) : "v"(seed)))
KERNEL(div_fmas_f64, double, seed + threadIdx.x + c, asm volatile("v_div_fmas_f64 %0, %0, %1, %0" : "+v"(x) : "v"(seed) : "vcc"))
KERNEL(fma_f32, float, seed + threadIdx.x + c, ASM3("v_fma_f32"))
KERNEL(add_u32, uint32_t, seed + threadIdx.x + c, ASM2("v_add_u32"))
KERNEL(xor_b32, uint32_t, seed + threadIdx.x + c, ASM2("v_xor_b32"))
KERNEL(mul_lo_u32, uint32_t, seed + threadIdx.x + c, ASM2("v_mul_lo_u32"))
KERNEL(mul_hi_u32, uint32_t, seed + threadIdx.x + c, ASM2("v_mul_hi_u32"))
KERNEL(cndmask_b32, uint32_t, seed + threadIdx.x + c, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(seed) : "vcc"))
KERNEL(mad_u64_u32, uint64_t, seed + threadIdx.x + c, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"((uint32_t)seed), "v"((uint32_t)threadIdx.x) : "vcc"))
KERNEL(cvt_f64_u32, double, seed + threadIdx.x + c, asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(x) : "v"((uint32_t)threadIdx.x)))
KERNEL(cmp_f64, double, seed + threadIdx.x + c, asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(x), "v"(seed) : "vcc"))
KERNEL(mov_b32, uint32_t, seed + threadIdx.x + c, asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(seed)))
KERNEL(cmp_lt_u64, uint64_t, seed + threadIdx.x + c, asm volatile("v_cmp_lt_u64_e64 s[20:21], %0, %1" : : "v"(x), "v"(seed) : "s20", "s21"))
KERNEL(cmp_lt_u32, uint32_t, seed + threadIdx.x + c, asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %1" : : "v"(x), "v"(seed) : "s20", "s21"))
KERNEL(cmp_eq_u64, uint64_t, seed + threadIdx.x + c, asm volatile("v_cmp_eq_u64_e64 s[20:21], %0, %1" : : "v"(x), "v"(seed) : "s20", "s21"))
KERNEL(cmp_addc_u64, uint64_t, seed + threadIdx.x + c, { uint32_t lo32 = (uint32_t)x; asm volatile("v_cmp_lt_u64_e64 s[20:21], %1, %2\n\tv_addc_co_u32_e64 %0, vcc, 0, %0, s[20:21]" : "+v"(lo32) : "v"(x), "v"(seed) : "s20", "s21", "vcc"); x = (x & 0xFFFFFFFF00000000ull) | lo32; })
KERNEL(cmp_addc_u32, uint32_t, seed + threadIdx.x + c, asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %1\n\tv_addc_co_u32_e64 %0, vcc, 0, %0, s[20:21]" : "+v"(x) : "v"(seed) : "s20", "s21", "vcc"))
KERNEL(cndmask_sgpr, uint32_t, seed + threadIdx.x + c, asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(x) : "v"(seed)))
KERNEL(cndmask_vcc_e32, uint32_t, seed + threadIdx.x + c, asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x) : "v"(seed)))
KERNEL(cndmask_vcc_e64, uint32_t, seed + threadIdx.x + c, asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(x) : "v"(seed)))
KERNEL(addc_vcc_e32, uint32_t, seed + threadIdx.x + c, asm volatile("v_addc_co_u32_e32 %0, vcc, %0, %1, vcc" : "+v"(x) : "v"(seed)))
KERNEL(cndmask_dpp_free, uint32_t, seed + threadIdx.x + c, asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(x) : "v"(seed)))
KERNEL(rcp_f32, float, seed + threadIdx.x + c, ASM1("v_rcp_f32"))
KERNEL(cvt_f32_f64, double, seed + threadIdx.x + c, asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(*(float*)&x) : "v"(seed)))
KERNEL(cvt_f64_f32, double, seed + threadIdx.x + c, asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(x) : "v"((float)threadIdx.x)))
KERNEL(min_f64, double, seed + threadIdx.x + c, ASM2("v_min_f64"))
KERNEL(and_b32, uint32_t, seed + threadIdx.x + c, ASM2("v_and_b32"))
KERNEL(mov_b64, double, seed + threadIdx.x + c, asm volatile("v_mov_b64 %0, %1" : "=v"(x) : "v"(seed)))
KERNEL(addc_only, uint32_t, seed + threadIdx.x + c, asm volatile("v_addc_co_u32_e64 %0, vcc, 0, %0, s[20:21]" : "+v"(x) : : "vcc"))

// mixes: is the transcendental pipe separate from the fp64 pipe (do the costs add or overlap)?  Every chain does one
// independent FMA per iteration; every 4th chain also issues one transcendental (or four xors) with a throw-away result.
// The printed figure is per FMA: compare with fma_f64 alone and with fma + the other op's cost / 4.
#define FMA1 asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(x) : "v"(seed))
KERNEL(mix_rcp64_per_4fma, double, seed + threadIdx.x + c, { if ((c & 3) == 0) { double t; asm volatile("v_rcp_f64 %0, %1" : "=v"(t) : "v"(seed)); } FMA1; })
KERNEL(mix_rcp32_per_4fma, double, seed + threadIdx.x + c, { if ((c & 3) == 0) { float t; asm volatile("v_rcp_f32 %0, %1" : "=v"(t) : "v"((float)seed)); } FMA1; })
KERNEL(mix_4xor_per_4fma, double, seed + threadIdx.x + c, { uint32_t t; asm volatile("v_xor_b32 %0, %1, %1" : "=v"(t) : "v"((uint32_t)c)); FMA1; })
KERNEL(mix_mad64_per_fma, double, seed + threadIdx.x + c, { uint64_t t; asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, 0" : "=v"(t) : "v"((uint32_t)c) : "vcc"); FMA1; })
KERNEL(exp_f32, float, seed + threadIdx.x + c, ASM1("v_exp_f32"))
KERNEL(bfi_b32, uint32_t, seed + threadIdx.x + c, ASM3("v_bfi_b32"))
KERNEL(bitop3_b32, uint32_t, seed + threadIdx.x + c, asm volatile("v_bitop3_b32 %0, %0, %1, %0 bitop3:0x96" : "+v"(x) : "v"(seed)))
KERNEL(bitop3_sgpr, uint32_t, seed + threadIdx.x + c, asm volatile("v_bitop3_b32 %0, %0, %1, s20 bitop3:0x96" : "+v"(x) : "v"(seed)))
KERNEL(lshl_add_u32, uint32_t, seed + threadIdx.x + c, asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x) : "v"(seed)))
KERNEL(add3_u32, uint32_t, seed + threadIdx.x + c, ASM3("v_add3_u32"))
KERNEL(frexp_exp_f64, double, seed + threadIdx.x + c, { uint32_t e; asm volatile("v_frexp_exp_i32_f64 %0, %1" : "=v"(e) : "v"(x)); })

template <typename T>
static void run(const char* name, void (*k)(T*, T), T seed) {
    T* d;
    hipMalloc(&d, 64);
    const int blocks = 256 * 4;  // 4 workgroups of 256 per CU = 4 waves per SIMD
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, seed);
    hipDeviceSynchronize();
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, seed);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    // wave-instructions per SIMD = 4 waves * ITERS * CHAINS; cycles at 2.4 GHz nominal
    const double insts_per_simd = 4.0 * ITERS * CHAINS;
    const double ns = best * 1e6;
    printf("%-16s %8.3f ms  %6.2f ns/wave-inst/SIMD  ~%5.2f cycles @2.4GHz\n", name, best, ns / insts_per_simd, ns / insts_per_simd * 2.4);
    hipFree(d);
}

int main() {
#define RUN(NAME, TYPE, SEED) run<TYPE>(#NAME, k_##NAME, (TYPE)SEED)
    RUN(fma_f64, double, 1.0000001); RUN(mul_f64, double, 1.0000001); RUN(add_f64, double, 1e-9); RUN(max_f64, double, 1.5);
    RUN(rcp_f64, double, 1.5); RUN(rsq_f64, double, 1.5); RUN(sqrt_f64, double, 1.5); RUN(ldexp_f64, double, 1e-300);
    RUN(rndne_f64, double, 1.5); RUN(frexp_mant_f64, double, 1.5);
    RUN(div_scale_f64, double, 1.5); RUN(div_fixup_f64, double, 1.5); RUN(div_fmas_f64, double, 1.0000001);
    RUN(fma_f32, float, 1.0000001f); RUN(add_u32, uint32_t, 3); RUN(xor_b32, uint32_t, 3); RUN(mul_lo_u32, uint32_t, 3);
    RUN(mul_hi_u32, uint32_t, 3); RUN(cndmask_b32, uint32_t, 3); RUN(mad_u64_u32, uint64_t, 3); RUN(cvt_f64_u32, double, 1.0);
    RUN(cmp_f64, double, 1.0); RUN(mov_b32, uint32_t, 3);
    RUN(cmp_lt_u64, uint64_t, 3); RUN(cmp_lt_u32, uint32_t, 3); RUN(cmp_eq_u64, uint64_t, 3);
    RUN(cmp_addc_u64, uint64_t, 3); RUN(cmp_addc_u32, uint32_t, 3); RUN(addc_only, uint32_t, 3);
    RUN(cndmask_sgpr, uint32_t, 3); RUN(cndmask_vcc_e32, uint32_t, 3); RUN(cndmask_vcc_e64, uint32_t, 3); RUN(addc_vcc_e32, uint32_t, 3); RUN(cndmask_dpp_free, uint32_t, 3); RUN(rcp_f32, float, 1.5f); RUN(cvt_f32_f64, double, 1.5); RUN(cvt_f64_f32, double, 1.5); RUN(min_f64, double, 1.5); RUN(and_b32, uint32_t, 3); RUN(mov_b64, double, 1.5);
    RUN(bfi_b32, uint32_t, 3); RUN(bitop3_b32, uint32_t, 3); RUN(bitop3_sgpr, uint32_t, 3); RUN(lshl_add_u32, uint32_t, 3); RUN(add3_u32, uint32_t, 3);
    RUN(frexp_exp_f64, double, 1.5); RUN(exp_f32, float, 1.5f);
    RUN(mix_rcp64_per_4fma, double, 1.0000001); RUN(mix_rcp32_per_4fma, double, 1.0000001); RUN(mix_4xor_per_4fma, double, 1.0000001);
    RUN(mix_mad64_per_fma, double, 1.0000001);
    return 0;
}
