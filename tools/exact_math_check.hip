// exact_math_check.hip -- exhaustive / randomised proof, ON THE HARDWARE, that the lean correctly-rounded sequences of
// nlos_device.h (sqrt_cr, rcp_cr, div_lean) return the bits of the compiler's IEEE sqrtf / 1.0f / x / a / b.
//
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math tools/exact_math_check.hip -o /tmp/exact_math_check
//   /tmp/exact_math_check            -> one JSON line (profiles/r05_exact_math.json)
//
// v_sqrt_f32 / v_rcp_f32 / v_rsq_f32 are deterministic functions of their input bits, so a sweep over ALL 2^32 bit
// patterns is a proof for this chip; the two-operand division is swept over 2^33 random pairs of the guarded range plus
// structured edge cases (mantissa all ones / zeros, powers of two, quotients next to rounding boundaries).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../nlos_surface_optimization_amd/csrc/nlos_device.h"

using namespace nlos;

// candidates that were measured against each other (only the winners live in nlos_device.h)
__device__ __forceinline__ float sqrt_cand_a(float x) {      // hardware sqrt + the +-1 ulp residual test (what the compiler emits, minus scaling)
    float y = __builtin_amdgcn_sqrtf(x);
    float ym = __uint_as_float(__float_as_uint(y) - 1u), yp = __uint_as_float(__float_as_uint(y) + 1u);
    float r1 = __fmaf_rn(-ym, y, x), r2 = __fmaf_rn(-yp, y, x);
    y = r1 <= 0.0f ? ym : y;
    y = r2 > 0.0f ? yp : y;
    return y;
}
__device__ __forceinline__ float sqrt_cand_b(float x) {      // hardware sqrt, one Newton step with h = rsq / 2
    float y = __builtin_amdgcn_sqrtf(x);
    float h = 0.5f * __builtin_amdgcn_rsqf(x);
    float r = __fmaf_rn(-y, y, x);
    return __fmaf_rn(r, h, y);
}
__device__ __forceinline__ float sqrt_cand_c(float x) {      // rsq only
    float g = __builtin_amdgcn_rsqf(x);
    float y = x * g, h = 0.5f * g;
    float r = __fmaf_rn(-y, y, x);
    return __fmaf_rn(r, h, y);
}
__device__ __forceinline__ float rcp_cand_a(float x) {       // one Newton step on v_rcp_f32
    float q = __builtin_amdgcn_rcpf(x);
    float e = __fmaf_rn(-x, q, 1.0f);
    return __fmaf_rn(e, q, q);
}
__device__ __forceinline__ float rcp_cand_b(float x) {       // two
    float q = __builtin_amdgcn_rcpf(x);
    float e = __fmaf_rn(-x, q, 1.0f);
    q = __fmaf_rn(e, q, q);
    e = __fmaf_rn(-x, q, 1.0f);
    return __fmaf_rn(e, q, q);
}

struct Counts { unsigned long long bad[16]; unsigned int first[16]; };

// every bit pattern with sign 0 and the exponent field in [elo, ehi]
__global__ void k_sweep_unary(Counts* c, uint32_t elo, uint32_t ehi) {
    const uint64_t n = ((uint64_t)(ehi - elo + 1)) << 23;
    unsigned long long bad[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t bits = (uint32_t)(i + ((uint64_t)elo << 23));
        const float x = __uint_as_float(bits);
        const uint32_t s = __float_as_uint(sqrtf(x));
        const uint32_t r = __float_as_uint(1.0f / x);
        const uint32_t got[8] = {__float_as_uint(sqrt_cand_a(x)), __float_as_uint(sqrt_cand_b(x)), __float_as_uint(sqrt_cand_c(x)),
                                 __float_as_uint(rcp_cand_a(x)),  __float_as_uint(rcp_cand_b(x)),  __float_as_uint(sqrt_cr(x)),
                                 __float_as_uint(rcp_cr(x)),      __float_as_uint(sqrt_cr0(x))};
        const uint32_t want[8] = {s, s, s, r, r, s, r, s};
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (got[k] != want[k]) { if (!bad[k]) atomicMin(&c->first[k], bits); bad[k] += 1; }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (bad[k]) atomicAdd(&c->bad[k], bad[k]);
}

__global__ void k_zero(Counts* c) {
    const float z = threadIdx.x & 1 ? -0.0f : 0.0f;
    if (__float_as_uint(sqrt_cr0(z)) != __float_as_uint(sqrtf(z)) && threadIdx.x == 0) atomicAdd(&c->bad[10], 1ull);
}

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// random float with the exponent field in [elo, ehi], random sign and mantissa; every 8th draw a structured mantissa
__device__ __forceinline__ float rnd_float(uint64_t h, uint32_t elo, uint32_t ehi) {
    const uint32_t e = elo + (uint32_t)((h >> 40) % (ehi - elo + 1));
    uint32_t m = (uint32_t)h & 0x7fffffu;
    const uint32_t kind = (uint32_t)(h >> 24) & 7u;
    if (kind == 0) m = 0u;
    else if (kind == 1) m = 0x7fffffu;
    else if (kind == 2) m &= 0x7u;
    else if (kind == 3) m |= 0x7ffff8u;
    return __uint_as_float(((uint32_t)(h >> 63) << 31) | (e << 23) | m);
}
__global__ void k_sweep_div(Counts* c, uint64_t n, uint64_t seed, uint32_t elo, uint32_t ehi, uint32_t dlo, uint32_t dhi) {
    unsigned long long bad = 0, bad_r = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t h1 = mix64(seed + 2 * i + 1), h2 = mix64(seed + 2 * i + 2);
        float a = rnd_float(h1, elo, ehi);
        const float b = rnd_float(h2, dlo, dhi);
        if ((h1 & 0xff0000000000ull) == 0) a = 0.0f;                       // zero numerators are allowed
        const float want = a / b;
        if (__float_as_uint(div_lean(a, b)) != __float_as_uint(want)) { if (!bad) atomicMin(&c->first[8], __float_as_uint(b)); bad += 1; }
        // the form the kernels use for a wave-uniform or shared denominator: reciprocal once, then div_by()
        const float r = rcp_refined(b);
        if (__float_as_uint(div_by(a, b, r)) != __float_as_uint(want)) { if (!bad_r) atomicMin(&c->first[9], __float_as_uint(b)); bad_r += 1; }
    }
    if (bad) atomicAdd(&c->bad[8], bad);
    if (bad_r) atomicAdd(&c->bad[9], bad_r);
}

int main(int argc, char** argv) {
    Counts* c;
    hipMalloc(&c, sizeof(Counts));
    Counts h;
    auto reset = [&]() { for (int k = 0; k < 16; ++k) { h.bad[k] = 0; h.first[k] = 0xffffffffu; } hipMemcpy(c, &h, sizeof(h), hipMemcpyHostToDevice); };
    auto fetch = [&]() { hipDeviceSynchronize(); hipMemcpy(&h, c, sizeof(h), hipMemcpyDeviceToHost); };
    const char* names[11] = {"sqrt_a(sqrt,+-1ulp)", "sqrt_b(sqrt+rsq)", "sqrt_c(rsq)", "rcp_a(1 step)", "rcp_b(2 steps)", "sqrt_cr", "rcp_cr",
                             "sqrt_cr0", "div_lean", "div_by(rcp_refined)", "sqrt_cr0(+-0)"};
    printf("{");
    // (1) all positive normal floats (exponent fields 1 ... 254), and the guarded range of the kernels separately
    const uint32_t ranges[3][2] = {{1, 254}, {kLeanExpLo, kLeanExpHi}, {127 - 24, 127}};
    const char* rname[3] = {"all_normal", "guarded_range", "unit_interval_2^-24..1"};
    for (int r = 0; r < 3; ++r) {
        reset();
        hipLaunchKernelGGL(k_sweep_unary, dim3(256 * 8), dim3(256), 0, 0, c, ranges[r][0], ranges[r][1]);
        fetch();
        printf("\"%s\": {\"inputs\": %llu", rname[r], (unsigned long long)(ranges[r][1] - ranges[r][0] + 1) << 23);
        for (int k = 0; k < 8; ++k) printf(", \"%s\": {\"differing\": %llu, \"first_bits\": \"0x%08x\"}", names[k], h.bad[k], h.first[k]);
        printf("}, ");
    }
    // (2) zero (the one value outside the range that the sample map produces: T = 0 once in 2^23 draws)
    reset();
    hipLaunchKernelGGL(k_zero, dim3(1), dim3(64), 0, 0, c);
    fetch();
    printf("\"%s\": {\"differing\": %llu}, ", names[10], h.bad[10]);
    // (3) two-operand division: random + structured pairs of the guarded range
    const uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : (1ull << 33);
    const uint32_t reg[2][4] = {{kLeanExpLo, kLeanExpHi, kLeanExpLo, kLeanExpHi}, {kLeanNumLo, kLeanNumHi, kLeanDenLo, kLeanDenHi}};
    for (int g = 0; g < 2; ++g) {
        reset();
        hipLaunchKernelGGL(k_sweep_div, dim3(256 * 8), dim3(256), 0, 0, c, n, 0x1234567ull + g, reg[g][0], reg[g][1], reg[g][2], reg[g][3]);
        fetch();
        printf("\"division_%d\": {\"pairs\": %llu, \"numerator_fields\": [%u, %u], \"denominator_fields\": [%u, %u]", g + 1, (unsigned long long)n,
               reg[g][0], reg[g][1], reg[g][2], reg[g][3]);
        for (int k = 8; k < 10; ++k) printf(", \"%s\": {\"differing\": %llu, \"first_den_bits\": \"0x%08x\"}", names[k], h.bad[k], h.first[k]);
        printf(g == 0 ? "}, " : "}}\n");
    }
    return 0;
}
