// nlos_device.h -- device-side numeric contract of the gfx950 transient renderer.
//
// Everything that decides accept/reject of a surface sample is IEEE fp32
// evaluated in a fixed order, exactly as written (this translation unit is built with
// -ffp-contract=off; correctly rounded sqrt/div are hipcc defaults), so the HIP
// kernels and the independent CPU oracle make bit-identical visibility decisions:
//   dot(a,b)   = fma(a.z, b.z, fma(a.y, b.y, a.x*b.x))          (explicit fma on both sides)
//   u*A+v*B+w*C = fma(w, C, fma(v, B, u*A)) per component         (explicit fma on both sides)
//   cross(a,b) = (fma(a.y, b.z, -(a.z*b.y)), fma(a.z, b.x, -(a.x*b.z)), fma(a.x, b.y, -(a.y*b.x)))   (explicit fma on both sides)
// Reference behaviour restated here (paths relative to transient_rendering_cython/):
//   sample map            smoothed_transient/transient_and_gradient.cpp:178-196
//   float from random bits stratified_transient_raytracer/rng_sse.h:33-42
//   triangle test         Embree 3 Moeller-Trumbore (published algorithm), rows I/E
//   GGX                   ggx/ggx_confocal.cpp:13-232
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nlos_contract.h"

namespace nlos {

struct V3 { float x, y, z; };

__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
// Numeric contract: a dot product is one multiplication and two fused multiply-adds, in this
// order -- the oracle's dot3() is the same expression (fmaf), so both sides round identically; everything that is not
// written as an explicit fma is a plain IEEE operation (-ffp-contract=off).  3 instructions instead of 5.
__device__ __forceinline__ float dot(V3 a, V3 b) { return __fmaf_rn(a.z, b.z, __fmaf_rn(a.y, b.y, a.x * b.x)); }
// one multiplication and one fused multiply-add per component (oracle: cross3(), the same expression).  Not exactly
// antisymmetric in its arguments: where two places must agree on a normal they evaluate the same call (make_tri below
// and load_face() in render_common.h both form cross(p1 - p0, p2 - p0)).
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return mk(__fmaf_rn(a.y, b.z, -(a.z * b.y)), __fmaf_rn(a.z, b.x, -(a.x * b.z)), __fmaf_rn(a.x, b.y, -(a.y * b.x)));
}
// u a + v b + w c, per component u a first, then the two fused multiply-adds (oracle: bary3())
__device__ __forceinline__ V3 bary(float u, V3 a, float v, V3 b, float w, V3 c) {
    return mk(__fmaf_rn(w, c.x, __fmaf_rn(v, b.x, u * a.x)), __fmaf_rn(w, c.y, __fmaf_rn(v, b.y, u * a.y)),
              __fmaf_rn(w, c.z, __fmaf_rn(v, b.z, u * a.z)));
}
__device__ __forceinline__ V3 ld3(const float* p) { return mk(p[0], p[1], p[2]); }

// ---------------------------------------------------------------- RNG (row R)
// k-th output of splitmix64 seeded with `seed`; S from the low, T from the high
// 32 bits; 23-bit mantissa floats in [0,1).  k = ((l*F + f)*spt + s) with the
// GLOBAL source index l and the ORIGINAL face index f, so results do not depend
// on source sharding or on the BVH's face ordering.
__device__ __forceinline__ float u32_to_unit(uint32_t x) {
    return __uint_as_float((x >> 9) | 0x3f800000u) - 1.0f;
}
__device__ __forceinline__ void sample_st(uint64_t seed, uint64_t k, float& S, float& T) {
    uint64_t z = seed + (k + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    S = u32_to_unit((uint32_t)(z & 0xffffffffull));
    T = u32_to_unit((uint32_t)(z >> 32));
}

// The same draw with the source's part of the key folded in on the scalar unit: for k = k0 + c (k0 wave-uniform, c a
// 32-bit per-lane count) seed + (k + 1) G = [seed + (k0 + 1) G] + c G  (mod 2^64), so a lane pays one 32 x 64-bit
// multiply-add instead of forming the 64-bit key, a 64 x 64-bit product and a 64-bit add: five quarter-rate
// instructions fewer per ray, the same bits.
__device__ __forceinline__ uint64_t sample_zbase(uint64_t seed, uint64_t k0) { return seed + (k0 + 1ull) * 0x9E3779B97F4A7C15ull; }
__device__ __forceinline__ void sample_st_c(uint64_t zbase, uint32_t c, float& S, float& T) {
    uint64_t z = zbase + (uint64_t)c * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    S = u32_to_unit((uint32_t)(z & 0xffffffffull));
    T = u32_to_unit((uint32_t)(z >> 32));
}

// ------------------------------------------------ lean correctly rounded sqrt / reciprocal / division
// hipcc's sqrtf and `/` are correctly rounded for EVERY input: v_sqrt_f32 / v_rcp_f32 (1 ulp) wrapped in a denormal
// pre-scale, the refinement, an un-scale and an inf / nan / zero fix-up -- 17 instructions per sqrtf, 12 per division,
// of which only the refinement does anything for operands of ordinary magnitude.  The functions below are that
// refinement alone: THE SAME BITS as sqrtf(x) / (1.0f / x) / (a / b) for operands whose exponent field lies in
// [kLeanExpLo, kLeanExpHi] (2^-60 <= |x| < 2^61; sqrt_cr and a numerator of div_*: also zero), proven on the hardware by
// tools/exact_math_check.hip (all 2^32 bit patterns for the unary functions; 2^33 random + structured pairs for the
// division, whose sequence is moreover the compiler's own minus v_div_scale / v_div_fixup, which are the identity on
// that range) -> profiles/r05_exact_math.json, tests/test_gpu_exact_math.py.  Callers guarantee the range
// (forward_grid.hip: per source through `frame_ok`, per lane by a guard that branches to the IEEE form).
constexpr uint32_t kLeanExpLo = 127 - 60, kLeanExpHi = 127 + 60;
constexpr float kLeanMin = 0x1p-60f, kLeanMax = 0x1p60f;
// second regime of the division (the form factor num / h / h and the bin (2 h - lb) / res): numerator fields
// [127 - 90, 127 + 60] (or zero) over denominator fields [127 - 30, 127 + 30] -- quotients stay normal, none of
// v_div_scale's conditions is met (numerator field > 23, exponent difference < 96), swept like the first
constexpr uint32_t kLeanNumLo = 127 - 90, kLeanNumHi = 127 + 60, kLeanDenLo = 127 - 30, kLeanDenHi = 127 + 30;
constexpr float kLeanNumMax = 0x1p30f;      // form-factor numerator: <= 2^30 keeps num / h inside the numerator range of the second division
// v_rsq_f32 and one residual step (candidates with v_sqrt_f32 -- + the compiler's own +-1 ulp test, or + a step with
// h = rsq / 2 -- are exact on the same range and cost 34 / 23 issue cycles against 17 for this one)
__device__ __forceinline__ float sqrt_cr(float x) {
    float g = __builtin_amdgcn_rsqf(x);
    float y = x * g, h = 0.5f * g;
    float r = __fmaf_rn(-y, y, x);
    return __fmaf_rn(r, h, y);
}
// the same for x in {0} u [2^-60, 2^61): the reciprocal square root of max(x, 2^-60) leaves 0 * g = 0 for x = 0
__device__ __forceinline__ float sqrt_cr0(float x) {
    float g = __builtin_amdgcn_rsqf(fmaxf(x, kLeanMin));
    float y = x * g, h = 0.5f * g;
    float r = __fmaf_rn(-y, y, x);
    return __fmaf_rn(r, h, y);
}
__device__ __forceinline__ float rcp_cr(float x) {
    float q = __builtin_amdgcn_rcpf(x);
    float e = __fmaf_rn(-x, q, 1.0f);
    return __fmaf_rn(e, q, q);
}
// a / b in two halves, for a denominator shared by several divisions (or wave-uniform): r = rcp_refined(b) once
__device__ __forceinline__ float rcp_refined(float b) {
    float r = __builtin_amdgcn_rcpf(b);
    float e = __fmaf_rn(-b, r, 1.0f);
    return __fmaf_rn(e, r, r);
}
__device__ __forceinline__ float div_by(float a, float b, float r) {
    float q = a * r;
    float e = __fmaf_rn(-b, q, a);
    q = __fmaf_rn(e, r, q);
    e = __fmaf_rn(-b, q, a);
    return __fmaf_rn(e, r, q);
}
__device__ __forceinline__ float div_lean(float a, float b) { return div_by(a, b, rcp_refined(b)); }

// ------------------------------------------------------------ triangle record
// 64-byte record (one cache-line half, 4 x dwordx4), sorted (Morton) order:
//   p0, e1 = p0-p1, e2 = p2-p0, ng = e2 x e1, zmin (smallest vertex z), original face id
// Grazing rule (numeric contract, DESIGN.md section 2, include/nlos_contract.h): a ray that meets a
// triangle's plane at less than asin(2^-10) = 0.056 degrees does not hit it: |ng . d| >= gmin = |ng| / 1024 =
// kGrazeRatio * area (for a unit direction; row E scales the bound by |d|) is part of the hit test, for the sampled
// face and for occluders alike.  Below that angle t = T / den is only as good as den and the reported hit can lie
// millimetres off the ray, where no culled query can follow it; above it every back-end (perspective grid, tiled
// grid, BVH packets, stackless BVH) enumerates exactly the hits of the all-faces definition.
constexpr float kGrazeRatio = NLOS_GRAZE_RATIO;
struct Tri { V3 p0, e1, e2, ng; float gmin; };
constexpr int kTriStride = 4;   // float4 per record: p0, e1, e2, ng | zmin, face id, area, 1 / (2 area)

__device__ __forceinline__ Tri make_tri(V3 p0, V3 p1, V3 p2) {
    Tri t;
    t.p0 = p0;
    t.e1 = p0 - p1;
    t.e2 = p2 - p0;
    t.ng = cross(-t.e1, t.e2);          // = cross(p1 - p0, p2 - p0) bit for bit (-e1 is p1 - p0 exactly)
    t.gmin = kGrazeRatio * (sqrtf(dot(t.ng, t.ng)) / 2.0f);
    return t;
}

__device__ __forceinline__ Tri load_tri(const float4* __restrict__ tris, int j) {
    float4 a = tris[kTriStride * j], b = tris[kTriStride * j + 1], c = tris[kTriStride * j + 2];
    Tri t;
    t.p0 = mk(a.x, a.y, a.z);
    t.e1 = mk(a.w, b.x, b.y);
    t.e2 = mk(b.z, b.w, c.x);
    t.ng = mk(c.y, c.z, c.w);
    t.gmin = kGrazeRatio * tris[kTriStride * j + 3].z;          // area, evaluated by the scene build
    return t;
}

// the 48-byte part only (candidate loops: the grazing bound is fetched by the rare lanes that have a valid hit)
__device__ __forceinline__ Tri load_tri48(const float4* __restrict__ tris, int j) {
    float4 a = tris[kTriStride * j], b = tris[kTriStride * j + 1], c = tris[kTriStride * j + 2];
    Tri t;
    t.p0 = mk(a.x, a.y, a.z);
    t.e1 = mk(a.w, b.x, b.y);
    t.e2 = mk(b.z, b.w, c.x);
    t.ng = mk(c.y, c.z, c.w);
    t.gmin = 0.0f;
    return t;
}

__device__ __forceinline__ float flipsign(float x, bool neg) { return neg ? -x : x; }

// Embree-3 style Moeller-Trumbore with tnear = 0, tfar = inf.  On a hit writes
// t and the barycentrics (u -> 2nd vertex, v -> 3rd vertex).
// LEAN: the reciprocal as rcp_cr() where |den| lies in its range (the rare lane outside takes the IEEE division): same bits.
template <bool LEAN = false>
__device__ __forceinline__ bool tri_test(const Tri& tr, V3 o, V3 d, float& t, float& u, float& v) {
    V3 c = tr.p0 - o;
    V3 r = cross(c, d);
    float den = dot(tr.ng, d);
    float aden = fabsf(den);
    bool sg = (__float_as_uint(den) >> 31) != 0u;
    float U = flipsign(dot(r, tr.e2), sg);
    float Vv = flipsign(dot(r, tr.e1), sg);
    bool ok = (den != 0.0f) && (U >= 0.0f) && (Vv >= 0.0f) && (U + Vv <= aden);
    if (!ok) return false;
    float Tn = flipsign(dot(tr.ng, c), sg);
    if (!(0.0f < Tn)) return false;
    if (!(aden >= tr.gmin)) return false;          // grazing rule
    float rcp;
    if (LEAN && __builtin_expect(aden >= kLeanMin && aden <= kLeanMax, 1)) rcp = rcp_cr(aden);
    else rcp = 1.0f / aden;
    u = U * rcp;
    v = Vv * rcp;
    t = Tn * rcp;
    return true;
}

// Same arithmetic as tri_test(), straight-line (no early-outs), distance only: used in the
// candidate loops, where a predicated test is cheaper than divergent branches.
__device__ __forceinline__ bool tri_hit_t(const Tri& tr, V3 o, V3 d, float& t) {
    V3 c = tr.p0 - o;
    V3 r = cross(c, d);
    float den = dot(tr.ng, d);
    float aden = fabsf(den);
    bool sg = (__float_as_uint(den) >> 31) != 0u;
    float U = flipsign(dot(r, tr.e2), sg);
    float Vv = flipsign(dot(r, tr.e1), sg);
    float Tn = flipsign(dot(tr.ng, c), sg);
    t = Tn * (1.0f / aden);
    return (den != 0.0f) & (U >= 0.0f) & (Vv >= 0.0f) & (U + Vv <= aden) & (0.0f < Tn) & (aden >= tr.gmin);
}

// Occluder test against the own-face hit at distance t_self (original face id self_fid):
// does triangle `tr` (original id via face_id[k]) give a valid hit that wins the closest-hit rule?
// Same arithmetic as tri_test(); the division is only reached by lanes with a valid hit (rare).
// `tr` may come from load_tri48(): the grazing bound of triangle k is read from `tris` by the lanes that need it.
__device__ __forceinline__ bool tri_occludes(const Tri& tr, V3 o, V3 d, float t_self, int self_fid,
                                             const int* __restrict__ face_id, int k,
                                             const float4* __restrict__ tris) {
    V3 c = tr.p0 - o;
    V3 r = cross(c, d);
    float den = dot(tr.ng, d);
    float aden = fabsf(den);
    bool sg = (__float_as_uint(den) >> 31) != 0u;
    float U = flipsign(dot(r, tr.e2), sg);
    float Vv = flipsign(dot(r, tr.e1), sg);
    float Tn = flipsign(dot(tr.ng, c), sg);
    bool valid = (den != 0.0f) & (U >= 0.0f) & (Vv >= 0.0f) & (U + Vv <= aden) & (0.0f < Tn);
    bool occ = false;
    if (valid) {
        const float gmin = kGrazeRatio * reinterpret_cast<const float*>(tris)[4 * (kTriStride * k + 3) + 2];
        // (one lane in twenty reaches this block, but nearly every wave does: the reciprocal in its lean form, same bits)
        float t;
        if (__builtin_expect(aden >= kLeanMin && aden <= kLeanMax, 1)) t = Tn * rcp_cr(aden);
        else t = Tn * (1.0f / aden);
        occ = (aden >= gmin) && t < t_self;
        if (aden >= gmin && t == t_self) occ = face_id[k] < self_fid;
    }
    return occ;
}

// -------------------------------------------------------------- BVH node (32 B)
// a = (lo.x, lo.y, lo.z, hi.x), b = (hi.y, hi.z, escape, link).  link >= 0: left child of an
// inner node; link < 0: leaf holding sorted triangle ~link.  On a box hit an inner node
// continues at `link`, otherwise (miss, or leaf) at `escape` (-1 = done) -- no stack.
// Node 0 is the root (see bvh_build.hip).
struct RayBox {
    float ox, oy, oz, ix, iy, iz;
};

__device__ __forceinline__ float safe_inv(float d) {
    // box tests only (conservative); the triangle test always uses the true d
    float ad = fabsf(d);
    float s = ad < 1e-20f ? copysignf(1e-20f, d) : d;
    return 1.0f / s;
}

__device__ __forceinline__ bool box_test(float4 a, float4 b, const RayBox& r, float tmax) {
    float t0x = (a.x - r.ox) * r.ix, t1x = (a.w - r.ox) * r.ix;
    float t0y = (a.y - r.oy) * r.iy, t1y = (b.x - r.oy) * r.iy;
    float t0z = (a.z - r.oz) * r.iz, t1z = (b.y - r.oz) * r.iz;
    float tn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z));
    float tf = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
    // widen the interval: boxes are padded at build time, this covers the slab arithmetic and the error of t itself
    // for triangles the ray meets at a grazing angle (t = T / den is only as good as den: <= ~4e-4 t at the 2^-10
    // cosine the grazing rule admits)
#ifndef NLOS_BOX_EPS
#define NLOS_BOX_EPS 1.6e-3f
#endif
    tn = tn - fabsf(tn) * NLOS_BOX_EPS;
    tf = tf + fabsf(tf) * NLOS_BOX_EPS;
    return fmaxf(tn, 0.0f) <= fminf(tf, tmax);
}

// Is the ray (o, d) blocked before the hit on triangle `self` at distance t_self?
// Equivalent to "closest hit over all faces (ties -> lowest original face id) is
// not `self`", the reference's acceptance rule primID == triangleIndex
// (smoothed_transient/transient_and_gradient.cpp:206), but with early exit.
__device__ __forceinline__ bool occluded(const float4* __restrict__ nodes, int n_nodes,
                                         const float4* __restrict__ tris,
                                         const int* __restrict__ face_id,
                                         V3 o, V3 d, float t_self, int self, int self_fid) {
    RayBox rb;
    rb.ox = o.x; rb.oy = o.y; rb.oz = o.z;
    rb.ix = safe_inv(d.x); rb.iy = safe_inv(d.y); rb.iz = safe_inv(d.z);
    int i = 0;
    while (i >= 0) {
        int leaf = -1;
        while (i >= 0) {
            float4 a = nodes[2 * i], b = nodes[2 * i + 1];
            bool hit = box_test(a, b, rb, t_self);
            int esc = __float_as_int(b.z);
            int link = __float_as_int(b.w);
            if (hit && link < 0) { leaf = ~link; i = esc; break; }
            i = hit ? link : esc;
        }
        if (leaf >= 0 && leaf != self) {
            Tri tr = load_tri(tris, leaf);
            float t, u, v;
            if (tri_test(tr, o, d, t, u, v)) {
                if (t < t_self || (t == t_self && face_id[leaf] < self_fid)) return true;
            }
        }
    }
    return false;
}

// Closest hit (row E).  Returns the sorted triangle index or -1.
__device__ __forceinline__ int closest_hit(const float4* __restrict__ nodes, int n_nodes,
                                           const float4* __restrict__ tris,
                                           const int* __restrict__ face_id,
                                           V3 o, V3 d, float& bt, float& bu, float& bv) {
    RayBox rb;
    rb.ox = o.x; rb.oy = o.y; rb.oz = o.z;
    rb.ix = safe_inv(d.x); rb.iy = safe_inv(d.y); rb.iz = safe_inv(d.z);
    int best = -1, best_fid = 0x7fffffff;
    bt = __int_as_float(0x7f800000);
    // row E takes the caller's direction as it is (embree_intersector/c_embree_intersector.cpp:20-45 hands it to Embree
    // unnormalised): |ng . d| scales with |d|, so the grazing bound does too -- the cut-off angle must not depend on
    // the direction's length
    const float dl = sqrtf(dot(d, d));
    int i = 0;
    while (i >= 0) {
        float4 a = nodes[2 * i], b = nodes[2 * i + 1];
        bool hit = box_test(a, b, rb, bt);
        int esc = __float_as_int(b.z);
        int link = __float_as_int(b.w);
        int tri = ~link;
        if (hit && link < 0) {
            Tri tr = load_tri(tris, tri);
            tr.gmin = tr.gmin * dl;
            float t, u, v;
            if (tri_test(tr, o, d, t, u, v)) {
                int fid = face_id[tri];
                if (best < 0 || t < bt || (t == bt && fid < best_fid)) {
                    best = tri; best_fid = fid; bt = t; bu = u; bv = v;
                }
            }
        }
        i = (hit && link >= 0) ? link : esc;
    }
    return best;
}

// ---------------------------------------------------------------- GGX (row B)
// ggx/ggx_confocal.cpp:13-232 -- scalar functions of nw = dot(normal, w).
// M_PI terms evaluate in double exactly as the reference's expressions do.
#define NLOS_PI 3.14159265358979323846
__device__ __forceinline__ float ggx_D(float a, float nw) {
    if (nw <= 0) return 0.0f;
    float nw2 = nw * nw;
    float bex = (1.0f - nw2) / (a * a) / nw2;
    float root = (1.0f + bex) * nw2;
    float result = (float)(1.0f / (NLOS_PI * a * a * root * root));
    if (result * nw < 1e-20f) result = 0;
    return result;
}
__device__ __forceinline__ float ggx_G1(float a, float nw) {
    if (nw <= 0) return 0.0f;
    if ((nw >= 1.0f) || (nw <= -1.0f)) return 1.0f;
    float root = a * a + (1.0f - a * a) * nw * nw;
    return 2.0f / (nw + sqrtf(root));
}
__device__ __forceinline__ float ggx_G(float a, float nw) { float g = ggx_G1(a, nw); return g * g; }
__device__ __forceinline__ float ggx_eval(float a, float nw) {
    if (nw <= 0) return 0.0f;
    float Dv = ggx_D(a, nw);
    if (Dv == 0) return 0.0f;
    return Dv * ggx_G(a, nw) / 4.0f;
}
__device__ __forceinline__ float ggx_D_adiff(float a, float nw) {
    if (nw <= 0) return 0.0f;
    float nw2 = nw * nw, a2 = a * a;
    float val = a2 * nw2 - nw2 + 1;
    return (float)(-(2.0f * a * (a2 * nw2 + nw2 - 1)) / (NLOS_PI * val * val * val));
}
__device__ __forceinline__ float ggx_G1_adiff(float a, float nw) {
    if (nw <= 0) return 0.0f;
    if ((nw >= 1.0f) || (nw <= -1.0f)) return 0.0f;
    float nw2 = nw * nw;
    float val = sqrtf(a * a - nw2 * (a * a - 1));
    float root = nw + val;
    return 2.0f * a * (nw2 - 1.0f) / (val * root * root);
}
__device__ __forceinline__ float ggx_eval_adiff(float a, float nw) {
    if (nw <= 0) return 0.0f;
    float Dv = ggx_D(a, nw);
    if (Dv == 0) return 0.0f;
    float Gv = ggx_G(a, nw);
    float Dp = ggx_D_adiff(a, nw);
    float Gp = 2.0f * ggx_G1_adiff(a, nw) * ggx_G1(a, nw);
    return (Dp * Gv + Gp * Dv) / 4.0f;
}
__device__ __forceinline__ float ggx_D_ndiff(float a, float nw) {
    if (nw <= 0) return 0.0f;
    float nw2 = nw * nw, a2 = a * a;
    float root = (a2 - 1.0f) * nw2 + 1.0f;
    return (float)(-(4.0f * a2 * nw * (a2 - 1.0f)) / (NLOS_PI * root * root * root));
}
__device__ __forceinline__ float ggx_G1_ndiff(float a, float nw) {
    if (nw <= 0) return 0.0f;
    if ((nw >= 1.0f) || (nw <= -1.0f)) return 0.0f;
    float nw2 = nw * nw, a2 = a * a;
    float temp = sqrtf(a2 - nw2 * (a2 - 1.0f));
    float root = nw + temp;
    return -2.0f * (1.0f - (nw * (a2 - 1.0f)) / temp) / root / root;
}
// scalar s of eval_nwdiff: dnormal = s*w, dw = s*normal (0 on the early-outs)
__device__ __forceinline__ float ggx_eval_nwsdiff(float a, float nw) {
    if (nw <= 0) return 0.0f;
    float Dv = ggx_D(a, nw);
    if (Dv == 0) return 0.0f;
    float Gv = ggx_G(a, nw);
    float Gp = 2.0f * ggx_G1_ndiff(a, nw) * ggx_G1(a, nw);
    float Dp = ggx_D_ndiff(a, nw);
    return (Dp * Gv + Gp * Dv) / 4.0f;
}


// Pass 2's forms of the same functions (round 6).  Pass 2 decides nothing -- the sample was accepted by pass 1 and its bins come from
// pass 1's h -- so its gradient VECTORS may use the 1-ulp reciprocal and square root and single precision throughout (DESIGN.md
// section 2, as the Lambertian branch has since round 3): ggx_D() above divides in DOUBLE (NLOS_PI is one) and the *_ndiff
// functions hold three IEEE divisions and a square root each -- about 250 of the GGX sample loop's instructions.  The values move by
// ~1e-7 relative, the gradient tolerance is 1e-4.  Early-outs and the 1e-20 cut as above.
__device__ __forceinline__ float ggx_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float ggx_D_fast(float a, float nw) {
    if (nw <= 0) return 0.0f;
    const float nw2 = nw * nw, a2 = a * a;
    const float root = nw2 + (1.0f - nw2) * ggx_rcp(a2);            // (1 + (1 - nw2) / a2 / nw2) nw2
    float result = ggx_rcp(3.14159265358979323846f * a2 * root * root);
    if (result * nw < 1e-20f) result = 0;
    return result;
}
__device__ __forceinline__ float ggx_G1_fast(float a, float nw, float& temp_out) {
    temp_out = __builtin_amdgcn_sqrtf(a * a + (1.0f - a * a) * nw * nw);
    if (nw <= 0) return 0.0f;
    if ((nw >= 1.0f) || (nw <= -1.0f)) return 1.0f;
    return 2.0f * ggx_rcp(nw + temp_out);
}
// brdf = ggx_eval, s = ggx_eval_nwsdiff, sharing D, G1 and the square root
__device__ __forceinline__ void ggx_eval_and_nwsdiff_fast(float a, float nw, float& brdf, float& s) {
    brdf = 0.0f; s = 0.0f;
    if (nw <= 0) return;
    const float Dv = ggx_D_fast(a, nw);
    if (Dv == 0) return;
    float temp;
    const float G1 = ggx_G1_fast(a, nw, temp);
    const float Gv = G1 * G1;
    brdf = Dv * Gv * 0.25f;
    const float a2 = a * a;
    float G1n = 0.0f;
    if (!((nw >= 1.0f) || (nw <= -1.0f))) {
        const float ir = ggx_rcp(nw + temp);
        G1n = -2.0f * (1.0f - (nw * (a2 - 1.0f)) * ggx_rcp(temp)) * (ir * ir);
    }
    const float root = (a2 - 1.0f) * nw * nw + 1.0f;
    const float Dp = -(4.0f * a2 * nw * (a2 - 1.0f)) * ggx_rcp(3.14159265358979323846f * root * root * root);
    s = (Dp * Gv + (2.0f * G1n * G1) * Dv) * 0.25f;
}

// d brdf / d alpha (scalar alpha gradient of pass 2), the same way
__device__ __forceinline__ float ggx_eval_adiff_fast(float a, float nw) {
    if (nw <= 0) return 0.0f;
    const float Dv = ggx_D_fast(a, nw);
    if (Dv == 0) return 0.0f;
    float temp;
    const float G1 = ggx_G1_fast(a, nw, temp);
    const float Gv = G1 * G1;
    const float nw2 = nw * nw, a2 = a * a;
    const float val = a2 * nw2 - nw2 + 1;
    const float Dp = -(2.0f * a * (a2 * nw2 + nw2 - 1)) * ggx_rcp(3.14159265358979323846f * val * val * val);
    float G1a = 0.0f;
    if (!((nw >= 1.0f) || (nw <= -1.0f))) {
        const float root = nw + temp;
        G1a = 2.0f * a * (nw2 - 1.0f) * ggx_rcp(temp * root * root);
    }
    return (Dp * Gv + (2.0f * G1a * G1) * Dv) * 0.25f;
}

// GGX for a (laser, sensor) pair (row N with the GGX branch; the reference has neither a kernel nor a prototype):
// the half-vector form of the same microfacet model,
//     brdf(n, wa, wb) = D(n.h) G1(n.wa) G1(n.wb) / 4,   h = (wa + wb) / |wa + wb|,
// with ggx_confocal.cpp's early-outs; it is ggx_eval() for wa == wb.  GRAD: derivatives with respect to wa, wb
// (unconstrained) and n from D' = dD/d(n.h), G1' = dG1/d(n.w) (ggx_confocal.cpp:113-150, :176-232).
struct GgxPair { float brdf; V3 ga, gb, gn; };
// FAST (pass 2's gradient vectors only): the non-decision forms above
template <bool GRAD, bool FAST = false>
__device__ __forceinline__ GgxPair ggx_pair(float a, V3 n, V3 wa, V3 wb) {
    if constexpr (FAST) {
        GgxPair o;
        o.brdf = 0.0f;
        o.ga = o.gb = o.gn = mk(0.0f, 0.0f, 0.0f);
        const float na = dot(n, wa), nb = dot(n, wb);
        if (na <= 0 || nb <= 0) return o;
        const V3 hv = wa + wb;
        const float h2 = dot(hv, hv);
        if (!(h2 > 0.0f)) return o;
        const float ihl = __builtin_amdgcn_rsqf(h2);
        const V3 hn = hv * ihl;
        const float nh = dot(n, hn);
        if (nh <= 0) return o;
        const float Dv = ggx_D_fast(a, nh);
        if (Dv == 0) return o;
        float ta, tb;
        const float Ga = ggx_G1_fast(a, na, ta), Gb = ggx_G1_fast(a, nb, tb);
        o.brdf = Dv * Ga * Gb * 0.25f;
        if (GRAD) {
            const float a2 = a * a;
            auto g1n = [&](float nw, float temp) -> float {
                if ((nw >= 1.0f) || (nw <= -1.0f)) return 0.0f;
                const float ir = ggx_rcp(nw + temp);
                return -2.0f * (1.0f - (nw * (a2 - 1.0f)) * ggx_rcp(temp)) * (ir * ir);
            };
            const float root = (a2 - 1.0f) * nh * nh + 1.0f;
            const float Dn = -(4.0f * a2 * nh * (a2 - 1.0f)) * ggx_rcp(3.14159265358979323846f * root * root * root);
            const float cD = Dn * Ga * Gb * 0.25f;
            const float cA = Dv * g1n(na, ta) * Gb * 0.25f;
            const float cB = Dv * Ga * g1n(nb, tb) * 0.25f;
            const V3 dnh = (n - hn * nh) * ihl;
            o.ga = (dnh * cD) + (n * cA);
            o.gb = (dnh * cD) + (n * cB);
            o.gn = ((hn * cD) + (wa * cA)) + (wb * cB);
        }
        return o;
    }
    GgxPair o;
    o.brdf = 0.0f;
    o.ga = o.gb = o.gn = mk(0.0f, 0.0f, 0.0f);
    const float na = dot(n, wa), nb = dot(n, wb);
    if (na <= 0 || nb <= 0) return o;
    const V3 hv = wa + wb;
    const float hl = sqrtf(dot(hv, hv));
    if (!(hl > 0.0f)) return o;
    const V3 hn = hv * (1.0f / hl);
    const float nh = dot(n, hn);
    if (nh <= 0) return o;
    const float Dv = ggx_D(a, nh);
    if (Dv == 0) return o;
    const float Ga = ggx_G1(a, na), Gb = ggx_G1(a, nb);
    o.brdf = Dv * Ga * Gb / 4.0f;
    if (GRAD) {
        const float cD = ggx_D_ndiff(a, nh) * Ga * Gb / 4.0f;
        const float cA = Dv * ggx_G1_ndiff(a, na) * Gb / 4.0f;
        const float cB = Dv * Ga * ggx_G1_ndiff(a, nb) / 4.0f;
        const V3 dnh = (n - hn * nh) * (1.0f / hl);          // d(n.h)/dwa = d(n.h)/dwb
        o.ga = (dnh * cD) + (n * cA);
        o.gb = (dnh * cD) + (n * cB);
        o.gn = ((hn * cD) + (wa * cA)) + (wb * cB);
    }
    return o;
}

}  // namespace nlos
