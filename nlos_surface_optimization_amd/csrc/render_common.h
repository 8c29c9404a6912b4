// render_common.h -- per-sample building blocks shared by the render kernels (internal header).
//
// Everything a (source, face, stratum) sample needs, restated once from the reference
// (paths relative to transient_rendering_cython/):
//   load_face / sample_geo     smoothed_transient/transient_and_gradient.cpp:157-159, :178-223 (rows S, I own-face part)
//   sample_geo_nc              row N (non-confocal pairs; prototypes only in the reference)
//   grad_vectors(_nc)          :944-966, ggx/transient_and_gradient.cpp:750-783 (t1, t2, intensity)
//   tap_bin / grouped_taps     :973-999 (the K-tap loop, factored into two scalar sums)
// The kernels live in forward_bvh.hip, forward_grid.hip, gradient.hip and render_kernels.hip.
#pragma once
#include "nlos_device.h"
#include "nlos_kernels.h"

#include <algorithm>
#include <cmath>

namespace nlos {
namespace {

constexpr int FEAT_VN = 1, FEAT_ALB = 2, FEAT_GGX = 4;

struct Face {
    V3 p0, p1, p2;
    int fid, i0, i1, i2;
    V3 fn;
    float area, inv2a;       // inv2a = 1 / (2 area), the build's value
    bool degenerate;
};

__device__ __forceinline__ Face load_face(const float4* __restrict__ rec, int j) {
    float4 a = rec[4 * j], b = rec[4 * j + 1], c = rec[4 * j + 2], d = rec[4 * j + 3];
    Face f;
    f.p0 = mk(a.x, a.y, a.z);
    f.p1 = mk(a.w, b.x, b.y);
    f.p2 = mk(b.z, b.w, c.x);
    f.fid = __float_as_int(c.y);
    f.i0 = __float_as_int(c.z);
    f.i1 = __float_as_int(c.w);
    f.i2 = __float_as_int(d.x);
    // smoothed_transient/transient_and_gradient.cpp:157-159
    V3 nr = cross(f.p1 - f.p0, f.p2 - f.p0);
    f.area = sqrtf(dot(nr, nr)) / 2.0f;
    f.degenerate = !(f.area > 0.0f);
    f.inv2a = 1.0f / (2.0f * f.area);
    f.fn = nr * f.inv2a;
    return f;
}

// Face + triangle of sorted slot j for the sample map.  Same values as load_face() / load_tri() (render_common.h,
// nlos_device.h): e1, e2 are make_tri()'s own subtractions, ng == cross(p1 - p0, p2 - p0) bit for bit, and area,
// 1 / (2 area) were evaluated by the scene build with load_face()'s expressions -- once per step instead of once
// per (source, face).
template <int FEAT>
__device__ __forceinline__ void load_face_tri(const SceneView& sc, int j, Face& f, Tri& tr) {
    const float4 fa = sc.facerec[4 * j], fb = sc.facerec[4 * j + 1], fc = sc.facerec[4 * j + 2];
    const float4 tc = sc.tris[kTriStride * j + 2], td = sc.tris[kTriStride * j + 3];
    f.p0 = mk(fa.x, fa.y, fa.z); f.p1 = mk(fa.w, fb.x, fb.y); f.p2 = mk(fb.z, fb.w, fc.x);
    f.fid = __float_as_int(fc.y);
    f.i0 = __float_as_int(fc.z); f.i1 = __float_as_int(fc.w);
    f.i2 = (FEAT & (FEAT_VN | FEAT_ALB)) ? __float_as_int(sc.facerec[4 * j + 3].x) : 0;
    tr.p0 = f.p0; tr.e1 = f.p0 - f.p1; tr.e2 = f.p2 - f.p0;
    tr.ng = mk(tc.y, tc.z, tc.w);
    tr.gmin = kGrazeRatio * td.z;
    f.area = td.z;
    f.degenerate = !(f.area > 0.0f);
    f.inv2a = td.w;
    f.fn = tr.ng * td.w;
}

// per-sample geometry of an own-face hit
struct Geo {
    float u, v, w, h;
    V3 dir, n;
    float alb;
};

// Row S + the own-face part of row I: stratified sample -> ray -> hit on face j.
// Returns false if the ray misses its own triangle (edge rounding) or the path
// length is outside [lb/2, ub/2].
// LEAN (the grid kernel's trace: every vertex of the scene lies between 2^-29 and 2^28 in front of / around the wall
// point, forward_grid.hip: source_frame()): the square roots and reciprocals as sqrt_cr() / rcp_cr() (nlos_device.h) --
// the same bits for 40 % of the instructions; the draw comes keyed as (zbase, c), see sample_st_c().
template <int FEAT, bool LEAN = false>
__device__ __forceinline__ bool sample_geo_st(const Face& f, const Tri& tr, V3 o, float S, float T,
                                              float lb, float ub, const float* __restrict__ vn,
                                              const float* __restrict__ alb, Geo& g, float& t_self) {
    float sq = LEAN ? sqrt_cr0(T) : sqrtf(T);
    float u = 1 - sq;
    float v = (1 - S) * sq;
    float w = S * sq;
    V3 p = bary(u, f.p0, v, f.p1, w, f.p2);
    V3 d = p - o;
    float rs = LEAN ? rcp_cr(sqrt_cr(dot(d, d))) : 1.0f / sqrtf(dot(d, d));
    g.dir = d * rs;
    float hu, hv;
    if (!tri_test<LEAN>(tr, o, g.dir, t_self, hu, hv)) return false;
    g.v = hu;
    g.w = hv;
    g.u = 1.0f - g.v - g.w;
    V3 q = bary(g.u, f.p0, g.v, f.p1, g.w, f.p2);
    V3 dq = q - o;
    g.h = LEAN ? sqrt_cr(dot(dq, dq)) : sqrtf(dot(dq, dq));
    if (!((g.h <= ub / 2.0f) && (g.h >= lb / 2.0f))) return false;
    g.n = f.fn;
    if (FEAT & FEAT_VN) {
        g.n = bary(g.u, ld3(vn + 3 * (size_t)f.i0), g.v, ld3(vn + 3 * (size_t)f.i1), g.w,
                   ld3(vn + 3 * (size_t)f.i2));
    }
    g.alb = 1.0f;
    if (FEAT & FEAT_ALB) g.alb = g.u * alb[f.i0] + g.v * alb[f.i1] + g.w * alb[f.i2];
    return true;
}
template <int FEAT>
__device__ __forceinline__ bool sample_geo(const Face& f, const Tri& tr, V3 o, uint64_t seed, uint64_t k,
                                           float lb, float ub, const float* __restrict__ vn,
                                           const float* __restrict__ alb, Geo& g, float& t_self) {
    float S, T;
    sample_st(seed, k, S, T);
    return sample_geo_st<FEAT, false>(f, tr, o, S, T, lb, ub, vn, alb, g, t_self);
}

// The same Geo from pass 1's geometry cache: the ray's direction, the hit's barycentrics (v, w) and h, bit for bit what
// sample_geo() computed.  (A 12-byte record of h, v, w with the direction taken as the unit vector towards the hit point
// was built first: for rays that graze their face the hit's barycentrics -- and with them that direction -- carry the
// 1 / |cos| error of the triangle test, and the gradient moved by up to 7e-5 of its norm in the sweep; pass 2 must see the
// SAMPLED direction, as the reference's does.)
template <int FEAT>
__device__ __forceinline__ void cached_geo(const Face& f, V3 dir, float hv, float hw, float h, const float* __restrict__ vn,
                                           const float* __restrict__ alb, Geo& g) {
    g.v = hv;
    g.w = hw;
    g.u = 1.0f - g.v - g.w;
    g.h = h;
    g.dir = dir;
    g.n = f.fn;
    if (FEAT & FEAT_VN) {
        g.n = bary(g.u, ld3(vn + 3 * (size_t)f.i0), g.v, ld3(vn + 3 * (size_t)f.i1), g.w,
                   ld3(vn + 3 * (size_t)f.i2));
    }
    g.alb = 1.0f;
    if (FEAT & FEAT_ALB) g.alb = g.u * alb[f.i0] + g.v * alb[f.i1] + g.w * alb[f.i2];
}

// Slope-space frame of a source: bounding rectangle of the projection of the BVH's (padded) root box.
// Shared by the grid kernel and the tile-binning kernel, which must agree bit for bit.
struct SourceFrame { bool ok; float gx0, gy0, wx, wy, zr0, zr1, hmin; };
__device__ __forceinline__ SourceFrame source_frame(const float4* __restrict__ nodes, V3 o) {
    SourceFrame fr;
    const float4 ra = nodes[0], rb = nodes[1];
    fr.zr0 = ra.z - o.z;
    fr.zr1 = rb.y - o.z;
    const float ext = fmaxf(fmaxf(ra.w - ra.x, rb.x - ra.y), rb.y - ra.z);
    fr.ok = fr.zr0 > 0.02f * ext && fr.zr0 > 0.0f;
    const float i0 = 1.0f / fmaxf(fr.zr0, 1e-30f), i1 = 1.0f / fmaxf(fr.zr1, 1e-30f);
    const float xl = ra.x - o.x, xh = ra.w - o.x, yl = ra.y - o.y, yh = rb.x - o.y;
    const float gx0 = fminf(xl * i0, xl * i1), gx1 = fmaxf(xh * i0, xh * i1);
    const float gy0 = fminf(yl * i0, yl * i1), gy1 = fmaxf(yh * i0, yh * i1);
    fr.wx = fmaxf(gx1 - gx0, 1e-12f);
    fr.wy = fmaxf(gy1 - gy0, 1e-12f);
    fr.gx0 = gx0 - 1e-3f * fr.wx;
    fr.gy0 = gy0 - 1e-3f * fr.wy;
    // graze_scale(): plane distance below which a triangle of this scene can be seen at |cos| < 2^-6 (x 1.1)
    const float fx = fmaxf(fabsf(xl), fabsf(xh)), fy = fmaxf(fabsf(yl), fabsf(yh)), fz = fmaxf(fabsf(fr.zr0), fabsf(fr.zr1));
    fr.hmin = sqrtf(fx * fx + fy * fy + fz * fz) * (1.1f / 64.0f);
    // The grid trace evaluates its square roots, reciprocals and divisions in the lean forms of nlos_device.h, which are the
    // IEEE results for operands between 2^-60 and 2^60: every vertex at least 2^-29 in front of the wall point and at most
    // 2^28 away along every axis keeps |p - o|^2, |p - o|, the edge cross products and their reciprocals inside that range.
    // (A scene outside it -- nanometres or light-seconds in a renderer of metre-sized objects -- takes the BVH query.)
    fr.ok = fr.ok && fr.zr0 >= 0x1p-29f && fmaxf(fmaxf(fx, fy), fz) <= 0x1p28f;
    return fr;
}

// The grid trace divides by `res` in the lean form (nlos_device.h: div_by), exact for 2^-30 <= res <= 2^30 and path
// lengths below 2^29; a time window outside that range is rendered through the BVH back-end (reason 7).
inline bool lean_params_ok(const SampleParams& sp) {
    return sp.res >= 0x1p-30f && sp.res <= 0x1p30f && fabsf(sp.lb) <= 0x1p29f && fabsf(sp.ub) <= 0x1p29f;
}

// num / h / h, the (unclamped) form factor of a leg of length h.  LEAN: through one refined reciprocal of h where the numerator
// lies in the range on which div_by() is the IEEE division (nlos_device.h); the rare lane outside takes the IEEE form.  Same bits.
template <bool LEAN>
__device__ __forceinline__ float form_factor(float num, float h) {
    if (LEAN && __builtin_expect(fabsf(num) >= kLeanMin && fabsf(num) <= kLeanNumMax, 1)) {
        const float rh = rcp_refined(h);
        return div_by(div_by(num, h, rh), h, rh);
    }
    return num / h / h;
}

// Pass 2's regenerated samples (no geometry cache: pairs excluded, spt > 8, jitter, scalar modes, per-face-word layouts) in the
// lean forms as well, where the source's frame and the launch's window guarantee the operand range: the same bits as pass 1's,
// whichever form pass 1 itself used.  `lean` is wave-uniform (a property of the source).
template <int FEAT>
__device__ __forceinline__ bool sample_geo_keyed(const Face& f, const Tri& tr, V3 o, uint64_t zbase, uint32_t c, bool lean,
                                                 float lb, float ub, const float* __restrict__ vn,
                                                 const float* __restrict__ alb, Geo& g, float& t_self) {
    float S, T;
    sample_st_c(zbase, c, S, T);
    return lean ? sample_geo_st<FEAT, true>(f, tr, o, S, T, lb, ub, vn, alb, g, t_self)
                : sample_geo_st<FEAT, false>(f, tr, o, S, T, lb, ub, vn, alb, g, t_self);
}

__device__ __forceinline__ float emax0(float x) { return 0.0f < x ? x : 0.0f; }

// fp64 add into a row held in LDS, as ds_add_f64 whatever the optimiser thinks of the surrounding branches.  Written as
// unsafeAtomicAdd(&s_row[bin], x) next to an unsafeAtomicAdd(&grow[bin], x) for the rows kept in HBM, the two calls are sunk
// into ONE flat_atomic_add_f64 on a generic pointer (src_shared_base for the LDS case).  A FLAT operation counts in vmcnt
// as well as lgkmcnt, which looked like the store stall of DESIGN.md 4.2 over again; measured, it is not (1.353 vs 1.349 ms,
// profiles/r03_ab_flat.log) -- the LDS instruction is kept because it is what the code says.
__device__ __forceinline__ void lds_add_f64(double* lds_ptr, double x) {
#ifdef NLOS_DIAG_FLAT_ROW_ADD      // diagnostic builds only: the form the optimiser merges into a FLAT atomic
    unsafeAtomicAdd(lds_ptr, x);
#else
    typedef __attribute__((address_space(3))) double lds_double;
    __builtin_amdgcn_ds_atomic_fadd_f64((lds_double*)lds_ptr, x);
#endif
}

__device__ __forceinline__ int wave_ticket(int* counter) {
    int b = 0;
    if ((threadIdx.x & 63) == 0) b = atomicAdd(counter, 1);
    return __builtin_amdgcn_readfirstlane(b);
}

// ------------------------------------------------------- forward, row N (pairs)
// Non-confocal pair l = (laser a, sensor b).  No native reference kernel exists;
// geometry follows the reference prototypes (transient_rendering_python/rendering.py:37-93: nearest
// hit from the laser, second segment visible from the sensor, path d1 + d2) with the v2
// conventions of rows F/G, so a == b gives the confocal rows bit for bit.
struct GeoNC {
    float u, v, w, d1, d2;
    V3 dirA, dirB, n;
    float alb;
};

// LEAN: the lean forms of sqrt / reciprocal (nlos_device.h) -- the same bits where BOTH wall points' frames guarantee the
// operand range (source_frame() of the laser and of the sensor; the callers check both).
template <int FEAT, bool LEAN = false>
__device__ __forceinline__ bool sample_geo_nc(const Face& f, const Tri& tr, V3 oa, V3 ob, uint64_t seed, uint64_t k,
                                              float lb, float ub, const float* __restrict__ vn,
                                              const float* __restrict__ alb, GeoNC& g, float& tA, float& tB) {
    float S, T;
    sample_st(seed, k, S, T);
    float sq = LEAN ? sqrt_cr0(T) : sqrtf(T);
    float u = 1 - sq;
    float v = (1 - S) * sq;
    float w = S * sq;
    V3 p = bary(u, f.p0, v, f.p1, w, f.p2);
    V3 dA = p - oa;
    g.dirA = dA * (LEAN ? rcp_cr(sqrt_cr(dot(dA, dA))) : 1.0f / sqrtf(dot(dA, dA)));
    float hu, hv;
    if (!tri_test<LEAN>(tr, oa, g.dirA, tA, hu, hv)) return false;
    g.v = hu;
    g.w = hv;
    g.u = 1.0f - g.v - g.w;
    V3 qA = bary(g.u, f.p0, g.v, f.p1, g.w, f.p2);
    V3 eA = qA - oa;
    g.d1 = LEAN ? sqrt_cr(dot(eA, eA)) : sqrtf(dot(eA, eA));
    V3 dB = p - ob;
    g.dirB = dB * (LEAN ? rcp_cr(sqrt_cr(dot(dB, dB))) : 1.0f / sqrtf(dot(dB, dB)));
    float bu, bv;
    if (!tri_test<LEAN>(tr, ob, g.dirB, tB, bu, bv)) return false;
    V3 qB = bary(1.0f - bu - bv, f.p0, bu, f.p1, bv, f.p2);
    V3 eB = qB - ob;
    g.d2 = LEAN ? sqrt_cr(dot(eB, eB)) : sqrtf(dot(eB, eB));
    const float tot = g.d1 + g.d2;
    if (!((tot <= ub) && (tot >= lb))) return false;
    g.n = f.fn;
    if (FEAT & FEAT_VN) {
        g.n = bary(g.u, ld3(vn + 3 * (size_t)f.i0), g.v, ld3(vn + 3 * (size_t)f.i1), g.w,
                   ld3(vn + 3 * (size_t)f.i2));
    }
    g.alb = 1.0f;
    if (FEAT & FEAT_ALB) g.alb = g.u * alb[f.i0] + g.v * alb[f.i1] + g.w * alb[f.i2];
    return true;
}
// (wave-uniform choice of the form: a property of the pair)
template <int FEAT>
__device__ __forceinline__ bool sample_geo_nc_rt(const Face& f, const Tri& tr, V3 oa, V3 ob, uint64_t seed, uint64_t k, bool lean,
                                                 float lb, float ub, const float* __restrict__ vn,
                                                 const float* __restrict__ alb, GeoNC& g, float& tA, float& tB) {
    return lean ? sample_geo_nc<FEAT, true>(f, tr, oa, ob, seed, k, lb, ub, vn, alb, g, tA, tB)
                : sample_geo_nc<FEAT, false>(f, tr, oa, ob, seed, k, lb, ub, vn, alb, g, tA, tB);
}

// ------------------------------------------------------------------- gradient
// Per accepted sample: vectors t1, t2 and the intensity
// (smoothed_transient/transient_and_gradient.cpp:944-966, ggx/...:750-783).
struct GVec { V3 t1, t2; float inten_f; };

// Pass 2 decides nothing: the sample was accepted by pass 1 and its bins come from sample_geo()'s h, which stays the
// contract's arithmetic.  The gradient vectors themselves may therefore use the 1-ulp reciprocal (v_rcp_f32) instead of
// the correctly rounded division sequence (~10 instructions each) and fused multiply-adds: the result moves by ~1e-7
// relative per sample, against a gradient tolerance of 1e-4 and the ~4e-7 the factored tap loop already differs by.
__device__ __forceinline__ float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ V3 grad_axpy(V3 a, float s, V3 b) {
    return mk(__fmaf_rn(a.x, s, b.x), __fmaf_rn(a.y, s, b.y), __fmaf_rn(a.z, s, b.z));
}
__device__ __forceinline__ V3 fmadd(V3 a, float s, V3 b) {       // a * s + b
    return mk(__fmaf_rn(a.x, s, b.x), __fmaf_rn(a.y, s, b.y), __fmaf_rn(a.z, s, b.z));
}

template <int FEAT>
__device__ __forceinline__ void grad_vectors(const Face& f, const Geo& g, V3 on, int normal_term, int v1_style,
                                             float alpha, GVec& out) {
    float c2 = dot(on, g.dir);
    float c3 = dot(g.n, -g.dir);
    if (c2 < 0) c2 = 0;
    if (c3 < 0) c3 = 0;
    const float h2 = g.h * g.h;
    V3 t1, gn = mk(0, 0, 0);
    if (FEAT & FEAT_GGX) {
#ifdef NLOS_DIAG_GGX_IEEE            // diagnostic builds only: the contract's divisions (rounds 1 - 5)
        const float ff = c2 * c3 / g.h / g.h;
        const float h4 = h2 * h2, h5 = h4 * g.h;
        const float ih = 1.0f / g.h, ih5 = 1.0f / h5, ih4 = 1.0f / h4;
        V3 wv = -g.dir;
        float nw = dot(g.n, wv);
        float brdf = ggx_eval(alpha, nw);
        float s = ggx_eval_nwsdiff(alpha, nw);
#else
        // (round 6: non-decision arithmetic like the Lambertian branch -- 1-ulp reciprocals, the BRDF and its derivative from
        // one evaluation in single precision, nlos_device.h: ggx_eval_and_nwsdiff_fast)
        const float ih = rcp_fast(g.h), ih2 = ih * ih;
        const float ff = (c2 * c3) * ih2;
        const float ih4 = ih2 * ih2, ih5 = ih4 * ih;
        V3 wv = -g.dir;
        float nw = dot(g.n, wv);
        float brdf, s;
        ggx_eval_and_nwsdiff_fast(alpha, nw, brdf, s);
#endif
        const V3 inner = ((on * c3) - (g.n * c2)) + ((((-g.dir) * 4.0f) * c2) * c3);
        V3 dn = wv * s, dw = g.n * s;
        V3 dx = (-dw) + ((g.dir * dot(g.dir, dw)) * ih);
        out.inten_f = (float)(double)(g.alb * ff * ff * brdf);
        V3 t11 = inner * (2 * c2 * c3);
        t11 = t11 * ih5;
        t11 = t11 * brdf;
        t1 = t11 + dx * (ff * ff);
        if (normal_term) {
            gn = ((((g.dir * -2.0f) * c3) * c2) * c2) * brdf;
            gn = gn * ih4;
            gn = gn + dn * (ff * ff);
            float ct = dot(gn, g.n);
            gn = gn - g.n * ct;
        }
    } else {
        const float ih2 = rcp_fast(h2);
        const float ff = (c2 * c3) * ih2;
        out.inten_f = g.alb * ff * ff;
        const float sc = (v1_style ? (2 * c2 * c3) : (2 * g.alb * c2 * c3)) * (ih2 * ih2) * rcp_fast(g.h);
        t1 = fmadd(on, c3, fmadd(g.n, -c2, g.dir * (-4.0f * c2 * c3))) * sc;
        if (normal_term) {
            const float s0 = (v1_style ? -2.0f : (-2 * g.alb)) * c3 * c2 * c2 * (ih2 * ih2);
            gn = g.dir * s0;
            gn = fmadd(g.n, -dot(gn, g.n), gn);
        }
    }
    out.t1 = t1;
    out.t2 = fmadd(g.n, out.inten_f, gn) * f.inv2a;
}

// Row N: t1 = alb (ff_b grad ff_a + ff_a grad ff_b), grad ff = (n_o c3 - n c2 - 4 dir c2 c3) / d^3;
// normal term dI/dn projected as in the confocal rows
template <int FEAT>
__device__ __forceinline__ void grad_vectors_nc(const Face& f, const GeoNC& g, V3 na, V3 nb, int normal_term, float alpha,
                                                GVec& out) {
    float c2a = dot(na, g.dirA), c3a = dot(g.n, -g.dirA);
    float c2b = dot(nb, g.dirB), c3b = dot(g.n, -g.dirB);
    if (c2a < 0) c2a = 0;
    if (c3a < 0) c3a = 0;
    if (c2b < 0) c2b = 0;
    if (c3b < 0) c3b = 0;
    const float i1 = rcp_fast(g.d1), i2 = rcp_fast(g.d2);            // (non-decision arithmetic, see rcp_fast)
    const float ffa = c2a * c3a * (i1 * i1), ffb = c2b * c3b * (i2 * i2);
    const V3 ia = fmadd(na, c3a, fmadd(g.n, -c2a, g.dirA * (-4.0f * c2a * c3a)));
    const V3 ib = fmadd(nb, c3b, fmadd(g.n, -c2b, g.dirB * (-4.0f * c2b * c3b)));
    const V3 ga = ia * ((i1 * i1) * i1);
    const V3 gb = ib * ((i2 * i2) * i2);
    out.inten_f = g.alb * ffa * ffb;
    out.t1 = fmadd(ga, ffb, gb * ffa) * g.alb;
    V3 gn = mk(0, 0, 0);
    if (normal_term) {
        gn = fmadd(g.dirA, c3b, g.dirB * c3a);
        gn = gn * (-(g.alb * c2a * c2b) * ((i1 * i1) * (i2 * i2)));
    }
    if (FEAT & FEAT_GGX) {
        // I = I_lambert * brdf(n, wa, wb), wx = -dirx:  dI/dp = brdf dI_l/dp + I_l (J_a^T ga + J_b^T gb) with
        // J_x = d wx / dp = -(1 - wx wx^T) / d_x;  dI/dn = brdf dI_l/dn + I_l d brdf/dn
        const V3 wa = -g.dirA, wb = -g.dirB;
#ifdef NLOS_DIAG_GGX_IEEE
        const GgxPair gp = ggx_pair<true>(alpha, g.n, wa, wb);
#else
        const GgxPair gp = ggx_pair<true, true>(alpha, g.n, wa, wb);      // (non-decision arithmetic: round 6)
#endif
        const float il = out.inten_f;
        const V3 pa = fmadd(wa, -dot(wa, gp.ga), gp.ga) * (-i1);
        const V3 pb = fmadd(wb, -dot(wb, gp.gb), gp.gb) * (-i2);
        out.t1 = (out.t1 * gp.brdf) + ((pa + pb) * il);
        if (normal_term) gn = (gn * gp.brdf) + (gp.gn * il);
        out.inten_f = il * gp.brdf;
    }
    if (normal_term) gn = fmadd(g.n, -dot(gn, g.n), gn);
    out.t2 = fmadd(g.n, out.inten_f, gn) * f.inv2a;
}

// bin of tap i: floor((2h + delta_i - lb) / res) in double
// (smoothed_transient/transient_and_gradient.cpp:975-976); reciprocal multiply with an
// exact-division fallback when the quotient is within 1e-9 of an integer.
__device__ __forceinline__ int tap_bin(double twoh, double delta, double lb, double res, double inv_res) {
    double num = (twoh + delta) - lb;
    double x = num * inv_res;
    double fl = floor(x);
    double fr = x - fl;
    if (fr < 1e-9 || fr > 1.0 - 1e-9) fl = floor(num / res);
    return (int)fl;
}

// Grouped taps (mode 0).  bin_i is non-decreasing in i and takes at most 4*sigma_bin+2 distinct
// values, so sum_i w_i d[bin_i] = sum_b d[b] * (P0[end_b] - P0[start_b]) with host-side prefix sums
// P0 = cumsum(float(w)), P1 = cumsum(g * float(w)).  The boundary tap of every bin is located from
// the closed form and then verified with the exact per-tap bin formula, so tap->bin assignment is
// identical to the reference's literal loop (the exact check runs only when the closed-form boundary
// falls within 1e-4 of a tap, 100x the rounding of the tap offsets).
struct TapTables {
    const double* delta;   // [K]
    const double* p0;      // [K+1]
    const double* p1;      // [K+1]
    int K, two_rs, refine;
    double r_over_res;     // refine / res
    // Round 5: the bin weights themselves.  With the first boundary at tap ic (1 <= ic <= refine) the taps fall into
    // nb = 4 sigma_bin + 1 bins whose boundaries are ic, ic + refine, ...: the per-bin differences of the prefix sums depend on
    // ic alone, so the host tabulates them -- wt[(ic * nb + k) * 2 + {0, 1}] = P{0,1}[end_k] - P{0,1}[start_k], the very
    // subtractions the loop below performs (same doubles, same bits) -- and a sample pays two loads and two fused
    // multiply-adds per bin instead of the boundary arithmetic (20 of the ~300 VALU instructions of a sample, five times).
    const double* wt;      // or null
    int nb;
};

// s_m2diff: the residual row as (double)(float)(-2 d)
__device__ __forceinline__ void grouped_taps(const TapTables& tt, const double* __restrict__ s_m2diff, int T,
                                             double twoh, double lbd, double resd, double inv_res,
                                             double& s0, double& s1) {
    s0 = 0.0;
    s1 = 0.0;
    const int K = tt.K;
    const int b_first = tap_bin(twoh, tt.delta[0], lbd, resd, inv_res);
    // first tap whose bin exceeds b: delta_i >= (b+1)*res + lb - 2h, i.e. a boundary y (in continuous tap
    // index, minus two_rs); consecutive boundaries are exactly `refine` taps apart
    const double thr0 = ((double)(b_first + 1) * resd + lbd) - twoh;
    const double y0 = thr0 * tt.r_over_res;
    const double yc0 = ceil(y0);
    // delta_i carries the fp32 rounding of (i - two_rs) * res / refine (<= ~1e-6 taps): the closed form is the
    // reference's per-tap assignment unless the boundary is that close to a tap.  All boundaries of a sample
    // share the fractional part of y0 (to ~1e-13), so one test decides for the whole sample.
    const bool clear = yc0 - y0 >= 1e-4 && y0 - (yc0 - 1.0) >= 1e-4;
    if (tt.wt && clear) {
        // tabulated bin weights: with the first boundary at tap ic in [1, refine] the K = 4 refine sigma + 1 taps meet
        // 4 sigma boundaries (ic + k refine <= K - 1 for k <= 4 sigma - 1), i.e. nb = 4 sigma + 1 bins from b_first on --
        // the last tap's bin needs no evaluation of its own
        const int ic = (int)yc0 + tt.two_rs;
        if (ic >= 1 && ic <= tt.refine) {
            const double* w = tt.wt + 2 * (size_t)(ic * tt.nb);
            if (tt.nb == 5 && b_first >= 0 && b_first + 5 <= T) {          // (sigma_bin = 1, the window inside the row: no checks)
                const double* d = s_m2diff + b_first;
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    s0 = fma(d[k], w[2 * k], s0);
                    s1 = fma(d[k], w[2 * k + 1], s1);
                }
                return;
            }
            for (int k = 0; k < tt.nb; ++k) {
                const int b = b_first + k;
                if (b >= 0 && b < T) {
                    const double dd = s_m2diff[b];
                    s0 = fma(dd, w[2 * k], s0);
                    s1 = fma(dd, w[2 * k + 1], s1);
                }
            }
            return;
        }
    }
    const int b_last = tap_bin(twoh, tt.delta[K - 1], lbd, resd, inv_res);
    if (b_first == b_last || clear) {
        int ic = (int)yc0 + tt.two_rs;
        int i_start = 0;
        double q0 = tt.p0[0], q1 = tt.p1[0];
        for (int b = b_first; b <= b_last; ++b) {
            const int ie = b < b_last ? max(i_start, min(K, ic)) : K;
            const double e0 = tt.p0[ie], e1 = tt.p1[ie];
            if (b >= 0 && b < T) {
                const double dd = s_m2diff[b];
                s0 = fma(dd, e0 - q0, s0);       // fp64 sums of fp32-exact products: fused or not is below 1e-16
                s1 = fma(dd, e1 - q1, s1);
            }
            q0 = e0; q1 = e1;
            i_start = ie;
            ic += tt.refine;
        }
        return;
    }
    // a boundary within 1e-4 of a tap: every boundary verified with the exact per-tap bin formula
    int i_start = 0;
    for (int b = b_first; b <= b_last; ++b) {
        int ie = K;
        if (b < b_last) {
            const double thr = ((double)(b + 1) * resd + lbd) - twoh;
            const double y = thr * tt.r_over_res;
            int ic = (int)ceil(y) + tt.two_rs;
            ic = max(i_start, min(K, ic));
            while (ic > i_start && tap_bin(twoh, tt.delta[ic - 1], lbd, resd, inv_res) > b) --ic;
            while (ic < K && tap_bin(twoh, tt.delta[ic], lbd, resd, inv_res) <= b) ++ic;
            ie = ic;
        }
        if (b >= 0 && b < T && ie > i_start) {
            const double dd = s_m2diff[b];
            s0 += dd * (tt.p0[ie] - tt.p0[i_start]);
            s1 += dd * (tt.p1[ie] - tt.p1[i_start]);
        }
        i_start = ie;
    }
}

// MODE 0: per-vertex gradient [V,3]; 1: scalar d/d albedo; 2: scalar d/d alpha (GGX);
//      3: single-vertex per-bin gradient [T,3]
// two 512-thread workgroups per CU (LDS: ~76 KB each) need <= 128 VGPRs
inline int feat_of(const SceneView& sc, const SampleParams& sp) {
    return (sc.vertex_normal ? FEAT_VN : 0) | (sc.albedo ? FEAT_ALB : 0) | (sp.use_ggx ? FEAT_GGX : 0);
}


}  // namespace
}  // namespace nlos
