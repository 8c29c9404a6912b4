// render_kernels.hip -- forward transient, residual and vertex-gradient kernels (gfx950).
//
// HIP counterparts of the reference's per-(source, triangle) task functions
// (paths relative to transient_rendering_cython/):
//   k_forward   <- streamedRayTraceTriangle / streamedRayTraceIntensity
//                  (smoothed_transient/transient_and_gradient.cpp:122-237, :22-119)
//                  + thread reduction of render_smoothed_transients (:320-341)
//   k_smooth    <- refined-histogram Gaussian + fold (:348-371)
//   k_residual  <- difference = (data - transient)[^3*2] * weight
//                  (smoothed_transient/stratifiedStreamedGradientRenderer.cpp:543-550)
//   k_boxfilter <- v1 residual smoothing
//                  (stratified_transient_raytracer/stratifiedStreamedGradientRenderer.cpp:447-462)
//   k_gradient  <- streamedRayTraceTriangleGradient / ...GradientAlbedo /
//                  ...GradientAlpha / ...VertexGradient (:843-1007, :571-695,
//                  ggx/transient_and_gradient.cpp:385-512, :697-840) + reduction (:561-565)
//
// Mapping to the hardware (MI355X, wave64):
//   * one workgroup per source: every ray of the workgroup starts at the same wall
//     point, the histogram row of that source lives in LDS (ds_add_f64), and is
//     written back once with coalesced stores -- no global atomics in pass 1;
//   * one lane per face (Morton order), looping over that face's `spt` strata:
//     neighbouring lanes shoot at neighbouring faces -> coherent BVH traversal, and
//     the nine per-vertex gradient sums of a (source, face) pair are reduced in
//     registers before touching LDS;
//   * waves pull 64-face blocks from an LDS ticket counter (back-facing blocks cost
//     almost nothing, so static striping would leave waves idle);
//   * the visibility of every accepted sample is cached as one bit by pass 1; pass 2
//     never traces a ray (the reference traces all rays twice);
//   * samples whose clamped form factor is zero contribute exactly 0 to both passes,
//     so their rays are never traced at all.
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
    float area;
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
    f.fn = nr * (1.0f / (2.0f * f.area));
    return f;
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
template <int FEAT>
__device__ __forceinline__ bool sample_geo(const Face& f, const Tri& tr, V3 o, uint64_t seed, uint64_t k,
                                           float lb, float ub, const float* __restrict__ vn,
                                           const float* __restrict__ alb, Geo& g, float& t_self) {
    float S, T;
    sample_st(seed, k, S, T);
    float sq = sqrtf(T);
    float u = 1 - sq;
    float v = (1 - S) * sq;
    float w = S * sq;
    V3 p = bary(u, f.p0, v, f.p1, w, f.p2);
    V3 d = p - o;
    float rs = 1.0f / sqrtf(dot(d, d));
    g.dir = d * rs;
    float hu, hv;
    if (!tri_test(tr, o, g.dir, t_self, hu, hv)) return false;
    g.v = hu;
    g.w = hv;
    g.u = 1.0f - g.v - g.w;
    V3 q = bary(g.u, f.p0, g.v, f.p1, g.w, f.p2);
    V3 dq = q - o;
    g.h = sqrtf(dot(dq, dq));
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

__device__ __forceinline__ float emax0(float x) { return 0.0f < x ? x : 0.0f; }

__device__ __forceinline__ int wave_ticket(int* counter) {
    int b = 0;
    if ((threadIdx.x & 63) == 0) b = atomicAdd(counter, 1);
    return __builtin_amdgcn_readfirstlane(b);
}

// ------------------------------------------------------------------- forward
// Packet occlusion query.  The CH rays of a chunk leave the same wall point towards the same
// small triangle, so one traversal serves all of them: a node is entered when its (padded) box
// meets the pyramid  { o + s*(mx, my, 1) : mx in [mxlo,mxhi], my in [mylo,myhi], 0 <= s <= zmax }
// spanned by the rays' slopes dx/dz, dy/dz and the deepest own-face hit.  Every point of every ray
// segment lies in that pyramid, so no occluder can be missed; leaves run the exact per-ray
// triangle test.  Requires dz > 0 for all rays (the wall faces the scene); the caller falls back
// to the per-ray traversal otherwise.  Returns the still-unoccluded subset of `alive`.
template <int CH>
__device__ __forceinline__ uint32_t trace_packet(const float4* __restrict__ nodes, int n_nodes,
                                                 const float4* __restrict__ tris,
                                                 const int* __restrict__ face_id, V3 o,
                                                 const float (&dx)[CH], const float (&dy)[CH],
                                                 const float (&dz)[CH], const float (&ts)[CH],
                                                 uint32_t alive, int self, int self_fid) {
    const float big = 3.0e38f;
    float mxlo = big, mxhi = -big, mylo = big, myhi = -big, zmax = 0.0f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        if (alive & (1u << c)) {
            float iz = 1.0f / dz[c];
            float mx = dx[c] * iz, my = dy[c] * iz;
            mxlo = fminf(mxlo, mx); mxhi = fmaxf(mxhi, mx);
            mylo = fminf(mylo, my); myhi = fmaxf(myhi, my);
            zmax = fmaxf(zmax, ts[c] * dz[c]);
        }
    }
    // a few ulp of slack on top of the build-time box padding
    mxlo -= 2e-6f * (1.0f + fabsf(mxlo)); mxhi += 2e-6f * (1.0f + fabsf(mxhi));
    mylo -= 2e-6f * (1.0f + fabsf(mylo)); myhi += 2e-6f * (1.0f + fabsf(myhi));
    zmax += 2e-6f * zmax;
    int i = 0;
    while (i >= 0 && alive) {
        int leaf = -1;
        while (i >= 0) {
            const float4 a = nodes[2 * i], b = nodes[2 * i + 1];
            const float za = fmaxf(a.z - o.z, 0.0f);
            const float zb = fminf(b.y - o.z, zmax);
            const float fxlo = fminf(za * mxlo, zb * mxlo), fxhi = fmaxf(za * mxhi, zb * mxhi);
            const float fylo = fminf(za * mylo, zb * mylo), fyhi = fmaxf(za * myhi, zb * myhi);
            const bool hit = (za <= zb) && (a.x - o.x <= fxhi) && (a.w - o.x >= fxlo) &&
                             (a.y - o.y <= fyhi) && (b.x - o.y >= fylo);
            const int esc = __float_as_int(b.z);
            const int link = __float_as_int(b.w);
            if (hit && link < 0) { leaf = ~link; i = esc; break; }
            i = hit ? link : esc;
        }
        if (leaf >= 0 && leaf != self) {
            const Tri tr = load_tri(tris, leaf);
            int lfid = -1;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (alive & (1u << c)) {
                    float t, u, v;
                    if (tri_test(tr, o, mk(dx[c], dy[c], dz[c]), t, u, v)) {
                        bool occ = t < ts[c];
                        if (!occ && t == ts[c]) {
                            if (lfid < 0) lfid = face_id[leaf];
                            occ = lfid < self_fid;
                        }
                        if (occ) alive &= ~(1u << c);
                    }
                }
            }
        }
    }
    return alive;
}

template <int FEAT, int CH>
__global__ __launch_bounds__(256) void k_forward(ForwardArgs a, int rows_in_lds) {
    // one dynamic LDS block: [ticket counter (8 B)][histogram row]; no static LDS in
    // front of it, so the doubles stay 8-byte aligned
    extern __shared__ double s_lds[];
    int* s_next = reinterpret_cast<int*>(s_lds);
    double* s_row = s_lds + 1;

    const int l = blockIdx.x;
    const int nbins = a.sp.nbins;
    const int F = a.sc.F;
    if (rows_in_lds)
        for (int i = threadIdx.x; i < nbins; i += blockDim.x) s_row[i] = 0.0;
    if (threadIdx.x == 0) *s_next = 0;
    __syncthreads();

    const V3 o = ld3(a.src.origin + 3 * (size_t)l);
    const V3 on = ld3(a.src.normal + 3 * (size_t)l);
    const uint64_t lg = (uint64_t)(a.src.source_offset + l);
    const int spt = a.sp.spt;
    const float lb = a.sp.lb, ub = a.sp.ub, res = a.sp.res;
    double* grow = a.rows ? a.rows + (size_t)l * nbins : nullptr;
    const int nblocks = (F + 63) >> 6;
    const int lane = threadIdx.x & 63;

    for (;;) {
        const int b = wave_ticket(s_next);
        if (b >= nblocks) break;
        const int j = (b << 6) + lane;
        if (j >= F) continue;
        const Face f = load_face(a.sc.facerec, j);
        uint32_t* visp = a.vis ? a.vis + ((size_t)l * a.vis_words) * F + j : nullptr;
        if (f.degenerate) {
            if (visp)
                for (int wi = 0; wi < a.vis_words; ++wi) visp[(size_t)wi * F] = 0u;
            continue;
        }
        const Tri tr = load_tri(a.sc.tris, j);
        const uint64_t kbase = (lg * (uint64_t)F + (uint64_t)f.fid) * (uint64_t)spt;
        uint32_t word = 0;
        double inten = 0.0;
        for (int c0 = 0; c0 < spt; c0 += CH) {
            float dx[CH], dy[CH], dz[CH], ts[CH], val[CH];
            int bin[CH];
            uint32_t alive = 0;
            bool zmajor = true;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int s = c0 + c;
                Geo g;
                float t_self = 0.0f;
                bool ok = s < spt;
                if (ok)
                    ok = sample_geo<FEAT>(f, tr, o, a.sp.seed, kbase + (uint64_t)s, lb, ub, a.sc.vertex_normal,
                                          a.sc.albedo, g, t_self);
                float vv = 0.0f;
                int bb = -1;
                if (ok) {
                    float ff = -dot(g.n, g.dir) * dot(on, g.dir) / g.h / g.h;
                    if (a.sp.clamp) {
                        ff = emax0(ff);
                        ok = ff > 0.0f;      // zero contribution in both passes: never trace
                    }
                    vv = f.area * g.alb * ff * ff;
                    if (FEAT & FEAT_GGX) vv = vv * ggx_eval(a.sp.ggx_alpha, dot(g.n, -g.dir));
                    bb = (int)floorf((2.0f * g.h - lb) / res);
                }
                dx[c] = ok ? g.dir.x : 0.0f;
                dy[c] = ok ? g.dir.y : 0.0f;
                dz[c] = ok ? g.dir.z : 1.0f;
                ts[c] = t_self;
                val[c] = vv;
                bin[c] = bb;
                if (ok) {
                    alive |= 1u << c;
                    zmajor = zmajor && (g.dir.z >= 0.05f);
                }
            }
            if (alive) {
                if (zmajor) {
                    alive = trace_packet<CH>(a.sc.nodes, a.sc.n_nodes, a.sc.tris, a.sc.face_id, o, dx, dy, dz, ts,
                                             alive, j, f.fid);
                } else {
#pragma unroll
                    for (int c = 0; c < CH; ++c)
                        if ((alive & (1u << c)) &&
                            occluded(a.sc.nodes, a.sc.n_nodes, a.sc.tris, a.sc.face_id, o, mk(dx[c], dy[c], dz[c]),
                                     ts[c], j, f.fid))
                            alive &= ~(1u << c);
                }
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (alive & (1u << c)) {
                    if (a.mode_intensity) {
                        inten += (double)val[c] / (double)spt;
                    } else if (bin[c] >= 0 && bin[c] < nbins) {
                        double cc = (double)val[c] / (double)spt;
                        if (rows_in_lds) unsafeAtomicAdd(&s_row[bin[c]], cc);
                        else unsafeAtomicAdd(&grow[bin[c]], cc);
                    }
                }
            }
            word |= alive << (c0 & 31);
            if (((c0 + CH) & 31) == 0 || c0 + CH >= spt) {
                if (visp) visp[(size_t)(c0 >> 5) * F] = word;
                word = 0;
            }
        }
        if (a.mode_intensity && inten != 0.0) unsafeAtomicAdd(&a.intensity[f.fid], inten);
    }
    if (rows_in_lds && grow) {
        __syncthreads();
        for (int i = threadIdx.x; i < nbins; i += blockDim.x) grow[i] = s_row[i];
    }
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

template <int FEAT>
__device__ __forceinline__ bool sample_geo_nc(const Face& f, const Tri& tr, V3 oa, V3 ob, uint64_t seed, uint64_t k,
                                              float lb, float ub, const float* __restrict__ vn,
                                              const float* __restrict__ alb, GeoNC& g, float& tA, float& tB) {
    float S, T;
    sample_st(seed, k, S, T);
    float sq = sqrtf(T);
    float u = 1 - sq;
    float v = (1 - S) * sq;
    float w = S * sq;
    V3 p = bary(u, f.p0, v, f.p1, w, f.p2);
    V3 dA = p - oa;
    g.dirA = dA * (1.0f / sqrtf(dot(dA, dA)));
    float hu, hv;
    if (!tri_test(tr, oa, g.dirA, tA, hu, hv)) return false;
    g.v = hu;
    g.w = hv;
    g.u = 1.0f - g.v - g.w;
    V3 qA = bary(g.u, f.p0, g.v, f.p1, g.w, f.p2);
    V3 eA = qA - oa;
    g.d1 = sqrtf(dot(eA, eA));
    V3 dB = p - ob;
    g.dirB = dB * (1.0f / sqrtf(dot(dB, dB)));
    float bu, bv;
    if (!tri_test(tr, ob, g.dirB, tB, bu, bv)) return false;
    V3 qB = bary(1.0f - bu - bv, f.p0, bu, f.p1, bv, f.p2);
    V3 eB = qB - ob;
    g.d2 = sqrtf(dot(eB, eB));
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

// one leg of a chunk: packet traversal when every live ray is z-major, per-ray traversal otherwise
template <int CH>
__device__ __forceinline__ uint32_t trace_leg(const SceneView& sc, V3 o, const float (&dx)[CH], const float (&dy)[CH],
                                              const float (&dz)[CH], const float (&ts)[CH], uint32_t alive,
                                              bool zmajor, int self, int self_fid) {
    if (!alive) return 0u;
    if (zmajor) return trace_packet<CH>(sc.nodes, sc.n_nodes, sc.tris, sc.face_id, o, dx, dy, dz, ts, alive, self, self_fid);
#pragma unroll
    for (int c = 0; c < CH; ++c)
        if ((alive & (1u << c)) &&
            occluded(sc.nodes, sc.n_nodes, sc.tris, sc.face_id, o, mk(dx[c], dy[c], dz[c]), ts[c], self, self_fid))
            alive &= ~(1u << c);
    return alive;
}

template <int FEAT, int CH>
__global__ __launch_bounds__(256) void k_forward_nc(ForwardArgs a, int rows_in_lds) {
    extern __shared__ double s_lds[];       // [ticket (8 B)][histogram row]
    int* s_next = reinterpret_cast<int*>(s_lds);
    double* s_row = s_lds + 1;

    const int l = blockIdx.x;
    const int nbins = a.sp.nbins;
    const int F = a.sc.F;
    if (rows_in_lds)
        for (int i = threadIdx.x; i < nbins; i += blockDim.x) s_row[i] = 0.0;
    if (threadIdx.x == 0) *s_next = 0;
    __syncthreads();

    const V3 oa = ld3(a.src.origin + 3 * (size_t)l), na = ld3(a.src.normal + 3 * (size_t)l);
    const V3 ob = ld3(a.src.sensor + 3 * (size_t)l), nb = ld3(a.src.sensor_normal + 3 * (size_t)l);
    const uint64_t lg = (uint64_t)(a.src.source_offset + l);
    const int spt = a.sp.spt;
    const float lb = a.sp.lb, ub = a.sp.ub, res = a.sp.res;
    double* grow = a.rows + (size_t)l * nbins;
    const int nblocks = (F + 63) >> 6;
    const int lane = threadIdx.x & 63;

    for (;;) {
        const int b = wave_ticket(s_next);
        if (b >= nblocks) break;
        const int j = (b << 6) + lane;
        if (j >= F) continue;
        const Face f = load_face(a.sc.facerec, j);
        uint32_t* visp = a.vis ? a.vis + ((size_t)l * a.vis_words) * F + j : nullptr;
        if (f.degenerate) {
            if (visp)
                for (int wi = 0; wi < a.vis_words; ++wi) visp[(size_t)wi * F] = 0u;
            continue;
        }
        const Tri tr = load_tri(a.sc.tris, j);
        const uint64_t kbase = (lg * (uint64_t)F + (uint64_t)f.fid) * (uint64_t)spt;
        uint32_t word = 0;
        for (int c0 = 0; c0 < spt; c0 += CH) {
            float ax[CH], ay[CH], az[CH], ta[CH], bx[CH], by[CH], bz[CH], tb[CH], val[CH];
            int bin[CH];
            uint32_t alive = 0;
            bool zmA = true, zmB = true;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int s = c0 + c;
                GeoNC g;
                float tA = 0.0f, tB = 0.0f;
                bool ok = s < spt;
                if (ok)
                    ok = sample_geo_nc<FEAT>(f, tr, oa, ob, a.sp.seed, kbase + (uint64_t)s, lb, ub, a.sc.vertex_normal,
                                             a.sc.albedo, g, tA, tB);
                float vv = 0.0f;
                int bb = -1;
                if (ok) {
                    const float ffa = emax0(-dot(g.n, g.dirA) * dot(na, g.dirA) / g.d1 / g.d1);
                    const float ffb = emax0(-dot(g.n, g.dirB) * dot(nb, g.dirB) / g.d2 / g.d2);
                    ok = ffa > 0.0f && ffb > 0.0f;      // zero contribution in both passes: never trace
                    vv = f.area * g.alb * ffa * ffb;
                    bb = (int)floorf(((g.d1 + g.d2) - lb) / res);
                }
                ax[c] = ok ? g.dirA.x : 0.0f; ay[c] = ok ? g.dirA.y : 0.0f; az[c] = ok ? g.dirA.z : 1.0f;
                bx[c] = ok ? g.dirB.x : 0.0f; by[c] = ok ? g.dirB.y : 0.0f; bz[c] = ok ? g.dirB.z : 1.0f;
                ta[c] = tA; tb[c] = tB;
                val[c] = vv;
                bin[c] = bb;
                if (ok) {
                    alive |= 1u << c;
                    zmA = zmA && (g.dirA.z >= 0.05f);
                    zmB = zmB && (g.dirB.z >= 0.05f);
                }
            }
            alive = trace_leg<CH>(a.sc, oa, ax, ay, az, ta, alive, zmA, j, f.fid);
            alive = trace_leg<CH>(a.sc, ob, bx, by, bz, tb, alive, zmB, j, f.fid);
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if ((alive & (1u << c)) && bin[c] >= 0 && bin[c] < nbins) {
                    double cc = (double)val[c] / (double)spt;
                    if (rows_in_lds) unsafeAtomicAdd(&s_row[bin[c]], cc);
                    else unsafeAtomicAdd(&grow[bin[c]], cc);
                }
            }
            word |= alive << (c0 & 31);
            if (((c0 + CH) & 31) == 0 || c0 + CH >= spt) {
                if (visp) visp[(size_t)(c0 >> 5) * F] = word;
                word = 0;
            }
        }
    }
    if (rows_in_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < nbins; i += blockDim.x) grow[i] = s_row[i];
    }
}

// ------------------------------------------------------------- forward (grid)
// Per-source perspective grid.  Every ray of a workgroup starts at the same wall point o, so the
// triangles that can block the ray towards slope (mx, my) = (dx/dz, dy/dz) are exactly those whose
// perspective projection from o covers that slope point.  Per source the workgroup builds, in LDS:
//   * an R x R grid over slope space in CSR form (counting pass, block scan, fill pass) holding,
//     per cell, the triangles whose projection (conservatively rasterised) overlaps the cell AND
//     that are not entirely deeper than the deepest live face seen through that cell -- nothing
//     deeper can be in front of any ray that will ever look the cell up;
//   * a table of projected bounding boxes, quantised outwards to 1/256 of the grid extent.
// A ray then walks only its own cell's list, rejects a candidate from the LDS bounding box (two
// byte compares, no global access) and runs the exact triangle test on the few that remain.  The
// kernel is bound by the vector-memory pipeline (divergent 48-byte record gathers), so everything
// in front of the record load lives in LDS.  The accepted samples are identical to the BVH
// path's: lists and boxes are supersets of what the exact test could report.  Sources for which
// the scene is not strictly in front of the wall point, or whose grid overflows its LDS budget,
// fall back to the stackless BVH traversal.
struct GridView {
    float gx0, gy0, inv_cw, inv_ch;   // cell = floor((m - g0) * inv_c)
    float z0, inv_qz;                 // quantised depth = floor((z - z0) * inv_qz), zmax levels over the scene
    int ib, zmax;                     // entry = index (ib bits) | x0:3 x1:3 y0:3 y1:3 | depth (32-12-ib bits)
    int R;
};

__device__ __forceinline__ int cell_coord(float m, float g0, float inv_c, int R) {
    int c = (int)floorf((m - g0) * inv_c);
    return min(max(c, 0), R - 1);
}

struct Proj2 { float ax, ay, bx, by, cx, cy; };

__device__ __forceinline__ Proj2 project_tri(V3 o, V3 p0, V3 p1, V3 p2) {
    // conservative uses only: approximate reciprocals are covered by the margins below
    const float iz0 = __builtin_amdgcn_rcpf(p0.z - o.z), iz1 = __builtin_amdgcn_rcpf(p1.z - o.z),
                iz2 = __builtin_amdgcn_rcpf(p2.z - o.z);
    Proj2 q;
    q.ax = (p0.x - o.x) * iz0; q.ay = (p0.y - o.y) * iz0;
    q.bx = (p1.x - o.x) * iz1; q.by = (p1.y - o.y) * iz1;
    q.cx = (p2.x - o.x) * iz2; q.cy = (p2.y - o.y) * iz2;
    return q;
}

// Cell-list entry (32 bit, LDS), most significant first:
//   [31:25] smallest depth of the triangle, quantised downwards to 128 levels over the scene's depth range
//   [24:19] y mask, [18:13] x mask: which sixths of THIS cell the triangle's projected bounding box touches
//   [12:0]  index of the triangle (Morton order); the grid path is limited to F <= 8191
// A ray carries  rlim = (its own hit depth level << 25) | 0x1FFFFFF  and  rmask = its sub-cell bit in
// both masks; a candidate survives iff  w <= rlim  (not entirely behind the hit),  (w & rmask) == rmask
// (the slope point is inside the box) and it is not the ray's own face: three compares on one LDS word.
struct BBoxF { float x0, x1, y0, y1; };
constexpr int kSub = 6;          // sub-cell levels per axis
constexpr int kIdxBits = 13;      // single-workgroup grid: 7 depth bits
constexpr int kIdxBitsTiled = 14; // tiled grid: subsets up to 16383 triangles, 6 depth bits

__device__ __forceinline__ uint32_t make_entry(const GridView& g, const BBoxF& bb, int xx, int yy, uint32_t zq, int k) {
    // the box is widened by 0.02 sub-cells: > 50x the fp32 error of the two projections (rcp, 1 ulp)
    const float fx0 = ((bb.x0 - g.gx0) * g.inv_cw - (float)xx) * (float)kSub - 0.02f;
    const float fx1 = ((bb.x1 - g.gx0) * g.inv_cw - (float)xx) * (float)kSub + 0.02f;
    const float fy0 = ((bb.y0 - g.gy0) * g.inv_ch - (float)yy) * (float)kSub - 0.02f;
    const float fy1 = ((bb.y1 - g.gy0) * g.inv_ch - (float)yy) * (float)kSub + 0.02f;
    const int a0 = min(max((int)floorf(fx0), 0), kSub - 1), a1 = min(max((int)floorf(fx1), 0), kSub - 1);
    const int b0 = min(max((int)floorf(fy0), 0), kSub - 1), b1 = min(max((int)floorf(fy1), 0), kSub - 1);
    const uint32_t xm = (2u << a1) - (1u << a0), ym = (2u << b1) - (1u << b0);
    return (zq << (g.ib + 2 * kSub)) | (ym << (g.ib + kSub)) | (xm << g.ib) | (uint32_t)k;
}

// conservative rasterisation of a projected triangle: fn(xx, yy) for every overlapped cell
template <class Fn>
__device__ __forceinline__ void raster_tri(const GridView& g, const Proj2& q, Fn fn) {
    const float cw = __builtin_amdgcn_rcpf(g.inv_cw), ch = __builtin_amdgcn_rcpf(g.inv_ch);
    const float mgx = 1e-3f * cw, mgy = 1e-3f * ch;          // >> fp32 rounding of the projection
    const int cx0 = cell_coord(fminf(fminf(q.ax, q.bx), q.cx) - mgx, g.gx0, g.inv_cw, g.R);
    const int cx1 = cell_coord(fmaxf(fmaxf(q.ax, q.bx), q.cx) + mgx, g.gx0, g.inv_cw, g.R);
    const int cy0 = cell_coord(fminf(fminf(q.ay, q.by), q.cy) - mgy, g.gy0, g.inv_ch, g.R);
    const int cy1 = cell_coord(fmaxf(fmaxf(q.ay, q.by), q.cy) + mgy, g.gy0, g.inv_ch, g.R);
    // edge functions, oriented so that the inside is >= 0
    const float area = (q.bx - q.ax) * (q.cy - q.ay) - (q.by - q.ay) * (q.cx - q.ax);
    const float sgn = area < 0.0f ? -1.0f : 1.0f;
    const bool thin = fabsf(area) < 1e-4f * cw * ch;         // edge-on: bbox cells only
    const float A0 = -(q.by - q.ay) * sgn, B0 = (q.bx - q.ax) * sgn, C0 = -(A0 * q.ax + B0 * q.ay);
    const float A1 = -(q.cy - q.by) * sgn, B1 = (q.cx - q.bx) * sgn, C1 = -(A1 * q.bx + B1 * q.by);
    const float A2 = -(q.ay - q.cy) * sgn, B2 = (q.ax - q.cx) * sgn, C2 = -(A2 * q.cx + B2 * q.cy);
    const float t0 = 2e-3f * (fabsf(A0) * cw + fabsf(B0) * ch);
    const float t1 = 2e-3f * (fabsf(A1) * cw + fabsf(B1) * ch);
    const float t2 = 2e-3f * (fabsf(A2) * cw + fabsf(B2) * ch);
    for (int yy = cy0; yy <= cy1; ++yy) {
        const float y0 = g.gy0 + (float)yy * ch, y1 = y0 + ch;
        for (int xx = cx0; xx <= cx1; ++xx) {
            const float x0 = g.gx0 + (float)xx * cw, x1 = x0 + cw;
            bool in = true;
            if (!thin) {
                in = (A0 * (A0 > 0 ? x1 : x0) + B0 * (B0 > 0 ? y1 : y0) + C0 >= -t0) &&
                     (A1 * (A1 > 0 ? x1 : x0) + B1 * (B1 > 0 ? y1 : y0) + C1 >= -t1) &&
                     (A2 * (A2 > 0 ? x1 : x0) + B2 * (B2 > 0 ? y1 : y0) + C2 >= -t2);
            }
            if (in) fn(xx, yy);
        }
    }
}

// Slope-space frame of a source: bounding rectangle of the projection of the BVH's (padded) root box.
// Shared by the grid kernel and the tile-binning kernel, which must agree bit for bit.
struct SourceFrame { bool ok; float gx0, gy0, wx, wy, zr0, zr1; };
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
    return fr;
}

// Tile binning for the tiled grid: one workgroup per source appends every triangle to the subset of each
// slope-space tile its projected bounding box meets (with the rasteriser's margin).  O(F) per source -- the
// tiles' workgroups then read their subset instead of scanning the whole mesh each.
__global__ __launch_bounds__(512) void k_tile_bin(ForwardArgs a, int R) {
    const int l = blockIdx.x;
    const V3 o = ld3(a.src.origin + 3 * (size_t)l);
    const SourceFrame fr = source_frame(a.sc.nodes, o);
    if (!fr.ok) return;                                   // tile 0 handles such a source alone
    const int ntx = a.tiles_x, nty = a.tiles_y;
    const float tw = fr.wx * 1.002f / (float)ntx, th = fr.wy * 1.002f / (float)nty;
    const float inv_tw = 1.0f / tw, inv_th = 1.0f / th;
    const float mx = 2e-3f * tw / (float)R, my = 2e-3f * th / (float)R;
    // slot allocation with LDS counters (one workgroup owns the whole source): global atomics on the few
    // per-tile counters were the bottleneck (2.5 ms for 1024 sources x 20 k faces)
    __shared__ int s_cnt[1024];
    const int ntile = ntx * nty;
    for (int i = threadIdx.x; i < ntile; i += blockDim.x) s_cnt[i] = 0;
    __syncthreads();
    uint32_t* lists = a.tile_list + (size_t)l * ntile * a.tile_cap;
    for (int j = threadIdx.x; j < a.sc.F; j += blockDim.x) {
        const float4 q0 = a.sc.facerec[4 * j], q1 = a.sc.facerec[4 * j + 1], q2 = a.sc.facerec[4 * j + 2];
        const Proj2 q = project_tri(o, mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y), mk(q1.z, q1.w, q2.x));
        const float bx0 = fminf(fminf(q.ax, q.bx), q.cx) - mx, bx1 = fmaxf(fmaxf(q.ax, q.bx), q.cx) + mx;
        const float by0 = fminf(fminf(q.ay, q.by), q.cy) - my, by1 = fmaxf(fmaxf(q.ay, q.by), q.cy) + my;
        // one extra tile on each side covers the rounding of the tile origins (gx0 + t * tw)
        const int t0 = max((int)floorf((bx0 - fr.gx0) * inv_tw - 1e-3f), 0), t1 = min((int)floorf((bx1 - fr.gx0) * inv_tw + 1e-3f), ntx - 1);
        const int u0 = max((int)floorf((by0 - fr.gy0) * inv_th - 1e-3f), 0), u1 = min((int)floorf((by1 - fr.gy0) * inv_th + 1e-3f), nty - 1);
        for (int u = u0; u <= u1; ++u)
            for (int t = t0; t <= t1; ++t) {
                const int ti = u * ntx + t;
                const int pos = atomicAdd(&s_cnt[ti], 1);
                if (pos < a.tile_cap) lists[(size_t)ti * a.tile_cap + pos] = (uint32_t)j;
            }
    }
    __syncthreads();
    int* cnt = a.tile_count + (size_t)l * ntile;
    for (int i = threadIdx.x; i < ntile; i += blockDim.x) cnt[i] = s_cnt[i];
}

// two 512-thread workgroups per CU = 4 waves per SIMD: keep the kernel within 128 VGPRs
// NCM (row N, non-confocal pairs): 0 = confocal; 1 = visibility-only pass from the SENSOR of each pair
// (bits -> a.vis2, no histogram); 2 = pass from the LASER that evaluates both legs' geometry, ANDs the
// sensor-leg bits and traces only the laser leg.  One perspective grid serves one origin, so a pair costs
// two grid passes (about 2x the confocal forward) instead of two BVH traversals per sample (9x).
//
// TILED (meshes whose cell lists do not fit one workgroup's LDS, F > ~7.6 k): slope space is cut into
// tiles_x * tiles_y tiles and one workgroup handles one (source, tile): it selects the triangles whose
// projected bounding box meets its tile (ids -> a.tile_list, at most 8191: the 13-bit entry index is then
// an index into that list), builds the grid over the tile only, and traces exactly the samples whose
// slope point falls into the tile -- every sample is owned by one tile, decided from the source's global
// frame so that all workgroups of a source agree.  Rows and visibility words are combined with atomics
// (the launcher zeroes them).  A tile whose subset overflows iterates over all faces and uses the BVH
// query for its own samples; a source whose scene is not strictly in front is handled by tile 0 alone.
// Tiles whose cell lists overflow the normal LDS share (two workgroups per CU) flag themselves and leave
// before writing anything; a second launch (`pass` = 1) with one workgroup per CU and ~150 KB of LDS redoes
// exactly those tiles (grazing views pile thousands of sliver triangles into a few tiles).
template <int FEAT, int NCM = 0, bool TILED = false>
__global__ __launch_bounds__(512, 4) void k_forward_grid(ForwardArgs a, int rows_in_lds, int R, int cap, int pass = 0) {
    // dynamic LDS: [ctl: ticket, bad, total, n_live (16 B)][row nbins f64][cells R*R+1 u32]
    //   [union { build: depth bound per 2x2 cells R2*R2 u32, block masks nblk u64 ;
    //            trace: 8 waves x (128 queued pairs + 2 mask words) }][entries cap u32]
    // (the bucketed live-face list lives in global scratch: it is read once per 64-face block)
    extern __shared__ double s_lds[];
    constexpr int IB = TILED ? kIdxBitsTiled : kIdxBits;
    if (TILED && pass == 1 && a.tile_count[gridDim.x + blockIdx.x] == 0) return;     // only the flagged tiles
    int* s_ctl = reinterpret_cast<int*>(s_lds);      // 8 ints: ticket, bad, total entries, n_live, tile subset size
    double* s_row = s_lds + 4;
    const int nbins = a.sp.nbins;
    const int ncell = R * R;
    const int R2 = (R + 1) >> 1;
    const int F = a.sc.F;
    const int ntiles = TILED ? a.tiles_x * a.tiles_y : 1;
    const int tile = TILED ? (int)(blockIdx.x % (unsigned)ntiles) : 0;
    const int tile_x = TILED ? tile % a.tiles_x : 0, tile_y = TILED ? tile / a.tiles_x : 0;
    const int mask_blocks = TILED ? max((a.tile_cap + 63) >> 6, (F + 63) >> 6) : (F + 63) >> 6;   // LDS sizing only
    uint32_t* s_cell = reinterpret_cast<uint32_t*>(s_row + (rows_in_lds ? nbins : 0));
    uint32_t* s_union = s_cell + ((ncell + 2) & ~1);
    uint32_t* s_zc = s_union;                                                   // build phase
    unsigned long long* s_mask = reinterpret_cast<unsigned long long*>(s_zc + ((R2 * R2 + 1) & ~1));
    uint32_t* s_queue = s_union;                                                // trace phase
    const int union_words = max(((R2 * R2 + 1) & ~1) + 2 * mask_blocks, 8 * 130);
    uint32_t* s_ent = s_union + ((union_words + 1) & ~1);
    uint16_t* g_live = a.live + (size_t)blockIdx.x * (TILED ? a.tile_cap : F);
    uint32_t* tl = TILED ? a.tile_list + (size_t)blockIdx.x * a.tile_cap : nullptr;
    __shared__ uint32_t s_scan[512];

    const int l = TILED ? (int)(blockIdx.x / (unsigned)ntiles) : (int)blockIdx.x;
    const int tid = threadIdx.x, NT = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nwaves = NT >> 6;
    const V3 o = ld3((NCM == 1 ? a.src.sensor : a.src.origin) + 3 * (size_t)l);
    const V3 on = ld3((NCM == 1 ? a.src.sensor_normal : a.src.normal) + 3 * (size_t)l);
    const V3 ob = NCM == 2 ? ld3(a.src.sensor + 3 * (size_t)l) : o;
    const V3 onb = NCM == 2 ? ld3(a.src.sensor_normal + 3 * (size_t)l) : on;
    uint32_t* const visout = NCM == 1 ? a.vis2 : a.vis;
#ifdef NLOS_FWD_STAMPS
    // diagnostic build only: per-phase cycles summed over workgroups -> a.dbg[0..5]
    long long t_prev = clock64();
    int t_slot = 0;
#define FWD_STAMP() do { __syncthreads(); if (tid == 0 && a.dbg) { long long t_now = clock64(); atomicAdd((unsigned long long*)&a.dbg[t_slot], (unsigned long long)(t_now - t_prev)); ++t_slot; t_prev = t_now; } } while (0)
#else
#define FWD_STAMP() do { } while (0)
#endif

    // ---- grid frame from the (padded) root box of the BVH: O(1) per source ------------------
    const SourceFrame fr = source_frame(a.sc.nodes, o);
    const float zr0 = fr.zr0, zr1 = fr.zr1;
    const bool frame_ok = fr.ok;
    GridView g;
    g.R = R;
    float Gx0 = 0.0f, Gy0 = 0.0f, inv_tw = 0.0f, inv_th = 0.0f;     // TILED: the source's global frame
    {
        const float wx = fr.wx, wy = fr.wy;
        g.gx0 = fr.gx0;
        g.gy0 = fr.gy0;
        g.inv_cw = (float)R / (wx * 1.002f);
        g.inv_ch = (float)R / (wy * 1.002f);
        if (TILED) {
            // the source's frame [G0, G0 + W) is cut into tiles; ownership of a slope point is decided
            // with (G0, inv_tw) only, which every workgroup of the source computes identically
            Gx0 = g.gx0; Gy0 = g.gy0;
            const float tw = wx * 1.002f / (float)a.tiles_x, th = wy * 1.002f / (float)a.tiles_y;
            inv_tw = 1.0f / tw; inv_th = 1.0f / th;
            g.gx0 = Gx0 + (float)tile_x * tw;
            g.gy0 = Gy0 + (float)tile_y * th;
            g.inv_cw = (float)R / tw;
            g.inv_ch = (float)R / th;
        }
        g.ib = IB;
        g.zmax = (1 << (32 - 2 * kSub - IB)) - 1;
        g.z0 = zr0;
        g.inv_qz = (float)g.zmax / fmaxf(zr1 - zr0, 1e-12f);
    }

    if (rows_in_lds)
        for (int i = tid; i < nbins; i += NT) s_row[i] = 0.0;
    for (int i = tid; i <= ncell; i += NT) s_cell[i] = 0u;
    for (int i = tid; i < R2 * R2; i += NT) s_zc[i] = 0u;
    if (tid == 0) { s_ctl[0] = 0; s_ctl[1] = frame_ok ? 0 : 1; s_ctl[2] = 0; s_ctl[3] = 0; s_ctl[4] = 0; }
    __syncthreads();

    // Fl faces are iterated by this workgroup; `ident`: local index == sorted face index
    int Fl = F;
    bool ident = true;
    if (TILED) {
        if (!frame_ok) {
            if (tile != 0) return;                     // tile 0 handles such a source alone (BVH queries)
        } else {
            // ---- tile subset, binned by k_tile_bin
            const int nsel = a.tile_count[blockIdx.x];
#ifdef NLOS_FWD_STAMPS
            if (tid == 0 && a.dbg) {
                if (nsel > a.tile_cap) atomicAdd((unsigned long long*)&a.dbg[22], 1ull);
                atomicMax((unsigned long long*)&a.dbg[23], (unsigned long long)nsel);
            }
#endif
            if (nsel > a.tile_cap || nsel > (1 << IB) - 1) {
                if (tid == 0) s_ctl[1] = 1;            // subset overflow: all faces, BVH query, own samples only
            } else {
                Fl = nsel;
                ident = false;
            }
            __syncthreads();
        }
    }
    const int nblocks = (Fl + 63) >> 6;
    auto gid = [&](int j) -> int { return (TILED && !ident) ? (int)tl[j] : j; };
    const bool compact = !(TILED && ident);            // ident tiles may exceed the u16 live list: no compaction

    // ---- which faces can contribute at all?  (order-preserving compaction, per 64-face block) ----
    // With face normals and the clamped form factor, -dot(n,dir) has the sign of dist(o, plane(f))
    // for every sample of f: if the wall point is clearly behind the face (and the face in front
    // of the wall), every contribution is exactly 0 -- nothing to sample, nothing to trace.
    // Live faces also record, per 2x2 block of cells they project to, the largest depth at which
    // a ray of this source can end.
    auto face_dark = [&](const Face& f) -> bool {
        bool dark = f.degenerate;
        if (!dark && !(FEAT & FEAT_VN) && a.sp.clamp) {
            const float dist = dot(f.fn, o - f.p0);
            const float sc = fabsf(o.x - f.p0.x) + fabsf(o.y - f.p0.y) + fabsf(o.z - f.p0.z);
            const bool behind = dist < -1e-4f * sc;
            const bool infront = dot(on, f.p0 - o) > 1e-4f * sc && dot(on, f.p1 - o) > 1e-4f * sc &&
                                 dot(on, f.p2 - o) > 1e-4f * sc;
            dark = behind && infront;
        }
        return dark;
    };
    for (int b = wave; b < nblocks; b += nwaves) {
        const int j = (b << 6) + lane;
        bool live = false;
        if (j < Fl) {
            const int jg = gid(j);
            const Face f = load_face(a.sc.facerec, jg);
            const bool dark = face_dark(f);
            live = !dark;
            if (!TILED && dark && visout) {
                uint32_t* visp = visout + ((size_t)l * a.vis_words) * F + j;
                for (int wi = 0; wi < a.vis_words; ++wi) visp[(size_t)wi * F] = 0u;
            }
            if (live && frame_ok && !(TILED && ident)) {
                const float zfar = fmaxf(fmaxf(f.p0.z, f.p1.z), f.p2.z) - o.z;
                const uint32_t zb = __float_as_uint(fmaxf(zfar, 0.0f) * 1.0001f + 1e-30f);
                const Proj2 q = project_tri(o, f.p0, f.p1, f.p2);
                raster_tri(g, q, [&](int xx, int yy) { atomicMax(&s_zc[(yy >> 1) * R2 + (xx >> 1)], zb); });
            }
        }
        const unsigned long long m = __ballot(live);
        if (lane == 0) s_mask[b] = m;
    }
    __syncthreads();
    FWD_STAMP();   // 0: setup + live-face masks + depth bounds

    if (frame_ok && !(TILED && ident)) {
        // ---- counting pass ---------------------------------------------------------------------------
        for (int jl = tid; jl < Fl; jl += NT) {
            const int j = gid(jl);
            const float4 q0 = a.sc.facerec[4 * j], q1 = a.sc.facerec[4 * j + 1], q2 = a.sc.facerec[4 * j + 2];
            const Proj2 q = project_tri(o, mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y), mk(q1.z, q1.w, q2.x));
            const uint32_t zn = __float_as_uint(fmaxf(fminf(fminf(q0.z, q1.y), q2.x) - o.z, 0.0f));
            raster_tri(g, q, [&](int xx, int yy) {
                if (zn <= s_zc[(yy >> 1) * R2 + (xx >> 1)]) atomicAdd(&s_cell[yy * R + xx], 1u);
            });
        }
    }
    __syncthreads();
    FWD_STAMP();   // 1: counting pass
    if (tid == 0) {
        uint32_t run = 0;
        for (int b = 0; b < nblocks; ++b) run += (uint32_t)__popcll(s_mask[b]);
        s_ctl[3] = (int)run;
    }
    if (frame_ok) {
        // ---- exclusive scan of the cell counts (each thread owns a contiguous slice) -------------
        const int per = (ncell + NT - 1) / NT;
        const int c0 = min(tid * per, ncell), c1 = min(c0 + per, ncell);
        uint32_t sum = 0;
        for (int c = c0; c < c1; ++c) sum += s_cell[c];
        s_scan[tid] = sum;
        __syncthreads();
        for (int off = 1; off < NT; off <<= 1) {
            uint32_t v = tid >= off ? s_scan[tid - off] : 0u;
            __syncthreads();
            s_scan[tid] += v;
            __syncthreads();
        }
        uint32_t run = s_scan[tid] - sum;
        for (int c = c0; c < c1; ++c) { uint32_t n = s_cell[c]; s_cell[c] = run; run += n; }
        if (tid == NT - 1) { s_ctl[2] = (int)s_scan[tid]; if ((int)s_scan[tid] > cap) s_ctl[1] = 1; }
    }
    __syncthreads();
    if (TILED && pass == 0 && frame_ok && !ident && s_ctl[1] != 0) {
        if (tid == 0) a.tile_count[gridDim.x + blockIdx.x] = 1;      // cell lists overflow: redo with the big-LDS launch
        return;
    }
    FWD_STAMP();   // 2: scans
    // ---- fill pass: s_cell[c] is the write cursor, afterwards the END of cell c ------------------
    if (frame_ok && s_ctl[1] == 0) {
        for (int jl = tid; jl < Fl; jl += NT) {
            const int j = gid(jl);
            const float4 q0 = a.sc.facerec[4 * j], q1 = a.sc.facerec[4 * j + 1], q2 = a.sc.facerec[4 * j + 2];
            const Proj2 q = project_tri(o, mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y), mk(q1.z, q1.w, q2.x));
            const float zmin_rel = fmaxf(fminf(fminf(q0.z, q1.y), q2.x) - o.z, 0.0f);
            const uint32_t zn = __float_as_uint(zmin_rel);
            const uint32_t zq = (uint32_t)min(max((int)floorf((zmin_rel - g.z0) * g.inv_qz) - 1, 0), g.zmax);
            BBoxF bb;
            bb.x0 = fminf(fminf(q.ax, q.bx), q.cx); bb.x1 = fmaxf(fmaxf(q.ax, q.bx), q.cx);
            bb.y0 = fminf(fminf(q.ay, q.by), q.cy); bb.y1 = fmaxf(fmaxf(q.ay, q.by), q.cy);
            raster_tri(g, q, [&](int xx, int yy) {
                if (zn <= s_zc[(yy >> 1) * R2 + (xx >> 1)]) {
                    uint32_t pos = atomicAdd(&s_cell[yy * R + xx], 1u);
                    s_ent[pos] = make_entry(g, bb, xx, yy, zq, jl);
                }
            });
        }
    }
    __syncthreads();
    FWD_STAMP();   // 3: fill pass
    // ---- live list, bucketed by the length of the list of the face's centroid cell ----------------
    // Cell lists have a heavy tail (mean 16, max > 60 entries) and the filter walk below is a
    // lockstep loop: a wave is as slow as its longest list.  Handing out the live faces in
    // buckets of similar list length (longest first) makes the 64 lists of a wave comparable.
    const bool use_grid = s_ctl[1] == 0;
#ifdef NLOS_FWD_STAMPS
    if (TILED && tid == 0 && a.dbg && !use_grid && !ident) atomicAdd((unsigned long long*)&a.dbg[20], 1ull);   // entry overflow
    if (TILED && tid == 0 && a.dbg) atomicMax((unsigned long long*)&a.dbg[21], (unsigned long long)s_ctl[2]);
#endif
    constexpr int NB = 16;                                   // buckets of 4 entries
    auto face_bucket = [&](int j) -> int {
        if (!use_grid) return 0;
        // longest list among the cells under the face's projected bounding box
        const int jg = gid(j);
        const float4 q0 = a.sc.facerec[4 * jg], q1 = a.sc.facerec[4 * jg + 1], q2 = a.sc.facerec[4 * jg + 2];
        const Proj2 q = project_tri(o, mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y), mk(q1.z, q1.w, q2.x));
        const int cx0 = cell_coord(fminf(fminf(q.ax, q.bx), q.cx), g.gx0, g.inv_cw, R);
        const int cx1 = cell_coord(fmaxf(fmaxf(q.ax, q.bx), q.cx), g.gx0, g.inv_cw, R);
        const int cy0 = cell_coord(fminf(fminf(q.ay, q.by), q.cy), g.gy0, g.inv_ch, R);
        const int cy1 = cell_coord(fmaxf(fmaxf(q.ay, q.by), q.cy), g.gy0, g.inv_ch, R);
        uint32_t n = 0;
        for (int yy = cy0; yy <= cy1; ++yy)
            for (int xx = cx0; xx <= cx1; ++xx) {
                const int c = yy * R + xx;
                n = max(n, s_cell[c] - (c > 0 ? s_cell[c - 1] : 0u));
            }
        return (NB - 1) - (int)min(n >> 2, (uint32_t)(NB - 1));   // bucket 0 = longest lists
    };
    if (tid < 2 * NB) s_scan[tid] = 0u;
    __syncthreads();
    for (int b = wave; compact && b < nblocks; b += nwaves) {
        if ((s_mask[b] >> lane) & 1ull) atomicAdd(&s_scan[face_bucket((b << 6) + lane)], 1u);
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int q = 0; q < NB; ++q) { s_scan[NB + q] = run; run += s_scan[q]; }
    }
    __syncthreads();
    for (int b = wave; compact && b < nblocks; b += nwaves) {
        if ((s_mask[b] >> lane) & 1ull) {
            const int j = (b << 6) + lane;
            g_live[atomicAdd(&s_scan[NB + face_bucket(j)], 1u)] = (uint16_t)j;
        }
    }
    __syncthreads();
    FWD_STAMP();   // 4: bucketed live list
    const int n_live = compact ? s_ctl[3] : Fl;
    const int live_blocks = (n_live + 63) >> 6;

    // ---- trace + histogram: dense lanes over the live faces ----------------------------------------
    // Per sample the 64 rays of a wave are handled in two wave-synchronous stages:
    //  (1) filter: every lane walks its own cell list in lockstep with LDS-only work (entry index +
    //      quantised projected box) and appends the survivors, as (owner lane, triangle) pairs, to a
    //      wave-private LDS queue (ballot + prefix rank);
    //  (2) exact test: whenever 64 pairs are queued (and at the end) each lane takes ONE pair, pulls
    //      the owner's ray through ds_bpermute, gathers the 48-byte record and runs the triangle
    //      test; hits are OR-ed into the wave's occlusion mask.
    // The kernel is VALU-issue bound and cell lists have a heavy tail (mean 17, wave-max 36
    // entries; 5.7 exact tests per ray at 22 % lane occupancy when done in place), so the expensive
    // stage must run on dense lanes and must not wait for the longest list.
    const uint64_t lg = (uint64_t)(a.src.source_offset + l);
    const int spt = a.sp.spt;
    const float lb = a.sp.lb, ub = a.sp.ub, res = a.sp.res;
    double* grow = a.rows ? a.rows + (size_t)l * nbins : nullptr;
#ifdef NLOS_FWD_STAMPS
    unsigned long long c_rays = 0, c_pairs = 0, c_iters = 0, c_mt = 0, c_mtw = 0;   // diagnostic build only
    long long tg = 0, ts = 0, tx = 0, th = 0, tmark = 0;
#define TMARK() (tmark = clock64())
#define TACC(v) do { long long now_ = clock64(); v += now_ - tmark; tmark = now_; } while (0)
#else
#define TMARK() do { } while (0)
#define TACC(v) do { } while (0)
#endif
    uint32_t* wq = s_queue + wave * 128;                 // this wave's pair queue (aliases the build-phase tables)
    uint32_t* wocc = s_queue + nwaves * 128 + wave * 2;  // this wave's 64-bit occlusion mask
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    for (;;) {
        const int b = wave_ticket(&s_ctl[0]);
        if (b >= live_blocks) break;
        const int li = (b << 6) + lane;
        bool has_face = li < n_live;
        const int j = has_face ? (compact ? (int)g_live[li] : li) : 0;     // index within this workgroup's face set
        const int jg = gid(j);                                              // sorted-face index
        const Face f = load_face(a.sc.facerec, jg);
        if (!compact && has_face) has_face = !face_dark(f);                 // (the block masks are gone: the queue reuses their LDS)
        if (TILED && !compact && has_face && frame_ok) {
            // overflowed subset: every face is visited, most of them lie outside this tile
            const Proj2 q = project_tri(o, f.p0, f.p1, f.p2);
            const float cwm = __builtin_amdgcn_rcpf(g.inv_cw) * (1.0f + 4e-3f), chm = __builtin_amdgcn_rcpf(g.inv_ch) * (1.0f + 4e-3f);
            has_face = fmaxf(fmaxf(q.ax, q.bx), q.cx) >= g.gx0 - 4e-3f * cwm && fminf(fminf(q.ax, q.bx), q.cx) <= g.gx0 + (float)R * cwm &&
                       fmaxf(fmaxf(q.ay, q.by), q.cy) >= g.gy0 - 4e-3f * chm && fminf(fminf(q.ay, q.by), q.cy) <= g.gy0 + (float)R * chm;
        }
        uint32_t* visp = (visout && has_face) ? visout + ((size_t)l * a.vis_words) * F + jg : nullptr;
        const uint32_t* visb = (NCM == 2 && has_face) ? a.vis2 + ((size_t)l * a.vis_words) * F + jg : nullptr;
        const Tri tr = load_tri(a.sc.tris, jg);
        const uint64_t kbase = (lg * (uint64_t)F + (uint64_t)f.fid) * (uint64_t)spt;
        uint32_t word = 0, word_b = 0;
        double inten = 0.0;
        for (int s = 0; s < spt; ++s) {
            TMARK();
            V3 dir = mk(0.0f, 0.0f, 1.0f);
            float t_self = 0.0f, val = 0.0f;
            int bin = -1;
            bool ok = has_face;
            if (NCM == 2 && (s & 31) == 0 && has_face) word_b = visb[(size_t)(s >> 5) * F];
            if (NCM == 0) {
                Geo gg;
                if (ok)
                    ok = sample_geo<FEAT>(f, tr, o, a.sp.seed, kbase + (uint64_t)s, lb, ub, a.sc.vertex_normal,
                                          a.sc.albedo, gg, t_self);
                if (ok) {
                    float ff = -dot(gg.n, gg.dir) * dot(on, gg.dir) / gg.h / gg.h;
                    if (a.sp.clamp) {
                        ff = emax0(ff);
                        ok = ff > 0.0f;
                    }
                    val = f.area * gg.alb * ff * ff;
                    if (FEAT & FEAT_GGX) val = val * ggx_eval(a.sp.ggx_alpha, dot(gg.n, -gg.dir));
                    bin = (int)floorf((2.0f * gg.h - lb) / res);
                    dir = gg.dir;
                }
            } else if (NCM == 1) {
                // sensor leg only: is the stratified point the closest hit seen from the sensor?
                if (ok) {
                    float S, T;
                    sample_st(a.sp.seed, kbase + (uint64_t)s, S, T);
                    const float sq = sqrtf(T);
                    const V3 p = bary(1 - sq, f.p0, (1 - S) * sq, f.p1, S * sq, f.p2);
                    const V3 d = p - o;
                    dir = d * (1.0f / sqrtf(dot(d, d)));
                    float hu, hv;
                    ok = tri_test(tr, o, dir, t_self, hu, hv);
                    // a leg whose form factor is exactly zero is rejected by the laser pass anyway: skip its ray
                    if (ok && !(FEAT & FEAT_VN)) ok = (-dot(f.fn, dir) * dot(on, dir)) > 0.0f;
                }
            } else {
                GeoNC gc;
                float t_b;
                if (ok)
                    ok = sample_geo_nc<FEAT>(f, tr, o, ob, a.sp.seed, kbase + (uint64_t)s, lb, ub, a.sc.vertex_normal,
                                             a.sc.albedo, gc, t_self, t_b);
                if (ok) {
                    const float ffa = emax0(-dot(gc.n, gc.dirA) * dot(on, gc.dirA) / gc.d1 / gc.d1);
                    const float ffb = emax0(-dot(gc.n, gc.dirB) * dot(onb, gc.dirB) / gc.d2 / gc.d2);
                    ok = ffa > 0.0f && ffb > 0.0f && ((word_b >> (s & 31)) & 1u);
                    val = f.area * gc.alb * ffa * ffb;
                    bin = (int)floorf(((gc.d1 + gc.d2) - lb) / res);
                    dir = gc.dirA;
                }
            }
            if (TILED && ok && frame_ok) {
                // the tile that owns this sample: from the source's global frame, identical in every workgroup
                const float izo = __builtin_amdgcn_rcpf(dir.z);
                const int ti = min(max((int)floorf((dir.x * izo - Gx0) * inv_tw), 0), a.tiles_x - 1);
                const int tj = min(max((int)floorf((dir.y * izo - Gy0) * inv_th), 0), a.tiles_y - 1);
                ok = dir.z > 0.0f ? (ti == tile_x && tj == tile_y) : tile == 0;
            }
            if (!ok) dir = mk(0.0f, 0.0f, 1.0f);
            const bool grid_ray = ok && use_grid && dir.z > 0.0f;
            if (ok && !grid_ray)
                ok = !occluded(a.sc.nodes, a.sc.n_nodes, a.sc.tris, a.sc.face_id, o, dir, t_self, jg, f.fid);

            // ---- stage 1 + 2 (wave-synchronous; every lane takes part) ----
            uint32_t e = 0, e1 = 0, rmask = 0, rlim = 0;
            if (grid_ray) {
                const float iz = __builtin_amdgcn_rcpf(dir.z);   // lookups only: 1-ulp rcp is fine
                const float ux = (dir.x * iz - g.gx0) * g.inv_cw, uy = (dir.y * iz - g.gy0) * g.inv_ch;
                const int cxx = min(max((int)floorf(ux), 0), R - 1), cyy = min(max((int)floorf(uy), 0), R - 1);
                const int sx = min(max((int)floorf((ux - (float)cxx) * (float)kSub), 0), kSub - 1);
                const int sy = min(max((int)floorf((uy - (float)cyy) * (float)kSub), 0), kSub - 1);
                rmask = (1u << (IB + sx)) | (1u << (IB + kSub + sy));
                // depth level of the own-face hit, rounded up: anything quantised deeper cannot occlude
                const float zs = t_self * dir.z;
                const uint32_t rq = (uint32_t)min(max((int)floorf((zs * 1.00002f - g.z0) * g.inv_qz) + 1, 0), g.zmax);
                rlim = (rq << (IB + 2 * kSub)) | ((1u << (IB + 2 * kSub)) - 1u);
                const int c = cyy * R + cxx;
                e1 = s_cell[c];
                e = c > 0 ? s_cell[c - 1] : 0u;
            }
            if (lane < 2) wocc[lane] = 0u;
            TACC(tg);
            int qn = 0;                                        // wave-uniform
            auto exact_round = [&](int n) {
#ifdef NLOS_FWD_STAMPS
                if (lane == 0) c_mtw += 1;
#endif
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const uint32_t pr = wq[lane < n ? lane : 0];
                const int owner = (int)(pr >> 16), k = (int)(pr & 0xFFFFu);
                const V3 od = mk(__shfl(dir.x, owner), __shfl(dir.y, owner), __shfl(dir.z, owner));
                const float ot = __shfl(t_self, owner);
                const int ofid = __shfl(f.fid, owner);
                if (lane < n) {
                    const int kg = gid(k);
                    const Tri tk = load_tri(a.sc.tris, kg);
                    if (tri_occludes(tk, o, od, ot, ofid, a.sc.face_id, kg))
                        atomicOr(&wocc[owner >> 5], 1u << (owner & 31));
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            };
            // lockstep filter walk (LDS only): entry index + packed box/depth word.  (A fully
            // flattened walk -- pairs spread evenly over the lanes with a prefix-sum owner search --
            // halves the iterations but its dependent ds_bpermute chain makes it slower; measured.)
#ifdef NLOS_FWD_STAMPS
            if (grid_ray) c_rays += 1;
#endif
            constexpr uint32_t imask = (1u << IB) - 1u;
            auto push = [&](bool pass, int k) {
                const unsigned long long m = __ballot(pass);
                if (m) {
                    if (pass) wq[qn + __popcll(m & lt_mask)] = ((uint32_t)lane << 16) | (uint32_t)k;
                    qn += __popcll(m);
                    if (qn >= 64) {
                        TACC(ts);
                        exact_round(64);
                        TACC(tx);
                        qn -= 64;
                        const uint32_t mv = wq[64 + (lane < qn ? lane : 0)];
                        __builtin_amdgcn_wave_barrier();
                        if (lane < qn) wq[lane] = mv;
                    }
                }
            };
            // two entries per trip (one ds_read2_b32): halves the loop overhead of the lockstep walk
            while (__any(e < e1)) {
                bool p0 = false, p1 = false;
                int k0 = 0, k1 = 0;
                if (e < e1) {
                    const uint32_t w0 = s_ent[e], w1 = s_ent[e + 1];
                    k0 = (int)(w0 & imask);
                    k1 = (int)(w1 & imask);
                    p0 = (w0 <= rlim) & ((w0 & rmask) == rmask) & (k0 != j);
                    p1 = (e + 1 < e1) & (w1 <= rlim) & ((w1 & rmask) == rmask) & (k1 != j);
#ifdef NLOS_FWD_STAMPS
                    if (grid_ray) c_pairs += (e + 1 < e1) ? 2 : 1;
#endif
                    e += 2;
                }
#ifdef NLOS_FWD_STAMPS
                if (lane == 0) c_iters += 1;
                if (p0) c_mt += 1;
                if (p1) c_mt += 1;
#endif
                push(p0, k0);
                push(p1, k1);
            }
            TACC(ts);
            if (qn > 0) exact_round(qn);
            TACC(tx);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (grid_ray) ok = ((wocc[lane >> 5] >> (lane & 31)) & 1u) == 0u;
            __builtin_amdgcn_wave_barrier();

            if (ok) {
                word |= 1u << (s & 31);
                if (NCM == 1) {
                    // visibility only
                } else if (a.mode_intensity) {
                    inten += (double)val / (double)spt;
                } else if (bin >= 0 && bin < nbins) {
                    double cc = (double)val / (double)spt;
                    if (rows_in_lds) unsafeAtomicAdd(&s_row[bin], cc);
                    else unsafeAtomicAdd(&grow[bin], cc);
                }
            }
            if ((s & 31) == 31 || s == spt - 1) {
                if (visp) {
                    if (!TILED) visp[(size_t)(s >> 5) * F] = word;
                    else if (word) atomicOr(&visp[(size_t)(s >> 5) * F], word);   // a face may straddle tiles
                }
                word = 0;
            }
            TACC(th);
        }
        if (a.mode_intensity && has_face && inten != 0.0) unsafeAtomicAdd(&a.intensity[f.fid], inten);
    }
#ifdef NLOS_FWD_STAMPS
    if (a.dbg) {   // diagnostic build only: work counters -> a.dbg[8..12]
        atomicAdd((unsigned long long*)&a.dbg[8], c_rays);
        atomicAdd((unsigned long long*)&a.dbg[9], c_pairs);
        atomicAdd((unsigned long long*)&a.dbg[10], c_iters);
        atomicAdd((unsigned long long*)&a.dbg[11], c_mt);
        atomicAdd((unsigned long long*)&a.dbg[12], c_mtw);
        if (tid == 0) atomicAdd((unsigned long long*)&a.dbg[13], (unsigned long long)s_ctl[2]);
        if (lane == 0) {
            atomicAdd((unsigned long long*)&a.dbg[14], (unsigned long long)tg);
            atomicAdd((unsigned long long*)&a.dbg[15], (unsigned long long)ts);
            atomicAdd((unsigned long long*)&a.dbg[16], (unsigned long long)tx);
            atomicAdd((unsigned long long*)&a.dbg[17], (unsigned long long)th);
        }
    }
#endif
    FWD_STAMP();   // 5: sample + trace + histogram
    if (rows_in_lds && grow) {
        __syncthreads();
        if (!TILED) {
            for (int i = tid; i < nbins; i += NT) grow[i] = s_row[i];
        } else {
            for (int i = tid; i < nbins; i += NT)
                if (s_row[i] != 0.0) unsafeAtomicAdd(&grow[i], s_row[i]);     // one partial row per tile
        }
    }
}

// --------------------------------------------------------------------- smooth
__global__ __launch_bounds__(256) void k_smooth(SmoothArgs a) {
    const int l = blockIdx.x;
    const int R = a.refine, K = a.K, half = a.offset;
    const int rb = a.T * R;
    const double* fine = a.fine + (size_t)l * rb;
    for (int t = threadIdx.x; t < a.T; t += blockDim.x) {
        double acc = 0.0;
        for (int q = 0; q < R; ++q) {
            // y[b + half] with y = full convolution of fine (*) kernel
            int c = t * R + q + half;
            double y = 0.0;
            for (int j = 0; j < K; ++j) {
                int i = c - j;
                if (i >= 0 && i < rb) y += fine[i] * a.kernel[j];
            }
            acc += y;
        }
        a.transient[(size_t)l * a.T + t] = acc;
    }
}

// ------------------------------------------------------------------- residual
__global__ __launch_bounds__(256) void k_residual(ResidualArgs a) {
    const size_t n = (size_t)a.L * a.T;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double d = a.data[i] - a.transient[i];
        if (a.loss_test == 1) d = 2 * d * d * d;
        if (a.weight) d = d * a.weight[i];
        a.diff[i] = d;
    }
    if (a.pathlengths && blockIdx.x == 0)
        for (int i = threadIdx.x; i < a.T; i += blockDim.x) a.pathlengths[i] = (double)(a.lb + i * a.res);
}

__global__ __launch_bounds__(256) void k_boxfilter(double* diff, int T, int w) {
    extern __shared__ double s_buf[];   // 2*T
    double* x = s_buf;
    double* y = s_buf + T;
    double* row = diff + (size_t)blockIdx.x * T;
    const double k = 1.0 / ((double)2 * w + 1);
    for (int i = threadIdx.x; i < T; i += blockDim.x) x[i] = row[i];
    __syncthreads();
    for (int i = threadIdx.x; i < T; i += blockDim.x) {
        double s = 0;
        for (int j = -w; j <= w; ++j) { int q = i + j; if (q >= 0 && q < T) s += x[q] * k; }
        y[i] = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < T; i += blockDim.x) {
        double s = 0;
        for (int j = -w; j <= w; ++j) { int q = i + j; if (q >= 0 && q < T) s += y[q] * k; }
        row[i] = s;
    }
}

// ------------------------------------------------------------------- gradient
// Per accepted sample: vectors t1, t2 and the intensity
// (smoothed_transient/transient_and_gradient.cpp:944-966, ggx/...:750-783).
struct GVec { V3 t1, t2; float inten_f; };

template <int FEAT>
__device__ __forceinline__ void grad_vectors(const Face& f, const Geo& g, V3 on, int normal_term, int v1_style,
                                             float alpha, GVec& out) {
    float c2 = dot(on, g.dir);
    float c3 = dot(g.n, -g.dir);
    if (c2 < 0) c2 = 0;
    if (c3 < 0) c3 = 0;
    float ff = c2 * c3 / g.h / g.h;
    float h2 = g.h * g.h, h4 = h2 * h2, h5 = h4 * g.h;
    V3 inner = ((on * c3) - (g.n * c2)) + ((((-g.dir) * 4.0f) * c2) * c3);
    V3 t1, gn = mk(0, 0, 0);
    if (FEAT & FEAT_GGX) {
        V3 wv = -g.dir;
        float nw = dot(g.n, wv);
        float brdf = ggx_eval(alpha, nw);
        float s = ggx_eval_nwsdiff(alpha, nw);
        V3 dn = wv * s, dw = g.n * s;
        V3 dx = (-dw) + ((g.dir * dot(g.dir, dw)) * (1.0f / g.h));
        out.inten_f = (float)(double)(g.alb * ff * ff * brdf);
        V3 t11 = inner * (2 * c2 * c3);
        t11 = t11 * (1.0f / h5);
        t11 = t11 * brdf;
        t1 = t11 + dx * (ff * ff);
        if (normal_term) {
            gn = ((((g.dir * -2.0f) * c3) * c2) * c2) * brdf;
            gn = gn * (1.0f / h4);
            gn = gn + dn * (ff * ff);
            float ct = dot(gn, g.n);
            gn = gn - g.n * ct;
        }
    } else {
        out.inten_f = g.alb * ff * ff;
        float sc = v1_style ? (2 * c2 * c3) : (2 * g.alb * c2 * c3);
        t1 = inner * sc;
        t1 = t1 * (1.0f / h5);
        if (normal_term) {
            float s0 = v1_style ? -2.0f : (-2 * g.alb);
            gn = (((g.dir * s0) * c3) * c2) * c2;
            gn = gn * (1.0f / h4);
            float ct = dot(gn, g.n);
            gn = gn - g.n * ct;
        }
    }
    V3 t2 = g.n * out.inten_f;
    t2 = (t2 + gn) * (1.0f / (2 * f.area));
    out.t1 = t1;
    out.t2 = t2;
}

// Row N: t1 = alb (ff_b grad ff_a + ff_a grad ff_b), grad ff = (n_o c3 - n c2 - 4 dir c2 c3) / d^3;
// normal term dI/dn projected as in the confocal rows
template <int FEAT>
__device__ __forceinline__ void grad_vectors_nc(const Face& f, const GeoNC& g, V3 na, V3 nb, int normal_term, GVec& out) {
    float c2a = dot(na, g.dirA), c3a = dot(g.n, -g.dirA);
    float c2b = dot(nb, g.dirB), c3b = dot(g.n, -g.dirB);
    if (c2a < 0) c2a = 0;
    if (c3a < 0) c3a = 0;
    if (c2b < 0) c2b = 0;
    if (c3b < 0) c3b = 0;
    const float ffa = c2a * c3a / g.d1 / g.d1, ffb = c2b * c3b / g.d2 / g.d2;
    const V3 ia = ((na * c3a) - (g.n * c2a)) + ((((-g.dirA) * 4.0f) * c2a) * c3a);
    const V3 ib = ((nb * c3b) - (g.n * c2b)) + ((((-g.dirB) * 4.0f) * c2b) * c3b);
    const V3 ga = ia * (1.0f / ((g.d1 * g.d1) * g.d1));
    const V3 gb = ib * (1.0f / ((g.d2 * g.d2) * g.d2));
    out.inten_f = g.alb * ffa * ffb;
    out.t1 = ((ga * ffb) + (gb * ffa)) * g.alb;
    V3 gn = mk(0, 0, 0);
    if (normal_term) {
        gn = (g.dirA * c3b) + (g.dirB * c3a);
        gn = gn * (-(g.alb * c2a * c2b));
        gn = gn * (1.0f / ((g.d1 * g.d1) * (g.d2 * g.d2)));
        float ct = dot(gn, g.n);
        gn = gn - g.n * ct;
    }
    V3 t2 = g.n * out.inten_f;
    out.t2 = (t2 + gn) * (1.0f / (2 * f.area));
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
    int K, two_rs;
    double r_over_res;     // refine / res
};

__device__ __forceinline__ void grouped_taps(const TapTables& tt, const double* __restrict__ s_diff, int T,
                                             double twoh, double lbd, double resd, double inv_res,
                                             double& s0, double& s1) {
    s0 = 0.0;
    s1 = 0.0;
    const int K = tt.K;
    const int b_first = tap_bin(twoh, tt.delta[0], lbd, resd, inv_res);
    const int b_last = tap_bin(twoh, tt.delta[K - 1], lbd, resd, inv_res);
    int i_start = 0;
    for (int b = b_first; b <= b_last; ++b) {
        int ie = K;
        if (b < b_last) {
            // first tap whose bin exceeds b: delta_i >= (b+1)*res + lb - 2h
            const double thr = ((double)(b + 1) * resd + lbd) - twoh;
            const double y = thr * tt.r_over_res;          // boundary in (continuous) tap index, minus two_rs
            const double yc = ceil(y);
            int ic = (int)yc + tt.two_rs;
            ic = max(i_start, min(K, ic));
            // delta_i carries the fp32 rounding of (i - two_rs) * res / refine (<= ~1e-6 taps): the closed
            // form is the reference's per-tap assignment unless the boundary is that close to a tap
            if (yc - y < 1e-4 || y - (yc - 1.0) < 1e-4) {
                while (ic > i_start && tap_bin(twoh, tt.delta[ic - 1], lbd, resd, inv_res) > b) --ic;
                while (ic < K && tap_bin(twoh, tt.delta[ic], lbd, resd, inv_res) <= b) ++ic;
            }
            ie = ic;
        }
        if (b >= 0 && b < T && ie > i_start) {
            double dd = (double)(float)((-2) * s_diff[b]);
            s0 += dd * (tt.p0[ie] - tt.p0[i_start]);
            s1 += dd * (tt.p1[ie] - tt.p1[i_start]);
        }
        i_start = ie;
    }
}

// MODE 0: per-vertex gradient [V,3]; 1: scalar d/d albedo; 2: scalar d/d alpha (GGX);
//      3: single-vertex per-bin gradient [T,3]
// two 512-thread workgroups per CU (LDS: ~76 KB each) need <= 128 VGPRs
#ifndef NLOS_GRAD_NT
#define NLOS_GRAD_NT 512
#define NLOS_GRAD_WPS 4
#endif
template <int FEAT, int MODE, bool NC = false>
__global__ __launch_bounds__(NLOS_GRAD_NT, NLOS_GRAD_WPS) void k_gradient(GradientArgs a) {
    extern __shared__ double s_mem[];       // [ticket (8 B)][diff row T][tap tables 3K+2][grad 3V][masks][bases][live]
    int* s_next = reinterpret_cast<int*>(s_mem);
    const int T = a.sp.nbins;
    const int K = a.K;
    const int F = a.sc.F, V = a.sc.V;
    const int nblocks = (F + 63) >> 6;
    double* s_diff = s_mem + 1;             // [T]
    double* s_delta = s_diff + T;           // [K]
    double* s_p0 = s_delta + K;             // [K+1]
    double* s_p1 = s_p0 + K + 1;            // [K+1]
    double* s_grad = s_p1 + K + 1;          // [3V] when lds_grad
    unsigned long long* s_mask = reinterpret_cast<unsigned long long*>(s_grad + (((MODE == 0 || MODE == 4) && a.lds_grad) ? 3 * V : 0));
    uint32_t* s_base = reinterpret_cast<uint32_t*>(s_mask + nblocks);              // [nblocks+1]
    uint16_t* s_live = reinterpret_cast<uint16_t*>(s_base + ((nblocks + 2) & ~1));   // [F] sorted face slots (compact only)
    const int spt = a.sp.spt;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int Ltot = a.src.total_sources > 0 ? a.src.total_sources : a.src.L;
    const double lbd = (double)a.sp.lb, resd = (double)a.sp.res, inv_res = 1.0 / resd;

    if (MODE == 4) {
        // jitter taps: s_delta <- (float) jitter_weight, s_p0 <- jitter_grad
        for (int i = threadIdx.x; i < K; i += blockDim.x) { s_delta[i] = (double)(float)a.tap_w[i]; s_p0[i] = a.tap_g[i]; }
    } else {
        for (int i = threadIdx.x; i < K; i += blockDim.x) s_delta[i] = a.tap_delta[i];
        for (int i = threadIdx.x; i <= K; i += blockDim.x) { s_p0[i] = a.tap_p0[i]; s_p1[i] = a.tap_p1[i]; }
    }
    if ((MODE == 0 || MODE == 4) && a.lds_grad)
        for (int i = threadIdx.x; i < 3 * V; i += blockDim.x) s_grad[i] = 0.0;
    double scalar_acc = 0.0;
    TapTables tt;
    tt.delta = s_delta; tt.p0 = s_p0; tt.p1 = s_p1; tt.K = K; tt.two_rs = a.two_rs; tt.r_over_res = a.r_over_res;

    for (int l = blockIdx.x; l < a.src.L; l += gridDim.x) {
        __syncthreads();                    // previous source done with s_diff
        for (int i = threadIdx.x; i < T; i += blockDim.x) s_diff[i] = a.diff[(size_t)l * T + i];
        if (threadIdx.x == 0) *s_next = 0;
        // faces with at least one accepted sample, compacted in order (pass 1 left the masks)
        for (int b = wave; b < nblocks; b += nwaves) {
            const int j = (b << 6) + lane;
            uint32_t any = 0;
            if (j < F) {
                const uint32_t* visp = a.vis + ((size_t)l * a.vis_words) * F + j;
                for (int wi = 0; wi < a.vis_words; ++wi) any |= visp[(size_t)wi * F];
            }
            const unsigned long long m = __ballot(any != 0u);
            if (lane == 0) s_mask[b] = m;
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            // wave 0: exclusive scan of the per-block counts
            uint32_t run = 0;
            for (int b0 = 0; b0 < nblocks; b0 += 64) {
                const int b = b0 + lane;
                uint32_t n = b < nblocks ? (uint32_t)__popcll(s_mask[b]) : 0u;
                uint32_t incl = n;
                for (int off = 1; off < 64; off <<= 1) {
                    uint32_t v = __shfl_up(incl, off);
                    if (lane >= off) incl += v;
                }
                if (b < nblocks) s_base[b] = run + incl - n;
                run += __shfl(incl, 63);
            }
            if (lane == 0) s_base[nblocks] = run;
        }
        __syncthreads();
        for (int b = wave; b < nblocks; b += nwaves) {
            const unsigned long long m = s_mask[b];
            if (a.compact && ((m >> lane) & 1ull))
                s_live[s_base[b] + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)((b << 6) + lane);
        }
        __syncthreads();
        const int n_live = a.compact ? (int)s_base[nblocks] : F;
        const int live_blocks = (n_live + 63) >> 6;
        const V3 o = ld3(a.src.origin + 3 * (size_t)l);
        const V3 on = ld3(a.src.normal + 3 * (size_t)l);
        const V3 ob = NC ? ld3(a.src.sensor + 3 * (size_t)l) : o;
        const V3 onb = NC ? ld3(a.src.sensor_normal + 3 * (size_t)l) : on;
        const uint64_t lg = (uint64_t)(a.src.source_offset + l);

        for (;;) {
            const int b = wave_ticket(s_next);
            if (b >= live_blocks) break;
            const int li = (b << 6) + lane;
            if (li >= n_live) continue;
            const int j = a.compact ? (int)s_live[li] : li;
            const uint32_t* visp = a.vis + ((size_t)l * a.vis_words) * F + j;
            const Face f = load_face(a.sc.facerec, j);
            if (MODE == 3 && f.i0 != a.vertex_num && f.i1 != a.vertex_num && f.i2 != a.vertex_num) continue;
            const Tri tr = load_tri(a.sc.tris, j);
            const uint64_t kbase = (lg * (uint64_t)F + (uint64_t)f.fid) * (uint64_t)spt;
            double acc[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) acc[q] = 0.0;
            double sacc = 0.0;

            for (int wi = 0; wi < a.vis_words; ++wi) {
                uint32_t word = visp[(size_t)wi * F];
                while (word) {
                    const int bit = __ffs(word) - 1;
                    word &= word - 1;
                    const int s = (wi << 5) + bit;
                    if (NC) {
                        // row N: two legs, d(d1 + d2)/dp = dirA + dirB; P1 carries the confocal factor 2
                        GeoNC gc;
                        float tA, tB;
                        if (!sample_geo_nc<FEAT>(f, tr, o, ob, a.sp.seed, kbase + (uint64_t)s, a.sp.lb, a.sp.ub,
                                                 a.sc.vertex_normal, a.sc.albedo, gc, tA, tB))
                            continue;
                        GVec gv;
                        grad_vectors_nc<FEAT>(f, gc, on, onb, a.normal_term, gv);
                        const V3 e0 = f.p2 - f.p1, e1 = f.p0 - f.p2, e2 = f.p1 - f.p0;
                        const V3 ce[3] = {cross(gv.t2, e0), cross(gv.t2, e1), cross(gv.t2, e2)};
                        double s0, s1;
                        grouped_taps(tt, s_diff, T, (double)(gc.d1 + gc.d2), lbd, resd, inv_res, s0, s1);
                        const V3 di = ((gc.dirA + gc.dirB) * 0.5f) * gv.inten_f;
                        const float bw[3] = {gc.u, gc.v, gc.w};
#pragma unroll
                        for (int q = 0; q < 3; ++q) {
                            V3 A1 = gv.t1 * bw[q] + ce[q];
                            V3 A2 = di * bw[q];
                            acc[3 * q + 0] += (double)A1.x * s0 + (double)A2.x * s1;
                            acc[3 * q + 1] += (double)A1.y * s0 + (double)A2.y * s1;
                            acc[3 * q + 2] += (double)A1.z * s0 + (double)A2.z * s1;
                        }
                        continue;
                    }
                    Geo g;
                    float t_self;
                    if (!sample_geo<FEAT>(f, tr, o, a.sp.seed, kbase + (uint64_t)s, a.sp.lb, a.sp.ub,
                                          a.sc.vertex_normal, a.sc.albedo, g, t_self))
                        continue;   // cannot happen: pass 1 accepted this sample with the same arithmetic
                    const double twoh = (double)(2.0f * g.h);
                    if (MODE == 1 || MODE == 2) {
                        // rows A / GGX alpha: scalar gradients (literal tap loop, double weights)
                        float c2 = dot(on, g.dir);
                        float c3 = dot(g.n, -g.dir);
                        if (c2 < 0) c2 = 0;
                        if (c3 < 0) c3 = 0;
                        float ff = c2 * c3 / g.h / g.h;
                        double g0;
                        if (MODE == 2) g0 = (double)(g.alb * ff * ff * ggx_eval_adiff(a.sp.ggx_alpha, dot(g.n, -g.dir)));
                        else g0 = (double)(ff * ff);
                        double s0 = 0.0;
                        for (int i = 0; i < K; ++i) {
                            int bin = tap_bin(twoh, s_delta[i], lbd, resd, inv_res);
                            if (bin >= 0 && bin < T) s0 += a.tap_w[i] * (-2) * s_diff[bin];
                        }
                        sacc += (double)f.area * g0 * s0 / (double)spt;
                        continue;
                    }
                    GVec gv;
                    grad_vectors<FEAT>(f, g, on, a.normal_term, a.v1_style, a.sp.ggx_alpha, gv);
                    const V3 e0 = f.p2 - f.p1, e1 = f.p0 - f.p2, e2 = f.p1 - f.p0;
                    const V3 ce0 = cross(gv.t2, e0), ce1 = cross(gv.t2, e1), ce2 = cross(gv.t2, e2);
                    if (MODE == 3) {
                        // single-vertex per-bin gradient: output indexed by the tap's bin
                        V3 ce; float bw;
                        if (a.vertex_num == f.i0) { ce = ce0; bw = g.u; }
                        else if (a.vertex_num == f.i1) { ce = ce1; bw = g.v; }
                        else { ce = ce2; bw = g.w; }
                        for (int i = 0; i < K; ++i) {
                            int bin = tap_bin(twoh, s_delta[i], lbd, resd, inv_res);
                            if (bin < 0 || bin >= T) continue;
                            V3 gg = g.dir * (float)a.tap_g[i];
                            V3 q = ((gv.t1 + gg * gv.inten_f) * bw + ce) * (float)a.tap_w[i];
                            double sc = 1.0 / ((double)spt * (double)Ltot);
                            unsafeAtomicAdd(&a.out[3 * bin + 0], (double)(f.area * q.x) * sc);
                            unsafeAtomicAdd(&a.out[3 * bin + 1], (double)(f.area * q.y) * sc);
                            unsafeAtomicAdd(&a.out[3 * bin + 2], (double)(f.area * q.z) * sc);
                        }
                        continue;
                    }
                    // MODE 0: the K-tap loop factors into two scalar sums per sample:
                    //   sum_i (t1*b + t2 x e) w_i d_i  +  b * I * dir * sum_i g_i w_i d_i
                    double s0, s1;
                    V3 di;
                    if (MODE == 4) {
                        // jitter/transient_and_gradient.cpp:944-969: tap i -> bin b0 + (i - offset),
                        //   g = (t1 w_i + jitter_grad_i * I * (-2) * dir / res) * b + (t2 x e) w_i
                        const int b0 = (int)floorf((2.0f * g.h - a.sp.lb) / a.sp.res) - a.two_rs;
                        const double m2i = (double)gv.inten_f * (-2);
                        s0 = 0.0;
                        s1 = 0.0;
                        const int i0 = max(0, -b0), i1 = min(K, T - b0);
                        for (int i = i0; i < i1; ++i) {
                            const float dd = (float)((-2) * s_diff[b0 + i]);
                            s0 += (double)((float)s_delta[i] * dd);
                            s1 += (double)(((float)(s_p0[i] * m2i) / a.sp.res) * dd);
                        }
                        di = g.dir;
                    } else {
                        grouped_taps(tt, s_diff, T, twoh, lbd, resd, inv_res, s0, s1);
                        di = g.dir * gv.inten_f;
                    }
                    const float bw[3] = {g.u, g.v, g.w};
                    const V3 ce[3] = {ce0, ce1, ce2};
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        V3 A1 = gv.t1 * bw[q] + ce[q];
                        V3 A2 = di * bw[q];
                        acc[3 * q + 0] += (double)A1.x * s0 + (double)A2.x * s1;
                        acc[3 * q + 1] += (double)A1.y * s0 + (double)A2.y * s1;
                        acc[3 * q + 2] += (double)A1.z * s0 + (double)A2.z * s1;
                    }
                }
            }
            if (MODE == 0 || MODE == 4) {
                const double sc = (double)f.area / (double)spt;
                const int vi[3] = {f.i0, f.i1, f.i2};
#pragma unroll
                for (int q = 0; q < 3; ++q) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        double val = acc[3 * q + c] * sc;
                        if (a.lds_grad) unsafeAtomicAdd(&s_grad[3 * vi[q] + c], val);
                        else unsafeAtomicAdd(&a.out[3 * (size_t)vi[q] + c], val / (double)Ltot);
                    }
                }
            } else if (MODE == 1 || MODE == 2) {
                scalar_acc += sacc;
            }
        }
    }
    __syncthreads();
    if ((MODE == 0 || MODE == 4) && a.lds_grad) {
        const double invL = 1.0 / (double)Ltot;
        for (int i = threadIdx.x; i < 3 * V; i += blockDim.x) {
            double v = s_grad[i];
            if (v != 0.0) unsafeAtomicAdd(&a.out[i], v * invL);
        }
    }
    if (MODE == 1 || MODE == 2) {
        for (int off = 32; off > 0; off >>= 1) scalar_acc += __shfl_down(scalar_acc, off);
        if (lane == 0 && scalar_acc != 0.0) unsafeAtomicAdd(&a.out[0], scalar_acc / (double)Ltot);
    }
}

// ------------------------------------------------- gradient, large meshes (face-major)
// When 3V doubles do not fit LDS, k_gradient above falls back to nine global atomics per (source, face)
// pair -- 74 M of them per step at F = 20 k, executed memory-side across the eight XCDs (4.7 ms).  This
// variant turns the loop nest around: a workgroup owns a CHUNK of kFmChunk consecutive (Morton-sorted)
// faces and a group of sources, keeps the nine sums of every face of the chunk in LDS across all its
// sources (ds_add_f64, one writer lane per item), and touches the vertex gradient only once per face at
// the end.  Per batch of kFmBatch sources it loads the residual rows, compacts the (source, face) items
// with accepted samples into an LDS list (ballot + one LDS counter) and hands them to the lanes densely.
// Same per-sample arithmetic as k_gradient<FEAT, 0>; only the fp64 summation order differs.
constexpr int kFmChunk = 512, kFmBatch = 4, kFmThreads = 256;

template <int FEAT>
__global__ __launch_bounds__(kFmThreads) void k_gradient_fm(GradientArgs a, int src_per_group) {
    extern __shared__ double s_fm[];      // [acc 9*CHUNK][rows BATCH*T][delta K][p0 K+1][p1 K+1][list BATCH*CHUNK u16][ctl]
    const int T = a.sp.nbins, K = a.K, F = a.sc.F;
    double* s_acc = s_fm;
    double* s_rows = s_acc + 9 * kFmChunk;
    double* s_delta = s_rows + kFmBatch * T;
    double* s_p0 = s_delta + K;
    double* s_p1 = s_p0 + K + 1;
    uint16_t* s_list = reinterpret_cast<uint16_t*>(s_p1 + K + 1);
    int* s_cnt = reinterpret_cast<int*>(s_list + kFmBatch * kFmChunk);
    const int tid = threadIdx.x, lane = tid & 63;
    const int f0 = blockIdx.x * kFmChunk, nf = min(kFmChunk, F - f0);
    const int l0 = blockIdx.y * src_per_group, l1 = min(l0 + src_per_group, a.src.L);
    const int spt = a.sp.spt;
    const int Ltot = a.src.total_sources > 0 ? a.src.total_sources : a.src.L;
    const double lbd = (double)a.sp.lb, resd = (double)a.sp.res, inv_res = 1.0 / resd;
    for (int i = tid; i < 9 * kFmChunk; i += kFmThreads) s_acc[i] = 0.0;
    for (int i = tid; i < K; i += kFmThreads) s_delta[i] = a.tap_delta[i];
    for (int i = tid; i <= K; i += kFmThreads) { s_p0[i] = a.tap_p0[i]; s_p1[i] = a.tap_p1[i]; }
    TapTables tt;
    tt.delta = s_delta; tt.p0 = s_p0; tt.p1 = s_p1; tt.K = K; tt.two_rs = a.two_rs; tt.r_over_res = a.r_over_res;

    for (int lb0 = l0; lb0 < l1; lb0 += kFmBatch) {
        const int nb = min(kFmBatch, l1 - lb0);
        __syncthreads();                                   // previous batch done with rows / list
        for (int i = tid; i < nb * T; i += kFmThreads) s_rows[i] = a.diff[(size_t)lb0 * T + i];
        if (tid == 0) *s_cnt = 0;
        __syncthreads();
        // (source, face) items of this batch with at least one accepted sample
        for (int it = tid; it < nb * kFmChunk; it += kFmThreads) {
            const int bl = it / kFmChunk, jl = it - bl * kFmChunk;
            uint32_t any = 0;
            if (jl < nf) {
                const uint32_t* visp = a.vis + ((size_t)(lb0 + bl) * a.vis_words) * F + f0 + jl;
                for (int wi = 0; wi < a.vis_words; ++wi) any |= visp[(size_t)wi * F];
            }
            const unsigned long long m = __ballot(any != 0u);
            int base = 0;
            if (lane == 0 && m) base = atomicAdd(s_cnt, __popcll(m));
            base = __shfl(base, 0);
            if (any) s_list[base + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)((bl << 12) | jl);
        }
        __syncthreads();
        const int n_items = *s_cnt;
        for (int it = tid; it < n_items; it += kFmThreads) {
            const int code = s_list[it];
            const int bl = code >> 12, jl = code & 0xFFF;
            const int l = lb0 + bl, j = f0 + jl;
            const double* s_diff = s_rows + bl * T;
            const Face f = load_face(a.sc.facerec, j);
            const Tri tr = load_tri(a.sc.tris, j);
            const V3 o = ld3(a.src.origin + 3 * (size_t)l);
            const V3 on = ld3(a.src.normal + 3 * (size_t)l);
            const uint64_t kbase = ((uint64_t)(a.src.source_offset + l) * (uint64_t)F + (uint64_t)f.fid) * (uint64_t)spt;
            const uint32_t* visp = a.vis + ((size_t)l * a.vis_words) * F + j;
            double acc[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) acc[q] = 0.0;
            for (int wi = 0; wi < a.vis_words; ++wi) {
                uint32_t word = visp[(size_t)wi * F];
                while (word) {
                    const int bit = __ffs(word) - 1;
                    word &= word - 1;
                    Geo g;
                    float t_self;
                    if (!sample_geo<FEAT>(f, tr, o, a.sp.seed, kbase + (uint64_t)((wi << 5) + bit), a.sp.lb, a.sp.ub,
                                          a.sc.vertex_normal, a.sc.albedo, g, t_self))
                        continue;
                    GVec gv;
                    grad_vectors<FEAT>(f, g, on, a.normal_term, a.v1_style, a.sp.ggx_alpha, gv);
                    const V3 e0 = f.p2 - f.p1, e1 = f.p0 - f.p2, e2 = f.p1 - f.p0;
                    const V3 ce[3] = {cross(gv.t2, e0), cross(gv.t2, e1), cross(gv.t2, e2)};
                    double s0, s1;
                    grouped_taps(tt, s_diff, T, (double)(2.0f * g.h), lbd, resd, inv_res, s0, s1);
                    const V3 di = g.dir * gv.inten_f;
                    const float bw[3] = {g.u, g.v, g.w};
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const V3 A1 = gv.t1 * bw[q] + ce[q];
                        const V3 A2 = di * bw[q];
                        acc[3 * q + 0] += (double)A1.x * s0 + (double)A2.x * s1;
                        acc[3 * q + 1] += (double)A1.y * s0 + (double)A2.y * s1;
                        acc[3 * q + 2] += (double)A1.z * s0 + (double)A2.z * s1;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 9; ++q) unsafeAtomicAdd(&s_acc[9 * jl + q], acc[q]);
        }
    }
    __syncthreads();
    // one pass over the chunk: scale and scatter to the vertices
    for (int jl = tid; jl < nf; jl += kFmThreads) {
        const Face f = load_face(a.sc.facerec, f0 + jl);
        if (f.degenerate) continue;
        const double sc = (double)f.area / (double)spt / (double)Ltot;
        const int vi[3] = {f.i0, f.i1, f.i2};
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double val = s_acc[9 * jl + 3 * q + c];
                if (val != 0.0) unsafeAtomicAdd(&a.out[3 * (size_t)vi[q] + c], val * sc);
            }
    }
}

template <int FEAT>
bool gradient_fm_launch(const GradientArgs& a, hipStream_t stream) {
    // only the plain vertex gradient of meshes whose 3V accumulator cannot live in LDS
    if (a.mode != 0 || a.src.sensor || a.sp.nbins * kFmBatch > 8192) return false;
    const size_t lds = ((size_t)9 * kFmChunk + (size_t)kFmBatch * a.sp.nbins + 3 * (size_t)a.K + 2) * sizeof(double) +
                       (size_t)kFmBatch * kFmChunk * 2 + 16;
    if (lds > 80 * 1024) return false;
    const int nchunks = (a.sc.F + kFmChunk - 1) / kFmChunk;
    // enough workgroups to fill the chip (256 CUs x 2), sources in multiples of the batch
    int groups = (1024 + nchunks - 1) / nchunks;
    int per = (a.src.L + groups - 1) / groups;
    per = ((per + kFmBatch - 1) / kFmBatch) * kFmBatch;
    if (per < kFmBatch) per = kFmBatch;
    groups = (a.src.L + per - 1) / per;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gradient_fm<FEAT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gradient_fm<FEAT>), dim3(nchunks, groups), dim3(kFmThreads), lds, stream, a, per);
    return true;
}

// ------------------------------------------------------------------ intersect
__global__ __launch_bounds__(256) void k_intersect(IntersectArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    V3 o = ld3(a.origins + 3 * (size_t)i);
    V3 d = ld3(a.dirs + 3 * (size_t)i);
    float t, u, v;
    int best = closest_hit(a.sc.nodes, a.sc.n_nodes, a.sc.tris, a.sc.face_id, o, d, t, u, v);
    if (a.out3) {
        if (best < 0) {
            a.out3[3 * (size_t)i] = -1.0f;      // u, v untouched (c_embree_intersector.cpp:39-45)
        } else {
            a.out3[3 * (size_t)i] = (float)a.sc.face_id[best];
            a.out3[3 * (size_t)i + 1] = u;
            a.out3[3 * (size_t)i + 2] = v;
        }
    }
    if (a.out1) a.out1[i] = best < 0 ? -1.0f : (float)a.sc.face_id[best];
}

__global__ __launch_bounds__(256) void k_bary_to_world(const float* V, const int32_t* F, const float* bary, int n,
                                                       float* out) {
    // embree_intersector/c_embree_intersector.cpp:75-92
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int fid = (int)bary[3 * (size_t)i];
    if (fid < 0) return;
    float u = bary[3 * (size_t)i + 1], v = bary[3 * (size_t)i + 2];
    int a = F[3 * fid], b = F[3 * fid + 1], c = F[3 * fid + 2];
    for (int k = 0; k < 3; ++k)
        out[3 * (size_t)i + k] = (1 - u - v) * V[3 * a + k] + u * V[3 * b + k] + v * V[3 * c + k];
}

__global__ __launch_bounds__(256) void k_zero_f64(double* p, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = 0.0;
}

// LDS budget of the grid kernel: two 512-thread workgroups per CU (160 KiB / 2, minus slack)
constexpr size_t kGridLdsBudget = 78 * 1024;

template <int FEAT, int NCM = 0>
bool forward_grid_launch(const ForwardArgs& a, int rows_in_lds, hipStream_t stream) {
    if (a.force_bvh || a.tile_list || a.sc.F > 8191 || a.sc.F < 64) return false;     // 13-bit triangle index in the cell entries
    int R = (int)lrintf(sqrtf(0.5f * (float)a.sc.F));
    R = std::min(std::max(R, 8), 96);
    const size_t nblk = ((size_t)a.sc.F + 63) / 64;
    const size_t R2 = ((size_t)R + 1) / 2;
    size_t union_words = ((R2 * R2 + 1) & ~(size_t)1) + 2 * nblk;
    if (union_words < 8 * 130) union_words = 8 * 130;
    const size_t fixed = 32 + (rows_in_lds ? (size_t)a.sp.nbins * sizeof(double) : 0) + (((size_t)R * R + 2) & ~(size_t)1) * 4 +
                         ((union_words + 1) & ~(size_t)1) * 4;
    if (fixed + 4 * 2 * (size_t)a.sc.F > kGridLdsBudget || !a.live) return false;      // want room for >= 2 entries per face
    size_t cap = (kGridLdsBudget - fixed) / 4;
    const size_t lds = fixed + cap * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_forward_grid<FEAT, NCM>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    // one slot of slack: the walk reads entries in pairs and may touch the slot after the last list
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_forward_grid<FEAT, NCM>), dim3(a.src.L), dim3(512), lds, stream, a, rows_in_lds, R,
                       (int)cap - 1);
    return true;
}

// meshes beyond one workgroup's LDS: one workgroup per (source, slope-space tile)
template <int FEAT>
bool forward_tiled_launch(const ForwardArgs& a, int rows_in_lds, hipStream_t stream) {
    if (a.force_bvh || !a.tile_list || !a.tile_count || !a.live || a.tiles_x * a.tiles_y > 1024 || a.tiles_x < 1 || a.tiles_y < 1 || a.mode_intensity || a.src.sensor) return false;
    const int R = 32;
    const size_t R2 = (R + 1) / 2;
    const size_t mask_blocks = std::max(((size_t)a.tile_cap + 63) / 64, ((size_t)a.sc.F + 63) / 64);
    size_t union_words = ((R2 * R2 + 1) & ~(size_t)1) + 2 * mask_blocks;
    if (union_words < 8 * 130) union_words = 8 * 130;
    const size_t fixed = 32 + (rows_in_lds ? (size_t)a.sp.nbins * sizeof(double) : 0) + (((size_t)R * R + 2) & ~(size_t)1) * 4 +
                         ((union_words + 1) & ~(size_t)1) * 4;
    if (fixed + 4 * 4096 > kGridLdsBudget) return false;
    const size_t cap = (kGridLdsBudget - fixed) / 4;
    const size_t lds = fixed + cap * 4;
    // partial rows / visibility words of the tiles are combined with atomics: start from zero
    if (rows_in_lds && a.rows) launch_zero_f64(a.rows, (size_t)a.src.L * a.sp.nbins, stream);
    if (a.vis) (void)hipMemsetAsync(a.vis, 0, sizeof(uint32_t) * (size_t)a.src.L * a.vis_words * a.sc.F, stream);
    const size_t nwg = (size_t)a.src.L * a.tiles_x * a.tiles_y;
    (void)hipMemsetAsync(a.tile_count, 0, sizeof(int) * 2 * nwg, stream);      // subset sizes + retry flags
    hipLaunchKernelGGL(k_tile_bin, dim3(a.src.L), dim3(512), 0, stream, a, R);
    // second launch for the tiles whose cell lists overflow: the whole CU's LDS for one workgroup
    const size_t lds_big = 150 * 1024;
    const size_t cap_big = (lds_big - fixed) / 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_forward_grid<FEAT, 0, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_big);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_forward_grid<FEAT, 0, true>), dim3((unsigned)nwg), dim3(512), lds, stream, a, rows_in_lds, R,
                       (int)cap - 1, 0);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_forward_grid<FEAT, 0, true>), dim3((unsigned)nwg), dim3(512), lds_big, stream, a, rows_in_lds,
                       R, (int)cap_big - 1, 1);
    return true;
}

template <int FEAT>
void forward_launch(const ForwardArgs& a, int rows_in_lds, size_t lds, hipStream_t stream) {
    if (a.src.sensor) {
        if constexpr ((FEAT & FEAT_GGX) == 0) {
            // row N: one grid pass per end point of the pair (sensor-leg visibility bits first, then the
            // laser pass that ANDs them and bins); ...
            if (a.vis2 && !a.mode_intensity) {
                ForwardArgs p1 = a;
                p1.rows = nullptr;
                if (forward_grid_launch<FEAT, 1>(p1, 0, stream) && forward_grid_launch<FEAT, 2>(a, rows_in_lds, stream)) return;
            }
            // ... or, for meshes the grid cannot hold, two shadow legs per sample through the BVH
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_forward_nc<FEAT, 4>), dim3(a.src.L), dim3(256), lds, stream, a, rows_in_lds);
        }
        return;
    }
    if (forward_grid_launch<FEAT>(a, rows_in_lds, stream)) return;
    if (forward_tiled_launch<FEAT>(a, rows_in_lds, stream)) return;
    // chunk = rays traced together per (source, face): 4 when spt <= 4, else 8
    if (a.sp.spt <= 4)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_forward<FEAT, 4>), dim3(a.src.L), dim3(256), lds, stream, a, rows_in_lds);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_forward<FEAT, 8>), dim3(a.src.L), dim3(256), lds, stream, a, rows_in_lds);
}

template <int FEAT, int MODE>
void gradient_launch2(const GradientArgs& a, int grid, size_t lds, hipStream_t stream) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gradient<FEAT, MODE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gradient<FEAT, MODE>), dim3(grid), dim3(NLOS_GRAD_NT), lds, stream, a);
}

template <int FEAT>
void gradient_launch(const GradientArgs& a, int grid, size_t lds, hipStream_t stream) {
    if (a.src.sensor) {
        if constexpr ((FEAT & FEAT_GGX) == 0) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gradient<FEAT, 0, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gradient<FEAT, 0, true>), dim3(grid), dim3(NLOS_GRAD_NT), lds, stream, a);
        }
        return;
    }
    switch (a.mode) {
        case 0: gradient_launch2<FEAT, 0>(a, grid, lds, stream); break;
        case 1: gradient_launch2<FEAT, 1>(a, grid, lds, stream); break;
        case 2: gradient_launch2<FEAT, 2>(a, grid, lds, stream); break;
        case 4:
            if constexpr ((FEAT & (FEAT_GGX | FEAT_ALB)) == 0) gradient_launch2<FEAT, 4>(a, grid, lds, stream);
            break;
        default: gradient_launch2<FEAT, 3>(a, grid, lds, stream); break;
    }
}

int feat_of(const SceneView& sc, const SampleParams& sp) {
    return (sc.vertex_normal ? FEAT_VN : 0) | (sc.albedo ? FEAT_ALB : 0) | (sp.use_ggx ? FEAT_GGX : 0);
}

}  // namespace

void launch_forward(const ForwardArgs& a, hipStream_t stream) {
    if (a.src.L <= 0) return;
    const size_t row_bytes = (size_t)a.sp.nbins * sizeof(double);
    const int rows_in_lds = (!a.mode_intensity && row_bytes <= 60 * 1024) ? 1 : 0;
    const size_t lds = 8 + (rows_in_lds ? row_bytes : 0);
    if (!rows_in_lds && !a.mode_intensity) launch_zero_f64(a.rows, (size_t)a.src.L * a.sp.nbins, stream);
    switch (feat_of(a.sc, a.sp)) {
        case 0: forward_launch<0>(a, rows_in_lds, lds, stream); break;
        case 1: forward_launch<1>(a, rows_in_lds, lds, stream); break;
        case 2: forward_launch<2>(a, rows_in_lds, lds, stream); break;
        case 3: forward_launch<3>(a, rows_in_lds, lds, stream); break;
        case 4: forward_launch<4>(a, rows_in_lds, lds, stream); break;
        case 5: forward_launch<5>(a, rows_in_lds, lds, stream); break;
        case 6: forward_launch<6>(a, rows_in_lds, lds, stream); break;
        default: forward_launch<7>(a, rows_in_lds, lds, stream); break;
    }
}

void launch_smooth(const SmoothArgs& a, hipStream_t stream) {
    if (a.L <= 0) return;
    hipLaunchKernelGGL(k_smooth, dim3(a.L), dim3(256), 0, stream, a);
}

void launch_residual(const ResidualArgs& a, hipStream_t stream) {
    const size_t n = (size_t)a.L * a.T;
    int grid = (int)((n + 255) / 256);
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_residual, dim3(grid), dim3(256), 0, stream, a);
    if (a.w_width > 0 && a.L > 0)
        hipLaunchKernelGGL(k_boxfilter, dim3(a.L), dim3(256), 2 * (size_t)a.T * sizeof(double), stream, a.diff, a.T,
                           a.w_width);
}

void launch_gradient(const GradientArgs& a_in, hipStream_t stream) {
    if (a_in.src.L <= 0) return;
    GradientArgs a = a_in;
    const size_t nblk = ((size_t)a.sc.F + 63) / 64;
    size_t lds = 8 + ((size_t)a.sp.nbins + 3 * (size_t)a.K + 2) * sizeof(double) + nblk * 8 + ((nblk + 2) & ~(size_t)1) * 4;
    // compacted list of the faces with accepted samples (u16): skipped for meshes it cannot index / hold
    a.compact = (a.sc.F <= 65535 && lds + 2 * (size_t)a.sc.F + 16 <= 64 * 1024) ? 1 : 0;
    if (a.compact) lds += (2 * (size_t)a.sc.F + 15) & ~(size_t)15;
    // per-workgroup 3V-double accumulator while it fits beside the rest (one workgroup per CU at worst)
    const size_t acc = 3 * (size_t)a.sc.V * sizeof(double);
    a.lds_grad = ((a.mode == 0 || a.mode == 4) && a_in.lds_grad && lds + acc <= 150 * 1024) ? 1 : 0;
    if (!a.lds_grad && a.mode == 0 && !a.src.sensor && a_in.lds_grad) {
        // large meshes: face-major variant (per-face sums in LDS across sources, one scatter per face)
        bool done = false;
        switch (feat_of(a.sc, a.sp)) {
            case 0: done = gradient_fm_launch<0>(a, stream); break;
            case 1: done = gradient_fm_launch<1>(a, stream); break;
            case 2: done = gradient_fm_launch<2>(a, stream); break;
            case 3: done = gradient_fm_launch<3>(a, stream); break;
            case 4: done = gradient_fm_launch<4>(a, stream); break;
            case 5: done = gradient_fm_launch<5>(a, stream); break;
            case 6: done = gradient_fm_launch<6>(a, stream); break;
            default: done = gradient_fm_launch<7>(a, stream); break;
        }
        if (done) return;
    }
    if (a.lds_grad) lds += acc;
    // persistent workgroups: as many as can be co-resident (512 threads each, <= 128 VGPRs -> 4 per CU)
    int per_cu = (int)(160 * 1024 / (lds + 64));
    if (per_cu > 4) per_cu = 4;
    if (per_cu < 1) per_cu = 1;
    int grid = 256 * per_cu;
    if (grid > a.src.L) grid = a.src.L;
    switch (feat_of(a.sc, a.sp)) {
        case 0: gradient_launch<0>(a, grid, lds, stream); break;
        case 1: gradient_launch<1>(a, grid, lds, stream); break;
        case 2: gradient_launch<2>(a, grid, lds, stream); break;
        case 3: gradient_launch<3>(a, grid, lds, stream); break;
        case 4: gradient_launch<4>(a, grid, lds, stream); break;
        case 5: gradient_launch<5>(a, grid, lds, stream); break;
        case 6: gradient_launch<6>(a, grid, lds, stream); break;
        default: gradient_launch<7>(a, grid, lds, stream); break;
    }
}

void launch_intersect(const IntersectArgs& a, hipStream_t stream) {
    if (a.n <= 0) return;
    hipLaunchKernelGGL(k_intersect, dim3((a.n + 255) / 256), dim3(256), 0, stream, a);
}

void launch_zero_f64(double* p, size_t n, hipStream_t stream) {
    if (n == 0) return;
    size_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_zero_f64, dim3((unsigned)g), dim3(256), 0, stream, p, n);
}

void launch_bary_to_world(const float* V, const int32_t* F, const float* bary, int n, float* out,
                          hipStream_t stream) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_bary_to_world, dim3((n + 255) / 256), dim3(256), 0, stream, V, F, bary, n, out);
}

}  // namespace nlos
