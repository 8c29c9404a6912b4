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
template <int FEAT>
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
        for (int s = 0; s < spt; ++s) {
            Geo g;
            float t_self;
            bool ok = sample_geo<FEAT>(f, tr, o, a.sp.seed, kbase + (uint64_t)s, lb, ub, a.sc.vertex_normal,
                                       a.sc.albedo, g, t_self);
            float val = 0.0f;
            if (ok) {
                float ff = -dot(g.n, g.dir) * dot(on, g.dir) / g.h / g.h;
                if (a.sp.clamp) {
                    ff = emax0(ff);
                    ok = ff > 0.0f;      // zero contribution in both passes: never trace
                }
                val = f.area * g.alb * ff * ff;
                if (FEAT & FEAT_GGX) val = val * ggx_eval(a.sp.ggx_alpha, dot(g.n, -g.dir));
            }
            if (ok) ok = !occluded(a.sc.nodes, a.sc.n_nodes, a.sc.tris, a.sc.face_id, o, g.dir, t_self, j, f.fid);
            if (ok) {
                word |= 1u << (s & 31);
                if (a.mode_intensity) {
                    inten += (double)val / (double)spt;
                } else {
                    int bin = (int)floorf((2.0f * g.h - lb) / res);
                    if (bin >= 0 && bin < nbins) {
                        double c = (double)val / (double)spt;
                        if (rows_in_lds) unsafeAtomicAdd(&s_row[bin], c);
                        else unsafeAtomicAdd(&grow[bin], c);
                    }
                }
            }
            if ((s & 31) == 31 || s == spt - 1) {
                if (visp) visp[(size_t)(s >> 5) * F] = word;
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

// --------------------------------------------------------------------- smooth
__global__ __launch_bounds__(256) void k_smooth(SmoothArgs a) {
    const int l = blockIdx.x;
    const int R = a.refine, K = a.K, half = 2 * a.refine * a.sigma_bin;
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

template <int FEAT>
__global__ __launch_bounds__(512) void k_gradient(GradientArgs a) {
    extern __shared__ double s_mem[];       // [ticket (8 B)][diff row T][grad 3V]
    int* s_next = reinterpret_cast<int*>(s_mem);
    const int T = a.sp.nbins;
    double* s_diff = s_mem + 1;             // [T]
    double* s_grad = s_mem + 1 + T;         // [3V] when lds_grad
    const int F = a.sc.F, V = a.sc.V;
    const int spt = a.sp.spt;
    const int lane = threadIdx.x & 63;
    const int nblocks = (F + 63) >> 6;
    const int Ltot = a.src.total_sources > 0 ? a.src.total_sources : a.src.L;
    const double lbd = (double)a.sp.lb, resd = (double)a.sp.res, inv_res = 1.0 / resd;

    if (a.mode == 0 && a.lds_grad)
        for (int i = threadIdx.x; i < 3 * V; i += blockDim.x) s_grad[i] = 0.0;
    double scalar_acc = 0.0;

    for (int l = blockIdx.x; l < a.src.L; l += gridDim.x) {
        __syncthreads();                    // previous source done with s_diff
        for (int i = threadIdx.x; i < T; i += blockDim.x) s_diff[i] = a.diff[(size_t)l * T + i];
        if (threadIdx.x == 0) *s_next = 0;
        __syncthreads();
        const V3 o = ld3(a.src.origin + 3 * (size_t)l);
        const V3 on = ld3(a.src.normal + 3 * (size_t)l);
        const uint64_t lg = (uint64_t)(a.src.source_offset + l);

        for (;;) {
            const int b = wave_ticket(s_next);
            if (b >= nblocks) break;
            const int j = (b << 6) + lane;
            if (j >= F) continue;
            const uint32_t* visp = a.vis + ((size_t)l * a.vis_words) * F + j;
            uint32_t any = 0;
            for (int wi = 0; wi < a.vis_words; ++wi) any |= visp[(size_t)wi * F];
            if (!any) continue;
            const Face f = load_face(a.sc.facerec, j);
            if (a.mode == 3 && f.i0 != a.vertex_num && f.i1 != a.vertex_num && f.i2 != a.vertex_num) continue;
            const Tri tr = load_tri(a.sc.tris, j);
            const uint64_t kbase = (lg * (uint64_t)F + (uint64_t)f.fid) * (uint64_t)spt;
            const V3 e0 = f.p2 - f.p1, e1 = f.p0 - f.p2, e2 = f.p1 - f.p0;
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
                    Geo g;
                    float t_self;
                    if (!sample_geo<FEAT>(f, tr, o, a.sp.seed, kbase + (uint64_t)s, a.sp.lb, a.sp.ub,
                                          a.sc.vertex_normal, a.sc.albedo, g, t_self))
                        continue;   // cannot happen: pass 1 accepted this sample with the same arithmetic
                    const double twoh = (double)(2.0f * g.h);
                    if (a.mode == 1 || a.mode == 2) {
                        // rows A / GGX alpha: scalar gradients
                        float c2 = dot(on, g.dir);
                        float c3 = dot(g.n, -g.dir);
                        if (c2 < 0) c2 = 0;
                        if (c3 < 0) c3 = 0;
                        float ff = c2 * c3 / g.h / g.h;
                        double g0;
                        if (a.mode == 2) g0 = (double)(g.alb * ff * ff * ggx_eval_adiff(a.sp.ggx_alpha, dot(g.n, -g.dir)));
                        else g0 = (double)(ff * ff);
                        double s0 = 0.0;
                        for (int i = 0; i < a.K; ++i) {
                            int bin = tap_bin(twoh, a.tap_delta[i], lbd, resd, inv_res);
                            if (bin >= 0 && bin < T) s0 += a.tap_w[i] * (-2) * s_diff[bin];
                        }
                        sacc += (double)f.area * g0 * s0 / (double)spt;
                        continue;
                    }
                    GVec gv;
                    grad_vectors<FEAT>(f, g, on, a.normal_term, a.v1_style, a.sp.ggx_alpha, gv);
                    const V3 ce0 = cross(gv.t2, e0), ce1 = cross(gv.t2, e1), ce2 = cross(gv.t2, e2);
                    if (a.mode == 3) {
                        // single-vertex per-bin gradient: output indexed by the tap's bin
                        V3 ce; float bw;
                        if (a.vertex_num == f.i0) { ce = ce0; bw = g.u; }
                        else if (a.vertex_num == f.i1) { ce = ce1; bw = g.v; }
                        else { ce = ce2; bw = g.w; }
                        for (int i = 0; i < a.K; ++i) {
                            int bin = tap_bin(twoh, a.tap_delta[i], lbd, resd, inv_res);
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
                    // mode 0: the K-tap loop factors into two scalar sums per sample:
                    //   sum_i (t1*b + t2 x e) w_i d_i  +  b * I * dir * sum_i g_i w_i d_i
                    double s0 = 0.0, s1 = 0.0;
                    for (int i = 0; i < a.K; ++i) {
                        int bin = tap_bin(twoh, a.tap_delta[i], lbd, resd, inv_res);
                        if (bin >= 0 && bin < T) {
                            float dd = (float)((-2) * s_diff[bin]);
                            double wd = (double)((float)a.tap_w[i] * dd);
                            s0 += wd;
                            s1 += a.tap_g[i] * wd;
                        }
                    }
                    const V3 di = g.dir * gv.inten_f;
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
            if (a.mode == 0) {
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
            } else if (a.mode == 1 || a.mode == 2) {
                scalar_acc += sacc;
            }
        }
    }
    __syncthreads();
    if (a.mode == 0 && a.lds_grad) {
        const double invL = 1.0 / (double)Ltot;
        for (int i = threadIdx.x; i < 3 * V; i += blockDim.x) {
            double v = s_grad[i];
            if (v != 0.0) unsafeAtomicAdd(&a.out[i], v * invL);
        }
    }
    if (a.mode == 1 || a.mode == 2) {
        for (int off = 32; off > 0; off >>= 1) scalar_acc += __shfl_down(scalar_acc, off);
        if (lane == 0 && scalar_acc != 0.0) unsafeAtomicAdd(&a.out[0], scalar_acc / (double)Ltot);
    }
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

template <int FEAT>
void forward_launch(const ForwardArgs& a, int rows_in_lds, size_t lds, hipStream_t stream) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_forward<FEAT>), dim3(a.src.L), dim3(256), lds, stream, a, rows_in_lds);
}

template <int FEAT>
void gradient_launch(const GradientArgs& a, int grid, size_t lds, hipStream_t stream) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gradient<FEAT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)lds);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gradient<FEAT>), dim3(grid), dim3(512), lds, stream, a);
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

void launch_gradient(const GradientArgs& a, hipStream_t stream) {
    if (a.src.L <= 0) return;
    size_t lds = 8 + (size_t)a.sp.nbins * sizeof(double);
    if (a.mode == 0 && a.lds_grad) lds += 3 * (size_t)a.sc.V * sizeof(double);
    // persistent workgroups: as many as can be co-resident (512 threads each)
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
