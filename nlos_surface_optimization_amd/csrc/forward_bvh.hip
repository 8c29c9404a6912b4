// forward_bvh.hip -- pass 1 through the stackless BVH (gfx950).
//
//   k_forward     <- streamedRayTraceTriangle / streamedRayTraceIntensity
//                    (smoothed_transient/transient_and_gradient.cpp:122-237, :22-119)
//                    + thread reduction of render_smoothed_transients (:320-341)
//   k_forward_nc  <- row N (non-confocal pairs), two shadow legs per sample
// One workgroup per source, histogram row in LDS, one lane per face (Morton order) looping over the
// face's strata; the CH rays of a (source, face) chunk traverse the BVH together as a packet.  This is
// the back-end for tiny meshes (F < 64) and `force_bvh`; the grid kernels (forward_grid.hip) are the
// fast path and cover everything else.
#include "render_common.h"

namespace nlos {
namespace {

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
    const uint64_t lg = (uint64_t)(a.src.source_offset + (long long)l * a.src.source_stride);
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
                    ok = sample_geo<FEAT>(f, tr, o, a.sp.seed, kbase + (uint64_t)s, a.sp.sampled_point ? -3.0e38f : lb,
                                          a.sp.sampled_point ? 3.0e38f : ub, a.sc.vertex_normal, a.sc.albedo, g, t_self);
                if (ok && a.sp.sampled_point) {
                    // v1 non-streamed body (stratified_transient_raytracer/stratifiedTransientRenderer.cpp:91-124): the SAMPLED
                    // point's barycentrics and distance instead of the reported hit's (same ray, same visibility query)
                    float S, T;
                    sample_st(a.sp.seed, kbase + (uint64_t)s, S, T);
                    const float sq = sqrtf(T);
                    g.u = 1 - sq; g.v = (1 - S) * sq; g.w = S * sq;
                    const V3 p = bary(g.u, f.p0, g.v, f.p1, g.w, f.p2);
                    const V3 dq = p - o;
                    g.h = sqrtf(dot(dq, dq));
                    ok = (g.h <= ub / 2.0f) && (g.h >= lb / 2.0f);
                    if (FEAT & FEAT_VN)
                        g.n = bary(g.u, ld3(a.sc.vertex_normal + 3 * (size_t)f.i0), g.v, ld3(a.sc.vertex_normal + 3 * (size_t)f.i1), g.w,
                                   ld3(a.sc.vertex_normal + 3 * (size_t)f.i2));
                    if (FEAT & FEAT_ALB) g.alb = g.u * a.sc.albedo[f.i0] + g.v * a.sc.albedo[f.i1] + g.w * a.sc.albedo[f.i2];
                }
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
                        if (rows_in_lds) lds_add_f64(&s_row[bin[c]], cc);
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
    const uint64_t lg = (uint64_t)(a.src.source_offset + (long long)l * a.src.source_stride);
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
                    if (FEAT & FEAT_GGX) vv = vv * ggx_pair<false>(a.sp.ggx_alpha, g.n, -g.dirA, -g.dirB).brdf;
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
                    if (rows_in_lds) lds_add_f64(&s_row[bin[c]], cc);
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

template <int FEAT>
void bvh_launch(const ForwardArgs& a, int rows_in_lds, size_t lds, hipStream_t stream) {
    if (a.src.sensor) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_forward_nc<FEAT, 4>), dim3(a.src.L), dim3(256), lds, stream, a, rows_in_lds);
        return;
    }
    // chunk = rays traced together per (source, face): 4 when spt <= 4, else 8
    if (a.sp.spt <= 4)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_forward<FEAT, 4>), dim3(a.src.L), dim3(256), lds, stream, a, rows_in_lds);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_forward<FEAT, 8>), dim3(a.src.L), dim3(256), lds, stream, a, rows_in_lds);
}

}  // namespace

void launch_forward_bvh(const ForwardArgs& a, int rows_in_lds, hipStream_t stream) {
    const size_t lds = 8 + (rows_in_lds ? (size_t)a.sp.nbins * sizeof(double) : 0);
    switch (feat_of(a.sc, a.sp)) {
        case 0: bvh_launch<0>(a, rows_in_lds, lds, stream); break;
        case 1: bvh_launch<1>(a, rows_in_lds, lds, stream); break;
        case 2: bvh_launch<2>(a, rows_in_lds, lds, stream); break;
        case 3: bvh_launch<3>(a, rows_in_lds, lds, stream); break;
        case 4: bvh_launch<4>(a, rows_in_lds, lds, stream); break;
        case 5: bvh_launch<5>(a, rows_in_lds, lds, stream); break;
        case 6: bvh_launch<6>(a, rows_in_lds, lds, stream); break;
        default: bvh_launch<7>(a, rows_in_lds, lds, stream); break;
    }
}

}  // namespace nlos
