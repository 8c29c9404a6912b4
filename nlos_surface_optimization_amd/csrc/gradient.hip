// gradient.hip -- pass 2: analytic per-vertex / scalar gradients from the visibility cache (gfx950).
//
//   k_gradient<FEAT, MODE, NC>  <- streamedRayTraceTriangleGradient / ...GradientAlbedo / ...GradientAlpha /
//                                  ...VertexGradient (smoothed_transient/transient_and_gradient.cpp:843-1007,
//                                  :571-695, ggx/transient_and_gradient.cpp:385-512, :697-840) + reduction
//                                  (:561-565); jitter taps (jitter/transient_and_gradient.cpp:944-969)
//   k_gradient_fm<FEAT>         the vertex gradient for meshes whose 3V-double accumulator does not fit LDS
// No ray is traced here: pass 1 left one bit per accepted sample.
#include "render_common.h"

#include <cstdlib>
#include <cstring>

namespace nlos {
namespace {

// accepted-sample bits (spt <= 32) of live-list entry li out of the source's item masks: ray r = li * spt + s is bit r & 63
// of item r >> 6 (items[0] = header: number of live faces in its low 16 bits -- forward_grid.hip --, the masks follow)
__device__ __forceinline__ uint32_t item_bits(const unsigned long long* __restrict__ items, int li, int spt) {
    const uint32_t r0 = (uint32_t)li * (uint32_t)spt;
    const uint32_t w = r0 >> 6, sh = r0 & 63u;
    unsigned long long m = items[1 + w] >> sh;
    if (sh + (uint32_t)spt > 64u) m |= items[2 + w] << (64u - sh);
    return (uint32_t)(m & ((1ull << spt) - 1ull));
}

// item masks -> per-face words [L, 1, F] (spt <= 32), for the consumers that index the cache by face: the face-major
// gradient kernel and the diagnostic read-back.  One workgroup per source.
__global__ __launch_bounds__(256) void k_items_to_words(const unsigned long long* __restrict__ items, int items_stride,
                                                        const uint16_t* __restrict__ live, int F, int spt,
                                                        uint32_t* __restrict__ words) {
    const int l = blockIdx.x;
    uint32_t* w = words + (size_t)l * F;
    for (int j = threadIdx.x; j < F; j += blockDim.x) w[j] = 0u;
    __syncthreads();
    const unsigned long long* it = items + (size_t)l * (size_t)items_stride;
    const int n_live = (int)(it[0] & 0xffffull);
    for (int li = threadIdx.x; li < n_live; li += blockDim.x) w[live[(size_t)l * F + li]] = item_bits(it, li, spt);
}

// Per accepted sample the nine sums of a face grow by (t1 b_q + t2 x e_q) s0 + di b_q s1, q = 0, 1, 2.  The edges e_q belong
// to the FACE, so sum_s (t2_s x e_q) s0_s = (sum_s t2_s s0_s) x e_q: the sample adds b_q P to the nine sums, P = t1 s0 + di s1,
// and t2 s0 to three more (all in fp64), and the three cross products are taken once per face -- 30 instead of 36
// half-rate operations per sample, and none of the three fp32 cross products, nine edge subtractions and eighteen A1 / A2
// operations the per-sample form needed (round 5: pass 2 0.383 -> see HISTORY.md R5).
struct FaceSums {
    double p[9];     // sum b_q P[c]
    double x[3];     // sum t2[c] s0
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int q = 0; q < 9; ++q) p[q] = 0.0;
        x[0] = x[1] = x[2] = 0.0;
    }
    __device__ __forceinline__ void add(V3 t1, V3 t2, V3 di, float b0, float b1, float b2, double s0, double s1) {
        const double px = fma((double)t1.x, s0, (double)di.x * s1), py = fma((double)t1.y, s0, (double)di.y * s1),
                     pz = fma((double)t1.z, s0, (double)di.z * s1);
        x[0] = fma((double)t2.x, s0, x[0]); x[1] = fma((double)t2.y, s0, x[1]); x[2] = fma((double)t2.z, s0, x[2]);
        const double bq[3] = {(double)b0, (double)b1, (double)b2};
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            p[3 * q + 0] = fma(bq[q], px, p[3 * q + 0]);
            p[3 * q + 1] = fma(bq[q], py, p[3 * q + 1]);
            p[3 * q + 2] = fma(bq[q], pz, p[3 * q + 2]);
        }
    }
    // the nine sums: p + x cross e_q, e_0 = p2 - p1, e_1 = p0 - p2, e_2 = p1 - p0 (fp32 differences, as the per-sample form took them)
    __device__ __forceinline__ void finish(const Face& f, double* out9) const {
        const V3 e[3] = {f.p2 - f.p1, f.p0 - f.p2, f.p1 - f.p0};
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const double ex = (double)e[q].x, ey = (double)e[q].y, ez = (double)e[q].z;
            out9[3 * q + 0] = p[3 * q + 0] + fma(x[1], ez, -(x[2] * ey));
            out9[3 * q + 1] = p[3 * q + 1] + fma(x[2], ex, -(x[0] * ez));
            out9[3 * q + 2] = p[3 * q + 2] + fma(x[0], ey, -(x[1] * ex));
        }
    }
};

#ifndef NLOS_GRAD_NT
#define NLOS_GRAD_NT 512
#define NLOS_GRAD_WPS 4
#endif
// NT = 1024: one workgroup per CU with the same sixteen waves, for meshes whose 3V-double accumulator leaves no
// room for two workgroups of 512 (V > ~2600)
// GEO: the samples' h and hit barycentrics come from pass 1's geometry cache (confocal vertex gradient on the item-mask
// layout) instead of being recomputed (hash, two square roots, own-face test); an instance of its own so that the
// recomputing path keeps its registers
template <int FEAT, int MODE, bool NC = false, int NT = NLOS_GRAD_NT, bool GEO = false>
__global__ __launch_bounds__(NT, NLOS_GRAD_WPS) void k_gradient(GradientArgs a) {
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
    // per-bin tap weights (TapTables::wt; vertex gradient with Gaussian taps) when the launcher made room for them
    const int n_wt = (MODE == 0 && a.tap_wt && a.wt_in_lds) ? 2 * (a.refine + 1) * a.tap_nb : 0;
    double* s_wt = s_p1 + K + 1;            // [n_wt]
    double* s_grad = s_wt + n_wt;           // [3V] when lds_grad
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
        // scalar gradients (modes 1, 2) group their taps over the double weights
        for (int i = threadIdx.x; i <= K; i += blockDim.x) { s_p0[i] = (MODE == 1 || MODE == 2) ? a.tap_pw[i] : a.tap_p0[i]; s_p1[i] = a.tap_p1[i]; }
    }
    for (int i = threadIdx.x; i < n_wt; i += blockDim.x) s_wt[i] = a.tap_wt[i];
    if ((MODE == 0 || MODE == 4) && a.lds_grad)
        for (int i = threadIdx.x; i < 3 * V; i += blockDim.x) s_grad[i] = 0.0;
    double scalar_acc = 0.0;
    TapTables tt;
    tt.delta = s_delta; tt.p0 = s_p0; tt.p1 = s_p1; tt.K = K; tt.two_rs = a.two_rs; tt.r_over_res = a.r_over_res; tt.refine = a.refine;
    tt.wt = n_wt ? s_wt : nullptr; tt.nb = a.tap_nb;

    for (int l = blockIdx.x; l < a.src.L; l += gridDim.x) {
        __syncthreads();                    // previous source done with s_diff
        // vertex-gradient modes read the row only as (float)(-2 d): rounded once here instead of once per tap group
        for (int i = threadIdx.x; i < T; i += blockDim.x) {
            double d;
            if (a.inline_residual) {
                // row D here instead of in a launch of its own: difference = (data - transient) [-> 2 d^3] * weight
                d = a.res_data[(size_t)l * T + i] - a.res_transient[(size_t)l * T + i];
                if (a.res_loss_test == 1) d = 2 * d * d * d;
                if (a.res_weight) d = d * a.res_weight[(size_t)l * T + i];
            } else {
                d = a.diff[(size_t)l * T + i];
            }
            s_diff[i] = (MODE == 0 || MODE == 4) ? (double)(float)((-2) * d) : (-2) * d;      // scalar modes: -2 d in double
        }
        if (threadIdx.x == 0) *s_next = 0;
        // Item-mask layout of the cache (nlos_kernels.h, ForwardArgs::vis_items): the entries to look at are those of the
        // source's bucketed live list (half of the faces), their bits come out of one or two 64-bit item masks
        const unsigned long long* it_l = a.vis_items ? a.vis_items + (size_t)l * (size_t)a.items_stride : nullptr;
        const uint16_t* live_l = a.vis_items ? a.live + (size_t)l * F : nullptr;
        // pass 1's geometry cache (vertex-gradient modes of confocal renders): h and the hit's barycentrics per ray of the live list
        typedef float f4_t __attribute__((ext_vector_type(4)));
        const f4_t* geo_l = GEO ? reinterpret_cast<const f4_t*>(a.geo) + (size_t)l * (size_t)a.geo_stride : nullptr;
        typedef float f2_t __attribute__((ext_vector_type(2)));
        const f2_t* geo_w = GEO ? reinterpret_cast<const f2_t*>(a.geo + 4 * (size_t)a.geo_sources * (size_t)a.geo_stride) + (size_t)l * (size_t)a.geo_stride : nullptr;
        const int n_src = it_l ? (int)(it_l[0] & 0xffffull) : F;
        // Per-face words: faces with at least one accepted sample, compacted in order (pass 1 left the words).  Item masks: no
        // compaction (round 5) -- a lane takes a live-list ENTRY as it stands and an entry without accepted samples idles its
        // lane for one block: with pass 2's sample loop at half of its round-4 length the three passes and barriers of the
        // compaction cost more than the 11 % of idle lanes they saved (0.3634 -> 0.3485 ms, profiles/r05_ab_grad_sources.log).
        // (pairs keep the compaction: their samples are regenerated -- two legs, two triangle tests -- and an idle lane costs
        // that much more: 0.555 vs 0.588 ms without)
        const bool skip_compact = it_l != nullptr && !NC;
        for (int b = wave; !skip_compact && b < nblocks; b += nwaves) {
            const int j = (b << 6) + lane;
            uint32_t any = 0;
            if (j < n_src) {
                if (it_l) {
                    any = item_bits(it_l, j, spt);
                } else {
                    const uint32_t* visp = a.vis + ((size_t)l * a.vis_words) * F + j;
                    for (int wi = 0; wi < a.vis_words; ++wi) any |= visp[(size_t)wi * F];
                }
            }
            const unsigned long long m = __ballot(any != 0u);
            if (lane == 0) s_mask[b] = m;
        }
        if (!skip_compact) __syncthreads();
        if (!skip_compact && threadIdx.x < 64) {
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
        if (!skip_compact) __syncthreads();
        for (int b = wave; !skip_compact && b < nblocks; b += nwaves) {
            const unsigned long long m = s_mask[b];
            if (a.compact && ((m >> lane) & 1ull))
                s_live[s_base[b] + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)((b << 6) + lane);
        }
        __syncthreads();
        const int n_live = (a.compact && !skip_compact) ? (int)s_base[nblocks] : n_src;
        const int live_blocks = (n_live + 63) >> 6;
        // measurement l: source l, pair l, or -- pass 2 of the L x S product -- the pair (laser l / S, sensor l % S)
        const size_t la = (NC && a.src.n_sensors > 0) ? (size_t)(l / a.src.n_sensors) : (size_t)l;
        const size_t lb_ = (NC && a.src.n_sensors > 0) ? (size_t)(l % a.src.n_sensors) : (size_t)l;
        const V3 o = ld3(a.src.origin + 3 * la);
        const V3 on = ld3(a.src.normal + 3 * la);
        const V3 ob = NC ? ld3(a.src.sensor + 3 * lb_) : o;
        const V3 onb = NC ? ld3(a.src.sensor_normal + 3 * lb_) : on;
        const uint64_t lg = (uint64_t)(a.src.source_offset + (long long)l * a.src.source_stride);
        // regenerated samples (no geometry cache): the draw keyed per source like pass 1's, and the lean forms of sqrt / reciprocal
        // where this source's frame and the window guarantee their range (render_common.h: sample_geo_keyed) -- the same bits
        const uint64_t zbase = sample_zbase(a.sp.seed, lg * (uint64_t)F * (uint64_t)spt);
        const bool lean_src = !GEO && a.lean_params && source_frame(a.sc.nodes, o).ok && (!NC || source_frame(a.sc.nodes, ob).ok);

#ifndef NLOS_GRAD_LOOKAHEAD
#define NLOS_GRAD_LOOKAHEAD 1
#endif
        // Block lookahead (round 6; the geometry-cache instance of Lambertian renders): ticket -> live-list entry -> item masks ->
        // face record -> first geometry record are dependent round trips in front of a block's arithmetic, and four waves per
        // SIMD do not hide them.  The ticket, the entry's face slot and its accepted-sample bits of the NEXT block are requested
        // behind this block's face-record loads (three registers across the block): metric pass 2 0.353 -> 0.348 ms, 32 x 32
        // sources 0.106 -> 0.103 ms.  Measured and NOT enabled elsewhere (profiles/r06_ab_grad_lookahead.log): the regenerating
        // instance loses (mannequin 0.638 -> 0.661 ms: its registers are worth more), the GGX one is even (0.674 -> 0.678);
        // requesting the next block's first geometry record as well (six more registers, 7 spilled) loses everywhere (0.358).
        const bool ahead = NLOS_GRAD_LOOKAHEAD && skip_compact && GEO && !(FEAT & FEAT_GGX);
        int b_nx = 0, j_nx = 0;
        uint32_t ib_nx = 0u;
        auto fetch_next = [&]() {
            b_nx = wave_ticket(s_next);
            const int li1 = (b_nx << 6) + lane;
            if (b_nx < live_blocks && li1 < n_live) { j_nx = (int)live_l[li1]; ib_nx = item_bits(it_l, li1, spt); }
        };
        if (ahead) fetch_next();
        for (;;) {
            const int b = ahead ? b_nx : wave_ticket(s_next);
            if (b >= live_blocks) break;
            const int li = (b << 6) + lane;
            const bool has = li < n_live;
            const int e = (a.compact && !skip_compact) ? (has ? (int)s_live[li] : 0) : li;   // a face slot, or (item masks) an entry of the live list
            const int j = ahead ? j_nx : ((it_l && has) ? (int)live_l[e] : e);
            const uint32_t ibits = ahead ? ib_nx : ((it_l && has) ? item_bits(it_l, e, spt) : 0u);
            Face f;
            Tri tr;
            if (has) load_face_tri<FEAT | FEAT_VN>(a.sc, j, f, tr);    // vertices, ids, and the per-face constants the scene build evaluated
            if (ahead) fetch_next();                                   // (behind this block's loads: wave-uniform ticket, every lane takes part)
            if (!has) continue;
            const uint32_t* visp = it_l ? nullptr : a.vis + ((size_t)l * a.vis_words) * F + j;
            const int n_words = it_l ? 1 : a.vis_words;
            if (MODE == 3 && f.i0 != a.vertex_num && f.i1 != a.vertex_num && f.i2 != a.vertex_num) continue;
            const uint64_t kbase = (lg * (uint64_t)F + (uint64_t)f.fid) * (uint64_t)spt;
            FaceSums fs;
            fs.clear();
            double sacc = 0.0;

            for (int wi = 0; wi < n_words; ++wi) {
                uint32_t word = it_l ? ibits : visp[(size_t)wi * F];
                // GEO: the record of the NEXT accepted sample is requested one sample ahead (a dependent load per
                // sample in front of its arithmetic left the kernel waiting: 0.558 -> 0.503 ms only, profiles/r04_ab_geo_cache.log)
                f4_t nrec = {0.0f, 0.0f, 0.0f, 0.0f};
                f2_t nw = {0.0f, 0.0f};
                if (GEO && word) {
                    const size_t at = (size_t)(__ffs(word) - 1) * (size_t)F + (size_t)e;      // [stratum][live-list entry]
                    nrec = __builtin_nontemporal_load(geo_l + at);
                    nw = __builtin_nontemporal_load(geo_w + at);
                }
                while (word) {
                    const int bit = __ffs(word) - 1;
                    word &= word - 1;
                    const int s = (wi << 5) + bit;
                    const V3 cdir = mk(nrec.x, nrec.y, nrec.z);
                    const float cv = nrec.w, cw = nw.x, ch = nw.y;
                    if (GEO && word) {
                        const size_t at = (size_t)(__ffs(word) - 1) * (size_t)F + (size_t)e;      // [stratum][live-list entry]
                        nrec = __builtin_nontemporal_load(geo_l + at);
                        nw = __builtin_nontemporal_load(geo_w + at);
                    }
                    if (NC) {
                        // row N: two legs, d(d1 + d2)/dp = dirA + dirB; P1 carries the confocal factor 2
                        GeoNC gc;
                        float tA, tB;
                        if (!sample_geo_nc_rt<FEAT>(f, tr, o, ob, a.sp.seed, kbase + (uint64_t)s, lean_src, a.sp.lb, a.sp.ub,
                                                    a.sc.vertex_normal, a.sc.albedo, gc, tA, tB))
                            continue;
                        GVec gv;
                        grad_vectors_nc<FEAT>(f, gc, on, onb, a.normal_term, a.sp.ggx_alpha, gv);
                        double s0, s1;
                        grouped_taps(tt, s_diff, T, (double)(gc.d1 + gc.d2), lbd, resd, inv_res, s0, s1);
                        const V3 di = ((gc.dirA + gc.dirB) * 0.5f) * gv.inten_f;
                        fs.add(gv.t1, gv.t2, di, gc.u, gc.v, gc.w, s0, s1);
                        continue;
                    }
                    Geo g;
                    float t_self;
                    if (GEO) {
                        cached_geo<FEAT>(f, cdir, cv, cw, ch, a.sc.vertex_normal, a.sc.albedo, g);
                    }
#ifdef NLOS_DIAG_GRAD_FREE_VWH      // diagnostic builds only (results are wrong): what a 12-byte (v, w, h) record from pass 1 could
                                    // save the regenerating pass 2 at most -- the draw and the sampled direction stay, the own-face
                                    // test, the hit point and its square root are free (tools: --diagnostic-no-gate)
                    else {
                        float S_, T_;
                        sample_st_c(zbase, (uint32_t)f.fid * (uint32_t)spt + (uint32_t)s, S_, T_);
                        const float sq_ = sqrt_cr0(T_);
                        g.u = 1 - sq_; g.v = (1 - S_) * sq_; g.w = S_ * sq_;
                        const V3 p_ = bary(g.u, f.p0, g.v, f.p1, g.w, f.p2);
                        const V3 d_ = p_ - o;
                        const float dd_ = dot(d_, d_);
                        const float rs_ = rcp_cr(sqrt_cr(dd_));
                        g.dir = d_ * rs_;
                        g.h = dd_ * rs_;
                        g.n = f.fn;
                        if (FEAT & FEAT_VN)
                            g.n = bary(g.u, ld3(a.sc.vertex_normal + 3 * (size_t)f.i0), g.v, ld3(a.sc.vertex_normal + 3 * (size_t)f.i1), g.w,
                                       ld3(a.sc.vertex_normal + 3 * (size_t)f.i2));
                        g.alb = 1.0f;
                        if (FEAT & FEAT_ALB) g.alb = g.u * a.sc.albedo[f.i0] + g.v * a.sc.albedo[f.i1] + g.w * a.sc.albedo[f.i2];
                        t_self = 0.0f;
                    }
#else
                    else if (!sample_geo_keyed<FEAT>(f, tr, o, zbase, (uint32_t)f.fid * (uint32_t)spt + (uint32_t)s, lean_src, a.sp.lb, a.sp.ub,
                                                     a.sc.vertex_normal, a.sc.albedo, g, t_self))
                        continue;   // cannot happen: pass 1 accepted this sample with the same arithmetic
#endif
                    const double twoh = (double)(2.0f * g.h);
                    if (MODE == 1 || MODE == 2) {
                        // rows A / GGX alpha: scalar gradients (literal tap loop, double weights)
                        float c2 = dot(on, g.dir);
                        float c3 = dot(g.n, -g.dir);
                        if (c2 < 0) c2 = 0;
                        if (c3 < 0) c3 = 0;
#ifdef NLOS_DIAG_GGX_IEEE            // diagnostic builds only: the contract's divisions (rounds 1 - 5)
                        float ff = c2 * c3 / g.h / g.h;
                        double g0;
                        if (MODE == 2) g0 = (double)(g.alb * ff * ff * ggx_eval_adiff(a.sp.ggx_alpha, dot(g.n, -g.dir)));
                        else g0 = (double)(ff * ff);
#else
                        // (non-decision arithmetic, round 6: 1-ulp reciprocal, the BRDF derivative in single precision)
                        const float ihs = rcp_fast(g.h);
                        float ff = (c2 * c3) * (ihs * ihs);
                        double g0;
                        if (MODE == 2) g0 = (double)(g.alb * ff * ff * ggx_eval_adiff_fast(a.sp.ggx_alpha, dot(g.n, -g.dir)));
                        else g0 = (double)(ff * ff);
#endif
                        // sum_i w_i (-2) d[bin_i], the taps grouped by bin over the prefix sums of w (the literal
                        // 41-tap loop with one fp64 floor per tap cost 2.5x the rest of the sample)
                        double s0, s1_unused;
                        grouped_taps(tt, s_diff, T, twoh, lbd, resd, inv_res, s0, s1_unused);
                        sacc += (double)f.area * g0 * s0 / (double)spt;
                        continue;
                    }
                    GVec gv;
                    grad_vectors<FEAT>(f, g, on, a.normal_term, a.v1_style, a.sp.ggx_alpha, gv);
                    if (MODE == 3) {
                        // single-vertex per-bin gradient: output indexed by the tap's bin
                        const V3 e0 = f.p2 - f.p1, e1 = f.p0 - f.p2, e2 = f.p1 - f.p0;
                        const V3 ce0 = cross(gv.t2, e0), ce1 = cross(gv.t2, e1), ce2 = cross(gv.t2, e2);
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
                            const float dd = (float)s_diff[b0 + i];      // (float)(-2 d), see the row load
                            s0 += (double)((float)s_delta[i] * dd);
                            s1 += (double)(((float)(s_p0[i] * m2i) / a.sp.res) * dd);
                        }
                        di = g.dir;
                    } else {
                        grouped_taps(tt, s_diff, T, twoh, lbd, resd, inv_res, s0, s1);
                        di = g.dir * gv.inten_f;
                    }
                    fs.add(gv.t1, gv.t2, di, g.u, g.v, g.w, s0, s1);
                }
            }
            if (MODE == 0 || MODE == 4) {
                double acc[9];
                fs.finish(f, acc);
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
#ifndef NLOS_FM_CHUNK
#define NLOS_FM_CHUNK 512
#define NLOS_FM_BATCH 4
#define NLOS_FM_THREADS 256
#define NLOS_FM_WPS 1
#endif
constexpr int kFmChunk = NLOS_FM_CHUNK, kFmBatch = NLOS_FM_BATCH, kFmThreads = NLOS_FM_THREADS;

// JIT: the taps of a measured jitter kernel (mode 4 of k_gradient: tap i -> bin b0 + i - offset)
template <int FEAT, bool NC = false, bool JIT = false>
__global__ __launch_bounds__(kFmThreads, NLOS_FM_WPS) void k_gradient_fm(GradientArgs a, int src_per_group) {
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
    // A chunk is kFmChunk / 8 blocks of 8 consecutive faces taken round-robin over the Morton order (block b
    // of the chunk = 8-face block b * gridDim.x + blockIdx.x): every chunk then holds a similar mix of faces
    // that see the wall and faces that do not -- consecutive chunks differ by 100x in accepted samples and the
    // heavy ones set the kernel time (512 consecutive faces: 1.45 ms at F = 20 k; 64-face blocks 1.11; 8-face
    // blocks 1.02; finer changes nothing).  Eight faces are one 32-byte piece of a visibility row.
#ifndef NLOS_FM_BLK_SHIFT
#define NLOS_FM_BLK_SHIFT 3
#endif
    constexpr int kBs = NLOS_FM_BLK_SHIFT;
    auto face_of = [&](int jl) -> int { return (((jl >> kBs) * (int)gridDim.x + (int)blockIdx.x) << kBs) + (jl & ((1 << kBs) - 1)); };
    const int l0 = blockIdx.y * src_per_group, l1 = min(l0 + src_per_group, a.src.L);
    const int spt = a.sp.spt;
    const int Ltot = a.src.total_sources > 0 ? a.src.total_sources : a.src.L;
    const double lbd = (double)a.sp.lb, resd = (double)a.sp.res, inv_res = 1.0 / resd;
    for (int i = tid; i < 9 * kFmChunk; i += kFmThreads) s_acc[i] = 0.0;
    if (JIT) {
        // jitter taps: s_delta <- (float) jitter_weight, s_p0 <- jitter_grad
        for (int i = tid; i < K; i += kFmThreads) { s_delta[i] = (double)(float)a.tap_w[i]; s_p0[i] = a.tap_g[i]; }
    } else {
        for (int i = tid; i < K; i += kFmThreads) s_delta[i] = a.tap_delta[i];
        for (int i = tid; i <= K; i += kFmThreads) { s_p0[i] = a.tap_p0[i]; s_p1[i] = a.tap_p1[i]; }
    }
    TapTables tt;
    tt.delta = s_delta; tt.p0 = s_p0; tt.p1 = s_p1; tt.K = K; tt.two_rs = a.two_rs; tt.r_over_res = a.r_over_res; tt.refine = a.refine;
    // (its LDS is full: the 880 bytes of tabulated bin weights are read through the cache)
    tt.wt = (!JIT && a.tap_wt && a.tap_nb > 0 && K > 1) ? a.tap_wt : nullptr; tt.nb = a.tap_nb;

    for (int lb0 = l0; lb0 < l1; lb0 += kFmBatch) {
        const int nb = min(kFmBatch, l1 - lb0);
        __syncthreads();                                   // previous batch done with rows / list
        for (int i = tid; i < nb * T; i += kFmThreads) s_rows[i] = (double)(float)((-2) * a.diff[(size_t)lb0 * T + i]);
        if (tid == 0) *s_cnt = 0;
        __syncthreads();
        // (source, face) items of this batch with at least one accepted sample
        for (int it = tid; it < nb * kFmChunk; it += kFmThreads) {
            const int bl = it / kFmChunk, jl = it - bl * kFmChunk;
            uint32_t any = 0;
            if (face_of(jl) < F) {
                const uint32_t* visp = a.vis + ((size_t)(lb0 + bl) * a.vis_words) * F + face_of(jl);
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
            const int l = lb0 + bl, j = face_of(jl);
            const double* s_diff = s_rows + bl * T;
            Face f;
            Tri tr;
            load_face_tri<FEAT | FEAT_VN>(a.sc, j, f, tr);    // vertices, ids, and the per-face constants the scene build evaluated
            // (pass 2 of the L x S product: measurement l is the pair (laser l / S, sensor l % S))
            const size_t la = (NC && a.src.n_sensors > 0) ? (size_t)(l / a.src.n_sensors) : (size_t)l;
            const size_t lsn = (NC && a.src.n_sensors > 0) ? (size_t)(l % a.src.n_sensors) : (size_t)l;
            const V3 o = ld3(a.src.origin + 3 * la);
            const V3 on = ld3(a.src.normal + 3 * la);
            const uint64_t lgf = (uint64_t)(a.src.source_offset + (long long)l * a.src.source_stride);
            const uint64_t kbase = (lgf * (uint64_t)F + (uint64_t)f.fid) * (uint64_t)spt;
            const uint64_t zbase = sample_zbase(a.sp.seed, lgf * (uint64_t)F * (uint64_t)spt);
            const V3 obf = NC ? ld3(a.src.sensor + 3 * lsn) : o;
            const bool lean_src = a.lean_params && source_frame(a.sc.nodes, o).ok && (!NC || source_frame(a.sc.nodes, obf).ok);      // (per lane here: items mix sources)
            const uint32_t* visp = a.vis + ((size_t)l * a.vis_words) * F + j;
            FaceSums fs;
            fs.clear();
            for (int wi = 0; wi < a.vis_words; ++wi) {
                uint32_t word = visp[(size_t)wi * F];
                while (word) {
                    const int bit = __ffs(word) - 1;
                    word &= word - 1;
                    GVec gv;
                    V3 di;
                    float bw[3];
                    double twoh;
                    int jit_b0 = 0;
                    if (NC) {
                        // row N: two legs, d(d1 + d2)/dp = dirA + dirB; P1 carries the confocal factor 2
                        GeoNC gc;
                        float tA, tB;
                        if (!sample_geo_nc_rt<FEAT>(f, tr, o, obf, a.sp.seed, kbase + (uint64_t)((wi << 5) + bit), lean_src, a.sp.lb, a.sp.ub,
                                                    a.sc.vertex_normal, a.sc.albedo, gc, tA, tB))
                            continue;
                        grad_vectors_nc<FEAT>(f, gc, on, ld3(a.src.sensor_normal + 3 * lsn), a.normal_term, a.sp.ggx_alpha, gv);
                        di = ((gc.dirA + gc.dirB) * 0.5f) * gv.inten_f;
                        bw[0] = gc.u; bw[1] = gc.v; bw[2] = gc.w;
                        twoh = (double)(gc.d1 + gc.d2);
                    } else {
                        Geo g;
                        float t_self;
                        if (!sample_geo_keyed<FEAT>(f, tr, o, zbase, (uint32_t)f.fid * (uint32_t)spt + (uint32_t)((wi << 5) + bit), lean_src,
                                                    a.sp.lb, a.sp.ub, a.sc.vertex_normal, a.sc.albedo, g, t_self))
                            continue;
                        grad_vectors<FEAT>(f, g, on, a.normal_term, a.v1_style, a.sp.ggx_alpha, gv);
                        di = JIT ? g.dir : g.dir * gv.inten_f;
                        bw[0] = g.u; bw[1] = g.v; bw[2] = g.w;
                        twoh = (double)(2.0f * g.h);
                        jit_b0 = (int)floorf((2.0f * g.h - a.sp.lb) / a.sp.res) - a.two_rs;
                    }
                    double s0, s1;
                    if (JIT) {
                        // jitter/transient_and_gradient.cpp:944-969, as in k_gradient<FEAT, 4>
                        const double m2i = (double)gv.inten_f * (-2);
                        s0 = 0.0;
                        s1 = 0.0;
                        const int i0 = max(0, -jit_b0), i1 = min(K, T - jit_b0);
                        for (int i = i0; i < i1; ++i) {
                            const float dd = (float)s_diff[jit_b0 + i];      // (float)(-2 d), see the row load
                            s0 += (double)((float)s_delta[i] * dd);
                            s1 += (double)(((float)(s_p0[i] * m2i) / a.sp.res) * dd);
                        }
                    } else {
                        grouped_taps(tt, s_diff, T, twoh, lbd, resd, inv_res, s0, s1);
                    }
                    fs.add(gv.t1, gv.t2, di, bw[0], bw[1], bw[2], s0, s1);
                }
            }
            double acc[9];
            fs.finish(f, acc);
#pragma unroll
            for (int q = 0; q < 9; ++q) unsafeAtomicAdd(&s_acc[9 * jl + q], acc[q]);
        }
    }
    __syncthreads();
    // one pass over the chunk: scale and scatter to the vertices
    for (int jl = tid; jl < kFmChunk; jl += kFmThreads) {
        if (face_of(jl) >= F) continue;
        const Face f = load_face(a.sc.facerec, face_of(jl));
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
    // only the vertex gradient (Gaussian or jitter taps) of meshes whose 3V accumulator cannot live in LDS
    if ((a.mode != 0 && a.mode != 4) || a.sp.nbins * kFmBatch > 8192) return false;
    if (a.src.sensor && ((FEAT & FEAT_GGX) || a.mode == 4)) return false;
    if (a.mode == 4 && (FEAT & (FEAT_GGX | FEAT_ALB))) return false;
    const size_t lds = ((size_t)9 * kFmChunk + (size_t)kFmBatch * a.sp.nbins + 3 * (size_t)a.K + 2) * sizeof(double) +
                       (size_t)kFmBatch * kFmChunk * 2 + 16;
    if (lds > 80 * 1024) return false;
    const int nchunks = (a.sc.F + kFmChunk - 1) / kFmChunk;
    // enough workgroups to fill the chip (four per CU: 1024 on the 256 CUs of an unpartitioned MI355X), sources in multiples of the batch
    const int want = 4 * device_cu_count();
    int groups = (want + nchunks - 1) / nchunks;
    int per = (a.src.L + groups - 1) / groups;
    per = ((per + kFmBatch - 1) / kFmBatch) * kFmBatch;
    if (per < kFmBatch) per = kFmBatch;
    groups = (a.src.L + per - 1) / per;
    if (a.src.sensor) {
        if constexpr ((FEAT & FEAT_GGX) == 0) {
            note_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gradient_fm<FEAT, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute(dynamic LDS)");
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gradient_fm<FEAT, true>), dim3(nchunks, groups), dim3(kFmThreads), lds, stream, a, per);
        }
        return true;
    }
    if (a.mode == 4) {
        if constexpr ((FEAT & (FEAT_GGX | FEAT_ALB)) == 0) {
            note_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gradient_fm<FEAT, false, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute(dynamic LDS)");
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gradient_fm<FEAT, false, true>), dim3(nchunks, groups), dim3(kFmThreads), lds, stream, a, per);
        }
        return true;
    }
    note_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gradient_fm<FEAT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds), "hipFuncSetAttribute(dynamic LDS)");
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gradient_fm<FEAT>), dim3(nchunks, groups), dim3(kFmThreads), lds, stream, a, per);
    return true;
}

template <int FEAT, int MODE>
void gradient_launch2(const GradientArgs& a, int grid, size_t lds, hipStream_t stream, bool wide = false) {
    if constexpr (MODE == 0) {
        if (a.geo && a.vis_items) {      // pass 1's geometry cache (and the item masks that index it)
            if (wide) {
                note_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gradient<FEAT, 0, false, 1024, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute(dynamic LDS)");
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gradient<FEAT, 0, false, 1024, true>), dim3(grid), dim3(1024), lds, stream, a);
                return;
            }
            note_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gradient<FEAT, 0, false, NLOS_GRAD_NT, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute(dynamic LDS)");
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gradient<FEAT, 0, false, NLOS_GRAD_NT, true>), dim3(grid), dim3(NLOS_GRAD_NT), lds, stream, a);
            return;
        }
        if (wide) {
            note_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gradient<FEAT, 0, false, 1024>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute(dynamic LDS)");
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gradient<FEAT, 0, false, 1024>), dim3(grid), dim3(1024), lds, stream, a);
            return;
        }
    }
    note_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gradient<FEAT, MODE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute(dynamic LDS)");
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gradient<FEAT, MODE>), dim3(grid), dim3(NLOS_GRAD_NT), lds, stream, a);
}

template <int FEAT>
void gradient_launch(const GradientArgs& a, int grid, size_t lds, hipStream_t stream, bool wide) {
    if (a.src.sensor) {
        {
            if (wide) {
                note_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gradient<FEAT, 0, true, 1024>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute(dynamic LDS)");
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gradient<FEAT, 0, true, 1024>), dim3(grid), dim3(1024), lds, stream, a);
                return;
            }
            note_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gradient<FEAT, 0, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute(dynamic LDS)");
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gradient<FEAT, 0, true>), dim3(grid), dim3(NLOS_GRAD_NT), lds, stream, a);
        }
        return;
    }
    switch (a.mode) {
        case 0: gradient_launch2<FEAT, 0>(a, grid, lds, stream, wide); break;
        case 1: gradient_launch2<FEAT, 1>(a, grid, lds, stream); break;
        case 2: gradient_launch2<FEAT, 2>(a, grid, lds, stream); break;
        case 4:
            if constexpr ((FEAT & (FEAT_GGX | FEAT_ALB)) == 0) gradient_launch2<FEAT, 4>(a, grid, lds, stream);
            break;
        default: gradient_launch2<FEAT, 3>(a, grid, lds, stream); break;
    }
}

}  // namespace

void launch_items_to_words(const unsigned long long* items, int items_stride, const uint16_t* live, int L, int F, int spt,
                           uint32_t* words, hipStream_t stream) {
    if (L <= 0) return;
    hipLaunchKernelGGL(k_items_to_words, dim3(L), dim3(256), 0, stream, items, items_stride, live, F, spt, words);
}

void launch_gradient(const GradientArgs& a_in, hipStream_t stream) {
    if (a_in.src.L <= 0) return;
    GradientArgs a = a_in;
    const size_t nblk = ((size_t)a.sc.F + 63) / 64;
    size_t lds = 8 + ((size_t)a.sp.nbins + 3 * (size_t)a.K + 2) * sizeof(double) + nblk * 8 + ((nblk + 2) & ~(size_t)1) * 4;
    // per-bin tap weights beside the prefix sums (TapTables::wt) while the table is small (refine 10, sigma_bin 1: 880 B)
    a.lean_params = lean_params_ok(a.sp) ? 1 : 0;
    a.wt_in_lds = 0;
    if (a.mode == 0 && a.tap_wt && a.tap_nb > 0 && a.K > 1) {
        const size_t wt = sizeof(double) * 2 * (size_t)(a.refine + 1) * (size_t)a.tap_nb;
        if (wt <= 4096) { a.wt_in_lds = 1; lds += wt; }
    }
    // compacted list of the faces with accepted samples (u16): skipped for meshes it cannot index / hold
    a.compact = (a.sc.F <= 65535 && lds + 2 * (size_t)a.sc.F + 16 <= 64 * 1024) ? 1 : 0;
    if (a.compact) lds += (2 * (size_t)a.sc.F + 15) & ~(size_t)15;
    // per-workgroup 3V-double accumulator while it fits beside the rest (one workgroup per CU at worst)
    const size_t acc = 3 * (size_t)a.sc.V * sizeof(double);
    a.lds_grad = ((a.mode == 0 || a.mode == 4) && a_in.lds_grad && lds + acc <= 150 * 1024) ? 1 : 0;
    if (a.inline_residual && !a.lds_grad && (a.mode == 0 || a.mode == 4) && a_in.lds_grad) {
        // the face-major kernel reads residual rows: form them first
        ResidualArgs ra;
        std::memset(&ra, 0, sizeof(ra));
        ra.data = a.res_data; ra.weight = a.res_weight; ra.transient = a.res_transient; ra.diff = a.diff_scratch;
        ra.L = a.src.L; ra.T = a.sp.nbins; ra.loss_test = a.res_loss_test;
        launch_residual(ra, stream);
        a.diff = a.diff_scratch; a.inline_residual = 0;
    }
    if (!a.lds_grad && (a.mode == 0 || a.mode == 4) && a_in.lds_grad) {
        // large meshes: face-major variant (per-face sums in LDS across sources, one scatter per face)
        if (a.vis_items && a.vis_scratch) {
            // it indexes the cache by face: per-face words out of the item masks first
            launch_items_to_words(a.vis_items, a.items_stride, a.live, a.src.L, a.sc.F, a.sp.spt, a.vis_scratch, stream);
            a.vis = a.vis_scratch; a.vis_words = 1; a.vis_items = nullptr;
        }
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
        if (done) { if (tl_note) tl_note->gradient_kernel = 3; return; }
    }
    if (tl_note) tl_note->gradient_kernel = a.lds_grad ? 1 : 2;
    if (a.lds_grad) lds += acc;
    // persistent workgroups: as many as can be co-resident (512 threads each, <= 128 VGPRs -> 4 per CU)
    int per_cu = (int)(160 * 1024 / (lds + 64));
    if (per_cu > 4) per_cu = 4;
    if (per_cu < 1) per_cu = 1;
    int grid = device_cu_count() * per_cu;
    if (grid > a.src.L) grid = a.src.L;
    // few sources (a rank's share of a strong split): a workgroup that owns ONE source pays the zeroing and the flush of its
    // 3V-double accumulator for that one source; NLOS_GRAD_MIN_SOURCES > 1 hands every workgroup at least that many
    // (fewer, longer workgroups).  Measured in round 6 (profiles/r06_ab_grad_min_sources.log); default from there.
    const int min_src = env_switches().grad_min_sources;
    if (a.lds_grad && min_src > 1 && a.src.L / min_src >= 1 && grid > a.src.L / min_src) grid = a.src.L / min_src;
    // a single workgroup per CU: give it the sixteen waves two workgroups would have had
    const int wide_ok = env_switches().grad_wide;
    const bool wide = per_cu == 1 && a.mode == 0 && wide_ok;
    switch (feat_of(a.sc, a.sp)) {
        case 0: gradient_launch<0>(a, grid, lds, stream, wide); break;
        case 1: gradient_launch<1>(a, grid, lds, stream, wide); break;
        case 2: gradient_launch<2>(a, grid, lds, stream, wide); break;
        case 3: gradient_launch<3>(a, grid, lds, stream, wide); break;
        case 4: gradient_launch<4>(a, grid, lds, stream, wide); break;
        case 5: gradient_launch<5>(a, grid, lds, stream, wide); break;
        case 6: gradient_launch<6>(a, grid, lds, stream, wide); break;
        default: gradient_launch<7>(a, grid, lds, stream, wide); break;
    }
}

}  // namespace nlos
