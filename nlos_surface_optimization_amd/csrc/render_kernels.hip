// render_kernels.hip -- the small kernels around the two passes and the pass-1 dispatcher (gfx950).
//
//   k_smooth    <- refined-histogram Gaussian + fold (smoothed_transient/transient_and_gradient.cpp:348-371);
//                  also the jitter kernel's convolution (jitter/transient_and_gradient.cpp:331-347)
//   k_residual  <- difference = (data - transient)[^3*2] * weight
//                  (smoothed_transient/stratifiedStreamedGradientRenderer.cpp:543-550)
//   k_boxfilter <- v1 residual smoothing
//                  (stratified_transient_raytracer/stratifiedStreamedGradientRenderer.cpp:447-462)
//   k_intersect, k_bary_to_world <- embree_intersector/c_embree_intersector.cpp:20-104
// Pass 1 lives in forward_grid.hip (fast path) and forward_bvh.hip, pass 2 in gradient.hip, the shared
// per-sample code in render_common.h.
//
// Mapping of the two passes to the hardware (MI355X, wave64):
//   * one workgroup per source: every ray of the workgroup starts at the same wall point, the histogram
//     row of that source lives in LDS (ds_add_f64) and is written back once with coalesced stores;
//   * one lane per face (Morton order), looping over that face's `spt` strata; the nine per-vertex
//     gradient sums of a (source, face) pair are reduced in registers before touching LDS;
//   * waves pull 64-face blocks from an LDS ticket counter;
//   * the visibility of every accepted sample is cached as one bit by pass 1; pass 2 never traces a ray
//     (the reference traces all rays twice);
//   * samples whose clamped form factor is zero contribute exactly 0 to both passes, so their rays are
//     never traced at all.
#include "render_common.h"

#include <cstdlib>

namespace nlos {
namespace {

// --------------------------------------------------------------------- smooth
// transient[t] = sum_q y[t R + q + half] with y = fine (*) kernel (full convolution), regrouped per input
// sample: transient[t] = sum_i fine[i] W[t R + half - i],  W[m] = sum_{q < R} kernel[m + q] -- K + R - 1
// multiply-adds per output bin instead of K R (201 x 10 at sigma_bin = 5: 2.46 -> see DESIGN ms for 1024 rows),
// the row and W staged in LDS.  Same terms as the literal double loop, summed in another order (fp64).
__global__ __launch_bounds__(256) void k_smooth(SmoothArgs a, int row_in_lds) {
    extern __shared__ double s_sm[];          // [W: K + R - 1][fine row: T R when it fits]
    const int l = blockIdx.x;
    const int R = a.refine, K = a.K, half = a.offset;
    const int rb = a.T * R;
    const int nw = K + R - 1;
    double* s_w = s_sm;
    double* s_f = s_sm + nw;
    const double* gfine = a.fine + (size_t)l * rb;
    for (int m = threadIdx.x; m < nw; m += blockDim.x) {
        double w = 0.0;
        for (int q = 0; q < R; ++q) {
            const int j = m - (R - 1) + q;
            if (j >= 0 && j < K) w += a.kernel[j];
        }
        s_w[m] = w;                           // W[m - (R - 1)]
    }
    if (row_in_lds)
        for (int i = threadIdx.x; i < rb; i += blockDim.x) s_f[i] = gfine[i];
    __syncthreads();
    const double* fine = row_in_lds ? s_f : gfine;
    for (int t = threadIdx.x; t < a.T; t += blockDim.x) {
        const int c = t * R + half;
        const int i0 = max(0, c - (K - 1)), i1 = min(rb - 1, c + R - 1);
        double acc = 0.0;
        for (int i = i0; i <= i1; ++i) acc += fine[i] * s_w[c - i + R - 1];
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
    if (a.zero)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.zero_n; i += stride) a.zero[i] = 0.0;
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

// ------------------------------------------------------------ row N as a product
// Combine kernel of the L x S product (include/nlos_hip.h, nlos_render_args.n_sensors): one workgroup per (laser i,
// sensor j), its row in LDS.  The record pass (forward_grid.hip, NCM = 3) left, per wall point and sample r = sorted face
// slot * spt + s, the leg's path length and clamped form factor (0: not seen from there); a pair accepts the sample iff
// both legs did and d1 + d2 lies in the window -- the expressions of the pair kernels (k_forward_grid<FEAT, 2>,
// sample_geo_nc) term for term: val = (area ffa) ffb, bin = floor(((d1 + d2) - lb) / res), (double)val / spt.
// One lane per face, its spt strata in turn; the accepted-sample word of (pair, face) is what pass 2 reads.
// EXT (round 6): scenes with vertex normals / albedo.  The pair's normal and albedo are the laser leg's (interpolated at ITS hit,
// sample_geo_nc()), so the laser record carries them and its own clamped form factor, the sensor record the leg's unit
// direction and length; the sensor's form factor is formed here, with sample_geo_nc()'s / k_forward_grid<FEAT, 2>'s expressions
// term for term: numb = -dot(n, dirB) * dot(onb, dirB), ffb = max(0, numb / d2 / d2), val = ((area * alb) * ffa) * ffb.
template <bool EXT>
__global__ __launch_bounds__(256) void k_product_combine(ProductArgs a) {
    extern __shared__ double s_prow[];
    const int pair = blockIdx.x;
    const int i = pair / a.Sb, j = pair - i * a.Sb;
    const int F = a.sc.F, spt = a.spt, nbins = a.nbins;
    for (int b = threadIdx.x; b < nbins; b += blockDim.x) s_prow[b] = 0.0;
    __syncthreads();
    const size_t R = (size_t)F * (size_t)spt;
    const float* __restrict__ dA = a.d_a + (size_t)i * R;
    const float* __restrict__ fA = a.ff_a + (size_t)i * R;
    const float* __restrict__ dB = a.d_b + (size_t)j * R;
    const float* __restrict__ fB = a.ff_b + (size_t)j * R;
    const double inv_spt = 1.0 / (double)spt;
    const float* __restrict__ xA = EXT ? a.ext_a + (size_t)i * R : nullptr;
    const float* __restrict__ xB = EXT ? a.ext_b + (size_t)j * R : nullptr;
    const size_t es = a.ext_stride;
    const V3 onb = EXT ? ld3(a.sensor_normal + 3 * (size_t)j) : mk(0.0f, 0.0f, 1.0f);
    for (int jf = threadIdx.x; jf < F; jf += blockDim.x) {
        uint32_t word = 0u;
        const size_t r0 = (size_t)jf * (size_t)spt;
        float area = 0.0f;
        for (int s = 0; s < spt; ++s) {
            const float ffa = fA[r0 + s];
            if (!(ffa > 0.0f)) continue;
            float ffb, alb = 1.0f;
            const float d2 = dB[r0 + s];
            if (EXT) {
                if (!(d2 > 0.0f)) continue;                                  // not seen from the sensor
                const size_t at = r0 + s;
                const V3 n = mk(xA[3 * es + at], xA[4 * es + at], xA[5 * es + at]);
                const V3 dirB = mk(xB[at], xB[es + at], xB[2 * es + at]);
                const float numb = -dot(n, dirB) * dot(onb, dirB);
                ffb = emax0(form_factor<false>(numb, d2));
                alb = xA[6 * es + at];
            } else {
                ffb = fB[r0 + s];
            }
            if (!(ffb > 0.0f)) continue;
            const float d1 = dA[r0 + s];
            const float tot = d1 + d2;
            if (!((tot <= a.ub) && (tot >= a.lb))) continue;
            word |= 1u << s;
            if (area == 0.0f) area = a.sc.tris[kTriStride * jf + 3].z;
            const int bin = (int)floorf((tot - a.lb) / a.res);
            if (bin < 0 || bin >= nbins) continue;
            const float val = EXT ? area * alb * ffa * ffb : area * ffa * ffb;
            lds_add_f64(&s_prow[bin], (double)val * inv_spt);
        }
        if (a.vis) a.vis[(size_t)pair * F + jf] = word;
    }
    __syncthreads();
    double* row = a.rows + (size_t)pair * nbins;
    for (int b = threadIdx.x; b < nbins; b += blockDim.x) row[b] = s_prow[b];
}

// the enumerated pairs of a product, for the renders the record pass does not carry: pair p = (laser p / Sb, sensor p % Sb)
__global__ __launch_bounds__(256) void k_expand_pairs(const float* __restrict__ laser, const float* __restrict__ lnormal,
                                                      const float* __restrict__ sensor, const float* __restrict__ snormal,
                                                      int La, int Sb, float* __restrict__ out_l, float* __restrict__ out_ln,
                                                      float* __restrict__ out_s, float* __restrict__ out_sn) {
    const size_t n = 3 * (size_t)La * (size_t)Sb;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) {
        const size_t p = k / 3, c = k - 3 * p;
        const size_t i = p / (size_t)Sb, j = p - i * (size_t)Sb;
        out_l[k] = laser[3 * i + c]; out_ln[k] = lnormal[3 * i + c];
        out_s[k] = sensor[3 * j + c]; out_sn[k] = snormal[3 * j + c];
    }
}

__global__ __launch_bounds__(256) void k_zero_f64(double* p, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = 0.0;
}

}  // namespace

thread_local LaunchNote* tl_note = nullptr;

// Diagnostics (nlos_ctx_debug_read what = 3, tools/soak.py): a position-dependent 2 x 64-bit digest of `n` 32-bit words --
// of the accepted-sample words per (source, face), which do not depend on the order in which workgroups, waves and atomics
// happened to run: two renders of the same inputs must give the same digest.
__global__ __launch_bounds__(256) void k_digest_u32(const uint32_t* __restrict__ w, size_t n, unsigned long long* __restrict__ out) {
    unsigned long long a = 0ull, b = 0ull;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = ((unsigned long long)w[i] + 0x9E3779B97F4A7C15ull) * (2ull * i + 1ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        a += z ^ (z >> 31);
        b ^= z * (i + 0x632BE59BD9B4E019ull);
    }
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_xor(a, off);
        b ^= __shfl_xor(b, off);
    }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], a); atomicXor(&out[1], b); }
}
// One workgroup: bounding box of the source positions, 3 x 10-bit Morton keys, bitonic sort of (key, index) in LDS.
__device__ __forceinline__ uint32_t spread10(uint32_t x) {
    x &= 0x3FFu;
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}
__global__ __launch_bounds__(1024) void k_order_sources(const float* __restrict__ origin, int L, int n_pad, int* __restrict__ perm) {
    extern __shared__ unsigned long long s_key[];
    __shared__ int s_lo[3], s_hi[3];           // ordered-integer images of the floats
    if (threadIdx.x < 3) { s_lo[threadIdx.x] = 0x7FFFFFFF; s_hi[threadIdx.x] = (int)0x80000000; }
    __syncthreads();
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int i = threadIdx.x; i < L; i += blockDim.x)
        for (int c = 0; c < 3; ++c) { const float x = origin[3 * (size_t)i + c]; lo[c] = fminf(lo[c], x); hi[c] = fmaxf(hi[c], x); }
    for (int c = 0; c < 3; ++c) {
        for (int off = 32; off > 0; off >>= 1) { lo[c] = fminf(lo[c], __shfl_xor(lo[c], off)); hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], off)); }
        if ((threadIdx.x & 63) == 0) {
            // (float min / max through the ordered-integer image: positive and negative values alike)
            auto enc = [](float f) { const int b = __float_as_int(f); return b >= 0 ? b : b ^ 0x7FFFFFFF; };
            atomicMin(&s_lo[c], enc(lo[c]));
            atomicMax(&s_hi[c], enc(hi[c]));
        }
    }
    __syncthreads();
    float blo[3], inv[3];
    for (int c = 0; c < 3; ++c) {
        auto dec = [](int b) { return __int_as_float(b >= 0 ? b : b ^ 0x7FFFFFFF); };
        blo[c] = dec(s_lo[c]);
        const float ext = dec(s_hi[c]) - blo[c];
        inv[c] = ext > 0.0f ? 1023.0f / ext : 0.0f;
    }
    for (int i = threadIdx.x; i < n_pad; i += blockDim.x) {
        unsigned long long k = ~0ull;                                   // padding sorts behind every real source
        if (i < L) {
            uint32_t q[3];
            for (int c = 0; c < 3; ++c) {
                const float t = (origin[3 * (size_t)i + c] - blo[c]) * inv[c];
                q[c] = (uint32_t)min(max((int)t, 0), 1023);            // (NaN -> 0)
            }
            const uint32_t m = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
            k = ((unsigned long long)m << 32) | (unsigned long long)(uint32_t)i;
        }
        s_key[i] = k;
    }
    __syncthreads();
    for (int k = 2; k <= n_pad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n_pad; i += blockDim.x) {
                const int p = i ^ j;
                if (p > i) {
                    const unsigned long long x = s_key[i], y = s_key[p];
                    const bool asc = (i & k) == 0;
                    if (asc ? x > y : x < y) { s_key[i] = y; s_key[p] = x; }
                }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < L; i += blockDim.x) perm[i] = (int)(uint32_t)(s_key[i] & 0xFFFFFFFFull);
}
bool launch_order_sources(const float* origin, int L, int* perm, hipStream_t stream) {
    if (L < 2 || L > 8192) return false;
    int n_pad = 2;
    while (n_pad < L) n_pad <<= 1;
    const size_t lds = (size_t)n_pad * sizeof(unsigned long long);
    note_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_order_sources), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
             "hipFuncSetAttribute(k_order_sources)");
    hipLaunchKernelGGL(k_order_sources, dim3(1), dim3(1024), lds, stream, origin, L, n_pad, perm);
    return true;
}
void launch_digest_u32(const uint32_t* w, size_t n, unsigned long long* out, hipStream_t stream) {
    note_hip(hipMemsetAsync(out, 0, 16, stream), "hipMemsetAsync(digest)");
    hipLaunchKernelGGL(k_digest_u32, dim3(1024), dim3(256), 0, stream, w, n, out);
}

const EnvSwitches& env_switches() {
    static const EnvSwitches e = [] {
        auto geti = [](const char* n, long long d) { const char* v = std::getenv(n); return v ? std::atoll(v) : d; };
        EnvSwitches s;
        s.tile_threshold = (int)geti("NLOS_TILE_THRESHOLD", 6200);
        s.lazy_tree = geti("NLOS_LAZY_TREE", 1) != 0;
        s.fuse_residual = geti("NLOS_FUSE_RESIDUAL", 1) != 0;
        const long long tt = geti("NLOS_TILE_TRIS", 3000);
        s.tile_tris = tt > 0 ? (int)tt : 3000;
        const char* sm = std::getenv("NLOS_TILE_SCRATCH_MAX");
        const unsigned long long smv = sm ? std::strtoull(sm, nullptr, 10) : 0ull;
        s.tile_scratch_max = smv > 0 ? smv : (32ull << 30);
        s.vis_items = geti("NLOS_VIS_ITEMS", 1) != 0;
        s.geo_cache = geti("NLOS_GEO_CACHE", 1) != 0;
        const char* gm = std::getenv("NLOS_GEO_CACHE_MAX_GB");
        s.geo_cache_max_gb = gm ? std::atof(gm) : -1.0;
        s.row_lds_max = (size_t)geti("NLOS_ROW_LDS_MAX", 10 * 1024);
        s.grad_wide = (int)geti("NLOS_GRAD_WIDE", 1);
        s.grad_min_sources = (int)geti("NLOS_GRAD_MIN_SOURCES", 1);
        s.fwd_order = geti("NLOS_FWD_ORDER", 1) != 0;
        s.geo_max_spt = (int)geti("NLOS_GEO_MAX_SPT", 8);
        return s;
    }();
    return e;
}

void launch_forward(const ForwardArgs& a, hipStream_t stream) {
    if (a.src.L <= 0) return;
    const size_t row_bytes = (size_t)a.sp.nbins * sizeof(double);
    // the histogram row of a source lives in LDS while it leaves room for the grid (<= 10 KB: 1280 bins);
    // longer rows are accumulated with global atomics -- only ~12 k accepted samples per source land in them
    // (measured: 2048 bins 5.4 ms with the row in LDS and the grid squeezed, 2.6 ms this way).  Round 6: the bound went from
    // 9 KB to 10 KB to cover the 1 200 bins every experiment script of the reference uses (exp_bunny/test.py:33-34): bunny_5k
    // 64x64x1200 forward 1.582 -> 1.391 ms with the row in LDS (64 of 4 096 sources coarsen their grid for the 400 bytes);
    // at 1 536 bins the row in LDS loses (1.532 -> 1.634 ms, 2 027 sources coarsen): profiles/r06_row_lds_max.log
    const size_t row_lds_max = env_switches().row_lds_max;
    const int rows_in_lds = (!a.mode_intensity && row_bytes <= row_lds_max) ? 1 : 0;
    if (!rows_in_lds && !a.mode_intensity) launch_zero_f64(a.rows, (size_t)a.src.L * a.sp.nbins, stream);
    if (tl_note) tl_note->rows_in_lds = rows_in_lds;
    if (launch_forward_grid(a, rows_in_lds, stream)) return;
    if (tl_note) tl_note->backend = 3;       // NLOS_PATH_BVH; the reason was recorded by the grid dispatcher
    if (tl_note && tl_note->lazy_build && !tl_note->tree_built) {
        // the scene was built for the grid (records only): the BVH back-end needs the tree after all
        launch_build_tree(*tl_note->lazy_build, false, stream);
        tl_note->tree_built = true;
    }
    launch_forward_bvh(a, rows_in_lds, stream);
}

void launch_smooth(const SmoothArgs& a, hipStream_t stream) {
    if (a.L <= 0) return;
    const size_t wbytes = ((size_t)a.K + a.refine - 1) * sizeof(double);
    const size_t row = (size_t)a.T * a.refine * sizeof(double);
    const int row_in_lds = wbytes + row <= 64 * 1024 ? 1 : 0;       // two or more workgroups per CU; W alone is <= 16 KB (API limit of 2048 taps)
    const size_t lds = wbytes + (row_in_lds ? row : 0);
    note_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_smooth), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
             "hipFuncSetAttribute(k_smooth)");
    hipLaunchKernelGGL(k_smooth, dim3(a.L), dim3(256), lds, stream, a, row_in_lds);
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

void launch_product_combine(const ProductArgs& a, hipStream_t stream) {
    if (a.La <= 0 || a.Sb <= 0) return;
    const size_t lds = (size_t)a.nbins * sizeof(double);
    if (a.ext_a && a.ext_b) {
        note_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_product_combine<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                 "hipFuncSetAttribute(k_product_combine)");
        hipLaunchKernelGGL(k_product_combine<true>, dim3((unsigned)((size_t)a.La * a.Sb)), dim3(256), lds, stream, a);
        return;
    }
    note_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_product_combine<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
             "hipFuncSetAttribute(k_product_combine)");
    hipLaunchKernelGGL(k_product_combine<false>, dim3((unsigned)((size_t)a.La * a.Sb)), dim3(256), lds, stream, a);
}

void launch_expand_pairs(const float* laser, const float* lnormal, const float* sensor, const float* snormal, int La, int Sb,
                         float* out_l, float* out_ln, float* out_s, float* out_sn, hipStream_t stream) {
    const size_t n = 3 * (size_t)La * (size_t)Sb;
    if (n == 0) return;
    size_t g = (n + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_expand_pairs, dim3((unsigned)g), dim3(256), 0, stream, laser, lnormal, sensor, snormal, La, Sb, out_l, out_ln,
                       out_s, out_sn);
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

