// bvh_build.hip -- single-launch LBVH builder for gfx950.
//
// Replaces the per-call Embree scene build of the reference
// (smoothed_transient/stratifiedStreamedGradientRenderer.cpp:473-511:
// rtcNewScene / RTC_BUILD_QUALITY_HIGH / rtcCommitScene every call).  The mesh
// moves every optimisation step, so the build must be cheap and stay on the
// device: one 1024-thread workgroup runs all phases back to back
//   bounds -> 30-bit Morton keys -> LDS-counted radix sort (10 x 3 bit) ->
//   Karras radix tree -> bottom-up box refit -> DFS pre-order emission
// with workgroup barriers between phases (no host round trip, one launch,
// graph-capturable).  Output is a stackless BVH: 32-byte nodes in pre-order with
// escape indices, 48-byte triangle records and 64-byte face records in Morton
// order (see nlos_device.h).  Morton order is also the order in which the render
// kernels hand faces to lanes, so the rays of a wave are spatially coherent.
#include "nlos_device.h"
#include "nlos_kernels.h"

namespace nlos {

namespace {

constexpr int BT = 1024;          // build threads (one workgroup)
constexpr int RBITS = 3;          // radix bits per pass
constexpr int RDIG = 1 << RBITS;  // 8 digits
constexpr int RPASS = 10;         // 30-bit keys

__device__ __forceinline__ uint32_t expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__device__ __forceinline__ int delta(const uint32_t* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    uint32_t a = keys[i], b = keys[j];
    if (a == b) return 32 + __clz((uint32_t)(i ^ j));
    return __clz(a ^ b);
}

__device__ __forceinline__ int clamp_index(int v, int nV, int* status) {
    if (v < 0 || v >= nV) { atomicOr(status, 1); return 0; }
    return v;
}

}  // namespace

__global__ __launch_bounds__(1024) void k_build_bvh(BuildArgs a) {
    __shared__ uint32_t s_cnt[RDIG * BT];     // 32 KB radix counters [digit][thread]
    __shared__ uint32_t s_part[BT];
    __shared__ float s_red[6 * 16];           // per-wave bounds
    __shared__ float s_bounds[8];             // lo[3], hi[3], pad

    const int tid = threadIdx.x;
    const int F = a.F;
    const int lane = tid & 63, wave = tid >> 6;

    // ---- phase 1: centroid bounds + scene extent -------------------------------
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    float ext = 0.0f;
    for (int f = tid; f < F; f += BT) {
        for (int k = 0; k < 3; ++k) {
            int vi = clamp_index(a.faces[3 * f + k], a.V, a.status);
            for (int c = 0; c < 3; ++c) {
                float x = a.vertices[3 * (size_t)vi + c];
                lo[c] = fminf(lo[c], x);
                hi[c] = fmaxf(hi[c], x);
                ext = fmaxf(ext, fabsf(x));
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        for (int c = 0; c < 3; ++c) {
            lo[c] = fminf(lo[c], __shfl_down(lo[c], off));
            hi[c] = fmaxf(hi[c], __shfl_down(hi[c], off));
        }
    }
    if (lane == 0)
        for (int c = 0; c < 3; ++c) { s_red[wave * 6 + c] = lo[c]; s_red[wave * 6 + 3 + c] = hi[c]; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < BT / 64; ++w)
            for (int c = 0; c < 3; ++c) {
                lo[c] = fminf(lo[c], s_red[w * 6 + c]);
                hi[c] = fmaxf(hi[c], s_red[w * 6 + 3 + c]);
            }
        float e = 0.0f;
        for (int c = 0; c < 3; ++c) {
            s_bounds[c] = lo[c];
            s_bounds[3 + c] = hi[c];
            e = fmaxf(e, fmaxf(fabsf(lo[c]), fabsf(hi[c])));
        }
        s_bounds[6] = 1e-4f * e + 1e-30f;      // box padding >> fp32 rounding of hit points
    }
    __syncthreads();
    const float pad = s_bounds[6];

    // ---- phase 2: Morton keys ---------------------------------------------------
    uint32_t* keys_in = a.keys0;
    uint32_t* keys_out = a.keys1;
    int* idx_in = a.idx0;
    int* idx_out = a.idx1;
    {
        float sx = s_bounds[3] - s_bounds[0], sy = s_bounds[4] - s_bounds[1], sz = s_bounds[5] - s_bounds[2];
        float ix = sx > 0 ? 1.0f / sx : 0.0f, iy = sy > 0 ? 1.0f / sy : 0.0f, iz = sz > 0 ? 1.0f / sz : 0.0f;
        for (int f = tid; f < F; f += BT) {
            float c[3] = {0, 0, 0};
            for (int k = 0; k < 3; ++k) {
                int vi = clamp_index(a.faces[3 * f + k], a.V, a.status);
                for (int q = 0; q < 3; ++q) c[q] += a.vertices[3 * (size_t)vi + q];
            }
            float nx = (c[0] * (1.0f / 3.0f) - s_bounds[0]) * ix;
            float ny = (c[1] * (1.0f / 3.0f) - s_bounds[1]) * iy;
            float nz = (c[2] * (1.0f / 3.0f) - s_bounds[2]) * iz;
            uint32_t qx = (uint32_t)fminf(fmaxf(nx * 1024.0f, 0.0f), 1023.0f);
            uint32_t qy = (uint32_t)fminf(fmaxf(ny * 1024.0f, 0.0f), 1023.0f);
            uint32_t qz = (uint32_t)fminf(fmaxf(nz * 1024.0f, 0.0f), 1023.0f);
            keys_in[f] = (expand_bits(qx) << 2) | (expand_bits(qy) << 1) | expand_bits(qz);
            idx_in[f] = f;
        }
    }
    __syncthreads();

    // ---- phase 3: stable LSD radix sort, one private counter column per thread ----
    const int chunk = (F + BT - 1) / BT;
    const int c0 = min(tid * chunk, F), c1 = min(c0 + chunk, F);
    for (int pass = 0; pass < RPASS; ++pass) {
        const int shift = pass * RBITS;
        for (int d = 0; d < RDIG; ++d) s_cnt[d * BT + tid] = 0;
        for (int i = c0; i < c1; ++i) s_cnt[((keys_in[i] >> shift) & (RDIG - 1)) * BT + tid] += 1;
        __syncthreads();
        // exclusive scan of s_cnt in (digit, thread) order: thread t owns entries [8t, 8t+8)
        uint32_t loc[RDIG];
        uint32_t sum = 0;
        for (int q = 0; q < RDIG; ++q) { loc[q] = sum; sum += s_cnt[tid * RDIG + q]; }
        s_part[tid] = sum;
        __syncthreads();
        for (int off = 1; off < BT; off <<= 1) {
            uint32_t v = tid >= off ? s_part[tid - off] : 0;
            __syncthreads();
            s_part[tid] += v;
            __syncthreads();
        }
        uint32_t base = s_part[tid] - sum;
        for (int q = 0; q < RDIG; ++q) s_cnt[tid * RDIG + q] = base + loc[q];
        __syncthreads();
        for (int i = c0; i < c1; ++i) {
            uint32_t k = keys_in[i];
            uint32_t dst = s_cnt[((k >> shift) & (RDIG - 1)) * BT + tid]++;
            keys_out[dst] = k;
            idx_out[dst] = idx_in[i];
        }
        __syncthreads();
        uint32_t* tk = keys_in; keys_in = keys_out; keys_out = tk;
        int* ti = idx_in; idx_in = idx_out; idx_out = ti;
    }
    // sorted data is in keys_in / idx_in (RPASS even -> back in keys0 / idx0)
    const uint32_t* keys = keys_in;
    const int* order = idx_in;

    // ---- phase 4: Karras radix tree ------------------------------------------------
    // temp node ids: internal i -> i (0..F-2), leaf j -> F-1+j
    const int n_int = F - 1;
    for (int i = tid; i < n_int; i += BT) {
        int d = (delta(keys, F, i, i + 1) - delta(keys, F, i, i - 1)) >= 0 ? 1 : -1;
        int dmin = delta(keys, F, i, i - d);
        int lmax = 2;
        while (delta(keys, F, i, i + lmax * d) > dmin) lmax <<= 1;
        int l = 0;
        for (int t = lmax >> 1; t >= 1; t >>= 1)
            if (delta(keys, F, i, i + (l + t) * d) > dmin) l += t;
        int j = i + l * d;
        int dnode = delta(keys, F, i, j);
        int s = 0, t = l;
        do {
            t = (t + 1) >> 1;
            if (delta(keys, F, i, i + (s + t) * d) > dnode) s += t;
        } while (t > 1);
        int gamma = i + s * d + min(d, 0);
        int first = min(i, j), last = max(i, j);
        int left = (first == gamma) ? (n_int + gamma) : gamma;
        int right = (last == gamma + 1) ? (n_int + gamma + 1) : (gamma + 1);
        a.child[2 * i] = left;
        a.child[2 * i + 1] = right;
        a.range[2 * i] = first;
        a.range[2 * i + 1] = last;
        a.parent[left] = i;
        a.parent[right] = i;
        a.arrive[i] = 0;
    }
    if (tid == 0) a.parent[F > 1 ? 0 : n_int] = -1;
    __syncthreads();

    // ---- phase 5: leaf records + bottom-up refit -------------------------------------
    for (int j = tid; j < F; j += BT) {
        int f = order[j];
        int i0 = clamp_index(a.faces[3 * f], a.V, a.status);
        int i1 = clamp_index(a.faces[3 * f + 1], a.V, a.status);
        int i2 = clamp_index(a.faces[3 * f + 2], a.V, a.status);
        V3 p0 = ld3(a.vertices + 3 * (size_t)i0);
        V3 p1 = ld3(a.vertices + 3 * (size_t)i1);
        V3 p2 = ld3(a.vertices + 3 * (size_t)i2);
        Tri tr = make_tri(p0, p1, p2);
        a.tris[3 * j] = make_float4(tr.p0.x, tr.p0.y, tr.p0.z, tr.e1.x);
        a.tris[3 * j + 1] = make_float4(tr.e1.y, tr.e1.z, tr.e2.x, tr.e2.y);
        a.tris[3 * j + 2] = make_float4(tr.e2.z, tr.ng.x, tr.ng.y, tr.ng.z);
        a.facerec[4 * j] = make_float4(p0.x, p0.y, p0.z, p1.x);
        a.facerec[4 * j + 1] = make_float4(p1.y, p1.z, p2.x, p2.y);
        a.facerec[4 * j + 2] = make_float4(p2.z, __int_as_float(f), __int_as_float(i0), __int_as_float(i1));
        a.facerec[4 * j + 3] = make_float4(__int_as_float(i2), 0.0f, 0.0f, 0.0f);
        a.face_id[j] = f;
        a.tri_zmin[j] = fminf(fminf(p0.z, p1.z), p2.z);
        float* b = a.box + 6 * (size_t)(n_int + j);
        b[0] = fminf(fminf(p0.x, p1.x), p2.x) - pad;
        b[1] = fminf(fminf(p0.y, p1.y), p2.y) - pad;
        b[2] = fminf(fminf(p0.z, p1.z), p2.z) - pad;
        b[3] = fmaxf(fmaxf(p0.x, p1.x), p2.x) + pad;
        b[4] = fmaxf(fmaxf(p0.y, p1.y), p2.y) + pad;
        b[5] = fmaxf(fmaxf(p0.z, p1.z), p2.z) + pad;
        __threadfence();
        int node = a.parent[n_int + j];
        while (node >= 0) {
            int old = atomicAdd(&a.arrive[node], 1);
            if (old == 0) break;                   // sibling subtree not finished yet
            __threadfence();
            const float* bl = a.box + 6 * (size_t)a.child[2 * node];
            const float* br = a.box + 6 * (size_t)a.child[2 * node + 1];
            float* bo = a.box + 6 * (size_t)node;
            // the sibling's box was written by another wave: read it through the L2
            for (int c = 0; c < 6; ++c) {
                float x = __hip_atomic_load(bl + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                float y = __hip_atomic_load(br + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bo[c] = c < 3 ? fminf(x, y) : fmaxf(x, y);
            }
            __threadfence();
            node = a.parent[node];
        }
    }
    __syncthreads();

    // ---- phase 6: DFS pre-order emission ---------------------------------------------
    // pre(node) = 2 * first_leaf(node) + (number of ancestors entered through their LEFT child)
    const int n_nodes = 2 * F - 1;
    for (int t = tid; t < n_nodes; t += BT) {
        bool leaf = t >= n_int;
        int first = leaf ? (t - n_int) : a.range[2 * t];
        int nleaves = leaf ? 1 : (a.range[2 * t + 1] - a.range[2 * t] + 1);
        int lt = 0;
        int node = t, p = a.parent[t];
        while (p >= 0) {
            if (a.child[2 * p] == node) ++lt;
            node = p;
            p = a.parent[p];
        }
        int pre = 2 * first + lt;
        int esc = pre + 2 * nleaves - 1;
        const float* b = a.box + 6 * (size_t)t;
        a.nodes[2 * pre] = make_float4(b[0], b[1], b[2], b[3]);
        a.nodes[2 * pre + 1] = make_float4(b[4], b[5], __int_as_float(esc), __int_as_float(leaf ? first : -1));
    }
}

void launch_build_bvh(const BuildArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(k_build_bvh, dim3(1), dim3(BT), 0, stream, a);
}

}  // namespace nlos
