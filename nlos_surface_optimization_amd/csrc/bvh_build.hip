// bvh_build.hip -- single-launch LBVH builder for gfx950.
//
// Replaces the per-call Embree scene build of the reference
// (smoothed_transient/stratifiedStreamedGradientRenderer.cpp:473-511:
// rtcNewScene / RTC_BUILD_QUALITY_HIGH / rtcCommitScene every call).  The mesh
// moves every optimisation step, so the build must be cheap and stay on the
// device: one 1024-thread workgroup runs all phases back to back
//   bounds -> Morton keys (24 bits in the single-workgroup builder) -> stable LSD radix sort in LDS (4 x 6 bit, wave counters) ->
//   Karras radix tree + escape links -> bottom-up box refit (in LDS while the inner nodes fit) + node emission
// with workgroup barriers between phases (no host round trip, one launch,
// graph-capturable).  Output is a stackless BVH: 32-byte nodes
//   a = (lo.x, lo.y, lo.z, hi.x), b = (hi.y, hi.z, escape, link)
// where link >= 0 is the left child of an inner node and link < 0 marks a leaf
// holding triangle ~link; on a box hit an inner node continues at `link`,
// otherwise (miss, or leaf) at `escape` (-1 = done).  Node ids: inner i in
// [0, F-2] (0 = root), leaf j -> F-1+j.  The escape link needs no top-down pass:
// it is the node whose leaf range starts right after this node's range, i.e.
// inner node s = last+1 if that node's range starts at s, else leaf s.
// Triangle (48 B) and face (64 B) records are emitted in Morton order, which is
// also the order in which the render kernels hand faces to lanes.
#include "nlos_device.h"
#include "nlos_kernels.h"

#include <algorithm>

namespace nlos {

namespace {

constexpr int BT = 1024;          // build threads (one workgroup)
constexpr int RBITS = 4;          // radix bits per pass
constexpr int RDIG = 1 << RBITS;  // 16 digits: the counters (66 KB) leave room for keys and indices in LDS
constexpr int RPASS = 6;          // 24-bit keys (an even number of passes: the result lands in the first buffer)
constexpr int QBITS = 8;          // Morton bits per axis of the single-workgroup builder (<= 5.8 k faces in 256^3 cells; the
                                  // chip-wide front end for larger meshes keeps 10): two radix passes less
constexpr int KF = 6;             // faces per thread the single-workgroup builder keeps in registers (F <= 6144)
constexpr int RCNT_WORDS = RDIG * BT + RDIG * BT / 32 + 32;   // skewed counter table

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

// order-preserving float <-> uint32 map (for LDS integer min/max atomics on box coordinates)
__device__ __forceinline__ uint32_t fkey(float x) {
    const uint32_t u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

__device__ __forceinline__ void emit_node(const BuildArgs& a, int id, const float* b, int esc, int link) {
    a.nodes[2 * id] = make_float4(b[0], b[1], b[2], b[3]);
    a.nodes[2 * id + 1] = make_float4(b[4], b[5], __int_as_float(esc), __int_as_float(link));
}

// Karras (2012) radix-tree node i over the sorted keys: children, covered leaf range, parent links
// LEAN: only what the staged builder reads back (parent links, range starts).  (The three searches of the six nodes of
// a thread run one after the other; running them in lock step -- their LDS probes issued together -- was tried and
// is slower, 48 k -> 90 k cycles: the phase is bound by instruction issue at four waves per SIMD, not by latency.)
template <bool LEAN = false>
__device__ __forceinline__ void karras_node(const BuildArgs& a, const uint32_t* __restrict__ keys, int F, int i,
                                            int& left, int& right, int& last_out) {
    const int n_int = F - 1;
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
    left = (first == gamma) ? (n_int + gamma) : gamma;
    right = (last == gamma + 1) ? (n_int + gamma + 1) : (gamma + 1);
    if (!LEAN) {
        a.child[2 * i] = left;
        a.child[2 * i + 1] = right;
        a.range[2 * i + 1] = last;
        a.arrive[i] = 0;
    }
    a.range[2 * i] = first;
    last_out = last;
    a.parent[left] = i;
    a.parent[right] = i;
}

// escape link of the node covering leaves [.., last]
__device__ __forceinline__ int escape_link(const BuildArgs& a, int F, int last) {
    const int n_int = F - 1;
    const int s = last + 1;
    if (s >= F) return -1;
    if (s < n_int && a.range[2 * s] == s) return s;      // inner node s starts at leaf s
    return n_int + s;                                    // otherwise the leaf itself
}

// the three vertex indices of face f, clamped without side effects (the caller reports `bad` once): loads that are
// followed by a possible atomic cannot be hoisted over one another, and a chain of dependent loads per face is what
// the single-workgroup builder spends its time on
__device__ __forceinline__ void face_indices(const BuildArgs& a, int f, int (&vi)[3], bool& bad) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int v = a.faces[3 * f + q];
        const bool ok = (unsigned)v < (unsigned)a.V;
        bad = bad || !ok;
        vi[q] = ok ? v : 0;
    }
}

// leaf_records() from vertices already in registers
__device__ __forceinline__ void leaf_records_from(const BuildArgs& a, int j, int f, const int (&vi)[3], V3 p0, V3 p1, V3 p2, float pad,
                                                  float (&lb6)[6]) {
    const int i0 = vi[0], i1 = vi[1], i2 = vi[2];
    Tri tr = make_tri(p0, p1, p2);
    a.tris[kTriStride * j] = make_float4(tr.p0.x, tr.p0.y, tr.p0.z, tr.e1.x);
    a.tris[kTriStride * j + 1] = make_float4(tr.e1.y, tr.e1.z, tr.e2.x, tr.e2.y);
    a.tris[kTriStride * j + 2] = make_float4(tr.e2.z, tr.ng.x, tr.ng.y, tr.ng.z);
    const float area = sqrtf(dot(tr.ng, tr.ng)) / 2.0f;
    const float inv2a = 1.0f / (2.0f * area);
    a.tris[kTriStride * j + 3] = make_float4(fminf(fminf(p0.z, p1.z), p2.z), __int_as_float(f), area, inv2a);
    a.facerec[4 * j] = make_float4(p0.x, p0.y, p0.z, p1.x);
    a.facerec[4 * j + 1] = make_float4(p1.y, p1.z, p2.x, p2.y);
    a.facerec[4 * j + 2] = make_float4(p2.z, __int_as_float(f), __int_as_float(i0), __int_as_float(i1));
    // + the unit normal (load_face()'s fn): the grid build's grazing test reads it beside the vertices
    a.facerec[4 * j + 3] = make_float4(__int_as_float(i2), tr.ng.x * inv2a, tr.ng.y * inv2a, tr.ng.z * inv2a);
    a.face_id[j] = f;
    a.tri_zmin[j] = fminf(fminf(p0.z, p1.z), p2.z);
    lb6[0] = fminf(fminf(p0.x, p1.x), p2.x) - pad;
    lb6[1] = fminf(fminf(p0.y, p1.y), p2.y) - pad;
    lb6[2] = fminf(fminf(p0.z, p1.z), p2.z) - pad;
    lb6[3] = fmaxf(fmaxf(p0.x, p1.x), p2.x) + pad;
    lb6[4] = fmaxf(fmaxf(p0.y, p1.y), p2.y) + pad;
    lb6[5] = fmaxf(fmaxf(p0.z, p1.z), p2.z) + pad;
}

// triangle / face records of sorted slot j and its padded box
__device__ __forceinline__ void leaf_records(const BuildArgs& a, const int* __restrict__ order, int j, float pad,
                                             float (&lb6)[6]) {
    int f = order[j];
    int i0 = clamp_index(a.faces[3 * f], a.V, a.status);
    int i1 = clamp_index(a.faces[3 * f + 1], a.V, a.status);
    int i2 = clamp_index(a.faces[3 * f + 2], a.V, a.status);
    V3 p0 = ld3(a.vertices + 3 * (size_t)i0);
    V3 p1 = ld3(a.vertices + 3 * (size_t)i1);
    V3 p2 = ld3(a.vertices + 3 * (size_t)i2);
    Tri tr = make_tri(p0, p1, p2);
    a.tris[kTriStride * j] = make_float4(tr.p0.x, tr.p0.y, tr.p0.z, tr.e1.x);
    a.tris[kTriStride * j + 1] = make_float4(tr.e1.y, tr.e1.z, tr.e2.x, tr.e2.y);
    a.tris[kTriStride * j + 2] = make_float4(tr.e2.z, tr.ng.x, tr.ng.y, tr.ng.z);
    // per-face constants of the sample map, evaluated once per build with load_face()'s own expressions
    // (render_common.h; ng == cross(p1 - p0, p2 - p0) bit for bit): area and 1 / (2 area)
    const float area = sqrtf(dot(tr.ng, tr.ng)) / 2.0f;
    const float inv2a = 1.0f / (2.0f * area);
    a.tris[kTriStride * j + 3] = make_float4(fminf(fminf(p0.z, p1.z), p2.z), __int_as_float(f), area, inv2a);
    a.facerec[4 * j] = make_float4(p0.x, p0.y, p0.z, p1.x);
    a.facerec[4 * j + 1] = make_float4(p1.y, p1.z, p2.x, p2.y);
    a.facerec[4 * j + 2] = make_float4(p2.z, __int_as_float(f), __int_as_float(i0), __int_as_float(i1));
    // + the unit normal (load_face()'s fn): the grid build's grazing test reads it beside the vertices
    a.facerec[4 * j + 3] = make_float4(__int_as_float(i2), tr.ng.x * inv2a, tr.ng.y * inv2a, tr.ng.z * inv2a);
    a.face_id[j] = f;
    a.tri_zmin[j] = fminf(fminf(p0.z, p1.z), p2.z);
    lb6[0] = fminf(fminf(p0.x, p1.x), p2.x) - pad;
    lb6[1] = fminf(fminf(p0.y, p1.y), p2.y) - pad;
    lb6[2] = fminf(fminf(p0.z, p1.z), p2.z) - pad;
    lb6[3] = fmaxf(fmaxf(p0.x, p1.x), p2.x) + pad;
    lb6[4] = fmaxf(fmaxf(p0.y, p1.y), p2.y) + pad;
    lb6[5] = fmaxf(fmaxf(p0.z, p1.z), p2.z) + pad;
}

}  // namespace

// ---- chip-wide front end for large meshes: bounds, Morton keys, LSD radix sort (8-bit digits) -----------
// The single-workgroup builder needs 2 ms for its first three phases at F = 79 k; these kernels do the same
// work with every CU (same keys, same stable order, hence the same tree).
constexpr int ST = 256;            // threads per sort block
constexpr int SKEYS = 8;           // keys per thread
constexpr int STILE = ST * SKEYS;  // keys per block

__global__ __launch_bounds__(256) void k_build_bounds(BuildArgs a, uint32_t* bkeys /* [6]: min keys 0..2, max keys 3..5 */) {
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < a.F; f += gridDim.x * blockDim.x) {
        for (int k = 0; k < 3; ++k) {
            int vi = clamp_index(a.faces[3 * f + k], a.V, a.status);
            for (int c = 0; c < 3; ++c) {
                float x = a.vertices[3 * (size_t)vi + c];
                lo[c] = fminf(lo[c], x);
                hi[c] = fmaxf(hi[c], x);
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1)
        for (int c = 0; c < 3; ++c) {
            lo[c] = fminf(lo[c], __shfl_down(lo[c], off));
            hi[c] = fmaxf(hi[c], __shfl_down(hi[c], off));
        }
    if ((threadIdx.x & 63) == 0)
        for (int c = 0; c < 3; ++c) {
            atomicMin(&bkeys[c], fkey(lo[c]));
            atomicMax(&bkeys[3 + c], fkey(hi[c]));
        }
}

__global__ __launch_bounds__(256) void k_build_morton(BuildArgs a, const uint32_t* bkeys) {
    float lo[3], hi[3];
    float e = 0.0f;
    for (int c = 0; c < 3; ++c) {
        lo[c] = fkey_inv(bkeys[c]);
        hi[c] = fkey_inv(bkeys[3 + c]);
        e = fmaxf(e, fmaxf(fabsf(lo[c]), fabsf(hi[c])));
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) a.box[6 * (size_t)(2 * a.F - 1)] = 1.6e-3f * e + 1e-30f;   // box padding
    const float sx = hi[0] - lo[0], sy = hi[1] - lo[1], sz = hi[2] - lo[2];
    const float ix = sx > 0 ? 1.0f / sx : 0.0f, iy = sy > 0 ? 1.0f / sy : 0.0f, iz = sz > 0 ? 1.0f / sz : 0.0f;
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= a.F) return;
    float c[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k) {
        int vi = clamp_index(a.faces[3 * f + k], a.V, a.status);
        for (int q = 0; q < 3; ++q) c[q] += a.vertices[3 * (size_t)vi + q];
    }
    float nx = (c[0] * (1.0f / 3.0f) - lo[0]) * ix;
    float ny = (c[1] * (1.0f / 3.0f) - lo[1]) * iy;
    float nz = (c[2] * (1.0f / 3.0f) - lo[2]) * iz;
    uint32_t qx = (uint32_t)fminf(fmaxf(nx * 1024.0f, 0.0f), 1023.0f);
    uint32_t qy = (uint32_t)fminf(fmaxf(ny * 1024.0f, 0.0f), 1023.0f);
    uint32_t qz = (uint32_t)fminf(fmaxf(nz * 1024.0f, 0.0f), 1023.0f);
    a.keys0[f] = (expand_bits(qx) << 2) | (expand_bits(qy) << 1) | expand_bits(qz);
    a.idx0[f] = f;
}

__global__ __launch_bounds__(ST) void k_sort_hist(const uint32_t* __restrict__ keys, int n, int shift, uint32_t* gh, int nb) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    for (int k = 0; k < SKEYS; ++k) {
        const int i = blockIdx.x * STILE + k * ST + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    gh[(size_t)threadIdx.x * nb + blockIdx.x] = h[threadIdx.x];      // digit-major: the flat scan gives stable offsets
}

__global__ __launch_bounds__(1024) void k_sort_scan(uint32_t* gh, int total) {
    __shared__ uint32_t s_w[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (total + 1023) / 1024;
    const int c0 = min(tid * per, total), c1 = min(c0 + per, total);
    uint32_t sum = 0;
    for (int i = c0; i < c1; ++i) sum += gh[i];
    uint32_t incl = sum;
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; ++w) base += s_w[w];
    uint32_t run = base + incl - sum;
    for (int i = c0; i < c1; ++i) { uint32_t v = gh[i]; gh[i] = run; run += v; }
}

__global__ __launch_bounds__(ST) void k_sort_scatter(const uint32_t* __restrict__ keys_in, const int* __restrict__ idx_in,
                                                     uint32_t* __restrict__ keys_out, int* __restrict__ idx_out, int n, int shift,
                                                     const uint32_t* __restrict__ gh, int nb) {
    __shared__ uint32_t cnt[256];          // next free global slot of every digit for this block
    __shared__ uint32_t cntw[ST / 64][256]; // per wave: keys of that digit in the current round
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    cnt[tid] = gh[(size_t)tid * nb + blockIdx.x];
    for (int w = 0; w < ST / 64; ++w) cntw[w][tid] = 0;
    __syncthreads();
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int r = 0; r < SKEYS; ++r) {          // rounds in key order: the sort stays stable
        const int i = blockIdx.x * STILE + r * ST + tid;
        const bool valid = i < n;
        const uint32_t key = valid ? keys_in[i] : 0u;
        const uint32_t d = (key >> shift) & 255u;
        unsigned long long m = __ballot(valid);
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bb = __ballot((d >> b) & 1u);
            m &= ((d >> b) & 1u) ? bb : ~bb;
        }
        const int rank = __popcll(m & lt);
        if (valid && rank == 0) cntw[wave][d] = (uint32_t)__popcll(m);
        __syncthreads();
        if (valid) {
            uint32_t off = cnt[d];
            for (int w = 0; w < wave; ++w) off += cntw[w][d];
            keys_out[off + rank] = key;
            idx_out[off + rank] = idx_in[i];
        }
        __syncthreads();
        uint32_t add = 0;
        for (int w = 0; w < ST / 64; ++w) { add += cntw[w][tid]; cntw[w][tid] = 0; }
        cnt[tid] += add;
        __syncthreads();
    }
}

// Large meshes (inner nodes beyond the LDS refit): the single workgroup stops after the sort and two
// chip-wide launches finish the job -- the tree (one thread per inner node) and the refit (one thread per
// leaf; the bottom-up walk is then as long as the deepest chain, ~35 levels, instead of F/1024 walks per
// thread: 8.6 M cycles -> ~0.1 ms at F = 79 k).
// records of a lazy build: one thread per sorted leaf (what k_build_refit does for its leaves, without the tree)
__global__ __launch_bounds__(256) void k_build_records(BuildArgs a) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.F) return;
    float lb6[6];
    leaf_records(a, a.idx0, j, 0.0f, lb6);
}

__global__ __launch_bounds__(256) void k_build_tree(BuildArgs a, int conditional) {
    if (conditional && a.need_tree && __hip_atomic_load(a.need_tree, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
    const int F = a.F, n_int = F - 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) a.parent[F > 1 ? 0 : n_int] = -1;
    if (i >= n_int) return;
    int left, right;
    int last_unused;
    karras_node(a, a.keys0, F, i, left, right, last_unused);
}

__global__ __launch_bounds__(256) void k_build_refit(BuildArgs a, int conditional) {
    if (conditional && a.need_tree && __hip_atomic_load(a.need_tree, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
    const int F = a.F, n_int = F - 1;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= F) return;
    const float pad = a.box[6 * (size_t)(2 * F - 1)];      // left there by k_build_bvh
    float lb6[6];
    leaf_records(a, a.idx0, j, pad, lb6);
    emit_node(a, n_int + j, lb6, escape_link(a, F, j), ~j);
    float* b = a.box + 6 * (size_t)(n_int + j);
    for (int c = 0; c < 6; ++c) b[c] = lb6[c];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    int node = a.parent[n_int + j];
    while (node >= 0) {
        int old = atomicAdd(&a.arrive[node], 1);
        if (old == 0) break;                   // sibling subtree not finished yet
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const float* bl = a.box + 6 * (size_t)a.child[2 * node];
        const float* br = a.box + 6 * (size_t)a.child[2 * node + 1];
        float* bo = a.box + 6 * (size_t)node;
        float nb[6];
        for (int c = 0; c < 6; ++c) {
            float x = __hip_atomic_load(bl + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            float y = __hip_atomic_load(br + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            nb[c] = c < 3 ? fminf(x, y) : fmaxf(x, y);
            __hip_atomic_store(bo + c, nb[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        emit_node(a, node, nb, escape_link(a, F, a.range[2 * node + 1]), a.child[2 * node]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        node = a.parent[node];
    }
}

// inclusive prefix sum over the 64 lanes of a wave on the DPP network (row shifts inside the rows of 16, then the two row
// broadcasts of gfx9): six VALU operations instead of six ds_bpermute round trips
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);     // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);     // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);     // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);     // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);     // row_bcast:15 -> rows 1 and 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);     // row_bcast:31 -> rows 2 and 3
    return x;
}

__global__ __launch_bounds__(1024) void k_build_bvh(BuildArgs a, int lds_words, int split) {
    extern __shared__ uint32_t s_dyn[];       // radix counters [RDIG * BT] (64 KB + skew), sort buffers; later the LDS refit
    __shared__ uint32_t s_wsum[BT / 64];
    __shared__ float s_red[6 * 16];           // per-wave bounds
    __shared__ float s_bounds[8];             // lo[3], hi[3], pad
    uint32_t* s_cnt = s_dyn;
    // flat index e -> e + e/32: the scan's stride-32 walk (thread t owns [32t, 32t+32)) then
    // lands on 33-word strides, i.e. distinct LDS banks across a wave
    auto SK = [](int e) { return e + (e >> 5); };

    const int tid = threadIdx.x;
    const int F = a.F;
    const int lane = tid & 63, wave = tid >> 6;
#ifdef NLOS_BUILD_STAMPS
    // diagnostic build only: cycles per phase -> status[4..9] (never read by the product path)
    long long t_prev = clock64();
    int t_slot = 4;
#define NLOS_STAMP() do { __syncthreads(); if (tid == 0) { long long t_now = clock64(); a.status[t_slot++] = (int)(t_now - t_prev); t_prev = t_now; } } while (0)
#else
#define NLOS_STAMP() do { } while (0)
#endif

    // ---- phase 1: scene bounds --------------------------------------------------
    // Faces tid + k * BT, k < KF, are read once: their centroid sums stay in registers for the Morton keys (the loop
    // is unrolled, so the index and vertex loads of all k are in flight together instead of one dependent chain per
    // face and phase: 36 k -> ~14 k cycles for both phases at F = 5 k).
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    float csum[KF][3];
    const bool in_regs = F <= KF * BT;
    bool bad_index = false;
    {
        int vi[KF][3];
#pragma unroll
        for (int k = 0; k < KF; ++k) {
            const int f = tid + k * BT;
            vi[k][0] = vi[k][1] = vi[k][2] = 0;
            if (in_regs && f < F) face_indices(a, f, vi[k], bad_index);
        }
#pragma unroll
        for (int k = 0; k < KF; ++k) {
            const int f = tid + k * BT;
            csum[k][0] = csum[k][1] = csum[k][2] = 0.0f;
            if (in_regs && f < F) {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const V3 x = ld3(a.vertices + 3 * (size_t)vi[k][q]);
                    lo[0] = fminf(lo[0], x.x); hi[0] = fmaxf(hi[0], x.x); csum[k][0] += x.x;
                    lo[1] = fminf(lo[1], x.y); hi[1] = fmaxf(hi[1], x.y); csum[k][1] += x.y;
                    lo[2] = fminf(lo[2], x.z); hi[2] = fmaxf(hi[2], x.z); csum[k][2] += x.z;
                }
            }
        }
    }
    if (bad_index) atomicOr(a.status, 1);
    for (int f = tid; !in_regs && f < F; f += BT) {
        for (int k = 0; k < 3; ++k) {
            int vi = clamp_index(a.faces[3 * f + k], a.V, a.status);
            for (int c = 0; c < 3; ++c) {
                float x = a.vertices[3 * (size_t)vi + c];
                lo[c] = fminf(lo[c], x);
                hi[c] = fmaxf(hi[c], x);
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
        s_bounds[6] = 1.6e-3f * e + 1e-30f;      // box padding >> fp32 rounding of hit points
    }
    __syncthreads();
    const float pad = s_bounds[6];
    NLOS_STAMP();

    // ---- phase 2: Morton keys ---------------------------------------------------
    // Keys and (16-bit) indices live in LDS behind the counters while they are sorted: a pass then costs
    // ~100 LDS operations per thread instead of a round trip through L2 (sort 119 k -> ~20 k cycles at F = 5 k).
    // Meshes too large for that (never the case below the LDS-refit limit) ping-pong through global scratch.
    const bool sort_in_lds = F <= 65535 && RCNT_WORDS + 3 * F + 4 <= lds_words;
    uint32_t* s_keyA = s_dyn + RCNT_WORDS;
    uint32_t* s_keyB = s_keyA + F;
    uint16_t* s_idxA = reinterpret_cast<uint16_t*>(s_keyB + F);
    uint16_t* s_idxB = s_idxA + F + (F & 1);
    {
        float sx = s_bounds[3] - s_bounds[0], sy = s_bounds[4] - s_bounds[1], sz = s_bounds[5] - s_bounds[2];
        float ix = sx > 0 ? 1.0f / sx : 0.0f, iy = sy > 0 ? 1.0f / sy : 0.0f, iz = sz > 0 ? 1.0f / sz : 0.0f;
        // 2^QBITS cells per axis while the registers hold the faces (always, with today's launcher: the LDS refit
        // bounds this kernel to 5.8 k faces), else the 10 bits of the generic loops
        auto morton = [&](float cx, float cy, float cz, float cells) -> uint32_t {
            float nx = (cx * (1.0f / 3.0f) - s_bounds[0]) * ix;
            float ny = (cy * (1.0f / 3.0f) - s_bounds[1]) * iy;
            float nz = (cz * (1.0f / 3.0f) - s_bounds[2]) * iz;
            uint32_t qx = (uint32_t)fminf(fmaxf(nx * cells, 0.0f), cells - 1.0f);
            uint32_t qy = (uint32_t)fminf(fmaxf(ny * cells, 0.0f), cells - 1.0f);
            uint32_t qz = (uint32_t)fminf(fmaxf(nz * cells, 0.0f), cells - 1.0f);
            return (expand_bits(qx) << 2) | (expand_bits(qy) << 1) | expand_bits(qz);
        };
#pragma unroll
        for (int k = 0; k < KF; ++k) {
            const int f = tid + k * BT;
            if (in_regs && f < F) {
                const uint32_t key = morton(csum[k][0], csum[k][1], csum[k][2], (float)(1 << QBITS));
                if (sort_in_lds) { s_keyA[f] = key; s_idxA[f] = (uint16_t)f; }
                else { a.keys0[f] = key; a.idx0[f] = f; }
            }
        }
        for (int f = tid; !in_regs && f < F; f += BT) {
            float c[3] = {0, 0, 0};
            for (int k = 0; k < 3; ++k) {
                int vi = clamp_index(a.faces[3 * f + k], a.V, a.status);
                for (int q = 0; q < 3; ++q) c[q] += a.vertices[3 * (size_t)vi + q];
            }
            const uint32_t key = morton(c[0], c[1], c[2], 1024.0f);
            if (sort_in_lds) { s_keyA[f] = key; s_idxA[f] = (uint16_t)f; }
            else { a.keys0[f] = key; a.idx0[f] = f; }
        }
    }
    __syncthreads();
    NLOS_STAMP();

    // ---- phase 3: stable LSD radix sort, one private counter column per thread ----
    // counters are laid out [digit][thread]; the exclusive scan runs over that flat order
    const int chunk = (F + BT - 1) / BT;
    const int c0 = min(tid * chunk, F), c1 = min(c0 + chunk, F);
    auto radix_sort = [&](auto* keys_in, auto* keys_out, auto* idx_in, auto* idx_out) {
        for (int pass = 0; pass < (in_regs ? RPASS : 8); ++pass) {        // 3 * QBITS-bit keys, or the 30 bits of the generic loop
            const int shift = pass * RBITS;
            for (int d = 0; d < RDIG; ++d) s_cnt[SK(d * BT + tid)] = 0;
            for (int i = c0; i < c1; ++i) s_cnt[SK(((keys_in[i] >> shift) & (RDIG - 1)) * BT + tid)] += 1;
            __syncthreads();
            // thread t owns flat entries [RDIG*t, RDIG*t + RDIG)
            uint32_t sum = 0;
            for (int q = 0; q < RDIG; ++q) sum += s_cnt[SK(tid * RDIG + q)];
            uint32_t incl = sum;
            for (int off = 1; off < 64; off <<= 1) {
                uint32_t v = __shfl_up(incl, off);
                if (lane >= off) incl += v;
            }
            if (lane == 63) s_wsum[wave] = incl;
            __syncthreads();
            uint32_t wbase = 0;
            for (int w = 0; w < wave; ++w) wbase += s_wsum[w];
            uint32_t run = wbase + incl - sum;
            for (int q = 0; q < RDIG; ++q) { uint32_t n = s_cnt[SK(tid * RDIG + q)]; s_cnt[SK(tid * RDIG + q)] = run; run += n; }
            __syncthreads();
            for (int i = c0; i < c1; ++i) {
                uint32_t k = keys_in[i];
                uint32_t dst = s_cnt[SK(((k >> shift) & (RDIG - 1)) * BT + tid)]++;
                keys_out[dst] = k;
                idx_out[dst] = idx_in[i];
            }
            __syncthreads();
            auto* tk = keys_in; keys_in = keys_out; keys_out = tk;
            auto* ti = idx_in; idx_in = idx_out; idx_out = ti;
        }
    };
    // The single-workgroup case proper (24-bit keys, keys and indices in LDS): 4 passes x 6 bits with one counter column
    // per WAVE instead of one per thread -- 64 digits x 16 waves = one counter per thread for the scan, no private
    // columns to zero, sum and rewrite.  Every wave owns a contiguous run of the input; within a 64-key round the rank of
    // a key among the round's keys of the same digit comes from six ballots (match-any), so the scatter is stable by
    // construction.  (Private columns: 6 passes x ~60 LDS operations per thread = 51 k cycles at F = 5 k; this: ~18 k.)
    auto wave_sort = [&](uint32_t* keys_in, uint32_t* keys_out, uint16_t* idx_in, uint16_t* idx_out) {
        constexpr int WBITS = 6, WDIG = 1 << WBITS, NW = BT / 64;
        static_assert(WDIG * NW == BT, "one counter per thread");
        static_assert(4 * WBITS == 3 * QBITS, "four passes cover the key");
        const int per_wave = (F + NW - 1) / NW;
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);     // scalar loop bounds
        const int w0 = min(wave_u * per_wave, F), w1 = min(w0 + per_wave, F);
        // (typed as LDS: a volatile access through a generic pointer is compiled to a system-coherent FLAT operation with a
        // full wait behind it)
        typedef volatile __attribute__((address_space(3))) uint32_t lds_vu32;
        lds_vu32* s_hist = (lds_vu32*)s_cnt;                         // [wave][digit]: a wave's 64 counters lie in 64 banks (digit-major,
                                                                     // 16 words apart, they share four: 16-way conflicts, 4 k cycles per pass)
#ifdef NLOS_BUILD_STAMPS
        long long ts_prev = clock64(), ts_acc[3] = {0, 0, 0};     // summed in registers: a read-modify-write of the status words would stall the next phase
#define NLOS_SUBSTAMP(k) do { long long t_now = clock64(); ts_acc[k] += t_now - ts_prev; ts_prev = t_now; } while (0)
#else
#define NLOS_SUBSTAMP(k) do { } while (0)
#endif
        constexpr int NR = KF;                                        // rounds of 64 keys per wave: per_wave <= KF * 64
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = pass * WBITS;
            // the wave's keys and indices, once per pass and all in flight together; they feed the histogram AND the scatter
            uint32_t k[NR], id[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int i = w0 + 64 * r + lane;
                k[r] = i < w1 ? keys_in[i] : 0u;
                id[r] = i < w1 ? (uint32_t)idx_in[i] : 0u;
            }
            // Ranks of all rounds first (ballots only): within a 64-key round, the lanes that hold my digit (match-any), my rank
            // among them, and the group's first lane, which speaks for the group in both LDS phases -- the histogram takes one
            // add of the group's size (sorted input puts whole rounds on one counter: 64 single adds serialise), and the
            // scatter one returning add that claims the group's run.
            uint32_t rank[NR], cnt[NR], dig[NR];
            int leader[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                rank[r] = 1u; cnt[r] = 0u; leader[r] = 0; dig[r] = 0u;
                if (w0 + 64 * r >= w1) continue;                    // (wave-uniform) no key in this round
                const bool valid = w0 + 64 * r + lane < w1;
                const uint32_t d = (k[r] >> shift) & (WDIG - 1);
                // per digit bit one ballot and, per half of the mask, one three-input boolean: peers &= ~(ballot ^ (my bit ? ~0 : 0))
                const unsigned long long vb = __ballot(valid);
                uint32_t plo = (uint32_t)vb, phi = (uint32_t)(vb >> 32);
#pragma unroll
                for (int b = 0; b < WBITS; ++b) {
                    const int sb = __builtin_amdgcn_sbfe((int)d, b, 1);      // 0 or -1
                    const unsigned long long m = __ballot(sb != 0);
                    plo &= ~((uint32_t)m ^ (uint32_t)sb);
                    phi &= ~((uint32_t)(m >> 32) ^ (uint32_t)sb);
                }
                dig[r] = d;
                cnt[r] = (uint32_t)(__popc(plo) + __popc(phi));
                leader[r] = plo ? __ffs((int)plo) - 1 : (phi ? 31 + __ffs((int)phi) : 0);
                rank[r] = valid ? __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u)) : 1u;    // 0: this lane speaks for its group
            }
            s_hist[tid] = 0u;
            __syncthreads();
#pragma unroll
            for (int r = 0; r < NR; ++r)
                if (rank[r] == 0u) atomicAdd(&s_cnt[wave_u * WDIG + dig[r]], cnt[r]);
            __syncthreads();
            NLOS_SUBSTAMP(0);                                       // loads + ranks + zero + histogram
            // the scan runs digit-major (all waves' counts of digit 0, then digit 1 ...): thread t holds (digit t / 16, wave t % 16)
            const int mine = (tid & (NW - 1)) * WDIG + (tid >> 4);
            const uint32_t v = s_hist[mine];
            const uint32_t incl = wave_incl_scan(v);
            if (lane == 63) s_wsum[wave] = incl;
            __syncthreads();
            uint32_t wbase = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) { const uint32_t x = s_wsum[w]; wbase += w < wave_u ? x : 0u; }   // 16 broadcast reads in flight, not a dependent chain
            s_hist[mine] = wbase + incl - v;                        // exclusive: where the keys of (digit, wave) start
            __syncthreads();
            NLOS_SUBSTAMP(1);                                       // scan
            // a wave's LDS operations execute in order, so round r + 1 sees what round r added without a wait in between;
            // then every lane fetches its leader's base: three LDS round trips per pass instead of three per round
            uint32_t at[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                at[r] = 0u;
                if (rank[r] == 0u) at[r] = atomicAdd(&s_cnt[wave_u * WDIG + dig[r]], cnt[r]);
            }
#pragma unroll
            for (int r = 0; r < NR; ++r) at[r] = __shfl(at[r], leader[r]);
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                if (w0 + 64 * r + lane < w1) {
                    keys_out[at[r] + rank[r]] = k[r];
                    idx_out[at[r] + rank[r]] = (uint16_t)id[r];
                }
            }
            __syncthreads();
            NLOS_SUBSTAMP(2);                                       // ranks + scatter
            uint32_t* tk = keys_in; keys_in = keys_out; keys_out = tk;
            uint16_t* ti = idx_in; idx_in = idx_out; idx_out = ti;
        }
#ifdef NLOS_BUILD_STAMPS
        if (tid == 0) for (int k = 0; k < 3; ++k) a.status[10 + k] = (int)ts_acc[k];
#endif
    };
    if (sort_in_lds) {
#ifdef NLOS_DIAG_THREAD_SORT       // diagnostic builds only: the private-column sort for every size
        radix_sort(s_keyA, s_keyB, s_idxA, s_idxB);
#else
        if (in_regs) wave_sort(s_keyA, s_keyB, s_idxA, s_idxB);    // an even number of passes: the result is back in A
        else radix_sort(s_keyA, s_keyB, s_idxA, s_idxB);
#endif
        for (int f = tid; f < F; f += BT) { a.keys0[f] = s_keyA[f]; a.idx0[f] = (int)s_idxA[f]; }
        __syncthreads();
    } else {
        radix_sort(a.keys0, a.keys1, a.idx0, a.idx1);
    }
    uint32_t* keys_in = a.keys0;              // the sorted keys / order (even number of passes)
    int* idx_in = a.idx0;
    const uint32_t* keys = keys_in;           // == a.keys0 / a.idx0 (even number of passes)
    const int* order = idx_in;
    NLOS_STAMP();
    if (split) {
        if (tid == 0) {
            a.box[6 * (size_t)(2 * F - 1)] = pad;      // for k_build_refit
            if (a.lazy) {
                // The root box -- all the perspective grid reads of the tree (source_frame()).  Bit for bit what the refit
                // arrives at: the union of the leaves' padded boxes, and x -> fl(x -/+ pad) is monotonic.
                a.nodes[0] = make_float4(s_bounds[0] - pad, s_bounds[1] - pad, s_bounds[2] - pad, s_bounds[3] + pad);
                a.nodes[1] = make_float4(s_bounds[4] + pad, s_bounds[5] + pad, __int_as_float(-1), __int_as_float(F > 1 ? 1 : ~0));
                if (a.need_tree) *a.need_tree = 0;
                if (a.host_status)
                    __hip_atomic_store(a.host_status, __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        return;
    }

    // ---- phase 4: Karras radix tree ------------------------------------------------
    const int n_int = F - 1;
    // Refit in LDS when the inner nodes fit (28 B each: parent word with an "arrived" flag in bit 31 +
    // six order-preserving box keys).  The bottom-up walk is a chain of up to ~30 dependent levels
    // towards the root; with global atomics + fences every level costs microseconds (0.2 ms in
    // total, 70 % of the whole build), in LDS a few hundred cycles.
    const bool lds_refit = n_int > 0 && 7 * n_int <= lds_words;
    uint32_t* s_par = s_dyn;                     // [n_int]
    uint32_t* s_box = s_dyn + n_int;             // [6 * n_int]: 0..2 min keys, 3..5 max keys
    // The sorted keys are still in LDS behind the counters (inside the future s_box area, clear of s_par): the
    // binary searches of the tree construction read them there; the boxes are initialised afterwards.
    const uint32_t* tree_keys = sort_in_lds ? s_keyA : keys;
    if (lds_refit && tid == 0) s_par[0] = 0x7FFFFFFFu;  // root: no parent
    // staged = the common case (LDS refit, faces in registers): nodes tid + k * BT keep their left child and the end of
    // their leaf range in registers for the node emission below
    const bool staged = lds_refit && in_regs;
    int nd_left[KF], nd_last[KF];
#pragma unroll
    for (int k = 0; k < KF; ++k) {
        const int i = tid + k * BT;
        nd_left[k] = nd_last[k] = 0;
        if (staged && i < n_int) {
            int left, right, last;
            karras_node<true>(a, tree_keys, F, i, left, right, last);
            if (left < n_int) s_par[left] = (uint32_t)i;
            if (right < n_int) s_par[right] = (uint32_t)i;
            nd_left[k] = left;
            nd_last[k] = last;
        }
    }
    for (int i = tid; !staged && i < n_int; i += BT) {
        int left, right, last_unused;
        karras_node(a, tree_keys, F, i, left, right, last_unused);
        if (lds_refit) {
            if (left < n_int) s_par[left] = (uint32_t)i;
            if (right < n_int) s_par[right] = (uint32_t)i;
        }
    }
    if (lds_refit) {
        __syncthreads();                                 // every search is done with the keys
        for (int i = tid; i < n_int; i += BT)
            for (int c = 0; c < 3; ++c) { s_box[6 * i + c] = 0xFFFFFFFFu; s_box[6 * i + 3 + c] = 0u; }
    }
    if (tid == 0) a.parent[F > 1 ? 0 : n_int] = -1;
    __syncthreads();
    NLOS_STAMP();

    auto escape_of = [&](int last) -> int { return escape_link(a, F, last); };

    // ---- phase 5: leaf records + bottom-up refit + node emission -------------------------
    if (staged) {
        // Leaves tid + k * BT: every round of dependent loads (order -> indices -> vertices; parents and escape links
        // beside them) is issued for all k before anything is stored -- a store between two rounds would be waited for
        // with the loads behind it (vmcnt is in order), and the generic loop below walked one dependent chain per leaf:
        // 108 k -> ~60 k cycles for this phase at F = 5 k.
        constexpr int KH = 3;                    // leaves per batch: three dependent rounds of loads per batch, 45 registers
        bool bad = false;
#pragma unroll 1
        for (int h = 0; h < KF; h += KH) {
            int fo[KH], par[KH], esc[KH], vi[KH][3];
#pragma unroll
            for (int k = 0; k < KH; ++k) {
                const int j = tid + (h + k) * BT;
                fo[k] = 0; par[k] = -1; esc[k] = -1;
                if (j < F) {
                    fo[k] = order[j];
                    par[k] = F > 1 ? a.parent[n_int + j] : -1;
                    esc[k] = escape_of(j);
                }
            }
#pragma unroll
            for (int k = 0; k < KH; ++k) {
                vi[k][0] = vi[k][1] = vi[k][2] = 0;
                if (tid + (h + k) * BT < F) face_indices(a, fo[k], vi[k], bad);
            }
            V3 P[KH][3];
#pragma unroll
            for (int k = 0; k < KH; ++k)
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    P[k][q] = (tid + (h + k) * BT < F) ? ld3(a.vertices + 3 * (size_t)vi[k][q]) : mk(0.0f, 0.0f, 0.0f);
#pragma unroll
            for (int k = 0; k < KH; ++k) {
                const int j = tid + (h + k) * BT;
                if (j >= F) continue;
                float lb6[6];
                leaf_records_from(a, j, fo[k], vi[k], P[k][0], P[k][1], P[k][2], pad, lb6);
                emit_node(a, n_int + j, lb6, esc[k], ~j);
                // merge into the parent's box with LDS min/max atomics, then raise the parent's flag; the
                // second arriver finds the box complete, takes it as its own and continues one level up
                uint32_t key[6];
                for (int c = 0; c < 6; ++c) key[c] = fkey(lb6[c]);
                uint32_t node = par[k] >= 0 ? (uint32_t)par[k] : 0x7FFFFFFFu;
                while (node != 0x7FFFFFFFu) {
                    for (int c = 0; c < 3; ++c) atomicMin(&s_box[6 * node + c], key[c]);
                    for (int c = 3; c < 6; ++c) atomicMax(&s_box[6 * node + c], key[c]);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    const uint32_t old = atomicOr(&s_par[node], 0x80000000u);
                    if (!(old & 0x80000000u)) break;          // sibling subtree not finished yet
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    for (int c = 0; c < 6; ++c)
                        key[c] = __hip_atomic_load(&s_box[6 * node + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    node = old & 0x7FFFFFFFu;
                }
            }
        }
        if (bad) atomicOr(a.status, 1);
        __syncthreads();
        // inner nodes: the escape links first (one more load each), then the stores
        int nesc[KF];
#pragma unroll
        for (int k = 0; k < KF; ++k) nesc[k] = (tid + k * BT < n_int) ? escape_of(nd_last[k]) : -1;
#pragma unroll
        for (int k = 0; k < KF; ++k) {
            const int i = tid + k * BT;
            if (i >= n_int) continue;
            float nb[6];
            for (int c = 0; c < 6; ++c) nb[c] = fkey_inv(s_box[6 * i + c]);
            emit_node(a, i, nb, nesc[k], nd_left[k]);
        }
        // the bad-index word goes to the host by a store of its own (every atomicOr on it happened before the barrier
        // above): no copy operation behind the build in the stream
        if (tid == 0 && a.host_status)
            __hip_atomic_store(a.host_status, __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        NLOS_STAMP();
        return;
    }
    for (int j = tid; j < F; j += BT) {
        float lb6[6];
        leaf_records(a, order, j, pad, lb6);
        emit_node(a, n_int + j, lb6, escape_of(j), ~j);
        if (lds_refit) {
            // merge into the parent's box with LDS min/max atomics, then raise the parent's flag; the
            // second arriver finds the box complete, takes it as its own and continues one level up
            uint32_t key[6];
            for (int c = 0; c < 6; ++c) key[c] = fkey(lb6[c]);
            uint32_t node = F > 1 ? (uint32_t)a.parent[n_int + j] : 0x7FFFFFFFu;
            while (node != 0x7FFFFFFFu) {
                for (int c = 0; c < 3; ++c) atomicMin(&s_box[6 * node + c], key[c]);
                for (int c = 3; c < 6; ++c) atomicMax(&s_box[6 * node + c], key[c]);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                const uint32_t old = atomicOr(&s_par[node], 0x80000000u);
                if (!(old & 0x80000000u)) break;          // sibling subtree not finished yet
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                for (int c = 0; c < 6; ++c)
                    key[c] = __hip_atomic_load(&s_box[6 * node + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                node = old & 0x7FFFFFFFu;
            }
            continue;
        }
        float* b = a.box + 6 * (size_t)(n_int + j);
        for (int c = 0; c < 6; ++c) b[c] = lb6[c];
        // One workgroup = one CU: a workgroup-scope release (s_waitcnt vmcnt(0): the write-through
        // stores have reached the L2) orders the box before the arrival count; the second arriver
        // reads the sibling's box with L1-bypassing loads.  No device-wide cache flushes needed.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        int node = a.parent[n_int + j];
        while (node >= 0) {
            int old = atomicAdd(&a.arrive[node], 1);
            if (old == 0) break;                   // sibling subtree not finished yet
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const float* bl = a.box + 6 * (size_t)a.child[2 * node];
            const float* br = a.box + 6 * (size_t)a.child[2 * node + 1];
            float* bo = a.box + 6 * (size_t)node;
            float nb[6];
            for (int c = 0; c < 6; ++c) {
                float x = __hip_atomic_load(bl + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                float y = __hip_atomic_load(br + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                nb[c] = c < 3 ? fminf(x, y) : fmaxf(x, y);
                bo[c] = nb[c];
            }
            emit_node(a, node, nb, escape_of(a.range[2 * node + 1]), a.child[2 * node]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            node = a.parent[node];
        }
    }
    if (lds_refit) {
        __syncthreads();
        for (int i = tid; i < n_int; i += BT) {
            float nb[6];
            for (int c = 0; c < 6; ++c) nb[c] = fkey_inv(s_box[6 * i + c]);
            emit_node(a, i, nb, escape_of(a.range[2 * i + 1]), a.child[2 * i]);
        }
    }
    __syncthreads();
    if (tid == 0 && a.host_status)
        __hip_atomic_store(a.host_status, __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    NLOS_STAMP();
}

bool launch_build_bvh(const BuildArgs& a, hipStream_t stream) {
    // dynamic LDS: the radix counters (66 KB) + sort buffers (12 B per face), or 28 B per inner node for the LDS refit if that is more
    // and still fits beside the static arrays (160 KB per CU)
    // (radix counters + keys and 16-bit indices of the in-LDS sort)
    size_t lds = ((size_t)RCNT_WORDS + 3 * (size_t)a.F + 4) * sizeof(uint32_t);
    const size_t refit = 7 * sizeof(uint32_t) * (size_t)(a.F > 1 ? a.F - 1 : 0);
    const size_t lds_max = 160 * 1024 - 1024;
    if (lds > lds_max) lds = (size_t)RCNT_WORDS * sizeof(uint32_t);      // the sort then goes through global scratch
    if (refit > lds && refit <= lds_max) lds = refit;
    const int split = (refit > lds_max && a.F > 1) ? 1 : 0;      // beyond the LDS refit: chip-wide launches
    if (a.lazy && a.F > 1 && ((size_t)RCNT_WORDS + 3 * (size_t)a.F + 4) * sizeof(uint32_t) <= lds_max && a.F <= KF * BT) {
        // lazy: the single-workgroup front end (bounds, keys, in-LDS sort, root box), then the records chip-wide
        const size_t lds_sort = ((size_t)RCNT_WORDS + 3 * (size_t)a.F + 4) * sizeof(uint32_t);
        note_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_build_bvh), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds_sort), "hipFuncSetAttribute(dynamic LDS)");
        hipLaunchKernelGGL(k_build_bvh, dim3(1), dim3(BT), lds_sort, stream, a, (int)(lds_sort / sizeof(uint32_t)), 1);
        hipLaunchKernelGGL(k_build_records, dim3((a.F + 255) / 256), dim3(256), 0, stream, a);
        return a.host_status != nullptr;
    }
    if (!split) {
        note_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_build_bvh), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds), "hipFuncSetAttribute(dynamic LDS)");
        hipLaunchKernelGGL(k_build_bvh, dim3(1), dim3(BT), lds, stream, a, (int)(lds / sizeof(uint32_t)), 0);
        return a.host_status != nullptr;
    }
    // bounds -> Morton keys -> 4 x (histogram, scan, scatter) -> tree -> refit
    uint32_t* bkeys = reinterpret_cast<uint32_t*>(a.status) + 16;          // 6 order-preserving bound keys
    (void)hipMemsetAsync(bkeys, 0xFF, 3 * sizeof(uint32_t), stream);
    (void)hipMemsetAsync(bkeys + 3, 0x00, 3 * sizeof(uint32_t), stream);
    const int fb = (a.F + 255) / 256;
    hipLaunchKernelGGL(k_build_bounds, dim3(std::min(fb, 1024)), dim3(256), 0, stream, a, bkeys);
    hipLaunchKernelGGL(k_build_morton, dim3(fb), dim3(256), 0, stream, a, bkeys);
    const int nb = (a.F + STILE - 1) / STILE;
    uint32_t* gh = reinterpret_cast<uint32_t*>(a.child);                   // 256 * nb words, free until k_build_tree
    uint32_t *kin = a.keys0, *kout = a.keys1;
    int *iin = a.idx0, *iout = a.idx1;
    for (int pass = 0; pass < 4; ++pass) {                                 // 32 key bits, the result lands in keys0 / idx0
        hipLaunchKernelGGL(k_sort_hist, dim3(nb), dim3(ST), 0, stream, kin, a.F, 8 * pass, gh, nb);
        hipLaunchKernelGGL(k_sort_scan, dim3(1), dim3(1024), 0, stream, gh, 256 * nb);
        hipLaunchKernelGGL(k_sort_scatter, dim3(nb), dim3(ST), 0, stream, kin, iin, kout, iout, a.F, 8 * pass, gh, nb);
        std::swap(kin, kout);
        std::swap(iin, iout);
    }
    hipLaunchKernelGGL(k_build_tree, dim3((a.F + 255) / 256), dim3(256), 0, stream, a, 0);
    hipLaunchKernelGGL(k_build_refit, dim3((a.F + 255) / 256), dim3(256), 0, stream, a, 0);
    return false;
}

void launch_build_tree(const BuildArgs& a, bool conditional, hipStream_t stream) {
    if (a.F <= 1) return;
    hipLaunchKernelGGL(k_build_tree, dim3((a.F + 255) / 256), dim3(256), 0, stream, a, conditional ? 1 : 0);
    hipLaunchKernelGGL(k_build_refit, dim3((a.F + 255) / 256), dim3(256), 0, stream, a, conditional ? 1 : 0);
}

}  // namespace nlos
