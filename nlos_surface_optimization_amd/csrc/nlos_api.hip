// nlos_api.hip -- C ABI of libnlos_hip.so (see include/nlos_hip.h).
//
// Host-side driver of the render: the counterpart of the reference's
// streamed_render_transient / streamed_render_gradient drivers
// (smoothed_transient/stratifiedStreamedTransientRenderer.cpp:81-153,
//  smoothed_transient/stratifiedStreamedGradientRenderer.cpp:471-578): build the
// acceleration structure, pass 1, residual, pass 2 -- all enqueued on one HIP
// stream with no host synchronisation in between.
#include "../../include/nlos_hip.h"
#include "nlos_kernels.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e__ = (expr);                                                            \
        if (e__ != hipSuccess)                                                              \
            return fail(NLOS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));  \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return NLOS_OK;
        if (p) { hipError_t e = hipFree(p); (void)e; p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(NLOS_ERR_HIP, std::string("hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e));
        }
        cap = want;
        return NLOS_OK;
    }
    void release() {
        if (p) { hipError_t e = hipFree(p); (void)e; }
        p = nullptr; cap = 0;
    }
    // For scratch that is an optimisation, never a reason to fail (the geometry cache): exactly `bytes`, allocated BEFORE the
    // old buffer is given up -- a growth that does not succeed leaves what worked in place -- and without touching the
    // last-error string.  false: no buffer of that size.
    bool try_ensure(size_t bytes) {
        if (bytes <= cap) return true;
        void* q = nullptr;
        if (hipMalloc(&q, bytes) != hipSuccess) { (void)hipGetLastError(); return false; }
        if (p) { hipError_t e = hipFree(p); (void)e; }
        p = q; cap = bytes;
        return true;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) {
            if (hipSetDevice(dev) == hipSuccess) changed = true;
        }
    }
    ~DeviceGuard() {
        if (changed) { hipError_t e = hipSetDevice(prev); (void)e; }
    }
};

}  // namespace

struct nlos_ctx {
    int device = 0;
    // BVH scratch + outputs
    DevBuf keys0, keys1, idx0, idx1, child, range, parent, arrive, box, status;
    DevBuf nodes, tris, facerec, face_id, tri_zmin;
    int built_F = -1, built_V = -1;
    // lazy scene build: the grid back-end reads the sorted records and the root box only, so a render
    // that will use it builds just those; tree_complete == false means nodes[] beyond the root may not exist.  The
    // launchers complete the tree when a back-end needs it (lazy_args: what that takes).
    bool tree_complete = false;
    nlos::BuildArgs lazy_args;
    DevBuf lazy_flag;
    // visibility cache as item masks (nlos_kernels.h, ForwardArgs::vis_items): what pass 1 of the single-workgroup grid
    // records for confocal renders with spt <= 32; `vis_is_items` says which layout the cache of `vis_gen` is in
    DevBuf vis_items;
    bool vis_is_items = false;
    int vis_items_stride = 0;
    // stale-cache protection: every scene build / recorded pass 1 takes the next value of one counter
    int64_t gen_counter = 0, mesh_gen = 0, vis_gen = 0;
    // what the last render did (nlos_ctx_last_path)
    nlos_path_info path;
    int path_retry_workgroups = 0;
    int path_items_sources = 0;      // > 0: the last pass 1 recorded item masks for this many sources (their headers carry its ray counts)
    // deferred device-side status (bad face index seen by the scene build): copied to pinned host memory behind
    // the build, looked at -- without synchronising -- by the next nlos_render, or by nlos_ctx_check
    int* h_status = nullptr;
    hipEvent_t status_ev = nullptr;
    bool status_pending = false;
    bool status_clear = true;        // status[0] is sticky across builds; cleared once its error has been reported
    // render scratch
    DevBuf vis, diff, fine, taps, rows_tmp, grad_tmp, live;
    DevBuf reg_normal, reg_area, reg_owner;
    DevBuf vis2, tile_list, tile_count, cov;
    DevBuf geo;                      // pass 1 -> pass 2 geometry cache (h, v, w per ray of the live lists)
    int64_t geo_gen = 0;             // the visibility generation the geometry cache was recorded with (0: none)
    int geo_stride = 0, geo_sources = 0;
    // pass 1's source order (ForwardArgs::perm): recomputed when the caller's origin array or its length changes -- an array
    // mutated in place keeps a stale order, which costs time at worst (any permutation renders the same rows)
    DevBuf src_perm;
    const float* perm_origin = nullptr;
    int perm_L = -1;
    DevBuf prod_rec, prod_pairs;     // the product of row N: per-wall-point records; enumerated pairs of the fallback
    int tap_refine = -1, tap_sigma = -1; float tap_res = -1.0f; int tap_kind = -1;
    // host-pointer path staging
    DevBuf io[16];
    // host-pointer drop-ins (round 6): a render stream and a copy stream of the context's own (non-blocking: a numpy call no
    // longer stalls the legacy default stream or anybody's torch stream), and the events that order them --
    // `host_p1`: pass 1 has finished (the rows are final: their download may start, behind pass 2);
    // `host_up`: data / weight have arrived (pass 2 waits for it, pass 1 does not)
    hipStream_t host_render = nullptr, host_copy = nullptr;
    hipEvent_t host_p1 = nullptr, host_up = nullptr;
    // what the visibility cache currently describes
    struct VisKey { int L = -1, F = -1, V = -1, spt = -1; long long off = -1; int stride = 1; uint64_t seed = 0; float lb = 0, ub = 0;
                    int feat = -1; int64_t mesh_gen = -1; } vis_key;
    // timing
    // ring of event sets: no host sync inside a timed loop, read back after the final sync
    static constexpr int kRing = 256;
    bool timing = false;
    std::vector<hipEvent_t> ring;    // kRing * 5 events, created on enable
    int ring_head = 0;               // next slot to record into
    int ring_count = 0;              // slots recorded since the last reset (saturates at kRing)
    hipEvent_t* ev = nullptr;        // slot being recorded by the current render
    bool ev_valid = false;
};

namespace {

std::mutex g_mu;
std::vector<nlos_ctx*> g_default_ctx;     // per device, for the host-pointer drop-ins
uint64_t g_default_seed = 0;
int g_default_device = 0;
int g_reg_overwrite = 0;

int get_default_ctx(nlos_ctx** out) {
    std::lock_guard<std::mutex> lk(g_mu);
    int dev = g_default_device;
    if ((int)g_default_ctx.size() <= dev) g_default_ctx.resize(dev + 1, nullptr);
    if (!g_default_ctx[dev]) {
        int rc = nlos_ctx_create(dev, &g_default_ctx[dev]);
        if (rc) return rc;
    }
    *out = g_default_ctx[dev];
    return NLOS_OK;
}

// Gaussian taps of the gradient pass, computed exactly as the reference does
// (smoothed_transient/transient_and_gradient.cpp:537-547, :973-974): K =
// 4*refine*sigma_bin+1, sigma = res*sigma_bin/2.355 (float*int, then double),
// delta_i evaluated in float.  Layout: [w(K) | delta(K) | g(K) | P0(K+1) | P1(K+1) | PW(K+1)] with the
// prefix sums the grouped-tap gradient kernels use (P0, P1 over the float-rounded weights of the vertex
// gradient; PW over the double weights of the scalar gradients).
void host_prefix(std::vector<double>& t, int K) {
    double p0 = 0.0, p1 = 0.0, pw = 0.0;
    t[3 * (size_t)K] = 0.0;
    t[4 * (size_t)K + 1] = 0.0;
    t[5 * (size_t)K + 2] = 0.0;
    for (int i = 0; i < K; ++i) {
        double wf = (double)(float)t[i];
        p0 += wf;
        p1 += t[2 * (size_t)K + i] * wf;
        pw += t[i];
        t[3 * (size_t)K + 1 + i] = p0;
        t[4 * (size_t)K + 2 + i] = p1;
        t[5 * (size_t)K + 3 + i] = pw;
    }
}

void host_taps(int refine, int sigma_bin, float res, std::vector<double>& t) {
    const int K = 4 * refine * sigma_bin + 1;
    t.assign(6 * (size_t)K + 3, 0.0);
    const double sigma = res * sigma_bin / 2.355;
    const double sigma_square = sigma * sigma;
    const double normalization = 1 / sigma / std::sqrt(2 * M_PI) * res / refine;
    for (int i = 0; i < K; ++i) {
        double tt = (-2 * refine * sigma_bin + i) * res / refine / sigma;
        t[i] = std::exp(-(tt * tt) / 2) * normalization;
        float d = (-2 * refine * sigma_bin + i) * res / refine;
        double dl = (double)d;
        t[K + i] = dl;
        t[2 * K + i] = (double)(float)(dl / sigma_square * 2);
    }
    host_prefix(t, K);
    // the per-bin weights of the grouped-tap loop, by the tap ic at which the first bin boundary falls (render_common.h,
    // TapTables::wt): bins k = 0 .. nb - 1 hold the taps [min(K, ic + (k - 1) refine), min(K, ic + k refine)), the last one up to K
    const int nb = 4 * sigma_bin + 1;
    const size_t base = t.size();
    t.resize(base + 2 * (size_t)(refine + 1) * nb, 0.0);
    const double* P0 = t.data() + 3 * (size_t)K;
    const double* P1 = t.data() + 4 * (size_t)K + 1;
    for (int ic = 0; ic <= refine; ++ic) {
        int prev = 0;
        for (int k = 0; k < nb; ++k) {
            const int ie = k < nb - 1 ? std::max(prev, std::min(K, ic + k * refine)) : K;
            t[base + 2 * ((size_t)ic * nb + k)] = P0[ie] - P0[prev];
            t[base + 2 * ((size_t)ic * nb + k) + 1] = P1[ie] - P1[prev];
            prev = ie;
        }
    }
}

// single unit tap (v1 gradient: delta 0, weight 1)
void host_taps_unit(std::vector<double>& t) {
    t.assign(9, 0.0);
    t[0] = 1.0;
    host_prefix(t, 1);
}

int ensure_taps(nlos_ctx* c, int kind, int refine, int sigma_bin, float res, hipStream_t st, int* K_out) {
    const int K = kind == 1 ? 1 : 4 * refine * sigma_bin + 1;
    *K_out = K;
    if (c->tap_kind == kind && c->tap_refine == refine && c->tap_sigma == sigma_bin && c->tap_res == res) return NLOS_OK;
    std::vector<double> t;
    if (kind == 1) host_taps_unit(t); else host_taps(refine, sigma_bin, res, t);
    int rc = c->taps.ensure(t.size() * sizeof(double));
    if (rc) return rc;
    // synchronous upload (tiny; happens only when the tap parameters change)
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy(c->taps.p, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice));
    c->tap_kind = kind; c->tap_refine = refine; c->tap_sigma = sigma_bin; c->tap_res = res;
    return NLOS_OK;
}

// Deferred status of the scene builds (face index out of range).  `wait` synchronises with the copy.
int check_status(nlos_ctx* c, bool wait) {
    if (!c->status_pending || !c->status_ev || !c->h_status) return NLOS_OK;
    if (wait) {
        HIP_TRY(hipEventSynchronize(c->status_ev));
    } else {
        hipError_t q = hipEventQuery(c->status_ev);
        if (q != hipSuccess) { (void)hipGetLastError(); return NLOS_OK; }     // not there yet: look again next time
    }
    c->status_pending = false;
    if (c->h_status[0] & 1) {
        c->h_status[0] = 0;
        c->status_clear = true;
        return fail(NLOS_ERR_ARG, "face index out of range [0, numVertices) in an earlier render on this context "
                                  "(the out-of-range indices were read as vertex 0)");
    }
    return NLOS_OK;
}

int ensure_bvh(nlos_ctx* c, const float* V, int nV, const int32_t* F, int nF, bool reuse, int64_t reuse_gen, hipStream_t st,
               bool lazy = false) {
    if (reuse) {
        // the caller vouches that vertices/faces are those of generation reuse_gen; any build since then
        // (another mesh, another vertex position) makes that claim stale
        if (c->built_F == nF && c->built_V == nV && reuse_gen != 0 && reuse_gen == c->mesh_gen) {
            if (!c->tree_complete) {
                // a lazily built scene: completing the tree reads the mesh again -- through THIS call's pointers
                c->lazy_args.vertices = V; c->lazy_args.faces = F;
                if (!lazy) {
                    nlos::launch_build_tree(c->lazy_args, false, st);
                    HIP_TRY(hipGetLastError());
                    c->tree_complete = true;
                }
            }
            return NLOS_OK;
        }
        return fail(NLOS_ERR_ARG, "nlos_render: reuse_bvh requested but the context no longer holds the tree of that "
                                  "mesh generation (pass the value nlos_ctx_mesh_generation() returned after the render to reuse)");
    }
    const size_t n_nodes = 2 * (size_t)nF - 1;
    int rc = 0;
    rc |= c->keys0.ensure(sizeof(uint32_t) * nF);
    rc |= c->keys1.ensure(sizeof(uint32_t) * nF);
    rc |= c->idx0.ensure(sizeof(int) * nF);
    rc |= c->idx1.ensure(sizeof(int) * nF);
    rc |= c->child.ensure(sizeof(int) * 2 * (size_t)nF);
    rc |= c->range.ensure(sizeof(int) * 2 * (size_t)nF);
    rc |= c->parent.ensure(sizeof(int) * n_nodes);
    rc |= c->arrive.ensure(sizeof(int) * (size_t)nF);
    rc |= c->box.ensure(sizeof(float) * (6 * n_nodes + 8));      // + the box padding handed to k_build_refit
    rc |= c->status.ensure(sizeof(int) * 128);      // (64 ints of status + scratch; the diagnostic builds' 40 counters)
    rc |= c->nodes.ensure(sizeof(float4) * 2 * n_nodes);
    rc |= c->tris.ensure(sizeof(float4) * 4 * (size_t)nF);
    rc |= c->facerec.ensure(sizeof(float4) * 4 * (size_t)nF);
    rc |= c->face_id.ensure(sizeof(int) * (size_t)nF);
    rc |= c->tri_zmin.ensure(sizeof(float) * (size_t)nF);
    rc |= c->lazy_flag.ensure(sizeof(int) * 4);
    if (rc) return NLOS_ERR_HIP;
    // status[0] (bad face index) is sticky until the host has reported it; the rest is scratch its users initialise
    if (c->status_clear) { HIP_TRY(hipMemsetAsync(c->status.p, 0, sizeof(int), st)); c->status_clear = false; }
#ifdef NLOS_BUILD_STAMPS
    HIP_TRY(hipMemsetAsync(c->status.as<int>() + 1, 0, sizeof(int) * 63, st));      // (the chip-wide front end initialises its own words)
#endif
    // the build's status word follows it to pinned host memory; nobody waits for it here
    if (!c->h_status) {
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_status), 64, hipHostMallocMapped));
        c->h_status[0] = 0;
        HIP_TRY(hipEventCreateWithFlags(&c->status_ev, hipEventDisableTiming));
    }
    int* h_status_dev = nullptr;
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&h_status_dev), c->h_status, 0) != hipSuccess) { h_status_dev = nullptr; (void)hipGetLastError(); }
    nlos::BuildArgs b;
    b.vertices = V; b.faces = F; b.V = nV; b.F = nF;
    b.host_status = h_status_dev;
    b.keys0 = c->keys0.as<uint32_t>(); b.keys1 = c->keys1.as<uint32_t>();
    b.idx0 = c->idx0.as<int>(); b.idx1 = c->idx1.as<int>();
    b.child = c->child.as<int>(); b.range = c->range.as<int>(); b.parent = c->parent.as<int>();
    b.arrive = c->arrive.as<int>(); b.box = c->box.as<float>(); b.status = c->status.as<int>();
    b.nodes = c->nodes.as<float4>(); b.tris = c->tris.as<float4>(); b.facerec = c->facerec.as<float4>();
    b.face_id = c->face_id.as<int>();
    b.tri_zmin = c->tri_zmin.as<float>();
    // lazy: only where the single-workgroup front end applies (the chip-wide builder of larger meshes always finishes)
    b.lazy = (lazy && nF >= 64 && nF <= 6144) ? 1 : 0;
    b.need_tree = c->lazy_flag.as<int>();
    const bool status_stored = nlos::launch_build_bvh(b, st);
    c->tree_complete = !b.lazy;
    c->lazy_args = b;
    HIP_TRY(hipGetLastError());
#ifdef NLOS_BUILD_STAMPS
    {   // diagnostic builds only: per-phase cycles of the build kernel
        int h[16];
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipMemcpy(h, c->status.p, sizeof(h), hipMemcpyDeviceToHost));
        std::fprintf(stderr, "[build stamps] bounds %d morton %d sort %d (histogram %d scan %d scatter %d) karras %d refit %d (cycles)\n", h[4], h[5], h[6], h[10], h[11], h[12], h[7], h[8]);
    }
#endif
    c->built_F = nF; c->built_V = nV;
    c->mesh_gen = ++c->gen_counter;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { cs = hipStreamCaptureStatusNone; (void)hipGetLastError(); }
    if (cs == hipStreamCaptureStatusNone) {     // (a captured render replays without the host looking on)
        if (!status_stored) HIP_TRY(hipMemcpyAsync(c->h_status, c->status.p, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipEventRecord(c->status_ev, st));
        c->status_pending = true;
    }
    return NLOS_OK;
}

nlos::SceneView scene_view(const nlos_ctx* c, int nF, int nV, const float* vn, const float* alb) {
    nlos::SceneView s;
    s.nodes = c->nodes.as<float4>(); s.tris = c->tris.as<float4>(); s.facerec = c->facerec.as<float4>();
    s.face_id = c->face_id.as<int>();
    s.tri_zmin = c->tri_zmin.as<float>();
    s.n_nodes = 2 * nF - 1; s.F = nF; s.V = nV;
    s.vertex_normal = vn; s.albedo = alb;
    return s;
}

void mark(nlos_ctx* c, int i, hipStream_t st) {
    if (!c->timing || c->ring.empty()) return;
    if (i == 0) c->ev = &c->ring[(size_t)c->ring_head * 5];
    hipError_t e = hipEventRecord(c->ev[i], st);
    (void)e;
    if (i == 4) {
        c->ring_head = (c->ring_head + 1) % nlos_ctx::kRing;
        if (c->ring_count < nlos_ctx::kRing) ++c->ring_count;
    }
}

}  // namespace

extern "C" {

const char* nlos_last_error(void) { return g_err.c_str(); }

int nlos_version(void) { return 100; }

int nlos_env_report(char* buf, int cap) {
    const nlos::EnvSwitches& e = nlos::env_switches();
    char tmp[1024];
    const int n = std::snprintf(tmp, sizeof(tmp),
        "NLOS_TILE_THRESHOLD=%d (6200)\nNLOS_LAZY_TREE=%d (1)\nNLOS_FUSE_RESIDUAL=%d (1)\nNLOS_TILE_TRIS=%d (3000)\n"
        "NLOS_TILE_SCRATCH_MAX=%llu (34359738368)\nNLOS_VIS_ITEMS=%d (1)\nNLOS_GEO_CACHE=%d (1)\nNLOS_GEO_CACHE_MAX_GB=%g (-1)\n"
        "NLOS_ROW_LDS_MAX=%zu (10240)\nNLOS_GRAD_WIDE=%d (1)\nNLOS_GRAD_MIN_SOURCES=%d (1)\nNLOS_FWD_ORDER=%d (1)\nNLOS_GEO_MAX_SPT=%d (8)\n",
        e.tile_threshold, (int)e.lazy_tree, (int)e.fuse_residual, e.tile_tris, e.tile_scratch_max, (int)e.vis_items, (int)e.geo_cache,
        e.geo_cache_max_gb, e.row_lds_max, e.grad_wide, e.grad_min_sources, (int)e.fwd_order, e.geo_max_spt);
    if (buf && cap > 0) { std::snprintf(buf, (size_t)cap, "%s", tmp); }
    return n;
}

int nlos_sizeof_render_args(void) { return (int)sizeof(nlos_render_args); }

int nlos_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int nlos_num_bins(float lb, float ub, float res) {
    // smoothed_transient/stratifiedStreamedGradientRenderer.cpp:514-515 (float32 ceil)
    return (int)std::ceil((ub - lb) / res);
}

void nlos_set_default_seed(uint64_t seed) { g_default_seed = seed; }
void nlos_set_default_device(int device) { g_default_device = device; }
void nlos_set_regulariser_overwrite(int overwrite) { g_reg_overwrite = overwrite ? 1 : 0; }

int nlos_ctx_create(int device, nlos_ctx** out) {
    if (!out) return fail(NLOS_ERR_ARG, "nlos_ctx_create: out is NULL");
    int n = nlos_device_count();
    if (n <= 0) return fail(NLOS_ERR_NO_DEVICE, "no HIP device visible: libnlos_hip needs an AMD GPU (no CPU fallback)");
    if (device < 0 || device >= n) return fail(NLOS_ERR_ARG, "nlos_ctx_create: bad device index");
    nlos_ctx* c = new nlos_ctx();
    c->device = device;
    *out = c;
    return NLOS_OK;
}

void nlos_ctx_destroy(nlos_ctx* c) {
    if (!c) return;
    DeviceGuard g(c->device);
    DevBuf* all[] = {&c->keys0, &c->keys1, &c->idx0, &c->idx1, &c->child, &c->range, &c->parent, &c->arrive,
                     &c->box, &c->status, &c->nodes, &c->tris, &c->facerec, &c->face_id, &c->tri_zmin, &c->vis, &c->diff,
                     &c->fine, &c->taps, &c->rows_tmp, &c->grad_tmp, &c->live, &c->reg_normal, &c->reg_area, &c->reg_owner, &c->vis2, &c->tile_list, &c->tile_count, &c->cov, &c->lazy_flag, &c->vis_items, &c->prod_rec, &c->prod_pairs, &c->geo, &c->src_perm};
    for (DevBuf* b : all) b->release();
    for (DevBuf& b : c->io) b.release();
    for (hipEvent_t& e : c->ring) if (e) { hipError_t r = hipEventDestroy(e); (void)r; e = nullptr; }
    if (c->status_ev) { hipError_t r = hipEventDestroy(c->status_ev); (void)r; }
    if (c->h_status) { hipError_t r = hipHostFree(c->h_status); (void)r; }
    if (c->host_p1) { hipError_t r = hipEventDestroy(c->host_p1); (void)r; }
    if (c->host_up) { hipError_t r = hipEventDestroy(c->host_up); (void)r; }
    if (c->host_render) { hipError_t r = hipStreamDestroy(c->host_render); (void)r; }
    if (c->host_copy) { hipError_t r = hipStreamDestroy(c->host_copy); (void)r; }
    delete c;
}

int64_t nlos_ctx_scratch_bytes(const nlos_ctx* c) {
    if (!c) return 0;
    const DevBuf* all[] = {&c->keys0, &c->keys1, &c->idx0, &c->idx1, &c->child, &c->range, &c->parent, &c->arrive,
                           &c->box, &c->status, &c->nodes, &c->tris, &c->facerec, &c->face_id, &c->tri_zmin, &c->vis, &c->diff,
                           &c->fine, &c->taps, &c->rows_tmp, &c->grad_tmp, &c->live, &c->reg_normal, &c->reg_area, &c->reg_owner, &c->vis2, &c->tile_list, &c->tile_count, &c->cov, &c->lazy_flag, &c->vis_items, &c->prod_rec, &c->prod_pairs, &c->geo, &c->src_perm};
    int64_t s = 0;
    for (const DevBuf* b : all) s += (int64_t)b->cap;
    for (const DevBuf& b : c->io) s += (int64_t)b.cap;
    return s;
}

void nlos_ctx_enable_timing(nlos_ctx* c, int enable) {
    if (!c) return;
    DeviceGuard g(c->device);
    c->timing = enable != 0;
    if (c->timing && c->ring.empty()) {
        c->ring.assign((size_t)nlos_ctx::kRing * 5, nullptr);
        for (hipEvent_t& e : c->ring) { hipError_t r = hipEventCreate(&e); (void)r; }
    }
    c->ring_head = 0; c->ring_count = 0; c->ev_valid = false;
}

void nlos_ctx_timing_reset(nlos_ctx* c) {
    if (!c) return;
    c->ring_head = 0; c->ring_count = 0; c->ev_valid = false;
}

static int slot_times(nlos_ctx* c, int slot, float* ms4) {
    hipEvent_t* ev = &c->ring[(size_t)slot * 5];
    for (int i = 0; i < 4; ++i) {
        ms4[i] = 0.0f;
        hipError_t e = hipEventElapsedTime(&ms4[i], ev[i], ev[i + 1]);
        if (e != hipSuccess) return fail(NLOS_ERR_HIP, std::string("hipEventElapsedTime: ") + hipGetErrorString(e));
    }
    return NLOS_OK;
}

int nlos_ctx_last_timing(nlos_ctx* c, float* ms4) {
    if (!c || !ms4) return fail(NLOS_ERR_ARG, "nlos_ctx_last_timing: NULL argument");
    if (!c->timing || c->ring_count == 0) return fail(NLOS_ERR_ARG, "timing not enabled or no render recorded");
    DeviceGuard g(c->device);
    int slot = (c->ring_head + nlos_ctx::kRing - 1) % nlos_ctx::kRing;
    return slot_times(c, slot, ms4);
}

int nlos_ctx_timing_mean(nlos_ctx* c, float* ms4, int* count) {
    if (!c || !ms4) return fail(NLOS_ERR_ARG, "nlos_ctx_timing_mean: NULL argument");
    if (!c->timing || c->ring_count == 0) return fail(NLOS_ERR_ARG, "timing not enabled or no render recorded");
    DeviceGuard g(c->device);
    double acc[4] = {0, 0, 0, 0};
    for (int k = 0; k < c->ring_count; ++k) {
        int slot = (c->ring_head + nlos_ctx::kRing - 1 - k) % nlos_ctx::kRing;
        float t[4];
        int rc = slot_times(c, slot, t);
        if (rc) return rc;
        for (int i = 0; i < 4; ++i) acc[i] += t[i];
    }
    for (int i = 0; i < 4; ++i) ms4[i] = (float)(acc[i] / c->ring_count);
    if (count) *count = c->ring_count;
    // this call has waited for the renders' events: a bad-face-index flag of any of their scene builds has arrived
    return check_status(c, true);
}

void nlos_render_args_init(nlos_render_args* a) {
    if (!a) return;
    std::memset(a, 0, sizeof(*a));
    a->refine_scale = 1;
    a->sigma_bin = 1;
    a->normal_term = -1;
    a->clamp = 1;
    a->vertex_num = -1;
}

static int render_product(nlos_ctx* c, const nlos_render_args* a, void* stream);

// argument checks shared by nlos_render and the product's fast path (which does not go through nlos_render's own):
// time window, temporal kernel, outputs.  0 or a failed status.
static int check_window_and_taps(const nlos_render_args* a) {
    if (!(a->resolution > 0.0f) || !(a->upper_bound > a->lower_bound))
        if (a->mode != NLOS_MODE_INTENSITY) return fail(NLOS_ERR_ARG, "nlos_render: need resolution > 0 and upper_bound > lower_bound");
    if (a->refine_scale < 1 || a->sigma_bin < 1) return fail(NLOS_ERR_ARG, "nlos_render: refine_scale and sigma_bin must be >= 1");
    // the temporal kernel's taps are staged in LDS by the smoothing and gradient kernels
    if ((long long)4 * a->refine_scale * a->sigma_bin + a->refine_scale > 2048 || a->jitter_length > 2048)
        return fail(NLOS_ERR_ARG, "nlos_render: temporal kernel longer than 2048 taps (4 * refine_scale * sigma_bin + 1, or jitter_length)");
    return NLOS_OK;
}

}  // extern "C" (re-opened below)
namespace {
// Host drop-ins only (section 1): called by nlos_render on its stream between pass 1 and the first kernel that reads
// `data` / `weight` -- the host entry uploads those two matrices THERE, on its copy stream, while pass 1 runs, and starts
// the download of the finished rows behind pass 2.  Null for every other caller.
struct AfterPass1 { int (*fn)(void* self, hipStream_t st); void* self; };
thread_local AfterPass1* tl_after_pass1 = nullptr;
}  // namespace
extern "C" {

int nlos_render(nlos_ctx* c, const nlos_render_args* a, void* stream) {
    if (c && a && a->n_sensors > 0) return render_product(c, a, stream);
    if (!c || !a) return fail(NLOS_ERR_ARG, "nlos_render: NULL ctx/args");
    if (a->F <= 0 || a->V <= 0) return fail(NLOS_ERR_ARG, "nlos_render: empty mesh");
    if (a->L < 0) return fail(NLOS_ERR_ARG, "nlos_render: negative source count");
    if (!a->vertices || !a->faces) return fail(NLOS_ERR_ARG, "nlos_render: vertices/faces are NULL");
    if (a->L > 0 && (!a->origin || !a->normal)) return fail(NLOS_ERR_ARG, "nlos_render: origin/normal are NULL");
    if (a->num_samples <= 0) return fail(NLOS_ERR_ARG, "nlos_render: num_samples must be positive");
    if (int rcw = check_window_and_taps(a)) return rcw;
    const int mode = a->mode;
    const bool needs_grad = mode == NLOS_MODE_GRADIENT || mode == NLOS_MODE_GRAD_ALBEDO || mode == NLOS_MODE_GRAD_ALPHA ||
                            mode == NLOS_MODE_GRADIENT_V1;
    if (needs_grad && !a->residual && (!a->data || !a->transient))
        return fail(NLOS_ERR_ARG, "nlos_render: gradient modes need data and transient (or residual)");
    if (a->residual && mode != NLOS_MODE_GRADIENT) return fail(NLOS_ERR_ARG, "nlos_render: residual is only valid in GRADIENT mode");
    if (a->reuse_visibility && !a->residual) return fail(NLOS_ERR_ARG, "nlos_render: reuse_visibility needs residual");
    if ((mode == NLOS_MODE_GRADIENT || mode == NLOS_MODE_GRADIENT_V1 || mode == NLOS_MODE_VERTEX_GRADIENT) && !a->gradient)
        return fail(NLOS_ERR_ARG, "nlos_render: gradient output is NULL");
    if ((mode == NLOS_MODE_GRAD_ALBEDO || mode == NLOS_MODE_GRAD_ALPHA) && !a->scalar_out)
        return fail(NLOS_ERR_ARG, "nlos_render: scalar_out is NULL");
    if (mode == NLOS_MODE_TRANSIENT && !a->transient) return fail(NLOS_ERR_ARG, "nlos_render: transient is NULL");
    if (mode == NLOS_MODE_INTENSITY && !a->intensity) return fail(NLOS_ERR_ARG, "nlos_render: intensity is NULL");
    if (mode < 0 || mode > NLOS_MODE_GRADIENT_V1) return fail(NLOS_ERR_ARG, "nlos_render: unknown mode");
    if (mode == NLOS_MODE_VERTEX_GRADIENT && (a->vertex_num < 0 || a->vertex_num >= a->V))
        return fail(NLOS_ERR_ARG, "nlos_render: vertex_num out of range");
    const bool jitter = a->jitter_weight != nullptr;
    if (jitter) {
        if (mode != NLOS_MODE_TRANSIENT && mode != NLOS_MODE_GRADIENT)
            return fail(NLOS_ERR_ARG, "nlos_render: the jitter kernel applies to TRANSIENT and GRADIENT modes only");
        if (a->jitter_length < 1 || a->jitter_offset < 0 || a->jitter_offset >= a->jitter_length)
            return fail(NLOS_ERR_ARG, "nlos_render: need 0 <= jitter_offset < jitter_length");
        if (mode == NLOS_MODE_GRADIENT && !a->jitter_grad) return fail(NLOS_ERR_ARG, "nlos_render: jitter_grad is NULL");
        if (a->use_ggx || a->sensor || !a->clamp || (mode == NLOS_MODE_GRADIENT && a->albedo))
            return fail(NLOS_ERR_ARG, "nlos_render: the jitter variant is confocal, Lambertian, without per-vertex albedo in the gradient");
    }
    if (a->sensor) {
        if (!a->sensor_normal) return fail(NLOS_ERR_ARG, "nlos_render: sensor needs sensor_normal");
        if (mode != NLOS_MODE_TRANSIENT && mode != NLOS_MODE_GRADIENT)
            return fail(NLOS_ERR_ARG, "nlos_render: non-confocal pairs support TRANSIENT and GRADIENT modes only");
        if (!a->clamp) return fail(NLOS_ERR_ARG, "nlos_render: non-confocal pairs use clamped form factors");
    }

    DeviceGuard guard(c->device);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int L = a->L, nF = a->F, nV = a->V;
    const float lb = a->lower_bound, ub = a->upper_bound, res = a->resolution;
    const int T = mode == NLOS_MODE_INTENSITY ? 1 : nlos_num_bins(lb, ub, res);
    if (T <= 0) return fail(NLOS_ERR_ARG, "nlos_render: zero bins");
    const int spt = 1 + ((a->num_samples - 1) / nF);
    const int vis_words = (spt + 31) / 32;

    int rc = check_status(c, false);
    if (rc) return rc;
    nlos::LaunchNote note;
    struct NoteScope { NoteScope(nlos::LaunchNote* n) { nlos::tl_note = n; } ~NoteScope() { nlos::tl_note = nullptr; } } note_scope(&note);
    std::memset(&c->path, 0, sizeof(c->path));
    c->path.workgroups = c->path.coarsened = c->path.big_lds = c->path.bvh_queries = -1;
    c->path.rays_traced = c->path.samples_accepted = -1;
    c->path_items_sources = 0;
    c->path_retry_workgroups = 0;

    mark(c, 0, st);
    const int tile_threshold_b = nlos::env_switches().tile_threshold;
    const bool lazy_enabled = nlos::env_switches().lazy_tree;
    // the single-workgroup grid back-end reads the records and the root box only: build the tree when (if) it is needed
    const bool v1_point = a->v1_sampled_point && mode == NLOS_MODE_TRANSIENT && !a->sensor;      // BVH back-end only
    const bool lazy_build = lazy_enabled && nF <= tile_threshold_b && a->force_bvh != 1 && !v1_point;
    rc = ensure_bvh(c, a->vertices, nV, a->faces, nF, a->reuse_bvh != 0, a->mesh_generation, st, lazy_build);
    if (rc) return rc;
    if (!c->tree_complete) note.lazy_build = &c->lazy_args;
    mark(c, 1, st);

    nlos::SourceView src;
    src.origin = a->origin; src.normal = a->normal; src.L = L;
    src.source_offset = a->source_offset; src.total_sources = a->total_sources;
    src.source_stride = a->shared_samples ? 0 : (a->source_stride > 1 ? a->source_stride : 1);
    src.n_sensors = 0;
    src.sensor = a->sensor; src.sensor_normal = a->sensor ? a->sensor_normal : nullptr;

    nlos::SampleParams sp;
    sp.seed = a->seed; sp.spt = spt; sp.lb = lb; sp.ub = ub;
    sp.clamp = a->clamp; sp.use_ggx = a->use_ggx; sp.ggx_alpha = a->ggx_alpha;
    sp.sampled_point = (a->v1_sampled_point && mode == NLOS_MODE_TRANSIENT && !a->sensor) ? 1 : 0;

    const float* vn = a->vertex_normal;
    const float* alb = a->albedo;
    if (mode == NLOS_MODE_VERTEX_GRADIENT || mode == NLOS_MODE_GRADIENT_V1) { vn = nullptr; alb = nullptr; sp.use_ggx = 0; }
    if (mode == NLOS_MODE_GRAD_ALBEDO) { vn = nullptr; sp.use_ggx = 0; }
    if (mode == NLOS_MODE_GRAD_ALPHA) { sp.use_ggx = 1; }
    if (mode == NLOS_MODE_GRADIENT_V1) sp.clamp = 1;
    nlos::SceneView sc = scene_view(c, nF, nV, vn, alb);

    // ---- pass 1 -------------------------------------------------------------------
    const bool two_pass = mode != NLOS_MODE_TRANSIENT && mode != NLOS_MODE_INTENSITY;
    int fwd_refine = a->refine_scale;
    // smoothed_transient/stratifiedStreamedGradientRenderer.cpp:521-524
    if (two_pass) fwd_refine = a->sigma_bin < 5 ? 1 : a->refine_scale;
    if (mode == NLOS_MODE_GRADIENT_V1 || mode == NLOS_MODE_INTENSITY || jitter) fwd_refine = 1;
    const int rb = T * fwd_refine;

    nlos::ForwardArgs fa;
    fa.sc = sc; fa.src = src; fa.sp = sp;
    fa.sp.res = fwd_refine > 1 ? res / fwd_refine : res;
    fa.sp.nbins = rb;
    fa.vis = nullptr; fa.vis_words = vis_words;
    fa.intensity = a->intensity; fa.mode_intensity = mode == NLOS_MODE_INTENSITY ? 1 : 0;
    fa.force_bvh = (a->force_bvh == 1 || v1_point) ? 1 : 0;
    fa.dbg = nullptr;
    fa.rec_d = fa.rec_ff = nullptr;
    fa.rec_ext = nullptr; fa.rec_ext_stride = 0;
    fa.perm = nullptr;
    fa.geo = nullptr; fa.geo_stride = 0; fa.geo_sources = 0;
    // Residual formed by pass 2 itself (round 4): vertex gradient of this call's own forward rows, confocal; k_residual's
    // other chores (clearing the gradient output, the pathlengths) then ride in the grid kernel's first workgroups, and the
    // step holds no residual launch.  Whether the launcher that runs carries them is known after the launch (note.prologue_done).
    const bool fuse_enabled = nlos::env_switches().fuse_residual;
    const bool fuse_candidate = fuse_enabled && mode == NLOS_MODE_GRADIENT && !a->residual && !a->sensor && !a->reuse_visibility;
    fa.zero = nullptr; fa.zero_n = 0; fa.pathlengths = nullptr; fa.path_lb = lb; fa.path_res = res; fa.path_T = T;
    if (fuse_candidate) {
        if (a->zero_gradient) { fa.zero = a->gradient; fa.zero_n = 3 * (size_t)nV; }
        fa.pathlengths = a->pathlengths;
    }
    fa.live = nullptr;
    fa.cov = nullptr;
    fa.tile_list = nullptr;
    fa.tile_count = nullptr;
    fa.retry = nullptr;
    fa.need_tree = c->tree_complete ? nullptr : c->lazy_flag.as<int>();
    fa.tiles_x = fa.tiles_y = fa.tile_cap = 0;
    const int tile_threshold = nlos::env_switches().tile_threshold;
    int chunk_L = L > 0 ? L : 1;                       // sources per pass-1 launch
    if (nF <= tile_threshold && a->force_bvh != 1 && !v1_point) {
        rc = c->live.ensure(sizeof(uint16_t) * (size_t)(L > 0 ? L : 1) * nF + 16);
        if (!rc) rc = c->cov.ensure(sizeof(uint16_t) * (size_t)(L > 0 ? L : 1) * nF + 16);
        if (!rc) rc = c->tile_count.ensure(sizeof(int) * (size_t)(L > 0 ? L : 1) + 16);
        if (rc) return rc;
        fa.live = c->live.as<uint16_t>();
        fa.cov = c->cov.as<uint16_t>();
        fa.retry = c->tile_count.as<int>();
    } else if (a->force_bvh != 1 && !v1_point && L > 0) {
        // tiled grid: ~3000 triangles per slope-space tile on average (small tiles leave the 512 threads idle); the densest tiles of a closed surface
        // (front + back side, several depth layers) hold up to ~4.5x the mean, and the subset capacity is bounded by
        // the 14-bit entry index (overflowing tiles fall back to the BVH query by themselves);
        // scratch = 8 B per (source, tile, slot)
        const int tile_tris = nlos::env_switches().tile_tris;
        const int nt = (nF + tile_tris - 1) / tile_tris;
        int side = 1;
        while (side * side < nt) ++side;
        const long long tiles = (long long)side * side;
        long long tcap = 6LL * nF / tiles + 512;
        if (tcap > 16383) tcap = 16383;
        if (a->force_bvh == 2) tcap = 64;              // diagnostic: force the subset-overflow fallback of the tiles
        // the per-(source, tile) subsets are the largest scratch of the path (8 B per slot): bounded to 32 GB by
        // rendering the sources in chunks (NLOS_TILE_SCRATCH_MAX overrides the bound; nlos_ctx_last_path reports
        // the number of chunks)
        const unsigned long long scratch_max = nlos::env_switches().tile_scratch_max;
        const unsigned long long per_source = (unsigned long long)tiles * (unsigned long long)tcap;
        unsigned long long max_l = scratch_max / (8ull * per_source);
        if (max_l < 1) max_l = 1;
        if ((unsigned long long)L > max_l) chunk_L = (int)max_l;
        const unsigned long long slots = (unsigned long long)chunk_L * per_source;
        rc = c->live.ensure(sizeof(uint16_t) * slots + 16);
        if (!rc) rc = c->cov.ensure(sizeof(uint16_t) * slots + 16);
        if (!rc) rc = c->tile_list.ensure(sizeof(uint32_t) * slots + 16);
        if (!rc) rc = c->tile_count.ensure(sizeof(int) * 2 * (size_t)chunk_L * tiles + 16);
        if (rc) return rc;
        fa.tile_count = c->tile_count.as<int>();
        fa.retry = fa.tile_count + (size_t)chunk_L * tiles;
        fa.live = c->live.as<uint16_t>();
        fa.cov = c->cov.as<uint16_t>();
        fa.tile_list = c->tile_list.as<uint32_t>();
        fa.tiles_x = fa.tiles_y = side;
        fa.tile_cap = (int)tcap;
    }
#if defined(NLOS_FWD_STAMPS) || defined(NLOS_FWD_STAMPS_LIGHT)
    HIP_TRY(hipMemsetAsync(c->status.p, 0, 48 * sizeof(long long), st));
    fa.dbg = c->status.as<long long>();     // 40 x int64 (diagnostic build only)
#endif
    fa.rows = nullptr;
    fa.vis2 = nullptr;
    if (a->sensor && fa.live) {
        // sensor-leg visibility bits of the two-pass grid path for non-confocal pairs
        rc = c->vis2.ensure(sizeof(uint32_t) * (size_t)(L > 0 ? L : 1) * vis_words * nF + 16);
        if (rc) return rc;
        fa.vis2 = c->vis2.as<uint32_t>();
    }
    nlos_ctx::VisKey key;
    key.L = L; key.F = nF; key.V = nV; key.spt = spt; key.off = a->source_offset; key.stride = a->shared_samples ? 0 : (a->source_stride > 1 ? a->source_stride : 1); key.seed = a->seed;
    key.lb = lb; key.ub = ub;
    key.feat = (vn ? 1 : 0) | (alb ? 2 : 0) | (sp.use_ggx ? 4 : 0) | (sp.clamp ? 8 : 0) | (a->sensor ? 16 : 0);
    key.mesh_gen = c->mesh_gen;
    const bool skip_pass1 = a->reuse_visibility != 0;
    if (skip_pass1) {
        // the cache must be the one the caller saw being recorded (generation), recorded on the tree the ctx
        // holds now (pass 2 reads the sorted face records of that tree), for the same sources and samples
        const nlos_ctx::VisKey& k = c->vis_key;
        if (a->visibility_generation == 0 || a->visibility_generation != c->vis_gen || k.mesh_gen != c->mesh_gen)
            return fail(NLOS_ERR_ARG, "nlos_render: reuse_visibility requested but the context no longer holds the visibility "
                                      "cache of that generation (another render or scene build ran on it since)");
        if (!(k.L == key.L && k.F == key.F && k.V == key.V && k.spt == key.spt && k.off == key.off && k.stride == key.stride && k.seed == key.seed &&
              k.lb == key.lb && k.ub == key.ub && k.feat == key.feat))
            return fail(NLOS_ERR_ARG, "nlos_render: reuse_visibility requested but the cache does not match this render");
    }
    fa.vis_items = nullptr; fa.items_stride = 0;
    if (two_pass || a->keep_visibility) {
        rc = c->vis.ensure(sizeof(uint32_t) * (size_t)L * vis_words * nF + 16);
        if (rc) return rc;
        fa.vis = c->vis.as<uint32_t>();
        // item masks beside the words: the launcher that runs decides which of the two pass 1 records (the grid kernel
        // of confocal renders takes the masks; pairs, tiled grid and BVH back-end keep the per-face words)
        const bool items_enabled = nlos::env_switches().vis_items;
        if (items_enabled && !skip_pass1 && spt <= 32 && fa.live && !fa.tile_list) {
            const int stride = (int)(((size_t)nF * spt + 63) / 64) + 2;
            rc = c->vis_items.ensure(sizeof(unsigned long long) * (size_t)L * stride + 16);
            if (rc) return rc;
            fa.vis_items = c->vis_items.as<unsigned long long>();
            fa.items_stride = stride;
            // geometry cache for the pass 2 of THIS call (vertex-gradient modes, confocal): 24 B per ray of the live lists
            const bool geo_enabled = nlos::env_switches().geo_cache;
            // (spt <= 8 only: a 64-ray item writes spt segments of 64 / spt consecutive records; at spt = 19 -- the 1 055-face
            // mannequin -- those are 54-byte pieces, and the streaming stores of partial lines triple pass 1: 1.09 -> 2.80 ms,
            // profiles/r04_side_bench.log; such renders keep the recomputing pass 2)
            if (geo_enabled && spt <= nlos::env_switches().geo_max_spt && !a->sensor && !jitter && (mode == NLOS_MODE_GRADIENT || mode == NLOS_MODE_GRADIENT_V1 || mode == NLOS_MODE_TRANSIENT)) {
                // an optimisation, never a reason to fail: bounded -- NLOS_GEO_CACHE_MAX_GB if set, else 32 GB or half of what
                // the device has free, whichever is less -- and skipped when the allocation does not succeed (pass 2 then
                // regenerates its samples).  DevBuf::try_ensure: exact size, the old buffer survives a failed growth, the
                // last-error string stays clean.
                const double geo_env_gb = nlos::env_switches().geo_cache_max_gb;
                const size_t geo_bytes = sizeof(float) * 6 * (size_t)L * (size_t)nF * (size_t)spt + 16;
                size_t geo_max = (size_t)32 << 30;
                if (geo_env_gb >= 0.0) {
                    geo_max = (size_t)(geo_env_gb * 1073741824.0);
                } else if (geo_bytes > c->geo.cap) {
                    size_t free_b = 0, total_b = 0;
                    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) geo_max = std::min(geo_max, (free_b + c->geo.cap) / 2);
                    else (void)hipGetLastError();
                }
                if (geo_bytes <= geo_max && c->geo.try_ensure(geo_bytes)) {
                    fa.geo = c->geo.as<float>();
                    fa.geo_stride = nF * spt;
                    fa.geo_sources = L;
                }
            }
        }
        // a pass 1 that records no geometry invalidates whatever the cache held (new visibility generation): a large buffer goes
        // back to the device (small ones stay: renders of different kinds may alternate, and hipMalloc / hipFree synchronise)
        if (!skip_pass1 && !fa.geo && c->geo.p && c->geo.cap >= ((size_t)4 << 30)) c->geo.release();
        c->vis_key = key;
        if (!skip_pass1) c->vis_gen = ++c->gen_counter;
    } else {
        c->vis_key = nlos_ctx::VisKey();
        c->vis_gen = 0;
    }
    double* transient = a->transient;
    if (mode == NLOS_MODE_GRADIENT && a->residual && !transient) {
        rc = c->rows_tmp.ensure(sizeof(double) * (size_t)L * T + 16);
        if (rc) return rc;
        transient = c->rows_tmp.as<double>();
    }
    if (mode == NLOS_MODE_VERTEX_GRADIENT && !transient) {
        rc = c->rows_tmp.ensure(sizeof(double) * (size_t)L * T + 16);
        if (rc) return rc;
        transient = c->rows_tmp.as<double>();
    }
    if (mode != NLOS_MODE_INTENSITY) {
        if (fwd_refine > 1 || jitter) {
            rc = c->fine.ensure(sizeof(double) * (size_t)L * rb + 16);
            if (rc) return rc;
            fa.rows = c->fine.as<double>();
        } else {
            fa.rows = transient;
        }
    }
    // sources in Z-order of the wall (ForwardArgs::perm): one-workgroup-per-source launches of confocal renders and of pairs
    if (!skip_pass1 && fa.live && !fa.tile_list && L >= 512 && L <= 8192 && nlos::env_switches().fwd_order) {
        if (c->perm_origin != a->origin || c->perm_L != L) {
            rc = c->src_perm.ensure(sizeof(int) * (size_t)L + 16);
            if (rc) return rc;
            c->perm_origin = nullptr; c->perm_L = -1;
            if (nlos::launch_order_sources(a->origin, L, c->src_perm.as<int>(), st)) { c->perm_origin = a->origin; c->perm_L = L; }
        }
        if (c->perm_origin == a->origin && c->perm_L == L) fa.perm = c->src_perm.as<int>();
    }
    int n_chunks = 0;
    for (int l0 = 0; !skip_pass1 && l0 < L; l0 += chunk_L, ++n_chunks) {
        nlos::ForwardArgs fc = fa;
        fc.src.L = L - l0 < chunk_L ? L - l0 : chunk_L;
        fc.src.origin += 3 * (size_t)l0; fc.src.normal += 3 * (size_t)l0;
        if (fc.src.sensor) { fc.src.sensor += 3 * (size_t)l0; fc.src.sensor_normal += 3 * (size_t)l0; }
        fc.src.source_offset += (long long)l0 * fc.src.source_stride;
        if (fc.rows) fc.rows += (size_t)l0 * rb;
        if (fc.vis) fc.vis += (size_t)l0 * vis_words * nF;
        if (fc.vis2) fc.vis2 += (size_t)l0 * vis_words * nF;
        if (fc.tile_count) fc.retry = fc.tile_count + (size_t)fc.src.L * fc.tiles_x * fc.tiles_y;   // flags follow this chunk's subset sizes
        nlos::launch_forward(fc, st);
    }
    c->path.backend = note.backend; c->path.reason = note.reason; c->path.grid_R = note.grid_R;
    c->path.tiles = note.tiles; c->path.tile_cap = note.tile_cap; c->path.chunks = n_chunks;
    c->path.rows_in_lds = note.rows_in_lds;
    c->path_retry_workgroups = note.retry_workgroups;
    if (note.tree_built) c->tree_complete = true;
    if (!skip_pass1 && fa.vis) { c->vis_is_items = note.vis_items != 0; c->vis_items_stride = fa.items_stride; }
    c->path_items_sources = (!skip_pass1 && fa.vis && note.vis_items != 0 && n_chunks == 1) ? L : 0;
    if (!skip_pass1) {     // the geometry cache belongs to the visibility generation it was recorded with
        c->geo_gen = (fa.geo && fa.vis && note.vis_items != 0) ? c->vis_gen : 0;
        c->geo_stride = fa.geo_stride;
        c->geo_sources = fa.geo_sources;
    }
#ifdef NLOS_FWD_STAMPS_LIGHT
    {
        long long h[48];
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipMemcpy(h, c->status.p, sizeof(h), hipMemcpyDeviceToHost));
        double tot = (double)(h[0] + h[1] + h[2] + h[3] + h[4] + h[5] + h[6]);
        std::fprintf(stderr, "[fwd light stamps] setup %.1f%% count %.1f%% scan %.1f%% fill %.1f%% live-buckets %.1f%% trace (wave 0) %.1f%% tail %.1f%% | Mcycles per source %.3f\n",
                     100 * h[0] / tot, 100 * h[1] / tot, 100 * h[2] / tot, 100 * h[3] / tot, 100 * h[4] / tot, 100 * h[5] / tot, 100 * h[6] / tot,
                     tot / 1e6 / (double)(L > 0 ? L : 1));
    }
#endif
#ifdef NLOS_FWD_STAMPS
    {
        long long h[48];
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipMemcpy(h, c->status.p, sizeof(h), hipMemcpyDeviceToHost));
        double tot = (double)(h[0] + h[1] + h[2] + h[3] + h[4] + h[5]);
        std::fprintf(stderr, "[fwd stamps] setup %.1f%% count %.1f%% scan %.1f%% fill %.1f%% live-buckets %.1f%% trace %.1f%% | Mcycles per source %.3f\n",
                     100 * h[0] / tot, 100 * h[1] / tot, 100 * h[2] / tot, 100 * h[3] / tot, 100 * h[4] / tot, 100 * h[5] / tot,
                     tot / 1e6 / (double)(L > 0 ? L : 1));
        double tt = (double)(h[14] + h[15] + h[16] + h[17]);
        std::fprintf(stderr, "[fwd trace shares] generate %.1f%% filter-scan %.1f%% exact-rounds %.1f%% histogram+vis %.1f%%\n",
                     100 * h[14] / tt, 100 * h[15] / tt, 100 * h[16] / tt, 100 * h[17] / tt);
        double rw = (double)h[8] / 64.0;
        std::fprintf(stderr, "[fwd counters] rays %lld entries/source %.0f | scan: lane-it %.1f/ray, wave-it %.1f/ray-wave | queued %.2f/ray | exact rounds %.2f/ray-wave\n",
                     h[8], (double)h[13] / (double)(L > 0 ? L : 1), (double)h[9] / (double)(h[8] ? h[8] : 1), (double)h[10] / rw,
                     (double)h[11] / (double)(h[8] ? h[8] : 1), (double)h[12] / rw);
        std::fprintf(stderr, "[fwd occlusion] rays found occluded %.1f%% | exact tests %lld, of which an occluder %.1f%%\n",
                     100.0 * (double)h[19] / (double)(h[8] ? h[8] : 1), h[6], 100.0 * (double)h[18] / (double)(h[6] ? h[6] : 1));
        // round 6: trip counts for the dynamic instruction histogram (tools/dynamic_histogram.py) and the depth-order bound
        std::fprintf(stderr, "[fwd trips] items %lld walk_trips %lld push_slots %lld exact_rounds %lld count_face_waves %lld count_cell_lane_it %lld count_cell_wave_it %lld "
                             "fill_face_waves %lld fill_entry_lane_it %lld fill_entry_wave_it %lld | walked entries %lld of which fail the depth term %lld (%.1f%%)\n",
                     h[26], h[10], h[7], h[12], h[27], h[29], h[30], h[33], h[32], h[31], h[9], h[28], 100.0 * (double)h[28] / (double)(h[9] ? h[9] : 1));
        std::fprintf(stderr, "[fwd pair filter bound] exact tests %lld | would survive true sub-cell coverage at 3 / 4 / 6 / 8 per side: %.1f%% %.1f%% %.1f%% %.1f%% | ideal point test %.1f%%\n",
                     h[6], 100.0 * h[34] / (double)(h[6] ? h[6] : 1), 100.0 * h[35] / (double)(h[6] ? h[6] : 1), 100.0 * h[36] / (double)(h[6] ? h[6] : 1),
                     100.0 * h[37] / (double)(h[6] ? h[6] : 1), 100.0 * h[38] / (double)(h[6] ? h[6] : 1));
        if (h[25] > 0) {
            // s_memtime counts shader-clock cycles, s_memrealtime the constant 100 MHz reference: their ratio over the
            // workgroups' lifetimes is the engine clock the forward kernel ran at (tools/issue_model.py uses it)
            const double ghz = 0.1 * (double)h[24] / (double)h[25];
            std::fprintf(stderr, "[fwd clock] %.4f GHz over %d sources\n", ghz, L);
            if (const char* cj = std::getenv("NLOS_CLOCK_JSON")) {
                if (FILE* fj = std::fopen(cj, "w")) {
                    std::fprintf(fj, "{\"clock_ghz\": %.5f, \"sources\": %d, \"basis\": \"sum of s_memtime ticks / sum of s_memrealtime ticks (100 MHz) over the "
                                     "workgroups of k_forward_grid, -DNLOS_FWD_STAMPS build\"}\n", ghz, L);
                    std::fclose(fj);
                }
            }
        }
        if (!fa.tile_list && fa.cov && (nF & 1) == 0 && L > 0 && note.backend == NLOS_PATH_GRID) {
            // per-source lifetimes of the grid kernel's workgroups (cycles): the spread that bounds strong scaling
            std::vector<unsigned long long> cyc((size_t)L);
            if (hipMemcpy2D(cyc.data(), 8, fa.cov, sizeof(uint16_t) * (size_t)nF, 8, (size_t)L, hipMemcpyDeviceToHost) == hipSuccess) {
                std::sort(cyc.begin(), cyc.end());
                double sum = 0; for (auto v : cyc) sum += (double)v;
                const double mean = sum / L;
                auto pct = [&](double p) { return (double)cyc[(size_t)std::min<double>(L - 1, p * (L - 1))]; };
                std::fprintf(stderr, "[fwd per-source cycles] min %.0f p50 %.0f mean %.0f p90 %.0f p99 %.0f max %.0f | max/mean %.3f p99/mean %.3f over %d sources\n",
                             (double)cyc.front(), pct(0.5), mean, pct(0.9), pct(0.99), (double)cyc.back(), cyc.back() / mean, pct(0.99) / mean, L);
            }
        }
        if (fa.tile_list)
            std::fprintf(stderr, "[fwd tiles] %d x %d tiles, capacity %d: %lld overflowing tiles, largest subset %lld | entry overflow: %lld workgroups, most entries %lld\n",
                         fa.tiles_x, fa.tiles_y, fa.tile_cap, h[22], h[23], h[20], h[21]);
    }
#endif
    if (!skip_pass1 && mode != NLOS_MODE_INTENSITY && fwd_refine > 1) {
        // Gaussian of the refined histogram (row FD): kernel = the gradient taps' w
        int K = 0;
        rc = ensure_taps(c, 0, fwd_refine, a->sigma_bin, res, st, &K);
        if (rc) return rc;
        nlos::SmoothArgs sm;
        sm.fine = c->fine.as<double>(); sm.transient = transient; sm.kernel = c->taps.as<double>();
        sm.L = L; sm.T = T; sm.refine = fwd_refine; sm.sigma_bin = a->sigma_bin; sm.K = K;
        sm.offset = 2 * fwd_refine * a->sigma_bin;
        nlos::launch_smooth(sm, st);
    }
    if (!skip_pass1 && jitter) {
        // jitter/transient_and_gradient.cpp:331-347: row = conv(histogram, jitter_weight)[offset : offset + T]
        nlos::SmoothArgs sm;
        sm.fine = c->fine.as<double>(); sm.transient = transient; sm.kernel = a->jitter_weight;
        sm.L = L; sm.T = T; sm.refine = 1; sm.sigma_bin = 0; sm.K = a->jitter_length; sm.offset = a->jitter_offset;
        nlos::launch_smooth(sm, st);
    }
    mark(c, 2, st);
    if (tl_after_pass1) {
        AfterPass1* h = tl_after_pass1;
        tl_after_pass1 = nullptr;                  // once per host call
        rc = h->fn(h->self, st);
        if (rc) return rc;
    }

    // ---- residual + pathlengths -----------------------------------------------------
    if (mode == NLOS_MODE_TRANSIENT || (mode == NLOS_MODE_VERTEX_GRADIENT)) {
        if (a->pathlengths) {
            nlos::ResidualArgs ra;
            std::memset(&ra, 0, sizeof(ra));
            ra.pathlengths = a->pathlengths; ra.L = 0; ra.T = T; ra.lb = lb; ra.res = res;
            nlos::launch_residual(ra, st);
        }
    }
    const double* diff_ptr = a->residual;
    // zero_gradient: the output is cleared by the residual kernel of this render (no fill operation of its own)
    double* zero_ptr = nullptr;
    size_t zero_n = 0;
    if (a->zero_gradient && (mode == NLOS_MODE_GRADIENT || mode == NLOS_MODE_VERTEX_GRADIENT)) {
        zero_ptr = a->gradient;
        zero_n = mode == NLOS_MODE_GRADIENT ? 3 * (size_t)nV : 3 * (size_t)T;
    }
    if (needs_grad && a->residual && (a->pathlengths || zero_ptr)) {
        nlos::ResidualArgs ra;
        std::memset(&ra, 0, sizeof(ra));
        ra.pathlengths = a->pathlengths; ra.L = 0; ra.T = T; ra.lb = lb; ra.res = res;
        ra.zero = zero_ptr; ra.zero_n = zero_n;
        nlos::launch_residual(ra, st);
        zero_ptr = nullptr;
    }
    const bool fused_residual = fuse_candidate && note.prologue_done != 0;
    if (fused_residual) {
        rc = c->diff.ensure(sizeof(double) * (size_t)L * T + 16);      // (scratch of the face-major fallback only)
        if (rc) return rc;
        diff_ptr = c->diff.as<double>();
        zero_ptr = nullptr;                                            // cleared by the grid kernel's first workgroups
    }
    if (needs_grad && !a->residual && !fused_residual) {
        rc = c->diff.ensure(sizeof(double) * (size_t)L * T + 16);
        if (rc) return rc;
        nlos::ResidualArgs ra;
        ra.data = a->data; ra.weight = mode == NLOS_MODE_GRADIENT_V1 ? nullptr : a->weight;
        ra.transient = transient; ra.diff = c->diff.as<double>();
        ra.pathlengths = a->pathlengths; ra.L = L; ra.T = T;
        ra.loss_test = mode == NLOS_MODE_GRADIENT_V1 ? 0 : a->loss_test;
        ra.lb = lb; ra.res = res;
        ra.w_width = mode == NLOS_MODE_GRADIENT_V1 ? a->w_width : 0;
        ra.zero = zero_ptr; ra.zero_n = zero_n;
        nlos::launch_residual(ra, st);
        zero_ptr = nullptr;
        diff_ptr = c->diff.as<double>();
    }
    if (zero_ptr) nlos::launch_zero_f64(zero_ptr, zero_n, st);      // (a mode without a residual launch)
    mark(c, 3, st);

    // ---- pass 2 -----------------------------------------------------------------------
    if (two_pass) {
        int K = 0;
        if (jitter) K = a->jitter_length;
        else rc = ensure_taps(c, mode == NLOS_MODE_GRADIENT_V1 ? 1 : 0, a->refine_scale, a->sigma_bin, res, st, &K);
        if (rc) return rc;
        nlos::GradientArgs ga;
        ga.sc = sc; ga.src = src; ga.sp = sp;
        ga.sp.res = res; ga.sp.nbins = T;
        ga.vis = c->vis.as<uint32_t>(); ga.vis_words = vis_words;
        ga.vis_items = c->vis_is_items ? c->vis_items.as<unsigned long long>() : nullptr;
        ga.live = c->live.as<uint16_t>();
        ga.items_stride = c->vis_items_stride;
        ga.vis_scratch = c->vis.as<uint32_t>();
        // the geometry cache recorded together with the visibility cache this pass 2 reads (by pass 1 of this call, or --
        // reuse_visibility -- by the render that recorded that generation), as item masks (the cache's index)
        ga.geo = (c->vis_is_items && c->geo_gen != 0 && c->geo_gen == c->vis_gen && c->geo.p) ? c->geo.as<float>() : nullptr;
        ga.geo_stride = c->geo_stride;
        ga.geo_sources = c->geo_sources;
        ga.inline_residual = fused_residual ? 1 : 0;
        ga.res_data = a->data; ga.res_weight = a->weight; ga.res_transient = transient; ga.res_loss_test = a->loss_test;
        ga.diff_scratch = c->diff.as<double>();
        ga.tap_w = c->taps.as<double>(); ga.tap_delta = ga.tap_w + K; ga.tap_g = ga.tap_w + 2 * K;
        ga.tap_p0 = ga.tap_w + 3 * K; ga.tap_p1 = ga.tap_w + 4 * K + 1; ga.tap_pw = ga.tap_w + 5 * K + 2;
        ga.tap_wt = nullptr; ga.tap_nb = 0;
        ga.K = K;
        const bool v1 = mode == NLOS_MODE_GRADIENT_V1;
        ga.two_rs = v1 ? 0 : 2 * a->refine_scale * a->sigma_bin;
        ga.r_over_res = v1 ? 1.0 : (double)a->refine_scale / (double)res;
        ga.refine = v1 ? 1 : a->refine_scale;
        ga.v1_style = mode == NLOS_MODE_GRADIENT_V1 ? 1 : 0;
        if (mode == NLOS_MODE_GRADIENT && !jitter) {       // per-bin weights of the Gaussian taps (host_taps)
            ga.tap_wt = ga.tap_w + 6 * (size_t)K + 3;
            ga.tap_nb = 4 * a->sigma_bin + 1;
        }
        ga.vertex_num = a->vertex_num;
        ga.diff = diff_ptr;
        switch (mode) {
            case NLOS_MODE_GRADIENT:
                ga.mode = 0;
                ga.normal_term = a->normal_term < 0 ? ((a->testing_flag == 0 && vn != nullptr) ? 1 : 0) : (a->normal_term ? 1 : 0);
                ga.out = a->gradient;
                if (jitter) {
                    ga.mode = 4;
                    ga.tap_w = a->jitter_weight; ga.tap_g = a->jitter_grad;
                    ga.tap_delta = ga.tap_p0 = ga.tap_p1 = nullptr;
                    ga.two_rs = a->jitter_offset;
                }
                break;
            case NLOS_MODE_GRADIENT_V1:
                ga.mode = 0; ga.normal_term = 1; ga.out = a->gradient;
                // v1 zeroes the output (stratified_transient_raytracer/stratifiedStreamedGradientRenderer.cpp:419)
                nlos::launch_zero_f64(a->gradient, 3 * (size_t)nV, st);
                break;
            case NLOS_MODE_GRAD_ALBEDO:
                ga.mode = 1; ga.normal_term = 0; ga.out = a->scalar_out;
                nlos::launch_zero_f64(a->scalar_out, 1, st);
                break;
            case NLOS_MODE_GRAD_ALPHA:
                ga.mode = 2; ga.normal_term = 0; ga.out = a->scalar_out;
                nlos::launch_zero_f64(a->scalar_out, 1, st);
                break;
            default:  // NLOS_MODE_VERTEX_GRADIENT
                ga.mode = 3; ga.normal_term = 1; ga.out = a->gradient;
                ga.diff = transient;   // unused by mode 3; any valid [L,T] buffer
                break;
        }
        ga.lds_grad = 1;          // allowed; launch_gradient decides from the LDS the mesh needs
        ga.compact = 0;
        nlos::launch_gradient(ga, st);
    }
    mark(c, 4, st);
    c->ev_valid = c->timing;
    c->path.gradient_kernel = note.gradient_kernel;
    if (note.err != hipSuccess)
        return fail(NLOS_ERR_HIP, std::string(note.err_what ? note.err_what : "launch") + ": " + hipGetErrorString(note.err));
    HIP_TRY(hipGetLastError());
    return NLOS_OK;
}

// Row N as the product of a laser set and a sensor set (include/nlos_hip.h, nlos_render_args.n_sensors).
//   fast path  (Lambertian, single-workgroup grid, spt <= 32, unrefined rows that fit LDS; face normals, or -- round 6 --
//              vertex normals / albedo with the extended records):
//              one record pass per wall point (k_forward_grid<FEAT, 3>: leg length + form factor of every sample the point
//              sees; extended: + the leg's direction and the normal / albedo at its hit), one combine kernel over the pairs
//              (rows + accepted-sample words), then residual and the pair gradient kernel over the L x S measurements --
//              O(L + S) grid passes instead of 2 L S;
//   otherwise  the pairs are enumerated into scratch arrays and rendered by the pair path on shared samples.
// Both are the same function of the inputs (the product is defined as its pairs).
static int render_product(nlos_ctx* c, const nlos_render_args* a, void* stream) {
    if (a->L < 0) return fail(NLOS_ERR_ARG, "nlos_render (product): negative laser count");
    if (!a->sensor || !a->sensor_normal || (a->L > 0 && (!a->origin || !a->normal)))
        return fail(NLOS_ERR_ARG, "nlos_render (product): laser / sensor arrays are NULL");
    if (a->mode != NLOS_MODE_TRANSIENT && a->mode != NLOS_MODE_GRADIENT)
        return fail(NLOS_ERR_ARG, "nlos_render (product): TRANSIENT and GRADIENT modes only");
    if (a->residual || a->reuse_visibility || a->keep_visibility || a->jitter_weight || !a->clamp)
        return fail(NLOS_ERR_ARG, "nlos_render (product): residual / visibility reuse / jitter / unclamped form factors are not offered");
    if (a->F <= 0 || a->V <= 0 || !a->vertices || !a->faces) return fail(NLOS_ERR_ARG, "nlos_render: empty mesh");
    if (a->num_samples <= 0) return fail(NLOS_ERR_ARG, "nlos_render: num_samples must be positive");
    // (the fast path below does not pass through nlos_render's checks: a refine_scale or sigma_bin below 1 would divide by
    // zero in the tap tables or size them negatively, an inverted window would render garbage)
    if (int rcw = check_window_and_taps(a)) return rcw;
    if (!a->transient) return fail(NLOS_ERR_ARG, "nlos_render: transient is NULL");
    if (a->mode == NLOS_MODE_GRADIENT && (!a->data || !a->gradient))
        return fail(NLOS_ERR_ARG, "nlos_render: gradient modes need data and a gradient output");
    DeviceGuard guard(c->device);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int La = a->L, Sb = a->n_sensors, nF = a->F, nV = a->V;
    if ((long long)La * Sb > (1LL << 30)) return fail(NLOS_ERR_ARG, "nlos_render (product): too many pairs");
    const int P = La * Sb;
    const float lb = a->lower_bound, ub = a->upper_bound, res = a->resolution;
    const int T = nlos_num_bins(lb, ub, res);
    if (T <= 0) return fail(NLOS_ERR_ARG, "nlos_render: zero bins");
    const int spt = 1 + ((a->num_samples - 1) / nF);
    const bool grad = a->mode == NLOS_MODE_GRADIENT;
    const int fwd_refine = grad ? (a->sigma_bin < 5 ? 1 : a->refine_scale) : a->refine_scale;
    const int tile_threshold = nlos::env_switches().tile_threshold;
    const int Ltot = (a->total_sources > 0 ? a->total_sources : La) * Sb;      // measurements the gradient is averaged over
    const bool ext = a->vertex_normal || a->albedo;      // extended records (round 6): normal + albedo of the laser leg, direction of the sensor leg
    const bool fast = !a->product_pairs && !a->use_ggx && a->force_bvh == 0 && nF >= 64 &&
                      nF <= tile_threshold && spt <= 32 && fwd_refine == 1 && (size_t)T * sizeof(double) <= 48 * 1024 && La > 0 &&
                      a->source_stride <= 1;
    // ---- the enumerated pairs, on shared samples (what the product is defined as)
    auto as_pairs = [&]() -> int {
        if (P == 0) return fail(NLOS_ERR_ARG, "nlos_render (product): no pairs");
        int rcp = c->prod_pairs.ensure(sizeof(float) * 12 * (size_t)P + 64);
        if (rcp) return rcp;
        float* pl = c->prod_pairs.as<float>();
        float *pln = pl + 3 * (size_t)P, *ps = pl + 6 * (size_t)P, *psn = pl + 9 * (size_t)P;
        nlos::launch_expand_pairs(a->origin, a->normal, a->sensor, a->sensor_normal, La, Sb, pl, pln, ps, psn, st);
        nlos_render_args b = *a;
        b.n_sensors = 0; b.product_pairs = 0;
        b.origin = pl; b.normal = pln; b.sensor = ps; b.sensor_normal = psn;
        b.L = P; b.shared_samples = 1; b.source_stride = 0; b.total_sources = Ltot;
        return nlos_render(c, &b, stream);
    };
    if (!fast) return as_pairs();

    if (!a->transient) return fail(NLOS_ERR_ARG, "nlos_render: transient is NULL");
    if (grad && (!a->data || !a->gradient)) return fail(NLOS_ERR_ARG, "nlos_render: gradient modes need data and a gradient output");
    int rc = check_status(c, false);
    if (rc) return rc;
    nlos::LaunchNote note;
    struct NoteScope { NoteScope(nlos::LaunchNote* n) { nlos::tl_note = n; } ~NoteScope() { nlos::tl_note = nullptr; } } note_scope(&note);
    std::memset(&c->path, 0, sizeof(c->path));
    c->path.workgroups = c->path.coarsened = c->path.big_lds = c->path.bvh_queries = -1;
    c->path.rays_traced = c->path.samples_accepted = -1;
    c->path_items_sources = 0;
    c->path_retry_workgroups = 0;
    mark(c, 0, st);
    const bool lazy_enabled = nlos::env_switches().lazy_tree;
    rc = ensure_bvh(c, a->vertices, nV, a->faces, nF, a->reuse_bvh != 0, a->mesh_generation, st, lazy_enabled);
    if (rc) return rc;
    if (!c->tree_complete) note.lazy_build = &c->lazy_args;
    mark(c, 1, st);
    nlos::SceneView sc = scene_view(c, nF, nV, a->vertex_normal, a->albedo);

    // ---- record pass: lasers, then sensors (the same records serve both roles when the two sets are one array)
    const bool same_set = a->sensor == a->origin && a->sensor_normal == a->normal && Sb == La;
    const int W = same_set ? La : La + Sb;
    const size_t R = (size_t)nF * (size_t)spt;
    const int Wmax = La > Sb ? La : Sb;
    rc = c->prod_rec.ensure(sizeof(float) * (ext ? 9 : 2) * (size_t)W * R + 64);
    if (!rc) rc = c->live.ensure(sizeof(uint16_t) * (size_t)Wmax * nF + 16);
    if (!rc) rc = c->cov.ensure(sizeof(uint16_t) * (size_t)Wmax * nF + 16);
    if (!rc) rc = c->tile_count.ensure(sizeof(int) * (size_t)Wmax + 16);
    if (rc) return rc;
    float* rec_d = c->prod_rec.as<float>();
    float* rec_ff = rec_d + (size_t)W * R;
    HIP_TRY(hipMemsetAsync(rec_ff, 0, sizeof(float) * (size_t)W * R, st));      // ff = 0: not seen (d is read only where ff > 0)
    float* rec_ext = ext ? rec_ff + (size_t)W * R : nullptr;                     // 7 arrays of W x R floats (ForwardArgs::rec_ext)
    if (ext) HIP_TRY(hipMemsetAsync(rec_d, 0, sizeof(float) * (size_t)W * R, st));   // extended records: d = 0 means "not seen"
    nlos::ForwardArgs fa;
    std::memset(&fa, 0, sizeof(fa));
    fa.sc = sc;
    fa.sp.seed = a->seed; fa.sp.spt = spt; fa.sp.lb = lb; fa.sp.ub = ub; fa.sp.res = res; fa.sp.nbins = T;
    fa.sp.clamp = 1; fa.sp.use_ggx = 0; fa.sp.ggx_alpha = 0.0f; fa.sp.sampled_point = 0;
    fa.src.source_offset = a->source_offset; fa.src.source_stride = 0; fa.src.total_sources = Ltot; fa.src.n_sensors = 0;
    fa.vis_words = 1;
    fa.live = c->live.as<uint16_t>(); fa.cov = c->cov.as<uint16_t>(); fa.retry = c->tile_count.as<int>();
    bool ok_launch = true;
    for (int side = 0; side < (same_set ? 1 : 2) && ok_launch; ++side) {
        fa.src.origin = side ? a->sensor : a->origin;
        fa.src.normal = side ? a->sensor_normal : a->normal;
        fa.src.L = side ? Sb : La;
        fa.rec_d = rec_d + (side ? (size_t)La * R : 0);
        fa.rec_ff = rec_ff + (side ? (size_t)La * R : 0);
        fa.rec_ext = ext ? rec_ext + (side ? (size_t)La * R : 0) : nullptr;
        fa.rec_ext_stride = (size_t)W * R;
        fa.need_tree = c->tree_complete || note.tree_built ? nullptr : c->lazy_flag.as<int>();
        ok_launch = nlos::launch_forward_record(fa, st);
        if (note.tree_built) c->tree_complete = true;
    }
    if (!ok_launch) return as_pairs();      // (the grid kernel's LDS does not hold this scene: nothing was launched)
    c->path.backend = note.backend; c->path.reason = note.reason; c->path.grid_R = note.grid_R;
    c->path.tiles = 1; c->path.chunks = 1; c->path.rows_in_lds = 1;
    c->path_retry_workgroups = note.retry_workgroups;
    // the visibility cache of the context now describes nothing a later reuse_visibility could ask for
    c->vis_key = nlos_ctx::VisKey();
    c->vis_gen = 0;

    // ---- combine: rows [La, Sb, T] (+ the accepted-sample words of every pair for pass 2)
    nlos::ProductArgs pa;
    pa.sc = sc;
    pa.d_a = rec_d; pa.ff_a = rec_ff;
    pa.d_b = same_set ? rec_d : rec_d + (size_t)La * R;
    pa.ff_b = same_set ? rec_ff : rec_ff + (size_t)La * R;
    pa.ext_a = ext ? rec_ext : nullptr;
    pa.ext_b = ext ? (same_set ? rec_ext : rec_ext + (size_t)La * R) : nullptr;
    pa.ext_stride = (size_t)W * R;
    pa.sensor_normal = a->sensor_normal;
    pa.La = La; pa.Sb = Sb; pa.spt = spt; pa.nbins = T; pa.lb = lb; pa.ub = ub; pa.res = res;
    pa.rows = a->transient;
    pa.vis = nullptr;
    if (grad) {
        rc = c->vis.ensure(sizeof(uint32_t) * (size_t)P * nF + 16);
        if (rc) return rc;
        pa.vis = c->vis.as<uint32_t>();
    }
    nlos::launch_product_combine(pa, st);
    mark(c, 2, st);

    // ---- residual + pathlengths, then pass 2 over the pairs
    nlos::ResidualArgs ra;
    std::memset(&ra, 0, sizeof(ra));
    ra.pathlengths = a->pathlengths; ra.T = T; ra.lb = lb; ra.res = res;
    if (grad) {
        rc = c->diff.ensure(sizeof(double) * (size_t)P * T + 16);
        if (rc) return rc;
        ra.data = a->data; ra.weight = a->weight; ra.transient = a->transient; ra.diff = c->diff.as<double>();
        ra.L = P; ra.loss_test = a->loss_test;
        if (a->zero_gradient) { ra.zero = a->gradient; ra.zero_n = 3 * (size_t)nV; }
    }
    if (grad || a->pathlengths) nlos::launch_residual(ra, st);
    mark(c, 3, st);
    if (grad) {
        int K = 0;
        rc = ensure_taps(c, 0, a->refine_scale, a->sigma_bin, res, st, &K);
        if (rc) return rc;
        nlos::GradientArgs ga;
        std::memset(&ga, 0, sizeof(ga));
        ga.sc = sc; ga.sp = fa.sp;
        ga.src.origin = a->origin; ga.src.normal = a->normal; ga.src.sensor = a->sensor; ga.src.sensor_normal = a->sensor_normal;
        ga.src.L = P; ga.src.n_sensors = Sb; ga.src.source_offset = a->source_offset; ga.src.source_stride = 0;
        ga.src.total_sources = Ltot;
        ga.vis = c->vis.as<uint32_t>(); ga.vis_words = 1; ga.vis_scratch = c->vis.as<uint32_t>();
        ga.tap_w = c->taps.as<double>(); ga.tap_delta = ga.tap_w + K; ga.tap_g = ga.tap_w + 2 * K;
        ga.tap_p0 = ga.tap_w + 3 * K; ga.tap_p1 = ga.tap_w + 4 * K + 1; ga.tap_pw = ga.tap_w + 5 * K + 2;
        ga.tap_wt = nullptr; ga.tap_nb = 0;
        ga.K = K;
        ga.two_rs = 2 * a->refine_scale * a->sigma_bin;
        ga.r_over_res = (double)a->refine_scale / (double)res;
        ga.refine = a->refine_scale;
        ga.tap_wt = ga.tap_w + 6 * (size_t)K + 3;
        ga.tap_nb = 4 * a->sigma_bin + 1;
        ga.diff = c->diff.as<double>();
        ga.mode = 0;
        // (nlos_render's rule: smoothed_transient/...GradientRenderer.cpp's normal term only with vertex normals and testing_flag 0)
        ga.normal_term = a->normal_term < 0 ? ((a->testing_flag == 0 && a->vertex_normal != nullptr) ? 1 : 0) : (a->normal_term ? 1 : 0);
        ga.out = a->gradient;
        ga.lds_grad = 1;
        nlos::launch_gradient(ga, st);
    }
    mark(c, 4, st);
    c->ev_valid = c->timing;
    c->path.gradient_kernel = note.gradient_kernel;
    if (note.err != hipSuccess)
        return fail(NLOS_ERR_HIP, std::string(note.err_what ? note.err_what : "launch") + ": " + hipGetErrorString(note.err));
    HIP_TRY(hipGetLastError());
    return NLOS_OK;
}

int64_t nlos_ctx_mesh_generation(const nlos_ctx* c) { return c ? c->mesh_gen : 0; }
int64_t nlos_ctx_visibility_generation(const nlos_ctx* c) { return c ? c->vis_gen : 0; }

int nlos_ctx_check(nlos_ctx* c) {
    if (!c) return fail(NLOS_ERR_ARG, "nlos_ctx_check: NULL ctx");
    DeviceGuard guard(c->device);
    HIP_TRY(hipDeviceSynchronize());
    return check_status(c, true);
}

int nlos_ctx_last_path(nlos_ctx* c, nlos_path_info* out, int count_workgroups) {
    if (!c || !out) return fail(NLOS_ERR_ARG, "nlos_ctx_last_path: NULL argument");
    *out = c->path;
    if (!count_workgroups || c->path_retry_workgroups <= 0) return NLOS_OK;
    DeviceGuard guard(c->device);
    // the flags of the (last chunk's) grid launches: single-workgroup grid -> tile_count[0 .. L), tiled grid ->
    // behind the subset sizes
    const int n = c->path_retry_workgroups;
    const int* flags = c->tile_count.as<int>() + (c->path.backend == NLOS_PATH_TILED_GRID ? (size_t)n : 0);
    std::vector<int> h((size_t)n);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(h.data(), flags, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
    {   // a synchronising entry point: report a pending bad-face-index flag here rather than in some later render
        const int rcs = check_status(c, true);
        if (rcs) return rcs;
    }
    out->workgroups = n; out->coarsened = 0; out->big_lds = 0; out->bvh_queries = 0;
    for (int v : h) {
        if (v == 1) ++out->big_lds;
        else if (v == 0x200) ++out->bvh_queries;
        else if (v >= 0x100 && v < 0x200) ++out->coarsened;
    }
    if (c->path_items_sources > 0 && c->vis_items.p && c->path.backend == NLOS_PATH_GRID) {
        // header word of every source's item masks (forward_grid.hip): [39:16] rays traced, [63:40] samples accepted
        const int L = c->path_items_sources;
        std::vector<unsigned long long> hd((size_t)L);
        HIP_TRY(hipMemcpy2D(hd.data(), sizeof(unsigned long long), c->vis_items.p, sizeof(unsigned long long) * (size_t)c->vis_items_stride,
                            sizeof(unsigned long long), (size_t)L, hipMemcpyDeviceToHost));
        out->rays_traced = 0; out->samples_accepted = 0;
        for (unsigned long long v : hd) { out->rays_traced += (int64_t)((v >> 16) & 0xffffffull); out->samples_accepted += (int64_t)(v >> 40); }
    }
    return NLOS_OK;
}

int nlos_intersect(nlos_ctx* c, const float* origins, const float* dirs, int n_rays, const float* vertices, int V,
                   const int32_t* faces, int F, float* out3, float* out1, void* stream) {
    if (!c) return fail(NLOS_ERR_ARG, "nlos_intersect: NULL ctx");
    if (F <= 0 || V <= 0 || !vertices || !faces) return fail(NLOS_ERR_ARG, "nlos_intersect: empty mesh");
    if (n_rays < 0 || (n_rays > 0 && (!origins || !dirs))) return fail(NLOS_ERR_ARG, "nlos_intersect: bad rays");
    DeviceGuard guard(c->device);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int rc = ensure_bvh(c, vertices, V, faces, F, false, 0, st);
    if (rc) return rc;
    nlos::IntersectArgs ia;
    ia.sc = scene_view(c, F, V, nullptr, nullptr);
    ia.origins = origins; ia.dirs = dirs; ia.n = n_rays; ia.out3 = out3; ia.out1 = out1;
    nlos::launch_intersect(ia, st);
    HIP_TRY(hipGetLastError());
    return NLOS_OK;
}

int64_t nlos_ctx_debug_read(nlos_ctx* c, int what, void* host_out, int64_t max_bytes) {
    if (!c || !host_out || max_bytes < 0) return -(int64_t)fail(NLOS_ERR_ARG, "nlos_ctx_debug_read: bad arguments");
    DeviceGuard guard(c->device);
    // what = 2: per-workgroup path codes of the last single-workgroup grid launch (0 normal, 1 redone with the
    // whole CU's LDS, 0x100 + R: redone on a grid coarsened to R x R)
    // what = 3: a 16-byte digest of the accepted-sample words (what = 0's array), computed on the device
    const DevBuf* b = (what == 0 || what == 3) ? &c->vis : (what == 1 ? &c->face_id : (what == 2 ? &c->tile_count : nullptr));
    if (!b || !b->p) return -(int64_t)fail(NLOS_ERR_ARG, "nlos_ctx_debug_read: nothing to read");
    size_t n = 0;
    if (what == 0 || what == 3) n = sizeof(uint32_t) * (size_t)c->vis_key.L * (size_t)((c->vis_key.spt + 31) / 32) * (size_t)c->vis_key.F;
    else if (what == 1) n = sizeof(int) * (size_t)c->built_F;
    else n = (size_t)max_bytes;
    if (what != 3 && n > (size_t)max_bytes) n = (size_t)max_bytes;
    if (n > b->cap) n = b->cap;
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess && (what == 0 || what == 3) && c->vis_is_items) {
        // the cache is held as item masks: per-face words for the reader
        nlos::launch_items_to_words(c->vis_items.as<unsigned long long>(), c->vis_items_stride, c->live.as<uint16_t>(), c->vis_key.L,
                                    c->vis_key.F, c->vis_key.spt, c->vis.as<uint32_t>(), nullptr);
        e = hipDeviceSynchronize();
    }
    if (e == hipSuccess && what == 3) {
        if (max_bytes < 16 || c->status.cap < 512) return -(int64_t)fail(NLOS_ERR_ARG, "nlos_ctx_debug_read(3): needs 16 bytes");
        unsigned long long* dg = c->status.as<unsigned long long>() + 56;      // scratch words behind the status / counters
        nlos::launch_digest_u32(c->vis.as<uint32_t>(), n / sizeof(uint32_t), dg, nullptr);
        e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipMemcpy(host_out, dg, 16, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return -(int64_t)fail(NLOS_ERR_HIP, std::string("nlos_ctx_debug_read: ") + hipGetErrorString(e));
        return 16;
    }
    if (e == hipSuccess) e = hipMemcpy(host_out, b->p, n, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return -(int64_t)fail(NLOS_ERR_HIP, std::string("nlos_ctx_debug_read: ") + hipGetErrorString(e));
    return (int64_t)n;
}

int nlos_adam_modified_step(nlos_ctx* c, float* params, const double* grad_f64, const float* grad_f32, float* exp_avg,
                            float* exp_avg_sq, float* max_exp_avg_sq, const uint8_t* row_mask, int rows, int cols,
                            int step, double lr, double beta1, double beta2, double eps, double weight_decay,
                            void* stream) {
    if (!c) return fail(NLOS_ERR_ARG, "nlos_adam_modified_step: NULL ctx");
    if (rows < 0 || cols < 1 || cols > 8) return fail(NLOS_ERR_ARG, "nlos_adam_modified_step: need rows >= 0 and 1 <= cols <= 8");
    if (rows > 0 && (!params || !exp_avg || !exp_avg_sq)) return fail(NLOS_ERR_ARG, "nlos_adam_modified_step: NULL state");
    if ((grad_f64 != nullptr) == (grad_f32 != nullptr)) return fail(NLOS_ERR_ARG, "nlos_adam_modified_step: pass exactly one of grad_f64 / grad_f32");
    if (step < 1) return fail(NLOS_ERR_ARG, "nlos_adam_modified_step: step counts from 1");
    if (!(lr >= 0.0) || !(eps >= 0.0) || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0))
        return fail(NLOS_ERR_ARG, "nlos_adam_modified_step: invalid lr / eps / betas");      // adam_modified.py:33-40
    DeviceGuard guard(c->device);
    nlos::AdamArgs a;
    a.params = params; a.grad64 = grad_f64; a.grad32 = grad_f32; a.exp_avg = exp_avg; a.exp_avg_sq = exp_avg_sq;
    a.max_exp_avg_sq = max_exp_avg_sq; a.row_mask = row_mask; a.rows = rows; a.cols = cols;
    a.beta1 = (float)beta1; a.beta2 = (float)beta2; a.eps = (float)eps; a.weight_decay = (float)weight_decay;
    a.one_minus_beta1 = (float)(1 - beta1); a.one_minus_beta2 = (float)(1 - beta2);
    // adam_modified.py:102-104, python float (double) arithmetic
    const double bc1 = 1 - std::pow(beta1, step), bc2 = 1 - std::pow(beta2, step);
    a.step_size = (float)(lr * std::sqrt(bc2) / bc1);
    nlos::launch_adam_modified(a, reinterpret_cast<hipStream_t>(stream));
    HIP_TRY(hipGetLastError());
    return NLOS_OK;
}

int nlos_create_weighting(nlos_ctx* c, const double* data, int rows, int cols, double gamma, double* weight, void* stream) {
    if (!c) return fail(NLOS_ERR_ARG, "nlos_create_weighting: NULL ctx");
    if (rows < 0 || cols < 0 || ((size_t)rows * cols > 0 && (!data || !weight))) return fail(NLOS_ERR_ARG, "nlos_create_weighting: bad arguments");
    DeviceGuard guard(c->device);
    nlos::launch_weighting(data, (size_t)rows * cols, gamma, weight, reinterpret_cast<hipStream_t>(stream));
    HIP_TRY(hipGetLastError());
    return NLOS_OK;
}

int nlos_weighted_l2(nlos_ctx* c, const double* transient, const double* data, const double* weight, int rows, int cols,
                     double* out, void* stream) {
    if (!c) return fail(NLOS_ERR_ARG, "nlos_weighted_l2: NULL ctx");
    if (rows < 0 || cols < 0 || !out || ((size_t)rows * cols > 0 && (!transient || !data))) return fail(NLOS_ERR_ARG, "nlos_weighted_l2: bad arguments");
    DeviceGuard guard(c->device);
    nlos::launch_weighted_l2(transient, data, weight, (size_t)rows * cols, rows, out, reinterpret_cast<hipStream_t>(stream));
    HIP_TRY(hipGetLastError());
    return NLOS_OK;
}

int nlos_mesh_regulariser(nlos_ctx* c, const float* vertices, int V, const int32_t* faces, int F,
                          const int32_t* face_affinity, double* gradient, double* value, int overwrite, void* stream) {
    if (!c) return fail(NLOS_ERR_ARG, "nlos_mesh_regulariser: NULL ctx");
    if (V <= 0 || F < 0 || !vertices || (F > 0 && !faces) || !gradient)
        return fail(NLOS_ERR_ARG, "nlos_mesh_regulariser: bad mesh / gradient");
    if (face_affinity && !value) return fail(NLOS_ERR_ARG, "nlos_mesh_regulariser: normal smoothing needs a value output");
    DeviceGuard guard(c->device);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int rc = c->reg_normal.ensure(sizeof(double) * 3 * (size_t)(F > 0 ? F : 1));
    if (!rc) rc = c->reg_area.ensure(sizeof(double) * (size_t)(F > 0 ? F : 1));
    if (!rc && overwrite) rc = c->reg_owner.ensure(sizeof(int) * (size_t)V);
    if (rc) return rc;
    nlos::RegulariserArgs ra;
    ra.vertices = vertices; ra.faces = faces; ra.affinity = face_affinity; ra.V = V; ra.F = F;
    ra.normal = c->reg_normal.as<double>(); ra.area = c->reg_area.as<double>();
    ra.owner = overwrite ? c->reg_owner.as<int>() : nullptr;
    ra.gradient = gradient; ra.value = value; ra.overwrite = overwrite ? 1 : 0;
    nlos::launch_regulariser(ra, st);
    HIP_TRY(hipGetLastError());
    return NLOS_OK;
}

}  // extern "C"

// =====================================================================================
// Section 1: host-pointer drop-ins.  Upload -> nlos_render -> download, synchronous.
// =====================================================================================
namespace {

struct HostCall {
    nlos_ctx* c = nullptr;
    int slot = 0;
    int rc = NLOS_OK;
    std::vector<std::pair<void*, std::pair<void*, size_t>>> downloads;  // host <- dev

    template <class T>
    T* up(const T* host, size_t count) {
        if (rc || !host || count == 0) return nullptr;
        DevBuf& b = c->io[slot++];
        rc = b.ensure(count * sizeof(T));
        if (rc) return nullptr;
        hipError_t e = hipMemcpy(b.p, host, count * sizeof(T), hipMemcpyHostToDevice);
        if (e != hipSuccess) { rc = fail(NLOS_ERR_HIP, std::string("hipMemcpy H2D: ") + hipGetErrorString(e)); return nullptr; }
        return b.as<T>();
    }
    template <class T>
    T* inout(T* host, size_t count, bool upload) {
        if (rc || !host || count == 0) return nullptr;
        DevBuf& b = c->io[slot++];
        rc = b.ensure(count * sizeof(T));
        if (rc) return nullptr;
        if (upload) {
            hipError_t e = hipMemcpy(b.p, host, count * sizeof(T), hipMemcpyHostToDevice);
            if (e != hipSuccess) { rc = fail(NLOS_ERR_HIP, std::string("hipMemcpy H2D: ") + hipGetErrorString(e)); return nullptr; }
        }
        downloads.push_back({host, {b.p, count * sizeof(T)}});
        return b.as<T>();
    }
    // ---- pipelined form (host_render, round 6) ------------------------------------------------------------------
    // Large inputs that pass 1 does not read (data, weight) get their device buffer now and their bytes LATER: after_pass1()
    // -- called by nlos_render between the passes -- copies them on the context's copy stream while pass 1 runs on the
    // render stream, and makes the render stream wait for their arrival.  The rows (transient) are final when pass 1 is,
    // so their download runs on the copy stream behind pass 2.  Small arrays keep the synchronous copies.
    struct Deferred { void* dev; const void* host; size_t bytes; };
    std::vector<Deferred> deferred;
    void* early_host = nullptr; void* early_dev = nullptr; size_t early_bytes = 0;   // the rows: downloaded behind pass 2
    bool hook_ran = false;
    template <class T>
    T* up_deferred(const T* host, size_t count) {
        if (rc || !host || count == 0) return nullptr;
        DevBuf& b = c->io[slot++];
        rc = b.ensure(count * sizeof(T));
        if (rc) return nullptr;
        deferred.push_back({b.p, host, count * sizeof(T)});
        return b.as<T>();
    }
    template <class T>
    T* out_early(T* host, size_t count) {
        if (rc || !host || count == 0) return nullptr;
        DevBuf& b = c->io[slot++];
        rc = b.ensure(count * sizeof(T));
        if (rc) return nullptr;
        early_host = host; early_dev = b.p; early_bytes = count * sizeof(T);
        return b.as<T>();
    }
    int ensure_streams() {
        if (!c->host_render) {
            HIP_TRY(hipStreamCreateWithFlags(&c->host_render, hipStreamNonBlocking));
            HIP_TRY(hipStreamCreateWithFlags(&c->host_copy, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&c->host_p1, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&c->host_up, hipEventDisableTiming));
        }
        return NLOS_OK;
    }
    int run_deferred(hipStream_t render) {
        // (pageable sources: the call returns when the bytes are staged / on the wire, the GPU keeps running pass 1 meanwhile;
        // 55 GB/s on the round-6 box, tools/host_copy_probe.hip)
        for (auto& d : deferred) HIP_TRY(hipMemcpyAsync(d.dev, d.host, d.bytes, hipMemcpyHostToDevice, c->host_copy));
        if (!deferred.empty()) {
            HIP_TRY(hipEventRecord(c->host_up, c->host_copy));
            HIP_TRY(hipStreamWaitEvent(render, c->host_up, 0));
        }
        deferred.clear();
        return NLOS_OK;
    }
    static int after_pass1_thunk(void* self, hipStream_t st) {
        HostCall* h = static_cast<HostCall*>(self);
        h->hook_ran = true;
        HIP_TRY(hipEventRecord(h->c->host_p1, st));
        return h->run_deferred(st);
    }
    // after nlos_render has returned (everything is enqueued): the rows come down behind pass 2, the small outputs behind the
    // render stream; no device-wide synchronisation
    int finish_pipelined(hipStream_t render) {
        if (rc) return rc;
        hipError_t e;
        if (early_host) {
            if (hook_ran) e = hipStreamWaitEvent(c->host_copy, c->host_p1, 0);
            else e = hipStreamSynchronize(render);          // (a path that did not pass the hook: after everything)
            if (e == hipSuccess) e = hipMemcpyAsync(early_host, early_dev, early_bytes, hipMemcpyDeviceToHost, c->host_copy);
            if (e != hipSuccess) return fail(NLOS_ERR_HIP, std::string("download of the rows: ") + hipGetErrorString(e));
        }
        e = hipStreamSynchronize(render);
        if (e != hipSuccess) return fail(NLOS_ERR_HIP, std::string("hipStreamSynchronize(render): ") + hipGetErrorString(e));
        for (auto& d : downloads) {
            e = hipMemcpyAsync(d.first, d.second.first, d.second.second, hipMemcpyDeviceToHost, render);
            if (e != hipSuccess) return fail(NLOS_ERR_HIP, std::string("hipMemcpy D2H: ") + hipGetErrorString(e));
        }
        int st[4] = {0, 0, 0, 0};
        e = hipMemcpyAsync(st, c->status.p, sizeof(st), hipMemcpyDeviceToHost, render);
        if (e == hipSuccess) e = hipStreamSynchronize(render);
        hipError_t e2 = hipStreamSynchronize(c->host_copy);
        if (e2 != hipSuccess) return fail(NLOS_ERR_HIP, std::string("hipStreamSynchronize(copy): ") + hipGetErrorString(e2));
        c->status_pending = false;
        if (e == hipSuccess && (st[0] & 1)) {
            c->status_clear = true;
            return fail(NLOS_ERR_ARG, "face index out of range [0, numVertices)");
        }
        return NLOS_OK;
    }
    int finish() {
        if (rc) return rc;
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) return fail(NLOS_ERR_HIP, std::string("hipDeviceSynchronize: ") + hipGetErrorString(e));
        for (auto& d : downloads) {
            e = hipMemcpy(d.first, d.second.first, d.second.second, hipMemcpyDeviceToHost);
            if (e != hipSuccess) return fail(NLOS_ERR_HIP, std::string("hipMemcpy D2H: ") + hipGetErrorString(e));
        }
        int st[4] = {0, 0, 0, 0};
        e = hipMemcpy(st, c->status.p, sizeof(st), hipMemcpyDeviceToHost);
        c->status_pending = false;
        if (e == hipSuccess && (st[0] & 1)) {
            c->status_clear = true;
            return fail(NLOS_ERR_ARG, "face index out of range [0, numVertices)");
        }
        return NLOS_OK;
    }
};

struct HostRender {
    int mode = NLOS_MODE_TRANSIENT;
    double *data = nullptr, *weight = nullptr;
    float *origin = nullptr, *normal = nullptr, *vertices = nullptr, *vnormal = nullptr, *albedo = nullptr;
    float *sensor = nullptr, *sensor_normal = nullptr;
    double *jitter_weight = nullptr, *jitter_grad = nullptr;
    int jitter_offset = 0, jitter_length = 0;
    int* faces = nullptr;
    int L = 0, V = 0, F = 0, num_samples = 0;
    float lb = 0, ub = 0, res = 1;
    double *transient = nullptr, *pathlengths = nullptr, *gradient = nullptr, *intensity = nullptr, *scalar = nullptr;
    int refine = 1, sigma_bin = 1, testing_flag = 0, loss_test = 0, use_ggx = 0, clamp = 1, w_width = 0, vertex_num = -1;
    int sampled_point = 0;
    float alpha = 0;
    int n_sensors = 0;          // > 0: the L x S product (sensor arrays of n_sensors rows, rows [L, n_sensors, T])
};

int host_render(const HostRender& h) {
    if (h.F <= 0 || h.V <= 0 || !h.vertices || !h.faces) return fail(NLOS_ERR_ARG, "empty mesh");
    if (h.L < 0) return fail(NLOS_ERR_ARG, "negative source count");
    for (size_t i = 0; i < 3 * (size_t)h.F; ++i)
        if (h.faces[i] < 0 || h.faces[i] >= h.V) return fail(NLOS_ERR_ARG, "face index out of range [0, numVertices)");
    nlos_ctx* c = nullptr;
    int rc = get_default_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_mu);
    DeviceGuard guard(c->device);
    HostCall hc;
    hc.c = c;
    rc = hc.ensure_streams();
    if (rc) return rc;
    // data / weight are first read after pass 1 (residual / pass 2): their upload rides behind it (HostCall, pipelined form);
    // the product entry consumes them inside its own sequence of launches: uploaded up front there
    const bool defer = h.n_sensors == 0;
    const int T = h.mode == NLOS_MODE_INTENSITY ? 0 : nlos_num_bins(h.lb, h.ub, h.res);
    nlos_render_args a;
    nlos_render_args_init(&a);
    a.mode = h.mode;
    a.origin = hc.up(h.origin, 3 * (size_t)h.L);
    a.normal = hc.up(h.normal, 3 * (size_t)h.L);
    const size_t n_sens = h.n_sensors > 0 ? (size_t)h.n_sensors : (size_t)h.L;      // sensor points
    const size_t n_meas = h.n_sensors > 0 ? (size_t)h.L * (size_t)h.n_sensors : (size_t)h.L;   // rows of transient / data / weight
    a.n_sensors = h.n_sensors;
    a.sensor = hc.up(h.sensor, 3 * n_sens);
    a.sensor_normal = hc.up(h.sensor_normal, 3 * n_sens);
    a.jitter_weight = hc.up(h.jitter_weight, (size_t)h.jitter_length);
    a.jitter_grad = hc.up(h.jitter_grad, (size_t)h.jitter_length);
    a.jitter_offset = h.jitter_offset; a.jitter_length = h.jitter_length;
    a.L = h.L; a.source_offset = 0; a.total_sources = h.L;
    a.vertices = hc.up(h.vertices, 3 * (size_t)h.V); a.V = h.V;
    a.faces = hc.up(h.faces, 3 * (size_t)h.F); a.F = h.F;
    a.vertex_normal = hc.up(h.vnormal, 3 * (size_t)h.V);
    a.albedo = hc.up(h.albedo, (size_t)h.V);
    a.num_samples = h.num_samples;
    a.lower_bound = h.lb; a.upper_bound = h.ub; a.resolution = h.res;
    a.refine_scale = h.refine; a.sigma_bin = h.sigma_bin;
    a.seed = g_default_seed;
    a.data = defer ? hc.up_deferred(h.data, n_meas * T) : hc.up(h.data, n_meas * T);
    a.weight = defer ? hc.up_deferred(h.weight, n_meas * T) : hc.up(h.weight, n_meas * T);
    a.transient = hc.out_early(h.transient, n_meas * T);
    a.pathlengths = hc.inout(h.pathlengths, (size_t)T, false);
    if (h.mode == NLOS_MODE_VERTEX_GRADIENT) a.gradient = hc.inout(h.gradient, 3 * (size_t)T, true);
    else a.gradient = hc.inout(h.gradient, 3 * (size_t)h.V, h.mode != NLOS_MODE_GRADIENT_V1);
    a.intensity = hc.inout(h.intensity, (size_t)h.F, true);
    a.scalar_out = hc.inout(h.scalar, 1, false);
    a.testing_flag = h.testing_flag; a.loss_test = h.loss_test;
    a.normal_term = -1; a.clamp = h.clamp; a.use_ggx = h.use_ggx; a.ggx_alpha = h.alpha;
    a.vertex_num = h.vertex_num; a.w_width = h.w_width;
    a.v1_sampled_point = h.sampled_point;
    if (hc.rc) return hc.rc;
    if (h.L == 0) {
        // nothing to render: outputs keep the reference's semantics (transient is [0,T])
        if (h.pathlengths) for (int i = 0; i < T; ++i) h.pathlengths[i] = (double)(h.lb + i * h.res);
        if (h.scalar) *h.scalar = 0.0;
        return NLOS_OK;
    }
    AfterPass1 hook{&HostCall::after_pass1_thunk, &hc};
    tl_after_pass1 = defer ? &hook : nullptr;
    rc = nlos_render(c, &a, c->host_render);
    tl_after_pass1 = nullptr;
    if (rc) {
        // nothing may still be reading the staging buffers or writing host memory when the caller sees the error
        hipError_t e1 = hipStreamSynchronize(c->host_render), e2 = hipStreamSynchronize(c->host_copy);
        (void)e1; (void)e2;
        return rc;
    }
    if (!hc.hook_ran && !hc.deferred.empty()) {
        // cannot happen on the paths above (every non-product render passes the hook); never leave inputs behind silently
        return fail(NLOS_ERR_ARG, "host drop-in: deferred inputs were not consumed");
    }
    return hc.finish_pipelined(c->host_render);
}

}  // namespace

extern "C" {

int nlos_streamed_render_transient(float* origin, int numSources, float* normal, float* vertices, int numVertices,
                                   float* vertexNormal, float* vertexAlbedo, int* triangles, int numTriangles,
                                   int numSamples, float lowerBound, float upperBound, float resolution,
                                   double* transient, double* pathlengths, int refine_scale, int sigma_bin) {
    HostRender h;
    h.mode = NLOS_MODE_TRANSIENT;
    h.origin = origin; h.L = numSources; h.normal = normal; h.vertices = vertices; h.V = numVertices;
    h.vnormal = vertexNormal; h.albedo = vertexAlbedo; h.faces = triangles; h.F = numTriangles;
    h.num_samples = numSamples; h.lb = lowerBound; h.ub = upperBound; h.res = resolution;
    h.transient = transient; h.pathlengths = pathlengths; h.refine = refine_scale; h.sigma_bin = sigma_bin;
    return host_render(h);
}

int nlos_nonconfocal_render_transient(float* laser, float* laserNormal, float* sensor, float* sensorNormal,
                                      int numPairs, float* vertices, int numVertices, float* vertexNormal,
                                      float* vertexAlbedo, int* triangles, int numTriangles, int numSamples,
                                      float lowerBound, float upperBound, float resolution, double* transient,
                                      double* pathlengths, int refine_scale, int sigma_bin) {
    if (numPairs > 0 && (!sensor || !sensorNormal)) return fail(NLOS_ERR_ARG, "non-confocal render: sensor arrays are NULL");
    HostRender h;
    h.mode = NLOS_MODE_TRANSIENT;
    h.origin = laser; h.normal = laserNormal; h.sensor = sensor; h.sensor_normal = sensorNormal; h.L = numPairs;
    h.vertices = vertices; h.V = numVertices;
    h.vnormal = vertexNormal; h.albedo = vertexAlbedo; h.faces = triangles; h.F = numTriangles;
    h.num_samples = numSamples; h.lb = lowerBound; h.ub = upperBound; h.res = resolution;
    h.transient = transient; h.pathlengths = pathlengths; h.refine = refine_scale; h.sigma_bin = sigma_bin;
    return host_render(h);
}

int nlos_nonconfocal_render_gradient(double* data, double* weight, float* laser, float* laserNormal, float* sensor,
                                     float* sensorNormal, int numPairs, float* vertices, int numVertices,
                                     float* vertexNormal, float* vertexAlbedo, int* triangles, int numTriangles,
                                     int numSamples, float lowerBound, float upperBound, float resolution,
                                     double* transient, double* pathlengths, double* gradient, int refine_scale,
                                     int sigma_bin, int testing_flag, int loss_test) {
    if (numPairs > 0 && (!sensor || !sensorNormal)) return fail(NLOS_ERR_ARG, "non-confocal render: sensor arrays are NULL");
    HostRender h;
    h.mode = NLOS_MODE_GRADIENT;
    h.data = data; h.weight = weight;
    h.origin = laser; h.normal = laserNormal; h.sensor = sensor; h.sensor_normal = sensorNormal; h.L = numPairs;
    h.vertices = vertices; h.V = numVertices;
    h.vnormal = vertexNormal; h.albedo = vertexAlbedo; h.faces = triangles; h.F = numTriangles;
    h.num_samples = numSamples; h.lb = lowerBound; h.ub = upperBound; h.res = resolution;
    h.transient = transient; h.pathlengths = pathlengths; h.gradient = gradient;
    h.refine = refine_scale; h.sigma_bin = sigma_bin; h.testing_flag = testing_flag; h.loss_test = loss_test;
    return host_render(h);
}

int nlos_nonconfocal_product_render_transient(float* laser, float* laserNormal, int numLasers, float* sensor,
                                              float* sensorNormal, int numSensors, float* vertices, int numVertices,
                                              int* triangles, int numTriangles, int numSamples, float lowerBound,
                                              float upperBound, float resolution, double* transient, double* pathlengths) {
    if (numSensors <= 0 || !sensor || !sensorNormal) return fail(NLOS_ERR_ARG, "non-confocal product: sensor arrays are NULL or empty");
    HostRender h;
    h.mode = NLOS_MODE_TRANSIENT;
    h.origin = laser; h.normal = laserNormal; h.sensor = sensor; h.sensor_normal = sensorNormal; h.L = numLasers;
    h.n_sensors = numSensors;
    h.vertices = vertices; h.V = numVertices; h.faces = triangles; h.F = numTriangles;
    h.num_samples = numSamples; h.lb = lowerBound; h.ub = upperBound; h.res = resolution;
    h.transient = transient; h.pathlengths = pathlengths;
    return host_render(h);
}

int nlos_nonconfocal_product_render_gradient(double* data, double* weight, float* laser, float* laserNormal, int numLasers,
                                             float* sensor, float* sensorNormal, int numSensors, float* vertices,
                                             int numVertices, int* triangles, int numTriangles, int numSamples,
                                             float lowerBound, float upperBound, float resolution, double* transient,
                                             double* pathlengths, double* gradient, int refine_scale, int sigma_bin,
                                             int testing_flag, int loss_test) {
    if (numSensors <= 0 || !sensor || !sensorNormal) return fail(NLOS_ERR_ARG, "non-confocal product: sensor arrays are NULL or empty");
    HostRender h;
    h.mode = NLOS_MODE_GRADIENT;
    h.data = data; h.weight = weight;
    h.origin = laser; h.normal = laserNormal; h.sensor = sensor; h.sensor_normal = sensorNormal; h.L = numLasers;
    h.n_sensors = numSensors;
    h.vertices = vertices; h.V = numVertices; h.faces = triangles; h.F = numTriangles;
    h.num_samples = numSamples; h.lb = lowerBound; h.ub = upperBound; h.res = resolution;
    h.transient = transient; h.pathlengths = pathlengths; h.gradient = gradient;
    h.refine = refine_scale; h.sigma_bin = sigma_bin; h.testing_flag = testing_flag; h.loss_test = loss_test;
    return host_render(h);
}

int nlos_ggx_nonconfocal_render_transient(float* laser, float* laserNormal, float* sensor, float* sensorNormal,
                                          int numPairs, float* vertices, int numVertices, float* vertexNormal,
                                          float* vertexAlbedo, int* triangles, int numTriangles, float alpha,
                                          int numSamples, float lowerBound, float upperBound, float resolution,
                                          double* transient, double* pathlengths, int refine_scale, int sigma_bin) {
    if (numPairs > 0 && (!sensor || !sensorNormal)) return fail(NLOS_ERR_ARG, "non-confocal render: sensor arrays are NULL");
    HostRender h;
    h.mode = NLOS_MODE_TRANSIENT; h.use_ggx = 1; h.alpha = alpha;
    h.origin = laser; h.normal = laserNormal; h.sensor = sensor; h.sensor_normal = sensorNormal; h.L = numPairs;
    h.vertices = vertices; h.V = numVertices;
    h.vnormal = vertexNormal; h.albedo = vertexAlbedo; h.faces = triangles; h.F = numTriangles;
    h.num_samples = numSamples; h.lb = lowerBound; h.ub = upperBound; h.res = resolution;
    h.transient = transient; h.pathlengths = pathlengths; h.refine = refine_scale; h.sigma_bin = sigma_bin;
    return host_render(h);
}

int nlos_ggx_nonconfocal_render_gradient(double* data, double* weight, float* laser, float* laserNormal, float* sensor,
                                         float* sensorNormal, int numPairs, float* vertices, int numVertices,
                                         float* vertexNormal, float* vertexAlbedo, int* triangles, int numTriangles,
                                         float alpha, int numSamples, float lowerBound, float upperBound,
                                         float resolution, double* transient, double* pathlengths, double* gradient,
                                         int refine_scale, int sigma_bin, int testing_flag, int loss_test) {
    if (numPairs > 0 && (!sensor || !sensorNormal)) return fail(NLOS_ERR_ARG, "non-confocal render: sensor arrays are NULL");
    HostRender h;
    h.mode = NLOS_MODE_GRADIENT; h.use_ggx = 1; h.alpha = alpha;
    h.data = data; h.weight = weight;
    h.origin = laser; h.normal = laserNormal; h.sensor = sensor; h.sensor_normal = sensorNormal; h.L = numPairs;
    h.vertices = vertices; h.V = numVertices;
    h.vnormal = vertexNormal; h.albedo = vertexAlbedo; h.faces = triangles; h.F = numTriangles;
    h.num_samples = numSamples; h.lb = lowerBound; h.ub = upperBound; h.res = resolution;
    h.transient = transient; h.pathlengths = pathlengths; h.gradient = gradient;
    h.refine = refine_scale; h.sigma_bin = sigma_bin; h.testing_flag = testing_flag; h.loss_test = loss_test;
    return host_render(h);
}

int nlos_jitter_streamed_render_transient(float* origin, int numSources, float* normal, float* vertices,
                                          int numVertices, float* vertexNormal, float* vertexAlbedo, int* triangles,
                                          int numTriangles, int numSamples, float lowerBound, float upperBound,
                                          float resolution, double* weight, int weight_offset, int weight_length,
                                          double* transient, double* pathlengths) {
    if (!weight || weight_length < 1) return fail(NLOS_ERR_ARG, "jitter render: weight is NULL / empty");
    HostRender h;
    h.mode = NLOS_MODE_TRANSIENT;
    h.origin = origin; h.L = numSources; h.normal = normal; h.vertices = vertices; h.V = numVertices;
    h.vnormal = vertexNormal; h.albedo = vertexAlbedo; h.faces = triangles; h.F = numTriangles;
    h.num_samples = numSamples; h.lb = lowerBound; h.ub = upperBound; h.res = resolution;
    h.jitter_weight = weight; h.jitter_offset = weight_offset; h.jitter_length = weight_length;
    h.transient = transient; h.pathlengths = pathlengths;
    return host_render(h);
}

int nlos_jitter_streamed_render_gradient(double* data, double* weight, float* origin, int measurement, float* normal,
                                         float* vertices, int numVertices, float* vertexNormal, int* triangles,
                                         int numTriangles, int numSamples, float lowerBound, float upperBound,
                                         float resolution, double* jitter_weight, double* jitter_grad,
                                         int weight_offset, int weight_length, double* transient,
                                         double* pathlengths, double* gradient, int testing_flag) {
    if (!jitter_weight || !jitter_grad || weight_length < 1) return fail(NLOS_ERR_ARG, "jitter render: jitter arrays are NULL / empty");
    HostRender h;
    h.mode = NLOS_MODE_GRADIENT;
    h.data = data; h.weight = weight;
    h.origin = origin; h.L = measurement; h.normal = normal; h.vertices = vertices; h.V = numVertices;
    h.vnormal = vertexNormal; h.faces = triangles; h.F = numTriangles;
    h.num_samples = numSamples; h.lb = lowerBound; h.ub = upperBound; h.res = resolution;
    h.jitter_weight = jitter_weight; h.jitter_grad = jitter_grad; h.jitter_offset = weight_offset;
    h.jitter_length = weight_length;
    h.transient = transient; h.pathlengths = pathlengths; h.gradient = gradient;
    h.testing_flag = testing_flag;
    return host_render(h);
}

int nlos_streamed_render_intensity(float* origin, int numSources, float* normal, float* vertices, int numVertices,
                                   float* vertexNormal, int* triangles, int numTriangles, int numSamples,
                                   float lowerBound, float upperBound, double* intensity) {
    HostRender h;
    h.mode = NLOS_MODE_INTENSITY;
    h.origin = origin; h.L = numSources; h.normal = normal; h.vertices = vertices; h.V = numVertices;
    h.vnormal = vertexNormal; h.faces = triangles; h.F = numTriangles;
    h.num_samples = numSamples; h.lb = lowerBound; h.ub = upperBound; h.res = 1.0f;
    h.intensity = intensity;
    return host_render(h);
}

int nlos_streamed_render_gradient(double* data, double* weight, float* origin, int measurement, float* normal,
                                  float* vertices, int numVertices, float* vertexNormal, int* triangles,
                                  int numTriangles, int numSamples, float lowerBound, float upperBound,
                                  float resolution, double* transient, double* pathlengths, double* gradient,
                                  int refine_scale, int sigma_bin, int testing_flag, int loss_test) {
    HostRender h;
    h.mode = NLOS_MODE_GRADIENT;
    h.data = data; h.weight = weight;
    h.origin = origin; h.L = measurement; h.normal = normal; h.vertices = vertices; h.V = numVertices;
    h.vnormal = vertexNormal; h.faces = triangles; h.F = numTriangles;
    h.num_samples = numSamples; h.lb = lowerBound; h.ub = upperBound; h.res = resolution;
    h.transient = transient; h.pathlengths = pathlengths; h.gradient = gradient;
    h.refine = refine_scale; h.sigma_bin = sigma_bin; h.testing_flag = testing_flag; h.loss_test = loss_test;
    return host_render(h);
}

int nlos_streamed_render_gradient_w_albedo(double* data, double* weight, float* origin, int measurement,
                                           float* normal, float* vertices, int numVertices, float* albedo,
                                           int* triangles, int numTriangles, int numSamples, float lowerBound,
                                           float upperBound, float resolution, double* transient,
                                           double* pathlengths, double* gradient, int refine_scale, int sigma_bin,
                                           int testing_flag, int loss_test) {
    HostRender h;
    h.mode = NLOS_MODE_GRADIENT;
    h.data = data; h.weight = weight;
    h.origin = origin; h.L = measurement; h.normal = normal; h.vertices = vertices; h.V = numVertices;
    h.albedo = albedo; h.faces = triangles; h.F = numTriangles;
    h.num_samples = numSamples; h.lb = lowerBound; h.ub = upperBound; h.res = resolution;
    h.transient = transient; h.pathlengths = pathlengths; h.gradient = gradient;
    h.refine = refine_scale; h.sigma_bin = sigma_bin; h.testing_flag = testing_flag; h.loss_test = loss_test;
    return host_render(h);
}

int nlos_streamed_render_gradient_albedo(double* data, double* weight, float* origin, int measurement, float* normal,
                                         float* vertices, int numVertices, float* albedo, int* triangles,
                                         int numTriangles, int numSamples, float lowerBound, float upperBound,
                                         float resolution, double* transient, double* pathlengths, int refine_scale,
                                         int sigma_bin, int testing_flag, int loss_test, double* grad_out) {
    HostRender h;
    h.mode = NLOS_MODE_GRAD_ALBEDO;
    h.data = data; h.weight = weight;
    h.origin = origin; h.L = measurement; h.normal = normal; h.vertices = vertices; h.V = numVertices;
    h.albedo = albedo; h.faces = triangles; h.F = numTriangles;
    h.num_samples = numSamples; h.lb = lowerBound; h.ub = upperBound; h.res = resolution;
    h.transient = transient; h.pathlengths = pathlengths; h.scalar = grad_out;
    h.refine = refine_scale; h.sigma_bin = sigma_bin; h.testing_flag = testing_flag; h.loss_test = loss_test;
    return host_render(h);
}

int nlos_streamed_render_vertex_gradient(int vertex_num, float* origin, int measurement, float* normal,
                                         float* vertices, int numVertices, int* triangles, int numTriangles,
                                         int numSamples, float lowerBound, float upperBound, float resolution,
                                         double* gradient, int refine_scale, int sigma_bin) {
    HostRender h;
    h.mode = NLOS_MODE_VERTEX_GRADIENT;
    h.origin = origin; h.L = measurement; h.normal = normal; h.vertices = vertices; h.V = numVertices;
    h.faces = triangles; h.F = numTriangles;
    h.num_samples = numSamples; h.lb = lowerBound; h.ub = upperBound; h.res = resolution;
    h.gradient = gradient; h.refine = refine_scale; h.sigma_bin = sigma_bin; h.vertex_num = vertex_num;
    return host_render(h);
}

int nlos_ggx_streamed_render_transient(float* origin, int numSources, float* normal, float* vertices,
                                       int numVertices, float* vertexNormal, float* vertexAlbedo, int* triangles,
                                       int numTriangles, float alpha, int numSamples, float lowerBound,
                                       float upperBound, float resolution, double* transient, double* pathlengths,
                                       int refine_scale, int sigma_bin) {
    HostRender h;
    h.mode = NLOS_MODE_TRANSIENT; h.use_ggx = 1; h.alpha = alpha;
    h.origin = origin; h.L = numSources; h.normal = normal; h.vertices = vertices; h.V = numVertices;
    h.vnormal = vertexNormal; h.albedo = vertexAlbedo; h.faces = triangles; h.F = numTriangles;
    h.num_samples = numSamples; h.lb = lowerBound; h.ub = upperBound; h.res = resolution;
    h.transient = transient; h.pathlengths = pathlengths; h.refine = refine_scale; h.sigma_bin = sigma_bin;
    return host_render(h);
}

int nlos_ggx_streamed_render_intensity(float* origin, int numSources, float* normal, float* vertices,
                                       int numVertices, float* vertexNormal, int* triangles, int numTriangles,
                                       float alpha, int numSamples, float lowerBound, float upperBound,
                                       double* intensity) {
    HostRender h;
    h.mode = NLOS_MODE_INTENSITY; h.use_ggx = 1; h.alpha = alpha;
    h.origin = origin; h.L = numSources; h.normal = normal; h.vertices = vertices; h.V = numVertices;
    h.vnormal = vertexNormal; h.faces = triangles; h.F = numTriangles;
    h.num_samples = numSamples; h.lb = lowerBound; h.ub = upperBound; h.res = 1.0f;
    h.intensity = intensity;
    return host_render(h);
}

int nlos_ggx_streamed_render_gradient(double* data, double* weight, float* origin, int measurement, float* normal,
                                      float* vertices, int numVertices, float* vertexNormal, int* triangles,
                                      int numTriangles, float alpha, int numSamples, float lowerBound,
                                      float upperBound, float resolution, double* transient, double* pathlengths,
                                      double* gradient, int refine_scale, int sigma_bin, int testing_flag) {
    HostRender h;
    h.mode = NLOS_MODE_GRADIENT; h.use_ggx = 1; h.alpha = alpha;
    h.data = data; h.weight = weight;
    h.origin = origin; h.L = measurement; h.normal = normal; h.vertices = vertices; h.V = numVertices;
    h.vnormal = vertexNormal; h.faces = triangles; h.F = numTriangles;
    h.num_samples = numSamples; h.lb = lowerBound; h.ub = upperBound; h.res = resolution;
    h.transient = transient; h.pathlengths = pathlengths; h.gradient = gradient;
    h.refine = refine_scale; h.sigma_bin = sigma_bin; h.testing_flag = testing_flag; h.loss_test = 0;
    return host_render(h);
}

int nlos_ggx_streamed_render_gradient_alpha(double* data, double* weight, float* origin, int measurement,
                                            float* normal, float* vertices, int numVertices, float* vertexNormal,
                                            int* triangles, int numTriangles, float alpha, int numSamples,
                                            float lowerBound, float upperBound, float resolution, double* transient,
                                            double* pathlengths, int refine_scale, int sigma_bin, double* grad_out) {
    HostRender h;
    h.mode = NLOS_MODE_GRAD_ALPHA; h.use_ggx = 1; h.alpha = alpha;
    h.data = data; h.weight = weight;
    h.origin = origin; h.L = measurement; h.normal = normal; h.vertices = vertices; h.V = numVertices;
    h.vnormal = vertexNormal; h.faces = triangles; h.F = numTriangles;
    h.num_samples = numSamples; h.lb = lowerBound; h.ub = upperBound; h.res = resolution;
    h.transient = transient; h.pathlengths = pathlengths; h.scalar = grad_out;
    h.refine = refine_scale; h.sigma_bin = sigma_bin;
    return host_render(h);
}

int nlos_v1_streamed_render_gradient(double* data, float* origin, int measurement, float* normal, float* vertices,
                                     int numVertices, int* triangles, int numTriangles, int numSamples,
                                     float lowerBound, float upperBound, float resolution, int w_width,
                                     double* transient, double* pathlengths, double* gradient) {
    HostRender h;
    h.mode = NLOS_MODE_GRADIENT_V1;
    h.data = data;
    h.origin = origin; h.L = measurement; h.normal = normal; h.vertices = vertices; h.V = numVertices;
    h.faces = triangles; h.F = numTriangles;
    h.num_samples = numSamples; h.lb = lowerBound; h.ub = upperBound; h.res = resolution;
    h.transient = transient; h.pathlengths = pathlengths; h.gradient = gradient; h.w_width = w_width;
    return host_render(h);
}

int nlos_v1_streamed_render_transient(float* origin, int numSources, float* normal, float* vertices, int numVertices,
                                      float* vertexNormal, float* vertexAlbedo, int* triangles, int numTriangles,
                                      int numSamples, float lowerBound, float upperBound, float resolution,
                                      double* transient, double* pathlengths) {
    HostRender h;
    h.mode = NLOS_MODE_TRANSIENT; h.clamp = 0;
    h.origin = origin; h.L = numSources; h.normal = normal; h.vertices = vertices; h.V = numVertices;
    h.vnormal = vertexNormal; h.albedo = vertexAlbedo; h.faces = triangles; h.F = numTriangles;
    h.num_samples = numSamples; h.lb = lowerBound; h.ub = upperBound; h.res = resolution;
    h.transient = transient; h.pathlengths = pathlengths;
    return host_render(h);
}

int nlos_v1_render_transient(float* origin, float* normal, float* vertices, int numVertices, int* triangles,
                             int numTriangles, int numSamples, float lowerBound, float upperBound, float resolution,
                             double* transient, double* pathlengths) {
    // stratified_transient_raytracer/stratifiedTransientRenderer.cpp:132-218: one source, rows [numBins]; the body
    // (:96-124) bins with the SAMPLED point's distance, unclamped
    HostRender h;
    h.mode = NLOS_MODE_TRANSIENT; h.clamp = 0; h.sampled_point = 1;
    h.origin = origin; h.L = 1; h.normal = normal; h.vertices = vertices; h.V = numVertices;
    h.faces = triangles; h.F = numTriangles;
    h.num_samples = numSamples; h.lb = lowerBound; h.ub = upperBound; h.res = resolution;
    h.transient = transient; h.pathlengths = pathlengths;
    return host_render(h);
}

static int host_intersect(float* origins, float* directions, int num_ray, float* vertices, int num_vertices,
                          int* triangles, int num_triangles, float* out, bool shortform) {
    if (num_triangles <= 0 || num_vertices <= 0 || !vertices || !triangles) return fail(NLOS_ERR_ARG, "empty mesh");
    if (num_ray < 0) return fail(NLOS_ERR_ARG, "negative ray count");
    if (num_ray == 0) return NLOS_OK;
    if (!origins || !directions || !out) return fail(NLOS_ERR_ARG, "NULL ray buffers");
    for (size_t i = 0; i < 3 * (size_t)num_triangles; ++i)
        if (triangles[i] < 0 || triangles[i] >= num_vertices) return fail(NLOS_ERR_ARG, "face index out of range [0, numVertices)");
    nlos_ctx* c = nullptr;
    int rc = get_default_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_mu);
    DeviceGuard guard(c->device);
    HostCall hc;
    hc.c = c;
    const float* o = hc.up(origins, 3 * (size_t)num_ray);
    const float* d = hc.up(directions, 3 * (size_t)num_ray);
    const float* v = hc.up(vertices, 3 * (size_t)num_vertices);
    const int* f = hc.up(triangles, 3 * (size_t)num_triangles);
    // [N,3] output keeps the caller's u,v on a miss (c_embree_intersector.cpp:39-45): upload it first
    float* r = hc.inout(out, shortform ? (size_t)num_ray : 3 * (size_t)num_ray, !shortform);
    if (hc.rc) return hc.rc;
    rc = nlos_intersect(c, o, d, num_ray, v, num_vertices, f, num_triangles, shortform ? nullptr : r,
                        shortform ? r : nullptr, nullptr);
    if (rc) return rc;
    return hc.finish();
}

int nlos_embree3_tbb_line_intersection(float* origins, float* directions, int num_ray, float* vertices,
                                       int num_vertices, int* triangles, int num_triangles, float* intersect) {
    return host_intersect(origins, directions, num_ray, vertices, num_vertices, triangles, num_triangles, intersect, false);
}

int nlos_embree3_tbb_short_line_intersection(float* origins, float* directions, int num_ray, float* vertices,
                                             int num_vertices, int* triangles, int num_triangles, float* intersect) {
    return host_intersect(origins, directions, num_ray, vertices, num_vertices, triangles, num_triangles, intersect, true);
}

int nlos_barycentric_to_world_n(float* vertices, int num_vertices, int* triangles, int num_triangles,
                                float* barycoord, int num_ray, float* intersection_p) {
    if (num_ray < 0 || num_vertices <= 0 || num_triangles <= 0 || !vertices || !triangles ||
        (num_ray > 0 && (!barycoord || !intersection_p)))
        return fail(NLOS_ERR_ARG, "nlos_barycentric_to_world: bad arguments");
    if (num_ray == 0) return NLOS_OK;
    nlos_ctx* c = nullptr;
    int rc = get_default_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_mu);
    DeviceGuard guard(c->device);
    HostCall hc;
    hc.c = c;
    const float* v = hc.up(vertices, 3 * (size_t)num_vertices);
    const int* f = hc.up(triangles, 3 * (size_t)num_triangles);
    const float* b = hc.up(barycoord, 3 * (size_t)num_ray);
    // rows whose face id is negative keep the caller's values (c_embree_intersector.cpp:79-80)
    float* out = hc.inout(intersection_p, 3 * (size_t)num_ray, true);
    if (hc.rc) return hc.rc;
    nlos::launch_bary_to_world(v, f, b, num_ray, out, nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NLOS_ERR_HIP, std::string("bary_to_world launch: ") + hipGetErrorString(e));
    hipError_t e2 = hipDeviceSynchronize();
    if (e2 != hipSuccess) return fail(NLOS_ERR_HIP, std::string("hipDeviceSynchronize: ") + hipGetErrorString(e2));
    for (auto& d : hc.downloads) {
        e = hipMemcpy(d.first, d.second.first, d.second.second, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return fail(NLOS_ERR_HIP, std::string("hipMemcpy D2H: ") + hipGetErrorString(e));
    }
    return NLOS_OK;
}

namespace {
int host_regulariser(float* vertices, int numVertices, int* triangles, int numTriangles, int* face_affinity,
                     double* curvature_grad, double* value_out) {
    if (numVertices <= 0 || numTriangles < 0 || !vertices || (numTriangles > 0 && !triangles) || !curvature_grad)
        return fail(NLOS_ERR_ARG, "mesh regulariser: bad arguments");
    for (size_t i = 0; i < 3 * (size_t)numTriangles; ++i) {
        if (triangles[i] < 0 || triangles[i] >= numVertices) return fail(NLOS_ERR_ARG, "face index out of range [0, numVertices)");
        if (face_affinity && face_affinity[i] >= numTriangles) return fail(NLOS_ERR_ARG, "face_affinity entry out of range");
    }
    nlos_ctx* c = nullptr;
    int rc = get_default_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_mu);
    DeviceGuard guard(c->device);
    HostCall hc;
    hc.c = c;
    const float* v = hc.up(vertices, 3 * (size_t)numVertices);
    const int* f = hc.up(triangles, 3 * (size_t)numTriangles);
    const int* aff = hc.up(face_affinity, 3 * (size_t)numTriangles);
    double* g = hc.inout(curvature_grad, 3 * (size_t)numVertices, false);
    double dummy = 0.0;
    double* val = hc.inout(value_out ? value_out : &dummy, 1, false);
    if (hc.rc) return hc.rc;
    if (face_affinity && numTriangles > 0 && !aff) return fail(NLOS_ERR_HIP, "mesh regulariser: staging failed");
    rc = nlos_mesh_regulariser(c, v, numVertices, f, numTriangles, aff, g, val, g_reg_overwrite, nullptr);
    if (rc) return rc;
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return fail(NLOS_ERR_HIP, std::string("hipDeviceSynchronize: ") + hipGetErrorString(e));
    for (auto& d : hc.downloads) {
        e = hipMemcpy(d.first, d.second.first, d.second.second, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return fail(NLOS_ERR_HIP, std::string("hipMemcpy D2H: ") + hipGetErrorString(e));
    }
    return NLOS_OK;
}
}  // namespace

int nlos_streamed_render_normal_smoothing(float* vertices, int numVertices, int* triangles, int numTriangles,
                                          int* face_affinity, double* curvature_grad, double* value_out) {
    if (!face_affinity || !value_out) return fail(NLOS_ERR_ARG, "normal smoothing: face_affinity / value_out is NULL");
    return host_regulariser(vertices, numVertices, triangles, numTriangles, face_affinity, curvature_grad, value_out);
}

int nlos_streamed_render_curvature_grad(float* vertices, int numVertices, int* triangles, int numTriangles,
                                        double* curvature_grad) {
    return host_regulariser(vertices, numVertices, triangles, numTriangles, nullptr, curvature_grad, nullptr);
}

}  // extern "C"
