// nlos_kernels.h -- host-visible launchers of the gfx950 kernels (internal header).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nlos {

// What the launchers decided for the current render (host side only).  nlos_render points `tl_note` at
// its note before it calls the launchers; they record the back-end they chose, why, and the first HIP
// error of an attribute call or launch -- nothing falls back or fails silently (nlos_ctx_last_path).
struct LaunchNote {
    int backend = 0, reason = 0, grid_R = 0, tiles = 0, tile_cap = 0, rows_in_lds = 0, gradient_kernel = 0;
    int retry_workgroups = 0;        // entries of the retry / path-code array the grid launches wrote
    int prologue_done = 0;           // 1: the pass-1 kernel cleared ForwardArgs::zero and wrote ForwardArgs::pathlengths itself
    int vis_items = 0;               // 1: pass 1 recorded the visibility cache as item masks (ForwardArgs::vis_items), not per-face words
    // lazy scene build (grid back-end): the records exist, the tree does not yet.  The launchers complete it -- on a
    // device-side flag between the grid's two launches, unconditionally in front of a BVH back-end
    const struct BuildArgs* lazy_build = nullptr;
    bool tree_built = false;         // set when the tree was completed unconditionally
    hipError_t err = hipSuccess;
    const char* err_what = nullptr;
};
extern thread_local LaunchNote* tl_note;

// Every behaviour-changing environment switch of the library, read ONCE per process by one accessor (round 6: the launchers
// and nlos_render used to read them through separate function-local statics); nlos_env_report() lists them with their values.
struct EnvSwitches {
    int tile_threshold;                  // NLOS_TILE_THRESHOLD   faces above which pass 1 uses the tiled grid            (6200)
    bool lazy_tree;                      // NLOS_LAZY_TREE        build the BVH only if a back-end asks for it            (1)
    bool fuse_residual;                  // NLOS_FUSE_RESIDUAL    pass 2 forms the residual itself                        (1)
    int tile_tris;                       // NLOS_TILE_TRIS        triangles per slope-space tile                          (3000)
    unsigned long long tile_scratch_max; // NLOS_TILE_SCRATCH_MAX bytes of tile-subset scratch before sources are chunked (32 GiB)
    bool vis_items;                      // NLOS_VIS_ITEMS        visibility cache as item masks                          (1)
    bool geo_cache;                      // NLOS_GEO_CACHE        pass 1 -> pass 2 geometry cache                         (1)
    double geo_cache_max_gb;             // NLOS_GEO_CACHE_MAX_GB bound of that cache (< 0: min(32 GB, half of free))     (-1)
    size_t row_lds_max;                  // NLOS_ROW_LDS_MAX      bytes of histogram row kept in LDS                      (10240)
    int grad_wide;                       // NLOS_GRAD_WIDE        pass 2 may use its wide instance                        (1)
    int grad_min_sources;                // NLOS_GRAD_MIN_SOURCES sources per pass-2 workgroup at least (small L)         (1)
    bool fwd_order;                      // NLOS_FWD_ORDER        pass 1 takes its sources in Z-order of the wall         (1)
    int geo_max_spt;                     // NLOS_GEO_MAX_SPT      strata per face up to which pass 1 records the geometry cache (8)
};
const EnvSwitches& env_switches();
void launch_digest_u32(const uint32_t* w, size_t n, unsigned long long* out, hipStream_t stream);
// perm = the sources in Z-order of their positions (ties: ascending index); false: L beyond what the kernel sorts
bool launch_order_sources(const float* origin, int L, int* perm, hipStream_t stream);
inline void note_hip(hipError_t e, const char* what) {
    if (e != hipSuccess && tl_note && tl_note->err == hipSuccess) { tl_note->err = e; tl_note->err_what = what; }
}

// Compute units of the current device (hipDeviceAttributeMultiprocessorCount): the persistent grids are sized from it, so
// that a partitioned MI355X (CPX / DPX modes expose 32 / 128 CUs per device) gets as many workgroups as it can hold, not
// the 256-CU chip's.  Queried once per device.
inline int device_cu_count() {
    static int cache[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (cache[dev] <= 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cache[dev] = n;
    }
    return cache[dev];
}

// ------------------------------------------------------------------ BVH build
struct BuildArgs {
    const float* vertices;   // [V,3]
    const int32_t* faces;    // [F,3]
    int V, F;
    // scratch
    uint32_t *keys0, *keys1;
    int *idx0, *idx1;
    int* child;              // [2*(F-1)]
    int* range;              // [2*(F-1)]
    int* parent;             // [2F-1]
    int* arrive;             // [F-1]
    float* box;              // [6*(2F-1) + 8]
    int* status;             // [1] bit0: face index out of range
    int lazy;                // 1: bounds, keys, sort, records and the root box only (what the perspective grid reads); the tree
                             //    follows by launch_build_tree() if somebody needs occluded() / closest_hit()
    int* need_tree;          // lazy builds: device word the grid kernels raise when a workgroup needs the BVH query (or null)
    int* host_status;        // device-visible pinned host word (or null): the single-workgroup builder leaves status[0] there on its way out
    // outputs
    float4* nodes;           // [2*(2F-1)]  pre-order, 32 B per node
    float4* tris;            // [4*F]       64 B per triangle, Morton order
    float4* facerec;         // [4*F]       64 B per face, Morton order
    int* face_id;            // [F]         original face index of sorted slot j
    float* tri_zmin;         // [F]         smallest vertex z of sorted triangle j
};
// true: the kernel itself stored the status word to a.host_status (no copy needed behind the build)
bool launch_build_bvh(const BuildArgs& a, hipStream_t stream);
// completes a lazy build (tree, refit, node emission; chip-wide kernels).  conditional: the kernels leave at once
// unless *a.need_tree != 0
void launch_build_tree(const BuildArgs& a, bool conditional, hipStream_t stream);

// ------------------------------------------------------------------ rendering
struct SceneView {
    const float4* nodes;
    const float4* tris;
    const float4* facerec;
    const int* face_id;
    const float* tri_zmin;
    int n_nodes, F, V;
    const float* vertex_normal;   // [V,3] or null
    const float* albedo;          // [V]   or null
};

struct SourceView {
    const float* origin;     // [L,3]
    const float* normal;     // [L,3]
    int L;
    long long source_offset; // global index of origin[0]
    int source_stride;       // global index of origin[l] = source_offset + l * source_stride (0: every source draws the same samples)
    int n_sensors;           // > 0 (pass 2 of the product): measurement l is the pair (origin[l / n_sensors], sensor[l % n_sensors])
    int total_sources;       // for 1/L
    // row N (non-confocal pairs): sensor point / wall normal of pair l; null = confocal
    const float* sensor;     // [L,3] or null
    const float* sensor_normal;  // [L,3] or null
};

struct SampleParams {
    uint64_t seed;
    int spt;                 // samples per (source, face)
    float lb, ub, res;       // res = bin width used for forward binning (res/refine when refined)
    int nbins;               // bins of the forward histogram row (T * refine)
    int clamp;               // 1: v2 clamped form factor
    int use_ggx;
    float ggx_alpha;
    int sampled_point;       // 1: v1 non-streamed forward body (path length, bin, normal, albedo from the SAMPLED point); BVH back-end only
};

// forward (rows S, I, F): histogram rows + optional visibility cache
struct ForwardArgs {
    SceneView sc;
    SourceView src;
    SampleParams sp;
    double* rows;            // [L, nbins]   overwritten
    uint32_t* vis;           // [L, vis_words, F] accepted-sample bitmasks (or null)
    int vis_words;
    // The same cache in the order pass 1 produces it (single-workgroup grid, confocal, spt <= 32): per source
    // [0] = header (bits 15:0 number of live faces, 39:16 rays traced, 63:40 samples accepted), [1 + i] = the 64-bit accepted mask of 64-ray item i; ray r = li * spt + s of the
    // bucketed live list `live` (which then stays valid until the next pass 1).  One 8-byte store per item instead
    // of a word per (face, 32 strata) plus the dark faces' zero words: 4.5 MB instead of 145 MB on the metric workload.
    unsigned long long* vis_items;   // [L, items_stride] or null
    int items_stride;
    double* intensity;       // [F] (mode intensity: accumulated with atomics; rows unused)
    int mode_intensity;
    int force_bvh;           // 1: never use the per-source perspective grid (tests / large meshes)
    long long* dbg;          // diagnostic builds only (NLOS_FWD_STAMPS); null in the product
    uint16_t* live;          // [L, F] scratch: per-source bucketed list of contributing faces (grid kernel)
    uint16_t* cov;           // [L, F] scratch: cells every triangle enters (counting pass -> fill pass of the grid kernel)
    uint32_t* vis2;          // [L, vis_words, F] scratch: sensor-leg visibility of non-confocal pairs (grid path) or null
    // tiled grid (meshes beyond one workgroup's LDS): slope space is cut into tiles_x * tiles_y tiles, one
    // workgroup per (source, tile); tile_list holds each workgroup's triangle subset (tile_cap ids each)
    uint32_t* tile_list;     // [L * tiles, tile_cap] scratch or null
    int* tile_count;         // [L * tiles] subset sizes (may exceed tile_cap: overflow), filled by k_tile_bin
    int tiles_x, tiles_y, tile_cap;
    int* retry;              // [workgroups] flags of the big-LDS second launch (grid kernels) or null
    // What k_residual does besides the residual, carried by the first workgroups of the single-workgroup grid kernel when
    // pass 2 computes the residual itself (GradientArgs::inline_residual): the step then holds no residual launch.
    double* zero;            // [zero_n] cleared (the gradient output of zero_gradient renders) or null
    size_t zero_n;
    double* pathlengths;     // [path_T] = (double)(path_lb + i * path_res) or null
    float path_lb, path_res;
    int path_T;
    // Geometry cache pass 1 -> pass 2 (round 4; confocal single-workgroup grid with item masks only): per ray r = li * spt + s
    // of the bucketed live list the five numbers pass 2 cannot get cheaply -- the sampled direction and the hit's
    // barycentrics (v, w) of sample_geo() -- so that pass 2 neither hashes nor repeats the own-face triangle test.
    float* geo;              // [L, geo_stride] float4 (dir.x, dir.y, dir.z, v) followed by [L, geo_stride] float2 (w, h): one dwordx4 + one
                             // dwordx2 per ray (five dwords of a 20-byte record cost pass 2 0.29 ms: profiles/r04_ab_geo_cache.log); or null
    int geo_sources;         // L of the allocation (where the float part starts)
    int geo_stride;          // rays per source the cache holds (F * spt)
    // record pass of the product (row N as L x S): per wall point and sample r = sorted face slot * spt + s, the path
    // length of the leg and its clamped form factor, 0 where the sample is not seen from that wall point
    float* rec_d;            // [L, F * spt] or null
    float* rec_ff;           // [L, F * spt]
    // scenes with vertex normals / albedo (round 6): a pair's normal and albedo are interpolated at the LASER leg's hit, so the
    // sensor leg's form factor depends on the laser; what makes the legs separable again -- this leg's unit direction (the
    // sensor role: the combine kernel forms its cosine with the laser's normal), and the normal + albedo interpolated at this
    // leg's own hit (the laser role).  rec_ext: 7 arrays of [L, F * spt] floats: dir.x, dir.y, dir.z, n.x, n.y, n.z, albedo;
    // rec_d > 0 then means "seen from this wall point" (rec_ff may be 0 there: it is the laser role's form factor).
    float* rec_ext;          // or null (face normals, no albedo: rec_d / rec_ff alone)
    size_t rec_ext_stride;   // floats between two of the seven arrays
    // Order in which the one-workgroup-per-source launch takes its sources (round 6): blockIdx -> source through `perm`, the
    // sources sorted along a Z-order curve over their wall positions, so that the ~512 workgroups resident at any time
    // render a compact patch of the wall instead of eight full rows of it (they then gather the same face records and
    // finish together: forward 1.348 -> 1.316 ms on the metric, profiles/r06_source_order_probe.log).  Any permutation gives
    // the same results; null = identity.
    const int* perm;         // [L] or null
    int* need_tree;          // lazy scene build: raised by first-launch workgroups that need the BVH query (they then leave it to
                             // the second launch, in front of which the tree is completed); null = the tree exists
};
void launch_forward(const ForwardArgs& a, hipStream_t stream);
// record pass of the product (single-workgroup grid, face normals, Lambertian); false: not available for this scene
bool launch_forward_record(const ForwardArgs& a, hipStream_t stream);
// combine kernel of the product: rows [La * Sb, T] (overwritten) and, if `vis` is given, the accepted-sample words
// [La * Sb, 1, F] pass 2 reads
struct ProductArgs {
    SceneView sc;
    const float* d_a; const float* ff_a;   // laser records [La, F * spt]
    const float* d_b; const float* ff_b;   // sensor records [Sb, F * spt]
    const float* ext_a; const float* ext_b;   // extended records (ForwardArgs::rec_ext) of the laser / sensor set, or null
    size_t ext_stride;
    const float* sensor_normal;            // [Sb, 3] (extended records: the sensor's wall cosine is formed in the combine)
    int La, Sb, spt, nbins;
    float lb, ub, res;
    double* rows;
    uint32_t* vis;
};
void launch_product_combine(const ProductArgs& a, hipStream_t stream);
void launch_expand_pairs(const float* laser, const float* lnormal, const float* sensor, const float* snormal, int La, int Sb,
                         float* out_l, float* out_ln, float* out_s, float* out_sn, hipStream_t stream);
// the two back-ends behind launch_forward (forward_grid.hip returns false when the BVH back-end is needed)
bool launch_forward_grid(const ForwardArgs& a, int rows_in_lds, hipStream_t stream);
void launch_forward_bvh(const ForwardArgs& a, int rows_in_lds, hipStream_t stream);

// refined-histogram Gaussian + fold (row FD, refine > 1)
struct SmoothArgs {
    const double* fine;      // [L, T*refine]
    double* transient;       // [L, T]
    const double* kernel;    // [K]
    int L, T, refine, sigma_bin, K;
    int offset;              // y index of output bin 0 (2*refine*sigma_bin for the Gaussian, jitter_offset for row J)
};
void launch_smooth(const SmoothArgs& a, hipStream_t stream);

// residual (row D) + pathlengths + optional v1 box filter (row W)
struct ResidualArgs {
    const double* data;      // [L,T]
    const double* weight;    // [L,T] or null
    const double* transient; // [L,T]
    double* diff;            // [L,T]
    double* pathlengths;     // [T] or null
    int L, T;
    int loss_test;
    float lb, res;
    int w_width;             // > 0: box(2w+1) (*) box(2w+1) per row, 'same' crop
    double* zero;            // [zero_n] or null: cleared by the same launch (the gradient output of zero_gradient renders)
    size_t zero_n;
};
void launch_residual(const ResidualArgs& a, hipStream_t stream);

// gradient (rows G, GD, A, G1, vertex gradient)
struct GradientArgs {
    SceneView sc;
    SourceView src;
    SampleParams sp;         // res = caller's resolution, nbins = T
    const double* diff;      // [L,T]
    const uint32_t* vis;     // [L, vis_words, F]
    int vis_words;
    const unsigned long long* vis_items;   // item-mask layout of the cache (ForwardArgs::vis_items) or null
    const uint16_t* live;                  // [L, F] bucketed live lists the item masks refer to
    int items_stride;
    uint32_t* vis_scratch;                 // [L, vis_words, F]: where the face-major kernel gets per-face words from item masks
    const float* geo;                      // geometry cache of pass 1 (ForwardArgs::geo; item-mask layout only) or null
    int geo_stride, geo_sources;
    // inline_residual = 1: `diff` is scratch, the kernel forms (data - transient)[^3 * 2] * weight itself while it stages a
    // source's row in LDS (row D, smoothed_transient/stratifiedStreamedGradientRenderer.cpp:543-550); the face-major kernel
    // does not, the launcher then runs k_residual into `diff_scratch` first
    int inline_residual;
    const double* res_data;                // [L,T]
    const double* res_weight;              // [L,T] or null
    const double* res_transient;           // [L,T]
    int res_loss_test;
    double* diff_scratch;                  // [L,T]
    const double* tap_w;     // [K] weighting_kernal
    const double* tap_delta; // [K] delta_length (float-evaluated, widened)
    const double* tap_g;     // [K] (float)(delta/sigma^2*2) widened
    const double* tap_p0;    // [K+1] prefix sums of (double)(float)w
    const double* tap_p1;    // [K+1] prefix sums of g * (double)(float)w
    const double* tap_pw;    // [K+1] prefix sums of w (double): scalar gradients
    const double* tap_wt;    // [(refine + 1) * tap_nb * 2] per-bin weights by first-boundary tap (render_common.h, TapTables::wt) or null
    int tap_nb;              // 4 sigma_bin + 1
    int wt_in_lds;           // set by launch_gradient: k_gradient stages tap_wt in LDS
    int lean_params;         // set by launch_gradient: the window lies in the range of the lean arithmetic (render_common.h: lean_params_ok)
    int K;
    int two_rs;              // 2*refine*sigma_bin (index of the centre tap)
    double r_over_res;       // refine / resolution
    int refine;              // taps per bin (consecutive bin boundaries are this many taps apart)
    int normal_term;         // 0/1 (already resolved)
    int v1_style;            // 1: G1 (t1 without albedo)
    int mode;                // 0 vertex gradient [V,3], 1 scalar albedo, 2 scalar alpha, 3 single vertex per bin,
                             // 4 vertex gradient with jitter taps (tap_w = jitter_weight, tap_g = jitter_grad,
                             //   K = jitter_length, two_rs = jitter_offset; delta/p0/p1 unused)
    int vertex_num;
    double* out;             // gradient [V,3] | scalar [1] | [T,3]
    int lds_grad;            // in: 1 = per-workgroup LDS accumulator allowed; the launcher clears it when 3V doubles do not fit
    int compact;             // set by the launcher: 1 = compacted u16 list of faces with accepted samples in LDS
};
void launch_gradient(const GradientArgs& a, hipStream_t stream);
// item-mask visibility cache -> per-face words [L, 1, F] (spt <= 32)
void launch_items_to_words(const unsigned long long* items, int items_stride, const uint16_t* live, int L, int F, int spt,
                           uint32_t* words, hipStream_t stream);

// closest-hit batch (row E)
struct IntersectArgs {
    SceneView sc;
    const float* origins;    // [N,3]
    const float* dirs;       // [N,3]
    int n;
    float* out3;             // [N,3] or null
    float* out1;             // [N]   or null
};
void launch_intersect(const IntersectArgs& a, hipStream_t stream);

// mesh regularisers: area ("curvature") gradient and normal smoothing
struct RegulariserArgs {
    const float* vertices;   // [V,3]
    const int32_t* faces;    // [F,3]
    const int32_t* affinity; // [F,3] neighbour face per edge or -1; null = area gradient only
    int V, F;
    double* normal;          // [3F] scratch
    double* area;            // [F]  scratch
    int* owner;              // [V]  scratch (overwrite semantics only)
    double* gradient;        // [V,3] zeroed by the launcher
    double* value;           // [1]   zeroed by the launcher (may be null without affinity)
    int overwrite;           // 0: accumulate over incident faces, 1: highest incident face wins
};
void launch_regulariser(const RegulariserArgs& a, hipStream_t stream);

// optimiser-side steps that follow the render in every loop of the reference
struct AdamArgs {
    float* params;           // [rows, cols] updated in place
    const double* grad64;    // [rows, cols] (renderer output) or null
    const float* grad32;     // [rows, cols] or null (exactly one of the two)
    float* exp_avg;          // [rows, cols]
    float* exp_avg_sq;       // [rows, cols]
    float* max_exp_avg_sq;   // [rows, cols] (amsgrad) or null
    const uint8_t* row_mask; // [rows] rows with 0 are left untouched; null = all rows
    int rows, cols;
    float beta1, beta2, eps, weight_decay;
    float one_minus_beta1, one_minus_beta2;
    float step_size;         // lr * sqrt(1 - beta2^t) / (1 - beta1^t), evaluated in double on the host
};
void launch_adam_modified(const AdamArgs& a, hipStream_t stream);
// weight = ((data / max(data) + 0.1)^gamma), rescaled to sum to n (exp_bunny/rendering.py:208-217)
void launch_weighting(const double* data, size_t n, double gamma, double* weight, hipStream_t stream);
// out[0] = sum weight * (transient - data)^2 / rows (exp_bunny/rendering.py:360-364, L1)
void launch_weighted_l2(const double* transient, const double* data, const double* weight, size_t n, int rows,
                        double* out, hipStream_t stream);

// small utilities
void launch_zero_f64(double* p, size_t n, hipStream_t stream);
void launch_bary_to_world(const float* V, const int32_t* F, const float* bary, int n, float* out,
                          hipStream_t stream);

// LDS budget the gradient kernel may use for its accumulator (bytes)
constexpr int kGradLdsBudget = 150 * 1024;

}  // namespace nlos
