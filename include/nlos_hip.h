/*
 * nlos_hip.h -- C ABI of libnlos_hip.so, the MI355X (gfx950) transient renderer.
 *
 * Drop-in boundary for the native interface of cmu-ci-lab/nlos_surface_optimization
 * (paths relative to transient_rendering_cython/ in the reference):
 *   - section 1 mirrors, parameter for parameter, the C++ free functions the
 *     reference's Cython modules bind (`cdef extern from "...h"`); they take HOST
 *     pointers, are synchronous and stateless, and return an int status where the
 *     reference returns void (0 = ok, see nlos_last_error()).
 *   - section 2 is the device-resident family (device pointers + hipStream_t as
 *     void*), an allowed addition (SURVEY.md section 8b "Ownership") used by the
 *     torch path, the benchmark and the multi-GPU source sharding.
 * Plain pointers and sizes only; no torch / C++ types cross this boundary.
 */
#ifndef NLOS_HIP_H
#define NLOS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NLOS_OK              0
#define NLOS_ERR_ARG        -1
#define NLOS_ERR_HIP        -2
#define NLOS_ERR_NO_DEVICE  -3

/* thread-local message of the last failing call */
const char *nlos_last_error(void);
/* number of visible HIP devices (0 when none / no driver); never throws */
int nlos_device_count(void);
/* library version: major*10000 + minor*100 + patch */
int nlos_version(void);
/* Every behaviour-changing environment switch of the library with the value this process read (once, at first use) and its
 * default in parentheses, one "NAME=value (default)" per line.  Writes at most `cap` bytes (NUL-terminated) into `buf`
 * (which may be NULL) and returns the length of the full text. */
int nlos_env_report(char *buf, int cap);

/* ------------------------------------------------------------------------
 * Section 1 -- host-pointer drop-ins (reference signatures)
 * ------------------------------------------------------------------------ */

/* smoothed_transient/stratifiedStreamedTransientRenderer.h:3-5 (v2 forward;
 * vertexNormal / vertexAlbedo may be NULL) */
int nlos_streamed_render_transient(float *origin, int numSources, float *normal,
        float *vertices, int numVertices, float *vertexNormal, float *vertexAlbedo,
        int *triangles, int numTriangles, int numSamples, float lowerBound,
        float upperBound, float resolution, double *transient, double *pathlengths,
        int refine_scale, int sigma_bin);

/* smoothed_transient/stratifiedStreamedTransientRenderer.h (streamed_render_intensity);
 * intensity[numTriangles] is accumulated into */
int nlos_streamed_render_intensity(float *origin, int numSources, float *normal,
        float *vertices, int numVertices, float *vertexNormal, int *triangles,
        int numTriangles, int numSamples, float lowerBound, float upperBound,
        double *intensity);

/* smoothed_transient/stratifiedStreamedGradientRenderer.h:3-13 */
int nlos_streamed_render_gradient(double *data, double *weight, float *origin,
        int measurement, float *normal, float *vertices, int numVertices,
        float *vertexNormal, int *triangles, int numTriangles, int numSamples,
        float lowerBound, float upperBound, float resolution, double *transient,
        double *pathlengths, double *gradient, int refine_scale, int sigma_bin,
        int testing_flag, int loss_test);

int nlos_streamed_render_gradient_w_albedo(double *data, double *weight, float *origin,
        int measurement, float *normal, float *vertices, int numVertices,
        float *albedo, int *triangles, int numTriangles, int numSamples,
        float lowerBound, float upperBound, float resolution, double *transient,
        double *pathlengths, double *gradient, int refine_scale, int sigma_bin,
        int testing_flag, int loss_test);

/* reference returns the scalar; here it is written to *grad_out */
int nlos_streamed_render_gradient_albedo(double *data, double *weight, float *origin,
        int measurement, float *normal, float *vertices, int numVertices,
        float *albedo, int *triangles, int numTriangles, int numSamples,
        float lowerBound, float upperBound, float resolution, double *transient,
        double *pathlengths, int refine_scale, int sigma_bin, int testing_flag,
        int loss_test, double *grad_out);

/* gradient is [numBins,3], accumulated into (renderer.pyx:78-88) */
int nlos_streamed_render_vertex_gradient(int vertex_num, float *origin, int measurement,
        float *normal, float *vertices, int numVertices, int *triangles,
        int numTriangles, int numSamples, float lowerBound, float upperBound,
        float resolution, double *gradient, int refine_scale, int sigma_bin);

/* ggx/stratifiedStreamed{Transient,Gradient}Renderer.h: as above with `float alpha`
 * after numTriangles */
int nlos_ggx_streamed_render_transient(float *origin, int numSources, float *normal,
        float *vertices, int numVertices, float *vertexNormal, float *vertexAlbedo,
        int *triangles, int numTriangles, float alpha, int numSamples, float lowerBound,
        float upperBound, float resolution, double *transient, double *pathlengths,
        int refine_scale, int sigma_bin);
int nlos_ggx_streamed_render_intensity(float *origin, int numSources, float *normal,
        float *vertices, int numVertices, float *vertexNormal, int *triangles,
        int numTriangles, float alpha, int numSamples, float lowerBound,
        float upperBound, double *intensity);
int nlos_ggx_streamed_render_gradient(double *data, double *weight, float *origin,
        int measurement, float *normal, float *vertices, int numVertices,
        float *vertexNormal, int *triangles, int numTriangles, float alpha,
        int numSamples, float lowerBound, float upperBound, float resolution,
        double *transient, double *pathlengths, double *gradient, int refine_scale,
        int sigma_bin, int testing_flag);
int nlos_ggx_streamed_render_gradient_alpha(double *data, double *weight, float *origin,
        int measurement, float *normal, float *vertices, int numVertices,
        float *vertexNormal, int *triangles, int numTriangles, float alpha,
        int numSamples, float lowerBound, float upperBound, float resolution,
        double *transient, double *pathlengths, int refine_scale, int sigma_bin,
        double *grad_out);

/* stratified_transient_raytracer/stratifiedStreamedGradientRenderer.h:4 (v1) and
 * stratifiedStreamedTransientRenderer.h (v1 forward, unclamped form factor) */
int nlos_v1_streamed_render_gradient(double *data, float *origin, int measurement,
        float *normal, float *vertices, int numVertices, int *triangles,
        int numTriangles, int numSamples, float lowerBound, float upperBound,
        float resolution, int w_width, double *transient, double *pathlengths,
        double *gradient);
int nlos_v1_streamed_render_transient(float *origin, int numSources, float *normal,
        float *vertices, int numVertices, float *vertexNormal, float *vertexAlbedo,
        int *triangles, int numTriangles, int numSamples, float lowerBound,
        float upperBound, float resolution, double *transient, double *pathlengths);

/* stratified_transient_raytracer/stratifiedTransientRenderer.h (v1, NOT streamed: one wall point, rows are
 * 1-D; renderer.pyx:93-102 `renderTransient`, still called by stratified_transient_raytracer/test.py:37).
 * Unclamped form factor, face normals, no albedo.  The reference body (stratifiedTransientRenderer.cpp:96-124) takes
 * path length, bin and barycentrics from the SAMPLED point of the stratified map, not from the hit Embree reports
 * back; so does this entry (nlos_render_args.v1_sampled_point; oracle option `sampled_point`).  Remaining deviation: the
 * ray direction is normalised by a multiply with the IEEE reciprocal, as everywhere in the contract, where the v1
 * body divides. */
int nlos_v1_render_transient(float *origin, float *normal, float *vertices, int numVertices,
        int *triangles, int numTriangles, int numSamples, float lowerBound, float upperBound,
        float resolution, double *transient, double *pathlengths);

/* Non-confocal (laser, sensor) pairs -- SURVEY.md 8a row N.  No native reference function exists
 * (prototypes: transient_rendering_python/rendering.py:8-93, mesh_optimization/rendering.py:739-797);
 * the parameter lists extend streamed_render_transient / streamed_render_gradient by the sensor
 * arrays.  Pair i = (laser[i], sensor[i]).  gradient is accumulated into (v2 semantics). */
int nlos_nonconfocal_render_transient(float *laser, float *laserNormal, float *sensor,
        float *sensorNormal, int numPairs, float *vertices, int numVertices,
        float *vertexNormal, float *vertexAlbedo, int *triangles, int numTriangles,
        int numSamples, float lowerBound, float upperBound, float resolution,
        double *transient, double *pathlengths, int refine_scale, int sigma_bin);
int nlos_nonconfocal_render_gradient(double *data, double *weight, float *laser,
        float *laserNormal, float *sensor, float *sensorNormal, int numPairs,
        float *vertices, int numVertices, float *vertexNormal, float *vertexAlbedo,
        int *triangles, int numTriangles, int numSamples, float lowerBound, float upperBound,
        float resolution, double *transient, double *pathlengths, double *gradient,
        int refine_scale, int sigma_bin, int testing_flag, int loss_test);

/* The product of a laser set and a sensor set (nlos_render_args.n_sensors): transient / data / weight are
 * [numLasers, numSensors, numBins]; every (laser, sensor) combination is a pair of the functions above, rendered on
 * samples shared by all wall points. */
int nlos_nonconfocal_product_render_transient(float *laser, float *laserNormal, int numLasers, float *sensor,
        float *sensorNormal, int numSensors, float *vertices, int numVertices, int *triangles, int numTriangles,
        int numSamples, float lowerBound, float upperBound, float resolution,
        double *transient, double *pathlengths);
int nlos_nonconfocal_product_render_gradient(double *data, double *weight, float *laser, float *laserNormal,
        int numLasers, float *sensor, float *sensorNormal, int numSensors, float *vertices, int numVertices,
        int *triangles, int numTriangles, int numSamples, float lowerBound, float upperBound,
        float resolution, double *transient, double *pathlengths, double *gradient,
        int refine_scale, int sigma_bin, int testing_flag, int loss_test);

/* the same with the GGX BRDF of the `ggx` module (`float alpha` after numTriangles, as ggx/...Renderer.h place it):
 * brdf = D(n.h) G1(n.w_laser) G1(n.w_sensor) / 4 with the half vector h -- ggx_confocal.cpp's eval() for
 * w_laser == w_sensor */
int nlos_ggx_nonconfocal_render_transient(float *laser, float *laserNormal, float *sensor,
        float *sensorNormal, int numPairs, float *vertices, int numVertices,
        float *vertexNormal, float *vertexAlbedo, int *triangles, int numTriangles, float alpha,
        int numSamples, float lowerBound, float upperBound, float resolution,
        double *transient, double *pathlengths, int refine_scale, int sigma_bin);
int nlos_ggx_nonconfocal_render_gradient(double *data, double *weight, float *laser,
        float *laserNormal, float *sensor, float *sensorNormal, int numPairs,
        float *vertices, int numVertices, float *vertexNormal, float *vertexAlbedo,
        int *triangles, int numTriangles, float alpha, int numSamples, float lowerBound, float upperBound,
        float resolution, double *transient, double *pathlengths, double *gradient,
        int refine_scale, int sigma_bin, int testing_flag, int loss_test);

/* smoothed_transient/stratifiedStreamedGradientRenderer.h (streamed_render_normal_smoothing,
 * streamed_render_curvature_grad; bodies :27-180).  curvature_grad [numVertices,3] is zeroed and
 * filled; the reference returns the smoothing value, here it is written to *value_out.
 * Per-vertex results are ACCUMULATED over the incident faces (the gradient of the formulas); the
 * reference stores them with `=`, i.e. keeps whichever incident face wrote last --
 * nlos_set_regulariser_overwrite(1) selects that behaviour deterministically (highest face index
 * wins, the outcome of a serial run). */
int nlos_streamed_render_normal_smoothing(float *vertices, int numVertices, int *triangles,
        int numTriangles, int *face_affinity, double *curvature_grad, double *value_out);
int nlos_streamed_render_curvature_grad(float *vertices, int numVertices, int *triangles,
        int numTriangles, double *curvature_grad);
void nlos_set_regulariser_overwrite(int overwrite);

/* jitter/stratifiedStreamedTransientRenderer.h, jitter/stratifiedStreamedGradientRenderer.h:10
 * (the `jitter` extension module: measured SPAD jitter kernel instead of the Gaussian) */
int nlos_jitter_streamed_render_transient(float *origin, int numSources, float *normal,
        float *vertices, int numVertices, float *vertexNormal, float *vertexAlbedo,
        int *triangles, int numTriangles, int numSamples, float lowerBound,
        float upperBound, float resolution, double *weight, int weight_offset,
        int weight_length, double *transient, double *pathlengths);
int nlos_jitter_streamed_render_gradient(double *data, double *weight, float *origin,
        int measurement, float *normal, float *vertices, int numVertices,
        float *vertexNormal, int *triangles, int numTriangles, int numSamples,
        float lowerBound, float upperBound, float resolution, double *jitter_weight,
        double *jitter_grad, int weight_offset, int weight_length, double *transient,
        double *pathlengths, double *gradient, int testing_flag);

/* embree_intersector/c_embree_intersector.h:3-9 */
int nlos_embree3_tbb_line_intersection(float *origins, float *directions, int num_ray,
        float *vertices, int num_vertices, int *triangles, int num_triangles,
        float *intersect /* [num_ray,3] */);
int nlos_embree3_tbb_short_line_intersection(float *origins, float *directions,
        int num_ray, float *vertices, int num_vertices, int *triangles,
        int num_triangles, float *intersect /* [num_ray] */);
/* barycentric_to_world (c_embree_intersector.h:4) takes no array sizes; the device
 * path needs them to stage V and F, so the two counts are added */
int nlos_barycentric_to_world_n(float *vertices, int num_vertices, int *triangles,
        int num_triangles, float *barycoord, int num_ray, float *intersection_p);

/* process-wide knobs used by the host-pointer drop-ins (the reference has none:
 * its RNG seeding is fixed, STR/sampler.cpp:20-34) */
void nlos_set_default_seed(uint64_t seed);
void nlos_set_default_device(int device);

/* ------------------------------------------------------------------------
 * Section 2 -- device-resident family
 * ------------------------------------------------------------------------ */
typedef struct nlos_ctx nlos_ctx;   /* per-device scratch: BVH, visibility cache, residual */

int  nlos_ctx_create(int device, nlos_ctx **out);
void nlos_ctx_destroy(nlos_ctx *ctx);
/* bytes of device scratch currently held */
int64_t nlos_ctx_scratch_bytes(const nlos_ctx *ctx);

enum {
    NLOS_MODE_TRANSIENT       = 0,  /* rows S,I,F,FD */
    NLOS_MODE_GRADIENT        = 1,  /* + D,G,GD */
    NLOS_MODE_INTENSITY       = 2,  /* row X */
    NLOS_MODE_GRAD_ALBEDO     = 3,  /* row A  */
    NLOS_MODE_GRAD_ALPHA      = 4,  /* GGX d/d alpha */
    NLOS_MODE_VERTEX_GRADIENT = 5,  /* single-vertex per-bin gradient */
    NLOS_MODE_GRADIENT_V1     = 6   /* rows W,G1 */
};

/* All pointers are DEVICE pointers (same dtypes/layouts as section 1). */
typedef struct nlos_render_args {
    int32_t mode;
    /* sources (this rank's block) */
    const float *origin;        /* [L,3] */
    const float *normal;        /* [L,3] */
    int32_t L;
    int64_t source_offset;      /* global index of origin[0]; keys the RNG */
    int32_t total_sources;      /* global L for the 1/L normalisation; 0 -> L */
    /* mesh */
    const float *vertices;      /* [V,3] */
    int32_t V;
    const int32_t *faces;       /* [F,3] */
    int32_t F;
    const float *vertex_normal; /* [V,3] or NULL */
    const float *albedo;        /* [V]   or NULL */
    /* sampling / binning */
    int32_t num_samples;
    float lower_bound, upper_bound, resolution;
    int32_t refine_scale, sigma_bin;
    uint64_t seed;
    /* measurement */
    const double *data;         /* [L,T] */
    const double *weight;       /* [L,T] (NULL -> 1) */
    /* outputs */
    double *transient;          /* [L,T]  overwritten */
    double *pathlengths;        /* [T]    overwritten (may be NULL) */
    double *gradient;           /* [V,3] accumulated into ([T,3] for VERTEX_GRADIENT;
                                   zeroed first for GRADIENT_V1) */
    double *intensity;          /* [F]   accumulated into */
    double *scalar_out;         /* [1]   overwritten (GRAD_ALBEDO / GRAD_ALPHA) */
    /* flags */
    int32_t testing_flag, loss_test;
    int32_t normal_term;        /* -1 reference rule, 0 off, 1 on */
    int32_t clamp;              /* 1 v2 (default), 0 v1 unclamped forward */
    int32_t use_ggx;
    float   ggx_alpha;
    int32_t vertex_num;         /* VERTEX_GRADIENT */
    int32_t w_width;            /* GRADIENT_V1 */
    int32_t reuse_bvh;          /* 1: skip the scene build and use the tree of generation `mesh_generation`
                                   (fails unless that is still the tree the ctx holds) */
    /* autograd support (additions; the reference always derives the residual itself) */
    const double *residual;     /* [L,T] or NULL. GRADIENT: use this as `difference` instead of
                                   (data - transient) * weight; data/weight may then be NULL */
    int32_t keep_visibility;    /* TRANSIENT: also record the per-sample visibility cache */
    int32_t reuse_visibility;   /* GRADIENT with residual: skip pass 1 and reuse the visibility cache of
                                   generation `visibility_generation` (fails unless the ctx still holds that
                                   cache, recorded on the tree it currently holds, for the same sources,
                                   samples and seed) */
    int32_t force_bvh;          /* 1: occlusion by BVH traversal only (default 0: per-source perspective
                                   grid in LDS, tiled over several workgroups per source for meshes beyond
                                   6.2 k faces; identical results).  2 (diagnostic): tiled grid with a tiny
                                   per-tile capacity, to exercise the tiles' overflow fallback */
    /* non-confocal pairs (the reference has only Python prototypes of it:
     * transient_rendering_python/rendering.py:8-93, mesh_optimization/rendering.py:739-797).
     * NULL = confocal.  Otherwise measurement l is the pair (laser origin[l], sensor[l]): the
     * surface point must be the closest hit seen from BOTH wall points, the path length is
     * d1 + d2, the form factor is the product of the two legs' clamped form factors; v2
     * conventions otherwise, so sensor == origin reproduces the confocal rows.  TRANSIENT and
     * GRADIENT modes; with use_ggx the BRDF is the half-vector form D(n.h) G1(n.wa) G1(n.wb) / 4. */
    const float *sensor;        /* [L,3] or NULL */
    const float *sensor_normal; /* [L,3] (required with sensor) */
    /* SPAD jitter variant (the reference's `jitter` module: jitter/transient_and_gradient.cpp:271-355,
     * :944-969).  jitter_weight != NULL (TRANSIENT / GRADIENT modes, confocal, Lambertian): the
     * forward rows are the plain histogram convolved with jitter_weight (transient[b] = y[b +
     * jitter_offset], y the full convolution) and the gradient uses tap i -> bin floor((2h-lb)/res)
     * + i - jitter_offset with weight jitter_weight[i] and time-derivative weight jitter_grad[i];
     * refine_scale / sigma_bin are ignored. */
    const double *jitter_weight; /* [jitter_length] or NULL */
    const double *jitter_grad;   /* [jitter_length] (GRADIENT mode) */
    int32_t jitter_offset, jitter_length;
    /* stale-cache protection for reuse_bvh / reuse_visibility: the values nlos_ctx_mesh_generation() /
     * nlos_ctx_visibility_generation() returned after the render whose tree / cache is to be reused */
    int64_t mesh_generation, visibility_generation;
    /* 1: `gradient` is overwritten instead of accumulated into (GRADIENT / VERTEX_GRADIENT modes; v1's semantics,
     * stratified_transient_raytracer/stratifiedStreamedGradientRenderer.cpp:419, which GRADIENT_V1 always has).  The
     * zeroing rides in the residual kernel of the same render: an optimisation loop that calls with zero_gradient = 1
     * holds no separate fill operation per step. */
    int32_t zero_gradient;
    /* 1 (TRANSIENT mode, confocal): the v1 NON-streamed forward body of renderTransient
     * (stratified_transient_raytracer/stratifiedTransientRenderer.cpp:96-124): path length, bin, shading normal and albedo
     * come from the SAMPLED point of the stratified map, not from the hit the intersector reports back (the two differ by
     * the rounding of the hit's barycentrics, i.e. a sample within ~1e-7 of a bin edge changes bins).  Runs on the BVH
     * back-end; nlos_v1_render_transient sets it together with clamp = 0. */
    int32_t v1_sampled_point;
    /* Global index of origin[l] = source_offset + l * source_stride (0 or 1: a contiguous block, the reference's own
     * batching, exp_bunny/test.py:66-67).  N > 1 with source_offset = rank: the STRIDED partition of an N-way
     * split (l = rank mod N) -- every rank then holds an even sample of the wall instead of one corner of it, which
     * balances the ranks (sources under the object are the slowest).  RNG keys are global, so the union of the
     * shards' rows and the sum of their gradients do not depend on the partition. */
    int32_t source_stride;
    /* 1: every source draws the SAME sample points -- the RNG key of sample s of face f is (source_offset * F + f) * spt + s
     * for all of them (the product below is defined on such samples; also valid for plain confocal / pair renders). */
    int32_t shared_samples;
    /* Row N as a PRODUCT (north_star's L x S x T histogram): n_sensors > 0 makes `sensor` / `sensor_normal` arrays of
     * n_sensors wall points of their own, and the measurements the L * n_sensors (laser, sensor) combinations:
     * transient, data, weight (and residual rows) are [L, n_sensors, T], measurement (l, j) = pair (origin[l], sensor[j])
     * of row N on shared samples (shared_samples is implied).  The reference has no native counterpart (prototype
     * formulas: transient_rendering_python/mesh_optimization/rendering.py:739-797); the product is DEFINED as its
     * pairs: results equal nlos_render of the L * n_sensors enumerated pairs with shared_samples = 1 up to fp64
     * summation order.  With face normals, no albedo, Lambertian, 64 <= F <= 6200, spt <= 32 the work is O(L + S) grid
     * passes (one visibility + geometry record per wall point) plus one combine kernel over the pairs, instead of two
     * grid passes per pair; anything else is rendered as the enumerated pairs.  TRANSIENT and GRADIENT modes.
     * gradient normalisation: 1 / (total_sources * n_sensors).  product_pairs = 1 (diagnostic): always enumerate. */
    int32_t n_sensors;
    int32_t product_pairs;
} nlos_render_args;

int  nlos_sizeof_render_args(void);                 /* for FFI layout checks */
/* Deferred status of the device-pointer path.  nlos_render() cannot validate face indices on the host: the scene
 * build reads an out-of-range index as vertex 0 (nothing faults), raises a flag, and the flag travels to the host
 * behind the build.  It is reported -- as NLOS_ERR_ARG, once -- by the first of: a later nlos_render() that finds it
 * arrived (that render returns before it enqueues anything: ITS outputs are untouched, and the mesh at fault is the
 * one of an EARLIER call on this context), nlos_ctx_check(), nlos_ctx_timing_mean(), nlos_ctx_last_path(count = 1).
 * The render that read the bad mesh itself returns NLOS_OK with outputs computed as if the index were 0.  Callers
 * that cannot vouch for their indices call nlos_ctx_check() after the render (it synchronises).  A stream-captured
 * render never reports (the graph replays without the host looking on). */
void nlos_render_args_init(nlos_render_args *a);   /* zero + defaults (clamp=1, normal_term=-1, refine=1, sigma_bin=1) */

/* Enqueue one render on `stream` (hipStream_t as void*; NULL = default stream).
 * Asynchronous: returns after the launches are queued. */
int nlos_render(nlos_ctx *ctx, const nlos_render_args *args, void *stream);

/* Every scene build gives the ctx a new mesh generation, every pass 1 that records the visibility cache a
 * new visibility generation (0 = the ctx holds none).  A caller that wants a later render to reuse either
 * (reuse_bvh / reuse_visibility) reads the generation after the render that produced it and hands it back;
 * any build or pass 1 in between makes the reuse fail instead of silently contracting against another
 * mesh's tree or visibility. */
int64_t nlos_ctx_mesh_generation(const nlos_ctx *ctx);
int64_t nlos_ctx_visibility_generation(const nlos_ctx *ctx);

/* Which kernels the last nlos_render on this ctx took, and why (nothing falls back silently). */
enum {
    NLOS_PATH_NONE       = 0,
    NLOS_PATH_GRID       = 1,   /* per-source perspective grid, one workgroup per source */
    NLOS_PATH_TILED_GRID = 2,   /* one workgroup per (source, slope-space tile) */
    NLOS_PATH_BVH        = 3    /* stackless BVH traversal per ray packet (5-6x slower on the bunny) */
};
enum {
    NLOS_REASON_NONE          = 0,
    NLOS_REASON_FORCED        = 1,  /* force_bvh = 1 */
    NLOS_REASON_TINY_MESH     = 2,  /* F < 64 */
    NLOS_REASON_LDS           = 3,  /* rows + cell tables leave no room for the cell lists */
    NLOS_REASON_TILE_LIMITS   = 4,  /* more than 1024 tiles, or the tile tables do not fit LDS */
    NLOS_REASON_GGX_PAIRS     = 5,  /* non-confocal pairs in a mode the grid passes do not carry (per-face intensity) */
    NLOS_REASON_LARGE_MESH    = 6,  /* F beyond the single-workgroup grid: tiled */
    NLOS_REASON_WINDOW_RANGE  = 7   /* resolution outside [2^-30, 2^30] or a bound beyond 2^29: outside the range on which the
                                       grid trace's lean division is the IEEE one (csrc/nlos_device.h) */
};
typedef struct nlos_path_info {
    int32_t backend;            /* NLOS_PATH_* of pass 1 (NONE if pass 1 was skipped) */
    int32_t reason;             /* NLOS_REASON_* : why not NLOS_PATH_GRID */
    int32_t grid_R;             /* cells per side of the (tile's) grid */
    int32_t tiles, tile_cap;    /* tiled grid: tiles per source, triangle capacity of a tile subset */
    int32_t chunks;             /* source chunks pass 1 was split into (tile scratch bounded to 32 GB) */
    int32_t rows_in_lds;        /* 1: histogram rows accumulated in LDS, 0: global atomics */
    int32_t gradient_kernel;    /* 0 none, 1 source-major with LDS accumulator, 2 source-major with global
                                   atomics, 3 face-major */
    /* per-workgroup outcomes of the grid launches (sources, or (source, tile) pairs); -1 unless counted */
    int64_t workgroups, coarsened, big_lds, bvh_queries;
    /* what pass 1 did with the L * F * spt surface samples of the render (summed over its sources); -1 unless counted:
     * single-workgroup grid with item masks only (confocal and pairs, spt <= 32).  `rays_traced`: samples that hit their
     * own face inside the time window with a non-zero form factor and went through the occlusion query (faces the wall
     * point sees from behind are dropped before sampling: exact under the clamped form factor); `samples_accepted`: the
     * ones found visible and binned. */
    int64_t rays_traced, samples_accepted;
} nlos_path_info;
/* count_workgroups != 0 synchronises the device and fills the counters from the launch's flags and the item-mask headers:
 * `coarsened` restarted on a coarser grid inside the kernel, `big_lds` were redone by the second launch
 * with the whole CU's LDS, `bvh_queries` traced their rays through the in-kernel BVH query (scene not
 * strictly in front of the wall point, or cell lists that fit nowhere). */
int nlos_ctx_last_path(nlos_ctx *ctx, nlos_path_info *out, int count_workgroups);
/* Deferred device-side status of earlier renders on this ctx (face index out of range seen by the scene
 * build): nlos_render checks it without synchronising whenever the flag has already arrived; this call
 * synchronises the device and checks it now.  NLOS_OK or NLOS_ERR_ARG. */
int nlos_ctx_check(nlos_ctx *ctx);

/* closest-hit batch (row E) on device pointers: out3 [N,3] and/or out1 [N] */
int nlos_intersect(nlos_ctx *ctx, const float *origins, const float *dirs, int n_rays,
                   const float *vertices, int V, const int32_t *faces, int F,
                   float *out3, float *out1, void *stream);

/* mesh regularisers on device pointers: face_affinity NULL -> area gradient (value may be NULL),
 * else normal smoothing with *value (device, [1]) overwritten.  gradient [V,3] is overwritten. */
int nlos_mesh_regulariser(nlos_ctx *ctx, const float *vertices, int V, const int32_t *faces, int F,
                          const int32_t *face_affinity, double *gradient, double *value,
                          int overwrite, void *stream);

/* ---- the steps that follow the render in every loop of the reference, on device pointers ----
 * Adam_Modified (exp_bunny/adam_modified.py:62-107): Adam with ONE denominator per row, the mean of
 * sqrt(exp_avg_sq) + eps over the row's columns.  params / exp_avg / exp_avg_sq (/ max_exp_avg_sq
 * for amsgrad, else NULL) are float32 [rows, cols]; the gradient is the renderer's float64 output
 * (grad_f64, narrowed like `torch.from_numpy(grad).float()`, exp_bunny/test.py:212-213) or a float32
 * array (grad_f32); exactly one of the two.  step is the 1-based step count.  row_mask (uint8
 * [rows], NULL = all) leaves rows with 0 untouched: the reference keeps two parameter groups with
 * different learning rates over disjoint vertex sets (exp_bunny/test.py:56-60).  cols <= 8. */
int nlos_adam_modified_step(nlos_ctx *ctx, float *params, const double *grad_f64, const float *grad_f32,
                            float *exp_avg, float *exp_avg_sq, float *max_exp_avg_sq,
                            const uint8_t *row_mask, int rows, int cols, int step, double lr,
                            double beta1, double beta2, double eps, double weight_decay, void *stream);
/* create_weighting_function (exp_bunny/rendering.py:208-217): weight = (data/max(data) + 0.1)^gamma,
 * rescaled to sum to rows*cols.  data, weight: float64 [rows, cols]. */
int nlos_create_weighting(nlos_ctx *ctx, const double *data, int rows, int cols, double gamma,
                          double *weight, void *stream);
/* L1 term of evaluate_loss_with_* (exp_bunny/rendering.py:360-364):
 * out[0] = sum weight * (transient - data)^2 / rows  (weight NULL -> 1). */
int nlos_weighted_l2(nlos_ctx *ctx, const double *transient, const double *data, const double *weight,
                     int rows, int cols, double *out, void *stream);

/* diagnostics: copy internal scratch of the last render to HOST memory (synchronises the device).
 * what = 0: the visibility cache, uint32 [L, words, F] in Morton-sorted face order (needs
 * keep_visibility or a gradient mode); what = 1: int32 [F] original face id of each sorted slot;
 * what = 2: int32 [L] path code of every source in the last single-workgroup grid launch (F <= 6200):
 * 0 = normal, 0x100 + R = cell lists overflowed and the source was redone on a grid coarsened to R x R,
 * 1 = redone with the whole CU's LDS.
 * Returns the number of bytes copied (<= max_bytes) or a negative status. */
int64_t nlos_ctx_debug_read(nlos_ctx *ctx, int what, void *host_out, int64_t max_bytes);

/* number of bins the reference computes in float32: ceil((ub-lb)/res) */
int nlos_num_bins(float lower_bound, float upper_bound, float resolution);

/* timing of the kernels of the last nlos_render on this ctx, measured with HIP
 * events on the launch stream (ms): [0]=bvh build, [1]=forward, [2]=residual,
 * [3]=gradient; valid after the stream is synchronised.  enable first. */
void nlos_ctx_enable_timing(nlos_ctx *ctx, int enable);
int  nlos_ctx_last_timing(nlos_ctx *ctx, float *ms4);
/* the event sets live in a ring (256 renders), so a timed loop needs no host sync:
 * reset before the loop, read the per-stage means after the final synchronise */
void nlos_ctx_timing_reset(nlos_ctx *ctx);
int  nlos_ctx_timing_mean(nlos_ctx *ctx, float *ms4, int *count);

#ifdef __cplusplus
}
#endif
#endif
