/*
 * nlos_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * See nlos_oracle.h for scope, citations and parity status ("parity unpinned"
 * against the original Embree binaries; pinned by tests/golden fixtures).
 *
 * Numeric contract (shared, by independent implementation, with the HIP path):
 *   - all per-sample math in IEEE fp32, no contraction by the compiler (-ffp-contract=off),
 *     correctly rounded sqrt/div; accumulation in fp64; two expressions are fused by definition,
 *     written as explicit fmaf() here and __fmaf_rn() on the device:
 *   - dot(a,b)   = fma(a.z, b.z, fma(a.y, b.y, a.x*b.x))
 *   - a*v1 + b*v2 + c*v3 = fma(c, v3, fma(b, v2, a*v1)), per component
 *   - cross(a,b) = (fma(a.y, b.z, -(a.z*b.y)), fma(a.z, b.x, -(a.x*b.z)), fma(a.x, b.y, -(a.y*b.x)))
 *   - triangle test = Embree 3 Moeller-Trumbore (published algorithm,
 *     kernels/geometry/triangle_intersector_moeller.h), restated in tri_test();
 *     Embree's rcp()/rsqrt() Newton estimates are replaced by IEEE 1/x, 1/sqrt;
 *   - closest hit = smallest t over all faces, ties -> lowest face index;
 *   - pow(h,5), pow(h,4) are evaluated as (h*h)*(h*h)*h and (h*h)*(h*h).
 * Documented deviations from the reference: out-of-range
 * bins/taps are skipped (Q3), degenerate (zero-area) faces contribute nothing,
 * v1 index typos are not reproduced (Q5), intensity adds are race-free (Q10),
 * GGX eval_nwdiff early-outs yield zero vectors (uninitialised in reference).
 */
#include "nlos_oracle.h"
#include "../include/nlos_contract.h"   /* NLOS_GRAZE_RATIO: the one constant both sides compile in */

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* threads > 0: that many OpenMP threads; 0: the host's default again (omp_set_num_threads is sticky: a caller that asked
 * for one thread must not leave every later "default" call single-threaded -- the CPU suite ran 5x longer for that) */
static void oracle_set_threads(int threads) {
#ifdef _OPENMP
    static int default_threads = 0;
    if (default_threads == 0) default_threads = omp_get_max_threads();
    omp_set_num_threads(threads > 0 ? threads : default_threads);
#else
    (void)threads;
#endif
}

/* ------------------------------------------------------------------ vectors */
typedef struct { float x, y, z; } v3;

static inline v3 mk(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 ld3(const float *p) { return mk(p[0], p[1], p[2]); }
static inline v3 add3(v3 a, v3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 neg3(v3 a) { return mk(-a.x, -a.y, -a.z); }
static inline v3 scl3(v3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
/* Numeric contract: one multiplication and two fused multiply-adds, in this order (the device's dot() is the same
 * expression); everything not written as fmaf() is a plain IEEE operation (-ffp-contract=off). */
static inline float dot3(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline v3 cross3(v3 a, v3 b) {
    return mk(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
/* u*a + v*b + w*c, the reference's "u * v1 + v * v2 + w * v3" */
static inline v3 bary3(float u, v3 a, float v, v3 b, float w, v3 c) {
    return mk(fmaf(w, c.x, fmaf(v, b.x, u * a.x)), fmaf(w, c.y, fmaf(v, b.y, u * a.y)), fmaf(w, c.z, fmaf(v, b.z, u * a.z)));
}
static inline float comp(v3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

/* ---------------------------------------------------------------------- RNG */
/* Row R (semantics only): the reference draws two floats per sample from a
 * per-thread SFMT stream (STR/rng_sse.h:19-61, STR/sampler.cpp:20-34), which is
 * not reproducible under TBB scheduling.  Replaced by the k-th
 * output of splitmix64 seeded with `seed`; S from the low, T from the high
 * 32 bits, each mapped to [0,1) exactly as STR/rng_sse.h:33-42 does. */
static inline float u32_to_unit(uint32_t x) {
    union { uint32_t u; float f; } c;
    c.u = (x >> 9) | 0x3f800000u;
    return c.f - 1.0f;
}
void nlos_oracle_sample(uint64_t seed, uint64_t k, float *S, float *T) {
    uint64_t z = seed + (k + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    *S = u32_to_unit((uint32_t)(z & 0xffffffffull));
    *T = u32_to_unit((uint32_t)(z >> 32));
}

/* global index of source l of this call (source sharding: a contiguous block from source_offset, or every
 * source_stride-th source from it); the sample keys are made of it */
static inline uint64_t global_source(const nlos_oracle_opts *op, int l) {
    if (op->shared_samples) return (uint64_t)op->source_offset;
    return (uint64_t)(op->source_offset + (int64_t)l * (op->source_stride > 1 ? op->source_stride : 1));
}

void nlos_oracle_default_opts(nlos_oracle_opts *o) {
    memset(o, 0, sizeof(*o));
    o->normal_term = -1;
    o->clamp = 1;
}

int nlos_oracle_num_bins(float lb, float ub, float res) {
    /* SMO/stratifiedStreamedGradientRenderer.cpp:514-515 (float32 ceil) */
    return (int)ceilf((ub - lb) / res);
}

/* ------------------------------------------------------------ triangle test */
/* Grazing rule (numeric contract, DESIGN.md section 2, include/nlos_contract.h): a ray that meets a triangle's plane
 * at less than asin(2^-10) = 0.056 degrees does not hit it -- |ng . d| >= |ng| |d| / 1024 is part of the hit test, for
 * the sampled face and for occluders alike.  The reference (Embree) has no such rule
 * (SMO/transient_and_gradient.cpp:199-206 accepts every hit rtcIntersect1M reports).  It is what makes "closest hit
 * over ALL faces" reproducible in fp32: t = T / den and the barycentrics are only as accurate as den, so for
 * den -> 0 the test reports hits millimetres away from the ray, which no culled query (BVH slabs, depth bounds, the
 * GPU's perspective grid) can be made to follow.  With the rule the reported hit is within ~4e-4 t of the true one
 * and every conservative cull is padded by a multiple of that.  Energy-wise an excluded sample carries
 * cos^2 < 1e-6 of a frontal sample's weight.
 * nlos_oracle_set_graze_ratio(0) switches the rule off: with accel == 0 (all faces, brute force) that is the
 * reference's rule-free definition, against which tests/test_oracle.py bounds what the rule changes. */
static float g_graze_ratio = NLOS_GRAZE_RATIO;      /* gmin = ratio * area = |ng| / 1024 (area = |ng| / 2) */
void nlos_oracle_set_graze_ratio(float ratio) { g_graze_ratio = ratio < 0.0f ? NLOS_GRAZE_RATIO : ratio; }
float nlos_oracle_graze_ratio(void) { return g_graze_ratio; }

typedef struct { v3 p0, e1, e2, ng; float gmin; } tri_t;   /* e1 = p0-p1, e2 = p2-p0, ng = e2 x e1 */

static inline tri_t make_tri(v3 p0, v3 p1, v3 p2) {
    tri_t t;
    t.p0 = p0;
    t.e1 = sub3(p0, p1);
    t.e2 = sub3(p2, p0);
    t.ng = cross3(neg3(t.e1), t.e2);     /* = cross(p1 - p0, p2 - p0) bit for bit (-e1 is p1 - p0 exactly) */
    t.gmin = g_graze_ratio * (sqrtf(dot3(t.ng, t.ng)) / 2.0f);
    return t;
}

static inline float flipsign(float x, int neg) { return neg ? -x : x; }

/* Embree 3 Moeller-Trumbore, tnear = 0, tfar = inf (Row I).  Returns 1 on hit
 * and writes (t, u, v); u weights the 2nd, v the 3rd vertex
 * (SMO/transient_and_gradient.cpp:208-211). */
/* gscale: |d| for row E's unnormalised directions (the cut-off angle must not depend on the direction's length);
 * the render paths pass unit directions and gscale = 1 (x * 1.0f is exact). */
static inline int tri_test_scaled(const tri_t *tr, v3 o, v3 d, float gscale, float *t, float *u, float *v) {
    v3 c = sub3(tr->p0, o);
    v3 r = cross3(c, d);
    float den = dot3(tr->ng, d);
    float aden = fabsf(den);
    int sg = signbit(den) ? 1 : 0;
    float U = flipsign(dot3(r, tr->e2), sg);
    float Vv = flipsign(dot3(r, tr->e1), sg);
    if (!(den != 0.0f)) return 0;
    if (!(U >= 0.0f)) return 0;
    if (!(Vv >= 0.0f)) return 0;
    if (!(U + Vv <= aden)) return 0;
    float Tn = flipsign(dot3(tr->ng, c), sg);
    if (!(0.0f < Tn)) return 0;           /* absDen*tnear < T with tnear = 0 */
    if (!(aden >= tr->gmin * gscale)) return 0;    /* grazing rule */
    float rcp = 1.0f / aden;
    *u = U * rcp;
    *v = Vv * rcp;
    *t = Tn * rcp;
    return 1;
}

static inline int tri_test(const tri_t *tr, v3 o, v3 d, float *t, float *u, float *v) {
    return tri_test_scaled(tr, o, d, 1.0f, t, u, v);
}

/* -------------------------------------------------------------------- scene */
typedef struct { float lo[3], hi[3]; int left, right, first, count; } bnode_t;

typedef struct {
    int nF, nV;
    const float *V;
    const int32_t *F;
    tri_t *tris;            /* [nF] */
    /* BVH (accel == 1) */
    bnode_t *nodes; int n_nodes;
    int *order;             /* leaf triangle order */
} scene_t;

static void tri_bounds(const scene_t *sc, int f, float lo[3], float hi[3]) {
    for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; }
    for (int k = 0; k < 3; ++k) {
        const float *p = sc->V + 3 * (size_t)sc->F[3 * f + k];
        for (int a = 0; a < 3; ++a) { if (p[a] < lo[a]) lo[a] = p[a]; if (p[a] > hi[a]) hi[a] = p[a]; }
    }
}

typedef struct { float key; int id; } keyid_t;
static int cmp_keyid(const void *a, const void *b) {
    const keyid_t *x = (const keyid_t *)a, *y = (const keyid_t *)b;
    if (x->key < y->key) return -1;
    if (x->key > y->key) return 1;
    return (x->id > y->id) - (x->id < y->id);
}

static int bvh_build_rec(scene_t *sc, int first, int count, const float *cent, float pad) {
    int me = sc->n_nodes++;
    bnode_t *n = &sc->nodes[me];
    float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int a = 0; a < 3; ++a) { n->lo[a] = INFINITY; n->hi[a] = -INFINITY; }
    for (int i = first; i < first + count; ++i) {
        float lo[3], hi[3];
        int f = sc->order[i];
        tri_bounds(sc, f, lo, hi);
        for (int a = 0; a < 3; ++a) {
            if (lo[a] < n->lo[a]) n->lo[a] = lo[a];
            if (hi[a] > n->hi[a]) n->hi[a] = hi[a];
            float c = cent[3 * f + a];
            if (c < clo[a]) clo[a] = c;
            if (c > chi[a]) chi[a] = c;
        }
    }
    for (int a = 0; a < 3; ++a) { n->lo[a] -= pad; n->hi[a] += pad; }
    n->first = first; n->count = count; n->left = n->right = -1;
    if (count <= 4) return me;
    int ax = 0;
    if (chi[1] - clo[1] > chi[ax] - clo[ax]) ax = 1;
    if (chi[2] - clo[2] > chi[ax] - clo[ax]) ax = 2;
    keyid_t *tmp = (keyid_t *)malloc(sizeof(keyid_t) * (size_t)count);
    for (int i = 0; i < count; ++i) { tmp[i].id = sc->order[first + i]; tmp[i].key = cent[3 * tmp[i].id + ax]; }
    qsort(tmp, (size_t)count, sizeof(keyid_t), cmp_keyid);
    for (int i = 0; i < count; ++i) sc->order[first + i] = tmp[i].id;
    free(tmp);
    int half = count / 2;
    int l = bvh_build_rec(sc, first, half, cent, pad);
    int r = bvh_build_rec(sc, first + half, count - half, cent, pad);
    sc->nodes[me].left = l; sc->nodes[me].right = r; sc->nodes[me].count = 0;
    return me;
}

static int scene_init(scene_t *sc, const float *V, int nV, const int32_t *F, int nF, int accel) {
    memset(sc, 0, sizeof(*sc));
    sc->nF = nF; sc->nV = nV; sc->V = V; sc->F = F;
    for (int i = 0; i < 3 * nF; ++i) if (F[i] < 0 || F[i] >= nV) return -1;
    sc->tris = (tri_t *)malloc(sizeof(tri_t) * (size_t)(nF > 0 ? nF : 1));
    for (int f = 0; f < nF; ++f)
        sc->tris[f] = make_tri(ld3(V + 3 * (size_t)F[3 * f]), ld3(V + 3 * (size_t)F[3 * f + 1]),
                               ld3(V + 3 * (size_t)F[3 * f + 2]));
    if (accel && nF > 0) {
        float *cent = (float *)malloc(sizeof(float) * 3 * (size_t)nF);
        float slo[3] = {INFINITY, INFINITY, INFINITY}, shi[3] = {-INFINITY, -INFINITY, -INFINITY};
        sc->order = (int *)malloc(sizeof(int) * (size_t)nF);
        for (int f = 0; f < nF; ++f) {
            float lo[3], hi[3];
            tri_bounds(sc, f, lo, hi);
            for (int a = 0; a < 3; ++a) {
                cent[3 * f + a] = 0.5f * (lo[a] + hi[a]);
                if (lo[a] < slo[a]) slo[a] = lo[a];
                if (hi[a] > shi[a]) shi[a] = hi[a];
            }
            sc->order[f] = f;
        }
        float ext = 0.0f;
        for (int a = 0; a < 3; ++a) {
            float e = fmaxf(fabsf(slo[a]), fabsf(shi[a]));
            if (e > ext) ext = e;
        }
        float pad = 1.6e-3f * ext + 1e-30f;   /* conservative: >> fp32 rounding of hit points, > the grazing-rule error */
        sc->nodes = (bnode_t *)malloc(sizeof(bnode_t) * (size_t)(2 * nF));
        sc->n_nodes = 0;
        bvh_build_rec(sc, 0, nF, cent, pad);
        free(cent);
    }
    return 0;
}

static void scene_free(scene_t *sc) {
    free(sc->tris); free(sc->nodes); free(sc->order);
    memset(sc, 0, sizeof(*sc));
}

typedef struct { int prim; float t, u, v; } hit_t;

static inline void hit_update(hit_t *h, int f, float t, float u, float v) {
    if (h->prim < 0 || t < h->t || (t == h->t && f < h->prim)) {
        h->prim = f; h->t = t; h->u = u; h->v = v;
    }
}

static hit_t closest_brute_scaled(const scene_t *sc, v3 o, v3 d, float gscale) {
    hit_t h; h.prim = -1; h.t = INFINITY; h.u = h.v = 0.0f;
    for (int f = 0; f < sc->nF; ++f) {
        float t, u, v;
        if (tri_test_scaled(&sc->tris[f], o, d, gscale, &t, &u, &v)) hit_update(&h, f, t, u, v);
    }
    return h;
}
static hit_t closest_brute(const scene_t *sc, v3 o, v3 d) { return closest_brute_scaled(sc, o, d, 1.0f); }

/* conservative slab test against padded boxes; never culls a box that holds a
 * triangle whose tri_test() hit has t <= tmax */
static inline int box_hit(const bnode_t *n, const float o[3], const float inv[3], float tmax) {
    float t0 = 0.0f, t1 = tmax;
    for (int a = 0; a < 3; ++a) {
        float ta = (n->lo[a] - o[a]) * inv[a];
        float tb = (n->hi[a] - o[a]) * inv[a];
        float tn = fminf(ta, tb), tf = fmaxf(ta, tb);   /* fmin/fmax drop NaN (0*inf) */
        tn = tn - fabsf(tn) * 1.6e-3f;                  /* > the error of t under the grazing rule (~4e-4) */
        tf = tf + fabsf(tf) * 1.6e-3f;
        if (tn > t0) t0 = tn;
        if (tf < t1) t1 = tf;
    }
    return t0 <= t1;
}

static hit_t closest_bvh_scaled(const scene_t *sc, v3 o, v3 d, float gscale) {
    hit_t h; h.prim = -1; h.t = INFINITY; h.u = h.v = 0.0f;
    if (sc->n_nodes == 0) return h;
    float of[3] = {o.x, o.y, o.z}, inv[3];
    float df[3] = {d.x, d.y, d.z};
    for (int a = 0; a < 3; ++a) inv[a] = 1.0f / df[a];   /* +-inf for zero components */
    int stack[128], sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
        const bnode_t *n = &sc->nodes[stack[--sp]];
        if (!box_hit(n, of, inv, h.t)) continue;
        if (n->left < 0) {
            for (int i = n->first; i < n->first + n->count; ++i) {
                int f = sc->order[i];
                float t, u, v;
                if (tri_test_scaled(&sc->tris[f], o, d, gscale, &t, &u, &v)) hit_update(&h, f, t, u, v);
            }
        } else {
            if (sp + 2 > 128) { /* cannot happen for median splits */ return closest_brute_scaled(sc, o, d, gscale); }
            stack[sp++] = n->left;
            stack[sp++] = n->right;
        }
    }
    return h;
}

static hit_t closest_bvh(const scene_t *sc, v3 o, v3 d) { return closest_bvh_scaled(sc, o, d, 1.0f); }

static inline hit_t closest_hit(const scene_t *sc, v3 o, v3 d, int accel) {
    return accel ? closest_bvh(sc, o, d) : closest_brute(sc, o, d);
}

/* ---------------------------------------------------------- row E functions */
int nlos_oracle_intersect(const float *origins, const float *dirs, int n_rays,
                          const float *V, int nV, const int32_t *F, int nF,
                          float *out3, float *out1, int accel, int threads) {
    scene_t sc;
    if (scene_init(&sc, V, nV, F, nF, accel)) return -1;
#ifdef _OPENMP
    oracle_set_threads(threads);
#endif
    (void)threads;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n_rays; ++i) {
        /* EMB/c_embree_intersector.cpp:20-45 hands the caller's direction to Embree as it is: any length */
        const v3 d = ld3(dirs + 3 * (size_t)i);
        const float dl = sqrtf(dot3(d, d));
        hit_t h = accel ? closest_bvh_scaled(&sc, ld3(origins + 3 * (size_t)i), d, dl)
                        : closest_brute_scaled(&sc, ld3(origins + 3 * (size_t)i), d, dl);
        if (out3) {
            if (h.prim < 0) out3[3 * (size_t)i] = -1.0f;
            else { out3[3 * (size_t)i] = (float)h.prim; out3[3 * (size_t)i + 1] = h.u; out3[3 * (size_t)i + 2] = h.v; }
        }
        if (out1) out1[i] = h.prim < 0 ? -1.0f : (float)h.prim;
    }
    scene_free(&sc);
    return 0;
}

void nlos_oracle_barycentric_to_world(const float *V, const int32_t *F,
                                      const float *bary, int n, float *out) {
    for (int i = 0; i < n; ++i) {
        int fid = (int)bary[3 * (size_t)i];
        if (fid < 0) continue;
        float u = bary[3 * (size_t)i + 1], v = bary[3 * (size_t)i + 2];
        int a = F[3 * fid], b = F[3 * fid + 1], c = F[3 * fid + 2];
        for (int k = 0; k < 3; ++k)
            out[3 * (size_t)i + k] = (1 - u - v) * V[3 * a + k] + u * V[3 * b + k] + v * V[3 * c + k];
    }
}

/* ----------------------------------------------------------------- GGX (B) */
static float ggx_D(float a, float nw) {
    if (nw <= 0) return 0.0f;
    float nw2 = nw * nw;
    float bex = (1.0f - nw2) / (a * a) / nw2;
    float root = (1.0f + bex) * nw2;
    float result = (float)(1.0f / (M_PI * a * a * root * root));
    if (result * nw < 1e-20f) result = 0;
    return result;
}
static float ggx_G1(float a, float nw) {
    if (nw <= 0) return 0.0f;
    if ((nw >= 1.0f) || (nw <= -1.0f)) return 1.0f;
    float root = a * a + (1.0f - a * a) * nw * nw;
    return 2.0f / (nw + sqrtf(root));
}
static float ggx_G(float a, float nw) { float g = ggx_G1(a, nw); return g * g; }
static float ggx_eval_nw(float a, float nw) {
    if (nw <= 0) return 0.0f;
    float Dv = ggx_D(a, nw);
    if (Dv == 0) return 0.0f;
    return Dv * ggx_G(a, nw) / 4.0f;
}
static float ggx_D_adiff(float a, float nw) {
    if (nw <= 0) return 0.0f;
    float nw2 = nw * nw, a2 = a * a;
    float val = a2 * nw2 - nw2 + 1;
    return (float)(-(2.0f * a * (a2 * nw2 + nw2 - 1)) / (M_PI * val * val * val));
}
static float ggx_G1_adiff(float a, float nw) {
    if (nw <= 0) return 0.0f;
    if ((nw >= 1.0f) || (nw <= -1.0f)) return 0.0f;
    float nw2 = nw * nw;
    float val = sqrtf(a * a - nw2 * (a * a - 1));
    float root = nw + val;
    return 2.0f * a * (nw2 - 1.0f) / (val * root * root);
}
static float ggx_eval_adiff_nw(float a, float nw) {
    if (nw <= 0) return 0.0f;
    float Dv = ggx_D(a, nw);
    if (Dv == 0) return 0.0f;
    float Gv = ggx_G(a, nw);
    float Dp = ggx_D_adiff(a, nw);
    float Gp = 2.0f * ggx_G1_adiff(a, nw) * ggx_G1(a, nw);
    return (Dp * Gv + Gp * Dv) / 4.0f;
}
static float ggx_D_ndiff(float a, float nw) {
    if (nw <= 0) return 0.0f;
    float nw2 = nw * nw, a2 = a * a;
    float root = (a2 - 1.0f) * nw2 + 1.0f;
    return (float)(-(4.0f * a2 * nw * (a2 - 1.0f)) / (M_PI * root * root * root));
}
static float ggx_G1_ndiff(float a, float nw) {
    if (nw <= 0) return 0.0f;
    if ((nw >= 1.0f) || (nw <= -1.0f)) return 0.0f;
    float nw2 = nw * nw, a2 = a * a;
    float temp = sqrtf(a2 - nw2 * (a2 - 1.0f));
    float root = nw + temp;
    return -2.0f * (1.0f - (nw * (a2 - 1.0f)) / temp) / root / root;
}
static float ggx_eval_nwsdiff_nw(float a, float nw) {
    if (nw <= 0) return 0.0f;
    float Dv = ggx_D(a, nw);
    if (Dv == 0) return 0.0f;
    float Gv = ggx_G(a, nw);
    float Gp = 2.0f * ggx_G1_ndiff(a, nw) * ggx_G1(a, nw);
    float Dp = ggx_D_ndiff(a, nw);
    return (Dp * Gv + Gp * Dv) / 4.0f;
}
/* GGX for a (laser, sensor) pair -- row N with the GGX branch.  The reference has neither a kernel nor a prototype
 * for it; the definition is the half-vector form of the same microfacet model,
 *     brdf(n, wa, wb) = D(n.h) G1(n.wa) G1(n.wb) / 4,   h = (wa + wb) / |wa + wb|,
 * with ggx_confocal.cpp's early-outs (a cosine <= 0, D below the cut-off), which is eval() for wa == wb.
 * Derivatives with respect to wa, wb (unconstrained) and n, from D' = dD/d(n.h) and G1' = dG1/d(n.w)
 * (ggx_confocal.cpp:113-150, :176-232). */
typedef struct { float brdf; v3 ga, gb, gn; } ggx_pair_t;
static void ggx_pair(float a, v3 n, v3 wa, v3 wb, int want_grad, ggx_pair_t *o) {
    o->brdf = 0.0f; o->ga = o->gb = o->gn = mk(0, 0, 0);
    float na = dot3(n, wa), nb = dot3(n, wb);
    if (na <= 0 || nb <= 0) return;
    v3 hv = add3(wa, wb);
    float hl = sqrtf(dot3(hv, hv));
    if (!(hl > 0.0f)) return;
    v3 hn = scl3(hv, 1.0f / hl);
    float nh = dot3(n, hn);
    if (nh <= 0) return;
    float Dv = ggx_D(a, nh);
    if (Dv == 0) return;
    float Ga = ggx_G1(a, na), Gb = ggx_G1(a, nb);
    o->brdf = Dv * Ga * Gb / 4.0f;
    if (!want_grad) return;
    float cD = ggx_D_ndiff(a, nh) * Ga * Gb / 4.0f;
    float cA = Dv * ggx_G1_ndiff(a, na) * Gb / 4.0f;
    float cB = Dv * Ga * ggx_G1_ndiff(a, nb) / 4.0f;
    v3 dnh = scl3(sub3(n, scl3(hn, nh)), 1.0f / hl);          /* d(n.h)/dwa = d(n.h)/dwb */
    o->ga = add3(scl3(dnh, cD), scl3(n, cA));
    o->gb = add3(scl3(dnh, cD), scl3(n, cB));
    o->gn = add3(add3(scl3(hn, cD), scl3(wa, cA)), scl3(wb, cB));
}

float nlos_oracle_ggx_eval(float alpha, const float *n, const float *w) {
    return ggx_eval_nw(alpha, dot3(ld3(n), ld3(w)));
}
float nlos_oracle_ggx_eval_adiff(float alpha, const float *n, const float *w) {
    return ggx_eval_adiff_nw(alpha, dot3(ld3(n), ld3(w)));
}
float nlos_oracle_ggx_eval_nwsdiff(float alpha, const float *n, const float *w) {
    return ggx_eval_nwsdiff_nw(alpha, dot3(ld3(n), ld3(w)));
}

/* ------------------------------------------------------- per-(l,f) context */
typedef struct {
    v3 o, on;               /* source point and wall normal */
    v3 p0, p1, p2;          /* "v1,v2,v3" of the reference */
    v3 fn; float area;      /* face normal, face area */
    v3 n0, n1, n2; int has_vn;
    float a0, a1, a2; int has_alb;
    int i0, i1, i2;
    int degenerate;
} task_t;

static void task_setup(task_t *t, const scene_t *sc, const float *origin, const float *normal,
                       int l, int f, const float *vnormal, const float *albedo) {
    t->o = ld3(origin + 3 * (size_t)l);
    t->on = ld3(normal + 3 * (size_t)l);
    t->i0 = sc->F[3 * f]; t->i1 = sc->F[3 * f + 1]; t->i2 = sc->F[3 * f + 2];
    t->p0 = ld3(sc->V + 3 * (size_t)t->i0);
    t->p1 = ld3(sc->V + 3 * (size_t)t->i1);
    t->p2 = ld3(sc->V + 3 * (size_t)t->i2);
    /* SMO/transient_and_gradient.cpp:157-159 */
    v3 nr = cross3(sub3(t->p1, t->p0), sub3(t->p2, t->p0));
    t->area = sqrtf(dot3(nr, nr)) / 2.0f;
    t->degenerate = !(t->area > 0.0f);
    t->fn = scl3(nr, 1.0f / (2.0f * t->area));
    t->has_vn = vnormal != NULL;
    t->n0 = t->n1 = t->n2 = mk(0, 0, 1);
    if (vnormal) {
        t->n0 = ld3(vnormal + 3 * (size_t)t->i0);
        t->n1 = ld3(vnormal + 3 * (size_t)t->i1);
        t->n2 = ld3(vnormal + 3 * (size_t)t->i2);
    }
    t->has_alb = albedo != NULL;
    t->a0 = t->a1 = t->a2 = 1.0f;
    if (albedo) { t->a0 = albedo[t->i0]; t->a1 = albedo[t->i1]; t->a2 = albedo[t->i2]; }
}

/* Row S: stratified sample -> ray direction (SMO/transient_and_gradient.cpp:178-196) */
static inline v3 sample_dir(const task_t *t, float S, float T) {
    float sq = sqrtf(T);
    float u = 1 - sq;
    float v = (1 - S) * sq;
    float w = S * sq;
    v3 p = bary3(u, t->p0, v, t->p1, w, t->p2);
    v3 d = sub3(p, t->o);
    float rs = 1.0f / sqrtf(dot3(d, d));
    return scl3(d, rs);
}

/* accepted-hit geometry shared by all kernels (SMO/...:206-223) */
typedef struct { float u, v, w, h; v3 dir, n; float alb; } geo_t;

static inline int accept_sample_ex(const task_t *t, const scene_t *sc, int f, int accel,
                                   float S, float T, float lb, float ub, geo_t *g, int sampled_point) {
    v3 dir = sample_dir(t, S, T);
    hit_t h = closest_hit(sc, t->o, dir, accel);
    if (h.prim != f) return 0;
    g->v = h.u; g->w = h.v;
    g->u = 1.0f - g->v - g->w;
    if (sampled_point) {
        /* STR/stratifiedTransientRenderer.cpp:91-101,108-124: halfLength = |point - origin| of the sampled point, and
         * u, v, w of the sample map weight the normals / albedos */
        float sq = sqrtf(T);
        g->u = 1 - sq; g->v = (1 - S) * sq; g->w = S * sq;
    }
    v3 p = bary3(g->u, t->p0, g->v, t->p1, g->w, t->p2);
    v3 d = sub3(p, t->o);
    g->h = sqrtf(dot3(d, d));
    if (!((g->h <= ub / 2.0f) && (g->h >= lb / 2.0f))) return 0;
    g->dir = dir;
    g->n = t->fn;
    if (t->has_vn) g->n = bary3(g->u, t->n0, g->v, t->n1, g->w, t->n2);
    g->alb = 1.0f;
    if (t->has_alb) g->alb = g->u * t->a0 + g->v * t->a1 + g->w * t->a2;
    return 1;
}
static inline int accept_sample(const task_t *t, const scene_t *sc, int f, int accel,
                                float S, float T, float lb, float ub, geo_t *g) {
    return accept_sample_ex(t, sc, f, accel, S, T, lb, ub, g, 0);
}

static inline float emax0(float x) { return 0.0f < x ? x : 0.0f; }   /* embree::max(0.f, x) */

/* ------------------------------------------------------------------ forward */
/* Row F: streamedRayTraceTriangle (SMO/transient_and_gradient.cpp:122-237),
 * GGX variant GGX/transient_and_gradient.cpp:236-237, v1 (no clamp)
 * STR/stratifiedStreamedTransientRenderer.cpp:130-137. */
static void forward_task(const scene_t *sc, const float *origin, const float *normal,
                         const float *vnormal, const float *albedo, int l, int f,
                         float lb, float ub, float res, int spt, int nbins, double *row,
                         const nlos_oracle_opts *op) {
    task_t t;
    task_setup(&t, sc, origin, normal, l, f, vnormal, albedo);
    if (t.degenerate) return;
    uint64_t kbase = ((global_source(op, l)) * (uint64_t)sc->nF + (uint64_t)f) * (uint64_t)spt;
    for (int s = 0; s < spt; ++s) {
        float S, T;
        geo_t g;
        nlos_oracle_sample(op->seed, kbase + (uint64_t)s, &S, &T);
        if (!accept_sample_ex(&t, sc, f, op->accel, S, T, lb, ub, &g, op->sampled_point)) continue;
        float ff = -dot3(g.n, g.dir) * dot3(t.on, g.dir) / g.h / g.h;
        if (op->clamp) ff = emax0(ff);
        int bin = (int)floorf((2.0f * g.h - lb) / res);
        if (bin < 0 || bin >= nbins) continue;                 /* deviation Q3 */
        float val = t.area * g.alb * ff * ff;
        if (op->use_ggx) val = val * ggx_eval_nw(op->ggx_alpha, dot3(g.n, neg3(g.dir)));
        row[bin] += (double)val / (double)spt;
    }
}

static void gauss_kernel(double *k, int refine, int sigma_bin, float res) {
    /* SMO/transient_and_gradient.cpp:350-355 / :538-544 */
    int K = 4 * refine * sigma_bin + 1;
    double sigma = res * sigma_bin / 2.355;
    double normalization = 1 / sigma / sqrt(2 * M_PI) * res / refine;
    for (int i = 0; i < K; ++i) {
        double t = (-2 * refine * sigma_bin + i) * res / refine / sigma;
        k[i] = exp(-(t * t) / 2) * normalization;
    }
}

/* Row FD: render_smoothed_transients (SMO/transient_and_gradient.cpp:271-376) */
static int forward_driver(const scene_t *sc, const float *origin, int L, const float *normal,
                          const float *vnormal, const float *albedo, int num_samples,
                          float lb, float ub, float res, int nbins, int refine, int sigma_bin,
                          double *transient, const nlos_oracle_opts *op) {
    const int nF = sc->nF;
    const int spt = 1 + ((num_samples - 1) / nF);
    const int rb = nbins * refine;
    memset(transient, 0, sizeof(double) * (size_t)L * (size_t)nbins);
    int nth = 1;
#ifdef _OPENMP
    oracle_set_threads(op->threads);
    nth = omp_get_max_threads();
#endif
    /* per-thread private histograms like the reference (:301-316) */
    double *priv = (double *)calloc((size_t)nth * (size_t)L * (size_t)rb, sizeof(double));
    if (!priv) return -2;
    const float res_r = res / refine;
    const long long ntask = (long long)L * nF;
#pragma omp parallel
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        double *mine = priv + (size_t)tid * (size_t)L * (size_t)rb;
#pragma omp for schedule(dynamic, 64)
        for (long long idx = 0; idx < ntask; ++idx) {
            int f = (int)(idx % nF), l = (int)(idx / nF);
            forward_task(sc, origin, normal, vnormal, albedo, l, f, lb, ub, res_r, spt, rb,
                         mine + (size_t)l * (size_t)rb, op);
        }
    }
    if (refine <= 1) {
        for (int l = 0; l < L; ++l)
            for (int th = 0; th < nth; ++th)
                for (int b = 0; b < nbins; ++b)
                    transient[(size_t)l * nbins + b] += priv[((size_t)th * L + l) * (size_t)nbins + b];
        free(priv);
        return 0;
    }
    /* refined histogram -> Gaussian -> fold (:348-371); MKL full convolution
     * (convolution_mkl.cpp:3-11) restated as a direct sum */
    const int K = 4 * refine * sigma_bin + 1;
    double *kern = (double *)malloc(sizeof(double) * (size_t)K);
    double *fine = (double *)malloc(sizeof(double) * (size_t)rb);
    double *y = (double *)malloc(sizeof(double) * (size_t)(rb + K - 1));
    gauss_kernel(kern, refine, sigma_bin, res);
    for (int l = 0; l < L; ++l) {
        memset(fine, 0, sizeof(double) * (size_t)rb);
        for (int th = 0; th < nth; ++th)
            for (int b = 0; b < rb; ++b) fine[b] += priv[((size_t)th * L + l) * (size_t)rb + b];
        memset(y, 0, sizeof(double) * (size_t)(rb + K - 1));
        for (int i = 0; i < rb; ++i)
            for (int j = 0; j < K; ++j) y[i + j] += fine[i] * kern[j];
        for (int b = 0; b < rb; ++b)
            transient[(size_t)l * nbins + b / refine] += y[b + 2 * refine * sigma_bin];
    }
    free(kern); free(fine); free(y); free(priv);
    return 0;
}

static void fill_pathlengths(double *pathlengths, int nbins, float lb, float res) {
    /* SMO/stratifiedStreamedGradientRenderer.cpp:517-520 */
    for (int i = 0; i < nbins; ++i) pathlengths[i] = (double)(lb + i * res);
}

int nlos_oracle_render_transient(const float *origin, int L, const float *normal,
                                 const float *V, int nV, const float *vnormal,
                                 const float *albedo, const int32_t *F, int nF,
                                 int num_samples, float lb, float ub, float res,
                                 double *transient, double *pathlengths,
                                 int refine, int sigma_bin,
                                 const nlos_oracle_opts *opts) {
    nlos_oracle_opts dflt;
    if (!opts) { nlos_oracle_default_opts(&dflt); opts = &dflt; }
    if (nF <= 0 || L < 0 || refine < 1) return -1;
    scene_t sc;
    if (scene_init(&sc, V, nV, F, nF, opts->accel)) return -1;
    int nbins = nlos_oracle_num_bins(lb, ub, res);
    if (pathlengths) fill_pathlengths(pathlengths, nbins, lb, res);
    int rc = forward_driver(&sc, origin, L, normal, vnormal, albedo, num_samples, lb, ub, res,
                            nbins, refine, sigma_bin, transient, opts);
    scene_free(&sc);
    return rc;
}

/* ----------------------------------------------------------------- residual */
/* Row D: SMO/stratifiedStreamedGradientRenderer.cpp:543-550 */
static void residual(const double *data, const double *weight, const double *transient,
                     size_t n, int loss_test, double *diff) {
    for (size_t i = 0; i < n; ++i) {
        double d = data[i] - transient[i];
        if (loss_test == 1) d = 2 * d * d * d;
        diff[i] = d * weight[i];
    }
}

/* ----------------------------------------------------------------- gradient */
typedef struct {
    int K, refine, sigma_bin;
    double *w;              /* weighting_kernal[K] */
    double sigma_square;
    float res;
} taps_t;

static void taps_init(taps_t *tp, int refine, int sigma_bin, float res) {
    tp->refine = refine; tp->sigma_bin = sigma_bin; tp->res = res;
    tp->K = 4 * refine * sigma_bin + 1;
    tp->w = (double *)malloc(sizeof(double) * (size_t)tp->K);
    gauss_kernel(tp->w, refine, sigma_bin, res);
    double sigma = res * sigma_bin / 2.355;
    tp->sigma_square = sigma * sigma;
}

static inline double tap_delta(const taps_t *tp, int i) {
    /* SMO/...:973: int*float/int evaluates in float, then widens */
    float d = (-2 * tp->refine * tp->sigma_bin + i) * tp->res / tp->refine;
    return (double)d;
}

static inline int tap_bin(float h, double delta, float lb, float res) {
    /* SMO/...:975-976: float + double -> double arithmetic, double floor */
    return (int)floor((2.0f * h + delta - lb) / res);
}

/* per-sample vectors t1, t2 and intensity (SMO/...:944-966, GGX/...:750-783) */
typedef struct { v3 t1, t2; double intensity; } gvec_t;

static void grad_vectors(const task_t *t, const geo_t *g, int normal_term,
                         const nlos_oracle_opts *op, int v1_style, gvec_t *out) {
    float c2 = dot3(t->on, g->dir);
    float c3 = dot3(g->n, neg3(g->dir));
    if (c2 < 0) c2 = 0;
    if (c3 < 0) c3 = 0;
    float ff = c2 * c3 / g->h / g->h;
    float h2 = g->h * g->h, h4 = h2 * h2, h5 = h4 * g->h;
    v3 inner = add3(sub3(scl3(t->on, c3), scl3(g->n, c2)), scl3(scl3(scl3(neg3(g->dir), 4), c2), c3));
    v3 t1, gn = mk(0, 0, 0);
    if (op->use_ggx) {
        v3 wv = neg3(g->dir);
        float nw = dot3(g->n, wv);
        float brdf = ggx_eval_nw(op->ggx_alpha, nw);
        float s = ggx_eval_nwsdiff_nw(op->ggx_alpha, nw);
        v3 dn = scl3(wv, s), dw = scl3(g->n, s);
        v3 dx = add3(neg3(dw), scl3(scl3(g->dir, dot3(g->dir, dw)), 1.0f / g->h));
        out->intensity = (double)(g->alb * ff * ff * brdf);
        v3 t11 = scl3(inner, 2 * c2 * c3);
        t11 = scl3(t11, 1.0f / h5);
        t11 = scl3(t11, brdf);
        v3 t12 = scl3(dx, ff * ff);
        t1 = add3(t11, t12);
        if (normal_term) {
            gn = scl3(scl3(scl3(scl3(scl3(g->dir, -2), c3), c2), c2), brdf);
            gn = scl3(gn, 1.0f / h4);
            gn = add3(gn, scl3(dn, ff * ff));
            float ct = dot3(gn, g->n);
            gn = sub3(gn, scl3(g->n, ct));
        }
    } else {
        out->intensity = (double)(g->alb * ff * ff);
        float sc = v1_style ? (2 * c2 * c3) : (2 * g->alb * c2 * c3);
        t1 = scl3(inner, sc);
        t1 = scl3(t1, 1.0f / h5);
        if (normal_term) {
            float s0 = v1_style ? -2.0f : (-2 * g->alb);
            gn = scl3(scl3(scl3(scl3(g->dir, s0), c3), c2), c2);
            gn = scl3(gn, 1.0f / h4);
            float ct = dot3(gn, g->n);
            gn = sub3(gn, scl3(g->n, ct));
        }
    }
    v3 t2 = scl3(g->n, (float)out->intensity);
    t2 = scl3(add3(t2, gn), 1.0f / (2 * t->area));
    out->t1 = t1; out->t2 = t2;
}

/* Row G: streamedRayTraceTriangleGradient (SMO/transient_and_gradient.cpp:843-1007) */
static void gradient_task(const scene_t *sc, const float *origin, const float *normal,
                          const float *vnormal, const float *albedo, int l, int f,
                          float lb, float ub, float res, int spt, int nbins,
                          const double *diff_row, const taps_t *tp, int normal_term,
                          double *grad, const nlos_oracle_opts *op) {
    task_t t;
    task_setup(&t, sc, origin, normal, l, f, vnormal, albedo);
    if (t.degenerate) return;
    uint64_t kbase = ((global_source(op, l)) * (uint64_t)sc->nF + (uint64_t)f) * (uint64_t)spt;
    const v3 e0 = sub3(t.p2, t.p1), e1 = sub3(t.p0, t.p2), e2 = sub3(t.p1, t.p0);
    const int vi[3] = {t.i0, t.i1, t.i2};
    for (int s = 0; s < spt; ++s) {
        float S, T;
        geo_t g;
        gvec_t gv;
        nlos_oracle_sample(op->seed, kbase + (uint64_t)s, &S, &T);
        if (!accept_sample(&t, sc, f, op->accel, S, T, lb, ub, &g)) continue;
        grad_vectors(&t, &g, normal_term, op, 0, &gv);
        const float bw[3] = {g.u, g.v, g.w};
        const v3 ce[3] = {cross3(gv.t2, e0), cross3(gv.t2, e1), cross3(gv.t2, e2)};
        for (int i = 0; i < tp->K; ++i) {
            double delta = tap_delta(tp, i);
            int bin = tap_bin(g.h, delta, lb, res);
            if (bin < 0 || bin >= nbins) continue;              /* deviation Q3 */
            v3 gg = scl3(g.dir, (float)(delta / tp->sigma_square * 2));
            v3 base = add3(gv.t1, scl3(gg, (float)gv.intensity));
            float wk = (float)tp->w[i];
            float dd = (float)((-2) * diff_row[bin]);
            for (int j = 0; j < 3; ++j) {
                v3 q = add3(scl3(base, bw[j]), ce[j]);
                q = scl3(q, wk);
                q = scl3(q, dd);
                grad[3 * (size_t)vi[j] + 0] += (double)(t.area * q.x) / (double)spt;
                grad[3 * (size_t)vi[j] + 1] += (double)(t.area * q.y) / (double)spt;
                grad[3 * (size_t)vi[j] + 2] += (double)(t.area * q.z) / (double)spt;
            }
        }
    }
}

/* Row GD: render_smoothed_gradients (SMO/transient_and_gradient.cpp:506-569) */
static int gradient_driver(const scene_t *sc, const float *origin, int L, const float *normal,
                           const float *vnormal, const float *albedo, int num_samples,
                           float lb, float ub, float res, int nbins, int refine, int sigma_bin,
                           const double *diff, int normal_term, double *gradient,
                           const nlos_oracle_opts *op) {
    const int nF = sc->nF, nV = sc->nV;
    const int spt = 1 + ((num_samples - 1) / nF);
    int nth = 1;
#ifdef _OPENMP
    oracle_set_threads(op->threads);
    nth = omp_get_max_threads();
#endif
    double *priv = (double *)calloc((size_t)nth * 3 * (size_t)nV, sizeof(double));
    if (!priv) return -2;
    taps_t tp;
    taps_init(&tp, refine, sigma_bin, res);
    const long long ntask = (long long)L * nF;
#pragma omp parallel
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        double *mine = priv + (size_t)tid * 3 * (size_t)nV;
#pragma omp for schedule(dynamic, 64)
        for (long long idx = 0; idx < ntask; ++idx) {
            int f = (int)(idx % nF), l = (int)(idx / nF);
            gradient_task(sc, origin, normal, vnormal, albedo, l, f, lb, ub, res, spt, nbins,
                          diff + (size_t)l * nbins, &tp, normal_term, mine, op);
        }
    }
    const int Ltot = op->total_sources > 0 ? op->total_sources : L;
    for (int th = 0; th < nth; ++th)
        for (size_t i = 0; i < 3 * (size_t)nV; ++i)
            gradient[i] += priv[(size_t)th * 3 * (size_t)nV + i] / Ltot;
    free(tp.w); free(priv);
    return 0;
}

int nlos_oracle_render_gradient(const double *data, const double *weight,
                                const float *origin, int L, const float *normal,
                                const float *V, int nV, const float *vnormal,
                                const float *albedo, const int32_t *F, int nF,
                                int num_samples, float lb, float ub, float res,
                                double *transient, double *pathlengths,
                                double *gradient, int refine, int sigma_bin,
                                int testing_flag, int loss_test,
                                const nlos_oracle_opts *opts) {
    nlos_oracle_opts dflt;
    if (!opts) { nlos_oracle_default_opts(&dflt); opts = &dflt; }
    if (nF <= 0 || L < 0 || refine < 1 || sigma_bin < 1) return -1;
    scene_t sc;
    if (scene_init(&sc, V, nV, F, nF, opts->accel)) return -1;
    int nbins = nlos_oracle_num_bins(lb, ub, res);
    if (pathlengths) fill_pathlengths(pathlengths, nbins, lb, res);
    /* SMO/stratifiedStreamedGradientRenderer.cpp:521-524 (Q2) */
    int fwd_refine = sigma_bin < 5 ? 1 : refine;
    int rc = forward_driver(&sc, origin, L, normal, vnormal, albedo, num_samples, lb, ub, res,
                            nbins, fwd_refine, sigma_bin, transient, opts);
    if (rc == 0) {
        size_t n = (size_t)L * (size_t)nbins;
        double *diff = (double *)malloc(sizeof(double) * (n ? n : 1));
        residual(data, weight, transient, n, loss_test, diff);
        int nt = opts->normal_term < 0 ? (testing_flag == 0 && vnormal != NULL) : opts->normal_term;
        rc = gradient_driver(&sc, origin, L, normal, vnormal, albedo, num_samples, lb, ub, res,
                             nbins, refine, sigma_bin, diff, nt, gradient, opts);
        free(diff);
    }
    scene_free(&sc);
    return rc;
}

/* Row A: streamedRayTraceTriangleGradientAlbedo (SMO/...:571-695) and
 * streamedRayTraceTriangleGradientAlpha (GGX/...:385-512) */
static double scalar_task(const scene_t *sc, const float *origin, const float *normal,
                          const float *albedo, int l, int f, float lb, float ub, float res,
                          int spt, int nbins, const double *diff_row, const taps_t *tp,
                          int wrt_alpha, const nlos_oracle_opts *op) {
    task_t t;
    double acc = 0;
    task_setup(&t, sc, origin, normal, l, f, NULL, albedo);
    if (t.degenerate) return 0;
    uint64_t kbase = ((global_source(op, l)) * (uint64_t)sc->nF + (uint64_t)f) * (uint64_t)spt;
    for (int s = 0; s < spt; ++s) {
        float S, T;
        geo_t g;
        nlos_oracle_sample(op->seed, kbase + (uint64_t)s, &S, &T);
        if (!accept_sample(&t, sc, f, op->accel, S, T, lb, ub, &g)) continue;
        float c2 = dot3(t.on, g.dir);
        float c3 = dot3(g.n, neg3(g.dir));
        if (c2 < 0) c2 = 0;
        if (c3 < 0) c3 = 0;
        float ff = c2 * c3 / g.h / g.h;
        double g0;
        if (wrt_alpha) {
            float da = ggx_eval_adiff_nw(op->ggx_alpha, dot3(g.n, neg3(g.dir)));
            g0 = g.alb * ff * ff * da;
        } else {
            g0 = ff * ff;
        }
        for (int i = 0; i < tp->K; ++i) {
            double delta = tap_delta(tp, i);
            int bin = tap_bin(g.h, delta, lb, res);
            if (bin < 0 || bin >= nbins) continue;
            if (wrt_alpha)
                acc += (double)t.area * g0 * tp->w[i] * (-2) * diff_row[bin] / (double)spt;
            else {
                double gg = g0 * tp->w[i] * (-2) * diff_row[bin];
                acc += (double)(t.area * gg) / (double)spt;
            }
        }
    }
    return acc;
}

double nlos_oracle_render_gradient_scalar(const double *data, const double *weight,
                                const float *origin, int L, const float *normal,
                                const float *V, int nV, const float *albedo,
                                const int32_t *F, int nF,
                                int num_samples, float lb, float ub, float res,
                                double *transient, double *pathlengths,
                                int refine, int sigma_bin, int loss_test,
                                int wrt_alpha, const nlos_oracle_opts *opts) {
    nlos_oracle_opts dflt;
    if (!opts) { nlos_oracle_default_opts(&dflt); opts = &dflt; }
    if (nF <= 0 || L < 0 || refine < 1 || sigma_bin < 1) return NAN;
    scene_t sc;
    if (scene_init(&sc, V, nV, F, nF, opts->accel)) return NAN;
    int nbins = nlos_oracle_num_bins(lb, ub, res);
    if (pathlengths) fill_pathlengths(pathlengths, nbins, lb, res);
    int fwd_refine = sigma_bin < 5 ? 1 : refine;
    double total = 0;
    if (forward_driver(&sc, origin, L, normal, NULL, albedo, num_samples, lb, ub, res, nbins,
                       fwd_refine, sigma_bin, transient, opts) == 0) {
        size_t n = (size_t)L * (size_t)nbins;
        double *diff = (double *)malloc(sizeof(double) * (n ? n : 1));
        residual(data, weight, transient, n, loss_test, diff);
        taps_t tp;
        taps_init(&tp, refine, sigma_bin, res);
        const int spt = 1 + ((num_samples - 1) / nF);
        const long long ntask = (long long)L * nF;
#ifdef _OPENMP
        oracle_set_threads(opts->threads);
#endif
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : total)
        for (long long idx = 0; idx < ntask; ++idx) {
            int f = (int)(idx % nF), l = (int)(idx / nF);
            total += scalar_task(&sc, origin, normal, albedo, l, f, lb, ub, res, spt, nbins,
                                 diff + (size_t)l * nbins, &tp, wrt_alpha, opts);
        }
        const int Ltot = opts->total_sources > 0 ? opts->total_sources : L;
        total /= Ltot;
        free(tp.w); free(diff);
    } else total = NAN;
    scene_free(&sc);
    return total;
}

/* ---------------------------------------------------------------- intensity */
/* Row X: streamedRayTraceIntensity (SMO/transient_and_gradient.cpp:22-119) */
int nlos_oracle_render_intensity(const float *origin, int L, const float *normal,
                                 const float *V, int nV, const float *vnormal,
                                 const int32_t *F, int nF, int num_samples,
                                 float lb, float ub, double *intensity,
                                 const nlos_oracle_opts *opts) {
    nlos_oracle_opts dflt;
    if (!opts) { nlos_oracle_default_opts(&dflt); opts = &dflt; }
    if (nF <= 0 || L < 0) return -1;
    scene_t sc;
    if (scene_init(&sc, V, nV, F, nF, opts->accel)) return -1;
    const int spt = 1 + ((num_samples - 1) / nF);
#ifdef _OPENMP
    oracle_set_threads(opts->threads);
#endif
#pragma omp parallel for schedule(dynamic, 16)
    for (int f = 0; f < nF; ++f) {
        double acc = 0;
        for (int l = 0; l < L; ++l) {
            task_t t;
            task_setup(&t, &sc, origin, normal, l, f, vnormal, NULL);
            if (t.degenerate) continue;
            uint64_t kbase = ((global_source(opts, l)) * (uint64_t)nF + (uint64_t)f) * (uint64_t)spt;
            for (int s = 0; s < spt; ++s) {
                float S, T;
                geo_t g;
                nlos_oracle_sample(opts->seed, kbase + (uint64_t)s, &S, &T);
                if (!accept_sample(&t, &sc, f, opts->accel, S, T, lb, ub, &g)) continue;
                float ff = -dot3(g.n, g.dir) * dot3(t.on, g.dir) / g.h / g.h;
                ff = emax0(ff);
                float val = t.area * 1.0f * ff * ff;
                if (opts->use_ggx) val = val * ggx_eval_nw(opts->ggx_alpha, dot3(g.n, neg3(g.dir)));
                acc += (double)val / (double)spt;
            }
        }
        intensity[f] += acc;
    }
    scene_free(&sc);
    return 0;
}

/* ---------------------------------------------------------- vertex gradient */
/* streamedRayTraceTriangleVertexGradient (SMO/transient_and_gradient.cpp:697-840),
 * driver :379-439.  gradient is [nbins,3]; /L normalisation as the reference. */
int nlos_oracle_render_vertex_gradient(int vertex_num, const float *origin, int L,
                                 const float *normal, const float *V, int nV,
                                 const int32_t *F, int nF, int num_samples,
                                 float lb, float ub, float res, double *gradient,
                                 int refine, int sigma_bin,
                                 const nlos_oracle_opts *opts) {
    nlos_oracle_opts dflt;
    if (!opts) { nlos_oracle_default_opts(&dflt); opts = &dflt; }
    if (nF <= 0 || L < 0 || refine < 1 || sigma_bin < 1) return -1;
    scene_t sc;
    if (scene_init(&sc, V, nV, F, nF, opts->accel)) return -1;
    int nbins = nlos_oracle_num_bins(lb, ub, res);
    const int spt = 1 + ((num_samples - 1) / nF);
    taps_t tp;
    taps_init(&tp, refine, sigma_bin, res);
    double *acc = (double *)calloc(3 * (size_t)nbins, sizeof(double));
    for (int l = 0; l < L; ++l)
        for (int f = 0; f < nF; ++f) {
            task_t t;
            task_setup(&t, &sc, origin, normal, l, f, NULL, NULL);
            if (t.i0 != vertex_num && t.i1 != vertex_num && t.i2 != vertex_num) continue;
            if (t.degenerate) continue;
            uint64_t kbase = ((global_source(opts, l)) * (uint64_t)nF + (uint64_t)f) * (uint64_t)spt;
            for (int s = 0; s < spt; ++s) {
                float S, T;
                geo_t g;
                gvec_t gv;
                nlos_oracle_sample(opts->seed, kbase + (uint64_t)s, &S, &T);
                if (!accept_sample(&t, &sc, f, opts->accel, S, T, lb, ub, &g)) continue;
                grad_vectors(&t, &g, 1, opts, 0, &gv);
                v3 e; float b;
                if (vertex_num == t.i0) { e = sub3(t.p2, t.p1); b = g.u; }
                else if (vertex_num == t.i1) { e = sub3(t.p0, t.p2); b = g.v; }
                else { e = sub3(t.p1, t.p0); b = g.w; }
                v3 ce = cross3(gv.t2, e);
                for (int i = 0; i < tp.K; ++i) {
                    double delta = tap_delta(&tp, i);
                    int bin = tap_bin(g.h, delta, lb, res);
                    if (bin < 0 || bin >= nbins) continue;
                    v3 gg = scl3(g.dir, (float)(delta / tp.sigma_square * 2));
                    v3 q = add3(scl3(add3(gv.t1, scl3(gg, (float)gv.intensity)), b), ce);
                    q = scl3(q, (float)tp.w[i]);
                    acc[3 * bin + 0] += (double)(t.area * q.x) / (double)spt;
                    acc[3 * bin + 1] += (double)(t.area * q.y) / (double)spt;
                    acc[3 * bin + 2] += (double)(t.area * q.z) / (double)spt;
                }
            }
        }
    const int Ltot = opts->total_sources > 0 ? opts->total_sources : L;
    for (int i = 0; i < 3 * nbins; ++i) gradient[i] += acc[i] / Ltot;
    free(acc); free(tp.w);
    scene_free(&sc);
    return 0;
}

/* ------------------------------------------------------------------ v1 path */
/* Rows W, G1: STR/stratifiedStreamedGradientRenderer.cpp:350-487.  Pass 1 is
 * the clamped transient kernel the v1 gradient driver calls; residual =
 * data - transient, optionally box(2w+1) (*) box(2w+1) filtered ('same' crop,
 * :447-462); pass 2 = G with one tap (delta 0, weight 1, normal term on,
 * t1 without albedo, :253-296) -- evident intent, typos not reproduced. */
static void box_same(const double *x, double *y, int n, int w) {
    double k = 1.0 / ((double)2 * w + 1);
    for (int i = 0; i < n; ++i) {
        double s = 0;
        for (int j = -w; j <= w; ++j) { int q = i + j; if (q >= 0 && q < n) s += x[q] * k; }
        y[i] = s;
    }
}

int nlos_oracle_render_gradient_v1(const double *data, const float *origin, int L,
                                const float *normal, const float *V, int nV,
                                const int32_t *F, int nF, int num_samples,
                                float lb, float ub, float res, int w_width,
                                double *transient, double *pathlengths,
                                double *gradient, const nlos_oracle_opts *opts) {
    nlos_oracle_opts dflt;
    if (!opts) { nlos_oracle_default_opts(&dflt); opts = &dflt; }
    if (nF <= 0 || L < 0) return -1;
    scene_t sc;
    if (scene_init(&sc, V, nV, F, nF, opts->accel)) return -1;
    int nbins = nlos_oracle_num_bins(lb, ub, res);
    if (pathlengths) fill_pathlengths(pathlengths, nbins, lb, res);
    nlos_oracle_opts op = *opts;
    op.use_ggx = 0;
    int rc = forward_driver(&sc, origin, L, normal, NULL, NULL, num_samples, lb, ub, res, nbins,
                            1, 1, transient, &op);
    if (rc) { scene_free(&sc); return rc; }
    size_t n = (size_t)L * (size_t)nbins;
    double *diff = (double *)malloc(sizeof(double) * (n ? n : 1));
    for (size_t i = 0; i < n; ++i) diff[i] = data[i] - transient[i];
    if (w_width > 0) {
        double *tmp = (double *)malloc(sizeof(double) * (size_t)nbins);
        for (int l = 0; l < L; ++l) {
            box_same(diff + (size_t)l * nbins, tmp, nbins, w_width);
            box_same(tmp, diff + (size_t)l * nbins, nbins, w_width);
        }
        free(tmp);
    }
    memset(gradient, 0, sizeof(double) * 3 * (size_t)nV);   /* v1 zeroes (STR/...:419) */
    const int spt = 1 + ((num_samples - 1) / nF);
    const int Ltot = op.total_sources > 0 ? op.total_sources : L;
    double *acc = (double *)calloc(3 * (size_t)nV, sizeof(double));
    for (int l = 0; l < L; ++l)
        for (int f = 0; f < nF; ++f) {
            task_t t;
            task_setup(&t, &sc, origin, normal, l, f, NULL, NULL);
            if (t.degenerate) continue;
            uint64_t kbase = ((global_source(&op, l)) * (uint64_t)nF + (uint64_t)f) * (uint64_t)spt;
            const v3 e[3] = {sub3(t.p2, t.p1), sub3(t.p0, t.p2), sub3(t.p1, t.p0)};
            const int vi[3] = {t.i0, t.i1, t.i2};
            for (int s = 0; s < spt; ++s) {
                float S, T;
                geo_t g;
                gvec_t gv;
                nlos_oracle_sample(op.seed, kbase + (uint64_t)s, &S, &T);
                if (!accept_sample(&t, &sc, f, op.accel, S, T, lb, ub, &g)) continue;
                int bin = (int)floorf((2.0f * g.h - lb) / res);
                if (bin < 0 || bin >= nbins) continue;
                grad_vectors(&t, &g, 1, &op, 1, &gv);
                const float bw[3] = {g.u, g.v, g.w};
                float dd = (float)((-2) * diff[(size_t)l * nbins + bin]);
                for (int j = 0; j < 3; ++j) {
                    v3 q = add3(scl3(gv.t1, bw[j]), cross3(gv.t2, e[j]));
                    q = scl3(q, dd);
                    acc[3 * (size_t)vi[j] + 0] += (double)(t.area * q.x) / (double)spt;
                    acc[3 * (size_t)vi[j] + 1] += (double)(t.area * q.y) / (double)spt;
                    acc[3 * (size_t)vi[j] + 2] += (double)(t.area * q.z) / (double)spt;
                }
            }
        }
    for (size_t i = 0; i < 3 * (size_t)nV; ++i) gradient[i] += acc[i] / Ltot;
    free(acc); free(diff);
    scene_free(&sc);
    return 0;
}

/* -------------------------------------------------------------- regularisers */
/* SMO/stratifiedStreamedGradientRenderer.cpp:27-180.
 * curvature_grad: per face g_j = cross(n, e_j / 2) = d(area)/d v_j           (:27-56, :159-180)
 * normal smoothing: value = sum_f A_f (1 - nbar_f . n_f), nbar_f = normalise(A_f n_f + sum_nbr A_g n_g),
 *                   per face g_j = cross(n_f - nbar_f, e_j / 2)              (:58-156)
 * The reference stores the per-vertex result with `=` and (in normal smoothing) lets every TBB
 * thread write thread 0's buffer, so its output is "whichever incident face wrote last".
 * overwrite = 0 (default here): accumulate over the incident faces -- the gradient the formulas
 * describe.  overwrite = 1: the incident face with the highest index wins -- what a serial run of
 * the reference produces.  Degenerate faces (area 0) are skipped in both roles, and so is a face
 * whose area-weighted neighbourhood normal cancels (shorter than 1e-3 of the summed areas: 0/0 or
 * rounding noise in the reference). */
static void face_normal_area(const float *V, const int32_t *F, int f, v3 *n, float *area, v3 *p) {
    p[0] = ld3(V + 3 * (size_t)F[3 * f]);
    p[1] = ld3(V + 3 * (size_t)F[3 * f + 1]);
    p[2] = ld3(V + 3 * (size_t)F[3 * f + 2]);
    v3 nr = cross3(sub3(p[1], p[0]), sub3(p[2], p[0]));
    *area = sqrtf(dot3(nr, nr)) / 2;
    *n = scl3(nr, 1.0f / (2 * *area));
}

double nlos_oracle_mesh_regulariser(const float *V, int nV, const int32_t *F, int nF,
                                    const int32_t *affinity, double *gradient, int overwrite) {
    double *normal = (double *)malloc(sizeof(double) * 3 * (size_t)nF);
    double *area = (double *)malloc(sizeof(double) * (size_t)nF);
    double value = 0;
    v3 p[3];
    for (int f = 0; f < nF; ++f) {
        v3 n; float a;
        face_normal_area(V, F, f, &n, &a, p);
        area[f] = a;
        normal[3 * f] = n.x; normal[3 * f + 1] = n.y; normal[3 * f + 2] = n.z;
    }
    memset(gradient, 0, sizeof(double) * 3 * (size_t)nV);
    for (int f = 0; f < nF; ++f) {          /* ascending order: "=" keeps the highest incident face */
        if (!(area[f] > 0)) continue;
        v3 fn = mk((float)normal[3 * f], (float)normal[3 * f + 1], (float)normal[3 * f + 2]);
        v3 d = fn;
        if (affinity) {
            v3 n = scl3(fn, (float)area[f]);
            float wsum = (float)area[f];
            for (int i = 0; i < 3; ++i) {
                int g = affinity[3 * f + i];
                if (g < 0 || !(area[g] > 0)) continue;
                v3 n1 = mk((float)normal[3 * g], (float)normal[3 * g + 1], (float)normal[3 * g + 2]);
                n = add3(n, scl3(n1, (float)area[g]));
                wsum += (float)area[g];
            }
            float len = sqrtf(dot3(n, n));
            if (!(len > 1e-3f * wsum)) {        /* neighbourhood normals cancel (0/0 or noise in the reference) */
                if (overwrite)                   /* the face still "writes last": zeros */
                    for (int j = 0; j < 3; ++j) memset(gradient + 3 * (size_t)F[3 * f + j], 0, 3 * sizeof(double));
                continue;
            }
            n = scl3(n, 1.0f / len);
            value += area[f] * (1 - dot3(n, fn));
            d = sub3(fn, n);
        }
        v3 nn; float aa;
        face_normal_area(V, F, f, &nn, &aa, p);
        const v3 e[3] = {sub3(p[2], p[1]), sub3(p[0], p[2]), sub3(p[1], p[0])};
        for (int j = 0; j < 3; ++j) {
            v3 g = cross3(d, scl3(e[j], 0.5f));
            double *o = gradient + 3 * (size_t)F[3 * f + j];
            if (overwrite) { o[0] = g.x; o[1] = g.y; o[2] = g.z; }
            else { o[0] += g.x; o[1] += g.y; o[2] += g.z; }
        }
    }
    free(normal); free(area);
    return value;
}

/* ------------------------------------------------------------------- jitter */
/* SPAD jitter variant, transient_rendering_cython/jitter/ ("JIT"):
 * forward = plain histogram convolved with the measured jitter kernel
 * (JIT/transient_and_gradient.cpp:271-355: y = full conv, transient[b] += y[b + weight_offset]);
 * gradient taps i -> bin floor((2h-lb)/res) + (i - jitter_offset), weights jitter_weight[i], and the
 * time-derivative term jitter_grad[i] * I * (-2) * dir / res (JIT/...:944-969). */
static void jitter_forward(const scene_t *sc, const float *origin, int L, const float *normal,
                           const float *vnormal, const float *albedo, int num_samples,
                           float lb, float ub, float res, int nbins, const double *jw, int joff,
                           int jlen, double *transient, const nlos_oracle_opts *op) {
    const int nF = sc->nF;
    const int spt = 1 + ((num_samples - 1) / nF);
    memset(transient, 0, sizeof(double) * (size_t)L * (size_t)nbins);
#pragma omp parallel
    {
        double *fine = (double *)malloc(sizeof(double) * (size_t)nbins);
        double *y = (double *)malloc(sizeof(double) * (size_t)(nbins + jlen - 1));
#pragma omp for schedule(dynamic, 1)
        for (int l = 0; l < L; ++l) {
            memset(fine, 0, sizeof(double) * (size_t)nbins);
            for (int f = 0; f < nF; ++f)
                forward_task(sc, origin, normal, vnormal, albedo, l, f, lb, ub, res, spt, nbins, fine, op);
            memset(y, 0, sizeof(double) * (size_t)(nbins + jlen - 1));
            for (int i = 0; i < nbins; ++i)
                for (int j = 0; j < jlen; ++j) y[i + j] += fine[i] * jw[j];
            for (int b = 0; b < nbins; ++b) transient[(size_t)l * nbins + b] += y[b + joff];
        }
        free(fine); free(y);
    }
}

static void jitter_gradient_task(const scene_t *sc, const float *origin, const float *normal,
                                 const float *vnormal, int l, int f, float lb, float ub, float res,
                                 int spt, int nbins, const double *diff_row, const double *jw,
                                 const double *jg, int joff, int jlen, int normal_term,
                                 double *grad, const nlos_oracle_opts *op) {
    task_t t;
    task_setup(&t, sc, origin, normal, l, f, vnormal, NULL);
    if (t.degenerate) return;
    uint64_t kbase = ((global_source(op, l)) * (uint64_t)sc->nF + (uint64_t)f) * (uint64_t)spt;
    const v3 e0 = sub3(t.p2, t.p1), e1 = sub3(t.p0, t.p2), e2 = sub3(t.p1, t.p0);
    const int vi[3] = {t.i0, t.i1, t.i2};
    for (int s = 0; s < spt; ++s) {
        float S, T;
        geo_t g;
        gvec_t gv;
        nlos_oracle_sample(op->seed, kbase + (uint64_t)s, &S, &T);
        if (!accept_sample(&t, sc, f, op->accel, S, T, lb, ub, &g)) continue;
        grad_vectors(&t, &g, normal_term, op, 0, &gv);
        const float bw[3] = {g.u, g.v, g.w};
        const v3 ce[3] = {cross3(gv.t2, e0), cross3(gv.t2, e1), cross3(gv.t2, e2)};
        const int b0 = (int)floorf((2.0f * g.h - lb) / res);       /* JIT/...:949-950, float floor */
        for (int i = 0; i < jlen; ++i) {
            int bin = b0 + (i - joff);
            if (bin < 0 || bin >= nbins) continue;                  /* deviation Q3 */
            float wk = (float)jw[i];
            /* jitter_grad[i] * intensity * (-2) in double, narrowed when it meets the float vector */
            v3 tg = scl3(g.dir, (float)(jg[i] * gv.intensity * (-2)));
            tg = mk(tg.x / res, tg.y / res, tg.z / res);
            v3 base = add3(scl3(gv.t1, wk), tg);
            float dd = (float)((-2) * diff_row[bin]);
            for (int j = 0; j < 3; ++j) {
                v3 q = add3(scl3(base, bw[j]), scl3(ce[j], wk));
                q = scl3(q, dd);
                grad[3 * (size_t)vi[j] + 0] += (double)(t.area * q.x) / (double)spt;
                grad[3 * (size_t)vi[j] + 1] += (double)(t.area * q.y) / (double)spt;
                grad[3 * (size_t)vi[j] + 2] += (double)(t.area * q.z) / (double)spt;
            }
        }
    }
}

/* JIT/stratifiedStreamedGradientRenderer.cpp:475-578 (gradient != NULL) and
 * JIT/stratifiedStreamedTransientRenderer.cpp (forward only: data == NULL / gradient == NULL).
 * jitter_grad may be NULL for the forward-only call. */
int nlos_oracle_render_jitter(const double *data, const double *weight,
                              const float *origin, int L, const float *normal,
                              const float *V, int nV, const float *vnormal, const float *albedo,
                              const int32_t *F, int nF, int num_samples,
                              float lb, float ub, float res,
                              const double *jitter_weight, const double *jitter_grad,
                              int jitter_offset, int jitter_length,
                              double *transient, double *pathlengths, double *gradient,
                              int testing_flag, const nlos_oracle_opts *opts) {
    nlos_oracle_opts dflt;
    if (!opts) { nlos_oracle_default_opts(&dflt); opts = &dflt; }
    if (nF <= 0 || L < 0 || jitter_length < 1 || jitter_offset < 0 || jitter_offset >= jitter_length) return -1;
    scene_t sc;
    if (scene_init(&sc, V, nV, F, nF, opts->accel)) return -1;
    const int nbins = nlos_oracle_num_bins(lb, ub, res);
    const int spt = 1 + ((num_samples - 1) / nF);
    if (pathlengths) fill_pathlengths(pathlengths, nbins, lb, res);
#ifdef _OPENMP
    oracle_set_threads(opts->threads);
#endif
    jitter_forward(&sc, origin, L, normal, vnormal, albedo, num_samples, lb, ub, res, nbins,
                   jitter_weight, jitter_offset, jitter_length, transient, opts);
    if (data && gradient && jitter_grad) {
        size_t n = (size_t)L * (size_t)nbins;
        double *diff = (double *)malloc(sizeof(double) * (n ? n : 1));
        residual(data, weight, transient, n, 0, diff);
        int nt = opts->normal_term < 0 ? (testing_flag == 0 && vnormal != NULL) : opts->normal_term;
        int nth = 1;
#ifdef _OPENMP
        nth = omp_get_max_threads();
#endif
        double *priv = (double *)calloc((size_t)nth * 3 * (size_t)nV, sizeof(double));
#pragma omp parallel
        {
            int tid = 0;
#ifdef _OPENMP
            tid = omp_get_thread_num();
#endif
            double *mine = priv + (size_t)tid * 3 * (size_t)nV;
#pragma omp for schedule(dynamic, 64)
            for (long long idx = 0; idx < (long long)L * nF; ++idx) {
                int f = (int)(idx % nF), l = (int)(idx / nF);
                jitter_gradient_task(&sc, origin, normal, vnormal, l, f, lb, ub, res, spt, nbins,
                                     diff + (size_t)l * nbins, jitter_weight, jitter_grad, jitter_offset,
                                     jitter_length, nt, mine, opts);
            }
        }
        const int Ltot = opts->total_sources > 0 ? opts->total_sources : L;
        for (int th = 0; th < nth; ++th)
            for (size_t i = 0; i < 3 * (size_t)nV; ++i)
                gradient[i] += priv[(size_t)th * 3 * (size_t)nV + i] / Ltot;
        free(priv); free(diff);
    }
    scene_free(&sc);
    return 0;
}

/* ------------------------------------------------------------ non-confocal */
/* Row N: laser point a != sensor point b.  The reference has no native
 * kernel for it; its prototypes (transient_rendering_python/rendering.py:37-93, angular sampling;
 * mesh_optimization/rendering.py:739-797) fix the geometry -- the surface point must be the
 * nearest hit seen from the laser AND visible from the sensor, path length d1 + d2 -- and this
 * restatement keeps the v2 conventions of rows F/G (stratified samples per (pair, face), wall
 * cosines, clamped form factors, bin = floor((d1 + d2 - lb)/res)), so that a == b reproduces the
 * confocal rows: the forward pass bit for bit, the gradient up to fp32 rounding of
 *     t1 = alb * (ff_b * grad ff_a + ff_a * grad ff_b),  grad ff = (n_o c3 - n c2 - 4 dir c2 c3)/d^3
 * against  2 alb c2 c3 (...) / h^5. */
typedef struct { float u, v, w, d1, d2; v3 dirA, dirB, n; float alb; } geo_nc_t;

static inline int accept_sample_nc(const task_t *t, v3 b, const scene_t *sc, int f, int accel,
                                   float S, float T, float lb, float ub, geo_nc_t *g) {
    float sq = sqrtf(T);
    float u = 1 - sq, v = (1 - S) * sq, w = S * sq;
    v3 p = bary3(u, t->p0, v, t->p1, w, t->p2);
    /* laser leg: identical to accept_sample() */
    v3 dA = sub3(p, t->o);
    v3 dirA = scl3(dA, 1.0f / sqrtf(dot3(dA, dA)));
    hit_t hA = closest_hit(sc, t->o, dirA, accel);
    if (hA.prim != f) return 0;
    g->v = hA.u; g->w = hA.v; g->u = 1.0f - g->v - g->w;
    v3 qA = bary3(g->u, t->p0, g->v, t->p1, g->w, t->p2);
    v3 eA = sub3(qA, t->o);
    g->d1 = sqrtf(dot3(eA, eA));
    /* sensor leg: same construction from b towards the same stratified point */
    v3 dB = sub3(p, b);
    v3 dirB = scl3(dB, 1.0f / sqrtf(dot3(dB, dB)));
    hit_t hB = closest_hit(sc, b, dirB, accel);
    if (hB.prim != f) return 0;
    v3 qB = bary3(1.0f - hB.u - hB.v, t->p0, hB.u, t->p1, hB.v, t->p2);
    v3 eB = sub3(qB, b);
    g->d2 = sqrtf(dot3(eB, eB));
    float tot = g->d1 + g->d2;
    if (!((tot <= ub) && (tot >= lb))) return 0;
    g->dirA = dirA; g->dirB = dirB;
    g->n = t->fn;
    if (t->has_vn) g->n = bary3(g->u, t->n0, g->v, t->n1, g->w, t->n2);
    g->alb = 1.0f;
    if (t->has_alb) g->alb = g->u * t->a0 + g->v * t->a1 + g->w * t->a2;
    return 1;
}

static void forward_task_nc(const scene_t *sc, const float *laser, const float *lnormal,
                            const float *sensor, const float *snormal, const float *vnormal,
                            const float *albedo, int l, int f, float lb, float ub, float res,
                            int spt, int nbins, double *row, const nlos_oracle_opts *op) {
    task_t t;
    task_setup(&t, sc, laser, lnormal, l, f, vnormal, albedo);
    if (t.degenerate) return;
    const v3 b = ld3(sensor + 3 * (size_t)l), bn = ld3(snormal + 3 * (size_t)l);
    uint64_t kbase = ((global_source(op, l)) * (uint64_t)sc->nF + (uint64_t)f) * (uint64_t)spt;
    for (int s = 0; s < spt; ++s) {
        float S, T;
        geo_nc_t g;
        nlos_oracle_sample(op->seed, kbase + (uint64_t)s, &S, &T);
        if (!accept_sample_nc(&t, b, sc, f, op->accel, S, T, lb, ub, &g)) continue;
        float ffa = emax0(-dot3(g.n, g.dirA) * dot3(t.on, g.dirA) / g.d1 / g.d1);
        float ffb = emax0(-dot3(g.n, g.dirB) * dot3(bn, g.dirB) / g.d2 / g.d2);
        int bin = (int)floorf(((g.d1 + g.d2) - lb) / res);
        if (bin < 0 || bin >= nbins) continue;
        float val = t.area * g.alb * ffa * ffb;
        if (op->use_ggx) {
            ggx_pair_t gp;
            ggx_pair(op->ggx_alpha, g.n, neg3(g.dirA), neg3(g.dirB), 0, &gp);
            val = val * gp.brdf;
        }
        row[bin] += (double)val / (double)spt;
    }
}

static void grad_vectors_nc(const task_t *t, v3 bn, const geo_nc_t *g, int normal_term, int use_ggx, float alpha,
                            gvec_t *out) {
    float c2a = dot3(t->on, g->dirA), c3a = dot3(g->n, neg3(g->dirA));
    float c2b = dot3(bn, g->dirB), c3b = dot3(g->n, neg3(g->dirB));
    if (c2a < 0) c2a = 0;
    if (c3a < 0) c3a = 0;
    if (c2b < 0) c2b = 0;
    if (c3b < 0) c3b = 0;
    float ffa = c2a * c3a / g->d1 / g->d1, ffb = c2b * c3b / g->d2 / g->d2;
    v3 ia = add3(sub3(scl3(t->on, c3a), scl3(g->n, c2a)), scl3(scl3(scl3(neg3(g->dirA), 4), c2a), c3a));
    v3 ib = add3(sub3(scl3(bn, c3b), scl3(g->n, c2b)), scl3(scl3(scl3(neg3(g->dirB), 4), c2b), c3b));
    v3 ga = scl3(ia, 1.0f / ((g->d1 * g->d1) * g->d1));
    v3 gb = scl3(ib, 1.0f / ((g->d2 * g->d2) * g->d2));
    out->intensity = (double)(g->alb * ffa * ffb);
    v3 t1 = scl3(add3(scl3(ga, ffb), scl3(gb, ffa)), g->alb);
    v3 gn = mk(0, 0, 0);
    if (normal_term) {
        /* d I / d n,  I = alb c2a c3a c2b c3b / (d1^2 d2^2),  c3x = -n.dirx */
        gn = add3(scl3(g->dirA, c3b), scl3(g->dirB, c3a));
        gn = scl3(gn, -(g->alb * c2a * c2b));
        gn = scl3(gn, 1.0f / ((g->d1 * g->d1) * (g->d2 * g->d2)));
    }
    if (use_ggx) {
        /* I = I_lambert * brdf(n, wa, wb), wx = -dirx:  dI/dp = brdf dI_l/dp + I_l (J_a^T ga + J_b^T gb) with
         * J_x = d wx / dp = -(1 - wx wx^T) / d_x;  dI/dn = brdf dI_l/dn + I_l d brdf/dn */
        ggx_pair_t gp;
        v3 wa = neg3(g->dirA), wb = neg3(g->dirB);
        ggx_pair(alpha, g->n, wa, wb, 1, &gp);
        float il = (float)out->intensity;
        v3 pa = scl3(sub3(gp.ga, scl3(wa, dot3(wa, gp.ga))), -1.0f / g->d1);
        v3 pb = scl3(sub3(gp.gb, scl3(wb, dot3(wb, gp.gb))), -1.0f / g->d2);
        t1 = add3(scl3(t1, gp.brdf), scl3(add3(pa, pb), il));
        if (normal_term) gn = add3(scl3(gn, gp.brdf), scl3(gp.gn, il));
        out->intensity = (double)(il * gp.brdf);
    }
    if (normal_term) {
        float ct = dot3(gn, g->n);
        gn = sub3(gn, scl3(g->n, ct));
    }
    v3 t2 = scl3(g->n, (float)out->intensity);
    t2 = scl3(add3(t2, gn), 1.0f / (2 * t->area));
    out->t1 = t1; out->t2 = t2;
}

static void gradient_task_nc(const scene_t *sc, const float *laser, const float *lnormal,
                             const float *sensor, const float *snormal, const float *vnormal,
                             const float *albedo, int l, int f, float lb, float ub, float res,
                             int spt, int nbins, const double *diff_row, const taps_t *tp,
                             int normal_term, double *grad, const nlos_oracle_opts *op) {
    task_t t;
    task_setup(&t, sc, laser, lnormal, l, f, vnormal, albedo);
    if (t.degenerate) return;
    const v3 b = ld3(sensor + 3 * (size_t)l), bn = ld3(snormal + 3 * (size_t)l);
    uint64_t kbase = ((global_source(op, l)) * (uint64_t)sc->nF + (uint64_t)f) * (uint64_t)spt;
    const v3 e0 = sub3(t.p2, t.p1), e1 = sub3(t.p0, t.p2), e2 = sub3(t.p1, t.p0);
    const int vi[3] = {t.i0, t.i1, t.i2};
    for (int s = 0; s < spt; ++s) {
        float S, T;
        geo_nc_t g;
        gvec_t gv;
        nlos_oracle_sample(op->seed, kbase + (uint64_t)s, &S, &T);
        if (!accept_sample_nc(&t, b, sc, f, op->accel, S, T, lb, ub, &g)) continue;
        /* a leg whose clamped form factor is zero contributes nothing (d max(0,x)/dx = 0 there;
         * the confocal expressions vanish by themselves through their c2*c3 factor) */
        if (!(emax0(-dot3(g.n, g.dirA) * dot3(t.on, g.dirA) / g.d1 / g.d1) > 0.0f)) continue;
        if (!(emax0(-dot3(g.n, g.dirB) * dot3(bn, g.dirB) / g.d2 / g.d2) > 0.0f)) continue;
        grad_vectors_nc(&t, bn, &g, normal_term, op->use_ggx, op->ggx_alpha, &gv);
        const float bw[3] = {g.u, g.v, g.w};
        const v3 ce[3] = {cross3(gv.t2, e0), cross3(gv.t2, e1), cross3(gv.t2, e2)};
        const v3 dsum = add3(g.dirA, g.dirB);
        const float tot = g.d1 + g.d2;
        for (int i = 0; i < tp->K; ++i) {
            double delta = tap_delta(tp, i);
            int bin = (int)floor((tot + delta - lb) / res);
            if (bin < 0 || bin >= nbins) continue;
            /* d(d1 + d2)/dp = dirA + dirB  (confocal: 2 dir, SMO/...:974) */
            v3 gg = scl3(dsum, (float)(delta / tp->sigma_square));
            v3 base = add3(gv.t1, scl3(gg, (float)gv.intensity));
            float wk = (float)tp->w[i];
            float dd = (float)((-2) * diff_row[bin]);
            for (int j = 0; j < 3; ++j) {
                v3 q = add3(scl3(base, bw[j]), ce[j]);
                q = scl3(q, wk);
                q = scl3(q, dd);
                grad[3 * (size_t)vi[j] + 0] += (double)(t.area * q.x) / (double)spt;
                grad[3 * (size_t)vi[j] + 1] += (double)(t.area * q.y) / (double)spt;
                grad[3 * (size_t)vi[j] + 2] += (double)(t.area * q.z) / (double)spt;
            }
        }
    }
}

/* P (laser, sensor) pairs; data == NULL: forward only.  Forward is always the plain histogram
 * (refine applies to the gradient taps only, as in the v2 driver with sigma_bin < 5). */
int nlos_oracle_render_nonconfocal(const double *data, const double *weight,
                                   const float *laser, const float *laser_normal,
                                   const float *sensor, const float *sensor_normal, int P,
                                   const float *V, int nV, const float *vnormal,
                                   const float *albedo, const int32_t *F, int nF,
                                   int num_samples, float lb, float ub, float res,
                                   double *transient, double *pathlengths, double *gradient,
                                   int refine, int sigma_bin, int testing_flag, int loss_test,
                                   const nlos_oracle_opts *opts) {
    nlos_oracle_opts dflt;
    if (!opts) { nlos_oracle_default_opts(&dflt); opts = &dflt; }
    if (nF <= 0 || P < 0 || refine < 1 || sigma_bin < 1) return -1;
    scene_t sc;
    if (scene_init(&sc, V, nV, F, nF, opts->accel)) return -1;
    const int nbins = nlos_oracle_num_bins(lb, ub, res);
    const int spt = 1 + ((num_samples - 1) / nF);
    if (pathlengths) fill_pathlengths(pathlengths, nbins, lb, res);
    memset(transient, 0, sizeof(double) * (size_t)P * (size_t)nbins);
#ifdef _OPENMP
    oracle_set_threads(opts->threads);
#endif
    /* forward refinement rule of the v2 gradient driver (SMO/stratifiedStreamedGradientRenderer.cpp:521-524);
     * forward-only calls (data == NULL) refine whenever refine > 1, like renderStreamedTransient */
    const int fr = (data && gradient) ? (sigma_bin < 5 ? 1 : refine) : refine;
    const int rb = nbins * fr;
    const int Kf = 4 * fr * sigma_bin + 1;
    double *kern = (double *)malloc(sizeof(double) * (size_t)Kf);
    gauss_kernel(kern, fr, sigma_bin, res);
    /* one pair per task: rows are private to the task, no reduction needed */
#pragma omp parallel
    {
        double *fine = (double *)malloc(sizeof(double) * (size_t)rb);
        double *y = (double *)malloc(sizeof(double) * (size_t)(rb + Kf - 1));
#pragma omp for schedule(dynamic, 1)
        for (int l = 0; l < P; ++l) {
            double *row = transient + (size_t)l * nbins;
            memset(fine, 0, sizeof(double) * (size_t)rb);
            for (int f = 0; f < nF; ++f)
                forward_task_nc(&sc, laser, laser_normal, sensor, sensor_normal, vnormal, albedo, l, f,
                                lb, ub, res / fr, spt, rb, fine, opts);
            if (fr <= 1) {
                memcpy(row, fine, sizeof(double) * (size_t)nbins);
            } else {
                /* Gaussian + fold exactly as forward_driver() (SMO/transient_and_gradient.cpp:348-371) */
                memset(y, 0, sizeof(double) * (size_t)(rb + Kf - 1));
                for (int i = 0; i < rb; ++i)
                    for (int j = 0; j < Kf; ++j) y[i + j] += fine[i] * kern[j];
                for (int b = 0; b < rb; ++b) row[b / fr] += y[b + 2 * fr * sigma_bin];
            }
        }
        free(fine); free(y);
    }
    free(kern);
    if (data && gradient) {
        size_t n = (size_t)P * (size_t)nbins;
        double *diff = (double *)malloc(sizeof(double) * (n ? n : 1));
        residual(data, weight, transient, n, loss_test, diff);
        int nt = opts->normal_term < 0 ? (testing_flag == 0 && vnormal != NULL) : opts->normal_term;
        taps_t tp;
        taps_init(&tp, refine, sigma_bin, res);
        int nth = 1;
#ifdef _OPENMP
        nth = omp_get_max_threads();
#endif
        double *priv = (double *)calloc((size_t)nth * 3 * (size_t)nV, sizeof(double));
#pragma omp parallel
        {
            int tid = 0;
#ifdef _OPENMP
            tid = omp_get_thread_num();
#endif
            double *mine = priv + (size_t)tid * 3 * (size_t)nV;
#pragma omp for schedule(dynamic, 64)
            for (long long idx = 0; idx < (long long)P * nF; ++idx) {
                int f = (int)(idx % nF), l = (int)(idx / nF);
                gradient_task_nc(&sc, laser, laser_normal, sensor, sensor_normal, vnormal, albedo,
                                 l, f, lb, ub, res, spt, nbins, diff + (size_t)l * nbins, &tp, nt,
                                 mine, opts);
            }
        }
        const int Ptot = opts->total_sources > 0 ? opts->total_sources : P;
        for (int th = 0; th < nth; ++th)
            for (size_t i = 0; i < 3 * (size_t)nV; ++i)
                gradient[i] += priv[(size_t)th * 3 * (size_t)nV + i] / Ptot;
        free(tp.w); free(priv); free(diff);
    }
    scene_free(&sc);
    return 0;
}

/* ------------------------------------------------------------ sample tracer */
int nlos_oracle_trace_sample(const float *origin_l, const float *normal_l,
                             int64_t l_global, int f, int s, int spt,
                             const float *V, int nV, const int32_t *F, int nF,
                             float lb, float ub, float res, uint64_t seed,
                             double *out) {
    scene_t sc;
    if (scene_init(&sc, V, nV, F, nF, 0)) return -1;
    task_t t;
    geo_t g;
    float S, T;
    int ok = 0;
    task_setup(&t, &sc, origin_l, normal_l, 0, f, NULL, NULL);
    uint64_t k = (((uint64_t)l_global) * (uint64_t)nF + (uint64_t)f) * (uint64_t)spt + (uint64_t)s;
    nlos_oracle_sample(seed, k, &S, &T);
    for (int i = 0; i < 7; ++i) out[i] = 0;
    if (!t.degenerate && accept_sample(&t, &sc, f, 0, S, T, lb, ub, &g)) {
        v3 p = bary3(g.u, t.p0, g.v, t.p1, g.w, t.p2);
        float ff = emax0(-dot3(g.n, g.dir) * dot3(t.on, g.dir) / g.h / g.h);
        out[0] = p.x; out[1] = p.y; out[2] = p.z; out[3] = g.h;
        out[4] = floorf((2.0f * g.h - lb) / res);
        out[5] = ff;
        out[6] = (double)(t.area * g.alb * ff * ff) / (double)spt;
        ok = 1;
    }
    scene_free(&sc);
    return ok;
}
