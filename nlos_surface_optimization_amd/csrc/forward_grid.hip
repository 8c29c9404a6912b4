// forward_grid.hip -- pass 1 through the per-source perspective grid (gfx950): the fast path.
//
//   k_forward_grid<FEAT, NCM, TILED>  one workgroup per source (or per (source, slope-space tile) for
//                                     meshes beyond one workgroup's LDS); NCM = the two passes of a
//                                     non-confocal pair
//   k_tile_bin                        triangle -> tile subsets for the tiled variant
// Same samples, same accept/reject decisions as forward_bvh.hip (smoothed_transient/
// transient_and_gradient.cpp:122-237); only the "is anything in front of this ray" query differs.
#include "render_common.h"

#include <type_traits>

namespace nlos {
namespace {

// ------------------------------------------------------------- forward (grid)
// Per-source perspective grid.  Every ray of a workgroup starts at the same wall point o, so the
// triangles that can block the ray towards slope (mx, my) = (dx/dz, dy/dz) are exactly those whose
// perspective projection from o covers that slope point.  Per source the workgroup builds, in LDS, an
// R x R grid over slope space in CSR form (counting pass, block scan, fill pass) holding, per cell, one
// 32-bit entry for every triangle whose projection (conservatively rasterised) overlaps the cell AND that
// is not entirely deeper than the deepest live face seen through that cell -- nothing deeper can be in
// front of any ray that will ever look the cell up.  The entry carries the triangle's quantised depth and
// the sub-cell masks of its projected bounding box (layout below).
// A ray walks only its own cell's list, rejects a candidate with three integer compares on the LDS word and
// queues the survivors for the exact triangle test, which runs on dense lanes (wave-cooperative rounds).
// The kernel is bound by VALU issue, so everything in front of the exact test is
// integer work on LDS.  The accepted samples are identical to the BVH path's: lists and masks are
// supersets of what the exact test could report.  Sources for which the scene is not strictly in front of
// the wall point fall back to the stackless BVH traversal; sources whose cell lists overflow the entry
// capacity restart on a coarser grid (grid_body<COARSE>), then with the whole CU's LDS, and only then
// fall back to the BVH.
struct GridView {
    float gx0, gy0, inv_cw, inv_ch;   // cell = floor((m - g0) * inv_c)
    float z0, inv_qz;                 // quantised depth = floor((z - z0) * inv_qz), zmax levels over the scene
    int ib, zmax;                     // entry = depth (32 - 2 kSub - ib bits) | y mask | x mask | index (ib bits)
    int R;
    float padx, pady;                 // sub-cell mask margin: kSubFrac + kSlopeAbs in sub-cells
};

__device__ __forceinline__ int cell_coord(float m, float g0, float inv_c, int R) {
    int c = (int)floorf((m - g0) * inv_c);
    return min(max(c, 0), R - 1);
}

struct Proj2 { float ax, ay, bx, by, cx, cy; };

__device__ __forceinline__ Proj2 project_tri(V3 o, V3 p0, V3 p1, V3 p2) {
    // conservative uses only: approximate reciprocals are covered by the margins below
    const float iz0 = __builtin_amdgcn_rcpf(p0.z - o.z), iz1 = __builtin_amdgcn_rcpf(p1.z - o.z),
                iz2 = __builtin_amdgcn_rcpf(p2.z - o.z);
    Proj2 q;
    q.ax = (p0.x - o.x) * iz0; q.ay = (p0.y - o.y) * iz0;
    q.bx = (p1.x - o.x) * iz1; q.by = (p1.y - o.y) * iz1;
    q.cx = (p2.x - o.x) * iz2; q.cy = (p2.y - o.y) * iz2;
    return q;
}

// Cell-list entry (32 bit, LDS), most significant first:
//   [31:21] smallest depth of the triangle, quantised downwards to 2048 levels over the scene's depth range
//   [20:17] y mask, [16:13] x mask: which quarters of THIS cell the triangle's projected bounding box touches
//   [12:0]  index of the triangle (Morton order); the grid path is limited to F <= 8191
// (tiled grid: 14 index bits, 10 depth bits).  Depth resolution is worth more than mask resolution: the
// candidates that survive are mostly the own face's neighbours, and with sixths / eighths of a cell and 7 / 3
// depth bits the kernel is 1.5 % / 7 % slower (measured on the bunny, the mannequin and a 20 k-face mesh).
// A ray carries  rlim = (its own hit depth level << 21) | 0x1FFFFF  and  rmask = its sub-cell bit in
// both masks; a candidate survives iff  w <= rlim  (not entirely behind the hit),  (w & rmask) == rmask
// (the slope point is inside the box) and it is not the ray's own face: three compares on one LDS word.
struct BBoxF { float x0, x1, y0, y1; };
// Conservative margins of the grid.  What they must cover is the error of a hit the exact test
// reports, and that error grows like 1 / |cos| of the angle between ray and triangle normal: ~6 eps / |cos| of t along
// the ray, ~3 eps |C| / |cos| beside it.  The BASE margins below are sized for |cos| >= 2^-6 (2.3e-5 t, 6e-6 m at
// |C| ~ 0.5 m; they are round 2's, which passed 7 590 fuzz scenes under a rule that cut off there).  The grazing rule
// of the contract now admits hits down to |cos| = NLOS_GRAZE_RATIO / 2 = 2^-10, sixteen times more error -- but only on
// triangles that this source really sees that edge-on.  So every margin is scaled PER (source, triangle) by
//     graze_scale = clamp(1 / (64 cos_min), 1, kGrazeMaxScale),   cos_min = dist(o, plane) / max_i |p_i - o|
// (the smallest |cos| any ray from o can have where it meets the triangle's plane inside the triangle: n . (p - o) is
// the same for every p of the plane).  98 % of the triangles keep the base margins; uniform margins of 8x / 16x cost
// 12 % / 51 % of the kernel (measured, profiles/r03_ab_margins.log), the scaled ones nothing measurable.
// Lateral margins are a fraction of a cell plus an absolute slope term (fine grids: the tiles of large meshes have
// cells of 3e-3 slope units), depth margins a relative term plus one quantisation level.
#ifndef NLOS_MARGIN_SCALE
#define NLOS_MARGIN_SCALE 1.0f            // x the base margins (diagnostic builds: the A/B of uniform margins)
#endif
constexpr float kSlopeAbs = 5e-5f * NLOS_MARGIN_SCALE;                  // absolute lateral margin, slope units
constexpr float kBoxFrac = 2e-3f * NLOS_MARGIN_SCALE;                   // bounding-box margin, cells
constexpr float kEdgeFrac = 4e-3f * NLOS_MARGIN_SCALE;                  // edge-function slack, cells
constexpr float kSubFrac = 0.02f * NLOS_MARGIN_SCALE;                   // sub-cell mask margin, sub-cells
constexpr float kDepthEps = 1e-4f * NLOS_MARGIN_SCALE;                  // relative depth margin (x graze_scale)
constexpr float kDepthRel = 1.0f + kDepthEps;                           // the ray's own depth level (arithmetic of zs only)
constexpr float kGrazeMaxScale = 1.1f * (2.0f / (64.0f * NLOS_GRAZE_RATIO));   // 2^-6 / (ratio / 2), 10 % up

// margin multiplier of triangle (p0, p1, p2) seen from o; nhat = its unit normal (facerec, the scene build's
// ng / |ng|).  hmin = 1.1 / 64 x the distance from o to the farthest corner of the scene's box: a triangle whose plane
// passes o at more than that cannot be seen at |cos| < 2^-6 -- one dot product and a compare for 98 % of the triangles.
// Degenerate triangles (nhat = NaN) never report a hit; they get the maximum.
__device__ __forceinline__ float graze_scale(V3 o, V3 p0, V3 p1, V3 p2, V3 nhat, float hmin) {
#ifdef NLOS_DIAG_NO_GRAZE_SCALE   // diagnostic builds only (what the scaled margins cost): results can differ from the oracle's
    return 1.0f;
#endif
    const V3 c0 = p0 - o;
    const float h = fabsf(dot(nhat, c0));                                           // distance of o from the plane
    float ms = 1.0f;
    if (!(h >= hmin)) {
        const V3 c1 = p1 - o, c2 = p2 - o;
        const float d2 = fmaxf(fmaxf(dot(c0, c0), dot(c1, c1)), dot(c2, c2));
        const float sc = sqrtf(d2) * __builtin_amdgcn_rcpf(h * (64.0f / 1.1f));     // 1.1 / (64 cos_min)
        ms = fmaxf(fminf(sc, kGrazeMaxScale), 1.0f);                                // (NaN -> kGrazeMaxScale -> >= 1)
    }
    return ms;
}
#ifndef NLOS_KSUB
#define NLOS_KSUB 4
#endif
constexpr int kSub = NLOS_KSUB;   // sub-cell levels per axis
#ifndef NLOS_MAX_COARSEN
#define NLOS_MAX_COARSEN 2
#endif
constexpr int kIdxBits = 13;      // single-workgroup grid: 11 depth bits
constexpr int kIdxBitsTiled = 14; // tiled grid: subsets up to 16383 triangles, 10 depth bits
#ifndef NLOS_EXACT_ROUND
#define NLOS_EXACT_ROUND 64
#endif
constexpr int kRound = NLOS_EXACT_ROUND;   // pairs per exact-test round: one per lane (two -- two record gathers in flight -- paid at four
                                            // waves per SIMD; at six the registers are worth more: 1.91 -> 1.89 ms)
#ifndef NLOS_SCAN_WIDTH
#define NLOS_SCAN_WIDTH 4
#endif
constexpr int kScan = NLOS_SCAN_WIDTH;      // cell-list entries per trip of the lockstep walk
constexpr int kQueueCap = kRound + 64;      // a trip appends at most 64 pairs per slot before the drain check
constexpr int kQueueWords = kQueueCap + 3;  // + a dump slot for the lanes that append nothing + the wave's 64-bit occlusion mask
#ifndef NLOS_GRID_NT
#define NLOS_GRID_NT 768
#endif
// Threads per workgroup of the grid kernel.  Two workgroups share a CU's LDS, so the block size sets the occupancy:
// 512 threads = 4 waves per SIMD (128 VGPRs), 768 = 6 waves (80 VGPRs).  With one ray per lane the trace loop fits
// the smaller budget, and the two extra waves fill the issue slots the lockstep walk and the record gathers leave
// idle: 2.16 -> 1.95 ms (640 threads: 10 waves do not spread evenly over 4 SIMDs, 2.9 ms; 1024: 64 VGPRs spill, 3.3 ms).
constexpr int kGridNT = NLOS_GRID_NT;
constexpr int kGridWaves = kGridNT / 64;

// Diagnostic builds only (tools/pad_test.sh, round 6): N extra VALU instructions at one point of the kernel -- what an
// instruction costs IN SITU.  If the SIMDs' issue slots are saturated the launch grows by N x trips x the instruction's
// issue cost / (SIMDs x clock); if they are not, by less: the slope is the share of an instruction's issue time that removing
// it would give back.  OP 0 = v_add_u32 (full rate), 1 = v_lshlrev_b32, 2 = v_cmp_lt_u32, 3 = v_min_u32, 4 = v_mul_lo_u32 (half rate
// back to back in tools/issue_rate.hip), 5 = v_fma_f32; two independent temporaries.
#ifndef NLOS_DIAG_PAD_OP
#define NLOS_DIAG_PAD_OP 0
#endif
template <int N>
__device__ __forceinline__ void diag_pad() {
    if constexpr (N > 0) {
        uint32_t t0, t1;
        asm volatile("v_mov_b32 %0, 1\n\tv_mov_b32 %1, 2" : "=v"(t0), "=v"(t1));
#pragma unroll
        for (int i = 0; i < N / 2; ++i) {
#if NLOS_DIAG_PAD_OP == 0
            asm volatile("v_add_u32 %0, %0, %0\n\tv_add_u32 %1, %1, %1" : "+v"(t0), "+v"(t1));
#elif NLOS_DIAG_PAD_OP == 1
            asm volatile("v_lshlrev_b32 %0, 1, %0\n\tv_lshlrev_b32 %1, 1, %1" : "+v"(t0), "+v"(t1));
#elif NLOS_DIAG_PAD_OP == 2
            asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cmp_lt_u32 vcc, %1, %0" : "+v"(t0), "+v"(t1) : : "vcc");
#elif NLOS_DIAG_PAD_OP == 3
            asm volatile("v_min_u32 %0, %0, %1\n\tv_min_u32 %1, %1, %0" : "+v"(t0), "+v"(t1));
#elif NLOS_DIAG_PAD_OP == 4
            asm volatile("v_mul_lo_u32 %0, %0, %0\n\tv_mul_lo_u32 %1, %1, %1" : "+v"(t0), "+v"(t1));
#else
            asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1" : "+v"(t0), "+v"(t1));
#endif
        }
    }
}
#ifndef NLOS_DIAG_PAD_WALK
#define NLOS_DIAG_PAD_WALK 0
#endif
#ifndef NLOS_DIAG_PAD_GEN
#define NLOS_DIAG_PAD_GEN 0
#endif
#ifndef NLOS_DIAG_PAD_COUNT
#define NLOS_DIAG_PAD_COUNT 0
#endif

__device__ __forceinline__ uint32_t make_entry(const GridView& g, const BBoxF& bb, int xx, int yy, uint32_t zq, int k, float ms) {
    // the box is widened by ms x (kSubFrac sub-cells + kSlopeAbs) (g.padx, g.pady: in sub-cells; ms = graze_scale)
    const float px = g.padx * ms, py = g.pady * ms;
    const float fx0 = ((bb.x0 - g.gx0) * g.inv_cw - (float)xx) * (float)kSub - px;
    const float fx1 = ((bb.x1 - g.gx0) * g.inv_cw - (float)xx) * (float)kSub + px;
    const float fy0 = ((bb.y0 - g.gy0) * g.inv_ch - (float)yy) * (float)kSub - py;
    const float fy1 = ((bb.y1 - g.gy0) * g.inv_ch - (float)yy) * (float)kSub + py;
    const int a0 = min(max((int)floorf(fx0), 0), kSub - 1), a1 = min(max((int)floorf(fx1), 0), kSub - 1);
    const int b0 = min(max((int)floorf(fy0), 0), kSub - 1), b1 = min(max((int)floorf(fy1), 0), kSub - 1);
    const uint32_t xm = (2u << a1) - (1u << a0), ym = (2u << b1) - (1u << b0);
    return (zq << (g.ib + 2 * kSub)) | (ym << (g.ib + kSub)) | (xm << g.ib) | (uint32_t)k;
}

// The same entry from the triangle's box in GLOBAL sub-cell coordinates (round 6): the four floors are taken once per triangle
// (sub_box()), an entry pays two integer subtractions and clamps per axis instead of four float subtract / scale / floor /
// convert chains.  floor(((A - xx) kSub) - p) and floor(A kSub - p) - kSub xx differ by the rounding of A kSub (A < 96 cells:
// 3e-5 sub-cells); kSubSlack on either side covers it -- the masks may only be too LARGE (they are a filter in front of the
// exact test).
struct SubBox { int x0, x1, y0, y1; };
constexpr float kSubSlack = 1e-4f;
__device__ __forceinline__ SubBox sub_box(const GridView& g, const BBoxF& bb, float ms) {
    const float px = g.padx * ms + kSubSlack, py = g.pady * ms + kSubSlack;
    SubBox s;
    s.x0 = (int)floorf((bb.x0 - g.gx0) * g.inv_cw * (float)kSub - px);
    s.x1 = (int)floorf((bb.x1 - g.gx0) * g.inv_cw * (float)kSub + px);
    s.y0 = (int)floorf((bb.y0 - g.gy0) * g.inv_ch * (float)kSub - py);
    s.y1 = (int)floorf((bb.y1 - g.gy0) * g.inv_ch * (float)kSub + py);
    return s;
}
__device__ __forceinline__ uint32_t make_entry_i(const GridView& g, const SubBox& sb, int xx, int yy, uint32_t zq, int k) {
    const int ox = xx * kSub, oy = yy * kSub;
    const int a0 = min(max(sb.x0 - ox, 0), kSub - 1), a1 = min(max(sb.x1 - ox, 0), kSub - 1);
    const int b0 = min(max(sb.y0 - oy, 0), kSub - 1), b1 = min(max(sb.y1 - oy, 0), kSub - 1);
    const uint32_t xm = (2u << a1) - (1u << a0), ym = (2u << b1) - (1u << b0);
    return (zq << (g.ib + 2 * kSub)) | (ym << (g.ib + kSub)) | (xm << g.ib) | (uint32_t)k;
}

// conservative rasterisation of a projected triangle: fn(xx, yy) for every overlapped cell.  `pre(cx0, cx1, cy0, cy1)`
// sees the cell range of the bounding box first and may drop the triangle before the edge functions are set up.
// Cells are `cw x ch` wide from (gx0, gy0); `Rx` cells per side.  An edge function is evaluated at the corner of
// the cell that lies deepest inside the half-plane and stepped from cell to cell by one addition: its drift over a
// bounding box (a few ulps of the largest term) is three orders of magnitude below the slack t_i, which is sized
// for the rounding of the projection.  (Round 1 re-evaluated the three edge functions from the cell coordinates
// in every cell: 30 VALU operations per cell against 7 here, and the build phases were 41 % of the kernel.)
struct RasterAll { __device__ __forceinline__ bool operator()(int, int, int, int) const { return true; } };
template <bool BBOX = false, class Fn, class Pre = RasterAll>
__device__ __forceinline__ void raster_cells(float gx0, float gy0, float inv_cw, float inv_ch, int Rx, const Proj2& q, float ms, Fn fn,
                                             Pre pre = Pre(), unsigned long long* cnt = nullptr) {
    const float cw = __builtin_amdgcn_rcpf(inv_cw), ch = __builtin_amdgcn_rcpf(inv_ch);
    // slack: the projection's rounding (1e-7) and, above all, the error of a reported hit under the grazing rule
    const float mgx = ms * (kBoxFrac * cw + kSlopeAbs), mgy = ms * (kBoxFrac * ch + kSlopeAbs);
    const int cx0 = cell_coord(fminf(fminf(q.ax, q.bx), q.cx) - mgx, gx0, inv_cw, Rx);
    const int cx1 = cell_coord(fmaxf(fmaxf(q.ax, q.bx), q.cx) + mgx, gx0, inv_cw, Rx);
    const int cy0 = cell_coord(fminf(fminf(q.ay, q.by), q.cy) - mgy, gy0, inv_ch, Rx);
    const int cy1 = cell_coord(fmaxf(fmaxf(q.ay, q.by), q.cy) + mgy, gy0, inv_ch, Rx);
    if (!pre(cx0, cx1, cy0, cy1)) return;
    if (BBOX) {                     // bounding box only (conservative superset of the cells the triangle enters)
        for (int yy = cy0; yy <= cy1; ++yy)
            for (int xx = cx0; xx <= cx1; ++xx) fn(xx, yy);
        return;
    }
    {   // edge functions, oriented so that the inside is >= 0.  (Conservative arithmetic, so `#pragma clang fp contract(fast)`
        // would be legitimate here and in make_entry(): 20 of the counting pass's 497 instructions, and no measurable
        // time -- forward 1.346 vs 1.345 ms, profiles/r04_ab_contract_freeh.log; not kept.)
    const float area = (q.bx - q.ax) * (q.cy - q.ay) - (q.by - q.ay) * (q.cx - q.ax);
    const float sgn = area < 0.0f ? -1.0f : 1.0f;
    const bool thin = fabsf(area) < 1e-4f * cw * ch;         // edge-on: bbox cells only
    const float A0 = -(q.by - q.ay) * sgn, B0 = (q.bx - q.ax) * sgn, C0 = -(A0 * q.ax + B0 * q.ay);
    const float A1 = -(q.cy - q.by) * sgn, B1 = (q.cx - q.bx) * sgn, C1 = -(A1 * q.bx + B1 * q.by);
    const float A2 = -(q.ay - q.cy) * sgn, B2 = (q.ax - q.cx) * sgn, C2 = -(A2 * q.cx + B2 * q.cy);
    const float ecw = ms * (kEdgeFrac * cw + kSlopeAbs), ech = ms * (kEdgeFrac * ch + kSlopeAbs);
    const float t0 = fabsf(A0) * ecw + fabsf(B0) * ech;
    const float t1 = fabsf(A1) * ecw + fabsf(B1) * ech;
    const float t2 = fabsf(A2) * ecw + fabsf(B2) * ech;
    // value at the first cell's inside-most corner (+ slack), and the steps per cell
    const float x00 = gx0 + (float)cx0 * cw, y00 = gy0 + (float)cy0 * ch;
    float E0 = A0 * (x00 + (A0 > 0 ? cw : 0.0f)) + B0 * (y00 + (B0 > 0 ? ch : 0.0f)) + (C0 + t0);
    float E1 = A1 * (x00 + (A1 > 0 ? cw : 0.0f)) + B1 * (y00 + (B1 > 0 ? ch : 0.0f)) + (C1 + t1);
    float E2 = A2 * (x00 + (A2 > 0 ? cw : 0.0f)) + B2 * (y00 + (B2 > 0 ? ch : 0.0f)) + (C2 + t2);
    const float ax0 = A0 * cw, ax1 = A1 * cw, ax2 = A2 * cw;
    const float by0 = B0 * ch, by1 = B1 * ch, by2 = B2 * ch;
    for (int yy = cy0; yy <= cy1; ++yy) {
        float r0 = E0, r1 = E1, r2 = E2;
        for (int xx = cx0; xx <= cx1; ++xx) {
#ifdef NLOS_FWD_STAMPS
            if (cnt) { cnt[0] += 1; if ((int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) cnt[1] += 1; }   // lane- and wave-iterations
#endif
            if (thin | ((r0 >= 0.0f) & (r1 >= 0.0f) & (r2 >= 0.0f))) fn(xx, yy);
            r0 += ax0; r1 += ax1; r2 += ax2;
        }
        E0 += by0; E1 += by1; E2 += by2;
    }
    }
}
// first cell of the bounding box as raster_cells() derives it (same expressions, same bits)
__device__ __forceinline__ void raster_origin(const GridView& g, const Proj2& q, float ms, int& cx0, int& cy0) {
    const float cw = __builtin_amdgcn_rcpf(g.inv_cw), ch = __builtin_amdgcn_rcpf(g.inv_ch);
    const float mgx = ms * (kBoxFrac * cw + kSlopeAbs), mgy = ms * (kBoxFrac * ch + kSlopeAbs);
    cx0 = cell_coord(fminf(fminf(q.ax, q.bx), q.cx) - mgx, g.gx0, g.inv_cw, g.R);
    cy0 = cell_coord(fminf(fminf(q.ay, q.by), q.cy) - mgy, g.gy0, g.inv_ch, g.R);
}
template <class Fn, class Pre = RasterAll>
__device__ __forceinline__ void raster_tri(const GridView& g, const Proj2& q, float ms, Fn fn, Pre pre = Pre(), unsigned long long* cnt = nullptr) {
    raster_cells(g.gx0, g.gy0, g.inv_cw, g.inv_ch, g.R, q, ms, fn, pre, cnt);
}
// the same on the coarse map of the depth bounds (one cell = 2 x 2 cells of the grid, R2 = (R + 1) / 2 per side): a
// coarse cell is touched iff one of its four cells is (up to the slack), at a quarter of the cells
template <class Fn>
__device__ __forceinline__ void raster_tri_coarse(const GridView& g, int R2, const Proj2& q, Fn fn) {
    // (the rays towards a LIVE face end inside its projection up to rounding: base margins; what scales with the
    // face's grazing angle is the depth its own hits are reported at, see zb at the call sites)
    // From the bounding box alone (round 3): a coarse cell is 2 x 2 cells and a triangle about one, so its box covers
    // little more than the triangle does, and a depth bound may only be too LARGE (it culls less, never wrongly); the
    // edge functions cost the pass half of its instructions.  Forward 1.358 -> 1.341 ms (profiles/r03_ab_zc_bbox.log).
#ifdef NLOS_DIAG_ZC_EDGES          // diagnostic builds only: the exact coverage
    raster_cells(g.gx0, g.gy0, 0.5f * g.inv_cw, 0.5f * g.inv_ch, R2, q, 1.0f, fn);
#else
    raster_cells<true>(g.gx0, g.gy0, 0.5f * g.inv_cw, 0.5f * g.inv_ch, R2, q, 1.0f, fn);
#endif
}

// (SourceFrame / source_frame(): render_common.h -- pass 2 asks the same question about a source's arithmetic range)

// Tile binning for the tiled grid: one workgroup per source appends every triangle to the subset of each
// slope-space tile its projected bounding box meets (with the rasteriser's margin).  O(F) per source -- the
// tiles' workgroups then read their subset instead of scanning the whole mesh each.
__global__ __launch_bounds__(512) void k_tile_bin(ForwardArgs a, int R, int from_sensor) {
    const int l = blockIdx.x;
    const V3 o = ld3((from_sensor ? a.src.sensor : a.src.origin) + 3 * (size_t)l);
    const SourceFrame fr = source_frame(a.sc.nodes, o);
    if (!fr.ok) return;                                   // tile 0 handles such a source alone
    const int ntx = a.tiles_x, nty = a.tiles_y;
    const float tw = fr.wx * 1.002f / (float)ntx, th = fr.wy * 1.002f / (float)nty;
    const float inv_tw = 1.0f / tw, inv_th = 1.0f / th;
    const float mx0 = kEdgeFrac * tw / (float)R + kSlopeAbs, my0 = kEdgeFrac * th / (float)R + kSlopeAbs;
    // slot allocation with LDS counters (one workgroup owns the whole source): global atomics on the few
    // per-tile counters were the bottleneck (2.5 ms for 1024 sources x 20 k faces)
    __shared__ int s_cnt[1024];
    const int ntile = ntx * nty;
    for (int i = threadIdx.x; i < ntile; i += blockDim.x) s_cnt[i] = 0;
    __syncthreads();
    uint32_t* lists = a.tile_list + (size_t)l * ntile * a.tile_cap;
    for (int j = threadIdx.x; j < a.sc.F; j += blockDim.x) {
        const float4 q0 = a.sc.facerec[4 * j], q1 = a.sc.facerec[4 * j + 1], q2 = a.sc.facerec[4 * j + 2];
        const Proj2 q = project_tri(o, mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y), mk(q1.z, q1.w, q2.x));
        const float4 q3 = a.sc.facerec[4 * j + 3];
        const float ms = graze_scale(o, mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y), mk(q1.z, q1.w, q2.x), mk(q3.y, q3.z, q3.w), fr.hmin);
        const float mx = ms * mx0, my = ms * my0;
        const float bx0 = fminf(fminf(q.ax, q.bx), q.cx) - mx, bx1 = fmaxf(fmaxf(q.ax, q.bx), q.cx) + mx;
        const float by0 = fminf(fminf(q.ay, q.by), q.cy) - my, by1 = fmaxf(fmaxf(q.ay, q.by), q.cy) + my;
        // one extra tile on each side covers the rounding of the tile origins (gx0 + t * tw)
        const int t0 = max((int)floorf((bx0 - fr.gx0) * inv_tw - 1e-3f), 0), t1 = min((int)floorf((bx1 - fr.gx0) * inv_tw + 1e-3f), ntx - 1);
        const int u0 = max((int)floorf((by0 - fr.gy0) * inv_th - 1e-3f), 0), u1 = min((int)floorf((by1 - fr.gy0) * inv_th + 1e-3f), nty - 1);
        for (int u = u0; u <= u1; ++u)
            for (int t = t0; t <= t1; ++t) {
                const int ti = u * ntx + t;
                const int pos = atomicAdd(&s_cnt[ti], 1);
                if (pos < a.tile_cap) lists[(size_t)ti * a.tile_cap + pos] = (uint32_t)j;
            }
    }
    __syncthreads();
    int* cnt = a.tile_count + (size_t)l * ntile;
    for (int i = threadIdx.x; i < ntile; i += blockDim.x) cnt[i] = s_cnt[i];
}

// two 512-thread workgroups per CU = 4 waves per SIMD: keep the kernel within 128 VGPRs
// NCM (row N, non-confocal pairs): 0 = confocal; 1 = visibility-only pass from the SENSOR of each pair
// (bits -> a.vis2, no histogram); 2 = pass from the LASER that evaluates both legs' geometry, ANDs the
// sensor-leg bits and traces only the laser leg.  One perspective grid serves one origin, so a pair costs
// two grid passes (about 2x the confocal forward) instead of two BVH traversals per sample (9x).
//
// TILED (meshes whose cell lists do not fit one workgroup's LDS, F > ~7.6 k): slope space is cut into
// tiles_x * tiles_y tiles and one workgroup handles one (source, tile): it selects the triangles whose
// projected bounding box meets its tile (ids -> a.tile_list, at most 8191: the 13-bit entry index is then
// an index into that list), builds the grid over the tile only, and traces exactly the samples whose
// slope point falls into the tile -- every sample is owned by one tile, decided from the source's global
// frame so that all workgroups of a source agree.  Rows and visibility words are combined with atomics
// (the launcher zeroes them).  A tile whose subset overflows iterates over all faces and uses the BVH
// query for its own samples; a source whose scene is not strictly in front is handled by tile 0 alone.
// Workgroups (sources, or tiles) whose cell lists overflow the normal LDS share (two workgroups per CU) start
// over on a coarser grid (grid_body<COARSE = true>, same workgroup, same launch); if even that does not fit they
// flag themselves in a.retry and leave before binning anything, and a second launch (`pass` = 1) with one
// workgroup per CU and ~150 KB of LDS redoes exactly those (grazing views pile thousands of sliver triangles
// into a few cells / tiles).
// The body of the grid kernel.  COARSE = false: R is the launch constant (the common path; keeping it constant
// is worth 1.5 % there).  When the cell lists overflow the entry capacity it returns true before binning
// anything, and the kernel runs the COARSE = true instance for the same source in the same workgroup: that one
// starts at 3/4 of the resolution and coarsens further (x 3/4, NLOS_MAX_COARSEN times) until the lists fit.
// Only what still does not fit is flagged for the big-LDS launch.  (Redoing such sources in a separate launch was
// tried: a few heavy workgroups fill the chip badly -- 4.4 vs 3.2 ms where a third of the sources overflow.)
template <int FEAT, int NCM, bool TILED, bool COARSE>
__device__ __forceinline__ bool grid_body(const ForwardArgs& a, const int rows_in_lds, const int R_launch, const int cap, const int pass,
                                          const int last_pass, uint32_t* s_scan, uint32_t* s_bkt, const unsigned slot) {
    constexpr int kCoarsen = COARSE ? NLOS_MAX_COARSEN : 0;
    // R_launch fixes the LDS layout; the grid actually used (R, R2, ncell) may be coarsened below when the
    // cell lists of this source do not fit
    int R = R_launch;
    // dynamic LDS: [ctl: ticket, bad, total, n_live (16 B)][row nbins f64][cells R*R+1 u32]
    //   [union { build: depth bound per 2x2 cells R2*R2 u32, block masks nblk u64 ;
    //            trace: 8 waves x (128 queued pairs + 2 mask words) }][entries cap u32]
    // (the bucketed live-face list lives in global scratch: it is read once per 64-face block)
    extern __shared__ double s_lds[];
    constexpr int IB = TILED ? kIdxBitsTiled : kIdxBits;
    int* s_ctl = reinterpret_cast<int*>(s_lds);      // 8 ints: ticket, bad, total entries, n_live, tile subset size
    double* s_row = s_lds + 4;
    const int nbins = a.sp.nbins;
    int ncell = R * R;
    int R2 = (R + 1) >> 1;
    const int F = a.sc.F;
    const int ntiles = TILED ? a.tiles_x * a.tiles_y : 1;
    const int tile = TILED ? (int)(slot % (unsigned)ntiles) : 0;       // (TILED: wgid == slot == blockIdx.x)
    const int tile_x = TILED ? tile % a.tiles_x : 0, tile_y = TILED ? tile / a.tiles_x : 0;
    const int mask_blocks = TILED ? (a.tile_cap + 63) >> 6 : (F + 63) >> 6;   // LDS sizing only (tiles that visit all faces keep no masks)
    uint32_t* s_cell = reinterpret_cast<uint32_t*>(s_row + (rows_in_lds ? nbins : 0));
    uint32_t* s_union = s_cell + ((ncell + 2) & ~1);
    uint32_t* s_zc = s_union;                                                   // build phase
    unsigned long long* s_mask = reinterpret_cast<unsigned long long*>(s_zc + ((R2 * R2 + 1) & ~1));
    uint32_t* s_queue = s_union;                                                // trace phase
    const int union_words = max(((R2 * R2 + 1) & ~1) + 2 * mask_blocks, kGridWaves * kQueueWords);
    uint32_t* s_ent = s_union + ((union_words + 1) & ~1);
    // build phase only: list length per cell (clamped to 255) in the part of the union the masks leave free,
    // so that the fill pass can bucket the live faces while it has their projection at hand
    uint8_t* s_len8 = reinterpret_cast<uint8_t*>(s_mask + mask_blocks);
    const bool len_ok = (((R2 * R2 + 1) & ~1) + 2 * mask_blocks) * 4 + ncell <= union_words * 4;
    // build-phase scratch of this workgroup: bucketed live list, per-triangle cell coverage.  (A pool of ~1500 regions
    // handed from workgroup to workgroup instead of one region per source was tried, to keep the scratch in cache:
    // the regions then migrate between the XCDs' L2s -- 1.75 GB of fabric traffic per launch instead of 0.49 GB and
    // 2.29 ms instead of 1.91 ms.)
    // (one workgroup per source: the source this workgroup renders, in the launcher's order when it has one -- wgid replaces
    // blockIdx.x everywhere below; the tiled grid keeps its (source, tile) numbering)
    // `slot`: blockIdx.x
    const unsigned wgid = (!TILED && a.perm) ? (unsigned)a.perm[slot] : slot;
    uint16_t* g_live = a.live + (size_t)wgid * (TILED ? a.tile_cap : F);
    uint16_t* g_cov = a.cov + (size_t)wgid * (TILED ? a.tile_cap : F);     // per-triangle cell coverage, count -> fill pass
    uint32_t* tl = TILED ? a.tile_list + (size_t)wgid * a.tile_cap : nullptr;

    const int l = TILED ? (int)(wgid / (unsigned)ntiles) : (int)wgid;
    const int tid = threadIdx.x, NT = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nwaves = NT >> 6;
    const V3 o = ld3((NCM == 1 ? a.src.sensor : a.src.origin) + 3 * (size_t)l);
    const V3 on = ld3((NCM == 1 ? a.src.sensor_normal : a.src.normal) + 3 * (size_t)l);
    const V3 ob = NCM == 2 ? ld3(a.src.sensor + 3 * (size_t)l) : o;
    const V3 onb = NCM == 2 ? ld3(a.src.sensor_normal + 3 * (size_t)l) : on;
    uint32_t* const visout = NCM == 1 ? a.vis2 : a.vis;
    // item-mask layout of the visibility cache (launcher: single-workgroup grid, confocal; then a.vis == nullptr)
    // geometry cache for pass 2 (confocal renders that record item masks: its index is the item masks' ray index)
    float* const geo_l = (!TILED && NCM == 0 && a.geo && a.vis_items) ? a.geo + 4 * (size_t)l * (size_t)a.geo_stride : nullptr;
    float* const geo_w = geo_l ? a.geo + 4 * (size_t)a.geo_sources * (size_t)a.geo_stride + 2 * (size_t)l * (size_t)a.geo_stride : nullptr;
    // (the laser pass of non-confocal pairs records the pair's accepted samples the same way, round 4)
    unsigned long long* const vitems = (!TILED && (NCM == 0 || NCM == 2) && a.vis_items) ? a.vis_items + (size_t)l * (size_t)a.items_stride : nullptr;
#ifdef NLOS_FWD_STAMPS
    // diagnostic build only: per-phase cycles summed over workgroups -> a.dbg[0..5]
    long long t_prev = clock64();
    int t_slot = 0;
    const long long c_start = t_prev, w_start = wall_clock64();   // shader-clock ticks vs the constant 100 MHz counter -> a.dbg[24], [25]
#define FWD_STAMP() do { __syncthreads(); if (tid == 0 && a.dbg) { long long t_now = clock64(); atomicAdd((unsigned long long*)&a.dbg[t_slot], (unsigned long long)(t_now - t_prev)); ++t_slot; t_prev = t_now; } } while (0)
    // round 6: trip counts of the build loops (lane- and wave-iterations) for the dynamic instruction histogram
    unsigned long long c_cnt[2] = {0ull, 0ull}, c_cfw = 0ull, c_fill[2] = {0ull, 0ull}, c_ffw = 0ull;
#define FWD_WAVE_FIRST() ((int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1)
#elif defined(NLOS_FWD_STAMPS_LIGHT)
    // diagnostic build only (round 6): the six phase stamps alone, taken by thread 0 at barriers the kernel has anyway -- no per-item
    // clocks, no counters: the phase shares of a workgroup's life in a build that runs at the product's speed (the full stamps
    // triple the trace's time and with it every share: profiles/r06_light_stamps.log)
    long long t_prev = clock64();
    int t_slot = 0;
    const long long c_start = t_prev;
#define FWD_STAMP() do { if (tid == 0 && a.dbg) { long long t_now = clock64(); atomicAdd((unsigned long long*)&a.dbg[t_slot], (unsigned long long)(t_now - t_prev)); ++t_slot; t_prev = t_now; } } while (0)
#else
#define FWD_STAMP() do { } while (0)
#endif

    // ---- grid frame from the (padded) root box of the BVH: O(1) per source ------------------
    const SourceFrame fr = source_frame(a.sc.nodes, o);
    const float zr0 = fr.zr0, zr1 = fr.zr1;
    const bool frame_ok = fr.ok;
    GridView g;
    g.R = R;
    float Gx0 = 0.0f, Gy0 = 0.0f, inv_tw = 0.0f, inv_th = 0.0f;     // TILED: the source's global frame
    {
        const float wx = fr.wx, wy = fr.wy;
        g.gx0 = fr.gx0;
        g.gy0 = fr.gy0;
        g.inv_cw = (float)R / (wx * 1.002f);
        g.inv_ch = (float)R / (wy * 1.002f);
        if (TILED) {
            // the source's frame [G0, G0 + W) is cut into tiles; ownership of a slope point is decided
            // with (G0, inv_tw) only, which every workgroup of the source computes identically
            Gx0 = g.gx0; Gy0 = g.gy0;
            const float tw = wx * 1.002f / (float)a.tiles_x, th = wy * 1.002f / (float)a.tiles_y;
            inv_tw = 1.0f / tw; inv_th = 1.0f / th;
            g.gx0 = Gx0 + (float)tile_x * tw;
            g.gy0 = Gy0 + (float)tile_y * th;
            g.inv_cw = (float)R / tw;
            g.inv_ch = (float)R / th;
        }
        g.padx = kSubFrac + kSlopeAbs * (float)kSub * g.inv_cw;
        g.pady = kSubFrac + kSlopeAbs * (float)kSub * g.inv_ch;
        g.ib = IB;
        g.zmax = (1 << (32 - 2 * kSub - IB)) - 1;
        g.z0 = zr0;
        g.inv_qz = (float)g.zmax / fmaxf(zr1 - zr0, 1e-12f);
    }
    // a coarser grid over the same frame (fewer, longer cell lists), inside the LDS laid out for R_launch
    auto set_grid_res = [&](int Rn) {
        R = Rn; ncell = Rn * Rn; R2 = (Rn + 1) >> 1;
        g.R = Rn;
        if (TILED) {
            g.inv_cw = (float)Rn / (fr.wx * 1.002f / (float)a.tiles_x);
            g.inv_ch = (float)Rn / (fr.wy * 1.002f / (float)a.tiles_y);
        } else {
            g.inv_cw = (float)Rn / (fr.wx * 1.002f);
            g.inv_ch = (float)Rn / (fr.wy * 1.002f);
        }
        g.padx = kSubFrac + kSlopeAbs * (float)kSub * g.inv_cw;
        g.pady = kSubFrac + kSlopeAbs * (float)kSub * g.inv_ch;
    };

    if (!TILED && !frame_ok && a.need_tree && a.retry && pass < last_pass) {
        // lazy scene build: the tree does not exist yet.  This source (scene not strictly in front of its wall point)
        // needs the BVH query: flag it for the second launch, in front of which the tree is completed
        if (tid == 0) {
            a.retry[wgid] = pass + 1;
            __hip_atomic_store(a.need_tree, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return false;
    }
    if (COARSE) set_grid_res(max(8, (R * 3) >> 2));      // the lists overflowed at R_launch: start one step coarser
    if (rows_in_lds)
        for (int i = tid; i < nbins; i += NT) s_row[i] = 0.0;
    for (int i = tid; i <= ncell; i += NT) s_cell[i] = 0u;
    for (int i = tid; i < R2 * R2; i += NT) s_zc[i] = 0u;
    if (tid == 0) { s_ctl[0] = 0; s_ctl[1] = frame_ok ? 0 : 1; s_ctl[2] = 0; s_ctl[3] = 0; s_ctl[4] = 0; s_ctl[5] = 0; s_ctl[6] = 0; }
    if (tid < 32) s_bkt[tid] = 0u;
    __syncthreads();

    // Fl faces are iterated by this workgroup; `ident`: local index == sorted face index
    int Fl = F;
    bool ident = true;
    if (TILED) {
        if (!frame_ok) {
            if (tile != 0) return false;               // tile 0 handles such a source alone (BVH queries)
        } else {
            // ---- tile subset, binned by k_tile_bin
            const int nsel = a.tile_count[wgid];
#ifdef NLOS_FWD_STAMPS
            if (tid == 0 && a.dbg) {
                if (nsel > a.tile_cap) atomicAdd((unsigned long long*)&a.dbg[22], 1ull);
                atomicMax((unsigned long long*)&a.dbg[23], (unsigned long long)nsel);
            }
#endif
            if (nsel > a.tile_cap || nsel > (1 << IB) - 1) {
                if (tid == 0) s_ctl[1] = 1;            // subset overflow: all faces, BVH query, own samples only
            } else {
                Fl = nsel;
                ident = false;
            }
            __syncthreads();
        }
    }
    const int nblocks = (Fl + 63) >> 6;
    auto gid = [&](int j) -> int { return (TILED && !ident) ? (int)tl[j] : j; };
    const bool compact = !(TILED && ident);            // ident tiles may exceed the u16 live list: no compaction

    // ---- which faces can contribute at all?  (order-preserving compaction, per 64-face block) ----
    // With face normals and the clamped form factor, -dot(n,dir) has the sign of dist(o, plane(f))
    // for every sample of f: if the wall point is clearly behind the face (and the face in front
    // of the wall), every contribution is exactly 0 -- nothing to sample, nothing to trace.
    // Live faces also record, per 2x2 block of cells they project to, the largest depth at which
    // a ray of this source can end.
    auto face_dark = [&](const Face& f) -> bool {
        bool dark = f.degenerate;
        if (!dark && a.sp.clamp) {
            const float sc = fabsf(o.x - f.p0.x) + fabsf(o.y - f.p0.y) + fabsf(o.z - f.p0.z);
            const bool infront = dot(on, f.p0 - o) > 1e-4f * sc && dot(on, f.p1 - o) > 1e-4f * sc &&
                                 dot(on, f.p2 - o) > 1e-4f * sc;
            bool behind;
            if (!(FEAT & FEAT_VN)) {
                behind = dot(f.fn, o - f.p0) < -1e-4f * sc;
            } else {
                // interpolated normals: n . (o - p) = sum_i b_i n_i . (o - p) is linear in p for each n_i, so it is
                // negative on the whole face if it is at the three corners, for all three vertex normals
                // (b_i >= 0 up to the rounding of the hit's barycentrics, 1e-7 against the 1e-4 margin)
                const float* vn = a.sc.vertex_normal;
                const V3 n3[3] = {ld3(vn + 3 * (size_t)f.i0), ld3(vn + 3 * (size_t)f.i1), ld3(vn + 3 * (size_t)f.i2)};
                const V3 d0 = o - f.p0, d1 = o - f.p1, d2 = o - f.p2;
                const float sc2 = fmaxf(sc, fmaxf(fabsf(d1.x) + fabsf(d1.y) + fabsf(d1.z), fabsf(d2.x) + fabsf(d2.y) + fabsf(d2.z)));
                behind = true;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const float m = -1e-4f * sc2 * (fabsf(n3[i].x) + fabsf(n3[i].y) + fabsf(n3[i].z));
                    behind = behind && dot(n3[i], d0) < m && dot(n3[i], d1) < m && dot(n3[i], d2) < m;
                }
            }
            dark = behind && infront;
        }
        return dark;
    };
    for (int b = wave; b < nblocks; b += nwaves) {
        const int j = (b << 6) + lane;
        bool live = false;
        if (j < Fl) {
            const int jg = gid(j);
            Face f;
            Tri tr_unused;
            load_face_tri<FEAT>(a.sc, jg, f, tr_unused);
            const bool dark = face_dark(f);
            live = !dark;
            // The margin scale of this face as an occluder (graze_scale(): 1 unless the source sees it at |cos| < 2^-6),
            // evaluated HERE, once, where the vertices and the normal are in registers, and handed to the counting and
            // the fill pass as a 16-bit fixed-point number (64ths, rounded up) in the live list's scratch, which is free
            // until the bucket placement behind the fill pass.  (Testing in those passes instead -- a dependent record
            // load and a divergent region in two thirds of their wave iterations -- cost 7 % of the kernel.)
            const float msv = graze_scale(o, f.p0, f.p1, f.p2, f.fn, fr.hmin);
            if (frame_ok && !(TILED && ident)) g_live[j] = (uint16_t)ceilf(msv * 64.0f);
#ifdef NLOS_DIAG_NO_DARKZERO       // diagnostic builds only
            if (false) {
#else
            if (!TILED && dark && visout) {
#endif
                // dark faces: no sample is ever accepted (live faces receive their words from the trace)
                uint32_t* visp = visout + ((size_t)l * a.vis_words) * F + j;
                for (int wi = 0; wi < a.vis_words; ++wi) visp[(size_t)wi * F] = 0u;
            }
#ifdef NLOS_DIAG_NO_ZC             // diagnostic builds only
            if (false) {
#else
            if (live && frame_ok && !(TILED && ident)) {
#endif
                // the largest depth at which an own hit of this face can be REPORTED: its farthest vertex, plus the error
                // of t at the face's grazing angle
                const float zfar = fmaxf(fmaxf(f.p0.z, f.p1.z), f.p2.z) - o.z;
                const float msf = msv;
                const uint32_t zb = __float_as_uint(fmaxf(zfar, 0.0f) * (1.0f + kDepthEps * msf) + 1e-30f);
                const Proj2 q = project_tri(o, f.p0, f.p1, f.p2);
                raster_tri_coarse(g, R2, q, [&](int cx, int cy) { atomicMax(&s_zc[cy * R2 + cx], zb); });
            }
        }
        const unsigned long long m = __ballot(live);
        if (lane == 0 && compact) s_mask[b] = m;      // (a tile that visits all F faces re-derives `dark` in the trace)
    }
    __syncthreads();
    FWD_STAMP();   // 0: setup + live-face masks + depth bounds
    if (tid == 0 && compact) {
        uint32_t run = 0;
        for (int b = 0; b < nblocks; ++b) run += (uint32_t)__popcll(s_mask[b]);
        s_ctl[3] = (int)run;
    }
    // A triangle that lies deeper than every depth bound under its bounding box enters no cell: skip it before the
    // edge functions are set up.  (The far side of the object is behind the live faces of its cells, and Morton
    // order keeps such triangles together -- whole waves leave here.)
    auto reachable = [&](uint32_t zn) {
        return [&, zn](int cx0, int cx1, int cy0, int cy1) -> bool {
            bool any = false;
            for (int yy = cy0 >> 1; yy <= (cy1 >> 1); ++yy)
                for (int xx = cx0 >> 1; xx <= (cx1 >> 1); ++xx) any = any || zn <= s_zc[yy * R2 + xx];
            return any;
        };
    };
    // ---- count + scan; if the cell lists do not fit the entry capacity, coarsen the grid (x 3/4, up to
    // twice) and count again: fewer cells per triangle, longer lists.  That costs one more depth-bound and
    // counting pass (~15 % of the kernel) where the alternative is the big-LDS relaunch at half the occupancy.
    for (int attempt = 0;; ++attempt) {
        if (attempt > 0) {
            set_grid_res(max(8, (R * 3) >> 2));
            for (int i = tid; i <= ncell; i += NT) s_cell[i] = 0u;
            for (int i = tid; i < R2 * R2; i += NT) s_zc[i] = 0u;
            if (tid == 0) s_ctl[1] = 0;
            __syncthreads();
            for (int b = wave; b < nblocks; b += nwaves) {
                const int j = (b << 6) + lane;
                if (j < Fl && ((s_mask[b] >> lane) & 1ull)) {
                    const int jg = gid(j);
                    const float4 q0 = a.sc.facerec[4 * jg], q1 = a.sc.facerec[4 * jg + 1], q2 = a.sc.facerec[4 * jg + 2];
                    const float zfar = fmaxf(fmaxf(q0.z, q1.y), q2.x) - o.z;
                    const float msf = (float)g_live[j] * (1.0f / 64.0f);          // the setup pass's margin scale of this face
                    const uint32_t zb = __float_as_uint(fmaxf(zfar, 0.0f) * (1.0f + kDepthEps * msf) + 1e-30f);
                    const Proj2 q = project_tri(o, mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y), mk(q1.z, q1.w, q2.x));
                    raster_tri_coarse(g, R2, q, [&](int cx, int cy) { atomicMax(&s_zc[cy * R2 + cx], zb); });
                }
            }
            __syncthreads();
        }
        if (frame_ok && !(TILED && ident)) {
            // ---- counting pass -----------------------------------------------------------------------
            int pend_jl = -1;
            uint16_t pend_cov = 0;
            // (Round 6, measured and removed: requesting the NEXT triangle's record and margin scale before this one is rasterised,
            // here and in the fill pass -- 1.299 -> 1.300 ms for this pass alone, 1.313 ms with the fill pass as well: the build
            // phases are 45 % of a workgroup's life (profiles/r06_light_stamps.log) but not because of these round trips.)
            // (Round 4, measured and removed: a pre-pass that applies the reach test to every triangle and compacts the
            // survivors so that the pass proper runs on dense lanes -- the pre-pass repeats projection, bounding box and
            // reach test, and costs more than the idle lanes did: forward 1.334 -> 1.433 ms, profiles/r04_ab_count_compact.log.)
#if defined(NLOS_DIAG_NO_COUNT)    // diagnostic builds only (tools/ab_pmc.sh)
            for (int jl = Fl; jl < Fl; jl += NT) {
#else
            for (int jl = tid; jl < Fl; jl += NT) {
#endif
                const int j = gid(jl);
                diag_pad<NLOS_DIAG_PAD_COUNT>();
#ifdef NLOS_FWD_STAMPS
                if (attempt == 0 && FWD_WAVE_FIRST()) c_cfw += 1;
#endif
                const float4 q0 = a.sc.facerec[4 * j], q1 = a.sc.facerec[4 * j + 1], q2 = a.sc.facerec[4 * j + 2];
                // the setup pass's margin scale of this triangle -- requested BEFORE the store below: loads and stores
                // return in order (vmcnt), so a load behind the store would wait for the store's round trip to the L2
                const uint16_t msq = g_live[jl];
                if (pend_jl >= 0) g_cov[pend_jl] = pend_cov;       // the previous triangle's cells, behind this one's loads (see the trace)
                // margins of this triangle as an occluder: x 1 unless the source sees it at less than 0.9 degrees
                const float ms = (float)msq * (1.0f / 64.0f);
                const Proj2 q = project_tri(o, mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y), mk(q1.z, q1.w, q2.x));
                // smallest depth at which a hit on it can be REPORTED (its nearest vertex, minus the error of t)
                const uint32_t zn = __float_as_uint(fmaxf(fminf(fminf(q0.z, q1.y), q2.x) - o.z, 0.0f) * (1.0f - kDepthEps * ms));
                // the cells the triangle enters are remembered for the fill pass (16 bits per triangle in global
                // scratch): a bit per cell of a bounding box of up to 4 x 4 cells, relative to its first cell (which
                // the fill pass re-derives from the projection, with the same margin scale); 0xFFFF = rasterise again
                // (larger boxes, and the rare box that is entered in all sixteen cells)
                int bx0 = 0, by0 = 0;
                bool big = false, reach = false;
                uint32_t cv = 0u;
                raster_tri(g, q, ms, [&](int xx, int yy) {
                    if (zn <= s_zc[(yy >> 1) * R2 + (xx >> 1)]) {
                        atomicAdd(&s_cell[yy * R + xx], 1u);
                        cv |= 1u << ((((yy - by0) & 3) << 2) + ((xx - bx0) & 3));
                    }
                }, [&](int cx0, int cx1, int cy0, int cy1) -> bool {
                    bx0 = cx0; by0 = cy0;
                    big = cx1 - cx0 > 3 || cy1 - cy0 > 3;
                    reach = reachable(zn)(cx0, cx1, cy0, cy1);
                    return reach;
#ifdef NLOS_FWD_STAMPS
                }, c_cnt);
#else
                });
#endif
                pend_jl = jl; pend_cov = (uint16_t)(!reach ? 0u : big ? 0xFFFFu : cv);
            }
            if (pend_jl >= 0) g_cov[pend_jl] = pend_cov;
        }
        __syncthreads();
        if (attempt == 0) FWD_STAMP();   // 1: counting pass
        if (frame_ok) {
            // ---- exclusive scan of the cell counts (each thread owns a contiguous slice) ---------
            const int per = (ncell + NT - 1) / NT;
            const int c0 = min(tid * per, ncell), c1 = min(c0 + per, ncell);
            uint32_t sum = 0;
            for (int c = c0; c < c1; ++c) sum += s_cell[c];
#ifdef NLOS_DIAG_HS_SCAN             // diagnostic builds only: round 1's Hillis-Steele scan over the workgroup (20 barriers)
            s_scan[tid] = sum;
            __syncthreads();
            for (int off = 1; off < NT; off <<= 1) {
                uint32_t v = tid >= off ? s_scan[tid - off] : 0u;
                __syncthreads();
                s_scan[tid] += v;
                __syncthreads();
            }
#else
            // inclusive scan of the per-thread sums: within the wave by lane shuffles, across the waves through their
            // totals in LDS (round 6: 2 barriers instead of 20)
            uint32_t incl = sum;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t up = __shfl_up(incl, off);
                if (lane >= off) incl += up;
            }
            if (lane == 63) s_scan[wave] = incl;
            __syncthreads();
            uint32_t base = 0u;
            for (int wv = 0; wv < wave; ++wv) base += s_scan[wv];
            __syncthreads();
            s_scan[tid] = incl + base;                       // (tid NT - 1 reads the grand total below)
            __syncthreads();
#endif
            uint32_t run = s_scan[tid] - sum;
            for (int c = c0; c < c1; ++c) {
                uint32_t n = s_cell[c];
                s_cell[c] = run;
                run += n;
                if (len_ok) s_len8[c] = (uint8_t)min(n, 255u);
            }
            if (tid == NT - 1) { s_ctl[2] = (int)s_scan[tid]; if ((int)s_scan[tid] > cap) s_ctl[1] = 1; }
        }
        __syncthreads();
        const bool overflow = frame_ok && !(TILED && ident) && s_ctl[1] != 0;
        if (!overflow || attempt >= kCoarsen || R <= 8) break;
        __syncthreads();                 // everyone has seen the flag before thread 0 clears it
    }
    if (frame_ok && !(TILED && ident) && s_ctl[1] != 0) {
        if (!COARSE && R > 8) return true;                     // cell lists overflow: once more, coarser, right here
        if (pass < last_pass && a.retry) {
            if (tid == 0) {
                a.retry[wgid] = pass + 1;      // still too many entries: redo in the big-LDS launch
                // (which may have to fall back to the BVH query: a lazily built scene gets its tree now)
                if (a.need_tree) __hip_atomic_store(a.need_tree, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return false;
        }
    }
    FWD_STAMP();   // 2: scans
    // ---- fill pass: s_cell[c] is the write cursor, afterwards the END of cell c ------------------
    // Live faces are bucketed here as well (longest list under the face's projected bounding box, see the
    // live list below): the bucket of the it-th face of a thread is kept as a nibble for the placement loop.
#ifndef NLOS_NB
#define NLOS_NB 16
#define NLOS_NB_SHIFT 2
#endif
    constexpr int NB = NLOS_NB;                              // buckets of (1 << NLOS_NB_SHIFT) entries
    const bool fill_buckets = len_ok && compact;
    unsigned long long nib0 = 0ull, nib1 = 0ull;
    if (frame_ok && s_ctl[1] == 0) {
        int it = 0;
#ifdef NLOS_DIAG_NO_FILL           // diagnostic builds only
        for (int jl = Fl; jl < Fl; jl += NT, ++it) {
#else
        for (int jl = tid; jl < Fl; jl += NT, ++it) {
#endif
            // what the counting pass found: nothing to enter (about half of the triangles: the far side of the
            // object), a mask of up to 4 x 4 cells, or a large bounding box that is rasterised again
            const uint32_t cov = g_cov[jl];
            const uint16_t msq = g_live[jl];
#ifdef NLOS_FWD_STAMPS
            if (FWD_WAVE_FIRST()) c_ffw += 1;
#endif
            const bool counts_as_live = fill_buckets && ((s_mask[jl >> 6] >> (jl & 63)) & 1ull);
            if (cov == 0u && !counts_as_live) continue;
            uint32_t nmax = 0u;                              // longest list among the cells the triangle enters
            if (cov) {
                const int j = gid(jl);
                const float4 q0 = a.sc.facerec[4 * j], q1 = a.sc.facerec[4 * j + 1], q2 = a.sc.facerec[4 * j + 2];
                const Proj2 q = project_tri(o, mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y), mk(q1.z, q1.w, q2.x));
                // the margin scale the counting pass used (the same 16-bit number)
                const float ms = (float)msq * (1.0f / 64.0f);
                const float zmin_rel = fmaxf(fminf(fminf(q0.z, q1.y), q2.x) - o.z, 0.0f) * (1.0f - kDepthEps * ms);
                const uint32_t zn = __float_as_uint(zmin_rel);
                const uint32_t zq = (uint32_t)min(max((int)floorf((zmin_rel - g.z0) * g.inv_qz) - 1, 0), g.zmax);
                BBoxF bb;
                bb.x0 = fminf(fminf(q.ax, q.bx), q.cx); bb.x1 = fmaxf(fmaxf(q.ax, q.bx), q.cx);
                bb.y0 = fminf(fminf(q.ay, q.by), q.cy); bb.y1 = fmaxf(fmaxf(q.ay, q.by), q.cy);
#ifndef NLOS_DIAG_FLOAT_ENTRY      // diagnostic builds only: the A/B of the integer entry
                const SubBox sbx = sub_box(g, bb, ms);
#define NLOS_MAKE_ENTRY(xx, yy) make_entry_i(g, sbx, xx, yy, zq, jl)
#else
#define NLOS_MAKE_ENTRY(xx, yy) make_entry(g, bb, xx, yy, zq, jl, ms)
#endif
                if (cov == 0xFFFFu) {
                    raster_tri(g, q, ms, [&](int xx, int yy) {
                        if (zn <= s_zc[(yy >> 1) * R2 + (xx >> 1)]) {
                            const int c = yy * R + xx;
                            uint32_t pos = atomicAdd(&s_cell[c], 1u);
                            s_ent[pos] = NLOS_MAKE_ENTRY(xx, yy);
                            if (len_ok) nmax = max(nmax, (uint32_t)s_len8[c]);
                        }
                    }, reachable(zn));
                } else {
                    int bx0, by0;
                    raster_origin(g, q, ms, bx0, by0);
                    for (uint32_t m = cov; m; m &= m - 1u) {
#ifdef NLOS_FWD_STAMPS
                        c_fill[0] += 1; if (FWD_WAVE_FIRST()) c_fill[1] += 1;
#endif
                        const int bit = __ffs((int)m) - 1;
                        const int xx = bx0 + (bit & 3), yy = by0 + (bit >> 2);
                        const int c = yy * R + xx;
                        uint32_t pos = atomicAdd(&s_cell[c], 1u);
                        s_ent[pos] = NLOS_MAKE_ENTRY(xx, yy);
                        if (len_ok) nmax = max(nmax, (uint32_t)s_len8[c]);
                    }
                }
            }
            if (counts_as_live) {
                const int bkt = (NB - 1) - (int)min(nmax >> NLOS_NB_SHIFT, (uint32_t)(NB - 1));   // bucket 0 = longest lists
                atomicAdd(&s_bkt[bkt], 1u);
                if (it < 16) nib0 |= (unsigned long long)bkt << (4 * it);
                else nib1 |= (unsigned long long)bkt << (4 * (it - 16));
            }
        }
    }
    __syncthreads();
    FWD_STAMP();   // 3: fill pass
    // ---- live list, bucketed by the length of the list of the face's centroid cell ----------------
    // Cell lists have a heavy tail (mean 16, max > 60 entries) and the filter walk below is a
    // lockstep loop: a wave is as slow as its longest list.  Handing out the live faces in
    // buckets of similar list length (longest first) makes the 64 lists of a wave comparable.
    const bool use_grid = s_ctl[1] == 0;
    // diagnostics (nlos_ctx_last_path): this workgroup traces its rays through the in-kernel BVH query
    if (!use_grid && a.retry && tid == 0 && (TILED || pass > 0)) a.retry[wgid] = 0x200;   // (first pass, one workgroup per source: at the end)
#ifdef NLOS_FWD_STAMPS
    if (TILED && tid == 0 && a.dbg && !use_grid && !ident) atomicAdd((unsigned long long*)&a.dbg[20], 1ull);   // entry overflow
    if (TILED && tid == 0 && a.dbg) atomicMax((unsigned long long*)&a.dbg[21], (unsigned long long)s_ctl[2]);
#endif
    auto face_bucket = [&](int j) -> int {
        if (!use_grid) return 0;
        // longest list among the cells under the face's projected bounding box
        const int jg = gid(j);
        const float4 q0 = a.sc.facerec[4 * jg], q1 = a.sc.facerec[4 * jg + 1], q2 = a.sc.facerec[4 * jg + 2];
        const Proj2 q = project_tri(o, mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y), mk(q1.z, q1.w, q2.x));
        const int cx0 = cell_coord(fminf(fminf(q.ax, q.bx), q.cx), g.gx0, g.inv_cw, R);
        const int cx1 = cell_coord(fmaxf(fmaxf(q.ax, q.bx), q.cx), g.gx0, g.inv_cw, R);
        const int cy0 = cell_coord(fminf(fminf(q.ay, q.by), q.cy), g.gy0, g.inv_ch, R);
        const int cy1 = cell_coord(fmaxf(fmaxf(q.ay, q.by), q.cy), g.gy0, g.inv_ch, R);
        uint32_t n = 0;
        for (int yy = cy0; yy <= cy1; ++yy)
            for (int xx = cx0; xx <= cx1; ++xx) {
                const int c = yy * R + xx;
                n = max(n, s_cell[c] - (c > 0 ? s_cell[c - 1] : 0u));
            }
        return (NB - 1) - (int)min(n >> NLOS_NB_SHIFT, (uint32_t)(NB - 1));   // bucket 0 = longest lists
    };
    const bool have_nibbles = fill_buckets && use_grid;      // the fill pass ran and counted the buckets
    for (int b = wave; compact && !have_nibbles && b < nblocks; b += nwaves) {
        if ((s_mask[b] >> lane) & 1ull) atomicAdd(&s_bkt[face_bucket((b << 6) + lane)], 1u);
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int q = 0; q < NB; ++q) { s_bkt[NB + q] = run; run += s_bkt[q]; }
    }
    __syncthreads();
    {
        int it = 0;                                          // block b = wave + it * nwaves holds faces tid + it * NT
        for (int b = wave; compact && b < nblocks; b += nwaves, ++it) {
            if ((s_mask[b] >> lane) & 1ull) {
                const int j = (b << 6) + lane;
                const int bkt = have_nibbles ? (int)(((it < 16 ? nib0 >> (4 * it) : nib1 >> (4 * (it - 16)))) & 15ull) : face_bucket(j);
                g_live[atomicAdd(&s_bkt[NB + bkt], 1u)] = (uint16_t)j;
            }
        }
    }
    __syncthreads();
    FWD_STAMP();   // 4: bucketed live list
    const int n_live = compact ? s_ctl[3] : Fl;
    if (!TILED && visout) {
        // The trace stores a visibility word whole when all its strata sit in one 64-ray item, and ORs the pieces of
        // a word that straddles two items: those words -- at most one per item boundary -- start from zero.
        const uint32_t n_r = (uint32_t)n_live * (uint32_t)a.sp.spt;
        for (uint32_t rb = 64u * (uint32_t)(tid + 1); rb < n_r; rb += 64u * (uint32_t)NT) {
            const uint32_t li = rb / (uint32_t)a.sp.spt;
            const uint32_t sb = rb - li * (uint32_t)a.sp.spt;
            if (sb & 31u) visout[((size_t)l * a.vis_words + (sb >> 5)) * F + gid(compact ? (int)g_live[li] : (int)li)] = 0u;
        }
        __syncthreads();
    }

    // ---- trace + histogram: one RAY per lane ---------------------------------------------------------
    // The rays of a source are the (live face, stratum) pairs r = li * spt + s in the order of the bucketed
    // live list; a wave takes 64 consecutive rays through an LDS ticket.  (Round 1 gave every lane a FACE and
    // looped over its strata: the face's vertices, edges and normal then stay live across the walk and the
    // exact-test rounds of every stratum -- 128 VGPRs with 17 spilled and scratch reloads inside the sample
    // loop.  With one ray per lane the face data is dead once the ray exists.)  Per item the 64 rays go through
    // two wave-synchronous stages:
    //  (1) filter: every lane walks its own cell list in lockstep with LDS-only work (entry index +
    //      quantised projected box) and appends the survivors, as (owner lane, triangle) pairs, to a
    //      wave-private LDS queue (ballot + prefix rank);
    //  (2) exact test: whenever kRound pairs are queued (and at the end) each lane takes its share of the
    //      pairs, pulls the owner's ray through ds_bpermute, gathers the 48-byte record and runs the triangle
    //      test; hits are OR-ed into the wave's occlusion mask.
    // The kernel is VALU-issue bound and cell lists have a heavy tail, so the expensive stage must run on
    // dense lanes and must not wait for the longest list.
    const uint64_t lg = (uint64_t)(a.src.source_offset + (long long)l * a.src.source_stride);
    const int spt = a.sp.spt;
    const float lb = a.sp.lb, ub = a.sp.ub, res = a.sp.res;
    double* grow = a.rows ? a.rows + (size_t)l * nbins : nullptr;
    const uint32_t n_rays = (uint32_t)n_live * (uint32_t)spt;       // <= num_samples + F
#ifdef NLOS_DIAG_NO_TRACE          // diagnostic builds only (tools/ab_pmc.sh): cost of the build phases alone
    const int n_items = 0;
#else
    const int n_items = (int)((n_rays + 63u) >> 6);
#endif
    const uint64_t kbase0 = lg * (uint64_t)F;
    const uint64_t zbase = sample_zbase(a.sp.seed, kbase0 * (uint64_t)spt);      // sample_st_c(): key = kbase0 spt + (fid spt + s)
    const bool lean_b = NCM == 2 ? source_frame(a.sc.nodes, ob).ok : true;        // pairs: the sensor's frame (the laser's is frame_ok)
    const float rres = rcp_refined(res);                                          // div_by(x, res, rres) == x / res (launcher: res within the lean range)
    const double inv_spt = 1.0 / (double)spt;
#ifdef NLOS_FWD_STAMPS
    unsigned long long c_rays = 0, c_pairs = 0, c_iters = 0, c_mt = 0, c_mtw = 0, c_hit = 0, c_occ = 0, c_tested = 0;   // diagnostic build only
    unsigned long long c_items = 0, c_dfail = 0, c_push = 0;   // round 6: items, walked entries that fail the depth term, non-empty push slots
    // round 6: of the pairs that reach the exact test, how many would survive a filter that knew the candidate's true projected
    // coverage of the ray's sub-cell at 3 / 4 / 6 / 8 sub-cells per cell side, and the ideal one (the slope point itself, margin kSlopeAbs)
    unsigned long long c_cov[5] = {0ull, 0ull, 0ull, 0ull, 0ull};
    long long tg = 0, ts = 0, tx = 0, th = 0, tmark = 0;
#define TMARK() (tmark = clock64())
#define TACC(v) do { long long now_ = clock64(); v += now_ - tmark; tmark = now_; } while (0)
#else
#define TMARK() do { } while (0)
#define TACC(v) do { } while (0)
#endif
    // what this source did (nlos_ctx_last_path, bench.py: rays_traced_per_step / accepted_per_step): rays that reached the
    // occlusion query and samples that were binned, counted per wave on the scalar unit
    int n_traced = 0, n_accepted = 0;
    uint32_t* wq = s_queue + wave * (kQueueCap + 1);     // this wave's pair queue (aliases the build-phase tables) + dump slot
    uint32_t* wocc = s_queue + nwaves * (kQueueCap + 1) + wave * 2;  // this wave's 64-bit occlusion mask

    // GRID (workgroup-uniform): occlusion through the cell lists; otherwise (scene not strictly in front of the
    // wall point, or lists that fit nowhere) every ray runs the stackless BVH query.  Two instances of the loop,
    // so that the traversal's registers and masks are not live in the grid loop.  With a valid frame every
    // vertex lies in front of the wall point, hence dir.z > 0 for every ray of the grid loop.
    auto trace = [&](auto grid_c) {
        constexpr bool GRID = decltype(grid_c)::value;
        // The accepted-sample words of an item are written one item LATER, right after that item's face records have
        // been requested: a store counts in vmcnt like the loads, in order, so the wait for the next loads after a
        // store also waits for the store's acknowledgement from the L2 -- with the store at the end of the item every
        // wave stalled for it before it could start the next item (measured without the stores: 12 % of the kernel on
        // the bunny, 34 % on the 1 055-face mannequin, for 1.7 % of the instructions).  Behind the loads the two
        // latencies overlap: forward 1.84 -> 1.75 ms (bunny), 1.92 -> 1.60 ms (mannequin).  (Issued after the records
        // have been consumed, or after the sample arithmetic, it is no faster: measured.)  pend_at: word offset within the source's rows, bit 31 = OR the piece in (a word shared
        // with another item or tile) instead of storing it; ~0 = nothing pending.
        uint32_t pend_at = ~0u, pend_bits = 0u;
        uint32_t* const vrow = visout ? visout + (size_t)l * a.vis_words * F : nullptr;
        // item-mask layout: the ballot of an item and its index are wave-uniform (scalar registers): one 8-byte store by
        // lane 0, deferred like the words (behind the next item's loads)
        int pend_item = -1;
        unsigned long long pend_mask = 0ull;
        auto flush_item = [&]() {
#ifndef NLOS_DIAG_NO_ITEMSTORE     // diagnostic builds only: what the one store per item still costs
            if (pend_item >= 0 && lane == 0) vitems[1 + pend_item] = pend_mask;
#endif
            pend_item = -1;
        };
        auto flush_pending = [&]() {
            if (pend_at != ~0u) {
                uint32_t* wp = vrow + (pend_at & 0x7FFFFFFFu);
                if (pend_at & 0x80000000u) atomicOr(wp, pend_bits);
                else *wp = pend_bits;
            }
            pend_at = ~0u;
        };
#ifndef NLOS_SLOT_MAGIC
#define NLOS_SLOT_MAGIC 1
#endif
        // ray -> (live-list slot, stratum): one v_mul_hi with M = ceil(2^32 / spt) where that is exact for every ray of
        // this workgroup (r (M spt - 2^32) < 2^32: the error term then stays below 1 / spt), the generic division otherwise
        const uint32_t spt_magic = spt > 1 ? 0xFFFFFFFFu / (uint32_t)spt + 1u : 0u;
        const bool magic_ok = NLOS_SLOT_MAGIC && spt > 1 && (uint64_t)n_rays * (uint64_t)(uint32_t)(spt_magic * (uint32_t)spt) < (1ull << 32);
        // (Claiming the next item and requesting its live-list entries one item ahead, so that ticket, entry and face
        // records are not three dependent round trips in front of every item's arithmetic, changes nothing: the kernel
        // waits for its VALU, not for these.  Measured, profiles/r03_ab_lookahead.log.)
#ifndef NLOS_LOOKAHEAD
#define NLOS_LOOKAHEAD 0
#endif
        // NLOS_LOOKAHEAD (round 6, re-measured on this kernel): the ticket and the live-list entry of the NEXT item are requested
        // behind this item's face-record loads, so that ticket -> live list -> face record are two dependent round trips in front
        // of an item's arithmetic instead of three (one scalar and one vector register live across the item).
        auto slot_of = [&](uint32_t r, bool has) -> uint32_t { return has ? (magic_ok ? __umulhi(r, spt_magic) : r / (uint32_t)spt) : 0u; };
        int b_next = NLOS_LOOKAHEAD ? wave_ticket(&s_ctl[0]) : 0;
        int j_next = 0;
        if (NLOS_LOOKAHEAD && b_next < n_items) {
            const uint32_t r0 = ((uint32_t)b_next << 6) + (uint32_t)lane;
            const uint32_t li0 = slot_of(r0, r0 < n_rays);
            j_next = compact ? (int)g_live[li0] : (int)li0;
        }
        for (;;) {
            const int b = NLOS_LOOKAHEAD ? b_next : wave_ticket(&s_ctl[0]);
            if (b >= n_items) break;
            TMARK();
            const uint32_t r = ((uint32_t)b << 6) + (uint32_t)lane;
            bool has_ray = r < n_rays;
            const uint32_t li = slot_of(r, has_ray);
            const int s = has_ray ? (int)(r - li * (uint32_t)spt) : 0;
            const int j = NLOS_LOOKAHEAD ? j_next : (compact ? (int)g_live[li] : (int)li);   // index within this workgroup's face set
            const int jg = gid(j);                                          // sorted-face index
            Face f;
            Tri tr;
            load_face_tri<FEAT>(a.sc, jg, f, tr);
            if (NLOS_LOOKAHEAD) {
                b_next = wave_ticket(&s_ctl[0]);
                if (b_next < n_items) {
                    const uint32_t r1 = ((uint32_t)b_next << 6) + (uint32_t)lane;
                    const uint32_t li1 = slot_of(r1, r1 < n_rays);
                    j_next = compact ? (int)g_live[li1] : (int)li1;
                }
            }
            diag_pad<NLOS_DIAG_PAD_GEN>();
#ifdef NLOS_FWD_STAMPS
            if (lane == 0) c_items += 1;
#endif
            if (visout) flush_pending();                                    // the previous item's words, behind this item's loads
            if (vitems) flush_item();
            if (!compact && has_ray) has_ray = !face_dark(f);               // (the block masks are gone: the queue reuses their LDS)
            if (TILED && !compact && has_ray && frame_ok) {
                // overflowed subset: every face is visited, most of them lie outside this tile
                const Proj2 q = project_tri(o, f.p0, f.p1, f.p2);
                const float cwm = __builtin_amdgcn_rcpf(g.inv_cw) * (1.0f + kEdgeFrac), chm = __builtin_amdgcn_rcpf(g.inv_ch) * (1.0f + kEdgeFrac);
                has_ray = fmaxf(fmaxf(q.ax, q.bx), q.cx) >= g.gx0 - kEdgeFrac * cwm - kSlopeAbs && fminf(fminf(q.ax, q.bx), q.cx) <= g.gx0 + (float)R * cwm + kSlopeAbs &&
                          fmaxf(fmaxf(q.ay, q.by), q.cy) >= g.gy0 - kEdgeFrac * chm - kSlopeAbs && fminf(fminf(q.ay, q.by), q.cy) <= g.gy0 + (float)R * chm + kSlopeAbs;
            }
            const uint64_t key = (kbase0 + (uint64_t)f.fid) * (uint64_t)spt + (uint64_t)s;
            V3 dir = mk(0.0f, 0.0f, 1.0f);
            float t_self = 0.0f, val = 0.0f;
            int bin = -1;
            bool ok = has_ray;
            if (NCM == 0) {
                Geo gg;
                // GRID: the lean forms of sqrt / reciprocal / division (nlos_device.h: the same bits on the range frame_ok
                // and the launcher guarantee), the draw keyed per source on the scalar unit
                constexpr bool LEAN = GRID;
                if (ok) {
                    float S, T;
                    if (LEAN) sample_st_c(zbase, (uint32_t)f.fid * (uint32_t)spt + (uint32_t)s, S, T);
                    else sample_st(a.sp.seed, key, S, T);
                    ok = sample_geo_st<FEAT, LEAN>(f, tr, o, S, T, lb, ub, a.sc.vertex_normal, a.sc.albedo, gg, t_self);
                }
                if (ok) {
                    float ff = form_factor<LEAN>(-dot(gg.n, gg.dir) * dot(on, gg.dir), gg.h);
                    if (a.sp.clamp) {
                        ff = emax0(ff);
                        ok = ff > 0.0f;
                    }
                    val = f.area * gg.alb * ff * ff;
                    if (FEAT & FEAT_GGX) val = val * ggx_eval(a.sp.ggx_alpha, dot(gg.n, -gg.dir));
                    bin = (int)floorf(LEAN ? div_by(2.0f * gg.h - lb, res, rres) : (2.0f * gg.h - lb) / res);
                    dir = gg.dir;
                    if (geo_l && ok) {
                        // What pass 2 needs of this sample: the sampled direction, the hit's barycentrics and h (24 B: float4 + float2),
                        // written HERE, before the trace (the barycentrics are dead afterwards; an occluded ray's entry is never
                        // read).  Record of (live-list entry li, stratum s) at [s][li]: pass 2 -- one lane per entry, one stratum
                        // per trip -- reads 64 consecutive records per load and every cache line once.  Streaming stores here and
                        // streaming loads there: ~1 GB written once and read once per launch stays out of the way of the scene
                        // records the L2 serves all the time.  Measured (profiles/r04_ab_geo_cache.log, _nt.log, _layout.log):
                        // forward 1.341 -> 1.366 ms, pass 2 0.548 -> 0.474 ms.  What was tried on the way: a 12-byte record (h, v, w,
                        // direction rebuilt from the hit point) is faster (0.447 ms) but moves the gradient by up to 7e-5 on grazing
                        // rays; records at [li][s] are fetched spt times by the streaming loads (0.60 ms); cached loads or stores
                        // cost 2-3 % of the step; the store deferred behind the next item's loads costs three registers across
                        // the trace (+0.03 ms).
                        typedef float f4_t __attribute__((ext_vector_type(4)));
                        const size_t at = (size_t)s * (size_t)F + (size_t)li;
                        f4_t* gp = reinterpret_cast<f4_t*>(geo_l) + at;
                        const f4_t rec = {gg.dir.x, gg.dir.y, gg.dir.z, gg.v};
                        typedef float f2_t __attribute__((ext_vector_type(2)));
                        const f2_t rec2 = {gg.w, gg.h};
#ifdef NLOS_GEO_CACHED_STORES          // diagnostic builds only: write-back stores (many strata per face: the pieces of a line merge in the L2)
                        *gp = rec;
                        *(reinterpret_cast<f2_t*>(geo_w) + at) = rec2;
#else
                        __builtin_nontemporal_store(rec, gp);
                        __builtin_nontemporal_store(rec2, reinterpret_cast<f2_t*>(geo_w) + at);
#endif
                    }
                }
            } else if (NCM == 3) {
                // record pass of the product (row N as L x S): this wall point's leg of every pair it takes part in --
                // the expressions of sample_geo_nc() for one end point.  Path length of the leg and its clamped form factor
                // travel through the trace in `bin` / `val` and are stored for the samples the wall point sees.
                // (GRID: the lean forms, as in the confocal branch -- this wall point's frame guarantees their range)
                constexpr bool LEAN = GRID;
                if (ok) {
                    float S, T;
                    sample_st(a.sp.seed, key, S, T);
                    const float sq = LEAN ? sqrt_cr0(T) : sqrtf(T);
                    const V3 p = bary(1 - sq, f.p0, (1 - S) * sq, f.p1, S * sq, f.p2);
                    const V3 d = p - o;
                    dir = d * (LEAN ? rcp_cr(sqrt_cr(dot(d, d))) : 1.0f / sqrtf(dot(d, d)));
                    float hu, hv;
                    ok = tri_test<LEAN>(tr, o, dir, t_self, hu, hv);
                    if (ok) {
                        const float hw = 1.0f - hu - hv;
                        const V3 q = bary(hw, f.p0, hu, f.p1, hv, f.p2);
                        const V3 e = q - o;
                        const float dist = LEAN ? sqrt_cr(dot(e, e)) : sqrtf(dot(e, e));
                        // normal and albedo at THIS leg's hit (sample_geo_nc()'s expressions: they are the pair's when this
                        // wall point is its laser)
                        V3 nn = f.fn;
                        if (FEAT & FEAT_VN)
                            nn = bary(hw, ld3(a.sc.vertex_normal + 3 * (size_t)f.i0), hu, ld3(a.sc.vertex_normal + 3 * (size_t)f.i1), hv,
                                      ld3(a.sc.vertex_normal + 3 * (size_t)f.i2));
                        val = emax0(form_factor<LEAN>(-dot(nn, dir) * dot(on, dir), dist));
                        bin = __float_as_int(dist);
                        if (FEAT & (FEAT_VN | FEAT_ALB)) {
                            // extended record: written HERE (normal and albedo are dead after this block; an occluded ray's record
                            // is invalidated by rec_d staying 0); the sensor role needs the leg even where this form factor is 0
                            const size_t at = (size_t)l * ((size_t)F * (size_t)spt) + (size_t)jg * (size_t)spt + (size_t)s;
                            float* ex = a.rec_ext + at;
                            const size_t es = a.rec_ext_stride;
                            ex[0] = dir.x; ex[es] = dir.y; ex[2 * es] = dir.z;
                            ex[3 * es] = nn.x; ex[4 * es] = nn.y; ex[5 * es] = nn.z;
                            float al = 1.0f;
                            if (FEAT & FEAT_ALB) al = hw * a.sc.albedo[f.i0] + hu * a.sc.albedo[f.i1] + hv * a.sc.albedo[f.i2];
                            ex[6 * es] = al;
                        } else {
                            ok = val > 0.0f;
                        }
                    }
                }
            } else if (NCM == 1) {
                // sensor leg only: is the stratified point the closest hit seen from the sensor?
                constexpr bool LEAN = GRID;
                if (ok) {
                    float S, T;
                    sample_st(a.sp.seed, key, S, T);
                    const float sq = LEAN ? sqrt_cr0(T) : sqrtf(T);
                    const V3 p = bary(1 - sq, f.p0, (1 - S) * sq, f.p1, S * sq, f.p2);
                    const V3 d = p - o;
                    dir = d * (LEAN ? rcp_cr(sqrt_cr(dot(d, d))) : 1.0f / sqrtf(dot(d, d)));
                    float hu, hv;
                    ok = tri_test<LEAN>(tr, o, dir, t_self, hu, hv);
                    // a leg whose form factor is exactly zero is rejected by the laser pass anyway: skip its ray
                    if (ok && !(FEAT & FEAT_VN)) ok = (-dot(f.fn, dir) * dot(on, dir)) > 0.0f;
                }
            } else {
                GeoNC gc;
                float t_b;
                // (the lean forms where the laser's frame -- GRID -- and the sensor's -- lean_b, evaluated once per workgroup --
                // both guarantee the range)
                const bool lean_nc = GRID && lean_b;
                if (ok)
                    ok = sample_geo_nc_rt<FEAT>(f, tr, o, ob, a.sp.seed, key, lean_nc, lb, ub, a.sc.vertex_normal, a.sc.albedo, gc, t_self, t_b);
                if (ok) {
                    const uint32_t word_b = a.vis2[((size_t)l * a.vis_words + (size_t)(s >> 5)) * F + jg];
                    const float numa = -dot(gc.n, gc.dirA) * dot(on, gc.dirA), numb = -dot(gc.n, gc.dirB) * dot(onb, gc.dirB);
                    const float ffa = emax0(lean_nc ? form_factor<true>(numa, gc.d1) : form_factor<false>(numa, gc.d1));
                    const float ffb = emax0(lean_nc ? form_factor<true>(numb, gc.d2) : form_factor<false>(numb, gc.d2));
                    ok = ffa > 0.0f && ffb > 0.0f && ((word_b >> (s & 31)) & 1u);
                    val = f.area * gc.alb * ffa * ffb;
                    if (FEAT & FEAT_GGX) val = val * ggx_pair<false>(a.sp.ggx_alpha, gc.n, -gc.dirA, -gc.dirB).brdf;
                    bin = (int)floorf(lean_nc ? div_by((gc.d1 + gc.d2) - lb, res, rres) : ((gc.d1 + gc.d2) - lb) / res);
                    dir = gc.dirA;
                }
            }
            const int fid = f.fid;
            if (TILED && ok && frame_ok) {
                // the tile that owns this sample: from the source's global frame, identical in every workgroup
                const float izo = __builtin_amdgcn_rcpf(dir.z);
                const int ti = min(max((int)floorf((dir.x * izo - Gx0) * inv_tw), 0), a.tiles_x - 1);
                const int tj = min(max((int)floorf((dir.y * izo - Gy0) * inv_th), 0), a.tiles_y - 1);
                ok = dir.z > 0.0f ? (ti == tile_x && tj == tile_y) : tile == 0;
            }
            if (!ok) dir = mk(0.0f, 0.0f, 1.0f);
            if (vitems) n_traced += __popcll(__ballot(ok));
            if (!GRID && ok)
                ok = !occluded(a.sc.nodes, a.sc.n_nodes, a.sc.tris, a.sc.face_id, o, dir, t_self, jg, fid);

            if (GRID) {
                // ---- stage 1 + 2 (wave-synchronous; every lane takes part) ----
                const bool grid_ray = ok;
                uint32_t e = 0, e1 = 0, rmask = 0, rlim = 0;
                if (grid_ray) {
                    const float iz = __builtin_amdgcn_rcpf(dir.z);   // lookups only: 1-ulp rcp is fine
                    const float ux = (dir.x * iz - g.gx0) * g.inv_cw, uy = (dir.y * iz - g.gy0) * g.inv_ch;
                    const int cxx = min(max((int)floorf(ux), 0), R - 1), cyy = min(max((int)floorf(uy), 0), R - 1);
                    const int sx = min(max((int)floorf((ux - (float)cxx) * (float)kSub), 0), kSub - 1);
                    const int sy = min(max((int)floorf((uy - (float)cyy) * (float)kSub), 0), kSub - 1);
                    rmask = (1u << (IB + sx)) | (1u << (IB + kSub + sy));
                    // depth level of the own-face hit, rounded up: anything quantised deeper cannot occlude
                    const float zs = t_self * dir.z;
                    const uint32_t rq = (uint32_t)min(max((int)floorf((zs * kDepthRel - g.z0) * g.inv_qz) + 1, 0), g.zmax);
                    rlim = (rq << (IB + 2 * kSub)) | ((1u << (IB + 2 * kSub)) - 1u);
                    const int c = cyy * R + cxx;
                    e1 = s_cell[c];
                    e = c > 0 ? s_cell[c - 1] : 0u;
#ifdef NLOS_DIAG_NO_WALK           // diagnostic builds only
                    e = e1;
#endif
                }
                if (lane < 2) wocc[lane] = 0u;
                TACC(tg);
                int qn = 0;                                        // wave-uniform
                auto exact_round = [&](int n) {
#ifdef NLOS_DIAG_NO_EXACT          // diagnostic builds only: results are wrong, the counters show what the rounds cost
                    return;
#endif
#ifdef NLOS_FWD_STAMPS
                    if (lane == 0) c_mtw += 1;
#endif
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int h = 0; h < kRound / 64; ++h) {
                        const int qi = lane + 64 * h;
                        const uint32_t pr = wq[qi < n ? qi : 0];
                        const int owner = (int)(pr >> 16), k = (int)(pr & 0xFFFFu);
                        const V3 od = mk(__shfl(dir.x, owner), __shfl(dir.y, owner), __shfl(dir.z, owner));
                        const float ot = __shfl(t_self, owner);
                        const int ofid = __shfl(fid, owner);
                        if (qi < n) {
                            const int kg = gid(k);
                            const Tri tk = load_tri48(a.sc.tris, kg);
                            const bool hit_k = tri_occludes(tk, o, od, ot, ofid, a.sc.face_id, kg, a.sc.tris);
#ifdef NLOS_FWD_STAMPS
                            c_tested += 1;
                            if (hit_k) c_hit += 1;
                            {
                                const V3 k0 = tk.p0, k1 = tk.p0 - tk.e1, k2 = tk.p0 + tk.e2;
                                const Proj2 qk = project_tri(o, k0, k1, k2);
                                const float izd = 1.0f / od.z;
                                const float mx = od.x * izd, my = od.y * izd;
                                const float area = (qk.bx - qk.ax) * (qk.cy - qk.ay) - (qk.by - qk.ay) * (qk.cx - qk.ax);
                                const float sgn = area < 0.0f ? -1.0f : 1.0f;
                                const float A[3] = {-(qk.by - qk.ay) * sgn, -(qk.cy - qk.by) * sgn, -(qk.ay - qk.cy) * sgn};
                                const float B[3] = {(qk.bx - qk.ax) * sgn, (qk.cx - qk.bx) * sgn, (qk.ax - qk.cx) * sgn};
                                const float C[3] = {-(A[0] * qk.ax + B[0] * qk.ay), -(A[1] * qk.bx + B[1] * qk.by), -(A[2] * qk.cx + B[2] * qk.cy)};
                                const float bx0 = fminf(fminf(qk.ax, qk.bx), qk.cx), bx1 = fmaxf(fmaxf(qk.ax, qk.bx), qk.cx);
                                const float by0 = fminf(fminf(qk.ay, qk.by), qk.cy), by1 = fmaxf(fmaxf(qk.ay, qk.by), qk.cy);
                                const float cwf = 1.0f / g.inv_cw, chf = 1.0f / g.inv_ch;
                                const int subs[5] = {3, 4, 6, 8, 0};
                                for (int si = 0; si < 5; ++si) {
                                    float x0, x1, y0, y1;
                                    if (subs[si] == 0) { x0 = x1 = mx; y0 = y1 = my; }
                                    else {
                                        const float S = (float)subs[si];
                                        const float ux = (mx - g.gx0) * g.inv_cw, uy = (my - g.gy0) * g.inv_ch;
                                        x0 = g.gx0 + floorf(ux * S) / S * cwf; x1 = x0 + cwf / S;
                                        y0 = g.gy0 + floorf(uy * S) / S * chf; y1 = y0 + chf / S;
                                    }
                                    const float mg = kSlopeAbs + 0.02f * cwf / 4.0f;       // today's mask margin, in slope units
                                    bool ov = bx1 + mg >= x0 && bx0 - mg <= x1 && by1 + mg >= y0 && by0 - mg <= y1;
                                    for (int ei = 0; ei < 3 && ov; ++ei) {
                                        const float ex = A[ei] > 0.0f ? x1 : x0, ey = B[ei] > 0.0f ? y1 : y0;
                                        ov = A[ei] * ex + B[ei] * ey + C[ei] + (fabsf(A[ei]) + fabsf(B[ei])) * mg >= 0.0f;
                                    }
                                    if (ov || fabsf(area) < 1e-4f * cwf * chf) c_cov[si] += 1;
                                }
                            }
#endif
                            if (hit_k)
                                atomicOr(&wocc[owner >> 5], 1u << (owner & 31));
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                };
                // lockstep filter walk (LDS only): entry index + packed box/depth word.  (A fully
                // flattened walk -- pairs spread evenly over the lanes with a prefix-sum owner search --
                // halves the iterations but its dependent ds_bpermute chain makes it slower; measured.)
#ifdef NLOS_FWD_STAMPS
                if (grid_ray) c_rays += 1;
#endif
                constexpr uint32_t imask = (1u << IB) - 1u;
                // `m` is the wave's 64-bit mask of the lanes that append (an SGPR pair straight out of the compares: the
                // ballot of a boolean that was AND-ed together costs a v_cndmask + v_cmp per slot on this compiler).
                auto push = [&](unsigned long long m, uint32_t w) {
                    if (m) {
                        // every lane stores -- the ones that append nothing into the dump slot: a select on the mask
                        // instead of a divergent region
#ifdef NLOS_DIAG_MBCNT_ADD         // diagnostic builds only: the separate v_add_u32 this used to cost
                        const uint32_t at = (uint32_t)qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
#else
                        const uint32_t at = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, (uint32_t)qn));   // v_mbcnt adds its last operand: qn rides along
#endif
                        uint32_t slot;
                        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(slot) : "v"((uint32_t)kQueueCap), "v"(at), "s"(m));
                        wq[slot] = ((uint32_t)lane << 16) | (w & imask);
                        qn += __popcll(m);
                        if (qn >= kRound) {
                            TACC(ts);
                            exact_round(kRound);
                            TACC(tx);
                            qn -= kRound;
                            const uint32_t mv = wq[kRound + (lane < qn ? lane : 0)];       // qn < 64 left over
                            __builtin_amdgcn_wave_barrier();
                            if (lane < qn) wq[lane] = mv;
#ifndef NLOS_DIAG_NO_EARLY_OUT
                            // a ray the round has just found occluded needs no more candidates: it stops walking its list
                            // (rlim = 0 fails the depth term of whatever it still reads) -- fewer pairs, and shorter trips
                            // where it held the longest list of the wave.  The decision "occluded" is an OR over the
                            // candidates, so skipping the rest changes nothing.  (Round 3: forward 1.649 -> 1.600 ms.)
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                            if ((wocc[lane >> 5] >> (lane & 31)) & 1u) { rlim = 0u; e = e1; }
#endif
                        }
                    }
                };
                // The filter on one word: with x = w ^ j (the index field of x is zero for the ray's own face)
                //   x <= rlim                       <=>  w <= rlim   (rlim's low bits are all ones: only the depth field decides)
                //   (x & (rmask | imask)) > rmask   <=>  both mask bits of the ray are set AND the index field is not zero
                // (the two bits of rmask lie above the index field, so a word that misses one of them stays below rmask
                // whatever its index) -- four VALU operations per entry instead of six.
                const uint32_t jx = (uint32_t)j, m2 = rmask | imask;
                constexpr int kULT = 36, kULE = 37, kUGT = 34;       // llvm.amdgcn.icmp predicates
                // kScan entries per trip: the lockstep walk pays its loop overhead (any(), branch, counters) once per
                // trip; lists average 24 entries, so wider trips waste more slots at the end (2: 2.49 ms, 4: 2.43 ms)
                while (__any(e < e1)) {
                    diag_pad<NLOS_DIAG_PAD_WALK>();
                    // every lane reads its next kScan words (finished lanes re-read the slots behind their list and
                    // fail the range term): no divergent region, the compare masks are combined on the scalar unit
                    const uint32_t rem = e < e1 ? e1 - e : 0u;
                    unsigned long long p[kScan];
                    uint32_t w[kScan];
#pragma unroll
                    for (int q = 0; q < kScan; ++q) {
                        w[q] = s_ent[e + q];
                        const uint32_t x = w[q] ^ jx;
                        p[q] = __builtin_amdgcn_uicmp((uint32_t)q, rem, kULT) & __builtin_amdgcn_uicmp(x, rlim, kULE) &
                               __builtin_amdgcn_uicmp(x & m2, rmask, kUGT);
                    }
#ifdef NLOS_FWD_STAMPS
                    if (grid_ray) c_pairs += min((uint32_t)kScan, rem);
                    if (lane == 0) c_iters += 1;
#pragma unroll
                    for (int q = 0; q < kScan; ++q) if ((p[q] >> lane) & 1ull) c_mt += 1;
#pragma unroll
                    for (int q = 0; q < kScan; ++q) {
                        if (grid_ray && (uint32_t)q < rem && (w[q] ^ jx) > rlim) c_dfail += 1;
                        if (lane == 0 && p[q]) c_push += 1;
                    }
#endif
                    e += min((uint32_t)kScan, rem);
#pragma unroll
                    for (int q = 0; q < kScan; ++q) push(p[q], w[q]);
                }
                TACC(ts);
                if (qn > 0) exact_round(qn);
                TACC(tx);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (grid_ray) ok = ((wocc[lane >> 5] >> (lane & 31)) & 1u) == 0u;
#ifdef NLOS_FWD_STAMPS
                if (grid_ray && !ok) c_occ += 1;
#endif
                __builtin_amdgcn_wave_barrier();
            } else {
                TACC(tg);
            }

            if (NCM == 3) {
                if (ok) {
                    const size_t at = (size_t)l * ((size_t)F * (size_t)spt) + (size_t)jg * (size_t)spt + (size_t)s;
                    a.rec_d[at] = __int_as_float(bin);
                    a.rec_ff[at] = val;
                }
            } else if (ok && NCM != 1) {
                // (double)val / spt as the reference bins it, up to the rounding of 1 / spt (below the order-of-
                // summation noise of the fp64 rows)
                const double cc = (double)val * inv_spt;
                if (a.mode_intensity) unsafeAtomicAdd(&a.intensity[fid], cc);
                else if (bin >= 0 && bin < nbins) {
                    if (rows_in_lds) lds_add_f64(&s_row[bin], cc);
                    else unsafeAtomicAdd(&grow[bin], cc);
                }
            }
#ifdef NLOS_DIAG_NO_VIS            // diagnostic builds only (tools/ab_traffic.sh): HBM traffic without the accepted-sample words
            if (false) {
#else
            if (visout) {
#endif
                // accepted-sample bits: the strata of a (face, word) that this wave holds sit in consecutive lanes;
                // the first of them writes the piece.  A word that lies completely inside the wave is stored (its
                // only writer); a face that straddles two items -- or tiles -- ORs its pieces into the zeroed word.
                const unsigned long long acc = __ballot(ok);
                if (has_ray && (lane == 0 || (s & 31) == 0)) {
                    const int in_word = min(spt - s, 32 - (s & 31));                  // strata left in this word
                    const int cnt = min(in_word, 64 - lane);                          // ... of which this wave holds
                    const uint32_t bits = (uint32_t)((acc >> lane) & ((1ull << cnt) - 1ull));
                    const uint32_t at = (uint32_t)(s >> 5) * (uint32_t)F + (uint32_t)jg;
                    if (!TILED && (s & 31) == 0 && cnt == in_word) { pend_at = at; pend_bits = bits; }
                    else if (bits) { pend_at = at | 0x80000000u; pend_bits = bits << (s & 31); }
                }
            }
            if (vitems) { pend_item = b; pend_mask = __ballot(ok); n_accepted += __popcll(pend_mask); }
            TACC(th);
        }
        if (visout) flush_pending();
        if (vitems) flush_item();
    };
    if (use_grid) trace(std::true_type{});
    else trace(std::false_type{});
#ifdef NLOS_FWD_STAMPS
    if (a.dbg) {   // diagnostic build only: work counters -> a.dbg[8..12]
        atomicAdd((unsigned long long*)&a.dbg[8], c_rays);
        atomicAdd((unsigned long long*)&a.dbg[9], c_pairs);
        atomicAdd((unsigned long long*)&a.dbg[10], c_iters);
        atomicAdd((unsigned long long*)&a.dbg[11], c_mt);
        atomicAdd((unsigned long long*)&a.dbg[12], c_mtw);
        atomicAdd((unsigned long long*)&a.dbg[18], c_hit);
        atomicAdd((unsigned long long*)&a.dbg[19], c_occ);
        atomicAdd((unsigned long long*)&a.dbg[6], c_tested);
        atomicAdd((unsigned long long*)&a.dbg[7], c_push);
        atomicAdd((unsigned long long*)&a.dbg[26], c_items);
        atomicAdd((unsigned long long*)&a.dbg[27], c_cfw);
        atomicAdd((unsigned long long*)&a.dbg[28], c_dfail);
        atomicAdd((unsigned long long*)&a.dbg[29], c_cnt[0]);
        atomicAdd((unsigned long long*)&a.dbg[30], c_cnt[1]);
        atomicAdd((unsigned long long*)&a.dbg[31], c_fill[1]);
        atomicAdd((unsigned long long*)&a.dbg[32], c_fill[0]);
        atomicAdd((unsigned long long*)&a.dbg[33], c_ffw);
        for (int si = 0; si < 5; ++si) atomicAdd((unsigned long long*)&a.dbg[34 + si], c_cov[si]);
        if (tid == 0) atomicAdd((unsigned long long*)&a.dbg[13], (unsigned long long)s_ctl[2]);
        if (lane == 0) {
            atomicAdd((unsigned long long*)&a.dbg[14], (unsigned long long)tg);
            atomicAdd((unsigned long long*)&a.dbg[15], (unsigned long long)ts);
            atomicAdd((unsigned long long*)&a.dbg[16], (unsigned long long)tx);
            atomicAdd((unsigned long long*)&a.dbg[17], (unsigned long long)th);
        }
    }
#endif
    FWD_STAMP();   // 5: sample + trace + histogram
#ifdef NLOS_FWD_STAMPS
    if (tid == 0 && a.dbg) {   // the shader clock this kernel really ran at: sum of s_memtime ticks / sum of s_memrealtime ticks (100 MHz)
        atomicAdd((unsigned long long*)&a.dbg[24], (unsigned long long)(clock64() - c_start));
        atomicAdd((unsigned long long*)&a.dbg[25], (unsigned long long)(wall_clock64() - w_start));
        // this workgroup's lifetime in shader-clock cycles, into the first bytes of its own (now dead) coverage scratch:
        // the per-source spread behind strong scaling (one resident round of workgroups lasts as long as its heaviest source)
        if (!TILED && (F & 1) == 0) {
            const unsigned long long dt = (unsigned long long)(clock64() - c_start);
            uint32_t* cw = reinterpret_cast<uint32_t*>(g_cov);
            cw[0] = (uint32_t)dt; cw[1] = (uint32_t)(dt >> 32);
        }
    }
#endif
    // diagnostics (nlos_ctx_debug_read what = 2): the coarsened resolution this workgroup ended up with; the
    // big-LDS launch only looks for the value 1
    // Every first-pass workgroup of the one-workgroup-per-source launch writes its code on its way out (0 = plain
    // grid), so the array needs no memset before the launch.
    if (vitems) {
        // header word of the source's item masks: [15:0] live faces (pass 2 walks the live list; g_live stays valid),
        // [39:16] rays traced, [63:40] samples accepted (F <= 8191, F spt < 2^24 on this path)
        if (lane == 0) { atomicAdd(&s_ctl[5], n_traced); atomicAdd(&s_ctl[6], n_accepted); }
        __syncthreads();
        if (tid == 0) vitems[0] = (unsigned long long)n_live | ((unsigned long long)(uint32_t)s_ctl[5] << 16) | ((unsigned long long)(uint32_t)s_ctl[6] << 40);
    }
    if (a.retry && tid == 0) {
        if (!TILED && pass == 0) a.retry[wgid] = COARSE ? 0x100 + R : (use_grid ? 0 : 0x200);
        else if (COARSE) a.retry[wgid] = 0x100 + R;
    }
    if (rows_in_lds && grow) {
        __syncthreads();
        if (!TILED) {
            for (int i = tid; i < nbins; i += NT) grow[i] = s_row[i];
        } else {
            for (int i = tid; i < nbins; i += NT)
                if (s_row[i] != 0.0) unsafeAtomicAdd(&grow[i], s_row[i]);     // one partial row per tile
        }
    }
#ifdef NLOS_FWD_STAMPS_LIGHT
    __syncthreads();
    FWD_STAMP();   // 6: waiting for the slowest wave of the trace + header + row flush
#endif
    return false;
}

// PASS >= 0 fixes the pass at compile time (the single-workgroup grid: the usually idle big-LDS launch then
// shows up under its own kernel name in profiles); PASS = -1 takes it from the argument.
template <int FEAT, int NCM = 0, bool TILED = false, int PASS = -1>
__global__ __launch_bounds__(kGridNT, kGridNT / 128) void k_forward_grid(ForwardArgs a, int rows_in_lds, int R, int cap, int pass_arg = 0,
                                                         int last_pass = 1) {
    __shared__ uint32_t s_scan[kGridNT];
    __shared__ uint32_t s_bkt[32];                   // live faces per list-length bucket, then the write cursors
    const int pass = PASS >= 0 ? PASS : pass_arg;
    if (pass >= 1 && a.retry[(!TILED && a.perm) ? a.perm[blockIdx.x] : blockIdx.x] != pass) return;  // second launch: only the workgroups flagged for it
    if (!TILED && NCM == 0 && pass == 0) {
        // the chores of the residual launch (ForwardArgs::zero / pathlengths): done here when pass 2 forms the residual itself
        if (a.zero)
            for (size_t i = (size_t)blockIdx.x * kGridNT + threadIdx.x; i < a.zero_n; i += (size_t)gridDim.x * kGridNT) a.zero[i] = 0.0;
        if (a.pathlengths && blockIdx.x == 0)
            for (int i = threadIdx.x; i < a.path_T; i += kGridNT) a.pathlengths[i] = (double)(a.path_lb + i * a.path_res);
    }
    // (Round 6, built, measured, removed: PERSISTENT workgroups -- 2 per CU drawing sources from a self-resetting device ticket,
    // no workgroup dispatch between two sources of a CU.  The loop around the body costs the kernel its register allocation:
    // 85 - 120 VGPR + 304 - 404 SGPR spills instead of 17 + 125, forward 1.29 -> 1.84 ms; a non-inlined body moves the spills
    // into the callee (scratch 572 B).  profiles/r06_ab_persistent_scan.log.)
    if (grid_body<FEAT, NCM, TILED, false>(a, rows_in_lds, R, cap, pass, last_pass, s_scan, s_bkt, blockIdx.x)) {
        __syncthreads();
        grid_body<FEAT, NCM, TILED, true>(a, rows_in_lds, R, cap, pass, last_pass, s_scan, s_bkt, blockIdx.x);
    }
}

// LDS budget of the grid kernel: two 512-thread workgroups per CU -- 160 KiB / 2, minus the kernel's static
// arrays (s_scan 2 KiB, s_bkt 128 B); one byte more and only one workgroup fits a CU (2.4 -> 3.9 ms)
constexpr size_t kGridLdsBudget = 78 * 1024 - 128 - (kGridNT - 512) * 4;

template <int FEAT, int NCM = 0>
bool forward_grid_launch(const ForwardArgs& a_in, int rows_in_lds, hipStream_t stream) {
    // visibility cache: item masks where this launch can record them (confocal; the live list is its index), per-face
    // words otherwise -- never both
    ForwardArgs b = a_in;
    const bool items = (NCM == 0 || NCM == 2) && b.vis_items && b.vis;
    if (items) b.vis = nullptr;
    else b.vis_items = nullptr;
    if (!items || NCM != 0) b.geo = nullptr;
    const ForwardArgs& a = b;
    LaunchNote scratch_note;
    LaunchNote& note = tl_note ? *tl_note : scratch_note;
    if (a.force_bvh) { note.reason = 1; return false; }
    if (a.sc.F < 64) { note.reason = 2; return false; }
    if (a.tile_list || a.sc.F > 8191) { note.reason = 6; return false; }              // 13-bit triangle index in the cell entries
    if (!lean_params_ok(a.sp)) { note.reason = 7; return false; }
#ifndef NLOS_GRID_RSCALE
#define NLOS_GRID_RSCALE 0.5f
#endif
    // finer cells shorten the per-sample walk for as long as the cell lists still fit: R = sqrt(F/2) always, and up to
    // sqrt(NLOS_GRID_RSCALE_MAX F) while a worst-case estimate of the entry count (each reachable face covering
    // (1 + 1.38 R/sqrt F)^2 cells -- the footprint calibrated on where bunny_5k starts to overflow --, every face reachable as on a
    // height field seen from the wall, 30 % margin)
    // stays below the capacity that resolution leaves.  Measured at 1 055 faces: forward 2.08 -> 1.93 ms.
#ifndef NLOS_GRID_RSCALE_MAX
#define NLOS_GRID_RSCALE_MAX 1.6f
#endif
    const size_t nblk = ((size_t)a.sc.F + 63) / 64;
    auto fixed_bytes = [&](int r) {
        const size_t r2 = ((size_t)r + 1) / 2;
        size_t union_words = ((r2 * r2 + 1) & ~(size_t)1) + 2 * nblk;
        if (union_words < (size_t)kGridWaves * kQueueWords) union_words = (size_t)kGridWaves * kQueueWords;
        return 32 + (rows_in_lds ? (size_t)a.sp.nbins * sizeof(double) : 0) + (((size_t)r * r + 2) & ~(size_t)1) * 4 +
               ((union_words + 1) & ~(size_t)1) * 4;
    };
    int R = (int)lrintf(sqrtf(NLOS_GRID_RSCALE * (float)a.sc.F));
    R = std::min(std::max(R, 8), 96);
    {
        const int r_hi = std::min((int)lrintf(sqrtf(NLOS_GRID_RSCALE_MAX * (float)a.sc.F)), 96);
        const float sf = sqrtf((float)a.sc.F);
        for (int r = r_hi; r > R; --r) {
            const size_t fx = fixed_bytes(r);
            if (fx >= kGridLdsBudget) continue;
            const float per_face = 1.f + 1.38f * (float)r / sf;
            const float est = 1.3f * (float)a.sc.F * per_face * per_face;
            if (est <= (float)((kGridLdsBudget - fx) / 4)) { R = r; break; }
        }
    }
    const size_t fixed = fixed_bytes(R);
    if (fixed + 4 * 2 * (size_t)a.sc.F > kGridLdsBudget || !a.live || !a.cov) { note.reason = 3; return false; }   // want room for >= 2 entries per face
    size_t cap = (kGridLdsBudget - fixed) / 4;
    const size_t lds = fixed + cap * 4;
    const size_t lds_big = 150 * 1024;
    note_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_forward_grid<FEAT, NCM, false, 0>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute(k_forward_grid, LDS)");
    if (a.retry)
        note_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_forward_grid<FEAT, NCM, false, 1>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_big), "hipFuncSetAttribute(k_forward_grid big-LDS)");
    note.backend = 1; note.reason = 0; note.grid_R = R; note.tiles = 1; note.tile_cap = 0;
    note.retry_workgroups = a.retry ? a.src.L : 0;
    if (items) note.vis_items = 1;
    if (NCM == 0) note.prologue_done = 1;      // (ForwardArgs::zero / pathlengths, if any, are handled by the kernel's first workgroups)
    // kScan slots of slack: the walk reads kScan entries per trip and may touch the slots after the last list
    const int last_pass = a.retry ? 1 : 0;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_forward_grid<FEAT, NCM, false, 0>), dim3(a.src.L), dim3(kGridNT), lds, stream, a, rows_in_lds, R,
                       (int)cap - kScan, 0, last_pass);
    // lazy scene build: the tree follows now if a first-launch workgroup asked for it (two launches that leave at once otherwise)
    if (a.retry && a.need_tree && note.lazy_build) launch_build_tree(*note.lazy_build, true, stream);
    if (a.retry)      // sources whose cell lists overflowed even on the coarsened grid: once more with the whole CU's LDS
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_forward_grid<FEAT, NCM, false, 1>), dim3(a.src.L), dim3(kGridNT), lds_big, stream, a,
                           rows_in_lds, R, (int)((lds_big - fixed) / 4) - kScan, 1, last_pass);
    return true;
}

// meshes beyond one workgroup's LDS: one workgroup per (source, slope-space tile)
template <int FEAT, int NCM = 0>
bool forward_tiled_launch(const ForwardArgs& a, int rows_in_lds, hipStream_t stream) {
    LaunchNote scratch_note;
    LaunchNote& note = tl_note ? *tl_note : scratch_note;
    if (a.force_bvh) { note.reason = 1; return false; }
    if (!a.tile_list || !a.tile_count || !a.live || !a.cov) return false;                      // (reason recorded by the single-workgroup launcher)
    if (!lean_params_ok(a.sp)) { note.reason = 7; return false; }
    if (a.tiles_x * a.tiles_y > 1024 || a.tiles_x < 1 || a.tiles_y < 1) { note.reason = 4; return false; }
    if ((NCM != 0) != (a.src.sensor != nullptr)) return false;
    uint32_t* const visout = NCM == 1 ? a.vis2 : a.vis;
#ifndef NLOS_TILE_R
#define NLOS_TILE_R 32
#endif
    const int R = NLOS_TILE_R;
    const size_t R2 = (R + 1) / 2;
    const size_t mask_blocks = ((size_t)a.tile_cap + 63) / 64;
    size_t union_words = ((R2 * R2 + 1) & ~(size_t)1) + 2 * mask_blocks;
    if (union_words < (size_t)kGridWaves * kQueueWords) union_words = (size_t)kGridWaves * kQueueWords;
    const size_t fixed = 32 + (rows_in_lds ? (size_t)a.sp.nbins * sizeof(double) : 0) + (((size_t)R * R + 2) & ~(size_t)1) * 4 +
                         ((union_words + 1) & ~(size_t)1) * 4;
    if (fixed + 4 * 4096 > kGridLdsBudget) { note.reason = 4; return false; }
    const size_t cap = (kGridLdsBudget - fixed) / 4;
    const size_t lds = fixed + cap * 4;
    // partial rows / visibility words of the tiles are combined with atomics: start from zero
    if (NCM != 1 && rows_in_lds && a.rows) launch_zero_f64(a.rows, (size_t)a.src.L * a.sp.nbins, stream);
    if (visout) note_hip(hipMemsetAsync(visout, 0, sizeof(uint32_t) * (size_t)a.src.L * a.vis_words * a.sc.F, stream), "hipMemsetAsync(vis)");
    const size_t nwg = (size_t)a.src.L * a.tiles_x * a.tiles_y;
    note_hip(hipMemsetAsync(a.tile_count, 0, sizeof(int) * 2 * nwg, stream), "hipMemsetAsync(tile_count)");      // subset sizes + retry flags
    note.backend = 2; note.reason = 6; note.grid_R = R; note.tiles = a.tiles_x * a.tiles_y; note.tile_cap = a.tile_cap;
    note.retry_workgroups = (int)nwg;
    hipLaunchKernelGGL(k_tile_bin, dim3(a.src.L), dim3(512), 0, stream, a, R, NCM == 1 ? 1 : 0);
    // second launch for the tiles whose cell lists overflow: the whole CU's LDS for one workgroup
    const size_t lds_big = 150 * 1024;
    const size_t cap_big = (lds_big - fixed) / 4;
    note_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_forward_grid<FEAT, NCM, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_big), "hipFuncSetAttribute(k_forward_grid tiled)");
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_forward_grid<FEAT, NCM, true>), dim3((unsigned)nwg), dim3(kGridNT), lds, stream, a, rows_in_lds, R,
                       (int)cap - kScan, 0, 1);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_forward_grid<FEAT, NCM, true>), dim3((unsigned)nwg), dim3(kGridNT), lds_big, stream, a, rows_in_lds,
                       R, (int)cap_big - kScan, 1, 1);
    return true;
}

template <int FEAT>
bool grid_dispatch(const ForwardArgs& a, int rows_in_lds, hipStream_t stream) {
    if (a.src.sensor) {
        {
            // row N: one grid pass per end point of the pair (sensor-leg visibility bits first, then the
            // laser pass that ANDs them and bins)
            if (a.vis2 && !a.mode_intensity) {
                ForwardArgs p1 = a;
                p1.rows = nullptr;
                if (a.tile_list)     // large mesh: both passes through the tiled grid
                    return forward_tiled_launch<FEAT, 1>(p1, 0, stream) && forward_tiled_launch<FEAT, 2>(a, rows_in_lds, stream);
                return forward_grid_launch<FEAT, 1>(p1, 0, stream) && forward_grid_launch<FEAT, 2>(a, rows_in_lds, stream);
            }
        }
        if (tl_note) tl_note->reason = a.force_bvh ? 1 : 5;
        return false;
    }
    if (forward_grid_launch<FEAT>(a, rows_in_lds, stream)) return true;
    return forward_tiled_launch<FEAT>(a, rows_in_lds, stream);
}

}  // namespace

// record pass of the product: one launch of the single-workgroup grid per set of wall points (face normals, Lambertian);
// false: this scene is outside that kernel's range (the caller then renders the enumerated pairs)
bool launch_forward_record(const ForwardArgs& a, hipStream_t stream) {
    if (!a.rec_d || !a.rec_ff || a.src.sensor || a.tile_list) return false;
    const int feat = feat_of(a.sc, a.sp);
    if (feat != 0 && !a.rec_ext) return false;
#ifdef NLOS_ONLY_FEAT0
    return feat == 0 && forward_grid_launch<0, 3>(a, 0, stream);
#else
    switch (feat) {
        case 0: return forward_grid_launch<0, 3>(a, 0, stream);
        case FEAT_VN: return forward_grid_launch<FEAT_VN, 3>(a, 0, stream);
        case FEAT_ALB: return forward_grid_launch<FEAT_ALB, 3>(a, 0, stream);
        case FEAT_VN | FEAT_ALB: return forward_grid_launch<FEAT_VN | FEAT_ALB, 3>(a, 0, stream);
        default: return false;          // GGX: the pair path
    }
#endif
}

// true: launched (single-workgroup grid, tiled grid or the two passes of non-confocal pairs);
// false: this render needs the BVH back-end
bool launch_forward_grid(const ForwardArgs& a, int rows_in_lds, hipStream_t stream) {
#ifdef NLOS_ONLY_FEAT0             // ISA studies and syntax checks: one feature set instead of eight (12 s instead of 100 s of hipcc)
    return grid_dispatch<0>(a, rows_in_lds, stream);
#else
    switch (feat_of(a.sc, a.sp)) {
        case 0: return grid_dispatch<0>(a, rows_in_lds, stream);
        case 1: return grid_dispatch<1>(a, rows_in_lds, stream);
        case 2: return grid_dispatch<2>(a, rows_in_lds, stream);
        case 3: return grid_dispatch<3>(a, rows_in_lds, stream);
        case 4: return grid_dispatch<4>(a, rows_in_lds, stream);
        case 5: return grid_dispatch<5>(a, rows_in_lds, stream);
        case 6: return grid_dispatch<6>(a, rows_in_lds, stream);
        default: return grid_dispatch<7>(a, rows_in_lds, stream);
    }
#endif
}

}  // namespace nlos
