// regulariser.hip -- mesh regulariser gradients on the device.
//
// Reference: smoothed_transient/stratifiedStreamedGradientRenderer.cpp:27-180
//   streamed_render_curvature_grad   : g_j = cross(n, e_j / 2) = d(area)/d v_j per face
//   streamed_render_normal_smoothing : value = sum_f A_f (1 - nbar_f . n_f),
//                                      nbar_f = normalise(A_f n_f + sum_nbr A_g n_g),
//                                      g_j = cross(n_f - nbar_f, e_j / 2)
// The reference stores per-vertex results with `=` (and shares one buffer between its TBB threads),
// so what it returns is "whichever incident face wrote last".  overwrite = 0 accumulates over the
// incident faces (the gradient the formulas describe, the default); overwrite = 1 reproduces the
// serial outcome deterministically: the incident face with the highest index wins (atomicMax
// ownership pass, then one writer per vertex).  O(F) work, three tiny launches; the point of having
// it on the device is that the whole gradient stays in HBM until the optimiser step.
#include "nlos_device.h"
#include "nlos_kernels.h"

namespace nlos {
namespace {

struct FaceGeo { V3 p0, p1, p2, n; float area; int i0, i1, i2; };

__device__ __forceinline__ FaceGeo face_geo(const float* __restrict__ V, const int32_t* __restrict__ F, int f) {
    FaceGeo g;
    g.i0 = F[3 * f]; g.i1 = F[3 * f + 1]; g.i2 = F[3 * f + 2];
    g.p0 = ld3(V + 3 * (size_t)g.i0);
    g.p1 = ld3(V + 3 * (size_t)g.i1);
    g.p2 = ld3(V + 3 * (size_t)g.i2);
    V3 nr = cross(g.p1 - g.p0, g.p2 - g.p0);
    g.area = sqrtf(dot(nr, nr)) / 2;
    g.n = nr * (1.0f / (2 * g.area));
    return g;
}

__global__ __launch_bounds__(256) void k_reg_normal_area(RegulariserArgs a) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= a.F) return;
    const FaceGeo g = face_geo(a.vertices, a.faces, f);
    a.area[f] = (double)g.area;
    a.normal[3 * (size_t)f] = (double)g.n.x;
    a.normal[3 * (size_t)f + 1] = (double)g.n.y;
    a.normal[3 * (size_t)f + 2] = (double)g.n.z;
    if (a.overwrite && g.area > 0.0f) {
        atomicMax(&a.owner[g.i0], f);
        atomicMax(&a.owner[g.i1], f);
        atomicMax(&a.owner[g.i2], f);
    }
}

__global__ __launch_bounds__(256) void k_reg_gradient(RegulariserArgs a) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    double val = 0.0;
    if (f < a.F) {
        const FaceGeo g = face_geo(a.vertices, a.faces, f);
        bool live = g.area > 0.0f;
        V3 d = g.n;
        if (live) {
            if (a.affinity) {
                V3 n = g.n * g.area;
                float wsum = g.area;
                for (int i = 0; i < 3; ++i) {
                    const int nb = a.affinity[3 * (size_t)f + i];
                    if (nb < 0) continue;
                    const float an = (float)a.area[nb];
                    if (!(an > 0.0f)) continue;
                    const V3 n1 = mk((float)a.normal[3 * (size_t)nb], (float)a.normal[3 * (size_t)nb + 1],
                                     (float)a.normal[3 * (size_t)nb + 2]);
                    n = n + n1 * an;
                    wsum += an;
                }
                const float len = sqrtf(dot(n, n));
                live = len > 1e-3f * wsum;       // neighbourhood normals cancel: skipped (0/0 or noise in the reference)
                if (live) {
                    n = n * (1.0f / len);
                    val = (double)g.area * (double)(1 - dot(n, g.n));
                    d = g.n - n;
                }
            }
        }
        if (live) {
            const V3 e[3] = {g.p2 - g.p1, g.p0 - g.p2, g.p1 - g.p0};
            const int vi[3] = {g.i0, g.i1, g.i2};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const V3 q = cross(d, e[j] * 0.5f);
                double* o = a.gradient + 3 * (size_t)vi[j];
                if (a.overwrite) {
                    if (a.owner[vi[j]] == f) { o[0] = (double)q.x; o[1] = (double)q.y; o[2] = (double)q.z; }
                } else {
                    unsafeAtomicAdd(&o[0], (double)q.x);
                    unsafeAtomicAdd(&o[1], (double)q.y);
                    unsafeAtomicAdd(&o[2], (double)q.z);
                }
            }
        }
    }
    if (a.value) {
        for (int off = 32; off > 0; off >>= 1) val += __shfl_down(val, off);
        if ((threadIdx.x & 63) == 0 && val != 0.0) unsafeAtomicAdd(a.value, val);
    }
}

__global__ __launch_bounds__(256) void k_reg_init(RegulariserArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3 * a.V) a.gradient[i] = 0.0;
    if (a.owner && i < a.V) a.owner[i] = -1;
    if (a.value && i == 0) *a.value = 0.0;
}

}  // namespace

void launch_regulariser(const RegulariserArgs& a, hipStream_t stream) {
    if (a.V <= 0) return;
    hipLaunchKernelGGL(k_reg_init, dim3((3 * a.V + 255) / 256), dim3(256), 0, stream, a);
    if (a.F <= 0) return;
    const dim3 grid((a.F + 255) / 256);
    hipLaunchKernelGGL(k_reg_normal_area, grid, dim3(256), 0, stream, a);
    hipLaunchKernelGGL(k_reg_gradient, grid, dim3(256), 0, stream, a);
}

}  // namespace nlos
