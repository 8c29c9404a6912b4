// optimiser.hip -- the steps right after the render in every optimisation loop of the reference,
// kept on the device so that transient, gradient and vertices never cross PCIe.
//
//   Adam_Modified   exp_bunny/adam_modified.py:62-107 -- Adam whose denominator is the ROW MEAN of
//                   sqrt(v) + eps (one step length per vertex, shared by x, y, z)
//   weighting       exp_bunny/rendering.py:208-217 create_weighting_function
//   weighted L2     exp_bunny/rendering.py:360-364 (L1 term of evaluate_loss_with_*)
#include "nlos_kernels.h"

namespace nlos {
namespace {

constexpr int kMaxCols = 8;

__global__ __launch_bounds__(256) void k_adam_modified(AdamArgs a) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.rows) return;
    if (a.row_mask && !a.row_mask[r]) return;
    const int C = a.cols;
    float m[kMaxCols];
    float denom_sum = 0.0f;
    for (int c = 0; c < C; ++c) {
        const size_t i = (size_t)r * C + c;
        // p.grad.data = torch.from_numpy(grad).float()   (exp_bunny/test.py:212-213)
        float g = a.grad64 ? (float)a.grad64[i] : a.grad32[i];
        if (a.weight_decay != 0.0f) g = g + a.weight_decay * a.params[i];
        const float mi = a.exp_avg[i] * a.beta1 + a.one_minus_beta1 * g;
        const float vi = a.exp_avg_sq[i] * a.beta2 + a.one_minus_beta2 * g * g;
        a.exp_avg[i] = mi;
        a.exp_avg_sq[i] = vi;
        float vd = vi;
        if (a.max_exp_avg_sq) {
            vd = fmaxf(a.max_exp_avg_sq[i], vi);
            a.max_exp_avg_sq[i] = vd;
        }
        m[c] = mi;
        denom_sum += sqrtf(vd) + a.eps;
    }
    const float new_denom = denom_sum / (float)C;          // torch.mean(denom, dim=1, keepdim=True)
    for (int c = 0; c < C; ++c) {
        const size_t i = (size_t)r * C + c;
        a.params[i] = a.params[i] + (-a.step_size) * (m[c] / new_denom);   // addcdiv_(-step_size, exp_avg, new_denom)
    }
}

// one 1024-thread workgroup: max, then sum of the powered terms, then the rescale.  Runs once per
// optimisation (the weights are fixed), so three passes by one workgroup beat three launches + atomics
// and are deterministic.
__global__ __launch_bounds__(1024) void k_weighting(const double* __restrict__ data, size_t n, double gamma,
                                                    double* __restrict__ weight) {
    __shared__ double s_red[16];
    __shared__ double s_bcast;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double mx = -1.0e308;
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) mx = fmax(mx, data[i]);
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_down(mx, off));
    if (lane == 0) s_red[wave] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = s_red[0];
        for (int w = 1; w < 16; ++w) m = fmax(m, s_red[w]);
        s_bcast = m;
    }
    __syncthreads();
    const double i_max = s_bcast;
    double sum = 0.0;
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) {
        const double w = pow(data[i] / i_max + 0.1, gamma);
        weight[i] = w;
        sum += w;
    }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
    __syncthreads();
    if (lane == 0) s_red[wave] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += s_red[w];
        s_bcast = t;
    }
    __syncthreads();
    const double total = s_bcast;
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) weight[i] = weight[i] / total * (double)n;
}

__global__ __launch_bounds__(256) void k_weighted_l2(const double* __restrict__ transient, const double* __restrict__ data,
                                                     const double* __restrict__ weight, size_t n, double inv_rows,
                                                     double* out) {
    double acc = 0.0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double d = transient[i] - data[i];
        acc += (weight ? weight[i] : 1.0) * d * d;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0 && acc != 0.0) unsafeAtomicAdd(out, acc * inv_rows);
}

}  // namespace

void launch_adam_modified(const AdamArgs& a, hipStream_t stream) {
    if (a.rows <= 0 || a.cols <= 0 || a.cols > kMaxCols) return;
    hipLaunchKernelGGL(k_adam_modified, dim3((a.rows + 255) / 256), dim3(256), 0, stream, a);
}

void launch_weighting(const double* data, size_t n, double gamma, double* weight, hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_weighting, dim3(1), dim3(1024), 0, stream, data, n, gamma, weight);
}

void launch_weighted_l2(const double* transient, const double* data, const double* weight, size_t n, int rows,
                        double* out, hipStream_t stream) {
    launch_zero_f64(out, 1, stream);
    if (n == 0 || rows <= 0) return;
    size_t g = (n + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(k_weighted_l2, dim3((unsigned)g), dim3(256), 0, stream, transient, data, weight, n,
                       1.0 / (double)rows, out);
}

}  // namespace nlos
