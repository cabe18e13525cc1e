// dca_train.hip — BatchNorm1d in TRAINING mode for the cost-to-go network's training step.
//
// Why this exists: in nnet_utils.train_nnet (reference utils/nnet_utils.py:53-118) the network runs with
// nn.BatchNorm1d in train() (pytorch_models.py:57-86: ten BN layers over [batch, 5000] / [batch, 1000] activations).
// rocprofv3 of the PyTorch-ROCm step at batch 10 000 shows the framework's per-column reduction kernels
// (batch_norm_collect_statistics / batch_norm_backward_reduce, "channels_last" path for 2-D input) at 274 us and
// 424 us per call — 41 % of the whole step for tensors of 40-200 MB that HBM streams in 10-40 us.  These kernels do
// the same arithmetic as column reductions over the row-major activation matrix, optionally fused with the residual
// add and the ReLU that follow the BatchNorm in the network:
//
//   forward   mean_c, var_c (biased) over the n rows;  y = relu?( (x-mean)*invstd*gamma + beta (+ skip) )
//   backward  g = dy * (y > 0 if relu);  dbeta = sum g;  dgamma = sum g*xhat;
//             dx = gamma*invstd * (g - dbeta/n - xhat*dgamma/n);  dskip = g
//
// Layout: x, y, dy, dx, skip are row-major [n, c] float32 (what the GEMMs produce).  A workgroup owns a tile of
// 256 columns (one float4 per lane, 64 lanes) x a slice of rows; 4 waves stride the rows, so every wave instruction
// reads 1 KiB contiguous.  Column sums are accumulated in fp64 (no cancellation in E[x^2]-E[x]^2), written as
// per-slice partials and folded by a small second kernel — deterministic, no atomics.
#include "dca_common.h"

namespace dca {

constexpr int kBnSlices = 128;  // row slices per column tile: (c/256) x 128 workgroups fill the chip for c >= 1000

struct BnAcc {
    double a[4], b[4];
};

// stage 1 of both reductions.  MODE 0: a = sum x, b = sum x^2.  MODE 1: a = sum g, b = sum g*xhat.
template <int MODE>
__global__ __launch_bounds__(256) void k_bn_reduce(const float* __restrict__ x, const float* __restrict__ dy,
                                                   const float* __restrict__ y, const float* __restrict__ mean,
                                                   const float* __restrict__ invstd, int64_t n, int64_t c, int relu,
                                                   double* __restrict__ part /*[2][kBnSlices][c]*/) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t col = ((int64_t)blockIdx.x * 64 + lane) * 4;
    const int64_t rows_per = (n + kBnSlices - 1) / kBnSlices;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per, r1 = r0 + rows_per < n ? r0 + rows_per : n;
    const bool vec = col + 4 <= c && (c & 3) == 0;
    BnAcc acc{};
    float mu[4] = {0, 0, 0, 0}, is[4] = {0, 0, 0, 0};
    if (MODE == 1)
        for (int k = 0; k < 4; k++)
            if (col + k < c) {
                mu[k] = mean[col + k];
                is[k] = invstd[col + k];
            }
    if (col < c) {
        for (int64_t r = r0 + wv; r < r1; r += 4) {
            float xv[4] = {0, 0, 0, 0}, gv[4] = {0, 0, 0, 0}, yv[4] = {1, 1, 1, 1};
            const int64_t o = r * c + col;
            if (vec) {
                const float4 t = *reinterpret_cast<const float4*>(x + o);
                xv[0] = t.x, xv[1] = t.y, xv[2] = t.z, xv[3] = t.w;
                if (MODE == 1) {
                    const float4 d = *reinterpret_cast<const float4*>(dy + o);
                    gv[0] = d.x, gv[1] = d.y, gv[2] = d.z, gv[3] = d.w;
                    if (relu) {
                        const float4 q = *reinterpret_cast<const float4*>(y + o);
                        yv[0] = q.x, yv[1] = q.y, yv[2] = q.z, yv[3] = q.w;
                    }
                }
            } else {
                for (int k = 0; k < 4; k++)
                    if (col + k < c) {
                        xv[k] = x[o + k];
                        if (MODE == 1) {
                            gv[k] = dy[o + k];
                            if (relu) yv[k] = y[o + k];
                        }
                    }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (MODE == 0) {
                    acc.a[k] += (double)xv[k];
                    acc.b[k] += (double)xv[k] * (double)xv[k];
                } else {
                    const float g = yv[k] > 0.f ? gv[k] : 0.f;
                    acc.a[k] += (double)g;
                    acc.b[k] += (double)g * (double)((xv[k] - mu[k]) * is[k]);
                }
            }
        }
    }
    __shared__ BnAcc sh[4][64];
    sh[wv][lane] = acc;
    __syncthreads();
    if (wv == 0 && col < c) {
        for (int w = 1; w < 4; w++)
            for (int k = 0; k < 4; k++) {
                acc.a[k] += sh[w][lane].a[k];
                acc.b[k] += sh[w][lane].b[k];
            }
        for (int k = 0; k < 4; k++)
            if (col + k < c) {
                part[((int64_t)0 * kBnSlices + blockIdx.y) * c + col + k] = acc.a[k];
                part[((int64_t)1 * kBnSlices + blockIdx.y) * c + col + k] = acc.b[k];
            }
    }
}

// stage 2, forward: mean / invstd / unbiased variance per column
__global__ __launch_bounds__(256) void k_bn_stats_final(const double* __restrict__ part, int64_t n, int64_t c, double eps,
                                                        float* __restrict__ mean, float* __restrict__ invstd,
                                                        float* __restrict__ var_unbiased) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= c) return;
    double s = 0.0, q = 0.0;
    for (int b = 0; b < kBnSlices; b++) {
        s += part[((int64_t)0 * kBnSlices + b) * c + j];
        q += part[((int64_t)1 * kBnSlices + b) * c + j];
    }
    const double m = s / (double)n;
    double var = q / (double)n - m * m;  // biased (what normalises the batch), exact enough in fp64
    var = var > 0.0 ? var : 0.0;
    mean[j] = (float)m;
    invstd[j] = (float)(1.0 / sqrt(var + eps));
    var_unbiased[j] = (float)(n > 1 ? var * (double)n / (double)(n - 1) : var);  // feeds running_var
}

// stage 2, backward: dgamma / dbeta per column
__global__ __launch_bounds__(256) void k_bn_grad_final(const double* __restrict__ part, int64_t c, float* __restrict__ dgamma,
                                                       float* __restrict__ dbeta) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= c) return;
    double s = 0.0, q = 0.0;
    for (int b = 0; b < kBnSlices; b++) {
        s += part[((int64_t)0 * kBnSlices + b) * c + j];
        q += part[((int64_t)1 * kBnSlices + b) * c + j];
    }
    dbeta[j] = (float)s;
    dgamma[j] = (float)q;
}

// elementwise passes.  Same tiling as the reductions (a workgroup = 256 columns x a slice of rows, one float4 per lane,
// 4 waves striding the rows): the per-column parameters live in registers and no index arithmetic is left in the loop.
// FWD: y = relu?(xhat*gamma + beta (+ skip)).
// BWD: g = dy*(y>0); dx = gamma*invstd*(g - dbeta/n - xhat*dgamma/n); dskip = g.
template <bool FWD>
__global__ __launch_bounds__(256) void k_bn_apply(const float* __restrict__ x, const float* __restrict__ other /*skip | dy*/,
                                                  const float* __restrict__ y_in, const float* __restrict__ mean,
                                                  const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, const float* __restrict__ dgamma,
                                                  const float* __restrict__ dbeta, int64_t n, int64_t c, int relu,
                                                  float* __restrict__ out, float* __restrict__ out2 /*dskip or null*/) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t col = ((int64_t)blockIdx.x * 64 + lane) * 4;
    if (col >= c) return;
    const int64_t rows_per = (n + gridDim.y - 1) / gridDim.y;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per, r1 = r0 + rows_per < n ? r0 + rows_per : n;
    const bool vec = col + 4 <= c && (c & 3) == 0;
    const float inv_n = 1.0f / (float)n;
    // per column: y = x*sc + sh (fwd);  dx = g*ga - (x*sc2 + sh2) (bwd)
    float mu[4], is[4], p0[4], p1[4], p2[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int64_t j = col + k < c ? col + k : c - 1;
        mu[k] = mean[j];
        is[k] = invstd[j];
        if (FWD) {
            p0[k] = gamma[j];
            p1[k] = beta[j];
            p2[k] = 0.f;
        } else {
            p0[k] = gamma[j] * is[k];                 // gamma*invstd
            p1[k] = dbeta[j] * inv_n;                 // dbeta/n
            p2[k] = dgamma[j] * inv_n;                // dgamma/n
        }
    }
    for (int64_t r = r0 + wv; r < r1; r += 4) {
        const int64_t o = r * c + col;
        float xv[4] = {0, 0, 0, 0}, ov[4] = {0, 0, 0, 0}, yv[4] = {1, 1, 1, 1}, res[4], res2[4];
        if (vec) {
            const float4 t = *reinterpret_cast<const float4*>(x + o);
            xv[0] = t.x, xv[1] = t.y, xv[2] = t.z, xv[3] = t.w;
            if (other) {
                const float4 d = *reinterpret_cast<const float4*>(other + o);
                ov[0] = d.x, ov[1] = d.y, ov[2] = d.z, ov[3] = d.w;
            }
            if (!FWD && relu) {
                const float4 q = *reinterpret_cast<const float4*>(y_in + o);
                yv[0] = q.x, yv[1] = q.y, yv[2] = q.z, yv[3] = q.w;
            }
        } else {
            for (int k = 0; k < 4; k++)
                if (col + k < c) {
                    xv[k] = x[o + k];
                    if (other) ov[k] = other[o + k];
                    if (!FWD && relu) yv[k] = y_in[o + k];
                }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float xh = (xv[k] - mu[k]) * is[k];
            if (FWD) {
                const float v = xh * p0[k] + p1[k] + ov[k];
                res[k] = relu ? fmaxf(v, 0.f) : v;
            } else {
                const float g = yv[k] > 0.f ? ov[k] : 0.f;
                res[k] = p0[k] * (g - p1[k] - xh * p2[k]);
                res2[k] = g;
            }
        }
        if (vec) {
            *reinterpret_cast<float4*>(out + o) = make_float4(res[0], res[1], res[2], res[3]);
            if (!FWD && out2) *reinterpret_cast<float4*>(out2 + o) = make_float4(res2[0], res2[1], res2[2], res2[3]);
        } else {
            for (int k = 0; k < 4; k++)
                if (col + k < c) {
                    out[o + k] = res[k];
                    if (!FWD && out2) out2[o + k] = res2[k];
                }
        }
    }
}

}  // namespace dca

using namespace dca;

namespace {
// column tiles x row slices: >= ~2000 workgroups, >= 16 rows per slice
dim3 apply_grid(int64_t n, int64_t c) {
    const unsigned gx = (unsigned)((c + 255) / 256);
    int64_t gy = 4096 / gx;
    if (gy > (n + 15) / 16) gy = (n + 15) / 16;
    if (gy < 1) gy = 1;
    return dim3(gx, (unsigned)gy);
}
}  // namespace

extern "C" {

int64_t dca_bn_workspace_bytes(int64_t c) { return (int64_t)2 * kBnSlices * c * (int64_t)sizeof(double); }

int dca_bn_train_forward(const float* x, const float* skip, const float* gamma, const float* beta, int64_t n, int64_t c,
                         double eps, int relu, float* y, float* mean, float* invstd, float* var_unbiased, void* workspace,
                         int64_t workspace_bytes, void* stream) {
    DCA_ARG(x && gamma && beta && y && mean && invstd && var_unbiased && workspace);
    DCA_ARG(n >= 1 && c >= 1 && n * c < (1ll << 40) && workspace_bytes >= dca_bn_workspace_bytes(c));
    hipStream_t s = (hipStream_t)stream;
    double* part = reinterpret_cast<double*>(workspace);
    const dim3 rg((unsigned)((c + 255) / 256), kBnSlices);
    hipLaunchKernelGGL(k_bn_reduce<0>, rg, dim3(256), 0, s, x, nullptr, nullptr, nullptr, nullptr, n, c, 0, part);
    hipLaunchKernelGGL(k_bn_stats_final, dim3((unsigned)((c + 255) / 256)), dim3(256), 0, s, part, n, c, eps, mean, invstd,
                       var_unbiased);
    hipLaunchKernelGGL(k_bn_apply<true>, apply_grid(n, c), dim3(256), 0, s, x, skip, nullptr, mean, invstd, gamma, beta, nullptr, nullptr,
                       n, c, relu, y, nullptr);
    return launch_check("dca_bn_train_forward");
}

int dca_bn_train_backward(const float* dy, const float* x, const float* y, const float* mean, const float* invstd,
                          const float* gamma, int64_t n, int64_t c, int relu, float* dx, float* dskip, float* dgamma,
                          float* dbeta, void* workspace, int64_t workspace_bytes, void* stream) {
    DCA_ARG(dy && x && mean && invstd && gamma && dx && dgamma && dbeta && workspace && (y || !relu));
    DCA_ARG(n >= 1 && c >= 1 && n * c < (1ll << 40) && workspace_bytes >= dca_bn_workspace_bytes(c));
    hipStream_t s = (hipStream_t)stream;
    double* part = reinterpret_cast<double*>(workspace);
    const dim3 rg((unsigned)((c + 255) / 256), kBnSlices);
    hipLaunchKernelGGL(k_bn_reduce<1>, rg, dim3(256), 0, s, x, dy, y, mean, invstd, n, c, relu, part);
    hipLaunchKernelGGL(k_bn_grad_final, dim3((unsigned)((c + 255) / 256)), dim3(256), 0, s, part, c, dgamma, dbeta);
    hipLaunchKernelGGL(k_bn_apply<false>, apply_grid(n, c), dim3(256), 0, s, x, dy, y, mean, invstd, gamma, nullptr, dgamma, dbeta, n, c,
                       relu, dx, dskip);
    return launch_check("dca_bn_train_backward");
}

}  // extern "C"
