// adam.hip -- C ABI: ampnet_adam_step_f32, one launch for a whole list of tensors (multi-tensor Adam).
// Replaces torch.optim.Adam.step as configured by the reference (train_pointnet-attention.py:140-141,469-470:
// lr, betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad):
//   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// `grad_scale` multiplies g on the fly (1 / world_size after a sum all-reduce).
#include "common.h"
#include <cmath>

namespace ampnet {

constexpr int ADAM_MAX_TENSORS = 72;
constexpr int ADAM_CHUNK = 2048;

struct AdamArgs {
    float *p[ADAM_MAX_TENSORS];
    const float *g[ADAM_MAX_TENSORS];
    float *m[ADAM_MAX_TENSORS];
    float *v[ADAM_MAX_TENSORS];
    int n[ADAM_MAX_TENSORS];
    float b1, b2, eps, step_size, inv_sqrt_bc2, grad_scale;
};

__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a)
{
    const int t = blockIdx.y;
    const int n = a.n[t];
    const int base = blockIdx.x * ADAM_CHUNK;
    if (base >= n) return;
    float *__restrict__ p = a.p[t];
    const float *__restrict__ g = a.g[t];
    float *__restrict__ m = a.m[t];
    float *__restrict__ v = a.v[t];
    const int end = min(base + ADAM_CHUNK, n);
    for (int i = base + threadIdx.x; i < end; i += 256) {
        const float gi = g[i] * a.grad_scale;
        const float mi = a.b1 * m[i] + (1.0f - a.b1) * gi;
        const float vi = a.b2 * v[i] + (1.0f - a.b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= a.step_size * mi / (sqrtf(vi) * a.inv_sqrt_bc2 + a.eps);
    }
}

}  // namespace ampnet

using namespace ampnet;

extern "C" int ampnet_adam_step_f32(float *const *params_host, const float *const *grads_host, float *const *m_host,
                                    float *const *v_host, const long *numel_host, int n_tensors, float lr, float beta1,
                                    float beta2, float eps, int step, float grad_scale, void *stream)
{
    AMPNET_REQUIRE(params_host && grads_host && m_host && v_host && numel_host, "ampnet_adam_step_f32: null pointer");
    AMPNET_REQUIRE(n_tensors >= 1 && step >= 1, "ampnet_adam_step_f32: n_tensors=%d step=%d", n_tensors, step);
    hipStream_t st = (hipStream_t)stream;
    const double bc1 = 1.0 - std::pow((double)beta1, step), bc2 = 1.0 - std::pow((double)beta2, step);
    for (int t0 = 0; t0 < n_tensors; t0 += ADAM_MAX_TENSORS) {
        AdamArgs a;
        const int cnt = n_tensors - t0 < ADAM_MAX_TENSORS ? n_tensors - t0 : ADAM_MAX_TENSORS;
        long nmax = 0;
        for (int i = 0; i < cnt; ++i) {
            AMPNET_REQUIRE(numel_host[t0 + i] >= 0 && numel_host[t0 + i] < (1L << 31), "ampnet_adam_step_f32: tensor %d too large", t0 + i);
            a.p[i] = params_host[t0 + i];
            a.g[i] = grads_host[t0 + i];
            a.m[i] = m_host[t0 + i];
            a.v[i] = v_host[t0 + i];
            a.n[i] = (int)numel_host[t0 + i];
            if (numel_host[t0 + i] > nmax) nmax = numel_host[t0 + i];
        }
        a.b1 = beta1;
        a.b2 = beta2;
        a.eps = eps;
        a.step_size = (float)((double)lr / bc1);
        a.inv_sqrt_bc2 = (float)(1.0 / std::sqrt(bc2));
        a.grad_scale = grad_scale;
        if (nmax == 0) continue;
        hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((nmax + ADAM_CHUNK - 1) / ADAM_CHUNK), cnt), dim3(256), 0, st, a);
        int rc = check_launch("adam_kernel");
        if (rc != AMPNET_OK) return rc;
    }
    return AMPNET_OK;
}
