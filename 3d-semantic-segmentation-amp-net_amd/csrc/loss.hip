// loss.hip -- the loss recipe of train_pointnet-attention.py:138,445,463-467 outside the head kernel:
//   reg = || I - F F^T ||_F over the whole [n, 64, 64] stack of feature transforms (forward and gradient),
//   weighted cross-entropy gradient d(ce)/d(logits) for logits [B, C, P], targets [B, P] (-1 = ignore).
#include "kernels.h"

namespace ampnet {

// one block per matrix: G = I - F F^T held in LDS; out_part[m] = sum G^2; optionally keeps G for the backward
// 1024 threads per matrix (four outputs each): B = 64 matrices are 64 workgroups, so the launch is as long as ONE workgroup takes
constexpr int REG_T = 1024;
__global__ __launch_bounds__(REG_T) void reg_fwd_kernel(const float *__restrict__ F, float *__restrict__ part, float *__restrict__ G)
{
    __shared__ float sF[64][65];
    __shared__ float red[REG_T / 64];
    const int m = blockIdx.x, tid = threadIdx.x;
    for (int e = tid; e < 4096; e += REG_T) sF[e / 64][e % 64] = F[(size_t)m * 4096 + e];
    __syncthreads();
    float acc = 0.f;
    for (int e = tid; e < 4096; e += REG_T) {
        const int i = e / 64, j = e % 64;
        float d = 0.f;
#pragma unroll 8
        for (int k = 0; k < 64; ++k) d = fmaf(sF[i][k], sF[j][k], d);
        const float g = (i == j ? 1.0f : 0.0f) - d;
        if (G) G[(size_t)m * 4096 + e] = g;
        acc = fmaf(g, g, acc);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < REG_T / 64; ++w) v += red[w];
        part[m] = v;
    }
}

__global__ void reg_finalize_kernel(const float *__restrict__ part, int n, float *__restrict__ out)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += (double)part[i];
        out[0] = (float)sqrt(s);
    }
}

// dF += coef * d(reg)/dF,  d(reg)/dF = -2 G F / reg   (G symmetric)
// lead > 0: dF is a stack of lead + n matrices that is WRITTEN: zeros in the first `lead` (the windows the regulariser does not see), the
// gradient itself in the last n (no accumulate): the caller then needs no zero fill of the stack (9.4 MB and a launch per step)
__global__ __launch_bounds__(REG_T) void reg_bwd_kernel(const float *__restrict__ F, const float *__restrict__ G,
                                                     const float *__restrict__ reg, float coef, float *__restrict__ dF, int lead, int write)
{
    __shared__ float sF[64][65], sG[64][65];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < lead) {
        float4 *z = reinterpret_cast<float4 *>(dF + (size_t)blockIdx.x * 4096);
        for (int e = tid; e < 1024; e += REG_T) z[e] = float4{0.f, 0.f, 0.f, 0.f};
        return;
    }
    const int m = (int)blockIdx.x - lead;
    dF += (size_t)lead * 4096;
    for (int e = tid; e < 4096; e += REG_T) {
        sF[e / 64][e % 64] = F[(size_t)m * 4096 + e];
        sG[e / 64][e % 64] = G[(size_t)m * 4096 + e];
    }
    __syncthreads();
    const float r = reg[0];
    const float k = r > 0.f ? -2.0f * coef / r : 0.f;
    for (int e = tid; e < 4096; e += REG_T) {
        const int i = e / 64, j = e % 64;
        float d = 0.f;
#pragma unroll 8
        for (int l = 0; l < 64; ++l) d = fmaf(sG[i][l], sF[l][j], d);
        if (write) dF[(size_t)m * 4096 + e] = k * d;
        else dF[(size_t)m * 4096 + e] += k * d;
    }
}

// dlogits[b, c, p] = gscale * w[t] / sum_w * (softmax(logits[b, :, p])[c] - [c == t]); 0 where t == -1
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float *__restrict__ logits, const long long *__restrict__ targets,
                                                    const float *__restrict__ class_w, const float *__restrict__ loss2,
                                                    float gscale, int B, int C, int P, float *__restrict__ dlogits)
{
    const size_t row = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (row >= (size_t)B * P) return;
    const int b = (int)(row / P), p = (int)(row % P);
    const long long t = targets[row];
    float l[8];
    float m = -__builtin_inff();
    for (int c = 0; c < C; ++c) {
        l[c] = logits[((size_t)b * C + c) * P + p];
        m = fmaxf(m, l[c]);
    }
    float se = 0.f;
    for (int c = 0; c < C; ++c) {
        l[c] = expf(l[c] - m);
        se += l[c];
    }
    const bool live = t >= 0 && t < C;
    const float w = live ? (class_w ? class_w[t] : 1.0f) : 0.f;
    const float k = live ? gscale * w / loss2[1] : 0.f;
    for (int c = 0; c < C; ++c) dlogits[((size_t)b * C + c) * P + p] = k * (l[c] / se - (c == (int)t ? 1.0f : 0.0f));
}

}  // namespace ampnet

using namespace ampnet;

extern "C" int ampnet_reg_loss_fwd_f32(const float *feat_T, int n, float *reg_out, float *G, float *part, void *stream)
{
    AMPNET_REQUIRE(feat_T && reg_out && part && n >= 1, "ampnet_reg_loss_fwd_f32: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(reg_fwd_kernel, dim3(n), dim3(REG_T), 0, st, feat_T, part, G);
    hipLaunchKernelGGL(reg_finalize_kernel, dim3(1), dim3(64), 0, st, part, n, reg_out);
    return check_launch("reg_loss_fwd");
}

extern "C" int ampnet_reg_loss_bwd_f32(const float *feat_T, const float *G, const float *reg, float coef, int n, float *d_feat_T,
                                       void *stream)
{
    AMPNET_REQUIRE(feat_T && G && reg && d_feat_T && n >= 1, "ampnet_reg_loss_bwd_f32: bad arguments");
    hipLaunchKernelGGL(reg_bwd_kernel, dim3(n), dim3(REG_T), 0, (hipStream_t)stream, feat_T, G, reg, coef, d_feat_T, 0, 0);
    return check_launch("reg_loss_bwd");
}

extern "C" int ampnet_reg_loss_bwd_stack_f32(const float *feat_T, const float *G, const float *reg, float coef, int n, int n_total, float *d_feat_T_stack,
                                             void *stream)
{
    AMPNET_REQUIRE(feat_T && G && reg && d_feat_T_stack && n >= 1 && n_total >= n, "ampnet_reg_loss_bwd_stack_f32: bad arguments");
    hipLaunchKernelGGL(reg_bwd_kernel, dim3(n_total), dim3(REG_T), 0, (hipStream_t)stream, feat_T, G, reg, coef, d_feat_T_stack, n_total - n, 1);
    return check_launch("reg_loss_bwd_stack");
}

extern "C" int ampnet_ce_bwd_f32(const float *logits, const long long *targets, const float *class_w, const float *loss2,
                                 float grad_scale, int B, int C, int P, float *dlogits, void *stream)
{
    AMPNET_REQUIRE(logits && targets && loss2 && dlogits, "ampnet_ce_bwd_f32: null pointer");
    AMPNET_REQUIRE(B >= 1 && P >= 1 && C >= 1 && C <= 8, "ampnet_ce_bwd_f32: bad sizes");
    const size_t rows = (size_t)B * P;
    hipLaunchKernelGGL(ce_bwd_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, logits, targets, class_w,
                       loss2, grad_scale, B, C, P, dlogits);
    return check_launch("ce_bwd_kernel");
}

// ---- a11 on the device: confusion counts of one batch (utils/get_metrics.py:6-31 are quotients of these) ---------------------------
namespace ampnet {
constexpr int CONF_MAX_C = 8;
__global__ __launch_bounds__(256) void confusion_kernel(const long long *__restrict__ preds, const long long *__restrict__ targets, long long n, int C,
                                                       unsigned long long *__restrict__ counts)
{
    __shared__ unsigned int h[CONF_MAX_C * CONF_MAX_C + 1];
    for (int i = threadIdx.x; i <= C * C; i += blockDim.x) h[i] = 0u;
    __syncthreads();
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long t = targets[i], p = preds[i];
        if (t < 0 || t >= C || p < 0 || p >= C) atomicAdd(&h[C * C], 1u);          // ignored (-1 = padding)
        else atomicAdd(&h[(int)t * C + (int)p], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i <= C * C; i += blockDim.x)
        if (h[i]) atomicAdd(&counts[i], (unsigned long long)h[i]);                  // integer: order-independent, exact
}
}  // namespace ampnet

extern "C" int ampnet_confusion_i64(const long long *preds, const long long *targets, long long n, int n_classes, long long *counts, void *stream)
{
    AMPNET_REQUIRE(preds && targets && counts, "ampnet_confusion_i64: null pointer");
    AMPNET_REQUIRE(n >= 0 && n_classes >= 1 && n_classes <= ampnet::CONF_MAX_C, "ampnet_confusion_i64: n=%lld classes=%d", n, n_classes);
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(counts, 0, (size_t)(n_classes * n_classes + 1) * sizeof(long long), st) != hipSuccess)
        return ampnet::fail(AMPNET_E_LAUNCH, "ampnet_confusion_i64: memset failed");
    if (n == 0) return AMPNET_OK;
    const int blocks = (int)((n + 256LL * 16 - 1) / (256LL * 16)) < 1024 ? (int)((n + 256LL * 16 - 1) / (256LL * 16)) : 1024;
    hipLaunchKernelGGL(ampnet::confusion_kernel, dim3(blocks), dim3(256), 0, st, preds, targets, n, n_classes,
                       reinterpret_cast<unsigned long long *>(counts));
    return ampnet::check_launch("confusion_kernel");
}

// ---- a5: the key-padding mask of train_loop, on the device ------------------------------------------------------------------------------
// train_pointnet-attention.py:428-431 builds it from the cluster-concatenated targets [B, W * N] as
//     (targets_pc.view(B, -1, W) == -1).all(dim=1)        i.e. mask[b, w] = all_i (targets[b, i * W + w] == -1)
// (a view(B, -1, W) of cluster-major data, restated literally: SURVEY F-notes).  As torch ops on the GPU that is a compare, two fills, a
// strided reduction and a copy: five launches, 46 us per step for 9 MB; here one workgroup per sample reads its row once.
namespace ampnet {
// one workgroup of 1024 threads per sample: its row is read once, up to eight loads in flight per thread
__global__ __launch_bounds__(1024) void pad_mask_kernel(const long long *__restrict__ targets, int P, int W, uint8_t *__restrict__ mask)
{
    __shared__ unsigned int s_seen[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    const long long *row = targets + (size_t)b * P;
    unsigned int seen = 0u;                                   // bit w: some element of column w is not -1
    for (int base = tid; base < P; base += 1024 * 8) {
        long long v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = base + 1024 * u < P ? row[base + 1024 * u] : -1;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (v[u] != -1) seen |= 1u << ((base + 1024 * u) % W);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) seen |= __shfl_xor(seen, o);
    if ((tid & 63) == 0) s_seen[tid >> 6] = seen;
    __syncthreads();
    if (tid < W) {
        unsigned int all = 0u;
#pragma unroll
        for (int i = 0; i < 16; ++i) all |= s_seen[i];
        mask[(size_t)b * W + tid] = ((all >> tid) & 1u) ? 0 : 1;
    }
}
}  // namespace ampnet

extern "C" int ampnet_pad_mask_i64(const long long *targets, int B, int P, int W, uint8_t *mask, void *stream)
{
    AMPNET_REQUIRE(targets && mask, "ampnet_pad_mask_i64: null pointer");
    AMPNET_REQUIRE(B >= 1 && P >= 1 && W >= 1 && W <= 32 && P % W == 0, "ampnet_pad_mask_i64: B=%d P=%d W=%d (W <= 32, P %% W == 0)", B, P, W);
    hipLaunchKernelGGL(ampnet::pad_mask_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, targets, P, W, mask);
    return ampnet::check_launch("pad_mask_kernel");
}
