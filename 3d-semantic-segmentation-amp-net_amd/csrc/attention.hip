// attention.hip -- the small kernels of SegmentationWithAttention (pointNet/model/pointnetAtt.py:176-209):
// positional encoding (:183-185), the per-(sample, head) attention core of nn.MultiheadAttention(256, 8)
// (:187-190; sequence = the W <= 32 cluster tokens of one sample, head_dim 32) and the tail of the output layer (:207):
// transposed logits store, argmax and the loss recipe of train_pointnet-attention.py:138,445-450 (weighted CE partials).
// The in/out projections and conv_2 / conv_3 / conv_4 run on pw_gemm (head.hip).
#include "head.h"

namespace ampnet {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void posenc_tokens_kernel(const float *__restrict__ gl, const float *__restrict__ cent,
                                                           const float *__restrict__ w1, const float *__restrict__ b1,
                                                           const float *__restrict__ w2, const float *__restrict__ b2,
                                                           float *__restrict__ tok, float *__restrict__ hid_out, float *__restrict__ slope_out)
{
    __shared__ float hid[16];
    const int q = blockIdx.x, e = threadIdx.x;
    if (e < 16) {
        const float v = fmaf(cent[q * 2 + 1], w1[e * 2 + 1], fmaf(cent[q * 2 + 0], w1[e * 2 + 0], b1[e]));
        hid[e] = v > 0.f ? v : 0.01f * v;                        // F.leaky_relu_, slope 0.01
        if (hid_out) {                                           // train mode: what the backward of the two Linear layers needs
            hid_out[q * 16 + e] = hid[e];
            slope_out[q * 16 + e] = v > 0.f ? 1.0f : 0.01f;
        }
    }
    __syncthreads();
    float acc = b2[e];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc = fmaf(hid[k], w2[e * 16 + k], acc);
    tok[(size_t)q * HEAD_E + e] = gl[(size_t)q * HEAD_E + e] + acc;
}

int posenc_tokens(const float *gl, const float *cent, const float *w1, const float *b1, const float *w2, const float *b2,
                  float *tok, int Q, hipStream_t st, float *hid_out, float *slope_out)
{
    hipLaunchKernelGGL(posenc_tokens_kernel, dim3(Q), dim3(HEAD_E), 0, st, gl, cent, w1, b1, w2, b2, tok, hid_out, slope_out);
    return check_launch("posenc_tokens_kernel");
}

// one wave per (sample, head): W x W scores in LDS
__global__ __launch_bounds__(64) void attention_core_kernel(const float *__restrict__ qkv, const uint8_t *__restrict__ mask,
                                                           float *__restrict__ probs, float *__restrict__ ctx, int W,
                                                           float drop_p, uint32_t drop_base)
{
    __shared__ float sq[HEAD_MAX_W][HEAD_D + 1], sk[HEAD_MAX_W][HEAD_D + 1], sv[HEAD_MAX_W][HEAD_D + 1];
    __shared__ float sp[HEAD_MAX_W][HEAD_MAX_W + 1];
    const int b = blockIdx.x, hd = blockIdx.y, lane = threadIdx.x;
    const float qscale = 0.17677669529663687f;                   // 1 / sqrt(32)
    for (int e = lane; e < W * HEAD_D; e += 64) {
        const int i = e / HEAD_D, d = e % HEAD_D;
        const float *row = qkv + (size_t)(b * W + i) * (3 * HEAD_E) + hd * HEAD_D + d;
        sq[i][d] = row[0] * qscale;
        sk[i][d] = row[HEAD_E];
        sv[i][d] = row[2 * HEAD_E];
    }
    __syncthreads();
    for (int e = lane; e < W * W; e += 64) {
        const int i = e / W, j = e % W;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < HEAD_D; ++d) s = fmaf(sq[i][d], sk[j][d], s);
        if (mask && mask[b * W + j]) s = -__builtin_inff();
        sp[i][j] = s;
    }
    __syncthreads();
    if (lane < W) {
        const int i = lane;
        float m = -__builtin_inff();
        for (int j = 0; j < W; ++j) m = fmaxf(m, sp[i][j]);
        float sum = 0.f;
        for (int j = 0; j < W; ++j) {
            const float p = (m == -__builtin_inff()) ? 0.f : __expf(sp[i][j] - m);   // fully padded sample: zeros, not NaN
            sp[i][j] = p;
            sum += p;
        }
        const float inv = sum > 0.f ? 1.0f / sum : 0.f;
        for (int j = 0; j < W; ++j) sp[i][j] *= inv;
    }
    __syncthreads();
    const uint32_t thr = drop_threshold(drop_p);
    const float dscale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    for (int e = lane; e < W * W; e += 64) {
        const int i = e / W, j = e % W;
        const size_t o = ((size_t)(b * HEAD_HEADS + hd) * W + i) * W + j;
        float p = sp[i][j];
        if (probs) probs[o] = p;                                   // post-softmax, pre-dropout (saved for backward)
        if (drop_p > 0.f) p = (mix32((uint32_t)o ^ drop_base) >= thr) ? p * dscale : 0.f;
        sp[i][j] = p;
    }
    __syncthreads();
    for (int e = lane; e < W * HEAD_D; e += 64) {
        const int i = e / HEAD_D, d = e % HEAD_D;
        float acc = 0.f;
        for (int j = 0; j < W; ++j) acc = fmaf(sp[i][j], sv[j][d], acc);
        ctx[(size_t)(b * W + i) * HEAD_E + hd * HEAD_D + d] = acc;
    }
}

int attention_core(const float *qkv, const uint8_t *key_pad_mask, float *probs, float *ctx, int B, int W, float drop_p,
                   uint32_t drop_base, hipStream_t st)
{
    AMPNET_REQUIRE(W >= 1 && W <= HEAD_MAX_W, "attention_core: %d cluster tokens per sample, supported 1..%d", W, HEAD_MAX_W);
    hipLaunchKernelGGL(attention_core_kernel, dim3(B, HEAD_HEADS), dim3(64), 0, st, qkv, key_pad_mask, probs, ctx, W, drop_p, drop_base);
    return check_launch("attention_core_kernel");
}

// ----------------------------------------------------------------------------------------------------
// head_logits: the tail of the head once conv_4 itself has run on the matrix cores (pw_gemm<64,32> with the bn_3 + ReLU +
// dropout prologue, z4 [R, ldz4]): transposed logits store [B, C, P], argmax, weighted cross-entropy partials.
// thread = row; a workgroup's 256 rows are consecutive points of one sample (or straddle two: handled per row).
// ----------------------------------------------------------------------------------------------------
constexpr int HL_ROWS = 256;

__global__ __launch_bounds__(HL_ROWS) void head_logits_kernel(HeadOut a, const float *__restrict__ z4, int ldz4)
{
    __shared__ float red[HL_ROWS / 64][2];
    const int tid = threadIdx.x, row = blockIdx.x * HL_ROWS + tid;
    float wnll = 0.f, wsum = 0.f;
    if (row < a.R) {
        float acc[HEAD_MAX_CLASSES];
#pragma unroll
        for (int c = 0; c < HEAD_MAX_CLASSES; ++c) acc[c] = c < a.C ? z4[(size_t)row * ldz4 + c] : 0.f;
        const int b = row / a.P, p = row % a.P;
        float m = acc[0];
        int am = 0;
#pragma unroll
        for (int c = 0; c < HEAD_MAX_CLASSES; ++c) {
            if (c < a.C) {
                a.logits[((size_t)b * a.C + c) * a.P + p] = acc[c];
                if (acc[c] > m) {          // strict: first maximum wins, like torch.max(dim)
                    m = acc[c];
                    am = c;
                }
            }
        }
        if (a.preds) a.preds[row] = am;
        if (a.targets) {
            const long long t = a.targets[row];
            if (t >= 0 && t < a.C) {
                float se = 0.f, lt = 0.f;
#pragma unroll
                for (int c = 0; c < HEAD_MAX_CLASSES; ++c) {
                    if (c < a.C) {
                        se += expf(acc[c] - m);
                        if (c == (int)t) lt = acc[c];
                    }
                }
                const float w = a.class_w ? a.class_w[t] : 1.0f;
                wnll = w * ((m + logf(se)) - lt);
                wsum = w;
            }
        }
    }
    if (a.loss_part) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            wnll += __shfl_xor(wnll, off);
            wsum += __shfl_xor(wsum, off);
        }
        if ((tid & 63) == 0) {
            red[tid >> 6][0] = wnll;
            red[tid >> 6][1] = wsum;
        }
        __syncthreads();
        if (tid == 0) {
            a.loss_part[blockIdx.x * 2 + 0] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
            a.loss_part[blockIdx.x * 2 + 1] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
        }
    }
}

int head_logits(const HeadOut &a, const float *z4, int ldz4, int *n_blocks, hipStream_t st)
{
    AMPNET_REQUIRE(z4 && a.logits && ldz4 >= a.C, "head_logits: null pointer / ldz4");
    AMPNET_REQUIRE(a.C >= 1 && a.C <= HEAD_MAX_CLASSES, "head_logits: %d classes, supported 1..%d", a.C, HEAD_MAX_CLASSES);
    AMPNET_REQUIRE(a.P >= 1 && a.R % a.P == 0, "head_logits: rows %d not a multiple of points per sample %d", a.R, a.P);
    const int blocks = cdiv(a.R, HL_ROWS);
    if (n_blocks) *n_blocks = blocks;
    hipLaunchKernelGGL(head_logits_kernel, dim3(blocks), dim3(HL_ROWS), 0, st, a, z4, ldz4);
    return check_launch("head_logits_kernel");
}

__global__ __launch_bounds__(256) void loss_finalize_kernel(const float *__restrict__ part, int n, float *__restrict__ out)
{
    __shared__ double sa[256], sb[256];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        a += (double)part[2 * i];
        b += (double)part[2 * i + 1];
    }
    sa[threadIdx.x] = a;
    sb[threadIdx.x] = b;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (threadIdx.x < s) {
            sa[threadIdx.x] += sa[threadIdx.x + s];
            sb[threadIdx.x] += sb[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = (float)(sa[0] / sb[0]);
        out[1] = (float)sb[0];
    }
}

int loss_finalize(const float *loss_part, int n_blocks, float *loss_out, hipStream_t st)
{
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, loss_part, n_blocks, loss_out);
    return check_launch("loss_finalize_kernel");
}

}  // namespace ampnet
