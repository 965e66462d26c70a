// cls_head.hip -- C ABI: ampnet_cls_head_fwd_f32 / _bwd_f32 = ClassificationWithAttention.forward (pointNet/model/pointnetAtt.py:115-151)
// and its autograd backward; SURVEY row f4.
//
//   attn_output, weights = MultiheadAttention(gl, gl, gl, key_padding_mask)          [W, B, 256], [B, W, W] (mean over the heads)
//   x = relu(conv_1(attn_output.view(-1, W, 256)))          conv_1 = Conv1d(num_w -> 1, 1): a weighted sum over the W "channels"
//   out = fc_3(relu(bn_2(fc_2(x.view(-1, 256)))))           BatchNorm1d over the B rows
// The .view(-1, W, 256) of the sequence-first [W, B, 256] tensor is a re-interpretation of memory, not a transpose: row (b', w') of
// the view is flat row f = b' W + w' of attn_output, i.e. token (w = f / B, b = f % B).  Restated literally (rowmap below).
// The reference's dropout_1 is constructed and never applied; the only dropout is the attention's.
// Launch sequence: pw_gemm 256 -> 768 (in_proj) -> attention_core -> pw_gemm 256 -> 256 (out_proj) -> cls_mix -> sgemm (fc_2)
// -> cls_bn_act (batch statistics over B rows, running update) -> cls_out (fc_3); everything after the attention is [B, <= 256].
#include "bwd_misc.h"
#include "head.h"

namespace ampnet {
namespace {

// ORDER = params.CLS_HEAD_PARAMS of the Python package = state_dict order of the reference module
enum ClsParam { CP_INPROJ_W = 0, CP_INPROJ_B, CP_OUTPROJ_W, CP_OUTPROJ_B, CP_CONV1_W, CP_CONV1_B, CP_FC2_W, CP_FC2_B, CP_FC3_W, CP_FC3_B,
                CP_BN2_W, CP_BN2_B, CP_COUNT };
constexpr int CLS_HID = 128, CLS_MAX_CLASSES = 16;

struct ClsWs {
    float *qkv, *probs, *ctx, *o;      // [Q,768] [B,8,W,W] [Q,256] [Q,256]
    float *x, *z, *act;                // [B,256] relu(conv_1), [B,128] fc_2 + bias (pre-BN), [B,128] relu(bn_2)
    float *scale, *shift, *mean, *invstd;   // bn_2 [128]
    int *tok_off;                      // {0, Q}
    size_t bytes;
};
struct ClsBwdWs {
    float *da, *dy, *dx, *d_o, *d_ctx, *d_qkv, *slot_ab, *cwpart;   // [B,128] [B,128] [B,256] [Q,256] [Q,256] [Q,768] [128,2] [B, W + 1]
    size_t bytes;
};
struct Carver {
    char *base;
    size_t off = 0;
    template <typename T>
    T *take(size_t n)
    {
        off = align_up(off, 256);
        T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};
void carve(int B, int W, void *base, ClsWs &w)
{
    Carver c{reinterpret_cast<char *>(base)};
    const size_t Q = (size_t)B * W;
    w.qkv = c.take<float>(Q * 768);
    w.probs = c.take<float>((size_t)B * HEAD_HEADS * W * W);
    w.ctx = c.take<float>(Q * 256);
    w.o = c.take<float>(Q * 256);
    w.x = c.take<float>((size_t)B * 256);
    w.z = c.take<float>((size_t)B * CLS_HID);
    w.act = c.take<float>((size_t)B * CLS_HID);
    w.scale = c.take<float>(CLS_HID);
    w.shift = c.take<float>(CLS_HID);
    w.mean = c.take<float>(CLS_HID);
    w.invstd = c.take<float>(CLS_HID);
    w.tok_off = c.take<int>(2);
    w.bytes = align_up(c.off, 256);
}
void carve_bwd(int B, int W, void *base, ClsBwdWs &w)
{
    Carver c{reinterpret_cast<char *>(base)};
    const size_t Q = (size_t)B * W;
    w.da = c.take<float>((size_t)B * CLS_HID);
    w.dy = c.take<float>((size_t)B * CLS_HID);
    w.dx = c.take<float>((size_t)B * 256);
    w.d_o = c.take<float>(Q * 256);
    w.d_ctx = c.take<float>(Q * 256);
    w.d_qkv = c.take<float>(Q * 768);
    w.slot_ab = c.take<float>(2 * CLS_HID);
    w.cwpart = c.take<float>((size_t)B * (W + 1));
    w.bytes = align_up(c.off, 256);
}

__device__ __forceinline__ int rowmap(int b, int w, int B, int W)     // row of o (= token b_src * W + w_src) behind view row (b, w)
{
    const int f = b * W + w;
    return (f % B) * W + f / B;
}

// x[b][e] = relu(sum_w cw[w] o[rowmap(b, w)][e] + cb)
__global__ __launch_bounds__(256) void cls_mix_kernel(const float *__restrict__ o, const float *__restrict__ cw, const float *__restrict__ cb,
                                                     int B, int W, float *__restrict__ x)
{
    const int b = blockIdx.x, e = threadIdx.x;
    float acc = cb[0];
    for (int w = 0; w < W; ++w) acc = fmaf(cw[w], o[(size_t)rowmap(b, w, B, W) * 256 + e], acc);
    x[(size_t)b * 256 + e] = fmaxf(acc, 0.f);
}

// averaged attention weights [B, W, W]: mean over the heads of the (dropped, in train mode) probabilities -- need_weights=True
__global__ void cls_weights_kernel(const float *__restrict__ probs, int B, int W, float drop_p, uint32_t drop_base_, float *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * W * W) return;
    const int b = i / (W * W), ij = i % (W * W);
    const uint32_t thr = drop_threshold(drop_p);
    const float dscale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    float s = 0.f;
    for (int hd = 0; hd < HEAD_HEADS; ++hd) {
        const size_t o = ((size_t)(b * HEAD_HEADS + hd) * W * W) + ij;
        float p = probs[o];
        if (drop_p > 0.f) p = (mix32((uint32_t)o ^ drop_base_) >= thr) ? p * dscale : 0.f;
        s += p;
    }
    out[i] = s * (1.0f / HEAD_HEADS);
}

// z = y + b2; train: batch statistics over the B rows (double), running update, act = relu(bn(z)); eval: running statistics
__global__ __launch_bounds__(64) void cls_bn_act_kernel(float *__restrict__ z, const float *__restrict__ b2, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, float *__restrict__ rmean, float *__restrict__ rvar, int B,
                                                       int train, float *__restrict__ scale, float *__restrict__ shift, float *__restrict__ mean,
                                                       float *__restrict__ invstd, float *__restrict__ act)
{
    const int c = blockIdx.x * 64 + threadIdx.x;
    const float bias = b2[c];
    double m = 0.0, var = 0.0;
    if (train) {
        double s = 0.0;
        for (int b = 0; b < B; ++b) s += (double)(z[(size_t)b * CLS_HID + c] + bias);
        m = s / B;
        double q = 0.0;
        for (int b = 0; b < B; ++b) {
            const double d = (double)(z[(size_t)b * CLS_HID + c] + bias) - m;
            q += d * d;
        }
        var = q / B;
        rmean[c] = 0.9f * rmean[c] + 0.1f * (float)m;
        rvar[c] = 0.9f * rvar[c] + 0.1f * (float)(B > 1 ? q / (B - 1) : q);
    } else {
        m = rmean[c];
        var = rvar[c];
    }
    const float is = (float)(1.0 / sqrt(var + 1e-5)), sc = gamma[c] * is, sh = beta[c] - (float)m * sc;
    scale[c] = sc; shift[c] = sh; mean[c] = (float)m; invstd[c] = is;
    for (int b = 0; b < B; ++b) {
        const float v = z[(size_t)b * CLS_HID + c] + bias;
        z[(size_t)b * CLS_HID + c] = v;
        act[(size_t)b * CLS_HID + c] = fmaxf(fmaf(v, sc, sh), 0.f);
    }
}

// out[b][k] = act[b] . W3[k] + b3[k]
__global__ __launch_bounds__(64) void cls_out_kernel(const float *__restrict__ act, const float *__restrict__ W3, const float *__restrict__ b3, int C,
                                                    float *__restrict__ out)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const float a0 = act[(size_t)b * CLS_HID + lane], a1 = act[(size_t)b * CLS_HID + 64 + lane];
    for (int k = 0; k < C; ++k) {
        float v = a0 * W3[(size_t)k * CLS_HID + lane] + a1 * W3[(size_t)k * CLS_HID + 64 + lane];
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) out[(size_t)b * C + k] = v + b3[k];
    }
}

// conv_1 backward: dpre = dx * [x > 0]; d_o[rowmap(b, w)][e] = cw[w] dpre[b][e]; per-sample partials of d cw[w], d cb
__global__ __launch_bounds__(256) void cls_mix_bwd_kernel(const float *__restrict__ dx, const float *__restrict__ x, const float *__restrict__ o,
                                                         const float *__restrict__ cw, int B, int W, float *__restrict__ d_o,
                                                         float *__restrict__ cwpart)
{
    __shared__ float red[4];
    const int b = blockIdx.x, e = threadIdx.x, lane = e & 63, wave = e >> 6;
    const float dpre = x[(size_t)b * 256 + e] > 0.f ? dx[(size_t)b * 256 + e] : 0.f;
    for (int w = 0; w <= W; ++w) {
        float v = dpre;                                        // w == W: the bias
        if (w < W) {
            const size_t row = (size_t)rowmap(b, w, B, W) * 256 + e;
            v = dpre * o[row];
            d_o[row] = cw[w] * dpre;
        }
        for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s);
        if (lane == 0) red[wave] = v;
        __syncthreads();
        if (e == 0) cwpart[(size_t)b * (W + 1) + w] = (red[0] + red[1]) + (red[2] + red[3]);
        __syncthreads();
    }
}

}  // namespace
}  // namespace ampnet

using namespace ampnet;

#define TRY(x)                            \
    do {                                  \
        int rc_ = (x);                    \
        if (rc_ != AMPNET_OK) return rc_; \
    } while (0)

extern "C" size_t ampnet_cls_head_workspace_bytes(int B, int W)
{
    if (B < 1 || W < 1) return 0;
    ClsWs w;
    carve(B, W, nullptr, w);
    return w.bytes;
}

extern "C" int ampnet_cls_head_fwd_f32(const float *const *params_host, float *const *buffers_host, const float *gl, const uint8_t *key_pad_mask,
                                       int B, int W, int n_classes, int train, float drop_p, uint32_t seed, float *out, float *attn_weights,
                                       void *workspace, size_t workspace_bytes, void *stream)
{
    AMPNET_REQUIRE(params_host && buffers_host && gl && out && workspace, "ampnet_cls_head_fwd_f32: null pointer");
    AMPNET_REQUIRE(B >= 1 && W >= 1 && W <= HEAD_MAX_W, "ampnet_cls_head_fwd_f32: B=%d W=%d (W <= %d)", B, W, HEAD_MAX_W);
    AMPNET_REQUIRE(n_classes >= 1 && n_classes <= CLS_MAX_CLASSES, "ampnet_cls_head_fwd_f32: n_classes=%d", n_classes);
    AMPNET_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "ampnet_cls_head_fwd_f32: dropout p=%f", drop_p);
    AMPNET_REQUIRE(!(train && sync_bn_on()), "ampnet_cls_head_fwd_f32: global-batch BatchNorm (ampnet_set_collective) does not cover this head");
    hipStream_t st = (hipStream_t)stream;
    ClsWs ws;
    carve(B, W, workspace, ws);
    if (ws.bytes > workspace_bytes) return fail(AMPNET_E_WORKSPACE, "ampnet_cls_head_fwd_f32: workspace %zu B < %zu B", workspace_bytes, ws.bytes);
    const float *const *P = params_host;
    const int Q = B * W;
    const float dp = train ? drop_p : 0.f;
    auto tok_gemm = [&](const float *A, const float *Wm, const float *bias, int cout, float *Z) {
        PwGemm g;
        g.A = A; g.lda = 256; g.cin = 256;
        g.W = Wm; g.ldw = 256; g.bias = bias;
        g.Z = Z; g.ldz = cout; g.cout = cout;
        g.uniform_rows = Q; g.Q = 1; g.chunk_rows = 128; g.chunks = cdiv(Q, 128); g.rows_hint = Q;
        return pw_gemm(g, st);
    };
    TRY(tok_gemm(gl, P[CP_INPROJ_W], P[CP_INPROJ_B], 768, ws.qkv));
    TRY(attention_core(ws.qkv, key_pad_mask, ws.probs, ws.ctx, B, W, dp, drop_base(seed, 0), st));
    TRY(tok_gemm(ws.ctx, P[CP_OUTPROJ_W], P[CP_OUTPROJ_B], 256, ws.o));
    if (attn_weights) {
        hipLaunchKernelGGL(cls_weights_kernel, dim3(cdiv(B * W * W, 256)), dim3(256), 0, st, ws.probs, B, W, dp, drop_base(seed, 0), attn_weights);
        TRY(check_launch("cls_weights_kernel"));
    }
    hipLaunchKernelGGL(cls_mix_kernel, dim3(B), dim3(256), 0, st, ws.o, P[CP_CONV1_W], P[CP_CONV1_B], B, W, ws.x);
    TRY(check_launch("cls_mix_kernel"));
    TRY(sgemm_small(0, 1, B, CLS_HID, 256, ws.x, 256, P[CP_FC2_W], 256, ws.z, CLS_HID, 0, st));
    hipLaunchKernelGGL(cls_bn_act_kernel, dim3(CLS_HID / 64), dim3(64), 0, st, ws.z, P[CP_FC2_B], P[CP_BN2_W], P[CP_BN2_B], buffers_host[0],
                       buffers_host[1], B, train, ws.scale, ws.shift, ws.mean, ws.invstd, ws.act);
    TRY(check_launch("cls_bn_act_kernel"));
    hipLaunchKernelGGL(cls_out_kernel, dim3(B), dim3(64), 0, st, ws.act, P[CP_FC3_W], P[CP_FC3_B], n_classes, out);
    return check_launch("cls_out_kernel");
}

extern "C" size_t ampnet_cls_head_bwd_workspace_bytes(int B, int W)
{
    if (B < 1 || W < 1) return 0;
    ClsBwdWs w;
    carve_bwd(B, W, nullptr, w);
    return w.bytes;
}

extern "C" int ampnet_cls_head_bwd_f32(const float *const *params_host, float *const *grads_host, const float *gl, int B, int W, int n_classes,
                                       float drop_p, uint32_t seed, const float *d_out, float *d_gl, void *fwd_workspace,
                                       size_t fwd_workspace_bytes, void *bwd_workspace, size_t bwd_workspace_bytes, void *stream)
{
    AMPNET_REQUIRE(params_host && grads_host && gl && d_out && d_gl && fwd_workspace && bwd_workspace, "ampnet_cls_head_bwd_f32: null pointer");
    AMPNET_REQUIRE(B >= 1 && W >= 1 && W <= HEAD_MAX_W && n_classes >= 1 && n_classes <= CLS_MAX_CLASSES, "ampnet_cls_head_bwd_f32: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    ClsWs f;
    carve(B, W, fwd_workspace, f);
    ClsBwdWs b;
    carve_bwd(B, W, bwd_workspace, b);
    if (f.bytes > fwd_workspace_bytes) return fail(AMPNET_E_WORKSPACE, "ampnet_cls_head_bwd_f32: forward workspace %zu B < %zu B", fwd_workspace_bytes, f.bytes);
    if (b.bytes > bwd_workspace_bytes) return fail(AMPNET_E_WORKSPACE, "ampnet_cls_head_bwd_f32: backward workspace %zu B < %zu B", bwd_workspace_bytes, b.bytes);
    const float *const *P = params_host;
    float *const *G = grads_host;
    const int Q = B * W, C = n_classes;
    // fc_3
    TRY(sgemm_linear_bwd(B, C, CLS_HID, d_out, C, f.act, CLS_HID, P[CP_FC3_W], CLS_HID, G[CP_FC3_W], CLS_HID, b.da, CLS_HID, st));
    TRY(colsum(d_out, B, C, G[CP_FC3_B], st));
    // bn_2 + ReLU over the B rows (global batch under ampnet_set_collective), then fc_2
    TRY(fc_bn_bwd(b.da, f.z, f.scale, f.shift, f.mean, f.invstd, 1, B, CLS_HID, b.dy, b.slot_ab, st));
    {
        BnGradItem it = {b.slot_ab, G[CP_BN2_W], G[CP_BN2_B], CLS_HID, 1};
        TRY(bn_param_grads(&it, 1, st));
    }
    TRY(sgemm_linear_bwd(B, CLS_HID, 256, b.dy, CLS_HID, f.x, 256, P[CP_FC2_W], 256, G[CP_FC2_W], 256, b.dx, 256, st));
    TRY(colsum(b.dy, B, CLS_HID, G[CP_FC2_B], st));
    // conv_1 over the re-viewed attention output
    hipLaunchKernelGGL(cls_mix_bwd_kernel, dim3(B), dim3(256), 0, st, b.dx, f.x, f.o, P[CP_CONV1_W], B, W, b.d_o, b.cwpart);
    TRY(check_launch("cls_mix_bwd_kernel"));
    TRY(colsum(b.cwpart, B, W + 1, b.dx, st));                          // dx is free now: [0 .. W) = d conv_1.weight, [W] = d conv_1.bias
    if (hipMemcpyAsync(G[CP_CONV1_W], b.dx, (size_t)W * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess ||
        hipMemcpyAsync(G[CP_CONV1_B], b.dx + W, sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return fail(AMPNET_E_LAUNCH, "ampnet_cls_head_bwd_f32: copy of the conv_1 gradients failed");
    // out_proj, attention, in_proj
    TRY(sgemm_linear_bwd(Q, 256, 256, b.d_o, 256, f.ctx, 256, P[CP_OUTPROJ_W], 256, G[CP_OUTPROJ_W], 256, b.d_ctx, 256, st));
    TRY(colsum(b.d_o, Q, 256, G[CP_OUTPROJ_B], st));
    TRY(attention_core_bwd(f.qkv, f.probs, b.d_ctx, b.d_qkv, B, W, drop_p, drop_base(seed, 0), st));
    TRY(sgemm_linear_bwd(Q, 768, 256, b.d_qkv, 768, gl, 256, P[CP_INPROJ_W], 256, G[CP_INPROJ_W], 256, d_gl, 256, st));
    TRY(colsum(b.d_qkv, Q, 768, G[CP_INPROJ_B], st));
    return AMPNET_OK;
}
