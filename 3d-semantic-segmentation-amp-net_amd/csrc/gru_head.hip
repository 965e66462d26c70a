// gru_head.hip -- C ABI: ampnet_gru_head_fwd_f32 / _bwd_f32 = SegmentationWithGRU.forward (pointNet/model/pointnetAtt.py:212-258)
// and its autograd backward; SURVEY row f4 (the GRU variant of the sequence model, pointNet/rnn/train_pointnetGRU.py:335-441).
//
// The reference runs nn.GRU(256 -> 64, batch_first) over the W window tokens of a sample, repeats every hidden state over the points
// of its window, concatenates with the 64 local features and applies conv_2 / bn_2 / conv_3 / bn_3 / conv_4 -- the same five lines as
// the attention head (:244-248 vs :203-207).  Here, as in head.hip, the repeat + cat is never built:
//     conv_2(cat(local, h)) = W2[:, :64] . local[point] + (W2[:, 64:] . h[window] + b2)
// so the GRU only has to produce one 128-float bias row per window (gbias) and head_points_fwd / head_points_bwd do the rest.
// Launch sequence (forward): pw_gemm 256 -> 192 (W_ih x + b_ih for all B*W tokens) -> gru_seq_kernel (the W dependent steps, one wave
// per sample, lane = hidden unit) -> pw_gemm 64 -> 128 (token half of conv_2) -> head_points_fwd.
#include "bwd_misc.h"
#include "head.h"

namespace ampnet {
namespace {

// ORDER = params.GRU_HEAD_PARAMS of the Python package = state_dict order of the reference module
enum GruParam { GP_WIH = 0, GP_WHH, GP_BIH, GP_BHH, GP_CONV2_W, GP_CONV2_B, GP_CONV3_W, GP_CONV3_B, GP_CONV4_W, GP_CONV4_B,
                GP_BN2_W, GP_BN2_B, GP_BN3_W, GP_BN3_B, GP_COUNT };
constexpr int G3 = 3 * GRU_H;

__device__ inline float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// torch.nn.GRU cell (gate order r, z, n):  r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r * gh_n), h' = (1 - z) n + z h
// with gi = W_ih x + b_ih (precomputed for every step) and gh = W_hh h + b_hh.  One wave per sample, lane j owns hidden unit j;
// W_hh^T sits in LDS as [k][gate unit] so that lane j's three weights of step k are conflict-free reads, h_k comes by readlane.
__global__ __launch_bounds__(64) void gru_seq_kernel(const float *__restrict__ gi, const float *__restrict__ Whh, const float *__restrict__ bhh,
                                                    float *__restrict__ h_out, float *__restrict__ gates, int W)
{
    extern __shared__ float sW[];                      // [64][G3 + 1]
    const int b = blockIdx.x, j = threadIdx.x;
    for (int e = j; e < G3 * GRU_H; e += 64) sW[(e & 63) * (G3 + 1) + (e >> 6)] = Whh[e];
    const float br = bhh[j], bz = bhh[GRU_H + j], bn = bhh[2 * GRU_H + j];
    __syncthreads();
    float h = 0.f;                                     // initHidden: zeros (:253-256)
    for (int t = 0; t < W; ++t) {
        const size_t q = (size_t)b * W + t;
        const float gr = gi[q * G3 + j], gz = gi[q * G3 + GRU_H + j], gn = gi[q * G3 + 2 * GRU_H + j];
        float ar = br, az = bz, an = bn;
#pragma unroll
        for (int k = 0; k < GRU_H; ++k) {
            const float hk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h), k));
            const float *w = sW + k * (G3 + 1);
            ar = fmaf(w[j], hk, ar);
            az = fmaf(w[GRU_H + j], hk, az);
            an = fmaf(w[2 * GRU_H + j], hk, an);
        }
        const float r = sigmoidf_(gr + ar), z = sigmoidf_(gz + az), n = tanhf(fmaf(r, an, gn));
        h = fmaf(z, h - n, n);                         // (1 - z) n + z h
        h_out[q * GRU_H + j] = h;
        if (gates) {
            float *g = gates + q * 4 * GRU_H;
            g[j] = r;
            g[GRU_H + j] = z;
            g[2 * GRU_H + j] = n;
            g[3 * GRU_H + j] = an;
        }
    }
}

// back-propagation through time of the cell above: dH [Q, 64] is the gradient every h_t receives from conv_2;
// writes d(gi) [Q, 192], d(gh) [Q, 192] (what W_hh h + b_hh receives) and h_{t-1} [Q, 64] for the weight-gradient GEMMs
__global__ __launch_bounds__(64) void gru_seq_bwd_kernel(const float *__restrict__ dH, const float *__restrict__ gates, const float *__restrict__ h_all,
                                                        const float *__restrict__ Whh, float *__restrict__ dgi, float *__restrict__ dgh,
                                                        float *__restrict__ hprev_out, int W)
{
    extern __shared__ float sW[];                      // [G3][64] row-major: lane j reads W_hh[k][j]
    const int b = blockIdx.x, j = threadIdx.x;
    for (int e = j; e < G3 * GRU_H; e += 64) sW[e] = Whh[e];
    __syncthreads();
    float dh_next = 0.f;
    for (int t = W - 1; t >= 0; --t) {
        const size_t q = (size_t)b * W + t;
        const float *g = gates + q * 4 * GRU_H;
        const float r = g[j], z = g[GRU_H + j], n = g[2 * GRU_H + j], ghn = g[3 * GRU_H + j];
        const float hp = t > 0 ? h_all[(q - 1) * GRU_H + j] : 0.f;
        const float dh = dH[q * GRU_H + j] + dh_next;
        const float dnp = dh * (1.0f - z) * (1.0f - n * n);
        const float dzp = dh * (hp - n) * z * (1.0f - z);
        const float drp = dnp * ghn * r * (1.0f - r);
        const float dnr = dnp * r;
        dgi[q * G3 + j] = drp;
        dgi[q * G3 + GRU_H + j] = dzp;
        dgi[q * G3 + 2 * GRU_H + j] = dnp;
        dgh[q * G3 + j] = drp;
        dgh[q * G3 + GRU_H + j] = dzp;
        dgh[q * G3 + 2 * GRU_H + j] = dnr;
        hprev_out[q * GRU_H + j] = hp;
        float acc = dh * z;
#pragma unroll
        for (int k = 0; k < GRU_H; ++k) {
            const float a0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(drp), k));
            const float a1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dzp), k));
            const float a2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dnr), k));
            acc = fmaf(a0, sW[k * GRU_H + j], acc);
            acc = fmaf(a1, sW[(GRU_H + k) * GRU_H + j], acc);
            acc = fmaf(a2, sW[(2 * GRU_H + k) * GRU_H + j], acc);
        }
        dh_next = acc;
    }
}

HeadPointParams point_params(const float *const *P, float *const *bufs)
{
    HeadPointParams pp;
    pp.conv2_w = P[GP_CONV2_W]; pp.conv2_ld = 64 + GRU_H;
    pp.conv3_w = P[GP_CONV3_W]; pp.conv3_b = P[GP_CONV3_B]; pp.conv4_w = P[GP_CONV4_W]; pp.conv4_b = P[GP_CONV4_B];
    pp.bn2_w = P[GP_BN2_W]; pp.bn2_b = P[GP_BN2_B]; pp.bn3_w = P[GP_BN3_W]; pp.bn3_b = P[GP_BN3_B];
    if (bufs) {
        pp.bn2_mean = bufs[HB_BN2_MEAN]; pp.bn2_var = bufs[HB_BN2_VAR];
        pp.bn3_mean = bufs[HB_BN3_MEAN]; pp.bn3_var = bufs[HB_BN3_VAR];
    }
    return pp;
}

}  // namespace
}  // namespace ampnet

using namespace ampnet;

#define TRY(x)                            \
    do {                                  \
        int rc_ = (x);                    \
        if (rc_ != AMPNET_OK) return rc_; \
    } while (0)

extern "C" size_t ampnet_gru_head_workspace_bytes(int B, int W, int total_rows, int max_rows, int n_classes, int train)
{
    if (B < 1 || W < 1 || total_rows < 1 || max_rows < 1) return 0;
    HeadWs ws;
    head_carve(head_shape(B, W, total_rows, max_rows, n_classes, train, HEAD_KIND_GRU), nullptr, ws);
    return ws.bytes;
}

extern "C" int ampnet_gru_head_fwd_f32(const float *const *params_host, float *const *buffers_host, const float *gl, const float *lo,
                                       const int32_t *win_off, int B, int W, int total_rows, int max_rows, int n_classes, int train,
                                       float drop_p, uint32_t seed, float *logits, const long long *targets, const float *class_w,
                                       long long *preds, float *loss_out, void *workspace, size_t workspace_bytes, void *stream)
{
    AMPNET_REQUIRE(params_host && buffers_host && gl && lo && win_off && logits && workspace, "ampnet_gru_head_fwd_f32: null pointer");
    AMPNET_REQUIRE(B >= 1 && W >= 1, "ampnet_gru_head_fwd_f32: B=%d W=%d", B, W);
    AMPNET_REQUIRE(total_rows >= 1 && total_rows % B == 0, "ampnet_gru_head_fwd_f32: total_rows %d not a multiple of B %d", total_rows, B);
    AMPNET_REQUIRE(n_classes >= 1 && n_classes <= HEAD_MAX_CLASSES, "ampnet_gru_head_fwd_f32: n_classes=%d", n_classes);
    AMPNET_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "ampnet_gru_head_fwd_f32: dropout p=%f", drop_p);
    AMPNET_REQUIRE(!loss_out || targets, "ampnet_gru_head_fwd_f32: loss_out needs targets");
    hipStream_t st = (hipStream_t)stream;
    const HeadShape s = head_shape(B, W, total_rows, max_rows, n_classes, train, HEAD_KIND_GRU);
    HeadWs ws;
    head_carve(s, workspace, ws);
    if (ws.bytes > workspace_bytes) return fail(AMPNET_E_WORKSPACE, "ampnet_gru_head_fwd_f32: workspace %zu B < %zu B", workspace_bytes, ws.bytes);
    const float *const *P = params_host;
    const bool tr = train != 0;
    const int Q = s.Q;
    if (tr) ws_tag_set(workspace, matrix_precision());

    auto tok_gemm = [&](const float *A, int cin, const float *Wm, int ldw, const float *bias, int cout, float *Z) {
        PwGemm g;
        g.A = A; g.lda = cin; g.cin = cin;
        g.W = Wm; g.ldw = ldw; g.bias = bias;
        g.Z = Z; g.ldz = cout; g.cout = cout;
        g.uniform_rows = Q; g.Q = 1; g.chunk_rows = s.tok_chunk_rows; g.chunks = s.tok_chunks; g.rows_hint = Q;
        return pw_gemm(g, st);
    };
    TRY(tok_gemm(gl, 256, P[GP_WIH], 256, P[GP_BIH], G3, ws.qkv));                              // gi = W_ih x + b_ih, all steps at once
    hipLaunchKernelGGL(gru_seq_kernel, dim3(B), dim3(64), (size_t)GRU_H * (G3 + 1) * sizeof(float), st, ws.qkv, P[GP_WHH], P[GP_BHH], ws.g2,
                       tr ? ws.ctx : nullptr, W);
    TRY(check_launch("gru_seq_kernel"));
    TRY(tok_gemm(ws.g2, GRU_H, P[GP_CONV2_W] + 64, 64 + GRU_H, P[GP_CONV2_B], 128, ws.gbias));   // hidden-state half of conv_2 + its bias

    HeadLossArgs o;
    o.logits = logits; o.targets = targets; o.class_w = class_w; o.preds = preds; o.loss_out = loss_out;
    return head_points_fwd(s, ws, point_params(P, buffers_host), lo, win_off, tr ? drop_p : 0.f, seed, o, st);
}

extern "C" size_t ampnet_gru_head_bwd_workspace_bytes(int B, int W, int total_rows, int max_rows, int n_classes)
{
    if (B < 1 || W < 1 || total_rows < 1 || max_rows < 1) return 0;
    HeadBwdWs w;
    head_bwd_carve(head_shape(B, W, total_rows, max_rows, n_classes, 1, HEAD_KIND_GRU), nullptr, w);
    return w.bytes;
}

extern "C" int ampnet_gru_head_bwd_f32(const float *const *params_host, float *const *grads_host, const float *gl, const float *lo,
                                       const int32_t *win_off, int B, int W, int total_rows, int max_rows, int n_classes, float drop_p,
                                       uint32_t seed, const float *dlogits, float *d_lo, float *d_gl, void *fwd_workspace,
                                       size_t fwd_workspace_bytes, void *bwd_workspace, size_t bwd_workspace_bytes, void *stream)
{
    AMPNET_REQUIRE(params_host && grads_host && gl && lo && win_off && dlogits && d_lo && d_gl && fwd_workspace && bwd_workspace,
                   "ampnet_gru_head_bwd_f32: null pointer");
    AMPNET_REQUIRE(B >= 1 && W >= 1 && total_rows % B == 0, "ampnet_gru_head_bwd_f32: bad sizes");
    AMPNET_REQUIRE(n_classes >= 1 && n_classes <= HEAD_MAX_CLASSES, "ampnet_gru_head_bwd_f32: n_classes=%d", n_classes);
    TRY(ws_tag_check(fwd_workspace, "ampnet_gru_head_bwd_f32"));
    hipStream_t st = (hipStream_t)stream;
    const HeadShape s = head_shape(B, W, total_rows, max_rows, n_classes, 1, HEAD_KIND_GRU);
    HeadWs f;
    head_carve(s, fwd_workspace, f);
    HeadBwdWs b;
    head_bwd_carve(s, bwd_workspace, b);
    if (f.bytes > fwd_workspace_bytes) return fail(AMPNET_E_WORKSPACE, "ampnet_gru_head_bwd_f32: forward workspace %zu B < %zu B", fwd_workspace_bytes, f.bytes);
    if (b.bytes > bwd_workspace_bytes) return fail(AMPNET_E_WORKSPACE, "ampnet_gru_head_bwd_f32: backward workspace %zu B < %zu B", bwd_workspace_bytes, b.bytes);
    const float *const *P = params_host;
    float *const *G = grads_host;
    const int Q = s.Q;

    HeadPointGrads pg;
    pg.conv2_w = G[GP_CONV2_W]; pg.conv2_ld = 64 + GRU_H; pg.conv2_b = G[GP_CONV2_B];
    pg.conv3_w = G[GP_CONV3_W]; pg.conv3_b = G[GP_CONV3_B]; pg.conv4_w = G[GP_CONV4_W]; pg.conv4_b = G[GP_CONV4_B];
    pg.bn2_w = G[GP_BN2_W]; pg.bn2_b = G[GP_BN2_B]; pg.bn3_w = G[GP_BN3_W]; pg.bn3_b = G[GP_BN3_B];
    TRY(head_points_bwd(s, f, b, point_params(P, nullptr), pg, lo, win_off, drop_p, seed, dlogits, d_lo, st));
    // gbias = h W2[:, 64:]^T + b2: weight gradient of the hidden-state half and dL/dh of every step
    TRY(sgemm_linear_bwd(Q, 128, GRU_H, b.dgb, 128, f.g2, GRU_H, P[GP_CONV2_W] + 64, 64 + GRU_H, G[GP_CONV2_W] + 64, 64 + GRU_H, b.d_g2, GRU_H, st));
    hipLaunchKernelGGL(gru_seq_bwd_kernel, dim3(B), dim3(64), (size_t)G3 * GRU_H * sizeof(float), st, b.d_g2, f.ctx, f.g2, P[GP_WHH], b.d_qkv,
                       b.d_ctx, b.hid, W);
    TRY(check_launch("gru_seq_bwd_kernel"));
    // gi = x W_ih^T + b_ih: dW_ih = d(gi)^T x, dx = d(gi) W_ih;  gh = h_{t-1} W_hh^T + b_hh: dW_hh = d(gh)^T h_{t-1}
    TRY(sgemm_linear_bwd(Q, G3, 256, b.d_qkv, G3, gl, 256, P[GP_WIH], 256, G[GP_WIH], 256, d_gl, 256, st));
    TRY(colsum(b.d_qkv, Q, G3, G[GP_BIH], st));
    TRY(sgemm_small(1, 0, G3, GRU_H, Q, b.d_ctx, G3, b.hid, GRU_H, G[GP_WHH], GRU_H, 0, st));
    TRY(colsum(b.d_ctx, Q, G3, G[GP_BHH], st));
    return AMPNET_OK;
}
