// head.h -- parameter tables and workspace layout shared by the head forward and backward.
#pragma once
#include "kernels.h"

namespace ampnet {

// ORDER = params.HEAD_PARAMS / HEAD_BUFFERS of the Python package
enum HeadParam { HP_FC1_W = 0, HP_FC1_B, HP_FC2_W, HP_FC2_B, HP_INPROJ_W, HP_INPROJ_B, HP_OUTPROJ_W, HP_OUTPROJ_B,
                 HP_CONV2_W, HP_CONV2_B, HP_CONV3_W, HP_CONV3_B, HP_CONV4_W, HP_CONV4_B, HP_BN2_W, HP_BN2_B,
                 HP_BN3_W, HP_BN3_B, HP_COUNT };
enum { HB_BN2_MEAN = 0, HB_BN2_VAR, HB_BN3_MEAN, HB_BN3_VAR, HB_COUNT };

constexpr int HEAD_E = 256, HEAD_HEADS = 8, HEAD_D = 32, HEAD_MAX_W = 32, HEAD_MAX_CLASSES = 8;

enum { HEAD_KIND_ATTENTION = 0, HEAD_KIND_GRU = 1 };
constexpr int GRU_H = 64;          // HIDDEN_SIZE (pointNet/rnn/train_pointnetGRU.py:28)

struct HeadShape {
    int B, W, Q, R, max_rows, train, n_classes;
    int kind;                      // which token producer feeds conv_2: attention (256-d token) or GRU (64-d hidden state)
    int chunk_rows, chunks;        // point layers
    int tok_chunk_rows, tok_chunks;   // token GEMMs: one window of Q rows
    int logit_B;                   // the logits are [logit_B, C, R / logit_B]: B, or 1 for ampnet_head_fwd_files_f32
};

struct BnSlot1 {
    float *scale, *shift, *mean, *invstd, *smean, *suvar;
    int C;
};

struct HeadWs {
    // attention: tok [Q,256] qkv [Q,768] probs [B,8,W,W] ctx [Q,256] g2 [Q,256] (out_proj output = the token)
    // GRU:       qkv = gi [Q,192] (W_ih x + b_ih), ctx = gates [Q,4,64] (r, z, n, W_hn h + b_hn), g2 = h [Q,64]; tok / probs unused
    float *pe_hid, *pe_slope;                      // [Q,16] hidden layer of the positional encoding and its leaky-ReLU slope (attention, for the backward)
    float *tok, *qkv, *probs, *ctx, *g2, *gbias;   // gbias [Q,128]: token half of conv_2 + its bias, one row per window
    float *z2, *z3;                                // [R,128] [R,64]
    int *part_rows;                                // [1024] rows per per-workgroup statistics partial (PwGemm.part_rows)
    float *part_sum, *part_sq;                     // [max(Q*chunks, tok_chunks), 128]
    float *merge;                                  // two-stage bn_finalize scratch
    float *loss_part;                              // [blocks, 2]
    float *z4;                                     // [R, 8] conv_4 output (row-major, padded)
    BnSlot1 bn2, bn3;
    size_t bytes;
};

HeadShape head_shape(int B, int W, int R, int max_rows, int n_classes, int train, int kind = HEAD_KIND_ATTENTION);
void head_carve(const HeadShape &s, void *base, HeadWs &ws);

// ---- the per-point layers both heads share: conv_2 (local half + per-window token bias) -> bn_2 -> ReLU -> dropout -> conv_3 -> bn_3
// -> ReLU -> dropout -> conv_4 (pointnetAtt.py:203-207 and :244-248 are the same five lines) -------------------------------------
struct HeadPointParams {
    const float *conv2_w = nullptr;      // [128, conv2_ld]: columns 0..63 multiply the local features, the rest the token
    int conv2_ld = 0;
    const float *conv3_w = nullptr, *conv3_b = nullptr, *conv4_w = nullptr, *conv4_b = nullptr;
    const float *bn2_w = nullptr, *bn2_b = nullptr, *bn3_w = nullptr, *bn3_b = nullptr;
    float *bn2_mean = nullptr, *bn2_var = nullptr, *bn3_mean = nullptr, *bn3_var = nullptr;   // running statistics (forward only)
};
struct HeadPointGrads {
    float *conv2_w = nullptr;            // [128, conv2_ld]: only the local half (columns 0..63) is written here
    int conv2_ld = 0;
    float *conv2_b = nullptr, *conv3_w = nullptr, *conv3_b = nullptr, *conv4_w = nullptr, *conv4_b = nullptr;
    float *bn2_w = nullptr, *bn2_b = nullptr, *bn3_w = nullptr, *bn3_b = nullptr;
};
struct HeadLossArgs {
    float *logits = nullptr;             // [B, C, P]
    const long long *targets = nullptr;
    const float *class_w = nullptr;
    long long *preds = nullptr;
    float *loss_out = nullptr;
};
// needs ws.gbias filled by the token stage
int head_points_fwd(const HeadShape &s, HeadWs &ws, const HeadPointParams &p, const float *lo, const int32_t *win_off, float drop_p,
                    uint32_t seed, const HeadLossArgs &o, hipStream_t st);

struct HeadBwdWs {
    float *dy3, *dy2;              // [R,64] [R,128]
    float *wpart, *dbpart, *dgb;   // [Q * wchunks, 128*128], [Q * wchunks, 128], [Q, 128] gradient of the per-window token bias
    float *w4part;                 // [blocks, C*64 + C]
    float *part_a, *part_b;        // [max(Q*chunks, blocks), 128]
    float *P1[2], *P2[2], *P3[2], *slot_ab[2];   // bn2 (128), bn3 (64)
    // attention: d_g2 [Q,256] d_ctx [Q,256] d_qkv [Q,768] hid / slope / d_hid [Q,16]
    // GRU:       d_g2 = dL/dh from conv_2 [Q,64], d_qkv = d(gi) [Q,192], d_ctx = d(gh) [Q,192], hid = h_{t-1} [Q,64]
    float *d_g2, *d_ctx, *d_qkv, *hid, *slope, *d_hid;
    int *tot_off;                  // {0, R}
    size_t bytes;
};
void head_bwd_carve(const HeadShape &s, void *base, HeadBwdWs &w);
// reverse of head_points_fwd: writes d_lo [R,64], every gradient in g, and leaves b.dgb = dL/d(gbias) [Q,128] for the token stage
int head_points_bwd(const HeadShape &s, HeadWs &f, HeadBwdWs &b, const HeadPointParams &p, const HeadPointGrads &g, const float *lo,
                    const int32_t *win_off, float drop_p, uint32_t seed, const float *dlogits, float *d_lo, hipStream_t st);

// tok[q, :] = gl[q, :] + fc2(leaky_relu(fc1(centroids[q, :])))     (pointnetAtt.py:183-185)
int posenc_tokens(const float *gl, const float *cent, const float *w1, const float *b1, const float *w2, const float *b2,
                  float *tok, int Q, hipStream_t st, float *hid_out = nullptr, float *slope_out = nullptr);
// softmax(q k^T / sqrt(d) + mask) [dropout] v per (sample, head)      (nn.MultiheadAttention core)
int attention_core(const float *qkv, const uint8_t *key_pad_mask, float *probs, float *ctx, int B, int W, float drop_p,
                   uint32_t drop_base, hipStream_t st);
// gradients of the above wrt q, k, v given d(ctx): dqkv [Q, 768]
int attention_core_bwd(const float *qkv, const float *probs, const float *dctx, float *dqkv, int B, int W, float drop_p, uint32_t drop_base,
                       hipStream_t st);
// logits[b, c, p] = conv_4(dropout(relu(bn_3(z3))))[row = b * P + p]; optional weighted CE partials + argmax
struct HeadOut {
    const float *z3 = nullptr;             // [R, 64]
    const float *scale = nullptr, *shift = nullptr;   // bn_3 affine [64]
    const float *W = nullptr, *bias = nullptr;        // [C, 64], [C]
    float drop_p = 0.f;
    uint32_t drop_seed = 0;
    int R = 0, P = 0, C = 0;
    float *logits = nullptr;               // [B, C, P]
    const long long *targets = nullptr;    // [B, P] int64, -1 = ignore; may be nullptr
    const float *class_w = nullptr;        // [C]
    long long *preds = nullptr;            // [B, P] int64 or nullptr
    float *loss_part = nullptr;            // [blocks, 2]: sum w * nll, sum w
};
// conv_4 runs as a pw_gemm (z4 [R, ldz4]); this is its tail: logits / preds / CE partials (HeadOut.z3, W, bias unused)
int head_logits(const HeadOut &a, const float *z4, int ldz4, int *n_blocks, hipStream_t st);
int loss_finalize(const float *loss_part, int n_blocks, float *loss_out, hipStream_t st);   // loss_out[0] = ce, [1] = sum w

}  // namespace ampnet
