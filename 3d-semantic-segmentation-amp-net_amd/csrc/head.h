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

struct HeadShape {
    int B, W, Q, R, max_rows, train, n_classes;
    int chunk_rows, chunks;        // point layers
    int tok_chunk_rows, tok_chunks;   // token GEMMs: one window of Q rows
};

struct BnSlot1 {
    float *scale, *shift, *mean, *invstd, *smean, *suvar;
    int C;
};

struct HeadWs {
    float *tok, *qkv, *probs, *ctx, *g2, *gbias;   // [Q,256] [Q,768] [B,8,W,W] [Q,256] [Q,256] [Q,128]
    float *z2, *z3;                                // [R,128] [R,64]
    int *tok_off;                                  // [2] = {0, Q}
    float *part_sum, *part_sq;                     // [max(Q*chunks, tok_chunks), 128]
    float *merge;                                  // two-stage bn_finalize scratch
    float *loss_part;                              // [blocks, 2]
    float *z4;                                     // [R, 8] conv_4 output (row-major, padded)
    BnSlot1 bn2, bn3;
    size_t bytes;
};

HeadShape head_shape(int B, int W, int R, int max_rows, int n_classes, int train);
void head_carve(const HeadShape &s, void *base, HeadWs &ws);

// tok[q, :] = gl[q, :] + fc2(leaky_relu(fc1(centroids[q, :])))     (pointnetAtt.py:183-185)
int posenc_tokens(const float *gl, const float *cent, const float *w1, const float *b1, const float *w2, const float *b2,
                  float *tok, int Q, hipStream_t st);
// softmax(q k^T / sqrt(d) + mask) [dropout] v per (sample, head)      (nn.MultiheadAttention core)
int attention_core(const float *qkv, const uint8_t *key_pad_mask, float *probs, float *ctx, int B, int W, float drop_p,
                   uint32_t drop_base, hipStream_t st);
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
