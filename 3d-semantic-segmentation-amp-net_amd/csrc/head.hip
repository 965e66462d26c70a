// head.hip -- C ABI: ampnet_head_fwd_f32 = SegmentationWithAttention.forward (pointNet/model/pointnetAtt.py:176-209)
// plus the loss/argmax of train_pointnet-attention.py:445-450.
//
// What the reference materialises and this path does not: the per-cluster repeat + cat of the 256-d attention
// token over every point (:192-201, a [B, 320, P] tensor = 755 MB at B = 32).  conv_2 over cat(local, token) is
//     W[:, :64] . local[point]  +  (W[:, 64:] . token[window] + bias)
// so the token part is a per-WINDOW bias vector [Q, 128] computed once by a small GEMM and added in the
// epilogue of the 64 -> 128 per-point GEMM.  Launch sequence:
//   posenc_tokens -> pw_gemm 256->768 (in_proj) -> attention_core -> pw_gemm 256->256 (out_proj)
//   -> pw_gemm 256->128 (token half of conv_2) -> pw_gemm 64->128 (+per-window bias, bn_2 stats)
//   -> pw_gemm 128->64 (bn_2+ReLU+dropout prologue, bn_3 stats) -> pw_gemm 64->C (bn_3+ReLU+dropout prologue)
//   -> head_logits (transposed store, CE, argmax)
#include "head.h"

namespace ampnet {

HeadShape head_shape(int B, int W, int R, int max_rows, int n_classes, int train, int kind)
{
    HeadShape s;
    s.kind = kind;
    s.B = B;
    s.W = W;
    s.Q = B * W;
    s.R = R;
    s.max_rows = max_rows;
    s.train = train;
    s.n_classes = n_classes;
    s.chunk_rows = 512;
    s.chunks = cdiv(max_rows, s.chunk_rows);
    s.tok_chunk_rows = 128;
    s.tok_chunks = cdiv(s.Q, s.tok_chunk_rows);
    s.logit_B = B;
    return s;
}

namespace {
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
void carve_bn(Carver &c, BnSlot1 &b, int C)
{
    b.C = C;
    b.scale = c.take<float>(C);
    b.shift = c.take<float>(C);
    b.mean = c.take<float>(C);
    b.invstd = c.take<float>(C);
    b.smean = c.take<float>(C);
    b.suvar = c.take<float>(C);
}
}  // namespace

void head_carve(const HeadShape &s, void *base, HeadWs &ws)
{
    Carver c{reinterpret_cast<char *>(base)};
    const size_t Q = (size_t)s.Q, R = (size_t)s.R;
    const bool gru = s.kind == HEAD_KIND_GRU;
    ws.tok = c.take<float>(gru ? 0 : Q * 256);
    ws.pe_hid = c.take<float>(gru ? 0 : Q * 16);
    ws.pe_slope = c.take<float>(gru ? 0 : Q * 16);
    ws.qkv = c.take<float>(Q * (gru ? 3 * GRU_H : 768));
    ws.probs = c.take<float>(gru ? 0 : (size_t)s.B * HEAD_HEADS * s.W * s.W);
    ws.ctx = c.take<float>(Q * 256);
    ws.g2 = c.take<float>(Q * (gru ? GRU_H : 256));
    ws.gbias = c.take<float>(Q * 128);
    ws.z2 = c.take<float>(R * 128);
    ws.z3 = c.take<float>(R * 64);
    ws.part_rows = c.take<int>(1024);
    const size_t np = (Q * (size_t)s.chunks > (size_t)s.tok_chunks ? Q * (size_t)s.chunks : (size_t)s.tok_chunks) * 128;
    ws.part_sum = c.take<float>(np);
    ws.part_sq = c.take<float>(np);
    ws.merge = c.take<float>(bn_finalize_merge_floats(1, 128));
    ws.loss_part = c.take<float>((size_t)cdiv(s.R, 128) * 2);
    ws.z4 = c.take<float>(R * HEAD_MAX_CLASSES);
    carve_bn(c, ws.bn2, 128);
    carve_bn(c, ws.bn3, 64);
    ws.bytes = align_up(c.off, 256);
}

}  // namespace ampnet

using namespace ampnet;

#define TRY(x)                            \
    do {                                  \
        int rc_ = (x);                    \
        if (rc_ != AMPNET_OK) return rc_; \
    } while (0)

namespace ampnet {

int head_points_fwd(const HeadShape &s, HeadWs &ws, const HeadPointParams &p, const float *lo, const int32_t *win_off, float dp,
                    uint32_t seed, const HeadLossArgs &lo_, hipStream_t st)
{
    const bool tr = s.train != 0;
    const int Q = s.Q, total_rows = s.R, max_rows = s.max_rows, n_classes = s.n_classes;
    const int zb = z_storage_bf16() ? 1 : 0;         // z2 / z3 are stored as bf16 (precision mode 3)
    if (!tr) {
        BnFoldItem items[2] = {{p.bn2_w, p.bn2_b, p.bn2_mean, p.bn2_var, ws.bn2.scale, ws.bn2.shift, 128},
                               {p.bn3_w, p.bn3_b, p.bn3_mean, p.bn3_var, ws.bn3.scale, ws.bn3.shift, 64}};
        TRY(bn_fold(items, 2, 1e-5f, st));
    }
    auto finalize = [&](BnSlot1 &b, const float *gamma, const float *beta) {
        BnFinalize f;
        f.part_sum = ws.part_sum; f.part_sq = ws.part_sq; f.chunk_rows = s.chunk_rows;
        f.win_off = win_off; f.n_slots = 1; f.C = b.C;
        f.part_rows = ws.part_rows; f.Q = pw_gemm_stat_plan(Q, s.chunks, 1).parts; f.chunks = 1;      // one partial per workgroup
        f.gamma = gamma; f.beta = beta;
        f.scale = b.scale; f.shift = b.shift; f.mean = b.mean; f.invstd = b.invstd; f.stat_mean = b.smean; f.stat_uvar = b.suvar;
        f.merge_ws = ws.merge;
        return bn_finalize(f, st);
    };
    {   // conv_2: local half + per-window token bias
        PwGemm g;
        g.A = lo; g.lda = 64; g.cin = 64;
        g.W = p.conv2_w; g.ldw = p.conv2_ld;
        g.bias = ws.gbias; g.bias_win_stride = 128;
        g.Z = ws.z2; g.ldz = 128; g.cout = 128; g.z_bf16 = zb;
        if (tr) { g.part_sum = ws.part_sum; g.part_sq = ws.part_sq; g.part_rows = ws.part_rows; g.stat_lanes = pw_gemm_stat_plan(Q, s.chunks, 1).lanes; }
        g.win_off = win_off; g.Q = Q; g.chunk_rows = s.chunk_rows; g.chunks = s.chunks; g.rows_hint = total_rows;
        TRY(pw_gemm(g, st));
        if (tr) TRY(finalize(ws.bn2, p.bn2_w, p.bn2_b));
    }
    {   // conv_3 on dropout(relu(bn_2(z2)))
        PwGemm g;
        g.A = ws.z2; g.lda = 128; g.cin = 128; g.a_bf16 = zb;
        g.W = p.conv3_w; g.ldw = 128; g.bias = p.conv3_b;
        g.pro_scale = ws.bn2.scale; g.pro_shift = ws.bn2.shift;
        g.drop_p = dp; g.drop_seed = drop_base(seed, 1);
        g.Z = ws.z3; g.ldz = 64; g.cout = 64; g.z_bf16 = zb;
        if (tr) { g.part_sum = ws.part_sum; g.part_sq = ws.part_sq; g.part_rows = ws.part_rows; g.stat_lanes = pw_gemm_stat_plan(Q, s.chunks, 1).lanes; }
        g.win_off = win_off; g.Q = Q; g.chunk_rows = s.chunk_rows; g.chunks = s.chunks; g.rows_hint = total_rows;
        TRY(pw_gemm(g, st));
        if (tr) TRY(finalize(ws.bn3, p.bn3_w, p.bn3_b));
    }
    {
        HeadOut o;
        o.z3 = ws.z3; o.scale = ws.bn3.scale; o.shift = ws.bn3.shift;
        o.W = p.conv4_w; o.bias = p.conv4_b;
        o.drop_p = dp; o.drop_seed = drop_base(seed, 2);
        o.R = total_rows; o.P = total_rows / s.logit_B; o.C = n_classes;
        o.logits = lo_.logits; o.targets = lo_.targets; o.class_w = lo_.class_w; o.preds = lo_.preds;
        o.loss_part = lo_.loss_out ? ws.loss_part : nullptr;
        int blocks = 0;
        // conv_4 on the matrix cores (the 5 output columns ride in one 32-column MFMA tile), then the row-wise tail
        PwGemm g;
        g.A = ws.z3; g.lda = 64; g.cin = 64; g.a_bf16 = zb;
        g.W = p.conv4_w; g.ldw = 64; g.bias = p.conv4_b;
        g.pro_scale = ws.bn3.scale; g.pro_shift = ws.bn3.shift;
        g.drop_p = dp; g.drop_seed = drop_base(seed, 2);
        g.Z = ws.z4; g.ldz = HEAD_MAX_CLASSES; g.cout = n_classes;
        g.win_off = win_off; g.Q = Q; g.chunk_rows = s.chunk_rows; g.chunks = s.chunks; g.rows_hint = total_rows;
        TRY(pw_gemm(g, st));
        TRY(head_logits(o, ws.z4, HEAD_MAX_CLASSES, &blocks, st));
        if (lo_.loss_out) TRY(loss_finalize(ws.loss_part, blocks, lo_.loss_out, st));
    }
    if (tr) {
        BnRunItem items[2] = {{ws.bn2.smean, ws.bn2.suvar, p.bn2_mean, p.bn2_var, 128, 1},
                              {ws.bn3.smean, ws.bn3.suvar, p.bn3_mean, p.bn3_var, 64, 1}};
        TRY(bn_running_update(items, 2, 0.1f, st));
    }
    return AMPNET_OK;
}

}  // namespace ampnet

extern "C" size_t ampnet_head_workspace_bytes(int B, int W, int total_rows, int max_rows, int n_classes, int train)
{
    if (B < 1 || W < 1 || total_rows < 1 || max_rows < 1) return 0;
    HeadWs ws;
    head_carve(head_shape(B, W, total_rows, max_rows, n_classes, train), nullptr, ws);
    return ws.bytes;
}

// logit_B: how many equal parts the rows are cut into for the [logit_B, C, P] logits / [logit_B, P] predictions: B for the module's
// contract (every sample has the same number of points), 1 for the several-files-per-launch inference form (samples of unequal size)
static int head_fwd_impl(const float *const *params_host, float *const *buffers_host, const float *gl,
                         const float *lo, const float *centroids, const int32_t *win_off,
                         const uint8_t *key_pad_mask, int B, int W, int total_rows, int max_rows, int n_classes,
                         int train, float drop_p, uint32_t seed, float *logits, const long long *targets,
                         const float *class_w, long long *preds, float *loss_out, void *workspace,
                         size_t workspace_bytes, void *stream, int logit_B)
{
    AMPNET_REQUIRE(params_host && buffers_host && gl && lo && centroids && win_off && logits && workspace, "ampnet_head_fwd_f32: null pointer");
    AMPNET_REQUIRE(B >= 1 && W >= 1 && W <= HEAD_MAX_W, "ampnet_head_fwd_f32: B=%d W=%d (W <= %d)", B, W, HEAD_MAX_W);
    AMPNET_REQUIRE(total_rows >= 1 && total_rows % logit_B == 0, "ampnet_head_fwd_f32: total_rows %d not a multiple of B %d", total_rows, logit_B);
    AMPNET_REQUIRE(n_classes >= 1 && n_classes <= HEAD_MAX_CLASSES, "ampnet_head_fwd_f32: n_classes=%d", n_classes);
    AMPNET_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "ampnet_head_fwd_f32: dropout p=%f", drop_p);
    AMPNET_REQUIRE(!loss_out || targets, "ampnet_head_fwd_f32: loss_out needs targets");
    hipStream_t st = (hipStream_t)stream;
    HeadShape s = head_shape(B, W, total_rows, max_rows, n_classes, train);
    s.logit_B = logit_B;
    HeadWs ws;
    head_carve(s, workspace, ws);
    if (ws.bytes > workspace_bytes) return fail(AMPNET_E_WORKSPACE, "ampnet_head_fwd_f32: workspace %zu B < %zu B", workspace_bytes, ws.bytes);
    const float *const *P = params_host;
    const bool tr = train != 0;
    const float dp = tr ? drop_p : 0.f;
    const int Q = s.Q;
    if (tr) ws_tag_set(workspace, matrix_precision());

    TRY(posenc_tokens(gl, centroids, P[HP_FC1_W], P[HP_FC1_B], P[HP_FC2_W], P[HP_FC2_B], ws.tok, Q, st, s.train ? ws.pe_hid : nullptr, s.train ? ws.pe_slope : nullptr));
    auto tok_gemm = [&](const float *A, const float *Wm, int ldw, const float *bias, int cout, float *Z) {
        PwGemm g;
        g.A = A; g.lda = 256; g.cin = 256;
        g.W = Wm; g.ldw = ldw; g.bias = bias;
        g.Z = Z; g.ldz = cout; g.cout = cout;
        g.uniform_rows = Q; g.Q = 1; g.chunk_rows = s.tok_chunk_rows; g.chunks = s.tok_chunks; g.rows_hint = Q;
        return pw_gemm(g, st);
    };
    TRY(tok_gemm(ws.tok, P[HP_INPROJ_W], 256, P[HP_INPROJ_B], 768, ws.qkv));
    TRY(attention_core(ws.qkv, key_pad_mask, ws.probs, ws.ctx, B, W, dp, drop_base(seed, 0), st));
    TRY(tok_gemm(ws.ctx, P[HP_OUTPROJ_W], 256, P[HP_OUTPROJ_B], 256, ws.g2));
    TRY(tok_gemm(ws.g2, P[HP_CONV2_W] + 64, 320, P[HP_CONV2_B], 128, ws.gbias));   // token half of conv_2 + its bias

    HeadPointParams pp;
    pp.conv2_w = P[HP_CONV2_W]; pp.conv2_ld = 320;
    pp.conv3_w = P[HP_CONV3_W]; pp.conv3_b = P[HP_CONV3_B]; pp.conv4_w = P[HP_CONV4_W]; pp.conv4_b = P[HP_CONV4_B];
    pp.bn2_w = P[HP_BN2_W]; pp.bn2_b = P[HP_BN2_B]; pp.bn3_w = P[HP_BN3_W]; pp.bn3_b = P[HP_BN3_B];
    pp.bn2_mean = buffers_host[HB_BN2_MEAN]; pp.bn2_var = buffers_host[HB_BN2_VAR];
    pp.bn3_mean = buffers_host[HB_BN3_MEAN]; pp.bn3_var = buffers_host[HB_BN3_VAR];
    HeadLossArgs lo_args;
    lo_args.logits = logits; lo_args.targets = targets; lo_args.class_w = class_w; lo_args.preds = preds; lo_args.loss_out = loss_out;
    return head_points_fwd(s, ws, pp, lo, win_off, dp, seed, lo_args, st);
}

extern "C" int ampnet_head_fwd_f32(const float *const *params_host, float *const *buffers_host, const float *gl,
                                   const float *lo, const float *centroids, const int32_t *win_off,
                                   const uint8_t *key_pad_mask, int B, int W, int total_rows, int max_rows, int n_classes,
                                   int train, float drop_p, uint32_t seed, float *logits, const long long *targets,
                                   const float *class_w, long long *preds, float *loss_out, void *workspace,
                                   size_t workspace_bytes, void *stream)
{
    return head_fwd_impl(params_host, buffers_host, gl, lo, centroids, win_off, key_pad_mask, B, W, total_rows, max_rows, n_classes, train, drop_p,
                         seed, logits, targets, class_w, preds, loss_out, workspace, workspace_bytes, stream, B);
}

// Inference over SEVERAL FILES in one launch sequence (test_pointnet_att_segmen.py:127-181 runs one file per step, batch 1): file f owns the W
// window slots f * W .. f * W + W - 1, of which its real clusters come first; the unused slots are windows of ZERO rows
// (win_off repeats) whose key_pad_mask entry is 1, so the attention of a file sees exactly its own clusters.  Files have different point
// counts, hence logits [n_classes, total_rows] and preds [total_rows] are over the concatenated rows (row order = file, cluster, point).
extern "C" int ampnet_head_fwd_files_f32(const float *const *params_host, float *const *buffers_host, const float *gl, const float *lo,
                                         const float *centroids, const int32_t *win_off, const uint8_t *key_pad_mask, int n_files, int W,
                                         int total_rows, int max_rows, int n_classes, float *logits, long long *preds, void *workspace,
                                         size_t workspace_bytes, void *stream)
{
    AMPNET_REQUIRE(key_pad_mask, "ampnet_head_fwd_files_f32: the slot mask is required");
    return head_fwd_impl(params_host, buffers_host, gl, lo, centroids, win_off, key_pad_mask, n_files, W, total_rows, max_rows, n_classes, 0, 0.f, 0,
                         logits, nullptr, nullptr, preds, nullptr, workspace, workspace_bytes, stream, 1);
}
