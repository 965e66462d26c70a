// encoder.hip -- C ABI: ampnet_encoder_fwd_f32 = BasePointNet.forward (pointNet/model/pointnetAtt.py:80-112,
// with TransformationNet.forward :28-47 twice) for ALL windows of a step in one launch sequence.
//
// The reference runs the encoder W times per step, once per cluster slot (train_pointnet-attention.py:396-410),
// so every BatchNorm sees the B windows of one slot.  Here window q = b * W + w and slot(q) = q % n_slots: the
// statistics stay per slot, the launches cover all Q = B * W windows.  Launch sequence (train mode; eval drops
// the bn_finalize launches and folds running statistics once):
//   pw_input(K=3) -> pw_gemm 64->128 -> pw_gemm 128->256 + maxpool -> FC 256->256->128->9 (+I)          input T-Net
//   pw_input(K=12, per-window T3 folded into conv_1) -> pw_gemm 64->64                                   conv_1, conv_2
//   pw_gemm 64->64 -> 64->128 -> 128->256 + maxpool -> FC 256->256->128->4096 (+I)                       feature T-Net
//   pw_gemm 64->64 with per-window weights T64 (the torch.bmm)                                           local features
//   pw_gemm 64->64 -> 64->128 -> 128->128 -> 128->256 + maxpool                                          conv_3..6, global
// Every layer stores its PRE-BatchNorm output; BatchNorm + ReLU are applied by the consumer's prologue, the
// 256-channel pooled layers are never materialised in either mode (only one signed extreme per window chunk and channel,
// its row and its value: the backward of those layers is algebraic, encoder_bwd.hip).
#include "encoder.h"

namespace ampnet {

static const int kBnC[BN_ENC_COUNT] = {64, 128, 256, 256, 128, 64, 64, 64, 128, 256, 256, 128, 64, 128, 128, 256};

EncShape enc_shape(int Q, int n_slots, int R, int max_rows, int train)
{
    EncShape s;
    s.Q = Q;
    s.n_slots = n_slots;
    s.R = R;
    s.max_rows = max_rows;
    s.train = train;
    s.chunk_rows = 512;
    s.chunks = cdiv(max_rows, s.chunk_rows);
    s.x_chunk_rows = 128;
    s.x_chunks = cdiv(max_rows, s.x_chunk_rows);
    s.fc_rows = Q / n_slots;
    s.fc_chunk_rows = 128;
    s.fc_chunks = cdiv(s.fc_rows, s.fc_chunk_rows);
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
}  // namespace

void enc_carve(const EncShape &s, void *base, EncWs &ws)
{
    Carver c{reinterpret_cast<char *>(base)};
    const size_t R = (size_t)s.R, Q = (size_t)s.Q;
    if (s.train) {
        // the 256-channel pooled layers are not stored in train mode either: their backward is algebraic (encoder_bwd.hip)
        ws.z_t1 = c.take<float>(R * 64);
        ws.z_t2 = c.take<float>(R * 128);
        ws.z_t3 = nullptr;
        ws.z_c1 = c.take<float>(R * 64);
        ws.z_c2 = c.take<float>(R * 64);
        ws.z_f1 = c.take<float>(R * 64);
        ws.z_f2 = c.take<float>(R * 128);
        ws.z_f3 = nullptr;
        ws.z_c3 = c.take<float>(R * 64);
        ws.z_c4 = c.take<float>(R * 128);
        ws.z_c5 = c.take<float>(R * 128);
        ws.z_c6 = nullptr;
    } else {
        // eval: four rotating buffers; the 256-channel layers are never stored
        float *a = c.take<float>(R * 64), *b = c.take<float>(R * 128), *cc = c.take<float>(R * 64), *d = c.take<float>(R * 128);
        ws.z_t1 = a; ws.z_t2 = b; ws.z_t3 = nullptr;
        ws.z_c1 = a; ws.z_c2 = cc;
        ws.z_f1 = a; ws.z_f2 = b; ws.z_f3 = nullptr;
        ws.z_c3 = a; ws.z_c4 = b; ws.z_c5 = d; ws.z_c6 = nullptr;
    }
    ws.pool_t = c.take<float>(Q * 256);
    ws.z_tf1 = c.take<float>(Q * 256);
    ws.z_tf2 = c.take<float>(Q * 128);
    ws.T3 = c.take<float>(Q * 12);
    ws.pool_f = c.take<float>(Q * 256);
    ws.z_ff1 = c.take<float>(Q * 256);
    ws.z_ff2 = c.take<float>(Q * 128);
    ws.arg_t = c.take<int>(Q * 256);
    ws.arg_f = c.take<int>(Q * 256);
    ws.arg_c = c.take<int>(Q * 256);
    ws.zext_t = c.take<float>(Q * 256);
    ws.zext_f = c.take<float>(Q * 256);
    ws.zext_c = c.take<float>(Q * 256);
    ws.part_rows = c.take<int>(2 * enc_fwd_part_region_rows(s));
    ws.merge = c.take<float>(bn_finalize_merge_floats(s.n_slots, 256));     // two-stage bn_finalize scratch
    const int pchunks = s.x_chunks > s.chunks ? s.x_chunks : s.chunks;        // sized for either kernel family (the precision mode may change between calls)
    const size_t np = Q * (size_t)(pchunks > s.fc_chunks ? pchunks : s.fc_chunks) * 256;
    // the fused backward indexes its BatchNorm sums by workgroup (<= 256 + n_slots of them) + one row per window
    const size_t two_regions = 2 * enc_bwd_part_region_floats(s);
    size_t np_bwd = np + (size_t)(320 + s.n_slots) * 256 > two_regions ? np + (size_t)(320 + s.n_slots) * 256 : two_regions;
    // the forward alternates between two regions of per-workgroup statistics partials (a consumer merges its producer's while it writes its own)
    if (np_bwd < 2 * enc_fwd_part_region_floats(s)) np_bwd = 2 * enc_fwd_part_region_floats(s);
    ws.part_sum = c.take<float>(np_bwd);
    ws.part_sq = c.take<float>(np_bwd);
    ws.part_max = c.take<float>(np);
    ws.part_amax = c.take<int>(np);
    for (int i = 0; i < BN_ENC_COUNT; ++i) {
        const size_t n = (size_t)s.n_slots * kBnC[i];
        ws.bn[i].C = kBnC[i];
        ws.bn[i].scale = c.take<float>(n);
        ws.bn[i].shift = c.take<float>(n);
        ws.bn[i].mean = c.take<float>(n);
        ws.bn[i].invstd = c.take<float>(n);
        ws.bn[i].smean = c.take<float>(n);
        ws.bn[i].suvar = c.take<float>(n);
    }
    ws.bytes = align_up(c.off, 256);
}

namespace {

struct BnParamRef {
    const float *gamma, *beta;
    float *rmean, *rvar;
};

struct EncRun {
    hipStream_t st;
    EncShape s;
    EncWs ws;
    const float *const *P;
    float *const *Bf;
    const int *win_off;
    BnParamRef bnp[BN_ENC_COUNT];
    bool zb = false;           // precision mode 3: the z tensors of the workspace are bf16
    // BatchNorm statistics in flight: the producer of layer `bn` left `parts` per-workgroup partials in region `region` of part_sum / part_sq /
    // part_rows; the next point layer merges them itself (PwGemm.pfin_*) or settle() launches bn_finalize.  Producers alternate regions.
    struct Pending {
        int bn = -1, parts = 0, region = 0;
    };
    mutable Pending pend;
    mutable int region = 0;
    float *psum(int r) const { return ws.part_sum + (size_t)r * enc_fwd_part_region_floats(s); }
    float *psq(int r) const { return ws.part_sq + (size_t)r * enc_fwd_part_region_floats(s); }
    int *prows(int r) const { return ws.part_rows + (size_t)r * enc_fwd_part_region_rows(s); }
    static bool split_layer(int cin, int cout) { return pw_gemm_stat_lane_cap(cin, cout) != 512; }     // this layer runs on the split kernels
    bool consumer_fin() const
    {
        static const bool off = [] { const char *v = getenv("AMPNET_FWD_FIN_IN_KERNEL"); return v && v[0] == '0'; }();
        return s.train && !off && !sync_bn_on();
    }
    void note_stats(int bn, int lanes) const
    {
        pend.bn = bn;
        pend.parts = cdiv(lanes, s.n_slots) * s.n_slots;
        pend.region = region;
        region ^= 1;
    }

    void bind_bn()
    {
        auto tn = [&](int base_p, int base_b, int first) {
            for (int i = 0; i < 5; ++i) {
                bnp[first + i] = {P[base_p + TP_BN1_W + 2 * i], P[base_p + TP_BN1_B + 2 * i], Bf[base_b + 2 * i], Bf[base_b + 2 * i + 1]};
            }
        };
        tn(EP_IT, EB_IT, BN_T1);
        tn(EP_FT, EB_FT, BN_F1);
        const int main_ids[6] = {BN_C1, BN_C2, BN_C3, BN_C4, BN_C5, BN_C6};
        for (int i = 0; i < 6; ++i)
            bnp[main_ids[i]] = {P[EP_BN1_W + 2 * i], P[EP_BN1_B + 2 * i], Bf[EB_MAIN + 2 * i], Bf[EB_MAIN + 2 * i + 1]};
    }

    // point layer on the real windows
    // a_is_z / z_is_z: the operand is one of the workspace's pre-BatchNorm tensors (stored as bf16 in precision mode 3), not `local`
    PwGemm point_layer(const float *A, int cin, const float *W, int cout, int pro_bn, float *Z, bool stats, bool pool, int pool_bn = -1,
                       bool a_is_z = true, bool z_is_z = true) const
    {
        PwGemm g;
        g.A = A; g.lda = cin; g.cin = cin;
        g.a_bf16 = (zb && a_is_z) ? 1 : 0;
        g.z_bf16 = (zb && z_is_z && Z) ? 1 : 0;
        g.W = W; g.ldw = cin;
        if (pro_bn >= 0) { g.pro_scale = ws.bn[pro_bn].scale; g.pro_shift = ws.bn[pro_bn].shift; }
        g.n_slots = s.train ? s.n_slots : 1;
        g.Z = Z; g.ldz = cout; g.cout = cout;
        if (stats) { g.part_sum = psum(region); g.part_sq = psq(region); }
        // eval forward in fp32: only the extremes are tracked (no argmax rows: nothing reads them without a backward)
        const bool rows_too = s.train || !precision_is_f32();
        if (pool) { g.part_max = ws.part_max; g.part_amax = rows_too ? ws.part_amax : nullptr; g.pool_gamma = bnp[pool_bn].gamma; }
        g.win_off = win_off; g.Q = s.Q; g.chunk_rows = s.chunk_rows; g.chunks = s.chunks; g.rows_hint = s.R;
        if (split_layer(cin, cout)) { g.chunk_rows = s.x_chunk_rows; g.chunks = s.x_chunks; }     // one wave per block of rows (pw_gemm.hip)
        if (stats) {                                             // one partial per workgroup: bn_finalize in one stage (kernels.h)
            g.part_rows = prows(region);
            g.stat_lanes = pw_gemm_stat_plan(s.Q, g.chunks, g.n_slots, pw_gemm_stat_lane_cap(cin, cout)).lanes;
        }
        if (stats && pro_bn >= 0 && pend.bn == pro_bn && consumer_fin() && (cin == 64 || cin == 128)) {
            // this launch finishes its input's BatchNorm itself (kernels.h: pfin_*)
            const BnSlot &b = ws.bn[pro_bn];
            g.pfin_sum = psum(pend.region); g.pfin_sq = psq(pend.region); g.pfin_rows = prows(pend.region); g.pfin_parts = pend.parts;
            g.pfin_gamma = bnp[pro_bn].gamma; g.pfin_beta = bnp[pro_bn].beta;
            g.pfin_scale = b.scale; g.pfin_shift = b.shift; g.pfin_mean = b.mean; g.pfin_invstd = b.invstd; g.pfin_smean = b.smean; g.pfin_suvar = b.suvar;
        }
        return g;
    }
    // launch a point layer: what is pending and not consumed by it is finalized first; its own statistics become the pending ones
    int run_point(const PwGemm &g, int stats_bn) const
    {
        if (g.pfin_sum) pend.bn = -1;
        else if (int rc = settle(); rc != AMPNET_OK) return rc;
        if (int rc = pw_gemm(g, st); rc != AMPNET_OK) return rc;
        if (stats_bn >= 0 && s.train) note_stats(stats_bn, g.stat_lanes);
        return AMPNET_OK;
    }
    // the pending statistics through a bn_finalize launch (consumers that do not merge them themselves: the pool, eval-free paths)
    int settle() const
    {
        if (pend.bn < 0 || !s.train) return AMPNET_OK;
        const int bn = pend.bn;
        pend.bn = -1;
        BnFinalize f;
        f.part_sum = psum(pend.region); f.part_sq = psq(pend.region); f.part_rows = prows(pend.region);
        f.win_off = win_off;
        f.Q = pend.parts; f.chunks = 1; f.chunk_rows = s.chunk_rows; f.uniform_rows = 0;
        f.n_slots = s.n_slots; f.C = ws.bn[bn].C;
        f.gamma = bnp[bn].gamma; f.beta = bnp[bn].beta;
        f.scale = ws.bn[bn].scale; f.shift = ws.bn[bn].shift; f.mean = ws.bn[bn].mean; f.invstd = ws.bn[bn].invstd;
        f.stat_mean = ws.bn[bn].smean; f.stat_uvar = ws.bn[bn].suvar;
        f.merge_ws = ws.merge;
        return bn_finalize(f, st);
    }
    // FC layer on the pooled rows: n_slots windows of Q / n_slots rows (one window of Q rows in eval mode)
    // stats_bn >= 0: this layer's BatchNorm.  A slot's rows are one block of rows when fc_chunks == 1 (B <= 128): the workgroup then writes
    // the BatchNorm constants itself and finalize_fc() launches nothing.  Not under global-batch BatchNorm (the ranks' partials are merged).
    bool fc_direct() const { return s.train && s.fc_chunks == 1 && !sync_bn_on(); }
    PwGemm fc_layer(const float *A, int cin, const float *W, int cout, const float *bias, int pro_bn, float *Z, int ldz, int stats_bn) const
    {
        PwGemm g;
        g.A = A; g.lda = cin; g.cin = cin;
        g.W = W; g.ldw = cin; g.bias = bias;
        if (pro_bn >= 0) { g.pro_scale = ws.bn[pro_bn].scale; g.pro_shift = ws.bn[pro_bn].shift; }
        g.n_slots = s.train ? s.n_slots : 1;
        g.Z = Z; g.ldz = ldz; g.cout = cout;
        g.uniform_rows = s.fc_rows; g.Q = s.n_slots; g.chunk_rows = s.fc_chunk_rows; g.chunks = s.fc_chunks; g.rows_hint = s.Q;
        if (stats_bn >= 0 && s.train) {
            g.part_sum = ws.part_sum; g.part_sq = ws.part_sq; g.part_rows = ws.part_rows;
            g.stat_lanes = pw_gemm_stat_plan(g.Q, g.chunks, g.n_slots).lanes;
            if (fc_direct()) {
                const BnSlot &b = ws.bn[stats_bn];
                g.fin_gamma = bnp[stats_bn].gamma; g.fin_beta = bnp[stats_bn].beta;
                g.fin_scale = b.scale; g.fin_shift = b.shift; g.fin_mean = b.mean; g.fin_invstd = b.invstd; g.fin_smean = b.smean; g.fin_suvar = b.suvar;
            }
        }
        return g;
    }
    // BatchNorm constants of an FC layer whose slot is more than one block of rows (B > 128); otherwise the GEMM wrote them itself
    int finalize_fc(int bn) const
    {
        if (!s.train || fc_direct()) return AMPNET_OK;
        BnFinalize f;
        f.part_sum = ws.part_sum; f.part_sq = ws.part_sq; f.part_rows = ws.part_rows;
        f.win_off = win_off;
        f.Q = cdiv(pw_gemm_stat_plan(s.n_slots, s.fc_chunks, s.n_slots).lanes, s.n_slots) * s.n_slots;    // partial slots (kernels.h: PwStatPlan.parts)
        f.chunks = 1; f.chunk_rows = s.fc_chunk_rows; f.uniform_rows = 0;
        f.n_slots = s.n_slots; f.C = ws.bn[bn].C;
        f.gamma = bnp[bn].gamma; f.beta = bnp[bn].beta;
        f.scale = ws.bn[bn].scale; f.shift = ws.bn[bn].shift; f.mean = ws.bn[bn].mean; f.invstd = ws.bn[bn].invstd;
        f.stat_mean = ws.bn[bn].smean; f.stat_uvar = ws.bn[bn].suvar;
        f.merge_ws = ws.merge;
        return bn_finalize(f, st);
    }
    int pool(int bn, float *pooled, int *arg, float *zext, bool slot_major) const
    {
        PoolFinalize p;
        if (pend.bn == bn && consumer_fin() && ws.bn[bn].C == 256) {             // the pooled layer's own BatchNorm, finished by this launch
            const BnSlot &bs = ws.bn[bn];
            p.pfin_sum = psum(pend.region); p.pfin_sq = psq(pend.region); p.pfin_rows = prows(pend.region); p.pfin_parts = pend.parts;
            p.pfin_gamma = bnp[bn].gamma; p.pfin_beta = bnp[bn].beta;
            p.pfin_scale = bs.scale; p.pfin_shift = bs.shift; p.pfin_mean = bs.mean; p.pfin_invstd = bs.invstd; p.pfin_smean = bs.smean; p.pfin_suvar = bs.suvar;
            pend.bn = -1;
        } else if (int rc = settle(); rc != AMPNET_OK) return rc;
        const bool rows_too = s.train || !precision_is_f32();
        p.part_max = ws.part_max; p.part_amax = rows_too ? ws.part_amax : nullptr;
        p.scale = ws.bn[bn].scale; p.shift = ws.bn[bn].shift;
        p.Q = s.Q; p.chunks = split_layer(128, 256) ? s.x_chunks : s.chunks; p.n_slots = s.train ? s.n_slots : 1; p.C = 256;
        p.out_slot_major = (slot_major && s.train) ? 1 : 0;
        p.pooled = pooled; p.arg = rows_too ? arg : nullptr; p.zext = rows_too ? zext : nullptr;
        return pool_finalize(p, st);
    }
};

#define TRY(x)                   \
    do {                         \
        int rc_ = (x);           \
        if (rc_ != AMPNET_OK) return rc_; \
    } while (0)

// one T-Net: conv stack on `A0` ([R, 64] pre-BN with prologue pro0, or the K=3 input layer), FC head -> T [Q, k*k]
int run_tnet(const EncRun &e, int pbase, int bn0, const float *x_or_A, int pro0, bool input_k3, float *z1, float *z2, float *z3,
             float *pooled, int *arg, float *zext, float *zf1, float *zf2, float *T, int k)
{
    const bool tr = e.s.train;
    if (input_k3) {
        PwInput in;
        in.x = x_or_A; in.W = e.P[pbase + TP_CONV1]; in.mode = 0; in.Z = z1; in.z_bf16 = e.zb ? 1 : 0;
        TRY(e.settle());
        if (tr) { in.part_sum = e.psum(e.region); in.part_sq = e.psq(e.region); }
        in.win_off = e.win_off; in.Q = e.s.Q; in.chunk_rows = e.s.chunk_rows; in.chunks = e.s.chunks;
        if (tr) { in.part_rows = e.prows(e.region); in.stat_lanes = pw_input_stat_lanes(e.s.Q, e.s.chunks, e.s.n_slots); in.n_slots = e.s.n_slots; }
        TRY(pw_input(in, e.st));
        if (tr) e.note_stats(bn0 + 0, in.stat_lanes);
    } else {
        TRY(e.run_point(e.point_layer(x_or_A, 64, e.P[pbase + TP_CONV1], 64, pro0, z1, tr, false), bn0 + 0));
    }
    TRY(e.run_point(e.point_layer(z1, 64, e.P[pbase + TP_CONV2], 128, bn0 + 0, z2, tr, false), bn0 + 1));
    TRY(e.run_point(e.point_layer(z2, 128, e.P[pbase + TP_CONV3], 256, bn0 + 1, z3, tr, true, bn0 + 2), bn0 + 2));
    TRY(e.pool(bn0 + 2, pooled, arg, zext, true));
    // FC head on [Q, 256]
    TRY(pw_gemm(e.fc_layer(pooled, 256, e.P[pbase + TP_FC1], 256, nullptr, -1, zf1, 256, bn0 + 3), e.st));
    TRY(e.finalize_fc(bn0 + 3));
    TRY(pw_gemm(e.fc_layer(zf1, 256, e.P[pbase + TP_FC2], 128, nullptr, bn0 + 3, zf2, 128, bn0 + 4), e.st));
    TRY(e.finalize_fc(bn0 + 4));
    {
        PwGemm g = e.fc_layer(zf2, 128, e.P[pbase + TP_FC3_W], k * k, e.P[pbase + TP_FC3_B], bn0 + 4, T, k * k, -1);
        g.identity_k = k;                                        // + I (pointnetAtt.py:42-46) in the GEMM's bias, no extra launch
        TRY(pw_gemm(g, e.st));
    }
    return AMPNET_OK;
}

}  // namespace
}  // namespace ampnet

using namespace ampnet;

extern "C" size_t ampnet_encoder_workspace_bytes(int Q, int n_slots, int total_rows, int max_rows, int train)
{
    if (Q < 1 || n_slots < 1 || total_rows < 1 || max_rows < 1) return 0;
    EncWs ws;
    enc_carve(enc_shape(Q, n_slots, total_rows, max_rows, train), nullptr, ws);
    return ws.bytes;
}

extern "C" int ampnet_encoder_fwd_f32(const float *const *params_host, float *const *buffers_host, const float *x,
                                      const int32_t *win_off, int Q, int n_slots, int total_rows, int max_rows, int train,
                                      float *local, float *global_feat, float *feat_T, float *in_T, void *workspace,
                                      size_t workspace_bytes, void *stream)
{
    AMPNET_REQUIRE(params_host && buffers_host && x && win_off && local && global_feat && feat_T && workspace, "ampnet_encoder_fwd_f32: null pointer");
    AMPNET_REQUIRE(Q >= 1 && n_slots >= 1 && total_rows >= 1 && max_rows >= 1, "ampnet_encoder_fwd_f32: bad sizes");
    AMPNET_REQUIRE(!train || Q % n_slots == 0, "ampnet_encoder_fwd_f32: train mode needs Q (%d) %% n_slots (%d) == 0", Q, n_slots);
    EncRun e;
    e.st = (hipStream_t)stream;
    e.s = enc_shape(Q, train ? n_slots : 1, total_rows, max_rows, train);
    enc_carve(e.s, workspace, e.ws);
    if (e.ws.bytes > workspace_bytes) return fail(AMPNET_E_WORKSPACE, "ampnet_encoder_fwd_f32: workspace %zu B < %zu B", workspace_bytes, e.ws.bytes);
    e.P = params_host;
    e.Bf = buffers_host;
    e.win_off = win_off;
    e.bind_bn();
    e.zb = z_storage_bf16();
    const bool tr = train != 0;
    if (tr) ws_tag_set(workspace, matrix_precision());      // the backward checks it ran in the same precision mode

    if (!tr) {
        BnFoldItem items[BN_ENC_COUNT];
        for (int i = 0; i < BN_ENC_COUNT; ++i)
            items[i] = {e.bnp[i].gamma, e.bnp[i].beta, e.bnp[i].rmean, e.bnp[i].rvar, e.ws.bn[i].scale, e.ws.bn[i].shift, e.ws.bn[i].C};
        TRY(bn_fold(items, BN_ENC_COUNT, 1e-5f, e.st));
    }

    // input T-Net on xyz
    TRY(run_tnet(e, EP_IT, BN_T1, x, -1, true, e.ws.z_t1, e.ws.z_t2, e.ws.z_t3, e.ws.pool_t, e.ws.arg_t, e.ws.zext_t, e.ws.z_tf1, e.ws.z_tf2, e.ws.T3, 3));
    // conv_1 on cat(xyz * T3, x), conv_2
    {
        PwInput in;
        in.x = x; in.W = e.P[EP_CONV1]; in.T = e.ws.T3; in.mode = 1;
        in.perwin_slot_major = tr ? 1 : 0; in.n_slots = e.s.n_slots; in.Z = e.ws.z_c1; in.z_bf16 = e.zb ? 1 : 0;
        TRY(e.settle());
        if (tr) { in.part_sum = e.psum(e.region); in.part_sq = e.psq(e.region); }
        in.win_off = win_off; in.Q = Q; in.chunk_rows = e.s.chunk_rows; in.chunks = e.s.chunks;
        if (tr) { in.part_rows = e.prows(e.region); in.stat_lanes = pw_input_stat_lanes(Q, e.s.chunks, e.s.n_slots); }
        TRY(pw_input(in, e.st));
        if (tr) e.note_stats(BN_C1, in.stat_lanes);
    }
    TRY(e.run_point(e.point_layer(e.ws.z_c1, 64, e.P[EP_CONV2], 64, BN_C1, e.ws.z_c2, tr, false), BN_C2));
    // feature T-Net on relu(bn_2(z_c2))
    TRY(run_tnet(e, EP_FT, BN_F1, e.ws.z_c2, BN_C2, false, e.ws.z_f1, e.ws.z_f2, e.ws.z_f3, e.ws.pool_f, e.ws.arg_f, e.ws.zext_f, e.ws.z_ff1, e.ws.z_ff2, feat_T, 64));
    // local = relu(bn_2(z_c2)) x T64[window]  (torch.bmm, pointnetAtt.py:96)
    {
        PwGemm g = e.point_layer(e.ws.z_c2, 64, feat_T, 64, BN_C2, local, false, false, -1, true, false);     // local leaves as fp32
        g.w_win_stride = 64 * 64;
        g.perwin_slot_major = tr ? 1 : 0;
        TRY(e.run_point(g, -1));
    }
    TRY(e.run_point(e.point_layer(local, 64, e.P[EP_CONV3], 64, -1, e.ws.z_c3, tr, false, -1, false, true), BN_C3));
    TRY(e.run_point(e.point_layer(e.ws.z_c3, 64, e.P[EP_CONV4], 128, BN_C3, e.ws.z_c4, tr, false), BN_C4));
    TRY(e.run_point(e.point_layer(e.ws.z_c4, 128, e.P[EP_CONV5], 128, BN_C4, e.ws.z_c5, tr, false), BN_C5));
    TRY(e.run_point(e.point_layer(e.ws.z_c5, 128, e.P[EP_CONV6], 256, BN_C5, e.ws.z_c6, tr, true, BN_C6), BN_C6));
    TRY(e.pool(BN_C6, global_feat, e.ws.arg_c, e.ws.zext_c, false));

    if (tr) {
        BnRunItem items[BN_ENC_COUNT];
        for (int i = 0; i < BN_ENC_COUNT; ++i)
            items[i] = {e.ws.bn[i].smean, e.ws.bn[i].suvar, e.bnp[i].rmean, e.bnp[i].rvar, e.ws.bn[i].C, e.s.n_slots};
        TRY(bn_running_update(items, BN_ENC_COUNT, 0.1f, e.st));
    }
    if (in_T) {
        hipError_t err = hipMemcpyAsync(in_T, e.ws.T3, (size_t)Q * 9 * sizeof(float), hipMemcpyDeviceToDevice, e.st);
        if (err != hipSuccess) return fail(AMPNET_E_LAUNCH, "ampnet_encoder_fwd_f32: copy of the input transform: %s", hipGetErrorString(err));
    }
    return AMPNET_OK;
}
