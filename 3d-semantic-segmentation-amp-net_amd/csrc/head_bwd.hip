// head_bwd.hip -- C ABI: ampnet_head_bwd_f32 = autograd backward of SegmentationWithAttention.forward
// (pointNet/model/pointnetAtt.py:176-209) given dL/dlogits; returns dL/d(local features), dL/d(window tokens)
// and writes every parameter gradient.  Reverse of head.hip's launch sequence:
//   head_out_bwd (conv_4, dropout, bn_3 ReLU mask)  -> pw_wgrad/pw_dgrad conv_3 -> pw_wgrad/pw_dgrad conv_2
//   (the per-window column sums of conv_2's gradient ARE the gradient of the per-window token bias)
//   -> small SGEMMs for the token half of conv_2, out_proj, in_proj -> attention_core_bwd -> posenc backward.
#include <cstdlib>
#include "bwd_misc.h"
#include "head.h"

namespace ampnet {
namespace {

constexpr int HB_WAVES = 4, HB_WROWS = 256, HB_ROWS = HB_WAVES * HB_WROWS;   // rows per wave / per workgroup

struct HeadOutBwd {
    const float *dlogits;      // [B, C, P]
    const float *z3;           // [R, 64]
    int z_bf16;                // z3 is a bf16 tensor (precision mode 3)
    const float *scale, *shift, *mean, *invstd;   // bn_3 [64]
    const float *W;            // [C, 64]
    float drop_p;
    uint32_t drop_seed;
    int R, P, C;
    float *dy3;                // [R, 64] masked gradient wrt bn_3 output
    float *part_a, *part_b;    // [blocks, 64]
    float *dWpart;             // [blocks, C * 64 + C]
};

// conv_4 backward + dropout + bn_3/ReLU mask, no LDS in the row loop: lane = channel k of the 64, a wave walks its rows.
// Per row the lane loads z3[row][k] (one 256-byte line per wave), rebuilds a3 = dropout(relu(bn_3(z3))), forms
// da3[k] = sum_c dlogits[row][c] W4[c][k] from five wave-uniform gradients (loaded 64 rows at a time with lane = row and
// broadcast by readlane), masks it and accumulates dW4[c][k], sum dy3, sum dy3 * zhat in registers.
template <bool ZB> __global__ __launch_bounds__(64 * HB_WAVES) void head_out_bwd_kernel(HeadOutBwd a)
{
    __shared__ float red[HB_WAVES][HEAD_MAX_CLASSES + 3][64];
    const int tid = threadIdx.x, k = tid & 63, wave = tid >> 6;
    const float sc = a.scale[k], sh = a.shift[k], me = a.mean[k], is = a.invstd[k];
    float w[HEAD_MAX_CLASSES], dW[HEAD_MAX_CLASSES], db[HEAD_MAX_CLASSES];
#pragma unroll
    for (int c = 0; c < HEAD_MAX_CLASSES; ++c) {
        w[c] = c < a.C ? a.W[c * 64 + k] : 0.f;
        dW[c] = 0.f;
        db[c] = 0.f;
    }
    float pa = 0.f, pb = 0.f;
    const bool has_drop = a.drop_p > 0.f;
    const uint32_t thr = drop_threshold(a.drop_p);
    const float dscale = has_drop ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    const int wrow0 = blockIdx.x * HB_ROWS + wave * HB_WROWS;
    for (int r0 = wrow0; r0 < min(wrow0 + HB_WROWS, a.R); r0 += 64) {
        // gradients of 64 rows: lane = row
        float dreg[HEAD_MAX_CLASSES];
        {
            const int row = r0 + k;
            const bool ok = row < a.R;
            const int b = ok ? row / a.P : 0, p = ok ? row % a.P : 0;
#pragma unroll
            for (int c = 0; c < HEAD_MAX_CLASSES; ++c) dreg[c] = (ok && c < a.C) ? a.dlogits[((size_t)b * a.C + c) * a.P + p] : 0.f;
        }
        // the next eight rows are requested before the current eight are processed (register double buffer)
        float zn[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) zn[u] = ld_act_t<ZB>(a.z3, (size_t)min(r0 + u, a.R - 1) * 64 + k);
#pragma unroll 1
        for (int i0 = 0; i0 < 64; i0 += 8) {
            if (r0 + i0 >= a.R) break;
            float zv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) zv[u] = zn[u];
            if (i0 + 8 < 64) {
#pragma unroll
                for (int u = 0; u < 8; ++u) zn[u] = ld_act_t<ZB>(a.z3, (size_t)min(r0 + i0 + 8 + u, a.R - 1) * 64 + k);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int row = r0 + i0 + u;
                const bool valid = row < a.R;
                float d[HEAD_MAX_CLASSES];
#pragma unroll
                for (int c = 0; c < HEAD_MAX_CLASSES; ++c)
                    d[c] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dreg[c]), i0 + u));   // 0 for rows past the end
                const float zh = (zv[u] - me) * is;
                float av = fmaxf(fmaf(zv[u], sc, sh), 0.f);
                if (has_drop) av = (mix32(((uint32_t)row * 64u + (uint32_t)k) ^ a.drop_seed) >= thr) ? av * dscale : 0.f;
                float v = 0.f;
#pragma unroll
                for (int c = 0; c < HEAD_MAX_CLASSES; ++c) v = fmaf(d[c], w[c], v);
                v *= dscale;
                const float dy = av > 0.f ? v : 0.f;                  // relu > 0 and kept by dropout
                if (valid) a.dy3[(size_t)row * 64 + k] = dy;
#pragma unroll
                for (int c = 0; c < HEAD_MAX_CLASSES; ++c) {
                    dW[c] = fmaf(d[c], av, dW[c]);
                    db[c] += d[c];
                }
                pa += dy;
                pb = fmaf(dy, zh, pb);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < HEAD_MAX_CLASSES; ++c) red[wave][c][k] = dW[c];
    red[wave][HEAD_MAX_CLASSES][k] = pa;
    red[wave][HEAD_MAX_CLASSES + 1][k] = pb;
    red[wave][HEAD_MAX_CLASSES + 2][k] = 0.f;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < HEAD_MAX_CLASSES; ++c)
        if (k == c) red[wave][HEAD_MAX_CLASSES + 2][c] = db[c];       // every lane holds the same sums
    __syncthreads();
    if (wave == 0) {
        auto sum4 = [&](int j, int col) { return (red[0][j][col] + red[1][j][col]) + (red[2][j][col] + red[3][j][col]); };
        float *dst = a.dWpart + (size_t)blockIdx.x * (a.C * 64 + a.C);
        for (int c = 0; c < a.C; ++c) dst[c * 64 + k] = sum4(c, k);
        if (k < a.C) dst[a.C * 64 + k] = sum4(HEAD_MAX_CLASSES + 2, k);
        a.part_a[(size_t)blockIdx.x * 64 + k] = sum4(HEAD_MAX_CLASSES, k);
        a.part_b[(size_t)blockIdx.x * 64 + k] = sum4(HEAD_MAX_CLASSES + 1, k);
    }
}

// per (sample, head): gradients of softmax(q k^T) [dropout] v  wrt q, k, v
__global__ __launch_bounds__(64) void attention_core_bwd_kernel(const float *__restrict__ qkv, const float *__restrict__ probs,
                                                               const float *__restrict__ dctx, float *__restrict__ dqkv, int W,
                                                               float drop_p, uint32_t drop_base_)
{
    __shared__ float sq[HEAD_MAX_W][HEAD_D + 1], sk[HEAD_MAX_W][HEAD_D + 1], sv[HEAD_MAX_W][HEAD_D + 1], sdc[HEAD_MAX_W][HEAD_D + 1];
    __shared__ float sp[HEAD_MAX_W][HEAD_MAX_W + 1], spd[HEAD_MAX_W][HEAD_MAX_W + 1], sds[HEAD_MAX_W][HEAD_MAX_W + 1];
    const int b = blockIdx.x, hd = blockIdx.y, lane = threadIdx.x;
    const float qscale = 0.17677669529663687f;
    for (int e = lane; e < W * HEAD_D; e += 64) {
        const int i = e / HEAD_D, d = e % HEAD_D;
        const size_t o = (size_t)(b * W + i) * (3 * HEAD_E) + hd * HEAD_D + d;
        sq[i][d] = qkv[o] * qscale;
        sk[i][d] = qkv[o + HEAD_E];
        sv[i][d] = qkv[o + 2 * HEAD_E];
        sdc[i][d] = dctx[(size_t)(b * W + i) * HEAD_E + hd * HEAD_D + d];
    }
    const uint32_t thr = drop_threshold(drop_p);
    const float dscale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    for (int e = lane; e < W * W; e += 64) {
        const int i = e / W, j = e % W;
        const size_t o = ((size_t)(b * HEAD_HEADS + hd) * W + i) * W + j;
        const float p = probs[o];
        float keep = dscale;
        if (drop_p > 0.f) keep = (mix32((uint32_t)o ^ drop_base_) >= thr) ? dscale : 0.f;
        sp[i][j] = p;
        spd[i][j] = p * keep;          // dropped probabilities (what multiplied v in the forward)
        sds[i][j] = keep;              // keep factor, reused below
    }
    __syncthreads();
    // dv[j][d] = sum_i pd[i][j] dctx[i][d]
    for (int e = lane; e < W * HEAD_D; e += 64) {
        const int j = e / HEAD_D, d = e % HEAD_D;
        float acc = 0.f;
        for (int i = 0; i < W; ++i) acc = fmaf(spd[i][j], sdc[i][d], acc);
        dqkv[(size_t)(b * W + j) * (3 * HEAD_E) + 2 * HEAD_E + hd * HEAD_D + d] = acc;
    }
    // dP[i][j] = keep * sum_d dctx[i][d] v[j][d]
    for (int e = lane; e < W * W; e += 64) {
        const int i = e / W, j = e % W;
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < HEAD_D; ++d) acc = fmaf(sdc[i][d], sv[j][d], acc);
        spd[i][j] = acc * sds[i][j];
    }
    __syncthreads();
    // dS = P * (dP - rowsum(dP * P))
    if (lane < W) {
        const int i = lane;
        float dot = 0.f;
        for (int j = 0; j < W; ++j) dot = fmaf(spd[i][j], sp[i][j], dot);
        for (int j = 0; j < W; ++j) sds[i][j] = sp[i][j] * (spd[i][j] - dot);
    }
    __syncthreads();
    for (int e = lane; e < W * HEAD_D; e += 64) {
        const int i = e / HEAD_D, d = e % HEAD_D;
        float dq = 0.f, dk = 0.f;
        for (int j = 0; j < W; ++j) {
            dq = fmaf(sds[i][j], sk[j][d], dq);
            dk = fmaf(sds[j][i], sq[j][d], dk);
        }
        const size_t o = (size_t)(b * W + i) * (3 * HEAD_E) + hd * HEAD_D + d;
        dqkv[o] = dq * qscale;
        dqkv[o + HEAD_E] = dk;
    }
}

// positional encoding backward: hidden activations recomputed per token
}  // namespace
}  // namespace ampnet

using namespace ampnet;

#define TRY(x)                            \
    do {                                  \
        int rc_ = (x);                    \
        if (rc_ != AMPNET_OK) return rc_; \
    } while (0)

namespace ampnet {

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

void head_bwd_carve(const HeadShape &s, void *base, HeadBwdWs &w)
{
    Carver c{reinterpret_cast<char *>(base)};
    const size_t Q = (size_t)s.Q, R = (size_t)s.R;
    const size_t blocks = (size_t)cdiv(s.R, HB_ROWS);
    w.dy3 = c.take<float>(R * 64);
    w.dy2 = c.take<float>(R * 128);
    const size_t wch = (size_t)cdiv(s.max_rows, 1024);
    // + 256: the fused backward writes one partial per workgroup (<= 256 of them) instead of one per (window, chunk)
    w.wpart = c.take<float>((Q * wch + 256) * 128 * 128);
    w.dbpart = c.take<float>((Q * wch + 256) * 128);
    w.dgb = c.take<float>(Q * 128);
    w.w4part = c.take<float>(blocks * (HEAD_MAX_CLASSES * 64 + HEAD_MAX_CLASSES));
    const size_t np = ((Q * (size_t)s.chunks > blocks ? Q * (size_t)s.chunks : blocks) + 256) * 128;
    w.part_a = c.take<float>(np);
    w.part_b = c.take<float>(np);
    const int Cs[2] = {128, 64};
    for (int i = 0; i < 2; ++i) {
        w.P1[i] = c.take<float>(Cs[i]);
        w.P2[i] = c.take<float>(Cs[i]);
        w.P3[i] = c.take<float>(Cs[i]);
        w.slot_ab[i] = c.take<float>(2 * Cs[i]);
    }
    const bool gru = s.kind == HEAD_KIND_GRU;
    w.d_g2 = c.take<float>(Q * (gru ? GRU_H : 256));
    w.d_ctx = c.take<float>(Q * 256);
    w.d_qkv = c.take<float>(Q * (gru ? 3 * GRU_H : 768));
    w.hid = c.take<float>(Q * (gru ? GRU_H : 16));
    w.slope = c.take<float>(gru ? 0 : Q * 16);
    w.d_hid = c.take<float>(gru ? 0 : Q * 16);
    w.tot_off = c.take<int>(2);
    w.bytes = align_up(c.off, 256);
}


int attention_core_bwd(const float *qkv, const float *probs, const float *dctx, float *dqkv, int B, int W, float drop_p, uint32_t drop_base_,
                       hipStream_t st)
{
    hipLaunchKernelGGL(attention_core_bwd_kernel, dim3(B, HEAD_HEADS), dim3(64), 0, st, qkv, probs, dctx, dqkv, W, drop_p, drop_base_);
    return check_launch("attention_core_bwd_kernel");
}

int head_points_bwd(const HeadShape &s, HeadWs &f, HeadBwdWs &b, const HeadPointParams &p_, const HeadPointGrads &g, const float *lo,
                    const int32_t *win_off, float drop_p, uint32_t seed, const float *dlogits, float *d_lo, hipStream_t st)
{
    const int B = s.B, Q = s.Q, R = s.R, C = s.n_classes, total_rows = s.R, max_rows = s.max_rows;
    const int blocks = cdiv(R, HB_ROWS);
    const int wch = cdiv(max_rows, 1024);
    const int zb = z_storage_bf16() ? 1 : 0;         // z2 / z3 of this forward workspace are bf16 tensors (precision mode 3)
    // the parameter-gradient partials of the three layers stay in their regions of wpart / dbpart / w4part and are reduced by ONE launch at
    // the end (seven reduce_windows launches before)
    ReduceItem red[REDUCE_MULTI_MAX];
    int n_red = 0;
    // BatchNorm-backward constants inside the consuming fused kernel (PwBwd.fin_*) instead of two bn_bwd_finalize launches: fp32 fused path only
    const char *fenv0 = getenv("AMPNET_FUSED_BWD"), *kenv = getenv("AMPNET_BWD_FIN_IN_KERNEL");
    const bool fin_in_kernel = !(fenv0 && fenv0[0] == '0') && !(kenv && kenv[0] == '0') && !bwd_operands_bf16() && !sync_bn_on();
    float *pa3 = b.part_a + align_up((size_t)blocks * 64, 64), *pb3 = b.part_b + align_up((size_t)blocks * 64, 64);   // conv_3's partials: behind head_out_bwd's
    // ---- conv_4 + dropout + bn_3/ReLU mask ------------------------------------------------------------
    {
        HeadOutBwd o;
        o.dlogits = dlogits; o.z3 = f.z3; o.z_bf16 = zb;
        o.scale = f.bn3.scale; o.shift = f.bn3.shift; o.mean = f.bn3.mean; o.invstd = f.bn3.invstd;
        o.W = p_.conv4_w; o.drop_p = drop_p; o.drop_seed = drop_base(seed, 2);
        o.R = R; o.P = R / B; o.C = C;
        o.dy3 = b.dy3; o.part_a = b.part_a; o.part_b = b.part_b; o.dWpart = b.w4part;
        if (o.z_bf16) hipLaunchKernelGGL(head_out_bwd_kernel<true>, dim3(blocks), dim3(64 * HB_WAVES), 0, st, o);
        else hipLaunchKernelGGL(head_out_bwd_kernel<false>, dim3(blocks), dim3(64 * HB_WAVES), 0, st, o);
        TRY(check_launch("head_out_bwd_kernel"));
        red[n_red++] = ReduceItem{b.w4part, blocks, (long)(C * 64 + C), 1, C * 64, C * 64, g.conv4_w, C * 64};
        red[n_red++] = ReduceItem{b.w4part + C * 64, blocks, (long)(C * 64 + C), 1, C, C, g.conv4_b, C};
        if (!fin_in_kernel) {                    // else conv_3's fused backward forms bn_3's constants from these partials itself
            BnBwdFinalize fz;
            fz.part_a = b.part_a; fz.part_b = b.part_b; fz.win_off = nullptr; fz.Q = 1; fz.chunks = blocks; fz.n_slots = 1; fz.C = 64;
            fz.uniform_rows = R;
            fz.gamma = p_.bn3_w; fz.mean = f.bn3.mean; fz.invstd = f.bn3.invstd;
            fz.P1 = b.P1[1]; fz.P2 = b.P2[1]; fz.P3 = b.P3[1]; fz.slot_ab = b.slot_ab[1];
            TRY(bn_bwd_finalize(fz, st));
        }
    }
    // ---- conv_3: z3 = dropout(relu(bn_2(z2))) W3^T + b3 --------------------------------------------------
    GradSrc g3;
    g3.dy = b.dy3; g3.z = f.z3; g3.C = 64; g3.P1 = b.P1[1]; g3.P2 = b.P2[1]; g3.P3 = b.P3[1]; g3.z_bf16 = zb;
    ActSrc a2;
    a2.z = f.z2; a2.C = 128; a2.z_bf16 = zb; a2.s = f.bn2.scale; a2.t = f.bn2.shift; a2.drop_p = drop_p; a2.drop_seed = drop_base(seed, 1);
    const char *fenv = getenv("AMPNET_FUSED_BWD");
    const bool fused = !(fenv && fenv[0] == '0');
    size_t conv3_w_floats = 0, conv3_b_floats = 0;             // conv_3's partial regions (fused path): conv_2's go behind them
    int conv3_parts = 0;                                       // > 0: bn_2's constants are still owed (conv_2's fused kernel forms them)
    AMPNET_REQUIRE(!zb || fused, "ampnet_head_bwd_f32: bf16 activation storage needs the fused backward");
    if (fused) {
        // one pass over (dy3, z3, z2): weight + bias gradient partials and dy2 with bn_2's backward sums
        PwBwd p;
        p.g = g3; p.prev = a2; p.prev_mean = f.bn2.mean; p.prev_invstd = f.bn2.invstd;
        p.W = p_.conv3_w; p.ldw = 128; p.out = b.dy2; p.dWpart = b.wpart; p.dbpart = b.dbpart;
        p.part_a = fin_in_kernel ? pa3 : b.part_a; p.part_b = fin_in_kernel ? pb3 : b.part_b;
        p.win_off = win_off; p.Q = Q; p.n_slots = 1; p.max_rows = max_rows; p.rows_hint = R;
        p.blocks_per_slot = pw_bwd_blocks(Q, 1, max_rows);
        const int nblk = p.blocks_per_slot;
        if (fin_in_kernel) {
            p.fin_part_a = b.part_a; p.fin_part_b = b.part_b; p.fin_parts = blocks; p.fin_rows = R;
            p.fin_gamma = p_.bn3_w; p.fin_mean = f.bn3.mean; p.fin_invstd = f.bn3.invstd;
            p.fin_P1 = b.P1[1]; p.fin_P2 = b.P2[1]; p.fin_P3 = b.P3[1]; p.fin_slot_ab = b.slot_ab[1];
        }
        TRY(pw_bwd_fused(p, st));
        red[n_red++] = ReduceItem{b.wpart, nblk, 64L * 128, 64, 128, 128, g.conv3_w, 128};
        red[n_red++] = ReduceItem{b.dbpart, nblk, 64L, 1, 64, 64, g.conv3_b, 64};
        conv3_w_floats = align_up((size_t)nblk * 64 * 128, 64);
        conv3_b_floats = align_up((size_t)nblk * 64, 64);
        conv3_parts = nblk;
        const int cpw2 = cdiv(max_rows, pw_bwd_item_rows());
        int ipb2 = cpw2 < 4 ? cpw2 : 4;
        while (cpw2 % ipb2) --ipb2;
        if (!(fin_in_kernel && cpw2 / ipb2 <= wch)) {          // conv_2 will not run the fused kernel (or not in-kernel): bn_2's constants by launch
            BnBwdFinalize fz;
            fz.part_a = p.part_a; fz.part_b = p.part_b; fz.win_off = win_off; fz.Q = Q; fz.chunks = 1; fz.part_Q = nblk; fz.n_slots = 1; fz.C = 128;
            fz.uniform_rows = (long)max_rows * Q == (long)total_rows ? max_rows : 0;
            fz.gamma = p_.bn2_w; fz.mean = f.bn2.mean; fz.invstd = f.bn2.invstd;
            fz.P1 = b.P1[0]; fz.P2 = b.P2[0]; fz.P3 = b.P3[0]; fz.slot_ab = b.slot_ab[0];
            TRY(bn_bwd_finalize(fz, st));
            conv3_parts = 0;
        }
    } else {
        if (n_red) TRY(reduce_windows_multi(red, n_red, st));
        n_red = 0;
        PwWgrad w;
        w.x = g3; w.y = a2; w.dWpart = b.wpart; w.ldp = 128; w.dbpart = b.dbpart;
        w.win_off = win_off; w.Q = Q; w.n_slots = 1; w.rows_hint = R; w.chunk_rows = 1024; w.chunks = wch;
        TRY(pw_wgrad(w, st));
        TRY(reduce_windows(b.wpart, Q * wch, 64 * 128, 64, 128, 128, g.conv3_w, 128, 0, st));
        TRY(reduce_windows(b.dbpart, Q * wch, 64, 1, 64, 64, g.conv3_b, 64, 0, st));
        PwDgrad d;
        d.g = g3; d.W = p_.conv3_w; d.ldw = 128; d.prev = a2; d.prev_mean = f.bn2.mean; d.prev_invstd = f.bn2.invstd;
        d.out = b.dy2; d.cp = 128; d.part_a = b.part_a; d.part_b = b.part_b;
        d.win_off = win_off; d.Q = Q; d.n_slots = 1; d.chunk_rows = s.chunk_rows; d.chunks = s.chunks; d.rows_hint = R;
        TRY(pw_dgrad(d, st));
        BnBwdFinalize fz;
        fz.part_a = b.part_a; fz.part_b = b.part_b; fz.win_off = win_off; fz.Q = Q; fz.chunks = s.chunks; fz.n_slots = 1; fz.C = 128;
        fz.uniform_rows = (long)max_rows * Q == (long)total_rows ? max_rows : 0;
        fz.gamma = p_.bn2_w; fz.mean = f.bn2.mean; fz.invstd = f.bn2.invstd;
        fz.P1 = b.P1[0]; fz.P2 = b.P2[0]; fz.P3 = b.P3[0]; fz.slot_ab = b.slot_ab[0];
        TRY(bn_bwd_finalize(fz, st));
    }
    // ---- conv_2: z2 = lo W2[:, :64]^T + (token W2[:, 64:]^T + b2)[window] ------------------------------------
    GradSrc g2;
    g2.dy = b.dy2; g2.z = f.z2; g2.C = 128; g2.P1 = b.P1[0]; g2.P2 = b.P2[0]; g2.P3 = b.P3[0]; g2.z_bf16 = zb;
    // fused form: every workgroup stays inside one window, so its column sums of g2 are a per-window partial of the
    // token-bias gradient
    const int cpw = cdiv(max_rows, pw_bwd_item_rows());
    int ipb = cpw < 4 ? cpw : 4;
    while (cpw % ipb) --ipb;
    const int bpw = cpw / ipb;                                  // workgroups per window
    AMPNET_REQUIRE(!zb || bpw <= wch, "ampnet_head_bwd_f32: bf16 activation storage needs the fused conv_2 backward");
    if (fused && bpw <= wch) {
        PwBwd p;
        p.g = g2; p.prev.z = lo; p.prev.C = 64;
        float *wp2 = b.wpart + conv3_w_floats, *dbp2 = b.dbpart + conv3_b_floats;
        p.W = p_.conv2_w; p.ldw = p_.conv2_ld; p.out = d_lo; p.dWpart = wp2; p.dbpart = dbp2;
        p.win_off = win_off; p.Q = Q; p.n_slots = 1; p.max_rows = max_rows; p.rows_hint = R;
        p.items_per_block = ipb; p.blocks_per_slot = Q * bpw;
        if (conv3_parts > 0) {
            p.fin_part_a = pa3; p.fin_part_b = pb3; p.fin_parts = conv3_parts; p.fin_rows = R;
            p.fin_gamma = p_.bn2_w; p.fin_mean = f.bn2.mean; p.fin_invstd = f.bn2.invstd;
            p.fin_P1 = b.P1[0]; p.fin_P2 = b.P2[0]; p.fin_P3 = b.P3[0]; p.fin_slot_ab = b.slot_ab[0];
        }
        TRY(pw_bwd_fused(p, st));
        red[n_red++] = ReduceItem{wp2, Q * bpw, 128L * 64, 128, 64, 64, g.conv2_w, g.conv2_ld};
        red[n_red++] = ReduceItem{dbp2, bpw, 128L, Q, 128, bpw * 128, b.dgb, 128};       // per window: the gradient of its token bias
        red[n_red++] = ReduceItem{dbp2, Q * bpw, 128L, 1, 128, 128, g.conv2_b, 128};      // all of them: conv_2.bias
    } else {
        if (n_red) TRY(reduce_windows_multi(red, n_red, st));     // wpart / dbpart are about to be reused from their start
        n_red = 0;
        ActSrc yl;
        yl.z = lo; yl.C = 64;
        PwWgrad w;
        w.x = g2; w.y = yl; w.dWpart = b.wpart; w.ldp = 64; w.dbpart = b.dbpart;      // dbpart[q] = d(token bias of window q)
        w.win_off = win_off; w.Q = Q; w.n_slots = 1; w.rows_hint = R; w.chunk_rows = 1024; w.chunks = wch;
        TRY(pw_wgrad(w, st));
        TRY(reduce_windows(b.wpart, Q * wch, 128 * 64, 128, 64, 64, g.conv2_w, g.conv2_ld, 0, st));
        // per-window sums of the chunk partials = gradient of the per-window token bias; their sum = conv_2.bias gradient
        TRY(reduce_windows(b.dbpart, wch, 128, Q, 128, wch * 128, b.dgb, 128, 0, st));
        TRY(reduce_windows(b.dgb, Q, 128, 1, 128, 128, g.conv2_b, 128, 0, st));
        PwDgrad d;
        d.g = g2; d.W = p_.conv2_w; d.ldw = p_.conv2_ld; d.out = d_lo; d.cp = 64;
        d.win_off = win_off; d.Q = Q; d.n_slots = 1; d.chunk_rows = s.chunk_rows; d.chunks = s.chunks; d.rows_hint = R;
        TRY(pw_dgrad(d, st));
    }
    if (n_red) TRY(reduce_windows_multi(red, n_red, st));
    // ---- BatchNorm weight / bias gradients ---------------------------------------------------------------------------
    {
        BnGradItem items[2] = {{b.slot_ab[0], g.bn2_w, g.bn2_b, 128, 1}, {b.slot_ab[1], g.bn3_w, g.bn3_b, 64, 1}};
        TRY(bn_param_grads(items, 2, st));
    }
    return AMPNET_OK;
}

}  // namespace ampnet

extern "C" size_t ampnet_head_bwd_workspace_bytes(int B, int W, int total_rows, int max_rows, int n_classes)
{
    if (B < 1 || W < 1 || total_rows < 1 || max_rows < 1) return 0;
    HeadBwdWs w;
    head_bwd_carve(head_shape(B, W, total_rows, max_rows, n_classes, 1), nullptr, w);
    return w.bytes;
}

extern "C" int ampnet_head_bwd_f32(const float *const *params_host, float *const *grads_host, const float *lo,
                                   const float *centroids, const int32_t *win_off, int B, int W, int total_rows, int max_rows,
                                   int n_classes, float drop_p, uint32_t seed, const float *dlogits, float *d_lo, float *d_gl,
                                   void *fwd_workspace, size_t fwd_workspace_bytes, void *bwd_workspace,
                                   size_t bwd_workspace_bytes, void *stream)
{
    AMPNET_REQUIRE(params_host && grads_host && lo && centroids && win_off && dlogits && d_lo && d_gl && fwd_workspace && bwd_workspace,
                   "ampnet_head_bwd_f32: null pointer");
    AMPNET_REQUIRE(B >= 1 && W >= 1 && W <= HEAD_MAX_W && total_rows % B == 0, "ampnet_head_bwd_f32: bad sizes");
    AMPNET_REQUIRE(n_classes >= 1 && n_classes <= HEAD_MAX_CLASSES, "ampnet_head_bwd_f32: n_classes=%d", n_classes);
    TRY(ws_tag_check(fwd_workspace, "ampnet_head_bwd_f32"));
    hipStream_t st = (hipStream_t)stream;
    const HeadShape s = head_shape(B, W, total_rows, max_rows, n_classes, 1);
    HeadWs f;
    head_carve(s, fwd_workspace, f);
    HeadBwdWs b;
    head_bwd_carve(s, bwd_workspace, b);
    if (f.bytes > fwd_workspace_bytes) return fail(AMPNET_E_WORKSPACE, "ampnet_head_bwd_f32: forward workspace %zu B < %zu B", fwd_workspace_bytes, f.bytes);
    if (b.bytes > bwd_workspace_bytes) return fail(AMPNET_E_WORKSPACE, "ampnet_head_bwd_f32: backward workspace %zu B < %zu B", bwd_workspace_bytes, b.bytes);
    const float *const *P = params_host;
    float *const *G = grads_host;
    const int Q = s.Q;
    HeadPointParams pp;
    pp.conv2_w = P[HP_CONV2_W]; pp.conv2_ld = 320;
    pp.conv3_w = P[HP_CONV3_W]; pp.conv3_b = P[HP_CONV3_B]; pp.conv4_w = P[HP_CONV4_W]; pp.conv4_b = P[HP_CONV4_B];
    pp.bn2_w = P[HP_BN2_W]; pp.bn2_b = P[HP_BN2_B]; pp.bn3_w = P[HP_BN3_W]; pp.bn3_b = P[HP_BN3_B];
    HeadPointGrads pg;
    pg.conv2_w = G[HP_CONV2_W]; pg.conv2_ld = 320; pg.conv2_b = G[HP_CONV2_B];
    pg.conv3_w = G[HP_CONV3_W]; pg.conv3_b = G[HP_CONV3_B]; pg.conv4_w = G[HP_CONV4_W]; pg.conv4_b = G[HP_CONV4_B];
    pg.bn2_w = G[HP_BN2_W]; pg.bn2_b = G[HP_BN2_B]; pg.bn3_w = G[HP_BN3_W]; pg.bn3_b = G[HP_BN3_B];
    TRY(head_points_bwd(s, f, b, pp, pg, lo, win_off, drop_p, seed, dlogits, d_lo, st));
    // ---- token path: gbias = g2tok W2[:, 64:]^T + b2 ; g2tok = ctx Wo^T + bo ; qkv = tok Wi^T + bi -------------
    const float *d_gbias = b.dgb;                                                           // [Q, 128]
    TRY(sgemm_linear_bwd(Q, 128, 256, d_gbias, 128, f.g2, 256, P[HP_CONV2_W] + 64, 320, G[HP_CONV2_W] + 64, 320, b.d_g2, 256, st));
    {
        LinBwdOpt o;
        o.db = G[HP_OUTPROJ_B];
        TRY(sgemm_linear_bwd(Q, 256, 256, b.d_g2, 256, f.ctx, 256, P[HP_OUTPROJ_W], 256, G[HP_OUTPROJ_W], 256, b.d_ctx, 256, st, o));
    }
    TRY(attention_core_bwd(f.qkv, f.probs, b.d_ctx, b.d_qkv, B, W, drop_p, drop_base(seed, 0), st));
    {
        LinBwdOpt o;
        o.db = G[HP_INPROJ_B];
        TRY(sgemm_linear_bwd(Q, 768, 256, b.d_qkv, 768, f.tok, 256, P[HP_INPROJ_W], 256, G[HP_INPROJ_W], 256, d_gl, 256, st, o));   // d_tok = d_gl = d_pos
    }
    // ---- positional encoding: pos = leaky(cent W1^T + b1) W2^T + b2 --------------------------------------------
    // the hidden layer and its leaky-ReLU slope were kept by the forward (HeadWs.pe_*); the slope multiplies the data gradient as it is written
    {
        LinBwdOpt o;
        o.db = G[HP_FC2_B];
        o.dx_mul = f.pe_slope;
        TRY(sgemm_linear_bwd(Q, 256, 16, d_gl, 256, f.pe_hid, 16, P[HP_FC2_W], 16, G[HP_FC2_W], 16, b.d_hid, 16, st, o));
    }
    TRY(sgemm_wgrad_bias(Q, 16, 2, b.d_hid, 16, centroids, 2, G[HP_FC1_W], 2, G[HP_FC1_B], st));
    return AMPNET_OK;
}
