// encoder_bwd.hip -- C ABI: ampnet_encoder_bwd_f32 = autograd backward of BasePointNet.forward for all windows
// of a step (the reference: loss.backward() at train_pointnet-attention.py:467 through pointnetAtt.py:80-112).
//
// Walks the forward launch sequence of encoder.hip in reverse.  Per BatchNorm'ed layer l (reverse order):
//     [pool_bwd | bn_bwd_finalize]  ->  pw_bwd_fused(l): weight gradient partials + dy_{l-1} in one pass  ->  reduce_windows
//     -> next layer's finalize            (AMPNET_FUSED_BWD=0: the separate pw_wgrad / pw_dgrad launches of pw_bwd.hip)
// using only what the train-mode forward left in its workspace (pre-BatchNorm z, batch mean / invstd / affine,
// argmax rows) plus two ping-pong dy buffers.  Train mode only (batch statistics).
#include <cstdlib>
#include "bwd_misc.h"
#include "encoder.h"

namespace ampnet {
// [32, 64] ones and zeros: the "BatchNorm-backward constants" of a gradient that is passed through as it is (g = dy: the fused bmm backward).
// Device globals filled once per device and process instead of once per step (a 5 us launch).
__device__ float g_ident_ones[32 * 64];
__device__ float g_ident_zeros[32 * 64];
static int identity_constants(int n_slots, const float **ones, const float **zeros, hipStream_t st)
{
    AMPNET_REQUIRE(n_slots <= 32, "identity_constants: %d slots", n_slots);
    static bool ready[64] = {};
    int dev = 0;
    float *po = nullptr, *pz = nullptr;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || hipGetSymbolAddress(reinterpret_cast<void **>(&po), HIP_SYMBOL(g_ident_ones)) != hipSuccess ||
        hipGetSymbolAddress(reinterpret_cast<void **>(&pz), HIP_SYMBOL(g_ident_zeros)) != hipSuccess)
        return fail(AMPNET_E_LAUNCH, "identity_constants: device symbol lookup failed");
    if (!ready[dev]) {
        int rc = fill_f32_pair(po, 1.0f, pz, 0.0f, (size_t)32 * 64, st);
        if (rc != AMPNET_OK) return rc;
        if (hipStreamSynchronize(st) != hipSuccess) return fail(AMPNET_E_LAUNCH, "identity_constants: synchronize failed");
        ready[dev] = true;
    }
    *ones = po;
    *zeros = pz;
    return AMPNET_OK;
}

namespace {

struct BnBwdSlot {
    float *P1, *P2, *P3, *slot_ab;   // [n_slots, C] x3, [n_slots, C, 2]
};

struct EncBwdWs {
    float *dyA, *dyB;          // [R, 128] ping-pong
    float *d_local, *d_h;      // [R, 64]
    float *wpart, *dbpart;     // [Q, 256 * 128], [Q, 256]
    float *dpm;                // [Q, 256] masked pooled gradient
    float *a1, *a2, *da2, *g2, *da1, *g1, *d_pool;   // FC: [Q,256] [Q,128] [Q,128] [Q,128] [Q,256] [Q,256] [Q,256]
    float *fc_split;                                 // [8, Q, 128] K-split partials of the 4096-output fc_3 data gradient
    float *dT64, *dT64t;       // [Q, 4096]
    float *dWeff, *dT3;        // [Q, 576], [Q, 12]
    float *srows, *Gm, *c0, *gram, *asum, *wgram;   // [Q*256,128] [S,128,128] [S,128] [S,128,128] [S,128] [S,256,128]: pooled-layer algebra
    int *srow_row, *srow_cnt;  // [Q*256], [Q]
    BnBwdSlot bn[BN_ENC_COUNT];
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

constexpr int WG_CHUNK_ROWS = 1024;     // rows per pw_wgrad workgroup
static int wg_chunks(const EncShape &s) { return cdiv(s.max_rows, WG_CHUNK_ROWS); }

static const int kBnC2[BN_ENC_COUNT] = {64, 128, 256, 256, 128, 64, 64, 64, 128, 256, 256, 128, 64, 128, 128, 256};

void enc_bwd_carve(const EncShape &s, void *base, EncBwdWs &w)
{
    Carver c{reinterpret_cast<char *>(base)};
    const size_t R = (size_t)s.R, Q = (size_t)s.Q;
    w.dyA = c.take<float>(R * 128);
    w.dyB = c.take<float>(R * 128);
    w.d_local = c.take<float>(R * 64);
    w.d_h = c.take<float>(R * 64);
    // + 320 + n_slots: the fused backward writes one partial per workgroup (<= 256 + n_slots of them)
    w.wpart = c.take<float>((Q * (size_t)wg_chunks(s) + 320 + s.n_slots) * 256 * 128);
    w.dbpart = c.take<float>((Q * (size_t)wg_chunks(s) + 320 + s.n_slots) * 256);
    w.dpm = c.take<float>(Q * 256);
    w.a1 = c.take<float>(Q * 256);
    w.a2 = c.take<float>(Q * 128);
    w.da2 = c.take<float>(Q * 128);
    w.fc_split = c.take<float>(8 * Q * 128);
    w.g2 = c.take<float>(Q * 128);
    w.da1 = c.take<float>(Q * 256);
    w.g1 = c.take<float>(Q * 256);
    w.d_pool = c.take<float>(Q * 256);
    w.dT64 = c.take<float>(Q * 4096);
    w.dT64t = c.take<float>(Q * (size_t)s.chunks * 4096);     // per (window, chunk) partials of the per-window transform gradient
    w.dWeff = c.take<float>(Q * 576);
    w.dT3 = c.take<float>(Q * 12);
    w.srows = c.take<float>(Q * 256 * 128);
    w.Gm = c.take<float>((size_t)s.n_slots * 128 * 128);
    w.c0 = c.take<float>((size_t)s.n_slots * 128);
    w.gram = c.take<float>((size_t)s.n_slots * 128 * 128);
    w.wgram = c.take<float>((size_t)s.n_slots * 256 * 128);
    w.asum = c.take<float>((size_t)s.n_slots * 128);
    w.srow_row = c.take<int>(Q * 256);
    w.srow_cnt = c.take<int>(Q);
    for (int i = 0; i < BN_ENC_COUNT; ++i) {
        const size_t n = (size_t)s.n_slots * kBnC2[i];
        w.bn[i].P1 = c.take<float>(n);
        w.bn[i].P2 = c.take<float>(n);
        w.bn[i].P3 = c.take<float>(n);
        w.bn[i].slot_ab = c.take<float>(2 * n);
    }
    w.bytes = align_up(c.off, 256);
}

#define TRY(x)                            \
    do {                                  \
        int rc_ = (x);                    \
        if (rc_ != AMPNET_OK) return rc_; \
    } while (0)

struct EncBwd {
    hipStream_t st;
    EncShape s;
    EncWs f;          // forward workspace (saved activations)
    EncBwdWs b;
    const float *const *P;
    float *const *G;
    const int *win_off;
    const float *gamma[BN_ENC_COUNT];
    bool fused = true;         // AMPNET_FUSED_BWD=0 falls back to the separate pw_wgrad / pw_dgrad launches
    bool zb = false;           // precision mode 3: the forward workspace holds its z tensors as bf16
    // weight-gradient partials of the fused layers stay in wpart (one region per layer) and are reduced by ONE launch at the end
    mutable size_t wpart_used = 0;
    size_t wpart_cap = 0;
    mutable ReduceItem deferred[REDUCE_MULTI_MAX];
    mutable int n_deferred = 0;
    // BatchNorm-backward sums: every producer writes its partials (sum dy, sum dy zhat) into one of TWO regions of the forward
    // workspace's partial arrays, alternating, and the constants of that layer are owed: the next fused layer forms them itself from the
    // region the producer wrote (PwBwd.fin_*: no bn_bwd_finalize launch) while it writes its own partials into the other region;
    // consumers that are not the fp32 fused kernel get the launch (settle()).
    struct Owed {
        bool open = false;
        int bn = -1, C = 0, parts = 0, chunks = 1;
        const float *pa = nullptr, *pb = nullptr;
    };
    mutable Owed owed;
    mutable int part_region = 0;
    bool in_kernel_fin = false;      // fp32 fused kernel, uniform windows, no global-batch BatchNorm
    size_t region_floats() const { return enc_bwd_part_region_floats(s); }
    float *part_a_out() const { return f.part_sum + (size_t)part_region * region_floats(); }
    float *part_b_out() const { return f.part_sq + (size_t)part_region * region_floats(); }
    int settle() const            // launch the finalize that is still owed (its consumer cannot do it in-kernel)
    {
        if (!owed.open) return AMPNET_OK;
        owed.open = false;
        BnBwdFinalize fz;
        fz.part_a = owed.pa; fz.part_b = owed.pb; fz.win_off = win_off;
        fz.Q = s.Q; fz.chunks = owed.chunks; fz.part_Q = owed.parts; fz.n_slots = s.n_slots; fz.C = owed.C;
        fz.uniform_rows = (long)s.max_rows * s.Q == (long)s.R ? s.max_rows : 0;
        const int bn = owed.bn;
        fz.gamma = gamma[bn]; fz.mean = f.bn[bn].mean; fz.invstd = f.bn[bn].invstd;
        fz.P1 = b.bn[bn].P1; fz.P2 = b.bn[bn].P2; fz.P3 = b.bn[bn].P3; fz.slot_ab = b.bn[bn].slot_ab;
        return bn_bwd_finalize(fz, st);
    }
    // the input layers' weight-gradient kernel as consumer of an owed finalize (bwd_misc.h: PwInputWgrad.fin_*)
    int input_fin(PwInputWgrad &w, int bn) const
    {
        if (owed.open && in_kernel_fin && owed.bn == bn && owed.chunks == 1 && owed.C == 64) {
            w.fin_part_a = owed.pa; w.fin_part_b = owed.pb; w.fin_parts = owed.parts;
            w.fin_rows = (s.Q / s.n_slots) * s.max_rows;
            w.fin_gamma = gamma[bn]; w.fin_mean = f.bn[bn].mean; w.fin_invstd = f.bn[bn].invstd;
            w.fin_P1 = b.bn[bn].P1; w.fin_P2 = b.bn[bn].P2; w.fin_P3 = b.bn[bn].P3; w.fin_slot_ab = b.bn[bn].slot_ab;
            owed.open = false;
            return AMPNET_OK;
        }
        return settle();
    }
    int flush_deferred() const
    {
        if (n_deferred == 0) return AMPNET_OK;
        const int n = n_deferred;
        n_deferred = 0;
        wpart_used = 0;
        return reduce_windows_multi(deferred, n, st);
    }

    GradSrc dense(const float *dy, const float *z, int bn, int C) const
    {
        GradSrc g;
        g.dy = dy; g.z = z; g.C = C;
        g.z_bf16 = (zb && bn >= 0) ? 1 : 0;            // every BatchNorm'ed z of the workspace (the bmm path passes bn < 0 and fp32 tensors)
        if (bn >= 0) { g.P1 = b.bn[bn].P1; g.P2 = b.bn[bn].P2; g.P3 = b.bn[bn].P3; }
        return g;
    }
    GradSrc sparse(const int *arg, const float *dpm, int slot_major, const float *z, int bn) const
    {
        GradSrc g;
        g.arg = arg; g.dpool = dpm; g.dpool_slot_major = slot_major; g.z = z; g.C = 256;
        g.P1 = b.bn[bn].P1; g.P2 = b.bn[bn].P2; g.P3 = b.bn[bn].P3;
        return g;
    }
    ActSrc act(const float *z, int bn, int C) const
    {
        ActSrc a;
        a.z = z; a.C = C;
        a.z_bf16 = (zb && bn >= 0) ? 1 : 0;
        if (bn >= 0) { a.s = f.bn[bn].scale; a.t = f.bn[bn].shift; }
        return a;
    }
    // dW[cx][cy] of a shared weight: per-window partials then ordered reduction
    int wgrad(const GradSrc &x, const ActSrc &y, float *dW) const
    {
        PwWgrad w;
        w.x = x; w.y = y; w.dWpart = b.wpart; w.ldp = y.C;
        w.win_off = win_off; w.Q = s.Q; w.n_slots = s.n_slots; w.rows_hint = s.R;
        w.chunk_rows = WG_CHUNK_ROWS; w.chunks = wg_chunks(s);
        TRY(pw_wgrad(w, st));
        return reduce_windows(b.wpart, s.Q * w.chunks, (long)x.C * y.C, x.C, y.C, y.C, dW, y.C, 0, st);
    }
    // dy of layer `prev_bn` (masked) + its BatchNorm-backward partial sums, then that layer's constants
    int dgrad(const GradSrc &g, const float *W, int ldw, const float *prev_z, int prev_bn, int cp, const float *add, float *out) const
    {
        PwDgrad d;
        d.g = g; d.W = W; d.ldw = ldw; d.add = add; d.out = out; d.cp = cp;
        if (prev_z) {
            d.prev = act(prev_z, prev_bn, cp);
            d.prev_mean = f.bn[prev_bn].mean; d.prev_invstd = f.bn[prev_bn].invstd;
            d.part_a = f.part_sum; d.part_b = f.part_sq;
        }
        d.win_off = win_off; d.Q = s.Q; d.n_slots = s.n_slots; d.chunk_rows = s.chunk_rows; d.chunks = s.chunks; d.rows_hint = s.R;
        TRY(pw_dgrad(d, st));
        if (prev_z) {
            BnBwdFinalize fz;
            fz.part_a = f.part_sum; fz.part_b = f.part_sq; fz.win_off = win_off;
            fz.Q = s.Q; fz.chunks = s.chunks; fz.n_slots = s.n_slots; fz.C = cp;
            fz.uniform_rows = (long)s.max_rows * s.Q == (long)s.R ? s.max_rows : 0;
            fz.gamma = gamma[prev_bn]; fz.mean = f.bn[prev_bn].mean; fz.invstd = f.bn[prev_bn].invstd;
            fz.P1 = b.bn[prev_bn].P1; fz.P2 = b.bn[prev_bn].P2; fz.P3 = b.bn[prev_bn].P3; fz.slot_ab = b.bn[prev_bn].slot_ab;
            TRY(bn_bwd_finalize(fz, st));
        }
        return AMPNET_OK;
    }
    // weight gradient + data gradient of one shared layer in one pass over (dy, z, z_prev) (pw_bwd_fused.hip); prev_bn < 0:
    // the layer's input is prev_z itself (no activation, no mask, no sums)
    int layer_bwd(const GradSrc &g, const float *W, float *dW, const float *prev_z, int prev_bn, int cy, const float *add, float *out) const
    {
        if (!fused || !pw_bwd_supported(g.C, cy)) {
            TRY(settle());                    // the separate kernels read the constants from memory
            ActSrc y;
            if (prev_bn >= 0) y = act(prev_z, prev_bn, cy);
            else { y.z = prev_z; y.C = cy; }
            TRY(flush_deferred());            // wgrad() reuses wpart from its start
            TRY(wgrad(g, y, dW));
            return dgrad(g, W, cy, prev_bn >= 0 ? prev_z : nullptr, prev_bn, cy, add, out);
        }
        PwBwd p;
        p.g = g;
        // the constants of the layer g belongs to: formed inside this launch when they are still owed (and g is that layer), else launched now
        if (owed.open && in_kernel_fin && g.P1 == b.bn[owed.bn].P1 && owed.chunks == 1) {
            p.fin_part_a = owed.pa; p.fin_part_b = owed.pb; p.fin_parts = owed.parts;
            p.fin_rows = (s.Q / s.n_slots) * s.max_rows;
            p.fin_gamma = gamma[owed.bn]; p.fin_mean = f.bn[owed.bn].mean; p.fin_invstd = f.bn[owed.bn].invstd;
            p.fin_P1 = b.bn[owed.bn].P1; p.fin_P2 = b.bn[owed.bn].P2; p.fin_P3 = b.bn[owed.bn].P3; p.fin_slot_ab = b.bn[owed.bn].slot_ab;
            owed.open = false;
            part_region ^= 1;                 // this launch writes the other region
        } else {
            TRY(settle());
        }
        if (prev_bn >= 0) {
            p.prev = act(prev_z, prev_bn, cy);
            p.prev_mean = f.bn[prev_bn].mean; p.prev_invstd = f.bn[prev_bn].invstd;
            p.part_a = part_a_out(); p.part_b = part_b_out();
        } else {
            p.prev.z = prev_z; p.prev.C = cy;
        }
        p.W = W; p.ldw = cy; p.add = add; p.out = out;
        p.win_off = win_off; p.Q = s.Q; p.n_slots = s.n_slots; p.max_rows = s.max_rows; p.rows_hint = s.R;
        p.blocks_per_slot = pw_bwd_blocks(s.Q, s.n_slots, s.max_rows);
        const int nblk = p.blocks_per_slot * s.n_slots;
        const size_t need = align_up((size_t)nblk * g.C * cy, 64);
        const bool defer = n_deferred < REDUCE_MULTI_MAX && wpart_used + need <= wpart_cap;
        float *part = b.wpart + (defer ? wpart_used : 0);
        if (!defer) TRY(flush_deferred());                       // the region at offset 0 is about to be overwritten
        p.dWpart = part;
        TRY(pw_bwd_fused(p, st));
        if (defer) {
            deferred[n_deferred++] = ReduceItem{part, nblk, (long)g.C * cy, g.C, cy, cy, dW, cy};
            wpart_used += need;
        } else {
            TRY(reduce_windows(part, nblk, (long)g.C * cy, g.C, cy, cy, dW, cy, 0, st));
        }
        if (prev_bn >= 0) TRY(finalize_prev(prev_bn, cy, nblk, 1, p.part_a, p.part_b));
        return AMPNET_OK;
    }
    // the partials (pa, pb) of layer prev_bn are complete: its constants are owed to the next consumer
    int finalize_prev(int prev_bn, int cp, int part_Q, int chunks, const float *pa, const float *pb) const
    {
        TRY(settle());
        owed.open = true; owed.bn = prev_bn; owed.C = cp; owed.parts = part_Q > 0 ? part_Q : s.Q * chunks; owed.chunks = part_Q > 0 ? chunks : chunks;
        owed.pa = pa; owed.pb = pb;
        if (part_Q <= 0 || chunks != 1) return settle();      // per-(window, chunk) partials of the unfused path: finalize now
        return AMPNET_OK;
    }
    // Backward of a 128 -> 256 layer followed by BatchNorm + ReLU + MaxPool, WITHOUT its [rows, 256] output.
    // With z = W a, dz = P1 dy + P2 z + P3 and dy non-zero only on the argmax rows (kernels.h, "backward of a max-pooled
    // layer"): the data gradient is a 128 -> 128 GEMM with per-slot weights G = W^T diag(P2) W plus c0 = P3 W plus a few
    // scattered rows, the weight gradient needs only the per-slot Gram matrix of the 128-channel input.
    int pooled_layer(const float *d_pooled, int slot_major, const int *arg, const float *zext, int bn, const float *W, float *dW,
                     const float *z_prev, int prev_bn, float *dy_out) const
    {
        PoolBwd p;
        p.d_pooled = d_pooled; p.slot_major = slot_major; p.arg = arg; p.zext = zext;
        p.scale = f.bn[bn].scale; p.shift = f.bn[bn].shift; p.mean = f.bn[bn].mean; p.invstd = f.bn[bn].invstd;
        p.win_off = win_off; p.Q = s.Q; p.n_slots = s.n_slots; p.C = 256;
        p.dpm = b.dpm; p.P1 = b.bn[bn].P1; p.P2 = b.bn[bn].P2; p.P3 = b.bn[bn].P3; p.slot_ab = b.bn[bn].slot_ab;
        TRY(pool_bwd(p, st));
        if (!fused) {
            SparseRows sr;
            sr.arg = arg; sr.dpm = b.dpm; sr.slot_major = slot_major; sr.P1 = b.bn[bn].P1; sr.W = W;
            sr.Q = s.Q; sr.n_slots = s.n_slots; sr.srows = b.srows; sr.srow_row = b.srow_row; sr.srow_cnt = b.srow_cnt;
            TRY(sparse_rows(sr, st));
        }
        TRY(slot_mats(W, b.bn[bn].P2, b.bn[bn].P3, s.n_slots, 256, 128, b.Gm, b.c0, st));
        PooledWgrad pw;
        pw.W = W; pw.P1 = b.bn[bn].P1; pw.P2 = b.bn[bn].P2; pw.P3 = b.bn[bn].P3; pw.gram = b.gram; pw.asum = b.asum;
        pw.arg = arg; pw.dpm = b.dpm; pw.slot_major = slot_major;
        pw.z_prev = z_prev; pw.s_prev = f.bn[prev_bn].scale; pw.t_prev = f.bn[prev_bn].shift; pw.z_bf16 = zb ? 1 : 0;
        pw.Q = s.Q; pw.n_slots = s.n_slots; pw.dW = dW;
        if (s.n_slots <= 10) pw.wgram = b.wgram;
        SparseFix sf;
        sf.srows = b.srows; sf.srow_row = b.srow_row; sf.srow_cnt = b.srow_cnt; sf.z_prev = z_prev;
        sf.s_prev = f.bn[prev_bn].scale; sf.t_prev = f.bn[prev_bn].shift; sf.mean_prev = f.bn[prev_bn].mean; sf.invstd_prev = f.bn[prev_bn].invstd;
        sf.Q = s.Q; sf.n_slots = s.n_slots; sf.out = dy_out;
        if (fused) {
            // one pass over z_prev: per-slot Gram matrix + column sums of a = relu(bn_prev(z_prev)) AND
            // dy_prev = (a G[slot] + c0[slot]) masked by the previous layer's ReLU, with its BatchNorm-backward sums
            PwBwd p;
            p.g.z = z_prev; p.g.C = 128; p.g.act = 1; p.g.P2 = f.bn[prev_bn].scale; p.g.P3 = f.bn[prev_bn].shift; p.g.z_bf16 = zb ? 1 : 0;
            p.prev = act(z_prev, prev_bn, 128);
            p.prev_mean = f.bn[prev_bn].mean; p.prev_invstd = f.bn[prev_bn].invstd;
            p.W = b.Gm; p.ldw = 128; p.w_slot_stride = 128 * 128; p.bias_slot = b.c0;
            p.blocks_per_slot = pw_bwd_blocks(s.Q, s.n_slots, s.max_rows);
            const int nblk = p.blocks_per_slot * s.n_slots;
            if (wpart_used + (size_t)nblk * 128 * 128 > wpart_cap) TRY(flush_deferred());       // the Gram partials go behind the deferred regions
            float *gpart = b.wpart + wpart_used;
            TRY(settle());                    // (nothing is owed at the points this is called from; cheap insurance)
            p.out = dy_out; p.dWpart = gpart; p.dbpart = b.dbpart; p.part_a = part_a_out(); p.part_b = part_b_out();
            p.win_off = win_off; p.Q = s.Q; p.n_slots = s.n_slots; p.max_rows = s.max_rows; p.rows_hint = s.R;
            TRY(pw_bwd_fused(p, st));
            TRY(reduce_slots2(gpart, 128 * 128, b.gram, b.dbpart, 128, b.asum, nblk, 1, s.n_slots, st));
            TRY(pooled_wgrad(pw, st));
            // the scattered rows: added after the dense part, their share of the sums goes behind the workgroup partials
            SparseScatter ss;
            ss.arg = arg; ss.dpm = b.dpm; ss.slot_major = slot_major; ss.P1 = b.bn[bn].P1; ss.W = W;
            ss.z_prev = z_prev; ss.s_prev = f.bn[prev_bn].scale; ss.t_prev = f.bn[prev_bn].shift; ss.z_bf16 = zb ? 1 : 0;
            ss.mean_prev = f.bn[prev_bn].mean; ss.invstd_prev = f.bn[prev_bn].invstd;
            ss.Q = s.Q; ss.n_slots = s.n_slots; ss.out = dy_out;
            ss.part_a = p.part_a + (size_t)nblk * 128; ss.part_b = p.part_b + (size_t)nblk * 128;
            ss.part_chunks = 1; ss.slot_idx = 0;
            TRY(sparse_scatter(ss, st));
            return finalize_prev(prev_bn, 128, nblk + s.Q, 1, p.part_a, p.part_b);
        }
        // Gram and column sums of a = relu(bn_prev(z_prev)) per slot, then dW
        {
            PwWgrad w;
            w.x.z = z_prev; w.x.C = 128; w.x.act = 1; w.x.P2 = f.bn[prev_bn].scale; w.x.P3 = f.bn[prev_bn].shift;
            w.y = act(z_prev, prev_bn, 128);
            w.dWpart = b.wpart; w.ldp = 128; w.dbpart = b.dbpart;
            w.win_off = win_off; w.Q = s.Q; w.n_slots = s.n_slots; w.rows_hint = s.R;
            w.chunk_rows = WG_CHUNK_ROWS; w.chunks = wg_chunks(s);
            TRY(pw_wgrad(w, st));
            TRY(reduce_slots(b.wpart, s.Q, w.chunks, s.n_slots, 128 * 128, b.gram, st));
            TRY(reduce_slots(b.dbpart, s.Q, w.chunks, s.n_slots, 128, b.asum, st));
            TRY(pooled_wgrad(pw, st));
        }
        // data gradient: dy_prev = (a G[slot] + c0[slot] + scattered rows) masked by the previous layer's ReLU
        PwDgrad d;
        d.g.z = z_prev; d.g.C = 128; d.g.act = 1; d.g.P2 = f.bn[prev_bn].scale; d.g.P3 = f.bn[prev_bn].shift;
        d.W = b.Gm; d.ldw = 128; d.w_slot_stride = 128 * 128; d.bias_slot = b.c0;
        d.prev = act(z_prev, prev_bn, 128);
        d.prev_mean = f.bn[prev_bn].mean; d.prev_invstd = f.bn[prev_bn].invstd;
        d.part_a = f.part_sum; d.part_b = f.part_sq; d.part_chunks = s.chunks + 1;
        d.out = dy_out; d.cp = 128;
        d.win_off = win_off; d.Q = s.Q; d.n_slots = s.n_slots; d.chunk_rows = s.chunk_rows; d.chunks = s.chunks; d.rows_hint = s.R;
        TRY(pw_dgrad(d, st));
        // the scattered rows: added after the dense part, with their share of the BatchNorm-backward sums in an extra slot
        sf.part_a = f.part_sum; sf.part_b = f.part_sq;
        sf.part_chunks = s.chunks + 1; sf.slot_idx = s.chunks;
        TRY(sparse_fix(sf, st));
        return finalize_prev(prev_bn, 128, 0, s.chunks + 1, f.part_sum, f.part_sq);
    }

    // T-Net FC head backward: g3 [Q, kk] (slot-major rows) -> parameter grads, d_pool [Q, 256]
    int tnet_fc_bwd(int pbase, int bn0, const float *g3, int kk, const float *pooled, const float *zf1, const float *zf2) const
    {
        const int Q = s.Q, per = s.fc_rows, ns = s.n_slots;
        // the activations a1 = relu(bn_4(zf1)), a2 = relu(bn_5(zf2)) in ONE launch (fusing them into the GEMMs' operand loads was measured:
        // + 8 .. 50 us per launch, the loads of a latency-bound kernel tripled); fc_3's bias gradient rides inside its weight-gradient problem
        TRY(fc_act_pair(zf1, f.bn[bn0 + 3].scale, f.bn[bn0 + 3].shift, 256, b.a1, zf2, f.bn[bn0 + 4].scale, f.bn[bn0 + 4].shift, 128, b.a2, Q, per, st));
        LinBwdOpt o3;
        o3.db = G[pbase + TP_FC3_B];
        // fc_3: z3 = a2 W3^T + b3
        static const bool ksplit = [] { const char *v = getenv("AMPNET_FC3_KSPLIT"); return !(v && v[0] == '0'); }();
        if (kk >= 1024 && ksplit)
            TRY(sgemm_linear_bwd_ksplit(Q, kk, 128, g3, kk, b.a2, 128, P[pbase + TP_FC3_W], 128, G[pbase + TP_FC3_W], 128, b.da2, 128, b.fc_split, 8, st, o3));
        else
            TRY(sgemm_linear_bwd(Q, kk, 128, g3, kk, b.a2, 128, P[pbase + TP_FC3_W], 128, G[pbase + TP_FC3_W], 128, b.da2, 128, st, o3));
        TRY(fc_bn_bwd(b.da2, zf2, f.bn[bn0 + 4].scale, f.bn[bn0 + 4].shift, f.bn[bn0 + 4].mean, f.bn[bn0 + 4].invstd, ns, per, 128, b.g2,
                      b.bn[bn0 + 4].slot_ab, st));
        // fc_2
        TRY(sgemm_linear_bwd(Q, 128, 256, b.g2, 128, b.a1, 256, P[pbase + TP_FC2], 256, G[pbase + TP_FC2], 256, b.da1, 256, st));
        TRY(fc_bn_bwd(b.da1, zf1, f.bn[bn0 + 3].scale, f.bn[bn0 + 3].shift, f.bn[bn0 + 3].mean, f.bn[bn0 + 3].invstd, ns, per, 256, b.g1,
                      b.bn[bn0 + 3].slot_ab, st));
        // fc_1 on the pooled features
        TRY(sgemm_linear_bwd(Q, 256, 256, b.g1, 256, pooled, 256, P[pbase + TP_FC1], 256, G[pbase + TP_FC1], 256, b.d_pool, 256, st));
        return AMPNET_OK;
    }
};

}  // namespace
}  // namespace ampnet

using namespace ampnet;

extern "C" size_t ampnet_encoder_bwd_workspace_bytes(int Q, int n_slots, int total_rows, int max_rows)
{
    if (Q < 1 || n_slots < 1 || total_rows < 1 || max_rows < 1) return 0;
    EncBwdWs w;
    enc_bwd_carve(enc_shape(Q, n_slots, total_rows, max_rows, 1), nullptr, w);
    return w.bytes;
}

extern "C" int ampnet_encoder_bwd_f32(const float *const *params_host, float *const *grads_host, const float *x,
                                      const int32_t *win_off, int Q, int n_slots, int total_rows, int max_rows,
                                      const float *local, const float *d_local, const float *d_global, const float *d_feat_T, const float *feat_T,
                                      void *fwd_workspace, size_t fwd_workspace_bytes, void *bwd_workspace,
                                      size_t bwd_workspace_bytes, void *stream)
{
    AMPNET_REQUIRE(params_host && grads_host && x && win_off && local && d_global && feat_T && fwd_workspace && bwd_workspace, "ampnet_encoder_bwd_f32: null pointer");
    AMPNET_REQUIRE(Q >= 1 && n_slots >= 1 && Q % n_slots == 0, "ampnet_encoder_bwd_f32: Q=%d n_slots=%d", Q, n_slots);
    TRY(ws_tag_check(fwd_workspace, "ampnet_encoder_bwd_f32"));
    EncBwd e;
    e.st = (hipStream_t)stream;
    e.s = enc_shape(Q, n_slots, total_rows, max_rows, 1);
    enc_carve(e.s, fwd_workspace, e.f);
    enc_bwd_carve(e.s, bwd_workspace, e.b);
    if (e.f.bytes > fwd_workspace_bytes) return fail(AMPNET_E_WORKSPACE, "ampnet_encoder_bwd_f32: forward workspace %zu B < %zu B", fwd_workspace_bytes, e.f.bytes);
    if (e.b.bytes > bwd_workspace_bytes) return fail(AMPNET_E_WORKSPACE, "ampnet_encoder_bwd_f32: backward workspace %zu B < %zu B", bwd_workspace_bytes, e.b.bytes);
    e.P = params_host;
    e.G = grads_host;
    e.win_off = win_off;
    {
        const char *env = getenv("AMPNET_FUSED_BWD");
        e.fused = !(env && env[0] == '0');
    }
    e.zb = z_storage_bf16();
    AMPNET_REQUIRE(!e.zb || e.fused, "ampnet_encoder_bwd_f32: bf16 activation storage needs the fused backward");
    {
        const char *env = getenv("AMPNET_BWD_FIN_IN_KERNEL");
        e.in_kernel_fin = e.fused && !bwd_operands_bf16() && !sync_bn_on() && (long)max_rows * Q == (long)total_rows && !(env && env[0] == '0');
    }
    e.wpart_cap = e.fused ? ((size_t)Q * (size_t)wg_chunks(e.s) + 320 + n_slots) * 256 * 128 : 0;      // floats in b.wpart (enc_bwd_carve)
    const float *const *P = params_host;
    float *const *G = grads_host;
    for (int i = 0; i < 5; ++i) {
        e.gamma[BN_T1 + i] = P[EP_IT + TP_BN1_W + 2 * i];
        e.gamma[BN_F1 + i] = P[EP_FT + TP_BN1_W + 2 * i];
    }
    {
        const int ids[6] = {BN_C1, BN_C2, BN_C3, BN_C4, BN_C5, BN_C6};
        for (int i = 0; i < 6; ++i) e.gamma[ids[i]] = P[EP_BN1_W + 2 * i];
    }
    const EncWs &f = e.f;
    const EncBwdWs &b = e.b;
    hipStream_t st = e.st;

    // ---- conv_6 .. conv_3 --------------------------------------------------------------------------
    TRY(e.pooled_layer(d_global, 0, f.arg_c, f.zext_c, BN_C6, P[EP_CONV6], G[EP_CONV6], f.z_c5, BN_C5, b.dyA));
    {
        const GradSrc g5 = e.dense(b.dyA, f.z_c5, BN_C5, 128);
        TRY(e.layer_bwd(g5, P[EP_CONV5], G[EP_CONV5], f.z_c4, BN_C4, 128, nullptr, b.dyB));
    }
    {
        const GradSrc g4 = e.dense(b.dyB, f.z_c4, BN_C4, 128);
        TRY(e.layer_bwd(g4, P[EP_CONV4], G[EP_CONV4], f.z_c3, BN_C3, 64, nullptr, b.dyA));
    }
    {
        // conv_3 reads `local` (the torch.bmm output, not activated); the head's gradient d_local joins here
        const GradSrc g3 = e.dense(b.dyA, f.z_c3, BN_C3, 64);
        TRY(e.layer_bwd(g3, P[EP_CONV3], G[EP_CONV3], local, -1, 64, d_local, b.d_local));
    }
    // ---- local = h x T64[window]: dT64 (per window, no reduction) and d_h ---------------------------------
    {
        // items (256-row chunks) per workgroup: a divisor of the chunks per window (a workgroup stays inside one window), at most
        // s.chunks partials per window (buffer size), and the fullest last round of 256 workgroups
        const int cpw = cdiv(max_rows, pw_bwd_item_rows());
        int ipb = 0;
        double best = -1.0;
        for (int d = 1; d <= cpw && d <= 8; ++d) {
            if (cpw % d || cpw / d > e.s.chunks) continue;
            const long blocks = (long)Q * (cpw / d);
            const double eff = (double)blocks / (double)(((blocks + 255) / 256) * 256);
            if (eff >= best) {
                best = eff;
                ipb = d;
            }
        }
        const int bpw = ipb ? cpw / ipb : 0;                        // workgroups per window
        if (e.fused && ipb > 0) {
            // one pass over (d_local, z_c2): dT^T[j][k] = sum_rows d_local[row][j] h[row][k] per window AND
            // d_h[row][k] = sum_j d_local[row][j] T[k][j] (masked by conv_2's ReLU here already: the mask is idempotent and the
            // feature T-Net's conv_1 backward applies it again after adding its own term).  g = dy via identity constants.
            TRY(e.settle());
            const float *ones64 = nullptr, *zeros64 = nullptr;
            TRY(identity_constants(n_slots, &ones64, &zeros64, st));
            PwBwd p;
            p.g.dy = b.d_local; p.g.z = b.d_local; p.g.C = 64; p.g.P1 = ones64; p.g.P2 = zeros64; p.g.P3 = zeros64;
            p.prev = e.act(f.z_c2, BN_C2, 64);
            p.W = feat_T; p.ldw = 64; p.w_win_stride = 4096; p.perwin_slot_major = 1;
            p.out = b.d_h; p.dWpart = b.dT64t;
            p.win_off = win_off; p.Q = Q; p.n_slots = n_slots; p.max_rows = max_rows; p.rows_hint = total_rows;
            p.items_per_block = ipb; p.blocks_per_slot = (Q / n_slots) * bpw;
            TRY(pw_bwd_fused(p, st));
            TRY(transpose64_slot_major(b.dT64t, b.dT64, Q, n_slots, bpw, 1, d_feat_T, st));
        } else {
            PwWgrad w;                                   // dT^T[j][k] = sum_rows d_local[row][j] * h[row][k]
            w.x = e.dense(b.d_local, nullptr, -1, 64);
            w.y = e.act(f.z_c2, BN_C2, 64);
            w.dWpart = b.dT64t; w.ldp = 64;
            w.win_off = win_off; w.Q = Q; w.n_slots = n_slots; w.rows_hint = total_rows;
            w.chunk_rows = e.s.chunk_rows; w.chunks = e.s.chunks;      // several workgroups per window: no half-empty last round
            TRY(pw_wgrad(w, st));
            // window q's matrix (sum of its chunk partials) belongs at the slot-major row the forward used for feat_T
            TRY(transpose64_slot_major(b.dT64t, b.dT64, Q, n_slots, e.s.chunks, 0, d_feat_T, st));
            PwDgrad d;                                   // d_h[row][k] = sum_j d_local[row][j] * T[k][j]
            d.g = e.dense(b.d_local, nullptr, -1, 64);
            d.W = feat_T; d.w_win_stride = 4096; d.perwin_slot_major = 1;
            d.out = b.d_h; d.cp = 64;
            d.win_off = win_off; d.Q = Q; d.n_slots = n_slots; d.chunk_rows = e.s.chunk_rows; d.chunks = e.s.chunks; d.rows_hint = total_rows;
            TRY(pw_dgrad(d, st));
        }
    }
    // ---- feature T-Net ---------------------------------------------------------------------------------------
    TRY(e.tnet_fc_bwd(EP_FT, BN_F1, b.dT64, 4096, f.pool_f, f.z_ff1, f.z_ff2));
    TRY(e.pooled_layer(b.d_pool, 1, f.arg_f, f.zext_f, BN_F3, P[EP_FT + TP_CONV3], G[EP_FT + TP_CONV3], f.z_f2, BN_F2, b.dyA));
    {
        const GradSrc g = e.dense(b.dyA, f.z_f2, BN_F2, 128);
        TRY(e.layer_bwd(g, P[EP_FT + TP_CONV2], G[EP_FT + TP_CONV2], f.z_f1, BN_F1, 64, nullptr, b.dyB));
    }
    {
        const GradSrc g = e.dense(b.dyB, f.z_f1, BN_F1, 64);
        TRY(e.layer_bwd(g, P[EP_FT + TP_CONV1], G[EP_FT + TP_CONV1], f.z_c2, BN_C2, 64, b.d_h, b.dyA));      // + the bmm path into h
    }
    // ---- conv_2, conv_1 -----------------------------------------------------------------------------------------
    {
        const GradSrc g = e.dense(b.dyA, f.z_c2, BN_C2, 64);
        TRY(e.layer_bwd(g, P[EP_CONV2], G[EP_CONV2], f.z_c1, BN_C1, 64, nullptr, b.dyB));
    }
    {
        PwInputWgrad w;
        w.x = x; w.dy = b.dyB; w.W = P[EP_CONV1]; w.T = f.T3; w.mode = 1; w.perwin_slot_major = 1;
        w.P1 = b.bn[BN_C1].P1; w.P2 = b.bn[BN_C1].P2; w.P3 = b.bn[BN_C1].P3;
        TRY(e.input_fin(w, BN_C1));            // bn_1's constants: in the kernel when they are still owed, else from memory
        w.dWeff = b.dWeff; w.win_off = win_off; w.Q = Q; w.n_slots = n_slots;
        TRY(pw_input_wgrad(w, st));
        TRY(input_param_grads(b.dWeff, P[EP_CONV1], f.T3, Q, n_slots, 1, 1, G[EP_CONV1], b.dT3, st));
    }
    // ---- input T-Net ----------------------------------------------------------------------------------------------
    TRY(e.tnet_fc_bwd(EP_IT, BN_T1, b.dT3, 9, f.pool_t, f.z_tf1, f.z_tf2));
    TRY(e.pooled_layer(b.d_pool, 1, f.arg_t, f.zext_t, BN_T3, P[EP_IT + TP_CONV3], G[EP_IT + TP_CONV3], f.z_t2, BN_T2, b.dyA));
    {
        const GradSrc g = e.dense(b.dyA, f.z_t2, BN_T2, 128);
        TRY(e.layer_bwd(g, P[EP_IT + TP_CONV2], G[EP_IT + TP_CONV2], f.z_t1, BN_T1, 64, nullptr, b.dyB));
    }
    {
        PwInputWgrad w;
        w.x = x; w.dy = b.dyB; w.W = P[EP_IT + TP_CONV1]; w.mode = 0;
        w.P1 = b.bn[BN_T1].P1; w.P2 = b.bn[BN_T1].P2; w.P3 = b.bn[BN_T1].P3;
        TRY(e.input_fin(w, BN_T1));
        w.dWeff = b.dWeff; w.win_off = win_off; w.Q = Q; w.n_slots = n_slots;
        TRY(pw_input_wgrad(w, st));
        TRY(input_param_grads(b.dWeff, P[EP_IT + TP_CONV1], nullptr, Q, n_slots, 0, 0, G[EP_IT + TP_CONV1], nullptr, st));
    }
    TRY(e.settle());
    TRY(e.flush_deferred());           // every fused layer's weight gradient: one reduction launch
    // ---- BatchNorm weight / bias gradients: sums over the slots of (sum dy, sum dy * zhat) -----------------------------
    {
        const int ids[16] = {BN_T1, BN_T2, BN_T3, BN_T4, BN_T5, BN_F1, BN_F2, BN_F3, BN_F4, BN_F5, BN_C1, BN_C2, BN_C3, BN_C4, BN_C5, BN_C6};
        BnGradItem items[16];
        for (int i = 0; i < 16; ++i) {
            const int id = ids[i];
            const int gw = id <= BN_T5 ? EP_IT + TP_BN1_W + 2 * (id - BN_T1)
                                       : (id >= BN_F1 && id <= BN_F5 ? EP_FT + TP_BN1_W + 2 * (id - BN_F1)
                                                                     : (id <= BN_C2 ? EP_BN1_W + 2 * (id - BN_C1) : EP_BN3_W + 2 * (id - BN_C3)));
            items[i] = {b.bn[id].slot_ab, G[gw], G[gw + 1], f.bn[id].C, n_slots};
        }
        TRY(bn_param_grads(items, 16, st));
    }
    return AMPNET_OK;
}
