// baseline_train.hip -- train-mode forward and backward of the baseline single-window PointNet segmentation model (SURVEY row a12,
// BASELINE.json config 1): pointNet/model/pointnet.py (:6-44 TransformationNet, :47-97 BasePointNet, :128-154 SegmentationPointNet;
// 1024-d, convolutions with bias, T-Net on x[:, :, :3]) and pointNet/model/light_pointnet_256.py (256-d, no conv / fc bias, T-Net on
// x[:, :, :2]), as trained by pointNet/baseline/train_segmentation.py:274-328 (loss.backward() through the whole module).
//
// Config 1 is the reference's plumbing case ([4, 512, 9]); like the eval forward (baseline.hip) this path is built for parity, not
// speed: every layer is sgemm_small + elementwise passes over materialised activations (a tape in the caller's workspace):
//     z = A W^T + b (+ per-window addend)   ->   batch statistics (double, fixed order) + running update   ->   a = relu(bn(z))
// and the backward walks the tape: BatchNorm/ReLU backward (two column sums + one elementwise pass), dW = dz^T A, db = sum dz,
// dA = dz W, MaxPool1d backward as a scatter to the argmax rows, the two torch.bmm, the T-Net identity.
#include "kernels.h"
#include "bwd_misc.h"

namespace ampnet {
namespace {

struct TLayer {                // device pointers of one layer: weight, bias, BatchNorm weight / bias / running mean / running var
    const float *W, *b, *g, *be;
    float *rm, *rv;
};
struct TGrad {                 // gradients of the same (nullptr where the layer has no such parameter)
    float *dW, *db, *dg, *dbe;
};

struct Dims {
    int k, G, F1, F2, H1, H2, H3;
    int C1, C2;                // ClassificationPointNet: fc_1 G -> C1, fc_2 C1 -> C2 (pointnet.py:109-111, light_pointnet_256.py:109-111)
};

bool dims_for(int variant, Dims *d)
{
    if (variant == 0) *d = Dims{3, 1024, 512, 256, 512, 256, 128, 512, 256};
    else if (variant == 1) *d = Dims{2, 256, 256, 128, 256, 128, 64, 128, 64};
    else return false;
    return true;
}

// ---- elementwise / reduction kernels -------------------------------------------------------------------------------------------
// z[row][c] += bias[c] + add[row / per][c]
__global__ void bt_bias_kernel(float *__restrict__ z, size_t n, int C, const float *__restrict__ bias, const float *__restrict__ add, int per)
{
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C);
    float v = z[i];
    if (bias) v += bias[c];
    if (add) v += add[(i / C / per) * C + c];
    z[i] = v;
}

// batch statistics of the columns of z [M, C] (biased variance, eps inside invstd) + the running update of nn.BatchNorm1d
// (momentum 0.1, unbiased variance).  block = 64 columns x 4 row groups, double accumulation, fixed order.
__global__ __launch_bounds__(256) void bt_stats_kernel(const float *__restrict__ z, int M, int C, float eps, float momentum, float *__restrict__ mean,
                                                       float *__restrict__ invstd, float *__restrict__ rm, float *__restrict__ rv)
{
    __shared__ double red[4][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    double s = 0.0;
    if (c < C)
        for (int r = g; r < M; r += 4) s += (double)z[(size_t)r * C + c];
    red[g][cl] = s;
    __syncthreads();
    const double mu = ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) / M;
    __syncthreads();
    double q = 0.0;
    if (c < C)
        for (int r = g; r < M; r += 4) {
            const double d = (double)z[(size_t)r * C + c] - mu;
            q += d * d;
        }
    red[g][cl] = q;
    __syncthreads();
    if (g == 0 && c < C) {
        const double m2 = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
        const double var = m2 / M;
        mean[c] = (float)mu;
        invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        rm[c] = (1.0f - momentum) * rm[c] + momentum * (float)mu;
        rv[c] = (1.0f - momentum) * rv[c] + momentum * (float)(M > 1 ? m2 / (M - 1) : m2);
    }
}

// a = relu((z - mean) * invstd * gamma + beta)
__global__ void bt_bn_act_kernel(const float *__restrict__ z, size_t n, int C, const float *__restrict__ mean, const float *__restrict__ invstd,
                                 const float *__restrict__ g, const float *__restrict__ be, float *__restrict__ a)
{
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C);
    a[i] = fmaxf(fmaf((z[i] - mean[c]) * invstd[c], g[c], be[c]), 0.f);
}

// column sums of the BatchNorm + ReLU backward: dbeta[c] = sum_r g, dgamma[c] = sum_r g * zhat with g = da * (a > 0)
__global__ __launch_bounds__(256) void bt_bn_bwd_sums_kernel(const float *__restrict__ da, const float *__restrict__ a, const float *__restrict__ z, int M,
                                                             int C, const float *__restrict__ mean, const float *__restrict__ invstd,
                                                             float *__restrict__ dg, float *__restrict__ dbe)
{
    __shared__ double ra[4][64], rb[4][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    double sa = 0.0, sb = 0.0;
    if (c < C) {
        const float mu = mean[c], is = invstd[c];
        for (int r = g; r < M; r += 4) {
            const size_t o = (size_t)r * C + c;
            const float gv = a[o] > 0.f ? da[o] : 0.f;
            sa += (double)gv;
            sb += (double)gv * (double)((z[o] - mu) * is);
        }
    }
    ra[g][cl] = sa;
    rb[g][cl] = sb;
    __syncthreads();
    if (g == 0 && c < C) {
        dbe[c] = (float)((ra[0][cl] + ra[1][cl]) + (ra[2][cl] + ra[3][cl]));
        dg[c] = (float)((rb[0][cl] + rb[1][cl]) + (rb[2][cl] + rb[3][cl]));
    }
}

// dz = gamma * invstd * (g - dbeta / M - zhat * dgamma / M), in place over da
__global__ void bt_bn_bwd_apply_kernel(float *__restrict__ da, const float *__restrict__ a, const float *__restrict__ z, size_t n, int M, int C,
                                       const float *__restrict__ mean, const float *__restrict__ invstd, const float *__restrict__ gamma,
                                       const float *__restrict__ dg, const float *__restrict__ dbe)
{
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C);
    const float gv = a[i] > 0.f ? da[i] : 0.f;
    const float zh = (z[i] - mean[c]) * invstd[c];
    da[i] = gamma[c] * invstd[c] * (gv - dbe[c] / M - zh * dg[c] / M);
}

// MaxPool1d(num_points): [B, N, C] -> value + argmax row (first maximum)
__global__ __launch_bounds__(256) void bt_rowmax_kernel(const float *__restrict__ a, int N, int C, float *__restrict__ out, int *__restrict__ arg)
{
    __shared__ float rv[4][64];
    __shared__ int ri[4][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6, c = blockIdx.x * 64 + cl, b = blockIdx.y;
    float m = -__builtin_inff();
    int mi = 0;
    if (c < C)
        for (int r = g; r < N; r += 4) {
            const float v = a[((size_t)b * N + r) * C + c];
            if (v > m) {
                m = v;
                mi = r;
            }
        }
    rv[g][cl] = m;
    ri[g][cl] = mi;
    __syncthreads();
    if (g == 0 && c < C) {
        for (int k = 1; k < 4; ++k)
            if (rv[k][cl] > m || (rv[k][cl] == m && ri[k][cl] < mi)) {
                m = rv[k][cl];
                mi = ri[k][cl];
            }
        out[(size_t)b * C + c] = m;
        arg[(size_t)b * C + c] = mi;
    }
}

// d_a[b, arg[b, c], c] = d_pool[b, c] on a zeroed d_a
__global__ void bt_pool_scatter_kernel(const float *__restrict__ d_pool, const int *__restrict__ arg, int B, int N, int C, float *__restrict__ d_a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i % C;
    d_a[((size_t)b * N + arg[i]) * C + c] = d_pool[i];
}

// cat([x[:, :, :k] @ T, x[:, :, k:]], 2)
__global__ void bt_mix_kernel(const float *__restrict__ x, const float *__restrict__ T, int R, int N, int k, float *__restrict__ out)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= R) return;
    const float *t = T + (size_t)(row / N) * k * k;
    const float *xi = x + (size_t)row * 9;
    float *o = out + (size_t)row * 9;
    for (int j = 0; j < k; ++j) {
        float acc = 0.f;
        for (int i = 0; i < k; ++i) acc = fmaf(xi[i], t[i * k + j], acc);
        o[j] = acc;
    }
    for (int j = k; j < 9; ++j) o[j] = xi[j];
}

// dT[b][i][j] = sum_rows x[row][i] * d_xin[row][j]   (i, j < k); one block per window
__global__ __launch_bounds__(256) void bt_mix_bwd_kernel(const float *__restrict__ x, const float *__restrict__ d_xin, int N, int k, float *__restrict__ dT)
{
    __shared__ double red[256];
    const int b = blockIdx.x;
    for (int e = 0; e < k * k; ++e) {
        const int i = e / k, j = e % k;
        double s = 0.0;
        for (int r = threadIdx.x; r < N; r += 256) s += (double)x[((size_t)b * N + r) * 9 + i] * (double)d_xin[((size_t)b * N + r) * 9 + j];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
            __syncthreads();
        }
        if (threadIdx.x == 0) dT[(size_t)b * k * k + e] = (float)red[0];
        __syncthreads();
    }
}

// out[b][c] = sum over the N rows of window b of dz[row][c]
__global__ __launch_bounds__(256) void bt_segsum_kernel(const float *__restrict__ dz, int N, int C, float *__restrict__ out)
{
    __shared__ double red[4][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6, c = blockIdx.x * 64 + cl, b = blockIdx.y;
    double s = 0.0;
    if (c < C)
        for (int r = g; r < N; r += 4) s += (double)dz[((size_t)b * N + r) * C + c];
    red[g][cl] = s;
    __syncthreads();
    if (g == 0 && c < C) out[(size_t)b * C + c] = (float)((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]));
}

// z [B * N, C] <-> logits [B, C, N]
__global__ void bt_logits_kernel(const float *__restrict__ z, int B, int N, int C, float *__restrict__ logits, int to_rows)
{
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= (size_t)B * N * C) return;
    const int p = (int)(i % N), c = (int)((i / N) % C), b = (int)(i / N / C);
    if (to_rows) const_cast<float *>(z)[((size_t)b * N + p) * C + c] = logits[i];
    else logits[i] = z[((size_t)b * N + p) * C + c];
}

__global__ void bt_fill_kernel(float *__restrict__ p, size_t n, float v)
{
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ void bt_add_kernel(float *__restrict__ y, const float *__restrict__ x, size_t n)
{
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) y[i] += x[i];
}

// eval mode: the "batch statistics" of a layer are its running statistics
__global__ void bt_running_stats_kernel(const float *__restrict__ rm, const float *__restrict__ rv, int C, float eps, float *__restrict__ mean,
                                        float *__restrict__ invstd)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        mean[c] = rm[c];
        invstd[c] = 1.0f / sqrtf(rv[c] + eps);
    }
}

// nn.Dropout(p) on a [n] tensor: out = keep(i) ? a * 1 / (1 - p) : 0 with the package's counter hash (kernels.h: mix32; the oracle's keep_mask);
// the backward is the same map on the gradient
__global__ void bt_dropout_kernel(const float *__restrict__ a, size_t n, uint32_t base, uint32_t thr, float scale, float *__restrict__ out)
{
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = (mix32((uint32_t)i ^ base) >= thr) ? a[i] * scale : 0.f;
}

// F.log_softmax(z, dim=1) on [B, C] (C <= 64): one thread per row
__global__ void bt_log_softmax_kernel(const float *__restrict__ z, int B, int C, float *__restrict__ out)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float m = -__builtin_inff();
    for (int c = 0; c < C; ++c) m = fmaxf(m, z[(size_t)b * C + c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(z[(size_t)b * C + c] - m);
    const float lse = m + logf(s);
    for (int c = 0; c < C; ++c) out[(size_t)b * C + c] = z[(size_t)b * C + c] - lse;
}

// backward of log_softmax: dz = d_out - softmax(z) * sum_c d_out
__global__ void bt_log_softmax_bwd_kernel(const float *__restrict__ z, const float *__restrict__ d_out, int B, int C, float *__restrict__ dz)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float m = -__builtin_inff(), tot = 0.f;
    for (int c = 0; c < C; ++c) {
        m = fmaxf(m, z[(size_t)b * C + c]);
        tot += d_out[(size_t)b * C + c];
    }
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(z[(size_t)b * C + c] - m);
    for (int c = 0; c < C; ++c) dz[(size_t)b * C + c] = d_out[(size_t)b * C + c] - expf(z[(size_t)b * C + c] - m) / s * tot;
}

#define BT_TRY(expr)                      \
    do {                                  \
        int rc_ = (expr);                 \
        if (rc_ != AMPNET_OK) return rc_; \
    } while (0)

inline unsigned nb(size_t n) { return (unsigned)((n + 255) / 256); }

// ---- the tape ----------------------------------------------------------------------------------------------------------------------
struct Rec {                   // one linear (+ BatchNorm + ReLU) layer on the tape
    float *z, *a;              // [M, cout] pre-BatchNorm (with bias / addend) and activation; a == z for layers without BatchNorm
    float *mean, *invstd;      // [cout]
    int M, K, cout;
};

struct Ws {
    Rec L[AMPNET_POINTNET_LAYERS];
    float *xin, *pool_i, *pool_f, *pool_c, *T3, *T64, *local, *gb;
    float *a_drop;             // classification head: dropout(relu(bn_2(fc_2))) [B, C2]
    int n_layers;
    int *arg_i, *arg_f, *arg_c;
    // backward scratch
    float *dA, *dB, *d_local, *d_c2, *d_pool, *d_gb, *dT64, *dT3, *d_xin, *dgs, *dbs;
    size_t bytes;
};

// cls: the 20-layer classification model (17 base layers + fc_1, fc_2, fc_3 on the global feature) instead of the 21-layer segmentation model
void carve(const Dims &d, int B, int N, int C, void *base, Ws &w, bool cls = false)
{
    size_t off = 0;
    auto take = [&](size_t n_floats) {
        float *p = base ? reinterpret_cast<float *>(static_cast<char *>(base) + off) : nullptr;
        off += align_up(n_floats * sizeof(float), 256);
        return p;
    };
    const size_t R = (size_t)B * N;
    const int kk = d.k * d.k;
    int couts[AMPNET_POINTNET_LAYERS] = {64, 128, d.G, d.F1, d.F2, kk, 64, 128, d.G, d.F1, d.F2, 4096, 64, 64, 64, 128, d.G, d.H1, d.H2, d.H3, C};
    int Ks[AMPNET_POINTNET_LAYERS] = {d.k, 64, 128, d.G, d.F1, d.F2, 64, 64, 128, d.G, d.F1, d.F2, 9, 64, 64, 64, 128, 64, d.H1, d.H2, d.H3};
    bool point[AMPNET_POINTNET_LAYERS] = {1, 1, 1, 0, 0, 0, 1, 1, 1, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1};
    bool bn[AMPNET_POINTNET_LAYERS] = {1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1, 1, 0};
    w.n_layers = cls ? AMPNET_POINTNET_CLS_LAYERS : AMPNET_POINTNET_LAYERS;
    if (cls) {
        couts[17] = d.C1; couts[18] = d.C2; couts[19] = C;
        Ks[17] = d.G; Ks[18] = d.C1; Ks[19] = d.C2;
        point[17] = point[18] = point[19] = false;
        bn[17] = bn[18] = true; bn[19] = false;
    }
    int maxc = 64;
    for (int i = 0; i < w.n_layers; ++i) {
        Rec &r = w.L[i];
        r.M = point[i] ? (int)R : B;
        r.K = Ks[i];
        r.cout = couts[i];
        r.z = take((size_t)r.M * r.cout);
        r.a = bn[i] ? take((size_t)r.M * r.cout) : r.z;
        r.mean = take(r.cout);
        r.invstd = take(r.cout);
        if (point[i] && r.cout > maxc) maxc = r.cout;
    }
    w.xin = take(R * 9);
    w.pool_i = take((size_t)B * d.G); w.pool_f = take((size_t)B * d.G); w.pool_c = take((size_t)B * d.G);
    w.arg_i = reinterpret_cast<int *>(take((size_t)B * d.G));
    w.arg_f = reinterpret_cast<int *>(take((size_t)B * d.G));
    w.arg_c = reinterpret_cast<int *>(take((size_t)B * d.G));
    w.T3 = take((size_t)B * 9);
    w.T64 = take((size_t)B * 4096);
    w.local = take(R * 64);
    w.gb = take((size_t)B * d.H1);
    w.a_drop = take((size_t)B * d.C2);
    w.dA = take(R * maxc); w.dB = take(R * maxc);
    w.d_local = take(R * 64); w.d_c2 = take(R * 64);
    w.d_pool = take((size_t)B * d.G); w.d_gb = take((size_t)B * d.H1);
    w.dT64 = take((size_t)B * 4096); w.dT3 = take((size_t)B * 9);
    w.d_xin = take(R * 9);
    w.dgs = take(4096); w.dbs = take(4096);
    w.bytes = off;
}

struct Ctx {
    hipStream_t st;
    const TLayer *L;
    const TGrad *G;
    Ws *w;
    bool running = false;      // eval mode: BatchNorm with the running statistics (no update)
};

// forward of layer i: z = A[:, :K] W[:, w0 : w0 + K]^T + b (+ add[row / per]); statistics; activation
int fwd_layer(const Ctx &c, int i, const float *A, int lda, int ldw, int w0, const float *add, int per)
{
    const TLayer &l = c.L[i];
    Rec &r = c.w->L[i];
    BT_TRY(sgemm_small(0, 1, r.M, r.cout, r.K, A, lda, l.W + w0, ldw, r.z, r.cout, 0, c.st));
    const size_t n = (size_t)r.M * r.cout;
    if (l.b || add) hipLaunchKernelGGL(bt_bias_kernel, dim3(nb(n)), dim3(256), 0, c.st, r.z, n, r.cout, l.b, add, per > 0 ? per : 1);
    if (l.g) {
        if (c.running) hipLaunchKernelGGL(bt_running_stats_kernel, dim3(cdiv(r.cout, 256)), dim3(256), 0, c.st, l.rm, l.rv, r.cout, 1e-5f, r.mean, r.invstd);
        else hipLaunchKernelGGL(bt_stats_kernel, dim3(cdiv(r.cout, 64)), dim3(256), 0, c.st, r.z, r.M, r.cout, 1e-5f, 0.1f, r.mean, r.invstd, l.rm, l.rv);
        hipLaunchKernelGGL(bt_bn_act_kernel, dim3(nb(n)), dim3(256), 0, c.st, r.z, n, r.cout, r.mean, r.invstd, l.g, l.be, r.a);
    }
    return check_launch("baseline train layer forward");
}

// backward of layer i: `da` [M, cout] (overwritten with dz) -> parameter gradients; dA [M, K] (=, or += when accumulate) unless nullptr
int bwd_layer(const Ctx &c, int i, float *da, const float *A, int lda, int ldw, int w0, float *dA, int ld_dA, int accumulate)
{
    const TLayer &l = c.L[i];
    const TGrad &g = c.G[i];
    const Rec &r = c.w->L[i];
    const size_t n = (size_t)r.M * r.cout;
    if (l.g) {
        float *dg = g.dg ? g.dg : c.w->dgs, *dbe = g.dbe ? g.dbe : c.w->dbs;
        hipLaunchKernelGGL(bt_bn_bwd_sums_kernel, dim3(cdiv(r.cout, 64)), dim3(256), 0, c.st, da, r.a, r.z, r.M, r.cout, r.mean, r.invstd, dg, dbe);
        hipLaunchKernelGGL(bt_bn_bwd_apply_kernel, dim3(nb(n)), dim3(256), 0, c.st, da, r.a, r.z, n, r.M, r.cout, r.mean, r.invstd, l.g, dg, dbe);
    }
    if (g.dW) BT_TRY(sgemm_small(1, 0, r.cout, r.K, r.M, da, r.cout, A, lda, g.dW + w0, ldw, 0, c.st));
    if (g.db) BT_TRY(colsum(da, r.M, r.cout, g.db, c.st));
    if (dA) BT_TRY(sgemm_small(0, 0, r.M, r.K, r.cout, da, r.cout, l.W + w0, ldw, dA, ld_dA, accumulate, c.st));
    return check_launch("baseline train layer backward");
}

int tnet_fwd(const Ctx &c, const Dims &d, int base, const float *A, int lda, int k, int B, int N, float *pool, int *arg, float *T)
{
    Ws &w = *c.w;
    BT_TRY(fwd_layer(c, base + 0, A, lda, k, 0, nullptr, 0));
    BT_TRY(fwd_layer(c, base + 1, w.L[base + 0].a, 64, 64, 0, nullptr, 0));
    BT_TRY(fwd_layer(c, base + 2, w.L[base + 1].a, 128, 128, 0, nullptr, 0));
    hipLaunchKernelGGL(bt_rowmax_kernel, dim3(cdiv(d.G, 64), B), dim3(256), 0, c.st, w.L[base + 2].a, N, d.G, pool, arg);
    BT_TRY(fwd_layer(c, base + 3, pool, d.G, d.G, 0, nullptr, 0));
    BT_TRY(fwd_layer(c, base + 4, w.L[base + 3].a, d.F1, d.F1, 0, nullptr, 0));
    BT_TRY(fwd_layer(c, base + 5, w.L[base + 4].a, d.F2, d.F2, 0, nullptr, 0));
    if (hipMemcpyAsync(T, w.L[base + 5].z, (size_t)B * k * k * sizeof(float), hipMemcpyDeviceToDevice, c.st) != hipSuccess)
        return fail(AMPNET_E_LAUNCH, "baseline train: copy of a transform failed");
    return add_identity(T, B, k, c.st);
}

// dT [B, k * k] -> gradients of the T-Net's six layers; d_in (the T-Net's input activation gradient, [R, k_in]) accumulated when given
int tnet_bwd(const Ctx &c, const Dims &d, int base, float *dT, const float *A, int lda, int k, int B, int N, const float *pool, const int *arg, float *d_in,
             int ld_in)
{
    Ws &w = *c.w;
    const size_t R = (size_t)B * N;
    BT_TRY(bwd_layer(c, base + 5, dT, w.L[base + 4].a, d.F2, d.F2, 0, w.dA, d.F2, 0));                 // fc_3 (no BatchNorm): dT is dz
    BT_TRY(bwd_layer(c, base + 4, w.dA, w.L[base + 3].a, d.F1, d.F1, 0, w.dB, d.F1, 0));
    BT_TRY(bwd_layer(c, base + 3, w.dB, pool, d.G, d.G, 0, w.d_pool, d.G, 0));
    hipLaunchKernelGGL(bt_fill_kernel, dim3(nb(R * d.G)), dim3(256), 0, c.st, w.dA, R * d.G, 0.f);
    hipLaunchKernelGGL(bt_pool_scatter_kernel, dim3(nb((size_t)B * d.G)), dim3(256), 0, c.st, w.d_pool, arg, B, N, d.G, w.dA);
    BT_TRY(bwd_layer(c, base + 2, w.dA, w.L[base + 1].a, 128, 128, 0, w.dB, 128, 0));
    BT_TRY(bwd_layer(c, base + 1, w.dB, w.L[base + 0].a, 64, 64, 0, w.dA, 64, 0));
    return bwd_layer(c, base + 0, w.dA, A, lda, k, 0, d_in, ld_in, 1);
}

int read_tables(const float *const *layers_host, float *const *grads_host, TLayer *L, TGrad *G, int n_layers = AMPNET_POINTNET_LAYERS)
{
    for (int i = 0; i < n_layers; ++i) {
        const float *const *p = layers_host + 6 * i;
        L[i] = TLayer{p[0], p[1], p[2], p[3], const_cast<float *>(p[4]), const_cast<float *>(p[5])};
        AMPNET_REQUIRE(L[i].W, "baseline PointNet: layer %d has no weight", i);
        AMPNET_REQUIRE(!L[i].g || (L[i].be && L[i].rm && L[i].rv), "baseline PointNet: layer %d: incomplete BatchNorm", i);
        if (grads_host) {
            float *const *q = grads_host + 4 * i;
            G[i] = TGrad{q[0], q[1], q[2], q[3]};
            AMPNET_REQUIRE(G[i].dW, "baseline PointNet: layer %d has no weight gradient", i);
        }
    }
    return AMPNET_OK;
}

// layers 0 .. 16 of both models (BasePointNet.forward up to the max-pool, pointnet.py:66-90): leaves local, pool_c / arg_c, T64 on the tape
int base_fwd(const Ctx &c, const Dims &d, const float *x, int B, int N)
{
    Ws &w = *c.w;
    const int R = B * N, k = d.k;
    BT_TRY(tnet_fwd(c, d, 0, x, 9, k, B, N, w.pool_i, w.arg_i, w.T3));
    hipLaunchKernelGGL(bt_mix_kernel, dim3(cdiv(R, 256)), dim3(256), 0, c.st, x, w.T3, R, N, k, w.xin);
    BT_TRY(fwd_layer(c, 12, w.xin, 9, 9, 0, nullptr, 0));
    BT_TRY(fwd_layer(c, 13, w.L[12].a, 64, 64, 0, nullptr, 0));
    BT_TRY(tnet_fwd(c, d, 6, w.L[13].a, 64, 64, B, N, w.pool_f, w.arg_f, w.T64));
    for (int b = 0; b < B; ++b)
        BT_TRY(sgemm_small(0, 0, N, 64, 64, w.L[13].a + (size_t)b * N * 64, 64, w.T64 + (size_t)b * 4096, 64, w.local + (size_t)b * N * 64, 64, 0, c.st));
    BT_TRY(fwd_layer(c, 14, w.local, 64, 64, 0, nullptr, 0));
    BT_TRY(fwd_layer(c, 15, w.L[14].a, 64, 64, 0, nullptr, 0));
    BT_TRY(fwd_layer(c, 16, w.L[15].a, 128, 128, 0, nullptr, 0));
    hipLaunchKernelGGL(bt_rowmax_kernel, dim3(cdiv(d.G, 64), B), dim3(256), 0, c.st, w.L[16].a, N, d.G, w.pool_c, w.arg_c);
    return check_launch("baseline PointNet base forward");
}

// backward of the same given w.d_pool (gradient of the global feature) and, when local_accumulate, w.d_local (the segmentation head's
// gradient of the local features); d_feat_T [B, 64, 64] or nullptr joins at the feature transform
int base_bwd(const Ctx &c, const Dims &d, const float *x, int B, int N, const float *d_feat_T, int local_accumulate)
{
    Ws &w = *c.w;
    const int k = d.k;
    const size_t R = (size_t)B * N;
    // ---- conv_5 .. conv_3 ----
    hipLaunchKernelGGL(bt_fill_kernel, dim3(nb(R * d.G)), dim3(256), 0, c.st, w.dA, R * d.G, 0.f);
    hipLaunchKernelGGL(bt_pool_scatter_kernel, dim3(nb((size_t)B * d.G)), dim3(256), 0, c.st, w.d_pool, w.arg_c, B, N, d.G, w.dA);
    BT_TRY(bwd_layer(c, 16, w.dA, w.L[15].a, 128, 128, 0, w.dB, 128, 0));
    BT_TRY(bwd_layer(c, 15, w.dB, w.L[14].a, 64, 64, 0, w.dA, 64, 0));
    BT_TRY(bwd_layer(c, 14, w.dA, w.local, 64, 64, 0, w.d_local, 64, local_accumulate));                   // (+= the head's d_local)
    // ---- local = a_c2 x T64: dT64 = a_c2^T d_local (+ reg-loss gradient), d_a_c2 = d_local T64^T ----
    for (int b = 0; b < B; ++b) {
        BT_TRY(sgemm_small(1, 0, 64, 64, N, w.L[13].a + (size_t)b * N * 64, 64, w.d_local + (size_t)b * N * 64, 64, w.dT64 + (size_t)b * 4096, 64, 0, c.st));
        BT_TRY(sgemm_small(0, 1, N, 64, 64, w.d_local + (size_t)b * N * 64, 64, w.T64 + (size_t)b * 4096, 64, w.d_c2 + (size_t)b * N * 64, 64, 0, c.st));
    }
    if (d_feat_T) hipLaunchKernelGGL(bt_add_kernel, dim3(nb((size_t)B * 4096)), dim3(256), 0, c.st, w.dT64, d_feat_T, (size_t)B * 4096);
    BT_TRY(tnet_bwd(c, d, 6, w.dT64, w.L[13].a, 64, 64, B, N, w.pool_f, w.arg_f, w.d_c2, 64));             // += into d_a_c2
    // ---- conv_2, conv_1 ----
    BT_TRY(bwd_layer(c, 13, w.d_c2, w.L[12].a, 64, 64, 0, w.dA, 64, 0));
    BT_TRY(bwd_layer(c, 12, w.dA, w.xin, 9, 9, 0, w.d_xin, 9, 0));
    hipLaunchKernelGGL(bt_mix_bwd_kernel, dim3(B), dim3(256), 0, c.st, x, w.d_xin, N, k, w.dT3);
    // ---- input T-Net (its input is x: no gradient needed) ----
    return tnet_bwd(c, d, 0, w.dT3, x, 9, k, B, N, w.pool_i, w.arg_i, nullptr, 0);
}

}  // namespace
}  // namespace ampnet

using namespace ampnet;

extern "C" size_t ampnet_pointnet_seg_train_workspace_bytes(int variant, int B, int N, int n_classes)
{
    Dims d;
    if (!dims_for(variant, &d) || B < 1 || N < 1 || n_classes < 1) return 0;
    Ws w;
    carve(d, B, N, n_classes, nullptr, w);
    return w.bytes;
}

extern "C" int ampnet_pointnet_seg_train_fwd_f32(const float *const *layers_host, int variant, const float *x, int B, int N, int n_classes,
                                                 float *logits, float *feat_T, void *workspace, size_t workspace_bytes, void *stream)
{
    Dims d;
    AMPNET_REQUIRE(dims_for(variant, &d), "ampnet_pointnet_seg_train_fwd_f32: variant %d", variant);
    AMPNET_REQUIRE(layers_host && x && logits && feat_T && workspace, "ampnet_pointnet_seg_train_fwd_f32: null pointer");
    AMPNET_REQUIRE(B >= 2 && N >= 1 && n_classes >= 1 && n_classes <= 64 && (long long)B * N < (1LL << 24),
                   "ampnet_pointnet_seg_train_fwd_f32: bad shape B=%d (>= 2: batch statistics of the T-Net FC layers) N=%d classes=%d", B, N, n_classes);
    Ws w;
    carve(d, B, N, n_classes, workspace, w);
    AMPNET_REQUIRE(workspace_bytes >= w.bytes, "ampnet_pointnet_seg_train_fwd_f32: workspace %zu bytes, need %zu", workspace_bytes, w.bytes);
    TLayer L[AMPNET_POINTNET_LAYERS];
    BT_TRY(read_tables(layers_host, nullptr, L, nullptr));
    Ctx c{static_cast<hipStream_t>(stream), L, nullptr, &w};
    const int R = B * N;
    BT_TRY(base_fwd(c, d, x, B, N));
    const int ldw = d.G + 64;
    BT_TRY(sgemm_small(0, 1, B, d.H1, d.G, w.pool_c, d.G, L[17].W, ldw, w.gb, d.H1, 0, c.st));
    BT_TRY(fwd_layer(c, 17, w.local, 64, ldw, d.G, w.gb, N));
    BT_TRY(fwd_layer(c, 18, w.L[17].a, d.H1, d.H1, 0, nullptr, 0));
    BT_TRY(fwd_layer(c, 19, w.L[18].a, d.H2, d.H2, 0, nullptr, 0));
    BT_TRY(fwd_layer(c, 20, w.L[19].a, d.H3, d.H3, 0, nullptr, 0));
    const size_t n = (size_t)R * n_classes;
    hipLaunchKernelGGL(bt_logits_kernel, dim3(nb(n)), dim3(256), 0, c.st, w.L[20].z, B, N, n_classes, logits, 0);
    if (hipMemcpyAsync(feat_T, w.T64, (size_t)B * 4096 * sizeof(float), hipMemcpyDeviceToDevice, c.st) != hipSuccess)
        return fail(AMPNET_E_LAUNCH, "ampnet_pointnet_seg_train_fwd_f32: copy of the feature transform failed");
    return check_launch("ampnet_pointnet_seg_train_fwd_f32");
}

extern "C" int ampnet_pointnet_seg_bwd_f32(const float *const *layers_host, float *const *grads_host, int variant, const float *x, int B, int N,
                                           int n_classes, const float *dlogits, const float *d_feat_T, void *workspace, size_t workspace_bytes,
                                           void *stream)
{
    Dims d;
    AMPNET_REQUIRE(dims_for(variant, &d), "ampnet_pointnet_seg_bwd_f32: variant %d", variant);
    AMPNET_REQUIRE(layers_host && grads_host && x && dlogits && workspace, "ampnet_pointnet_seg_bwd_f32: null pointer");
    Ws w;
    carve(d, B, N, n_classes, workspace, w);
    AMPNET_REQUIRE(workspace_bytes >= w.bytes, "ampnet_pointnet_seg_bwd_f32: workspace %zu bytes, need %zu", workspace_bytes, w.bytes);
    TLayer L[AMPNET_POINTNET_LAYERS];
    TGrad G[AMPNET_POINTNET_LAYERS];
    BT_TRY(read_tables(layers_host, grads_host, L, G));
    Ctx c{static_cast<hipStream_t>(stream), L, G, &w};
    const int ldw = d.G + 64;
    const size_t R = (size_t)B * N;
    // ---- segmentation head ----
    hipLaunchKernelGGL(bt_logits_kernel, dim3(nb(R * n_classes)), dim3(256), 0, c.st, w.dA, B, N, n_classes, const_cast<float *>(dlogits), 1);
    BT_TRY(bwd_layer(c, 20, w.dA, w.L[19].a, d.H3, d.H3, 0, w.dB, d.H3, 0));
    BT_TRY(bwd_layer(c, 19, w.dB, w.L[18].a, d.H2, d.H2, 0, w.dA, d.H2, 0));
    BT_TRY(bwd_layer(c, 18, w.dA, w.L[17].a, d.H1, d.H1, 0, w.dB, d.H1, 0));
    // conv_1 on cat([global, local]): local half through the generic layer, global half = one row per window
    BT_TRY(bwd_layer(c, 17, w.dB, w.local, 64, ldw, d.G, w.d_local, 64, 0));                              // dW[:, G:], db, BatchNorm; d_local
    hipLaunchKernelGGL(bt_segsum_kernel, dim3(cdiv(d.H1, 64), B), dim3(256), 0, c.st, w.dB, N, d.H1, w.d_gb);
    BT_TRY(sgemm_small(1, 0, d.H1, d.G, B, w.d_gb, d.H1, w.pool_c, d.G, G[17].dW, ldw, 0, c.st));          // dW[:, :G]
    BT_TRY(sgemm_small(0, 0, B, d.G, d.H1, w.d_gb, d.H1, L[17].W, ldw, w.d_pool, d.G, 0, c.st));           // d global feature
    BT_TRY(base_bwd(c, d, x, B, N, d_feat_T, 1));
    return check_launch("ampnet_pointnet_seg_bwd_f32");
}

// ---- f4: ClassificationPointNet (pointnet.py:100-125, light_pointnet_256.py:100-125) on the same tape ---------------------------------
extern "C" size_t ampnet_pointnet_cls_workspace_bytes(int variant, int B, int N, int n_classes)
{
    Dims d;
    if (!dims_for(variant, &d) || B < 1 || N < 1 || n_classes < 1) return 0;
    Ws w;
    carve(d, B, N, n_classes, nullptr, w, true);
    return w.bytes;
}

extern "C" int ampnet_pointnet_cls_fwd_f32(const float *const *layers_host, int variant, const float *x, int B, int N, int n_classes, int train,
                                           float drop_p, uint32_t seed, float *log_probs, float *feat_T, void *workspace, size_t workspace_bytes,
                                           void *stream)
{
    Dims d;
    AMPNET_REQUIRE(dims_for(variant, &d), "ampnet_pointnet_cls_fwd_f32: variant %d", variant);
    AMPNET_REQUIRE(layers_host && x && log_probs && feat_T && workspace, "ampnet_pointnet_cls_fwd_f32: null pointer");
    AMPNET_REQUIRE(B >= (train ? 2 : 1) && N >= 1 && n_classes >= 1 && n_classes <= 64 && (long long)B * N < (1LL << 24),
                   "ampnet_pointnet_cls_fwd_f32: bad shape B=%d (train mode needs >= 2: batch statistics of the FC layers) N=%d classes=%d", B, N, n_classes);
    AMPNET_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "ampnet_pointnet_cls_fwd_f32: dropout p=%f", drop_p);
    Ws w;
    carve(d, B, N, n_classes, workspace, w, true);
    AMPNET_REQUIRE(workspace_bytes >= w.bytes, "ampnet_pointnet_cls_fwd_f32: workspace %zu bytes, need %zu", workspace_bytes, w.bytes);
    TLayer L[AMPNET_POINTNET_LAYERS];
    BT_TRY(read_tables(layers_host, nullptr, L, nullptr, AMPNET_POINTNET_CLS_LAYERS));
    Ctx c{static_cast<hipStream_t>(stream), L, nullptr, &w};
    c.running = train == 0;
    BT_TRY(base_fwd(c, d, x, B, N));
    BT_TRY(fwd_layer(c, 17, w.pool_c, d.G, d.G, 0, nullptr, 0));
    BT_TRY(fwd_layer(c, 18, w.L[17].a, d.C1, d.C1, 0, nullptr, 0));
    const float *a2 = w.L[18].a;
    if (train && drop_p > 0.f) {
        const size_t n = (size_t)B * d.C2;
        hipLaunchKernelGGL(bt_dropout_kernel, dim3(nb(n)), dim3(256), 0, c.st, a2, n, drop_base(seed, 0), drop_threshold(drop_p), 1.0f / (1.0f - drop_p), w.a_drop);
        a2 = w.a_drop;
    }
    BT_TRY(fwd_layer(c, 19, a2, d.C2, d.C2, 0, nullptr, 0));
    hipLaunchKernelGGL(bt_log_softmax_kernel, dim3(cdiv(B, 64)), dim3(64), 0, c.st, w.L[19].z, B, n_classes, log_probs);
    if (hipMemcpyAsync(feat_T, w.T64, (size_t)B * 4096 * sizeof(float), hipMemcpyDeviceToDevice, c.st) != hipSuccess)
        return fail(AMPNET_E_LAUNCH, "ampnet_pointnet_cls_fwd_f32: copy of the feature transform failed");
    return check_launch("ampnet_pointnet_cls_fwd_f32");
}

extern "C" int ampnet_pointnet_cls_bwd_f32(const float *const *layers_host, float *const *grads_host, int variant, const float *x, int B, int N,
                                           int n_classes, float drop_p, uint32_t seed, const float *d_log_probs, const float *d_feat_T,
                                           void *workspace, size_t workspace_bytes, void *stream)
{
    Dims d;
    AMPNET_REQUIRE(dims_for(variant, &d), "ampnet_pointnet_cls_bwd_f32: variant %d", variant);
    AMPNET_REQUIRE(layers_host && grads_host && x && d_log_probs && workspace, "ampnet_pointnet_cls_bwd_f32: null pointer");
    AMPNET_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "ampnet_pointnet_cls_bwd_f32: dropout p=%f", drop_p);
    Ws w;
    carve(d, B, N, n_classes, workspace, w, true);
    AMPNET_REQUIRE(workspace_bytes >= w.bytes, "ampnet_pointnet_cls_bwd_f32: workspace %zu bytes, need %zu", workspace_bytes, w.bytes);
    TLayer L[AMPNET_POINTNET_LAYERS];
    TGrad G[AMPNET_POINTNET_LAYERS];
    BT_TRY(read_tables(layers_host, grads_host, L, G, AMPNET_POINTNET_CLS_LAYERS));
    Ctx c{static_cast<hipStream_t>(stream), L, G, &w};
    // log_softmax, fc_3 (its input is the dropped activation), dropout, fc_2, fc_1
    hipLaunchKernelGGL(bt_log_softmax_bwd_kernel, dim3(cdiv(B, 64)), dim3(64), 0, c.st, w.L[19].z, d_log_probs, B, n_classes, w.dA);
    const bool drop = drop_p > 0.f;
    BT_TRY(bwd_layer(c, 19, w.dA, drop ? w.a_drop : w.L[18].a, d.C2, d.C2, 0, w.dB, d.C2, 0));
    if (drop) {
        const size_t n = (size_t)B * d.C2;
        hipLaunchKernelGGL(bt_dropout_kernel, dim3(nb(n)), dim3(256), 0, c.st, w.dB, n, drop_base(seed, 0), drop_threshold(drop_p), 1.0f / (1.0f - drop_p), w.dB);
    }
    BT_TRY(bwd_layer(c, 18, w.dB, w.L[17].a, d.C1, d.C1, 0, w.dA, d.C1, 0));
    BT_TRY(bwd_layer(c, 17, w.dA, w.pool_c, d.G, d.G, 0, w.d_pool, d.G, 0));
    BT_TRY(base_bwd(c, d, x, B, N, d_feat_T, 0));
    return check_launch("ampnet_pointnet_cls_bwd_f32");
}
