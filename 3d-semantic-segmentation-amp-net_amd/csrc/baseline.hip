// baseline.hip -- eval forward of the baseline single-window PointNet segmentation model:
// pointNet/model/pointnet.py (TransformationNet :6-44, BasePointNet :47-97, SegmentationPointNet :128-154; 1024-d,
// convolutions WITH bias, T-Net on x[:, :, :3]) and pointNet/model/light_pointnet_256.py (:128-153; 256-d, no
// conv/fc bias, T-Net on x[:, :, :2]).  Config 1 of BASELINE.json is the reference's CPU plumbing case ([4, 512, 9]):
// this path is built for parity, not speed -- every layer is sgemm_small + one affine pass.  BatchNorm uses the
// running statistics (module.eval()); training this model is not part of the HIP path.
#include "kernels.h"

namespace ampnet {
namespace {

struct Layer {                 // six device pointers per layer, any of b / BatchNorm may be null
    const float *W, *b, *g, *be, *rm, *rv;
};

struct Dims {
    int k, G, F1, F2, H1, H2, H3;
};

bool dims_for(int variant, Dims *d)
{
    if (variant == 0) *d = Dims{3, 1024, 512, 256, 512, 256, 128};
    else if (variant == 1) *d = Dims{2, 256, 256, 128, 256, 128, 64};
    else return false;
    return true;
}

// scale / shift of "conv bias, then eval BatchNorm": y = (z + b - rm) * g / sqrt(rv + eps) + be
__global__ void bl_fold_kernel(Layer l, int C, float eps, float *__restrict__ scale, float *__restrict__ shift)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float b = l.b ? l.b[c] : 0.f;
    if (l.g) {
        const float s = l.g[c] / sqrtf(l.rv[c] + eps);
        scale[c] = s;
        shift[c] = fmaf(b - l.rm[c], s, l.be[c]);
    } else {
        scale[c] = 1.0f;
        shift[c] = b;
    }
}

// in place: z[row][c] = act((z + add[row / per][c]) * scale[c] + shift[c])
__global__ void bl_affine_kernel(float *__restrict__ z, size_t n, int C, const float *__restrict__ scale,
                                 const float *__restrict__ shift, const float *__restrict__ add, int per, int relu)
{
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C);
    float v = z[i];
    if (add) v += add[(i / C / per) * C + c];
    v = fmaf(v, scale[c], shift[c]);
    z[i] = relu ? fmaxf(v, 0.f) : v;
}

// [B, N, C] -> [B, C] maximum over the N rows of a window; block = (64 channels, window), 4 row groups
__global__ __launch_bounds__(256) void bl_rowmax_kernel(const float *__restrict__ a, int N, int C, float *__restrict__ out)
{
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6, b = blockIdx.y;
    float m = -__builtin_inff();
    if (c < C)
        for (int r = g; r < N; r += 4) m = fmaxf(m, a[((size_t)b * N + r) * C + c]);
    red[g][threadIdx.x & 63] = m;
    __syncthreads();
    if (g == 0 && c < C) out[(size_t)b * C + c] = fmaxf(fmaxf(red[0][threadIdx.x], red[1][threadIdx.x]), fmaxf(red[2][threadIdx.x], red[3][threadIdx.x]));
}

// cat([x[:, :, :k] @ T, x[:, :, k:]], 2): pointnet.py:71-74 (k = 3), light_pointnet_256.py:71-74 (k = 2)
__global__ void bl_mix_kernel(const float *__restrict__ x, const float *__restrict__ T, int R, int N, int k, float *__restrict__ out)
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

// z [B * N, C] -> logits [B, C, N]
__global__ void bl_logits_kernel(const float *__restrict__ z, int B, int N, int C, float *__restrict__ logits)
{
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= (size_t)B * N * C) return;
    const int p = (int)(i % N), c = (int)((i / N) % C), b = (int)(i / N / C);
    logits[i] = z[((size_t)b * N + p) * C + c];
}

struct Ctx {
    hipStream_t st;
    float *scale, *shift;
    float eps;
};

#define BL_TRY(expr)            \
    do {                        \
        int rc_ = (expr);       \
        if (rc_ != AMPNET_OK) return rc_; \
    } while (0)

// out [M, cout] = act(bn(A [M, K] (row stride lda) @ W[:, w0 : w0 + K]^T + b)) (+ per-window addend before the affine)
int layer(const Ctx &c, const float *A, int lda, int M, int K, const Layer &l, int ldw, int w0, int cout, float *out,
          const float *add, int per, int relu)
{
    AMPNET_REQUIRE(l.W, "baseline PointNet: a layer has no weight pointer");
    BL_TRY(sgemm_small(0, 1, M, cout, K, A, lda, l.W + w0, ldw, out, cout, 0, c.st));
    hipLaunchKernelGGL(bl_fold_kernel, dim3(cdiv(cout, 256)), dim3(256), 0, c.st, l, cout, c.eps, c.scale, c.shift);
    const size_t n = (size_t)M * cout;
    hipLaunchKernelGGL(bl_affine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.st, out, n, cout, c.scale, c.shift, add,
                       per > 0 ? per : 1, relu);
    return check_launch("baseline layer");
}

struct Ws {
    float *xin, *a64a, *a64b, *a128, *big, *pool, *f1, *f2, *T3, *T64, *local, *h2, *h3, *gb, *zc, *scale, *shift;
    size_t bytes;
};

void carve(const Dims &d, int B, int N, int C, void *base, Ws &w)
{
    size_t off = 0;
    const size_t R = (size_t)B * N;
    auto take = [&](size_t n_floats) {
        float *p = base ? reinterpret_cast<float *>(static_cast<char *>(base) + off) : nullptr;
        off += align_up(n_floats * sizeof(float), 256);
        return p;
    };
    const int big = d.G > d.H1 ? d.G : d.H1;
    w.xin = take(R * 9);
    w.a64a = take(R * 64);
    w.a64b = take(R * 64);
    w.a128 = take(R * 128);
    w.big = take(R * big);                 // conv 128 -> G output, later the segmentation head's first layer
    w.pool = take((size_t)B * d.G);
    w.f1 = take((size_t)B * d.F1);
    w.f2 = take((size_t)B * d.F2);
    w.T3 = take((size_t)B * 9);
    w.T64 = take((size_t)B * 4096);
    w.local = take(R * 64);
    w.h2 = take(R * d.H2);
    w.h3 = take(R * d.H3);
    w.gb = take((size_t)B * d.H1);
    w.zc = take(R * C);
    w.scale = take(4096);
    w.shift = take(4096);
    w.bytes = off;
}

// T-Net (pointnet.py:26-44): A [R, k] (row stride lda) -> T [B, k, k]
int tnet(const Ctx &c, const Dims &d, const Layer *L, const float *A, int lda, int k, int B, int N, Ws &w, float *T)
{
    const int R = B * N;
    BL_TRY(layer(c, A, lda, R, k, L[0], k, 0, 64, w.a64a, nullptr, 0, 1));
    BL_TRY(layer(c, w.a64a, 64, R, 64, L[1], 64, 0, 128, w.a128, nullptr, 0, 1));
    BL_TRY(layer(c, w.a128, 128, R, 128, L[2], 128, 0, d.G, w.big, nullptr, 0, 1));
    hipLaunchKernelGGL(bl_rowmax_kernel, dim3(cdiv(d.G, 64), B), dim3(256), 0, c.st, w.big, N, d.G, w.pool);
    BL_TRY(layer(c, w.pool, d.G, B, d.G, L[3], d.G, 0, d.F1, w.f1, nullptr, 0, 1));
    BL_TRY(layer(c, w.f1, d.F1, B, d.F1, L[4], d.F1, 0, d.F2, w.f2, nullptr, 0, 1));
    BL_TRY(layer(c, w.f2, d.F2, B, d.F2, L[5], d.F2, 0, k * k, T, nullptr, 0, 0));
    return add_identity(T, B, k, c.st);
}

}  // namespace
}  // namespace ampnet

using namespace ampnet;

extern "C" size_t ampnet_pointnet_seg_workspace_bytes(int variant, int B, int N, int n_classes)
{
    Dims d;
    if (!dims_for(variant, &d) || B < 1 || N < 1 || n_classes < 1) return 0;
    Ws w;
    carve(d, B, N, n_classes, nullptr, w);
    return w.bytes;
}

extern "C" int ampnet_pointnet_seg_fwd_f32(const float *const *layers_host, int variant, const float *x, int B, int N,
                                           int n_classes, float *logits, float *feat_T, void *workspace,
                                           size_t workspace_bytes, void *stream)
{
    Dims d;
    AMPNET_REQUIRE(dims_for(variant, &d), "ampnet_pointnet_seg_fwd_f32: variant %d (0 = pointnet.py, 1 = light_pointnet_256.py)", variant);
    AMPNET_REQUIRE(layers_host && x && logits && workspace, "ampnet_pointnet_seg_fwd_f32: null pointer");
    AMPNET_REQUIRE(B >= 1 && N >= 1 && n_classes >= 1 && n_classes <= 64 && (long long)B * N < (1LL << 30),
                   "ampnet_pointnet_seg_fwd_f32: bad shape B=%d N=%d classes=%d", B, N, n_classes);
    Ws w;
    carve(d, B, N, n_classes, workspace, w);
    AMPNET_REQUIRE(workspace_bytes >= w.bytes, "ampnet_pointnet_seg_fwd_f32: workspace %zu bytes, need %zu", workspace_bytes, w.bytes);
    Layer L[AMPNET_POINTNET_LAYERS];
    for (int i = 0; i < AMPNET_POINTNET_LAYERS; ++i) {
        const float *const *p = layers_host + 6 * i;
        L[i] = Layer{p[0], p[1], p[2], p[3], p[4], p[5]};
        AMPNET_REQUIRE(L[i].W, "ampnet_pointnet_seg_fwd_f32: layer %d has no weight", i);
        AMPNET_REQUIRE(!L[i].g || (L[i].be && L[i].rm && L[i].rv), "ampnet_pointnet_seg_fwd_f32: layer %d: incomplete BatchNorm", i);
    }
    Ctx c{static_cast<hipStream_t>(stream), w.scale, w.shift, 1e-5f};
    const int R = B * N, k = d.k;
    // input transform on x[:, :, :k], folded back into the 9 input channels
    BL_TRY(tnet(c, d, L + 0, x, 9, k, B, N, w, w.T3));
    hipLaunchKernelGGL(bl_mix_kernel, dim3(cdiv(R, 256)), dim3(256), 0, c.st, x, w.T3, R, N, k, w.xin);
    BL_TRY(layer(c, w.xin, 9, R, 9, L[12], 9, 0, 64, w.a64a, nullptr, 0, 1));
    BL_TRY(layer(c, w.a64a, 64, R, 64, L[13], 64, 0, 64, w.a64b, nullptr, 0, 1));
    // feature transform, local features = bmm(x, T64)
    BL_TRY(tnet(c, d, L + 6, w.a64b, 64, 64, B, N, w, w.T64));
    for (int b = 0; b < B; ++b)
        BL_TRY(sgemm_small(0, 0, N, 64, 64, w.a64b + (size_t)b * N * 64, 64, w.T64 + (size_t)b * 4096, 64, w.local + (size_t)b * N * 64, 64, 0, c.st));
    BL_TRY(layer(c, w.local, 64, R, 64, L[14], 64, 0, 64, w.a64a, nullptr, 0, 1));
    BL_TRY(layer(c, w.a64a, 64, R, 64, L[15], 64, 0, 128, w.a128, nullptr, 0, 1));
    BL_TRY(layer(c, w.a128, 128, R, 128, L[16], 128, 0, d.G, w.big, nullptr, 0, 1));
    hipLaunchKernelGGL(bl_rowmax_kernel, dim3(cdiv(d.G, 64), B), dim3(256), 0, c.st, w.big, N, d.G, w.pool);
    // segmentation head on cat([global, local]) (pointnet.py:95): the global part is one row per window
    const int ldw = d.G + 64;
    BL_TRY(sgemm_small(0, 1, B, d.H1, d.G, w.pool, d.G, L[17].W, ldw, w.gb, d.H1, 0, c.st));
    BL_TRY(layer(c, w.local, 64, R, 64, L[17], ldw, d.G, d.H1, w.big, w.gb, N, 1));
    BL_TRY(layer(c, w.big, d.H1, R, d.H1, L[18], d.H1, 0, d.H2, w.h2, nullptr, 0, 1));
    BL_TRY(layer(c, w.h2, d.H2, R, d.H2, L[19], d.H2, 0, d.H3, w.h3, nullptr, 0, 1));
    BL_TRY(layer(c, w.h3, d.H3, R, d.H3, L[20], d.H3, 0, n_classes, w.zc, nullptr, 0, 0));
    const size_t n = (size_t)R * n_classes;
    hipLaunchKernelGGL(bl_logits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.st, w.zc, B, N, n_classes, logits);
    if (feat_T && hipMemcpyAsync(feat_T, w.T64, (size_t)B * 4096 * sizeof(float), hipMemcpyDeviceToDevice, c.st) != hipSuccess)
        return fail(AMPNET_E_LAUNCH, "ampnet_pointnet_seg_fwd_f32: copy of the feature transform failed");
    return check_launch("ampnet_pointnet_seg_fwd_f32");
}
