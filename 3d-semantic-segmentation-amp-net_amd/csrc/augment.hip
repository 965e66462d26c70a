// augment.hip -- the per-step input pipeline of train_loop on the device: the reference
// augments on the host with numpy and copies nine windows up one by one (train_pointnet-attention.py:390-405,
// utils/utils.py:582-632).  Here the collated batch goes up once, as it left collate_seq_padd, and ONE kernel applies
//     the cluster permutation shared by the batch          (shuffle_clusters, utils.py:620-632)
//     the z-rotation of x, y, z with one angle per step     (rotate_point_cloud_z, utils.py:582-604: float64 product of the
//                                                            float32 coordinates with [[c, s, 0], [-s, c, 0], [0, 0, 1]],
//                                                            rounded to float32)
//     one point permutation per window shared by the batch  (shuffle_data, utils.py:607-617)
// and the [B, N, 9, W] -> [B, W, N, 9] re-layout the encoder wants.  The permutations and the angle are drawn on the host
// from numpy's global RNG in the reference's order (amp_step.augment_batch_device), so a seeded run sees the reference's batch.
#include "common.h"

#pragma clang fp contract(off)

namespace ampnet {

__global__ __launch_bounds__(256) void augment_kernel(const float *__restrict__ pc, const long long *__restrict__ tg,
                                                      const int *__restrict__ cluster_perm, const int *__restrict__ point_perm,
                                                      double c, double s, int rotate, int B, int N, int W, float *__restrict__ x_out,
                                                      long long *__restrict__ t_out)
{
    const long long i = blockIdx.x * 256LL + threadIdx.x;           // output row (b, w, n)
    if (i >= (long long)B * W * N) return;
    const int n = (int)(i % N), w = (int)((i / N) % W), b = (int)(i / N / W);
    const int sw = cluster_perm[w];
    const int sn = point_perm ? point_perm[(size_t)w * N + n] : n;
    const float *src = pc + (((size_t)b * N + sn) * 9) * W + sw;      // feature f at src[f * W]
    float v[9];
#pragma unroll
    for (int f = 0; f < 9; ++f) v[f] = src[(size_t)f * W];
    if (rotate) {
        const double x = (double)v[0], y = (double)v[1], z = (double)v[2];
        v[0] = (float)((x * c + y * (-s)) + z * 0.0);                 // np.dot(xyz, rot), column by column, in float64
        v[1] = (float)((x * s + y * c) + z * 0.0);
        v[2] = (float)((x * 0.0 + y * 0.0) + z * 1.0);
    }
    float *dst = x_out + (size_t)i * 9;
#pragma unroll
    for (int f = 0; f < 9; ++f) dst[f] = v[f];
    if (tg) t_out[i] = tg[((size_t)b * N + sn) * W + sw];
}

// The same kernel fed by the RAGGED batch of collate_seq_ragged (package pointNet/collate_fns.py): the resampling to N points and the
// padding to W clusters that collate_seq_padd does on the host (pointNet/collate_fns.py:33-45 -- 42 MB of gathers and copies per batch of
// 64 in a DataLoader worker) happen HERE, while the row is gathered anyway:
//     padded[b][p][f][c] = pts_b[idx[b][p]][f][min(c, w_b - 1)]          ("replicate" padding of the cluster axis)
//     targets[b][p][c]   = c < w_b ? lab_b[idx[b][p]][c] : -1            (constant -1 padding)
// with pts_b [n_b, 9, w_b] and lab_b [n_b, w_b] (int8) of sample b at pts + meta[b].pts_off / lab + meta[b].lab_off.
struct RaggedMeta {
    int n, w, pts_off, lab_off;       // points and clusters of the sample, element offsets of its two arrays
};
__global__ __launch_bounds__(256) void collate_augment_kernel(const float *__restrict__ pts, const signed char *__restrict__ lab,
                                                              const int *__restrict__ idx, const RaggedMeta *__restrict__ meta,
                                                              const int *__restrict__ cluster_perm, const int *__restrict__ point_perm,
                                                              double c, double s, int rotate, int B, int N, int W, float *__restrict__ x_out,
                                                              long long *__restrict__ t_out)
{
    const long long i = blockIdx.x * 256LL + threadIdx.x;           // output row (b, w, n)
    if (i >= (long long)B * W * N) return;
    const int n = (int)(i % N), w = (int)((i / N) % W), b = (int)(i / N / W);
    const RaggedMeta m = meta[b];
    const int sw = cluster_perm[w];
    const int sn = point_perm ? point_perm[(size_t)w * N + n] : n;
    const int row = idx[(size_t)b * N + sn];                        // the sample's own row behind padded row sn
    const int sc = sw < m.w ? sw : m.w - 1;                         // padded clusters replicate the last real one
    const float *src = pts + m.pts_off + ((size_t)row * 9) * m.w + sc;      // feature f at src[f * w_b]
    float v[9];
#pragma unroll
    for (int f = 0; f < 9; ++f) v[f] = src[(size_t)f * m.w];
    if (rotate) {
        const double x = (double)v[0], y = (double)v[1], z = (double)v[2];
        v[0] = (float)((x * c + y * (-s)) + z * 0.0);
        v[1] = (float)((x * s + y * c) + z * 0.0);
        v[2] = (float)((x * 0.0 + y * 0.0) + z * 1.0);
    }
    float *dst = x_out + (size_t)i * 9;
#pragma unroll
    for (int f = 0; f < 9; ++f) dst[f] = v[f];
    t_out[i] = sw < m.w ? (long long)lab[m.lab_off + (size_t)row * m.w + sw] : -1LL;
}

}  // namespace ampnet

extern "C" int ampnet_collate_augment_f32(const float *pts, const signed char *labels, const int32_t *idx, const int32_t *meta,
                                          const int32_t *cluster_perm, const int32_t *point_perm, double cos_a, double sin_a, int rotate,
                                          int B, int N, int W, float *x_out, long long *t_out, void *stream)
{
    using namespace ampnet;
    AMPNET_REQUIRE(pts && labels && idx && meta && cluster_perm && x_out && t_out, "ampnet_collate_augment_f32: null pointer");
    AMPNET_REQUIRE(B >= 1 && N >= 1 && W >= 1 && (long long)B * N * W < (1LL << 31), "ampnet_collate_augment_f32: bad shape B=%d N=%d W=%d", B, N, W);
    static_assert(sizeof(RaggedMeta) == 4 * sizeof(int32_t), "meta is [B, 4] int32");
    const long long rows = (long long)B * N * W;
    hipLaunchKernelGGL(collate_augment_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pts, labels, idx,
                       reinterpret_cast<const RaggedMeta *>(meta), cluster_perm, point_perm, cos_a, sin_a, rotate, B, N, W, x_out, t_out);
    return check_launch("collate_augment_kernel");
}

extern "C" int ampnet_augment_f32(const float *pc, const long long *targets, const int32_t *cluster_perm, const int32_t *point_perm,
                                  double cos_a, double sin_a, int rotate, int B, int N, int W, float *x_out, long long *t_out, void *stream)
{
    using namespace ampnet;
    AMPNET_REQUIRE(pc && cluster_perm && x_out && (targets == nullptr) == (t_out == nullptr), "ampnet_augment_f32: null pointer");
    AMPNET_REQUIRE(B >= 1 && N >= 1 && W >= 1 && (long long)B * N * W < (1LL << 31), "ampnet_augment_f32: bad shape B=%d N=%d W=%d", B, N, W);
    const long long rows = (long long)B * N * W;
    hipLaunchKernelGGL(augment_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pc, targets, cluster_perm,
                       point_perm, cos_a, sin_a, rotate, B, N, W, x_out, t_out);
    return check_launch("augment_kernel");
}
