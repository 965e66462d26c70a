// fps.hip -- greedy farthest-point sampling, one workgroup per cloud (C ABI: ampnet_fps_f32).
//
// Replaces utils/utils.py:889-933 of the reference.  The algorithm is S-1 dependent rounds; a round is
//   d[j] = min(d[j], |p_last - p_j|^2)  for every point, then argmax_j d[j] (first index on ties).
// Mapping to CDNA4: the cloud lives in REGISTERS for the whole kernel (thread t owns points t, t+T, ...:
// coalesced 4-byte loads once, 12 B xyz + 4 B running minimum per point = the 16 B per (candidate, round)
// the roofline counts, served from the register file instead of HBM), a round costs
//   VALU update  ->  64-lane argmax by DPP/shuffles  ->  one LDS slot per wave  ->  ONE barrier
// and the winner's coordinates travel with its slot, so no thread ever indexes its register array at run
// time and nothing is re-read from global memory.  Slots are double-buffered so a round needs one barrier.
//
// Bit parity: distances are float32 ((dx*dx + dy*dy) + dz*dz) with one rounding per operation -- this file
// is compiled with -ffp-contract=off and the pragma below repeats it; picked points carry -1 so that
// min(d, -1) keeps them out for good (every true distance is >= 0); ties go to the lowest index.
#include "common.h"

#pragma clang fp contract(off)

namespace ampnet {

struct Cand {
    float d;
    int i;
};

__device__ __forceinline__ bool better(float da, int ia, float db, int ib)
{
    return (da > db) || (da == db && ia < ib);
}

template <int T, int P>
__global__ __launch_bounds__(T) void fps_kernel(const float *__restrict__ xyz, int n, int ld, int s,
                                                int32_t *__restrict__ idx)
{
    constexpr int NW = T / WAVE;
    // slot = {d, i, x, y, z} per wave, two generations
    __shared__ float s_d[2][NW];
    __shared__ int s_i[2][NW];
    __shared__ float s_x[2][NW], s_y[2][NW], s_z[2][NW];

    const int tid = threadIdx.x;
    const int wave = tid / WAVE;
    const float *cloud = xyz + (size_t)blockIdx.x * n * ld;
    int32_t *out = idx + (size_t)blockIdx.x * s;

    float px[P], py[P], pz[P], dist[P];
#pragma unroll
    for (int k = 0; k < P; ++k) {
        const int j = tid + k * T;
        const bool ok = j < n;
        px[k] = ok ? cloud[(size_t)j * ld + 0] : 0.f;
        py[k] = ok ? cloud[(size_t)j * ld + 1] : 0.f;
        pz[k] = ok ? cloud[(size_t)j * ld + 2] : 0.f;
        dist[k] = ok ? __builtin_inff() : -2.0f;   // -2: padding lanes never win, never change
    }
    // seed: point 0 (utils.py:907-908)
    float lx = cloud[0], ly = cloud[1], lz = cloud[2];
    if (tid == 0) {
        out[0] = 0;
        dist[0] = -1.0f;
    }

    for (int r = 1; r < s; ++r) {
        float bd = -3.0f, bx = 0.f, by = 0.f, bz = 0.f;
        int bi = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const float dx = lx - px[k];
            const float dy = ly - py[k];
            const float dz = lz - pz[k];
            const float d = (dx * dx + dy * dy) + dz * dz;
            const float m = fminf(d, dist[k]);          // picked (-1) and padding (-2) stay as they are
            dist[k] = m;
            if (m > bd) {                               // strict: first (lowest) index wins inside a thread
                bd = m;
                bi = tid + k * T;
                bx = px[k];
                by = py[k];
                bz = pz[k];
            }
        }
        // 64-lane argmax on (d, i)
        float wd = bd;
        int wi = bi;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float od = __shfl_xor(wd, off);
            const int oi = __shfl_xor(wi, off);
            if (better(od, oi, wd, wi)) {
                wd = od;
                wi = oi;
            }
        }
        const int g = r & 1;
        if (wi == bi) {          // exactly one lane of the wave owns the wave's winner (indices are unique)
            s_d[g][wave] = bd;
            s_i[g][wave] = bi;
            s_x[g][wave] = bx;
            s_y[g][wave] = by;
            s_z[g][wave] = bz;
        }
        __syncthreads();
        // every thread folds the NW slots (LDS broadcast reads)
        float gd = s_d[g][0];
        int gi = s_i[g][0];
        int gw = 0;
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            const float od = s_d[g][w];
            const int oi = s_i[g][w];
            if (better(od, oi, gd, gi)) {
                gd = od;
                gi = oi;
                gw = w;
            }
        }
        lx = s_x[g][gw];
        ly = s_y[g][gw];
        lz = s_z[g][gw];
        if (tid == 0) out[r] = gi;
        // retire the winner in its owner's registers: compile-time k, run-time predicate
        const int own_t = gi % T;
        if (tid == own_t) {
            const int own_k = gi / T;
#pragma unroll
            for (int k = 0; k < P; ++k)
                if (k == own_k) dist[k] = -1.0f;
        }
    }
}

template <int T, int P>
static int launch(const float *xyz, int n_clouds, int n, int ld, int s, int32_t *idx, hipStream_t st)
{
    hipLaunchKernelGGL((fps_kernel<T, P>), dim3(n_clouds), dim3(T), 0, st, xyz, n, ld, s, idx);
    return check_launch("fps_kernel");
}

__global__ void gather_rows_kernel(const float *__restrict__ src, const int32_t *__restrict__ idx, int n,
                                   int ld, int s, float *__restrict__ out)
{
    const int c = blockIdx.y;
    const size_t total = (size_t)s * ld;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total;
         e += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / ld), f = (int)(e % ld);
        const int j = idx[(size_t)c * s + i];
        out[(size_t)c * total + e] = src[((size_t)c * n + j) * ld + f];
    }
}

}  // namespace ampnet

extern "C" int ampnet_fps_f32(const float *xyz, int n_clouds, int n, int ld, int s, int32_t *idx, void *stream)
{
    using namespace ampnet;
    AMPNET_REQUIRE(xyz && idx, "ampnet_fps_f32: null pointer");
    AMPNET_REQUIRE(n_clouds >= 1 && n >= 1 && ld >= 3, "ampnet_fps_f32: bad shape n_clouds=%d n=%d ld=%d", n_clouds, n, ld);
    AMPNET_REQUIRE(s >= 1 && s <= n, "ampnet_fps_f32: n_samples=%d must be in [1, n=%d]", s, n);
    AMPNET_REQUIRE(n <= 16384, "ampnet_fps_f32: n=%d exceeds 16384 points per cloud", n);
    hipStream_t st = (hipStream_t)stream;
    if (n <= 256) return launch<256, 1>(xyz, n_clouds, n, ld, s, idx, st);
    if (n <= 1024) return launch<256, 4>(xyz, n_clouds, n, ld, s, idx, st);
    if (n <= 2048) return launch<512, 4>(xyz, n_clouds, n, ld, s, idx, st);
    if (n <= 4096) return launch<1024, 4>(xyz, n_clouds, n, ld, s, idx, st);
    if (n <= 8192) return launch<1024, 8>(xyz, n_clouds, n, ld, s, idx, st);
    return launch<1024, 16>(xyz, n_clouds, n, ld, s, idx, st);
}

extern "C" int ampnet_gather_rows_f32(const float *src, const int32_t *idx, int n_clouds, int n, int ld, int s,
                                      float *out, void *stream)
{
    using namespace ampnet;
    AMPNET_REQUIRE(src && idx && out, "ampnet_gather_rows_f32: null pointer");
    AMPNET_REQUIRE(n_clouds >= 1 && n >= 1 && ld >= 1 && s >= 1, "ampnet_gather_rows_f32: bad shape");
    const int blocks = cdiv(s * ld, 256) < 1024 ? cdiv(s * ld, 256) : 1024;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks, n_clouds), dim3(256), 0, (hipStream_t)stream, src, idx, n, ld, s, out);
    return check_launch("gather_rows_kernel");
}
