// kmeans.hip -- size-constrained k-means on the GPU (C ABI: ampnet_kmeans_balanced_f32), SURVEY.md section 8(f) rank 2.
//
// Replaces the window grouping of data_proc/3_kmeans.py:54-82 and utils/utils.py:473-535, which call the third-party
// k_means_constrained.KMeansConstrained (min-cost-flow assignment; not part of the reference repository, version unpinned):
// parity with it is UNPINNED by construction.  The spec built here is deterministic and restated in oracle/kmeans_oracle.py:
//   seeding    farthest-point seeding in feature space from a start point (init 0: point 0; init t: hash(seed, t) % n)
//   assignment greedy by ascending (distance, point, cluster): sweep 1 gives every cluster its first size_min points, sweep 2 places
//              the rest with capacity size_max -- clusters end with size_min <= size <= size_max (exactly n / k when both are n / k)
//   update     cluster means (double accumulation, fixed order); stop when the summed squared centre shift <= tol * mean feature variance
//   n_init     restarts, the lowest inertia wins (ties: the earlier init)
// CDNA4 mapping: distances are float32 ((d0*d0 + d1*d1) + d2*d2) (no FMA, this file is compiled with -ffp-contract=off) packed with
// the (point, cluster) id into 64-bit keys; the keys are sorted by a bitonic network whose strides below 2048 run inside LDS
// (one launch per merge stage instead of eleven); the capacity sweep is inherently ordered and runs on one wave with the pick
// bitmap in LDS; iterations run without host synchronisation (a device flag turns the remaining launches into no-ops).
#include "common.h"

#pragma clang fp contract(off)

namespace ampnet {
namespace {

constexpr int KM_MAXK = 32;             // clusters (the reference uses <= 18)
constexpr int KM_LOCAL = 4096;          // keys sorted inside one workgroup's LDS
constexpr int KM_T = 1024;

struct KmState {                        // device-side control block
    int done;                           // Lloyd loop converged: the remaining launches of this init do nothing
    int iters;
    float shift;
    float tol_abs;
    double inertia;
    double best_inertia;
    int best_init;
};

__device__ __forceinline__ uint32_t km_hash(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

// tol_abs = tol * mean over the features of the (population) variance; one workgroup
__global__ __launch_bounds__(KM_T) void km_prepare_kernel(const float *__restrict__ F, int n, float tol, KmState *st)
{
    __shared__ double red[KM_T];
    double var_sum = 0.0;
    for (int f = 0; f < 3; ++f) {
        double s = 0.0;
        for (int i = threadIdx.x; i < n; i += KM_T) s += (double)F[(size_t)i * 3 + f];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int w = KM_T / 2; w > 0; w >>= 1) {
            if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
            __syncthreads();
        }
        const double mean = red[0] / n;
        __syncthreads();
        double q = 0.0;
        for (int i = threadIdx.x; i < n; i += KM_T) {
            const double d = (double)F[(size_t)i * 3 + f] - mean;
            q += d * d;
        }
        red[threadIdx.x] = q;
        __syncthreads();
        for (int w = KM_T / 2; w > 0; w >>= 1) {
            if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
            __syncthreads();
        }
        var_sum += red[0] / n;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        st->tol_abs = (float)((double)tol * var_sum / 3.0);
        st->best_inertia = 1e300;
        st->best_init = -1;
    }
}

// farthest-point seeding of k centres from start point s (one workgroup; distances float32, first maximum on ties)
__global__ __launch_bounds__(KM_T) void km_seed_kernel(const float *__restrict__ F, int n, int k, uint32_t seed, int init, float *__restrict__ C,
                                                       float *__restrict__ dmin, KmState *st)
{
    __shared__ float s_val[KM_T];
    __shared__ int s_idx[KM_T];
    __shared__ int s_last;
    const int tid = threadIdx.x;
    if (tid == 0) {
        s_last = init == 0 ? 0 : (int)(km_hash(seed + 0x9E3779B9U * (uint32_t)init) % (uint32_t)n);
        st->done = 0;
        st->iters = 0;
    }
    for (int i = tid; i < n; i += KM_T) dmin[i] = __builtin_inff();
    __syncthreads();
    for (int c = 0; c < k; ++c) {
        const int last = s_last;
        const float lx = F[(size_t)last * 3], ly = F[(size_t)last * 3 + 1], lz = F[(size_t)last * 3 + 2];
        if (tid < 3) C[c * 3 + tid] = F[(size_t)last * 3 + tid];
        float bv = -1.0f;
        int bi = 0;
        for (int i = tid; i < n; i += KM_T) {
            const float dx = lx - F[(size_t)i * 3], dy = ly - F[(size_t)i * 3 + 1], dz = lz - F[(size_t)i * 3 + 2];
            const float d = (dx * dx + dy * dy) + dz * dz;
            const float m = fminf(d, dmin[i]);
            dmin[i] = m;
            if (m > bv) {            // ascending i inside the thread: first maximum
                bv = m;
                bi = i;
            }
        }
        s_val[tid] = bv;
        s_idx[tid] = bi;
        __syncthreads();
        for (int w = KM_T / 2; w > 0; w >>= 1) {
            if (tid < w) {
                const float ov = s_val[tid + w];
                const int oi = s_idx[tid + w];
                if (ov > s_val[tid] || (ov == s_val[tid] && oi < s_idx[tid])) {
                    s_val[tid] = ov;
                    s_idx[tid] = oi;
                }
            }
            __syncthreads();
        }
        if (tid == 0) s_last = s_idx[0];
        __syncthreads();
    }
}

// keys[p] = (float bits of the squared distance) << 32 | (point * 32 + cluster); padding keys = all ones
__global__ void km_keys_kernel(const float *__restrict__ F, const float *__restrict__ C, int n, int k, size_t n_keys, unsigned long long *__restrict__ keys,
                               const KmState *st)
{
    if (st->done) return;
    const size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (p >= n_keys) return;
    const int i = (int)(p / k), c = (int)(p % k);
    if (i >= n) {
        keys[p] = ~0ull;
        return;
    }
    const float dx = F[(size_t)i * 3] - C[c * 3], dy = F[(size_t)i * 3 + 1] - C[c * 3 + 1], dz = F[(size_t)i * 3 + 2] - C[c * 3 + 2];
    const float d = (dx * dx + dy * dy) + dz * dz;
    keys[p] = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)((uint32_t)i * KM_MAXK + (uint32_t)c);
}

// bitonic network, ascending.  Local form: one workgroup sorts / merges KM_LOCAL keys in LDS -- all strides j < KM_LOCAL of stage kk
// (kk = 0: the full local sort, every stage up to KM_LOCAL).  Global form: one compare-exchange step (kk, j >= KM_LOCAL / 2 ... ).
__global__ __launch_bounds__(KM_T) void km_sort_local_kernel(unsigned long long *__restrict__ keys, size_t kk, const KmState *st)
{
    if (st->done) return;
    __shared__ unsigned long long s[KM_LOCAL];
    const size_t base = (size_t)blockIdx.x * KM_LOCAL;
    for (int e = threadIdx.x; e < KM_LOCAL; e += KM_T) s[e] = keys[base + e];
    __syncthreads();
    auto step = [&](size_t stage, int j) {
        for (int t = threadIdx.x; t < KM_LOCAL / 2; t += KM_T) {
            const int lo = 2 * t - (t & (j - 1));          // element with bit j clear
            const int hi = lo + j;
            const bool asc = (((base + lo) & stage) == 0);
            const unsigned long long a = s[lo], b = s[hi];
            if ((a > b) == asc) {
                s[lo] = b;
                s[hi] = a;
            }
        }
        __syncthreads();
    };
    if (kk == 0) {
        for (size_t stage = 2; stage <= KM_LOCAL; stage <<= 1)
            for (int j = (int)(stage >> 1); j > 0; j >>= 1) step(stage, j);
    } else {
        for (int j = KM_LOCAL / 2; j > 0; j >>= 1) step(kk, j);
    }
    for (int e = threadIdx.x; e < KM_LOCAL; e += KM_T) keys[base + e] = s[e];
}

__global__ void km_sort_global_kernel(unsigned long long *__restrict__ keys, size_t n_keys, size_t kk, size_t j, const KmState *st)
{
    if (st->done) return;
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= n_keys / 2) return;
    const size_t lo = 2 * t - (t & (j - 1)), hi = lo + j;
    const bool asc = ((lo & kk) == 0);
    const unsigned long long a = keys[lo], b = keys[hi];
    if ((a > b) == asc) {
        keys[lo] = b;
        keys[hi] = a;
    }
}

// the capacity sweep: pairs in ascending order, a pair is taken when its point is free and its cluster below the cap.
// One wave.  A batch of 64 consecutive pairs is FILTERED in parallel against the state before the batch (point already placed, cluster
// already full: both only ever become more true), the survivors are then replayed in order with wave-uniform control flow, re-checking
// against the live state in LDS.  Most pairs die in the filter, so the serial part is a few iterations per point.
__global__ __launch_bounds__(64) void km_sweep_kernel(const unsigned long long *__restrict__ keys, size_t n_pairs, int n, int k, int size_min, int size_max,
                                                      int *__restrict__ labels, int *__restrict__ counts_out, const KmState *st)
{
    if (st->done) return;
    __shared__ uint32_t taken[65536 / 32];
    __shared__ int s_cnt[KM_MAXK];
    const int lane = threadIdx.x;
    for (int w = lane; w < 65536 / 32; w += 64) taken[w] = 0u;
    if (lane < KM_MAXK) s_cnt[lane] = 0;
    __syncthreads();
    int assigned = 0;                                             // wave-uniform
    for (int phase = 0; phase < 2; ++phase) {
        const int cap = phase == 0 ? size_min : size_max;
        const int target = phase == 0 ? (size_min * k < n ? size_min * k : n) : n;
        for (size_t p0 = 0; p0 < n_pairs && assigned < target; p0 += 64) {
            const bool valid = p0 + lane < n_pairs;
            const uint32_t id = valid ? (uint32_t)keys[p0 + lane] : 0u;
            const int i = (int)(id / KM_MAXK), c = (int)(id % KM_MAXK);
            const bool live = valid && !((taken[i >> 5] >> (i & 31)) & 1u) && s_cnt[c] < cap;
            unsigned long long mask = __ballot(live);
            while (mask && assigned < target) {
                const int b = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const int ib = __builtin_amdgcn_readlane(i, b), cb = __builtin_amdgcn_readlane(c, b);
                const uint32_t word = taken[ib >> 5];
                const int cc = s_cnt[cb];
                if (((word >> (ib & 31)) & 1u) || cc >= cap) continue;      // placed / filled earlier in this batch
                if (lane == 0) {
                    taken[ib >> 5] = word | (1u << (ib & 31));
                    s_cnt[cb] = cc + 1;
                    labels[ib] = cb;
                }
                __builtin_amdgcn_wave_barrier();
                ++assigned;
            }
        }
    }
    if (lane < k) counts_out[lane] = s_cnt[lane];
}

// new centres = cluster means (block = cluster; double sums in a fixed order), centre shift, inertia of the assignment w.r.t. the OLD centres
__global__ __launch_bounds__(KM_T) void km_update_kernel(const float *__restrict__ F, const int *__restrict__ labels, int n, int k, float *__restrict__ C,
                                                         double *__restrict__ part, const KmState *st)
{
    if (st->done) return;
    __shared__ double red[4][KM_T];
    const int c = blockIdx.x, tid = threadIdx.x;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, cnt = 0.0;
    for (int i = tid; i < n; i += KM_T) {
        if (labels[i] == c) {
            s0 += (double)F[(size_t)i * 3];
            s1 += (double)F[(size_t)i * 3 + 1];
            s2 += (double)F[(size_t)i * 3 + 2];
            cnt += 1.0;
        }
    }
    red[0][tid] = s0; red[1][tid] = s1; red[2][tid] = s2; red[3][tid] = cnt;
    __syncthreads();
    for (int w = KM_T / 2; w > 0; w >>= 1) {
        if (tid < w)
            for (int f = 0; f < 4; ++f) red[f][tid] += red[f][tid + w];
        __syncthreads();
    }
    if (tid == 0) {
        const double m = red[3][0] > 0.0 ? red[3][0] : 1.0;
        double sh = 0.0;
        for (int f = 0; f < 3; ++f) {
            const float nc = (float)(red[f][0] / m);
            const double d = (double)nc - (double)C[c * 3 + f];
            sh += d * d;
            C[c * 3 + f] = nc;
        }
        part[c] = sh;
    }
}

// after the update of every cluster: shift -> convergence flag; the final pass of an init (final != 0) computes the inertia of the labels
// against the centres they were assigned with and keeps the best init's labels / centres
__global__ __launch_bounds__(KM_T) void km_control_kernel(const float *__restrict__ F, const int *__restrict__ labels, const float *__restrict__ C,
                                                          const double *__restrict__ part, int n, int k, int final_pass, int init, int max_iter,
                                                          int *__restrict__ best_labels, float *__restrict__ best_C, KmState *st)
{
    __shared__ double red[KM_T];
    __shared__ int s_better;
    const int tid = threadIdx.x;
    if (!final_pass) {
        if (tid == 0 && !st->done) {
            double sh = 0.0;
            for (int c = 0; c < k; ++c) sh += part[c];
            st->shift = (float)sh;
            st->iters += 1;
            if (sh <= (double)st->tol_abs || st->iters >= max_iter) st->done = 1;
        }
        return;
    }
    double q = 0.0;
    for (int i = tid; i < n; i += KM_T) {
        const int c = labels[i];
        const float dx = F[(size_t)i * 3] - C[c * 3], dy = F[(size_t)i * 3 + 1] - C[c * 3 + 1], dz = F[(size_t)i * 3 + 2] - C[c * 3 + 2];
        q += (double)((dx * dx + dy * dy) + dz * dz);
    }
    red[tid] = q;
    __syncthreads();
    for (int w = KM_T / 2; w > 0; w >>= 1) {
        if (tid < w) red[tid] += red[tid + w];
        __syncthreads();
    }
    if (tid == 0) {
        st->inertia = red[0];
        s_better = red[0] < st->best_inertia;
        if (s_better) {
            st->best_inertia = red[0];
            st->best_init = init;
        }
    }
    __syncthreads();
    if (s_better) {
        for (int i = tid; i < n; i += KM_T) best_labels[i] = labels[i];
        for (int e = tid; e < 3 * k; e += KM_T) best_C[e] = C[e];
    }
}

// the loop of an init is over: open the gate again for the final assignment
__global__ void km_reopen_kernel(KmState *st) { st->done = 0; }

size_t km_pow2(size_t x)
{
    size_t p = KM_LOCAL;
    while (p < x) p <<= 1;
    return p;
}

struct KmWs {
    unsigned long long *keys;
    float *C, *dmin;
    double *part;
    int *labels, *counts;
    KmState *st;
    size_t bytes;
};

void km_carve(int n, int k, void *base, KmWs &w)
{
    size_t off = 0;
    auto take = [&](size_t nbytes) {
        void *p = base ? static_cast<char *>(base) + off : nullptr;
        off += align_up(nbytes, 256);
        return p;
    };
    w.keys = static_cast<unsigned long long *>(take(km_pow2((size_t)n * k) * 8));
    w.C = static_cast<float *>(take(KM_MAXK * 3 * 4));
    w.dmin = static_cast<float *>(take((size_t)n * 4));
    w.part = static_cast<double *>(take(KM_MAXK * 8));
    w.labels = static_cast<int *>(take((size_t)n * 4));
    w.counts = static_cast<int *>(take(KM_MAXK * 4));
    w.st = static_cast<KmState *>(take(sizeof(KmState)));
    w.bytes = off;
}

int km_sort(unsigned long long *keys, size_t n_keys, const KmState *st, hipStream_t s)
{
    const unsigned blocks = (unsigned)(n_keys / KM_LOCAL);
    hipLaunchKernelGGL(km_sort_local_kernel, dim3(blocks), dim3(KM_T), 0, s, keys, (size_t)0, st);
    for (size_t kk = 2 * (size_t)KM_LOCAL; kk <= n_keys; kk <<= 1) {
        for (size_t j = kk >> 1; j >= KM_LOCAL; j >>= 1)
            hipLaunchKernelGGL(km_sort_global_kernel, dim3((unsigned)((n_keys / 2 + 255) / 256)), dim3(256), 0, s, keys, n_keys, kk, j, st);
        hipLaunchKernelGGL(km_sort_local_kernel, dim3(blocks), dim3(KM_T), 0, s, keys, kk, st);
    }
    return check_launch("km_sort");
}

}  // namespace
}  // namespace ampnet

using namespace ampnet;

extern "C" size_t ampnet_kmeans_workspace_bytes(int n, int k)
{
    if (n < 1 || k < 1 || k > KM_MAXK) return 0;
    KmWs w;
    km_carve(n, k, nullptr, w);
    return w.bytes;
}

extern "C" int ampnet_kmeans_balanced_f32(const float *feat, int n, int k, int size_min, int size_max, int n_init, int max_iter, float tol,
                                          uint32_t seed, int32_t *labels, float *centres, double *inertia, void *workspace, size_t workspace_bytes,
                                          void *stream)
{
    AMPNET_REQUIRE(feat && labels && centres && workspace, "ampnet_kmeans_balanced_f32: null pointer");
    AMPNET_REQUIRE(n >= 1 && n <= 65536 && k >= 1 && k <= KM_MAXK && k <= n, "ampnet_kmeans_balanced_f32: n=%d (<= 65536), k=%d (<= %d)", n, k, KM_MAXK);
    AMPNET_REQUIRE(size_min >= 0 && (long)size_min * k <= n && size_max >= 1 && (long)size_max * k >= n && size_min <= size_max,
                   "ampnet_kmeans_balanced_f32: sizes [%d, %d] cannot hold %d points in %d clusters", size_min, size_max, n, k);
    AMPNET_REQUIRE(n_init >= 1 && max_iter >= 1 && tol >= 0.f, "ampnet_kmeans_balanced_f32: n_init=%d max_iter=%d tol=%g", n_init, max_iter, (double)tol);
    KmWs w;
    km_carve(n, k, workspace, w);
    if (w.bytes > workspace_bytes) return fail(AMPNET_E_WORKSPACE, "ampnet_kmeans_balanced_f32: workspace %zu B < %zu B", workspace_bytes, w.bytes);
    hipStream_t st = (hipStream_t)stream;
    const size_t n_pairs = (size_t)n * k, n_keys = km_pow2(n_pairs);
    hipLaunchKernelGGL(km_prepare_kernel, dim3(1), dim3(KM_T), 0, st, feat, n, tol, w.st);
    auto assign = [&]() -> int {
        hipLaunchKernelGGL(km_keys_kernel, dim3((unsigned)((n_keys + 255) / 256)), dim3(256), 0, st, feat, w.C, n, k, n_keys, w.keys, w.st);
        int rc = km_sort(w.keys, n_keys, w.st, st);
        if (rc != AMPNET_OK) return rc;
        hipLaunchKernelGGL(km_sweep_kernel, dim3(1), dim3(64), 0, st, w.keys, n_pairs, n, k, size_min, size_max, w.labels, w.counts, w.st);
        return check_launch("km_sweep_kernel");
    };
    for (int init = 0; init < n_init; ++init) {
        hipLaunchKernelGGL(km_seed_kernel, dim3(1), dim3(KM_T), 0, st, feat, n, k, seed, init, w.C, w.dmin, w.st);
        for (int it = 0; it < max_iter; ++it) {
            int rc = assign();
            if (rc != AMPNET_OK) return rc;
            hipLaunchKernelGGL(km_update_kernel, dim3(k), dim3(KM_T), 0, st, feat, w.labels, n, k, w.C, w.part, w.st);
            hipLaunchKernelGGL(km_control_kernel, dim3(1), dim3(KM_T), 0, st, feat, w.labels, w.C, w.part, n, k, 0, init, max_iter, labels, centres, w.st);
        }
        // final assignment with the final centres (labels and centres consistent), inertia, best-of-n_init bookkeeping
        hipLaunchKernelGGL(km_reopen_kernel, dim3(1), dim3(1), 0, st, w.st);
        int rc = assign();
        if (rc != AMPNET_OK) return rc;
        hipLaunchKernelGGL(km_control_kernel, dim3(1), dim3(KM_T), 0, st, feat, w.labels, w.C, w.part, n, k, 1, init, max_iter, labels, centres, w.st);
    }
    if (inertia && hipMemcpyAsync(inertia, &w.st->best_inertia, sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return fail(AMPNET_E_LAUNCH, "ampnet_kmeans_balanced_f32: copy of the inertia failed");
    return check_launch("ampnet_kmeans_balanced_f32");
}
