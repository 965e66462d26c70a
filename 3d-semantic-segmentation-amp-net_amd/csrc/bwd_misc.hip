// bwd_misc.hip -- the small backward kernels: BatchNorm-backward constants, MaxPool backward, a plain tiled
// SGEMM for the [Q, <=4096] T-Net FC / attention projections, and the K<=12 input layers' weight gradients.
#include <cstdlib>
#include <cstdint>
#include "bwd_misc.h"

namespace ampnet {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ----------------------------------------------------------------------------------------------------
// bn_bwd_finalize: block = (slot, 64 channels) x 4 groups over the slot's partials, fixed order
// ----------------------------------------------------------------------------------------------------
// CPB channels per block x (1024 / CPB) groups over the partials: 64 x 16 by default; 16 x 64 when there are few (slot, 64-channel) blocks
// and many partials (the head: one slot, 64 channels, one partial per workgroup of head_out_bwd -- 49 us in ONE block before)
template <int CPB>
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(BnBwdFinalize a)
{
    constexpr int BFIN_G = 1024 / CPB;
    __shared__ double ra[BFIN_G][CPB], rb[BFIN_G][CPB];
    __shared__ int rows_s;
    const int slot = blockIdx.x, cl = threadIdx.x % CPB, g = threadIdx.x / CPB;
    const int c = blockIdx.y * CPB + cl;
    const int per_slot = (a.Q - slot + a.n_slots - 1) / a.n_slots;
    const int part_per_slot = a.part_Q > 0 ? (a.part_Q - slot + a.n_slots - 1) / a.n_slots : per_slot;
    double sa = 0.0, sb = 0.0;
    if (c < a.C) {
        const int total = part_per_slot * a.chunks;
        for (int e0 = g; e0 < total; e0 += BFIN_G * 8) {          // eight independent loads per array in flight
            float va[8], vb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + BFIN_G * u;
                const int q = slot + (e / a.chunks) * a.n_slots;
                const size_t o = (size_t)(q * a.chunks + e % a.chunks) * a.C + c;
                va[u] = e < total ? a.part_a[o] : 0.f;
                vb[u] = e < total ? a.part_b[o] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                sa += (double)va[u];
                sb += (double)vb[u];
            }
        }
    }
    ra[g][cl] = sa;
    rb[g][cl] = sb;
    if (threadIdx.x == 0) rows_s = 0;
    __syncthreads();
    if (a.uniform_rows > 0) {
        if (threadIdx.x == 0) rows_s = per_slot * a.uniform_rows;
    } else {
        int rows = 0;
        for (int i = threadIdx.x; i < per_slot; i += 1024) {
            const int q = slot + i * a.n_slots;
            rows += a.win_off[q + 1] - a.win_off[q];
        }
        if (rows) atomicAdd(&rows_s, rows);       // integer, order-independent
    }
    __syncthreads();
    if (g == 0 && c < a.C) {
        double A = 0.0, Bs = 0.0;
#pragma unroll
        for (int k = 0; k < BFIN_G; ++k) {
            A += ra[k][cl];
            Bs += rb[k][cl];
        }
        const double n = (double)rows_s;
        const size_t o = (size_t)slot * a.C + c;
        const double invstd = a.invstd[o], mean = a.mean[o];
        const double s = (double)a.gamma[c] * invstd;
        const double p2 = -s * invstd * Bs / n;
        a.P1[o] = (float)s;
        a.P2[o] = (float)p2;
        a.P3[o] = (float)(-s * A / n - p2 * mean);
        a.slot_ab[o * 2 + 0] = (float)A;
        a.slot_ab[o * 2 + 1] = (float)Bs;
    }
}

int bn_bwd_finalize(const BnBwdFinalize &a, hipStream_t st)
{
    AMPNET_REQUIRE(a.part_a && a.part_b && (a.win_off || a.uniform_rows > 0) && a.gamma && a.mean && a.invstd && a.P1 && a.P2 && a.P3 && a.slot_ab, "bn_bwd_finalize: null pointer");
    const long parts = (long)(a.part_Q > 0 ? cdiv(a.part_Q, a.n_slots) : cdiv(a.Q, a.n_slots)) * a.chunks;
    if (a.n_slots * cdiv(a.C, 64) < 8 && parts >= 512)
        hipLaunchKernelGGL(bn_bwd_finalize_kernel<16>, dim3(a.n_slots, cdiv(a.C, 16)), dim3(1024), 0, st, a);
    else
        hipLaunchKernelGGL(bn_bwd_finalize_kernel<64>, dim3(a.n_slots, cdiv(a.C, 64)), dim3(1024), 0, st, a);
    int rc = check_launch("bn_bwd_finalize_kernel");
    if (rc == AMPNET_OK && sync_bn_on())       // global batch: the constants again, from the all-reduced sums (slot_ab stays the rank's own)
        rc = sync_bn_bwd_constants(a.slot_ab, a.win_off, a.Q, a.n_slots, a.uniform_rows, a.C, a.gamma, nullptr, a.mean, a.invstd, a.P1, a.P2, a.P3, st);
    return rc;
}

constexpr int MAX_BN_ITEMS = 24;
struct BnGradArgs {
    BnGradItem it[MAX_BN_ITEMS];
    int n;
};

__global__ void bn_param_grads_kernel(BnGradArgs a)
{
    const BnGradItem it = a.it[blockIdx.x];
    for (int c = threadIdx.x; c < it.C; c += blockDim.x) {
        float db = 0.f, dg = 0.f;
        for (int s = 0; s < it.n_slots; ++s) {
            db += it.slot_ab[((size_t)s * it.C + c) * 2 + 0];
            dg += it.slot_ab[((size_t)s * it.C + c) * 2 + 1];
        }
        it.dbeta[c] = db;
        it.dgamma[c] = dg;
    }
}

int bn_param_grads(const BnGradItem *items, int n, hipStream_t st)
{
    AMPNET_REQUIRE(n >= 1 && n <= MAX_BN_ITEMS, "bn_param_grads: %d items", n);
    BnGradArgs a;
    for (int i = 0; i < n; ++i) a.it[i] = items[i];
    a.n = n;
    hipLaunchKernelGGL(bn_param_grads_kernel, dim3(n), dim3(256), 0, st, a);
    return check_launch("bn_param_grads_kernel");
}

// ----------------------------------------------------------------------------------------------------
// pool_bwd: block = (slot, 64 channels); loops the slot's windows.  The only rows with a gradient are the argmax
// rows and the forward kept their pre-BatchNorm values (zext), so the whole BatchNorm-backward reduction of the
// pooled layer is B table look-ups per channel: the [rows, 256] output is never needed again.
// ----------------------------------------------------------------------------------------------------
constexpr int PB_G = 16;       // window groups per channel (block = 64 channels x 16 groups)

__global__ __launch_bounds__(64 * PB_G) void pool_bwd_kernel(PoolBwd a)
{
    __shared__ double rA[PB_G][64], rB[PB_G][64];
    __shared__ int rR[PB_G];
    const int slot = blockIdx.x, cl = threadIdx.x & 63, g = threadIdx.x >> 6, c = blockIdx.y * 64 + cl;
    const bool ok = c < a.C;
    const size_t so = (size_t)slot * a.C + (ok ? c : 0);
    const float sc = a.scale[so], sh = a.shift[so], mean = a.mean[so], invstd = a.invstd[so];
    const int per_slot = (a.Q - slot + a.n_slots - 1) / a.n_slots;
    double A = 0.0, Bs = 0.0;
    int rows = 0;
    // four windows per trip, every load issued before the first use
    for (int i0 = g; i0 < per_slot; i0 += PB_G * 4) {
        int rowv[4], prw[4];
        float zv[4], dv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + PB_G * u;
            const bool live = i < per_slot && ok;
            const int q = slot + (live ? i : 0) * a.n_slots;
            prw[u] = a.slot_major ? (q % a.n_slots) * (a.Q / a.n_slots) + q / a.n_slots : q;
            rowv[u] = live ? a.arg[(size_t)q * a.C + c] : -2;
            zv[u] = live ? a.zext[(size_t)q * a.C + c] : 0.f;
            dv[u] = live ? a.d_pooled[(size_t)prw[u] * a.C + c] : 0.f;
            if (cl == 0 && i < per_slot) rows += a.win_off[q + 1] - a.win_off[q];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (rowv[u] == -2) continue;
            float d = 0.f;
            if (rowv[u] >= 0 && fmaf(zv[u], sc, sh) > 0.f) {
                d = dv[u];
                A += (double)d;
                Bs += (double)d * (double)((zv[u] - mean) * invstd);
            }
            a.dpm[(size_t)prw[u] * a.C + c] = d;
        }
    }
    rA[g][cl] = A;
    rB[g][cl] = Bs;
    if (cl == 0) rR[g] = rows;
    __syncthreads();
    if (g != 0 || !ok) return;
    A = 0.0;
    Bs = 0.0;
    rows = 0;
#pragma unroll
    for (int k = 0; k < PB_G; ++k) {
        A += rA[k][cl];
        Bs += rB[k][cl];
        rows += rR[k];
    }
    const double n = (double)rows;
    const double gamma_invstd = (double)sc;      // scale = gamma * invstd
    const double p2 = -gamma_invstd * (double)invstd * Bs / n;
    a.P1[so] = sc;
    a.P2[so] = (float)p2;
    a.P3[so] = (float)(-gamma_invstd * A / n - p2 * (double)mean);
    a.slot_ab[so * 2 + 0] = (float)A;
    a.slot_ab[so * 2 + 1] = (float)Bs;
}

int pool_bwd(const PoolBwd &a, hipStream_t st)
{
    AMPNET_REQUIRE(a.d_pooled && a.arg && a.zext && a.scale && a.shift && a.mean && a.invstd && a.win_off && a.dpm && a.P1 && a.P2 && a.P3 && a.slot_ab, "pool_bwd: null pointer");
    hipLaunchKernelGGL(pool_bwd_kernel, dim3(a.n_slots, cdiv(a.C, 64)), dim3(64 * PB_G), 0, st, a);
    int rc = check_launch("pool_bwd_kernel");
    if (rc == AMPNET_OK && sync_bn_on())
        rc = sync_bn_bwd_constants(a.slot_ab, a.win_off, a.Q, a.n_slots, 0, a.C, nullptr, a.scale, a.mean, a.invstd, a.P1, a.P2, a.P3, st);
    return rc;
}

// ----------------------------------------------------------------------------------------------------
// sgemm_small: 32 x 32 output tile, 32-deep LDS slices, 2 x 2 outputs per thread (VALU fp32); the next slice is
// fetched into registers while the current one is multiplied.  The problems are tiny (M or N = a few hundred
// rows): small tiles keep every CU busy, which matters more than per-block efficiency.
// ----------------------------------------------------------------------------------------------------
// Split-K inside the workgroup: the problems are latency-bound (K of a few hundred, a few dozen workgroups), so SK = 4 groups
// of 256 threads each walk a quarter of K with their own LDS tiles and the partial 32 x 32 tiles are summed through LDS in a
// fixed order -- a quarter of the dependent (load -> barrier -> multiply) trips per launch.
constexpr int SG_SK = 4;
constexpr int SG_KD = 32;               // depth of a k slice per group and trip.  (64 made the hot path's launches ~10 % shorter before they moved to
                                       // sgemm_mfma below; this kernel is now the AMPNET_SGEMM_VALU=1 A/B form only)

struct SgProblem {
    int M, N, K, ta, tb, lda, ldb, ldc, accumulate;
    const float *A, *B;
    float *C;
    const float *mul = nullptr;        // [M, N] with leading dimension ldc: C = (product) * mul elementwise (sgemm_mfma only)
    const float *kscale = nullptr;     // [K]: op(A)[m][k] is multiplied by kscale[k] as it is loaded (sgemm_mfma only): A^T diag(kscale) B in one product
    float *db = nullptr;               // [M]: row sums of op(A) over K (the bias gradient G^T 1 of a weight-gradient problem), taken by the
                                       // workgroups of the first column of tiles from the A tile they stage anyway: no extra problem, no extra launch
};
constexpr int SG_MAX_PROBLEMS = 10;
struct SgArgs {
    SgProblem p[SG_MAX_PROBLEMS];      // blockIdx.z picks the problem: independent products of one backward step share a launch
};

__global__ __launch_bounds__(256 * SG_SK) void sgemm_small_kernel(SgArgs args)
{
    __shared__ float sA[SG_SK][SG_KD][33], sB[SG_SK][SG_KD][33];     // [group][k][m], [group][k][n]
    const SgProblem &g = args.p[blockIdx.z];
    const int M = g.M, N = g.N, K = g.K, lda = g.lda, ldb = g.ldb;
    const bool TA = g.ta != 0, TB = g.tb != 0;
    const float *__restrict__ A = g.A, *__restrict__ B = g.B;
    const int grp = threadIdx.x >> 8, tid = threadIdx.x & 255, tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    if (m0 >= M || n0 >= N) return;                             // the grid covers the larger of the two problems (uniform per block)
    // group g owns the k tiles g, g + SK, ... (SG_KD deep each)
    float acc[2][2] = {};
    float rs0 = 0.f, rs1 = 0.f;                                 // row sums of op(A) (g.db): rows ty and ty + 16 of the tile, threads tx == 0 of tile column 0
    const bool do_rs = g.db != nullptr && blockIdx.x == 0 && tx == 0;
    constexpr int NF = SG_KD * 32 / 256;                        // elements of each operand a thread stages per trip
    float ra[NF], rb[NF];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int e = tid + 256 * i;
            const int m = TA ? e % 32 : e / SG_KD, k = TA ? e / 32 : e % SG_KD;
            const int gm = m0 + m, gk = k0 + k;
            ra[i] = (gm < M && gk < K) ? (TA ? A[(size_t)gk * lda + gm] : A[(size_t)gm * lda + gk]) : 0.f;
            const int n = TB ? e / SG_KD : e % 32, k2 = TB ? e % SG_KD : e / 32;
            const int gn = n0 + n, gk2 = k0 + k2;
            rb[i] = (gn < N && gk2 < K) ? (TB ? B[(size_t)gn * ldb + gk2] : B[(size_t)gk2 * ldb + gn]) : 0.f;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int e = tid + 256 * i;
            if (TA) sA[grp][e / 32][e % 32] = ra[i]; else sA[grp][e % SG_KD][e / SG_KD] = ra[i];
            if (TB) sB[grp][e % SG_KD][e / SG_KD] = rb[i]; else sB[grp][e / 32][e % 32] = rb[i];
        }
    };
    const int ktiles = (K + SG_KD - 1) / SG_KD;
    const int trips = (ktiles + SG_SK - 1) / SG_SK;            // the same for every group (the barriers are workgroup-wide)
    fetch(SG_KD * grp);
    for (int i = 0; i < trips; ++i) {
        const int kt = grp + i * SG_SK;
        __syncthreads();
        stash();
        __syncthreads();
        if (i + 1 < trips) fetch(SG_KD * (kt + SG_SK));        // tiles past K load zeros
        if (kt < ktiles) {
#pragma unroll
            for (int k = 0; k < SG_KD; ++k) {
                const float a0 = sA[grp][k][ty], a1 = sA[grp][k][ty + 16], b0 = sB[grp][k][tx], b1 = sB[grp][k][tx + 16];
                acc[0][0] = fmaf(a0, b0, acc[0][0]);
                acc[0][1] = fmaf(a0, b1, acc[0][1]);
                acc[1][0] = fmaf(a1, b0, acc[1][0]);
                acc[1][1] = fmaf(a1, b1, acc[1][1]);
                if (do_rs) {
                    rs0 += a0;
                    rs1 += a1;
                }
            }
        }
    }
    // sum the SK partial tiles in group order
    __syncthreads();
    float *red = &sA[0][0][0];                                  // [SK][4][256]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) red[(grp * 4 + i * 2 + j) * 256 + tid] = acc[i][j];
    __syncthreads();
    if (g.db && blockIdx.x == 0) {                              // uniform per block: the SK groups' row sums, group order
        __syncthreads();
        float *rsum = &sB[0][0][0];                             // [SK][32]
        if (tx == 0) {
            rsum[grp * 32 + ty] = rs0;
            rsum[grp * 32 + ty + 16] = rs1;
        }
        __syncthreads();
        if (grp == 0 && tid < 32 && m0 + tid < M) {
            float v = 0.f;
#pragma unroll
            for (int g2 = 0; g2 < SG_SK; ++g2) v += rsum[g2 * 32 + tid];
            g.db[m0 + tid] = v;
        }
    }
    if (grp != 0) return;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int gm = m0 + ty + 16 * i;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gn = n0 + tx + 16 * j;
            if (gm < M && gn < N) {
                float v = 0.f;
#pragma unroll
                for (int g2 = 0; g2 < SG_SK; ++g2) v += red[(g2 * 4 + i * 2 + j) * 256 + tid];
                float *d = g.C + (size_t)gm * g.ldc + gn;
                *d = g.accumulate ? *d + v : v;
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------------
// sgemm_mfma: the same problems on the fp32 matrix pipe, without LDS staging and without dependent trips.
// One workgroup of 16 waves per 32 x 32 output tile; the waves split K between them (contiguous ranges of 8-deep chunks) and every
// wave feeds v_mfma_f32_32x32x2_f32 STRAIGHT from global memory: lane (r, h) supplies op(A)[m0 + r][k] and op(B)[k][n0 + r] for the
// k values 8 c + 4 h + i, i = 0..3, of chunk c (any pairing of k values is a valid contraction order as long as both operands use the
// same one).  An operand whose k runs along ROWS of the stored matrix (G and X of a weight gradient G^T X, W of a data gradient G W)
// is four coalesced 128-byte row segments per chunk; one whose k is contiguous in memory (G of G W) is one 16-byte load per lane.
// All loads of up to eight chunks per wave are issued before the first MFMA: K = 576 (the rows of a per-window problem) is ONE memory
// round trip and <= 20 MFMAs per wave, where the VALU kernel above makes three load -> barrier -> multiply trips that are bound by
// its LDS reads (4 ds_read_b32 per 4 FMAs, two 1024-thread workgroups per CU): 19 .. 46 us per launch there, the launch floor here.
// The 16 partial tiles are summed through LDS in wave order: fixed order, no atomics, bitwise reproducible.
// ----------------------------------------------------------------------------------------------------
constexpr int SGM_WAVES = 16;
constexpr int SGM_PASS = 8;              // chunks of 8 k values a wave has in flight per pass

typedef float sg_f32x16 __attribute__((ext_vector_type(16)));
typedef float sg_f32x4 __attribute__((ext_vector_type(4)));

// four k values 8 c + 4 h + i of row / column x of an operand; kmaj: k contiguous in memory (element (x, k) at base[x * ld + k])
__device__ __forceinline__ sg_f32x4 sgm_fetch(const float *__restrict__ base, int ld, bool kmaj, bool vec, int x, int X, int k, int K)
{
    sg_f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (x >= X) return v;
    if (kmaj) {
        const float *p = base + (size_t)x * ld + k;
        if (vec && k + 3 < K) return *reinterpret_cast<const sg_f32x4 *>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (k + i < K) v[i] = p[i];
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (k + i < K) v[i] = base[(size_t)(k + i) * ld + x];
    }
    return v;
}

__global__ __launch_bounds__(64 * SGM_WAVES) void sgemm_mfma_kernel(SgArgs args)
{
    extern __shared__ __attribute__((aligned(16))) float sgm_smem[];
    float (*red)[16][64] = reinterpret_cast<float (*)[16][64]>(sgm_smem);                       // [SGM_WAVES][16][64]
    float (*rsum)[32] = reinterpret_cast<float (*)[32]>(sgm_smem + SGM_WAVES * 16 * 64);        // [SGM_WAVES][32]
    const SgProblem &g = args.p[blockIdx.z];
    const int M = g.M, N = g.N, K = g.K;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    if (m0 >= M || n0 >= N) return;                             // uniform per block: the grid covers the largest problem of the launch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const bool akm = g.ta == 0, bkm = g.tb != 0;                // k contiguous in memory
    const bool avec = akm && (g.lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.A) & 15) == 0);
    const bool bvec = bkm && (g.ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.B) & 15) == 0);
    const int chunks = (K + 7) / 8;
    const int per_wave = (chunks + SGM_WAVES - 1) / SGM_WAVES;
    const int c_begin = min(wave * per_wave, chunks), c_end = min(c_begin + per_wave, chunks);
    const bool do_rs = g.db != nullptr && blockIdx.x == 0;
    sg_f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    float rs = 0.f;
    for (int c0 = c_begin; c0 < c_end; c0 += SGM_PASS) {
        sg_f32x4 av[SGM_PASS], bv[SGM_PASS];
#pragma unroll
        for (int u = 0; u < SGM_PASS; ++u) {
            if (c0 + u < c_end) {                                // wave-uniform
                const int k = 8 * (c0 + u) + 4 * h;
                av[u] = sgm_fetch(g.A, g.lda, akm, avec, m0 + r, M, k, K);
                bv[u] = sgm_fetch(g.B, g.ldb, bkm, bvec, n0 + r, N, k, K);
                if (g.kscale) {                                  // uniform per problem
#pragma unroll
                    for (int i = 0; i < 4; ++i) av[u][i] *= (k + i < K) ? g.kscale[k + i] : 0.f;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < SGM_PASS; ++u) {
            if (c0 + u < c_end) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][i], bv[u][i], acc, 0, 0, 0);
                if (do_rs) rs += (av[u][0] + av[u][1]) + (av[u][2] + av[u][3]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) red[wave][e][lane] = acc[e];
    if (do_rs) {
        rs += __shfl_xor(rs, 32);
        if (h == 0) rsum[wave][r] = rs;
    }
    __syncthreads();
    // thread -> one output element: e = tid >> 6 (accumulator register), lane as in the MFMA output map
    {
        const int e = tid >> 6;
        const int gm = m0 + (e & 3) + 8 * (e >> 2) + 4 * h, gn = n0 + r;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < SGM_WAVES; ++w) v += red[w][e][lane];
        if (gm < M && gn < N) {
            float *d = g.C + (size_t)gm * g.ldc + gn;
            if (g.mul) v *= g.mul[(size_t)gm * g.ldc + gn];
            *d = g.accumulate ? *d + v : v;
        }
    }
    if (do_rs && tid < 32 && m0 + tid < M) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < SGM_WAVES; ++w) v += rsum[w][tid];
        g.db[m0 + tid] = v;
    }
}

static int sgemm_launch(const SgProblem *probs, int n, hipStream_t st)
{
    AMPNET_REQUIRE(n >= 1 && n <= SG_MAX_PROBLEMS, "sgemm_small: %d problems", n);
    SgArgs a;
    int gx = 1, gy = 1;
    double flops = 0.0, bytes = 0.0;
    for (int i = 0; i < n; ++i) {
        const SgProblem &p = probs[i];
        AMPNET_REQUIRE(p.A && p.B && p.C && p.M >= 1 && p.N >= 1 && p.K >= 1, "sgemm_small: bad arguments");
        a.p[i] = p;
        gx = cdiv(p.N, 32) > gx ? cdiv(p.N, 32) : gx;
        gy = cdiv(p.M, 32) > gy ? cdiv(p.M, 32) : gy;
        flops += 2.0 * p.M * p.N * p.K;
        bytes += 4.0 * ((double)p.M * p.K + (double)p.K * p.N + (double)p.M * p.N);
    }
    if (n == 1) a.p[1] = a.p[0];
    // AMPNET_SGEMM_VALU=1: the VALU kernel (A/B timing, tests/test_small_gemm_gpu.py compares the two)
    static const bool valu_env = [] { const char *e = getenv("AMPNET_SGEMM_VALU"); return e && e[0] == '1'; }();
    bool valu = valu_env;
    for (int i = 0; i < n; ++i) valu = valu && !probs[i].mul && !probs[i].kscale;      // the multiply epilogue / k scaling live in the matrix-core kernel only
    if (valu) {
        ProfScope prof("sgemm_small", flops, bytes, st);
        hipLaunchKernelGGL(sgemm_small_kernel, dim3(gx, gy, n), dim3(256 * SG_SK), 0, st, a);
        return check_launch("sgemm_small_kernel");
    }
    constexpr size_t lds = (size_t)(SGM_WAVES * 16 * 64 + SGM_WAVES * 32) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(sgemm_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(AMPNET_E_LAUNCH, "sgemm_mfma: hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
        attr_set = true;
    }
    ProfScope prof("sgemm_mfma", flops, bytes, st);
    hipLaunchKernelGGL(sgemm_mfma_kernel, dim3(gx, gy, n), dim3(64 * SGM_WAVES), lds, st, a);
    return check_launch("sgemm_mfma_kernel");
}

int sgemm_small(int transA, int transB, int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc,
                int accumulate, hipStream_t st)
{
    SgProblem p = {M, N, K, transA, transB, lda, ldb, ldc, accumulate, A, B, C};
    return sgemm_launch(&p, 1, st);
}

// weight gradient + bias gradient of the LAST layer of a chain (no data gradient wanted): dW [n_out, n_in] = G^T X, db [n_out] = column sums of G
int sgemm_wgrad_bias(int rows, int n_out, int n_in, const float *G, int ldg, const float *X, int ldx, float *dW, int lddw, float *db, hipStream_t st)
{
    SgProblem p = {n_out, n_in, rows, 1, 0, ldg, ldx, lddw, 0, G, X, dW};
    p.db = db;
    return sgemm_launch(&p, 1, st);
}

// the weight gradient dW = G^T X ([N_out, N_in] = [rows, N_out]^T [rows, N_in]) and the data gradient dX = G W of one linear layer
// on [rows, *] activations, one launch (the two products are independent); o.db: the bias gradient rides in the dW problem
int sgemm_linear_bwd(int rows, int n_out, int n_in, const float *G, int ldg, const float *X, int ldx, const float *W, int ldw, float *dW, int lddw,
                     float *dX, int lddx, hipStream_t st, const LinBwdOpt &o)
{
    SgProblem p[2];
    p[0] = {n_out, n_in, rows, 1, 0, ldg, ldx, lddw, 0, G, X, dW};
    p[0].db = o.db;
    p[1] = {rows, n_in, n_out, 0, 0, ldg, ldw, lddx, 0, G, W, dX};
    p[1].mul = o.dx_mul;
    return sgemm_launch(p, 2, st);
}

// the same for a layer with thousands of outputs (the feature T-Net's fc_3: 128 -> 4096): dX = G W has few output tiles and a K of n_out, so
// one workgroup per tile walks 128 k tiles in sequence (114 us for 1.2 GFLOP).  The K range is cut into `splits` problems of the same
// launch, each writing its own partial [rows, n_in]; a fixed-order reduction adds them (no atomics: reproducible).
int sgemm_linear_bwd_ksplit(int rows, int n_out, int n_in, const float *G, int ldg, const float *X, int ldx, const float *W, int ldw, float *dW,
                            int lddw, float *dX, int lddx, float *scratch, int splits, hipStream_t st, const LinBwdOpt &o)
{
    AMPNET_REQUIRE(scratch && splits >= 2 && splits < SG_MAX_PROBLEMS && n_out % splits == 0, "sgemm_linear_bwd_ksplit: %d outputs in %d splits", n_out, splits);
    SgProblem p[SG_MAX_PROBLEMS];
    p[0] = {n_out, n_in, rows, 1, 0, ldg, ldx, lddw, 0, G, X, dW};
    p[0].db = o.db;
    const int kc = n_out / splits;
    for (int c = 0; c < splits; ++c)
        p[1 + c] = {rows, n_in, kc, 0, 0, ldg, ldw, n_in, 0, G + (size_t)c * kc, W + (size_t)c * kc * ldw, scratch + (size_t)c * rows * n_in};
    const int np = 1 + splits;
    int rc = sgemm_launch(p, np, st);
    if (rc != AMPNET_OK) return rc;
    return reduce_windows(scratch, splits, (long)rows * n_in, rows, n_in, n_in, dX, lddx, 0, st);
}

// ----------------------------------------------------------------------------------------------------
// FC-layer helpers on [rows, C] with rows grouped in n_slots contiguous blocks of `per` rows
// ----------------------------------------------------------------------------------------------------
__global__ void fc_act_kernel(const float *__restrict__ z, const float *__restrict__ s, const float *__restrict__ t, int rows, int C,
                              int per, float *__restrict__ act)
{
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * C) return;
    const int row = (int)(i / C), c = (int)(i % C);
    const size_t so = (size_t)(row / per) * C + c;
    act[i] = fmaxf(fmaf(z[i], s[so], t[so]), 0.f);
}

// two activations (the T-Net's a1 [rows, C0] and a2 [rows, C1]) in one launch
__global__ void fc_act_pair_kernel(const float *__restrict__ z0, const float *__restrict__ s0, const float *__restrict__ t0, int C0, float *__restrict__ a0,
                                   const float *__restrict__ z1, const float *__restrict__ s1, const float *__restrict__ t1, int C1, float *__restrict__ a1,
                                   int rows, int per)
{
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, n0 = (size_t)rows * C0, n1 = (size_t)rows * C1;
    if (i < n0) {
        const int row = (int)(i / C0), c = (int)(i % C0);
        const size_t so = (size_t)(row / per) * C0 + c;
        a0[i] = fmaxf(fmaf(z0[i], s0[so], t0[so]), 0.f);
    } else if (i < n0 + n1) {
        const size_t k = i - n0;
        const int row = (int)(k / C1), c = (int)(k % C1);
        const size_t so = (size_t)(row / per) * C1 + c;
        a1[k] = fmaxf(fmaf(z1[k], s1[so], t1[so]), 0.f);
    }
}

int fc_act_pair(const float *z0, const float *s0, const float *t0, int C0, float *a0, const float *z1, const float *s1, const float *t1, int C1, float *a1,
                int rows, int per, hipStream_t st)
{
    const size_t n = (size_t)rows * (C0 + C1);
    hipLaunchKernelGGL(fc_act_pair_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, z0, s0, t0, C0, a0, z1, s1, t1, C1, a1, rows, per);
    return check_launch("fc_act_pair_kernel");
}

int fc_act(const float *z, const float *s, const float *t, int rows, int C, int per, float *act, hipStream_t st)
{
    const size_t n = (size_t)rows * C;
    hipLaunchKernelGGL(fc_act_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, z, s, t, rows, C, per, act);
    return check_launch("fc_act_kernel");
}

// da [rows, C] = grad wrt relu(bn(z)); block = (slot, 64 channels): full BatchNorm backward over the slot's `per` rows
constexpr int FCB_G = 8;       // row groups per channel (block = 64 channels x 8 groups)

__global__ __launch_bounds__(64 * FCB_G) void fc_bn_bwd_kernel(const float *__restrict__ da, const float *__restrict__ z,
                                                              const float *__restrict__ scale, const float *__restrict__ shift,
                                                              const float *__restrict__ mean, const float *__restrict__ invstd, int per, int C,
                                                              float *__restrict__ g, float *__restrict__ slot_ab)
{
    __shared__ double rA[FCB_G][64], rB[FCB_G][64];
    const int s = blockIdx.x, cl = threadIdx.x & 63, grp = threadIdx.x >> 6, c = blockIdx.y * 64 + cl;
    const bool ok = c < C;
    const size_t so = (size_t)s * C + (ok ? c : 0);
    const float sc = scale[so], sh = shift[so], mu = mean[so], is = invstd[so];
    double A = 0.0, Bs = 0.0;
    // eight rows per trip, their sixteen loads in flight together (a dependent chain of 2 * per loads otherwise)
    for (int i0 = grp; i0 < per; i0 += FCB_G * 8) {
        float zv[8], dv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + FCB_G * u;
            const bool live = ok && i < per;
            const size_t o = (size_t)(s * per + (live ? i : 0)) * C + (ok ? c : 0);
            zv[u] = live ? z[o] : 0.f;
            dv[u] = live ? da[o] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float dy = fmaf(zv[u], sc, sh) > 0.f ? dv[u] : 0.f;      // dead rows: dv = 0
            A += (double)dy;
            Bs += (double)dy * (double)((zv[u] - mu) * is);
        }
    }
    rA[grp][cl] = A;
    rB[grp][cl] = Bs;
    __syncthreads();
    A = 0.0;
    Bs = 0.0;
#pragma unroll
    for (int k = 0; k < FCB_G; ++k) {
        A += rA[k][cl];
        Bs += rB[k][cl];
    }
    const float an = (float)(A / per), bn = (float)(Bs / per);
    for (int i0 = grp; i0 < per; i0 += FCB_G * 8) {
        float zv[8], dv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + FCB_G * u;
            const bool live = ok && i < per;
            const size_t o = (size_t)(s * per + (live ? i : 0)) * C + (ok ? c : 0);
            zv[u] = live ? z[o] : 0.f;
            dv[u] = live ? da[o] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + FCB_G * u;
            if (!(ok && i < per)) continue;
            const float dy = fmaf(zv[u], sc, sh) > 0.f ? dv[u] : 0.f;
            g[(size_t)(s * per + i) * C + c] = sc * (dy - an - (zv[u] - mu) * is * bn);
        }
    }
    if (grp == 0 && ok) {
        slot_ab[so * 2 + 0] = (float)A;
        slot_ab[so * 2 + 1] = (float)Bs;
    }
}

int fc_bn_bwd(const float *da, const float *z, const float *scale, const float *shift, const float *mean, const float *invstd,
              int n_slots, int per, int C, float *g, float *slot_ab, hipStream_t st)
{
    hipLaunchKernelGGL(fc_bn_bwd_kernel, dim3(n_slots, cdiv(C, 64)), dim3(64 * FCB_G), 0, st, da, z, scale, shift, mean, invstd, per, C, g, slot_ab);
    int rc = check_launch("fc_bn_bwd_kernel");
    if (rc == AMPNET_OK && sync_bn_on()) rc = sync_bn_fc_apply(da, z, scale, shift, mean, invstd, slot_ab, n_slots, per, C, g, st);
    return rc;
}

// block = 32 columns x 8 row groups, fixed summation order
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ x, int rows, int C, float *__restrict__ out)
{
    __shared__ float red[8][32];
    const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float s0 = 0.f, s1 = 0.f;
    if (c < C) {
        int r = g;
        for (; r + 24 < rows; r += 32) {
            const float v0 = x[(size_t)r * C + c], v1 = x[(size_t)(r + 8) * C + c], v2 = x[(size_t)(r + 16) * C + c], v3 = x[(size_t)(r + 24) * C + c];
            s0 += v0 + v2;
            s1 += v1 + v3;
        }
        for (; r < rows; r += 8) s0 += x[(size_t)r * C + c];
    }
    red[g][cl] = s0 + s1;
    __syncthreads();
    if (g == 0 && c < C) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += red[k][cl];
        out[c] = s;
    }
}

int colsum(const float *x, int rows, int C, float *out, hipStream_t st)
{
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(C, 32)), dim3(256), 0, st, x, rows, C, out);
    return check_launch("colsum_kernel");
}

// ----------------------------------------------------------------------------------------------------
// input layers (K = 3 / 9): per-window dWeff[q][c][f] = sum_rows g[row][c] * x[row][f], lane = channel c
// ----------------------------------------------------------------------------------------------------
constexpr int IW_WAVES = 8, IW_U = 16;     // waves per window, rows in flight per wave: the kernel is one pass over dy (HBM)
__global__ __launch_bounds__(64 * IW_WAVES) void pw_input_wgrad_kernel(PwInputWgrad a)
{
    __shared__ __attribute__((aligned(16))) float sx[256 * 12];          // 12-float row pitch: three 16-byte reads per row
    __shared__ __attribute__((aligned(16))) float red[IW_WAVES][64][9];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = blockIdx.x;
    const int row_begin = a.win_off[q], row_end = a.win_off[q + 1];
    const int slot = a.n_slots > 1 ? q % a.n_slots : 0;
    float p1, p2, p3;
    if (a.fin_part_a) {
        // wave w sums the slot's partials w, w + IW_WAVES, ... (all loads of a trip in flight), the waves merge through LDS in wave order
        double sa = 0.0, sb = 0.0;
        const int per_slot_parts = (a.fin_parts - slot + a.n_slots - 1) / a.n_slots;
        for (int k0 = wave; k0 < per_slot_parts; k0 += IW_WAVES * 8) {
            float va[8], vb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + IW_WAVES * u;
                const size_t o = (size_t)(slot + (k < per_slot_parts ? k : 0) * a.n_slots) * 64 + lane;
                va[u] = a.fin_part_a[o];
                vb[u] = a.fin_part_b[o];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (k0 + IW_WAVES * u < per_slot_parts) {
                    sa += (double)va[u];
                    sb += (double)vb[u];
                }
            }
        }
        double *fr = reinterpret_cast<double *>(&red[0][0][0]);       // [IW_WAVES][64][2]
        fr[(wave * 64 + lane) * 2 + 0] = sa;
        fr[(wave * 64 + lane) * 2 + 1] = sb;
        __syncthreads();
        double A = 0.0, Bs = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < IW_WAVES; ++w2) {
            A += fr[(w2 * 64 + lane) * 2 + 0];
            Bs += fr[(w2 * 64 + lane) * 2 + 1];
        }
        __syncthreads();                                               // red[] is reused below
        const size_t o = (size_t)slot * 64 + lane;
        const double n = (double)a.fin_rows, invstd = a.fin_invstd[o], mean = a.fin_mean[o];
        const double s = (double)a.fin_gamma[lane] * invstd;
        const double q2 = -s * invstd * Bs / n;
        p1 = (float)s;
        p2 = (float)q2;
        p3 = (float)(-s * A / n - q2 * mean);
        if (q < a.n_slots && wave == 0) {                              // the first window of each slot: the arrays other kernels read
            a.fin_slot_ab[o * 2 + 0] = (float)A;
            a.fin_slot_ab[o * 2 + 1] = (float)Bs;
            a.fin_P1[o] = p1;
            a.fin_P2[o] = p2;
            a.fin_P3[o] = p3;
        }
    } else {
        p1 = a.P1[(size_t)slot * 64 + lane];
        p2 = a.P2[(size_t)slot * 64 + lane];
        p3 = a.P3[(size_t)slot * 64 + lane];
    }
    // effective weights of this lane's channel, exactly as pw_input forms them (pw_misc.hip)
    float w[9];
    if (a.mode == 0) {
#pragma unroll
        for (int f = 0; f < 9; ++f) w[f] = f < 3 ? a.W[lane * 3 + f] : 0.f;
    } else {
        const int pidx = a.perwin_slot_major ? (q % a.n_slots) * (a.Q / a.n_slots) + q / a.n_slots : q;
        const float *T = a.T + (size_t)pidx * 9;
#pragma unroll
        for (int f = 0; f < 9; ++f) {
            float v = a.W[lane * 12 + 3 + f];
            if (f < 3) v += T[f * 3 + 0] * a.W[lane * 12 + 0] + T[f * 3 + 1] * a.W[lane * 12 + 1] + T[f * 3 + 2] * a.W[lane * 12 + 2];
            w[f] = v;
        }
    }
    const bool three = a.mode == 0;
    float acc[9];
#pragma unroll
    for (int f = 0; f < 9; ++f) acc[f] = 0.f;
    for (int base = row_begin; base < row_end; base += 256) {
        const int n = min(256, row_end - base);
        __syncthreads();
        for (int e = tid; e < n * 9; e += 64 * IW_WAVES) sx[(e / 9) * 12 + e % 9] = a.x[(size_t)base * 9 + e];
        __syncthreads();
        // (Round 3 also tried: the dy loads of the next trip issued before the current trip's FMAs -- 90 -> 92 us, the kernel is VALU-bound
        // (~26 instructions per row and wave) with 576 workgroups on 256 CUs, not waiting for loads; and a three-feature branch for the xyz-only
        // T-Net layer -- 90 -> 120 us, the branch inside the unrolled trip costs more than the six FMAs it saves.)
        for (int i0 = wave; i0 < n; i0 += IW_WAVES * IW_U) {        // IW_U rows per trip: all loads first, then the FMAs
            float dyv[IW_U];
#pragma unroll
            for (int u = 0; u < IW_U; ++u) {
                const int i = min(i0 + IW_WAVES * u, n - 1);
                dyv[u] = a.dy[(size_t)(base + i) * 64 + lane];
            }
#pragma unroll
            for (int u = 0; u < IW_U; ++u) {
                const int i = i0 + IW_WAVES * u;
                if (i < n) {
                    const sg_f32x4 x0 = *reinterpret_cast<const sg_f32x4 *>(sx + i * 12), x1 = *reinterpret_cast<const sg_f32x4 *>(sx + i * 12 + 4);
                    const float xr[9] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3], sx[i * 12 + 8]};
                    float z = 0.f;
                    if (three) {
                        z = xr[0] * w[0] + xr[1] * w[1] + xr[2] * w[2];
                    } else {
#pragma unroll
                        for (int f = 0; f < 9; ++f) z = fmaf(xr[f], w[f], z);
                    }
                    const float g = fmaf(dyv[u], p1, fmaf(z, p2, p3));
#pragma unroll
                    for (int f = 0; f < 9; ++f) acc[f] = fmaf(g, xr[f], acc[f]);
                }
            }
        }
    }
#pragma unroll
    for (int f = 0; f < 9; ++f) red[wave][lane][f] = acc[f];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int f = 0; f < 9; ++f)
        {
            float v = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < IW_WAVES; ++w2) v += red[w2][lane][f];
            a.dWeff[((size_t)q * 64 + lane) * 9 + f] = v;
        }
    }
}

int pw_input_wgrad(const PwInputWgrad &a, hipStream_t st)
{
    AMPNET_REQUIRE(a.x && a.dy && a.W && a.dWeff && a.win_off && (a.fin_part_a || (a.P1 && a.P2 && a.P3)), "pw_input_wgrad: null pointer");
    AMPNET_REQUIRE(!a.fin_part_a || (a.fin_part_b && a.fin_parts >= a.n_slots && a.fin_rows >= 1 && a.fin_gamma && a.fin_mean && a.fin_invstd && a.fin_P1 && a.fin_P2 && a.fin_P3 &&
                                     a.fin_slot_ab && a.Q >= a.n_slots),
                   "pw_input_wgrad: in-kernel BatchNorm-backward constants need the producer's partials, the layer's statistics and the output arrays");
    AMPNET_REQUIRE(a.mode == 0 || a.T, "pw_input_wgrad: mode 1 needs T");
    hipLaunchKernelGGL(pw_input_wgrad_kernel, dim3(a.Q), dim3(64 * IW_WAVES), 0, st, a);
    return check_launch("pw_input_wgrad_kernel");
}

// mode 0 (T-Net conv_1 on xyz): dW[c][f] = sum_q dWeff[q][c][f], f < 3
// mode 1 (encoder conv_1):      dW[c][3+f] = sum_q dWeff[q][c][f];  dW[c][d] = sum_q sum_i T[p(q)][i][d] * dWeff[q][c][i];
//                               dT[p(q)][i][d] = sum_c dWeff[q][c][i] * W[c][d]
constexpr int IPG_WB = 16;     // weight-gradient blocks of input_param_grads (4 channels each)
__global__ __launch_bounds__(256) void input_param_grads_kernel(const float *__restrict__ dWeff, const float *__restrict__ W,
                                                               const float *__restrict__ T, int Q, int n_slots, int slot_major, int mode,
                                                               float *__restrict__ dW, float *__restrict__ dT)
{
    if (blockIdx.x < IPG_WB) {
        // blocks 0..15: the weight gradient of 4 channels each, thread = (channel c, window group g of 64); groups summed in fixed order.
        // 64 groups: a thread walks Q / 64 windows, four per trip with their loads (9 gradient values, 9 transform entries each) issued before
        // the first use -- three dependent trips at Q = 576 (nine with the 4 x 16-group blocks before: 21.6 us)
        __shared__ float red[64][4][12];
        const int cl = threadIdx.x & 3, c = blockIdx.x * 4 + cl, g = threadIdx.x >> 2;
        float s[12];
#pragma unroll
        for (int f = 0; f < 12; ++f) s[f] = 0.f;
        for (int q0 = g; q0 < Q; q0 += 64 * 4) {
            float ev[4][9], tv[4][9];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = q0 + 64 * u;
                const bool live = q < Q;
                const float *e = dWeff + ((size_t)(live ? q : 0) * 64 + c) * 9;
                const int p = slot_major ? ((live ? q : 0) % n_slots) * (Q / n_slots) + (live ? q : 0) / n_slots : (live ? q : 0);
#pragma unroll
                for (int f = 0; f < 9; ++f) {
                    ev[u][f] = live ? e[f] : 0.f;
                    tv[u][f] = (live && mode != 0) ? T[p * 9 + f] : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (mode == 0) {
#pragma unroll
                    for (int f = 0; f < 3; ++f) s[f] += ev[u][f];
                } else {
#pragma unroll
                    for (int f = 0; f < 9; ++f) s[3 + f] += ev[u][f];
#pragma unroll
                    for (int d = 0; d < 3; ++d) s[d] += tv[u][0 * 3 + d] * ev[u][0] + tv[u][1 * 3 + d] * ev[u][1] + tv[u][2 * 3 + d] * ev[u][2];
                }
            }
        }
#pragma unroll
        for (int f = 0; f < 12; ++f) red[g][cl][f] = s[f];
        __syncthreads();
        if (g == 0) {
            const int nf = mode == 0 ? 3 : 12;
            for (int f = 0; f < nf; ++f) {
                float v = 0.f;
#pragma unroll 16
                for (int k = 0; k < 64; ++k) v += red[k][cl][f];
                dW[c * nf + f] = v;
            }
        }
    } else if (mode == 1) {
        // blocks IPG_WB..: dT for windows, one (q, i, d) per thread
        const int idx = (blockIdx.x - IPG_WB) * 256 + threadIdx.x;
        if (idx >= Q * 9) return;
        const int q = idx / 9, i = (idx % 9) / 3, d = idx % 3;
        const int p = slot_major ? (q % n_slots) * (Q / n_slots) + q / n_slots : q;
        float s = 0.f;
#pragma unroll 8
        for (int cc = 0; cc < 64; ++cc) s = fmaf(dWeff[((size_t)q * 64 + cc) * 9 + i], W[cc * 12 + d], s);
        dT[p * 9 + i * 3 + d] = s;
    }
}

int input_param_grads(const float *dWeff, const float *W, const float *T, int Q, int n_slots, int slot_major, int mode, float *dW,
                      float *dT, hipStream_t st)
{
    const int blocks = IPG_WB + (mode == 1 ? cdiv(Q * 9, 256) : 0);
    hipLaunchKernelGGL(input_param_grads_kernel, dim3(blocks), dim3(256), 0, st, dWeff, W, T, Q, n_slots, slot_major, mode, dW, dT);
    return check_launch("input_param_grads_kernel");
}

__global__ void axpy_kernel(const float *__restrict__ x, float alpha, size_t n, float *__restrict__ y)
{
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) y[i] = fmaf(alpha, x[i], y[i]);
}

__global__ void fill_kernel(float *__restrict__ p, size_t n, float v)
{
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

int fill_f32(float *p, size_t n, float v, hipStream_t st)
{
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, n, v);
    return check_launch("fill_kernel");
}

__global__ void fill_pair_kernel(float *__restrict__ p0, float v0, float *__restrict__ p1, float v1, size_t n)
{
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) {
        p0[i] = v0;
        p1[i] = v1;
    }
}

// two constant arrays of n floats in one launch
int fill_f32_pair(float *p0, float v0, float *p1, float v1, size_t n, hipStream_t st)
{
    hipLaunchKernelGGL(fill_pair_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p0, v0, p1, v1, n);
    return check_launch("fill_pair_kernel");
}

int axpy(const float *x, float alpha, size_t n, float *y, hipStream_t st)
{
    hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, alpha, n, y);
    return check_launch("axpy_kernel");
}

// masked pooled gradient for FC inputs: nothing to do (pool_bwd handles the ReLU of the pooled layer)

}  // namespace ampnet

namespace ampnet {
// dst[p(q)] = src[q]^T with p(q) the slot-major row of window q
__global__ __launch_bounds__(256) void transpose64_kernel(const float *__restrict__ src, float *__restrict__ dst, int Q, int n_slots, int chunks, int by_workgroup,
                                                          const float *__restrict__ add)
{
    __shared__ float t[64][65];
    const int q = blockIdx.x;
    const size_t p = (size_t)(q % n_slots) * (Q / n_slots) + q / n_slots;
    // the window's partials, summed in chunk order; four elements x up to eight chunks of loads in flight per thread (the plain nest waited for
    // every load in turn: 42 us for 75 MB)
    // by_workgroup: partials indexed like pw_bwd_fused's workgroups, ((q / n_slots) * chunks + ch) * n_slots + q % n_slots
    for (int e0 = threadIdx.x; e0 < 4096; e0 += 256 * 4) {
        float sacc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c0 = 0; c0 < chunks; c0 += 8) {
            float v[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int ch = c0 + i < chunks ? c0 + i : c0;
                    const size_t pi = by_workgroup ? ((size_t)(q / n_slots) * chunks + ch) * n_slots + q % n_slots : (size_t)q * chunks + ch;
                    v[u][i] = src[pi * 4096 + e0 + 256 * u];
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (c0 + i < chunks) sacc[u] += v[u][i];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) t[(e0 + 256 * u) / 64][(e0 + 256 * u) % 64] = sacc[u];
    }
    __syncthreads();
    // + add (same layout as dst): the gradient that reaches the transforms from outside (the regulariser), no separate axpy launch
    for (int e = threadIdx.x; e < 4096; e += 256) dst[p * 4096 + e] = t[e % 64][e / 64] + (add ? add[p * 4096 + e] : 0.f);
}

int transpose64_slot_major(const float *src, float *dst, int Q, int n_slots, int chunks, int by_workgroup, const float *add, hipStream_t st)
{
    hipLaunchKernelGGL(transpose64_kernel, dim3(Q), dim3(256), 0, st, src, dst, Q, n_slots, chunks, by_workgroup, add);
    return check_launch("transpose64_kernel");
}
}  // namespace ampnet

namespace ampnet {

// ----------------------------------------------------------------------------------------------------
// sparse_rows: one workgroup per window.  Channels that share an argmax row are merged (in channel order, so the sums
// are reproducible); the merged rows (P1 dy) W go to srows[q * C + i], their row numbers to srow_row.
// ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sparse_rows_kernel(SparseRows a)
{
    __shared__ int sArg[256], sIdx[256], sDup[256];
    __shared__ float sCoef[256];
    __shared__ int s_wave_cnt[4], s_dup_cnt[4], s_unique, s_ndup;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int slot = a.n_slots > 1 ? q % a.n_slots : 0;
    const int prow = a.slot_major ? (q % a.n_slots) * (a.Q / a.n_slots) + q / a.n_slots : q;
    const int c = tid;                                        // C <= 256 = blockDim
    int r = -1;
    if (c < a.C) r = a.arg[(size_t)q * a.C + c];
    sArg[c] = r;
    sCoef[c] = r >= 0 ? a.P1[(size_t)slot * a.C + c] * a.dpm[(size_t)prow * a.C + c] : 0.f;
    __syncthreads();
    // first[c] = lowest channel with the same argmax row; owners (first == c) get compact indices in channel order
    int first = -1;
    if (r >= 0) {
        first = c;
        for (int p = 0; p < c; ++p)
            if (sArg[p] == r) {
                first = p;
                break;
            }
    }
    const bool owner = r >= 0 && first == c, dup = r >= 0 && first != c;
    const unsigned long long bo = __ballot(owner), bd = __ballot(dup);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (lane == 0) {
        s_wave_cnt[wv] = __popcll(bo);
        s_dup_cnt[wv] = __popcll(bd);
    }
    __syncthreads();
    int base = 0, dbase = 0;
    for (int w = 0; w < wv; ++w) {
        base += s_wave_cnt[w];
        dbase += s_dup_cnt[w];
    }
    if (tid == 0) {
        s_unique = s_wave_cnt[0] + s_wave_cnt[1] + s_wave_cnt[2] + s_wave_cnt[3];
        s_ndup = s_dup_cnt[0] + s_dup_cnt[1] + s_dup_cnt[2] + s_dup_cnt[3];
    }
    if (owner) sIdx[c] = base + __popcll(bo & below);
    if (dup) sDup[dbase + __popcll(bd & below)] = c;         // the merged channels, in channel order
    __syncthreads();
    if (dup) sIdx[c] = sIdx[first];
    if (owner) a.srow_row[(size_t)q * a.C + sIdx[c]] = r;
    if (tid == 0) a.srow_cnt[q] = s_unique;
    __syncthreads();
    // every owner's row = coef * W[c][:] (one wave per channel, lanes over k: coalesced) ...
    float *out = a.srows + (size_t)q * a.C * a.cp;
    __shared__ unsigned char sIsDup[256];
    sIsDup[c] = dup ? 1 : 0;
    __syncthreads();
    {
        // thread = (channel group, 16-byte column): cp / 4 columns, 256 / (cp / 4) channels per trip, eight trips in flight
        const int qn = a.cp / 4, cstep = 256 / qn, kq = tid % qn, cg = tid / qn;
        for (int c0 = cg; c0 < a.C; c0 += cstep * 8) {
            f32x4 wv4[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int cc = c0 + cstep * u;
                const bool own = cc < a.C && sArg[cc] >= 0 && !sIsDup[cc];
                wv4[u] = own ? *reinterpret_cast<const f32x4 *>(a.W + (size_t)cc * a.cp + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int cc = c0 + cstep * u;
                if (cc < a.C && sArg[cc] >= 0 && !sIsDup[cc]) *reinterpret_cast<f32x4 *>(out + (size_t)sIdx[cc] * a.cp + 4 * kq) = wv4[u] * sCoef[cc];
            }
        }
    }
    __syncthreads();
    // ... then the few merged channels are added in channel order (same thread per column: reproducible)
    if (tid < a.cp) {
        const int k = tid;
        for (int d = 0; d < s_ndup; ++d) {
            const int cc = sDup[d];
            out[(size_t)sIdx[cc] * a.cp + k] = fmaf(sCoef[cc], a.W[(size_t)cc * a.cp + k], out[(size_t)sIdx[cc] * a.cp + k]);
        }
    }
}

int sparse_rows(const SparseRows &a, hipStream_t st)
{
    AMPNET_REQUIRE(a.arg && a.dpm && a.P1 && a.W && a.srows && a.srow_row && a.srow_cnt, "sparse_rows: null pointer");
    AMPNET_REQUIRE(a.cp <= 256 && a.C <= 256 && a.cp % 4 == 0 && 256 % (a.cp / 4) == 0, "sparse_rows: C=%d cp=%d", a.C, a.cp);
    hipLaunchKernelGGL(sparse_rows_kernel, dim3(a.Q), dim3(256), 0, st, a);
    return check_launch("sparse_rows_kernel");
}

// one workgroup per window, thread = column; walks the window's merged rows in order (reproducible sums)
__global__ __launch_bounds__(128) void sparse_fix_kernel(SparseFix a)
{
    const int q = blockIdx.x, k = threadIdx.x;
    if (k >= a.cp) return;
    const int slot = a.n_slots > 1 ? q % a.n_slots : 0;
    const float sc = a.s_prev[(size_t)slot * a.cp + k], sh = a.t_prev[(size_t)slot * a.cp + k];
    const float mu = a.mean_prev[(size_t)slot * a.cp + k], is = a.invstd_prev[(size_t)slot * a.cp + k];
    const int n = a.srow_cnt[q];
    float sa = 0.f, sb = 0.f;
    // eight rows per trip: their loads (row index, z, the scattered row, the current output) are all in flight before
    // the first store -- the rows of one window are distinct, so the read-modify-writes do not alias
    for (int i0 = 0; i0 < n; i0 += 8) {
        int rowv[8];
        float zv[8], sv[8], ov[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) rowv[u] = a.srow_row[(size_t)q * a.C + min(i0 + u, n - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            zv[u] = a.z_prev[(size_t)rowv[u] * a.cp + k];
            sv[u] = a.srows[((size_t)q * a.C + min(i0 + u, n - 1)) * a.cp + k];
            ov[u] = a.out[(size_t)rowv[u] * a.cp + k];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (i0 + u >= n) continue;
            const float v = fmaf(zv[u], sc, sh) > 0.f ? sv[u] : 0.f;
            a.out[(size_t)rowv[u] * a.cp + k] = ov[u] + v;
            sa += v;
            sb = fmaf(v, (zv[u] - mu) * is, sb);
        }
    }
    const size_t o = (size_t)(q * a.part_chunks + a.slot_idx) * a.cp + k;
    a.part_a[o] = sa;
    a.part_b[o] = sb;
}

int sparse_fix(const SparseFix &a, hipStream_t st)
{
    AMPNET_REQUIRE(a.srows && a.srow_row && a.srow_cnt && a.z_prev && a.s_prev && a.t_prev && a.mean_prev && a.invstd_prev && a.out && a.part_a && a.part_b, "sparse_fix: null pointer");
    AMPNET_REQUIRE(a.cp <= 128, "sparse_fix: cp=%d", a.cp);
    hipLaunchKernelGGL(sparse_fix_kernel, dim3(a.Q), dim3(128), 0, st, a);
    return check_launch("sparse_fix_kernel");
}

// ----------------------------------------------------------------------------------------------------
// sparse_scatter = sparse_rows + sparse_fix without the [Q * C, cp] intermediate: one workgroup per window finds, for every
// distinct argmax row, the chain of channels that picked it (channel order, so the sums are reproducible), and adds
//   out[row][k] += mask(row, k) * sum_chain P1[c] dpm[c] W[c][k]
// together with that window's share of the BatchNorm-backward sums.  Eight rows per trip, all loads before the first use.
// ----------------------------------------------------------------------------------------------------
template <bool ZB> __global__ __launch_bounds__(512) void sparse_scatter_kernel(SparseScatter a)
{
    __shared__ int sArg[256], sOwn[256], sNext[256];
    __shared__ float sCoef[256];
    __shared__ int s_wave_cnt[8];
    __shared__ float sRed[4][2][128];
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int slot = a.n_slots > 1 ? q % a.n_slots : 0;
    const int prow = a.slot_major ? (q % a.n_slots) * (a.Q / a.n_slots) + q / a.n_slots : q;
    const int c = tid;                                        // C <= 256; threads 256..511 only join the row loop
    int r = -1;
    if (c < a.C) r = a.arg[(size_t)q * a.C + c];
    if (c < 256) {
        sArg[c] = r;
        sCoef[c] = r >= 0 ? a.P1[(size_t)slot * a.C + c] * a.dpm[(size_t)prow * a.C + c] : 0.f;
        sNext[c] = -1;
    }
    __syncthreads();
    // pred = the closest lower channel with the same argmax row (none: this channel owns the row)
    int pred = -1;
    if (r >= 0 && c < 256) {
        for (int p = c - 1; p >= 0; --p)
            if (sArg[p] == r) {
                pred = p;
                break;
            }
    }
    const bool owner = r >= 0 && pred < 0;
    if (pred >= 0) sNext[pred] = c;                           // every channel has at most one successor
    const unsigned long long bo = __ballot(owner);
    if (lane == 0) s_wave_cnt[wv] = __popcll(bo);
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wv; ++w) base += s_wave_cnt[w];
    const int n_own = s_wave_cnt[0] + s_wave_cnt[1] + s_wave_cnt[2] + s_wave_cnt[3];      // waves 4..7 own nothing
    if (owner) sOwn[base + __popcll(bo & ((1ull << lane) - 1ull))] = c;      // owners in channel order
    __syncthreads();

    const int k = tid & 127, grp = tid >> 7;                 // four row groups x cp columns
    float sa = 0.f, sb = 0.f;
    if (k < a.cp) {
        const float sc = a.s_prev[(size_t)slot * a.cp + k], sh = a.t_prev[(size_t)slot * a.cp + k];
        const float mu = a.mean_prev[(size_t)slot * a.cp + k], is = a.invstd_prev[(size_t)slot * a.cp + k];
        for (int i0 = grp; i0 < n_own; i0 += 32) {
            int ch[8], rowv[8];
            float wv8[8], zv[8], ov[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = min(i0 + 4 * u, n_own - 1);
                ch[u] = sOwn[i];
                rowv[u] = sArg[ch[u]];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                wv8[u] = a.W[(size_t)ch[u] * a.cp + k];
                zv[u] = ld_act_t<ZB>(a.z_prev, (size_t)rowv[u] * a.cp + k);
                ov[u] = a.out[(size_t)rowv[u] * a.cp + k];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (i0 + 4 * u >= n_own) continue;
                float sv = sCoef[ch[u]] * wv8[u];
                for (int cc = sNext[ch[u]]; cc >= 0; cc = sNext[cc]) sv = fmaf(sCoef[cc], a.W[(size_t)cc * a.cp + k], sv);   // merged channels (rare)
                const float v = fmaf(zv[u], sc, sh) > 0.f ? sv : 0.f;
                a.out[(size_t)rowv[u] * a.cp + k] = ov[u] + v;            // distinct rows: no conflict
                sa += v;
                sb = fmaf(v, (zv[u] - mu) * is, sb);
            }
        }
    }
    sRed[grp][0][k] = sa;
    sRed[grp][1][k] = sb;
    __syncthreads();
    if (grp == 0 && k < a.cp) {
        const size_t o = (size_t)(q * a.part_chunks + a.slot_idx) * a.cp + k;
        a.part_a[o] = (sRed[0][0][k] + sRed[1][0][k]) + (sRed[2][0][k] + sRed[3][0][k]);
        a.part_b[o] = (sRed[0][1][k] + sRed[1][1][k]) + (sRed[2][1][k] + sRed[3][1][k]);
    }
}

int sparse_scatter(const SparseScatter &a, hipStream_t st)
{
    AMPNET_REQUIRE(a.arg && a.dpm && a.P1 && a.W && a.z_prev && a.s_prev && a.t_prev && a.mean_prev && a.invstd_prev && a.out && a.part_a && a.part_b,
                   "sparse_scatter: null pointer");
    AMPNET_REQUIRE(a.cp <= 128 && a.C <= 256, "sparse_scatter: C=%d cp=%d", a.C, a.cp);
    if (a.z_bf16) hipLaunchKernelGGL(sparse_scatter_kernel<true>, dim3(a.Q), dim3(512), 0, st, a);
    else hipLaunchKernelGGL(sparse_scatter_kernel<false>, dim3(a.Q), dim3(512), 0, st, a);
    return check_launch("sparse_scatter_kernel");
}

// workgroup = (16 rows j of G[slot] | the c0 row, slot): W streams through LDS once per workgroup in chunks of 32 channels
// (the next chunk's loads are in flight while the current one is multiplied), thread = column k x 8 of the 16 rows
constexpr int SM_J = 16, SM_C = 32;

__global__ __launch_bounds__(256) void slot_mats_kernel(const float *__restrict__ W, const float *__restrict__ P2, const float *__restrict__ P3,
                                                       int C, int cp, float *__restrict__ G, float *__restrict__ c0)
{
    __shared__ float sWk[SM_C][128], sWj[SM_C][SM_J];
    const int s = blockIdx.y, jb = blockIdx.x, tid = threadIdx.x;
    const bool c0_block = jb * SM_J >= cp;
    const int j0 = jb * SM_J;
    const int k = tid & 127, jg = tid >> 7;                     // rows j0 + 8 jg .. + 7
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    // staging roles: 16 floats of the chunk per thread; chunk element e = c * cp + kk
    float rw[16];
    auto fetch = [&](int cbase) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int e = tid + 256 * i, cc = e / 128, kk = e % 128;
            rw[i] = (kk < cp && cbase + cc < C) ? W[(size_t)(cbase + cc) * cp + kk] : 0.f;
        }
    };
    fetch(0);
    for (int cbase = 0; cbase < C; cbase += SM_C) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int e = tid + 256 * i;
            sWk[e / 128][e % 128] = rw[i];
        }
        __syncthreads();
        for (int e = tid; e < SM_C * SM_J; e += 256) {
            const int cc = e / SM_J, jj = e % SM_J, c = cbase + cc;
            float v = 0.f;
            if (c < C) {
                if (c0_block) v = jj == 0 ? P3[(size_t)s * C + c] : 0.f;
                else v = (j0 + jj < cp) ? sWk[cc][j0 + jj] * P2[(size_t)s * C + c] : 0.f;
            }
            sWj[cc][jj] = v;
        }
        if (cbase + SM_C < C) fetch(cbase + SM_C);
        __syncthreads();
#pragma unroll 8
        for (int cc = 0; cc < SM_C; ++cc) {
            const float wk = sWk[cc][k];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = fmaf(sWj[cc][8 * jg + i], wk, acc[i]);
        }
    }
    if (k >= cp) return;
    if (c0_block) {
        if (jg == 0) c0[(size_t)s * cp + k] = acc[0];
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (j0 + 8 * jg + i < cp) G[((size_t)s * cp + j0 + 8 * jg + i) * cp + k] = acc[i];
    }
}

int slot_mats(const float *W, const float *P2, const float *P3, int n_slots, int C, int cp, float *G, float *c0, hipStream_t st)
{
    AMPNET_REQUIRE(W && P2 && P3 && G && c0 && cp >= 1 && cp <= 128 && C >= 1, "slot_mats: bad arguments (cp=%d)", cp);
    if (n_slots + 1 <= SG_MAX_PROBLEMS) {
        // G[s] = W^T diag(P2[s]) W and c0 = P3 W as ONE launch of the matrix-core small GEMM: n_slots problems [cp, cp, K = C] whose A operand is
        // scaled along k as it is loaded, + one [n_slots, cp, K = C] problem (the VALU kernel below made eight load -> barrier -> multiply trips
        // on 81 workgroups: 24.6 us per pooled layer)
        SgProblem p[SG_MAX_PROBLEMS];
        for (int s2 = 0; s2 < n_slots; ++s2) {
            p[s2] = {cp, cp, C, 1, 0, cp, cp, cp, 0, W, W, G + (size_t)s2 * cp * cp};
            p[s2].kscale = P2 + (size_t)s2 * C;
        }
        p[n_slots] = {n_slots, cp, C, 0, 0, C, cp, cp, 0, P3, W, c0};
        return sgemm_launch(p, n_slots + 1, st);
    }
    hipLaunchKernelGGL(slot_mats_kernel, dim3(cdiv(cp, SM_J) + 1, n_slots), dim3(256), 0, st, W, P2, P3, C, cp, G, c0);
    return check_launch("slot_mats_kernel");
}

// blockIdx.z picks one of two (partials, output) pairs: the Gram matrix and the column sums of a pooled layer share a launch
__global__ __launch_bounds__(256) void reduce_slots_kernel(const float *__restrict__ part0, int n_el0, float *__restrict__ out0,
                                                          const float *__restrict__ part1, int n_el1, float *__restrict__ out1, int Q, int chunks,
                                                          int n_slots)
{
    const float *__restrict__ part = blockIdx.z ? part1 : part0;
    float *__restrict__ out = blockIdx.z ? out1 : out0;
    const int n_el = blockIdx.z ? n_el1 : n_el0;
    const int s = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n_el) return;
    float a0 = 0.f, a1 = 0.f;
    for (int q = s; q < Q; q += n_slots)
        for (int ch = 0; ch < chunks; ch += 2) {
            a0 += part[(size_t)(q * chunks + ch) * n_el + e];
            if (ch + 1 < chunks) a1 += part[(size_t)(q * chunks + ch + 1) * n_el + e];
        }
    out[(size_t)s * n_el + e] = a0 + a1;
}

int reduce_slots(const float *part, int Q, int chunks, int n_slots, int n_el, float *out, hipStream_t st)
{
    hipLaunchKernelGGL(reduce_slots_kernel, dim3(cdiv(n_el, 256), n_slots, 1), dim3(256), 0, st, part, n_el, out, part, n_el, out, Q, chunks, n_slots);
    return check_launch("reduce_slots_kernel");
}

int reduce_slots2(const float *part0, int n_el0, float *out0, const float *part1, int n_el1, float *out1, int Q, int chunks, int n_slots, hipStream_t st)
{
    const int m = n_el0 > n_el1 ? n_el0 : n_el1;
    hipLaunchKernelGGL(reduce_slots_kernel, dim3(cdiv(m, 256), n_slots, 2), dim3(256), 0, st, part0, n_el0, out0, part1, n_el1, out1, Q, chunks, n_slots);
    return check_launch("reduce_slots_kernel");
}

// block = output channel c, thread = input channel k
constexpr int PWG_G = 8;       // thread groups per output channel: with sixteen gathers in flight per thread a channel's 576 rows are five trips

template <bool ZB> __global__ __launch_bounds__(128 * PWG_G) void pooled_wgrad_kernel(PooledWgrad a)
{
    __shared__ float sW[256];
    __shared__ float sRed[PWG_G][128];
    extern __shared__ int sDyn[];         // [Q] argmax row (clamped to 0), [Q] coefficient, [S][cp] scale, [S][cp] shift
    int *sRow = sDyn;
    float *sCoef = reinterpret_cast<float *>(sDyn + a.Q);
    float *sS = sCoef + a.Q, *sT = sS + a.n_slots * a.cp;
    const int c = blockIdx.x, tid = threadIdx.x, k = tid & 127, grp = tid >> 7;
    for (int j = tid; j < a.cp; j += 128 * PWG_G) sW[j] = a.W[(size_t)c * a.cp + j];
    for (int q = tid; q < a.Q; q += 128 * PWG_G) {
        const int r = a.arg[(size_t)q * a.C + c];
        const int slot = a.n_slots > 1 ? q % a.n_slots : 0;
        const int prow = a.slot_major ? (q % a.n_slots) * (a.Q / a.n_slots) + q / a.n_slots : q;
        sRow[q] = r < 0 ? 0 : r;
        sCoef[q] = r < 0 ? 0.f : a.P1[(size_t)slot * a.C + c] * a.dpm[(size_t)prow * a.C + c];
    }
    for (int e = tid; e < a.n_slots * a.cp; e += 128 * PWG_G) {
        sS[e] = a.s_prev[e];
        sT[e] = a.t_prev[e];
    }
    __syncthreads();
    float acc = 0.f;
    if (k < a.cp) {
        if (a.wgram) {
            // dense part from W Gram[s] (a.wgram [S, C, cp], one launch of the matrix-core small GEMM before this kernel): two loads per slot,
            // all independent -- the walk over Gram rows below is sixteen dependent trips of eight L2 loads per slot (33 of this kernel's 46 us)
            for (int s = grp; s < a.n_slots; s += PWG_G) {
                acc = fmaf(a.P2[(size_t)s * a.C + c], a.wgram[((size_t)s * a.C + c) * a.cp + k], acc);
                acc = fmaf(a.P3[(size_t)s * a.C + c], a.asum[(size_t)s * a.cp + k], acc);
            }
        } else {
            // dense part: slots s = grp, grp + 4, ...
            for (int s = grp; s < a.n_slots; s += PWG_G) {
                float m = 0.f;
                const float *g = a.gram + (size_t)s * a.cp * a.cp;
#pragma unroll 8
                for (int j = 0; j < a.cp; ++j) m = fmaf(sW[j], g[(size_t)j * a.cp + k], m);
                acc = fmaf(a.P2[(size_t)s * a.C + c], m, acc);
                acc = fmaf(a.P3[(size_t)s * a.C + c], a.asum[(size_t)s * a.cp + k], acc);
            }
        }
        // sparse part: gathers of the argmax rows, windows q = grp, grp + 4, ...; independent loads so that they pipeline
        const float *__restrict__ zp = a.z_prev;
        for (int q0 = grp; q0 < a.Q; q0 += PWG_G * 16) {
            float zv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int q = q0 + PWG_G * u;
                zv[u] = ld_act_t<ZB>(zp, (size_t)sRow[q < a.Q ? q : q0] * a.cp + k);
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int q = q0 + PWG_G * u;
                if (q < a.Q) {
                    const int slot = a.n_slots > 1 ? q % a.n_slots : 0;
                    acc = fmaf(sCoef[q], fmaxf(fmaf(zv[u], sS[slot * a.cp + k], sT[slot * a.cp + k]), 0.f), acc);
                }
            }
        }
    }
    sRed[grp][k] = acc;
    __syncthreads();
    if (grp == 0 && k < a.cp) {
        float v = 0.f;
#pragma unroll
        for (int g2 = 0; g2 < PWG_G; ++g2) v += sRed[g2][k];
        a.dW[(size_t)c * a.cp + k] = v;
    }
}

int pooled_wgrad(const PooledWgrad &a, hipStream_t st)
{
    AMPNET_REQUIRE(a.W && a.P1 && a.P2 && a.P3 && a.gram && a.asum && a.arg && a.dpm && a.z_prev && a.s_prev && a.t_prev && a.dW, "pooled_wgrad: null pointer");
    const size_t lds = ((size_t)a.Q * 2 + (size_t)a.n_slots * a.cp * 2) * sizeof(int);
    AMPNET_REQUIRE(a.cp <= 128 && lds <= 60 * 1024, "pooled_wgrad: cp=%d Q=%d n_slots=%d", a.cp, a.Q, a.n_slots);
    if (a.wgram) {
        AMPNET_REQUIRE(a.n_slots <= SG_MAX_PROBLEMS, "pooled_wgrad: %d slots in one GEMM launch", a.n_slots);
        SgProblem p[SG_MAX_PROBLEMS];
        for (int s2 = 0; s2 < a.n_slots; ++s2)
            p[s2] = {a.C, a.cp, a.cp, 0, 0, a.cp, a.cp, a.cp, 0, a.W, a.gram + (size_t)s2 * a.cp * a.cp, a.wgram + (size_t)s2 * a.C * a.cp};
        int rc = sgemm_launch(p, a.n_slots, st);
        if (rc != AMPNET_OK) return rc;
    }
    if (a.z_bf16) hipLaunchKernelGGL(pooled_wgrad_kernel<true>, dim3(a.C), dim3(128 * PWG_G), lds, st, a);
    else hipLaunchKernelGGL(pooled_wgrad_kernel<false>, dim3(a.C), dim3(128 * PWG_G), lds, st, a);
    return check_launch("pooled_wgrad_kernel");
}

}  // namespace ampnet

// ---- the token-level products of the backward behind the C ABI (tests/test_small_gemm_gpu.py) ----------------------------------------
extern "C" int ampnet_small_gemm_f32(int trans_a, int trans_b, int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc,
                                     int accumulate, float *row_sums, const float *k_scale, void *stream)
{
    AMPNET_REQUIRE(A && B && C, "ampnet_small_gemm_f32: null pointer");
    AMPNET_REQUIRE(M >= 1 && N >= 1 && K >= 1 && lda >= 1 && ldb >= 1 && ldc >= N, "ampnet_small_gemm_f32: M=%d N=%d K=%d lda=%d ldb=%d ldc=%d", M, N, K, lda, ldb, ldc);
    ampnet::SgProblem p = {M, N, K, trans_a, trans_b, lda, ldb, ldc, accumulate, A, B, C};
    p.db = row_sums;
    p.kscale = k_scale;
    return ampnet::sgemm_launch(&p, 1, (hipStream_t)stream);
}

