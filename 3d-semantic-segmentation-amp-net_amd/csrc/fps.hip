// fps.hip -- greedy farthest-point sampling, one workgroup per cloud (C ABI: ampnet_fps_f32).
//
// Replaces utils/utils.py:889-933 of the reference.  The algorithm is S-1 dependent rounds; a round is
//   d[j] = min(d[j], |p_last - p_j|^2)  for every point, then argmax_j d[j] (first index on ties).
// Mapping to CDNA4: the cloud lives in REGISTERS for the whole kernel (thread t owns points t, t+T, ...:
// coalesced 4-byte loads once, 12 B xyz + 4 B running minimum per point = the 16 B per (candidate, round)
// the roofline counts, served from the register file instead of HBM), a round costs
//   VALU update (branch-free)  ->  64-lane argmax by DPP row operations + 4 readlanes  ->  one LDS slot per wave
//   ->  ONE barrier  ->  16-lane DPP fold of the slots
// and the winner's coordinates travel with its slot, so no thread ever indexes its register array at run
// time and nothing is re-read from global memory.  Slots are double-buffered so a round needs one barrier.
//
// Bit parity: distances are float32 ((dx*dx + dy*dy) + dz*dz) with one rounding per operation -- this file
// is compiled with -ffp-contract=off and the pragma below repeats it; picked points carry -1 so that
// min(d, -1) keeps them out for good (every true distance is >= 0); ties go to the lowest index.
#include "common.h"

#pragma clang fp contract(off)

namespace ampnet {

// (distance, index) as one 64-bit key whose unsigned order is "larger distance, then LOWER index":
//   hi = order-preserving image of the float (picked = -1 and padding = -2 sort below every real distance >= 0),
//   lo = ~index.
__device__ __forceinline__ uint32_t fkey(float d)
{
    const int b = __float_as_int(d);
    return (uint32_t)(b ^ ((b >> 31) | (int)0x80000000));
}

template <int CTRL>
__device__ __forceinline__ void dpp_max64(uint32_t &hi, uint32_t &lo)
{
    const uint32_t ohi = (uint32_t)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, 0xF, 0xF, false);
    const uint32_t olo = (uint32_t)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, 0xF, 0xF, false);
    const bool take = (((unsigned long long)ohi << 32) | olo) > (((unsigned long long)hi << 32) | lo);      // one v_cmp_gt_u64
    hi = take ? ohi : hi;
    lo = take ? olo : lo;
}

// max over each row of 16 lanes, result in all 16 lanes: quad swaps, then half-row and row mirrors (DPP, no LDS)
__device__ __forceinline__ void row16_max64(uint32_t &hi, uint32_t &lo)
{
    dpp_max64<0xB1>(hi, lo);     // quad_perm [1,0,3,2]
    dpp_max64<0x4E>(hi, lo);     // quad_perm [2,3,0,1]
    dpp_max64<0x141>(hi, lo);    // row_half_mirror
    dpp_max64<0x140>(hi, lo);    // row_mirror
}

__device__ __forceinline__ void smax64(uint32_t &hi, uint32_t &lo, uint32_t ohi, uint32_t olo)
{
    const bool take = (((unsigned long long)ohi << 32) | olo) > (((unsigned long long)hi << 32) | lo);
    hi = take ? ohi : hi;
    lo = take ? olo : lo;
}

// max over groups of 8 / 4 consecutive lanes (the slot fold of workgroups with <= 8 / <= 4 waves: fewer dependent steps)
__device__ __forceinline__ void row8_max64(uint32_t &hi, uint32_t &lo)
{
    dpp_max64<0xB1>(hi, lo);
    dpp_max64<0x4E>(hi, lo);
    dpp_max64<0x141>(hi, lo);    // row_half_mirror: lane l <-> 7 - l inside each half row
}
__device__ __forceinline__ void row4_max64(uint32_t &hi, uint32_t &lo)
{
    dpp_max64<0xB1>(hi, lo);
    dpp_max64<0x4E>(hi, lo);
}

// LXYZ: a copy of the cloud's coordinates lives in LDS (n * 12 bytes <= 144 KB), so a round only carries (distance,
// index) through the reductions and reads the winner's coordinates with three broadcast LDS loads; otherwise the
// coordinates travel with the per-thread / per-wave winner.
template <int T, int P, bool LXYZ>
__global__ __launch_bounds__(T) void fps_kernel(const float *__restrict__ xyz, int n, int ld, int s,
                                                int32_t *__restrict__ idx)
{
    constexpr int NW = T / WAVE;
    static_assert(NW <= 16, "one row of 16 lanes folds the wave slots");
    extern __shared__ __attribute__((aligned(16))) float s_cloud[];      // LXYZ: x[n], y[n], z[n]
    // per-wave slot = winner key (+ its coordinates), two generations (one barrier per round)
    __shared__ uint32_t s_hi[2][16], s_lo[2][16];
    __shared__ float s_x[2][16], s_y[2][16], s_z[2][16];

    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
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
        if (LXYZ && ok) {
            s_cloud[j] = px[k];
            s_cloud[n + j] = py[k];
            s_cloud[2 * n + j] = pz[k];
        }
    }
    // seed: point 0 (utils.py:907-908)
    float lx = cloud[0], ly = cloud[1], lz = cloud[2];
    if (tid == 0) {
        out[0] = 0;
        dist[0] = -1.0f;
    }
    if (tid < 32) {                                  // unused slots of the 16-lane fold never win
        s_hi[tid >> 4][tid & 15] = 0u;
        s_lo[tid >> 4][tid & 15] = 0u;
    }
    __syncthreads();

    for (int r = 1; r < s; ++r) {
        // ---- update + thread-local argmax, branch-free: ascending k and a strict compare keep the lowest index ----
        float bd = -3.0f, bx = 0.f, by = 0.f, bz = 0.f;
        int bi = 0;
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const float dx = lx - px[k];
            const float dy = ly - py[k];
            const float dz = lz - pz[k];
            const float d = (dx * dx + dy * dy) + dz * dz;
            const float m = fminf(d, dist[k]);          // picked (-1) and padding (-2) stay as they are
            dist[k] = m;
            const bool take = m > bd;
            bd = take ? m : bd;
            bi = take ? k : bi;
            if (!LXYZ) {
                bx = take ? px[k] : bx;
                by = take ? py[k] : by;
                bz = take ? pz[k] : bz;
            }
        }
        const uint32_t bhi = fkey(bd), blo = ~(uint32_t)(tid + bi * T);
        // ---- 64-lane argmax: DPP inside the four rows of 16, then four scalars ----
        uint32_t whi = bhi, wlo = blo;
        row16_max64(whi, wlo);
        uint32_t ghi = (uint32_t)__builtin_amdgcn_readlane((int)whi, 0), glo = (uint32_t)__builtin_amdgcn_readlane((int)wlo, 0);
        smax64(ghi, glo, (uint32_t)__builtin_amdgcn_readlane((int)whi, 16), (uint32_t)__builtin_amdgcn_readlane((int)wlo, 16));
        smax64(ghi, glo, (uint32_t)__builtin_amdgcn_readlane((int)whi, 32), (uint32_t)__builtin_amdgcn_readlane((int)wlo, 32));
        smax64(ghi, glo, (uint32_t)__builtin_amdgcn_readlane((int)whi, 48), (uint32_t)__builtin_amdgcn_readlane((int)wlo, 48));
        const int g = r & 1;
        if (LXYZ) {
            if (lane == 0) {
                s_hi[g][wave] = ghi;
                s_lo[g][wave] = glo;
            }
        } else if (blo == glo && bhi == ghi) {   // exactly one lane of the wave owns the wave's winner (indices are unique)
            s_hi[g][wave] = bhi;
            s_lo[g][wave] = blo;
            s_x[g][wave] = bx;
            s_y[g][wave] = by;
            s_z[g][wave] = bz;
        }
        __syncthreads();
        // ---- fold the wave slots: lane l of every row reads slot l & 15, one row-of-16 DPP max, every lane has it ----
        constexpr int FOLD = NW <= 4 ? 4 : (NW <= 8 ? 8 : 16);          // slots replicated every FOLD lanes
        const uint32_t mhi = s_hi[g][lane & (FOLD - 1)], mlo = s_lo[g][lane & (FOLD - 1)];
        uint32_t fhi = mhi, flo = mlo;
        if (FOLD == 4) row4_max64(fhi, flo);
        else if (FOLD == 8) row8_max64(fhi, flo);
        else row16_max64(fhi, flo);
        const int gi = (int)~flo;
        if (LXYZ) {
            lx = s_cloud[gi];
            ly = s_cloud[n + gi];
            lz = s_cloud[2 * n + gi];
        } else {
            const unsigned long long own = __ballot(mlo == flo && mhi == fhi);
            const int gw = __ffsll((long long)own) - 1;        // slot of the winner (lanes 0..15 carry the slots)
            lx = s_x[g][gw];
            ly = s_y[g][gw];
            lz = s_z[g][gw];
        }
        if (tid == 0) out[r] = gi;
        // retire the winner in its owner's registers: compile-time k, run-time predicate
        const int own_t = gi % T, own_k = gi / T;
#pragma unroll
        for (int k = 0; k < P; ++k) dist[k] = (tid == own_t && k == own_k) ? -1.0f : dist[k];
    }
}

template <int T, int P>
static int launch(const float *xyz, int n_clouds, int n, int ld, int s, int32_t *idx, hipStream_t st)
{
    const size_t lds = (size_t)n * 3 * sizeof(float);
    if (lds <= 144 * 1024) {
        static bool attr_set = false;
        auto kern = fps_kernel<T, P, true>;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
            if (e != hipSuccess) return fail(AMPNET_E_LAUNCH, "fps: hipFuncSetAttribute: %s", hipGetErrorString(e));
            attr_set = true;
        }
        hipLaunchKernelGGL(kern, dim3(n_clouds), dim3(T), lds, st, xyz, n, ld, s, idx);
    } else {
        hipLaunchKernelGGL((fps_kernel<T, P, false>), dim3(n_clouds), dim3(T), 0, st, xyz, n, ld, s, idx);
    }
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
    // 16 points per thread from 4096 points on: a round is bound by the argmax reduction, and half the waves halve it
    if (n <= 4096) return launch<512, 8>(xyz, n_clouds, n, ld, s, idx, st);
    if (n <= 8192) return launch<512, 16>(xyz, n_clouds, n, ld, s, idx, st);
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
