// fps.hip -- greedy farthest-point sampling, one workgroup per cloud (C ABI: ampnet_fps_f32).
//
// Replaces utils/utils.py:889-933 of the reference.  The algorithm is S-1 dependent rounds; a round is
//   d[j] = min(d[j], |p_last - p_j|^2)  for every point, then argmax_j d[j] (first index on ties).
// Mapping to CDNA4: the cloud lives in REGISTERS for the whole kernel (12 B xyz + 4 B running minimum per point = the
// 16 B per (candidate, round) the roofline counts, served from the register file instead of HBM); a round is VALU work
// (12 instructions per point) plus ONE dependent chain: wave reduction of the value by 32-bit DPP -> one LDS slot per
// wave -> one barrier -> 16-lane DPP fold -> one LDS read of the winner's coordinates (fps_kernel below).  Clouds of more
// than 16384 points keep only the running minima in registers and stream the coordinates from L2 (fps_stream_kernel).
//
// Bit parity: distances are float32 ((dx*dx + dy*dy) + dz*dz) with one rounding per operation -- this file
// is compiled with -ffp-contract=off and the pragma below repeats it; ties go to the lowest index.
#include <cstdlib>
#include "common.h"

#pragma clang fp contract(off)

typedef float f32x2 __attribute__((ext_vector_type(2)));

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

// ---- 32-bit DPP reductions (value only): one VALU instruction per step -------------------------------------------
// Running minima are compared as INTEGERS: the bit pattern of a float >= 0 orders like the float, and the negative markers
// (-2 padding, -3 / -4 "nothing yet") are negative integers, below every real distance.  Integer max needs no NaN
// canonicalisation (fmaxf costs an extra v_max per operand) and folds into v_max_i32_dpp.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int dpp_i32(int v)
{
    return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xF, ROW_MASK == 0xF);     // full masks + bound_ctrl: `old` is dead, the move folds into the consumer
}
// maximum over the 64 lanes, valid in lane 63 (rows of 16 by quad / mirror steps, then row_bcast:15 and row_bcast:31)
__device__ __forceinline__ int wave_max_i32(int v)
{
    v = max(v, dpp_i32<0xB1>(v));            // quad_perm [1,0,3,2]
    v = max(v, dpp_i32<0x4E>(v));            // quad_perm [2,3,0,1]
    v = max(v, dpp_i32<0x141>(v));           // row_half_mirror
    v = max(v, dpp_i32<0x140>(v));           // row_mirror: every lane of a row holds the row's maximum
    v = max(v, dpp_i32<0x142, 0xA>(v));      // row_bcast:15 into rows 1 and 3
    v = max(v, dpp_i32<0x143, 0xC>(v));      // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}

// v_min_f32 as it stands: fminf() lowers to a canonicalising v_max_f32 x, x in front of every v_min (IEEE minNum for signalling
// NaNs), one more VALU instruction per point and round.  Distances are never NaN here.
__device__ __forceinline__ float min_f32(float a, float b)
{
    float m;
    asm("v_min_f32_e32 %0, %1, %2" : "=v"(m) : "v"(a), "v"(b));
    return m;
}

// One workgroup per cloud, the cloud in REGISTERS (thread t owns the P consecutive points t*P ..), a copy of the
// coordinates in LDS as float4 (LDSXYZ, n <= 9216) or re-read from global memory / L2 for the winner only.
// A round = P x (distance, running minimum, strict-> argmax: 12 VALU per point)
//   -> wave maximum of the VALUE by six 32-bit DPP steps + one readlane; ballot of the lanes that hold it, lowest lane,
//      readlane of its k: the wave's (maximum, lowest index) in scalars, one 8-byte LDS slot per wave
//   -> ONE barrier (slots double-buffered by round parity)
//   -> every lane reads slot lane & (FOLD - 1); log2(FOLD) DPP steps give the maximum, as many more the lowest index among its holders
//   -> one broadcast ds_read_b128 of the winner's coordinates.
// Picked points need no marker: a picked point's running minimum is exactly 0 after the next update (its distance to itself),
// so while the maximum M is > 0 no picked point can win.  M == 0 means every point left is a duplicate of a picked one and
// the running minima can never change again: the remaining picks are the unpicked indices in ascending order (what the
// reference's argmax over an all-zero `dists[points_left]` returns, utils.py:927-931), emitted from a bitmap of the picks.
// STAMP (diagnostic build, ampnet_fps_round_stamps): thread 0 writes s_memtime after the update, after the barrier, after the
// fold and after the coordinate read of every round into a buffer nothing else reads.
template <int T, int P, bool LDSXYZ, bool STAMP = false>
__global__ __launch_bounds__(T) void fps_kernel(const float *__restrict__ xyz, int n, int ld, int s,
                                                int32_t *__restrict__ idx, unsigned long long *__restrict__ stamps = nullptr,
                                                const int *__restrict__ cloud_off = nullptr, const int *__restrict__ out_off = nullptr)
{
    constexpr int NW = T / WAVE;
    static_assert(NW <= 16, "one row of 16 lanes folds the wave slots");
    extern __shared__ __attribute__((aligned(16))) float s_cloud[];      // LDSXYZ: float4[n]
    __shared__ uint2 s_slot[2][16];                                      // (bits of the wave's maximum, its lowest index)
    __shared__ uint32_t s_picked[16384 / 32];

    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wave = tid / WAVE;
    const float *cloud = xyz + (size_t)blockIdx.x * n * ld;
    int32_t *out = idx + (size_t)blockIdx.x * s;
    if (cloud_off) {
        // ragged batch (ampnet_fps_ragged_f32): cloud b = rows cloud_off[b] .. cloud_off[b + 1] of one [total, ld] array, its samples go to
        // idx[out_off[b] .. out_off[b + 1]); n <= T * P is the launcher's promise (the template is picked for the largest cloud)
        const int c0 = cloud_off[blockIdx.x], o0 = out_off[blockIdx.x];
        const int n_launch = n;                      // max_n of the launch: registers (T * P) and the LDS image are sized for it
        n = cloud_off[blockIdx.x + 1] - c0;
        s = min(out_off[blockIdx.x + 1] - o0, n);
        cloud = xyz + (size_t)c0 * ld;
        out = idx + o0;
        if (s <= 0) return;                          // uniform per workgroup
        if (n > n_launch) {                          // a cloud larger than the caller's max_n (the host cannot see device offsets): refuse it
            if (tid == 0) out[0] = -1;               // instead of overrunning LDS; its first index reads -1
            return;
        }
    }

    // the P points of a thread as P / 2 PAIRS: differences, squares and the two sums are v_pk_add_f32 / v_pk_mul_f32 on a pair (correctly
    // rounded per element, so the distances stay bit-identical to numpy's) -- 8 instead of 16 VALU instructions per pair
    constexpr int P2 = (P + 1) / 2;                             // odd P: the last pair's second element is padding
    f32x2 px[P2], py[P2], pz[P2], dist[P2];
#pragma unroll
    for (int k = 0; k < 2 * P2; ++k) {
        const int j = tid * P + k;
        const bool ok = k < P && j < n;
        const float x = ok ? cloud[(size_t)j * ld + 0] : 0.f, y = ok ? cloud[(size_t)j * ld + 1] : 0.f, z = ok ? cloud[(size_t)j * ld + 2] : 0.f;
        px[k / 2][k & 1] = x;
        py[k / 2][k & 1] = y;
        pz[k / 2][k & 1] = z;
        dist[k / 2][k & 1] = ok ? __builtin_inff() : -2.0f;   // -2: padding never wins (M > 0 in every round that picks), never changes
        if (LDSXYZ && ok) *reinterpret_cast<float4 *>(s_cloud + 4 * (size_t)j) = make_float4(x, y, z, 0.f);
    }
    for (int w = tid; w < 16384 / 32; w += T) s_picked[w] = w == 0 ? 1u : 0u;      // seed: point 0 (utils.py:907-908)
    if (tid < 32) s_slot[tid >> 4][tid & 15] = make_uint2(__float_as_uint(-4.0f), 0x7FFFFFFFu);   // unused slots never win
    float lx = cloud[0], ly = cloud[1], lz = cloud[2];
    if (tid == 0) out[0] = 0;
    __syncthreads();

    int r = 1;
    for (; r < s; ++r) {
        // ---- update + thread-local argmax, branch-free: ascending k and a strict compare keep the lowest index ----
        float bd = -3.0f;
        int bi = 0;
        const f32x2 lx2 = {lx, lx}, ly2 = {ly, ly}, lz2 = {lz, lz};
#pragma unroll
        for (int k2 = 0; k2 < P2; ++k2) {
            const f32x2 dx = lx2 - px[k2];
            const f32x2 dy = ly2 - py[k2];
            const f32x2 dz = lz2 - pz[k2];
            const f32x2 d = (dx * dx + dy * dy) + dz * dz;
            const float m0 = min_f32(d[0], dist[k2][0]), m1 = min_f32(d[1], dist[k2][1]);
            dist[k2] = f32x2{m0, m1};
            const bool t0 = m0 > bd;
            bd = t0 ? m0 : bd;
            bi = t0 ? 2 * k2 : bi;
            const bool t1 = m1 > bd;
            bd = t1 ? m1 : bd;
            bi = t1 ? 2 * k2 + 1 : bi;
        }
        if (STAMP && tid == 0) stamps[4 * (size_t)r + 0] = __builtin_amdgcn_s_memtime();
        // ---- the wave's maximum and the lowest index that holds it (indices ascend with the lane, then with k) ----
        const int bdi = __float_as_int(bd);
        const int wm = wave_max_i32(bdi);
        const unsigned long long holders = __ballot(bdi == wm);
        const int wl = __ffsll((long long)holders) - 1;
        const int wk = __builtin_amdgcn_readlane(bi, wl);
        const int g = r & 1;
        if (lane == 0) s_slot[g][wave] = make_uint2((uint32_t)wm, (uint32_t)((wave * WAVE + wl) * P + wk));
        __syncthreads();
        if (STAMP && tid == 0) stamps[4 * (size_t)r + 1] = __builtin_amdgcn_s_memtime();
        // ---- fold the slots (replicated in every row of 16 lanes): maximum, then lowest index among its holders ----
        constexpr int FOLD = NW <= 4 ? 4 : (NW <= 8 ? 8 : 16);          // slots replicated every FOLD lanes: every lane ends with the result
        const uint2 sl = s_slot[g][lane & (FOLD - 1)];
        const int sm = (int)sl.x;
        int Mi = sm;
        Mi = max(Mi, dpp_i32<0xB1>(Mi));
        Mi = max(Mi, dpp_i32<0x4E>(Mi));
        if (NW > 4) Mi = max(Mi, dpp_i32<0x141>(Mi));
        if (NW > 8) Mi = max(Mi, dpp_i32<0x140>(Mi));
        int gi = sm == Mi ? (int)sl.y : 0x7FFFFFFF;
        gi = min(gi, dpp_i32<0xB1>(gi));
        gi = min(gi, dpp_i32<0x4E>(gi));
        if (NW > 4) gi = min(gi, dpp_i32<0x141>(gi));
        if (NW > 8) gi = min(gi, dpp_i32<0x140>(gi));
        const float M = __int_as_float(Mi);
        if (STAMP && tid == 0) stamps[4 * (size_t)r + 2] = __builtin_amdgcn_s_memtime() + (gi & 0);   // after the fold
        if (!(M > 0.0f)) break;                                  // uniform: only duplicates of picked points are left
        if (LDSXYZ) {
            const float4 c = *reinterpret_cast<const float4 *>(s_cloud + 4 * (size_t)gi);
            lx = c.x;
            ly = c.y;
            lz = c.z;
        } else {
            lx = cloud[(size_t)gi * ld + 0];
            ly = cloud[(size_t)gi * ld + 1];
            lz = cloud[(size_t)gi * ld + 2];
        }
        if (tid == 0) {
            out[r] = gi;
            s_picked[gi >> 5] |= 1u << (gi & 31);                // only this thread writes the bitmap
        }
        if (STAMP && tid == 0) stamps[4 * (size_t)r + 3] = __builtin_amdgcn_s_memtime() + (__float_as_int(lx) & 0);   // coordinates landed
    }
    if (r < s) {                                                 // the all-duplicates tail (see above); rare, serial
        __syncthreads();
        if (tid == 0) {
            for (int j = 0; j < n && r < s; ++j)
                if (!((s_picked[j >> 5] >> (j & 31)) & 1u)) out[r++] = j;
        }
    }
}

// n > 16384: nothing of the cloud fits on chip for the whole kernel, so a round STREAMS it: coordinates and running minima
// live in a structure-of-arrays workspace in global memory (x[n], y[n], z[n], d[n] per cloud; 16 B per point, it stays in
// the XCD's L2), thread t walks points t, t + T, ... with coalesced loads, rewrites a running minimum only when it changed,
// and the reduction carries 64-bit (distance, ~index) keys (lowest index on ties for any ownership).  A picked point is
// retired by its owner during the next round's walk (its minimum becomes -1: every real distance is >= 0).
template <int T>
__global__ __launch_bounds__(T) void fps_stream_kernel(float *__restrict__ soa, int n, int s, int32_t *__restrict__ idx,
                                                       const int *__restrict__ cloud_off = nullptr, const int *__restrict__ out_off = nullptr)
{
    __shared__ uint32_t s_hi[2][16], s_lo[2][16];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    size_t base = (size_t)blockIdx.x * 4 * n;
    int32_t *out = idx + (size_t)blockIdx.x * s;
    if (cloud_off) {                                 // ragged batch: the workspace holds 4 * n_b floats per cloud, in cloud order
        const int c0 = cloud_off[blockIdx.x], o0 = out_off[blockIdx.x];
        const int n_launch = n;
        n = cloud_off[blockIdx.x + 1] - c0;
        s = min(out_off[blockIdx.x + 1] - o0, n);
        base = (size_t)c0 * 4;
        out = idx + o0;
        if (s <= 0) return;
        if (n > n_launch) {                          // beyond the caller's max_n: refused like in fps_kernel (first index -1)
            if (tid == 0) out[0] = -1;
            return;
        }
    }
    const float *X = soa + base, *Y = X + n, *Z = Y + n;
    float *D = soa + base + 3 * (size_t)n;
    float lx = X[0], ly = Y[0], lz = Z[0];
    int last = 0;
    if (tid == 0) out[0] = 0;
    if (tid < 32) {
        s_hi[tid >> 4][tid & 15] = 0u;
        s_lo[tid >> 4][tid & 15] = 0u;
    }
    __syncthreads();
    for (int r = 1; r < s; ++r) {
        float bd = -3.0f;
        int bj = 0;
#pragma unroll 4
        for (int j = tid; j < n; j += T) {
            const float old = D[j];
            const float dx = lx - X[j];
            const float dy = ly - Y[j];
            const float dz = lz - Z[j];
            const float d = (dx * dx + dy * dy) + dz * dz;
            const float m = j == last ? -1.0f : fminf(d, old);       // picked points stay at -1: min(d, -1) = -1
            if (m != old) D[j] = m;
            const bool take = m > bd;                                 // ascending j, strict: lowest index of the thread
            bd = take ? m : bd;
            bj = take ? j : bj;
        }
        uint32_t whi = fkey(bd), wlo = ~(uint32_t)bj;
        row16_max64(whi, wlo);
        uint32_t ghi = (uint32_t)__builtin_amdgcn_readlane((int)whi, 0), glo = (uint32_t)__builtin_amdgcn_readlane((int)wlo, 0);
        smax64(ghi, glo, (uint32_t)__builtin_amdgcn_readlane((int)whi, 16), (uint32_t)__builtin_amdgcn_readlane((int)wlo, 16));
        smax64(ghi, glo, (uint32_t)__builtin_amdgcn_readlane((int)whi, 32), (uint32_t)__builtin_amdgcn_readlane((int)wlo, 32));
        smax64(ghi, glo, (uint32_t)__builtin_amdgcn_readlane((int)whi, 48), (uint32_t)__builtin_amdgcn_readlane((int)wlo, 48));
        const int g = r & 1;
        if (lane == 0) {
            s_hi[g][wave] = ghi;
            s_lo[g][wave] = glo;
        }
        __syncthreads();
        uint32_t fhi = s_hi[g][lane & 15], flo = s_lo[g][lane & 15];
        row16_max64(fhi, flo);
        last = (int)~flo;
        lx = X[last];
        ly = Y[last];
        lz = Z[last];
        if (tid == 0) out[r] = last;
    }
}

// [n_clouds, n, ld] rows -> [n_clouds, 4, n] structure of arrays x, y, z, running minimum (the stream kernel's workspace)
__global__ void fps_soa_kernel(const float *__restrict__ xyz, int n, int ld, float *__restrict__ soa, const int *__restrict__ cloud_off = nullptr)
{
    const int c = blockIdx.y;
    size_t row0 = (size_t)c * n;
    if (cloud_off) {
        row0 = (size_t)cloud_off[c];
        n = cloud_off[c + 1] - cloud_off[c];
    }
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const float *p = xyz + (row0 + j) * ld;
        float *o = soa + row0 * 4;
        o[j] = p[0];
        o[n + j] = p[1];
        o[2 * (size_t)n + j] = p[2];
        o[3 * (size_t)n + j] = j == 0 ? -1.0f : __builtin_inff();       // seed: point 0 is picked (utils.py:907-908)
    }
}

template <int T, int P>
static int launch(const float *xyz, int n_clouds, int n, int ld, int s, int32_t *idx, hipStream_t st, const int *cloud_off = nullptr,
                  const int *out_off = nullptr)
{
    const size_t lds = (size_t)n * 4 * sizeof(float);
    if (lds <= 144 * 1024) {
        static bool attr_set = false;
        auto kern = fps_kernel<T, P, true>;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
            if (e != hipSuccess) return fail(AMPNET_E_LAUNCH, "fps: hipFuncSetAttribute: %s", hipGetErrorString(e));
            attr_set = true;
        }
        hipLaunchKernelGGL(kern, dim3(n_clouds), dim3(T), lds, st, xyz, n, ld, s, idx, (unsigned long long *)nullptr, cloud_off, out_off);
    } else {
        hipLaunchKernelGGL((fps_kernel<T, P, false>), dim3(n_clouds), dim3(T), 0, st, xyz, n, ld, s, idx, (unsigned long long *)nullptr, cloud_off, out_off);
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

extern "C" size_t ampnet_fps_workspace_bytes(int n_clouds, int n)
{
    if (n_clouds < 1 || n <= AMPNET_FPS_RESIDENT_MAX) return 0;
    return (size_t)n_clouds * 4 * (size_t)n * sizeof(float);
}

namespace ampnet {
// n = points per cloud (uniform batch) or the LARGEST cloud of a ragged batch (cloud_off / out_off device arrays, n_clouds + 1 entries)
static int fps_dispatch(const float *xyz, int n_clouds, int n, int ld, int s, int32_t *idx, float *soa, hipStream_t st, const int *cloud_off,
                        const int *out_off)
{
    if (n > AMPNET_FPS_RESIDENT_MAX) {
        hipLaunchKernelGGL(fps_soa_kernel, dim3(cdiv(n, 256) < 256 ? cdiv(n, 256) : 256, n_clouds), dim3(256), 0, st, xyz, n, ld, soa, cloud_off);
        hipLaunchKernelGGL((fps_stream_kernel<1024>), dim3(n_clouds), dim3(1024), 0, st, soa, n, s, idx, cloud_off, out_off);
        return check_launch("fps_stream_kernel");
    }
    if (n <= 256) return launch<256, 1>(xyz, n_clouds, n, ld, s, idx, st, cloud_off, out_off);
    if (n <= 1024) return launch<256, 4>(xyz, n_clouds, n, ld, s, idx, st, cloud_off, out_off);
    if (n <= 2048) return launch<512, 4>(xyz, n_clouds, n, ld, s, idx, st, cloud_off, out_off);
    if (n <= 4096) return launch<512, 8>(xyz, n_clouds, n, ld, s, idx, st, cloud_off, out_off);
    static const int t8k = [] { const char *e = getenv("AMPNET_FPS_T8K"); return e ? atoi(e) : 512; }();      // tuning hook
    if (n <= 8192) {
        if (t8k == 1024) return launch<1024, 8>(xyz, n_clouds, n, ld, s, idx, st, cloud_off, out_off);
        if (t8k == 768) return launch<768, 11>(xyz, n_clouds, n, ld, s, idx, st, cloud_off, out_off);
        return launch<512, 16>(xyz, n_clouds, n, ld, s, idx, st, cloud_off, out_off);
    }
    return launch<1024, 16>(xyz, n_clouds, n, ld, s, idx, st, cloud_off, out_off);
}
}  // namespace ampnet

extern "C" int ampnet_fps_f32(const float *xyz, int n_clouds, int n, int ld, int s, int32_t *idx, void *workspace,
                              size_t workspace_bytes, void *stream)
{
    using namespace ampnet;
    AMPNET_REQUIRE(xyz && idx, "ampnet_fps_f32: null pointer");
    AMPNET_REQUIRE(n_clouds >= 1 && n >= 1 && ld >= 3, "ampnet_fps_f32: bad shape n_clouds=%d n=%d ld=%d", n_clouds, n, ld);
    AMPNET_REQUIRE(s >= 1 && s <= n, "ampnet_fps_f32: n_samples=%d must be in [1, n=%d]", s, n);
    AMPNET_REQUIRE(n <= AMPNET_FPS_MAX_POINTS, "ampnet_fps_f32: n=%d exceeds %d points per cloud", n, AMPNET_FPS_MAX_POINTS);
    if (n > AMPNET_FPS_RESIDENT_MAX) {
        const size_t need = ampnet_fps_workspace_bytes(n_clouds, n);
        if (!workspace || workspace_bytes < need) return fail(AMPNET_E_WORKSPACE, "ampnet_fps_f32: n=%d needs a workspace of %zu B (ampnet_fps_workspace_bytes)", n, need);
    }
    return fps_dispatch(xyz, n_clouds, n, ld, s, idx, reinterpret_cast<float *>(workspace), (hipStream_t)stream, nullptr, nullptr);
}

extern "C" size_t ampnet_fps_ragged_workspace_bytes(int total_rows, int max_n)
{
    if (total_rows < 1 || max_n <= AMPNET_FPS_RESIDENT_MAX) return 0;
    return (size_t)total_rows * 4 * sizeof(float);
}

extern "C" int ampnet_fps_ragged_f32(const float *rows, int ld, const int32_t *cloud_off, const int32_t *out_off, int n_clouds, int total_rows,
                                     int max_n, int32_t *idx, void *workspace, size_t workspace_bytes, void *stream)
{
    using namespace ampnet;
    AMPNET_REQUIRE(rows && cloud_off && out_off && idx, "ampnet_fps_ragged_f32: null pointer");
    AMPNET_REQUIRE(n_clouds >= 1 && ld >= 3 && max_n >= 1 && total_rows >= max_n, "ampnet_fps_ragged_f32: bad shape n_clouds=%d ld=%d max_n=%d total_rows=%d",
                   n_clouds, ld, max_n, total_rows);
    AMPNET_REQUIRE(max_n <= AMPNET_FPS_MAX_POINTS, "ampnet_fps_ragged_f32: max_n=%d exceeds %d points per cloud", max_n, AMPNET_FPS_MAX_POINTS);
    if (max_n > AMPNET_FPS_RESIDENT_MAX) {
        const size_t need = ampnet_fps_ragged_workspace_bytes(total_rows, max_n);
        if (!workspace || workspace_bytes < need)
            return fail(AMPNET_E_WORKSPACE, "ampnet_fps_ragged_f32: max_n=%d needs a workspace of %zu B (ampnet_fps_ragged_workspace_bytes)", max_n, need);
    }
    return fps_dispatch(rows, n_clouds, max_n, ld, max_n, idx, reinterpret_cast<float *>(workspace), (hipStream_t)stream, cloud_off, out_off);
}

extern "C" int ampnet_fps_round_stamps(const float *xyz, int n, int ld, int s, int32_t *idx, unsigned long long *stamps, void *stream)
{
    using namespace ampnet;
    AMPNET_REQUIRE(xyz && idx && stamps, "ampnet_fps_round_stamps: null pointer");
    AMPNET_REQUIRE(n > 4096 && n <= 8192 && ld >= 3 && s >= 1 && s <= n, "ampnet_fps_round_stamps: built for one cloud of 4097..8192 points");
    auto kern = fps_kernel<512, 16, true, true>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    if (e != hipSuccess) return fail(AMPNET_E_LAUNCH, "fps: hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(kern, dim3(1), dim3(512), (size_t)n * 16, (hipStream_t)stream, xyz, n, ld, s, idx, stamps, (const int *)nullptr, (const int *)nullptr);
    return check_launch("fps_kernel (stamps)");
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
