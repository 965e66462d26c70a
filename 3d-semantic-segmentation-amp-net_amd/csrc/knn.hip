// knn.hip -- exact brute-force k-nearest-neighbour grouping of FPS centres (C ABI: ampnet_knn_f32).
//
// BUILD-DEFINED: the reference has no k-NN or ball query anywhere (data_proc/sample_fps.py only calls
// utils.fps; window grouping is KMeansConstrained).  BASELINE.json's north_star and config 5 name "FPS + k-NN grouping"
// as the large-window stress kernel, so the spec is fixed here and pinned by the build's own CPU restatement
// (oracle/fps_oracle.py: knn_indices) -- parity against the reference is "unpinned" by construction:
//     out[c][i][0..k) = the k points j of cloud c with the smallest (|p_j - p_centre(i)|^2, j), ascending;
//     distance = float32 ((dx*dx + dy*dy) + dz*dz), one rounding per operation (this file is compiled with
//     -ffp-contract=off like fps.hip); equal distances go to the lower index; the centre itself (distance 0) comes first.
//
// Mapping to CDNA4: the cloud's coordinates sit in LDS as three planes (n * 12 bytes <= 144 KB), one workgroup serves a
// slice of the centres of one cloud, one WAVE per centre.  Lane l scans candidates l, l + 64, ... (conflict-free LDS
// reads) keeping its four smallest (distance, index) keys in registers; then k rounds of a 64-lane DPP minimum pop the
// global order.  A lane that runs out of its four (rare: the k winners spread over 64 lanes) rescans for its next four.
// HBM traffic is the cloud once per workgroup; the algorithmic figure SURVEY.md section 8(d) counts is 12 B per
// (candidate, centre), served from LDS.
#include "common.h"

#pragma clang fp contract(off)

namespace ampnet {

constexpr int KNN_WAVES = 16;     // 4 waves per SIMD hide the LDS latency of the candidate scan
constexpr unsigned long long KNN_INF = ~0ull;

__device__ __forceinline__ unsigned long long dpp_min64_step(unsigned long long v, int ctrl_sel)
{
    uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v, ohi, olo;
    switch (ctrl_sel) {
    case 0:
        ohi = (uint32_t)__builtin_amdgcn_update_dpp((int)hi, (int)hi, 0xB1, 0xF, 0xF, false);
        olo = (uint32_t)__builtin_amdgcn_update_dpp((int)lo, (int)lo, 0xB1, 0xF, 0xF, false);
        break;
    case 1:
        ohi = (uint32_t)__builtin_amdgcn_update_dpp((int)hi, (int)hi, 0x4E, 0xF, 0xF, false);
        olo = (uint32_t)__builtin_amdgcn_update_dpp((int)lo, (int)lo, 0x4E, 0xF, 0xF, false);
        break;
    case 2:
        ohi = (uint32_t)__builtin_amdgcn_update_dpp((int)hi, (int)hi, 0x141, 0xF, 0xF, false);
        olo = (uint32_t)__builtin_amdgcn_update_dpp((int)lo, (int)lo, 0x141, 0xF, 0xF, false);
        break;
    default:
        ohi = (uint32_t)__builtin_amdgcn_update_dpp((int)hi, (int)hi, 0x140, 0xF, 0xF, false);
        olo = (uint32_t)__builtin_amdgcn_update_dpp((int)lo, (int)lo, 0x140, 0xF, 0xF, false);
        break;
    }
    const unsigned long long o = ((unsigned long long)ohi << 32) | olo;
    return o < v ? o : v;
}

// minimum of a 64-bit key over the wave, in every lane: four DPP steps inside the rows of 16, then four scalars
__device__ __forceinline__ unsigned long long wave_min64(unsigned long long v)
{
    v = dpp_min64_step(v, 0);
    v = dpp_min64_step(v, 1);
    v = dpp_min64_step(v, 2);
    v = dpp_min64_step(v, 3);
    unsigned long long m = KNN_INF;
#pragma unroll
    for (int l = 0; l < 64; l += 16) {
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;
        m = o < m ? o : m;
    }
    return m;
}

__global__ __launch_bounds__(64 * KNN_WAVES) void knn_kernel(const float *__restrict__ xyz, int n, int ld, const int32_t *__restrict__ centres,
                                                            int s, int k, int centres_per_block, int32_t *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float s_cloud[];      // x[n], y[n], z[n]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cloud_i = blockIdx.y;
    const float *cloud = xyz + (size_t)cloud_i * n * ld;
    for (int j = tid; j < n; j += 64 * KNN_WAVES) {
        s_cloud[j] = cloud[(size_t)j * ld + 0];
        s_cloud[n + j] = cloud[(size_t)j * ld + 1];
        s_cloud[2 * n + j] = cloud[(size_t)j * ld + 2];
    }
    __syncthreads();
    const int c_begin = blockIdx.x * centres_per_block, c_end = min(c_begin + centres_per_block, s);
    for (int ci = c_begin + wave; ci < c_end; ci += KNN_WAVES) {
        const int cidx = centres[(size_t)cloud_i * s + ci];
        const float cx = s_cloud[cidx], cy = s_cloud[n + cidx], cz = s_cloud[2 * n + cidx];
        // the lane's four smallest keys greater than `floor_key` (first scan: everything)
        unsigned long long t0, t1, t2, t3, floor_key = 0;
        bool first = true;
        // branch-free sorted insert of `key` into t0 <= t1 <= t2 <= t3 (keys are unique; KNN_INF never displaces anything)
        auto insert = [&](unsigned long long key) {
            unsigned long long a = key < t3 ? key : t3;         // new t3 candidate
            unsigned long long lo = a < t2 ? a : t2, hi = a < t2 ? t2 : a;
            t3 = hi;
            a = lo;
            lo = a < t1 ? a : t1, hi = a < t1 ? t1 : a;
            t2 = hi;
            a = lo;
            t1 = a < t0 ? t0 : a;
            t0 = a < t0 ? a : t0;
        };
        auto scan = [&]() {
            t0 = t1 = t2 = t3 = KNN_INF;
            // four candidates per trip: their twelve LDS reads are in flight together
            for (int j0 = lane; j0 < n; j0 += 256) {
                float px[4], py[4], pz[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = min(j0 + 64 * u, n - 1);
                    px[u] = s_cloud[j];
                    py[u] = s_cloud[n + j];
                    pz[u] = s_cloud[2 * n + j];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = j0 + 64 * u;
                    const float dx = cx - px[u], dy = cy - py[u], dz = cz - pz[u];
                    const float d = (dx * dx + dy * dy) + dz * dz;
                    unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (uint32_t)j;
                    if (j >= n || (!first && key <= floor_key)) key = KNN_INF;
                    insert(key);
                }
            }
        };
        scan();
        first = false;
        int held = 4;                                   // keys of the current batch not yet popped
        unsigned long long last_popped = 0;
        int32_t *dst = out + ((size_t)cloud_i * s + ci) * k;
        for (int r = 0; r < k; ++r) {
            // a lane whose batch is used up may still own smaller keys than the other lanes' heads: refill first
            const bool need = held == 0 && t0 == KNN_INF && last_popped != KNN_INF;
            if (__any(need)) {
                if (need) {
                    floor_key = last_popped;
                    scan();
                    held = 4;
                    if (t0 == KNN_INF) last_popped = KNN_INF;      // nothing left in this lane's stripe
                }
            }
            const unsigned long long m = wave_min64(t0);
            if (lane == 0) dst[r] = m == KNN_INF ? -1 : (int32_t)(uint32_t)m;
            if (t0 == m && m != KNN_INF) {                      // keys are unique: exactly one lane pops
                last_popped = t0;
                t0 = t1;
                t1 = t2;
                t2 = t3;
                t3 = KNN_INF;
                --held;
            }
        }
    }
}

}  // namespace ampnet

extern "C" int ampnet_knn_f32(const float *xyz, int n_clouds, int n, int ld, const int32_t *centres, int s, int k, int32_t *out,
                              void *stream)
{
    using namespace ampnet;
    AMPNET_REQUIRE(xyz && centres && out, "ampnet_knn_f32: null pointer");
    AMPNET_REQUIRE(n_clouds >= 1 && n >= 1 && ld >= 3 && s >= 1, "ampnet_knn_f32: bad shape n_clouds=%d n=%d ld=%d s=%d", n_clouds, n, ld, s);
    AMPNET_REQUIRE(k >= 1 && k <= n, "ampnet_knn_f32: k=%d must be in [1, n=%d]", k, n);
    const size_t lds = (size_t)n * 3 * sizeof(float);
    AMPNET_REQUIRE(lds <= 144 * 1024, "ampnet_knn_f32: n=%d exceeds %d points per cloud (coordinates must fit LDS)", n, 144 * 1024 / 12);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(knn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
        if (e != hipSuccess) return fail(AMPNET_E_LAUNCH, "ampnet_knn_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    // enough workgroups to fill 256 CUs, each amortising its copy of the cloud over >= 32 centres
    int per_block = cdiv(s * n_clouds, 1024);
    if (per_block < 32) per_block = 32;
    if (per_block > s) per_block = s;
    hipLaunchKernelGGL(knn_kernel, dim3(cdiv(s, per_block), n_clouds), dim3(64 * KNN_WAVES), lds, (hipStream_t)stream, xyz, n, ld, centres, s, k,
                       per_block, out);
    return check_launch("knn_kernel");
}
