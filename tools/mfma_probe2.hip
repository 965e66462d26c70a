// fp32 MFMA issue rate vs number of independent accumulators, and with LDS reads in the shadow (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int NLDS>
__global__ __launch_bounds__(256) void probe(float *out, int iters, float seed)
{
    __shared__ __attribute__((aligned(16))) float lds[64 * 132];
    for (int e = threadIdx.x; e < 64 * 132; e += 256) lds[e] = seed + e;
    __syncthreads();
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t)
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    float a = seed + threadIdx.x;
    f32x4 b[4];
    for (int t = 0; t < 4; ++t) b[t] = *reinterpret_cast<f32x4 *>(lds + (threadIdx.x & 31) * 132 + 4 * t);
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        // 16 MFMAs per iteration, round-robin over NACC accumulators
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[i & 3][i >> 2], acc[i % NACC], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int l = 0; l < NLDS; ++l) b[l & 3] = *reinterpret_cast<f32x4 *>(lds + ((lane & 31) + 32 * (l & 1)) * 132 + 4 * ((it + l) & 31));
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t)
        for (int e = 0; e < 16; ++e) s += acc[t][e];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int NACC, int NLDS>
void run(int wg_per_cu)
{
    float *out;
    (void)hipMalloc(&out, 4096);
    const int iters = 5000, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<NACC, NLDS><<<grid, 256>>>(out, 100, 1.0f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    probe<NACC, NLDS><<<grid, 256>>>(out, iters, 1.0f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("accumulators %d, ds_read_b128 per 16 MFMA %d, wg/cu %d: %7.1f ns per MFMA per SIMD\n", NACC, NLDS, wg_per_cu, ms * 1e6 / iters / 16 / wg_per_cu);
    (void)hipFree(out);
}

int main()
{
    for (int w = 1; w <= 2; ++w) {
        run<4, 0>(w); run<2, 0>(w); run<1, 0>(w);
        run<4, 6>(w); run<2, 6>(w); run<4, 12>(w);
    }
    return 0;
}
