// Does fp32 VALU work overlap with v_mfma_f32_32x32x2_f32 on gfx950?  (and with v_mfma_f32_32x32x16_bf16?)
// hipcc --offload-arch=gfx950 -O3 -o tools/mfma_probe tools/mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int NV, bool BF>
__global__ __launch_bounds__(256) void probe(float *out, int iters, float seed)
{
    f32x16 acc0, acc1, acc2, acc3;
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; acc2[e] = 0.f; acc3[e] = 0.f; }
    float a = seed + threadIdx.x, b = seed * 0.5f;
    bf16x8 ab, bb;
    for (int e = 0; e < 8; ++e) { ab[e] = (__bf16)a; bb[e] = (__bf16)b; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i;
    const bool do_mfma = MODE != 1 && (MODE != 2 || (threadIdx.x >> 6) % 2 == 0 || true);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2) {
            if (BF) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc3, 0, 0, 0);
            } else {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc3, 0, 0, 0);
            }
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int k = 0; k < NV; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], 1.0001f, 0.5f);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
    for (int e = 0; e < 16; ++e) s += acc0[e] + acc1[e] + acc2[e] + acc3[e];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
    (void)do_mfma;
}

template <int MODE, int NV, bool BF>
float run(int wg_per_cu, const char *what)
{
    float *out;
    hipMalloc(&out, 4096);
    const int iters = 20000, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE, NV, BF><<<grid, 256>>>(out, 100, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<MODE, NV, BF><<<grid, 256>>>(out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-52s wg/cu %d: %8.3f ms   = %6.1f ns per iteration (4 MFMA + %d fma per wave)\n", what, wg_per_cu, ms, ms * 1e6 / iters, MODE == 0 ? 0 : NV);
    hipFree(out);
    return ms;
}

int main()
{
    for (int w = 1; w <= 2; ++w) {
        run<0, 0, false>(w, "fp32 MFMA only");
        run<1, 16, false>(w, "16 fma only");
        run<2, 16, false>(w, "fp32 MFMA + 16 fma");
        run<1, 32, false>(w, "32 fma only");
        run<2, 32, false>(w, "fp32 MFMA + 32 fma");
        run<0, 0, true>(w, "bf16 MFMA only");
        run<2, 16, true>(w, "bf16 MFMA + 16 fma");
        run<2, 32, true>(w, "bf16 MFMA + 32 fma");
    }
    return 0;
}
