// Probe of ds_read_b64_tr_b16 + v_mfma_f32_32x32x16_bf16 operand maps as pw_bwd_bf16.hip uses them (run on the MI355X):
//   hipcc --offload-arch=gfx950 -O2 tools/tr_probe.hip -o tools/tr_probe && tools/tr_probe
// C[m][n] = sum_k G[k][m] * Y[k][n] with G, Y row-major [16][32] bf16 tiles read transposed must equal the host product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int LD = 32 + 32;
__device__ bf16x8 tr_operand(const __bf16 *tile, int ld, int row0, int col0, int lane)
{
    const int g4 = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const __bf16 *src = tile + (row0 + 8 * (g4 >> 1) + q) * ld + col0 + 16 * (g4 & 1) + 4 * p;
    typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(src));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(src + 4 * ld));
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    // whole-vector bit casts + one shuffle: an element-by-element short -> __bf16 copy is miscompiled by this hipcc (it keeps only the
    // first dword of each read)
    return __builtin_shufflevector(__builtin_bit_cast(bf16x4, lo), __builtin_bit_cast(bf16x4, hi), 0, 1, 2, 3, 4, 5, 6, 7);
}
__global__ void probe(const float *G, const float *Y, float *C)
{
    __shared__ __attribute__((aligned(16))) __bf16 sG[16 * LD], sY[16 * LD];
    for (int e = threadIdx.x; e < 16 * 32; e += 64) { sG[(e / 32) * LD + e % 32] = (__bf16)G[e]; sY[(e / 32) * LD + e % 32] = (__bf16)Y[e]; }
    __syncthreads();
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_operand(sG, LD, 0, 0, lane), tr_operand(sY, LD, 0, 0, lane), acc, 0, 0, 0);
    for (int e = 0; e < 16; ++e) C[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r] = acc[e];
}
int main()
{
    float G[512], Y[512], C[1024], *dG, *dY, *dC;
    for (int i = 0; i < 512; ++i) { G[i] = (float)((i * 7 + 3) % 13 - 6); Y[i] = (float)((i * 5 + 1) % 11 - 5); }   // small integers: exact in bf16
    hipMalloc(&dG, sizeof G); hipMalloc(&dY, sizeof Y); hipMalloc(&dC, sizeof C);
    hipMemcpy(dG, G, sizeof G, hipMemcpyHostToDevice); hipMemcpy(dY, Y, sizeof Y, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dG, dY, dC);
    hipMemcpy(C, dC, sizeof C, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int m = 0; m < 32; ++m)
        for (int n = 0; n < 32; ++n) {
            float want = 0;
            for (int k = 0; k < 16; ++k) want += G[k * 32 + m] * Y[k * 32 + n];
            if (C[m * 32 + n] != want && bad++ < 5) printf("mismatch C[%d][%d] = %g, want %g\n", m, n, C[m * 32 + n], want);
        }
    printf(bad ? "tr_probe: %d mismatches\n" : "tr_probe: OK (transposed operands + 32x32x16 bf16 MFMA as assumed)\n", bad);
    return bad != 0;
}
