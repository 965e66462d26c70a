// Raw map of ds_read_b64_tr_b16 with STRIDED rows: lane 4q+p of each 16-lane group reads row (8*(g>>1) + q), columns 16*(g&1) + 4p .. +3 of a
// [16][64] tile of ids (id = 100 * row + col); prints, per lane, the 4 ids it received.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(short *out)
{
    __shared__ __attribute__((aligned(16))) short lds[16 * 64];
    for (int e = threadIdx.x; e < 16 * 64; e += 64) lds[e] = (short)(100 * (e / 64) + e % 64);
    __syncthreads();
    const int lane = threadIdx.x, g4 = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(lds + (8 * (g4 >> 1) + q) * 64 + 16 * (g4 & 1) + 4 * p));
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = v[i];
}
int main()
{
    short h[256], *d;
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
    return 0;
}
