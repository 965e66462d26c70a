// pw_gemm.hip -- the shared per-point MLP layer (Conv1d k=1 / Linear) on the fp32 matrix cores.
//
// Reference ops covered (pointNet/model/pointnetAtt.py): every conv_k / fc_k with K >= 64 of
// TransformationNet (:31-40), BasePointNet (:90-103, and the two torch.bmm at :85/:96 as per-window weights)
// and SegmentationWithAttention (:203-207), each with the PREVIOUS layer's BatchNorm+ReLU(+Dropout) fused as
// a prologue on the A operand and THIS layer's BatchNorm statistics / MaxPool1d fused as epilogue reductions.
//
// CDNA4 mapping
//   * v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD = the fp32 matrix peak).  A wave owns a
//     32-row x (32*NT)-column output tile: NT accumulators of 16 VGPRs.
//   * MFMA "A" = activations (rows = points), "B" = weights, so the accumulator has the OUTPUT CHANNEL on
//     the lane and 16 rows in registers: per-channel sums / max / argmax are in-lane reductions, and a store
//     instruction writes two full 128-byte row segments.
//   * K permutation: lane (r, h) loads 16 bytes A[row r][8j + 4h .. +3] straight from HBM/L2 into registers
//     and feeds them to four consecutive MFMAs; the weight fragment W[col][8j + 4h .. +3] comes from LDS
//     with one ds_read_b128 (row stride CIN + 4 floats: conflict-free).  The MFMA sums over k in a permuted
//     order, both operands agree on it.
//   * weights (<= 128 x 256 fp32 = 133 KB) are staged in LDS once per workgroup and reused for every row tile
//     of the workgroup's chunk; A fragments are prefetched one 128-byte line per row ahead of the MFMAs.
#include "kernels.h"

namespace ampnet {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int PW_NW = 4;   // waves per workgroup

__device__ __forceinline__ int pidx_of(int q, int n_slots, int Q, int slot_major)
{
    return slot_major ? (q % n_slots) * (Q / n_slots) + q / n_slots : q;
}

template <int CIN, int NT>
__global__ __launch_bounds__(PW_NW * 64, 2) void pw_gemm_kernel(PwGemm a)
{
    constexpr int CB = 32 * NT;
    constexpr int LDW = CIN + 4;
    constexpr int NBLK = CIN / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sW = smem;                  // [CB][LDW]
    float *sPro = smem + CB * LDW;     // scale[CIN], shift[CIN]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r = lane & 31;
    const int h = lane >> 5;
    const int q = blockIdx.y;
    const int chunk = blockIdx.x;
    const int cb0 = blockIdx.z * CB;

    const int w_begin = a.win_off[q];
    const int w_end = a.win_off[q + 1];
    const int row_begin = w_begin + chunk * a.chunk_rows;
    const int row_end = min(w_end, row_begin + a.chunk_rows);
    const int nrows = max(row_end - row_begin, 0);
    const int slot = (a.n_slots > 1) ? (q % a.n_slots) : 0;
    const int pidx = pidx_of(q, a.n_slots, a.Q, a.perwin_slot_major);

    // ---- stage weights (transposing when they are k-major) and the prologue affine ----
    if (nrows > 0) {
        if (a.w_win_stride == 0) {
            const float *Wg = a.W;
            for (int e = tid; e < CB * (CIN / 4); e += PW_NW * 64) {
                const int j = e / (CIN / 4), k4 = e % (CIN / 4);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (cb0 + j < a.cout) v = *reinterpret_cast<const f32x4 *>(Wg + (size_t)(cb0 + j) * a.ldw + 4 * k4);
                *reinterpret_cast<f32x4 *>(sW + j * LDW + 4 * k4) = v;
            }
        } else {
            const float *Wg = a.W + (size_t)pidx * a.w_win_stride;   // [CIN][cout]
            for (int e = tid; e < CIN * CB; e += PW_NW * 64) {
                const int k = e / CB, j = e % CB;
                sW[j * LDW + k] = (cb0 + j < a.cout) ? Wg[(size_t)k * a.cout + cb0 + j] : 0.f;
            }
        }
        if (a.pro_scale) {
            for (int e = tid; e < CIN; e += PW_NW * 64) {
                sPro[e] = a.pro_scale[(size_t)slot * CIN + e];
                sPro[CIN + e] = a.pro_shift[(size_t)slot * CIN + e];
            }
        }
    }
    __syncthreads();

    const bool has_pro = a.pro_scale != nullptr;
    const bool has_drop = a.drop_p > 0.f;
    const uint32_t dthr = drop_threshold(a.drop_p);
    const uint32_t dbase = a.drop_seed;
    const float dscale = has_drop ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    const bool do_stats = a.part_sum != nullptr;
    const bool do_pool = a.part_max != nullptr;

    // BatchNorm statistics are accumulated as sums of (v - z0) and (v - z0)^2 with z0 = the wave's first
    // row: E[z^2] - mean^2 in fp32 loses everything when a channel's spread is small against its mean
    // (the T-Net FC layers normalise over only B rows of near-identical pooled features).
    float s_sum[NT], s_sq[NT], s_max[NT], s_min[NT], s_z0[NT];
    int s_amax[NT], s_amin[NT];
    int s_cnt = 0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        s_z0[t] = 0.f;
        s_sum[t] = 0.f;
        s_sq[t] = 0.f;
        s_max[t] = -__builtin_inff();
        s_min[t] = __builtin_inff();
        s_amax[t] = -1;
        s_amin[t] = -1;
    }
    float bias_v[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = cb0 + 32 * t + r;
        bias_v[t] = (a.bias && col < a.cout) ? a.bias[(size_t)(a.bias_win_stride ? pidx : 0) * a.bias_win_stride + col] : 0.f;
    }

    const int ntiles = (nrows + 31) / 32;
    // A fragments: blocks of 4 j (= 32 k = one 128-byte line per row), the next block prefetched in registers
    // while the current one feeds 16 * NT MFMAs.  The prefetch runs across tile boundaries.
    f32x4 a_cur[4], a_nxt[4];

    auto frag_ptr = [&](int tile, int kb) -> const float * {
        const int row0 = row_begin + tile * 32;
        const int valid = min(32, row_end - row0);
        const int arow = row0 + min(r, valid - 1);
        return a.A + (size_t)arow * a.lda + 32 * kb + 4 * h;
    };

    int tile = wave;
    if (tile < ntiles) {
        const float *ap = frag_ptr(tile, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) a_cur[j] = *reinterpret_cast<const f32x4 *>(ap + 8 * j);
    }
    for (; tile < ntiles; tile += PW_NW) {
        const int row0 = row_begin + tile * 32;
        const int valid = min(32, row_end - row0);
        const int arow = row0 + min(r, valid - 1);
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

#pragma unroll 1
        for (int kb = 0; kb < NBLK; ++kb) {
            // prefetch the next block (same tile, or block 0 of this wave's next tile)
            {
                int ptile = tile, pkb = kb + 1;
                if (pkb == NBLK) {
                    ptile = tile + PW_NW;
                    pkb = 0;
                }
                if (ptile < ntiles) {
                    const float *ap = frag_ptr(ptile, pkb);
#pragma unroll
                    for (int j = 0; j < 4; ++j) a_nxt[j] = *reinterpret_cast<const f32x4 *>(ap + 8 * j);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k0 = 32 * kb + 8 * j + 4 * h;
                f32x4 av = a_cur[j];
                if (has_pro) {
                    const f32x4 sc = *reinterpret_cast<const f32x4 *>(sPro + k0);
                    const f32x4 sh = *reinterpret_cast<const f32x4 *>(sPro + CIN + k0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) av[i] = fmaxf(fmaf(av[i], sc[i], sh[i]), 0.f);
                    if (has_drop) {
                        const uint32_t e0 = (uint32_t)arow * (uint32_t)CIN + (uint32_t)k0;
#pragma unroll
                        for (int i = 0; i < 4; ++i) av[i] = (mix32((e0 + i) ^ dbase) >= dthr) ? av[i] * dscale : 0.f;
                    }
                }
                f32x4 bv[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) bv[t] = *reinterpret_cast<const f32x4 *>(sW + (32 * t + r) * LDW + k0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[t][i], acc[t], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) a_cur[j] = a_nxt[j];
        }

        // ---- epilogue: lane = output channel, registers = 16 rows ----
        if (do_stats) {
            if (tile == wave) {
#pragma unroll
                for (int t = 0; t < NT; ++t) s_z0[t] = __shfl(acc[t][0] + bias_v[t], r);   // row0 + 0 lives in lane r, register 0
            }
            s_cnt += valid;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int col = cb0 + 32 * t + r;
            const bool cok = col < a.cout;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rr = (e & 3) + 8 * (e >> 2) + 4 * h;
                const float v = acc[t][e] + bias_v[t];
                const bool ok = rr < valid;
                if (a.Z && ok && cok) a.Z[(size_t)(row0 + rr) * a.ldz + col] = v;
                if (do_stats && ok) {
                    const float d = v - s_z0[t];
                    s_sum[t] += d;
                    s_sq[t] = fmaf(d, d, s_sq[t]);
                }
                if (do_pool && ok) {
                    if (v > s_max[t]) {
                        s_max[t] = v;
                        s_amax[t] = row0 + rr;
                    }
                    if (v < s_min[t]) {
                        s_min[t] = v;
                        s_amin[t] = row0 + rr;
                    }
                }
            }
        }
    }

    if (!do_stats && !do_pool) return;

    // ---- combine the two half-waves, then the waves (LDS scratch reuses the weight tile) ----
    __syncthreads();
    float *red_f = smem;                                           // [PW_NW][CB][5]: S1, S2, max, min, z0
    int *red_i = reinterpret_cast<int *>(smem + PW_NW * CB * 5);   // [PW_NW][CB][2]: amax, amin
    int *red_n = red_i + PW_NW * CB * 2;                           // [PW_NW] rows seen by the wave
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float o_sum = __shfl_xor(s_sum[t], 32), o_sq = __shfl_xor(s_sq[t], 32);
        const float o_max = __shfl_xor(s_max[t], 32), o_min = __shfl_xor(s_min[t], 32);
        const int o_amax = __shfl_xor(s_amax[t], 32), o_amin = __shfl_xor(s_amin[t], 32);
        float f_max = s_max[t], f_min = s_min[t];
        int f_amax = s_amax[t], f_amin = s_amin[t];
        if (o_max > f_max || (o_max == f_max && (unsigned)o_amax < (unsigned)f_amax)) {
            f_max = o_max;
            f_amax = o_amax;
        }
        if (o_min < f_min || (o_min == f_min && (unsigned)o_amin < (unsigned)f_amin)) {
            f_min = o_min;
            f_amin = o_amin;
        }
        if (h == 0) {
            const int c = 32 * t + r;
            red_f[(wave * CB + c) * 5 + 0] = s_sum[t] + o_sum;
            red_f[(wave * CB + c) * 5 + 1] = s_sq[t] + o_sq;
            red_f[(wave * CB + c) * 5 + 2] = f_max;
            red_f[(wave * CB + c) * 5 + 3] = f_min;
            red_f[(wave * CB + c) * 5 + 4] = s_z0[t];
            red_i[(wave * CB + c) * 2 + 0] = f_amax;
            red_i[(wave * CB + c) * 2 + 1] = f_amin;
        }
    }
    if (lane == 0) red_n[wave] = s_cnt;
    __syncthreads();
    for (int c = tid; c < CB; c += PW_NW * 64) {
        const int col = cb0 + c;
        if (col >= a.cout) continue;
        double n = 0.0, mean = 0.0, m2 = 0.0;                      // Chan's pairwise merge, fixed wave order
        float mx = -__builtin_inff(), mn = __builtin_inff();
        int amx = -1, amn = -1;
#pragma unroll
        for (int w = 0; w < PW_NW; ++w) {
            const double nw = (double)red_n[w];
            if (do_stats && nw > 0.0) {
                const double s1 = red_f[(w * CB + c) * 5 + 0], s2 = red_f[(w * CB + c) * 5 + 1];
                const double mw = (double)red_f[(w * CB + c) * 5 + 4] + s1 / nw;
                const double m2w = s2 - s1 * s1 / nw;
                const double nn = n + nw, delta = mw - mean;
                mean += delta * nw / nn;
                m2 += m2w + delta * delta * n * nw / nn;
                n = nn;
            }
            const float vmx = red_f[(w * CB + c) * 5 + 2], vmn = red_f[(w * CB + c) * 5 + 3];
            const int imx = red_i[(w * CB + c) * 2 + 0], imn = red_i[(w * CB + c) * 2 + 1];
            if (vmx > mx || (vmx == mx && (unsigned)imx < (unsigned)amx)) {
                mx = vmx;
                amx = imx;
            }
            if (vmn < mn || (vmn == mn && (unsigned)imn < (unsigned)amn)) {
                mn = vmn;
                amn = imn;
            }
        }
        const size_t o = (size_t)(q * a.chunks + chunk) * a.cout + col;
        if (do_stats) {
            a.part_sum[o] = (float)mean;            // chunk mean
            a.part_sq[o] = (float)(m2 < 0.0 ? 0.0 : m2);   // chunk sum of squared deviations
        }
        if (do_pool) {
            a.part_max[o] = mx;
            a.part_min[o] = mn;
            a.part_amax[o] = amx;
            a.part_amin[o] = amn;
        }
    }
}

template <int CIN, int NT>
static int launch_pw(const PwGemm &a, hipStream_t st)
{
    constexpr int CB = 32 * NT;
    constexpr size_t lds_main = (size_t)(CB * (CIN + 4) + 2 * CIN) * sizeof(float);
    constexpr size_t lds_red = (size_t)(PW_NW * CB * 7 + PW_NW) * sizeof(float);
    constexpr size_t lds = lds_main > lds_red ? lds_main : lds_red;
    static bool attr_set = false;
    auto kern = pw_gemm_kernel<CIN, NT>;
    if (!attr_set) {
        if (lds > 65536) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return fail(AMPNET_E_LAUNCH, "pw_gemm: hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
        }
        attr_set = true;
    }
    dim3 grid(a.chunks, a.Q, cdiv(a.cout, CB));
    char name[64];
    snprintf(name, sizeof(name), "pw_gemm<%d,%d>%s%s", CIN, 32 * NT, a.Z ? "+store" : "", a.part_max ? "+pool" : "");
    const double rows = (double)a.rows_hint;
    ProfScope prof(name, 2.0 * rows * CIN * a.cout, rows * 4.0 * ((double)CIN * cdiv(a.cout, CB) + (a.Z ? a.cout : 0)), st);
    hipLaunchKernelGGL(kern, grid, dim3(PW_NW * 64), lds, st, a);
    return check_launch("pw_gemm_kernel");
}

int pw_gemm(const PwGemm &a, hipStream_t st)
{
    AMPNET_REQUIRE(a.A && a.W && a.win_off, "pw_gemm: null pointer");
    AMPNET_REQUIRE(a.Q >= 1 && a.chunks >= 1 && a.cout >= 1, "pw_gemm: bad sizes Q=%d chunks=%d cout=%d", a.Q, a.chunks, a.cout);
    AMPNET_REQUIRE(a.lda % 4 == 0 && (a.w_win_stride != 0 || a.ldw % 4 == 0), "pw_gemm: lda/ldw must be multiples of 4");
    AMPNET_REQUIRE(a.n_slots >= 1 && (!a.perwin_slot_major || a.Q % a.n_slots == 0), "pw_gemm: Q %% n_slots != 0");
    AMPNET_REQUIRE((a.part_sum == nullptr) == (a.part_sq == nullptr), "pw_gemm: part_sum/part_sq must come together");
    AMPNET_REQUIRE(!a.part_max || (a.part_min && a.part_amax && a.part_amin), "pw_gemm: pool partials incomplete");
    const int nt = a.cout > 64 ? 4 : (a.cout > 32 ? 2 : 1);
    switch (a.cin) {
    case 64:
        if (nt == 4) return launch_pw<64, 4>(a, st);
        if (nt == 2) return launch_pw<64, 2>(a, st);
        return launch_pw<64, 1>(a, st);
    case 128:
        if (nt == 4) return launch_pw<128, 4>(a, st);
        if (nt == 2) return launch_pw<128, 2>(a, st);
        return launch_pw<128, 1>(a, st);
    case 256:
        if (nt == 4) return launch_pw<256, 4>(a, st);
        if (nt == 2) return launch_pw<256, 2>(a, st);
        return launch_pw<256, 1>(a, st);
    default:
        return fail(AMPNET_E_ARG, "pw_gemm: cin=%d not in {64,128,256}", a.cin);
    }
}

}  // namespace ampnet
