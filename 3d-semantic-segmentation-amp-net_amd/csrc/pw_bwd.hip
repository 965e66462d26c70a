// pw_bwd.hip -- backward of the shared per-point MLP layers on the fp32 matrix cores.
//
// For a layer z_l = a_{l-1} W_l^T (+ b_l), a_{l-1} = dropout(relu(bn_{l-1}(z_{l-1}))), autograd of the reference
// (train_pointnet-attention.py:467 loss.backward()) computes, per layer,
//     g      = dL/dz_l  = BatchNorm-backward(dy_l)           -> folded here into g = dy*P1 + z*P2 + P3 per channel
//     dW_l   = g^T a_{l-1},  db_l = sum g                    -> pw_wgrad (per-window partials + ordered reduction)
//     dy_{l-1} = (g W_l) * [y_{l-1} > 0] * dropout mask       -> pw_dgrad, which also emits the partial sums
//                sum dy_{l-1}, sum dy_{l-1} * zhat_{l-1} that the NEXT BatchNorm backward needs.
// Nothing but z (saved by the forward) and dy (one buffer per layer) is read from HBM: activations, masks and
// normalised values are recomputed in registers.  MaxPool backward never materialises its sparse gradient:
// dy[row][c] = (row == argmax[window][c]) ? dpool[window][c] : 0 is formed while loading.
//
// MFMA mapping: same as pw_gemm.hip (A = rows x K from HBM in 16-byte fragments, B = weights from LDS,
// accumulator = output column on the lane); the weight gradient contracts over ROWS, so there both operands
// are staged row-major in LDS (already transformed) and read with conflict-free ds_read_b32.
#include "kernels.h"

namespace ampnet {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int BW_NW = 4;

__device__ __forceinline__ int pidx_of_b(int q, int n_slots, int Q, int slot_major)
{
    return slot_major ? (q % n_slots) * (Q / n_slots) + q / n_slots : q;
}

// ----------------------------------------------------------------------------------------------------
// pw_dgrad
// ----------------------------------------------------------------------------------------------------
template <int K, int NT, int NW, bool EXTRA>
__global__ __launch_bounds__(NW * 64, 2) void pw_dgrad_kernel(PwDgrad a)
{
    constexpr int CB = 32 * NT;
    constexpr int LDW = K + 4;
    constexpr int NBLK = K / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sW = smem;                       // [CB][LDW]: sW[j][k] = W[k][cb0 + j]
    float *sP = smem + CB * LDW;            // P1[K], P2[K], P3[K]
    float *sDp = sP + 3 * K;                // sparse: dpool[K]
    int *sArg = reinterpret_cast<int *>(sDp + K);   // sparse: arg[K]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int q = blockIdx.y, chunk = blockIdx.x, cb0 = blockIdx.z * CB;
    const int row_begin = a.win_off[q] + chunk * a.chunk_rows;
    const int row_end = min(a.win_off[q + 1], row_begin + a.chunk_rows);
    const int nrows = max(row_end - row_begin, 0);
    const int slot = (a.n_slots > 1) ? (q % a.n_slots) : 0;
    const int pidx = pidx_of_b(q, a.n_slots, a.Q, a.perwin_slot_major);
    const bool act = a.g.act != 0;                       // operand = relu(z * P2 + P3)
    const bool sparse = a.g.dy == nullptr && !act;
    const bool has_bn = a.g.P1 != nullptr || act;        // z is loaded

    if (nrows > 0) {
        if (a.w_win_stride == 0) {
            const float *Wsh = a.W + (size_t)slot * a.w_slot_stride;
            // shared weight W[k][j] (torch [cout_l = K][cin_l]): transpose while staging
            // consecutive lanes take consecutive k: the LDS writes are conflict-free, the 16-byte global reads are
            // strided but L2-resident (the whole weight is <= 128 KB and every workgroup reads it)
            for (int e = tid; e < K * (CB / 4); e += NW * 64) {
                const int k = e % K, j4 = e / K;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (cb0 + 4 * j4 < a.cp) v = *reinterpret_cast<const f32x4 *>(Wsh + (size_t)k * a.ldw + cb0 + 4 * j4);
#pragma unroll
                for (int i = 0; i < 4; ++i) sW[(4 * j4 + i) * LDW + k] = v[i];
            }
        } else {
            const float *Wg = a.W + (size_t)pidx * a.w_win_stride;   // T[j][k], k contiguous, K columns
            for (int e = tid; e < CB * (K / 4); e += NW * 64) {
                const int j = e / (K / 4), k4 = e % (K / 4);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (cb0 + j < a.cp) v = *reinterpret_cast<const f32x4 *>(Wg + (size_t)(cb0 + j) * K + 4 * k4);
                *reinterpret_cast<f32x4 *>(sW + j * LDW + 4 * k4) = v;
            }
        }
        for (int e = tid; e < K; e += NW * 64) {
            sP[e] = a.g.P1 ? a.g.P1[(size_t)slot * K + e] : 1.0f;
            sP[K + e] = has_bn ? a.g.P2[(size_t)slot * K + e] : 0.0f;
            sP[2 * K + e] = has_bn ? a.g.P3[(size_t)slot * K + e] : 0.0f;
            if (sparse) {
                const int prow = pidx_of_b(q, a.n_slots, a.Q, a.g.dpool_slot_major);
                sDp[e] = a.g.dpool[(size_t)prow * K + e];
                sArg[e] = a.g.arg[(size_t)q * K + e];
            }
        }
    }
    __syncthreads();

    const bool has_prev = a.prev.z != nullptr;
    const bool has_drop = a.prev.drop_p > 0.f;
    const uint32_t dthr = drop_threshold(a.prev.drop_p);
    const float dscale = has_drop ? 1.0f / (1.0f - a.prev.drop_p) : 1.0f;
    const bool do_part = a.part_a != nullptr;

    float c_s[NT], c_t[NT], c_m[NT], c_i[NT], c_b[NT], s_a[NT], s_b[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = cb0 + 32 * t + r;
        c_b[t] = (a.bias_slot && col < a.cp) ? a.bias_slot[(size_t)slot * a.cp + col] : 0.f;
        const bool ok = has_prev && col < a.cp && a.prev.s != nullptr;
        c_s[t] = ok ? a.prev.s[(size_t)slot * a.cp + col] : 1.0f;
        c_t[t] = ok ? a.prev.t[(size_t)slot * a.cp + col] : 0.0f;
        c_m[t] = (ok && a.prev_mean) ? a.prev_mean[(size_t)slot * a.cp + col] : 0.0f;
        c_i[t] = (ok && a.prev_invstd) ? a.prev_invstd[(size_t)slot * a.cp + col] : 0.0f;
        s_a[t] = 0.f;
        s_b[t] = 0.f;
    }

    const int ntiles = (nrows + 31) / 32;
    f32x4 dy_cur[4], dy_nxt[4], z_cur[4], z_nxt[4];
    auto arow_of = [&](int tile) {
        const int row0 = row_begin + tile * 32;
        const int valid = min(32, row_end - row0);
        return row0 + min(r, valid - 1);
    };
    auto load_blk = [&](int tile, int kb, f32x4 (&dy)[4], f32x4 (&z)[4]) {
        const size_t o = (size_t)arow_of(tile) * K + 32 * kb + 4 * h;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!sparse && !act) dy[j] = *reinterpret_cast<const f32x4 *>(a.g.dy + o + 8 * j);
            if (has_bn) z[j] = *reinterpret_cast<const f32x4 *>(a.g.z + o + 8 * j);
        }
    };

    int tile = wave;
    if (tile < ntiles) load_blk(tile, 0, dy_cur, z_cur);
    for (; tile < ntiles; tile += NW) {
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
            {
                int ptile = tile, pkb = kb + 1;
                if (pkb == NBLK) {
                    ptile = tile + NW;
                    pkb = 0;
                }
                if (ptile < ntiles) load_blk(ptile, pkb, dy_nxt, z_nxt);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k0 = 32 * kb + 8 * j + 4 * h;
                f32x4 dyv;
                if (sparse) {
                    const i32x4 ar = *reinterpret_cast<const i32x4 *>(sArg + k0);
                    const f32x4 dp = *reinterpret_cast<const f32x4 *>(sDp + k0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) dyv[i] = (ar[i] == arow && r < valid) ? dp[i] : 0.f;
                } else if (!act) {
                    dyv = dy_cur[j];
                }
                f32x4 gv;
                if (act) {
                    const f32x4 p2 = *reinterpret_cast<const f32x4 *>(sP + K + k0);
                    const f32x4 p3 = *reinterpret_cast<const f32x4 *>(sP + 2 * K + k0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) gv[i] = fmaxf(fmaf(z_cur[j][i], p2[i], p3[i]), 0.f);
                } else if (has_bn) {
                    const f32x4 p1 = *reinterpret_cast<const f32x4 *>(sP + k0);
                    const f32x4 p2 = *reinterpret_cast<const f32x4 *>(sP + K + k0);
                    const f32x4 p3 = *reinterpret_cast<const f32x4 *>(sP + 2 * K + k0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) gv[i] = fmaf(dyv[i], p1[i], fmaf(z_cur[j][i], p2[i], p3[i]));
                } else {
                    gv = dyv;
                }
                f32x4 bv[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) bv[t] = *reinterpret_cast<const f32x4 *>(sW + (32 * t + r) * LDW + k0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(gv[i], bv[t][i], acc[t], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dy_cur[j] = dy_nxt[j];
                z_cur[j] = z_nxt[j];
            }
        }

        // ---- epilogue: lane = column of layer l-1, registers = rows.  All global loads of a 32 x 32 sub-tile are
        // issued back to back BEFORE anything consumes them (a load-use-load-use chain costs one memory round trip
        // per element: 64 per tile, ten times the tile's MFMA time).
        int rmap[16];
        if (EXTRA && a.rowmap) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rr = (e & 3) + 8 * (e >> 2) + 4 * h;
                rmap[e] = rr < valid ? a.rowmap[row0 + rr] : -1;
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int col = cb0 + 32 * t + r;
            const bool cok = col < a.cp;
            const int ccol = cok ? col : 0;
            float zv[16], xv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rr = (e & 3) + 8 * (e >> 2) + 4 * h;
                const size_t o = (size_t)(row0 + min(rr, valid - 1)) * a.cp + ccol;      // clamped: always a valid address
                zv[e] = has_prev ? a.prev.z[o] : 0.f;
                if (EXTRA) {
                    float x = a.add ? a.add[o] : 0.f;
                    if (a.rowmap && rmap[e] >= 0) x += a.srows[(size_t)rmap[e] * a.cp + ccol];
                    xv[e] = x;
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rr = (e & 3) + 8 * (e >> 2) + 4 * h;
                if (!(rr < valid && cok)) continue;
                const size_t o = (size_t)(row0 + rr) * a.cp + col;
                float v = acc[t][e] + c_b[t];
                if (EXTRA) v += xv[e];
                if (has_prev) {
                    bool keep = fmaf(zv[e], c_s[t], c_t[t]) > 0.f;
                    if (a.prev.s == nullptr) keep = true;            // identity activation: no mask
                    if (has_drop) {
                        const bool dk = mix32(((uint32_t)(row0 + rr) * (uint32_t)a.cp + (uint32_t)col) ^ a.prev.drop_seed) >= dthr;
                        v = dk ? v * dscale : 0.f;
                    }
                    v = keep ? v : 0.f;
                    if (do_part) {
                        s_a[t] += v;
                        s_b[t] = fmaf(v, (zv[e] - c_m[t]) * c_i[t], s_b[t]);
                    }
                }
                a.out[o] = v;
            }
        }
    }

    if (!do_part) return;
    __syncthreads();
    float *red = smem;     // [NW][CB][2]
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float oa = __shfl_xor(s_a[t], 32), ob = __shfl_xor(s_b[t], 32);
        if (h == 0) {
            red[(wave * CB + 32 * t + r) * 2 + 0] = s_a[t] + oa;
            red[(wave * CB + 32 * t + r) * 2 + 1] = s_b[t] + ob;
        }
    }
    __syncthreads();
    for (int c = tid; c < CB; c += NW * 64) {
        const int col = cb0 + c;
        if (col >= a.cp) continue;
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            sa += red[(w * CB + c) * 2 + 0];
            sb += red[(w * CB + c) * 2 + 1];
        }
        const size_t o = (size_t)(q * (a.part_chunks ? a.part_chunks : a.chunks) + chunk) * a.cp + col;
        a.part_a[o] = sa;
        a.part_b[o] = sb;
    }
}

template <int K, int NT, int NW, bool EXTRA>
static int launch_dgrad_x(const PwDgrad &a, hipStream_t st)
{
    constexpr int CB = 32 * NT;
    constexpr size_t lds_main = (size_t)(CB * (K + 4) + 5 * K) * sizeof(float);
    constexpr size_t lds_red = (size_t)NW * CB * 2 * sizeof(float);
    constexpr size_t lds = lds_main > lds_red ? lds_main : lds_red;
    static bool attr_set = false;
    auto kern = pw_dgrad_kernel<K, NT, NW, EXTRA>;
    if (!attr_set) {
        if (lds > 65536) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return fail(AMPNET_E_LAUNCH, "pw_dgrad: hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
        }
        attr_set = true;
    }
    char name[64];
    snprintf(name, sizeof(name), "pw_dgrad<%d,%d>%s", K, CB, a.g.act ? "+act" : (a.g.dy ? "" : "+sparse"));
    const double rows = (double)a.rows_hint;
    ProfScope prof(name, 2.0 * rows * K * a.cp, rows * 4.0 * ((a.g.dy ? K : 0) + (a.g.P1 ? K : 0) + (a.prev.z ? 2.0 : 1.0) * a.cp), st);
    hipLaunchKernelGGL(kern, dim3(a.chunks, a.Q, cdiv(a.cp, CB)), dim3(NW * 64), lds, st, a);
    return check_launch("pw_dgrad_kernel");
}

template <int K, int NT, int NW>
static int launch_dgrad(const PwDgrad &a, hipStream_t st)
{
    return (a.add || a.rowmap) ? launch_dgrad_x<K, NT, NW, true>(a, st) : launch_dgrad_x<K, NT, NW, false>(a, st);
}

int pw_dgrad(const PwDgrad &a, hipStream_t st)
{
    AMPNET_REQUIRE(a.W && a.out && a.win_off, "pw_dgrad: null pointer");
    AMPNET_REQUIRE(a.g.dy || a.g.act || (a.g.arg && a.g.dpool), "pw_dgrad: neither dense nor sparse gradient source");
    AMPNET_REQUIRE(!(a.g.P1 || a.g.act) || (a.g.P2 && a.g.P3 && a.g.z), "pw_dgrad: BatchNorm constants incomplete");
    AMPNET_REQUIRE(!a.rowmap || a.srows, "pw_dgrad: rowmap without srows");
    AMPNET_REQUIRE(a.cp >= 1 && a.cp % 4 == 0 && (a.w_win_stride != 0 || a.ldw % 4 == 0), "pw_dgrad: cp / ldw must be multiples of 4");
    AMPNET_REQUIRE(!a.part_a || (a.part_b && a.prev.z), "pw_dgrad: partial sums need the previous layer");
    const int nt = a.cp > 64 ? 4 : (a.cp > 32 ? 2 : 1);
    switch (a.g.C) {
    case 64:
        return nt == 4 ? launch_dgrad<64, 4, 4>(a, st) : (nt == 2 ? launch_dgrad<64, 2, 4>(a, st) : launch_dgrad<64, 1, 4>(a, st));
    case 128:
        return nt == 4 ? launch_dgrad<128, 4, 4>(a, st) : (nt == 2 ? launch_dgrad<128, 2, 4>(a, st) : launch_dgrad<128, 1, 4>(a, st));
    case 256:
        // 128 x 260 floats of weights = 133 KB of LDS = one workgroup per CU: give that workgroup 8 waves
        return nt == 4 ? launch_dgrad<256, 4, 8>(a, st) : (nt == 2 ? launch_dgrad<256, 2, 4>(a, st) : launch_dgrad<256, 1, 4>(a, st));
    default:
        return fail(AMPNET_E_ARG, "pw_dgrad: K=%d not in {64,128,256}", a.g.C);
    }
}

// ----------------------------------------------------------------------------------------------------
// pw_wgrad: one workgroup = one chunk of one window x one (32*TX) x (32*TY) block of dW.  The block is split over
// the 4 waves by TILES (never by rows), so every wave issues MFMAs whatever the layer shape:
//   (TX,TY) = (2,2): 1 tile per wave    (4,2): wave w -> row-tile w, both column tiles    (2,4): transposed
//   (4,4): 2 x 2 tiles per wave.
// Operands are transformed once while staged into LDS (double buffered, one barrier per ROWS rows).
// ----------------------------------------------------------------------------------------------------
template <int TX, int TY, int ROWS>
__global__ __launch_bounds__(256, 2) void pw_wgrad_kernel(PwWgrad a)
{
    constexpr int CXB = 32 * TX, CYB = 32 * TY;
    constexpr int WX = (TX == 4 && TY == 2) ? 4 : ((TX == 2 && TY == 4) ? 1 : 2);
    constexpr int WY = 4 / WX;
    constexpr int TXW = TX / WX, TYW = TY / WY;
    constexpr int QX = CXB / 4, QY = CYB / 4;            // float4 columns of a staged row
    constexpr int SX = 256 / QX, SY = 256 / QY;          // rows covered per staging iteration
    constexpr int NIX = ROWS / SX, NIY = ROWS / SY;
    __shared__ __attribute__((aligned(16))) float sX[2][ROWS][CXB];
    __shared__ __attribute__((aligned(16))) float sY[2][ROWS][CYB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int q = blockIdx.x / a.chunks, chunk = blockIdx.x % a.chunks;
    const int cx0 = blockIdx.y * CXB, cy0 = blockIdx.z * CYB;
    const int row_begin = a.win_off[q] + chunk * a.chunk_rows;
    const int row_end = min(a.win_off[q + 1], row_begin + a.chunk_rows);
    const int slot = (a.n_slots > 1) ? (q % a.n_slots) : 0;
    const int CX = a.x.C, CY = a.y.C;
    const bool x_act = a.x.act != 0;
    const bool sparse = a.x.dy == nullptr && !x_act;
    const bool has_bn = a.x.P1 != nullptr && !x_act;
    const bool y_act = a.y.s != nullptr;
    const bool y_drop = a.y.drop_p > 0.f;
    const uint32_t dthr = drop_threshold(a.y.drop_p);
    const float dscale = y_drop ? 1.0f / (1.0f - a.y.drop_p) : 1.0f;

    // staging roles
    const int cqx = tid % QX, rsx = tid / QX, cqy = tid % QY, rsy = tid / QY;
    const int xc = cx0 + 4 * cqx, yc = cy0 + 4 * cqy;
    const bool xok = xc < CX, yok = yc < CY;
    f32x4 p1 = {1.f, 1.f, 1.f, 1.f}, p2 = {0.f, 0.f, 0.f, 0.f}, p3 = {0.f, 0.f, 0.f, 0.f}, ys = p1, yt = p2, dp = p2;
    i32x4 ar = {-1, -1, -1, -1};
    if (xok) {
        if (has_bn) p1 = *reinterpret_cast<const f32x4 *>(a.x.P1 + (size_t)slot * CX + xc);
        if (has_bn || x_act) {
            p2 = *reinterpret_cast<const f32x4 *>(a.x.P2 + (size_t)slot * CX + xc);
            p3 = *reinterpret_cast<const f32x4 *>(a.x.P3 + (size_t)slot * CX + xc);
        }
        if (sparse) {
            const int prow = pidx_of_b(q, a.n_slots, a.Q, a.x.dpool_slot_major);
            dp = *reinterpret_cast<const f32x4 *>(a.x.dpool + (size_t)prow * CX + xc);
            ar = *reinterpret_cast<const i32x4 *>(a.x.arg + (size_t)q * CX + xc);
        }
    }
    if (yok && y_act) {
        ys = *reinterpret_cast<const f32x4 *>(a.y.s + (size_t)slot * CY + yc);
        yt = *reinterpret_cast<const f32x4 *>(a.y.t + (size_t)slot * CY + yc);
    }

    f32x4 rx_dy[NIX], rx_z[NIX], ry_z[NIY];
    auto load_regs = [&](int blk_row0) {
#pragma unroll
        for (int i = 0; i < NIX; ++i) {
            const int row = blk_row0 + rsx + SX * i;
            const size_t rr = (size_t)(row < row_end ? row : row_begin);
            if (xok) {
                if (!sparse && !x_act) rx_dy[i] = *reinterpret_cast<const f32x4 *>(a.x.dy + rr * CX + xc);
                if (has_bn || x_act) rx_z[i] = *reinterpret_cast<const f32x4 *>(a.x.z + rr * CX + xc);
            }
        }
#pragma unroll
        for (int i = 0; i < NIY; ++i) {
            const int row = blk_row0 + rsy + SY * i;
            const size_t rr = (size_t)(row < row_end ? row : row_begin);
            if (yok) ry_z[i] = *reinterpret_cast<const f32x4 *>(a.y.z + rr * CY + yc);
        }
    };
    f32x4 dbacc = {0.f, 0.f, 0.f, 0.f};
    auto write_lds = [&](int buf, int blk_row0) {
#pragma unroll
        for (int i = 0; i < NIX; ++i) {
            const int row = blk_row0 + rsx + SX * i;
            f32x4 xv = {0.f, 0.f, 0.f, 0.f};
            if (row < row_end && xok) {
                f32x4 dyv = {0.f, 0.f, 0.f, 0.f};
                if (sparse) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) dyv[c] = (ar[c] == row) ? dp[c] : 0.f;
                } else if (!x_act) {
                    dyv = rx_dy[i];
                }
                if (x_act) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) xv[c] = fmaxf(fmaf(rx_z[i][c], p2[c], p3[c]), 0.f);
                } else if (has_bn) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) xv[c] = fmaf(dyv[c], p1[c], fmaf(rx_z[i][c], p2[c], p3[c]));
                } else {
                    xv = dyv;
                }
                dbacc += xv;
            }
            *reinterpret_cast<f32x4 *>(&sX[buf][rsx + SX * i][4 * cqx]) = xv;
        }
#pragma unroll
        for (int i = 0; i < NIY; ++i) {
            const int row = blk_row0 + rsy + SY * i;
            f32x4 yv = {0.f, 0.f, 0.f, 0.f};
            if (row < row_end && yok) {
                yv = ry_z[i];
                if (y_act) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) yv[c] = fmaxf(fmaf(yv[c], ys[c], yt[c]), 0.f);
                    if (y_drop) {
                        const uint32_t e0 = (uint32_t)row * (uint32_t)CY + (uint32_t)yc;
#pragma unroll
                        for (int c = 0; c < 4; ++c) yv[c] = (mix32((e0 + c) ^ a.y.drop_seed) >= dthr) ? yv[c] * dscale : 0.f;
                    }
                }
            }
            *reinterpret_cast<f32x4 *>(&sY[buf][rsy + SY * i][4 * cqy]) = yv;
        }
    };

    const int tx0 = (wave / WY) * TXW, ty0 = (wave % WY) * TYW;
    f32x16 acc[TXW][TYW];
#pragma unroll
    for (int i = 0; i < TXW; ++i)
#pragma unroll
        for (int j = 0; j < TYW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nblk = (max(row_end - row_begin, 0) + ROWS - 1) / ROWS;
    if (nblk > 0) {
        load_regs(row_begin);
        write_lds(0, row_begin);
    }
    __syncthreads();
    for (int b = 0; b < nblk; ++b) {
        const int cur = b & 1;
        const bool more = b + 1 < nblk;
        if (more) load_regs(row_begin + (b + 1) * ROWS);
#pragma unroll 4
        for (int s2 = 0; s2 < ROWS / 2; ++s2) {
            const int kr = 2 * s2 + h;
            float xa[TXW], yb[TYW];
#pragma unroll
            for (int i = 0; i < TXW; ++i) xa[i] = sX[cur][kr][32 * (tx0 + i) + r];
#pragma unroll
            for (int j = 0; j < TYW; ++j) yb[j] = sY[cur][kr][32 * (ty0 + j) + r];
#pragma unroll
            for (int i = 0; i < TXW; ++i)
#pragma unroll
                for (int j = 0; j < TYW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[i], yb[j], acc[i][j], 0, 0, 0);
        }
        if (more) write_lds(cur ^ 1, row_begin + (b + 1) * ROWS);
        __syncthreads();
    }

    // ---- the chunk's partial: accumulator row = cx (registers), column = cy (lane) ----
    const size_t pbase = (size_t)blockIdx.x * CX;
#pragma unroll
    for (int i = 0; i < TXW; ++i)
#pragma unroll
        for (int j = 0; j < TYW; ++j) {
            const int cy = cy0 + 32 * (ty0 + j) + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int cx = cx0 + 32 * (tx0 + i) + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (cx < CX && cy < CY) a.dWpart[(pbase + cx) * a.ldp + cy] = acc[i][j][e];
            }
        }
    if (a.dbpart && blockIdx.z == 0) {
        float *red = &sX[0][0][0];        // [SX][CXB]
        __syncthreads();
        *reinterpret_cast<f32x4 *>(red + rsx * CXB + 4 * cqx) = dbacc;
        __syncthreads();
        if (tid < CXB && cx0 + tid < CX) {
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < SX; ++g) s += red[g * CXB + tid];
            a.dbpart[pbase + cx0 + tid] = s;
        }
    }
}

int pw_wgrad(const PwWgrad &a, hipStream_t st)
{
    AMPNET_REQUIRE(a.dWpart && a.win_off && a.y.z, "pw_wgrad: null pointer");
    AMPNET_REQUIRE(a.x.dy || a.x.act || (a.x.arg && a.x.dpool), "pw_wgrad: neither dense nor sparse gradient source");
    AMPNET_REQUIRE(!(a.x.P1 || a.x.act) || (a.x.P2 && a.x.P3 && a.x.z), "pw_wgrad: BatchNorm constants incomplete");
    AMPNET_REQUIRE(a.x.C % 4 == 0 && a.y.C % 4 == 0 && a.ldp >= a.y.C, "pw_wgrad: channel counts must be multiples of 4");
    AMPNET_REQUIRE(a.chunks >= 1 && a.chunk_rows % 64 == 0, "pw_wgrad: chunk_rows must be a multiple of 64");
    char name[64];
    snprintf(name, sizeof(name), "pw_wgrad<%d,%d>%s", a.x.C, a.y.C, a.x.act ? "+gram" : (a.x.dy ? "" : "+sparse"));
    const double rows = (double)a.rows_hint;
    const int tx = a.x.C > 64 ? 4 : 2, ty = a.y.C > 64 ? 4 : 2;
    ProfScope prof(name, 2.0 * rows * a.x.C * a.y.C,
                   rows * 4.0 * (((a.x.dy ? 1 : 0) + (a.x.P1 ? 1 : 0)) * (double)a.x.C * cdiv(a.y.C, 32 * ty) + (double)a.y.C * cdiv(a.x.C, 32 * tx)), st);
    dim3 grid(a.Q * a.chunks, cdiv(a.x.C, 32 * tx), cdiv(a.y.C, 32 * ty));
    if (tx == 2 && ty == 2) hipLaunchKernelGGL((pw_wgrad_kernel<2, 2, 64>), grid, dim3(256), 0, st, a);
    else if (tx == 4 && ty == 2) hipLaunchKernelGGL((pw_wgrad_kernel<4, 2, 32>), grid, dim3(256), 0, st, a);
    else if (tx == 2 && ty == 4) hipLaunchKernelGGL((pw_wgrad_kernel<2, 4, 32>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((pw_wgrad_kernel<4, 4, 32>), grid, dim3(256), 0, st, a);
    return check_launch("pw_wgrad_kernel");
}

// ----------------------------------------------------------------------------------------------------
// reduce_windows: dst[r][c] (=|+=) sum_q part[q][r][c]; block = 32 elements x 8 partial groups, fixed order
// ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void reduce_windows_kernel(const float *__restrict__ part, int Q, long stride, int rows, int cols,
                                                            int ld_part, float *__restrict__ dst, int ld_dst, int accumulate)
{
    __shared__ float red[8][32];
    const int el = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + el;
    const bool ok = i < rows * cols;
    const int rr = ok ? i / cols : 0, c = ok ? i % cols : 0;
    const float *p = part + (size_t)rr * ld_part + c;
    float s0 = 0.f, s1 = 0.f;
    if (ok) {
        int qi = g;
        for (; qi + 56 < Q; qi += 64) {             // 8 independent loads per trip
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(qi + 8 * u) * stride];
            s0 += (v[0] + v[2]) + (v[4] + v[6]);
            s1 += (v[1] + v[3]) + (v[5] + v[7]);
        }
        for (; qi < Q; qi += 8) s0 += p[(size_t)qi * stride];
    }
    red[g][el] = s0 + s1;
    __syncthreads();
    if (g == 0 && ok) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += red[k][el];
        float *d = dst + (size_t)rr * ld_dst + c;
        *d = accumulate ? *d + s : s;
    }
}

// several reductions in one launch (the weight-gradient partials of the fused layer backward, deferred to the end of a backward pass)
struct ReduceMultiArgs {
    ReduceItem it[REDUCE_MULTI_MAX];
    int first_block[REDUCE_MULTI_MAX + 1];
    int n;
};

__global__ __launch_bounds__(256) void reduce_windows_multi_kernel(ReduceMultiArgs a)
{
    __shared__ float red[8][32];
    int k = 0;
    while (k + 1 < a.n && (int)blockIdx.x >= a.first_block[k + 1]) ++k;
    const ReduceItem it = a.it[k];
    const int el = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int i = ((int)blockIdx.x - a.first_block[k]) * 32 + el;
    const bool ok = i < it.rows * it.cols;
    const int rr = ok ? i / it.cols : 0, c = ok ? i % it.cols : 0;
    const float *p = it.part + (size_t)rr * it.ld_part + c;
    float s0 = 0.f, s1 = 0.f;
    if (ok) {
        int qi = g;
        for (; qi + 56 < it.Q; qi += 64) {             // 8 independent loads per trip, the order of reduce_windows_kernel
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(qi + 8 * u) * it.stride];
            s0 += (v[0] + v[2]) + (v[4] + v[6]);
            s1 += (v[1] + v[3]) + (v[5] + v[7]);
        }
        for (; qi < it.Q; qi += 8) s0 += p[(size_t)qi * it.stride];
    }
    red[g][el] = s0 + s1;
    __syncthreads();
    if (g == 0 && ok) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += red[j][el];
        it.dst[(size_t)rr * it.ld_dst + c] = s;
    }
}

int reduce_windows_multi(const ReduceItem *items, int n, hipStream_t st)
{
    AMPNET_REQUIRE(n >= 1 && n <= REDUCE_MULTI_MAX, "reduce_windows_multi: %d items", n);
    ReduceMultiArgs a;
    a.n = n;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        AMPNET_REQUIRE(items[i].part && items[i].dst && items[i].Q >= 1 && items[i].rows >= 1 && items[i].cols >= 1, "reduce_windows_multi: bad item %d", i);
        a.it[i] = items[i];
        a.first_block[i] = blocks;
        blocks += cdiv(items[i].rows * items[i].cols, 32);
    }
    a.first_block[n] = blocks;
    hipLaunchKernelGGL(reduce_windows_multi_kernel, dim3(blocks), dim3(256), 0, st, a);
    return check_launch("reduce_windows_multi_kernel");
}

int reduce_windows(const float *part, int Q, long stride, int rows, int cols, int ld_part, float *dst, int ld_dst, int accumulate,
                   hipStream_t st)
{
    AMPNET_REQUIRE(part && dst && Q >= 1 && rows >= 1 && cols >= 1, "reduce_windows: bad arguments");
    hipLaunchKernelGGL(reduce_windows_kernel, dim3(cdiv(rows * cols, 32)), dim3(256), 0, st, part, Q, stride, rows, cols, ld_part, dst,
                       ld_dst, accumulate);
    return check_launch("reduce_windows_kernel");
}

}  // namespace ampnet
