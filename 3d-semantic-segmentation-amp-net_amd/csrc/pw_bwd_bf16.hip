// pw_bwd_bf16.hip -- the fused backward of pw_bwd_fused.hip with bf16 MFMA operands (fp32 accumulation).
//
// Same launch interface (PwBwd), same work split, same roles, same fp32 tensors in HBM and the same fp32 epilogue as the fp32
// kernel; what changes is the arithmetic of the two products and the LDS image that feeds it:
//     sG[row][cx] = bf16(g)            g = dy * P1 + z * P2 + P3 in fp32 (or relu(z * P2 + P3) for the Gram form), rounded once
//     sY[row][cy] = bf16(act(z_prev))  the layer's input as the forward pass saw it (BatchNorm + ReLU (+ dropout) in fp32)
//     sZ[row][cy] = z_prev in fp32     only for the epilogue of the data gradient (ReLU mask, zhat of the BatchNorm-backward sums)
//     sWt[cy][cx] = bf16(W[cx][cy])
// W waves:  dW[cx][cy] += sum_rows g[row][cx] * y[row][cy]  on v_mfma_f32_32x32x16_bf16 with k = rows.  Both operands are COLUMNS of a
//           row-major tile; ds_read_b64_tr_b16 delivers them (a 4-row x 16-column block per 16 lanes, column-major to the lanes):
//           lane (r, h) of the operand gets rows 16 s + 8 h .. + 7 of channel 32 t + r from two such reads.
// D waves:  dy_prev[row][cy] = sum_cx g[row][cx] * W[cx][cy]  with k = cx: rows of sG and of sWt, one ds_read_b128 per operand and step.
// With the products 16 x cheaper than on the fp32 matrix path every shape of this kernel is bound by its HBM traffic
// (bytes: kernels.h / DESIGN.md); row stride of the bf16 tiles = 2 C + 64 bytes, which puts the four rows of a transposed
// read on four disjoint groups of 16 banks.
// Selected by ampnet_set_matrix_precision(AMPNET_PRECISION_BF16_TRAIN); BASELINE.json config 3 ("bf16 MFMA MLP").
#include <type_traits>
#include "kernels.h"

namespace ampnet {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int FBB_THREADS = 512;
constexpr int FBB_ITEM_ROWS = 256;      // must equal pw_bwd_item_rows() (the host sizes per-window shares with it)

__device__ __forceinline__ bf16x4 to_bf16x4(const f32x4 &v)
{
    bf16x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (__bf16)v[i];       // v_cvt_pk_bf16_f32, round to nearest even
    return o;
}

// MFMA operand whose k runs over the ROWS of a row-major bf16 tile: rows row0 .. row0 + 15, channel col0 + (lane & 31).
// EXEC must be all ones (the read gathers across lanes): only called from wave-uniform code.
__device__ __forceinline__ bf16x8 tr_operand(const __bf16 *tile, int ld, int row0, int col0, int lane)
{
    const int g4 = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const __bf16 *src = tile + (row0 + 8 * (g4 >> 1) + q) * ld + col0 + 16 * (g4 & 1) + 4 * p;
    typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(src));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(src + 4 * ld));
    // whole-vector bit casts + one shuffle: an element-by-element short -> __bf16 copy is miscompiled by this hipcc (ROCm 7.2: it keeps
    // only the first dword of each read; tools/tr_probe.hip checks the operand map on the hardware)
    return __builtin_shufflevector(__builtin_bit_cast(bf16x4, lo), __builtin_bit_cast(bf16x4, hi), 0, 1, 2, 3, 4, 5, 6, 7);
}

// ZG / ZP: g.z / prev.z are bf16 tensors (activation storage of precision mode 3) -- compile-time, like every other mode of this kernel
template <int CX, int CY, int ROWS, bool GRAM, bool YACT, bool ADD, bool DROP, bool ZG, bool ZP>
__global__ __launch_bounds__(FBB_THREADS, 1) void pw_bwd_bf16_kernel(PwBwd a)
{
    constexpr int LDG = CX + 32, LDY = CY + 32;     // bf16 elements per row of the operand tiles (2 C + 64 bytes)
    constexpr int LDZ = CY + 4;                     // fp32 row of the raw z_prev tile
    constexpr int LDW = CX + 8;                     // bf16 row of the transposed weight
    constexpr int TXN = CX / 32, TYN = CY / 32;
    constexpr int WXN = (TXN == 4 && TYN == 2) ? 4 : 2, WYN = 4 / WXN;
    constexpr int TXW = TXN / WXN, TYW = TYN / WYN;
    constexpr int STAGE = FBB_THREADS;              // all eight waves stage (two register sets of loads in flight per thread, see the loop): the
                                                    // D waves' waits are on loads two blocks old, older than any store they still have pending
    constexpr int QX = CX / 4, QY = CY / 4, SX = STAGE / QX, SY = STAGE / QY;
    constexpr int NIX = ROWS / SX, NIY = ROWS / SY;
    static_assert((ROWS / 32) * TYN == 4, "one dgrad tile per D wave");
    static_assert(NIX >= 1 && NIY >= 1 && ROWS % 16 == 0, "staging shape");
    constexpr bool NEED_Y = !GRAM;                  // Gram form: x and y are the same activated tile
    constexpr bool NEED_Z = YACT;                   // raw z_prev for the epilogue
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16 *sG = reinterpret_cast<__bf16 *>(smem_raw);                       // [2][ROWS][LDG]
    __bf16 *sY = sG + 2 * ROWS * LDG;                                        // [2][ROWS][LDY]   (NEED_Y)
    float *sZ = reinterpret_cast<float *>(sY + (NEED_Y ? 2 * ROWS * LDY : 0));   // [2][ROWS][LDZ]   (NEED_Z)
    __bf16 *sWt = reinterpret_cast<__bf16 *>(sZ + (NEED_Z ? 2 * ROWS * LDZ : 0));   // [CY][LDW]
    float *red = reinterpret_cast<float *>(sWt + CY * LDW);                  // final reductions: max(CX * SX, 8 * CY) floats

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int slot = blockIdx.x % a.n_slots, jb = blockIdx.x / a.n_slots;
    constexpr bool x_act = GRAM, has_bn = !GRAM, y_act = YACT;

    // ---- work split: items = (window of this slot, chunk of FBB_ITEM_ROWS rows), contiguous share per workgroup ----
    const int per_slot = (a.Q - slot + a.n_slots - 1) / a.n_slots;
    const int cpw = (a.max_rows + FBB_ITEM_ROWS - 1) / FBB_ITEM_ROWS;
    const int n_items = per_slot * cpw;
    const int ipb = a.items_per_block > 0 ? a.items_per_block : (n_items + a.blocks_per_slot - 1) / a.blocks_per_slot;
    const int item_begin = min(jb * ipb, n_items), item_end = min(item_begin + ipb, n_items);

    // ---- the transposed weight in bf16 ----
    if (a.w_win_stride != 0) {
        const int bi = item_begin / cpw;
        const int pidx = a.perwin_slot_major ? slot * (a.Q / a.n_slots) + bi : bi * a.n_slots + slot;
        const float *Tq = a.W + (size_t)pidx * a.w_win_stride;               // [cy][cx]: already transposed
        for (int e = tid; e < CY * (CX / 4); e += FBB_THREADS) {
            const int j = e / (CX / 4), k4 = e % (CX / 4);
            *reinterpret_cast<bf16x4 *>(sWt + j * LDW + 4 * k4) = to_bf16x4(*reinterpret_cast<const f32x4 *>(Tq + (size_t)j * CX + 4 * k4));
        }
    } else {
        const float *Wsh = a.W + (size_t)slot * a.w_slot_stride;
        for (int e = tid; e < CX * (CY / 4); e += FBB_THREADS) {
            const int k = e % CX, j4 = e / CX;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(Wsh + (size_t)k * a.ldw + 4 * j4);
#pragma unroll
            for (int i = 0; i < 4; ++i) sWt[(4 * j4 + i) * LDW + k] = (__bf16)v[i];
        }
    }
    const int stid = tid & (STAGE - 1);
    const int cqx = stid % QX, rsx = stid / QX, cqy = stid % QY, rsy = stid / QY;
    f32x4 p1 = {1.f, 1.f, 1.f, 1.f}, p2 = {0.f, 0.f, 0.f, 0.f}, p3 = {0.f, 0.f, 0.f, 0.f};
    if (has_bn) p1 = *reinterpret_cast<const f32x4 *>(a.g.P1 + (size_t)slot * CX + 4 * cqx);
    p2 = *reinterpret_cast<const f32x4 *>(a.g.P2 + (size_t)slot * CX + 4 * cqx);
    p3 = *reinterpret_cast<const f32x4 *>(a.g.P3 + (size_t)slot * CX + 4 * cqx);
    f32x4 ys = {1.f, 1.f, 1.f, 1.f}, yt = {0.f, 0.f, 0.f, 0.f};             // the input activation's affine, staging view
    if (y_act && NEED_Y) {
        ys = *reinterpret_cast<const f32x4 *>(a.prev.s + (size_t)slot * CY + 4 * cqy);
        yt = *reinterpret_cast<const f32x4 *>(a.prev.t + (size_t)slot * CY + 4 * cqy);
    }
    const uint32_t dthr = drop_threshold(a.prev.drop_p);
    const float dscale = DROP ? 1.0f / (1.0f - a.prev.drop_p) : 1.0f;

    struct Pos {
        int item, row0, row_end;
    };
    auto open_item = [&](int item, Pos &p) -> bool {
        for (; item < item_end; ++item) {
            const int q = (item / cpw) * a.n_slots + slot, ch = item % cpw;
            const int rb = a.win_off[q] + ch * FBB_ITEM_ROWS;
            const int re = min(a.win_off[q + 1], rb + FBB_ITEM_ROWS);
            if (rb < re) {
                p.item = item;
                p.row0 = rb;
                p.row_end = re;
                return true;
            }
        }
        return false;
    };
    auto advance = [&](Pos &p) -> bool {
        if (p.row0 + ROWS < p.row_end) {
            p.row0 += ROWS;
            return true;
        }
        return open_item(p.item + 1, p);
    };

    // the z tensors may be stored as bf16 (precision mode 3): 8-byte loads, kept as they arrive and widened when the tile is written to LDS.
    // Two register sets: the loads of block n + 2 are issued while block n + 1 is still in flight (see the loop) -- with one set a CU has
    // 16 .. 24 KB on the wire, which at the ~1 us of a loaded HBM round trip caps the kernel near 5 TB/s whatever the format (Little's law:
    // the bf16-stored tensors moved half the bytes in the same time).
    using ZGT = std::conditional_t<ZG, bf16x4, f32x4>;
    using ZPT = std::conditional_t<ZP, bf16x4, f32x4>;
    struct Regs {
        f32x4 dy[NIX];
        ZGT xz[NIX];
        ZPT yz[NIY];
    };
    auto widen = [](const auto &v) -> f32x4 { return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]}; };
    auto load_regs = [&](const Pos &p, Regs &R) {
#pragma unroll
        for (int i = 0; i < NIX; ++i) {
            const int row = p.row0 + rsx + SX * i;
            const size_t rr = (size_t)(row < p.row_end ? row : p.row0);
            if (!x_act) R.dy[i] = ld_stream(reinterpret_cast<const f32x4 *>(a.g.dy + rr * CX + 4 * cqx));
            R.xz[i] = ld_stream(reinterpret_cast<const ZGT *>(reinterpret_cast<const std::conditional_t<ZG, __bf16, float> *>(a.g.z) + rr * CX + 4 * cqx));
        }
        if (!GRAM) {
#pragma unroll
            for (int i = 0; i < NIY; ++i) {
                const int row = p.row0 + rsy + SY * i;
                const size_t rr = (size_t)(row < p.row_end ? row : p.row0);
                R.yz[i] = ld_stream(reinterpret_cast<const ZPT *>(reinterpret_cast<const std::conditional_t<ZP, __bf16, float> *>(a.prev.z) + rr * CY + 4 * cqy));
            }
        }
    };
    f32x4 dbacc = {0.f, 0.f, 0.f, 0.f};
    auto write_lds = [&](int buf, const Pos &p, const Regs &R) {
        __bf16 *g = sG + buf * ROWS * LDG;
        __bf16 *y = sY + buf * ROWS * LDY;
        float *z = sZ + buf * ROWS * LDZ;
#pragma unroll
        for (int i = 0; i < NIX; ++i) {
            const int lrow = rsx + SX * i, row = p.row0 + lrow;
            f32x4 xv = {0.f, 0.f, 0.f, 0.f};
            const f32x4 xz = widen(R.xz[i]);
            if (row < p.row_end) {
                if (x_act) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) xv[c] = fmaxf(fmaf(xz[c], p2[c], p3[c]), 0.f);
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) xv[c] = fmaf(R.dy[i][c], p1[c], fmaf(xz[c], p2[c], p3[c]));
                }
                dbacc += xv;
            }
            *reinterpret_cast<bf16x4 *>(g + lrow * LDG + 4 * cqx) = to_bf16x4(xv);
            if (GRAM && NEED_Z) *reinterpret_cast<f32x4 *>(z + lrow * LDZ + 4 * cqx) = xz;     // CX == CY here
        }
        if (!GRAM) {
#pragma unroll
            for (int i = 0; i < NIY; ++i) {
                const int lrow = rsy + SY * i, row = p.row0 + lrow;
                const f32x4 yz = widen(R.yz[i]);
                f32x4 yv = yz;
                if (y_act) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) yv[c] = fmaxf(fmaf(yv[c], ys[c], yt[c]), 0.f);
                    if (DROP) {
                        const uint32_t el = (uint32_t)row * (uint32_t)CY + (uint32_t)(4 * cqy);
#pragma unroll
                        for (int c = 0; c < 4; ++c) yv[c] = (mix32((el + c) ^ a.prev.drop_seed) >= dthr) ? yv[c] * dscale : 0.f;
                    }
                }
                if (!(row < p.row_end)) yv = f32x4{0.f, 0.f, 0.f, 0.f};        // rows past the block's end contribute nothing to dW
                *reinterpret_cast<bf16x4 *>(y + lrow * LDY + 4 * cqy) = to_bf16x4(yv);
                if (NEED_Z) *reinterpret_cast<f32x4 *>(z + lrow * LDZ + 4 * cqy) = yz;
            }
        }
    };

    // ---- role state ----
    const bool w_role = wave < 4;
    const int ww = wave & 3;
    const int tx0 = (ww / WYN) * TXW, ty0 = (ww % WYN) * TYW;
    f32x16 acc_w[TXW][TYW];
#pragma unroll
    for (int i = 0; i < TXW; ++i)
#pragma unroll
        for (int j = 0; j < TYW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc_w[i][j][e] = 0.f;
    // D role: tile (rt, dty), lane = output column
    const int rt = ww / TYN, dty = ww % TYN, dcol = 32 * dty + r;
    const float c_b = a.bias_slot ? a.bias_slot[(size_t)slot * CY + dcol] : 0.f;
    const float c_s = y_act ? a.prev.s[(size_t)slot * CY + dcol] : 1.0f;
    const float c_t = y_act ? a.prev.t[(size_t)slot * CY + dcol] : 0.0f;
    const float c_m = (y_act && a.prev_mean) ? a.prev_mean[(size_t)slot * CY + dcol] : 0.0f;
    const float c_i = (y_act && a.prev_invstd) ? a.prev_invstd[(size_t)slot * CY + dcol] : 0.0f;
    const bool do_part = a.part_a != nullptr;
    float s_a = 0.f, s_b = 0.f;

    // Positions of blocks n (in LDS), n + 1 and n + 2 (in registers, in flight).  A tail position that does not exist repeats the last
    // real one: the loads are issued unconditionally (a conditional load would make every wait in the loop a vmcnt(0), which is the
    // one-deep pipeline again) and their data is simply not written.
    Pos cur, nxt, nx2;
    bool live = open_item(item_begin, cur);
    bool more1 = false, more2 = false;
    nxt = cur;
    if (live) more1 = advance(nxt);
    if (!more1) nxt = cur;
    nx2 = nxt;
    if (more1) more2 = advance(nx2);
    if (!more2) nx2 = nxt;
    Regs S0, S1;
    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): every constant has landed before the loop (see pw_bwd_fused.hip)
    if (live) {
        load_regs(cur, S0);
        load_regs(nxt, S1);
    }
    __syncthreads();                     // sWt staged
    if (live) {
        write_lds(0, cur, S0);
        load_regs(nx2, S0);
    }
    __syncthreads();
    int buf = 0;
    // one block of rows: compute block n from LDS, write block n + 1 (register set A) into the other buffer, refill A with block n + 3
    // The two roles run the loop as two separate instantiations (role_tag): the wait for a register set is then a static count of the
    // role's own younger memory operations (W: the other set's loads; D: those + its 32 stores) instead of the conservative merge of
    // both paths, which made the D waves wait for the stores they had just issued.  Same number of barriers on both sides.
    auto step = [&](Regs &A, auto role_tag) {
        constexpr bool W_ROLE = decltype(role_tag)::value;
        const __bf16 *g = sG + buf * ROWS * LDG;
        const __bf16 *y = GRAM ? g : sY + buf * ROWS * LDY;
        constexpr int LDYY = GRAM ? LDG : LDY;
        const float *z = sZ + buf * ROWS * LDZ;
        if constexpr (W_ROLE) {
#pragma unroll
            for (int s2 = 0; s2 < ROWS / 16; ++s2) {
                bf16x8 xa[TXW], yb[TYW];
#pragma unroll
                for (int i = 0; i < TXW; ++i) xa[i] = tr_operand(g, LDG, 16 * s2, 32 * (tx0 + i), lane);
#pragma unroll
                for (int j = 0; j < TYW; ++j) yb[j] = tr_operand(y, LDYY, 16 * s2, 32 * (ty0 + j), lane);
#pragma unroll
                for (int i = 0; i < TXW; ++i)
#pragma unroll
                    for (int j = 0; j < TYW; ++j) acc_w[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[i], yb[j], acc_w[i][j], 0, 0, 0);
            }
        } else {
            const int valid = min(ROWS, cur.row_end - cur.row0) - 32 * rt;       // rows of this tile that exist (may be <= 0)
            const int trow0 = cur.row0 + 32 * rt;
            float addv[16];
            if (ADD) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int rr = (e & 3) + 8 * (e >> 2) + 4 * h;
                    addv[e] = rr < valid ? a.add[(size_t)(trow0 + rr) * CY + dcol] : 0.f;
                }
            }
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            const __bf16 *ga = g + (32 * rt + r) * LDG + 8 * h;
            const __bf16 *wb = sWt + dcol * LDW + 8 * h;
#pragma unroll
            for (int s2 = 0; s2 < CX / 16; ++s2) {
                const bf16x8 av = *reinterpret_cast<const bf16x8 *>(ga + 16 * s2);
                const bf16x8 bv = *reinterpret_cast<const bf16x8 *>(wb + 16 * s2);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
            }
            float zv[16];
            if (YACT) {
#pragma unroll
                for (int e = 0; e < 16; ++e) zv[e] = z[(32 * rt + (e & 3) + 8 * (e >> 2) + 4 * h) * LDZ + dcol];
            }
            float *op = a.out + (size_t)(trow0 + 4 * h) * CY + dcol;
            auto finish = [&](auto full_tag) {
                constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int rr = (e & 3) + 8 * (e >> 2) + 4 * h;
                    const bool ok = FULL || rr < valid;
                    float v = acc[e] + c_b;
                    if (ADD) v += addv[e];
                    if (YACT) {
                        if (DROP) {
                            const uint32_t el = (uint32_t)(trow0 + rr) * (uint32_t)CY + (uint32_t)dcol;
                            v = (mix32(el ^ a.prev.drop_seed) >= dthr) ? v * dscale : 0.f;
                        }
                        v = fmaf(zv[e], c_s, c_t) > 0.f ? v : 0.f;
                        const float vs = ok ? v : 0.f;
                        s_a += vs;
                        s_b = fmaf(vs, (zv[e] - c_m) * c_i, s_b);
                    }
                    if (ok) st_stream(v, &op[((e & 3) + 8 * (e >> 2)) * CY]);
                }
            };
            if (valid >= 32) finish(std::true_type{});
            else finish(std::false_type{});
        }
        Pos nx3 = nx2;
        const bool more3 = more2 && advance(nx3);
        if (!more3) nx3 = nx2;
        if (more1) write_lds(buf ^ 1, nxt, A);
        load_regs(nx3, A);
        __syncthreads();
        buf ^= 1;
        cur = nxt;
        nxt = nx2;
        nx2 = nx3;
        live = more1;
        more1 = more2;
        more2 = more3;
    };
    if (w_role) {
        while (live) {
            step(S1, std::true_type{});
            if (!live) break;
            step(S0, std::true_type{});
        }
    } else {
        while (live) {
            step(S1, std::false_type{});
            if (!live) break;
            step(S0, std::false_type{});
        }
    }

    // ---- flush: weight-gradient partial of this workgroup, bias sums, BatchNorm-backward sums ----
    if (w_role) {
        float *dst = a.dWpart + (size_t)blockIdx.x * CX * CY;
#pragma unroll
        for (int i = 0; i < TXW; ++i)
#pragma unroll
            for (int j = 0; j < TYW; ++j) {
                const int cy = 32 * (ty0 + j) + r;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int cx = 32 * (tx0 + i) + (e & 3) + 8 * (e >> 2) + 4 * h;
                    dst[(size_t)cx * CY + cy] = acc_w[i][j][e];
                }
            }
    }
    if (a.dbpart) {
        *reinterpret_cast<f32x4 *>(red + rsx * CX + 4 * cqx) = dbacc;
        __syncthreads();
        if (tid < CX) {
            float s = 0.f;
#pragma unroll
            for (int gi = 0; gi < SX; ++gi) s += red[gi * CX + tid];
            a.dbpart[(size_t)blockIdx.x * CX + tid] = s;
        }
        __syncthreads();
    }
    if (do_part) {
        const float oa = __shfl_xor(s_a, 32), ob = __shfl_xor(s_b, 32);
        if (!w_role && h == 0) {
            red[(rt * CY + dcol) * 2 + 0] = s_a + oa;
            red[(rt * CY + dcol) * 2 + 1] = s_b + ob;
        }
        __syncthreads();
        if (tid < CY) {
            float sa = 0.f, sb = 0.f;
#pragma unroll
            for (int t = 0; t < ROWS / 32; ++t) {
                sa += red[(t * CY + tid) * 2 + 0];
                sb += red[(t * CY + tid) * 2 + 1];
            }
            a.part_a[(size_t)blockIdx.x * CY + tid] = sa;
            a.part_b[(size_t)blockIdx.x * CY + tid] = sb;
        }
    }
}

template <int CX, int CY, int ROWS, bool GRAM, bool YACT, bool ADD, bool DROP, bool ZG, bool ZP>
static int launch_bf16_z(const PwBwd &a, hipStream_t st)
{
    constexpr int SX = FBB_THREADS / (CX / 4);
    constexpr size_t red_floats = (size_t)(CX * SX > 8 * CY ? CX * SX : 8 * CY);
    constexpr size_t lds = (size_t)2 * ROWS * (CX + 32) * 2 + (GRAM ? 0 : (size_t)2 * ROWS * (CY + 32) * 2) + (YACT ? (size_t)2 * ROWS * (CY + 4) * 4 : 0) +
                           (size_t)CY * (CX + 8) * 2 + red_floats * 4;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static bool attr_set = false;
    auto kern = pw_bwd_bf16_kernel<CX, CY, ROWS, GRAM, YACT, ADD, DROP, ZG, ZP>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(AMPNET_E_LAUNCH, "pw_bwd_bf16: hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
        attr_set = true;
    }
    char name[64];
    snprintf(name, sizeof(name), "pw_bwd<%d,%d>%s bf16", CX, CY, a.g.act ? "+gram" : "");
    const double rows = (double)a.rows_hint;
    const bool same = a.g.act && a.g.z == a.prev.z;
    const double zgb = a.g.z_bf16 ? 2.0 : 4.0, zpb = a.prev.z_bf16 ? 2.0 : 4.0;
    ProfScope prof(name, 4.0 * rows * CX * CY, rows * ((a.g.dy ? 4.0 * CX : 0.0) + ((a.g.P1 || a.g.act) ? zgb * CX : 0.0) + (same ? 0.0 : zpb * CY) + 4.0 * CY + (a.add ? 4.0 * CY : 0.0)), st);
    hipLaunchKernelGGL(kern, dim3(a.blocks_per_slot * a.n_slots), dim3(FBB_THREADS), lds, st, a);
    return check_launch("pw_bwd_bf16_kernel");
}

// storage formats of the two z tensors: both fp32 (modes 1 / 2), both bf16, or one of them an fp32 tensor that crosses the ABI
// (`local` as a layer input: ZP = false; the bmm backward's d_local as "z": ZG = false)
template <int CX, int CY, int ROWS, bool GRAM, bool YACT, bool ADD, bool DROP = false>
static int launch_bf16_x(const PwBwd &a, hipStream_t st)
{
    const bool zg = a.g.z_bf16 != 0, zp = a.prev.z_bf16 != 0;
    if (!zg && !zp) return launch_bf16_z<CX, CY, ROWS, GRAM, YACT, ADD, DROP, false, false>(a, st);
    if (zg && zp) return launch_bf16_z<CX, CY, ROWS, GRAM, YACT, ADD, DROP, true, true>(a, st);
    if constexpr (!GRAM && !DROP) {
        if (zg) return launch_bf16_z<CX, CY, ROWS, GRAM, YACT, ADD, DROP, true, false>(a, st);
        if constexpr (CX == 64 && CY == 64 && YACT && !ADD) return launch_bf16_z<CX, CY, ROWS, GRAM, YACT, ADD, DROP, false, true>(a, st);
    }
    return fail(AMPNET_E_ARG, "pw_bwd_bf16<%d,%d>: storage combination (g.z bf16 = %d, prev.z bf16 = %d) not built", CX, CY, (int)zg, (int)zp);
}

template <int CX, int CY, int ROWS>
static int launch_bf16(const PwBwd &a, hipStream_t st)
{
    const bool gram = a.g.act != 0, yact = a.prev.s != nullptr, add = a.add != nullptr;
    if constexpr (CX == 64 && CY == 128) {
        if (gram || add || !yact) return fail(AMPNET_E_ARG, "pw_bwd_bf16: 64 x 128 is built for an activated input without addend");
        return a.prev.drop_p > 0.f ? launch_bf16_x<CX, CY, ROWS, false, true, false, true>(a, st) : launch_bf16_x<CX, CY, ROWS, false, true, false, false>(a, st);
    }
    if (a.prev.drop_p > 0.f) return fail(AMPNET_E_ARG, "pw_bwd_bf16: dropout only built for 64 x 128");
    if (gram) {
        if (CX != CY || !yact || add) return fail(AMPNET_E_ARG, "pw_bwd_bf16: Gram form needs CX == CY, an activated input and no addend");
        if constexpr (CX == CY) return launch_bf16_x<CX, CY, ROWS, true, true, false>(a, st);
    }
    if (add) {
        if constexpr (CX == 64 && CY == 64)
            return yact ? launch_bf16_x<CX, CY, ROWS, false, true, true>(a, st) : launch_bf16_x<CX, CY, ROWS, false, false, true>(a, st);
        return fail(AMPNET_E_ARG, "pw_bwd_bf16: addend only built for 64 x 64");
    }
    return yact ? launch_bf16_x<CX, CY, ROWS, false, true, false>(a, st) : launch_bf16_x<CX, CY, ROWS, false, false, false>(a, st);
}

// same argument contract as pw_bwd_fused (it validates before dispatching here)
int pw_bwd_fused_bf16(const PwBwd &a, hipStream_t st)
{
    static_assert(FBB_ITEM_ROWS == 256, "item size shared with pw_bwd_fused.hip");
    if (a.g.C == 128 && a.prev.C == 128) return launch_bf16<128, 128, 32>(a, st);
    if (a.g.C == 128 && a.prev.C == 64) return launch_bf16<128, 64, 64>(a, st);
    if (a.g.C == 64 && a.prev.C == 64) return launch_bf16<64, 64, 64>(a, st);
    if (a.g.C == 64 && a.prev.C == 128) return launch_bf16<64, 128, 32>(a, st);
    return fail(AMPNET_E_ARG, "pw_bwd_bf16: %d x %d not built", a.g.C, a.prev.C);
}

}  // namespace ampnet
