// pw_bwd_fused.hip -- weight gradient AND data gradient of a shared-weight per-point layer in one pass.
//
// pw_wgrad and pw_dgrad (pw_bwd.hip) each stream (dy_l, z_l, z_{l-1}) from HBM; at 64..128 channels both sit on the
// memory system, not on the matrix cores.  Here a workgroup stages one block of rows ONCE:
//     sG[row][cx] = g = dy * P1 + z * P2 + P3   (or relu(z * P2 + P3) for the Gram form of the pooled layers)
//     sZ[row][cy] = z_{l-1} (raw: the activation is applied when read, the ReLU mask / zhat come from the same tile)
// and its 8 waves split by ROLE, one wave of each role per SIMD:
//     waves 0-3 (W): dW[cx][cy] += sum_rows sG[row][cx] * act(sZ[row][cy])       accumulators live across the whole grid-stride
//     waves 4-7 (D): out[row][cy] = mask * (sum_cx sG[row][cx] * W[cx][cy] + ...)  one 32 x 32 tile per wave, + the sums
//                    sum dy_{l-1}, sum dy_{l-1} * zhat_{l-1} of the next BatchNorm backward
// Both roles issue the same number of MFMAs per block of rows (it is the same rows x CX x CY product).
// Workgroups are persistent: (slot, j) walks a contiguous share of the (window, chunk) items of its slot, so the
// per-workgroup partials are few (grid of them, not windows x chunks), every one belongs to one BatchNorm slot, and
// weights / constants are staged once.  Fixed assignment, no atomics: bitwise reproducible.
#include <cstdlib>
#include <type_traits>
#include "kernels.h"

namespace ampnet {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int FB_THREADS = 512;
constexpr int FB_ITEM_ROWS = 256;      // granularity of the work split inside a slot

// GRAM: X operand = relu(z * P2 + P3) of the SAME tensor as z_{l-1} (pooled layers); otherwise dense dy with BatchNorm
// constants.  YACT: the layer's input is relu(bn(z_{l-1})) (mask + sums), else z_{l-1} itself.  ADD: extra addend.
// The modes are compile-time so that the prefetch loads sit in one basic block (no conservative vmcnt(0) between them).
template <int CX, int CY, int ROWS, bool GRAM, bool YACT, bool ADD, bool DROP = false>
__global__ __launch_bounds__(FB_THREADS, 1) void pw_bwd_kernel(PwBwd a)
{
    constexpr int LDG = CX + 4, LDZ = CY + 4;
    constexpr int TXN = CX / 32, TYN = CY / 32;
    constexpr int WXN = (TXN == 4 && TYN == 2) ? 4 : 2, WYN = 4 / WXN;
    constexpr int TXW = TXN / WXN, TYW = TYN / WYN;
    constexpr int FB_STAGE = 256;                   // the four W waves stage; the D waves only compute and store, so
                                                    // they never wait on a load behind their own pending stores
    constexpr int QX = CX / 4, QY = CY / 4, SX = FB_STAGE / QX, SY = FB_STAGE / QY;
    constexpr int NIX = ROWS / SX, NIY = ROWS / SY;
    // (Round 3 tried twice the rows per block for the 64-input layers -- <64,64,128>, <64,128,64>, two data-gradient tiles per D wave, half the barriers
    // per MFMA: 255 -> 335 us and 475 -> 495 us, the doubled staging registers spill and the roles pipeline worse.  Static wave priorities for either
    // role: +-0.5 %.  Both A/B'd on one box, gpurun_out r3A / r3z.)
    static_assert((ROWS / 32) * TYN == 4, "one dgrad tile per D wave");
    static_assert(NIX >= 1 && NIY >= 1, "staging shape");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sG = smem;                               // [2][ROWS][LDG]
    float *sZ = sG + 2 * ROWS * LDG;                // [2][ROWS][LDZ]
    float *sWt = sZ + 2 * ROWS * LDZ;               // [CY][LDG]: sWt[j][k] = W[k][j]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int slot = blockIdx.x % a.n_slots, jb = blockIdx.x / a.n_slots;
    constexpr bool x_act = GRAM, has_bn = !GRAM, same = GRAM, y_act = YACT;
    // SACT: the layer's input activation relu(bn(z_{l-1})) is applied ONCE while the tile is staged and sZ holds a = relu(z s + t): the four
    // weight-gradient waves used to recompute it for every k step (2 VALU per MFMA, and VALU time is matrix-pipe time here); the
    // data-gradient role gets its ReLU mask as a > 0 and zhat = (z - mean) invstd = (a - beta) / gamma where the mask holds
    // Gram form: sG already IS that activated tile (x and y are the same tensor), so the raw copy in sZ is not written at all.
    // DROP (dropout on that activation): the staged value is the DROPPED activation a' = keep ? a / (1 - p) : 0 -- the hash of the element
    // index is taken once per element here instead of once per k step in four waves plus once per output element; a' > 0 is ReLU mask
    // and keep mask in one, a = a' (1 - p) where it holds.
    constexpr bool SACT = YACT;

    // ---- work split: items = (window of this slot, chunk of FB_ITEM_ROWS rows), contiguous share per workgroup ----
    const int per_slot = (a.Q - slot + a.n_slots - 1) / a.n_slots;
    const int cpw = (a.max_rows + FB_ITEM_ROWS - 1) / FB_ITEM_ROWS;
    const int n_items = per_slot * cpw;
    const int ipb = a.items_per_block > 0 ? a.items_per_block : (n_items + a.blocks_per_slot - 1) / a.blocks_per_slot;
    const int item_begin = min(jb * ipb, n_items), item_end = min(item_begin + ipb, n_items);

    // ---- stage the transposed weight and load the per-thread constants ----
    if (a.w_win_stride != 0) {
        // per-window matrix T[pidx][cy][cx] (the bmm transform, already "transposed": out[row][cy] = sum_cx g[row][cx] T[cy][cx]);
        // the host keeps every workgroup inside one window (items_per_block divides the chunks per window)
        const int bi = item_begin / cpw;
        const int pidx = a.perwin_slot_major ? slot * (a.Q / a.n_slots) + bi : bi * a.n_slots + slot;
        const float *Tq = a.W + (size_t)pidx * a.w_win_stride;
        for (int e = tid; e < CY * (CX / 4); e += FB_THREADS) {
            const int j = e / (CX / 4), k4 = e % (CX / 4);
            *reinterpret_cast<f32x4 *>(sWt + j * LDG + 4 * k4) = *reinterpret_cast<const f32x4 *>(Tq + (size_t)j * CX + 4 * k4);
        }
    } else {
        const float *Wsh = a.W + (size_t)slot * a.w_slot_stride;
        for (int e = tid; e < CX * (CY / 4); e += FB_THREADS) {
            const int k = e % CX, j4 = e / CX;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(Wsh + (size_t)k * a.ldw + 4 * j4);
#pragma unroll
            for (int i = 0; i < 4; ++i) sWt[(4 * j4 + i) * LDG + k] = v[i];
        }
    }
    const int stid = tid & (FB_STAGE - 1);
    const int cqx = stid % QX, rsx = stid / QX, cqy = stid % QY, rsy = stid / QY;
    f32x4 p1 = {1.f, 1.f, 1.f, 1.f}, p2 = {0.f, 0.f, 0.f, 0.f}, p3 = {0.f, 0.f, 0.f, 0.f};
    if (has_bn && a.fin_part_a) {
        // The BatchNorm-backward constants of this layer, formed HERE from the partial sums its producer left (kernels.h): SX groups of
        // threads split the slot's partials (group gq takes partials gq, gq + SX, ... of the slot, eight loads in flight), double sums, one
        // fixed-order merge through LDS.  ~30 partials of <= 512 B per workgroup out of L2: noise next to the pass over the rows, and a
        // launch (5 .. 12 us of an idle chip) less per layer.
        double sa[4] = {0.0, 0.0, 0.0, 0.0}, sb[4] = {0.0, 0.0, 0.0, 0.0};
        const int per_slot_parts = (a.fin_parts - slot + a.n_slots - 1) / a.n_slots;
        if ((tid < FB_STAGE)) {
            for (int k0 = rsx; k0 < per_slot_parts; k0 += SX * 8) {
                f32x4 va[8], vb[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = k0 + SX * u;
                    const size_t o = (size_t)(slot + (k < per_slot_parts ? k : 0) * a.n_slots) * CX + 4 * cqx;
                    va[u] = *reinterpret_cast<const f32x4 *>(a.fin_part_a + o);
                    vb[u] = *reinterpret_cast<const f32x4 *>(a.fin_part_b + o);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (k0 + SX * u < per_slot_parts) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            sa[c] += (double)va[u][c];
                            sb[c] += (double)vb[u][c];
                        }
                    }
                }
            }
            double *red = reinterpret_cast<double *>(smem);          // [SX][CX][2]: the staging buffers are not in use yet
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                red[((size_t)rsx * CX + 4 * cqx + c) * 2 + 0] = sa[c];
                red[((size_t)rsx * CX + 4 * cqx + c) * 2 + 1] = sb[c];
            }
        }
        __syncthreads();
        if ((tid < FB_STAGE)) {
            const double *red = reinterpret_cast<const double *>(smem);
            const double n = (double)a.fin_rows;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ch = 4 * cqx + c;
                double A = 0.0, Bs = 0.0;
                for (int gq = 0; gq < SX; ++gq) {
                    A += red[((size_t)gq * CX + ch) * 2 + 0];
                    Bs += red[((size_t)gq * CX + ch) * 2 + 1];
                }
                const size_t o = (size_t)slot * CX + ch;
                const double invstd = a.fin_invstd[o], mean = a.fin_mean[o];
                const double s = (double)a.fin_gamma[ch] * invstd;
                const double q2 = -s * invstd * Bs / n;
                p1[c] = (float)s;
                p2[c] = (float)q2;
                p3[c] = (float)(-s * A / n - q2 * mean);
                if (jb == 0 && rsx == 0) {                       // one writer per slot: the arrays other kernels read
                    a.fin_slot_ab[o * 2 + 0] = (float)A;
                    a.fin_slot_ab[o * 2 + 1] = (float)Bs;
                }
            }
            if (jb == 0 && rsx == 0) {
                *reinterpret_cast<f32x4 *>(a.fin_P1 + (size_t)slot * CX + 4 * cqx) = p1;
                *reinterpret_cast<f32x4 *>(a.fin_P2 + (size_t)slot * CX + 4 * cqx) = p2;
                *reinterpret_cast<f32x4 *>(a.fin_P3 + (size_t)slot * CX + 4 * cqx) = p3;
            }
        }
        __syncthreads();                                             // smem is about to be overwritten by the weights / first tile
    } else if (has_bn) {
        p1 = *reinterpret_cast<const f32x4 *>(a.g.P1 + (size_t)slot * CX + 4 * cqx);
    }
    if ((has_bn && !a.fin_part_a) || x_act) {
        p2 = *reinterpret_cast<const f32x4 *>(a.g.P2 + (size_t)slot * CX + 4 * cqx);
        p3 = *reinterpret_cast<const f32x4 *>(a.g.P3 + (size_t)slot * CX + 4 * cqx);
    }
    const uint32_t dthr = drop_threshold(a.prev.drop_p);
    const float dscale = DROP ? 1.0f / (1.0f - a.prev.drop_p) : 1.0f;
    f32x4 ys4 = {1.f, 1.f, 1.f, 1.f}, yt4 = {0.f, 0.f, 0.f, 0.f};
    if (SACT && !GRAM) {
        ys4 = *reinterpret_cast<const f32x4 *>(a.prev.s + (size_t)slot * CY + 4 * cqy);
        yt4 = *reinterpret_cast<const f32x4 *>(a.prev.t + (size_t)slot * CY + 4 * cqy);
    }

    // ---- the walk over blocks of ROWS rows (crosses item boundaries so that the prefetch never drains) ----
    struct Pos {
        int item, row0, row_end;        // current block = rows [row0, min(row0 + ROWS, row_end))
    };
    auto open_item = [&](int item, Pos &p) -> bool {       // first block of the next non-empty item at or after `item`
        for (; item < item_end; ++item) {
            const int q = (item / cpw) * a.n_slots + slot, ch = item % cpw;
            const int rb = a.win_off[q] + ch * FB_ITEM_ROWS;
            const int re = min(a.win_off[q + 1], rb + FB_ITEM_ROWS);
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

    f32x4 rx_dy[NIX], rx_z[NIX], ry_z[NIY];
    // lane offsets (floats) inside a block of rows: row (rsx + SX i), columns 4 cqx ..; the block's first row is workgroup-uniform, so a
    // full block is addressed as (uniform base in SGPRs) + (32-bit lane offset) + (compile-time i * SX * CX): no 64-bit VALU address
    // arithmetic and no row clamps per load (99 VALU instructions for 20 loads before -- and VALU time is matrix-pipe time here)
    // (Tried in round 3 and dropped: buffer addressing for these loads -- resource in SGPRs, one constant VGPR offset, SALU row offsets, i.e.
    // zero VALU instructions per load instead of ~5 for the 64-bit flat address -- made every instantiation 3 .. 10 % SLOWER on the MI355X
    // in a same-box A/B (gpurun_out r3j: <128,64> 439 -> 480 us, <64,64> 245 -> 291 us), as did reading the inputs from an L2-resident
    // window, which changed nothing: the loads are not what this kernel waits for.)
    auto load_regs = [&](const Pos &p) {
#pragma unroll
        for (int i = 0; i < NIX; ++i) {
            const int row = p.row0 + rsx + SX * i;
            size_t rr = (size_t)(row < p.row_end ? row : p.row0);
            if (a.dbg_row_wrap) rr &= (size_t)(a.dbg_row_wrap - 1);       // power of two
            if (!x_act) rx_dy[i] = ld_stream(reinterpret_cast<const f32x4 *>(a.g.dy + rr * CX + 4 * cqx));
            if (has_bn || x_act) rx_z[i] = ld_stream(reinterpret_cast<const f32x4 *>(a.g.z + rr * CX + 4 * cqx));
        }
        if (!same) {
#pragma unroll
            for (int i = 0; i < NIY; ++i) {
                const int row = p.row0 + rsy + SY * i;
                size_t rr = (size_t)(row < p.row_end ? row : p.row0);
                if (a.dbg_row_wrap) rr &= (size_t)(a.dbg_row_wrap - 1);
                ry_z[i] = ld_stream(reinterpret_cast<const f32x4 *>(a.prev.z + rr * CY + 4 * cqy));
            }
        }
    };
    f32x4 dbacc = {0.f, 0.f, 0.f, 0.f};
    const bool want_db = a.dbpart != nullptr;          // uniform: the layers without a bias skip 32 adds per block
    auto write_lds = [&](int buf, const Pos &p) {
        float *g = sG + buf * ROWS * LDG, *z = sZ + buf * ROWS * LDZ;
#pragma unroll
        for (int i = 0; i < NIX; ++i) {
            const int row = p.row0 + rsx + SX * i;
            f32x4 xv = {0.f, 0.f, 0.f, 0.f};
            if (row < p.row_end) {
                if (x_act) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) xv[c] = fmaxf(fmaf(rx_z[i][c], p2[c], p3[c]), 0.f);
                } else if (has_bn) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) xv[c] = fmaf(rx_dy[i][c], p1[c], fmaf(rx_z[i][c], p2[c], p3[c]));
                } else {
                    xv = rx_dy[i];
                }
                if (want_db) dbacc += xv;
            }
            *reinterpret_cast<f32x4 *>(g + (rsx + SX * i) * LDG + 4 * cqx) = xv;
            if (same && !SACT) *reinterpret_cast<f32x4 *>(z + (rsx + SX * i) * LDZ + 4 * cqx) = rx_z[i];   // CX == CY here
        }
        if (!same) {
#pragma unroll
            for (int i = 0; i < NIY; ++i) {
                f32x4 zv4 = ry_z[i];
                if (SACT) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) zv4[c] = fmaxf(fmaf(zv4[c], ys4[c], yt4[c]), 0.f);
                    if (DROP) {
                        const uint32_t e0 = (uint32_t)(p.row0 + rsy + SY * i) * (uint32_t)CY + (uint32_t)(4 * cqy);
#pragma unroll
                        for (int c = 0; c < 4; ++c) zv4[c] = (mix32((e0 + c) ^ a.prev.drop_seed) >= dthr) ? zv4[c] * dscale : 0.f;
                    }
                }
                *reinterpret_cast<f32x4 *>(z + (rsy + SY * i) * LDZ + 4 * cqy) = zv4;
            }
        }
    };

    // ---- role state ----
    const bool w_role = wave < 4;
    const int ww = wave & 3;
    // W role: tiles (tx0 .. tx0 + TXW) x (ty0 .. ty0 + TYW)
    const int tx0 = (ww / WYN) * TXW, ty0 = (ww % WYN) * TYW;
    f32x16 acc_w[TXW][TYW];
#pragma unroll
    for (int i = 0; i < TXW; ++i)
#pragma unroll
        for (int j = 0; j < TYW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc_w[i][j][e] = 0.f;
    // Gram form at 128 x 128: the matrix is symmetric, so the four W waves own the 10 tiles of its upper triangle
    // (3, 3, 2, 2) instead of all 16 and mirror them when they flush: 112 instead of 128 MFMAs per SIMD and block of rows
    constexpr bool SYM = GRAM && CX == 128 && CY == 128;
    // waves 0, 1 own the three tiles over two column blocks {B0, B1} = {0, 1} / {2, 3}: (B0,B0) (B0,B1) (B1,B1); waves 2, 3 own
    // (0,B2) (1,B2) with B2 = 2 / 3.  Operand A of a tile and operand B of another are the same LDS words (same lane mapping), so a
    // wave reads each of its 2 or 3 column blocks ONCE per k step: 10 LDS reads per 10 MFMAs instead of 20 (an LDS read costs
    // issue cycles the matrix pipe does not get back, tools/mfma_probe2.hip)
    const int sym_b0 = ww == 1 ? 2 : 0, sym_b1 = ww == 1 ? 3 : 1, sym_b2 = ww == 3 ? 3 : 2;
    const bool sym3 = ww < 2;
    float wys[TYW], wyt[TYW];
#pragma unroll
    for (int j = 0; j < TYW; ++j) {
        const int col = 32 * (ty0 + j) + r;
        wys[j] = y_act ? a.prev.s[(size_t)slot * CY + col] : 1.0f;
        wyt[j] = y_act ? a.prev.t[(size_t)slot * CY + col] : 0.0f;
    }
    // D role: tile (rt, ty), lane = output column
    const int rt = ww / TYN, dty = ww % TYN, dcol = 32 * dty + r;
    const float c_b = a.bias_slot ? a.bias_slot[(size_t)slot * CY + dcol] : 0.f;
    const float c_s = y_act ? a.prev.s[(size_t)slot * CY + dcol] : 1.0f;
    const float c_t = y_act ? a.prev.t[(size_t)slot * CY + dcol] : 0.0f;
    const float c_m = (y_act && a.prev_mean) ? a.prev_mean[(size_t)slot * CY + dcol] : 0.0f;
    const float c_i = (y_act && a.prev_invstd) ? a.prev_invstd[(size_t)slot * CY + dcol] : 0.0f;
    const float c_beta = fmaf(c_m, c_s, c_t);                      // t = beta - mean s
    const float c_invg = c_s != 0.f ? c_i / c_s : 0.f;             // gamma = s / invstd; gamma = 0: P2 = 0, any finite zhat will do
    const float c_undrop = DROP ? 1.0f - a.prev.drop_p : 1.0f;     // a = a' (1 - p) for a kept element
    const bool do_part = a.part_a != nullptr;
    float s_a = 0.f, s_b = 0.f;
    // (Round 4, tried and dropped: the phase skew that helps the split kernel pw_bwd_x3.hip -- the W wave stages block n + 1 BEFORE its
    // products of block n, so that one wave's VALU phase lies under the other's MFMA phase -- made THIS kernel slower in a same-box A/B:
    // <128,64> 484 -> 536 us, <64,64> 271 -> 288 us, <64,128> +2 % (gpurun_out/r4_ab7): next to the 64-cycle fp32 MFMA a VALU instruction
    // of the partner wave is not hidden, it only competes.)
    Pos cur, nxt;
    bool live = open_item(item_begin, cur);
    if (live && w_role) load_regs(cur);
    __syncthreads();                     // sWt staged
    if (live && w_role) write_lds(0, cur);
    __syncthreads();
    int buf = 0;
    // every constant loaded above has landed before the loop: inside it, a wait on them would also wait for the
    // D waves' own stores (loads and stores share vmcnt on gfx9-family parts)
    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
    while (live) {
        nxt = cur;
        const bool more = advance(nxt);
        if (more && w_role) load_regs(nxt);
        const float *g = sG + buf * ROWS * LDG, *z = sZ + buf * ROWS * LDZ;
        if (w_role && SYM) {
            // Ten tiles over four waves would be 3, 3, 2, 2 per k step (48 / 32 MFMAs per block of rows next to the D wave's 64 on the same
            // SIMD: the workgroup runs at the pace of the 112).  The diagonal tiles (1,1) and (3,3) are therefore split over the rows:
            // waves 0 / 1 take them for the first half of the k steps, waves 2 / 3 (which read that column block anyway) for the
            // second half into an accumulator of their own, added at the flush: 40 MFMAs per wave and block of rows.
            const float *gc0 = g + 32 * sym_b0 + r, *gc1 = g + 32 * sym_b1 + r, *gc2 = g + 32 * sym_b2 + r;
            constexpr int HALF = ROWS / 4;
            if (sym3) {
                float c0_n = gc0[h * LDG], c1_n = gc1[h * LDG];
#pragma unroll
                for (int s2 = 0; s2 < ROWS / 2; ++s2) {
                    const float c0 = c0_n, c1 = c1_n;
                    if (s2 + 1 < ROWS / 2) {
                        c0_n = gc0[(2 * s2 + 2 + h) * LDG];
                        c1_n = gc1[(2 * s2 + 2 + h) * LDG];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    acc_w[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(c0, c0, acc_w[0][0], 0, 0, 0);
                    acc_w[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(c0, c1, acc_w[0][1], 0, 0, 0);
                    if (s2 < HALF) acc_w[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(c1, c1, acc_w[1][0], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                float c0_n = gc0[h * LDG], c1_n = gc1[h * LDG], c2_n = gc2[h * LDG];
#pragma unroll
                for (int s2 = 0; s2 < ROWS / 2; ++s2) {
                    const float c0 = c0_n, c1 = c1_n, c2 = c2_n;
                    if (s2 + 1 < ROWS / 2) {
                        c0_n = gc0[(2 * s2 + 2 + h) * LDG];
                        c1_n = gc1[(2 * s2 + 2 + h) * LDG];
                        c2_n = gc2[(2 * s2 + 2 + h) * LDG];
                    }
                    const float cd = ww == 2 ? c1 : c2;                // wave 2 helps tile (1,1), wave 3 tile (3,3)
                    __builtin_amdgcn_sched_barrier(0);
                    acc_w[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(c0, c2, acc_w[0][0], 0, 0, 0);
                    acc_w[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(c1, c2, acc_w[0][1], 0, 0, 0);
                    if (s2 >= HALF) acc_w[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cd, cd, acc_w[1][0], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else if (w_role) {
            // software pipeline: the operands of step s2 + 1 are read from LDS before the MFMAs of step s2 issue, so the
            // matrix pipe never waits on an LDS round trip.  Gram form: x and y are the same activated tile (sG).
            float xa_n[TXW], yb_n[TYW];
            auto fetch = [&](int s2) {
                const int kr = 2 * s2 + h;
#pragma unroll
                for (int i = 0; i < TXW; ++i) xa_n[i] = g[kr * LDG + 32 * (tx0 + i) + r];
#pragma unroll
                for (int j = 0; j < TYW; ++j) yb_n[j] = GRAM ? g[kr * LDG + 32 * (ty0 + j) + r] : z[kr * LDZ + 32 * (ty0 + j) + r];
            };
            fetch(0);
#pragma unroll 8
            for (int s2 = 0; s2 < ROWS / 2; ++s2) {
                const int kr = 2 * s2 + h;
                float xa[TXW], yb[TYW];
#pragma unroll
                for (int i = 0; i < TXW; ++i) xa[i] = xa_n[i];
#pragma unroll
                for (int j = 0; j < TYW; ++j) {
                    yb[j] = (y_act && !GRAM && !SACT) ? fmaxf(fmaf(yb_n[j], wys[j], wyt[j]), 0.f) : yb_n[j];   // DROP: the raw tile + mask
                    if (DROP && !SACT) {
                        const uint32_t el = (uint32_t)(cur.row0 + kr) * (uint32_t)CY + (uint32_t)(32 * (ty0 + j) + r);
                        yb[j] = (mix32(el ^ a.prev.drop_seed) >= dthr) ? yb[j] * dscale : 0.f;
                    }
                }
                if (s2 + 1 < ROWS / 2) fetch(s2 + 1);
                __builtin_amdgcn_sched_barrier(0);        // keep the reads above the MFMAs (the scheduler sinks them to their use)
#pragma unroll
                for (int i = 0; i < TXW; ++i)
#pragma unroll
                    for (int j = 0; j < TYW; ++j) acc_w[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[i], yb[j], acc_w[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // the optional addend: all sixteen loads in flight before the MFMAs
            const int valid = min(ROWS, cur.row_end - cur.row0) - 32 * rt;       // rows of this tile that exist (may be <= 0)
            const int trow0 = cur.row0 + 32 * rt;
            float addv[16];
            if (ADD) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int rr = (e & 3) + 8 * (e >> 2) + 4 * h;
                    addv[e] = rr < valid ? ld_stream(&a.add[(size_t)(trow0 + rr) * CY + dcol]) : 0.f;
                }
            }
            // one accumulator, started at the per-slot bias: dependent fp32 MFMAs issue back to back at full rate (tools/mfma_probe2.hip),
            // and every add saved in the epilogue is matrix-pipe time (VALU does not overlap with fp32 MFMA here)
            f32x16 acc0;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc0[e] = c_b;
            const float *ga = g + (32 * rt + r) * LDG + 4 * h;
            const float *wb = sWt + dcol * LDG + 4 * h;
            f32x4 g0 = *reinterpret_cast<const f32x4 *>(ga), w0 = *reinterpret_cast<const f32x4 *>(wb);
            f32x4 g1 = *reinterpret_cast<const f32x4 *>(ga + 8), w1 = *reinterpret_cast<const f32x4 *>(wb + 8);
#pragma unroll 4
            for (int j = 0; j < CX / 8; j += 2) {
                const f32x4 cg0 = g0, cw0 = w0, cg1 = g1, cw1 = w1;
                if (j + 2 < CX / 8) {                       // next pair of fragments in flight behind the eight MFMAs below
                    g0 = *reinterpret_cast<const f32x4 *>(ga + 8 * j + 16);
                    w0 = *reinterpret_cast<const f32x4 *>(wb + 8 * j + 16);
                    g1 = *reinterpret_cast<const f32x4 *>(ga + 8 * j + 24);
                    w1 = *reinterpret_cast<const f32x4 *>(wb + 8 * j + 24);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(cg0[i], cw0[i], acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(cg1[i], cw1[i], acc0, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // epilogue: the sixteen z_{l-1} values of the lane come from LDS in one batch (rows past the block's end hold
            // finite filler), everything else is predicated -- no load sits between two stores
            float zv[16];
            if (YACT) {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    zv[e] = (GRAM && SACT) ? g[(32 * rt + (e & 3) + 8 * (e >> 2) + 4 * h) * LDG + dcol] : z[(32 * rt + (e & 3) + 8 * (e >> 2) + 4 * h) * LDZ + dcol];
            }
            // one base pointer per lane, compile-time row offsets: the stores need no per-element address arithmetic
            float *op = a.out + (size_t)(trow0 + 4 * h) * CY + dcol;
            auto finish = [&](auto full_tag) {
                constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int rr = (e & 3) + 8 * (e >> 2) + 4 * h;
                    const bool ok = FULL || rr < valid;
                    float v = acc0[e];
                    if (ADD) v += addv[e];
                    if (YACT) {
                        if (DROP && !SACT) {
                            const uint32_t el = (uint32_t)(trow0 + rr) * (uint32_t)CY + (uint32_t)dcol;
                            v = (mix32(el ^ a.prev.drop_seed) >= dthr) ? v * dscale : 0.f;
                        }
                        if (DROP && SACT) v *= dscale;                       // kept elements only survive the mask below
                        v = (SACT ? zv[e] : fmaf(zv[e], c_s, c_t)) > 0.f ? v : 0.f;
                        const float vs = ok ? v : 0.f;
                        s_a += vs;
                        s_b = fmaf(vs, SACT ? (zv[e] * c_undrop - c_beta) * c_invg : (zv[e] - c_m) * c_i, s_b);
                    }
                    if (ok) st_stream(v, &op[((e & 3) + 8 * (e >> 2)) * CY]);
                }
            };
            if (valid >= 32) finish(std::true_type{});
            else finish(std::false_type{});
        }
        if (more && w_role) write_lds(buf ^ 1, nxt);
        __syncthreads();
        buf ^= 1;
        cur = nxt;
        live = more;
    }

    // ---- flush: weight-gradient partial of this workgroup, bias sums, BatchNorm-backward sums ----
    if (SYM) {
        // the second-half partials of the diagonal tiles (waves 2, 3) join the first halves (waves 0, 1); the staged tiles are consumed
        float *xch = sG;                                       // [2][16][64]
        if (w_role && !sym3) {
#pragma unroll
            for (int e = 0; e < 16; ++e) xch[((ww - 2) * 16 + e) * 64 + lane] = acc_w[1][0][e];
        }
        __syncthreads();
        if (w_role && sym3) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc_w[1][0][e] += xch[(ww * 16 + e) * 64 + lane];
        }
        __syncthreads();
    }
    if (w_role && SYM) {
        float *dst = a.dWpart + (size_t)blockIdx.x * CX * CY;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            if (t >= (sym3 ? 3 : 2)) continue;
            const int ta = sym3 ? (t == 2 ? sym_b1 : sym_b0) : (t == 0 ? sym_b0 : sym_b1);
            const int tb = sym3 ? (t == 0 ? sym_b0 : sym_b1) : sym_b2;
            const f32x16 &acc = t == 0 ? acc_w[0][0] : (t == 1 ? acc_w[0][1] : acc_w[1][0]);
            const int cy = 32 * tb + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int cx = 32 * ta + (e & 3) + 8 * (e >> 2) + 4 * h;
                dst[(size_t)cx * CY + cy] = acc[e];
                if (ta != tb) dst[(size_t)cy * CY + cx] = acc[e];          // the mirrored tile
            }
        }
    } else if (w_role) {
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
    float *red = sG;                      // everything staged has been consumed (barrier at the loop's end)
    if (a.dbpart) {
        if (w_role) *reinterpret_cast<f32x4 *>(red + rsx * CX + 4 * cqx) = dbacc;
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
        // D wave (rt, dty): column dcol, two half-waves
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

int pw_bwd_blocks(int Q, int n_slots, int max_rows)
{
    // one workgroup per CU (8 waves, > 80 KB of LDS): as many as there are CUs, split evenly over the slots
    const int cus = 256;
    int bps = cus / n_slots;
    if (bps < 1) bps = 1;
    const int per_slot = (Q + n_slots - 1) / n_slots;
    const int items = per_slot * ((max_rows + FB_ITEM_ROWS - 1) / FB_ITEM_ROWS);
    return bps < items ? bps : (items > 0 ? items : 1);
}

int pw_bwd_item_rows() { return FB_ITEM_ROWS; }

bool pw_bwd_supported(int cx, int cy) { return (cx == 128 && (cy == 128 || cy == 64)) || (cx == 64 && (cy == 64 || cy == 128)); }

template <int CX, int CY, int ROWS, bool GRAM, bool YACT, bool ADD, bool DROP = false>
static int launch_fused_x(const PwBwd &a, hipStream_t st)
{
    constexpr size_t lds = (size_t)(2 * ROWS * (CX + 4) + 2 * ROWS * (CY + 4) + CY * (CX + 4)) * sizeof(float);
    static bool attr_set = false;
    auto kern = pw_bwd_kernel<CX, CY, ROWS, GRAM, YACT, ADD, DROP>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(AMPNET_E_LAUNCH, "pw_bwd_fused: hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
        attr_set = true;
    }
    char name[64];
    // one event name per kernel symbol (the rocprofv3 stats and bench.py's table then name the same thing): +gram, input without activation
    // (lin), with an addend (+add), with dropout (+drop)
    snprintf(name, sizeof(name), "pw_bwd<%d,%d>%s%s%s%s", CX, CY, a.g.act ? "+gram" : "", YACT ? "" : " lin", ADD ? "+add" : "", DROP ? "+drop" : "");
    const double rows = (double)a.rows_hint;
    const bool same = a.g.act && a.g.z == a.prev.z;
    // flops the launch executes: the symmetric Gram form multiplies 10 of the 16 tiles
    const double wflops = (GRAM && CX == 128 && CY == 128) ? 2.0 * rows * CX * CY * 10.0 / 16.0 : 2.0 * rows * CX * CY;
    ProfScope prof(name, wflops + 2.0 * rows * CX * CY, rows * 4.0 * ((a.g.dy ? CX : 0) + ((a.g.P1 || a.g.act) ? CX : 0) + (same ? 0 : CY) + CY + (a.add ? CY : 0)), st);
    hipLaunchKernelGGL(kern, dim3(a.blocks_per_slot * a.n_slots), dim3(FB_THREADS), lds, st, a);
    return check_launch("pw_bwd_kernel");
}

template <int CX, int CY, int ROWS>
static int launch_fused(const PwBwd &a, hipStream_t st)
{
    const bool gram = a.g.act != 0, yact = a.prev.s != nullptr, add = a.add != nullptr;
    if constexpr (CX == 64 && CY == 128) {
        // the head's conv_3: its input is dropout(relu(bn_2(z2)))
        if (gram || add || !yact) return fail(AMPNET_E_ARG, "pw_bwd_fused: 64 x 128 is built for an activated input without addend");
        return a.prev.drop_p > 0.f ? launch_fused_x<CX, CY, ROWS, false, true, false, true>(a, st) : launch_fused_x<CX, CY, ROWS, false, true, false, false>(a, st);
    }
    if (a.prev.drop_p > 0.f) return fail(AMPNET_E_ARG, "pw_bwd_fused: dropout only built for 64 x 128");
    if (gram) {
        if (CX != CY || !yact || add) return fail(AMPNET_E_ARG, "pw_bwd_fused: Gram form needs CX == CY, an activated input and no addend");
        if constexpr (CX == CY) return launch_fused_x<CX, CY, ROWS, true, true, false>(a, st);
    }
    if (add) {
        // only the 64 x 64 layers have an addend (conv_3: the head's d_local, feature T-Net conv_1: the bmm path)
        if constexpr (CX == 64 && CY == 64)
            return yact ? launch_fused_x<CX, CY, ROWS, false, true, true>(a, st) : launch_fused_x<CX, CY, ROWS, false, false, true>(a, st);
        return fail(AMPNET_E_ARG, "pw_bwd_fused: addend only built for 64 x 64");
    }
    return yact ? launch_fused_x<CX, CY, ROWS, false, true, false>(a, st) : launch_fused_x<CX, CY, ROWS, false, false, false>(a, st);
}

int pw_bwd_fused(const PwBwd &a_in, hipStream_t st)
{
    PwBwd a = a_in;
    {
        static const int wrap = [] { const char *e = getenv("AMPNET_PWBWD_ROWWRAP"); return e ? atoi(e) : 0; }();
        a.dbg_row_wrap = wrap;
    }
    AMPNET_REQUIRE(a.W && a.out && a.dWpart && a.win_off && a.prev.z, "pw_bwd_fused: null pointer");
    AMPNET_REQUIRE(a.g.act ? a.g.z == a.prev.z : (a.g.dy && a.g.P1), "pw_bwd_fused: dense gradient with BatchNorm constants, or the Gram form of one tensor");
    AMPNET_REQUIRE(a.g.P2 && a.g.P3 && a.g.z, "pw_bwd_fused: BatchNorm constants incomplete");
    AMPNET_REQUIRE(!a.part_a || (a.part_b && a.prev.s), "pw_bwd_fused: partial sums need the previous layer's BatchNorm");
    AMPNET_REQUIRE(a.w_win_stride == 0 || (a.items_per_block > 0 && ((a.max_rows + FB_ITEM_ROWS - 1) / FB_ITEM_ROWS) % a.items_per_block == 0 && a.Q % a.n_slots == 0),
                   "pw_bwd_fused: per-window weights need workgroups that stay inside one window");
    AMPNET_REQUIRE(a.ldw % 4 == 0 && a.Q >= 1 && a.n_slots >= 1 && a.max_rows >= 1 && a.blocks_per_slot >= 1, "pw_bwd_fused: bad shape");
    AMPNET_REQUIRE(a.prev.C == 0 || pw_bwd_supported(a.g.C, a.prev.C), "pw_bwd_fused: %d x %d not built", a.g.C, a.prev.C);
    AMPNET_REQUIRE(!a.fin_part_a || (a.fin_part_b && a.fin_parts >= a.n_slots && a.fin_rows >= 1 && a.fin_gamma && a.fin_mean && a.fin_invstd && a.fin_P1 && a.fin_P2 &&
                                     a.fin_P3 && a.fin_slot_ab && !a.g.act && a.fin_part_a != a.part_a),
                   "pw_bwd_fused: in-kernel BatchNorm-backward constants need the partials of the producer (not this launch's), the layer's statistics and outputs");
    AMPNET_REQUIRE(!a.fin_part_a || !bwd_operands_bf16(), "pw_bwd_fused: in-kernel BatchNorm-backward constants are built for the fp32 kernel");
    if (precision_split() && pw_bwd_x3_supported(a)) return pw_bwd_fused_x3(a, st);     // fp32 results from the bf16 pipe (pw_bwd_x3.hip)
    if (bwd_operands_bf16()) return pw_bwd_fused_bf16(a, st);
    AMPNET_REQUIRE(!a.g.z_bf16 && !a.prev.z_bf16, "pw_bwd_fused: bf16 activation tensors need a bf16 precision mode");      // bf16 MFMA operands (pw_bwd_bf16.hip)
    if (a.g.C == 128 && a.prev.C == 128) return launch_fused<128, 128, 32>(a, st);
    if (a.g.C == 128 && a.prev.C == 64) return launch_fused<128, 64, 64>(a, st);
    if (a.g.C == 64 && a.prev.C == 64) return launch_fused<64, 64, 64>(a, st);
    if (a.g.C == 64 && a.prev.C == 128) return launch_fused<64, 128, 32>(a, st);
    return fail(AMPNET_E_ARG, "pw_bwd_fused: %d x %d not built", a.g.C, a.prev.C);
}

}  // namespace ampnet
