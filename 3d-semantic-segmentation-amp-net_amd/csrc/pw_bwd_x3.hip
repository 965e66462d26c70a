// pw_bwd_x3.hip -- the fused backward of pw_bwd_fused.hip for the MFMA-bound 128 x 128 shapes in precision mode
// AMPNET_PRECISION_F32_SPLIT: fp32 results from the bf16 matrix pipe.
//
// Same launch interface (PwBwd), work split, roles and fp32 epilogue as the fp32 kernel; every operand of the two products is split into
// three bf16 terms x = x1 + x2 + x3 (round to nearest each time: the sum is the fp32 value exactly) and a product is the six partial
// products x1 y1, x1 y2, x2 y1, x1 y3, x2 y2, x3 y1 on v_mfma_f32_32x32x16_bf16 (each exact in fp32, the dropped terms are below
// 2^-23 |x y|), accumulated in fp32: 6 / 16 of the fp32 MFMA time.
//     sG[i][row][cx] = i-th bf16 term of g = dy * P1 + z * P2 + P3 (or of a = relu(z * P2 + P3) for the Gram form), split ONCE while staged
//     sY[i][row][cy] = i-th term of a_prev = relu(bn(z_prev))  (dense form only; the Gram form's y IS g)
//     sZ[row][cy]    = the ACTIVATED input a_prev in fp32: ReLU mask (a > 0) and zhat = (a - beta) / gamma for the data gradient's epilogue
// W waves:  dW[cx][cy] += sum_rows g[row][cx] * y[row][cy], k = rows: operands are columns of the row-major tiles (ds_read_b64_tr_b16)
// D waves:  dy_prev[row][cy] = sum_cx g[row][cx] * W[cx][cy], k = cx; a D wave owns ONE 32-column block for the whole launch, so its 8 x 3
//           weight fragments (per-slot matrix of the pooled layers, shared weights otherwise) live in 96 VGPRs -- no weight image in LDS,
//           which three images of two staged tiles already fill.
// Built for: Gram form 128 x 128 (the three pooled layers' backward) and the dense 128 x 128 layer (conv_5), activated input, no addend.
#include <type_traits>
#include "kernels.h"

namespace ampnet {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// Development probe (tools/build_stamps.sh, tools/x3_stamps.py bwd): wave 0 (W role) and wave 4 (D role) of workgroup 0 record (label, cycle)
#ifdef AMPNET_PW_STAMPS
__device__ unsigned long long g_bx_stamps[2][2048];
#define BX_STAMP(id)                                                                                             \
    do {                                                                                                         \
        if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256) && stamp_n < 2000)                        \
            g_bx_stamps[threadIdx.x >> 8][1 + stamp_n++] = ((unsigned long long)(id) << 48) | (__builtin_readcyclecounter() & 0xffffffffffffull); \
    } while (0)
#else
#define BX_STAMP(id) do { } while (0)
#endif

constexpr int X3B_THREADS = 512;
constexpr int X3B_ITEM_ROWS = 256;      // must equal pw_bwd_item_rows() (the host sizes per-window shares with it)

namespace {

// the three-term split on pairs (pw_gemm.hip has the same helpers; see there for why the conversion is an instruction)
__device__ __forceinline__ uint32_t cvt_pk(const f32x2 &v)
{
    uint32_t p;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p) : "v"(v[0]), "v"(v[1]));
    return p;
}
__device__ __forceinline__ f32x2 widen_pk(uint32_t p) { return f32x2{__builtin_bit_cast(float, p << 16), __builtin_bit_cast(float, p & 0xffff0000u)}; }
__device__ __forceinline__ void split_pair(const f32x2 &x, uint32_t &p1, uint32_t &p2, uint32_t &p3)
{
    p1 = cvt_pk(x);
    const f32x2 r = x - widen_pk(p1);
    p2 = cvt_pk(r);
    p3 = cvt_pk(r - widen_pk(p2));
}
__device__ __forceinline__ void split4(const f32x4 &v, bf16x4 &p1, bf16x4 &p2, bf16x4 &p3)
{
    uint32_t q1[2], q2[2], q3[2];
    split_pair(f32x2{v[0], v[1]}, q1[0], q2[0], q3[0]);
    split_pair(f32x2{v[2], v[3]}, q1[1], q2[1], q3[1]);
    p1 = __builtin_bit_cast(bf16x4, u32x2{q1[0], q1[1]});
    p2 = __builtin_bit_cast(bf16x4, u32x2{q2[0], q2[1]});
    p3 = __builtin_bit_cast(bf16x4, u32x2{q3[0], q3[1]});
}
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8 &p1, bf16x8 &p2, bf16x8 &p3)
{
    uint32_t q1[4], q2[4], q3[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split_pair(f32x2{v[2 * i], v[2 * i + 1]}, q1[i], q2[i], q3[i]);
    p1 = __builtin_bit_cast(bf16x8, u32x4{q1[0], q1[1], q1[2], q1[3]});
    p2 = __builtin_bit_cast(bf16x8, u32x4{q2[0], q2[1], q2[2], q2[3]});
    p3 = __builtin_bit_cast(bf16x8, u32x4{q3[0], q3[1], q3[2], q3[3]});
}

// MFMA operand whose k runs over the ROWS of a row-major bf16 tile: rows row0 .. row0 + 15, channel col0 + (lane & 31) (pw_bwd_bf16.hip)
__device__ __forceinline__ bf16x8 tr_operand(const __bf16 *tile, int ld, int row0, int col0, int lane)
{
    const int g4 = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const __bf16 *src = tile + (row0 + 8 * (g4 >> 1) + q) * ld + col0 + 16 * (g4 & 1) + 4 * p;
    typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(src));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(src + 4 * ld));
    return __builtin_shufflevector(__builtin_bit_cast(bf16x4, lo), __builtin_bit_cast(bf16x4, hi), 0, 1, 2, 3, 4, 5, 6, 7);
}

// ---- the g images are swizzled: the 16-byte granule gi of row `lrow` sits at gi ^ ((lrow >> 2) & 3) ----
// The row pitch (2 C + 64 bytes) makes the weight-gradient role's transposed reads conflict-free (four rows x 64 bytes per 32 lanes), but rows
// r, r + 4, r + 8, r + 12 then start in the same bank group, and the data-gradient role's row-major ds_read_b128 (lane = row) replayed four times:
// SQ_LDS_BANK_CONFLICT was 40 % of SQ_LDS_IDX_ACTIVE in these kernels.  The XOR moves those four rows to four different granules of an aligned
// group of four; a transposed read touches four rows of ONE such quartet (same XOR for all of them: still four disjoint 64-byte segments), the
// staging writes permute within a row.  Both readers and the writer below go through these helpers.
__device__ __forceinline__ int swz_col(int lrow, int c) { return ((((c >> 3) ^ ((lrow >> 2) & 3)) << 3) | (c & 7)); }
struct TrLane {           // per-lane constants of a swizzled transposed read (rows row0 .. row0 + 15, row0 a multiple of 16; col0 a multiple of 32)
    int lo, hi;           // element offsets of the lane's two 8-byte pieces relative to tile + row0 * ld + col0
};
__device__ __forceinline__ TrLane tr_lane(int ld, int lane)
{
    const int g4 = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int row = 8 * (g4 >> 1) + q, col = 16 * (g4 & 1) + 4 * p;
    return TrLane{row * ld + swz_col(row, col), (row + 4) * ld + swz_col(row + 4, col)};
}
__device__ __forceinline__ bf16x8 tr_operand_swz(const __bf16 *tile, int ld, int row0, int col0, const TrLane &t)
{
    const __bf16 *src = tile + row0 * ld + col0;
    typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(src + t.lo));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(src + t.hi));
    return __builtin_shufflevector(__builtin_bit_cast(bf16x4, lo), __builtin_bit_cast(bf16x4, hi), 0, 1, 2, 3, 4, 5, 6, 7);
}

// acc += x * y from the three terms of each: six exact partial products, the small ones first
__device__ __forceinline__ void mfma6(f32x16 &acc, const bf16x8 (&x)[3], const bf16x8 (&y)[3])
{
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[2], y[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], y[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[1], y[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[1], y[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], y[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], y[0], acc, 0, 0, 0);
}

}  // namespace

template <bool GRAM>
__global__ __launch_bounds__(X3B_THREADS, 1) void pw_bwd_x3_kernel(PwBwd a)
{
    constexpr int CX = 128, CY = 128, ROWS = 32;
    constexpr int LDG = CX + 32, LDY = CY + 32;     // bf16 elements per row of the operand tiles (2 C + 64 bytes: transposed reads conflict-free)
    constexpr int LDZ = CY + 8;                     // fp32 row of the activated-input tile (4 rows = 32 banks on: the two row groups of a ds_read_b32 do not collide)
    constexpr int TYN = CY / 32;
    constexpr int TXW = 2, TYW = 2;                 // W role: 2 x 2 tiles per wave
    // Gram form: the four W waves stage (their role is the lighter one: no epilogue, no stores), the D waves only multiply, mask and store --
    // with all eight staging, the D wave's products + epilogue (4600 cycles per block) and then its staging (2000) were the block's critical
    // path while the W wave waited at the barrier (in-kernel stamps, tools/x3_stamps.py).  Dense form: everybody stages (three tensors).
    constexpr int STAGE = GRAM ? X3B_THREADS / 2 : X3B_THREADS;
    constexpr int QX = CX / 4, SX = STAGE / QX;     // 32 channel quads, 8 (Gram) or 16 row groups
    constexpr int NIX = ROWS / SX;                  // 4 (Gram) or 2 quads per staging thread and tensor
    constexpr int IMG = ROWS * LDG;                 // elements of one image of one buffer
    static_assert(CX == CY && LDG == LDY && (ROWS / 32) * TYN == 4, "one dgrad tile per D wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16 *sG = reinterpret_cast<__bf16 *>(smem_raw);                                   // [2 buffers][3 terms][ROWS][LDG]
    __bf16 *sY = sG + 2 * 3 * IMG;                                                       // the same for a_prev (dense form)
    float *sZ = reinterpret_cast<float *>(sY + (GRAM ? 0 : 2 * 3 * IMG));                // [2][ROWS][LDZ] activated input, fp32
    float *red = reinterpret_cast<float *>(smem_raw);                                    // reductions alias the tiles (before / after the loop)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const TrLane trl = tr_lane(LDG, lane);
    const int slot = blockIdx.x % a.n_slots, jb = blockIdx.x / a.n_slots;

    // ---- work split: items = (window of this slot, chunk of X3B_ITEM_ROWS rows), contiguous share per workgroup ----
    const int per_slot = (a.Q - slot + a.n_slots - 1) / a.n_slots;
    const int cpw = (a.max_rows + X3B_ITEM_ROWS - 1) / X3B_ITEM_ROWS;
    const int n_items = per_slot * cpw;
    const int ipb = a.items_per_block > 0 ? a.items_per_block : (n_items + a.blocks_per_slot - 1) / a.blocks_per_slot;
    const int item_begin = min(jb * ipb, n_items), item_end = min(item_begin + ipb, n_items);

    const int cqx = tid % QX, rsx = (tid % STAGE) / QX;
    const bool stager = tid < STAGE;                 // (Gram form: waves 0 .. 3 = the W role)
    f32x4 p1 = {1.f, 1.f, 1.f, 1.f}, p2 = {0.f, 0.f, 0.f, 0.f}, p3 = {0.f, 0.f, 0.f, 0.f};
    if (!GRAM && a.fin_part_a) {
        // the BatchNorm-backward constants of this layer from the partial sums its producer left (pw_bwd_fused.hip: fin_*)
        double sa[4] = {0.0, 0.0, 0.0, 0.0}, sb[4] = {0.0, 0.0, 0.0, 0.0};
        const int per_slot_parts = (a.fin_parts - slot + a.n_slots - 1) / a.n_slots;
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
        double *redd = reinterpret_cast<double *>(smem_raw);          // [SX][CX][2]: the staging buffers are not in use yet
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            redd[((size_t)rsx * CX + 4 * cqx + c) * 2 + 0] = sa[c];
            redd[((size_t)rsx * CX + 4 * cqx + c) * 2 + 1] = sb[c];
        }
        __syncthreads();
        const double n = (double)a.fin_rows;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int ch = 4 * cqx + c;
            double A = 0.0, Bs = 0.0;
            for (int gq = 0; gq < SX; ++gq) {
                A += redd[((size_t)gq * CX + ch) * 2 + 0];
                Bs += redd[((size_t)gq * CX + ch) * 2 + 1];
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
        __syncthreads();                                             // the scratch becomes the first tile
    } else {
        if (!GRAM) p1 = *reinterpret_cast<const f32x4 *>(a.g.P1 + (size_t)slot * CX + 4 * cqx);
        p2 = *reinterpret_cast<const f32x4 *>(a.g.P2 + (size_t)slot * CX + 4 * cqx);
        p3 = *reinterpret_cast<const f32x4 *>(a.g.P3 + (size_t)slot * CX + 4 * cqx);
    }
    f32x4 ys = {1.f, 1.f, 1.f, 1.f}, yt = {0.f, 0.f, 0.f, 0.f};             // the input activation's affine, staging view (dense form)
    if (!GRAM) {
        ys = *reinterpret_cast<const f32x4 *>(a.prev.s + (size_t)slot * CY + 4 * cqx);
        yt = *reinterpret_cast<const f32x4 *>(a.prev.t + (size_t)slot * CY + 4 * cqx);
    }

    struct Pos {
        int item, row0, row_end;
    };
    auto open_item = [&](int item, Pos &p) -> bool {
        for (; item < item_end; ++item) {
            const int q = (item / cpw) * a.n_slots + slot, ch = item % cpw;
            const int rb = a.win_off[q] + ch * X3B_ITEM_ROWS;
            const int re = min(a.win_off[q + 1], rb + X3B_ITEM_ROWS);
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

    // two register sets of loads in flight (blocks n + 1 and n + 2), issued unconditionally (pw_bwd_bf16.hip explains both)
    struct Regs {
        f32x4 dy[GRAM ? 1 : NIX];
        f32x4 xz[NIX];
        f32x4 yz[GRAM ? 1 : NIX];
    };
    auto load_regs = [&](const Pos &p, Regs &R) {
#pragma unroll
        for (int i = 0; i < NIX; ++i) {
            const int row = p.row0 + rsx + SX * i;
            const size_t rr = (size_t)(row < p.row_end ? row : p.row0);
            if (!GRAM) R.dy[i] = ld_stream(reinterpret_cast<const f32x4 *>(a.g.dy + rr * CX + 4 * cqx));
            R.xz[i] = ld_stream(reinterpret_cast<const f32x4 *>(a.g.z + rr * CX + 4 * cqx));
            if (!GRAM) R.yz[i] = ld_stream(reinterpret_cast<const f32x4 *>(a.prev.z + rr * CY + 4 * cqx));
        }
    };
    f32x4 dbacc = {0.f, 0.f, 0.f, 0.f};
    const bool want_db = a.dbpart != nullptr;
    auto write_lds = [&](int buf_, const Pos &p, const Regs &R) {
        int bsel = buf_;
        asm volatile("" : "+s"(bsel));               // (see step(): keeps the addresses of the two buffers from being hoisted apart)
        __bf16 *g = sG + bsel * 3 * IMG;
        __bf16 *y = sY + bsel * 3 * IMG;
        float *z = sZ + bsel * ROWS * LDZ;
#pragma unroll
        for (int i = 0; i < NIX; ++i) {
            const int lrow = rsx + SX * i, row = p.row0 + lrow;
            const bool live_row = row < p.row_end;
            f32x4 xv = {0.f, 0.f, 0.f, 0.f};
            if (live_row) {
                if (GRAM) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) xv[c] = fmaxf(fmaf(R.xz[i][c], p2[c], p3[c]), 0.f);
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) xv[c] = fmaf(R.dy[i][c], p1[c], fmaf(R.xz[i][c], p2[c], p3[c]));
                }
                if (want_db) dbacc += xv;
            }
            bf16x4 t1, t2, t3;
            split4(xv, t1, t2, t3);
            const int gofs = lrow * LDG + swz_col(lrow, 4 * cqx);
            *reinterpret_cast<bf16x4 *>(g + gofs) = t1;
            *reinterpret_cast<bf16x4 *>(g + IMG + gofs) = t2;
            *reinterpret_cast<bf16x4 *>(g + 2 * IMG + gofs) = t3;
            if (GRAM) {
                *reinterpret_cast<f32x4 *>(z + lrow * LDZ + 4 * cqx) = xv;           // the activated input IS g here
            } else {
                f32x4 yv;
#pragma unroll
                for (int c = 0; c < 4; ++c) yv[c] = fmaxf(fmaf(R.yz[i][c], ys[c], yt[c]), 0.f);
                *reinterpret_cast<f32x4 *>(z + lrow * LDZ + 4 * cqx) = yv;           // (finite filler on rows past the end: masked in the epilogue)
                if (!live_row) yv = f32x4{0.f, 0.f, 0.f, 0.f};                       // rows past the block's end contribute nothing to dW
                split4(yv, t1, t2, t3);
                *reinterpret_cast<bf16x4 *>(y + lrow * LDY + 4 * cqx) = t1;
                *reinterpret_cast<bf16x4 *>(y + IMG + lrow * LDY + 4 * cqx) = t2;
                *reinterpret_cast<bf16x4 *>(y + 2 * IMG + lrow * LDY + 4 * cqx) = t3;
            }
        }
    };

    // ---- role state ----
    const bool w_role = wave < 4;
    const int ww = wave & 3;
    const int tx0 = (ww / 2) * TXW, ty0 = (ww % 2) * TYW;
    f32x16 acc_w[TXW][TYW];
#pragma unroll
    for (int i = 0; i < TXW; ++i)
#pragma unroll
        for (int j = 0; j < TYW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc_w[i][j][e] = 0.f;
    // D role: tile (rows 0 .. 31, columns 32 ww ..), lane = output column; its weight column block as three bf16 terms in registers
    const int dcol = 32 * ww + r;
    bf16x8 wf[CX / 16][3];
    if (!w_role) {
        const float *Wsh = a.W + (size_t)slot * a.w_slot_stride;
#pragma unroll
        for (int s2 = 0; s2 < CX / 16; ++s2) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = Wsh[(size_t)(16 * s2 + 8 * h + j) * a.ldw + dcol];
            split8(v, wf[s2][0], wf[s2][1], wf[s2][2]);
        }
    }
    const float c_b = a.bias_slot ? a.bias_slot[(size_t)slot * CY + dcol] : 0.f;
    const float c_s = a.prev.s[(size_t)slot * CY + dcol];
    const float c_t = a.prev.t[(size_t)slot * CY + dcol];
    const float c_m = a.prev_mean ? a.prev_mean[(size_t)slot * CY + dcol] : 0.0f;
    const float c_i = a.prev_invstd ? a.prev_invstd[(size_t)slot * CY + dcol] : 0.0f;
    const float c_beta = fmaf(c_m, c_s, c_t);                      // t = beta - mean s
    const float c_invg = c_s != 0.f ? c_i / c_s : 0.f;             // gamma = s / invstd; gamma = 0: P2 = 0, any finite zhat will do
    const bool do_part = a.part_a != nullptr;
    float s_a = 0.f, s_b = 0.f;

    // Positions of block n (in LDS) and of the blocks whose loads are in flight.  Gram form (8 registers per block and thread): TWO register
    // sets, blocks n + 1 and n + 2 in flight (one set caps the bytes in flight per CU below what 0.3 ms per launch needs).  Dense form
    // (dy, z and z_prev: 24 registers per block and thread, next to the D role's 96 weight registers): ONE set -- a block of rows is 96 MFMAs
    // per SIMD here, 3072 cycles, which covers a trip to HBM, and the second set spilled into the loop.  A tail position that does not exist
    // repeats the last real one: the loads are issued unconditionally (a conditional load makes every wait a vmcnt(0)), their data is not written.
    constexpr int NSETS = GRAM ? 2 : 1;
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
    if (live && stager) {
        load_regs(cur, S0);
        if (NSETS == 2) load_regs(nxt, S1);
        write_lds(0, cur, S0);
        load_regs(NSETS == 2 ? nx2 : nxt, S0);
    }
    __syncthreads();
    int buf = 0;
    // one block of rows: compute block n from LDS, write block n + 1 (register set A) into the other buffer, refill A.  The two roles run
    // the loop as two separate instantiations (role_tag): every wait is then a static count of the role's own younger memory operations.
    [[maybe_unused]] int stamp_n = 0;
    auto step = [&](Regs &A, auto role_tag) {
        constexpr bool W_ROLE = decltype(role_tag)::value;
        BX_STAMP(1);                                   // block begins
        // Gram form: the W wave stages block n + 1 FIRST and multiplies block n afterwards, the D wave multiplies first and runs its epilogue
        // afterwards -- so on every SIMD one wave's VALU phase (staging / epilogue) lies under the other's MFMA phase.  With both roles
        // multiplying first the matrix pipe idled for the second half of every block (stamps: products done at 3100 of 5850 cycles).
        Pos nx3 = nx2;
        bool more3 = false;
        if constexpr (NSETS == 2) {
            more3 = more2 && advance(nx3);
            if (!more3) nx3 = nx2;
#ifndef X3B_NO_SKEW                             // (A/B build switch, tools/ab_lib.sh: the W wave stages AFTER its products)
            if (W_ROLE && GRAM) {
                if (more1) write_lds(buf ^ 1, nxt, A);
                load_regs(nx3, A);
                BX_STAMP(3);                           // next block staged
            }
#endif
        }
        // the buffer index stays a run-time scalar: folded (the loop alternates two step bodies) every LDS address of both buffers is hoisted
        // into a register of its own, ~60 of them, and they spill into the loop
        int bsel = buf;
        asm volatile("" : "+s"(bsel));
        const __bf16 *g = sG + bsel * 3 * IMG;
        const __bf16 *y = GRAM ? g : sY + bsel * 3 * IMG;
        const float *z = sZ + bsel * ROWS * LDZ;
        if constexpr (W_ROLE && GRAM) {
            // the Gram matrix is symmetric: the four W waves own the 10 tiles of its upper triangle and mirror them at the flush -- waves 0 / 1
            // the three tiles over column blocks {0, 1} / {2, 3}: (B0,B0) (B0,B1) (B1,B1), waves 2 / 3 the tiles (0,c) (1,c) with c = 2 / 3.
            // An operand of a tile and an operand of another are the same registers: 18 / 12 MFMAs per wave and k step instead of 24.
#pragma unroll
            for (int s2 = 0; s2 < ROWS / 16; ++s2) {
                if (ww < 2) {
                    bf16x8 c0[3], c1[3];
#pragma unroll
                    for (int m = 0; m < 3; ++m) {
                        c0[m] = tr_operand_swz(g + m * IMG, LDG, 16 * s2, 32 * (2 * ww), trl);
                        c1[m] = tr_operand_swz(g + m * IMG, LDG, 16 * s2, 32 * (2 * ww + 1), trl);
                    }
                    mfma6(acc_w[0][0], c0, c0);
                    mfma6(acc_w[0][1], c0, c1);
                    if (s2 == 0) mfma6(acc_w[1][1], c1, c1);       // the second k step of this diagonal tile belongs to wave ww + 2: 30 MFMAs per wave and block
                } else {
                    bf16x8 c0[3], c1[3], c2[3];
#pragma unroll
                    for (int m = 0; m < 3; ++m) {
                        c0[m] = tr_operand_swz(g + m * IMG, LDG, 16 * s2, 0, trl);
                        c1[m] = tr_operand_swz(g + m * IMG, LDG, 16 * s2, 32, trl);
                        c2[m] = tr_operand_swz(g + m * IMG, LDG, 16 * s2, 32 * ww, trl);
                    }
                    mfma6(acc_w[0][0], c0, c2);
                    mfma6(acc_w[1][0], c1, c2);
                    // + the second k step of wave ww - 2's tile (1,1) / (3,3), whose operand this wave holds anyway (merged at the flush)
                    if (s2 == 1) {
                        if (ww == 2) mfma6(acc_w[1][1], c1, c1);
                        else mfma6(acc_w[1][1], c2, c2);
                    }
                }
            }
        } else if constexpr (W_ROLE) {
#pragma unroll
            for (int s2 = 0; s2 < ROWS / 16; ++s2) {
                bf16x8 xa[TXW][3], yb[TYW][3];
#pragma unroll
                for (int i = 0; i < TXW; ++i)
#pragma unroll
                    for (int m = 0; m < 3; ++m) xa[i][m] = tr_operand_swz(g + m * IMG, LDG, 16 * s2, 32 * (tx0 + i), trl);
#pragma unroll
                for (int j = 0; j < TYW; ++j)
#pragma unroll
                    for (int m = 0; m < 3; ++m) yb[j][m] = tr_operand(y + m * IMG, LDG, 16 * s2, 32 * (ty0 + j), lane);
#pragma unroll
                for (int i = 0; i < TXW; ++i)
#pragma unroll
                    for (int j = 0; j < TYW; ++j) mfma6(acc_w[i][j], xa[i], yb[j]);
            }
        } else {
            const int valid = min(ROWS, cur.row_end - cur.row0);           // rows of this tile that exist
            const int trow0 = cur.row0;
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = c_b;
            // the lane's row of g, swizzled (see swz_col): k steps 2 j and 2 j + 1 sit in the granules (2 * 0 + h) ^ x and (2 * 1 + h) ^ x of quartet j
            const __bf16 *ga2[2] = {g + r * LDG + 8 * ((0 + h) ^ ((r >> 2) & 3)), g + r * LDG + 8 * ((2 + h) ^ ((r >> 2) & 3))};
            // the epilogue's sixteen activations first (LDS returns in order: they are long there when the products finish), then the A
            // fragments ONE k step ahead of their six MFMAs -- left to the compiler, a step's reads sat right in front of its first MFMA and
            // the lone D wave of a SIMD ate an LDS round trip per step (1200 of its 5400 cycles per block)
            float zv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) zv[e] = z[((e & 3) + 8 * (e >> 2) + 4 * h) * LDZ + dcol];
            bf16x8 av[2][3];
#pragma unroll
            for (int m = 0; m < 3; ++m) av[0][m] = *reinterpret_cast<const bf16x8 *>(ga2[0] + m * IMG);
#pragma unroll
            for (int s2 = 0; s2 < CX / 16; ++s2) {
                if (s2 + 1 < CX / 16) {
#pragma unroll
                    for (int m = 0; m < 3; ++m) av[(s2 + 1) & 1][m] = *reinterpret_cast<const bf16x8 *>(ga2[(s2 + 1) & 1] + m * IMG + 32 * ((s2 + 1) >> 1));
                }
                __builtin_amdgcn_sched_barrier(0);
                mfma6(acc, av[s2 & 1], wf[s2]);
                __builtin_amdgcn_sched_barrier(0);
            }
            float *op = a.out + (size_t)(trow0 + 4 * h) * CY + dcol;
            auto finish = [&](auto full_tag) {
                constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int rr = (e & 3) + 8 * (e >> 2) + 4 * h;
                    const bool ok = FULL || rr < valid;
                    float v = acc[e];
                    v = zv[e] > 0.f ? v : 0.f;                               // ReLU mask of the layer's input: a > 0
                    const float vs = ok ? v : 0.f;
                    s_a += vs;
                    s_b = fmaf(vs, (zv[e] - c_beta) * c_invg, s_b);          // zhat = (a - beta) / gamma where the mask holds
                    if (ok) st_stream(v, &op[((e & 3) + 8 * (e >> 2)) * CY]);
                }
            };
            if (valid >= 32) finish(std::true_type{});
            else finish(std::false_type{});
        }
        BX_STAMP(2);                                   // the role's products (and the D role's epilogue) are issued
        if constexpr (NSETS == 2) {
#ifdef X3B_NO_SKEW
            if (W_ROLE || !GRAM) {
#else
            if (!GRAM) {
#endif
                if (more1) write_lds(buf ^ 1, nxt, A);
                load_regs(nx3, A);
            }
            __syncthreads();
            BX_STAMP(4);                               // barrier passed
            buf ^= 1;
            cur = nxt;
            nxt = nx2;
            nx2 = nx3;
            live = more1;
            more1 = more2;
            more2 = more3;
        } else {
            if (more1) write_lds(buf ^ 1, nxt, A);
            load_regs(nx2, A);                       // block n + 2 (or a repeat of the last real one)
            __syncthreads();
            buf ^= 1;
            cur = nxt;
            nxt = nx2;
            live = more1;
            more1 = more2;
            more2 = more1 && advance(nx2);
            if (!more2) nx2 = nxt;
        }
    };
    if (w_role) {
        while (live) {
            step(NSETS == 2 ? S1 : S0, std::true_type{});
            if (!live) break;
            step(S0, std::true_type{});
        }
    } else {
        // (s_setprio 2 for this role -- the critical path of a block -- was A/B'd on one box: +1.5 %, dropped)
        while (live) {
            step(NSETS == 2 ? S1 : S0, std::false_type{});
            if (!live) break;
            step(S0, std::false_type{});
        }
    }

#ifdef AMPNET_PW_STAMPS
    if (GRAM && blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) g_bx_stamps[threadIdx.x >> 8][0] = (unsigned long long)stamp_n;
#endif
    // ---- flush: weight-gradient partial of this workgroup, bias sums, BatchNorm-backward sums ----
    if (GRAM) {
        // the halves of the two shared diagonal tiles meet: waves 2 / 3 hand theirs to waves 0 / 1 (fixed order: bitwise reproducible);
        // the staged tiles are consumed, the scratch sits past the bias-sum scratch used below
        float *xch = reinterpret_cast<float *>(smem_raw) + 4096;       // [2][16][64]
        if (w_role && ww >= 2) {
#pragma unroll
            for (int e = 0; e < 16; ++e) xch[((ww - 2) * 16 + e) * 64 + lane] = acc_w[1][1][e];
        }
        __syncthreads();
        if (w_role && ww < 2) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc_w[1][1][e] += xch[(ww * 16 + e) * 64 + lane];
        }
    }
    if (w_role && GRAM) {
        // the upper-triangle tiles and their mirror images (see the W role above)
        float *dst = a.dWpart + (size_t)blockIdx.x * CX * CY;
        auto put = [&](const f32x16 &acc, int ta, int tb) {
            const int cy = 32 * tb + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int cx = 32 * ta + (e & 3) + 8 * (e >> 2) + 4 * h;
                dst[(size_t)cx * CY + cy] = acc[e];
                if (ta != tb) dst[(size_t)cy * CY + cx] = acc[e];
            }
        };
        if (ww < 2) {
            put(acc_w[0][0], 2 * ww, 2 * ww);
            put(acc_w[0][1], 2 * ww, 2 * ww + 1);
            put(acc_w[1][1], 2 * ww + 1, 2 * ww + 1);
        } else {
            put(acc_w[0][0], 0, ww);
            put(acc_w[1][0], 1, ww);
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
    if (a.dbpart) {
        if (stager) *reinterpret_cast<f32x4 *>(red + rsx * CX + 4 * cqx) = dbacc;
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
            red[dcol * 2 + 0] = s_a + oa;
            red[dcol * 2 + 1] = s_b + ob;
        }
        __syncthreads();
        if (tid < CY) {
            a.part_a[(size_t)blockIdx.x * CY + tid] = red[tid * 2 + 0];
            a.part_b[(size_t)blockIdx.x * CY + tid] = red[tid * 2 + 1];
        }
    }
}

template <bool GRAM>
static int launch_x3(const PwBwd &a, hipStream_t st)
{
    constexpr size_t img = (size_t)32 * 160 * 2;
    constexpr size_t lds = 2 * 3 * img * (GRAM ? 1 : 2) + (size_t)2 * 32 * 136 * 4;
    static_assert(lds <= 160 * 1024 && lds >= (size_t)16 * 128 * 2 * 8, "LDS budget (tiles; the prologue's double scratch aliases them)");
    static bool attr_set = false;
    auto kern = pw_bwd_x3_kernel<GRAM>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(AMPNET_E_LAUNCH, "pw_bwd_x3: hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
        attr_set = true;
    }
    char name[64];
    snprintf(name, sizeof(name), "pw_bwd<128,128>%s x3", GRAM ? "+gram" : "");
    const double rows = (double)a.rows_hint;
    // algorithmic flops the launch stands for (each executed as six bf16 partial products): the symmetric Gram form multiplies 10 of its 16 tiles
    const double wflops = GRAM ? 2.0 * rows * 128 * 128 * 10.0 / 16.0 : 2.0 * rows * 128 * 128;
    ProfScope prof(name, wflops + 2.0 * rows * 128 * 128, rows * 4.0 * ((a.g.dy ? 128 : 0) + 128 + (GRAM ? 0 : 128) + 128), st);
    hipLaunchKernelGGL(kern, dim3(a.blocks_per_slot * a.n_slots), dim3(X3B_THREADS), lds, st, a);
    return check_launch("pw_bwd_x3_kernel");
}

// ---- 128 x 64 (the four launches behind the feature T-Net / conv_4: dy [R, 128], z [R, 128], z_prev [R, 64] -> out [R, 64]) -------------------
// The fp32 kernel of this shape was the step's dominant one at 0.57 matrix-pipe busy and 3.9 TB/s (neither roof); in split arithmetic its
// MFMA time is 3 / 8, which leaves HBM.  Blocks of 32 rows (three images of two staged tiles of 64 rows would not fit LDS) give the data
// gradient TWO 32 x 32 tiles and the weight gradient eight, so the eight waves take three roles, one MFMA wave and one VALU wave per SIMD:
//     waves 0, 1, 6, 7 (S): stage -- loads two blocks ahead, g = dy P1 + z P2 + P3 and a = relu(bn(z_prev)) in fp32, the three-term
//                            split, LDS writes; bias / column sums of g
//     waves 2, 3       (W): dW[cx][cy] += sum_rows g a, four 32 x 32 tiles each (k = rows, transposed LDS reads): 48 MFMAs per block
//     waves 4, 5       (D): out[row][32 d ..] = mask (g W), weight column block as 8 x 3 bf16 fragments in registers: 48 MFMAs per block
// A workgroup's waves are dealt to the SIMDs 0, 1, 2, 3, 0, 1, 2, 3: every SIMD holds one MFMA wave and one staging wave.
template <int CX, int CY, bool YACT, bool DROP, bool ADD>
__global__ __launch_bounds__(X3B_THREADS, 1) void pw_bwd_x3n_kernel(PwBwd a)
{
    static_assert((CX == 128 && CY == 64) || (CX == 64 && CY == 128) || (CX == 64 && CY == 64), "48 + 48 MFMAs per block (24 + 24 at 64 x 64: HBM-bound)");
    static_assert(!ADD || (CX == 64 && CY == 64), "the addend exists on the 64 x 64 launches");
    static_assert(!DROP || YACT, "dropout sits on an activated input");
    constexpr int ROWS = 32;
    constexpr int LDG = CX + 32, LDY = CY + 32;     // bf16 rows of the operand tiles (2 C + 64 bytes: transposed reads conflict-free)
    constexpr int LDZ = CY + 8;                     // fp32 row of the (activated) input tile (4 rows = 32 banks on)
    constexpr int IMGG = ROWS * LDG, IMGY = ROWS * LDY;
    constexpr int NS = 256;                         // staging threads
    constexpr int QX = CX / 4, QY = CY / 4;         // channel quads per row
    constexpr int SX = NS / QX, SY = NS / QY;       // row groups
    constexpr int NIX = ROWS / SX, NIY = ROWS / SY; // quads per staging thread (4 + 2 at 128 x 64, 2 + 4 at 64 x 128)
    constexpr int FG = X3B_THREADS / QX;            // prologue: groups of channel quads over the whole workgroup
    constexpr int TXW = CX / 64, TYW = CY / 32;     // W role: x blocks per wave, y blocks (all of them): 4 tiles per wave
    constexpr int DB = CY / 64, KS = CX / 16;       // D role: column blocks per wave, k steps: DB * KS * 6 = 48 MFMAs per block
    constexpr size_t BUF = (size_t)3 * IMGG * 2 + (size_t)3 * IMGY * 2 + (size_t)ROWS * LDZ * 4;      // bytes of one staged block
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *red = reinterpret_cast<float *>(smem_raw);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const TrLane trl = tr_lane(LDG, lane);
    const int slot = blockIdx.x % a.n_slots, jb = blockIdx.x / a.n_slots;
    const bool s_role = wave < 2 || wave >= 6, w_role = wave == 2 || wave == 3, d_role = wave == 4 || wave == 5;
    const int stid = ((wave < 2 ? wave : wave - 4) << 6) | lane;       // 0 .. 255 over the four staging waves
    const int cqx = stid % QX, rsx = stid / QX, cqy = stid % QY, rsy = stid / QY;

    const int per_slot = (a.Q - slot + a.n_slots - 1) / a.n_slots;
    const int cpw = (a.max_rows + X3B_ITEM_ROWS - 1) / X3B_ITEM_ROWS;
    const int n_items = per_slot * cpw;
    const int ipb = a.items_per_block > 0 ? a.items_per_block : (n_items + a.blocks_per_slot - 1) / a.blocks_per_slot;
    const int item_begin = min(jb * ipb, n_items), item_end = min(item_begin + ipb, n_items);

    // ---- BatchNorm-backward constants of the layer g belongs to (formed here from the producer's partial sums, or read) ----
    f32x4 p1 = {1.f, 1.f, 1.f, 1.f}, p2 = {0.f, 0.f, 0.f, 0.f}, p3 = {0.f, 0.f, 0.f, 0.f};
    if (a.fin_part_a) {
        // every thread takes part (512 = FG groups of QX channel quads), pw_bwd_fused.hip: fin_*
        const int fq = tid % QX, fg = tid / QX;
        double sa[4] = {0.0, 0.0, 0.0, 0.0}, sb[4] = {0.0, 0.0, 0.0, 0.0};
        const int per_slot_parts = (a.fin_parts - slot + a.n_slots - 1) / a.n_slots;
        for (int k0 = fg; k0 < per_slot_parts; k0 += FG * 8) {
            f32x4 va[8], vb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + FG * u;
                const size_t o = (size_t)(slot + (k < per_slot_parts ? k : 0) * a.n_slots) * CX + 4 * fq;
                va[u] = *reinterpret_cast<const f32x4 *>(a.fin_part_a + o);
                vb[u] = *reinterpret_cast<const f32x4 *>(a.fin_part_b + o);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (k0 + FG * u < per_slot_parts) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        sa[c] += (double)va[u][c];
                        sb[c] += (double)vb[u][c];
                    }
                }
            }
        }
        double *redd = reinterpret_cast<double *>(smem_raw);          // [FG][CX][2] = 32 KB: the staging buffers are not in use yet
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            redd[((size_t)fg * CX + 4 * fq + c) * 2 + 0] = sa[c];
            redd[((size_t)fg * CX + 4 * fq + c) * 2 + 1] = sb[c];
        }
        __syncthreads();
        const double n = (double)a.fin_rows;
        f32x4 q1, q2, q3;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int ch = 4 * cqx + c;
            double A = 0.0, Bs = 0.0;
            for (int gq = 0; gq < FG; ++gq) {
                A += redd[((size_t)gq * CX + ch) * 2 + 0];
                Bs += redd[((size_t)gq * CX + ch) * 2 + 1];
            }
            const size_t o = (size_t)slot * CX + ch;
            const double invstd = a.fin_invstd[o], mean = a.fin_mean[o];
            const double sc = (double)a.fin_gamma[ch] * invstd;
            const double qq = -sc * invstd * Bs / n;
            q1[c] = (float)sc;
            q2[c] = (float)qq;
            q3[c] = (float)(-sc * A / n - qq * mean);
            if (jb == 0 && tid < QX) {                          // one writer per slot (threads 0 .. 31 = wave 0: cqx = tid)
                a.fin_slot_ab[o * 2 + 0] = (float)A;
                a.fin_slot_ab[o * 2 + 1] = (float)Bs;
            }
        }
        p1 = q1; p2 = q2; p3 = q3;
        if (jb == 0 && tid < QX) {
            *reinterpret_cast<f32x4 *>(a.fin_P1 + (size_t)slot * CX + 4 * cqx) = p1;
            *reinterpret_cast<f32x4 *>(a.fin_P2 + (size_t)slot * CX + 4 * cqx) = p2;
            *reinterpret_cast<f32x4 *>(a.fin_P3 + (size_t)slot * CX + 4 * cqx) = p3;
        }
        __syncthreads();
    } else {
        p1 = *reinterpret_cast<const f32x4 *>(a.g.P1 + (size_t)slot * CX + 4 * cqx);
        p2 = *reinterpret_cast<const f32x4 *>(a.g.P2 + (size_t)slot * CX + 4 * cqx);
        p3 = *reinterpret_cast<const f32x4 *>(a.g.P3 + (size_t)slot * CX + 4 * cqx);
    }
    f32x4 ys = {1.f, 1.f, 1.f, 1.f}, yt = {0.f, 0.f, 0.f, 0.f};
    if (YACT) {
        ys = *reinterpret_cast<const f32x4 *>(a.prev.s + (size_t)slot * CY + 4 * cqy);
        yt = *reinterpret_cast<const f32x4 *>(a.prev.t + (size_t)slot * CY + 4 * cqy);
    }

    struct Pos {
        int item, row0, row_end;
    };
    auto open_item = [&](int item, Pos &p) -> bool {
        for (; item < item_end; ++item) {
            const int q = (item / cpw) * a.n_slots + slot, ch = item % cpw;
            const int rb = a.win_off[q] + ch * X3B_ITEM_ROWS;
            const int re = min(a.win_off[q + 1], rb + X3B_ITEM_ROWS);
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

    struct Regs {
        f32x4 dy[NIX], xz[NIX], yz[NIY];
    };
    auto load_regs = [&](const Pos &p, Regs &R) {
#pragma unroll
        for (int i = 0; i < NIX; ++i) {
            const int row = p.row0 + rsx + SX * i;
            const size_t rr = (size_t)(row < p.row_end ? row : p.row0);
            R.dy[i] = ld_stream(reinterpret_cast<const f32x4 *>(a.g.dy + rr * CX + 4 * cqx));
            R.xz[i] = ld_stream(reinterpret_cast<const f32x4 *>(a.g.z + rr * CX + 4 * cqx));
        }
#pragma unroll
        for (int i = 0; i < NIY; ++i) {
            const int row = p.row0 + rsy + SY * i;
            const size_t rr = (size_t)(row < p.row_end ? row : p.row0);
            R.yz[i] = ld_stream(reinterpret_cast<const f32x4 *>(a.prev.z + rr * CY + 4 * cqy));
        }
    };
    f32x4 dbacc = {0.f, 0.f, 0.f, 0.f};
    const bool want_db = a.dbpart != nullptr;
    const uint32_t dthr = drop_threshold(a.prev.drop_p);
    const float dscale = DROP ? 1.0f / (1.0f - a.prev.drop_p) : 1.0f;
    auto write_lds = [&](int buf_, const Pos &p, const Regs &R) {
        int bsel = buf_;
        asm volatile("" : "+s"(bsel));               // keeps the addresses of the two buffers from being hoisted apart (see pw_bwd_x3_kernel)
        __bf16 *g = reinterpret_cast<__bf16 *>(smem_raw + bsel * BUF);
        __bf16 *y = g + 3 * IMGG;
        float *z = reinterpret_cast<float *>(y + 3 * IMGY);
#pragma unroll
        for (int i = 0; i < NIX; ++i) {
            const int lrow = rsx + SX * i, row = p.row0 + lrow;
            f32x4 xv = {0.f, 0.f, 0.f, 0.f};
            if (row < p.row_end) {
#pragma unroll
                for (int c = 0; c < 4; ++c) xv[c] = fmaf(R.dy[i][c], p1[c], fmaf(R.xz[i][c], p2[c], p3[c]));
                if (want_db) dbacc += xv;
            }
            bf16x4 t1, t2, t3;
            split4(xv, t1, t2, t3);
            const int gofs = lrow * LDG + swz_col(lrow, 4 * cqx);
            *reinterpret_cast<bf16x4 *>(g + gofs) = t1;
            *reinterpret_cast<bf16x4 *>(g + IMGG + gofs) = t2;
            *reinterpret_cast<bf16x4 *>(g + 2 * IMGG + gofs) = t3;
        }
#pragma unroll
        for (int i = 0; i < NIY; ++i) {
            const int lrow = rsy + SY * i, row = p.row0 + lrow;
            f32x4 yv = R.yz[i];
            if (YACT) {
#pragma unroll
                for (int c = 0; c < 4; ++c) yv[c] = fmaxf(fmaf(yv[c], ys[c], yt[c]), 0.f);
                if (DROP) {                  // the layer's input went through dropout: one hash per staged element (pw_bwd_fused.hip)
                    const uint32_t e0 = (uint32_t)row * (uint32_t)CY + (uint32_t)(4 * cqy);
#pragma unroll
                    for (int c = 0; c < 4; ++c) yv[c] = (mix32((e0 + c) ^ a.prev.drop_seed) >= dthr) ? yv[c] * dscale : 0.f;
                }
                *reinterpret_cast<f32x4 *>(z + lrow * LDZ + 4 * cqy) = yv;           // (finite filler on rows past the end: masked in the epilogue)
            }
            if (!(row < p.row_end)) yv = f32x4{0.f, 0.f, 0.f, 0.f};               // rows past the block's end contribute nothing to dW
            bf16x4 t1, t2, t3;
            split4(yv, t1, t2, t3);
            *reinterpret_cast<bf16x4 *>(y + lrow * LDY + 4 * cqy) = t1;
            *reinterpret_cast<bf16x4 *>(y + IMGY + lrow * LDY + 4 * cqy) = t2;
            *reinterpret_cast<bf16x4 *>(y + 2 * IMGY + lrow * LDY + 4 * cqy) = t3;
        }
    };

    // ---- role state ----
    const int wi = wave & 1;                               // W: x blocks TXW wi .. of cx, every y block;  D: column blocks DB wi .. of cy
    f32x16 acc_w[TXW][TYW];
#pragma unroll
    for (int i = 0; i < TXW; ++i)
#pragma unroll
        for (int j = 0; j < TYW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc_w[i][j][e] = 0.f;
    int dcol[DB];
    bf16x8 wf[DB][KS][3];                                  // the wave's column blocks of the weight: 96 VGPRs at either shape
    float c_b[DB], c_beta[DB], c_invg[DB];
#pragma unroll
    for (int bq = 0; bq < DB; ++bq) {
        dcol[bq] = 32 * (DB * wi + bq) + r;
        c_b[bq] = a.bias_slot ? a.bias_slot[(size_t)slot * CY + dcol[bq]] : 0.f;
        const float c_s = YACT ? a.prev.s[(size_t)slot * CY + dcol[bq]] : 1.0f;
        const float c_t = YACT ? a.prev.t[(size_t)slot * CY + dcol[bq]] : 0.0f;
        const float c_m = (YACT && a.prev_mean) ? a.prev_mean[(size_t)slot * CY + dcol[bq]] : 0.0f;
        const float c_i = (YACT && a.prev_invstd) ? a.prev_invstd[(size_t)slot * CY + dcol[bq]] : 0.0f;
        c_beta[bq] = fmaf(c_m, c_s, c_t);
        c_invg[bq] = c_s != 0.f ? c_i / c_s : 0.f;
    }
    const float c_undrop = DROP ? 1.0f - a.prev.drop_p : 1.0f;      // a = a' (1 - p) for a kept element
    const bool do_part = a.part_a != nullptr;
    float s_a[DB], s_b[DB];
#pragma unroll
    for (int bq = 0; bq < DB; ++bq) s_a[bq] = s_b[bq] = 0.f;

    // blocks n (in LDS), n + 1 and n + 2 (in the stagers' two register sets); a tail position repeats the last real one (unconditional loads)
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
    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): every constant has landed before the loop
    if (live && s_role) {
        load_regs(cur, S0);
        load_regs(nxt, S1);
        write_lds(0, cur, S0);
        load_regs(nx2, S0);
    }
    __syncthreads();
    int buf = 0;
    auto step = [&](Regs &A, auto role_tag) {
        constexpr int ROLE = decltype(role_tag)::value;          // 0 stage, 1 weight gradient, 2 data gradient
        int bsel = buf;
        asm volatile("" : "+s"(bsel));
        const __bf16 *g = reinterpret_cast<const __bf16 *>(smem_raw + bsel * BUF);
        const __bf16 *y = g + 3 * IMGG;
        const float *z = reinterpret_cast<const float *>(y + 3 * IMGY);
        Pos nx3 = nx2;
        const bool more3 = more2 && advance(nx3);
        if (!more3) nx3 = nx2;
        if constexpr (ROLE == 0) {
            if (more1) write_lds(buf ^ 1, nxt, A);
            load_regs(nx3, A);
        } else if constexpr (ROLE == 1) {
#pragma unroll
            for (int s2 = 0; s2 < ROWS / 16; ++s2) {
                bf16x8 xa[TXW][3], yb[TYW][3];
#pragma unroll
                for (int i = 0; i < TXW; ++i)
#pragma unroll
                    for (int m = 0; m < 3; ++m) xa[i][m] = tr_operand_swz(g + m * IMGG, LDG, 16 * s2, 32 * (TXW * wi + i), trl);
#pragma unroll
                for (int j = 0; j < TYW; ++j)
#pragma unroll
                    for (int m = 0; m < 3; ++m) yb[j][m] = tr_operand(y + m * IMGY, LDY, 16 * s2, 32 * j, lane);
#pragma unroll
                for (int i = 0; i < TXW; ++i)
#pragma unroll
                    for (int j = 0; j < TYW; ++j) mfma6(acc_w[i][j], xa[i], yb[j]);
            }
        } else {
            const int valid = min(ROWS, cur.row_end - cur.row0);
            const int trow0 = cur.row0;
            f32x16 acc[DB];
            // one column block: the sixteen activations of its epilogue are requested up front (LDS returns in order: long there when the products
            // finish); two: after the products, the second block's under the first block's epilogue
            float zv[2][16];
#pragma unroll
            for (int bq = 0; bq < DB; ++bq) {
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[bq][e] = c_b[bq];
            }
            if (YACT && DB == 1) {
#pragma unroll
                for (int e = 0; e < 16; ++e) zv[0][e] = z[((e & 3) + 8 * (e >> 2) + 4 * h) * LDZ + dcol[0]];
            }
            float addv[16];                       // ADD: the gradient that joins this layer's input from another branch (DB == 1 there)
            if (ADD) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int rr = (e & 3) + 8 * (e >> 2) + 4 * h;
                    addv[e] = rr < valid ? ld_stream(&a.add[(size_t)(trow0 + rr) * CY + dcol[0]]) : 0.f;
                }
            }
            const __bf16 *ga2[2] = {g + r * LDG + 8 * ((0 + h) ^ ((r >> 2) & 3)), g + r * LDG + 8 * ((2 + h) ^ ((r >> 2) & 3))};     // swizzled row (swz_col)
            // A fragments one k step ahead of their MFMAs (two register sets) when one column block leaves room for them; with two column
            // blocks (12 MFMAs per step cover most of an LDS round trip, and 24 more VGPRs would spill into this loop) one set, read per step
            constexpr int AVS = DB == 1 ? 2 : 1;
            bf16x8 av[AVS][3];
#pragma unroll
            for (int m = 0; m < 3; ++m) av[0][m] = *reinterpret_cast<const bf16x8 *>(ga2[0] + m * IMGG);
#pragma unroll
            for (int s2 = 0; s2 < KS; ++s2) {
                if (AVS == 2 && s2 + 1 < KS) {
#pragma unroll
                    for (int m = 0; m < 3; ++m) av[(s2 + 1) % AVS][m] = *reinterpret_cast<const bf16x8 *>(ga2[(s2 + 1) & 1] + m * IMGG + 32 * ((s2 + 1) >> 1));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int bq = 0; bq < DB; ++bq) mfma6(acc[bq], av[s2 % AVS], wf[bq][s2]);
                if (AVS == 1 && s2 + 1 < KS) {
#pragma unroll
                    for (int m = 0; m < 3; ++m) av[0][m] = *reinterpret_cast<const bf16x8 *>(ga2[(s2 + 1) & 1] + m * IMGG + 32 * ((s2 + 1) >> 1));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (YACT && DB > 1) {             // two column blocks: no room for the activations next to 96 weight VGPRs + two accumulators during the products
#pragma unroll
                for (int e = 0; e < 16; ++e) zv[0][e] = z[((e & 3) + 8 * (e >> 2) + 4 * h) * LDZ + dcol[0]];
            }
#pragma unroll
            for (int bq = 0; bq < DB; ++bq) {
                if (YACT && bq + 1 < DB) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) zv[(bq + 1) & 1][e] = z[((e & 3) + 8 * (e >> 2) + 4 * h) * LDZ + dcol[bq + 1]];
                }
                float *op = a.out + (size_t)(trow0 + 4 * h) * CY + dcol[bq];
                auto finish = [&](auto full_tag) {
                    constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int rr = (e & 3) + 8 * (e >> 2) + 4 * h;
                        const bool ok = FULL || rr < valid;
                        float v = acc[bq][e];
                        if (ADD) v += addv[e];
                        if (YACT) {
                            if (DROP) v *= dscale;                               // kept elements only survive the mask below
                            v = zv[bq & 1][e] > 0.f ? v : 0.f;                   // ReLU (and dropout) mask of the layer's input: a > 0
                            const float vs = ok ? v : 0.f;
                            s_a[bq] += vs;
                            s_b[bq] = fmaf(vs, (zv[bq & 1][e] * c_undrop - c_beta[bq]) * c_invg[bq], s_b[bq]);      // zhat = (a - beta) / gamma where the mask holds
                        }
                        if (ok) st_stream(v, &op[((e & 3) + 8 * (e >> 2)) * CY]);
                    }
                };
                if (valid >= 32) finish(std::true_type{});
                else finish(std::false_type{});
            }
        }
        __syncthreads();
        buf ^= 1;
        cur = nxt;
        nxt = nx2;
        nx2 = nx3;
        live = more1;
        more1 = more2;
        more2 = more3;
    };
    if (s_role) {
        while (live) {
            step(S1, std::integral_constant<int, 0>{});
            if (!live) break;
            step(S0, std::integral_constant<int, 0>{});
        }
    } else if (w_role) {
        while (live) step(S0, std::integral_constant<int, 1>{});
    } else {
        // the weight fragments are fetched HERE, after the staging waves' priming block above: loaded before it they were live across it
        // (96 VGPRs next to two staging register sets) and the allocator spilled one of them into this loop
        const float *Wsh = a.W + (size_t)slot * a.w_slot_stride;
        const float *Tq = nullptr;
        if (a.w_win_stride != 0) {
            // per-window matrix T[pidx][cy][cx] (the bmm transform: out[row][cy] = sum_cx g[row][cx] T[cy][cx]); the host keeps every workgroup
            // inside one window (pw_bwd_fused.hip)
            const int bi = item_begin / cpw;
            const int pidx = a.perwin_slot_major ? slot * (a.Q / a.n_slots) + bi : bi * a.n_slots + slot;
            Tq = a.W + (size_t)pidx * a.w_win_stride;
        }
#pragma unroll
        for (int bq = 0; bq < DB; ++bq)
#pragma unroll
            for (int s2 = 0; s2 < KS; ++s2) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    v[j] = Tq ? Tq[(size_t)dcol[bq] * CX + 16 * s2 + 8 * h + j] : Wsh[(size_t)(16 * s2 + 8 * h + j) * a.ldw + dcol[bq]];
                split8(v, wf[bq][s2][0], wf[bq][s2][1], wf[bq][s2][2]);
            }
        while (live) step(S0, std::integral_constant<int, 2>{});
    }

    // ---- flush ----
    if (w_role) {
        float *dst = a.dWpart + (size_t)blockIdx.x * CX * CY;
#pragma unroll
        for (int i = 0; i < TXW; ++i)
#pragma unroll
            for (int j = 0; j < TYW; ++j) {
                const int cy = 32 * j + r;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int cx = 32 * (TXW * wi + i) + (e & 3) + 8 * (e >> 2) + 4 * h;
                    dst[(size_t)cx * CY + cy] = acc_w[i][j][e];
                }
            }
    }
    if (a.dbpart) {
        if (s_role) *reinterpret_cast<f32x4 *>(red + rsx * CX + 4 * cqx) = dbacc;
        __syncthreads();
        if (tid < CX) {
            float sum = 0.f;
#pragma unroll
            for (int gi = 0; gi < SX; ++gi) sum += red[gi * CX + tid];
            a.dbpart[(size_t)blockIdx.x * CX + tid] = sum;
        }
        __syncthreads();
    }
    if (do_part) {
#pragma unroll
        for (int bq = 0; bq < DB; ++bq) {
            const float oa = __shfl_xor(s_a[bq], 32), ob = __shfl_xor(s_b[bq], 32);
            if (d_role && h == 0) {
                red[dcol[bq] * 2 + 0] = s_a[bq] + oa;
                red[dcol[bq] * 2 + 1] = s_b[bq] + ob;
            }
        }
        __syncthreads();
        if (tid < CY) {
            a.part_a[(size_t)blockIdx.x * CY + tid] = red[tid * 2 + 0];
            a.part_b[(size_t)blockIdx.x * CY + tid] = red[tid * 2 + 1];
        }
    }
}

template <int CX, int CY, bool YACT, bool DROP, bool ADD = false>
static int launch_x3n(const PwBwd &a, hipStream_t st)
{
    constexpr size_t buf = (size_t)3 * 32 * (CX + 32) * 2 + (size_t)3 * 32 * (CY + 32) * 2 + (size_t)32 * (CY + 8) * 4;
    constexpr size_t lds = 2 * buf;
    static_assert(lds <= 160 * 1024 && lds >= (size_t)16 * 128 * 2 * 8, "LDS budget (tiles; the prologue's double scratch aliases them)");
    static bool attr_set = false;
    auto kern = pw_bwd_x3n_kernel<CX, CY, YACT, DROP, ADD>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(AMPNET_E_LAUNCH, "pw_bwd_x3n: hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
        attr_set = true;
    }
    char name[64];
    snprintf(name, sizeof(name), "pw_bwd<%d,%d>%s%s%s x3", CX, CY, YACT ? "" : " lin", ADD ? "+add" : "", DROP ? "+drop" : "");
    const double rows = (double)a.rows_hint;
    ProfScope prof(name, 4.0 * rows * CX * CY, rows * 4.0 * (2 * CX + 2 * CY + (ADD ? CY : 0)), st);
    hipLaunchKernelGGL(kern, dim3(a.blocks_per_slot * a.n_slots), dim3(X3B_THREADS), lds, st, a);
    return check_launch("pw_bwd_x3n_kernel");
}

// the shapes the split backward is built for; pw_bwd_fused() asks before it dispatches here
bool pw_bwd_x3_supported(const PwBwd &a)
{
    const bool gram = a.g.act != 0;
    if (a.g.z_bf16 || a.prev.z_bf16) return false;
    if (a.w_win_stride && !(a.g.C == 64 && a.prev.C == 64 && a.items_per_block > 0)) return false;      // per-window weights: the bmm backward only
    const bool dense_g = !gram && a.g.dy != nullptr && a.g.z != nullptr && (a.g.P1 != nullptr || a.fin_part_a != nullptr);
    // 64 -> 64 (conv_2 / conv_3 of the encoder and the feature T-Net's conv_1; some with the addend of a joining branch): the built variants
    if (a.g.C == 64 && a.prev.C == 64) return dense_g && !(a.prev.drop_p > 0.f) && (a.prev.s != nullptr || a.add != nullptr);
    if (a.add) return false;
    // 64 -> 128 (the head's conv_3: its input is an activation that went through dropout)
    if (a.g.C == 64 && a.prev.C == 128) return dense_g && a.prev.s != nullptr && a.prev.drop_p < 1.f;
    if (a.prev.drop_p > 0.f) return false;
    if (a.g.C == 128 && a.prev.C == 128) return a.prev.s != nullptr && (gram ? a.g.z == a.prev.z : (a.g.dy != nullptr));
    if (a.g.C == 128 && a.prev.C == 64) return dense_g;
    return false;
}

// same argument contract as pw_bwd_fused (it validates before dispatching here)
int pw_bwd_fused_x3(const PwBwd &a, hipStream_t st)
{
    static_assert(X3B_ITEM_ROWS == 256, "item size shared with pw_bwd_fused.hip");
    AMPNET_REQUIRE(pw_bwd_x3_supported(a), "pw_bwd_x3: shape not built");
    if (a.g.C == 64 && a.prev.C == 64) {
        if (a.prev.s) return a.add ? launch_x3n<64, 64, true, false, true>(a, st) : launch_x3n<64, 64, true, false, false>(a, st);
        return launch_x3n<64, 64, false, false, true>(a, st);
    }
    if (a.g.C == 64) return a.prev.drop_p > 0.f ? launch_x3n<64, 128, true, true>(a, st) : launch_x3n<64, 128, true, false>(a, st);
    if (a.prev.C == 64) return a.prev.s ? launch_x3n<128, 64, true, false>(a, st) : launch_x3n<128, 64, false, false>(a, st);
    return a.g.act ? launch_x3<true>(a, st) : launch_x3<false>(a, st);
}

}  // namespace ampnet

#ifdef AMPNET_PW_STAMPS
extern "C" int ampnet_debug_bx_stamps(unsigned long long *host_out, int max_entries)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    const size_t n = (size_t)(max_entries < 4096 ? max_entries : 4096);
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(ampnet::g_bx_stamps), n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
