// pw_misc.hip -- the small kernels around pw_gemm: the K<=12 input layers (VALU), BatchNorm statistic
// finalisation / folding / running update, and the MaxPool finalisation.
//
// Reference ops (pointNet/model/pointnetAtt.py): input T-Net conv_1 (:31, K = 3), encoder conv_1 on
// cat(xyz * T, x) (:85-90, K = 12), every nn.BatchNorm1d (:17-22, :73-78, :173-174; eps 1e-5, momentum 0.1,
// unbiased running variance) and nn.MaxPool1d(num_points) (:35, :104).
#include "kernels.h"

namespace ampnet {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ----------------------------------------------------------------------------------------------------
// pw_input: 64 output channels, lane = channel, a wave walks rows.  HBM-bound (36 B in, 256 B out per
// point), so the layout that matters is the coalesced 256-byte row store; x rows are staged through LDS.
// ----------------------------------------------------------------------------------------------------
constexpr int IN_ROWS = 256;   // rows staged per pass

__global__ __launch_bounds__(256) void pw_input_kernel(PwInput a)
{
    __shared__ __attribute__((aligned(16))) float sx[IN_ROWS * 12];
    __shared__ float red[4][64][3];
    __shared__ int red_n[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Two grids.  part_rows == nullptr: block (chunk, q) handles one block of rows and writes its own statistics partial (eval mode: none).
    // part_rows != nullptr (train): a 1-D grid of stat_lanes persistent blocks; block (slot, j) = (blockIdx.x % n_slots, / n_slots) walks
    // the blocks of rows j, j + L_slot, ... of its slot and leaves ONE partial (like pw_gemm: bn_finalize then runs in one stage).
    const bool wg = a.part_rows != nullptr;
    const StatLane sl = wg ? stat_lane_of(blockIdx.x, a.stat_lanes, a.n_slots, a.Q, a.chunks) : StatLane{0, 0, 1};
    const int my_slot = sl.slot, slot_lanes = sl.L;
    const int part_idx = sl.slot + sl.j * a.n_slots;
    const int items = wg ? ((a.Q - my_slot + a.n_slots - 1) / a.n_slots) * a.chunks : 1;
    double run_mean = 0.0, run_m2 = 0.0;          // wave 0, lane = channel
    int run_n = 0;
    for (int it = wg ? sl.j : 0; it < items; it += slot_lanes) {
        const int q = wg ? my_slot + (it / a.chunks) * a.n_slots : blockIdx.y, chunk = wg ? it % a.chunks : blockIdx.x;
        const int row_begin = a.win_off[q] + chunk * a.chunk_rows;
        const int row_end = min(a.win_off[q + 1], row_begin + a.chunk_rows);

        // effective weights of this lane's output channel
        float w[9];
        if (a.mode == 0) {
#pragma unroll
            for (int f = 0; f < 9; ++f) w[f] = f < 3 ? a.W[lane * 3 + f] : 0.f;
        } else {
            const int pidx = a.perwin_slot_major ? (q % a.n_slots) * (a.Q / a.n_slots) + q / a.n_slots : q;
            const float *T = a.T + (size_t)pidx * 9;
#pragma unroll
            for (int f = 0; f < 9; ++f) {
                float v = a.W[lane * 12 + 3 + f];
                if (f < 3) v += T[f * 3 + 0] * a.W[lane * 12 + 0] + T[f * 3 + 1] * a.W[lane * 12 + 1] + T[f * 3 + 2] * a.W[lane * 12 + 2];
                w[f] = v;
            }
        }
        const int nf = a.mode == 0 ? 3 : 9;
        float s = 0.f, sq = 0.f, z0 = 0.f;     // sums of (z - z0), (z - z0)^2 with z0 = the wave's first row (see pw_gemm.hip)
        int cnt = 0;
        for (int base = row_begin; base < row_end; base += IN_ROWS) {
            const int n = min(IN_ROWS, row_end - base);
            __syncthreads();
            {
                // rows on a 12-float pitch: a row's nine features are three 16-byte LDS reads below instead of nine 4-byte ones
                float tmp[9];
#pragma unroll
                for (int u = 0; u < 9; ++u) tmp[u] = (tid + 256 * u) < n * 9 ? a.x[(size_t)base * 9 + tid + 256 * u] : 0.f;
#pragma unroll
                for (int u = 0; u < 9; ++u) {
                    const int e = tid + 256 * u;
                    if (e < n * 9) sx[(e / 9) * 12 + e % 9] = tmp[u];
                }
            }
            __syncthreads();
            for (int i = wave; i < n; i += 4) {
                float z = 0.f;
                const f32x4 x0 = *reinterpret_cast<const f32x4 *>(sx + i * 12);
                if (nf == 3) {
                    z = x0[0] * w[0] + x0[1] * w[1] + x0[2] * w[2];
                } else {
                    const f32x4 x1 = *reinterpret_cast<const f32x4 *>(sx + i * 12 + 4);
                    const float x8 = sx[i * 12 + 8];
#pragma unroll
                    for (int f = 0; f < 4; ++f) z = fmaf(x0[f], w[f], z);
#pragma unroll
                    for (int f = 0; f < 4; ++f) z = fmaf(x1[f], w[4 + f], z);
                    z = fmaf(x8, w[8], z);
                }
                if (a.z_bf16) reinterpret_cast<__bf16 *>(a.Z)[(size_t)(base + i) * 64 + lane] = (__bf16)z;
                else a.Z[(size_t)(base + i) * 64 + lane] = z;
                if (cnt == 0) z0 = z;
                const float d = z - z0;
                s += d;
                sq = fmaf(d, d, sq);
                ++cnt;
            }
        }
        if (!a.part_sum) continue;
        __syncthreads();                           // the previous item's merge has read red[]
        red[wave][lane][0] = s;
        red[wave][lane][1] = sq;
        red[wave][lane][2] = z0;
        if (lane == 0) red_n[wave] = cnt;
        __syncthreads();
        if (wave == 0) {
            double n = 0.0, mean = 0.0, m2 = 0.0;
            for (int w2 = 0; w2 < 4; ++w2) {
                const double nw = (double)red_n[w2];
                if (nw <= 0.0) continue;
                const double s1 = red[w2][lane][0], s2 = red[w2][lane][1];
                const double mw = (double)red[w2][lane][2] + s1 / nw, m2w = s2 - s1 * s1 / nw;
                const double nn = n + nw, delta = mw - mean;
                mean += delta * nw / nn;
                m2 += m2w + delta * delta * n * nw / nn;
                n = nn;
            }
            if (m2 < 0.0) m2 = 0.0;
            if (!wg) {
                const size_t o = (size_t)(q * a.chunks + chunk) * 64 + lane;
                a.part_sum[o] = (float)mean;
                a.part_sq[o] = (float)m2;
            } else if (n > 0.0) {                  // merge into the block's running partial (Chan)
                const double rn = (double)run_n, nn = rn + n, delta = mean - run_mean;
                run_mean += delta * n / nn;
                run_m2 += m2 + delta * delta * rn * n / nn;
                run_n += (int)n;
            }
        }
    }
    if (wg && a.part_sum && wave == 0) {
        a.part_sum[(size_t)part_idx * 64 + lane] = (float)run_mean;
        a.part_sq[(size_t)part_idx * 64 + lane] = (float)run_m2;
        if (lane == 0) {
            a.part_rows[part_idx] = run_n;
            if (sl.j == sl.L - 1 && sl.L < (a.stat_lanes + a.n_slots - 1) / a.n_slots) a.part_rows[part_idx + a.n_slots] = 0;
        }
    }
}

// persistent blocks for pw_input's statistics: the kernel is HBM-bound and light on registers / LDS, so it wants several blocks per CU:
// up to 1024 (four per CU), each walking the blocks of rows of one slot
int pw_input_stat_lanes(int Q, int chunks, int n_slots)
{
    const long items = (long)Q * chunks;
    int lanes = items < 1024 ? (int)items : 1024;
    return lanes < n_slots ? n_slots : lanes;
}

int pw_input(const PwInput &a, hipStream_t st)
{
    AMPNET_REQUIRE(a.x && a.W && a.Z && a.win_off, "pw_input: null pointer");
    AMPNET_REQUIRE(a.mode == 0 || a.T, "pw_input: mode 1 needs T");
    AMPNET_REQUIRE((a.part_sum == nullptr) == (a.part_sq == nullptr), "pw_input: partials");
    AMPNET_REQUIRE(!a.part_rows || (a.part_sum && a.stat_lanes >= a.n_slots && a.n_slots >= 1), "pw_input: per-block statistics need part_sum and a plan");
    if (a.part_rows) hipLaunchKernelGGL(pw_input_kernel, dim3(a.stat_lanes), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(pw_input_kernel, dim3(a.chunks, a.Q), dim3(256), 0, st, a);
    return check_launch("pw_input_kernel");
}

// ----------------------------------------------------------------------------------------------------
// bn_finalize: block = (slot, 64 channels) x 4 groups; merges the per-chunk (mean, M2) partials of the slot's
// windows in double, in a fixed order: bitwise reproducible run to run.
// ----------------------------------------------------------------------------------------------------
constexpr int FIN_G = 16;      // partial groups per channel (block = 64 channels x 16 groups)
constexpr int FIN_V = 16;      // two-stage form: sub-slots per slot merged first (few slots x thousands of partials)

// MERGE: write the merged (mean, M2, rows) of the block's (sub-)slot as a partial instead of the BatchNorm constants.
template <bool MERGE>
__global__ __launch_bounds__(64 * FIN_G) void bn_finalize_kernel(BnFinalize a, float *out_sum, float *out_sq, int *out_rows)
{
    // two passes over the slot's chunk partials (n_i, mean_i, M2_i), no division inside the loops:
    //   mean = sum n_i mean_i / N;   M2 = sum (M2_i + n_i (mean_i - mean)^2)
    __shared__ double rn[FIN_G][64], rs[FIN_G][64];
    const int slot = blockIdx.x;
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    const int per_slot = (a.Q - slot + a.n_slots - 1) / a.n_slots;     // windows q = slot, slot + n_slots, ...
    const int total = per_slot * a.chunks;
    const bool ok = c < a.C;
    const int sh = (a.chunks & (a.chunks - 1)) == 0 ? __ffs(a.chunks) - 1 : -1;    // chunks = 2^sh: no division
    // element e of the slot -> (rows of that chunk, offset of its partial); one division at most
    auto decode = [&](int e, int &rows, size_t &off) {
        const int qi = sh >= 0 ? e >> sh : e / a.chunks;
        const int ch = e - qi * a.chunks;
        const int q = slot + qi * a.n_slots;
        const int p = q * a.chunks + ch;
        if (a.gather_ranks > 0) {        // partial qi = rank qi's merged slot: segment {mean, M2, rows}
            const float *seg = a.part_sum + (size_t)qi * a.gather_seg;
            rows = reinterpret_cast<const int *>(seg + (size_t)2 * a.n_slots * a.C)[slot];
            off = (size_t)qi * a.gather_seg + (size_t)slot * a.C + c;
            return;
        }
        if (a.part_rows) {
            rows = a.part_rows[p];
        } else {
            const int wrows = a.uniform_rows > 0 ? a.uniform_rows : a.win_off[q + 1] - a.win_off[q];
            rows = max(min(wrows - ch * a.chunk_rows, a.chunk_rows), 0);
        }
        off = (size_t)p * a.C + c;
    };
    // the partial loads are issued eight at a time before anything consumes them: a load -> use -> load chain costs one
    // memory round trip per partial
    double n = 0.0, sm = 0.0;
    for (int e0 = g; e0 < total; e0 += FIN_G * 8) {
        float v[8];
        int rw[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + FIN_G * u;
            size_t off = 0;
            rw[u] = 0;
            if (e < total && ok) decode(e, rw[u], off);
            v[u] = rw[u] > 0 ? a.part_sum[off] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            n += (double)rw[u];
            sm += (double)rw[u] * (double)v[u];
        }
    }
    rn[g][cl] = n;
    rs[g][cl] = sm;
    __syncthreads();
    double N = 0.0, S = 0.0;
#pragma unroll
    for (int k = 0; k < FIN_G; ++k) {
        N += rn[k][cl];
        S += rs[k][cl];
    }
    const double mean = N > 0.0 ? S / N : 0.0;
    __syncthreads();
    double m2 = 0.0;
    for (int e0 = g; e0 < total; e0 += FIN_G * 8) {
        float v[8], w[8];
        int rw[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + FIN_G * u;
            size_t off = 0;
            rw[u] = 0;
            if (e < total && ok) decode(e, rw[u], off);
            v[u] = rw[u] > 0 ? a.part_sum[off] : 0.f;
            w[u] = rw[u] > 0 ? a.part_sq[off] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double d = (double)v[u] - mean;
            m2 += rw[u] > 0 ? (double)w[u] + (double)rw[u] * d * d : 0.0;
        }
    }
    rs[g][cl] = m2;
    __syncthreads();
    if (g == 0 && ok) {
        m2 = 0.0;
#pragma unroll
        for (int k = 0; k < FIN_G; ++k) m2 += rs[k][cl];
        const size_t o = (size_t)slot * a.C + c;
        if (MERGE) {
            out_sum[o] = (float)mean;
            out_sq[o] = (float)m2;
            if (c == 0) out_rows[slot] = (int)N;
            return;
        }
        const double var = N > 0.0 ? m2 / N : 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)a.eps));
        const float sc = a.gamma[c] * invstd;
        a.scale[o] = sc;
        a.shift[o] = a.beta[c] - (float)mean * sc;
        if (a.mean) a.mean[o] = (float)mean;
        if (a.invstd) a.invstd[o] = invstd;
        if (a.stat_mean) {
            a.stat_mean[o] = (float)mean;
            a.stat_uvar[o] = (float)(N > 1.0 ? m2 / (N - 1.0) : m2);
        }
    }
}

int bn_finalize(const BnFinalize &a, hipStream_t st)
{
    AMPNET_REQUIRE(a.part_sum && a.part_sq && (a.win_off || a.part_rows || a.uniform_rows > 0) && a.gamma && a.beta && a.scale && a.shift, "bn_finalize: null pointer");
    if (sync_bn_on()) {
        // global batch: merge the rank's partials per slot, all-gather the (rows, mean, M2) of every rank, finalize from those
        const size_t seg = (size_t)a.n_slots * (2 * (size_t)a.C + 1);
        float *loc = nullptr, *all = nullptr;
        int rc = sync_bn_gather(seg, &loc, &all);
        if (rc != AMPNET_OK) return rc;
        float *m_sum = loc, *m_sq = loc + (size_t)a.n_slots * a.C;
        int *m_rows = reinterpret_cast<int *>(m_sq + (size_t)a.n_slots * a.C);
        hipLaunchKernelGGL(bn_finalize_kernel<true>, dim3(a.n_slots, cdiv(a.C, 64)), dim3(64 * FIN_G), 0, st, a, m_sum, m_sq, m_rows);
        rc = check_launch("bn_finalize_kernel (rank merge)");
        if (rc != AMPNET_OK) return rc;
        rc = sync_bn_exchange(AMPNET_COLLECTIVE_ALLGATHER, loc, all, seg, st);
        if (rc != AMPNET_OK) return rc;
        BnFinalize s2 = a;
        s2.part_sum = all; s2.part_sq = all + (size_t)a.n_slots * a.C;
        s2.gather_ranks = sync_bn_world(); s2.gather_seg = (long)seg;
        s2.Q = sync_bn_world() * a.n_slots; s2.chunks = 1; s2.uniform_rows = 0; s2.part_rows = nullptr;
        hipLaunchKernelGGL(bn_finalize_kernel<false>, dim3(a.n_slots, cdiv(a.C, 64)), dim3(64 * FIN_G), 0, st, s2, nullptr, nullptr, nullptr);
        return check_launch("bn_finalize_kernel (global batch)");
    }
    const long per_slot_parts = (long)((a.Q + a.n_slots - 1) / a.n_slots) * a.chunks;
    if (a.merge_ws && per_slot_parts > 192 && a.n_slots * FIN_V <= a.Q) {
        // few slots, thousands of partials each (the head: one slot): merge FIN_V sub-slots per slot on FIN_V x more
        // workgroups first, then finalize those
        const int V = a.n_slots * FIN_V;
        float *m_sum = a.merge_ws, *m_sq = m_sum + (size_t)V * a.C;
        int *m_rows = reinterpret_cast<int *>(m_sq + (size_t)V * a.C);
        BnFinalize s1 = a;
        s1.n_slots = V;
        hipLaunchKernelGGL(bn_finalize_kernel<true>, dim3(V, cdiv(a.C, 64)), dim3(64 * FIN_G), 0, st, s1, m_sum, m_sq, m_rows);
        BnFinalize s2 = a;
        s2.part_sum = m_sum; s2.part_sq = m_sq; s2.part_rows = m_rows;
        s2.Q = V; s2.chunks = 1; s2.uniform_rows = 0;
        hipLaunchKernelGGL(bn_finalize_kernel<false>, dim3(a.n_slots, cdiv(a.C, 64)), dim3(64 * FIN_G), 0, st, s2, nullptr, nullptr, nullptr);
        return check_launch("bn_finalize_kernel (two stages)");
    }
    hipLaunchKernelGGL(bn_finalize_kernel<false>, dim3(a.n_slots, cdiv(a.C, 64)), dim3(64 * FIN_G), 0, st, a, nullptr, nullptr, nullptr);
    return check_launch("bn_finalize_kernel");
}

// ----------------------------------------------------------------------------------------------------
// bn_fold / bn_running_update: a list of layers per launch (items in kernel arguments, <= 24 layers)
// ----------------------------------------------------------------------------------------------------
constexpr int MAX_ITEMS = 24;
struct FoldArgs {
    BnFoldItem it[MAX_ITEMS];
    int n;
    float eps;
};
struct RunArgs {
    BnRunItem it[MAX_ITEMS];
    int n;
    float momentum;
};

__global__ void bn_fold_kernel(FoldArgs a)
{
    const BnFoldItem it = a.it[blockIdx.x];
    for (int c = threadIdx.x; c < it.C; c += blockDim.x) {
        const float invstd = 1.0f / sqrtf(it.rvar[c] + a.eps);
        const float sc = it.gamma[c] * invstd;
        it.scale[c] = sc;
        it.shift[c] = it.beta[c] - it.rmean[c] * sc;
    }
}

__global__ void bn_running_kernel(RunArgs a)
{
    const BnRunItem it = a.it[blockIdx.x];
    for (int c = threadIdx.x; c < it.C; c += blockDim.x) {
        float m = it.rmean[c], v = it.rvar[c];
        for (int s = 0; s < it.n_slots; ++s) {      // slot order = the order of the reference's W encoder calls
            m = (1.0f - a.momentum) * m + a.momentum * it.stat_mean[(size_t)s * it.C + c];
            v = (1.0f - a.momentum) * v + a.momentum * it.stat_uvar[(size_t)s * it.C + c];
        }
        it.rmean[c] = m;
        it.rvar[c] = v;
    }
}

int bn_fold(const BnFoldItem *items, int n, float eps, hipStream_t st)
{
    AMPNET_REQUIRE(n >= 1 && n <= MAX_ITEMS, "bn_fold: %d items", n);
    FoldArgs a;
    for (int i = 0; i < n; ++i) a.it[i] = items[i];
    a.n = n;
    a.eps = eps;
    hipLaunchKernelGGL(bn_fold_kernel, dim3(n), dim3(256), 0, st, a);
    return check_launch("bn_fold_kernel");
}

int bn_running_update(const BnRunItem *items, int n, float momentum, hipStream_t st)
{
    AMPNET_REQUIRE(n >= 1 && n <= MAX_ITEMS, "bn_running_update: %d items", n);
    RunArgs a;
    for (int i = 0; i < n; ++i) a.it[i] = items[i];
    a.n = n;
    a.momentum = momentum;
    hipLaunchKernelGGL(bn_running_kernel, dim3(n), dim3(256), 0, st, a);
    return check_launch("bn_running_kernel");
}

// ----------------------------------------------------------------------------------------------------
// pool_finalize: relu(scale * extreme + shift) where extreme = max for scale >= 0, min otherwise
// (BatchNorm then ReLU are monotone per channel, so MaxPool commutes with them up to the sign of scale).
// ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pool_finalize_kernel(PoolFinalize a)
{
    const int q = blockIdx.x;
    const int slot = a.n_slots > 1 ? q % a.n_slots : 0;
    const int orow = a.out_slot_major ? (q % a.n_slots) * (a.Q / a.n_slots) + q / a.n_slots : q;
    for (int c = threadIdx.x; c < a.C; c += blockDim.x) {
        const float sc = a.scale[(size_t)slot * a.C + c], sh = a.shift[(size_t)slot * a.C + c];
        float best = 0.f;
        int arg = -1;
        const bool use_max = sc >= 0.f;
        for (int ch = 0; ch < a.chunks; ++ch) {
            const size_t o = (size_t)(q * a.chunks + ch) * a.C + c;
            const float v = a.part_max[o];
            const int i = a.part_amax ? a.part_amax[o] : (__builtin_isinf(v) ? -1 : 0);    // no rows tracked (eval): an empty chunk holds -+inf
            if (i < 0) continue;
            if (arg < 0 || (use_max ? v > best : v < best)) {     // chunks ascend in row order: first extreme wins
                best = v;
                arg = i;
            }
        }
        a.pooled[(size_t)orow * a.C + c] = fmaxf(fmaf(best, sc, sh), 0.f);
        if (a.arg) a.arg[(size_t)q * a.C + c] = arg;
        if (a.zext) a.zext[(size_t)q * a.C + c] = best;
    }
}

// The same with the layer's BatchNorm finished in the kernel (PoolFinalize.pfin_*): workgroup = (slot, four windows of it), 1024 threads =
// 4 groups x 256 channels.  Group g first sums the slot's partials g, g + 4, ... (n, n mean, M2 + n mean^2 in double, sixteen partials in
// flight per thread: one memory round trip for the ~57 a slot has), the groups merge through LDS in group order, every thread forms its
// channel's scale / shift, and group g then pools window 4 wg + g.  Workgroup 0 of a slot writes the BatchNorm arrays.
__global__ __launch_bounds__(1024) void pool_finalize_bn_kernel(PoolFinalize a)
{
    __shared__ double rs[4][256][2];
    __shared__ double rn[4];
    const int slot = blockIdx.x % a.n_slots, wg = blockIdx.x / a.n_slots;
    const int c = threadIdx.x & 255, g = threadIdx.x >> 8;
    const int per_slot_parts = a.pfin_parts / a.n_slots;
    double sn = 0.0, sm = 0.0, sq = 0.0;
    for (int k0 = g; k0 < per_slot_parts; k0 += 4 * 16) {
        float v[16], w[16];
        int rw[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int kk = k0 + 4 * u;
            const int idx = slot + (kk < per_slot_parts ? kk : 0) * a.n_slots;
            rw[u] = kk < per_slot_parts ? a.pfin_rows[idx] : 0;
            v[u] = a.pfin_sum[(size_t)idx * 256 + c];
            w[u] = a.pfin_sq[(size_t)idx * 256 + c];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (rw[u] > 0) {
                const double nn = (double)rw[u], m = (double)v[u];
                sn += nn;
                sm += nn * m;
                sq += (double)w[u] + nn * m * m;
            }
        }
    }
    rs[g][c][0] = sm;
    rs[g][c][1] = sq;
    if (c == 0) rn[g] = sn;
    __syncthreads();
    const double N = (rn[0] + rn[1]) + (rn[2] + rn[3]);
    const double S = (rs[0][c][0] + rs[1][c][0]) + (rs[2][c][0] + rs[3][c][0]);
    const double Q2 = (rs[0][c][1] + rs[1][c][1]) + (rs[2][c][1] + rs[3][c][1]);
    const double mean = N > 0.0 ? S / N : 0.0;
    double m2 = Q2 - N * mean * mean;
    if (m2 < 0.0) m2 = 0.0;
    const double var = N > 0.0 ? m2 / N : 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)a.pfin_eps));
    const float sc = a.pfin_gamma[c] * invstd;
    const float sh = a.pfin_beta[c] - (float)mean * sc;
    if (wg == 0 && g == 0) {
        const size_t o = (size_t)slot * 256 + c;
        a.pfin_scale[o] = sc;
        a.pfin_shift[o] = sh;
        a.pfin_mean[o] = (float)mean;
        a.pfin_invstd[o] = invstd;
        a.pfin_smean[o] = (float)mean;
        a.pfin_suvar[o] = (float)(N > 1.0 ? m2 / (N - 1.0) : m2);
    }
    const int q = slot + (wg * 4 + g) * a.n_slots;
    if (q >= a.Q) return;
    const int orow = a.out_slot_major ? (q % a.n_slots) * (a.Q / a.n_slots) + q / a.n_slots : q;
    float best = 0.f;
    int arg = -1;
    const bool use_max = sc >= 0.f;
    for (int ch = 0; ch < a.chunks; ++ch) {
        const size_t o = (size_t)(q * a.chunks + ch) * 256 + c;
        const float v = a.part_max[o];
        const int i = a.part_amax[o];
        if (i < 0) continue;
        if (arg < 0 || (use_max ? v > best : v < best)) {     // chunks ascend in row order: first extreme wins
            best = v;
            arg = i;
        }
    }
    a.pooled[(size_t)orow * 256 + c] = fmaxf(fmaf(best, sc, sh), 0.f);
    if (a.arg) a.arg[(size_t)q * 256 + c] = arg;
    if (a.zext) a.zext[(size_t)q * 256 + c] = best;
}

int pool_finalize(const PoolFinalize &a, hipStream_t st)
{
    AMPNET_REQUIRE(a.part_max && a.pooled && (a.pfin_sum ? a.part_amax != nullptr : (a.scale && a.shift)), "pool_finalize: null pointer");
    AMPNET_REQUIRE(a.part_amax || (!a.arg), "pool_finalize: argmax rows requested but not tracked by the producer");
    AMPNET_REQUIRE(!a.out_slot_major || a.Q % a.n_slots == 0, "pool_finalize: Q %% n_slots != 0");
    if (a.pfin_sum) {
        AMPNET_REQUIRE(a.C == 256 && a.pfin_sq && a.pfin_rows && a.pfin_parts >= a.n_slots && a.pfin_parts % a.n_slots == 0 && a.pfin_gamma && a.pfin_beta && a.pfin_scale &&
                           a.pfin_shift && a.pfin_mean && a.pfin_invstd && a.pfin_smean && a.pfin_suvar,
                       "pool_finalize: in-kernel BatchNorm finalize needs 256 channels, the producer's partials and every output array");
        const int per_slot = cdiv(a.Q, a.n_slots);
        hipLaunchKernelGGL(pool_finalize_bn_kernel, dim3(a.n_slots * cdiv(per_slot, 4)), dim3(1024), 0, st, a);
        return check_launch("pool_finalize_bn_kernel");
    }
    hipLaunchKernelGGL(pool_finalize_kernel, dim3(a.Q), dim3(256), 0, st, a);
    return check_launch("pool_finalize_kernel");
}

// ----------------------------------------------------------------------------------------------------
__global__ void add_identity_kernel(float *T, int n_mats, int k)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_mats * k) T[(size_t)(i / k) * k * k + (i % k) * (k + 1)] += 1.0f;
}

int add_identity(float *T, int n_mats, int k, hipStream_t st)
{
    hipLaunchKernelGGL(add_identity_kernel, dim3(cdiv(n_mats * k, 256)), dim3(256), 0, st, T, n_mats, k);
    return check_launch("add_identity_kernel");
}

__global__ void ramp_kernel(int *dst, int n, int step)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = i * step;
}

int fill_i32_ramp(int *dst, int n, int step, hipStream_t st)
{
    hipLaunchKernelGGL(ramp_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, dst, n, step);
    return check_launch("ramp_kernel");
}

}  // namespace ampnet
