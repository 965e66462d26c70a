// kernels.h -- internal launch interface between the C-ABI orchestration (encoder.hip, head.hip, ...)
// and the kernels.  Not part of the public ABI.
#pragma once
#include "common.h"

namespace ampnet {

// Streaming accesses: the per-row tensors of a layer (hundreds of MB each, read or written once per launch) are loaded and stored with the
// non-temporal hint, so that they do not push the weights / partials / the other stream's lines out of L2 and the infinity cache on their way
// through.  Same-box A/B on the 128 -> 64 fused backward: 0.371 -> 0.352 ms (<true>), 0.441 -> 0.398 ms (<false>), step kernel time -0.14 ms.
// AMPNET_NO_STREAM_HINT at compile time turns them back into plain accesses.
#if defined(__HIPCC__) || defined(__HIP__)
#ifdef AMPNET_NO_STREAM_HINT
template <typename T> __device__ __forceinline__ T ld_stream(const T *p) { return *p; }
template <typename T> __device__ __forceinline__ void st_stream(T v, T *p) { *p = v; }
#else
template <typename T> __device__ __forceinline__ T ld_stream(const T *p) { return __builtin_nontemporal_load(p); }
#ifdef AMPNET_NO_STREAM_STORE
template <typename T> __device__ __forceinline__ void st_stream(T v, T *p) { *p = v; }
#else
template <typename T> __device__ __forceinline__ void st_stream(T v, T *p) { __builtin_nontemporal_store(v, p); }
#endif
#endif
#endif


// ----------------------------------------------------------------------------------------------------
// Per-point linear layer ("shared MLP" = Conv1d k=1) on fp32 MFMA:  Z[row, :] = pro(A[row, :]) * W^T + bias
//   rows are grouped in windows (win_off[q] .. win_off[q+1]); a workgroup owns one chunk of one window and
//   one block of <=128 output channels.
//   pro(a)[k] = relu(a[k] * pro_scale[slot][k] + pro_shift[slot][k]) (the previous layer's BatchNorm + ReLU,
//   optionally followed by dropout) or the identity when pro_scale == nullptr.
//   Epilogue options: store Z; per-chunk column sums / sums of squares (BatchNorm statistics of THIS layer);
//   per-chunk column max / min with their row index (MaxPool1d over the window, before the affine).
// ----------------------------------------------------------------------------------------------------
struct PwGemm {
    const float *A = nullptr;      // [rows, lda]
    int lda = 0;
    int cin = 0;                   // 64, 128 or 256
    const float *W = nullptr;      // shared: [cout, ldw]; per window: [.., cin, cout] k-major (see w_win_stride)
    int ldw = 0;
    long w_win_stride = 0;         // != 0: weights of window q start at W + pidx(q) * w_win_stride, layout [cin][cout]
    const float *bias = nullptr;   // [cout] or per window [.., cout]
    long bias_win_stride = 0;
    int perwin_slot_major = 0;     // pidx(q) = (q % n_slots) * (Q / n_slots) + q / n_slots instead of q
    const float *pro_scale = nullptr;   // [pro_slots, cin]
    const float *pro_shift = nullptr;
    int n_slots = 1;               // slot(q) = q % n_slots (batch-statistics groups); 1 in eval mode
    float drop_p = 0.f;            // dropout on pro(a): keep iff hash(row * cin + k) >= p * 2^32, scaled 1/(1-p)
    uint32_t drop_seed = 0;
    float *Z = nullptr;            // [rows, ldz] or nullptr
    int ldz = 0;
    int cout = 0;
    float *part_sum = nullptr;     // [Q * chunks, cout] or nullptr: per-chunk MEAN of each output column
    float *part_sq = nullptr;      //                               per-chunk sum of squared deviations from it
    float *part_max = nullptr;     // [Q * chunks, cout] or nullptr: per-chunk extreme of each column (max if gamma >= 0 else min)
    int *part_amax = nullptr;      //                               and the row it sits in
    const float *pool_gamma = nullptr;   // [cout] BatchNorm weight of THIS layer: its sign picks max or min; nullptr = max
    const int *win_off = nullptr;  // [Q + 1] device
    int Q = 0;
    int chunk_rows = 512;
    int chunks = 1;                // cdiv(max window rows, chunk_rows)
    long rows_hint = 0;            // total rows (profiling only: algorithmic flops / bytes of the launch)
    int a_bf16 = 0, z_bf16 = 0;    // A / Z are bf16 tensors ([rows, lda] / [rows, ldz] ELEMENTS): activation storage of precision mode 3
    int uniform_rows = 0;          // > 0: window q is rows q * uniform_rows .. (win_off may be nullptr): no offset array to fill
    int identity_k = 0;            // > 0: + 1 on the output columns i * (k + 1) (the identity a T-Net adds to its k x k transform)
    // per-WORKGROUP statistics (not with the pool epilogue): part_rows != nullptr -> part_sum / part_sq are [stat_lanes, cout],
    // partial `lane` belongs to slot lane % n_slots, part_rows[lane] = rows it covers; stat_lanes from pw_gemm_stat_plan()
    int *part_rows = nullptr;
    int stat_lanes = 0;
    // plan.direct (one block of rows per slot): the workgroup finishes the BatchNorm constants itself, no partial, no bn_finalize launch
    const float *fin_gamma = nullptr, *fin_beta = nullptr;
    float *fin_scale = nullptr, *fin_shift = nullptr, *fin_mean = nullptr, *fin_invstd = nullptr, *fin_smean = nullptr, *fin_suvar = nullptr;
    float fin_eps = 1e-5f;
    // Consumer-side finalize of the INPUT's BatchNorm (train mode, per-workgroup statistics, PRO >= 1): instead of a bn_finalize launch
    // between producer and consumer (~11 us of a mostly idle chip per layer), every workgroup of the consumer merges the producer's
    // per-workgroup partials of ITS slot (pfin_parts / n_slots <= 114 partials of cin channels, out of L2) into the prologue constants it
    // is about to stage; lane j == 0 of each slot (column block 0) also writes the arrays the backward, later consumers and the
    // running-statistics update read.  The producer's partials must not be the buffers this launch writes its own partials to.
    const float *pfin_sum = nullptr, *pfin_sq = nullptr;
    const int *pfin_rows = nullptr;
    int pfin_parts = 0;                       // partial slots (a multiple of n_slots; empty ones carry rows 0)
    const float *pfin_gamma = nullptr, *pfin_beta = nullptr;
    float *pfin_scale = nullptr, *pfin_shift = nullptr, *pfin_mean = nullptr, *pfin_invstd = nullptr, *pfin_smean = nullptr, *pfin_suvar = nullptr;
};
struct PwStatPlan {
    int lanes = 1;                 // workgroups that take part
    int parts = 1;                 // partial slots they write: (lanes / n_slots rounded up) * n_slots, partial p belongs to slot p % n_slots
    bool direct = false;
};
// Which blocks of rows a lane of the per-workgroup statistics walks.  `lanes` workgroups are dealt to the slots (lanes / n_slots each, one
// more for the first lanes % n_slots slots); inside a slot lane j takes the blocks j, j + L, ... of the slot's I blocks, so I % L lanes
// walk one block more than the others.  Lane ids are handed out HEAVY LANES FIRST: with two workgroups per CU (ids c and c + 256) every CU
// then gets one lane of each kind -- 2304 blocks over 512 lanes are 256 x 5 + 256 x 4, nine per CU, as even as a slot-blind walk.
struct StatLane {
    int slot, j, L;                // my slot, my index in it, lanes of that slot
};
__device__ __forceinline__ StatLane stat_lane_of(int lane_id, int lanes, int n_slots, int Q, int chunks)
{
    const int base = lanes / n_slots, extra = lanes % n_slots;
    int rem = lane_id;
    for (int pass = 0; pass < 2; ++pass) {
        for (int s = 0; s < n_slots; ++s) {
            const int L = base + (s < extra ? 1 : 0);
            const int I = ((Q - s + n_slots - 1) / n_slots) * chunks;
            const int h = L > 0 ? (I >= L ? I % L : I) : 0;                 // lanes of slot s that walk one block more
            const int cnt = pass == 0 ? h : L - h;
            if (rem < cnt) return StatLane{s, pass == 0 ? rem : h + rem, L};
            rem -= cnt;
        }
    }
    return StatLane{0, 0, 1};      // not reached for lane_id < lanes
}
PwStatPlan pw_gemm_stat_plan(int Q, int chunks, int n_slots, int max_lanes = 512);
int pw_gemm_stat_lane_cap(int cin, int cout);      // 512, or what the split kernels of this shape hold at once (precision mode 4)
int pw_gemm(const PwGemm &a, hipStream_t st);
// one element of an activation tensor that is fp32 or (precision mode 3) bf16
__device__ __forceinline__ float ld_act(const float *base, size_t elem, int bf16)
{
    return bf16 ? (float)reinterpret_cast<const __bf16 *>(base)[elem] : base[elem];
}
// the same with the format fixed at compile time: a runtime flag puts a branch around every load, and the gather loops of the backward
// (eight or sixteen independent loads per trip) then wait for each one in turn -- measured 2.2 x on pw_input_wgrad in mode 3
template <bool ZB> __device__ __forceinline__ float ld_act_t(const float *base, size_t elem)
{
    if constexpr (ZB) return (float)reinterpret_cast<const __bf16 *>(base)[elem];
    else return base[elem];
}
// activations kept for the backward (the nine pre-BatchNorm z tensors of the encoder, z2 / z3 of the head) are stored as bf16
inline bool z_storage_bf16() { return matrix_precision() == AMPNET_PRECISION_BF16_STORE; }
// bf16 MFMA operands in the fused backward (modes 2 and 3)
inline bool bwd_operands_bf16() { return matrix_precision() == AMPNET_PRECISION_BF16_TRAIN || matrix_precision() == AMPNET_PRECISION_BF16_STORE; }
// fp32 semantics: exact fp32 MFMA (mode 0) or the three-term bf16 split of the operands (mode 4: fp32 results from the bf16 pipe)
inline bool precision_is_f32() { return matrix_precision() == AMPNET_PRECISION_F32 || matrix_precision() == AMPNET_PRECISION_F32_SPLIT; }
inline bool precision_split() { return matrix_precision() == AMPNET_PRECISION_F32_SPLIT; }

// First layers with a tiny contraction (K = 3 or 12), VALU: Z[row, 0:64] = x[row, cols] * Weff^T
//   mode 0: Weff = W[64][3] on x[:, 0:3]                                     (T-Net conv_1 on xyz)
//   mode 1: Weff[c][f] = W[c][3+f] + (f < 3 ? sum_d T[q][f][d] * W[c][d] : 0)  on x[:, 0:9]
//           = conv_1 of cat(xyz * T, x)                                       (pointnetAtt.py:85-90)
struct PwInput {
    const float *x = nullptr;      // [rows, 9]
    const float *W = nullptr;      // mode 0: [64, 3]; mode 1: [64, 12]
    const float *T = nullptr;      // mode 1: [.., 3, 3] per window (slot-major index if perwin_slot_major)
    int mode = 0;
    int perwin_slot_major = 0;
    int n_slots = 1;
    float *Z = nullptr;            // [rows, 64]
    int z_bf16 = 0;                // Z is a bf16 tensor (precision mode 3)
    float *part_sum = nullptr, *part_sq = nullptr;    // [Q * chunks, 64] or nullptr (chunk mean, chunk M2)
    int *part_rows = nullptr;      // != nullptr: one partial per persistent block, [stat_lanes, 64] + rows (see PwGemm.part_rows)
    int stat_lanes = 0;
    const int *win_off = nullptr;
    int Q = 0, chunk_rows = 512, chunks = 1;
};
int pw_input_stat_lanes(int Q, int chunks, int n_slots);
int pw_input(const PwInput &a, hipStream_t st);

// BatchNorm statistics -> affine.  One block per (slot, 64 channels).
struct BnFinalize {
    const float *part_sum = nullptr, *part_sq = nullptr;   // [Q * chunks, C] chunk mean, chunk M2
    int chunk_rows = 512;
    int uniform_rows = 0;          // > 0: every window has this many rows (no win_off look-ups in the loops)
    const int *win_off = nullptr;
    int Q = 0, chunks = 1, n_slots = 1, C = 0;
    const float *gamma = nullptr, *beta = nullptr;
    float eps = 1e-5f;
    float *scale = nullptr, *shift = nullptr;   // [n_slots, C]   y = z * scale + shift
    float *mean = nullptr, *invstd = nullptr;   // [n_slots, C]   (saved for backward)
    float *stat_mean = nullptr, *stat_uvar = nullptr;   // [n_slots, C] batch mean / unbiased var for the running update
    const int *part_rows = nullptr; // explicit rows per partial [Q * chunks] (instead of win_off / chunk_rows)
    float *merge_ws = nullptr;     // optional scratch, bn_finalize_merge_floats(n_slots, C) floats: enables the two-stage form
    // global-batch form (sync_bn.hip): the partial arrays are `gather_ranks` segments of `gather_seg` floats, one per rank, each
    // {mean [n_slots, C], M2 [n_slots, C], rows [n_slots] as int}; partial (rank, slot) belongs to slot `slot`
    int gather_ranks = 0;
    long gather_seg = 0;
};
// ---- global-batch BatchNorm across data-parallel ranks (sync_bn.hip; ampnet_set_collective) ----
bool sync_bn_on();
int sync_bn_world();
int sync_bn_gather(size_t seg_floats, float **local_seg, float **gathered);
int sync_bn_exchange(int op, float *send, float *recv, size_t n_floats, hipStream_t st);
int sync_bn_bwd_constants(const float *slot_ab, const int *win_off, int Q, int n_slots, int uniform_rows, int C, const float *gamma,
                          const float *scale, const float *mean, const float *invstd, float *P1, float *P2, float *P3, hipStream_t st);
int sync_bn_fc_apply(const float *da, const float *z, const float *scale, const float *shift, const float *mean, const float *invstd,
                     const float *slot_ab, int n_slots, int per, int C, float *g, hipStream_t st);
inline size_t bn_finalize_merge_floats(int n_slots, int C) { return (size_t)n_slots * 16 * (2 * (size_t)C + 1); }
int bn_finalize(const BnFinalize &a, hipStream_t st);

// eval mode: scale/shift from running statistics for a list of layers, one launch
struct BnFoldItem {
    const float *gamma, *beta, *rmean, *rvar;
    float *scale, *shift;
    int C;
};
int bn_fold(const BnFoldItem *items_host, int n_items, float eps, hipStream_t st);

// train mode: running statistics update for a list of layers, sequential over slots like W encoder calls
struct BnRunItem {
    const float *stat_mean, *stat_uvar;   // [n_slots, C]
    float *rmean, *rvar;
    int C, n_slots;
};
int bn_running_update(const BnRunItem *items_host, int n_items, float momentum, hipStream_t st);

// MaxPool1d over each window after BatchNorm + ReLU:
//   pooled[orow(q), c] = relu(scale * extreme + shift), extreme = max for scale >= 0 else min (pw_gemm tracked the one
//   sign(gamma) = sign(scale) asks for), arg[q, c] = row index of that extreme
struct PoolFinalize {
    const float *part_max = nullptr;
    const int *part_amax = nullptr;
    const float *scale = nullptr, *shift = nullptr;   // [n_slots, C]
    int Q = 0, chunks = 1, n_slots = 1, C = 0;
    int out_slot_major = 0;       // orow(q) = (q % n_slots) * (Q / n_slots) + q / n_slots, else q
    float *pooled = nullptr;      // [Q, C]
    int *arg = nullptr;           // [Q, C] or nullptr
    float *zext = nullptr;        // [Q, C] or nullptr: the pre-BatchNorm extreme itself (row q)
    // train mode, C == 256: the pooled layer's BatchNorm is finished HERE from the per-workgroup partials of the layer's GEMM (as
    // PwGemm.pfin_*: no bn_finalize launch in between); scale / shift above are then OUTPUTS like the other arrays
    const float *pfin_sum = nullptr, *pfin_sq = nullptr;
    const int *pfin_rows = nullptr;
    int pfin_parts = 0;
    const float *pfin_gamma = nullptr, *pfin_beta = nullptr;
    float pfin_eps = 1e-5f;
    float *pfin_scale = nullptr, *pfin_shift = nullptr, *pfin_mean = nullptr, *pfin_invstd = nullptr, *pfin_smean = nullptr, *pfin_suvar = nullptr;
};
int pool_finalize(const PoolFinalize &a, hipStream_t st);

// small elementwise helpers
int add_identity(float *T, int n_mats, int k, hipStream_t st);                    // T[m] += I_k
int fill_i32_ramp(int *dst, int n, int step, hipStream_t st);                     // dst[i] = i * step

// dropout hash shared by kernels (lowbias32), restated in oracle/ampnet_oracle.py:keep_mask
__host__ __device__ inline uint32_t mix32(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
__host__ __device__ inline uint32_t drop_base(uint32_t seed, uint32_t stream) { return mix32(seed + stream * 0x9E3779B9U); }
__host__ __device__ inline uint32_t drop_threshold(float p)
{
    double t = (double)p * 4294967296.0;
    return t >= 4294967295.0 ? 0xFFFFFFFFU : (uint32_t)t;
}

}  // namespace ampnet

// =====================================================================================================
// backward kernels (pw_bwd.hip, bwd_misc.hip)
// =====================================================================================================
namespace ampnet {

// How the gradient wrt a layer's PRE-BatchNorm output z_l [rows, C] is produced on the fly:
//   g = dy * P1[slot][c] + z * P2[slot][c] + P3[slot][c]          (BatchNorm backward folded into three constants)
//   dy dense : dy[rows, C] = dL/d(bn output) already masked by ReLU/dropout
//   dy sparse: dy[row][c] = (arg[q][c] == row) ? dpool[prow(q)][c] : 0   (backward of MaxPool1d)
//   P1 == nullptr: g = dy (layers with no BatchNorm behind them)
struct GradSrc {
    const float *dy = nullptr;     // [rows, C] or nullptr (sparse)
    const float *z = nullptr;      // [rows, C] (needed when P1 != nullptr)
    const int *arg = nullptr;      // [Q, C]
    const float *dpool = nullptr;  // [Q, C]
    int dpool_slot_major = 0;
    const float *P1 = nullptr, *P2 = nullptr, *P3 = nullptr;   // [n_slots, C]
    int act = 0;                   // 1: g = relu(z * P2 + P3) -- a recomputed forward activation used as the operand
    int C = 0;
    int z_bf16 = 0;                // z is a bf16 tensor (activation storage of precision mode 3; fused bf16 kernels only)
};

// How the forward activation a_{l-1} [rows, C] is recomputed: relu(z * s + t) (+ dropout), or z itself
struct ActSrc {
    const float *z = nullptr;      // [rows, C]
    const float *s = nullptr, *t = nullptr;   // [n_slots, C] or nullptr = identity
    float drop_p = 0.f;
    uint32_t drop_seed = 0;
    int C = 0;
    int z_bf16 = 0;                // z is a bf16 tensor (precision mode 3)
};

// data gradient: out[row, j] = sum_k g[row, k] * W[k, j]  (+ add[row, j]), then optionally the ReLU/dropout mask
// of layer l-1 and the partial sums its BatchNorm backward needs
struct PwDgrad {
    GradSrc g;                     // K = g.C in {64, 128, 256}
    const float *W = nullptr;      // shared torch weight [K = cout_l][ldw] (row k, column j); per window: see below
    int ldw = 0;
    long w_win_stride = 0;         // != 0: per-window matrix T[pidx][j][k] (the bmm transform), rows j, k contiguous
    int perwin_slot_major = 0;
    long w_slot_stride = 0;        // != 0: shared-layout weights per SLOT at W + slot * w_slot_stride
    const float *bias_slot = nullptr;   // [n_slots, cp] added to every row of the slot
    const int *rowmap = nullptr;   // [rows]: -1 or an index into srows whose row is added (sparse max-pool part)
    const float *srows = nullptr;  // [*, cp]
    const float *add = nullptr;    // [rows, cp] or nullptr
    ActSrc prev;                   // prev.z == nullptr: raw output, no mask, no partial sums
    const float *prev_mean = nullptr, *prev_invstd = nullptr;   // [n_slots, cp]
    float *out = nullptr;          // [rows, cp]
    int cp = 0;                    // output columns (<= 128 per launch block, any multiple of 32 up to 256)
    float *part_a = nullptr, *part_b = nullptr;   // [Q * part_chunks, cp]: sum dy, sum dy * zhat of layer l-1
    int part_chunks = 0;           // partial slots per window (0 = chunks); > chunks leaves room for sparse_fix's slot
    const int *win_off = nullptr;
    int Q = 0, n_slots = 1, chunk_rows = 512, chunks = 1;
    long rows_hint = 0;
};
int pw_dgrad(const PwDgrad &a, hipStream_t st);

// weight gradient per window chunk: dWpart[q * chunks + chunk][cx][cy] = sum_rows X[row, cx] * Y[row, cy]; X = GradSrc, Y = ActSrc.
// Optional dbpart[q][cx] = sum_rows X[row, cx].
struct PwWgrad {
    GradSrc x;
    ActSrc y;
    float *dWpart = nullptr;       // [Q][x.C][ldp] (ldp >= y.C)
    int ldp = 0;
    float *dbpart = nullptr;       // [Q][x.C] or nullptr
    const int *win_off = nullptr;
    int Q = 0, n_slots = 1;
    int chunk_rows = 1 << 30, chunks = 1;   // partial index = q * chunks + chunk; dWpart is [Q * chunks][x.C][ldp]
    long rows_hint = 0;
};
int pw_wgrad(const PwWgrad &a, hipStream_t st);

// Fused backward of one shared-weight layer z_l = a_{l-1} W^T with a_{l-1} = relu(bn(z_{l-1})) (or z_{l-1} itself):
// ONE pass over (dy_l, z_l, z_{l-1}) produces both the weight-gradient partials and dy_{l-1} with the BatchNorm-backward
// sums of layer l-1 (pw_bwd_fused.hip).  Persistent workgroups: grid = blocks_per_slot * n_slots, workgroup
// (slot, j) = blockIdx.x % n_slots, / n_slots walks a contiguous share of the slot's windows, so every partial
// (dWpart, dbpart, part_a / part_b) is indexed by blockIdx.x and belongs to exactly one slot (index % n_slots).
struct PwBwd {
    GradSrc g;                     // dense (dy, z, P1..P3 | P1 == nullptr: g = dy) or act (z, P2 = scale, P3 = shift); CX = g.C
    ActSrc prev;                   // z_{l-1} [rows, CY]; s == nullptr: identity (no mask, no sums); no dropout here
    const float *prev_mean = nullptr, *prev_invstd = nullptr;   // [n_slots, CY]
    const float *W = nullptr;      // [CX][ldw] torch layout (row = output channel of layer l, column = input channel)
    int ldw = 0;
    long w_slot_stride = 0;        // != 0: per-slot weights at W + slot * w_slot_stride
    long w_win_stride = 0;         // != 0: per-window matrix T[pidx(q)][CY][CX] (the bmm transform); needs items_per_block
    int perwin_slot_major = 0;     // pidx(q) = (q % n_slots) * (Q / n_slots) + q / n_slots instead of q
    const float *bias_slot = nullptr;   // [n_slots, CY] added to every dgrad row of the slot
    const float *add = nullptr;    // [rows, CY] added to the dgrad output
    float *out = nullptr;          // [rows, CY] dy_{l-1} (masked)
    float *dWpart = nullptr;       // [grid][CX][CY]
    float *dbpart = nullptr;       // [grid][CX] or nullptr: sum of g
    float *part_a = nullptr, *part_b = nullptr;   // [grid][CY] or nullptr
    const int *win_off = nullptr;
    int Q = 0, n_slots = 1, max_rows = 0;
    int blocks_per_slot = 0;       // filled by pw_bwd_blocks()
    int items_per_block = 0;       // > 0: fixed share of (window, pw_bwd_item_rows()-row chunk) items per workgroup; a divisor of the
                                   // chunks per window keeps every workgroup inside one window (per-window dbpart sums)
    long rows_hint = 0;
    // Consumer-side BatchNorm-backward constants (fp32 kernel; no bn_bwd_finalize launch between two layers): fin_part_a != nullptr ->
    // every workgroup sums the fin_parts partials (sum dy, sum dy zhat) [fin_parts, CX] of ITS slot (partial i belongs to slot
    // i % n_slots) that the producer of g.dy left, in a fixed order, and forms P1..P3 itself (g.P1..P3 are then outputs: the first
    // workgroup of every slot writes them and slot_ab for the parameter gradients).  The partial arrays must not be the ones this
    // launch writes (part_a / part_b): the callers alternate two regions.
    const float *fin_part_a = nullptr, *fin_part_b = nullptr;
    int fin_parts = 0;
    int fin_rows = 0;              // rows of one slot (uniform windows)
    const float *fin_gamma = nullptr, *fin_mean = nullptr, *fin_invstd = nullptr;   // [CX], [n_slots, CX] x2 of the layer g belongs to
    float *fin_P1 = nullptr, *fin_P2 = nullptr, *fin_P3 = nullptr, *fin_slot_ab = nullptr;
    int dbg_row_wrap = 0;          // TIMING EXPERIMENT ONLY (AMPNET_PWBWD_ROWWRAP=n): the staging loads read row % n -- wrong results, cache-resident inputs
};
int pw_bwd_blocks(int Q, int n_slots, int max_rows);     // blocks_per_slot for this shape (grid = that * n_slots)
bool pw_bwd_supported(int cx, int cy);
int pw_bwd_item_rows();
int pw_bwd_fused(const PwBwd &a, hipStream_t st);
int pw_bwd_fused_bf16(const PwBwd &a, hipStream_t st);     // the same pass with bf16 MFMA operands (pw_bwd_bf16.hip); pw_bwd_fused dispatches
bool pw_bwd_x3_supported(const PwBwd &a);                  // precision mode 4: the shapes pw_bwd_x3.hip is built for (128 x 128, Gram or dense)
int pw_bwd_fused_x3(const PwBwd &a, hipStream_t st);       // the same pass with three-term bf16 split operands (fp32 results); pw_bwd_fused dispatches

// dst[i] (= or +=) sum_q part[q * stride + i], i < n, fixed order; dst row-remap for strided destinations:
// element i = (r, c) with c < cols -> dst[r * ld_dst + c]
int reduce_windows(const float *part, int Q, long stride, int rows, int cols, int ld_part, float *dst, int ld_dst, int accumulate,
                   hipStream_t st);
// the same reduction (overwrite form) for up to REDUCE_MULTI_MAX (partials, destination) pairs in ONE launch
constexpr int REDUCE_MULTI_MAX = 12;
struct ReduceItem {
    const float *part;
    int Q;
    long stride;
    int rows, cols, ld_part;
    float *dst;
    int ld_dst;
};
int reduce_windows_multi(const ReduceItem *items, int n, hipStream_t st);

// BatchNorm backward constants of one layer from the partial sums of pw_dgrad / head_out_bwd / pool_bwd:
//   per slot: A = sum dy, Bs = sum dy * zhat  ->  P1 = s, P2 = -s * invstd * Bs / n, P3 = -s * A / n - P2 * mean,
//   slot_ab[slot][c] = (A, Bs) for the parameter gradients (dbeta = sum_slots A, dgamma = sum_slots Bs)
struct BnBwdFinalize {
    const float *part_a = nullptr, *part_b = nullptr;   // [Q * chunks, C]
    const int *win_off = nullptr;                       // rows per slot are counted from it
    int uniform_rows = 0;                               // > 0: every window has this many rows
    int Q = 0, chunks = 1, n_slots = 1, C = 0;
    int part_Q = 0;                                     // > 0: the partial arrays hold part_Q * chunks rows (index % n_slots = slot)
    const float *gamma = nullptr, *mean = nullptr, *invstd = nullptr;   // gamma [C]; mean / invstd [n_slots, C]
    float *P1 = nullptr, *P2 = nullptr, *P3 = nullptr;  // [n_slots, C]
    float *slot_ab = nullptr;                           // [n_slots, C, 2]
};
int bn_bwd_finalize(const BnBwdFinalize &a, hipStream_t st);

struct BnGradItem {
    const float *slot_ab;   // [n_slots, C, 2]
    float *dgamma, *dbeta;  // [C]
    int C, n_slots;
};
int bn_param_grads(const BnGradItem *items_host, int n, hipStream_t st);

// MaxPool backward + the BatchNorm-backward constants of the pooled layer in one kernel
struct PoolBwd {
    const float *d_pooled = nullptr;   // [Q, C] grad wrt pooled (post-ReLU) activations, row = prow(q)
    int slot_major = 0;
    const int *arg = nullptr;          // [Q, C]
    const float *zext = nullptr;       // [Q, C] pre-BN value of the pooled layer at its argmax row (pool_finalize)
    const float *scale = nullptr, *shift = nullptr, *mean = nullptr, *invstd = nullptr;   // [n_slots, C]
    const int *win_off = nullptr;
    int Q = 0, n_slots = 1, C = 256;
    float *dpm = nullptr;              // [Q, C] masked pooled grads, row = prow(q)
    float *P1 = nullptr, *P2 = nullptr, *P3 = nullptr, *slot_ab = nullptr;
};
int pool_bwd(const PoolBwd &a, hipStream_t st);

// ---- backward of a max-pooled layer without its [rows, 256] output ------------------------------------------------
// With z = W a, dz = P1 dy + P2 z + P3 and dy non-zero only on the argmax rows:
//   dgrad:  dz W = a G + c0 + S,   G = W^T diag(P2) W (per slot),  c0 = P3 W,  S = rows scattered from (P1 dy) W
//   wgrad:  dz^T a = diag(P2) W Gram + P3 (x) asum + sparse,   Gram = a^T a, asum = sum a (per slot)
struct SparseRows {        // S rows of one layer: srows[q * C + i][:] and rowmap[row] = q * C + i
    const int *arg = nullptr;           // [Q, C]
    const float *dpm = nullptr;         // [Q, C] masked pooled gradient (row prow(q))
    int slot_major = 0;
    const float *P1 = nullptr;          // [n_slots, C]
    const float *W = nullptr;           // [C, cp]
    int Q = 0, n_slots = 1, C = 256, cp = 128;
    float *srows = nullptr;             // [Q * C, cp] merged rows of window q at [q * C + i], i < srow_cnt[q]
    int *srow_row = nullptr;            // [Q * C] the row each merged row belongs to
    int *srow_cnt = nullptr;            // [Q]
};
int sparse_rows(const SparseRows &a, hipStream_t st);
// adds the merged sparse rows into the masked data gradient and writes their share of the BatchNorm-backward sums
// into partial slot `slot_idx` of every window:  out[row] += mask * S;  part[(q * part_chunks + slot_idx)] = sums
struct SparseFix {
    const float *srows = nullptr;
    const int *srow_row = nullptr, *srow_cnt = nullptr;
    const float *z_prev = nullptr, *s_prev = nullptr, *t_prev = nullptr, *mean_prev = nullptr, *invstd_prev = nullptr;
    int Q = 0, n_slots = 1, C = 256, cp = 128;
    float *out = nullptr;               // [rows, cp]
    float *part_a = nullptr, *part_b = nullptr;
    int part_chunks = 1, slot_idx = 0;
};
int sparse_fix(const SparseFix &a, hipStream_t st);
// sparse_rows + sparse_fix in one kernel, no [Q * C, cp] intermediate (the fused backward uses this one)
struct SparseScatter {
    int z_bf16 = 0;                     // z_prev is a bf16 tensor (precision mode 3)
    const int *arg = nullptr;           // [Q, C]
    const float *dpm = nullptr;         // [Q, C]
    int slot_major = 0;
    const float *P1 = nullptr, *W = nullptr;   // [n_slots, C], [C, cp]
    const float *z_prev = nullptr, *s_prev = nullptr, *t_prev = nullptr, *mean_prev = nullptr, *invstd_prev = nullptr;
    int Q = 0, n_slots = 1, C = 256, cp = 128;
    float *out = nullptr;               // [rows, cp]
    float *part_a = nullptr, *part_b = nullptr;
    int part_chunks = 1, slot_idx = 0;
};
int sparse_scatter(const SparseScatter &a, hipStream_t st);
// G[s][j][k] = sum_c W[c][j] P2[s][c] W[c][k];  c0[s][k] = sum_c P3[s][c] W[c][k]
int slot_mats(const float *W, const float *P2, const float *P3, int n_slots, int C, int cp, float *G, float *c0, hipStream_t st);
// out[s][e] = sum over windows q = s (mod n_slots), chunks: part[(q * chunks + ch) * n_el + e]
int reduce_slots(const float *part, int Q, int chunks, int n_slots, int n_el, float *out, hipStream_t st);
// two such reductions over the same (Q, chunks, n_slots) in one launch
int reduce_slots2(const float *part0, int n_el0, float *out0, const float *part1, int n_el1, float *out1, int Q, int chunks, int n_slots, hipStream_t st);
struct PooledWgrad {
    int z_bf16 = 0;                     // z_prev is a bf16 tensor (precision mode 3)
    const float *W = nullptr, *P2 = nullptr, *P3 = nullptr, *gram = nullptr, *asum = nullptr;   // [C,cp] [S,C] [S,C] [S,cp,cp] [S,cp]
    const int *arg = nullptr;           // [Q, C]
    const float *dpm = nullptr, *P1 = nullptr;
    int slot_major = 0;
    const float *z_prev = nullptr, *s_prev = nullptr, *t_prev = nullptr;   // [rows, cp], [S, cp]
    int Q = 0, n_slots = 1, C = 256, cp = 128;
    float *dW = nullptr;                // [C, cp]
    float *wgram = nullptr;             // optional scratch [S, C, cp]: W Gram[s] by one small-GEMM launch instead of a walk over Gram rows per thread
};
int pooled_wgrad(const PooledWgrad &a, hipStream_t st);

// C[M, N] = op(A) * op(B) (+ C if accumulate); row-major, small problems (T-Net FC layers, attention projections)
// (AMPNET_SGEMM_VALU=1 selects the VALU kernel sgemm_small_kernel for an A/B; the matrix-core kernel sgemm_mfma serves every caller,
// the baseline PointNets included: their fixtures are held to the reference's own float32-to-float64 distance, not to a summation order)
int sgemm_small(int transA, int transB, int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc,
                int accumulate, hipStream_t st);
// backward of a linear layer Y = X W^T on [rows, *] activations, both products in ONE launch: dW [n_out, n_in] = G^T X, dX [rows, n_in] = G W
struct LinBwdOpt {
    float *db = nullptr;                         // [n_out]: the bias gradient (column sums of G), taken inside the weight-gradient problem
    const float *dx_mul = nullptr;               // [rows, n_in] (same leading dimension as dX): dX is multiplied elementwise as it is written
                                                 // (the derivative of the activation in front of the layer)
};
int sgemm_linear_bwd(int rows, int n_out, int n_in, const float *G, int ldg, const float *X, int ldx, const float *W, int ldw, float *dW, int lddw,
                     float *dX, int lddx, hipStream_t st, const LinBwdOpt &o = LinBwdOpt());
int sgemm_wgrad_bias(int rows, int n_out, int n_in, const float *G, int ldg, const float *X, int ldx, float *dW, int lddw, float *db, hipStream_t st);
// n_out in the thousands: dX's K range split over `splits` problems of the launch + a fixed-order reduction; scratch [splits, rows, n_in]
int sgemm_linear_bwd_ksplit(int rows, int n_out, int n_in, const float *G, int ldg, const float *X, int ldx, const float *W, int ldw, float *dW,
                            int lddw, float *dX, int lddx, float *scratch, int splits, hipStream_t st, const LinBwdOpt &o = LinBwdOpt());

}  // namespace ampnet
