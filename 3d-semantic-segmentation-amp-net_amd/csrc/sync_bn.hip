// sync_bn.hip -- global-batch BatchNorm across data-parallel ranks (SURVEY section 8(e) option A), behind ampnet_set_collective().
//
// The reference is single-device: its BatchNorm sees the whole batch.  Under data parallelism every rank holds B / world samples;
// with a collective registered, every train-mode BatchNorm of the encoder / head launch sequences uses the statistics of the GLOBAL batch:
//   forward   per-slot (rows, mean, M2) of the rank  -> all-gather -> Chan merge of the world's partials  (bn_finalize)
//   backward  per-slot (sum dy, sum dy zhat, rows)    -> all-reduce -> BatchNorm-backward constants / FC gradients from the global sums
//             (bn_bwd_finalize, pool_bwd, fc_bn_bwd); the per-rank sums still feed the gamma / beta gradients, which the gradient
//             all-reduce adds up like every other parameter gradient.
// The exchange itself is the caller's: a C callback (Python: torch.distributed on views of the scratch buffer), called on the host
// between two launches and ordered on the stream it is given.  18 + 18 latency-bound collectives per step: off by default.
#include "bwd_misc.h"
#include "kernels.h"

namespace ampnet {

namespace {
ampnet_collective_fn g_fn = nullptr;
void *g_ctx = nullptr;
int g_rank = 0, g_world = 1;
float *g_scratch = nullptr;
size_t g_scratch_floats = 0;

constexpr size_t SEG_MAX = (size_t)AMPNET_SYNC_MAX_SLOTS * (2 * AMPNET_SYNC_MAX_CHANNELS + 1);   // floats of one rank's segment

__global__ void sync_pack_kernel(const float *__restrict__ slot_ab, const int *__restrict__ win_off, int Q, int n_slots, int uniform_rows, int C,
                                 float *__restrict__ comm)
{
    const int s = blockIdx.x;
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) comm[(size_t)s * 2 * C + i] = slot_ab[(size_t)s * 2 * C + i];
    if (threadIdx.x == 0) {
        const int per_slot = (Q - s + n_slots - 1) / n_slots;
        long rows = 0;
        if (uniform_rows > 0) rows = (long)per_slot * uniform_rows;
        else
            for (int i = 0; i < per_slot; ++i) rows += win_off[s + i * n_slots + 1] - win_off[s + i * n_slots];
        comm[(size_t)n_slots * 2 * C + s] = (float)rows;
    }
}

// P1 = s, P2 = -s invstd B / n, P3 = -s A / n - P2 mean with s = gamma invstd (bn_bwd_finalize_kernel / pool_bwd_kernel), global A, B, n
__global__ void sync_constants_kernel(const float *__restrict__ comm, const float *__restrict__ gamma, const float *__restrict__ scale,
                                      const float *__restrict__ mean, const float *__restrict__ invstd, int n_slots, int C, float *__restrict__ P1,
                                      float *__restrict__ P2, float *__restrict__ P3)
{
    const int s = blockIdx.x;
    const double n = (double)comm[(size_t)n_slots * 2 * C + s];
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const size_t o = (size_t)s * C + c;
        const double A = comm[o * 2 + 0], B = comm[o * 2 + 1];
        const double is = invstd[o];
        const double sc = scale ? (double)scale[o] : (double)gamma[c] * is;
        const double p2 = -sc * is * B / n;
        P1[o] = (float)sc;
        P2[o] = (float)p2;
        P3[o] = (float)(-sc * A / n - p2 * (double)mean[o]);
    }
}

// FC BatchNorm backward with the global means of dy and dy zhat (fc_bn_bwd_kernel's second pass)
__global__ void sync_fc_apply_kernel(const float *__restrict__ da, const float *__restrict__ z, const float *__restrict__ scale,
                                     const float *__restrict__ shift, const float *__restrict__ mean, const float *__restrict__ invstd,
                                     const float *__restrict__ comm, int n_slots, int per, int C, float *__restrict__ g)
{
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= (size_t)n_slots * per * C) return;
    const int row = (int)(i / C), c = (int)(i % C), s = row / per;
    const size_t so = (size_t)s * C + c;
    const float n = comm[(size_t)n_slots * 2 * C + s];
    const float an = comm[so * 2 + 0] / n, bn = comm[so * 2 + 1] / n;
    const float sc = scale[so], zv = z[i];
    const float dy = fmaf(zv, sc, shift[so]) > 0.f ? da[i] : 0.f;
    g[i] = sc * (dy - an - (zv - mean[so]) * invstd[so] * bn);
}
}  // namespace

// a registered collective is used whatever the world size: with one rank every exchange is the identity and the step equals the plain one
// (tests/test_rccl_gpu.py drives the RCCL path that way on a one-GPU box); the Python side only registers one when it wants the exchanges
bool sync_bn_on() { return g_fn != nullptr; }
int sync_bn_world() { return g_world; }

int sync_bn_gather(size_t seg_floats, float **local_seg, float **gathered)
{
    if (seg_floats > SEG_MAX) return fail(AMPNET_E_ARG, "sync BatchNorm: segment of %zu floats > %zu", seg_floats, SEG_MAX);
    *local_seg = g_scratch;
    *gathered = g_scratch + SEG_MAX;
    return AMPNET_OK;
}

int sync_bn_exchange(int op, float *send, float *recv, size_t n_floats, hipStream_t st)
{
    const int rc = g_fn(g_ctx, op, send, recv, n_floats, (void *)st);
    if (rc != 0) return fail(AMPNET_E_LAUNCH, "sync BatchNorm: the registered collective returned %d", rc);
    return AMPNET_OK;
}

// after a backward kernel wrote the rank's slot_ab [n_slots, C, 2]: all-reduce (A, B, rows) and return the comm buffer with the global sums
static int reduce_ab(const float *slot_ab, const int *win_off, int Q, int n_slots, int uniform_rows, int C, float **comm_out, hipStream_t st)
{
    const size_t n = (size_t)n_slots * (2 * C + 1);
    if (n > SEG_MAX) return fail(AMPNET_E_ARG, "sync BatchNorm: %d slots x %d channels exceed the scratch segment", n_slots, C);
    float *comm = g_scratch;
    hipLaunchKernelGGL(sync_pack_kernel, dim3(n_slots), dim3(256), 0, st, slot_ab, win_off, Q, n_slots, uniform_rows, C, comm);
    int rc = check_launch("sync_pack_kernel");
    if (rc != AMPNET_OK) return rc;
    rc = sync_bn_exchange(AMPNET_COLLECTIVE_ALLREDUCE_SUM, comm, comm, n, st);
    *comm_out = comm;
    return rc;
}

int sync_bn_bwd_constants(const float *slot_ab, const int *win_off, int Q, int n_slots, int uniform_rows, int C, const float *gamma,
                          const float *scale, const float *mean, const float *invstd, float *P1, float *P2, float *P3, hipStream_t st)
{
    float *comm = nullptr;
    int rc = reduce_ab(slot_ab, win_off, Q, n_slots, uniform_rows, C, &comm, st);
    if (rc != AMPNET_OK) return rc;
    hipLaunchKernelGGL(sync_constants_kernel, dim3(n_slots), dim3(256), 0, st, comm, gamma, scale, mean, invstd, n_slots, C, P1, P2, P3);
    return check_launch("sync_constants_kernel");
}

int sync_bn_fc_apply(const float *da, const float *z, const float *scale, const float *shift, const float *mean, const float *invstd,
                     const float *slot_ab, int n_slots, int per, int C, float *g, hipStream_t st)
{
    float *comm = nullptr;
    int rc = reduce_ab(slot_ab, nullptr, n_slots, n_slots, per, C, &comm, st);     // one "window" of `per` rows per slot
    if (rc != AMPNET_OK) return rc;
    const size_t n = (size_t)n_slots * per * C;
    hipLaunchKernelGGL(sync_fc_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, da, z, scale, shift, mean, invstd, comm, n_slots, per, C, g);
    return check_launch("sync_fc_apply_kernel");
}

}  // namespace ampnet

extern "C" size_t ampnet_collective_scratch_bytes(int world_size)
{
    if (world_size < 1) return 0;
    return ((size_t)world_size + 1) * ampnet::SEG_MAX * sizeof(float);
}

extern "C" int ampnet_set_collective(ampnet_collective_fn fn, void *ctx, int rank, int world_size, void *scratch, size_t scratch_bytes)
{
    using namespace ampnet;
    if (!fn) {
        g_fn = nullptr; g_ctx = nullptr; g_rank = 0; g_world = 1; g_scratch = nullptr; g_scratch_floats = 0;
        return AMPNET_OK;
    }
    AMPNET_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, "ampnet_set_collective: rank %d of %d", rank, world_size);
    AMPNET_REQUIRE(scratch && scratch_bytes >= ampnet_collective_scratch_bytes(world_size), "ampnet_set_collective: scratch of %zu B < %zu B",
                   scratch_bytes, ampnet_collective_scratch_bytes(world_size));
    g_fn = fn; g_ctx = ctx; g_rank = rank; g_world = world_size;
    g_scratch = reinterpret_cast<float *>(scratch);
    g_scratch_floats = scratch_bytes / sizeof(float);
    return AMPNET_OK;
}
