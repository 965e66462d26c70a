// encoder.h -- parameter tables and workspace layout shared by the encoder forward and backward.
#pragma once
#include "kernels.h"

namespace ampnet {

// index tables: ORDER = params.ENC_PARAMS / ENC_BUFFERS of the Python package (tests/test_abi.py checks it)
enum TnetParam { TP_CONV1 = 0, TP_CONV2, TP_CONV3, TP_BN1_W, TP_BN1_B, TP_BN2_W, TP_BN2_B, TP_BN3_W, TP_BN3_B,
                 TP_BN4_W, TP_BN4_B, TP_BN5_W, TP_BN5_B, TP_FC1, TP_FC2, TP_FC3_W, TP_FC3_B, TP_COUNT };
enum EncParam {
    EP_IT = 0, EP_FT = TP_COUNT, EP_CONV1 = 2 * TP_COUNT, EP_CONV2, EP_CONV3, EP_CONV4, EP_CONV5, EP_CONV6,
    EP_BN1_W, EP_BN1_B, EP_BN2_W, EP_BN2_B, EP_BN3_W, EP_BN3_B, EP_BN4_W, EP_BN4_B, EP_BN5_W, EP_BN5_B,
    EP_BN6_W, EP_BN6_B, EP_COUNT
};
// buffers: (running_mean, running_var) pairs: IT bn_1..5, FT bn_1..5, main bn_1..6
enum { EB_IT = 0, EB_FT = 10, EB_MAIN = 20, EB_COUNT = 32 };

// BatchNorm layers of the encoder in execution order
enum EncBn { BN_T1 = 0, BN_T2, BN_T3, BN_T4, BN_T5, BN_C1, BN_C2, BN_F1, BN_F2, BN_F3, BN_F4, BN_F5,
             BN_C3, BN_C4, BN_C5, BN_C6, BN_ENC_COUNT };

struct BnSlot {              // per-layer derived arrays, each [n_slots, C]
    float *scale, *shift, *mean, *invstd, *smean, *suvar;
    int C;
};

struct EncShape {
    int Q, n_slots, R, max_rows, train;
    int chunk_rows, chunks;            // point layers
    int x_chunk_rows, x_chunks;        // point layers that run on the split kernels (precision mode 4): a block of rows is one wave's
    int fc_rows, fc_chunk_rows, fc_chunks;   // T-Net FC layers: n_slots windows of Q / n_slots rows
};

struct EncWs {
    // pre-BatchNorm activations, [R, C]
    float *z_t1, *z_t2, *z_t3, *z_c1, *z_c2, *z_f1, *z_f2, *z_f3, *z_c3, *z_c4, *z_c5, *z_c6;
    // pooled / FC activations, [Q, C] (rows slot-major), transforms
    float *pool_t, *z_tf1, *z_tf2, *T3, *pool_f, *z_ff1, *z_ff2;
    int *arg_t, *arg_f, *arg_c;        // [Q, 256] row index of the pooled extreme
    float *zext_t, *zext_f, *zext_c;   // [Q, 256] its pre-BatchNorm value (all the backward needs of the 256-channel layers)
    int *part_rows;                    // [1024] rows covered by each per-workgroup statistics partial (pw_gemm_stat_plan)
    float *merge;                      // two-stage bn_finalize scratch
    float *part_sum, *part_sq, *part_max;   // [Q * chunks, 256]
    int *part_amax;
    BnSlot bn[BN_ENC_COUNT];
    size_t bytes;
};

EncShape enc_shape(int Q, int n_slots, int R, int max_rows, int train);
// the backward keeps two alternating regions of BatchNorm-backward partials in part_sum / part_sq: one per fused workgroup (<= 320 + n_slots)
// + one per window (the scattered rows of the pooled layers), 256 channels wide
inline size_t enc_bwd_part_region_floats(const EncShape &s) { return (size_t)(320 + s.n_slots + s.Q) * 256; }
// the forward's per-workgroup statistics partials: <= 1024 lanes (pw_input) rounded up to whole slots, 256 channels wide, two regions
inline size_t enc_fwd_part_region_rows(const EncShape &s) { return (size_t)1024 + 2 * (size_t)s.n_slots; }
inline size_t enc_fwd_part_region_floats(const EncShape &s) { return enc_fwd_part_region_rows(s) * 256; }
// carve the workspace; base may be nullptr to only compute ws.bytes
void enc_carve(const EncShape &s, void *base, EncWs &ws);

}  // namespace ampnet
