// bwd_misc.h -- small backward kernels (bwd_misc.hip); internal.
#pragma once
#include "kernels.h"

namespace ampnet {

int fc_act(const float *z, const float *s, const float *t, int rows, int C, int per, float *act, hipStream_t st);
int fc_act_pair(const float *z0, const float *s0, const float *t0, int C0, float *a0, const float *z1, const float *s1, const float *t1, int C1, float *a1,
                int rows, int per, hipStream_t st);
int fc_bn_bwd(const float *da, const float *z, const float *scale, const float *shift, const float *mean, const float *invstd,
              int n_slots, int per, int C, float *g, float *slot_ab, hipStream_t st);
int colsum(const float *x, int rows, int C, float *out, hipStream_t st);
int axpy(const float *x, float alpha, size_t n, float *y, hipStream_t st);

struct PwInputWgrad {
    const float *x = nullptr;                 // [rows, 9]
    const float *dy = nullptr;                // [rows, 64]
    // the layer's pre-BatchNorm output z is RECOMPUTED from the staged x rows (3 or 9 FMAs per element, the arithmetic of pw_input) instead of
    // read back: the kernel sits on HBM, and z was half of its bytes
    const float *W = nullptr, *T = nullptr;   // as PwInput: mode 0 W [64, 3]; mode 1 W [64, 12], T [.., 3, 3] per window
    int mode = 0, perwin_slot_major = 0;
    // optional: this layer's BatchNorm-backward constants formed in the kernel from the partial sums its producer left (as PwBwd.fin_*,
    // kernels.h); the first window of each slot also writes P1 / P2 / P3 / slot_ab for bn_param_grads
    const float *fin_part_a = nullptr, *fin_part_b = nullptr;
    int fin_parts = 0, fin_rows = 0;
    const float *fin_gamma = nullptr, *fin_mean = nullptr, *fin_invstd = nullptr;
    float *fin_P1 = nullptr, *fin_P2 = nullptr, *fin_P3 = nullptr, *fin_slot_ab = nullptr;
    const float *P1 = nullptr, *P2 = nullptr, *P3 = nullptr;   // [n_slots, 64]
    float *dWeff = nullptr;                   // [Q, 64, 9]
    const int *win_off = nullptr;
    int Q = 0, n_slots = 1;
};
int pw_input_wgrad(const PwInputWgrad &a, hipStream_t st);
int input_param_grads(const float *dWeff, const float *W, const float *T, int Q, int n_slots, int slot_major, int mode, float *dW,
                      float *dT, hipStream_t st);
// dst[p(q)] = (sum of the window's `chunks` partials)^T (+ add[p(q)]), 64 x 64, p(q) = the slot-major row of window q
int transpose64_slot_major(const float *src, float *dst, int Q, int n_slots, int chunks, int by_workgroup, const float *add, hipStream_t st);
int fill_f32(float *p, size_t n, float v, hipStream_t st);
int fill_f32_pair(float *p0, float v0, float *p1, float v1, size_t n, hipStream_t st);

}  // namespace ampnet
