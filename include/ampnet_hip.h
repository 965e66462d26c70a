/* ampnet_hip.h -- C ABI of libampnet_hip.so: the MI355X (gfx950) implementation of the AMP-Net
 * per-window hot path of marionacaros/3D-semantic-segmentation-AMP-Net.
 *
 * The reference has no FFI: its boundary for this path is Python (nn.Module.forward signatures,
 * function signatures, state_dict keys).  Each entry point below names the reference interface it
 * replaces (paths relative to the reference root); the Python package
 * `3d-semantic-segmentation-amp-net_amd/` binds them with ctypes behind modules of the reference's
 * names (see INTEGRATION.md for the stub a maintainer of the reference would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host; caller owns every buffer;
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it and never synchronised;
 *   - return value 0 = ok, negative = error (AMPNET_E_*); ampnet_last_error() gives the text
 *     (thread-local).  No exception crosses the ABI;
 *   - the library keeps no state between calls except the thread-local error string.
 *   - float tensors are fp32, row-major, point-major: activations are [rows, channels].
 *
 * Window batching: the reference calls its encoder W times per step, each time on the B windows that
 * occupy cluster slot w (train_pointnet-attention.py:396-410); BatchNorm statistics are therefore
 * per slot.  Here all Q = B*W windows go through one launch sequence; window q = b*W + w (sample-major,
 * the order of lo_feats in the reference, train_pointnet-attention.py:415-417) and `n_slots` = W tells
 * the kernels that windows with equal q % n_slots share batch statistics.  Eval mode ignores slots.
 */
#ifndef AMPNET_HIP_H
#define AMPNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMPNET_ABI_VERSION 1

enum {
    AMPNET_OK = 0,
    AMPNET_E_ARG = -1,        /* bad shape / null pointer / unsupported size */
    AMPNET_E_LAUNCH = -2,     /* hipGetLastError() after a launch            */
    AMPNET_E_WORKSPACE = -3,  /* workspace too small                         */
    AMPNET_E_DEVICE = -4      /* not a gfx950 device / no device             */
};

int ampnet_abi_version(void);
const char *ampnet_last_error(void);

/* ---- a1: farthest-point sampling ---------------------------------------------------------------
 * replaces utils/utils.py:889-933 `fps(pc, n_samples)` (driver data_proc/sample_fps.py:23-31).
 * xyz: [n_clouds, n, ld] fp32, columns 0..2 = x,y,z (ld >= 3 lets the caller pass whole rows).
 * idx: [n_clouds, s] int32, selection order; idx[c][0] == 0 (utils.py:907-908).
 * Bit-exact with the reference: float32 ((dx*dx + dy*dy) + dz*dz), running minimum, first maximum.
 * 1 <= s <= n <= 16384.                                                                           */
int ampnet_fps_f32(const float *xyz, int n_clouds, int n, int ld, int s, int32_t *idx, void *stream);

/* rows gather: out[c][i][:] = src[c][idx[c][i]][:]  (the `pc[sample_inds]` of utils.py:933)         */
int ampnet_gather_rows_f32(const float *src, const int32_t *idx, int n_clouds, int n, int ld, int s,
                           float *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* AMPNET_HIP_H */
