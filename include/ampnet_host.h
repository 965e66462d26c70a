/* ampnet_host.h -- host-side (CPU) entry points of the input pipeline: what the DataLoader workers run per sample.
 * Plain C ABI, no HIP, no torch: libampnet_host.so is built with g++ and may be loaded in forked worker processes.
 *
 * These functions restate, in one pass over the sample, what the reference's Dataset.__getitem__ does with numpy
 * (pointNet/datasets.py:295-460 LidarKmeansDataset) -- the values are bit-identical to the numpy statement kept in
 * <package>/pointNet/datasets.py (tests/test_host_cpu.py compares the two on seeded files). */
#ifndef AMPNET_HOST_H
#define AMPNET_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define AMPNET_HOST_ABI_VERSION 1
int ampnet_host_abi_version(void);

/* One `kmeans_<name>.pt` sample, LidarKmeansDataset.__getitem__ for task == 'segmentation' (reference pointNet/datasets.py:330-458):
 *   pc        [n][feats][w] float32, feats >= 10, columns x, y, HAG, class code, I, R, G, B, NIR, NDVI, ...
 *   a point ROW is dropped when ANY of its w clusters carries a noise class code (30, 7, 2, 8, 13, 14) (:339-350);
 *   pts_out   [n_kept][9][w]: columns 0, 1, 2, 4 .. 9 of the surviving rows, x and y <- 2 v - 1 (float32) (:352-360);
 *   labels_out[n_kept][w] int64: class code -> 0 background, 1 tower (15), 2 lines (14), 3 low / medium vegetation (3, 4),
 *             4 high vegetation (5); anything that is not exactly one of those values is background (:449-458);
 *   cent_out  [2][w] (may be NULL): the mean over the surviving rows of the scaled x and y of each cluster, float32 sums in row
 *             order divided by n_kept (numpy's mean(0) of a [n_kept, w] float32 view) (:362-366).
 * pts_out / labels_out must hold n rows.  Returns the number of surviving rows (>= 0), or -1 on a bad argument. */
long ampnet_host_kmeans_sample_f32(const float *pc, long n, int feats, int w, float *pts_out, long long *labels_out, float *cent_out);

/* The same sample straight from its file into a slice of the RAGGED batch (collate_fns.collate_seq_ragged, RaggedBatch.pts / .lab):
 * `byte_offset` = where the [n][feats][w] float32 storage record starts inside the torch.save archive (the zip stores it uncompressed;
 * _safe_load.pt_tensor_header finds it), read with pread into a per-thread scratch buffer that is reused from call to call; labels leave
 * as int8 (what the device-side gather takes).  Returns the surviving rows, -1 on a bad argument, -2 when the file cannot be read. */
long ampnet_host_kmeans_file_ragged_f32(const char *path, long long byte_offset, long n, int feats, int w, float *pts_out, signed char *lab_out,
                                        float *cent_out);

#ifdef __cplusplus
}
#endif
#endif
