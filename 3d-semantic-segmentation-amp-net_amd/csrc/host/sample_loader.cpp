// Host side of the input pipeline (include/ampnet_host.h): LidarKmeansDataset.__getitem__ in one pass over the sample.
// The numpy statement of the same steps (pointNet/datasets.py in this package, reference pointNet/datasets.py:330-458) makes six
// temporaries of the whole [n, 13, w] array per sample (~1 ms in a DataLoader worker); 64 samples per 10 ms train step then need
// more worker time than the GPU box's CPU quota has.  Built with g++ (-ffp-contract=off: the float32 roundings are numpy's).
#include "../../../include/ampnet_host.h"

#include <cstring>
#include <vector>

#include <fcntl.h>
#include <unistd.h>

extern "C" int ampnet_host_abi_version(void) { return AMPNET_HOST_ABI_VERSION; }

namespace {
// class code -> label (0 .. 4) or NOISE, for the exact integer codes 0 .. 31; every other value (fractional, negative, larger, NaN) is an
// ordinary background point, as in the numpy statement (`codes == 15` etc. are exact float comparisons)
constexpr signed char NOISE = -1;
struct CodeTable {
    signed char t[32];
    constexpr CodeTable() : t{}
    {
        for (int i = 0; i < 32; ++i) t[i] = 0;
        t[15] = 1; t[14] = NOISE /* 14 is deleted as noise before it could become label 2 (reference datasets.py:339-350 runs first) */;
        t[3] = 3; t[4] = 3; t[5] = 4;
        t[30] = NOISE; t[7] = NOISE; t[2] = NOISE; t[8] = NOISE; t[13] = NOISE;
    }
};
constexpr CodeTable CODES;
inline signed char classify(float c)
{
    const int k = (int)c;                         // (out-of-range / NaN conversions are caught by the round-trip test below)
    return (c >= 0.f && c < 32.f && (float)k == c) ? CODES.t[k] : (signed char)0;
}
}  // namespace

template <typename L>
static long kmeans_sample(const float *pc, long n, int feats, int w, float *pts_out, L *labels_out, float *cent_out)
{
    float sx[64], sy[64];
    for (int j = 0; j < w; ++j) sx[j] = sy[j] = 0.f;
    const long row = (long)feats * w;
    long kept = 0;
    for (long i = 0; i < n; ++i) {
        const float *p = pc + i * row;
        const float *code = p + 3L * w;
        signed char cls[64];
        int drop = 0;
        for (int j = 0; j < w; ++j) {
            cls[j] = classify(code[j]);
            drop |= cls[j];                       // NOISE = -1 sets the sign bit, labels 0 .. 4 never do
        }
        if (drop < 0) continue;
        float *o = pts_out + kept * 9L * w;
        L *l = labels_out + kept * (long)w;
        for (int j = 0; j < w; ++j) {
            const float x = p[j] * 2.f - 1.f, y = p[w + j] * 2.f - 1.f;      // v * 2 is exact, one rounding in the subtraction
            o[j] = x;
            o[w + j] = y;
            sx[j] += x;                                                      // numpy's mean(0) of a [rows, w >= 2] view: rows added in order, float32
            sy[j] += y;
            l[j] = (L)cls[j];
        }
        for (int j = 0; j < w; ++j) o[2 * w + j] = p[2 * w + j];             // HAG
        for (int j = 0; j < 6 * w; ++j) o[3 * w + j] = p[4 * w + j];         // I, R, G, B, NIR, NDVI
        ++kept;
    }
    if (cent_out) {
        // numpy's _mean: the float32 sum divided by the row count as a float32 (a python int is a weak scalar)
        const float cnt = (float)kept;
        for (int j = 0; j < w; ++j) {
            cent_out[j] = sx[j] / cnt;
            cent_out[w + j] = sy[j] / cnt;
        }
    }
    return kept;
}

extern "C" long ampnet_host_kmeans_sample_f32(const float *pc, long n, int feats, int w, float *pts_out, long long *labels_out, float *cent_out)
{
    if (!pc || !pts_out || !labels_out || n < 0 || feats < 10 || w < 1 || w > 64) return -1;
    return kmeans_sample(pc, n, feats, w, pts_out, labels_out, cent_out);
}

extern "C" long ampnet_host_kmeans_file_ragged_f32(const char *path, long long byte_offset, long n, int feats, int w, float *pts_out,
                                                   signed char *lab_out, float *cent_out)
{
    if (!path || !pts_out || !lab_out || byte_offset < 0 || n < 0 || feats < 10 || w < 1 || w > 64) return -1;
    static thread_local std::vector<float> scratch;                          // reused: no fresh pages per sample
    const size_t count = (size_t)n * feats * w;
    if (scratch.size() < count) scratch.resize(count);
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return -2;
    size_t got = 0;
    const size_t want = count * sizeof(float);
    char *dst = reinterpret_cast<char *>(scratch.data());
    while (got < want) {
        const ssize_t r = pread(fd, dst + got, want - got, (off_t)byte_offset + (off_t)got);
        if (r <= 0) {
            close(fd);
            return -2;
        }
        got += (size_t)r;
    }
    close(fd);
    return kmeans_sample(scratch.data(), n, feats, w, pts_out, lab_out, cent_out);
}
