/* ORACLE -- test infrastructure, not product code.
 *
 * Plain-C restatement of the reference's greedy farthest-point sampling
 * (utils/utils.py:889-933 of marionacaros/3D-semantic-segmentation-AMP-Net; see oracle/fps_oracle.py
 * for the line-by-line semantics).  Compiled with -ffp-contract=off: the reference computes
 * ((dx*dx + dy*dy) + dz*dz) in float32 with one rounding per operation, so no FMA contraction.
 *
 * Used by tests/ (checker for the HIP kernel at sizes numpy is slow for) and by bench.py's
 * cpu_baseline leg.  Never linked into the product library.
 */
#include <stdint.h>
#include <stdlib.h>

/* xyz: [n, ld] float32 row-major, columns 0..2 are x,y,z.  idx: [s] int32 out.
 * returns 0, or -1 on bad arguments. */
int fps_oracle_f32(const float *xyz, int n, int ld, int s, int32_t *idx)
{
    if (!xyz || !idx || n <= 0 || ld < 3 || s <= 0 || s > n) return -1;
    float *dist = (float *)malloc(sizeof(float) * (size_t)n);
    if (!dist) return -2;
    for (int j = 0; j < n; ++j) dist[j] = __builtin_inff();
    int last = 0;
    idx[0] = 0;
    dist[0] = -1.0f;
    for (int i = 1; i < s; ++i) {
        const float lx = xyz[(size_t)last * ld + 0];
        const float ly = xyz[(size_t)last * ld + 1];
        const float lz = xyz[(size_t)last * ld + 2];
        float best = -2.0f;
        int besti = 0;
        for (int j = 0; j < n; ++j) {
            float dj = dist[j];
            if (dj < 0.0f) continue;                 /* already picked */
            const float dx = lx - xyz[(size_t)j * ld + 0];
            const float dy = ly - xyz[(size_t)j * ld + 1];
            const float dz = lz - xyz[(size_t)j * ld + 2];
            const float d = (dx * dx + dy * dy) + dz * dz;
            if (d < dj) dj = d;                      /* np.minimum */
            dist[j] = dj;
            if (dj > best) { best = dj; besti = j; } /* first maximum wins */
        }
        idx[i] = besti;
        dist[besti] = -1.0f;
        last = besti;
    }
    free(dist);
    return 0;
}
