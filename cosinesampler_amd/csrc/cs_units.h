// cs_units.h -- host functions that cross translation units of libcosine_sampler_hip.so.
// cs_abi.hip owns the C ABI and the path choice; the kernels of a path that is compiled on its own
// (cs_coherent.hip, cs_sort.hip) are reached through these launch functions.  Device code never crosses units.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "cs_kernels_direct.cuh"

namespace cs {
namespace coh {

constexpr int64_t MAX_SIZE_HOST = 32766;   // = cs::coh::MAX_SIZE (cs_coherent.cuh): cell coordinates are packed into 15 bits

// one backward stage on the coherent-points path (cs_coherent.cuh).  `cq` = padded channel count / 4 (1, 2, 4 or 8),
// `kernel` the blending kernel enum, `sdt` the stream element type (0 fp32, 1 half, 2 bfloat16); `acc` the zeroed
// channels-last accumulator [N][H*W][4*cq] that receives the input-shaped gradient (null in backward / bb: the gradient is
// not wanted, nothing is scattered).  Return: 0 or a hipError_t.
struct Launch {
    Dims d;
    Flags f;
    int kernel, sdt, cq;
    hipStream_t stream;
    bool nsum = false;   // CS_SUM_OVER_N: shared points and cotangents, per-point results summed over the tables (supported_nsum)
};
bool supported(const Launch &L);
bool supported_nsum(const Launch &L);
int forward(const Launch &L, const float *icl, const float *grid, const float *offset, void *output);
int backward(const Launch &L, const void *gOut, const float *icl, const float *grid, const float *offset, float *acc,
             float *grad_grid);
int bb(const Launch &L, const float *cG, const float *icl, const float *grid, const void *gOut, const float *offset,
       float *acc, float *gGrid, void *ggOut);   // grad_out_input absent (with it the stage stays on the general path)
int bbb(const Launch &L, const float *icl, const float *grid, const void *gOut, const float *cG, const float *hG,
        const void *hO, const float *offset, float *acc, void *ggOut);
void set_chunk(int samples_per_wave, int ablation_bits);   // tuning / experiments (cs_debug_coherent_tuning)

}  // namespace coh

namespace sort {
size_t workspace_bytes(int64_t P);
// order P points by the (tile, cell) of table 0 they fall into: perm[j] = index of the j-th point, sorted[j] = points[perm[j]]
int sort_points(int dim, const float *points, int64_t P, int64_t D, int64_t H, int64_t W, int padding_mode,
                int align_corners, int multicell, float *sorted_points, int32_t *perm, void *workspace,
                size_t workspace_bytes, hipStream_t stream);
// out[r][j] = in[r][index[j]] for `rows` arrays of P elements of `width` floats (cs_carry_points)
int carry_points(const float *in, float *out, const int32_t *index, int64_t rows, int64_t P, int width, hipStream_t stream);
// fraction-of-coherence measure: number of times the tile of consecutive points changes, per table 0
int count_tile_changes(int dim, const float *points, int64_t P, int64_t D, int64_t H, int64_t W, int padding_mode,
                       int align_corners, int multicell, uint32_t *count /* device, one word */, hipStream_t stream);
int sample_tile_changes(int dim, const float *points, int64_t P, int64_t D, int64_t H, int64_t W, int padding_mode,
                        int align_corners, int multicell, int segments, uint32_t *count, hipStream_t stream);
}  // namespace sort
}  // namespace cs
