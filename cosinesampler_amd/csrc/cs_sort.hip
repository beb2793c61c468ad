// cs_sort.hip -- ordering a point set by cell, once, at set-up (cs2d_sort_points / cs3d_sort_points) and measuring how
// coherent a given order is (cs_points_tile_changes).  Not on the per-step path: PIXEL-style callers draw their
// collocation points once and re-use them every step (reference test/test_2d.py:28-38), so the order is theirs to
// choose.  The key pass is ours; the sort itself is rocPRIM's device radix sort (stable: equal cells keep the caller's
// order, so the result is reproducible).
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "cs_math.cuh"
#include "cs_units.h"

namespace cs {
namespace sort {
namespace {

constexpr int TS = 8;   // = cs::coh::TS: the tiles the coherent kernels anchor their windows on
constexpr uint64_t KEY_LAST = ~0ull;

struct KeyDims {
    int dim, size[3], nt[3];
    Flags f;
};

// (tile, quad of 2 x 2 (x 2) cells inside the tile, cell inside the quad) of the point in table 0 (offset 0), each
// row-major: the coherent kernels reduce runs of equal quad and hold one row of a tile's quads on chip
// (cs_coherent.cuh); points that touch no node go last
__device__ __forceinline__ uint64_t cell_key(const float *pt, const KeyDims &k, bool tile_only) {
    uint64_t tile = 0, quad = 0, sub = 0;
    for (int j = k.dim - 1; j >= 0; --j) {
        float mu;
        const float i = source_index(pt[j], k.size[j], k.f.pad, k.f.align, 0.0f, k.f.multicell, mu);
        if (!(i > -1073741824.0f && i < 1073741824.0f)) return KEY_LAST;
        const int u = (int)floorf(i) + 1;
        if (u < 0 || u > k.size[j]) return KEY_LAST;
        tile = tile * (uint64_t)k.nt[j] + (uint64_t)(u / TS);
        quad = quad * (TS / 2) + (uint64_t)((u % TS) >> 1);
        sub = sub * 2 + (uint64_t)(u & 1);
    }
    return tile_only ? tile : tile * (TS * TS * TS) + quad * 8 + sub;
}

__global__ __launch_bounds__(256) void make_keys(const float *__restrict__ pts, int64_t P, KeyDims k,
                                                 uint64_t *__restrict__ keys, int32_t *__restrict__ idx) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    keys[p] = cell_key(pts + p * k.dim, k, false);
    idx[p] = (int32_t)p;
}
__global__ __launch_bounds__(256) void gather_points(const float *__restrict__ pts, const int32_t *__restrict__ perm,
                                                     int64_t P, int dim, float *__restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const int64_t s = perm[p];
    for (int j = 0; j < dim; ++j) out[p * dim + j] = pts[s * dim + j];
}
// out[r][j][k] = in[r][index[j]][k]: `rows` arrays of P elements of `width` floats each taken in the order `index` (a
// permutation of the P points, or its inverse).  grid (ceil(P / 256), rows): consecutive workgroups share a row, so the
// random 4-byte reads stay inside one 4 MiB row at a time (the XCD's L2) while the writes are coalesced.
__global__ __launch_bounds__(256) void carry_rows(const float *__restrict__ in, const int32_t *__restrict__ index, int64_t P,
                                                  int width, float *__restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= P) return;
    const int64_t row = (int64_t)blockIdx.y * P, s = index[j];
    for (int k = 0; k < width; ++k) out[(row + j) * width + k] = in[(row + s) * width + k];
}
__global__ void zero_word(uint32_t *w) { *w = 0; }
// one counter update per workgroup (with an unordered set every wave has changes: one atomic per wave on the one word
// took 107 us for 2^20 points; per workgroup it is ~5 us)
__global__ __launch_bounds__(256) void tile_changes(const float *__restrict__ pts, int64_t P, KeyDims k,
                                                    uint32_t *__restrict__ count) {
    __shared__ uint32_t wsum[4];
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool change = false;
    if (p < P) change = p == 0 || cell_key(pts + p * k.dim, k, true) != cell_key(pts + (p - 1) * k.dim, k, true);
    const uint64_t m = __ballot(change);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (t) atomicAdd(count, t);
    }
}

// the same count over `segments` runs of 1024 consecutive points, seg_stride points apart (a run's first point has no
// predecessor inside the run and counts as no change): a prefix says nothing about the rest of the set
__global__ __launch_bounds__(256) void tile_changes_sampled(const float *__restrict__ pts, int64_t P, KeyDims k,
                                                            int64_t seg_stride, uint32_t *__restrict__ count) {
    __shared__ uint32_t wsum[4];
    const int64_t p = (int64_t)(blockIdx.x >> 2) * seg_stride + (int64_t)(blockIdx.x & 3) * 256 + threadIdx.x;
    const bool first = (blockIdx.x & 3) == 0 && threadIdx.x == 0;
    bool change = false;
    if (p < P && !first) change = cell_key(pts + p * k.dim, k, true) != cell_key(pts + (p - 1) * k.dim, k, true);
    const uint64_t m = __ballot(change);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (t) atomicAdd(count, t);
    }
}

int key_dims(KeyDims &k, int dim, int64_t D, int64_t H, int64_t W, int pad, int align, int multicell) {
    if ((dim != 2 && dim != 3) || H < 1 || W < 1 || (dim == 3 && D < 1) || pad < 0 || pad > 2) return -1;
    if (W > (1 << 28) || H > (1 << 28) || D > (1 << 28)) return -2;
    k.dim = dim;
    k.size[0] = (int)W;
    k.size[1] = (int)H;
    k.size[2] = dim == 3 ? (int)D : 1;
    double tiles = 1;
    for (int j = 0; j < 3; ++j) {
        k.nt[j] = k.size[j] / TS + 1;
        if (j < dim) tiles *= k.nt[j];
    }
    if (tiles * TS * TS * TS >= 9.0e18) return -2;
    k.f.pad = pad;
    k.f.align = align ? 1 : 0;
    k.f.multicell = multicell ? 1 : 0;
    k.f.exact = 0;
    k.f.pair16 = 0;
    return 0;
}
size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
size_t radix_temp_bytes(int64_t P) {
    size_t temp = 0;
    (void)rocprim::radix_sort_pairs(nullptr, temp, (uint64_t *)nullptr, (uint64_t *)nullptr, (int32_t *)nullptr,
                                    (int32_t *)nullptr, (size_t)P, 0, 64, (hipStream_t)0);
    return temp;
}

}  // namespace

size_t workspace_bytes(int64_t P) {
    if (P <= 0) return 0;
    return 2 * align256((size_t)P * 8) + align256((size_t)P * 4) + align256(radix_temp_bytes(P));
}

int sort_points(int dim, const float *points, int64_t P, int64_t D, int64_t H, int64_t W, int padding_mode,
                int align_corners, int multicell, float *sorted_points, int32_t *perm, void *workspace,
                size_t ws_bytes, hipStream_t stream) {
    KeyDims k;
    int rc = key_dims(k, dim, D, H, W, padding_mode, align_corners, multicell);
    if (rc) return rc;
    if (P < 0 || P > INT32_MAX) return -2;
    if (P == 0) return 0;
    if (!points || !sorted_points || !perm) return -1;
    if (!workspace || ws_bytes < workspace_bytes(P)) return -3;
    char *b = (char *)workspace;
    uint64_t *keys_in = (uint64_t *)b;
    b += align256((size_t)P * 8);
    uint64_t *keys_out = (uint64_t *)b;
    b += align256((size_t)P * 8);
    int32_t *idx = (int32_t *)b;
    b += align256((size_t)P * 4);
    size_t temp = radix_temp_bytes(P);
    const unsigned nb = (unsigned)((P + 255) / 256);
    make_keys<<<nb, 256, 0, stream>>>(points, P, k, keys_in, idx);
    hipError_t e = rocprim::radix_sort_pairs((void *)b, temp, keys_in, keys_out, idx, perm, (size_t)P, 0, 64, stream);
    if (e != hipSuccess) return (int)e;
    gather_points<<<nb, 256, 0, stream>>>(points, perm, P, dim, sorted_points);
    e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int carry_points(const float *in, float *out, const int32_t *index, int64_t rows, int64_t P, int width, hipStream_t stream) {
    if (rows < 0 || P < 0 || width < 1 || width > 4 || rows > 65535 || (P + 255) / 256 > (int64_t)INT32_MAX) return -1;
    if (rows == 0 || P == 0) return 0;
    if (!in || !out || !index) return -1;
    carry_rows<<<dim3((unsigned)((P + 255) / 256), (unsigned)rows), 256, 0, stream>>>(in, index, P, width, out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int count_tile_changes(int dim, const float *points, int64_t P, int64_t D, int64_t H, int64_t W, int padding_mode,
                       int align_corners, int multicell, uint32_t *count, hipStream_t stream) {
    KeyDims k;
    int rc = key_dims(k, dim, D, H, W, padding_mode, align_corners, multicell);
    if (rc) return rc;
    if (P < 0) return -1;
    if (!count || (P > 0 && !points)) return -1;
    zero_word<<<1, 1, 0, stream>>>(count);
    if (P > 0) tile_changes<<<(unsigned)((P + 255) / 256), 256, 0, stream>>>(points, P, k, count);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int sample_tile_changes(int dim, const float *points, int64_t P, int64_t D, int64_t H, int64_t W, int padding_mode,
                        int align_corners, int multicell, int segments, uint32_t *count, hipStream_t stream) {
    if (segments < 1 || segments > 65536) return -1;
    if (P <= (int64_t)segments * 1024)      // the whole set is no more than the sample would be
        return count_tile_changes(dim, points, P, D, H, W, padding_mode, align_corners, multicell, count, stream);
    KeyDims k;
    int rc = key_dims(k, dim, D, H, W, padding_mode, align_corners, multicell);
    if (rc) return rc;
    if (!count || !points) return -1;
    zero_word<<<1, 1, 0, stream>>>(count);
    tile_changes_sampled<<<(unsigned)(4 * segments), 256, 0, stream>>>(points, P, k, P / segments, count);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

}  // namespace sort
}  // namespace cs
