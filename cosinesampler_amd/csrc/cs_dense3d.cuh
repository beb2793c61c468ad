// cs_dense3d.cuh -- 3D with small, crowded tables (the reference's test_3d.py: 50 tables of 16^3 cells, C = 4,
// 10^5-10^6 points).  A whole table is a few hundred KiB, so the row atomics of the general 3D path all land on
// the same few thousand L2 lines and serialise (1.3 ms for 5 M samples; 0.15 ms for the gathers next to it).
// Same cure as cs_tiled.cuh's cell_scatter for 2D: the plan bins the samples by CELL (counting sort, the
// scan kernels are cs_tiled.cuh's), the channels-last point kernels leave p-ordered rows
// [cotangent values | node coefficients] instead of adding them, and one WAVE owns one (n, cell) bucket: lanes
// stride through the bucket keeping 8*C sums, a halving exchange leaves every lane with one (node, channel)
// value, which it adds to grad_input -- in the caller's layout, so the channels-last accumulator, its clear and
// its unpack disappear as well.
#pragma once
#include "cs_points_cl.cuh"
#include "cs_tiled.cuh"

namespace cs {
namespace dense3 {

using tiled::Plan;

struct Cell3 {
    int bin;
    bool valid;
};
// cell coordinates u = lo + 1 in [0, size] per axis (lo = -1: only the high node is in range)
__device__ __forceinline__ Cell3 locate3(const float *g, const Dims &d, const Flags &f, float off, const Plan &pl) {
    int u[3];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float mu;
        float i = source_index(g[j], d.size[j], f.pad, f.align, off, f.multicell, mu);
        bool sane = (i > -1073741824.0f) && (i < 1073741824.0f);
        u[j] = sane ? (int)floorf(i) + 1 : -4;
        ok = ok && u[j] >= 0 && u[j] <= d.size[j];
    }
    Cell3 c;
    c.valid = ok;
    c.bin = (u[2] * pl.nty + u[1]) * pl.ntx + u[0];
    return c;
}

// (chunks, N) workgroups: histogram of the cells of one chunk of one n   (Plan: ntx = W+1, nty = H+1, ntiles = cells)
__global__ __launch_bounds__(256) void plan_count3(const float *__restrict__ grid, const float *__restrict__ offset,
                                                   Plan pl, Dims d, Flags f) {
    extern __shared__ uint32_t hist[];
    const int n = blockIdx.y, chunk = blockIdx.x;
    for (int b = threadIdx.x; b < pl.ntiles; b += 256) hist[b] = 0;
    __syncthreads();
    const float off = offset[n];
    const int64_t p0 = (int64_t)chunk * pl.chunk;
    for (int i = threadIdx.x; i < pl.chunk; i += 256) {
        int64_t p = p0 + i;
        if (p < d.P) {
            Cell3 q = locate3(grid + d.gpt(n, p) * 3, d, f, off, pl);
            if (q.valid) atomicAdd(&hist[q.bin], 1u);
        }
    }
    __syncthreads();
    uint32_t *dst = pl.block_hist + ((int64_t)n * pl.chunks + chunk) * pl.ntiles;
    for (int b = threadIdx.x; b < pl.ntiles; b += 256) dst[b] = hist[b];
}

// (chunks, N): every sample takes a slot of its cell's bucket -- the final order
__global__ __launch_bounds__(256) void plan_scatter3(const float *__restrict__ grid, const float *__restrict__ offset,
                                                     Plan pl, Dims d, Flags f) {
    extern __shared__ uint32_t cursor[];
    const int n = blockIdx.y, chunk = blockIdx.x;
    const uint32_t *excl = pl.block_hist + ((int64_t)n * pl.chunks + chunk) * pl.ntiles;
    const uint32_t *tb = pl.tile_begin + (int64_t)n * pl.ntiles;
    for (int b = threadIdx.x; b < pl.ntiles; b += 256) cursor[b] = tb[b] + excl[b];
    __syncthreads();
    const float off = offset[n];
    const int64_t p0 = (int64_t)chunk * pl.chunk;
    for (int i = threadIdx.x; i < pl.chunk; i += 256) {
        int64_t p = p0 + i;
        if (p < d.P) {
            int64_t s = (int64_t)n * d.P + p;
            Cell3 q = locate3(grid + d.gpt(n, p) * 3, d, f, off, pl);
            if (q.valid) pl.sorted[atomicAdd(&cursor[q.bin], 1u)] = (uint32_t)s;
        }
    }
}

// one wave per (n, cell) bucket.  Rows: cl::Rec<3, CQ, MODE> without the ids = [g (C) | (h (C)) | coef (8) | (coef2 (8))]
template <int CQ, int MODE>
__global__ __launch_bounds__(256) void cell_scatter3(const float *__restrict__ rows, Plan pl,
                                                     float *__restrict__ grad_input, Dims d) {
    using R = cl::Rec<3, CQ, MODE>;
    constexpr int C = 4 * CQ, NC = 8, NV = NC * C;
    static_assert(NV <= 128, "at most two (node, channel) values per lane");
    const int64_t bucket = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bucket >= (int64_t)d.N * pl.ntiles) return;
    const uint32_t b0 = pl.tile_begin[bucket], b1 = pl.tile_begin[bucket + 1];
    if (b0 == b1) return;
    const int lane = threadIdx.x & 63;
    float v[NV];   // v[a * C + c]
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = 0.f;
    for (uint32_t j = b0 + lane; j < b1; j += 64) {
        const float *row = rows + (int64_t)pl.sorted[j] * R::IDS;
        float g[C], h[C], ka[NC], kb[NC];
#pragma unroll
        for (int q = 0; q < CQ; ++q) {
            const float4 t = *reinterpret_cast<const float4 *>(row + 4 * q);
            g[4 * q] = t.x; g[4 * q + 1] = t.y; g[4 * q + 2] = t.z; g[4 * q + 3] = t.w;
            if (MODE == 2) {
                const float4 u = *reinterpret_cast<const float4 *>(row + C + 4 * q);
                h[4 * q] = u.x; h[4 * q + 1] = u.y; h[4 * q + 2] = u.z; h[4 * q + 3] = u.w;
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float4 t = *reinterpret_cast<const float4 *>(row + R::COEF + 4 * q);
            ka[4 * q] = t.x; ka[4 * q + 1] = t.y; ka[4 * q + 2] = t.z; ka[4 * q + 3] = t.w;
            if (MODE == 2) {
                const float4 u = *reinterpret_cast<const float4 *>(row + R::COEF + NC + 4 * q);
                kb[4 * q] = u.x; kb[4 * q + 1] = u.y; kb[4 * q + 2] = u.z; kb[4 * q + 3] = u.w;
            }
        }
#pragma unroll
        for (int a = 0; a < NC; ++a)
#pragma unroll
            for (int c = 0; c < C; ++c) {
                float t = fmaf(ka[a], g[c], v[a * C + c]);
                if (MODE == 2) t = fmaf(kb[a], h[c], t);
                v[a * C + c] = t;
            }
    }
    // halving exchange (as tiled::cell_scatter): six steps leave NV / 64 fully reduced values per lane (C = 16: values
    // 2 l and 2 l + 1 on lane l); with NV < 64 the last steps are plain butterflies and 64 / NV lanes share a value
    constexpr int VALS = NV >= 64 ? NV / 64 : 1, SHARE = NV >= 64 ? 1 : 64 / NV;
    int cur = NV;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        if (cur > VALS) {
            const int half = cur / 2;
            const bool up = (lane & m) != 0;
#pragma unroll
            for (int i = 0; i < half; ++i) {
                const float lo = v[i], hi = v[i + half];
                const float got = __shfl_xor(up ? lo : hi, m);
                v[i] = (up ? hi : lo) + got;
            }
            cur = half;
        } else {
            v[0] += __shfl_xor(v[0], m);
        }
    }
    if (lane % SHARE) return;
    const int n = (int)(bucket / pl.ntiles);
    int cell = (int)(bucket - (int64_t)n * pl.ntiles);
    const int ux = cell % pl.ntx;
    cell /= pl.ntx;
    const int uy = cell % pl.nty, uz = cell / pl.nty;
#pragma unroll
    for (int jv = 0; jv < VALS; ++jv) {
        const int idx = (lane / SHARE) * VALS + jv, a = idx / C, c = idx % C;
        const int x = ux - 1 + (a & 1), y = uy - 1 + ((a >> 1) & 1), z = uz - 1 + (a >> 2);
        const float r = v[jv];
        if (c >= d.C || x < 0 || x >= d.size[0] || y < 0 || y >= d.size[1] || z < 0 || z >= d.size[2] || r == 0.f) continue;
        unsafeAtomicAdd(grad_input + ((int64_t)n * d.C + c) * d.vol + ((int64_t)z * d.size[1] + y) * d.size[0] + x, r);
    }
}

}  // namespace dense3
}  // namespace cs
