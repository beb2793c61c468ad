// cs_tiled.cuh -- the fast 2D path for MI355X: no scattered atomics in any hot loop.
//
// Why (measured on MI355X, tools/microbench.hip, profiles/round1_microbench.txt):
//   * global fp32 atomics retire ~20 G requests/s chip-wide whether a request is one float or a
//     64-byte row: the reference's 4*C atomics per sample (2d.cu:469-472, :709, :885) cost
//     51 ms per stage at N=16 C=16 P=2^20 and 3.4 ms even with channels-last rows;
//   * LDS float atomics run at ~0.2 T lane-ops/s chip-wide: accumulating in LDS is no way out;
//   * plain 64-byte row stores to random slots run at ~3 TB/s, sequential row reads at ~6 TB/s;
//   * random 4 x 64 B node gathers from a channels-last table run at ~12.5 TB/s (L2->L1 bound).
//
// Structure of one backward stage (grad_input part):
//   plan   (once per grid)  sort samples by (n, 16x16-cell tile, cell): final slot `rank[s]`, first
//                           slot of every tile (`tile_begin`) and of every cell in it (`cell_begin`).
//   point kernel (p-order)  one lane per sample, streams coalesced, node vectors gathered from
//                           the channels-last copy of `input`; computes every p-ordered output
//                           (grad_grid / ggOut / ...) and writes, per sample, a 64-byte payload
//                           row (the C cotangent values) plus a 16-byte coefficient record
//                           (4 node weights) into its tile-sorted slot.
//   tile kernel ("walkers") one workgroup per (n, tile); 16 lanes = the C channels of one
//                           walker, one walker per cell row of the tile.  A walker visits its
//                           samples in cell order, keeps the 4 node sums of the current cell in
//                           registers, hands the right-hand pair to the next cell (shared nodes)
//                           and stores finished node sums to LDS without atomics; the tile's
//                           (TX+1)x(TY+1) nodes are then added to grad_input (NCHW), lanes along x.
//
// Reference maths per stage: see cs_kernels_direct.cuh (same formulas, same quirks).
#pragma once
#include "cs_kernels_direct.cuh"

namespace cs {
namespace tiled {

constexpr int TX = 16, TY = 16;            // cells per tile
constexpr int CELLS = TX * TY;             // 256 -> local cell id fits a byte
constexpr int CHUNK = 4096;                // samples per plan workgroup
constexpr uint32_t INVALID = 0xFFFFFFFFu;

struct Plan {
    uint32_t *rank;        // [S]  sample -> final slot: sorted by (n, tile, cell)  (INVALID: touches no node)
    uint32_t *sid;         // [S]  scratch: tile-sorted slot -> sample
    uint8_t *cell1;        // [S]  scratch: tile-sorted slot -> local cell id
    uint32_t *tile_begin;  // [N*ntiles + 1]      first slot of every tile bucket
    uint32_t *cell_begin;  // [N*ntiles*(CELLS+1)] bucket-relative first slot of every cell of every tile
    uint32_t *block_hist;  // [N*chunks*ntiles] scratch
    int ntx, nty, ntiles, chunks;
};

struct Geo2 {  // tile coordinates of a sample; u = lo + 1 so that lo = -1 (only the high node valid) is cell 0
    int tile, cell;
    bool valid;
};

__device__ __forceinline__ Geo2 locate(float gx, float gy, const Dims &d, const Flags &f, float off, int ntx) {
    float mu;
    float ix = source_index(gx, d.size[0], f.pad, f.align, off, f.multicell, mu);
    float iy = source_index(gy, d.size[1], f.pad, f.align, off, f.multicell, mu);
    bool sx = (ix > -1073741824.0f) && (ix < 1073741824.0f), sy = (iy > -1073741824.0f) && (iy < 1073741824.0f);
    int ux = sx ? (int)floorf(ix) + 1 : -4, uy = sy ? (int)floorf(iy) + 1 : -4;
    Geo2 g;
    g.valid = ux >= 0 && ux <= d.size[0] && uy >= 0 && uy <= d.size[1];   // at least one node in range
    int tx = ux / TX, ty = uy / TY;
    g.tile = ty * ntx + tx;
    g.cell = (uy - ty * TY) * TX + (ux - tx * TX);
    return g;
}

// ------------------------------------------------------------------------------------------------
// channels-last repack:  in (N,C,vol) -> out (N,vol,C), C % 4 == 0
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_channels_last(const float *__restrict__ in, float *__restrict__ out,
                                                          int C, int64_t vol) {
    extern __shared__ float tile[];  // [C][65]
    const int n = blockIdx.y;
    const int64_t v0 = (int64_t)blockIdx.x * 64;
    const int CQ = C >> 2;
    for (int idx = threadIdx.x; idx < C * 64; idx += 256) {
        int c = idx >> 6, v = idx & 63;
        float x = 0.0f;
        if (v0 + v < vol) x = in[((int64_t)n * C + c) * vol + v0 + v];
        tile[c * 65 + v] = x;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < CQ * 64; idx += 256) {
        int v = idx / CQ, q = idx - v * CQ;
        if (v0 + v < vol) {
            float4 r = make_float4(tile[(4 * q) * 65 + v], tile[(4 * q + 1) * 65 + v], tile[(4 * q + 2) * 65 + v],
                                   tile[(4 * q + 3) * 65 + v]);
            *reinterpret_cast<float4 *>(out + (((int64_t)n * vol + v0 + v) * C + 4 * q)) = r;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// plan kernels
// ------------------------------------------------------------------------------------------------
// (chunks, N) workgroups: histogram of tile ids of one chunk of one n
__global__ __launch_bounds__(256) void plan_count(const float *__restrict__ grid, const float *__restrict__ offset,
                                                  Plan pl, Dims d, Flags f) {
    extern __shared__ uint32_t hist[];
    const int n = blockIdx.y, chunk = blockIdx.x;
    for (int b = threadIdx.x; b < pl.ntiles; b += 256) hist[b] = 0;
    __syncthreads();
    const float off = offset[n];
    const int64_t p0 = (int64_t)chunk * CHUNK;
    for (int i = threadIdx.x; i < CHUNK; i += 256) {
        int64_t p = p0 + i;
        if (p < d.P) {
            float2 g = *reinterpret_cast<const float2 *>(grid + ((int64_t)n * d.P + p) * 2);
            Geo2 q = locate(g.x, g.y, d, f, off, pl.ntx);
            if (q.valid) atomicAdd(&hist[q.tile], 1u);
        }
    }
    __syncthreads();
    uint32_t *dst = pl.block_hist + ((int64_t)n * pl.chunks + chunk) * pl.ntiles;
    for (int b = threadIdx.x; b < pl.ntiles; b += 256) dst[b] = hist[b];
}

// one thread per (n, tile): exclusive prefix over chunks (in place) and the bucket size
__global__ __launch_bounds__(256) void plan_scan_chunks(Plan pl, int N, uint32_t *__restrict__ totals) {
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)N * pl.ntiles) return;
    int n = (int)(t / pl.ntiles), b = (int)(t - (int64_t)n * pl.ntiles);
    uint32_t run = 0;
    uint32_t *h = pl.block_hist + (int64_t)n * pl.chunks * pl.ntiles + b;
    for (int c = 0; c < pl.chunks; ++c) {
        uint32_t v = h[(int64_t)c * pl.ntiles];
        h[(int64_t)c * pl.ntiles] = run;
        run += v;
    }
    totals[t] = run;
}

// single workgroup: exclusive scan of the bucket sizes -> tile_begin[0..count]
__global__ __launch_bounds__(1024) void plan_scan_tiles(const uint32_t *__restrict__ totals,
                                                        uint32_t *__restrict__ tile_begin, int64_t count) {
    __shared__ uint32_t part[1024];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < count; base += 1024) {
        int64_t i = base + threadIdx.x;
        uint32_t v = i < count ? totals[i] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int s = 1; s < 1024; s <<= 1) {  // Hillis-Steele inclusive scan
            uint32_t a = threadIdx.x >= (unsigned)s ? part[threadIdx.x - s] : 0;
            __syncthreads();
            part[threadIdx.x] += a;
            __syncthreads();
        }
        if (i < count) tile_begin[i] = carry + part[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_begin[count] = carry;
}

// (chunks, N): give every sample a slot inside its tile bucket (any order), remember who sits there
__global__ __launch_bounds__(256) void plan_scatter(const float *__restrict__ grid, const float *__restrict__ offset,
                                                    Plan pl, Dims d, Flags f) {
    extern __shared__ uint32_t cursor[];
    const int n = blockIdx.y, chunk = blockIdx.x;
    const uint32_t *excl = pl.block_hist + ((int64_t)n * pl.chunks + chunk) * pl.ntiles;
    const uint32_t *tb = pl.tile_begin + (int64_t)n * pl.ntiles;
    for (int b = threadIdx.x; b < pl.ntiles; b += 256) cursor[b] = tb[b] + excl[b];
    __syncthreads();
    const float off = offset[n];
    const int64_t p0 = (int64_t)chunk * CHUNK;
    for (int i = threadIdx.x; i < CHUNK; i += 256) {
        int64_t p = p0 + i;
        if (p < d.P) {
            int64_t s = (int64_t)n * d.P + p;
            float2 g = *reinterpret_cast<const float2 *>(grid + s * 2);
            Geo2 q = locate(g.x, g.y, d, f, off, pl.ntx);
            if (q.valid) {
                uint32_t r = atomicAdd(&cursor[q.tile], 1u);
                pl.cell1[r] = (uint8_t)q.cell;
                pl.sid[r] = (uint32_t)s;
            } else {
                pl.rank[s] = INVALID;
            }
        }
    }
}

// one workgroup per (n, tile): counting sort of the bucket by local cell id -> final slots
__global__ __launch_bounds__(256) void plan_tile_sort(Plan pl) {
    __shared__ uint32_t cnt[CELLS];
    __shared__ uint32_t scan[CELLS];
    const int64_t t = blockIdx.x;
    const uint32_t b0 = pl.tile_begin[t], b1 = pl.tile_begin[t + 1];
    uint32_t *cbeg = pl.cell_begin + t * (CELLS + 1);
    cnt[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t j = b0 + threadIdx.x; j < b1; j += 256) atomicAdd(&cnt[pl.cell1[j]], 1u);
    __syncthreads();
    uint32_t v = cnt[threadIdx.x];
    scan[threadIdx.x] = v;
    __syncthreads();
    for (int s = 1; s < CELLS; s <<= 1) {
        uint32_t a = threadIdx.x >= (unsigned)s ? scan[threadIdx.x - s] : 0;
        __syncthreads();
        scan[threadIdx.x] += a;
        __syncthreads();
    }
    uint32_t start = scan[threadIdx.x] - v;   // exclusive start of each cell, also the running cursor
    cnt[threadIdx.x] = start;
    cbeg[threadIdx.x] = start;
    if (threadIdx.x == CELLS - 1) cbeg[CELLS] = b1 - b0;
    __syncthreads();
    for (uint32_t j = b0 + threadIdx.x; j < b1; j += 256) {
        uint32_t pos = atomicAdd(&cnt[pl.cell1[j]], 1u);
        pl.rank[pl.sid[j]] = b0 + pos;
    }
}

// ------------------------------------------------------------------------------------------------
// point kernels: one lane per sample, channels-last gathers, payload rows to tile-sorted slots.
// Launch: grid (ceil(P/256), N), 256 threads; CQ = C/4 is a template parameter so that every
// gather of a sample (4 nodes x CQ float4) is in flight before the first one is consumed.
// ------------------------------------------------------------------------------------------------
// Stream accesses (each element touched once per kernel) are marked nontemporal so that they do
// not displace the feature table from L2 / Infinity Cache.
#ifndef CS_NT_STREAMS
#define CS_NT_STREAMS 1
#endif
__device__ __forceinline__ void st_stream(float *p, float v) {
#if CS_NT_STREAMS
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
__device__ __forceinline__ float ld_stream(const float *p) {
#if CS_NT_STREAMS
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

struct Sample2 {
    int n;
    int64_t p, s;
    Axis ax[2];
    uint32_t node[4];  // node index (y*W + x) inside one n; 0 when zero-padded (see ok[])
    bool ok[4];
    float W[4];

    template <int KERNEL, int ORDER>
    __device__ __forceinline__ bool load(const float *grid, const float *offset, const Dims &d, const Flags &f,
                                         int align) {
        n = blockIdx.y;
        int64_t pp = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        bool live = pp < d.P;
        p = live ? pp : d.P - 1;           // keep every lane of the wave busy: cooperative row writes follow
        s = (int64_t)n * d.P + p;
        float off = offset[n];
        float2 g = *reinterpret_cast<const float2 *>(grid + s * 2);
        ax[0] = make_axis<KERNEL, ORDER>(g.x, d.size[0], f, align, off);
        ax[1] = make_axis<KERNEL, ORDER>(g.y, d.size[1], f, align, off);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            int x = ax[0].lo + (a & 1), y = ax[1].lo + (a >> 1);
            ok[a] = x >= 0 && x < d.size[0] && y >= 0 && y < d.size[1];
            node[a] = ok[a] ? (uint32_t)(y * d.size[0] + x) : 0u;
            W[a] = ax[0].w[a & 1] * ax[1].w[a >> 1];
        }
        return live;
    }
    __device__ __forceinline__ float first(int a, int j) const {
        float sgn = ((a >> j) & 1) ? ax[j].d1 : -ax[j].d1;
        return sgn * ax[1 - j].w[(a >> (1 - j)) & 1];
    }
    __device__ __forceinline__ float pure2(int a, int j) const {
        float sgn = ((a >> j) & 1) ? -ax[j].d2 : ax[j].d2;
        return sgn * ax[1 - j].w[(a >> (1 - j)) & 1];
    }
};

__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ float4 fma4(float s, float4 a, float4 acc) {
    return make_float4(fmaf(s, a.x, acc.x), fmaf(s, a.y, acc.y), fmaf(s, a.z, acc.z), fmaf(s, a.w, acc.w));
}
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// all 4*CQ node vectors of a sample; zero-padded nodes read node 0 and are masked afterwards
template <int CQ>
__device__ __forceinline__ void gather_nodes(const float4 *tab, const Sample2 &sm, float4 (&v)[4][CQ]) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int q = 0; q < CQ; ++q) v[a][q] = tab[sm.node[a] * CQ + q];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int q = 0; q < CQ; ++q)
            if (!sm.ok[a]) v[a][q] = zero4();
}

template <int CQ>
__device__ __forceinline__ void load_stream(const float *src, int64_t P, float4 (&g)[CQ]) {
#pragma unroll
    for (int q = 0; q < CQ; ++q)
        g[q] = make_float4(ld_stream(src + (4 * q) * P), ld_stream(src + (4 * q + 1) * P),
                           ld_stream(src + (4 * q + 2) * P), ld_stream(src + (4 * q + 3) * P));
}
template <int CQ>
__device__ __forceinline__ void store_stream(float *dst, int64_t P, const float4 (&o)[CQ]) {
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
        st_stream(dst + (4 * q) * P, o[q].x);
        st_stream(dst + (4 * q + 1) * P, o[q].y);
        st_stream(dst + (4 * q + 2) * P, o[q].z);
        st_stream(dst + (4 * q + 3) * P, o[q].w);
    }
}

// Cooperative write of one wave's 64 payload rows (C floats each) staged in LDS as stage[row][C]:
// consecutive lanes write consecutive 16-byte pieces, so each row goes out as whole 64-byte sectors.
template <int CQ>
__device__ __forceinline__ void write_rows(const float *stage, const uint32_t *slots, float *rows) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < CQ; ++i) {
        int item = i * 64 + lane;
        int r = item / CQ, k = item % CQ;
        uint32_t slot = slots[r];
        if (slot != INVALID) {
            float4 v = *reinterpret_cast<const float4 *>(stage + r * (4 * CQ) + 4 * k);
            *reinterpret_cast<float4 *>(rows + (int64_t)slot * (4 * CQ) + 4 * k) = v;
        }
    }
}
template <int CQ>
__device__ __forceinline__ void stage_row(float *stage, int lane, const float4 (&g)[CQ]) {
#pragma unroll
    for (int q = 0; q < CQ; ++q) *reinterpret_cast<float4 *>(stage + lane * (4 * CQ) + 4 * q) = g[q];
}

template <int KERNEL, int CQ>
__global__ __launch_bounds__(256) void point_forward(const float *__restrict__ icl, const float *__restrict__ grid,
                                                     const float *__restrict__ offset, float *__restrict__ out,
                                                     Dims d, Flags f) {
    constexpr int C = 4 * CQ;
    Sample2 sm;
    if (!sm.load<KERNEL, 0>(grid, offset, d, f, 1)) return;   // 2D forward: align_corners = 1 (2d.cu:307-308)
    const float4 *tab = reinterpret_cast<const float4 *>(icl + (int64_t)sm.n * d.vol * C);
    float4 v[4][CQ];
    gather_nodes<CQ>(tab, sm, v);
    float4 o[CQ];
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
        float4 acc = zero4();
#pragma unroll
        for (int a = 0; a < 4; ++a) acc = fma4(sm.W[a], v[a][q], acc);
        o[q] = acc;
    }
    store_stream<CQ>(out + (int64_t)sm.n * C * d.P + sm.p, d.P, o);
}

// LDS of the point kernels: per wave  slots[64] + NROWS * stage[64][C]
template <int KERNEL, int CQ>
__global__ __launch_bounds__(256) void point_backward(const float *__restrict__ gOut, const float *__restrict__ icl,
                                                      const float *__restrict__ grid, const float *__restrict__ offset,
                                                      const uint32_t *__restrict__ rank, float *__restrict__ rows,
                                                      float4 *__restrict__ coef, float *__restrict__ grad_grid,
                                                      Dims d, Flags f) {
    extern __shared__ float lds[];
    constexpr int C = 4 * CQ;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *stage = lds + wave * (64 + 64 * C);
    uint32_t *slots = reinterpret_cast<uint32_t *>(stage);
    stage += 64;
    Sample2 sm;
    bool live = sm.load<KERNEL, 1>(grid, offset, d, f, f.align);
    uint32_t slot = (live && rows) ? rank[sm.s] : INVALID;
    slots[lane] = slot;
    const float4 *tab = reinterpret_cast<const float4 *>(icl + (int64_t)sm.n * d.vol * C);
    float4 g[CQ], v[4][CQ];
    load_stream<CQ>(gOut + (int64_t)sm.n * C * d.P + sm.p, d.P, g);
    gather_nodes<CQ>(tab, sm, v);
    // d/dx: -/+ on the x side weighted by the y weights; d/dy likewise
    const float wy0 = sm.ax[1].w[0], wy1 = sm.ax[1].w[1], wx0 = sm.ax[0].w[0], wx1 = sm.ax[0].w[1];
    float gx = 0.f, gy = 0.f;
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
        float d0 = dot4(v[0][q], g[q]), d1 = dot4(v[1][q], g[q]), d2 = dot4(v[2][q], g[q]), d3 = dot4(v[3][q], g[q]);
        gx += wy0 * (d1 - d0) + wy1 * (d3 - d2);
        gy += wx0 * (d2 - d0) + wx1 * (d3 - d1);
    }
    if (live) {
        *reinterpret_cast<float2 *>(grad_grid + sm.s * 2) = make_float2(sm.ax[0].d1 * gx, sm.ax[1].d1 * gy);
        if (slot != INVALID) coef[slot] = make_float4(sm.W[0], sm.W[1], sm.W[2], sm.W[3]);
    }
    if (rows) {
        stage_row<CQ>(stage, lane, g);
        __syncthreads();
        write_rows<CQ>(stage, slots, rows);
    }
}

// second backward, point part.  cIcl = channels-last copy of gOutInput (nullable).
template <int KERNEL, int CQ, bool HAS_CI>
__global__ __launch_bounds__(256) void point_bb(const float *__restrict__ cIcl, const float *__restrict__ cG,
                                                const float *__restrict__ icl, const float *__restrict__ grid,
                                                const float *__restrict__ gOut, const float *__restrict__ offset,
                                                const uint32_t *__restrict__ rank, float *__restrict__ rows,
                                                float4 *__restrict__ coef, float *__restrict__ gGrid,
                                                float *__restrict__ ggOut, Dims d, Flags f) {
    extern __shared__ float lds[];
    constexpr int C = 4 * CQ;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *stage = lds + wave * (64 + 64 * C);
    uint32_t *slots = reinterpret_cast<uint32_t *>(stage);
    stage += 64;
    Sample2 sm;
    bool live = sm.load<KERNEL, 2>(grid, offset, d, f, f.align);
    uint32_t slot = live ? rank[sm.s] : INVALID;
    slots[lane] = slot;
    float2 cg = cG ? *reinterpret_cast<const float2 *>(cG + sm.s * 2) : make_float2(0.f, 0.f);
    const float4 *tab = reinterpret_cast<const float4 *>(icl + (int64_t)sm.n * d.vol * C);
    float4 g[CQ], v[4][CQ];
    load_stream<CQ>(gOut + (int64_t)sm.n * C * d.P + sm.p, d.P, g);
    gather_nodes<CQ>(tab, sm, v);
    float Dm[4], Sx[4], Sy[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        Dm[a] = sm.first(a, 0) * cg.x + sm.first(a, 1) * cg.y;
        Sx[a] = sm.pure2(a, 0) * cg.x;   // 2D: pure second derivatives only (2d.cu:705-706)
        Sy[a] = sm.pure2(a, 1) * cg.y;
    }
    float4 o[CQ];
    float sx = 0.f, sy = 0.f;
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
        float4 acc = zero4(), tx = zero4(), ty = zero4();
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            acc = fma4(Dm[a], v[a][q], acc);
            tx = fma4(Sx[a], v[a][q], tx);
            ty = fma4(Sy[a], v[a][q], ty);
        }
        sx += dot4(tx, g[q]);
        sy += dot4(ty, g[q]);
        o[q] = acc;
    }
    if (HAS_CI) {   // + sum_a gOutInput[q_a] * W_a   (2d.cu:694-697)
        const float4 *ctab = reinterpret_cast<const float4 *>(cIcl + (int64_t)sm.n * d.vol * C);
        float4 u[4][CQ];
        gather_nodes<CQ>(ctab, sm, u);
#pragma unroll
        for (int q = 0; q < CQ; ++q)
#pragma unroll
            for (int a = 0; a < 4; ++a) o[q] = fma4(sm.W[a], u[a][q], o[q]);
    }
    if (live) {
        store_stream<CQ>(ggOut + (int64_t)sm.n * C * d.P + sm.p, d.P, o);
        *reinterpret_cast<float2 *>(gGrid + sm.s * 2) = make_float2(sx, sy);
        if (slot != INVALID) coef[slot] = make_float4(Dm[0], Dm[1], Dm[2], Dm[3]);
    }
    stage_row<CQ>(stage, lane, g);
    __syncthreads();
    write_rows<CQ>(stage, slots, rows);
}

// fused third backward, point part: rows1/coef1 carry (gOut, E), rows2/coef2 carry (hO, D)
template <int KERNEL, int CQ>
__global__ __launch_bounds__(256) void point_bbb(const float *__restrict__ icl, const float *__restrict__ grid,
                                                 const float *__restrict__ gOut, const float *__restrict__ cG,
                                                 const float *__restrict__ hG, const float *__restrict__ hO,
                                                 const float *__restrict__ offset, const uint32_t *__restrict__ rank,
                                                 float *__restrict__ rows1, float4 *__restrict__ coef1,
                                                 float *__restrict__ rows2, float4 *__restrict__ coef2,
                                                 float *__restrict__ ggOut, Dims d, Flags f) {
    extern __shared__ float lds[];
    constexpr int C = 4 * CQ;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *stage1 = lds + wave * (64 + 128 * C);
    uint32_t *slots = reinterpret_cast<uint32_t *>(stage1);
    stage1 += 64;
    float *stage2 = stage1 + 64 * C;
    Sample2 sm;
    bool live = sm.load<KERNEL, 2>(grid, offset, d, f, f.align);
    uint32_t slot = live ? rank[sm.s] : INVALID;
    slots[lane] = slot;
    float2 cg = cG ? *reinterpret_cast<const float2 *>(cG + sm.s * 2) : make_float2(0.f, 0.f);
    float2 hg = hG ? *reinterpret_cast<const float2 *>(hG + sm.s * 2) : make_float2(0.f, 0.f);
    const float4 *tab = reinterpret_cast<const float4 *>(icl + (int64_t)sm.n * d.vol * C);
    float4 g[CQ], v[4][CQ];
    load_stream<CQ>(gOut + (int64_t)sm.n * C * d.P + sm.p, d.P, g);
    gather_nodes<CQ>(tab, sm, v);
    float Dm[4], Em[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        Dm[a] = sm.first(a, 0) * cg.x + sm.first(a, 1) * cg.y;
        Em[a] = sm.pure2(a, 0) * (hg.x * cg.x) + sm.pure2(a, 1) * (hg.y * cg.y);   // 2d.cu:876
    }
    float4 o[CQ];
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
        float4 acc = zero4();
#pragma unroll
        for (int a = 0; a < 4; ++a) acc = fma4(Em[a], v[a][q], acc);
        o[q] = acc;
    }
    if (live) {
        store_stream<CQ>(ggOut + (int64_t)sm.n * C * d.P + sm.p, d.P, o);
        if (slot != INVALID) {
            coef1[slot] = make_float4(Em[0], Em[1], Em[2], Em[3]);
            if (hO) coef2[slot] = make_float4(Dm[0], Dm[1], Dm[2], Dm[3]);
        }
    }
    stage_row<CQ>(stage1, lane, g);
    if (hO) {
        float4 h[CQ];
        load_stream<CQ>(hO + (int64_t)sm.n * C * d.P + sm.p, d.P, h);
        stage_row<CQ>(stage2, lane, h);
    }
    __syncthreads();
    write_rows<CQ>(stage1, slots, rows1);
    if (hO) write_rows<CQ>(stage2, slots, rows2);
}

// ------------------------------------------------------------------------------------------------
// tile kernel: grad_input[n,c,node] += sum over the tile's samples of coef_a * row[c]
// one workgroup per (n, tile); C lanes per walker; walker w owns cell rows w, w+NW, ...
// ------------------------------------------------------------------------------------------------
template <int LOGC, bool TWO>
__global__ __launch_bounds__(256) void tile_scatter(const float *__restrict__ rows1, const float4 *__restrict__ coef1,
                                                    const float *__restrict__ rows2, const float4 *__restrict__ coef2,
                                                    Plan pl, float *__restrict__ grad_input, Dims d) {
    constexpr int C = 1 << LOGC;
    constexpr int NW = 256 >> LOGC;            // walkers per workgroup
    constexpr int NS = C + 1;                  // padded node stride (floats) in LDS
    constexpr int U = 4;                       // samples in flight per walker
    __shared__ float top[TY * (TX + 1) * NS];  // node sums seen from the cell row below-right / above
    __shared__ float bot[TY * (TX + 1) * NS];
    __shared__ uint32_t cb[CELLS + 1];         // bucket-relative first slot of every cell

    const int64_t t = blockIdx.x;
    const uint32_t b0 = pl.tile_begin[t], b1 = pl.tile_begin[t + 1];
    if (b0 == b1) return;                      // empty tile: grad_input was zero-filled
    const int n = (int)(t / pl.ntiles), tl = (int)(t - (int64_t)n * pl.ntiles);
    const int ty = tl / pl.ntx, tx = tl - ty * pl.ntx;
    {
        const uint32_t *cbeg = pl.cell_begin + t * (CELLS + 1);
        cb[threadIdx.x] = cbeg[threadIdx.x];
        if (threadIdx.x == 0) cb[CELLS] = cbeg[CELLS];
    }
    __syncthreads();

    const int w = threadIdx.x >> LOGC, c = threadIdx.x & (C - 1);
    for (int ly = w; ly < TY; ly += NW) {
        float ct = 0.f, cbm = 0.f;             // sums carried to the next cell: its left nodes are our right nodes
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int cur = 0;                           // current local x
        float *trow = top + ly * (TX + 1) * NS + c;
        float *brow = bot + ly * (TX + 1) * NS + c;
        const uint32_t *cbr = cb + ly * TX;
        const uint32_t j1 = cbr[TX];
        uint32_t nb = cbr[1];                  // first slot of the next cell
        for (uint32_t j = cbr[0]; j < j1; j += U) {
            float4 k[U], k2[U];
            float g[U], h[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {      // all loads of the batch first: the slots are consecutive
                uint32_t r = b0 + min(j + u, j1 - 1);
                k[u] = coef1[r];
                g[u] = rows1[(int64_t)r * C + c];
                if (TWO) {
                    k2[u] = coef2[r];
                    h[u] = rows2[(int64_t)r * C + c];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (j + u < j1) {
                    while (j + u >= nb) {      // close cells up to the one holding this slot
                        trow[cur * NS] = ct + a0;
                        brow[cur * NS] = cbm + a2;
                        ct = a1; cbm = a3;
                        a0 = a1 = a2 = a3 = 0.f;
                        ++cur;
                        nb = cbr[cur + 1];
                    }
                    a0 = fmaf(k[u].x, g[u], a0); a1 = fmaf(k[u].y, g[u], a1);
                    a2 = fmaf(k[u].z, g[u], a2); a3 = fmaf(k[u].w, g[u], a3);
                    if (TWO) {
                        a0 = fmaf(k2[u].x, h[u], a0); a1 = fmaf(k2[u].y, h[u], a1);
                        a2 = fmaf(k2[u].z, h[u], a2); a3 = fmaf(k2[u].w, h[u], a3);
                    }
                }
            }
        }
        while (cur < TX) {
            trow[cur * NS] = ct + a0;
            brow[cur * NS] = cbm + a2;
            ct = a1; cbm = a3;
            a0 = a1 = a2 = a3 = 0.f;
            ++cur;
        }
        trow[TX * NS] = ct;
        brow[TX * NS] = cbm;
    }
    __syncthreads();

    // node (ly, lx) of the tile = global node (ty*TY + ly - 1, tx*TX + lx - 1); its sum is
    // top[ly][lx] (cell row ly, low-y nodes) + bot[ly-1][lx] (cell row ly-1, high-y nodes)
    const int W = d.size[0], H = d.size[1];
    float *gi = grad_input + (int64_t)n * C * d.vol;
    for (int idx = threadIdx.x; idx < C * (TY + 1) * (TX + 1); idx += 256) {
        int lx = idx % (TX + 1);
        int rest = idx / (TX + 1);
        int ly = rest % (TY + 1);
        int ch = rest / (TY + 1);
        int gx = tx * TX + lx - 1, gy = ty * TY + ly - 1;
        if (gx < 0 || gx >= W || gy < 0 || gy >= H) continue;
        float v = 0.f;
        if (ly < TY) v += top[(ly * (TX + 1) + lx) * NS + ch];
        if (ly > 0) v += bot[((ly - 1) * (TX + 1) + lx) * NS + ch];
        if (v != 0.f) unsafeAtomicAdd(gi + (int64_t)ch * d.vol + (int64_t)gy * W + gx, v);
    }
}

}  // namespace tiled
}  // namespace cs
