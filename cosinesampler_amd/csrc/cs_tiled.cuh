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
//   plan   (once per grid)  bin samples by (n, 16x16-cell tile): tile-sorted slot `rank1[s]`,
//                           and inside each tile bucket the cell-sorted visiting order `ord[]`.
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
    uint32_t *rank1;       // [S]  sample -> tile-sorted slot (INVALID: touches no node)
    uint8_t *cell1;        // [S]  tile-sorted slot -> local cell id
    uint32_t *ord;         // [S]  cell-sorted position (bucket-relative) -> bucket-relative slot
    uint8_t *ocell;        // [S]  cell-sorted position -> local cell id
    uint32_t *tile_begin;  // [N*ntiles + 1]
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

// (chunks, N): give every sample its tile-sorted slot
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
            uint32_t r = INVALID;
            if (q.valid) {
                r = atomicAdd(&cursor[q.tile], 1u);
                pl.cell1[r] = (uint8_t)q.cell;
            }
            pl.rank1[s] = r;
        }
    }
}

// one workgroup per (n, tile): counting sort of the bucket by local cell id
__global__ __launch_bounds__(256) void plan_tile_sort(Plan pl) {
    __shared__ uint32_t cnt[CELLS];
    __shared__ uint32_t scan[CELLS];
    const int64_t t = blockIdx.x;
    const uint32_t b0 = pl.tile_begin[t], b1 = pl.tile_begin[t + 1];
    if (b0 == b1) return;
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
    cnt[threadIdx.x] = scan[threadIdx.x] - v;  // exclusive start of each cell = running cursor
    __syncthreads();
    for (uint32_t j = b0 + threadIdx.x; j < b1; j += 256) {
        uint8_t c = pl.cell1[j];
        uint32_t pos = atomicAdd(&cnt[c], 1u);
        pl.ord[b0 + pos] = j - b0;
        pl.ocell[b0 + pos] = c;
    }
}

// ------------------------------------------------------------------------------------------------
// point kernels: one lane per sample, channels-last gathers, payload rows to tile-sorted slots
// ------------------------------------------------------------------------------------------------
struct Sample2 {
    int n;
    int64_t p, s;
    Axis ax[2];
    int64_t node[4];   // node index (y*W + x) inside one n, or -1
    float W[4];

    template <int KERNEL, int ORDER>
    __device__ __forceinline__ bool load(const float *grid, const float *offset, const Dims &d, const Flags &f,
                                         int align) {
        s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        bool live = s < d.S;
        int64_t sc = live ? s : d.S - 1;   // keep every lane of the wave busy: cooperative row writes follow
        n = (int)(sc / d.P);
        p = sc - (int64_t)n * d.P;
        float off = offset[n];
        float2 g = *reinterpret_cast<const float2 *>(grid + sc * 2);
        ax[0] = make_axis<KERNEL, ORDER>(g.x, d.size[0], f, align, off);
        ax[1] = make_axis<KERNEL, ORDER>(g.y, d.size[1], f, align, off);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            int x = ax[0].lo + (a & 1), y = ax[1].lo + (a >> 1);
            bool ok = x >= 0 && x < d.size[0] && y >= 0 && y < d.size[1];
            node[a] = ok ? (int64_t)y * d.size[0] + x : -1;
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

__device__ __forceinline__ float4 ld4(const float *base, int64_t node, int C, int q) {
    return node >= 0 ? *reinterpret_cast<const float4 *>(base + node * C + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ float4 fma4(float s, float4 a, float4 acc) {
    return make_float4(fmaf(s, a.x, acc.x), fmaf(s, a.y, acc.y), fmaf(s, a.z, acc.z), fmaf(s, a.w, acc.w));
}

// Cooperative write of one wave's 64 payload rows (C floats each) staged in LDS as stage[row][C]:
// consecutive lanes write consecutive 16-byte pieces, so each row goes out as whole 64-byte sectors.
__device__ __forceinline__ void write_rows(const float *stage, const uint32_t *slots, float *rows, int C) {
    const int CQ = C >> 2;
    const int lane = threadIdx.x & 63;
    for (int i = 0; i < CQ; ++i) {
        int item = i * 64 + lane;
        int r = item / CQ, k = item - r * CQ;
        uint32_t slot = slots[r];
        if (slot != INVALID) {
            float4 v = *reinterpret_cast<const float4 *>(stage + r * C + 4 * k);
            *reinterpret_cast<float4 *>(rows + (int64_t)slot * C + 4 * k) = v;
        }
    }
}

template <int KERNEL>
__global__ __launch_bounds__(256) void point_forward(const float *__restrict__ icl, const float *__restrict__ grid,
                                                     const float *__restrict__ offset, float *__restrict__ out,
                                                     Dims d, Flags f) {
    Sample2 sm;
    if (!sm.load<KERNEL, 0>(grid, offset, d, f, 1)) return;   // 2D forward: align_corners = 1 (2d.cu:307-308)
    const int C = d.C, CQ = C >> 2;
    const float *base = icl + (int64_t)sm.n * d.vol * C;
    float *o = out + (int64_t)sm.n * C * d.P + sm.p;
    for (int q = 0; q < CQ; ++q) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int a = 0; a < 4; ++a)
            if (sm.node[a] >= 0) acc = fma4(sm.W[a], ld4(base, sm.node[a], C, q), acc);
        o[0] = acc.x;
        o[d.P] = acc.y;
        o[2 * d.P] = acc.z;
        o[3 * d.P] = acc.w;
        o += 4 * d.P;
    }
}

// LDS of the point kernels: per wave  slots[64] + NROWS * stage[64][C]
template <int KERNEL>
__global__ __launch_bounds__(256) void point_backward(const float *__restrict__ gOut, const float *__restrict__ icl,
                                                      const float *__restrict__ grid, const float *__restrict__ offset,
                                                      const uint32_t *__restrict__ rank1, float *__restrict__ rows,
                                                      float4 *__restrict__ coef, float *__restrict__ grad_grid,
                                                      Dims d, Flags f) {
    extern __shared__ float lds[];
    const int C = d.C, CQ = C >> 2, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *stage = lds + wave * (64 + 64 * C);
    uint32_t *slots = reinterpret_cast<uint32_t *>(stage);
    stage += 64;
    Sample2 sm;
    bool live = sm.load<KERNEL, 1>(grid, offset, d, f, f.align);
    uint32_t slot = (live && rows) ? rank1[sm.s] : INVALID;
    slots[lane] = slot;
    const float *base = icl + (int64_t)sm.n * d.vol * C;
    const float *go = gOut + (int64_t)sm.n * C * d.P + sm.p;
    float ox0 = -sm.ax[1].w[0], ox1 = sm.ax[1].w[0], ox2 = -sm.ax[1].w[1], ox3 = sm.ax[1].w[1];   // d/dx: -+ on x
    float oy0 = -sm.ax[0].w[0], oy1 = -sm.ax[0].w[1], oy2 = sm.ax[0].w[0], oy3 = sm.ax[0].w[1];   // d/dy: -+ on y
    float gx = 0.f, gy = 0.f;
    for (int q = 0; q < CQ; ++q) {
        float4 g = make_float4(go[0], go[d.P], go[2 * d.P], go[3 * d.P]);
        go += 4 * d.P;
        float4 v0 = ld4(base, sm.node[0], C, q), v1 = ld4(base, sm.node[1], C, q);
        float4 v2 = ld4(base, sm.node[2], C, q), v3 = ld4(base, sm.node[3], C, q);
        gx += ox0 * dot4(v0, g) + ox1 * dot4(v1, g) + ox2 * dot4(v2, g) + ox3 * dot4(v3, g);
        gy += oy0 * dot4(v0, g) + oy1 * dot4(v1, g) + oy2 * dot4(v2, g) + oy3 * dot4(v3, g);
        *reinterpret_cast<float4 *>(stage + lane * C + 4 * q) = g;
    }
    if (live) {
        *reinterpret_cast<float2 *>(grad_grid + sm.s * 2) = make_float2(sm.ax[0].d1 * gx, sm.ax[1].d1 * gy);
        if (slot != INVALID) coef[slot] = make_float4(sm.W[0], sm.W[1], sm.W[2], sm.W[3]);
    }
    if (rows) {
        __builtin_amdgcn_wave_barrier();
        __syncthreads();
        write_rows(stage, slots, rows, C);
    }
}

// second backward, point part.  cIcl = channels-last copy of gOutInput (nullable).
template <int KERNEL>
__global__ __launch_bounds__(256) void point_bb(const float *__restrict__ cIcl, const float *__restrict__ cG,
                                                const float *__restrict__ icl, const float *__restrict__ grid,
                                                const float *__restrict__ gOut, const float *__restrict__ offset,
                                                const uint32_t *__restrict__ rank1, float *__restrict__ rows,
                                                float4 *__restrict__ coef, float *__restrict__ gGrid,
                                                float *__restrict__ ggOut, Dims d, Flags f) {
    extern __shared__ float lds[];
    const int C = d.C, CQ = C >> 2, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *stage = lds + wave * (64 + 64 * C);
    uint32_t *slots = reinterpret_cast<uint32_t *>(stage);
    stage += 64;
    Sample2 sm;
    bool live = sm.load<KERNEL, 2>(grid, offset, d, f, f.align);
    uint32_t slot = live ? rank1[sm.s] : INVALID;
    slots[lane] = slot;
    float2 cg = cG ? *reinterpret_cast<const float2 *>(cG + (live ? sm.s : 0) * 2) : make_float2(0.f, 0.f);
    float Dm[4], Sx[4], Sy[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        Dm[a] = sm.first(a, 0) * cg.x + sm.first(a, 1) * cg.y;
        Sx[a] = sm.pure2(a, 0) * cg.x;   // 2D: pure second derivatives only (2d.cu:705-706)
        Sy[a] = sm.pure2(a, 1) * cg.y;
    }
    const float *base = icl + (int64_t)sm.n * d.vol * C;
    const float *cbase = cIcl ? cIcl + (int64_t)sm.n * d.vol * C : nullptr;
    const float *go = gOut + (int64_t)sm.n * C * d.P + sm.p;
    float *ggo = ggOut + (int64_t)sm.n * C * d.P + sm.p;
    float sx = 0.f, sy = 0.f;
    for (int q = 0; q < CQ; ++q) {
        float4 g = make_float4(go[0], go[d.P], go[2 * d.P], go[3 * d.P]);
        go += 4 * d.P;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f), tx = o, ty = o;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float4 v = ld4(base, sm.node[a], C, q);
            o = fma4(Dm[a], v, o);
            tx = fma4(Sx[a], v, tx);
            ty = fma4(Sy[a], v, ty);
            if (cbase) o = fma4(sm.W[a], ld4(cbase, sm.node[a], C, q), o);
        }
        sx += dot4(tx, g);
        sy += dot4(ty, g);
        if (live) {
            ggo[0] = o.x;
            ggo[d.P] = o.y;
            ggo[2 * d.P] = o.z;
            ggo[3 * d.P] = o.w;
        }
        ggo += 4 * d.P;
        *reinterpret_cast<float4 *>(stage + lane * C + 4 * q) = g;
    }
    if (live) {
        *reinterpret_cast<float2 *>(gGrid + sm.s * 2) = make_float2(sx, sy);
        if (slot != INVALID) coef[slot] = make_float4(Dm[0], Dm[1], Dm[2], Dm[3]);
    }
    __builtin_amdgcn_wave_barrier();
    __syncthreads();
    write_rows(stage, slots, rows, C);
}

// fused third backward, point part: rows1/coef1 carry (gOut, E), rows2/coef2 carry (hO, D)
template <int KERNEL>
__global__ __launch_bounds__(256) void point_bbb(const float *__restrict__ icl, const float *__restrict__ grid,
                                                 const float *__restrict__ gOut, const float *__restrict__ cG,
                                                 const float *__restrict__ hG, const float *__restrict__ hO,
                                                 const float *__restrict__ offset, const uint32_t *__restrict__ rank1,
                                                 float *__restrict__ rows1, float4 *__restrict__ coef1,
                                                 float *__restrict__ rows2, float4 *__restrict__ coef2,
                                                 float *__restrict__ ggOut, Dims d, Flags f) {
    extern __shared__ float lds[];
    const int C = d.C, CQ = C >> 2, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *stage1 = lds + wave * (64 + 128 * C);
    uint32_t *slots = reinterpret_cast<uint32_t *>(stage1);
    stage1 += 64;
    float *stage2 = stage1 + 64 * C;
    Sample2 sm;
    bool live = sm.load<KERNEL, 2>(grid, offset, d, f, f.align);
    uint32_t slot = live ? rank1[sm.s] : INVALID;
    slots[lane] = slot;
    int64_t so = (live ? sm.s : 0) * 2;
    float2 cg = cG ? *reinterpret_cast<const float2 *>(cG + so) : make_float2(0.f, 0.f);
    float2 hg = hG ? *reinterpret_cast<const float2 *>(hG + so) : make_float2(0.f, 0.f);
    float Dm[4], Em[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        Dm[a] = sm.first(a, 0) * cg.x + sm.first(a, 1) * cg.y;
        Em[a] = sm.pure2(a, 0) * (hg.x * cg.x) + sm.pure2(a, 1) * (hg.y * cg.y);   // 2d.cu:876
    }
    const float *base = icl + (int64_t)sm.n * d.vol * C;
    const float *go = gOut + (int64_t)sm.n * C * d.P + sm.p;
    const float *ho = hO ? hO + (int64_t)sm.n * C * d.P + sm.p : nullptr;
    float *ggo = ggOut + (int64_t)sm.n * C * d.P + sm.p;
    for (int q = 0; q < CQ; ++q) {
        float4 g = make_float4(go[0], go[d.P], go[2 * d.P], go[3 * d.P]);
        go += 4 * d.P;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int a = 0; a < 4; ++a) o = fma4(Em[a], ld4(base, sm.node[a], C, q), o);
        if (live) {
            ggo[0] = o.x;
            ggo[d.P] = o.y;
            ggo[2 * d.P] = o.z;
            ggo[3 * d.P] = o.w;
        }
        ggo += 4 * d.P;
        *reinterpret_cast<float4 *>(stage1 + lane * C + 4 * q) = g;
        if (ho) {
            float4 h = make_float4(ho[0], ho[d.P], ho[2 * d.P], ho[3 * d.P]);
            ho += 4 * d.P;
            *reinterpret_cast<float4 *>(stage2 + lane * C + 4 * q) = h;
        }
    }
    if (live && slot != INVALID) {
        coef1[slot] = make_float4(Em[0], Em[1], Em[2], Em[3]);
        if (hO) coef2[slot] = make_float4(Dm[0], Dm[1], Dm[2], Dm[3]);
    }
    __builtin_amdgcn_wave_barrier();
    __syncthreads();
    write_rows(stage1, slots, rows1, C);
    if (hO) write_rows(stage2, slots, rows2, C);
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
    __shared__ float top[TY * (TX + 1) * NS];  // node sums from the cell row above the node row's upper side
    __shared__ float bot[TY * (TX + 1) * NS];
    __shared__ uint32_t rowb[TY + 1];          // bucket-relative start of each cell row (cell-sorted order)

    const int64_t t = blockIdx.x;
    const uint32_t b0 = pl.tile_begin[t], b1 = pl.tile_begin[t + 1];
    if (b0 == b1) return;                      // empty tile: grad_input was zero-filled
    const int n = (int)(t / pl.ntiles), tl = (int)(t - (int64_t)n * pl.ntiles);
    const int ty = tl / pl.ntx, tx = tl - ty * pl.ntx;
    const uint32_t cnt = b1 - b0;

    // start of each cell row: first cell-sorted position whose cell id >= ly*TX (binary search, ocell is sorted)
    if (threadIdx.x <= TY) {
        uint32_t key = threadIdx.x * TX, lo = 0, hi = cnt;
        while (lo < hi) {
            uint32_t mid = (lo + hi) >> 1;
            if (pl.ocell[b0 + mid] < key) lo = mid + 1; else hi = mid;
        }
        rowb[threadIdx.x] = lo;
    }
    __syncthreads();

    const int w = threadIdx.x >> LOGC, c = threadIdx.x & (C - 1);
    for (int ly = w; ly < TY; ly += NW) {
        float ct = 0.f, cb = 0.f;              // sums carried to the next cell: its left nodes are our right nodes
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int cur = 0;                           // current local x
        float *trow = top + ly * (TX + 1) * NS + c;
        float *brow = bot + ly * (TX + 1) * NS + c;
        const uint32_t jb = rowb[ly], je = rowb[ly + 1];
        for (uint32_t j = jb; j < je; ++j) {
            const int x = pl.ocell[b0 + j] & (TX - 1);
            const uint32_t r = b0 + pl.ord[b0 + j];
            while (cur < x) {                  // close cells cur .. x-1 (wave-uniform inside one walker)
                trow[cur * NS] = ct + a0;
                brow[cur * NS] = cb + a2;
                ct = a1; cb = a3;
                a0 = a1 = a2 = a3 = 0.f;
                ++cur;
            }
            float4 k = coef1[r];
            float g = rows1[(int64_t)r * C + c];
            a0 = fmaf(k.x, g, a0); a1 = fmaf(k.y, g, a1); a2 = fmaf(k.z, g, a2); a3 = fmaf(k.w, g, a3);
            if (TWO) {
                float4 k2 = coef2[r];
                float h = rows2[(int64_t)r * C + c];
                a0 = fmaf(k2.x, h, a0); a1 = fmaf(k2.y, h, a1); a2 = fmaf(k2.z, h, a2); a3 = fmaf(k2.w, h, a3);
            }
        }
        while (cur < TX) {
            trow[cur * NS] = ct + a0;
            brow[cur * NS] = cb + a2;
            ct = a1; cb = a3;
            a0 = a1 = a2 = a3 = 0.f;
            ++cur;
        }
        trow[TX * NS] = ct;
        brow[TX * NS] = cb;
    }
    __syncthreads();

    // node (ly, lx) of the tile = global node (ty*TY + ly - 1, tx*TX + lx - 1); its sum is
    // top[ly][lx] (cells below-right in index space) + bot[ly-1][lx]
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
