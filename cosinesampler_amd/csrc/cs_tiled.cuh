// cs_tiled.cuh -- the fast 2D path for MI355X: no scattered atomics in any hot loop.
//
// Why (measured on MI355X, tools/microbench.hip, profiles/round1_microbench.txt):
//   * global fp32 atomics retire ~20 G requests/s chip-wide whether a request is one float or a
//     64-byte row: the reference's 4*C atomics per sample (2d.cu:469-472, :709, :885) cost
//     51 ms per stage at N=16 C=16 P=2^20 and 3.4 ms even with channels-last rows;
//   * LDS float atomics run at ~0.2 T lane-ops/s chip-wide: accumulating in LDS is no way out;
//   * plain 64-byte row stores to random slots run at ~3 TB/s, sequential row reads at ~6 TB/s;
//   * random 4 x 64 B node gathers from a channels-last table run at ~200 G rows/s when one 4 MiB table at a
//     time is hot in the L2, a third of that with all N tables hot -- and they do NOT overlap HBM streams
//     (tools/microbench_sum.hip O1-O5): a kernel costs t_gather + bytes / 6.5 TB/s, so gathers are issued with as
//     little register footprint as possible (C/4 lanes per sample) and streams are kept to the algorithmic bytes
//     plus the fat rows;
//   * every read that misses the L2 moves a 128-byte line (round 2 counters: TCC_EA0_RDREQ_128B = all requests): an
//     80-byte row fetched by id costs 1.5 lines on average, a 16-byte record a whole one; random line reads run at
//     6.9 TB/s, random 64-byte writes at 3 TB/s -- so records are WRITTEN in p-order and FETCHED by id, never the reverse.
//
// Structure of one backward stage (grad_input part; round 2: the first scatter stage of a step also leaves grad_output in
// cell order inside the plan -- Plan::Gs -- and the later stages write only their coefficients):
//   plan   (once per grid)  sort the samples by (n, 16x16-cell tile, cell): `sorted[j]` = sample id
//                           at sorted position j, first position of every tile (`tile_begin`) and
//                           of every cell in it (`cell_begin`).
//   point kernel (p-order)  every stream access coalesced (lane = sample), node vectors gathered from the
//                           channels-last copy of `input` (C/4 lanes per sample); computes every
//                           p-ordered output (grad_grid / ggOut / ...) and leaves, per sample, one
//                           contiguous "fat row" in p-order: the C cotangent values followed by the
//                           4 node coefficients (third backward: two of each).  Sequential writes.
//   tile kernel ("walkers") one workgroup per (n, tile); C/4 lanes form a walker (one float4 of
//                           channels each), one walker per run of C/4 cells of one cell row.  A
//                           walker visits its samples in cell order -- fetching their fat rows by
//                           sample id, the only random access of the stage, ~80-160 contiguous
//                           bytes each --, keeps the 4 node sums of the current cell in registers,
//                           hands the right-hand pair to the next cell (shared nodes) and stores
//                           finished node sums to LDS without atomics; the tile's (TX+1)x(TY+1)
//                           nodes are then added to grad_input (NCHW), lanes along x.
//   crowded tables          (>= 128 samples per cell, PIXEL's 16x16 tables): the plan bins by cell directly and
//                           cell_scatter gives every (n, cell) bucket a wave (end of this file).
//
// Reference maths per stage: see cs_kernels_direct.cuh (same formulas, same quirks).
#pragma once
#include "cs_kernels_direct.cuh"

namespace cs {
namespace tiled {

constexpr int TX = 16, TY = 16;            // cells per tile
constexpr int CELLS = TX * TY;             // 256 -> local cell id fits a byte
// floats per fat row: [payload (C) | 4 coefficients], third backward [2 payloads | 2 x 4 coefficients]
// (80-byte rows padded and aligned to 128 bytes measured slower: +0.11-0.16 ms per stage, profiles/round1_ablation.txt)
__host__ __device__ constexpr int row1(int C) { return C + 4; }
__host__ __device__ constexpr int row2(int C) { return 2 * C + 8; }
constexpr int CHUNK = 16384;               // samples per plan workgroup, at most (Plan::chunk)

struct Plan {
    uint32_t *sorted;      // [S]  sorted position (by n, tile, cell) -> sample id n*P+p; only the first
                           //      tile_begin[N*ntiles] entries are defined (samples touching no node are dropped)
    uint32_t *key;         // [S]  scratch: tile-sorted slot -> (p << 8) | local cell id; with more than 2^24 points per
                           //      table (`cellb` set): the point index alone, the cell id in cellb[slot]
    uint8_t *cellb;        // [S]  scratch, only when P > 2^24
    uint32_t *tile_begin;  // [N*ntiles + 1]      first sorted position of every tile bucket
    uint32_t *cell_begin;  // [N*ntiles*(CELLS+1)] bucket-relative first position of every cell of every tile
    uint32_t *block_hist;  // [N*chunks*ntiles] scratch
    int ntx, nty, ntiles, chunks;
    int chunk;             // samples per plan workgroup (CHUNK; larger when the histogram is large)
    int dense;             // crowded tables: the bins ARE the cells -- ntx = W+1, ntiles = (W+1)(H+1), `sorted` is written
                           // by plan_scatter directly, no per-tile pass, no cell_begin; consumed by cell_scatter
    float *Gs;             // [S*CP] (walker plans only) grad_output rows in SORTED order, channels-last: left by the
                           // first tile kernel that fetched them by sample id, streamed by the later stages' walkers
};

struct Geo2 {  // tile coordinates of a sample; u = lo + 1 so that lo = -1 (only the high node valid) is cell 0
    int tile, cell;
    int ux, uy;
    bool valid;
    __device__ __forceinline__ int bin(const Plan &pl) const { return pl.dense ? uy * pl.ntx + ux : tile; }
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
    g.ux = ux;
    g.uy = uy;
    return g;
}

// ------------------------------------------------------------------------------------------------
// channels-last repack:  in (N,C,vol) -> out (N,vol,CP), CP = C rounded up to a multiple of 4 (extra channels zero)
// ------------------------------------------------------------------------------------------------
// (shift, slots: the z-paired 3D layout, see pack_cl4 in cs_kernels_direct.cuh; 0, 1 otherwise)
static __global__ __launch_bounds__(256) void pack_channels_last(const float *__restrict__ in, float *__restrict__ out,
                                                          int C, int CP, int64_t vol, int64_t shift, int slots) {
    extern __shared__ float tile[];  // [CP][65]
    const int n = blockIdx.y, slot = blockIdx.z;
    const int64_t v0 = (int64_t)blockIdx.x * 64, src0 = v0 + slot * shift;
    const int CQ = CP >> 2;
    for (int idx = threadIdx.x; idx < CP * 64; idx += 256) {
        int c = idx >> 6, v = idx & 63;
        float x = 0.0f;
        if (c < C && src0 + v < vol) x = in[((int64_t)n * C + c) * vol + src0 + v];
        tile[c * 65 + v] = x;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < CQ * 64; idx += 256) {
        int v = idx / CQ, q = idx - v * CQ;
        if (v0 + v < vol) {
            float4 r = make_float4(tile[(4 * q) * 65 + v], tile[(4 * q + 1) * 65 + v], tile[(4 * q + 2) * 65 + v],
                                   tile[(4 * q + 3) * 65 + v]);
            *reinterpret_cast<float4 *>(out + ((((int64_t)n * vol + v0 + v) * slots + slot) * CP + 4 * q)) = r;
        }
    }
}

// XCD-aware order of the backward point kernels' workgroups (a speed choice only: every order computes the same values;
// Dims::xcd, set by the host where it was measured to pay).  The launch is (ceil(P/256), N) workgroups, dealt round-robin
// over the 8 XCDs (MI355X_MICROARCH.md, workgroup dispatch: blocks b and b + 8 share one): in launch order the whole chip
// works on table n, then on n + 1.  With Dims::xcd the XCDs form CS_XCD_GROUPS groups, each owning whole tables (n = g,
// g + G, ... one after the other) and starting a G-th of the points further on than the last, so that G tables are in flight
// at different p.  BASELINE configs[1], fp32 streams: first / second backward -0.05 / -0.04 ms, the step -1.8 % -- with 2, 4
// or 8 groups alike, so it is not the table meeting fewer L2s that pays (profiles/round4_ablation.txt section 12, 15).
// Where it LOSES and stays off: the forward point kernel (-0.02 ms: it has nothing but its 16 output rows per table to
// stream), 16-bit streams (+5-7 % on the step) and the crowded-table path (+2-3 %).
#ifndef CS_XCD_TABLES
#define CS_XCD_TABLES 1
#endif
#ifndef CS_XCD_STAGGER
#define CS_XCD_STAGGER 1
#endif
struct PBlk {
    int x, n;      // which 256 points, which table
};
#ifndef CS_XCD_GROUPS
#define CS_XCD_GROUPS 2       // tables in flight: the XCDs form this many groups, each group owns whole tables (2, 4 or 8: measured alike)
#endif
__device__ __forceinline__ PBlk pblk(int on) {      // on: Dims::xcd, the host's choice (0: the launch order)
    PBlk b{(int)blockIdx.x, (int)blockIdx.y};
#if CS_XCD_TABLES
    constexpr uint32_t G = CS_XCD_GROUPS, M = 8 / G;      // M XCDs per group
    if (on && G > 1 && (gridDim.y % G) == 0 && (uint64_t)gridDim.x * gridDim.y < (1ull << 31)) {
        const uint32_t l = blockIdx.y * gridDim.x + blockIdx.x, xcd = l & 7u, g = xcd % G, r = xcd / G;
        const uint32_t k = (l >> 3) * M + r, t = k / gridDim.x;      // the group's k-th workgroup: table t of the group
        b.x = (int)(k - t * gridDim.x);
        b.n = (int)(g + G * t);
#if CS_XCD_STAGGER
        // ... and each of the groups starts its tables a G-th of the points further on: the tables in flight are then
        // read and written at different p (without this the gain depends on the device: section 12)
        b.x = (int)(((uint32_t)b.x + g * ((gridDim.x + G - 1) / G)) % gridDim.x);
#endif
    }
#endif
    return b;
}

// ------------------------------------------------------------------------------------------------
// plan kernels
// ------------------------------------------------------------------------------------------------
// (chunks, N) workgroups: histogram of tile ids of one chunk of one n
static __global__ __launch_bounds__(256) void plan_count(const float *__restrict__ grid, const float *__restrict__ offset,
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
            float2 g = *reinterpret_cast<const float2 *>(grid + d.gpt(n, p) * 2);
            Geo2 q = locate(g.x, g.y, d, f, off, pl.ntx);
            if (q.valid) atomicAdd(&hist[q.bin(pl)], 1u);
        }
    }
    __syncthreads();
    uint32_t *dst = pl.block_hist + ((int64_t)n * pl.chunks + chunk) * pl.ntiles;
    for (int b = threadIdx.x; b < pl.ntiles; b += 256) dst[b] = hist[b];
}

// one thread per (n, tile): exclusive prefix over chunks (in place) and the bucket size
static __global__ __launch_bounds__(256) void plan_scan_chunks(Plan pl, int N, uint32_t *__restrict__ totals) {
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

// exclusive scan of the bucket sizes -> tile_begin[0..count], in three small launches: (1) every workgroup scans its
// 1024 entries in LDS and leaves its total in bsum, (2) one workgroup scans bsum, (3) the workgroup prefixes are added.
// (A single workgroup sweeping the array took 0.45 ms for the 2.5e5 cell buckets of the reference's 3D test shapes.)
__device__ __forceinline__ uint32_t block_scan_1024(uint32_t v, uint32_t *part) {   // inclusive, all 1024 threads
    part[threadIdx.x] = v;
    __syncthreads();
    for (int s = 1; s < 1024; s <<= 1) {
        uint32_t a = threadIdx.x >= (unsigned)s ? part[threadIdx.x - s] : 0;
        __syncthreads();
        part[threadIdx.x] += a;
        __syncthreads();
    }
    return part[threadIdx.x];
}
static __global__ __launch_bounds__(1024) void plan_scan_tiles_local(const uint32_t *__restrict__ totals,
                                                              uint32_t *__restrict__ tile_begin,
                                                              uint32_t *__restrict__ bsum, int64_t count) {
    __shared__ uint32_t part[1024];
    const int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    const uint32_t v = i < count ? totals[i] : 0;
    const uint32_t inc = block_scan_1024(v, part);
    if (i < count) tile_begin[i] = inc - v;
    if (threadIdx.x == 1023) bsum[blockIdx.x] = inc;
}
static __global__ __launch_bounds__(1024) void plan_scan_tiles_sums(uint32_t *__restrict__ bsum, int64_t nblocks) {
    __shared__ uint32_t part[1024];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < nblocks; base += 1024) {   // one sweep up to 2^20 buckets
        const int64_t i = base + threadIdx.x;
        const uint32_t v = i < nblocks ? bsum[i] : 0;
        const uint32_t inc = block_scan_1024(v, part);
        if (i < nblocks) bsum[i] = carry + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) bsum[nblocks] = carry;             // grand total
}
static __global__ __launch_bounds__(1024) void plan_scan_tiles_add(uint32_t *__restrict__ tile_begin,
                                                            const uint32_t *__restrict__ bsum, int64_t count,
                                                            int64_t nblocks) {
    const int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    if (i < count) tile_begin[i] += bsum[blockIdx.x];
    if (i == count) tile_begin[count] = bsum[nblocks];
}

// (chunks, N): give every sample a slot inside its tile bucket (any order), remember who sits there
static __global__ __launch_bounds__(256) void plan_scatter(const float *__restrict__ grid, const float *__restrict__ offset,
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
            float2 g = *reinterpret_cast<const float2 *>(grid + d.gpt(n, p) * 2);
            Geo2 q = locate(g.x, g.y, d, f, off, pl.ntx);
            if (q.valid) {
                uint32_t r = atomicAdd(&cursor[q.bin(pl)], 1u);
                if (pl.dense) pl.sorted[r] = (uint32_t)s;                 // bins are cells: this is the final order
                else if (pl.cellb) { pl.key[r] = (uint32_t)p; pl.cellb[r] = (uint8_t)q.cell; }
                else pl.key[r] = ((uint32_t)p << 8) | (uint32_t)q.cell;   // one scattered word per sample
            }
        }
    }
}

// one workgroup per (n, tile): counting sort of the bucket by local cell id -> final order
static __global__ __launch_bounds__(256) void plan_tile_sort(Plan pl, int64_t P) {
    __shared__ uint32_t cnt[CELLS];
    __shared__ uint32_t scan[CELLS];
    const int64_t t = blockIdx.x;
    const uint32_t b0 = pl.tile_begin[t], b1 = pl.tile_begin[t + 1];
    uint32_t *cbeg = pl.cell_begin + t * (CELLS + 1);
    cnt[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t j = b0 + threadIdx.x; j < b1; j += 256) atomicAdd(&cnt[pl.cellb ? pl.cellb[j] : (pl.key[j] & 0xFFu)], 1u);
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
    const uint32_t sbase = (uint32_t)(t / pl.ntiles) * (uint32_t)P;   // n * P
    for (uint32_t j = b0 + threadIdx.x; j < b1; j += 256) {
        const uint32_t k = pl.key[j];
        const uint32_t cell = pl.cellb ? pl.cellb[j] : (k & 0xFFu), pp = pl.cellb ? k : (k >> 8);
        uint32_t pos = atomicAdd(&cnt[cell], 1u);
        pl.sorted[b0 + pos] = sbase + pp;
    }
}

// ------------------------------------------------------------------------------------------------
// point kernels.  Launch: grid (ceil(P/256), N), 256 threads; a wave owns 64 consecutive points of
// one n.  Every point kernel alternates between two lane layouts, exchanging data through the wave's LDS:
//   lane = sample                   for everything p-ordered: coordinates and weights, the channel-major
//                                   streams (256 contiguous bytes per wave instruction; 64-byte segments
//                                   measured 45 % slower), the fat rows;
//   lane = (sample, channel quad)   for the node gathers: CQ = C/4 lanes share a sample, one wave instruction
//                                   fetches 64/CQ whole node rows with 4 float4 registers per lane instead of
//                                   16, so many more requests are in flight per SIMD.
// ------------------------------------------------------------------------------------------------
// Stream accesses (each element touched once per kernel) must not displace the feature table from the L2:
// loads are nontemporal; output stores are sc1 (agent-scope relaxed: written through and dropped from the XCD's
// L2) -- worth it because every wave instruction writes whole lines (256 contiguous bytes); the fat rows leave
// with nontemporal stores (flush_rows).
__device__ __forceinline__ void st_stream_wt(float *p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T>
__device__ __forceinline__ void st_stream_wt(T *p, float v) { stream_store(p, v); }   // 16-bit streams: nontemporal
template <typename T>
__device__ __forceinline__ float ld_stream(const T *p) { return stream_load(p); }

// ---- 16-bit streams (float16 / bfloat16 elements, fp32 arithmetic) --------------------------------------------
// When the channel rows are dword aligned and P is even (Flags::pair16) they move two samples per dword, straight
// between HBM and the wave's LDS rows (StreamRegs / store_rows16 below); otherwise element by element.
// the wave's first sample and how many of its 64 exist
__device__ __forceinline__ int64_t wave_p0(int xcd) { return (int64_t)pblk(xcd).x * 256 + (threadIdx.x & ~63); }
__device__ __forceinline__ int wave_nlive(int64_t P, int xcd) {
    const int64_t r = P - wave_p0(xcd);
    return r <= 0 ? 0 : (r > 64 ? 64 : (int)r);
}
template <typename T>
__device__ __forceinline__ float half_of(uint32_t w, bool hi) {
    const uint16_t b = hi ? (uint16_t)(w >> 16) : (uint16_t)w;
    return (float)__builtin_bit_cast(T, b);
}
template <typename T>
__device__ __forceinline__ uint32_t bits16(float v) { return (uint32_t)__builtin_bit_cast(uint16_t, (T)v); }

__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ float4 fma4(float s, float4 a, float4 acc) {
    return make_float4(fmaf(s, a.x, acc.x), fmaf(s, a.y, acc.y), fmaf(s, a.z, acc.z), fmaf(s, a.w, acc.w));
}
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// record fields (one float/uint per sample, SoA over the wave's 64 samples)
enum { R_NODE = 0, R_WX0 = 4, R_WX1, R_WY0, R_WY1, R_FIELDS };
constexpr uint32_t NO_NODE = 0xFFFFFFFFu;
constexpr int REC_FLOATS = R_FIELDS * 64;   // per wave
constexpr int OUT_LD = 68;                  // row pitch of the result tile: channel quads land 16 banks apart

// phase 1 of point_forward: lane = sample; everything phase 2 needs goes to `rec`
template <int KERNEL>
__device__ __forceinline__ void point_phase1(float *rec, const float *grid, const float *offset, const Dims &d,
                                             const Flags &f, int align) {
    const int lane = threadIdx.x & 63;
    const PBlk wb = pblk(0);        // (the forward stage only: see pblk)
    const int n = wb.n;
    int64_t p = (int64_t)wb.x * 256 + threadIdx.x;
    if (p >= d.P) p = d.P - 1;
    const float off = offset[n];
    float2 g = *reinterpret_cast<const float2 *>(grid + d.gpt(n, p) * 2);
    Axis ax = make_axis<KERNEL, 0>(g.x, d.size[0], f, align, off);
    Axis ay = make_axis<KERNEL, 0>(g.y, d.size[1], f, align, off);
    uint32_t *ru = reinterpret_cast<uint32_t *>(rec);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        int x = ax.lo + (a & 1), y = ay.lo + (a >> 1);
        bool ok = x >= 0 && x < d.size[0] && y >= 0 && y < d.size[1];
        ru[(R_NODE + a) * 64 + lane] = ok ? (uint32_t)(y * d.size[0] + x) : NO_NODE;
    }
    rec[R_WX0 * 64 + lane] = ax.w[0];
    rec[R_WX1 * 64 + lane] = ax.w[1];
    rec[R_WY0 * 64 + lane] = ay.w[0];
    rec[R_WY1 * 64 + lane] = ay.w[1];
}

// what a phase-2 lane knows about its sample
struct QuadSample {
    int sl;            // sample slot inside the wave (0..63)
    int q;             // channel quad
    int64_t p;         // point index inside n (valid only if live)
    bool live;
    uint32_t node[4];
    float wx[2], wy[2];
    float W[4];
    __device__ __forceinline__ void read(const float *rec, int sub, int CQ, const Dims &d) {
        const int lane = threadIdx.x & 63;
        const int per = 64 / CQ;
        sl = sub * per + lane / CQ;
        q = lane % CQ;
        p = wave_p0(0) + sl;
        live = p < d.P;
        const uint32_t *ru = reinterpret_cast<const uint32_t *>(rec);
#pragma unroll
        for (int a = 0; a < 4; ++a) node[a] = ru[(R_NODE + a) * 64 + sl];
        wx[0] = rec[R_WX0 * 64 + sl];
        wx[1] = rec[R_WX1 * 64 + sl];
        wy[0] = rec[R_WY0 * 64 + sl];
        wy[1] = rec[R_WY1 * 64 + sl];
#pragma unroll
        for (int a = 0; a < 4; ++a) W[a] = wx[a & 1] * wy[a >> 1];
    }
};

template <int CQ>
__device__ __forceinline__ void gather4(const float4 *tab, const QuadSample &qs, float4 (&v)[4]) {
#pragma unroll
    for (int a = 0; a < 4; ++a) v[a] = tab[(qs.node[a] == NO_NODE ? 0u : qs.node[a]) * CQ + qs.q];
#pragma unroll
    for (int a = 0; a < 4; ++a)
        if (qs.node[a] == NO_NODE) v[a] = zero4();
}
// the 4 channels 4q..4q+3 of a channel-major stream; channels >= cv do not exist (C < 4 runs zero-padded as CQ = 1)
template <typename T>
__device__ __forceinline__ float4 load_quad(const T *src, int64_t P, int cv) {
    float4 r;
    if constexpr (sizeof(T) == 2) {   // all four loads in flight before the first conversion (see StreamRegs)
        if (cv <= 0) return zero4();
        const int last = cv > 3 ? 3 : cv - 1;
        T t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = __builtin_nontemporal_load(src + (int64_t)(k < last ? k : last) * P);
        r.x = (float)t[0];
        r.y = cv > 1 ? (float)t[1] : 0.0f;
        r.z = cv > 2 ? (float)t[2] : 0.0f;
        r.w = cv > 3 ? (float)t[3] : 0.0f;
        return r;
    }
    r.x = cv > 0 ? ld_stream(src) : 0.0f;   // cv <= 0: a quad of padding channels (C padded up to a supported count)
    r.y = cv > 1 ? ld_stream(src + P) : 0.0f;
    r.z = cv > 2 ? ld_stream(src + 2 * P) : 0.0f;
    r.w = cv > 3 ? ld_stream(src + 3 * P) : 0.0f;
    return r;
}
template <int KERNEL, int CQ, typename ST = float>
__global__ __launch_bounds__(256) void point_forward(const float *__restrict__ icl, const float *__restrict__ grid,
                                                     const float *__restrict__ offset, ST *__restrict__ out,
                                                     Dims d, Flags f) {
    extern __shared__ float lds[];
    constexpr int C = 4 * CQ;
    float *rec = lds + (threadIdx.x >> 6) * REC_FLOATS;
    point_phase1<KERNEL>(rec, grid, offset, d, f, 1);   // 2D forward: align_corners = 1 (2d.cu:307-308)
    __syncthreads();
    const int n = pblk(0).n;
    const float4 *tab = reinterpret_cast<const float4 *>(icl + (int64_t)n * d.vol * C);
    ST *obase = out + (int64_t)n * d.out_ns;   // out_ns: d.C * d.P for a contiguous stream (d.C: the caller's channel count)
    float *ot = lds + 4 * REC_FLOATS + (threadIdx.x >> 6) * (C * OUT_LD);   // this wave's [C][64] result tile
    QuadSample qsv[CQ];
    float4 vv[CQ][4];
#pragma unroll
    for (int sub = 0; sub < CQ; ++sub) {   // every pass's node rows in flight before the first is blended
        qsv[sub].read(rec, sub, CQ, d);
        gather4<CQ>(tab, qsv[sub], vv[sub]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int sub = 0; sub < CQ; ++sub) {
        const QuadSample &qs = qsv[sub];
        const float4(&v)[4] = vv[sub];
        float4 acc = zero4();
#pragma unroll
        for (int a = 0; a < 4; ++a) acc = fma4(qs.W[a], v[a], acc);
        // results go back to lane = sample through LDS: 256 contiguous bytes per store instruction instead of
        // four 64-byte segments (measured: forward 0.70 -> 0.51 ms; the stores alone cost 0.34 ms the other way)
        ot[(4 * qs.q + 0) * OUT_LD + qs.sl] = acc.x;
        ot[(4 * qs.q + 1) * OUT_LD + qs.sl] = acc.y;
        ot[(4 * qs.q + 2) * OUT_LD + qs.sl] = acc.z;
        ot[(4 * qs.q + 3) * OUT_LD + qs.sl] = acc.w;
    }
    __syncthreads();
    const int64_t p = (int64_t)pblk(0).x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    if constexpr (sizeof(ST) == 2) if (f.pair16) {   // lanes 0..31: channel c, lanes 32..63: channel c+1 (store_rows16)
        const int L = lane & 31, up = lane >> 5;
        if (2 * L >= wave_nlive(d.P, 0)) return;
        ST *row = obase + wave_p0(0) + 2 * L;
#pragma unroll
        for (int c0 = 0; c0 < C; c0 += 2) {
            const int c = c0 + up;
            const float2 v = *reinterpret_cast<const float2 *>(ot + c * OUT_LD + 2 * L);
            if (c < d.C)
                __builtin_nontemporal_store(bits16<ST>(v.x) | (bits16<ST>(v.y) << 16),
                                            reinterpret_cast<uint32_t *>(row + (int64_t)c * d.P));
        }
        return;
    }
    if (p < d.P) {
#pragma unroll
        for (int c = 0; c < C; ++c)
            if (c < d.C) st_stream_wt(obase + (int64_t)c * d.P + p, ot[c * OUT_LD + lane]);
    }
}

// ---- backward point kernels: what a lane = sample phase knows about its sample ----
struct Sample2 {
    int n;
    int64_t p, s;
    bool live;
    Axis ax[2];
    uint32_t node[4];  // node index (y*W + x) inside one n; NO_NODE when zero-padded
    float W[4];

    float2 g;          // the coordinates as loaded (begin), consumed by finish
    // begin: which sample, and its coordinate load goes out; finish: the geometry.  The kernels issue their cotangent
    // stream loads in between, so that one memory round trip covers both (a wave used to wait for its coordinates,
    // work out the geometry and only then ask for the streams).
    __device__ __forceinline__ void begin(const float *grid, const Dims &d) {
        const PBlk wb = pblk(d.xcd);
        n = wb.n;
        int64_t pp = (int64_t)wb.x * blockDim.x + threadIdx.x;
        live = pp < d.P;
        p = live ? pp : d.P - 1;
        s = (int64_t)n * d.P + p;
        g = *reinterpret_cast<const float2 *>(grid + d.gpt(n, p) * 2);
    }
    template <int KERNEL, int ORDER>
    __device__ __forceinline__ void finish(const float *offset, const Dims &d, const Flags &f) {
        float off = offset[n];
        ax[0] = make_axis<KERNEL, ORDER>(g.x, d.size[0], f, f.align, off);
        ax[1] = make_axis<KERNEL, ORDER>(g.y, d.size[1], f, f.align, off);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            int x = ax[0].lo + (a & 1), y = ax[1].lo + (a >> 1);
            bool ok = x >= 0 && x < d.size[0] && y >= 0 && y < d.size[1];
            node[a] = ok ? (uint32_t)(y * d.size[0] + x) : NO_NODE;
            W[a] = ax[0].w[a & 1] * ax[1].w[a >> 1];
        }
    }
    __device__ __forceinline__ float first(int a, int j) const {
        float sgn = ((a >> j) & 1) ? ax[j].d1 : -ax[j].d1;
        return sgn * ax[1 - j].w[(a >> (1 - j)) & 1];
    }
    __device__ __forceinline__ float pure2(int a, int j) const {
        float sgn = ((a >> j) & 1) ? -ax[j].d2 : ax[j].d2;
        return sgn * ax[1 - j].w[(a >> (1 - j)) & 1];
    }
    __device__ __forceinline__ float mixed2(int a) const {   // d2 W_a / dx dy
        float sx = (a & 1) ? ax[0].d1 : -ax[0].d1, sy = (a & 2) ? ax[1].d1 : -ax[1].d1;
        return sx * sy;
    }
};

template <int CQ, typename T>
__device__ __forceinline__ void load_stream(const T *src, int64_t P, float4 (&g)[CQ], int C) {
#pragma unroll
    for (int q = 0; q < CQ; ++q) g[q] = load_quad(src + (int64_t)(4 * q) * P, P, C - 4 * q);
}
template <int CQ, typename T>
__device__ __forceinline__ void store_stream(T *dst, int64_t P, const float4 (&o)[CQ], int C) {
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
        T *p = dst + (int64_t)(4 * q) * P;
        const int cv = C - 4 * q;
        if (cv > 0) st_stream_wt(p, o[q].x);
        if (cv > 1) st_stream_wt(p + P, o[q].y);
        if (cv > 2) st_stream_wt(p + 2 * P, o[q].z);
        if (cv > 3) st_stream_wt(p + 3 * P, o[q].w);
    }
}
template <int CQ>
__device__ __forceinline__ void put_payload(float *row, const float4 (&g)[CQ]);
// 16-bit stream <-> the wave's LDS rows without going through lane = sample registers: lanes 0..31 take the 32 sample
// pairs of channel c, lanes 32..63 those of channel c+1 -- every instruction moves two contiguous 128-byte runs, each
// covered by consecutive lanes.  `chan0_n` = channel 0 of this n at sample 0.
template <int CQ, typename T>
__device__ __forceinline__ void store_rows16(const float *stage, int stride, T *base, int64_t P, int nlive, int C) {
    const int lane = threadIdx.x & 63, L = lane & 31, up = lane >> 5;
    if (2 * L >= nlive) return;
    const float *r0 = stage + (2 * L) * stride + up;
#pragma unroll
    for (int i = 0; i < 2 * CQ; ++i) {
        const int c = 2 * i + up;
        const uint32_t w = bits16<T>(r0[2 * i]) | (bits16<T>(r0[stride + 2 * i]) << 16);
        if (c < C) __builtin_nontemporal_store(w, reinterpret_cast<uint32_t *>(base + (int64_t)c * P + 2 * L));
    }
}
// A cotangent stream on its way to the wave's stage rows, in two steps so that the loads can be issued early:
// issue() sends the loads (no use of the data: nothing waits), to_rows() converts and writes the LDS rows -- this
// lane's row, or with 16-bit pairs two channels of every row of the wave (callers sync before reading rows).
template <int CQ, typename T>
struct StreamRegs {
    float4 g[CQ];          // fp32 streams
    T t[4 * CQ];           // 16-bit streams, element path
    uint32_t w[2 * CQ];    // 16-bit streams, dword path: lanes 0..31 hold the 32 sample pairs of channel 2i, lanes 32..63 of 2i+1
    __device__ __forceinline__ void issue(const T *chan0_n, int64_t p, int64_t P, int C, bool pair16) {
        if constexpr (sizeof(T) == 2) {
            if (pair16) {
                const int lane = threadIdx.x & 63, L = lane & 31, up = lane >> 5;
                const bool live = 2 * L < wave_nlive(P, 0);      // (16-bit streams: always the launch order, Dims::xcd)
                const T *base = chan0_n + (live ? wave_p0(0) + 2 * L : 0);
#pragma unroll
                for (int i = 0; i < 2 * CQ; ++i) {
                    const int c = 2 * i + up;
                    w[i] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(base + (int64_t)(c < C ? c : C - 1) * P));
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4 * CQ; ++k) t[k] = __builtin_nontemporal_load(chan0_n + p + (int64_t)(k < C ? k : C - 1) * P);
            }
        } else {
            load_stream<CQ>(chan0_n + p, P, g, C);
        }
    }
    __device__ __forceinline__ void to_rows(float *stage, int stride, int off, int64_t P, int C, bool pair16) {
        const int lane = threadIdx.x & 63;
        if constexpr (sizeof(T) == 2) {
            if (pair16) {
                const int L = lane & 31, up = lane >> 5;
                const bool live = 2 * L < wave_nlive(P, 0);
                float *r0 = stage + off + (2 * L) * stride + up;
#pragma unroll
                for (int i = 0; i < 2 * CQ; ++i) {
                    const bool ok = live && 2 * i + up < C;
                    r0[2 * i] = ok ? half_of<T>(w[i], false) : 0.0f;
                    r0[stride + 2 * i] = ok ? half_of<T>(w[i], true) : 0.0f;
                }
                return;
            }
#pragma unroll
            for (int q = 0; q < CQ; ++q)
                g[q] = make_float4(4 * q < C ? (float)t[4 * q] : 0.0f, 4 * q + 1 < C ? (float)t[4 * q + 1] : 0.0f,
                                   4 * q + 2 < C ? (float)t[4 * q + 2] : 0.0f, 4 * q + 3 < C ? (float)t[4 * q + 3] : 0.0f);
        }
        put_payload<CQ>(stage + lane * stride + off, g);
    }
};

// fat row of sample s: payload(s) then coefficient record(s), contiguous, 16-byte aligned.
// A lane first puts its row into the wave's LDS stage; flush_rows then streams the wave's 64 rows
// (one contiguous 64*STRIDE*4-byte block of the p-ordered array) with consecutive lanes writing
// consecutive 16-byte pieces -- a lane-strided struct store would touch every line 5-10 times.
template <int CQ>
__device__ __forceinline__ void put_payload(float *row, const float4 (&g)[CQ]) {
#pragma unroll
    for (int q = 0; q < CQ; ++q) *reinterpret_cast<float4 *>(row + 4 * q) = g[q];
}
// 16-byte stores of p-ordered records and of the sorted grad_output copy: through a buffer descriptor of the wave's /
// workgroup's own range (wave-uniform base, 32-bit lane offset) so that the cache policy is a parameter -- aux 2 =
// nontemporal, 16 = sc1 (written through and dropped from the XCD's L2), 0 = plain.
#ifndef CS_ROWS_AUX
#define CS_ROWS_AUX 2
#endif
#ifndef CS_EMIT_AUX
#define CS_EMIT_AUX 0
#endif
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
template <int AUX>
__device__ __forceinline__ void st_row16(__amdgpu_buffer_rsrc_t r, uint32_t byteoff, float4 v) {
    const v4f_t t = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_t, t), r, (int)byteoff, 0, AUX);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_rsrc(const void *uniform_base) {
    const uint64_t a = (uint64_t)uniform_base;       // wave-uniform by construction; say so to the compiler
    const uint64_t u = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                       (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    return __builtin_amdgcn_make_buffer_rsrc((void *)u, 0, -1, 0x00020000);
}
template <int STRIDE>
__device__ __forceinline__ void flush_rows(const float *stage, float *fat, int n, const Dims &d) {
    const int lane = threadIdx.x & 63;
    const int64_t p0 = wave_p0(d.xcd);     // first point of this wave
    if (p0 >= d.P) return;
    const int nlive = (int)min((int64_t)64, d.P - p0);
    const __amdgpu_buffer_rsrc_t dst = rows_rsrc(fat + ((int64_t)n * d.P + p0) * STRIDE);
    const float4 *src = reinterpret_cast<const float4 *>(stage);
    constexpr int PIECES = STRIDE / 4;                                        // float4 per row
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        int item = i * 64 + lane;
        // nontemporal: the rows are next read by the tile kernel, long after they left the caches (-0.28 ms per step
        // against plain stores)
        if (item < nlive * PIECES) st_row16<CS_ROWS_AUX>(dst, (uint32_t)item * 16u, src[item]);
    }
}

// ------------------------------------------------------------------------------------------------
// Three-phase point kernels.  (Their first version kept lane = sample throughout: 16 float4 gathers per lane
// cost 64 VGPRs, 136 in total, 3 waves per SIMD, and the gathers alone took 0.67 ms against 0.29 ms with CQ
// lanes per sample -- an ablation with -D switches, profiles/round1_ablation.txt.)
//   phase 1  lane = sample   geometry, coalesced stream loads; the fat row [payload | coefficients] is built in
//                            the wave's LDS stage, node ids go to `rec`; the rows are flushed to HBM
//   phase 2  CQ lanes = one sample (lane q owns channels 4q..4q+3), 64/CQ samples per pass: 4 float4 gathers
//                            per lane, all passes in flight; cotangent quad read back from the stage row, per
//                            channel results written over it, per sample dot products reduced by shuffles
//   phase 3  lane = sample   results leave as 256 contiguous bytes per store instruction
// ------------------------------------------------------------------------------------------------
constexpr int QREC = 6 * 64;   // per wave: node[4] (uint), then two floats per sample for phase 2 -> 3 results

__device__ __forceinline__ void q_put_nodes(float *rec, const Sample2 &sm) {
    uint32_t *ru = reinterpret_cast<uint32_t *>(rec);
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int a = 0; a < 4; ++a) ru[a * 64 + lane] = sm.node[a];
}
template <int CQ>
__device__ __forceinline__ void q_gather(const float4 *tab, const float *rec, int sl, int q, float4 (&v)[4]) {
    const uint32_t *ru = reinterpret_cast<const uint32_t *>(rec);
    uint32_t node[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) node[a] = ru[a * 64 + sl];
#pragma unroll
    for (int a = 0; a < 4; ++a) v[a] = tab[(node[a] == NO_NODE ? 0u : node[a]) * CQ + q];
#pragma unroll
    for (int a = 0; a < 4; ++a)
        if (node[a] == NO_NODE) v[a] = zero4();
}
// sum over the CQ lanes of a sample (CQ in {1,2,4,8}: lanes of one sample are adjacent)
template <int CQ>
__device__ __forceinline__ float q_reduce(float x) {
    if (CQ >= 2) x += __shfl_xor(x, 1);
    if (CQ >= 4) x += __shfl_xor(x, 2);
    if (CQ >= 8) x += __shfl_xor(x, 4);
    return x;
}
template <int CQ, typename T>
__device__ __forceinline__ void q_store_rows(const float *stage, int stride, T *chan0_n, int64_t p, int64_t P, bool live,
                                             int C, bool pair16) {
    if constexpr (sizeof(T) == 2) if (pair16) {
        store_rows16<CQ>(stage, stride, chan0_n + wave_p0(0), P, wave_nlive(P, 0), C);
        return;
    }
    if (!live) return;
    const float *row = stage + (threadIdx.x & 63) * stride;
    float4 o[CQ];
#pragma unroll
    for (int q = 0; q < CQ; ++q) o[q] = *reinterpret_cast<const float4 *>(row + 4 * q);
    store_stream<CQ, T>(chan0_n + p, P, o, C);
}

// first backward.  LDS stage row = the fat row [g | W0..W3] (flushed when WANT_ROWS: grad_input is wanted);
// `co` = [4][64]: wx0 wx1 wy0 wy1 for the grad_grid dot products of phase 2.
template <int KERNEL, int CQ, bool WANT_ROWS, typename ST = float>
__global__ __launch_bounds__(256) void point_backward(const ST *__restrict__ gOut, const float *__restrict__ icl,
                                                      const float *__restrict__ grid, const float *__restrict__ offset,
                                                      float *__restrict__ fat, float *__restrict__ grad_grid,
                                                      Dims d, Flags f) {
    constexpr int C = 4 * CQ, STRIDE = row1(C), CO = 4 * 64;
    extern __shared__ float lds[];
    float *stage = lds + (threadIdx.x >> 6) * (64 * STRIDE + QREC + CO);
    float *rec = stage + 64 * STRIDE;
    float *co = rec + QREC;
    const int lane = threadIdx.x & 63;
    Sample2 sm;
    sm.begin(grid, d);
    {
        StreamRegs<CQ, ST> sg;
        sg.issue(gOut + (int64_t)sm.n * d.go_ns, sm.p, d.P, d.C, f.pair16);
        sm.finish<KERNEL, 1>(offset, d, f);
        sg.to_rows(stage, STRIDE, 0, d.P, d.C, f.pair16);
        float *row = stage + lane * STRIDE;
        if (WANT_ROWS) *reinterpret_cast<float4 *>(row + C) = make_float4(sm.W[0], sm.W[1], sm.W[2], sm.W[3]);
        co[lane] = sm.ax[0].w[0];
        co[64 + lane] = sm.ax[0].w[1];
        co[128 + lane] = sm.ax[1].w[0];
        co[192 + lane] = sm.ax[1].w[1];
        q_put_nodes(rec, sm);
    }
    __syncthreads();
    const float4 *tab = reinterpret_cast<const float4 *>(icl + (int64_t)sm.n * d.vol * C);
    const int q = lane % CQ;
    float4 vv[CQ][4];
#pragma unroll
    for (int sub = 0; sub < CQ; ++sub) q_gather<CQ>(tab, rec, sub * (64 / CQ) + lane / CQ, q, vv[sub]);
    __builtin_amdgcn_sched_barrier(0);   // every pass's gathers issued before anything consumes one
    if (WANT_ROWS) flush_rows<STRIDE>(stage, fat, sm.n, d);   // while the gathers are in flight
#pragma unroll
    for (int sub = 0; sub < CQ; ++sub) {
        const int sl = sub * (64 / CQ) + lane / CQ;
        const float4(&v)[4] = vv[sub];
        const float4 g4 = *reinterpret_cast<const float4 *>(stage + sl * STRIDE + 4 * q);
        const float wx0 = co[sl], wx1 = co[64 + sl], wy0 = co[128 + sl], wy1 = co[192 + sl];
        float d0 = dot4(v[0], g4), d1 = dot4(v[1], g4), d2 = dot4(v[2], g4), d3 = dot4(v[3], g4);
        float gx = q_reduce<CQ>(wy0 * (d1 - d0) + wy1 * (d3 - d2));
        float gy = q_reduce<CQ>(wx0 * (d2 - d0) + wx1 * (d3 - d1));
        if (q == 0) {
            rec[4 * 64 + sl] = gx;
            rec[5 * 64 + sl] = gy;
        }
    }
    __syncthreads();
    if (sm.live)
        *reinterpret_cast<float2 *>(grad_grid + sm.s * 2) =
            make_float2(sm.ax[0].d1 * rec[4 * 64 + lane], sm.ax[1].d1 * rec[5 * 64 + lane]);
}

// second backward.  LDS stage row: g | Dm[4] (the fat row) and, in `rec`, node ids; the per sample coefficient
// sets Sx, Sy (and W for HAS_CI) live in a second record block `co`: [12][64]
// ROWS: 0 nothing for the scatter (grad_input not wanted), 1 the fat rows [gOut | D], 2 the 16-byte D record alone
// (p-ordered, `fat` = S float4): the walkers then take gOut from the sorted copy an earlier stage left (Plan::Gs)
template <int KERNEL, int CQ, bool HAS_CI, int ROWS, typename ST = float>
__global__ __launch_bounds__(256) void point_bb(const float *__restrict__ cIcl, const float *__restrict__ cG,
                                                  const float *__restrict__ icl, const float *__restrict__ grid,
                                                  const ST *__restrict__ gOut, const float *__restrict__ offset,
                                                  float *__restrict__ fat, float *__restrict__ gGrid,
                                                  ST *__restrict__ ggOut, Dims d, Flags f) {
    constexpr int C = 4 * CQ, STRIDE = row1(C), CO = 12 * 64;
    extern __shared__ float lds[];
    float *stage = lds + (threadIdx.x >> 6) * (64 * STRIDE + QREC + CO);
    float *rec = stage + 64 * STRIDE;
    float *co = rec + QREC;
    const int lane = threadIdx.x & 63;
    Sample2 sm;
    sm.begin(grid, d);
    {
        float2 cg = cG ? *reinterpret_cast<const float2 *>(cG + d.gpt(sm.n, sm.p) * 2) : make_float2(0.f, 0.f);
        StreamRegs<CQ, ST> sg;
        sg.issue(gOut + (int64_t)sm.n * d.go_ns, sm.p, d.P, d.C, f.pair16);
        sm.finish<KERNEL, 2>(offset, d, f);
        sg.to_rows(stage, STRIDE, 0, d.P, d.C, f.pair16);
        float *row = stage + lane * STRIDE;
        float Dm[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            Dm[a] = sm.first(a, 0) * cg.x + sm.first(a, 1) * cg.y;
            float sx = sm.pure2(a, 0) * cg.x, sy = sm.pure2(a, 1) * cg.y;   // 2D keeps pure second derivatives only
            if (f.exact) {                                                    // (2d.cu:705-706) unless asked otherwise
                const float mx = sm.mixed2(a);
                sx = fmaf(mx, cg.y, sx);
                sy = fmaf(mx, cg.x, sy);
            }
            co[a * 64 + lane] = sx;
            co[(4 + a) * 64 + lane] = sy;
            if (HAS_CI) co[(8 + a) * 64 + lane] = sm.W[a];
        }
        *reinterpret_cast<float4 *>(row + C) = make_float4(Dm[0], Dm[1], Dm[2], Dm[3]);
        if (ROWS == 2 && sm.live) {
            const int64_t s0w = (int64_t)sm.n * d.P + wave_p0(d.xcd);          // the wave's first sample: uniform
            st_row16<CS_ROWS_AUX>(rows_rsrc(fat + s0w * 4), (uint32_t)(threadIdx.x & 63) * 16u, make_float4(Dm[0], Dm[1], Dm[2], Dm[3]));
        }
        q_put_nodes(rec, sm);
    }
    __syncthreads();
    const float4 *tab = reinterpret_cast<const float4 *>(icl + (int64_t)sm.n * d.vol * C);
    const float4 *ctab = HAS_CI ? reinterpret_cast<const float4 *>(cIcl + (int64_t)sm.n * d.vol * C) : nullptr;
    const int q = lane % CQ;
    float4 vv[CQ][4];   // every pass's gathers are issued before any result is written (LDS writes would fence them)
#pragma unroll
    for (int sub = 0; sub < CQ; ++sub) q_gather<CQ>(tab, rec, sub * (64 / CQ) + lane / CQ, q, vv[sub]);
    __builtin_amdgcn_sched_barrier(0);   // every pass's gathers issued before anything consumes one
    if (ROWS == 1) flush_rows<STRIDE>(stage, fat, sm.n, d);   // while the gathers are in flight
#pragma unroll
    for (int sub = 0; sub < CQ; ++sub) {
        const int sl = sub * (64 / CQ) + lane / CQ;
        const float4(&v)[4] = vv[sub];
        float *row = stage + sl * STRIDE;
        const float4 g4 = *reinterpret_cast<const float4 *>(row + 4 * q);
        const float4 Dm = *reinterpret_cast<const float4 *>(row + C);
        float4 acc = zero4(), tx = zero4(), ty = zero4();
        const float dm[4] = {Dm.x, Dm.y, Dm.z, Dm.w};
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            acc = fma4(dm[a], v[a], acc);
            tx = fma4(co[a * 64 + sl], v[a], tx);
            ty = fma4(co[(4 + a) * 64 + sl], v[a], ty);
        }
        if (HAS_CI) {   // + sum_a gOutInput[q_a] * W_a   (2d.cu:694-697)
            float4 u[4];
            q_gather<CQ>(ctab, rec, sl, q, u);
#pragma unroll
            for (int a = 0; a < 4; ++a) acc = fma4(co[(8 + a) * 64 + sl], u[a], acc);
        }
        float sx = q_reduce<CQ>(dot4(tx, g4)), sy = q_reduce<CQ>(dot4(ty, g4));
        *reinterpret_cast<float4 *>(row + 4 * q) = acc;   // over the cotangent quad this lane just consumed
        if (q == 0) {
            rec[4 * 64 + sl] = sx;
            rec[5 * 64 + sl] = sy;
        }
    }
    __syncthreads();
    q_store_rows<CQ>(stage, STRIDE, ggOut + (int64_t)sm.n * d.out_ns, sm.p, d.P, sm.live, d.C, f.pair16);
    if (sm.live) *reinterpret_cast<float2 *>(gGrid + sm.s * 2) = make_float2(rec[4 * 64 + lane], rec[5 * 64 + lane]);
}

// fused third backward.  Stage row = the fat row: TWO ? [gOut | hO | E | D] : [gOut | E];
// LEAN (with TWO): [hO | E | D] -- grad_output is neither read nor written again, the walkers stream its sorted copy
// (padding these rows to whole 128-byte lines was measured and dropped: point kernel +0.09 ms, walkers -0.02 ms)
__host__ __device__ constexpr int row3(int C) { return C + 8; }
template <int KERNEL, int CQ, bool TWO, bool LEAN = false, typename ST = float>
__global__ __launch_bounds__(256) void point_bbb(const float *__restrict__ icl, const float *__restrict__ grid,
                                                   const ST *__restrict__ gOut, const float *__restrict__ cG,
                                                   const float *__restrict__ hG, const ST *__restrict__ hO,
                                                   const float *__restrict__ offset, float *__restrict__ fat,
                                                   ST *__restrict__ ggOut, Dims d, Flags f) {
    static_assert(!LEAN || TWO, "the lean rows carry grad_out_ggout");
    constexpr int C = 4 * CQ, STRIDE = LEAN ? row3(C) : TWO ? row2(C) : row1(C), EOFF = LEAN ? C : TWO ? 2 * C : C;
    extern __shared__ float lds[];
    float *stage = lds + (threadIdx.x >> 6) * (64 * STRIDE + QREC);
    float *rec = stage + 64 * STRIDE;
    const int lane = threadIdx.x & 63;
    Sample2 sm;
    sm.begin(grid, d);
    {
        float2 cg = cG ? *reinterpret_cast<const float2 *>(cG + d.gpt(sm.n, sm.p) * 2) : make_float2(0.f, 0.f);
        float2 hg = hG ? *reinterpret_cast<const float2 *>(hG + d.gpt(sm.n, sm.p) * 2) : make_float2(0.f, 0.f);
        StreamRegs<CQ, ST> sg, sh;
        if (!LEAN) sg.issue(gOut + (int64_t)sm.n * d.go_ns, sm.p, d.P, d.C, f.pair16);
        if (TWO) sh.issue(hO + (int64_t)sm.n * d.ho_ns, sm.p, d.P, d.C, f.pair16);
        sm.finish<KERNEL, 2>(offset, d, f);
        float Dm[4], Em[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            Dm[a] = sm.first(a, 0) * cg.x + sm.first(a, 1) * cg.y;
            Em[a] = sm.pure2(a, 0) * (hg.x * cg.x) + sm.pure2(a, 1) * (hg.y * cg.y);   // 2d.cu:876
            if (f.exact) Em[a] = fmaf(sm.mixed2(a), hg.x * cg.y + hg.y * cg.x, Em[a]);
        }
        float *row = stage + lane * STRIDE;
        if (!LEAN) sg.to_rows(stage, STRIDE, 0, d.P, d.C, f.pair16);
        if (TWO) {
            sh.to_rows(stage, STRIDE, LEAN ? 0 : C, d.P, d.C, f.pair16);
            *reinterpret_cast<float4 *>(row + EOFF + 4) = make_float4(Dm[0], Dm[1], Dm[2], Dm[3]);
        }
        *reinterpret_cast<float4 *>(row + EOFF) = make_float4(Em[0], Em[1], Em[2], Em[3]);
        q_put_nodes(rec, sm);
    }
    __syncthreads();
    const float4 *tab = reinterpret_cast<const float4 *>(icl + (int64_t)sm.n * d.vol * C);
    const int q = lane % CQ;
    float4 vv[CQ][4];   // every pass's gathers are issued before any result is written (LDS writes would fence them)
#pragma unroll
    for (int sub = 0; sub < CQ; ++sub) q_gather<CQ>(tab, rec, sub * (64 / CQ) + lane / CQ, q, vv[sub]);
    __builtin_amdgcn_sched_barrier(0);   // every pass's gathers issued before anything consumes one
    flush_rows<STRIDE>(stage, fat, sm.n, d);   // while the gathers are in flight
#pragma unroll
    for (int sub = 0; sub < CQ; ++sub) {
        const int sl = sub * (64 / CQ) + lane / CQ;
        const float4(&v)[4] = vv[sub];
        float *row = stage + sl * STRIDE;
        const float4 E = *reinterpret_cast<const float4 *>(row + EOFF);
        float4 acc = fma4(E.x, v[0], zero4());
        acc = fma4(E.y, v[1], acc);
        acc = fma4(E.z, v[2], acc);
        acc = fma4(E.w, v[3], acc);
        *reinterpret_cast<float4 *>(row + 4 * q) = acc;   // over the (already flushed) gOut quad
    }
    __syncthreads();
    q_store_rows<CQ>(stage, STRIDE, ggOut + (int64_t)sm.n * d.out_ns, sm.p, d.P, sm.live, d.C, f.pair16);
}

// ------------------------------------------------------------------------------------------------
// tile kernel: grad_input[n,c,node] += sum over the tile's samples of coef_a * payload[c]
// one workgroup per (n, tile).  CQ lanes = one walker (lane q owns channels 4q..4q+3); walker
// (ly, seg) owns cells [seg*CQ, (seg+1)*CQ) of cell row ly: 256/CQ walkers = TY rows x TX/CQ runs.
// ------------------------------------------------------------------------------------------------
// fat-row loads stay plain: the CQ lanes of a walker and the payload / coefficient loads of a lane share lines
// (nontemporal loads here: tile kernels 15 % slower)
__device__ __forceinline__ float4 ld_row(const float *p) { return *reinterpret_cast<const float4 *>(p); }
template <int CQ>
constexpr size_t tile_scatter_lds() {   // top + bot images + the cell table
    return (size_t)2 * TY * (TX / CQ) * (CQ + 1) * CQ * 16 + (CELLS + 1 + 3) / 4 * 16;
}
// SRC says where a sample's payload(s) and coefficients come from:
//   0  fat rows [gOut | k]                 fetched by sample id          (first / second backward)
//   1  fat rows [gOut | hO | E | D]        fetched by sample id          (fused third backward)
//   2  gOut streamed from Plan::Gs (sorted), the 16-byte record k fetched by id          (second backward)
//   3  gOut streamed from Plan::Gs, rows [hO | E | D] fetched by id                      (fused third backward)
// EMIT (SRC 0, 1): the gOut quads also leave in sorted order to Plan::Gs -- sequential 64-byte rows, each walker its run
template <int CQ, int SRC, bool EMIT = false>
__global__ __launch_bounds__(256) void tile_scatter(const float *__restrict__ fat, Plan pl,
                                                    float *__restrict__ grad_input, Dims d) {
    constexpr int C = 4 * CQ;
    constexpr bool TWO = SRC == 1 || SRC == 3;
    constexpr int STRIDE = SRC == 0 ? row1(C) : SRC == 1 ? row2(C) : SRC == 2 ? 4 : row3(C);
    static_assert(!EMIT || SRC <= 1, "only the rows that carry gOut can emit it");
    constexpr int SEGW = CQ;                   // cells per walker
    constexpr int NSEG = TX / SEGW;            // walkers per cell row
    constexpr int NODES = NSEG * (SEGW + 1);   // node slots per cell row (run ends are duplicated)
    constexpr int U = TWO ? 4 : 8;             // samples in flight per walker (2..16 measured alike)
    static_assert(TY * NSEG * CQ == 256, "one workgroup = all walkers of a tile");
    extern __shared__ float4 tile_lds[];       // dynamic: 74 KiB at CQ = 8 (tile_scatter_lds)
    float4 *top = tile_lds;                    // sums for the nodes on the low-y side of each cell row
    float4 *bot = top + TY * NODES * CQ;       // ... on the high-y side
    uint32_t *cb = reinterpret_cast<uint32_t *>(bot + TY * NODES * CQ);   // bucket-relative first position of every cell

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

    const int w = threadIdx.x / CQ, q = threadIdx.x % CQ;
    const int ly = w / NSEG, seg = w % NSEG;
    {
        float4 ct = zero4(), cbm = zero4();     // sums carried to the next cell: its left nodes are our right nodes
        float4 a0 = zero4(), a1 = zero4(), a2 = zero4(), a3 = zero4();
        int cur = 0;                            // current cell of the run
        float4 *trow = top + ((ly * NSEG + seg) * (SEGW + 1)) * CQ + q;
        float4 *brow = bot + ((ly * NSEG + seg) * (SEGW + 1)) * CQ + q;
        const uint32_t *cbr = cb + ly * TX + seg * SEGW;
        const uint32_t j1 = cbr[SEGW];
        uint32_t nb = cbr[1];                   // first position of the next cell
        const uint32_t *sorted = pl.sorted + b0;
        float *gs = pl.Gs + (int64_t)b0 * C + 4 * q;     // this bucket's rows of the sorted gOut copy, this lane's quad
        const __amdgpu_buffer_rsrc_t gs_r = rows_rsrc(pl.Gs + (int64_t)b0 * C);   // (the same rows, for the EMIT stores)
        const uint32_t jbeg = cbr[0];
        uint32_t ids[U];                        // sample ids of the NEXT batch: fetched one batch ahead so that
#pragma unroll                                  // the row fetches never wait on the id fetch
        for (int u = 0; u < U; ++u) ids[u] = jbeg < j1 ? sorted[min(jbeg + u, j1 - 1)] : 0u;
        for (uint32_t j = jbeg; j < j1; j += U) {
            float4 g[U], k[U], h[U], k2[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {       // all row loads of the batch first
                const float *row = fat + (int64_t)ids[u] * STRIDE;
                if (SRC == 0) {
                    g[u] = ld_row(row + 4 * q);
                    k[u] = ld_row(row + C);
                } else if (SRC == 1) {
                    g[u] = ld_row(row + 4 * q);
                    h[u] = ld_row(row + C + 4 * q);
                    k[u] = ld_row(row + 2 * C);
                    k2[u] = ld_row(row + 2 * C + 4);
                } else {
                    g[u] = ld_row(gs + (int64_t)min(j + u, j1 - 1) * C);
                    if (SRC == 2) {
                        k[u] = ld_row(row);
                    } else {
                        h[u] = ld_row(row + 4 * q);
                        k[u] = ld_row(row + C);
                        k2[u] = ld_row(row + C + 4);
                    }
                }
            }
            if (EMIT) {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (j + u < j1) st_row16<CS_EMIT_AUX>(gs_r, (uint32_t)(((j + u) * C + 4 * q) * 4), g[u]);
            }
            if (j + U < j1) {
#pragma unroll
                for (int u = 0; u < U; ++u) ids[u] = sorted[min(j + U + u, j1 - 1)];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (j + u < j1) {
                    while (j + u >= nb) {       // close cells up to the one holding this position
                        trow[cur * CQ] = make_float4(ct.x + a0.x, ct.y + a0.y, ct.z + a0.z, ct.w + a0.w);
                        brow[cur * CQ] = make_float4(cbm.x + a2.x, cbm.y + a2.y, cbm.z + a2.z, cbm.w + a2.w);
                        ct = a1; cbm = a3;
                        a0 = a1 = a2 = a3 = zero4();
                        ++cur;
                        nb = cbr[cur + 1];
                    }
                    a0 = fma4(k[u].x, g[u], a0); a1 = fma4(k[u].y, g[u], a1);
                    a2 = fma4(k[u].z, g[u], a2); a3 = fma4(k[u].w, g[u], a3);
                    if (TWO) {
                        a0 = fma4(k2[u].x, h[u], a0); a1 = fma4(k2[u].y, h[u], a1);
                        a2 = fma4(k2[u].z, h[u], a2); a3 = fma4(k2[u].w, h[u], a3);
                    }
                }
            }
        }
        while (cur < SEGW) {
            trow[cur * CQ] = make_float4(ct.x + a0.x, ct.y + a0.y, ct.z + a0.z, ct.w + a0.w);
            brow[cur * CQ] = make_float4(cbm.x + a2.x, cbm.y + a2.y, cbm.z + a2.z, cbm.w + a2.w);
            ct = a1; cbm = a3;
            a0 = a1 = a2 = a3 = zero4();
            ++cur;
        }
        trow[SEGW * CQ] = ct;
        brow[SEGW * CQ] = cbm;
    }
    __syncthreads();

    // node (ly, lx) of the tile = global node (ty*TY + ly - 1, tx*TX + lx - 1).  Column lx is slot
    // lx % SEGW of run lx / SEGW and, when lx is a run boundary, also the last slot of the run before.
    const int W = d.size[0], H = d.size[1];
    const float *topf = reinterpret_cast<const float *>(top), *botf = reinterpret_cast<const float *>(bot);
    float *gi = grad_input + (int64_t)n * d.C * d.vol;
    for (int idx = threadIdx.x; idx < d.C * (TY + 1) * (TX + 1); idx += 256) {   // ch < d.C: padded channels are dropped
        int lx = idx % (TX + 1);
        int rest = idx / (TX + 1);
        int lyy = rest % (TY + 1);
        int ch = rest / (TY + 1);
        int gx = tx * TX + lx - 1, gy = ty * TY + lyy - 1;
        if (gx < 0 || gx >= W || gy < 0 || gy >= H) continue;
        int sg = lx / SEGW, sl = lx - sg * SEGW;
        float v = 0.f;
#pragma unroll
        for (int side = 0; side < 2; ++side) {          // side 0: this run's slot; side 1: previous run's end slot
            int s2 = side ? sg - 1 : sg, l2 = side ? SEGW : sl;
            if (side && sl != 0) continue;
            if (s2 < 0 || s2 >= NSEG) continue;
            if (lyy < TY) v += topf[((lyy * NSEG + s2) * (SEGW + 1) + l2) * C + ch];
            if (lyy > 0) v += botf[(((lyy - 1) * NSEG + s2) * (SEGW + 1) + l2) * C + ch];
        }
        if (v != 0.f) unsafeAtomicAdd(gi + (int64_t)ch * d.vol + (int64_t)gy * W + gx, v);
    }
}

// ------------------------------------------------------------------------------------------------
// cell_scatter: the walkers' job for CROWDED tables (PIXEL's own: 96 tables of 16x16 cells, 10^5-10^6 points:
// hundreds to thousands of samples per cell, and a whole table is one tile).  There a walker would run through
// its cells' samples one by one while most of the chip idles (tile_scatter: 0.58 ms for 9.6 M samples of 16 B).
// Here the plan's bins are the cells themselves (Plan::dense) and one WAVE owns one (n, cell) bucket: lanes
// stride through the bucket, each fetching the fat rows of its samples and keeping 4 node sums of C channels, and
// a halving exchange (lane pairs 32, 16, ... apart swap half of what they hold) leaves every lane with one fully
// reduced (node, channel) value -- 4C-1 shuffles per cell instead of 6 x 4C -- which it adds to grad_input.
// ------------------------------------------------------------------------------------------------
template <int CQ, bool TWO>
__global__ __launch_bounds__(256) void cell_scatter(const float *__restrict__ fat, Plan pl,
                                                    float *__restrict__ grad_input, Dims d) {
    constexpr int C = 4 * CQ, NV = 4 * C;
    constexpr int STRIDE = TWO ? row2(C) : row1(C);
    const int64_t bucket = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bucket >= (int64_t)d.N * pl.ntiles) return;
    const uint32_t b0 = pl.tile_begin[bucket], b1 = pl.tile_begin[bucket + 1];
    if (b0 == b1) return;
    const int lane = threadIdx.x & 63;
    float v[NV];   // v[a * C + c]
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = 0.f;
#pragma unroll 2
    for (uint32_t j = b0 + lane; j < b1; j += 64) {
        const float *row = fat + (int64_t)pl.sorted[j] * STRIDE;
        const float4 k = ld_row(row + (TWO ? 2 * C : C));
        float4 k2 = zero4();
        if (TWO) k2 = ld_row(row + 2 * C + 4);
#pragma unroll
        for (int q = 0; q < CQ; ++q) {
            const float4 g = ld_row(row + 4 * q);
            float4 h = zero4();
            if (TWO) h = ld_row(row + C + 4 * q);
            const float ka[4] = {k.x, k.y, k.z, k.w}, kb[4] = {k2.x, k2.y, k2.z, k2.w};
            const float gc[4] = {g.x, g.y, g.z, g.w}, hc[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = fmaf(ka[a], gc[e], v[a * C + 4 * q + e]);
                    if (TWO) t = fmaf(kb[a], hc[e], t);
                    v[a * C + 4 * q + e] = t;
                }
        }
    }
    // halving exchange: at the step with partner distance m a lane keeps the half of its values selected by its
    // bit m and receives the partner's copy of that half; six steps (m = 32 .. 1) leave NV / 64 fully reduced values
    // per lane (NV = 128: values 2 l and 2 l + 1 on lane l); with NV < 64 the last steps are plain butterflies and
    // 64 / NV lanes share value l / (64 / NV)
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
    const int n = (int)(bucket / pl.ntiles), cell = (int)(bucket - (int64_t)n * pl.ntiles);
    const int uy = cell / pl.ntx, ux = cell - uy * pl.ntx;
#pragma unroll
    for (int jv = 0; jv < VALS; ++jv) {
        const int idx = (lane / SHARE) * VALS + jv, a = idx / C, c = idx % C;
        const int x = ux - 1 + (a & 1), y = uy - 1 + (a >> 1);
        const float r = v[jv];
        if (c >= d.C || x < 0 || x >= d.size[0] || y < 0 || y >= d.size[1] || r == 0.f) continue;
        unsafeAtomicAdd(grad_input + ((int64_t)n * d.C + c) * d.vol + (int64_t)y * d.size[0] + x, r);
    }
}

}  // namespace tiled
}  // namespace cs
