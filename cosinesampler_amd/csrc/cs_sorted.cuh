// cs_sorted.cuh -- round 2: the cell-sorted ("s") domain of the fast 2D path.
//
// Round 1 moved every backward stage's scatter through "fat rows": the point kernel wrote, per sample, one p-ordered
// row [cotangent values | node coefficients] (80-160 bytes) and the tile walkers fetched those rows by sample id -- a
// random fetch and a full rewrite of the SAME grad_output payload in each of the three backward stages, ~11 GB per step
// next to 9.2 GB of algorithmic traffic.  Here the transport happens ONCE per tensor:
//
//   plan (once per grid)      a single-level counting sort by (n, cell): every sample gets `rank[s]`, its position in
//                             cell order, computed in p-order (chunk histogram in LDS as packed 16-bit counters, one
//                             returning LDS atomic per sample, no scattered writes except the 8-byte sorted copy of
//                             the coordinates `coord[rank]`) -- tools/microbench_l2.hip `hist`: 0.07 ms per 2^24.
//   point kernels (p-order)   as before the only place where the table is gathered and every p-ordered output is formed;
//                             the stage's NEW payload leaves as 64-byte channels-last rows scattered to `rank[s]`
//                             (grad_output in the first stage that sees it -> Plan2::G, reused by the later stages;
//                             grad_out_ggout in the fused third backward) and its 8-byte cotangents of the grid likewise.
//   tile kernels (s-order)    stream the sorted payload SEQUENTIALLY (each walker owns a run of cells = one contiguous
//                             piece of the sorted arrays) and evaluate the node coefficients themselves from the sorted
//                             coordinates -- no sample ids, no random fetch, no coefficient rows.
//
// The point kernels are "quad-transposed": in the gather passes lane (sl, q) of pass `sub` works on point CQ*sl + sub of
// its wave, so after the CQ passes it holds channels 4q..4q+3 of CQ CONSECUTIVE points: channel-major streams are read
// and written as runs of CQ floats per lane (a wave instruction touches 256 contiguous bytes in each of CQ planes)
// straight from / into registers, and a lane's own point (lane index) is again the one it handled in phase 1.
// Node rows arrive by LDS-DMA (cs_tiled.cuh, dma_issue).
//
// Applies when the cell histogram fits the LDS as 16-bit counters ((W+1)(H+1) <= 75 000: a 256x256 table has 66 049)
// and the table is not a crowded one (those keep round 1's wave-per-cell path); everything else keeps round 1's path.
// Reference maths per stage: cs_kernels_direct.cuh (same formulas, same quirks; 2d.cu:464-505, :661-712, :850-888).
#pragma once
#include "cs_tiled.cuh"

namespace cs {
namespace sorted {

using namespace cs::tiled;

constexpr uint32_t NONE = 0xFFFFFFFFu;   // rank of a sample that touches no node
constexpr int P2_THREADS = 1024;
constexpr int P2_CHUNK = 32768;          // points per plan workgroup: counts of a (chunk, cell) fit 16 bits
constexpr int P2_MAX_BINS = 75000;       // 2 bytes each in LDS (<= 150 000 of the 160 KiB)

struct Plan2 {
    uint32_t *rank;        // [S]           sample n*P+p -> position in (n, cell) order; NONE: dropped
    float2 *coord;         // [S]           sorted position -> grid coordinates of the sample sitting there
    uint32_t *cell_begin;  // [N*bins + 1]  first sorted position of every (n, cell), cells row-major (uy, ux)
    uint32_t *cnt;         // scratch [N*chunks*words]  packed 16-bit counts of every (chunk, cell)
    uint32_t *excl;        // scratch [N*chunks*bins]   first sorted position of every (chunk, cell)
    float *G;              // [S*CP]        cell-sorted channels-last copy of grad_output (whoever wrote it last)
    float2 *cG;            // [S]           cell-sorted copy of grad_out_grid
    int bx, by, bins, words, chunks, chunk;   // bins = bx*by, bx = W+1, by = H+1 (cell u = low node + 1)
    int ntx, nty;          // 16x16-cell tiles of the scatter kernels
};

// ------------------------------------------------------------------------------------------------
// plan
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t cnt_get(const uint32_t *w, int b) { return (b & 1) ? (w[b >> 1] >> 16) : (w[b >> 1] & 0xffffu); }

// (chunks, N) workgroups of 1024: histogram of one chunk of one n over ALL cells, in LDS, two 16-bit counters per word;
// the value the atomic returns is the sample's rank inside its (chunk, cell) -> rank[s] (finished by p2_rank)
__global__ __launch_bounds__(P2_THREADS) void p2_count(const float *__restrict__ grid, const float *__restrict__ offset,
                                                      Plan2 pl, Dims d, Flags f) {
    extern __shared__ uint32_t h[];
    for (int i = threadIdx.x; i < pl.words; i += P2_THREADS) h[i] = 0;
    __syncthreads();
    const int n = blockIdx.y, chunk = blockIdx.x;
    const float off = offset[n];
    const int64_t p0 = (int64_t)chunk * pl.chunk;
    for (int i = threadIdx.x; i < pl.chunk; i += P2_THREADS) {
        const int64_t p = p0 + i;
        if (p < d.P) {
            const int64_t s = (int64_t)n * d.P + p;
            const float2 g = *reinterpret_cast<const float2 *>(grid + s * 2);
            const Geo2 q = locate(g.x, g.y, d, f, off, 1);
            uint32_t r = NONE;
            if (q.valid) {
                const int b = q.uy * pl.bx + q.ux;
                const uint32_t old = atomicAdd(&h[b >> 1], (b & 1) ? 0x10000u : 1u);
                r = (b & 1) ? (old >> 16) : (old & 0xffffu);
            }
            pl.rank[s] = r;
        }
    }
    __syncthreads();
    uint32_t *dst = pl.cnt + ((int64_t)n * pl.chunks + chunk) * pl.words;
    for (int i = threadIdx.x; i < pl.words; i += P2_THREADS) dst[i] = h[i];
}
// one thread per (n, cell): bucket size = sum over chunks
__global__ __launch_bounds__(256) void p2_totals(Plan2 pl, int N, uint32_t *__restrict__ totals) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)N * pl.bins) return;
    const int n = (int)(t / pl.bins), b = (int)(t - (int64_t)n * pl.bins);
    const uint32_t *c = pl.cnt + (int64_t)n * pl.chunks * pl.words;
    uint32_t sum = 0;
    for (int k = 0; k < pl.chunks; ++k) sum += cnt_get(c + (int64_t)k * pl.words, b);
    totals[t] = sum;
}
// one thread per (n, cell): first position of every (chunk, cell) = cell_begin + counts of the earlier chunks
__global__ __launch_bounds__(256) void p2_excl(Plan2 pl, int N) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)N * pl.bins) return;
    const int n = (int)(t / pl.bins), b = (int)(t - (int64_t)n * pl.bins);
    const uint32_t *c = pl.cnt + (int64_t)n * pl.chunks * pl.words;
    uint32_t *e = pl.excl + (int64_t)n * pl.chunks * pl.bins + b;
    uint32_t run = pl.cell_begin[t];
    for (int k = 0; k < pl.chunks; ++k) {
        e[(int64_t)k * pl.bins] = run;
        run += cnt_get(c + (int64_t)k * pl.words, b);
    }
}
// p-order: rank[s] = first position of the sample's (chunk, cell) + its rank inside; the coordinates go to their
// sorted slot (the one scattered write of the plan: 8 bytes per sample)
__global__ __launch_bounds__(256) void p2_rank(const float *__restrict__ grid, const float *__restrict__ offset, Plan2 pl,
                                               Dims d, Flags f) {
    const int n = blockIdx.y;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= d.P) return;
    const int64_t s = (int64_t)n * d.P + p;
    const uint32_t r = pl.rank[s];
    if (r == NONE) return;
    const float2 g = *reinterpret_cast<const float2 *>(grid + s * 2);
    const Geo2 q = locate(g.x, g.y, d, f, offset[n], 1);
    const int b = q.uy * pl.bx + q.ux;
    const uint32_t pos = pl.excl[((int64_t)n * pl.chunks + (int)(p / pl.chunk)) * pl.bins + b] + r;
    pl.rank[s] = pos;
    pl.coord[pos] = g;
}

// ------------------------------------------------------------------------------------------------
// point kernels: shared pieces
// ------------------------------------------------------------------------------------------------
// CQ consecutive points of one channel plane, starting at p (a multiple of CQ inside the plane): one vector access when
// the plane allows it (16-byte aligned run, all CQ points exist), scalars at the tail
template <int CQ>
__device__ __forceinline__ void load_run(const float *p, int nlive, float (&v)[CQ]) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef float v2f __attribute__((ext_vector_type(2)));
    constexpr int VB = (CQ >= 4 ? 16 : CQ * 4);
    if (nlive >= CQ && ((uintptr_t)p & (VB - 1)) == 0) {
        if (CQ >= 4) {
#pragma unroll
            for (int k = 0; k < CQ / 4; ++k) {
                const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p) + k);
                v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w;
            }
        } else if (CQ == 2) {
            const v2f t = __builtin_nontemporal_load(reinterpret_cast<const v2f *>(p));
            v[0] = t.x; v[CQ - 1] = t.y;
        } else {
            v[0] = __builtin_nontemporal_load(p);
        }
    } else {
#pragma unroll
        for (int k = 0; k < CQ; ++k) v[k] = k < nlive ? __builtin_nontemporal_load(p + k) : 0.0f;
    }
}
// the 4 channel planes 4q..4q+3 of a stream, CQ points each; planes >= C do not exist (C < 4 runs zero-padded)
template <int CQ>
__device__ __forceinline__ void load_quads(const float *base, int64_t P, int q, int C, int nlive, float (&g)[4][CQ]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (4 * q + j < C) {
            load_run<CQ>(base + (int64_t)(4 * q + j) * P, nlive, g[j]);
        } else {
#pragma unroll
            for (int k = 0; k < CQ; ++k) g[j][k] = 0.0f;
        }
    }
}
template <int CQ>
__device__ __forceinline__ void store_quads(float *base, int64_t P, int q, int C, int nlive, const float4 (&r)[CQ]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (4 * q + j < C) {
            float v[CQ];
#pragma unroll
            for (int k = 0; k < CQ; ++k) v[k] = j == 0 ? r[k].x : j == 1 ? r[k].y : j == 2 ? r[k].z : r[k].w;
            store_run<CQ>(base + (int64_t)(4 * q + j) * P, v, nlive);
        }
    }
}
template <int CQ>
__device__ __forceinline__ float4 quad_of(const float (&g)[4][CQ], int sub) {
    return make_float4(g[0][sub], g[1][sub], g[2][sub], g[3][sub]);
}
// a 16-byte piece of a sorted channels-last row
__device__ __forceinline__ void put_piece(float *rows, uint32_t rank, int C, int q, float4 v) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    if (rank != NONE) {
        const v4f t = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(t, reinterpret_cast<v4f *>(rows + (int64_t)rank * C + 4 * q));
    }
}
// the gather passes: `issue(sub, buf)` starts the DMA loads of pass sub into landing zone buf (LPP loads per lane),
// `use(sub, buf)` consumes them; NBUF zones in flight.  vmcnt counts in issue order, so "all but the youngest
// LPP*(passes still in flight)" is exactly "pass sub has landed" -- as long as `use` issues no vector-memory operation.
template <int CQ, int NBUF, int LPP, typename Issue, typename Use>
__device__ __forceinline__ void dma_passes(Issue issue, Use use) {
#pragma unroll
    for (int sub = 0; sub < NBUF - 1 && sub < CQ; ++sub) issue(sub, sub % NBUF);
#pragma unroll
    for (int sub = 0; sub < CQ; ++sub) {
        if (sub + NBUF - 1 < CQ) issue(sub + NBUF - 1, (sub + NBUF - 1) % NBUF);
        const int ahead = (sub + NBUF - 1 < CQ) ? NBUF - 1 : CQ - 1 - sub;   // passes issued after this one
        if (ahead * LPP == 0) dma_wait_keep<0>();
        else if (ahead * LPP == 4) dma_wait_keep<4>();
        else if (ahead * LPP == 8) dma_wait_keep<8>();
        else if (ahead * LPP == 12) dma_wait_keep<12>();
        else if (ahead * LPP == 16) dma_wait_keep<16>();
        else dma_wait_keep<24>();
        use(sub, sub % NBUF);
    }
}
__device__ __forceinline__ void rec_nodes(const float *rec, int s, uint32_t (&node)[4]) {
    const uint32_t *ru = reinterpret_cast<const uint32_t *>(rec);
#pragma unroll
    for (int a = 0; a < 4; ++a) node[a] = ru[a * 64 + s];
}

// ------------------------------------------------------------------------------------------------
// first backward.  rec per wave: node[4], wx0 wx1 wy0 wy1, rank  (9 x 64 words)
//   grad_grid[s,j] = d1_j * sum_c gOut[c] * (signed node sums)          (2d.cu:476-503)
//   WANT_G: the gOut rows go to their sorted slots (Plan2::G) for the tile kernels of this and the later stages
// ------------------------------------------------------------------------------------------------
constexpr int BW_FIELDS = 9;
template <int CQ, int NBUF>
constexpr size_t bwd_lds() { return (size_t)4 * (BW_FIELDS * 64 + NBUF * DMA_FLOATS) * 4; }
template <int KERNEL, int CQ, int NBUF, bool WANT_G>
__global__ __launch_bounds__(256) void point_bwd_s(const float *__restrict__ gOut, const float *__restrict__ icl,
                                                   const float *__restrict__ grid, const float *__restrict__ offset,
                                                   const uint32_t *__restrict__ rank, float *__restrict__ Gs,
                                                   float *__restrict__ grad_grid, Dims d, Flags f) {
    constexpr int C = 4 * CQ;
    extern __shared__ float lds[];
    float *rec = lds + (threadIdx.x >> 6) * (BW_FIELDS * 64 + NBUF * DMA_FLOATS);
    float *dma = rec + BW_FIELDS * 64;
    const int lane = threadIdx.x & 63, sl = lane / CQ, q = lane % CQ, n = blockIdx.y;
    const int64_t pw = (int64_t)blockIdx.x * 256 + (threadIdx.x & ~63);   // first point of this wave
    const int64_t p0 = pw + CQ * sl;                                        // first of this lane's CQ points
    const int nlive = (int)max((int64_t)0, min((int64_t)CQ, d.P - p0));
    float g[4][CQ];
    load_quads<CQ>(gOut + (int64_t)n * d.go_ns + min(p0, d.P - 1), d.P, q, d.C, nlive, g);
    Sample2 sm;
    sm.load<KERNEL, 1>(grid, offset, d, f);
    {
        q_put_nodes(rec, sm);
        rec[4 * 64 + lane] = sm.ax[0].w[0];
        rec[5 * 64 + lane] = sm.ax[0].w[1];
        rec[6 * 64 + lane] = sm.ax[1].w[0];
        rec[7 * 64 + lane] = sm.ax[1].w[1];
        if (WANT_G) reinterpret_cast<uint32_t *>(rec)[8 * 64 + lane] = sm.live ? rank[sm.s] : NONE;
    }
    __syncthreads();
    const float4 *tab = reinterpret_cast<const float4 *>(icl + (int64_t)n * d.vol * C);
    float mgx = 0.f, mgy = 0.f;
    dma_passes<CQ, NBUF, 4>(
        [&](int sub, int buf) {
            uint32_t node[4];
            rec_nodes(rec, CQ * sl + sub, node);
            dma_issue<CQ>(tab, node, q, dma + buf * DMA_FLOATS);
        },
        [&](int sub, int buf) {
            const int s = CQ * sl + sub;
            const float wx0 = rec[4 * 64 + s], wx1 = rec[5 * 64 + s], wy0 = rec[6 * 64 + s], wy1 = rec[7 * 64 + s];
            const float4 g4 = quad_of<CQ>(g, sub);
            uint32_t node[4];
            rec_nodes(rec, s, node);
            float4 v[4];
            dma_read4(dma + buf * DMA_FLOATS, node, v);
            const float d0 = dot4(v[0], g4), d1 = dot4(v[1], g4), d2 = dot4(v[2], g4), d3 = dot4(v[3], g4);
            const float gx = q_reduce<CQ>(wy0 * (d1 - d0) + wy1 * (d3 - d2));
            const float gy = q_reduce<CQ>(wx0 * (d2 - d0) + wx1 * (d3 - d1));
            if (q == sub) { mgx = gx; mgy = gy; }   // lane (sl, q) keeps point CQ*sl + q = its own lane index
        });
    if (sm.live) *reinterpret_cast<float2 *>(grad_grid + sm.s * 2) = make_float2(sm.ax[0].d1 * mgx, sm.ax[1].d1 * mgy);
    if (WANT_G) {
        const uint32_t *ru = reinterpret_cast<const uint32_t *>(rec);
#pragma unroll
        for (int sub = 0; sub < CQ; ++sub) put_piece(Gs, ru[8 * 64 + CQ * sl + sub], C, q, quad_of<CQ>(g, sub));
    }
}

// ------------------------------------------------------------------------------------------------
// second backward.  rec per wave: node[4], Sx[4], Sy[4], Dm[4], (W[4] for HAS_CI), rank
//   ggOut[c] = sum_a input[q_a] D_a (+ sum_a gOutInput[q_a] W_a);  gGrid[j] = sum_c gOut[c] sum_a input[q_a] S_j[a]
//   (2d.cu:691-706).  The sorted copy of grad_out_grid (Plan2::cG) is written here for the tile kernel; the gOut rows
//   only when no earlier stage of the step left them (WANT_G).
// ------------------------------------------------------------------------------------------------
template <bool HAS_CI>
constexpr int bb_fields() { return 4 + 12 + (HAS_CI ? 4 : 0) + 1; }
template <int CQ, int NBUF, bool HAS_CI>
constexpr size_t bb_lds() { return (size_t)4 * (bb_fields<HAS_CI>() * 64 + NBUF * (HAS_CI ? 2 : 1) * DMA_FLOATS) * 4; }
template <int KERNEL, int CQ, int NBUF, bool HAS_CI, bool SCATTER>
__global__ __launch_bounds__(256, HAS_CI ? 2 : 4) void point_bb_s(const float *__restrict__ cIcl, const float *__restrict__ cG,
                                                  const float *__restrict__ icl, const float *__restrict__ grid,
                                                  const float *__restrict__ gOut, const float *__restrict__ offset,
                                                  const uint32_t *__restrict__ rank, float *__restrict__ Gs,
                                                  float2 *__restrict__ cGs, float *__restrict__ gGrid,
                                                  float *__restrict__ ggOut, Dims d, Flags f, int want_g) {
    constexpr int C = 4 * CQ, F = bb_fields<HAS_CI>(), ZONES = HAS_CI ? 2 : 1, R_RANK = F - 1;
    extern __shared__ float lds[];
    float *rec = lds + (threadIdx.x >> 6) * (F * 64 + NBUF * ZONES * DMA_FLOATS);
    float *dma = rec + F * 64;
    const int lane = threadIdx.x & 63, sl = lane / CQ, q = lane % CQ, n = blockIdx.y;
    const int64_t pw = (int64_t)blockIdx.x * 256 + (threadIdx.x & ~63);
    const int64_t p0 = pw + CQ * sl;
    const int nlive = (int)max((int64_t)0, min((int64_t)CQ, d.P - p0));
    float g[4][CQ];
    load_quads<CQ>(gOut + (int64_t)n * d.go_ns + min(p0, d.P - 1), d.P, q, d.C, nlive, g);
    Sample2 sm;
    sm.load<KERNEL, 2>(grid, offset, d, f);
    {
        const float2 cg = cG ? *reinterpret_cast<const float2 *>(cG + sm.s * 2) : make_float2(0.f, 0.f);
        q_put_nodes(rec, sm);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float sx = sm.pure2(a, 0) * cg.x, sy = sm.pure2(a, 1) * cg.y;   // 2D keeps pure second derivatives only
            if (f.exact) {                                                    // (2d.cu:705-706) unless asked otherwise
                const float mx = sm.mixed2(a);
                sx = fmaf(mx, cg.y, sx);
                sy = fmaf(mx, cg.x, sy);
            }
            rec[(4 + a) * 64 + lane] = sx;
            rec[(8 + a) * 64 + lane] = sy;
            rec[(12 + a) * 64 + lane] = sm.first(a, 0) * cg.x + sm.first(a, 1) * cg.y;
            if (HAS_CI) rec[(16 + a) * 64 + lane] = sm.W[a];
        }
        if (SCATTER) {
            const uint32_t r = sm.live ? rank[sm.s] : NONE;
            reinterpret_cast<uint32_t *>(rec)[R_RANK * 64 + lane] = r;
            if (r != NONE) cGs[r] = cg;
        }
    }
    __syncthreads();
    const float4 *tab = reinterpret_cast<const float4 *>(icl + (int64_t)n * d.vol * C);
    const float4 *ctab = HAS_CI ? reinterpret_cast<const float4 *>(cIcl + (int64_t)n * d.vol * C) : nullptr;
    float msx = 0.f, msy = 0.f;
    float4 res[CQ];
    dma_passes<CQ, NBUF, 4 * ZONES>(
        [&](int sub, int buf) {
            uint32_t node[4];
            rec_nodes(rec, CQ * sl + sub, node);
            dma_issue<CQ>(tab, node, q, dma + buf * ZONES * DMA_FLOATS);
            if (HAS_CI) dma_issue<CQ>(ctab, node, q, dma + (buf * ZONES + 1) * DMA_FLOATS);
        },
        [&](int sub, int buf) {
            const int s = CQ * sl + sub;
            const float *z = dma + buf * ZONES * DMA_FLOATS;
            const float4 g4 = quad_of<CQ>(g, sub);
            uint32_t node[4];
            rec_nodes(rec, s, node);
            float4 acc = zero4(), v[4];
            dma_read4(z, node, v);
            // sum_c gOut[c] sum_a S[a] input[q_a][c] = sum_a S[a] <input[q_a], gOut>: four dot products serve both axes
            float sx = 0.f, sy = 0.f;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                acc = fma4(rec[(12 + a) * 64 + s], v[a], acc);
                const float dd = dot4(v[a], g4);
                sx = fmaf(rec[(4 + a) * 64 + s], dd, sx);
                sy = fmaf(rec[(8 + a) * 64 + s], dd, sy);
            }
            if (HAS_CI) {   // + sum_a gOutInput[q_a] * W_a   (2d.cu:694-697)
                float4 u[4];
                dma_read4(z + DMA_FLOATS, node, u);
#pragma unroll
                for (int a = 0; a < 4; ++a) acc = fma4(rec[(16 + a) * 64 + s], u[a], acc);
            }
            sx = q_reduce<CQ>(sx);
            sy = q_reduce<CQ>(sy);
            if (q == sub) { msx = sx; msy = sy; }
            res[sub] = acc;
        });
    if (sm.live) *reinterpret_cast<float2 *>(gGrid + sm.s * 2) = make_float2(msx, msy);
    if (nlive > 0) store_quads<CQ>(ggOut + (int64_t)n * d.C * d.P + p0, d.P, q, d.C, nlive, res);
    if (SCATTER && want_g) {
        const uint32_t *ru = reinterpret_cast<const uint32_t *>(rec);
#pragma unroll
        for (int sub = 0; sub < CQ; ++sub) put_piece(Gs, ru[R_RANK * 64 + CQ * sl + sub], C, q, quad_of<CQ>(g, sub));
    }
}

// ------------------------------------------------------------------------------------------------
// fused third backward.  rec per wave: node[4], E[4], rank
//   grad_grad_out[c] = sum_a input[q_a] E_a   (2d.cu:876-882);  the D_a * grad_out_ggout part of grad_input
//   (modules_2d.py:109-111) needs no gather: the hO rows and the grid cotangents just travel to their sorted slots.
// ------------------------------------------------------------------------------------------------
constexpr int B3_FIELDS = 9;
template <int CQ, int NBUF>
constexpr size_t bbb_lds() { return (size_t)4 * (B3_FIELDS * 64 + NBUF * DMA_FLOATS) * 4; }
template <int KERNEL, int CQ, int NBUF, bool TWO>
__global__ __launch_bounds__(256, 4) void point_bbb_s(const float *__restrict__ icl, const float *__restrict__ grid,
                                                   const float *__restrict__ gOut, const float *__restrict__ cG,
                                                   const float *__restrict__ hG, const float *__restrict__ hO,
                                                   const float *__restrict__ offset, const uint32_t *__restrict__ rank,
                                                   float *__restrict__ Gs, float2 *__restrict__ cGs,
                                                   float *__restrict__ hOs, float2 *__restrict__ hGs,
                                                   float *__restrict__ ggOut, Dims d, Flags f, int want_g, int want_cg) {
    constexpr int C = 4 * CQ;
    extern __shared__ float lds[];
    float *rec = lds + (threadIdx.x >> 6) * (B3_FIELDS * 64 + NBUF * DMA_FLOATS);
    float *dma = rec + B3_FIELDS * 64;
    const int lane = threadIdx.x & 63, sl = lane / CQ, q = lane % CQ, n = blockIdx.y;
    const int64_t pw = (int64_t)blockIdx.x * 256 + (threadIdx.x & ~63);
    const int64_t p0 = pw + CQ * sl;
    const int nlive = (int)max((int64_t)0, min((int64_t)CQ, d.P - p0));
    float h[4][CQ];
    if (TWO) load_quads<CQ>(hO + (int64_t)n * d.ho_ns + min(p0, d.P - 1), d.P, q, d.C, nlive, h);
    Sample2 sm;
    sm.load<KERNEL, 2>(grid, offset, d, f);
    {
        const float2 cg = cG ? *reinterpret_cast<const float2 *>(cG + sm.s * 2) : make_float2(0.f, 0.f);
        const float2 hg = hG ? *reinterpret_cast<const float2 *>(hG + sm.s * 2) : make_float2(0.f, 0.f);
        q_put_nodes(rec, sm);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float E = sm.pure2(a, 0) * (hg.x * cg.x) + sm.pure2(a, 1) * (hg.y * cg.y);   // 2d.cu:876
            if (f.exact) E = fmaf(sm.mixed2(a), hg.x * cg.y + hg.y * cg.x, E);
            rec[(4 + a) * 64 + lane] = E;
        }
        const uint32_t r = sm.live ? rank[sm.s] : NONE;
        reinterpret_cast<uint32_t *>(rec)[8 * 64 + lane] = r;
        if (r != NONE) {
            hGs[r] = hg;
            if (want_cg) cGs[r] = cg;
        }
    }
    __syncthreads();
    const float4 *tab = reinterpret_cast<const float4 *>(icl + (int64_t)n * d.vol * C);
    float4 res[CQ];
    dma_passes<CQ, NBUF, 4>(
        [&](int sub, int buf) {
            uint32_t node[4];
            rec_nodes(rec, CQ * sl + sub, node);
            dma_issue<CQ>(tab, node, q, dma + buf * DMA_FLOATS);
        },
        [&](int sub, int buf) {
            const int s = CQ * sl + sub;
            const float *z = dma + buf * DMA_FLOATS;
            uint32_t node[4];
            rec_nodes(rec, s, node);
            float4 acc = zero4(), v[4];
            dma_read4(z, node, v);
#pragma unroll
            for (int a = 0; a < 4; ++a) acc = fma4(rec[(4 + a) * 64 + s], v[a], acc);
            res[sub] = acc;
        });
    if (nlive > 0) store_quads<CQ>(ggOut + (int64_t)n * d.C * d.P + p0, d.P, q, d.C, nlive, res);
    const uint32_t *ru = reinterpret_cast<const uint32_t *>(rec);
    if (TWO) {
#pragma unroll
        for (int sub = 0; sub < CQ; ++sub) put_piece(hOs, ru[8 * 64 + CQ * sl + sub], C, q, quad_of<CQ>(h, sub));
    }
    if (want_g) {   // no earlier stage of this step sorted grad_output: do it here (one more 1 GiB read)
        float g[4][CQ];
        load_quads<CQ>(gOut + (int64_t)n * d.go_ns + min(p0, d.P - 1), d.P, q, d.C, nlive, g);
#pragma unroll
        for (int sub = 0; sub < CQ; ++sub) put_piece(Gs, ru[8 * 64 + CQ * sl + sub], C, q, quad_of<CQ>(g, sub));
    }
}

// ------------------------------------------------------------------------------------------------
// tile kernels: grad_input[n,c,node] += sum over the tile's samples of coef_a * payload[c], payload and coordinates
// streamed in cell order.  One workgroup per (n, 16x16-cell tile); CQ lanes = one walker (lane q owns channels
// 4q..4q+3) of CQ cells of one cell row = ONE contiguous run of the sorted arrays; node sums in registers, the
// right-hand pair handed to the next cell, finished nodes to LDS without atomics, the tile's 17x17 nodes added to the
// caller's NCHW grad_input with lanes along x (round 1's tile_scatter, minus the ids and the row fetch).
//   MODE 0  first backward:   W_a * gOut
//   MODE 1  second backward:  D_a * gOut,            D_a = sum_j dW_a/dg_j cG_j                      (2d.cu:709)
//   MODE 2  fused third:      E_a * gOut + D_a * hO   (2d.cu:885; modules_2d.py:109-111)
//   MODE 3  third without grad_out_ggout: E_a * gOut
// ------------------------------------------------------------------------------------------------
template <int CQ>
constexpr size_t tile_s_lds() { return (size_t)2 * TY * (TX / CQ) * (CQ + 1) * CQ * 16 + (size_t)TY * (TX + 1) * 4; }
template <int KERNEL, int CQ, int MODE>
__global__ __launch_bounds__(256) void tile_s(Plan2 pl, const float *__restrict__ hOs, const float2 *__restrict__ hGs,
                                              const float *__restrict__ offset, float *__restrict__ grad_input, Dims d,
                                              Flags f) {
    constexpr int C = 4 * CQ;
    constexpr int SEGW = CQ, NSEG = TX / SEGW, NODES = NSEG * (SEGW + 1);
    constexpr int U = (MODE == 2) ? 4 : 8;            // samples in flight per walker
    constexpr int ORDER = MODE == 0 ? 0 : MODE == 1 ? 1 : 2;
    static_assert(TY * NSEG * CQ == 256, "one workgroup = all walkers of a tile");
    extern __shared__ float4 tile_lds[];
    float4 *top = tile_lds;
    float4 *bot = top + TY * NODES * CQ;
    uint32_t *cb = reinterpret_cast<uint32_t *>(bot + TY * NODES * CQ);   // [TY][TX+1] first sorted position of every cell

    const int ntl = pl.ntx * pl.nty;
    const int n = blockIdx.x / ntl, tl = blockIdx.x - n * ntl;
    const int ty = tl / pl.ntx, tx = tl - ty * pl.ntx;
    const uint32_t *cbn = pl.cell_begin + (int64_t)n * pl.bins;
    for (int i = threadIdx.x; i < TY * (TX + 1); i += 256) {
        const int ly = i / (TX + 1), lx = i - ly * (TX + 1);
        const int uy = ty * TY + ly, ux = min(tx * TX + lx, pl.bx);   // one past the row's end = first cell of the next row
        cb[i] = uy < pl.by ? cbn[(int64_t)uy * pl.bx + ux] : 0u;
    }
    __syncthreads();
    {   // empty tile: grad_input was zero-filled
        bool any = false;
        for (int ly = 0; ly < TY; ++ly) any |= cb[ly * (TX + 1)] != cb[ly * (TX + 1) + TX];
        if (!any) return;
    }
    const float off = offset[n];
    const int w = threadIdx.x / CQ, q = threadIdx.x % CQ;
    const int ly = w / NSEG, seg = w % NSEG;
    {
        float4 ct = zero4(), cbm = zero4();
        float4 a0 = zero4(), a1 = zero4(), a2 = zero4(), a3 = zero4();
        int cur = 0;
        float4 *trow = top + ((ly * NSEG + seg) * (SEGW + 1)) * CQ + q;
        float4 *brow = bot + ((ly * NSEG + seg) * (SEGW + 1)) * CQ + q;
        const uint32_t *cbr = cb + ly * (TX + 1) + seg * SEGW;
        const uint32_t j0 = cbr[0], j1 = cbr[SEGW];
        uint32_t nb = cbr[1];
        for (uint32_t j = j0; j < j1; j += U) {
            float4 g[U], h[U];
            float2 xy[U], cg[U], hg[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {       // all loads of the batch first: consecutive sorted positions
                const uint32_t jj = min(j + u, j1 - 1);
                g[u] = ld_row(pl.G + (int64_t)jj * C + 4 * q);
                xy[u] = pl.coord[jj];
                if (MODE >= 1) cg[u] = pl.cG[jj];
                if (MODE >= 2) hg[u] = hGs[jj];
                if (MODE == 2) h[u] = ld_row(hOs + (int64_t)jj * C + 4 * q);
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
                    Sample2 sm;                  // only the two axes are used
                    sm.ax[0] = make_axis<KERNEL, ORDER>(xy[u].x, d.size[0], f, f.align, off);
                    sm.ax[1] = make_axis<KERNEL, ORDER>(xy[u].y, d.size[1], f, f.align, off);
                    float k[4], k2[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        if (MODE == 0) {
                            k[a] = sm.ax[0].w[a & 1] * sm.ax[1].w[a >> 1];
                        } else if (MODE == 1) {
                            k[a] = sm.first(a, 0) * cg[u].x + sm.first(a, 1) * cg[u].y;
                        } else {
                            float E = sm.pure2(a, 0) * (hg[u].x * cg[u].x) + sm.pure2(a, 1) * (hg[u].y * cg[u].y);
                            if (f.exact) E = fmaf(sm.mixed2(a), hg[u].x * cg[u].y + hg[u].y * cg[u].x, E);
                            k[a] = E;
                            if (MODE == 2) k2[a] = sm.first(a, 0) * cg[u].x + sm.first(a, 1) * cg[u].y;
                        }
                    }
                    a0 = fma4(k[0], g[u], a0); a1 = fma4(k[1], g[u], a1);
                    a2 = fma4(k[2], g[u], a2); a3 = fma4(k[3], g[u], a3);
                    if (MODE == 2) {
                        a0 = fma4(k2[0], h[u], a0); a1 = fma4(k2[1], h[u], a1);
                        a2 = fma4(k2[2], h[u], a2); a3 = fma4(k2[3], h[u], a3);
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

    // node (lyy, lx) of the tile = global node (ty*TY + lyy - 1, tx*TX + lx - 1).  Column lx is slot lx % SEGW of run
    // lx / SEGW and, when lx is a run boundary, also the last slot of the run before.
    const int W = d.size[0], H = d.size[1];
    const float *topf = reinterpret_cast<const float *>(top), *botf = reinterpret_cast<const float *>(bot);
    float *gi = grad_input + (int64_t)n * d.C * d.vol;
    for (int idx = threadIdx.x; idx < d.C * (TY + 1) * (TX + 1); idx += 256) {   // ch < d.C: padded channels are dropped
        const int lx = idx % (TX + 1);
        const int rest = idx / (TX + 1);
        const int lyy = rest % (TY + 1);
        const int ch = rest / (TY + 1);
        const int gx = tx * TX + lx - 1, gy = ty * TY + lyy - 1;
        if (gx < 0 || gx >= W || gy < 0 || gy >= H) continue;
        const int sg = lx / SEGW, sl = lx - sg * SEGW;
        float v = 0.f;
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const int s2 = side ? sg - 1 : sg, l2 = side ? SEGW : sl;
            if (side && sl != 0) continue;
            if (s2 < 0 || s2 >= NSEG) continue;
            if (lyy < TY) v += topf[((lyy * NSEG + s2) * (SEGW + 1) + l2) * C + ch];
            if (lyy > 0) v += botf[(((lyy - 1) * NSEG + s2) * (SEGW + 1) + l2) * C + ch];
        }
        if (v != 0.f) unsafeAtomicAdd(gi + (int64_t)ch * d.vol + (int64_t)gy * W + gx, v);
    }
}

}  // namespace sorted
}  // namespace cs
