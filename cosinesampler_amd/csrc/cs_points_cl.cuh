// cs_points_cl.cuh -- point kernels for any dimensionality gathering float4 channel quads from the
// channels-last copy of `input` (one node = one contiguous C-float row).  Used for 3D with C <= 16 (padded to 4, 8 or 16 channels): a
// trilinear sample touches 8 node rows instead of 8*C separate lines of the NCDHW tensor.  They produce the
// p-ordered outputs of a stage and, with SCATTER, the input-shaped gradient too: each lane leaves
// [cotangent values | node coefficients | node ids] of its sample in LDS and the wave then adds whole node rows
// (x-neighbour pairs when a row is under 64 bytes) into the channels-last accumulator with C or 2C lanes per
// sample -- the row atomics of row_scatter (cs_kernels_direct.cuh) without a second kernel that recomputes the
// geometry and re-reads the cotangents channel by channel.  Formulas as in cs_kernels_direct.cuh.
#pragma once
#include "cs_kernels_direct.cuh"

namespace cs {
namespace cl {

// waves per SIMD the backward point kernels are compiled for (A/B builds).  They hold 144-184 registers (2-3 waves per
// SIMD); asking the compiler for more waves only makes it spill: BASELINE configs[3] 3.22 ms as is, 3.31 / 3.99 / 5.26 at
// 3 / 4 / 5 waves per SIMD (profiles/round4_ablation.txt) -- more occupancy has to come from fewer live values per lane.
#ifndef CS_CL_WAVES
#define CS_CL_WAVES 1
#endif

__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 fma4(float s, float4 a, float4 acc) {
    return make_float4(fmaf(s, a.x, acc.x), fmaf(s, a.y, acc.y), fmaf(s, a.z, acc.z), fmaf(s, a.w, acc.w));
}
__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
// the 4 channels of a quad of a channel-major stream; cv = how many of them exist (C = 1..3 runs zero-padded)
template <typename T>
__device__ __forceinline__ float4 load_quad(const T *src, int64_t P, int cv) {
    float4 r;
    if constexpr (sizeof(T) == 2) {   // 16-bit: the four loads are issued before the first conversion -- behind a branch
        if (cv <= 0) return zero4();  // each one would be waited for on its own (cs_tiled.cuh StreamRegs)
        const int last = cv > 3 ? 3 : cv - 1;
        T t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = __builtin_nontemporal_load(src + (int64_t)(k < last ? k : last) * P);
        return make_float4((float)t[0], cv > 1 ? (float)t[1] : 0.0f, cv > 2 ? (float)t[2] : 0.0f, cv > 3 ? (float)t[3] : 0.0f);
    }
    r.x = cv > 0 ? stream_load(src) : 0.0f;   // cv <= 0: a quad of padding channels (C padded up to a supported count)
    r.y = cv > 1 ? stream_load(src + P) : 0.0f;
    r.z = cv > 2 ? stream_load(src + 2 * P) : 0.0f;
    r.w = cv > 3 ? stream_load(src + 3 * P) : 0.0f;
    return r;
}
template <typename T>
__device__ __forceinline__ void store_quad(T *dst, int64_t P, float4 o, int cv) {
    if (cv > 0) stream_store(dst, o.x);
    if (cv > 1) stream_store(dst + P, o.y);
    if (cv > 2) stream_store(dst + 2 * P, o.z);
    if (cv > 3) stream_store(dst + 3 * P, o.w);
}

// node rows of one channel quad; zero-padded nodes read row 0 and are masked.
// 3D tables are z-paired (pack_cl4: node v holds [its own row | the row of the node one z-plane above]): a corner on the
// sample's upper z-plane is read from the second slot of the node below it whenever that node's plane exists, so that the
// four rows (x, x+1) x (z, z+1) of one y are one contiguous run of 4 * C floats.  `plane` = H * W.
template <int DIM, int CQ>
__device__ __forceinline__ void gather_quad(const float4 *tab, const Sample<DIM> &sm, int q, float4 (&v)[1 << DIM],
                                            int64_t plane = 0) {
#pragma unroll
    for (int a = 0; a < (1 << DIM); ++a) {
        int64_t row = sm.node[a] < 0 ? 0 : sm.node[a];
        if (DIM == 3) row = ((a & 4) && sm.ax[DIM - 1].lo >= 0 && sm.node[a] >= 0) ? (row - plane) * 2 + 1 : row * 2;
        v[a] = tab[row * CQ + q];
    }
#pragma unroll
    for (int a = 0; a < (1 << DIM); ++a)
        if (sm.node[a] < 0) v[a] = zero4();
}

// ---- fused row-atomic scatter -------------------------------------------------------------------
// MODE 0: W_a * gOut   MODE 1: D_a * gOut   MODE 2: E_a * gOut + D_a * hO     (as row_scatter)
template <int DIM, int CQ, int MODE>
struct Rec {
    static constexpr int NC = 1 << DIM, C = 4 * CQ, NP = MODE == 2 ? 2 : 1;
    static constexpr int COEF = C * NP, IDS = COEF + NC * NP, WORDS = IDS + NC;   // [payloads | coefficients | node ids]
};
template <int DIM, int CQ, int MODE>
__device__ __forceinline__ void rec_put_nodes(float *rec, const Sample<DIM> &sm, const Dims &d, bool live) {
    int *ids = reinterpret_cast<int *>(rec + Rec<DIM, CQ, MODE>::IDS);
#pragma unroll
    for (int a = 0; a < (1 << DIM); ++a)
        ids[a] = (live && sm.node[a] >= 0) ? (int)((int64_t)sm.n * d.vol + sm.node[a]) : -1;
}
template <int DIM, int CQ, int MODE>
__device__ __forceinline__ void scatter_phase(const float *stage, float *__restrict__ acc_cl) {
    using R = Rec<DIM, CQ, MODE>;
    constexpr int NC = R::NC, C = R::C;
    constexpr bool PAIR = C <= 8;                 // rows under 64 bytes go as x-neighbour pairs
    constexpr int L = PAIR ? 2 * C : C;           // lanes per sample
    const int lane = threadIdx.x & 63, sub = lane % L, c = sub % C, xbit = PAIR ? sub / C : 0;
#pragma unroll 2
    for (int pass = 0; pass < L; ++pass) {        // 64 / L samples per pass
        const float *r = stage + (pass * (64 / L) + lane / L) * R::WORDS;
        const float g = r[c];
        const float h = MODE == 2 ? r[C + c] : 0.0f;
        const int *ids = reinterpret_cast<const int *>(r + R::IDS);
#pragma unroll
        for (int k = 0; k < (PAIR ? NC / 2 : NC); ++k) {
            const int a = PAIR ? 2 * k + xbit : k;
            const int node = ids[a];
            if (node < 0) continue;
            float v = r[R::COEF + a] * g;
            if (MODE == 2) v = fmaf(r[R::COEF + NC + a], h, v);
            unsafeAtomicAdd(acc_cl + (int64_t)node * C + c, v);
        }
    }
}

// ---- the summing mode (Dims::nsum; CS_SUM_OVER_N in 3D, round 4) -------------------------------------------------
// The PIXEL pattern sampler(cells, grid.repeat(N,..)).sum(0) (reference test/test_3d.py:31-32): ONE set of points and of
// cotangents for all N tables, per-point results summed over the tables.  A lane owns a point and walks the tables
// (`Walk`): the gathers of a 3D table are random HBM lines whichever table they come from, so walking the tables costs
// nothing in locality, and the (N,C,P) streams of the plain op, their sums over n and the expanded cotangents disappear.
// The records for the scatter stay per (table, point): the scatter kernels are the plain op's.
struct Walk {
    int64_t t;       // the lane's index: a sample (plain) or a point (nsum)
    int iters;       // tables this lane walks: N (nsum) or 1
    __device__ __forceinline__ Walk(const Dims &d) : t((int64_t)blockIdx.x * blockDim.x + threadIdx.x), iters(d.nsum ? d.N : 1) {}
    // the sample of iteration `it` (past the end: no sample)
    __device__ __forceinline__ int64_t sample(const Dims &d, int it) const {
        return d.nsum ? (t < d.P ? (int64_t)it * d.P + t : d.S) : t;
    }
    // first sample of this lane's wave in iteration `it`, and how many of its 64 exist
    __device__ __forceinline__ int64_t wave_first(const Dims &d, int it) const {
        return (d.nsum ? (int64_t)it * d.P : 0) + (int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63);
    }
    __device__ __forceinline__ int wave_live(const Dims &d) const {
        const int64_t lim = d.nsum ? d.P : d.S, w0 = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63);
        return w0 >= lim ? 0 : (int)(lim - w0 < 64 ? lim - w0 : 64);
    }
};

// SCATTER == 2 (crowded tables, cs_dense3d.cuh): instead of adding them, the wave's records -- payloads and
// coefficients, without the node ids -- go to HBM as p-ordered rows of Rec::IDS floats for cell_scatter3.
template <int DIM, int CQ, int MODE>
__device__ __forceinline__ void flush_records(const float *stage, float *__restrict__ rows, const Dims &d, int64_t s0, int nlive) {
    using R = Rec<DIM, CQ, MODE>;
    constexpr int PIECES = R::IDS / 4;   // float4 per row
    const int lane = threadIdx.x & 63;
    if (nlive <= 0) return;
    float4 *dst = reinterpret_cast<float4 *>(rows + s0 * R::IDS);
    for (int item = lane; item < nlive * PIECES; item += 64) {
        const int row = item / PIECES, piece = item - row * PIECES;
        dst[item] = *reinterpret_cast<const float4 *>(stage + row * R::WORDS + 4 * piece);
    }
}

template <int DIM, int KERNEL, int CQ, typename ST = float>
__global__ __launch_bounds__(256) void forward(const float *__restrict__ icl, const float *__restrict__ grid,
                                               const float *__restrict__ offset, ST *__restrict__ out, Dims d,
                                               Flags f) {
    constexpr int NC = 1 << DIM, C = 4 * CQ;
    const Walk wk(d);
    float4 acc[CQ];
#pragma unroll
    for (int q = 0; q < CQ; ++q) acc[q] = zero4();
    int n_out = 0;
    int64_t p_out = -1;
    for (int it = 0; it < wk.iters; ++it) {
        Sample<DIM> sm;
        if (!sm.template load_at<KERNEL, 0>(wk.sample(d, it), grid, offset, d, f, DIM == 2 ? 1 : f.align)) break;
        float W[NC];
        sm.weights(W);
        const int64_t plane = (int64_t)d.size[0] * d.size[1];                 // 3D tables are z-paired: two rows per node
        const float4 *tab = reinterpret_cast<const float4 *>(icl + (int64_t)sm.n * d.vol * C * (DIM == 3 ? 2 : 1));
        float4 v[CQ][NC];
#pragma unroll
        for (int q = 0; q < CQ; ++q) gather_quad<DIM, CQ>(tab, sm, q, v[q], plane);
#pragma unroll
        for (int q = 0; q < CQ; ++q)
#pragma unroll
            for (int a = 0; a < NC; ++a) acc[q] = fma4(W[a], v[q][a], acc[q]);
        n_out = d.nsum ? 0 : sm.n;
        p_out = sm.p;
    }
    if (p_out < 0) return;
    ST *o = out + (int64_t)n_out * d.out_ns + p_out;   // out_ns: d.C * d.P for a contiguous stream (d.C: the caller's channel count, C the padded one)
#pragma unroll
    for (int q = 0; q < CQ; ++q) store_quad(o + (int64_t)(4 * q) * d.P, d.P, acc[q], d.C - 4 * q);
}

template <int DIM, int KERNEL, int CQ, int SCATTER, typename ST = float>
__global__ __launch_bounds__(256, CS_CL_WAVES) void backward(const ST *__restrict__ gOut, const float *__restrict__ icl,
                                                const float *__restrict__ grid, const float *__restrict__ offset,
                                                float *__restrict__ grad_grid, float *__restrict__ acc_cl, Dims d,
                                                Flags f) {
    constexpr int NC = 1 << DIM, C = 4 * CQ;
    using R = Rec<DIM, CQ, 0>;
    extern __shared__ float lds[];
    float *stage = lds + (threadIdx.x >> 6) * (64 * R::WORDS);
    float *rec = stage + (threadIdx.x & 63) * R::WORDS;
    const Walk wk(d);
    float gsum[DIM];
#pragma unroll
    for (int j = 0; j < DIM; ++j) gsum[j] = 0.0f;
    int64_t gg_out = -1;
    for (int it = 0; it < wk.iters; ++it) {
    Sample<DIM> sm;
    const bool live = sm.template load_at<KERNEL, 1>(wk.sample(d, it), grid, offset, d, f, f.align);
    if (!SCATTER && !live) break;
    if (SCATTER) rec_put_nodes<DIM, CQ, 0>(rec, sm, d, live);
    if (live) {
    const int64_t plane = (int64_t)d.size[0] * d.size[1];                 // 3D tables are z-paired: two rows per node
    const float4 *tab = reinterpret_cast<const float4 *>(icl + (int64_t)sm.n * d.vol * C * (DIM == 3 ? 2 : 1));
    const ST *go = gOut + (int64_t)sm.n * d.go_ns + sm.p;
    float4 v[CQ][NC], g[CQ];
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
        g[q] = load_quad(go + (int64_t)(4 * q) * d.P, d.P, d.C - 4 * q);
        gather_quad<DIM, CQ>(tab, sm, q, v[q], plane);
    }
    // every gather in flight before the first one is consumed: left alone, the scheduler issued the second quad's node
    // rows two at a time between the uses of the first (four more HBM round trips per wave; 3D config 3: 0.72 -> 0.54 ms)
    __builtin_amdgcn_sched_barrier(0);
    float oth[DIM][NC];   // (worked out while the rows fly: fewer registers live across the wait)
#pragma unroll
    for (int j = 0; j < DIM; ++j)
#pragma unroll
        for (int a = 0; a < NC; ++a) oth[j][a] = ((a >> j) & 1) ? sm.others(a, j) : -sm.others(a, j);
    float acc[DIM];
#pragma unroll
    for (int j = 0; j < DIM; ++j) acc[j] = 0.0f;
#pragma unroll
    for (int q = 0; q < CQ; ++q)
#pragma unroll
        for (int j = 0; j < DIM; ++j) {
            float4 t = zero4();
#pragma unroll
            for (int a = 0; a < NC; ++a) t = fma4(oth[j][a], v[q][a], t);
            acc[j] += dot4(t, g[q]);
        }
    gg_out = ((int64_t)(d.nsum ? 0 : sm.n) * d.P + sm.p) * DIM;
#pragma unroll
    for (int j = 0; j < DIM; ++j) gsum[j] = it == 0 ? sm.ax[j].d1 * acc[j] : fmaf(sm.ax[j].d1, acc[j], gsum[j]);
    if (SCATTER) {
        float W[NC];
        sm.weights(W);
#pragma unroll
        for (int q = 0; q < CQ; ++q) *reinterpret_cast<float4 *>(rec + 4 * q) = g[q];
#pragma unroll
        for (int a = 0; a < NC; ++a) rec[R::COEF + a] = W[a];
    }
    }
    if (SCATTER) {
        __syncthreads();
        if (SCATTER == 1) scatter_phase<DIM, CQ, 0>(stage, acc_cl);
        else flush_records<DIM, CQ, 0>(stage, acc_cl, d, wk.wave_first(d, it), wk.wave_live(d));
        if (it + 1 < wk.iters) __syncthreads();           // the stage is written again for the next table
    }
    }
    if (gg_out >= 0) {
#pragma unroll
        for (int j = 0; j < DIM; ++j) grad_grid[gg_out + j] = gsum[j];
    }
}

template <int DIM, int KERNEL, int CQ, bool HAS_CI, int SCATTER, typename ST = float>
__global__ __launch_bounds__(256, CS_CL_WAVES) void backward_backward(const float *__restrict__ cIcl, const float *__restrict__ cG,
                                                         const float *__restrict__ icl, const float *__restrict__ grid,
                                                         const ST *__restrict__ gOut, const float *__restrict__ offset,
                                                         float *__restrict__ gGrid, ST *__restrict__ ggOut,
                                                         float *__restrict__ acc_cl, Dims d, Flags f) {
    constexpr int NC = 1 << DIM, C = 4 * CQ;
    constexpr bool FULL = (DIM == 3);   // mixed second derivatives + gOutInput -> grad_grid (3d.cu:758-771, :837-839)
    using R = Rec<DIM, CQ, 1>;
    extern __shared__ float lds[];
    float *stage = lds + (threadIdx.x >> 6) * (64 * R::WORDS);
    float *rec = stage + (threadIdx.x & 63) * R::WORDS;
    const Walk wk(d);
    float gsum[DIM];
    float4 osum[CQ];
#pragma unroll
    for (int j = 0; j < DIM; ++j) gsum[j] = 0.0f;
#pragma unroll
    for (int q = 0; q < CQ; ++q) osum[q] = zero4();
    int n_out = 0;
    int64_t p_out = -1;
    for (int it = 0; it < wk.iters; ++it) {
    Sample<DIM> sm;
    const bool live = sm.template load_at<KERNEL, 2>(wk.sample(d, it), grid, offset, d, f, f.align);
    if (!SCATTER && !live) break;
    if (SCATTER) rec_put_nodes<DIM, CQ, 1>(rec, sm, d, live);
    if (live) {
    // loads first, all of them (see bbb): grad_out_grid through a pointer that is valid either way, the cotangent quads, the
    // node rows; the coefficient tables are worked out while they fly
    const float *cgp = (cG ? cG : grid) + d.gpt(sm.n, sm.p) * DIM;
    float cg[DIM];
#pragma unroll
    for (int j = 0; j < DIM; ++j) cg[j] = cgp[j];
    const int64_t plane = (int64_t)d.size[0] * d.size[1];                 // 3D tables are z-paired: two rows per node
    const float4 *tab = reinterpret_cast<const float4 *>(icl + (int64_t)sm.n * d.vol * C * (DIM == 3 ? 2 : 1));
    const float4 *ctab = reinterpret_cast<const float4 *>(cIcl + (int64_t)sm.n * d.vol * C * (DIM == 3 ? 2 : 1));
    const ST *go = gOut + (int64_t)sm.n * d.go_ns + sm.p;
    float4 gq[CQ], vq[CQ][NC];   // all node rows in flight at once (see backward)
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
        gq[q] = load_quad(go + (int64_t)(4 * q) * d.P, d.P, d.C - 4 * q);
        gather_quad<DIM, CQ>(tab, sm, q, vq[q], plane);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < DIM; ++j) cg[j] = cG ? cg[j] : 0.0f;
    float W[NC], Dm[NC], F[DIM][NC], Sg[DIM][NC];
    sm.weights(W);
#pragma unroll
    for (int a = 0; a < NC; ++a) {
        float dsum = 0.0f;
#pragma unroll
        for (int j = 0; j < DIM; ++j) {
            F[j][a] = sm.first(a, j);
            dsum = fmaf(F[j][a], cg[j], dsum);
            float s = sm.pure2(a, j) * cg[j];
            if (FULL) {
#pragma unroll
                for (int k = 0; k < DIM; ++k)
                    if (k != j) s = fmaf(sm.mixed2(a, j, k), cg[k], s);
            }
            Sg[j][a] = s;
        }
        Dm[a] = dsum;
    }
    float acc[DIM];
#pragma unroll
    for (int j = 0; j < DIM; ++j) acc[j] = 0.0f;
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
        const float4 g = gq[q];
        if (SCATTER) *reinterpret_cast<float4 *>(rec + 4 * q) = g;
        const float4(&v)[NC] = vq[q];
        float4 o = zero4();
#pragma unroll
        for (int a = 0; a < NC; ++a) o = fma4(Dm[a], v[a], o);
#pragma unroll
        for (int j = 0; j < DIM; ++j) {
            float4 t = zero4();
#pragma unroll
            for (int a = 0; a < NC; ++a) t = fma4(Sg[j][a], v[a], t);
            acc[j] += dot4(t, g);
        }
        if (HAS_CI) {
            float4 u[NC];
            gather_quad<DIM, CQ>(ctab, sm, q, u, plane);
#pragma unroll
            for (int a = 0; a < NC; ++a) o = fma4(W[a], u[a], o);
            if (FULL) {
#pragma unroll
                for (int j = 0; j < DIM; ++j) {
                    float4 t = zero4();
#pragma unroll
                    for (int a = 0; a < NC; ++a) t = fma4(F[j][a], u[a], t);
                    acc[j] += dot4(t, g);
                }
            }
        }
        // (the first table's value as it is: the plain op's results stay bit for bit what they were)
        osum[q] = it == 0 ? o : make_float4(osum[q].x + o.x, osum[q].y + o.y, osum[q].z + o.z, osum[q].w + o.w);
    }
    n_out = d.nsum ? 0 : sm.n;
    p_out = sm.p;
#pragma unroll
    for (int j = 0; j < DIM; ++j) gsum[j] = it == 0 ? acc[j] : gsum[j] + acc[j];
    if (SCATTER) {
#pragma unroll
        for (int a = 0; a < NC; ++a) rec[R::COEF + a] = Dm[a];
    }
    }
    if (SCATTER) {
        __syncthreads();
        if (SCATTER == 1) scatter_phase<DIM, CQ, 1>(stage, acc_cl);
        else flush_records<DIM, CQ, 1>(stage, acc_cl, d, wk.wave_first(d, it), wk.wave_live(d));
        if (it + 1 < wk.iters) __syncthreads();           // the stage is written again for the next table
    }
    }
    if (p_out >= 0) {
        ST *ggo = ggOut + (int64_t)n_out * d.out_ns + p_out;
#pragma unroll
        for (int q = 0; q < CQ; ++q) store_quad(ggo + (int64_t)(4 * q) * d.P, d.P, osum[q], d.C - 4 * q);
        float *gg = gGrid + ((int64_t)n_out * d.P + p_out) * DIM;
#pragma unroll
        for (int j = 0; j < DIM; ++j) gg[j] = gsum[j];
    }
}

template <int DIM, int KERNEL, int CQ, int SCATTER, typename ST = float>
__global__ __launch_bounds__(256, CS_CL_WAVES) void bbb(const float *__restrict__ icl, const float *__restrict__ grid,
                                           const ST *__restrict__ gOut, const float *__restrict__ cG,
                                           const float *__restrict__ hG, const ST *__restrict__ hO,
                                           const float *__restrict__ offset, ST *__restrict__ ggOut,
                                           float *__restrict__ acc_cl, Dims d, Flags f) {
    constexpr int NC = 1 << DIM, C = 4 * CQ;
    using R = Rec<DIM, CQ, 2>;
    extern __shared__ float lds[];
    float *stage = lds + (threadIdx.x >> 6) * (64 * R::WORDS);
    float *rec = stage + (threadIdx.x & 63) * R::WORDS;
    const Walk wk(d);
    float4 osum[CQ];
#pragma unroll
    for (int q = 0; q < CQ; ++q) osum[q] = zero4();
    int n_out = 0;
    int64_t p_out = -1;
    for (int it = 0; it < wk.iters; ++it) {
    Sample<DIM> sm;
    const bool live = sm.template load_at<KERNEL, 2>(wk.sample(d, it), grid, offset, d, f, f.align);
    if (!SCATTER && !live) break;
    if (SCATTER) rec_put_nodes<DIM, CQ, 2>(rec, sm, d, live);
    if (live) {
    // every load of the sample goes out before anything is computed from it: the cotangents of the grid (read through a
    // pointer that is valid either way and masked afterwards -- a load inside `p ? *p : 0` is waited for on its own),
    // the stream quads, then the node rows
    const int64_t o0 = d.gpt(sm.n, sm.p) * DIM;
    const float *cgp = cG ? cG + o0 : grid + o0, *hgp = hG ? hG + o0 : grid + o0;
    float cgv[DIM], hgv[DIM];
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        cgv[j] = cgp[j];
        hgv[j] = hgp[j];
    }
    float4 gq[CQ], hq[CQ];
    if (SCATTER) {
        const ST *go = gOut + (int64_t)sm.n * d.go_ns + sm.p;
#pragma unroll
        for (int q = 0; q < CQ; ++q) {
            gq[q] = load_quad(go + (int64_t)(4 * q) * d.P, d.P, d.C - 4 * q);
            hq[q] = zero4();
            if (hO) hq[q] = load_quad(hO + (int64_t)sm.n * d.ho_ns + sm.p + (int64_t)(4 * q) * d.P, d.P, d.C - 4 * q);
        }
    }
    const int64_t plane = (int64_t)d.size[0] * d.size[1];                 // 3D tables are z-paired: two rows per node
    const float4 *tab = reinterpret_cast<const float4 *>(icl + (int64_t)sm.n * d.vol * C * (DIM == 3 ? 2 : 1));
    float4 v[CQ][NC];
#pragma unroll
    for (int q = 0; q < CQ; ++q) gather_quad<DIM, CQ>(tab, sm, q, v[q], plane);
    __builtin_amdgcn_sched_barrier(0);
    float Em[NC], Dm[NC];
#pragma unroll
    for (int a = 0; a < NC; ++a) Em[a] = Dm[a] = 0.0f;
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        const float cgj = cG ? cgv[j] : 0.0f;
        const float hgj = hG ? hgv[j] : 0.0f;
        float e = cgj * hgj;
#pragma unroll
        for (int a = 0; a < NC; ++a) {
            Em[a] = fmaf(sm.pure2(a, j), e, Em[a]);   // pure terms only (3d.cu:1008-1010) ...
            if (SCATTER) Dm[a] = fmaf(sm.first(a, j), cgj, Dm[a]);
        }
        if (f.exact) {                                // ... unless the mixed ones are asked for
#pragma unroll
            for (int k = 0; k < DIM; ++k) {
                if (k == j) continue;
                float ek = hgj * (cG ? cgv[k] : 0.0f);
#pragma unroll
                for (int a = 0; a < NC; ++a) Em[a] = fmaf(sm.mixed2(a, j, k), ek, Em[a]);
            }
        }
    }
    if (SCATTER) {   // cotangent streams and coefficients of the scatter: E_a * gOut + D_a * hO
#pragma unroll
        for (int q = 0; q < CQ; ++q) {
            *reinterpret_cast<float4 *>(rec + 4 * q) = gq[q];
            *reinterpret_cast<float4 *>(rec + C + 4 * q) = hq[q];
        }
#pragma unroll
        for (int a = 0; a < NC; ++a) {
            rec[R::COEF + a] = Em[a];
            rec[R::COEF + NC + a] = Dm[a];
        }
    }
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
        float4 o = zero4();
#pragma unroll
        for (int a = 0; a < NC; ++a) o = fma4(Em[a], v[q][a], o);
        osum[q] = it == 0 ? o : make_float4(osum[q].x + o.x, osum[q].y + o.y, osum[q].z + o.z, osum[q].w + o.w);
    }
    n_out = d.nsum ? 0 : sm.n;
    p_out = sm.p;
    }
    if (SCATTER) {
        __syncthreads();
        if (SCATTER == 1) scatter_phase<DIM, CQ, 2>(stage, acc_cl);
        else flush_records<DIM, CQ, 2>(stage, acc_cl, d, wk.wave_first(d, it), wk.wave_live(d));
        if (it + 1 < wk.iters) __syncthreads();           // the stage is written again for the next table
    }
    }
    if (p_out >= 0) {
        ST *ggo = ggOut + (int64_t)n_out * d.out_ns + p_out;
#pragma unroll
        for (int q = 0; q < CQ; ++q) store_quad(ggo + (int64_t)(4 * q) * d.P, d.P, osum[q], d.C - 4 * q);
    }
}

}  // namespace cl
}  // namespace cs
