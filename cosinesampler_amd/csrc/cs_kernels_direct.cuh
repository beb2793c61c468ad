// cs_kernels_direct.cuh -- "direct" kernels: one lane per output sample (n,p), all C channels in
// a register loop, node values gathered straight from the caller's NC[D]HW feature grid and the
// grad_input scatter done with hardware fp32 atomics (global_atomic_add_f32).
//
// These are the general-purpose path: any C, any padding mode, any size.  They already remove
// the reference's pure overheads (SURVEY section 7 "hard part 3"): no read-modify-write of
// `output` through global memory (2d.cu:341-352), no atomics + 1 GiB memset for the
// lane-private grad_grad_out (2d.cu:699-703, 2d.cpp:101), grad_out_grid read once per sample
// (2d.cu:688-689), grad_grid stored once (2d.cu:501-505), and K4 + the extra K3 launch of
// modules_2d.py:106-111 fused into one pass.
//
// Stage maths: SURVEY.md Appendix A; per-stage reference text:
//   fwd  2d.cu:297-355   3d.cu:287-370
//   bwd  2d.cu:406-506   3d.cu:428-583
//   bb   2d.cu:569-716   3d.cu:652-868   (3D only: mixed second derivatives + gOutInput->grad_grid)
//   bbb  2d.cu:774-890   3d.cu:932-1070  (pure second derivatives only)
#pragma once
#include "cs_math.cuh"

namespace cs {

struct Dims {
    int N, C;
    int size[3];    // x (W), y (H), z (D); size[2] = 1 for 2D
    int64_t P;      // samples per n
    int64_t S;      // N * P
    int64_t vol;    // D*H*W elements of one (n,c) volume
    // layout of the table the kernels gather `input` from: element (n, c, node) sits at
    // n*C*vol + node*tab_ns + c*tab_cs  -- (1, vol) for the caller's NC[D]HW tensor, (C, 1) for the
    // channels-last copy made by pack_channels_last (one node = one contiguous C-float row)
    int64_t tab_ns, tab_cs;
    int64_t go_ns, ho_ns;   // elements between consecutive n of gOut / hO: C*P, or 0 for an n-broadcast (expanded) tensor
    int64_t out_ns;         // ... of the stream a stage WRITES (output, grad_grad_out): C*P, or more when the caller's tensor has
                            // more channels than this call computes (a channel range of a wider stream: cs_cotangent_layout)
    // points between consecutive n of grid, grad_out_grid and grad_out_ggrid: P, or 0 when ONE set of P points serves
    // every n (CS_GRID_BROADCAST: PIXEL's grid.repeat(N, ...), test/test_2d.py:38, without the repeat)
    int64_t grid_ns;
    // CS_SUM_OVER_N on the channels-last point kernels (cs_points_cl.cuh, 3D): non-zero = one lane owns a POINT for all N
    // tables and the per-point results leave summed over them (the streams then have no n: go_ns = ho_ns = grid_ns = 0)
    int nsum;
    int xcd;       // the tiled 2D backward point kernels take their workgroups in the XCD-aware order (cs_tiled.cuh pblk)
    __host__ __device__ __forceinline__ int64_t gpt(int n, int64_t p) const { return (int64_t)n * grid_ns + p; }
};

template <int DIM>
struct Sample {
    static constexpr int NC = 1 << DIM;
    int n;
    int64_t p;
    Axis ax[DIM];
    int64_t node[NC];  // element offset inside one (n,c) volume, or -1 when zero-padded

    template <int KERNEL, int ORDER>
    __device__ __forceinline__ bool load(const float *grid, const float *offset, const Dims &d, const Flags &f,
                                         int align) {
        return load_at<KERNEL, ORDER>((int64_t)blockIdx.x * blockDim.x + threadIdx.x, grid, offset, d, f, align);
    }
    template <int KERNEL, int ORDER>
    __device__ __forceinline__ bool load_at(int64_t s, const float *grid, const float *offset, const Dims &d,
                                            const Flags &f, int align) {
        if (s >= d.S) return false;
        n = (int)(s / d.P);
        p = s - (int64_t)n * d.P;
        float off = offset[n];
        const float *g = grid + d.gpt(n, p) * DIM;
        float gc[DIM];
        if (DIM == 2) {
            float2 v = *reinterpret_cast<const float2 *>(g);
            gc[0] = v.x;
            gc[1] = v.y;
        } else {
#pragma unroll
            for (int j = 0; j < DIM; ++j) gc[j] = g[j];
        }
#pragma unroll
        for (int j = 0; j < DIM; ++j) ax[j] = make_axis<KERNEL, ORDER>(gc[j], d.size[j], f, align, off);
#pragma unroll
        for (int a = 0; a < NC; ++a) {
            bool ok = true;
            int64_t o = 0;
            int64_t stride = 1;
#pragma unroll
            for (int j = 0; j < DIM; ++j) {
                int idx = ax[j].lo + ((a >> j) & 1);
                ok = ok && (idx >= 0) && (idx < d.size[j]);
                o += (int64_t)idx * stride;
                stride *= d.size[j];
            }
            node[a] = ok ? o : -1;
        }
        return true;
    }

    // W_a = prod_j w_j[bit_j(a)], multiplied in x, y, z order
    __device__ __forceinline__ void weights(float (&W)[NC]) const {
#pragma unroll
        for (int a = 0; a < NC; ++a) {
            float w = ax[0].w[a & 1];
#pragma unroll
            for (int j = 1; j < DIM; ++j) w *= ax[j].w[(a >> j) & 1];
            W[a] = w;
        }
    }
    // product of the blending weights of every axis except j (and except k, if k >= 0)
    __device__ __forceinline__ float others(int a, int j, int k = -1) const {
        float w = 1.0f;
#pragma unroll
        for (int m = 0; m < DIM; ++m)
            if (m != j && m != k) w *= ax[m].w[(a >> m) & 1];
        return w;
    }
    // F_j[a] = dW_a/dg_j
    __device__ __forceinline__ float first(int a, int j) const {
        float s = ((a >> j) & 1) ? ax[j].d1 : -ax[j].d1;
        return s * others(a, j);
    }
    // H_jj[a] = d2W_a/dg_j^2
    __device__ __forceinline__ float pure2(int a, int j) const {
        float s = ((a >> j) & 1) ? -ax[j].d2 : ax[j].d2;
        return s * others(a, j);
    }
    // H_jk[a] = d2W_a/dg_j dg_k, j != k
    __device__ __forceinline__ float mixed2(int a, int j, int k) const {
        float sj = ((a >> j) & 1) ? ax[j].d1 : -ax[j].d1;
        float sk = ((a >> k) & 1) ? ax[k].d1 : -ax[k].d1;
        return sj * sk * others(a, j, k);
    }
};

template <int DIM>
__device__ __forceinline__ void gather(const float *vol, const int64_t (&node)[1 << DIM], float (&v)[1 << DIM],
                                       int64_t ns = 1) {
#pragma unroll
    for (int a = 0; a < (1 << DIM); ++a) v[a] = node[a] >= 0 ? vol[node[a] * ns] : 0.0f;
}

// ----------------------------------------------------------------------------------------------
// forward:  out[n,c,p] = sum_a W_a * input[n,c,q_a]
// ----------------------------------------------------------------------------------------------
template <int DIM, int KERNEL>
__global__ __launch_bounds__(256) void direct_forward(const float *__restrict__ input, const float *__restrict__ grid,
                                                      const float *__restrict__ offset, float *__restrict__ out,
                                                      Dims d, Flags f) {
    constexpr int NC = 1 << DIM;
    Sample<DIM> sm;
    // 2D forward: align_corners hard-wired to 1 in the reference (2d.cu:307-308); 3D honours it.
    if (!sm.template load<KERNEL, 0>(grid, offset, d, f, DIM == 2 ? 1 : f.align)) return;
    float W[NC];
    sm.weights(W);
    const float *in = input + (int64_t)sm.n * d.C * d.vol;
    float *o = out + (int64_t)sm.n * d.out_ns + sm.p;
    for (int c = 0; c < d.C; ++c) {
        float v[NC];
        gather<DIM>(in, sm.node, v, d.tab_ns);
        float acc = 0.0f;
#pragma unroll
        for (int a = 0; a < NC; ++a)
            if (sm.node[a] >= 0) acc = fmaf(v[a], W[a], acc);
        *o = acc;
        in += d.tab_cs;
        o += d.P;
    }
}

// ----------------------------------------------------------------------------------------------
// backward:  grad_input[n,c,q_a] += W_a * gOut ;  grad_grid[s,j] = sum_c gOut * sum_a F_j[a] * input[q_a]
// ----------------------------------------------------------------------------------------------
template <int DIM, int KERNEL>
__global__ __launch_bounds__(256) void direct_backward(const float *__restrict__ gOut, const float *__restrict__ input,
                                                       const float *__restrict__ grid, const float *__restrict__ offset,
                                                       float *__restrict__ grad_input, float *__restrict__ grad_grid,
                                                       Dims d, Flags f) {
    constexpr int NC = 1 << DIM;
    Sample<DIM> sm;
    if (!sm.template load<KERNEL, 1>(grid, offset, d, f, f.align)) return;
    float W[NC];
    sm.weights(W);
    // sum_a F_j[a] v_a = d1_j * sum_a (+/-) others(a,j) v_a : keep the signed sums, scale at the end
    float oth[DIM][NC];
#pragma unroll
    for (int j = 0; j < DIM; ++j)
#pragma unroll
        for (int a = 0; a < NC; ++a) oth[j][a] = ((a >> j) & 1) ? sm.others(a, j) : -sm.others(a, j);
    float acc[DIM];
#pragma unroll
    for (int j = 0; j < DIM; ++j) acc[j] = 0.0f;

    const float *in = input + (int64_t)sm.n * d.C * d.vol;
    float *gi = grad_input ? grad_input + (int64_t)sm.n * d.C * d.vol : nullptr;
    const float *go = gOut + (int64_t)sm.n * d.go_ns + sm.p;
    for (int c = 0; c < d.C; ++c) {
        float g = *go;
        float v[NC];
        gather<DIM>(in, sm.node, v, d.tab_ns);
        if (gi) {
#pragma unroll
            for (int a = 0; a < NC; ++a)
                if (sm.node[a] >= 0) unsafeAtomicAdd(gi + sm.node[a], W[a] * g);
            gi += d.vol;
        }
#pragma unroll
        for (int j = 0; j < DIM; ++j) {
            float t = 0.0f;
#pragma unroll
            for (int a = 0; a < NC; ++a) t = fmaf(v[a], oth[j][a], t);
            acc[j] = fmaf(t, g, acc[j]);
        }
        in += d.tab_cs;
        go += d.P;
    }
    float *gg = grad_grid + ((int64_t)sm.n * d.P + sm.p) * DIM;
#pragma unroll
    for (int j = 0; j < DIM; ++j) gg[j] = sm.ax[j].d1 * acc[j];
}

// ----------------------------------------------------------------------------------------------
// backward_backward (cotangents cI = gOutInput [nullable], cG = gOutGrid [nullable = 0]):
//   D_a = sum_j F_j[a] cG_j
//   gInput[n,c,q_a] += gOut * D_a
//   ggOut[n,c,s]     = sum_a input[q_a] D_a  (+ sum_a cI[q_a] W_a)
//   gGrid[s,j]       = sum_c gOut sum_a input[q_a] S_j[a]  (+ 3D: sum_c gOut sum_a cI[q_a] F_j[a])
//   S_j[a] = H_jj[a] cG_j                       (2D: pure terms only, 2d.cu:705-706)
//          = sum_k H_jk[a] cG_k                 (3D, 3d.cu:848-856)
// ----------------------------------------------------------------------------------------------
template <int DIM, int KERNEL>
__global__ __launch_bounds__(256) void direct_backward_backward(
    const float *__restrict__ cI, const float *__restrict__ cG, const float *__restrict__ input,
    const float *__restrict__ grid, const float *__restrict__ gOut, const float *__restrict__ offset,
    float *__restrict__ gInput, float *__restrict__ gGrid, float *__restrict__ ggOut, Dims d, Flags f) {
    constexpr int NC = 1 << DIM;
    const bool FULL = (DIM == 3) || f.exact;   // 2D: the reference keeps pure terms only unless asked otherwise
    Sample<DIM> sm;
    if (!sm.template load<KERNEL, 2>(grid, offset, d, f, f.align)) return;
    float cg[DIM];
#pragma unroll
    for (int j = 0; j < DIM; ++j) cg[j] = cG ? cG[d.gpt(sm.n, sm.p) * DIM + j] : 0.0f;

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

    const float *in = input + (int64_t)sm.n * d.C * d.vol;
    const float *ci = cI ? cI + (int64_t)sm.n * d.C * d.vol : nullptr;
    float *gi = gInput ? gInput + (int64_t)sm.n * d.C * d.vol : nullptr;
    const float *go = gOut + (int64_t)sm.n * d.go_ns + sm.p;
    float *ggo = ggOut + (int64_t)sm.n * d.out_ns + sm.p;
    for (int c = 0; c < d.C; ++c) {
        float g = *go;
        float v[NC];
        gather<DIM>(in, sm.node, v, d.tab_ns);
        float o = 0.0f;
#pragma unroll
        for (int a = 0; a < NC; ++a) o = fmaf(v[a], Dm[a], o);
        if (ci) {
            float u[NC];
            gather<DIM>(ci, sm.node, u);
#pragma unroll
            for (int a = 0; a < NC; ++a) o = fmaf(u[a], W[a], o);
            if (FULL) {
#pragma unroll
                for (int j = 0; j < DIM; ++j) {
                    float t = 0.0f;
#pragma unroll
                    for (int a = 0; a < NC; ++a) t = fmaf(u[a], F[j][a], t);
                    acc[j] = fmaf(t, g, acc[j]);
                }
            }
            ci += d.vol;
        }
        *ggo = o;
#pragma unroll
        for (int j = 0; j < DIM; ++j) {
            float t = 0.0f;
#pragma unroll
            for (int a = 0; a < NC; ++a) t = fmaf(v[a], Sg[j][a], t);
            acc[j] = fmaf(t, g, acc[j]);
        }
        if (gInput) {
#pragma unroll
            for (int a = 0; a < NC; ++a)
                if (sm.node[a] >= 0) unsafeAtomicAdd(gi + sm.node[a], g * Dm[a]);
            gi += d.vol;
        }
        in += d.tab_cs;
        go += d.P;
        ggo += d.P;
    }
    float *gg = gGrid + ((int64_t)sm.n * d.P + sm.p) * DIM;
#pragma unroll
    for (int j = 0; j < DIM; ++j) gg[j] = acc[j];
}

// ----------------------------------------------------------------------------------------------
// third backward, fused (cotangents hG = gOutgGrid [nullable = 0], hO = gOutggOut [nullable = 0]):
//   E_a = sum_j H_jj[a] hG_j cG_j                (pure terms only, 2d.cu:833-834, 3d.cu:1008-1010)
//   ggOut[n,c,s]      = sum_a input[q_a] E_a
//   gInput[n,c,q_a]  += gOut * E_a + hO * D_a    (the D_a part is the reference's extra
//                                                 backward_backward launch, modules_2d.py:109-111)
// ----------------------------------------------------------------------------------------------
template <int DIM, int KERNEL>
__global__ __launch_bounds__(256) void direct_bbb_fused(
    const float *__restrict__ input, const float *__restrict__ grid, const float *__restrict__ gOut,
    const float *__restrict__ cG, const float *__restrict__ hG, const float *__restrict__ hO,
    const float *__restrict__ offset, float *__restrict__ gInput, float *__restrict__ ggOut, Dims d, Flags f) {
    constexpr int NC = 1 << DIM;
    Sample<DIM> sm;
    if (!sm.template load<KERNEL, 2>(grid, offset, d, f, f.align)) return;
    float cg[DIM], hg[DIM];
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        int64_t o = d.gpt(sm.n, sm.p) * DIM + j;
        cg[j] = cG ? cG[o] : 0.0f;
        hg[j] = hG ? hG[o] : 0.0f;
    }
    float Dm[NC], Em[NC];
#pragma unroll
    for (int a = 0; a < NC; ++a) {
        float dsum = 0.0f, esum = 0.0f;
#pragma unroll
        for (int j = 0; j < DIM; ++j) {
            dsum = fmaf(sm.first(a, j), cg[j], dsum);
            esum = fmaf(sm.pure2(a, j), hg[j] * cg[j], esum);
            if (f.exact) {   // + the mixed terms H_jk hG_j cG_k the reference drops (2d.cu:833-834, 3d.cu:1008-1010)
#pragma unroll
                for (int k = 0; k < DIM; ++k)
                    if (k != j) esum = fmaf(sm.mixed2(a, j, k), hg[j] * cg[k], esum);
            }
        }
        Dm[a] = dsum;
        Em[a] = esum;
    }
    const float *in = input + (int64_t)sm.n * d.C * d.vol;
    float *gi = gInput ? gInput + (int64_t)sm.n * d.C * d.vol : nullptr;
    const float *go = gOut + (int64_t)sm.n * d.go_ns + sm.p;
    const float *ho = hO ? hO + (int64_t)sm.n * d.ho_ns + sm.p : nullptr;
    float *ggo = ggOut + (int64_t)sm.n * d.out_ns + sm.p;
    for (int c = 0; c < d.C; ++c) {
        float g = *go;
        float h = ho ? *ho : 0.0f;
        float v[NC];
        gather<DIM>(in, sm.node, v, d.tab_ns);
        float o = 0.0f;
#pragma unroll
        for (int a = 0; a < NC; ++a) o = fmaf(v[a], Em[a], o);
        *ggo = o;
        if (gInput) {
#pragma unroll
            for (int a = 0; a < NC; ++a)
                if (sm.node[a] >= 0) unsafeAtomicAdd(gi + sm.node[a], fmaf(g, Em[a], h * Dm[a]));
            gi += d.vol;
        }
        in += d.tab_cs;
        go += d.P;
        ggo += d.P;
        if (ho) ho += d.P;
    }
}

// ----------------------------------------------------------------------------------------------
// NOT in the reference (its third backward returns no gradient for grid, modules_2d.py:111): d/dgrid of
//   Phi = <gGrid, hG> + <ggOut, hO>,   gGrid[s,j] = sum_c G sum_a I_a sum_k H_jk[a] cG_k,   ggOut = sum_a I_a D_a
// (the second backward's outputs with every mixed term, gOutInput absent):
//   gGrid3[s,l] = sum_a ( A_l[a] <G, I_a> + B_l[a] <hO, I_a> ),   A_l[a] = sum_jk T_jkl[a] hG_j cG_k,  B_l[a] = sum_k H_kl[a] cG_k
// with T_jkl = d3 W_a / dg_j dg_k dg_l.  Per axis m and node bit b the factor of derivative order n is
//   n=0: w[b]   n=1: +-d1 (+ for the high node)   n=2: -+d2   n=3: +-d3,   d3 = mu^3 k3(t), k3 = third derivative of the blending kernel.
// One lane per sample, nodes gathered from the caller's tensor: an opt-in path (u_xxx, u_xxy), not a hot one.
// ----------------------------------------------------------------------------------------------
template <int DIM, int KERNEL>
__global__ __launch_bounds__(256) void direct_bbb_grid(
    const float *__restrict__ input, const float *__restrict__ grid, const float *__restrict__ gOut,
    const float *__restrict__ cG, const float *__restrict__ hG, const float *__restrict__ hO,
    const float *__restrict__ offset, float *__restrict__ gGrid3, Dims d, Flags f) {
    constexpr int NC = 1 << DIM;
    Sample<DIM> sm;
    if (!sm.template load<KERNEL, 2>(grid, offset, d, f, f.align)) return;
    const int64_t so = d.gpt(sm.n, sm.p) * DIM;
    float cg[DIM], hg[DIM], d3[DIM];
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        cg[j] = cG ? cG[so + j] : 0.0f;
        hg[j] = hG ? hG[so + j] : 0.0f;
        d3[j] = third_coef<KERNEL>(grid[so + j], d.size[j], f, f.align, offset[sm.n]);
    }
    auto fac = [&](int m, int n, int b) -> float {
        if (n == 0) return sm.ax[m].w[b];
        if (n == 1) return b ? sm.ax[m].d1 : -sm.ax[m].d1;
        if (n == 2) return b ? -sm.ax[m].d2 : sm.ax[m].d2;
        return b ? d3[m] : -d3[m];
    };
    float A[DIM][NC], B[DIM][NC];
#pragma unroll
    for (int a = 0; a < NC; ++a)
#pragma unroll
        for (int l = 0; l < DIM; ++l) {
            float asum = 0.0f, bsum = 0.0f;
#pragma unroll
            for (int k = 0; k < DIM; ++k) {
                float hkl = 1.0f;
#pragma unroll
                for (int m = 0; m < DIM; ++m) hkl *= fac(m, (m == k) + (m == l), (a >> m) & 1);
                bsum = fmaf(hkl, cg[k], bsum);
#pragma unroll
                for (int j = 0; j < DIM; ++j) {
                    float tj = 1.0f;
#pragma unroll
                    for (int m = 0; m < DIM; ++m) tj *= fac(m, (m == j) + (m == k) + (m == l), (a >> m) & 1);
                    asum = fmaf(tj, hg[j] * cg[k], asum);
                }
            }
            A[l][a] = asum;
            B[l][a] = bsum;
        }
    float dg[NC], dh[NC];
#pragma unroll
    for (int a = 0; a < NC; ++a) dg[a] = dh[a] = 0.0f;
    const float *in = input + (int64_t)sm.n * d.C * d.vol;
    const float *go = gOut + (int64_t)sm.n * d.go_ns + sm.p;
    const float *ho = hO ? hO + (int64_t)sm.n * d.ho_ns + sm.p : nullptr;
    for (int c = 0; c < d.C; ++c) {
        float v[NC];
        gather<DIM>(in, sm.node, v, d.tab_ns);
        const float g = *go, h = ho ? *ho : 0.0f;
#pragma unroll
        for (int a = 0; a < NC; ++a) {
            dg[a] = fmaf(g, v[a], dg[a]);
            dh[a] = fmaf(h, v[a], dh[a]);
        }
        in += d.tab_cs;
        go += d.P;
        if (ho) ho += d.P;
    }
#pragma unroll
    for (int l = 0; l < DIM; ++l) {
        float r = 0.0f;
#pragma unroll
        for (int a = 0; a < NC; ++a) r = fmaf(A[l][a], dg[a], fmaf(B[l][a], dh[a], r));
        gGrid3[((int64_t)sm.n * d.P + sm.p) * DIM + l] = r;   // results are per (n, p) whatever the grid's layout
    }
}

// ----------------------------------------------------------------------------------------------
// Row-atomic scatter for shapes outside the tiled path (3D; 2D with C = 1,2,32,64): the
// input-shaped gradient of any backward stage, accumulated into a channels-last scratch
// (N,spatial...,C).  C lanes share a sample, so one wave instruction adds 64/C whole C-float node
// rows: the memory side retires ~20 G atomic REQUESTS/s whether a request is one float or a
// contiguous row (tools/microbench.hip MB4), i.e. C times fewer requests than lane-per-sample.
// PAIR: rows shorter than 64 bytes (C <= 8) are added two at a time -- the x-neighbours of a sample are adjacent in
// the channels-last scratch, 2C lanes cover both, and the request rate is the same up to 64 bytes whatever the
// alignment (tools/microbench_sum.hip A1-A3: 2^25 32-byte rows 1.66 ms, the same data as 2^24 64-byte rows 0.83 ms).
//   MODE 0: W_a * gOut                 (backward,           2d.cu:469-472 / 3d.cu:507-522)
//   MODE 1: D_a * gOut                 (backward_backward,  2d.cu:709     / 3d.cu:858-860)
//   MODE 2: E_a * gOut + D_a * hO      (third backward,     2d.cu:885 / 3d.cu:1063 + modules_2d.py:109-111)
// ----------------------------------------------------------------------------------------------
template <int DIM, int KERNEL, int MODE, bool PAIR>
__global__ __launch_bounds__(256) void row_scatter(const float *__restrict__ grid, const float *__restrict__ offset,
                                                   const float *__restrict__ gOut, const float *__restrict__ cG,
                                                   const float *__restrict__ hG, const float *__restrict__ hO,
                                                   float *__restrict__ acc_cl, Dims d, Flags f, int logC) {
    constexpr int NC = 1 << DIM;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c = (int)(t & ((1 << logC) - 1));
    const int xbit = PAIR ? (int)((t >> logC) & 1) : 0;   // PAIR: this lane's nodes are the ones on its x side
    Sample<DIM> sm;
    if (!sm.template load_at<KERNEL, (MODE == 0 ? 0 : (MODE == 1 ? 1 : 2))>(t >> (logC + (PAIR ? 1 : 0)), grid, offset, d, f, f.align)) return;
    const int64_t so = d.gpt(sm.n, sm.p) * DIM;
    float cg[DIM], hg[DIM];
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        cg[j] = (MODE >= 1 && cG) ? cG[so + j] : 0.0f;
        hg[j] = (MODE == 2 && hG) ? hG[so + j] : 0.0f;
    }
    const float g = gOut[(int64_t)sm.n * d.go_ns + (int64_t)c * d.P + sm.p];
    const float h = (MODE == 2 && hO) ? hO[(int64_t)sm.n * d.ho_ns + (int64_t)c * d.P + sm.p] : 0.0f;
    float *dst = acc_cl + (int64_t)sm.n * d.vol * d.C + c;
    // contribution of node a (compile-time a: no dynamically indexed registers)
    auto value = [&](int a) -> float {
        if (MODE == 0) {
            float w = sm.ax[0].w[a & 1];
#pragma unroll
            for (int j = 1; j < DIM; ++j) w *= sm.ax[j].w[(a >> j) & 1];
            return w * g;
        }
        float dsum = 0.0f, esum = 0.0f;
#pragma unroll
        for (int j = 0; j < DIM; ++j) {
            dsum = fmaf(sm.first(a, j), cg[j], dsum);
            if (MODE == 2) {
                esum = fmaf(sm.pure2(a, j), hg[j] * cg[j], esum);
                if (f.exact) {
#pragma unroll
                    for (int k = 0; k < DIM; ++k)
                        if (k != j) esum = fmaf(sm.mixed2(a, j, k), hg[j] * cg[k], esum);
                }
            }
        }
        return MODE == 1 ? g * dsum : fmaf(g, esum, h * dsum);
    };
    if (PAIR) {
#pragma unroll
        for (int k = 0; k < NC / 2; ++k) {   // lanes with xbit = 0 take node 2k, the others its x-neighbour 2k+1
            const auto node = xbit ? sm.node[2 * k + 1] : sm.node[2 * k];
            const float v = xbit ? value(2 * k + 1) : value(2 * k);
            if (node >= 0) unsafeAtomicAdd(dst + node * d.C, v);
        }
    } else {
#pragma unroll
        for (int a = 0; a < NC; ++a)
            if (sm.node[a] >= 0) unsafeAtomicAdd(dst + sm.node[a] * d.C, value(a));
    }
}

// channels-last scratch (N,vol,C) -> caller's (N,C,vol); any C <= 64.  NV nodes per workgroup: every channel
// plane receives NV*4 contiguous bytes per workgroup (1 KiB at NV = 256; 64 nodes measured 3.6 TB/s).
__host__ __device__ constexpr int unpack_nv(int C) { return C > 32 ? 128 : 256; }   // tile stays under 64 KiB of LDS
static __global__ __launch_bounds__(256) void unpack_channels_last(const float *__restrict__ in, float *__restrict__ out,
                                                            int C, int CP, int64_t vol) {   // CP: channels of `in` (padded)
    extern __shared__ float tile[];  // [NV][C+1]
    const int NV = unpack_nv(C);
    const int n = blockIdx.y;
    const int64_t v0 = (int64_t)blockIdx.x * NV;
    for (int idx = threadIdx.x; idx < C * NV; idx += 256) {
        int v = idx / C, c = idx - v * C;
        tile[v * (C + 1) + c] = (v0 + v < vol) ? in[((int64_t)n * vol + v0 + v) * CP + c] : 0.0f;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < C * NV; idx += 256) {
        int c = idx / NV, v = idx - c * NV;
        if (v0 + v < vol) out[((int64_t)n * C + c) * vol + v0 + v] = tile[v * (C + 1) + c];
    }
}

// ------------------------------------------------------------------------------------------------
// The same two layout changes with 16-byte accesses on both sides (CP a multiple of 4, vol a multiple of 4): a
// workgroup moves NV nodes x CP channels through an LDS tile [CP][NV + 4]; channels-last rows are read / written as
// float4 quads, channel planes as float4 runs of 4 consecutive nodes.  The scalar kernels above moved 1 GiB in
// 0.28-0.30 ms (3D config: 512 MiB table / accumulator).
// ------------------------------------------------------------------------------------------------
__host__ __device__ constexpr int cl4_nv(int CP) { return CP <= 8 ? 1024 : CP <= 16 ? 512 : CP <= 32 ? 256 : 128; }
__host__ __device__ constexpr size_t cl4_lds(int CP) { return (size_t)CP * (cl4_nv(CP) + 4) * 4; }
// (N,vol,CP) channels-last -> (N,C,vol) planes
static __global__ __launch_bounds__(256) void unpack_cl4(const float *__restrict__ in, float *__restrict__ out, int C, int CP,
                                                  int64_t vol) {
    extern __shared__ float tile[];
    const int NV = cl4_nv(CP), LD = NV + 4, CQ = CP >> 2;
    const int n = blockIdx.y;
    const int64_t v0 = (int64_t)blockIdx.x * NV;
    const float4 *src = reinterpret_cast<const float4 *>(in + ((int64_t)n * vol + v0) * CP);
    for (int i = threadIdx.x; i < NV * CQ; i += 256) {
        const int v = i / CQ, cq = i - v * CQ;
        if (v0 + v < vol) {
            const float4 x = src[i];
            tile[(4 * cq) * LD + v] = x.x;
            tile[(4 * cq + 1) * LD + v] = x.y;
            tile[(4 * cq + 2) * LD + v] = x.z;
            tile[(4 * cq + 3) * LD + v] = x.w;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * (NV / 4); i += 256) {
        const int c = i / (NV / 4), v4 = i - c * (NV / 4);
        if (v0 + 4 * v4 < vol)   // vol is a multiple of 4: whole quads only
            *reinterpret_cast<float4 *>(out + ((int64_t)n * C + c) * vol + v0 + 4 * v4) =
                *reinterpret_cast<const float4 *>(tile + c * LD + 4 * v4);
    }
}
// (N,C,vol) planes -> (N,vol,CP) channels-last, channels >= C zero.
// 3D tables are packed Z-PAIRED (slots = 2, gridDim.z = 2, shift = H*W): node v gets two rows, its own and the one of the
// node a z-plane above (zeros past the last plane), so that the 2x2 node rows a sample needs per y -- (x, x+1) x (z, z+1) --
// are ONE contiguous run of 4*CP floats (cs_points_cl.cuh gather_quad): 3 lines per sample instead of 5.4 at C = 8.
static __global__ __launch_bounds__(256) void pack_cl4(const float *__restrict__ in, float *__restrict__ out, int C, int CP,
                                                int64_t vol, int64_t shift, int slots) {
    extern __shared__ float tile[];   // [slots][CP][LD]
    const int NV = cl4_nv(CP) / slots, LD = NV + 4, CQ = CP >> 2;
    const int n = blockIdx.y;
    const int64_t v0 = (int64_t)blockIdx.x * NV;
    for (int i = threadIdx.x; i < slots * CP * (NV / 4); i += 256) {
        const int slot = i / (CP * (NV / 4)), r = i - slot * (CP * (NV / 4)), c = r / (NV / 4), v4 = r - c * (NV / 4);
        const int64_t src = v0 + 4 * v4 + slot * shift;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < C && v0 + 4 * v4 < vol && src < vol) x = *reinterpret_cast<const float4 *>(in + ((int64_t)n * C + c) * vol + src);
        *reinterpret_cast<float4 *>(tile + (slot * CP + c) * LD + 4 * v4) = x;
    }
    __syncthreads();
    float4 *dst = reinterpret_cast<float4 *>(out + ((int64_t)n * vol + v0) * slots * CP);   // [node][slot][CP]: contiguous
    for (int i = threadIdx.x; i < NV * slots * CQ; i += 256) {
        const int v = i / (slots * CQ), r = i - v * (slots * CQ), slot = r / CQ, cq = r - slot * CQ;
        const float *t = tile + (slot * CP + 4 * cq) * LD + v;
        if (v0 + v < vol) dst[i] = make_float4(t[0], t[LD], t[2 * LD], t[3 * LD]);
    }
}

// The z-paired pack reading the table ONCE (round 3): a workgroup owns NV nodes of one (y, x) range and walks ZS z-planes
// with two LDS tiles -- the plane it is writing rows for and the plane above it, which becomes the lower one of the next
// step -- so every input element is read once (+1 plane per ZS) instead of twice by two different workgroups; the next
// plane's loads are in flight while the current rows leave.  Needs plane % NV == 0.  grid (plane / NV, ceil(D / ZS), N).
static __global__ __launch_bounds__(256) void pack_cl4_zcol(const float *__restrict__ in, float *__restrict__ out, int C, int CP,
                                                     int64_t plane, int D, int ZS) {
    extern __shared__ float tile[];   // [2][CP][LD]
    const int NV = cl4_nv(CP) / 2, LD = NV + 4, CQ = CP >> 2, Q = NV / 4;
    constexpr int MAXR = 8;           // float4 per thread and plane: CP * Q / 256 <= 8 for every CP <= 64 (CP * cl4_nv(CP) = 8192)
    const int per = (CP * Q + 255) / 256;
    const int n = blockIdx.z, z0 = blockIdx.y * ZS;
    const int64_t v0 = (int64_t)blockIdx.x * NV, vol = plane * D;
    const int z1 = min(z0 + ZS, D);
    float4 R[MAXR];
    auto issue = [&](int z) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < MAXR; ++k) {
            if (k < per) {
                const int i = k * 256 + threadIdx.x, c = i / Q, v4 = i - c * Q;
                R[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < CP * Q && c < C && z < D)
                    R[k] = *reinterpret_cast<const float4 *>(in + ((int64_t)n * C + c) * vol + (int64_t)z * plane + v0 + 4 * v4);
            }
        }
    };
    auto stash = [&](float *t) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < MAXR; ++k) {
            if (k < per) {
                const int i = k * 256 + threadIdx.x, c = i / Q, v4 = i - c * Q;
                if (i < CP * Q) *reinterpret_cast<float4 *>(t + c * LD + 4 * v4) = R[k];
            }
        }
    };
    float *cur = tile, *nxt = tile + CP * LD;
    issue(z0);
    stash(cur);
    issue(z0 + 1);
    for (int z = z0; z < z1; ++z) {
        stash(nxt);
        __syncthreads();
        if (z + 1 < z1) issue(z + 2);
        float4 *dst = reinterpret_cast<float4 *>(out + ((int64_t)n * vol + (int64_t)z * plane + v0) * 2 * CP);   // [node][slot][CP]
        for (int i = threadIdx.x; i < NV * 2 * CQ; i += 256) {
            const int v = i / (2 * CQ), r = i - v * (2 * CQ), slot = r / CQ, cq = r - slot * CQ;
            const float *t = (slot ? nxt : cur) + (4 * cq) * LD + v;
            dst[i] = make_float4(t[0], t[LD], t[2 * LD], t[3 * LD]);
        }
        __syncthreads();
        float *sw = cur;
        cur = nxt;
        nxt = sw;
    }
}

}  // namespace cs
