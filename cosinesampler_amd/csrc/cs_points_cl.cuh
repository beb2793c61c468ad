// cs_points_cl.cuh -- point kernels for any dimensionality gathering float4 channel quads from the
// channels-last copy of `input` (one node = one contiguous C-float row).  They produce the
// p-ordered outputs of a stage only; the input-shaped gradient comes from row_scatter
// (cs_kernels_direct.cuh).  Used for 3D with C in {4,8,16}: a trilinear sample touches 8 node rows
// instead of 8*C separate lines of the NCDHW tensor.  Formulas as in cs_kernels_direct.cuh.
#pragma once
#include "cs_kernels_direct.cuh"

namespace cs {
namespace cl {

__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 fma4(float s, float4 a, float4 acc) {
    return make_float4(fmaf(s, a.x, acc.x), fmaf(s, a.y, acc.y), fmaf(s, a.z, acc.z), fmaf(s, a.w, acc.w));
}
__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ float4 load_quad(const float *src, int64_t P) {
    return make_float4(__builtin_nontemporal_load(src), __builtin_nontemporal_load(src + P),
                       __builtin_nontemporal_load(src + 2 * P), __builtin_nontemporal_load(src + 3 * P));
}
__device__ __forceinline__ void store_quad(float *dst, int64_t P, float4 o) {
    __builtin_nontemporal_store(o.x, dst);
    __builtin_nontemporal_store(o.y, dst + P);
    __builtin_nontemporal_store(o.z, dst + 2 * P);
    __builtin_nontemporal_store(o.w, dst + 3 * P);
}

// node rows of one channel quad; zero-padded nodes read row 0 and are masked
template <int DIM, int CQ>
__device__ __forceinline__ void gather_quad(const float4 *tab, const Sample<DIM> &sm, int q, float4 (&v)[1 << DIM]) {
#pragma unroll
    for (int a = 0; a < (1 << DIM); ++a) v[a] = tab[(sm.node[a] < 0 ? 0 : sm.node[a]) * CQ + q];
#pragma unroll
    for (int a = 0; a < (1 << DIM); ++a)
        if (sm.node[a] < 0) v[a] = zero4();
}

template <int DIM, int KERNEL, int CQ>
__global__ __launch_bounds__(256) void forward(const float *__restrict__ icl, const float *__restrict__ grid,
                                               const float *__restrict__ offset, float *__restrict__ out, Dims d,
                                               Flags f) {
    constexpr int NC = 1 << DIM, C = 4 * CQ;
    Sample<DIM> sm;
    if (!sm.template load<KERNEL, 0>(grid, offset, d, f, DIM == 2 ? 1 : f.align)) return;
    float W[NC];
    sm.weights(W);
    const float4 *tab = reinterpret_cast<const float4 *>(icl + (int64_t)sm.n * d.vol * C);
    float *o = out + (int64_t)sm.n * C * d.P + sm.p;
    float4 v[CQ][NC];
#pragma unroll
    for (int q = 0; q < CQ; ++q) gather_quad<DIM, CQ>(tab, sm, q, v[q]);
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
        float4 acc = zero4();
#pragma unroll
        for (int a = 0; a < NC; ++a) acc = fma4(W[a], v[q][a], acc);
        store_quad(o + (int64_t)(4 * q) * d.P, d.P, acc);
    }
}

template <int DIM, int KERNEL, int CQ>
__global__ __launch_bounds__(256) void backward(const float *__restrict__ gOut, const float *__restrict__ icl,
                                                const float *__restrict__ grid, const float *__restrict__ offset,
                                                float *__restrict__ grad_grid, Dims d, Flags f) {
    constexpr int NC = 1 << DIM, C = 4 * CQ;
    Sample<DIM> sm;
    if (!sm.template load<KERNEL, 1>(grid, offset, d, f, f.align)) return;
    float oth[DIM][NC];
#pragma unroll
    for (int j = 0; j < DIM; ++j)
#pragma unroll
        for (int a = 0; a < NC; ++a) oth[j][a] = ((a >> j) & 1) ? sm.others(a, j) : -sm.others(a, j);
    float acc[DIM];
#pragma unroll
    for (int j = 0; j < DIM; ++j) acc[j] = 0.0f;
    const float4 *tab = reinterpret_cast<const float4 *>(icl + (int64_t)sm.n * d.vol * C);
    const float *go = gOut + (int64_t)sm.n * d.go_ns + sm.p;
    float4 v[CQ][NC], g[CQ];
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
        g[q] = load_quad(go + (int64_t)(4 * q) * d.P, d.P);
        gather_quad<DIM, CQ>(tab, sm, q, v[q]);
    }
#pragma unroll
    for (int q = 0; q < CQ; ++q)
#pragma unroll
        for (int j = 0; j < DIM; ++j) {
            float4 t = zero4();
#pragma unroll
            for (int a = 0; a < NC; ++a) t = fma4(oth[j][a], v[q][a], t);
            acc[j] += dot4(t, g[q]);
        }
    float *gg = grad_grid + ((int64_t)sm.n * d.P + sm.p) * DIM;
#pragma unroll
    for (int j = 0; j < DIM; ++j) gg[j] = sm.ax[j].d1 * acc[j];
}

template <int DIM, int KERNEL, int CQ, bool HAS_CI>
__global__ __launch_bounds__(256) void backward_backward(const float *__restrict__ cIcl, const float *__restrict__ cG,
                                                         const float *__restrict__ icl, const float *__restrict__ grid,
                                                         const float *__restrict__ gOut, const float *__restrict__ offset,
                                                         float *__restrict__ gGrid, float *__restrict__ ggOut, Dims d,
                                                         Flags f) {
    constexpr int NC = 1 << DIM, C = 4 * CQ;
    constexpr bool FULL = (DIM == 3);   // mixed second derivatives + gOutInput -> grad_grid (3d.cu:758-771, :837-839)
    Sample<DIM> sm;
    if (!sm.template load<KERNEL, 2>(grid, offset, d, f, f.align)) return;
    float cg[DIM];
#pragma unroll
    for (int j = 0; j < DIM; ++j) cg[j] = cG ? cG[((int64_t)sm.n * d.P + sm.p) * DIM + j] : 0.0f;
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
    const float4 *tab = reinterpret_cast<const float4 *>(icl + (int64_t)sm.n * d.vol * C);
    const float4 *ctab = reinterpret_cast<const float4 *>(cIcl + (int64_t)sm.n * d.vol * C);
    const float *go = gOut + (int64_t)sm.n * d.go_ns + sm.p;
    float *ggo = ggOut + (int64_t)sm.n * C * d.P + sm.p;
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
        float4 g = load_quad(go + (int64_t)(4 * q) * d.P, d.P);
        float4 v[NC];
        gather_quad<DIM, CQ>(tab, sm, q, v);
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
            gather_quad<DIM, CQ>(ctab, sm, q, u);
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
        store_quad(ggo + (int64_t)(4 * q) * d.P, d.P, o);
    }
    float *gg = gGrid + ((int64_t)sm.n * d.P + sm.p) * DIM;
#pragma unroll
    for (int j = 0; j < DIM; ++j) gg[j] = acc[j];
}

template <int DIM, int KERNEL, int CQ>
__global__ __launch_bounds__(256) void bbb(const float *__restrict__ icl, const float *__restrict__ grid,
                                           const float *__restrict__ cG, const float *__restrict__ hG,
                                           const float *__restrict__ offset, float *__restrict__ ggOut, Dims d,
                                           Flags f) {
    constexpr int NC = 1 << DIM, C = 4 * CQ;
    Sample<DIM> sm;
    if (!sm.template load<KERNEL, 2>(grid, offset, d, f, f.align)) return;
    float Em[NC];
#pragma unroll
    for (int a = 0; a < NC; ++a) Em[a] = 0.0f;
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        int64_t o = ((int64_t)sm.n * d.P + sm.p) * DIM + j;
        float e = (cG ? cG[o] : 0.0f) * (hG ? hG[o] : 0.0f);
#pragma unroll
        for (int a = 0; a < NC; ++a) Em[a] = fmaf(sm.pure2(a, j), e, Em[a]);   // pure terms only (3d.cu:1008-1010)
    }
    const float4 *tab = reinterpret_cast<const float4 *>(icl + (int64_t)sm.n * d.vol * C);
    float *ggo = ggOut + (int64_t)sm.n * C * d.P + sm.p;
    float4 v[CQ][NC];
#pragma unroll
    for (int q = 0; q < CQ; ++q) gather_quad<DIM, CQ>(tab, sm, q, v[q]);
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
        float4 o = zero4();
#pragma unroll
        for (int a = 0; a < NC; ++a) o = fma4(Em[a], v[q][a], o);
        store_quad(ggo + (int64_t)(4 * q) * d.P, d.P, o);
    }
}

}  // namespace cl
}  // namespace cs
