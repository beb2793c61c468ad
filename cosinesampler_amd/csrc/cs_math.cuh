// cs_math.cuh -- per-sample geometry of the sampler, shared by every kernel (gfx950, fp32).
//
// What the reference does in grid_sampler_compute_source_index[_set_grad] (2d.cu:54-236,
// 3d.cu:64-247) and in the kernel heads (2d.cu:304-335, :413-453, :575-643, :781-835;
// 3d.cu:295-339, :436-492, :660-772, :940-1011), restated once:
//
//   i_j   = unnormalize(g_j) (+ padding)         source coordinate on axis j
//   mu_j  = d i_j / d g_j                        (0 where border/reflection clip it)
//   l_j   = floor(i_j),  t_j = (l_j + 1) - i_j
//   w_j[0] = k(t_j) (low node), w_j[1] = 1 - k(t_j) (high node)
//   d1_j  = mu_j   * k'(t_j)  =  d w_j[1] / d g_j  = -d w_j[0] / d g_j
//   d2_j  = mu_j^2 * k''(t_j) =  d2 w_j[0] / d g_j^2 = -d2 w_j[1] / d g_j^2
//
// One convention (t = high - i, k(t) on the LOW node) is used for every stage; the reference
// mixes it with tau = i - low / k(tau) on the HIGH node in 3d.cu:313-329 (K5-K7) -- the same
// numbers in exact arithmetic because k(1-t) = 1-k(t) for all three kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cs {

enum { PAD_ZEROS = 0, PAD_BORDER = 1, PAD_REFLECTION = 2 };
enum { K_COSINE = 0, K_LINEAR = 1, K_SMOOTHSTEP = 2 };

constexpr float kPi = 3.14159265358979323846f;

// Element types of the channel-major STREAMS (output, grad_output, grad_grad_out, grad_out_ggout): fp32, IEEE half or
// bfloat16 (include/cosine_sampler.h, CS_STREAM_F16 / CS_STREAM_BF16).  The reference dispatches half too (2d.cu:905)
// and, like here, evaluates weights and sums in float whatever the tensor type (2d.cu:239-261, :430-431).  Everything
// else -- table, grid and its cotangents, the input-shaped gradients -- stays fp32.
typedef _Float16 stream_f16;
typedef __bf16 stream_bf16;
template <typename T>
__device__ __forceinline__ float stream_load(const T *p) { return (float)__builtin_nontemporal_load(p); }
template <typename T>
__device__ __forceinline__ void stream_store(T *p, float v) { __builtin_nontemporal_store((T)v, p); }

struct Flags {
    int pad;
    int align;      // as given by the caller
    int multicell;
    int exact;      // CS_KERNEL_EXACT_MIXED: keep the mixed second derivatives the reference drops (include/cosine_sampler.h)
    int pair16;     // 16-bit streams whose rows are 4-byte aligned with P even: lane pairs move whole dwords (cs_tiled.cuh)
};

// k, k', k'' at t in [0,1] (2d.cu:239-261).  ORDER = highest derivative wanted.
template <int KERNEL, int ORDER>
__device__ __forceinline__ void kern_eval(float t, float &k0, float &k1, float &k2) {
    k1 = 1.0f;
    k2 = 0.0f;
    if (KERNEL == K_COSINE) {
        // cos(pi t), sin(pi t) on the hardware trig unit: v_cos_f32 / v_sin_f32 take their argument
        // in revolutions, so pi*t is t/2 with no range reduction to get wrong.  Max abs error on
        // [0,1] measured on MI355X: 1.25e-7 (tools/microbench.hip MB5) -- tighter than the
        // __cosf/__sinf the reference gets under --use_fast_math (setup.py:37; 2.7e-7).
        float c = __builtin_amdgcn_cosf(0.5f * t);
        k0 = 0.5f * (1.0f - c);
        if (ORDER >= 1) k1 = (0.5f * kPi) * __builtin_amdgcn_sinf(0.5f * t);
        if (ORDER >= 2) k2 = (0.5f * kPi * kPi) * c;
    } else if (KERNEL == K_SMOOTHSTEP) {
        k0 = t * t * (3.0f - 2.0f * t);
        if (ORDER >= 1) k1 = 6.0f * t * (1.0f - t);
        if (ORDER >= 2) k2 = 6.0f - 12.0f * t;
    } else {
        k0 = t;
    }
}

// Coordinate arithmetic is pinned operation by operation: t = frac(i) inherits the absolute
// rounding error of i (one ulp of a coordinate ~ size) and k'(t), k''(t) amplify it, so two
// evaluations of the same formula that differ only in where the compiler fused a multiply-add
// differ by ~1e-5 relative at H=W=256.  Compiler contraction is therefore OFF here and the two
// multiply-adds a GPU build of the reference performs (nvcc/hipcc fuse `x*(size-1) + offset` and
// `(g+1)*size - 1`, as does the HIP build of torch's grid_sampler_unnormalize) are written as
// explicit fmaf -- in this file and in oracle/cs_oracle.c alike -- so the source index is
// bit-identical to the CPU oracle's and to torch.nn.functional.grid_sample's.
__device__ __forceinline__ float clip_coord(float in, int limit, float &g) {  // 2d.cu:99-116
#pragma clang fp contract(off)
    if (in <= 0.0f) { g = 0.0f; return 0.0f; }
    float hi = (float)(limit - 1);
    if (in >= hi) { g = 0.0f; return hi; }
    g = 1.0f;
    return in;
}

__device__ __forceinline__ float reflect_coord(float in, int twice_low, int twice_high, float &g) {  // 2d.cu:145-171
#pragma clang fp contract(off)
    if (twice_low == twice_high) { g = 0.0f; return 0.0f; }
    float lo = (float)twice_low * 0.5f;
    float span = (float)(twice_high - twice_low) * 0.5f;
    in = in - lo;
    float sgn = 1.0f;
    if (in < 0.0f) { sgn = -1.0f; in = -in; }
    float extra = fmodf(in, span);
    int flips = (int)floorf(in / span);
    if ((flips & 1) == 0) { g = sgn; return extra + lo; }
    g = -sgn;
    return span - extra + lo;
}

// source coordinate and mu = d i / d g (2d.cu:212-236)
__device__ __forceinline__ float source_index(float g, int size, int pad, int align, float off, int multicell,
                                              float &mu) {
#pragma clang fp contract(off)
    float c;
    if (align) {
        int s = multicell ? size - 1 : size;            // 2d.cu:57-59
        mu = (float)(s - 1) * 0.5f;                     // 2d.cu:80
        c = fmaf((g + 1.0f) * 0.5f, (float)(s - 1), off); // 2d.cu:61
    } else {
        mu = (float)size * 0.5f;                        // 2d.cu:84
        c = fmaf(g + 1.0f, (float)size, -1.0f) * 0.5f + off;   // 2d.cu:64
    }
    if (pad == PAD_BORDER) {
        float gc;
        c = clip_coord(c, size, gc);
        mu *= gc;
    } else if (pad == PAD_REFLECTION) {
        float gr, gc;
        if (align) c = reflect_coord(c, 0, 2 * (size - 2), gr);  // NB size-2 (2d.cu:185, :227)
        else       c = reflect_coord(c, -1, 2 * size - 1, gr);
        c = clip_coord(c, size, gc);
        mu *= gr * gc;
    }
    return c;
}

struct Axis {
    int lo;       // low node index (may be out of range: zero padding)
    float w[2];   // blending weights of the low / high node
    float d1;     // mu k'(t)
    float d2;     // mu^2 k''(t)
    float t;      // (lo + 1) - i, the blending argument: 1 - t is the position inside the cell (cs_coherent.cuh orders by it)
};

template <int KERNEL, int ORDER>
__device__ __forceinline__ Axis make_axis(float g, int size, const Flags &f, int align, float off) {
#pragma clang fp contract(off)
    Axis a;
    float mu;
    float i = source_index(g, size, f.pad, align, off, f.multicell, mu);
    float fl = floorf(i);
    // anything absurd (NaN, |i| >= 2^30) is pushed fully out of range: no node is touched
    bool sane = (i > -1073741824.0f) && (i < 1073741824.0f);
    a.lo = sane ? (int)fl : -4;
    float t = (fl + 1.0f) - i;  // "ix_right - ix", 2d.cu:315
    // ... and gets finite weights, so that such a sample yields zeros on every path instead of NaN * 0
    // (the reference leaves this to undefined float->int conversions)
    if (!sane) { t = 0.5f; i = 0.5f; fl = 0.0f; mu = 0.0f; }
    float k0, k1, k2;
    kern_eval<KERNEL, ORDER>(t, k0, k1, k2);
    a.w[0] = k0;
    // linear: torch.nn.functional.grid_sample's own form (ix - ix_nw), so that the linear /
    // multicell=False specialisation reproduces it bit for bit; the reference's 1-(ix_right-ix)
    // (2d.cu:329) differs from it by at most one ulp.
    a.w[1] = (KERNEL == K_LINEAR) ? (i - fl) : (1.0f - k0);
    a.d1 = mu * k1;
    a.d2 = mu * mu * k2;
    a.t = t;
    return a;
}

// mu^3 k3(t), k3 = third derivative of the blending kernel of one axis -- only for the opt-in third-order grid gradient (direct_bbb_grid); the hot kernels never
// need it, so it is not part of Axis.  Same coordinate arithmetic as make_axis.
template <int KERNEL>
__device__ __forceinline__ float third_coef(float g, int size, const Flags &f, int align, float off) {
#pragma clang fp contract(off)
    float mu;
    float i = source_index(g, size, f.pad, align, off, f.multicell, mu);
    if (!((i > -1073741824.0f) && (i < 1073741824.0f))) return 0.0f;
    float t = (floorf(i) + 1.0f) - i;
    float k3 = 0.0f;
    if (KERNEL == K_COSINE) k3 = (-0.5f * kPi * kPi * kPi) * __builtin_amdgcn_sinf(0.5f * t);
    else if (KERNEL == K_SMOOTHSTEP) k3 = -12.0f;
    return mu * mu * mu * k3;
}

}  // namespace cs
