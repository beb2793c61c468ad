/*
 * oracle/cs_oracle.c -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.
 *
 * A plain-C, single-precision CPU restatement of the eight device kernels of
 * NamGyuKang/CosineSampler (2D + 3D: forward, backward, backward_backward,
 * backward_backward_backward), written from the text of the reference's CUDA
 * sources, quirks included.  It exists so that the hand-written HIP kernels in
 * cosinesampler_amd/csrc can be checked element by element on the same seeded
 * inputs.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library; the product (cosinesampler_amd/) never does.
 *
 * Parity pinning: this restatement is pinned by tests/test_oracle_golden.py
 * against golden vectors captured from the reference's own pure-PyTorch ground
 * truth (test/grid_sampler.py, differentiated by autograd) -- see
 * tests/golden/make_golden.py -- and against torch.nn.functional.grid_sample
 * for the linear kernel.  The reference's CUDA kernels themselves cannot be
 * compiled in the build container (no nvcc, no CUDA device): see DESIGN.md.
 *
 * Reference files followed (all under /root/reference):
 *   2d.cu = cosine_sampler_2d/csrc/cosine_sampler_2d_kernel.cu
 *   3d.cu = cosine_sampler_3d/csrc/cosine_sampler_3d_kernel.cu
 *   mod2d.py / mod3d.py = cosine_sampler_{2,3}d/modules_{2,3}d.py
 *
 * Layouts (all contiguous fp32, as the reference's CHECK_CONTIGUOUS demands):
 *   2D: input (N,C,H,W)   grid (N,Ho,Wo,2)     output/gOut (N,C,Ho,Wo)
 *   3D: input (N,C,D,H,W) grid (N,Do,Ho,Wo,3)  output/gOut (N,C,Do,Ho,Wo)
 *   grid[...,0] indexes W (x), [...,1] indexes H (y), [...,2] indexes D (z).
 *   P below = Ho*Wo (2D) or Do*Ho*Wo (3D): the kernels never use the split.
 *
 * Threads: samples of different n touch disjoint slices of every output, so
 * the n loop is the (optional) OpenMP loop; inside one n everything is serial
 * and therefore deterministic.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define CS_PI_F 3.141592654f /* CUDART_PI_F */

/* The input-shaped gradients are sums of many per-sample terms per node (64 at BASELINE configs[1], 1600 at the
 * reference's own test shapes, test/test_2d.py:26-38).  The reference adds them with fp32 atomics in arrival order
 * (2d.cu:469-472); this restatement adds them serially in fp32, p by p -- a third, equally arbitrary order.  Built with
 * -DCS_ORACLE_ACC_DOUBLE (make dacc: _build/libcs_oracle_dacc.so) the TERMS are still the fp32 products of the reference
 * but their sum is kept in double and rounded once, so that a comparison of a crowded table's gradient measures the
 * kernel under test and not this checker's own serial rounding. */
#include <stdlib.h>
#ifdef CS_ORACLE_ACC_DOUBLE
typedef double acc_t;
static acc_t *acc_open(float *out, size_t count) { return out ? (acc_t *)calloc(count ? count : 1, sizeof(acc_t)) : NULL; }
static int acc_close(acc_t *a, float *out, size_t count) {
    if (!out) return 0;
    if (!a) return 1;
    for (size_t i = 0; i < count; ++i) out[i] = (float)a[i];
    free(a);
    return 0;
}
#else
typedef float acc_t;
static acc_t *acc_open(float *out, size_t count) {
    if (out) memset(out, 0, sizeof(float) * count); /* zeros_like, 2d.cpp:75, :99, :119 */
    return out;
}
static int acc_close(acc_t *a, float *out, size_t count) { (void)a; (void)out; (void)count; return 0; }
#endif

enum { CS_PAD_ZEROS = 0, CS_PAD_BORDER = 1, CS_PAD_REFLECTION = 2 }; /* mod2d.py:4-10 */
enum { CS_K_COSINE = 0, CS_K_LINEAR = 1, CS_K_SMOOTHSTEP = 2 };        /* mod2d.py:12-18 */

/* ------------------------------------------------------------------ */
/* 1-D blending kernels and their derivatives: 2d.cu:239-261, 3d.cu:29-50 */
/* ------------------------------------------------------------------ */
static float kern0(int kernel, float t) {
    if (kernel == CS_K_COSINE) return 0.5f * (1 - cosf(CS_PI_F * t));
    if (kernel == CS_K_SMOOTHSTEP) return t * t * (3.0f - 2.0f * t);
    return t;
}
static float kern1(int kernel, float t) {
    if (kernel == CS_K_COSINE) return 0.5f * CS_PI_F * sinf(CS_PI_F * t);
    if (kernel == CS_K_SMOOTHSTEP) return 6 * t * (1.0f - t);
    return 1.0f;
}
static float kern2(int kernel, float t) {
    if (kernel == CS_K_COSINE) return 0.5f * CS_PI_F * CS_PI_F * cosf(CS_PI_F * t);
    if (kernel == CS_K_SMOOTHSTEP) return 6.0f - 12.0f * t;
    return 0.0f;
}

/* ------------------------------------------------------------------ */
/* grid coordinate -> source index (+ d index / d coordinate)          */
/* 2d.cu:54-236, 3d.cu:64-247 (identical text in both files)           */
/* ------------------------------------------------------------------ */
static float unnormalize(float g, int64_t size, int align, float off, int multicell, float *mult) {
    if (align) {
        if (multicell) size = size - 1;          /* 2d.cu:57-59, :77-79 */
        *mult = (float)(size - 1) / 2;           /* 2d.cu:80 */
        /* a GPU build fuses this multiply-add (nvcc fmad / hipcc contraction); made explicit so
         * that the checker and the kernels agree bit for bit on the source index */
        return fmaf((g + 1) / 2, (float)(size - 1), off); /* 2d.cu:61, :81 */
    }
    *mult = (float)size / 2;                     /* 2d.cu:84 */
    return fmaf(g + 1, (float)size, -1.0f) / 2 + off; /* 2d.cu:64, :85 */
}

static float clip_grad(float in, int64_t limit, float *g) { /* 2d.cu:99-116 */
    if (in <= 0.0f) { *g = 0.0f; return 0.0f; }
    float hi = (float)(limit - 1);
    if (in >= hi) { *g = 0.0f; return hi; }
    *g = 1.0f;
    return in;
}

static float reflect_grad(float in, int64_t twice_low, int64_t twice_high, float *g) { /* 2d.cu:145-171 */
    if (twice_low == twice_high) { *g = 0.0f; return 0.0f; }
    int sgn;
    float lo = (float)twice_low / 2;
    float span = (float)(twice_high - twice_low) / 2;
    in = in - lo;
    if (in < 0.0f) { sgn = -1; in = -in; } else { sgn = 1; }
    float extra = fmodf(in, span);
    int flips = (int)floorf(in / span);
    if (flips % 2 == 0) { *g = (float)sgn; return extra + lo; }
    *g = (float)(-sgn);
    return span - extra + lo;
}

/* grid_sampler_compute_source_index_set_grad, 2d.cu:212-236.  The non-grad
 * variant (2d.cu:197-205) yields the same coordinate; clip_coordinates
 * (2d.cu:91-93) and clip_coordinates_set_grad return the same value. */
static float source_index(float g, int64_t size, int pad, int align, float off, int multicell, float *mult) {
    float gc, gr;
    float c = unnormalize(g, size, align, off, multicell, mult);
    if (pad == CS_PAD_BORDER) {
        c = clip_grad(c, size, &gc);
        *mult = (*mult) * gc;
    } else if (pad == CS_PAD_REFLECTION) {
        if (align) c = reflect_grad(c, 0, 2 * (size - 2), &gr); /* NB size-2, not torch's size-1 */
        else       c = reflect_grad(c, -1, 2 * size - 1, &gr);
        c = clip_grad(c, size, &gc);
        *mult = (*mult) * gr * gc;
    }
    return c;
}

static int inb2(int y, int x, int64_t H, int64_t W) { return y >= 0 && y < H && x >= 0 && x < W; }
static int inb3(int z, int y, int x, int64_t D, int64_t H, int64_t W) {
    return z >= 0 && z < D && y >= 0 && y < H && x >= 0 && x < W;
}

/* =================================================================== */
/*                                2D                                    */
/* =================================================================== */

/* K1 -- cosine_sampler_kernel, 2d.cu:265-356.
 * NB 2d.cu:307-308 passes the literal 1 for align_corners: the 2D forward
 * ignores the caller's flag.  Reproduced. */
int cs2d_forward_cpu(const float *input, const float *grid, const float *offset, float *output,
                     int64_t N, int64_t C, int64_t H, int64_t W, int64_t P,
                     int pad, int align_corners, int kernel, int multicell) {
    (void)align_corners;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        for (int64_t p = 0; p < P; ++p) {
            const float *g = grid + (n * P + p) * 2;
            float mx, my;
            float ix = source_index(g[0], W, pad, 1, offset[n], multicell, &mx);
            float iy = source_index(g[1], H, pad, 1, offset[n], multicell, &my);
            int xl = (int)floorf(ix), yt = (int)floorf(iy);
            int xr = xl + 1, yb = yt + 1;
            float wxl = kern0(kernel, xr - ix); /* "dx_right": weight of the LEFT column */
            float wyt = kern0(kernel, yb - iy);
            float wxr = 1.0f - wxl, wyb = 1.0f - wyt;
            float nw = wxl * wyt, ne = wxr * wyt, sw = wxl * wyb, se = wxr * wyb;
            for (int64_t c = 0; c < C; ++c) {
                const float *in = input + (n * C + c) * H * W;
                float acc = 0.0f;
                if (inb2(yt, xl, H, W)) acc += in[yt * W + xl] * nw;
                if (inb2(yt, xr, H, W)) acc += in[yt * W + xr] * ne;
                if (inb2(yb, xl, H, W)) acc += in[yb * W + xl] * sw;
                if (inb2(yb, xr, H, W)) acc += in[yb * W + xr] * se;
                output[(n * C + c) * P + p] = acc;
            }
        }
    }
    return 0;
}

/* K2 -- cosine_sampler_backward_kernel, 2d.cu:359-507.
 * grad_input may be NULL (input_requires_grad == false, 2d.cpp:73-79);
 * otherwise it is zero-filled here (2d.cpp:75). */
int cs2d_backward_cpu(const float *gOut, const float *input, const float *grid, const float *offset,
                      float *grad_input, float *grad_grid,
                      int64_t N, int64_t C, int64_t H, int64_t W, int64_t P,
                      int pad, int align_corners, int kernel, int multicell) {
    const size_t acc_count = (size_t)(N * C * H * W);
    acc_t *accum = acc_open(grad_input, acc_count);
    if (grad_input && !accum) return 1;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        for (int64_t p = 0; p < P; ++p) {
            const float *g = grid + (n * P + p) * 2;
            float mx, my;
            float ix = source_index(g[0], W, pad, align_corners, offset[n], multicell, &mx);
            float iy = source_index(g[1], H, pad, align_corners, offset[n], multicell, &my);
            int xl = (int)floorf(ix), yt = (int)floorf(iy);
            int xr = xl + 1, yb = yt + 1;
            float tx = xr - ix, ty = yb - iy;
            float dkx = kern1(kernel, tx), dky = kern1(kernel, ty); /* 2d.cu:430-443 */
            float wxl = kern0(kernel, tx), wyt = kern0(kernel, ty);
            float wxr = 1.0f - wxl, wyb = 1.0f - wyt;
            float nw = wxl * wyt, ne = wxr * wyt, sw = wxl * wyb, se = wxr * wyb;
            float gix = 0.0f, giy = 0.0f;
            for (int64_t c = 0; c < C; ++c) {
                const float *in = input + (n * C + c) * H * W;
                float go = gOut[(n * C + c) * P + p];
                if (grad_input) { /* safe_add_2d x4, 2d.cu:469-472 */
                    acc_t *gi = accum + (n * C + c) * H * W;
                    if (inb2(yt, xl, H, W)) gi[yt * W + xl] += nw * go;
                    if (inb2(yt, xr, H, W)) gi[yt * W + xr] += ne * go;
                    if (inb2(yb, xl, H, W)) gi[yb * W + xl] += sw * go;
                    if (inb2(yb, xr, H, W)) gi[yb * W + xr] += se * go;
                }
                if (inb2(yt, xl, H, W)) { float v = in[yt * W + xl]; gix -= v * wyt * go;          giy -= v * wxl * go; }
                if (inb2(yt, xr, H, W)) { float v = in[yt * W + xr]; gix += v * wyt * go;          giy -= v * (1 - wxl) * go; }
                if (inb2(yb, xl, H, W)) { float v = in[yb * W + xl]; gix -= v * (1 - wyt) * go;    giy += v * wxl * go; }
                if (inb2(yb, xr, H, W)) { float v = in[yb * W + xr]; gix += v * (1 - wyt) * go;    giy += v * (1 - wxl) * go; }
            }
            /* 2d.cu:501-503 (the reference re-stores this every channel; same final value) */
            grad_grid[(n * P + p) * 2 + 0] = mx * gix * dkx;
            grad_grid[(n * P + p) * 2 + 1] = my * giy * dky;
        }
    }
    return acc_close(accum, grad_input, acc_count);
}

/* K3 -- cosine_sampler_backward_backward_kernel, 2d.cu:509-717.
 * gOutInput may be NULL (mod2d.py:87-89: all-zero/absent cotangent).
 * Only PURE second derivatives feed gGrid and the gOutInput->gGrid term is
 * absent.  Reproduced. */
int cs2d_backward_backward_cpu(const float *gOutInput, const float *gOutGrid,
                               const float *input, const float *grid, const float *gOut, const float *offset,
                               float *gInput, float *gGrid, float *ggOut,
                               int64_t N, int64_t C, int64_t H, int64_t W, int64_t P,
                               int pad, int align_corners, int kernel, int multicell) {
    const size_t acc_count = (size_t)(N * C * H * W);
    acc_t *accum = acc_open(gInput, acc_count); /* 2d.cpp:99 */
    if (!accum) return 1;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        for (int64_t p = 0; p < P; ++p) {
            const float *g = grid + (n * P + p) * 2;
            float mx, my;
            float ix = source_index(g[0], W, pad, align_corners, offset[n], multicell, &mx);
            float iy = source_index(g[1], H, pad, align_corners, offset[n], multicell, &my);
            int xi[2], yi[2];
            xi[0] = (int)floorf(ix); yi[0] = (int)floorf(iy);
            xi[1] = xi[0] + 1;       yi[1] = yi[0] + 1;
            float tx = xi[1] - ix, ty = yi[1] - iy;
            /* per axis: [corner side][0 weight, 1 first derivative, 2 second derivative]; 2d.cu:591-627 */
            float ax[2][3], ay[2][3];
            ax[0][2] = (kernel == CS_K_LINEAR) ? 0.0f : mx * mx * kern2(kernel, tx);
            ay[0][2] = (kernel == CS_K_LINEAR) ? 0.0f : my * my * kern2(kernel, ty);
            ax[0][1] = -mx * kern1(kernel, tx);
            ay[0][1] = -my * kern1(kernel, ty);
            ax[0][0] = kern0(kernel, tx);
            ay[0][0] = kern0(kernel, ty);
            for (int k = 0; k < 3; ++k) { ax[1][k] = -ax[0][k]; ay[1][k] = -ay[0][k]; }
            ax[1][0] = 1.0f - ax[0][0];
            ay[1][0] = 1.0f - ay[0][0];
            float sc[4], d1x[4], d2x[4], d1y[4], d2y[4];
            for (int a = 0; a < 4; ++a) { /* 2d.cu:633-643 */
                int px = a & 1, py = (a >> 1) & 1;
                sc[a]  = ax[px][0] * ay[py][0];
                d1x[a] = ay[py][0] * ax[px][1];
                d2x[a] = ay[py][0] * ax[px][2];
                d1y[a] = ax[px][0] * ay[py][1];
                d2y[a] = ax[px][0] * ay[py][2];
            }
            float ggx = gOutGrid[(n * P + p) * 2 + 0], ggy = gOutGrid[(n * P + p) * 2 + 1];
            float s2x = 0.0f, s2y = 0.0f;
            for (int64_t c = 0; c < C; ++c) {
                const float *in = input + (n * C + c) * H * W;
                const float *goi = gOutInput ? gOutInput + (n * C + c) * H * W : NULL;
                acc_t *gi = accum + (n * C + c) * H * W;
                float go = gOut[(n * C + c) * P + p];
                float acc = 0.0f; /* the reference accumulates this through atomics on a zeroed ggOut */
                for (int a = 0; a < 4; ++a) {
                    int x = xi[a & 1], y = yi[(a >> 1) & 1];
                    if (!inb2(y, x, H, W)) continue;
                    float v = in[y * W + x];
                    float dLdx = go * d1x[a], dLdy = go * d1y[a];
                    float delta = v * (d1x[a] * ggx + d1y[a] * ggy); /* 2d.cu:691-692 */
                    if (goi) delta += goi[y * W + x] * sc[a];        /* 2d.cu:694-697 */
                    acc += delta;
                    s2x += v * go * (d2x[a] * ggx);                  /* 2d.cu:705 */
                    s2y += v * go * (d2y[a] * ggy);                  /* 2d.cu:706 */
                    gi[y * W + x] += dLdx * ggx + dLdy * ggy;        /* 2d.cu:709 */
                }
                ggOut[(n * C + c) * P + p] = acc;
            }
            gGrid[(n * P + p) * 2 + 0] = s2x; /* 2d.cu:714-715 */
            gGrid[(n * P + p) * 2 + 1] = s2y;
        }
    }
    return acc_close(accum, gInput, acc_count);
}

/* K4 -- cosine_sampler_backward_backward_backward_kernel, 2d.cu:722-891.
 * Pure second derivatives only; input_requires_grad is unused by the
 * reference kernel. */
int cs2d_backward_backward_backward_cpu(const float *input, const float *grid, const float *gOut,
                                        const float *gOutGrid, const float *gOutgGrid, const float *offset,
                                        float *gInput, float *ggOut,
                                        int64_t N, int64_t C, int64_t H, int64_t W, int64_t P,
                                        int pad, int align_corners, int kernel, int multicell) {
    const size_t acc_count = (size_t)(N * C * H * W);
    acc_t *accum = acc_open(gInput, acc_count); /* 2d.cpp:119 */
    if (!accum) return 1;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        for (int64_t p = 0; p < P; ++p) {
            const float *g = grid + (n * P + p) * 2;
            float mx, my;
            float ix = source_index(g[0], W, pad, align_corners, offset[n], multicell, &mx);
            float iy = source_index(g[1], H, pad, align_corners, offset[n], multicell, &my);
            int xi[2], yi[2];
            xi[0] = (int)floorf(ix); yi[0] = (int)floorf(iy);
            xi[1] = xi[0] + 1;       yi[1] = yi[0] + 1;
            float tx = xi[1] - ix, ty = yi[1] - iy;
            float ax[2][2], ay[2][2]; /* [side][0 weight, 1 second derivative]; 2d.cu:798-824 */
            ax[0][1] = (kernel == CS_K_LINEAR) ? 0.0f : mx * mx * kern2(kernel, tx);
            ay[0][1] = (kernel == CS_K_LINEAR) ? 0.0f : my * my * kern2(kernel, ty);
            ax[0][0] = kern0(kernel, tx);
            ay[0][0] = kern0(kernel, ty);
            ax[1][0] = 1.0f - ax[0][0]; ay[1][0] = 1.0f - ay[0][0];
            ax[1][1] = -ax[0][1];       ay[1][1] = -ay[0][1];
            float d2x[4], d2y[4];
            for (int a = 0; a < 4; ++a) { /* 2d.cu:829-835 */
                int px = a & 1, py = (a >> 1) & 1;
                d2x[a] = ay[py][0] * ax[px][1];
                d2y[a] = ax[px][0] * ay[py][1];
            }
            float ggx = gOutGrid[(n * P + p) * 2 + 0],  ggy = gOutGrid[(n * P + p) * 2 + 1];
            float hgx = gOutgGrid[(n * P + p) * 2 + 0], hgy = gOutgGrid[(n * P + p) * 2 + 1];
            for (int64_t c = 0; c < C; ++c) {
                const float *in = input + (n * C + c) * H * W;
                acc_t *gi = accum + (n * C + c) * H * W;
                float go = gOut[(n * C + c) * P + p];
                float acc = 0.0f;
                for (int a = 0; a < 4; ++a) {
                    int x = xi[a & 1], y = yi[(a >> 1) & 1];
                    if (!inb2(y, x, H, W)) continue;
                    float v = in[y * W + x];
                    float e = d2x[a] * hgx * ggx + d2y[a] * hgy * ggy; /* 2d.cu:876, :885 */
                    acc += v * e;
                    gi[y * W + x] += go * e;
                }
                ggOut[(n * C + c) * P + p] = acc;
            }
        }
    }
    return acc_close(accum, gInput, acc_count);
}

/* =================================================================== */
/*                                3D                                    */
/* =================================================================== */

/* K5 -- 3d.cu:250-371.  Uses tau = i - floor(i) and k(tau) as the HIGH-side
 * weight (3d.cu:313-329); honours align_corners (3d.cu:299-301). */
int cs3d_forward_cpu(const float *input, const float *grid, const float *offset, float *output,
                     int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int64_t P,
                     int pad, int align_corners, int kernel, int multicell) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        for (int64_t p = 0; p < P; ++p) {
            const float *g = grid + (n * P + p) * 3;
            float m;
            float ix = source_index(g[0], W, pad, align_corners, offset[n], multicell, &m);
            float iy = source_index(g[1], H, pad, align_corners, offset[n], multicell, &m);
            float iz = source_index(g[2], D, pad, align_corners, offset[n], multicell, &m);
            int xi[2], yi[2], zi[2];
            xi[0] = (int)floorf(ix); yi[0] = (int)floorf(iy); zi[0] = (int)floorf(iz);
            xi[1] = xi[0] + 1; yi[1] = yi[0] + 1; zi[1] = zi[0] + 1;
            float wx[2], wy[2], wz[2];
            wx[1] = kern0(kernel, ix - xi[0]); wy[1] = kern0(kernel, iy - yi[0]); wz[1] = kern0(kernel, iz - zi[0]);
            wx[0] = 1.0f - wx[1]; wy[0] = 1.0f - wy[1]; wz[0] = 1.0f - wz[1];
            float w8[8];
            for (int a = 0; a < 8; ++a) w8[a] = wx[a & 1] * wy[(a >> 1) & 1] * wz[(a >> 2) & 1]; /* 3d.cu:332-339 */
            for (int64_t c = 0; c < C; ++c) {
                const float *in = input + (n * C + c) * D * H * W;
                float acc = 0.0f;
                for (int a = 0; a < 8; ++a) { /* tnw,tne,tsw,tse,bnw,bne,bsw,bse: 3d.cu:345-368 */
                    int x = xi[a & 1], y = yi[(a >> 1) & 1], z = zi[(a >> 2) & 1];
                    if (inb3(z, y, x, D, H, W)) acc += in[(z * H + y) * W + x] * w8[a];
                }
                output[(n * C + c) * P + p] = acc;
            }
        }
    }
    return 0;
}

/* K6 -- 3d.cu:373-584 */
int cs3d_backward_cpu(const float *gOut, const float *input, const float *grid, const float *offset,
                      float *grad_input, float *grad_grid,
                      int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int64_t P,
                      int pad, int align_corners, int kernel, int multicell) {
    const size_t acc_count = (size_t)(N * C * D * H * W);
    acc_t *accum = acc_open(grad_input, acc_count);
    if (grad_input && !accum) return 1;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        for (int64_t p = 0; p < P; ++p) {
            const float *g = grid + (n * P + p) * 3;
            float mx, my, mz;
            float ix = source_index(g[0], W, pad, align_corners, offset[n], multicell, &mx);
            float iy = source_index(g[1], H, pad, align_corners, offset[n], multicell, &my);
            float iz = source_index(g[2], D, pad, align_corners, offset[n], multicell, &mz);
            int xi[2], yi[2], zi[2];
            xi[0] = (int)floorf(ix); yi[0] = (int)floorf(iy); zi[0] = (int)floorf(iz);
            xi[1] = xi[0] + 1; yi[1] = yi[0] + 1; zi[1] = zi[0] + 1;
            float ux = ix - xi[0], uy = iy - yi[0], uz = iz - zi[0];
            float dkx = kern1(kernel, ux), dky = kern1(kernel, uy), dkz = kern1(kernel, uz); /* 3d.cu:460-478 */
            float wx[2], wy[2], wz[2];
            wx[1] = kern0(kernel, ux); wy[1] = kern0(kernel, uy); wz[1] = kern0(kernel, uz);
            wx[0] = 1.0f - wx[1]; wy[0] = 1.0f - wy[1]; wz[0] = 1.0f - wz[1];
            float sx = 0.0f, sy = 0.0f, sz = 0.0f;
            for (int64_t c = 0; c < C; ++c) {
                const float *in = input + (n * C + c) * D * H * W;
                acc_t *gi = grad_input ? accum + (n * C + c) * D * H * W : NULL;
                float go = gOut[(n * C + c) * P + p];
                for (int a = 0; a < 8; ++a) {
                    int px = a & 1, py = (a >> 1) & 1, pz = (a >> 2) & 1;
                    int x = xi[px], y = yi[py], z = zi[pz];
                    if (!inb3(z, y, x, D, H, W)) continue;
                    int64_t el = (z * H + y) * W + x;
                    if (gi) gi[el] += wx[px] * wy[py] * wz[pz] * go;   /* 3d.cu:507-522 */
                    float v = in[el];                                  /* 3d.cu:525-572 */
                    float cx = v * wy[py] * wz[pz] * go;
                    float cy = v * wx[px] * wz[pz] * go;
                    float cz = v * wx[px] * wy[py] * go;
                    if (px) sx += cx; else sx -= cx;
                    if (py) sy += cy; else sy -= cy;
                    if (pz) sz += cz; else sz -= cz;
                }
            }
            grad_grid[(n * P + p) * 3 + 0] = mx * sx * dkx; /* 3d.cu:579-582 */
            grad_grid[(n * P + p) * 3 + 1] = my * sy * dky;
            grad_grid[(n * P + p) * 3 + 2] = mz * sz * dkz;
        }
    }
    return acc_close(accum, grad_input, acc_count);
}

/* K7 -- 3d.cu:587-870.  Has the mixed second derivatives (3d.cu:758-771) and
 * the gOutInput -> grad_grid term (3d.cu:837-839) that the 2D kernel lacks. */
int cs3d_backward_backward_cpu(const float *gOutInput, const float *gOutGrid,
                               const float *input, const float *grid, const float *gOut, const float *offset,
                               float *gInput, float *gGrid, float *ggOut,
                               int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int64_t P,
                               int pad, int align_corners, int kernel, int multicell) {
    const size_t acc_count = (size_t)(N * C * D * H * W);
    acc_t *accum = acc_open(gInput, acc_count);
    if (!accum) return 1;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        for (int64_t p = 0; p < P; ++p) {
            const float *g = grid + (n * P + p) * 3;
            float m[3];
            float ic[3];
            ic[0] = source_index(g[0], W, pad, align_corners, offset[n], multicell, &m[0]);
            ic[1] = source_index(g[1], H, pad, align_corners, offset[n], multicell, &m[1]);
            ic[2] = source_index(g[2], D, pad, align_corners, offset[n], multicell, &m[2]);
            int idx[3][2];
            float A[3][2][3]; /* [axis][side][0 weight, 1 first, 2 second]; 3d.cu:680-746 */
            for (int j = 0; j < 3; ++j) {
                idx[j][0] = (int)floorf(ic[j]);
                idx[j][1] = idx[j][0] + 1;
                float u = ic[j] - idx[j][0];
                float d1 = m[j], d2 = 0.0f;
                if (kernel != CS_K_LINEAR) {
                    d1 *= kern1(kernel, u);
                    d2 = m[j] * m[j] * kern2(kernel, u);
                }
                A[j][1][0] = kern0(kernel, u); A[j][1][1] = d1;  A[j][1][2] = d2;
                A[j][0][0] = 1.0f - A[j][1][0]; A[j][0][1] = -d1; A[j][0][2] = -d2;
            }
            float sc[8], od[8][12];
            for (int a = 0; a < 8; ++a) { /* 3d.cu:751-772 */
                int px = a & 1, py = (a >> 1) & 1, pz = (a >> 2) & 1;
                const float *X = A[0][px], *Y = A[1][py], *Z = A[2][pz];
                sc[a] = X[0] * Y[0] * Z[0];
                od[a][0]  = Y[0] * Z[0] * X[1];
                od[a][1]  = Y[0] * Z[0] * X[2];
                od[a][2]  = Y[1] * Z[0] * X[1];
                od[a][3]  = Y[0] * Z[1] * X[1];
                od[a][4]  = X[0] * Z[0] * Y[1];
                od[a][5]  = X[0] * Z[0] * Y[2];
                od[a][6]  = X[1] * Z[0] * Y[1];
                od[a][7]  = X[0] * Z[1] * Y[1];
                od[a][8]  = X[0] * Y[0] * Z[1];
                od[a][9]  = X[0] * Y[0] * Z[2];
                od[a][10] = X[1] * Y[0] * Z[1];
                od[a][11] = X[0] * Y[1] * Z[1];
            }
            const float *gg = gOutGrid + (n * P + p) * 3;
            float s2x = 0.0f, s2y = 0.0f, s2z = 0.0f;
            for (int64_t c = 0; c < C; ++c) {
                const float *in = input + (n * C + c) * D * H * W;
                const float *goi = gOutInput ? gOutInput + (n * C + c) * D * H * W : NULL;
                acc_t *gi = accum + (n * C + c) * D * H * W;
                float go = gOut[(n * C + c) * P + p];
                float acc = 0.0f;
                for (int a = 0; a < 8; ++a) {
                    int x = idx[0][a & 1], y = idx[1][(a >> 1) & 1], z = idx[2][(a >> 2) & 1];
                    if (!inb3(z, y, x, D, H, W)) continue;
                    int64_t el = (z * H + y) * W + x;
                    float v = in[el];
                    float dLdx = go * od[a][0], dLdy = go * od[a][4], dLdz = go * od[a][8];
                    float delta = v * (od[a][0] * gg[0] + od[a][4] * gg[1] + od[a][8] * gg[2]); /* 3d.cu:830-832 */
                    if (goi) {                                                                  /* 3d.cu:834-840 */
                        float t = goi[el];
                        delta += t * sc[a];
                        s2x += dLdx * t; s2y += dLdy * t; s2z += dLdz * t;
                    }
                    acc += delta;
                    s2x += v * go * (od[a][1] * gg[0] + od[a][2] * gg[1] + od[a][3] * gg[2]);   /* 3d.cu:848-856 */
                    s2y += v * go * (od[a][6] * gg[0] + od[a][5] * gg[1] + od[a][7] * gg[2]);
                    s2z += v * go * (od[a][10] * gg[0] + od[a][11] * gg[1] + od[a][9] * gg[2]);
                    gi[el] += dLdx * gg[0] + dLdy * gg[1] + dLdz * gg[2];                       /* 3d.cu:858-860 */
                }
                ggOut[(n * C + c) * P + p] = acc;
            }
            gGrid[(n * P + p) * 3 + 0] = s2x; /* 3d.cu:865-867 */
            gGrid[(n * P + p) * 3 + 1] = s2y;
            gGrid[(n * P + p) * 3 + 2] = s2z;
        }
    }
    return acc_close(accum, gInput, acc_count);
}

/* K8 -- 3d.cu:875-1071.  Back to the t = right - i convention with k(t) as
 * the LOW-side weight (3d.cu:960-999); pure second derivatives only. */
int cs3d_backward_backward_backward_cpu(const float *input, const float *grid, const float *gOut,
                                        const float *gOutGrid, const float *gOutgGrid, const float *offset,
                                        float *gInput, float *ggOut,
                                        int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int64_t P,
                                        int pad, int align_corners, int kernel, int multicell) {
    const size_t acc_count = (size_t)(N * C * D * H * W);
    acc_t *accum = acc_open(gInput, acc_count);
    if (!accum) return 1;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        for (int64_t p = 0; p < P; ++p) {
            const float *g = grid + (n * P + p) * 3;
            float m[3], ic[3];
            ic[0] = source_index(g[0], W, pad, align_corners, offset[n], multicell, &m[0]);
            ic[1] = source_index(g[1], H, pad, align_corners, offset[n], multicell, &m[1]);
            ic[2] = source_index(g[2], D, pad, align_corners, offset[n], multicell, &m[2]);
            int idx[3][2];
            float A[3][2][2]; /* [axis][side][0 weight, 1 second derivative] */
            for (int j = 0; j < 3; ++j) {
                idx[j][0] = (int)floorf(ic[j]);
                idx[j][1] = idx[j][0] + 1;
                float t = idx[j][1] - ic[j];
                float d2 = (kernel == CS_K_LINEAR) ? 0.0f : m[j] * m[j] * kern2(kernel, t);
                A[j][0][0] = kern0(kernel, t);   A[j][0][1] = d2;
                A[j][1][0] = 1.0f - A[j][0][0];  A[j][1][1] = -d2;
            }
            float od[8][3];
            for (int a = 0; a < 8; ++a) { /* 3d.cu:1003-1011 */
                const float *X = A[0][a & 1], *Y = A[1][(a >> 1) & 1], *Z = A[2][(a >> 2) & 1];
                od[a][0] = Y[0] * Z[0] * X[1];
                od[a][1] = X[0] * Z[0] * Y[1];
                od[a][2] = X[0] * Y[0] * Z[1];
            }
            const float *gg = gOutGrid + (n * P + p) * 3;
            const float *hg = gOutgGrid + (n * P + p) * 3;
            for (int64_t c = 0; c < C; ++c) {
                const float *in = input + (n * C + c) * D * H * W;
                acc_t *gi = accum + (n * C + c) * D * H * W;
                float go = gOut[(n * C + c) * P + p];
                float acc = 0.0f;
                for (int a = 0; a < 8; ++a) {
                    int x = idx[0][a & 1], y = idx[1][(a >> 1) & 1], z = idx[2][(a >> 2) & 1];
                    if (!inb3(z, y, x, D, H, W)) continue;
                    int64_t el = (z * H + y) * W + x;
                    float e = od[a][0] * hg[0] * gg[0] + od[a][1] * hg[1] * gg[1] + od[a][2] * hg[2] * gg[2];
                    acc += in[el] * e;   /* 3d.cu:1054 */
                    gi[el] += go * e;    /* 3d.cu:1063-1065 */
                }
                ggOut[(n * C + c) * P + p] = acc;
            }
        }
    }
    return acc_close(accum, gInput, acc_count);
}
