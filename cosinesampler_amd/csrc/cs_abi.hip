// cs_abi.hip -- extern "C" entry points of libcosine_sampler_hip.so (see include/cosine_sampler.h).
// Host side only: argument checks, stage -> kernel dispatch, launches on the caller's stream.
// Nothing here allocates, frees, copies to the host or synchronises.
#include <hip/hip_runtime.h>

#include "../../include/cosine_sampler.h"
#include "cs_kernels_direct.cuh"

namespace {

using cs::Dims;
using cs::Flags;

constexpr int kBlock = 256;

struct Problem {
    Dims d;
    Flags f;
    int dim;
    int kernel;
    hipStream_t stream;
    unsigned blocks;
};

int make_problem(Problem &pb, int dim, int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int64_t P,
                 int padding_mode, int align_corners, int kernel, int multicell, void *stream) {
    if (N < 0 || C < 0 || P < 0 || D < 1 || H < 1 || W < 1) return CS_ERR_INVALID;
    if (padding_mode < 0 || padding_mode > 2 || kernel < 0 || kernel > 2) return CS_ERR_INVALID;
    // node indices and sizes are kept in 32-bit registers; element offsets are 64-bit
    if (N > INT32_MAX || C > INT32_MAX || D > (1 << 28) || H > (1 << 28) || W > (1 << 28)) return CS_ERR_UNSUPPORTED;
    int64_t S = N * P;
    if ((S + kBlock - 1) / kBlock > (int64_t)INT32_MAX) return CS_ERR_UNSUPPORTED;
    pb.dim = dim;
    pb.kernel = kernel;
    pb.d.N = (int)N;
    pb.d.C = (int)C;
    pb.d.size[0] = (int)W;
    pb.d.size[1] = (int)H;
    pb.d.size[2] = (int)D;
    pb.d.P = P;
    pb.d.S = S;
    pb.d.vol = D * H * W;
    pb.f.pad = padding_mode;
    pb.f.align = align_corners ? 1 : 0;
    pb.f.multicell = multicell ? 1 : 0;
    pb.stream = (hipStream_t)stream;
    pb.blocks = (unsigned)((S + kBlock - 1) / kBlock);
    return CS_OK;
}

int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? CS_OK : (int)e;
}

int zero_async(float *p, int64_t elems, hipStream_t s) {
    if (!p || elems <= 0) return CS_OK;
    hipError_t e = hipMemsetAsync(p, 0, (size_t)elems * sizeof(float), s);
    return e == hipSuccess ? CS_OK : (int)e;
}

// kernel-enum dispatch: KERNEL is a template parameter so the unused derivative paths fold away
#define CS_DISPATCH_KERNEL(kernel_enum, ...)                                  \
    switch (kernel_enum) {                                                    \
        case CS_KERNEL_COSINE: { constexpr int KERNEL = cs::K_COSINE; __VA_ARGS__; } break;       \
        case CS_KERNEL_LINEAR: { constexpr int KERNEL = cs::K_LINEAR; __VA_ARGS__; } break;       \
        default:               { constexpr int KERNEL = cs::K_SMOOTHSTEP; __VA_ARGS__; } break;   \
    }

template <int DIM>
int run_forward(const Problem &pb, const float *input, const float *grid, const float *offset, float *output) {
    if (pb.d.S == 0 || pb.d.C == 0) return CS_OK;
    CS_DISPATCH_KERNEL(pb.kernel, (cs::direct_forward<DIM, KERNEL>
                                   <<<pb.blocks, kBlock, 0, pb.stream>>>(input, grid, offset, output, pb.d, pb.f)));
    return launch_status();
}

template <int DIM>
int run_backward(const Problem &pb, const float *gOut, const float *input, const float *grid, const float *offset,
                 float *grad_input, float *grad_grid) {
    int rc = zero_async(grad_input, (int64_t)pb.d.N * pb.d.C * pb.d.vol, pb.stream);
    if (rc) return rc;
    if (pb.d.S == 0) return CS_OK;
    if (pb.d.C == 0) return zero_async(grad_grid, pb.d.S * DIM, pb.stream);  // empty channel axis: d/dgrid = 0
    CS_DISPATCH_KERNEL(pb.kernel, (cs::direct_backward<DIM, KERNEL><<<pb.blocks, kBlock, 0, pb.stream>>>(
                                      gOut, input, grid, offset, grad_input, grad_grid, pb.d, pb.f)));
    return launch_status();
}

template <int DIM>
int run_bb(const Problem &pb, const float *cI, const float *cG, const float *input, const float *grid,
           const float *gOut, const float *offset, float *gInput, float *gGrid, float *ggOut) {
    int rc = zero_async(gInput, (int64_t)pb.d.N * pb.d.C * pb.d.vol, pb.stream);
    if (rc) return rc;
    if (pb.d.S == 0) return CS_OK;
    if (pb.d.C == 0) return zero_async(gGrid, pb.d.S * DIM, pb.stream);
    CS_DISPATCH_KERNEL(pb.kernel, (cs::direct_backward_backward<DIM, KERNEL><<<pb.blocks, kBlock, 0, pb.stream>>>(
                                      cI, cG, input, grid, gOut, offset, gInput, gGrid, ggOut, pb.d, pb.f)));
    return launch_status();
}

template <int DIM>
int run_bbb(const Problem &pb, const float *input, const float *grid, const float *gOut, const float *cG,
            const float *hG, const float *hO, const float *offset, float *gInput, float *ggOut) {
    int rc = zero_async(gInput, (int64_t)pb.d.N * pb.d.C * pb.d.vol, pb.stream);
    if (rc) return rc;
    if (pb.d.S == 0 || pb.d.C == 0) return CS_OK;
    CS_DISPATCH_KERNEL(pb.kernel, (cs::direct_bbb_fused<DIM, KERNEL><<<pb.blocks, kBlock, 0, pb.stream>>>(
                                      input, grid, gOut, cG, hG, hO, offset, gInput, ggOut, pb.d, pb.f)));
    return launch_status();
}

bool any_null(std::initializer_list<const void *> ps) {
    for (const void *p : ps)
        if (!p) return true;
    return false;
}

}  // namespace

extern "C" {

int cs_abi_version(void) { return CS_ABI_VERSION; }

const char *cs_error_string(int code) {
    switch (code) {
        case CS_OK: return "ok";
        case CS_ERR_INVALID: return "cosine_sampler: invalid argument (null pointer, negative size or unknown enum)";
        case CS_ERR_UNSUPPORTED: return "cosine_sampler: unsupported size or dtype";
        case CS_ERR_WORKSPACE: return "cosine_sampler: workspace missing or too small";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "cosine_sampler: unknown error";
    }
}

size_t cs_workspace_bytes(int dim, int stage, int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int64_t P) {
    (void)dim; (void)stage; (void)N; (void)C; (void)D; (void)H; (void)W; (void)P;
    return 0;  // the direct kernels need no scratch
}

#define CS_PROBLEM(dim, D)                                                                                        \
    Problem pb;                                                                                                   \
    (void)workspace; (void)workspace_bytes;                                                                       \
    {                                                                                                             \
        int rc_ = make_problem(pb, dim, N, C, D, H, W, P, padding_mode, align_corners, kernel, multicell, stream); \
        if (rc_) return rc_;                                                                                      \
    }

// zero-element tensors legitimately come with null data pointers
#define CS_NEED(...)                                                            \
    if (pb.d.S > 0 && pb.d.C > 0 && any_null({__VA_ARGS__})) return CS_ERR_INVALID;

// ---- 2D ----
int cs2d_forward(const float *input, const float *grid, const float *offset, float *output, int64_t N, int64_t C,
                 int64_t H, int64_t W, int64_t P, int padding_mode, int align_corners, int kernel, int multicell,
                 void *workspace, size_t workspace_bytes, void *stream) {
    CS_PROBLEM(2, 1)
    CS_NEED(input, grid, offset, output)
    return run_forward<2>(pb, input, grid, offset, output);
}

int cs2d_backward(const float *grad_output, const float *input, const float *grid, const float *offset,
                  float *grad_input, float *grad_grid, int64_t N, int64_t C, int64_t H, int64_t W, int64_t P,
                  int padding_mode, int align_corners, int kernel, int multicell, void *workspace,
                  size_t workspace_bytes, void *stream) {
    CS_PROBLEM(2, 1)
    CS_NEED(grad_output, input, grid, offset, grad_grid)
    return run_backward<2>(pb, grad_output, input, grid, offset, grad_input, grad_grid);
}

int cs2d_backward_backward(const float *grad_out_input, const float *grad_out_grid, const float *input,
                           const float *grid, const float *grad_output, const float *offset, float *grad_input,
                           float *grad_grid, float *grad_grad_out, int64_t N, int64_t C, int64_t H, int64_t W,
                           int64_t P, int padding_mode, int align_corners, int kernel, int multicell,
                           void *workspace, size_t workspace_bytes, void *stream) {
    CS_PROBLEM(2, 1)
    CS_NEED(input, grid, grad_output, offset, grad_input, grad_grid, grad_grad_out)
    return run_bb<2>(pb, grad_out_input, grad_out_grid, input, grid, grad_output, offset, grad_input, grad_grid,
                     grad_grad_out);
}

int cs2d_backward_backward_backward(const float *input, const float *grid, const float *grad_output,
                                    const float *grad_out_grid, const float *grad_out_ggrid, const float *offset,
                                    float *grad_input, float *grad_grad_out, int64_t N, int64_t C, int64_t H,
                                    int64_t W, int64_t P, int padding_mode, int align_corners, int kernel,
                                    int multicell, void *workspace, size_t workspace_bytes, void *stream) {
    CS_PROBLEM(2, 1)
    CS_NEED(input, grid, grad_output, grad_out_grid, grad_out_ggrid, offset, grad_input, grad_grad_out)
    return run_bbb<2>(pb, input, grid, grad_output, grad_out_grid, grad_out_ggrid, nullptr, offset, grad_input,
                      grad_grad_out);
}

int cs2d_bbb_fused(const float *input, const float *grid, const float *grad_output, const float *grad_out_grid,
                   const float *grad_out_ggrid, const float *grad_out_ggout, const float *offset, float *grad_input,
                   float *grad_grad_out, int64_t N, int64_t C, int64_t H, int64_t W, int64_t P, int padding_mode,
                   int align_corners, int kernel, int multicell, void *workspace, size_t workspace_bytes,
                   void *stream) {
    CS_PROBLEM(2, 1)
    CS_NEED(input, grid, grad_output, offset, grad_input, grad_grad_out)
    return run_bbb<2>(pb, input, grid, grad_output, grad_out_grid, grad_out_ggrid, grad_out_ggout, offset,
                      grad_input, grad_grad_out);
}

// ---- 3D ----
int cs3d_forward(const float *input, const float *grid, const float *offset, float *output, int64_t N, int64_t C,
                 int64_t D, int64_t H, int64_t W, int64_t P, int padding_mode, int align_corners, int kernel,
                 int multicell, void *workspace, size_t workspace_bytes, void *stream) {
    CS_PROBLEM(3, D)
    CS_NEED(input, grid, offset, output)
    return run_forward<3>(pb, input, grid, offset, output);
}

int cs3d_backward(const float *grad_output, const float *input, const float *grid, const float *offset,
                  float *grad_input, float *grad_grid, int64_t N, int64_t C, int64_t D, int64_t H, int64_t W,
                  int64_t P, int padding_mode, int align_corners, int kernel, int multicell, void *workspace,
                  size_t workspace_bytes, void *stream) {
    CS_PROBLEM(3, D)
    CS_NEED(grad_output, input, grid, offset, grad_grid)
    return run_backward<3>(pb, grad_output, input, grid, offset, grad_input, grad_grid);
}

int cs3d_backward_backward(const float *grad_out_input, const float *grad_out_grid, const float *input,
                           const float *grid, const float *grad_output, const float *offset, float *grad_input,
                           float *grad_grid, float *grad_grad_out, int64_t N, int64_t C, int64_t D, int64_t H,
                           int64_t W, int64_t P, int padding_mode, int align_corners, int kernel, int multicell,
                           void *workspace, size_t workspace_bytes, void *stream) {
    CS_PROBLEM(3, D)
    CS_NEED(input, grid, grad_output, offset, grad_input, grad_grid, grad_grad_out)
    return run_bb<3>(pb, grad_out_input, grad_out_grid, input, grid, grad_output, offset, grad_input, grad_grid,
                     grad_grad_out);
}

int cs3d_backward_backward_backward(const float *input, const float *grid, const float *grad_output,
                                    const float *grad_out_grid, const float *grad_out_ggrid, const float *offset,
                                    float *grad_input, float *grad_grad_out, int64_t N, int64_t C, int64_t D,
                                    int64_t H, int64_t W, int64_t P, int padding_mode, int align_corners,
                                    int kernel, int multicell, void *workspace, size_t workspace_bytes,
                                    void *stream) {
    CS_PROBLEM(3, D)
    CS_NEED(input, grid, grad_output, grad_out_grid, grad_out_ggrid, offset, grad_input, grad_grad_out)
    return run_bbb<3>(pb, input, grid, grad_output, grad_out_grid, grad_out_ggrid, nullptr, offset, grad_input,
                      grad_grad_out);
}

int cs3d_bbb_fused(const float *input, const float *grid, const float *grad_output, const float *grad_out_grid,
                   const float *grad_out_ggrid, const float *grad_out_ggout, const float *offset, float *grad_input,
                   float *grad_grad_out, int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int64_t P,
                   int padding_mode, int align_corners, int kernel, int multicell, void *workspace,
                   size_t workspace_bytes, void *stream) {
    CS_PROBLEM(3, D)
    CS_NEED(input, grid, grad_output, offset, grad_input, grad_grad_out)
    return run_bbb<3>(pb, input, grid, grad_output, grad_out_grid, grad_out_ggrid, grad_out_ggout, offset,
                      grad_input, grad_grad_out);
}

}  // extern "C"
