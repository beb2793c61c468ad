// cs_abi.hip -- extern "C" entry points of libcosine_sampler_hip.so (see include/cosine_sampler.h).
// Host side only: argument checks, path choice, workspace carving, launches on the caller's stream.
// Nothing here allocates, frees, copies to the host or synchronises.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "../../include/cosine_sampler.h"
#include "cs_kernels_direct.cuh"
#include "cs_points_cl.cuh"
#include "cs_tiled.cuh"
#include "cs_dense3d.cuh"
#include "cs_units.h"

namespace {

using cs::Dims;
using cs::Flags;
namespace tl = cs::tiled;

constexpr int kBlock = 256;
std::atomic<int> g_force_path{0};  // cs_debug_force_path
// The testing knobs act only in a process that asked for them: COSINESAMPLER_DEBUG=1 in the environment, read once.
bool debug_enabled() {
    static const bool on = [] {
        const char *e = std::getenv("COSINESAMPLER_DEBUG");
        return e && std::strcmp(e, "1") == 0;
    }();
    return on;
}

struct Problem {
    Dims d;
    Flags f;
    int dim;
    int kernel;
    int sdt;      // element type of the channel-major streams: 0 fp32, 1 half, 2 bfloat16 (CS_STREAM_*)
    bool coherent;   // CS_POINTS_COHERENT: the caller says consecutive points share cells
    bool sum_n;      // CS_SUM_OVER_N: shared points and cotangents, per-point results summed over the tables
    int acc_kind;    // cs_cotangent_layout.accumulate_grad_input: grad_input is the caller's step accumulator (CS_ACC_*)
    hipStream_t stream;
    unsigned blocks;
};

int make_problem(Problem &pb, int dim, int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int64_t P,
                 int padding_mode, int align_corners, int kernel, int multicell, void *stream) {
    if (N < 0 || C < 0 || P < 0 || D < 1 || H < 1 || W < 1) return CS_ERR_INVALID;
    const int exact = (kernel & CS_KERNEL_EXACT_MIXED) ? 1 : 0;
    if ((kernel & CS_STREAM_F16) && (kernel & CS_STREAM_BF16)) return CS_ERR_INVALID;
    pb.sdt = (kernel & CS_STREAM_F16) ? 1 : (kernel & CS_STREAM_BF16) ? 2 : 0;
    const bool grid_bc = (kernel & CS_GRID_BROADCAST) != 0;
    pb.coherent = (kernel & CS_POINTS_COHERENT) != 0;
    pb.sum_n = (kernel & CS_SUM_OVER_N) != 0;
    pb.acc_kind = CS_ACC_NONE;
    if (pb.sum_n && (!grid_bc || exact)) return CS_ERR_UNSUPPORTED;     // one set of points for every table; the reference's derivatives
    kernel &= ~(CS_KERNEL_EXACT_MIXED | CS_STREAM_F16 | CS_STREAM_BF16 | CS_GRID_BROADCAST | CS_POINTS_COHERENT | CS_SUM_OVER_N);
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
    pb.d.tab_ns = 1;          // gathers read the caller's NC[D]HW tensor unless a stage switches to a channels-last copy
    pb.d.tab_cs = pb.d.vol;
    pb.d.go_ns = pb.d.ho_ns = pb.d.out_ns = C * P;   // contiguous streams unless the entry point is given a layout
    pb.d.grid_ns = grid_bc ? 0 : P;
    pb.d.nsum = 0;            // set by use_nsum3() where a 3D stage runs in the summing mode
    pb.d.xcd = 0;             // set by prepare() for the tiled 2D backward point kernels
    pb.f.pad = padding_mode;
    pb.f.align = align_corners ? 1 : 0;
    pb.f.multicell = multicell ? 1 : 0;
    pb.f.exact = exact;
    pb.f.pair16 = 0;
    pb.stream = (hipStream_t)stream;
    pb.blocks = (unsigned)((S + kBlock - 1) / kBlock);
    return CS_OK;
}

int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? CS_OK : (int)e;
}

// A kernel, not hipMemsetAsync: in round 1 (ROCm 7.2, HIP runtime 70226015) a memset node inside a captured stage was
// seen to be skipped on a replay (tests/test_parity_gpu.py::test_stages_can_be_captured_in_a_hip_graph); kernel nodes
// replay faithfully.  The minimal form of it -- memset + one kernel captured, replayed 8 times at 1 / 64 / 512 MiB
// (tools/memset_graph_repro.hip, profiles/round2_memset_graph_repro.txt) -- does NOT reproduce the skip on the same
// runtime, so the cause may have been the capture of a memset between kernels of this library rather than memset nodes
// as such; the kernel costs the same (11 us per 64 MiB) and stays until that is understood.
__global__ __launch_bounds__(256) void zero_fill(float *__restrict__ p, int64_t elems) {
    const int64_t stride = (int64_t)gridDim.x * 256 * 4;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < elems; i += stride) {
        if (i + 4 <= elems && ((uintptr_t)(p + i) & 15) == 0) {
            *reinterpret_cast<float4 *>(p + i) = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            for (int64_t k = i; k < elems && k < i + 4; ++k) p[k] = 0.f;
        }
    }
}
int zero_async(float *p, int64_t elems, hipStream_t s) {
    if (!p || elems <= 0) return CS_OK;
    const int64_t blocks = std::min<int64_t>((elems / 4 + 255) / 256 + 1, 256 * 32);
    zero_fill<<<(unsigned)blocks, 256, 0, s>>>(p, elems);
    return launch_status();
}

// channels-last scratch (N,vol,CP) -> the caller's (N,C,vol)
int unpack_cl(const float *acc, float *out, int64_t N, int64_t C, int64_t CP, int64_t vol, hipStream_t s) {
    if ((vol & 3) == 0 && (CP & 3) == 0 && CP <= 64 && (((uintptr_t)acc | (uintptr_t)out) & 15) == 0) {
        const int nv = cs::cl4_nv((int)CP);
        dim3 g((unsigned)((vol + nv - 1) / nv), (unsigned)N);
        cs::unpack_cl4<<<g, 256, cs::cl4_lds((int)CP), s>>>(acc, out, (int)C, (int)CP, vol);
        return launch_status();
    }
    const int nv = cs::unpack_nv((int)C);
    dim3 g((unsigned)((vol + nv - 1) / nv), (unsigned)N);
    cs::unpack_channels_last<<<g, 256, (size_t)nv * (C + 1) * 4, s>>>(acc, out, (int)C, (int)CP, vol);
    return launch_status();
}

// channel count of a channels-last copy: 1, 2, 4, 8 or 16 float4 quads -- the counts the fast kernels are built for.
// Other channel counts run zero-padded up to the next one (C = 1..3 as 4, 5..7 as 8, 9..15 as 16, 17..31 as 32): the
// reference takes any C in one loop (2d.cu:340-354), and the direct kernels' scattered atomics are 30-100x slower.
int64_t cpad(int64_t C) {
    int64_t q = 1;
    while (q * 4 < C) q *= 2;
    return q * 4;
}

// channel-count dispatch of the fast paths: CQ = cpad(C)/4 in {1, 2, 4}
#define CS_DISPATCH_CQ(C_, ...)                                               \
    switch (cpad(C_) / 4) {                                                   \
        case 1:  { constexpr int CQ = 1; __VA_ARGS__; } break;                \
        case 2:  { constexpr int CQ = 2; __VA_ARGS__; } break;                \
        default: { constexpr int CQ = 4; __VA_ARGS__; } break;                \
    }

// the 2D tiled path also takes 32 channels (8 quads); the 3D channels-last kernels stop at 16
#define CS_DISPATCH_CQT(C_, ...)                                              \
    switch (cpad(C_) / 4) {                                                   \
        case 1:  { constexpr int CQ = 1; __VA_ARGS__; } break;                \
        case 2:  { constexpr int CQ = 2; __VA_ARGS__; } break;                \
        case 4:  { constexpr int CQ = 4; __VA_ARGS__; } break;                \
        default: { constexpr int CQ = 8; __VA_ARGS__; } break;                \
    }
// dynamic LDS beyond the default 64 KiB limit needs an opt-in per kernel (idempotent, no device work)
template <typename K>
int allow_lds(K kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return CS_OK;
    hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return e == hipSuccess ? CS_OK : (int)e;
}

// kernel-enum dispatch: KERNEL is a template parameter so the unused derivative paths fold away; ST is the element type
// of the channel-major streams (`pb.sdt`, the fast paths only: kernels that do not take it are the same symbol thrice)
#define CS_DISPATCH_KERNEL_(kernel_enum, ...)                                 \
    switch (kernel_enum) {                                                    \
        case CS_KERNEL_COSINE: { constexpr int KERNEL = cs::K_COSINE; __VA_ARGS__; } break;       \
        case CS_KERNEL_LINEAR: { constexpr int KERNEL = cs::K_LINEAR; __VA_ARGS__; } break;       \
        default:               { constexpr int KERNEL = cs::K_SMOOTHSTEP; __VA_ARGS__; } break;   \
    }
#define CS_DISPATCH_KERNEL(kernel_enum, ...)                                                              \
    switch (pb.sdt) {                                                                                     \
        case 1:  { using ST = cs::stream_f16; (void)sizeof(ST); CS_DISPATCH_KERNEL_(kernel_enum, __VA_ARGS__) } break;   \
        case 2:  { using ST = cs::stream_bf16; (void)sizeof(ST); CS_DISPATCH_KERNEL_(kernel_enum, __VA_ARGS__) } break;  \
        default: { using ST = float; (void)sizeof(ST); CS_DISPATCH_KERNEL_(kernel_enum, __VA_ARGS__) } break;             \
    }

// ------------------------------------------------------------------------------------------------
// direct path (any shape)
// ------------------------------------------------------------------------------------------------
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
    // gInput == nullptr: p-ordered outputs only (the caller scatters separately)
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

// ------------------------------------------------------------------------------------------------
// row-atomic scatter (any dim, C a power of two <= 64): p-ordered outputs by the direct kernels with
// their scatter switched off, input-shaped gradient by row_scatter into a channels-last scratch
// ------------------------------------------------------------------------------------------------
int log2_exact(int64_t C) {
    for (int l = 0; l <= 6; ++l)
        if (C == ((int64_t)1 << l)) return l;
    return -1;
}

bool rows_applies(int64_t N, int64_t C, int64_t P, int64_t vol) {
    int mode = g_force_path.load(std::memory_order_relaxed);
    if (mode == 1) return false;
    if (log2_exact(C) < 1) return false;                       // C = 1: rows are single floats, nothing to gain
    if (N * P * C >= ((int64_t)1 << 40) || vol * C >= ((int64_t)1 << 31)) return false;
    if (N > 65535) return false;                               // unpack_channels_last launches with gridDim.y = N
    if (mode >= 2) return true;
    return N * P >= (1 << 16);
}

template <int DIM, int MODE>
int row_scatter_into(const Problem &pb, const float *grid, const float *offset, const float *gOut, const float *cG,
                     const float *hG, const float *hO, float *out_grad, float *acc);

template <int DIM, int MODE>
int run_row_scatter(const Problem &pb, const float *grid, const float *offset, const float *gOut, const float *cG,
                    const float *hG, const float *hO, float *out_grad, void *workspace, size_t workspace_bytes) {
    const int64_t T = (int64_t)pb.d.N * pb.d.C * pb.d.vol;
    if (!workspace || workspace_bytes < (size_t)T * 4) return CS_ERR_WORKSPACE;
    return row_scatter_into<DIM, MODE>(pb, grid, offset, gOut, cG, hG, hO, out_grad, (float *)workspace);
}

template <int DIM, int MODE>
int row_scatter_into(const Problem &pb, const float *grid, const float *offset, const float *gOut, const float *cG,
                     const float *hG, const float *hO, float *out_grad, float *acc) {
    const int64_t T = (int64_t)pb.d.N * pb.d.C * pb.d.vol;
    int rc = zero_async(acc, T, pb.stream);
    if (rc) return rc;
    const int logC = log2_exact(pb.d.C);
    if (logC < 0 || pb.d.N > 65535) return CS_ERR_UNSUPPORTED;   // channel masks are shifts; gridDim.y = N in the unpack
    const bool pair = pb.d.C <= 8;   // node rows under 64 bytes go in x-neighbour pairs (same request rate, half the requests)
    const int64_t lanes = pb.d.S << (logC + (pair ? 1 : 0));
    if ((lanes + kBlock - 1) / kBlock > (int64_t)INT32_MAX) return CS_ERR_UNSUPPORTED;
    const unsigned nb = (unsigned)((lanes + kBlock - 1) / kBlock);
    if (pair) {
        CS_DISPATCH_KERNEL(pb.kernel, (cs::row_scatter<DIM, KERNEL, MODE, true><<<nb, kBlock, 0, pb.stream>>>(
                                          grid, offset, gOut, cG, hG, hO, acc, pb.d, pb.f, logC)));
    } else {
        CS_DISPATCH_KERNEL(pb.kernel, (cs::row_scatter<DIM, KERNEL, MODE, false><<<nb, kBlock, 0, pb.stream>>>(
                                          grid, offset, gOut, cG, hG, hO, acc, pb.d, pb.f, logC)));
    }
    rc = launch_status();
    if (rc) return rc;
    return unpack_cl(acc, out_grad, pb.d.N, pb.d.C, pb.d.C, pb.d.vol, pb.stream);
}

// ------------------------------------------------------------------------------------------------
// tiled path (2D, any C <= 32: run zero-padded to 4, 8, 16 or 32 channels, cpad)
// ------------------------------------------------------------------------------------------------
constexpr int64_t kTiledMinSamples = 1 << 16;  // below this the launch count matters more than atomics

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

bool tiled_applies(int dim, int64_t N, int64_t C, int64_t H, int64_t W, int64_t P) {
    int mode = g_force_path.load(std::memory_order_relaxed);
    if (mode == 1 || dim != 2) return false;
    if (C > 32) return false;                             // other counts run zero-padded to 4, 8, 16 or 32 (cpad)
    int64_t S = N * P;
    if (S <= 0 || S >= (int64_t)0xFFFFFFF0ll) return false;
    int64_t ntx = (W + 1 + tl::TX - 1) / tl::TX, nty = (H + 1 + tl::TY - 1) / tl::TY;
    if (ntx * nty > 12288) return false;                 // tile histogram lives in LDS (48 KiB)
    if (N * ntx * nty >= (int64_t)INT32_MAX) return false;
    if (N > 65535 || H * W * C >= ((int64_t)1 << 31)) return false;   // gridDim.y = N; 32-bit node offsets
    if (mode >= 2) return true;
    return S >= kTiledMinSamples;
}

constexpr int64_t PLAN_WGS = 1024;   // 2048 and 4096 measured the same
// samples per plan workgroup: enough workgroups to balance the chip (the plan kernels are bound by LDS atomics per
// CU), few enough that the per-chunk histograms (chunks x bins words) stay small
int plan_chunk(int64_t N, int64_t P, int64_t bins) {
    if (bins > 12288) return 4 * tl::CHUNK;
    int64_t want = (N * P + PLAN_WGS - 1) / PLAN_WGS;            // samples per workgroup for ~PLAN_WGS workgroups
    want = (want + 255) / 256 * 256;
    return (int)std::min<int64_t>(std::max<int64_t>(want, 1024), tl::CHUNK);
}

struct PlanLayout {
    int ntx, nty, ntiles, chunks, chunk, dense;
    size_t off_sorted, off_key, off_tile_begin, off_cell_begin, off_block_hist, off_totals, off_bsum, off_G, off_cellb, bytes;
};

// Crowded tables (the reference's own test shapes: 96 tables of 16x16 cells, 10^5 points): the plan bins by cell
// directly and cell_scatter gives every (n, cell) bucket a wave.  Needs the cell histogram in LDS (48 KiB) and
// pays off from about two waves' worth of samples per cell: measured at N=16 C=16 P=2^20, stages + plan,
// 32^2 cells 4.75 vs 7.47 ms, 64^2 4.70 vs 5.24, 96^2 4.98 vs 4.85 (the plan's per-chunk histograms grow with the cells).
bool dense_applies(int64_t N, int64_t H, int64_t W, int64_t P) {
    if (g_force_path.load(std::memory_order_relaxed) == 3) return false;   // testing: tile walkers only
    const int64_t cells = (W + 1) * (H + 1);
    return cells <= 12288 && P >= 128 * cells && N * cells < (int64_t)INT32_MAX;
}

PlanLayout plan_layout(int64_t N, int64_t C, int64_t H, int64_t W, int64_t P) {
    PlanLayout L;
    L.dense = dense_applies(N, H, W, P) ? 1 : 0;
    L.ntx = L.dense ? (int)(W + 1) : (int)((W + 1 + tl::TX - 1) / tl::TX);
    L.nty = L.dense ? (int)(H + 1) : (int)((H + 1 + tl::TY - 1) / tl::TY);
    L.ntiles = L.ntx * L.nty;
    L.chunk = plan_chunk(N, P, L.ntiles);
    L.chunks = (int)((P + L.chunk - 1) / L.chunk);
    int64_t S = N * P;
    size_t o = 0;
    L.off_sorted = o;     o += align256((size_t)S * 4);
    L.off_key = o;        o += L.dense ? 0 : align256((size_t)S * 4);
    L.off_tile_begin = o; o += align256(((size_t)N * L.ntiles + 1) * 4);
    L.off_cell_begin = o; o += L.dense ? 0 : align256((size_t)N * L.ntiles * (tl::CELLS + 1) * 4);
    L.off_block_hist = o; o += align256((size_t)N * L.chunks * L.ntiles * 4);
    L.off_totals = o;     o += align256((size_t)N * L.ntiles * 4);
    L.off_bsum = o;       o += align256(((size_t)N * L.ntiles / 1024 + 3) * 4);
    L.off_G = o;          o += L.dense ? 0 : align256((size_t)S * cpad(C) * 4);   // Plan::Gs (walker plans only)
    L.off_cellb = o;      o += (L.dense || P <= ((int64_t)1 << 24)) ? 0 : align256((size_t)S);   // keys no longer hold the cell
    L.bytes = o;
    return L;
}

// exclusive scan of `totals` (nt bucket sizes) into tile_begin[0..nt]
int scan_buckets(const uint32_t *totals, uint32_t *tile_begin, uint32_t *bsum, int64_t nt, hipStream_t s) {
    const int64_t nb = nt / 1024 + 1;   // covers index nt itself, where the grand total goes
    tl::plan_scan_tiles_local<<<(unsigned)nb, 1024, 0, s>>>(totals, tile_begin, bsum, nt);
    tl::plan_scan_tiles_sums<<<1, 1024, 0, s>>>(bsum, nb);
    tl::plan_scan_tiles_add<<<(unsigned)nb, 1024, 0, s>>>(tile_begin, bsum, nt, nb);
    return launch_status();
}

tl::Plan plan_view(const PlanLayout &L, void *blob) {
    char *b = (char *)blob;
    tl::Plan p;
    p.sorted = (uint32_t *)(b + L.off_sorted);
    p.key = (uint32_t *)(b + L.off_key);
    p.tile_begin = (uint32_t *)(b + L.off_tile_begin);
    p.cell_begin = (uint32_t *)(b + L.off_cell_begin);
    p.block_hist = (uint32_t *)(b + L.off_block_hist);
    p.Gs = L.dense ? nullptr : (float *)(b + L.off_G);
    p.cellb = (L.off_cellb == L.bytes || L.dense) ? nullptr : (uint8_t *)(b + L.off_cellb);
    p.ntx = L.ntx;
    p.nty = L.nty;
    p.ntiles = L.ntiles;
    p.chunks = L.chunks;
    p.chunk = L.chunk;
    p.dense = L.dense;
    return p;
}

int build_plan(const Problem &pb, const float *grid, const float *offset, void *blob) {
    PlanLayout L = plan_layout(pb.d.N, pb.d.C, pb.d.size[1], pb.d.size[0], pb.d.P);
    tl::Plan pl = plan_view(L, blob);
    uint32_t *totals = (uint32_t *)((char *)blob + L.off_totals);
    dim3 g((unsigned)L.chunks, (unsigned)pb.d.N);
    size_t shm = (size_t)L.ntiles * 4;
    tl::plan_count<<<g, 256, shm, pb.stream>>>(grid, offset, pl, pb.d, pb.f);
    int64_t nt = (int64_t)pb.d.N * L.ntiles;
    tl::plan_scan_chunks<<<(unsigned)((nt + 255) / 256), 256, 0, pb.stream>>>(pl, pb.d.N, totals);
    scan_buckets(totals, pl.tile_begin, (uint32_t *)((char *)blob + L.off_bsum), nt, pb.stream);
    tl::plan_scatter<<<g, 256, shm, pb.stream>>>(grid, offset, pl, pb.d, pb.f);
    if (!L.dense) tl::plan_tile_sort<<<(unsigned)nt, 256, 0, pb.stream>>>(pl, pb.d.P);
    return launch_status();
}

// plane > 0: the z-paired layout of the 3D tables (two rows per node, see cs::pack_cl4), plane = H * W
int pack_cl(const float *in, float *out, int64_t N, int64_t C, int64_t vol, hipStream_t s, int64_t plane = 0) {
    if (N == 0 || C == 0 || vol == 0) return CS_OK;
    const int64_t CP = cpad(C);
    const int slots = plane > 0 ? 2 : 1;
    if ((vol & 3) == 0 && (plane & 3) == 0 && CP <= 64 && (((uintptr_t)in | (uintptr_t)out) & 15) == 0) {   // 16-byte accesses on both sides
        const int nv = cs::cl4_nv((int)CP) / slots;                 // both slots of a node leave from one workgroup
        const size_t shm = (size_t)slots * CP * (nv + 4) * 4;
        if (slots == 2 && plane % nv == 0 && vol % plane == 0 && vol / plane <= INT32_MAX && g_force_path.load(std::memory_order_relaxed) != 6) {
            // z-paired and the planes divide into whole node ranges: the column-wise pack reads the table once
            const int64_t D = vol / plane;
            const int ZS = 16;
            dim3 gz((unsigned)(plane / nv), (unsigned)((D + ZS - 1) / ZS), (unsigned)N);
            if (gz.y <= 65535 && gz.z <= 65535) {
                cs::pack_cl4_zcol<<<gz, 256, shm, s>>>(in, out, (int)C, (int)CP, plane, (int)D, ZS);
                return launch_status();
            }
        }
        dim3 g((unsigned)((vol + nv - 1) / nv), (unsigned)N);
        cs::pack_cl4<<<g, 256, shm, s>>>(in, out, (int)C, (int)CP, vol, plane, slots);
        return launch_status();
    }
    dim3 g((unsigned)((vol + 63) / 64), (unsigned)N, (unsigned)slots);
    tl::pack_channels_last<<<g, 256, (size_t)CP * 65 * 4, s>>>(in, out, (int)C, (int)CP, vol, plane, slots);
    return launch_status();
}
// floats of a channels-last table copy: 3D tables are z-paired
size_t table_floats(int dim, int64_t N, int64_t C, int64_t vol) { return (size_t)N * cpad(C) * vol * (dim == 3 ? 2 : 1); }

// carve the workspace in the order cs_workspace_bytes adds it up: [input_cl?][plan?][gOutInput channels-last?][fat rows | accumulator]
struct Carve {
    char *base;
    size_t used, cap;
    void *take(size_t bytes) {
        void *p = base + used;
        used += align256(bytes);
        return p;
    }
    bool ok() const { return used <= cap; }
};

size_t tiled_workspace(int stage, int64_t N, int64_t C, int64_t H, int64_t W, int64_t P, int have_cl, int have_plan,
                       int have_cI, bool coherent, bool accumulate) {
    const int64_t CP = cpad(C);
    size_t T = align256((size_t)N * CP * H * W * 4);
    size_t S = (size_t)N * P;
    size_t need = 0;
    if (!have_cl) need += T;
    if (stage == CS_STAGE_FORWARD) return need;
    if (coherent && !(stage == CS_STAGE_BACKWARD_BACKWARD && have_cI))   // no plan, no records: the channels-last accumulator is all
        return need + (accumulate ? 0 : T);                              // (... and with a step accumulator it is the caller's)
    if (!have_plan) need += plan_layout(N, C, H, W, P).bytes;
    if (stage == CS_STAGE_BACKWARD_BACKWARD && have_cI) need += T;
    need += align256(S * (size_t)(stage == CS_STAGE_BBB_FUSED ? tl::row2((int)CP) : tl::row1((int)CP)) * 4);   // fat rows
    return need;
}

// SRC / EMIT: see tl::tile_scatter.  Crowded tables (wave per cell) know rows with everything in them only.
template <int SRC, bool EMIT = false>
int launch_tile_scatter(const Problem &pb, const tl::Plan &pl, const float *fat, float *grad_input) {
    if (pl.dense) {   // one wave per (n, cell) bucket
        static_assert(SRC <= 1 || true, "");
        if (SRC > 1) return CS_ERR_INVALID;
        unsigned nbk = (unsigned)(((int64_t)pb.d.N * pl.ntiles + 3) / 4);
        CS_DISPATCH_CQT(pb.d.C, (tl::cell_scatter<CQ, SRC == 1><<<nbk, 256, 0, pb.stream>>>(fat, pl, grad_input, pb.d)));
        return launch_status();
    }
    unsigned nb = (unsigned)((int64_t)pb.d.N * pl.ntiles);
    int rc = CS_OK;
    CS_DISPATCH_CQT(pb.d.C, {
        constexpr size_t shm = tl::tile_scatter_lds<CQ>();
        rc = allow_lds(tl::tile_scatter<CQ, SRC, EMIT>, shm);
        if (!rc) tl::tile_scatter<CQ, SRC, EMIT><<<nb, 256, shm, pb.stream>>>(fat, pl, grad_input, pb.d);
    });
    return rc ? rc : launch_status();
}

struct Prepared {
    const float *icl;
    tl::Plan plan;
    bool plan_is_callers;   // it outlives this call: worth leaving the sorted gOut copy in it
    Dims d;                 // the problem as the backward point kernels get it (Dims::xcd decided here)
};

// resolve input_cl / plan: use the caller's, or build into the workspace
int prepare(const Problem &pb, int stage, const float *input, const float *grid, const float *offset,
            const float *input_cl, const void *plan, Carve &ws, Prepared &out) {
    const int64_t T = (int64_t)pb.d.N * cpad(pb.d.C) * pb.d.vol;
    if (input_cl) {
        out.icl = input_cl;
    } else {
        float *buf = (float *)ws.take((size_t)T * 4);
        if (!ws.ok()) return CS_ERR_WORKSPACE;
        int rc = pack_cl(input, buf, pb.d.N, pb.d.C, pb.d.vol, pb.stream);
        if (rc) return rc;
        out.icl = buf;
    }
    out.plan_is_callers = false;
    out.d = pb.d;
    if (stage == CS_STAGE_FORWARD) return CS_OK;
    PlanLayout L = plan_layout(pb.d.N, pb.d.C, pb.d.size[1], pb.d.size[0], pb.d.P);
    if (plan) {
        out.plan = plan_view(L, const_cast<void *>(plan));
        out.plan_is_callers = g_force_path.load(std::memory_order_relaxed) != 4;   // 4: testing, no payload re-use
    } else {
        void *blob = ws.take(L.bytes);
        if (!ws.ok()) return CS_ERR_WORKSPACE;
        int rc = build_plan(pb, grid, offset, blob);
        if (rc) return rc;
        out.plan = plan_view(L, blob);
    }
    // the XCD-aware workgroup order of the backward point kernels (cs_tiled.cuh pblk): where it was measured to pay -- fp32
    // streams, tile walkers (not the crowded-table path); the kernels fall back by themselves when N does not divide
    out.d.xcd = (pb.sdt == 0 && !out.plan.dense) ? 1 : 0;
    return CS_OK;
}

dim3 point_grid(const Problem &pb) { return dim3((unsigned)((pb.d.P + kBlock - 1) / kBlock), (unsigned)pb.d.N); }

// three-phase point kernels: per wave 64 fat rows + the node/result record (+ the coefficient record of point_bb)
size_t q_lds(int stride, int co_fields) { return (size_t)4 * (64 * stride + tl::QREC + co_fields * 64) * 4; }
// point_forward, per wave: one geometry record block + the [C][64] result tile
size_t point_lds(int C) { return (size_t)4 * (tl::REC_FLOATS + C * tl::OUT_LD) * 4; }

cs::coh::Launch coh_launch(const Problem &pb);
bool coherent_applies(const Problem &pb);
int sum_n_check(const Problem &pb, bool has_gOut, bool has_hO);

int tiled_forward(const Problem &pb, const float *input, const float *grid, const float *offset, float *output,
                  const float *input_cl, void *workspace, size_t workspace_bytes) {
    Carve ws{(char *)workspace, 0, workspace ? workspace_bytes : 0};
    Prepared pr;
    int rc = sum_n_check(pb, false, false);
    if (rc) return rc;
    rc = prepare(pb, CS_STAGE_FORWARD, input, grid, offset, input_cl, nullptr, ws, pr);
    if (rc) return rc;
    if (coherent_applies(pb)) return cs::coh::forward(coh_launch(pb), pr.icl, grid, offset, output);
    CS_DISPATCH_KERNEL(pb.kernel, CS_DISPATCH_CQT(pb.d.C, (tl::point_forward<KERNEL, CQ, ST><<<point_grid(pb), kBlock, point_lds((int)cpad(pb.d.C)), pb.stream>>>(
                                      pr.icl, grid, offset, (ST *)output, pb.d, pb.f))));
    return launch_status();
}

// the coherent-points kernels (cs_coherent.cuh) for this problem
cs::coh::Launch coh_launch(const Problem &pb) {
    cs::coh::Launch L;
    L.d = pb.d;
    L.f = pb.f;
    L.kernel = pb.kernel;
    L.sdt = pb.sdt;
    L.cq = (int)(cpad(pb.d.C) / 4);
    L.stream = pb.stream;
    L.nsum = pb.sum_n;
    return L;
}
// CS_SUM_OVER_N lives on the coherent kernels alone (whatever the order of the points: only their speed depends on it)
bool coherent_applies(const Problem &pb) {
    if (pb.sum_n) return true;
    return pb.coherent && g_force_path.load(std::memory_order_relaxed) != 5 && cs::coh::supported(coh_launch(pb));
}
// ... for the shapes and flags the summing kernels are built for; the shared cotangents have no n-stride
int sum_n_check(const Problem &pb, bool has_gOut, bool has_hO) {
    if (!pb.sum_n) return CS_OK;
    if (!cs::coh::supported_nsum(coh_launch(pb))) return CS_ERR_UNSUPPORTED;
    if ((has_gOut && pb.d.go_ns != 0) || (has_hO && pb.d.ho_ns != 0)) return CS_ERR_INVALID;
    return CS_OK;
}
// the zeroed channels-last accumulator the coherent kernels add into
// (cs_cotangent_layout.accumulate_grad_input: the caller's `grad_input` IS that accumulator, already holding the sums of
// the step's earlier stages -- nothing to clear here and nothing to unpack afterwards)
int coherent_accumulator(const Problem &pb, Carve &ws, float *grad_input, float *&acc) {
    if (pb.acc_kind == CS_ACC_NCHW) return CS_ERR_UNSUPPORTED;
    if (pb.acc_kind == CS_ACC_CHANNELS_LAST) {
        acc = grad_input;
        return CS_OK;
    }
    const int64_t T = (int64_t)pb.d.N * cpad(pb.d.C) * pb.d.vol;
    acc = (float *)ws.take((size_t)T * 4);
    if (!ws.ok()) return CS_ERR_WORKSPACE;
    return zero_async(acc, T, pb.stream);
}
int coherent_finish(const Problem &pb, const float *acc, float *grad_input) {
    if (pb.acc_kind) return CS_OK;
    return unpack_cl(acc, grad_input, pb.d.N, pb.d.C, cpad(pb.d.C), pb.d.vol, pb.stream);
}
// the walkers / the wave-per-cell kernel add their sums to NC[D]HW grad_input with float atomics: a step accumulator of
// that layout is simply not cleared
int walker_target(const Problem &pb, float *grad_input) {
    if (pb.acc_kind == CS_ACC_CHANNELS_LAST) return CS_ERR_UNSUPPORTED;
    if (pb.acc_kind == CS_ACC_NCHW) return CS_OK;
    return zero_async(grad_input, (int64_t)pb.d.N * pb.d.C * pb.d.vol, pb.stream);
}

int tiled_backward(const Problem &pb, const float *gOut, const float *input, const float *grid, const float *offset,
                   float *grad_input, float *grad_grid, const float *input_cl, const void *plan, void *workspace,
                   size_t workspace_bytes, bool g_leave) {
    Carve ws{(char *)workspace, 0, workspace ? workspace_bytes : 0};
    Prepared pr;
    // without grad_input nothing is scattered: no plan, no fat rows -- the point kernel gathers and is all there is
    const bool coh = coherent_applies(pb);
    int rc = sum_n_check(pb, true, false);
    if (rc) return rc;
    rc = prepare(pb, grad_input && !coh ? CS_STAGE_BACKWARD : CS_STAGE_FORWARD, input, grid, offset, input_cl, plan, ws, pr);
    if (rc) return rc;
    if (coh) {
        float *acc = nullptr;
        if (grad_input) rc = coherent_accumulator(pb, ws, grad_input, acc);
        if (rc) return rc;
        rc = cs::coh::backward(coh_launch(pb), gOut, pr.icl, grid, offset, acc, grad_grid);
        return rc || !grad_input ? rc : coherent_finish(pb, acc, grad_input);
    }
    if (!grad_input) {
        CS_DISPATCH_KERNEL(pb.kernel, CS_DISPATCH_CQT(pb.d.C, (tl::point_backward<KERNEL, CQ, false, ST><<<point_grid(pb), kBlock, q_lds(tl::row1((int)cpad(pb.d.C)), 4), pb.stream>>>(
                                          (const ST *)gOut, pr.icl, grid, offset, nullptr, grad_grid, pr.d, pb.f))));
        return launch_status();
    }
    // with grad_input: the same point kernel also leaves the fat rows [gOut | W_a]; the tile walkers add them up
    float *fat = (float *)ws.take((size_t)pb.d.S * tl::row1((int)cpad(pb.d.C)) * 4);
    if (!ws.ok()) return CS_ERR_WORKSPACE;
    rc = walker_target(pb, grad_input);
    if (rc) return rc;
    CS_DISPATCH_KERNEL(pb.kernel, CS_DISPATCH_CQT(pb.d.C, (tl::point_backward<KERNEL, CQ, true, ST><<<point_grid(pb), kBlock, q_lds(tl::row1((int)cpad(pb.d.C)), 4), pb.stream>>>(
                                      (const ST *)gOut, pr.icl, grid, offset, fat, grad_grid, pr.d, pb.f))));
    rc = launch_status();
    if (rc) return rc;
    // a caller's plan outlives this call: when asked, leave the sorted copy of grad_output in it for later stages' walkers
    if (g_leave && pr.plan_is_callers && !pr.plan.dense) return launch_tile_scatter<0, true>(pb, pr.plan, fat, grad_input);
    return launch_tile_scatter<0>(pb, pr.plan, fat, grad_input);
}

int tiled_bb(const Problem &pb, const float *cI, const float *cG, const float *input, const float *grid,
             const float *gOut, const float *offset, float *gInput, float *gGrid, float *ggOut,
             const float *input_cl, const void *plan, void *workspace, size_t workspace_bytes, bool g_sorted, bool g_leave) {
    Carve ws{(char *)workspace, 0, workspace ? workspace_bytes : 0};
    Prepared pr;
    // without gInput nothing is scattered: no plan, no fat rows
    const bool coh = !cI && coherent_applies(pb);   // (with grad_out_input: the general path)
    if (pb.sum_n && cI) return CS_ERR_UNSUPPORTED;
    int rc = sum_n_check(pb, true, false);
    if (rc) return rc;
    rc = prepare(pb, gInput && !coh ? CS_STAGE_BACKWARD_BACKWARD : CS_STAGE_FORWARD, input, grid, offset, input_cl, plan, ws, pr);
    if (rc) return rc;
    const float *cIcl = nullptr;
    if (cI) {
        float *buf = (float *)ws.take((size_t)pb.d.N * cpad(pb.d.C) * pb.d.vol * 4);
        if (!ws.ok()) return CS_ERR_WORKSPACE;
        rc = pack_cl(cI, buf, pb.d.N, pb.d.C, pb.d.vol, pb.stream);
        if (rc) return rc;
        cIcl = buf;
    }
    if (coh) {
        float *acc = nullptr;
        if (gInput) rc = coherent_accumulator(pb, ws, gInput, acc);
        if (rc) return rc;
        rc = cs::coh::bb(coh_launch(pb), cG, pr.icl, grid, gOut, offset, acc, gGrid, ggOut);
        return rc || !gInput ? rc : coherent_finish(pb, acc, gInput);
    }
    // the plan already holds THIS grad_output in sorted order (an earlier stage of the step left it): the point kernel
    // writes the 16-byte D record only and the walkers stream the payload
    const bool lean = gInput && g_sorted && pr.plan_is_callers && !pr.plan.dense;
    float *fat = nullptr;
    if (gInput) {
        fat = (float *)ws.take((size_t)pb.d.S * (lean ? 4 : tl::row1((int)cpad(pb.d.C))) * 4);
        if (!ws.ok()) return CS_ERR_WORKSPACE;
        rc = walker_target(pb, gInput);
        if (rc) return rc;
    }
    const size_t shm = q_lds(tl::row1((int)cpad(pb.d.C)), 12);
#define CS_TILED_BB(HAS_CI, ROWS)                                                                                      \
    CS_DISPATCH_KERNEL(pb.kernel, CS_DISPATCH_CQT(pb.d.C, (tl::point_bb<KERNEL, CQ, HAS_CI, ROWS, ST><<<point_grid(pb), kBlock, shm, pb.stream>>>( \
                                      cIcl, cG, pr.icl, grid, (const ST *)gOut, offset, fat, gGrid, (ST *)ggOut, pr.d, pb.f))))
    if (!gInput) { if (cIcl) { CS_TILED_BB(true, 0); } else { CS_TILED_BB(false, 0); } }
    else if (lean) { if (cIcl) { CS_TILED_BB(true, 2); } else { CS_TILED_BB(false, 2); } }
    else { if (cIcl) { CS_TILED_BB(true, 1); } else { CS_TILED_BB(false, 1); } }
#undef CS_TILED_BB
    rc = launch_status();
    if (rc || !gInput) return rc;
    if (lean) return launch_tile_scatter<2>(pb, pr.plan, fat, gInput);
    if (g_leave && pr.plan_is_callers && !pr.plan.dense) return launch_tile_scatter<0, true>(pb, pr.plan, fat, gInput);
    return launch_tile_scatter<0>(pb, pr.plan, fat, gInput);
}

int tiled_bbb(const Problem &pb, const float *input, const float *grid, const float *gOut, const float *cG,
              const float *hG, const float *hO, const float *offset, float *gInput, float *ggOut,
              const float *input_cl, const void *plan, void *workspace, size_t workspace_bytes, bool g_sorted, bool g_leave) {
    Carve ws{(char *)workspace, 0, workspace ? workspace_bytes : 0};
    Prepared pr;
    const bool coh = coherent_applies(pb);
    int rc = sum_n_check(pb, true, hO != nullptr);
    if (rc) return rc;
    rc = prepare(pb, coh ? CS_STAGE_FORWARD : CS_STAGE_BBB_FUSED, input, grid, offset, input_cl, plan, ws, pr);
    if (rc) return rc;
    if (coh) {
        float *acc;
        rc = coherent_accumulator(pb, ws, gInput, acc);
        if (rc) return rc;
        rc = cs::coh::bbb(coh_launch(pb), pr.icl, grid, gOut, cG, hG, hO, offset, acc, ggOut);
        return rc ? rc : coherent_finish(pb, acc, gInput);
    }
    const bool lean = hO && g_sorted && pr.plan_is_callers && !pr.plan.dense;   // see tiled_bb
    float *fat = (float *)ws.take((size_t)pb.d.S * (lean ? tl::row3((int)cpad(pb.d.C)) : tl::row2((int)cpad(pb.d.C))) * 4);
    if (!ws.ok()) return CS_ERR_WORKSPACE;
    rc = walker_target(pb, gInput);
    if (rc) return rc;
    if (lean) {
        const size_t shm = q_lds(tl::row3((int)cpad(pb.d.C)), 0);
        CS_DISPATCH_KERNEL(pb.kernel, CS_DISPATCH_CQT(pb.d.C, {
            rc = allow_lds(tl::point_bbb<KERNEL, CQ, true, true, ST>, shm);
            if (!rc) tl::point_bbb<KERNEL, CQ, true, true, ST><<<point_grid(pb), kBlock, shm, pb.stream>>>(
                         pr.icl, grid, (const ST *)gOut, cG, hG, (const ST *)hO, offset, fat, (ST *)ggOut, pr.d, pb.f);
        }));
        if (rc) return rc;
        rc = launch_status();
        if (rc) return rc;
        return launch_tile_scatter<3>(pb, pr.plan, fat, gInput);
    }
    if (hO) {
        const size_t shm = q_lds(tl::row2((int)cpad(pb.d.C)), 0);   // 80 KiB at 32 channels
        CS_DISPATCH_KERNEL(pb.kernel, CS_DISPATCH_CQT(pb.d.C, {
            rc = allow_lds(tl::point_bbb<KERNEL, CQ, true, false, ST>, shm);
            if (!rc) tl::point_bbb<KERNEL, CQ, true, false, ST><<<point_grid(pb), kBlock, shm, pb.stream>>>(
                         pr.icl, grid, (const ST *)gOut, cG, hG, (const ST *)hO, offset, fat, (ST *)ggOut, pr.d, pb.f);
        }));
        if (rc) return rc;
        rc = launch_status();
        if (rc) return rc;
        if (g_leave && pr.plan_is_callers && !pr.plan.dense) return launch_tile_scatter<1, true>(pb, pr.plan, fat, gInput);
        return launch_tile_scatter<1>(pb, pr.plan, fat, gInput);
    } else {
        CS_DISPATCH_KERNEL(pb.kernel, CS_DISPATCH_CQT(pb.d.C, (tl::point_bbb<KERNEL, CQ, false, false, ST><<<point_grid(pb), kBlock, q_lds(tl::row1((int)cpad(pb.d.C)), 0), pb.stream>>>(
                                          pr.icl, grid, (const ST *)gOut, cG, hG, (const ST *)hO, offset, fat, (ST *)ggOut, pr.d, pb.f))));
        rc = launch_status();
        if (rc) return rc;
        if (g_leave && pr.plan_is_callers && !pr.plan.dense) return launch_tile_scatter<0, true>(pb, pr.plan, fat, gInput);
        return launch_tile_scatter<0>(pb, pr.plan, fat, gInput);
    }
}

// ------------------------------------------------------------------------------------------------
// 3D with C <= 16 (zero-padded to 4, 8 or 16 channels): channels-last point kernels + fused row atomics, or the dense path for crowded tables
// ------------------------------------------------------------------------------------------------
bool rows_cl_applies(int dim, int64_t N, int64_t C, int64_t P, int64_t vol) {
    const int mode = g_force_path.load(std::memory_order_relaxed);
    if (mode == 1 || dim != 3 || C > 16) return false;   // other counts run zero-padded to 4, 8 or 16 (cpad)
    if (N * P >= ((int64_t)1 << 31) || N * vol >= ((int64_t)1 << 31) || vol * cpad(C) >= ((int64_t)1 << 30)) return false;   // (the table copy holds two rows per node)
    if (N > 65535) return false;                               // pack / unpack launch with gridDim.y = N
    return mode >= 2 || N * P >= (1 << 16);   // (global node ids of the fused scatter are 32-bit)
}

// CS_SUM_OVER_N in 3D (round 4): the channels-last point kernels walk the N tables per point (cs_points_cl.cuh, Walk).
// One lane per POINT: the launch covers P, the streams have no n.  fp32 streams, like the 2D summing kernels.
bool sum_n3_applies(const Problem &pb) {
    return pb.sum_n && pb.dim == 3 && pb.sdt == 0 && pb.d.grid_ns == 0 && pb.d.N > 1 && pb.d.S > 0 && pb.d.C > 0 &&
           rows_cl_applies(3, pb.d.N, pb.d.C, pb.d.P, pb.d.vol);
}
// the problem as the summing launch sees it; the scatter kernels that follow get the plain one (their work is per table)
Problem nsum3(const Problem &pb) {
    Problem q = pb;
    if (sum_n3_applies(pb)) {
        q.d.nsum = 1;
        q.blocks = (unsigned)((pb.d.P + kBlock - 1) / kBlock);
    }
    return q;
}
int sum_n3_check(const Problem &pb, bool has_gO, bool has_hO) {
    if (!pb.sum_n) return CS_OK;
    if (!sum_n3_applies(pb)) return CS_ERR_UNSUPPORTED;
    if ((has_gO && pb.d.go_ns != 0) || (has_hO && pb.d.ho_ns != 0)) return CS_ERR_INVALID;      // one cotangent for every table
    return CS_OK;
}


// 3D crowded tables (cs_dense3d.cuh): cells fit the LDS histogram, one (node, channel) value per lane in cell_scatter3
// (two at 16 channels), and enough samples per cell for a wave per cell to pay: measured with 32^3-cell tables, 28 samples per cell
// 1.9 vs 4.9 ms per stage, 2.8 per cell 1.47 vs 1.26 ms -- the threshold is 8
// the cell histogram of the 3D plan lives in LDS: 160 KiB per workgroup on gfx950 = 40960 bins (a 33^3-cell table fits)
constexpr int64_t kDense3MaxCells = 40000;
bool dense3_applies(int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int64_t P) {
    if (g_force_path.load(std::memory_order_relaxed) == 3) return false;   // testing: row atomics only
    const int64_t cells = (W + 1) * (H + 1) * (D + 1);
    return C <= 16 && cells <= kDense3MaxCells && P >= 8 * cells && N * cells < (int64_t)INT32_MAX &&
           N * P < (int64_t)0xFFFFFFF0ll && N <= 65535;
}
struct Plan3Layout {
    int ntx, nty, ntiles, chunks, chunk;
    size_t off_sorted, off_tile_begin, off_block_hist, off_totals, off_bsum, bytes;
};
Plan3Layout plan3_layout(int64_t N, int64_t D, int64_t H, int64_t W, int64_t P) {
    Plan3Layout L;
    L.ntx = (int)(W + 1);
    L.nty = (int)(H + 1);
    L.ntiles = (int)((W + 1) * (H + 1) * (D + 1));
    L.chunk = plan_chunk(N, P, L.ntiles);   // fewer, larger chunks when a histogram is 50-160 KiB
    L.chunks = (int)((P + L.chunk - 1) / L.chunk);
    size_t o = 0;
    L.off_sorted = o;     o += align256((size_t)N * P * 4);
    L.off_tile_begin = o; o += align256(((size_t)N * L.ntiles + 1) * 4);
    L.off_block_hist = o; o += align256((size_t)N * L.chunks * L.ntiles * 4);
    L.off_totals = o;     o += align256((size_t)N * L.ntiles * 4);
    L.off_bsum = o;       o += align256(((size_t)N * L.ntiles / 1024 + 3) * 4);
    L.bytes = o;
    return L;
}
tl::Plan plan3_view(const Plan3Layout &L, void *blob) {
    char *b = (char *)blob;
    tl::Plan p;
    p.sorted = (uint32_t *)(b + L.off_sorted);
    p.key = nullptr;
    p.cellb = nullptr;
    p.Gs = nullptr;
    p.tile_begin = (uint32_t *)(b + L.off_tile_begin);
    p.cell_begin = nullptr;
    p.block_hist = (uint32_t *)(b + L.off_block_hist);
    p.ntx = L.ntx;
    p.nty = L.nty;
    p.ntiles = L.ntiles;
    p.chunks = L.chunks;
    p.chunk = L.chunk;
    p.dense = 1;
    return p;
}
int build_plan3(const Problem &pb, const float *grid, const float *offset, void *blob) {
    Plan3Layout L = plan3_layout(pb.d.N, pb.d.size[2], pb.d.size[1], pb.d.size[0], pb.d.P);
    tl::Plan pl = plan3_view(L, blob);
    uint32_t *totals = (uint32_t *)((char *)blob + L.off_totals);
    dim3 g((unsigned)L.chunks, (unsigned)pb.d.N);
    size_t shm = (size_t)L.ntiles * 4;
    if (shm > 64 * 1024) {   // beyond the default dynamic-LDS limit: opt in (idempotent, no device work)
        hipError_t e1 = hipFuncSetAttribute((const void *)cs::dense3::plan_count3,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        hipError_t e2 = hipFuncSetAttribute((const void *)cs::dense3::plan_scatter3,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e1 != hipSuccess || e2 != hipSuccess) return (int)(e1 != hipSuccess ? e1 : e2);
    }
    cs::dense3::plan_count3<<<g, 256, shm, pb.stream>>>(grid, offset, pl, pb.d, pb.f);
    int64_t nt = (int64_t)pb.d.N * L.ntiles;
    tl::plan_scan_chunks<<<(unsigned)((nt + 255) / 256), 256, 0, pb.stream>>>(pl, pb.d.N, totals);
    scan_buckets(totals, pl.tile_begin, (uint32_t *)((char *)blob + L.off_bsum), nt, pb.stream);
    cs::dense3::plan_scatter3<<<g, 256, shm, pb.stream>>>(grid, offset, pl, pb.d, pb.f);
    return launch_status();
}
// 3D tables too large for the cell histogram (cs_dense3d.cuh, tiles3): grad_input cut into 16x4x4-node tiles, every sample
// listed in the tiles that own its corners, one wave per tile sums its list without atomics and writes the tile out
namespace t3 = cs::tiles3;
struct Tiles3Dims { int64_t ntx, nty, ntz, ntiles; };
Tiles3Dims tiles3_dims(int64_t D, int64_t H, int64_t W) {
    Tiles3Dims t;
    t.ntx = (W + t3::M3X - 1) / t3::M3X;
    t.nty = (H + t3::M3Y - 1) / t3::M3Y;
    t.ntz = (D + t3::M3Z - 1) / t3::M3Z;
    t.ntiles = t.ntx * t.nty * t.ntz;
    return t;
}
constexpr int64_t kTiles3MinSamples = 1 << 18;
constexpr int64_t kTiles3MaxLists = 8;   // a sample is listed at most once per corner
bool tiles3_applies(int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int64_t P) {
    const int mode = g_force_path.load(std::memory_order_relaxed);
    if (mode == 1 || mode == 3) return false;              // testing: direct kernels / row atomics only
    if (dense3_applies(N, C, D, H, W, P)) return false;    // crowded tables: a wave per cell
    const Tiles3Dims t = tiles3_dims(D, H, W);
    if (C > 16 || t.ntiles > 12288 || N * t.ntiles >= (int64_t)INT32_MAX) return false;   // tile histogram in 48 KiB of LDS
    // keys are (p << 9) | code; list positions are 32-bit
    if (P >= ((int64_t)1 << 23) || N * P * kTiles3MaxLists >= (int64_t)0xFFFFFFF0ll || N > 65535) return false;
    return mode >= 2 || N * P >= kTiles3MinSamples;
}
struct Plan3TLayout {
    Tiles3Dims t;
    int chunks, chunk;
    size_t off_key, off_tile_begin, off_block_hist, off_totals, off_bsum, bytes;
};
Plan3TLayout plan3t_layout(int64_t N, int64_t D, int64_t H, int64_t W, int64_t P) {
    Plan3TLayout L;
    L.t = tiles3_dims(D, H, W);
    L.chunk = plan_chunk(N, P, (int)L.t.ntiles);
    L.chunks = (int)((P + L.chunk - 1) / L.chunk);
    const size_t E = (size_t)N * P * kTiles3MaxLists;   // list entries, worst case (1.66 per sample at 16x4x4 nodes)
    size_t o = 0;
    L.off_key = o;        o += align256(E * 4);
    L.off_tile_begin = o; o += align256(((size_t)N * L.t.ntiles + 1) * 4);
    L.off_block_hist = o; o += align256((size_t)N * L.chunks * L.t.ntiles * 4);
    L.off_totals = o;     o += align256((size_t)N * L.t.ntiles * 4);
    L.off_bsum = o;       o += align256(((size_t)N * L.t.ntiles / 1024 + 3) * 4);
    L.bytes = o;
    return L;
}
tl::Plan plan3t_view(const Plan3TLayout &L, void *blob) {
    char *b = (char *)blob;
    tl::Plan p;
    p.sorted = nullptr;
    p.key = (uint32_t *)(b + L.off_key);      // the lists themselves: (p << 9) | code, grouped by tile
    p.cellb = nullptr;
    p.Gs = nullptr;
    p.tile_begin = (uint32_t *)(b + L.off_tile_begin);
    p.cell_begin = nullptr;
    p.block_hist = (uint32_t *)(b + L.off_block_hist);
    p.ntx = (int)L.t.ntx;
    p.nty = (int)L.t.nty;
    p.ntiles = (int)L.t.ntiles;
    p.chunks = L.chunks;
    p.chunk = L.chunk;
    p.dense = 0;
    return p;
}
int build_plan3t(const Problem &pb, const float *grid, const float *offset, void *blob) {
    Plan3TLayout L = plan3t_layout(pb.d.N, pb.d.size[2], pb.d.size[1], pb.d.size[0], pb.d.P);
    tl::Plan pl = plan3t_view(L, blob);
    uint32_t *totals = (uint32_t *)((char *)blob + L.off_totals);
    dim3 g((unsigned)L.chunks, (unsigned)pb.d.N);
    const size_t shm = (size_t)pl.ntiles * 4;
    t3::plan_count3t<<<g, 256, shm, pb.stream>>>(grid, offset, pl, pb.d, pb.f);
    const int64_t nt = (int64_t)pb.d.N * pl.ntiles;
    tl::plan_scan_chunks<<<(unsigned)((nt + 255) / 256), 256, 0, pb.stream>>>(pl, pb.d.N, totals);
    scan_buckets(totals, pl.tile_begin, (uint32_t *)((char *)blob + L.off_bsum), nt, pb.stream);
    t3::plan_scatter3t<<<g, 256, shm, pb.stream>>>(grid, offset, pl, pb.d, pb.f);
    return launch_status();
}
// floats per p-ordered row of the 3D dense path: cl::Rec without the node ids
int dense3_row_floats(int64_t C, int stage) { return (int)((cpad(C) + 8) * (stage == CS_STAGE_BBB_FUSED ? 2 : 1)); }

// (these three are 3D paths: a channels-last TABLE copy is z-paired, twice the accumulator's size)
size_t rows_cl_workspace(int stage, int64_t N, int64_t C, int64_t vol, int have_cl, int have_cI) {
    size_t T = align256((size_t)N * cpad(C) * vol * 4), T2 = align256(table_floats(3, N, C, vol) * 4), need = 0;
    if (!have_cl) need += T2;
    if (stage == CS_STAGE_FORWARD) return need;
    if (stage == CS_STAGE_BACKWARD_BACKWARD && have_cI) need += T2;
    return need + T;   // + the channels-last accumulator of row_scatter
}
size_t dense3_workspace(int stage, int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int64_t P, int have_cl,
                        int have_plan, int have_cI) {
    size_t T = align256(table_floats(3, N, C, D * H * W) * 4), need = 0;
    if (!have_cl) need += T;
    if (stage == CS_STAGE_FORWARD) return need;
    if (stage == CS_STAGE_BACKWARD_BACKWARD && have_cI) need += T;
    if (!have_plan) need += align256(plan3_layout(N, D, H, W, P).bytes);
    return need + align256((size_t)N * P * dense3_row_floats(C, stage) * 4);
}

size_t tiles3_workspace(int stage, int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int64_t P, int have_cl,
                        int have_plan, int have_cI) {
    size_t T = align256(table_floats(3, N, C, D * H * W) * 4), need = 0;
    if (!have_cl) need += T;
    if (stage == CS_STAGE_FORWARD) return need;
    if (stage == CS_STAGE_BACKWARD_BACKWARD && have_cI) need += T;
    if (!have_plan) need += align256(plan3t_layout(N, D, H, W, P).bytes);
    return need + align256((size_t)N * P * dense3_row_floats(C, stage) * 4);
}

// resolve the channels-last table (caller's or packed into the workspace)
int rows_cl_table(const Problem &pb, const float *input, const float *input_cl, Carve &ws, const float *&icl) {
    if (input_cl) { icl = input_cl; return CS_OK; }
    float *buf = (float *)ws.take(table_floats(3, pb.d.N, pb.d.C, pb.d.vol) * 4);
    if (!ws.ok()) return CS_ERR_WORKSPACE;
    icl = buf;
    return pack_cl(input, buf, pb.d.N, pb.d.C, pb.d.vol, pb.stream, (int64_t)pb.d.size[0] * pb.d.size[1]);
}

template <int DIM>
int rcl_forward(const Problem &pb, const float *input, const float *grid, const float *offset, float *output,
                const float *input_cl, void *workspace, size_t workspace_bytes) {
    Carve ws{(char *)workspace, 0, workspace ? workspace_bytes : 0};
    const float *icl;
    int rc = sum_n3_check(pb, false, false);
    if (rc) return rc;
    rc = rows_cl_table(pb, input, input_cl, ws, icl);
    if (rc) return rc;
    const Problem pk = nsum3(pb);      // (the summing mode: one lane per point, walking the tables)
    CS_DISPATCH_KERNEL(pb.kernel, CS_DISPATCH_CQ(pb.d.C, (cs::cl::forward<DIM, KERNEL, CQ, ST><<<pk.blocks, kBlock, 0, pb.stream>>>(
                                      icl, grid, offset, (ST *)output, pk.d, pb.f))));
    return launch_status();
}

// LDS of the fused scatter: 4 waves x 64 records of [payloads | coefficients | node ids]
template <int DIM>
size_t rcl_lds(int C, int mode) {
    const int NC = 1 << DIM, np = mode == 2 ? 2 : 1;
    const int CP = (int)cpad(C);
    return (size_t)256 * (CP * np + NC * np + NC) * 4;
}
// accumulator -> caller's layout
int rcl_finish(const Problem &pb, const float *acc, float *out_grad) {
    return unpack_cl(acc, out_grad, pb.d.N, pb.d.C, cpad(pb.d.C), pb.d.vol, pb.stream);
}
int rcl_accumulator(const Problem &pb, Carve &ws, float *&acc) {
    const int64_t T = (int64_t)pb.d.N * cpad(pb.d.C) * pb.d.vol;
    acc = (float *)ws.take((size_t)T * 4);
    if (!ws.ok()) return CS_ERR_WORKSPACE;
    return zero_async(acc, T, pb.stream);
}

// 3D dense path: plan (caller's or built into the workspace), p-ordered rows, one wave per (n, cell) bucket
int dense3_prepare(const Problem &pb, const float *grid, const float *offset, const void *plan, Carve &ws, int stage,
                   tl::Plan &pl, float *&rows) {
    Plan3Layout L = plan3_layout(pb.d.N, pb.d.size[2], pb.d.size[1], pb.d.size[0], pb.d.P);
    if (plan) {
        pl = plan3_view(L, const_cast<void *>(plan));
    } else {
        void *blob = ws.take(L.bytes);
        if (!ws.ok()) return CS_ERR_WORKSPACE;
        int rc = build_plan3(pb, grid, offset, blob);
        if (rc) return rc;
        pl = plan3_view(L, blob);
    }
    rows = (float *)ws.take((size_t)pb.d.S * dense3_row_floats(pb.d.C, stage) * 4);
    return ws.ok() ? CS_OK : CS_ERR_WORKSPACE;
}
template <int MODE>
int dense3_scatter(const Problem &pb, const tl::Plan &pl, const float *rows, float *grad_input) {
    unsigned nbk = (unsigned)(((int64_t)pb.d.N * pl.ntiles + 3) / 4);
    CS_DISPATCH_CQ(pb.d.C, (cs::dense3::cell_scatter3<CQ, MODE><<<nbk, 256, 0, pb.stream>>>(rows, pl, grad_input, pb.d)));
    return launch_status();
}

// 3D tile path: plan (caller's or built into the workspace), p-ordered rows
int tiles3_prepare(const Problem &pb, const float *grid, const float *offset, const void *plan, Carve &ws, int stage,
                   tl::Plan &pl, float *&rows) {
    Plan3TLayout L = plan3t_layout(pb.d.N, pb.d.size[2], pb.d.size[1], pb.d.size[0], pb.d.P);
    if (plan) {
        pl = plan3t_view(L, const_cast<void *>(plan));
    } else {
        void *blob = ws.take(L.bytes);
        if (!ws.ok()) return CS_ERR_WORKSPACE;
        int rc = build_plan3t(pb, grid, offset, blob);
        if (rc) return rc;
        pl = plan3t_view(L, blob);
    }
    rows = (float *)ws.take((size_t)pb.d.S * dense3_row_floats(pb.d.C, stage) * 4);
    return ws.ok() ? CS_OK : CS_ERR_WORKSPACE;
}
template <int MODE>
int tiles3_scatter(const Problem &pb, const tl::Plan &pl, const float *rows, float *grad_input) {
    const int CP = (int)cpad(pb.d.C);
    const int waves = CP <= 8 ? 4 : 2;                                   // 4 / 8 / 16 KiB of LDS image per wave
    const size_t shm = (size_t)waves * (t3::OWNED * CP + t3::CNT_WORDS) * 4;   // images + the waves' code counters
    const int64_t nb = (int64_t)pb.d.N * pl.ntiles;
    CS_DISPATCH_CQ(pb.d.C, (t3::tile3_scatter<CQ, MODE><<<(unsigned)((nb + waves - 1) / waves), 64 * waves, shm, pb.stream>>>(
                               rows, pl, grad_input, pb.d, waves)));
    return launch_status();
}
// which scatter a 3D stage with grad_input uses: 0 fused row atomics, 1 wave per cell (crowded), 2 wave per tile (no atomics)
template <int DIM>
int scatter3_way(const Problem &pb) {
    if (DIM != 3) return 0;
    if (dense3_applies(pb.d.N, pb.d.C, pb.d.size[2], pb.d.size[1], pb.d.size[0], pb.d.P)) return 1;
    return tiles3_applies(pb.d.N, pb.d.C, pb.d.size[2], pb.d.size[1], pb.d.size[0], pb.d.P) ? 2 : 0;
}

template <int DIM>
int rcl_backward(const Problem &pb, const float *gOut, const float *input, const float *grid, const float *offset,
                 float *grad_input, float *grad_grid, const float *input_cl, const void *plan, void *workspace,
                 size_t workspace_bytes) {
    Carve ws{(char *)workspace, 0, workspace ? workspace_bytes : 0};
    const float *icl;
    int rc = sum_n3_check(pb, true, false);
    if (rc) return rc;
    rc = rows_cl_table(pb, input, input_cl, ws, icl);
    if (rc) return rc;
    const Problem pk = nsum3(pb);
    if (!grad_input) {
        CS_DISPATCH_KERNEL(pb.kernel, CS_DISPATCH_CQ(pb.d.C, (cs::cl::backward<DIM, KERNEL, CQ, 0, ST><<<pk.blocks, kBlock, 0, pb.stream>>>(
                                          (const ST *)gOut, icl, grid, offset, grad_grid, nullptr, pk.d, pb.f))));
        return launch_status();
    }
    const size_t shm = rcl_lds<DIM>(pb.d.C, 0);
    if (const int way = scatter3_way<DIM>(pb)) {
        tl::Plan pl;
        float *rows;
        rc = way == 1 ? dense3_prepare(pb, grid, offset, plan, ws, CS_STAGE_BACKWARD, pl, rows)
                      : tiles3_prepare(pb, grid, offset, plan, ws, CS_STAGE_BACKWARD, pl, rows);
        if (rc) return rc;
        if (way == 1) rc = zero_async(grad_input, (int64_t)pb.d.N * pb.d.C * pb.d.vol, pb.stream);
        if (rc) return rc;
        CS_DISPATCH_KERNEL(pb.kernel, CS_DISPATCH_CQ(pb.d.C, (cs::cl::backward<DIM, KERNEL, CQ, 2, ST><<<pk.blocks, kBlock, shm, pb.stream>>>(
                                          (const ST *)gOut, icl, grid, offset, grad_grid, rows, pk.d, pb.f))));
        rc = launch_status();
        if (rc) return rc;
        return way == 1 ? dense3_scatter<0>(pb, pl, rows, grad_input) : tiles3_scatter<0>(pb, pl, rows, grad_input);
    }
    float *acc;
    rc = rcl_accumulator(pb, ws, acc);
    if (rc) return rc;
    CS_DISPATCH_KERNEL(pb.kernel, CS_DISPATCH_CQ(pb.d.C, (cs::cl::backward<DIM, KERNEL, CQ, 1, ST><<<pk.blocks, kBlock, shm, pb.stream>>>(
                                      (const ST *)gOut, icl, grid, offset, grad_grid, acc, pk.d, pb.f))));
    rc = launch_status();
    if (rc) return rc;
    return rcl_finish(pb, acc, grad_input);
}

template <int DIM>
int rcl_bb(const Problem &pb, const float *cI, const float *cG, const float *input, const float *grid,
           const float *gOut, const float *offset, float *gInput, float *gGrid, float *ggOut, const float *input_cl,
           const void *plan, void *workspace, size_t workspace_bytes) {
    Carve ws{(char *)workspace, 0, workspace ? workspace_bytes : 0};
    const float *icl;
    if (pb.sum_n && cI) return CS_ERR_UNSUPPORTED;       // (the summed op has no cotangent of grad_input: not the PIXEL pattern)
    int rc = sum_n3_check(pb, true, false);
    if (rc) return rc;
    rc = rows_cl_table(pb, input, input_cl, ws, icl);
    if (rc) return rc;
    const Problem pk = nsum3(pb);
    const float *cIcl = nullptr;
    if (cI) {
        float *buf = (float *)ws.take(table_floats(DIM, pb.d.N, pb.d.C, pb.d.vol) * 4);
        if (!ws.ok()) return CS_ERR_WORKSPACE;
        rc = pack_cl(cI, buf, pb.d.N, pb.d.C, pb.d.vol, pb.stream, DIM == 3 ? (int64_t)pb.d.size[0] * pb.d.size[1] : 0);
        if (rc) return rc;
        cIcl = buf;
    }
    const int way = gInput ? scatter3_way<DIM>(pb) : 0;
    const bool dense = way != 0;   // the point kernel leaves p-ordered rows (SCATTER == 2)
    float *acc = nullptr;     // channels-last accumulator (row atomics) or the p-ordered rows (dense / tiles)
    tl::Plan pl;
    if (way == 1) {
        rc = dense3_prepare(pb, grid, offset, plan, ws, CS_STAGE_BACKWARD_BACKWARD, pl, acc);
        if (rc) return rc;
        rc = zero_async(gInput, (int64_t)pb.d.N * pb.d.C * pb.d.vol, pb.stream);
        if (rc) return rc;
    } else if (way == 2) {
        rc = tiles3_prepare(pb, grid, offset, plan, ws, CS_STAGE_BACKWARD_BACKWARD, pl, acc);
        if (rc) return rc;
    } else if (gInput) {
        rc = rcl_accumulator(pb, ws, acc);
        if (rc) return rc;
    }
    const size_t shm = gInput ? rcl_lds<DIM>(pb.d.C, 1) : 0;
#define CS_RCL_BB(HAS_CI, SCATTER)                                                                                   \
    CS_DISPATCH_KERNEL(pb.kernel, CS_DISPATCH_CQ(pb.d.C, (cs::cl::backward_backward<DIM, KERNEL, CQ, HAS_CI, SCATTER, ST>  \
                                      <<<pk.blocks, kBlock, shm, pb.stream>>>(cIcl, cG, icl, grid, (const ST *)gOut, offset, gGrid, (ST *)ggOut, acc, pk.d, pb.f))))
    if (cIcl && dense) { CS_RCL_BB(true, 2); }
    else if (cIcl && gInput) { CS_RCL_BB(true, 1); }
    else if (cIcl) { CS_RCL_BB(true, 0); }
    else if (dense) { CS_RCL_BB(false, 2); }
    else if (gInput) { CS_RCL_BB(false, 1); }
    else { CS_RCL_BB(false, 0); }
#undef CS_RCL_BB
    rc = launch_status();
    if (rc || !gInput) return rc;
    if (way == 1) return dense3_scatter<1>(pb, pl, acc, gInput);
    if (way == 2) return tiles3_scatter<1>(pb, pl, acc, gInput);
    return rcl_finish(pb, acc, gInput);
}

template <int DIM>
int rcl_bbb(const Problem &pb, const float *input, const float *grid, const float *gOut, const float *cG,
            const float *hG, const float *hO, const float *offset, float *gInput, float *ggOut,
            const float *input_cl, const void *plan, void *workspace, size_t workspace_bytes) {
    Carve ws{(char *)workspace, 0, workspace ? workspace_bytes : 0};
    const float *icl;
    int rc = sum_n3_check(pb, true, hO != nullptr);
    if (rc) return rc;
    rc = rows_cl_table(pb, input, input_cl, ws, icl);
    if (rc) return rc;
    const Problem pk = nsum3(pb);
    const size_t shm = rcl_lds<DIM>(pb.d.C, 2);
    if (const int way = scatter3_way<DIM>(pb)) {
        tl::Plan pl;
        float *rows;
        rc = way == 1 ? dense3_prepare(pb, grid, offset, plan, ws, CS_STAGE_BBB_FUSED, pl, rows)
                      : tiles3_prepare(pb, grid, offset, plan, ws, CS_STAGE_BBB_FUSED, pl, rows);
        if (rc) return rc;
        if (way == 1) rc = zero_async(gInput, (int64_t)pb.d.N * pb.d.C * pb.d.vol, pb.stream);
        if (rc) return rc;
        CS_DISPATCH_KERNEL(pb.kernel, CS_DISPATCH_CQ(pb.d.C, (cs::cl::bbb<DIM, KERNEL, CQ, 2, ST><<<pk.blocks, kBlock, shm, pb.stream>>>(
                                          icl, grid, (const ST *)gOut, cG, hG, (const ST *)hO, offset, (ST *)ggOut, rows, pk.d, pb.f))));
        rc = launch_status();
        if (rc) return rc;
        return way == 1 ? dense3_scatter<2>(pb, pl, rows, gInput) : tiles3_scatter<2>(pb, pl, rows, gInput);
    }
    float *acc;
    rc = rcl_accumulator(pb, ws, acc);
    if (rc) return rc;
    CS_DISPATCH_KERNEL(pb.kernel, CS_DISPATCH_CQ(pb.d.C, (cs::cl::bbb<DIM, KERNEL, CQ, 1, ST><<<pk.blocks, kBlock, shm, pb.stream>>>(
                                      icl, grid, (const ST *)gOut, cG, hG, (const ST *)hO, offset, (ST *)ggOut, acc, pk.d, pb.f))));
    rc = launch_status();
    if (rc) return rc;
    return rcl_finish(pb, acc, gInput);
}

bool any_null(std::initializer_list<const void *> ps) {
    for (const void *p : ps)
        if (!p) return true;
    return false;
}

// Opt-in third-order grid gradient (include/cosine_sampler.h): one direct kernel for both dimensionalities.
template <int DIM>
int bbb_grid_impl(Problem &pb, const float *input, const float *grid, const float *grad_output,
                         const float *grad_out_grid, const float *grad_out_ggrid, const float *grad_out_ggout,
                         const float *offset, float *grad_grid3) {
    if (pb.sum_n) return CS_ERR_UNSUPPORTED;
    if (pb.d.S == 0) return CS_OK;
    if (pb.d.C == 0) return zero_async(grad_grid3, pb.d.S * DIM, pb.stream);
    CS_DISPATCH_KERNEL(pb.kernel, (cs::direct_bbb_grid<DIM, KERNEL><<<pb.blocks, kBlock, 0, pb.stream>>>(
                                      input, grid, grad_output, grad_out_grid, grad_out_ggrid, grad_out_ggout, offset,
                                      grad_grid3, pb.d, pb.f)));
    return launch_status();
}

// 16-bit streams move as dwords shared by lane pairs (cs_tiled.cuh ld_pair16) when every channel row starts on a dword
void pair_streams(Problem &pb, std::initializer_list<const void *> streams) {
    bool ok = pb.sdt != 0 && pb.d.P % 2 == 0 && pb.d.go_ns % 2 == 0 && pb.d.ho_ns % 2 == 0 && pb.d.out_ns % 2 == 0;
    for (const void *q : streams) ok = ok && (reinterpret_cast<uintptr_t>(q) & 3) == 0;
    pb.f.pair16 = ok ? 1 : 0;
}

}  // namespace

static bool misaligned_ptr(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; }

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

int cs_debug_force_path(int mode) {
    if (!debug_enabled()) return 0;
    g_force_path.store(mode, std::memory_order_relaxed);
    return 1;
}
int cs_debug_coherent_tuning(int samples_per_wave, int ablation_bits) {
    if (!debug_enabled()) return 0;
    cs::coh::set_chunk(samples_per_wave, ablation_bits);
    return 1;
}

int cs_accumulator_kind(int dim, int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int64_t P, int kernel) {
    if (dim != 2 || N <= 0 || C <= 0 || P <= 0 || H <= 0 || W <= 0) return CS_ACC_NONE;
    Problem pb;
    if (make_problem(pb, 2, N, C, 1, H, W, P, 0, 1, kernel, 1, nullptr)) return CS_ACC_NONE;
    if (pb.f.exact || !tiled_applies(2, N, C, H, W, P)) return CS_ACC_NONE;
    return coherent_applies(pb) ? CS_ACC_CHANNELS_LAST : CS_ACC_NCHW;
}
size_t cs_accumulator_bytes(int dim, int kind, int64_t N, int64_t C, int64_t D, int64_t H, int64_t W) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || (dim == 3 && D <= 0)) return 0;
    const int64_t vol = (dim == 3 ? D : 1) * H * W;
    if (kind == CS_ACC_NCHW) return (size_t)N * C * vol * 4;
    if (kind == CS_ACC_CHANNELS_LAST) return (size_t)N * cpad(C) * vol * 4;
    return 0;
}
int cs_accumulator_finish(int dim, int kind, const float *acc, float *grad_input, int64_t N, int64_t C, int64_t D,
                          int64_t H, int64_t W, void *stream) {
    if ((dim != 2 && dim != 3) || N < 0 || C < 0 || H < 1 || W < 1 || (dim == 3 && D < 1)) return CS_ERR_INVALID;
    const int64_t vol = (dim == 3 ? D : 1) * H * W;
    if (N * C == 0) return CS_OK;
    if (!acc || !grad_input || misaligned_ptr(acc) || misaligned_ptr(grad_input)) return CS_ERR_INVALID;
    if (kind == CS_ACC_NCHW) {
        if (acc == grad_input) return CS_OK;
        hipError_t e = hipMemcpyAsync(grad_input, acc, (size_t)N * C * vol * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream);
        return e == hipSuccess ? CS_OK : (int)e;
    }
    if (kind != CS_ACC_CHANNELS_LAST || N > 65535) return CS_ERR_INVALID;
    return unpack_cl(acc, grad_input, N, C, cpad(C), vol, (hipStream_t)stream);
}

size_t cs_sort_points_bytes(int64_t P) { return cs::sort::workspace_bytes(P); }
int cs2d_sort_points(const float *points, float *sorted_points, int32_t *perm, int64_t P, int64_t H, int64_t W,
                     int padding_mode, int align_corners, int multicell, void *workspace, size_t workspace_bytes,
                     void *stream) {
    return cs::sort::sort_points(2, points, P, 1, H, W, padding_mode, align_corners, multicell, sorted_points, perm,
                                 workspace, workspace_bytes, (hipStream_t)stream);
}
int cs3d_sort_points(const float *points, float *sorted_points, int32_t *perm, int64_t P, int64_t D, int64_t H, int64_t W,
                     int padding_mode, int align_corners, int multicell, void *workspace, size_t workspace_bytes,
                     void *stream) {
    return cs::sort::sort_points(3, points, P, D, H, W, padding_mode, align_corners, multicell, sorted_points, perm,
                                 workspace, workspace_bytes, (hipStream_t)stream);
}
int cs_carry_points(const float *in, float *out, const int32_t *index, int64_t rows, int64_t P, int width, void *stream) {
    const int rc = cs::sort::carry_points(in, out, index, rows, P, width, (hipStream_t)stream);
    return rc < 0 ? CS_ERR_INVALID : rc;
}
int cs_points_tile_changes(int dim, const float *points, uint32_t *count, int64_t P, int64_t D, int64_t H, int64_t W,
                           int padding_mode, int align_corners, int multicell, void *stream) {
    return cs::sort::count_tile_changes(dim, points, P, D, H, W, padding_mode, align_corners, multicell, count,
                                        (hipStream_t)stream);
}

int cs_points_tile_changes_sampled(int dim, const float *points, uint32_t *count, int64_t P, int64_t D, int64_t H, int64_t W,
                                   int padding_mode, int align_corners, int multicell, int segments, void *stream) {
    return cs::sort::sample_tile_changes(dim, points, P, D, H, W, padding_mode, align_corners, multicell, segments, count,
                                         (hipStream_t)stream);
}

size_t cs_workspace_bytes(int dim, int stage, int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int64_t P,
                          int have_input_cl, int have_plan, int have_cI) {
    if (N <= 0 || C <= 0 || P <= 0 || H <= 0 || W <= 0 || (dim == 3 && D <= 0)) return 0;
    const bool coherent = (stage & CS_STAGE_POINTS_COHERENT) != 0 && dim == 2 && W <= cs::coh::MAX_SIZE_HOST &&
                          H <= cs::coh::MAX_SIZE_HOST && g_force_path.load(std::memory_order_relaxed) != 5;
    const bool accumulate = (stage & CS_STAGE_ACCUMULATE) != 0;
    stage &= ~(CS_STAGE_POINTS_COHERENT | CS_STAGE_ACCUMULATE);
    if (stage & CS_STAGE_NO_GRAD_INPUT) {
        // grad_input == NULL: nothing is scattered -- no plan, no rows, no accumulator; what is left is the
        // channels-last copy of the table (and of grad_out_input, when the second backward carries one)
        stage &= ~CS_STAGE_NO_GRAD_INPUT;
        if (stage != CS_STAGE_BACKWARD && stage != CS_STAGE_BACKWARD_BACKWARD) return 0;   // the third backward always scatters
        const size_t T = cs_pack_bytes(dim, N, C, D, H, W, P);
        return (have_input_cl ? 0 : T) + (stage == CS_STAGE_BACKWARD_BACKWARD && have_cI ? T : 0);
    }
    if (tiled_applies(dim, N, C, H, W, P)) return tiled_workspace(stage, N, C, H, W, P, have_input_cl, have_plan, have_cI, coherent, accumulate);
    const int64_t vol = (dim == 3 ? D : 1) * H * W;
    if (rows_cl_applies(dim, N, C, P, vol)) {
        if (dense3_applies(N, C, D, H, W, P)) return dense3_workspace(stage, N, C, D, H, W, P, have_input_cl, have_plan, have_cI);
        if (tiles3_applies(N, C, D, H, W, P)) return tiles3_workspace(stage, N, C, D, H, W, P, have_input_cl, have_plan, have_cI);
        return rows_cl_workspace(stage, N, C, vol, have_input_cl, have_cI);
    }
    if (stage != CS_STAGE_FORWARD && rows_applies(N, C, P, vol)) return align256((size_t)N * C * vol * 4);
    return 0;
}

int cs_half_streams_supported(int dim, int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int64_t P) {
    if (N <= 0 || C <= 0 || P <= 0 || H <= 0 || W <= 0 || D <= 0) return 0;
    if (dim == 2) return tiled_applies(2, N, C, H, W, P) ? 1 : 0;
    return (dim == 3 && rows_cl_applies(3, N, C, P, D * H * W)) ? 1 : 0;
}

int cs2d_sum_over_n_supported(int64_t N, int64_t C, int64_t H, int64_t W, int64_t P, int padding_mode, int align_corners) {
    if (N <= 0 || C <= 0 || P <= 0 || H <= 0 || W <= 0) return 0;
    if (!tiled_applies(2, N, C, H, W, P)) return 0;
    Problem pb;
    if (make_problem(pb, 2, N, C, 1, H, W, P, padding_mode, align_corners, CS_GRID_BROADCAST | CS_SUM_OVER_N, 1, nullptr)) return 0;
    return cs::coh::supported_nsum(coh_launch(pb)) ? 1 : 0;
}

int cs_sum_over_n_supported(int dim, int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int64_t P, int padding_mode,
                            int align_corners) {
    if (dim == 2) return cs2d_sum_over_n_supported(N, C, H, W, P, padding_mode, align_corners);
    if (dim != 3 || N <= 1 || C <= 0 || P <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    Problem pb;
    if (make_problem(pb, 3, N, C, D, H, W, P, padding_mode, align_corners, CS_GRID_BROADCAST | CS_SUM_OVER_N, 1, nullptr)) return 0;
    return sum_n3_applies(pb) ? 1 : 0;
}

size_t cs_pack_bytes(int dim, int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int64_t P) {
    if (N <= 0 || C <= 0 || P <= 0 || H <= 0 || W <= 0 || D <= 0) return 0;
    const int64_t vol = (dim == 3 ? D : 1) * H * W;
    if (!tiled_applies(dim, N, C, H, W, P) && !rows_cl_applies(dim, N, C, P, vol)) return 0;
    return align256(table_floats(dim, N, C, vol) * 4);
}

int cs_pack_input(int dim, const float *input, float *input_cl, int64_t N, int64_t C, int64_t D, int64_t H,
                  int64_t W, void *stream) {
    if (dim != 2 && dim != 3) return CS_ERR_INVALID;
    if (N < 0 || C < 0 || D < 1 || H < 1 || W < 1) return CS_ERR_INVALID;
    if (N * C > 0 && (!input || !input_cl)) return CS_ERR_INVALID;
    if (cpad(C) * 65 * 4 > 64 * 1024) return CS_ERR_UNSUPPORTED;
    return pack_cl(input, input_cl, N, C, (dim == 3 ? D : 1) * H * W, (hipStream_t)stream, dim == 3 ? H * W : 0);
}

size_t cs2d_plan_bytes(int64_t N, int64_t C, int64_t H, int64_t W, int64_t P) {
    if (N <= 0 || C <= 0 || P <= 0 || H <= 0 || W <= 0) return 0;
    if (!tiled_applies(2, N, C, H, W, P)) return 0;
    return plan_layout(N, C, H, W, P).bytes;
}

int cs2d_plan_build(const float *grid, const float *offset, void *plan, size_t plan_bytes, int64_t N, int64_t C,
                    int64_t H, int64_t W, int64_t P, int padding_mode, int align_corners, int multicell,
                    int flags, void *stream) {
    Problem pb;
    if (flags & ~CS_GRID_BROADCAST) return CS_ERR_INVALID;
    int rc = make_problem(pb, 2, N, C, 1, H, W, P, padding_mode, align_corners, flags, multicell, stream);
    if (rc) return rc;
    if (!tiled_applies(2, N, C, H, W, P)) return CS_ERR_UNSUPPORTED;
    if (!grid || !offset || !plan) return CS_ERR_INVALID;
    if (plan_bytes < plan_layout(N, C, H, W, P).bytes) return CS_ERR_WORKSPACE;
    return build_plan(pb, grid, offset, plan);
}

int cs2d_plan_keeps_sorted_copy(int64_t N, int64_t C, int64_t H, int64_t W, int64_t P) {
    if (N <= 0 || C <= 0 || P <= 0 || H <= 0 || W <= 0 || !tiled_applies(2, N, C, H, W, P)) return 0;
    return (!plan_layout(N, C, H, W, P).dense && g_force_path.load(std::memory_order_relaxed) != 4) ? 1 : 0;
}

size_t cs3d_plan_bytes(int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int64_t P) {
    if (N <= 0 || C <= 0 || P <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    if (!rows_cl_applies(3, N, C, P, D * H * W)) return 0;
    if (dense3_applies(N, C, D, H, W, P)) return plan3_layout(N, D, H, W, P).bytes;
    return tiles3_applies(N, C, D, H, W, P) ? plan3t_layout(N, D, H, W, P).bytes : 0;
}

int cs3d_plan_build(const float *grid, const float *offset, void *plan, size_t plan_bytes, int64_t N, int64_t C,
                    int64_t D, int64_t H, int64_t W, int64_t P, int padding_mode, int align_corners, int multicell,
                    int flags, void *stream) {
    Problem pb;
    if (flags & ~CS_GRID_BROADCAST) return CS_ERR_INVALID;
    int rc = make_problem(pb, 3, N, C, D, H, W, P, padding_mode, align_corners, flags, multicell, stream);
    if (rc) return rc;
    if (!rows_cl_applies(3, N, C, P, D * H * W)) return CS_ERR_UNSUPPORTED;
    const bool dense = dense3_applies(N, C, D, H, W, P);
    if (!dense && !tiles3_applies(N, C, D, H, W, P)) return CS_ERR_UNSUPPORTED;
    if (!grid || !offset || !plan) return CS_ERR_INVALID;
    if (plan_bytes < (dense ? plan3_layout(N, D, H, W, P).bytes : plan3t_layout(N, D, H, W, P).bytes)) return CS_ERR_WORKSPACE;
    return dense ? build_plan3(pb, grid, offset, plan) : build_plan3t(pb, grid, offset, plan);
}

#define CS_PROBLEM(dim, D)                                                                                        \
    Problem pb;                                                                                                   \
    {                                                                                                             \
        int rc_ = make_problem(pb, dim, N, C, D, H, W, P, padding_mode, align_corners, kernel, multicell, stream); \
        if (rc_) return rc_;                                                                                      \
    }                                                                                                             \
    const bool tiled = pb.d.S > 0 && pb.d.C > 0 && tiled_applies(dim, N, C, H, W, P);                             \
    if (pb.sum_n && !tiled && !(dim == 3 && sum_n3_applies(pb))) return CS_ERR_UNSUPPORTED;                       \
    bool g_sorted = false, g_leave = false;                                                                       \
    (void)g_sorted; (void)g_leave;                                                                                \
    const bool rows = !tiled && pb.d.S > 0 && pb.d.C > 0 && rows_applies(N, C, P, pb.d.vol);                      \
    (void)tiled; (void)rows; (void)input_cl; (void)plan; (void)workspace; (void)workspace_bytes;              \
    /* The direct kernels could gather from the channels-last copy too (Dims::tab_ns/tab_cs), but with   \
     * one scalar load per channel that measured SLOWER than the NC[D]HW gathers (3D config 4 forward    \
     * 2.88 vs 2.14 ms): they stay on the caller's tensor until they get float4 node loads. */           \
    const float *table_ = input;

// n-strides of the channel-major cotangents (include/cosine_sampler.h, cs_cotangent_layout)
#define CS_LAYOUT()                                                                                  \
    if (layout) {                                                                                    \
        if (layout->grad_output_stride_n < 0 || layout->grad_out_ggout_stride_n < 0) return CS_ERR_INVALID; \
        pb.d.go_ns = layout->grad_output_stride_n;                                                   \
        pb.d.ho_ns = layout->grad_out_ggout_stride_n;                                                \
        if (layout->grad_grad_out_stride_n) {                                                        \
            if (layout->grad_grad_out_stride_n < C * P) return CS_ERR_INVALID;                       \
            pb.d.out_ns = layout->grad_grad_out_stride_n;                                            \
        }                                                                                            \
        g_sorted = layout->sorted_grad_output_valid != 0;                                            \
        g_leave = layout->leave_sorted_grad_output != 0;                                             \
        if (layout->accumulate_grad_input < 0 || layout->accumulate_grad_input > CS_ACC_CHANNELS_LAST) return CS_ERR_INVALID; \
        pb.acc_kind = layout->accumulate_grad_input;                                                 \
        /* only the 2D fast paths add into a caller-held accumulator (cs_accumulator_kind says so beforehand) */ \
        if (pb.acc_kind && !tiled) return CS_ERR_UNSUPPORTED;                                        \
    }

// the alignment contract of include/cosine_sampler.h: the kernels issue 8-byte accesses on grid-shaped tensors and
// 16-byte accesses on input-shaped ones, the prepared objects and the workspace
static bool misaligned(std::initializer_list<const void *> ps, uintptr_t mask) {
    for (const void *p : ps)
        if (reinterpret_cast<uintptr_t>(p) & mask) return true;
    return false;
}
#define CS_LIST(...) {__VA_ARGS__}
#define CS_ALIGNED(grids, tables)                                                        \
    if (misaligned(CS_LIST grids, pb.dim == 2 ? 7 : 3)) return CS_ERR_INVALID;           \
    if (misaligned(CS_LIST tables, 15)) return CS_ERR_INVALID;

// zero-element tensors legitimately come with null data pointers
#define CS_NEED(...)                                                            \
    if (pb.d.S > 0 && pb.d.C > 0 && any_null({__VA_ARGS__})) return CS_ERR_INVALID;

// ---- 2D ----
int cs2d_forward(const float *input, const float *grid, const float *offset, float *output, int64_t N, int64_t C,
                 int64_t H, int64_t W, int64_t P, int padding_mode, int align_corners, int kernel, int multicell,
                 const float *input_cl, const void *plan, void *workspace, size_t workspace_bytes, void *stream) {
    CS_PROBLEM(2, 1)
    CS_NEED(input, grid, offset, output)
    CS_ALIGNED((grid), (input, input_cl, plan, workspace))
    pair_streams(pb, {output});
    if (tiled) return tiled_forward(pb, input, grid, offset, output, input_cl, workspace, workspace_bytes);
    if (pb.sdt) return CS_ERR_UNSUPPORTED;   // 16-bit streams: fast paths only (cs_half_streams_supported)
    return run_forward<2>(pb, table_, grid, offset, output);
}

int cs2d_backward(const float *grad_output, const float *input, const float *grid, const float *offset,
                  float *grad_input, float *grad_grid, int64_t N, int64_t C, int64_t H, int64_t W, int64_t P,
                  int padding_mode, int align_corners, int kernel, int multicell, const cs_cotangent_layout *layout, const float *input_cl,
                  const void *plan, void *workspace, size_t workspace_bytes, void *stream) {
    CS_PROBLEM(2, 1)
    CS_LAYOUT()
    CS_NEED(grad_output, input, grid, offset, grad_grid)
    CS_ALIGNED((grid, grad_grid), (input, grad_input, input_cl, plan, workspace))
    pair_streams(pb, {grad_output});
    if (tiled)
        return tiled_backward(pb, grad_output, input, grid, offset, grad_input, grad_grid, input_cl, plan, workspace,
                              workspace_bytes, g_leave);
    if (pb.sdt) return CS_ERR_UNSUPPORTED;
    if (rows && grad_input) {
        int rc = run_backward<2>(pb, grad_output, table_, grid, offset, nullptr, grad_grid);
        if (rc) return rc;
        return run_row_scatter<2, 0>(pb, grid, offset, grad_output, nullptr, nullptr, nullptr, grad_input, workspace,
                                     workspace_bytes);
    }
    return run_backward<2>(pb, grad_output, table_, grid, offset, grad_input, grad_grid);
}

int cs2d_backward_backward(const float *grad_out_input, const float *grad_out_grid, const float *input,
                           const float *grid, const float *grad_output, const float *offset, float *grad_input,
                           float *grad_grid, float *grad_grad_out, int64_t N, int64_t C, int64_t H, int64_t W,
                           int64_t P, int padding_mode, int align_corners, int kernel, int multicell, const cs_cotangent_layout *layout,
                           const float *input_cl, const void *plan, void *workspace, size_t workspace_bytes,
                           void *stream) {
    CS_PROBLEM(2, 1)
    CS_LAYOUT()
    CS_NEED(input, grid, grad_output, offset, grad_grid, grad_grad_out)   // grad_input may be NULL: not wanted
    CS_ALIGNED((grid, grad_grid, grad_out_grid), (input, grad_input, grad_out_input, input_cl, plan, workspace))
    // exact + grad_out_input: the grad_out_input -> grad_grid term is only in the direct kernel (run_bb)
    // (the row-atomic scatter needs C a power of two >= 2: C = 1, 3 keep the direct kernel, which scatters itself)
    pair_streams(pb, {grad_output, grad_grad_out});
    const bool exact_ci = tiled && pb.f.exact && grad_out_input;
    if (pb.acc_kind && exact_ci) return CS_ERR_UNSUPPORTED;   // that call leaves the fast path: it defines, it cannot add
    const bool via_rows = rows || (exact_ci && log2_exact(C) >= 1 && N <= 65535);
    if (pb.sdt && (!tiled || exact_ci)) return CS_ERR_UNSUPPORTED;
    if (tiled && !exact_ci)
        return tiled_bb(pb, grad_out_input, grad_out_grid, input, grid, grad_output, offset, grad_input, grad_grid,
                        grad_grad_out, input_cl, plan, workspace, workspace_bytes, g_sorted, g_leave);
    if (via_rows) {
        int rc = run_bb<2>(pb, grad_out_input, grad_out_grid, table_, grid, grad_output, offset, nullptr, grad_grid,
                           grad_grad_out);
        if (rc || !grad_input) return rc;
        return run_row_scatter<2, 1>(pb, grid, offset, grad_output, grad_out_grid, nullptr, nullptr, grad_input,
                                     workspace, workspace_bytes);
    }
    return run_bb<2>(pb, grad_out_input, grad_out_grid, table_, grid, grad_output, offset, grad_input, grad_grid,
                     grad_grad_out);
}

int cs2d_backward_backward_backward(const float *input, const float *grid, const float *grad_output,
                                    const float *grad_out_grid, const float *grad_out_ggrid, const float *offset,
                                    float *grad_input, float *grad_grad_out, int64_t N, int64_t C, int64_t H,
                                    int64_t W, int64_t P, int padding_mode, int align_corners, int kernel,
                                    int multicell, const cs_cotangent_layout *layout, const float *input_cl, const void *plan, void *workspace,
                                    size_t workspace_bytes, void *stream) {
    CS_PROBLEM(2, 1)
    CS_LAYOUT()
    CS_NEED(input, grid, grad_output, grad_out_grid, grad_out_ggrid, offset, grad_input, grad_grad_out)
    CS_ALIGNED((grid, grad_out_grid, grad_out_ggrid), (input, grad_input, input_cl, plan, workspace))
    pair_streams(pb, {grad_output, grad_grad_out});
    if (tiled)
        return tiled_bbb(pb, input, grid, grad_output, grad_out_grid, grad_out_ggrid, nullptr, offset, grad_input,
                         grad_grad_out, input_cl, plan, workspace, workspace_bytes, g_sorted, g_leave);
    if (pb.sdt) return CS_ERR_UNSUPPORTED;
    if (rows) {
        int rc = run_bbb<2>(pb, table_, grid, grad_output, grad_out_grid, grad_out_ggrid, nullptr, offset, nullptr,
                            grad_grad_out);
        if (rc) return rc;
        return run_row_scatter<2, 2>(pb, grid, offset, grad_output, grad_out_grid, grad_out_ggrid, nullptr, grad_input,
                                     workspace, workspace_bytes);
    }
    return run_bbb<2>(pb, table_, grid, grad_output, grad_out_grid, grad_out_ggrid, nullptr, offset, grad_input,
                      grad_grad_out);
}

int cs2d_bbb_fused(const float *input, const float *grid, const float *grad_output, const float *grad_out_grid,
                   const float *grad_out_ggrid, const float *grad_out_ggout, const float *offset, float *grad_input,
                   float *grad_grad_out, int64_t N, int64_t C, int64_t H, int64_t W, int64_t P, int padding_mode,
                   int align_corners, int kernel, int multicell, const cs_cotangent_layout *layout,
                   const float *input_cl, const void *plan,
                   void *workspace, size_t workspace_bytes, void *stream) {
    CS_PROBLEM(2, 1)
    CS_LAYOUT()
    CS_NEED(input, grid, grad_output, offset, grad_input, grad_grad_out)
    CS_ALIGNED((grid, grad_out_grid, grad_out_ggrid), (input, grad_input, input_cl, plan, workspace))
    pair_streams(pb, {grad_output, grad_grad_out, grad_out_ggout});
    if (tiled)
        return tiled_bbb(pb, input, grid, grad_output, grad_out_grid, grad_out_ggrid, grad_out_ggout, offset,
                         grad_input, grad_grad_out, input_cl, plan, workspace, workspace_bytes, g_sorted, g_leave);
    if (pb.sdt) return CS_ERR_UNSUPPORTED;
    if (rows) {
        int rc = run_bbb<2>(pb, table_, grid, grad_output, grad_out_grid, grad_out_ggrid, grad_out_ggout, offset,
                            nullptr, grad_grad_out);
        if (rc) return rc;
        return run_row_scatter<2, 2>(pb, grid, offset, grad_output, grad_out_grid, grad_out_ggrid, grad_out_ggout,
                                     grad_input, workspace, workspace_bytes);
    }
    return run_bbb<2>(pb, table_, grid, grad_output, grad_out_grid, grad_out_ggrid, grad_out_ggout, offset,
                      grad_input, grad_grad_out);
}

// ---- 3D ----
int cs3d_forward(const float *input, const float *grid, const float *offset, float *output, int64_t N, int64_t C,
                 int64_t D, int64_t H, int64_t W, int64_t P, int padding_mode, int align_corners, int kernel,
                 int multicell, const float *input_cl, const void *plan, void *workspace, size_t workspace_bytes,
                 void *stream) {
    CS_PROBLEM(3, D)
    CS_NEED(input, grid, offset, output)
    CS_ALIGNED((grid), (input, input_cl, plan, workspace))
    if (pb.d.S > 0 && pb.d.C > 0 && rows_cl_applies(3, N, C, P, pb.d.vol))
        return rcl_forward<3>(pb, input, grid, offset, output, input_cl, workspace, workspace_bytes);
    if (pb.sdt) return CS_ERR_UNSUPPORTED;
    return run_forward<3>(pb, table_, grid, offset, output);
}

int cs3d_backward(const float *grad_output, const float *input, const float *grid, const float *offset,
                  float *grad_input, float *grad_grid, int64_t N, int64_t C, int64_t D, int64_t H, int64_t W,
                  int64_t P, int padding_mode, int align_corners, int kernel, int multicell, const cs_cotangent_layout *layout, const float *input_cl,
                  const void *plan, void *workspace, size_t workspace_bytes, void *stream) {
    CS_PROBLEM(3, D)
    CS_LAYOUT()
    CS_NEED(grad_output, input, grid, offset, grad_grid)
    CS_ALIGNED((grid, grad_grid), (input, grad_input, input_cl, plan, workspace))
    if (pb.d.S > 0 && pb.d.C > 0 && rows_cl_applies(3, N, C, P, pb.d.vol))
        return rcl_backward<3>(pb, grad_output, input, grid, offset, grad_input, grad_grid, input_cl, plan, workspace,
                               workspace_bytes);
    if (pb.sdt) return CS_ERR_UNSUPPORTED;
    if (rows && grad_input) {
        int rc = run_backward<3>(pb, grad_output, table_, grid, offset, nullptr, grad_grid);
        if (rc) return rc;
        return run_row_scatter<3, 0>(pb, grid, offset, grad_output, nullptr, nullptr, nullptr, grad_input, workspace,
                                     workspace_bytes);
    }
    return run_backward<3>(pb, grad_output, table_, grid, offset, grad_input, grad_grid);
}

int cs3d_backward_backward(const float *grad_out_input, const float *grad_out_grid, const float *input,
                           const float *grid, const float *grad_output, const float *offset, float *grad_input,
                           float *grad_grid, float *grad_grad_out, int64_t N, int64_t C, int64_t D, int64_t H,
                           int64_t W, int64_t P, int padding_mode, int align_corners, int kernel, int multicell, const cs_cotangent_layout *layout,
                           const float *input_cl, const void *plan, void *workspace, size_t workspace_bytes,
                           void *stream) {
    CS_PROBLEM(3, D)
    CS_LAYOUT()
    CS_NEED(input, grid, grad_output, offset, grad_grid, grad_grad_out)   // grad_input may be NULL: not wanted
    CS_ALIGNED((grid, grad_grid, grad_out_grid), (input, grad_input, grad_out_input, input_cl, plan, workspace))
    if (pb.d.S > 0 && pb.d.C > 0 && rows_cl_applies(3, N, C, P, pb.d.vol))
        return rcl_bb<3>(pb, grad_out_input, grad_out_grid, input, grid, grad_output, offset, grad_input, grad_grid,
                         grad_grad_out, input_cl, plan, workspace, workspace_bytes);
    if (pb.sdt) return CS_ERR_UNSUPPORTED;
    if (rows) {
        int rc = run_bb<3>(pb, grad_out_input, grad_out_grid, table_, grid, grad_output, offset, nullptr, grad_grid,
                           grad_grad_out);
        if (rc || !grad_input) return rc;
        return run_row_scatter<3, 1>(pb, grid, offset, grad_output, grad_out_grid, nullptr, nullptr, grad_input,
                                     workspace, workspace_bytes);
    }
    return run_bb<3>(pb, grad_out_input, grad_out_grid, table_, grid, grad_output, offset, grad_input, grad_grid,
                     grad_grad_out);
}

int cs3d_backward_backward_backward(const float *input, const float *grid, const float *grad_output,
                                    const float *grad_out_grid, const float *grad_out_ggrid, const float *offset,
                                    float *grad_input, float *grad_grad_out, int64_t N, int64_t C, int64_t D,
                                    int64_t H, int64_t W, int64_t P, int padding_mode, int align_corners,
                                    int kernel, int multicell, const cs_cotangent_layout *layout, const float *input_cl, const void *plan,
                                    void *workspace, size_t workspace_bytes, void *stream) {
    CS_PROBLEM(3, D)
    CS_LAYOUT()
    CS_NEED(input, grid, grad_output, grad_out_grid, grad_out_ggrid, offset, grad_input, grad_grad_out)
    CS_ALIGNED((grid, grad_out_grid, grad_out_ggrid), (input, grad_input, input_cl, plan, workspace))
    if (pb.d.S > 0 && pb.d.C > 0 && rows_cl_applies(3, N, C, P, pb.d.vol))
        return rcl_bbb<3>(pb, input, grid, grad_output, grad_out_grid, grad_out_ggrid, nullptr, offset, grad_input,
                          grad_grad_out, input_cl, plan, workspace, workspace_bytes);
    if (pb.sdt) return CS_ERR_UNSUPPORTED;
    if (rows) {
        int rc = run_bbb<3>(pb, table_, grid, grad_output, grad_out_grid, grad_out_ggrid, nullptr, offset, nullptr,
                            grad_grad_out);
        if (rc) return rc;
        return run_row_scatter<3, 2>(pb, grid, offset, grad_output, grad_out_grid, grad_out_ggrid, nullptr, grad_input,
                                     workspace, workspace_bytes);
    }
    return run_bbb<3>(pb, table_, grid, grad_output, grad_out_grid, grad_out_ggrid, nullptr, offset, grad_input,
                      grad_grad_out);
}

int cs3d_bbb_fused(const float *input, const float *grid, const float *grad_output, const float *grad_out_grid,
                   const float *grad_out_ggrid, const float *grad_out_ggout, const float *offset, float *grad_input,
                   float *grad_grad_out, int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int64_t P,
                   int padding_mode, int align_corners, int kernel, int multicell, const cs_cotangent_layout *layout, const float *input_cl,
                   const void *plan, void *workspace, size_t workspace_bytes, void *stream) {
    CS_PROBLEM(3, D)
    CS_LAYOUT()
    CS_NEED(input, grid, grad_output, offset, grad_input, grad_grad_out)
    CS_ALIGNED((grid, grad_out_grid, grad_out_ggrid), (input, grad_input, input_cl, plan, workspace))
    if (pb.d.S > 0 && pb.d.C > 0 && rows_cl_applies(3, N, C, P, pb.d.vol))
        return rcl_bbb<3>(pb, input, grid, grad_output, grad_out_grid, grad_out_ggrid, grad_out_ggout, offset,
                          grad_input, grad_grad_out, input_cl, plan, workspace, workspace_bytes);
    if (pb.sdt) return CS_ERR_UNSUPPORTED;
    if (rows) {
        int rc = run_bbb<3>(pb, table_, grid, grad_output, grad_out_grid, grad_out_ggrid, grad_out_ggout, offset,
                            nullptr, grad_grad_out);
        if (rc) return rc;
        return run_row_scatter<3, 2>(pb, grid, offset, grad_output, grad_out_grid, grad_out_ggrid, grad_out_ggout,
                                     grad_input, workspace, workspace_bytes);
    }
    return run_bbb<3>(pb, table_, grid, grad_output, grad_out_grid, grad_out_ggrid, grad_out_ggout, offset,
                      grad_input, grad_grad_out);
}

int cs2d_bbb_grid(const float *input, const float *grid, const float *grad_output, const float *grad_out_grid,
                  const float *grad_out_ggrid, const float *grad_out_ggout, const float *offset, float *grad_grid3,
                  int64_t N, int64_t C, int64_t H, int64_t W, int64_t P, int padding_mode, int align_corners,
                  int kernel, int multicell, const cs_cotangent_layout *layout, void *stream) {
    Problem pb;
    int rc = make_problem(pb, 2, N, C, 1, H, W, P, padding_mode, align_corners, kernel, multicell, stream);
    if (rc) return rc;
    bool g_sorted = false, g_leave = false;
    const bool tiled = false;     // (this entry point produces no input-shaped gradient: nothing to accumulate)
    CS_LAYOUT()
    (void)g_sorted; (void)g_leave;
    CS_NEED(input, grid, grad_output, grad_out_grid, offset, grad_grid3)
    CS_ALIGNED((grid, grad_out_grid, grad_out_ggrid, grad_grid3), (input))
    if (pb.sdt) return CS_ERR_UNSUPPORTED;
    return bbb_grid_impl<2>(pb, input, grid, grad_output, grad_out_grid, grad_out_ggrid, grad_out_ggout, offset, grad_grid3);
}

int cs3d_bbb_grid(const float *input, const float *grid, const float *grad_output, const float *grad_out_grid,
                  const float *grad_out_ggrid, const float *grad_out_ggout, const float *offset, float *grad_grid3,
                  int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int64_t P, int padding_mode,
                  int align_corners, int kernel, int multicell, const cs_cotangent_layout *layout, void *stream) {
    Problem pb;
    int rc = make_problem(pb, 3, N, C, D, H, W, P, padding_mode, align_corners, kernel, multicell, stream);
    if (rc) return rc;
    bool g_sorted = false, g_leave = false;
    const bool tiled = false;     // (this entry point produces no input-shaped gradient: nothing to accumulate)
    CS_LAYOUT()
    (void)g_sorted; (void)g_leave;
    CS_NEED(input, grid, grad_output, grad_out_grid, offset, grad_grid3)
    CS_ALIGNED((grid, grad_out_grid, grad_out_ggrid, grad_grid3), (input))
    if (pb.sdt) return CS_ERR_UNSUPPORTED;
    return bbb_grid_impl<3>(pb, input, grid, grad_output, grad_out_grid, grad_out_ggrid, grad_out_ggout, offset, grad_grid3);
}

}  // extern "C"
