/*
 * cosine_sampler.h -- C ABI of libcosine_sampler_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the hot path of NamGyuKang/CosineSampler: the four native entry points
 * per dimensionality that the reference exposes through pybind11 as `_cosine_2d` / `_cosine_3d`
 *   reference cosine_sampler_2d/csrc/cosine_sampler_2d.cpp:130-135   (2D)
 *   reference cosine_sampler_3d/csrc/cosine_sampler_3d.cpp:133-138   (3D)
 * restated as plain C: raw device pointers, explicit sizes, integer flags, a HIP stream handle.
 * No C++ or torch types cross this boundary, nothing is allocated, freed or synchronised inside,
 * every call only ENQUEUES work on `stream` (graph-capturable).
 *
 * Conventions
 *   - All tensors are contiguous fp32 in device memory (the reference's CHECK_CUDA /
 *     CHECK_CONTIGUOUS, 2d.cpp:4-6); the channel-major streams may be half / bfloat16 instead (CS_STREAM_F16 /
 *     CS_STREAM_BF16).  fp64 is not built: the Python layer converts.
 *   - 2D: input (N,C,H,W), grid (N,Ho,Wo,2), out/gOut (N,C,Ho,Wo);  P = Ho*Wo.
 *     3D: input (N,C,D,H,W), grid (N,Do,Ho,Wo,3), out/gOut (N,C,Do,Ho,Wo);  P = Do*Ho*Wo.
 *     grid[...,0] addresses W, [...,1] H, [...,2] D, each in [-1,1].
 *   - `offset` = N floats, the per-cell "multicell" shift the reference's Python builds with
 *     torch.linspace(0, 1-1/N, N) (modules_2d.py:24-27); zeros when multicell == 0.
 *   - padding_mode: 0 zeros, 1 border, 2 reflection (modules_2d.py:4-10)
 *     kernel:       0 cosine, 1 linear, 2 smoothstep (modules_2d.py:12-18)
 *   - Outputs are caller-allocated and need NO pre-zeroing: every element of every output is
 *     defined by the call (the reference needs zeros_like for grad_input / ggOut, 2d.cpp:75,99-101).
 *   - Nullable pointers replace the reference's `input_requires_grad` flags; see each function.
 *   - `workspace`: scratch device memory, at least cs_workspace_bytes(...) bytes, 256-byte
 *     aligned, owned by the caller, contents undefined before and after.  May be NULL when the
 *     query returns 0.
 *   - `input_cl`, `plan` (both nullable): prepared objects that let consecutive stages of one
 *     training step share work.  `input_cl` is the channels-last copy of `input` made by
 *     cs_pack_input(); `plan` is the point-binning plan made by cs2d_plan_build() for THIS grid,
 *     offset, sizes and flags.  The library cannot check that they match (they live in device
 *     memory and nothing here synchronises): passing a stale one is undefined behaviour.  When
 *     NULL, a stage that wants them builds them inside `workspace` (cs_workspace_bytes accounts
 *     for it).  Problems outside the fast paths (C > 32 in 2D / > 16 in 3D -- smaller counts run zero-padded to 4, 8, 16
 *     or 32 channels --, tiny S) ignore both.
 *   - grid (N,...,dim); with CS_GRID_BROADCAST in `kernel` a single (...,dim) set of points shared by every n.
 *   - Alignment: `grid`-shaped tensors (grid, grad_grid, grad_out_grid, grad_out_ggrid) on 8 bytes in 2D, 4 in 3D; `input`-shaped
 *     tensors (input, grad_input, grad_out_input), `input_cl`, `plan` and `workspace` on 16 bytes; streams on their
 *     element size.  The kernels use 8- and 16-byte accesses on them; a violation returns CS_ERR_INVALID.
 *   - Return value: 0 on success, a negative CS_ERR_* for argument errors, or a positive
 *     hipError_t from the launch.  cs_error_string() describes either.
 *   - Thread-safe and re-entrant: the library keeps no mutable global state.  (The two testing knobs at the end of
 *     this header, cs_debug_force_path and cs_debug_coherent_tuning, are inert unless the process was started with
 *     COSINESAMPLER_DEBUG=1 in its environment -- read once, when the library is first used -- and the switches of the
 *     latter that make results wrong exist only in libraries built with -DCS_COH_DEBUG.)
 */
#ifndef COSINE_SAMPLER_H
#define COSINE_SAMPLER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CS_ABI_VERSION 15

enum { CS_OK = 0, CS_ERR_INVALID = -1, CS_ERR_UNSUPPORTED = -2, CS_ERR_WORKSPACE = -3 };
enum { CS_PAD_ZEROS = 0, CS_PAD_BORDER = 1, CS_PAD_REFLECTION = 2 };
enum { CS_KERNEL_COSINE = 0, CS_KERNEL_LINEAR = 1, CS_KERNEL_SMOOTHSTEP = 2 };
/* OR-ed into `kernel` (not in the reference): keep the MIXED second derivatives d2W/dg_j dg_k, j != k, that the
 * reference drops -- in the 2D second backward (grad_grid, 2d.cu:705-706; also its grad_out_input -> grad_grid term,
 * 2d.cu has none, 3d.cu:837-839 does) and in the third backward of both dimensionalities (2d.cu:833-834,
 * 3d.cu:1008-1010).  With it u_xy and d(u_xy)/d(input) obtained through autograd are the exact derivatives of the
 * interpolant; without it (default) results are the reference's. */
#define CS_KERNEL_EXACT_MIXED 0x100
/* OR-ed into `kernel` (the reference dispatches half too, 2d.cu:905, :948, :1009, :1076): the channel-major STREAMS --
 * output, grad_output, grad_grad_out, grad_out_ggout, the tensors of 4*N*C*P bytes each -- hold IEEE half / bfloat16
 * instead of fp32; their pointers are then `_Float16*` / `__bf16*` passed through the `float*` parameters.  Weights,
 * sums and every other tensor (input, grid and its cotangents, the input-shaped gradients) stay fp32, as the reference
 * evaluates its weights in float whatever the dispatch type (2d.cu:239-261).  Native on the fast paths only
 * (cs_half_streams_supported); elsewhere the calls return CS_ERR_UNSUPPORTED and the caller converts. */
#define CS_STREAM_F16 0x1000
#define CS_STREAM_BF16 0x2000
/* OR-ed into `kernel` (and the `flags` of the plan builders; not in the reference, whose callers materialise
 * grid.repeat(N, 1, 1, 1), test/test_2d.py:38): ONE set of P points serves every n.  `grid`, `grad_out_grid` and
 * `grad_out_ggrid` are then [P, dim] arrays; the per-point OUTPUTS (grad_grid of the first and second backward) are
 * still written per n, [N, P, dim] -- the gradient with respect to the shared points is their sum over n, which the
 * caller takes (cosinesampler_amd/ops.py does).  Every path accepts it. */
#define CS_GRID_BROADCAST 0x4000
/* OR-ed into `kernel` (a HINT about the caller's data, never about the result): consecutive points of `grid` fall into
 * the same or neighbouring cells -- the order cs2d_sort_points produces, which a PIXEL-style caller, whose collocation
 * points are a fixed set re-used every step (test/test_2d.py:28-38), establishes once at set-up.  The 2D backward stages
 * that produce grad_input then run on kernels that keep everything on chip (cs_coherent.cuh: a wave reduces runs of
 * equal cell in registers and adds them to a private LDS image of the tile it is walking; the reference's per-sample
 * atomics, 2d.cu:464-505, :661-712, :850-888): no plan, no records, about a third of the time.  Results are the same for
 * ANY order of the points (up to the summation order of fp32 adds); a wrong hint only costs time -- an unordered set
 * empties its window at almost every sample.  cs_points_tile_changes measures an order.  Every 2D stage of the fast path
 * has a coherent kernel (forward and the stages without grad_input included); ignored where it does not apply (3D, the
 * second backward with grad_out_input, problems outside the 2D fast path). */
#define CS_POINTS_COHERENT 0x8000
/* OR-ed into `kernel`, together with CS_GRID_BROADCAST (not in the reference): the PIXEL pattern
 * features = sampler(cells, grid.repeat(N,1,1,1)).sum(0) (test/test_2d.py:38, :51) as ONE op.  Every per-point tensor loses
 * its n: the inputs `grid`, `grad_out_grid`, `grad_out_ggrid` are [P,2] and `grad_output`, `grad_out_ggout` are [C,P] (one
 * cotangent for every table: pass n-strides 0 in cs_cotangent_layout); the per-point RESULTS come back summed over the
 * tables -- `output` and `grad_grad_out` [C,P], `grad_grid` [P,2].  Input-shaped gradients stay [N,C,H,W].  Equal to the plain
 * op on repeated / expanded inputs followed by sums over n, without the N-fold streams in between.  2D: runs on the
 * coherent-points kernels (fast for points in cell order, correct for any; fp32 streams, zeros padding with align_corners;
 * workspace as for CS_STAGE_POINTS_COHERENT).  3D (round 4): the channels-last point kernels walk the tables per point, the
 * scatter is the plain op's (plan, records, tile / cell / row-atomic scatter: workspace as for the plain call).
 * cs_sum_over_n_supported says where; CS_ERR_UNSUPPORTED otherwise -- the caller then sums himself. */
#define CS_SUM_OVER_N 0x10000
/* stage ids for cs_workspace_bytes */
enum { CS_STAGE_FORWARD = 0, CS_STAGE_BACKWARD = 1, CS_STAGE_BACKWARD_BACKWARD = 2, CS_STAGE_BBB_FUSED = 3 };
/* OR-ed into the stage id: the call will pass grad_input == NULL (first / second backward only) -- nothing is
 * scattered, so neither a plan nor scatter scratch is needed (every derivative a PINN takes with
 * autograd.grad(u, x, create_graph=True) is such a call) */
#define CS_STAGE_NO_GRAD_INPUT 0x10
/* OR-ed into the stage id: the call will carry CS_POINTS_COHERENT (no plan; one channels-last accumulator of scratch) */
#define CS_STAGE_POINTS_COHERENT 0x20
/* OR-ed into the stage id: the call will carry cs_cotangent_layout.accumulate_grad_input (the accumulator is the
 * caller's: no scratch accumulator) */
#define CS_STAGE_ACCUMULATE 0x40

/* How the channel-major cotangents of a backward stage lie in memory.  The reference demands contiguous
 * (N,C,[Do,]Ho,Wo) tensors (CHECK_CONTIGUOUS, 2d.cpp:5), so PIXEL-style callers, which sum the sampled features
 * over n before the MLP, pay a 4*N*C*P-byte `.contiguous()` copy of an *expanded* gradient on every backward
 * call.  Here the n-stride of those tensors is a parameter: 0 = one (C,P) block shared by every n (what
 * `Tensor.expand` produces), C*P = contiguous; within one n the block is always contiguous (C,P).
 * A NULL layout pointer means contiguous. */
typedef struct cs_cotangent_layout {
    int64_t grad_output_stride_n;      /* elements between consecutive n of grad_output */
    int64_t grad_out_ggout_stride_n;   /* same for grad_out_ggout (cs*_bbb_fused only) */
    /* Non-zero: the caller's `plan` already holds the cell-sorted copy of THIS grad_output (2D walker plans only;
     * ignored elsewhere and when plan == NULL).  The backward stages of one training step are handed the same
     * grad_output (modules_2d.py:62, :95 save and re-use it); the first stage that scatters with a caller's plan leaves
     * its rows in the plan in sorted order, and a later stage that is told so streams them instead of writing and
     * fetching them again.  The library cannot check (the bytes live in device memory): set it only if an earlier
     * call with this plan was given this very tensor, unchanged, and a non-NULL grad_input.  Zero is always safe. */
    int32_t sorted_grad_output_valid;
    /* Non-zero: a stage that fetches grad_output's rows by sample id anyway should also leave them in the plan in
     * sorted order (one more sequential 4*N*C*P-byte write: +0.25 ms at N=16 C=16 P=2^20) because a later stage will be
     * called with sorted_grad_output_valid.  Worth it from the second use on; a stage that does not leave its copy does
     * not touch the one the plan holds. */
    int32_t leave_sorted_grad_output;
    /* Elements between consecutive n of the stream the stage WRITES, grad_grad_out (second / third backward); 0 = contiguous
     * (C*P).  Larger when grad_grad_out is a channel range of a wider (N, C_total, P) tensor -- how tables with more channels
     * than the fast paths hold are run as channel groups without gathering the groups' results afterwards. */
    int64_t grad_grad_out_stride_n;
    /* Non-zero (CS_ACC_NCHW or CS_ACC_CHANNELS_LAST, what cs_accumulator_kind returned for this problem): `grad_input`
     * is not a result to be defined but the step's ACCUMULATOR -- cs_accumulator_bytes bytes that the caller zeroed
     * before the first stage of the step -- and the stage ADDS its input-shaped gradient to it: no clear, no layout
     * conversion.  A training step wants the SUM of the gradients its backward stages produce (the autograd engine adds
     * them up into cells.grad; a multi-GPU step all-reduces that sum once): with the stages adding into one
     * buffer a step pays one clear and one conversion (cs_accumulator_finish) instead of one per stage and no adding
     * passes.  Replaces the zeros_like + per-stage result of 2d.cpp:75, :99, :119.  A stage whose path keeps its sums in
     * the other layout returns CS_ERR_UNSUPPORTED (nothing has been written: the caller runs it without the flag and adds). */
    int32_t accumulate_grad_input;
    int32_t reserved_;
} cs_cotangent_layout;

/* ---- the step accumulator (not in the reference) ---------------------------------------------------------------------
 * cs_accumulator_kind: in which layout the backward stages of this problem can ADD into a caller-held accumulator --
 *   CS_ACC_NONE           not at all on this path (3D fast paths, channel counts beyond the fast path, ...)
 *   CS_ACC_NCHW           the caller's own (N,C,[D,]H,W) layout: the accumulator can be the final tensor itself
 *   CS_ACC_CHANNELS_LAST  a padded channels-last image (the coherent-points kernels, CS_POINTS_COHERENT in `kernel`)
 * `kernel` = the kernel argument of the stage calls, flags included.  cs_accumulator_bytes: its size.
 * cs_accumulator_finish: the accumulated sum in the caller's layout, `grad_input` (N,C,[D,]H,W); for CS_ACC_NCHW a copy
 * unless acc == grad_input.  Enqueues on `stream` like everything else. */
enum { CS_ACC_NONE = 0, CS_ACC_NCHW = 1, CS_ACC_CHANNELS_LAST = 2 };
int cs_accumulator_kind(int dim, int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int64_t P, int kernel);
size_t cs_accumulator_bytes(int dim, int kind, int64_t N, int64_t C, int64_t D, int64_t H, int64_t W);
int cs_accumulator_finish(int dim, int kind, const float *acc, float *grad_input, int64_t N, int64_t C, int64_t D,
                          int64_t H, int64_t W, void *stream);

int cs_abi_version(void);
const char *cs_error_string(int code);

/* Scratch bytes stage `stage` needs for this problem.  dim = 2 or 3; D is ignored for dim 2.
 * have_input_cl / have_plan: the caller will pass those prepared objects; have_cI: the
 * backward_backward call will carry a grad_out_input (it needs its own channels-last copy). */
size_t cs_workspace_bytes(int dim, int stage, int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int64_t P,
                          int have_input_cl, int have_plan, int have_cI);

/* 1 if the summing kernels of CS_SUM_OVER_N are built for this problem, else 0 (the caller sums himself).  2D: the
 * coherent-points kernels (fp32 streams, zeros padding with align_corners).  3D (round 4): the channels-last point kernels
 * walking the N tables per point (fast path: C <= 16, N > 1, fp32 streams; every padding mode, any order of the points). */
int cs2d_sum_over_n_supported(int64_t N, int64_t C, int64_t H, int64_t W, int64_t P, int padding_mode, int align_corners);
int cs_sum_over_n_supported(int dim, int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int64_t P, int padding_mode,
                            int align_corners);
/* 1 if this problem runs on a path whose kernels take 16-bit streams (CS_STREAM_F16 / CS_STREAM_BF16), else 0. */
int cs_half_streams_supported(int dim, int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int64_t P);

/* Channels-last copy (N,spatial...,CP) of an (N,C,spatial...) tensor, CP = C padded with zero channels to 4, 8, 16 or 32
 * (2D tables with 1..3 channels run as one float4 quad).  In 3D the copy is z-paired -- every node carries its own row and
 * the row of the node one z-plane above, twice the bytes -- so that the 2x2 rows a sample needs per y are contiguous.
 * Returns the byte size of the copy / makes it; an opaque object for the stage calls' `input_cl`.  cs_pack_bytes returns 0
 * when no fast path applies. */
size_t cs_pack_bytes(int dim, int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int64_t P);
int cs_pack_input(int dim, const float *input, float *input_cl, int64_t N, int64_t C, int64_t D, int64_t H,
                  int64_t W, void *stream);

/* Point-binning plan of one grid (2D fast path): the sample ids sorted by (n, 16x16-cell tile, cell)
 * with the first position of every tile and of every cell.  Depends on grid, offset, N, H, W, P and
 * the three flags, not on the blending kernel (C only decides whether the fast path applies).
 * cs2d_plan_bytes returns 0 when the fast path does not apply. */
size_t cs2d_plan_bytes(int64_t N, int64_t C, int64_t H, int64_t W, int64_t P);
int cs2d_plan_build(const float *grid, const float *offset, void *plan, size_t plan_bytes,
                    int64_t N, int64_t C, int64_t H, int64_t W, int64_t P,
                    int padding_mode, int align_corners, int multicell, int flags /* 0 or CS_GRID_BROADCAST */,
                    void *stream);

/* 1 if a plan of this problem is the kind that can hold the cell-sorted copy of grad_output (cs_cotangent_layout:
 * walker plans; crowded tables bin by cell and keep none), else 0: a caller only sets sorted_grad_output_valid after a
 * stage that was asked to leave the copy on such a plan. */
int cs2d_plan_keeps_sorted_copy(int64_t N, int64_t C, int64_t H, int64_t W, int64_t P);

/* The same for 3D (C <= 16, run zero-padded to 4, 8 or 16 channels).  Two kinds of plan, chosen by the sizes: small
 * crowded tables (cells = (D+1)(H+1)(W+1) <= 40000 and P >= 8 cells: the reference's test_3d.py shapes) -- samples binned
 * by cell; other tables of up to 12288 tiles of 16x4x4 nodes, P < 2^23 per table (BASELINE configs[3]) -- every sample
 * listed in the tiles that own its corner nodes.  cs3d_plan_bytes returns 0 where neither applies (row atomics). */
size_t cs3d_plan_bytes(int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int64_t P);
int cs3d_plan_build(const float *grid, const float *offset, void *plan, size_t plan_bytes,
                    int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int64_t P,
                    int padding_mode, int align_corners, int multicell, int flags /* 0 or CS_GRID_BROADCAST */,
                    void *stream);

/* ---- ordering a point set (not in the reference; set-up time, not the per-step path) ------------------------------
 * cs{2,3}d_sort_points orders P points by the cell of table 0 (offset 0) they fall into -- by 8-cell tiles, then by
 * cell inside the tile, points that touch no node last; equal cells keep the caller's order (stable, reproducible).
 *   perm[j]          = index of the point that comes j-th          (int32, P entries)
 *   sorted_points[j] = points[perm[j]]                              ([P, dim] floats)
 * A caller orders its collocation points ONCE with this, keeps `perm` if it has per-point data to carry along, and
 * passes CS_POINTS_COHERENT from then on.  `workspace`: cs_sort_points_bytes(P) bytes of device scratch.
 * cs_points_tile_changes counts, on the device, how often the tile changes between consecutive points (one uint32 at
 * `count`, device memory; P for an unordered set, about the number of occupied tiles for an ordered one): the hint pays
 * when changes * 256 <= P.  Nothing here synchronises; the caller reads `count` when it likes. */
size_t cs_sort_points_bytes(int64_t P);
int cs2d_sort_points(const float *points, float *sorted_points, int32_t *perm, int64_t P, int64_t H, int64_t W,
                     int padding_mode, int align_corners, int multicell, void *workspace, size_t workspace_bytes,
                     void *stream);
int cs3d_sort_points(const float *points, float *sorted_points, int32_t *perm, int64_t P, int64_t D, int64_t H, int64_t W,
                     int padding_mode, int align_corners, int multicell, void *workspace, size_t workspace_bytes,
                     void *stream);
int cs_points_tile_changes(int dim, const float *points, uint32_t *count, int64_t P, int64_t D, int64_t H, int64_t W,
                           int padding_mode, int align_corners, int multicell, void *stream);
/* Per-point data carried along with a permutation of the points: out[r][j][k] = in[r][index[j]][k] for `rows` arrays of P
 * elements of `width` (1..4) floats -- a (C,P) stream has rows = C, width = 1; a [P,2] grid-shaped tensor rows = 1,
 * width = 2.  `index` = the perm of cs2d_sort_points (into cell order) or its inverse (back).  How the summed op
 * (CS_SUM_OVER_N) serves points in the order they were drawn: its per-point tensors are N times smaller than the plain op's. */
int cs_carry_points(const float *in, float *out, const int32_t *index, int64_t rows, int64_t P, int width, void *stream);
/* The same count taken on a SAMPLE: `segments` (1..65536) runs of 1024 consecutive points spread evenly over the set (all
 * of it when P <= 1024 * segments) -- a few microseconds whatever P is, and unlike a prefix it sees every part of the set.
 * The hint pays when changes * 256 <= min(P, 1024 * segments). */
int cs_points_tile_changes_sampled(int dim, const float *points, uint32_t *count, int64_t P, int64_t D, int64_t H,
                                   int64_t W, int padding_mode, int align_corners, int multicell, int segments,
                                   void *stream);
/* Tuning / experiments on the coherent kernels: samples_per_wave (a multiple of 64; 0: the per-stage defaults) and ablation_bits,
 * which switch parts of the kernels OFF to see what each costs (results are then wrong): 1 no scatter-reduce, 2 no window
 * flush, 4 no products / outputs; 0 = the product.  Process-wide.  INERT unless COSINESAMPLER_DEBUG=1 was in the
 * environment when the library was first used, and the ablation bits only exist in a library built with -DCS_COH_DEBUG
 * (tools/ab.sh): no call can make the shipped library compute wrong results.  Returns 1 if the call took effect. */
int cs_debug_coherent_tuning(int samples_per_wave, int ablation_bits);

/* Testing knob: 0 = choose the path from the shapes (default), 1 = always the direct (atomics)
 * kernels, 2 = the fast paths wherever they are implemented, whatever the size, 3 = as 2 but crowded tables
 * go through the tile walkers, not the wave-per-cell kernel, 4 = as 2 without the re-use of the sorted grad_output
 * copy between the stages of a step, 5 = as 0 but CS_POINTS_COHERENT is ignored (A/B of the hint), 6 = as 2 but 3D tables
 * are packed by the two-reads kernel instead of the column-wise one (A/B of cs_pack_input).  Every mode computes the
 * same results.  Process-wide; INERT unless COSINESAMPLER_DEBUG=1 was in the environment when the library was first
 * used (the test suite sets it).  Returns 1 if the call took effect. */
int cs_debug_force_path(int mode);

/* ---- 2D -------------------------------------------------------------------------------- */

/* Replaces `_cosine_2d.forward` (2d.cpp:47-62 -> launch_cosine_sampler_forward_kernel, 2d.cu:897).
 * NB the reference 2D forward ignores align_corners (2d.cu:307-308); so does this, for parity. */
int cs2d_forward(const float *input, const float *grid, const float *offset, float *output,
                 int64_t N, int64_t C, int64_t H, int64_t W, int64_t P,
                 int padding_mode, int align_corners, int kernel, int multicell,
                 const float *input_cl, const void *plan, void *workspace, size_t workspace_bytes, void *stream);

/* Replaces `_cosine_2d.backward` (2d.cpp:64-85 -> 2d.cu:938).
 * grad_input == NULL  <=>  input_requires_grad == false (2d.cpp:73-79). */
int cs2d_backward(const float *grad_output, const float *input, const float *grid, const float *offset,
                  float *grad_input /* nullable */, float *grad_grid,
                  int64_t N, int64_t C, int64_t H, int64_t W, int64_t P,
                  int padding_mode, int align_corners, int kernel, int multicell, const cs_cotangent_layout *layout /* nullable */,
                  const float *input_cl, const void *plan, void *workspace, size_t workspace_bytes, void *stream);

/* Replaces `_cosine_2d.backward_backward` (2d.cpp:87-106 -> 2d.cu:990).
 * grad_out_input == NULL <=> input_requires_grad == false (modules_2d.py:87-89).
 * grad_out_grid == NULL is read as all zeros.
 * grad_input == NULL: the caller has no use for d/d input of this stage (the reference always computes it and
 * autograd throws it away when only d/d grid is asked for, e.g. u_xx = grad(u_x, x)); the scatter half is skipped. */
int cs2d_backward_backward(const float *grad_out_input /* nullable */, const float *grad_out_grid /* nullable */,
                           const float *input, const float *grid, const float *grad_output, const float *offset,
                           float *grad_input /* nullable */, float *grad_grid, float *grad_grad_out,
                           int64_t N, int64_t C, int64_t H, int64_t W, int64_t P,
                           int padding_mode, int align_corners, int kernel, int multicell, const cs_cotangent_layout *layout /* nullable */,
                           const float *input_cl, const void *plan, void *workspace, size_t workspace_bytes, void *stream);

/* Replaces `_cosine_2d.backward_backward_backward` (2d.cpp:108-127 -> 2d.cu:1058). */
int cs2d_backward_backward_backward(const float *input, const float *grid, const float *grad_output,
                                    const float *grad_out_grid, const float *grad_out_ggrid, const float *offset,
                                    float *grad_input, float *grad_grad_out,
                                    int64_t N, int64_t C, int64_t H, int64_t W, int64_t P,
                                    int padding_mode, int align_corners, int kernel, int multicell, const cs_cotangent_layout *layout /* nullable */,
                                    const float *input_cl, const void *plan, void *workspace, size_t workspace_bytes, void *stream);

/* The whole of CosineSamplerBackwardBackward.backward (modules_2d.py:98-111) in one pass: the
 * kernel above PLUS the reference's second backward_backward launch with gOut := grad_out_ggout,
 * gOutInput := ones, of which only grad_input is kept and added.
 *   grad_input[n,c,q]   = sum_s  gOut*E_a + grad_out_ggout*D_a
 *   grad_grad_out[n,c,s] = sum_a input[q_a]*E_a
 * grad_out_ggrid == NULL and/or grad_out_ggout == NULL are read as all zeros. */
int cs2d_bbb_fused(const float *input, const float *grid, const float *grad_output,
                   const float *grad_out_grid, const float *grad_out_ggrid /* nullable */,
                   const float *grad_out_ggout /* nullable */, const float *offset,
                   float *grad_input, float *grad_grad_out,
                   int64_t N, int64_t C, int64_t H, int64_t W, int64_t P,
                   int padding_mode, int align_corners, int kernel, int multicell, const cs_cotangent_layout *layout /* nullable */,
                   const float *input_cl, const void *plan, void *workspace, size_t workspace_bytes, void *stream);

/* ---- 3D: same contracts; 3d.cpp:50-131 -> 3d.cu:1073,1115,1167,1241; modules_3d.py:87-100 --- */

int cs3d_forward(const float *input, const float *grid, const float *offset, float *output,
                 int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int64_t P,
                 int padding_mode, int align_corners, int kernel, int multicell,
                 const float *input_cl, const void *plan, void *workspace, size_t workspace_bytes, void *stream);

int cs3d_backward(const float *grad_output, const float *input, const float *grid, const float *offset,
                  float *grad_input /* nullable */, float *grad_grid,
                  int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int64_t P,
                  int padding_mode, int align_corners, int kernel, int multicell, const cs_cotangent_layout *layout /* nullable */,
                  const float *input_cl, const void *plan, void *workspace, size_t workspace_bytes, void *stream);

int cs3d_backward_backward(const float *grad_out_input /* nullable */, const float *grad_out_grid /* nullable */,
                           const float *input, const float *grid, const float *grad_output, const float *offset,
                           float *grad_input /* nullable */, float *grad_grid, float *grad_grad_out,
                           int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int64_t P,
                           int padding_mode, int align_corners, int kernel, int multicell, const cs_cotangent_layout *layout /* nullable */,
                           const float *input_cl, const void *plan, void *workspace, size_t workspace_bytes, void *stream);

int cs3d_backward_backward_backward(const float *input, const float *grid, const float *grad_output,
                                    const float *grad_out_grid, const float *grad_out_ggrid, const float *offset,
                                    float *grad_input, float *grad_grad_out,
                                    int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int64_t P,
                                    int padding_mode, int align_corners, int kernel, int multicell, const cs_cotangent_layout *layout /* nullable */,
                                    const float *input_cl, const void *plan, void *workspace, size_t workspace_bytes, void *stream);

int cs3d_bbb_fused(const float *input, const float *grid, const float *grad_output,
                   const float *grad_out_grid, const float *grad_out_ggrid /* nullable */,
                   const float *grad_out_ggout /* nullable */, const float *offset,
                   float *grad_input, float *grad_grad_out,
                   int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int64_t P,
                   int padding_mode, int align_corners, int kernel, int multicell, const cs_cotangent_layout *layout /* nullable */,
                   const float *input_cl, const void *plan, void *workspace, size_t workspace_bytes, void *stream);

/* ---- not in the reference: third-order gradient w.r.t. grid ----------------------------------
 * The reference's third backward returns no gradient for `grid` (modules_2d.py:111, modules_3d.py:100), so u_xxx or
 * u_xxy cannot be had from it.  This is d/dgrid of  <grad_grid2, grad_out_ggrid> + <grad_grad_out, grad_out_ggout>
 * where (grad_grid2, grad_grad_out) are the second backward's outputs with every mixed term (CS_KERNEL_EXACT_MIXED
 * semantics, whatever the flag in `kernel`) and grad_out_input absent.  grad_grid3 has grid's shape.  Needs no workspace. */
int cs2d_bbb_grid(const float *input, const float *grid, const float *grad_output, const float *grad_out_grid,
                  const float *grad_out_ggrid /* nullable */, const float *grad_out_ggout /* nullable */,
                  const float *offset, float *grad_grid3,
                  int64_t N, int64_t C, int64_t H, int64_t W, int64_t P,
                  int padding_mode, int align_corners, int kernel, int multicell,
                  const cs_cotangent_layout *layout /* nullable */, void *stream);
int cs3d_bbb_grid(const float *input, const float *grid, const float *grad_output, const float *grad_out_grid,
                  const float *grad_out_ggrid /* nullable */, const float *grad_out_ggout /* nullable */,
                  const float *offset, float *grad_grid3,
                  int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int64_t P,
                  int padding_mode, int align_corners, int kernel, int multicell,
                  const cs_cotangent_layout *layout /* nullable */, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* COSINE_SAMPLER_H */
