// cs_coherent.cuh -- the 2D stages for COHERENT point sets (CS_POINTS_COHERENT): consecutive samples of one table
// fall into the same or neighbouring cells, which is what a caller gets by ordering its collocation points once
// (cs2d_sort_points; PIXEL draws a fixed random set and re-uses it every step, reference test/test_2d.py:28-38).
// Then nothing has to be moved to where it is needed: no plan, no p-ordered records, no fetch by sample id.
// Replaces the gather / scatter loops of the reference, 2d.cu:340-354, :464-505, :661-712, :850-888.
//
// One kernel template, four stages (forward, first / second / fused third backward).  A WAVE owns `chunk`
// consecutive samples of one table and walks them 64 at a time, lane = sample:
//   * streams (coordinates, cotangents) are loaded coalesced a batch ahead and leave coalesced from registers;
//   * the table is read through a WINDOW in LDS: the 4 x 10 node rows around the quad row of the tile the wave is in,
//     loaded once (coalesced); a lane takes its four node rows from there (16-byte LDS reads) and does its own
//     products with them -- per-sample work never crosses lanes;
//   * only the scatter-reduce into grad_input crosses lanes.  Samples are grouped by the QUAD (2 x 2 cells) of the
//     un-shifted point: with the multicell shift n/N a point set ordered for table 0 moves by less than one cell in
//     table n, so every sample of a quad touches nodes of the same 4 x 4 block.  A sample's four coefficients sit in a
//     16-float block over those nodes and the sum over a run of equal quad of  block^T x cotangents --
//     (16 x m) x (m x C) -- is taken by the matrix core in exact fp32 (v_mfma_f32_16x16x4_f32, four samples per
//     instruction, operands one dword per lane from LDS; run bounds are wave-uniform, from a ballot).  The result is
//     added once per run to the wave's private accumulator window (plain read-modify-write, one writer);
//   * when the samples leave the window (next quad row, next tile, any jump of an unordered set) the wave adds the
//     accumulator window to the channels-last accumulator with whole-row float atomics, clears it and reloads the
//     table window.  Correct for ANY order of the points; fast when the order is coherent: ten bytes of atomics per
//     sample instead of four rows.
// Why the matrix core in a gather/scatter op: the reduction needs every lane to see every sample's coefficients; with
// vector lanes that is two LDS reads per sample and lane, with the MFMA operand layout two per FOUR samples.  Why not for
// the per-sample products as well (the first version of this file did): they are lane-local, 4 of 16 block entries are
// non-zero, and the fp32 matrix rate equals the vector rate -- the kernels were instruction-issue bound (rocprofv3:
// 700-1050 vector + 330-640 scalar instructions per 64 samples; profiles/round3_coherent_issue_bound.txt).
#pragma once
#include "cs_tiled.cuh"

namespace cs {
namespace coh {

namespace tl = cs::tiled;

constexpr int TS = 8;            // cells per side of the tiles the window is anchored on (= cs_sort.hip)
constexpr int TQ = TS / 2;       // quads per tile side
constexpr int WN = TS + 2;       // nodes per window row: a tile, +1 for the multicell shift, +1 for the far nodes
constexpr int WNY = 4;           // window rows: the nodes of one row of quads
constexpr int KB = 16;           // floats per coefficient block: the 4 x 4 nodes of a quad
constexpr int KP = 17;           // pitch of the blocks in LDS: the lanes' stores of their own block fall on different banks
constexpr int MAX_SIZE = 32766;  // quad coordinates are packed into 15 bits
constexpr uint32_t KEY_NONE = 0xFFFFFFFFu;
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { FWD = 0, BWD = 1, BB = 2, BBB = 3 };

// batches of stream loads a wave keeps in flight (register sets); the summing kernels' chunk is 64 * depth(mode)
#ifndef CS_COH_DEPTH_BWD
#define CS_COH_DEPTH_BWD 2
#endif
#ifndef CS_COH_DEPTH_BB
#define CS_COH_DEPTH_BB 2
#endif
__host__ __device__ constexpr int depth(int mode) {
    return mode == FWD ? 4 : mode == BWD ? CS_COH_DEPTH_BWD : mode == BB ? CS_COH_DEPTH_BB : 2;
}

template <int C>
struct Lay {
    static constexpr int CQ = C / 4;
    static constexpr int NH = C > 16 ? C / 16 : 1;     // 16-channel halves (one accumulator tile each)
    static constexpr int PT = 66;                      // pitch of the channel-major cotangent rows [C][PT]: the matrix
                                                       // core's B reads (16 channels x 2 samples per half wave) hit 32 banks
    static constexpr int PTH = 34;                     // ... when the rows hold half a batch (half_batch()): same banks
#ifndef CS_COH_ROW_PAD
#define CS_COH_ROW_PAD 4
#endif
    // floats per table-window row: 10 nodes + one 16-byte slot.  A lane's 16-byte reads of its node rows (ds_read_b128: 16
    // lanes share the 64 banks) then fall on different slots for all 16 nodes of a quad's 4 x 4 block -- slot (4 lx + 9 ly) mod 16
    // at C = 16 -- where the unpadded pitch put node (y+1, x) on the banks of node (y, x+2): samples that the multicell shift
    // moved to the next cell collided with their neighbours' (rocprofv3: 56 of the 154 LDS cycles of a batch's 16 reads)
    static constexpr int ROWP = WN * C + CS_COH_ROW_PAD;
    static constexpr int WIN = WNY * ROWP;
    static constexpr int ROWA = WN * C + 16;           // row pitch of the accumulator window: block rows r, r+1 of the
    static constexpr int WINA = WNY * ROWA;            // matrix core's result (half a wave) land 16 banks apart
};

// The first backward stages the scatter-reduce's operands HALF a batch at a time (32 samples: lanes 0..31, then lanes
// 32..63 through the same LDS rows).  It is the stage with the fewest bytes per sample to hide behind and the one whose
// registers (<= 128) admit four waves per SIMD; the whole-batch operands (8.4 KiB of a wave's 13.7) kept it at 2.75, and
// capping the waves per CU showed these kernels follow their occupancy (8 instead of 11 waves per CU: +15 %).  Halved,
// a wave takes 9.6 KiB -> 16 per CU.  The second / third backward hold 140-168 registers (3 waves per SIMD at best) and
// keep whole batches: they would pay the second round of stores for one more wave per CU.
#ifdef CS_COH_NO_HALF          // A/B builds (tools/ab.sh)
__host__ __device__ constexpr bool half_batch(int) { return false; }
#else
#ifdef CS_COH_HALF_BB
__host__ __device__ constexpr bool half_batch(int mode) { return mode == BWD || mode == BB; }
#else
__host__ __device__ constexpr bool half_batch(int mode) { return mode == BWD; }
#endif
#endif
// per wave (floats): cotangent rows GT, coefficient blocks KA, the table window TW, the accumulator window AW -- in
// this order: the reads of a run's last, partly masked group of four may run past GT / KA into what follows
template <int C>
__host__ __device__ constexpr int wave_floats(int mode) {
    return mode == FWD ? Lay<C>::WIN
                       : C * (half_batch(mode) ? Lay<C>::PTH : Lay<C>::PT) + (half_batch(mode) ? 32 : 64) * KP + Lay<C>::WIN + Lay<C>::WINA;
}

// LDS traffic of one wave is in program order; this only keeps the compiler from moving accesses of OTHER lanes' data
// across a phase boundary
__device__ __forceinline__ void wave_sync() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// geometry of one sample + the quad it is grouped by
struct Geo {
    Axis ax[2];
    uint32_t akey;   // (qy << 15) | qx, q = cell of the UN-SHIFTED point / 2; KEY_NONE: touches no node
    int lx, ly;      // the sample's low node inside its quad's 4 x 4 block: 0..2 (its other nodes: +1 in x, +1 in y)
    __device__ __forceinline__ float first(int a, int j) const {
        float sgn = ((a >> j) & 1) ? ax[j].d1 : -ax[j].d1;
        return sgn * ax[1 - j].w[(a >> (1 - j)) & 1];
    }
    __device__ __forceinline__ float pure2(int a, int j) const {
        float sgn = ((a >> j) & 1) ? -ax[j].d2 : ax[j].d2;
        return sgn * ax[1 - j].w[(a >> (1 - j)) & 1];
    }
    __device__ __forceinline__ float mixed2(int a) const {
        float sx_ = (a & 1) ? ax[0].d1 : -ax[0].d1, sy_ = (a & 2) ? ax[1].d1 : -ax[1].d1;
        return sx_ * sy_;
    }
    __device__ __forceinline__ int slot0() const { return 4 * ly + lx; }
};

template <int KERNEL, int ORDER>
__device__ __forceinline__ void make_geo(Geo &g, float2 xy, float off, const Dims &d, const Flags &f, bool live) {
    const int align = ORDER == 0 ? 1 : f.align;             // 2D forward: align_corners = 1 whatever the caller says (2d.cu:307-308)
    g.ax[0] = make_axis<KERNEL, ORDER>(xy.x, d.size[0], f, align, off);
    g.ax[1] = make_axis<KERNEL, ORDER>(xy.y, d.size[1], f, align, off);
    const int ux = g.ax[0].lo + 1, uy = g.ax[1].lo + 1;      // u = low node + 1: 0 .. size when a node is touched
    const bool valid = live && ux >= 0 && ux <= d.size[0] && uy >= 0 && uy <= d.size[1];
    // 1 - t = position inside the cell: the multicell shift `off` carried the sample over the cell boundary iff it is
    // below off.  Only the GROUPING depends on this; every sample is added at its own low node.
    const int sx = (g.ax[0].t > 1.0f - off && ux > 0) ? 1 : 0;
    const int sy = (g.ax[1].t > 1.0f - off && uy > 0) ? 1 : 0;
    const int cx = ux - sx, cy = uy - sy;                    // cell of the un-shifted point (what the caller's order groups)
    g.lx = valid ? (cx & 1) + sx : 0;
    g.ly = valid ? (cy & 1) + sy : 0;
    g.akey = valid ? ((uint32_t)(cy >> 1) << 15) | (uint32_t)(cx >> 1) : KEY_NONE;
}

// ---- the two windows: WNY x WN nodes of one table, anchored together ------------------------------------------------
// TW: the table's values (read), AW: the sums for grad_input (read-modify-write).  Anchor = (first quad of a tile in x,
// quad row): window node (iy, ix) = table node (2 aqy - 1 + iy, 2 tqx - 1 + ix); the 4 x 4 block of quad (qx, aqy)
// starts at window column 2 (qx - tqx).
template <int C, bool ACC>
struct Windows {
    using L = Lay<C>;
    float *tw, *aw;
    int tqx, aqy;
    bool dirty, noflush, noload;
    __device__ __forceinline__ void init(float *t, float *a) {
        tw = t;
        aw = a;
        tqx = aqy = -(1 << 20);
        dirty = false;
        noflush = false;
        noload = false;
        if (ACC)
            for (int i = threadIdx.x & 63; i < L::WINA; i += 64) aw[i] = 0.0f;
    }
    __device__ __forceinline__ bool holds(uint32_t key) const {
        return (int)(key & 0x7FFCu) == tqx && (int)(key >> 15) == aqy;
    }
    __device__ __forceinline__ int column(uint32_t key) const { return 2 * ((int)(key & 0x7FFFu) - tqx); }
    // nodes outside the table read as zero (zero padding).  In two halves: the loads go out when the wave knows where it
    // moves, the rows land in LDS after the old accumulator window has been flushed -- the flush (LDS reads, atomics)
    // runs inside the loads' round trip instead of in front of it.
    static constexpr int Q = WN * L::CQ;                    // float4 per window row
    static constexpr int NV = (WNY * Q + 63) / 64;
    __device__ __forceinline__ void load_issue(const float *__restrict__ tab_n, const Dims &d, float4 (&v)[NV]) const {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = i * 64 + lane, iy = idx / Q, c4 = idx - iy * Q;
            const int gy = 2 * aqy - 1 + iy, gx = 2 * tqx - 1 + c4 / L::CQ;
            const bool ok = idx < WNY * Q && gy >= 0 && gy < d.size[1] && gx >= 0 && gx < d.size[0];
            const float4 *src = reinterpret_cast<const float4 *>(tab_n + ((int64_t)(ok ? gy : 0) * d.size[0] + (ok ? gx : 0)) * C) + c4 % L::CQ;
            v[i] = *src;
            if (!ok) v[i] = tl::zero4();
        }
    }
    __device__ __forceinline__ void load_land(const float4 (&v)[NV]) {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = i * 64 + lane, iy = idx / Q, c4 = idx - iy * Q;
            if (idx < WNY * Q) *reinterpret_cast<float4 *>(tw + iy * L::ROWP + 4 * c4) = v[i];
        }
    }
    // AW -> the channels-last accumulator, WN*C contiguous floats per row: whole-line float atomics; AW is left zero
    __device__ __forceinline__ void flush(float *__restrict__ acc_n, const Dims &d) { flush_at(acc_n, d, tqx, aqy); }
    __device__ __forceinline__ void flush_at(float *__restrict__ acc_n, const Dims &d, int tqx, int aqy) {
        if (!ACC || !dirty) return;
        dirty = false;
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int iy = 0; iy < WNY; ++iy) {
            const int gy = 2 * aqy - 1 + iy;
            float *wrow = aw + iy * L::ROWA;
            const bool yok = gy >= 0 && gy < d.size[1] && !noflush;
            float *arow = acc_n + ((int64_t)gy * d.size[0] + 2 * tqx - 1) * C;
#pragma unroll
            for (int i0 = 0; i0 < WN * C; i0 += 64) {
                const int idx = i0 + lane;
                if (idx < WN * C) {
                    const int gx = 2 * tqx - 1 + idx / C;
                    const float v = wrow[idx];
                    wrow[idx] = 0.0f;
                    if (yok && gx >= 0 && gx < d.size[0] && v != 0.0f) unsafeAtomicAdd(arow + idx, v);
                }
            }
        }
    }
    // move both windows to the quad row of `key`
    __device__ __forceinline__ void anchor(uint32_t key, const float *__restrict__ tab_n, float *__restrict__ acc_n,
                                           const Dims &d) {
        const int otqx = tqx, oaqy = aqy;
        tqx = (int)(key & 0x7FFCu);
        aqy = (int)(key >> 15);
        float4 v[NV];
#ifdef CS_COH_NO_ANCHOR_OVERLAP      // A/B builds: the flush in front of the loads, as before
        flush_at(acc_n, d, otqx, oaqy);
        if (!noload) load_issue(tab_n, d, v);
#else
        if (!noload) load_issue(tab_n, d, v);
        flush_at(acc_n, d, otqx, oaqy);
#endif
        wave_sync();
        if (!noload) load_land(v);
        wave_sync();
    }
};

// ---- the scatter-reduce of one run ----------------------------------------------------------------------------------
// S[slot][c] += sum_j K[j][slot] * G[j][c] over the run's samples j in [js, js + len), by v_mfma_f32_16x16x4_f32: rows =
// the 16 slots of the quad's block, k = samples (4 per instruction), columns = channels.  A one dword per lane, lane
// (k, i) = A[i][k] = KA[sample][slot i]; B lane (k, j) = GT[channel j][sample]; D lane (r, j), register v =
// S[4 r + v][j] = block row r, column v -- a k-ordered fma chain in exact fp32.  Groups of four samples are counted from
// the run's first sample; the lanes of the last group that belong to the next run are masked on both operands (their
// reads may run past the arrays into the wave's own LDS: any value will do).  Then into AW at block column bx.
template <int C, int PT>
__device__ __forceinline__ void scatter_run(const float *KA, const float *GT, float *aw, int js, int len, int bx) {
    using L = Lay<C>;
    constexpr int NH = L::NH;
    const int lane = threadIdx.x & 63, k = lane >> 4, m = lane & 15, cc = m < C ? m : C - 1;
    f32x4 D[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) D[h] = f32x4{0.f, 0.f, 0.f, 0.f};
    float *wp = aw + k * L::ROWA + bx * C + cc;
    float old[NH][4];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int v = 0; v < 4; ++v) old[h][v] = wp[16 * h + v * C];
    const float *ka = KA + (js + k) * KP + m;
    const float *gp = GT + cc * PT + js + k;
#pragma unroll
    for (int u0 = 0; u0 < 16; u0 += 4) {
        if (4 * u0 >= len) break;                            // wave-uniform
        float a[4], b[4][NH];
#pragma unroll
        for (int u = 0; u < 4; ++u) {                        // every operand of four instructions before the first
            a[u] = ka[(u0 + u) * 4 * KP];
#pragma unroll
            for (int h = 0; h < NH; ++h) b[u][h] = gp[16 * h * PT + (u0 + u) * 4];
        }
        if (4 * (u0 + 4) <= len) {                           // sixteen samples of the run: nothing to mask
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int h = 0; h < NH; ++h) D[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u][h], D[h], 0, 0, 0);
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool in = 4 * (u0 + u) + k < len;
                const float av = in ? a[u] : 0.0f;
#pragma unroll
                for (int h = 0; h < NH; ++h)
                    D[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, in ? b[u][h] : 0.0f, D[h], 0, 0, 0);
            }
        }
    }
    if (m < C) {
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int v = 0; v < 4; ++v) wp[16 * h + v * C] = old[h][v] + D[h][v];
    }
}

// ---- streams ------------------------------------------------------------------------------------------
// Every global address of the kernel is a wave-uniform base (scalar registers) plus a 32-bit byte offset of the lane
// inside the wave's chunk: one address register per lane instead of a 64-bit pointer per channel row.
template <typename T>
__device__ __forceinline__ const T *at(const T *ubase, uint32_t byteoff) {
    return reinterpret_cast<const T *>(reinterpret_cast<const char *>(ubase) + byteoff);
}
template <typename T>
__device__ __forceinline__ T *at(T *ubase, uint32_t byteoff) {
    return reinterpret_cast<T *>(reinterpret_cast<char *>(ubase) + byteoff);
}
// The channel-major streams go through BUFFER instructions: a 128-bit descriptor (scalar registers) of the wave's first
// sample in channel row 16 g; row c of the group = a scalar byte offset (c mod 4) * P * sizeof(T) -- four scalar registers
// worked out once, shared by every stream of the kernel -- plus the lane's offset inside the chunk moved by (c / 4) * 4 rows
// (four vector registers per batch, three adds, shared by the streams as well: all have the row pitch P).  No other address
// arithmetic is left in the batch loop: the pointer form spent six scalar instructions per row and batch (96 of the ~180
// scalar instructions of a batch at 16 channels) on 64-bit row bases it could not keep.  (Sixteen scalar row offsets
// instead of 4 + 4 made the second and third backward spill scalar registers to vector lanes: 100 / 240 v_readlane.)
// Reach of one descriptor: 15 rows + a chunk, in 32 bits (Launch: supported()).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr int RG = 16;                                  // channel rows per descriptor
constexpr int RQ = 4;                                   // rows per scalar offset set
struct RowOffs {
    uint32_t s[RQ];        // (c mod 4) rows, bytes: wave-uniform, loop-invariant
    uint32_t quad;         // 4 rows, bytes
    int Cv;                // the caller's channel count (<= the padded count C the kernel is built for)
    template <typename T>
    __device__ __forceinline__ void init(int64_t P, int Cv_) {
        Cv = Cv_;
        const uint32_t row = (uint32_t)P * (uint32_t)sizeof(T);
#pragma unroll
        for (int i = 0; i < RQ; ++i) s[i] = (uint32_t)i * row;
        quad = RQ * row;
    }
    // The caller's channel count for the per-row tests of the padded case, opaque to the optimiser: left visible, the
    // sixteen wave-uniform comparisons c < Cv are hoisted out of the batch loop as sixteen 64-bit masks -- thirty-two scalar
    // registers, which the second and third backward then spill to vector lanes (100 / 240 v_readlane per batch pair).
    __device__ __forceinline__ int channels() const {
        int c = Cv;
        asm volatile("" : "+s"(c));
        return c;
    }
};
// the lane's byte offset inside the chunk for each of the four row quads of a group
struct LaneOffs {
    uint32_t v[RG / RQ];
    __device__ __forceinline__ void set(uint32_t byteoff, const RowOffs &ro) {
#pragma unroll
        for (int q = 0; q < RG / RQ; ++q) v[q] = byteoff + (uint32_t)q * ro.quad;
    }
};
__device__ __forceinline__ rsrc_t make_rsrc(const void *p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, -1, 0x00020000);
}
template <int C, typename T>
struct Rows {
    static constexpr int NG = (C + RG - 1) / RG;
    rsrc_t r[NG];
    __device__ __forceinline__ void init(const T *row0, int64_t P) {
#pragma unroll
        for (int g = 0; g < NG; ++g) r[g] = make_rsrc(row0 ? row0 + (int64_t)g * RG * P : nullptr);
    }
};
template <typename T>
__device__ __forceinline__ T load_elem(rsrc_t r, uint32_t voff, uint32_t soff) {
    if constexpr (sizeof(T) == 4) {
        return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 2));        // 2: nontemporal
    } else {
        return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b16(r, (int)voff, (int)soff, 2));
    }
}
// outputs leave nontemporal: the table is read through the windows, nothing here wants to stay in the L2 (forward:
// 0.300 -> 0.242 ms against write-through stores, the other stages unchanged)
#ifndef CS_COH_OUT_AUX
#define CS_COH_OUT_AUX 2      // cache policy of the output stores: 2 nontemporal, 16 sc1 (write-through), 0 plain (A/B builds)
#endif
template <typename T>
__device__ __forceinline__ void store_elem(rsrc_t r, uint32_t voff, uint32_t soff, float v) {
    const T t = (T)v;
    if constexpr (sizeof(T) == 4) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, t), r, (int)voff, (int)soff, CS_COH_OUT_AUX);
    } else {
        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, t), r, (int)voff, (int)soff, CS_COH_OUT_AUX);
    }
}
// channel c of this lane's sample; channels >= Cv do not exist (C padded up to a supported count)
template <int C, typename T>
struct StreamRegs {
    T raw[C];
    // every row is loaded, unconditionally (a load inside a wave-uniform branch is waited for at the join: serialized
    // round trips); rows the caller does not have -- C padded up to a supported count -- re-read an existing row and are
    // zeroed in arrived()
    __device__ __forceinline__ void issue(const Rows<C, T> &rows, const LaneOffs &lo, const RowOffs &ro) {
        if (ro.Cv == C) {
#pragma unroll
            for (int c = 0; c < C; ++c) raw[c] = load_elem<T>(rows.r[c / RG], lo.v[(c % RG) / RQ], ro.s[c % RQ]);
        } else {
            const int cv = ro.channels();
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const bool have = c < cv;
                raw[c] = load_elem<T>(rows.r[have ? c / RG : 0], have ? lo.v[(c % RG) / RQ] : lo.v[0], have ? ro.s[c % RQ] : 0u);
            }
        }
    }
    // channels past Cv read as zero (wave-uniform: nothing to do when no channel is padded)
    __device__ __forceinline__ void arrived(const RowOffs &ro) {
        if (ro.Cv < C) {
            const int cv = ro.channels();
#pragma unroll
            for (int c = 0; c < C; ++c)
                if (c >= cv) raw[c] = (T)0.0f;
        }
    }
    __device__ __forceinline__ float val(int c) const { return (float)raw[c]; }
};
// a sample's coefficients -> its block (the block is zero everywhere else: cleared by the same lane after the batch)
__device__ __forceinline__ void put_coefs(float *blk, int slot0, const float (&k)[4]) {
    blk[slot0] = k[0];
    blk[slot0 + 1] = k[1];
    blk[slot0 + 4] = k[2];
    blk[slot0 + 5] = k[3];
}
template <int C, int PT, typename R>
__device__ __forceinline__ void put_rows(float *GT, const R &regs, int col) {
#pragma unroll
    for (int c = 0; c < C; ++c) GT[c * PT + col] = regs.val(c);
}

struct Args {
    const float *icl, *grid, *offset;    // channels-last table, points, multicell offsets
    const void *gOut, *hO;               // cotangent streams (hO: fused third backward only, may be null)
    const float *cG, *hG;                // grid-shaped cotangents (null: zeros)
    float *acc;                          // zeroed channels-last accumulator of the input-shaped gradient
    float *out_grid;                     // grad_grid (first / second backward)
    void *out_stream;                    // output (forward) / grad_grad_out (second / third backward)
};

// =====================================================================================================
// MODE FWD  (2d.cu:265-356):  output = sum_a W_a input[q_a]
//      BWD  (2d.cu:406-506):  grad_input += W gOut;  grad_grid from the products input[q_a] . gOut
//      BB   (2d.cu:569-716), grad_out_input absent:  grad_input += D gOut;  grad_grad_out = sum_a D_a input[q_a];
//                             grad_grid from the products with the second-derivative coefficients
//      BBB  (2d.cu:774-890 + the extra second backward of modules_2d.py:106-111):  grad_input += E gOut (+ D hO with
//                             TWO);  grad_grad_out = sum_a E_a input[q_a]
// Launch: grid (ceil(ceil(P / chunk) / waves per block), N), 64 x waves threads; every wave is on its own.
//
// A wave keeps DEPTH batches of stream loads in flight (DEPTH register sets; with one, a batch took one loaded memory
// latency, ~4 us, whatever it computed: 11 waves per CU are all the LDS admits).  Global memory operations of a wave share
// ONE in-order counter, so a batch is arranged around one wait at its top:  [wait: the set's loads, issued DEPTH batches
// ago]  geometry  ->  LDS, vector and matrix work only  ->  the set's next loads go out, then this batch's outputs leave.
// The windows' global traffic (table rows in, atomics out) happens when the wave moves to another quad row: once per
// ~4 batches of an ordered set.
// `dbg`: experiments only, compiled in with -DCS_COH_DEBUG (cs_debug_coherent_tuning): 1 no scatter-reduce, 2 no window
// flush, 4 no per-sample products, 8 no stores of the output stream, 32 no table-window loads, 64 no scatter operands
// written to LDS.
// =====================================================================================================
// COMMON: zeros padding with align_corners -- every BASELINE config and the reference's defaults -- as compile-time
// constants: the padding variants are wave-uniform branches, but they cost scalar registers (the general kernels spill them
// to vector lanes), scalar instructions and ~40 branches per batch in kernels that are instruction-issue bound.
//
// NSUM (CS_SUM_OVER_N; SURVEY 8f-1, the PIXEL pattern features = sampler(cells, grid).sum(0), reference test/test_2d.py:38,
// :51): ONE set of points and ONE set of cotangents serve every table (grid [P,2], gOut / hO [C,P], cG / hG [P,2]) and
// every per-point result is wanted summed over the tables (output / grad_grad_out [C,P], grad_grid [P,2]); only the
// input-shaped gradient stays per table.  A wave then owns 128 points for ALL tables: their streams are loaded once and
// stay in registers, the wave walks the tables one after the other -- table n's window, scatter-reduce into table n's
// accumulator, products -- adding the per-point results up in registers, and writes them once: the (N,C,P) streams of
// the plain op, 1 GiB each at BASELINE configs[1], shrink N-fold and the caller's sums over n disappear.
template <int KERNEL, int CQ, int MODE, bool TWO, bool SCAT, typename ST, bool COMMON = false, bool NSUM = false>
#ifndef CS_COH_BWD_WAVES
#define CS_COH_BWD_WAVES 4       // waves per SIMD the first backward is compiled for (A/B builds)
#endif
__global__ __launch_bounds__(256, (NSUM || CQ > 4) ? 2 : (half_batch(MODE) && SCAT) ? CS_COH_BWD_WAVES : 3) void stage(Args a, Dims d, Flags f_, int chunk, int dbg) {
    Flags f = f_;
#ifndef CS_COH_DEBUG
    dbg = 0;      // the ablation switches (wrong results by design) exist only in experiment builds (-DCS_COH_DEBUG, tools/ab.sh)
#endif
    if constexpr (COMMON) {
        f.pad = PAD_ZEROS;
        f.align = 1;
    }
    constexpr int C = 4 * CQ;
    using L = Lay<C>;
    constexpr bool IN = MODE != FWD;                     // reads the cotangent stream gOut
    constexpr bool ACC = IN && SCAT;                     // scatters into grad_input (SCAT false: grad_input not wanted)
    constexpr bool PROD = MODE == BWD || MODE == BB;     // needs the products input[q_a] . gOut
    constexpr bool OUTS = MODE != BWD;                   // produces a channel stream
    constexpr int ORD = MODE == FWD ? 0 : MODE == BWD ? 1 : 2;
    constexpr int DEPTH = depth(MODE);                   // batches of stream loads in flight per wave
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    int n = NSUM ? 0 : blockIdx.y;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // wave in block: a scalar
    const int64_t wv = (int64_t)blockIdx.x * (blockDim.x >> 6) + wib;
    const int64_t p_begin = wv * chunk;
    if (p_begin >= d.P) return;
    const int count = (int)min((int64_t)chunk, d.P - p_begin);       // samples of this wave
    constexpr bool HALF = ACC && half_batch(MODE);       // the scatter-reduce's operands go through LDS 32 samples at a time
    constexpr int PTV = HALF ? L::PTH : L::PT;           // pitch of the cotangent rows
    constexpr int NB = HALF ? 32 : 64;                   // coefficient blocks
    float *GT = lds + wib * wave_floats<C>(ACC ? MODE : FWD);
    float *KA = GT + C * PTV;
    float *TW = ACC ? KA + NB * KP : GT;
    float *AW = TW + L::WIN;
    float off = a.offset[n];
    const float *tab_n = a.icl + (int64_t)n * d.vol * C;
    float *acc_n = ACC ? a.acc + (int64_t)n * d.vol * C : nullptr;
    // wave-uniform bases: the first sample of the chunk (NSUM: n = 0 here, the shared inputs and the summed outputs have no n)
    const float *grid_w = a.grid + (d.gpt(n, p_begin)) * 2;
    const float *cg_w = (MODE >= BB && a.cG) ? a.cG + (d.gpt(n, p_begin)) * 2 : nullptr;
    const float *hg_w = (MODE == BBB && a.hG) ? a.hG + (d.gpt(n, p_begin)) * 2 : nullptr;
    Rows<IN ? C : 1, ST> go_r;
    Rows<TWO ? C : 1, ST> ho_r;
    Rows<OUTS ? C : 1, ST> os_r;
    go_r.init(IN ? (const ST *)a.gOut + (int64_t)n * d.go_ns + p_begin : nullptr, d.P);
    ho_r.init(TWO ? (const ST *)a.hO + (int64_t)n * d.ho_ns + p_begin : nullptr, d.P);
    os_r.init(OUTS ? (const ST *)a.out_stream + (int64_t)n * d.out_ns + p_begin : nullptr, d.P);
    RowOffs ro;
    ro.init<ST>(d.P, d.C);
    float *og_w = (MODE == BWD || MODE == BB) ? a.out_grid + ((int64_t)n * d.P + p_begin) * 2 : nullptr;

    Windows<C, ACC> w;
    w.init(TW, AW);
    w.noflush = (dbg & 2) != 0;
    w.noload = (dbg & 32) != 0;
    if (ACC) {
        for (int i = lane; i < NB * KP; i += 64) KA[i] = 0.0f;
    }
    // DEPTH register sets of stream loads, each re-issued for the batch DEPTH ahead as soon as its batch is done with it
    struct Pre {
        float2 xy, cg, hg;
        StreamRegs<IN ? C : 1, ST> sg;
        StreamRegs<TWO ? C : 1, ST> sh;
    };
    auto issue = [&](Pre &s, int b) __attribute__((always_inline)) {
        const uint32_t r = (uint32_t)min(b + lane, count - 1);
        s.xy = *at(reinterpret_cast<const float2 *>(grid_w), r * 8u);
        s.cg = s.hg = make_float2(0.f, 0.f);
        if (cg_w) s.cg = *at(reinterpret_cast<const float2 *>(cg_w), r * 8u);
        if (hg_w) s.hg = *at(reinterpret_cast<const float2 *>(hg_w), r * 8u);
        if constexpr (IN) {
            LaneOffs lo;
            lo.set(r * (uint32_t)sizeof(ST), ro);
            s.sg.issue(go_r, lo, ro);
            if constexpr (TWO) s.sh.issue(ho_r, lo, ro);
        }
    };
    // per-point results of one batch; NSUM: summed over the tables before they leave
    struct Res {
        float O[OUTS ? C : 1];
        float gx, gy;
    };
    auto clear = [&](Res &r) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < (OUTS ? C : 1); ++c) r.O[c] = 0.0f;
        r.gx = r.gy = 0.0f;
    };
    auto emit = [&](const Res &r, int b0) __attribute__((always_inline)) {
        const int rel = b0 + lane;
        const bool live = rel < count;
        if ((MODE == BWD || MODE == BB) && live)
            *at(reinterpret_cast<float2 *>(og_w), (uint32_t)rel * 8u) = make_float2(r.gx, r.gy);
        if (OUTS && live && !(dbg & 8)) {
            LaneOffs lo;
            lo.set((uint32_t)rel * (uint32_t)sizeof(ST), ro);
            if (ro.Cv == C) {
#pragma unroll
                for (int c = 0; c < C; ++c) store_elem<ST>(os_r.r[c / RG], lo.v[(c % RG) / RQ], ro.s[c % RQ], r.O[c]);
            } else {
                const int cv = ro.channels();
#pragma unroll
                for (int c = 0; c < C; ++c)
                    if (c < cv) store_elem<ST>(os_r.r[c / RG], lo.v[(c % RG) / RQ], ro.s[c % RQ], r.O[c]);
            }
        }
    };
    auto batch = [&](Pre &s, int b0, Res &res) __attribute__((always_inline)) {
        const int rel = b0 + lane;                            // this lane's sample inside the chunk
        const bool live = rel < count;
        Geo g;
        make_geo<KERNEL, ORD>(g, s.xy, off, d, f, live);    // the wait of the batch: loads issued DEPTH batches ago
        if constexpr (IN) s.sg.arrived(ro);
        if constexpr (TWO) s.sh.arrived(ro);
        const float2 cgb = s.cg, hgb = s.hg;

        // coefficients: kS scatters gOut and weights the table rows of the output stream, kH scatters hO (TWO)
        float kS[4], kH[4], Sx[4], Sy[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (MODE == FWD || MODE == BWD) {
                kS[q] = g.ax[0].w[q & 1] * g.ax[1].w[q >> 1];
            } else {
                const float Dm = g.first(q, 0) * cgb.x + g.first(q, 1) * cgb.y;
                if (MODE == BB) {
                    kS[q] = Dm;
                    Sx[q] = g.pure2(q, 0) * cgb.x;                // pure second derivatives only (2d.cu:705-706)
                    Sy[q] = g.pure2(q, 1) * cgb.y;
                    if (f.exact) {
                        const float mx = g.mixed2(q);
                        Sx[q] = fmaf(mx, cgb.y, Sx[q]);
                        Sy[q] = fmaf(mx, cgb.x, Sy[q]);
                    }
                } else {
                    kH[q] = Dm;
                    kS[q] = g.pure2(q, 0) * (hgb.x * cgb.x) + g.pure2(q, 1) * (hgb.y * cgb.y);   // 2d.cu:876
                    if (f.exact) kS[q] = fmaf(g.mixed2(q), hgb.x * cgb.y + hgb.y * cgb.x, kS[q]);
                }
            }
        }
        const bool valid = g.akey != KEY_NONE;
        float *blk = KA + (HALF ? (lane & 31) : lane) * KP;
        const int s0 = g.slot0();
        const float zero4[4] = {0.f, 0.f, 0.f, 0.f};
        int rhalf = -1;                                     // HALF: whose operands sit in LDS (lanes 32 rhalf .. 32 rhalf + 31)
        if constexpr (ACC && !TWO && !HALF) {               // the scatter operands of the whole batch, once
            if (!(dbg & 64)) {
                put_coefs(blk, s0, kS);
                put_rows<C, PTV>(GT, s.sg, lane);
            }
        }
        float Y[4] = {0.f, 0.f, 0.f, 0.f};
        float(&O)[OUTS ? C : 1] = res.O;
        if constexpr (!NSUM) clear(res);

        // runs of equal quad: heads = first lane of every maximal stretch
        const uint32_t prev = (uint32_t)__shfl_up((int)g.akey, 1);
        const uint64_t heads = __ballot(lane == 0 || g.akey != prev);
        uint64_t todo = __ballot(valid);
        // segments: everything that fits the windows where they are (or where the first waiting sample puts them)
        while (todo) {
            const int lead = __ffsll((unsigned long long)todo) - 1;
            const uint32_t key0 = (uint32_t)__builtin_amdgcn_readlane((int)g.akey, lead);
            if (!w.holds(key0)) w.anchor(key0, tab_n, acc_n, d);
            const bool mine = valid && w.holds(g.akey);
            const uint64_t seg = __ballot(mine) & todo;
            todo &= ~seg;
            const bool inseg = (seg >> lane) & 1;
            if (ACC && !(dbg & 1)) {
#pragma unroll
                for (int pass = TWO ? 0 : 1; pass < 2; ++pass) {
                    if constexpr (!HALF) {
                        if constexpr (TWO) {                    // (hO, D) then (gOut, E) through the same LDS rows
                            wave_sync();
                            put_coefs(blk, s0, pass ? kS : kH);
                            if (pass) put_rows<C, PTV>(GT, s.sg, lane);
                            else put_rows<C, PTV>(GT, s.sh, lane);
                        }
                        wave_sync();
                        uint64_t hs = heads & seg;
                        while (hs) {
                            const int js = __ffsll((unsigned long long)hs) - 1;
                            hs &= hs - 1;
                            const uint64_t later = heads & ~((2ull << js) - 1);
                            const int je = later ? __ffsll((unsigned long long)later) - 1 : 64;
                            const uint32_t key = (uint32_t)__builtin_amdgcn_readlane((int)g.akey, js);
                            scatter_run<C, PTV>(KA, GT, w.aw, js, je - js, w.column(key));
                        }
                    } else {
                        // half a batch at a time through the same rows and blocks; a run that crosses the middle of the
                        // batch is reduced in two parts (lane 32 counts as a head)
                        const uint64_t hd = heads | (1ull << 32);
#pragma unroll
                        for (int half = 0; half < 2; ++half) {
                            const uint64_t hm = half ? 0xFFFFFFFF00000000ull : 0x00000000FFFFFFFFull;
                            uint64_t hs = hd & seg & hm;
                            if (!hs) continue;                      // wave-uniform
                            wave_sync();
                            if (rhalf >= 0 && rhalf != half) {          // the other half's blocks go back to zero first: same storage
                                if ((lane >> 5) == rhalf) put_coefs(blk, s0, zero4);
                                wave_sync();
                            }
                            if ((lane >> 5) == half) {
                                if constexpr (TWO) {
                                    put_coefs(blk, s0, pass ? kS : kH);
                                    if (pass) put_rows<C, PTV>(GT, s.sg, lane & 31);
                                    else put_rows<C, PTV>(GT, s.sh, lane & 31);
                                } else {
                                    put_coefs(blk, s0, kS);
                                    put_rows<C, PTV>(GT, s.sg, lane & 31);
                                }
                            }
                            rhalf = half;
                            wave_sync();
                            while (hs) {
                                const int js = __ffsll((unsigned long long)hs) - 1;
                                hs &= hs - 1;
                                const uint64_t later = hd & ~((2ull << js) - 1);
                                const int je = later ? __ffsll((unsigned long long)later) - 1 : 64;
                                const uint32_t key = (uint32_t)__builtin_amdgcn_readlane((int)g.akey, js);
                                scatter_run<C, PTV>(KA, GT, w.aw, js - 32 * half, je - js, w.column(key));
                            }
                        }
                    }
                    w.dirty = true;
                }
            }
            if ((PROD || OUTS) && !(dbg & 4) && inseg) {     // lane-local: this sample's four node rows from the window
                const float *tp = w.tw + g.ly * L::ROWP + (w.column(g.akey) + g.lx) * C;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q == 2) __builtin_amdgcn_sched_barrier(0);     // two node rows' worth of registers at a time
                    const float4 *row = reinterpret_cast<const float4 *>(tp + (q >> 1) * L::ROWP + (q & 1) * C);
                    float y = 0.0f;
#pragma unroll
                    for (int c4 = 0; c4 < CQ; ++c4) {
                        const float4 t = row[c4];
                        if constexpr (PROD) {
                            y = fmaf(s.sg.val(4 * c4), t.x, y);
                            y = fmaf(s.sg.val(4 * c4 + 1), t.y, y);
                            y = fmaf(s.sg.val(4 * c4 + 2), t.z, y);
                            y = fmaf(s.sg.val(4 * c4 + 3), t.w, y);
                        }
                        if constexpr (OUTS) {
                            O[4 * c4] = fmaf(kS[q], t.x, O[4 * c4]);
                            O[4 * c4 + 1] = fmaf(kS[q], t.y, O[4 * c4 + 1]);
                            O[4 * c4 + 2] = fmaf(kS[q], t.z, O[4 * c4 + 2]);
                            O[4 * c4 + 3] = fmaf(kS[q], t.w, O[4 * c4 + 3]);
                        }
                    }
                    Y[q] = y;
                }
            }
        }
        if constexpr (ACC) {                                // the blocks go back to zero for the next batch
            wave_sync();
            if (!HALF || (lane >> 5) == rhalf) put_coefs(blk, s0, zero4);
        }
        __builtin_amdgcn_sched_barrier(0);
        // the set is free: its next loads go out, DEPTH batches of work to arrive in (NSUM: the sets are loaded once)
        if constexpr (!NSUM)
            if (b0 + 64 * DEPTH < count) issue(s, b0 + 64 * DEPTH);
        // this batch's per-point results
        if (MODE == BWD) {
            const float gx = g.ax[1].w[0] * (Y[1] - Y[0]) + g.ax[1].w[1] * (Y[3] - Y[2]);
            const float gy = g.ax[0].w[0] * (Y[2] - Y[0]) + g.ax[0].w[1] * (Y[3] - Y[1]);
            res.gx = fmaf(g.ax[0].d1, gx, res.gx);
            res.gy = fmaf(g.ax[1].d1, gy, res.gy);
        }
        if (MODE == BB) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                res.gx = fmaf(Sx[q], Y[q], res.gx);
                res.gy = fmaf(Sy[q], Y[q], res.gy);
            }
        }
        if constexpr (!NSUM) emit(res, b0);
    };
    Pre S[DEPTH];
#pragma unroll
    for (int i = 0; i < DEPTH; ++i)
        if (64 * i < count) issue(S[i], 64 * i);
    if constexpr (NSUM) {
        // chunk = 64 * DEPTH points: both sets stay in registers while the wave walks the tables
        Res R[DEPTH];
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) clear(R[i]);
        for (int nn = 0; nn < d.N; ++nn) {
            n = nn;
            off = a.offset[n];
            tab_n = a.icl + (int64_t)n * d.vol * C;
            if (ACC) acc_n = a.acc + (int64_t)n * d.vol * C;
#pragma unroll
            for (int i = 0; i < DEPTH; ++i)
                if (64 * i < count) batch(S[i], 64 * i, R[i]);
            w.flush(acc_n, d);                 // table n's sums leave before the windows move to table n + 1
            w.tqx = w.aqy = -(1 << 20);
        }
#pragma unroll
        for (int i = 0; i < DEPTH; ++i)
            if (64 * i < count) emit(R[i], 64 * i);
        return;
    }
    for (int b0 = 0; b0 < count; b0 += 64 * DEPTH) {
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) {
            Res res;
            if (b0 + 64 * i < count) batch(S[i], b0 + 64 * i, res);
        }
    }
    w.flush(acc_n, d);
}

}  // namespace coh
}  // namespace cs
