// cs_coherent.cuh -- the 2D backward stages for COHERENT point sets (CS_POINTS_COHERENT): consecutive samples of
// one table fall into the same or neighbouring cells, which is what a caller gets by ordering its collocation
// points once (cs2d_sort_points; PIXEL draws a fixed random set and re-uses it every step, reference
// test/test_2d.py:28-38).  Then nothing has to be moved to where it is needed: no plan, no p-ordered records,
// no fetch by sample id.  Replaces the scatter loops of the reference, 2d.cu:464-505, :661-712, :850-888.
//
//   * a WAVE owns a run of `chunk` consecutive samples of one table and walks it 64 samples at a time; the next
//     batch's coordinate and stream loads are issued a batch ahead, the previous batch's outputs leave a batch late
//     (why: the comment above the kernels);
//   * lane = sample does the geometry and takes the coalesced stream loads; the cotangents and the coefficients go
//     to the wave's LDS rows -- and stay there;
//   * the table is read through a window as well: the WNY x (WN+1) node rows around the tile the wave is walking are
//     loaded ONCE into LDS (coalesced), no per-sample gathers;
//   * the scatter-reduce.  Samples are grouped by the cell of the UN-SHIFTED point: with the multicell shift n/N a point
//     set ordered for table 0 alternates between up to four cells of table n, but always inside the 3 x 3 nodes around
//     the un-shifted cell.  Each sample's four coefficients are laid out as a 3 x 4 block over those nodes, and the sum
//     over a run of equal cell of  block x cotangents  -- (12 x m) x (m x C) -- is taken by the matrix core in exact fp32
//     (v_mfma_f32_16x16x4_f32, four samples per instruction, operands one dword per lane straight from the LDS rows; run
//     bounds are wave-uniform, from a ballot).  The result is added ONCE per run to the wave's private LDS window, a
//     WNY x WN-node image of grad_input: plain read-modify-write, the wave is the only writer;
//   * the per-sample products with the table (what the gathers were for) are block-local too:  Y = G T^T  and
//     O = K T  on the same 3 x 4 nodes, by the same instruction;
//   * when a run falls outside the windows (next tile of the caller's order, or any jump of an unordered set) the wave
//     adds the touched part of the accumulator window to the channels-last accumulator with whole-row float atomics,
//     moves both windows and reloads the table window.  Correct for ANY order of the points; fast when the order is
//     coherent: a few bytes of atomics per sample instead of 4 rows.
// Why the matrix core for a gather/scatter op: the reduction needs every lane to see every sample's coefficients; with
// vector lanes that is 6 LDS cycles per sample (broadcast reads), with the MFMA operand layout 1.
#pragma once
#include "cs_tiled.cuh"

namespace cs {
namespace coh {

namespace tl = cs::tiled;
using tl::dot4;
using tl::fma4;
using tl::zero4;

constexpr int TS = 8;            // cells per side of the tiles the window is anchored on
constexpr int WN = TS + 2;       // nodes per window row: a tile, +1 for the multicell shift, +1 for the far nodes
constexpr int WNY = 4;           // window rows: two rows of cells (a wave's 64 ordered samples wrap to the next cell row at most once)
constexpr int MAX_SIZE = 32766;  // cell coordinates are packed into 15 bits
constexpr uint32_t KEY_NONE = 0xFFFFFFFFu;
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int C>
struct Lay {
    static constexpr int CQ = C / 4;
    static constexpr int NH = C > 16 ? C / 16 : 1;     // 16-channel halves (one accumulator tile each)
    static constexpr int PT = 68;                      // pitch of the channel-major payload rows [C][PT]: the 16 channel
                                                       // rows of one sample quad fall on 16 banks x 2
    static constexpr int ROWP = (WN + 1) * C;          // floats per window row (one spare node: column 3 of a block at the
                                                       // window's right edge, always multiplied by zero but read)
    static constexpr int WIN = WNY * ROWP;
};

// per wave (floats): payloads GT (and HT) channel-major, coefficient blocks KA (and KB), per-sample products YB,
// the table window TW and the accumulator window AW
template <int C>
__host__ __device__ constexpr int wave_floats(bool two, bool yb) {
    return (two ? 2 : 1) * (C * Lay<C>::PT + 64 * 12) + (yb ? 64 * 12 : 0) + 2 * Lay<C>::WIN;
}

// LDS traffic of one wave is in program order; this only keeps the compiler from moving accesses of OTHER lanes' data
// across a phase boundary
__device__ __forceinline__ void wave_sync() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// geometry of one sample + the cell it is grouped by
struct Geo {
    Axis ax[2];
    float W[4];
    uint32_t akey;   // (uy - sy) << 15 | (ux - sx), u = lo + 1: the cell of the UN-SHIFTED point, which is what the caller's
                     // order groups; the sample's own low node is that cell + (sx, sy), s in {0,1}^2.  KEY_NONE: touches no node
    int sx, sy;
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
    // slot of this sample's low node in its group's 3 x 4 block (its other nodes: +1, +4, +5)
    __device__ __forceinline__ int slot0() const { return 4 * sy + sx; }
};

template <int KERNEL, int ORDER>
__device__ __forceinline__ void make_geo(Geo &g, float2 xy, float off, const Dims &d, const Flags &f, bool live) {
    g.ax[0] = make_axis<KERNEL, ORDER>(xy.x, d.size[0], f, f.align, off);
    g.ax[1] = make_axis<KERNEL, ORDER>(xy.y, d.size[1], f, f.align, off);
#pragma unroll
    for (int a = 0; a < 4; ++a) g.W[a] = g.ax[0].w[a & 1] * g.ax[1].w[a >> 1];
    const int ux = g.ax[0].lo + 1, uy = g.ax[1].lo + 1;
    const bool valid = live && ux >= 0 && ux <= d.size[0] && uy >= 0 && uy <= d.size[1];   // touches a node
    // 1 - t = position inside the cell: the multicell shift `off` carried the sample over the cell boundary iff it is
    // below off.  Only the GROUPING depends on this; every sample is added at its own low node, cell + (sx, sy).
    g.sx = (g.ax[0].t > 1.0f - off && ux > 0) ? 1 : 0;
    g.sy = (g.ax[1].t > 1.0f - off && uy > 0) ? 1 : 0;
    g.akey = valid ? ((uint32_t)(uy - g.sy) << 15) | (uint32_t)(ux - g.sx) : KEY_NONE;
}

// The coefficients k[0..3] of a sample's four nodes (x fastest) as a 3 x 4 block over the nodes of its GROUP's cell:
// row r, column x holds the coefficient of node cell + (x, r); column 3 and everything the sample does not touch is zero.
__device__ __forceinline__ void put_block(float *K12, const float (&k)[4], int sx, int sy, bool valid) {
    const float z = 0.0f;
    const float r0a = sy ? z : k[0], r0b = sy ? z : k[1];      // row 0: the sample's low row unless shifted
    const float r1a = sy ? k[0] : k[2], r1b = sy ? k[1] : k[3];
    const float r2a = sy ? k[2] : z, r2b = sy ? k[3] : z;
    float4 R0 = make_float4(sx ? z : r0a, sx ? r0a : r0b, sx ? r0b : z, z);
    float4 R1 = make_float4(sx ? z : r1a, sx ? r1a : r1b, sx ? r1b : z, z);
    float4 R2 = make_float4(sx ? z : r2a, sx ? r2a : r2b, sx ? r2b : z, z);
    if (!valid) R0 = R1 = R2 = zero4();
    *reinterpret_cast<float4 *>(K12) = R0;
    *reinterpret_cast<float4 *>(K12 + 4) = R1;
    *reinterpret_cast<float4 *>(K12 + 8) = R2;
}

// ---- the two windows: WNY x (WN+1) nodes of one table, anchored together ------------------------------------------
// TW: the table's values (read), AW: the sums for grad_input (read-modify-write).  Window node (iy, ix) = table node
// (ay - 1 + iy, ax - 1 + ix); ax a multiple of TS, ay the cell row of the first run (u coordinates, u = lo + 1).
template <int C>
struct Windows {
    using L = Lay<C>;
    float *tw, *aw;
    int ax, ay;
    int ylo, yhi;    // rows of AW touched since the last flush
    bool noflush;
    __device__ __forceinline__ void init(float *t, float *a) {
        tw = t;
        aw = a;
        ax = ay = -(1 << 20);
        ylo = WNY;
        yhi = -1;
        noflush = false;
        for (int i = threadIdx.x & 63; i < L::WIN; i += 64) aw[i] = 0.0f;
    }
    // table rows ay-1 .. ay+WNY-2, columns ax-1 .. ax+WN-1 -> TW; nodes outside the table read as zero (zero padding)
    __device__ __forceinline__ void load_table(const float *__restrict__ tab_n, const Dims &d) {
        const int lane = threadIdx.x & 63;
        constexpr int Q = (WN + 1) * L::CQ;                 // float4 per window row
        constexpr int NV = (WNY * Q + 63) / 64;
        float4 v[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = i * 64 + lane, iy = idx / Q, c4 = idx - iy * Q;
            const int gy = ay - 1 + iy, gx = ax - 1 + c4 / L::CQ;
            const bool ok = idx < WNY * Q && gy >= 0 && gy < d.size[1] && gx >= 0 && gx < d.size[0];
            const float4 *src = reinterpret_cast<const float4 *>(tab_n + ((int64_t)(ok ? gy : 0) * d.size[0] + (ok ? gx : 0)) * C) + c4 % L::CQ;
            v[i] = *src;
            if (!ok) v[i] = zero4();
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = i * 64 + lane;
            if (idx < WNY * Q) reinterpret_cast<float4 *>(tw)[idx] = v[i];
        }
    }
    // touched rows of AW -> the channels-last accumulator, WN*C contiguous floats per row: whole-line float atomics
    __device__ __forceinline__ void flush(float *__restrict__ acc_n, const Dims &d) {
        const int lane = threadIdx.x & 63;
        if (noflush) { ylo = WNY; yhi = -1; return; }
        for (int iy = ylo; iy <= yhi; ++iy) {
            const int gy = ay - 1 + iy;
            float *wrow = aw + iy * L::ROWP;
            const bool yok = gy >= 0 && gy < d.size[1];
#pragma unroll
            for (int i0 = 0; i0 < WN * C; i0 += 64) {
                const int idx = i0 + lane;
                if (idx < WN * C) {
                    const int gx = ax - 1 + idx / C;
                    const float v = wrow[idx];
                    wrow[idx] = 0.0f;
                    if (yok && gx >= 0 && gx < d.size[0] && v != 0.0f)
                        unsafeAtomicAdd(acc_n + ((int64_t)gy * d.size[0] + ax - 1) * C + idx, v);
                }
            }
        }
        ylo = WNY;
        yhi = -1;
    }
    // the block of the run with group key `key` inside the windows -> (ix, iy); re-anchors (flush AW, reload TW) when the
    // block does not fit: the first run of the wave, the next tile of the caller's order, any jump of an unordered set
    __device__ __forceinline__ void place(uint32_t key, int &ix, int &iy, const float *__restrict__ tab_n,
                                          float *__restrict__ acc_n, const Dims &d) {
        const int kx = (int)(key & 0x7FFFu), ky = (int)(key >> 15);
        ix = kx - ax;
        iy = ky - ay;
        if ((unsigned)ix >= (unsigned)TS || (unsigned)iy > (unsigned)(WNY - 3)) {
            flush(acc_n, d);
            ax = kx / TS * TS;
            ay = ky;
            ix = kx - ax;
            iy = 0;
            wave_sync();
            load_table(tab_n, d);
            wave_sync();
        }
        ylo = min(ylo, iy);
        yhi = max(yhi, iy + 2);
    }
};

// ---- one wave's worth of block-local products ---------------------------------------------------------------------
// Everything a sample needs from the table and gives to grad_input lives on the 3 x 4 nodes around its group's cell, so
// the three sums of a run [js, je) of one group are small dense products, taken by the matrix core in exact fp32
// (v_mfma_f32_16x16x4_f32: A one dword per lane, lane (k, i) = A[i][k]; B lane (k, j) = B[k][j]; D lane (., j),
// register v = D[4 (lane / 16) + v][j]; a k-ordered fma chain), every operand a single LDS dword at a lane-constant
// address plus a wave-uniform offset:
//   scatter  S[m][c] += sum_j K[j][m] * G[j][c]          rows = slots m, k = samples (4 per instruction), columns = channels
//   products Y[j][m]  = sum_c G[j][c] * T[m][c]          rows = samples (16 per instruction), k = channels, columns = slots
//   outputs  O[j][c]  = sum_m K[j][m] * T[m][c]          rows = samples, k = slots (3 x 4), columns = channels
// Rows / columns past the 12 slots or the C channels hold whatever the clamped addresses deliver and are never stored.
template <int C, bool TWO>
struct Blocks {
    using L = Lay<C>;
    static constexpr int NH = L::NH, CQ = L::CQ;
    const float *GT, *HT, *KA, *KB;
    int lane, k, m, mm, cc;
    __device__ __forceinline__ Blocks(const float *g, const float *h, const float *ka, const float *kb)
        : GT(g), HT(h), KA(ka), KB(kb) {
        lane = threadIdx.x & 63;
        k = lane >> 4;
        m = lane & 15;
        mm = m < 12 ? m : 11;
        cc = m < C ? m : C - 1;
    }
    // S += K^T G (+ KB^T H) over the run; then into AW at block origin (ix, iy).  Four groups of four samples per
    // round, every operand read before the first product (a run of an ordered set is ~1 round), the window's old values
    // fetched under the products.
    __device__ __forceinline__ void scatter(int js, int je, int ix, int iy, float *aw) const {
        f32x4 D[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) D[h] = f32x4{0.f, 0.f, 0.f, 0.f};
        float *wp = aw + (iy + (k < 3 ? k : 0)) * L::ROWP + ix * C + cc;
        float old[NH][3];
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int x = 0; x < 3; ++x) old[h][x] = wp[16 * h + x * C];
        const float *ka = KA + k * 12 + mm, *kb = KB + k * 12 + mm;
        const float *gp = GT + cc * L::PT + k, *hp = HT + cc * L::PT + k;
        for (int j0 = js; j0 < je; j0 += 16) {
            float a[4], a2[4], b[4][NH], b2[4][NH];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = min(j0 + 4 * u + k, 63) - k;         // (uniform part + clamp) - k: ka / gp carry k
                a[u] = ka[j * 12];
                if (TWO) a2[u] = kb[j * 12];
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    b[u][h] = gp[16 * h * L::PT + j];
                    if (TWO) b2[u][h] = hp[16 * h * L::PT + j];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (j0 + 4 * u < je) {                              // wave-uniform
                    const bool in = j0 + 4 * u + k < je;
                    const float av = in ? a[u] : 0.0f, av2 = TWO ? (in ? a2[u] : 0.0f) : 0.0f;
#pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        D[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[u][h], D[h], 0, 0, 0);
                        if (TWO) D[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(av2, b2[u][h], D[h], 0, 0, 0);
                    }
                }
            }
        }
        if (k < 3 && m < C) {      // lane (r = k, c = m): row r of the block, columns 0..2
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int x = 0; x < 3; ++x) wp[16 * h + x * C] = old[h][x] + D[h][x];
        }
    }
    // Y = G T^T for the run's samples -> YB[sample][12]
    __device__ __forceinline__ void products(int js, int je, int ix, int iy, const float *tw, float *YB, float *dump) const {
        float tb[CQ];                             // T[slot mm][channels CQ k .. CQ k + CQ - 1]
        const float *tp = tw + (iy + (mm >> 2)) * L::ROWP + (ix + (mm & 3)) * C + (C / 4) * k;
#pragma unroll
        for (int s = 0; s < C / 4; ++s) tb[s] = tp[s];
        for (int i0 = js; i0 < je; i0 += 16) {
            const int row = min(i0 + m, 63);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < C / 4; ++s)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(GT[((C / 4) * k + s) * L::PT + row], tb[s], acc, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < 4; ++v) {       // rows of other runs and slots past 12 go to a dump word: no lane masks
                const int smp = i0 + 4 * k + v;
                float *dst = (smp < je && m < 12) ? YB + smp * 12 + m : dump;
                *dst = acc[v];
            }
        }
    }
    // O = K T for the run's samples -> over their payload columns in GT (the run is done with them)
    __device__ __forceinline__ void outputs(int js, int je, int ix, int iy, const float *tw, float *GTw, float *dump) const {
        float tb[3][NH];                          // T[slot 4 s + k = (row s, column k)][channel]
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int h = 0; h < NH; ++h) tb[s][h] = tw[(iy + s) * L::ROWP + (ix + k) * C + 16 * h + cc];
        for (int i0 = js; i0 < je; i0 += 16) {
            const int row = min(i0 + m, 63);
            f32x4 acc[NH];
#pragma unroll
            for (int h = 0; h < NH; ++h) acc[h] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const float a = KA[row * 12 + 4 * s + k];
#pragma unroll
                for (int h = 0; h < NH; ++h) acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, tb[s][h], acc[h], 0, 0, 0);
            }
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int smp = i0 + 4 * k + v;
                    float *dst = (smp < je && m < C) ? GTw + (16 * h + m) * L::PT + smp : dump;
                    *dst = acc[h][v];
                }
        }
    }
};

// runs of the wave: heads = first lane of every maximal stretch of equal group key
__device__ __forceinline__ uint64_t run_heads(uint32_t akey) {
    const int lane = threadIdx.x & 63;
    const uint32_t prev = (uint32_t)__shfl_up((int)akey, 1);
    return __ballot(lane == 0 || akey != prev);
}

// ---- stream loads -----------------------------------------------------------------------------------
// channel c of this lane's sample; channels >= Cv do not exist (C padded up to a supported count)
template <int C, typename T>
struct StreamRegs {
    T raw[C];
    __device__ __forceinline__ void issue(const T *chan0_p, int64_t P, int Cv) {
#pragma unroll
        for (int c = 0; c < C; ++c) raw[c] = __builtin_nontemporal_load(chan0_p + (int64_t)(c < Cv ? c : Cv - 1) * P);
    }
    float val[C];
    // the loads have to have arrived HERE (while nothing but loads is outstanding, see the kernels): pin the conversion
    __device__ __forceinline__ void settle(int Cv) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            val[c] = c < Cv ? (float)raw[c] : 0.0f;
            asm volatile("" : "+v"(val[c]));
        }
    }
    // -> the channel-major LDS rows: row c, column = this lane's sample
    __device__ __forceinline__ void to_rows(float *GT, int pt) const {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int c = 0; c < C; ++c) GT[c * pt + lane] = val[c];
    }
};
template <typename T>
__device__ __forceinline__ void store_out(T *p, float v) { __builtin_nontemporal_store((T)v, p); }
__device__ __forceinline__ void store_out(float *p, float v) { tl::st_stream_wt(p, v); }

// the wave's LDS slice
template <int C, bool TWO, bool YBUF>
struct Slice {
    float *GT, *HT, *KA, *KB, *YB, *TW, *AW;
    __device__ __forceinline__ Slice(float *lds) {
        using L = Lay<C>;
        float *p = lds + (threadIdx.x >> 6) * wave_floats<C>(TWO, YBUF);
        GT = p;
        p += C * L::PT;
        HT = p;
        if (TWO) p += C * L::PT;
        KA = p;
        p += 64 * 12;
        KB = p;
        if (TWO) p += 64 * 12;
        YB = YBUF ? p : KA;      // without a buffer of its own the products go over the coefficient blocks (done with)
        if (YBUF) p += 64 * 12;
        TW = p;
        p += L::WIN;
        AW = p;
    }
};

// A wave owns `chunk` consecutive samples of one table and walks them 64 at a time
struct WaveJob {
    int n, lane;
    int64_t p_begin, p_end;
    __device__ __forceinline__ WaveJob(const Dims &d, int chunk) {
        n = blockIdx.y;
        lane = threadIdx.x & 63;
        const int64_t wv = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        p_begin = wv * chunk;
        p_end = min(d.P, p_begin + chunk);
    }
    __device__ __forceinline__ bool empty() const { return p_begin >= p_end; }
    __device__ __forceinline__ int64_t clamp(int64_t p) const { return min(p, p_end - 1); }
};

// =====================================================================================================
// The three stages.  Launch: grid (ceil(ceil(P / chunk) / 4), N), 256 threads = four independent waves.
//
// Global memory operations of a wave share ONE in-order counter (loads, stores and atomics alike), and the compiler waits
// for all of them whenever a load's result is needed while a store is outstanding.  So a batch is arranged to wait once, at
// its top:   [wait]  outputs of the PREVIOUS batch leave (they sat in LDS / registers)  ->  this batch's geometry and LDS
// rows  ->  the NEXT batch's coordinate and stream loads go out  ->  LDS and matrix-core work only (the table comes from
// the window)  ->  next batch.  Loads and stores have a whole batch to complete before anything asks for them.  The
// windows' global traffic (table rows in, atomics out) happens when the wave moves to another tile: once per ~16 batches.
// `dbg`: experiments only (cs_debug_coherent_tuning): 1 no scatter-reduce, 2 no window flush, 4 no products / outputs.
// =====================================================================================================

// walk the runs of the batch: fn(js, je, ix, iy) for every run that touches a node
template <int C, typename F>
__device__ __forceinline__ void for_each_run(uint32_t akey, Windows<C> &w, const float *__restrict__ tab_n,
                                             float *__restrict__ acc_n, const Dims &d, F fn) {
    uint64_t heads = run_heads(akey);
    while (heads) {
        const int js = __ffsll((unsigned long long)heads) - 1;
        heads &= heads - 1;
        const int je = heads ? __ffsll((unsigned long long)heads) - 1 : 64;
        const uint32_t key = (uint32_t)__builtin_amdgcn_readlane((int)akey, js);
        if (key == KEY_NONE) continue;      // samples that touch no node: nothing to add, their outputs are zero
        int ix, iy;
        w.place(key, ix, iy, tab_n, acc_n, d);
        fn(js, je, ix, iy);
    }
}

// first backward (2d.cu:406-506): grad_grid per sample, grad_input through the window
template <int KERNEL, int CQ, typename ST>
__global__ __launch_bounds__(256, 3) void backward(const ST *__restrict__ gOut, const float *__restrict__ icl,
                                                const float *__restrict__ grid, const float *__restrict__ offset,
                                                float *__restrict__ acc, float *__restrict__ grad_grid, Dims d, Flags f,
                                                int chunk, int dbg) {
    constexpr int C = 4 * CQ;
    using L = Lay<C>;
    extern __shared__ float lds[];
    Slice<C, false, false> sl(lds);
    const WaveJob job(d, chunk);
    if (job.empty()) return;
    const int lane = job.lane, n = job.n;
    const float off = offset[n];
    const float *tab_n = icl + (int64_t)n * d.vol * C;
    float *acc_n = acc + (int64_t)n * d.vol * C;
    const ST *go_n = gOut + (int64_t)n * d.go_ns;
    Windows<C> w;
    w.init(sl.TW, sl.AW);
    w.noflush = (dbg & 2) != 0;
    const Blocks<C, false> bl(sl.GT, nullptr, sl.KA, nullptr);
    float2 xy;
    StreamRegs<C, ST> sg;
    {
        const int64_t p = job.clamp(job.p_begin + lane);
        xy = *reinterpret_cast<const float2 *>(grid + d.gpt(n, p) * 2);
        sg.issue(go_n + p, d.P, d.C);
    }
    float2 out = make_float2(0.f, 0.f);     // the previous batch's result, stored behind the next batch's wait
    int64_t out_p = -1;
    for (int64_t p0 = job.p_begin; p0 < job.p_end; p0 += 64) {
        const int64_t p = p0 + lane;
        const bool live = p < job.p_end;
        Geo g;
        make_geo<KERNEL, 1>(g, xy, off, d, f, live);      // the one wait of the batch: the loads issued a batch ago
        sg.settle(d.C);
        __builtin_amdgcn_sched_barrier(0);
        if (p0 + 64 < job.p_end) {                        // the next batch's loads: a whole batch to arrive
            const int64_t pn = job.clamp(p + 64);         // (two batches ahead was measured: no difference)
            xy = *reinterpret_cast<const float2 *>(grid + d.gpt(n, pn) * 2);
            sg.issue(go_n + pn, d.P, d.C);
        }
        if (out_p >= 0) *reinterpret_cast<float2 *>(grad_grid + ((int64_t)n * d.P + out_p) * 2) = out;
        __builtin_amdgcn_sched_barrier(0);
        put_block(sl.KA + lane * 12, g.W, g.sx, g.sy, g.akey != KEY_NONE);
        sg.to_rows(sl.GT, L::PT);
        wave_sync();
        // the products go over the coefficient blocks' memory: only after the run's scatter has read them
        for_each_run<C>(g.akey, w, tab_n, acc_n, d, [&](int js, int je, int ix, int iy) {
            if (!(dbg & 1)) bl.scatter(js, je, ix, iy, w.aw);
            wave_sync();
            if (!(dbg & 4)) bl.products(js, je, ix, iy, w.tw, sl.YB, sl.GT + 64);
        });
        wave_sync();
        float gx = 0.0f, gy = 0.0f;
        if (g.akey != KEY_NONE) {
            const float *y = sl.YB + lane * 12 + g.slot0();
            const float d0 = y[0], d1 = y[1], d2 = y[4], d3 = y[5];
            gx = g.ax[1].w[0] * (d1 - d0) + g.ax[1].w[1] * (d3 - d2);
            gy = g.ax[0].w[0] * (d2 - d0) + g.ax[0].w[1] * (d3 - d1);
        }
        out = make_float2(g.ax[0].d1 * gx, g.ax[1].d1 * gy);
        out_p = live ? p : -1;
        wave_sync();
    }
    if (out_p >= 0) *reinterpret_cast<float2 *>(grad_grid + ((int64_t)n * d.P + out_p) * 2) = out;
    w.flush(acc_n, d);
}

// the previous batch's grad_grad_out rows leave from the payload rows they were written over
template <int C, typename ST>
__device__ __forceinline__ void store_rows(const float *GT, ST *ggo_n, int64_t P, int64_t p, int Cv, bool valid) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < C; ++c)
        if (c < Cv) store_out(ggo_n + (int64_t)c * P + p, valid ? GT[c * Lay<C>::PT + lane] : 0.0f);
}

// second backward (2d.cu:569-716), grad_out_input absent
template <int KERNEL, int CQ, typename ST>
__global__ __launch_bounds__(256, 3) void bb(const float *__restrict__ cG, const float *__restrict__ icl,
                                          const float *__restrict__ grid, const ST *__restrict__ gOut,
                                          const float *__restrict__ offset, float *__restrict__ acc,
                                          float *__restrict__ gGrid, ST *__restrict__ ggOut, Dims d, Flags f, int chunk,
                                          int dbg) {
    constexpr int C = 4 * CQ;
    using L = Lay<C>;
    extern __shared__ float lds[];
    Slice<C, false, true> sl(lds);
    const WaveJob job(d, chunk);
    if (job.empty()) return;
    const int lane = job.lane, n = job.n;
    const float off = offset[n];
    const float *tab_n = icl + (int64_t)n * d.vol * C;
    float *acc_n = acc + (int64_t)n * d.vol * C;
    const ST *go_n = gOut + (int64_t)n * d.go_ns;
    ST *ggo_n = ggOut + (int64_t)n * d.C * d.P;
    Windows<C> w;
    w.init(sl.TW, sl.AW);
    w.noflush = (dbg & 2) != 0;
    const Blocks<C, false> bl(sl.GT, nullptr, sl.KA, nullptr);
    float2 xy, cg = make_float2(0.f, 0.f);
    StreamRegs<C, ST> sg;
    {
        const int64_t p = job.clamp(job.p_begin + lane);
        xy = *reinterpret_cast<const float2 *>(grid + d.gpt(n, p) * 2);
        if (cG) cg = *reinterpret_cast<const float2 *>(cG + d.gpt(n, p) * 2);
        sg.issue(go_n + p, d.P, d.C);
    }
    float2 out = make_float2(0.f, 0.f);
    int64_t out_p = -1;
    bool out_valid = false;
    for (int64_t p0 = job.p_begin; p0 < job.p_end; p0 += 64) {
        const int64_t p = p0 + lane;
        const bool live = p < job.p_end;
        Geo g;
        make_geo<KERNEL, 2>(g, xy, off, d, f, live);      // the one wait of the batch: the loads issued a batch ago
        sg.settle(d.C);
        const float2 cgb = cg;
        __builtin_amdgcn_sched_barrier(0);
        if (p0 + 64 < job.p_end) {                        // the next batch's loads: a whole batch to arrive
            const int64_t pn = job.clamp(p + 64);
            xy = *reinterpret_cast<const float2 *>(grid + d.gpt(n, pn) * 2);
            if (cG) cg = *reinterpret_cast<const float2 *>(cG + d.gpt(n, pn) * 2);
            sg.issue(go_n + pn, d.P, d.C);
        }
        if (out_p >= 0) {                                 // the previous batch's outputs leave, behind the wait
            store_rows<C>(sl.GT, ggo_n, d.P, out_p, d.C, out_valid);
            *reinterpret_cast<float2 *>(gGrid + ((int64_t)n * d.P + out_p) * 2) = out;
        }
        __builtin_amdgcn_sched_barrier(0);
        float Sx[4], Sy[4];
        {
            float Dm[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                Dm[a] = g.first(a, 0) * cgb.x + g.first(a, 1) * cgb.y;
                Sx[a] = g.pure2(a, 0) * cgb.x;                // pure second derivatives only (2d.cu:705-706)
                Sy[a] = g.pure2(a, 1) * cgb.y;
                if (f.exact) {
                    const float mx = g.mixed2(a);
                    Sx[a] = fmaf(mx, cgb.y, Sx[a]);
                    Sy[a] = fmaf(mx, cgb.x, Sy[a]);
                }
            }
            put_block(sl.KA + lane * 12, Dm, g.sx, g.sy, g.akey != KEY_NONE);
        }
        wave_sync();                                  // the stores have read the rows that are overwritten now
        sg.to_rows(sl.GT, L::PT);
        wave_sync();
        for_each_run<C>(g.akey, w, tab_n, acc_n, d, [&](int js, int je, int ix, int iy) {
            if (!(dbg & 1)) bl.scatter(js, je, ix, iy, w.aw);
            if (!(dbg & 4)) bl.products(js, je, ix, iy, w.tw, sl.YB, sl.GT + 64);
            wave_sync();
            if (!(dbg & 4)) bl.outputs(js, je, ix, iy, w.tw, sl.GT, sl.GT + 64);      // over the run's cotangents: last
        });
        wave_sync();
        float sx = 0.0f, sy = 0.0f;
        out_valid = g.akey != KEY_NONE;
        if (out_valid) {
            const float *y = sl.YB + lane * 12 + g.slot0();
            const float ya[4] = {y[0], y[1], y[4], y[5]};
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                sx = fmaf(Sx[a], ya[a], sx);
                sy = fmaf(Sy[a], ya[a], sy);
            }
        }
        out = make_float2(sx, sy);
        out_p = live ? p : -1;
        wave_sync();
    }
    if (out_p >= 0) {
        store_rows<C>(sl.GT, ggo_n, d.P, out_p, d.C, out_valid);
        *reinterpret_cast<float2 *>(gGrid + ((int64_t)n * d.P + out_p) * 2) = out;
    }
    w.flush(acc_n, d);
}

// fused third backward (2d.cu:774-890 + the extra second backward of modules_2d.py:106-111):
// grad_input += gOut * E  (+ hO * D with TWO);  grad_grad_out = sum_a input[q_a] * E_a.
// With TWO the batch is walked twice over the same LDS rows -- (gOut, E) then (hO, D) -- instead of holding both pairs:
// 13 KiB of LDS per wave instead of 20, i.e. 12 waves per CU instead of 7.
template <int KERNEL, int CQ, bool TWO, typename ST>
__global__ __launch_bounds__(256, 3) void bbb(const float *__restrict__ icl, const float *__restrict__ grid,
                                           const ST *__restrict__ gOut, const float *__restrict__ cG,
                                           const float *__restrict__ hG, const ST *__restrict__ hO,
                                           const float *__restrict__ offset, float *__restrict__ acc,
                                           ST *__restrict__ ggOut, Dims d, Flags f, int chunk, int dbg) {
    constexpr int C = 4 * CQ;
    using L = Lay<C>;
    extern __shared__ float lds[];
    Slice<C, false, false> sl(lds);
    const WaveJob job(d, chunk);
    if (job.empty()) return;
    const int lane = job.lane, n = job.n;
    const float off = offset[n];
    const float *tab_n = icl + (int64_t)n * d.vol * C;
    float *acc_n = acc + (int64_t)n * d.vol * C;
    const ST *go_n = gOut + (int64_t)n * d.go_ns;
    const ST *ho_n = TWO ? hO + (int64_t)n * d.ho_ns : nullptr;
    ST *ggo_n = ggOut + (int64_t)n * d.C * d.P;
    Windows<C> w;
    w.init(sl.TW, sl.AW);
    w.noflush = (dbg & 2) != 0;
    const Blocks<C, false> bl(sl.GT, nullptr, sl.KA, nullptr);
    float2 xy, cg = make_float2(0.f, 0.f), hg = make_float2(0.f, 0.f);
    StreamRegs<C, ST> sg, sh;
    {
        const int64_t p = job.clamp(job.p_begin + lane);
        xy = *reinterpret_cast<const float2 *>(grid + d.gpt(n, p) * 2);
        if (cG) cg = *reinterpret_cast<const float2 *>(cG + d.gpt(n, p) * 2);
        if (hG) hg = *reinterpret_cast<const float2 *>(hG + d.gpt(n, p) * 2);
        sg.issue(go_n + p, d.P, d.C);
        if (TWO) sh.issue(ho_n + p, d.P, d.C);
    }
    int64_t out_p = -1;
    bool out_valid = false;
    for (int64_t p0 = job.p_begin; p0 < job.p_end; p0 += 64) {
        const int64_t p = p0 + lane;
        const bool live = p < job.p_end;
        Geo g;
        make_geo<KERNEL, 2>(g, xy, off, d, f, live);      // the one wait of the batch: the loads issued a batch ago
        sg.settle(d.C);
        float hval[TWO ? C : 1];
        if (TWO) {
            sh.settle(d.C);
#pragma unroll
            for (int c = 0; c < C; ++c) hval[c] = sh.val[c];
        }
        const float2 cgb = cg, hgb = hg;
        __builtin_amdgcn_sched_barrier(0);
        if (p0 + 64 < job.p_end) {                        // the next batch's loads: a whole batch to arrive
            const int64_t pn = job.clamp(p + 64);
            xy = *reinterpret_cast<const float2 *>(grid + d.gpt(n, pn) * 2);
            if (cG) cg = *reinterpret_cast<const float2 *>(cG + d.gpt(n, pn) * 2);
            if (hG) hg = *reinterpret_cast<const float2 *>(hG + d.gpt(n, pn) * 2);
            sg.issue(go_n + pn, d.P, d.C);
            if (TWO) sh.issue(ho_n + pn, d.P, d.C);
        }
        if (out_p >= 0) store_rows<C>(sl.GT, ggo_n, d.P, out_p, d.C, out_valid);   // the previous batch's outputs leave
        __builtin_amdgcn_sched_barrier(0);
        float Dm[4], Em[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            Dm[a] = g.first(a, 0) * cgb.x + g.first(a, 1) * cgb.y;
            Em[a] = g.pure2(a, 0) * (hgb.x * cgb.x) + g.pure2(a, 1) * (hgb.y * cgb.y);   // 2d.cu:876
            if (f.exact) Em[a] = fmaf(g.mixed2(a), hgb.x * cgb.y + hgb.y * cgb.x, Em[a]);
        }
        const bool valid = g.akey != KEY_NONE;
        wave_sync();                                  // the stores have read the rows that are overwritten now
        if (TWO) {                                    // first (hO, D): grad_input only
            put_block(sl.KA + lane * 12, Dm, g.sx, g.sy, valid);
#pragma unroll
            for (int c = 0; c < C; ++c) sl.GT[c * L::PT + lane] = hval[c];
            wave_sync();
            for_each_run<C>(g.akey, w, tab_n, acc_n, d, [&](int js, int je, int ix, int iy) {
                if (!(dbg & 1)) bl.scatter(js, je, ix, iy, w.aw);
            });
            wave_sync();
        }
        put_block(sl.KA + lane * 12, Em, g.sx, g.sy, valid);   // then (gOut, E): grad_input and grad_grad_out
        sg.to_rows(sl.GT, L::PT);
        wave_sync();
        for_each_run<C>(g.akey, w, tab_n, acc_n, d, [&](int js, int je, int ix, int iy) {
            if (!(dbg & 1)) bl.scatter(js, je, ix, iy, w.aw);
            wave_sync();
            if (!(dbg & 4)) bl.outputs(js, je, ix, iy, w.tw, sl.GT, sl.GT + 64);
        });
        out_valid = valid;
        out_p = live ? p : -1;
        wave_sync();
    }
    if (out_p >= 0) store_rows<C>(sl.GT, ggo_n, d.P, out_p, d.C, out_valid);
    w.flush(acc_n, d);
}

}  // namespace coh
}  // namespace cs
