// cs_coherent.cuh -- the 2D backward stages for COHERENT point sets (CS_POINTS_COHERENT): consecutive samples of
// one table fall into the same or neighbouring cells, which is what a caller gets by ordering its collocation
// points once (cs2d_sort_points; PIXEL draws a fixed random set and re-uses it every step, reference
// test/test_2d.py:28-38).  Then nothing has to be moved to where it is needed: no plan, no p-ordered records,
// no fetch by sample id.  Replaces the scatter loops of the reference, 2d.cu:464-505, :661-712, :850-888.
//
//   * a WAVE takes 64 consecutive samples of one table (launch geometry of the tiled path's point kernels: many short
//     independent waves -- a wave that walked a long run of samples with loads issued ahead was measured first and lost:
//     loads, stores and atomics share one in-order counter, so every wait for a node row also waited for the
//     prefetch and for the previous outputs);
//   * lane = sample does the geometry and takes the coalesced stream loads; the cotangents and the coefficients go
//     to the wave's LDS rows -- and stay there;
//   * CQ lanes per sample gather the node rows (L1 / L2 hits: neighbours share them) and form the per-sample
//     products, exactly as the tiled path's point kernels;
//   * the scatter-reduce.  Samples are grouped by the cell of the UN-SHIFTED point: with the multicell shift n/N a point
//     set ordered for table 0 alternates between up to four cells of table n, but always inside the 3 x 3 nodes around
//     the un-shifted cell.  Each sample's four coefficients are laid out as a 3 x 4 block over those nodes, and the sum
//     over a run of equal cell of  block x cotangents  -- (12 x m) x (m x C) -- is taken by the matrix core in exact fp32
//     (v_mfma_f32_16x16x4_f32, four samples per instruction, operands one dword per lane straight from the LDS rows; run
//     bounds are wave-uniform, from a ballot).  The result is added ONCE per run to the wave's private LDS window, a
//     WNY x WN-node image of grad_input: plain read-modify-write, the wave is the only writer;
//   * at its end (and whenever a run falls outside the window: next tile of the caller's order, or any jump of an
//     unordered set) the wave adds the touched part of the window to the channels-last accumulator with whole-row
//     float atomics.  Correct for ANY order of the points; fast when the order is coherent: ~20 bytes of atomics per
//     sample instead of 4 rows.
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
    static constexpr int NH = C > 16 ? C / 16 : 1;     // 16-channel halves (one accumulator tile each)
    static constexpr int ROWP = (WN + 1) * C;          // floats per window row (C = 16: rows r, r+1 on disjoint banks)
    static constexpr int WIN = WNY * ROWP;
};

// per wave (floats): payload rows G (and H), coefficient rows KA (and KB), node / result block, per-corner
// coefficients for the per-sample products, the window
template <int C>
__host__ __device__ constexpr int wave_floats(bool two, int co_fields) {
    return (two ? 2 : 1) * (64 * C + 64 * 12) + tl::QREC + co_fields * 64 + Lay<C>::WIN;
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
    uint32_t node[4];
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
};

template <int KERNEL, int ORDER>
__device__ __forceinline__ void make_geo(Geo &g, float2 xy, float off, const Dims &d, const Flags &f, bool live) {
    g.ax[0] = make_axis<KERNEL, ORDER>(xy.x, d.size[0], f, f.align, off);
    g.ax[1] = make_axis<KERNEL, ORDER>(xy.y, d.size[1], f, f.align, off);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        int x = g.ax[0].lo + (a & 1), y = g.ax[1].lo + (a >> 1);
        bool ok = x >= 0 && x < d.size[0] && y >= 0 && y < d.size[1];
        g.node[a] = ok ? (uint32_t)(y * d.size[0] + x) : tl::NO_NODE;
        g.W[a] = g.ax[0].w[a & 1] * g.ax[1].w[a >> 1];
    }
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

// ---- the window ---------------------------------------------------------------------------------
template <int C>
struct Window {
    using L = Lay<C>;
    float *win;
    int ax, ay;      // anchor in u coordinates (u = lo + 1), multiples of TS; window node (iy, ix) = table node (ay-1+iy, ax-1+ix)
    int ylo, yhi;    // rows of the window touched since the last flush
    bool noflush = false;
    __device__ __forceinline__ void init(float *w) {
        win = w;
        ax = ay = -(1 << 20);
        ylo = WNY;
        yhi = -1;
        for (int i = threadIdx.x & 63; i < L::WIN; i += 64) win[i] = 0.0f;
    }
    // touched rows -> the channels-last accumulator, WN*C contiguous floats per row: whole-line float atomics
    __device__ __forceinline__ void flush(float *__restrict__ acc_n, const Dims &d) {
        const int lane = threadIdx.x & 63;
        if (noflush) { ylo = WNY; yhi = -1; return; }
        for (int iy = ylo; iy <= yhi; ++iy) {
            const int gy = ay - 1 + iy;
            float *wrow = win + iy * L::ROWP;
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
};

// ---- the scatter-reduce over one batch -------------------------------------------------------------
// A run = a maximal stretch of samples grouped by one cell (equal akey); its bounds are wave-uniform (ballot).  The sum
// over a run of  coefficient block (3 x 4 nodes) x payload (C channels)  is a small dense product, so the matrix core
// does it: v_mfma_f32_16x16x4_f32, four samples per instruction, exact fp32 (a k-ordered fma chain), operands one dword
// per lane straight from the LDS rows -- A = the blocks (lane (k, m): node slot m of sample 4t+k), B = the payloads
// (lane (k, c): channel c of sample 4t+k).  Lane (r, c) of the result holds row r of the block for channel c in its four
// registers; at the end of a run it adds them to the window: plain read-modify-write, the wave is the only writer.
// A group of four samples that straddles runs is issued once per run with the other samples' blocks masked to zero.
template <int C, bool TWO>
struct Scatter {
    using L = Lay<C>;
    static constexpr int NH = L::NH;
    f32x4 D[NH];
    bool open;
    int ix, iy;
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int h = 0; h < NH; ++h) D[h] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __device__ __forceinline__ void close(Window<C> &w) {
        if (!open) return;
        const int lane = threadIdx.x & 63, r = lane >> 4, c = lane & 15;
        if (r < 3 && c < C) {
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                float *wp = w.win + (iy + r) * L::ROWP + ix * C + 16 * h + c;
                wp[0] += D[h][0];
                wp[C] += D[h][1];
                wp[2 * C] += D[h][2];
            }
        }
        clear();
        open = false;
    }
    __device__ __forceinline__ void begin(uint32_t key, Window<C> &w, float *__restrict__ acc_n, const Dims &d) {
        if (key == KEY_NONE) return;    // samples that touch no node: their blocks are zero, nothing to add
        const int kx = (int)(key & 0x7FFFu), ky = (int)(key >> 15);
        ix = kx - w.ax;
        iy = ky - w.ay;
        if ((unsigned)ix >= (unsigned)TS || (unsigned)iy > (unsigned)(WNY - 3)) {
            // the run is outside the window: empty it and anchor it on this cell's row and tile column
            w.flush(acc_n, d);
            w.ax = kx / TS * TS;
            w.ay = ky;
            ix = kx - w.ax;
            iy = ky - w.ay;
        }
        w.ylo = min(w.ylo, iy);
        w.yhi = max(w.yhi, iy + 2);
        open = true;
    }
    // one batch: G / H payload rows [64][C], KA / KB block rows [64][12]; akey: this lane's (= sample's) group key
    __device__ __forceinline__ void batch(const float *G, const float *H, const float *KA, const float *KB, uint32_t akey,
                                          Window<C> &w, float *__restrict__ acc_n, const Dims &d) {
        const int lane = threadIdx.x & 63, k = lane >> 4, m = lane & 15;
        const uint32_t prev = (uint32_t)__shfl_up((int)akey, 1);
        const uint64_t heads = __ballot(lane == 0 || akey != prev);
        open = false;
        clear();
        const bool am = m < 12, bm = m < C;
        const float *ka = KA + k * 12 + (am ? m : 0), *kb = KB + k * 12 + (am ? m : 0);
        const float *gb = G + k * C + (bm ? m : 0), *hb = H + k * C + (bm ? m : 0);
#pragma unroll
        for (int t0 = 0; t0 < 16; t0 += 8) {
            float a[8], a2[8], b[8][NH], b2[8][NH];
#pragma unroll
            for (int u = 0; u < 8; ++u) {          // every operand of eight groups first: none depends on the runs
                const int t = t0 + u;
                a[u] = ka[t * 48];
                if (TWO) a2[u] = kb[t * 48];
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    b[u][h] = gb[t * 4 * C + 16 * h];
                    if (TWO) b2[u][h] = hb[t * 4 * C + 16 * h];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + u;
                const float av = am ? a[u] : 0.0f, av2 = TWO ? (am ? a2[u] : 0.0f) : 0.0f;
                float bv[NH], bv2[NH];
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    bv[h] = bm ? b[u][h] : 0.0f;
                    bv2[h] = TWO ? (bm ? b2[u][h] : 0.0f) : 0.0f;
                }
                const uint32_t hbits = (uint32_t)(heads >> (4 * t)) & 0xFu;
                if (hbits == 0) {                  // the whole group continues the open run
#pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        D[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[h], D[h], 0, 0, 0);
                        if (TWO) D[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(av2, bv2[h], D[h], 0, 0, 0);
                    }
                    continue;
                }
                int s = 0;
                while (s < 4) {
                    if ((hbits >> s) & 1u) {
                        close(w);
                        begin((uint32_t)__builtin_amdgcn_readlane((int)akey, 4 * t + s), w, acc_n, d);
                    }
                    const uint32_t later = hbits >> (s + 1);
                    const int e = later ? s + 1 + (__ffs((int)later) - 1) : 4;
                    const bool in = k >= s && k < e;
#pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        D[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(in ? av : 0.0f, bv[h], D[h], 0, 0, 0);
                        if (TWO) D[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(in ? av2 : 0.0f, bv2[h], D[h], 0, 0, 0);
                    }
                    s = e;
                }
            }
        }
        close(w);
    }
};

// ---- stream loads, one batch ahead ----------------------------------------------------------------
// channel c of this lane's sample; channels >= Cv do not exist (C padded up to a supported count)
template <int C, typename T>
struct StreamAhead {
    T raw[C];
    __device__ __forceinline__ void issue(const T *chan0_p, int64_t P, int Cv) {
#pragma unroll
        for (int c = 0; c < C; ++c) raw[c] = __builtin_nontemporal_load(chan0_p + (int64_t)(c < Cv ? c : Cv - 1) * P);
    }
    __device__ __forceinline__ void to_row(float *row, int Cv) const {
#pragma unroll
        for (int q = 0; q < C / 4; ++q)
            *reinterpret_cast<float4 *>(row + 4 * q) =
                make_float4(4 * q < Cv ? (float)raw[4 * q] : 0.0f, 4 * q + 1 < Cv ? (float)raw[4 * q + 1] : 0.0f,
                            4 * q + 2 < Cv ? (float)raw[4 * q + 2] : 0.0f, 4 * q + 3 < Cv ? (float)raw[4 * q + 3] : 0.0f);
    }
};
template <typename T>
__device__ __forceinline__ void store_out(T *p, float v) { __builtin_nontemporal_store((T)v, p); }
__device__ __forceinline__ void store_out(float *p, float v) { tl::st_stream_wt(p, v); }

__device__ __forceinline__ void put_nodes(float *rec, int r, const Geo &g) {
    uint32_t *ru = reinterpret_cast<uint32_t *>(rec);
#pragma unroll
    for (int a = 0; a < 4; ++a) ru[a * 64 + r] = g.node[a];
}

// node rows of one pass, RAW: every load of every pass goes out before anything looks at a result (a select on the
// loaded value right behind the loads makes the compiler wait for them pass by pass); rows of nodes outside the table
// are read from node 0 and masked when they are used (mask_rows)
template <int CQ>
__device__ __forceinline__ void gather_raw(const float4 *tab, const float *rec, int sl, int q, float4 (&v)[4]) {
    const uint32_t *ru = reinterpret_cast<const uint32_t *>(rec);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const uint32_t nd = ru[a * 64 + sl];
        v[a] = tab[(nd == tl::NO_NODE ? 0u : nd) * CQ + q];
    }
}
__device__ __forceinline__ void mask_rows(const float *rec, int sl, float4 (&v)[4]) {
    const uint32_t *ru = reinterpret_cast<const uint32_t *>(rec);
#pragma unroll
    for (int a = 0; a < 4; ++a)
        if (ru[a * 64 + sl] == tl::NO_NODE) v[a] = zero4();
}

// the wave's LDS slice
template <int C, bool TWO, int CO>
struct Slice {
    float *G, *H, *KA, *KB, *rec, *co, *win;
    __device__ __forceinline__ Slice(float *lds) {
        float *p = lds + (threadIdx.x >> 6) * wave_floats<C>(TWO, CO);
        G = p;
        p += 64 * C;
        H = p;
        if (TWO) p += 64 * C;
        KA = p;
        p += 64 * 12;
        KB = p;
        if (TWO) p += 64 * 12;
        rec = p;
        p += tl::QREC;
        co = p;
        p += CO * 64;
        win = p;
    }
};

// =====================================================================================================
// The three stages.  Launch: grid (ceil(P/256), N), 256 threads = four independent waves of 64 consecutive samples.
// Order of the memory operations in a wave (they share one in-order counter): stream loads -> node rows -> [LDS and
// matrix-core work] -> outputs -> the window's atomics last, nothing waits behind them.
// `dbg`: experiments only (cs_debug_coherent_tuning): 1 no scatter-reduce, 2 no window flush, 4 no node rows.
// =====================================================================================================
struct WaveId {
    int n, lane;
    int64_t p;      // this lane's sample (clamped to the last one of the table)
    bool live;
    __device__ __forceinline__ WaveId(const Dims &d) {
        n = blockIdx.y;
        lane = threadIdx.x & 63;
        const int64_t pp = (int64_t)blockIdx.x * 256 + threadIdx.x;
        live = pp < d.P;
        p = live ? pp : d.P - 1;
    }
    __device__ __forceinline__ bool wave_empty(const Dims &d) const {
        return (int64_t)blockIdx.x * 256 + (threadIdx.x & ~63) >= d.P;
    }
};

// first backward (2d.cu:406-506): grad_grid per sample, grad_input through the window
template <int KERNEL, int CQ, typename ST>
__global__ __launch_bounds__(256) void backward(const ST *__restrict__ gOut, const float *__restrict__ icl,
                                                const float *__restrict__ grid, const float *__restrict__ offset,
                                                float *__restrict__ acc, float *__restrict__ grad_grid, Dims d, Flags f,
                                                int dbg) {
    constexpr int C = 4 * CQ, CO = 4;
    extern __shared__ float lds[];
    Slice<C, false, CO> sl_(lds);
    float *G = sl_.G, *KA = sl_.KA, *rec = sl_.rec, *co = sl_.co;
    const WaveId id(d);
    if (id.wave_empty(d)) return;
    const int lane = id.lane, n = id.n, q = lane % CQ;
    const float4 *tab = reinterpret_cast<const float4 *>(icl + (int64_t)n * d.vol * C);
    float *acc_n = acc + (int64_t)n * d.vol * C;
    const float2 xy = *reinterpret_cast<const float2 *>(grid + d.gpt(n, id.p) * 2);
    StreamAhead<C, ST> sa;
    sa.issue(gOut + (int64_t)n * d.go_ns + id.p, d.P, d.C);
    Window<C> w;
    w.init(sl_.win);
    w.noflush = (dbg & 2) != 0;
    Scatter<C, false> sc;
    Geo g;
    make_geo<KERNEL, 1>(g, xy, offset[n], d, f, id.live);
    sa.to_row(G + lane * C, d.C);
    put_block(KA + lane * 12, g.W, g.sx, g.sy, g.akey != KEY_NONE);
    co[lane] = g.ax[0].w[0];
    co[64 + lane] = g.ax[0].w[1];
    co[128 + lane] = g.ax[1].w[0];
    co[192 + lane] = g.ax[1].w[1];
    put_nodes(rec, lane, g);
    wave_sync();
    float4 vv[CQ][4];
#pragma unroll
    for (int sub = 0; sub < CQ; ++sub) {
        if (dbg & 4) vv[sub][0] = vv[sub][1] = vv[sub][2] = vv[sub][3] = zero4();
        else gather_raw<CQ>(tab, rec, sub * (64 / CQ) + lane / CQ, q, vv[sub]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!(dbg & 1)) sc.batch(G, nullptr, KA, nullptr, g.akey, w, acc_n, d);   // while the node rows are in flight
#pragma unroll
    for (int sub = 0; sub < CQ; ++sub) {
        const int sl = sub * (64 / CQ) + lane / CQ;
        float4(&v)[4] = vv[sub];
        mask_rows(rec, sl, v);
        const float4 g4 = *reinterpret_cast<const float4 *>(G + sl * C + 4 * q);
        const float wx0 = co[sl], wx1 = co[64 + sl], wy0 = co[128 + sl], wy1 = co[192 + sl];
        float d0 = dot4(v[0], g4), d1 = dot4(v[1], g4), d2 = dot4(v[2], g4), d3 = dot4(v[3], g4);
        float gx = tl::q_reduce<CQ>(wy0 * (d1 - d0) + wy1 * (d3 - d2));
        float gy = tl::q_reduce<CQ>(wx0 * (d2 - d0) + wx1 * (d3 - d1));
        if (q == 0) {
            rec[4 * 64 + sl] = gx;
            rec[5 * 64 + sl] = gy;
        }
    }
    wave_sync();
    if (id.live)
        *reinterpret_cast<float2 *>(grad_grid + ((int64_t)n * d.P + id.p) * 2) =
            make_float2(g.ax[0].d1 * rec[4 * 64 + lane], g.ax[1].d1 * rec[5 * 64 + lane]);
    w.flush(acc_n, d);
}

// second backward (2d.cu:569-716).  co: D[4], Sx[4], Sy[4], (W[4] with HAS_CI)
template <int KERNEL, int CQ, bool HAS_CI, typename ST>
__global__ __launch_bounds__(256) void bb(const float *__restrict__ cIcl, const float *__restrict__ cG,
                                          const float *__restrict__ icl, const float *__restrict__ grid,
                                          const ST *__restrict__ gOut, const float *__restrict__ offset,
                                          float *__restrict__ acc, float *__restrict__ gGrid, ST *__restrict__ ggOut,
                                          Dims d, Flags f, int dbg) {
    constexpr int C = 4 * CQ, CO = 16;
    extern __shared__ float lds[];
    Slice<C, false, CO> sl_(lds);
    float *G = sl_.G, *KA = sl_.KA, *rec = sl_.rec, *co = sl_.co;
    const WaveId id(d);
    if (id.wave_empty(d)) return;
    const int lane = id.lane, n = id.n, q = lane % CQ;
    const float4 *tab = reinterpret_cast<const float4 *>(icl + (int64_t)n * d.vol * C);
    const float4 *ctab = HAS_CI ? reinterpret_cast<const float4 *>(cIcl + (int64_t)n * d.vol * C) : nullptr;
    ST *ggo_n = ggOut + (int64_t)n * d.C * d.P;
    float *acc_n = acc + (int64_t)n * d.vol * C;
    const float2 xy = *reinterpret_cast<const float2 *>(grid + d.gpt(n, id.p) * 2);
    const float2 cg = cG ? *reinterpret_cast<const float2 *>(cG + d.gpt(n, id.p) * 2) : make_float2(0.f, 0.f);
    StreamAhead<C, ST> sa;
    sa.issue(gOut + (int64_t)n * d.go_ns + id.p, d.P, d.C);
    Window<C> w;
    w.init(sl_.win);
    w.noflush = (dbg & 2) != 0;
    Scatter<C, false> sc;
    Geo g;
    make_geo<KERNEL, 2>(g, xy, offset[n], d, f, id.live);
    sa.to_row(G + lane * C, d.C);
    {
        float Dm[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            Dm[a] = g.first(a, 0) * cg.x + g.first(a, 1) * cg.y;
            float sx = g.pure2(a, 0) * cg.x, sy = g.pure2(a, 1) * cg.y;   // pure second derivatives only (2d.cu:705-706)
            if (f.exact) {
                const float mx = g.mixed2(a);
                sx = fmaf(mx, cg.y, sx);
                sy = fmaf(mx, cg.x, sy);
            }
            co[a * 64 + lane] = Dm[a];
            co[(4 + a) * 64 + lane] = sx;
            co[(8 + a) * 64 + lane] = sy;
            if (HAS_CI) co[(12 + a) * 64 + lane] = g.W[a];
        }
        put_block(KA + lane * 12, Dm, g.sx, g.sy, g.akey != KEY_NONE);
    }
    put_nodes(rec, lane, g);
    wave_sync();
    float4 vv[CQ][4];
#pragma unroll
    for (int sub = 0; sub < CQ; ++sub) {
        if (dbg & 4) vv[sub][0] = vv[sub][1] = vv[sub][2] = vv[sub][3] = zero4();
        else gather_raw<CQ>(tab, rec, sub * (64 / CQ) + lane / CQ, q, vv[sub]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!(dbg & 1)) sc.batch(G, nullptr, KA, nullptr, g.akey, w, acc_n, d);
#pragma unroll
    for (int sub = 0; sub < CQ; ++sub) {
        const int sl = sub * (64 / CQ) + lane / CQ;
        float4(&v)[4] = vv[sub];
        mask_rows(rec, sl, v);
        float *row = G + sl * C;
        const float4 g4 = *reinterpret_cast<const float4 *>(row + 4 * q);
        float4 o = zero4(), tx = zero4(), ty = zero4();
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            o = fma4(co[a * 64 + sl], v[a], o);
            tx = fma4(co[(4 + a) * 64 + sl], v[a], tx);
            ty = fma4(co[(8 + a) * 64 + sl], v[a], ty);
        }
        if (HAS_CI) {   // + sum_a gOutInput[q_a] * W_a   (2d.cu:694-697)
            float4 u[4];
            tl::q_gather<CQ>(ctab, rec, sl, q, u);
#pragma unroll
            for (int a = 0; a < 4; ++a) o = fma4(co[(12 + a) * 64 + sl], u[a], o);
        }
        float sx = tl::q_reduce<CQ>(dot4(tx, g4)), sy = tl::q_reduce<CQ>(dot4(ty, g4));
        *reinterpret_cast<float4 *>(row + 4 * q) = o;   // over the cotangent quad (the scatter is done with it)
        if (q == 0) {
            rec[4 * 64 + sl] = sx;
            rec[5 * 64 + sl] = sy;
        }
    }
    wave_sync();
    if (id.live) {
        const float *row = G + lane * C;
#pragma unroll
        for (int c = 0; c < C; ++c)
            if (c < d.C) store_out(ggo_n + (int64_t)c * d.P + id.p, row[c]);
        *reinterpret_cast<float2 *>(gGrid + ((int64_t)n * d.P + id.p) * 2) = make_float2(rec[4 * 64 + lane], rec[5 * 64 + lane]);
    }
    w.flush(acc_n, d);
}

// fused third backward (2d.cu:774-890 + the extra second backward of modules_2d.py:106-111):
// grad_input += gOut * E  (+ hO * D with TWO);  grad_grad_out = sum_a input[q_a] * E_a
template <int KERNEL, int CQ, bool TWO, typename ST>
__global__ __launch_bounds__(256) void bbb(const float *__restrict__ icl, const float *__restrict__ grid,
                                           const ST *__restrict__ gOut, const float *__restrict__ cG,
                                           const float *__restrict__ hG, const ST *__restrict__ hO,
                                           const float *__restrict__ offset, float *__restrict__ acc,
                                           ST *__restrict__ ggOut, Dims d, Flags f, int dbg) {
    constexpr int C = 4 * CQ, CO = 4;
    extern __shared__ float lds[];
    Slice<C, TWO, CO> sl_(lds);
    float *G = sl_.G, *H = sl_.H, *KA = sl_.KA, *KB = sl_.KB, *rec = sl_.rec, *co = sl_.co;
    const WaveId id(d);
    if (id.wave_empty(d)) return;
    const int lane = id.lane, n = id.n, q = lane % CQ;
    const float4 *tab = reinterpret_cast<const float4 *>(icl + (int64_t)n * d.vol * C);
    ST *ggo_n = ggOut + (int64_t)n * d.C * d.P;
    float *acc_n = acc + (int64_t)n * d.vol * C;
    const float2 xy = *reinterpret_cast<const float2 *>(grid + d.gpt(n, id.p) * 2);
    const float2 cg = cG ? *reinterpret_cast<const float2 *>(cG + d.gpt(n, id.p) * 2) : make_float2(0.f, 0.f);
    const float2 hg = hG ? *reinterpret_cast<const float2 *>(hG + d.gpt(n, id.p) * 2) : make_float2(0.f, 0.f);
    StreamAhead<C, ST> sa, sh;
    sa.issue(gOut + (int64_t)n * d.go_ns + id.p, d.P, d.C);
    if (TWO) sh.issue(hO + (int64_t)n * d.ho_ns + id.p, d.P, d.C);
    Window<C> w;
    w.init(sl_.win);
    w.noflush = (dbg & 2) != 0;
    Scatter<C, TWO> sc;
    Geo g;
    make_geo<KERNEL, 2>(g, xy, offset[n], d, f, id.live);
    sa.to_row(G + lane * C, d.C);
    if (TWO) sh.to_row(H + lane * C, d.C);
    {
        float Dm[4], Em[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            Dm[a] = g.first(a, 0) * cg.x + g.first(a, 1) * cg.y;
            Em[a] = g.pure2(a, 0) * (hg.x * cg.x) + g.pure2(a, 1) * (hg.y * cg.y);   // 2d.cu:876
            if (f.exact) Em[a] = fmaf(g.mixed2(a), hg.x * cg.y + hg.y * cg.x, Em[a]);
            co[a * 64 + lane] = Em[a];
        }
        put_block(KA + lane * 12, Em, g.sx, g.sy, g.akey != KEY_NONE);
        if (TWO) put_block(KB + lane * 12, Dm, g.sx, g.sy, g.akey != KEY_NONE);
    }
    put_nodes(rec, lane, g);
    wave_sync();
    float4 vv[CQ][4];
#pragma unroll
    for (int sub = 0; sub < CQ; ++sub) {
        if (dbg & 4) vv[sub][0] = vv[sub][1] = vv[sub][2] = vv[sub][3] = zero4();
        else gather_raw<CQ>(tab, rec, sub * (64 / CQ) + lane / CQ, q, vv[sub]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!(dbg & 1)) sc.batch(G, H, KA, KB, g.akey, w, acc_n, d);
#pragma unroll
    for (int sub = 0; sub < CQ; ++sub) {
        const int sl = sub * (64 / CQ) + lane / CQ;
        float4(&v)[4] = vv[sub];
        mask_rows(rec, sl, v);
        float4 o = fma4(co[sl], v[0], zero4());
        o = fma4(co[64 + sl], v[1], o);
        o = fma4(co[128 + sl], v[2], o);
        o = fma4(co[192 + sl], v[3], o);
        *reinterpret_cast<float4 *>(G + sl * C + 4 * q) = o;   // over the cotangent quad (the scatter is done with it)
    }
    wave_sync();
    if (id.live) {
        const float *row = G + lane * C;
#pragma unroll
        for (int c = 0; c < C; ++c)
            if (c < d.C) store_out(ggo_n + (int64_t)c * d.P + id.p, row[c]);
    }
    w.flush(acc_n, d);
}

}  // namespace coh
}  // namespace cs
