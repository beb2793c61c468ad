// cs_dense3d.cuh -- 3D with small, crowded tables (the reference's test_3d.py: 50 tables of 16^3 cells, C = 4,
// 10^5-10^6 points).  A whole table is a few hundred KiB, so the row atomics of the general 3D path all land on
// the same few thousand L2 lines and serialise (1.3 ms for 5 M samples; 0.15 ms for the gathers next to it).
// Same cure as cs_tiled.cuh's cell_scatter for 2D: the plan bins the samples by CELL (counting sort, the
// scan kernels are cs_tiled.cuh's), the channels-last point kernels leave p-ordered rows
// [cotangent values | node coefficients] instead of adding them, and one WAVE owns one (n, cell) bucket: lanes
// stride through the bucket keeping 8*C sums, a halving exchange leaves every lane with one (node, channel)
// value, which it adds to grad_input -- in the caller's layout, so the channels-last accumulator, its clear and
// its unpack disappear as well.
#pragma once
#include "cs_points_cl.cuh"
#include "cs_tiled.cuh"

namespace cs {
namespace dense3 {

using tiled::Plan;

struct Cell3 {
    int bin;
    bool valid;
};
// cell coordinates u = lo + 1 in [0, size] per axis (lo = -1: only the high node is in range)
__device__ __forceinline__ Cell3 locate3(const float *g, const Dims &d, const Flags &f, float off, const Plan &pl) {
    int u[3];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float mu;
        float i = source_index(g[j], d.size[j], f.pad, f.align, off, f.multicell, mu);
        bool sane = (i > -1073741824.0f) && (i < 1073741824.0f);
        u[j] = sane ? (int)floorf(i) + 1 : -4;
        ok = ok && u[j] >= 0 && u[j] <= d.size[j];
    }
    Cell3 c;
    c.valid = ok;
    c.bin = (u[2] * pl.nty + u[1]) * pl.ntx + u[0];
    return c;
}

// (chunks, N) workgroups: histogram of the cells of one chunk of one n   (Plan: ntx = W+1, nty = H+1, ntiles = cells)
static __global__ __launch_bounds__(256) void plan_count3(const float *__restrict__ grid, const float *__restrict__ offset,
                                                   Plan pl, Dims d, Flags f) {
    extern __shared__ uint32_t hist[];
    const int n = blockIdx.y, chunk = blockIdx.x;
    for (int b = threadIdx.x; b < pl.ntiles; b += 256) hist[b] = 0;
    __syncthreads();
    const float off = offset[n];
    const int64_t p0 = (int64_t)chunk * pl.chunk;
    for (int i = threadIdx.x; i < pl.chunk; i += 256) {
        int64_t p = p0 + i;
        if (p < d.P) {
            Cell3 q = locate3(grid + d.gpt(n, p) * 3, d, f, off, pl);
            if (q.valid) atomicAdd(&hist[q.bin], 1u);
        }
    }
    __syncthreads();
    uint32_t *dst = pl.block_hist + ((int64_t)n * pl.chunks + chunk) * pl.ntiles;
    for (int b = threadIdx.x; b < pl.ntiles; b += 256) dst[b] = hist[b];
}

// (chunks, N): every sample takes a slot of its cell's bucket -- the final order
static __global__ __launch_bounds__(256) void plan_scatter3(const float *__restrict__ grid, const float *__restrict__ offset,
                                                     Plan pl, Dims d, Flags f) {
    extern __shared__ uint32_t cursor[];
    const int n = blockIdx.y, chunk = blockIdx.x;
    const uint32_t *excl = pl.block_hist + ((int64_t)n * pl.chunks + chunk) * pl.ntiles;
    const uint32_t *tb = pl.tile_begin + (int64_t)n * pl.ntiles;
    for (int b = threadIdx.x; b < pl.ntiles; b += 256) cursor[b] = tb[b] + excl[b];
    __syncthreads();
    const float off = offset[n];
    const int64_t p0 = (int64_t)chunk * pl.chunk;
    for (int i = threadIdx.x; i < pl.chunk; i += 256) {
        int64_t p = p0 + i;
        if (p < d.P) {
            int64_t s = (int64_t)n * d.P + p;
            Cell3 q = locate3(grid + d.gpt(n, p) * 3, d, f, off, pl);
            if (q.valid) pl.sorted[atomicAdd(&cursor[q.bin], 1u)] = (uint32_t)s;
        }
    }
}

// one wave per (n, cell) bucket.  Rows: cl::Rec<3, CQ, MODE> without the ids = [g (C) | (h (C)) | coef (8) | (coef2 (8))]
template <int CQ, int MODE>
__global__ __launch_bounds__(256) void cell_scatter3(const float *__restrict__ rows, Plan pl,
                                                     float *__restrict__ grad_input, Dims d) {
    using R = cl::Rec<3, CQ, MODE>;
    constexpr int C = 4 * CQ, NC = 8, NV = NC * C;
    static_assert(NV <= 128, "at most two (node, channel) values per lane");
    const int64_t bucket = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bucket >= (int64_t)d.N * pl.ntiles) return;
    const uint32_t b0 = pl.tile_begin[bucket], b1 = pl.tile_begin[bucket + 1];
    if (b0 == b1) return;
    const int lane = threadIdx.x & 63;
    float v[NV];   // v[a * C + c]
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = 0.f;
    for (uint32_t j = b0 + lane; j < b1; j += 64) {
        const float *row = rows + (int64_t)pl.sorted[j] * R::IDS;
        float g[C], h[C], ka[NC], kb[NC];
#pragma unroll
        for (int q = 0; q < CQ; ++q) {
            const float4 t = *reinterpret_cast<const float4 *>(row + 4 * q);
            g[4 * q] = t.x; g[4 * q + 1] = t.y; g[4 * q + 2] = t.z; g[4 * q + 3] = t.w;
            if (MODE == 2) {
                const float4 u = *reinterpret_cast<const float4 *>(row + C + 4 * q);
                h[4 * q] = u.x; h[4 * q + 1] = u.y; h[4 * q + 2] = u.z; h[4 * q + 3] = u.w;
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float4 t = *reinterpret_cast<const float4 *>(row + R::COEF + 4 * q);
            ka[4 * q] = t.x; ka[4 * q + 1] = t.y; ka[4 * q + 2] = t.z; ka[4 * q + 3] = t.w;
            if (MODE == 2) {
                const float4 u = *reinterpret_cast<const float4 *>(row + R::COEF + NC + 4 * q);
                kb[4 * q] = u.x; kb[4 * q + 1] = u.y; kb[4 * q + 2] = u.z; kb[4 * q + 3] = u.w;
            }
        }
#pragma unroll
        for (int a = 0; a < NC; ++a)
#pragma unroll
            for (int c = 0; c < C; ++c) {
                float t = fmaf(ka[a], g[c], v[a * C + c]);
                if (MODE == 2) t = fmaf(kb[a], h[c], t);
                v[a * C + c] = t;
            }
    }
    // halving exchange (as tiled::cell_scatter): six steps leave NV / 64 fully reduced values per lane (C = 16: values
    // 2 l and 2 l + 1 on lane l); with NV < 64 the last steps are plain butterflies and 64 / NV lanes share a value
    constexpr int VALS = NV >= 64 ? NV / 64 : 1, SHARE = NV >= 64 ? 1 : 64 / NV;
    int cur = NV;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        if (cur > VALS) {
            const int half = cur / 2;
            const bool up = (lane & m) != 0;
#pragma unroll
            for (int i = 0; i < half; ++i) {
                const float lo = v[i], hi = v[i + half];
                const float got = __shfl_xor(up ? lo : hi, m);
                v[i] = (up ? hi : lo) + got;
            }
            cur = half;
        } else {
            v[0] += __shfl_xor(v[0], m);
        }
    }
    if (lane % SHARE) return;
    const int n = (int)(bucket / pl.ntiles);
    int cell = (int)(bucket - (int64_t)n * pl.ntiles);
    const int ux = cell % pl.ntx;
    cell /= pl.ntx;
    const int uy = cell % pl.nty, uz = cell / pl.nty;
#pragma unroll
    for (int jv = 0; jv < VALS; ++jv) {
        const int idx = (lane / SHARE) * VALS + jv, a = idx / C, c = idx % C;
        const int x = ux - 1 + (a & 1), y = uy - 1 + ((a >> 1) & 1), z = uz - 1 + (a >> 2);
        const float r = v[jv];
        if (c >= d.C || x < 0 || x >= d.size[0] || y < 0 || y >= d.size[1] || z < 0 || z >= d.size[2] || r == 0.f) continue;
        unsafeAtomicAdd(grad_input + ((int64_t)n * d.C + c) * d.vol + ((int64_t)z * d.size[1] + y) * d.size[0] + x, r);
    }
}

}  // namespace dense3

// ------------------------------------------------------------------------------------------------------------------
// tiles3 -- large, sparsely hit 3D tables (BASELINE configs[3]: 128^3 nodes, 0.25 samples per cell) WITHOUT atomics.
// The fused row atomics of cs_points_cl.cuh sit on a floor that ordering cannot move: 2^24 pair-row atomics take
// 0.83 ms wherever they land, and LDS float atomics are slower still (profiles/round2_microbench_atomics.txt).  So:
//   plan    grad_input is cut into tiles of 16x4x4 NODES, each owned by one wave.  A sample is listed in every tile that
//           owns one of its 8 corner nodes (1.66 tiles on average), with a 9-bit code = where its low corner sits
//           relative to that tile (-1..15, -1..3, -1..3); the 2D plan's scan kernels, no order inside a bucket
//   points  the channels-last point kernels leave p-ordered rows [cotangents | coefficients] (SCATTER == 2, as for the
//           crowded tables above)
//   tile3_scatter   a wave sums its tile in an LDS image, list entries taken 64 at a time.  Entries with DIFFERENT
//           codes never meet at a node in the same round (a round = one corner); entries with the same code are
//           ranked inside the batch and rank r is added in pass r -- plain LDS read-modify-writes,
//           wave-synchronous, no atomics -- and corners owned by another tile are skipped (that tile lists the sample
//           too).  The tile then leaves for grad_input in the caller's NCDHW layout with plain 64-byte row stores:
//           every node of every tile exactly once, so no accumulator, no clear, no unpack.
// (A first version with cell tiles, one list entry per sample and 9x9x5-node images summed by a second kernel moved
// 3 GB per stage through those images and took 0.72 ms for the two kernels.)
// ------------------------------------------------------------------------------------------------------------------
namespace tiles3 {

using tiled::Plan;
constexpr int M3X = 16, M3Y = 4, M3Z = 4, OWNED = M3X * M3Y * M3Z;        // nodes per tile
constexpr int K3X = M3X + 1, K3Y = M3Y + 1, K3Z = M3Z + 1;               // code range per axis: low corner slot + 1
constexpr int BINS = 512;                                                // codes take 9 bits: K3X * K3Y * K3Z = 425
static_assert(K3X * K3Y * K3Z <= BINS, "a code fits 9 bits");

// the low corner of a sample per axis, lo in [-1, size-1]; valid = it touches at least one node of the table
struct Low3 {
    int lo[3];
    bool valid;
};
__device__ __forceinline__ Low3 locate3m(const float *g, const Dims &d, const Flags &f, float off) {
    Low3 q;
    q.valid = true;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float mu;
        float i = source_index(g[j], d.size[j], f.pad, f.align, off, f.multicell, mu);
        bool sane = (i > -1073741824.0f) && (i < 1073741824.0f);
        q.lo[j] = sane ? (int)floorf(i) : -4;
        q.valid = q.valid && q.lo[j] >= -1 && q.lo[j] <= d.size[j] - 1;
    }
    return q;
}
// calls emit(tile, code) for every tile that owns a corner node of the sample
template <typename F>
__device__ __forceinline__ void for_each_tile(const Low3 &q, const Dims &d, const Plan &pl, F emit) {
    int t[3][2], cnt[3];
    constexpr int M[3] = {M3X, M3Y, M3Z};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int a = q.lo[j] >= 0 ? q.lo[j] / M[j] : -1;                          // tile of node lo (if it exists)
        const int b = q.lo[j] + 1 <= d.size[j] - 1 ? (q.lo[j] + 1) / M[j] : -1;    // tile of node lo + 1
        cnt[j] = 0;
        if (a >= 0) t[j][cnt[j]++] = a;
        if (b >= 0 && b != a) t[j][cnt[j]++] = b;
    }
    for (int kz = 0; kz < cnt[2]; ++kz)
        for (int ky = 0; ky < cnt[1]; ++ky)
            for (int kx = 0; kx < cnt[0]; ++kx) {
                const int tx = t[0][kx], ty = t[1][ky], tz = t[2][kz];
                const int cx = q.lo[0] - tx * M3X + 1, cy = q.lo[1] - ty * M3Y + 1, cz = q.lo[2] - tz * M3Z + 1;
                emit((tz * pl.nty + ty) * pl.ntx + tx, (cz * K3Y + cy) * K3X + cx);
            }
}

// (chunks, N) workgroups: histogram of the tile lists of one chunk of one n
static __global__ __launch_bounds__(256) void plan_count3t(const float *__restrict__ grid, const float *__restrict__ offset,
                                                    Plan pl, Dims d, Flags f) {
    extern __shared__ uint32_t hist[];
    const int n = blockIdx.y, chunk = blockIdx.x;
    for (int b = threadIdx.x; b < pl.ntiles; b += 256) hist[b] = 0;
    __syncthreads();
    const float off = offset[n];
    const int64_t p0 = (int64_t)chunk * pl.chunk;
    for (int i = threadIdx.x; i < pl.chunk; i += 256) {
        int64_t p = p0 + i;
        if (p < d.P) {
            Low3 q = locate3m(grid + d.gpt(n, p) * 3, d, f, off);
            if (q.valid) for_each_tile(q, d, pl, [&](int tile, int) { atomicAdd(&hist[tile], 1u); });
        }
    }
    __syncthreads();
    uint32_t *dst = pl.block_hist + ((int64_t)n * pl.chunks + chunk) * pl.ntiles;
    for (int b = threadIdx.x; b < pl.ntiles; b += 256) dst[b] = hist[b];
}
// (chunks, N): every list entry takes a slot of its tile's bucket (any order); key = (p << 9) | code
static __global__ __launch_bounds__(256) void plan_scatter3t(const float *__restrict__ grid, const float *__restrict__ offset,
                                                      Plan pl, Dims d, Flags f) {
    extern __shared__ uint32_t cursor[];
    const int n = blockIdx.y, chunk = blockIdx.x;
    const uint32_t *excl = pl.block_hist + ((int64_t)n * pl.chunks + chunk) * pl.ntiles;
    const uint32_t *tb = pl.tile_begin + (int64_t)n * pl.ntiles;
    for (int b = threadIdx.x; b < pl.ntiles; b += 256) cursor[b] = tb[b] + excl[b];
    __syncthreads();
    const float off = offset[n];
    const int64_t p0 = (int64_t)chunk * pl.chunk;
    for (int i = threadIdx.x; i < pl.chunk; i += 256) {
        int64_t p = p0 + i;
        if (p < d.P) {
            Low3 q = locate3m(grid + d.gpt(n, p) * 3, d, f, off);
            if (q.valid)
                for_each_tile(q, d, pl, [&](int tile, int code) {
                    pl.key[atomicAdd(&cursor[tile], 1u)] = ((uint32_t)p << 9) | (uint32_t)code;
                });
        }
    }
}
// One wave per (n, tile) bucket; blockDim = 64 * waves, dynamic LDS = waves * OWNED * CQ float4 (4 / 8 / 16 KiB per wave:
// the CU fills with waves, which is what hides the chain  bucket bounds -> ids -> rows -> LDS -> output  of a tile; a
// version that took four tiles per wave and fetched the next tile's first batch ahead was no faster).
template <int CQ, int MODE>
struct Batch3 {   // one lane's share of a batch: a sample's row
    using R = cl::Rec<3, CQ, MODE>;
    float4 g[CQ], h[CQ], ka[2], kb[2];
    __device__ __forceinline__ void load(const float *rows, uint32_t id) {
        const float *row = rows + (int64_t)id * R::IDS;
#pragma unroll
        for (int q = 0; q < CQ; ++q) {
            g[q] = *reinterpret_cast<const float4 *>(row + 4 * q);
            if (MODE == 2) h[q] = *reinterpret_cast<const float4 *>(row + 4 * CQ + 4 * q);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            ka[q] = *reinterpret_cast<const float4 *>(row + R::COEF + 4 * q);
            if (MODE == 2) kb[q] = *reinterpret_cast<const float4 *>(row + R::COEF + 8 + 4 * q);
        }
    }
};
constexpr int CNT_WORDS = 128;     // per wave: one byte counter per code (BINS / 4 words), kept zero between batches
// add one batch (lane = list entry) to the wave's image [z][y][x][CQ]
template <int CQ, int MODE>
__device__ __forceinline__ void sum_batch3(float4 *img, uint32_t *cnt, const Batch3<CQ, MODE> &b, int code, bool live) {
    // rank of this entry among the entries of the batch with the same code (the bucket is in no particular order): a
    // returning LDS add on the code's byte counter hands every entry of a code a different rank -- one instruction where
    // the first version read and compared all 63 other lanes' codes (~190 instructions per batch; the kernel is bound by
    // the instructions its short waves issue: profiles/round4_ablation.txt).  The entries put their counters back to zero.
    int rank = 0;
    const int sh = 8 * (code & 3);
    if (live) rank = (int)((atomicAdd(&cnt[code >> 2], 1u << sh) >> sh) & 0xFFu);
    int maxrank = rank;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) maxrank = max(maxrank, __shfl_xor(maxrank, m));
    const int cx = code % K3X, cyz = code / K3X, cy = cyz % K3Y, cz = cyz / K3Y;
    const int sx = cx - 1, sy = cy - 1, sz = cz - 1;             // slot of the low corner inside the tile, -1 .. M-1
    const float ca[8] = {b.ka[0].x, b.ka[0].y, b.ka[0].z, b.ka[0].w, b.ka[1].x, b.ka[1].y, b.ka[1].z, b.ka[1].w};
    float cb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (MODE == 2) { cb[0] = b.kb[0].x; cb[1] = b.kb[0].y; cb[2] = b.kb[0].z; cb[3] = b.kb[0].w; cb[4] = b.kb[1].x; cb[5] = b.kb[1].y; cb[6] = b.kb[1].z; cb[7] = b.kb[1].w; }
    for (int r = 0; r <= maxrank; ++r) {
        if (live && rank == r) {
#pragma unroll
            for (int a = 0; a < 8; ++a) {      // a round = one corner: different codes -> different nodes
                const int x = sx + (a & 1), y = sy + ((a >> 1) & 1), z = sz + (a >> 2);
                if (x >= 0 && x < M3X && y >= 0 && y < M3Y && z >= 0 && z < M3Z) {   // else: another tile's node
                    float4 *node = img + ((z * M3Y + y) * M3X + x) * CQ;
#pragma unroll
                    for (int q = 0; q < CQ; ++q) {
                        float4 t = cl::fma4(ca[a], b.g[q], node[q]);
                        if (MODE == 2) t = cl::fma4(cb[a], b.h[q], t);
                        node[q] = t;
                    }
                }
                // The next round's nodes are other LANES' nodes of this round (code c's high-x corner is code c+1's
                // low-x one).  One thread's own addresses differ by constants, so without this the compiler batches
                // the eight reads ahead of the eight writes; the hardware needs nothing (LDS runs a wave in order).
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // ranks of one code follow each other
    }
    if (live) cnt[code >> 2] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}
template <int CQ, int MODE>
__global__ __launch_bounds__(256) void tile3_scatter(const float *__restrict__ rows, Plan pl, float *__restrict__ out,
                                                     Dims d, int waves) {
    extern __shared__ float4 img_all[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t bucket = (int64_t)blockIdx.x * waves + wv;
    if (bucket >= (int64_t)d.N * pl.ntiles) return;
    float4 *img = img_all + wv * (OWNED * CQ);     // this wave's image: LDS operations of ONE wave execute in order,
                                                   // so no barrier is needed anywhere below
    uint32_t *cnt = reinterpret_cast<uint32_t *>(img_all + waves * (OWNED * CQ)) + wv * CNT_WORDS;   // the wave's code counters
    for (int i = lane; i < CNT_WORDS; i += 64) cnt[i] = 0u;
    const uint32_t b0 = pl.tile_begin[bucket], b1 = pl.tile_begin[bucket + 1];
    const uint32_t sbase = (uint32_t)(bucket / pl.ntiles) * (uint32_t)d.P;          // n * P
    for (int i = lane; i < OWNED * CQ; i += 64) img[i] = cl::zero4();
    for (uint32_t j0 = b0; j0 < b1; j0 += 128) {   // two entries per lane: a tile lists ~100 at config 3
        const bool liveA = j0 + lane < b1, liveB = j0 + 64 + lane < b1;
        const uint32_t jA = liveA ? j0 + lane : b1 - 1, jB = liveB ? j0 + 64 + lane : b1 - 1;
        const uint32_t kA = pl.key[jA], kB = pl.key[jB];                            // (p << 9) | code
        const int codeA = (int)(kA & (BINS - 1)), codeB = (int)(kB & (BINS - 1));
        Batch3<CQ, MODE> A, B;
        A.load(rows, sbase + (kA >> 9));
        B.load(rows, sbase + (kB >> 9));
        sum_batch3<CQ, MODE>(img, cnt, A, codeA, liveA);
        if (j0 + 64 < b1) sum_batch3<CQ, MODE>(img, cnt, B, codeB, liveB);
    }
    // the tile leaves: rows of 16 nodes along x of one (channel, z, y), 64 contiguous bytes each
    const int W = d.size[0], H = d.size[1], D = d.size[2];
    const int n = (int)(bucket / pl.ntiles);
    int tl = (int)(bucket - (int64_t)n * pl.ntiles);
    const int tx = tl % pl.ntx;
    tl /= pl.ntx;
    const int ty = tl % pl.nty, tz = tl / pl.nty;
    const int x0 = tx * M3X, y0 = ty * M3Y, z0 = tz * M3Z;
    float *o = out + (int64_t)n * d.C * d.vol;
    const bool vec = (W & 3) == 0;
    // one item = the four channels of a quad at four nodes along x: four 16-byte LDS reads, four 16-byte stores (one per
    // channel plane) -- the first version took one channel per item with four scalar LDS reads each: four times the loop
    // iterations and index arithmetic for the same stores
    for (int i = lane; i < CQ * M3Z * M3Y * (M3X / 4); i += 64) {
        const int x4 = i % (M3X / 4), r = i / (M3X / 4), y = r % M3Y, zq = r / M3Y, z = zq % M3Z, q = zq / M3Z;
        const int gx = x0 + 4 * x4, gy = y0 + y, gz = z0 + z;
        if (gy >= H || gz >= D || gx >= W) continue;
        const float4 *src = img + ((z * M3Y + y) * M3X + 4 * x4) * CQ + q;
        const float4 n0 = src[0], n1 = src[CQ], n2 = src[2 * CQ], n3 = src[3 * CQ];
        const float4 ch[4] = {make_float4(n0.x, n1.x, n2.x, n3.x), make_float4(n0.y, n1.y, n2.y, n3.y),
                              make_float4(n0.z, n1.z, n2.z, n3.z), make_float4(n0.w, n1.w, n2.w, n3.w)};
        float *dst0 = o + (int64_t)(4 * q) * d.vol + ((int64_t)gz * H + gy) * W + gx;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (4 * q + k >= d.C) break;
            float *dst = dst0 + (int64_t)k * d.vol;
            const float4 v = ch[k];
            if (vec) {
                *reinterpret_cast<float4 *>(dst) = v;                     // W % 4 == 0: whole quads are in range
            } else {
                dst[0] = v.x;
                if (gx + 1 < W) dst[1] = v.y;
                if (gx + 2 < W) dst[2] = v.z;
                if (gx + 3 < W) dst[3] = v.w;
            }
        }
    }
}

}  // namespace tiles3
}  // namespace cs
