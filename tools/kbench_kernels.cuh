// tools/kbench_kernels.cuh -- EXPERIMENTAL kernels of round 2 that did not make it into the product (kept so that
// tools/kbench.hip reproduces the numbers quoted in DESIGN.md): LDS-DMA node gathers and the "quad-transposed"
// forward kernel built on them.  Faster than the shipped forward only with warm caches (0.43 vs 0.51 ms), equal cold.
#pragma once
#include "../cosinesampler_amd/csrc/cs_tiled.cuh"

namespace cs {
namespace tiled {

// ------------------------------------------------------------------------------------------------
// Node rows travel from the XCD's L2 STRAIGHT INTO LDS (global_load_lds_dwordx4, "LDS-DMA"), not through the VGPRs.
// Counters (profiles/round2_sq_tcp_summary.json: the L2 answers a row request in its unloaded 175 cycles while the
// vector L1 is busy every cycle) and tools/microbench_gather.hip put the bound of scattered 64-byte row gathers in the
// L1's return path into the register file: 2^24 samples x 4 rows take 0.336 ms as dwordx4 loads and 0.235 ms as DMA
// (0.26-0.29 with 128-byte pairs, 16 lanes per row or sc1 loads; nt 0.69).  Lane (sample, quad q) asks for its 16 bytes
// of each of the sample's 4 node rows; the words land at dma[a][lane] and come back with one ds_read_b128 each.
// ------------------------------------------------------------------------------------------------
constexpr int DMA_FLOATS = 4 * 64 * 4;   // per wave and pass: [4 nodes][64 lanes] float4
template <int CQ>
__device__ __forceinline__ void dma_issue(const float4 *tab, const uint32_t (&node)[4], int q, float *dma) {
#pragma unroll
    for (int a = 0; a < 4; ++a)   // a zero-padded node fetches row 0 and is dropped by dma_read
        __builtin_amdgcn_global_load_lds(tab + (size_t)(node[a] == NO_NODE ? 0u : node[a]) * CQ + q, dma + a * 256, 16, 0, 0);
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// The landing zone is read back in inline assembly: for an ordinary LDS load the compiler's wait-count pass puts an
// `s_waitcnt vmcnt(0)` in front -- it cannot tell which DMA is still in flight into WHICH zone -- and that would drain
// the next pass's loads too (seen in the ISA: NBUF = 2 ran like NBUF = 1).  The counted waits are dma_wait_keep's.
__device__ __forceinline__ void dma_read4(const float *zone, const uint32_t (&node)[4], float4 (&v)[4]) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f a, b, c, d;
    const unsigned addr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void *)(zone + (threadIdx.x & 63) * 4);
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\t"
                 "ds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(addr) : "memory");
    v[0] = node[0] == NO_NODE ? zero4() : make_float4(a.x, a.y, a.z, a.w);
    v[1] = node[1] == NO_NODE ? zero4() : make_float4(b.x, b.y, b.z, b.w);
    v[2] = node[2] == NO_NODE ? zero4() : make_float4(c.x, c.y, c.z, c.w);
    v[3] = node[3] == NO_NODE ? zero4() : make_float4(d.x, d.y, d.z, d.w);
}

// ------------------------------------------------------------------------------------------------
// point_forward3: "quad-transposed" sample order.  Phase 1 as point_forward (lane = sample, geometry -> rec).  In
// phase 2 lane (sl, q) of pass `sub` works on sample CQ*sl + sub of its wave, so that after the CQ passes it holds
// channels 4q..4q+3 of CQ CONSECUTIVE points: the results leave straight from the registers, one store of CQ floats
// per channel (a wave instruction writes 256 contiguous bytes in each of CQ planes) -- no result tile in LDS, no
// second barrier.  Node rows arrive by LDS-DMA, NBUF passes in flight.
// ------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void dma_wait_keep() {   // wait until at most N vector-memory operations are outstanding
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int CQ>
__device__ __forceinline__ void store_run(float *p, const float (&v)[CQ], int nlive) {   // CQ consecutive points of one plane
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef float v2f __attribute__((ext_vector_type(2)));
    if (nlive >= CQ && ((uintptr_t)p & (CQ * 4 - 1)) == 0) {
        if (CQ >= 4) {
#pragma unroll
            for (int k = 0; k < CQ / 4; ++k) {
                v4f t = {v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
                __builtin_nontemporal_store(t, reinterpret_cast<v4f *>(p) + k);
            }
        } else if (CQ == 2) {
            v2f t = {v[0], v[1]};
            __builtin_nontemporal_store(t, reinterpret_cast<v2f *>(p));
        } else {
            __builtin_nontemporal_store(v[0], p);
        }
    } else {
#pragma unroll
        for (int k = 0; k < CQ; ++k)
            if (k < nlive) __builtin_nontemporal_store(v[k], p + k);
    }
}
template <int CQ, int NBUF>
constexpr size_t f3_lds() { return (size_t)4 * (REC_FLOATS + NBUF * DMA_FLOATS) * 4; }
template <int KERNEL, int CQ, int NBUF, int ABL = 0>   // ABL: ablation bits for tools/kbench.hip (1 no gathers, 2 no stores, 4 no grid load)
__global__ __launch_bounds__(256) void point_forward3(const float *__restrict__ icl, const float *__restrict__ grid,
                                                      const float *__restrict__ offset, float *__restrict__ out,
                                                      Dims d, Flags f) {
    extern __shared__ float lds[];
    constexpr int C = 4 * CQ;
    float *rec = lds + (threadIdx.x >> 6) * (REC_FLOATS + NBUF * DMA_FLOATS);
    float *dma = rec + REC_FLOATS;
    if (ABL & 4) {   // positions from a hash of the point index: no HBM read in front of the gathers
        uint32_t h = (uint32_t)(blockIdx.x * 256 + threadIdx.x) * 2654435761u + 17u;
        h ^= h >> 16; h *= 0x7feb352dU; h ^= h >> 15; h *= 0x846ca68bU; h ^= h >> 16;
        const int x = (h & 0xffff) % (d.size[0] - 1), y = (h >> 16) % (d.size[1] - 1), ln = threadIdx.x & 63;
        uint32_t *ru0 = reinterpret_cast<uint32_t *>(rec);
        for (int a = 0; a < 4; ++a) ru0[(R_NODE + a) * 64 + ln] = (uint32_t)((y + (a >> 1)) * d.size[0] + x + (a & 1));
        rec[R_WX0 * 64 + ln] = 0.25f; rec[R_WX1 * 64 + ln] = 0.75f; rec[R_WY0 * 64 + ln] = 0.5f; rec[R_WY1 * 64 + ln] = 0.5f;
    } else {
        point_phase1<KERNEL>(rec, grid, offset, d, f, 1);   // 2D forward: align_corners = 1 (2d.cu:307-308)
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, sl = lane / CQ, q = lane % CQ, n = blockIdx.y;
    const float4 *tab = reinterpret_cast<const float4 *>(icl + (int64_t)n * d.vol * C);
    const uint32_t *ru = reinterpret_cast<const uint32_t *>(rec);
    uint32_t node[CQ][4];
#pragma unroll
    for (int sub = 0; sub < CQ; ++sub)
#pragma unroll
        for (int a = 0; a < 4; ++a) node[sub][a] = ru[(R_NODE + a) * 64 + CQ * sl + sub];
    float4 acc[CQ];
#pragma unroll
    for (int sub = 0; sub < NBUF - 1 && sub < CQ; ++sub)
        if (!(ABL & 1)) dma_issue<CQ>(tab, node[sub], q, dma + sub * DMA_FLOATS);
#pragma unroll
    for (int sub = 0; sub < CQ; ++sub) {
        if (sub + NBUF - 1 < CQ && !(ABL & 1)) dma_issue<CQ>(tab, node[sub + NBUF - 1], q, dma + ((sub + NBUF - 1) % NBUF) * DMA_FLOATS);
        if (sub + NBUF - 1 < CQ) dma_wait_keep<4 * (NBUF - 1)>();
        else if (sub + 1 < CQ && NBUF > 1) { if (CQ - 1 - sub == 1) dma_wait_keep<4>(); else if (CQ - 1 - sub == 2) dma_wait_keep<8>(); else dma_wait_keep<12>(); }
        else dma_wait();
        const int s = CQ * sl + sub;
        const float wx0 = rec[R_WX0 * 64 + s], wx1 = rec[R_WX1 * 64 + s], wy0 = rec[R_WY0 * 64 + s], wy1 = rec[R_WY1 * 64 + s];
        const float W[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};
        float4 v[4], r = zero4();
        dma_read4(dma + (sub % NBUF) * DMA_FLOATS, node[sub], v);
#pragma unroll
        for (int a = 0; a < 4; ++a) r = fma4(W[a], v[a], r);
        acc[sub] = r;
    }
    const int64_t p0 = (int64_t)blockIdx.x * 256 + (threadIdx.x & ~63) + CQ * sl;   // first of this lane's CQ points
    if (ABL & 2) {
        float t = 0.f;
        for (int sub = 0; sub < CQ; ++sub) t += acc[sub].x + acc[sub].y + acc[sub].z + acc[sub].w;
        if (t != -12345.f) return;
    }
    if (p0 >= d.P) return;
    const int nlive = (int)min((int64_t)CQ, d.P - p0);
    float *obase = out + (int64_t)n * d.C * d.P + p0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (4 * q + j < d.C) {
            float v[CQ];
#pragma unroll
            for (int sub = 0; sub < CQ; ++sub) v[sub] = j == 0 ? acc[sub].x : j == 1 ? acc[sub].y : j == 2 ? acc[sub].z : acc[sub].w;
            store_run<CQ>(obase + (int64_t)(4 * q + j) * d.P, v, nlive);
        }
    }
}

}  // namespace tiled
}  // namespace cs
