// tools/microbench_l2.hip -- which resource do the sampler's kernels run out of?  (round 2; not product code)
//
// The round-1 "additive law" (L2-hit gathers and HBM streams never overlap) was a wall-clock observation.  This file
// holds the same access patterns as separate, individually named kernels so that one `rocprofv3 --pmc` pass per counter
// group (tools/l2_pmc.sh) can say what each of them does to the L2 (TCC), the vector L1 (TCP/TA) and the fabric (EA):
//   gather64      2^24 samples x 4 random 64-B node rows of a 4 MiB channels-last table (one table at a time)  [L2 hits]
//   read4/read16  1 GiB streamed in, 4 B or 16 B per lane, nontemporal                                         [HBM -> L2 -> CU]
//   write4/write16 1 GiB streamed out, sc1 dword stores (as the product's outputs) / nontemporal dwordx4       [CU -> L2 -> HBM]
//   copy16        1 GiB in + 1 GiB out
//   mixed         gather64 and copy16 in ONE launch, workgroups alternate roles
//   scat64/scat8  2^24 rows of 64 B / 8 B written to random slots of a 1 GiB / 128 MiB array  (p-order -> cell order)
//   fetch64       2^24 random 64-B row reads from a 1 GiB array                               (cell order <- p-order)
//   hist          2^24 LDS integer atomics on a 66k-bin u16-pair histogram (the round-2 plan's count pass)
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench_l2 tools/microbench_l2.hip && tools/microbench_l2
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x)                                                                        \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

__global__ __launch_bounds__(256) void gather64(const float4 *table, int64_t nodes_per_n, int W, int64_t P, float *out) {
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t p = t >> 2;
    int q = (int)(t & 3), n = blockIdx.y;
    if (p >= P) return;
    uint32_t h = hash32((uint32_t)p * 2654435761u + 17u);
    float fx = (h & 0xffff) * (1.0f / 65536.0f) * (W - 2), fy = (h >> 16) * (1.0f / 65536.0f) * (W - 2);
    float off = n * (1.0f / 16.0f);
    int x = (int)(fx + off), y = (int)(fy + off);
    const float4 *base = table + ((int64_t)n * nodes_per_n + (int64_t)y * W + x) * 4 + q;
    float4 a = base[0], b = base[4], c = base[(int64_t)W * 4], d = base[(int64_t)W * 4 + 4];
    float r = a.x + b.y + c.z + d.w;
    if (r == -12345.f) out[0] = r;
}

__global__ __launch_bounds__(256) void read4(const float *__restrict__ src, int64_t n, float *out) {
    // as the product's stream loads: lane = point, 16 channel planes one after the other (256 B per wave instruction)
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t plane = n / 16;
    float r = 0.f;
    if (i < plane) {
#pragma unroll
        for (int c = 0; c < 16; ++c) r += __builtin_nontemporal_load(src + (int64_t)c * plane + i);
    }
    if (r == -12345.f) out[0] = r;
}
// 16-bit planes (0.5 GiB): how should lanes share them?  n = elements in all 16 planes, block = 256 samples unless noted
__global__ __launch_bounds__(256) void read2(const uint16_t *__restrict__ src, int64_t n, float *out) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // one 2-byte load per lane and plane (128 B per wave instruction)
    const int64_t plane = n / 16;
    uint32_t r = 0;
#pragma unroll
    for (int c = 0; c < 16; ++c) r += __builtin_nontemporal_load(src + (int64_t)c * plane + i);
    if (r == 0x12345u) out[0] = r;
}
__global__ __launch_bounds__(256) void read2_halves(const uint16_t *__restrict__ src, int64_t n, float *out) {
    // lanes 0..31: sample pairs of plane c, lanes 32..63: of plane c+1 (two 128-B runs per instruction, 8 loads per lane)
    const int lane = threadIdx.x & 63, L = lane & 31, up = lane >> 5;
    const int64_t p0 = (int64_t)blockIdx.x * 256 + (threadIdx.x & ~63), plane = n / 16;
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        r += __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(src + (int64_t)(2 * i + up) * plane + p0 + 2 * L));
    if (r == 0x12345u) out[0] = r;
}
__global__ __launch_bounds__(256) void read2_rows(const uint16_t *__restrict__ src, int64_t n, float *out) {
    // wave w takes planes w, w+4, w+8, w+12 for the block's 256 samples: 8 B per lane, 512 contiguous bytes per instruction
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t p0 = (int64_t)blockIdx.x * 256, plane = n / 16;
    uint32_t r = 0;
    typedef uint32_t v2u __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v2u t = __builtin_nontemporal_load(reinterpret_cast<const v2u *>(src + (int64_t)(w + 4 * i) * plane + p0 + 4 * lane));
        r += t.x + t.y;
    }
    if (r == 0x12345u) out[0] = r;
}
__global__ __launch_bounds__(256) void read2_wide(const uint16_t *__restrict__ src, int64_t n, float *out) {
    // block = 512 samples, lane = sample pair: one dword per lane and plane (256 B per wave instruction, 16 loads per lane)
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t plane = n / 16;
    uint32_t r = 0;
#pragma unroll
    for (int c = 0; c < 16; ++c) r += __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(src + (int64_t)c * plane) + i);
    if (r == 0x12345u) out[0] = r;
}
__global__ __launch_bounds__(256) void read4_half(const float *__restrict__ src, int64_t n, float *out) {
    // control: fp32 planes, 8 of them (the same 0.5 GiB, the product's fp32 pattern)
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t plane = n / 8;
    float r = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) r += __builtin_nontemporal_load(src + (int64_t)c * plane + i);
    if (r == -12345.f) out[0] = r;
}
__global__ __launch_bounds__(256) void read16(const v4f *__restrict__ src, int64_t n4, float *out) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float r = 0.f;
    if (i < n4) { v4f t = __builtin_nontemporal_load(src + i); r = t.x + t.w; }
    if (r == -12345.f) out[0] = r;
}
__global__ __launch_bounds__(256) void write4(float *__restrict__ dst, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t plane = n / 16;
    if (i < plane) {
#pragma unroll
        for (int c = 0; c < 16; ++c) __hip_atomic_store(dst + (int64_t)c * plane + i, (float)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__global__ __launch_bounds__(256) void write4_plain(float *__restrict__ dst, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t plane = n / 16;
    if (i < plane) {
#pragma unroll
        for (int c = 0; c < 16; ++c) dst[(int64_t)c * plane + i] = (float)c;
    }
}
__global__ __launch_bounds__(256) void write16(v4f *__restrict__ dst, int64_t n4) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) { v4f t = {1.f, 2.f, 3.f, (float)i}; __builtin_nontemporal_store(t, dst + i); }
}

// ---- round 3: what the coherent kernels' stream pattern can reach -------------------------------------------------
__global__ __launch_bounds__(256) void write4_nt(float *__restrict__ dst, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t plane = n / 16;
    if (i < plane) {
#pragma unroll
        for (int c = 0; c < 16; ++c) __builtin_nontemporal_store((float)c, dst + (int64_t)c * plane + i);
    }
}
// the same 16 planes, but a wave writes its 64 points as 4 instructions of 16 B per lane: 16 lanes per plane row
__global__ __launch_bounds__(256) void write16_rows(float *__restrict__ dst, int64_t n) {
    const int64_t plane = n / 16;
    const int lane = threadIdx.x & 63;
    const int64_t p0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (p0 < plane) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (lane >> 4) + 4 * i;
            v4f t = {1.f, 2.f, 3.f, (float)row};
            __builtin_nontemporal_store(t, reinterpret_cast<v4f *>(dst + (int64_t)row * plane + p0 + 4 * (lane & 15)));
        }
    }
}
// a wave walks `chunk` consecutive points 64 at a time, DEPTH register sets of 16 plane loads in flight, one 8-byte
// store per point: the coherent first backward without any of its work.  Launched with one wave per workgroup and `lds`
// bytes of dynamic LDS to set the number of waves per CU.
template <int DEPTH, bool WRITE16>
__global__ __launch_bounds__(64) void stream_walk(const float *__restrict__ src, int64_t plane, int chunk, float *__restrict__ out,
                                                  float *__restrict__ out16) {
    extern __shared__ float lds_[];
    const int lane = threadIdx.x;
    const int64_t p0 = (int64_t)blockIdx.x * chunk;
    if (p0 >= plane) return;
    float v[DEPTH][16];
    auto issue = [&](float (&s)[16], int b) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < 16; ++c) s[c] = __builtin_nontemporal_load(src + (int64_t)c * plane + p0 + b + lane);
    };
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) issue(v[i], 64 * i);
    for (int b0 = 0; b0 < chunk; b0 += 64 * DEPTH) {
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) {
            const int b = b0 + 64 * i;
            float r = 0.f, q = 0.f;
#pragma unroll
            for (int c = 0; c < 16; ++c) { r += v[i][c]; q = fmaf(v[i][c], r, q); }
            if (WRITE16) {
#pragma unroll
                for (int c = 0; c < 16; ++c) __builtin_nontemporal_store(v[i][c] + r, out16 + (int64_t)c * plane + p0 + b + lane);
            }
            if (b + 64 * DEPTH < chunk) issue(v[i], b + 64 * DEPTH);
            if (lds_[lane] == -1.f) r = 0.f;
            reinterpret_cast<float2 *>(out)[p0 + b + lane] = make_float2(r, q);
        }
    }
}
__global__ __launch_bounds__(256) void copy16(const v4f *__restrict__ src, v4f *__restrict__ dst, int64_t n4) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) { v4f t = __builtin_nontemporal_load(src + i); __builtin_nontemporal_store(t, dst + i); }
}
__global__ __launch_bounds__(256) void mixed(const float4 *table, int64_t nodes_per_n, int W, int64_t P, float *out,
                                             const v4f *__restrict__ src, v4f *__restrict__ dst, int64_t n4,
                                             int64_t gather_blocks_per_n) {
    const int64_t b = blockIdx.x >> 1;
    if (blockIdx.x & 1) {
        int64_t i = b * 256 + threadIdx.x;
        if (i < n4) { v4f t = __builtin_nontemporal_load(src + i); __builtin_nontemporal_store(t, dst + i); }
    } else {
        int n = (int)(b / gather_blocks_per_n);
        if (n >= 16) return;
        int64_t t = (b - (int64_t)n * gather_blocks_per_n) * 256 + threadIdx.x;
        int64_t p = t >> 2;
        int q = (int)(t & 3);
        if (p >= P) return;
        uint32_t h = hash32((uint32_t)p * 2654435761u + 17u);
        float fx = (h & 0xffff) * (1.0f / 65536.0f) * (W - 2), fy = (h >> 16) * (1.0f / 65536.0f) * (W - 2);
        float off = n * (1.0f / 16.0f);
        int x = (int)(fx + off), y = (int)(fy + off);
        const float4 *base = table + ((int64_t)n * nodes_per_n + (int64_t)y * W + x) * 4 + q;
        float4 a = base[0], bb = base[4], c = base[(int64_t)W * 4], d = base[(int64_t)W * 4 + 4];
        float r = a.x + bb.y + c.z + d.w;
        if (r == -12345.f) out[0] = r;
    }
}
// gathers + reads of the SAME launch in the SAME wave (what a point kernel does): 16 dword stream loads + 4 row gathers
__global__ __launch_bounds__(256) void gather_read(const float4 *table, int64_t nodes_per_n, int W, int64_t P, float *out,
                                                   const float *__restrict__ src) {
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;   // 4 lanes per sample for the gathers
    int64_t p = t >> 2;
    int q = (int)(t & 3), n = blockIdx.y;
    if (p >= P) return;
    uint32_t h = hash32((uint32_t)p * 2654435761u + 17u);
    float fx = (h & 0xffff) * (1.0f / 65536.0f) * (W - 2), fy = (h >> 16) * (1.0f / 65536.0f) * (W - 2);
    float off = n * (1.0f / 16.0f);
    int x = (int)(fx + off), y = (int)(fy + off);
    const float4 *base = table + ((int64_t)n * nodes_per_n + (int64_t)y * W + x) * 4 + q;
    float4 a = base[0], b = base[4], c = base[(int64_t)W * 4], d = base[(int64_t)W * 4 + 4];
    // this lane's share of the 16 x 4 B stream words of its wave's 16 samples: 4 planes x 1 dword, coalesced over the wave
    const int lane = threadIdx.x & 63;
    const int64_t p0 = ((int64_t)blockIdx.x * 256 + (threadIdx.x & ~63)) >> 2;     // first sample of this wave
    float r = a.x + b.y + c.z + d.w;
    const float *s = src + (int64_t)n * 16 * P;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int cpl = k * 4 + (lane >> 4);                                               // plane
        r += __builtin_nontemporal_load(s + (int64_t)cpl * P + p0 + (lane & 15));
    }
    if (r == -12345.f) out[0] = r;
}

__global__ void make_perm(uint32_t *perm, int64_t n, uint32_t mask) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t x = (uint32_t)i;
    x = (x * 2654435761u) & mask; x ^= x >> 7; x = (x * 0x9E3779B1u) & mask; x ^= x >> 11; x = (x * 0x85EBCA6Bu | 1u) & mask;
    perm[i] = x & mask;
}
__global__ __launch_bounds__(256) void scat64(float4 *dst, const uint32_t *perm, int64_t rows) {
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t r = t >> 2;
    int q = t & 3;
    if (r >= rows) return;
    dst[(int64_t)perm[r] * 4 + q] = make_float4((float)r, 1.f, 2.f, (float)q);
}
__global__ __launch_bounds__(256) void scat64_nt(v4f *dst, const uint32_t *perm, int64_t rows) {
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t r = t >> 2;
    int q = t & 3;
    if (r >= rows) return;
    v4f v = {(float)r, 1.f, 2.f, (float)q};
    __builtin_nontemporal_store(v, dst + (int64_t)perm[r] * 4 + q);
}
__global__ __launch_bounds__(256) void scat8(float2 *dst, const uint32_t *perm, int64_t rows) {
    int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    dst[perm[r]] = make_float2((float)r, 1.f);
}
__global__ __launch_bounds__(256) void scat16(float4 *dst, const uint32_t *perm, int64_t rows) {
    int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    dst[perm[r]] = make_float4((float)r, 1.f, 2.f, 3.f);
}
__global__ __launch_bounds__(256) void fetch64(const float4 *src, const uint32_t *perm, int64_t rows, float *out) {
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t r = t >> 2;
    int q = t & 3;
    if (r >= rows) return;
    float4 v = src[(int64_t)perm[r] * 4 + q];
    if (v.x + v.y + v.z + v.w == -12345.f) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void fetch8(const float2 *src, const uint32_t *perm, int64_t rows, float *out) {
    int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    float2 v = src[perm[r]];
    if (v.x + v.y == -12345.f) out[0] = 1.f;
}
// sequential rows read with the walkers' access shape: 4 lanes = one walker, each walker owns a run of `per` rows
__global__ __launch_bounds__(256) void walk64(const float4 *src, int64_t rows, int per, float *out) {
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int q = (int)(t & 3);
    int64_t w = t >> 2;
    int64_t j0 = w * per;
    if (j0 >= rows) return;
    float4 acc = {};
    for (int i = 0; i < per; i += 8) {
        float4 g[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) g[u] = src[(j0 + i + u) * 4 + q];
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc.x += g[u].x; acc.y += g[u].y; acc.z += g[u].z; acc.w += g[u].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == -12345.f) out[0] = 1.f;
}

// the round-2 plan's count pass: one workgroup = one chunk of one table's points, a u16-pair histogram of every cell
// in LDS (66 049 bins of a 256x256 table = 132 KiB), returning atomics give the sample its rank inside (chunk, cell)
__global__ __launch_bounds__(1024) void hist(const float2 *__restrict__ grid, uint32_t *__restrict__ rank, int W, int chunk,
                                             int64_t P, uint32_t *dump) {
    extern __shared__ uint32_t h[];
    const int bins = (W + 1) * (W + 1), words = (bins + 1) / 2;
    for (int i = threadIdx.x; i < words; i += 1024) h[i] = 0;
    __syncthreads();
    const int n = blockIdx.y;
    const int64_t p0 = (int64_t)blockIdx.x * chunk;
    const float off = n * (1.0f / 16.0f);
    for (int i = threadIdx.x; i < chunk; i += 1024) {
        int64_t p = p0 + i;
        if (p < P) {
            float2 g = grid[(int64_t)n * P + p];
            int ux = (int)floorf((g.x + 1.f) * 0.5f * (W - 2) + off) + 1, uy = (int)floorf((g.y + 1.f) * 0.5f * (W - 2) + off) + 1;
            int b = uy * (W + 1) + ux;
            uint32_t old = atomicAdd(&h[b >> 1], (b & 1) ? 0x10000u : 1u);
            rank[(int64_t)n * P + p] = (b & 1) ? (old >> 16) : (old & 0xffffu);
        }
    }
    __syncthreads();
    uint32_t *d = dump + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * words;
    for (int i = threadIdx.x; i < words; i += 1024) d[i] = h[i];
}
__global__ void fill_grid(float2 *grid, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t h = hash32((uint32_t)i * 2654435761u + 3u);
    grid[i] = make_float2((h & 0xffff) * (2.0f / 65536.0f) - 1.f, (h >> 16) * (2.0f / 65536.0f) - 1.f);
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }
#define TIME(label, ...)                                                                    \
    do {                                                                                    \
        float best = 1e9f;                                                                  \
        for (int rep = 0; rep < reps; ++rep) {                                              \
            CK(hipEventRecord(e0)); __VA_ARGS__; CK(hipEventRecord(e1));                    \
            CK(hipEventSynchronize(e1));                                                    \
            float ms = time_ms(e0, e1);                                                     \
            if ((rep || reps == 1) && ms < best) best = ms;                                 \
        }                                                                                   \
        CK(hipGetLastError());                                                              \
        printf("%-92s %.3f ms\n", label, best);                                             \
    } while (0)

int main(int argc, char **argv) {
    const int reps = (argc > 1 && !strcmp(argv[1], "once")) ? 1 : 4;   // `once`: one launch per kernel (PMC passes)
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float *dout; CK(hipMalloc(&dout, 64));
    const int W = 256, N = 16;
    const int64_t nodes = (int64_t)W * W, P = 1 << 20, S = P * N;
    float4 *table; CK(hipMalloc(&table, (size_t)N * nodes * 64 + 4096)); CK(hipMemset(table, 0, (size_t)N * nodes * 64));
    const int64_t n4 = (int64_t)1 << 26;   // 1 GiB of float4
    v4f *src, *dst;
    CK(hipMalloc(&src, n4 * 16)); CK(hipMalloc(&dst, n4 * 16)); CK(hipMemset(src, 0, n4 * 16)); CK(hipMemset(dst, 0, n4 * 16));
    uint32_t *perm; CK(hipMalloc(&perm, S * 4));
    make_perm<<<S / 256, 256>>>(perm, S, (uint32_t)(S - 1));
    CK(hipDeviceSynchronize());
    dim3 gg((unsigned)(P * 4 / 256), 16);
    TIME("gather64   2^24 samples x 4 rows of 64 B, table by table (L2 hits)", (gather64<<<gg, 256>>>(table, nodes, W, P, dout)));
    TIME("read4      1 GiB in, 4 B/lane nontemporal (16 planes, as the product streams)", (read4<<<S / 256, 256>>>((const float *)src, S * 16, dout)));
    TIME("read2      0.5 GiB in, 16 planes of 16-bit elements, 2 B/lane", (read2<<<S / 256, 256>>>((const uint16_t *)src, S * 16, dout)));
    TIME("read2h     the same planes, dwords: lanes 0..31 plane c, lanes 32..63 plane c+1", (read2_halves<<<S / 256, 256>>>((const uint16_t *)src, S * 16, dout)));
    TIME("read2r     the same planes, a wave takes whole 512-B block rows (8 B/lane, 4 planes per wave)", (read2_rows<<<S / 256, 256>>>((const uint16_t *)src, S * 16, dout)));
    TIME("read2w     the same planes, 512 samples per block: lane = sample pair, one dword per plane", (read2_wide<<<S / 512, 256>>>((const uint16_t *)src, S * 16, dout)));
    TIME("read4half  control: 0.5 GiB as 8 fp32 planes, 4 B/lane", (read4_half<<<S / 256, 256>>>((const float *)src, S * 8, dout)));
    if (argc > 1 && !strcmp(argv[1], "half")) return 0;
    TIME("read16     1 GiB in, 16 B/lane nontemporal", (read16<<<n4 / 256, 256>>>(src, n4, dout)));
    TIME("write4     1 GiB out, 4 B/lane sc1 stores (as the product outputs)", (write4<<<S / 256, 256>>>((float *)dst, S * 16)));
    TIME("write4p    1 GiB out, 4 B/lane plain stores", (write4_plain<<<S / 256, 256>>>((float *)dst, S * 16)));
    TIME("write16    1 GiB out, 16 B/lane nontemporal", (write16<<<n4 / 256, 256>>>(dst, n4)));
    TIME("write4nt   1 GiB out, 4 B/lane nontemporal stores (the coherent kernels' outputs)", (write4_nt<<<S / 256, 256>>>((float *)dst, S * 16)));
    TIME("write16r   1 GiB out, the same 16 planes, 16 B/lane nontemporal: a wave's 64 points as 4 x (4 rows x 256 B)", (write16_rows<<<S / 256, 256>>>((float *)dst, S * 16)));
    {
        float *o2; CK(hipMalloc(&o2, S * 8));
        CK(hipFuncSetAttribute((const void *)stream_walk<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        CK(hipFuncSetAttribute((const void *)stream_walk<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        CK(hipFuncSetAttribute((const void *)stream_walk<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        const int chunk = 512;
        TIME("walk d2 11w  1 GiB in + 128 MiB out: wave walks 512 points, 2 sets in flight, 13.9 KiB LDS (11 waves/CU)", (stream_walk<2, false><<<S / chunk, 64, 14240>>>((const float *)src, S, chunk, o2, nullptr)));
        TIME("walk d4 11w  the same, 4 sets in flight", (stream_walk<4, false><<<S / chunk, 64, 14240>>>((const float *)src, S, chunk, o2, nullptr)));
        TIME("walk d2 16w  2 sets, 10 KiB LDS (16 waves/CU)", (stream_walk<2, false><<<S / chunk, 64, 10240>>>((const float *)src, S, chunk, o2, nullptr)));
        TIME("walk d2 32w  2 sets, 4 KiB LDS (registers bound the waves)", (stream_walk<2, false><<<S / chunk, 64, 4096>>>((const float *)src, S, chunk, o2, nullptr)));
        TIME("walk d2 11w 2k  2 sets, 11 waves/CU, 2048 points per wave", (stream_walk<2, false><<<S / 2048, 64, 14240>>>((const float *)src, S, 2048, o2, nullptr)));
        TIME("walkw d2 11w  1 GiB in + 1 GiB out (16 dword stores per point) + 128 MiB, 11 waves/CU", (stream_walk<2, true><<<S / chunk, 64, 14240>>>((const float *)src, S, chunk, o2, (float *)dst)));
        TIME("walkw d2 32w  the same, 4 KiB LDS", (stream_walk<2, true><<<S / chunk, 64, 4096>>>((const float *)src, S, chunk, o2, (float *)dst)));
        CK(hipFree(o2));
    }
    if (argc > 1 && !strcmp(argv[1], "walk")) return 0;
    TIME("copy16     1 GiB in + 1 GiB out", (copy16<<<n4 / 256, 256>>>(src, dst, n4)));
    {
        const int64_t gpn = P * 4 / 256;
        TIME("mixed      gather64 + copy16 in one launch (workgroups alternate roles)", (mixed<<<(unsigned)(2 * 16 * gpn), 256>>>(table, nodes, W, P, dout, src, dst, n4, gpn)));
    }
    TIME("gatherread gather64 + 1 GiB of 4 B/lane stream loads in the same waves", (gather_read<<<gg, 256>>>(table, nodes, W, P, dout, (const float *)src)));
    TIME("scat64     2^24 rows of 64 B to random slots of 1 GiB (plain stores)", (scat64<<<S * 4 / 256, 256>>>((float4 *)dst, perm, S)));
    TIME("scat64nt   the same, nontemporal stores", (scat64_nt<<<S * 4 / 256, 256>>>(dst, perm, S)));
    TIME("scat16     2^24 x 16 B to random slots of 256 MiB", (scat16<<<S / 256, 256>>>((float4 *)dst, perm, S)));
    TIME("scat8      2^24 x 8 B to random slots of 128 MiB", (scat8<<<S / 256, 256>>>((float2 *)dst, perm, S)));
    TIME("fetch64    2^24 random 64-B row reads from 1 GiB", (fetch64<<<S * 4 / 256, 256>>>((const float4 *)src, perm, S, dout)));
    TIME("fetch8     2^24 random 8-B reads from 128 MiB", (fetch8<<<S / 256, 256>>>((const float2 *)src, perm, S, dout)));
    TIME("walk64     1 GiB of 64-B rows, sequential, walker shape (4 lanes/row, runs of 64 rows)", (walk64<<<S / 64 * 4 / 256, 256>>>((const float4 *)src, S, 64, dout)));
    {
        float2 *grid; CK(hipMalloc(&grid, S * 8));
        fill_grid<<<S / 256, 256>>>(grid, S);
        const int chunk = 32768, chunks = (int)(P / chunk);
        const int bins = (W + 1) * (W + 1), words = (bins + 1) / 2;
        uint32_t *dump; CK(hipMalloc(&dump, (size_t)N * chunks * words * 4));
        CK(hipFuncSetAttribute((const void *)hist, hipFuncAttributeMaxDynamicSharedMemorySize, words * 4));
        TIME("hist       2^24 samples, u16-pair LDS histogram of 66 049 cells per (table, 32k chunk) + ranks", (hist<<<dim3(chunks, N), 1024, words * 4>>>(grid, perm, W, chunk, P, dump)));
    }
    return 0;
}
