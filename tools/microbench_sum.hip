// Microbenchmarks for the sum-over-n ("PIXEL pattern") kernels: what does it cost when ONE lane walks all N
// tables for its point (all 64 MiB of tables hot at once, instead of one 4 MiB table per launch phase), and
// how fast are random row fetches from an array that fits the 256 MiB Infinity Cache but not the L2?
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench_sum tools/microbench_sum.hip && tools/microbench_sum
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                        \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// lane = point, loop over n: 4 nodes x 4 float4 per n (C = 16), position the same for all n (+ offset n/N)
template <int NB>
__global__ __launch_bounds__(256) void point_all_n(const float4 *table, int64_t nodes_per_n, int W, int n0, int64_t P,
                                                   float *out) {
    int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    uint32_t h = hash32((uint32_t)p * 2654435761u + 17u);
    float fx = (h & 0xffff) * (1.0f / 65536.0f) * (W - 2), fy = (h >> 16) * (1.0f / 65536.0f) * (W - 2);
    float4 acc[4] = {};
#pragma unroll 2
    for (int k = 0; k < NB; ++k) {
        int n = n0 + k;
        float off = n * (1.0f / 16.0f);
        int x = (int)(fx + off), y = (int)(fy + off);
        const float4 *base = table + ((int64_t)n * nodes_per_n + (int64_t)y * W + x) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 a = base[q], b = base[4 + q], c = base[(int64_t)W * 4 + q], d = base[(int64_t)W * 4 + 4 + q];
            acc[q].x += a.x + b.x + c.x + d.x;
            acc[q].y += a.y + b.y + c.y + d.y;
            acc[q].z += a.z + b.z + c.z + d.z;
            acc[q].w += a.w + b.w + c.w + d.w;
        }
    }
    float r = 0.f;
    for (int q = 0; q < 4; ++q) r += acc[q].x + acc[q].y + acc[q].z + acc[q].w;
    if (r == -12345.f) out[0] = r;
}

// 4 lanes = one point (each lane one float4 of the channels), loop over n
template <int NB>
__global__ __launch_bounds__(256) void quad_all_n(const float4 *table, int64_t nodes_per_n, int W, int n0, int64_t P,
                                                  float *out) {
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t p = t >> 2;
    int q = (int)(t & 3);
    if (p >= P) return;
    uint32_t h = hash32((uint32_t)p * 2654435761u + 17u);
    float fx = (h & 0xffff) * (1.0f / 65536.0f) * (W - 2), fy = (h >> 16) * (1.0f / 65536.0f) * (W - 2);
    float4 acc = {};
#pragma unroll 4
    for (int k = 0; k < NB; ++k) {
        int n = n0 + k;
        float off = n * (1.0f / 16.0f);
        int x = (int)(fx + off), y = (int)(fy + off);
        const float4 *base = table + ((int64_t)n * nodes_per_n + (int64_t)y * W + x) * 4 + q;
        float4 a = base[0], b = base[4], c = base[(int64_t)W * 4], d = base[(int64_t)W * 4 + 4];
        acc.x += a.x + b.x + c.x + d.x;
        acc.y += a.y + b.y + c.y + d.y;
        acc.z += a.z + b.z + c.z + d.z;
        acc.w += a.w + b.w + c.w + d.w;
    }
    float r = acc.x + acc.y + acc.z + acc.w;
    if (r == -12345.f) out[0] = r;
}

// one sample per lane-quad, table by table (what the shipped kernels do): sample s -> n = s / P
__global__ __launch_bounds__(256) void quad_per_sample(const float4 *table, int64_t nodes_per_n, int W, int64_t P,
                                                       int64_t S, float *out) {
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t s = t >> 2;
    int q = (int)(t & 3);
    if (s >= S) return;
    int n = (int)(s / P);
    int64_t p = s - (int64_t)n * P;
    uint32_t h = hash32((uint32_t)p * 2654435761u + 17u);
    float fx = (h & 0xffff) * (1.0f / 65536.0f) * (W - 2), fy = (h >> 16) * (1.0f / 65536.0f) * (W - 2);
    float off = n * (1.0f / 16.0f);
    int x = (int)(fx + off), y = (int)(fy + off);
    const float4 *base = table + ((int64_t)n * nodes_per_n + (int64_t)y * W + x) * 4 + q;
    float4 a = base[0], b = base[4], c = base[(int64_t)W * 4], d = base[(int64_t)W * 4 + 4];
    float r = a.x + b.y + c.z + d.w;
    if (r == -12345.f) out[0] = r;
}

// walker-style fetch: 4 lanes fetch one row of R float4 (only the first 4+1 used), row id = random over `rows`
// (rows * R * 16 bytes = the array size: 80 MiB fits the Infinity Cache, 1.25 GiB does not)
template <int R>
__global__ __launch_bounds__(256) void row_fetch(const float4 *src, uint32_t row_mask, int64_t fetches, float *out) {
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int q = (int)(t & 3);
    int64_t w = t >> 2;                       // walker id
    const int64_t per = 16;                   // fetches per walker
    float4 acc = {};
    for (int64_t i = 0; i < per; ++i) {
        int64_t j = w * per + i;
        if (j >= fetches) break;
        uint32_t r = hash32((uint32_t)j * 2654435761u + 99u) & row_mask;
        const float4 *row = src + (int64_t)r * R;
        float4 g = row[q], c = row[4];
        acc.x += g.x * c.x; acc.y += g.y * c.y; acc.z += g.z * c.z; acc.w += g.w * c.w;
    }
    float r = acc.x + acc.y + acc.z + acc.w;
    if (r == -12345.f) out[0] = r;
}

// row atomics into a 512 MiB accumulator: LANES lanes add one contiguous row of LANES floats; the row starts at a
// multiple of ALIGN floats (ALIGN = LANES: aligned rows; ALIGN = LANES/2: every other row straddles its natural boundary)
template <int LANES, int ALIGN>
__global__ __launch_bounds__(256) void atomic_rows(float *dst, uint32_t slot_mask, int64_t rows) {
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t r = t / LANES;
    int c = (int)(t % LANES);
    if (r >= rows) return;
    uint32_t slot = hash32((uint32_t)r * 2654435761u + 7u) & slot_mask;
    unsafeAtomicAdd(dst + (int64_t)slot * ALIGN + c, 1.0f);
}

// the same 64-byte row atomics with LOCALITY: a workgroup adds 256 rows to random slots of ONE 16 KiB window of the
// accumulator (what a pass over cell-sorted samples of a 3D tile would do) -- do the line fills go away?
__global__ __launch_bounds__(256) void atomic_rows_local(float *dst, uint32_t window_mask, int passes) {
    const uint32_t window = hash32(blockIdx.x * 2654435761u + 11u) & window_mask;
    const int c = threadIdx.x & 15;
    for (int i = 0; i < passes; ++i) {
        const uint32_t r = (uint32_t)i * 16u + (threadIdx.x >> 4);
        const uint32_t slot = hash32((blockIdx.x * 4096u + r) * 2246822519u + 3u) & 255u;
        unsafeAtomicAdd(dst + ((int64_t)window * 256 + slot) * 16 + c, 1.0f);
    }
}

// LDS float atomics: a workgroup adds `adds` values per thread to hashed slots of a 13 KiB image (a 3D tile's 9x9x5 nodes
// of 8 channels), then stores the image (so that nothing is optimised away).  RUN = 8: a lane adds 8 consecutive floats
// (one node row of 8 channels), as a per-corner lane of a 3D scatter would.
template <int RUN>
__global__ __launch_bounds__(256) void lds_atomics(float *out, int adds) {
    __shared__ float img[3240];
    for (int i = threadIdx.x; i < 3240; i += 256) img[i] = 0.f;
    __syncthreads();
    uint32_t h = hash32(blockIdx.x * 256u + threadIdx.x + 1u);
    for (int i = 0; i < adds; i += RUN) {
        h = hash32(h + 0x9e3779b9u);
        const uint32_t node = h % 405u;
#pragma unroll
        for (int c = 0; c < RUN; ++c) atomicAdd(&img[node * 8 + (RUN == 8 ? c : (h >> 16) & 7)], 1.0f);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3240; i += 256) out[(int64_t)(blockIdx.x & 1023) * 3240 + i] = img[i];
}

// gathers only, table by table without the 64-bit division of S1: blockIdx.y = n
__global__ __launch_bounds__(256) void quad_gather_n(const float4 *table, int64_t nodes_per_n, int W, int64_t P, float *out) {
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
// pure stream: copy n4 float4
__global__ __launch_bounds__(256) void stream_copy(const float4 *__restrict__ src, float4 *__restrict__ dst, int64_t n4) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(src) + i);
        __builtin_nontemporal_store(t, reinterpret_cast<v4f *>(dst) + i);
    }
}

// one launch, two roles: even workgroups gather (as quad_gather_n), odd workgroups copy (as stream_copy)
__global__ __launch_bounds__(256) void mixed_roles(const float4 *table, int64_t nodes_per_n, int W, int64_t P, float *out,
                                                   const float4 *__restrict__ src, float4 *__restrict__ dst, int64_t n4,
                                                   int64_t gather_blocks_per_n) {
    const int64_t b = blockIdx.x >> 1;
    if (blockIdx.x & 1) {
        int64_t i = b * 256 + threadIdx.x;
        if (i < n4) {
            typedef float v4f __attribute__((ext_vector_type(4)));
            v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(src) + i);
            __builtin_nontemporal_store(t, reinterpret_cast<v4f *>(dst) + i);
        }
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

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

#define TIME(label, ...)                                                                    \
    do {                                                                                    \
        float best = 1e9f;                                                                  \
        for (int rep = 0; rep < 4; ++rep) {                                                 \
            CK(hipEventRecord(e0)); __VA_ARGS__; CK(hipEventRecord(e1));                    \
            CK(hipEventSynchronize(e1));                                                    \
            float ms = time_ms(e0, e1);                                                     \
            if (rep && ms < best) best = ms;                                                \
        }                                                                                   \
        CK(hipGetLastError());                                                              \
        printf("%-88s %.3f ms\n", label, best);                                             \
    } while (0)

int main() {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float *dout; CK(hipMalloc(&dout, 64));
    const int W = 256, N = 16;
    const int64_t nodes = (int64_t)W * W, P = 1 << 20, S = P * N;
    float4 *table; CK(hipMalloc(&table, (size_t)N * nodes * 64 + 4096)); CK(hipMemset(table, 0, (size_t)N * nodes * 64));
    TIME("S1 per-sample quads, table by table (shipped layout), 2^24 samples", (quad_per_sample<<<S * 4 / 256, 256>>>(table, nodes, W, P, S, dout)));
    TIME("S2 lane=point, loop over all 16 tables", (point_all_n<16><<<P / 256, 256>>>(table, nodes, W, 0, P, dout)));
    TIME("S3 quad=point, loop over all 16 tables", (quad_all_n<16><<<P * 4 / 256, 256>>>(table, nodes, W, 0, P, dout)));
    TIME("S4 quad=point, 2 launches x 8 tables", ({ for (int g = 0; g < 2; ++g) quad_all_n<8><<<P * 4 / 256, 256>>>(table, nodes, W, g * 8, P, dout); }));
    TIME("S5 quad=point, 4 launches x 4 tables", ({ for (int g = 0; g < 4; ++g) quad_all_n<4><<<P * 4 / 256, 256>>>(table, nodes, W, g * 4, P, dout); }));
    TIME("S6 quad=point, 8 launches x 2 tables", ({ for (int g = 0; g < 8; ++g) quad_all_n<2><<<P * 4 / 256, 256>>>(table, nodes, W, g * 2, P, dout); }));
    TIME("S7 quad=point, 16 launches x 1 table", ({ for (int g = 0; g < 16; ++g) quad_all_n<1><<<P * 4 / 256, 256>>>(table, nodes, W, g, P, dout); }));
    TIME("S8 lane=point, 4 launches x 4 tables", ({ for (int g = 0; g < 4; ++g) point_all_n<4><<<P / 256, 256>>>(table, nodes, W, g * 4, P, dout); }));
    // row fetches: 2^24 fetches of 80 B rows (R = 5 float4) / 96 B (R = 6)
    float4 *rows; CK(hipMalloc(&rows, (size_t)(1 << 24) * 96)); CK(hipMemset(rows, 0, (size_t)(1 << 24) * 96));
    const int64_t F = 1 << 24;
    TIME("R1 2^24 fetches of 80 B rows, random over 2^24 rows (1.25 GiB, HBM)", (row_fetch<5><<<F / 16 * 4 / 256, 256>>>(rows, (1u << 24) - 1, F, dout)));
    TIME("R2 2^24 fetches of 80 B rows, random over 2^20 rows (80 MiB, Infinity Cache)", (row_fetch<5><<<F / 16 * 4 / 256, 256>>>(rows, (1u << 20) - 1, F, dout)));
    TIME("R3 2^24 fetches of 80 B rows, random over 2^16 rows (5 MiB, ~L2)", (row_fetch<5><<<F / 16 * 4 / 256, 256>>>(rows, (1u << 16) - 1, F, dout)));
    TIME("R4 2^24 fetches of 128 B-stride rows, random over 2^20 rows (128 MiB)", (row_fetch<8><<<F / 16 * 4 / 256, 256>>>(rows, (1u << 20) - 1, F, dout)));
    TIME("R5 2^24 fetches of 64 B-stride rows (G only + coef from next row), 2^20 rows (64 MiB)", (row_fetch<4><<<F / 16 * 4 / 256, 256>>>(rows, (1u << 20) - 1, F, dout)));
    {   // do L2-resident gathers and HBM streams overlap when they run side by side (two streams)?
        hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
        hipEvent_t f1, f2; CK(hipEventCreate(&f1)); CK(hipEventCreate(&f2));
        float4 *src, *dst; const int64_t n4 = (int64_t)1 << 26;   // 1 GiB each way
        CK(hipMalloc(&src, n4 * 16)); CK(hipMalloc(&dst, n4 * 16)); CK(hipMemset(src, 0, n4 * 16));
        dim3 gg((unsigned)(P * 4 / 256), 16);
        TIME("O1 gathers alone (2^24 samples, table by table)", (quad_gather_n<<<gg, 256>>>(table, nodes, W, P, dout)));
        TIME("O2 stream copy alone (1 GiB read + 1 GiB write)", (stream_copy<<<n4 / 256, 256>>>(src, dst, n4)));
        TIME("O3 both, one after the other on one stream", ({ quad_gather_n<<<gg, 256>>>(table, nodes, W, P, dout); stream_copy<<<n4 / 256, 256>>>(src, dst, n4); }));
        TIME("O4 both, side by side on two streams", ({
            CK(hipEventRecord(f1, 0)); CK(hipStreamWaitEvent(s1, f1, 0)); CK(hipStreamWaitEvent(s2, f1, 0));
            quad_gather_n<<<gg, 256, 0, s1>>>(table, nodes, W, P, dout);
            stream_copy<<<n4 / 256, 256, 0, s2>>>(src, dst, n4);
            CK(hipEventRecord(f1, s1)); CK(hipEventRecord(f2, s2));
            CK(hipStreamWaitEvent(0, f1, 0)); CK(hipStreamWaitEvent(0, f2, 0)); }));
        {   // gather blocks: 16 * P*4/256 = 262144 ; copy blocks: n4/256 = 262144 -> interleave 1:1
            const int64_t gpn = P * 4 / 256;
            TIME("O5 both in ONE launch, workgroups alternate roles", (mixed_roles<<<(unsigned)(2 * 16 * gpn), 256>>>(table, nodes, W, P, dout, src, dst, n4, gpn)));
        }
        CK(hipFree(src)); CK(hipFree(dst));
    }
    {   // 3D config 4: 2^25 node-row updates of 32 B (C = 8) into 512 MiB
        float *acc; CK(hipMalloc(&acc, (size_t)512 << 20)); CK(hipMemset(acc, 0, (size_t)512 << 20));
        const int64_t R = 1 << 25;
        TIME("A1 2^25 row atomics,  8 lanes x 4 B = 32 B rows, 32 B aligned", (atomic_rows<8, 8><<<R * 8 / 256, 256>>>(acc, (1u << 24) - 1, R)));
        TIME("A2 2^24 row atomics, 16 lanes x 4 B = 64 B rows, 64 B aligned (pairs of the above)", (atomic_rows<16, 16><<<(R / 2) * 16 / 256, 256>>>(acc, (1u << 23) - 1, R / 2)));
        TIME("A3 2^24 row atomics, 16 lanes x 4 B = 64 B rows, 32 B aligned (half of them straddle)", (atomic_rows<16, 8><<<(R / 2) * 16 / 256, 256>>>(acc, (1u << 24) - 2, R / 2)));
        TIME("A4 2^23 row atomics, 32 lanes x 4 B = 128 B rows, 128 B aligned", (atomic_rows<32, 32><<<(R / 4) * 32 / 256, 256>>>(acc, (1u << 22) - 1, R / 4)));
        TIME("A6 2^24 64-B row atomics, each workgroup's 256 rows inside one random 16 KiB window", (atomic_rows_local<<<(R / 2) / 256, 256>>>(acc, (1u << 15) - 1, 16)));
        TIME("A7 2^24 64-B row atomics, 64 rows per workgroup and window (4x as many workgroups)", (atomic_rows_local<<<(R / 2) / 64, 256>>>(acc, (1u << 15) - 1, 4)));
        TIME("L1 2^28 LDS float atomics, random floats of a 13 KiB image, 65536 workgroups x 256 threads x 16", (lds_atomics<1><<<65536, 256>>>(acc, 16)));
        TIME("L2 2^28 LDS float atomics, runs of 8 consecutive floats (node rows)", (lds_atomics<8><<<65536, 256>>>(acc, 16)));
        TIME("A5 2^23 row atomics, 32 lanes x 4 B = 128 B rows, 32 B aligned", (atomic_rows<32, 8><<<(R / 4) * 32 / 256, 256>>>(acc, (1u << 24) - 4, R / 4)));
    }
    return 0;
}
