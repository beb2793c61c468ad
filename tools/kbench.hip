// tools/kbench.hip -- the PRODUCT's hot kernels on synthetic config-2 inputs, launched back to back from C++ (no Python,
// no allocator, no other kernels in between): what a kernel costs by itself, and -D ablations of it.  Not product code.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DKB_...] -o tools/kbench tools/kbench.hip && tools/kbench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "kbench_kernels.cuh"

#define CK(x)                                                                        \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

__device__ __forceinline__ uint32_t khash(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__global__ void fill_grid_rep(float2 *grid, int64_t P, int N) {   // PIXEL pattern: the same P points for every n
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    uint32_t h = khash((uint32_t)i * 2654435761u + 3u);
    float2 g = make_float2((h & 0xffff) * (2.0f / 65536.0f) - 1.f, (h >> 16) * (2.0f / 65536.0f) - 1.f);
    for (int n = 0; n < N; ++n) grid[(int64_t)n * P + i] = g;
}
__global__ void fill_rand(float *x, int64_t n, uint32_t seed) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] = (khash((uint32_t)i * 2654435761u + seed) & 0xffffff) * (1.0f / 16777216.0f);
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }
#define TIME(label, ...)                                                                    \
    do {                                                                                    \
        float best = 1e9f;                                                                  \
        for (int rep = 0; rep < 5; ++rep) {                                                 \
            CK(hipEventRecord(e0)); __VA_ARGS__; CK(hipEventRecord(e1));                    \
            CK(hipEventSynchronize(e1));                                                    \
            float ms = time_ms(e0, e1);                                                     \
            if (rep && ms < best) best = ms;                                                \
        }                                                                                   \
        CK(hipGetLastError());                                                              \
        printf("%-80s %.3f ms\n", label, best);                                             \
    } while (0)

__global__ __launch_bounds__(256) void pollute(const float4 *__restrict__ a, float4 *__restrict__ b, int64_t n4) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) { float4 v = a[i]; v.x += 1.f; b[i] = v; }
}
#define TIME_COLD(label, ...)                                                               \
    do {                                                                                    \
        float best = 1e9f;                                                                  \
        for (int rep = 0; rep < 5; ++rep) {                                                 \
            pollute<<<(unsigned)(pn4 / 256), 256>>>(pa, pb, pn4);                           \
            CK(hipEventRecord(e0)); __VA_ARGS__; CK(hipEventRecord(e1));                    \
            CK(hipEventSynchronize(e1));                                                    \
            float ms = time_ms(e0, e1);                                                     \
            if (rep && ms < best) best = ms;                                                \
        }                                                                                   \
        CK(hipGetLastError());                                                              \
        printf("%-80s %.3f ms (after 2 GiB of other traffic)\n", label, best);              \
    } while (0)

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    using namespace cs;
    namespace tl = cs::tiled;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 16, C = 16, H = 256, W = 256;
    const int64_t P = 1 << 20, S = P * N, vol = (int64_t)H * W;
    Dims d{};
    d.N = N; d.C = C; d.size[0] = W; d.size[1] = H; d.size[2] = 1; d.P = P; d.S = S; d.vol = vol;
    d.tab_ns = 1; d.tab_cs = vol; d.go_ns = d.ho_ns = (int64_t)C * P;
    Flags f{0, 1, 1, 0};
    float *input, *icl, *grid, *out, *gOut, *offset;
    CK(hipMalloc(&input, N * C * vol * 4)); CK(hipMalloc(&icl, N * C * vol * 4)); CK(hipMalloc(&grid, S * 8));
    CK(hipMalloc(&out, S * C * 4)); CK(hipMalloc(&gOut, S * C * 4)); CK(hipMalloc(&offset, N * 4));
    fill_rand<<<(unsigned)((N * C * vol + 255) / 256), 256>>>(input, N * C * vol, 1);
    fill_rand<<<(unsigned)((S * C + 255) / 256), 256>>>(gOut, S * C, 2);
    fill_grid_rep<<<(unsigned)(P / 256), 256>>>((float2 *)grid, P, N);
    float hoff[N];
    for (int n = 0; n < N; ++n) hoff[n] = (float)n / N;
    CK(hipMemcpy(offset, hoff, sizeof(hoff), hipMemcpyHostToDevice));
    tl::pack_channels_last<<<dim3((unsigned)((vol + 63) / 64), N), 256, (size_t)C * 65 * 4>>>(input, icl, C, C, vol);
    CK(hipDeviceSynchronize());
    dim3 pg((unsigned)(P / 256), N);
    TIME("point_forward (round 1 as shipped: VGPR gathers, LDS result tile)",
         (tl::point_forward<0, 4><<<pg, 256, (size_t)4 * (tl::REC_FLOATS + C * tl::OUT_LD) * 4>>>(icl, grid, offset, out, d, f)));
    TIME("point_forward3 NBUF=2", (tl::point_forward3<0, 4, 2><<<pg, 256, tl::f3_lds<4, 2>()>>>(icl, grid, offset, out, d, f)));
    TIME("point_forward3 NBUF=1", (tl::point_forward3<0, 4, 1><<<pg, 256, tl::f3_lds<4, 1>()>>>(icl, grid, offset, out, d, f)));
    float4 *pa, *pb; const int64_t pn4 = (int64_t)1 << 26;
    CK(hipMalloc(&pa, pn4 * 16)); CK(hipMalloc(&pb, pn4 * 16)); CK(hipMemset(pa, 0, pn4 * 16));
    TIME_COLD("point_forward", (tl::point_forward<0, 4><<<pg, 256, (size_t)4 * (tl::REC_FLOATS + C * tl::OUT_LD) * 4>>>(icl, grid, offset, out, d, f)));
    TIME_COLD("point_forward3 NBUF=2", (tl::point_forward3<0, 4, 2><<<pg, 256, tl::f3_lds<4, 2>()>>>(icl, grid, offset, out, d, f)));
    TIME_COLD("point_forward3 NBUF=1", (tl::point_forward3<0, 4, 1><<<pg, 256, tl::f3_lds<4, 1>()>>>(icl, grid, offset, out, d, f)));
    TIME_COLD("point_forward3 NBUF=2, no gathers", (tl::point_forward3<0, 4, 2, 1><<<pg, 256, tl::f3_lds<4, 2>()>>>(icl, grid, offset, out, d, f)));
    TIME_COLD("point_forward3 NBUF=2, no stores", (tl::point_forward3<0, 4, 2, 2><<<pg, 256, tl::f3_lds<4, 2>()>>>(icl, grid, offset, out, d, f)));
    TIME_COLD("point_forward3 NBUF=2, no grid load (hashed positions)", (tl::point_forward3<0, 4, 2, 4><<<pg, 256, tl::f3_lds<4, 2>()>>>(icl, grid, offset, out, d, f)));
    TIME_COLD("point_forward3 NBUF=2, no grid load, no stores", (tl::point_forward3<0, 4, 2, 6><<<pg, 256, tl::f3_lds<4, 2>()>>>(icl, grid, offset, out, d, f)));
    TIME_COLD("point_forward3 NBUF=2, no grid load, no gathers (stores only)", (tl::point_forward3<0, 4, 2, 5><<<pg, 256, tl::f3_lds<4, 2>()>>>(icl, grid, offset, out, d, f)));
    TIME_COLD("point_forward3 NBUF=2, grid load only", (tl::point_forward3<0, 4, 2, 3><<<pg, 256, tl::f3_lds<4, 2>()>>>(icl, grid, offset, out, d, f)));
    TIME_COLD("pack + point_forward3 NBUF=2", ({ tl::pack_channels_last<<<dim3((unsigned)((vol + 63) / 64), N), 256, (size_t)C * 65 * 4>>>(input, icl, C, C, vol); tl::point_forward3<0, 4, 2><<<pg, 256, tl::f3_lds<4, 2>()>>>(icl, grid, offset, out, d, f); }));
    return 0;
}
