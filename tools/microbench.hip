// tools/microbench.hip -- design microbenchmarks for the scatter/gather paths (not product code).
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench tools/microbench.hip && ./tools/microbench
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// ---- MB1: LDS float atomics, random addresses in an nwords array ------------------------------
__global__ __launch_bounds__(256) void lds_atomic(float *out, int nwords, int iters) {
    extern __shared__ float acc[];
    for (int i = threadIdx.x; i < nwords; i += 256) acc[i] = 0.f;
    __syncthreads();
    uint32_t s = hash32(blockIdx.x * 256 + threadIdx.x + 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            s = s * 1664525u + 1013904223u;
            uint32_t a = __umulhi(s, (uint32_t)nwords);
            atomicAdd(&acc[a], 1.0f);
        }
    }
    __syncthreads();
    float t = 0.f;
    for (int i = threadIdx.x; i < nwords; i += 256) t += acc[i];
    if (t == -1.f) out[0] = t;
}

// ---- MB2: scatter-write rows of ROWB bytes to random row slots ---------------------------------
template <int ROWF>  // floats per row handled by one lane-quad group: 4 lanes x float4 = 16 floats
__global__ __launch_bounds__(256) void scatter_rows64(float4 *dst, const uint32_t *perm, int64_t rows) {
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t r = t >> 2;
    int q = t & 3;
    if (r >= rows) return;
    uint32_t slot = perm[r];
    float4 v = make_float4((float)r, 1.f, 2.f, (float)q);
    dst[(int64_t)slot * 4 + q] = v;
}
__global__ __launch_bounds__(256) void scatter_16B(float4 *dst, const uint32_t *perm, int64_t rows) {
    int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    dst[perm[r]] = make_float4((float)r, 1.f, 2.f, 3.f);
}
__global__ __launch_bounds__(256) void scatter_8B(float2 *dst, const uint32_t *perm, int64_t rows) {
    int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    dst[perm[r]] = make_float2((float)r, 1.f);
}
// gather rows of 64 B (lane quad) from random slots, reduce so it is not optimised away
__global__ __launch_bounds__(256) void gather_rows64(const float4 *src, const uint32_t *perm, int64_t rows, float *out) {
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t r = t >> 2;
    int q = t & 3;
    if (r >= rows) return;
    float4 v = src[(int64_t)perm[r] * 4 + q];
    if (v.x + v.y + v.z + v.w == -12345.f) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void stream_read(const float4 *src, int64_t n4, float *out) {
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n4) return;
    float4 v = src[t];
    if (v.x + v.y + v.z + v.w == -12345.f) out[0] = 1.f;
}
__global__ void make_perm(uint32_t *perm, int64_t n, uint32_t mask, uint32_t mul) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    // bijection on [0, 2^k): odd multiplier + xor-shift rounds
    uint32_t x = (uint32_t)i;
    x = (x * mul) & mask; x ^= x >> 7; x = (x * 0x9E3779B1u) & mask; x ^= x >> 11; x = (x * 0x85EBCA6Bu | 1u) & mask;
    perm[i] = x & mask;
}

// ---- MB4: global float atomics: 16 lanes add to one random 64 B row (4 rows per wave instr) ----
__global__ __launch_bounds__(256) void atomic_rows64(float *dst, const uint32_t *perm, int64_t rows, int reps) {
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t r = t >> 4;
    int c = t & 15;
    if (r >= rows) return;
    for (int k = 0; k < reps; ++k) {
        uint32_t slot = perm[(r + (int64_t)k * 7919) % rows];
        unsafeAtomicAdd(dst + (int64_t)slot * 16 + c, 1.0f);
    }
}
__global__ __launch_bounds__(256) void atomic_scalar(float *dst, const uint32_t *perm, int64_t n) {
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    unsafeAtomicAdd(dst + perm[t], 1.0f);
}

// ---- MB5: hardware sin/cos accuracy on [0,1] ---------------------------------------------------
__global__ void trig_err(float *maxerr, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float t = (float)i / (float)(n - 1);
    float c_hw = __builtin_amdgcn_cosf(0.5f * t);   // v_cos_f32: input in revolutions
    float s_hw = __builtin_amdgcn_sinf(0.5f * t);
    double c = cos(M_PI * (double)t), s = sin(M_PI * (double)t);
    float e1 = (float)fabs((double)c_hw - c), e2 = (float)fabs((double)s_hw - s);
    float e3 = (float)fabs((double)cospif(t) - c), e4 = (float)fabs((double)sinpif(t) - s);
    float e5 = (float)fabs((double)__cosf(3.141592654f * t) - c), e6 = (float)fabs((double)__sinf(3.141592654f * t) - s);
    atomicMax((int *)&maxerr[0], __float_as_int(e1));
    atomicMax((int *)&maxerr[1], __float_as_int(e2));
    atomicMax((int *)&maxerr[2], __float_as_int(e3));
    atomicMax((int *)&maxerr[3], __float_as_int(e4));
    atomicMax((int *)&maxerr[4], __float_as_int(e5));
    atomicMax((int *)&maxerr[5], __float_as_int(e6));
}

// ---- MB6: quad gather of 64 B node rows from a channels-last table (4 corners per sample) ------
__global__ __launch_bounds__(256) void quad_gather(const float4 *table, int64_t nodes_per_n, int W, int n_count,
                                                   int64_t samples, float *out, int xcd_affine) {
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t s = t >> 2;
    int q = t & 3;
    if (s >= samples) return;
    int64_t per_n = samples / n_count;
    int n;
    if (xcd_affine) {   // blocks b, b+8, ... share an XCD: give them the same n
        int64_t blocks_per_n = per_n * 4 / 256;
        int64_t b = blockIdx.x;
        int x = b % 8;
        int64_t k = b / 8;
        n = (int)((k / blocks_per_n) * 8 + x);
        if (n >= n_count) n = n % n_count;
    } else {
        n = (int)(s / per_n);
    }
    uint32_t h = hash32((uint32_t)s * 2654435761u + 17u);
    int x = h % (W - 1), y = (h >> 12) % (W - 1);
    const float4 *base = table + ((int64_t)n * nodes_per_n) * 4 + q;
    float4 a = base[((int64_t)y * W + x) * 4];
    float4 b = base[((int64_t)y * W + x + 1) * 4];
    float4 c = base[((int64_t)(y + 1) * W + x) * 4];
    float4 d = base[((int64_t)(y + 1) * W + x + 1) * 4];
    float r = a.x + b.y + c.z + d.w;
    if (r == -12345.f) out[0] = r;
}

// MB6 variants: mode 1 = x forced even (nw|ne share one 128 B line), mode 2 = nontemporal loads,
// mode 3 = even x + 8 lanes per sample, each lane 2 loads (row pair = 128 B contiguous per 8 lanes)
typedef float vf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ntload(const float4 *p) {
    vf4 v = __builtin_nontemporal_load(reinterpret_cast<const vf4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
template <int MODE>
__global__ __launch_bounds__(256) void quad_gather_v(const float4 *table, int64_t nodes_per_n, int W, int n_count,
                                                     int64_t samples, float *out) {
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    constexpr int LPS = (MODE == 3) ? 8 : 4;
    int64_t s = t / LPS;
    int q = t % LPS;
    if (s >= samples) return;
    int n = (int)(s / (samples / n_count));
    uint32_t h = hash32((uint32_t)s * 2654435761u + 17u);
    int x = h % (W - 1), y = (h >> 12) % (W - 1);
    if (MODE == 1 || MODE == 3) x &= ~1;
    const float4 *base = table + ((int64_t)n * nodes_per_n) * 4 + q;
    float r;
    if (MODE == 3) {
        float4 a = base[((int64_t)y * W + x) * 4];
        float4 c = base[((int64_t)(y + 1) * W + x) * 4];
        r = a.x + c.z;
    } else if (MODE == 2) {
        float4 a = ntload(&base[((int64_t)y * W + x) * 4]);
        float4 b = ntload(&base[((int64_t)y * W + x + 1) * 4]);
        float4 c = ntload(&base[((int64_t)(y + 1) * W + x) * 4]);
        float4 d = ntload(&base[((int64_t)(y + 1) * W + x + 1) * 4]);
        r = a.x + b.y + c.z + d.w;
    } else {
        float4 a = base[((int64_t)y * W + x) * 4];
        float4 b = base[((int64_t)y * W + x + 1) * 4];
        float4 c = base[((int64_t)(y + 1) * W + x) * 4];
        float4 d = base[((int64_t)(y + 1) * W + x + 1) * 4];
        r = a.x + b.y + c.z + d.w;
    }
    if (r == -12345.f) out[0] = r;
}
// lane = sample, 16 x dwordx4 per lane (the layout of the direct kernels, channels-last table)
__global__ __launch_bounds__(256) void lane_gather(const float4 *table, int64_t nodes_per_n, int W, int n_count,
                                                   int64_t samples, float *out) {
    int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= samples) return;
    int n = (int)(s / (samples / n_count));
    uint32_t h = hash32((uint32_t)s * 2654435761u + 17u);
    int x = h % (W - 1), y = (h >> 12) % (W - 1);
    const float4 *base = table + ((int64_t)n * nodes_per_n) * 4;
    float r = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float4 a = base[((int64_t)y * W + x) * 4 + q];
        float4 b = base[((int64_t)y * W + x + 1) * 4 + q];
        float4 c = base[((int64_t)(y + 1) * W + x) * 4 + q];
        float4 d = base[((int64_t)(y + 1) * W + x + 1) * 4 + q];
        r += a.x + b.y + c.z + d.w;
    }
    if (r == -12345.f) out[0] = r;
}

// MB7: lane = sample writes a row of R float4 to a random slot (row stride = R*16 bytes)
template <int R>
__global__ __launch_bounds__(256) void scatter_rows_lane(float4 *dst, const uint32_t *perm, int64_t rows) {
    int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    float4 *d = dst + (int64_t)perm[r] * R;
#pragma unroll
    for (int k = 0; k < R; ++k) d[k] = make_float4((float)r, 1.f, (float)k, 3.f);
}
template <int R>
__global__ __launch_bounds__(256) void read_rows_lane(const float4 *src, int64_t rows, float *out) {
    int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float4 *d = src + r * R;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < R; ++k) { float4 v = d[k]; acc += v.x + v.w; }
    if (acc == -12345.f) out[0] = acc;
}
template <int R>
void run_rows(float4 *buf, const uint32_t *perm, int64_t rows, float *dout, hipEvent_t e0, hipEvent_t e1) {
    float ms_w = 0, ms_r = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        scatter_rows_lane<R><<<rows / 256, 256>>>(buf, perm, rows);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms_w, e0, e1));
        CK(hipEventRecord(e0));
        read_rows_lane<R><<<rows / 256, 256>>>(buf, rows, dout);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms_r, e0, e1));
    }
    printf("MB7 lane-per-row scatter-write 2^24 rows of %3d B: %.3f ms (%.0f GB/s); lane-per-row sequential read: %.3f ms (%.0f GB/s)\n",
           R * 16, ms_w, rows * R * 16.0 / ms_w * 1e-6, ms_r, rows * R * 16.0 / ms_r * 1e-6);
}

// MB8: walker-style gather of fat rows: 16 lanes = one walker, reads for each of its rows 16 floats
// (one per lane) + NCOEF float4 coefficient records broadcast; rows visited in `perm` order.
// STRIDE = floats between rows.
template <int STRIDE, int NPAY, int NCOEF>
__global__ __launch_bounds__(256) void walker_gather(const float *fat, const uint32_t *perm, int64_t rows, float *out) {
    const int w = (blockIdx.x * 256 + threadIdx.x) >> 4, c = threadIdx.x & 15;
    const int64_t per = 256;                       // rows per walker
    int64_t j0 = (int64_t)w * per;
    if (j0 >= rows) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int64_t j = j0; j < j0 + per; j += 4) {
        uint32_t r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = perm[j + u];
        float g[4][NPAY];
        float4 k[4][NCOEF];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float *row = fat + (int64_t)r[u] * STRIDE;
#pragma unroll
            for (int q = 0; q < NPAY; ++q) g[u][q] = row[16 * q + c];
#pragma unroll
            for (int q = 0; q < NCOEF; ++q) k[u][q] = *reinterpret_cast<const float4 *>(row + 16 * NPAY + 4 * q);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < NPAY; ++q) {
                float4 kk = k[u][q < NCOEF ? q : 0];
                a0 = fmaf(kk.x, g[u][q], a0); a1 = fmaf(kk.y, g[u][q], a1);
                a2 = fmaf(kk.z, g[u][q], a2); a3 = fmaf(kk.w, g[u][q], a3);
            }
    }
    if (a0 + a1 + a2 + a3 == -12345.f) out[0] = a0;
}
template <int STRIDE, int NPAY, int NCOEF>
void run_walker(const float *buf, const uint32_t *perm, int64_t rows, float *dout, hipEvent_t e0, hipEvent_t e1, const char *what) {
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        walker_gather<STRIDE, NPAY, NCOEF><<<(unsigned)(rows / 256 * 16 / 256), 256>>>(buf, perm, rows, dout);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("MB8 walker gather %-34s stride %3d B, useful %3d B/row: %.3f ms (%.0f GB/s useful)\n", what, STRIDE * 4,
           (16 * NPAY + 4 * NCOEF) * 4, ms, rows * (16.0 * NPAY + 4 * NCOEF) * 4 / ms * 1e-6);
}
__global__ void make_ident(uint32_t *perm, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) perm[i] = (uint32_t)i;
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main() {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float *dout; CK(hipMalloc(&dout, 64));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs %d\n", prop.gcnArchName, prop.multiProcessorCount);

    // MB5
    {
        float *d; CK(hipMalloc(&d, 32)); CK(hipMemset(d, 0, 32));
        int n = 1 << 22;
        trig_err<<<(n + 255) / 256, 256>>>(d, n);
        float h[6]; CK(hipMemcpy(h, d, 24, hipMemcpyDeviceToHost));
        printf("MB5 max abs err on [0,1]: v_cos %.3e v_sin %.3e | cospif %.3e sinpif %.3e | __cosf %.3e __sinf %.3e\n",
               h[0], h[1], h[2], h[3], h[4], h[5]);
    }
    // MB1
    for (int nwords : {289 * 16, 1089 * 16, 16384}) {
        int blocks = 256 * 8, iters = 256;
        size_t shm = (size_t)nwords * 4;
        CK(hipFuncSetAttribute((const void *)lds_atomic, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        lds_atomic<<<blocks, 256, shm>>>(dout, nwords, 4);
        CK(hipEventRecord(e0));
        lds_atomic<<<blocks, 256, shm>>>(dout, nwords, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        double ops = (double)blocks * 256 * iters * 16;
        float ms = time_ms(e0, e1);
        printf("MB1 LDS atomics random over %6d words: %.3f ms, %.1f Gops/s chip (2^30 ops = %.3f ms)\n", nwords, ms,
               ops / ms * 1e-6, 1073741824.0 / (ops / ms));
    }
    // MB2/3/4
    {
        const int64_t rows = 1 << 24;           // 2^24 rows x 64 B = 1 GiB
        float4 *buf; uint32_t *perm;
        CK(hipMalloc(&buf, rows * 64)); CK(hipMalloc(&perm, rows * 4));
        make_perm<<<rows / 256, 256>>>(perm, rows, (uint32_t)(rows - 1), 2654435761u);
        CK(hipMemset(buf, 0, rows * 64));
        CK(hipDeviceSynchronize());
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            scatter_rows64<16><<<rows * 4 / 256, 256>>>(buf, perm, rows);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        printf("MB2 scatter-write 2^24 random 64 B rows: %.3f ms (%.0f GB/s payload)\n", time_ms(e0, e1), rows * 64.0 / time_ms(e0, e1) * 1e-6);
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            scatter_16B<<<rows / 256, 256>>>(buf, perm, rows);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        printf("MB2 scatter-write 2^24 random 16 B (dense 256 MiB target): %.3f ms (%.0f GB/s payload)\n", time_ms(e0, e1), rows * 16.0 / time_ms(e0, e1) * 1e-6);
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            scatter_8B<<<rows / 256, 256>>>((float2 *)buf, perm, rows);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        printf("MB2 scatter-write 2^24 random 8 B (dense 128 MiB target): %.3f ms (%.0f GB/s payload)\n", time_ms(e0, e1), rows * 8.0 / time_ms(e0, e1) * 1e-6);
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            gather_rows64<<<rows * 4 / 256, 256>>>(buf, perm, rows, dout);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        printf("MB3 gather-read 2^24 random 64 B rows of 1 GiB: %.3f ms (%.0f GB/s)\n", time_ms(e0, e1), rows * 64.0 / time_ms(e0, e1) * 1e-6);
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            stream_read<<<rows * 4 / 256, 256>>>(buf, rows * 4, dout);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        printf("MB3 stream-read 1 GiB: %.3f ms (%.0f GB/s)\n", time_ms(e0, e1), rows * 64.0 / time_ms(e0, e1) * 1e-6);
        // atomics on a 64 MiB table (2^20 rows of 64 B)
        const int64_t trows = 1 << 20;
        make_perm<<<trows / 256, 256>>>(perm, trows, (uint32_t)(trows - 1), 2654435761u);
        CK(hipDeviceSynchronize());
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            atomic_rows64<<<trows * 16 / 256, 256>>>((float *)buf, perm, trows, 16);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        {
            double reqs = (double)trows * 16;
            printf("MB4 global atomics, random 64 B rows (16 lanes/row): %.3f ms for 2^24 row-adds -> %.1f G rows/s, %.0f GB/s added\n",
                   time_ms(e0, e1), reqs / time_ms(e0, e1) * 1e-6, reqs * 64 / time_ms(e0, e1) * 1e-6);
        }
        make_perm<<<rows / 256, 256>>>(perm, rows, (uint32_t)(rows - 1), 2654435761u);
        CK(hipDeviceSynchronize());
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            atomic_scalar<<<rows / 256, 256>>>((float *)buf, perm, rows);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        printf("MB4 global atomics, random 4 B scalars over 64 MiB: %.3f ms for 2^24 -> %.1f G/s\n", time_ms(e0, e1), rows / time_ms(e0, e1) * 1e-6);
        // MB6
        for (int ncount : {1, 2, 8, 16}) for (int aff = 0; aff < 2; ++aff) {
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                quad_gather<<<rows * 4 / 256, 256>>>(buf, 65536, 256, ncount, rows, dout, aff);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            }
            printf("MB6 quad gather 4x64 B per sample, 2^24 samples, %d tables of 4 MiB, xcd_affine=%d: %.3f ms (%.0f GB/s L1 traffic)\n",
                   ncount, aff, time_ms(e0, e1), rows * 256.0 / time_ms(e0, e1) * 1e-6);
        }
        CK(hipFree(buf));
        CK(hipMalloc(&buf, rows * 192));
        make_perm<<<rows / 256, 256>>>(perm, rows, (uint32_t)(rows - 1), 2654435761u);
        run_rows<4>(buf, perm, rows, dout, e0, e1);
        run_rows<5>(buf, perm, rows, dout, e0, e1);
        run_rows<6>(buf, perm, rows, dout, e0, e1);
        run_rows<8>(buf, perm, rows, dout, e0, e1);
        run_rows<10>(buf, perm, rows, dout, e0, e1);
        run_rows<12>(buf, perm, rows, dout, e0, e1);
        CK(hipFree(buf));
        CK(hipMalloc(&buf, rows * 256));
        CK(hipMemset(buf, 0, rows * 256));
        make_perm<<<rows / 256, 256>>>(perm, rows, (uint32_t)(rows - 1), 2654435761u);
        run_walker<16, 1, 0>((float *)buf, perm, rows, dout, e0, e1, "random, G only");
        run_walker<20, 1, 1>((float *)buf, perm, rows, dout, e0, e1, "random, G+coef packed");
        run_walker<32, 1, 1>((float *)buf, perm, rows, dout, e0, e1, "random, G+coef in a 128 B line");
        run_walker<40, 2, 2>((float *)buf, perm, rows, dout, e0, e1, "random, 2 payloads + 2 coef packed");
        run_walker<48, 2, 2>((float *)buf, perm, rows, dout, e0, e1, "random, 2+2 stride 192");
        run_walker<64, 2, 2>((float *)buf, perm, rows, dout, e0, e1, "random, 2+2 in 256 B");
        make_ident<<<rows / 256, 256>>>(perm, rows);
        run_walker<16, 1, 0>((float *)buf, perm, rows, dout, e0, e1, "sequential, G only");
        run_walker<20, 1, 1>((float *)buf, perm, rows, dout, e0, e1, "sequential, G+coef packed");
        run_walker<40, 2, 2>((float *)buf, perm, rows, dout, e0, e1, "sequential, 2+2 packed");
        for (int rep = 0; rep < 2; ++rep) { CK(hipEventRecord(e0)); quad_gather_v<1><<<rows * 4 / 256, 256>>>(buf, 65536, 256, 16, rows, dout); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); }
        printf("MB6b quad gather, x even (pairs share a 128 B line): %.3f ms\n", time_ms(e0, e1));
        for (int rep = 0; rep < 2; ++rep) { CK(hipEventRecord(e0)); quad_gather_v<2><<<rows * 4 / 256, 256>>>(buf, 65536, 256, 16, rows, dout); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); }
        printf("MB6c quad gather, nontemporal loads: %.3f ms\n", time_ms(e0, e1));
        for (int rep = 0; rep < 2; ++rep) { CK(hipEventRecord(e0)); quad_gather_v<3><<<rows * 8 / 256, 256>>>(buf, 65536, 256, 16, rows, dout); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); }
        printf("MB6d oct gather, x even, 8 lanes x 2 loads (128 B per row pair): %.3f ms\n", time_ms(e0, e1));
        for (int rep = 0; rep < 2; ++rep) { CK(hipEventRecord(e0)); lane_gather<<<rows / 256, 256>>>(buf, 65536, 256, 16, rows, dout); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); }
        printf("MB6e lane=sample gather 16 x dwordx4: %.3f ms\n", time_ms(e0, e1));
    }
    return 0;
}
