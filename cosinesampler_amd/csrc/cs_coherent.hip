// cs_coherent.hip -- launchers of the coherent-points path (cs_coherent.cuh); its own translation unit so that the
// path compiles in parallel with the rest of the library.
#include <hip/hip_runtime.h>

#include <atomic>

#include "cs_coherent.cuh"
#include "cs_units.h"

namespace cs {
namespace coh {
namespace {

std::atomic<int> g_dbg{0}, g_chunk{1024}, g_wpb{2};

int status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
template <typename K>
int allow_lds(K kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return 0;
    hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return e == hipSuccess ? 0 : (int)e;
}
struct Geometry {
    dim3 grid;
    int block, chunk, dbg;
};
Geometry geometry(const Launch &L) {
    Geometry g;
    g.chunk = g_chunk.load(std::memory_order_relaxed);
    g.dbg = g_dbg.load(std::memory_order_relaxed);
    const int64_t waves = (L.d.P + g.chunk - 1) / g.chunk;
    const int wpb = g_wpb.load(std::memory_order_relaxed);
    g.grid = dim3((unsigned)((waves + wpb - 1) / wpb), (unsigned)L.d.N);
    g.block = 64 * wpb;
    return g;
}

#define COH_CQ(cq_, ...)                                            \
    switch (cq_) {                                                  \
        case 1:  { constexpr int CQ = 1; __VA_ARGS__; } break;      \
        case 2:  { constexpr int CQ = 2; __VA_ARGS__; } break;      \
        case 4:  { constexpr int CQ = 4; __VA_ARGS__; } break;      \
        default: { constexpr int CQ = 8; __VA_ARGS__; } break;      \
    }
#define COH_KERNEL_(k_, ...)                                                     \
    switch (k_) {                                                                \
        case 0:  { constexpr int KERNEL = K_COSINE; __VA_ARGS__; } break;        \
        case 1:  { constexpr int KERNEL = K_LINEAR; __VA_ARGS__; } break;        \
        default: { constexpr int KERNEL = K_SMOOTHSTEP; __VA_ARGS__; } break;    \
    }
#define COH_KERNEL(L_, ...)                                                                          \
    switch ((L_).sdt) {                                                                              \
        case 1:  { using ST = stream_f16; COH_KERNEL_((L_).kernel, __VA_ARGS__) } break;             \
        case 2:  { using ST = stream_bf16; COH_KERNEL_((L_).kernel, __VA_ARGS__) } break;            \
        default: { using ST = float; COH_KERNEL_((L_).kernel, __VA_ARGS__) } break;                  \
    }

}  // namespace

void set_chunk(int samples_per_wave, int ablation_bits) {   // experiments: 1 no scatter-reduce, 2 no window flush, 4 no products
    if (samples_per_wave >= 64) g_chunk.store((samples_per_wave + 63) / 64 * 64, std::memory_order_relaxed);
    g_dbg.store(ablation_bits & 7, std::memory_order_relaxed);
    if ((ablation_bits >> 4) >= 1 && (ablation_bits >> 4) <= 4) g_wpb.store(ablation_bits >> 4, std::memory_order_relaxed);   // waves per workgroup
}

bool supported(const Launch &L) {
    return L.d.size[0] <= MAX_SIZE && L.d.size[1] <= MAX_SIZE && L.d.N <= 65535 && L.d.P > 0 &&
           (L.d.P + 255) / 256 <= (int64_t)INT32_MAX &&
           (L.cq == 1 || L.cq == 2 || L.cq == 4 || L.cq == 8);
}

int backward(const Launch &L, const void *gOut, const float *icl, const float *grid, const float *offset, float *acc,
             float *grad_grid) {
    const Geometry g = geometry(L);
    int rc = 0;
    COH_KERNEL(L, COH_CQ(L.cq, {
        const size_t shm = (size_t)(g.block / 64) * wave_floats<4 * CQ>(false, false) * 4;
        rc = allow_lds(backward<KERNEL, CQ, ST>, shm);
        if (!rc) backward<KERNEL, CQ, ST><<<g.grid, g.block, shm, L.stream>>>((const ST *)gOut, icl, grid, offset, acc, grad_grid, L.d, L.f, g.chunk, g.dbg);
    }));
    return rc ? rc : status();
}

int bb(const Launch &L, const float *cG, const float *icl, const float *grid, const void *gOut, const float *offset,
       float *acc, float *gGrid, void *ggOut) {
    const Geometry g = geometry(L);
    int rc = 0;
    COH_KERNEL(L, COH_CQ(L.cq, {
        const size_t shm = (size_t)(g.block / 64) * wave_floats<4 * CQ>(false, true) * 4;
        rc = allow_lds(bb<KERNEL, CQ, ST>, shm);
        if (!rc) bb<KERNEL, CQ, ST><<<g.grid, g.block, shm, L.stream>>>(cG, icl, grid, (const ST *)gOut, offset, acc, gGrid, (ST *)ggOut, L.d, L.f, g.chunk, g.dbg);
    }));
    return rc ? rc : status();
}

int bbb(const Launch &L, const float *icl, const float *grid, const void *gOut, const float *cG, const float *hG,
        const void *hO, const float *offset, float *acc, void *ggOut) {
    const Geometry g = geometry(L);
    int rc = 0;
#define COH_BBB(TWO)                                                                                                    \
    COH_KERNEL(L, COH_CQ(L.cq, {                                                                                        \
        const size_t shm = (size_t)(g.block / 64) * wave_floats<4 * CQ>(false, false) * 4;                                             \
        rc = allow_lds(bbb<KERNEL, CQ, TWO, ST>, shm);                                                                  \
        if (!rc) bbb<KERNEL, CQ, TWO, ST><<<g.grid, g.block, shm, L.stream>>>(icl, grid, (const ST *)gOut, cG, hG, (const ST *)hO, offset, acc, (ST *)ggOut, L.d, L.f, g.chunk, g.dbg); \
    }))
    if (hO) { COH_BBB(true); } else { COH_BBB(false); }
#undef COH_BBB
    return rc ? rc : status();
}

}  // namespace coh
}  // namespace cs
