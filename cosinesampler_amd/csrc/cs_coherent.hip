// cs_coherent.hip -- launchers of the coherent-points path (cs_coherent.cuh); its own translation unit so that the
// path compiles in parallel with the rest of the library.
#include <hip/hip_runtime.h>

#include <atomic>

#include "cs_coherent.cuh"
#include "cs_units.h"

namespace cs {
namespace coh {
// the summing kernels (CS_SUM_OVER_N) are compiled in their own unit, cs_coherent_sum.hip
int launch_nsum(int mode, bool two, bool scat, const Launch &L, const Args &a, int dbg);
namespace {

std::atomic<int> g_dbg{0}, g_chunk{0}, g_wpb{1};      // g_chunk 0: the per-stage defaults below
// samples per wave.  A wave pays one table-window load when it starts, so longer chunks amortise better, but the grid is
// ~12 rounds of single-wave workgroups and shorter chunks leave a smaller idle tail.  Swept INSIDE the step (tools/
// chunk_sweep.sh: the ordered-points step of bench.py; a stage timed back to back with itself ranks them differently --
// profiles/round4_ablation.txt): forward 0.322 / 0.293 / 0.278 / 0.329 / 0.346 ms at 128 / 192 / 256 / 384 / 512 (256 = the
// four batches of coordinates it keeps in flight: every load of the wave goes out at once), first backward 0.358 / 0.342 /
// 0.347 at 256 / 384 / 512, second 0.557 / 0.548 / 0.561, third 0.836 / 0.819 / 0.824.
#ifndef CS_COH_CHUNKS
#define CS_COH_CHUNKS {256, 384, 384, 384}
#endif
constexpr int kChunk[4] = CS_COH_CHUNKS;
#ifndef CS_COH_LDS_EXTRA
#define CS_COH_LDS_EXTRA 0     // experiments (tools/ab.sh): unused LDS bytes per wave, to cap the waves a CU holds
#endif

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
Geometry geometry(const Launch &L, int mode) {
    Geometry g;
    g.chunk = g_chunk.load(std::memory_order_relaxed);
    if (g.chunk >= (1 << 16)) g.chunk = 64 * ((g.chunk >> (8 * mode)) & 0xFF);   // experiments: one byte per stage, in batches of 64
    if (g.chunk <= 0) g.chunk = kChunk[mode];
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

template <int MODE, bool TWO, bool SCAT = true>
int launch(const Launch &L, const Args &a) {
    if (L.nsum) return launch_nsum(MODE, TWO, SCAT, L, a, g_dbg.load(std::memory_order_relaxed));   // cs_coherent_sum.hip
    const Geometry g = geometry(L, MODE);
    int rc = 0;
#ifndef CS_COH_NO_COMMON
    if (L.sdt == 0 && L.f.pad == PAD_ZEROS && L.f.align) {     // fp32 streams, zeros padding, align_corners: the specialised kernels
        COH_KERNEL_(L.kernel, COH_CQ(L.cq, {
            using ST = float;
            const size_t shm = (size_t)(g.block / 64) * (wave_floats<4 * CQ>(SCAT ? MODE : FWD) * 4 + CS_COH_LDS_EXTRA);
            rc = allow_lds(stage<KERNEL, CQ, MODE, TWO, SCAT, ST, true>, shm);
            if (!rc) stage<KERNEL, CQ, MODE, TWO, SCAT, ST, true><<<g.grid, g.block, shm, L.stream>>>(a, L.d, L.f, g.chunk, g.dbg);
        }));
        return rc ? rc : status();
    }
#endif
    COH_KERNEL(L, COH_CQ(L.cq, {
        const size_t shm = (size_t)(g.block / 64) * wave_floats<4 * CQ>(SCAT ? MODE : FWD) * 4;
        rc = allow_lds(stage<KERNEL, CQ, MODE, TWO, SCAT, ST>, shm);
        if (!rc) stage<KERNEL, CQ, MODE, TWO, SCAT, ST><<<g.grid, g.block, shm, L.stream>>>(a, L.d, L.f, g.chunk, g.dbg);
    }));
    return rc ? rc : status();
}

}  // namespace

void set_chunk(int samples_per_wave, int ablation_bits) {   // experiments: 1 no scatter-reduce, 2 no window flush, 4 no products
    // (values from 2^16 up: four bytes, one per stage, forward in the lowest: the samples per wave of that stage in batches of 64)
    g_chunk.store(samples_per_wave >= (1 << 16) ? samples_per_wave : samples_per_wave >= 64 ? (samples_per_wave + 63) / 64 * 64 : 0,
                  std::memory_order_relaxed);
    g_dbg.store((ablation_bits & 7) | ((ablation_bits >> 8) & 15) << 3, std::memory_order_relaxed);     // + 256 / 512: store policies (unused now), + 1024: no table-window loads, + 2048: no scatter operands to LDS
    if (((ablation_bits >> 4) & 15) >= 1 && ((ablation_bits >> 4) & 15) <= 4) g_wpb.store((ablation_bits >> 4) & 15, std::memory_order_relaxed);   // waves per workgroup
}

bool supported_nsum(const Launch &L) {
    // (the 2D forward ignores align_corners, 2d.cu:307-308: COMMON's align = 1 is what it uses anyway; every other stage
    // honours the flag, so the caller's must be set)
    return supported(L) && L.sdt == 0 && L.f.pad == PAD_ZEROS && L.f.align && L.d.grid_ns == 0;
}

bool supported(const Launch &L) {
    // (a stream's buffer descriptor reaches fifteen channel rows and a chunk past the wave's first sample in 32 bits:
    // beyond P = 2^26 points per table the general path takes over)
    return L.d.size[0] <= MAX_SIZE && L.d.size[1] <= MAX_SIZE && L.d.N <= 65535 && L.d.P > 0 &&
           (L.d.P + 255) / 256 <= (int64_t)INT32_MAX && L.d.P <= ((int64_t)1 << 26) &&
           (L.cq == 1 || L.cq == 2 || L.cq == 4 || L.cq == 8);
}

int forward(const Launch &L, const float *icl, const float *grid, const float *offset, void *output) {
    Args a{};
    a.icl = icl;
    a.grid = grid;
    a.offset = offset;
    a.out_stream = output;
    return launch<FWD, false>(L, a);
}

int backward(const Launch &L, const void *gOut, const float *icl, const float *grid, const float *offset, float *acc,
             float *grad_grid) {
    Args a{};
    a.icl = icl;
    a.grid = grid;
    a.offset = offset;
    a.gOut = gOut;
    a.acc = acc;
    a.out_grid = grad_grid;
    return acc ? launch<BWD, false>(L, a) : launch<BWD, false, false>(L, a);
}

int bb(const Launch &L, const float *cG, const float *icl, const float *grid, const void *gOut, const float *offset,
       float *acc, float *gGrid, void *ggOut) {
    Args a{};
    a.icl = icl;
    a.grid = grid;
    a.offset = offset;
    a.gOut = gOut;
    a.cG = cG;
    a.acc = acc;
    a.out_grid = gGrid;
    a.out_stream = ggOut;
    return acc ? launch<BB, false>(L, a) : launch<BB, false, false>(L, a);
}

int bbb(const Launch &L, const float *icl, const float *grid, const void *gOut, const float *cG, const float *hG,
        const void *hO, const float *offset, float *acc, void *ggOut) {
    Args a{};
    a.icl = icl;
    a.grid = grid;
    a.offset = offset;
    a.gOut = gOut;
    a.hO = hO;
    a.cG = cG;
    a.hG = hG;
    a.acc = acc;
    a.out_stream = ggOut;
    return hO ? launch<BBB, true>(L, a) : launch<BBB, false>(L, a);
}

}  // namespace coh
}  // namespace cs
