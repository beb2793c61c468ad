// cs_coherent_sum.hip -- launchers of the summing kernels (cs_coherent.cuh, NSUM; CS_SUM_OVER_N): their own translation
// unit so that they compile in parallel with the plain coherent kernels (cs_coherent.hip).
#include <hip/hip_runtime.h>

#include "cs_coherent.cuh"
#include "cs_units.h"

namespace cs {
namespace coh {
namespace {

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

#define SUM_CQ(cq_, ...)                                            \
    switch (cq_) {                                                  \
        case 1:  { constexpr int CQ = 1; __VA_ARGS__; } break;      \
        case 2:  { constexpr int CQ = 2; __VA_ARGS__; } break;      \
        case 4:  { constexpr int CQ = 4; __VA_ARGS__; } break;      \
        default: { constexpr int CQ = 8; __VA_ARGS__; } break;      \
    }
#define SUM_KERNEL(k_, ...)                                                      \
    switch (k_) {                                                                \
        case 0:  { constexpr int KERNEL = K_COSINE; __VA_ARGS__; } break;        \
        case 1:  { constexpr int KERNEL = K_LINEAR; __VA_ARGS__; } break;        \
        default: { constexpr int KERNEL = K_SMOOTHSTEP; __VA_ARGS__; } break;    \
    }

// A wave owns 64 * DEPTH points for every table.  Built for fp32 streams with zeros padding and align_corners (the COMMON
// specialisation): the PIXEL pattern; anything else is the caller's to sum (supported_nsum).
template <int MODE, bool TWO, bool SCAT>
int launch_one(const Launch &L, const Args &a, int dbg) {
    constexpr int chunk = 64 * depth(MODE);                // the kernel keeps exactly its DEPTH register sets: one chunk
    const int64_t waves = (L.d.P + chunk - 1) / chunk;
    int rc = 0;
    SUM_KERNEL(L.kernel, SUM_CQ(L.cq, {
        using ST = float;
        const size_t shm = (size_t)wave_floats<4 * CQ>(SCAT ? MODE : FWD) * 4;
        rc = allow_lds(stage<KERNEL, CQ, MODE, TWO, SCAT, ST, true, true>, shm);
        if (!rc) stage<KERNEL, CQ, MODE, TWO, SCAT, ST, true, true><<<dim3((unsigned)waves), 64, shm, L.stream>>>(a, L.d, L.f, chunk, dbg);
    }));
    return rc ? rc : status();
}

}  // namespace

int launch_nsum(int mode, bool two, bool scat, const Launch &L, const Args &a, int dbg) {
    switch (mode) {
        case FWD: return launch_one<FWD, false, true>(L, a, dbg);
        case BWD: return scat ? launch_one<BWD, false, true>(L, a, dbg) : launch_one<BWD, false, false>(L, a, dbg);
        case BB:  return scat ? launch_one<BB, false, true>(L, a, dbg) : launch_one<BB, false, false>(L, a, dbg);
        default:  return two ? launch_one<BBB, true, true>(L, a, dbg) : launch_one<BBB, false, true>(L, a, dbg);
    }
}

}  // namespace coh
}  // namespace cs
