// tools/memset_graph_repro.hip -- does a hipMemsetAsync captured into a HIP graph run on EVERY replay?
// Round 1 replaced the library's hipMemsetAsync by a zero-fill kernel after a memset node was seen to be skipped on a
// replay of a captured stage (cs_abi.hip, zero_fill).  This is the minimal form of that observation, with the runtime
// version, so that it can be re-checked on a newer ROCm:   hipcc --offload-arch=gfx950 -o /tmp/mgr tools/memset_graph_repro.hip && /tmp/mgr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
__global__ void add_one(float *p, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] += 1.0f; }
__global__ void poison(float *p, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = 1000.0f; }
int main() {
    int rt = 0, drv = 0;
    CK(hipRuntimeGetVersion(&rt)); CK(hipDriverGetVersion(&drv));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, HIP runtime %d, driver %d\n", prop.gcnArchName, rt, drv);
    int bad_total = 0;
    for (size_t mib : {1, 64, 512}) {
        const int n = (int)(mib << 18);
        float *d; CK(hipMalloc(&d, (size_t)n * 4));
        hipStream_t s; CK(hipStreamCreate(&s));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        CK(hipMemsetAsync(d, 0, (size_t)n * 4, s));
        add_one<<<(n + 255) / 256, 256, 0, s>>>(d, n);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        int bad = 0;
        std::vector<float> h(4);
        for (int rep = 0; rep < 8; ++rep) {
            poison<<<(n + 255) / 256, 256, 0, s>>>(d, n);          // what a skipped memset would leave: 1001 after add_one
            CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            CK(hipMemcpy(h.data(), d, 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(h.data() + 2, d + n - 2, 8, hipMemcpyDeviceToHost));
            if (h[0] != 1.0f || h[1] != 1.0f || h[2] != 1.0f || h[3] != 1.0f) { ++bad; printf("  %zu MiB replay %d: got %g %g %g %g (expected 1)\n", mib, rep, h[0], h[1], h[2], h[3]); }
        }
        printf("memset node of %zu MiB: %d of 8 replays wrong\n", mib, bad);
        bad_total += bad;
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(s)); CK(hipFree(d));
    }
    printf(bad_total ? "RESULT: memset nodes were skipped on replay\n" : "RESULT: memset nodes replayed faithfully in this form\n");
    return 0;
}
