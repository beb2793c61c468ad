/* oracle/asan_driver.c -- TEST INFRASTRUCTURE: runs every function of the CPU restatement (cs_oracle.c, included below so
 * that the sanitizers see its code) over exactly-sized heap buffers under AddressSanitizer + UBSan: every padding mode,
 * both align_corners settings, the three blending kernels, multicell on / off, points inside, on and far outside [-1, 1],
 * with and without the optional tensors.  `make -C oracle asan-run` builds and runs it; tests/test_oracle_golden.py does
 * that on the CPU ("CPU restatement under ASan/UBSan").  Exit status 0 = no finding. */
#include "cs_oracle.c"

#include <stdio.h>

static float *buf(size_t n, unsigned *seed, float lo, float hi) {
    float *p = (float *)malloc((n ? n : 1) * sizeof(float));
    if (!p) exit(2);
    for (size_t i = 0; i < n; ++i) {
        *seed = *seed * 1664525u + 1013904223u;
        p[i] = lo + (hi - lo) * (float)(*seed >> 8) / 16777216.0f;
    }
    return p;
}

int main(void) {
    unsigned seed = 12345u;
    int runs = 0;
    for (int dim = 2; dim <= 3; ++dim) {
        const int64_t N = 3, C = 2, D = dim == 3 ? 4 : 1, H = 5, W = 6, P = 41;
        const size_t vol = (size_t)(D * H * W);
        for (int pad = 0; pad < 3; ++pad)
            for (int align = 0; align < 2; ++align)
                for (int kernel = 0; kernel < 3; ++kernel)
                    for (int mc = 0; mc < 2; ++mc) {
                        float *input = buf(N * C * vol, &seed, 0.f, 1.f);
                        float *grid = buf(N * P * dim, &seed, -1.4f, 1.4f);
                        grid[0] = -1.f; grid[1] = 1.f; grid[2] = 0.f; grid[3] = 1e30f; grid[4] = -1e30f;
                        float *offset = buf(N, &seed, 0.f, mc ? 0.9f : 0.f);
                        float *gOut = buf(N * C * P, &seed, -1.f, 1.f), *hO = buf(N * C * P, &seed, -1.f, 1.f);
                        float *cG = buf(N * P * dim, &seed, -1.f, 1.f), *hG = buf(N * P * dim, &seed, -1.f, 1.f);
                        float *cI = buf(N * C * vol, &seed, -1.f, 1.f);
                        float *out = buf(N * C * P, &seed, 0.f, 0.f), *ggOut = buf(N * C * P, &seed, 0.f, 0.f);
                        float *gI = buf(N * C * vol, &seed, 0.f, 0.f), *gG = buf(N * P * dim, &seed, 0.f, 0.f);
                        int rc = 0;
                        if (dim == 2) {
                            rc |= cs2d_forward_cpu(input, grid, offset, out, N, C, H, W, P, pad, align, kernel, mc);
                            rc |= cs2d_backward_cpu(gOut, input, grid, offset, gI, gG, N, C, H, W, P, pad, align, kernel, mc);
                            rc |= cs2d_backward_cpu(gOut, input, grid, offset, NULL, gG, N, C, H, W, P, pad, align, kernel, mc);
                            rc |= cs2d_backward_backward_cpu(cI, cG, input, grid, gOut, offset, gI, gG, ggOut, N, C, H, W, P, pad, align, kernel, mc);
                            rc |= cs2d_backward_backward_cpu(NULL, cG, input, grid, gOut, offset, gI, gG, ggOut, N, C, H, W, P, pad, align, kernel, mc);
                            rc |= cs2d_backward_backward_backward_cpu(input, grid, gOut, cG, hG, offset, gI, ggOut, N, C, H, W, P, pad, align, kernel, mc);
                        } else {
                            rc |= cs3d_forward_cpu(input, grid, offset, out, N, C, D, H, W, P, pad, align, kernel, mc);
                            rc |= cs3d_backward_cpu(gOut, input, grid, offset, gI, gG, N, C, D, H, W, P, pad, align, kernel, mc);
                            rc |= cs3d_backward_cpu(gOut, input, grid, offset, NULL, gG, N, C, D, H, W, P, pad, align, kernel, mc);
                            rc |= cs3d_backward_backward_cpu(cI, cG, input, grid, gOut, offset, gI, gG, ggOut, N, C, D, H, W, P, pad, align, kernel, mc);
                            rc |= cs3d_backward_backward_cpu(NULL, cG, input, grid, gOut, offset, gI, gG, ggOut, N, C, D, H, W, P, pad, align, kernel, mc);
                            rc |= cs3d_backward_backward_backward_cpu(input, grid, gOut, cG, hG, offset, gI, ggOut, N, C, D, H, W, P, pad, align, kernel, mc);
                        }
                        if (rc) { fprintf(stderr, "oracle call failed\n"); return 1; }
                        runs += 6;
                        free(input); free(grid); free(offset); free(gOut); free(hO); free(cG); free(hG); free(cI);
                        free(out); free(ggOut); free(gI); free(gG);
                    }
    }
    printf("asan driver: %d oracle calls, no finding\n", runs);
    return 0;
}
