// Pins the semantics of the 64-bit DPP row broadcast used by stage 4 (csrc/rpf_xlane.h: fmac_rowbc / rowbc) on the device it
// runs on:  hipcc --offload-arch=gfx950 -O2 -o /tmp/dpp_bcast_probe scripts/microbench/dpp_bcast_probe.hip && /tmp/dpp_bcast_probe
// Expected: rowbc<N> gives lane L the value of lane (L & ~15) + N; fmac_rowbc<N>(acc, row, z) = acc + row[(L & ~15) + N] * z.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../raytracer-rpf_amd/csrc/rpf_xlane.h"

__global__ void probe(double *out) {
    const int lane = threadIdx.x;
    const double v = 100.0 + lane;
    out[0 * 64 + lane] = rpf::xl::rowbc<0>(v);
    out[1 * 64 + lane] = rpf::xl::rowbc<5, true>(v);
    out[2 * 64 + lane] = rpf::xl::rowbc<15>(v);
    double acc = 1000.0 * lane;
    rpf::xl::fmac_rowbc<3, true>(acc, v, 2.0);
    out[3 * 64 + lane] = acc;
    double acc2 = 0.5;
    rpf::xl::fmac_rowbc<9>(acc2, v, (double)lane);
    out[4 * 64 + lane] = acc2;
}

int main() {
    double *d, h[5 * 64];
    hipMalloc(&d, sizeof(h));
    probe<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int r = l & ~15;
        bad += h[l] != 100.0 + r;
        bad += h[64 + l] != 100.0 + r + 5;
        bad += h[128 + l] != 100.0 + r + 15;
        bad += h[192 + l] != 1000.0 * l + (100.0 + r + 3) * 2.0;
        bad += h[256 + l] != 0.5 + (100.0 + r + 9) * l;
    }
    const char *names[5] = {"rowbc<0>", "rowbc<5>", "rowbc<15>", "fmac_rowbc<3>(1000 L, v, 2)", "fmac_rowbc<9>(0.5, v, L)"};
    for (int t = 0; t < 5; ++t) {
        printf("%-30s", names[t]);
        for (int l = 0; l < 64; ++l) printf(" %g", h[t * 64 + l]);
        printf("\n");
    }
    printf("dpp_bcast_probe: %d mismatches\n", bad);
    return bad != 0;
}
