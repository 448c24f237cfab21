// glds_probe.hip -- semantics of __builtin_amdgcn_global_load_lds on gfx950 as stage 4 of the RPF kernels uses it: per-lane
// global address, LDS destination = wave-uniform base + lane * size.  Each lane gathers a[perm[lane]] (4 bytes) into LDS and
// reads its own word back.  Prints "glds ok" or the first mismatch.   hipcc --offload-arch=gfx950 -O2 glds_probe.hip -o glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(const float *a, const int *perm, float *out, int rounds) {
    __shared__ float buf[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
        const float *src = a + perm[(lane + 7 * r) & 63] + 64 * r;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)&buf[wv][0], 4, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += buf[wv][lane];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    out[threadIdx.x] = acc;
}

int main() {
    const int R = 5;
    std::vector<float> a(64 * R);
    std::vector<int> perm(64);
    for (int i = 0; i < 64 * R; ++i) a[i] = (float)(i * 3 + 1);
    for (int i = 0; i < 64; ++i) perm[i] = (i * 37 + 11) & 63;
    float *da, *dout; int *dp;
    hipMalloc(&da, a.size() * 4); hipMalloc(&dp, 256); hipMalloc(&dout, 256 * 4);
    hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dp, perm.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, da, dp, dout, R);
    std::vector<float> out(256);
    hipMemcpy(out.data(), dout, 1024, hipMemcpyDeviceToHost);
    for (int t = 0; t < 256; ++t) {
        const int lane = t & 63;
        float want = 0.f;
        for (int r = 0; r < R; ++r) want += a[perm[(lane + 7 * r) & 63] + 64 * r];
        if (out[t] != want) { printf("glds MISMATCH thread %d: got %g want %g\n", t, out[t], want); return 1; }
    }
    printf("glds ok\n");
    return 0;
}
