// Microbenchmark: cost of LDS atomics on gfx950 as the MI stage uses them (one wave64 per workgroup,
// random keys over B*B cells).  Build: hipcc --offload-arch=gfx950 -O3 -o lds_atomic_bench lds_atomic_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int ITER = 2000;
constexpr int KK = 8;

template <int MODE>
__global__ __launch_bounds__(64) void bench(const uint32_t *keys, uint32_t *out, unsigned long long *cycles, int cells) {
    __shared__ uint32_t hist[4096];
    const int lane = threadIdx.x;
    for (int i = lane; i < 4096; i += 64) hist[i] = 0;
    uint32_t key[KK];
    for (int k = 0; k < KK; ++k) key[k] = keys[(blockIdx.x * KK + k) * 64 + lane] % cells;
    if (MODE == 1) for (int k = 0; k < KK; ++k) key[k] = lane + 64 * k;   // conflict free
    if (MODE == 4) for (int k = 0; k < KK; ++k) key[k] = k;               // all lanes same address
    __syncthreads();
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int k = 0; k < KK; ++k) {
            if (MODE == 3) atomicAdd(&hist[key[k]], 1u);                  // no return
            else if (MODE == 5) acc += hist[key[k]];                      // plain read
            else if (MODE == 6) acc += __hip_atomic_fetch_add(&hist[key[k] >> 1], 1u << (16 * (key[k] & 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else acc += atomicAdd(&hist[key[k]], 1u);                     // returning
        }
        if (MODE == 5) { __builtin_amdgcn_s_waitcnt(0xc07f); }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cycles[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 64 + lane] = acc + hist[lane];
}

template <int MODE>
void run(const char *name, int blocks, int cells, const uint32_t *dkeys, uint32_t *dout, unsigned long long *dcyc) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(64), 0, 0, dkeys, dout, dcyc, cells);
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(64), 0, 0, dkeys, dout, dcyc, cells);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> c(blocks);
    CHECK(hipMemcpy(c.data(), dcyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double mean = 0; for (auto v : c) mean += v; mean /= blocks;
    const double ninst = (double)ITER * KK;
    // per-CU LDS time per wave-instruction when `blocks` waves share the chip: wall * 256 CUs... report both
    printf("%-34s blocks %5d cells %4d  in-wave cycles/instr %7.2f   wall ns/instr/CU-wave-slot %7.3f (ms %.3f)\n", name, blocks,
           cells, mean / ninst, ms * 1e6 / ninst / ((double)blocks / 256.0), ms);
}

int main() {
    const int maxblocks = 256 * 8;
    std::vector<uint32_t> keys((size_t)maxblocks * KK * 64);
    srand(1);
    for (auto &k : keys) k = (uint32_t)rand();
    uint32_t *dkeys, *dout; unsigned long long *dcyc;
    CHECK(hipMalloc(&dkeys, keys.size() * 4)); CHECK(hipMalloc(&dout, maxblocks * 64 * 4)); CHECK(hipMalloc(&dcyc, maxblocks * 8));
    CHECK(hipMemcpy(dkeys, keys.data(), keys.size() * 4, hipMemcpyHostToDevice));
    for (int blocks : {256, 256 * 4, 256 * 8}) {
        run<1>("add_rtn conflict-free", blocks, 4096, dkeys, dout, dcyc);
        run<2>("add_rtn random 289 cells", blocks, 289, dkeys, dout, dcyc);
        run<2>("add_rtn random 1521 cells", blocks, 1521, dkeys, dout, dcyc);
        run<3>("add (no rtn) random 289 cells", blocks, 289, dkeys, dout, dcyc);
        run<4>("add_rtn same address", blocks, 289, dkeys, dout, dcyc);
        run<5>("ds_read_b32 random 289 cells", blocks, 289, dkeys, dout, dcyc);
        run<6>("add_rtn u16-packed random 289", blocks, 289, dkeys, dout, dcyc);
    }
    return 0;
}
