// tools/micro/lds_atomic_bench.hip: how fast are LDS atomics on gfx950?
// Each workgroup (256 threads) runs ITER wave-instructions of one kind on
// lane-distinct addresses (stride 1 word: conflict-free for plain accesses) and
// reports cycles; with 1 / 4 / 7 workgroups per CU the per-CU rate in lane
// operations per clock follows. Build: hipcc --offload-arch=gfx950 -O3 -o
// lds_atomic_bench lds_atomic_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int ITER = 2048;

template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned long long* cycles, unsigned* sink, int pad_words)
{
    extern __shared__ unsigned lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096 + pad_words; i += 256)
        lds[i] = 0;
    __syncthreads();
    unsigned acc = 0;
    unsigned long long* lds64 = reinterpret_cast<unsigned long long*>(lds);
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 8
    for (int it = 0; it < ITER; ++it) {
        const int a = (tid + it * 67) & 2047;          /* distinct per lane within an instruction */
        if (KIND == 0) lds[a] = it;                                   /* ds_write_b32 */
        if (KIND == 1) acc += lds[a];                                 /* ds_read_b32 */
        if (KIND == 2) atomicAdd(&lds[a], 1u);                        /* ds_add_u32 */
        if (KIND == 3) acc += atomicAdd(&lds[a], 1u);                 /* ds_add_rtn_u32 */
        if (KIND == 4) acc += atomicCAS(&lds[a], 0u, (unsigned)it);   /* ds_cmpst_rtn_b32 */
        if (KIND == 5) atomicAdd(&lds64[a], 1ull);                    /* ds_add_u64 */
        if (KIND == 6) acc += (unsigned)atomicAdd(&lds64[a], 1ull);   /* ds_add_rtn_u64 */
        if (KIND == 7) atomicOr(&lds64[a], 1ull << (it & 63));        /* ds_or_b64 */
        if (KIND == 8) atomicAdd(&lds[(a & 15)], 1u);                 /* 16 addresses: 4 lanes each */
        if (KIND == 9) atomicAdd(&lds[0], 1u);                        /* one address */
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0)
        cycles[blockIdx.x] = t1 - t0;
    if (acc == 0xdeadbeef)
        sink[0] = acc;
}

template <int KIND>
void run(const char* name, int wgs_per_cu, unsigned long long* d_cycles, unsigned* d_sink)
{
    const int n_cu = 256;
    const int blocks = n_cu * wgs_per_cu;
    /* LDS per workgroup chosen so that exactly wgs_per_cu fit a CU (160 KB) */
    const size_t lds = wgs_per_cu == 1 ? 100 * 1024 : wgs_per_cu == 4 ? 36 * 1024 : 20 * 1024;
    const int pad_words = (int)(lds / 4) - 4096;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), lds, 0, d_cycles, d_sink, 0 * pad_words);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), lds, 0, d_cycles, d_sink, 0 * pad_words);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(blocks);
    hipMemcpy(c.data(), d_cycles, blocks * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : c)
        mean += (double)v;
    mean /= blocks;
    /* per CU: wgs_per_cu * 4 waves * ITER instructions * 64 lanes in `mean` cycles */
    const double lane_ops_per_clk = (double)wgs_per_cu * 4 * ITER * 64 / mean;
    printf("%-28s %d WG/CU: %9.0f cycles/WG  %7.1f cycles per wave-instr (own)  %6.2f lane-ops/clk/CU  kernel %.1f us\n",
           name, wgs_per_cu, mean, mean / ITER, lane_ops_per_clk, ms * 1e3);
}

int main()
{
    unsigned long long* d_cycles;
    unsigned* d_sink;
    hipMalloc(&d_cycles, 256 * 8 * 8);
    hipMalloc(&d_sink, 64);
    for (int w : { 1, 4, 7 }) {
        run<0>("ds_write_b32", w, d_cycles, d_sink);
        run<1>("ds_read_b32", w, d_cycles, d_sink);
        run<2>("ds_add_u32", w, d_cycles, d_sink);
        run<3>("ds_add_rtn_u32", w, d_cycles, d_sink);
        run<4>("ds_cmpst_rtn_b32", w, d_cycles, d_sink);
        run<5>("ds_add_u64", w, d_cycles, d_sink);
        run<6>("ds_add_rtn_u64", w, d_cycles, d_sink);
        run<7>("ds_or_b64", w, d_cycles, d_sink);
        run<8>("ds_add_u32 16 addresses", w, d_cycles, d_sink);
        run<9>("ds_add_u32 1 address", w, d_cycles, d_sink);
    }
    return 0;
}
