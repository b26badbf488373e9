/* tools/micro/pk_fma_bench.hip -- issue rate of v_pk_fma_f32 (two fp32 multiply-adds per lane
 * and instruction) against v_mad_u32_u24 and v_fma_f32 on gfx950: cycles per wave-instruction
 * with 1, 2 and 4 waves per SIMD. hipcc --offload-arch=gfx950 -O3 -o pk_fma_bench pk_fma_bench.hip */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long* out, int iters, float seed)
{
    typedef float float2v __attribute__((ext_vector_type(2)));
    float2v a[8];
    unsigned int u[16];
    float2v m = { seed, seed };
    unsigned int mi = (unsigned int)seed + 3u;
    for (int i = 0; i < 8; ++i) {
        a[i] = float2v{ (float)i, (float)(i + 1) };
        u[2 * i] = i;
        u[2 * i + 1] = i + 7;
    }
    float2v c = { 1.0f + threadIdx.x, 2.0f };
    unsigned int ci = threadIdx.x + 5;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(c), "v"(m));
#pragma unroll
            for (int i = 0; i < 8; ++i)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(c), "v"(m));
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(u[i]) : "v"(ci), "s"(mi));
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(c.x), "v"(m.x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].y) : "v"(c.y), "v"(m.x));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    unsigned int su = 0;
    for (int i = 0; i < 8; ++i) {
        s += a[i].x + a[i].y;
        su += u[2 * i] + u[2 * i + 1];
    }
    if (threadIdx.x == 0)
        out[blockIdx.x] = t1 - t0;
    if (s == 12345.f && su == 77)
        out[0] = 0;
}

int main()
{
    unsigned long long* d;
    (void)hipMalloc(&d, 65536 * 8);
    const int iters = 20000;
    const char* names[] = { "v_pk_fma_f32", "v_mad_u32_u24", "v_fma_f32" };
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode)
        for (int waves_per_simd : { 1, 2, 4, 8 }) {
            /* 256-thread workgroups (one wave per SIMD each); waves_per_simd of them per CU */
            const int blocks = 256 * waves_per_simd;
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.5f);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.5f);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.5f);
                (void)hipEventRecord(e1, 0);
                (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            const double wave_instr = (double)blocks * 4 * iters * 16.0;
            printf("%-14s %d waves/SIMD: %.3f ms, %.3e wave-instr/s chip-wide = %.3f per SIMD per ns\n", names[mode],
                   waves_per_simd, ms, wave_instr / (ms * 1e-3), wave_instr / (ms * 1e-3) / 1024 / 1e9);
        }
    return 0;
}
