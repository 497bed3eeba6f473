// ubench_valu.hip -- development aid: issue cost of the VALU instructions the int8 scan is made of, on gfx950.
// Each kernel runs `iters` rounds of 16 independent instructions per wave; cycles from s_memtime.
// waves/SIMD = blockDim.x / 256 (one workgroup per CU).  Build: hipcc --offload-arch=gfx950 -O2 ubench_valu.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define REP16(X) X X X X X X X X X X X X X X X X

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void k(uint64_t *out, int iters, float seed)
{
    float a[16];
    f32x2 p[16];
    uint32_t u[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        a[i] = seed + i;
        p[i] = f32x2{seed + i, seed - i};
        u[i] = (uint32_t)(seed * 1000) + i * 77u;
    }
    const float b = seed * 0.5f;
    const f32x2 pb = f32x2{b, b + 1.0f};
    const uint32_t lane4 = threadIdx.x * 4u;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0)
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            else if (OP == 1)
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
            else if (OP == 2)
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
            else if (OP == 3)
                asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(lane4), "s"(0x0c0c0004u));
            else if (OP == 4)
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            else if (OP == 5)
                asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(p[i]) : "s"(pb));
            else if (OP == 6)
                asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(pb));
            else if (OP == 7)
                asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        s += a[i] + p[i].x + p[i].y;
        x ^= u[i];
    }
    if (s == 123.456f || x == 0xdeadbeefu)
        out[1] = 1;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0 && threadIdx.x < 256)
        out[0] = t1 - t0;
    else if ((threadIdx.x & 63) == 0 && blockIdx.x == 0)
        out[2 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
void run(const char *name, uint64_t *d_out, int waves_per_simd)
{
    const int iters = 200000;
    uint64_t h[40] = {0};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) // warm the clocks
        hipLaunchKernelGGL((k<OP>), dim3(256), dim3(256 * waves_per_simd), 0, 0, d_out, iters, 1.25f);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<OP>), dim3(256), dim3(256 * waves_per_simd), 0, 0, d_out, iters, 1.25f);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    const double per_instr = (double)h[0] / ((double)iters * 16.0);
    // wall clock: instructions one SIMD issued (all its waves) per nanosecond -> cycles at 2.4 GHz
    const double ns_per_simd_instr = (double)ms * 1e6 / ((double)iters * 16.0 * waves_per_simd);
    std::printf("%-34s waves/SIMD %d: %.2f s_memtime ticks per instruction per wave; wall %.3f ns per SIMD issue = %.2f cycles at 2.4 GHz\n",
                name, waves_per_simd, per_instr, ns_per_simd_instr, ns_per_simd_instr * 2.4);
}

int main()
{
    uint64_t *d_out;
    hipMalloc((void **)&d_out, 40 * sizeof(uint64_t));
    hipMemset(d_out, 0, 40 * sizeof(uint64_t));
    for (int w : {1, 2, 4}) {
        run<0>("v_add_f32", d_out, w);
        run<4>("v_mul_f32", d_out, w);
        run<7>("v_fma_f32", d_out, w);
        run<1>("v_pk_add_f32", d_out, w);
        run<5>("v_pk_add_f32 (sgpr pair, op_sel)", d_out, w);
        run<2>("v_pk_mul_f32", d_out, w);
        run<6>("v_pk_fma_f32", d_out, w);
        run<3>("v_perm_b32", d_out, w);
    }
    hipFree(d_out);
    return 0;
}
