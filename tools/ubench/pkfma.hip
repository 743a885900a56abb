// Issue-rate microbenchmark: v_fma_f32 against v_pk_fma_f32 (and v_min/v_max, v_fma_f64), 4 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int iters) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float2v p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
    float2v pa = {a, a}, pb = {b, b};
    double d0 = x0, d1 = x1, d2 = x2, d3 = x3;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x0 = __builtin_fmaf(x0, a, b), x1 = __builtin_fmaf(x1, a, b), x2 = __builtin_fmaf(x2, a, b), x3 = __builtin_fmaf(x3, a, b);
                x4 = __builtin_fmaf(x4, a, b), x5 = __builtin_fmaf(x5, a, b), x6 = __builtin_fmaf(x6, a, b), x7 = __builtin_fmaf(x7, a, b);
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(pa), "v"(pb));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(pa), "v"(pb));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(pa), "v"(pb));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(pa), "v"(pb));
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x0 = fminf(x0, a), x1 = fmaxf(x1, b), x2 = fminf(x2, a), x3 = fmaxf(x3, b);
                x4 = fminf(x4, a), x5 = fmaxf(x5, b), x6 = fminf(x6, a), x7 = fmaxf(x7, b);
                asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                d0 = __builtin_fma(d0, (double)a, (double)b), d1 = __builtin_fma(d1, (double)a, (double)b);
                d2 = __builtin_fma(d2, (double)a, (double)b), d3 = __builtin_fma(d3, (double)a, (double)b);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + (float)(d0 + d1 + d2 + d3);
}
template <int MODE>
void run(const char* name, int per_iter, double flops_per_instr) {
    float* out;
    hipMalloc(&out, 256 * 1024 * 4 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const int iters = 20000, blocks = 256 * 4;  // 4 blocks of 4 waves per CU: 4 waves/SIMD
    k<MODE><<<blocks, 256>>>(out, 0.999f, 0.001f, 100);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, 0.999f, 0.001f, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)blocks * 4 * iters * per_iter;  // wave-instructions
    printf("%-14s %8.2f ms  %7.2f G wave-instr/s  %7.1f TFLOP/s\n", name, ms, instr / ms / 1e6, instr * 64 * flops_per_instr / ms / 1e9);
    hipFree(out);
}
int main() {
    run<0>("v_fma_f32", 64, 2);
    run<1>("v_pk_fma_f32", 32, 4);
    run<2>("v_min/max_f32", 64, 1);
    run<3>("v_fma_f64", 32, 2);
    return 0;
}
