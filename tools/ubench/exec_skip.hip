// Does a VALU instruction cost less when part of the wave is masked off?  Lanes >= `live` leave the loop at once;
// the rest run chains of v_min_f32 (slow class) or v_fma_f32 (fast class).  4 waves/SIMD as in valu_rates.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int iters, int live, int stride) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const int lane = threadIdx.x & 63;
    const bool on = stride == 1 ? lane < live : (lane % stride == 0 && lane / stride < live);
    if (on) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (MODE == 0) {
                    x0 = __builtin_fmaf(x0, a, b), x1 = __builtin_fmaf(x1, a, b), x2 = __builtin_fmaf(x2, a, b), x3 = __builtin_fmaf(x3, a, b);
                    x4 = __builtin_fmaf(x4, a, b), x5 = __builtin_fmaf(x5, a, b), x6 = __builtin_fmaf(x6, a, b), x7 = __builtin_fmaf(x7, a, b);
                } else {
                    x0 = fminf(x0, a), x1 = fmaxf(x1, b), x2 = fminf(x2, a), x3 = fmaxf(x3, b);
                    x4 = fminf(x4, a), x5 = fmaxf(x5, b), x6 = fminf(x6, a), x7 = fmaxf(x7, b);
                    asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
                }
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <int MODE>
void run(const char* name, int live, int stride) {
    float* out;
    hipMalloc(&out, 256 * 1024 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const int iters = 20000, blocks = 256 * 4;
    k<MODE><<<blocks, 256>>>(out, 0.999f, 0.001f, 100, live, stride);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, 0.999f, 0.001f, iters, live, stride);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)blocks * 4 * iters * 64;
    printf("%-10s live %2d stride %d: %8.2f ms  %7.2f G wave-instr/s  %.2f clocks/instr/SIMD at 2.4 GHz\n", name, live, stride, ms,
           instr / ms / 1e6, 1024 * 2.4e9 / (instr / ms * 1e3));
    hipFree(out);
}
int main() {
    for (int live : {64, 48, 32, 16, 8, 1}) run<0>("v_fma_f32", live, 1);
    for (int live : {64, 48, 32, 16, 8, 1}) run<1>("v_min/max", live, 1);
    run<1>("v_min/max", 16, 4);  // 16 live lanes spread over the wave
    run<0>("v_fma_f32", 16, 4);
    run<1>("v_min/max", 32, 2);
    return 0;
}
