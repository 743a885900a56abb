// Issue rates of single VALU instructions on gfx950 (wave64, 4 waves/SIMD, 8 independent chains per lane).
#include <hip/hip_runtime.h>
#include <cstdio>
#define OP8(INS)                                                                                       \
    asm volatile(INS " %0, %0, %8, %9\n" INS " %1, %1, %8, %9\n" INS " %2, %2, %8, %9\n" INS " %3, %3, %8, %9\n" \
                 INS " %4, %4, %8, %9\n" INS " %5, %5, %8, %9\n" INS " %6, %6, %8, %9\n" INS " %7, %7, %8, %9"   \
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b))
#define OP8_2(INS)                                                                             \
    asm volatile(INS " %0, %0, %8\n" INS " %1, %1, %8\n" INS " %2, %2, %8\n" INS " %3, %3, %8\n"         \
                 INS " %4, %4, %8\n" INS " %5, %5, %8\n" INS " %6, %6, %8\n" INS " %7, %7, %8"           \
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a))
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int iters) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    unsigned long long mask = 0x5555555555555555ull ^ (unsigned long long)iters, m0 = 0, m1 = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) OP8("v_fma_f32");
            if (MODE == 1) OP8_2("v_min_f32");
            if (MODE == 2) OP8("v_min3_f32");
            if (MODE == 3) OP8_2("v_mul_f32");
            if (MODE == 4) OP8_2("v_add_f32");
            if (MODE == 5) OP8_2("v_add_u32");
            if (MODE == 6) OP8_2("v_xor_b32");
            if (MODE == 7) OP8_2("v_mul_lo_u32");
            if (MODE == 8)
                asm volatile("v_cndmask_b32_e64 %0, %0, %8, %9\nv_cndmask_b32_e64 %1, %1, %8, %9\nv_cndmask_b32_e64 %2, %2, %8, %9\n"
                             "v_cndmask_b32_e64 %3, %3, %8, %9\nv_cndmask_b32_e64 %4, %4, %8, %9\nv_cndmask_b32_e64 %5, %5, %8, %9\n"
                             "v_cndmask_b32_e64 %6, %6, %8, %9\nv_cndmask_b32_e64 %7, %7, %8, %9"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "s"(mask));
            if (MODE == 9)
                asm volatile("v_cmp_lt_f32_e64 %0, %2, %3\nv_cmp_lt_f32_e64 %1, %3, %4\nv_cmp_lt_f32_e64 %0, %4, %5\nv_cmp_lt_f32_e64 %1, %5, %6\n"
                             "v_cmp_lt_f32_e64 %0, %6, %7\nv_cmp_lt_f32_e64 %1, %7, %8\nv_cmp_lt_f32_e64 %0, %8, %9\nv_cmp_lt_f32_e64 %1, %9, %2"
                             : "=s"(m0), "=s"(m1) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7));
            if (MODE == 10) OP8("v_med3_f32");
            if (MODE == 11) OP8_2("v_max_f32");
            if (MODE == 12) OP8_2("v_lshlrev_b32");
            if (MODE == 13) OP8("v_mad_u32_u24");
            if (MODE == 14) OP8("v_bfe_u32");
            if (MODE == 15) OP8("v_alignbit_b32");
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (float)(m0 + m1);
}
template <int MODE>
void run(const char* name) {
    float* out;
    (void)hipMalloc(&out, 256 * 1024 * 4 * sizeof(float));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    const int iters = 20000, blocks = 256 * 4;
    k<MODE><<<blocks, 256>>>(out, 0.999f, 0.001f, 100);
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, 0.999f, 0.001f, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)blocks * 4 * iters * 64;
    printf("%-26s %8.2f ms  %7.2f G wave-instr/s  %5.2f clocks per instruction per SIMD (2.4 GHz)\n", name, ms, instr / ms / 1e6,
           1024.0 * 2.4e9 / (instr / ms * 1e3));
    (void)hipFree(out);
}
int main() {
    run<0>("v_fma_f32");
    run<1>("v_min_f32");
    run<2>("v_min3_f32");
    run<3>("v_mul_f32");
    run<4>("v_add_f32");
    run<5>("v_add_u32");
    run<6>("v_xor_b32");
    run<7>("v_mul_lo_u32");
    run<8>("v_cndmask_b32 (sgpr mask)");
    run<9>("v_cmp_lt_f32 -> sgpr");
    run<10>("v_med3_f32");
    run<11>("v_max_f32");
    run<12>("v_lshlrev_b32");
    run<13>("v_mad_u32_u24");
    run<14>("v_bfe_u32");
    run<15>("v_alignbit_b32");
    return 0;
}
