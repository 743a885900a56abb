// f32 AOV kernel: closest hit of each pixel-centre ray through the SAME traversal and intersection code
// the render kernels use (trace_closest), so its mismatch rate against the f64 oracle measures exactly
// what f32 costs the render path (tolerance stated in tests/test_primary_hits.py).
#include "device_rt.h"
#include "kernels.h"

namespace rt {

constexpr int kPBlock = 64;

template <int STACK>
__global__ __launch_bounds__(kPBlock) void primary32_kernel(DevScene sc, uint32_t width, uint32_t height,
                                                            int32_t* __restrict__ prim_id,
                                                            double* __restrict__ t_out) {
    __shared__ int lds_stack[STACK * kPBlock];
    const uint32_t idx = blockIdx.x * kPBlock + threadIdx.x;
    if (idx >= width * height) return;
    const uint32_t pi = idx % width;
    const uint32_t pj = height - idx / width - 1;
    Ray ray;
    ray.o = d3(sc.cam.origin);
    ray.d = f3(sc.cam.ll_rel) + (float)(pi + 1) * f3(sc.cam.hor_px) + (float)(pj + 1) * f3(sc.cam.ver_px);
    TraceStats st;
    const HitRec h = trace_closest<false>(sc, ray, 0.001f, lds_stack + threadIdx.x, kPBlock, st);
    prim_id[idx] = h.slot >= 0 ? sc.slot_surface[h.slot] : (h.slot == -1 ? -1 : sc.loose[-2 - h.slot].w);
    t_out[idx] = h.slot != -1 ? (double)h.t : -1.0;
}

hipError_t launch_primary32(const DevScene& sc, uint32_t width, uint32_t height, uint32_t tree_depth,
                            int32_t* d_prim_id, double* d_t, hipStream_t stream) {
    const uint32_t n = width * height;
    const dim3 grid((n + kPBlock - 1) / kPBlock), block(kPBlock);
    if (tree_depth <= 24)
        hipLaunchKernelGGL(primary32_kernel<24>, grid, block, 0, stream, sc, width, height, d_prim_id, d_t);
    else if (tree_depth <= 40)
        hipLaunchKernelGGL(primary32_kernel<40>, grid, block, 0, stream, sc, width, height, d_prim_id, d_t);
    else if (tree_depth <= 64)
        hipLaunchKernelGGL(primary32_kernel<64>, grid, block, 0, stream, sc, width, height, d_prim_id, d_t);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace rt
