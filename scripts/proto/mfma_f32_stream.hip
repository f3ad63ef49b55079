// mfma_f32_stream.hip -- what does a REGISTER-ONLY stream of v_mfma_f32_32x32x2_f32 sustain on gfx950, as a function of the number
// of independent accumulators per wave (dependent distance) and of waves per SIMD?  (The exact-fp32 forward kernels run 2 or 4
// accumulators per wave with 3 / 2 waves per SIMD and reach 0.77-0.81 of the 157.3 TFLOP/s peak; k_conv_wgrad_tr<4>, 7 accumulators
// in one multiplying wave per SIMD, 0.91.)
// build: hipcc --offload-arch=gfx950 -O3 -o build/mfma_f32_stream scripts/proto/mfma_f32_stream.hip ; run: build/mfma_f32_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void k_stream(float* out, int iters, float a0, float b0) {
    floatx16 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
static int run(int wgs_per_cu, float* out, float a, float b) {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount, iters = 2000;
    const int grid = ncu * wgs_per_cu;   // 256 threads = 4 waves = one per SIMD per workgroup
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_stream<NACC>, dim3(grid), dim3(256), 0, 0, out, 10, a, b);
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_stream<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, a, b);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double flops = (double)grid * 4 * iters * 8 * NACC * 4096.0;
    printf("accumulators per wave %d, waves per SIMD %d: %.1f TFLOP/s (%.3f of 157.3)  [%s operands]\n", NACC, wgs_per_cu, flops / best / 1e9,
           flops / best / 1e9 / 157.3, a == 0.f ? "zero" : "random-ish");
    return 0;
}

int main() {
    float* out;
    CK(hipMalloc(&out, (size_t)256 * 16 * 256 * 4));
    for (int z = 0; z < 2; ++z) {
        const float a = z ? 0.f : 1.37f, b = z ? 0.f : -0.73f;
        for (int w = 1; w <= 3; ++w) {
            if (run<1>(w, out, a, b) || run<2>(w, out, a, b) || run<4>(w, out, a, b) || run<7>(w, out, a, b)) return 1;
        }
    }
    return 0;
}
