// memtime_cal.hip -- what does s_memtime count on gfx950?  Spin for 2^26 ticks, time with HIP events; then the same
// while every SIMD of the chip streams MFMAs (power-limited clocks), to see whether the tick rate follows the shader clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
__global__ void spin(unsigned long long ticks, unsigned long long* out) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long t1;
    do { t1 = __builtin_amdgcn_s_memtime(); } while (t1 - t0 < ticks);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
// every wave: N MFMAs back to back; reports the ticks wave 0 of block 0 needed
__global__ __launch_bounds__(256) void mfma_all(int n, float c, unsigned long long* out, int randomize) {
    floatx16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    half8_t x, y;
    for (int i = 0; i < 8; ++i) {
        x[i] = (_Float16)(randomize ? c * (float)((threadIdx.x * 7 + i * 13 + blockIdx.x) % 97) / 97.f - 0.5f : 0.f);
        y[i] = (_Float16)(randomize ? c * (float)((threadIdx.x * 11 + i * 5 + blockIdx.x) % 89) / 89.f - 0.5f : 0.f);
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < n; ++it) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a3, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (a0[0] + a1[1] + a2[2] + a3[3] == 1.2345f) out[1] = 1;
}
int main() {
    unsigned long long* d; hipMalloc(&d, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms; unsigned long long h[2];
    for (int r = 0; r < 2; ++r) {
        hipEventRecord(e0); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, 1ull << 26, d); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("spin: %llu ticks in %.3f ms -> %.1f MHz\n", h[0], ms, h[0] / ms / 1e3);
    }
    for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd)
        for (int rnd = 0; rnd < 2; ++rnd)
            for (int r = 0; r < 2; ++r) {
                const int n = 1 << 16;   // 4 * 65536 MFMAs per wave
                hipEventRecord(e0); hipLaunchKernelGGL(mfma_all, dim3(256 * waves_per_simd), dim3(256), 0, 0, n, 1.0f, d, rnd); hipEventRecord(e1);
                hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
                const double mf = 4.0 * n;
                printf("mfma on all CUs, %d wave/SIMD, %s operands: %.3f ms, %llu ticks (%.1f MHz), %.2f ticks/MFMA, %.1f ns/MFMA, chip %.0f TF\n",
                       waves_per_simd, rnd ? "random" : "zero", ms, h[0], h[0] / ms / 1e3, h[0] / mf, ms * 1e6 / mf,
                       mf * 32768.0 * 1024 * waves_per_simd / (ms * 1e-3) / 1e12);
            }
    return 0;
}
