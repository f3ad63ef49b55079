// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access patterns of the convolution kernels (MI355X_MICROARCH.md:
// "Other access widths ... are uncalibrated: calibrate on a known byte count in your own access pattern").
// Every kernel reads a known number of distinct bytes once out of a 1 GiB buffer (4 x the Infinity Cache):
//   k_stream      : 64 lanes x 16 B contiguous (the guide's case: FETCH_SIZE = bytes / 2)
//   k_half_lines  : groups of 4 lanes read 64 contiguous bytes, groups 128 B apart (first half of every 128-B line):
//                   the halo loads of k_conv_zr on a 32-channel tensor (one 16-channel chunk of a 128-B voxel record)
//   k_half_both   : the same, then the second halves in a second sweep (the next chunk, one phase later)
//   k_lines_256   : groups of 8 lanes read one full 128-B line, groups 256 B apart (k_conv_wgrad_zs: x pairs)
// build: hipcc --offload-arch=gfx950 -O3 scripts/proto/fetch_calib.hip -o build/proto/fetch_calib
// run:   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -o calib -- build/proto/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_stream(const float4* __restrict__ p, int64_t n, float* out) {
    float a = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = p[i];
        a += v.x + v.y + v.z + v.w;
    }
    if (a == 12345.f) *out = a;
}
// element i (16 B units): group = i / G lanes, lane in group = i % G; address = group * STRIDE16 + OFF16 + lane
template <int G, int STRIDE16>
__global__ void k_groups(const float4* __restrict__ p, int64_t ngroups, int off16, float* out) {
    float a = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < ngroups * G; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = p[(i / G) * STRIDE16 + off16 + (i % G)];
        a += v.x + v.y + v.z + v.w;
    }
    if (a == 12345.f) *out = a;
}
int main() {
    const int64_t bytes = 1ll << 30;
    float4* buf;
    float* out;
    (void)hipMalloc(&buf, bytes);
    (void)hipMalloc(&out, 4);
    (void)hipMemset(buf, 0, bytes);
    const int64_t n16 = bytes / 16;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, 0, buf, n16, out);                       // 1 GiB
        hipLaunchKernelGGL((k_groups<4, 8>), dim3(4096), dim3(256), 0, 0, buf, n16 / 8, 0, out);          // 512 MiB touched (first halves)
        hipLaunchKernelGGL((k_groups<4, 8>), dim3(4096), dim3(256), 0, 0, buf, n16 / 8, 4, out);          // 512 MiB touched (second halves)
        hipLaunchKernelGGL((k_groups<8, 16>), dim3(4096), dim3(256), 0, 0, buf, n16 / 16, 0, out);        // 512 MiB touched (every other line)
        hipLaunchKernelGGL((k_groups<2, 8>), dim3(4096), dim3(256), 0, 0, buf, n16 / 8, 0, out);          // 256 MiB touched (32 B of every line)
    }
    (void)hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
