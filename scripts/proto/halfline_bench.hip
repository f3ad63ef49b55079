// Does a read of 64 contiguous bytes out of every 128-byte line cost a 128-byte line of HBM bandwidth?
// (16-bit activation slices of a 2 x 32-channel concat buffer are exactly that pattern: profiles/r05_halfline.txt)
//   hipcc --offload-arch=gfx950 -O3 scripts/proto/halfline_bench.hip -o build/halfline_bench && build/halfline_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
// item i: 16 bytes at byte offset (i / per) * stride + (i % per) * 16; per = 16-byte pieces read per record
__global__ __launch_bounds__(256) void k_read(const unsigned char* __restrict__ p, int64_t nitems, int per, int stride, unsigned* out) {
    unsigned acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nitems; i += (int64_t)gridDim.x * 256) {
        const u4 v = __builtin_nontemporal_load(reinterpret_cast<const u4*>(p + (i / per) * stride + (i % per) * 16));
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *out = acc;
}
__global__ __launch_bounds__(256) void k_write(unsigned char* __restrict__ p, int64_t nitems, int per, int stride) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nitems; i += (int64_t)gridDim.x * 256) {
        const u4 v = {(unsigned)i, 1u, 2u, 3u};
        __builtin_nontemporal_store(v, reinterpret_cast<u4*>(p + (i / per) * stride + (i % per) * 16));
    }
}
int main() {
    const int64_t bytes = (int64_t)2 << 30;   // 2 GiB buffer: far beyond L2 + Infinity Cache
    unsigned char* p;
    unsigned* out;
    hipMalloc(&p, bytes);
    hipMalloc(&out, 4);
    hipMemset(p, 1, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    struct { const char* name; int per, stride; } cases[] = {
        {"dense 128 of 128", 8, 128}, {"first 64 of 128", 4, 128}, {"first 64 of 256", 4, 256}, {"first 128 of 256", 8, 256},
        {"first 32 of 128", 2, 128}, {"first 32 of 64", 2, 64}};
    for (int w = 0; w < 2; ++w)
        for (auto& c : cases) {
            const int64_t nrec = bytes / c.stride, nitems = nrec * c.per;
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                if (w) hipLaunchKernelGGL(k_write, dim3(256 * 16), dim3(256), 0, 0, p, nitems, c.per, c.stride);
                else hipLaunchKernelGGL(k_read, dim3(256 * 16), dim3(256), 0, 0, p, nitems, c.per, c.stride, out);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("%-5s %-18s useful %7.1f MB in %7.3f ms = %6.2f TB/s useful, %6.2f TB/s if whole 128-byte lines move\n", w ? "write" : "read",
                   c.name, nitems * 16 / 1e6, best, nitems * 16 / 1e9 / best, (double)nrec * (c.stride >= 128 ? ((c.per * 16 + 127) / 128 * 128) : c.stride) / 1e9 / best);
        }
    return 0;
}
