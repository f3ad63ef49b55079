// augment.hip -- on-device augmentations so the training loop never leaves HBM.
//
// Reference: torch_em/transform/augmentation.py -- KorniaAugmentationPipeline.forward (:203-223) with the default
// 3-D pipeline of H/V/D flips (:254-258) and RandomElasticDeformationStacked (:11-88), whose arithmetic lives in the
// third-party package kornia (setup.py:12, unpinned, NOT in this image): kornia.geometry.transform.elastic_transform2d
// = Gaussian-blur the 2-channel noise field (63x63 kernel, zero border), scale by alpha, add to the normalised
// identity grid of create_meshgrid (linspace(-1,1)), clamp to [-1,1], F.grid_sample(align_corners=False,
// padding_mode="reflection", bilinear | nearest).  Restated from kornia's published algorithm; parity is pinned
// against a torch-CPU restatement (oracle/augment_ref.py), not against kornia itself ("parity unpinned").
//
// All three kernels are pure data movement / small stencils: HBM-bound, one read and one write per element.
#include "tem_common.h"

// flags: device int [N][3] (z, y, x), non-zero = flip that axis of sample n.  planes = channels per sample.
__global__ __launch_bounds__(256) void k_flip3d(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                 const int* __restrict__ flags, int planes, int D, int H, int W,
                                                 int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        int64_t r = i / W;
        const int y = (int)(r % H);
        r /= H;
        const int z = (int)(r % D);
        r /= D;  // r = n * planes + c
        const int n = (int)(r / planes);
        const int sz = flags[n * 3 + 0] ? D - 1 - z : z;
        const int sy = flags[n * 3 + 1] ? H - 1 - y : y;
        const int sx = flags[n * 3 + 2] ? W - 1 - x : x;
        dst[i] = src[((r * D + sz) * H + sy) * W + sx];
    }
}

extern "C" int tem_flip3d(const void* src, void* dst, const int* flags_dev, int N, int planes, int D, int H, int W,
                          tem_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    TEM_REQUIRE(src && dst && flags_dev && src != dst, "tem_flip3d: null pointer or in-place call");
    TEM_REQUIRE(N > 0 && planes > 0 && D > 0 && H > 0 && W > 0, "tem_flip3d: bad sizes");
    const int64_t total = (int64_t)N * planes * D * H * W;
    hipLaunchKernelGGL(k_flip3d, dim3(tem_grid_1d(total, 256)), dim3(256), 0, s, (const uint32_t*)src, (uint32_t*)dst,
                       flags_dev, planes, D, H, W, total);
    TEM_CHECK_LAUNCH("tem_flip3d");
    return TEM_OK;
}

// displacement field: disp[0] = alpha0 * (noise[0] (*) G(sigma1)), disp[1] = alpha1 * (noise[1] (*) G(sigma0)),
// G = normalised outer product of 1-D Gaussians of length ks, zero ("constant") border.  gk: device [2][ks] 1-D kernels.
__global__ __launch_bounds__(256) void k_elastic_field(const float* __restrict__ noise, const float* __restrict__ gk,
                                                        int ks, int H, int W, float alpha0, float alpha1,
                                                        float* __restrict__ disp) {
    const int c = blockIdx.y;
    const float* g = gk + (c == 0 ? ks : 0);  // channel 0 uses sigma[1], channel 1 uses sigma[0]
    const int half = ks / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W) return;
    const int y = i / W, x = i % W;
    float acc = 0.f;
    for (int dy = 0; dy < ks; ++dy) {
        const int yy = y + dy - half;
        if (yy < 0 || yy >= H) continue;
        float row = 0.f;
        for (int dx = 0; dx < ks; ++dx) {
            const int xx = x + dx - half;
            if (xx >= 0 && xx < W) row = fmaf(g[dx], noise[((int64_t)c * H + yy) * W + xx], row);
        }
        acc = fmaf(g[dy], row, acc);
    }
    disp[((int64_t)c * H + y) * W + x] = acc * (c == 0 ? alpha0 : alpha1);
}

extern "C" int tem_elastic_field(const float* noise, const float* gauss1d, int ksize, int H, int W, float alpha0,
                                 float alpha1, float* disp, tem_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    TEM_REQUIRE(noise && gauss1d && disp, "tem_elastic_field: null pointer");
    TEM_REQUIRE(ksize > 0 && (ksize & 1) && H > 0 && W > 0, "tem_elastic_field: kernel size must be odd");
    hipLaunchKernelGGL(k_elastic_field, dim3((unsigned)tem_cdiv((int64_t)H * W, 256), 2), dim3(256), 0, s, noise, gauss1d,
                       ksize, H, W, alpha0, alpha1, disp);
    TEM_CHECK_LAUNCH("tem_elastic_field");
    return TEM_OK;
}

__device__ __forceinline__ float reflect_coord(float c, float lo2, float hi2) {
    // torch grid_sample reflect_coordinates(in, twice_low, twice_high)
    if (lo2 == hi2) return 0.f;
    const float mn = lo2 * 0.5f, span = (hi2 - lo2) * 0.5f;
    c = fabsf(c - mn);
    const float extra = fmodf(c, span);
    const int flips = (int)floorf(c / span);
    return (flips & 1) ? span - extra + mn : extra + mn;
}
__device__ __forceinline__ float src_index(float g, int size) {
    // align_corners=False unnormalise, reflection padding, clip (torch grid_sampler_compute_source_index)
    float c = ((g + 1.f) * size - 1.f) * 0.5f;
    c = reflect_coord(c, -1.f, 2.f * size - 1.f);
    return fminf(fmaxf(c, 0.f), (float)(size - 1));
}

// the same 2-D warp applied to every [H][W] plane: out[p][y][x] = sample(in[p], clamp(grid(y,x) + disp(y,x)))
__global__ __launch_bounds__(256) void k_elastic_warp(const float* __restrict__ src, const float* __restrict__ disp,
                                                       float* __restrict__ dst, int64_t planes, int H, int W, int nearest) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W) return;
    const int y = i / W, x = i % W;
    // create_meshgrid(normalized_coordinates=True): linspace(-1, 1, n)
    float gx = (W > 1 ? -1.f + 2.f * x / (float)(W - 1) : -1.f) + disp[(int64_t)y * W + x];
    float gy = (H > 1 ? -1.f + 2.f * y / (float)(H - 1) : -1.f) + disp[((int64_t)H + y) * W + x];
    gx = fminf(fmaxf(gx, -1.f), 1.f);
    gy = fminf(fmaxf(gy, -1.f), 1.f);
    const float ix = src_index(gx, W), iy = src_index(gy, H);
    const int64_t hw = (int64_t)H * W;
    if (nearest) {
        const int nx = (int)nearbyintf(ix), ny = (int)nearbyintf(iy);
        const bool in = nx >= 0 && nx < W && ny >= 0 && ny < H;
        for (int64_t p = blockIdx.y; p < planes; p += gridDim.y)
            dst[p * hw + i] = in ? src[p * hw + (int64_t)ny * W + nx] : 0.f;
        return;
    }
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    const bool bx0 = x0 >= 0 && x0 < W, bx1 = x1 >= 0 && x1 < W, by0 = y0 >= 0 && y0 < H, by1 = y1 >= 0 && y1 < H;
    for (int64_t p = blockIdx.y; p < planes; p += gridDim.y) {
        const float* s = src + p * hw;
        float v = 0.f;
        // torch's order: nw, ne, sw, se
        if (by0 && bx0) v += s[(int64_t)y0 * W + x0] * (wx0 * wy0);
        if (by0 && bx1) v += s[(int64_t)y0 * W + x1] * (wx1 * wy0);
        if (by1 && bx0) v += s[(int64_t)y1 * W + x0] * (wx0 * wy1);
        if (by1 && bx1) v += s[(int64_t)y1 * W + x1] * (wx1 * wy1);
        dst[p * hw + i] = v;
    }
}

extern "C" int tem_elastic_warp2d(const float* src, const float* disp, float* dst, int64_t planes, int H, int W,
                                  int nearest, tem_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    TEM_REQUIRE(src && disp && dst && src != dst, "tem_elastic_warp2d: null pointer or in-place call");
    TEM_REQUIRE(planes > 0 && H > 0 && W > 0, "tem_elastic_warp2d: bad sizes");
    const unsigned gy = (unsigned)(planes < 64 ? planes : 64);
    hipLaunchKernelGGL(k_elastic_warp, dim3((unsigned)tem_cdiv((int64_t)H * W, 256), gy), dim3(256), 0, s, src, disp, dst,
                       planes, H, W, nearest);
    TEM_CHECK_LAUNCH("tem_elastic_warp2d");
    return TEM_OK;
}

// ---------------------------------------------------------------------------
// 3-D affine warp: kornia RandomAffine3D / RandomRotation3D (reference transform/augmentation.py:235,240 configures them
// with degrees=(90,90,90), scale=(0.0,1.1)).  kornia draws the parameters, composes a 4x4 matrix about the volume
// centre and resamples with warp_affine3d; here the host composes the matrix from the drawn (or injected) parameters
// and hands over its INVERSE in voxel coordinates: mat[n] = 3x4 row-major, (sx, sy, sz) = A * (x, y, z, 1) maps an
// output voxel to its source position.  One pass: trilinear for images, nearest for labels, zeros outside
// (kornia's padding_mode="zeros").  kornia is not in this image: parity unpinned (oracle/augment_ref.py:affine_warp3d
// restates the resampling with F.affine_grid / F.grid_sample).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_affine_warp3d(const float* __restrict__ src, const float* __restrict__ mat,
                                                       float* __restrict__ dst, int planes, int D, int H, int W,
                                                       int nearest, int64_t total) {
    const int64_t dhw = (int64_t)D * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        int64_t r = i / W;
        const int y = (int)(r % H);
        r /= H;
        const int z = (int)(r % D);
        const int n = (int)(r / D);
        const float* A = mat + n * 12;
        const float sx = fmaf(A[0], x, fmaf(A[1], y, fmaf(A[2], z, A[3])));
        const float sy = fmaf(A[4], x, fmaf(A[5], y, fmaf(A[6], z, A[7])));
        const float sz = fmaf(A[8], x, fmaf(A[9], y, fmaf(A[10], z, A[11])));
        const float* s0 = src + (int64_t)n * planes * dhw;
        float* d0 = dst + (int64_t)n * planes * dhw + ((int64_t)z * H + y) * W + x;
        if (nearest) {
            const int nx = (int)nearbyintf(sx), ny = (int)nearbyintf(sy), nz = (int)nearbyintf(sz);
            const bool in = nx >= 0 && nx < W && ny >= 0 && ny < H && nz >= 0 && nz < D;
            const int64_t off = in ? ((int64_t)nz * H + ny) * W + nx : 0;
            for (int c = 0; c < planes; ++c) d0[c * dhw] = in ? s0[c * dhw + off] : 0.f;
            continue;
        }
        const float fx = floorf(sx), fy = floorf(sy), fz = floorf(sz);
        const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
        const float wx = sx - fx, wy = sy - fy, wz = sz - fz;
        // 8 corners: clamped addresses + zero weights outside, so the loads of one voxel are independent
        float wgt[8];
        int64_t off[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int cx = x0 + (k & 1), cy = y0 + ((k >> 1) & 1), cz = z0 + (k >> 2);
            const bool in = cx >= 0 && cx < W && cy >= 0 && cy < H && cz >= 0 && cz < D;
            wgt[k] = in ? ((k & 1) ? wx : 1.f - wx) * (((k >> 1) & 1) ? wy : 1.f - wy) * ((k >> 2) ? wz : 1.f - wz) : 0.f;
            off[k] = ((int64_t)min(max(cz, 0), D - 1) * H + min(max(cy, 0), H - 1)) * W + min(max(cx, 0), W - 1);
        }
        for (int c = 0; c < planes; ++c) {
            const float* s = s0 + c * dhw;
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) v = fmaf(s[off[k]], wgt[k], v);
            d0[c * dhw] = v;
        }
    }
}

extern "C" int tem_affine_warp3d(const float* src, const float* mat_dev, float* dst, int N, int planes, int D, int H, int W,
                                 int nearest, tem_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    TEM_REQUIRE(src && mat_dev && dst && src != dst, "tem_affine_warp3d: null pointer or in-place call");
    TEM_REQUIRE(N > 0 && planes > 0 && D > 0 && H > 0 && W > 0, "tem_affine_warp3d: bad sizes");
    const int64_t total = (int64_t)N * D * H * W;
    hipLaunchKernelGGL(k_affine_warp3d, dim3(tem_grid_1d(total, 256)), dim3(256), 0, s, src, mat_dev, dst, planes, D, H, W,
                       nearest, total);
    TEM_CHECK_LAUNCH("tem_affine_warp3d");
    return TEM_OK;
}
