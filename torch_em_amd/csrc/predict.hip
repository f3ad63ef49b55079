// predict.hip -- block gather / scatter of tiled inference with a halo.
//
// Reference: torch_em/util/prediction.py -- `_load_block` (:98-142: clip the (block + halo) box to the volume, then
// np.pad(mode="reflect") back to the full size) and the write-back of `predict_with_halo` (:270-302: crop the halo,
// zero outside the mask, store into the output volume).  The reference does both on the host with numpy per block
// and ships every block over PCIe twice; here the whole volume stays in HBM (288 GB) and a block is one gather
// kernel in, the forward kernels, and one scatter kernel out.  Pure data movement: HBM-bound, bit-exact.
#include "tem_common.h"

struct Box3 {
    int s0[3], len[3], padl[3], out[3];  // clipped segment start / length, left padding, padded block size (z, y, x)
};

// numpy "reflect" (no edge repeat) of index j relative to a segment of length L
__device__ __forceinline__ int reflect_idx(int j, int L) {
    if (L == 1) return 0;
    const int period = 2 * (L - 1);
    j %= period;
    if (j < 0) j += period;
    return j < L ? j : period - j;
}

__global__ __launch_bounds__(256) void k_block_load_reflect(const float* __restrict__ src, float* __restrict__ dst, int C,
                                                            int D, int H, int W, Box3 b) {
    const int64_t bv = (int64_t)b.out[0] * b.out[1] * b.out[2];
    const int64_t total = bv * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % b.out[2]);
        int64_t r = i / b.out[2];
        const int y = (int)(r % b.out[1]);
        r /= b.out[1];
        const int z = (int)(r % b.out[0]);
        const int c = (int)(r / b.out[0]);
        const int sz = b.s0[0] + reflect_idx(z - b.padl[0], b.len[0]);
        const int sy = b.s0[1] + reflect_idx(y - b.padl[1], b.len[1]);
        const int sx = b.s0[2] + reflect_idx(x - b.padl[2], b.len[2]);
        dst[i] = src[(((int64_t)c * D + sz) * H + sy) * W + sx];
    }
}

extern "C" int tem_block_load_reflect(const float* src, float* dst, int C, int D, int H, int W, const int* seg_start,
                                      const int* seg_len, const int* pad_left, const int* out_shape, tem_stream_t stream) {
    TEM_REQUIRE(src && dst && seg_start && seg_len && pad_left && out_shape, "tem_block_load_reflect: null pointer");
    Box3 b;
    const int dims[3] = {D, H, W};
    for (int a = 0; a < 3; ++a) {
        b.s0[a] = seg_start[a];
        b.len[a] = seg_len[a];
        b.padl[a] = pad_left[a];
        b.out[a] = out_shape[a];
        TEM_REQUIRE(b.len[a] > 0 && b.s0[a] >= 0 && b.s0[a] + b.len[a] <= dims[a] && b.out[a] > 0 && b.padl[a] >= 0,
                    "tem_block_load_reflect: bad box on axis %d", a);
    }
    TEM_REQUIRE(C > 0, "tem_block_load_reflect: bad channel count");
    const int64_t total = (int64_t)C * b.out[0] * b.out[1] * b.out[2];
    hipLaunchKernelGGL(k_block_load_reflect, dim3(tem_grid_1d(total, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, C,
                       D, H, W, b);
    TEM_CHECK_LAUNCH("tem_block_load_reflect");
    return TEM_OK;
}

// out[c][begin + i] = pred[c][inner + i] (0 where mask == 0) for i in the block's inner box `size`
__global__ __launch_bounds__(256) void k_block_store_inner(const float* __restrict__ pred, int pd, int ph, int pw,
                                                           float* __restrict__ out, int D, int H, int W, int C,
                                                           const unsigned char* __restrict__ mask, int i0, int i1, int i2,
                                                           int b0, int b1, int b2, int n0, int n1, int n2) {
    const int64_t total = (int64_t)C * n0 * n1 * n2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % n2);
        int64_t r = i / n2;
        const int y = (int)(r % n1);
        r /= n1;
        const int z = (int)(r % n0);
        const int c = (int)(r / n0);
        const int64_t vo = ((int64_t)(b0 + z) * H + (b1 + y)) * W + (b2 + x);
        float v = pred[(((int64_t)c * pd + (i0 + z)) * ph + (i1 + y)) * pw + (i2 + x)];
        if (mask && !mask[vo]) v = 0.f;
        out[(int64_t)c * D * H * W + vo] = v;
    }
}

extern "C" int tem_block_store_inner(const float* pred, const int* pred_shape, float* out, int C, int D, int H, int W,
                                     const unsigned char* mask, const int* inner_start, const int* out_start,
                                     const int* size, tem_stream_t stream) {
    TEM_REQUIRE(pred && pred_shape && out && inner_start && out_start && size, "tem_block_store_inner: null pointer");
    const int dims[3] = {D, H, W};
    for (int a = 0; a < 3; ++a)
        TEM_REQUIRE(size[a] > 0 && inner_start[a] >= 0 && inner_start[a] + size[a] <= pred_shape[a] && out_start[a] >= 0 &&
                        out_start[a] + size[a] <= dims[a],
                    "tem_block_store_inner: bad box on axis %d", a);
    const int64_t total = (int64_t)C * size[0] * size[1] * size[2];
    hipLaunchKernelGGL(k_block_store_inner, dim3(tem_grid_1d(total, 256)), dim3(256), 0, (hipStream_t)stream, pred,
                       pred_shape[0], pred_shape[1], pred_shape[2], out, D, H, W, C, mask, inner_start[0], inner_start[1],
                       inner_start[2], out_start[0], out_start[1], out_start[2], size[0], size[1], size[2]);
    TEM_CHECK_LAUNCH("tem_block_store_inner");
    return TEM_OK;
}

// AccumulateChannels (reference model/unet.py:15-44): out = cat([x[:, i0:i1], reduce(x[:, c0:c1], dim=1)]) with reduce in
// {mean 0, min 1, max 2}; export-time post-processing of affinity predictions.  x read through (sn, sc, sv) strides.
__global__ __launch_bounds__(256) void k_accumulate_channels(const float* __restrict__ x, int64_t sn, int64_t sc, int64_t sv,
                                                             float* __restrict__ out, int N, int64_t V, int i0, int i1,
                                                             int c0, int c1, int mode) {
    const int ninv = i1 - i0, cout = ninv + 1;
    const int64_t total = (int64_t)N * V;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / V);
        const int64_t v = i % V;
        const float* p = x + n * sn + v * sv;
        for (int c = 0; c < ninv; ++c) out[((int64_t)n * cout + c) * V + v] = p[(int64_t)(i0 + c) * sc];
        float acc = p[(int64_t)c0 * sc];
        for (int c = c0 + 1; c < c1; ++c) {
            const float t = p[(int64_t)c * sc];
            acc = mode == 0 ? acc + t : (mode == 1 ? fminf(acc, t) : fmaxf(acc, t));
        }
        if (mode == 0) acc /= (float)(c1 - c0);
        out[((int64_t)n * cout + ninv) * V + v] = acc;
    }
}

extern "C" int tem_accumulate_channels(const float* x, int64_t sn, int64_t sc, int64_t sv, float* out, int N, int C,
                                       int64_t V, int i0, int i1, int c0, int c1, int mode, tem_stream_t stream) {
    TEM_REQUIRE(x && out && N > 0 && V > 0, "tem_accumulate_channels: bad arguments");
    TEM_REQUIRE(0 <= i0 && i0 <= i1 && i1 <= C && 0 <= c0 && c0 < c1 && c1 <= C && mode >= 0 && mode <= 2,
                "tem_accumulate_channels: bad channel ranges");
    hipLaunchKernelGGL(k_accumulate_channels, dim3(tem_grid_1d((int64_t)N * V, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       sn, sc, sv, out, N, V, i0, i1, c0, c1, mode);
    TEM_CHECK_LAUNCH("tem_accumulate_channels");
    return TEM_OK;
}
