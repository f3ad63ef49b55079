// pool_upsample.hip -- nn.MaxPool3d(factor) and F.interpolate(trilinear, align_corners=False)
// for NDHWC fp32 tensors (reference model/unet.py:300-302,645 and :455-458).
// HBM-bound streaming kernels: one workgroup per output row (n, z, y), threads
// sweep (x, channel-quad) so every access is a coalesced 16-byte vector.
#include "tem_common.h"
#include "tem_act.h"

template <int VEC>
struct VecT;
template <>
struct VecT<4> {
    typedef float4 type;
};
template <>
struct VecT<1> {
    typedef float type;
};

#ifndef TEM_POOL_NT
#define TEM_POOL_NT 0
#endif
template <int VEC, typename T>
__device__ __forceinline__ void ld_vec(const T* p, float (&v)[VEC]) {
    if constexpr (VEC == 8) {   // 16-bit tensors: 8 channels = one 16-byte access, as 4 fp32 channels are
        act_ld8(p, v);
    } else if constexpr (VEC == 4) {
        const float4 t = act_ld4(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        v[0] = act_ld1(p);
    }
}
template <int VEC, typename T>
__device__ __forceinline__ void st_vec(T* p, const float (&v)[VEC]) {
    if constexpr (VEC == 8) {
        act_st8(p, v);
    } else if constexpr (VEC == 4) {
#if TEM_POOL_NT
        act_st4_nt(p, make_float4(v[0], v[1], v[2], v[3]));
#else
        act_st4(p, make_float4(v[0], v[1], v[2], v[3]));
#endif
    } else {
        act_st1(p, v[0]);
    }
}

// st: storage type of the tensors (a vector of 4 elements is 16 bytes of fp32, 8 bytes of fp16 / bf16)
static inline bool vec4_ok(int C, std::initializer_list<const void*> ptrs, std::initializer_list<int64_t> lds, int st = 0) {
    if (C % 4) return false;
    for (const void* p : ptrs)
        if (p && ((uintptr_t)p % tem_st_align4(st))) return false;
    for (int64_t l : lds)
        if (l % 4) return false;
    return true;
}

// 16-bit tensors whose rows allow 16-byte accesses of 8 channels (what a load / store instruction moves per lane decides the
// rate of these kernels: with 8-byte accesses the 16-bit max-pool backward of round 5's first cut ran SLOWER than fp32)
static inline bool vec8_ok(int C, std::initializer_list<const void*> ptrs, std::initializer_list<int64_t> lds, int st) {
    if (!st || C % 8) return false;
    for (const void* p : ptrs)
        if (p && ((uintptr_t)p % 16)) return false;
    for (int64_t l : lds)
        if (l % 8) return false;
    return true;
}

// ---------------------------------------------------------------------------
// max pool
// ---------------------------------------------------------------------------
// stat (optional, [N][Do * Ho][C][2], 256 % (C / VEC) == 0): the block's (sum, sum of squares) of every channel of its output
// row -- the first stage of the statistics of the norm that reads the pooled tensor next (tem_maxpool3d_fwd_stats)
template <int VEC, typename T>
__global__ __launch_bounds__(256) void k_maxpool_fwd(const T* __restrict__ x, int64_t x_ld, T* __restrict__ y,
                                                     int64_t y_ld, int D, int H, int W, int C, int fz, int fy, int fx,
                                                     float* __restrict__ stat) {
    extern __shared__ float pst[];   // stat: [256 / cq rows][C][2]
    const int Do = D / fz, Ho = H / fy, Wo = W / fx;
    const int cq = C / VEC;
    float s0[VEC], s1[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) s0[j] = s1[j] = 0.f;
    int row = blockIdx.x;  // (n, zo, yo)
    const int yo = row % Ho;
    row /= Ho;
    const int zo = row % Do;
    const int n = row / Do;
    const int nwin = fz * fy * fx;
    if (nwin <= 8) {  // the window as 8 independent loads (see k_maxpool_bwd)
        for (int i = threadIdx.x; i < Wo * cq; i += 256) {
            const int xo = i / cq, c0 = (i % cq) * VEC;
            float t[8][VEC], m[VEC];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int kk = k < nwin ? k : 0;
                const int dx = kk % fx, dy = (kk / fx) % fy, dz = kk / (fx * fy);
                ld_vec<VEC>(x + ((((int64_t)n * D + zo * fz + dz) * H + yo * fy + dy) * W + xo * fx + dx) * x_ld + c0, t[k]);
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j) m[j] = -INFINITY;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k < nwin) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j)
                        if (t[k][j] > m[j] || t[k][j] != t[k][j]) m[j] = t[k][j];
                }
            st_vec<VEC>(y + ((((int64_t)n * Do + zo) * Ho + yo) * Wo + xo) * y_ld + c0, m);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                s0[j] += m[j];
                s1[j] = fmaf(m[j], m[j], s1[j]);
            }
        }
    } else {
        for (int i = threadIdx.x; i < Wo * cq; i += 256) {
            const int xo = i / cq, c0 = (i % cq) * VEC;
            float m[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) m[j] = -INFINITY;
            for (int dz = 0; dz < fz; ++dz)
                for (int dy = 0; dy < fy; ++dy)
                    for (int dx = 0; dx < fx; ++dx) {
                        int64_t v = (((int64_t)n * D + zo * fz + dz) * H + yo * fy + dy) * W + xo * fx + dx;
                        float t[VEC];
                        ld_vec<VEC>(x + v * x_ld + c0, t);
#pragma unroll
                        for (int j = 0; j < VEC; ++j)
                            if (t[j] > m[j] || t[j] != t[j]) m[j] = t[j];
                    }
            int64_t vo = (((int64_t)n * Do + zo) * Ho + yo) * Wo + xo;
            st_vec<VEC>(y + vo * y_ld + c0, m);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                s0[j] += m[j];
                s1[j] = fmaf(m[j], m[j], s1[j]);
            }
        }
    }
    if (stat) {   // 256 % cq == 0: a thread keeps its channels over its trips; rows of threads share them
        const int q = threadIdx.x % cq, r = threadIdx.x / cq, rows = 256 / cq;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            pst[(r * C + q * VEC + j) * 2 + 0] = s0[j];
            pst[(r * C + q * VEC + j) * 2 + 1] = s1[j];
        }
        __syncthreads();
        for (int t = threadIdx.x; t < 2 * C; t += 256) {
            float a = 0.f;
            for (int k = 0; k < rows; ++k) a += pst[k * 2 * C + t];
            stat[(int64_t)blockIdx.x * 2 * C + t] = a;
        }
    }
}

template <int VEC, typename T>
__global__ __launch_bounds__(256) void k_maxpool_bwd(const T* __restrict__ gy, int64_t gy_ld,
                                                     const T* __restrict__ x, int64_t x_ld,
                                                     const T* __restrict__ gskip, int64_t gskip_ld, int relu_mask,
                                                     T* __restrict__ gx, int64_t gx_ld, int D, int H, int W, int C,
                                                     int fz, int fy, int fx, const float* __restrict__ gcoef,
                                                     int64_t gcoef_ld, const float* __restrict__ ycoef,
                                                     unsigned* __restrict__ amax) {
    const int Do = D / fz, Ho = H / fy, Wo = W / fx;
    const int cq = C / VEC;
    float amx = 0.f;
    int row = blockIdx.x;
    const int yo = row % Ho;
    row /= Ho;
    const int zo = row % Do;
    const int n = row / Do;
    const int nwin = fz * fy * fx;
    if (nwin <= 8) {
        // windows of <= 8 voxels (every pooling factor of the reference nets): the window is read ONCE into registers,
        // all loads independent (the runtime-bounded triple loops issued them one by one and read x twice)
        for (int i = threadIdx.x; i < Wo * cq; i += 256) {
            const int xo = i / cq, c0 = (i % cq) * VEC;
            int64_t vk[8];
            float t[8][VEC], o[8][VEC];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int kk = k < nwin ? k : 0;
                const int dx = kk % fx, dy = (kk / fx) % fy, dz = kk / (fx * fy);
                vk[k] = (((int64_t)n * D + zo * fz + dz) * H + yo * fy + dy) * W + xo * fx + dx;
                ld_vec<VEC>(x + vk[k] * x_ld + c0, t[k]);
                if (gskip) {
                    ld_vec<VEC>(gskip + vk[k] * gskip_ld + c0, o[k]);
                } else {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) o[k][j] = 0.f;
                }
            }
            float m[VEC];
            int am[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                m[j] = -INFINITY;
                am[j] = 0;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k < nwin) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j)
                        if (t[k][j] > m[j] || t[k][j] != t[k][j]) {
                            m[j] = t[k][j];
                            am[j] = k;
                        }
                }
            const int64_t vo = (((int64_t)n * Do + zo) * Ho + yo) * Wo + xo;
            float g[VEC];
            ld_vec<VEC>(gy + vo * gy_ld + c0, g);
            float4 kg[VEC], ky[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                if (gcoef) kg[j] = *reinterpret_cast<const float4*>(gcoef + (int64_t)n * gcoef_ld + (c0 + j) * 4);
                if (ycoef) {
                    ky[j] = *reinterpret_cast<const float4*>(ycoef + ((int64_t)n * C + c0 + j) * 4);
                    g[j] = ky[j].x * g[j] - ky[j].y - (m[j] - ky[j].w) * ky[j].z;  // see the general path below
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k < nwin) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        float v = o[k][j];
                        if (gcoef) v = kg[j].x * v - kg[j].y - (t[k][j] - kg[j].w) * kg[j].z;
                        if (am[j] == k) v += g[j];
                        if (relu_mask && !(t[k][j] > 0.f)) v = 0.f;
                        o[k][j] = v;
                        amx = __builtin_fmaxf(amx, __builtin_fabsf(v));
                    }
                    st_vec<VEC>(gx + vk[k] * gx_ld + c0, o[k]);
                }
        }
        if (amax) tem_amax_commit(amax, amx);
        return;
    }
    for (int i = threadIdx.x; i < Wo * cq; i += 256) {
        const int xo = i / cq, c0 = (i % cq) * VEC;
        float m[VEC];
        int am[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            m[j] = -INFINITY;
            am[j] = 0;
        }
        int k = 0;
        for (int dz = 0; dz < fz; ++dz)
            for (int dy = 0; dy < fy; ++dy)
                for (int dx = 0; dx < fx; ++dx, ++k) {
                    int64_t v = (((int64_t)n * D + zo * fz + dz) * H + yo * fy + dy) * W + xo * fx + dx;
                    float t[VEC];
                    ld_vec<VEC>(x + v * x_ld + c0, t);
#pragma unroll
                    for (int j = 0; j < VEC; ++j)
                        if (t[j] > m[j] || t[j] != t[j]) {
                            m[j] = t[j];
                            am[j] = k;
                        }
                }
        int64_t vo = (((int64_t)n * Do + zo) * Ho + yo) * Wo + xo;
        float g[VEC];
        ld_vec<VEC>(gy + vo * gy_ld + c0, g);
        if (ycoef) {
            // gy is the RAW data gradient behind the norm whose input is the POOLED tensor (first norm of the next
            // level's block): its backward is applied here -- the pooled value is the maximum m[j] just recomputed
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float4 kc = *reinterpret_cast<const float4*>(ycoef + ((int64_t)n * C + c0 + j) * 4);
                g[j] = kc.x * g[j] - kc.y - (m[j] - kc.w) * kc.z;
            }
        }
        k = 0;
        for (int dz = 0; dz < fz; ++dz)
            for (int dy = 0; dy < fy; ++dy)
                for (int dx = 0; dx < fx; ++dx, ++k) {
                    int64_t v = (((int64_t)n * D + zo * fz + dz) * H + yo * fy + dy) * W + xo * fx + dx;
                    float o[VEC];
                    if (gskip) {
                        ld_vec<VEC>(gskip + v * gskip_ld + c0, o);
                    } else {
#pragma unroll
                        for (int j = 0; j < VEC; ++j) o[j] = 0.f;
                    }
                    float t[VEC];
                    if (relu_mask || gcoef) ld_vec<VEC>(x + v * x_ld + c0, t);
                    if (gcoef) {
                        // gskip is the RAW data gradient of the decoder conv behind the concat norm: apply that norm's
                        // backward here (x is its input: the skip tensor) instead of in a pass of its own
#pragma unroll
                        for (int j = 0; j < VEC; ++j) {
                            const float4 kc = *reinterpret_cast<const float4*>(gcoef + (int64_t)n * gcoef_ld + (c0 + j) * 4);
                            o[j] = kc.x * o[j] - kc.y - (t[j] - kc.w) * kc.z;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        if (am[j] == k) o[j] += g[j];
                        if (relu_mask && !(t[j] > 0.f)) o[j] = 0.f;
                        amx = __builtin_fmaxf(amx, __builtin_fabsf(o[j]));
                    }
                    st_vec<VEC>(gx + v * gx_ld + c0, o);
                }
    }
    if (amax) tem_amax_commit(amax, amx);
}

static int maxpool3d_fwd_impl(const void* x, int64_t x_ld, void* y, int64_t y_ld, int N, int D, int H, int W, int C, int fz,
                              int fy, int fx, float* stat, int st, tem_stream_t stream) {
    TEM_REQUIRE(x && y && N > 0 && C > 0 && x_ld >= C && y_ld >= C, "tem_maxpool3d_fwd: bad arguments");
    TEM_REQUIRE(st >= 0 && st <= 2, "tem_maxpool3d_fwd: unknown storage type %d", st);
    TEM_REQUIRE(fz > 0 && fy > 0 && fx > 0 && D % fz == 0 && H % fy == 0 && W % fx == 0,
                "tem_maxpool3d_fwd: shape (%d,%d,%d) not divisible by factors (%d,%d,%d)", D, H, W, fz, fy, fx);
    int64_t rows = (int64_t)N * (D / fz) * (H / fy);
    TEM_REQUIRE(rows < (1ll << 31), "tem_maxpool3d_fwd: too many rows");
    const bool v4 = vec4_ok(C, {x, y}, {x_ld, y_ld}, st);
    const bool v8 = vec8_ok(C, {x, y}, {x_ld, y_ld}, st) && 256 % (C / 8) == 0 && tem_option(TEM_OPT_POOL_VEC8);
    const int cq = v8 ? C / 8 : v4 ? C / 4 : C;
    TEM_REQUIRE(!stat || (cq <= 256 && 256 % cq == 0), "tem_maxpool3d_fwd_stats: tem_maxpool3d_fwd_stat_blocks() == 0 for C = %d", C);
    const size_t lds = stat ? (size_t)(256 / cq) * C * 2 * sizeof(float) : 0;
    if (v8) {
        TEM_ST16_SWITCH(st, T, hipLaunchKernelGGL((k_maxpool_fwd<8, T>), dim3((unsigned)rows), dim3(256), lds, (hipStream_t)stream,
                                                  (const T*)x, x_ld, (T*)y, y_ld, D, H, W, C, fz, fy, fx, stat));
        TEM_CHECK_LAUNCH("tem_maxpool3d_fwd");
        return TEM_OK;
    }
    TEM_ST_SWITCH(st, T, {
        if (v4)
            hipLaunchKernelGGL((k_maxpool_fwd<4, T>), dim3((unsigned)rows), dim3(256), lds, (hipStream_t)stream, (const T*)x, x_ld,
                               (T*)y, y_ld, D, H, W, C, fz, fy, fx, stat);
        else
            hipLaunchKernelGGL((k_maxpool_fwd<1, T>), dim3((unsigned)rows), dim3(256), lds, (hipStream_t)stream, (const T*)x, x_ld,
                               (T*)y, y_ld, D, H, W, C, fz, fy, fx, stat);
    });
    TEM_CHECK_LAUNCH("tem_maxpool3d_fwd");
    return TEM_OK;
}

extern "C" int tem_maxpool3d_fwd(const float* x, int64_t x_ld, float* y, int64_t y_ld, int N, int D, int H, int W,
                                 int C, int fz, int fy, int fx, tem_stream_t stream) {
    return maxpool3d_fwd_impl(x, x_ld, y, y_ld, N, D, H, W, C, fz, fy, fx, nullptr, 0, stream);
}

// statistics partial rows per sample of tem_maxpool3d_fwd_stats: one per output row (zo, yo); 0: this channel count cannot
extern "C" int64_t tem_maxpool3d_fwd_stat_blocks(int D, int H, int C, int fz, int fy) {
    const int cq = C % 4 == 0 ? C / 4 : C;
    if (fz <= 0 || fy <= 0 || cq > 256 || 256 % cq) return 0;
    return (int64_t)(D / fz) * (H / fy);
}

// tem_maxpool3d_fwd that also writes stat_part [N][stat_blocks][C][2]: per output row the (sum, sum of squares) of every
// channel -- tem_norm_finalize_partials turns them into the statistics of the norm that reads y next (the first norm of the
// next encoder level: one pass over the pooled tensor and one launch less)
extern "C" int tem_maxpool3d_fwd_stats(const float* x, int64_t x_ld, float* y, int64_t y_ld, int N, int D, int H, int W, int C,
                                       int fz, int fy, int fx, float* stat_part, int64_t stat_blocks, tem_stream_t stream) {
    return tem_maxpool3d_fwd_st(x, x_ld, y, y_ld, N, D, H, W, C, fz, fy, fx, stat_part, stat_blocks, TEM_ST_F32, stream);
}

// tem_maxpool3d_fwd / _stats (stat_part != NULL) for tensors of storage type st
extern "C" int tem_maxpool3d_fwd_st(const void* x, int64_t x_ld, void* y, int64_t y_ld, int N, int D, int H, int W, int C,
                                    int fz, int fy, int fx, float* stat_part, int64_t stat_blocks, int st, tem_stream_t stream) {
    if (stat_part) {
        TEM_REQUIRE(stat_blocks > 0 && fz > 0 && fy > 0 && stat_blocks == tem_maxpool3d_fwd_stat_blocks(D, H, C, fz, fy),
                    "tem_maxpool3d_fwd_stats: stat_blocks must be tem_maxpool3d_fwd_stat_blocks() (and > 0)");
        TEM_REQUIRE(st >= 0 && st <= 2 && (C % 4 == 0) == vec4_ok(C, {x, y}, {x_ld, y_ld}, st),
                    "tem_maxpool3d_fwd_stats: x / y must be aligned to 4 elements with ld %% 4 == 0");
    }
    return maxpool3d_fwd_impl(x, x_ld, y, y_ld, N, D, H, W, C, fz, fy, fx, stat_part, st, stream);
}

// k_maxpool_bwd<8, T> for 16-bit tensors and windows of FZ x 2 x 2 voxels, written for REGISTERS: the generic kernel
// unpacks the whole window (2 x 8 voxels x 8 channels of floats + 16 coefficient rows = 256 VGPRs, two waves per SIMD: a
// streaming kernel that waits for its one batch of loads with a quarter of the chip's wave slots -- 305 us for the
// 128^3 level of cfg 2, 2.7 TB/s).  Here the 16-byte words stay packed; a thread walks its 8 channels in four PAIRS
// (unpack 2 x NW values, arg-max, norm backward, mask, re-pack into the word it came from), the coefficient rows of the
// block's sample sit in LDS.  Same arithmetic, same order of operations per element: bit-identical to the generic kernel.
template <int FZ, typename T>
__global__ __launch_bounds__(256, 3) void k_maxpool_bwd16(const T* __restrict__ gy, int64_t gy_ld, const T* __restrict__ x, int64_t x_ld,
                                                       const T* __restrict__ gskip, int64_t gskip_ld, int relu_mask,
                                                       T* __restrict__ gx, int64_t gx_ld, int D, int H, int W, int C,
                                                       const float* __restrict__ gcoef, int64_t gcoef_ld,
                                                       const float* __restrict__ ycoef, unsigned* __restrict__ amax) {
    static_assert(sizeof(T) == 2, "16-bit storage only");
    constexpr int NW = FZ * 4;
    extern __shared__ __attribute__((aligned(16))) float4 mp_coef[];   // [C] skip-norm rows, then [C] pooled-norm rows
    const int Do = D / FZ, Ho = H / 2, Wo = W / 2;
    const int cq = C / 8;
    int row = blockIdx.x;
    const int yo = row % Ho;
    row /= Ho;
    const int zo = row % Do;
    const int n = row / Do;
    if (gcoef)
        for (int c = threadIdx.x; c < C; c += 256) mp_coef[c] = *reinterpret_cast<const float4*>(gcoef + (int64_t)n * gcoef_ld + c * 4);
    if (ycoef)
        for (int c = threadIdx.x; c < C; c += 256) mp_coef[C + c] = *reinterpret_cast<const float4*>(ycoef + ((int64_t)n * C + c) * 4);
    if (gcoef || ycoef) __syncthreads();
    float amx = 0.f;
    for (int i = threadIdx.x; i < Wo * cq; i += 256) {
        const int xo = i / cq, c0 = (i % cq) * 8;
        unsigned tw[NW][4], ow[NW][4];
        // the window's first voxel per thread, its other voxels at wave-uniform distances
        const int64_t v0 = (((int64_t)n * D + zo * FZ) * H + yo * 2) * W + xo * 2;
        const T* const xb = x + v0 * x_ld + c0;
        const T* const sb = gskip ? gskip + v0 * gskip_ld + c0 : nullptr;
        T* const ob = gx + v0 * gx_ld + c0;
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const int64_t dv = ((int64_t)(k >> 2) * H + ((k >> 1) & 1)) * W + (k & 1);
            const act_u4 a = *reinterpret_cast<const act_u4*>(xb + dv * x_ld);
            tw[k][0] = a.x, tw[k][1] = a.y, tw[k][2] = a.z, tw[k][3] = a.w;
            act_u4 b = {0u, 0u, 0u, 0u};
            if (gskip) b = *reinterpret_cast<const act_u4*>(sb + dv * gskip_ld);
            ow[k][0] = b.x, ow[k][1] = b.y, ow[k][2] = b.z, ow[k][3] = b.w;
        }
        const int64_t vo = (((int64_t)n * Do + zo) * Ho + yo) * Wo + xo;
        const act_u4 gq = *reinterpret_cast<const act_u4*>(gy + vo * gy_ld + c0);
        const unsigned gw[4] = {gq.x, gq.y, gq.z, gq.w};
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
            float r[2][NW];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float t[NW];
#pragma unroll
                for (int k = 0; k < NW; ++k) t[k] = h ? act_hi<T>(tw[k][jp]) : act_lo<T>(tw[k][jp]);
                float m = -INFINITY;
                int am = 0;
#pragma unroll
                for (int k = 0; k < NW; ++k)
                    if (t[k] > m || t[k] != t[k]) {
                        m = t[k];
                        am = k;
                    }
                float g = h ? act_hi<T>(gw[jp]) : act_lo<T>(gw[jp]);
                const int c = c0 + jp * 2 + h;
                if (ycoef) {
                    const float4 ky = mp_coef[C + c];
                    g = ky.x * g - ky.y - (m - ky.w) * ky.z;
                }
                float4 kg = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gcoef) kg = mp_coef[c];
#pragma unroll
                for (int k = 0; k < NW; ++k) {
                    float v = h ? act_hi<T>(ow[k][jp]) : act_lo<T>(ow[k][jp]);
                    if (gcoef) v = kg.x * v - kg.y - (t[k] - kg.w) * kg.z;
                    if (am == k) v += g;
                    if (relu_mask && !(t[k] > 0.f)) v = 0.f;
                    amx = __builtin_fmaxf(amx, __builtin_fabsf(v));
                    r[h][k] = v;
                }
            }
#pragma unroll
            for (int k = 0; k < NW; ++k) ow[k][jp] = act_pk<T>(r[0][k], r[1][k]);   // one rounding per stored value
        }
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const int64_t dv = ((int64_t)(k >> 2) * H + ((k >> 1) & 1)) * W + (k & 1);
            *reinterpret_cast<act_u4*>(ob + dv * gx_ld) = act_u4{ow[k][0], ow[k][1], ow[k][2], ow[k][3]};
        }
    }
    if (amax) tem_amax_commit(amax, amx);
}

static int maxpool3d_bwd_impl(const void* gy, int64_t gy_ld, const void* x, int64_t x_ld, const void* gskip,
                              int64_t gskip_ld, int relu_mask, void* gx, int64_t gx_ld, int N, int D, int H, int W,
                              int C, int fz, int fy, int fx, const float* gcoef, int64_t gcoef_ld, const float* ycoef,
                              unsigned* amax, int st, tem_stream_t stream) {
    TEM_REQUIRE(gy && x && gx && N > 0 && C > 0 && x_ld >= C && gy_ld >= C && gx_ld >= C,
                "tem_maxpool3d_bwd: bad arguments");
    TEM_REQUIRE(st >= 0 && st <= 2, "tem_maxpool3d_bwd: unknown storage type %d", st);
    TEM_REQUIRE(fz > 0 && fy > 0 && fx > 0 && D % fz == 0 && H % fy == 0 && W % fx == 0,
                "tem_maxpool3d_bwd: shape (%d,%d,%d) not divisible by factors (%d,%d,%d)", D, H, W, fz, fy, fx);
    int64_t rows = (int64_t)N * (D / fz) * (H / fy);
    TEM_REQUIRE(rows < (1ll << 31), "tem_maxpool3d_bwd: too many rows");
    const bool v4 = vec4_ok(C, {gy, x, gskip, gx}, {gy_ld, x_ld, gskip ? gskip_ld : 0, gx_ld}, st);
    if (vec8_ok(C, {gy, x, gskip, gx}, {gy_ld, x_ld, gskip ? gskip_ld : 0, gx_ld}, st) && tem_option(TEM_OPT_POOL_VEC8)) {
        if (fy == 2 && fx == 2 && (fz == 1 || fz == 2) && C <= 2048 && tem_option(TEM_OPT_POOL_VEC8) >= 2) {
            const size_t ldsb = (gcoef || ycoef) ? (size_t)2 * C * sizeof(float4) : 0;
            TEM_ST16_SWITCH(st, T, {
                if (fz == 2)
                    hipLaunchKernelGGL((k_maxpool_bwd16<2, T>), dim3((unsigned)rows), dim3(256), ldsb, (hipStream_t)stream, (const T*)gy,
                                       gy_ld, (const T*)x, x_ld, (const T*)gskip, gskip_ld, relu_mask, (T*)gx, gx_ld, D, H, W, C, gcoef,
                                       gcoef_ld, ycoef, amax);
                else
                    hipLaunchKernelGGL((k_maxpool_bwd16<1, T>), dim3((unsigned)rows), dim3(256), ldsb, (hipStream_t)stream, (const T*)gy,
                                       gy_ld, (const T*)x, x_ld, (const T*)gskip, gskip_ld, relu_mask, (T*)gx, gx_ld, D, H, W, C, gcoef,
                                       gcoef_ld, ycoef, amax);
            });
            TEM_CHECK_LAUNCH("tem_maxpool3d_bwd");
            return TEM_OK;
        }
        TEM_ST16_SWITCH(st, T, hipLaunchKernelGGL((k_maxpool_bwd<8, T>), dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream,
                                                  (const T*)gy, gy_ld, (const T*)x, x_ld, (const T*)gskip, gskip_ld, relu_mask, (T*)gx,
                                                  gx_ld, D, H, W, C, fz, fy, fx, gcoef, gcoef_ld, ycoef, amax));
        TEM_CHECK_LAUNCH("tem_maxpool3d_bwd");
        return TEM_OK;
    }
    TEM_ST_SWITCH(st, T, {
        if (v4)
            hipLaunchKernelGGL((k_maxpool_bwd<4, T>), dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const T*)gy, gy_ld,
                               (const T*)x, x_ld, (const T*)gskip, gskip_ld, relu_mask, (T*)gx, gx_ld, D, H, W, C, fz, fy, fx, gcoef,
                               gcoef_ld, ycoef, amax);
        else
            hipLaunchKernelGGL((k_maxpool_bwd<1, T>), dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const T*)gy, gy_ld,
                               (const T*)x, x_ld, (const T*)gskip, gskip_ld, relu_mask, (T*)gx, gx_ld, D, H, W, C, fz, fy, fx, gcoef,
                               gcoef_ld, ycoef, amax);
    });
    TEM_CHECK_LAUNCH("tem_maxpool3d_bwd");
    return TEM_OK;
}

extern "C" int tem_maxpool3d_bwd(const float* gy, int64_t gy_ld, const float* x, int64_t x_ld, const float* gskip,
                                 int64_t gskip_ld, int relu_mask, float* gx, int64_t gx_ld, int N, int D, int H, int W,
                                 int C, int fz, int fy, int fx, tem_stream_t stream) {
    return maxpool3d_bwd_impl(gy, gy_ld, x, x_ld, gskip, gskip_ld, relu_mask, gx, gx_ld, N, D, H, W, C, fz, fy, fx, nullptr,
                              0, nullptr, nullptr, 0, stream);
}

// tem_maxpool3d_bwd whose skip gradient is still the RAW data gradient of the decoder conv behind the concat norm: that
// norm's backward (coefficients from tem_norm_bwd_coef, rows of gcoef_ld floats per sample, this tensor's channels
// first) is applied on the fly -- x, its input, is the tensor this kernel reads anyway.
// ycoef (optional, [N][C][4] dense): gy is likewise raw -- the gradient behind the norm whose input is the POOLED tensor
// (the first norm of the next level's block); its backward uses the maximum this kernel recomputes.
extern "C" int tem_maxpool3d_bwd_norm(const float* gy, int64_t gy_ld, const float* x, int64_t x_ld, const float* gskip,
                                      int64_t gskip_ld, int relu_mask, float* gx, int64_t gx_ld, int N, int D, int H,
                                      int W, int C, int fz, int fy, int fx, const float* gcoef, int64_t gcoef_ld,
                                      const float* ycoef, tem_stream_t stream) {
    TEM_REQUIRE(gcoef || ycoef, "tem_maxpool3d_bwd_norm: no coefficients given");
    return tem_maxpool3d_bwd_st(gy, gy_ld, x, x_ld, gskip, gskip_ld, relu_mask, gx, gx_ld, N, D, H, W, C, fz, fy, fx, gcoef,
                                gcoef_ld, ycoef, nullptr, TEM_ST_F32, stream);
}

// tem_maxpool3d_bwd / _bwd_norm (gcoef / ycoef != NULL) for tensors of storage type st; out_amax (optional): the device word
// that receives max |gx| (bit pattern, integer atomicMax) for the weight gradient that reads gx next
extern "C" int tem_maxpool3d_bwd_st(const void* gy, int64_t gy_ld, const void* x, int64_t x_ld, const void* gskip,
                                    int64_t gskip_ld, int relu_mask, void* gx, int64_t gx_ld, int N, int D, int H, int W, int C,
                                    int fz, int fy, int fx, const float* gcoef, int64_t gcoef_ld, const float* ycoef,
                                    unsigned* out_amax, int st, tem_stream_t stream) {
    TEM_REQUIRE(!gcoef || (gskip && gcoef_ld >= 4 * C && ((uintptr_t)gcoef % 16 == 0) && gcoef_ld % 4 == 0),
                "tem_maxpool3d_bwd_norm: bad skip coefficient arguments");
    TEM_REQUIRE(!ycoef || ((uintptr_t)ycoef % 16 == 0), "tem_maxpool3d_bwd_norm: ycoef must be 16-byte aligned");
    return maxpool3d_bwd_impl(gy, gy_ld, x, x_ld, gskip, gskip_ld, relu_mask, gx, gx_ld, N, D, H, W, C, fz, fy, fx, gcoef,
                              gcoef_ld, ycoef, out_amax, st, stream);
}

// ---------------------------------------------------------------------------
// linear upsampling, align_corners=False, integer factor f per axis.
// ATen: src = (dst + 0.5) / f - 0.5, clamped at 0; i0 = floor(src); i1 = i0 + (i0 < in-1);
// l1 = src - i0; l0 = 1 - l1   (area_pixel_compute_source_index)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void lin_src(int o, int f, int in, int& i0, int& i1, float& l0, float& l1) {
    float scale = 1.0f / (float)f;
    float src = scale * ((float)o + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int)src;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + ((i0 < in - 1) ? 1 : 0);
    l1 = src - (float)i0;
    l0 = 1.f - l1;
}

template <int VEC, typename T>
__global__ __launch_bounds__(256) void k_upsample_fwd(const T* __restrict__ x, int64_t x_ld, T* __restrict__ y,
                                                      int64_t y_ld, int D, int H, int W, int C, int fz, int fy, int fx) {
    const int Do = D * fz, Ho = H * fy, Wo = W * fx;
    const int cq = C / VEC;
    int row = blockIdx.x;  // (n, zo, yo)
    const int yo = row % Ho;
    row /= Ho;
    const int zo = row % Do;
    const int n = row / Do;
    int z0, z1, y0, y1;
    float lz0, lz1, ly0, ly1;
    lin_src(zo, fz, D, z0, z1, lz0, lz1);
    lin_src(yo, fy, H, y0, y1, ly0, ly1);
    const int64_t r00 = (((int64_t)n * D + z0) * H + y0) * W, r01 = (((int64_t)n * D + z0) * H + y1) * W;
    const int64_t r10 = (((int64_t)n * D + z1) * H + y0) * W, r11 = (((int64_t)n * D + z1) * H + y1) * W;
    for (int i = threadIdx.x; i < Wo * cq; i += 256) {
        const int xo = i / cq, c0 = (i % cq) * VEC;
        int x0, x1;
        float lx0, lx1;
        lin_src(xo, fx, W, x0, x1, lx0, lx1);
        float a[VEC], b[VEC], c[VEC], d[VEC], e[VEC], f[VEC], g[VEC], h[VEC], o[VEC];
        ld_vec<VEC>(x + (r00 + x0) * x_ld + c0, a);
        ld_vec<VEC>(x + (r00 + x1) * x_ld + c0, b);
        ld_vec<VEC>(x + (r01 + x0) * x_ld + c0, c);
        ld_vec<VEC>(x + (r01 + x1) * x_ld + c0, d);
        ld_vec<VEC>(x + (r10 + x0) * x_ld + c0, e);
        ld_vec<VEC>(x + (r10 + x1) * x_ld + c0, f);
        ld_vec<VEC>(x + (r11 + x0) * x_ld + c0, g);
        ld_vec<VEC>(x + (r11 + x1) * x_ld + c0, h);
#pragma unroll
        for (int j = 0; j < VEC; ++j)
            o[j] = lz0 * (ly0 * (lx0 * a[j] + lx1 * b[j]) + ly1 * (lx0 * c[j] + lx1 * d[j])) +
                   lz1 * (ly0 * (lx0 * e[j] + lx1 * f[j]) + ly1 * (lx0 * g[j] + lx1 * h[j]));
        int64_t vo = (((int64_t)n * Do + zo) * Ho + yo) * Wo + xo;
        st_vec<VEC>(y + vo * y_ld + c0, o);
    }
}

// weight with which output index o reads input index i along one axis
__device__ __forceinline__ float lin_w(int o, int f, int in, int i) {
    int i0, i1;
    float l0, l1;
    lin_src(o, f, in, i0, i1, l0, l1);
    float w = 0.f;
    if (i0 == i) w += l0;
    if (i1 == i) w += l1;
    return w;
}

// the fine positions lo..hi that read coarse index i with a non-zero weight, compacted into (index, weight) lists of 4
// (factor <= 2); unused entries repeat a valid index with weight 0.  Returns the count.
__device__ __forceinline__ int upb_taps(int lo, int hi, int f, int in, int i, int (&idx)[4], float (&wt)[4]) {
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        idx[k] = lo;
        wt[k] = 0.f;
    }
    for (int o = lo; o <= hi; ++o) {
        const float w = lin_w(o, f, in, i);
        if (w != 0.f && cnt < 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k == cnt) {
                    idx[k] = o;
                    wt[k] = w;
                }
            ++cnt;
        }
    }
    return cnt;
}

// (U^T U)[i][i+d], d = -1, 0, 1, and (U^T 1)[i] of the 1-D interpolation operator U (lin_w)
__device__ __forceinline__ void utu_axis(int i, int f, int in, float (&a)[3], float& s) {
    a[0] = a[1] = a[2] = 0.f;
    s = 0.f;
    const int olo = max(0, f * i - f), ohi = min(in * f - 1, f * i + 2 * f - 1);
    for (int o = olo; o <= ohi; ++o) {
        const float w = lin_w(o, f, in, i);
        if (w == 0.f) continue;
        s += w;
#pragma unroll
        for (int d = -1; d <= 1; ++d)
            if (i + d >= 0 && i + d < in) a[d + 1] = fmaf(w, lin_w(o, f, in, i + d), a[d + 1]);
    }
}

template <int VEC, typename T>
__global__ __launch_bounds__(256) void k_upsample_bwd(const T* __restrict__ gy, int64_t gy_ld,
                                                      T* __restrict__ gx, int64_t gx_ld, int D, int H, int W, int C,
                                                      int fz, int fy, int fx, const T* __restrict__ u, int64_t u_ld,
                                                      const float* __restrict__ ncoef, int64_t ncoef_ld) {
    const int Do = D * fz, Ho = H * fy, Wo = W * fx;
    const int cq = C / VEC;
    int row = blockIdx.x;  // (n, z, y) of the INPUT grid
    const int yi = row % H;
    row /= H;
    const int zi = row % D;
    const int n = row / D;
    const int zlo = max(0, fz * zi - fz), zhi = min(Do - 1, fz * zi + 2 * fz - 1);
    const int ylo = max(0, fy * yi - fy), yhi = min(Ho - 1, fy * yi + 2 * fy - 1);
    const bool fast = fz <= 2 && fy <= 2 && fx <= 2;  // <= 4 contributing positions per axis
    int zidx[4], yidx[4];
    float zw[4], yw[4];
    int nzt = 0;
    if (fast) {
        nzt = upb_taps(zlo, zhi, fz, D, zi, zidx, zw);
        upb_taps(ylo, yhi, fy, H, yi, yidx, yw);
    }
    float az[3] = {0.f, 0.f, 0.f}, ay[3] = {0.f, 0.f, 0.f}, sz = 0.f, sy = 0.f;  // the row's (z, y) stencil weights
    if (ncoef) {
        utu_axis(zi, fz, D, az, sz);
        utu_axis(yi, fy, H, ay, sy);
    }
    for (int i = threadIdx.x; i < W * cq; i += 256) {
        const int xi = i / cq, c0 = (i % cq) * VEC;
        const int xlo = max(0, fx * xi - fx), xhi = min(Wo - 1, fx * xi + 2 * fx - 1);
        float acc[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
        if (fast) {
            // the (<= 4) contributing fine positions per axis as compact (index, weight) lists: 16 independent loads per
            // z tap without a branch in between (the weight test per gathered element serialised the loads: this kernel
            // ran at the latency of ~90 dependent loads per item)
            int xidx[4];
            float xw[4];
            upb_taps(xlo, xhi, fx, W, xi, xidx, xw);
            for (int kz = 0; kz < nzt; ++kz) {
                const int64_t rz = ((int64_t)n * Do + zidx[kz]) * Ho;
                float t[4][4][VEC];
#pragma unroll
                for (int ky = 0; ky < 4; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 4; ++kx) ld_vec<VEC>(gy + ((rz + yidx[ky]) * Wo + xidx[kx]) * gy_ld + c0, t[ky][kx]);
#pragma unroll
                for (int ky = 0; ky < 4; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 4; ++kx) {
                        const float w = zw[kz] * yw[ky] * xw[kx];
#pragma unroll
                        for (int j = 0; j < VEC; ++j) acc[j] = fmaf(w, t[ky][kx][j], acc[j]);
                    }
            }
        } else
        for (int zo = zlo; zo <= zhi; ++zo) {
            float wz = lin_w(zo, fz, D, zi);
            if (wz == 0.f) continue;
            for (int yo = ylo; yo <= yhi; ++yo) {
                float wy = lin_w(yo, fy, H, yi);
                if (wy == 0.f) continue;
                const int64_t r = (((int64_t)n * Do + zo) * Ho + yo) * Wo;
                for (int xo = xlo; xo <= xhi; ++xo) {
                    float wx = lin_w(xo, fx, W, xi);
                    if (wx == 0.f) continue;
                    float t[VEC];
                    ld_vec<VEC>(gy + (r + xo) * gy_ld + c0, t);
                    const float w = wz * wy * wx;
#pragma unroll
                    for (int j = 0; j < VEC; ++j) acc[j] = fmaf(w, t[j], acc[j]);
                }
            }
        }
        int64_t v = (((int64_t)n * D + zi) * H + yi) * W + xi;
        if (ncoef) {
            // gy is the RAW data gradient g of the conv behind a norm whose input was upsample(u); the norm backward
            // g' = a*g - m1 - (x - mean)*m2r is linear and x = U u, so  U^T g' = a*U^T g - m1*U^T 1 - m2r*(U^T U u - mean*U^T 1):
            // a 27-point stencil on the LOW-RESOLUTION tensor u replaces a pass over the fine tensors
            float ax[3], sx;
            utu_axis(xi, fx, W, ax, sx);
            const float S = sz * sy * sx;
            float q[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) q[j] = 0.f;
            // all 27 taps unconditionally (out-of-range neighbours: clamped address, weight 0): independent loads
#pragma unroll
            for (int dz = -1; dz <= 1; ++dz) {
                const int zc = min(max(zi + dz, 0), D - 1);
                float t[3][3][VEC];
#pragma unroll
                for (int dy = -1; dy <= 1; ++dy) {
                    const int yc = min(max(yi + dy, 0), H - 1);
#pragma unroll
                    for (int dx = -1; dx <= 1; ++dx) {
                        const int xc = min(max(xi + dx, 0), W - 1);
                        ld_vec<VEC>(u + ((((int64_t)n * D + zc) * H + yc) * W + xc) * u_ld + c0, t[dy + 1][dx + 1]);
                    }
                }
#pragma unroll
                for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                    for (int dx = -1; dx <= 1; ++dx) {
                        const float wq = az[dz + 1] * ay[dy + 1] * ax[dx + 1];
#pragma unroll
                        for (int j = 0; j < VEC; ++j) q[j] = fmaf(wq, t[dy + 1][dx + 1][j], q[j]);
                    }
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float4 kc = *reinterpret_cast<const float4*>(ncoef + (int64_t)n * ncoef_ld + (c0 + j) * 4);
                acc[j] = kc.x * acc[j] - kc.y * S - kc.z * (q[j] - kc.w * S);
            }
        }
        st_vec<VEC>(gx + v * gx_ld + c0, acc);
    }
}

// ---------------------------------------------------------------------------
// Factor-2 fast paths (fy = fx = 2, fz = 1 or 2: every sampler of the U-Nets, reference model/unet.py:455-458).
// The generic kernels above gather per output element: 8 coarse loads per fine voxel forward, 64 fine loads per coarse
// voxel backward -- both bound by the texture-address path (16 B x 64 lanes per load instruction), 3-4x off the HBM
// time.  Here a thread owns a 2x2x2 block of outputs and evaluates the interpolation separably (x, then y, then z) in
// registers: 27 loads per 8 outputs forward, 216 per 8 backward (27 per output).
//   fine o = 2i reads coarse (i-1, i) with (0.25, 0.75) -- (x[0], x[0]) with (1, 0) at i = 0; o = 2i+1 reads (i, i+1) with
//   (0.75, 0.25), i+1 clamped: area_pixel_compute_source_index for scale 1/2, as lin_src().
// ---------------------------------------------------------------------------
// CH channels of one voxel: 4 (one 16-byte access of fp32, 8 bytes of a 16-bit type) or 8 (16 bytes of a 16-bit type)
template <int CH>
struct FV {
    float v[CH];
};
template <int CH, typename T>
__device__ __forceinline__ FV<CH> ldv(const T* p) {
    FV<CH> r;
    if constexpr (CH == 8) {
        act_ld8(p, r.v);
    } else {
        const float4 t = act_ld4(p);
        r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
    }
    return r;
}
template <int CH, typename T>
__device__ __forceinline__ void stv(T* p, const FV<CH>& a) {
    if constexpr (CH == 8) act_st8(p, a.v);
    else act_st4(p, make_float4(a.v[0], a.v[1], a.v[2], a.v[3]));
}
template <int CH>
__device__ __forceinline__ FV<CH> fv_zero() {
    FV<CH> r;
#pragma unroll
    for (int j = 0; j < CH; ++j) r.v[j] = 0.f;
    return r;
}
template <int CH>
__device__ __forceinline__ FV<CH> lerp2(float l0, const FV<CH>& a, float l1, const FV<CH>& b) {
    FV<CH> r;
#pragma unroll
    for (int j = 0; j < CH; ++j) r.v[j] = l0 * a.v[j] + l1 * b.v[j];
    return r;
}

// forward; optionally the (sum y, sum y^2) partials of the block's coarse row [N][D*H][C][2] (what tem_upsample_stats
// derives from u with a 27-point stencil: here the outputs are in registers anyway)
template <int FZ, typename T, int CH = 4>
__global__ __launch_bounds__(256) void k_upsample2_fwd(const T* __restrict__ x, int64_t x_ld, T* __restrict__ y,
                                                       int64_t y_ld, int D, int H, int W, int C,
                                                       float* __restrict__ part) {
    extern __shared__ float lsu[];  // [4 waves][C][2] when part
    const int Ho = 2 * H, Wo = 2 * W;
    const int cq = C / CH;
    int row = blockIdx.x;  // (n, zi, yi) of the coarse grid
    const int yi = row % H;
    row /= H;
    const int zi = row % D;
    const int n = row / D;
    const int ym = max(yi - 1, 0), yp = min(yi + 1, H - 1);
    const int zm = max(zi - 1, 0), zp = min(zi + 1, D - 1);
    const float ya0 = yi == 0 ? 1.f : 0.25f, ya1 = yi == 0 ? 0.f : 0.75f;   // fine 2yi: (ym, yi)
    const float za0 = zi == 0 ? 1.f : 0.25f, za1 = zi == 0 ? 0.f : 0.75f;
    const int zrow[3] = {zm, zi, zp}, yrow[3] = {ym, yi, yp};
    float s1[CH], s2[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) s1[j] = s2[j] = 0.f;
    for (int i = threadIdx.x; i < W * cq; i += 256) {
        const int xi = i / cq, c0 = (i % cq) * CH;
        const int xm = max(xi - 1, 0), xp = min(xi + 1, W - 1);
        const float xa0 = xi == 0 ? 1.f : 0.25f, xa1 = xi == 0 ? 0.f : 0.75f;
        FV<CH> Y[FZ == 2 ? 3 : 1][2][2];  // [coarse z][fine y parity][fine x parity]
#pragma unroll
        for (int kz = 0; kz < (FZ == 2 ? 3 : 1); ++kz) {
            const int zc = FZ == 2 ? zrow[kz] : zi;
            FV<CH> X[3][2];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const T* r = x + ((((int64_t)n * D + zc) * H + yrow[ky]) * W) * x_ld + c0;
                const FV<CH> a = ldv<CH>(r + (int64_t)xm * x_ld), b = ldv<CH>(r + (int64_t)xi * x_ld), c = ldv<CH>(r + (int64_t)xp * x_ld);
                X[ky][0] = lerp2(xa0, a, xa1, b);
                X[ky][1] = lerp2(0.75f, b, 0.25f, c);
            }
#pragma unroll
            for (int sx = 0; sx < 2; ++sx) {
                Y[kz][0][sx] = lerp2(ya0, X[0][sx], ya1, X[1][sx]);
                Y[kz][1][sx] = lerp2(0.75f, X[1][sx], 0.25f, X[2][sx]);
            }
            __builtin_amdgcn_sched_barrier(0);   // one plane's 9 loads in flight, not all 27 (184 VGPRs)
        }
#pragma unroll
        for (int sz = 0; sz < FZ; ++sz)
#pragma unroll
            for (int sy = 0; sy < 2; ++sy)
#pragma unroll
                for (int sx = 0; sx < 2; ++sx) {
                    FV<CH> o;
                    if constexpr (FZ == 2)
                        o = sz == 0 ? lerp2(za0, Y[0][sy][sx], za1, Y[1][sy][sx]) : lerp2(0.75f, Y[1][sy][sx], 0.25f, Y[2][sy][sx]);
                    else
                        o = Y[0][sy][sx];
                    const int64_t vo = (((int64_t)n * (D * FZ) + zi * FZ + sz) * Ho + 2 * yi + sy) * Wo + 2 * xi + sx;
                    if constexpr (sizeof(T) == 2) {   // the row sums describe the tensor AS STORED
#pragma unroll
                        for (int j = 0; j < CH; j += 2) {
                            const unsigned pk = act_pk<T>(o.v[j], o.v[j + 1]);
                            o.v[j] = act_lo<T>(pk);
                            o.v[j + 1] = act_hi<T>(pk);
                        }
                    }
                    stv<CH>(y + vo * y_ld + c0, o);
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        s1[j] += o.v[j];
                        s2[j] = fmaf(o.v[j], o.v[j], s2[j]);
                    }
                }
    }
    if (part) {
        const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            for (int o = cq; o < 64; o <<= 1) {  // lanes sharing a channel group sit cq apart (256 % cq == 0)
                s1[j] += __shfl_xor(s1[j], o, 64);
                s2[j] += __shfl_xor(s2[j], o, 64);
            }
            if (lane < cq) {
                lsu[(wv * C + lane * CH + j) * 2 + 0] = s1[j];
                lsu[(wv * C + lane * CH + j) * 2 + 1] = s2[j];
            }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < 2 * C; t += 256) {
            float a = 0.f;
            for (int w4 = 0; w4 < 4; ++w4) a += lsu[w4 * 2 * C + t];
            part[((int64_t)n * D * H + (int64_t)zi * H + yi) * 2 * C + t] = a;
        }
    }
}

// One axis of the adjoint for a coarse pair (i0, i0 + 1) at factor 2: the 6 fine positions 2*i0 - 1 + k and, per coarse
// c, the weights of k = 2c .. 2c + 3 (zero outside the volume / for a pair's missing second element).
struct UpbAxis {
    int idx[6];
    float w[2][4];
};
__device__ __forceinline__ UpbAxis upb_axis2(int i0, int in) {
    UpbAxis a;
    const int out = 2 * in;
#pragma unroll
    for (int k = 0; k < 6; ++k) a.idx[k] = min(max(2 * i0 - 1 + k, 0), out - 1);
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int o = 2 * i0 - 1 + 2 * c + k;
            a.w[c][k] = (o >= 0 && o < out && i0 + c < in) ? lin_w(o, 2, in, i0 + c) : 0.f;
        }
    return a;
}
// weight of fine position o (factor 2) in coarse i's adjoint; 0 outside the volume / for a missing pair element
__device__ __forceinline__ float upb_w(int o, int in, int i) {
    return (o >= 0 && o < 2 * in && i < in) ? lin_w(o, 2, in, i) : 0.f;
}
// a[d] for d in 0..2, else 0 -- without a dynamically indexed register array
__device__ __forceinline__ float pick3(const float (&a)[3], int d) {
    return d == 0 ? a[0] : d == 1 ? a[1] : d == 2 ? a[2] : 0.f;
}
// (U^T U)[i][i + d] and (U^T 1)[i] for the pair: positions i0 - 1 + k, k = 0..3; coarse c uses k = c .. c + 2
struct UtuAxis {
    int idx[4];
    float a[2][3];
    float s[2];
};
__device__ __forceinline__ UtuAxis utu_axis2(int i0, int f, int in) {
    UtuAxis r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r.idx[k] = min(max(i0 - 1 + k, 0), in - 1);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        r.a[c][0] = r.a[c][1] = r.a[c][2] = 0.f;
        r.s[c] = 0.f;
        if (i0 + c < in) utu_axis(i0 + c, f, in, r.a[c], r.s[c]);
    }
    return r;
}

template <int FZ, typename T, int CH = 4>
__global__ __launch_bounds__(256) void k_upsample2_bwd(const T* __restrict__ gy, int64_t gy_ld,
                                                       T* __restrict__ gx, int64_t gx_ld, int D, int H, int W, int C,
                                                       const T* __restrict__ u, int64_t u_ld,
                                                       const float* __restrict__ ncoef, int64_t ncoef_ld) {
    constexpr int CZ = FZ == 2 ? 2 : 1;      // coarse z per thread
    constexpr int NZF = FZ == 2 ? 6 : 1;     // fine z planes it reads
    const int Do = D * FZ, Ho = 2 * H, Wo = 2 * W;
    const int cq = C / CH;
    const int Hp = (H + 1) >> 1, Wp = (W + 1) >> 1, Dp = (D + CZ - 1) / CZ;
    int row = blockIdx.x;  // (n, z pair, y pair) of the coarse grid
    const int y0 = (row % Hp) * 2;
    row /= Hp;
    const int z0 = (row % Dp) * CZ;
    const int n = row / Dp;
    for (int i = threadIdx.x; i < Wp * cq; i += 256) {
        const int x0 = (i / cq) * 2, c0 = (i % cq) * CH;
        const UpbAxis ax = upb_axis2(x0, W);
        FV<CH> acc[CZ][2][2];
#pragma unroll
        for (int a = 0; a < CZ; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c) acc[a][b][c] = fv_zero<CH>();
        // the z and y loops stay rolled (plane / row index and weights recomputed from the loop counter: wave-uniform
        // scalar work): unrolled, the 216 loads and their addresses were all hoisted (500 VGPRs, occupancy 1)
#pragma unroll 1
        for (int kz = 0; kz < NZF; ++kz) {
            const int oz = FZ == 2 ? 2 * z0 - 1 + kz : z0;
            const int zf = min(max(oz, 0), Do - 1);
            float wz[CZ];
#pragma unroll
            for (int a = 0; a < CZ; ++a) wz[a] = FZ == 2 ? upb_w(oz, D, z0 + a) : 1.f;
            FV<CH> pp[2][2];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c) pp[b][c] = fv_zero<CH>();
#pragma unroll 2
            for (int ky = 0; ky < 6; ++ky) {
                const int oy = 2 * y0 - 1 + ky;
                const float wy[2] = {upb_w(oy, H, y0), upb_w(oy, H, y0 + 1)};
                const T* r = gy + ((((int64_t)n * Do + zf) * Ho + min(max(oy, 0), Ho - 1)) * Wo) * gy_ld + c0;
                FV<CH> t[6];
#pragma unroll
                for (int kx = 0; kx < 6; ++kx) t[kx] = ldv<CH>(r + (int64_t)ax.idx[kx] * gy_ld);
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        const float xr = ax.w[c][0] * t[2 * c].v[j] + ax.w[c][1] * t[2 * c + 1].v[j] +
                                         ax.w[c][2] * t[2 * c + 2].v[j] + ax.w[c][3] * t[2 * c + 3].v[j];
                        pp[0][c].v[j] = fmaf(wy[0], xr, pp[0][c].v[j]);
                        pp[1][c].v[j] = fmaf(wy[1], xr, pp[1][c].v[j]);
                    }
            }
#pragma unroll
            for (int a = 0; a < CZ; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int j = 0; j < CH; ++j) acc[a][b][c].v[j] = fmaf(wz[a], pp[b][c].v[j], acc[a][b][c].v[j]);
        }
        if (ncoef) {
            // U^T(norm backward(g)) = a*U^T g - m1*U^T 1 - m2r*(U^T U u - mean*U^T 1) (see k_upsample_bwd): the 27-point
            // stencil on u, separably over the pair's 4x4x4 neighbourhood
            const UtuAxis bx = utu_axis2(x0, 2, W), by = utu_axis2(y0, 2, H), bz = utu_axis2(z0, FZ, D);
            FV<CH> q[CZ][2][2];
#pragma unroll
            for (int a = 0; a < CZ; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int c = 0; c < 2; ++c) q[a][b][c] = fv_zero<CH>();
            constexpr int NZU = FZ == 2 ? 4 : 1;
#pragma unroll 1
            for (int kz = 0; kz < NZU; ++kz) {
                const int zc = FZ == 2 ? min(max(z0 - 1 + kz, 0), D - 1) : z0;
                float wz[CZ];
#pragma unroll
                for (int a = 0; a < CZ; ++a) wz[a] = FZ == 2 ? pick3(bz.a[a], kz - a) : 1.f;
                FV<CH> qp[2][2];
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int c = 0; c < 2; ++c) qp[b][c] = fv_zero<CH>();
#pragma unroll 2
                for (int ky = 0; ky < 4; ++ky) {
                    const float wy[2] = {pick3(by.a[0], ky), pick3(by.a[1], ky - 1)};
                    const T* r = u + ((((int64_t)n * D + zc) * H + min(max(y0 - 1 + ky, 0), H - 1)) * W) * u_ld + c0;
                    FV<CH> t[4];
#pragma unroll
                    for (int kx = 0; kx < 4; ++kx) t[kx] = ldv<CH>(r + (int64_t)bx.idx[kx] * u_ld);
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int j = 0; j < CH; ++j) {
                            const float xr = bx.a[c][0] * t[c].v[j] + bx.a[c][1] * t[c + 1].v[j] + bx.a[c][2] * t[c + 2].v[j];
                            qp[0][c].v[j] = fmaf(wy[0], xr, qp[0][c].v[j]);
                            qp[1][c].v[j] = fmaf(wy[1], xr, qp[1][c].v[j]);
                        }
                }
#pragma unroll
                for (int a = 0; a < CZ; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int c = 0; c < 2; ++c)
#pragma unroll
                            for (int j = 0; j < CH; ++j) q[a][b][c].v[j] = fmaf(wz[a], qp[b][c].v[j], q[a][b][c].v[j]);
            }
            float4 kc[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) kc[j] = *reinterpret_cast<const float4*>(ncoef + (int64_t)n * ncoef_ld + (c0 + j) * 4);
#pragma unroll
            for (int a = 0; a < CZ; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const float S = (FZ == 2 ? bz.s[a] : 1.f) * by.s[b] * bx.s[c];
#pragma unroll
                        for (int j = 0; j < CH; ++j)
                            acc[a][b][c].v[j] = kc[j].x * acc[a][b][c].v[j] - kc[j].y * S - kc[j].z * (q[a][b][c].v[j] - kc[j].w * S);
                    }
        }
#pragma unroll
        for (int a = 0; a < CZ; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    if (z0 + a >= D || y0 + b >= H || x0 + c >= W) continue;
                    const int64_t v = (((int64_t)n * D + z0 + a) * H + y0 + b) * W + x0 + c;
                    stv<CH>(gx + v * gx_ld + c0, acc[a][b][c]);
                }
    }
}

// Statistics of y = upsample(u) WITHOUT reading y: sum_o y[o] = sum_i (U^T 1)[i] u[i] and sum_o y[o]^2 = sum_i u[i] (U^T U u)[i]
// -- the same low-resolution 27-point stencil as tem_upsample_bwd_norm.  One block per low-resolution row (n, z, y);
// part: [N][D*H][C][2] partial sums (sum y, sum y^2) in the layout tem_norm_finalize_partials2 merges.
template <int VEC, typename T>
__global__ __launch_bounds__(256) void k_upsample_stats(const T* __restrict__ u, int64_t u_ld, int D, int H, int W,
                                                        int C, int fz, int fy, int fx, float* __restrict__ part) {
    extern __shared__ float lsu[];  // [4 waves][C][2]
    const int cq = C / VEC;         // VEC == 4: power of two <= 64 (launcher); a thread keeps its channel quad
    int row = blockIdx.x;
    const int yi = row % H;
    row /= H;
    const int zi = row % D;
    const int n = row / D;
    float az[3], ay[3], sz, sy;
    utu_axis(zi, fz, D, az, sz);
    utu_axis(yi, fy, H, ay, sy);
    float s1[VEC], s2[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) s1[j] = s2[j] = 0.f;
    for (int i = threadIdx.x; i < W * cq; i += 256) {
        const int xi = i / cq, c0 = (i % cq) * VEC;
        float ax[3], sx;
        utu_axis(xi, fx, W, ax, sx);
        const float S = sz * sy * sx;
        float q[VEC], uc[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) q[j] = 0.f;
        ld_vec<VEC>(u + ((((int64_t)n * D + zi) * H + yi) * W + xi) * u_ld + c0, uc);
#pragma unroll
        for (int dz = -1; dz <= 1; ++dz) {  // all 27 taps unconditionally (clamped address, weight 0 outside)
            const int zc = min(max(zi + dz, 0), D - 1);
            float t[3][3][VEC];
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy) {
                const int yc = min(max(yi + dy, 0), H - 1);
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int xc = min(max(xi + dx, 0), W - 1);
                    ld_vec<VEC>(u + ((((int64_t)n * D + zc) * H + yc) * W + xc) * u_ld + c0, t[dy + 1][dx + 1]);
                }
            }
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const float wq = az[dz + 1] * ay[dy + 1] * ax[dx + 1];
#pragma unroll
                    for (int j = 0; j < VEC; ++j) q[j] = fmaf(wq, t[dy + 1][dx + 1][j], q[j]);
                }
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            s1[j] = fmaf(S, uc[j], s1[j]);
            s2[j] = fmaf(uc[j], q[j], s2[j]);
        }
    }
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        for (int o = cq; o < 64; o <<= 1) {  // lanes sharing a quad sit cq apart (256 % cq == 0)
            s1[j] += __shfl_xor(s1[j], o, 64);
            s2[j] += __shfl_xor(s2[j], o, 64);
        }
        if (lane < cq) {
            lsu[(wv * C + lane * VEC + j) * 2 + 0] = s1[j];
            lsu[(wv * C + lane * VEC + j) * 2 + 1] = s2[j];
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * C; t += 256) {
        float a = 0.f;
        for (int w4 = 0; w4 < 4; ++w4) a += lsu[w4 * 2 * C + t];
        part[((int64_t)n * D * H + (int64_t)zi * H + yi) * 2 * C + t] = a;
    }
}

static int upsample_stats_impl(const void* u, int64_t u_ld, int N, int D, int H, int W, int C, int fz, int fy, int fx,
                               float* part, int st, tem_stream_t stream) {
    TEM_REQUIRE(u && part && N > 0 && C > 0 && u_ld >= C && D > 0 && H > 0 && W > 0 && fz > 0 && fy > 0 && fx > 0,
                "tem_upsample_stats: bad arguments");
    TEM_REQUIRE(st >= 0 && st <= 2, "tem_upsample_stats: unknown storage type %d", st);
    const int cq = C / 4;
    TEM_REQUIRE(C % 4 == 0 && cq <= 64 && (cq & (cq - 1)) == 0 && u_ld % 4 == 0 && ((uintptr_t)u % tem_st_align4(st) == 0),
                "tem_upsample_stats: needs C = 4 * 2^k <= 256 channels, rows aligned to 4 elements (got C=%d)", C);
    int64_t rows = (int64_t)N * D * H;
    TEM_REQUIRE(rows < (1ll << 31), "tem_upsample_stats: too many rows");
    TEM_ST_SWITCH(st, T, hipLaunchKernelGGL((k_upsample_stats<4, T>), dim3((unsigned)rows), dim3(256), (size_t)4 * C * 2 * sizeof(float),
                                            (hipStream_t)stream, (const T*)u, u_ld, D, H, W, C, fz, fy, fx, part));
    TEM_CHECK_LAUNCH("tem_upsample_stats");
    return TEM_OK;
}
extern "C" int tem_upsample_stats(const float* u, int64_t u_ld, int N, int D, int H, int W, int C, int fz, int fy, int fx,
                                  float* part, tem_stream_t stream) {
    return upsample_stats_impl(u, u_ld, N, D, H, W, C, fz, fy, fx, part, 0, stream);
}
extern "C" int tem_upsample_stats_st(const void* u, int64_t u_ld, int N, int D, int H, int W, int C, int fz, int fy, int fx,
                                     float* part, int st, tem_stream_t stream) {
    return upsample_stats_impl(u, u_ld, N, D, H, W, C, fz, fy, fx, part, st, stream);
}

// the factor-2 kernels: channel quads per thread, and (for the row partials) a quad per lane: C = 4 * 2^k <= 256
static inline bool upsample2_ok(int C, int fz, int fy, int fx) {
    return fy == 2 && fx == 2 && (fz == 1 || fz == 2) && C % 4 == 0 && !tem_option(TEM_OPT_UPSAMPLE_GENERIC);
}

// tem_upsample_fwd that also returns the first stage of y's statistics: part [N][D*H][C][2] (sum y, sum y^2 per coarse
// row -- the layout of tem_upsample_stats, which it replaces when the factor-2 kernel takes the shape).  Returns
// TEM_EINVAL when it does not (query tem_upsample_fwd_stats_ok first).
extern "C" int tem_upsample_fwd_stats_ok(int C, int fz, int fy, int fx) {
    const int cq = C / 4;
    return upsample2_ok(C, fz, fy, fx) && cq <= 64 && (cq & (cq - 1)) == 0;
}

static int upsample_fwd_impl(const void* x, int64_t x_ld, void* y, int64_t y_ld, int N, int D, int H, int W, int C, int fz,
                             int fy, int fx, float* part, int st, tem_stream_t stream) {
    TEM_REQUIRE(x && y && N > 0 && C > 0 && x_ld >= C && y_ld >= C && D > 0 && H > 0 && W > 0,
                "tem_upsample_fwd: bad arguments");
    TEM_REQUIRE(st >= 0 && st <= 2, "tem_upsample_fwd: unknown storage type %d", st);
    TEM_REQUIRE(fz > 0 && fy > 0 && fx > 0, "tem_upsample_fwd: bad factors");
    const bool v4 = vec4_ok(C, {x, y}, {x_ld, y_ld}, st);
    TEM_REQUIRE(!part || (tem_upsample_fwd_stats_ok(C, fz, fy, fx) && v4),
                "tem_upsample_fwd_stats: needs factors (1|2, 2, 2), C = 4 * 2^k <= 256 and rows aligned to 4 elements");
    int64_t rows = (int64_t)N * D * fz * H * fy;
    TEM_REQUIRE(rows < (1ll << 31), "tem_upsample_fwd: too many rows");
    const size_t ldsb = part ? (size_t)4 * C * 2 * sizeof(float) : 0;
    // 16-bit tensors: 8 channels per thread (16-byte accesses) where the row partials keep their layout (C / 8 a power of two)
    // (factor (1, 2, 2) only: with three coarse z-planes in registers the 8-channel variant needs 300 VGPRs)
    if (fz == 1 && upsample2_ok(C, fz, fy, fx) && vec8_ok(C, {x, y}, {x_ld, y_ld}, st) && ((C / 8) & (C / 8 - 1)) == 0 && C / 8 <= 64) {
        const int64_t crow = (int64_t)N * D * H;
        TEM_ST16_SWITCH(st, T, {
            hipLaunchKernelGGL((k_upsample2_fwd<1, T, 8>), dim3((unsigned)crow), dim3(256), ldsb, (hipStream_t)stream, (const T*)x,
                               x_ld, (T*)y, y_ld, D, H, W, C, part);
        });
        TEM_CHECK_LAUNCH("tem_upsample_fwd");
        return TEM_OK;
    }
    TEM_ST_SWITCH(st, T, {
        const T* xs = (const T*)x;
        T* ys = (T*)y;
        if (upsample2_ok(C, fz, fy, fx) && v4) {
            const int64_t crow = (int64_t)N * D * H;
            if (fz == 2)
                hipLaunchKernelGGL((k_upsample2_fwd<2, T>), dim3((unsigned)crow), dim3(256), ldsb, (hipStream_t)stream, xs, x_ld, ys,
                                   y_ld, D, H, W, C, part);
            else
                hipLaunchKernelGGL((k_upsample2_fwd<1, T>), dim3((unsigned)crow), dim3(256), ldsb, (hipStream_t)stream, xs, x_ld, ys,
                                   y_ld, D, H, W, C, part);
        } else if (v4)
            hipLaunchKernelGGL((k_upsample_fwd<4, T>), dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, xs, x_ld, ys,
                               y_ld, D, H, W, C, fz, fy, fx);
        else
            hipLaunchKernelGGL((k_upsample_fwd<1, T>), dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, xs, x_ld, ys,
                               y_ld, D, H, W, C, fz, fy, fx);
    });
    TEM_CHECK_LAUNCH("tem_upsample_fwd");
    return TEM_OK;
}

extern "C" int tem_upsample_fwd_stats(const float* x, int64_t x_ld, float* y, int64_t y_ld, int N, int D, int H, int W,
                                      int C, int fz, int fy, int fx, float* part, tem_stream_t stream) {
    TEM_REQUIRE(part, "tem_upsample_fwd_stats: bad arguments");
    return upsample_fwd_impl(x, x_ld, y, y_ld, N, D, H, W, C, fz, fy, fx, part, 0, stream);
}

extern "C" int tem_upsample_fwd(const float* x, int64_t x_ld, float* y, int64_t y_ld, int N, int D, int H, int W, int C,
                                int fz, int fy, int fx, tem_stream_t stream) {
    return upsample_fwd_impl(x, x_ld, y, y_ld, N, D, H, W, C, fz, fy, fx, nullptr, 0, stream);
}

// tem_upsample_fwd / _fwd_stats (part != NULL) for tensors of storage type st
extern "C" int tem_upsample_fwd_st(const void* x, int64_t x_ld, void* y, int64_t y_ld, int N, int D, int H, int W, int C,
                                   int fz, int fy, int fx, float* part, int st, tem_stream_t stream) {
    return upsample_fwd_impl(x, x_ld, y, y_ld, N, D, H, W, C, fz, fy, fx, part, st, stream);
}

static int upsample_bwd_impl(const void* gy, int64_t gy_ld, void* gx, int64_t gx_ld, int N, int D, int H, int W,
                             int C, int fz, int fy, int fx, const void* u, int64_t u_ld, const float* ncoef,
                             int64_t ncoef_ld, int st, tem_stream_t stream) {
    TEM_REQUIRE(gy && gx && N > 0 && C > 0 && gy_ld >= C && gx_ld >= C && D > 0 && H > 0 && W > 0,
                "tem_upsample_bwd: bad arguments");
    TEM_REQUIRE(st >= 0 && st <= 2, "tem_upsample_bwd: unknown storage type %d", st);
    TEM_REQUIRE(fz > 0 && fy > 0 && fx > 0, "tem_upsample_bwd: bad factors");
    int64_t rows = (int64_t)N * D * H;
    TEM_REQUIRE(rows < (1ll << 31), "tem_upsample_bwd: too many rows");
    // the 2x2x2-per-thread kernel has 1/8 of the gather kernel's threads: it needs a volume that still fills the chip
    const int64_t pairs = (int64_t)N * ((D + fz - 1) / fz) * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
    if (upsample2_ok(C, fz, fy, fx) && pairs >= 2 * 65536 && vec8_ok(C, {gy, gx, u}, {gy_ld, gx_ld, u ? u_ld : 0}, st) && tem_option(TEM_OPT_UPSAMPLE2_CH8)) {
        TEM_ST16_SWITCH(st, T, {
            if (fz == 2) {
                const int64_t prow = (int64_t)N * ((D + 1) / 2) * ((H + 1) / 2);
                hipLaunchKernelGGL((k_upsample2_bwd<2, T, 8>), dim3((unsigned)prow), dim3(256), 0, (hipStream_t)stream, (const T*)gy,
                                   gy_ld, (T*)gx, gx_ld, D, H, W, C, (const T*)u, u_ld, ncoef, ncoef_ld);
            } else {
                const int64_t prow = (int64_t)N * D * ((H + 1) / 2);
                hipLaunchKernelGGL((k_upsample2_bwd<1, T, 8>), dim3((unsigned)prow), dim3(256), 0, (hipStream_t)stream, (const T*)gy,
                                   gy_ld, (T*)gx, gx_ld, D, H, W, C, (const T*)u, u_ld, ncoef, ncoef_ld);
            }
        });
        TEM_CHECK_LAUNCH("tem_upsample_bwd");
        return TEM_OK;
    }
    TEM_ST_SWITCH(st, T, {
        const T* gys = (const T*)gy;
        const T* us = (const T*)u;
        T* gxs = (T*)gx;
        if (upsample2_ok(C, fz, fy, fx) && pairs >= 65536 && vec4_ok(C, {gy, gx, u}, {gy_ld, gx_ld, u_ld}, st)) {
            if (fz == 2) {
                const int64_t prow = (int64_t)N * ((D + 1) / 2) * ((H + 1) / 2);
                hipLaunchKernelGGL((k_upsample2_bwd<2, T>), dim3((unsigned)prow), dim3(256), 0, (hipStream_t)stream, gys, gy_ld, gxs,
                                   gx_ld, D, H, W, C, us, u_ld, ncoef, ncoef_ld);
            } else {
                const int64_t prow = (int64_t)N * D * ((H + 1) / 2);
                hipLaunchKernelGGL((k_upsample2_bwd<1, T>), dim3((unsigned)prow), dim3(256), 0, (hipStream_t)stream, gys, gy_ld, gxs,
                                   gx_ld, D, H, W, C, us, u_ld, ncoef, ncoef_ld);
            }
        } else if (vec4_ok(C, {gy, gx, u}, {gy_ld, gx_ld, u ? u_ld : 0}, st))
            hipLaunchKernelGGL((k_upsample_bwd<4, T>), dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, gys, gy_ld, gxs,
                               gx_ld, D, H, W, C, fz, fy, fx, us, u_ld, ncoef, ncoef_ld);
        else
            hipLaunchKernelGGL((k_upsample_bwd<1, T>), dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, gys, gy_ld, gxs,
                               gx_ld, D, H, W, C, fz, fy, fx, us, u_ld, ncoef, ncoef_ld);
    });
    TEM_CHECK_LAUNCH("tem_upsample_bwd");
    return TEM_OK;
}

extern "C" int tem_upsample_bwd(const float* gy, int64_t gy_ld, float* gx, int64_t gx_ld, int N, int D, int H, int W,
                                int C, int fz, int fy, int fx, tem_stream_t stream) {
    return upsample_bwd_impl(gy, gy_ld, gx, gx_ld, N, D, H, W, C, fz, fy, fx, nullptr, 0, nullptr, 0, 0, stream);
}

// tem_upsample_bwd of the RAW data gradient behind a norm whose input was upsample(u): U^T(norm backward(g)) =
// a*U^T g - m1*U^T 1 - m2r*(U^T U u - mean*U^T 1), evaluated with a 27-point stencil on the low-resolution u.
extern "C" int tem_upsample_bwd_norm(const float* gy, int64_t gy_ld, float* gx, int64_t gx_ld, int N, int D, int H,
                                     int W, int C, int fz, int fy, int fx, const float* u, int64_t u_ld,
                                     const float* ncoef, int64_t ncoef_ld, tem_stream_t stream) {
    TEM_REQUIRE(u && ncoef, "tem_upsample_bwd_norm: bad arguments");
    return tem_upsample_bwd_st(gy, gy_ld, gx, gx_ld, N, D, H, W, C, fz, fy, fx, u, u_ld, ncoef, ncoef_ld, TEM_ST_F32, stream);
}

// tem_upsample_bwd / _bwd_norm (u, ncoef != NULL) for tensors of storage type st
extern "C" int tem_upsample_bwd_st(const void* gy, int64_t gy_ld, void* gx, int64_t gx_ld, int N, int D, int H, int W, int C,
                                   int fz, int fy, int fx, const void* u, int64_t u_ld, const float* ncoef, int64_t ncoef_ld,
                                   int st, tem_stream_t stream) {
    TEM_REQUIRE((u == nullptr) == (ncoef == nullptr), "tem_upsample_bwd_norm: u and ncoef come together");
    if (ncoef) {
        TEM_REQUIRE(u_ld >= C && ncoef_ld >= 4 * C && ((uintptr_t)ncoef % 16 == 0) && ncoef_ld % 4 == 0,
                    "tem_upsample_bwd_norm: bad arguments");
        TEM_REQUIRE(st >= 0 && st <= 2 && (C % 4 || (u_ld % 4 == 0 && (uintptr_t)u % tem_st_align4(st) == 0)),
                    "tem_upsample_bwd_norm: u must be aligned to 4 elements");
    }
    return upsample_bwd_impl(gy, gy_ld, gx, gx_ld, N, D, H, W, C, fz, fy, fx, u, u_ld, ncoef, ncoef_ld, st, stream);
}
