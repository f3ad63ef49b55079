// norm.hip -- InstanceNorm / GroupNorm statistics and backward for NDHWC fp32 tensors.
// Replaces nn.InstanceNorm3d / nn.GroupNorm of get_norm_layer (reference model/unet.py:391-406).
// The normalisation itself is never materialised: tem_norm_stats emits per-(n,c)
// scale/shift that the conv kernels apply while staging their input (HBM: one
// read of x instead of read+write+read).  All reductions are two-stage and
// deterministic: fp32 per-thread partials -> fp32 per-block partials -> fp64 merge.
#include "tem_common.h"
#include "tem_act.h"

#define NORM_MAX_BLOCKS 512
#ifndef TEM_NORM_NT
#define TEM_NORM_NT 1   // nontemporal loads / stores in the backward apply pass (streams 3 tensors once): -0.1 ms/step
#endif
#ifndef TEM_NORM_NT2
#define TEM_NORM_NT2 0
#endif
template <typename T> __device__ __forceinline__ float4 nt2_load4(const T* p) {
#if TEM_NORM_NT2
    return act_ld4_nt(p);
#else
    return act_ld4(p);
#endif
}
template <typename T> __device__ __forceinline__ float4 nt_load4(const T* p) {
#if TEM_NORM_NT
    return act_ld4_nt(p);
#else
    return act_ld4(p);
#endif
}
template <typename T> __device__ __forceinline__ void nt_store4(T* p, float4 v) {
#if TEM_NORM_NT
    act_st4_nt(p, v);
#else
    act_st4(p, v);
#endif
}
#define NT2_LOAD4(p) nt2_load4(p)
#define NT_LOAD4(p) nt_load4(p)
#define NT_STORE4(p, v) nt_store4(p, v)
#define NORM_MAX_C 1024

struct NormGeom {
    int vec;       // 4 or 1 floats per thread item
    int cq;        // channel items per voxel (C / vec)
    int rows;      // voxel rows processed concurrently by one block
    int threads;   // cq * rows
    int nblk;      // blocks per sample
    int64_t vper;  // voxels per block
};

static NormGeom norm_geom(const void* p0, const void* p1, int64_t ld0, int64_t ld1, int64_t V, int C, int st = 0) {
    NormGeom g;
    const uintptr_t a4 = tem_st_align4(st);
    bool al = ((uintptr_t)p0 % a4 == 0) && (p1 == nullptr || (uintptr_t)p1 % a4 == 0);
    g.vec = (C % 4 == 0 && ld0 % 4 == 0 && (p1 == nullptr || ld1 % 4 == 0) && al) ? 4 : 1;
    g.cq = C / g.vec;
    g.rows = 256 / g.cq;
    if (g.rows < 1) g.rows = 1;
    g.threads = g.cq * g.rows;
    int64_t nb = tem_cdiv(V, (int64_t)g.rows * 16);
    if (nb > NORM_MAX_BLOCKS) nb = NORM_MAX_BLOCKS;
    if (nb < 1) nb = 1;
    g.nblk = (int)nb;
    g.vper = tem_cdiv(V, nb);
    return g;
}

extern "C" int64_t tem_norm_ws(int N, int64_t V, int C) {
    (void)V;
    // partials [N][NORM_MAX_BLOCKS][C][2] floats + coefficients [N][C][4] floats
    return ((int64_t)N * NORM_MAX_BLOCKS * C * 2 + (int64_t)N * C * 4) * (int64_t)sizeof(float);
}

// ---------------------------------------------------------------------------
// stage 1: per-block partial sums.  MODE 0: (sum x, sum x^2); MODE 1: (sum g, sum g*xn)
// ---------------------------------------------------------------------------
template <int VEC, int MODE, typename T>
__global__ __launch_bounds__(VEC == 4 ? 256 : 1024) void k_norm_partial(const T* __restrict__ x, int64_t x_ld,
                                                       const T* __restrict__ g, int64_t g_ld, int64_t V, int C,
                                                       int G, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, int cq, int rows, int64_t vper,
                                                       float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float sh[];  // [rows][C][2]
    const int n = blockIdx.y, b = blockIdx.x, nblk = gridDim.x;
    const int q = threadIdx.x % cq, r = threadIdx.x / cq;
    const int c0 = q * VEC;
    int64_t v0 = (int64_t)b * vper, v1 = v0 + vper;
    if (v1 > V) v1 = V;
    float a0[VEC], a1[VEC];
    float mu[VEC], rs[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        a0[j] = 0.f;
        a1[j] = 0.f;
        mu[j] = 0.f;
        rs[j] = 1.f;
    }
    if (MODE == 1) {
        const int cg = C / G;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            int grp = (c0 + j) / cg;
            mu[j] = mean[n * G + grp];
            rs[j] = rstd[n * G + grp];
        }
    }
    const T* xb = x + (int64_t)n * V * x_ld;
    const T* gb = (MODE == 1) ? g + (int64_t)n * V * g_ld : nullptr;
    int64_t vs = v0 + r;
    if constexpr (VEC == 4) {
        // four voxels per trip: 4 (MODE 0) / 8 (MODE 1) independent 16-byte loads in flight per thread
        for (; vs + 3 * (int64_t)rows < v1; vs += 4 * (int64_t)rows) {
            float4 t[4], u[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                t[k] = NT2_LOAD4(xb + (vs + k * (int64_t)rows) * x_ld + c0);
                if constexpr (MODE == 1) u[k] = NT2_LOAD4(gb + (vs + k * (int64_t)rows) * g_ld + c0);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float xv[4] = {t[k].x, t[k].y, t[k].z, t[k].w};
                const float gv[4] = {u[k].x, u[k].y, u[k].z, u[k].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (MODE == 0) {
                        a0[j] += xv[j];
                        a1[j] = fmaf(xv[j], xv[j], a1[j]);
                    } else {
                        float xn = (xv[j] - mu[j]) * rs[j];
                        a0[j] += gv[j];
                        a1[j] = fmaf(gv[j], xn, a1[j]);
                    }
                }
            }
        }
    }
    for (int64_t v = vs; v < v1; v += rows) {
        float xv[VEC], gv[VEC];
        if constexpr (VEC == 4) {
            float4 t = act_ld4(xb + v * x_ld + c0);
            xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
            if constexpr (MODE == 1) {
                float4 u = act_ld4(gb + v * g_ld + c0);
                gv[0] = u.x; gv[1] = u.y; gv[2] = u.z; gv[3] = u.w;
            }
        } else {
            xv[0] = act_ld1(xb + v * x_ld + c0);
            if constexpr (MODE == 1) gv[0] = act_ld1(gb + v * g_ld + c0);
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            if (MODE == 0) {
                a0[j] += xv[j];
                a1[j] = fmaf(xv[j], xv[j], a1[j]);
            } else {
                float xn = (xv[j] - mu[j]) * rs[j];
                a0[j] += gv[j];
                a1[j] = fmaf(gv[j], xn, a1[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        sh[((int64_t)r * C + c0 + j) * 2 + 0] = a0[j];
        sh[((int64_t)r * C + c0 + j) * 2 + 1] = a1[j];
    }
    __syncthreads();
    if (r == 0) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float s0 = 0.f, s1 = 0.f;
            for (int rr = 0; rr < rows; ++rr) {
                s0 += sh[((int64_t)rr * C + c0 + j) * 2 + 0];
                s1 += sh[((int64_t)rr * C + c0 + j) * 2 + 1];
            }
            int64_t o = (((int64_t)n * nblk + b) * C + c0 + j) * 2;
            part[o] = s0;
            part[o + 1] = s1;
        }
    }
}

__device__ __forceinline__ void block_sum2_d(double& a, double& b) {   // blocks of 4 .. 16 waves
    __shared__ double sh2[2][16];
    a = tem_wave_sum_d(a);
    b = tem_wave_sum_d(b);
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sh2[0][w] = a;
        sh2[1][w] = b;
    }
    __syncthreads();
    a = sh2[0][0] + sh2[0][1] + sh2[0][2] + sh2[0][3];
    b = sh2[1][0] + sh2[1][1] + sh2[1][2] + sh2[1][3];
    for (int k = 4; k < (int)(blockDim.x >> 6); ++k) {
        a += sh2[0][k];
        b += sh2[1][k];
    }
    __syncthreads();
}

// the (sum, sum of squares) pairs of channels [c0, c0 + cg) over the nblk partial rows of one sample ([nblk][C][2]), this
// thread's share.  Eight independent 8-byte loads per trip: the level-0 convs leave 16 384 rows, and a loop of one
// load + two dependent double adds per trip walked them in 64 round trips to L2 (42-46 us per call, r04 trace; 26 us
// with 8 per trip, the rest of the way with 1024 threads for such layers: norm_finalize_threads).
__device__ __forceinline__ void norm_sum_partials(const float* __restrict__ part, int nblk, int C, int cg, int c0,
                                                  double& s, double& ss) {
    const int total = nblk * cg, nt = (int)blockDim.x;
    for (int i0 = threadIdx.x; i0 < total; i0 += nt * 8) {
        float2 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {   // never `cond ? load : 0` per lane: that is a branch + s_waitcnt vmcnt(0) per load
            const int i = i0 + k * nt < total ? i0 + k * nt : total - 1;
            const int b = i / cg, c = c0 + i % cg;
            v[k] = *reinterpret_cast<const float2*>(part + ((int64_t)b * C + c) * 2);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const bool in = i0 + k * nt < total;
            s += in ? (double)v[k].x : 0.0;
            ss += in ? (double)v[k].y : 0.0;
        }
    }
}

// stage 2 (forward): one block per (n, group)
__global__ __launch_bounds__(1024) void k_norm_finalize(const float* __restrict__ part, int nblk, int64_t V, int C, int G,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float eps, float* __restrict__ mean, float* __restrict__ rstd,
                                                       float* __restrict__ scale, float* __restrict__ shift) {
    const int n = blockIdx.x / G, grp = blockIdx.x % G;
    const int cg = C / G;
    double s = 0.0, ss = 0.0;
    norm_sum_partials(part + (int64_t)n * nblk * C * 2, nblk, C, cg, grp * cg, s, ss);
    block_sum2_d(s, ss);
    double cnt = (double)V * (double)cg;
    double m = s / cnt;
    double var = ss / cnt - m * m;
    if (var < 0.0) var = 0.0;
    double r = 1.0 / sqrt(var + (double)eps);
    if (threadIdx.x == 0) {
        mean[n * G + grp] = (float)m;
        rstd[n * G + grp] = (float)r;
    }
    for (int i = threadIdx.x; i < cg; i += (int)blockDim.x) {
        int c = grp * cg + i;
        double ga = gamma ? (double)gamma[c] : 1.0;
        double be = beta ? (double)beta[c] : 0.0;
        scale[(int64_t)n * C + c] = (float)(r * ga);
        shift[(int64_t)n * C + c] = (float)(be - m * r * ga);
    }
}

// threads of a finalize block: its nblk * cg partials are one trip of 8 loads per thread up to 2048 of them; the conv
// epilogues of the 128^3 level leave 16 384 rows -- 1024 threads walk those in 2 trips instead of 8
static int norm_finalize_threads(int64_t nblk, int cg) { return nblk * cg > 2048 ? 1024 : 256; }

// stage 2 (forward) for a tensor whose channel halves have different producers (decoder concat): channels [0, CA) from
// partA [n][nblkA][CA][2], channels [CA, C) from partB [n][nblkB][C-CA][2]; a group lies inside one half
__global__ __launch_bounds__(1024) void k_norm_finalize2(const float* __restrict__ partA, int nblkA, int CA,
                                                        const float* __restrict__ partB, int nblkB, int64_t V, int C,
                                                        int G, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, float* __restrict__ mean,
                                                        float* __restrict__ rstd, float* __restrict__ scale,
                                                        float* __restrict__ shift) {
    const int n = blockIdx.x / G, grp = blockIdx.x % G;
    const int cg = C / G;
    const bool inA = grp * cg < CA;
    const float* part = inA ? partA : partB;
    const int nblk = inA ? nblkA : nblkB, Cs = inA ? CA : C - CA, cbase = inA ? grp * cg : grp * cg - CA;
    double s = 0.0, ss = 0.0;
    norm_sum_partials(part + (int64_t)n * nblk * Cs * 2, nblk, Cs, cg, cbase, s, ss);
    block_sum2_d(s, ss);
    double cnt = (double)V * (double)cg;
    double m = s / cnt;
    double var = ss / cnt - m * m;
    if (var < 0.0) var = 0.0;
    double r = 1.0 / sqrt(var + (double)eps);
    if (threadIdx.x == 0) {
        mean[n * G + grp] = (float)m;
        rstd[n * G + grp] = (float)r;
    }
    for (int i = threadIdx.x; i < cg; i += (int)blockDim.x) {
        int c = grp * cg + i;
        double ga = gamma ? (double)gamma[c] : 1.0;
        double be = beta ? (double)beta[c] : 0.0;
        scale[(int64_t)n * C + c] = (float)(r * ga);
        shift[(int64_t)n * C + c] = (float)(be - m * r * ga);
    }
}

// stage 2 (backward): one block per (n, group) -> coef[n][c] = {a, m1, m2r, mean}
//   gx = a*g - m1 - (x - mean)*m2r
__global__ __launch_bounds__(256) void k_norm_bwd_finalize(const float* __restrict__ part, int nblk, int64_t V, int C,
                                                           int G, const float* __restrict__ gamma,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, float* __restrict__ coef) {
    const int n = blockIdx.x / G, grp = blockIdx.x % G;
    const int cg = C / G;
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < nblk * cg; i += 256) {
        int b = i / cg, c = grp * cg + i % cg;
        double ga = gamma ? (double)gamma[c] : 1.0;
        int64_t o = (((int64_t)n * nblk + b) * C + c) * 2;
        s1 += ga * (double)part[o];
        s2 += ga * (double)part[o + 1];
    }
    block_sum2_d(s1, s2);
    double cnt = (double)V * (double)cg;
    double r = (double)rstd[n * G + grp], m = (double)mean[n * G + grp];
    for (int i = threadIdx.x; i < cg; i += 256) {
        int c = grp * cg + i;
        double ga = gamma ? (double)gamma[c] : 1.0;
        float* o = coef + ((int64_t)n * C + c) * 4;
        o[0] = (float)(r * ga);
        o[1] = (float)(r * s1 / cnt);
        o[2] = (float)(r * r * s2 / cnt);
        o[3] = (float)m;
    }
}

// dgamma[c] = sum_{n,b} part[..][1], dbeta[c] = sum part[..][0]; one block per channel (a single thread per channel
// walking up to N*512 partials took 96 us per call, 1.7 ms per GroupNorm training step)
__global__ __launch_bounds__(256) void k_norm_bwd_affine(const float* __restrict__ part, int N, int nblk, int C,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < N * nblk; i += 256) {
        int64_t o = ((int64_t)i * C + c) * 2;
        s1 += (double)part[o];
        s2 += (double)part[o + 1];
    }
    block_sum2_d(s1, s2);
    if (threadIdx.x == 0) {
        if (dbeta) dbeta[c] = (float)s1;
        if (dgamma) dgamma[c] = (float)s2;
    }
}

// amax (optional, out_amax of tem_norm_bwd_st): max |gx| as a by-product.  This kernel is SHORT: all its waves end at the same moment,
// so the per-wave "read the word, atomicMax if larger" of the long kernels degenerates into one atomic per wave on one
// address (~10 ns each: 8192 waves cost +80 us, measured).  With amax the launcher uses 1024-thread blocks and at most 256
// of them; a block reduces in LDS and issues ONE atomic.
template <int VEC, typename T>
__global__ __launch_bounds__(1024) void k_norm_bwd_apply(const T* __restrict__ gy, int64_t gy_ld,
                                                         const T* __restrict__ x, int64_t x_ld,
                                                         T* __restrict__ gx, int64_t gx_ld, int64_t V, int C,
                                                         const float* __restrict__ coef, int relu_mask,
                                                         unsigned* __restrict__ amax) {
    __shared__ unsigned smax[16];
    const int n = blockIdx.y;
    const int cq = C / VEC;
    const int64_t items = V * cq;
    const T* gb = gy + (int64_t)n * V * gy_ld;
    const T* xb = x + (int64_t)n * V * x_ld;
    T* ob = gx + (int64_t)n * V * gx_ld;
    const float* cf = coef + (int64_t)n * C * 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t dv = stride / cq;
    const int dq = (int)(stride % cq);
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t v = i / cq;
    int q = (int)(i % cq);
    float amx = 0.f;
    bool fast = false;
    if constexpr (VEC == 4) fast = dq == 0;
    if (fast) {
        if constexpr (VEC == 4) {
            // the grid stride is a multiple of the channel-quad count (every power-of-two width): a thread keeps its
            // four channels, so the 16 coefficients are loaded once instead of per voxel; two voxels per trip
            const int c0 = q * 4;
            float4 k[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) k[j] = *reinterpret_cast<const float4*>(cf + (int64_t)(c0 + j) * 4);
            for (; i < items; i += 2 * stride, v += 2 * dv) {
                const bool two = i + stride < items;
                const float4 g4 = NT_LOAD4(gb + v * gy_ld + c0);
                const float4 x4 = NT_LOAD4(xb + v * x_ld + c0);
                float4 g5 = g4, x5 = x4;
                if (two) {
                    g5 = NT_LOAD4(gb + (v + dv) * gy_ld + c0);
                    x5 = NT_LOAD4(xb + (v + dv) * x_ld + c0);
                }
                const float ga[4] = {g4.x, g4.y, g4.z, g4.w}, xa[4] = {x4.x, x4.y, x4.z, x4.w};
                const float gb2[4] = {g5.x, g5.y, g5.z, g5.w}, xb2[4] = {x5.x, x5.y, x5.z, x5.w};
                float oa[4], ob2[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float ra = k[j].x * ga[j] - k[j].y - (xa[j] - k[j].w) * k[j].z;
                    const float rb = k[j].x * gb2[j] - k[j].y - (xb2[j] - k[j].w) * k[j].z;
                    oa[j] = (relu_mask && !(xa[j] > 0.f)) ? 0.f : ra;
                    ob2[j] = (relu_mask && !(xb2[j] > 0.f)) ? 0.f : rb;
                }
                amx = tem_amax4(amx, oa[0], oa[1], oa[2], oa[3]);
                NT_STORE4(ob + v * gx_ld + c0, make_float4(oa[0], oa[1], oa[2], oa[3]));
                if (two) {
                    amx = tem_amax4(amx, ob2[0], ob2[1], ob2[2], ob2[3]);
                    NT_STORE4(ob + (v + dv) * gx_ld + c0, make_float4(ob2[0], ob2[1], ob2[2], ob2[3]));
                }
            }
        }
    } else {
        for (; i < items; i += stride, v += dv, q += dq) {
            if (q >= cq) {
                q -= cq;
                ++v;
            }
            const int c0 = q * VEC;
            if constexpr (VEC == 4) {
                float4 g4 = act_ld4(gb + v * gy_ld + c0);
                float4 x4 = act_ld4(xb + v * x_ld + c0);
                float gv[4] = {g4.x, g4.y, g4.z, g4.w}, xv[4] = {x4.x, x4.y, x4.z, x4.w}, ov[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float4 k = *reinterpret_cast<const float4*>(cf + (int64_t)(c0 + j) * 4);
                    float r = k.x * gv[j] - k.y - (xv[j] - k.w) * k.z;
                    ov[j] = (relu_mask && !(xv[j] > 0.f)) ? 0.f : r;
                }
                amx = tem_amax4(amx, ov[0], ov[1], ov[2], ov[3]);
                act_st4(ob + v * gx_ld + c0, make_float4(ov[0], ov[1], ov[2], ov[3]));
            } else {
                float gv = act_ld1(gb + v * gy_ld + c0), xv = act_ld1(xb + v * x_ld + c0);
                float4 k = *reinterpret_cast<const float4*>(cf + (int64_t)c0 * 4);
                float r = k.x * gv - k.y - (xv - k.w) * k.z;
                r = (relu_mask && !(xv > 0.f)) ? 0.f : r;
                amx = __builtin_fmaxf(amx, __builtin_fabsf(r));
                act_st1(ob + v * gx_ld + c0, r);
            }
        }
    }
    if (amax) {   // launch-uniform: one atomic per block
        unsigned u = __builtin_bit_cast(unsigned, amx);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) u = max(u, (unsigned)__shfl_xor((int)u, o, 64));
        if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = u;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < (int)(blockDim.x >> 6); ++w) u = max(u, smax[w]);
            if (u > __atomic_load_n(amax, __ATOMIC_RELAXED)) atomicMax(amax, u);
        }
    }
}

static int norm_stats_impl(const void* x, int64_t x_ld, int N, int64_t V, int C, int G, const float* gamma,
                           const float* beta, float eps, float* mean, float* rstd, float* scale, float* shift,
                           void* ws, int64_t ws_bytes, int st, tem_stream_t stream) {
    TEM_REQUIRE(x && mean && rstd && scale && shift && ws, "tem_norm_stats: null pointer");
    TEM_REQUIRE(st >= 0 && st <= 2, "tem_norm_stats: unknown storage type %d", st);
    TEM_REQUIRE(N > 0 && V > 0 && C > 0 && C <= NORM_MAX_C && x_ld >= C, "tem_norm_stats: bad shape (C=%d)", C);
    TEM_REQUIRE(G > 0 && C % G == 0, "tem_norm_stats: C=%d not divisible by G=%d", C, G);
    if (ws_bytes < tem_norm_ws(N, V, C)) {
        tem_set_error("tem_norm_stats: workspace too small");
        return TEM_EWS;
    }
    NormGeom g = norm_geom(x, nullptr, x_ld, 0, V, C, st);
    float* part = (float*)ws;
    size_t lds = (size_t)g.rows * C * 2 * sizeof(float);
    dim3 grid(g.nblk, N);
    TEM_ST_SWITCH(st, T, {
        if (g.vec == 4)
            hipLaunchKernelGGL((k_norm_partial<4, 0, T>), grid, dim3(g.threads), lds, (hipStream_t)stream, (const T*)x, x_ld,
                               (const T*)nullptr, (int64_t)0, V, C, G, nullptr, nullptr, g.cq, g.rows, g.vper, part);
        else
            hipLaunchKernelGGL((k_norm_partial<1, 0, T>), grid, dim3(g.threads), lds, (hipStream_t)stream, (const T*)x, x_ld,
                               (const T*)nullptr, (int64_t)0, V, C, G, nullptr, nullptr, g.cq, g.rows, g.vper, part);
    });
    hipLaunchKernelGGL(k_norm_finalize, dim3(N * G), dim3(norm_finalize_threads(g.nblk, C / G)), 0, (hipStream_t)stream, part,
                       g.nblk, V, C, G, gamma, beta, eps, mean, rstd, scale, shift);
    TEM_CHECK_LAUNCH("tem_norm_stats");
    return TEM_OK;
}
extern "C" int tem_norm_stats(const float* x, int64_t x_ld, int N, int64_t V, int C, int G, const float* gamma,
                              const float* beta, float eps, float* mean, float* rstd, float* scale, float* shift,
                              void* ws, int64_t ws_bytes, tem_stream_t stream) {
    return norm_stats_impl(x, x_ld, N, V, C, G, gamma, beta, eps, mean, rstd, scale, shift, ws, ws_bytes, 0, stream);
}
extern "C" int tem_norm_stats_st(const void* x, int64_t x_ld, int N, int64_t V, int C, int G, const float* gamma,
                                 const float* beta, float eps, float* mean, float* rstd, float* scale, float* shift,
                                 void* ws, int64_t ws_bytes, int st, tem_stream_t stream) {
    return norm_stats_impl(x, x_ld, N, V, C, G, gamma, beta, eps, mean, rstd, scale, shift, ws, ws_bytes, st, stream);
}

// Second stage only: the per-(sample, block, channel) partial sums (sum x, sum x^2) were written by the producer of x
// (tem_conv3d_fwd_stats), so x is not read again.  part: [N][nblk][C][2].
extern "C" int tem_norm_finalize_partials(const float* part, int64_t nblk, int N, int64_t V, int C, int G,
                                          const float* gamma, const float* beta, float eps, float* mean, float* rstd,
                                          float* scale, float* shift, tem_stream_t stream) {
    TEM_REQUIRE(part && mean && rstd && scale && shift, "tem_norm_finalize_partials: null pointer");
    TEM_REQUIRE(N > 0 && V > 0 && C > 0 && nblk > 0 && nblk < (1ll << 31) / 2, "tem_norm_finalize_partials: bad shape");
    TEM_REQUIRE(G > 0 && C % G == 0, "tem_norm_finalize_partials: C=%d not divisible by G=%d", C, G);
    hipLaunchKernelGGL(k_norm_finalize, dim3(N * G), dim3(norm_finalize_threads(nblk, C / G)), 0, (hipStream_t)stream, part,
                       (int)nblk, V, C, G, gamma, beta, eps, mean, rstd, scale, shift);
    TEM_CHECK_LAUNCH("tem_norm_finalize_partials");
    return TEM_OK;
}

// tem_norm_finalize_partials for a concatenated tensor: channels [0, CA) summarised by partA ([N][nblkA][CA][2], e.g.
// tem_upsample_stats), channels [CA, C) by partB ([N][nblkB][C-CA][2], e.g. tem_conv3d_fwd_stats of the skip tensor)
extern "C" int tem_norm_finalize_partials2(const float* partA, int64_t nblkA, int CA, const float* partB, int64_t nblkB,
                                           int N, int64_t V, int C, int G, const float* gamma, const float* beta,
                                           float eps, float* mean, float* rstd, float* scale, float* shift,
                                           tem_stream_t stream) {
    TEM_REQUIRE(partA && partB && mean && rstd && scale && shift, "tem_norm_finalize_partials2: null pointer");
    TEM_REQUIRE(N > 0 && V > 0 && C > 0 && CA > 0 && CA < C && nblkA > 0 && nblkB > 0 && nblkA < (1ll << 30) &&
                    nblkB < (1ll << 30),
                "tem_norm_finalize_partials2: bad shape");
    TEM_REQUIRE(G > 0 && C % G == 0 && CA % (C / G) == 0,
                "tem_norm_finalize_partials2: groups of %d channels must not straddle the split at %d", C / (G > 0 ? G : 1), CA);
    hipLaunchKernelGGL(k_norm_finalize2, dim3(N * G), dim3(norm_finalize_threads(nblkA > nblkB ? nblkA : nblkB, C / G)), 0,
                       (hipStream_t)stream, partA, (int)nblkA, CA, partB, (int)nblkB, V, C, G, gamma, beta, eps, mean, rstd, scale,
                       shift);
    TEM_CHECK_LAUNCH("tem_norm_finalize_partials2");
    return TEM_OK;
}

static int norm_bwd_impl(const void* gy, int64_t gy_ld, const void* x, int64_t x_ld, int N, int64_t V, int C,
                         int G, const float* gamma, const float* mean, const float* rstd, int relu_mask, void* gx,
                         int64_t gx_ld, float* dgamma, float* dbeta, const float* sums, float* coef_out, void* ws,
                         int64_t ws_bytes, tem_stream_t stream, int64_t sums_nblk, unsigned* amax, int st) {
    TEM_REQUIRE(gy && x && mean && rstd && (gx || coef_out) && ws, "tem_norm_bwd: null pointer");
    TEM_REQUIRE(st >= 0 && st <= 2, "tem_norm_bwd: unknown storage type %d", st);
    TEM_REQUIRE(sums_nblk >= 1 && sums_nblk < (1ll << 24), "tem_norm_bwd: bad number of partial rows");
    if (coef_out) gx_ld = C;
    TEM_REQUIRE(N > 0 && V > 0 && C > 0 && C <= NORM_MAX_C && x_ld >= C && gy_ld >= C && gx_ld >= C,
                "tem_norm_bwd: bad shape (C=%d)", C);
    TEM_REQUIRE(G > 0 && C % G == 0, "tem_norm_bwd: C=%d not divisible by G=%d", C, G);
    if (ws_bytes < tem_norm_ws(N, V, C)) {
        tem_set_error("tem_norm_bwd: workspace too small");
        return TEM_EWS;
    }
    NormGeom g = norm_geom(x, gy, x_ld, gy_ld, V, C, st);
    float* part = (float*)ws;
    float* coef = coef_out ? coef_out : part + (int64_t)N * NORM_MAX_BLOCKS * C * 2;
    size_t lds = (size_t)g.rows * C * 2 * sizeof(float);
    dim3 grid(g.nblk, N);
    int nblk = g.nblk;
    if (sums) {  // first stage delivered by the weight gradient (tem_conv3d_wgrad_sums): [N][1][C][2]
        part = const_cast<float*>(sums);
        nblk = (int)sums_nblk;   // [N][sums_nblk][C][2]; 1 for the weight gradient's sums
    } else {
        TEM_ST_SWITCH(st, T, {
            if (g.vec == 4)
                hipLaunchKernelGGL((k_norm_partial<4, 1, T>), grid, dim3(g.threads), lds, (hipStream_t)stream, (const T*)x, x_ld,
                                   (const T*)gy, gy_ld, V, C, G, mean, rstd, g.cq, g.rows, g.vper, part);
            else
                hipLaunchKernelGGL((k_norm_partial<1, 1, T>), grid, dim3(g.threads), lds, (hipStream_t)stream, (const T*)x, x_ld,
                                   (const T*)gy, gy_ld, V, C, G, mean, rstd, g.cq, g.rows, g.vper, part);
        });
    }
    hipLaunchKernelGGL(k_norm_bwd_finalize, dim3(N * G), dim3(256), 0, (hipStream_t)stream, part, nblk, V, C, G,
                       gamma, mean, rstd, coef);
    if (dgamma || dbeta)
        hipLaunchKernelGGL(k_norm_bwd_affine, dim3((unsigned)C), dim3(256), 0, (hipStream_t)stream, part,
                           N, nblk, C, dgamma, dbeta);
    if (coef_out) {  // the consumers of the gradient apply gx = a*gy - m1 - (x - mean)*m2r themselves
        TEM_CHECK_LAUNCH("tem_norm_bwd_coef");
        return TEM_OK;
    }
    const uintptr_t a4 = tem_st_align4(st);
    bool v4 = (C % 4 == 0) && gy_ld % 4 == 0 && x_ld % 4 == 0 && gx_ld % 4 == 0 && (uintptr_t)gy % a4 == 0 &&
              (uintptr_t)x % a4 == 0 && (uintptr_t)gx % a4 == 0;
    int64_t items = V * (v4 ? C / 4 : C);
    dim3 agrid(tem_grid_1d(items, 256, 2048), N);
    // gx is what a weight gradient reads next: max |gx| as a by-product when asked for (one atomic per 1024-thread block, see the kernel)
    const int threads = amax ? 1024 : 256;
    if (amax) agrid = dim3(tem_grid_1d(items, 1024, N > 0 ? (256 + N - 1) / N : 256), N);
    TEM_ST_SWITCH(st, T, {
        if (v4)
            hipLaunchKernelGGL((k_norm_bwd_apply<4, T>), agrid, dim3(threads), 0, (hipStream_t)stream, (const T*)gy, gy_ld,
                               (const T*)x, x_ld, (T*)gx, gx_ld, V, C, coef, relu_mask, amax);
        else
            hipLaunchKernelGGL((k_norm_bwd_apply<1, T>), agrid, dim3(threads), 0, (hipStream_t)stream, (const T*)gy, gy_ld,
                               (const T*)x, x_ld, (T*)gx, gx_ld, V, C, coef, relu_mask, amax);
    });
    TEM_CHECK_LAUNCH("tem_norm_bwd");
    return TEM_OK;
}

extern "C" int tem_norm_bwd(const float* gy, int64_t gy_ld, const float* x, int64_t x_ld, int N, int64_t V, int C,
                            int G, const float* gamma, const float* mean, const float* rstd, int relu_mask, float* gx,
                            int64_t gx_ld, float* dgamma, float* dbeta, void* ws, int64_t ws_bytes,
                            tem_stream_t stream) {
    return norm_bwd_impl(gy, gy_ld, x, x_ld, N, V, C, G, gamma, mean, rstd, relu_mask, gx, gx_ld, dgamma, dbeta, nullptr,
                         nullptr, ws, ws_bytes, stream, 1, nullptr, 0);
}

// tem_norm_bwd whose first stage -- sums[n][c] = (sum_v gy, sum_v gy * xn) -- was delivered by tem_conv3d_wgrad_sums
extern "C" int tem_norm_bwd_from_sums(const float* gy, int64_t gy_ld, const float* x, int64_t x_ld, int N, int64_t V,
                                      int C, int G, const float* gamma, const float* mean, const float* rstd,
                                      int relu_mask, float* gx, int64_t gx_ld, float* dgamma, float* dbeta,
                                      const float* sums, void* ws, int64_t ws_bytes, tem_stream_t stream) {
    TEM_REQUIRE(sums, "tem_norm_bwd_from_sums: null sums");
    return norm_bwd_impl(gy, gy_ld, x, x_ld, N, V, C, G, gamma, mean, rstd, relu_mask, gx, gx_ld, dgamma, dbeta, sums,
                         nullptr, ws, ws_bytes, stream, 1, nullptr, 0);
}

// ... was delivered as partial rows part[N][nblk][C][2] by the data gradient that wrote gy (TEM_BP_NORM_SUMS of tem_conv3d_fwd_ex)
extern "C" int tem_norm_bwd_from_partials(const float* gy, int64_t gy_ld, const float* x, int64_t x_ld, int N, int64_t V,
                                          int C, int G, const float* gamma, const float* mean, const float* rstd,
                                          int relu_mask, float* gx, int64_t gx_ld, float* dgamma, float* dbeta,
                                          const float* part, int64_t nblk, float* coef, void* ws, int64_t ws_bytes,
                                          tem_stream_t stream) {
    TEM_REQUIRE(part && (gx || coef), "tem_norm_bwd_from_partials: null pointer");
    return norm_bwd_impl(gy, gy_ld, x, x_ld, N, V, C, G, gamma, mean, rstd, relu_mask, coef ? nullptr : gx, gx_ld, dgamma, dbeta,
                         part, coef, ws, ws_bytes, stream, nblk, nullptr, 0);
}

// Reduction stage only: coef[n][c] = {a, m1, m2r, mean} with gx = a*gy - m1 - (x - mean)*m2r (and dgamma / dbeta).  The
// elementwise apply is left to the kernels that read the gradient next: tem_maxpool3d_bwd_norm (skip half of a decoder
// concat) and tem_upsample_bwd_norm (its upsampled half) -- the 3-tensor apply pass over the concat buffer disappears.
// sums: optional first stage from tem_conv3d_wgrad_sums (then gy / x are not read at all).
extern "C" int tem_norm_bwd_coef(const float* gy, int64_t gy_ld, const float* x, int64_t x_ld, int N, int64_t V, int C,
                                 int G, const float* gamma, const float* mean, const float* rstd, float* dgamma,
                                 float* dbeta, const float* sums, float* coef, void* ws, int64_t ws_bytes,
                                 tem_stream_t stream) {
    TEM_REQUIRE(coef, "tem_norm_bwd_coef: null coef");
    return norm_bwd_impl(gy, gy_ld, x, x_ld, N, V, C, G, gamma, mean, rstd, 0, nullptr, C, dgamma, dbeta, sums, coef, ws,
                         ws_bytes, stream, 1, nullptr, 0);
}

// All of the above for tensors of storage type st, every variant as explicit arguments:
//   part / part_nblk : first stage rows [N][part_nblk][C][2] delivered by a producer (tem_conv3d_wgrad_ex norm_sums: 1 row;
//                      tem_conv3d_fwd_ex norm sums: its nblk) or NULL = reduce gy and x here;
//   coef_out         : non-NULL = reduction only, write coef[N][C][4] (gx is not touched, may be NULL);
//   out_amax         : optional device word that receives max |gx| (see tem_maxpool3d_bwd_st).
extern "C" int tem_norm_bwd_st(const void* gy, int64_t gy_ld, const void* x, int64_t x_ld, int N, int64_t V, int C, int G,
                               const float* gamma, const float* mean, const float* rstd, int relu_mask, void* gx,
                               int64_t gx_ld, float* dgamma, float* dbeta, const float* part, int64_t part_nblk,
                               float* coef_out, unsigned* out_amax, void* ws, int64_t ws_bytes, int st, tem_stream_t stream) {
    return norm_bwd_impl(gy, gy_ld, x, x_ld, N, V, C, G, gamma, mean, rstd, coef_out ? 0 : relu_mask, coef_out ? nullptr : gx,
                         coef_out ? C : gx_ld, dgamma, dbeta, part, coef_out, ws, ws_bytes, stream, part ? part_nblk : 1,
                         coef_out ? nullptr : out_amax, st);
}
