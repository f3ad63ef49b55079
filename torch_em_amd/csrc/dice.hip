// dice.hip -- Dice score / loss (reference loss/dice.py:34-133) and the masked variant of
// LossWrapper + ApplyAndRemoveMask(masking_method="multiply") (loss/wrapper.py:84-87,129-152).
// Pure HBM-bound: prediction, target (and mask) are read ONCE in place through generic
// strides -- no flatten_samples() permute+copy (loss/dice.py:22-31), no p*m / t*m temporaries.
// Reduction: fp32 per-thread partials -> fp32 per-block partials -> fp64 per-channel sums.
#include "tem_common.h"

#define DICE_MAX_BLOCKS 2048

extern "C" int64_t tem_dice_ws(int N, int64_t V, int C) {
    (void)N;
    (void)V;
    return (int64_t)DICE_MAX_BLOCKS * C * 4 * (int64_t)sizeof(float);
}

// flags of the Dice family (reference loss/dice.py:136-253): TEM_DICE_LOGITS -- the prediction holds logits, the Dice
// terms use sigmoid(x) (DiceLossWithLogits, BCEDiceLossWithLogits); TEM_DICE_BCE -- a fourth per-channel sum carries the
// binary cross entropy (BCEDiceLoss: F.binary_cross_entropy with its log clamp at -100; with TEM_DICE_LOGITS:
// F.binary_cross_entropy_with_logits = max(x, 0) - x t + log1p(exp(-|x|)))
__device__ __forceinline__ float dice_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

// thread -> (channel c, voxel sub-row r); CFAST: consecutive threads walk channels (NDHWC
// prediction), else consecutive threads walk voxels (NCDHW prediction).  NS sums per channel: 3, or 4 with the BCE term.
template <bool CFAST>
__global__ void k_dice_partial(const float* __restrict__ p, int64_t p_sn, int64_t p_sc, int64_t p_sv,
                               const float* __restrict__ t, int64_t t_sn, int64_t t_sc, int64_t t_sv,
                               const float* __restrict__ mask, int N, int C, int64_t V, int rows, int64_t vper,
                               int nblk_per_n, float* __restrict__ part, int flags) {
    extern __shared__ float sh[];  // [rows][C][NS]
    const int NS = (flags & TEM_DICE_BCE) ? 4 : 3;
    const int n = blockIdx.x / nblk_per_n, b = blockIdx.x % nblk_per_n;
    int c, r;
    if (CFAST) {
        c = threadIdx.x % C;
        r = threadIdx.x / C;
    } else {
        r = threadIdx.x % rows;
        c = threadIdx.x / rows;
    }
    int64_t v0 = (int64_t)b * vper, v1 = v0 + vper;
    if (v1 > V) v1 = V;
    const float* pp = p + (int64_t)n * p_sn + (int64_t)c * p_sc;
    const float* tp = t + (int64_t)n * t_sn + (int64_t)c * t_sc;
    const float* mp = mask ? mask + (int64_t)n * t_sn + (int64_t)c * t_sc : nullptr;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int64_t v = v0 + r; v < v1; v += rows) {
        float pv = pp[v * p_sv], tv = tp[v * t_sv];
        if (flags) {   // launch-uniform
            const float xv = pv;
            if (flags & TEM_DICE_LOGITS) pv = dice_sigmoid(xv);
            if (flags & TEM_DICE_BCE) {
                if (flags & TEM_DICE_LOGITS)
                    s3 += fmaxf(xv, 0.f) - xv * tv + log1pf(expf(-fabsf(xv)));
                else
                    s3 -= tv * fmaxf(logf(pv), -100.f) + (1.f - tv) * fmaxf(logf(1.f - pv), -100.f);
            }
        }
        if (mp) {
            float m = mp[v * t_sv];
            pv *= m;
            tv *= m;
        }
        s0 = fmaf(pv, tv, s0);
        s1 = fmaf(pv, pv, s1);
        s2 = fmaf(tv, tv, s2);
    }
    float* my = sh + ((int64_t)r * C + c) * NS;
    my[0] = s0;
    my[1] = s1;
    my[2] = s2;
    if (NS == 4) my[3] = s3;
    __syncthreads();
    if (r == 0) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int rr = 0; rr < rows; ++rr) {
            const float* o = sh + ((int64_t)rr * C + c) * NS;
            a0 += o[0];
            a1 += o[1];
            a2 += o[2];
            if (NS == 4) a3 += o[3];
        }
        float* o = part + ((int64_t)blockIdx.x * C + c) * NS;
        o[0] = a0;
        o[1] = a1;
        o[2] = a2;
        if (NS == 4) o[3] = a3;
    }
}

// The same sums with one VOXEL per thread and a loop over the channels (C <= DICE_VOX_C), round 4.  The kernel above gives a
// channel to a lane: with a channels-last prediction and a channel-first target (what UNetFunction and the loaders
// produce) one of the two tensors is read with lanes a volume apart -- 13 cache lines per load instruction for the 12
// affinity channels of cfg 3, 1.5 TB/s.  Here both are coalesced: a lane reads its voxel's C contiguous prediction values
// (16-byte loads when C % 4 == 0) and, per channel, the wave reads 64 consecutive target values.
#define DICE_VOX_C 16
template <int VECP>   // prediction loads: 4 = float4 (p_sc == 1, C % 4 == 0, aligned), 2 = float2 (C == 2), 1 = scalar / any strides
__global__ __launch_bounds__(256) void k_dice_partial_vox(const float* __restrict__ p, int64_t p_sn, int64_t p_sc, int64_t p_sv,
                                                          const float* __restrict__ t, int64_t t_sn, int64_t t_sc, int64_t t_sv,
                                                          const float* __restrict__ mask, int N, int C, int64_t V, int64_t vper,
                                                          int nblk_per_n, float* __restrict__ part, int flags) {
    __shared__ float shv[4][DICE_VOX_C][4];
    const int NS = (flags & TEM_DICE_BCE) ? 4 : 3;
    const int n = blockIdx.x / nblk_per_n, b = blockIdx.x % nblk_per_n;
    (void)vper;
    const float* pn = p + (int64_t)n * p_sn;
    const float* tn = t + (int64_t)n * t_sn;
    const float* mn = mask ? mask + (int64_t)n * t_sn : nullptr;
    float s0[DICE_VOX_C], s1[DICE_VOX_C], s2[DICE_VOX_C], s3[DICE_VOX_C];
#pragma unroll
    for (int c = 0; c < DICE_VOX_C; ++c) s0[c] = s1[c] = s2[c] = s3[c] = 0.f;
    // tiles of 256 voxels dealt round-robin to the workgroups of a sample: with one contiguous range per workgroup (ranges
    // a power of two apart, the C target planes a power of two apart) all workgroups walked the same few HBM channels in
    // lockstep -- 461 us on cfg 3 where the gradient kernel (grid-stride, same loads plus a store) takes 358
    for (int64_t vb = (int64_t)b * 256 + threadIdx.x; vb < V; vb += (int64_t)nblk_per_n * 256) {
        float pq[DICE_VOX_C], tq[DICE_VOX_C], mq[DICE_VOX_C];
        const float* pv_ = pn + vb * p_sv;
        if constexpr (VECP == 4) {
#pragma unroll
            for (int c = 0; c < DICE_VOX_C; c += 4)
                if (c < C) {
                    const float4 q = *reinterpret_cast<const float4*>(pv_ + c);
                    pq[c] = q.x; pq[c + 1] = q.y; pq[c + 2] = q.z; pq[c + 3] = q.w;
                }
        } else if constexpr (VECP == 2) {
            const float2 q = *reinterpret_cast<const float2*>(pv_);
            pq[0] = q.x; pq[1] = q.y;
        } else {
#pragma unroll
            for (int c = 0; c < DICE_VOX_C; ++c)
                if (c < C) pq[c] = pv_[c * p_sc];
        }
#pragma unroll
        for (int c = 0; c < DICE_VOX_C; ++c)
            if (c < C) {
                tq[c] = tn[c * t_sc + vb * t_sv];
                mq[c] = mn ? mn[c * t_sc + vb * t_sv] : 1.f;
            }
#pragma unroll
        for (int c = 0; c < DICE_VOX_C; ++c)
            if (c < C) {
                float pv = pq[c], tv = tq[c];
                if (flags) {   // launch-uniform
                    const float xv = pv;
                    if (flags & TEM_DICE_LOGITS) pv = dice_sigmoid(xv);
                    if (flags & TEM_DICE_BCE) {
                        if (flags & TEM_DICE_LOGITS)
                            s3[c] += fmaxf(xv, 0.f) - xv * tv + log1pf(expf(-fabsf(xv)));
                        else
                            s3[c] -= tv * fmaxf(logf(pv), -100.f) + (1.f - tv) * fmaxf(logf(1.f - pv), -100.f);
                    }
                }
                if (mn) {
                    pv *= mq[c];
                    tv *= mq[c];
                }
                s0[c] = fmaf(pv, tv, s0[c]);
                s1[c] = fmaf(pv, pv, s1[c]);
                s2[c] = fmaf(tv, tv, s2[c]);
            }
    }
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < DICE_VOX_C; ++c)
        if (c < C) {
            const float a0 = tem_wave_sum(s0[c]), a1 = tem_wave_sum(s1[c]), a2 = tem_wave_sum(s2[c]);
            const float a3 = NS == 4 ? tem_wave_sum(s3[c]) : 0.f;
            if (lane == 0) {
                shv[wv][c][0] = a0;
                shv[wv][c][1] = a1;
                shv[wv][c][2] = a2;
                shv[wv][c][3] = a3;
            }
        }
    __syncthreads();
    for (int i = threadIdx.x; i < C * NS; i += 256) {
        const int c = i / NS, k = i % NS;
        part[((int64_t)blockIdx.x * C + c) * NS + k] = shv[0][c][k] + shv[1][c][k] + shv[2][c][k] + shv[3][c][k];
    }
}

__global__ __launch_bounds__(256) void k_dice_reduce(const float* __restrict__ part, int nblk, int C,
                                                     double* __restrict__ sums, int NS) {
    const int c = blockIdx.x;
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < nblk; b += 256) {
        const float* o = part + ((int64_t)b * C + c) * NS;
        a[0] += (double)o[0];
        a[1] += (double)o[1];
        a[2] += (double)o[2];
        if (NS == 4) a[3] += (double)o[3];
    }
    __shared__ double sh[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        a[k] = tem_wave_sum_d(a[k]);
        if ((threadIdx.x & 63) == 0) sh[k][threadIdx.x >> 6] = a[k];
    }
    __syncthreads();
    if (threadIdx.x < NS) sums[c * NS + threadIdx.x] = sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3];
}

static int dice_sums_impl(const float* p, int64_t p_sn, int64_t p_sc, int64_t p_sv, const float* t, int64_t t_sn,
                          int64_t t_sc, int64_t t_sv, const float* mask, int N, int C, int64_t V, double* sums,
                          void* ws, int64_t ws_bytes, int flags, tem_stream_t stream) {
    TEM_REQUIRE(p && t && sums && ws, "tem_dice_sums: null pointer");
    TEM_REQUIRE((flags & ~(TEM_DICE_LOGITS | TEM_DICE_BCE)) == 0 && (!(flags & TEM_DICE_BCE) || !mask),
                "tem_dice_sums2: unknown flags, or the BCE term together with a mask");
    const int NS = (flags & TEM_DICE_BCE) ? 4 : 3;
    TEM_REQUIRE(N > 0 && C > 0 && C <= 1024 && V > 0, "tem_dice_sums: bad shape (C=%d)", C);
    if (ws_bytes < tem_dice_ws(N, V, C)) {
        tem_set_error("tem_dice_sums: workspace too small");
        return TEM_EWS;
    }
    int rows = 256 / C;
    if (rows < 1) rows = 1;
    if (C <= DICE_VOX_C && tem_option(TEM_OPT_DICE_VOX)) rows = 256;   // one voxel per thread (k_dice_partial_vox)
    int threads = C <= DICE_VOX_C && tem_option(TEM_OPT_DICE_VOX) ? 256 : rows * C;
    int64_t nb = tem_cdiv(V, (int64_t)rows * 8);
    int64_t maxb = DICE_MAX_BLOCKS / N;
    if (maxb < 1) {
        tem_set_error("tem_dice_sums: batch too large (N=%d)", N);
        return TEM_EINVAL;
    }
    if (nb > maxb) nb = maxb;
    int64_t vper = tem_cdiv(V, nb);
    int nblk = (int)nb * N;
    size_t lds = (size_t)rows * C * NS * sizeof(float);
    float* part = (float*)ws;
    bool cfast = (p_sc == 1);
    if (C <= DICE_VOX_C && tem_option(TEM_OPT_DICE_VOX)) {
        const bool al = cfast && p_sn % 4 == 0 && p_sv % 4 == 0 && ((uintptr_t)p % 16 == 0);
        if (al && C % 4 == 0)
            hipLaunchKernelGGL((k_dice_partial_vox<4>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, p, p_sn, p_sc, p_sv, t, t_sn,
                               t_sc, t_sv, mask, N, C, V, vper, (int)nb, part, flags);
        else if (cfast && C == 2 && p_sn % 2 == 0 && p_sv == 2 && ((uintptr_t)p % 8 == 0))
            hipLaunchKernelGGL((k_dice_partial_vox<2>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, p, p_sn, p_sc, p_sv, t, t_sn,
                               t_sc, t_sv, mask, N, C, V, vper, (int)nb, part, flags);
        else
            hipLaunchKernelGGL((k_dice_partial_vox<1>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, p, p_sn, p_sc, p_sv, t, t_sn,
                               t_sc, t_sv, mask, N, C, V, vper, (int)nb, part, flags);
    } else if (cfast)
        hipLaunchKernelGGL((k_dice_partial<true>), dim3(nblk), dim3(threads), lds, (hipStream_t)stream, p, p_sn, p_sc,
                           p_sv, t, t_sn, t_sc, t_sv, mask, N, C, V, rows, vper, (int)nb, part, flags);
    else
        hipLaunchKernelGGL((k_dice_partial<false>), dim3(nblk), dim3(threads), lds, (hipStream_t)stream, p, p_sn, p_sc,
                           p_sv, t, t_sn, t_sc, t_sv, mask, N, C, V, rows, vper, (int)nb, part, flags);
    hipLaunchKernelGGL(k_dice_reduce, dim3(C), dim3(256), 0, (hipStream_t)stream, part, nblk, C, sums, NS);
    TEM_CHECK_LAUNCH("tem_dice_sums");
    return TEM_OK;
}

extern "C" int tem_dice_sums(const float* p, int64_t p_sn, int64_t p_sc, int64_t p_sv, const float* t, int64_t t_sn,
                             int64_t t_sc, int64_t t_sv, const float* mask, int N, int C, int64_t V, double* sums,
                             void* ws, int64_t ws_bytes, tem_stream_t stream) {
    return dice_sums_impl(p, p_sn, p_sc, p_sv, t, t_sn, t_sc, t_sv, mask, N, C, V, sums, ws, ws_bytes, 0, stream);
}

// sums[C][3] as tem_dice_sums, or [C][4] with TEM_DICE_BCE (4th column: the sum of the per-element cross entropy)
extern "C" int tem_dice_sums2(const float* p, int64_t p_sn, int64_t p_sc, int64_t p_sv, const float* t, int64_t t_sn,
                              int64_t t_sc, int64_t t_sv, int N, int C, int64_t V, double* sums, void* ws,
                              int64_t ws_bytes, int flags, tem_stream_t stream) {
    return dice_sums_impl(p, p_sn, p_sc, p_sv, t, t_sn, t_sc, t_sv, nullptr, N, C, V, sums, ws, ws_bytes, flags, stream);
}

// ---------------------------------------------------------------------------
// finalize: sums -> score/loss (+ the per-channel coefficients of d out / d p)
//   score_c = 2*num/max(den,eps)  (loss/dice.py:65-67), out_c = invert ? 1-score_c : score_c
//   reduce: 0 none, 1 sum, 2 mean, 3 max, 4 min (loss/dice.py:71-84)
//   d out_c / d p = ca_c * t + cb_c * p
// ---------------------------------------------------------------------------
__global__ void k_dice_finalize(const double* __restrict__ sums, int C, double eps, int channelwise, int invert,
                                int reduce, float* __restrict__ out, float* __restrict__ ca, float* __restrict__ cb, int NS) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double sgn = invert ? -1.0 : 1.0;
    if (!channelwise) {
        double num = 0.0, den = 0.0;
        for (int c = 0; c < C; ++c) {
            num += sums[c * NS];
            den += sums[c * NS + 1] + sums[c * NS + 2];
        }
        double cd = den < eps ? eps : den;
        double score = 2.0 * num / cd;
        out[0] = (float)(invert ? 1.0 - score : score);
        double a = sgn * 2.0 / cd, b = (den >= eps) ? -sgn * 4.0 * num / (cd * cd) : 0.0;
        for (int c = 0; c < C; ++c) {
            ca[c] = (float)a;
            cb[c] = (float)b;
        }
        return;
    }
    double acc = 0.0;
    int arg = 0;
    double best = 0.0;
    for (int c = 0; c < C; ++c) {
        double num = sums[c * NS], den = sums[c * NS + 1] + sums[c * NS + 2];
        double cd = den < eps ? eps : den;
        double score = 2.0 * num / cd;
        double val = invert ? 1.0 - score : score;
        ca[c] = (float)(sgn * 2.0 / cd);
        cb[c] = (float)((den >= eps) ? -sgn * 4.0 * num / (cd * cd) : 0.0);
        if (reduce == 0) out[c] = (float)val;
        acc += val;
        if (c == 0 || (reduce == 3 && val > best) || (reduce == 4 && val < best)) {
            best = val;
            arg = c;
        }
    }
    if (reduce == 1) out[0] = (float)acc;
    if (reduce == 2) {
        out[0] = (float)(acc / C);
        for (int c = 0; c < C; ++c) {
            ca[c] /= (float)C;
            cb[c] /= (float)C;
        }
    }
    if (reduce == 3 || reduce == 4) {
        out[0] = (float)best;
        for (int c = 0; c < C; ++c)
            if (c != arg) {
                ca[c] = 0.f;
                cb[c] = 0.f;
            }
    }
}

static int dice_finalize_impl(const double* sums, int C, double eps, int channelwise, int invert, int reduce, float* out,
                              float* ca, float* cb, int NS, tem_stream_t stream) {
    TEM_REQUIRE(sums && out && ca && cb && C > 0, "tem_dice_finalize: bad arguments");
    TEM_REQUIRE(reduce >= 0 && reduce <= 4, "tem_dice_finalize: Unsupported channel reduction %d", reduce);
    hipLaunchKernelGGL(k_dice_finalize, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, C, eps, channelwise, invert,
                       reduce, out, ca, cb, NS);
    TEM_CHECK_LAUNCH("tem_dice_finalize");
    return TEM_OK;
}
extern "C" int tem_dice_finalize(const double* sums, int C, double eps, int channelwise, int invert, int reduce,
                                 float* out, float* ca, float* cb, tem_stream_t stream) {
    return dice_finalize_impl(sums, C, eps, channelwise, invert, reduce, out, ca, cb, 3, stream);
}
// tem_dice_finalize on sums with `ncol` (3 or 4) columns per channel (tem_dice_sums2)
extern "C" int tem_dice_finalize2(const double* sums, int ncol, int C, double eps, int channelwise, int invert, int reduce,
                                  float* out, float* ca, float* cb, tem_stream_t stream) {
    TEM_REQUIRE(ncol == 3 || ncol == 4, "tem_dice_finalize2: ncol must be 3 or 4");
    return dice_finalize_impl(sums, C, eps, channelwise, invert, reduce, out, ca, cb, ncol, stream);
}

// gp = gout * (ca*t + cb*p) * mask   (p, t masked first)
template <bool CFAST>
__global__ __launch_bounds__(256) void k_dice_grad(const float* __restrict__ p, int64_t p_sn, int64_t p_sc,
                                                   int64_t p_sv, const float* __restrict__ t, int64_t t_sn,
                                                   int64_t t_sc, int64_t t_sv, const float* __restrict__ mask,
                                                   const float* __restrict__ ca, const float* __restrict__ cb,
                                                   const float* __restrict__ gout, int gout_per_channel,
                                                   float* __restrict__ gp, int64_t g_sn, int64_t g_sc, int64_t g_sv,
                                                   int C, int64_t V, int flags, float w_dice, float w_bce) {
    const int n = blockIdx.y;
    const int64_t items = V * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < items; i += (int64_t)gridDim.x * 256) {
        int c;
        int64_t v;
        if (CFAST) {
            v = i / C;
            c = (int)(i - v * C);
        } else {
            c = (int)(i / V);
            v = i - (int64_t)c * V;
        }
        float pv = p[(int64_t)n * p_sn + (int64_t)c * p_sc + v * p_sv];
        float tv = t[(int64_t)n * t_sn + (int64_t)c * t_sc + v * t_sv];
        float m = 1.f;
        if (mask) {
            m = mask[(int64_t)n * t_sn + (int64_t)c * t_sc + v * t_sv];
            pv *= m;
            tv *= m;
        }
        float go = gout ? gout[gout_per_channel ? c : 0] : 1.f;
        if (flags) {   // launch-uniform: the logits / BCE members of the family (no mask there)
            const float xv = pv;
            if (flags & TEM_DICE_LOGITS) pv = dice_sigmoid(xv);
            float d = w_dice * (ca[c] * tv + cb[c] * pv);
            if (flags & TEM_DICE_LOGITS) d *= pv * (1.f - pv);                       // d sigmoid / d x
            if (flags & TEM_DICE_BCE)   // aten binary_cross_entropy_backward: (p - t) / max(p (1 - p), 1e-12); with logits: sigmoid(x) - t
                d += w_bce * ((flags & TEM_DICE_LOGITS) ? (pv - tv) : (pv - tv) / fmaxf(pv * (1.f - pv), 1e-12f));
            gp[(int64_t)n * g_sn + (int64_t)c * g_sc + v * g_sv] = go * d;
            continue;
        }
        gp[(int64_t)n * g_sn + (int64_t)c * g_sc + v * g_sv] = go * (ca[c] * tv + cb[c] * pv) * m;
    }
}

// One voxel per thread (C <= DICE_VOX_C): see k_dice_partial_vox.  The gradient has the prediction's layout, so with a
// channels-last prediction both its load and the store are C contiguous floats per lane.
template <int VECP>
__global__ __launch_bounds__(256) void k_dice_grad_vox(const float* __restrict__ p, int64_t p_sn, int64_t p_sc, int64_t p_sv,
                                                       const float* __restrict__ t, int64_t t_sn, int64_t t_sc, int64_t t_sv,
                                                       const float* __restrict__ mask, const float* __restrict__ ca,
                                                       const float* __restrict__ cb, const float* __restrict__ gout,
                                                       int gout_per_channel, float* __restrict__ gp, int64_t g_sn, int64_t g_sc,
                                                       int64_t g_sv, int C, int64_t V, int flags, float w_dice, float w_bce) {
    const int n = blockIdx.y;
    const float* pn = p + (int64_t)n * p_sn;
    const float* tn = t + (int64_t)n * t_sn;
    const float* mn = mask ? mask + (int64_t)n * t_sn : nullptr;
    float* gn = gp + (int64_t)n * g_sn;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (int64_t)gridDim.x * 256) {
        float pq[DICE_VOX_C], tq[DICE_VOX_C], mq[DICE_VOX_C], gq[DICE_VOX_C];
        const float* pv_ = pn + v * p_sv;
        if constexpr (VECP == 4) {
#pragma unroll
            for (int c = 0; c < DICE_VOX_C; c += 4)
                if (c < C) {
                    const float4 q = *reinterpret_cast<const float4*>(pv_ + c);
                    pq[c] = q.x; pq[c + 1] = q.y; pq[c + 2] = q.z; pq[c + 3] = q.w;
                }
        } else if constexpr (VECP == 2) {
            const float2 q = *reinterpret_cast<const float2*>(pv_);
            pq[0] = q.x; pq[1] = q.y;
        } else {
#pragma unroll
            for (int c = 0; c < DICE_VOX_C; ++c)
                if (c < C) pq[c] = pv_[c * p_sc];
        }
#pragma unroll
        for (int c = 0; c < DICE_VOX_C; ++c)
            if (c < C) {
                tq[c] = tn[c * t_sc + v * t_sv];
                mq[c] = mn ? mn[c * t_sc + v * t_sv] : 1.f;
            }
#pragma unroll
        for (int c = 0; c < DICE_VOX_C; ++c)
            if (c < C) {
                float pv = pq[c], tv = tq[c];
                const float m = mq[c];
                if (mn) {
                    pv *= m;
                    tv *= m;
                }
                const float go = gout ? gout[gout_per_channel ? c : 0] : 1.f;   // wave-uniform addresses: scalar loads
                if (flags) {   // launch-uniform: the logits / BCE members of the family (no mask there)
                    const float xv = pv;
                    if (flags & TEM_DICE_LOGITS) pv = dice_sigmoid(xv);
                    float d = w_dice * (ca[c] * tv + cb[c] * pv);
                    if (flags & TEM_DICE_LOGITS) d *= pv * (1.f - pv);
                    if (flags & TEM_DICE_BCE)
                        d += w_bce * ((flags & TEM_DICE_LOGITS) ? (pv - tv) : (pv - tv) / fmaxf(pv * (1.f - pv), 1e-12f));
                    gq[c] = go * d;
                } else {
                    gq[c] = go * (ca[c] * tv + cb[c] * pv) * m;
                }
            }
        float* gv_ = gn + v * g_sv;
        if constexpr (VECP == 4) {
#pragma unroll
            for (int c = 0; c < DICE_VOX_C; c += 4)
                if (c < C) *reinterpret_cast<float4*>(gv_ + c) = make_float4(gq[c], gq[c + 1], gq[c + 2], gq[c + 3]);
        } else if constexpr (VECP == 2) {
            *reinterpret_cast<float2*>(gv_) = make_float2(gq[0], gq[1]);
        } else {
#pragma unroll
            for (int c = 0; c < DICE_VOX_C; ++c)
                if (c < C) gv_[c * g_sc] = gq[c];
        }
    }
}

static int dice_grad_impl(const float* p, int64_t p_sn, int64_t p_sc, int64_t p_sv, const float* t, int64_t t_sn,
                          int64_t t_sc, int64_t t_sv, const float* mask, const float* ca, const float* cb,
                          const float* gout, int gout_per_channel, float* gp, int64_t g_sn, int64_t g_sc,
                          int64_t g_sv, int N, int C, int64_t V, int flags, float w_dice, float w_bce, tem_stream_t stream) {
    TEM_REQUIRE(p && t && ca && cb && gp && N > 0 && C > 0 && V > 0, "tem_dice_grad: bad arguments");
    TEM_REQUIRE((flags & ~(TEM_DICE_LOGITS | TEM_DICE_BCE)) == 0 && (!flags || !mask), "tem_dice_grad2: unknown flags, or flags with a mask");
    dim3 grid(tem_grid_1d(V * C, 256, 2048), N);
    if (C <= DICE_VOX_C && tem_option(TEM_OPT_DICE_VOX)) {
        const dim3 vgrid(tem_grid_1d(V, 256, 4096), N);
        // vector accesses: prediction AND gradient channels-last with the same aligned pitch
        const bool cl = p_sc == 1 && g_sc == 1;
        const bool al4 = cl && C % 4 == 0 && p_sn % 4 == 0 && p_sv % 4 == 0 && g_sn % 4 == 0 && g_sv % 4 == 0 &&
                         ((uintptr_t)p % 16 == 0) && ((uintptr_t)gp % 16 == 0);
        const bool al2 = cl && C == 2 && p_sv == 2 && g_sv == 2 && p_sn % 2 == 0 && g_sn % 2 == 0 && ((uintptr_t)p % 8 == 0) &&
                         ((uintptr_t)gp % 8 == 0);
#define TEM_DGV(VP)                                                                                                            \
    hipLaunchKernelGGL((k_dice_grad_vox<VP>), vgrid, dim3(256), 0, (hipStream_t)stream, p, p_sn, p_sc, p_sv, t, t_sn, t_sc, t_sv, \
                       mask, ca, cb, gout, gout_per_channel, gp, g_sn, g_sc, g_sv, C, V, flags, w_dice, w_bce)
        if (al4) TEM_DGV(4);
        else if (al2) TEM_DGV(2);
        else TEM_DGV(1);
#undef TEM_DGV
    } else if (p_sc == 1)
        hipLaunchKernelGGL((k_dice_grad<true>), grid, dim3(256), 0, (hipStream_t)stream, p, p_sn, p_sc, p_sv, t, t_sn,
                           t_sc, t_sv, mask, ca, cb, gout, gout_per_channel, gp, g_sn, g_sc, g_sv, C, V, flags, w_dice, w_bce);
    else
        hipLaunchKernelGGL((k_dice_grad<false>), grid, dim3(256), 0, (hipStream_t)stream, p, p_sn, p_sc, p_sv, t, t_sn,
                           t_sc, t_sv, mask, ca, cb, gout, gout_per_channel, gp, g_sn, g_sc, g_sv, C, V, flags, w_dice, w_bce);
    TEM_CHECK_LAUNCH("tem_dice_grad");
    return TEM_OK;
}
extern "C" int tem_dice_grad(const float* p, int64_t p_sn, int64_t p_sc, int64_t p_sv, const float* t, int64_t t_sn,
                             int64_t t_sc, int64_t t_sv, const float* mask, const float* ca, const float* cb,
                             const float* gout, int gout_per_channel, float* gp, int64_t g_sn, int64_t g_sc,
                             int64_t g_sv, int N, int C, int64_t V, tem_stream_t stream) {
    return dice_grad_impl(p, p_sn, p_sc, p_sv, t, t_sn, t_sc, t_sv, mask, ca, cb, gout, gout_per_channel, gp, g_sn, g_sc,
                          g_sv, N, C, V, 0, 1.f, 0.f, stream);
}
// gp = gout * (w_dice * dDice/dx + w_bce * dBCE_sum/dx): the gradient of alpha * dice + beta * mean-BCE w.r.t. the
// prediction (probabilities, or logits with TEM_DICE_LOGITS), w_dice = alpha, w_bce = beta / (N C V)
extern "C" int tem_dice_grad2(const float* p, int64_t p_sn, int64_t p_sc, int64_t p_sv, const float* t, int64_t t_sn,
                              int64_t t_sc, int64_t t_sv, const float* ca, const float* cb, const float* gout,
                              int gout_per_channel, float* gp, int64_t g_sn, int64_t g_sc, int64_t g_sv, int N, int C,
                              int64_t V, int flags, float w_dice, float w_bce, tem_stream_t stream) {
    return dice_grad_impl(p, p_sn, p_sc, p_sv, t, t_sn, t_sc, t_sv, nullptr, ca, cb, gout, gout_per_channel, gp, g_sn,
                          g_sc, g_sv, N, C, V, flags, w_dice, w_bce, stream);
}
