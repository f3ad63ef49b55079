// spoco.hip -- SPOCO / contrastive embedding loss kernels.
//
// Reference: torch_em/loss/spoco_loss.py (compute_cluster_means :16-33, ContrastiveLossBase :117-298,
// ExtendedContrastiveLoss :301-430, SPOCOLoss :433-566), loss/contrastive_impl.py:14-129,
// loss/affinity_side_loss.py:9-172.  The reference runs these as dozens of full-volume torch ops plus a Python loop
// over instances; here every term is ONE or TWO streaming passes over the embeddings ([E, V] planes, voxel-fastest,
// read in place through a channel stride).
//
// Determinism: no floating-point atomics anywhere.
//   * per-label segment sums (cluster means, and the pull-term's per-label gradient sums) are accumulated in 64-bit
//     FIXED POINT (2^-28 resolution; integer adds commute, so LDS/global atomics give a bit-reproducible result);
//     a wave whose 64 voxels share one label first reduces with a fixed butterfly and adds once;
//   * every scalar is a two-stage reduction (fp32 thread/wave partials -> per-block partials -> fp64 in index order).
// All kernels are HBM/L2 streaming kernels (integer/float byte work, no GEMM shape): the roofline is HBM bandwidth.
#include "tem_common.h"

#define SP_FIX 268435456.0f  // 2^28
#define SP_MAXB 1024         // block-partial slots of the 1-D streaming kernels
#define SP_IT 8              // instances per tile of the V x C kernels
#define SP_NCH 8             // chunks per slice of the per-slice kernels
#define SP_MAXA 64           // anchors
#define SP_MAXK 32           // affinity offsets

typedef unsigned long long u64;

__device__ __forceinline__ void fix_add(u64* p, float v) {
    atomicAdd(p, (u64)(long long)__float2ll_rn(v * SP_FIX));
}
__device__ __forceinline__ float fix_get(u64 v) { return (float)((double)(long long)v * (1.0 / 268435456.0)); }

__device__ __forceinline__ long long shfl_xor_ll(long long v, int o) {
    int lo = __shfl_xor((int)(v & 0xffffffffll), o, 64), hi = __shfl_xor((int)(v >> 32), o, 64);
    return ((long long)hi << 32) | (unsigned int)lo;
}

// block sum of NV floats per thread; result valid in the first NV threads (as out[i]); deterministic order.
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* sh /*[nwaves*NV]*/, float* out /*[NV] in LDS*/) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float s = tem_wave_sum(v[i]);
        if (lane == 0) sh[wv * NV + i] = s;
    }
    __syncthreads();
    if ((int)threadIdx.x < NV) {
        float s = 0.f;
        for (int w = 0; w < nw; ++w) s += sh[w * NV + threadIdx.x];
        out[threadIdx.x] = s;
    }
    __syncthreads();
}

// Sum 8 per-lane values over the 64 lanes with 10 shuffles instead of 48: each butterfly step keeps half of the values
// and hands the other half to the partner lane (a transposing reduction).  Afterwards EVERY lane holds the wave total
// of value index reduce8_index(lane).  Fixed order => deterministic.
__device__ __forceinline__ int reduce8_index(int lane) { return (lane & 1) * 4 + ((lane >> 1) & 1) * 2 + ((lane >> 2) & 1); }
__device__ __forceinline__ float wave_reduce8(const float* v, int lane) {
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
    float w[4], u[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float send = b0 ? v[i] : v[i + 4], keep = b0 ? v[i + 4] : v[i];
        w[i] = keep + __shfl_xor(send, 1, 64);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float send = b1 ? w[i] : w[i + 2], keep = b1 ? w[i + 2] : w[i];
        u[i] = keep + __shfl_xor(send, 2, 64);
    }
    float t = (b2 ? u[1] : u[0]) + __shfl_xor(b2 ? u[0] : u[1], 4, 64);
    t += __shfl_xor(t, 8, 64);
    t += __shfl_xor(t, 16, 64);
    t += __shfl_xor(t, 32, 64);
    return t;
}

// deterministic block sum of one double per thread (butterfly inside a wave, waves in index order); valid in thread 0
__device__ __forceinline__ double block_sum_d(double v, double* sh /*[nwaves]*/) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    v = tem_wave_sum_d(v);
    __syncthreads();
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < nw; ++w) s += sh[w];
    return s;
}

// ------------------------------------------------------------------------------------------------
// label range
// ------------------------------------------------------------------------------------------------
__global__ void k_label_range(const int64_t* __restrict__ lbl, int64_t V, long long* __restrict__ part) {
    long long mn = INT64_MAX, mx = INT64_MIN;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (int64_t)gridDim.x * blockDim.x) {
        long long l = lbl[v];
        mn = l < mn ? l : mn;
        mx = l > mx ? l : mx;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        long long a = shfl_xor_ll(mn, o), b = shfl_xor_ll(mx, o);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    __shared__ long long sh[2][16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) {
        sh[0][wv] = mn;
        sh[1][wv] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {
            mn = sh[0][w] < mn ? sh[0][w] : mn;
            mx = sh[1][w] > mx ? sh[1][w] : mx;
        }
        part[blockIdx.x * 2] = mn;
        part[blockIdx.x * 2 + 1] = mx;
    }
}
__global__ __launch_bounds__(256) void k_label_range_final(const long long* __restrict__ part, int nb,
                                                            int64_t* __restrict__ out) {
    long long mn = INT64_MAX, mx = INT64_MIN;
    for (int b = threadIdx.x; b < nb; b += blockDim.x) {
        mn = part[b * 2] < mn ? part[b * 2] : mn;
        mx = part[b * 2 + 1] > mx ? part[b * 2 + 1] : mx;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        long long a = shfl_xor_ll(mn, o), b = shfl_xor_ll(mx, o);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    __shared__ long long sh[2][4];
    if ((threadIdx.x & 63) == 0) {
        sh[0][threadIdx.x >> 6] = mn;
        sh[1][threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            mn = sh[0][w] < mn ? sh[0][w] : mn;
            mx = sh[1][w] > mx ? sh[1][w] : mx;
        }
        out[0] = mn;
        out[1] = mx;
    }
}

extern "C" int64_t tem_spoco_ws(int C, int E, int64_t V, int nz, int A, int K) {
    (void)V;
    int64_t b = 0;
    b += (int64_t)C * (E + 1) * 8 * 3;                         // fixed-point accumulators (means, pull S, push)
    b += (int64_t)SP_MAXB * 8 * 4;                             // scalar block partials
    b += (int64_t)C * (2 * E + 4) * 8;                         // row values / combined dmeans
    b += (int64_t)nz * (C > 0 ? C : 1) * SP_NCH * 4 * 4;       // instance-dice partials
    b += (int64_t)nz * SP_NCH * (3 + (int64_t)(A > 0 ? A : 1) * E) * 4 + (int64_t)nz * 16 + (int64_t)A * E * 8;
    b += (int64_t)SP_MAXB * (K > 0 ? K : 1) * 3 * 4 + (int64_t)K * 24;
    return b + 4096;
}

extern "C" int tem_label_range(const int64_t* lbl, int64_t V, int64_t* out_minmax, void* ws, int64_t ws_bytes,
                               tem_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    TEM_REQUIRE(lbl && out_minmax && ws && V > 0, "tem_label_range: null pointer or empty input");
    const int nb = tem_grid_1d(V, 256, SP_MAXB);
    TEM_REQUIRE(ws_bytes >= (int64_t)nb * 16, "tem_label_range: workspace too small");
    hipLaunchKernelGGL(k_label_range, dim3(nb), dim3(256), 0, s, lbl, V, (long long*)ws);
    hipLaunchKernelGGL(k_label_range_final, dim3(1), dim3(256), 0, s, (const long long*)ws, nb, out_minmax);
    TEM_CHECK_LAUNCH("tem_label_range");
    return TEM_OK;
}

// ------------------------------------------------------------------------------------------------
// S1 / S2: per-label segment accumulation.  MODE 0: sums of e and counts.  MODE 1: pull term, value partial
// sum_v h^2/count_l and S_l = sum_{v in l} h * (e - mu_l)/|e - mu_l| (h = (|e-mu_l| - delta_var)+).
// ------------------------------------------------------------------------------------------------
template <int EM, int MODE>
__global__ __launch_bounds__(256) void k_seg_accum(const float* __restrict__ emb, int64_t cs,
                                                    const int64_t* __restrict__ lbl, int64_t V, int E, int C,
                                                    const float* __restrict__ means, const float* __restrict__ counts,
                                                    float delta_var, u64* __restrict__ gacc, int use_lds,
                                                    float* __restrict__ vpart) {
    extern __shared__ u64 lacc[];
    const int E1 = E + 1;
    u64* acc = use_lds ? lacc : gacc;
    if (use_lds) {
        for (int i = threadIdx.x; i < C * E1; i += blockDim.x) lacc[i] = 0;
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    float vsum = 0.f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t vend = (V + 63) / 64 * 64;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < vend; v += stride) {
        const bool valid = v < V;
        int l = -1;
        float val[EM];
#pragma unroll
        for (int e = 0; e < EM; ++e) val[e] = 0.f;
        if (valid) {
            l = (int)lbl[v];
#pragma unroll
            for (int e = 0; e < EM; ++e)
                if (e < E) val[e] = emb[(int64_t)e * cs + v];
            if (MODE == 1) {
                float d2 = 0.f;
#pragma unroll
                for (int e = 0; e < EM; ++e)
                    if (e < E) {
                        val[e] -= means[(int64_t)l * E + e];
                        d2 = fmaf(val[e], val[e], d2);
                    }
                const float d = sqrtf(d2);
                const float h = fmaxf(d - delta_var, 0.f);
                vsum += h * h / counts[l];
                const float sc = h > 0.f ? h / d : 0.f;
#pragma unroll
                for (int e = 0; e < EM; ++e) val[e] *= sc;
            }
        }
        const int l0 = __builtin_amdgcn_readfirstlane(l);
        if (__all(l == l0)) {
            if (l0 >= 0) {  // whole wave valid, one label: butterfly then a single add per channel
#pragma unroll
                for (int e = 0; e < EM; ++e)
                    if (e < E) {
                        const float sres = tem_wave_sum(val[e]);
                        if (lane == 0) fix_add(&acc[(int64_t)l0 * E1 + e], sres);
                    }
                if (MODE == 0 && lane == 0) atomicAdd(&acc[(int64_t)l0 * E1 + E], (u64)64);
            }
        } else if (valid) {
#pragma unroll
            for (int e = 0; e < EM; ++e)
                if (e < E) fix_add(&acc[(int64_t)l * E1 + e], val[e]);
            if (MODE == 0) atomicAdd(&acc[(int64_t)l * E1 + E], (u64)1);
        }
    }
    if (use_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < C * E1; i += blockDim.x)
            if (lacc[i]) atomicAdd(&gacc[i], lacc[i]);
    }
    if (MODE == 1) {
        __shared__ float sh[8], o1[1];
        float a[1] = {vsum};
        block_sum<1>(a, sh, o1);
        if (threadIdx.x == 0) vpart[blockIdx.x] = o1[0];
    }
}

__global__ void k_means_finalize(const u64* __restrict__ acc, int C, int E, float* __restrict__ means,
                                 float* __restrict__ counts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * E) return;
    const int c = i / E, e = i % E;
    const double cnt = (double)acc[(int64_t)c * (E + 1) + E];
    means[i] = (float)((double)(long long)acc[(int64_t)c * (E + 1) + e] * (1.0 / 268435456.0) / (cnt < 1.0 ? 1.0 : cnt));
    if (e == 0) counts[c] = (float)cnt;
}

__global__ __launch_bounds__(256) void k_sum_partials(const float* __restrict__ part, int n, double scale,
                                                       float* __restrict__ out) {
    __shared__ double sh[4];
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) v += (double)part[i];
    const double s = block_sum_d(v, sh);
    if (threadIdx.x == 0) out[0] = (float)(s * scale);
}

#define SP_DISPATCH_E(E, CALL)                       \
    do {                                             \
        if ((E) <= 8) { CALL(8); }                   \
        else if ((E) <= 16) { CALL(16); }            \
        else { CALL(32); }                           \
    } while (0)

static inline char* ws_take(char*& p, int64_t bytes) {
    char* r = p;
    p += (bytes + 255) / 256 * 256;
    return r;
}

extern "C" int tem_spoco_cluster_means(const float* emb, int64_t cs, const int64_t* lbl, int64_t V, int E, int C,
                                       float* means, float* counts, void* ws, int64_t ws_bytes, tem_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    TEM_REQUIRE(emb && lbl && means && counts && ws, "tem_spoco_cluster_means: null pointer");
    TEM_REQUIRE(V > 0 && C > 0 && E > 0 && E <= 32 && cs >= V, "tem_spoco_cluster_means: bad sizes (E<=32, cs>=V)");
    const int64_t accb = (int64_t)C * (E + 1) * 8;
    TEM_REQUIRE(ws_bytes >= accb, "tem_spoco_cluster_means: workspace too small");
    u64* acc = (u64*)ws;
    (void)hipMemsetAsync(acc, 0, accb, s);
    const int use_lds = accb <= 48 * 1024;
    const int nb = tem_grid_1d(V, 256, SP_MAXB);
#define CALL(EMV)                                                                                                \
    hipLaunchKernelGGL((k_seg_accum<EMV, 0>), dim3(nb), dim3(256), use_lds ? accb : 0, s, emb, cs, lbl, V, E, C, \
                       (const float*)nullptr, (const float*)nullptr, 0.f, acc, use_lds, (float*)nullptr)
    SP_DISPATCH_E(E, CALL);
#undef CALL
    hipLaunchKernelGGL(k_means_finalize, dim3((unsigned)tem_cdiv((int64_t)C * E, 256)), dim3(256), 0, s, acc, C, E, means,
                       counts);
    TEM_CHECK_LAUNCH("tem_spoco_cluster_means");
    return TEM_OK;
}

__global__ void k_fix_to_float(const u64* __restrict__ acc, int C, int E, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * E) return;
    out[i] = fix_get(acc[(int64_t)(i / E) * (E + 1) + i % E]);
}

// value_out[0] = sum_v h_v^2 / count_{l(v)} (NOT divided by the instance count); S_out [C,E].
extern "C" int tem_spoco_pull(const float* emb, int64_t cs, const int64_t* lbl, int64_t V, int E, int C,
                              const float* means, const float* counts, float delta_var, float* value_out, float* S_out,
                              void* ws, int64_t ws_bytes, tem_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    TEM_REQUIRE(emb && lbl && means && counts && value_out && S_out && ws, "tem_spoco_pull: null pointer");
    TEM_REQUIRE(V > 0 && C > 0 && E > 0 && E <= 32 && cs >= V, "tem_spoco_pull: bad sizes");
    const int64_t accb = (int64_t)C * (E + 1) * 8;
    char* p = (char*)ws;
    u64* acc = (u64*)ws_take(p, accb);
    float* vpart = (float*)ws_take(p, SP_MAXB * 4);
    TEM_REQUIRE(p - (char*)ws <= ws_bytes, "tem_spoco_pull: workspace too small");
    (void)hipMemsetAsync(acc, 0, accb, s);
    const int use_lds = accb <= 48 * 1024;
    const int nb = tem_grid_1d(V, 256, SP_MAXB);
#define CALL(EMV)                                                                                                \
    hipLaunchKernelGGL((k_seg_accum<EMV, 1>), dim3(nb), dim3(256), use_lds ? accb : 0, s, emb, cs, lbl, V, E, C, \
                       means, counts, delta_var, acc, use_lds, vpart)
    SP_DISPATCH_E(E, CALL);
#undef CALL
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, s, vpart, nb, 1.0, value_out);
    hipLaunchKernelGGL(k_fix_to_float, dim3((unsigned)tem_cdiv((int64_t)C * E, 256)), dim3(256), 0, s, acc, C, E, S_out);
    TEM_CHECK_LAUNCH("tem_spoco_pull");
    return TEM_OK;
}

// ------------------------------------------------------------------------------------------------
// S3 + S4 (means only): distance term, regulariser and their gradients wrt the means.  One block per row i.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_means_terms(const float* __restrict__ mu, int C, int E, float delta_dist,
                                                     int ignore_zero, double* __restrict__ rowval /*[C][2]*/,
                                                     float* __restrict__ ddist, float* __restrict__ dreg) {
    const int i = blockIdx.x, lane = threadIdx.x;
    const int Cn = ignore_zero ? C - 1 : C;
    const bool active = C > 1 && !(ignore_zero && C == 2);
    const float Z = (float)((double)Cn * (Cn - 1));
    float val = 0.f;
    float g[32];
#pragma unroll
    for (int e = 0; e < 32; ++e) g[e] = 0.f;
    if (active) {
        for (int j = lane; j < C; j += 64) {
            if (j == i) continue;
            float d2 = 0.f;
            for (int e = 0; e < E; ++e) {
                const float df = mu[i * E + e] - mu[j * E + e];
                d2 = fmaf(df, df, d2);
            }
            const float d = sqrtf(d2);
            if (ignore_zero && (i == 0 || j == 0)) {
                // reference scales these distances past the hinge (contrastive_impl.py:57-67): zero unless d == 0
                if (d == 0.f) val += 4.f * delta_dist * delta_dist;
                continue;
            }
            const float h = fmaxf(2.f * delta_dist - d, 0.f);
            val += h * h;
            if (h > 0.f && d > 0.f) {
                const float cf = -4.f * h / (d * Z);  // (i,j) and (j,i) both contain mu_i
#pragma unroll
                for (int e = 0; e < 32; ++e)
                    if (e < E) g[e] = fmaf(cf, mu[i * E + e] - mu[j * E + e], g[e]);
            }
        }
    }
    val = tem_wave_sum(val);
    float n2 = 0.f;
    for (int e = 0; e < E; ++e) n2 = fmaf(mu[i * E + e], mu[i * E + e], n2);
    const float nrm = sqrtf(n2);
#pragma unroll
    for (int e = 0; e < 32; ++e)
        if (e < E) {
            const float s = tem_wave_sum(g[e]);
            if (lane == 0) {
                ddist[i * E + e] = s;
                dreg[i * E + e] = nrm > 0.f ? mu[i * E + e] / (nrm * C) : 0.f;
            }
        }
    if (lane == 0) {
        rowval[i * 2] = active ? (double)val / (double)Z : 0.0;
        rowval[i * 2 + 1] = (double)nrm / C;
    }
}
__global__ void k_rows_final(const double* __restrict__ rowval, int C, float* __restrict__ out2) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int i = 0; i < C; ++i) {
            a += rowval[i * 2];
            b += rowval[i * 2 + 1];
        }
        out2[0] = (float)a;
        out2[1] = (float)b;
    }
}

extern "C" int tem_spoco_means_terms(const float* means, int C, int E, float delta_dist, int ignore_zero,
                                     float* values_out2, float* ddist, float* dreg, void* ws, int64_t ws_bytes,
                                     tem_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    TEM_REQUIRE(means && values_out2 && ddist && dreg && ws, "tem_spoco_means_terms: null pointer");
    TEM_REQUIRE(C > 0 && E > 0 && E <= 32, "tem_spoco_means_terms: bad sizes");
    TEM_REQUIRE(ws_bytes >= (int64_t)C * 16, "tem_spoco_means_terms: workspace too small");
    hipLaunchKernelGGL(k_means_terms, dim3(C), dim3(64), 0, s, means, C, E, delta_dist, ignore_zero, (double*)ws, ddist,
                       dreg);
    hipLaunchKernelGGL(k_rows_final, dim3(1), dim3(64), 0, s, (const double*)ws, C, values_out2);
    TEM_CHECK_LAUNCH("tem_spoco_means_terms");
    return TEM_OK;
}

// ------------------------------------------------------------------------------------------------
// S5: instance Dice term (value only -- the reference detaches it, spoco_loss.py:422).  The volume is [nz][slice];
// DiceLoss() sees the pmap as [1, nz, ...] so the first spatial axis plays "channel" (per-slice Dice, summed).
// grid (SP_NCH, nz, ceil((C-1)/SP_IT)); block 256.
// ------------------------------------------------------------------------------------------------
template <int EM>
__global__ __launch_bounds__(256) void k_instance_sums(const float* __restrict__ emb, int64_t cs,
                                                        const int64_t* __restrict__ lbl, int64_t slice, int E, int C,
                                                        const float* __restrict__ means, float inv_two_sigma,
                                                        float* __restrict__ part /*[nz][C][NCH][3]*/) {
    __shared__ float mu[SP_IT][EM];
    __shared__ float sh[4 * SP_IT * 3], outv[SP_IT * 3];
    const int z = blockIdx.y, ch = blockIdx.x, i0 = 1 + blockIdx.z * SP_IT;
    for (int t = threadIdx.x; t < SP_IT * EM; t += blockDim.x) {
        const int ii = i0 + t / EM, e = t % EM;
        mu[t / EM][e] = (ii < C && e < E) ? means[(int64_t)ii * E + e] : 0.f;
    }
    __syncthreads();
    float acc[SP_IT * 3];
#pragma unroll
    for (int t = 0; t < SP_IT * 3; ++t) acc[t] = 0.f;
    const int64_t per = (slice + SP_NCH - 1) / SP_NCH;
    const int64_t v0 = (int64_t)z * slice + ch * per;
    const int64_t zend = (int64_t)(z + 1) * slice;
    const int64_t v1 = v0 + per < zend ? v0 + per : zend;
    for (int64_t v = v0 + threadIdx.x; v < v1; v += blockDim.x) {
        float ev[EM];
#pragma unroll
        for (int e = 0; e < EM; ++e) ev[e] = emb[(int64_t)(e < E ? e : 0) * cs + v];   // unconditional loads (a predicated
#pragma unroll                                                                          // one is a branch + vmcnt(0) each)
        for (int e = 0; e < EM; ++e) ev[e] = e < E ? ev[e] : 0.f;
        const int l = (int)lbl[v];
#pragma unroll
        for (int t = 0; t < SP_IT; ++t) {
            float d2 = 0.f;
#pragma unroll
            for (int e = 0; e < EM; ++e) {
                const float df = ev[e] - mu[t][e];
                d2 = fmaf(df, df, d2);
            }
            // reference: d = norm(...); exp(-d*d/two_sigma)
            const float d = sqrtf(d2);
            const float p = expf(-(d * d) * inv_two_sigma);
            const bool m = l == i0 + t;
            acc[t * 3 + 0] += m ? p : 0.f;
            acc[t * 3 + 1] = fmaf(p, p, acc[t * 3 + 1]);
            acc[t * 3 + 2] += m ? 1.f : 0.f;
        }
    }
    block_sum<SP_IT * 3>(acc, sh, outv);
    if (threadIdx.x < SP_IT * 3) {
        const int t = threadIdx.x / 3, k = threadIdx.x % 3, ii = i0 + t;
        if (ii < C) part[(((int64_t)z * C + ii) * SP_NCH + ch) * 3 + k] = outv[threadIdx.x];
    }
}

__global__ __launch_bounds__(256) void k_instance_rows(const float* __restrict__ part, int nz, int C, double eps,
                                                        double* __restrict__ per_inst) {
    __shared__ double sh[4];
    const int i = blockIdx.x + 1;
    double loss = 0.0;
    for (int z = threadIdx.x; z < nz; z += blockDim.x) {
        double num = 0.0, pp = 0.0, mm = 0.0;
        for (int ch = 0; ch < SP_NCH; ++ch) {
            const float* q = part + (((int64_t)z * C + i) * SP_NCH + ch) * 3;
            num += q[0];
            pp += q[1];
            mm += q[2];
        }
        const double den = pp + mm;
        loss += 1.0 - 2.0 * (num / (den < eps ? eps : den));
    }
    const double s = block_sum_d(loss, sh);
    if (threadIdx.x == 0) per_inst[i] = s;
}
__global__ __launch_bounds__(256) void k_instance_final(const double* __restrict__ per_inst, int C,
                                                         float* __restrict__ out) {
    __shared__ double sh[4];
    double v = 0.0;
    for (int i = 1 + threadIdx.x; i < C; i += blockDim.x) v += per_inst[i];
    const double s = block_sum_d(v, sh);
    if (threadIdx.x == 0) out[0] = C > 1 ? (float)(s / (C - 1)) : 0.f;
}

extern "C" int tem_spoco_instance_dice(const float* emb, int64_t cs, const int64_t* lbl, int64_t V, int nz, int E, int C,
                                       const float* means, float two_sigma, float eps, float* value_out, void* ws,
                                       int64_t ws_bytes, tem_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    TEM_REQUIRE(emb && lbl && means && value_out && ws, "tem_spoco_instance_dice: null pointer");
    TEM_REQUIRE(V > 0 && nz > 0 && V % nz == 0 && C > 0 && E > 0 && E <= 32 && cs >= V,
                "tem_spoco_instance_dice: bad sizes");
    char* p = (char*)ws;
    float* part = (float*)ws_take(p, (int64_t)nz * C * SP_NCH * 3 * 4);
    double* per_inst = (double*)ws_take(p, (int64_t)C * 8);
    TEM_REQUIRE(p - (char*)ws <= ws_bytes, "tem_spoco_instance_dice: workspace too small");
    if (C > 1) {
        const int ntile = (C - 1 + SP_IT - 1) / SP_IT;
        TEM_REQUIRE(nz <= 65535 && ntile <= 65535, "tem_spoco_instance_dice: grid too large");
        // chunks past the end of a short slice write zeros (v1 <= v0), so every partial slot is defined
#define CALL(EMV)                                                                                                  \
    hipLaunchKernelGGL((k_instance_sums<EMV>), dim3(SP_NCH, nz, ntile), dim3(256), 0, s, emb, cs, lbl, V / nz, E, C, \
                       means, 1.f / two_sigma, part)
        SP_DISPATCH_E(E, CALL);
#undef CALL
    }
    if (C > 1) hipLaunchKernelGGL(k_instance_rows, dim3(C - 1), dim3(256), 0, s, part, nz, C, (double)eps, per_inst);
    hipLaunchKernelGGL(k_instance_final, dim3(1), dim3(256), 0, s, (const double*)per_inst, C, value_out);
    TEM_CHECK_LAUNCH("tem_spoco_instance_dice");
    return TEM_OK;
}

// ------------------------------------------------------------------------------------------------
// S4b: unlabeled (background) push, spoco_loss.py:162-190: value, d/d(emb) on background voxels, d/d(means).
//   push = 1/((C-1) n_bg) * sum_{i>=1} sum_{v in bg} (delta_dist - |e_v - mu_i|)+^2
// ------------------------------------------------------------------------------------------------
template <int EM>
__global__ __launch_bounds__(256) void k_push(const float* __restrict__ emb, int64_t cs, const int64_t* __restrict__ lbl,
                                               int64_t V, int E, int C, const float* __restrict__ means,
                                               const float* __restrict__ counts, float delta_dist, float grad_scale,
                                               float* __restrict__ grad, int64_t gcs, u64* __restrict__ gacc,
                                               float* __restrict__ vpart) {
    const int lane = threadIdx.x & 63;
    const int E1 = E + 1;
    const float norm = 1.f / (counts[0] * (float)(C - 1));
    float vsum = 0.f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t vend = (V + 63) / 64 * 64;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < vend; v += stride) {
        const bool bg = v < V && lbl[v] == 0;
        if (!__any(bg)) continue;
        float ev[EM], ge[EM];
        const int64_t vc = v < V ? v : V - 1;
#pragma unroll
        for (int e = 0; e < EM; ++e) ev[e] = emb[(int64_t)(e < E ? e : 0) * cs + vc];   // unconditional, dropped below
#pragma unroll
        for (int e = 0; e < EM; ++e) {
            ev[e] = (bg && e < E) ? ev[e] : 0.f;
            ge[e] = 0.f;
        }
        for (int i = 1; i < C; ++i) {
            float df[EM], d2 = 0.f;
#pragma unroll
            for (int e = 0; e < EM; ++e) {
                df[e] = e < E ? ev[e] - means[(int64_t)i * E + e] : 0.f;
                d2 = fmaf(df[e], df[e], d2);
            }
            const float d = sqrtf(d2);
            const float t = bg ? fmaxf(delta_dist - d, 0.f) : 0.f;
            if (!__any(t > 0.f)) continue;
            vsum = fmaf(t, t, vsum);
            const float sc = (t > 0.f && d > 0.f) ? t / d : 0.f;
#pragma unroll
            for (int e = 0; e < EM; ++e)
                if (e < E) {
                    const float u = sc * df[e];      // t * unit vector (bounded by delta_dist)
                    ge[e] -= u;                      // d/de_v of t^2 = -2 t u_hat
                    const float su = tem_wave_sum(u);
                    if (lane == 0 && su != 0.f) fix_add(&gacc[(int64_t)i * E1 + e], su);
                }
        }
        if (bg && grad) {
#pragma unroll
            for (int e = 0; e < EM; ++e)
                if (e < E) grad[(int64_t)e * gcs + v] += grad_scale * 2.f * norm * ge[e];
        }
    }
    __shared__ float sh[8], o1[1];
    float a[1] = {vsum};
    block_sum<1>(a, sh, o1);
    if (threadIdx.x == 0) vpart[blockIdx.x] = o1[0];
}
__global__ void k_push_final(const float* __restrict__ vpart, int nb, const u64* __restrict__ acc, int C, int E,
                             const float* __restrict__ counts, float* __restrict__ value_out, float* __restrict__ dpush) {
    const double norm = 1.0 / ((double)counts[0] * (C - 1));
    for (int i = threadIdx.x; i < C * E; i += blockDim.x)
        dpush[i] = (float)(2.0 * norm * (double)fix_get(acc[(int64_t)(i / E) * (E + 1) + i % E]));
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int b = 0; b < nb; ++b) s += vpart[b];
        value_out[0] = (float)(s * norm);
    }
}

extern "C" int tem_spoco_push(const float* emb, int64_t cs, const int64_t* lbl, int64_t V, int E, int C,
                              const float* means, const float* counts, float delta_dist, float* value_out,
                              float grad_scale, float* grad, int64_t gcs, float* dpush, void* ws, int64_t ws_bytes,
                              tem_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    TEM_REQUIRE(emb && lbl && means && counts && value_out && dpush && ws, "tem_spoco_push: null pointer");
    TEM_REQUIRE(V > 0 && C > 1 && E > 0 && E <= 32 && cs >= V, "tem_spoco_push: bad sizes (needs C > 1)");
    const int64_t accb = (int64_t)C * (E + 1) * 8;
    char* p = (char*)ws;
    u64* acc = (u64*)ws_take(p, accb);
    float* vpart = (float*)ws_take(p, SP_MAXB * 4);
    TEM_REQUIRE(p - (char*)ws <= ws_bytes, "tem_spoco_push: workspace too small");
    (void)hipMemsetAsync(acc, 0, accb, s);
    const int nb = tem_grid_1d(V, 256, SP_MAXB);
#define CALL(EMV)                                                                                               \
    hipLaunchKernelGGL((k_push<EMV>), dim3(nb), dim3(256), 0, s, emb, cs, lbl, V, E, C, means, counts, delta_dist, \
                       grad_scale, grad, gcs, acc, vpart)
    SP_DISPATCH_E(E, CALL);
#undef CALL
    hipLaunchKernelGGL(k_push_final, dim3(1), dim3(256), 0, s, vpart, nb, acc, C, E, counts, value_out, dpush);
    TEM_CHECK_LAUNCH("tem_spoco_push");
    return TEM_OK;
}

// ------------------------------------------------------------------------------------------------
// gradient of  w_var * var + <dmu, means>  wrt the embeddings (means depend on every voxel of their label):
//   grad_e[v] (+)= w_var * 2 h_v / (n_inst count_l) * u_hat_v + dmu_tot[l] / count_l
//   dmu_tot[l]  = -w_var * 2 / (n_inst count_l) * S_l + w_dist * ddist_l + w_reg * dreg_l + w_push * dpush_l
// ------------------------------------------------------------------------------------------------
__global__ void k_combine_dmeans(const float* __restrict__ S, const float* __restrict__ ddist,
                                 const float* __restrict__ dreg, const float* __restrict__ dpush,
                                 const float* __restrict__ counts, int C, int E, float w_var, float n_inst, float w_dist,
                                 float w_reg, float w_push, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * E) return;
    const float cnt = counts[i / E];
    float g = -w_var * 2.f / (n_inst * cnt) * S[i] + w_dist * ddist[i] + w_reg * dreg[i];
    if (dpush) g = fmaf(w_push, dpush[i], g);
    out[i] = g / cnt;  // pre-divided by the label size (d mean / d e_v = 1/count)
}
template <int EM>
__global__ __launch_bounds__(256) void k_embed_grad(const float* __restrict__ emb, int64_t cs,
                                                     const int64_t* __restrict__ lbl, int64_t V, int E,
                                                     const float* __restrict__ means, const float* __restrict__ counts,
                                                     const float* __restrict__ dmu, float delta_var, float cvar,
                                                     float* __restrict__ grad, int64_t gcs, int accumulate) {
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (int64_t)gridDim.x * blockDim.x) {
        const int l = (int)lbl[v];
        float df[EM], d2 = 0.f;
#pragma unroll
        for (int e = 0; e < EM; ++e) {
            df[e] = e < E ? emb[(int64_t)e * cs + v] - means[(int64_t)l * E + e] : 0.f;
            d2 = fmaf(df[e], df[e], d2);
        }
        const float d = sqrtf(d2);
        const float h = fmaxf(d - delta_var, 0.f);
        const float sc = h > 0.f ? cvar * 2.f * h / (counts[l] * d) : 0.f;
#pragma unroll
        for (int e = 0; e < EM; ++e)
            if (e < E) {
                const float g = fmaf(sc, df[e], dmu[(int64_t)l * E + e]);
                float* o = grad + (int64_t)e * gcs + v;
                *o = accumulate ? *o + g : g;
            }
    }
}

extern "C" int tem_spoco_embed_grad(const float* emb, int64_t cs, const int64_t* lbl, int64_t V, int E, int C,
                                    const float* means, const float* counts, const float* S, const float* ddist,
                                    const float* dreg, const float* dpush, float delta_var, float n_inst, float w_var,
                                    float w_dist, float w_reg, float w_push, float* grad, int64_t gcs, int accumulate,
                                    void* ws, int64_t ws_bytes, tem_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    TEM_REQUIRE(emb && lbl && means && counts && S && ddist && dreg && grad && ws, "tem_spoco_embed_grad: null pointer");
    TEM_REQUIRE(V > 0 && C > 0 && E > 0 && E <= 32 && cs >= V && gcs >= V && n_inst > 0.f,
                "tem_spoco_embed_grad: bad sizes");
    TEM_REQUIRE(ws_bytes >= (int64_t)C * E * 4, "tem_spoco_embed_grad: workspace too small");
    float* dmu = (float*)ws;
    hipLaunchKernelGGL(k_combine_dmeans, dim3((unsigned)tem_cdiv((int64_t)C * E, 256)), dim3(256), 0, s, S, ddist, dreg,
                       dpush, counts, C, E, w_var, n_inst, w_dist, w_reg, w_push, dmu);
    const int nb = tem_grid_1d(V, 256);
#define CALL(EMV)                                                                                                 \
    hipLaunchKernelGGL((k_embed_grad<EMV>), dim3(nb), dim3(256), 0, s, emb, cs, lbl, V, E, means, counts, dmu,     \
                       delta_var, w_var / n_inst, grad, gcs, accumulate)
    SP_DISPATCH_E(E, CALL);
#undef CALL
    TEM_CHECK_LAUNCH("tem_spoco_embed_grad");
    return TEM_OK;
}

// ------------------------------------------------------------------------------------------------
// S6 helpers: count the unlabeled (label == 0) voxels and find the k-th one in row-major order
// (the reference builds torch.nonzero(mask) -- three index tensors of the mask size -- per anchor, :513).
// ------------------------------------------------------------------------------------------------
#define ZCH 1024
__global__ __launch_bounds__(256) void k_zero_counts(const int64_t* __restrict__ lbl, int64_t V,
                                                      int* __restrict__ chunk_counts) {
    const int64_t base = (int64_t)blockIdx.x * ZCH;
    int c = 0;
#pragma unroll
    for (int k = 0; k < ZCH / 256; ++k) {
        const int64_t v = base + k * 256 + threadIdx.x;
        c += (v < V && lbl[v] == 0) ? 1 : 0;
    }
    __shared__ float sh[8], o1[1];
    float a[1] = {(float)c};
    block_sum<1>(a, sh, o1);  // <= 1024: exact in fp32
    if (threadIdx.x == 0) chunk_counts[blockIdx.x] = (int)o1[0];
}
__global__ __launch_bounds__(256) void k_zero_total(const int* __restrict__ chunk_counts, int64_t nchunk,
                                                     int64_t* __restrict__ total) {
    __shared__ double sh[4];
    double v = 0.0;  // integers < 2^53: exact
    for (int64_t i = threadIdx.x; i < nchunk; i += blockDim.x) v += (double)chunk_counts[i];
    const double s = block_sum_d(v, sh);
    if (threadIdx.x == 0) total[0] = (int64_t)s;
}
__global__ __launch_bounds__(64) void k_kth_zero(const int64_t* __restrict__ lbl, int64_t V,
                                                  const int* __restrict__ chunk_counts, int64_t nchunk,
                                                  const int64_t* __restrict__ ranks, int64_t* __restrict__ idx_out) {
    const int lane = threadIdx.x;
    long long k = ranks[blockIdx.x];
    long long running = 0;
    int64_t chunk = -1;
    for (int64_t base = 0; base < nchunk && chunk < 0; base += 64) {
        const int c = (base + lane < nchunk) ? chunk_counts[base + lane] : 0;
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        const int tot = __shfl(incl, 63, 64);
        if (running + tot > k) {
            const unsigned long long b = __ballot(running + incl > k);
            const int fl = __ffsll((long long)b) - 1;
            chunk = base + fl;
            const int incl_f = __shfl(incl, fl, 64), c_f = __shfl(c, fl, 64);
            k -= running + incl_f - c_f;
        } else {
            running += tot;
        }
    }
    int64_t res = -1;
    if (chunk >= 0) {
        for (int it = 0; it < ZCH / 64 && res < 0; ++it) {
            const int64_t v = chunk * ZCH + it * 64 + lane;
            const bool z = v < V && lbl[v] == 0;
            const unsigned long long b = __ballot(z);
            const int cnt = __popcll(b);
            if (k < cnt) {
                const int before = __popcll(b & ((1ull << lane) - 1ull));
                const unsigned long long hit = __ballot(z && before == (int)k);
                res = chunk * ZCH + it * 64 + (__ffsll((long long)hit) - 1);
            } else {
                k -= cnt;
            }
        }
    }
    if (lane == 0) idx_out[blockIdx.x] = res;
}

extern "C" int tem_zero_count(const int64_t* lbl, int64_t V, int* chunk_counts, int64_t* total_out, tem_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    TEM_REQUIRE(lbl && chunk_counts && total_out && V > 0, "tem_zero_count: null pointer or empty input");
    const int64_t nchunk = tem_cdiv(V, ZCH);
    hipLaunchKernelGGL(k_zero_counts, dim3((unsigned)nchunk), dim3(256), 0, s, lbl, V, chunk_counts);
    hipLaunchKernelGGL(k_zero_total, dim3(1), dim3(256), 0, s, chunk_counts, nchunk, total_out);
    TEM_CHECK_LAUNCH("tem_zero_count");
    return TEM_OK;
}
extern "C" int tem_zero_select(const int64_t* lbl, int64_t V, const int* chunk_counts, const int64_t* ranks, int A,
                               int64_t* idx_out, tem_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    TEM_REQUIRE(lbl && chunk_counts && ranks && idx_out && V > 0 && A > 0, "tem_zero_select: null pointer or empty input");
    hipLaunchKernelGGL(k_kth_zero, dim3(A), dim3(64), 0, s, lbl, V, chunk_counts, tem_cdiv(V, ZCH), ranks, idx_out);
    TEM_CHECK_LAUNCH("tem_zero_select");
    return TEM_OK;
}

// ------------------------------------------------------------------------------------------------
// S6: embedding consistency (spoco_loss.py:503-527): A anchors, q/k Gaussian pmaps, per-slice Dice over
// [A, nz, slice] (the first spatial axis is DiceLoss()'s channel axis), gradient wrt emb_q incl. the anchor voxels.
// ------------------------------------------------------------------------------------------------
__global__ void k_gather_anchors(const float* __restrict__ eq, const float* __restrict__ ek, int64_t cs,
                                 const int64_t* __restrict__ idx, int A, int E, float* __restrict__ anc /*[A][2][E]*/) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A * E) return;
    const int a = i / E, e = i % E;
    anc[(a * 2 + 0) * E + e] = eq[(int64_t)e * cs + idx[a]];
    anc[(a * 2 + 1) * E + e] = ek[(int64_t)e * cs + idx[a]];
}

template <int EM, bool GRAD>
__global__ __launch_bounds__(256) void k_consistency(const float* __restrict__ eq, const float* __restrict__ ek,
                                                      int64_t cs, int64_t slice, int E, int A,
                                                      const float* __restrict__ anc, float inv_two_sigma,
                                                      float* __restrict__ part /*[nz][NCH][3]*/,
                                                      const double* __restrict__ zs /*[nz][2]*/, double eps,
                                                      float grad_scale, float* __restrict__ grad, int64_t gcs,
                                                      float* __restrict__ apart /*[nz*NCH][A][E]*/) {
    extern __shared__ float smem[];
    float* an = smem;                      // [A][2][EM]
    float* wacc = smem + A * 2 * EM;       // GRAD: [4 waves][A][EM]
    __shared__ float sh[4 * 3], o3[3];
    const int z = blockIdx.y, ch = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int t = threadIdx.x; t < A * 2 * EM; t += blockDim.x) {
        const int e = t % EM, a2 = t / EM;
        an[t] = e < E ? anc[a2 * E + e] : 0.f;
    }
    if (GRAD)
        for (int t = threadIdx.x; t < 4 * A * EM; t += blockDim.x) wacc[t] = 0.f;
    __syncthreads();
    float c_k = 0.f, c_q = 0.f;  // dL/dq = c_k * k + c_q * q
    if (GRAD) {
        const double n = zs[z * 2], d = zs[z * 2 + 1];
        if (d > eps) {
            c_k = (float)(-2.0 / d);
            c_q = (float)(4.0 * n / (d * d));
        } else {
            c_k = (float)(-2.0 / eps);
        }
    }
    float acc[3] = {0.f, 0.f, 0.f};
    const int64_t per = (slice + SP_NCH - 1) / SP_NCH;
    const int64_t v0 = (int64_t)z * slice + ch * per;
    const int64_t zend = (int64_t)(z + 1) * slice;
    const int64_t v1 = v0 + per < zend ? v0 + per : zend;
    const int64_t vend = v0 + (v1 > v0 ? (v1 - v0 + 63) / 64 * 64 : 0);
    for (int64_t v = v0 + threadIdx.x; v < vend; v += blockDim.x) {
        const bool valid = v < v1;
        float q[EM], k[EM], ge[EM];
        const int64_t vc = valid ? v : v1 - 1;
#pragma unroll
        for (int e = 0; e < EM; ++e) {   // unconditional loads, dropped below
            q[e] = eq[(int64_t)(e < E ? e : 0) * cs + vc];
            k[e] = ek[(int64_t)(e < E ? e : 0) * cs + vc];
        }
#pragma unroll
        for (int e = 0; e < EM; ++e) {
            q[e] = (valid && e < E) ? q[e] : 0.f;
            k[e] = (valid && e < E) ? k[e] : 0.f;
            ge[e] = 0.f;
        }
        for (int a = 0; a < A; ++a) {
            float dq[EM], d2q = 0.f, d2k = 0.f;
#pragma unroll
            for (int e = 0; e < EM; ++e) {
                dq[e] = q[e] - an[(a * 2) * EM + e];
                const float dk = k[e] - an[(a * 2 + 1) * EM + e];
                d2q = fmaf(dq[e], dq[e], d2q);
                d2k = fmaf(dk, dk, d2k);
            }
            const float nq = sqrtf(d2q), nk = sqrtf(d2k);
            const float pq = valid ? expf(-(nq * nq) * inv_two_sigma) : 0.f;
            const float pk = valid ? expf(-(nk * nk) * inv_two_sigma) : 0.f;
            if (!GRAD) {
                acc[0] = fmaf(pq, pk, acc[0]);
                acc[1] = fmaf(pq, pq, acc[1]);
                acc[2] = fmaf(pk, pk, acc[2]);
            } else {
                // dq_pmap/de = pq * (-2/two_sigma) * (e - anchor); the anchor voxel receives the opposite
                const float cf = grad_scale * (c_k * pk + c_q * pq) * pq * (-2.f * inv_two_sigma);
                float gv[EM];
#pragma unroll
                for (int e = 0; e < EM; ++e) {
                    gv[e] = cf * dq[e];
                    ge[e] += gv[e];
                }
#pragma unroll
                for (int e0 = 0; e0 < EM; e0 += 8)
                    if (e0 < E) {
                        const float tot = wave_reduce8(gv + e0, lane);
                        if (lane < 8) wacc[(wv * A + a) * EM + e0 + reduce8_index(lane)] -= tot;
                    }
            }
        }
        if (GRAD && valid) {
#pragma unroll
            for (int e = 0; e < EM; ++e)
                if (e < E) grad[(int64_t)e * gcs + v] += ge[e];
        }
    }
    if (!GRAD) {
        block_sum<3>(acc, sh, o3);
        if (threadIdx.x < 3) part[((int64_t)z * SP_NCH + ch) * 3 + threadIdx.x] = o3[threadIdx.x];
    } else {
        __syncthreads();
        const int64_t blk = (int64_t)z * SP_NCH + ch;
        for (int t = threadIdx.x; t < A * E; t += blockDim.x) {
            const int a = t / E, e = t % E;
            float sres = 0.f;
            for (int w = 0; w < 4; ++w) sres += wacc[(w * A + a) * EM + e];
            apart[(blk * A + a) * E + e] = sres;
        }
    }
}
__global__ __launch_bounds__(256) void k_consistency_final(const float* __restrict__ part, int nz, double eps,
                                                            double* __restrict__ zs, float* __restrict__ out) {
    __shared__ double sh[4];
    double loss = 0.0;
    for (int z = threadIdx.x; z < nz; z += blockDim.x) {
        double n = 0.0, qq = 0.0, kk = 0.0;
        for (int ch = 0; ch < SP_NCH; ++ch) {
            const float* p = part + ((int64_t)z * SP_NCH + ch) * 3;
            n += p[0];
            qq += p[1];
            kk += p[2];
        }
        const double d = qq + kk;
        zs[z * 2] = n;
        zs[z * 2 + 1] = d;
        loss += 1.0 - 2.0 * (n / (d < eps ? eps : d));
    }
    const double s = block_sum_d(loss, sh);
    if (threadIdx.x == 0) out[0] = (float)s;
}
// anchor gradients: block (a, e) sums the per-block partials, then ONE thread per channel adds them to the anchor
// voxels sequentially (the same voxel may have been drawn twice).
__global__ __launch_bounds__(256) void k_anchor_reduce(const float* __restrict__ apart, int64_t nblk, int A, int E,
                                                        double* __restrict__ ganc /*[A][E]*/) {
    __shared__ double sh[4];
    const int a = blockIdx.x / E, e = blockIdx.x % E;
    double v = 0.0;
    for (int64_t b = threadIdx.x; b < nblk; b += blockDim.x) v += (double)apart[(b * A + a) * E + e];
    const double s = block_sum_d(v, sh);
    if (threadIdx.x == 0) ganc[a * E + e] = s;
}
__global__ void k_anchor_scatter(const double* __restrict__ ganc, const int64_t* __restrict__ idx, int A, int E,
                                 float* __restrict__ grad, int64_t gcs) {
    const int e = threadIdx.x;
    if (e >= E) return;
    for (int a = 0; a < A; ++a) grad[(int64_t)e * gcs + idx[a]] += (float)ganc[a * E + e];
}

extern "C" int tem_spoco_consistency(const float* emb_q, const float* emb_k, int64_t cs, int64_t V, int nz, int E,
                                     const int64_t* anchor_idx, int A, float two_sigma, float eps, float* value_out,
                                     float grad_scale, float* grad_q, int64_t gcs, void* ws, int64_t ws_bytes,
                                     tem_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    TEM_REQUIRE(emb_q && emb_k && anchor_idx && value_out && ws, "tem_spoco_consistency: null pointer");
    TEM_REQUIRE(V > 0 && nz > 0 && nz <= 65535 && V % nz == 0 && E > 0 && E <= 32 && A > 0 && A <= SP_MAXA && cs >= V,
                "tem_spoco_consistency: bad sizes (E<=32, anchors<=64)");
    char* p = (char*)ws;
    float* anc = (float*)ws_take(p, (int64_t)A * 2 * E * 4);
    float* part = (float*)ws_take(p, (int64_t)nz * SP_NCH * 3 * 4);
    double* zs = (double*)ws_take(p, (int64_t)nz * 16);
    float* apart = (float*)ws_take(p, (int64_t)nz * SP_NCH * A * E * 4);
    double* ganc = (double*)ws_take(p, (int64_t)A * E * 8);
    TEM_REQUIRE(p - (char*)ws <= ws_bytes, "tem_spoco_consistency: workspace too small");
    hipLaunchKernelGGL(k_gather_anchors, dim3((unsigned)tem_cdiv((int64_t)A * E, 64)), dim3(64), 0, s, emb_q, emb_k, cs,
                       anchor_idx, A, E, anc);
    const int64_t slice = V / nz;
#define CALL(EMV)                                                                                                    \
    do {                                                                                                             \
        hipLaunchKernelGGL((k_consistency<EMV, false>), dim3(SP_NCH, nz), dim3(256), (size_t)A * 2 * EMV * 4, s, emb_q, \
                           emb_k, cs, slice, E, A, anc, 1.f / two_sigma, part, (const double*)nullptr, (double)eps, 0.f, \
                           (float*)nullptr, (int64_t)0, (float*)nullptr);                                            \
        hipLaunchKernelGGL(k_consistency_final, dim3(1), dim3(256), 0, s, part, nz, (double)eps, zs, value_out);       \
        if (grad_q) {                                                                                                \
            hipLaunchKernelGGL((k_consistency<EMV, true>), dim3(SP_NCH, nz), dim3(256),                               \
                               (size_t)(A * 2 * EMV + 4 * A * EMV) * 4, s, emb_q, emb_k, cs, slice, E, A, anc,        \
                               1.f / two_sigma, part, (const double*)zs, (double)eps, grad_scale, grad_q, gcs, apart); \
            hipLaunchKernelGGL(k_anchor_reduce, dim3(A* E), dim3(256), 0, s, apart, (int64_t)nz* SP_NCH, A, E, ganc);  \
            hipLaunchKernelGGL(k_anchor_scatter, dim3(1), dim3(64), 0, s, (const double*)ganc, anchor_idx, A, E,      \
                               grad_q, gcs);                                                                         \
        }                                                                                                            \
    } while (0)
    SP_DISPATCH_E(E, CALL);
#undef CALL
    TEM_CHECK_LAUNCH("tem_spoco_consistency");
    return TEM_OK;
}

// ------------------------------------------------------------------------------------------------
// S7: affinity side loss (affinity_side_loss.py:92-172).  Partner of voxel x under offset o is clamp(x + o)
// (replication padding => a clamped partner is the voxel itself or its border neighbour).
//   a = 1 - ((2 delta - |e_x - e_p|) / (2 delta))+^2,  t = [label_x != label_p],  loss = sum_k (1 - dice_k(a, t))
// ------------------------------------------------------------------------------------------------
struct AffOffsets {
    int o[SP_MAXK][3];
};
__device__ __forceinline__ int clampi(int v, int n) { return v < 0 ? 0 : (v > n - 1 ? n - 1 : v); }

template <int EM>
__global__ __launch_bounds__(256) void k_aff_sums(const float* __restrict__ emb, int64_t cs,
                                                   const int64_t* __restrict__ lbl, int D, int H, int W, int E,
                                                   AffOffsets offs, float delta, float* __restrict__ part /*[K][nb][3]*/) {
    const int k = blockIdx.y;
    const int oz = offs.o[k][0], oy = offs.o[k][1], ox = offs.o[k][2];
    const int64_t V = (int64_t)D * H * W;
    float acc[3] = {0.f, 0.f, 0.f};
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(v % W), y = (int)((v / W) % H), z = (int)(v / ((int64_t)W * H));
        const int64_t p = ((int64_t)clampi(z + oz, D) * H + clampi(y + oy, H)) * W + clampi(x + ox, W);
        float d2 = 0.f;
#pragma unroll
        for (int e = 0; e < EM; ++e)
            if (e < E) {
                const float df = emb[(int64_t)e * cs + v] - emb[(int64_t)e * cs + p];
                d2 = fmaf(df, df, d2);
            }
        const float sv = fmaxf((2.f * delta - sqrtf(d2)) / (2.f * delta), 0.f);
        const float a = 1.f - sv * sv;
        const float t = lbl[v] != lbl[p] ? 1.f : 0.f;
        acc[0] = fmaf(a, t, acc[0]);
        acc[1] = fmaf(a, a, acc[1]);
        acc[2] += t;
    }
    __shared__ float sh[4 * 3], o3[3];
    block_sum<3>(acc, sh, o3);
    if (threadIdx.x < 3) part[((int64_t)k * gridDim.x + blockIdx.x) * 3 + threadIdx.x] = o3[threadIdx.x];
}
__global__ __launch_bounds__(256) void k_aff_rows(const float* __restrict__ part, int nb, double eps,
                                                   double* __restrict__ ks /*[K][3]*/) {
    __shared__ double sh[4];
    const int k = blockIdx.x;
    double n = 0.0, d = 0.0;
    for (int b = threadIdx.x; b < nb; b += blockDim.x) {
        const float* p = part + ((int64_t)k * nb + b) * 3;
        n += p[0];
        d += (double)p[1] + (double)p[2];
    }
    n = block_sum_d(n, sh);
    d = block_sum_d(d, sh);
    if (threadIdx.x == 0) {
        ks[k * 3] = n;
        ks[k * 3 + 1] = d;
        ks[k * 3 + 2] = 1.0 - 2.0 * (n / (d < eps ? eps : d));
    }
}
__global__ void k_aff_final(const double* __restrict__ ks, int K, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double loss = 0.0;
        for (int k = 0; k < K; ++k) loss += ks[k * 3 + 2];
        out[0] = (float)loss;
    }
}

// pre-image of u under v -> clamp(v + o, 0, n-1) as an inclusive range [lo, hi] (empty if lo > hi)
__device__ __forceinline__ void preimage(int u, int o, int n, int& lo, int& hi) {
    const bool at_lo = u == 0, at_hi = u == n - 1;
    lo = hi = u - o;  // interior: the single candidate u - o
    if (at_lo && at_hi) {
        lo = 0;
        hi = 0;
    } else if (at_lo) {  // every v with v + o <= 0
        lo = 0;
        hi = (-o < n - 1) ? -o : n - 1;
    } else if (at_hi) {  // every v with v + o >= n - 1
        lo = (n - 1 - o > 0) ? n - 1 - o : 0;
        hi = n - 1;
    } else if (lo < 0 || lo > n - 1) {
        lo = 1;
        hi = 0;
    }
}

template <int EM>
__global__ __launch_bounds__(256) void k_aff_grad(const float* __restrict__ emb, int64_t cs,
                                                   const int64_t* __restrict__ lbl, int D, int H, int W, int E, int K,
                                                   AffOffsets offs, float delta, const double* __restrict__ ks,
                                                   double eps, float grad_scale, float* __restrict__ grad, int64_t gcs) {
    const int64_t V = (int64_t)D * H * W;
    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < V; u += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(u % W), y = (int)((u / W) % H), z = (int)(u / ((int64_t)W * H));
        float eu[EM], g[EM];
#pragma unroll
        for (int e = 0; e < EM; ++e) eu[e] = emb[(int64_t)(e < E ? e : 0) * cs + u];   // unconditional loads
#pragma unroll
        for (int e = 0; e < EM; ++e) {
            eu[e] = e < E ? eu[e] : 0.f;
            g[e] = 0.f;
        }
        const int64_t lu = lbl[u];
        for (int k = 0; k < K; ++k) {
            const int oz = offs.o[k][0], oy = offs.o[k][1], ox = offs.o[k][2];
            const double n = ks[k * 3], dn = ks[k * 3 + 1];
            float c_t, c_a;  // dL/da = c_t * t + c_a * a
            if (dn > eps) {
                c_t = (float)(-2.0 / dn);
                c_a = (float)(4.0 * n / (dn * dn));
            } else {
                c_t = (float)(-2.0 / eps);
                c_a = 0.f;
            }
            // role 1: u is the first voxel of its own pair; role 2: u is the partner of every v in its pre-image.
            // Both give the same expression in (e_u - e_other): d a/d e_u = (s/delta) * (e_u - e_other)/|.|
            int zl, zh, yl, yh, xl, xh;
            preimage(z, oz, D, zl, zh);
            preimage(y, oy, H, yl, yh);
            preimage(x, ox, W, xl, xh);
            const int64_t p1 = ((int64_t)clampi(z + oz, D) * H + clampi(y + oy, H)) * W + clampi(x + ox, W);
            for (int role = 0; role < 2; ++role) {
                const int z0 = role ? zl : 0, z1 = role ? zh : 0;
                for (int vz = z0; vz <= z1; ++vz) {
                    const int y0 = role ? yl : 0, y1 = role ? yh : 0;
                    for (int vy = y0; vy <= y1; ++vy) {
                        const int x0 = role ? xl : 0, x1 = role ? xh : 0;
                        for (int vx = x0; vx <= x1; ++vx) {
                            const int64_t o = role ? ((int64_t)vz * H + vy) * W + vx : p1;
                            float df[EM], d2 = 0.f;
#pragma unroll
                            for (int e = 0; e < EM; ++e) {
                                df[e] = e < E ? eu[e] - emb[(int64_t)e * cs + o] : 0.f;
                                d2 = fmaf(df[e], df[e], d2);
                            }
                            const float d = sqrtf(d2);
                            const float sv = fmaxf((2.f * delta - d) / (2.f * delta), 0.f);
                            if (sv > 0.f && d > 0.f) {
                                const float a = 1.f - sv * sv;
                                const float t = lu != lbl[o] ? 1.f : 0.f;
                                const float cf = grad_scale * (c_t * t + c_a * a) * sv / (delta * d);
#pragma unroll
                                for (int e = 0; e < EM; ++e) g[e] = fmaf(cf, df[e], g[e]);
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int e = 0; e < EM; ++e)
            if (e < E) grad[(int64_t)e * gcs + u] += g[e];
    }
}

extern "C" int tem_affinity_side(const float* emb, int64_t cs, const int64_t* lbl, int D, int H, int W, int E,
                                 const int* offsets_zyx, int K, float delta, float eps, float* value_out,
                                 float grad_scale, float* grad, int64_t gcs, void* ws, int64_t ws_bytes, tem_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    TEM_REQUIRE(emb && lbl && offsets_zyx && value_out && ws, "tem_affinity_side: null pointer");
    const int64_t V = (int64_t)D * H * W;
    TEM_REQUIRE(D > 0 && H > 0 && W > 0 && E > 0 && E <= 32 && K > 0 && K <= SP_MAXK && cs >= V,
                "tem_affinity_side: bad sizes (E<=32, offsets<=32)");
    AffOffsets offs;
    for (int k = 0; k < K; ++k)
        for (int a = 0; a < 3; ++a) offs.o[k][a] = offsets_zyx[k * 3 + a];
    const int nb = tem_grid_1d(V, 256, SP_MAXB);
    char* p = (char*)ws;
    float* part = (float*)ws_take(p, (int64_t)K * nb * 3 * 4);
    double* ks = (double*)ws_take(p, (int64_t)K * 24);
    TEM_REQUIRE(p - (char*)ws <= ws_bytes, "tem_affinity_side: workspace too small");
#define CALL(EMV)                                                                                                   \
    do {                                                                                                            \
        hipLaunchKernelGGL((k_aff_sums<EMV>), dim3(nb, K), dim3(256), 0, s, emb, cs, lbl, D, H, W, E, offs, delta, part); \
        hipLaunchKernelGGL(k_aff_rows, dim3(K), dim3(256), 0, s, part, nb, (double)eps, ks);                        \
        hipLaunchKernelGGL(k_aff_final, dim3(1), dim3(64), 0, s, (const double*)ks, K, value_out);                   \
        if (grad)                                                                                                   \
            hipLaunchKernelGGL((k_aff_grad<EMV>), dim3(tem_grid_1d(V, 256)), dim3(256), 0, s, emb, cs, lbl, D, H, W, E, \
                               K, offs, delta, (const double*)ks, (double)eps, grad_scale, grad, gcs);               \
    } while (0)
    SP_DISPATCH_E(E, CALL);
#undef CALL
    TEM_CHECK_LAUNCH("tem_affinity_side");
    return TEM_OK;
}
