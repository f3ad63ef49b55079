// optim.hip -- fused optimizer steps over a flat parameter arena.
// tem_adamw_step: torch.optim.AdamW exactly as default_segmentation_trainer configures it
// (reference segmentation.py:543; amsgrad=False, maximize=False):
//   p *= 1 - lr*wd ; m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g^2 ;
//   p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// One launch for all 46 parameter tensors (HBM-bound: 4 reads + 3 writes of 85 MB).
// tem_ema_update: SPOCOTrainer._momentum_update (reference trainer/spoco_trainer.py:45-47).
#include "tem_common.h"
#include <math.h>

__global__ __launch_bounds__(256) void k_adamw(float* __restrict__ p, const float* __restrict__ g,
                                               float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                               float b1, float b2, float eps, float wd, float step_size,
                                               float inv_sqrt_bc2, float gscale) {
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 p4 = reinterpret_cast<float4*>(p)[i];
        float4 g4 = reinterpret_cast<const float4*>(g)[i];
        float4 m4 = reinterpret_cast<float4*>(m)[i];
        float4 v4 = reinterpret_cast<float4*>(v)[i];
        float pp[4] = {p4.x, p4.y, p4.z, p4.w}, gg[4] = {g4.x, g4.y, g4.z, g4.w};
        float mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float gr = gg[j] * gscale;
            pp[j] *= (1.f - lr * wd);
            mm[j] = mm[j] + (1.f - b1) * (gr - mm[j]);          // lerp_
            vv[j] = b2 * vv[j] + (1.f - b2) * gr * gr;            // mul_ + addcmul_
            float denom = sqrtf(vv[j]) * inv_sqrt_bc2 + eps;
            pp[j] -= step_size * (mm[j] / denom);
        }
        reinterpret_cast<float4*>(p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
        reinterpret_cast<float4*>(m)[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
        reinterpret_cast<float4*>(v)[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
    }
    // tail
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        float gr = g[i] * gscale;
        float pp = p[i] * (1.f - lr * wd);
        float mm = m[i] + (1.f - b1) * (gr - m[i]);
        float vv = b2 * v[i] + (1.f - b2) * gr * gr;
        float denom = sqrtf(vv) * inv_sqrt_bc2 + eps;
        p[i] = pp - step_size * (mm / denom);
        m[i] = mm;
        v[i] = vv;
    }
}

extern "C" int tem_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                              float beta1, float beta2, float eps, float weight_decay, int64_t step, float grad_scale,
                              tem_stream_t stream) {
    TEM_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "tem_adamw_step: bad arguments");
    TEM_REQUIRE(((uintptr_t)param % 16 == 0) && ((uintptr_t)grad % 16 == 0) && ((uintptr_t)exp_avg % 16 == 0) &&
                    ((uintptr_t)exp_avg_sq % 16 == 0),
                "tem_adamw_step: arena pointers must be 16-byte aligned");
    double bc1 = 1.0 - pow((double)beta1, (double)step);
    double bc2 = 1.0 - pow((double)beta2, (double)step);
    float step_size = (float)((double)lr / bc1);
    float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    hipLaunchKernelGGL(k_adamw, dim3(tem_grid_1d(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, param, grad,
                       exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step_size, inv_sqrt_bc2, grad_scale);
    TEM_CHECK_LAUNCH("tem_adamw_step");
    return TEM_OK;
}

// The same update with its scalars read from DEVICE memory: hyper = [lr, beta1, beta2, eps, weight_decay, step_size,
// inv_sqrt_bc2, grad_scale, skip, ...].  A launch captured in a HIP graph has its kernel arguments frozen; the step
// count (bias corrections) and the learning rate change every step, so a captured optimizer step takes them from a
// 12-float buffer the host refreshes before each replay.  skip != 0 leaves everything untouched (an overflowed
// mixed-precision step).
__global__ __launch_bounds__(256) void k_adamw_dev(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   const float* __restrict__ hyper) {
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], step_size = hyper[5],
                inv_sqrt_bc2 = hyper[6], gscale = hyper[7];
    if (hyper[8] != 0.f) return;
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 p4 = reinterpret_cast<float4*>(p)[i];
        float4 g4 = reinterpret_cast<const float4*>(g)[i];
        float4 m4 = reinterpret_cast<float4*>(m)[i];
        float4 v4 = reinterpret_cast<float4*>(v)[i];
        float pp[4] = {p4.x, p4.y, p4.z, p4.w}, gg[4] = {g4.x, g4.y, g4.z, g4.w};
        float mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float gr = gg[j] * gscale;
            pp[j] *= (1.f - lr * wd);
            mm[j] = mm[j] + (1.f - b1) * (gr - mm[j]);
            vv[j] = b2 * vv[j] + (1.f - b2) * gr * gr;
            float denom = sqrtf(vv[j]) * inv_sqrt_bc2 + eps;
            pp[j] -= step_size * (mm[j] / denom);
        }
        reinterpret_cast<float4*>(p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
        reinterpret_cast<float4*>(m)[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
        reinterpret_cast<float4*>(v)[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        float gr = g[i] * gscale;
        float pp = p[i] * (1.f - lr * wd);
        float mm = m[i] + (1.f - b1) * (gr - m[i]);
        float vv = b2 * v[i] + (1.f - b2) * gr * gr;
        float denom = sqrtf(vv) * inv_sqrt_bc2 + eps;
        p[i] = pp - step_size * (mm / denom);
        m[i] = mm;
        v[i] = vv;
    }
}

// ... and under dynamic loss scaling, where a step may be SKIPPED on the device (overflow) without the host knowing
// before it enqueues the next one: sstate = [scale, growth_tracker, found_inf, applied_steps] lives on the device, the
// host uploads the scalar rows of a WINDOW of step numbers, table = [lo, J, -, -] + J x 12 floats (row j: step lo + j,
// as tem_adamw_hyper fills it), and the kernel takes row applied_steps + 1 - lo.  found_inf != 0: no update.
__global__ __launch_bounds__(256) void k_adamw_tab(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   const float* __restrict__ table, float* __restrict__ sstate) {
    if (sstate[2] != 0.f) return;
    const int lo = (int)table[0], J = (int)table[1];
    const int row = (int)sstate[3] + 1 - lo;
    if (row < 0 || row >= J) {
        // The host's lower bound of the applied-step count is older than the window it uploaded (it must never be: the
        // owner of the graph bounds its lead, torch_em_amd/graph.py READBACK_SLOTS < TABLE_ROWS - 1).  Applying the bias
        // corrections of another step number would be silently wrong, so NO workgroup updates anything (all of them take
        // this branch: same inputs) and the step is reported like an overflow: skipped, visible in the scaler's state.
        if (blockIdx.x == 0 && threadIdx.x == 0) sstate[2] = 1.f;
        return;
    }
    const float* hyper = table + 4 + 12 * row;
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], step_size = hyper[5],
                inv_sqrt_bc2 = hyper[6], gscale = hyper[7];
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 p4 = reinterpret_cast<float4*>(p)[i];
        float4 g4 = reinterpret_cast<const float4*>(g)[i];
        float4 m4 = reinterpret_cast<float4*>(m)[i];
        float4 v4 = reinterpret_cast<float4*>(v)[i];
        float pp[4] = {p4.x, p4.y, p4.z, p4.w}, gg[4] = {g4.x, g4.y, g4.z, g4.w};
        float mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float gr = gg[j] * gscale;
            pp[j] *= (1.f - lr * wd);
            mm[j] = mm[j] + (1.f - b1) * (gr - mm[j]);
            vv[j] = b2 * vv[j] + (1.f - b2) * gr * gr;
            float denom = sqrtf(vv[j]) * inv_sqrt_bc2 + eps;
            pp[j] -= step_size * (mm[j] / denom);
        }
        reinterpret_cast<float4*>(p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
        reinterpret_cast<float4*>(m)[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
        reinterpret_cast<float4*>(v)[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        float gr = g[i] * gscale;
        float pp = p[i] * (1.f - lr * wd);
        float mm = m[i] + (1.f - b1) * (gr - m[i]);
        float vv = b2 * v[i] + (1.f - b2) * gr * gr;
        float denom = sqrtf(vv) * inv_sqrt_bc2 + eps;
        p[i] = pp - step_size * (mm / denom);
        m[i] = mm;
        v[i] = vv;
    }
}

extern "C" int tem_adamw_step_tab(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                  const float* table, float* sstate, tem_stream_t stream) {
    TEM_REQUIRE(param && grad && exp_avg && exp_avg_sq && table && sstate && n > 0, "tem_adamw_step_tab: bad arguments");
    TEM_REQUIRE(((uintptr_t)param % 16 == 0) && ((uintptr_t)grad % 16 == 0) && ((uintptr_t)exp_avg % 16 == 0) &&
                    ((uintptr_t)exp_avg_sq % 16 == 0),
                "tem_adamw_step_tab: arena pointers must be 16-byte aligned");
    hipLaunchKernelGGL(k_adamw_tab, dim3(tem_grid_1d(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, param, grad,
                       exp_avg, exp_avg_sq, n, table, sstate);
    TEM_CHECK_LAUNCH("tem_adamw_step_tab");
    return TEM_OK;
}

extern "C" int tem_adamw_hyper(float* hyper_host, float lr, float beta1, float beta2, float eps, float weight_decay,
                               int64_t step, float grad_scale) {
    TEM_REQUIRE(hyper_host && step >= 1, "tem_adamw_hyper: bad arguments");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hyper_host[0] = lr; hyper_host[1] = beta1; hyper_host[2] = beta2; hyper_host[3] = eps; hyper_host[4] = weight_decay;
    hyper_host[5] = (float)((double)lr / bc1);
    hyper_host[6] = (float)(1.0 / sqrt(bc2));
    hyper_host[7] = grad_scale;
    hyper_host[8] = 0.f;
    hyper_host[9] = hyper_host[10] = hyper_host[11] = 0.f;
    return TEM_OK;
}

extern "C" int tem_adamw_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                  const float* hyper, tem_stream_t stream) {
    TEM_REQUIRE(param && grad && exp_avg && exp_avg_sq && hyper && n > 0, "tem_adamw_step_dev: bad arguments");
    TEM_REQUIRE(((uintptr_t)param % 16 == 0) && ((uintptr_t)grad % 16 == 0) && ((uintptr_t)exp_avg % 16 == 0) &&
                    ((uintptr_t)exp_avg_sq % 16 == 0),
                "tem_adamw_step_dev: arena pointers must be 16-byte aligned");
    hipLaunchKernelGGL(k_adamw_dev, dim3(tem_grid_1d(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, param, grad,
                       exp_avg, exp_avg_sq, n, hyper);
    TEM_CHECK_LAUNCH("tem_adamw_step_dev");
    return TEM_OK;
}

__global__ __launch_bounds__(256) void k_ema(float* __restrict__ k, const float* __restrict__ q, int64_t n, float mom) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        k[i] = k[i] * mom + q[i] * (1.f - mom);
}

extern "C" int tem_ema_update(float* theta_k, const float* theta_q, int64_t n, float momentum, tem_stream_t stream) {
    TEM_REQUIRE(theta_k && theta_q && n > 0, "tem_ema_update: bad arguments");
    hipLaunchKernelGGL(k_ema, dim3(tem_grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream, theta_k, theta_q, n,
                       momentum);
    TEM_CHECK_LAUNCH("tem_ema_update");
    return TEM_OK;
}

// GradScaler.unscale_ of the mixed-precision path (reference trainer/default_trainer.py:789-794 drives
// torch.amp.GradScaler: _amp_foreach_non_finite_check_and_unscale_): g *= inv_scale in place; *found_inf = 1 when any
// element is inf/NaN (the fp16 operand rounding of use_mfma = 5 overflowed).  Every writer stores the same value.
__global__ __launch_bounds__(256) void k_amp_unscale(float* __restrict__ g, int64_t n, int64_t n4, float inv_scale,
                                                     float* __restrict__ found_inf) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 v = reinterpret_cast<float4*>(g)[i];
        v.x *= inv_scale;
        v.y *= inv_scale;
        v.z *= inv_scale;
        v.w *= inv_scale;
        bad |= !(isfinite(v.x) && isfinite(v.y) && isfinite(v.z) && isfinite(v.w));
        reinterpret_cast<float4*>(g)[i] = v;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float v = g[i] * inv_scale;
        bad |= !isfinite(v);
        g[i] = v;
    }
    if (bad) *found_inf = 1.f;
}

// GradScaler with its state on the device (HIP-graph capture; what torch.amp.GradScaler does with its scale tensor):
// sstate = [scale, growth_tracker, found_inf, applied_steps].
__global__ __launch_bounds__(256) void k_amp_unscale_dev(float* __restrict__ g, int64_t n, int64_t n4,
                                                         float* __restrict__ sstate) {
    const float inv_scale = 1.f / sstate[0];
    const int64_t stride = (int64_t)gridDim.x * 256;
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 v = reinterpret_cast<float4*>(g)[i];
        v.x *= inv_scale;
        v.y *= inv_scale;
        v.z *= inv_scale;
        v.w *= inv_scale;
        bad |= !(isfinite(v.x) && isfinite(v.y) && isfinite(v.z) && isfinite(v.w));
        reinterpret_cast<float4*>(g)[i] = v;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float v = g[i] * inv_scale;
        bad |= !isfinite(v);
        g[i] = v;
    }
    if (bad) sstate[2] = 1.f;
}

extern "C" int tem_amp_unscale_dev(float* grad, int64_t n, float* sstate, tem_stream_t stream) {
    TEM_REQUIRE(grad && sstate && n > 0, "tem_amp_unscale_dev: bad arguments");
    const int64_t n4 = ((uintptr_t)grad % 16 == 0) ? (n >> 2) : 0;
    hipLaunchKernelGGL(k_amp_unscale_dev, dim3(tem_grid_1d(n4 ? n / 4 + 1 : n, 256)), dim3(256), 0, (hipStream_t)stream,
                       grad, n, n4, sstate);
    TEM_CHECK_LAUNCH("tem_amp_unscale_dev");
    return TEM_OK;
}

// GradScaler.update(): torch's _amp_update_scale_ (halve after an overflow, double after `interval` clean steps), plus
// the count of applied optimizer steps (the row index of tem_adamw_step_tab) and the reset of the overflow flag.
__global__ void k_amp_update_dev(float* __restrict__ sstate, float growth, float backoff, int interval) {
    if (threadIdx.x || blockIdx.x) return;
    if (sstate[2] != 0.f) {
        sstate[0] *= backoff;
        sstate[1] = 0.f;
    } else {
        const float t = sstate[1] + 1.f;
        if ((int)t == interval) {
            sstate[0] *= growth;
            sstate[1] = 0.f;
        } else {
            sstate[1] = t;
        }
        sstate[3] += 1.f;
    }
    sstate[2] = 0.f;
}

extern "C" int tem_amp_update_dev(float* sstate, float growth, float backoff, int interval, tem_stream_t stream) {
    TEM_REQUIRE(sstate && growth > 1.f && backoff < 1.f && interval > 0, "tem_amp_update_dev: bad arguments");
    hipLaunchKernelGGL(k_amp_update_dev, dim3(1), dim3(64), 0, (hipStream_t)stream, sstate, growth, backoff, interval);
    TEM_CHECK_LAUNCH("tem_amp_update_dev");
    return TEM_OK;
}

extern "C" int tem_amp_unscale(float* grad, int64_t n, float inv_scale, float* found_inf, tem_stream_t stream) {
    TEM_REQUIRE(grad && found_inf && n > 0, "tem_amp_unscale: bad arguments");
    const int64_t n4 = ((uintptr_t)grad % 16 == 0) ? (n >> 2) : 0;  // unaligned views take the scalar path
    hipLaunchKernelGGL(k_amp_unscale, dim3(tem_grid_1d(n4 ? n / 4 + 1 : n, 256)), dim3(256), 0, (hipStream_t)stream, grad,
                       n, n4, inv_scale, found_inf);
    TEM_CHECK_LAUNCH("tem_amp_unscale");
    return TEM_OK;
}
