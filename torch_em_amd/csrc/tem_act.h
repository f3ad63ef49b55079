// tem_act.h -- element types of ACTIVATION tensors in HBM (round 5: 16-bit storage for the mixed-precision modes, the
// byte half of the reference's torch.autocast, /root/reference/torch_em/trainer/default_trainer.py:134-142, 781-794).
//
// A tensor between two kernels of the step is fp32 (TEM_ST_F32, every fp32-class mode), fp16 (TEM_ST_F16, "amp") or bf16
// (TEM_ST_BF16, "amp_bf16"); statistics, coefficients, parameters, the gradient arena, split-K partials and the network
// output stay fp32.  Kernels are templated on the element type; all arithmetic stays fp32 -- a value is widened when it is
// loaded and rounded ONCE (to nearest even) when it is stored, as autocast does per op.  Leading dimensions are in ELEMENTS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 tem_f16;
typedef __bf16 tem_bf16;

template <typename T> struct TemSt;
template <> struct TemSt<float> { static constexpr int id = 0; };
template <> struct TemSt<tem_f16> { static constexpr int id = 1; };
template <> struct TemSt<tem_bf16> { static constexpr int id = 2; };

static inline int tem_st_size(int st) { return st == 0 ? 4 : 2; }

// storage types of the call in flight on this thread (set by the *_st / *_ex entry points for the duration of the call only:
// internal plumbing between an entry point and the launchers it reaches, never visible across calls)
struct TemCallSt {
    int x;   // input-side tensors  (forward: x;      weight gradient: x)
    int y;   // output-side tensors (forward: y, ref; weight gradient: g and the gnorm y)
};
extern thread_local TemCallSt tem_call_st;
struct TemStScope {
    TemCallSt prev;
    TemStScope(int stx, int sty) : prev(tem_call_st) { tem_call_st = TemCallSt{stx, sty}; }
    ~TemStScope() { tem_call_st = prev; }
};

// Channel-CHUNK strides of the call in flight (tem_conv3d_fwd_ex / tem_conv3d_wgrad_ex: x_cs / y_cs, in elements; 0 = the
// channels of a voxel are contiguous).  With a stride the 32-channel chunk k of a voxel lives at base + k * stride + voxel * ld:
// the two halves of a 2 x 32-channel concat as two DENSE planes, so that a 16-bit half is a whole 128-byte line per two
// voxels instead of half of every line (DESIGN.md 6.R5 "half lines").  Only the z-reuse forward / data-gradient kernel and the
// transposing z-sliding weight gradient on 16-bit tensors take them; every other launch site refuses a call that carries one.
struct TemCallCs {
    int64_t x, y;
};
extern thread_local TemCallCs tem_call_cs;
struct TemCsScope {
    TemCallCs prev;
    TemCsScope(int64_t x_cs, int64_t y_cs) : prev(tem_call_cs) { tem_call_cs = TemCallCs{x_cs, y_cs}; }
    ~TemCsScope() { tem_call_cs = prev; }
};

// run `...` with T bound to the element type of storage id `st`
#define TEM_ST_SWITCH(st, T, ...)                                   \
    do {                                                            \
        switch (st) {                                               \
            case 0: { using T = float; __VA_ARGS__; } break;        \
            case 1: { using T = tem_f16; __VA_ARGS__; } break;      \
            default: { using T = tem_bf16; __VA_ARGS__; } break;    \
        }                                                           \
    } while (0)
// ... for the 16-bit types only (kernels whose fp32 instantiation is spelled out elsewhere)
#define TEM_ST16_SWITCH(st, T, ...)                                 \
    do {                                                            \
        if ((st) == 1) { using T = tem_f16; __VA_ARGS__; }          \
        else { using T = tem_bf16; __VA_ARGS__; }                   \
    } while (0)

typedef float act_f4 __attribute__((ext_vector_type(4)));
typedef float act_f2 __attribute__((ext_vector_type(2)));
typedef unsigned act_u2 __attribute__((ext_vector_type(2)));
typedef unsigned act_u4 __attribute__((ext_vector_type(4)));
typedef _Float16 act_h2 __attribute__((ext_vector_type(2)));
typedef __bf16 act_b2 __attribute__((ext_vector_type(2)));

// ---- two packed 16-bit values <-> two floats ----
template <typename T> __device__ __forceinline__ float act_lo(unsigned u);
template <typename T> __device__ __forceinline__ float act_hi(unsigned u);
template <> __device__ __forceinline__ float act_lo<tem_f16>(unsigned u) { return (float)__builtin_bit_cast(act_h2, u).x; }
template <> __device__ __forceinline__ float act_hi<tem_f16>(unsigned u) { return (float)__builtin_bit_cast(act_h2, u).y; }
template <> __device__ __forceinline__ float act_lo<tem_bf16>(unsigned u) { return __builtin_bit_cast(float, u << 16); }
template <> __device__ __forceinline__ float act_hi<tem_bf16>(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
template <typename T> __device__ __forceinline__ unsigned act_pk(float a, float b);
template <> __device__ __forceinline__ unsigned act_pk<tem_f16>(float a, float b) {
    const act_h2 v = {(_Float16)a, (_Float16)b};   // v_cvt_pk_f16_f32: round to nearest even, overflow -> inf (as torch's .half())
    return __builtin_bit_cast(unsigned, v);
}
template <> __device__ __forceinline__ unsigned act_pk<tem_bf16>(float a, float b) {
    const act_b2 v = {(__bf16)a, (__bf16)b};       // v_cvt_pk_bf16_f32: round to nearest even
    return __builtin_bit_cast(unsigned, v);
}

// ---- one element ----
__device__ __forceinline__ float act_ld1(const float* p) { return *p; }
__device__ __forceinline__ float act_ld1(const tem_f16* p) { return (float)*p; }
__device__ __forceinline__ float act_ld1(const tem_bf16* p) { return (float)*p; }
__device__ __forceinline__ void act_st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void act_st1(tem_f16* p, float v) { *p = (_Float16)v; }
__device__ __forceinline__ void act_st1(tem_bf16* p, float v) { *p = (__bf16)v; }

// ---- two consecutive elements (8 / 4 bytes) ----
__device__ __forceinline__ act_f2 act_ld2(const float* p) { return *reinterpret_cast<const act_f2*>(p); }
template <typename T> __device__ __forceinline__ act_f2 act_ld2(const T* p) {
    const unsigned u = *reinterpret_cast<const unsigned*>(p);
    return act_f2{act_lo<T>(u), act_hi<T>(u)};
}
__device__ __forceinline__ void act_st2(float* p, act_f2 v) { *reinterpret_cast<act_f2*>(p) = v; }
template <typename T> __device__ __forceinline__ void act_st2(T* p, act_f2 v) { *reinterpret_cast<unsigned*>(p) = act_pk<T>(v.x, v.y); }
// ... streaming stores (tensors far beyond the caches that the next kernel reads from HBM anyway)
__device__ __forceinline__ void act_st2_nt(float* p, act_f2 v) { __builtin_nontemporal_store(v, reinterpret_cast<act_f2*>(p)); }
template <typename T> __device__ __forceinline__ void act_st2_nt(T* p, act_f2 v) { __builtin_nontemporal_store(act_pk<T>(v.x, v.y), reinterpret_cast<unsigned*>(p)); }

// ---- four consecutive elements (16 / 8 bytes) ----
__device__ __forceinline__ float4 act_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <typename T> __device__ __forceinline__ float4 act_ld4(const T* p) {
    const act_u2 u = *reinterpret_cast<const act_u2*>(p);
    return make_float4(act_lo<T>(u.x), act_hi<T>(u.x), act_lo<T>(u.y), act_hi<T>(u.y));
}
__device__ __forceinline__ float4 act_ld4_nt(const float* p) {
    const act_f4 v = __builtin_nontemporal_load(reinterpret_cast<const act_f4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
template <typename T> __device__ __forceinline__ float4 act_ld4_nt(const T* p) {
    const act_u2 u = __builtin_nontemporal_load(reinterpret_cast<const act_u2*>(p));
    return make_float4(act_lo<T>(u.x), act_hi<T>(u.x), act_lo<T>(u.y), act_hi<T>(u.y));
}
__device__ __forceinline__ void act_st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
template <typename T> __device__ __forceinline__ void act_st4(T* p, float4 v) {
    *reinterpret_cast<act_u2*>(p) = act_u2{act_pk<T>(v.x, v.y), act_pk<T>(v.z, v.w)};
}
__device__ __forceinline__ void act_st4_nt(float* p, float4 v) {
    const act_f4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<act_f4*>(p));
}
template <typename T> __device__ __forceinline__ void act_st4_nt(T* p, float4 v) {
    const act_u2 t = {act_pk<T>(v.x, v.y), act_pk<T>(v.z, v.w)};
    __builtin_nontemporal_store(t, reinterpret_cast<act_u2*>(p));
}

// ---- eight consecutive elements of a 16-bit tensor (16 bytes) ----
template <typename T> __device__ __forceinline__ void act_ld8(const T* p, float (&v)[8]) {
    const act_u4 u = *reinterpret_cast<const act_u4*>(p);
    v[0] = act_lo<T>(u.x); v[1] = act_hi<T>(u.x); v[2] = act_lo<T>(u.y); v[3] = act_hi<T>(u.y);
    v[4] = act_lo<T>(u.z); v[5] = act_hi<T>(u.z); v[6] = act_lo<T>(u.w); v[7] = act_hi<T>(u.w);
}
template <typename T> __device__ __forceinline__ void act_st8(T* p, const float (&v)[8]) {
    *reinterpret_cast<act_u4*>(p) = act_u4{act_pk<T>(v[0], v[1]), act_pk<T>(v[2], v[3]), act_pk<T>(v[4], v[5]), act_pk<T>(v[6], v[7])};
}

// alignment a vector of 4 elements needs (bytes) / rows whose leading dimension keeps it
template <typename T> struct ActAlign4 { static constexpr uintptr_t bytes = sizeof(T) * 4; };
static inline uintptr_t tem_st_align4(int st) { return st == 0 ? 16 : 8; }

// run `...` with TX / TY bound to the element types of an (input, output) storage pair: both fp32, one side fp32 and the
// other 16-bit, or both the SAME 16-bit type (the pairs a training step produces); `bad` runs for any other pair
static inline bool tem_st2_ok(int stx, int sty) { return !(stx && sty && stx != sty); }   // the pairs TEM_ST2_SWITCH instantiates
#define TEM_ST2_SWITCH(stx, sty, TX, TY, bad, ...)                                                  \
    do {                                                                                            \
        const int k__ = (stx) * 3 + (sty);                                                          \
        switch (k__) {                                                                              \
            case 0: { using TX = float; using TY = float; __VA_ARGS__; } break;                     \
            case 1: { using TX = float; using TY = tem_f16; __VA_ARGS__; } break;                   \
            case 2: { using TX = float; using TY = tem_bf16; __VA_ARGS__; } break;                  \
            case 3: { using TX = tem_f16; using TY = float; __VA_ARGS__; } break;                   \
            case 4: { using TX = tem_f16; using TY = tem_f16; __VA_ARGS__; } break;                 \
            case 6: { using TX = tem_bf16; using TY = float; __VA_ARGS__; } break;                  \
            case 8: { using TX = tem_bf16; using TY = tem_bf16; __VA_ARGS__; } break;               \
            default: { bad; } break;                                                                \
        }                                                                                           \
    } while (0)
