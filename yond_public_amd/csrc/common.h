// Shared helpers for the gfx950 kernels of libyond_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/yond_hip.h"

// Experiment switches (grid sizes, tile widths of tools/ A/B runs) exist only in -DYOND_EXPERIMENTS builds
// (python -m yond_public_amd.build --experiments): the product library reads no environment variable.
#ifdef YOND_EXPERIMENTS
#include <stdlib.h>
static inline long yond_exp_long(const char* name, long dflt) {
    const char* e = getenv(name);
    return e ? atol(e) : dflt;
}
#else
#define yond_exp_long(name, dflt) (dflt)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define YOND_LAUNCH_CHECK()                          \
    do {                                             \
        hipError_t e__ = hipGetLastError();          \
        if (e__ != hipSuccess) return (int)e__;      \
    } while (0)

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }

__device__ __forceinline__ int reflect101(int i, int n) {
    // BORDER_REFLECT_101 / torch 'reflect' for any i (multiple reflections allowed), n >= 1
    if (n == 1) return 0;
    const int period = 2 * (n - 1);
    i %= period;
    if (i < 0) i += period;
    return i >= n ? period - i : i;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// compile-time loop: f(IntC<B>{}), ..., f(IntC<E-1>{}) -- the index is usable in constant expressions
template <int I>
struct IntC {
    static constexpr int value = I;
};
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(IntC<B>{});
        static_for<B + 1, E>(f);
    }
}
