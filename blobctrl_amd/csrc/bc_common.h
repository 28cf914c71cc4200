// Shared device/host helpers for libblobctrl_hip (gfx950 only: wave64, MFMA 32x32x16 f16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <atomic>
#include "../../include/blobctrl_hip.h"

typedef _Float16 h16;
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

void bc_set_error(const char* fmt, ...);

#define BC_CHECK_ARG(cond, ...)                                                \
    do {                                                                       \
        if (!(cond)) { bc_set_error(__VA_ARGS__); return 1; }                  \
    } while (0)

#define BC_CHECK_HIP(expr)                                                     \
    do {                                                                       \
        hipError_t _e = (expr);                                                \
        if (_e != hipSuccess) {                                                \
            bc_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return 2;                                                          \
        }                                                                      \
    } while (0)

#define BC_CHECK_LAUNCH() BC_CHECK_HIP(hipGetLastError())

// Raise a kernel's dynamic-LDS limit once PER DEVICE (the attribute belongs to the function on the current device).  One atomic
// bit per device ordinal: thread-safe, and racing threads at worst set the same value twice.  Returns a hipError_t.
static inline hipError_t bc_set_max_lds(std::atomic<unsigned long long>& done, const void* func, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
}

static inline int bc_ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float bc_silu_f(float x) { return x / (1.0f + __expf(-x)); }
// exact-erf GELU (activations.py:93-123 uses F.gelu default = erf form)
#ifdef BC_GELU_LIBM
__device__ __forceinline__ float bc_gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
#else
// erfc(|z|) = t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-z^2), t = 1 / (1 + p |z|)  (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7);
// gelu(x) = 0.5 x (2 - erfc) for x >= 0 and 0.5 x erfc for x < 0 (no cancellation on the negative side).  Max abs error 4.2e-7 over
// [-12, 12] against the fp64 erf form, under half an fp16 ulp of the result everywhere (tools/gelu_check.py).  libm's erff is ~4x the
// instructions and made the GEGLU epilogue VALU-bound: 19 of the 75 us of the 16384 x 2560 x 320 projection (-DBC_GELU_LIBM restores it).
__device__ __forceinline__ float bc_gelu_f(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float erfc = poly * __builtin_amdgcn_exp2f(-(z * z) * 1.4426950408889634f);
    return 0.5f * x * (x >= 0.0f ? 2.0f - erfc : erfc);
}
#endif

// CLIP's "quick_gelu": x * sigmoid(1.702 x)  (transformers activations.QuickGELUActivation)
__device__ __forceinline__ float bc_quick_gelu_f(float x) { return x / (1.0f + __expf(-1.702f * x)); }

__device__ __forceinline__ uint4 bc_ld16(const void* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void bc_st16(void* p, uint4 v) { *reinterpret_cast<uint4*>(p) = v; }

// XCD-aware workgroup remap (cdna_hip_programming.md T1): workgroups are dealt round-robin over the 8 XCDs, each with a
// private 4 MiB L2.  Remapping the linear id so that every XCD owns a CONTIGUOUS band of the grid makes tiles that share an
// operand panel (column tiles of one row band, query blocks of one head) hit the same L2.  Bijective for any grid size.
// Placement only affects speed, never results.
__device__ __forceinline__ int bc_xcd_remap(int id, int nwg) {
    const int xcd = id & 7, q = nwg >> 3, r = nwg & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (id >> 3);
}
