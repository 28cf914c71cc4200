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

// Two GELUs per instruction stream (round 5): the same operations in the same order as bc_gelu_f - bit-identical results - on
// v_pk_fma_f32 / v_pk_mul_f32 (two fp32 lanes per VALU issue on gfx90a+); the two transcendentals per element stay scalar.  The GEGLU
// epilogues are VALU-bound (row-chain feed-forward, 320 channels: 26k of a workgroup's 143k cycles before this).
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef BC_GELU_LIBM
__device__ __forceinline__ f32x2 bc_gelu_f2(f32x2 x) {
    const f32x2 ax = {fabsf(x.x), fabsf(x.y)};
    const f32x2 z = ax * 0.70710678118654752f;
    const f32x2 d = __builtin_elementwise_fma((f32x2)(0.3275911f), z, (f32x2)(1.0f));
    const f32x2 t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    f32x2 p = __builtin_elementwise_fma(t, (f32x2)(1.061405429f), (f32x2)(-1.453152027f));
    p = __builtin_elementwise_fma(t, p, (f32x2)(1.421413741f));
    p = __builtin_elementwise_fma(t, p, (f32x2)(-0.284496736f));
    p = __builtin_elementwise_fma(t, p, (f32x2)(0.254829592f));
    const f32x2 poly = t * p;
    const f32x2 ea = (z * z) * (-1.4426950408889634f);
    const f32x2 e = {__builtin_amdgcn_exp2f(ea.x), __builtin_amdgcn_exp2f(ea.y)};
    const f32x2 erfc = poly * e;
    const f32x2 w = {x.x >= 0.0f ? 2.0f - erfc.x : erfc.x, x.y >= 0.0f ? 2.0f - erfc.y : erfc.y};
    return (0.5f * x) * w;
}
#else
__device__ __forceinline__ f32x2 bc_gelu_f2(f32x2 x) { return (f32x2){bc_gelu_f(x.x), bc_gelu_f(x.y)}; }
#endif

// CLIP's "quick_gelu": x * sigmoid(1.702 x)  (transformers activations.QuickGELUActivation)
__device__ __forceinline__ float bc_quick_gelu_f(float x) { return x / (1.0f + __expf(-1.702f * x)); }

// ---- GroupNorm statistics as ORDER-INDEPENDENT totals (round 4) -------------------------------------------------------------------
// Per (image, channel) BC_GN_TOT_WORDS 64-bit integer accumulators: the sum and the sum of squares of the fp16 activation, each as three
// signed 40-bit slices of a fixed-point number (units 2^-60, 2^-20, 2^20).  Every producer workgroup reduces its rows in fp32 as
// before and then ADDS that partial with integer atomics: integer addition commutes, so the totals - and everything downstream - are
// bit-reproducible whatever the arrival order (an fp32 partial below 2^50 in magnitude converts exactly - fp16 activations give at
// most 2^46 per workgroup partial -; the slices cannot overflow below 2^14 addends).  A non-finite or absurd partial adds the POISON
// 2^50 to slice 2 instead: honest slice-2 sums stay below 2^44 in magnitude, so any count of 1 ... 8191 poisoned addends leaves
// |slice 2| >= 2^45 (the round-4 poison 2^60 cancelled itself after 16 addends); bc_gn_tot_read turns that into NaN.  Consumers read six words per channel instead of re-reducing `nslab` partials, and no finalize launch
// sits between a producer and a fused consumer.  The tables are zeroed once per replay (bc_memset_zero at the head of a segment).
__device__ __forceinline__ void bc_gn_slices(float v, long long (&sl)[3]) {
    if (!(fabsf(v) < 1.1e15f)) {                       // inf / NaN / absurd: poison the total (consumers turn it into NaN)
        sl[0] = 0; sl[1] = 0; sl[2] = 1ll << 50;
        return;
    }
    const double d = (double)v;
    const double h = floor(d * 9.5367431640625e-07);   // 2^-20
    const double r = d - h * 1048576.0;                // in [0, 2^20)
    const double m = floor(r * 1048576.0);
    const double r2 = r - m * 9.5367431640625e-07;     // in [0, 2^-20)
    sl[2] = (long long)h;
    sl[1] = (long long)m;
    sl[0] = (long long)floor(r2 * 1152921504606846976.0);   // 2^60
}
__device__ __forceinline__ void bc_gn_tot_add(unsigned long long* t, float s, float q) {
    long long a[3], b[3];
    bc_gn_slices(s, a);
    bc_gn_slices(q, b);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (a[i]) __hip_atomic_fetch_add(t + i, (unsigned long long)a[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (b[i]) __hip_atomic_fetch_add(t + 3 + i, (unsigned long long)b[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// Channels per totals BLOCK (round 6).  Atomics on one cache line execute one after the other, ~22 ns each (tools/gn_atomic_probe.hip): the 64
// row tiles of an image adding 2 - 5 slices per channel into the same 48 bytes kept every launch that produces statistics alive 3 - 5 us after
// its last store - 5 % of the batch-1 step (a build that stores instead of adding: 8.64 -> 8.20 ms).  GroupNorm never needs one channel's
// statistics, only a group's.  In a table whose width is a multiple of 320 - every GroupNorm input of the UNet / BlobNet: 32 groups of 10 k
// channels, also over concatenated sources - a producer therefore adds ONE partial per block of 10 consecutive channels, and it adds it into
// slot (block + spread % 10): the ten channel slots of a block are ten accumulators of the block's sums, picked by the producer's row-tile
// index.  A tenth of the atomics, a tenth of the chain per line, and no consumer changes: a group's sum over its channels' slots is the same
// sum.  MEANINGFUL ARE ONLY SUMS OVER WHOLE BLOCKS - launch.py refuses the producer's totals to a GroupNorm whose groups are not unions of
// blocks (it runs a statistics pass instead).  Tables of other widths (the VAE: 128 / 256 / 512 channels, groups of 4 - 16) stay per channel.
__host__ __device__ __forceinline__ int bc_gn_cg(int n_out) { return n_out % 320 == 0 ? 10 : 1; }
// The thread that holds column n of a workgroup covering columns [n_lo, n_hi) adds what is its to add: with cg == 1 its own column; else, if n is
// the first column of its block inside [n_lo, n_hi), the block's columns inside that range (col(k) -> (sum, sum of squares) of column k, from LDS).
template <class F>
__device__ __forceinline__ void bc_gn_tot_add_slot(unsigned long long* table_of_image, int n, int n_lo, int n_hi, int cg, int spread, F&& col) {
    if (cg <= 1) {
        const float2 v = col(n);
        bc_gn_tot_add(table_of_image + (size_t)n * BC_GN_TOT_WORDS, v.x, v.y);
        return;
    }
    const int d0 = n / cg * cg;
    if (n != (d0 > n_lo ? d0 : n_lo)) return;
    const int last = d0 + cg < n_hi ? d0 + cg : n_hi;
    float s = 0.f, q = 0.f;
    for (int k = n; k < last; ++k) {
        const float2 v = col(k);
        s += v.x;
        q += v.y;
    }
    bc_gn_tot_add(table_of_image + (size_t)(d0 + (unsigned)spread % (unsigned)cg) * BC_GN_TOT_WORDS, s, q);
}
// (sum, sum of squares) of one channel from its six words; a poisoned total reads as NaN
__device__ __forceinline__ void bc_gn_tot_decode(unsigned long long w0, unsigned long long w1, unsigned long long w2, unsigned long long w3,
                                                 unsigned long long w4, unsigned long long w5, double& s, double& q) {
    const long long a0 = (long long)w0, a1 = (long long)w1, a2 = (long long)w2;
    const long long b0 = (long long)w3, b1 = (long long)w4, b2 = (long long)w5;
    s = (double)a0 * 8.673617379884035e-19 + (double)a1 * 9.5367431640625e-07 + (double)a2 * 1048576.0;
    q = (double)b0 * 8.673617379884035e-19 + (double)b1 * 9.5367431640625e-07 + (double)b2 * 1048576.0;
    const long long lim = 1ll << 45;
    if (a2 >= lim || a2 <= -lim || b2 >= lim || b2 <= -lim) { s = __builtin_nan(""); q = s; }
}
__device__ __forceinline__ void bc_gn_tot_read(const unsigned long long* t, double& s, double& q) {
    bc_gn_tot_decode(t[0], t[1], t[2], t[3], t[4], t[5], s, q);
}

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
