// Issue-bound ceiling of the D = 40 flash-attention tile on gfx950 ("the stated target or a measured proof it is
// unreachable").  One 32-query x 64-key tile of csrc/attention.hip costs, per wave, 6 v_mfma_f32_32x32x16_f16 for S^T = K Q^T
// (2 key tiles x 3 k-steps of the head_dim padded to 48), then 17 v_max3_f32 + 32 v_exp_f32 + 16 v_cvt_pk_f16_f32 (+ 2 permlane
// swaps), then 8 MFMAs for O^T += V^T P^T (2 row tiles of the padded 64 x 2 key tiles x 2 k-steps).  This probe runs EXACTLY that
// instruction mix with the same dependencies (S -> softmax -> P -> PV) but with every operand in registers: no LDS reads, no
// LDS-DMA, no barriers, no global memory.  What it measures is therefore the ceiling the SIMD's issue port and matrix pipe allow for
// W co-resident waves per SIMD; whatever the real kernel loses below it is data supply (fragment reads: every MFMA needs a fresh
// 1-KiB operand from LDS), staging and barriers.
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form tools/attn_bound.hip -o attn_bound && ./attn_bound
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

template <int VALU>   // 1: full softmax VALU work, 0: MFMAs only (P taken from constants)
__global__ __launch_bounds__(256) void tile_loop(const _Float16* __restrict__ in, float* __restrict__ out, int tiles) {
    const int lane = threadIdx.x & 63;
    h16x8 kf[3], qf[3], vf[2];
    for (int s = 0; s < 3; ++s) {
        kf[s] = *reinterpret_cast<const h16x8*>(in + (lane * 3 + s) * 8);
        qf[s] = *reinterpret_cast<const h16x8*>(in + 2048 + (lane * 3 + s) * 8);
    }
    for (int s = 0; s < 2; ++s) vf[s] = *reinterpret_cast<const h16x8*>(in + 4096 + (lane * 2 + s) * 8);
    f32x16 oacc[2];
    for (int t = 0; t < 2; ++t)
        for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
    float mrun = 0.f;
    union PFrag { h16x8 v; h16x2 p[4]; };
    for (int it = 0; it < tiles; ++it) {
        f32x16 sacc[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kt], qf[0], z, 0, 0, 0);               // (two different key tiles)
            sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[(kt + 1) % 3], qf[1], sacc[kt], 0, 0, 0);
            sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[(kt + 2) % 3], qf[2], sacc[kt], 0, 0, 0);
        }
        PFrag pf[2][2];
        if (VALU) {
            float mx = max3f(sacc[0][0], sacc[0][1], sacc[0][2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) mx = max3f(mx, sacc[0][r], sacc[0][r + 1]);
            float mx1 = max3f(sacc[1][0], sacc[1][1], sacc[1][2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) mx1 = max3f(mx1, sacc[1][r], sacc[1][r + 1]);
            mx = max3f(mx, mx1, sacc[0][15]);
            mx = fmaxf(mx, sacc[1][15]);
            auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            mrun = fmaxf(mrun, mx);                      // (kept live; the real kernel compares it with its rescale threshold)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        f32x2 e;
                        e.x = __builtin_amdgcn_exp2f(sacc[kt][8 * s2 + 2 * jj]);
                        e.y = __builtin_amdgcn_exp2f(sacc[kt][8 * s2 + 2 * jj + 1]);
                        pf[kt][s2].p[jj] = __builtin_convertvector(e, h16x2);
                    }
        } else {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    pf[kt][s2].v = vf[s2];
                    asm volatile("" ::"v"(sacc[kt][8 * s2]), "v"(sacc[kt][8 * s2 + 7]));       // keep the S MFMAs alive
                }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
                    oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[(t + s2) & 1], pf[kt][s2].v, oacc[t], 0, 0, 0);
        // next tile's operands differ (as in the real kernel): rotate the register-resident fragments
        const h16x8 k0 = kf[0];
        kf[0] = kf[1]; kf[1] = kf[2]; kf[2] = k0;
    }
    float s = mrun;
    for (int t = 0; t < 2; ++t)
        for (int r = 0; r < 16; ++r) s += oacc[t][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    const int tiles = 4096;
    _Float16* in;
    float* out;
    hipMalloc(&in, 8192 * sizeof(_Float16));
    hipMalloc(&out, 256 * 16 * 1024 * sizeof(float));
    _Float16 h[8192];
    srand(1);
    for (int i = 0; i < 8192; ++i) h[i] = (_Float16)((rand() % 2001 - 1000) / 4000.0f);
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    printf("{\"device\": \"%s\", \"cus\": %d, \"tiles_per_wave\": %d, \"mfma_cycles_per_tile\": %d, \"runs\": [", prop.name, cus, tiles, 14 * 32);
    bool first = true;
    for (int valu = 1; valu >= 0; --valu)
        for (int w = 1; w <= 4; ++w) {                         // w waves per SIMD = w workgroups of 4 waves per CU
            const int grid = cus * w;
            float best = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(a, 0);
                if (valu) hipLaunchKernelGGL(tile_loop<1>, dim3(grid), dim3(256), 0, 0, in, out, tiles);
                else hipLaunchKernelGGL(tile_loop<0>, dim3(grid), dim3(256), 0, 0, in, out, tiles);
                hipEventRecord(b, 0);
                hipEventSynchronize(b);
                float ms;
                hipEventElapsedTime(&ms, a, b);
                if (rep > 0 && ms < best) best = ms;
            }
            // every SIMD runs w waves x tiles tiles; matrix-pipe time per tile = 14 MFMAs x 32 cycles
            const double tile_ns = best * 1e6 / ((double)tiles * w);                 // wall time per tile per SIMD
            const double flops = (double)grid * 4 * tiles * 14 * 2.0 * 32 * 32 * 16;
            printf("%s{\"softmax_valu\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"ns_per_tile_per_simd\": %.1f, \"mfma_tflops_padded\": %.0f}",
                   first ? "" : ", ", valu, w, best, tile_ns, flops / best / 1e9);
            first = false;
        }
    printf("]}\n");
    return 0;
}
