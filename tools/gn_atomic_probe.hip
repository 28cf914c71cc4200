// (GPU) What do the GroupNorm-statistics atomics cost a chain of dependent launches?  Each launch: `nwg` workgroups of 512 threads spin for
// ~`spin` cycles (the "kernel"), then 160 threads add `per_col` 64-bit integers each
//   mode 0: nothing (the floor)          mode 1: plain stores into the shared table
//   mode 2: atomics, all workgroups into the SAME 160 columns (the production pattern: 64 row tiles of an image)
//   mode 3: atomics, every workgroup into its OWN 160 columns (no same-address chain)
//   mode 4: atomics into 8 copies of the table, copy = workgroup index mod 8
// Timed: 50 back-to-back launches on one stream (each waits for the previous one to retire), HIP events.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ __launch_bounds__(512) void probe(unsigned long long* tab, int mode, int per_col, int spin, float* sink) {
    const long long t0 = __builtin_amdgcn_s_memtime();
    float acc = threadIdx.x;
    while (__builtin_amdgcn_s_memtime() - t0 < spin) acc = acc * 1.0001f + 0.5f;
    if (acc == 12345.678f) sink[0] = acc;
    if (threadIdx.x < 160 && mode) {
        const int wg = blockIdx.x;
        unsigned long long* t = tab + (size_t)threadIdx.x * 6;
        if (mode == 3) t += (size_t)wg * 160 * 6;
        if (mode == 4) t += (size_t)(wg & 7) * 160 * 6;
        for (int i = 0; i < per_col; ++i) {
            if (mode == 1) t[i] = (unsigned long long)(wg + i);
            else __hip_atomic_fetch_add(t + i, (unsigned long long)(wg + i + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int main() {
    unsigned long long* tab;
    float* sink;
    hipMalloc(&tab, (size_t)2048 * 160 * 6 * 8);
    hipMalloc(&sink, 64);
    hipMemset(tab, 0, (size_t)2048 * 160 * 6 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grids[] = {64, 128, 256, 1024};
    for (int gi = 0; gi < 4; ++gi)
        for (int per_col = 2; per_col <= 6; per_col += 2)
            for (int mode = 0; mode < 5; ++mode) {
                const int nwg = grids[gi], spin = 40000;      // ~20 us of "kernel"
                for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(probe, dim3(nwg), dim3(512), 0, 0, tab, mode, per_col, spin, sink);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                for (int w = 0; w < 50; ++w) hipLaunchKernelGGL(probe, dim3(nwg), dim3(512), 0, 0, tab, mode, per_col, spin, sink);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (per_col == 2 || mode >= 1) printf("grid %4d  per_col %d  mode %d: %7.2f us per launch\n", nwg, per_col, mode, ms * 1000 / 50);
            }
    return 0;
}
