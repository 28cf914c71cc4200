// Premise check for a split-K reduction INSIDE the convolution launch (the K split is the parallelism of the 16 x 32 and 8 x 16
// levels; its reduction is a second launch today):
//   every split workgroup adds its 128 x 160 fp32 partial tile, converted to 64-bit fixed point, into ONE int64 tile with integer
//   atomics (order-independent, so the result stays bit-reproducible), bumps a per-tile counter, and the last one to arrive reads the
//   tile back, runs the epilogue and leaves the tile zeroed for the next launch.
// Questions: (1) what do 20 480 64-bit atomics per workgroup cost at L2 scope (all splits of a tile on one XCD: workgroup id congruent
// mod 8) and at agent scope (memory-side, any placement), against writing an fp32 slab with plain stores; (2) is the workgroup -> XCD
// assignment really `id % 8`, also with a second queue busy.
//   hipcc -O3 --offload-arch=gfx950 tools/atomic_probe.hip -o atomic_probe && ./atomic_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f4v __attribute__((ext_vector_type(4)));
constexpr int TILE = 128 * 160;        // values of one output tile
constexpr int PER = TILE / 256;        // per thread

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15; }   // HW_REG_XCC_ID[3:0]

__device__ __forceinline__ long long to_fix(float x) {          // round(x * 2^30), |x| < 2^31
    const float hi = truncf(x);
    const float fr = x - hi;                                     // exact
    return ((long long)(int)hi << 30) + (long long)(int)rintf(fr * 1073741824.f);
}
__device__ __forceinline__ float from_fix(long long v) { return (float)((double)v * (1.0 / 1073741824.0)); }

// MODE 0: L2-scope atomics (splits of a tile share an XCD), 1: agent-scope atomics, 2: fp32 slab with plain stores (no reduction at all)
template <int MODE>
__global__ __launch_bounds__(256) void acc_kernel(long long* tiles, float* slabs, unsigned* counters, float* out, int ntiles, int sk, int* xcc, int* err) {
    const int wg = blockIdx.x, tid = threadIdx.x;
    const int x = wg & 7, j = wg >> 3;
    const int tile = (j / sk) * 8 + x, s = j % sk;
    if (tile >= ntiles) return;
    if (tid == 0 && xcc) xcc[wg] = xcc_id();
    float v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) v[i] = (float)((tid * 131 + i * 17 + s * 7) % 1000) * 0.001f - 0.3f + (float)tile;
    if (MODE == 2 || MODE == 3) {
        float* sl = slabs + ((size_t)s * ntiles + tile) * TILE;
#pragma unroll
        for (int i = 0; i < PER; i += 4) *reinterpret_cast<float4*>(sl + (i / 4) * 1024 + tid * 4) = make_float4(v[i], v[i + 1], v[i + 2], v[i + 3]);
        if (MODE == 2) return;
        // MODE 3: the last split to arrive (all on one XCD: plain stores are in that XCD's L2 once vmcnt drains) sums the slabs in split order
        __shared__ unsigned last3;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(counters + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            last3 = old == (unsigned)sk - 1;
            if (last3) __hip_atomic_store(counters + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __syncthreads();
        if (!last3) return;
        float r[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) r[i] = 0.f;
        for (int k = 0; k < sk; ++k) {
            if (k == s) {
#pragma unroll
                for (int i = 0; i < PER; ++i) r[i] += v[i];
                continue;
            }
            const float* q = slabs + ((size_t)k * ntiles + tile) * TILE;
            f4v w[PER / 4];
#pragma unroll
            for (int i = 0; i < PER / 4; ++i)                     // all 20 loads of a slab in flight, one wait
                asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(w[i]) : "v"(q + i * 1024 + tid * 4) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < PER / 4; ++i) {
                asm volatile("" : "+v"(w[i]));
                r[4 * i] += w[i].x; r[4 * i + 1] += w[i].y; r[4 * i + 2] += w[i].z; r[4 * i + 3] += w[i].w;
            }
        }
        float* o3 = out + (size_t)tile * TILE;
#pragma unroll
        for (int i = 0; i < PER; i += 4) *reinterpret_cast<float4*>(o3 + (i / 4) * 1024 + tid * 4) = make_float4(r[i], r[i + 1], r[i + 2], r[i + 3]);
        return;
    }
    long long* t = tiles + (size_t)tile * TILE;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        if (MODE == 0) __hip_atomic_fetch_add(t + i * 256 + tid, to_fix(v[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_add(t + i * 256 + tid, to_fix(v[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __shared__ unsigned last;
    if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (tid == 0) {
        const unsigned old = MODE == 0 ? __hip_atomic_fetch_add(counters + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                                       : __hip_atomic_fetch_add(counters + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = old == (unsigned)sk - 1;
        if (last) {
            if (MODE == 0) __hip_atomic_store(counters + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_store(counters + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (!last) return;
    if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    float* o = out + (size_t)tile * TILE;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        long long r;
        if (MODE == 0) r = __hip_atomic_exchange(t + i * 256 + tid, 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else r = __hip_atomic_exchange(t + i * 256 + tid, 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        o[i * 256 + tid] = from_fix(r);
    }
    (void)err;
}

// what a separate reducer launch does: sum sk slabs, write the tile
__global__ __launch_bounds__(256) void reduce_kernel(const float* slabs, float* out, int ntiles, int sk) {
    const size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= (size_t)ntiles * TILE) return;
    float4 r = make_float4(0, 0, 0, 0);
    for (int k = 0; k < sk; ++k) { const float4 w = *reinterpret_cast<const float4*>(slabs + (size_t)k * ntiles * TILE + e); r.x += w.x; r.y += w.y; r.z += w.z; r.w += w.w; }
    *reinterpret_cast<float4*>(out + e) = r;
}

__global__ void busy_kernel(float* p, int n) {
    float a = p[threadIdx.x];
    for (int i = 0; i < n; ++i) a = a * 1.0001f + 0.5f;
    p[threadIdx.x + blockIdx.x * blockDim.x] = a;
}

static float run_two(float* slabs, float* out, int ntiles, int sk, hipStream_t st, int iters);
template <int MODE>
static float run(long long* tiles, float* slabs, unsigned* cnt, float* out, int ntiles, int sk, int* xcc, hipStream_t st, int iters) {
    const int per_x = (ntiles + 7) / 8 * sk;
    const int grid = per_x * 8;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(acc_kernel<MODE>, dim3(grid), dim3(256), 0, st, tiles, slabs, cnt, out, ntiles, sk, xcc, nullptr);
    CHECK(hipEventRecord(a, st));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(acc_kernel<MODE>, dim3(grid), dim3(256), 0, st, tiles, slabs, cnt, out, ntiles, sk, xcc, nullptr);
    CHECK(hipEventRecord(b, st));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3f / iters;
}

static float run_two(float* slabs, float* out, int ntiles, int sk, hipStream_t st, int iters) {
    const int grid = (ntiles + 7) / 8 * sk * 8, rgrid = (ntiles * TILE / 4 + 255) / 256;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    CHECK(hipEventRecord(a, st));
    for (int i = 0; i < iters; ++i) {
        hipLaunchKernelGGL(acc_kernel<2>, dim3(grid), dim3(256), 0, st, nullptr, slabs, nullptr, out, ntiles, sk, nullptr, nullptr);
        hipLaunchKernelGGL(reduce_kernel, dim3(rgrid), dim3(256), 0, st, slabs, out, ntiles, sk);
    }
    CHECK(hipEventRecord(b, st));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3f / iters;
}

static double expect(int tile, int sk, int tid, int i) {
    double s = 0;
    for (int k = 0; k < sk; ++k) s += (double)((float)((tid * 131 + i * 17 + k * 7) % 1000) * 0.001f - 0.3f + (float)tile);
    return s;
}

int main() {
    const int MAXT = 64, MAXSK = 16;
    long long* tiles; float* slabs; unsigned* cnt; float* out; int* xcc; float* busy;
    CHECK(hipMalloc(&tiles, (size_t)MAXT * TILE * 8)); CHECK(hipMemset(tiles, 0, (size_t)MAXT * TILE * 8));
    CHECK(hipMalloc(&slabs, (size_t)MAXT * MAXSK * TILE * 4));
    CHECK(hipMalloc(&cnt, MAXT * 4)); CHECK(hipMemset(cnt, 0, MAXT * 4));
    CHECK(hipMalloc(&out, (size_t)MAXT * TILE * 4));
    CHECK(hipMalloc(&xcc, 4096 * 4));
    CHECK(hipMalloc(&busy, 1024 * 256 * 4)); CHECK(hipMemset(busy, 0, 1024 * 256 * 4));
    hipStream_t st, st2;
    CHECK(hipStreamCreate(&st)); CHECK(hipStreamCreate(&st2));
    std::vector<float> h((size_t)MAXT * TILE);
    std::vector<int> hx(4096);
    const int cases[][2] = {{16, 5}, {16, 10}, {16, 16}, {64, 3}, {64, 4}, {32, 8}};
    for (int with_busy = 0; with_busy < 2; ++with_busy) {
        for (auto& c : cases) {
            const int nt = c[0], sk = c[1];
            if (with_busy) hipLaunchKernelGGL(busy_kernel, dim3(128), dim3(256), 0, st2, busy, 40000000);
            CHECK(hipMemsetAsync(xcc, 0xff, 4096 * 4, st));
            const float t0 = run<0>(tiles, slabs, cnt, out, nt, sk, xcc, st, 50);
            CHECK(hipStreamSynchronize(st));
            CHECK(hipMemcpy(h.data(), out, (size_t)nt * TILE * 4, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(hx.data(), xcc, 4096 * 4, hipMemcpyDeviceToHost));
            int bad0 = 0, badx = 0;
            for (int t = 0; t < nt; ++t) for (int e = 0; e < TILE; e += 97) { const int i = e / 256, tid = e % 256; if (fabs(h[(size_t)t * TILE + e] - expect(t, sk, tid, i)) > 1e-4 * (1 + t) * sk) ++bad0; }
            const int grid = (nt + 7) / 8 * sk * 8;
            for (int w = 0; w < grid; ++w) if (hx[w] != hx[w & 7]) ++badx;
            CHECK(hipMemset(out, 0, (size_t)nt * TILE * 4));
            const float t1 = run<1>(tiles, slabs, cnt, out, nt, sk, nullptr, st, 50);
            CHECK(hipStreamSynchronize(st));
            CHECK(hipMemcpy(h.data(), out, (size_t)nt * TILE * 4, hipMemcpyDeviceToHost));
            int bad1 = 0;
            for (int t = 0; t < nt; ++t) for (int e = 0; e < TILE; e += 97) { const int i = e / 256, tid = e % 256; if (fabs(h[(size_t)t * TILE + e] - expect(t, sk, tid, i)) > 1e-4 * (1 + t) * sk) ++bad1; }
            const float t2 = run<2>(tiles, slabs, cnt, out, nt, sk, nullptr, st, 50);
            CHECK(hipStreamSynchronize(st));
            CHECK(hipMemset(out, 0, (size_t)nt * TILE * 4));
            const float t3 = run<3>(tiles, slabs, cnt, out, nt, sk, nullptr, st, 50);
            CHECK(hipStreamSynchronize(st));
            CHECK(hipMemcpy(h.data(), out, (size_t)nt * TILE * 4, hipMemcpyDeviceToHost));
            int bad3 = 0;
            for (int t = 0; t < nt; ++t) for (int e = 0; e < TILE; e += 97) { const int q4 = e / 1024, tid = (e % 1024) / 4, i = q4 * 4 + e % 4; if (fabs(h[(size_t)t * TILE + e] - expect(t, sk, tid, i)) > 1e-4 * (1 + t) * sk) ++bad3; }
            const float t4 = run_two(slabs, out, nt, sk, st, 50);
            CHECK(hipStreamSynchronize(st));
            printf("   slabs + last arriver on the XCD %7.2f us (wrong %d) | slab launch + reducer launch %7.2f us\n", t3, bad3, t4);
            if (with_busy) CHECK(hipStreamSynchronize(st2));
            printf("busy=%d tiles=%2d sk=%2d grid=%4d | L2-scope atomics %7.2f us (wrong %d) | agent-scope %7.2f us (wrong %d) | fp32 slab stores %7.2f us | xcd of wg 0..7: %d %d %d %d %d %d %d %d, wg with xcd != xcd[id%%8]: %d\n",
                   with_busy, nt, sk, grid, t0, bad0, t1, bad1, t2, hx[0], hx[1], hx[2], hx[3], hx[4], hx[5], hx[6], hx[7], badx);
            fflush(stdout);
        }
    }
    return 0;
}
