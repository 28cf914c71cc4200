// Stand-alone bring-up of the 8-wave, 8-phase 256 x 256 GEMM main loop that csrc/gemm256.hip ships ("a large-M GEMM
// at vendor rate inside the plan").  C[M][N] = A[M][K] . B[N][K]^T, both operands K-contiguous fp16, fp32 accumulation - the layout of
// every Linear / 1 x 1 convolution of diffusers/src/diffusers/models/attention.py:1161-1167, attention_processor.py:2191-2224,
// transformers/transformer_2d.py:479-527 as the plan runs them.
// Structure (cdna_hip_programming.md section 5 "The 256^2 8-phase template", rebuilt from its description - the example file is not in
// this image):
//   * 8 waves = 2 (M) x 4 (N), a wave owns 128 x 64 of the tile as 8 x 4 accumulator tiles of v_mfma_f32_16x16x32_f16 (128 registers);
//   * operands global -> LDS by LDS-DMA (global_load_lds_dwordx4), in HALF-TILES of 16 KB (B0, A0, B1, A1 per 64-k tile; "0" / "1" = the
//     part a wave needs in phases {0, 1} / {2, 3}), each wave 2 instructions per half-tile, one half-tile per phase, 7 half-tiles ahead;
//     counted s_waitcnt vmcnt(6) once per k-tile, raw s_barrier: 3..7 half-tiles in flight across the barriers;
//   * the LDS image is made of 1-KiB sub-tiles [16 rows][32 k] = one MFMA fragment, rows of 64 B with the 16-byte chunk index XOR-ed by
//     2 for rows >= 8 (conflict-free ds_read_b128 over the 4 x 16-lane service groups); the swizzle sits on the per-lane SOURCE address;
//   * a phase = {fragment reads + 2 LDS-DMA issues | s_barrier | 16 MFMAs (one 64 x 32 quadrant over the tile's 64 k) | s_barrier};
//     the two wave groups (M halves; one wave of each per SIMD) run one barrier apart: one group's MFMAs beside the other's loads.
//   hipcc -O3 --offload-arch=gfx950 tools/gemm8p_probe.hip -o gemm8p_probe && ./gemm8p_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <type_traits>
#include <vector>

typedef _Float16 h16;
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __forceinline__ void glds16(const void* src, char* lds_dst) { __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_dst, 16, 0, 0); }

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HT = 16384;              // half-tile bytes
constexpr int BUF = 4 * HT;            // one k-tile: [B0][A0][B1][A1]
constexpr int LDS_BYTES = 2 * BUF;     // 128 KiB

#define SB() __builtin_amdgcn_sched_barrier(0)
#define BAR() do { SB(); __builtin_amdgcn_s_barrier(); SB(); } while (0)

struct Frags {
    h16x8 a[4][2];          // current A sub-block: 4 row tiles x 2 k-steps
    h16x8 b0[2][2], b1[2][2];
};

// MODE 0: steady state (every phase stages a half-tile); 1: first tile of the draining pair (only phase 0 stages; phase 3 waits for
// everything); 2: last tile (nothing staged, nothing waited for)
template <int B, int MODE>
__device__ __forceinline__ void tile_phases(f32x4 (&acc)[8][4], Frags& f, const char* aB, const char* bB, char* stg, const h16* pB0, const h16* pA0,
                                            const h16* pB1, const h16* pA1, size_t k1, size_t k2) {
    constexpr int O = B * BUF, ON = (B ^ 1) * BUF;
    auto mfma = [&](auto ac, auto bc) __attribute__((always_inline)) {
        constexpr int a = decltype(ac)::value, b = decltype(bc)::value;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SB();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    acc[a * 4 + rt][b * 2 + ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b ? f.b1[ct][ks] : f.b0[ct][ks], f.a[rt][ks], acc[a * 4 + rt][b * 2 + ct], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    // ---- phase 0: B0 + A0 fragments; stage A1 of tile t + 1
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) f.b0[ct][ks] = *reinterpret_cast<const h16x8*>(bB + O + 0 * HT + ct * 2048 + ks * 1024);
    SB();
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) f.a[rt][ks] = *reinterpret_cast<const h16x8*>(aB + O + 1 * HT + rt * 2048 + ks * 1024);
    SB();
    if (MODE <= 1) {
        glds16(pA1 + k1, stg + ON + 3 * HT);
        glds16(pA1 + k1 + 32, stg + ON + 3 * HT + 1024);
    }
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");      // the B0 reads have returned: B0's slot may be restaged one phase later
    BAR();
    mfma(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    BAR();
    // ---- phase 1: B1 fragments; stage B0 of tile t + 2
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) f.b1[ct][ks] = *reinterpret_cast<const h16x8*>(bB + O + 2 * HT + ct * 2048 + ks * 1024);
    SB();
    if (MODE == 0) {
        glds16(pB0 + k2, stg + O + 0 * HT);
        glds16(pB0 + k2 + 32, stg + O + 0 * HT + 1024);
    }
    BAR();
    mfma(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
    BAR();
    // ---- phase 2: A1 fragments; stage A0 of tile t + 2
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) f.a[rt][ks] = *reinterpret_cast<const h16x8*>(aB + O + 3 * HT + rt * 2048 + ks * 1024);
    SB();
    if (MODE == 0) {
        glds16(pA0 + k2, stg + O + 1 * HT);
        glds16(pA0 + k2 + 32, stg + O + 1 * HT + 1024);
    }
    BAR();
    mfma(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
    BAR();
    // ---- phase 3: no reads; stage B1 of tile t + 2; the wait that retires tile t + 1
    if (MODE == 0) {
        glds16(pB1 + k2, stg + O + 2 * HT);
        glds16(pB1 + k2 + 32, stg + O + 2 * HT + 1024);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else if (MODE == 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    BAR();
    mfma(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
    BAR();
}

template <int VAR>
__global__ __launch_bounds__(512, 2) void gemm8p_kernel(const h16* __restrict__ A, const h16* __restrict__ Bm, h16* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int tiles_m = M / BM, tiles_n = N / BN, ntiles = tiles_m * tiles_n;
    int id = blockIdx.x;
    if (ntiles % 8 == 0) id = (id & 7) * (ntiles >> 3) + (id >> 3);        // XCD x (= blockIdx % 8) owns a contiguous run of tile ids
    constexpr int GM = 4;
    int tm, tn;
    if (tiles_m % GM == 0) {
        const int grp = id / (GM * tiles_n), r = id - grp * GM * tiles_n;
        tm = grp * GM + (r % GM);
        tn = r / GM;
    } else {
        tm = id / tiles_n;
        tn = id - tm * tiles_n;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    // staging: this wave fills sub-tiles 2 wave, 2 wave + 1 (the two k-steps of one 16-row group) of every half-tile; lane -> (row, chunk)
    // of the sub-tile slot it lands in (lane-linear), source chunk with the swizzle applied
    const int srow = lane >> 2;
    const int skq = (lane & 3) ^ (((lane >> 5) & 1) << 1);
    const h16* pB0 = Bm + (size_t)(n0 + (wave >> 1) * 64 + (wave & 1) * 16 + srow) * K + skq * 8;
    const h16* pB1 = pB0 + (size_t)32 * K;
    const h16* pA0 = A + (size_t)(m0 + (wave >> 2) * 128 + (wave & 3) * 16 + srow) * K + skq * 8;
    const h16* pA1 = pA0 + (size_t)64 * K;
    char* stg = smem + wave * 2048;
    // fragment reads: lane (r = lane & 15, kq = lane >> 4) reads chunk kq ^ (2 if r >= 8) of row r
    const int foff = (lane & 15) * 64 + ((((lane >> 4) ^ (((lane >> 3) & 1) << 1))) << 4);
    const char* aB = smem + wr * 8192 + foff;
    const char* bB = smem + wc * 4096 + foff;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    Frags f;

    const int T = K / BK;                       // even, >= 2
    // prologue: half-tiles 0..6 = tile 0 (B0 A0 B1 A1), tile 1 (B0 A0 B1)
    glds16(pB0, stg + 0 * HT);            glds16(pB0 + 32, stg + 0 * HT + 1024);
    glds16(pA0, stg + 1 * HT);            glds16(pA0 + 32, stg + 1 * HT + 1024);
    glds16(pB1, stg + 2 * HT);            glds16(pB1 + 32, stg + 2 * HT + 1024);
    glds16(pA1, stg + 3 * HT);            glds16(pA1 + 32, stg + 3 * HT + 1024);
    glds16(pB0 + 64, stg + BUF + 0 * HT); glds16(pB0 + 96, stg + BUF + 0 * HT + 1024);
    glds16(pA0 + 64, stg + BUF + 1 * HT); glds16(pA0 + 96, stg + BUF + 1 * HT + 1024);
    glds16(pB1 + 64, stg + BUF + 2 * HT); glds16(pB1 + 96, stg + BUF + 2 * HT + 1024);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    BAR();
    if (wr == 1) BAR();                         // the second wave group runs one barrier behind the first
    size_t k = 0;                               // element offset of tile t
    for (int it = 0; it < T / 2 - 1; ++it) {
        tile_phases<0, 0>(acc, f, aB, bB, stg, pB0, pA0, pB1, pA1, k + 64, k + 128);
        tile_phases<1, 0>(acc, f, aB, bB, stg, pB0, pA0, pB1, pA1, k + 128, k + 192);
        k += 128;
    }
    tile_phases<0, 1>(acc, f, aB, bB, stg, pB0, pA0, pB1, pA1, k + 64, 0);
    tile_phases<1, 2>(acc, f, aB, bB, stg, pB0, pA0, pB1, pA1, 0, 0);
    if (wr == 0) BAR();

    // D^T tile (i = 4 a + rt, j = 2 b + ct): rows m = 128 wr + 64 a + 16 rt + (lane & 15); columns n = 64 wc + 32 b + 16 ct + 4 (lane >> 4) + e
    const int mrow = m0 + wr * 128 + (lane & 15);
    const int ncol = n0 + wc * 64 + 4 * (lane >> 4);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        h16* crow = C + (size_t)(mrow + (i >> 2) * 64 + (i & 3) * 16) * N + ncol;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (h16)acc[i][j][e];
            *reinterpret_cast<h16x4*>(crow + (j >> 1) * 32 + (j & 1) * 16) = o;
        }
    }
}

__global__ void ref_kernel(const h16* A, const h16* B, float* out, int K, const int* rows, const int* cols, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const h16* a = A + (size_t)rows[i] * K;
    const h16* b = B + (size_t)cols[i] * K;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += (float)a[k] * (float)b[k];
    out[i] = s;
}

static void fill(std::vector<h16>& v, unsigned seed, float scale) {
    unsigned s = seed;
    for (auto& x : v) {                            // sum of four uniforms: near-normal, random mantissas (the power the data costs is real)
        float t = 0.f;
        for (int j = 0; j < 4; ++j) { s = s * 1664525u + 1013904223u; t += (float)(s >> 8) / 16777216.f - 0.5f; }
        x = (h16)(t * scale);
    }
}

template <int VAR>
static double run(int M, int N, int K, int reps, bool full_check) {
    std::vector<h16> ha((size_t)M * K), hb((size_t)N * K);
    fill(ha, 1u + M, 1.7f);
    fill(hb, 7u + N, 0.05f);
    h16 *dA, *dB, *dC;
    CHECK(hipMalloc(&dA, ha.size() * 2)); CHECK(hipMalloc(&dB, hb.size() * 2)); CHECK(hipMalloc(&dC, (size_t)M * N * 2));
    CHECK(hipMemcpy(dA, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dB, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemset(dC, 0, (size_t)M * N * 2));
    auto kern = &gemm8p_kernel<VAR>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    const int grid = (M / BM) * (N / BN);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS_BYTES, 0, dA, dB, dC, M, N, K);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS_BYTES, 0, dA, dB, dC, M, N, K);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    // correctness: sampled entries (or, full_check, one entry of EVERY 16 x 16 output tile) against an fp32 dot product of the same operands
    std::vector<int> hr, hc;
    unsigned s = 99;
    if (full_check) {
        for (int i = 0; i < M; i += 16)
            for (int j = 0; j < N; j += 16) { s = s * 1664525u + 1013904223u; hr.push_back(i + ((s >> 8) & 15)); hc.push_back(j + ((s >> 12) & 15)); }
    } else {
        for (int i = 0; i < 4096; ++i) { s = s * 1664525u + 1013904223u; hr.push_back((s >> 8) % M); s = s * 1664525u + 1013904223u; hc.push_back((s >> 8) % N); }
    }
    const int n = (int)hr.size();
    int *dr, *dc; float* dref;
    CHECK(hipMalloc(&dr, n * 4)); CHECK(hipMalloc(&dc, n * 4)); CHECK(hipMalloc(&dref, n * 4));
    CHECK(hipMemcpy(dr, hr.data(), n * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dc, hc.data(), n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(ref_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, dA, dB, dref, K, dr, dc, n);
    std::vector<float> href(n);
    std::vector<h16> hC((size_t)M * N);
    CHECK(hipMemcpy(href.data(), dref, n * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost));
    double worst = 0.0, scale = 0.0;
    int bad = 0;
    for (int i = 0; i < n; ++i) scale = fmax(scale, fabs(href[i]));
    for (int i = 0; i < n; ++i) {
        const double e = fabs((double)(float)hC[(size_t)hr[i] * N + hc[i]] - href[i]);
        worst = fmax(worst, e);
        if (e > 2e-3 * scale + 1e-3) ++bad;
    }
    // race screen: the same launch again must give the same bits
    std::vector<h16> hC2((size_t)M * N);
    int diff = 0;
    for (int rep = 0; rep < (full_check ? 6 : 1); ++rep) {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS_BYTES, 0, dA, dB, dC, M, N, K);
        CHECK(hipMemcpy(hC2.data(), dC, hC2.size() * 2, hipMemcpyDeviceToHost));
        if (memcmp(hC.data(), hC2.data(), hC.size() * 2) != 0) ++diff;
    }
    const double tf = 2.0 * M * N * K / us / 1e6;
    printf("var %d  %6d x %6d x %6d  grid %4d  %9.1f us  %7.1f TFLOP/s  (%.3f of 2.5 PF)  max err %.3e of scale %.3e, %d of %d bad, %d replays differ: %s\n", VAR, M, N, K, grid,
           us, tf, tf / 2500.0, worst, scale, bad, n, diff, bad == 0 && diff == 0 ? "ok" : "WRONG");
    fflush(stdout);
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dr); hipFree(dc); hipFree(dref);
    return tf;
}

int main(int argc, char** argv) {
    run<0>(512, 512, 512, 5, true);
    run<0>(1024, 768, 128, 5, true);
    run<0>(2048, 1280, 1280, 5, true);
    const int shapes[][3] = {{4096, 4096, 4096}, {8192, 8192, 8192}, {8192, 10240, 1280}, {8192, 1280, 5120}, {8192, 1280, 1280}, {8192, 3840, 1280},
                             {4096, 10240, 1280}, {4096, 1280, 1280}, {16384, 5120, 640}, {9216, 10240, 1280}, {2048, 10240, 1280}, {2048, 1280, 5120}, {8192, 1280, 6400}};
    for (auto& sh : shapes) run<0>(sh[0], sh[1], sh[2], 20, false);
    return 0;
}
