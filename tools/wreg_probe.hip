// Premise check for a "weights straight to VGPRs" form of the halo convolution (csrc/conv_halo.hip streams its weights by LDS-DMA and
// runs its tap loop at ~1500 cycles per tap against 640 cycles of MFMA: DESIGN 3.5).  This probe runs ONLY the proposed main loop:
//   workgroup = 128 pixels x 160 output channels, 10 MFMA waves = 5 column pairs x 2 K halves of each 64-channel chunk (+ 2 idle waves
//   standing in for the staging helpers), A fragments from an LDS-resident halo image (10 ds_read_b128 per kx serve 3 ky x 8 row tiles),
//   B fragments from a per-wave CONTIGUOUS fragment stream in global memory through an 18-slot register ring (two kx groups ahead),
//   one barrier per chunk.
// It reports cycles per tap for (a) every workgroup column reading the same weights (the 64 x 128 level: L2-hot) and (b) weights that
// are unique per workgroup column and cold (the low-resolution levels).
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form tools/wreg_probe.hip -o wreg_probe && ./wreg_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 h16;
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int HSTR = 18;
constexpr int HALO_BYTES = 24 * 1024;

// Balanced form: 8 waves, the 10 column tiles of the 160-column block split 3 | 3 | 2 | 2 over four column groups, two K halves each;
// wave w sits on SIMD w % 4, so SIMD s carries waves s and s + 4: groups are assigned so that every SIMD owns 5 tile columns.
//   wave:        0  1  2  3  4  5  6  7
//   tiles:       3  3  2  2  2  2  3  3
template <int NT, int MODE>
__device__ __forceinline__ void wave_loop(const char* smem, const h16* wp, int nch, int kg, int lane, float& sum) {
    int a_off[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int hx = (lane & 15) + kx;
        const int ch = 4 * kg + (lane >> 4);
        a_off[kx] = hx * 128 + ((ch ^ (((hx >> 1) & 3) << 1)) << 4);
    }
    f32x4v acc[8][NT];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[i][t] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    constexpr int G = 3 * NT;                                // fragments per kx group
    h16x8 ring[3 * G];
#pragma unroll
    for (int f = 0; f < 3 * G; ++f) ring[f] = *reinterpret_cast<const h16x8*>(wp + f * 512);
    for (int cl = 0; cl < nch; ++cl) {
        const char* hb = smem + (cl & 1) * HALO_BYTES;
        const h16* wnext = wp + (long long)(cl + 1) * 3 * G * 512;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            h16x8 a[3];
            a[0] = *reinterpret_cast<const h16x8*>(hb + a_off[kx]);
            a[1] = *reinterpret_cast<const h16x8*>(hb + a_off[kx] + HSTR * 128);
#pragma unroll
            for (int r = 0; r < 10; ++r) {
                if (MODE != 2 && r + 2 < 10) a[(r + 2) % 3] = *reinterpret_cast<const h16x8*>(hb + a_off[kx] + (r + 2) * HSTR * 128);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int i = r - ky;
                    if (i < 0 || i >= 8) continue;
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[MODE == 2 ? r % 2 : r % 3], ring[(kx * 3 + ky) * NT + t], acc[i][t], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE != 1 && cl + 1 < nch) {
#pragma unroll
                for (int f = 0; f < G; ++f) ring[kx * G + f] = *reinterpret_cast<const h16x8*>(wnext + (kx * G + f) * 512);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_barrier();
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t) sum += acc[i][t][0] + acc[i][t][1] + acc[i][t][2] + acc[i][t][3];
}

template <int MODE>   // 0: full loop, 1: no B loads (ring loaded once), 2: no A reads (fragments read once)
__global__ __launch_bounds__(512) void loop_kernel(const h16* __restrict__ wfrag, long long wave_stride, long long col_stride, int nch,
                                                   float* __restrict__ out, unsigned long long* __restrict__ stamps, int ntiles, int nsplit) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * HALO_BYTES / 4; i += 512) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u + (i & 255);
    __syncthreads();
    const int kg = wave & 1;                                  // (SIMD pairs 0/1 and 2/3 then hold the two K halves of a column group)
    const int grp = ((wave >> 1) & 1) | ((wave >> 2) << 1);   // column group 0..3 <- waves {0,1} {2,3} {4,5} {6,7}
    const bool three = grp == 0 || grp == 3;
    // per-wave stream = its tiles x 2 KiB... streams are laid out in units of one tile column: [tile column 0..9][K half]
    const int tile0 = grp == 0 ? 0 : grp == 1 ? 3 : grp == 2 ? 5 : 7;
    const int ntile = blockIdx.x % ntiles, split = (blockIdx.x / ntiles) % nsplit;
    const h16* wp = wfrag + (long long)ntile * col_stride + (long long)(tile0 * 2 + kg) * (wave_stride / 2) + (long long)split * nch * 9 * (three ? 3 : 2) * 512 + lane * 8;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    if (three) wave_loop<3, MODE>(smem, wp, nch, kg, lane, s);
    else wave_loop<2, MODE>(smem, wp, nch, kg, lane, s);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) stamps[blockIdx.x * 10 + wave] = t1 - t0;
    out[blockIdx.x * 512 + tid] = s;
}

template <int MODE>
static void run(const char* label, const h16* w, int ntiles, int nsplit, int nch, int grid, float* out,
                unsigned long long* stamps, void* thrash, size_t thrash_bytes) {
    // the weight set of one layer, laid out as the real kernel would: [column tile][5 column pairs x 2 K halves][all chunks][18 KiB]
    const long long wave_stride = (long long)nsplit * nch * 18 * 512, col_stride = wave_stride * 10;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&loop_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * HALO_BYTES);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    float best = 1e30f;
    double cyc = 0;
    for (int rep = 0; rep < 6; ++rep) {
        if (thrash) hipMemsetAsync(thrash, rep, thrash_bytes, 0);
        hipEventRecord(a, 0);
        hipLaunchKernelGGL((loop_kernel<MODE>), dim3(grid), dim3(512), 2 * HALO_BYTES, 0, w, wave_stride, col_stride, nch, out, stamps, ntiles, nsplit);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (rep > 0 && ms < best) {
            best = ms;
            std::vector<unsigned long long> h(grid * 10);
            hipMemset(stamps, 0, 0);
            hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
            cyc = 0;
            for (int g2 = 0; g2 < grid; ++g2)
                for (int w2 = 0; w2 < 8; ++w2) cyc += (double)h[g2 * 10 + w2];
            cyc /= grid * 8;
        }
    }
    // s_memtime ticks at 100 MHz on this part: report both ticks and the event time per tap
    printf("%-66s grid %4d  chunks %3d  weights %5.1f MB  kernel %7.1f us  = %6.1f ns per tap   (ticks per tap %.0f)\n", label, grid, nch,
           (double)col_stride * ntiles * 2 / 1e6, best * 1e3, best * 1e6 / (nch * 9), cyc / (nch * 9));
}

int main() {
    h16* w;
    const size_t wbytes = 64u << 20;
    hipMalloc(&w, wbytes);
    hipMemset(w, 0, wbytes);
    float* out;
    hipMalloc(&out, 512 * 768 * sizeof(float));
    unsigned long long* stamps;
    hipMalloc(&stamps, 512 * 10 * 8);
    void* thrash;
    const size_t tb = 640u << 20;
    hipMalloc(&thrash, tb);
    //                                                                         ntiles nsplit chunks grid
    run<0>("64x128, 320->320, batch 2 (UNet L0): hot", w, 2, 1, 5, 256, out, stamps, nullptr, 0);
    run<0>("64x128, 320->320, batch 2 (UNet L0): cold", w, 2, 1, 5, 256, out, stamps, thrash, tb);
    run<1>("  ... without B loads", w, 2, 1, 5, 256, out, stamps, nullptr, 0);
    run<2>("  ... without A reads", w, 2, 1, 5, 256, out, stamps, nullptr, 0);
    run<0>("64x128, 960->320 (up block conv1): cold", w, 2, 1, 15, 256, out, stamps, thrash, tb);
    run<0>("64x128, 320->320, batch 1 (BlobNet L0): cold, split 2", w, 2, 2, 3, 256, out, stamps, thrash, tb);
    run<0>("64x128, 320->320, batch 1 (BlobNet L0): cold, unsplit", w, 2, 1, 5, 128, out, stamps, thrash, tb);
    run<0>("32x64, 640->640, batch 2 (UNet L1): cold, split 2", w, 4, 2, 5, 256, out, stamps, thrash, tb);
    run<0>("32x64, 640->640, batch 2 (UNet L1): cold, unsplit", w, 4, 1, 10, 128, out, stamps, thrash, tb);
    run<0>("16x32, 1280->1280, batch 2 (UNet L2): cold, split 4", w, 8, 4, 5, 256, out, stamps, thrash, tb);
    run<0>("16x32, 1280->1280, batch 2 (UNet L2): cold, split 2", w, 8, 2, 10, 128, out, stamps, thrash, tb);
    run<0>("16x32, 1280->1280, batch 1 (BlobNet L2): cold, split 7", w, 8, 7, 3, 224, out, stamps, thrash, tb);
    run<0>("8x16, 1280->1280, batch 2 (UNet L3): cold, split 10", w, 8, 10, 2, 160, out, stamps, thrash, tb);
    run<0>("8x16, 1280->1280, batch 2 (UNet L3): cold, split 5", w, 8, 5, 4, 80, out, stamps, thrash, tb);
    run<0>("8x16, 1280->1280, batch 1 (BlobNet L3): cold, split 10", w, 8, 10, 2, 80, out, stamps, thrash, tb);
    return 0;
}
