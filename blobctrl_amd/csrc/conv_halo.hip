// 3x3 / stride-1 / pad-1 convolution with an LDS-RESIDENT INPUT HALO TILE and the GroupNorm(+SiLU) of its input fused into the
// staging path (gfx950, v_mfma_f32_16x16x32_f16).
//
// Replaces, for the ResBlock convolutions of the hot path (D/models/resnet.py:327-341 norm1 -> nonlinearity -> conv1 and
// :351-366 norm2 -> nonlinearity -> conv2, with the torch.cat([h, skip], 1) of unet_2d_blocks.py:2559,2719 in front of norm1):
// the separate GroupNorm-apply pass over the activation AND the implicit-GEMM kernel that re-gathered every input pixel nine times.
//
// Work decomposition
//   workgroup = 8 x 16 output pixels (BM = 128 rows) x BN = 160 output channels, 8 waves.
//   K loop    = input channels in chunks of 64; per chunk the (8+2) x (16+2) pixel halo of the input is brought into LDS ONCE
//               (LDS-DMA of the raw fp16 rows straight into their final, swizzled slots of the operand image - the swizzle is
//               applied to the SOURCE address - then an in-place register pass  y = silu?(a[b][c] * x + b[b][c]); out-of-image
//               pixels are written as zeros AFTER the activation = the convolution's zero padding) and serves all nine taps;
//               only the weights stream per tap (LDS-DMA, NSTG-stage ring = NSTG-1 taps in flight, counted vmcnt, raw s_barrier:
//               the low-resolution levels are bound by the weight bytes a CU keeps in flight).
//               LDS-DMA pieces per MFMA are ~0.45x the implicit-GEMM kernel's, and the 9x re-read of the input from L2 is gone.
//   waves     = 2 (pixel halves) x 2 (channel halves) x 2 (K halves of every 64-channel chunk); wave tile 64 x 80 = 4 x 5 MFMA tiles,
//               9 ds_read_b128 per 20 MFMAs; the two K halves are summed through LDS in the epilogue.
//   split-K   = over channel chunks (grid.z), fp32 slabs + the shared reducer (gemm.hip), for the low-resolution levels.
// Swizzle: 16-byte chunk c of a 128-byte operand row is stored at chunk c ^ (((r >> 1) & 3) << 1) with r = halo x (A image) or
// weight row (B image): with the 16x16x32 fragment map (lane -> row l & 15, chunk l >> 4) every ds_read_b128 lane group then
// touches 16 distinct 16-byte slots of the 256-byte bank row.
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include <type_traits>
#include "gemm_common.h"

using namespace bcg;

namespace {

__device__ __attribute__((aligned(16))) unsigned int g_zero_line_h[4] = {0u, 0u, 0u, 0u};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* src, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_dst, 16, 0, 0);
}

constexpr int TW = 16, TH = 8;
constexpr int HBM = TW * TH;                    // 128 output pixels per workgroup
constexpr int HBN = 160;                        // output channels per workgroup
constexpr int HSTR = TW + 2;                    // halo row stride (pixels)
constexpr int HPIX = (TH + 2) * HSTR;           // 180 halo pixels
constexpr int HALO_BYTES = 24 * 1024;           // one 64-channel chunk of the halo (swizzled operand image): 24 LDS-DMA pieces of 1 KiB
                                                // = 192 pixel slots, the last 12 are padding (the DMA count per wave stays uniform)
constexpr int B_PIECES = HBN / 8;               // 20 x 1 KiB per weight stage (one tap of one chunk)
constexpr int STAGE_BYTES = HBN * 128;
#ifndef BC_HALO_NSTG
#define BC_HALO_NSTG 3
#endif
constexpr int NSTG = BC_HALO_NSTG;              // weight ring depth
constexpr int DEPTH = NSTG - 1;                 // taps issued ahead
constexpr int OFF_HALO = 0;
constexpr int OFF_B = 2 * HALO_BYTES;
constexpr int OFF_AB = OFF_B + NSTG * STAGE_BYTES;
constexpr int MAX_CH = (160 * 1024 - OFF_AB) / 512 < 40 ? (160 * 1024 - OFF_AB) / 512 : 40;   // channel chunks per workgroup (the affine
                                                // table of the chunk range lives in LDS: 64 x (a, b) x 4 bytes per chunk)
constexpr int LDS_TOTAL = OFF_AB + MAX_CH * 64 * 8;
constexpr int TS = HBN + 4;                     // epilogue tile row stride (floats)
constexpr int OFF_SCR = HBM * TS * 4;           // GroupNorm-partial scratch behind the epilogue tile
static_assert(OFF_SCR + 24 * HBN * 2 * 4 <= LDS_TOTAL, "epilogue scratch does not fit");
static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget");
static_assert(MAX_CH >= 20 && HPIX * 128 <= HALO_BYTES, "halo layout");

template <int N>
__device__ __forceinline__ void wait_vm_c() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void wait_vm(int n) {      // n is wave-uniform
    switch (n) {
        case 0: wait_vm_c<0>(); break;
        case 2: wait_vm_c<2>(); break;
        case 1: wait_vm_c<1>(); break;
        case 3: wait_vm_c<3>(); break;
        case 4: wait_vm_c<4>(); break;
        case 5: wait_vm_c<5>(); break;
        case 6: wait_vm_c<6>(); break;
        case 7: wait_vm_c<7>(); break;
        case 8: wait_vm_c<8>(); break;
        case 9: wait_vm_c<9>(); break;
        case 10: wait_vm_c<10>(); break;
        case 11: wait_vm_c<11>(); break;
        case 12: wait_vm_c<12>(); break;
        case 13: wait_vm_c<13>(); break;
        case 14: wait_vm_c<14>(); break;
        default: wait_vm_c<15>(); break;
    }
}

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

constexpr int FIN_MAX_CH = 720;                 // in-kernel GroupNorm finalize: channel span (incl. group straddle) per workgroup

// AFFINE: 0 = plain convolution, 1 = affine table from global memory (bc_gn_finalize ran), 2 = GroupNorm finalize in the prologue
template <int AFFINE>
__global__ __launch_bounds__(512) void conv_halo_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const BcGemm& p = g.p;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave & 1, wm = (wave >> 1) & 1, wn = wave >> 2;

    const int plane = gridDim.x * gridDim.y;
    const int lin3 = bc_xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), plane * gridDim.z);
    const int lin = lin3 % plane;
    const int split = lin3 / plane;
    // an XCD owns a contiguous band of pixel tiles with all their column tiles, or (g.nband: the weights are the larger operand)
    // a band of column tiles with all their pixel tiles - see gemm_fast.hip
    const int tile = g.nband ? lin % (int)gridDim.y : lin / (int)gridDim.x;
    const int ntile = g.nband ? lin / (int)gridDim.y : lin % (int)gridDim.x;
    const int n0 = ntile * HBN;
    const int b = tile / g.halo_tpi;
    const int tin = tile - b * g.halo_tpi;
    const int ty0 = (tin / g.halo_tx) * TH, tx0 = (tin % g.halo_tx) * TW;
    const int H = p.Hin, W = p.Win;

    const int dbg = g.halo_dbg;                              // BC_HALO_DBG ablation bits (diagnostics only; results are wrong when set)
    unsigned long long* const stamps = g.halo_stamps;        // BC_HALO_STAMPS diagnostics (null in production)
    auto stamp = [&](int i) {
        if (stamps && tid == 0) stamps[(size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8 + i] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);
    const int c_begin = split * g.halo_cps;
    const int nch = min(g.halo_nch, c_begin + g.halo_cps) - c_begin;
    const int NT = nch * 9;

    const h16* __restrict__ A1 = reinterpret_cast<const h16*>(p.A);
    const h16* __restrict__ A2 = reinterpret_cast<const h16*>(p.A2);
    const h16* __restrict__ Wt = reinterpret_cast<const h16*>(p.W);
    const h16* zero = reinterpret_cast<const h16*>(g_zero_line_h);

    // ---- this lane's three halo slots: the 16 bytes it brings in by LDS-DMA are the 16 bytes it later transforms ----
    // LDS-DMA lands lane-linearly: slot (tid, q) IS bytes [tid * 16 + 8192 q, +16) of the operand image = halo pixel
    // hp = (tid >> 3) + 64 q, 16-byte slot tid & 7; the swizzle (chunk c lives in slot c ^ swz(hx)) is therefore applied to the
    // source: this lane fetches 8-channel sub-chunk csub = (tid & 7) ^ swz(hx).
    int pix[3];                                               // input pixel index (b, gy, gx) or -1 (outside the image / padding slot)
    int csub[3];                                              // 8-channel sub-chunk of the 64-channel chunk this slot holds
    bool in_halo[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int hp = (tid >> 3) + 64 * q;
        const int hy = hp / HSTR, hx = hp - hy * HSTR;
        const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
        in_halo[q] = hp < HPIX;
        const bool in_img = in_halo[q] && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        pix[q] = in_img ? (b * H + gy) * W + gx : -1;
        csub[q] = (tid & 7) ^ (((hx >> 1) & 3) << 1);
    }
    // ---- weight pieces of this wave (1 KiB = 8 rows x 128 B; waves 0-3 carry three, waves 4-7 two) ----
    const int LB = wave < 4 ? 3 : 2;
    int woff[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int n = (wave + 8 * q) * 8 + (lane >> 3);
        const int cs = (lane & 7) ^ (((n >> 1) & 3) << 1);
        woff[q] = (n0 + n) * p.ldw + cs * 8;
    }
    // ---- fragment read offsets ----
    int a_off[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int hx = (lane & 15) + kx;
        const int ch = 4 * kg + (lane >> 4);
        a_off[kx] = (wm * 4) * HSTR * 128 + hx * 128 + ((ch ^ (((hx >> 1) & 3) << 1)) << 4);
    }
    const int nl = wn * 80 + (lane & 15);
    const int b_off = OFF_B + nl * 128 + (((4 * kg + (lane >> 4)) ^ (((nl >> 1) & 3) << 1)) << 4);

    auto issue_a = [&](int cl) {                              // raw rows of chunk c_begin + cl -> operand image (cl & 1)
        const int k0 = (c_begin + cl) * 64;
        const bool second = A2 != nullptr && k0 >= p.C1;      // wave-uniform (C1 % 64 == 0)
        const h16* src = second ? A2 : A1;
        const long long stride = second ? p.lda2 : p.lda;
        const int kin = second ? k0 - p.C1 : k0;
        char* dst = smem + OFF_HALO + (cl & 1) * HALO_BYTES;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const h16* s = pix[q] >= 0 ? src + (long long)pix[q] * stride + kin + csub[q] * 8 : zero;
            glds16(s, dst + (wave + 8 * q) * 1024);
        }
    };
    auto issue_b = [&](int kt, int stage) {                   // weights of flattened (chunk, tap) index kt -> ring stage
        const int cl = kt / 9, tap = kt - cl * 9;
        const int koff = tap * p.Cin + (c_begin + cl) * 64;
#pragma unroll
        for (int q = 0; q < 3; ++q)
            if (q < LB) glds16(Wt + woff[q] + koff, smem + OFF_B + stage * STAGE_BYTES + (wave + 8 * q) * 1024);
    };
    unsigned slot_addr = (unsigned)(size_t)(lptr_t)(smem) + OFF_HALO + tid * 16;   // LDS byte address of this lane's slot 0, image 0
    unsigned ab_base = (unsigned)(size_t)(lptr_t)(smem) + OFF_AB;
    // In-place pass over this lane's slot q of the image that just landed, in two halves so that the LDS latency hides behind the
    // tap's MFMAs (only with a GroupNorm in front, AFFINE != 0: a plain convolution's rows are final as they land):
    //   tr_issue  (top of the tap, BEFORE the fragment reads so hipcc's own counted lgkmcnt waits stay conservative): the slot + the
    //             8 (a, b) pairs of its channels, no wait;
    //   tr_finish (after the MFMAs): wait, normalise + activate, store the slot back.
    // Inline asm throughout: hipcc orders every plain LDS access it cannot prove disjoint behind ALL LDS-DMA writes in flight
    // (`s_waitcnt vmcnt(0)`: it does not see the counted waits), which would drain the weight ring at every slice.  The lane's own
    // slot IS complete (see the wait accounting in the tap loop).
    struct Pending { u32x4v raw; f32x4v t0, t1, t2, t3; };
    auto tr_issue = [&](auto qc, int cl, Pending& pd) {
        constexpr int q = decltype(qc)::value;
        if (!AFFINE || !in_halo[q]) return;
        const unsigned sa = slot_addr + (cl & 1) * HALO_BYTES;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(pd.raw) : "v"(sa), "n"(8192 * q) : "memory");
        const unsigned ab_addr = ab_base + cl * 512 + csub[q] * 64;     // (64 channels x (a, b) x 4 bytes per chunk)
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\t"
                     "ds_read_b128 %3, %4 offset:48"
                     : "=&v"(pd.t0), "=&v"(pd.t1), "=&v"(pd.t2), "=&v"(pd.t3) : "v"(ab_addr) : "memory");
    };
    auto tr_finish = [&](auto qc, int cl, Pending& pd) {
        constexpr int q = decltype(qc)::value;
        if (!AFFINE || !in_halo[q]) return;
        {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pd.raw), "+v"(pd.t0), "+v"(pd.t1), "+v"(pd.t2), "+v"(pd.t3)::"memory");
            __builtin_amdgcn_sched_barrier(0);
            const float aa[8] = {pd.t0[0], pd.t0[2], pd.t1[0], pd.t1[2], pd.t2[0], pd.t2[2], pd.t3[0], pd.t3[2]};
            const float bb[8] = {pd.t0[1], pd.t0[3], pd.t1[1], pd.t1[3], pd.t2[1], pd.t2[3], pd.t3[1], pd.t3[3]};
            const u32x4v rawv = pd.raw;
            const h16* xin = reinterpret_cast<const h16*>(&rawv);
            u32x4v outraw;
            h16* o = reinterpret_cast<h16*>(&outraw);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = fmaf((float)xin[j], aa[j], bb[j]);
                if (p.a_act == BC_ACT_SILU) v = v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
                o[j] = (h16)v;
            }
            if (pix[q] < 0) outraw = (u32x4v){0u, 0u, 0u, 0u};       // zero padding is applied AFTER norm + activation
            pd.raw = outraw;
        }
        const unsigned sa = slot_addr + (cl & 1) * HALO_BYTES;
        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(sa), "v"(pd.raw), "n"(8192 * q) : "memory");
    };
    using Q0 = std::integral_constant<int, 0>;
    using Q1 = std::integral_constant<int, 1>;
    using Q2 = std::integral_constant<int, 2>;

    // ---- prologue: affine table of this chunk range -> LDS, first halo chunk, first DEPTH weight stages ----
    if (AFFINE == 2) {
        // GroupNorm finalize for the groups overlapping this workgroup's channels [k_lo, k_hi), from the producers' per-channel
        // partials (fixed summation order: bit-reproducible).  Scratch = the two halo images (before any LDS-DMA is issued).
        const int cpg = p.Cin / p.a_groups;
        const int k_lo = c_begin * 64, k_hi = k_lo + nch * 64;
        const int g_lo = k_lo / cpg, g_hi = min(p.a_groups, (k_hi + cpg - 1) / cpg);
        const int c_lo = g_lo * cpg, nc = g_hi * cpg - c_lo;
        double* scr = reinterpret_cast<double*>(smem + OFF_HALO);           // [nc][2] = (sum, sum of squares) per channel
        for (int cc = tid; cc < nc; cc += 512) {
            const int c = c_lo + cc;
            const bool second = p.A2 != nullptr && c >= p.C1;
            const int Cs = second ? p.Cin - p.C1 : (p.A2 ? p.C1 : p.Cin);
            const unsigned long long* t = (second ? p.a_tot2 : p.a_tot1) + ((size_t)b * Cs + (second ? c - p.C1 : c)) * BC_GN_TOT_WORDS;
            double s, q;
            bc_gn_tot_read(t, s, q);
            scr[cc * 2] = s;
            scr[cc * 2 + 1] = q;
        }
        __syncthreads();
        float* stat = reinterpret_cast<float*>(scr + nc * 2);             // [groups][2] = (mean, rstd)
        for (int gi = g_lo + wave; gi < g_hi; gi += 8) {
            double s = 0.0, q = 0.0;
            for (int cj = lane; cj < cpg; cj += 64) {
                s += scr[((gi - g_lo) * cpg + cj) * 2];
                q += scr[((gi - g_lo) * cpg + cj) * 2 + 1];
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                s += __shfl_xor(s, o);
                q += __shfl_xor(q, o);
            }
            const double n = (double)g.div_rpb.d * cpg;
            const double mean = s / n;
            double var = q / n - mean * mean;
            if (var < 0.0) var = 0.0;
            if (lane == 0) {
                stat[(gi - g_lo) * 2] = (float)mean;
                stat[(gi - g_lo) * 2 + 1] = (float)(1.0 / sqrt(var + (double)p.a_eps));
            }
        }
        __syncthreads();
        float* abt = reinterpret_cast<float*>(smem + OFF_AB);
        for (int i = tid; i < nch * 64; i += 512) {
            const int c = k_lo + i;
            const int gi = c / cpg - g_lo;
            const float a = stat[gi * 2 + 1] * p.a_gamma[c];
            abt[i * 2] = a;
            abt[i * 2 + 1] = p.a_beta[c] - stat[gi * 2] * a;
        }
        __syncthreads();                                                  // scratch is free again before the first halo is written
    }
    // (the in-kernel finalize above used the halo images as scratch: the DMAs start after it)
    stamp(1);
    issue_a(0);
    const int npre = DEPTH < NT ? DEPTH : NT;
    for (int i = 0; i < npre; ++i) issue_b(i, i);
    if (AFFINE == 1) {                                        // (its global loads fly together with the DMAs above)
        const float4* src = reinterpret_cast<const float4*>(p.a_affine + ((size_t)b * p.Cin + (size_t)c_begin * 64) * 2);
        float4* dst = reinterpret_cast<float4*>(smem + OFF_AB);
        for (int i = tid; i < nch * 32; i += 512) dst[i] = src[i];
    }
    wait_vm((dbg & (3 | 64)) ? 0 : npre * LB);                // my three halo slots have landed (weights may still fly)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                             // affine table visible
    asm volatile("" ::: "memory");
    stamp(2);
    {
        Pending p0, p1, p2;
        tr_issue(Q0{}, 0, p0);
        tr_issue(Q1{}, 0, p1);
        tr_issue(Q2{}, 0, p2);
        tr_finish(Q0{}, 0, p0);
        tr_finish(Q1{}, 0, p1);
        tr_finish(Q2{}, 0, p2);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    f32x4v acc[4][5];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = (f32x4v){0.f, 0.f, 0.f, 0.f};

    stamp(3);
    // The two waves of a SIMD (w, w + 4) would run every tap in lockstep - both waiting for their fragment reads, then both in
    // their MFMAs - so the matrix pipe idles through every read phase.  Waves 4-7 therefore run half a tap late: between two
    // barriers they first multiply with the fragments they read at the END of the previous interval, then read this tap's; one
    // partner's reads, DMA issue and in-place pass sit under the other's MFMAs (BC_HALO_DBG & 128 turns the stagger off).
    const bool late = wn == 1 && !(dbg & 128);
    h16x8 fa[4], fb[5];
    auto mfma_tap = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    };
    int st = 0;                                               // ring stage of the current tap
    for (int cl = 0; cl < nch; ++cl) {
        const bool has_next = cl + 1 < nch;
        const char* hb = smem + OFF_HALO + (cl & 1) * HALO_BYTES;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kt = cl * 9 + tap;
            // Counted wait: the weights of THIS tap must have landed.  What may stay in flight is whatever this wave issued after
            // them, in order: the taps kt+1 .. kt+DEPTH-1 and - the next chunk's halo rows are issued at tap 0 BEFORE that tap's
            // weight stage - those rows while they are younger than this tap's weights (taps 1 .. DEPTH-1), but no later than
            // tap 2: from tap 3 on the in-place pass reads them.
            const int ahead = NT - 1 - kt < DEPTH - 1 ? NT - 1 - kt : DEPTH - 1;
            int pend = ahead * LB;
            if (has_next && tap >= 1 && tap <= (DEPTH - 1 < 2 ? DEPTH - 1 : 2)) pend += 3;
            if (dbg & (3 | 64)) pend = 0;
            wait_vm(pend);
            if (!(dbg & 16)) __builtin_amdgcn_s_barrier();    // stage `st` complete for every wave; the stage of tap kt-1 is free
            asm volatile("" ::: "memory");
            if (late && (tap > 0 || cl > 0) && !(dbg & 4)) mfma_tap();          // (the previous tap's fragments)
            const int ky = tap / 3, kx = tap - ky * 3;
            // next chunk's operand image, one 16-byte slot per lane at a time, spread over taps 3..8; the two waves that share a
            // SIMD take alternate taps.  A lane passes over its OWN slots only, which are complete once its tap-3 wait has
            // passed; the other waves see them after the barriers that follow.
            const bool slice = AFFINE && has_next && tap >= 3 && ((tap - 3) & 1) == wn && !(dbg & (2 | 32));
            Pending pd;
            if (slice) {
                if (tap < 5) tr_issue(Q0{}, cl + 1, pd);
                else if (tap < 7) tr_issue(Q1{}, cl + 1, pd);
                else tr_issue(Q2{}, cl + 1, pd);
            }
            if (!(dbg & 8)) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    fa[i] = *reinterpret_cast<const h16x8*>(hb + a_off[kx] + (i + ky) * HSTR * 128);
#pragma unroll
                for (int j = 0; j < 5; ++j)
                    fb[j] = *reinterpret_cast<const h16x8*>(smem + b_off + st * STAGE_BYTES + j * 2048);
            }
            if (tap == 0 && has_next && !(dbg & (2 | 64))) issue_a(cl + 1);   // (that image was consumed during the previous chunk)
            if (kt + DEPTH < NT && !(dbg & 1)) issue_b(kt + DEPTH, st == 0 ? NSTG - 1 : st - 1);
            if (!late && !(dbg & 4)) mfma_tap();
            if (slice) {
                if (tap < 5) tr_finish(Q0{}, cl + 1, pd);
                else if (tap < 7) tr_finish(Q1{}, cl + 1, pd);
                else tr_finish(Q2{}, cl + 1, pd);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // my reads of this stage and my slot stores are done before the next barrier
            st = st + 1 == NSTG ? 0 : st + 1;
        }
    }
    if (late && !(dbg & 4)) mfma_tap();                       // the last tap's fragments

    // ------------------------------------------------------------------------------------------------ epilogue
    stamp(4);
    __syncthreads();
    float* tilef = reinterpret_cast<float*>(smem);
    const int er = (wm * 4) * 16 + (lane >> 4) * 4, ec = wn * 80 + (lane & 15);
    if (kg == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) tilef[(er + i * 16 + r) * TS + ec + j * 16] = acc[i][j][r];
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) tilef[(er + i * 16 + r) * TS + ec + j * 16] += acc[i][j][r];
    }
    __syncthreads();
    stamp(5);

    // row-major pass: 24 rows x 20 eight-column chunks per sweep (480 of the 512 threads)
    const int col8 = tid % 20, row0 = tid / 20;
    const bool act = tid < 480;
    const int rpb = (int)g.div_rpb.d;
    if (p.splitk > 1) {
        if (act) {
            float* slab = p.slab + (size_t)split * p.M * p.N;
            for (int row = row0; row < HBM; row += 24) {
                const int m = b * rpb + (ty0 + (row >> 4)) * W + tx0 + (row & 15);
                const float4 lo = *reinterpret_cast<const float4*>(tilef + row * TS + col8 * 8);
                const float4 hi = *reinterpret_cast<const float4*>(tilef + row * TS + col8 * 8 + 4);
                float* dst = slab + (size_t)m * p.N + n0 + col8 * 8;
                *reinterpret_cast<float4*>(dst) = lo;
                *reinterpret_cast<float4*>(dst + 4) = hi;
            }
        }
        stamp(6);
        return;
    }
    float gs[8], gq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { gs[j] = 0.f; gq[j] = 0.f; }
    if (act) {
        const float alpha = scalar_alpha(p);
        Cols8 cols;
        cols8_init(g, cols, n0 + col8 * 8, n0 + col8 * 8, false, alpha);
        const float gt[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int row = row0; row < HBM; row += 24) {
            const int m = b * rpb + (ty0 + (row >> 4)) * W + tx0 + (row & 15);
            const float4 lo = *reinterpret_cast<const float4*>(tilef + row * TS + col8 * 8);
            const float4 hi = *reinterpret_cast<const float4*>(tilef + row * TS + col8 * 8 + 4);
            float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            epi8_store(g, cols, v, gt, m, gs, gq);
        }
    }
    if (p.gn_tot) {
        float* scr = reinterpret_cast<float*>(smem + OFF_SCR);     // [24][160][2], behind the tile
        if (act) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                scr[(row0 * HBN + col8 * 8 + j) * 2] = gs[j];
                scr[(row0 * HBN + col8 * 8 + j) * 2 + 1] = gq[j];
            }
        }
        __syncthreads();
        if (tid < HBN)                            // one add per totals block (bc_gn_cg): the block's first column sums its columns' 24 row partials
            bc_gn_tot_add_slot(p.gn_tot + (size_t)b * g.n_out * BC_GN_TOT_WORDS, n0 + tid, n0, n0 + HBN, bc_gn_cg(g.n_out), tin, [&](int k) {
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int r = 0; r < 24; ++r) {
                    s += scr[(r * HBN + k - n0) * 2];
                    q += scr[(r * HBN + k - n0) * 2 + 1];
                }
                return make_float2(s, q);
            });
    }
    stamp(6);
}

}  // namespace

int bc_conv_halo_max_chunks_impl() { return MAX_CH; }

// Eligibility of the halo kernel for a conv problem (the host-side planners mirror this).
int bc_conv_halo_ok(const BcGemm& p) {
    if (p.a_mode != BC_A_CONV3X3 || p.stride != 1 || p.conv_nopad_lo) return 0;
    // BC_TILE_WREG also takes the exact 2x nearest upsample in front of a plain convolution (no GroupNorm prologue, single source)
    const bool ups2 = p.tile_cfg == BC_TILE_WREG && p.Hv == 2 * p.Hin && p.Wv == 2 * p.Win && !p.A2 && !p.a_affine && !p.a_tot1;
    if (!ups2 && (p.Hv != p.Hin || p.Wv != p.Win)) return 0;
    if (p.Hout != p.Hv || p.Wout != p.Wv) return 0;
    if (p.Cin % 64 != 0 || p.N % HBN != 0 || p.Wout % TW != 0 || p.Hout % TH != 0) return 0;
    if (p.A2 && (p.C1 % 64 != 0)) return 0;
    return 1;
}

int bc_conv_halo_launch(GemmArgs& g, hipStream_t stream) {
    BcGemm& p = g.p;
    g.halo_tx = p.Wout / TW;
    g.halo_tpi = g.halo_tx * (p.Hout / TH);
    g.halo_nch = p.Cin / 64;
#ifdef BC_DIAGNOSTICS
    // ablation bits of the kernel (1 no weight DMA, 2 no halo path, 4 no MFMA, 8 no fragment reads, 16 no barriers, 32 no transform, 64 no
    // raw DMA): WRONG results by design, so they exist only in a library built with -DBC_DIAGNOSTICS (round 6: no switch that can break
    // correctness is reachable from a production import)
    static const int dbg_env = getenv("BC_HALO_DBG") ? atoi(getenv("BC_HALO_DBG")) : 0;
    g.halo_dbg = dbg_env;
#else
    g.halo_dbg = 0;
#endif
    int sk = std::max(1, std::min(p.splitk, g.halo_nch));
    g.halo_cps = bc_ceil_div(g.halo_nch, sk);
    p.splitk = bc_ceil_div(g.halo_nch, g.halo_cps);
    BC_CHECK_ARG(g.halo_cps <= MAX_CH, "bc_gemm(halo conv): %d channel chunks per split exceed %d (raise splitk)", g.halo_cps, MAX_CH);
    BC_CHECK_ARG(p.splitk == 1 || p.slab != nullptr, "bc_gemm(halo conv): splitk=%d needs a slab", p.splitk);
    const int B = p.M / (p.Hout * p.Wout);
    dim3 grid(p.N / HBN, B * g.halo_tpi, p.splitk);
    {
        g.nband = (double)p.N * 9 > (double)p.M && grid.x >= 4;        // (threshold ratio 1: 0.25 and 4 measured worse)
    }
    // in-kernel cycle stamps of this (fallback) kernel: -DBC_DIAGNOSTICS builds only (BC_HALO_STAMPS=1; synchronises the stream)
#ifdef BC_DIAGNOSTICS
    static const bool want_stamps = getenv("BC_HALO_STAMPS") != nullptr;
#else
    const bool want_stamps = false;
#endif
    static unsigned long long* stamp_buf = nullptr;
    const size_t nwg_s = (size_t)grid.x * grid.y * grid.z;
    g.halo_stamps = nullptr;
    if (want_stamps && nwg_s <= 4096) {
        if (!stamp_buf) BC_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&stamp_buf), 4096 * 8 * sizeof(unsigned long long)));
        BC_CHECK_HIP(hipMemsetAsync(stamp_buf, 0, nwg_s * 8 * sizeof(unsigned long long), stream));
        g.halo_stamps = stamp_buf;
    }
    struct StampReport {
        hipStream_t stream; size_t n; unsigned long long* buf; const BcGemm& p; int cps;
        ~StampReport() {
            if (!buf) return;
            if (hipStreamSynchronize(stream) != hipSuccess) return;
            std::vector<unsigned long long> h(n * 8);
            if (hipMemcpy(h.data(), buf, n * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return;
            double d[6] = {0, 0, 0, 0, 0, 0};
            unsigned long long t0 = ~0ull, t1 = 0;
            for (size_t i = 0; i < n; ++i) {
                for (int k = 0; k < 6; ++k) d[k] += (double)(h[i * 8 + k + 1] - h[i * 8 + k]);
                t0 = std::min(t0, h[i * 8]);
                t1 = std::max(t1, h[i * 8 + 6]);
            }
            fprintf(stderr, "[halo stamps] M=%d N=%d Cin=%d sk=%d cps=%d wgs=%zu | avg ticks: setup %.0f, dma+table %.0f, first halo pass %.0f, "
                    "loop %.0f, k-half sum %.0f, stores %.0f | first entry -> last exit %llu ticks\n", p.M, p.N, p.Cin, p.splitk, cps, n,
                    d[0] / n, d[1] / n, d[2] / n, d[3] / n, d[4] / n, d[5] / n, t1 - t0);
        }
    } report{stream, nwg_s, g.halo_stamps, p, g.halo_cps};
    static std::atomic<unsigned long long> set_a{0}, set_p{0}, set_f{0};
    if (p.a_tot1) {
        BC_CHECK_ARG(p.a_gamma && p.a_beta && p.a_groups > 0 && p.Cin % p.a_groups == 0 && (!p.A2 || p.a_tot2),
                     "bc_gemm(halo conv): in-kernel GroupNorm finalize needs a_gamma, a_beta, a_groups | Cin and the partials of every source");
        const int cpg = p.Cin / p.a_groups;
        BC_CHECK_ARG(g.halo_cps * 64 + 2 * cpg <= FIN_MAX_CH && (8 * (g.halo_cps * 64 + 2 * cpg) + p.a_groups + 8) * 8 <= 2 * HALO_BYTES,
                     "bc_gemm(halo conv): channel span %d per workgroup too wide for the in-kernel GroupNorm finalize (max %d): use "
                     "bc_gn_finalize + a_affine or raise splitk", g.halo_cps * 64 + 2 * cpg, FIN_MAX_CH);
        BC_CHECK_HIP(bc_set_max_lds(set_f, reinterpret_cast<const void*>(&conv_halo_kernel<2>), LDS_TOTAL));
        hipLaunchKernelGGL((conv_halo_kernel<2>), grid, dim3(512), LDS_TOTAL, stream, g);
    } else if (p.a_affine) {
        BC_CHECK_HIP(bc_set_max_lds(set_a, reinterpret_cast<const void*>(&conv_halo_kernel<1>), LDS_TOTAL));
        hipLaunchKernelGGL((conv_halo_kernel<1>), grid, dim3(512), LDS_TOTAL, stream, g);
    } else {
        BC_CHECK_HIP(bc_set_max_lds(set_p, reinterpret_cast<const void*>(&conv_halo_kernel<0>), LDS_TOTAL));
        hipLaunchKernelGGL((conv_halo_kernel<0>), grid, dim3(512), LDS_TOTAL, stream, g);
    }
    BC_CHECK_LAUNCH();
    return 0;
}
