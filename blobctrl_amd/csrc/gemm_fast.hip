// Fast-path implicit GEMM for gfx950: LDS-DMA staged (global_load_lds_dwordx4), v_mfma_f32_32x32x16_f16.
//
// Same contract as gemm.hip (C = epilogue(A . W^T)), restricted to the shapes that carry the FLOPs of the hot path:
//   K % 64 == 0 (conv: Cin % 64 == 0; concat split C1 % 64 == 0), exact 2x nearest upsample or none.
// What is different from the generic kernel:
//   * operands go HBM/L2 -> LDS directly (no VGPR round trip, no ds_write): each wave issues 1-KiB pieces whose per-lane
//     SOURCE address carries the XOR swizzle ((row>>1)&7 on 16-byte chunks) while the LDS image stays lane-linear
//     (cdna_hip_programming.md section 5, rule 21); out-of-range rows / conv halo read a 16-byte zero line instead.
//   * im2col addressing is hoisted: per row one element offset + a 9-bit tap-validity mask are computed once; a K-tile
//     adds one wave-uniform scalar (tap offset + channel offset).
//   * two LDS stages, loads of tile k+1 in flight while tile k is multiplied; 64 KiB -> two workgroups per CU.
//   * epilogue in two phases: accumulators (+bias, +time-embedding row vector, activation, LayerScale, scale) are
//     parked as fp32 in the (now free) LDS tile, then re-read row-major so that the residual loads, the BlobNet
//     right-half add and the fp16 stores are all 16-byte coalesced; the same pass optionally emits per-channel
//     (sum, sumsq) partials of the fp16-rounded output for the consumer's GroupNorm (no extra pass over the tensor).
#include <stdlib.h>
#include "gemm_common.h"

using namespace bcg;

namespace {

__device__ __attribute__((aligned(16))) unsigned int g_zero_line[4] = {0u, 0u, 0u, 0u};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* src, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_dst, 16, 0, 0);
}

template <int BM, int BN, int WM, int WN, int NS, bool CONV, bool UPS>
__global__ __launch_bounds__(64 * WM * WN) void gemm_fast_kernel(const GemmArgs g) {
    constexpr int NT = 64 * WM * WN;
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int A_LOADS = BM / 8 / NW;      // 1-KiB pieces (8 rows x 128 B) per wave
    constexpr int B_LOADS = BN / 8 / NW;
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile/waves mismatch");
    constexpr int STAGE = (BM + BN) * 128;    // bytes per pipeline stage

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const BcGemm& p = g.p;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int plane = gridDim.x * gridDim.y;
    const int lin3 = bc_xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), plane * gridDim.z);
    const int lin = lin3 % plane;
    // an XCD (a run of consecutive `lin`) owns a contiguous band of row tiles with all their column tiles - its L2 then fetches 1/8
    // of the activation and the whole weight matrix - or, when the weights are the larger operand (g.nband), a band of column tiles
    // with all their row tiles: 1/8 of the weights, the whole activation
    const int m0 = (g.nband ? lin % (int)gridDim.y : lin / (int)gridDim.x) * BM;
    const int n0 = (g.nband ? lin / (int)gridDim.y : lin % (int)gridDim.x) * BN;
    const int split = lin3 / plane;

    const h16* __restrict__ A = reinterpret_cast<const h16*>(p.A);
    const h16* __restrict__ A2 = reinterpret_cast<const h16*>(p.A2);
    const h16* __restrict__ W = reinterpret_cast<const h16*>(p.W);
    const h16* zero = reinterpret_cast<const h16*>(g_zero_line);

    const int prow = lane >> 3;          // row inside a piece
    const int pslot = lane & 7;          // LDS slot inside the row

    // ---- per-piece source state (A) ----
    long long a_off[A_LOADS];            // element offset of this lane's chunk at k0 = 0 (tap (0,0) for convs)
    long long a2_off[A_LOADS];
    int a_mask[A_LOADS];                 // conv: 9-bit tap validity; dense: 0 / all ones
    int a_py[A_LOADS], a_px[A_LOADS];    // UPS only: output pixel coordinates
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
        const int row = (wave + i * NW) * 8 + prow;
        const int cs = pslot ^ ((row >> 1) & 7);           // global chunk this lane fetches (source-side swizzle)
        const int m = m0 + row;
        const bool ok = m < p.M;
        a2_off[i] = 0;
        a_py[i] = a_px[i] = 0;
        if (CONV) {
            const int mm = ok ? m : 0;
            const int hw = p.Hout * p.Wout;
            const int b = (int)fdiv((unsigned)mm, g.div_rpb);
            const int rem = mm - b * hw;
            const int oy = (int)fdiv((unsigned)rem, g.div_wout);
            const int ox = rem - oy * p.Wout;
            if (UPS) {
                a_off[i] = (long long)b * p.Hin * p.Win;           // pixel base; finished per K-tile
                a_py[i] = oy;
                a_px[i] = ox;
                a_mask[i] = ok ? 0x1ff : 0;
                a2_off[i] = cs * 8;
            } else {
                const int pad_lo = p.conv_nopad_lo ? 0 : 1;
                const int iy0 = oy * p.stride - pad_lo, ix0 = ox * p.stride - pad_lo;
                a_off[i] = (((long long)b * p.Hin + iy0) * p.Win + ix0) * p.Cin + cs * 8;
                int mask = 0;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int iy = iy0 + t / 3, ix = ix0 + t % 3;
                    if (ok && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win) mask |= 1 << t;
                }
                a_mask[i] = mask;
            }
        } else {
            a_off[i] = (long long)m * p.lda + cs * 8;
            a2_off[i] = (long long)m * p.lda2 + cs * 8 - p.C1;
            a_mask[i] = ok ? 0x1ff : 0;
        }
    }
    long long b_off[B_LOADS];
    bool b_ok[B_LOADS];
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
        const int row = (wave + i * NW) * 8 + prow;
        const int cs = pslot ^ ((row >> 1) & 7);
        const int n = n0 + row;
        b_ok[i] = n < p.N;
        b_off[i] = (long long)n * p.ldw + cs * 8;
    }

    const int kt_begin = split * g.kt_per_split;
    const int kt_end = min(g.nk, kt_begin + g.kt_per_split);

    auto issue = [&](int kt, int buf) {
        char* la = smem + buf * STAGE;
        char* lb = la + BM * 128;
        const int k0 = kt * BK;
        if (CONV) {
            const int tap = k0 / p.Cin;                      // wave-uniform (Cin % 64 == 0)
            const int c0 = k0 - tap * p.Cin;
            const int ky = tap / 3, kx = tap - ky * 3;
            if (UPS) {
#pragma unroll
                for (int i = 0; i < A_LOADS; ++i) {
                    const int iyv = a_py[i] + ky - 1, ixv = a_px[i] + kx - 1;
                    const bool ok = a_mask[i] && (unsigned)iyv < (unsigned)p.Hv && (unsigned)ixv < (unsigned)p.Wv;
                    const long long off = (a_off[i] + (long long)(iyv >> 1) * p.Win + (ixv >> 1)) * p.Cin + c0 + a2_off[i];
                    glds16(ok ? (const void*)(A + off) : (const void*)zero, la + (wave + i * NW) * 1024);
                }
            } else {
                const long long koff = (long long)(ky * p.Win + kx) * p.Cin + c0;
#pragma unroll
                for (int i = 0; i < A_LOADS; ++i) {
                    const bool ok = (a_mask[i] >> tap) & 1;
                    glds16(ok ? (const void*)(A + a_off[i] + koff) : (const void*)zero, la + (wave + i * NW) * 1024);
                }
            }
        } else {
            const bool second = (A2 != nullptr) && (k0 >= p.C1);   // wave-uniform (C1 % 64 == 0)
#pragma unroll
            for (int i = 0; i < A_LOADS; ++i) {
                const h16* src = second ? (A2 + a2_off[i] + k0) : (A + a_off[i] + k0);
                glds16(a_mask[i] ? (const void*)src : (const void*)zero, la + (wave + i * NW) * 1024);
            }
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i)
            glds16(b_ok[i] ? (const void*)(W + b_off[i] + k0) : (const void*)zero, lb + (wave + i * NW) * 1024);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int frow = lane & 31;
    const int fhalf = lane >> 5;

    // NS-stage LDS ring, NS-1 tiles of LDS-DMA in flight.  Counted vmcnt + raw s_barrier: a __syncthreads() here would
    // drain every outstanding LDS-DMA (cdna_hip_programming.md section 5 "Pipelining across barriers").
    constexpr int L = A_LOADS + B_LOADS;              // LDS-DMA instructions per wave per tile
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (kt_begin + s < kt_end) issue(kt_begin + s, s);
    int cur = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int remaining = kt_end - 1 - kt;        // tiles issued after tile kt
        if (NS >= 4 && remaining >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * L) : "memory");
        else if (NS >= 3 && remaining >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // tile kt landed for every wave; stage (kt-1)%NS is free again
        asm volatile("" ::: "memory");
        if (kt + NS - 1 < kt_end) {
            int nxt = cur + NS - 1;
            if (nxt >= NS) nxt -= NS;
            issue(kt + NS - 1, nxt);
        }
        const char* la = smem + cur * STAGE;
        const char* lb = la + BM * 128;
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            h16x8 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * (BM / WM) + i * 32 + frow;
                fa[i] = *reinterpret_cast<const h16x8*>(la + lds_off(row, 2 * s + fhalf));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn * (BN / WN) + j * 32 + frow;
                fb[j] = *reinterpret_cast<const h16x8*>(lb + lds_off(row, 2 * s + fhalf));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my LDS reads of this stage are done before I pass the next barrier
        cur = (cur + 1 == NS) ? 0 : cur + 1;
    }

    // ------------------------------------------------------------------------------------------------ epilogue
    __syncthreads();                                   // all waves are done reading the operand stages
    float* tile = reinterpret_cast<float*>(smem);      // raw fp32 accumulators [BM][BN]
    if (g.vec_transposed && p.splitk == 1) {
        acc_to_tile_t<TM, TN>(tile, BM + 4, acc, wm * (BM / WM), wn * (BN / WN), frow, fhalf);
        __syncthreads();
        tile_epilogue_transposed<BM, BN, NT>(g, tile, BM + 4, m0, n0, tid);
        return;
    }
    acc_to_tile<TM, TN>(tile, BN, acc, wm * (BM / WM), wn * (BN / WN), frow, fhalf);
    __syncthreads();
    if (p.splitk > 1 || !g.vec_epilogue) {
        tile_epilogue_scalar<BM, BN, NT>(g, tile, m0, n0, split, tid);
        return;
    }

    // ---- row-major pass, 8 output columns per thread: bias, time-embedding row vector, activation, LayerScale, scale,
    // ---- residual, BlobNet right-half residual, fp16 store (all 16-byte accesses) and GroupNorm partials ----
    const float alpha = scalar_alpha(p);
    const bool geglu = p.act == BC_ACT_GEGLU;
    const int TSO = geglu ? BN / 2 : BN;            // output columns of this block
    const int CPR = TSO / 8;                        // chunks per tile row (power of two: 4, 8, 16)
    const int col8 = tid & (CPR - 1);
    const int rstep = NT / CPR;
    const int c_out = col8 * 8;                     // first output column (block-local)
    const int n_first = (geglu ? n0 / 2 : n0) + c_out;
    const bool nok = n_first < g.n_out;
    const int cv = geglu ? (c_out >> 5) * 64 + (c_out & 31) : c_out;      // tile column of the (value) accumulators
    Cols8 cols;
    cols8_init(g, cols, n_first, n0 + cv, geglu, alpha);
    float gs[8], gq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { gs[j] = 0.f; gq[j] = 0.f; }
    for (int row = tid / CPR; row < BM; row += rstep) {
        const int m = m0 + row;
        if (m < p.M && nok) {
            const float4 lo = *reinterpret_cast<const float4*>(tile + row * BN + cv);
            const float4 hi = *reinterpret_cast<const float4*>(tile + row * BN + cv + 4);
            float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            float gt[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (geglu) {
                const float4 glo = *reinterpret_cast<const float4*>(tile + row * BN + cv + 32);
                const float4 ghi = *reinterpret_cast<const float4*>(tile + row * BN + cv + 36);
                gt[0] = glo.x; gt[1] = glo.y; gt[2] = glo.z; gt[3] = glo.w;
                gt[4] = ghi.x; gt[5] = ghi.y; gt[6] = ghi.z; gt[7] = ghi.w;
            }
            epi8_store(g, cols, v, gt, m, gs, gq);
        }
    }
    if (p.gn_tot) {
        // reduce the per-thread column partials over the rows of this block: lanes with equal col8 inside a wave, then waves
        for (int o = CPR; o < 64; o <<= 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                gs[j] += __shfl_xor(gs[j], o);
                gq[j] += __shfl_xor(gq[j], o);
            }
        }
        __syncthreads();                            // tile fully consumed; reuse its head as scratch [NW][TSO][2]
        float* scr = reinterpret_cast<float*>(smem);
        if (lane < CPR) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                scr[((wave * TSO) + c_out + j) * 2] = gs[j];
                scr[((wave * TSO) + c_out + j) * 2 + 1] = gq[j];
            }
        }
        __syncthreads();
        if (tid < TSO) {
            const int nb = geglu ? n0 / 2 : n0, n = nb + tid;
            if (n < g.n_out) {
                const int b = (int)fdiv((unsigned)m0, g.div_rpb);
                bc_gn_tot_add_slot(p.gn_tot + (size_t)b * g.n_out * BC_GN_TOT_WORDS, n, nb, min(nb + TSO, g.n_out), bc_gn_cg(g.n_out), m0 / BM, [&](int k) {
                    float s = 0.f, q = 0.f;
#pragma unroll
                    for (int w = 0; w < NW; ++w) {
                        s += scr[(w * TSO + k - nb) * 2];
                        q += scr[(w * TSO + k - nb) * 2 + 1];
                    }
                    return make_float2(s, q);
                });
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int NS, bool CONV, bool UPS>
int launch_fast(const GemmArgs& g_in, hipStream_t stream) {
    GemmArgs g = g_in;
    const BcGemm& p = g.p;
    dim3 grid(bc_ceil_div(p.N, BN), bc_ceil_div(p.M, BM), p.splitk);
    {
        // operand bytes one split touches: activation rows (a convolution's nine taps re-read the same pixels) vs weights
        const double a_bytes = (double)p.M * (CONV ? p.Cin : p.K) / p.splitk, w_bytes = (double)p.N * p.K / p.splitk;
        g.nband = w_bytes > a_bytes && grid.x >= 8;              // (threshold ratio 1: 0.25 and 4 measured worse, forced column bands +0.15 ms: DESIGN 3.1)
    }
    dim3 block(64 * WM * WN);
    size_t lds = std::max<size_t>((size_t)NS * (BM + BN) * 128, (size_t)BN * (BM + 4) * 4);   // stages | epilogue tile (| transposed, padded)
    static std::atomic<unsigned long long> lds_set{0};       // one bit per device ordinal
    BC_CHECK_HIP(bc_set_max_lds(lds_set, reinterpret_cast<const void*>(&gemm_fast_kernel<BM, BN, WM, WN, NS, CONV, UPS>), (int)lds));
    hipLaunchKernelGGL((gemm_fast_kernel<BM, BN, WM, WN, NS, CONV, UPS>), grid, block, lds, stream, g);
    BC_CHECK_LAUNCH();
    return 0;
}

template <int BM, int BN, int WM, int WN, int NS>
int launch_mode(const GemmArgs& g, bool conv, bool ups, hipStream_t stream) {
    if (conv) return ups ? launch_fast<BM, BN, WM, WN, NS, true, true>(g, stream) : launch_fast<BM, BN, WM, WN, NS, true, false>(g, stream);
    return launch_fast<BM, BN, WM, WN, NS, false, false>(g, stream);
}

}  // namespace

int bc_gemm_fast_try(const GemmArgs& g, hipStream_t stream) {
    const BcGemm& p = g.p;
    if (p.K % BK != 0) return -1;
    const bool conv = p.a_mode == BC_A_CONV3X3;
    bool ups = false;
    if (conv) {
        if (p.Cin % BK != 0) return -1;
        ups = (p.Hv != p.Hin) || (p.Wv != p.Win);
        if (ups && !(p.Hv == 2 * p.Hin && p.Wv == 2 * p.Win && p.stride == 1)) return -1;
    } else if (p.A2 && (p.C1 % BK != 0)) {
        return -1;
    }
    switch (g.cfg) {
        case BC_TILE_256x128:    return launch_mode<256, 128, 4, 2, 3>(g, conv, ups, stream);
        case BC_TILE_128x128_S3: return launch_mode<128, 128, 2, 2, 3>(g, conv, ups, stream);
        case BC_TILE_128x128_S2: return launch_mode<128, 128, 2, 2, 2>(g, conv, ups, stream);
        case BC_TILE_256x64_S2:  return launch_mode<256, 64, 4, 1, 2>(g, conv, ups, stream);
        case BC_TILE_256x64_S3:  return launch_mode<256, 64, 4, 1, 3>(g, conv, ups, stream);
        case BC_TILE_128x64:     return launch_mode<128, 64, 2, 2, 3>(g, conv, ups, stream);
        case BC_TILE_64x64:      return launch_mode<64, 64, 2, 2, 4>(g, conv, ups, stream);
        default: return -1;
    }
}
