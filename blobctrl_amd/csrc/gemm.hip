// Implicit GEMM for the BlobCtrl hot path on gfx950:  C[M][N] = epilogue( A[M][K] . W[N][K]^T ), fp16 in, fp32 accumulate.
//
//   * A operand is either dense rows (linear layers, 1x1 convs over NHWC, optional channel-concat of two tensors) or
//     gathered on the fly from an NHWC image (3x3 conv, pad 1, stride 1|2, optional fused nearest upsample).
//   * v_mfma_f32_32x32x16_f16, 64-wide wavefronts; block tile BM x BN x 64, register-staged double-buffered LDS with an
//     XOR swizzle ((row>>1)&7 on 16-byte chunks of 128-byte rows) that makes every ds_read_b128 fragment read
//     conflict-free (MI355X_MICROARCH.md LDS table: b128 reads are served in 16-lane groups over 64 banks).
//   * epilogue fuses bias, per-batch row vector (time embedding), GELU / GEGLU / SiLU, LayerScale, scalar scale
//     (BlobNet conditioning scale from a device table), residual add, BlobNet right-half residual add,
//     and transposed output (V^T for attention).
//   * split-K writes fp32 slabs; bc_splitk_reduce applies the same epilogue.
//
// Reference call sites replaced: see include/blobctrl_hip.h (BcGemm).
#include <stdlib.h>
#include "gemm_common.h"

using namespace bcg;
namespace {

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void gemm_kernel(const GemmArgs g) {
    constexpr int NT = 64 * WM * WN;
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int A_LOADS = BM * 8 / NT;
    constexpr int B_LOADS = BN * 8 / NT;
    constexpr int ROWS_PER_PASS = NT / 8;
    static_assert(BM * 8 % NT == 0 && BN * 8 % NT == 0, "tile/threads mismatch");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lds_a = smem;                        // [2][BM][128 B]
    char* lds_b = smem + 2 * BM * 128;         // [2][BN][128 B]

    const BcGemm& p = g.p;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int plane = gridDim.x * gridDim.y;
    const int lin3 = bc_xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), plane * gridDim.z);
    const int lin = lin3 % plane;
    const int m0 = (lin / (int)gridDim.x) * BM;        // an XCD owns a contiguous band of row tiles, all their column tiles
    const int n0 = (lin % (int)gridDim.x) * BN;
    const int split = lin3 / plane;

    const h16* __restrict__ A = reinterpret_cast<const h16*>(p.A);
    const h16* __restrict__ A2 = reinterpret_cast<const h16*>(p.A2);
    const h16* __restrict__ W = reinterpret_cast<const h16*>(p.W);

    const int cc = tid & 7;             // 16-byte chunk column handled by this thread
    const int r_base = tid >> 3;        // first row handled by this thread

    // ---- per-row state for the A gather ----
    int a_base[A_LOADS], a_iy0[A_LOADS], a_ix0[A_LOADS];
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
        int m = m0 + r_base + i * ROWS_PER_PASS;
        bool ok = m < p.M;
        if (p.a_mode == BC_A_CONV3X3) {
            int mm = ok ? m : 0;
            int hw = p.Hout * p.Wout;
            int b = (int)fdiv((unsigned)mm, g.div_rpb);        // rows_per_batch == Hout*Wout for convs
            int rem = mm - b * hw;
            int oy = (int)fdiv((unsigned)rem, g.div_wout);
            int ox = rem - oy * p.Wout;
            a_base[i] = b * p.Hin * p.Win;
            const int pad_lo = p.conv_nopad_lo ? 0 : 1;
            a_iy0[i] = ok ? oy * p.stride - pad_lo : -(1 << 20);
            a_ix0[i] = ox * p.stride - pad_lo;
        } else {
            a_base[i] = ok ? m : -1;
            a_iy0[i] = 0;
            a_ix0[i] = 0;
        }
    }
    const bool upsample = (p.Hv != p.Hin) || (p.Wv != p.Win);

    const int kt_begin = split * g.kt_per_split;
    const int kt_end = min(g.nk, kt_begin + g.kt_per_split);

    uint4 ra[A_LOADS], rb[B_LOADS];

    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
        const int k = k0 + cc * 8;
        // ---- A ----
        if (p.a_mode == BC_A_CONV3X3) {
            int tap, c;
            if (g.fast_k) {
                tap = k0 / p.Cin;               // uniform over the block
                c = k0 - tap * p.Cin + cc * 8;
            } else {
                tap = k / p.Cin;
                c = k - tap * p.Cin;
            }
            const int ky = tap / 3, kx = tap - ky * 3;
            const bool kok = k < p.K;
#pragma unroll
            for (int i = 0; i < A_LOADS; ++i) {
                int iyv = a_iy0[i] + ky, ixv = a_ix0[i] + kx;
                bool ok = kok && (unsigned)iyv < (unsigned)p.Hv && (unsigned)ixv < (unsigned)p.Wv;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (ok) {
                    int iy = iyv, ix = ixv;
                    if (upsample) {              // nearest: src = floor(dst * in / out)  (upsampling.py:169)
                        iy = iyv * p.Hin / p.Hv;
                        ix = ixv * p.Win / p.Wv;
                    }
                    size_t pix = (size_t)(a_base[i] + iy * p.Win + ix);
                    v = bc_ld16(A + pix * p.Cin + c);
                }
                ra[i] = v;
            }
        } else {
            const bool kok = k < p.K;
            const bool second = (A2 != nullptr) && (k >= p.C1);
#pragma unroll
            for (int i = 0; i < A_LOADS; ++i) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (kok && a_base[i] >= 0) {
                    const h16* src = second ? (A2 + (size_t)a_base[i] * p.lda2 + (k - p.C1))
                                            : (A + (size_t)a_base[i] * p.lda + k);
                    v = bc_ld16(src);
                }
                ra[i] = v;
            }
        }
        // ---- B (weights) ----
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            int n = n0 + r_base + i * ROWS_PER_PASS;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (n < p.N && k < p.K) v = bc_ld16(W + (size_t)n * p.ldw + k);
            rb[i] = v;
        }
    };

    auto store_tile = [&](int buf) {
        char* la = lds_a + buf * BM * 128;
        char* lb = lds_b + buf * BN * 128;
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) bc_st16(la + lds_off(r_base + i * ROWS_PER_PASS, cc), ra[i]);
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) bc_st16(lb + lds_off(r_base + i * ROWS_PER_PASS, cc), rb[i]);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int frow = lane & 31;      // fragment row inside a 32-row MFMA tile
    const int fhalf = lane >> 5;     // which 8-element k group

    if (kt_begin < kt_end) {
        load_tile(kt_begin);
        store_tile(0);
    }
    __syncthreads();

    int cur = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const bool more = (kt + 1) < kt_end;
        if (more) load_tile(kt + 1);

        const char* la = lds_a + cur * BM * 128;
        const char* lb = lds_b + cur * BN * 128;
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            h16x8 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                int row = wm * (BM / WM) + i * 32 + frow;
                fa[i] = *reinterpret_cast<const h16x8*>(la + lds_off(row, 2 * s + fhalf));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                int row = wn * (BN / WN) + j * 32 + frow;
                fb[j] = *reinterpret_cast<const h16x8*>(lb + lds_off(row, 2 * s + fhalf));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }

        if (more) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ------------------------------------------------------------------------------------------------ epilogue
    __syncthreads();                                   // operand stages are dead: reuse LDS as the fp32 output tile
    float* tile = reinterpret_cast<float*>(smem);
    acc_to_tile<TM, TN>(tile, BN, acc, wm * (BM / WM), wn * (BN / WN), frow, fhalf);
    __syncthreads();
    tile_epilogue_scalar<BM, BN, NT>(g, tile, m0, n0, split, tid);
}

// Sum split-K slabs and apply the epilogue.  One thread per output element, n fastest (coalesced slab reads).
__global__ void splitk_reduce_kernel(const GemmArgs g) {
    const BcGemm& p = g.p;
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)p.M * g.n_out;
    if (idx >= total) return;
    int m = (int)(idx / g.n_out);
    int n = (int)(idx - (long long)m * g.n_out);
    const float alpha = scalar_alpha(p);
    const size_t mn = (size_t)p.M * p.N;
    if (p.act == BC_ACT_GEGLU) {
        int nv = (n >> 5) * 64 + (n & 31), ng = nv + 32;
        float av = 0.f, ag = 0.f;
        for (int z = 0; z < p.splitk; ++z) {
            av += p.slab[z * mn + (size_t)m * p.N + nv];
            ag += p.slab[z * mn + (size_t)m * p.N + ng];
        }
        float v = pre_act(g, av, m, nv), gt = pre_act(g, ag, m, ng);
        epilogue_store(g, v * bc_gelu_f(gt), m, n, alpha);
        return;
    }
    float a = 0.f;
    for (int z = 0; z < p.splitk; ++z) a += p.slab[z * mn + (size_t)m * p.N + n];
    float v = pre_act(g, a, m, n);
    if (p.act == BC_ACT_GELU) v = bc_gelu_f(v);
    else if (p.act == BC_ACT_SILU) v = bc_silu_f(v);
    else if (p.act == BC_ACT_QUICK_GELU) v = bc_quick_gelu_f(v);
    epilogue_store(g, v, m, n, alpha);
}

// Vectorised reducer: a workgroup owns SK_ROWS rows x 64 output columns (8 columns per thread), sums the slabs with
// 16-byte loads, applies the shared 8-wide epilogue and (optionally) adds the GroupNorm statistics of its row block to the
// totals gn_tot[B][n_out][BC_GN_TOT_WORDS].
constexpr int SK_ROWS = 32;
__global__ __launch_bounds__(256) void splitk_reduce_vec_kernel(const GemmArgs g) {
    __shared__ float scr[4 * 64 * 2];
    const BcGemm& p = g.p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col8 = tid & 7, row = tid >> 3;                 // 8 chunks x 32 rows
    const int m = blockIdx.y * SK_ROWS + row;
    const int n_first = blockIdx.x * 64 + col8 * 8;
    const bool geglu = p.act == BC_ACT_GEGLU;
    const int ncol_v = geglu ? (n_first >> 5) * 64 + (n_first & 31) : n_first;
    const float alpha = scalar_alpha(p);
    Cols8 cols;
    cols8_init(g, cols, n_first, ncol_v, geglu, alpha);
    float gs[8], gq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { gs[j] = 0.f; gq[j] = 0.f; }
    if (m < p.M && cols.nok) {
        const size_t mn = (size_t)p.M * p.N;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gt[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        // slabs four at a time: eight independent 16-byte loads in flight per thread instead of one dependent round per slab
        // (split-K runs to 14 slabs; summation order is unchanged: z ascending)
        const float* base = p.slab + (size_t)m * p.N + ncol_v;
        int z = 0;
        for (; !geglu && z + 4 <= p.splitk; z += 4) {
            float4 lo[4], hi[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                lo[u] = *reinterpret_cast<const float4*>(base + (size_t)(z + u) * mn);
                hi[u] = *reinterpret_cast<const float4*>(base + (size_t)(z + u) * mn + 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[0] += lo[u].x; v[1] += lo[u].y; v[2] += lo[u].z; v[3] += lo[u].w;
                v[4] += hi[u].x; v[5] += hi[u].y; v[6] += hi[u].z; v[7] += hi[u].w;
            }
        }
        for (; z < p.splitk; ++z) {
            const float* src = base + (size_t)z * mn;
            const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
            v[0] += lo.x; v[1] += lo.y; v[2] += lo.z; v[3] += lo.w; v[4] += hi.x; v[5] += hi.y; v[6] += hi.z; v[7] += hi.w;
            if (geglu) {
                const float4 glo = *reinterpret_cast<const float4*>(src + 32), ghi = *reinterpret_cast<const float4*>(src + 36);
                gt[0] += glo.x; gt[1] += glo.y; gt[2] += glo.z; gt[3] += glo.w;
                gt[4] += ghi.x; gt[5] += ghi.y; gt[6] += ghi.z; gt[7] += ghi.w;
            }
        }
        epi8_store(g, cols, v, gt, m, gs, gq);
    }
    if (p.gn_tot) {
        for (int o = 8; o < 64; o <<= 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                gs[j] += __shfl_xor(gs[j], o);
                gq[j] += __shfl_xor(gq[j], o);
            }
        }
        if (lane < 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                scr[(wave * 64 + col8 * 8 + j) * 2] = gs[j];
                scr[(wave * 64 + col8 * 8 + j) * 2 + 1] = gq[j];
            }
        }
        __syncthreads();
        if (tid < 64) {
            const int nb = blockIdx.x * 64, n = nb + tid;
            if (n < g.n_out) {
                const int b = (int)fdiv((unsigned)(blockIdx.y * SK_ROWS), g.div_rpb);
                bc_gn_tot_add_slot(p.gn_tot + (size_t)b * g.n_out * BC_GN_TOT_WORDS, n, nb, min(nb + 64, g.n_out), bc_gn_cg(g.n_out), (int)blockIdx.y, [&](int k) {
                    float s = 0.f, q = 0.f;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        s += scr[(w * 64 + k - nb) * 2];
                        q += scr[(w * 64 + k - nb) * 2 + 1];
                    }
                    return make_float2(s, q);
                });
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
int launch_gemm(const GemmArgs& g, hipStream_t stream) {
    const BcGemm& p = g.p;
    dim3 grid(bc_ceil_div(p.N, BN), bc_ceil_div(p.M, BM), p.splitk);
    dim3 block(64 * WM * WN);
    size_t lds = 2 * (BM + BN) * 128;
    static std::atomic<unsigned long long> lds_set{0};       // one bit per device ordinal
    BC_CHECK_HIP(bc_set_max_lds(lds_set, reinterpret_cast<const void*>(&gemm_kernel<BM, BN, WM, WN>), (int)lds));
    hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN>), grid, block, lds, stream, g);
    BC_CHECK_LAUNCH();
    return 0;
}

}  // namespace

namespace {

struct TileCfg { int bm, bn, threads, ns, lds; };
// index = BC_TILE_*
const TileCfg kTiles[BC_TILE_COUNT] = {
    {0, 0, 0, 0, 0},
    {256, 128, 512, 3, 3 * (256 + 128) * 128},
    {128, 128, 256, 3, 3 * (128 + 128) * 128},
    {128, 128, 256, 2, 2 * (128 + 128) * 128},
    {256, 64, 256, 2, 2 * (256 + 64) * 128},
    {256, 64, 256, 3, 3 * (256 + 64) * 128},
    {128, 64, 256, 3, 3 * (128 + 64) * 128},
    {64, 64, 256, 4, 4 * (64 + 64) * 128},
};

int g_num_cu = 0;

int num_cu() {
    if (g_num_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) g_num_cu = prop.multiProcessorCount;
        if (g_num_cu <= 0) g_num_cu = 256;
    }
    return g_num_cu;
}

// Cost model (microseconds).  Operand staging is bounded by the per-CU L2->LDS rate (~70 GB/s measured for LDS-DMA
// gathers, MI355X_MICROARCH.md "Indexed rows: gather into LDS") and, when few bytes are in flight per CU, by latency
// (~1 us per dependent tile); MFMA at ~9.8 TFLOP/s per CU.  Split-K adds the fp32 slab round trip and a second launch.
double model_us(const TileCfg& t, int M, int N, int K, int sk) {
    const int cus = num_cu();
    const long long nblk = (long long)bc_ceil_div(M, t.bm) * bc_ceil_div(N, t.bn) * sk;
    const int nkb = bc_ceil_div(bc_ceil_div(K, BK), sk);
    const double stage = (double)(t.bm + t.bn) * 128.0;
    const int by_lds = std::max(1, (160 * 1024) / std::max(t.lds, t.bm * t.bn * 4));
    const int resident = (int)std::min<long long>(std::min(by_lds, 2048 / t.threads), (nblk + cus - 1) / cus);
    const double inflight = resident * (t.ns - 1) * stage;                 // bytes in flight per CU
    const double bw = std::min(70e3, inflight / 1.0);                      // bytes per microsecond per CU
    const double per_cu_blocks = (double)((nblk + cus - 1) / cus);
    const double t_load = per_cu_blocks * nkb * stage / bw;
    const double t_mfma = per_cu_blocks * nkb * (2.0 * t.bm * t.bn * BK) / (9.8e6 * 0.8);
    double us = std::max(t_load, t_mfma) + 3.0 + per_cu_blocks * (t.bm * t.bn) / 16384.0 * 1.0;   // + epilogue per block
    if (sk > 1) us += 3.0 + 2.0 * sk * (double)M * N * 4.0 / 3.0e6;
    return us;
}

}  // namespace

extern "C" int bc_gemm_plan(int M, int N, int K, int fast, int* tile_cfg, int* splitk, int* bm, int* bn) {
    BC_CHECK_ARG(tile_cfg && splitk && bm && bn && M > 0 && N > 0 && K > 0, "bc_gemm_plan: bad args");
    const int nk = bc_ceil_div(K, BK);
    int cfg = *tile_cfg, sk = *splitk;
    if (!fast) {
        // generic kernel: 128x128, or 256x64 when N would be padded by > 10 % on a 128-wide tile
        int n128 = bc_ceil_div(N, 128) * 128;
        cfg = ((double)n128 / N > 1.10) ? BC_TILE_256x64_S2 : BC_TILE_128x128_S2;
        if (sk <= 0) {
            long long tiles = (long long)bc_ceil_div(M, kTiles[cfg].bm) * bc_ceil_div(N, kTiles[cfg].bn);
            sk = 1;
            if (tiles < num_cu() * 6 / 10 && nk >= 8) sk = (int)std::max<long long>(1, std::min<long long>(std::min<long long>((num_cu() + tiles - 1) / tiles, nk / 4), 8));
        }
    } else if (cfg <= BC_TILE_AUTO || cfg >= BC_TILE_COUNT || sk <= 0) {
        double best = 1e30;
        int best_cfg = BC_TILE_128x128_S2, best_sk = 1;
        const int sks[] = {1, 2, 3, 4, 6, 8, 12};
        for (int c = 1; c < BC_TILE_COUNT; ++c) {
            if (cfg > BC_TILE_AUTO && cfg < BC_TILE_COUNT && c != cfg) continue;
            for (int s : sks) {
                if (sk > 0 && s != sk) continue;
                if (s > 1 && nk / s < 4) continue;
                double us = model_us(kTiles[c], M, N, K, s);
                if (us < best) { best = us; best_cfg = c; best_sk = s; }
            }
        }
        cfg = best_cfg;
        if (sk <= 0) sk = best_sk;
    }
    if (sk > nk) sk = nk;
    *tile_cfg = cfg;
    *splitk = sk;
    *bm = kTiles[cfg].bm;
    *bn = kTiles[cfg].bn;
    return 0;
}

extern "C" int bc_conv_halo_eligible(int Cin, int C1, int N, int Hin, int Win, int Hout, int Wout, int stride) {
    BcGemm p = {};
    p.a_mode = BC_A_CONV3X3; p.Cin = Cin; p.C1 = C1; p.N = N; p.Hin = p.Hv = Hin; p.Win = p.Wv = Win; p.Hout = Hout; p.Wout = Wout;
    p.stride = stride;
    p.A2 = C1 > 0 ? reinterpret_cast<const bc_half*>(&p) : nullptr;       // (only tested for non-null)
    return bc_conv_halo_ok(p);
}

extern "C" int bc_conv_halo_max_chunks(void) { return bc_conv_halo_max_chunks_impl(); }

// Timing probe of bc_plan_run_timed_kernels: an event recorded between a split-K GEMM's main kernel and its reducer.
static thread_local hipEvent_t tl_probe = nullptr;
static thread_local bool tl_probe_hit = false;
void bc_gemm_set_probe(hipEvent_t e) { tl_probe = e; tl_probe_hit = false; }
bool bc_gemm_probe_hit() { return tl_probe_hit; }

extern "C" int bc_gemm(const BcGemm* pp, bc_stream stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    BC_CHECK_ARG(pp != nullptr, "bc_gemm: null params");
    GemmArgs g;
    g.p = *pp;
    BcGemm& p = g.p;
    BC_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "bc_gemm: bad dims M=%d N=%d K=%d", p.M, p.N, p.K);
    BC_CHECK_ARG(p.A && p.W && p.C, "bc_gemm: null A/W/C");
    BC_CHECK_ARG(p.alpha_bstride == 0 || (p.alpha_bstride > 0 && p.alpha_dev && p.rows_per_batch > 0 && p.out_mode != BC_OUT_F16_T),
                 "bc_gemm: alpha_bstride needs alpha_dev, rows_per_batch > 0 and a row-major output");
    BC_CHECK_ARG(p.K % 8 == 0 && p.ldw % 8 == 0 && p.ldw >= p.K, "bc_gemm: K=%d ldw=%d must be multiples of 8, ldw>=K", p.K, p.ldw);
    BC_CHECK_ARG(((uintptr_t)p.A % 16 == 0) && ((uintptr_t)p.W % 16 == 0), "bc_gemm: A/W must be 16-byte aligned");
    if (p.a_mode == BC_A_CONV3X3) {
        BC_CHECK_ARG(p.Cin > 0 && p.Cin % 8 == 0 && p.K == 9 * p.Cin, "bc_gemm: conv needs Cin%%8==0 and K==9*Cin (Cin=%d K=%d)", p.Cin, p.K);
        BC_CHECK_ARG(p.stride == 1 || p.stride == 2, "bc_gemm: conv stride must be 1 or 2");
        BC_CHECK_ARG(p.Hin > 0 && p.Win > 0 && p.Hout > 0 && p.Wout > 0, "bc_gemm: conv geometry missing");
        if (p.Hv <= 0) p.Hv = p.Hin;
        if (p.Wv <= 0) p.Wv = p.Win;
        BC_CHECK_ARG(p.M % (p.Hout * p.Wout) == 0, "bc_gemm: conv M=%d not a multiple of Hout*Wout", p.M);
        const int pads = p.conv_nopad_lo ? 1 : 2;
        BC_CHECK_ARG(p.Hout == (p.Hv + pads - 3) / p.stride + 1 && p.Wout == (p.Wv + pads - 3) / p.stride + 1,
                     "bc_gemm: conv output size %dx%d inconsistent with input %dx%d stride %d", p.Hout, p.Wout, p.Hv, p.Wv, p.stride);
        BC_CHECK_ARG(p.A2 == nullptr || p.tile_cfg == BC_TILE_HALO || p.tile_cfg == BC_TILE_WREG, "bc_gemm: conv mode takes a single source (except BC_TILE_HALO / BC_TILE_WREG)");
        p.rows_per_batch = p.Hout * p.Wout;
        if (p.out_w <= 0) p.out_w = p.Wout;
        g.fast_k = (p.Cin % BK == 0);
    } else {
        BC_CHECK_ARG(p.a_mode == BC_A_DENSE, "bc_gemm: unknown a_mode %d", p.a_mode);
        BC_CHECK_ARG(p.lda % 8 == 0, "bc_gemm: lda=%d must be a multiple of 8", p.lda);
        if (p.A2) {
            BC_CHECK_ARG(p.C1 > 0 && p.C1 % 8 == 0 && p.C1 < p.K && p.lda2 % 8 == 0 && (uintptr_t)p.A2 % 16 == 0,
                         "bc_gemm: bad concat split C1=%d K=%d lda2=%d", p.C1, p.K, p.lda2);
            BC_CHECK_ARG(p.lda >= p.C1 && p.lda2 >= p.K - p.C1, "bc_gemm: concat strides too small");
        } else {
            BC_CHECK_ARG(p.lda >= p.K, "bc_gemm: lda=%d < K=%d", p.lda, p.K);
        }
        g.fast_k = 1;
    }
    if (p.rows_per_batch <= 0) p.rows_per_batch = p.M;
    if (p.out_w <= 0) p.out_w = p.rows_per_batch;
    if (p.splitk < 1) p.splitk = 1;
    g.n_out = p.N;
    if (p.act == BC_ACT_GEGLU) {
        BC_CHECK_ARG(p.N % 64 == 0, "bc_gemm: GEGLU needs N%%64==0 (got %d)", p.N);
        g.n_out = p.N / 2;
    }
    if (p.sm_group > 0) {                             // softmax epilogue: sm_keep of every sm_group columns are written
        BC_CHECK_ARG(p.sm_keep > 0 && p.sm_keep <= p.sm_group && p.N % p.sm_group == 0, "bc_gemm: bad sm_group / sm_keep (%d / %d, N=%d)", p.sm_group, p.sm_keep, p.N);
        g.n_out = p.N / p.sm_group * p.sm_keep;
    }
    if (p.R) BC_CHECK_ARG(p.ldr >= g.n_out, "bc_gemm: ldr too small");
    if (p.R2) BC_CHECK_ARG(p.ldr2 >= g.n_out && p.r2_bmod > 0, "bc_gemm: bad R2 params");
    if (p.rowvec) BC_CHECK_ARG(p.ld_rowvec >= p.N, "bc_gemm: ld_rowvec too small");
    const bool gw = bc_gemm_wreg_nt(p.tile_cfg) != 0;
    BC_CHECK_ARG(gw || !p.ln_colsum, "bc_gemm: ln_colsum needs a BC_TILE_GW* configuration");
    BC_CHECK_ARG(gw || p.tile_cfg == BC_TILE_G256 || !p.C_t, "bc_gemm: C_t needs a BC_TILE_GW* or the BC_TILE_G256 configuration");
    BC_CHECK_ARG(gw || (!p.w_bstride && !p.vec_bstride && !p.sm_group), "bc_gemm: w_bstride / vec_bstride / sm_group need a BC_TILE_GW* configuration");
    if (p.out_mode == BC_OUT_F16_T) {
        BC_CHECK_ARG(p.M % p.rows_per_batch == 0 && p.ldc >= p.rows_per_batch, "bc_gemm: transposed output needs M%%rows_per_batch==0, ldc>=rows_per_batch");
    } else {
        BC_CHECK_ARG(p.ldc >= (p.C_t ? p.n_t0 : g.n_out), "bc_gemm: ldc=%d < n_out=%d", p.ldc, g.n_out);
    }
    if (gw) {
        BC_CHECK_ARG(bc_gemm_wreg_ok(p, p.tile_cfg), "bc_gemm: BC_TILE_GW* needs dense A, M%%64==0, N%%(64 NT)==0, K%%320==0 (C1%%320==0), fp16 row-major "
                     "output, no split-K / row vector / affine table (a_tot1: one source, groups | K <= 2560, no activation) (M=%d N=%d K=%d C1=%d)", p.M, p.N, p.K, p.C1);
        g.nk = p.K / 32; g.kt_per_split = g.nk; p.splitk = 1;
        g.div_rpb = make_fastdiv((unsigned)p.rows_per_batch);
        g.div_outw = make_fastdiv((unsigned)p.out_w);
        g.div_wout = make_fastdiv(1u);
        g.cfg = p.tile_cfg; g.bm = 64; g.bn = 64 * bc_gemm_wreg_nt(p.tile_cfg);
        auto al16 = [](const void* q) { return ((uintptr_t)q % 16) == 0; };
        g.vec_epilogue = g.n_out % 8 == 0 && p.ldc % 8 == 0 && al16(p.C) && (!p.R || (p.ldr % 8 == 0 && al16(p.R))) &&
                         (!p.R2 || (p.ldr2 % 8 == 0 && al16(p.R2))) && (!p.C_t || al16(p.C_t));
        BC_CHECK_ARG(g.vec_epilogue, "bc_gemm: BC_TILE_GW* needs 16-byte aligned C / R / R2 / C_t and widths %% 8 == 0");
        g.vec_transposed = 0; g.nband = 1;
        return bc_gemm_wreg_launch(g, stream);
    }
    if (p.tile_cfg == BC_TILE_G256) {
        BC_CHECK_ARG(bc_gemm256_ok(p), "bc_gemm: BC_TILE_G256 needs dense A (C1%%128==0), M%%256==0, N%%256==0, K%%128==0, no split-K, fp16 row-major "
                     "or transposed (bias / alpha only) output, n_t0%%256==0 with C_t (M=%d N=%d K=%d out_mode=%d)", p.M, p.N, p.K, p.out_mode);
        g.nk = p.K / BK; g.kt_per_split = g.nk; p.splitk = 1;
        g.div_rpb = make_fastdiv((unsigned)p.rows_per_batch);
        g.div_outw = make_fastdiv((unsigned)p.out_w);
        g.div_wout = make_fastdiv(1u);
        g.cfg = BC_TILE_G256; g.bm = 256; g.bn = 256;
        auto al16 = [](const void* q) { return ((uintptr_t)q % 16) == 0; };
        g.vec_epilogue = g.n_out % 8 == 0 && p.ldc % 8 == 0 && al16(p.C) && (!p.R || (p.ldr % 8 == 0 && al16(p.R))) &&
                         (!p.R2 || (p.ldr2 % 8 == 0 && al16(p.R2))) && (!p.rowvec || p.ld_rowvec % 8 == 0) && (!p.C_t || al16(p.C_t)) &&
                         (!p.A2 || al16(p.A2));
        BC_CHECK_ARG(g.vec_epilogue, "bc_gemm: BC_TILE_G256 needs 16-byte aligned C / C_t / R / R2 / A2 and widths %% 8 == 0");
        BC_CHECK_ARG(!p.gn_tot || (p.rows_per_batch % 256 == 0 && p.M % p.rows_per_batch == 0),
                     "bc_gemm: BC_TILE_G256 with GroupNorm statistics needs rows_per_batch%%256==0 (a tile's rows inside one image)");
        g.vec_transposed = p.out_mode == BC_OUT_F16_T; g.nband = 0;
        return bc_gemm256_launch(g, stream);
    }
    const bool wreg = p.tile_cfg == BC_TILE_WREG;
    const bool halo = p.tile_cfg == BC_TILE_HALO || wreg;
    if (halo) {
        if (p.lda <= 0) p.lda = p.A2 ? p.C1 : p.Cin;          // pixel stride of A: 0 = its channel count (a wider NHWC view passes its own)
        if (p.A2 && p.lda2 <= 0) p.lda2 = p.Cin - p.C1;
        BC_CHECK_ARG(p.lda >= (p.A2 ? p.C1 : p.Cin) && p.lda % 8 == 0, "bc_gemm: BC_TILE_HALO lda=%d must be >= the channel count and a multiple of 8", p.lda);
        BC_CHECK_ARG(!p.A2 || (p.lda2 >= p.Cin - p.C1 && p.lda2 % 8 == 0), "bc_gemm: BC_TILE_HALO lda2=%d must be >= Cin - C1 and a multiple of 8", p.lda2);
        BC_CHECK_ARG(bc_conv_halo_ok(p), "bc_gemm: BC_TILE_HALO needs a 3x3 stride-1 pad-1 convolution with Cin%%64==0, N%%160==0, "
                                         "Wout%%16==0, Hout%%8==0 (Cin=%d N=%d %dx%d)", p.Cin, p.N, p.Hout, p.Wout);
        BC_CHECK_ARG(!p.a_affine || p.a_act == BC_ACT_NONE || p.a_act == BC_ACT_SILU, "bc_gemm: a_act must be NONE or SILU");
    } else {
        BC_CHECK_ARG(p.a_affine == nullptr && p.a_tot1 == nullptr, "bc_gemm: the fused GroupNorm prologue is only available on BC_TILE_HALO");
    }
    g.nk = bc_ceil_div(p.K, BK);
    if (halo) g.nk = p.Cin / BK;                    // split-K counts 64-channel chunks (each covers the nine taps)
    if (p.splitk > g.nk) p.splitk = g.nk;
    if (p.splitk > 1) BC_CHECK_ARG(p.slab != nullptr, "bc_gemm: splitk=%d needs a slab", p.splitk);
    g.kt_per_split = bc_ceil_div(g.nk, p.splitk);
    p.splitk = bc_ceil_div(g.nk, g.kt_per_split);   // no empty splits
    g.div_rpb = make_fastdiv((unsigned)p.rows_per_batch);
    g.div_outw = make_fastdiv((unsigned)p.out_w);
    g.div_wout = make_fastdiv((unsigned)(p.a_mode == BC_A_CONV3X3 ? p.Wout : 1));

    // ---- resolve tile configuration (explicit from the caller's plan, or heuristic) ----
    bool fast_ok = (p.K % BK == 0);
    if (p.a_mode == BC_A_CONV3X3) {
        bool ups = (p.Hv != p.Hin) || (p.Wv != p.Win);
        fast_ok = fast_ok && (p.Cin % BK == 0) && (!ups || (p.Hv == 2 * p.Hin && p.Wv == 2 * p.Win && p.stride == 1));
    } else if (p.A2) {
        fast_ok = fast_ok && (p.C1 % BK == 0);
    }
    static const bool force_generic = getenv("BC_GEMM_GENERIC") != nullptr;
    static const int force_tile = getenv("BC_GEMM_TILE") ? atoi(getenv("BC_GEMM_TILE")) : 0;
    if (force_generic) fast_ok = false;
    if (halo) {
        g.cfg = wreg ? BC_TILE_WREG : BC_TILE_HALO; g.bm = 128; g.bn = 160;
    } else {
        int cfg = (force_tile > 0 && force_tile < BC_TILE_COUNT) ? force_tile : p.tile_cfg;
        int sk = p.splitk, bm = 0, bn = 0;
        int rc0 = bc_gemm_plan(p.M, p.N, p.K, fast_ok ? 1 : 0, &cfg, &sk, &bm, &bn);
        if (rc0) return rc0;
        g.cfg = cfg; g.bm = bm; g.bn = bn;
    }
    auto aligned16 = [](const void* q) { return ((uintptr_t)q % 16) == 0; };
    g.vec_epilogue = p.out_mode == BC_OUT_F16 && g.n_out % 8 == 0 && p.ldc % 8 == 0 && aligned16(p.C) &&
                     (!p.R || (p.ldr % 8 == 0 && aligned16(p.R))) && (!p.R2 || (p.ldr2 % 8 == 0 && aligned16(p.R2)));
    g.vec_transposed = p.out_mode == BC_OUT_F16_T && p.rows_per_batch % 8 == 0 && p.ldc % 8 == 0 && aligned16(p.C) &&
                       p.act == BC_ACT_NONE && !p.rowvec && !p.colscale && !p.R && !p.R2;
    if (p.gn_tot) {
        // (a workgroup's rows must belong to one image: its statistics go to that image's totals)
        const int slab_rows = p.splitk > 1 ? SK_ROWS : g.bm;
        BC_CHECK_ARG((fast_ok || halo || p.splitk > 1) && g.vec_epilogue && p.N % 4 == 0 && p.rows_per_batch % slab_rows == 0 &&
                         p.M % p.rows_per_batch == 0,
                     "bc_gemm: fused GroupNorm statistics need the fast path or split-K, fp16 row-major output and rows_per_batch%%%d==0", slab_rows);
    }
    int rc = wreg ? bc_conv_wreg_launch(g, stream) : halo ? bc_conv_halo_launch(g, stream) : fast_ok ? bc_gemm_fast_try(g, stream) : -1;
    if (rc > 0) return rc;
    if (rc < 0) {
        BC_CHECK_ARG(!p.gn_tot || p.splitk > 1, "bc_gemm: fused GroupNorm statistics are only produced by the fast path or the split-K reducer");
        rc = (g.bn == 64) ? launch_gemm<256, 64, 4, 1>(g, stream) : launch_gemm<128, 128, 2, 2>(g, stream);
        if (rc) return rc;
    }
    if (p.splitk > 1) {
        if (tl_probe) {                                       // (profiling replays: where the main kernel ends and the reducer starts)
            BC_CHECK_HIP(hipEventRecord(tl_probe, stream));
            tl_probe_hit = true;
        }
        if (g.vec_epilogue && p.N % 4 == 0) {
            hipLaunchKernelGGL(splitk_reduce_vec_kernel, dim3(bc_ceil_div(g.n_out, 64), bc_ceil_div(p.M, SK_ROWS)), dim3(256), 0,
                               stream, g);
        } else {
            long long total = (long long)p.M * g.n_out;
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3(bc_ceil_div(total, 256)), dim3(256), 0, stream, g);
        }
        BC_CHECK_LAUNCH();
    }
    return 0;
}
