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
    const int m0 = blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;
    const int split = blockIdx.z;

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
            a_iy0[i] = ok ? oy * p.stride - 1 : -(1 << 20);
            a_ix0[i] = ox * p.stride - 1;
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
    float alpha = p.alpha;
    if (p.alpha_dev) alpha *= p.alpha_dev[p.alpha_idx ? *p.alpha_idx : 0];
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
    epilogue_store(g, v, m, n, alpha);
}

template <int BM, int BN, int WM, int WN>
int launch_gemm(const GemmArgs& g, hipStream_t stream) {
    const BcGemm& p = g.p;
    dim3 grid(bc_ceil_div(p.N, BN), bc_ceil_div(p.M, BM), p.splitk);
    dim3 block(64 * WM * WN);
    size_t lds = 2 * (BM + BN) * 128;
    static bool attr_set = false;
    if (!attr_set) {
        BC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<BM, BN, WM, WN>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN>), grid, block, lds, stream, g);
    BC_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" int bc_gemm_tile_rows(int N) {
    int n128 = bc_ceil_div(N, 128) * 128;
    return ((double)n128 / N > 1.10) ? 256 : 128;
}

extern "C" int bc_gemm(const BcGemm* pp, bc_stream stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    BC_CHECK_ARG(pp != nullptr, "bc_gemm: null params");
    GemmArgs g;
    g.p = *pp;
    BcGemm& p = g.p;
    BC_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "bc_gemm: bad dims M=%d N=%d K=%d", p.M, p.N, p.K);
    BC_CHECK_ARG(p.A && p.W && p.C, "bc_gemm: null A/W/C");
    BC_CHECK_ARG(p.K % 8 == 0 && p.ldw % 8 == 0 && p.ldw >= p.K, "bc_gemm: K=%d ldw=%d must be multiples of 8, ldw>=K", p.K, p.ldw);
    BC_CHECK_ARG(((uintptr_t)p.A % 16 == 0) && ((uintptr_t)p.W % 16 == 0), "bc_gemm: A/W must be 16-byte aligned");
    if (p.a_mode == BC_A_CONV3X3) {
        BC_CHECK_ARG(p.Cin > 0 && p.Cin % 8 == 0 && p.K == 9 * p.Cin, "bc_gemm: conv needs Cin%%8==0 and K==9*Cin (Cin=%d K=%d)", p.Cin, p.K);
        BC_CHECK_ARG(p.stride == 1 || p.stride == 2, "bc_gemm: conv stride must be 1 or 2");
        BC_CHECK_ARG(p.Hin > 0 && p.Win > 0 && p.Hout > 0 && p.Wout > 0, "bc_gemm: conv geometry missing");
        if (p.Hv <= 0) p.Hv = p.Hin;
        if (p.Wv <= 0) p.Wv = p.Win;
        BC_CHECK_ARG(p.M % (p.Hout * p.Wout) == 0, "bc_gemm: conv M=%d not a multiple of Hout*Wout", p.M);
        BC_CHECK_ARG(p.Hout == (p.Hv + 2 - 3) / p.stride + 1 && p.Wout == (p.Wv + 2 - 3) / p.stride + 1,
                     "bc_gemm: conv output size %dx%d inconsistent with input %dx%d stride %d", p.Hout, p.Wout, p.Hv, p.Wv, p.stride);
        BC_CHECK_ARG(p.A2 == nullptr, "bc_gemm: conv mode takes a single source");
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
    if (p.R) BC_CHECK_ARG(p.ldr >= g.n_out, "bc_gemm: ldr too small");
    if (p.R2) BC_CHECK_ARG(p.ldr2 >= g.n_out && p.r2_bmod > 0, "bc_gemm: bad R2 params");
    if (p.rowvec) BC_CHECK_ARG(p.ld_rowvec >= p.N, "bc_gemm: ld_rowvec too small");
    if (p.out_mode == BC_OUT_F16_T) {
        BC_CHECK_ARG(p.M % p.rows_per_batch == 0 && p.ldc >= p.rows_per_batch, "bc_gemm: transposed output needs M%%rows_per_batch==0, ldc>=rows_per_batch");
    } else {
        BC_CHECK_ARG(p.ldc >= g.n_out, "bc_gemm: ldc=%d < n_out=%d", p.ldc, g.n_out);
    }
    g.nk = bc_ceil_div(p.K, BK);
    if (p.splitk > g.nk) p.splitk = g.nk;
    if (p.splitk > 1) BC_CHECK_ARG(p.slab != nullptr, "bc_gemm: splitk=%d needs a slab", p.splitk);
    g.kt_per_split = bc_ceil_div(g.nk, p.splitk);
    p.splitk = bc_ceil_div(g.nk, g.kt_per_split);   // no empty splits
    g.div_rpb = make_fastdiv((unsigned)p.rows_per_batch);
    g.div_outw = make_fastdiv((unsigned)p.out_w);
    g.div_wout = make_fastdiv((unsigned)(p.a_mode == BC_A_CONV3X3 ? p.Wout : 1));

    // tile choice: 128x128 unless N would be padded by >10 % (N = 320 -> 5 x 64 columns)
    int n128 = bc_ceil_div(p.N, 128) * 128;
    g.narrow = (double)n128 / p.N > 1.10;
    auto aligned16 = [](const void* q) { return ((uintptr_t)q % 16) == 0; };
    g.vec_epilogue = p.out_mode == BC_OUT_F16 && g.n_out % 8 == 0 && p.ldc % 8 == 0 && aligned16(p.C) &&
                     (!p.R || (p.ldr % 8 == 0 && aligned16(p.R))) && (!p.R2 || (p.ldr2 % 8 == 0 && aligned16(p.R2)));
    if (p.gn_part) {
        int bm = g.narrow ? 256 : 128;
        BC_CHECK_ARG(p.splitk == 1 && g.vec_epilogue && p.K % BK == 0 && p.rows_per_batch % bm == 0 && p.M % p.rows_per_batch == 0,
                     "bc_gemm: fused GroupNorm partials need splitk==1, fp16 row-major output, K%%64==0 and rows_per_batch%%%d==0", bm);
    }
    static const bool force_generic = getenv("BC_GEMM_GENERIC") != nullptr;
    int rc = force_generic ? -1 : bc_gemm_fast_try(g, stream);
    if (rc > 0) return rc;
    if (rc < 0) {
        BC_CHECK_ARG(!p.gn_part, "bc_gemm: fused GroupNorm partials are only produced by the fast path");
        rc = g.narrow ? launch_gemm<256, 64, 4, 1>(g, stream) : launch_gemm<128, 128, 2, 2>(g, stream);
        if (rc) return rc;
    }
    if (p.splitk > 1) {
        long long total = (long long)p.M * g.n_out;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(bc_ceil_div(total, 256)), dim3(256), 0, stream, g);
        BC_CHECK_LAUNCH();
    }
    return 0;
}
