// Small-M GEMM with the WEIGHTS STREAMED STRAIGHT INTO VGPRs (gfx950, v_mfma_f32_16x16x32_f16): the projections of the 1280-channel
// levels of both nets (16 x 32, 8 x 16, mid: M = 128 ... 1024 token rows), BC_TILE_GW64x128 / GW64x256 / GW64x320.
//
// Reference ops (diffusers/src/diffusers/models/): attention_processor.py:2191-2224 (to_q / to_k / to_v / to_out), attention.py:447-541
// (norm1 / norm2 / norm3 in front of them, the residual adds behind them), activations.py:113-123 + attention.py:1161-1167 (GEGLU
// feed-forward), transformers/transformer_2d.py:484,521 (proj_in / proj_out), resnet.py:368 (conv_shortcut),
// blobctrl/models/blobnet.py:860-864,921-924 (zero-convs).
//
// Why another GEMM kernel.  At M <= 1024 a projection is a few GFLOP: the LDS-DMA tiles of gemm_fast.hip run their k-loop at the
// per-CU LDS-DMA rate (~22 B/clk) and spend as long again outside it.  Same recipe as conv_wreg.hip / rowchain.hip:
//   * a workgroup (4 waves) owns 64 token rows x BN = 64 NT columns; wave w owns the 16 NT columns [16 NT w, 16 NT (w + 1)) for ALL 64
//     rows, so no weight is shared between waves and the weights stream L2/HBM -> VGPR from a per-wave fragment stream
//     (bc_gemm_wreg_pack: lane l of a 1-KiB fragment holds W[n + (l & 15)][k + 8 (l >> 4) .. + 8]; k-step outermost) through a register
//     ring that stays RING fragments ahead of the MFMAs - no LDS, no barrier inside the k-loop;
//   * the activation rows are the operand the four waves SHARE: they go through LDS as X = [k-step][64 rows][32 k] (64-byte rows, 16-byte
//     chunks XOR-swizzled: every ds_read_b128 fragment read conflict-free) in chunks of 320 k, double-buffered; the next chunk's rows are
//     in flight in registers while the current one is multiplied; one barrier per chunk;
//   * the product is swapped (D^T = W . X^T): a lane holds 4 consecutive output channels of one row - one 16-byte LDS store parks them
//     in the fp32 epilogue tile; the row-major pass is the shared 8-column epilogue of the GEMM family (epi8_store: bias, GEGLU,
//     scale, residual, BlobNet right-half residual, GroupNorm partials);
//   * LayerNorm in front of the projection is FOLDED (BcGemm.ln_colsum): the host scales W by gamma, the rows enter raw, their
//     statistics are accumulated (v_dot2_f32_f16) while they are staged, and the epilogue applies
//         rstd_r (acc[r][n] - mean_r colsum[n]) + bias'[n],   colsum[n] = sum_k W'[n][k],  bias' = bias + W beta:
//     no LayerNorm launch, no normalised activation in HBM;
//   * q | k | V^T in one launch (BcGemm.C_t): column tiles from n_t0 on are written transposed [B][N - n_t0][ldc_t].
//   * per-image weight streams (BcGemm.w_bstride) and a row softmax over the workgroup's 128 columns in the epilogue (BcGemm.sm_group: a
//     64 x 128 workgroup = the keys of one head, padded): the two launches of a cross-attention whose prompt is folded into the weights
//     once per edit (bc_ctx_fold below) - scores = LN(x) QK^T never leave the launch.
// XCD placement: consecutive workgroups of an XCD share a COLUMN tile (all its row blocks), so an XCD's L2 fetches each weight
// stream once.
#include <stdlib.h>
#include <type_traits>
#include "gemm_common.h"

using namespace bcg;

namespace {

constexpr int GW_BM = 64;               // rows per workgroup
constexpr int GW_KC = 10;               // k-steps (of 32) per LDS chunk: 320 k
constexpr int GW_XBUF = GW_KC * 4096;   // one operand chunk: [10][64 rows][64 B]
constexpr int GW_NPF = 10;              // 16-byte pieces a thread stages per chunk (2 rows x 5)

typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ h16x8 as_h8(uint4 v) { return __builtin_bit_cast(h16x8, v); }

__device__ __forceinline__ void lds_barrier() {
    // LDS visibility only: the weight ring's global loads stay in flight across it (a __syncthreads() would drain them)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int R>
struct WRing {
    uint4 f[R];
    const uint4* p;          // this lane's next fragment to LOAD
};

// X operand image: byte offset of the 16-byte chunk kc (0..3) of `row` in k-step s (rowchain.hip's layout)
__device__ __forceinline__ int x_off(int s, int row, int kc) { return s * 4096 + row * 64 + ((kc ^ ((0 - ((row & 15) >> 2)) & 3)) << 4); }

// One chunk: acc[t][mt] += W-tile t (16 channels) x rows-tile mt (16 rows) over GW_KC k-steps.  Issue order per k-step pinned with
// sched_barrier as in rowchain.hip (left alone hipcc sinks the ring refills and the prefetch distance collapses): refill the slots the
// PREVIOUS step consumed, read the NEXT step's operand fragments, then this step's MFMAs run under both.  (NT * GW_KC) % R == 0, so the
// ring position is 0 at the top of every chunk.
template <int NT, int R>
__device__ __forceinline__ void gemm_chunk(f32x4v (&acc)[NT][4], WRing<R>& r, const char* xb) {
    h16x8 xq[2][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) xq[0][mt] = *reinterpret_cast<const h16x8*>(xb + mt * 1024);
#pragma unroll
    for (int s = 0; s < GW_KC; ++s) {
        if (s > 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t) r.f[((s - 1) * NT + t) % R] = r.p[t * 64];
            r.p += NT * 64;
        }
        if (s + 1 < GW_KC) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) xq[(s + 1) & 1][mt] = *reinterpret_cast<const h16x8*>(xb + (s + 1) * 4096 + mt * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const h16x8 w = as_h8(r.f[(s * NT + t) % R]);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, xq[s & 1][mt], acc[t][mt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) r.f[((GW_KC - 1) * NT + t) % R] = r.p[t * 64];
    r.p += NT * 64;
    __builtin_amdgcn_sched_barrier(0);
}

// fp32 epilogue tile [64][BN]: float4 slot (row, c0) lives at column c0 ^ ((row & 7) << 2) - the 8 lanes of a ds_write_b128 service
// group hold 8 different rows of one column block and land on 8 different 16-byte bank groups.
template <int BN>
__device__ __forceinline__ int tile_off(int row, int c0) { return row * BN + (c0 ^ ((row & 7) << 2)); }

// SM: the instantiation with the softmax epilogue (BcGemm.sm_group) - its own kernel, so that the projections' one stays as it was
template <int NT, int R, bool SM = false>
__global__ __launch_bounds__(256, 2) void gemm_wreg_kernel(const GemmArgs g) {
    constexpr int BN = 64 * NT;
    static_assert((NT * GW_KC) % R == 0, "a chunk must consume a whole number of rings");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const BcGemm& p = g.p;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, q = lane >> 4;
    const int nrb = p.M / GW_BM;
    const int wg = bc_xcd_remap(blockIdx.x, gridDim.x);
    const int jt = wg / nrb;                          // column tile (slow index: an XCD's workgroups share it)
    const int m0 = (wg - jt * nrb) * GW_BM, n0 = jt * BN;
    const int KS = p.K >> 5;
    const int nchunk = KS / GW_KC;

    // ---- GroupNorm in front of the projection (Transformer2D's `norm` -> proj_in, transformer_2d.py:481-484): the finalize runs here, from
    // the statistics totals (bc_common.h), into a per-channel (a, b) table in LDS that the row staging below applies: no GroupNorm pass,
    // no normalised activation in HBM.  Before the weight ring is filled, so that nothing of the ring is drained by its waits.
    constexpr int MAIN_BYTES = 2 * GW_XBUF > GW_BM * BN * 4 + GW_BM * 8 ? 2 * GW_XBUF : GW_BM * BN * 4 + GW_BM * 8;     // = launch_gw's LDS
    const bool gn = p.a_tot1 != nullptr;
    float* const abt = reinterpret_cast<float*>(smem + MAIN_BYTES);            // [K][2]
    if (gn) {
        const int bimg = (int)fdiv((unsigned)m0, g.div_rpb);
        double* scr = reinterpret_cast<double*>(smem);                           // [K][2] (the operand buffers are not in use yet)
        float* stt = reinterpret_cast<float*>(smem + p.K * 16);                  // [groups][2]
        for (int c = tid; c < p.K; c += 256) {
            double s_, q_;
            bc_gn_tot_read(p.a_tot1 + ((size_t)bimg * p.K + c) * BC_GN_TOT_WORDS, s_, q_);
            scr[c * 2] = s_;
            scr[c * 2 + 1] = q_;
        }
        __syncthreads();
        const int cpg = p.K / p.a_groups, sub = tid & 7;
        for (int gi = tid >> 3; gi < p.a_groups; gi += 32) {
            double s_ = 0.0, q_ = 0.0;
            for (int cj = sub; cj < cpg; cj += 8) {
                s_ += scr[(gi * cpg + cj) * 2];
                q_ += scr[(gi * cpg + cj) * 2 + 1];
            }
#pragma unroll
            for (int o = 4; o > 0; o >>= 1) {
                s_ += __shfl_xor(s_, o);
                q_ += __shfl_xor(q_, o);
            }
            const double n = (double)g.div_rpb.d * cpg;
            const double mean = s_ / n;
            double var = q_ / n - mean * mean;
            if (var < 0.0) var = 0.0;
            if (sub == 0) {
                stt[gi * 2] = (float)mean;
                stt[gi * 2 + 1] = (float)(1.0 / sqrt(var + (double)p.a_eps));
            }
        }
        __syncthreads();
        for (int c = tid; c < p.K; c += 256) {
            const int gi = c / cpg;
            const float av = stt[gi * 2 + 1] * p.a_gamma[c];
            abt[c * 2] = av;
            abt[c * 2 + 1] = p.a_beta[c] - stt[gi * 2] * av;
        }
        __syncthreads();
    }

    WRing<R> ring;
    const int wimg = p.w_bstride ? (int)fdiv((unsigned)m0, g.div_rpb) : 0;       // per-image weights: this row block's image
    ring.p = reinterpret_cast<const uint4*>(p.W + (size_t)wimg * p.w_bstride) + ((size_t)(jt * 4 + wave) * KS * NT) * 64 + lane;
#pragma unroll
    for (int i = 0; i < R; ++i) ring.f[i] = ring.p[i * 64];
    ring.p += R * 64;

    // ---- activation staging: thread (v8 = tid & 7, row = tid >> 3) moves the 16-byte pieces v8 + 8 k (k < 5) of the rows row, row + 32 of a chunk
    const int srow = tid >> 3, v8 = tid & 7;
    const int xbase = x_off(v8 >> 2, srow, v8 & 3);
    const h16* a1 = reinterpret_cast<const h16*>(p.A) + (size_t)(m0 + srow) * p.lda + v8 * 8;
    const h16* a2 = p.A2 ? reinterpret_cast<const h16*>(p.A2) + (size_t)(m0 + srow) * p.lda2 + v8 * 8 : nullptr;
    const int lda32 = 32 * p.lda, lda32_2 = 32 * p.lda2;
    uint4 pf[GW_NPF];
    auto issue = [&](int c) {
        const int k0 = c * (GW_KC * 32);
        const bool second = a2 != nullptr && k0 >= p.C1;           // (workgroup-uniform: C1 % 320 == 0)
        const h16* src = second ? a2 + (k0 - p.C1) : a1 + k0;
        const int ld32 = second ? lda32_2 : lda32;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            pf[2 * k] = *reinterpret_cast<const uint4*>(src + k * 64);
            pf[2 * k + 1] = *reinterpret_cast<const uint4*>(src + k * 64 + ld32);
        }
    };
    float st[2][2] = {{0.f, 0.f}, {0.f, 0.f}};       // LayerNorm fold: (sum, sum of squares) of this thread's pieces of its two rows
    const bool ln = p.ln_colsum != nullptr;
    auto land = [&](int c) {
        char* xb = smem + (c & 1) * GW_XBUF + xbase;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            float4 a4[4];
            if (gn) {                                   // (a, b) of this piece's 8 channels
                const float4* ab4 = reinterpret_cast<const float4*>(abt + (c * (GW_KC * 32) + (v8 + 8 * k) * 8) * 2);
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4) a4[j4] = ab4[j4];
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint4 d = pf[2 * k + j];
                if (gn) {
                    h16* e = reinterpret_cast<h16*>(&d);
#pragma unroll
                    for (int j4 = 0; j4 < 4; ++j4) {
                        e[2 * j4] = (h16)fmaf((float)e[2 * j4], a4[j4].x, a4[j4].y);
                        e[2 * j4 + 1] = (h16)fmaf((float)e[2 * j4 + 1], a4[j4].z, a4[j4].w);
                    }
                }
                if (ln) {
                    const h16x2 one = {(h16)1.0f, (h16)1.0f};
                    const h16x2* e = reinterpret_cast<const h16x2*>(&d);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        st[j][0] = __builtin_amdgcn_fdot2(e[i], one, st[j][0], false);
                        st[j][1] = __builtin_amdgcn_fdot2(e[i], e[i], st[j][1], false);
                    }
                }
                *reinterpret_cast<uint4*>(xb + k * 2 * 4096 + j * 32 * 64) = d;
            }
        }
    };

    f32x4v acc[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[t][mt] = (f32x4v){0.f, 0.f, 0.f, 0.f};

    issue(0);
    land(0);
    if (nchunk > 1) issue(1);
    lds_barrier();
    const int xfo = m * 64 + ((q ^ ((0 - (m >> 2)) & 3)) << 4);      // this lane's fragment offset inside a (k-step, row-tile) KiB
    for (int c = 0; c < nchunk; ++c) {
        gemm_chunk<NT, R>(acc, ring, smem + (c & 1) * GW_XBUF + xfo);
        if (c + 1 < nchunk) {
            land(c + 1);                              // (the buffer's last readers passed the previous barrier)
            if (c + 2 < nchunk) issue(c + 2);
        }
        lds_barrier();
    }

    // ------------------------------------------------------------------------------------------------ epilogue
    // (every wave is past the last barrier: the operand buffers are free)
    float* tile = reinterpret_cast<float*>(smem);
    float* stat = reinterpret_cast<float*>(smem + GW_BM * BN * 4);     // [64 rows][2] = (mean, rstd) of the LayerNorm fold
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int c0 = 16 * NT * wave + 16 * t + 4 * q;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) *reinterpret_cast<f32x4v*>(tile + tile_off<BN>(16 * mt + m, c0)) = acc[t][mt];
    }
    if (ln) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float s = st[j][0], ss = st[j][1];
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) {
                s += __shfl_xor(s, o);
                ss += __shfl_xor(ss, o);
            }
            if (v8 == 0) {
                const float mean = s / (float)p.K;
                const float var = fmaxf(ss / (float)p.K - mean * mean, 0.f);
                stat[(srow + 32 * j) * 2] = mean;
                stat[(srow + 32 * j) * 2 + 1] = __builtin_amdgcn_rsqf(var + p.ln_eps);
            }
        }
    }
    __syncthreads();

    if constexpr (SM) {
        {
            // row softmax over the tile's 128 columns (a head's padded keys, the first sm_valid real): four threads per row, 32 columns
            // each, maximum and sum through the quad; the first sm_keep columns are written, compacted to sm_keep per head
            const int valid = p.sm_valid, keep = p.sm_keep;
            const int vimg = p.vec_bstride ? (int)fdiv((unsigned)m0, g.div_rpb) : 0;
            const float* csp = ln ? p.ln_colsum + (size_t)vimg * p.vec_bstride + n0 : nullptr;
            const float* bp = p.bias ? p.bias + (size_t)vimg * p.vec_bstride + n0 : nullptr;
            const int row = tid >> 2, c0 = (tid & 3) * 32;
            const float mean = ln ? stat[row * 2] : 0.f, rstd = ln ? stat[row * 2 + 1] : 1.f;
            float v[32];
            float mx = -3.0e38f;
            // (round 6: the thread's 32 column sums and biases as sixteen 16-byte loads up front.  Read one by one inside the loop below - `if (ln)
            //  ... csp[j]`, `if (bp) ... bp[j]` - they were 64 scalar loads, each awaited where it was made: most of this launch's 23 us)
            float4 cs4[8], bb4[8];
#pragma unroll
            for (int j4 = 0; j4 < 8; ++j4) {
                cs4[j4] = make_float4(0.f, 0.f, 0.f, 0.f);
                bb4[j4] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (ln) {
#pragma unroll
                for (int j4 = 0; j4 < 8; ++j4) cs4[j4] = *reinterpret_cast<const float4*>(csp + c0 + j4 * 4);
            }
            if (bp) {
#pragma unroll
                for (int j4 = 0; j4 < 8; ++j4) bb4[j4] = *reinterpret_cast<const float4*>(bp + c0 + j4 * 4);
            }
#pragma unroll
            for (int j4 = 0; j4 < 8; ++j4) {
                const float4 t4 = *reinterpret_cast<const float4*>(tile + tile_off<BN>(row, c0 + j4 * 4));
                const float tv[4] = {t4.x, t4.y, t4.z, t4.w};
                const float cv[4] = {cs4[j4].x, cs4[j4].y, cs4[j4].z, cs4[j4].w}, bv[4] = {bb4[j4].x, bb4[j4].y, bb4[j4].z, bb4[j4].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = c0 + j4 * 4 + e;
                    float x = tv[e];
                    if (ln) x = rstd * (x - mean * cv[e]);
                    if (bp) x += bv[e];
                    x = j < valid ? x : -3.0e38f;
                    v[j4 * 4 + e] = x;
                    mx = fmaxf(mx, x);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 1));
            mx = fmaxf(mx, __shfl_xor(mx, 2));
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                v[j] = (c0 + j) < valid ? exp2f((v[j] - mx) * 1.44269504088896f) : 0.f;
                sum += v[j];
            }
            sum += __shfl_xor(sum, 1);
            sum += __shfl_xor(sum, 2);
            const float inv = 1.f / sum;
            h16* dst = reinterpret_cast<h16*>(p.C) + (size_t)(m0 + row) * p.ldc + jt * keep + c0;
#pragma unroll
            for (int j8 = 0; j8 < 4; ++j8) {
                if (c0 + j8 * 8 < keep) {
                    uint4 outraw;
                    h16* o = reinterpret_cast<h16*>(&outraw);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (h16)(v[j8 * 8 + e] * inv);
                    bc_st16(dst + j8 * 8, outraw);
                }
            }
            return;
        }
    }
    if (p.C_t && n0 >= p.n_t0) {
        // transposed output (V^T for the attention kernel): thread = (column, 8 consecutive tokens) -> one 16-byte store
        const int b = (int)fdiv((unsigned)m0, g.div_rpb);
        const int pix0 = m0 - b * (int)g.div_rpb.d;
        const int nt_out = p.N - p.n_t0;
        for (int idx = tid; idx < BN * 8; idx += 256) {
            const int col = idx >> 3, mc = idx & 7;
            const int n = n0 + col;
            const float bias = p.bias ? p.bias[n] : 0.f;
            const float cs = ln ? p.ln_colsum[n] : 0.f;
            uint4 outraw;
            h16* o = reinterpret_cast<h16*>(&outraw);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = mc * 8 + j;
                float v = tile[tile_off<BN>(row, col & ~3) + (col & 3)];
                if (ln) v = stat[row * 2 + 1] * (v - stat[row * 2] * cs);
                o[j] = (h16)(v + bias);
            }
            bc_st16(reinterpret_cast<h16*>(p.C_t) + ((size_t)b * nt_out + (n - p.n_t0)) * p.ldc_t + pix0 + mc * 8, outraw);
        }
        return;
    }

    // ---- row-major pass, 8 output columns per thread (gemm_fast.hip's, with the LayerNorm fold in front of the bias) ----
    const float alpha = scalar_alpha(p);
    const bool geglu = p.act == BC_ACT_GEGLU;
    const int TSO = geglu ? BN / 2 : BN;            // output columns of this block
    const int CPR = TSO / 8;                        // 8-column chunks per tile row
    const int RG = 256 / CPR;                       // rows one pass of the block covers
    const int col8 = tid % CPR, rg = tid / CPR;
    const bool active = rg < RG;
    const int c_out = col8 * 8;
    const int n_first = (geglu ? n0 / 2 : n0) + c_out;
    const int cv = geglu ? (c_out >> 5) * 64 + (c_out & 31) : c_out;      // tile column of the (value) accumulators
    Cols8 cols;
    cols8_init(g, cols, n_first, n0 + cv, geglu, alpha);
    float csv[8], csg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        csv[j] = ln ? p.ln_colsum[n0 + cv + j] : 0.f;
        csg[j] = (ln && geglu) ? p.ln_colsum[n0 + cv + 32 + j] : 0.f;
    }
    float gs[8], gq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { gs[j] = 0.f; gq[j] = 0.f; }
    const int lk = epi8_load_kind(p);               // (launch-uniform; gemm_common.h epi8_apply)
    if (active && lk == 1 && !geglu) {
        // residual launches (to_out, the two-source block end): the thread's residual chunks are all requested before the first is used - each
        // wait then leaves the younger requests and every store in flight; a workgroup of these launches has its CU to itself at M <= 1024
        constexpr int RGn = 256 / (BN / 8), NRW = (GW_BM + RGn - 1) / RGn;
        uint4 rr[NRW];
#pragma unroll
        for (int k = 0; k < NRW; ++k)
            rr[k] = bc_ld16(reinterpret_cast<const h16*>(p.R) + (size_t)(m0 + min(rg + k * RGn, GW_BM - 1)) * p.ldr + n_first);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < NRW; ++k) {
            const int row = rg + k * RGn;
            if (NRW * RGn != GW_BM && row >= GW_BM) break;
            const float4 lo = *reinterpret_cast<const float4*>(tile + tile_off<BN>(row, cv));
            const float4 hi = *reinterpret_cast<const float4*>(tile + tile_off<BN>(row, cv + 4));
            float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            if (ln) {
                const float mean = stat[row * 2], rstd = stat[row * 2 + 1];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = rstd * (v[j] - mean * csv[j]);
            }
            bc_st16(reinterpret_cast<h16*>(p.C) + (size_t)(m0 + row) * p.ldc + n_first, epi8_apply<1>(g, cols, v, v, gs, gq, rr[k]));
        }
    } else if (active) {
        // (two instantiations of the loop: launches without a load in this pass take the load-free one - with epi8_store's loads anywhere in the
        //  loop body, taken or not, its joins make every chunk wait for the previous chunk's store)
        auto rows = [&](auto NOLOAD) {
            for (int row = rg; row < GW_BM; row += RG) {
                const float4 lo = *reinterpret_cast<const float4*>(tile + tile_off<BN>(row, cv));
                const float4 hi = *reinterpret_cast<const float4*>(tile + tile_off<BN>(row, cv + 4));
                float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                float gt[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (geglu) {
                    const float4 glo = *reinterpret_cast<const float4*>(tile + tile_off<BN>(row, cv + 32));
                    const float4 ghi = *reinterpret_cast<const float4*>(tile + tile_off<BN>(row, cv + 36));
                    gt[0] = glo.x; gt[1] = glo.y; gt[2] = glo.z; gt[3] = glo.w;
                    gt[4] = ghi.x; gt[5] = ghi.y; gt[6] = ghi.z; gt[7] = ghi.w;
                }
                if (ln) {
                    const float mean = stat[row * 2], rstd = stat[row * 2 + 1];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        v[j] = rstd * (v[j] - mean * csv[j]);
                        gt[j] = rstd * (gt[j] - mean * csg[j]);
                    }
                }
                if (decltype(NOLOAD)::value)
                    bc_st16(reinterpret_cast<h16*>(p.C) + (size_t)(m0 + row) * p.ldc + n_first, epi8_apply<0>(g, cols, v, gt, gs, gq, make_uint4(0u, 0u, 0u, 0u)));
                else
                    epi8_store(g, cols, v, gt, m0 + row, gs, gq);
            }
        };
        if (lk == 0) rows(std::true_type{});
        else rows(std::false_type{});
    }
    if (p.gn_tot) {
        // per-channel (sum, sum of squares) of the fp16 output over the block's 64 rows: threads of equal col8 combine through LDS
        __syncthreads();                            // tile fully consumed; reuse its head as scratch [RG][TSO][2]
        float* scr = reinterpret_cast<float*>(smem);
        if (active) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                scr[((rg * TSO) + c_out + j) * 2] = gs[j];
                scr[((rg * TSO) + c_out + j) * 2 + 1] = gq[j];
            }
        }
        __syncthreads();
        for (int c = tid; c < TSO; c += 256) {
            const int b = (int)fdiv((unsigned)m0, g.div_rpb), nb = geglu ? n0 / 2 : n0;
            bc_gn_tot_add_slot(p.gn_tot + (size_t)b * g.n_out * BC_GN_TOT_WORDS, nb + c, nb, nb + TSO, bc_gn_cg(g.n_out), m0 / GW_BM, [&](int k) {
                float s = 0.f, qq = 0.f;
                for (int r = 0; r < RG; ++r) {
                    s += scr[(r * TSO + k - nb) * 2];
                    qq += scr[(r * TSO + k - nb) * 2 + 1];
                }
                return make_float2(s, qq);
            });
        }
    }
}

template <int NT, int R, bool SM = false>
int launch_gw(const GemmArgs& g, hipStream_t stream) {
    constexpr int BN = 64 * NT;
    // two operand buffers | the fp32 epilogue tile + the LayerNorm statistics behind it (the tile of NT <= 4 leaves room inside the 80 KiB)
    constexpr int LDS = 2 * GW_XBUF > GW_BM * BN * 4 + GW_BM * 8 ? 2 * GW_XBUF : GW_BM * BN * 4 + GW_BM * 8;
    static_assert(NT == 5 || 2 * LDS <= 160 * 1024, "two workgroups per CU");
    constexpr int LDS_MAX = LDS + 2560 * 8;                  // + the GroupNorm (a, b) table of up to 2560 channels (one workgroup per CU then)
    static std::atomic<unsigned long long> lds_set{0};
    BC_CHECK_HIP(bc_set_max_lds(lds_set, reinterpret_cast<const void*>(&gemm_wreg_kernel<NT, R, SM>), LDS_MAX));
    const int grid = (g.p.M / GW_BM) * (g.p.N / BN);
    hipLaunchKernelGGL((gemm_wreg_kernel<NT, R, SM>), dim3(grid), dim3(256), LDS + (g.p.a_tot1 ? g.p.K * 8 : 0), stream, g);
    BC_CHECK_LAUNCH();
    return 0;
}

__global__ void gw_pack_kernel(const h16* __restrict__ w, int ldw, int N, int K, int NT, uint4* __restrict__ out) {
    // one thread per 16-byte piece of the stream [column tile][wave][k-step][tile][lane = 16 q + r][8]
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)N * K / 8;
    if (idx >= total) return;
    const int KS = K / 32;
    const int lane = (int)(idx & 63);
    long long f = idx >> 6;                          // fragment index
    const int t = (int)(f % NT); f /= NT;
    const int s = (int)(f % KS); f /= KS;
    const int wave = (int)(f & 3);
    const int jt = (int)(f >> 2);
    const int n = jt * 64 * NT + wave * 16 * NT + 16 * t + (lane & 15);
    const int k = 32 * s + 8 * (lane >> 4);
    out[idx] = *reinterpret_cast<const uint4*>(w + (size_t)n * ldw + k);
}

// ---------------------------------------------------------------------------------------------- prompt folded into the weights (bc_ctx_fold)
constexpr int CF_KEYS = 80;             // keys per head the fold computes (BcGemm.sm_keep of the consumer: columns of the probabilities)
constexpr int CF_GROUP = 128;           // rows per head in the QK stream (BcGemm.sm_group: one 64 x 128 workgroup per head; rows >= 80 stay zero)
constexpr int CF_DMAX = 160;            // head width the LDS staging holds

// 16-byte piece of W[n][8 k8 .. + 8] inside a BC_TILE_GW* stream of NT tiles per wave and KS k-steps (gw_pack_kernel's order)
__device__ __forceinline__ size_t gw_piece(int n, int k8, int NT, int KS) {
    const int bn = 64 * NT;
    const int jt = n / bn, r = n - jt * bn;
    const int wave = r / (16 * NT), r2 = r - wave * 16 * NT;
    return ((((size_t)(jt * 4 + wave) * KS + (k8 >> 2)) * NT + (r2 >> 4)) * 64) + (k8 & 3) * 16 + (r2 & 15);
}

// QK[b][(h, j)][c] = scale sum_d k[b][j][h D + d] wq[h D + d][c]: workgroup = (20 keys, head, image), thread = (10 keys, 8 columns c);
// the column sums of the ROUNDED rows and the bias row scale k . bq come out of the same workgroup (fixed summation order).
__global__ __launch_bounds__(320) void ctx_fold_qk_kernel(const h16* __restrict__ k, int ldk, int T, int C, int heads, float scale,
                                                          const h16* __restrict__ wq, const float* __restrict__ bq, uint4* __restrict__ wqk,
                                                          long long wqk_stride16, float* __restrict__ colsum, float* __restrict__ qbias) {
    __shared__ float ks[20][CF_DMAX];
    __shared__ float psum[20][160];
    const int tid = threadIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int D = C / heads, j0 = blockIdx.x * 20;
    for (int i = tid; i < 20 * D; i += 320) {
        const int jl = i / D, d = i - jl * D;
        ks[jl][d] = (j0 + jl) < T ? (float)k[((size_t)b * T + j0 + jl) * ldk + h * D + d] : 0.f;
    }
    __syncthreads();
    const int half = tid / 160, kl = tid - half * 160;
    const int KS = C / 32, N = heads * CF_GROUP;
    uint4* out = wqk + (size_t)b * wqk_stride16;
    float cs[10];
#pragma unroll
    for (int jj = 0; jj < 10; ++jj) cs[jj] = 0.f;
    for (int k8 = kl; k8 < C / 8; k8 += 160) {
        float acc[10][8];
#pragma unroll
        for (int jj = 0; jj < 10; ++jj)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[jj][e] = 0.f;
        const h16* wp = wq + (size_t)h * D * C + k8 * 8;
        for (int d = 0; d < D; ++d) {
            const h16x8 w8 = *reinterpret_cast<const h16x8*>(wp + (size_t)d * C);
            float wf[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) wf[e] = (float)w8[e];
#pragma unroll
            for (int jj = 0; jj < 10; ++jj) {
                const float kv = ks[half * 10 + jj][d];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[jj][e] = fmaf(kv, wf[e], acc[jj][e]);
            }
        }
#pragma unroll
        for (int jj = 0; jj < 10; ++jj) {
            uint4 raw;
            h16* o = reinterpret_cast<h16*>(&raw);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[e] = (h16)(acc[jj][e] * scale);
                cs[jj] += (float)o[e];
            }
            out[gw_piece(h * CF_GROUP + j0 + half * 10 + jj, k8, 2, KS)] = raw;
        }
    }
#pragma unroll
    for (int jj = 0; jj < 10; ++jj) psum[half * 10 + jj][kl] = cs[jj];
    __syncthreads();
    if (tid < 20) {
        float s_ = 0.f;
        for (int i = 0; i < 160; ++i) s_ += psum[tid][i];
        float bv = 0.f;
        for (int d = 0; d < D; ++d) bv = fmaf(ks[tid][d], bq[h * D + d], bv);
        colsum[(size_t)b * N + h * CF_GROUP + j0 + tid] = s_;
        qbias[(size_t)b * N + h * CF_GROUP + j0 + tid] = bv * scale;
    }
}

// VO[b][n][(h, j)] = sum_d wo[n][h D + d] vt[b][h D + d][j]: workgroup = (256 output channels n, head, image), thread = one n, all 80 keys
__global__ __launch_bounds__(256) void ctx_fold_vo_kernel(const h16* __restrict__ vt, int ldvt, int T, int C, int heads, const h16* __restrict__ wo,
                                                          uint4* __restrict__ vwo, long long vwo_stride16) {
    __shared__ __attribute__((aligned(16))) float vs[CF_DMAX][CF_KEYS];
    const int tid = threadIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int D = C / heads;
    for (int i = tid; i < D * CF_KEYS; i += 256) {
        const int d = i / CF_KEYS, j = i - d * CF_KEYS;
        vs[d][j] = j < T ? (float)vt[((size_t)b * C + h * D + d) * ldvt + j] : 0.f;
    }
    __syncthreads();
    const int n = blockIdx.x * 256 + tid;
    if (n >= C) return;
    float acc[CF_KEYS];
#pragma unroll
    for (int j = 0; j < CF_KEYS; ++j) acc[j] = 0.f;
    const h16* wp = wo + (size_t)n * C + h * D;
    for (int d8 = 0; d8 < D; d8 += 8) {
        const h16x8 w8 = *reinterpret_cast<const h16x8*>(wp + d8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float wv = (float)w8[e];
            const float4* vr = reinterpret_cast<const float4*>(vs[d8 + e]);
#pragma unroll
            for (int j4 = 0; j4 < CF_KEYS / 4; ++j4) {
                const float4 v4 = vr[j4];
                acc[4 * j4] = fmaf(wv, v4.x, acc[4 * j4]);
                acc[4 * j4 + 1] = fmaf(wv, v4.y, acc[4 * j4 + 1]);
                acc[4 * j4 + 2] = fmaf(wv, v4.z, acc[4 * j4 + 2]);
                acc[4 * j4 + 3] = fmaf(wv, v4.w, acc[4 * j4 + 3]);
            }
        }
    }
    uint4* out = vwo + (size_t)b * vwo_stride16;
    const int KS = heads * CF_KEYS / 32;
#pragma unroll
    for (int j8 = 0; j8 < CF_KEYS / 8; ++j8) {
        uint4 raw;
        h16* o = reinterpret_cast<h16*>(&raw);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (h16)acc[j8 * 8 + e];
        out[gw_piece(n, h * (CF_KEYS / 8) + j8, 2, KS)] = raw;
    }
}

}  // namespace

int bc_gemm_wreg_nt(int tile_cfg) {
    return tile_cfg == BC_TILE_GW64x128 ? 2 : tile_cfg == BC_TILE_GW64x256 ? 4 : tile_cfg == BC_TILE_GW64x320 ? 5 : 0;
}

// 1 when bc_gemm can run this problem on the given BC_TILE_GW* configuration
int bc_gemm_wreg_ok(const BcGemm& p, int tile_cfg) {
    const int nt = bc_gemm_wreg_nt(tile_cfg);
    if (!nt || p.a_mode != BC_A_DENSE) return 0;
    if (p.M <= 0 || p.M % GW_BM || p.N % (64 * nt) || p.K % (32 * GW_KC)) return 0;
    if (p.A2 && (p.C1 % (32 * GW_KC) || p.C1 <= 0 || p.C1 >= p.K)) return 0;
    if (p.out_mode != BC_OUT_F16 || p.splitk > 1 || p.rowvec || p.a_affine) return 0;
    if (p.a_tot1) {                                   // GroupNorm finalized in the prologue and applied while the rows are staged
        const int rpb = p.rows_per_batch > 0 ? p.rows_per_batch : p.M;
        if (!p.a_gamma || !p.a_beta || p.a_groups <= 0 || p.K % p.a_groups || p.a_groups > 256 || p.K > 2560 || p.A2 || p.ln_colsum || rpb % GW_BM ||
            p.M % rpb || p.a_act != BC_ACT_NONE || p.K * 16 + p.a_groups * 8 > 2 * GW_XBUF)
            return 0;
    }
    if (p.act != BC_ACT_NONE && p.act != BC_ACT_GEGLU && p.act != BC_ACT_GELU && p.act != BC_ACT_SILU && p.act != BC_ACT_QUICK_GELU) return 0;
    if (p.ln_colsum && p.A2) return 0;               // (the statistics cover one source)
    if (p.w_bstride || p.vec_bstride) {               // per-image weights: a row block lies inside one image; streams 16-byte aligned
        const int rpb = p.rows_per_batch > 0 ? p.rows_per_batch : p.M;
        if (p.w_bstride < 0 || p.w_bstride % 8 || p.vec_bstride < 0 || rpb % GW_BM || p.M % rpb) return 0;
        if (p.vec_bstride && (!p.sm_group || p.vec_bstride < p.N)) return 0;       // (per-image bias / colsum: the softmax epilogue only)
    }
    if (p.sm_group) {                                 // softmax epilogue: the 64 x 128 workgroup = one group
        if (nt != 2 || p.sm_group != 64 * nt || p.sm_valid <= 0 || p.sm_valid > p.sm_keep || p.sm_keep > p.sm_group || p.sm_keep % 8 ||
            p.ldc < p.N / p.sm_group * p.sm_keep)
            return 0;
        if (p.act != BC_ACT_NONE || p.R || p.R2 || p.colscale || p.gn_tot || p.alpha_dev || p.alpha != 1.0f || p.C_t || p.a_tot1 || p.A2) return 0;
    }
    if (p.C_t) {
        const int rpb = p.rows_per_batch > 0 ? p.rows_per_batch : p.M;
        if (p.n_t0 <= 0 || p.n_t0 % (64 * nt) || p.n_t0 >= p.N || rpb % GW_BM || p.M % rpb || p.ldc_t < rpb || p.ldc_t % 8 ||
            p.act != BC_ACT_NONE || p.R || p.R2 || p.colscale || p.gn_tot || p.alpha_dev || p.alpha != 1.0f)
            return 0;
    }
    if (p.gn_tot) {
        const int rpb = p.rows_per_batch > 0 ? p.rows_per_batch : p.M;
        if (rpb % GW_BM || p.M % rpb) return 0;
    }
    return 1;
}

int bc_gemm_wreg_launch(const GemmArgs& g, hipStream_t stream) {
    switch (g.cfg) {
        case BC_TILE_GW64x128: return g.p.sm_group ? launch_gw<2, 20, true>(g, stream) : launch_gw<2, 20>(g, stream);
        case BC_TILE_GW64x256: return launch_gw<4, 20>(g, stream);
        case BC_TILE_GW64x320: return launch_gw<5, 10>(g, stream);
        default: bc_set_error("bc_gemm: not a BC_TILE_GW* configuration (%d)", g.cfg); return 1;
    }
}

extern "C" int bc_gemm_wreg_eligible(int M, int N, int K, int C1, int tile_cfg) {
    BcGemm p = {};
    p.a_mode = BC_A_DENSE; p.M = M; p.N = N; p.K = K; p.C1 = C1; p.out_mode = BC_OUT_F16; p.splitk = 1; p.alpha = 1.0f;
    p.A2 = C1 > 0 ? reinterpret_cast<const bc_half*>(&p) : nullptr;       // (only tested for non-null)
    return bc_gemm_wreg_ok(p, tile_cfg);
}

extern "C" long long bc_gemm_wreg_stream_elems(int N, int K) {
    // elements (bc_half) of a packed stream incl. the tail the register ring reads past the last fragment (never used)
    return (long long)N * K + 32 * 512;
}

extern "C" int bc_gemm_wreg_pack(const bc_half* w, int ldw, int N, int K, int tile_cfg, bc_half* out, bc_stream stream) {
    const int nt = bc_gemm_wreg_nt(tile_cfg);
    BC_CHECK_ARG(w && out && nt && N > 0 && N % (64 * nt) == 0 && K > 0 && K % 32 == 0 && ldw >= K && ldw % 8 == 0,
                 "bc_gemm_wreg_pack: needs a BC_TILE_GW* configuration, N %% %d == 0 and K %% 32 == 0 (N=%d K=%d)", 64 * nt, N, K);
    BC_CHECK_ARG(w != out, "bc_gemm_wreg_pack: out of place only");
    const long long total = (long long)N * K / 8;
    hipLaunchKernelGGL(gw_pack_kernel, dim3(bc_ceil_div(total, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const h16*>(w), ldw, N, K, nt, reinterpret_cast<uint4*>(out));
    BC_CHECK_LAUNCH();
    return 0;
}

extern "C" int bc_ctx_fold(const bc_half* k, int ldk, const bc_half* vt, int ldvt, int B, int T, int channels, int heads, float scale,
                           const bc_half* wq, const float* bq, const bc_half* wo, bc_half* wqk, float* qk_colsum, float* qk_bias, bc_half* vwo,
                           bc_stream stream) {
    BC_CHECK_ARG(k && vt && wq && bq && wo && wqk && qk_colsum && qk_bias && vwo, "bc_ctx_fold: null argument");
    BC_CHECK_ARG(B > 0 && heads > 0 && T > 0 && T <= CF_KEYS && channels > 0 && channels % (8 * heads) == 0 && channels / heads <= CF_DMAX &&
                 channels % 320 == 0 && (heads * CF_KEYS) % 320 == 0 && ldk >= channels && ldvt >= T,
                 "bc_ctx_fold: needs T <= 80, channels %% 320 == 0, head width %% 8 == 0 and <= 160, 80 heads %% 320 == 0 (B=%d T=%d C=%d heads=%d)",
                 B, T, channels, heads);
    const long long s_qk = bc_gemm_wreg_stream_elems(heads * CF_GROUP, channels), s_vo = bc_gemm_wreg_stream_elems(channels, heads * CF_KEYS);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(ctx_fold_qk_kernel, dim3(CF_KEYS / 20, heads, B), dim3(320), 0, st, reinterpret_cast<const h16*>(k), ldk, T, channels, heads,
                       scale, reinterpret_cast<const h16*>(wq), bq, reinterpret_cast<uint4*>(wqk), s_qk / 8, qk_colsum, qk_bias);
    BC_CHECK_LAUNCH();
    hipLaunchKernelGGL(ctx_fold_vo_kernel, dim3(bc_ceil_div(channels, 256), heads, B), dim3(256), 0, st, reinterpret_cast<const h16*>(vt), ldvt, T,
                       channels, heads, reinterpret_cast<const h16*>(wo), reinterpret_cast<uint4*>(vwo), s_vo / 8);
    BC_CHECK_LAUNCH();
    return 0;
}
