// Shared host/device pieces of the implicit-GEMM kernels (gemm.hip = generic kernel + dispatch, gemm_fast.hip = LDS-DMA kernel).
#pragma once
#include "bc_common.h"

namespace bcg {


constexpr int BK = 64;   // K-tile (elements); 128-byte LDS rows

struct FastDiv {          // exact floor(n / d) for 0 <= n < 2^31
    unsigned mul, shift, d;
};

static inline FastDiv make_fastdiv(unsigned d) {
    FastDiv f;
    if (d == 0) d = 1;
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    unsigned long long m = ((1ull << (31 + l)) + d - 1) / d;
    f.mul = (unsigned)m;
    f.shift = 31 + l;
    f.d = d;
    return f;
}

__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& f) {
    return (unsigned)(((unsigned long long)n * f.mul) >> f.shift);
}

struct GemmArgs {
    BcGemm p;
    FastDiv div_rpb;     // rows_per_batch
    FastDiv div_outw;    // out_w
    FastDiv div_wout;    // conv Wout
    int nk;              // number of K tiles
    int kt_per_split;
    int n_out;           // output columns (N, or N/2 for GEGLU)
    int fast_k;          // conv: Cin % BK == 0 ; dense: (C1 % BK == 0 or no A2)
    int cfg;             // resolved BC_TILE_* configuration
    int bm, bn;          // its tile shape
    int vec_epilogue;    // 1: LDS-staged 16-byte epilogue is legal (fp16 row-major output, widths % 8 == 0)
    int vec_transposed;  // 1: BC_OUT_F16_T with rows_per_batch % 8 == 0 and ldc % 8 == 0: 16-byte stores along the token axis
    int halo_tx, halo_tpi;   // halo conv: pixel tiles per image row / per image
    int halo_nch, halo_cps;  // halo conv: 64-channel chunks in total / per split
    int halo_dbg;            // halo conv: BC_HALO_DBG ablation bits (diagnostics)
    int nband;               // fast GEMM: 1 = an XCD owns a band of COLUMN tiles (all row tiles): it fetches 1/8 of the weights and the
                             // whole activation - chosen when the weight matrix is the larger operand (low-resolution levels)
    unsigned long long* halo_stamps;   // halo conv: BC_HALO_STAMPS=1 -> [workgroup][8] s_memtime stamps (diagnostics), else null
    FastDiv wr_plane, wr_gx, wr_gy, wr_tpi, wr_tx, wr_cpg;   // conv_wreg: grid plane / grid.x / grid.y / tiles per image / tiles per image row / channels per
                                                             // GroupNorm group (the kernel's prologue did seven integer divisions in SALU code)
};

__device__ __forceinline__ int lds_off(int row, int chunk) {   // byte offset inside a [rows][64] fp16 tile
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// Launch-wide scalar multiplier; with alpha_bstride > 0 the device table holds one scalar per image and is applied per row.
__device__ __forceinline__ float scalar_alpha(const BcGemm& p) {
    float alpha = p.alpha;
    if (p.alpha_dev && p.alpha_bstride == 0) alpha *= p.alpha_dev[p.alpha_idx ? *p.alpha_idx : 0];
    return alpha;
}
// Row vector of this launch: a fixed pointer, or the block of the current denoise step inside a per-edit table.
__device__ __forceinline__ const h16* rowvec_base(const BcGemm& p) {
    return reinterpret_cast<const h16*>(p.rowvec) + (p.rowvec_idx ? (size_t)(*p.rowvec_idx) * p.rowvec_step : 0);
}
__device__ __forceinline__ float batch_alpha(const GemmArgs& g, int b) {
    const BcGemm& p = g.p;
    return p.alpha_dev[(p.alpha_idx ? *p.alpha_idx : 0) * p.alpha_bstride + b];
}

// One output element: everything after the accumulator (+bias +rowvec, activation) has been applied by the caller.
__device__ __forceinline__ void epilogue_store(const GemmArgs& g, float v, int m, int n, float alpha) {
    const BcGemm& p = g.p;
    if (p.colscale) v *= p.colscale[n];
    v *= alpha;
    if (p.alpha_bstride > 0) v *= batch_alpha(g, (int)fdiv((unsigned)m, g.div_rpb));
    if (p.R) v += (float)reinterpret_cast<const h16*>(p.R)[(size_t)m * p.ldr + n];
    int b = 0, pix = m;
    if (p.R2 || p.out_mode == BC_OUT_F16_T) {
        b = (int)fdiv((unsigned)m, g.div_rpb);
        pix = m - b * (int)g.div_rpb.d;
    }
    if (p.R2) {
        int y = (int)fdiv((unsigned)pix, g.div_outw);
        int x = pix - y * (int)g.div_outw.d;
        if (x >= p.r2_xmin) {
            int bb = b % p.r2_bmod;
            v += (float)reinterpret_cast<const h16*>(p.R2)[((size_t)bb * g.div_rpb.d + pix) * p.ldr2 + n];
        }
    }
    if (p.out_mode == BC_OUT_F16) {
        reinterpret_cast<h16*>(p.C)[(size_t)m * p.ldc + n] = (h16)v;
    } else if (p.out_mode == BC_OUT_F32) {
        reinterpret_cast<float*>(p.C)[(size_t)m * p.ldc + n] = v;
    } else {
        reinterpret_cast<h16*>(p.C)[((size_t)b * g.n_out + n) * p.ldc + pix] = (h16)v;
    }
}

__device__ __forceinline__ float pre_act(const GemmArgs& g, float acc, int m, int ncol) {
    const BcGemm& p = g.p;
    float v = acc;
    if (p.bias) v += p.bias[ncol];
    if (p.rowvec) {
        int b = (int)fdiv((unsigned)m, g.div_rpb);
        v += (float)rowvec_base(p)[(size_t)b * p.ld_rowvec + ncol];
    }
    return v;
}


// ---- epilogue helpers shared by both kernels -----------------------------------------------------------------
// Accumulators are first parked RAW in LDS with compile-time register indices only (anything fancier inside the
// unrolled register loops makes hipcc give up unrolling and demote the accumulator array to scratch memory), then a
// plain per-element / per-chunk loop applies the epilogue with runtime indices.
template <int TM, int TN>
__device__ __forceinline__ void acc_to_tile(float* tile, int ts, const f32x16 (&acc)[TM][TN], int wave_m0, int wave_n0,
                                            int frow, int fhalf) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                tile[(wave_m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf) * ts + wave_n0 + j * 32 + frow] = acc[i][j][r];
}

template <int BM, int BN, int NT>
__device__ __forceinline__ void tile_epilogue_scalar(const GemmArgs& g, const float* tile, int m0, int n0, int split,
                                                     int tid) {
    const BcGemm& p = g.p;
    if (p.splitk > 1) {
        float* slab = p.slab + (size_t)split * p.M * p.N;
        for (int idx = tid; idx < BM * BN; idx += NT) {
            const int row = idx / BN, col = idx - row * BN;
            const int m = m0 + row, n = n0 + col;
            if (m < p.M && n < p.N) slab[(size_t)m * p.N + n] = tile[idx];
        }
        return;
    }
    const float alpha = scalar_alpha(p);
    if (p.act == BC_ACT_GEGLU) {
        constexpr int BNO = BN / 2;
        for (int idx = tid; idx < BM * BNO; idx += NT) {
            const int row = idx / BNO, c = idx - row * BNO;
            const int m = m0 + row;
            const int cv = (c >> 5) * 64 + (c & 31);           // tile column of the value; gate is 32 further
            const int nv = n0 + cv, ng = nv + 32;
            if (m < p.M && ng < p.N) {
                const float v = pre_act(g, tile[row * BN + cv], m, nv);
                const float gt = pre_act(g, tile[row * BN + cv + 32], m, ng);
                epilogue_store(g, v * bc_gelu_f(gt), m, n0 / 2 + c, alpha);
            }
        }
        return;
    }
    for (int idx = tid; idx < BM * BN; idx += NT) {
        const int row = idx / BN, col = idx - row * BN;
        const int m = m0 + row, n = n0 + col;
        if (m < p.M && n < p.N) {
            float v = pre_act(g, tile[idx], m, n);
            if (p.act == BC_ACT_GELU) v = bc_gelu_f(v);
            else if (p.act == BC_ACT_SILU) v = bc_silu_f(v);
            else if (p.act == BC_ACT_QUICK_GELU) v = bc_quick_gelu_f(v);
            epilogue_store(g, v, m, n, alpha);
        }
    }
}

// ---- vectorised (8 output columns) epilogue shared by the fast kernel's row-major pass and the split-K reducer ----
struct Cols8 {
    float bias_v[8], bias_g[8], cs[8];
    int n_first;        // first OUTPUT column
    bool nok;
};

__device__ __forceinline__ void cols8_init(const GemmArgs& g, Cols8& c, int n_first_out, int ncol_value, bool geglu, float alpha) {
    const BcGemm& p = g.p;
    c.n_first = n_first_out;
    c.nok = n_first_out < g.n_out;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        c.bias_v[j] = (p.bias && c.nok) ? p.bias[ncol_value + j] : 0.f;
        c.bias_g[j] = (geglu && p.bias && c.nok) ? p.bias[ncol_value + 32 + j] : 0.f;
        c.cs[j] = ((p.colscale && c.nok) ? p.colscale[n_first_out + j] : 1.f) * alpha;
    }
}

// v[8] = raw accumulators of the (value) columns, gt = raw gate accumulators (GEGLU only).  Applies bias, row vector,
// activation, LayerScale*alpha, residual, BlobNet right-half residual; stores 8 fp16; accumulates GroupNorm partials.
__device__ __forceinline__ void epi8_store(const GemmArgs& g, const Cols8& c, float (&v)[8], const float (&gt)[8], int m,
                                           float (&gs)[8], float (&gq)[8]) {
    const BcGemm& p = g.p;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += c.bias_v[j];
    int b = 0, pix = m;
    if (p.rowvec || p.R2 || p.alpha_bstride > 0) {
        b = (int)fdiv((unsigned)m, g.div_rpb);
        pix = m - b * (int)g.div_rpb.d;
    }
    if (p.rowvec) {
        const uint4 raw = bc_ld16(rowvec_base(p) + (size_t)b * p.ld_rowvec + c.n_first);
        const h16* rh = reinterpret_cast<const h16*>(&raw);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += (float)rh[j];
    }
    if (p.act == BC_ACT_GEGLU) {
#pragma unroll
        for (int j = 0; j < 8; j += 2) {             // (bc_gelu_f2 = bc_gelu_f bit for bit, two elements per VALU issue)
            const f32x2 gl = bc_gelu_f2((f32x2){gt[j] + c.bias_g[j], gt[j + 1] + c.bias_g[j + 1]});
            v[j] *= gl.x;
            v[j + 1] *= gl.y;
        }
    } else if (p.act == BC_ACT_GELU) {
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const f32x2 gl = bc_gelu_f2((f32x2){v[j], v[j + 1]});
            v[j] = gl.x;
            v[j + 1] = gl.y;
        }
    } else if (p.act == BC_ACT_SILU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = bc_silu_f(v[j]);
    } else if (p.act == BC_ACT_QUICK_GELU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = bc_quick_gelu_f(v[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= c.cs[j];
    if (p.alpha_bstride > 0) {
        const float ab = batch_alpha(g, b);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= ab;
    }
    if (p.R) {
        const uint4 raw = bc_ld16(reinterpret_cast<const h16*>(p.R) + (size_t)m * p.ldr + c.n_first);
        const h16* rh = reinterpret_cast<const h16*>(&raw);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += (float)rh[j];
    }
    if (p.R2) {
        const int y = (int)fdiv((unsigned)pix, g.div_outw);
        const int x = pix - y * (int)g.div_outw.d;
        if (x >= p.r2_xmin) {
            const int bb = b % p.r2_bmod;
            const uint4 raw = bc_ld16(reinterpret_cast<const h16*>(p.R2) + ((size_t)bb * g.div_rpb.d + pix) * p.ldr2 + c.n_first);
            const h16* rh = reinterpret_cast<const h16*>(&raw);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += (float)rh[j];
        }
    }
    uint4 outraw;
    h16* o = reinterpret_cast<h16*>(&outraw);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        o[j] = (h16)v[j];
        const float f = (float)o[j];
        gs[j] += f;
        gq[j] += f * f;
    }
    bc_st16(reinterpret_cast<h16*>(p.C) + (size_t)m * p.ldc + c.n_first, outraw);
}

// epi8_store without its global LOADS, for the launches that need none (LD 0) or only the residual R (LD 1: `rraw`, requested by the caller
// ahead of the stores); the arithmetic and its order are epi8_store's, bit for bit.  Why it exists (round 6): the compiler's s_waitcnt for a
// load that sits under a branch - epi8_store's row vector / residual / right-half residual, taken or not - is vmcnt(0) at the join, and vmcnt
// is in-order: every 8-column chunk waits for the previous chunk's STORE to be acknowledged, and a residual is requested and awaited chunk by
// chunk.  Where another workgroup of the CU covers that (gemm_fast: measured, no gain) epi8_store stays; gemm_wreg.hip's launches at M <= 1024
// have a CU to themselves.  The caller stores the returned 16 bytes.
template <int LD>
__device__ __forceinline__ uint4 epi8_apply(const GemmArgs& g, const Cols8& c, float (&v)[8], const float (&gt)[8], float (&gs)[8], float (&gq)[8],
                                            const uint4 rraw) {
    const BcGemm& p = g.p;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += c.bias_v[j];
    if (p.act == BC_ACT_GEGLU) {
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const f32x2 gl = bc_gelu_f2((f32x2){gt[j] + c.bias_g[j], gt[j + 1] + c.bias_g[j + 1]});
            v[j] *= gl.x;
            v[j + 1] *= gl.y;
        }
    } else if (p.act == BC_ACT_GELU) {
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const f32x2 gl = bc_gelu_f2((f32x2){v[j], v[j + 1]});
            v[j] = gl.x;
            v[j + 1] = gl.y;
        }
    } else if (p.act == BC_ACT_SILU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = bc_silu_f(v[j]);
    } else if (p.act == BC_ACT_QUICK_GELU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = bc_quick_gelu_f(v[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= c.cs[j];
    if (LD == 1) {
        const h16* rh = reinterpret_cast<const h16*>(&rraw);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += (float)rh[j];
    }
    uint4 outraw;
    h16* o = reinterpret_cast<h16*>(&outraw);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        o[j] = (h16)v[j];
        const float f = (float)o[j];
        gs[j] += f;
        gq[j] += f * f;
    }
    return outraw;
}
// which form a launch's row-major pass can take (launch-uniform): 2 = epi8_store
__device__ __forceinline__ int epi8_load_kind(const BcGemm& p) { return (p.rowvec || p.R2 || p.alpha_bstride > 0) ? 2 : p.R ? 1 : 0; }

// Transposed parking of the accumulators: tileT[n][m] with row stride ts (= BM + 4: the 32 lanes of a store hit 32 different
// rows, the +4 floats keep them on different banks).  Lane holds column n = lane&31 and 4 consecutive rows per register group.
template <int TM, int TN>
__device__ __forceinline__ void acc_to_tile_t(float* tile, int ts, const f32x16 (&acc)[TM][TN], int wave_m0, int wave_n0,
                                              int frow, int fhalf) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float4 v = make_float4(acc[i][j][4 * g4], acc[i][j][4 * g4 + 1], acc[i][j][4 * g4 + 2], acc[i][j][4 * g4 + 3]);
                *reinterpret_cast<float4*>(tile + (wave_n0 + j * 32 + frow) * ts + wave_m0 + i * 32 + 8 * g4 + 4 * fhalf) = v;
            }
}

// BC_OUT_F16_T epilogue (V^T for attention): thread = (output column n, 8 consecutive tokens) -> one 16-byte store into
// C[(b*N + n)*ldc + pix].  Supports bias and alpha (what the V projections use); anything else takes the scalar path.
template <int BM, int BN, int NT>
__device__ __forceinline__ void tile_epilogue_transposed(const GemmArgs& g, const float* tileT, int ts, int m0, int n0, int tid) {
    const BcGemm& p = g.p;
    const float alpha = scalar_alpha(p);
    constexpr int MCH = BM / 8;                       // 8-token chunks per column
    for (int idx = tid; idx < BN * MCH; idx += NT) {
        const int col = idx / MCH, mc = idx - col * MCH;
        const int n = n0 + col, m = m0 + mc * 8;
        if (n >= p.N || m >= p.M) continue;
        const float4 lo = *reinterpret_cast<const float4*>(tileT + col * ts + mc * 8);
        const float4 hi = *reinterpret_cast<const float4*>(tileT + col * ts + mc * 8 + 4);
        const float bias = p.bias ? p.bias[n] : 0.f;
        const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        const int b = (int)fdiv((unsigned)m, g.div_rpb);
        const int pix = m - b * (int)g.div_rpb.d;
        uint4 outraw;
        h16* o = reinterpret_cast<h16*>(&outraw);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (h16)((v[j] + bias) * alpha);
        h16* dst = reinterpret_cast<h16*>(p.C) + ((size_t)b * g.n_out + n) * p.ldc + pix;
        if (m + 8 <= p.M) {
            bc_st16(dst, outraw);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)               // (constant indices: a runtime-indexed o[] would live in scratch)
                if (j < p.M - m) dst[j] = o[j];
        }
    }
}

}  // namespace bcg

// gemm_fast.hip: returns 0 when it launched the GEMM, -1 when the problem is outside its fast path, >0 on error.
int bc_gemm_fast_try(const bcg::GemmArgs& g, hipStream_t stream);
// conv_halo.hip: LDS-resident input-halo 3x3 convolution with the fused GroupNorm prologue (BC_TILE_HALO).
int bc_conv_halo_ok(const BcGemm& p);
int bc_conv_halo_max_chunks_impl();
void bc_gemm_set_probe(hipEvent_t e);      // gemm.hip: event recorded between the main kernel and the split-K reducer (nullptr: off)
bool bc_gemm_probe_hit();
int bc_conv_halo_launch(bcg::GemmArgs& g, hipStream_t stream);
// conv_wreg.hip: the same convolution with the weights streamed into VGPRs from a packed fragment stream (BC_TILE_WREG).
int bc_conv_wreg_launch(bcg::GemmArgs& g, hipStream_t stream);
// gemm_wreg.hip: small-M projections with the weights streamed into VGPRs (BC_TILE_GW*).
int bc_gemm_wreg_nt(int tile_cfg);
int bc_gemm_wreg_ok(const BcGemm& p, int tile_cfg);
int bc_gemm_wreg_launch(const bcg::GemmArgs& g, hipStream_t stream);
// gemm256.hip: large-M dense projections on 256 x 256 tiles, 8 waves, 8-phase LDS-DMA pipeline, persistent (BC_TILE_G256).
int bc_gemm256_ok(const BcGemm& p);
int bc_gemm256_launch(const bcg::GemmArgs& g, hipStream_t stream);
