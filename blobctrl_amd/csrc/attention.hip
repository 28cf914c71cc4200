// Flash attention forward for gfx950 (wave64, v_mfma_f32_32x32x16_f16), head_dim D in {8,16,32,40,64,80,160}.
//
// Replaces F.scaled_dot_product_attention at D/models/attention_processor.py:2216-2218 (UNet/BlobNet self-attention with
// N = 8192/2048/512/128 tokens and head_dim 40/80/160, UNet cross-attention to 77 CLIP tokens) and the eager attention of
// transformers' Dinov2SelfAttention (257 tokens, head_dim 64).
//
// Structure (per workgroup: 4 waves x 32 query rows, KV tiles of 64 keys staged in LDS):
//   S^T = K . Q^T   "swapped" product: MFMA A-operand = K rows from LDS, B-operand = Q rows held in registers, so each
//                   lane owns ONE query column and 16 keys per 32-key tile -> row max / sum are in-lane + one xor-32 shuffle.
//   P   = exp2(S^T - m)   (scale * log2(e) is folded into Q once), fp32 online softmax.
//   O^T += V^T . P^T      the S^T accumulator registers, converted to fp16, ARE the B operand of the next MFMA
//                         (cdna_hip_programming.md section 3 "An accumulator tile as the next MFMA's operand"); the matching
//                         k-permutation is applied to the V^T fragment read (two ds_read_b64 per fragment).
//   V arrives TRANSPOSED from the to_v GEMM epilogue (BC_OUT_F16_T), so the V^T tile is a coalesced row copy.
//   When D is not a multiple of 32 the padded V^T tile carries a row of ones, which makes the MFMA produce the softmax
//   denominator for free (removes 32 v_add per tile from the VALU-bound D=40 case).
#include <type_traits>
#include "bc_common.h"

namespace {

constexpr int QW = 32;        // queries per wave
constexpr int KVT = 64;       // keys per tile

template <int D>
struct AttnCfg {
    static constexpr int D16 = (D + 15) / 16;          // QK^T k-steps
    static constexpr int DK = D16 * 16;                // padded head_dim for QK^T
    static constexpr int DT = (D + 31) / 32;           // O^T row tiles
    static constexpr int DP = DT * 32;
    static constexpr bool ONES = (D % 32) != 0;        // spare padded V^T row available for the row sum
    static constexpr bool BIAS = (D % 16) != 0;        // spare padded K column available: K[:, D] = 1, Q[:, D] = -m_ref
    static constexpr int K_STRIDE = DK + 8;            // halfs; (DK/8 + 1) odd 16-byte slots -> conflict-free ds_read_b128
    static constexpr int V_STRIDE = KVT + 4;           // halfs; 17 x 8-byte slots -> conflict-free ds_read_b64
    static constexpr int STAGE_HALFS = KVT * K_STRIDE + DP * V_STRIDE;
    static constexpr int LDS_BYTES = 2 * STAGE_HALFS * 2;          // two stages
};

constexpr float RESCALE_THR = 6.0f;    // log2 units: P <= 64 between reference updates (fp16 P, fp32 accumulation)

// NW = waves (x 32 queries) per workgroup.
//
// Pipeline: K / V^T tiles are double-buffered in LDS; the global loads of tile t+1 are issued into registers before tile t is
// multiplied and written to the other LDS stage afterwards (one barrier per tile, HBM/L2 latency hidden under the MFMAs).
// Softmax reference: instead of subtracting the running maximum from every score (32 VALU ops per tile and lane), the kernel
// keeps a per-query reference m_ref and, when head_dim leaves a padded K column (D = 40: columns 40..47), lets the MFMA do the
// subtraction: K[:, D] = 1 and Q[:, D] = -m_ref (fp16; any consistent reference is valid for online softmax).  m_ref only moves
// when a score exceeds it by more than RESCALE_THR (wave-uniform slow path), so the steady state is exp2 + max + pack only.
template <int D, int NW>
__global__ __launch_bounds__(64 * NW) void attn_fwd_kernel(const h16* __restrict__ Q, const h16* __restrict__ K,
                                                         const h16* __restrict__ Vt, h16* __restrict__ O, int Nq, int Nkv,
                                                         int ldq, int ldk, int ldvt, int ldo, long long q_bs, long long k_bs,
                                                         long long vt_bs, long long o_bs, float scale_log2e) {
    using C = AttnCfg<D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    h16* lds = reinterpret_cast<h16*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qcol = lane & 31, half = lane >> 5;
    // XCD-aware remap: an XCD's L2 then serves the K / V of a few (batch, head) pairs to all of their query blocks
    const int nwg = gridDim.x * gridDim.y * gridDim.z;
    const int lin = bc_xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), nwg);
    const int qblk = lin % (int)gridDim.x, head = (lin / (int)gridDim.x) % (int)gridDim.y, b = lin / (int)(gridDim.x * gridDim.y);
    constexpr int NT = 64 * NW;
    const int q0 = qblk * (QW * NW) + wave * QW;

    const h16* Qb = Q + (size_t)b * q_bs + (size_t)head * D;
    const h16* Kb = K + (size_t)b * k_bs + (size_t)head * D;
    const h16* Vb = Vt + (size_t)b * vt_bs + (size_t)head * D * ldvt;
    h16* Ob = O + (size_t)b * o_bs + (size_t)head * D;

    // ---- Q fragments (B operand): lane holds Q[q0 + qcol][16 s + 8 half .. +7], pre-scaled by scale*log2(e) ----
    h16x8 qf[C::D16];
    {
        const int q = q0 + qcol;
#pragma unroll
        for (int s = 0; s < C::D16; ++s) {
            int dcol = 16 * s + 8 * half;
            h16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (h16)0.f;
            if (q < Nq && dcol < D) {
                uint4 raw = bc_ld16(Qb + (size_t)q * ldq + dcol);
                const h16* h = reinterpret_cast<const h16*>(&raw);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (h16)((float)h[j] * scale_log2e);
            }
            qf[s] = v;
        }
    }
    // the bias slot: column D of the padded head_dim lives in fragment D/16, lane-half (D%16)/8, element D%8
    constexpr int BS = D / 16, BH = (D % 16) / 8, BJ = D % 8;

    // ---- constant parts of both LDS stages: K pad columns (column D = 1 when BIAS), V^T pad rows (row D = ones when ONES) ----
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        h16* ldsK = lds + st * C::STAGE_HALFS;
        h16* ldsV = ldsK + KVT * C::K_STRIDE;
        if (C::DK > D) {
            for (int i = tid; i < KVT * (C::DK - D); i += NT) {
                int key = i / (C::DK - D), c = D + i % (C::DK - D);
                ldsK[key * C::K_STRIDE + c] = (C::BIAS && c == D) ? (h16)1.0f : (h16)0.f;
            }
        }
        if (C::DP > D) {
            for (int i = tid; i < (C::DP - D) * KVT; i += NT) {
                int r = D + i / KVT, c = i % KVT;
                ldsV[r * C::V_STRIDE + c] = (C::ONES && r == D) ? (h16)1.0f : (h16)0.f;
            }
        }
    }

    // ---- register prefetch of one K / V^T tile ----
    constexpr int KCH = KVT * (D / 8), VCH = D * (KVT / 8);
    constexpr int KL = (KCH + NT - 1) / NT, VL = (VCH + NT - 1) / NT;
    uint4 kreg[KL], vreg[VL];
    auto load_tile = [&](const int kbase, auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
#pragma unroll
        for (int i = 0; i < KL; ++i) {
            const int idx = tid + i * NT;
            const int key = idx / (D / 8), ch = idx % (D / 8);
            uint4 v = make_uint4(0, 0, 0, 0);
            if (idx < KCH && (!MASKED || kbase + key < Nkv)) v = bc_ld16(Kb + (size_t)(kbase + key) * ldk + ch * 8);
            kreg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < VL; ++i) {
            const int idx = tid + i * NT;
            const int r = idx / (KVT / 8), ch = idx % (KVT / 8);
            uint4 v = make_uint4(0, 0, 0, 0);
            if (idx < VCH) v = bc_ld16(Vb + (size_t)r * ldvt + kbase + ch * 8);   // Vt is zero-padded beyond Nkv by contract
            vreg[i] = v;
        }
    };
    auto store_tile = [&](const int st) {
        h16* ldsK = lds + st * C::STAGE_HALFS;
        h16* ldsV = ldsK + KVT * C::K_STRIDE;
#pragma unroll
        for (int i = 0; i < KL; ++i) {
            const int idx = tid + i * NT;
            const int key = idx / (D / 8), ch = idx % (D / 8);
            if (idx < KCH) bc_st16(ldsK + key * C::K_STRIDE + ch * 8, kreg[i]);
        }
#pragma unroll
        for (int i = 0; i < VL; ++i) {
            const int idx = tid + i * NT;
            const int r = idx / (KVT / 8), ch = idx % (KVT / 8);
            if (idx < VCH) {      // V_STRIDE*2 bytes = 136 is only 8-byte aligned: two 8-byte stores
                uint2* dst = reinterpret_cast<uint2*>(ldsV + r * C::V_STRIDE + ch * 8);
                dst[0] = make_uint2(vreg[i].x, vreg[i].y);
                dst[1] = make_uint2(vreg[i].z, vreg[i].w);
            }
        }
    };

    f32x16 oacc[C::DT];
#pragma unroll
    for (int t = 0; t < C::DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
    float m_ref = 0.f, l_run = 0.f;     // m_ref: reference exponent of this query's running softmax (exactly fp16-representable)
    bool first = true;

    typedef float f32x2 __attribute__((ext_vector_type(2)));
    union PFrag { h16x8 v; h16x2 p[4]; };

    auto compute_tile = [&](const int st, const int kbase, auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        const h16* ldsK = lds + st * C::STAGE_HALFS;
        const h16* ldsV = ldsK + KVT * C::K_STRIDE;
        // ---- S'^T[kt] = K_tile[kt] . Q'^T  (two 32-key tiles); with BIAS the product already holds S - m_ref ----
        f32x16 sacc[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kt][r] = 0.f;
#pragma unroll
            for (int s = 0; s < C::D16; ++s) {
                const h16x8 kf = *reinterpret_cast<const h16x8*>(ldsK + (kt * 32 + qcol) * C::K_STRIDE + 16 * s + 8 * half);
                sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], sacc[kt], 0, 0, 0);
            }
        }
        if (!C::BIAS) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[kt][r] -= m_ref;
        }
        if (MASKED) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int key = kbase + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (key >= Nkv) sacc[kt][r] = -INFINITY;
                }
        }
        // ---- running reference: per query column; partner lane = lane ^ 32 holds the other 32 keys ----
        float mx = sacc[0][0];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[kt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const bool move = first || (mx > RESCALE_THR);
        if (__any(move)) {
            // slow path (wave-uniform): move the reference of the lanes that need it, rescale O and l, shift this tile's scores
            const float want = move ? (m_ref + mx) : m_ref;                 // every tile holds >= 1 valid key: mx is finite
            const float m_new = (float)(h16)want;                            // keep the reference fp16-exact (it rides in Q)
            const float delta = m_new - m_ref;
            const float alpha = __builtin_amdgcn_exp2f(-delta);
            m_ref = m_new;
            if (C::BIAS) qf[BS][BJ] = (half == BH) ? (h16)(-m_new) : qf[BS][BJ];
            l_run *= alpha;
#pragma unroll
            for (int t = 0; t < C::DT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[kt][r] -= delta;
            first = false;
        }
        float psum = 0.f;
        PFrag pf[2][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    f32x2 e;
                    e.x = __builtin_amdgcn_exp2f(sacc[kt][8 * s2 + 2 * jj]);
                    e.y = __builtin_amdgcn_exp2f(sacc[kt][8 * s2 + 2 * jj + 1]);
                    pf[kt][s2].p[jj] = __builtin_convertvector(e, h16x2);
                    if (!C::ONES) psum += e.x + e.y;
                }
        if (!C::ONES) l_run += psum + __shfl_xor(psum, 32);

        // ---- O^T[t] += V^T_tile[t] . P^T ;  A fragment element j of lane-half h must be key 16 s2 + 8 (j>>2) + 4 h + (j&3) ----
#pragma unroll
        for (int t = 0; t < C::DT; ++t) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const h16* src = ldsV + (t * 32 + qcol) * C::V_STRIDE + kt * 32 + 16 * s2 + 4 * half;
                    h16x4 lo = *reinterpret_cast<const h16x4*>(src);
                    h16x4 hi = *reinterpret_cast<const h16x4*>(src + 8);
                    h16x8 vf;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { vf[j] = lo[j]; vf[4 + j] = hi[j]; }
                    oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kt][s2].v, oacc[t], 0, 0, 0);
                }
        }
    };

    const int nfull = Nkv / KVT;
    const int ntiles = (Nkv + KVT - 1) / KVT;
    // prologue: tile 0 -> stage 0
    if (nfull > 0) load_tile(0, std::false_type{}); else load_tile(0, std::true_type{});
    store_tile(0);
    __syncthreads();
    for (int tile = 0; tile < ntiles; ++tile) {
        const int st = tile & 1;
        const int nxt = tile + 1;
        if (nxt < ntiles) {                                   // global loads of the next tile fly during this tile's MFMAs
            if (nxt < nfull) load_tile(nxt * KVT, std::false_type{}); else load_tile(nxt * KVT, std::true_type{});
        }
        if (tile < nfull) compute_tile(st, tile * KVT, std::false_type{}); else compute_tile(st, tile * KVT, std::true_type{});
        if (nxt < ntiles) store_tile(st ^ 1);                 // the other stage was last read one barrier ago
        __syncthreads();
    }

    // ---- epilogue: O[q][dd] = O^T[dd][q] / l ----
    float l = l_run;
    if (C::ONES) {
        // the ones row is row D of the padded V^T: tile D/32, in-tile row i = D%32 -> register r with
        // (r&3) + 8 (r>>2) = i - 4 h  on lane-half h = (i>>2)&1
        constexpr int TI = D / 32, I = D % 32, H = (I >> 2) & 1, RI = (I & 3) + 4 * (I >> 3);
        float mine = (half == H) ? oacc[TI][RI] : 0.f;
        l = mine + __shfl_xor(mine, 32);
    }
    const float inv = 1.0f / l;
    const int q = q0 + qcol;
    if (q < Nq) {
#pragma unroll
        for (int t = 0; t < C::DT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int dd = t * 32 + 8 * g + 4 * half;          // rows (r&3) + 8 (r>>2) + 4 half, r = 4 g .. 4 g + 3
                if (dd < D) {
                    h16x4 o4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o4[j] = (h16)(oacc[t][4 * g + j] * inv);
                    *reinterpret_cast<h16x4*>(Ob + (size_t)q * ldo + dd) = o4;
                }
            }
    }
}

template <int D, int NW>
int launch_attn_nw(const h16* Q, const h16* K, const h16* Vt, h16* O, int B, int heads, int Nq, int Nkv, int ldq, int ldk,
                   int ldvt, int ldo, long long qbs, long long kbs, long long vbs, long long obs, float scale,
                   hipStream_t stream) {
    using C = AttnCfg<D>;
    dim3 grid(bc_ceil_div(Nq, QW * NW), heads, B), block(64 * NW);
    static bool attr_set = false;
    if (!attr_set) {
        BC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<D, NW>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
        attr_set = true;
    }
    hipLaunchKernelGGL((attn_fwd_kernel<D, NW>), grid, block, C::LDS_BYTES, stream, Q, K, Vt, O, Nq, Nkv, ldq, ldk, ldvt, ldo,
                       qbs, kbs, vbs, obs, scale * 1.4426950408889634f);
    BC_CHECK_LAUNCH();
    return 0;
}

template <int D>
int launch_attn(const h16* Q, const h16* K, const h16* Vt, h16* O, int B, int heads, int Nq, int Nkv, int ldq, int ldk,
                int ldvt, int ldo, long long qbs, long long kbs, long long vbs, long long obs, float scale,
                hipStream_t stream) {
    // Measured on MI355X: splitting short sequences over more, smaller workgroups (NW = 2 / 1) is SLOWER (every workgroup
    // re-stages the whole K / V with fewer threads: D=160, N=512: 18 vs 35 TFLOP/s), so the 4-wave form is always used.
    return launch_attn_nw<D, 4>(Q, K, Vt, O, B, heads, Nq, Nkv, ldq, ldk, ldvt, ldo, qbs, kbs, vbs, obs, scale, stream);
}

}  // namespace

extern "C" int bc_attention(const bc_half* Q, const bc_half* K, const bc_half* Vt, bc_half* O, int B, int heads, int d,
                            int Nq, int Nkv, int ldq, int ldk, int ldvt, int ldo, long long q_bstride,
                            long long k_bstride, long long vt_bstride, long long o_bstride, float scale,
                            bc_stream stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    BC_CHECK_ARG(Q && K && Vt && O && B > 0 && heads > 0 && Nq > 0 && Nkv > 0, "bc_attention: bad args");
    BC_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && ldo % 4 == 0, "bc_attention: strides must be multiples of 8 (ldo: 4)");
    BC_CHECK_ARG(ldvt >= bc_ceil_div(Nkv, KVT) * KVT, "bc_attention: ldvt=%d must cover Nkv=%d rounded up to %d (zero padded)", ldvt, Nkv, KVT);
    BC_CHECK_ARG(q_bstride % 8 == 0 && k_bstride % 8 == 0 && vt_bstride % 8 == 0 && o_bstride % 4 == 0, "bc_attention: batch strides must be multiples of 8");
    const h16* q = reinterpret_cast<const h16*>(Q);
    const h16* k = reinterpret_cast<const h16*>(K);
    const h16* v = reinterpret_cast<const h16*>(Vt);
    h16* o = reinterpret_cast<h16*>(O);
#define BC_ATTN_CASE(DD)                                                                                         \
    case DD:                                                                                                     \
        return launch_attn<DD>(q, k, v, o, B, heads, Nq, Nkv, ldq, ldk, ldvt, ldo, q_bstride, k_bstride, vt_bstride, \
                               o_bstride, scale, stream);
    switch (d) {
        BC_ATTN_CASE(8)
        BC_ATTN_CASE(16)
        BC_ATTN_CASE(32)
        BC_ATTN_CASE(40)
        BC_ATTN_CASE(64)
        BC_ATTN_CASE(80)
        BC_ATTN_CASE(160)
        default:
            bc_set_error("bc_attention: head_dim %d not instantiated (supported: 8,16,32,40,64,80,160)", d);
            return 1;
    }
#undef BC_ATTN_CASE
}
