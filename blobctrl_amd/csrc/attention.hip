// Flash attention forward for gfx950 (wave64, v_mfma_f32_32x32x16_f16), head_dim D in {8,16,32,40,64,80,160}.
//
// Replaces F.scaled_dot_product_attention at D/models/attention_processor.py:2216-2218 (UNet/BlobNet self-attention with
// N = 8192/2048/512/128 tokens and head_dim 40/80/160, UNet cross-attention to 77 CLIP tokens) and the eager attention of
// transformers' Dinov2SelfAttention (257 tokens, head_dim 64).
//
// Structure (per workgroup: 4 waves x 1 or 2 blocks of 32 query rows, KV tiles of 64 keys staged in LDS):
//   S^T = K . Q^T   "swapped" product: MFMA A-operand = K rows from LDS, B-operand = Q rows held in registers, so each
//                   lane owns ONE query column and 16 keys per 32-key tile -> row max / sum are in-lane + one xor-32 shuffle.
//   P   = exp2(S^T - m)   (scale * log2(e) is folded into Q once), fp32 online softmax.
//   O^T += V^T . P^T      the S^T accumulator registers, converted to fp16, ARE the B operand of the next MFMA
//                         (cdna_hip_programming.md section 3 "An accumulator tile as the next MFMA's operand"); the K rows
//                         are fed in a bit-swapped order so that the matching V^T fragment is ONE aligned ds_read_b128.
//   V arrives TRANSPOSED from the to_v GEMM epilogue (BC_OUT_F16_T), so the V^T tile is a coalesced row copy.
//   When D is not a multiple of 32 the padded V^T tile carries a row of ones, which makes the MFMA produce the softmax
//   denominator for free (removes 32 v_add per tile from the VALU-bound D=40 case).
#include <stdlib.h>
#include <type_traits>
#include "bc_common.h"

namespace {

constexpr int QW = 32;        // queries per query block (one MFMA column tile)
constexpr int KVT = 64;       // keys per tile

template <int D>
struct AttnCfg {
    static constexpr int D16 = (D + 15) / 16;          // QK^T k-steps
    static constexpr int DK = D16 * 16;                // padded head_dim for QK^T
    static constexpr int DT = (D + 31) / 32;           // O^T row tiles
    static constexpr int DP = DT * 32;
    static constexpr bool ONES = (D % 32) != 0;        // spare padded V^T row available for the row sum
    static constexpr bool BIAS = (D % 16) != 0;        // spare padded K column available: K[:, D] = 1, Q[:, D] = -m_ref
    // LDS images are filled by LDS-DMA (buffer_load ... lds: 64 lanes x 16 bytes land lane-linearly), so rows cannot be padded
    // by the store; instead a row carries a dummy 16-byte chunk when needed to make its stride an ODD number of chunks, which is
    // what makes the ds_read_b128 fragment reads conflict-free.
    static constexpr int DC = D / 8;                   // real chunks per K row
    static constexpr int KC = (DC % 2) ? DC : DC + 1;  // K row stride in chunks
    static constexpr int VC = KVT / 8 + 1;             // V^T row stride in chunks (8 real + 1 dummy)
    static constexpr int KPIECES = KC;                 // 64 keys x KC chunks = KC pieces of 64 chunks (1 KiB)
    static constexpr int VPIECES = (D * VC + 63) / 64;
    static constexpr int K_BYTES = KPIECES * 1024;
    static constexpr int STAGE_BYTES = K_BYTES + VPIECES * 1024;
    // constants behind the two stages: 128 B of ones, 128 B of zeros (padded V^T rows), and the K bias chunk [1,0,..,0] twice,
    // KSPAN apart (the kt = 1 fragment read adds KSPAN to the address as an immediate)
    static constexpr int KSPAN = 32 * KC * 16;
    static constexpr int CONST_OFF = 2 * STAGE_BYTES;
    static constexpr int KCONST_OFF = CONST_OFF + 256;
    static constexpr int LDS_BYTES = KCONST_OFF + (BIAS ? KSPAN + 16 : 0);
};

constexpr float RESCALE_THR = 6.0f;    // log2 units: P <= 64 between reference updates (fp16 P, fp32 accumulation)

// NOTE: no inline asm on MFMA results - the compiler's hazard recogniser does not see inside asm strings and would not insert the
// wait states an MFMA -> VALU read needs (measured: wrong maxima on the first registers of a tile).
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
// max over the two 32-lane halves (lane l <-> lane l^32) with v_permlane32_swap: VALU-only, no LDS crossbar round trip
__device__ __forceinline__ float halves_max(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float halves_sum(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// NW = waves per workgroup, QB = 32-query blocks per wave (a wave's K / V^T fragment reads from LDS are shared by its QB blocks
// and the blocks' MFMA and softmax chains are independent, so the scheduler can run one block's exp2 under the other's MFMAs).
//
// Pipeline: K / V^T tiles are double-buffered in LDS; the global loads of tile t+1 are issued into registers before tile t is
// multiplied and written to the other LDS stage afterwards (one barrier per tile, HBM/L2 latency hidden under the MFMAs).  Each
// (wave, load slot) stages either a K piece or a V^T piece - a wave-uniform choice, so the staging code has no divergent
// branches.
// Key permutation: MFMA row i of the S^T tile is fed key pi(i) = i with bits 2 and 3 swapped.  The S^T accumulator registers
// 8 s2 .. 8 s2 + 7 of lane-half h then hold the 8 CONSECUTIVE keys 16 s2 + 8 h .. + 7, so the matching V^T fragment is one
// aligned ds_read_b128.
// Softmax reference: instead of subtracting the running maximum from every score (32 VALU ops per tile and lane), the kernel
// keeps a per-query reference m_ref and, when head_dim leaves a padded K column (D = 40: columns 40..47), lets the MFMA do the
// subtraction: K[:, D] = 1 and Q[:, D] = -m_ref (fp16; any consistent reference is valid for online softmax).  m_ref only moves
// when a score exceeds it by more than RESCALE_THR (wave-uniform slow path), so the steady state is exp2 + max + pack only.
// WPE = minimum waves per SIMD the register allocation must allow (1 = unconstrained).  WPE = 4 caps D <= 40 at 128 VGPRs: the
// main loop still fits (the few spills land in the ragged / causal tail), and a fourth resident wave per SIMD is worth +5 % when
// the grid is large enough to fill it (measured 647 vs 612 TFLOP/s at d=40, N=8192, CFG batch 2).
template <int D, int NW, int QB, int WPE = 1>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(WPE))) void attn_fwd_kernel(const h16* __restrict__ Q, const h16* __restrict__ K,
                                                         const h16* __restrict__ Vt, h16* __restrict__ O, int Nq, int Nkv,
                                                         int ldq, int ldk, int ldvt, int ldo, long long q_bs, long long k_bs,
                                                         long long vt_bs, long long o_bs, float scale_log2e, int causal) {
#if defined(__HIP_DEVICE_COMPILE__)    // the body uses device-only builtins (buffer resources, LDS-DMA): the host pass only needs the stub
    using C = AttnCfg<D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qcol = lane & 31, half = lane >> 5;
    const int krow = (qcol & ~12) | ((qcol & 4) << 1) | ((qcol & 8) >> 1);      // pi(qcol)
    // XCD-aware remap: an XCD's L2 then serves the K / V of a few (batch, head) pairs to all of their query blocks
    const int nwg = gridDim.x * gridDim.y * gridDim.z;
    const int lin = bc_xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), nwg);
    const int qblk = lin % (int)gridDim.x, head = (lin / (int)gridDim.x) % (int)gridDim.y, b = lin / (int)(gridDim.x * gridDim.y);
    constexpr int NT = 64 * NW;
    const int q0 = qblk * (QW * QB * NW) + wave * (QW * QB);

    const h16* Qb = Q + (size_t)b * q_bs + (size_t)head * D;
    const h16* Kb = K + (size_t)b * k_bs + (size_t)head * D;
    const h16* Vb = Vt + (size_t)b * vt_bs + (size_t)head * D * ldvt;
    h16* Ob = O + (size_t)b * o_bs + (size_t)head * D;

    // ---- Q fragments (B operand): lane holds Q[q][16 s + 8 half .. +7], pre-scaled by scale*log2(e) ----
    // (round 6: every chunk is requested from a clamped address before the first is converted.  With the loads under `if (q < Nq && dcol < D)` the
    //  compiler kept each chunk's conversion in the load's block and awaited every request where it was made: D / 16 + 1 serial round trips to
    //  memory in front of every workgroup's first key tile - ten of them at D = 160)
    h16x8 qf[QB][C::D16];
    uint4 qraw[QB][C::D16];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int q = q0 + qb * QW + qcol, qc = q < Nq ? q : Nq - 1;
#pragma unroll
        for (int s = 0; s < C::D16; ++s) {
            const int dcol = 16 * s + 8 * half;
            qraw[qb][s] = bc_ld16(Qb + (size_t)qc * ldq + (dcol < D ? dcol : 0));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int q = q0 + qb * QW + qcol;
#pragma unroll
        for (int s = 0; s < C::D16; ++s) {
            const bool ok = q < Nq && 16 * s + 8 * half < D;
            const h16* h = reinterpret_cast<const h16*>(&qraw[qb][s]);
            h16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = ok ? (h16)((float)h[j] * scale_log2e) : (h16)0.f;
            qf[qb][s] = v;
        }
    }
    // the bias slot: column D of the padded head_dim lives in fragment D/16, lane-half (D%16)/8, element D%8
    constexpr int BS = D / 16, BH = (D % 16) / 8, BJ = D % 8;

    // ---- constants in LDS ----
    char* const smem_c = smem;
    for (int i = tid; i < 128; i += NT) {
        reinterpret_cast<h16*>(smem_c + C::CONST_OFF)[i] = i < 64 ? (h16)1.0f : (h16)0.f;
    }
    if (C::BIAS && tid < 16) {
        const h16 v = (tid & 7) == 0 ? (h16)1.0f : (h16)0.f;
        reinterpret_cast<h16*>(smem_c + C::KCONST_OFF + (tid >> 3) * C::KSPAN)[tid & 7] = v;
    }

    // ---- LDS-DMA of one K / V^T tile: pieces of 64 x 16 bytes, piece p < KPIECES is K, the rest V^T.  Per lane only the byte
    // offset inside the (batch, head) buffer is kept; the tile advances through the SGPR offset, and the buffer range check
    // returns zeros for K rows past Nkv and for the dummy chunks (offset = "far out of range"). ----
    constexpr int TP = C::KPIECES + C::VPIECES;
    constexpr int NLD = (TP + NW - 1) / NW;
    constexpr int OOB = (int)0x7ffffff0;
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16*>(Kb), 0, (int)(((long long)(Nkv - 1) * ldk + D) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16*>(Vb), 0, (int)((long long)D * ldvt * 2), 0x00020000);
    int dvoff[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int piece = i * NW + wave;                   // wave-uniform
        if (piece < C::KPIECES) {
            const int q = piece * 64 + lane, key = q / C::KC, c = q % C::KC;
            dvoff[i] = c < C::DC ? (key * ldk + c * 8) * 2 : OOB;
        } else {
            const int q = (piece - C::KPIECES) * 64 + lane, r = q / C::VC, c = q % C::VC;
            dvoff[i] = (r < D && c < KVT / 8) ? (r * ldvt + c * 8) * 2 : OOB;
        }
    }
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto issue_tile = [&](const int tile, const int st) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int piece = i * NW + wave;
            if (piece < C::KPIECES) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lds_ptr)(smem + st * C::STAGE_BYTES + piece * 1024), 16, dvoff[i],
                                                         tile * (KVT * 2) * ldk, 0, 0);
            } else if (piece < TP) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lds_ptr)(smem + st * C::STAGE_BYTES + piece * 1024), 16, dvoff[i],
                                                         tile * (KVT * 2), 0, 0);
            }
        }
    };
    // ---- per-lane fragment read addresses (bytes) ----
    int kaddr[2], klast[2], vaddr[2][C::DT];
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        kaddr[st] = st * C::STAGE_BYTES + krow * (C::KC * 16) + half * 16;
        klast[st] = (C::BIAS && half == 1) ? C::KCONST_OFF - (C::D16 - 1) * 32 : kaddr[st];
#pragma unroll
        for (int t = 0; t < C::DT; ++t) {
            const int row = t * 32 + qcol;
            vaddr[st][t] = row < D ? st * C::STAGE_BYTES + C::K_BYTES + row * (C::VC * 16) + half * 16
                                   : ((C::ONES && row == D) ? C::CONST_OFF : C::CONST_OFF + 128);
        }
    }

    f32x16 oacc[QB][C::DT];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int t = 0; t < C::DT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[qb][t][r] = 0.f;
    float m_ref[QB], l_run[QB];     // m_ref: reference exponent of a query's running softmax (exactly fp16-representable)
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) { m_ref[qb] = 0.f; l_run[qb] = 0.f; }
    bool first = true;

    typedef float f32x2 __attribute__((ext_vector_type(2)));
    union PFrag { h16x8 v; h16x2 p[4]; };

    auto compute_tile = [&](auto stage_tag, const int kbase, auto masked_tag) {
        constexpr int st = decltype(stage_tag)::value;        // compile-time stage: the fragment addresses stay in fixed registers
        constexpr bool MASKED = decltype(masked_tag)::value;
        // ---- S'^T[qb][kt] = K_tile[kt] . Q'^T[qb]  (two 32-key tiles); with BIAS the product already holds S - m_ref ----
        f32x16 sacc[QB][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int s = 0; s < C::D16; ++s) {
                const int ka = (C::BIAS && s == C::D16 - 1) ? klast[st] : kaddr[st];
                const h16x8 kf = *reinterpret_cast<const h16x8*>(smem + ka + kt * C::KSPAN + s * 32);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    if (s == 0) {
                        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        sacc[qb][kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[qb][s], z, 0, 0, 0);
                    } else {
                        sacc[qb][kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[qb][s], sacc[qb][kt], 0, 0, 0);
                    }
                }
            }
        }
        PFrag pf[QB][2][2];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            if (!C::BIAS) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc[qb][kt][r] -= m_ref[qb];
            }
            if (MASKED) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        int key = kbase + kt * 32 + 16 * (r >> 3) + 8 * half + (r & 7);     // pi applied
                        if (key >= Nkv || (causal && key > q0 + qb * QW + qcol)) sacc[qb][kt][r] = -INFINITY;
                    }
            }
            // ---- running reference: per query column; partner lane = lane ^ 32 holds the other 32 keys ----
            float mx = max3f(sacc[qb][0][0], sacc[qb][0][1], sacc[qb][0][2]);
            mx = max3f(mx, sacc[qb][0][3], sacc[qb][0][4]);
#pragma unroll
            for (int r = 5; r < 15; r += 2) mx = max3f(mx, sacc[qb][0][r], sacc[qb][0][r + 1]);
            float mx1 = max3f(sacc[qb][1][0], sacc[qb][1][1], sacc[qb][1][2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) mx1 = max3f(mx1, sacc[qb][1][r], sacc[qb][1][r + 1]);
            mx = max3f(mx, mx1, sacc[qb][0][15]);
            mx = max3f(mx, sacc[qb][1][15], sacc[qb][1][15]);
            mx = halves_max(mx);
            const bool move = first || (mx > RESCALE_THR);
            if (__any(move)) {
                // slow path (wave-uniform): move the reference of the lanes that need it, rescale O and l, shift this tile's scores
                const float want = move ? (m_ref[qb] + mx) : m_ref[qb];         // every tile holds >= 1 valid key: mx is finite
                const float m_new = (float)(h16)want;                            // keep the reference fp16-exact (it rides in Q)
                const float delta = m_new - m_ref[qb];
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                m_ref[qb] = m_new;
                if (C::BIAS) qf[qb][BS][BJ] = (half == BH) ? (h16)(-m_new) : qf[qb][BS][BJ];
                l_run[qb] *= alpha;
#pragma unroll
                for (int t = 0; t < C::DT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[qb][t][r] *= alpha;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc[qb][kt][r] -= delta;
            }
            float psum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        f32x2 e;
                        e.x = __builtin_amdgcn_exp2f(sacc[qb][kt][8 * s2 + 2 * jj]);
                        e.y = __builtin_amdgcn_exp2f(sacc[qb][kt][8 * s2 + 2 * jj + 1]);
                        pf[qb][kt][s2].p[jj] = __builtin_convertvector(e, h16x2);
                        if (!C::ONES) psum += e.x + e.y;
                    }
            if (!C::ONES) l_run[qb] += halves_sum(psum);
        }
        first = false;

        // ---- O^T[qb][t] += V^T_tile[t] . P^T[qb] ;  A fragment = V^T[t*32 + row][16 s2 + 8 half .. + 7] (one b128) ----
#pragma unroll
        for (int t = 0; t < C::DT; ++t) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const h16x8 vf = *reinterpret_cast<const h16x8*>(smem + vaddr[st][t] + kt * 64 + s2 * 32);
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb)
                        oacc[qb][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[qb][kt][s2].v, oacc[qb][t], 0, 0, 0);
                }
        }
    };

    // causal: every tile takes the masked path, and tiles wholly above the diagonal of this workgroup's last query are skipped
    const int ntiles = causal ? min((Nkv + KVT - 1) / KVT, (min(Nq, qblk * (QW * QB * NW) + QW * QB * NW) + KVT - 1) / KVT)
                              : (Nkv + KVT - 1) / KVT;
    const int nfull = causal ? 0 : Nkv / KVT;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    // prologue: tile 0 -> stage 0
    issue_tile(0, 0);
    __syncthreads();                                          // (waits for this wave's DMA, then for everybody's)
    int tile = 0;
    for (; tile + 1 < nfull; tile += 2) {                     // two tiles per trip: the stage is a compile-time constant
        issue_tile(tile + 1, 1);                              // the other stage was last read one barrier ago
        compute_tile(S0{}, tile * KVT, std::false_type{});
        __syncthreads();
        if (tile + 2 < ntiles) issue_tile(tile + 2, 0);
        compute_tile(S1{}, (tile + 1) * KVT, std::false_type{});
        __syncthreads();
    }
    if (tile < nfull) {                                       // odd number of full tiles: one more at stage 0
        if (tile + 1 < ntiles) issue_tile(tile + 1, 1);
        compute_tile(S0{}, tile * KVT, std::false_type{});
        __syncthreads();
        ++tile;
    }
    for (; tile < ntiles; ++tile) {                           // ragged tail (one tile) or the causal tiles: masked scores
        if (tile + 1 < ntiles) issue_tile(tile + 1, (tile + 1) & 1);
        if (tile & 1) compute_tile(S1{}, tile * KVT, std::true_type{});
        else compute_tile(S0{}, tile * KVT, std::true_type{});
        __syncthreads();
    }

    // ---- epilogue: O[q][dd] = O^T[dd][q] / l ----
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        float l = l_run[qb];
        if (C::ONES) {
            // the ones row is row D of the padded V^T: tile D/32, in-tile row i = D%32 -> register r with
            // (r&3) + 8 (r>>2) = i - 4 h  on lane-half h = (i>>2)&1
            constexpr int TI = D / 32, I = D % 32, H = (I >> 2) & 1, RI = (I & 3) + 4 * (I >> 3);
            float mine = (half == H) ? oacc[qb][TI][RI] : 0.f;
            l = halves_sum(mine);
        }
        const float inv = 1.0f / l;
        const int q = q0 + qb * QW + qcol;
        if (q < Nq) {
#pragma unroll
            for (int t = 0; t < C::DT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    int dd = t * 32 + 8 * g + 4 * half;          // rows (r&3) + 8 (r>>2) + 4 half, r = 4 g .. 4 g + 3
                    if (dd < D) {
                        h16x4 o4;
#pragma unroll
                        for (int j = 0; j < 4; ++j) o4[j] = (h16)(oacc[qb][t][4 * g + j] * inv);
                        *reinterpret_cast<h16x4*>(Ob + (size_t)q * ldo + dd) = o4;
                    }
                }
        }
    }
#endif
}

template <int D, int NW, int QB, int WPE = 1>
int launch_attn_nw(const h16* Q, const h16* K, const h16* Vt, h16* O, int B, int heads, int Nq, int Nkv, int ldq, int ldk,
                   int ldvt, int ldo, long long qbs, long long kbs, long long vbs, long long obs, float scale,
                   int causal, hipStream_t stream) {
    using C = AttnCfg<D>;
    dim3 grid(bc_ceil_div(Nq, QW * QB * NW), heads, B), block(64 * NW);
    static std::atomic<unsigned long long> lds_set{0};       // one bit per device ordinal
    BC_CHECK_HIP(bc_set_max_lds(lds_set, reinterpret_cast<const void*>(&attn_fwd_kernel<D, NW, QB, WPE>), (int)C::LDS_BYTES));
    hipLaunchKernelGGL((attn_fwd_kernel<D, NW, QB, WPE>), grid, block, C::LDS_BYTES, stream, Q, K, Vt, O, Nq, Nkv, ldq, ldk, ldvt, ldo,
                       qbs, kbs, vbs, obs, scale * 1.4426950408889634f, causal);
    BC_CHECK_LAUNCH();
    return 0;
}

template <int D>
int launch_attn(const h16* Q, const h16* K, const h16* Vt, h16* O, int B, int heads, int Nq, int Nkv, int ldq, int ldk,
                int ldvt, int ldo, long long qbs, long long kbs, long long vbs, long long obs, float scale,
                int causal, hipStream_t stream) {
    // Measured on MI355X: splitting short sequences over more, smaller workgroups (NW = 2 / 1) is SLOWER (every workgroup
    // re-stages the whole K / V with fewer threads: D=160, N=512: 18 vs 35 TFLOP/s), so the 4-wave form is always used.
    // A second query block per wave (template parameter QB = 2: K / V fragments read once for 64 queries) was measured slower
    // than QB = 1 at every shape of the loop once the tiles arrive by LDS-DMA (d=40 N=8192: 556 vs 613 TFLOP/s; it costs
    // occupancy), so only QB = 1 is instantiated.
    if constexpr (D == 40) {
        // Round 2 (tools/attn_probe.py, L0 shape B=2 / 768^2 batch 4, TFLOP/s): 4 waves x 1 query block 633 / 775; 4 x 2 blocks
        // (each K / V^T fragment feeds two MFMAs: half the LDS reads) 634 / 793; 8 waves x 1 block (the tile is staged once for 256
        // queries) 659 / 804; 8 x 2 660 / 737.  Halving the LDS traffic buys nothing - the kernel is VALU-issue-bound - so the
        // 8-wave form is taken where the grid still gives every SIMD four waves, and the two-block forms are not instantiated.
        const long long wgs8 = (long long)bc_ceil_div(Nq, QW * 8) * heads * B;
        // Round 3: from ONE 8-wave workgroup per CU (was two): BlobNet's batch-1 self-attention at the 64 x 128 level (256 such
        // workgroups) then stages every K / V^T tile once per 256 queries too - step 10.59 -> 10.49 ms (same box, two rounds).
        static const bool no8 = getenv("BC_ATTN_NO8") != nullptr;      // (the 4-wave form for every grid: tests run both on one shape)
        if (wgs8 >= 256 && Nkv >= 1024 && !causal && !no8)
            return launch_attn_nw<D, 8, 1, 4>(Q, K, Vt, O, B, heads, Nq, Nkv, ldq, ldk, ldvt, ldo, qbs, kbs, vbs, obs, scale, causal, stream);
    }
    if constexpr (D == 80) {
        // Round 5: the 8-wave form at the 32 x 64 level too (2048 keys; the UNet's launch = 128 workgroups: a K / V^T tile staged once for 256
        // queries, two waves per SIMD on half the CUs instead of one wave per SIMD on all of them - their MFMA and softmax phases
        // interleave, and the other queue has the other CUs): six interleaved same-box pairs, ms per step, 4-wave vs 8-wave: 9.122 / 9.071,
        // 9.086 / 9.082, 9.094 / 9.076, 9.084 / 9.068, 8.937 / 8.887, 8.907 / 8.904.
        if (Nkv >= 1024 && !causal && Nq % (QW * 8) == 0)
            return launch_attn_nw<D, 8, 1>(Q, K, Vt, O, B, heads, Nq, Nkv, ldq, ldk, ldvt, ldo, qbs, kbs, vbs, obs, scale, causal, stream);
    }
    if constexpr (D <= 40) {
        // enough workgroups for 4 per CU and a long key loop: take the 128-VGPR build (4 waves per SIMD)
        const long long wgs = (long long)bc_ceil_div(Nq, QW * 4) * heads * B;
        if (wgs >= 4 * 256 && Nkv >= 1024 && !causal)
            return launch_attn_nw<D, 4, 1, 4>(Q, K, Vt, O, B, heads, Nq, Nkv, ldq, ldk, ldvt, ldo, qbs, kbs, vbs, obs, scale, causal, stream);
    }
    return launch_attn_nw<D, 4, 1>(Q, K, Vt, O, B, heads, Nq, Nkv, ldq, ldk, ldvt, ldo, qbs, kbs, vbs, obs, scale, causal, stream);
}

}  // namespace

static int attention_impl(const bc_half* Q, const bc_half* K, const bc_half* Vt, bc_half* O, int B, int heads, int d,
                            int Nq, int Nkv, int ldq, int ldk, int ldvt, int ldo, long long q_bstride,
                            long long k_bstride, long long vt_bstride, long long o_bstride, float scale,
                            int causal, bc_stream stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    BC_CHECK_ARG(Q && K && Vt && O && B > 0 && heads > 0 && Nq > 0 && Nkv > 0, "bc_attention: bad args");
    BC_CHECK_ARG(!causal || Nq == Nkv, "bc_attention_causal: needs Nq == Nkv (Nq=%d Nkv=%d)", Nq, Nkv);
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
                               o_bstride, scale, causal, stream);
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

extern "C" int bc_attention(const bc_half* Q, const bc_half* K, const bc_half* Vt, bc_half* O, int B, int heads, int d,
                            int Nq, int Nkv, int ldq, int ldk, int ldvt, int ldo, long long q_bstride,
                            long long k_bstride, long long vt_bstride, long long o_bstride, float scale,
                            bc_stream stream) {
    return attention_impl(Q, K, Vt, O, B, heads, d, Nq, Nkv, ldq, ldk, ldvt, ldo, q_bstride, k_bstride, vt_bstride, o_bstride,
                          scale, 0, stream);
}

extern "C" int bc_attention_causal(const bc_half* Q, const bc_half* K, const bc_half* Vt, bc_half* O, int B, int heads, int d,
                                   int Nq, int Nkv, int ldq, int ldk, int ldvt, int ldo, long long q_bstride,
                                   long long k_bstride, long long vt_bstride, long long o_bstride, float scale,
                                   bc_stream stream) {
    return attention_impl(Q, K, Vt, O, B, heads, d, Nq, Nkv, ldq, ldk, ldvt, ldo, q_bstride, k_bstride, vt_bstride, o_bstride,
                          scale, 1, stream);
}
