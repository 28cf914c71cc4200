// Row-chain kernel: the per-token part of a Transformer2D / BasicTransformerBlock as ONE launch per attention boundary.
//
// Reference ops (diffusers/src/diffusers/models/): transformers/transformer_2d.py:479-527 (GroupNorm -> proj_in ... proj_out +
// residual), attention.py:421-541 (norm1 -> attn1 -> +, norm2 -> attn2 -> +, norm3 -> ff -> +), attention_processor.py:2154-2236
// (to_q / to_k / to_v, to_out), activations.py:113-123 + attention.py:1161-1167 (GEGLU feed-forward); BlobNet's zero-conv
// (blobctrl/models/blobnet.py:860-864, 921-924, 936-938).  Everything between two attention calls acts on token rows independently,
// so a workgroup keeps a block of 64 token rows resident and runs the whole chain on it:
//   BC_CHAIN_IN   x -> GroupNorm affine -> proj_in -> h0 ; LayerNorm1 -> to_q | to_k (row-major) , to_v (written transposed)
//   BC_CHAIN_MID  attn1 out -> to_out + h0 -> h1 ; LayerNorm2 -> attn2.to_q                                        (UNet only)
//   BC_CHAIN_OUT  attn out -> to_out + h -> h2 ; LayerNorm3 -> GEGLU feed-forward (hidden in chunks of 128: the [rows x 4C]
//                 intermediate never exists) + h2 -> h3 ; proj_out + x (+ BlobNet right-half residual) -> out (+ GroupNorm partials)
//                 [BlobNet: -> zero-conv * conditioning scale -> residual tensor]
// replacing 14 launches (12 GEMMs, 3 LayerNorms, 1 GroupNorm pass) per UNet block by 3, and the HBM round trips between them.
//
// Structure (C = 320, 4 waves = 256 threads, 64 rows per workgroup, v_mfma_f32_16x16x32_f16):
//   * activations live in LDS as the B operand: X = [K/32][64 rows][32 k] fp16 (64-byte rows, 16-byte chunks XOR-swizzled so that
//     every ds_read_b128 fragment read is conflict-free); the product is SWAPPED (D^T = W . X^T) so that a lane ends up with 4
//     consecutive output channels of one token: 8-byte LDS writes into the next GEMM's operand image, row statistics by two
//     cross-lane adds;
//   * a wave owns 80 of the 320 output channels for all 64 rows (5 x 4 accumulator tiles), so weights are never shared between
//     waves: they stream HBM/L2 -> VGPR directly (no LDS, no barrier in any k-loop) from a per-wave stream that the host packs in
//     exact consumption order - every wave-instruction reads one contiguous KiB - through a register ring that stays R fragments
//     ahead of the MFMAs ACROSS GEMM boundaries;
//   * row-major global inputs / outputs go through an LDS staging image S in full 16-byte accesses (640-byte rows coalesced);
//   * LDS = X (40 KiB) + S (40 KiB; aliases the feed-forward's double-buffered hidden chunk) = 80 KiB: two workgroups per CU.
#include <stdlib.h>
#include <vector>
#include "bc_common.h"

namespace {

constexpr int RC_BM = 64;              // rows per workgroup
constexpr int RC_NT = 5;               // 16-channel tiles per wave of an N = C GEMM: a wave owns 80 output channels
constexpr int RC_HC = 128;             // feed-forward hidden chunk
constexpr int RC_RPAD = 20;            // padding fragments at the end of every wave's stream (>= the deepest ring)
// Ring depth R (fragments of 1 KiB per wave in flight ahead of the MFMAs): 13 for IN (12 at 640 channels; 14 spills three registers
// since the GroupNorm finalize moved into its prologue), 16 for MID, 15 for OUT (a feed-forward chunk
// must consume a whole number of rings): 56-64 KiB per 4 waves in flight, what ~70 GB/s per CU needs at ~1 us of L2 / HBM latency.

// Channel count C in {320, 640}: C / 80 waves (4 / 8), 64 rows per workgroup either way.
template <int C>
struct RCfg {
    static constexpr int NW = C / 80;                  // waves
    static constexpr int NTH = 64 * NW;                // threads
    static constexpr int KS = C / 32;                  // k-steps of a K = C GEMM
    static constexpr int NCH = 4 * C / RC_HC;          // feed-forward hidden chunks
    static constexpr int HW_ = RC_HC / NW;             // hidden units of a chunk per wave (32 / 16)
    static constexpr int TP = HW_ / 16;                // (value tile, gate tile) passes per chunk and wave (2 / 1)
    static constexpr int X_BYTES = KS * 4096;
    static constexpr int S_BYTES = RC_BM * C * 2;
    static constexpr int LDS = X_BYTES + S_BYTES;
    static constexpr int ROWS_PASS = NTH / 8;          // rows one pass of the copy threads covers (32 / 64)
    static constexpr int NPASS = RC_BM / ROWS_PASS;    // 2 / 1
    static constexpr int NKC = C / 64;                 // 16-byte chunks per row and thread (5 / 10)
    static constexpr int G = RC_NT * KS;               // weight fragments of one N = C GEMM per wave
};

struct RowChainArgs {
    int kind, M, rows_per_batch;
    const h16* x;            // IN: block input rows [M][C]; MID / OUT: attention output rows
    const float* affine;     // IN: GroupNorm affine [B][C][2] or null
    const unsigned long long* gn_in;   // IN: ... or the statistics totals of x [B][C][BC_GN_TOT_WORDS]: the finalize runs in the prologue
    const float* gn_gamma;   //     (gamma, beta, groups, eps of that GroupNorm)
    const float* gn_beta;
    int gn_groups;
    float gn_eps;
    const h16* res;          // MID / OUT: residual stream before this attention [M][C]
    const h16* res2;         // OUT: the block's input x (added after proj_out)
                             // MIDX (the struct is kept at its size: a larger one costs every kind scalar registers, and those spill into
                             // vector lanes): the K / V^T fragment streams [B][waves][XCfg::FR_WAVE + RC_RPAD][64 lanes] x 16 bytes
                             // (bc_rowchain_pack_kv); r2_xmin = context tokens (<= 80); alpha = softmax scale (head_dim ** -0.5)
    const h16* r2;           // OUT: BlobNet residual [bmod][rows_per_batch][C] added where pixel x >= r2_xmin, or null
    int r2_xmin, r2_bmod, out_w;
    const uint4* wstream;    // [4 waves][fragments in consumption order (+ RC_R of padding)][64 lanes] 16-byte fragments
    long long wave_frags;    // fragments per wave stream (incl. padding)
    const float* vec;        // fp32 vectors (biases, LayerNorm affine) in consumption order
    h16* out0;               // IN: h0 [M][C]; MID: h1; OUT: block output
    h16* out1;               // IN: q|k [M][2C]; MID: q [M][C]; OUT (BlobNet): zero-conv residual [M][C]
    h16* out2;               // IN: V^T [B][C][ldvt]
    int ldvt;
    unsigned long long* gn_tot;   // OUT: statistics totals of the block output [B][C][BC_GN_TOT_WORDS] (added to with integer atomics), or null
    float ln_eps;
    float alpha;             // OUT (BlobNet): zero-conv scale
    const float* alpha_dev;
    const int* alpha_idx;
    int alpha_bstride;
    float* part;             // OUT_FF (written) / OUT_TAIL (read): fp32 partial sums of the feed-forward [nsplit][M][C]
    int nsplit;              // OUT_FF: workgroups per 64-row block (each takes NCH / nsplit hidden chunks); OUT_TAIL: slabs to add
    unsigned long long* stamps;   // BC_RC_STAMPS diagnostics: [workgroup][16] s_memtime stamps (null in production)
};

// Cross-attention inside the MID launch (BC_CHAIN_MIDX): 8 heads, a wave's 80 channels are 2 heads of 40 (C = 320) or one of 80 (C = 640)
template <int C>
struct XCfg {
    static constexpr int D = C / 8;                    // head dimension
    static constexpr int HPW = 80 / D;                 // heads per wave
    static constexpr int QKS = D == 40 ? 2 : 3;        // 32-channel k-steps of the X image that cover one head's channels
    static constexpr int NKT = 5;                      // key tiles of 16: up to 80 context tokens
    static constexpr int VT = D == 40 ? 3 : 5;         // value (output channel) tiles a head touches
    static constexpr int FR_HEAD = NKT * QKS + VT * 3; // fragments per head: K then V^T
    static constexpr int FR_WAVE = HPW * FR_HEAD;
    static constexpr int P_BYTES = 2 * 4096 + 2048;    // one wave's probability image
};

typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int V> struct IC { static constexpr int value = V; };

__device__ __forceinline__ h16x8 as_h8(uint4 v) { return __builtin_bit_cast(h16x8, v); }

__device__ __forceinline__ void lds_barrier() {
    // LDS visibility only: the weight ring's global loads stay in flight across it (a __syncthreads() would drain them)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ---- weight ring: R fragments ahead of the consumer, static indices only ------------------------------------------------------
template <int R>
struct WRing {
    uint4 f[R];
    const uint4* p;          // this lane's next fragment to LOAD
};

template <int R>
__device__ __forceinline__ void ring_fill(WRing<R>& r) {
#pragma unroll
    for (int i = 0; i < R; ++i) r.f[i] = r.p[i * 64];
    r.p += R * 64;
}

// One GEMM segment: acc[t][mt] += W-tile t (16 channels) x rows-tile mt (16 rows) over KS k-steps of 32.
// POS = ring index of the segment's first fragment (compile time).
// SWAP: D^T[channel][row] (lane: 4 consecutive channels of one row); !SWAP: D[row][channel] (lane: 4 consecutive rows of one channel).
// Issue order per k-step, pinned with sched_barrier (left alone, hipcc sinks the ring refills by two k-steps - the prefetch distance
// collapses to vmcnt(0..1) - and reads the operand fragments right in front of the MFMAs that need them): refill the slots the
// PREVIOUS step consumed, read the NEXT step's operand fragments, then this step's MFMAs run under both.
template <int NT, int KS, int POS, bool SWAP, int RC_R>
__device__ __forceinline__ void gemm_seg(f32x4v (&acc)[NT][4], WRing<RC_R>& r, const char* xb) {
    h16x8 xq[2][4];                                  // operand fragments of the current / the next k-step (static ping-pong)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) xq[0][mt] = *reinterpret_cast<const h16x8*>(xb + mt * 1024);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if (s > 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t) r.f[(POS + (s - 1) * NT + t) % RC_R] = r.p[t * 64];
            r.p += NT * 64;
        }
        if (s + 1 < KS) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) xq[(s + 1) & 1][mt] = *reinterpret_cast<const h16x8*>(xb + (s + 1) * 4096 + mt * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const h16x8 w = as_h8(r.f[(POS + s * NT + t) % RC_R]);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                acc[t][mt] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_f16(w, xq[s & 1][mt], acc[t][mt], 0, 0, 0)
                                  : __builtin_amdgcn_mfma_f32_16x16x32_f16(xq[s & 1][mt], w, acc[t][mt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) r.f[(POS + (KS - 1) * NT + t) % RC_R] = r.p[t * 64];
    r.p += NT * 64;
    __builtin_amdgcn_sched_barrier(0);
}

// The same segment for the LAST GEMM of a weight stream that is followed, in consumption order, by fragments that live elsewhere (MIDX:
// the K / V^T fragments of the workgroup's image at `p2`): its refill number JUMP (of NT * KS) and all later ones come from `p2`, and
// the ring runs on along that stream.
template <int NT, int KS, int POS, int RC_R, int JUMP>
__device__ __forceinline__ void gemm_seg_jump(f32x4v (&acc)[NT][4], WRing<RC_R>& r, const uint4* p2, const char* xb) {
    h16x8 xq[2][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) xq[0][mt] = *reinterpret_cast<const h16x8*>(xb + mt * 1024);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if (s > 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
                r.f[(POS + (s - 1) * NT + t) % RC_R] = (s - 1) * NT + t >= JUMP ? p2[((s - 1) * NT + t - JUMP) * 64] : r.p[t * 64];
            r.p += NT * 64;
        }
        if (s + 1 < KS) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) xq[(s + 1) & 1][mt] = *reinterpret_cast<const h16x8*>(xb + (s + 1) * 4096 + mt * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const h16x8 w = as_h8(r.f[(POS + s * NT + t) % RC_R]);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, xq[s & 1][mt], acc[t][mt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) r.f[(POS + (KS - 1) * NT + t) % RC_R] = (KS - 1) * NT + t >= JUMP ? p2[((KS - 1) * NT + t - JUMP) * 64] : r.p[t * 64];
    r.p = p2 + (NT * KS - JUMP) * 64;
    __builtin_amdgcn_sched_barrier(0);
}

// O^T += V^T . P^T for one head (MIDX): NTV value tiles starting at accumulator tile TLO, three k-steps over the (<= 80, padded to 96)
// keys.  V^T fragments from the ring; P (fp16 probabilities of this wave's 64 rows) from the wave's private image `Pw`: k-steps 0 and 1
// in the X operand layout, k-step 2 (keys 64..79 only) as [64 rows][32 bytes] - its upper half is zero by construction.
template <int NTV, int TLO, int POS, int RC_R>
__device__ __forceinline__ void pv_seg(f32x4v (&o)[5][4], WRing<RC_R>& r, const char* Pw, int xfo, int m, int q) {
    h16x8 xq[2][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) xq[0][mt] = *reinterpret_cast<const h16x8*>(Pw + mt * 1024 + xfo);
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        if (s > 0) {
#pragma unroll
            for (int t = 0; t < NTV; ++t) r.f[(POS + (s - 1) * NTV + t) % RC_R] = r.p[t * 64];
            r.p += NTV * 64;
        }
        if (s == 0) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) xq[1][mt] = *reinterpret_cast<const h16x8*>(Pw + 4096 + mt * 1024 + xfo);
        }
        if (s == 1) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const h16x8 v = *reinterpret_cast<const h16x8*>(Pw + 8192 + (16 * mt + m) * 32 + (q & 1) * 16);
                xq[0][mt] = q < 2 ? v : (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NTV; ++t) {
            const h16x8 w = as_h8(r.f[(POS + s * NTV + t) % RC_R]);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) o[TLO + t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, xq[s & 1][mt], o[TLO + t][mt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < NTV; ++t) r.f[(POS + 2 * NTV + t) % RC_R] = r.p[t * 64];
    r.p += NTV * 64;
    __builtin_amdgcn_sched_barrier(0);
}

template <int NT>
__device__ __forceinline__ void zero_acc(f32x4v (&acc)[NT][4]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[t][mt] = (f32x4v){0.f, 0.f, 0.f, 0.f};
}

// ---- LDS images -----------------------------------------------------------------------------------------------------------------
// X / P operand image: [k-step][row][32 k]: byte offset of the 16-byte chunk kc (0..3) of `row` in k-step s
__device__ __forceinline__ int x_off(int s, int row, int kc) { return s * 4096 + row * 64 + ((kc ^ ((0 - ((row & 15) >> 2)) & 3)) << 4); }
// S staging image: [row][C] fp16, 8-byte units XOR-swizzled by ((row >> 1) & 7)
template <int RC_C>
__device__ __forceinline__ int s_off8(int row, int unit) { return row * (RC_C * 2) + ((unit ^ ((row >> 1) & 7)) << 3); }

__device__ __forceinline__ uint4 swap_halves(uint4 d) { return make_uint4(d.z, d.w, d.x, d.y); }

// ---- coalesced 16-byte copies between global rows and the LDS images ------------------------------------------------------------------
// Thread (v8 = tid & 7, r = tid >> 3) moves the chunks v = v8 + 8 k (k < C / 64) of the rows r + ROWS_PASS j: the swizzles depend on row
// bits 1..3 only, so every address is a per-thread base plus a compile-time offset (an idx / 40 mapping made the compiler keep twenty
// 64-bit addresses alive across the whole kernel - spills).
struct CopyMap {
    int row, v8, sw;         // sw = (row >> 1) & 7
    int s_base, x_base;
};

template <int RC_C>
__device__ __forceinline__ CopyMap copy_map(int tid) {
    CopyMap c;
    c.row = tid >> 3;
    c.v8 = tid & 7;
    c.sw = (c.row >> 1) & 7;
    c.s_base = c.row * (RC_C * 2) + ((c.v8 ^ (c.sw >> 1)) << 4);
    c.x_base = x_off(c.v8 >> 2, c.row, c.v8 & 3);
    return c;
}

// global rows [64][C] (ld = C) -> S; rows whose pixel x < xmin are staged as zeros when `masked`
template <int RC_C>
__device__ __forceinline__ void rows_to_S(const h16* __restrict__ src, char* S, const CopyMap& c, bool masked = false, int pix0 = 0,
                                          int out_w = 1, int xmin = 0) {
    const h16* g = src + (size_t)c.row * RC_C + c.v8 * 8;
#pragma unroll
    for (int j = 0; j < RCfg<RC_C>::NPASS; ++j) {
        const bool zero = masked && ((pix0 + c.row + RCfg<RC_C>::ROWS_PASS * j) % out_w) < xmin;
#pragma unroll
        for (int k = 0; k < RCfg<RC_C>::NKC; ++k) {
            uint4 d = *reinterpret_cast<const uint4*>(g + j * RCfg<RC_C>::ROWS_PASS * RC_C + k * 64);
            if (zero) d = make_uint4(0u, 0u, 0u, 0u);
            if (c.sw & 1) d = swap_halves(d);
            *reinterpret_cast<uint4*>(S + c.s_base + j * RCfg<RC_C>::ROWS_PASS * RC_C * 2 + k * 128) = d;
        }
    }
}

// S -> global rows (row stride ld elements)
template <int RC_C>
__device__ __forceinline__ void S_to_rows(h16* __restrict__ dst, int ld, const char* S, const CopyMap& c) {
    h16* g = dst + (size_t)c.row * ld + c.v8 * 8;
#pragma unroll
    for (int j = 0; j < RCfg<RC_C>::NPASS; ++j)
#pragma unroll
        for (int k = 0; k < RCfg<RC_C>::NKC; ++k) {
            uint4 d = *reinterpret_cast<const uint4*>(S + c.s_base + j * RCfg<RC_C>::ROWS_PASS * RC_C * 2 + k * 128);
            if (c.sw & 1) d = swap_halves(d);
            *reinterpret_cast<uint4*>(g + (size_t)j * RCfg<RC_C>::ROWS_PASS * ld + k * 64) = d;
        }
}

// global rows -> X operand image, optionally through the per-(image, channel) GroupNorm affine y = a x + b (table [C][2] in LDS at `abl`)
template <int RC_C>
__device__ __forceinline__ void rows_to_X(const h16* __restrict__ src, char* X, const CopyMap& c, const char* abl, bool ab) {
    const h16* g = src + (size_t)c.row * RC_C + c.v8 * 8;
#pragma unroll
    for (int k = 0; k < RCfg<RC_C>::NKC; ++k) {
        float4 a4[4];
        if (ab) {
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) a4[j4] = reinterpret_cast<const float4*>(abl + (c.v8 + 8 * k) * 64)[j4];   // (a, b) of 8 channels
        }
#pragma unroll
        for (int j = 0; j < RCfg<RC_C>::NPASS; ++j) {
            uint4 d = *reinterpret_cast<const uint4*>(g + j * RCfg<RC_C>::ROWS_PASS * RC_C + k * 64);
            if (ab) {
                h16* e = reinterpret_cast<h16*>(&d);
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4) {
                    e[2 * j4] = (h16)fmaf((float)e[2 * j4], a4[j4].x, a4[j4].y);
                    e[2 * j4 + 1] = (h16)fmaf((float)e[2 * j4 + 1], a4[j4].z, a4[j4].w);
                }
            }
            *reinterpret_cast<uint4*>(X + c.x_base + k * 2 * 4096 + j * RCfg<RC_C>::ROWS_PASS * 64) = d;
        }
    }
}

// S -> X (the block output as the zero-conv's operand)
template <int RC_C>
__device__ __forceinline__ void S_to_X(const char* S, char* X, const CopyMap& c) {
#pragma unroll
    for (int j = 0; j < RCfg<RC_C>::NPASS; ++j)
#pragma unroll
        for (int k = 0; k < RCfg<RC_C>::NKC; ++k) {
            uint4 d = *reinterpret_cast<const uint4*>(S + c.s_base + j * RCfg<RC_C>::ROWS_PASS * RC_C * 2 + k * 128);
            if (c.sw & 1) d = swap_halves(d);
            *reinterpret_cast<uint4*>(X + c.x_base + k * 2 * 4096 + j * RCfg<RC_C>::ROWS_PASS * 64) = d;
        }
}

__device__ __forceinline__ h16x4 pack4(const float (&v)[4]) { return (h16x4){(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]}; }

// ---- epilogues on the swapped accumulator layout: acc[t][mt][r] = (channel 80w + 16t + 4q + r, row 16mt + m) ----------------------
// acc += bias (+ the fp16 rows staged in S when RES)
template <int RC_C, bool RES>
__device__ __forceinline__ void epi_bias_res(f32x4v (&acc)[RC_NT][4], const float* __restrict__ bias, const char* S, int wave, int m, int q) {
#pragma unroll
    for (int t = 0; t < RC_NT; ++t) {
        const int c0 = 80 * wave + 16 * t + 4 * q;
        const float4 b = bias ? *reinterpret_cast<const float4*>(bias + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            acc[t][mt][0] += b.x; acc[t][mt][1] += b.y; acc[t][mt][2] += b.z; acc[t][mt][3] += b.w;
            if (RES) {
                const h16x4 rr = *reinterpret_cast<const h16x4*>(S + s_off8<RC_C>(16 * mt + m, c0 >> 2));
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[t][mt][r] += (float)rr[r];
            }
        }
    }
}

// acc (scaled) -> fp16 -> S
template <int RC_C>
__device__ __forceinline__ void acc_to_S(const f32x4v (&acc)[RC_NT][4], char* S, int wave, int m, int q, float scale = 1.0f) {
#pragma unroll
    for (int t = 0; t < RC_NT; ++t) {
        const int c0 = 80 * wave + 16 * t + 4 * q;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const float v[4] = {acc[t][mt][0] * scale, acc[t][mt][1] * scale, acc[t][mt][2] * scale, acc[t][mt][3] * scale};
            *reinterpret_cast<h16x4*>(S + s_off8<RC_C>(16 * mt + m, c0 >> 2)) = pack4(v);
        }
    }
}

// acc -> fp16 -> X operand image
__device__ __forceinline__ void acc_to_X(const f32x4v (&acc)[RC_NT][4], char* X, int wave, int m, int q) {
#pragma unroll
    for (int t = 0; t < RC_NT; ++t) {
        const int c0 = 80 * wave + 16 * t + 4 * q;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const float v[4] = {acc[t][mt][0], acc[t][mt][1], acc[t][mt][2], acc[t][mt][3]};
            *reinterpret_cast<h16x4*>(X + x_off(c0 >> 5, 16 * mt + m, (c0 & 31) >> 3) + ((c0 >> 2) & 1) * 8) = pack4(v);
        }
    }
}

// LayerNorm over the 320 channels of every row (two passes, fp32), written as the next GEMM's operand image.  `gb` = gamma | beta.
// Statistics cross the four waves through the head of X, which is free between the barriers below.
template <int RC_C>
__device__ __forceinline__ void layernorm_to_X(const f32x4v (&acc)[RC_NT][4], const float* __restrict__ gb, float eps, char* X, int wave,
                                               int m, int q) {
    constexpr int NW = RCfg<RC_C>::NW;
    float* st = reinterpret_cast<float*>(X);                       // [2][64 rows][NW waves]
    float mean[4], rstd[4];
    auto row_total = [&](const float* p) {                          // sum of the NW per-wave partials of a row
        float4 v = *reinterpret_cast<const float4*>(p);
        float t = (v.x + v.y) + (v.z + v.w);
        if (NW == 8) {
            v = *reinterpret_cast<const float4*>(p + 4);
            t += (v.x + v.y) + (v.z + v.w);
        }
        return t;
    };
    lds_barrier();                                                  // every wave has left the k-loop that read X
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < RC_NT; ++t) s += (acc[t][mt][0] + acc[t][mt][1]) + (acc[t][mt][2] + acc[t][mt][3]);
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (q == 0) st[(16 * mt + m) * NW + wave] = s;
    }
    lds_barrier();
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        mean[mt] = row_total(st + (16 * mt + m) * NW) * (1.0f / RC_C);
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < RC_NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = acc[t][mt][r] - mean[mt];
                s = fmaf(d, d, s);
            }
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (q == 0) st[64 * NW + (16 * mt + m) * NW + wave] = s;
    }
    lds_barrier();
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) rstd[mt] = __builtin_amdgcn_rsqf(row_total(st + 64 * NW + (16 * mt + m) * NW) * (1.0f / RC_C) + eps);
    lds_barrier();                                                  // statistics consumed: X may be overwritten
#pragma unroll
    for (int t = 0; t < RC_NT; ++t) {
        const int c0 = 80 * wave + 16 * t + 4 * q;
        const float4 g = *reinterpret_cast<const float4*>(gb + c0);
        const float4 b = *reinterpret_cast<const float4*>(gb + RC_C + c0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const float v[4] = {fmaf((acc[t][mt][0] - mean[mt]) * rstd[mt], g.x, b.x), fmaf((acc[t][mt][1] - mean[mt]) * rstd[mt], g.y, b.y),
                                fmaf((acc[t][mt][2] - mean[mt]) * rstd[mt], g.z, b.z), fmaf((acc[t][mt][3] - mean[mt]) * rstd[mt], g.w, b.w)};
            *reinterpret_cast<h16x4*>(X + x_off(c0 >> 5, 16 * mt + m, (c0 & 31) >> 3) + ((c0 >> 2) & 1) * 8) = pack4(v);
        }
    }
}

// per-channel (sum, sum of squares) of the fp16 rows in S over the 64 rows, added to the consumer's GroupNorm statistics totals
// (round 6: one add per totals block - bc_common.h bc_gn_cg - through `scratch`, RC_C * 8 bytes of LDS nobody reads meanwhile; two barriers)
template <int RC_C>
__device__ __forceinline__ void gn_partials_from_S(const char* S, char* scratch, unsigned long long* __restrict__ dst, int tid, int spread) {
    float* const sc = reinterpret_cast<float*>(scratch);
    if (tid < RC_C / 2) {
        float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
        const int unit = tid >> 1, half = tid & 1;
#pragma unroll 8
        for (int row = 0; row < RC_BM; ++row) {
            const h16x2 v = *reinterpret_cast<const h16x2*>(S + s_off8<RC_C>(row, unit) + half * 4);
            const float a = (float)v[0], b = (float)v[1];
            s0 += a; q0 = fmaf(a, a, q0);
            s1 += b; q1 = fmaf(b, b, q1);
        }
        *reinterpret_cast<float4*>(sc + 4 * tid) = make_float4(s0, q0, s1, q1);
    }
    lds_barrier();
    for (int c = tid; c < RC_C; c += (int)blockDim.x)
        bc_gn_tot_add_slot(dst, c, 0, RC_C, bc_gn_cg(RC_C), spread, [&](int k) { return make_float2(sc[2 * k], sc[2 * k + 1]); });
    lds_barrier();
}

template <int RC_C, int KIND, bool BLOB>
__global__ __launch_bounds__(RCfg<RC_C>::NTH, 2) void rowchain_kernel(const RowChainArgs a) {
    using CF = RCfg<RC_C>;
    constexpr int RC_KS = CF::KS, RC_NCH = CF::NCH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* X = smem;
    char* S = smem + CF::X_BYTES;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, q = lane >> 4;
    // OUT_FF: nsplit workgroups share a row block, each with its own slice of the hidden chunks (and its own weight stream)
    // OUT_FFP (round 5): OUT_FF that goes on THROUGH proj_out [and the zero-conv] on its own partial sum - both are linear, so the sum over
    // the slices moves behind them and OUT_TAIL's GEMMs run on nsplit times as many CUs; what is left is an elementwise sum of nsplit
    // fp16 partial outputs (rowchain_sum_kernel).  Its slices are SLICE-MAJOR in the remapped workgroup id: an XCD holds the workgroups
    // of one slice (nsplit = 4: two XCDs per slice), so its L2 fetches one quarter of the feed-forward weights instead of all of them.
    constexpr bool FFP = KIND == BC_CHAIN_OUT_FFP;
    constexpr bool FF = KIND == BC_CHAIN_OUT_FF || FFP, TAIL = KIND == BC_CHAIN_OUT_TAIL;
    const int wg = bc_xcd_remap(blockIdx.x, gridDim.x);
    const int nrb = a.M / RC_BM;
    const int z = FFP ? wg / nrb : (FF ? wg % a.nsplit : 0);
    const int m0 = (FFP ? wg % nrb : (FF ? wg / a.nsplit : wg)) * RC_BM;
    const int b = m0 / a.rows_per_batch;
    const int pix0 = m0 - b * a.rows_per_batch;
    const int xfo = m * 64 + ((q ^ ((0 - (m >> 2)) & 3)) << 4);        // this lane's fragment offset inside a (k-step, row-tile) KiB
    const float* vec = a.vec;
    const CopyMap cm = copy_map<RC_C>(tid);
    unsigned long long* const stamps = a.stamps;
    unsigned long long acc_t[3] = {0, 0, 0};         // (OUT, diagnostics) ticks inside ff1 GEMMs / GEGLU epilogues / ff2 GEMMs
    auto stamp = [&](int i) {
        if (stamps && tid == 0) stamps[(size_t)blockIdx.x * 16 + i] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);

    if (KIND == BC_CHAIN_IN && a.gn_in) {
        // (before the weight ring is filled: its registers stay out of this prologue)
        // GroupNorm finalize of this image from the statistics totals (bc_common.h) into S, which is free here: no bc_gn_finalize
        // launch in front of the block.  gamma / beta first (they are needed last), one thread per channel, eight lanes per group.
        constexpr int NTH = CF::NTH, PER = (RC_C + NTH - 1) / NTH;
        float* abl = reinterpret_cast<float*>(S);                       // [C][2]
        double* scr = reinterpret_cast<double*>(S + RC_C * 8);          // [C][2]
        float* stat = reinterpret_cast<float*>(S + RC_C * 24);          // [groups][2]
        float gp[PER], bp_[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int c = tid + NTH * j;
            gp[j] = c < RC_C ? a.gn_gamma[c] : 0.f;
            bp_[j] = c < RC_C ? a.gn_beta[c] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int c = tid + NTH * j;
            if (c < RC_C) {
                double s_, q_;
                bc_gn_tot_read(a.gn_in + ((size_t)b * RC_C + c) * BC_GN_TOT_WORDS, s_, q_);
                scr[c * 2] = s_;
                scr[c * 2 + 1] = q_;
            }
        }
        __syncthreads();
        const int cpg = RC_C / a.gn_groups, sub = tid & 7;
        for (int gi = tid >> 3; gi < a.gn_groups; gi += NTH / 8) {
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
            const double n = (double)a.rows_per_batch * cpg;
            const double mean = s_ / n;
            double var = q_ / n - mean * mean;
            if (var < 0.0) var = 0.0;
            if (sub == 0) {
                stat[gi * 2] = (float)mean;
                stat[gi * 2 + 1] = (float)(1.0 / sqrt(var + (double)a.gn_eps));
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int c = tid + NTH * j;
            if (c < RC_C) {
                const int gi = c / cpg;
                const float av = stat[gi * 2 + 1] * gp[j];
                abl[c * 2] = av;
                abl[c * 2 + 1] = bp_[j] - stat[gi * 2] * av;
            }
        }
        __syncthreads();
    } else if (KIND == BC_CHAIN_IN && a.affine) {
        // the table bc_gn_finalize wrote, staged in S like the in-kernel one (one way for rows_to_X to read it)
        const float4* src4 = reinterpret_cast<const float4*>(a.affine + (size_t)b * RC_C * 2);
        for (int i = tid; i < RC_C / 2; i += CF::NTH) reinterpret_cast<float4*>(S)[i] = src4[i];
        __syncthreads();
    }
    constexpr bool FFLOOP = KIND == BC_CHAIN_OUT || FF;      // kinds that run the feed-forward loop
    constexpr bool MIDX = KIND == BC_CHAIN_MIDX;
    constexpr int RC_R = RC_C == 320 ? (FFLOOP ? 15 : (KIND == BC_CHAIN_IN ? 13 : (MIDX ? 12 : 16)))
                                     : (FFLOOP ? 10 : (KIND == BC_CHAIN_IN ? 12 : (MIDX ? 8 : 12)));   // (8 waves: two per SIMD hide more latency)
    static_assert(RC_R <= RC_RPAD, "stream padding");
    WRing<RC_R> ring;
    ring.p = a.wstream + ((size_t)z * CF::NW + wave) * a.wave_frags * 64 + lane;
    ring_fill(ring);

    f32x4v acc[RC_NT][4];
    zero_acc(acc);

    if (KIND == BC_CHAIN_IN) {
        rows_to_X<RC_C>(a.x + (size_t)m0 * RC_C, X, cm, S, a.gn_in != nullptr || a.affine != nullptr);
        lds_barrier();
        stamp(1);
        // proj_in -> h0
        gemm_seg<RC_NT, RC_KS, 0, true>(acc, ring, X + xfo);
        stamp(2);
        epi_bias_res<RC_C, false>(acc, vec, S, wave, m, q);
        layernorm_to_X<RC_C>(acc, vec + RC_C, a.ln_eps, X, wave, m, q);
        stamp(3);
        acc_to_S<RC_C>(acc, S, wave, m, q);
        lds_barrier();                                                  // X = LN1(h0), S = h0
        S_to_rows<RC_C>(a.out0 + (size_t)m0 * RC_C, RC_C, S, cm);
        // to_q | to_k: two passes of 320 columns
        zero_acc(acc);
        gemm_seg<RC_NT, RC_KS, (RC_NT * RC_KS) % RC_R, true>(acc, ring, X + xfo);
        lds_barrier();                                                  // S has been copied out by every thread
        acc_to_S<RC_C>(acc, S, wave, m, q);
        lds_barrier();
        S_to_rows<RC_C>(a.out1 + (size_t)m0 * 2 * RC_C, 2 * RC_C, S, cm);
        zero_acc(acc);
        gemm_seg<RC_NT, RC_KS, (2 * RC_NT * RC_KS) % RC_R, true>(acc, ring, X + xfo);
        lds_barrier();
        acc_to_S<RC_C>(acc, S, wave, m, q);
        lds_barrier();
        S_to_rows<RC_C>(a.out1 + (size_t)m0 * 2 * RC_C + RC_C, 2 * RC_C, S, cm);
        // to_v, written transposed: D[row][channel] (lane: 4 consecutive rows of one channel) -> S^T [channel][64 rows]
        zero_acc(acc);
        gemm_seg<RC_NT, RC_KS, (3 * RC_NT * RC_KS) % RC_R, false>(acc, ring, X + xfo);
        lds_barrier();
#pragma unroll
        for (int t = 0; t < RC_NT; ++t) {
            const int ch = 80 * wave + 16 * t + m;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const float v[4] = {acc[t][mt][0], acc[t][mt][1], acc[t][mt][2], acc[t][mt][3]};
                *reinterpret_cast<h16x4*>(S + ch * 128 + (((4 * mt + q) ^ (ch & 15)) << 3)) = pack4(v);
            }
        }
        lds_barrier();
        {
            const int ch0 = tid >> 3, part = tid & 7, sw = ch0 & 15;
            h16* vt = a.out2 + ((size_t)b * RC_C + ch0) * a.ldvt + pix0 + part * 8;
            const char* sp = S + ch0 * 128 + ((part ^ (sw >> 1)) << 4);
#pragma unroll
            for (int i = 0; i < RC_C / CF::ROWS_PASS; ++i) {            // channels ch0 + ROWS_PASS i: the swizzle (ch & 15) keeps its low 4 bits
                uint4 d = *reinterpret_cast<const uint4*>(sp + i * CF::ROWS_PASS * 128);
                if (sw & 1) d = swap_halves(d);
                *reinterpret_cast<uint4*>(vt + (size_t)i * CF::ROWS_PASS * a.ldvt) = d;
            }
        }
        stamp(4);
        return;
    }

    // MID / OUT / OUT_FF: X = attention output rows, S = the residual stream
    if (!TAIL) {
        rows_to_X<RC_C>(a.x + (size_t)m0 * RC_C, X, cm, S, false);
        rows_to_S<RC_C>(a.res + (size_t)m0 * RC_C, S, cm);
        lds_barrier();
        stamp(1);
        gemm_seg<RC_NT, RC_KS, 0, true>(acc, ring, X + xfo);           // attn.to_out
        stamp(2);
        epi_bias_res<RC_C, true>(acc, vec, S, wave, m, q);                    // + bias + residual (own columns only: no barrier needed)
    }

    if (KIND == BC_CHAIN_MID) {
        layernorm_to_X<RC_C>(acc, vec + RC_C, a.ln_eps, X, wave, m, q);
        acc_to_S<RC_C>(acc, S, wave, m, q);
        lds_barrier();
        S_to_rows<RC_C>(a.out0 + (size_t)m0 * RC_C, RC_C, S, cm);           // h1
        zero_acc(acc);
        gemm_seg<RC_NT, RC_KS, (RC_NT * RC_KS) % RC_R, true>(acc, ring, X + xfo);   // attn2.to_q
        lds_barrier();
        acc_to_S<RC_C>(acc, S, wave, m, q);
        lds_barrier();
        S_to_rows<RC_C>(a.out1 + (size_t)m0 * RC_C, RC_C, S, cm);
        stamp(4);
        return;
    }
    if constexpr (MIDX) {
        // MID with the cross-attention of the block (attention_processor.py:2191-2224 on the <= 80 context tokens) behind to_q: the
        // query rows never leave the workgroup.  K and V^T of the workgroup's image arrive as fragment streams through the weight ring
        // (bc_rowchain_pack_kv: per wave, the heads its 80 channels hold), so S^T = K Q^T and O^T = V^T P^T are two more segments of the
        // chain; the softmax runs on the swapped accumulator layout (a lane holds 4 keys of one row: two cross-lane steps per row).
        using XC = XCfg<RC_C>;
        constexpr int G = RC_NT * RC_KS;
        layernorm_to_X<RC_C>(acc, vec + RC_C, a.ln_eps, X, wave, m, q);
        acc_to_S<RC_C>(acc, S, wave, m, q);
        lds_barrier();
        S_to_rows<RC_C>(a.out0 + (size_t)m0 * RC_C, RC_C, S, cm);           // h1
        zero_acc(acc);
        // attn2.to_q; the ring runs on into the K / V^T stream
        gemm_seg_jump<RC_NT, RC_KS, G % RC_R, RC_R, G - RC_R>(
            acc, ring, reinterpret_cast<const uint4*>(a.res2) + ((size_t)b * CF::NW + wave) * (XC::FR_WAVE + RC_RPAD) * 64 + lane, X + xfo);
        lds_barrier();                                                  // X (LayerNorm2 image) and S (h1) are free
        acc_to_X(acc, X, wave, m, q);                                   // Q of this wave's heads, as operand image (read back by this wave only)
        f32x4v o[RC_NT][4];
        zero_acc(o);
        char* Pw = S + wave * XC::P_BYTES;
        const float csc = a.alpha * 1.4426950408889634f;
        const int T = a.r2_xmin;
        auto head = [&](auto hc) {
            constexpr int hh = decltype(hc)::value;
            constexpr int POS_K = (2 * G + hh * XC::FR_HEAD) % RC_R, POS_V = (POS_K + XC::NKT * XC::QKS) % RC_R;
            constexpr int TLO = XC::D == 40 && hh == 1 ? 2 : 0;
            const int s0 = (80 * wave + XC::D * hh) >> 5;
            zero_acc(acc);
            gemm_seg<XC::NKT, XC::QKS, POS_K, true, RC_R>(acc, ring, X + s0 * 4096 + xfo);   // scores^T[key 16 t + 4 q + r][row 16 mt + m]
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                float mx = -3.0e38f;
#pragma unroll
                for (int t = 0; t < XC::NKT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (16 * t + 4 * q + r < T) mx = fmaxf(mx, acc[t][mt][r]);
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                float sum = 0.f;
#pragma unroll
                for (int t = 0; t < XC::NKT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = 16 * t + 4 * q + r < T ? __builtin_amdgcn_exp2f((acc[t][mt][r] - mx) * csc) : 0.f;
                        acc[t][mt][r] = e;
                        sum += e;
                    }
                sum += __shfl_xor(sum, 16);
                sum += __shfl_xor(sum, 32);
                const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
                for (int t = 0; t < XC::NKT; ++t) {
                    const float v[4] = {acc[t][mt][0] * inv, acc[t][mt][1] * inv, acc[t][mt][2] * inv, acc[t][mt][3] * inv};
                    char* dst = t < 4 ? Pw + x_off(t >> 1, 16 * mt + m, 2 * (t & 1) + (q >> 1)) + (q & 1) * 8 : Pw + 8192 + (16 * mt + m) * 32 + q * 8;
                    *reinterpret_cast<h16x4*>(dst) = pack4(v);
                }
            }
            pv_seg<XC::VT, TLO, POS_V, RC_R>(o, ring, Pw, xfo, m, q);
        };
        head(IC<0>{});
        if constexpr (XC::HPW == 2) head(IC<1>{});
        lds_barrier();                                                  // every wave has read its probability image: S may be overwritten
        // (lane-derived addresses are recomputed from a laundered thread id: kept alive across the attention they spill)
        int tid2 = threadIdx.x;
        asm volatile("" : "+v"(tid2));
        const int m2 = tid2 & 15, q2 = (tid2 & 63) >> 4;
        acc_to_S<RC_C>(o, S, wave, m2, q2);
        lds_barrier();
        S_to_rows<RC_C>(a.out1 + (size_t)m0 * RC_C, RC_C, S, copy_map<RC_C>(tid2));     // attention output rows
        stamp(4);
        return;
    }

    // ---- OUT: LayerNorm3 -> GEGLU feed-forward accumulated ON TOP of h2 (acc keeps the residual in fp32) ----
    // ---- OUT_FF: the same for the hidden chunks [z, z + 1) * NCH / nsplit; only slice 0 keeps h2 under its partial sum ----
    if (!TAIL) {
    layernorm_to_X<RC_C>(acc, vec + RC_C, a.ln_eps, X, wave, m, q);
    lds_barrier();
    stamp(3);
    if (FF && z != 0) zero_acc(acc);
    const int c_lo = FF ? z * (RC_NCH / a.nsplit) : 0, c_hi = FF ? c_lo + RC_NCH / a.nsplit : RC_NCH;
    {
        const float* b1 = vec + 3 * RC_C;                               // [chunk][wave][pass][value 16 | gate 16]
        constexpr int POS_FF = (RC_NT * RC_KS) % RC_R;
        constexpr int TP = CF::TP, HPW = CF::HW_;                        // (value, gate) tile passes per chunk, hidden units per wave
        static_assert(!FFLOOP || (TP * 2 * RC_KS + RC_NT * (RC_HC / 32)) % RC_R == 0,
                      "a feed-forward chunk must consume a whole number of rings");
        for (int c = c_lo; c < c_hi; ++c) {
            char* P = S + (c & 1) * 16384;
            const float* bb = b1 + (c * CF::NW + wave) * (2 * HPW) + 4 * q;
            // (the chunk's GEGLU biases are fetched here, a GEMM pass ahead of their use: behind the pass's sched_barriers they cost
            //  an exposed L2 round trip per pass)
            float4 bvg[2 * TP];
#pragma unroll
            for (int i = 0; i < 2 * TP; ++i) bvg[i] = *reinterpret_cast<const float4*>(bb + 16 * i);
            // ff.net.0.proj for this wave's hidden units of the chunk in (value tile, gate tile) pairs: 32 accumulator registers per
            // pass (the block's 80 stay resident underneath)
#pragma unroll
            for (int tp = 0; tp < TP; ++tp) {
                f32x4v a1[2][4];
                zero_acc(a1);
                const unsigned long long ta = stamps ? __builtin_amdgcn_s_memtime() : 0;
                if (tp == 0) gemm_seg<2, RC_KS, POS_FF, true>(a1, ring, X + xfo);
                else gemm_seg<2, RC_KS, (POS_FF + 2 * RC_KS) % RC_R, true>(a1, ring, X + xfo);
                const unsigned long long tb = stamps ? __builtin_amdgcn_s_memtime() : 0;
                acc_t[0] += tb - ta;
                const float4 bv = bvg[2 * tp], bg = bvg[2 * tp + 1];
                const int hc0 = HPW * wave + 16 * tp + 4 * q;           // hidden unit (inside the chunk) of this lane's first value
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    // (two elements per VALU issue: bc_gelu_f2 = bc_gelu_f bit for bit)
                    const f32x2 g01 = bc_gelu_f2((f32x2){a1[1][mt][0], a1[1][mt][1]} + (f32x2){bg.x, bg.y});
                    const f32x2 g23 = bc_gelu_f2((f32x2){a1[1][mt][2], a1[1][mt][3]} + (f32x2){bg.z, bg.w});
                    const f32x2 v01 = ((f32x2){a1[0][mt][0], a1[0][mt][1]} + (f32x2){bv.x, bv.y}) * g01;
                    const f32x2 v23 = ((f32x2){a1[0][mt][2], a1[0][mt][3]} + (f32x2){bv.z, bv.w}) * g23;
                    const float v[4] = {v01.x, v01.y, v23.x, v23.y};
                    *reinterpret_cast<h16x4*>(P + x_off(hc0 >> 5, 16 * mt + m, (hc0 & 31) >> 3) + ((hc0 >> 2) & 1) * 8) = pack4(v);
                }
                if (stamps) acc_t[1] += __builtin_amdgcn_s_memtime() - tb;
            }
            lds_barrier();                                              // the chunk's 128 hidden columns are complete
            const unsigned long long tc = stamps ? __builtin_amdgcn_s_memtime() : 0;
            gemm_seg<RC_NT, RC_HC / 32, (POS_FF + TP * 2 * RC_KS) % RC_R, true>(acc, ring, P + xfo);
            if (stamps) acc_t[2] += __builtin_amdgcn_s_memtime() - tc;
        }
    }
    }
    stamp(4);
    if (stamps && tid == 0) {
        stamps[(size_t)blockIdx.x * 16 + 8] = acc_t[0];
        stamps[(size_t)blockIdx.x * 16 + 9] = acc_t[1];
        stamps[(size_t)blockIdx.x * 16 + 10] = acc_t[2];
    }
    if (KIND == BC_CHAIN_OUT_FF) {                                       // this slice's partial sum (fp32), one 16-byte store per tile
        float* dst = a.part + ((size_t)z * a.M + m0) * RC_C + 80 * wave + 4 * q;
#pragma unroll
        for (int t = 0; t < RC_NT; ++t)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) *reinterpret_cast<f32x4v*>(dst + (size_t)(16 * mt + m) * RC_C + 16 * t) = acc[t][mt];
        stamp(7);
        return;
    }
    if (TAIL) {                                                          // h2 + the whole feed-forward = the slices' partial sums, in order
        const float* src = a.part + (size_t)m0 * RC_C + 80 * wave + 4 * q;
        for (int zz = 0; zz < a.nsplit; ++zz)
#pragma unroll
            for (int t = 0; t < RC_NT; ++t)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[t][mt] += *reinterpret_cast<const f32x4v*>(src + ((size_t)zz * a.M + 16 * mt + m) * RC_C + 16 * t);
    }
    // (lane-derived values are laundered here: without it hipcc keeps the LDS addresses of the epilogues before and after the
    //  feed-forward loop alive ACROSS it - common subexpressions - and spills two dozen registers around the loop)
    int m_ = m, q_ = q, xfo_ = xfo;
    CopyMap cm_ = cm;
    asm volatile("" : "+v"(m_), "+v"(q_), "+v"(xfo_), "+v"(cm_.row), "+v"(cm_.v8), "+v"(cm_.sw), "+v"(cm_.s_base), "+v"(cm_.x_base));
#define m m_
#define q q_
#define xfo xfo_
#define cm cm_
    // OUT_FFP: everything that is NOT linear in the partial sum (biases, the block input x, the BlobNet residual) belongs to slice 0 only
    const bool first = !FFP || z == 0;                                   // (workgroup-uniform)
    if (first) epi_bias_res<RC_C, false>(acc, vec + 3 * RC_C + 8 * RC_C, S, wave, m, q);  // + ff.net.2 bias -> h3
    lds_barrier();                                                      // every wave is done with the last hidden chunk (S) and with X
    acc_to_X(acc, X, wave, m, q);
    if (first) rows_to_S<RC_C>(a.res2 + (size_t)m0 * RC_C, S, cm);
    lds_barrier();
    zero_acc(acc);
    constexpr int POS_PO = TAIL ? 0 : (RC_NT * RC_KS) % RC_R;           // (the feed-forward consumed whole rings; OUT_TAIL's stream starts here)
    stamp(5);
    gemm_seg<RC_NT, RC_KS, POS_PO, true>(acc, ring, X + xfo);          // proj_out
    stamp(6);
    const float* bpo = vec + 3 * RC_C + 8 * RC_C + RC_C;
    if (first) epi_bias_res<RC_C, true>(acc, bpo, S, wave, m, q);             // + bias + x
    if (a.r2 && first) {                                                 // + BlobNet residual on the right-hand part of the canvas
        const bool any = ((pix0 % a.out_w) + RC_BM > a.r2_xmin) || (pix0 % a.out_w) + RC_BM > a.out_w;
        if (any) {                                                       // (workgroup-uniform)
            lds_barrier();
            rows_to_S<RC_C>(a.r2 + ((size_t)(b % a.r2_bmod) * a.rows_per_batch + pix0) * RC_C, S, cm, true, pix0, a.out_w, a.r2_xmin);
            lds_barrier();
            epi_bias_res<RC_C, true>(acc, nullptr, S, wave, m, q);
        }
    }
    lds_barrier();
    acc_to_S<RC_C>(acc, S, wave, m, q);
    lds_barrier();
    // OUT_FFP: this slice's fp16 partial OUTPUT [nsplit][M][C] (in `part`; behind it the zero-conv partials); the statistics of the
    // summed output are the sum kernel's
    h16* const p16 = reinterpret_cast<h16*>(a.part);
    S_to_rows<RC_C>(FFP ? p16 + ((size_t)z * a.M + m0) * RC_C : a.out0 + (size_t)m0 * RC_C, RC_C, S, cm);
    if (!FFP && a.gn_tot) gn_partials_from_S<RC_C>(S, X, a.gn_tot + (size_t)b * RC_C * BC_GN_TOT_WORDS, tid, m0 / RC_BM);    // (X is free here)
    stamp(7);
    if (BLOB) {
        // zero-conv of the block output (the BlobNet residual the UNet adds): r = (W out + b) * conditioning scale
        S_to_X<RC_C>(S, X, cm);
        lds_barrier();
        zero_acc(acc);
        gemm_seg<RC_NT, RC_KS, (POS_PO + RC_NT * RC_KS) % RC_R, true>(acc, ring, X + xfo);
        if (first) epi_bias_res<RC_C, false>(acc, bpo + RC_C, S, wave, m, q);
        float alpha = a.alpha;
        if (a.alpha_dev) alpha *= a.alpha_dev[(a.alpha_idx ? *a.alpha_idx : 0) * (a.alpha_bstride > 0 ? a.alpha_bstride : 1) + (a.alpha_bstride > 0 ? b : 0)];
        lds_barrier();                                                  // S (block output) copied out and transposed into X by everyone
        acc_to_S<RC_C>(acc, S, wave, m, q, alpha);
        lds_barrier();
        S_to_rows<RC_C>(FFP ? p16 + ((size_t)(a.nsplit + z) * a.M + m0) * RC_C : a.out1 + (size_t)m0 * RC_C, RC_C, S, cm);
    }
#undef m
#undef q
#undef xfo
#undef cm
}

// BC_RC_STAMPS=1 (diagnostics; synchronises the stream after every launch): where a workgroup's cycles go
struct RcStampReport {
    hipStream_t stream; size_t n; unsigned long long* buf; int kind;
    ~RcStampReport() {
        if (!buf) return;
        if (hipStreamSynchronize(stream) != hipSuccess) return;
        std::vector<unsigned long long> h(n * 16);
        if (hipMemcpy(h.data(), buf, n * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return;
        double d[8] = {0}, ff[3] = {0};
        unsigned long long t0 = ~0ull, t1 = 0;
        const int last = kind == BC_CHAIN_OUT ? 7 : 4;
        for (size_t i = 0; i < n; ++i) {
            for (int k = 0; k < last; ++k) d[k] += (double)(h[i * 16 + k + 1] - h[i * 16 + k]);
            for (int k = 0; k < 3; ++k) ff[k] += (double)h[i * 16 + 8 + k];
            t0 = std::min(t0, h[i * 16]);
            t1 = std::max(t1, h[i * 16 + last]);
        }
        if (kind == BC_CHAIN_OUT)
            fprintf(stderr, "[rowchain stamps out] wgs=%zu | avg ticks: prologue %.0f, to_out gemm %.0f, epilogue+LN %.0f, ff loop %.0f (ff1 gemms %.0f, "
                    "geglu %.0f, ff2 gemms %.0f), h3 + x staging %.0f, proj_out gemm %.0f, epilogue+stores %.0f | first entry -> last exit %llu ticks\n",
                    n, d[0] / n, d[1] / n, d[2] / n, d[3] / n, ff[0] / n, ff[1] / n, ff[2] / n, d[4] / n, d[5] / n, d[6] / n, t1 - t0);
        else
            fprintf(stderr, "[rowchain stamps %s] wgs=%zu | avg ticks: prologue %.0f, first gemm %.0f, epilogue+LN %.0f, rest %.0f | first entry -> last "
                    "exit %llu ticks\n", kind == BC_CHAIN_IN ? "in" : kind == BC_CHAIN_MIDX ? "midx" : "mid", n, d[0] / n, d[1] / n, d[2] / n, d[3] / n, t1 - t0);
    }
};

template <int RC_C, int KIND, bool BLOB>
int launch_chain(const RowChainArgs& a_in, hipStream_t stream) {
    using CF = RCfg<RC_C>;
    RowChainArgs a = a_in;
    static const bool want_stamps = getenv("BC_RC_STAMPS") != nullptr;
    static unsigned long long* stamp_buf = nullptr;
    const size_t nwg = (size_t)a.M / RC_BM;
    a.stamps = nullptr;
    if (want_stamps && nwg <= 8192) {
        if (!stamp_buf) BC_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&stamp_buf), 8192 * 16 * sizeof(unsigned long long)));
        BC_CHECK_HIP(hipMemsetAsync(stamp_buf, 0, nwg * 16 * sizeof(unsigned long long), stream));
        a.stamps = stamp_buf;
    }
    RcStampReport report{stream, nwg, a.stamps, KIND};
    static std::atomic<unsigned long long> lds_set{0};
    BC_CHECK_HIP(bc_set_max_lds(lds_set, reinterpret_cast<const void*>(&rowchain_kernel<RC_C, KIND, BLOB>), CF::LDS));
    hipLaunchKernelGGL((rowchain_kernel<RC_C, KIND, BLOB>), dim3(a.M / RC_BM * (KIND == BC_CHAIN_OUT_FF || KIND == BC_CHAIN_OUT_FFP ? a.nsplit : 1)), dim3(CF::NTH), CF::LDS, stream, a);
    BC_CHECK_LAUNCH();
    return 0;
}

#ifndef BC_ROWCHAIN_MIDX_TU
// ---- the block end's reduction (round 5): out = sum over the nsplit fp16 partial outputs of BC_CHAIN_OUT_FFP (fp32 sum in slice order:
// bit-reproducible), its GroupNorm statistics added to the consumer's totals; with out1 the zero-conv partials behind them likewise.
// One thread = 8 channels (16-byte accesses) of SUM_ROWS / RL rows; HBM-bound: (nsplit + 1) * M * C * 2 bytes.
constexpr int SUM_ROWS = 64;           // rows per workgroup = the producers' row block: the same number of statistics atomics as BC_CHAIN_OUT
template <int RC_C>
__global__ __launch_bounds__(640) void rowchain_sum_kernel(const h16* __restrict__ part, int nsplit, int M, int rows_per_batch, h16* __restrict__ out0,
                                                           unsigned long long* __restrict__ gn_tot, h16* __restrict__ out1) {
    constexpr int NC = RC_C / 8, RL = 640 / NC, RPT = SUM_ROWS / RL;       // column groups of 8 channels, row lanes (8 / 16), rows per thread (8 / 4)
    __shared__ float red[RL][NC][16];
    const int tid = threadIdx.x, cg = tid % NC, rl = tid / NC;
    const int m0 = blockIdx.x * SUM_ROWS;
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const size_t off = (size_t)(m0 + rl + RL * i) * RC_C + cg * 8;
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            if (o == 1 && !out1) break;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
            for (int z = 0; z < nsplit; ++z) {
                const h16x8 d = *reinterpret_cast<const h16x8*>(part + ((size_t)(o * nsplit + z) * M) * RC_C + off);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += (float)d[j];
            }
            h16x8 r;
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = (h16)v[j];
            *reinterpret_cast<h16x8*>((o ? out1 : out0) + off) = r;
            if (o == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float f = (float)r[j];                    // statistics of the fp16-ROUNDED output, as every other producer's
                    s[j] += f;
                    q[j] = fmaf(f, f, q[j]);
                }
            }
        }
    }
    if (gn_tot) {                                                   // (workgroup-uniform)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            red[rl][cg][j] = s[j];
            red[rl][cg][8 + j] = q[j];
        }
        __syncthreads();
        const int b = m0 / rows_per_batch;
        if (tid < RC_C)                                             // one add per totals block (bc_gn_cg: 10 channels), the row lanes in a fixed order
            bc_gn_tot_add_slot(gn_tot + (size_t)b * RC_C * BC_GN_TOT_WORDS, tid, 0, RC_C, bc_gn_cg(RC_C), (int)blockIdx.x, [&](int k) {
                float ss = 0.f, qq = 0.f;
#pragma unroll
                for (int r2 = 0; r2 < RL; ++r2) {
                    ss += red[r2][k >> 3][k & 7];
                    qq += red[r2][k >> 3][8 + (k & 7)];
                }
                return make_float2(ss, qq);
            });
    }
}

template <int RC_C>
long long stream_frags(int kind, int blobnet, int nsplit) {
    using CF = RCfg<RC_C>;
    const long long g = CF::G;
    const long long per_chunk = CF::TP * 2 * CF::KS + RC_NT * (RC_HC / 32);
    long long n = 0;
    if (kind == BC_CHAIN_IN) n = 4 * g;
    else if (kind == BC_CHAIN_MID || kind == BC_CHAIN_MIDX) n = 2 * g;      // (MIDX reads MID's stream)
    else if (kind == BC_CHAIN_OUT) n = 2 * g + CF::NCH * per_chunk + (blobnet ? g : 0);
    else if (kind == BC_CHAIN_OUT_FF && nsplit > 0 && CF::NCH % nsplit == 0) n = g + CF::NCH / nsplit * per_chunk;
    else if (kind == BC_CHAIN_OUT_FFP && nsplit > 0 && CF::NCH % nsplit == 0) n = 2 * g + CF::NCH / nsplit * per_chunk + (blobnet ? g : 0);
    else if (kind == BC_CHAIN_OUT_TAIL) n = g + (blobnet ? g : 0);
    else return -1;
    return n + RC_RPAD;
}

template <int RC_C>
int dispatch_chain(int kind, bool blob, const RowChainArgs& a, hipStream_t s) {
    switch (kind) {
        case BC_CHAIN_IN: return launch_chain<RC_C, BC_CHAIN_IN, false>(a, s);
        case BC_CHAIN_MID: return launch_chain<RC_C, BC_CHAIN_MID, false>(a, s);
        case BC_CHAIN_OUT_FF: return launch_chain<RC_C, BC_CHAIN_OUT_FF, false>(a, s);
        case BC_CHAIN_OUT_TAIL: return blob ? launch_chain<RC_C, BC_CHAIN_OUT_TAIL, true>(a, s) : launch_chain<RC_C, BC_CHAIN_OUT_TAIL, false>(a, s);
        case BC_CHAIN_OUT_FFP: return blob ? launch_chain<RC_C, BC_CHAIN_OUT_FFP, true>(a, s) : launch_chain<RC_C, BC_CHAIN_OUT_FFP, false>(a, s);
        default: return blob ? launch_chain<RC_C, BC_CHAIN_OUT, true>(a, s) : launch_chain<RC_C, BC_CHAIN_OUT, false>(a, s);
    }
}

}  // namespace

extern "C" int bc_rowchain_supported(int channels, int M, int rows_per_batch) {
    return (channels == 320 || channels == 640) && M > 0 && rows_per_batch > 0 && M % rows_per_batch == 0 && rows_per_batch % RC_BM == 0;
}

extern "C" long long bc_rowchain_stream_frags(int channels, int kind, int blobnet, int nsplit) {
    // fragments (1 KiB per wave-instruction) of ONE wave's weight stream, incl. the padding at the end; -1: unsupported
    if (channels == 320) return stream_frags<320>(kind, blobnet, nsplit);
    if (channels == 640) return stream_frags<640>(kind, blobnet, nsplit);
    return -1;
}

extern "C" int bc_rowchain(int kind, int channels, int M, int rows_per_batch, const bc_half* x, const float* affine,
                           const unsigned long long* gn_in, const float* gn_gamma, const float* gn_beta, int gn_groups, float gn_eps, const bc_half* res,
                           const bc_half* res2, const bc_half* r2, int r2_xmin, int r2_bmod, int out_w, const bc_half* wstream,
                           const float* vec, bc_half* out0, bc_half* out1, bc_half* out2, int ldvt, unsigned long long* gn_tot, float ln_eps,
                           float alpha, const float* alpha_dev, const int* alpha_idx, int alpha_bstride, float* part, int nsplit,
                           bc_stream stream) {
    BC_CHECK_ARG(bc_rowchain_supported(channels, M, rows_per_batch), "bc_rowchain: needs 320 or 640 channels, M %% rows_per_batch == 0 and "
                 "rows_per_batch %% %d == 0 (channels=%d M=%d rows_per_batch=%d)", RC_BM, channels, M, rows_per_batch);
    BC_CHECK_ARG((kind >= BC_CHAIN_IN && kind <= BC_CHAIN_OUT_TAIL) || kind == BC_CHAIN_OUT_FFP, "bc_rowchain: unknown kind %d", kind);
    BC_CHECK_ARG(wstream && vec && (x || kind == BC_CHAIN_OUT_TAIL) && (out0 || kind == BC_CHAIN_OUT_FF || kind == BC_CHAIN_OUT_FFP),
                 "bc_rowchain: null pointer");
    const bool blob = (kind == BC_CHAIN_OUT || kind == BC_CHAIN_OUT_TAIL || kind == BC_CHAIN_OUT_FFP) && out1 != nullptr;
    if (kind == BC_CHAIN_OUT_FF || kind == BC_CHAIN_OUT_TAIL || kind == BC_CHAIN_OUT_FFP)
        BC_CHECK_ARG(part && nsplit >= 1 && (4 * channels / 128) % nsplit == 0, "bc_rowchain(OUT_FF / OUT_TAIL): needs the partial-sum buffer and nsplit "
                     "dividing the %d hidden chunks (nsplit=%d)", 4 * channels / 128, nsplit);
    RowChainArgs a;
    a.kind = kind; a.M = M; a.rows_per_batch = rows_per_batch;
    a.x = reinterpret_cast<const h16*>(x); a.affine = affine;
    a.gn_in = kind == BC_CHAIN_IN ? gn_in : nullptr; a.gn_gamma = gn_gamma; a.gn_beta = gn_beta; a.gn_groups = gn_groups; a.gn_eps = gn_eps;
    if (a.gn_in)
        BC_CHECK_ARG(gn_gamma && gn_beta && gn_groups > 0 && channels % gn_groups == 0 && channels / gn_groups >= 1 && !affine,
                     "bc_rowchain(IN): the in-kernel GroupNorm finalize needs gamma, beta, groups | channels (and no affine table)");
    a.res = reinterpret_cast<const h16*>(res); a.res2 = reinterpret_cast<const h16*>(res2);
    a.r2 = reinterpret_cast<const h16*>(r2); a.r2_xmin = r2_xmin; a.r2_bmod = r2_bmod > 0 ? r2_bmod : 1; a.out_w = out_w > 0 ? out_w : 1;
    a.wstream = reinterpret_cast<const uint4*>(wstream);
    a.wave_frags = bc_rowchain_stream_frags(channels, kind, blob ? 1 : 0, nsplit);
    a.part = part; a.nsplit = nsplit > 0 ? nsplit : 1;
    a.vec = vec;
    a.out0 = reinterpret_cast<h16*>(out0); a.out1 = reinterpret_cast<h16*>(out1); a.out2 = reinterpret_cast<h16*>(out2);
    a.ldvt = ldvt; a.gn_tot = gn_tot; a.ln_eps = ln_eps;
    a.alpha = alpha; a.alpha_dev = alpha_dev; a.alpha_idx = alpha_idx; a.alpha_bstride = alpha_bstride;
    a.stamps = nullptr;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (kind == BC_CHAIN_IN)
        BC_CHECK_ARG(out1 && out2 && ldvt >= rows_per_batch && ldvt % 8 == 0, "bc_rowchain(IN): needs out1 (q|k), out2 (V^T) and ldvt >= rows_per_batch, ldvt %% 8 == 0");
    if (kind == BC_CHAIN_MID) BC_CHECK_ARG(res && out1, "bc_rowchain(MID): needs res and out1");
    if (kind == BC_CHAIN_OUT_FF) BC_CHECK_ARG(res, "bc_rowchain(OUT_FF): needs res");
    if (kind == BC_CHAIN_OUT_FFP) {
        BC_CHECK_ARG(res && res2 && !gn_tot, "bc_rowchain(OUT_FFP): needs res and res2; the statistics totals belong to bc_rowchain_sum");
        BC_CHECK_ARG(!r2 || (out_w > 0 && rows_per_batch % out_w == 0), "bc_rowchain(OUT_FFP): r2 needs out_w dividing rows_per_batch");
    }
    if (kind == BC_CHAIN_OUT || kind == BC_CHAIN_OUT_TAIL) {
        BC_CHECK_ARG((res || kind == BC_CHAIN_OUT_TAIL) && res2, "bc_rowchain(OUT / OUT_TAIL): needs res and res2");
        BC_CHECK_ARG(!r2 || (out_w > 0 && rows_per_batch % out_w == 0), "bc_rowchain(OUT): r2 needs out_w dividing rows_per_batch");
    }
    return channels == 320 ? dispatch_chain<320>(kind, blob, a, s) : dispatch_chain<640>(kind, blob, a, s);
}

extern "C" int bc_rowchain_sum(int channels, int M, int rows_per_batch, const bc_half* part, int nsplit, bc_half* out0, unsigned long long* gn_tot,
                               bc_half* out1, bc_stream stream) {
    BC_CHECK_ARG(bc_rowchain_supported(channels, M, rows_per_batch), "bc_rowchain_sum: needs 320 or 640 channels, M %% rows_per_batch == 0 and "
                 "rows_per_batch %% %d == 0 (channels=%d M=%d rows_per_batch=%d)", RC_BM, channels, M, rows_per_batch);
    BC_CHECK_ARG(part && out0 && nsplit >= 1 && nsplit <= 40, "bc_rowchain_sum: null pointer or nsplit = %d outside 1..40", nsplit);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (channels == 320)
        hipLaunchKernelGGL(rowchain_sum_kernel<320>, dim3(M / SUM_ROWS), dim3(640), 0, s, reinterpret_cast<const h16*>(part), nsplit, M, rows_per_batch,
                           reinterpret_cast<h16*>(out0), gn_tot, reinterpret_cast<h16*>(out1));
    else
        hipLaunchKernelGGL(rowchain_sum_kernel<640>, dim3(M / SUM_ROWS), dim3(640), 0, s, reinterpret_cast<const h16*>(part), nsplit, M, rows_per_batch,
                           reinterpret_cast<h16*>(out0), gn_tot, reinterpret_cast<h16*>(out1));
    BC_CHECK_LAUNCH();
    return 0;
}

#else   // BC_ROWCHAIN_MIDX_TU: rowchain_midx.hip
}  // namespace

// ---- cross-attention inside the chain (BC_CHAIN_MIDX) ------------------------------------------------------------------------------
namespace {

// K rows [B * T][ldk] and V^T [B][C][ldvt] of one block's context -> per (image, wave) fragment streams in the consumption order of
// the MIDX launch: per head of the wave, K as [k-step][key tile] then V^T as [key k-step][value tile], every fragment masked to the
// head's channels and to keys < T (zeros elsewhere: a k-step of the operand image also holds the neighbouring head's channels).
template <int RC_C>
__global__ void rowchain_pack_kv_kernel(const h16* __restrict__ k, int ldk, const h16* __restrict__ vt, int ldvt, int B, int T, long long kvf,
                                        uint4* __restrict__ out) {
    using XC = XCfg<RC_C>;
    constexpr int NW = RCfg<RC_C>::NW;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * NW * kvf * 64) return;
    const int lane = (int)(idx & 63);
    const long long fi = idx >> 6;
    const int f = (int)(fi % kvf), w = (int)((fi / kvf) % NW), b = (int)(fi / (kvf * NW));
    h16 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (h16)0.f;
    if (f < XC::FR_WAVE) {
        const int hh = f / XC::FR_HEAD, g = f % XC::FR_HEAD;
        const int c_h = 80 * w + XC::D * hh;
        if (g < XC::NKT * XC::QKS) {
            const int ks = g / XC::NKT, t = g % XC::NKT;
            const int key = 16 * t + (lane & 15), ch0 = 32 * ((c_h >> 5) + ks) + 8 * (lane >> 4);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (key < T && ch0 + j >= c_h && ch0 + j < c_h + XC::D) v[j] = k[((size_t)b * T + key) * ldk + ch0 + j];
        } else {
            const int g2 = g - XC::NKT * XC::QKS;
            const int ks = g2 / XC::VT, tv = g2 % XC::VT;
            const int tlo = XC::D == 40 && hh == 1 ? 2 : 0;
            const int ch = 80 * w + 16 * (tlo + tv) + (lane & 15), key0 = 32 * ks + 8 * (lane >> 4);
            if (ch >= c_h && ch < c_h + XC::D) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (key0 + j < T) v[j] = vt[((size_t)b * RC_C + ch) * ldvt + key0 + j];
            }
        }
    }
    out[idx] = *reinterpret_cast<const uint4*>(v);
}

}  // namespace

extern "C" long long bc_rowchain_kv_frags(int channels) {
    // fragments of ONE (image, wave) K / V^T stream incl. the padding the ring reads past its end; -1: unsupported
    if (channels == 320) return XCfg<320>::FR_WAVE + RC_RPAD;
    if (channels == 640) return XCfg<640>::FR_WAVE + RC_RPAD;
    return -1;
}

extern "C" int bc_rowchain_pack_kv(const bc_half* k, int ldk, const bc_half* vt, int ldvt, int B, int T, int channels, bc_half* out, bc_stream stream) {
    BC_CHECK_ARG((channels == 320 || channels == 640) && k && vt && out && B > 0 && T > 0 && T <= 80 && ldk >= channels && ldvt >= T,
                 "bc_rowchain_pack_kv: needs 320 or 640 channels (8 heads), 1 <= T <= 80 context tokens, ldk >= channels, ldvt >= T (channels=%d T=%d)", channels, T);
    const long long kvf = bc_rowchain_kv_frags(channels);
    const long long total = (long long)B * (channels / 80) * kvf * 64;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (channels == 320)
        hipLaunchKernelGGL(rowchain_pack_kv_kernel<320>, dim3(bc_ceil_div(total, 256)), dim3(256), 0, s, reinterpret_cast<const h16*>(k), ldk,
                           reinterpret_cast<const h16*>(vt), ldvt, B, T, kvf, reinterpret_cast<uint4*>(out));
    else
        hipLaunchKernelGGL(rowchain_pack_kv_kernel<640>, dim3(bc_ceil_div(total, 256)), dim3(256), 0, s, reinterpret_cast<const h16*>(k), ldk,
                           reinterpret_cast<const h16*>(vt), ldvt, B, T, kvf, reinterpret_cast<uint4*>(out));
    BC_CHECK_LAUNCH();
    return 0;
}

extern "C" int bc_rowchain_midx(int channels, int M, int rows_per_batch, const bc_half* x, const bc_half* res, const bc_half* wstream, const float* vec,
                                const bc_half* kvstream, int n_ctx, float attn_scale, bc_half* out0, bc_half* out1, float ln_eps, bc_stream stream) {
    BC_CHECK_ARG(bc_rowchain_supported(channels, M, rows_per_batch), "bc_rowchain_midx: needs 320 or 640 channels, M %% rows_per_batch == 0 and "
                 "rows_per_batch %% %d == 0 (channels=%d M=%d rows_per_batch=%d)", RC_BM, channels, M, rows_per_batch);
    BC_CHECK_ARG(x && res && wstream && vec && kvstream && out0 && out1 && n_ctx >= 1 && n_ctx <= 80 && attn_scale > 0.f,
                 "bc_rowchain_midx: null pointer, or n_ctx = %d outside 1..80", n_ctx);
    RowChainArgs a = {};
    a.kind = BC_CHAIN_MIDX; a.M = M; a.rows_per_batch = rows_per_batch;
    a.x = reinterpret_cast<const h16*>(x); a.res = reinterpret_cast<const h16*>(res);
    a.r2_bmod = 1; a.out_w = 1; a.nsplit = 1;
    a.wstream = reinterpret_cast<const uint4*>(wstream);
    a.wave_frags = bc_rowchain_stream_frags(channels, BC_CHAIN_MID, 0, 1);
    a.vec = vec;
    a.res2 = reinterpret_cast<const h16*>(kvstream); a.r2_xmin = n_ctx; a.alpha = attn_scale;      // (see RowChainArgs)
    a.out0 = reinterpret_cast<h16*>(out0); a.out1 = reinterpret_cast<h16*>(out1);
    a.ln_eps = ln_eps;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return channels == 320 ? launch_chain<320, BC_CHAIN_MIDX, false>(a, s) : launch_chain<640, BC_CHAIN_MIDX, false>(a, s);
}
#endif  // BC_ROWCHAIN_MIDX_TU
