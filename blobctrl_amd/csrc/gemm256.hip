// Large-M dense GEMM for gfx950 (BC_TILE_G256, round 6): 256 x 256 tiles, 8 waves, the "8-phase" LDS-DMA pipeline, persistent workgroups.
//
// Same contract as gemm.hip / gemm_fast.hip (C = epilogue(A . W^T), both operands K-contiguous fp16, fp32 accumulation) for the dense
// projections that carry the 1280-channel levels once a batch is large (M = B.H.W >= 2048 token rows): the Linear / 1 x 1 convolution
// layers of diffusers/src/diffusers/models/attention.py:1161-1167 (GEGLU feed-forward), attention_processor.py:2191-2224 (to_q / to_k /
// to_v / to_out), transformers/transformer_2d.py:479-527 (proj_in / proj_out) and BlobNet's zero-convs (blobctrl/models/blobnet.py:860-864).
// gemm_fast.hip's best tile (256 x 128, 3-stage LDS-DMA ring, one barrier per k-tile) runs those at 370-670 TFLOP/s (profiles/r6_c3_*);
// the vendor library on the same box and data does 1351 at 4096^3 (profiles/r5_blas_ceiling.txt).  This kernel: 1257 / 1263 TFLOP/s at
// 4096^3 / 8192^3 and 900-970 at the network's K = 1280 shapes stand-alone (tools/gemm8p_probe.hip, profiles/r6_gemm8p_probe.txt).
//
// Structure (cdna_hip_programming.md section 5 "The 256^2 8-phase template", rebuilt from its description):
//   * 8 waves = 2 (M halves) x 4 (N quarters); a wave owns 128 x 64 outputs as 8 x 4 accumulator tiles of v_mfma_f32_16x16x32_f16 (128
//     registers); the product is formed swapped (D^T = W . X^T), so a lane holds 4 consecutive output columns of one row;
//   * operands go global -> LDS by LDS-DMA in HALF-TILES of 16 KB: per 64-k tile B0, A0, B1, A1, where "0" / "1" is the part every wave
//     needs in phases {0, 1} / {2, 3} (A: its rows 0-63 / 64-127; B: its columns 0-31 / 32-63).  A wave issues 2 LDS-DMA instructions per
//     half-tile, one half-tile per phase, SEVEN half-tiles ahead of the phase that first reads it; one counted s_waitcnt vmcnt(6) per k-tile
//     and raw s_barriers keep 3..7 half-tiles in flight across the barriers;
//   * the LDS image is made of 1-KiB sub-tiles [16 rows][32 k] - exactly one MFMA operand fragment - with 64-byte rows whose 16-byte chunk
//     index is XOR-ed with 2 for rows >= 8: every ds_read_b128 of a fragment is conflict-free over the four 16-lane service groups
//     (MI355X_MICROARCH.md LDS).  LDS-DMA lands lane-linearly, so the swizzle sits on the per-lane SOURCE address;
//   * a phase = {fragment reads (12 / 4 / 8 / 0) + 2 LDS-DMA issues | s_barrier | 16 MFMAs = one 64 x 32 quadrant over the tile's 64 k |
//     s_barrier}; the two wave groups (one wave of each on every SIMD) run ONE BARRIER APART, so one group's MFMAs run beside the other
//     group's reads and DMA issues (the guide's 8-wave ping-pong);
//   * persistent workgroups (grid = min(tiles, CUs)): an XCD owns a contiguous run of tile ids (4-row grouped order), and the staging
//     stream simply runs on into the NEXT tile's first seven half-tiles during the last two k-tiles - the next tile's cold-start latency
//     hides under this tile's epilogue;
//   * epilogue through a 32-KiB LDS slab beside the staging buffers, 8 passes of 2 x 16 rows x 256 columns: accumulators parked raw
//     (XOR-swizzled 16-byte chunks: conflict-free both ways), re-read row-major, 8 output columns per thread = gemm_fast's shared
//     epilogue (bias, GEGLU / GELU / SiLU, scales, residual, BlobNet right-half residual, 16-byte stores, GroupNorm statistics totals);
//     BC_OUT_F16_T (V^T for the attention kernel) as 16-byte stores along the token axis.
//   * two-source A (BcGemm.A2 / C1 % 128 == 0: torch.cat([x, skip], 1) in front of a 1 x 1 shortcut, [g | h] in front of the fused
//     ff.net.2 + proj_out weight) by switching the staged source at k = C1; BcGemm.C_t / n_t0 (q | k row-major + V^T in one launch).
// Eligibility (bc_gemm256_eligible): dense A, M % 256 == 0, N % 256 == 0, K % 128 == 0, no split-K.
#include <stdlib.h>
#include <type_traits>
#include <vector>
#include "gemm_common.h"

using namespace bcg;

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __forceinline__ void glds16(const void* src, char* lds_dst) { __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_dst, 16, 0, 0); }

constexpr int G_BM = 256, G_BN = 256;
constexpr int HT = 16384;              // half-tile bytes
constexpr int KBUF = 4 * HT;           // one k-tile: [B0][A0][B1][A1]
constexpr int STG_BYTES = 2 * KBUF;    // 128 KiB of staging
constexpr int SLAB_BYTES = 32768;      // epilogue slab: 32 rows x 256 fp32
constexpr int G_LDS = STG_BYTES + SLAB_BYTES;

// LDS visibility only (the epilogue's slab): a __syncthreads() also waits vmcnt(0) - i.e. for the previous pass's global STORES to be
// acknowledged and for the next tile's run-ahead LDS-DMA: ~1 us per pass, 8 passes per tile (tools/g256_probe.py: 42.6 -> 35 us per K = 1280 tile)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Epilogue read-back of a wave's slab slice, as inline assembly: a compiler-visible LDS read with the next tile's run-ahead LDS-DMA in flight gets an
// s_waitcnt vmcnt(0) in front of it (the DMA's destination and the slab come from one shared array: "may alias"), which parks the epilogue until all
// seven half-tiles have landed.  lds_wait4 ties the four results to the wait, so that no use can be scheduled above it.
__device__ __forceinline__ f32x4 lds_read16_raw(const char* p) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((unsigned)(uintptr_t)(lptr_t)p) : "memory");
    return v;
}
__device__ __forceinline__ void lds_wait4(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
}

// The residual rows of the unrolled residual form, likewise: the compiler's own count for them became vmcnt(0) - the previous row-tile's stores
// included - once the read-back above was assembly.  Issue order per row-tile: wait, finish, the NEXT row-tile's two loads, this one's two stores;
// so vmcnt(2) at the next wait leaves exactly those stores in flight (row-tile 0: nothing was stored yet, vmcnt(0)).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 ld16_raw(const void* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int N>
__device__ __forceinline__ void vm_wait2(u32x4& a, u32x4& b) {
    if (N == 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b)::"memory");
    else asm volatile("s_waitcnt vmcnt(2)" : "+v"(a), "+v"(b)::"memory");
}

#define SB() __builtin_amdgcn_sched_barrier(0)
#define BAR() do { SB(); __builtin_amdgcn_s_barrier(); SB(); } while (0)

struct Frags {
    h16x8 a[4][2];                     // current A sub-block: 4 row tiles x 2 k-steps
    h16x8 b0[2][2], b1[2][2];          // B sub-blocks: 2 column tiles x 2 k-steps
};

template <int V> using IC = std::integral_constant<int, V>;

// The four phases of one 64-k tile held in staging buffer B.  MODE 0: every phase stages a half-tile (a1 = source of A1 of the NEXT
// k-tile; b0n / a0n / b1n = sources of B0 / A0 / B1 of the k-tile after that - possibly the next output tile's); 1: first tile of a
// draining pair (only phase 0 stages; phase 3 waits for everything); 2: last tile (nothing staged, nothing waited for).
template <int B, int MODE>
__device__ __forceinline__ void tile_phases(f32x4 (&acc)[8][4], Frags& f, const char* aB, const char* bB, char* stg, const char* a1, const char* b0n,
                                            const char* a0n, const char* b1n, unsigned voA1, unsigned voA0n, unsigned voB) {
    constexpr int O = B * KBUF, ON = (B ^ 1) * KBUF;
    auto mfma = [&](auto ac, auto bc) __attribute__((always_inline)) {
        constexpr int a = decltype(ac)::value, b = decltype(bc)::value;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SB();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    acc[a * 4 + rt][b * 2 + ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b ? f.b1[ct][ks] : f.b0[ct][ks], f.a[rt][ks], acc[a * 4 + rt][b * 2 + ct], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    // ---- phase 0: B0 + A0 fragments; stage A1 of the next k-tile
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) f.b0[ct][ks] = *reinterpret_cast<const h16x8*>(bB + O + 0 * HT + ct * 2048 + ks * 1024);
    SB();
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) f.a[rt][ks] = *reinterpret_cast<const h16x8*>(aB + O + 1 * HT + rt * 2048 + ks * 1024);
    SB();
    if (MODE <= 1) {
        glds16(a1 + voA1, stg + ON + 3 * HT);
        glds16(a1 + voA1 + 64, stg + ON + 3 * HT + 1024);
    }
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");      // the B0 reads have returned: B0's slot is restaged one phase later
    BAR();
    mfma(IC<0>{}, IC<0>{});
    BAR();
    // ---- phase 1: B1 fragments; stage B0 two k-tiles ahead
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) f.b1[ct][ks] = *reinterpret_cast<const h16x8*>(bB + O + 2 * HT + ct * 2048 + ks * 1024);
    SB();
    if (MODE == 0) {
        glds16(b0n + voB, stg + O + 0 * HT);
        glds16(b0n + voB + 64, stg + O + 0 * HT + 1024);
    }
    BAR();
    mfma(IC<0>{}, IC<1>{});
    BAR();
    // ---- phase 2: A1 fragments; stage A0 two k-tiles ahead
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) f.a[rt][ks] = *reinterpret_cast<const h16x8*>(aB + O + 3 * HT + rt * 2048 + ks * 1024);
    SB();
    if (MODE == 0) {
        glds16(a0n + voA0n, stg + O + 1 * HT);
        glds16(a0n + voA0n + 64, stg + O + 1 * HT + 1024);
    }
    BAR();
    mfma(IC<1>{}, IC<1>{});
    BAR();
    // ---- phase 3: no reads; stage B1 two k-tiles ahead; the wait that retires the next k-tile
    if (MODE == 0) {
        glds16(b1n + voB, stg + O + 2 * HT);
        glds16(b1n + voB + 64, stg + O + 2 * HT + 1024);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else if (MODE == 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    BAR();
    mfma(IC<1>{}, IC<0>{});
    BAR();
}

// The GEMM family's 8-column epilogue (gemm_common.h epi8_store) for this kernel, whose 256 x 256 tile has 128 outputs per thread and whose
// epilogue time is VALU issue + one exposed memory round trip (in-kernel stamps, -DBC_DIAGNOSTICS: 28.6k of a K = 1280 tile's 85k cycles before
// this form).  Differences, none of them in the arithmetic's order per output except the bias: (i) the BIAS is the accumulators' initial value
// (record: acc = b; acc += a w), so no bias load sits in the epilogue - vmcnt is in-order, and a load issued there waits behind the next
// tile's seven run-ahead half-tiles; (ii) column scales and GroupNorm statistics cost nothing when the launch does not use them
// (launch-uniform flags); (iii) the residual rows of a pass are requested before any of its arithmetic.
// LD: which global LOADS the pass loop may hold (the compiler's s_waitcnt insertion is exact in straight-line code, and falls back to vmcnt(0) -
// every store of the pass before and the whole run-ahead of the next tile - where a load sits under a branch: that alone was 2/3 of this epilogue):
// 0 none (bias / activation / scales / statistics), 1 the residual R (requested one pass ahead), 2 all of them (row vector, R2, per-batch alpha).
// ACTK: the activation known at compile time (0 none, 1 GEGLU) or 2 = the launch's (run-time chain).
template <int LD, int ACTK>
__device__ __forceinline__ uint4 epi8_finish(const GemmArgs& g, int n_first, const float alpha, float (&v)[8], const float (&gt)[8], int m, float (&gs)[8],
                                            float (&gq)[8], const uint4 rraw, const bool scaled, const bool stats) {
    const BcGemm& p = g.p;
    int b = 0, pix = m;
    if (LD == 2 && (p.rowvec || p.R2 || p.alpha_bstride > 0)) {
        b = (int)fdiv((unsigned)m, g.div_rpb);
        pix = m - b * (int)g.div_rpb.d;
    }
    if (LD == 2 && p.rowvec) {
        const uint4 raw = bc_ld16(rowvec_base(p) + (size_t)b * p.ld_rowvec + n_first);
        const h16* rh = reinterpret_cast<const h16*>(&raw);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += (float)rh[j];
    }
    if (ACTK == 0) {
    } else if (ACTK == 1 || p.act == BC_ACT_GEGLU) {
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const f32x2 gl = bc_gelu_f2((f32x2){gt[j], gt[j + 1]});
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
    if (scaled) {                                           // column scales (LD 2 only: a load) x the launch's scalar alpha
        if (LD == 2 && p.colscale) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= p.colscale[n_first + j] * alpha;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= alpha;
        }
    }
    if (LD == 2 && p.alpha_bstride > 0) {
        const float ab = batch_alpha(g, b);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= ab;
    }
    if (LD == 1 || (LD == 2 && p.R)) {
        const h16* rh = reinterpret_cast<const h16*>(&rraw);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += (float)rh[j];
    }
    if (LD == 2 && p.R2) {
        const int y = (int)fdiv((unsigned)pix, g.div_outw);
        const int x = pix - y * (int)g.div_outw.d;
        if (x >= p.r2_xmin) {
            int bmod = p.r2_bmod;                           // (laundered: the reciprocal of a hoisted modulus would stay live across the k-loop)
            asm volatile("" : "+s"(bmod));
            const int bb = b % bmod;
            const uint4 raw = bc_ld16(reinterpret_cast<const h16*>(p.R2) + ((size_t)bb * g.div_rpb.d + pix) * p.ldr2 + n_first);
            const h16* rh = reinterpret_cast<const h16*>(&raw);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += (float)rh[j];
        }
    }
    uint4 outraw;
    h16* o = reinterpret_cast<h16*>(&outraw);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (h16)v[j];
    if (stats) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = (float)o[j];
            gs[j] += f;
            gq[j] += f * f;
        }
    }
    return outraw;
}

// acc[i][0..3] for a RUN-TIME (wave-uniform) pass index i: an if-chain over compile-time indices (scalar branches), so that the
// accumulators keep static register numbers while the epilogue's pass loop stays a real loop.  Unrolled eight times the epilogue was 60 KB of
// code - the kernel 82 KB against a 64 KB instruction cache - and cost 13-14 us per tile whatever it did (K sweep, tools/g256_probe.py).
__device__ __forceinline__ void acc_row(const f32x4 (&acc)[8][4], int i, f32x4 (&w)[4]) {
#define BC_ROW(I) if (i == I) { w[0] = acc[I][0]; w[1] = acc[I][1]; w[2] = acc[I][2]; w[3] = acc[I][3]; }
    BC_ROW(0) else BC_ROW(1) else BC_ROW(2) else BC_ROW(3) else BC_ROW(4) else BC_ROW(5) else BC_ROW(6) else BC_ROW(7)
#undef BC_ROW
}

// tile index -> (row tile, column tile): runs of 4 row tiles walk the columns, so that consecutive ids share operand panels
__device__ __forceinline__ void tile_coords(int id, int tiles_m, int tiles_n, int& tm, int& tn) {
    constexpr int GM = 4;
    const int full = (tiles_m / GM) * GM * tiles_n;
    if (id < full) {
        const int grp = id / (GM * tiles_n), r = id - grp * GM * tiles_n;
        tm = grp * GM + (r % GM);
        tn = r / GM;
    } else {                                               // the last tiles_m % 4 row tiles: row-major
        const int r = id - full;
        tm = (tiles_m / GM) * GM + r / tiles_n;
        tn = r % tiles_n;
    }
}

// GEN: false = the three unrolled epilogue forms of the large projections (plain / GEGLU / residual), true = the rolled general form; two code
// objects because together the forms' tile-invariant values do not fit the 64 registers the accumulators and fragments leave across the k-loop.
__host__ __device__ inline bool g256_general(const BcGemm& p) {
    if (p.rowvec || p.R2 || p.alpha_bstride > 0 || p.colscale || p.gn_tot) return true;
    if (p.act == BC_ACT_NONE) return false;
    return !(p.act == BC_ACT_GEGLU && !p.R);
}

template <bool GEN>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const BcGemm& p = g.p;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int tiles_m = p.M / G_BM, tiles_n = p.N / G_BN, ntiles = tiles_m * tiles_n;
    // persistent schedule: XCD x (= blockIdx % 8) owns the contiguous run [start, start + cnt) of tile ids; its workgroups (slot = blockIdx / 8)
    // take the run round-robin, so that the tiles in flight on an XCD are neighbours
    const int G = gridDim.x;
    int j, jstep, jend;
    if (G >= 8) {
        const int xcd = blockIdx.x & 7, q = ntiles >> 3, r = ntiles & 7;
        const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        j = start + (blockIdx.x >> 3);
        jstep = G >> 3;
        jend = start + q + (xcd < r ? 1 : 0);
    } else {
        j = blockIdx.x; jstep = G; jend = ntiles;
    }
    if (j >= jend) return;                                  // (never with the launcher's grid; all waves leave together)

    const char* Ab = reinterpret_cast<const char*>(p.A);
    const char* A2b = reinterpret_cast<const char*>(p.A2);          // second source: columns k >= C1 (C1 % 128 == 0), row stride p.lda2
    const char* Wb = reinterpret_cast<const char*>(p.W);
    const long long lda2 = 2ll * p.lda, ldw2 = 2ll * p.ldw, ldb2 = 2ll * p.lda2;
    const long long c1b = A2b ? 2ll * p.C1 : (1ll << 60);           // byte offset inside a row where the second source starts
    // staging: this wave fills sub-tiles 2 wave, 2 wave + 1 (the two k-steps of one 16-row group) of every half-tile; a lane lands in
    // (row lane >> 2, chunk lane & 3) of the sub-tile (lane-linear) and fetches the chunk that the read-side swizzle expects there
    const int srow = lane >> 2;
    const int skq = (lane & 3) ^ (((lane >> 5) & 1) << 1);
    const unsigned voA = (unsigned)(((wave >> 2) * 128 + (wave & 3) * 16 + srow) * lda2 + skq * 16);       // + 64 rows: the "1" half
    const unsigned voA2 = (unsigned)(((wave >> 2) * 128 + (wave & 3) * 16 + srow) * ldb2 + skq * 16);      // the same rows of the second source
    const unsigned voB = (unsigned)(((wave >> 1) * 64 + (wave & 1) * 16 + srow) * ldw2 + skq * 16);        // + 32 rows: the "1" half
    const long long a_half = 64 * lda2, a2_half = 64 * ldb2, b_half = 32 * ldw2;
    char* stg = smem + wave * 2048;
    // fragment reads: lane (r = lane & 15, kq = lane >> 4) reads chunk kq ^ (2 if r >= 8) of row r
    const int foff = (lane & 15) * 64 + ((((lane >> 4) ^ (((lane >> 3) & 1) << 1))) << 4);
    const char* aB = smem + wr * 8192 + foff;
    const char* bB = smem + wc * 4096 + foff;
    float* slab = reinterpret_cast<float*>(smem + STG_BYTES);

    int tm, tn;
    tile_coords(j, tiles_m, tiles_n, tm, tn);
    const char* cA = Ab + (long long)tm * G_BM * lda2;     // wave-uniform bases of the current tile's operand panels (k = 0)
    const char* cA2 = A2b ? A2b + (long long)tm * G_BM * ldb2 : nullptr;
    const char* cB = Wb + (long long)tn * G_BN * ldw2;
    // source of the A half-tile `half` (0 / 1) at byte offset kb of the tile's rows: (wave-uniform pointer, per-lane offset) - the
    // first source below C1, the second from C1 on (every k-tile lies inside one source: C1 % 64 == 0)
    auto a_src = [&](const char* a, const char* a2, long long kb, int half, unsigned& vo) __attribute__((always_inline)) -> const char* {
        if (kb < c1b) { vo = voA; return a + kb + half * a_half; }
        vo = voA2;
        return a2 + (kb - c1b) + half * a2_half;
    };
    const int T = p.K / BK;                                 // even, >= 2
    // prologue: half-tiles 0..6 = k-tile 0 (B0 A0 B1 A1), k-tile 1 (B0 A0 B1)
    glds16(cB + voB, stg + 0 * HT);                    glds16(cB + voB + 64, stg + 0 * HT + 1024);
    glds16(cA + voA, stg + 1 * HT);                    glds16(cA + voA + 64, stg + 1 * HT + 1024);
    glds16(cB + b_half + voB, stg + 2 * HT);           glds16(cB + b_half + voB + 64, stg + 2 * HT + 1024);
    glds16(cA + a_half + voA, stg + 3 * HT);           glds16(cA + a_half + voA + 64, stg + 3 * HT + 1024);
    glds16(cB + 128 + voB, stg + KBUF + 0 * HT);       glds16(cB + 128 + voB + 64, stg + KBUF + 0 * HT + 1024);
    glds16(cA + 128 + voA, stg + KBUF + 1 * HT);       glds16(cA + 128 + voA + 64, stg + KBUF + 1 * HT + 1024);
    glds16(cB + b_half + 128 + voB, stg + KBUF + 2 * HT); glds16(cB + b_half + 128 + voB + 64, stg + KBUF + 2 * HT + 1024);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    BAR();

    f32x4 acc[8][4];
    Frags f;
    const float alpha = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(scalar_alpha(p))));      // (an SGPR: launch-uniform, live across the k-loop)
    // launch-uniform epilogue flags as ONE scalar, decoded inside the tile loop from a laundered copy (as booleans they are three SGPR pairs live
    // across the k-loop, and the allocator parks what does not fit in vector registers): GEGLU | transposed output | scaled (the per-column
    // scale multiply of the epilogue is skipped otherwise)
    const int eflags = __builtin_amdgcn_readfirstlane((p.act == BC_ACT_GEGLU ? 1 : 0) | (p.out_mode == BC_OUT_F16_T ? 2 : 0) | ((p.colscale != nullptr || alpha != 1.0f) ? 4 : 0));

#ifdef BC_DIAGNOSTICS
    unsigned long long* const stamps = g.halo_stamps;       // [workgroup][tile < 4][16]: tile start, k-loop done, epilogue done (s_memtime of wave 0)
    int tile_no = 0;
#define G256_STAMP(k) do { if (stamps && threadIdx.x == 0 && tile_no < 4) stamps[((size_t)blockIdx.x * 4 + tile_no) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define G256_STAMP(k) do { } while (0)
#endif
    for (;;) {
        G256_STAMP(0);
        const int jn = j + jstep;
        const bool has_next = jn < jend;                    // (workgroup-uniform)
        const char *nA = cA, *nA2 = cA2, *nB = cB;
        int tmn = tm, tnn = tn;
        if (has_next) {
            tile_coords(jn, tiles_m, tiles_n, tmn, tnn);
            nA = Ab + (long long)tmn * G_BM * lda2;
            nA2 = A2b ? A2b + (long long)tmn * G_BM * ldb2 : nullptr;
            nB = Wb + (long long)tnn * G_BN * ldw2;
        }
        // accumulators start at the BIAS of their four columns (epi8_finish i): lane (lane >> 4) owns columns 64 wc + 32 (jj >> 1) + 16 (jj & 1) + 4 (lane >> 4) ..+4
        const float* bias_e = p.bias;                       // (laundered: the hoisted null test is one more boolean across the k-loop)
        asm volatile("" : "+s"(bias_e));
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (bias_e) bv = *reinterpret_cast<const f32x4*>(bias_e + tn * G_BN + wc * 64 + (jj >> 1) * 32 + (jj & 1) * 16 + 4 * (lane >> 4));
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i][jj] = bv;
        }
        if (wr == 1) BAR();                                 // the second wave group runs one barrier behind the first
        long long kb = 0;                                   // byte offset of the pair's first k-tile inside a row
        for (int it = 0; it < T / 2; ++it) {
            // sources of the k-tiles two and three ahead: this tile's - or, in the last pair, the NEXT tile's k-tiles 0 and 1: the staging
            // stream runs on across the tile boundary (without a next tile it re-reads this tile's head: 7 half-tiles nobody uses)
            const bool last = it == T / 2 - 1;              // (uniform)
            const char *hA = last ? nA : cA, *hA2 = last ? nA2 : cA2;
            const long long hk = last ? 0 : kb + 256;       // byte offset of the k-tile two ahead inside ITS tile's rows
            const char* hB = last ? nB : cB + kb + 256;
            unsigned v1, v2, v3, v4;
            const char* s1 = a_src(cA, cA2, kb + 128, 1, v1);        // A1 of the next k-tile
            const char* s2 = a_src(hA, hA2, hk, 0, v2);              // A0 two ahead
            const char* s3 = a_src(hA, hA2, hk, 1, v3);              // A1 two ahead (staged in the second tile's phase 0)
            const char* s4 = a_src(hA, hA2, hk + 128, 0, v4);        // A0 three ahead
            tile_phases<0, 0>(acc, f, aB, bB, stg, s1, hB, s2, hB + b_half, v1, v2, voB);
            tile_phases<1, 0>(acc, f, aB, bB, stg, s3, hB + 128, s4, hB + b_half + 128, v3, v4, voB);
            kb += 256;
        }
        if (wr == 0) BAR();                                 // both groups in step again
        G256_STAMP(1);

        // ------------------------------------------------------------------------------------------------ epilogue of tile (tm, tn)
        // (lane-derived epilogue addresses come from a laundered thread id: they are tile-invariant, and hoisted out of the tile loop they
        //  would stay live across the k-loop, where the 192 accumulator + fragment registers leave no room for them)
        // (the id itself is rebuilt from the wave number, an SGPR, and the lane count - keeping threadIdx.x alive across the k-loop costs the
        //  register that tips the kernel into scratch)
        int tid_e;                                          // (volatile: the lane count itself must not be hoisted out of the tile loop either)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(tid_e));
        tid_e |= wave << 6;
#define tid tid_e
        const int lane_e = tid_e & 63;
#define lane lane_e
        int ef = eflags;
        asm volatile("" : "+s"(ef));
        const bool geglu = ef & 1, all_transposed = ef & 2, scaled = ef & 4;
        const int m0 = tm * G_BM, n0 = tn * G_BN;
        const int wrow = wr * 16 + (lane & 15);             // slab row this lane parks into
        char* wbase = reinterpret_cast<char*>(slab) + wrow * 1024;
        const int wsw = wrow & 7, wc4 = wc * 16 + (lane >> 4);
        // BcGemm.C_t: the column tiles from n_t0 on go TRANSPOSED into C_t (q | k row-major + V^T for the attention kernel out of one launch)
        const bool transposed = all_transposed || (p.C_t != nullptr && n0 >= p.n_t0);      // (tile-uniform)
        if (!transposed) {
            // Row-major tiles leave through WAVE-PRIVATE slices of the slab - no workgroup barrier in the epilogue (round 6: the two barriers per
            // 32-row pass of the shared-slab form, the waits they made every wave inherit and the run-time accumulator select were 20k of a tile's
            // cycles; this form is bounded by the LDS round trip alone).  Per accumulator row-tile i a wave parks its 16 x 64 fp32 block (4 KB,
            // 16-byte chunks XOR-swizzled by the row: reads and writes both spread evenly over the banks), and reads it back as
            // lane -> (row lane >> 2, the 8 columns (lane & 3) * 8 and the 8 columns 32 further): with GEGLU those are a value chunk and its gate,
            // otherwise two output chunks; a store instruction then covers 64 contiguous bytes of each of 16 rows.
            char* const ws = reinterpret_cast<char*>(slab) + wave * 4096;
            char* const wrowp = ws + (lane & 15) * 256;
            const int wx = lane & 15, wq = lane >> 4;
            // read-back: one 8-column output chunk per lane - GEGLU: (row lane >> 2, value chunk lane & 3 and its gate 32 columns further), 16 rows at once;
            // otherwise (row lane >> 3, chunk lane & 7), the block's rows 0..7 and 8..15 in two sub-passes
            const int nsub = geglu ? 1 : 2;
            const int rrow = geglu ? lane >> 2 : lane >> 3;
            const int rc = geglu ? (lane & 3) * 2 : (lane & 7) * 2;
            const int n_first = geglu ? n0 / 2 + wc * 32 + (lane & 3) * 8 : n0 + wc * 64 + (lane & 7) * 8;
            const int mrow = m0 + wr * 128 + rrow;
            const bool stats = p.gn_tot != nullptr;
            float gs[8], gq[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { gs[q] = 0.f; gq[q] = 0.f; }
            auto passes = [&](auto LDK, auto AK) {
                constexpr int LD = decltype(LDK)::value, ACTK = decltype(AK)::value;
                const h16* const Rb = reinterpret_cast<const h16*>(p.R) + n_first;
                u32x4 rn[2] = {(u32x4){0u, 0u, 0u, 0u}, (u32x4){0u, 0u, 0u, 0u}};
                if (LD == 1) {                              // residual rows of row-tile 0 (LD 1 is never GEGLU: two sub-passes)
                    rn[0] = ld16_raw(Rb + (size_t)mrow * p.ldr);
                    rn[1] = ld16_raw(Rb + (size_t)(mrow + 8) * p.ldr);
                }
                auto body = [&](const int i, const f32x4 (&w)[4]) __attribute__((always_inline)) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
                        *reinterpret_cast<f32x4*>(wrowp + ((((jj >> 1) * 8 + (jj & 1) * 4 + wq) ^ wx) << 4)) = w[jj];
                    uint4 rr[2] = {make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u)}, o[2];
                    float v[2][8];
                    f32x4 lo[2], hi[2];
#pragma unroll
                    for (int c = 0; c < 2; ++c) {           // GEGLU: c = value / gate of one row; otherwise the two sub-passes' rows
                        const int row = geglu ? rrow : rrow + 8 * c, ch = geglu ? rc + 8 * c : rc;
                        const char* const rp = ws + row * 256;
                        lo[c] = lds_read16_raw(rp + ((ch ^ row) << 4));
                        hi[c] = lds_read16_raw(rp + (((ch + 1) ^ row) << 4));
                    }
                    lds_wait4(lo[0], hi[0], lo[1], hi[1]);
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        v[c][0] = lo[c].x; v[c][1] = lo[c].y; v[c][2] = lo[c].z; v[c][3] = lo[c].w;
                        v[c][4] = hi[c].x; v[c][5] = hi[c].y; v[c][6] = hi[c].z; v[c][7] = hi[c].w;
                    }
                    const int m = mrow + i * 16;
                    if (LD == 1) {
                        if (i == 0) vm_wait2<0>(rn[0], rn[1]);
                        else vm_wait2<2>(rn[0], rn[1]);
#pragma unroll
                        for (int c = 0; c < 2; ++c) rr[c] = make_uint4(rn[c].x, rn[c].y, rn[c].z, rn[c].w);
                    }
                    if (ACTK == 1 || (ACTK == 2 && geglu)) {
                        if (LD == 2) rr[0] = p.R ? bc_ld16(Rb + (size_t)m * p.ldr) : make_uint4(0u, 0u, 0u, 0u);
                        o[0] = epi8_finish<LD, ACTK>(g, n_first, alpha, v[0], v[1], m, gs, gq, rr[0], scaled, LD == 2 && stats);
                    } else {
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            if (LD == 2) rr[c] = p.R ? bc_ld16(Rb + (size_t)(m + 8 * c) * p.ldr) : make_uint4(0u, 0u, 0u, 0u);
                            o[c] = epi8_finish<LD, ACTK>(g, n_first, alpha, v[c], v[c], m + 8 * c, gs, gq, rr[c], scaled, LD == 2 && stats);
                        }
                    }
                    if (LD == 1 && i < 7) {                 // the next row-tile's residual rows, ahead of this one's stores (vmcnt is in-order); every
                        const int mn = mrow + (i + 1) * 16; // such load MUST meet its vm_wait2: its registers are only the compiler's until then
                        rn[0] = ld16_raw(Rb + (size_t)mn * p.ldr);
                        rn[1] = ld16_raw(Rb + (size_t)(mn + 8) * p.ldr);
                    }
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        if (c >= nsub) continue;
#ifdef BC_DIAGNOSTICS
                        if ((g.halo_dbg & 2) && !(o[c].x == 0x7fff7fffu && o[c].y == 0x12345678u)) continue;      // stores skipped (values kept alive)
#endif
                        bc_st16(reinterpret_cast<h16*>(p.C) + (size_t)(m + 8 * c) * p.ldc + n_first, o[c]);
                    }
                };
                if (LD == 2) {                              // (the rarely used general form stays rolled: its finish is long)
#pragma unroll 1
                    for (int i = 0; i < 8; ++i) {
                        f32x4 w[4];
                        acc_row(acc, i, w);
                        body(i, w);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) body(i, acc[i]);
                }
            };
            using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
            if constexpr (GEN) {
                passes(I2{}, I2{});
            } else {                                        // unrolled: static accumulator rows, exact wait counts
                if (geglu) passes(I0{}, I1{});
                else if (p.R) passes(I1{}, I0{});
                else passes(I0{}, I0{});
            }
            if (GEN && stats) {
                // per-lane column partials -> the lanes of a wave that share a column chunk -> the wave's slice -> the two waves of a column group -> one atomic add per column
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    for (int o = geglu ? 4 : 8; o < 64; o <<= 1) {
                        gs[q] += __shfl_xor(gs[q], o);
                        gq[q] += __shfl_xor(gq[q], o);
                    }
                }
                int tid2;                                   // (a fresh lane count: the epilogue's stays out of the pass loop's live set)
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(tid2));
                const int lane2 = tid2;
                tid2 |= wave << 6;
                float* const wsf = reinterpret_cast<float*>(slab) + wave * 1024;
                if (lane2 < (geglu ? 4 : 8)) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        wsf[(lane2 * 8 + q) * 2] = gs[q];
                        wsf[(lane2 * 8 + q) * 2 + 1] = gq[q];
                    }
                }
                lds_barrier();
                const int TSO = geglu ? G_BN / 2 : G_BN, wcols = geglu ? 32 : 64;     // output columns of the tile / of a wave
                if (tid2 < TSO) {
                    const int b = (int)fdiv((unsigned)m0, g.div_rpb), nb = geglu ? n0 / 2 : n0;
                    bc_gn_tot_add_slot(p.gn_tot + (size_t)b * g.n_out * BC_GN_TOT_WORDS, nb + tid2, nb, nb + TSO, bc_gn_cg(g.n_out), tm, [&](int k) {
                        const int c = k - nb, cw = geglu ? c >> 5 : c >> 6, cc = c - cw * wcols;
                        float s = 0.f, q2 = 0.f;
#pragma unroll
                        for (int w = 0; w < 2; ++w) {
                            s += slab[(w * 4 + cw) * 1024 + cc * 2];
                            q2 += slab[(w * 4 + cw) * 1024 + cc * 2 + 1];
                        }
                        return make_float2(s, q2);
                    });
                }
                lds_barrier();
            }
        } else {
            // BC_OUT_F16_T: thread = (column, 8 consecutive tokens) -> one 16-byte store into C[(b N + n) ldc + pix]  (bias and alpha only)
#pragma unroll 1
            for (int i = 0; i < 8; ++i) {
                f32x4 w[4];
                acc_row(acc, i, w);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    *reinterpret_cast<f32x4*>(wbase + (((wc4 + (jj >> 1) * 8 + (jj & 1) * 4) ^ wsw) << 4)) = w[jj];
                lds_barrier();
#pragma unroll 1
                for (int c = 0; c < 2; ++c) {
                    const int item = tid + 512 * c;
                    const int col = item & 255, ch = item >> 8;                     // ch 0..3: rows 8 ch .. 8 ch + 7 of the slab
                    const int n = n0 + col;
                    const int m = m0 + (ch >> 1) * 128 + i * 16 + (ch & 1) * 8;
                    uint4 outraw;
                    h16* o = reinterpret_cast<h16*>(&outraw);
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const int rl = ch * 8 + r;
                        const float x = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(slab) + rl * 1024 + ((((col >> 2) ^ (rl & 7))) << 4) + (col & 3) * 4);
                        o[r] = (h16)(x * alpha);              // (the bias is the accumulators' initial value)
                    }
                    const int b = (int)fdiv((unsigned)m, g.div_rpb);
                    const int pix = m - b * (int)g.div_rpb.d;
                    if (all_transposed) bc_st16(reinterpret_cast<h16*>(p.C) + ((size_t)b * g.n_out + n) * p.ldc + pix, outraw);
                    else bc_st16(reinterpret_cast<h16*>(p.C_t) + ((size_t)b * (p.N - p.n_t0) + (n - p.n_t0)) * p.ldc_t + pix, outraw);
                }
                lds_barrier();
            }
        }
#undef tid
#undef lane
        G256_STAMP(2);
#ifdef BC_DIAGNOSTICS
        ++tile_no;
#endif
        if (!has_next) break;
        j = jn; tm = tmn; tn = tnn; cA = nA; cA2 = nA2; cB = nB;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the last tile's run-ahead LDS-DMA must have landed before the LDS is given back)
}

}  // namespace

int bc_gemm256_ok(const BcGemm& p) {
    if (p.a_mode != BC_A_DENSE || p.splitk > 1) return 0;
    if (p.A2 && (p.C1 <= 0 || p.C1 % 128 != 0 || p.C1 >= p.K)) return 0;                 // (two sources: every k-tile pair inside one of them)
    if (p.M <= 0 || p.M % G_BM != 0 || p.N % G_BN != 0 || p.K % 128 != 0 || p.K < 128) return 0;
    if (p.a_affine || p.a_tot1 || p.ln_colsum || p.w_bstride || p.vec_bstride || p.sm_group) return 0;
    if (p.bias && (reinterpret_cast<uintptr_t>(p.bias) & 15)) return 0;                  // (the accumulators start at the bias: 16-byte loads)
    if (p.C_t) {                                            // row-major columns [0, n_t0) + transposed columns [n_t0, N): plain projections only
        const int rpb = p.rows_per_batch > 0 ? p.rows_per_batch : p.M;
        if (p.out_mode != BC_OUT_F16 || p.n_t0 <= 0 || p.n_t0 % G_BN != 0 || p.n_t0 >= p.N || p.act != BC_ACT_NONE || p.R || p.R2 || p.gn_tot || p.rowvec ||
            p.colscale || p.alpha_bstride || rpb % 8 != 0 || p.M % rpb != 0 || p.ldc_t % 8 != 0)
            return 0;
    }
    if (p.out_mode == BC_OUT_F16_T) {
        if (p.act != BC_ACT_NONE || p.rowvec || p.colscale || p.R || p.R2 || p.gn_tot || p.alpha_bstride) return 0;
        const int rpb = p.rows_per_batch > 0 ? p.rows_per_batch : p.M;
        if (rpb % 8 != 0 || p.M % rpb != 0) return 0;
    } else if (p.out_mode != BC_OUT_F16) {
        return 0;
    }
    // per-lane source offsets are 32-bit: 256 rows of either operand must stay below 4 GiB
    if (256ll * 2 * std::max(std::max(p.lda, p.lda2), p.ldw) >= (1ll << 31)) return 0;
    return 1;
}

extern "C" int bc_gemm256_eligible(int M, int N, int K, int C1, int out_mode, int rows_per_batch, int want_gn) {
    BcGemm p = {};
    p.a_mode = BC_A_DENSE; p.M = M; p.N = N; p.K = K; p.lda = C1 > 0 ? C1 : K; p.ldw = K; p.out_mode = out_mode; p.splitk = 1; p.alpha = 1.0f;
    p.rows_per_batch = rows_per_batch;
    if (C1 > 0) { p.A2 = reinterpret_cast<const bc_half*>(&p); p.C1 = C1; p.lda2 = K - C1; }     // (only tested for non-null)
    if (want_gn && (rows_per_batch <= 0 || rows_per_batch % G_BM != 0)) return 0;      // a tile's rows must lie inside one image
    return bc_gemm256_ok(p);
}

int bc_gemm256_launch(const GemmArgs& g, hipStream_t stream) {
    const BcGemm& p = g.p;
    const int ntiles = (p.M / G_BM) * (p.N / G_BN);
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = 256;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    }
    int grid = std::min(ntiles, cus);
    if (grid >= 8) grid &= ~7;                              // (the schedule deals whole workgroups to the 8 XCDs)
    static std::atomic<unsigned long long> lds_set{0};
    static std::atomic<unsigned long long> lds_set_gen{0};
    const bool gen = g256_general(p);
    if (gen) BC_CHECK_HIP(bc_set_max_lds(lds_set_gen, reinterpret_cast<const void*>(&gemm256_kernel<true>), G_LDS));
    else BC_CHECK_HIP(bc_set_max_lds(lds_set, reinterpret_cast<const void*>(&gemm256_kernel<false>), G_LDS));
#ifdef BC_DIAGNOSTICS
    // BC_G256_STAMPS=1 (diagnostic builds: HIPCC_EXTRA=-DBC_DIAGNOSTICS; synchronises): where a workgroup's cycles go, per tile
    // value = bit 0 (always) | bit 1: the epilogue's global stores are skipped (WRONG results: what the stores cost) | (workgroups << 8): grid override
    static const int stamp_bits = getenv("BC_G256_STAMPS") ? atoi(getenv("BC_G256_STAMPS")) : 0;
    const bool want_stamps = stamp_bits != 0;
    if (want_stamps) {
        if (stamp_bits >> 8) grid = std::min(grid, stamp_bits >> 8);
        static unsigned long long* buf = nullptr;
        if (!buf) BC_CHECK_HIP(hipMalloc(&buf, 256 * 64 * sizeof(unsigned long long)));
        BC_CHECK_HIP(hipMemsetAsync(buf, 0, 256 * 64 * sizeof(unsigned long long), stream));
        GemmArgs g2 = g;
        g2.halo_stamps = buf;
        g2.halo_dbg = stamp_bits & 2;
        if (gen) hipLaunchKernelGGL(gemm256_kernel<true>, dim3(grid), dim3(512), G_LDS, stream, g2);
        else hipLaunchKernelGGL(gemm256_kernel<false>, dim3(grid), dim3(512), G_LDS, stream, g2);
        BC_CHECK_HIP(hipStreamSynchronize(stream));
        std::vector<unsigned long long> h(256 * 64);
        BC_CHECK_HIP(hipMemcpy(h.data(), buf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double kl[4] = {0}, ep[4] = {0};
        int n[4] = {0};
        for (int w = 0; w < grid; ++w)
            for (int t = 0; t < 4; ++t) {
                const unsigned long long* q = &h[(w * 4 + t) * 16];
                if (!q[2]) continue;
                kl[t] += (double)(q[1] - q[0]); ep[t] += (double)(q[2] - q[1]); ++n[t];
            }
        for (int t = 0; t < 4; ++t)
            if (n[t]) fprintf(stderr, "[g256 stamps] M=%d N=%d K=%d act=%d grid=%d tile %d: k-loop %.0f ticks, epilogue %.0f (wave 0; n=%d)\n", p.M, p.N, p.K, p.act, grid, t,
                              kl[t] / n[t], ep[t] / n[t], n[t]);
        return 0;
    }
#endif
    if (gen) hipLaunchKernelGGL(gemm256_kernel<true>, dim3(grid), dim3(512), G_LDS, stream, g);
    else hipLaunchKernelGGL(gemm256_kernel<false>, dim3(grid), dim3(512), G_LDS, stream, g);
    BC_CHECK_LAUNCH();
    return 0;
}
