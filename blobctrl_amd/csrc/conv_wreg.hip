// 3x3 / stride-1 / pad-1 convolution, LDS-resident input halo, WEIGHTS STREAMED STRAIGHT INTO VGPRs (gfx950, v_mfma_f32_16x16x32_f16).
//
// Same contract, tile (8 x 16 output pixels x 160 output channels per workgroup), halo staging (raw rows by LDS-DMA into their final
// swizzled slots, in-place GroupNorm + SiLU pass, zero padding after the activation), split-K over 64-channel chunks and epilogue as
// conv_halo.hip, which it replaces for the ResBlock convolutions (D/models/resnet.py:327-341, :351-366).  What changed is the main loop.
// conv_halo.hip brings the weights of every tap in by LDS-DMA (20 KiB per tap and workgroup, one barrier per tap) and measures ~1500
// cycles per tap against 640 cycles of MFMA: the LDS-DMA path of a CU delivers ~22 B/clk here (tools/wreg_probe.hip), the
// ordinary global-load path more than twice that.  So:
//   * the B operand never touches LDS: weights are pre-packed (bc_conv_wreg_pack / weights.pack_conv_wreg) into one CONTIGUOUS stream
//     of 1-KiB MFMA fragments per wave - lane l of a fragment holds W[n0 + (l & 15)][k0 + 8 (l >> 4) .. + 8] - read with one
//     global_load_dwordx4 per fragment into a register ring that runs a whole chunk (9 taps) ahead: 144-216 KiB in flight per CU;
//   * 8 waves = 4 column groups of 3 | 2 | 2 | 3 MFMA column tiles x 2 K halves of each 64-channel chunk; wave w runs on SIMD w % 4, the
//     groups are assigned so that every SIMD owns 5 tile columns (waves 0, 1, 6, 7 three tiles; 2 - 5 two): no weight byte is fetched
//     twice and the matrix pipes are evenly loaded;
//   * every wave covers all 8 pixel rows of the tile, so an A fragment (16 pixels of halo row r at x shift kx) serves the three taps
//     (ky = 0..2) that read row r: 10 ds_read_b128 per kx group instead of 24, 0.4 LDS reads per MFMA;
//   * one barrier per CHUNK: the halo images are triple-buffered; the four two-tile waves (which have matrix-pipe time to spare, and run
//     at raised priority) issue the LDS-DMA of chunk c + 2 and run the in-place pass over chunk c + 1 while everybody multiplies chunk c.
// Every LDS access of the loop and the whole weight ring are inline asm with hand-counted lgkmcnt / vmcnt waits: hipcc orders plain
// LDS reads behind every LDS-DMA in flight, and with an LDS-DMA in the same loop it drains the memory counter (vmcnt(0)) at the first
// use of any loaded register.  Two rules follow for registers the asm loads into (both learnt the hard way): they must
// reach their wait on a straight path (no load or wait under a run-time condition: the last chunk is PEELED instead), and nothing may
// still be in flight when such a register dies - hipcc re-uses it at once and the late data lands in somebody else's value.
// Vector-memory operations complete in issue order, so a counted vmcnt states exactly which of them have landed.
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include <vector>
#include "gemm_common.h"

using namespace bcg;

namespace {

__device__ __attribute__((aligned(16))) unsigned int g_zero_line_w[4] = {0u, 0u, 0u, 0u};

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
constexpr int HALO_BYTES = 24 * 1024;           // one 64-channel chunk of the halo: 192 pixel slots x 128 B (the last 12 are padding)
constexpr int NBUF = 3;                         // halo images: multiplied | being transformed | landing
constexpr int OFF_AB = NBUF * HALO_BYTES;
constexpr int MAX_CH = 40;                      // channel chunks per workgroup (affine table: 512 B per chunk)
constexpr int TS = HBN + 4;                     // split-K: row stride (floats) of the fp32 partial tile
constexpr int LDS_LOOP = OFF_AB + MAX_CH * 512;
constexpr int LDS_EPI = 8 * 16384;              // epilogue: one 16-KB slice per wave (64 pixels x 256 B); the split-K tile (HBM * TS * 4 bytes) fits below
constexpr int OFF_EPI = LDS_LOOP > LDS_EPI ? LDS_LOOP : LDS_EPI;   // bias[160] | time-embedding row[160] (floats), staged in the prologue
constexpr int LDS_TOTAL = OFF_EPI + 2 * HBN * 4;
static_assert(LDS_TOTAL <= 160 * 1024 && HPIX * 128 <= HALO_BYTES && HBM * TS * 4 <= LDS_EPI, "LDS budget");
constexpr int FIN_MAX_CH = 2752;                // in-kernel GroupNorm finalize: channel span (incl. group straddle) per workgroup
constexpr int OFF_FIN = LDS_LOOP;               // its scratch: [span][2] doubles + [groups][2] floats, behind the loop's LDS
constexpr int LDS_TOTAL_FIN = OFF_FIN + FIN_MAX_CH * 16 + 1024 > LDS_TOTAL ? OFF_FIN + FIN_MAX_CH * 16 + 1024 : LDS_TOTAL;
static_assert(LDS_TOTAL_FIN <= 160 * 1024, "LDS budget (in-kernel finalize)");

template <int N>
__device__ __forceinline__ void wait_vm_c() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

template <int V> using IC = std::integral_constant<int, V>;

// AFFINE: 0 = plain convolution, 1 = affine table from global memory (bc_gn_finalize ran), 2 = GroupNorm finalize in the prologue
template <int AFFINE>
__global__ __launch_bounds__(512) void conv_wreg_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const BcGemm& p = g.p;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave & 1;                                      // K half of every chunk
    const int grp = ((wave >> 1) & 1) | ((wave >> 2) << 1);       // column group <- waves {0,1} {2,3} {4,5} {6,7}
    const bool three = grp == 0 || grp == 3;                      // 3 | 2 | 2 | 3 column tiles
    const int tile0 = grp == 0 ? 0 : grp == 1 ? 3 : grp == 2 ? 5 : 7;

    // Every kernel-argument field the prologue reads is fetched HERE, in two clumps of scalar loads: left alone hipcc loads each field where
    // its first use sits, behind a dozen separate s_waitcnt - at the cold start of a launch each one a scalar-cache miss of its own in
    // front of the workgroup's first memory request (BC_WREG_STAMPS: 4.3k cycles from entry to the first request).
    asm volatile("" ::"s"(p.A), "s"(p.A2), "s"(p.W), "s"(p.Hv), "s"(p.Wv), "s"(p.Hin), "s"(p.Win), "s"(p.lda), "s"(p.lda2), "s"(p.C1), "s"(p.Cin),
                 "s"(p.splitk), "s"(p.a_act), "s"(g.halo_tpi), "s"(g.halo_tx), "s"(g.halo_nch), "s"(g.halo_cps), "s"(g.nband), "s"(g.halo_stamps));
    asm volatile("" ::"s"(g.wr_plane.mul), "s"(g.wr_plane.shift), "s"(g.wr_plane.d), "s"(g.wr_gx.mul), "s"(g.wr_gx.shift), "s"(g.wr_gx.d), "s"(g.wr_gy.mul),
                 "s"(g.wr_gy.shift), "s"(g.wr_gy.d), "s"(g.wr_tpi.mul), "s"(g.wr_tpi.shift), "s"(g.wr_tx.mul), "s"(g.wr_tx.shift), "s"(p.bias), "s"(p.rowvec),
                 "s"(p.rowvec_idx), "s"(p.rowvec_step), "s"(p.ld_rowvec), "s"(g.halo_dbg), "s"(g.div_rpb.d));
    if (AFFINE == 2)
        asm volatile("" ::"s"(p.a_tot1), "s"(p.a_tot2), "s"(p.a_gamma), "s"(p.a_beta), "s"(p.a_groups), "s"(p.a_eps), "s"(g.wr_cpg.mul), "s"(g.wr_cpg.shift));
    // (index arithmetic with host-made reciprocals: the seven integer divisions here were ~250 scalar instructions in front of the first
    //  memory request of every workgroup)
    const int plane = (int)g.wr_plane.d;
    const int lin3 = bc_xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), plane * gridDim.z);
    const int split = (int)fdiv((unsigned)lin3, g.wr_plane);
    const int lin = lin3 - split * plane;
    const int q_gx = (int)fdiv((unsigned)lin, g.wr_gx), q_gy = (int)fdiv((unsigned)lin, g.wr_gy);
    const int tile = g.nband ? lin - q_gy * (int)g.wr_gy.d : q_gx;
    const int ntile = g.nband ? q_gy : lin - q_gx * (int)g.wr_gx.d;
    const int n0 = ntile * HBN;
    const int b = (int)fdiv((unsigned)tile, g.wr_tpi);
    const int tin = tile - b * g.halo_tpi;
    const int ty_ = (int)fdiv((unsigned)tin, g.wr_tx);
    const int ty0 = ty_ * TH, tx0 = (tin - ty_ * g.halo_tx) * TW;
    // H x W = the image the convolution runs over; with UPS2 (exact nearest-neighbour 2x upsample in front: D/models/upsampling.py:
    // F.interpolate(scale_factor=2.0, mode="nearest") -> conv) that is the VIRTUAL image and a halo pixel reads source pixel (y/2, x/2)
    const int H = p.Hv, W = p.Wv;
    const bool ups2 = p.Hv != p.Hin;
    const int c_begin = split * g.halo_cps;
    const int nch = min(g.halo_nch, c_begin + g.halo_cps) - c_begin;

    unsigned long long* const stamps = g.halo_stamps;        // BC_WREG_STAMPS diagnostics (null in production): waves 0 (three tiles) and 2 (staging)
    auto stamp = [&](int i) {
        if (stamps && lane == 0 && (wave == 0 || wave == 2))
            stamps[((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 2 + (wave >> 1)) * 16 + i] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);
    // The time-embedding row of a captured step sits in a per-edit table behind a device-side step counter.  The counter is fetched
    // with a SCALAR load (lgkmcnt): as a vector load its wait (vmcnt is in-order) sat in front of, or behind, everything the prologue
    // requests - an exposed memory round trip before the halo rows were even asked for.
    // (load and wait are ONE asm statement, placed where the prologue has everything else in flight: between a bare s_load and a later wait
    //  hipcc is free to move or spill the destination SGPR - it cannot know a load is still on its way to it - and with the scalar file
    //  full, as in the half-CU kernel, it did: the first launch of the step read its time-embedding row from a garbage index)

    const h16* __restrict__ A1 = reinterpret_cast<const h16*>(p.A);
    const h16* __restrict__ A2 = reinterpret_cast<const h16*>(p.A2);
    const h16* zero = reinterpret_cast<const h16*>(g_zero_line_w);

    // ---- staging duty (waves 2-5, 256 lanes): lane owns six 16-byte slots of every halo image: bytes [sidx * 16 + 4096 q, + 16) =
    // halo pixel hp = (sidx >> 3) + 32 q, 16-byte slot sidx & 7.  The swizzle (chunk c lives in slot c ^ swz(hx)) is applied to the
    // SOURCE address; the lane that brings a slot in by LDS-DMA is the lane that later normalises it in place.
    const int sidx = ((wave >= 2 ? wave - 2 : 0) << 6) | lane;
    int pixv[6], csubv[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int hp = (sidx >> 3) + 32 * q;
        const int hy = hp / HSTR, hx = hp - hy * HSTR;
        const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
        const bool inh = hp < HPIX;
        const bool in_img = inh && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        pixv[q] = !in_img ? -1 : ups2 ? (b * p.Hin + (gy >> 1)) * p.Win + (gx >> 1) : (b * H + gy) * W + gx;
        csubv[q] = (sidx & 7) ^ (((hx >> 1) & 3) << 1);
    }
    // raw rows of chunk c_begin + cl -> halo image `buf` (staging waves only).  `valid` false: six reads of the zero line instead (the
    // tail of the loop keeps the SAME number of memory operations per chunk: hipcc's vmcnt bookkeeping takes the minimum over the
    // paths that join, so a conditional load anywhere in the loop makes every wait behind it stricter by its count)
    auto issue_a = [&](int cl, int buf, bool valid) {
        const int k0 = (c_begin + cl) * 64;
        const bool second = A2 != nullptr && k0 >= p.C1;      // wave-uniform (C1 % 64 == 0)
        const h16* src = second ? A2 : A1;
        const int stride = second ? p.lda2 : p.lda;           // (element offsets fit 32 bits: checked by the launcher)
        const int kin = second ? k0 - p.C1 : k0;
        char* dst = smem + buf * HALO_BYTES + (wave - 2) * 1024;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int off = pixv[q] * stride + kin + csubv[q] * 8;
            const h16* s = (valid && pixv[q] >= 0) ? src + off : zero;
            glds16(s, dst + q * 4096);
        }
    };
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)(smem);
    const unsigned slot_addr = lds0 + sidx * 16;
    const unsigned ab_base = lds0 + OFF_AB;
    struct Pending { u32x4v raw; f32x4v t0, t1, t2, t3; };
    // in-place pass over one 16-byte slot: `sa` = its LDS address, `ab_addr` = the (a, b) pairs of its 8 channels, `outside` = the pixel
    // lies outside the image (zero padding, applied AFTER norm + activation)
    auto tr_issue = [&](unsigned sa, unsigned ab_addr, Pending& pd) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(pd.raw) : "v"(sa) : "memory");
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\t"
                     "ds_read_b128 %3, %4 offset:48"
                     : "=&v"(pd.t0), "=&v"(pd.t1), "=&v"(pd.t2), "=&v"(pd.t3) : "v"(ab_addr) : "memory");
    };
    auto tr_finish = [&](unsigned sa, bool outside, bool store, Pending& pd) {
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
        if (outside) outraw = (u32x4v){0u, 0u, 0u, 0u};
        if (store) asm volatile("ds_write_b128 %0, %1" ::"v"(sa), "v"(outraw) : "memory");
    };
    // FIRST image (prologue): every lane of the workgroup owns three 16-byte slots for the in-place pass: bytes [tid * 16 + 8192 q,
    // + 16), q < 3, = halo pixel (tid >> 3) + 64 q, slot tid & 7 (the staging waves waited for the LDS-DMA that filled them and a
    // barrier has passed since).  tinfo: per q, bit 0 = inside the halo, bit 1 = outside the image, bits 2-4 = 8-channel sub-chunk.
    unsigned tinfo = 0;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int hp = (tid >> 3) + 64 * q;
        const int hy = hp / HSTR, hx = hp - hy * HSTR;
        const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
        const unsigned inh = hp < HPIX, outside = !((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W);
        tinfo |= (inh | outside << 1 | (unsigned)((tid & 7) ^ (((hx >> 1) & 3) << 1)) << 2) << (8 * q);
    }
    const unsigned tslot = lds0 + tid * 16;
    // in the loop the pass belongs to the staging waves (the three-tile waves have neither the registers nor the issue slots to spare):
    // two of the lane's six DMA slots after each kx group, reads of both issued before the arithmetic
    auto transform_pair = [&](auto kxc, int cl, int buf) {
        constexpr int q0 = 2 * decltype(kxc)::value;
        Pending pa, pb;
        const unsigned sa = slot_addr + buf * HALO_BYTES + 4096 * q0, ab = ab_base + cl * 512;      // (64 channels x (a, b) x 4 bytes per chunk)
        const bool in0 = (sidx >> 3) + 32 * q0 < HPIX, in1 = (sidx >> 3) + 32 * (q0 + 1) < HPIX;
        // (reads and arithmetic are unconditional - the padding slots of the image are readable - only the store is predicated: the
        // registers the asm reads land in reach their wait on a straight path)
        tr_issue(sa, ab + csubv[q0] * 64, pa);
        tr_issue(sa + 4096, ab + csubv[q0 + 1] * 64, pb);
        tr_finish(sa, pixv[q0] < 0, in0, pa);
        tr_finish(sa + 4096, pixv[q0 + 1] < 0, in1, pb);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    auto transform_first = [&]() {                                // image 0, all three slots with the reads issued together
        Pending pd[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) tr_issue(tslot + 8192 * q, ab_base + ((tinfo >> (8 * q + 2)) & 7) * 64, pd[q]);
#pragma unroll
        for (int q = 0; q < 3; ++q) tr_finish(tslot + 8192 * q, (tinfo >> (8 * q + 1)) & 1, (tinfo >> (8 * q)) & 1, pd[q]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    // ---- prologue: affine table of this chunk range -> LDS ----
    // ---- main loop + the K-half sum of the epilogue, per wave kind (NT = column tiles of this wave; STG = staging duty) ----
    float* tilef = reinterpret_cast<float*>(smem);
    auto body = [&](auto ntc, auto stgc) {
        constexpr int NT = decltype(ntc)::value;
        constexpr bool STG = decltype(stgc)::value != 0;
        constexpr int G = 3 * NT;                                 // fragments per kx group (3 ky x NT tiles)
        // this wave's fragment stream: [column tile block ntile][group, K half][chunk][kx][ky][tile][64 lanes x 8 halves]
        const long long per_chunk = 9 * 512;                      // halves per chunk and tile column
        // The ring is loaded and waited for by hand (inline asm): with the LDS-DMA of the staging waves in the same loop hipcc
        // treats the memory counter as out of order and drains it (vmcnt(0)) at every first use of a ring register.  Loads return in
        // order; at the top of a kx group the wave may have in flight: the two younger groups (2 G loads) and, in a staging wave, the
        // six LDS-DMA pieces of the chunk's halo prefetch - in whatever position, so they are always allowed for.
        u32x4v ring[3 * G];
        const unsigned lane16 = lane * 16;
        // fragments [GX * G, GX * G + G) of the chunk at BASE (wave-uniform); 13-bit immediate: one scalar base per four fragments
#define BC_WREG_LOAD_GROUP(GX, BASE)                                                                                                  \
    _Pragma("unroll") for (int f = 0; f < G; ++f) {                                                                                   \
        const h16* b4 = (BASE) + (((GX) * G + f) & ~3) * 512;                                                                          \
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(ring[(GX) * G + f]) : "v"(lane16), "s"(b4), "n"((((GX) * G + f) & 3) * 1024) : "memory"); \
    }
        const h16* wb = reinterpret_cast<const h16*>(p.W) + ((long long)ntile * 20 + 2 * tile0 + kg * NT) * g.halo_nch * per_chunk +
                        (long long)c_begin * NT * per_chunk;     // wave-uniform stream pointer (chunk cl)
        // Request order = order of use: the first two halo images, ring group 0, the tables; then the rest of chunk 0's weights.  (A
        // workgroup's first requests - 48 KiB of halo rows + 144-216 KiB of weights - take ~5000 cycles at the per-CU fetch rate: the
        // first MFMA waits only for what it needs.)
        // AFFINE == 2, channel span <= FIN_FAST_CH (round 5): the finalize no longer waits for the halo rows and the first weight fragments.
        // Its inputs (statistics totals, gamma, beta) are requested FIRST by the three-tile waves with inline-asm loads and waited for with
        // a counted vmcnt - vector-memory operations return in issue order, so they land ahead of everything requested behind them -
        // and only those waves compute: they issue no LDS-DMA, so hipcc has no reason to drain the memory counter in front of their
        // plain LDS accesses, and the two barriers inside are LDS-only (a __syncthreads() waits for vmcnt(0): in round 4 the first one
        // waited for 48 KiB of halo rows + the first ring group, and the rest of chunk 0's weights was requested behind the whole finalize:
        // 21k cycles to the first MFMA against 10k with a finalize launch - BC_WREG_STAMPS).
        constexpr int FIN_T = 4;                                  // channels per finalize thread (256 of them)
        const int fin_cpg = AFFINE == 2 ? p.Cin / p.a_groups : 1;
        const int fin_k_lo = c_begin * 64, fin_k_hi = fin_k_lo + nch * 64;
        const int fin_g_lo = (int)fdiv((unsigned)fin_k_lo, g.wr_cpg), fin_g_hi = AFFINE == 2 ? min(p.a_groups, (int)fdiv((unsigned)(fin_k_hi + fin_cpg - 1), g.wr_cpg)) : 1;
        const int fin_c_lo = fin_g_lo * fin_cpg, fin_nc = fin_g_hi * fin_cpg - fin_c_lo;
        const bool fin_fast = AFFINE == 2 && fin_nc <= 256 * FIN_T && !(g.halo_dbg & 0x100);   // (workgroup-uniform; nch * 64 <= fin_nc; BC_WREG_FIN_SLOW=1: the round-4 form)
        const int ft = ((wave < 2 ? wave : wave - 4) << 6) | lane;        // finalize thread id: waves 0, 1, 6, 7
        u32x4v fq[FIN_T][3];
        float fg[FIN_T], fb[FIN_T];
        if (AFFINE == 2 && !STG && fin_fast) {
            // (the two table pointers are pinned in scalar registers: a per-lane select between two kernel-argument FIELDS makes hipcc
            //  select the field's address and fetch the pointer itself with a vector load - an exposed round trip per channel)
            unsigned long long tp1 = (unsigned long long)p.a_tot1, tp2 = (unsigned long long)(p.A2 ? p.a_tot2 : p.a_tot1);
            asm volatile("" : "+s"(tp1), "+s"(tp2));
            const int C1s = p.A2 ? p.C1 : p.Cin, C2s = p.Cin - C1s;      // channels of the first / second source
#pragma unroll
            for (int j = 0; j < FIN_T; ++j) {
                const int cc = ft + 256 * j;
                const int c = fin_c_lo + (cc < fin_nc ? cc : 0);
                const bool second = c >= C1s;
                const unsigned long long ta = (second ? tp2 : tp1) + ((unsigned long long)((long long)b * (second ? C2s : C1s) + (second ? c - C1s : c))) * (BC_GN_TOT_WORDS * 8);
                const unsigned long long* t = reinterpret_cast<const unsigned long long*>(ta);
                asm volatile("global_load_dwordx4 %0, %3, off\n\tglobal_load_dwordx4 %1, %3, off offset:16\n\tglobal_load_dwordx4 %2, %3, off offset:32"
                             : "=&v"(fq[j][0]), "=&v"(fq[j][1]), "=&v"(fq[j][2]) : "v"(t) : "memory");
                const int i = ft + 256 * j;                               // table entry (channel fin_k_lo + i) this thread writes
                const int ci = fin_k_lo + (i < nch * 64 ? i : 0);
                asm volatile("global_load_dword %0, %2, off\n\tglobal_load_dword %1, %3, off" : "=&v"(fg[j]), "=&v"(fb[j]) : "v"(p.a_gamma + ci), "v"(p.a_beta + ci) : "memory");
            }
        }
        stamp(8);
        if (STG) {
            issue_a(0, 0, true);
            issue_a(1, 1, nch > 1);
        }
        BC_WREG_LOAD_GROUP(0, wb)
        stamp(9);
        if (AFFINE == 2 && fin_fast && !STG) {
            // the totals have landed (younger: this wave's ring group 0)
#define BC_FIN_TIE(j) "+v"(fq[j][0]), "+v"(fq[j][1]), "+v"(fq[j][2])
            asm volatile("s_waitcnt vmcnt(%12)" : BC_FIN_TIE(0), BC_FIN_TIE(1), BC_FIN_TIE(2), BC_FIN_TIE(3) : "n"(G) : "memory");
            asm volatile("s_waitcnt vmcnt(%8)" : "+v"(fg[0]), "+v"(fg[1]), "+v"(fg[2]), "+v"(fg[3]), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3])
                         : "n"(G) : "memory");
#undef BC_FIN_TIE
        }
        // bias and the time-embedding row of this image (epilogue): requested here - behind the finalize's counted wait, ahead of ring
        // groups 1 and 2 -, stored to LDS behind the first in-place pass.  Branch-free, raw bits: a conditional load or an immediate
        // fp16 -> fp32 conversion makes hipcc wait for it on the spot.
        stamp(10);
        unsigned rv_idx = 0;
        if (p.rowvec && p.rowvec_idx) asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rv_idx) : "s"(p.rowvec_idx) : "memory");
        const bool eok = p.splitk == 1 && tid < HBN;
        const float* epi_bp = (eok && p.bias) ? p.bias + n0 + tid : reinterpret_cast<const float*>(g_zero_line_w);
        const h16* epi_rp = (eok && p.rowvec) ? reinterpret_cast<const h16*>(p.rowvec) + (size_t)rv_idx * p.rowvec_step + (size_t)b * p.ld_rowvec + n0 + tid : zero;
        const float epi_bv = *epi_bp;
        const unsigned short epi_rraw = *reinterpret_cast<const unsigned short*>(epi_rp);
        if (AFFINE == 2 && fin_fast) {
            double* scr = reinterpret_cast<double*>(smem + OFF_FIN);           // [nc][2] = (sum, sum of squares) per channel
            float* stat = reinterpret_cast<float*>(scr + fin_nc * 2);          // [groups][2] = (mean, rstd)
            if (!STG) {
#pragma unroll
                for (int j = 0; j < FIN_T; ++j) {
                    const int cc = ft + 256 * j;
                    if (cc < fin_nc) {
                        double s_, q_;
                        auto w64 = [](unsigned lo, unsigned hi) { return (unsigned long long)lo | ((unsigned long long)hi << 32); };
                        bc_gn_tot_decode(w64(fq[j][0][0], fq[j][0][1]), w64(fq[j][0][2], fq[j][0][3]), w64(fq[j][1][0], fq[j][1][1]),
                                         w64(fq[j][1][2], fq[j][1][3]), w64(fq[j][2][0], fq[j][2][1]), w64(fq[j][2][2], fq[j][2][3]), s_, q_);
                        scr[cc * 2] = s_;
                        scr[cc * 2 + 1] = q_;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            stamp(11);
            if (!STG) {   // eight lanes per group: 32 groups per pass of the four finalize waves, fixed summation order
                const int sub = ft & 7;
                for (int gi = fin_g_lo + (ft >> 3); gi < fin_g_hi; gi += 32) {
                    double s_ = 0.0, q_ = 0.0;
                    for (int cj = sub; cj < fin_cpg; cj += 8) {
                        s_ += scr[((gi - fin_g_lo) * fin_cpg + cj) * 2];
                        q_ += scr[((gi - fin_g_lo) * fin_cpg + cj) * 2 + 1];
                    }
#pragma unroll
                    for (int o = 4; o > 0; o >>= 1) {
                        s_ += __shfl_xor(s_, o);
                        q_ += __shfl_xor(q_, o);
                    }
                    const double n = (double)g.div_rpb.d * fin_cpg;
                    const double mean = s_ / n;
                    double var = q_ / n - mean * mean;
                    if (var < 0.0) var = 0.0;
                    if (sub == 0) {
                        stat[(gi - fin_g_lo) * 2] = (float)mean;
                        stat[(gi - fin_g_lo) * 2 + 1] = (float)(1.0 / sqrt(var + (double)p.a_eps));
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            stamp(12);
            if (!STG) {
                float* abt = reinterpret_cast<float*>(smem + OFF_AB);
#pragma unroll
                for (int j = 0; j < FIN_T; ++j) {
                    const int i = ft + 256 * j;
                    if (i < nch * 64) {
                        const int gi = (int)fdiv((unsigned)(fin_k_lo + i), g.wr_cpg) - fin_g_lo;
                        const float a = stat[gi * 2 + 1] * fg[j];
                        abt[i * 2] = a;
                        abt[i * 2 + 1] = fb[j] - stat[gi * 2] * a;
                    }
                }
            }
            stamp(13);
            // (no barrier here: the table is read behind the "raw rows visible" barrier below)
        } else if (AFFINE == 2) {
            // GroupNorm finalize for the groups overlapping this workgroup's channels [k_lo, k_hi), from the statistics totals (six
            // words per channel, order-independent integer sums: bc_common.h), while the first halo rows and weight fragments are in
            // flight.  Scratch behind the loop's LDS (the halo images are already being written).
            const int cpg = p.Cin / p.a_groups;
            const int k_lo = c_begin * 64, k_hi = k_lo + nch * 64;
            const int g_lo = k_lo / cpg, g_hi = min(p.a_groups, (k_hi + cpg - 1) / cpg);
            const int c_lo = g_lo * cpg, nc = g_hi * cpg - c_lo;
            // gamma / beta of this thread's first two channels are requested FIRST: fetched behind the second barrier they were a
            // third exposed memory round trip of the prologue
            float gpre[2] = {0.f, 0.f}, bpre[2] = {0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (tid + 512 * j < nch * 64) {
                    gpre[j] = p.a_gamma[k_lo + tid + 512 * j];
                    bpre[j] = p.a_beta[k_lo + tid + 512 * j];
                }
            double* scr = reinterpret_cast<double*>(smem + OFF_FIN);           // [nc][2] = (sum, sum of squares) per channel
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
            {   // eight lanes per group: 64 groups per pass of the workgroup, fixed summation order
                const int sub = tid & 7;
                for (int gi = g_lo + (tid >> 3); gi < g_hi; gi += 64) {
                    double s = 0.0, q = 0.0;
                    for (int cj = sub; cj < cpg; cj += 8) {
                        s += scr[((gi - g_lo) * cpg + cj) * 2];
                        q += scr[((gi - g_lo) * cpg + cj) * 2 + 1];
                    }
#pragma unroll
                    for (int o = 4; o > 0; o >>= 1) {
                        s += __shfl_xor(s, o);
                        q += __shfl_xor(q, o);
                    }
                    const double n = (double)g.div_rpb.d * cpg;
                    const double mean = s / n;
                    double var = q / n - mean * mean;
                    if (var < 0.0) var = 0.0;
                    if (sub == 0) {
                        stat[(gi - g_lo) * 2] = (float)mean;
                        stat[(gi - g_lo) * 2 + 1] = (float)(1.0 / sqrt(var + (double)p.a_eps));
                    }
                }
            }
            __syncthreads();
            float* abt = reinterpret_cast<float*>(smem + OFF_AB);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int i = tid + 512 * j;
                if (i < nch * 64) {
                    const int gi = (k_lo + i) / cpg - g_lo;
                    const float a = stat[gi * 2 + 1] * gpre[j];
                    abt[i * 2] = a;
                    abt[i * 2 + 1] = bpre[j] - stat[gi * 2] * a;
                }
            }
            for (int i = tid + 1024; i < nch * 64; i += 512) {
                const int c = k_lo + i;
                const int gi = c / cpg - g_lo;
                const float a = stat[gi * 2 + 1] * p.a_gamma[c];
                abt[i * 2] = a;
                abt[i * 2 + 1] = p.a_beta[c] - stat[gi * 2] * a;
            }
            __syncthreads();                                              // (the scratch overlaps the epilogue vectors staged next)
        }
        if (AFFINE == 1) {   // affine table of the chunk range (bc_gn_finalize ran) -> LDS
            const float4* src = reinterpret_cast<const float4*>(p.a_affine + ((size_t)b * p.Cin + (size_t)c_begin * 64) * 2);
            float4* dst = reinterpret_cast<float4*>(smem + OFF_AB);
            for (int i = tid; i < nch * 32; i += 512) dst[i] = src[i];
        }
        BC_WREG_LOAD_GROUP(1, wb)
        BC_WREG_LOAD_GROUP(2, wb)
        stamp(1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G) : "memory");  // the halo rows (and ring group 0) have landed; groups 1 and 2 may fly
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                             // affine table and raw rows of chunks 0 and 1 visible
        asm volatile("" ::: "memory");
        stamp(6);
        if (AFFINE) transform_first();
        if (eok) {                                                // (the epilogue finds them in LDS; the region is clear of the finalize's scratch by now)
            float* ev = reinterpret_cast<float*>(smem + OFF_EPI);
            ev[tid] = epi_bv;
            ev[HBN + tid] = (float)__builtin_bit_cast(h16, epi_rraw);
        }
        stamp(7);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                             // halo image 0 complete
        asm volatile("" ::: "memory");
        stamp(2);

        int a_off[3];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int hx = (lane & 15) + kx;
            const int ch = 4 * kg + (lane >> 4);
            a_off[kx] = hx * 128 + ((ch ^ (((hx >> 1) & 3) << 1)) << 4);
        }
        f32x4v acc[8][NT];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[i][t] = (f32x4v){0.f, 0.f, 0.f, 0.f};

        // The staging waves carry 2/3 of a three-tile wave's MFMAs plus the in-place pass; their SIMD partner has nothing else to do.
        // Raised priority lets their MFMAs through first, the partner fills the matrix pipe while they normalise.
#ifndef BC_WREG_PRIO
#define BC_WREG_PRIO 1
#endif
        if ((BC_WREG_PRIO == 1 && STG) || (BC_WREG_PRIO == 2 && !STG)) __builtin_amdgcn_s_setprio(3);
        // The last chunk is peeled (the pass loop below is unrolled, `more` is a constant in each copy): a register the inline asm
        // loads into must reach its wait on a straight path - a refill under a run-time condition makes hipcc join two versions of
        // the ring at the end of the branch, possibly by a register copy issued while the load is still in flight.
        int buf = 0;                                              // image of chunk cl
        int cl = 0;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
        const bool more = pass == 0;
        const int cl_end = more ? nch - 1 : nch;
        for (; cl < cl_end; ++cl) {
            const int buf1 = buf + 1 == NBUF ? 0 : buf + 1;       // image of chunk cl + 1
            const int buf2 = buf1 + 1 == NBUF ? 0 : buf1 + 1;     // image of chunk cl + 2 (last read during chunk cl - 1)
            if (STG) issue_a(cl + 2, buf2, cl + 2 < nch);
            wb += 3 * G * 512;                                    // -> chunk cl + 1
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                // this group's fragments have landed.  Younger operations that may still fly: the two groups refilled after it and, in a
                // staging wave, the six LDS-DMA pieces issued at the top of the chunk; the LAST chunk refills nothing (a load still
                // in flight at the end of the loop would land in a register hipcc has meanwhile given to something else), so there
                // the count shrinks group by group
#define BC_WREG_WAIT(N)                                                                                                                 \
    if (NT == 3) asm volatile("s_waitcnt vmcnt(%9)" : "+v"(ring[kx * G]), "+v"(ring[kx * G + 1]), "+v"(ring[kx * G + 2]), "+v"(ring[kx * G + 3]), \
                              "+v"(ring[kx * G + 4]), "+v"(ring[kx * G + 5]), "+v"(ring[kx * G + 6]), "+v"(ring[kx * G + 7]),             \
                              "+v"(ring[kx * G + G - 1]) : "n"(N) : "memory");                                                          \
    else asm volatile("s_waitcnt vmcnt(%6)" : "+v"(ring[kx * G]), "+v"(ring[kx * G + 1]), "+v"(ring[kx * G + 2]), "+v"(ring[kx * G + 3]),       \
                      "+v"(ring[kx * G + 4]), "+v"(ring[kx * G + G - 1]) : "n"(N) : "memory");
                constexpr int D = STG ? 6 : 0;
                if (more || kx == 0) { BC_WREG_WAIT(2 * G + D) }
                else if (kx == 1) { BC_WREG_WAIT(G + D) }
                else { BC_WREG_WAIT(D) }
                const unsigned abase = lds0 + buf * HALO_BYTES + a_off[kx];
                u32x4v a[3];
                asm volatile("ds_read_b128 %0, %1" : "=v"(a[0]) : "v"(abase) : "memory");
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[1]) : "v"(abase), "n"(HSTR * 128) : "memory");
#pragma unroll
                for (int r = 0; r < 10; ++r) {
                    if (r + 2 < 10) {
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[(r + 2) % 3]) : "v"(abase), "n"((r + 2) * HSTR * 128) : "memory");
                        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a[r % 3])::"memory");
                    } else if (r + 1 < 10) {
                        asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(a[r % 3])::"memory");
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[r % 3])::"memory");
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const h16x8 af = __builtin_bit_cast(h16x8, a[r % 3]);
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const int i = r - ky;
                        if (i < 0 || i >= 8) continue;
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, ring[(kx * 3 + ky) * NT + t]), af, acc[i][t], 0, 0, 0);   // (swapped product: a lane holds 4 consecutive CHANNELS of one pixel)
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // in-place pass over chunk cl + 1 (staging waves; the rows landed before the barrier that ended chunk cl - 1)
                if (STG && AFFINE && more) {
                    if (kx == 0) transform_pair(IC<0>{}, cl + 1, buf1);
                    else if (kx == 1) transform_pair(IC<1>{}, cl + 1, buf1);
                    else transform_pair(IC<2>{}, cl + 1, buf1);
                }
                // this group's slots are free: the same group of the next chunk
                if (more) { BC_WREG_LOAD_GROUP(kx, wb) }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (STG) {                                            // the rows of chunk cl + 2 (LDS-DMA at the top of this chunk) have landed
                if (more) wait_vm_c<3 * G>();
                else wait_vm_c<0>();
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                         // image of chunk cl is free; image of chunk cl + 1 is complete
            asm volatile("" ::: "memory");
            buf = buf1;
        }
        }

        __builtin_amdgcn_s_setprio(0);
        stamp(3);
        // ---- the two K halves are summed through LDS (the halo images are dead: every wave passed the last barrier).  The partner
        // waves (same column group, other K half) hold the same fragment layout: each sends the half of its accumulators the OTHER one
        // finalises (K half 0: pixel rows 0-3, K half 1: rows 4-7) as 16-byte fragments, adds what it receives, and writes its own four
        // pixel rows of the row-major tile.
        char* xch = smem + tile0 * 8192;                          // [row tile 8][tile NT][lane 64] x 16 bytes per column group
        const int i_keep = kg * 4, i_send = 4 - i_keep;
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                *reinterpret_cast<f32x4v*>(xch + (((i_send + ii) * NT + t) * 64 + lane) * 16) = kg ? acc[ii][t] : acc[4 + ii][t];
        __syncthreads();
        f32x4v fin[4][NT];
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f32x4v o = *reinterpret_cast<const f32x4v*>(xch + (((i_keep + ii) * NT + t) * 64 + lane) * 16);
                fin[ii][t] = (kg ? acc[4 + ii][t] : acc[ii][t]) + o;
            }
        __syncthreads();                                          // every fragment has been read: the exchange area is free
        // fin[ii][t]: pixel (row tile i_keep + ii, x = lane & 15), channels tile0 * 16 + t * 16 + 4 (lane >> 4) .. + 3  (the swapped product)
        if (p.splitk > 1) {
            // split-K: the row-major fp32 tile for the slab store below
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    *reinterpret_cast<f32x4v*>(tilef + ((i_keep + ii) * 16 + (lane & 15)) * TS + tile0 * 16 + t * 16 + 4 * (lane >> 4)) = fin[ii][t];
            __syncthreads();
            return;
        }
        // ---- epilogue of an unsplit launch, per wave and without a workgroup barrier (round 6; before: the row-major fp32 tile of all eight waves,
        // a barrier, a 480-thread row pass - "k-half sum + stores" 13k of a 640-channel workgroup's 88k cycles, BC_WREG_STAMPS).  The wave parks its 64
        // pixels x NT * 16 channels in a private 16-KB slice (rows of 256 B, 16-byte chunk index XOR-ed with the row: conflict-free both
        // ways; LDS operations of one wave execute in order) and reads them back as (pixel, 8-channel chunk) per lane: NT = 3 - 8 pixels x 6 chunks
        // per pass (lanes with chunk position 6, 7 idle), 8 passes; NT = 2 - 16 pixels x 4 chunks, 4 passes.  Arithmetic as epi8_store, in its order.
        constexpr int RPP = NT == 3 ? 8 : 16, NP = 64 / RPP;
        char* const sl = smem + wave * 16384;
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int row = ii * 16 + (lane & 15), ck = t * 4 + (lane >> 4);
                *reinterpret_cast<f32x4v*>(sl + row * 256 + ((ck ^ (row & 15)) << 4)) = fin[ii][t];
            }
        int cpos, rsub;
        bool lane_on = true;
        if (NT == 3) {
            cpos = lane & 7;
            rsub = lane >> 3;
            lane_on = cpos < 6;
            if (!lane_on) cpos = 5;                               // (idle lanes repeat chunk 5 of their row; they do not count towards the statistics)
        } else {
            // 16 lanes served together read 4 pixels x 4 chunks: pixels {0, 1, 8, 9} + 2 (lane >> 4), so that their XOR-ed chunk sets are disjoint
            const int sub = (lane >> 2) & 3;
            cpos = lane & 3;
            rsub = 2 * (lane >> 4) + (sub & 1) + 8 * (sub >> 1);
        }
        const int ch0 = tile0 * 16 + cpos * 8, n_first = n0 + ch0;
        const int rpb = (int)g.div_rpb.d;
        const float alpha = scalar_alpha(p);
        const float* ev = reinterpret_cast<const float*>(smem + OFF_EPI) + ch0;
        float bias_v[8], rvec[8], cs[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            bias_v[j] = ev[j];
            rvec[j] = ev[HBN + j];
            cs[j] = (p.colscale ? p.colscale[n_first + j] : 1.0f) * alpha;
        }
        const float ab = p.alpha_bstride > 0 ? batch_alpha(g, b) : 1.0f;
        float gs[8], gq[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { gs[j] = 0.f; gq[j] = 0.f; }
        // HR: the launch has a residual (compile time: under a run-time `if (p.R)` the compiler moves each chunk's fp16 -> fp32 conversions up into
        // the load's block and awaits every request where it is made); H2: a right-half residual (per-lane loads under per-lane conditions)
        // (H2 = the general form: also an activation after the convolution - no layer of these networks has one; kept rolled, loads where used)
        auto passes = [&](auto H2K, auto HRK) {
            constexpr bool H2 = decltype(H2K)::value, HR = decltype(HRK)::value;
            uint4 rr[NP];
            int mrow[NP];
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int row = j * RPP + rsub;
                mrow[j] = b * rpb + (ty0 + i_keep + (row >> 4)) * W + tx0 + (row & 15);
                if (HR) rr[j] = bc_ld16(reinterpret_cast<const h16*>(p.R) + (size_t)mrow[j] * p.ldr + n_first);
            }
            if (!H2) __builtin_amdgcn_sched_barrier(0);               // (all requests before the first use: the waits then count exactly)
#pragma unroll H2 ? 1 : NP
            for (int j = 0; j < NP; ++j) {
                const int row = j * RPP + rsub;
                const f32x4v lo = *reinterpret_cast<const f32x4v*>(sl + row * 256 + (((2 * cpos) ^ (row & 15)) << 4));
                const f32x4v hi = *reinterpret_cast<const f32x4v*>(sl + row * 256 + (((2 * cpos + 1) ^ (row & 15)) << 4));
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] += bias_v[q];
                if (p.rowvec) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] += rvec[q];
                }
                if (H2) {
                    if (p.act == BC_ACT_GELU) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = bc_gelu_f(v[q]);
                    } else if (p.act == BC_ACT_SILU) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = bc_silu_f(v[q]);
                    } else if (p.act == BC_ACT_QUICK_GELU) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = bc_quick_gelu_f(v[q]);
                    }
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] *= cs[q];
                if (p.alpha_bstride > 0) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] *= ab;
                }
                if (HR) {
                    const h16* rh = reinterpret_cast<const h16*>(&rr[j]);
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] += (float)rh[q];
                }
                if (H2 && p.R) {
                    const uint4 r1 = bc_ld16(reinterpret_cast<const h16*>(p.R) + (size_t)mrow[j] * p.ldr + n_first);
                    const h16* rh = reinterpret_cast<const h16*>(&r1);
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] += (float)rh[q];
                }
                if (H2 && p.R2) {
                    const int pix = mrow[j] - b * rpb;
                    const int x = pix - (int)fdiv((unsigned)pix, g.div_outw) * (int)g.div_outw.d;
                    if (x >= p.r2_xmin) {
                        const uint4 r2 = bc_ld16(reinterpret_cast<const h16*>(p.R2) + ((size_t)(b % p.r2_bmod) * rpb + pix) * p.ldr2 + n_first);
                        const h16* rh = reinterpret_cast<const h16*>(&r2);
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] += (float)rh[q];
                    }
                }
                uint4 outraw;
                h16* o = reinterpret_cast<h16*>(&outraw);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    o[q] = (h16)v[q];
                    const float f = lane_on ? (float)o[q] : 0.f;
                    gs[q] += f;
                    gq[q] += f * f;
                }
                // (no branch around the store: the idle lanes of NT = 3 repeat chunk 5's store of their row, same address and data - under a per-lane
                //  branch the compiler cannot count the store, and every wait for a residual chunk covers the previous row's store again)
                bc_st16(reinterpret_cast<h16*>(p.C) + (size_t)mrow[j] * p.ldc + n_first, outraw);
            }
        };
        if (p.R2 || p.act != BC_ACT_NONE) passes(std::true_type{}, std::false_type{});
        else if (p.R) passes(std::false_type{}, std::true_type{});
        else passes(std::false_type{}, std::false_type{});
        if (p.gn_tot) {
            // column statistics of the fp16-rounded outputs.  The lanes of one chunk position hold eight column sums each; a REDUCE-SCATTER over three
            // lane bits (exchange 4, then 2, then 1 of them: 7 shuffles per statistic instead of the butterfly's 24 - 32) leaves lane (chunk position,
            // q) with column q's sum over the wave's 64 pixels (NT = 2: one more exchange over the fourth bit).  Fixed order: deterministic.  The
            // K-half-1 wave of the column group (pixels 64 .. 127) hands its sums to the K-half-0 wave through its own slice; one atomic add per
            // column and workgroup.
            constexpr int X0 = NT == 3 ? 8 : 4, X1 = 2 * X0, X2 = 4 * X0;
            const bool b0 = lane & X0, b1 = lane & X1, b2 = lane & X2;
            auto scatter = [&](const float (&v)[8]) __attribute__((always_inline)) {
                float k4[4], k2[2];
#pragma unroll
                for (int k = 0; k < 4; ++k) k4[k] = (b0 ? v[4 + k] : v[k]) + __shfl_xor(b0 ? v[k] : v[4 + k], X0);
#pragma unroll
                for (int k = 0; k < 2; ++k) k2[k] = (b1 ? k4[2 + k] : k4[k]) + __shfl_xor(b1 ? k4[k] : k4[2 + k], X1);
                float t = (b2 ? k2[1] : k2[0]) + __shfl_xor(b2 ? k2[0] : k2[1], X2);
                if (NT == 2) t += __shfl_xor(t, 32);
                return t;
            };
            const float s_own = scatter(gs), q2_own = scatter(gq);
            const int q_own = (b0 ? 4 : 0) + (b1 ? 2 : 0) + (b2 ? 1 : 0);
            const bool owner = NT == 3 ? lane_on : lane < 32;
            // both waves of the column group leave their column sums in their own slices; the K-half-0 wave adds per totals block (bc_gn_cg: one
            // partial per 10 channels for these networks, into the block's slot tin % 10 - bc_common.h)
            float* sc = reinterpret_cast<float*>(smem + wave * 16384);
            if (owner) {
                sc[(cpos * 8 + q_own) * 2] = s_own;
                sc[(cpos * 8 + q_own) * 2 + 1] = q2_own;
            }
            __syncthreads();
            if (kg == 0 && owner) {
                const float* sp = reinterpret_cast<const float*>(smem + (wave | 1) * 16384);
                const int n_lo = n0 + tile0 * 16;
                bc_gn_tot_add_slot(p.gn_tot + (size_t)b * g.n_out * BC_GN_TOT_WORDS, n_first + q_own, n_lo, n_lo + NT * 16, bc_gn_cg(g.n_out), tin, [&](int k) {
                    const int c = k - n_lo;
                    return make_float2(sc[c * 2] + sp[c * 2], sc[c * 2 + 1] + sp[c * 2 + 1]);
                });
            }
        }
    };
    if (three) body(IC<3>{}, IC<0>{});
    else body(IC<2>{}, IC<1>{});
    stamp(4);
    if (p.splitk > 1) {
        // split-K: this split's fp32 partial tile into its slab (24 rows x 20 eight-column chunks per sweep: 480 of the 512 threads)
        const int col8 = tid % 20, row0 = tid / 20;
        const int rpb = (int)g.div_rpb.d;
        if (tid < 480) {
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
    }
    stamp(5);
}

// out[frag stream] <- w[N][9 * Cin] (k = (ky * 3 + kx) * Cin + c, the layout of every other 3x3 path): one thread per 16-byte lane slot
__global__ void conv_wreg_pack_kernel(const h16* __restrict__ w, int N, int Cin, h16* __restrict__ out) {
    const long long total = (long long)N * 9 * Cin / 8;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int nchunks = Cin / 64;
    const long long per_ntile = (long long)20 * nchunks * 9 * 64;           // lane slots per 160-column block
    const int ntile = (int)(idx / per_ntile);
    long long rem = idx - (long long)ntile * per_ntile;
    // stream order inside a block: (group, K half) streams of nchunks * 9 * NT fragments, starting at (2 * tile0 + kg * NT) tile-streams
    const long long per_tile_stream = (long long)nchunks * 9 * 64;          // lane slots of one tile column and K half
    const int ts = (int)(rem / per_tile_stream);                             // 0..19 = 2 * tile0 + kg * NT + (position inside the wave stream)
    int grp, tile0, NT;
    if (ts < 6) { grp = 0; tile0 = 0; NT = 3; }
    else if (ts < 10) { grp = 1; tile0 = 3; NT = 2; }
    else if (ts < 14) { grp = 2; tile0 = 5; NT = 2; }
    else { grp = 3; tile0 = 7; NT = 3; }
    (void)grp;
    const long long in_grp = rem - (long long)2 * tile0 * per_tile_stream;  // lane slot inside the group's two wave streams
    const long long per_wave = per_tile_stream * NT;
    const int kg = (int)(in_grp / per_wave);
    long long s = in_grp - (long long)kg * per_wave;                         // lane slot inside the wave stream
    const int lane = (int)(s & 63);
    s >>= 6;                                                                 // fragment index = ((chunk * 3 + kx) * 3 + ky) * NT + t
    const int t = (int)(s % NT);
    s /= NT;
    const int ky = (int)(s % 3);
    s /= 3;
    const int kx = (int)(s % 3);
    const int chunk = (int)(s / 3);
    const int n = ntile * HBN + (tile0 + t) * 16 + (lane & 15);
    const long long k = (long long)(ky * 3 + kx) * Cin + chunk * 64 + kg * 32 + 8 * (lane >> 4);
    *reinterpret_cast<uint4*>(out + idx * 8) = *reinterpret_cast<const uint4*>(w + (long long)n * 9 * Cin + k);
}

}  // namespace

extern "C" int bc_conv_wreg_pack(const bc_half* w, int N, int Cin, bc_half* out, bc_stream stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    BC_CHECK_ARG(w && out && N > 0 && N % HBN == 0 && Cin > 0 && Cin % 64 == 0, "bc_conv_wreg_pack: N %% 160 == 0 and Cin %% 64 == 0 (N=%d Cin=%d)", N, Cin);
    BC_CHECK_ARG(((uintptr_t)w % 16 == 0) && ((uintptr_t)out % 16 == 0) && w != out, "bc_conv_wreg_pack: 16-byte aligned, out of place");
    const long long total = (long long)N * 9 * Cin / 8;
    hipLaunchKernelGGL(conv_wreg_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, reinterpret_cast<const h16*>(w), N, Cin,
                       reinterpret_cast<h16*>(out));
    BC_CHECK_LAUNCH();
    return 0;
}

int bc_conv_wreg_launch(GemmArgs& g, hipStream_t stream) {
    BcGemm& p = g.p;
    g.halo_tx = p.Wout / TW;
    g.halo_tpi = g.halo_tx * (p.Hout / TH);
    g.halo_nch = p.Cin / 64;
    g.halo_dbg = 0;                                  // (bit 0x100 = the round-4 in-prologue finalize: kept in the kernel for reference, not selectable)
    g.halo_stamps = nullptr;
    int sk = std::max(1, std::min(p.splitk, g.halo_nch));
    g.halo_cps = bc_ceil_div(g.halo_nch, sk);
    p.splitk = bc_ceil_div(g.halo_nch, g.halo_cps);
    BC_CHECK_ARG(g.halo_cps <= MAX_CH, "bc_gemm(wreg conv): %d channel chunks per split exceed %d (raise splitk)", g.halo_cps, MAX_CH);
    BC_CHECK_ARG(p.splitk == 1 || p.slab != nullptr, "bc_gemm(wreg conv): splitk=%d needs a slab", p.splitk);
    const int B = p.M / (p.Hout * p.Wout);
    BC_CHECK_ARG((long long)p.M * std::max(p.lda, p.lda2) < 2147483647LL, "bc_gemm(wreg conv): activation of %d pixels x stride %d exceeds 32-bit element offsets",
                 p.M, std::max(p.lda, p.lda2));
    dim3 grid(p.N / HBN, B * g.halo_tpi, p.splitk);
    g.wr_plane = make_fastdiv(grid.x * grid.y);
    g.wr_gx = make_fastdiv(grid.x);
    g.wr_gy = make_fastdiv(grid.y);
    g.wr_tpi = make_fastdiv((unsigned)g.halo_tpi);
    g.wr_tx = make_fastdiv((unsigned)g.halo_tx);
    g.wr_cpg = make_fastdiv((unsigned)(p.a_tot1 && p.a_groups > 0 ? p.Cin / p.a_groups : 1));
    {
        g.nband = (double)p.N * 9 > (double)p.M && grid.x >= 4;        // (threshold ratio 1: 0.25 and 4 measured worse)
    }
    // BC_WREG_STAMPS=1 (diagnostics; synchronises the stream after every launch): where the cycles of a three-tile wave and of a staging wave go
    static const bool want_stamps = getenv("BC_WREG_STAMPS") != nullptr;
    static unsigned long long* stamp_buf = nullptr;
    const size_t nwg_s = (size_t)grid.x * grid.y * grid.z;
    if (want_stamps && nwg_s <= 4096) {
        if (!stamp_buf) BC_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&stamp_buf), 4096 * 32 * sizeof(unsigned long long)));
        BC_CHECK_HIP(hipMemsetAsync(stamp_buf, 0, nwg_s * 32 * sizeof(unsigned long long), stream));
        g.halo_stamps = stamp_buf;
    }
    struct StampReport {
        hipStream_t stream; size_t n; unsigned long long* buf; const BcGemm& p; int cps;
        ~StampReport() {
            if (!buf) return;
            if (hipStreamSynchronize(stream) != hipSuccess) return;
            std::vector<unsigned long long> h(n * 32);
            if (hipMemcpy(h.data(), buf, n * 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return;
            for (int w = 0; w < 2; ++w) {
                double d[5] = {0, 0, 0, 0, 0};
                for (size_t i = 0; i < n; ++i)
                    for (int k = 0; k < 5; ++k) d[k] += (double)(h[(i * 2 + w) * 16 + k + 1] - h[(i * 2 + w) * 16 + k]);
                double e[3] = {0, 0, 0};                              // stamp 1 -> 6 (landing + barrier), 6 -> 7 (first in-place pass), 7 -> 2
                for (size_t i = 0; i < n; ++i) {
                    const unsigned long long* q = &h[(i * 2 + w) * 16];
                    e[0] += (double)(q[6] - q[1]);
                    e[1] += (double)(q[7] - q[6]);
                    e[2] += (double)(q[2] - q[7]);
                }
                double pr[6] = {0, 0, 0, 0, 0, 0};                   // prologue detail: stamps 8..13 relative to stamp 0 (0 where not taken)
                for (size_t i = 0; i < n; ++i)
                    for (int k = 0; k < 6; ++k) {
                        const unsigned long long* q = &h[(i * 2 + w) * 16];
                        if (q[8 + k]) pr[k] += (double)(q[8 + k] - q[0]);
                    }
                fprintf(stderr, "[wreg stamps] prologue detail (ticks since entry): inputs requested %.0f, ring group 0 requested %.0f, totals landed %.0f, "
                        "barrier 1 %.0f, barrier 2 %.0f, table written %.0f\n", pr[0] / n, pr[1] / n, pr[2] / n, pr[3] / n, pr[4] / n, pr[5] / n);
                fprintf(stderr, "[wreg stamps] M=%d N=%d Cin=%d sk=%d cps=%d wgs=%zu %s | avg ticks: setup+table %.0f, first halo %.0f (landing %.0f, pass %.0f, "
                        "next DMA + barrier %.0f), loop %.0f (%.0f per tap), k-half sum %.0f, stores %.0f\n", p.M, p.N, p.Cin, p.splitk, cps, n,
                        w ? "staging wave " : "3-tile wave  ", d[0] / n, d[1] / n, e[0] / n, e[1] / n, e[2] / n, d[2] / n, d[2] / n / (cps * 9), d[3] / n, d[4] / n);
            }
        }
    } report{stream, nwg_s, g.halo_stamps, p, g.halo_cps};
    static std::atomic<unsigned long long> set_a{0}, set_p{0}, set_f{0};
    if (p.a_tot1) {
        BC_CHECK_ARG(p.a_gamma && p.a_beta && p.a_groups > 0 && p.Cin % p.a_groups == 0 && (!p.A2 || p.a_tot2),
                     "bc_gemm(wreg conv): in-kernel GroupNorm finalize needs a_gamma, a_beta, a_groups | Cin and the partials of every source");
        const int cpg = p.Cin / p.a_groups;
        BC_CHECK_ARG(g.halo_cps * 64 + 2 * cpg <= FIN_MAX_CH && p.a_groups * 8 <= 1024,
                     "bc_gemm(wreg conv): channel span %d per workgroup too wide for the in-kernel GroupNorm finalize (max %d): use "
                     "bc_gn_finalize + a_affine or raise splitk", g.halo_cps * 64 + 2 * cpg, FIN_MAX_CH);
        BC_CHECK_HIP(bc_set_max_lds(set_f, reinterpret_cast<const void*>(&conv_wreg_kernel<2>), LDS_TOTAL_FIN));
        const int lds_fin = std::max(LDS_TOTAL, OFF_FIN + (g.halo_cps * 64 + 2 * cpg) * 16 + 1024);    // (scratch for this span only)
        hipLaunchKernelGGL((conv_wreg_kernel<2>), grid, dim3(512), lds_fin, stream, g);
    } else if (p.a_affine) {
        BC_CHECK_HIP(bc_set_max_lds(set_a, reinterpret_cast<const void*>(&conv_wreg_kernel<1>), LDS_TOTAL));
        hipLaunchKernelGGL((conv_wreg_kernel<1>), grid, dim3(512), LDS_TOTAL, stream, g);
    } else {
        BC_CHECK_HIP(bc_set_max_lds(set_p, reinterpret_cast<const void*>(&conv_wreg_kernel<0>), LDS_TOTAL));
        hipLaunchKernelGGL((conv_wreg_kernel<0>), grid, dim3(512), LDS_TOTAL, stream, g);
    }
    BC_CHECK_LAUNCH();
    return 0;
}
