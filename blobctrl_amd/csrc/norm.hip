// GroupNorm(+SiLU) over NHWC (optionally over a channel-concat of two tensors) and LayerNorm, gfx950.
// HBM-bound kernels: 16-byte (8 x fp16) loads/stores per lane, fp32 statistics, 64-wide wavefront reductions.
//
// Reference call sites replaced: F.group_norm + SiLU at D/models/resnet.py:327-328,351-363, transformer_2d.py:481,
// unet_2d_condition.py:1341-1343; torch.cat([h, skip], 1) at unet_2d_blocks.py:2559,2719 (folded into the two-source
// read); F.layer_norm at D/models/attention.py:447,491,517 and in transformers' Dinov2Layer.
#include "bc_common.h"

namespace {

constexpr int GN_PIX_PER_SLAB = 128;    // stand-alone statistics pass (same slab height as the smallest fused GEMM tiles)
constexpr int GN_MAX_GROUPS = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Per-CHANNEL (sum, sumsq) over the GN_PIX_PER_SLAB pixels of a slab, ADDED to the totals tot[b][c][BC_GN_TOT_WORDS] (integer atomics:
// order-independent, so GroupNorm - and everything downstream - stays bit-reproducible; bc_common.h).  Four waves split the slab's
// pixels; lane l of a wave owns 8-channel chunk(s) l, l + 64, ... (adjacent lanes read adjacent 16-byte chunks: coalesced rows); the
// four wave partials are combined through LDS in wave order.
// (Only used for tensors whose producer could not emit the statistics itself - see BcGemm.gn_tot.)
__global__ __launch_bounds__(256) void gn_stats_kernel(const h16* __restrict__ x, int C, int HW, unsigned long long* __restrict__ tot) {
    __shared__ float red[4][64][16];
    const int b = blockIdx.y, slab = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunk = C / 8;
    const int p0 = slab * GN_PIX_PER_SLAB;
    const int np = min(GN_PIX_PER_SLAB, HW - p0);
    const int per_wave = (np + 3) / 4;
    const int pw0 = min(np, wave * per_wave), pw1 = min(np, pw0 + per_wave);
    unsigned long long* dst = tot + (size_t)b * C * BC_GN_TOT_WORDS;
    for (int ch0 = 0; ch0 < nchunk; ch0 += 64) {
        const int ch = ch0 + lane;
        float s[8], q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
        if (ch < nchunk) {
            const h16* src = x + ((size_t)b * HW + p0) * C + ch * 8;
#pragma unroll 8
            for (int pl = pw0; pl < pw1; ++pl) {
                uint4 raw = bc_ld16(src + (size_t)pl * C);
                const h16* v = reinterpret_cast<const h16*>(&raw);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float f = (float)v[j];
                    s[j] += f;
                    q[j] += f * f;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { red[wave][lane][2 * j] = s[j]; red[wave][lane][2 * j + 1] = q[j]; }
        __syncthreads();
        if (wave == 0 && ch < nchunk) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float ss = 0.f, qq = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) { ss += red[w][lane][2 * j]; qq += red[w][lane][2 * j + 1]; }
                bc_gn_tot_add(dst + (size_t)(ch * 8 + j) * BC_GN_TOT_WORDS, ss, qq);
            }
        }
        __syncthreads();
    }
}

// ab[b][c] = (rstd*gamma[c], beta[c] - mean*rstd*gamma[c]) for the channel-concat of two sources, each with its own totals
// tot_i[B][C_i][BC_GN_TOT_WORDS].  grid (G, B), one wave per group: every lane reads the totals of its channels of the group (48 bytes
// each), fp64 butterfly => deterministic.
__global__ __launch_bounds__(64) void gn_finalize_kernel(const unsigned long long* __restrict__ tot1, int C1,
                                                           const unsigned long long* __restrict__ tot2, int C2, int HW,
                                                           int G, float eps, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ ab) {
    const int b = blockIdx.y;
    const int C = C1 + C2;
    const int cpg = C / G;
    const int lane = threadIdx.x;
    const int c_lo = blockIdx.x * cpg, c_hi = c_lo + cpg;
    double s = 0.0, q = 0.0;
    for (int c = c_lo + lane; c < c_hi; c += 64) {
        const unsigned long long* t = c < C1 ? tot1 + ((size_t)b * C1 + c) * BC_GN_TOT_WORDS : tot2 + ((size_t)b * C2 + (c - C1)) * BC_GN_TOT_WORDS;
        double cs, cq;
        bc_gn_tot_read(t, cs, cq);
        s += cs;
        q += cq;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o);
        q += __shfl_xor(q, o);
    }
    const double n = (double)HW * cpg;
    const double mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float meanf = (float)mean, rstd = (float)(1.0 / sqrt(var + (double)eps));
    for (int c = c_lo + lane; c < c_hi; c += 64) {
        const float a = rstd * gamma[c];
        ab[((size_t)b * C + c) * 2] = a;
        ab[((size_t)b * C + c) * 2 + 1] = beta[c] - meanf * a;
    }
}


// y = silu?(x * a + b) over the concat of two sources; grid (chunks of one image, batch), 32-bit index maths only.
__global__ __launch_bounds__(256) void gn_apply_kernel(const h16* __restrict__ x1, int C1, const h16* __restrict__ x2,
                                                         int C2, int HW, int nchunk, unsigned div_mul, unsigned div_shift,
                                                         const float* __restrict__ ab, int silu, h16* __restrict__ y) {
    const int C = C1 + C2;
    const int b = blockIdx.y;
    const unsigned total = (unsigned)HW * nchunk;
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const unsigned pl = (unsigned)(((unsigned long long)idx * div_mul) >> div_shift);     // idx / nchunk
        const int c = (int)(idx - pl * nchunk) * 8;
        const size_t pix = (size_t)b * HW + pl;
        uint4 raw = (c < C1) ? bc_ld16(x1 + pix * C1 + c) : bc_ld16(x2 + pix * C2 + (c - C1));
        const h16* v = reinterpret_cast<const h16*>(&raw);
        const float4* abp = reinterpret_cast<const float4*>(ab + ((size_t)b * C + c) * 2);
        uint4 outraw;
        h16* o = reinterpret_cast<h16*>(&outraw);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float4 t = abp[j];                       // (a0, b0, a1, b1)
            float r0 = (float)v[2 * j] * t.x + t.y;
            float r1 = (float)v[2 * j + 1] * t.z + t.w;
            if (silu) { r0 = bc_silu_f(r0); r1 = bc_silu_f(r1); }
            o[2 * j] = (h16)r0;
            o[2 * j + 1] = (h16)r1;
        }
        bc_st16(y + pix * C + c, outraw);
    }
}

// Fused finalize + apply: one launch per GroupNorm.  A workgroup owns a 64-channel range x a pixel range of one image; its
// prologue reads the statistics totals (bc_common.h: six words per channel) of the groups that overlap its channel range, one thread
// per channel, combines the channels of a group through LDS (one thread per group, fixed order, fp64) into (a, b) pairs, then
// streams its pixels: y = silu?(a*x + b), 16-byte accesses.
__global__ __launch_bounds__(256) void gn_apply_fused_kernel(const unsigned long long* __restrict__ tot1, int C1,
                                                               const unsigned long long* __restrict__ tot2, int C2,
                                                               const h16* __restrict__ x1, const h16* __restrict__ x2, int HW,
                                                               int G, float eps, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, int silu, int pix_per_block,
                                                               h16* __restrict__ y) {
    __shared__ float ab_s[64 * 2];
    __shared__ double sq_s[256 * 2];
    const int b = blockIdx.z;
    const int C = C1 + C2;
    const int cpg = C / G;
    const int c0 = blockIdx.x * 64, c1 = min(c0 + 64, C);
    const int g_lo = c0 / cpg, g_hi = (c1 - 1) / cpg;
    const int cA = g_lo * cpg, nch = (g_hi + 1) * cpg - cA;          // channels whose statistics this workgroup needs (<= 256)
    if ((int)threadIdx.x < nch) {
        const int c = cA + threadIdx.x;
        const unsigned long long* t = c < C1 ? tot1 + ((size_t)b * C1 + c) * BC_GN_TOT_WORDS : tot2 + ((size_t)b * C2 + (c - C1)) * BC_GN_TOT_WORDS;
        double s, q;
        bc_gn_tot_read(t, s, q);
        sq_s[threadIdx.x * 2] = s;
        sq_s[threadIdx.x * 2 + 1] = q;
    }
    __syncthreads();
    if ((int)threadIdx.x <= g_hi - g_lo) {                            // one thread per group: fixed-order combine
        const int g = g_lo + threadIdx.x;
        const int cl = g * cpg - cA;
        double s = 0.0, q = 0.0;
        for (int j = 0; j < cpg; ++j) {
            s += sq_s[(cl + j) * 2];
            q += sq_s[(cl + j) * 2 + 1];
        }
        const double n = (double)HW * cpg;
        const double mean = s / n;
        double var = q / n - mean * mean;
        if (var < 0.0) var = 0.0;
        const float meanf = (float)mean, rstd = (float)(1.0 / sqrt(var + (double)eps));
        for (int c = max(g * cpg, c0); c < min((g + 1) * cpg, c1); ++c) {
            const float a = rstd * gamma[c];
            ab_s[(c - c0) * 2] = a;
            ab_s[(c - c0) * 2 + 1] = beta[c] - meanf * a;
        }
    }
    __syncthreads();
    const int chunk = threadIdx.x & 7;                   // 8 channels
    const int c = c0 + chunk * 8;
    if (c >= c1) return;
    float a8[8], b8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { a8[j] = ab_s[(chunk * 8 + j) * 2]; b8[j] = ab_s[(chunk * 8 + j) * 2 + 1]; }
    const int p_begin = blockIdx.y * pix_per_block, p_end = min(HW, p_begin + pix_per_block);
    const bool first = c < C1;
    const h16* src = first ? x1 + (size_t)b * HW * C1 + c : x2 + (size_t)b * HW * C2 + (c - C1);
    const int sstride = first ? C1 : C2;
    h16* dst = y + (size_t)b * HW * C + c;
#pragma unroll 4
    for (int p = p_begin + (threadIdx.x >> 3); p < p_end; p += 32) {
        const uint4 raw = bc_ld16(src + (size_t)p * sstride);
        const h16* v = reinterpret_cast<const h16*>(&raw);
        uint4 outraw;
        h16* o = reinterpret_cast<h16*>(&outraw);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float r = (float)v[j] * a8[j] + b8[j];
            if (silu) r = bc_silu_f(r);
            o[j] = (h16)r;
        }
        bc_st16(dst + (size_t)p * C, outraw);
    }
}

// LayerNorm: one wave per row; the row (C <= 64*8*MAXC elements) lives in registers, two-pass statistics.
template <int MAXC>
__global__ __launch_bounds__(256) void layernorm_kernel(const h16* __restrict__ x, int rows, int C, int ldx,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, h16* __restrict__ y, int ldy) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = C / 8;
    float v[MAXC][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        int ch = lane + i * 64;
        if (ch < nchunk) {
            uint4 raw = bc_ld16(x + (size_t)row * ldx + ch * 8);
            const h16* h = reinterpret_cast<const h16*>(&raw);
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[i][j] = (float)h[j]; s += v[i][j]; }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        int ch = lane + i * 64;
        if (ch < nchunk) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { float d = v[i][j] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        int ch = lane + i * 64;
        if (ch < nchunk) {
            uint4 outraw;
            h16* o = reinterpret_cast<h16*>(&outraw);
            const float4* gp = reinterpret_cast<const float4*>(gamma + ch * 8);
            const float4* bp = reinterpret_cast<const float4*>(beta + ch * 8);
            float4 g0 = gp[0], g1 = gp[1], b0 = bp[0], b1 = bp[1];
            float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (h16)((v[i][j] - mean) * rstd * gg[j] + bb[j]);
            bc_st16(y + (size_t)row * ldy + ch * 8, outraw);
        }
    }
}

// Row softmax, one wave per row, fp32 maths, in place.  cols <= 64*8*MAXC.
template <int MAXC>
__global__ __launch_bounds__(256) void softmax_rows_kernel(h16* __restrict__ x, int rows, int cols, int ld) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = cols / 8;
    float v[MAXC][8];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int ch = lane + i * 64;
        if (ch < nchunk) {
            uint4 raw = bc_ld16(x + (size_t)row * ld + ch * 8);
            const h16* hh = reinterpret_cast<const h16*>(&raw);
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[i][j] = (float)hh[j]; mx = fmaxf(mx, v[i][j]); }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = -INFINITY;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[i][j] = __expf(v[i][j] - mx); sum += v[i][j]; }
    const float inv = 1.0f / wave_sum(sum);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int ch = lane + i * 64;
        if (ch < nchunk) {
            uint4 outraw;
            h16* o = reinterpret_cast<h16*>(&outraw);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (h16)(v[i][j] * inv);
            bc_st16(x + (size_t)row * ld + ch * 8, outraw);
        }
    }
}

__global__ __launch_bounds__(256) void zero_kernel(uint4* __restrict__ p, long long n16) {
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) p[i] = z;
}

// bc_dup_halves: up to six buffers [2][bytes_k]; the first half of each is copied over its second half (16-byte accesses)
struct DupArgs { uint4* p[6]; long long n16[6]; };
__global__ __launch_bounds__(256) void dup_halves_kernel(const DupArgs a) {
    uint4* __restrict__ p = a.p[blockIdx.y];
    const long long n = a.n16[blockIdx.y];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[n + i] = p[i];
}

}  // namespace

extern "C" int bc_dup_halves(void* p0, long long b0, void* p1, long long b1, void* p2, long long b2, void* p3, long long b3, void* p4, long long b4,
                             void* p5, long long b5, bc_stream stream_) {
    void* ps[6] = {p0, p1, p2, p3, p4, p5};
    const long long bs[6] = {b0, b1, b2, b3, b4, b5};
    DupArgs a;
    int n = 0;
    long long most = 0;
    for (int k = 0; k < 6; ++k) {
        if (!ps[k] || bs[k] <= 0) continue;
        BC_CHECK_ARG(bs[k] % 16 == 0 && ((uintptr_t)ps[k] % 16) == 0, "bc_dup_halves: buffer %d needs a 16-byte aligned pointer and half size", k);
        a.p[n] = reinterpret_cast<uint4*>(ps[k]);
        a.n16[n] = bs[k] / 16;
        most = std::max(most, a.n16[n]);
        ++n;
    }
    BC_CHECK_ARG(n > 0, "bc_dup_halves: nothing to copy");
    const int blocks = (int)std::min<long long>((most + 255) / 256, 1024);
    hipLaunchKernelGGL(dup_halves_kernel, dim3(blocks, n), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), a);
    BC_CHECK_LAUNCH();
    return 0;
}

extern "C" int bc_gn_stats(const bc_half* x, int C, int B, int HW, unsigned long long* tot, bc_stream stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    BC_CHECK_ARG(x && tot && B > 0 && HW > 0 && C > 0 && C % 8 == 0, "bc_gn_stats: bad args (C %% 8 == 0)");
    hipLaunchKernelGGL(gn_stats_kernel, dim3(bc_ceil_div(HW, GN_PIX_PER_SLAB), B), dim3(256), 0, stream, reinterpret_cast<const h16*>(x), C, HW, tot);
    BC_CHECK_LAUNCH();
    return 0;
}

extern "C" int bc_gn_finalize(const unsigned long long* tot1, int C1, const unsigned long long* tot2, int C2, int B, int HW,
                              int G, float eps, const float* gamma, const float* beta, float* ab, bc_stream stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!tot2) C2 = 0;
    int C = C1 + C2;
    BC_CHECK_ARG(tot1 && gamma && beta && ab && G > 0 && G <= GN_MAX_GROUPS && C % G == 0,
                 "bc_gn_finalize: bad args (groups <= %d, C %% G == 0)", GN_MAX_GROUPS);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(G, B), dim3(64), 0, stream, tot1, C1, tot2, C2, HW, G, eps, gamma, beta, ab);
    BC_CHECK_LAUNCH();
    return 0;
}

extern "C" int bc_memset_zero(void* ptr, long long bytes, bc_stream stream_) {
    // an ordinary kernel (a kernel node when captured), 16-byte stores: the statistics tables are 16-byte aligned multiples of 16 bytes
    BC_CHECK_ARG(ptr && bytes > 0 && bytes % 16 == 0 && ((uintptr_t)ptr % 16) == 0, "bc_memset_zero: needs a 16-byte aligned pointer and size");
    const long long n16 = bytes / 16;
    const int blocks = (int)std::min<long long>((n16 + 255) / 256, 2048);
    hipLaunchKernelGGL(zero_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), reinterpret_cast<uint4*>(ptr), n16);
    BC_CHECK_LAUNCH();
    return 0;
}

extern "C" int bc_gn_apply(const bc_half* x1, int C1, const bc_half* x2, int C2, int B, int HW, const float* ab,
                           int silu, bc_half* y, bc_stream stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!x2) C2 = 0;
    BC_CHECK_ARG(x1 && ab && y && C1 % 8 == 0 && C2 % 8 == 0, "bc_gn_apply: bad args");
    int nchunk = (C1 + C2) / 8;
    long long per_img = (long long)HW * nchunk;
    BC_CHECK_ARG(per_img < (1ll << 31), "bc_gn_apply: image too large");
    unsigned l = 0;
    while ((1ull << l) < (unsigned)nchunk) ++l;
    unsigned long long mul = ((1ull << (31 + l)) + nchunk - 1) / nchunk;
    int blocks = (int)std::min<long long>((per_img + 255) / 256, 2048);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(blocks, B), dim3(256), 0, stream, reinterpret_cast<const h16*>(x1), C1,
                       reinterpret_cast<const h16*>(x2), C2, HW, nchunk, (unsigned)mul, 31 + l, ab, silu,
                       reinterpret_cast<h16*>(y));
    BC_CHECK_LAUNCH();
    return 0;
}

extern "C" int bc_gn_apply_fused(const unsigned long long* tot1, int C1, const unsigned long long* tot2, int C2,
                                 const bc_half* x1, const bc_half* x2, int B, int HW, int G, float eps, const float* gamma,
                                 const float* beta, int silu, bc_half* y, bc_stream stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!tot2 || !x2) { C2 = 0; tot2 = nullptr; x2 = nullptr; }
    const int C = C1 + C2;
    BC_CHECK_ARG(tot1 && x1 && y && gamma && beta && G > 0 && C % G == 0 && C1 % 8 == 0 && C2 % 8 == 0, "bc_gn_apply_fused: bad args");
    BC_CHECK_ARG(C / G <= 96, "bc_gn_apply_fused: %d channels per group (> 96): use bc_gn_finalize + bc_gn_apply", C / G);
    const int cblocks = bc_ceil_div(C, 64);
    // ~512 workgroups in total, at least 32 pixels each
    int pblocks = std::max(1, std::min(bc_ceil_div(HW, 32), bc_ceil_div(512, cblocks * B)));
    int ppb = bc_ceil_div(bc_ceil_div(HW, pblocks), 32) * 32;
    pblocks = bc_ceil_div(HW, ppb);
    hipLaunchKernelGGL(gn_apply_fused_kernel, dim3(cblocks, pblocks, B), dim3(256), 0, stream, tot1, C1, tot2, C2,
                       reinterpret_cast<const h16*>(x1), reinterpret_cast<const h16*>(x2), HW, G, eps, gamma, beta, silu,
                       ppb, reinterpret_cast<h16*>(y));
    BC_CHECK_LAUNCH();
    return 0;
}

extern "C" int bc_layernorm(const bc_half* x, int rows, int C, int ldx, const float* gamma, const float* beta,
                            float eps, bc_half* y, int ldy, bc_stream stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    BC_CHECK_ARG(x && y && gamma && beta && rows > 0, "bc_layernorm: bad args");
    BC_CHECK_ARG(C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && C <= 64 * 8 * 4, "bc_layernorm: C=%d unsupported (C%%8==0, C<=2048)", C);
    dim3 grid(bc_ceil_div(rows, 4)), block(256);
    const h16* xi = reinterpret_cast<const h16*>(x);
    h16* yo = reinterpret_cast<h16*>(y);
    if (C <= 512) hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, stream, xi, rows, C, ldx, gamma, beta, eps, yo, ldy);
    else if (C <= 1024) hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, stream, xi, rows, C, ldx, gamma, beta, eps, yo, ldy);
    else hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, stream, xi, rows, C, ldx, gamma, beta, eps, yo, ldy);
    BC_CHECK_LAUNCH();
    return 0;
}

extern "C" int bc_softmax_rows(bc_half* x, int rows, int cols, int ld, bc_stream stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    BC_CHECK_ARG(x && rows > 0 && cols > 0 && cols % 8 == 0 && ld % 8 == 0 && ld >= cols && cols <= 64 * 8 * 18,
                 "bc_softmax_rows: cols=%d must be a multiple of 8 and <= 9216", cols);
    dim3 grid(bc_ceil_div(rows, 4)), block(256);
    h16* xi = reinterpret_cast<h16*>(x);
    if (cols <= 512) hipLaunchKernelGGL(softmax_rows_kernel<1>, grid, block, 0, stream, xi, rows, cols, ld);
    else if (cols <= 2048) hipLaunchKernelGGL(softmax_rows_kernel<4>, grid, block, 0, stream, xi, rows, cols, ld);
    else if (cols <= 4096) hipLaunchKernelGGL(softmax_rows_kernel<8>, grid, block, 0, stream, xi, rows, cols, ld);
    else hipLaunchKernelGGL(softmax_rows_kernel<18>, grid, block, 0, stream, xi, rows, cols, ld);
    BC_CHECK_LAUNCH();
    return 0;
}
