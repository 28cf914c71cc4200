// GroupNorm(+SiLU) over NHWC (optionally over a channel-concat of two tensors) and LayerNorm, gfx950.
// HBM-bound kernels: 16-byte (8 x fp16) loads/stores per lane, fp32 statistics, 64-wide wavefront reductions.
//
// Reference call sites replaced: F.group_norm + SiLU at D/models/resnet.py:327-328,351-363, transformer_2d.py:481,
// unet_2d_condition.py:1341-1343; torch.cat([h, skip], 1) at unet_2d_blocks.py:2559,2719 (folded into the two-source
// read); F.layer_norm at D/models/attention.py:447,491,517 and in transformers' Dinov2Layer.
#include "bc_common.h"

namespace {

constexpr int GN_PIX_PER_SLAB = 16;     // stand-alone statistics pass: small slabs => enough workgroups for 256 CUs
constexpr int GN_MAX_GROUPS = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// part[b][slab][c][2] = per-CHANNEL (sum, sumsq) over the GN_PIX_PER_SLAB pixels of a slab.  No atomics: every
// partial is produced by exactly one thread in a fixed order, so GroupNorm (and everything downstream) is bit-reproducible.
// Lane l owns 8-channel chunk(s) l, l + blockDim, ...; adjacent lanes read adjacent 16-byte chunks (coalesced rows).
// (Only used for tensors whose producer could not emit the partials itself - see BcGemm.gn_part.)
__global__ __launch_bounds__(256) void gn_stats_kernel(const h16* __restrict__ x, int C, int HW, float* __restrict__ part,
                                                         int nslab) {
    const int b = blockIdx.y, slab = blockIdx.x;
    const int nchunk = C / 8;
    const int p0 = slab * GN_PIX_PER_SLAB;
    const int np = min(GN_PIX_PER_SLAB, HW - p0);
    float* dst = part + ((size_t)b * nslab + slab) * C * 2;
    for (int ch = threadIdx.x; ch < nchunk; ch += blockDim.x) {
        const int c = ch * 8;
        const h16* src = x + ((size_t)b * HW + p0) * C + c;
        float s[8], q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
#pragma unroll 8
        for (int pl = 0; pl < np; ++pl) {
            uint4 raw = bc_ld16(src + (size_t)pl * C);
            const h16* v = reinterpret_cast<const h16*>(&raw);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float f = (float)v[j];
                s[j] += f;
                q[j] += f * f;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            dst[(c + j) * 2] = s[j];
            dst[(c + j) * 2 + 1] = q[j];
        }
    }
}

// ab[b][c] = (rstd*gamma[c], beta[c] - mean*rstd*gamma[c]) for the channel-concat of two sources, each with its own
// per-channel partial buffer part_i[B][nslab_i][C_i][2].  grid (ceil(G/4), B), one wave per group: its lanes sweep the
// flattened (channel-in-group, slab) items of both sources (all loads independent -> one memory latency, not cpg of them),
// fp64 accumulation in a fixed order => deterministic.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ part1, int nslab1, int C1,
                                                            const float* __restrict__ part2, int nslab2, int C2, int HW,
                                                            int G, float eps, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ ab) {
    const int b = blockIdx.y;
    const int C = C1 + C2;
    const int cpg = C / G;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = blockIdx.x * 4 + wave;
    if (g >= G) return;
    const int c_lo = g * cpg, c_hi = c_lo + cpg;
    const int n1 = max(0, min(c_hi, C1) - c_lo);               // channels of this group living in source 1
    const int n2 = cpg - n1;
    double s = 0.0, q = 0.0;
    {
        const float* base = part1 + ((size_t)b * nslab1 * C1 + c_lo) * 2;
        const int items = n1 * nslab1;
        for (int it = lane; it < items; it += 64) {
            const int sl = it / n1, cj = it - sl * n1;
            const float2 v = *reinterpret_cast<const float2*>(base + ((size_t)sl * C1 + cj) * 2);
            s += v.x;
            q += v.y;
        }
    }
    if (n2 > 0) {
        const int c2_lo = max(c_lo, C1) - C1;
        const float* base = part2 + ((size_t)b * nslab2 * C2 + c2_lo) * 2;
        const int items = n2 * nslab2;
        for (int it = lane; it < items; it += 64) {
            const int sl = it / n2, cj = it - sl * n2;
            const float2 v = *reinterpret_cast<const float2*>(base + ((size_t)sl * C2 + cj) * 2);
            s += v.x;
            q += v.y;
        }
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

}  // namespace

extern "C" int bc_gn_stats(const bc_half* x, int C, int B, int HW, float* part, int nslab, bc_stream stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    BC_CHECK_ARG(x && part && B > 0 && HW > 0 && C > 0 && C % 8 == 0, "bc_gn_stats: bad args (C %% 8 == 0)");
    BC_CHECK_ARG(nslab == bc_ceil_div(HW, GN_PIX_PER_SLAB), "bc_gn_stats: nslab must be ceil(HW/%d)", GN_PIX_PER_SLAB);
    int threads = std::min(256, ((C / 8 + 63) / 64) * 64);
    hipLaunchKernelGGL(gn_stats_kernel, dim3(nslab, B), dim3(threads), 0, stream, reinterpret_cast<const h16*>(x), C, HW, part,
                       nslab);
    BC_CHECK_LAUNCH();
    return 0;
}

extern "C" int bc_gn_finalize(const float* part1, int nslab1, int C1, const float* part2, int nslab2, int C2, int B, int HW,
                              int G, float eps, const float* gamma, const float* beta, float* ab, bc_stream stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!part2) { C2 = 0; nslab2 = 0; }
    int C = C1 + C2;
    BC_CHECK_ARG(part1 && gamma && beta && ab && G > 0 && G <= GN_MAX_GROUPS && C % G == 0 && nslab1 > 0,
                 "bc_gn_finalize: bad args (groups <= %d, C %% G == 0)", GN_MAX_GROUPS);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(bc_ceil_div(G, 4), B), dim3(256), 0, stream, part1, nslab1, C1, part2, nslab2, C2, HW, G, eps,
                       gamma, beta, ab);
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
