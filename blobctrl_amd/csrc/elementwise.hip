// Blob maths and loop glue kernels (all HBM/latency-trivial): Gaussian-blob splat rasteriser, input assembly,
// timestep embedding, CFG + scheduler step, layout conversion at the nn.Module boundary, DINOv2 embedding glue.
#include "bc_common.h"

namespace {

// blobctrl/utils/utils.py:145-194 for one blob per image, fp64 like the reference (numpy float64 -> torch float64).
//   delta = (grid - mu*(W,H)) / (W,H) ; m = delta^T cov^-1 delta ; s = min(1, 2*sigmoid(-m)) ; s = 1e-6f if size < 0.5
//   out[0] = (1 - s) * 1 (background after alpha compositing), out[1] = s.
struct SplatParams { double v[16 * 8]; };
__global__ void splat_kernel(const SplatParams prm, int h, int w, double* __restrict__ out) {
    const int n = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= h * w) return;
    const double* p = prm.v + n * 8;
    const double xs = p[0], ys = p[1], a = p[2], b = p[3], c = p[4], d = p[5], size = p[6];
    const int gy = idx / w, gx = idx - gy * w;
    const double dx = ((double)gx - xs * (double)w) / (double)w;      // ut:151-153
    const double dy = ((double)gy - ys * (double)h) / (double)h;
    // solve [[a b][c d]] z = delta  (ut:156, torch.linalg.solve = LU with partial pivoting)
    double z0, z1;
    if (fabs(a) >= fabs(c)) {
        const double f = c / a;
        const double u = d - f * b;
        z1 = (dy - f * dx) / u;
        z0 = (dx - b * z1) / a;
    } else {
        const double f = a / c;
        const double u = b - f * d;
        z1 = (dx - f * dy) / u;
        z0 = (dy - d * z1) / c;
    }
    const double m = dx * z0 + dy * z1;
    double s = 1.0 / (1.0 + exp(m));                                  // sigmoid(-m)  ut:162
    s = fmin(2.0 * s, 1.0);                                           // ut:163
    if (size < 0.5) s = (double)1e-6f;                                // ut:165-172 (float32 constant in the reference)
    double* o = out + (size_t)n * 2 * h * w;
    o[idx] = (1.0 - s);                                               // ut:179-181 alpha composite with bg score 1
    o[(size_t)h * w + idx] = s;
}

// pipeline_blobnet.py:724-739 + :706-721.  X[b][y][x][c], x in [0, 2w): left = clean image latents, right = noisy latents.
__global__ void assemble_kernel(const float* __restrict__ latents, int Blat, const float* __restrict__ img_lat,
                                const float* __restrict__ score, const float* __restrict__ feat, int Bimg, int F, int Bout,
                                int h, int w, int Cpad, int dup_score, h16* __restrict__ X) {
    const long long total = (long long)Bout * h * 2 * w * (Cpad / 8);
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int nch = Cpad / 8;
        int ch = (int)(idx % nch);
        long long pix = idx / nch;
        int x = (int)(pix % (2 * w));
        int y = (int)((pix / (2 * w)) % h);
        int b = (int)(pix / ((long long)2 * w * h));
        const bool right = x >= w;
        const int xs = right ? x - w : x;
        const int bi = b % Bimg;
        const float sc = score[((size_t)bi * h + y) * w + xs];
        uint4 raw;
        h16* o = reinterpret_cast<h16*>(&raw);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int c = ch * 8 + j;
            float v = 0.f;
            if (c < 4) {
                v = right ? latents[(((size_t)(b % Blat) * 4 + c) * h + y) * w + xs] : img_lat[(((size_t)bi * 4 + c) * h + y) * w + xs];
            } else if (c == 4) {
                v = sc;
            } else if (c < 5 + F) {
                v = sc * feat[(size_t)bi * F + c - 5];
            } else if (dup_score && c == 5) {
                v = sc;          // rank-1 collapsed feature channels: one extra copy of the score (see engine.py)
            }
            o[j] = (h16)v;
        }
        bc_st16(X + (size_t)pix * Cpad + ch * 8, raw);
    }
}

// The same input (8 channels: 4 latents, score, [score], 0, 0) written as the 3x3 im2col operand of conv_in: row = canvas pixel,
// k = tap * 8 + channel for the nine taps (zero outside the h x 2w canvas = the convolution's padding), zero-filled up to 128, so that
// conv_in (K = 72: outside the LDS-DMA GEMM's K % 64 == 0 fast path, 41 us per launch on the register-staged kernel) runs as a dense
// K = 128 GEMM.  One thread per (pixel, 16-byte chunk).
__global__ void assemble_im2col_kernel(const float* __restrict__ latents, int Blat, const float* __restrict__ img_lat,
                                       const float* __restrict__ score, int Bimg, int Bout, int h, int w, int dup_score,
                                       h16* __restrict__ X) {
    const long long total = (long long)Bout * h * 2 * w * 16;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int chunk = (int)(idx & 15);
        const long long pix = idx >> 4;
        const int x = (int)(pix % (2 * w));
        const int y = (int)((pix / (2 * w)) % h);
        const int b = (int)(pix / ((long long)2 * w * h));
        uint4 raw = make_uint4(0u, 0u, 0u, 0u);
        if (chunk < 9) {
            const int yy = y + chunk / 3 - 1, xx = x + chunk % 3 - 1;
            if ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)(2 * w)) {
                const bool right = xx >= w;
                const int xs = right ? xx - w : xx;
                const int bi = b % Bimg;
                const float sc = score[((size_t)bi * h + yy) * w + xs];
                h16* o = reinterpret_cast<h16*>(&raw);
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    o[c] = (h16)(right ? latents[(((size_t)(b % Blat) * 4 + c) * h + yy) * w + xs]
                                       : img_lat[(((size_t)bi * 4 + c) * h + yy) * w + xs]);
                o[4] = (h16)sc;
                o[5] = dup_score ? (h16)sc : (h16)0.f;
            }
        }
        bc_st16(X + (size_t)pix * 128 + chunk * 8, raw);
    }
}

// embeddings.py:27-78: [cos(t*f_k) | sin(t*f_k)], f_k = exp(-ln(10000) * k / half)
// rows_per_step > 0: row r belongs to step r / rows_per_step of the table (all steps of an edit at once)
__global__ void temb_kernel(const float* __restrict__ t_table, const int* __restrict__ t_idx, float t_value, int rows,
                            int dim, int rows_per_step, h16* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * dim) return;
    const int r = i / dim, c = i - r * dim;
    const int halfd = dim / 2;
    const float t = rows_per_step > 0 ? t_table[r / rows_per_step] : (t_table ? t_table[t_idx ? *t_idx : 0] : t_value);
    const int k = c < halfd ? c : c - halfd;
    const float freq = expf(-9.210340371976184f * (float)k / (float)halfd);
    const float a = t * freq;
    out[i] = (h16)(c < halfd ? cosf(a) : sinf(a));
}

__global__ void silu_kernel(const h16* __restrict__ x, h16* __restrict__ y, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        y[i] = (h16)bc_silu_f((float)x[i]);
}

// coef row layout (blobctrl_amd/schedulers.py): see bc_cfg_scheduler_step in the header.
//  [0] inv_alpha_t  [1] sigma_over_alpha      x0 = x*c0 - eps*c1
//  [2] use_corrector
//  [3] cc_x (last_sample) [4] cc_m0 (prev x0) [5] cc_m1 (prev-prev x0) [6] cc_mt (this x0)      x_c = sum
//  [7] cp_x (x_c)         [8] cp_m0 (this x0) [9] cp_m1 (prev x0)  [10] cp_eps (guided eps)     x_next = sum
//  [11] guidance scale (used when the launch argument is negative: lets a captured graph follow per-call values)
__global__ void cfg_step_kernel(const float* __restrict__ eps, float* __restrict__ latents, const float* __restrict__ coef,
                                int* __restrict__ step_idx, float* __restrict__ hist, float guidance, int B, int h, int w,
                                float* __restrict__ eps_out) {
    const int n = B * 4 * h * w;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int step = *step_idx;
    if (i < n) {
        const float* cf = coef + (size_t)step * 16;
        int xx = i % w;
        int yy = (i / w) % h;
        int c = (i / (w * h)) % 4;
        int b = i / (4 * w * h);
        // eps token-major [2B][h][2w][4]; right half, uncond = batch b, cond = batch B + b   (pipe:1092-1098)
        size_t pu = (((size_t)b * h + yy) * (2 * w) + (w + xx)) * 4 + c;
        size_t pc = (((size_t)(B + b) * h + yy) * (2 * w) + (w + xx)) * 4 + c;
        float eu = eps[pu], ec = eps[pc];
        const float gscale = guidance >= 0.f ? guidance : cf[11];     // < 0: read from the coefficient table (graph-replay safe)
        float e = eu + gscale * (ec - eu);
        if (eps_out) eps_out[i] = e;
        float x = latents[i];
        float* m0 = hist, *m1 = hist + n, *last = hist + 2 * (size_t)n;
        float x0 = x * cf[0] - e * cf[1];
        float xc = x;
        float pm0 = m0[i], pm1 = m1[i];
        if (cf[2] != 0.f) xc = cf[3] * last[i] + cf[4] * pm0 + cf[5] * pm1 + cf[6] * x0;
        float xn = cf[7] * xc + cf[8] * x0 + cf[9] * pm0 + cf[10] * e;
        m1[i] = pm0;
        m0[i] = x0;
        last[i] = xc;
        latents[i] = xn;
    }
}

__global__ void advance_kernel(int* step_idx) { *step_idx += 1; }

__global__ void nchw_to_nhwc_kernel(const void* __restrict__ src, int src_f32, int B, int C, int HW, int Cpad,
                                    h16* __restrict__ dst) {
    const long long total = (long long)B * HW * Cpad;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int c = (int)(i % Cpad);
        long long pix = i / Cpad;
        int p = (int)(pix % HW);
        int b = (int)(pix / HW);
        float v = 0.f;
        if (c < C) {
            size_t s = ((size_t)b * C + c) * HW + p;
            v = src_f32 ? reinterpret_cast<const float*>(src)[s] : (float)reinterpret_cast<const h16*>(src)[s];
        }
        dst[i] = (h16)v;
    }
}

__global__ void nhwc_to_nchw_kernel(const h16* __restrict__ src, int B, int C, int HW, int ldsrc, void* __restrict__ dst,
                                    int dst_f32) {
    const long long total = (long long)B * C * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int p = (int)(i % HW);
        int c = (int)((i / HW) % C);
        int b = (int)(i / ((long long)HW * C));
        float v = (float)src[((size_t)b * HW + p) * ldsrc + c];
        if (dst_f32) reinterpret_cast<float*>(dst)[i] = v;
        else reinterpret_cast<h16*>(dst)[i] = (h16)v;
    }
}

// out[b][0] = cls + pos[0]; out[b][1 + t] = patches[b][t] + pos[1 + t]
__global__ void add_cls_pos_kernel(const h16* __restrict__ patches, const float* __restrict__ cls,
                                   const float* __restrict__ pos, int B, int T, int D, h16* __restrict__ out) {
    const long long total = (long long)B * (T + 1) * D;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int d = (int)(i % D);
        int t = (int)((i / D) % (T + 1));
        int b = (int)(i / ((long long)D * (T + 1)));
        float v = (t == 0) ? cls[d] : (float)patches[((size_t)b * T + (t - 1)) * D + d];
        out[i] = (h16)(v + pos[(size_t)t * D + d]);
    }
}

// im2col for kernel = stride = patch: row (b, gy, gx), column (c, ky, kx) matching Conv2d weight.flatten(1)
__global__ void patchify_kernel(const float* __restrict__ px, int B, int H, int W, int patch, int Kpad, h16* __restrict__ out) {
    const int gh = H / patch, gw = W / patch;
    const long long total = (long long)B * gh * gw * Kpad;
    const int K = 3 * patch * patch;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int k = (int)(i % Kpad);
        long long row = i / Kpad;
        int gx = (int)(row % gw);
        int gy = (int)((row / gw) % gh);
        int b = (int)(row / ((long long)gw * gh));
        float v = 0.f;
        if (k < K) {
            int c = k / (patch * patch);
            int r = k - c * patch * patch;
            int ky = r / patch, kx = r - ky * patch;
            v = px[(((size_t)b * 3 + c) * H + gy * patch + ky) * W + gx * patch + kx];
        }
        out[i] = (h16)v;
    }
}

// vae.py:767-789: mean + exp(0.5 * clamp(logvar, -30, 20)) * noise, times the latent scaling factor
__global__ void gaussian_sample_kernel(const h16* __restrict__ mom, const float* __restrict__ noise, int B, int Cz, int HW,
                                       float scale, float* __restrict__ out) {
    const long long total = (long long)B * Cz * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int p = (int)(i % HW);
        int c = (int)((i / HW) % Cz);
        int b = (int)(i / ((long long)HW * Cz));
        const h16* m = mom + ((size_t)b * HW + p) * (2 * Cz);
        float mean = (float)m[c];
        float logvar = fminf(fmaxf((float)m[Cz + c], -30.f), 20.f);
        out[i] = (mean + expf(0.5f * logvar) * noise[i]) * scale;
    }
}

// CLIPTextEmbeddings (transformers models/clip/modeling_clip.py): token_embedding(ids) + position_embedding(arange(T))
__global__ void embed_tokens_kernel(const long long* __restrict__ ids, const h16* __restrict__ tok, const float* __restrict__ pos,
                                    int B, int T, int D, int vocab, h16* __restrict__ out) {
    const long long total = (long long)B * T * (D / 8);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % (D / 8));
        const long long row = i / (D / 8);
        const int t = (int)(row % T);
        long long id = ids[row];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        const uint4 raw = bc_ld16(tok + (size_t)id * D + ch * 8);
        const h16* e = reinterpret_cast<const h16*>(&raw);
        uint4 o;
        h16* oh = reinterpret_cast<h16*>(&o);
#pragma unroll
        for (int j = 0; j < 8; ++j) oh[j] = (h16)((float)e[j] + pos[(size_t)t * D + ch * 8 + j]);
        bc_st16(out + (size_t)row * D + ch * 8, o);
    }
}

inline int ew_blocks(long long total) { return (int)std::min<long long>((total + 255) / 256, 256 * 8); }

}  // namespace

extern "C" int bc_splat_scores(const double* params_host, int n, int h, int w, double* out, bc_stream stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    BC_CHECK_ARG(params_host && out && n > 0 && n <= 16 && h > 0 && w > 0, "bc_splat_scores: bad args (n<=16)");
    SplatParams prm;   // travels as a kernel argument: no allocation, no host->device copy, graph-capturable
    for (int i = 0; i < 8 * n; ++i) prm.v[i] = params_host[i];
    hipLaunchKernelGGL(splat_kernel, dim3(bc_ceil_div(h * w, 256), n), dim3(256), 0, stream, prm, h, w, out);
    BC_CHECK_LAUNCH();
    return 0;
}

extern "C" int bc_assemble_input(const float* latents, int Blat, const float* img_lat, const float* score,
                                 const float* feat, int Bimg, int F, int Bout, int h, int w, int Cpad, int dup_score,
                                 bc_half* X, bc_stream stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!feat) F = 0;
    BC_CHECK_ARG(latents && img_lat && score && X && Blat > 0 && Bout > 0 && Bimg > 0, "bc_assemble_input: bad args");
    BC_CHECK_ARG(Cpad % 8 == 0 && Cpad >= 5 + F, "bc_assemble_input: Cpad=%d must be a multiple of 8 and >= %d", Cpad, 5 + F);
    long long total = (long long)Bout * h * 2 * w * (Cpad / 8);
    hipLaunchKernelGGL(assemble_kernel, dim3(ew_blocks(total)), dim3(256), 0, stream, latents, Blat, img_lat, score, feat,
                       Bimg, F, Bout, h, w, Cpad, dup_score, reinterpret_cast<h16*>(X));
    BC_CHECK_LAUNCH();
    return 0;
}

extern "C" int bc_assemble_input_im2col(const float* latents, int Blat, const float* img_lat, const float* score, int Bimg, int Bout,
                                        int h, int w, int dup_score, bc_half* X, bc_stream stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    BC_CHECK_ARG(latents && img_lat && score && X && Blat > 0 && Bout > 0 && Bimg > 0, "bc_assemble_input_im2col: bad args");
    long long total = (long long)Bout * h * 2 * w * 16;
    hipLaunchKernelGGL(assemble_im2col_kernel, dim3(ew_blocks(total)), dim3(256), 0, stream, latents, Blat, img_lat, score, Bimg, Bout,
                       h, w, dup_score, reinterpret_cast<h16*>(X));
    BC_CHECK_LAUNCH();
    return 0;
}

extern "C" int bc_timestep_embedding(const float* t_table, const int* t_idx, float t_value, int rows, int dim,
                                     bc_half* out, bc_stream stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    BC_CHECK_ARG(out && rows > 0 && dim > 0 && dim % 2 == 0, "bc_timestep_embedding: bad args");
    hipLaunchKernelGGL(temb_kernel, dim3(bc_ceil_div(rows * dim, 256)), dim3(256), 0, stream, t_table, t_idx, t_value, rows,
                       dim, 0, reinterpret_cast<h16*>(out));
    BC_CHECK_LAUNCH();
    return 0;
}

extern "C" int bc_timestep_embedding_table(const float* t_table, int nsteps, int rows_per_step, int dim, bc_half* out,
                                           bc_stream stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    BC_CHECK_ARG(t_table && out && nsteps > 0 && rows_per_step > 0 && dim > 0 && dim % 2 == 0, "bc_timestep_embedding_table: bad args");
    const int rows = nsteps * rows_per_step;
    hipLaunchKernelGGL(temb_kernel, dim3(bc_ceil_div(rows * dim, 256)), dim3(256), 0, stream, t_table, nullptr, 0.f, rows, dim,
                       rows_per_step, reinterpret_cast<h16*>(out));
    BC_CHECK_LAUNCH();
    return 0;
}

extern "C" int bc_silu(const bc_half* x, bc_half* y, long long n, bc_stream stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    BC_CHECK_ARG(x && y && n > 0, "bc_silu: bad args");
    hipLaunchKernelGGL(silu_kernel, dim3(ew_blocks(n)), dim3(256), 0, stream, reinterpret_cast<const h16*>(x),
                       reinterpret_cast<h16*>(y), n);
    BC_CHECK_LAUNCH();
    return 0;
}

extern "C" int bc_cfg_scheduler_step(const float* eps, float* latents, const float* coef, int* step_idx, float* hist,
                                     float guidance_scale, int B, int h, int w, float* eps_out, int advance,
                                     bc_stream stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    BC_CHECK_ARG(eps && latents && coef && step_idx && hist && B > 0, "bc_cfg_scheduler_step: bad args");
    int n = B * 4 * h * w;
    hipLaunchKernelGGL(cfg_step_kernel, dim3(bc_ceil_div(n, 256)), dim3(256), 0, stream, eps, latents, coef, step_idx, hist,
                       guidance_scale, B, h, w, eps_out);
    BC_CHECK_LAUNCH();
    if (advance) {
        hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(1), 0, stream, step_idx);
        BC_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int bc_nchw_to_nhwc_f16(const void* src, int src_is_f32, int B, int C, int HW, int Cpad, bc_half* dst,
                                   bc_stream stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    BC_CHECK_ARG(src && dst && Cpad >= C, "bc_nchw_to_nhwc_f16: bad args");
    long long total = (long long)B * HW * Cpad;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(ew_blocks(total)), dim3(256), 0, stream, src, src_is_f32, B, C, HW, Cpad,
                       reinterpret_cast<h16*>(dst));
    BC_CHECK_LAUNCH();
    return 0;
}

extern "C" int bc_nhwc_to_nchw(const bc_half* src, int B, int C, int HW, int ldsrc, void* dst, int dst_is_f32,
                               bc_stream stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    BC_CHECK_ARG(src && dst && ldsrc >= C, "bc_nhwc_to_nchw: bad args");
    long long total = (long long)B * C * HW;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(ew_blocks(total)), dim3(256), 0, stream, reinterpret_cast<const h16*>(src), B,
                       C, HW, ldsrc, dst, dst_is_f32);
    BC_CHECK_LAUNCH();
    return 0;
}

extern "C" int bc_add_cls_pos(const bc_half* patches, const float* cls, const float* pos, int B, int T, int D,
                              bc_half* out, bc_stream stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    BC_CHECK_ARG(patches && cls && pos && out, "bc_add_cls_pos: bad args");
    long long total = (long long)B * (T + 1) * D;
    hipLaunchKernelGGL(add_cls_pos_kernel, dim3(ew_blocks(total)), dim3(256), 0, stream, reinterpret_cast<const h16*>(patches),
                       cls, pos, B, T, D, reinterpret_cast<h16*>(out));
    BC_CHECK_LAUNCH();
    return 0;
}

extern "C" int bc_patchify(const float* pixels, int B, int H, int W, int patch, int Kpad, bc_half* out, bc_stream stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    BC_CHECK_ARG(pixels && out && H % patch == 0 && W % patch == 0 && Kpad >= 3 * patch * patch && Kpad % 8 == 0,
                 "bc_patchify: bad args");
    long long total = (long long)B * (H / patch) * (W / patch) * Kpad;
    hipLaunchKernelGGL(patchify_kernel, dim3(ew_blocks(total)), dim3(256), 0, stream, pixels, B, H, W, patch, Kpad,
                       reinterpret_cast<h16*>(out));
    BC_CHECK_LAUNCH();
    return 0;
}

extern "C" int bc_gaussian_sample(const bc_half* moments, const float* noise, int B, int Cz, int HW, float scale, float* out,
                                  bc_stream stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    BC_CHECK_ARG(moments && noise && out && B > 0 && Cz > 0 && HW > 0, "bc_gaussian_sample: bad args");
    long long total = (long long)B * Cz * HW;
    hipLaunchKernelGGL(gaussian_sample_kernel, dim3(ew_blocks(total)), dim3(256), 0, stream,
                       reinterpret_cast<const h16*>(moments), noise, B, Cz, HW, scale, out);
    BC_CHECK_LAUNCH();
    return 0;
}

extern "C" int bc_embed_tokens(const long long* ids, const bc_half* tok_emb, const float* pos_emb, int B, int T, int D, int vocab,
                               bc_half* out, bc_stream stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    BC_CHECK_ARG(ids && tok_emb && pos_emb && out && B > 0 && T > 0 && D > 0 && D % 8 == 0 && vocab > 0, "bc_embed_tokens: bad args");
    long long total = (long long)B * T * (D / 8);
    hipLaunchKernelGGL(embed_tokens_kernel, dim3(ew_blocks(total)), dim3(256), 0, stream, ids, reinterpret_cast<const h16*>(tok_emb),
                       pos_emb, B, T, D, vocab, reinterpret_cast<h16*>(out));
    BC_CHECK_LAUNCH();
    return 0;
}
