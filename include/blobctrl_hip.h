/*
 * blobctrl_hip.h  --  C ABI of libblobctrl_hip.so, the MI355X (gfx950) kernels for the BlobCtrl denoising hot path.
 *
 * Boundary contract: the reference has no FFI layer, its hot path is a chain of ATen calls made from
 * Python nn.Modules.  This library replaces those ATen calls one for one; every entry point below cites the reference
 * call site(s) it stands in for (paths relative to the reference root, D/ = diffusers/src/diffusers/).
 *
 * Conventions
 *   - plain C: pointers are DEVICE pointers unless marked host; sizes are ints; no torch / C++ types.
 *   - activations are fp16 ("bc_half" = IEEE binary16) in NHWC / token-major layout [B][H*W][C]; accumulation,
 *     normalisation statistics, softmax and the scheduler state are fp32.
 *   - every K-contiguous dimension (channels, head_dim) is a multiple of 8 elements (16 bytes); callers pad.
 *   - all work is enqueued on the caller's hipStream_t (passed as void*); no internal threads, no allocation, no
 *     synchronisation: every launch function is hipGraph-capturable.  The caller owns all buffers.
 *   - return value: 0 on success, non-zero on error; bc_last_error() returns a thread-local message.
 */
#ifndef BLOBCTRL_HIP_H
#define BLOBCTRL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t bc_half;          /* IEEE fp16 bit pattern */
typedef void*    bc_stream;        /* hipStream_t */

const char* bc_last_error(void);
int bc_version(void);
/* Device properties of the current device: [0]=multiProcessorCount, [1]=warpSize, [2]=sharedMemPerBlock(KiB), [3]=gcn major*100+minor */
int bc_device_info(int* out4);

/* ---------------------------------------------------------------------------------------------------------------
 * Implicit GEMM on MFMA (v_mfma_f32_32x32x16_f16):   C[M][N] = epilogue( A[M][K] . W[N][K]^T )
 * Replaces: F.conv2d 3x3 / 1x1 (D/models/resnet.py:341,366,368; downsampling.py:147; upsampling.py:180;
 *           transformer_2d.py:484,521; unet_2d_condition.py:1172,1344; blobnet.py:840,862-864,881,922-924),
 *           F.linear (attention_processor.py:2191-2224; attention.py:1165-1167; activations.py:117-123;
 *           embeddings.py:576-588; resnet.py:343-350) and the patched right-half residual add
 *           (unet_2d_blocks.py:1303-1307 et al.).
 * --------------------------------------------------------------------------------------------------------------- */
enum { BC_A_DENSE = 0, BC_A_CONV3X3 = 1 };
enum { BC_ACT_NONE = 0, BC_ACT_GELU = 1, BC_ACT_GEGLU = 2, BC_ACT_SILU = 3, BC_ACT_QUICK_GELU = 4 /* x*sigmoid(1.702x), CLIP */ };
enum { BC_OUT_F16 = 0, BC_OUT_F16_T = 1, BC_OUT_F32 = 2 };

typedef struct BcGemm {
    /* ---- A operand ---- */
    const bc_half* A;        /* DENSE: [M][lda] rows; CONV3X3: NHWC image [B][Hin][Win][Cin] */
    const bc_half* A2;       /* optional second source (DENSE, or CONV3X3 on BC_TILE_HALO): columns / channels k >= C1 come from
                              * A2[..][k - C1] (channel concat) */
    int a_mode;              /* BC_A_* */
    int M, N, K;             /* GEMM dims; CONV3X3: M = B*Hout*Wout, K = 9*Cin; weights are [N][ky][kx][Cin] */
    int lda, lda2;           /* DENSE row strides (elements); CONV3X3 on BC_TILE_HALO: pixel strides of A / A2 (0 = Cin) */
    int C1;                  /* DENSE with A2: split point (multiple of 8); else ignored */
    int Cin;                 /* CONV3X3: input channels (multiple of 8) */
    int Hin, Win;            /* CONV3X3: stored input image size */
    int Hv, Wv;              /* CONV3X3: virtual (nearest-upsampled) input size the 3x3 window slides over; = Hin,Win when no upsample */
    int Hout, Wout;          /* CONV3X3: output size */
    int stride;              /* CONV3X3: 1 or 2 */
    int conv_nopad_lo;       /* CONV3X3: 0 = pad 1 on every side; 1 = no top/left padding (pad only bottom/right): the VAE encoder's
                              * Downsample2D(padding=0) = F.pad(x, (0,1,0,1)) + conv stride 2 (D/models/downsampling.py:141-147) */
    /* ---- B operand ---- */
    const bc_half* W;        /* [N][ldw], K contiguous */
    int ldw;
    /* ---- epilogue:  v = acc (+bias[n]) (+rowvec[m / rows_per_batch][n]) ; v = act(v) ; v *= colscale[n] ; v *= alpha ;
     *                 v += R[m][n] ; v += R2[(b % r2_bmod, pixel)][n] if x(m) >= r2_xmin ---- */
    const float*   bias;     /* [N] fp32 or NULL (GEGLU: [N] in the interleaved order of W) */
    const bc_half* rowvec;   /* per-batch row vector [B][ld_rowvec] (time-embedding projection) or NULL */
    int ld_rowvec;
    int rows_per_batch;      /* H*W of the output (for rowvec / R2 / transposed-output indexing); 0 => M */
    const int* rowvec_idx;   /* optional device step counter: the row vector is read at rowvec + *rowvec_idx * rowvec_step (a table
                              * with one block of row vectors per denoise step, computed once per edit) */
    int rowvec_step;         /* elements (bc_half) between the blocks of consecutive steps */
    int act;                 /* BC_ACT_* ; GEGLU: columns come in groups of 64 = 32 value | 32 gate, output has N/2 columns */
    const float*   colscale; /* [N] or NULL (DINOv2 LayerScale) */
    float alpha;             /* scalar multiplier (1.0f default) */
    const float*   alpha_dev;/* optional device scalar table: alpha *= alpha_dev[*alpha_idx] (BlobNet conditioning_scale*keep[i]) */
    const int*     alpha_idx;
    int alpha_bstride;       /* 0: one scalar per launch; B > 0: per-image scalars alpha_dev[*alpha_idx * B + m / rows_per_batch]
                              * (a batch of independent edit requests with their own conditioning scales) */
    const bc_half* R;        /* residual [M][ldr] or NULL */
    int ldr;
    const bc_half* R2;       /* BlobNet residual, token-major [r2_bmod][rows_per_batch][ldr2], added where x >= r2_xmin */
    int ldr2, r2_xmin, r2_bmod, out_w;   /* out_w = Wout (x = m % out_w) */
    /* ---- output ---- */
    int out_mode;            /* BC_OUT_F16: C[m*ldc + n]; BC_OUT_F16_T: C[(b*N + n)*ldc + (m % rows_per_batch)]; BC_OUT_F32: float C */
    void* C;
    int ldc;
    /* ---- split-K ---- */
    int splitk;              /* >=1 ; >1 needs `slab` of splitk*M*N floats */
    float* slab;
    /* ---- optional fused GroupNorm statistics of the (fp16-rounded) OUTPUT: every workgroup ADDS the per-channel (sum, sumsq) of its
     *      rows to the totals gn_tot[B][n_out][BC_GN_TOT_WORDS] (see "GroupNorm statistics totals" below; the table must be zero before
     *      the launch).  Fast path or split-K reducer only; needs BC_OUT_F16, widths % 8 == 0 and rows_per_batch % slab_rows == 0
     *      (slab_rows = bm of bc_gemm_plan, or 32 with split-K: a workgroup's rows lie in one image).  Replaces the bc_gn_stats pass. ---- */
    unsigned long long* gn_tot;
    /* ---- tile configuration: 0 = library heuristic, else one of BC_TILE_* (see bc_gemm_plan) ---- */
    int tile_cfg;
    /* ---- A-operand prologue (BC_TILE_HALO only): the GroupNorm(+SiLU) in front of a ResBlock convolution
     *      (D/models/resnet.py:327-328,351-363) applied while the input halo tile is staged:
     *      a := act( A[.., k] * a_affine[b][k][0] + a_affine[b][k][1] ), k over the channel concat (A | A2), a_affine fp32
     *      [B][Cin][2] as written by bc_gn_finalize; zero padding applies to the activated value.  NULL = plain convolution. ---- */
    const float* a_affine;
    int a_act;               /* BC_ACT_NONE or BC_ACT_SILU */
    /* ---- ... or the same prologue with the GroupNorm FINALIZE done inside the convolution (no bc_gn_finalize launch): every
     *      workgroup reads the statistics totals of the groups overlapping its channel range (a_tot1 [B][C1][BC_GN_TOT_WORDS] for the
     *      channels of A, a_tot2 [B][Cin - C1][..] for A2; same layout as gn_tot), then applies (x - mean) * rstd * a_gamma[k] +
     *      a_beta[k] and a_act.  Used when a_tot1 != NULL (a_affine is then ignored); needs the workgroup's channel span (+ group
     *      straddle) <= 2752 channels (BC_TILE_HALO: 720), else call bc_gn_finalize and pass a_affine. ---- */
    const unsigned long long* a_tot1;
    const unsigned long long* a_tot2;
    const float* a_gamma; const float* a_beta;
    int a_groups; float a_eps;
    /* ---- BC_TILE_GW* only: LayerNorm in front of the projection, FOLDED (attention.py:447,491,517 norm1/2/3 -> to_q|k|v, ff.net.0):
     *      W holds W * diag(gamma), bias holds bias + W beta, ln_colsum[n] = sum_k (fp16) W'[n][k]; A is the RAW residual stream, the
     *      kernel accumulates the row statistics while it stages the rows and applies
     *      rstd_m * (acc[m][n] - mean_m * ln_colsum[n]) + bias[n]  (eps = ln_eps).  NULL = plain projection. ---- */
    const float* ln_colsum; float ln_eps;
    /* ---- BC_TILE_GW* only: a second, TRANSPOSED output for the column tiles n >= n_t0 (q | k row-major into C, V^T for the attention
     *      kernel in the same launch): C_t[(b * (N - n_t0) + n - n_t0) * ldc_t + (m % rows_per_batch)].  NULL = none. ---- */
    void* C_t; int ldc_t, n_t0;
    /* ---- BC_TILE_GW* only: PER-IMAGE weights (round 5: the cross-attention of the 1280-channel blocks folded over the prompt, see
     *      bc_ctx_fold): the rows of image b = m / rows_per_batch multiply the stream W + b * w_bstride (bc_half elements; 0 = one
     *      stream for every row); with vec_bstride != 0 `bias` and `ln_colsum` are per image too ([B][vec_bstride] floats). ---- */
    long long w_bstride; int vec_bstride;
    /* ---- BC_TILE_GW64x128 only: row SOFTMAX over each workgroup's 128 columns (sm_group = 128: a head's keys, padded) of which the
     *      first `sm_valid` are keys; the first `sm_keep` columns (% 8 == 0, >= sm_valid; columns >= sm_valid come out as 0) of every
     *      group are written as fp16 probabilities, compacted: C[m][(n / 128) * sm_keep + n % 128].  The scores never exist in HBM (fp32
     *      inside the launch).  0 = none. ---- */
    int sm_group, sm_valid, sm_keep;
} BcGemm;

int bc_gemm(const BcGemm* p, bc_stream stream);
int bc_sizeof_gemm(void);            /* sizeof(BcGemm), lets FFI bindings verify their struct mirror */
/* Tile configurations of the LDS-DMA fast path (block tile BM x BN, waves, LDS stages). */
enum { BC_TILE_AUTO = 0, BC_TILE_256x128 = 1, BC_TILE_128x128_S3 = 2, BC_TILE_128x128_S2 = 3, BC_TILE_256x64_S2 = 4,
       BC_TILE_256x64_S3 = 5, BC_TILE_128x64 = 6, BC_TILE_64x64 = 7, BC_TILE_COUNT = 8,
       /* LDS-resident input-halo convolution (conv_halo.hip): 3x3 / stride 1 / pad 1, Cin % 64 == 0, N % 160 == 0, Wout % 16 == 0,
        * Hout % 8 == 0; workgroup = 8 x 16 pixels x 160 channels; takes two channel-concatenated sources (A | A2, C1 % 64 == 0,
        * lda / lda2 = their pixel strides) and the fused GroupNorm prologue (a_affine); splitk counts 64-channel chunks.  Never
        * chosen by BC_TILE_AUTO: callers ask for it (bc_conv_halo_eligible). */
       BC_TILE_HALO = 8,
       /* the same convolution, tile, prologue and epilogue with the weights streamed straight into VGPRs (conv_wreg.hip): `W` is the
        * fragment stream bc_conv_wreg_pack wrote for this layer (not the row-major matrix), ldw is ignored.  Same eligibility. */
       BC_TILE_WREG = 9,
       /* small-M projections with the weights streamed straight into VGPRs (gemm_wreg.hip): dense A (one or two sources), M % 64 == 0,
        * K % 320 == 0 (C1 % 320 == 0), workgroup = 64 rows x 128 / 256 / 320 columns (N a multiple of that); `W` is the fragment stream
        * bc_gemm_wreg_pack wrote for the SAME configuration, ldw is ignored; fp16 row-major output, no split-K, no row vector; takes
        * ln_colsum (folded LayerNorm), C_t (transposed second output) and a_tot1 / a_gamma / a_beta / a_groups / a_eps (the GroupNorm in front
        * of a Transformer2D's proj_in finalized in the prologue and applied while the rows are staged; one source, K <= 2560, no
        * activation).  Never chosen by BC_TILE_AUTO (bc_gemm_wreg_eligible). */
       BC_TILE_GW64x128 = 10, BC_TILE_GW64x256 = 11, BC_TILE_GW64x320 = 12,
       /* large-M dense projections (gemm256.hip, round 6): 256 x 256 tiles, 8 waves, the 8-phase LDS-DMA pipeline, persistent workgroups
        * (grid = min(tiles, CUs), the next tile's first loads under this tile's epilogue).  Dense A (one source, or two split at
        * C1 % 128 == 0), M % 256 == 0, N % 256 == 0, K % 128 == 0, no split-K; the whole shared epilogue (bias, row vector, GEGLU / GELU / SiLU, scales, residual,
        * BlobNet right-half residual, GroupNorm statistics totals when rows_per_batch % 256 == 0) for BC_OUT_F16, bias + alpha for
        * BC_OUT_F16_T.  Replaces, at batch >= 2, the Linear / 1 x 1 layers of the 1280-channel transformer blocks
        * (attention.py:1161-1167, attention_processor.py:2191-2224, transformer_2d.py:479-527).  Never chosen by BC_TILE_AUTO
        * (bc_gemm256_eligible). */
       BC_TILE_G256 = 13 };
/* Re-order a weight matrix w[N][ldw] into the per-wave fragment streams of a BC_TILE_GW* configuration (out of place;
 * bc_gemm_wreg_stream_elems(N, K) elements incl. the tail the register ring reads past the end): [column tile][wave 4][k-step K / 32]
 * [tile][64 lanes][8]; lane l holds w[n0 + 16 tile + (l & 15)][32 s + 8 (l >> 4) .. + 8]. */
int bc_gemm_wreg_pack(const bc_half* w, int ldw, int N, int K, int tile_cfg, bc_half* out, bc_stream stream);
long long bc_gemm_wreg_stream_elems(int N, int K);
/* 1 when a dense projection (C1 > 0: two-source A split at C1) can run on the given BC_TILE_GW* configuration. */
int bc_gemm_wreg_eligible(int M, int N, int K, int C1, int tile_cfg);
/* 1 when a dense projection (C1 > 0: two-source A split at C1, C1 % 128 == 0) can run on BC_TILE_G256 (out_mode BC_OUT_F16 / BC_OUT_F16_T;
 * want_gn = 1: the launch also produces GroupNorm statistics totals, which needs a tile's 256 rows inside one image).  BC_TILE_G256 also
 * takes C_t / ldc_t / n_t0 (n_t0 % 256 == 0: the column tiles from n_t0 on are written transposed, plain projections only). */
int bc_gemm256_eligible(int M, int N, int K, int C1, int out_mode, int rows_per_batch, int want_gn);
/* Re-order a 3x3 weight matrix w[N][9 * Cin] (k = (ky * 3 + kx) * Cin + c; N % 160 == 0, Cin % 64 == 0) into the per-wave fragment
 * streams of BC_TILE_WREG (same size, out of place): per 160-column block, per (column group 3|2|2|3 tiles, K half of the 64-channel
 * chunk) one contiguous stream [chunk][kx][ky][tile][64 lanes][8]; lane l holds w[n0 + 16 tile + (l & 15)][k0 + 8 (l >> 4) .. + 8]. */
int bc_conv_wreg_pack(const bc_half* w, int N, int Cin, bc_half* out, bc_stream stream);
/* 1 when a convolution can run on BC_TILE_HALO. */
int bc_conv_halo_eligible(int Cin, int C1, int N, int Hin, int Win, int Hout, int Wout, int stride);
/* most 64-channel chunks one workgroup of BC_TILE_HALO may take: callers keep ceil(Cin / 64 / splitk) <= this */
int bc_conv_halo_max_chunks(void);
/* Resolve the plan for a GEMM: in/out *tile_cfg (AUTO -> heuristic choice), in/out *splitk (<= 0 -> heuristic), out *bm,
 * *bn = tile shape.  `fast` = 1 when the problem meets the fast-path conditions
 * (K % 64 == 0, conv Cin % 64 == 0, concat split % 64 == 0); otherwise only 128x128 / 256x64 generic tiles exist. */
int bc_gemm_plan(int M, int N, int K, int fast, int* tile_cfg, int* splitk, int* bm, int* bn);

/* ---------------------------------------------------------------------------------------------------------------
 * GroupNorm (+SiLU) over NHWC, optionally over a channel-concat of two tensors.
 * Replaces F.group_norm + F.silu (resnet.py:327-328,351-363; transformer_2d.py:481; unet_2d_condition.py:1341-1343)
 * and torch.cat([h, skip], 1) (unet_2d_blocks.py:2559,2719) feeding it.
 * GroupNorm statistics totals (round 4): tot[B][C][BC_GN_TOT_WORDS] 64-bit words per (image, channel) = the sum and the sum of
 * squares of the fp16 activation over the image, each as three signed 40-bit slices of a fixed-point number (units 2^-60, 2^-20,
 * 2^20).  Producers (the GEMM / convolution epilogues: BcGemm.gn_tot, bc_rowchain, or bc_gn_stats) reduce their rows in fp32 and ADD
 * the result with integer atomics - integer sums do not depend on the arrival order, so results stay bit-reproducible - and a
 * consumer reads six words per channel.  A table must be zero before its producers run: bc_memset_zero (one per plan segment for
 * all tables of that segment).
 * BLOCKS (round 6): in a table whose width C is a multiple of 320 the GEMM / convolution / row-chain epilogues add ONE partial per block of
 * 10 consecutive channels, into slot block + (producer row tile % 10) - the block's ten channel slots are ten accumulators of the block's
 * sums (atomics on one cache line execute one after the other: a tenth of the adds, a tenth of the chain).  Only sums over WHOLE blocks are
 * meaningful in such a table: it serves a GroupNorm whose groups are unions of blocks - (C1 + C2) / G a multiple of 10 and C1 a multiple
 * of 10, which holds for every GroupNorm of the UNet / BlobNet - and no other; for any other consumer run bc_gn_stats (always one slot per
 * channel) into a table of its own.  Tables of other widths are per channel as before.
 *   bc_gn_stats    : statistics of ONE tensor added to tot[B][C][..].  Only needed when the producing GEMM did not emit them itself.
 *   bc_gn_finalize : per-channel affine of the concat (x1 | x2): ab[B][C1+C2][2] = (rstd*gamma, beta - mean*rstd*gamma),
 *                    from tot_i[B][C_i][..] (tot2 may be NULL)
 *   bc_gn_apply    : y[B][HW][C1+C2] = silu?( x*ab.x + ab.y )   (x = concat(x1, x2))
 * --------------------------------------------------------------------------------------------------------------- */
enum { BC_GN_TOT_WORDS = 6 };
int bc_gn_stats(const bc_half* x, int C, int B, int HW, unsigned long long* tot, bc_stream stream);
int bc_gn_finalize(const unsigned long long* tot1, int C1, const unsigned long long* tot2, int C2, int B, int HW, int G,
                   float eps, const float* gamma, const float* beta, float* ab, bc_stream stream);
int bc_gn_apply(const bc_half* x1, int C1, const bc_half* x2, int C2, int B, int HW,
                const float* ab, int silu, bc_half* y, bc_stream stream);
/* finalize + apply in ONE launch (what the engine uses): every workgroup reads the totals of the groups overlapping
 * its 64-channel range, then normalises its pixels. */
int bc_gn_apply_fused(const unsigned long long* tot1, int C1, const unsigned long long* tot2, int C2,
                      const bc_half* x1, const bc_half* x2, int B, int HW, int G, float eps, const float* gamma,
                      const float* beta, int silu, bc_half* y, bc_stream stream);
/* Zero `bytes` bytes at `ptr` (both multiples of 16) with a kernel on the caller's stream (a kernel node when captured): zeroes the
 * statistics totals at the head of a segment. */
int bc_memset_zero(void* ptr, long long bytes, bc_stream stream);
/* Up to six buffers laid out [2][bytes_k] (16-byte aligned, bytes_k % 16 == 0; null / 0 = unused slot): the first half of each is
 * copied over its second half, one launch.  Round 6: the CFG-invariant prefix of the UNet (pipe:1031 `torch.cat([latents] * 2)`: both
 * classifier-free-guidance images enter the UNet identical, the prompt first enters at attn2 of down_blocks.0.attentions.0,
 * attention.py:504-510) is computed for ONE image per pair; this op fans its four activations (skip #0, the ResBlock output, the
 * Transformer2D's h0 and self-attention output) and skip #0's GroupNorm statistics totals out to the pair. */
int bc_dup_halves(void* p0, long long bytes0, void* p1, long long bytes1, void* p2, long long bytes2, void* p3, long long bytes3,
                  void* p4, long long bytes4, void* p5, long long bytes5, bc_stream stream);

/* Row softmax in place on fp16 [rows][cols] (fp32 maths): the single-head, head_dim-512 attention of the VAE mid block is run
 * as GEMM (QK^T) -> softmax -> GEMM (PV)  (attention_processor.py:2216 with heads = 1). */
int bc_softmax_rows(bc_half* x, int rows, int cols, int ld, bc_stream stream);

/* CLIP text embeddings (first stage of `self.text_encoder(text_input_ids)`, pipeline_blobnet.py:599; transformers
 * models/clip/modeling_clip.py CLIPTextEmbeddings): out[b][t][:] = tok_emb[ids[b][t]][:] + pos_emb[t][:]  (ids int64, clamped to the
 * vocabulary). */
int bc_embed_tokens(const long long* ids, const bc_half* tok_emb, const float* pos_emb, int B, int T, int D, int vocab,
                    bc_half* out, bc_stream stream);

/* Posterior sample of the VAE encoder (D/models/autoencoders/vae.py:767-789): moments token-major [B][HW][2*Cz] (mean | logvar),
 * noise fp32 NCHW [B][Cz][HW]; out fp32 NCHW = (mean + exp(0.5*clamp(logvar,-30,20)) * noise) * scale. */
int bc_gaussian_sample(const bc_half* moments, const float* noise, int B, int Cz, int HW, float scale, float* out,
                       bc_stream stream);

/* LayerNorm over the last dim of [rows][C]  (attention.py:447,491,517; transformers Dinov2Layer norm1/norm2/layernorm). */
int bc_layernorm(const bc_half* x, int rows, int C, int ldx, const float* gamma, const float* beta, float eps,
                 bc_half* y, int ldy, bc_stream stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Flash attention forward, one launch per (self|cross) attention layer.
 * Replaces F.scaled_dot_product_attention (attention_processor.py:2216-2218) and the DINOv2 eager attention.
 *   Q  [B][Nq ][ldq] , head h at columns h*d .. h*d+d-1
 *   K  [B][Nkv][ldk]
 *   Vt [B][heads*d][ldvt]  (V TRANSPOSED: key index contiguous; ldvt >= Nkv rounded up to 64, padding zero)
 *   O  [B][Nq ][ldo]
 * softmax(scale * Q K^T) V with fp32 online softmax.  d in {8,16,32,40,64,80,160}.
 * --------------------------------------------------------------------------------------------------------------- */
int bc_attention(const bc_half* Q, const bc_half* K, const bc_half* Vt, bc_half* O,
                 int B, int heads, int d, int Nq, int Nkv,
                 int ldq, int ldk, int ldvt, int ldo,
                 long long q_bstride, long long k_bstride, long long vt_bstride, long long o_bstride,
                 float scale, bc_stream stream);
/* Same with the causal mask of the CLIP text encoder (key j visible to query i iff j <= i; pipe:599 -> transformers
 * CLIPTextTransformer); needs Nq == Nkv. */
int bc_attention_causal(const bc_half* Q, const bc_half* K, const bc_half* Vt, bc_half* O,
                        int B, int heads, int d, int Nq, int Nkv,
                        int ldq, int ldk, int ldvt, int ldo,
                        long long q_bstride, long long k_bstride, long long vt_bstride, long long o_bstride,
                        float scale, bc_stream stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Blob maths and loop glue
 * --------------------------------------------------------------------------------------------------------------- */
/* Gaussian-blob splat rasteriser: blobctrl/utils/utils.py:145-194 (tuple score_size branch, return_d_score=True).
 * params (HOST doubles, 8 per blob image): xs, ys, cov00, cov01, cov10, cov11, size, unused.
 * out: DEVICE double [n][2][h][w]  (channel 0 = background, 1 = foreground), computed in fp64 like the reference. */
int bc_splat_scores(const double* params_host, int n, int h, int w, double* out, bc_stream stream);

/* Input assembly: pipeline_blobnet.py:724-739 (construct_blobnet_input) fused with NCHW->NHWC, fp16 cast, channel padding
 * and the rank-1 feature splat (pipeline_blobnet.py:706-721).  Writes X[Bout][h][2w][Cpad]:
 *   left half  = (img_lat[bi][4], score[bi], score*feat[bi][0..F))   right half = (latents[b % Blat][4], score, score*feat)
 * with bi = b % Bimg.  latents fp32 [Blat][4][h][w] (NCHW), img_lat fp32 [Bimg][4][h][w], score fp32 [Bimg][h][w],
 * feat fp32 [Bimg][F] or NULL (F = 0).  dup_score != 0 (with F = 0) writes the score a second time into channel 5: the input of the
 * rank-1-collapsed BlobNet conv_in (the 1024 feature channels are score x vector, so their 3x3 conv equals a 1-channel conv of
 * the score with the per-edit kernel sum_c W[:, 5+c] * f_c). */
int bc_assemble_input(const float* latents, int Blat, const float* img_lat, const float* score, const float* feat,
                      int Bimg, int F, int Bout, int h, int w, int Cpad, int dup_score, bc_half* X, bc_stream stream);
/* The 8-channel form of the same input (4 latents, score, [second copy of the score], 0, 0) as the 3x3 im2col operand of conv_in:
 * X [Bout][h * 2w][128] fp16, k = tap * 8 + channel for the nine taps (zero outside the canvas), zero-filled from k = 72, so that
 * conv_in (pipe:724-739 -> unet_2d_condition.py:1166 / blobnet.py:812) is a dense K = 128 GEMM on the LDS-DMA fast path. */
int bc_assemble_input_im2col(const float* latents, int Blat, const float* img_lat, const float* score, int Bimg, int Bout,
                             int h, int w, int dup_score, bc_half* X, bc_stream stream);

/* Sinusoidal timestep embedding (embeddings.py:27-78, flip_sin_to_cos=True, shift 0) for `rows` identical rows.
 * t = t_table[*t_idx] when t_table != NULL else t_value.  out [rows][dim] fp16. */
int bc_timestep_embedding(const float* t_table, const int* t_idx, float t_value, int rows, int dim, bc_half* out,
                          bc_stream stream);
/* The same for EVERY step of an edit at once: out [nsteps * rows_per_step][dim], row r uses t_table[r / rows_per_step].  With
 * BcGemm.rowvec_idx the whole time-embedding path (embeddings.py:27-78, 576-588, resnet.py:343-350) then runs once per edit. */
int bc_timestep_embedding_table(const float* t_table, int nsteps, int rows_per_step, int dim, bc_half* out, bc_stream stream);

/* SiLU elementwise on fp16 (resnet.py:345 nonlinearity(temb)). */
int bc_silu(const bc_half* x, bc_half* y, long long n, bc_stream stream);

/* Crop (right half) + classifier-free guidance + scheduler step, all fp32 state
 * (pipeline_blobnet.py:1092-1102; scheduling_unipc_multistep.py:822-901 / scheduling_ddim.py:342-468 as linear
 * combinations with per-step host-precomputed coefficients).
 *   eps      : UNet output, fp32 token-major [2B][h][2w][4]  (uncond batch first, then cond)
 *   latents  : fp32 [B][4][h][w] NCHW, updated in place
 *   coef     : device table [nsteps][16] (layout in blobctrl_amd/schedulers.py), row = *step_idx
 *   hist     : fp32 [3][B*4*h*w] scheduler history (x0_prev, x0_prevprev, last_sample)
 *   eps_out  : optional fp32 [B][4][h][w] guided epsilon (for parity tracing) or NULL
 * guidance_scale < 0 => the scale is read from coef[*step_idx][11] (a captured graph then follows per-call values).
 * Increments *step_idx when advance != 0. */
int bc_cfg_scheduler_step(const float* eps, float* latents, const float* coef, int* step_idx, float* hist,
                          float guidance_scale, int B, int h, int w, float* eps_out, int advance, bc_stream stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Row-chain: everything of a Transformer2D block that acts on token rows independently, as ONE launch per attention
 * boundary (csrc/rowchain.hip).  Replaces, for 320- and 640-channel blocks (the 64 x 128 and 32 x 64 levels of SD-1.5; C / 80 waves
 * per 64-row workgroup), the call sites
 *   diffusers/src/diffusers/models/transformers/transformer_2d.py:479-527 (norm -> proj_in ... proj_out + residual),
 *   attention.py:421-541 (norm1/2/3, attn to_q / to_k / to_v / to_out, residual adds), activations.py:113-123 and
 *   attention.py:1161-1167 (GEGLU feed-forward), blobctrl/models/blobnet.py:860-864,921-924,936-938 (zero-conv x scale):
 *   BC_CHAIN_IN   x [M][C] --(GroupNorm affine [B][C][2], or the statistics totals gn_in [B][C][BC_GN_TOT_WORDS] of x with gn_gamma /
 *                 gn_beta / gn_groups / gn_eps: the finalize then runs in the kernel's prologue; or neither)--> proj_in -> out0 = h0 [M][C]; LayerNorm ->
 *                 out1 = q|k [M][2C] (row-major), out2 = V^T [B][C][ldvt]
 *   BC_CHAIN_MID  x = attention output, res = h0: to_out + res -> out0 = h1; LayerNorm -> out1 = attn2.to_q [M][C]
 *   BC_CHAIN_OUT  x = attention output, res = residual stream, res2 = the block's input: to_out + res -> LayerNorm ->
 *                 GEGLU feed-forward + residual -> proj_out + res2 (+ r2: BlobNet residual [r2_bmod][rows_per_batch][C] where
 *                 pixel x = (row % out_w) >= r2_xmin) -> out0 [M][C]; gn_tot [B][C][BC_GN_TOT_WORDS]: the statistics totals the
 *                 per-channel (sum, sum of squares) of the fp16 output rows are added to (NULL: none).  With out1 != NULL (BlobNet) the block
 *                 output also goes through the zero-conv: out1 = (W out0 + b) * alpha * alpha_dev[*alpha_idx (* bstride + image)].
 *   BC_CHAIN_OUT_FF + BC_CHAIN_OUT_TAIL  the same block end as two launches with the feed-forward's hidden chunks split over `nsplit`
 *                 workgroups per row block (a 64-row workgroup of the one-launch form does the whole feed-forward on one CU: at 640
 *                 channels that is 120 us of MFMA work; with few row blocks most of the chip idles meanwhile).  OUT_FF (x, res as for
 *                 OUT; grid = row blocks x nsplit): to_out + res -> LayerNorm -> its NCH / nsplit hidden chunks -> part[z][M][C] fp32
 *                 (slice 0 on top of the residual stream).  OUT_TAIL (res2, r2, out0, out1, gn_tot as for OUT): sum of the nsplit
 *                 slices + ff.net.2 bias -> proj_out + res2 (+ r2) [-> zero-conv].  nsplit divides 4C / 128.
 *   BC_CHAIN_OUT_FFP + bc_rowchain_sum  (round 5) the block end with the reduction moved BEHIND proj_out [and the zero-conv]: both are
 *                 linear, so every one of the `nsplit` workgroups of a row block runs to_out + res -> LayerNorm -> its hidden chunks ->
 *                 proj_out [-> zero-conv x alpha] on its own partial sum (slice 0 carries the residual stream, the biases, res2 and r2)
 *                 and writes an fp16 partial OUTPUT: part = bc_half [nsplit][M][C] (+ [nsplit][M][C] zero-conv partials behind it when
 *                 out1 != NULL; out1 is only the BlobNet flag here).  OUT_TAIL's GEMMs thus run on nsplit times as many CUs, its fp32
 *                 partial sums (nsplit x 160 KB read per workgroup) are gone, and the slices are slice-major over the XCDs (an XCD's L2
 *                 fetches 1 / nsplit of the feed-forward weights).  bc_rowchain_sum adds the partial outputs in slice order (fp32, rounded
 *                 once: bit-reproducible) into out0 [, out1] and adds the output's GroupNorm statistics to gn_tot.
 * `wstream` / `vec`: the block's weights packed by blobctrl_amd/weights.py:pack_rowchain (per-wave fragment streams in
 * consumption order, for OUT_FF one set per slice; bc_rowchain_stream_frags(channels, kind, blobnet, nsplit) gives the length).  M % rows_per_batch == 0,
 * rows_per_batch % 64 == 0.  Every workgroup streams the block's whole weight set (4.1 MB at 320 channels, 16.4 MB at 640): worth it
 * from a few dozen row blocks upwards (the engine takes the 640-channel form from 64 row blocks).
 * --------------------------------------------------------------------------------------------------------------- */
enum { BC_CHAIN_IN = 0, BC_CHAIN_MID = 1, BC_CHAIN_OUT = 2, BC_CHAIN_OUT_FF = 3, BC_CHAIN_OUT_TAIL = 4, BC_CHAIN_MIDX = 5, BC_CHAIN_OUT_FFP = 6 };
int bc_rowchain_supported(int channels, int M, int rows_per_batch);
long long bc_rowchain_stream_frags(int channels, int kind, int blobnet, int nsplit);
int bc_rowchain(int kind, int channels, int M, int rows_per_batch, const bc_half* x, const float* affine,
                const unsigned long long* gn_in, const float* gn_gamma, const float* gn_beta, int gn_groups, float gn_eps, const bc_half* res,
                const bc_half* res2, const bc_half* r2, int r2_xmin, int r2_bmod, int out_w, const bc_half* wstream,
                const float* vec, bc_half* out0, bc_half* out1, bc_half* out2, int ldvt, unsigned long long* gn_tot, float ln_eps,
                float alpha, const float* alpha_dev, const int* alpha_idx, int alpha_bstride, float* part, int nsplit, bc_stream stream);
int bc_rowchain_sum(int channels, int M, int rows_per_batch, const bc_half* part, int nsplit, bc_half* out0, unsigned long long* gn_tot,
                    bc_half* out1, bc_stream stream);
/* BC_CHAIN_MIDX: BC_CHAIN_MID with the block's cross-attention behind to_q, in the same launch (diffusers/src/diffusers/models/
 * attention.py:491-510 norm2 -> attn2, attention_processor.py:2191-2224 scaled-dot-product attention over the encoder tokens): x =
 * attn1 output, res = h0: to_out + res -> out0 = h1; LayerNorm2 -> to_q -> softmax(q K^T * attn_scale) V per head (8 heads) -> out1 =
 * attn2's attention output rows [M][C] (what BC_CHAIN_OUT takes as x).  The query rows stay in the workgroup; K and V^T of the
 * workgroup's image are read as per-wave fragment streams `kvstream` that bc_rowchain_pack_kv lays out once per edit from the
 * projected context (K rows [B * T][ldk], V^T [B][C][ldvt]; T <= 80 tokens; bc_rowchain_kv_frags(channels) * 64 * 16 bytes per image
 * and wave, channels / 80 waves).  `wstream` / `vec` are BC_CHAIN_MID's. */
long long bc_rowchain_kv_frags(int channels);
int bc_rowchain_pack_kv(const bc_half* k, int ldk, const bc_half* vt, int ldvt, int B, int T, int channels, bc_half* out, bc_stream stream);

/* Cross-attention with the prompt folded into the weights, once per edit (round 5; attention.py:504-510 attn2 of a BasicTransformerBlock,
 * attention_processor.py:2191-2224): K_h = (ctx W_k)_h and V_h = (ctx W_v)_h of an image are fixed over the edit, so
 *     softmax(scale LN(x) W_q,h^T K_h^T) V_h W_o,h^T  =  softmax(LN(x) QK_h^T) VO_h^T,   QK_h = scale K_h W_q,h  [T][C],   VO_h = W_o,h V_h^T  [C][T]:
 * `attn2.to_q` + the 77-key attention + `attn2.to_out` (3 launches, 4 C^2 + 4 T C flops per row) become two projections with per-image
 * weights (2 launches) - bc_gemm on BC_TILE_GW64x128 with sm_group = 128 / sm_valid = T / sm_keep = 80 and w_bstride / vec_bstride (one
 * 64 x 128 workgroup per head: 128 QK rows of which the first T are keys, the rest zero), then bc_gemm on BC_TILE_GW64x128 with w_bstride
 * over the 80 kept probabilities per head (column (h, j) = 80 h + j).
 *   k   [B][T][ldk] projected context rows, vt [B][C][ldvt] its V^T (what the attention launch read), D = C / heads (% 8 == 0), T <= 80;
 *   wq  [C][C] = attn2.to_q.weight * diag(gamma of norm2) (fp16), bq [C] = to_q.weight . beta (fp32), wo [C][C] = attn2.to_out.0.weight;
 *   wqk [B][bc_gemm_wreg_stream_elems(128 heads, C)]: the GW64x128 stream of QK, row (h, j) = 128 h + j (rows of j >= T zero: the
 *       rows j >= 80 are never written - the caller provides them zeroed);
 *   qk_colsum / qk_bias [B][128 heads] fp32: sum_c of the fp16-ROUNDED QK row (BcGemm.ln_colsum) / scale K_h . bq_h (BcGemm.bias);
 *   vwo [B][bc_gemm_wreg_stream_elems(C, 80 heads)]: the GW64x128 stream of VO (columns of keys >= T zero).
 * fp32 accumulation in a fixed order: the same bytes for the same inputs on every call. */
int bc_ctx_fold(const bc_half* k, int ldk, const bc_half* vt, int ldvt, int B, int T, int channels, int heads, float scale,
                const bc_half* wq, const float* bq, const bc_half* wo, bc_half* wqk, float* qk_colsum, float* qk_bias, bc_half* vwo,
                bc_stream stream);
int bc_rowchain_midx(int channels, int M, int rows_per_batch, const bc_half* x, const bc_half* res, const bc_half* wstream, const float* vec,
                     const bc_half* kvstream, int n_ctx, float attn_scale, bc_half* out0, bc_half* out1, float ln_eps, bc_stream stream);

/* Layout helpers at the nn.Module boundary (NCHW <-> token-major NHWC, fp32/fp16). */
int bc_nchw_to_nhwc_f16(const void* src, int src_is_f32, int B, int C, int HW, int Cpad, bc_half* dst, bc_stream stream);
int bc_nhwc_to_nchw(const bc_half* src, int B, int C, int HW, int ldsrc, void* dst, int dst_is_f32, bc_stream stream);
/* Concatenate a learned token in front of patch tokens and add position embeddings (DINOv2 embeddings). */
int bc_add_cls_pos(const bc_half* patches, const float* cls, const float* pos, int B, int T, int D, bc_half* out,
                   bc_stream stream);
/* im2col for the DINOv2 14x14/s14 patch embedding: pixels fp32 [B][3][H][W] -> [B*gh*gw][Kpad] fp16 (c,ky,kx order). */
int bc_patchify(const float* pixels, int B, int H, int W, int patch, int Kpad, bc_half* out, bc_stream stream);

/* ---------------------------------------------------------------------------------------------------------------
 * hipGraph helpers: capture everything enqueued on `stream` between begin/end, replay with bc_graph_launch.
 * --------------------------------------------------------------------------------------------------------------- */
int bc_graph_begin(bc_stream stream);
int bc_graph_end(bc_stream stream, void** graph_exec_out);
int bc_graph_launch(void* graph_exec, bc_stream stream);
int bc_graph_destroy(void* graph_exec);

/* Cross-stream dependencies (BlobNet and the UNet run concurrently on two streams; inside a captured graph these become
 * DAG edges).  bc_event_create_sync makes a timing-disabled event. */
int bc_event_create_sync(void** ev);
int bc_stream_wait_event(bc_stream stream, void* ev);

/* HIP-event timing on an arbitrary stream (torch.cuda.Event only sees torch's current stream). */
int bc_event_create(void** ev);
int bc_event_record(void* ev, bc_stream stream);
int bc_event_elapsed_ms(void* start, void* stop, float* ms);   /* synchronises on `stop` */
int bc_event_destroy(void* ev);

/* ---------------------------------------------------------------------------------------------------------------
 * Plan runtime (plan.hip): the loop body of StableDiffusionBlobNetPipeline.__call__ (pipeline_blobnet.py:1025-1123: BlobNet forward,
 * patched UNet forward, crop + CFG, scheduler step) - or any other forward pass of this library - as a STATIC list of launches that
 * is compiled once per (batch, canvas, steps) configuration and replayed without Python in the loop.
 *   build   : bc_plan_create -> bc_plan_segment ("prologue", "step_active", "step_inactive", ...) -> bc_plan_add_gemm / bc_plan_add_op
 *             (stream_id 0 = main stream, 1.. = side streams; BC_OP_SIGNAL / BC_OP_WAIT on plan events express the dependencies
 *             between them and become DAG edges when captured) -> bc_plan_set_slab (split-K scratch per stream id)
 *   run     : bc_step replays one segment on the caller's streams (its captured hipGraph when bc_plan_capture was called, else
 *             eagerly); bc_plan_capture_loop captures a whole sequence of segments (the N-step denoise loop) into ONE graph, replayed
 *             with bc_graph_launch; per-step scalars come from device tables indexed by a device-side step counter, so the replay
 *             needs no host arguments.
 *   persist : bc_plan_save writes a relocatable plan (every pointer = (buffer, offset) of the caller-declared buffer table, with
 *             the initial contents of weights / tables); bc_plan_load allocates one arena, uploads, patches the records; a plain C
 *             host then runs an edit: bc_plan_buffer to find the I/O buffers, bc_step / bc_plan_capture_loop (tests/c/plan_edit.c).
 * streams == NULL / nstreams == 0 makes the plan use three streams of its own.
 * --------------------------------------------------------------------------------------------------------------- */
typedef struct BcPlan BcPlan;
enum { BC_OP_GEMM = 0, BC_OP_GN_STATS = 1, BC_OP_GN_FINALIZE = 2, BC_OP_GN_APPLY_FUSED = 3, BC_OP_GN_APPLY = 4, BC_OP_LAYERNORM = 5,
       BC_OP_ATTENTION = 6, BC_OP_ATTENTION_CAUSAL = 7, BC_OP_ASSEMBLE_INPUT = 8, BC_OP_TIMESTEP_EMBEDDING = 9,
       BC_OP_TIMESTEP_EMBEDDING_TABLE = 10, BC_OP_CFG_SCHEDULER_STEP = 11, BC_OP_EMBED_TOKENS = 12, BC_OP_SOFTMAX_ROWS = 13,
       BC_OP_PATCHIFY = 14, BC_OP_ADD_CLS_POS = 15, BC_OP_SILU = 16, BC_OP_NCHW_TO_NHWC_F16 = 17, BC_OP_NHWC_TO_NCHW = 18,
       BC_OP_GAUSSIAN_SAMPLE = 19, BC_OP_SIGNAL = 20 /* arg: event id */, BC_OP_WAIT = 21 /* arg: event id */, BC_OP_ROWCHAIN = 22, BC_OP_ASSEMBLE_IM2COL = 23,
       BC_OP_MEMSET_ZERO = 24, BC_OP_ROWCHAIN_MIDX = 25, BC_OP_ROWCHAIN_PACK_KV = 26, BC_OP_ROWCHAIN_SUM = 27, BC_OP_CTX_FOLD = 28, BC_OP_DUP_HALVES = 29, BC_OP_COUNT = 30 };
typedef struct BcPlanBuffer {
    const char* name;        /* "" for anonymous workspace; named buffers are found again with bc_plan_buffer */
    const void* address;     /* the address the launch records were built against */
    long long   bytes;
    const void* host_data;   /* initial contents to store in the file (weights, tables) or NULL = zero-filled workspace */
} BcPlanBuffer;

int bc_plan_create(BcPlan** out);
int bc_plan_destroy(BcPlan* plan);
int bc_plan_segment(BcPlan* plan, const char* name);                       /* -> segment id, < 0 on error */
int bc_plan_find_segment(BcPlan* plan, const char* name);                  /* -> segment id or -1 */
int bc_plan_new_event(BcPlan* plan);                                       /* -> event id, < 0 on error */
int bc_plan_add_gemm(BcPlan* plan, int seg, int stream_id, const BcGemm* g);             /* -> launch index (< 0 on error); `slab` is taken from
                                                                                          * bc_plan_set_slab(stream_id) at run time */
/* args: the entry point's arguments in order WITHOUT the stream, one 64-bit word each (pointers as addresses, ints sign-extended,
 * floats as their 32-bit pattern); BC_OP_SIGNAL / BC_OP_WAIT take the event id. -> launch index, < 0 on error */
int bc_plan_add_op(BcPlan* plan, int seg, int stream_id, int op, const uint64_t* args, int nargs);
int bc_plan_set_slab(BcPlan* plan, int stream_id, float* slab);
int bc_plan_enable(BcPlan* plan, int seg, int index, int enabled);         /* diagnostics: skip / restore one launch (ablation probes) */
int bc_plan_num_launches(BcPlan* plan, int seg);
int bc_step(BcPlan* plan, int seg, const bc_stream* streams, int nstreams);
int bc_plan_capture(BcPlan* plan, int seg, const bc_stream* streams, int nstreams);
int bc_plan_release(BcPlan* plan, int seg);                                /* drop the segment's graph (back to eager replay) */
int bc_plan_capture_loop(BcPlan* plan, const int* seg_sequence, int n, const bc_stream* streams, int nstreams, void** graph_exec_out);
int bc_plan_run_timed(BcPlan* plan, int seg, bc_stream stream, float* ms_out /* [bc_plan_num_launches] */);
/* same, with a split-K GEMM's time divided into its main kernel and its reducer (ms_reduce[i] = 0 when launch i has none) */
int bc_plan_run_timed_kernels(BcPlan* plan, int seg, bc_stream stream, float* ms_main, float* ms_reduce);
/* Concurrent eager replay of a segment on its own streams with a timing event behind each launch listed in `marks` (ascending launch
 * indices), recorded on that launch's stream: ms_out[k] = milliseconds from the start of the replay to that event (diagnostics:
 * where the two queues really are inside an overlapped step). */
int bc_plan_run_marked(BcPlan* plan, int seg, const bc_stream* streams, int nstreams, const int* marks, int nmarks, float* ms_out);
int bc_plan_save(BcPlan* plan, const char* path, const BcPlanBuffer* buffers, int nbuffers);
int bc_plan_load(const char* path, BcPlan** out);
int bc_plan_buffer(BcPlan* plan, const char* name, void** ptr, long long* bytes);

#ifdef __cplusplus
}
#endif
#endif
