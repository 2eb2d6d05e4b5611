/*
 * dsc_hip.h - C ABI of libdsc_hip.so: the MI355X (gfx950) hot path of the spatially-controlled denoising
 * step of duongve13112002/DiffusionSpatialControl.
 *
 * Conventions (last row):
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless it says host;
 *   - the caller owns every buffer, including the workspace; nothing is allocated or freed inside;
 *   - every entry is asynchronous on the `stream` it is given (a hipStream_t passed as void*), re-entrant for
 *     distinct streams/workspaces, and safe to capture into a hipGraph (no sync, no malloc, no memcpy);
 *   - the return value is a status: DSC_OK (0) or a negative DSC_ERR_* code; no exception crosses the ABI;
 *   - strides are in ELEMENTS; the innermost (head-dim / channel) stride is always 1.
 *
 * Each entry replaces torch-op sequences issued by the reference's Python; the reference has no FFI of its
 * own (it is pure Python), so the "interface each one replaces" is the Python function cited beside it.
 * Paths below are relative to /root/reference/source/.
 */
#ifndef DSC_HIP_H
#define DSC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSC_OK                0
#define DSC_ERR_BAD_ARG      -1   /* null pointer, non-positive size, Bw does not divide Bc*H, ...        */
#define DSC_ERR_UNSUPPORTED  -2   /* head dim / dtype / alignment outside what the kernels are built for */
#define DSC_ERR_WORKSPACE    -3   /* workspace missing or smaller than the *_workspace_bytes() answer    */
#define DSC_ERR_LAUNCH       -4   /* hipLaunchKernel returned an error (hipGetLastError is left set)     */

/* dtype tags */
#define DSC_F16 0
/* flags for dsc_region_xattn_fwd */
#define DSC_FLAG_REF_FP16_ROUNDING 1u  /* round where the reference's fp16 tensors round: scores
                                          (attention_modify.py:90), std (0-dim fp16), the in-place bias add (:97)
                                          and the softmax output (:101) */
#define DSC_FLAG_REUSE_STATS       4u  /* measurement aid: skip the statistics launch and reuse the partial sums the
                                          previous identical call left in `workspace` (times the forward kernel alone) */
#define DSC_FLAG_ROWS_PADDED     256u  /* dsc_region_xattn_fwd_packed only: `region_rows` is [n_rows][100] fp32 (row stride 100
                                          floats = the kernel's LDS table, zeros beyond column S, 16-byte aligned) instead
                                          of [n_rows][S]: the table goes to LDS as a flat 16-byte copy.  S <= 96 only. */
#define DSC_FLAG_BIAS_IS_FINAL     2u  /* `region` already holds the additive bias (a custom weight_func was
                                          evaluated by the caller): add it as is, skip the statistics pass */

/* ABI version of this header; bumped on any signature change. */
int dsc_abi_version(void);
/* Static string naming the code-object target the library was built for ("gfx950"). */
const char* dsc_target_arch(void);
/* Diagnostic builds only: a 2-KiB device buffer that receives in-kernel clock stamps of workgroup 0 when a
 * call carries debug flag 32 (tools/mb_xattn.py); NULL (default) disables it.  No output tensor is ever touched. */
void dsc_debug_set_stamp_buffer(void* device_buffer_2KiB);
/* Tuning aid: force a tiling variant of the flash self-attention kernel (0 = automatic choice, 1 = one query tile per
 * wave, 2 = two query tiles per wave where the head dim allows it). */
void dsc_debug_set_self_attn_variant(int variant);
/* Diagnostic: 6 x u64 per-segment cycle sums of workgroup 0 / wave 0 of the flash self-attention kernel (NULL = off). */
void dsc_debug_set_self_attn_stamps(void* device_buffer_64B);
void dsc_debug_set_gemm_stamps(void* device_buffer);   /* 8 x int64 per workgroup of the next gemm_tn_f16 launches (tools/stamps_gemm.py) */
void dsc_debug_set_self_attn_stamp_wave(int wave);    /* which wave of workgroup 0 writes them (default 0) */
/* Human-readable text for a status code. */
const char* dsc_status_string(int status);

/*
 * Region-biased cross-attention forward - replaces modules/attention_modify.py:74-103
 * `scaled_dot_product_attention_regionstate` with weight_func = `w * sigma * qk.std()` (app.py:1004), i.e.
 * rows 2-10 of SURVEY.md 2b, and (with region == NULL) the plain cross-attention SDPA call at :483-485.
 *
 *   out[b,l,h,:] = softmax_s( scale*q[b,l,h,:].k[b,s,h,:] + region[bw(b,h),l,s] * sigma * std_g ) . v[b,s,h,:]
 *
 *   q, out : Bc x L x H x d   addressed as base + b*sb + l*sl + h*sh + i   (q_strides / o_strides = {sb, sl, sh})
 *   k, v   : Bc x S x H x d   addressed as base + b*sb + s*ss + h*sh + i   (k_strides / v_strides = {sb, ss, sh})
 *            so both the processor's [Bc, L, H*d] projection outputs (sl = H*d, sh = d) and a transposed
 *            [Bc, H, L, d] tensor (sh = L*d, sl = d) are consumed in place, and `out` is written directly in the
 *            [Bc, L, H*d] layout `to_out[0]` reads (attention_modify.py:487).
 *   region : fp32 dense [Bw, L, S], Bw divides Bc*H; flattened score row bh = b*H + h takes table row
 *            bh / (Bc*H/Bw)  (torch.repeat_interleave at :96-99).  NULL = no bias (std pass skipped).
 *   std_g  : unbiased (N-1) standard deviation of scale*q.k^T over std group g = b % n_std_groups, all heads,
 *            all L x S scores of the group's rows (:93-95: ONE global std when n_std_groups == 1).
 *   sigma  : `sigma_dev` (device fp32 scalar) if non-NULL, else `sigma_host`.  A device scalar lets a captured
 *            graph be replayed for every step without a host sync (the reference syncs, model_k_diffusion.py:1115).
 *   scale  : <= 0 means 1/sqrt(d)  (attention_modify.py:77; attn.scale is ignored there).
 *
 * Requirements: dtype DSC_F16; d % 8 == 0 and d <= 160; S <= 96; all strides % 8 == 0; pointers 16-byte aligned.
 * Workspace: dsc_region_xattn_workspace_bytes(); contents need no initialisation and carry nothing across calls.
 */
size_t dsc_region_xattn_workspace_bytes(int Bc, int H, int L, int S, int d, int n_std_groups);

int dsc_region_xattn_fwd(const void* q, const void* k, const void* v, void* out,
                         const float* region,
                         int Bc, int H, int L, int S, int d, int Bw, int n_std_groups,
                         const int64_t q_strides[3], const int64_t k_strides[3],
                         const int64_t v_strides[3], const int64_t o_strides[3],
                         float sigma_host, const float* sigma_dev, float scale,
                         int dtype, unsigned flags,
                         void* workspace, size_t workspace_bytes, void* stream);

/*
 * Statistics only: writes std_out[g] (fp32, n_std_groups values) = the std the call above would use.
 * Exposes `qk.std()` of app.py:1004 so that a caller-supplied weight_func that is not the default one can be
 * evaluated on the host side without materialising the scores more than once.
 */
int dsc_region_xattn_std(const void* q, const void* k,
                         int Bc, int H, int L, int S, int d, int n_std_groups,
                         const int64_t q_strides[3], const int64_t k_strides[3], float scale,
                         int dtype, unsigned flags, float* std_out,
                         void* workspace, size_t workspace_bytes, void* stream);
/*
 * The same with an additive attention mask inside the statistics - reference attention_modify.py:85-95 adds `attn_mask`
 * (float) to the scaled scores BEFORE weight_func sees them, so the std is over scale * q.k^T + mask; likewise
 * `get_attention_scores(attn, query, key, attention_mask)` (:39-70, baddbmm with beta = 1) on the AttnProcessor path
 * (:144,166).  mask: fp32, element (b * H + h, l, s) at mask[bh * mask_strides[0] + l * mask_strides[1] + s]; a stride of 0
 * broadcasts that dimension (an [L, S] mask: strides {0, S}; a [B*H, 1, S] one: {S, 0}).  NULL = dsc_region_xattn_std.
 */
int dsc_region_xattn_std_masked(const void* q, const void* k,
                                int Bc, int H, int L, int S, int d, int n_std_groups,
                                const int64_t q_strides[3], const int64_t k_strides[3], float scale,
                                int dtype, unsigned flags, const float* mask, const int64_t mask_strides[2],
                                float* std_out, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Prepared-operand path of the region cross-attention (the one the pipeline uses).
 *
 * dsc_xattn_kv_pack: the text keys / values of a cross-attention layer are step-invariant (the reference re-projects
 * them in every layer at every step, attention_modify.py:465-466); they are packed once per generation into the
 * register image the MFMAs consume (zero-padded K fragments, V^T fragments in the chained product's k order).
 * `packed` holds dsc_xattn_kv_pack_bytes() bytes, 16-byte aligned; k / v addressed as in dsc_region_xattn_fwd.
 *
 * dsc_region_xattn_fwd_packed: same result as dsc_region_xattn_fwd, with
 *   packed_kv   : the image written by dsc_xattn_kv_pack for the same (Bc, H, S, d);
 *   region_ids  : uint16 [Bw, L] - index of each table row into region_rows, or NULL for no bias;
 *   region_rows : fp32 [n_rows, S] - the DISTINCT rows of the dense table (n_rows <= 32; the reference's tables have
 *                 at most 2^regions distinct rows, encode_region_map_function.py:49-69).
 * The forward kernel DMAs the image into LDS (global_load_lds) and folds sigma * std into an LDS bias table once per
 * workgroup.  Flags: DSC_FLAG_REF_FP16_ROUNDING, DSC_FLAG_REUSE_STATS.  Workspace as dsc_region_xattn_workspace_bytes().
 * S <= 96: one image per (b, h).  96 < S <= 384 (prompts of several 77-token chunks, encoder_prompt_modify.py:691-812): one
 * image per (b, h, 96-key chunk); the statistics run over all chunks, the forward pass walks them with an online softmax;
 * fp32 scores only (DSC_FLAG_REF_FP16_ROUNDING -> DSC_ERR_UNSUPPORTED).
 */
size_t dsc_xattn_kv_pack_bytes(int Bc, int H, int S, int d);
int dsc_xattn_kv_pack(const void* k, const void* v, void* packed, int Bc, int H, int S, int d,
                      const int64_t k_strides[3], const int64_t v_strides[3], int dtype, void* stream);
int dsc_region_xattn_fwd_packed(const void* q, const void* packed_kv, void* out,
                                const uint16_t* region_ids, const float* region_rows, int n_rows,
                                int Bc, int H, int L, int S, int d, int Bw, int n_std_groups,
                                const int64_t q_strides[3], const int64_t o_strides[3],
                                float sigma_host, const float* sigma_dev, float scale,
                                int dtype, unsigned flags,
                                void* workspace, size_t workspace_bytes, void* stream);

/*
 * Flash self-attention forward - replaces `F.scaled_dot_product_attention(query, key, value)` on the self-attention
 * branch of the processors (modules/attention_modify.py:483-485) and `attn.get_attention_scores` + bmm (:187-188).
 *   out[b,l,h,:] = softmax_s(scale * q[b,l,h,:].k[b,s,h,:]) . v[b,s,h,:]     (no mask, no dropout)
 * Same addressing as dsc_region_xattn_fwd: q/out [Bc, L, H, d] and k/v [Bc, S, H, d] through {sb, sl|ss, sh} element
 * strides, so the three operands can be strided views of ONE fused [Bc, L, 3*H*d] QKV projection.  The L x S scores
 * are never written to memory (online softmax over 64-key tiles).  fp16, d % 8 == 0, d <= 160, strides % 8 == 0.
 * scale <= 0 means 1/sqrt(d).  No workspace.
 */
int dsc_self_attn_fwd(const void* q, const void* k, const void* v, void* out,
                      int Bc, int H, int L, int S, int d,
                      const int64_t q_strides[3], const int64_t k_strides[3],
                      const int64_t v_strides[3], const int64_t o_strides[3],
                      float scale, int dtype, void* stream);

/*
 * Fused sampler step for the k-diffusion DPM++ 2M loop with classifier-free guidance - replaces, per step,
 * `torch.cat([x]*2)` + `input * c_in` (model_k_diffusion.py:1097, external_k_diffusion.py:111),
 * `input + eps * c_out` (:114), the CFG combine (model_k_diffusion.py:1162-1166) and the 3-5 elementwise launches
 * of k_diffusion.sampling.sample_dpmpp_2m (un-vendored), by ONE launch:
 *
 *   D      = x - sigma * (eps_u + guidance * (eps_c - eps_u))          eps = [eps_u(n_img rows); eps_c(n_img rows)]
 *   x      = a * x + b * D + c * old ;   old = D                       (in place; fp32 math, one fp16 rounding each)
 *   x_in   = [x; x] * c_in_next ;  t_buf[0 .. 2 n_img) = t_next ;  sigma_buf[0] = sigma_next
 *
 * x, old: fp16 [n_img, chw]; eps, x_in: fp16 [2 n_img, chw]; t_buf fp32 [2 n_img]; sigma_buf fp32 [1].
 * dsc_prepare_unet_input does only the last line (before the first step).  chw % 8 == 0, 16-byte aligned pointers.
 */
/* row_src (optional, NULL = none): row_halfs fp16 values copied to each of the row_copies rows of row_dst in the same launch -
 * the pipeline keeps the per-ResNet time-embedding projections of ALL steps in a table computed once per schedule (they depend on
 * the timestep only, reference u_net_condition_modify.py:1040-1060 + diffusers ResnetBlock2D.time_emb_proj) and hands the
 * coming step's row to the static buffer the captured UNet step reads: three launches per step fewer.  row_halfs % 8 == 0. */
int dsc_prepare_unet_input(const void* x, float c_in, float t, float sigma,
                           void* x_in, float* t_buf, float* sigma_buf, int n_img, int chw, int dtype,
                           const void* row_src, void* row_dst, int row_halfs, int row_copies, void* stream);
int dsc_cfg_dpmpp2m_step(void* x, const void* eps, void* old, float sigma, float guidance,
                         float a, float b, float c, float c_in_next, float t_next, float sigma_next,
                         void* x_in, float* t_buf, float* sigma_buf, int n_img, int chw, int dtype,
                         const void* row_src, void* row_dst, int row_halfs, int row_copies, void* stream);
/* out = a*x + b*denoised + c*old  (old may be NULL when c == 0): the sampler update alone, for callers that keep
 * the reference's `sampler(model_fn, x, sigmas=...)` control flow.  n elements, n % 8 == 0. */
int dsc_dpmpp2m_update(const void* x, const void* denoised, const void* old, float a, float b, float c,
                       void* out, int64_t n, int dtype, void* stream);

/*
 * GroupNorm (+ optional SiLU) over NCHW fp16 - replaces `GroupNorm(32, eps)` -> `SiLU` pairs of the UNet
 * (u_net_condition_modify.py:465-470,1304-1306 and diffusers ResnetBlock2D / Transformer2DModel norms; SURVEY.md 2b
 * row 12).  y[b,c,:,:] = act((x - mean_bg) * rstd_bg * gamma[c] + beta[c]), statistics in fp32/fp64 over the
 * (C/groups) * hw elements of group g of row b (biased variance, as torch).  Two launches (partial sums, apply);
 * workspace from dsc_groupnorm_workspace_bytes(), no initialisation needed.  gamma/beta fp16 [C].
 * Requirements: C % groups == 0, 16-byte aligned x / y (hw % 8 != 0 takes a scalar-load path).
 */
size_t dsc_groupnorm_workspace_bytes(int B, int C, int hw, int groups);
int dsc_groupnorm_silu(const void* x, void* y, const void* gamma, const void* beta,
                       int B, int C, int hw, int groups, float eps, int apply_silu, int dtype,
                       void* workspace, size_t workspace_bytes, void* stream);

/*
 * GroupNorm (+ optional SiLU) over channels-last (NHWC) fp16 - the layout the UNet keeps its activations in so that
 * MIOpen's NHWC implicit-GEMM convolutions and the transformer's token-major GEMMs need no transposes.
 *   x, y  : [B, hw, C] (= a channels_last [B, C, h, w] tensor), C % 8 == 0, C/8 <= 512
 *   add   : optional fp16 [B, C] added to x before the statistics and the normalisation - fuses the ResNet block's
 *           `hidden + time_emb_proj(silu(temb))[:, :, None, None]` (diffusers ResnetBlock2D) into the norm that follows
 *   y[b,p,c] = act(((x[b,p,c] + add[b,c]) - mean_bg) * rstd_bg * gamma[c] + beta[c])
 * Every access is a coalesced 16-byte vector; two launches (per-chunk partial sums, apply); bit-reproducible.
 */
size_t dsc_groupnorm_nhwc_workspace_bytes(int B, int C, int hw, int groups);
/* diagnostics: 0 = choose the kernel from the shape, 2 = never a single-launch kernel (statistics + apply), 3 = also allow the
 * 1024-thread single-launch kernel (measured slower; kept for tools/mb_gn.py), 4 = 2 in the old three-launch form (fine row
 * chunks, separate finalize launch); 10 / 11 = statistics workgroups without / with channel slabs (default 11) */
void dsc_debug_set_gn_mode(int mode);
int dsc_groupnorm_silu_nhwc(const void* x, void* y, const void* gamma, const void* beta,
                            const void* add, int64_t add_row_stride,   /* elements between rows of `add` (>= C) */
                            int B, int C, int hw, int groups, float eps, int apply_silu, int dtype,
                            void* workspace, size_t workspace_bytes, void* stream);
/* The same over the channel concatenation [x1 | x2] that the caller never materialised - the up blocks'
 * `torch.cat([hidden_states, res_hidden_states], dim=1)` in front of every ResNet block (diffusers UpBlock2D /
 * CrossAttnUpBlock2D, reached from reference u_net_condition_modify.py:1274-1302): x1 [B, hw, C1], x2 [B, hw, C - C1],
 * C1 % 8 == 0.  The pass that reads the two sources for the statistics also writes the concatenation to `cat`
 * [B, hw, C] (the block's 1x1 shortcut reads it), so the separate concatenation kernel and one read of the tensor go away.
 * y as above.  cat must not alias x1, x2 or y.  Same workspace as dsc_groupnorm_silu_nhwc for (B, C, hw, groups). */
int dsc_groupnorm_silu_nhwc_cat(const void* x1, const void* x2, int C1, void* cat, void* y, const void* gamma,
                                const void* beta, const void* add, int64_t add_row_stride, int B, int C, int hw,
                                int groups, float eps, int apply_silu, int dtype, void* workspace, size_t workspace_bytes,
                                void* stream);

/* out[r, c] = a[r, c] + b[r, c] + bias[c] over fp16 [rows, C] (channels-last residual add with the convolution's
 * bias folded in: the ResNet block's `x + conv2(h)` where conv2 ran without its bias).  bias may be NULL. C % 8 == 0. */
int dsc_add_bias_residual(const void* a, const void* b, const void* bias, void* out, int64_t rows, int C,
                          int dtype, void* stream);

/*
 * Token-major fp16 linear with fused epilogues - replaces `F.linear` (+ the separate residual add / GEGLU kernels) for
 * the UNet's `to_q/to_k/to_v/to_out`, feed-forward and 1x1 projection layers (diffusers BasicTransformerBlock /
 * Transformer2DModel / ResnetBlock2D shortcut):
 *   geglu == 0:  out[m, n] = sum_k x[m,k] w[n,k] (+ bias[n]) (+ residual[m,n]),            out is [M, N]
 *   geglu == 1:  out[m, j] = (acc[m,j] + bias[j]) * gelu(acc[m,N/2+j] + bias[N/2+j]),      out is [M, N/2]
 * x [M, K] with row stride ldx, w [N, K] contiguous (torch Linear layout), residual / out with row strides ldr / ldo.
 * Requirements: fp16, K % 64 == 0, N % 64 == 0, strides % 8 == 0, 16-byte aligned pointers; geglu needs bias, no residual.
 * A residual of FEWER rows than M - the residual stream of layers that ran once per image in front of the first cross-attention,
 * added to a result that has a row per classifier-free-guidance branch (the reference repeats nothing there: its batch is
 * already doubled) - is passed as  ldr = row stride | (R << 32)  with R = its row count: row m adds residual row m % R.
 * R % 128 == 0, M % R == 0; same encoding in dsc_linear_ln_f16 and dsc_linear_gn_f16.  High bits 0: a residual of M rows.
 */
int dsc_linear_f16(const void* x, const void* w, const void* bias, const void* residual, void* out,
                   int64_t M, int N, int K, int64_t ldx, int64_t ldr, int64_t ldo, int geglu, int dtype, void* stream);

/*
 * 3x3 / stride 1 / pad 1 convolution over channels-last fp16 - replaces `F.conv2d` (MIOpen) for the ResnetBlock2D
 * conv1/conv2 and the Upsample2D conv of the UNet that reference source/modules/u_net_condition_modify.py assembles
 * from diffusers blocks:
 *   out[b,y,x,n] = sum_{dy,dx,c} x[b,y+dy-1,x+dx-1,c] * w[n,dy,dx,c]  (+ bias[n]) (+ residual[b,y,x,n]),  zero padding
 * x [B,H,W,Cin] with pixel stride ldx, w [Cout,3,3,Cin] contiguous (a torch conv weight in channels_last memory
 * format), residual / out [B,H,W,Cout] with pixel strides ldr / ldo.  fp32 accumulation, one fp16 rounding.
 * resample == DSC_CONV_UPSAMPLE2X: x is [B,H/2,W/2,Cin] and the convolution reads it through a nearest-neighbour 2x
 * upsampling (diffusers Upsample2D = F.interpolate(scale_factor=2, mode="nearest") + conv) without materialising the image.
 * resample == DSC_CONV_STRIDE2: only the even pixels are stored, out is [B,H/2,W/2,Cout] - the stride-2 / pad-1 convolution of
 * diffusers Downsample2D (3 per UNet step).  The taps still run at H x W (4x the necessary MFMA work); it is nevertheless
 * faster than MIOpen's stride-2 kernels here (24-27 us vs 31-37 us) and, unlike their atomic split-K, bit-reproducible.
 * out_nchw != 0: out is [B,Cout,H,W] (channel-major; the UNet's 4-channel conv_out hands its result back in the sampler's
 * layout).  Cout need not be a multiple of 64: a ragged last channel tile reads zero weight rows through the buffer bounds.
 * splits: number of input-channel ranges accumulated by separate workgroups (0 = chosen from the shape); splits > 1
 * needs `workspace` (dsc_conv3x3_workspace_bytes) and sums the partials in range order: bit-reproducible.
 * Supported (dsc_conv3x3_supported): Cin % 64 == 0, strides % 8 == 0 (when Cout % 8 == 0), 16-byte aligned pointers; any
 * H, W >= 1 (sides that are not multiples of the 8 x 16 / 8 x 8 pixel tile - 12 x 12, the lowest level of a 768 x 768 generation -
 * cost the overhang's MFMAs, nothing else); anything else returns DSC_ERR_UNSUPPORTED and the caller keeps the library convolution.
 */
#define DSC_CONV_UPSAMPLE2X 1
#define DSC_CONV_STRIDE2 2
#define DSC_CONV_STRIDE2_PAD_BR 3   /* stride 2 with zero padding on the bottom / right only (F.pad (0,1,0,1) + stride-2 conv of
                                     * the AutoencoderKL encoder's Downsample2D): the ODD pixels of the stride-1 / pad-1 taps */
int dsc_conv3x3_supported(int B, int H, int W, int Cin, int Cout);
/* diagnostics: 8 x int64 per workgroup (start / loop start / loop end / end in 100 MHz ticks, the three segment lengths in
 * shader clocks, XCC and HW ids) of every following dsc_conv3x3_nhwc_f16 call go to `device_buffer`; NULL switches it off */
void dsc_debug_set_conv_stamps(void* device_buffer);
/* diagnostics: 3 / 9 = force the weight-tile ring depth, 0 = by grid size and tuning profile; 200 + n = the split-count
 * model's per-step time for grids of <= 256 workgroups (n / 100 us); 300 / 301 = pixel tiles / channel blocks fastest
 * within an XCD (default: by shape); 400 / 401 / 402 = the nine-stage kernels without DMA-only loader waves / by rule / always */
void dsc_debug_set_conv_ring(int stages);
size_t dsc_conv3x3_workspace_bytes(int B, int H, int W, int Cin, int Cout, int splits);
int dsc_conv3x3_nhwc_f16(const void* x, const void* w, const void* bias, const void* residual, void* out,
                         int B, int H, int W, int Cin, int Cout, int64_t ldx, int64_t ldr, int64_t ldo,
                         int resample, int out_nchw, int splits, int dtype, void* workspace, size_t workspace_bytes,
                         void* stream);

/*
 * dsc_linear_f16 with a LayerNorm folded in on either side - removes the `nn.LayerNorm` launches of diffusers'
 * BasicTransformerBlock (norm1/norm2/norm3) between the token-major GEMMs of a block:
 *   ln_out != NULL  (producer): also writes, per output row and 64-column block, the (sum, sum of squares) of the fp16
 *                   output row segment to ln_out[M][N/64][2] (fp32) - the statistics of the residual stream s = out.
 *   ln_in  != NULL  (consumer): x is the UN-normalised s [M, K = C], w = W diag(gamma) (fp16), bias = beta.W^T + b,
 *                   ln_cvec[n] = sum_k w[n,k] (fp32); per row mu, rstd come from ln_in[M][ln_nb][2] (summed in block order) and
 *                   out[m,n] = rstd_m (acc[m,n] - mu_m cvec[n]) + bias[n]      ( = LayerNorm(s)[m,:] . W[n,:] + b[n] )
 * geglu may be combined with ln_in, not with ln_out.  Same shape limits as dsc_linear_f16.
 */
int dsc_linear_ln_f16(const void* x, const void* w, const void* bias, const void* residual, void* out,
                      int64_t M, int N, int K, int64_t ldx, int64_t ldr, int64_t ldo, int geglu,
                      const float* ln_in, int ln_nb, const float* ln_cvec, float ln_eps, float* ln_out,
                      int dtype, void* stream);
/*
 * The self-attention branch's fused q / k / v projection (reference attention_modify.py:458-474: `attn.to_q / to_k / to_v` on
 * the same hidden states, then `.view(batch, -1, heads, head_dim).transpose(1, 2)`) as ONE GEMM over w = [Wq; Wk; Wv]
 * ([3C, K]) whose epilogue already performs the head split for K and V: columns [0, C) -> q_out [M, C] (row stride ldq),
 * columns [C, 3C) -> kv_out [2][M / seq_len][heads][seq_len][C / heads] (fp16, 16-byte aligned) - each head's keys / values
 * contiguous, the layout dsc_self_attn_fwd's 64-key LDS-DMA tiles want (contiguous 1-KiB pieces instead of C/heads-element
 * row segments 3C elements apart).  Same LayerNorm-folding arguments as dsc_linear_ln_f16 (ln_in NULL = plain GEMM).
 * Needs C % 64 == 0, (C / heads) % 8 == 0, K % 64 == 0, M % seq_len == 0.
 */
/* diagnostics: launch variant of dsc_linear_f16 / _ln_f16 / _qkv_f16, one decimal field each (0 = the measured rule):
 *   v % 10            K-tile ring depth (2 or 3 stages)
 *   v / 10 % 1000     tile height (64 or 128 token rows)                          e.g. 643 = 64-row tiles, 3 stages
 *   v / 10000 % 10    DMA-only loader waves: 4 = for every tile, 9 = never
 *   v / 100000 % 10   128-column tiles: 1 = never, 2 = wherever N is a multiple of 128
 *   v / 1000000 % 10  workgroup order: 1 = plain blockIdx, 2 = XCD-aware for every shape
 *   v / 10000000 % 10 1 = non-temporal stores of the GEGLU output (measured: no effect)
 *   v < 0             -v = the grid (workgroups) a plain GEMM must keep to take 128-column tiles (default 512: the throughput
 *                     tier - 8 images per generation, coalesced requests; no batch-1 launch qualifies)
 * Every variant gives equal bytes (tests/test_unet_pipeline_gpu.py::test_linear_kernel_tilings_agree_bit_for_bit). */
void dsc_debug_set_gemm_stages(int stages);
int dsc_linear_qkv_f16(const void* x, const void* w, const void* bias, void* q_out, void* kv_out,
                       int64_t M, int C, int K, int64_t ldx, int64_t ldq, int heads, int seq_len,
                       const float* ln_in, int ln_nb, const float* ln_cvec, float ln_eps, int dtype, void* stream);

/*
 * 3x3 / pad 1 convolution with few input channels (<= 16) - the UNet's `conv_in` (4 -> 320; reference
 * u_net_condition_modify.py:352-356,1187): x [B,Cin,H,W] channel-major fp16 (the sampler's latent layout),
 * w_t [9*Cin, Cout] = weight.reshape(Cout, Cin*9).t() (k = (ci*3 + dy)*3 + dx), out [B,H,W,Cout] channels-last, bias fused.
 * Cout % 8 == 0, Cout <= 512, W % 8 == 0.
 */
int dsc_conv3x3_fewcin_f16(const void* x_nchw, const void* w_t, const void* bias, void* out_nhwc,
                           int B, int Cin, int H, int W, int Cout, int dtype, void* stream);

/*
 * out = x . w^T (+ bias) (+ residual) as ONE hipBLASLt launch (bias epilogue + beta = 1 with C = residual) - the plain
 * library GEMM for the token-major linears dsc_linear_f16 does not cover (few rows, long K); replaces `F.linear` + the
 * separate residual add of diffusers' FeedForward / Transformer2DModel.proj_out / ResnetBlock2D.conv_shortcut.
 * Same operand conventions as dsc_linear_f16; K, N and the strides multiples of 8.  The first call for a shape queries
 * the heuristic (and creates the handle / a 32 MiB workspace): run it once outside a graph capture.
 */
int dsc_linear_lt_f16(const void* x, const void* w, const void* bias, const void* residual, void* out,
                      int64_t M, int N, int K, int64_t ldx, int64_t ldr, int64_t ldo, int dtype, void* stream);
/* 1 when this build of the library contains the hipBLASLt path above (built with DSC_WITH_HIPBLASLT=1), 0 for the default
 * build, which neither contains nor links hipBLASLt: dsc_linear_lt_f16 then returns DSC_ERR_UNSUPPORTED for every shape and
 * the caller's own GEMM (dsc_linear_f16 / dsc_linear_splitk_f16) runs it - what the pipeline does by default since round 3. */
int dsc_has_library_gemm(void);
/* Generation slot of the calling host thread (0..3, default 0): which of the library-GEMM workspaces dsc_linear_lt_f16 hands
 * to hipBLASLt (its stream-K kernels keep partial tiles there).  A pipeline that keeps two generations in flight on two
 * streams drives each from its own thread / slot, so that neither the eager GEMMs nor the ones baked into the two captured
 * step graphs share a workspace.  Thread-local; returns DSC_ERR_BAD_ARG outside 0..3. */
int dsc_set_workspace_slot(int slot);
/*
 * Tuning profile of the launch rules, process-wide, read when a kernel is LAUNCHED (so a captured graph keeps the profile it
 * was captured under):
 *   DSC_TUNE_LATENCY     one generation at a time owns the chip (default): kernels may take a whole CU each - the 3x3
 *                        convolution's nine-stage weight ring for grids of <= 256 workgroups (all of a CU's LDS), the GEMM's
 *                        four extra DMA-only loader waves for its 64-row tiles.  +1.5 % images/s one at a time.
 *   DSC_TUNE_THROUGHPUT  several generations share the chip on their own streams: the three-stage ring / four-wave forms,
 *                        which leave room for the other stream's workgroups on the same CU.  +3 % images/s with two generations
 *                        in flight against the latency rules (7 interleaved runs each on one box: 10.81 vs 10.46).
 * This library's OWN kernels give equal bytes under the two (tests/...::test_conv3x3_and_gemm_profiles_give_equal_bytes); a GEMM
 * that is left to hipBLASLt (dsc_linear_lt_f16) may run another library algorithm under DSC_TUNE_THROUGHPUT, and two library
 * algorithms need not add in the same order: those results agree to rounding, not to the bit.
 * Returns DSC_ERR_BAD_ARG for any other value.
 */
#define DSC_TUNE_LATENCY 0
#define DSC_TUNE_THROUGHPUT 1
int dsc_set_tuning_profile(int profile);
int dsc_get_tuning_profile(void);
/* Diagnostic: out[0] = shapes planned so far, out[1] = library candidate algorithms the heuristic offered, out[2] = how
 * many of those need a workspace (stream-K / split-K kernels whose workgroups wait on each other's partial tiles) and were
 * therefore NOT eligible: dsc_linear_lt_f16 only ever runs workspace-free algorithms, which finish under any residency. */
void dsc_linear_lt_stats(long long out[3]);

/*
 * Few-row linear (M <= 8) for the time-embedding path: y[m,n] = act(sum_k x[m,k] w[n,k] + bias[n]) - `Timesteps` +
 * `TimestepEmbedding.linear_1/linear_2` (reference u_net_condition_modify.py:554-560, 1040-1060) and the per-ResNet
 * `time_emb_proj(silu(temb))` projections.  x [M,K] fp16 (row stride ldx) or, with DSC_ROWS_SINUSOID_IN, fp32 t[M] from which
 * the sinusoidal embedding [cos | sin] of width K is generated on the fly; DSC_ROWS_SILU_OUT applies SiLU to the result.
 */
#define DSC_ROWS_SINUSOID_IN 1
#define DSC_ROWS_SILU_OUT 2
int dsc_linear_rows_f16(const void* x, const void* w, const void* bias, void* out, int M, int N, int K,
                        int64_t ldx, int64_t ldo, int flags, int dtype, void* stream);

/*
 * (residual add +) LayerNorm over the last dimension - replaces the `x = attn(...) + x` elementwise add and the
 * `nn.LayerNorm` that follows it in diffusers' BasicTransformerBlock (norm1/norm2/norm3, eps 1e-5):
 *   s[r, :]  = x[r, :] + a[r, :]            (a == NULL: s = x)         -> written to `sum_out` when non-NULL (fp16)
 *   y[r, :]  = (s - mean_r) * rstd_r * gamma + beta                     statistics in fp32 on the fp16-rounded s
 * rows x C fp16, contiguous rows; C % 8 == 0, C <= 4096.  One wave per row, 16-byte accesses, one launch.
 */
int dsc_add_layernorm(const void* x, const void* a, const void* gamma, const void* beta, void* sum_out, void* y,
                      int64_t rows, int C, float eps, int dtype, void* stream);

/* GEGLU of the transformer feed-forward (diffusers GEGLU): y[r, j] = x[r, j] * gelu(x[r, n + j]), exact erf gelu.
 * x fp16 [rows, 2n] contiguous, y fp16 [rows, n]; n % 8 == 0. */
int dsc_geglu(const void* x, void* y, int64_t rows, int n, int dtype, void* stream);

/*
 * GroupNorm statistics from the PRODUCER's epilogue - the GroupNorms of the UNet's ResNet / transformer blocks
 * (modules/u_net_condition_modify.py:465-470,1304-1306 and the diffusers blocks it builds) at the 64x64 / 32x32 levels were a
 * statistics launch + an apply launch each, and the statistics launch re-read a tensor that the convolution / GEMM in front of it
 * had just written.  These entries let that producer emit the statistics:
 *   dsc_conv3x3_gn_nhwc_f16  = dsc_conv3x3_nhwc_f16 (no split, channels-last output) + `add`: an optional per-IMAGE bias row
 *                              add[b][c] (fp16, row stride add_ld: the ResNet block's time-embedding projection, which
 *                              diffusers adds between conv1 and norm2) + the partial sums of the fp16 tensor it stores;
 *   dsc_linear_gn_f16        = dsc_linear_f16 (bias / residual epilogue) whose rows are the pixels of images of
 *                              `rows_per_image` rows + the same partial sums (1x1 convolutions: proj_out, conv_shortcut);
 *   dsc_groupnorm_apply_nhwc = the whole GroupNorm (+ SiLU) in ONE launch given those partial sums.
 * gn_part is fp32 [B][rows][groups][2][2] with rows = dsc_*_gn_rows(...) pixel tiles per image (0 = shape not covered: 16-wide
 * convolution tiles without split-K, Cout % 64 == 0, groups of <= 64 channels, <= 128 tiles per image); slot [g][1] holds the
 * part of a group that lies beyond a 64-channel tile boundary.  Fixed summation orders: bit-reproducible.  No workgroup waits
 * for another - the partial sums cross the kernel boundary.
 */
int dsc_conv3x3_gn_rows(int B, int H, int W, int Cin, int Cout, int groups, int resample);
int dsc_conv3x3_gn_nhwc_f16(const void* x, const void* w, const void* bias, const void* add, int64_t add_ld,
                            const void* residual, void* out, int B, int H, int W, int Cin, int Cout, int64_t ldx,
                            int64_t ldr, int64_t ldo, int resample, float* gn_part, int groups, int dtype, void* stream);
int dsc_linear_gn_rows(int64_t M, int N, int K, int rows_per_image, int groups);
int dsc_linear_gn_f16(const void* x, const void* w, const void* bias, const void* residual, void* out,
                      int64_t M, int N, int K, int64_t ldx, int64_t ldr, int64_t ldo, int rows_per_image,
                      float* gn_part, int groups, int dtype, void* stream);
int dsc_groupnorm_apply_nhwc(const void* x, void* y, const void* gamma, const void* beta, const float* gn_part,
                             int part_rows, int B, int C, int hw, int groups, float eps, int apply_silu, int dtype,
                             void* stream);

/*
 * Split-K form of dsc_linear_f16 for few-row, long-K projections - out = x . w^T (+ bias) (+ residual) - the feed-forward
 * output projections (`ff.net.2`, K = 4C) and the 1x1 `conv_shortcut`s on concatenated skips (K = 1920 / 2560) of the 16x16 and
 * 8x8 UNet levels that modules/u_net_condition_modify.py builds from diffusers blocks (M = 128 / 512 token rows at batch 1).
 * Those GEMMs are bound by how many bytes of cold WEIGHTS are in flight, so the K tiles are dealt over `splits` workgroups per
 * output tile (splits <= 0: chosen from the shape), each writes its raw fp32 tile to `workspace` ([splits][M][N] fp32,
 * dsc_linear_splitk_workspace_bytes; 16-byte aligned) and a second launch adds the splits IN ORDER, then bias and residual, with
 * one fp16 rounding: bit-reproducible, no atomics, no workgroup waits on another.  Same operand constraints as dsc_linear_f16
 * (K % 64 == 0, N % 64 == 0, 16-byte aligned rows); splits == 1 after clamping runs dsc_linear_f16 itself.
 */
size_t dsc_linear_splitk_workspace_bytes(int64_t M, int N, int K, int splits);
int dsc_linear_splitk_f16(const void* x, const void* w, const void* bias, const void* residual, void* out,
                          int64_t M, int N, int K, int64_t ldx, int64_t ldr, int64_t ldo, int splits,
                          void* workspace, size_t workspace_bytes, int dtype, void* stream);

/*
 * Row softmax of materialised fp16 scores: probs[r, :] = softmax(scale * scores[r, :]), fp32 arithmetic, one fp16 rounding.
 * The middle step of the VAE decoder's / encoder's single 512-channel attention head (diffusers AutoencoderKL mid-block
 * `Attention`, reached from modules/model_k_diffusion.py:291-299 `decode_latents` and :600-606 `vae.encode`): too wide for
 * the flash kernel's registers, so it runs scores = q.k^T (dsc_linear_f16) -> this -> out = probs.v (dsc_linear_f16).
 * rows x n fp16, row strides ld_* in elements (% 8 == 0); n % 8 == 0, n <= 16384.  One workgroup per row, one launch.
 */
int dsc_softmax_rows_f16(const void* scores, void* probs, int64_t rows, int n, int64_t ld_scores, int64_t ld_probs,
                         float scale, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DSC_HIP_H */
