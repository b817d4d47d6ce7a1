/* mmgt_hip.h — C ABI of libmmgt_hip.so: the MI355X (gfx950) kernels behind MMGT's Stage-2 denoising path.
 *
 * This is the drop-in boundary.  The reference has no native layer at all: every call below replaces arithmetic that
 * the reference reaches through PyTorch / diffusers modules, cited per entry point (paths relative to the reference
 * checkout).  A maintainer binds these with ctypes; mmgt_amd/hip.py is exactly such a binding.
 *
 * Conventions
 *  - All pointers are DEVICE pointers unless stated; inputs are borrowed and never written; outputs are caller-owned.
 *  - `dtype` selects the storage type of activations and weights: MMGT_BF16 (product) or MMGT_F32 (the fp32-I/O parity
 *    mode of the same kernels).  Accumulation is always fp32.  Bias / norm affine / mask / scale vectors are fp32.
 *  - Activations are channels-last: an image tensor is (N, H, W, C) == a token matrix (N*H*W, C), row stride given.
 *  - `stream` is a hipStream_t (0 = default stream).  Calls only enqueue work; nothing synchronises, allocates or frees,
 *    so sequences of calls may be captured into a hipGraph.
 *  - Return 0 on success, non-zero on error with a message in mmgt_last_error() (thread-local).  Unsupported shapes are
 *    errors; there is no fallback path.
 *  - ONE DEVICE PER PROCESS, calls from one thread at a time: the library caches per-kernel launch state (the > 64 KB
 *    dynamic-LDS opt-in and the resident-workgroup count of each GEMM tile) for the device current at the first call.
 *    This is the deployment model of the path (one process per GPU); a process that switches devices must
 *    not reuse the library.
 */
#ifndef MMGT_HIP_H
#define MMGT_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define MMGT_F32 0
#define MMGT_BF16 1

#define MMGT_ACT_NONE 0
#define MMGT_ACT_GEGLU 1 /* weights packed per 64 rows as [32 h | 32 gate]; output has N/2 columns */
#define MMGT_ACT_SILU 2
#define MMGT_ACT_RELU 3
#define MMGT_ACT_QUICK_GELU 4 /* x * sigmoid(1.702 x): the CLIP vision tower's MLP (transformers activations.py) */
#define MMGT_ACT_GELU 5       /* exact-erf GELU: SMGA's feed-forward blocks (src/audio2pose_model/SMGA.py:92) */
#define MMGT_ACT_MISH 6       /* x * tanh(softplus(x)): SMGA's time MLP (src/audio2pose_model/model.py:370-374) */

int mmgt_abi_version(void);
const char* mmgt_last_error(void);
/* Benchmark-only knobs (A/B measurements and tests; the defaults are the product):
 *   "gemm_cfg" = 0 (heuristic tile choice) or 1, 3, 6, 9, 12, 16, 17, 19, 20 (force a GEMM / conv tile configuration; 19 = gemm16's 192 x 320 tile,
 *                20 = its 256 x 128 tile: measured slower than the 128 x 128 tile everywhere, kept for the record);
 *   "bm192"    = 1 (default) / 0: the 192-row gemm16 tile for shapes whose 256-row tile count leaves the last round of the grid half empty;
 *   "attn64"   = 1 (default) / 0: the 64-queries-per-wave spatial attention kernel at head_dim 40;
 *   "attn_nomax" = 1 (default) / 0: that kernel without the running maximum (the softmax reference stays what a row's first 32 keys set it to; a
 *                workgroup in which a denominator ends beyond 2^100 runs its tile loop again with the running maximum: csrc/attn64.hip);
 *   "attn80"   = 1 (default) / 0: head_dim 80 on the LDS-DMA-staged kernel of csrc/attn80.hip (0: attention.hip's register-staged kernel);
 *   "gn_slab"  = 1 (default) / 0: GroupNorm below the 64 x 64 level with the image x group slab in registers, one read of the tensor
 *                (0: the multi-pass kernels);
 *   "g16_stagger" = -1 (default: by shape) / 0 / n: every second CU's gemm16 workgroup starts n x 512 clocks late (short reductions with a
 *                residual epilogue: the chip is then not in the tile-end phase all at once); timing only, results are bitwise the same;
 *   "gn_rows"  = 0 (default: measured choice) or the GroupNorm rows per workgroup;
 *   "splitk"   = 1 (default) / 0: split-K of long reductions on grids of at most half a tile per CU (the 8x8-level convs);
 *   "ffn_dbg", "tleg_abl", "gnconv_abl", "rowgemm_dbg" 1 .. 4: timing ablations whose RESULTS ARE GARBAGE.  The product library does not
 *   contain them and refuses the keys with an error; they exist in libmmgt_hip_abl.so only (`make -C mmgt_amd/csrc abl`, -DMMGT_ABLATE),
 *   which the instruments under tools/ build and load explicitly (tools/abl_lib.py).
 *   "tailsplit" = 1 (default) / 0: convs whose tile count leaves the last round of the persistent grid half empty run that round's rows
 *               as a second launch with the reduction split in two (A/B switch).
 * One kernel per family ships: the measured-slower variants of earlier rounds (gemm16s / gemm16v, the phased and register-staged
 * head_dim-40 attention kernels, the producer / consumer FeedForward) are records under tools/micro/. */
int mmgt_tune(const char* key, int value);
/* Host-side switches kept in the same table (what mmgt_amd/unet3d.py, pipeline.py and smga.py consult; all default 1, 0 = the
 * unfused / stateless form, for same-box A/Bs): "fused_ff", "twin_attention", "shared_rows", "oz3", "rowgemm", "tleg", "up2" (the up-sampling convs in the four-phase form), "conv_out_taps" (conv_norm_out + SiLU + conv_out as a 36-column GEMM + a gather), "sc_cat" (conv_shortcut over [x | skip] as one two-source launch), "ffpo_cat" (ff2 + proj_out of a block as one two-source GEMM), "rconv" (a mask: 1 the 320-wide resnets, 2 the 640-wide, 4 the 1280-wide), "gnconv" (mmgt_amd/vae.py),
 * "zero_audio_skip", "window_state", "smga_graph".  mmgt_tune sets them, mmgt_tune_get reads them: one state describes a run. */
int mmgt_tune_get(const char* key, int* value);
/* Box calibration for bench.py's `box_calib` (csrc/calib.hip): a bare v_mfma_f32_16x16x32_bf16 loop on random operands, one wave per SIMD,
 * launched back to back for `warm_seconds` (0 .. 10); *mfma_mhz = the in-kernel clock of the last launch (delta s_memtime / delta
 * s_memrealtime, median over the workgroups), *mfma_tflops = the rate of the last batch of launches.  Measures the BOX (boxes differ by
 * several per cent in the clock they hold under load), not the product.  Host pointers; synchronises the stream. */
int mmgt_box_calib(float warm_seconds, float* mfma_mhz, float* mfma_tflops, void* stream);
/* Debug only (tools/trace_gemm16.py): p = device buffer of u64 [grid][32 tiles][2 wave groups][4] that gemm16's workgroups fill
 * with 100-MHz stamps at their tile phases; NULL (the default) switches the stamps off. */
void mmgt_gemm16_set_trace(void* p);
/* Debug only (tools/trace_ffn.py): u64 [workgroups][64] shader-clock stamps of mmgt_ff_fused's phases (wave 0); NULL = off. */
void mmgt_ffn_set_trace(void* p);

/* out[M,N] = epi(A[M,K] . W[N,K]^T):  v = acc + bias[n] + bias2[m / bias2_rows][n]; v = act(v);
 * v *= row_scale[m] * alpha; v += residual[m][n].   Optional batch (grid.z) with element strides bs*.
 * Replaces: nn.Linear / 1x1 nn.Conv2d / diffusers Attention.to_q,to_k,to_v,to_out / FeedForward(GEGLU) call sites
 * src/models/transformer_3d.py:176,253; src/models/attention.py:323-349,361,465,568-626,730-769;
 * src/models/motion_module.py:161,172,233,256; src/models/resnet.py:211-215,226,243; src/models/unet_3d.py:502. */
int mmgt_gemm(const void* A, long lda, const void* W, const float* bias, const float* bias2, int bias2_rows,
              const float* row_scale, float alpha, const void* residual, long ldr, void* out, long ldo, int M, int N,
              int K, int act, int batch, long bsA, long bsW, long bsR, long bsO, int dtype, void* stream);

/* The same with a second bias added AFTER the row scale:  v = (acc + bias[n]) * row_scale[m] * alpha + bias_post[n] +
 * residual[m][n].  Lets MM-HAA's  zero_conv_i(mask_i * to_out_i(a_i))  (attention.py:730-760: a 320x320 Linear, a per-token
 * mask multiply and a 1x1 conv per branch) run as ONE GEMM per branch on the host-merged weight W_z W_o: bias = W_z b_o,
 * row_scale = mask, alpha = motion_scale, bias_post = motion_scale * b_z. */
int mmgt_gemm_post(const void* A, long lda, const void* W, const float* bias, const float* row_scale, float alpha,
                   const float* bias_post, const void* residual, long ldr, void* out, long ldo, int M, int N, int K,
                   int dtype, void* stream);

/* 3x3 / pad 1 convolution on channels-last input as implicit GEMM.  x0 (NB,IH,IW,C0) and optional x1 (NB,IH,IW,C1) are
 * read as one (C0+C1)-channel tensor (the UNet skip concat, unet_3d_blocks.py:894,1057); `upsample` = the conv sees the
 * nearest-2x upsampled input (Upsample3D, resnet.py:70-88); stride 2 = Downsample3D (resnet.py:112-120).
 * stride -2 = stride 2 with the padding on the high side only (diffusers Downsample2D(padding=0): F.pad(x, (0,1,0,1)),
 * the AutoencoderKL encoder's downsampler).
 * upsample = 2: the same function as upsample = 1 from the FOUR-PHASE weight image [4][Cout][2][2][C0] (mmgt_amd/packing.py pack_conv3x3_up2): output
 * pixel (2 y + a, 2 x + b) reads the stored pixels (y + a - 1 .. y + a, x + b - 1 .. x + b) through the 3 x 3 taps summed per stored pixel -- four
 * 2 x 2 convs on the stored image (one per phase, grid.z), 16 instead of 36 multiply-adds per stored pixel and channel pair; bf16, one source,
 * bias only, Cout a multiple of 256 or 320 (the up-sampling convs of the UNet, the ReferenceNet and the VAE decoder).
 * Wp: [Cout][3][3][C0+C1].  out (NB,OH,OW,Cout) = act(conv + bias + bias2[pixel / bias2_rows]) + residual.
 * Replaces: InflatedConv3d (src/models/resnet.py:9-17) in ResnetBlock3D.conv1/conv2 (resnet.py:223,240), conv_in /
 * conv_out (unet_3d.py:517,620), PoseGuider convs (pose_guider.py:47-57), AutoencoderKL decoder convs (diffusers). */
int mmgt_conv3x3_nhwc(const void* x0, int C0, const void* x1, int C1, int NB, int IH, int IW, int stride, int upsample,
                      const void* Wp, const float* bias, const float* bias2, int bias2_rows, const void* residual,
                      void* out, int Cout, int act, int dtype, void* stream);

/* 1 x 1 conv over the channel concatenation of two channels-last tensors, out[p][o] = bias[o] + Wp[o] . [x0[p] | x1[p]] (+ residual[p][o]): the
 * resnets' conv_shortcut over [hidden | skip] (resnet.py:243-245; unet_3d_blocks.py:941-969 concatenates first) as one launch -- the conv gather's
 * two-source reduction with a single tap -- instead of two dense GEMMs chained through a residual.  bf16; `rows` pixels; C0, C1 multiples of 64, Cout a
 * multiple of 256 or 320; Wp [Cout][C0 + C1]; residual / out (rows, Cout). */
int mmgt_conv1x1_cat_nhwc(const void* x0, int C0, const void* x1, int C1, long rows, const void* Wp, const float* bias, const void* residual, void* out,
                          int Cout, int dtype, void* stream);

/* A 3 x 3 / pad 1 conv with FOUR output channels (conv_out, unet_3d.py:620) as a GEMM + this gather: Y (NB, H, W, ldY) bf16 holds, per pixel, the 36
 * products W[o][tap] . x[pixel] in column 4 tap + o (one GEMM over the pixel's channels for all nine taps: mmgt_rowgemm320 with norm = 3, the GroupNorm
 * tables + SiLU of conv_norm_out applied while the rows are loaded); out (NB, H, W, 8) bf16 = bias[o] + the sum over the taps of the NEIGHBOUR pixel's
 * product (out-of-image neighbours contribute nothing: the zero padding), channels 4 .. 7 zero. */
int mmgt_conv_taps_gather(const void* Y, int ldY, const float* bias, void* out, int NB, int H, int W, int dtype, void* stream);

/* GroupNorm over channels-last images, one statistic per (image, group), optional fused SiLU.
 * x,out (NB, HW, C) where C = C0 + C1 may be split over two sources (skip concat); workspace: NB*chunks*G*2 floats
 * with chunks = mmgt_groupnorm_chunks(HW).
 * Replaces: InflatedGroupNorm / nn.GroupNorm (+F.silu) at resnet.py:220-221,231,237; transformer_3d.py:174;
 * motion_module.py:156; unet_3d.py:618-619. */
int mmgt_groupnorm_chunks(int HW);
/* GroupNorm statistics only, as per-(image, channel) tables scale = rstd * gamma, shift = beta - mean * scale ([NB][C] fp32 each;
 * HW > 256): mmgt_rowgemm320(norm = 2) applies them while it loads x -- transformer_3d.py:174-188 (norm -> proj_in), motion_module.py:156-170. */
int mmgt_groupnorm_affine(const void* x, int C, const float* gamma, const float* beta, float* workspace, float* scale, float* shift,
                          int NB, int HW, int G, float eps, int dtype, void* stream);
/* ... of the channel concatenation x0 | x1 (a resnet's norm1 behind a skip connection: unet_3d_blocks.py:941-969; groups may straddle the seam);
 * x1 NULL / C1 0: one tensor. */
int mmgt_groupnorm_affine2(const void* x0, int C0, const void* x1, int C1, const float* gamma, const float* beta, float* workspace, float* scale,
                           float* shift, int NB, int HW, int G, float eps, int dtype, void* stream);
int mmgt_groupnorm_nhwc(const void* x0, int C0, const void* x1, int C1, const float* gamma, const float* beta, void* out,
                        float* workspace, int NB, int HW, int G, float eps, int silu, int dtype, void* stream);

/* LayerNorm over the last dimension of a (rows, C) token matrix, optional additive table pe[(row / pe_div) % pe_mod][C]
 * applied AFTER the affine transform (the temporal positional encoding enters q, k and v: motion_module.py:359-366).
 * With pe == NULL and pe_mod > 1, `beta` is itself a [pe_mod][C] table indexed the same way (the host folds
 * beta + pe once per model: one vector load per row less, 2x on the L0 temporal norms).
 * Replaces: nn.LayerNorm at attention.py:392-396,450-454,465,677-681,712-716,769; motion_module.py:244,256;
 * mutual_self_attention.py:122,195-199,210; audio_proj.py:119. */
int mmgt_layernorm(const void* x, long ldx, const float* gamma, const float* beta, float eps, const float* pe, int pe_div,
                   int pe_mod, void* out, long ldo, int rows, int C, int dtype, void* stream);

/* softmax(Q K^T * scale) V for `batch` x `heads` independent problems, flash style (scores never leave the chip).
 *   Q element (b, i, h, d) at q[(b / q_bdiv) * q_bs0 + (b % q_bdiv) * q_bs1 + i * q_ts + h * hd + d]; K, V, O likewise.
 *   Keys/values come from up to two segments: [K, V] with nk keys, then [K2, V2] with nk2 keys for batches
 *   b >= seg2_first_batch (the ReferenceNet feature bank that only the conditional CFG half attends to); segment 2
 *   uses batch index b / k2_bdiv2 with stride k2_bs.
 *   v_transposed = 1: V (and V2) are stored [b][h*hd + d][key] (key contiguous, row stride v_ts).
 * Replaces: diffusers Attention + AttnProcessor2_0 (F.scaled_dot_product_attention) at attention.py:323-349,568-626;
 * motion_module.py:377-383; and the bank concat + uncond recompute of mutual_self_attention.py:149-188. */
int mmgt_attention(const void* q, long q_bs0, long q_bs1, long q_ts, const void* k, long k_bs0, long k_bs1, long k_ts,
                   const void* v, long v_bs0, long v_bs1, long v_ts, void* o, long o_bs0, long o_bs1, long o_ts,
                   int bdiv, const void* k2, const void* v2, long k2_bs, long k2_ts, long v2_bs, long v2_ts, int k2_bdiv,
                   int nk2, int seg2_first_batch, int batch, int heads, int hd, int nq, int nk, float scale,
                   int v_transposed, int dtype, void* stream);

/* mmgt_attention with a per-row output multiplier per group of `os_heads` heads, applied in fp32 with the softmax normalisation:
 *   o[b][q][head] *= out_scale[(head / os_heads) * os_group_stride + b * nq + q]
 * Single key segment, row-major V, bdiv == 1.  MM-HAA (attention.py:730-760): the three masked audio cross-attentions write
 * mask_i * attn2_i(x, audio_i), so that  sum_i zero_conv_i(mask_i * to_out_i(.))  is ONE GEMM over the concatenated reduction. */
int mmgt_attention_scaled(const void* q, long q_bs0, long q_bs1, long q_ts, const void* k, long k_bs0, long k_bs1, long k_ts,
                          const void* v, long v_bs0, long v_bs1, long v_ts, void* o, long o_bs0, long o_bs1, long o_ts,
                          const float* out_scale, long os_group_stride, int os_heads, int batch, int heads, int hd, int nq, int nk,
                          float scale, int dtype, void* stream);

/* mmgt_attention whose every batch entry reads the second key segment, with a TWIN output: o_twin (o's strides) receives the attention over
 * the FIRST segment alone -- the state of the online softmax after its last tile --, o the attention over both.  The CFG pair of the first
 * reference-attention reader (mutual_self_attention.py:160-230): both rows enter with the same hidden states, the conditional row attends
 * [x | bank], the unconditional row [x]; one pass over x serves both.  bf16, head_dim 40, V transposed, nq % 256 == 0, nk % 64 == 0,
 * nk2 % 64 == 0 only (csrc/attn64.hip); anything else is an error. */
int mmgt_attention_twin(const void* q, long q_bs0, long q_bs1, long q_ts, const void* k, long k_bs0, long k_bs1, long k_ts,
                        const void* v, long v_bs0, long v_bs1, long v_ts, void* o, void* o_twin, long o_bs0, long o_bs1, long o_ts,
                        int bdiv, const void* k2, const void* v2, long k2_bs, long k2_ts, long v2_bs, long v2_ts, int k2_bdiv,
                        int nk2, int batch, int heads, int hd, int nq, int nk, float scale, int dtype, void* stream);

/* Row softmax of a (rows, cols) matrix scaled by `scale` (materialised-score attention of the VAE mid block). */
int mmgt_softmax_rows(const void* x, long ldx, void* out, long ldo, int rows, int cols, float scale, int dtype,
                      void* stream);

/* The bf16 path of the VAE mid-block attention (one 512-wide head over 4096 tokens per frame; replaces diffusers 0.24.0
 * `Attention` as called by `AutoencoderKL.decode` at pipeline_pose2vid_long.py:112-125).  Its logits reach hundreds with
 * sd-vae-ft-mse weights, so q . k keeps ~17 bits through hi / lo bf16 operand pieces on the bf16 MFMA path instead of an fp32 GEMM:
 *   mmgt_gemm_bf16_f32          out fp32 [M][ldo] = A bf16 [M][lda] . W bf16 [N][ldw]^T, raw accumulators, no epilogue
 *                               (K % 64 == 0, N % 8 == 0);
 *   mmgt_qk_split3              qk fp32 [rows][4C] = [t Wq_hi^T | t Wq_lo^T | t Wk_hi^T | t Wk_lo^T]  ->  q = qk0 + qk1 + bias_q,
 *                               k = qk2 + qk3 + bias_k,  Qp bf16 [rows][3C] = [q_hi | q_hi | q_lo],  Kp = [k_hi | k_lo | k_hi];
 *   mmgt_softmax_rows_f32_bf16  fp32 logits -> bf16 probabilities in one pass (cols a multiple of 256, at most 8192). */
int mmgt_gemm_bf16_f32(const void* A, long lda, const void* W, long ldw, float* out, long ldo, int M, int N, int K, void* stream);
int mmgt_qk_split3(const float* qk, const float* bias_q, const float* bias_k, void* Qp, void* Kp, long rows, int C, void* stream);
int mmgt_softmax_rows_f32_bf16(const float* x, long ldx, void* out, long ldo, int rows, int cols, float scale, void* stream);

/* Layout / dtype plumbing between the reference's (b, c, f, h, w) fp32 tensors and channels-last T:
 * out[(b*F + f), y, x, c] (c padded with zeros up to Cpad) <- in[b, c, f, y, x]; and the inverse (first C channels).
 * Replaces: the einops rearranges at resnet.py:13-15; transformer_3d.py:158,178-180,248-252,264. */
int mmgt_ncfhw_to_nhwc(const float* in, void* out, int B, int C, int F, int H, int W, int Cpad, float scale, int dtype,
                       void* stream);                       /* out = in * scale   (z / 0.18215 before the VAE, :114) */
int mmgt_nhwc_to_ncfhw(const void* in, float* out, int B, int C, int F, int H, int W, int Cpad, float scale, float shift,
                       int clamp01, int dtype, void* stream); /* out = clamp(in * scale + shift): (x/2+0.5).clamp(0,1), :122 */

/* Sinusoidal timestep features: out[b][0:half] = cos(t_b * f_i), out[b][half:] = sin(t_b * f_i), f_i = 10000^(-i/half).
 * Replaces: diffusers Timesteps(flip_sin_to_cos=True, shift 0) at unet_3d.py:496.  timesteps: fp32 [B] device. */
int mmgt_timestep_features(const float* timesteps, void* out, int B, int dim, int dtype, void* stream);

/* LayerNorm -> FeedForward(GEGLU) -> + residual of a transformer block as ONE launch, 320 channels, bf16 (csrc/ffn.hip):
 *   out[m] = residual[m] + bias2 + W2 . (h * gelu(g)),  [h | g] = W1 . LN(x[m]) + b1     (ln_gamma == NULL: no LayerNorm)
 * wimg: the weight image built by mmgt_amd/packing.py: pack_ff_fused (mmgt_ff_fused_image_bytes(C, inner) bytes; -1 if the shape
 * is not supported).  The hidden activations never reach memory: x is read once, out written once.
 * Replaces: the norm3 / ff_norm + `self.ff(...) + hidden_states` lines of src/models/attention.py:361,465,642,769 and
 * src/models/motion_module.py:253-254 (diffusers FeedForward) at the 64x64 level. */
int mmgt_ff_fused_image_bytes(int C, int inner);
int mmgt_ff_fused(const void* x, long ldx, const float* ln_gamma, const float* ln_beta, float eps, const void* wimg,
                  const float* bias2, const void* residual, long ldr, void* out, long ldo, int M, int C, int inner, int dtype,
                  void* stream);
/* ... and the block's proj_out on the end of the same launch:
 *   hidden = bf16(residual + bias2 + FeedForward(LN(x)))  (never stored),   out[m] = residual2[m] + bias_po + Wpo . hidden[m]
 * wpo: mmgt_amd/packing.py: pack_ff_proj_out(proj_out.weight (320, 320)), 204 800 bytes.
 * Replaces additionally: `hidden_states = self.proj_out(hidden_states)` + `output = hidden_states + residual` of
 * src/models/transformer_3d.py:262-268 and src/models/motion_module.py:178-182. */
int mmgt_ff_fused_po(const void* x, long ldx, const float* ln_gamma, const float* ln_beta, float eps, const void* wimg,
                     const float* bias2, const void* residual, long ldr, const void* wpo, const float* bias_po, const void* residual2,
                     long ldr2, void* out, long ldo, int M, int C, int inner, int dtype, void* stream);

/* [LayerNorm | GroupNorm ->] Linear(s) of the 320-channel level with the rows stationary in registers (csrc/rowgemm.hip), bf16:
 *   y[m, :] = [norm](x[m, :]) . W^T + bias [+ bias2[m / bias2_rows]] [+ residual[m, :]],   K = 320, N % 32 == 0, N <= 1920
 * norm = 0: none; 1: LayerNorm over the row (gamma [320], beta [pe_mod][320]); 2: x * gamma[g] + beta[g] with both tables
 * [pe_mod][320] and g = (m / pe_div) % pe_mod -- the second pass of a GroupNorm (tables from mmgt_groupnorm_affine, pe_div = HW,
 * pe_mod = NB) applied while the rows are loaded: the normalised tensor is never written.
 * Columns [0, n1) are written row-major to out[m * ldo + c]; columns [n1, N) TRANSPOSED per batch of n_tok rows to
 * out_t[(m / n_tok) * (N - n1) * npad + (c - n1) * npad + m % n_tok] (the V^T operand of mmgt_attention with v_transposed = 1), so the
 * q | k GEMM and the W . X^T GEMM of a self-attention and the LayerNorm in front of both are one launch that reads x once.
 * ln_gamma == NULL: no LayerNorm; ln_beta: [pe_mod][320], row (m / pe_div) % pe_mod (pe_mod <= 1: one row) -- the motion module's
 * positional encoding folded into beta as in mmgt_layernorm.  A residual needs n1 == N.  wimg: mmgt_amd/packing.py: pack_rowgemm
 * (mmgt_rowgemm320_image_bytes(N) bytes; -1 if N is not supported).
 * Replaces: norm1 / norm2 + to_q / to_k / to_v, and to_out[0] + residual, of src/models/attention.py:346-360,440-462,700-760
 * (diffusers Attention) and of src/models/motion_module.py:294-330 at the 64x64 level. */
long mmgt_rowgemm320_image_bytes(int N);
void mmgt_rowgemm_set_trace(void* stamps);   /* debug: u64 [workgroups][32] shader-clock stamps (mmgt_tune("rowgemm_dbg", 5)); NULL = off */
int mmgt_rowgemm320(const void* x, long ldx, int norm, const float* ln_gamma, const float* ln_beta, int pe_div, int pe_mod, float eps,
                    const void* wimg, const float* bias, const float* bias2, int bias2_rows, const void* residual, long ldr,
                    void* out, long ldo, int n1, void* out_t, int n_tok, int npad, int M, int N, int dtype, void* stream);

/* wav2vec2 feature extractor, element-wise pieces (csrc/wav2vec.hip; the convs / Linears / attention run on mmgt_gemm, mmgt_layernorm,
 * mmgt_attention).  x, out: (rows, C) channels-last in `dtype`.
 *   mmgt_channel_norm_gelu: out = gelu((x - mean_c) * rstd_c * gamma + beta), statistics per CHANNEL over the rows -- GroupNorm(C, C)
 *     over time + GELU of the first conv layer (transformers Wav2Vec2GroupNormConvLayer; src/models/wav2vec.py:73).
 *   mmgt_lerp_rows: rows_in -> rows_out frames, F.interpolate(mode="linear", align_corners=True) (src/models/wav2vec.py:196-209). */
int mmgt_channel_norm_gelu(const void* x, const float* gamma, const float* beta, void* out, int rows, int C, float eps, int dtype,
                           void* stream);
int mmgt_lerp_rows(const void* x, void* out, int rows_in, int rows_out, int C, int dtype, void* stream);

/* Elementwise x -> silu(x) (time embedding activation, resnet.py:226) over n elements. */
int mmgt_silu(const void* x, void* out, long n, int dtype, void* stream);

/* One CFG + DDIM update on fp32 latents (all (1,4,F,H,W) = n elements; pred_sum holds [uncond | cond] = 2n):
 *   eps = pred_sum / counter (per frame);  v = eps_u + s (eps_c - eps_u);
 *   x0 = sa_t x - sb_t v;  e = sa_t v + sb_t x;  x_prev = sa_p x0 + sb_p e          (v-prediction, eta = 0)
 * counter: fp32 [F]; frame index of element i = (i / hw) % F.
 * Replaces: src/pipelines/pipeline_pose2vid_long.py:622-635 + diffusers DDIMScheduler.step. */
int mmgt_cfg_ddim_step(const float* pred_sum, const float* counter, const float* latents, float* latents_out, long n,
                       int F, int hw, float guidance, float sa_t, float sb_t, float sa_p, float sb_p, void* stream);

/* pred_sum[:, :, idx[j]] += pred[:, :, j]; counter[idx[j]] += 1 for a window of Fw frames (closed-loop indices).
 * pred is channels-last T ((2*Fw), hw, Cpad) as produced by the UNet; pred_sum fp32 (2, C, F, hw).
 * Replaces: src/pipelines/pipeline_pose2vid_long.py:622-624. */
int mmgt_accumulate_window(const void* pred, float* pred_sum, float* counter, const int* idx, int Fw, int F, int C,
                           int Cpad, int hw, int dtype, void* stream);

/* The same for `rows` (1 or 2) CFG rows of pred_sum starting at row0: pred is ((rows*Fw), hw, Cpad).  The window-parallel
 * sampler (one long video over several GPUs) exchanges single CFG rows sliced to Cpad == C and accumulates them one by
 * one; bump_counter = 0 for the second row of a window so that counter counts windows, as :624 does. */
int mmgt_accumulate_window_rows(const void* pred, float* pred_sum, float* counter, const int* idx, int Fw, int F, int C,
                                int Cpad, int hw, int rows, int row0, int bump_counter, int dtype, void* stream);

/* ---- Stage-1 SMGA audio -> pose sampler: the element-wise glue between its Linear / attention / LayerNorm calls.
 * out = x rotated pairwise by the angle table cos_sin[(row % seq)][dim / 2][2] (cos, sin): RotaryEmbedding.rotate_queries_or_keys,
 * src/audio2pose_model/rotary_embedding_torch.py:38-61,106-113 (call sites model.py:121,267,298-299). */
int mmgt_rotary(const void* x, const float* cos_sin, void* out, long rows, int dim, int seq, int dtype, void* stream);
/* out = res (+ res2) + (scale[b] + 1) * x + shift[b], b = row / rows_per_batch, scale_shift fp32 rows of [scale(dim) | shift(dim)]
 * with row stride ld_ss: featurewise_affine of a DenseFiLM output plus the residual, model.py:44-64,231,246-259. */
int mmgt_film_residual(const void* x, const float* scale_shift, long ld_ss, const void* res, const void* res2, void* out, long rows,
                       int dim, int rows_per_batch, int dtype, void* stream);
/* out[b][c] (fp32) = mean over tokens of x[b][t][c]: model.py:460. */
int mmgt_mean_tokens(const void* x, float* out, int batch, int tokens, int dim, int dtype, void* stream);
/* out = act(x) element-wise for act in {MMGT_ACT_SILU, MMGT_ACT_GELU, MMGT_ACT_MISH}: DenseFiLM's Mish in front of its Linear
 * (model.py:50-52) where no GEMM epilogue can carry it. */
int mmgt_activation(const void* x, void* out, long n, int act, int dtype, void* stream);
/* x0 = clamp(u + (c - u) g, -1, 1); eps = (x sqrt(1/a) - x0) / sqrt(1/a - 1); out = last ? x0 : x0 sqrt(a') + c eps + sigma noise:
 * guided_forward + model_predictions + the DDIM update of src/audio2pose_model/diffusion.py:149-156,257-273 (fp32 sampler state). */
int mmgt_smga_ddim_step(const void* pred_uncond, const void* pred_cond, const float* x, const float* noise, float* out, long n,
                        float guidance, float sqrt_recip_acp, float sqrt_recipm1_acp, float sqrt_acp_next, float c, float sigma,
                        int last, int dtype, void* stream);

/* One launch per temporal-attention leg of a level-0 motion module (csrc/tleg.hip; src/models/motion_module.py:236-259,351-388):
 *   out = x + to_out(softmax_f(q k^T * scale) v),  q | k | v = (LayerNorm(x; ln_gamma, eps) + beta_pe[f]) . W^T   per pixel over its `frames` rows.
 * x / out (batch * frames * n_pix, 320) bf16, rows ordered (batch, frame, pixel), contiguous; out may alias x.  beta_pe (pe_rows >= frames, 320)
 * fp32 = LayerNorm bias + positional encoding of frame f.  wimg = packing.pack_tleg(Wq, Wk, Wv, Wo) (mmgt_temporal_leg320_image_bytes() bytes),
 * bias_o (320) fp32.  frames = 24 or 12 (48 rows per wave), n_pix a multiple of 4 * 48 / frames.  bf16 only: the fp32-I/O mode and every
 * other shape run rowgemm / LayerNorm + GEMM -> mmgt_attention -> GEMM + residual. */
long mmgt_temporal_leg320_image_bytes(void);
int mmgt_temporal_leg320(const void* x, void* out, const float* ln_gamma, const float* beta_pe, int pe_rows, const void* wimg, const float* bias_o,
                         int batch, int frames, int n_pix, float scale, float eps, int dtype, void* stream);

/* GroupNorm + SiLU + conv3x3 (stride 1, padding 1) in one launch for the VAE's 128- and 256-channel levels (csrc/gnconv.hip; diffusers
 * `ResnetBlock2D.forward` norm -> nonlinearity -> conv as `AutoencoderKL.decode` runs it for src/pipelines/pipeline_pose2vid_long.py:112-125):
 *   out[..., 0 .. Cout) = bias + conv3x3( silu( x * scale[n, c] + shift[n, c] ) ) (+ residual[..., 0 .. Cout))
 * x (nb, H, W, Cin) bf16 channels-last, H and W multiples of 16, smaller than 2 GiB; scale / shift (nb, Cin) fp32 = the tables of
 * mmgt_groupnorm_affine in ONE allocation (shift = scale + nb * Cin); wimg = packing.pack_gnconv(W) (mmgt_gn_silu_conv3x3_image_bytes(Cin, Cout) bytes); bias (Cout) fp32 or null;
 * residual / out: bf16 tensors of ldo >= Cout channels per pixel (a 256-wide output = two launches on its 128-channel halves), residual or null.
 * bf16; (Cin, Cout) = (128, 128), (256, 128), (128, 64: no residual).  Everything else runs mmgt_groupnorm -> mmgt_conv3x3. */
long mmgt_gn_silu_conv3x3_image_bytes(int cin, int cout);
int mmgt_gn_silu_conv3x3(const void* x, const float* scale, const float* shift, const void* wimg, const float* bias, const void* residual, void* out,
                         float* stats, int nb, int H, int W, int cin, int cout, int ldo, int dtype, void* stream);
/* `stats` (or null; Cout = 128 launches): the launch also writes, per 16 x 16 tile and 4-channel quad of its output, the pair (sum, sum of
 * squares) of the bf16 values it stores -- [nb * (H / 16) (W / 16)][ldo / 4][2] floats, a launch on the second half of a 256-wide output is handed
 * stats + 64.  mmgt_gn_stats_finalize folds them (fixed order) into the (scale, shift) tables of the GroupNorm that reads that output, so the
 * statistics pass of mmgt_groupnorm_affine over the tensor is not needed (resnet.py:20-28 `InflatedGroupNorm` / diffusers GroupNorm semantics:
 * biased variance over (H W C / G) values, eps inside the root).  scale | shift: one allocation, shift = scale + nb * C. */
int mmgt_gn_stats_finalize(const float* stats, const float* gamma, const float* beta, float* scale, float* shift, int nb, int tiles, int C, int G,
                           float eps, void* stream);

/* GroupNorm-apply + SiLU + conv3x3 (stride 1, padding 1) of the UNet's resnets in one launch (csrc/rconv.hip; ResnetBlock3D.forward,
 * src/models/resnet.py:217-247: norm1 -> nonlinearity -> conv1 (+ temb) and norm2 -> nonlinearity -> conv2 (+ shortcut), per frame as
 * InflatedGroupNorm / InflatedConv3d of :20-28, :156-196 run them):
 *   out = bias + bias2[n / b2_imgs] + conv3x3( silu( (x0 | x1) * scale[n, c] + shift[n, c] ) ) (+ residual)
 * x0 (nb, H, W, C0) and optionally x1 (nb, H, W, C1): the two halves of the channel concatenation in front of an up-block resnet; H, W multiples
 * of 16; C0, C1 multiples of 64; Cout a multiple of 320; scale_shift (2, nb, C0 + C1) fp32 = the tables of mmgt_groupnorm_affine2 in one
 * allocation; wimg = packing.pack_rconv(W) (mmgt_gn_silu_conv3x3_unet_image_bytes(C0 + C1, Cout) bytes); bias (Cout), bias2 (rows, Cout) fp32 or
 * null (the time-embedding projection: image n takes row n / b2_imgs); residual / out (nb, H, W, Cout) bf16.  bf16 only; everything else runs
 * mmgt_groupnorm_nhwc -> mmgt_conv3x3_nhwc. */
long mmgt_gn_silu_conv3x3_unet_image_bytes(int cin, int cout);
/* ... with the statistics of the GroupNorm that reads `out` from the launch's epilogue (the next leg's norm2, resnet.py:231): `stats` (or NULL) receives
 * [3][nb (H / 16) (W / 16) (16 / rows)][Cout] floats = (pivot, sum, sum of squares of the deviations from the pivot) of the bf16 values stored, per partial
 * of rows x 16 pixels and channel; rows = mmgt_gn_silu_conv3x3_unet_stats_rows(nb, H, W, Cout) (4 or 2: the cut the launch takes for this shape).
 * mmgt_gn_stats_finalize_unet combines the partials (Chan's pairwise update: no cancellation whatever the mean) into scale | shift (2, nb, C) of
 * GroupNorm(G, gamma, beta, eps) over the stored tensor (count = 16 rows values per partial and channel): no statistics pass over the tensor. */
int mmgt_gn_silu_conv3x3_unet_stats_rows(int nb, int H, int W, int cout);
int mmgt_gn_silu_conv3x3_unet_stats(const void* x0, int C0, const void* x1, int C1, const float* scale_shift, const void* wimg, const float* bias,
                                    const float* bias2, int b2_imgs, const void* residual, void* out, float* stats, int nb, int H, int W, int cout, int dtype,
                                    void* stream);
int mmgt_gn_stats_finalize_unet(const float* stats, const float* gamma, const float* beta, float* scale_shift, int nb, int parts_per_img, int count, int C,
                                int G, float eps, void* stream);
/* Debug (tools/trace_rconv.py): a device buffer of [workgroups][512] u64 receives 100-MHz stamps of the launches that follow; NULL = off. */
void mmgt_rconv_set_trace(void* buf);
int mmgt_gn_silu_conv3x3_unet(const void* x0, int C0, const void* x1, int C1, const float* scale_shift, const void* wimg, const float* bias,
                              const float* bias2, int b2_imgs, const void* residual, void* out, int nb, int H, int W, int cout, int dtype, void* stream);

/* ---- conditioning producers and the output path on the device; uint8 image buffers are device pointers.
 * blur_mask: (frames, H, W) u8 -> (frames, 64, 64) u8 = cv2.resize(64x64, bilinear) -> cv2.GaussianBlur(ksize, sigma from ksize,
 * reflect-101) -> cv2.normalize(MINMAX, 0, 255): scripts/pose2vid.py:94-114, scripts/audio2vid.py:131-151. */
int mmgt_blur_mask_u8(const unsigned char* masks, unsigned char* out, int frames, int H, int W, int ksize, void* stream);
/* (frames, S, S) u8 -> (frames, D, D): PIL's two-pass 8-bit resampling with the caller's integer coefficient tables
 * (bounds[D][2] = first tap, tap count; coeffs[D][ksize], 22 fractional bits), written as float / 255 (ToTensor) and / or u8:
 * torchvision Resize + ToTensor of src/dataset/image_processor.py:75-102,311-333. */
int mmgt_resample_u8(const unsigned char* in, float* out_f32, unsigned char* out_u8, int frames, int S, int D, const int* bounds,
                     const int* coeffs, int ksize, void* stream);
/* out[f][j] = x[clamp(f + j - half, 0, frames - 1)], j = 0 .. 2 half, rows of D floats: process_audio_emb, scripts/pose2vid.py:72-91. */
int mmgt_window_stack(const float* x, float* out, int frames, long D, int half, void* stream);
/* out (npix, 3) u8 = trunc(clamp(x[p][0..2] * scale + shift, 0, 1) * 255) from channels-last T (npix, cpad): decode_latents'
 * (x / 2 + 0.5).clamp(0, 1) (pipeline_pose2vid_long.py:121-123) fused with save_videos_grid's (x * 255).astype(uint8)
 * (src/utils/util.py:148-160). */
int mmgt_frames_to_u8(const void* x, unsigned char* out, long npix, int cpad, float scale, float shift, int dtype, void* stream);
/* SMGA key points -> the four frame streams of Stage 2, drawn on the device: kp (frames, 134, 3) fp32 = SMGA's normalised
 * (x, y, score) features.  Replaces data/extract_movment_mask_all.py:319-321 `pose_vid_generator` (denormalize :128-132, mask_leg :66-89,
 * process_keypoints :98-119), src/dwpose/__init__.py:220-283 `DWposeDetector_movment_mask.__call__` with draw_pose / draw_pose_mask_head /
 * _lips / _hand (:133-196) and src/dwpose/util.py:79-157,160-206,208-230,291-302,349-388 (cv2.ellipse2Poly, fillConvexPoly, line, circle
 * restated), and the mp4 write / read round trip of scripts/audio2vid.py:386,426-430.  pose (frames, 512, 512, 3) u8; hands / lips / face
 * masks (frames, 512, 512) u8 (face = face box + hand boxes with the reference's uint8 wrap-around).  H = W = 512 (the reference's canvas). */
int mmgt_dwpose_draw(const float* kp, unsigned char* pose, unsigned char* hands_mask, unsigned char* lips_mask, unsigned char* face_mask,
                     int frames, int H, int W, void* stream);

#ifdef __cplusplus
}
#endif
#endif
