/* objcavit_hip.h -- C ABI of libobjcavit_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the forward depth-inference hot path of ObjCAViT.  The
 * reference has no FFI of its own: the path sits behind Python
 * nn.Module.forward() calls that reach ATen kernels.  Each entry point below
 * replaces the ATen work behind the cited reference lines (paths relative to
 * the reference tree) and is what a ctypes binding in the reference's modules
 * would bind.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 (or uint8 for masks) unless
 *     stated otherwise; tensors are dense row-major with the strides given;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - entry points only enqueue work: no allocation, no synchronisation, no
 *     host<->device copies, so a caller may capture them into a hipGraph;
 *     scratch memory is supplied by the caller (`*_workspace_bytes`);
 *   - return 0 on success, a hipError_t (> 0) if a launch failed, -1 for a
 *     rejected argument; ocv_last_error() returns the message (thread-local).
 *   - results are fp32, and so are storage and accumulation.  A contraction runs in one of four forms, named by its entry point:
 *       exact fp32     v_mfma_f32_32x32x2_f32 (the *_exact_* / unsuffixed token entry points, OCV_* = exact / fp32 routes);
 *       fp16 pairs     every operand v = hi + lo with hi = fp16(v), lo = fp16(v - hi) (or lo' = fp16((v - hi) 2^11) for the token
 *                      kernels), every product hi*hi + hi*lo + lo*hi on v_mfma_f32_*_f16 with fp32 accumulation: relative error
 *                      of a product <= 2^-22, range +-65504 (range guard below).  THE DEFAULT of the decoder's and heads' 3x3 / 1x1
 *                      convolutions (f16 = 1 of the *_x_fwd entry points), the token stacks' layer tails, the self-attention core,
 *                      the few-key cross-attention and the bin head;
 *       bf16 pairs     the same with bf16 terms: <= 2^-17 per product, fp32's range.  The encoder's 1x1 layers, and the route every
 *                      fp16-pair kernel falls back to when the range guard trips (f16 = 0);
 *       bf16 triples   three terms, six products ("split3"): 2^-24, fp32's range: the token linears that need both.
 */
#ifndef OBJCAVIT_HIP_H
#define OBJCAVIT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ocv_stream_t;

#define OCV_ABI_VERSION 5 /* 5: round 6 ADDED ocv_attention_set_dispatch (the library reads no environment variable any more),
 * ocv_conv3x3_packed_taps_k / ocv_conv3x3_split_packed_taps_fwd; nothing removed or changed: a caller built against 4 keeps working;
 * 4: round 5 REMOVED the opt-in entry points that lost their A/Bs (ocv_tap_interp_skip_fwd, the squeeze-excite tail
 * family ocv_*_se_fwd / ocv_se_fold_gate_weights_fwd / ocv_se_tail_supported, ocv_conv3x3_winograd_split_fwd F(2x2)) and added the range
 * guard (ocv_range_flag_set, ocv_range_flag_take_fwd, ocv_attention_set_fp32_range), bin head route 4, ocv_mha_few_keys_h2_set_dispatch;
 * 2: ocv_encoder_layer_params starts with struct_size (round 3); 3: ocv_patch_embed_split_fwd and
                           ocv_conv3x3_winograd43_split_fwd take the split operands' element type (+ cscale / oscale), round 4 */
int ocv_abi_version(void);
const char* ocv_last_error(void);

/* Range guard of the fp16 pairs (round 5).  The reference's convolutions are nn.Conv2d in fp32 for ANY input
 * (modules/DenseFeatureExtractor.py:37-47,104-118); the two-term fp16 split of the decoder's activations ends at +-65504.  Every
 * launcher that writes fp16 "hl32" pairs (ocv_conv_nhwc_split_x_fwd, ocv_conv3x3_winograd43_split_fwd,
 * ocv_upsample_concat_split_x_fwd, ocv_tap_interp_*_fwd) ORs 1 into the word the CALLING THREAD armed with ocv_range_flag_set when
 * a value it converts exceeds 65504 / 16 = 4094 in magnitude (the first-batch calibration's own limit) (an atomic on that rare branch only; NULL = not armed, the default).  The word is
 * device memory owned by the caller, sticky until the caller clears it; ocv_range_flag_take_fwd copies it to `out` and zeroes it
 * on the stream (one tiny launch: capturable).  The host reads `out` where it reads results and re-runs the batch on bf16 pairs
 * (objcavit_amd/hip_ops/_core.py: RangeGuard, guarded_forward, bf16_pairs; objcavit_amd/graph.py GraphedGraphBins.checked).  Both
 * return 0 / -1. */
int ocv_range_flag_set(unsigned* flag);
int ocv_range_flag_take_fwd(unsigned* flag, unsigned* out, ocv_stream_t stream);
/* on != 0: ocv_attention_fwd (and every composite built on it) issued by THIS THREAD takes the exact-fp32 core whatever
 * ocv_attention_set_dispatch says -- the two-term fp16 core converts projected queries / keys / values to fp16 and ends at +-65504 like the
 * fp16 pairs; the range guard's fallback route (hip_ops.bf16_pairs) switches it together with them.  Returns 0. */
int ocv_attention_set_fp32_range(int on);
/* Core of ocv_attention_fwd and of every composite built on it (ocv_mha_fwd, ocv_mha_split3_fwd beyond 32 keys, ocv_encoder_layer_fwd,
 * ocv_encoder_stack_fwd): form 0 = two-term fp16 products on v_mfma_f32_32x32x16_f16 (default), 1 = exact fp32 MFMA (the A/B numerics
 * route).  Process-wide, like the other *_set_dispatch entry points; the library reads no environment variable.  Returns 0 / -1. */
int ocv_attention_set_dispatch(int form);

/* activation codes for ocv_linear_fwd */
#define OCV_ACT_NONE 0
#define OCV_ACT_RELU 1
#define OCV_ACT_LEAKY_RELU 2 /* slope 0.01 (nn.LeakyReLU default) */
#define OCV_ACT_SILU 3       /* x * sigmoid(x); convolution entry points only */
#define OCV_ACT_SIGMOID 4    /* ocv_pointwise_conv_nhwc_fwd only (squeeze-excite gate) */

/* out[z][m][n] = act( sum_k A[z][m][k] * W(n,k) + bias[n] )
 * W(n,k) = W[z][n*ldw + k] if w_kn == 0 (nn.Linear layout, [N,K])
 *        = W[z][k*ldw + n] if w_kn != 0 ([K,N] layout).
 * batch strides (elements) may be 0 to share an operand across the batch.
 * Replaces nn.Linear / nn.Sequential(Linear, LeakyReLU, ...) at
 * modules/ObjCAViT.py:257-282,289,299-303,324-330,378 and
 * modules/miniViT.py:16-20,33; also used for the in/out projections of
 * nn.MultiheadAttention and linear1 of nn.TransformerEncoderLayer. */
int ocv_linear_fwd(const float* A, int lda, long strideA, const float* W, int ldw, long strideW, int w_kn,
                   const float* bias, float* out, int ldo, long strideO, int batch, int M, int N, int K, int act,
                   ocv_stream_t stream);

/* out[m][:] = LayerNorm( residual[m][:] + A[m][:] W^T + bias ) * gamma + beta,  N == E <= 128... exactly 128.
 * zero_row_mask (uint8[M], nullable): rows with a non-zero entry are written as 0.0
 * (nested-tensor fast path of nn.TransformerEncoder).
 * Replaces out_proj + residual + norm1 and linear2 + residual + norm2 of
 * nn.TransformerEncoderLayer (modules/ObjCAViT.py:155-161,169,188; modules/layers.py:8-9,23). */
int ocv_linear_residual_layernorm_fwd(const float* A, int lda, const float* W, int ldw, const float* bias,
                                      const float* residual, int ldres, const float* gamma, const float* beta,
                                      float eps, const uint8_t* zero_row_mask, float* out, int ldo, int M, int N,
                                      int K, ocv_stream_t stream);

/* Fused feed-forward block of a post-norm transformer layer, hidden activations never leave the chip:
 *   out[m][:] = LayerNorm( x[m][:] + W2 relu(W1 x[m][:] + b1) + b2 ) * gamma + beta
 * x / out [M, E] dense (out may alias x), w1 [FF, E], w2 [E, FF] (nn.Linear layout); E == 128, FF % 128 == 0.
 * Replaces linear1 + ReLU + linear2 + residual + norm2 of nn.TransformerEncoderLayer
 * (modules/ObjCAViT.py:155-161,169,188; modules/layers.py:8-9,23). */
int ocv_ffn_residual_layernorm_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                                   const float* gamma, const float* beta, float eps, const uint8_t* zero_row_mask,
                                   float* out, int M, int E, int FF, ocv_stream_t stream);

/* out[m][:] = LayerNorm(x[m][:] (+ residual[m][:])) * gamma + beta over E columns (biased variance).
 * residual nullable.  Stand-alone form of the LayerNorm used above. */
int ocv_layernorm_residual_fwd(const float* x, const float* residual, const float* gamma, const float* beta, float eps,
                               float* out, int rows, int E, ocv_stream_t stream);

/* Attention core: ctx[b, i, h*d : (h+1)*d] = softmax_j( q_bih . k_bjh * scale + mask_bj ) v_bjh,  d == 32.
 * q/k/v/ctx are addressed as ptr + b*batch_stride + i*seq_stride + h*32 (strides in elements), so packed
 * [.., 3E] projections and seq-first (S x B x E) layouts need no copies.
 * key_padding_mask: uint8 [B, Sk], non-zero = ignore key (nullable).  A query row whose keys are ALL masked
 * yields NaN, as torch does. */
int ocv_attention_fwd(const float* q, long q_bs, int q_ss, const float* k, long k_bs, int k_ss, const float* v,
                      long v_bs, int v_ss, const uint8_t* key_padding_mask, float* ctx, long o_bs, int o_ss, int B,
                      int H, int Sq, int Sk, float scale, ocv_stream_t stream);

/* nn.MultiheadAttention(E, H, batch_first=True).forward(query, key, value, key_padding_mask, need_weights=False)
 * (modules/ObjCAViT.py:163-164,195-207): packed in_proj_weight [3E,E] / in_proj_bias [3E], out_proj [E,E] + [E].
 * q_src [B,Sq,E], k_src / v_src [B,Sk,E] dense; out [B,Sq,E].  E == 128, H == 4.
 * kv_limit: 0, or a number of leading keys such that EVERY key j >= kv_limit is masked in every batch row (the
 * caller knows it from the object counts: key_padding_mask is True for j >= n_b, so kv_limit = max_b n_b).  Keys
 * beyond it are then neither projected nor scored -- identical result, since they carry zero probability.  * With E = 128, H = 4 and at most 32 live keys (kv_limit <= 32, or Sk <= 32) the whole operation is one launch
 * (projections, scores, softmax, context, output projection per 32-query tile); otherwise five. */
size_t ocv_mha_workspace_bytes(int B, int Sq, int Sk, int E);
int ocv_mha_fwd(const float* q_src, const float* k_src, const float* v_src, const uint8_t* key_padding_mask,
                const float* in_proj_w, const float* in_proj_b, const float* out_w, const float* out_b, float* out,
                int B, int Sq, int Sk, int kv_limit, int E, int H, void* workspace, size_t workspace_bytes,
                ocv_stream_t stream);

/* ocv_mha_fwd with the four projections as three-term bf16 splits (fp32-faithful; weights packed once by
 * ocv_pack_split3_fwd: in_proj_p3 = packed in_proj_weight [3E, E], out_proj_p3 = packed out_proj.weight [E, E]); QK^T and
 * PV stay on exact fp32 MFMA.  With at most 32 live keys (the image <- object cross-attention, modules/ObjCAViT.py:192-201)
 * K and V are projected ONCE per image (a 32 KB record per image in the workspace, in the order the query tiles' lanes read it) and every 32-query tile is one fused launch
 * reading it; otherwise split3 linears around ocv_attention_fwd.  Same workspace size as ocv_mha_fwd. */
int ocv_mha_split3_fwd(const float* q_src, const float* k_src, const float* v_src, const uint8_t* key_padding_mask,
                       const void* in_proj_p3, const float* in_proj_b, const void* out_proj_p3, const float* out_b, float* out,
                       int B, int Sq, int Sk, int kv_limit, int E, int H, void* workspace, size_t workspace_bytes,
                       ocv_stream_t stream);

/* nn.MultiheadAttention forward with AT MOST 32 LIVE KEYS per image (kv_limit, or Sk itself, <= 32; E = 128, H = 4: the image <-
 * object cross-attention, modules/ObjCAViT.py:192-201) with EVERY contraction -- the four projections, Q K^T and P V -- as a
 * two-term fp16 split: x = hi + 2^-11 lo', hi = fp16(x), lo' = fp16((x - hi) 2^11), three v_mfma_f32_32x32x16_f16 per product
 * block (hi hi into one accumulator, hi lo' + lo' hi into a second that is added scaled by 2^-11); products carry 22 bits,
 * measured error = that of an fp32 FMA chain.  fp16's range applies: an activation beyond +-65504 turns its output rows inf / NaN
 * (packed weights saturate there); ocv_mha_split3_fwd has fp32's range.
 *   ocv_pack_split_h2_fwd: W [N][K] fp32 -> ocv_split_h2_packed_elems(N, K) fp16, 16-byte aligned, laid out
 *     packed[((jt * nsteps + s) * 2 + part) * 512 + lane * 8 + e] = part(W[32 jt + (lane & 31)][16 s + 8 (lane >> 5) + e]).
 *   in_proj_h2 / out_proj_h2 = packed in_proj_weight [3E, E] / out_proj.weight [E, E]; workspace: the K / V record,
 *   ocv_mha_few_keys_h2_workspace_bytes(B); every pointer 16-byte aligned.  ONE launch while the call is
 *   small (ceil(Sq / 32) B <= 320 workgroups: each tile projects its image's K / V itself -- same arithmetic, bit-identical
 *   results, the record unused), two launches beyond (K / V once per image, then the tiles).
 *   ocv_mha_few_keys_h2_set_dispatch(n): one launch up to n workgroups (0 = always two launches, < 0 = the default 320). */
int ocv_mha_few_keys_h2_set_dispatch(int one_launch_max_workgroups);
size_t ocv_split_h2_packed_elems(int N, int K);
int ocv_pack_split_h2_fwd(const float* W, int ldw, int N, int K, void* packed, ocv_stream_t stream);
size_t ocv_mha_few_keys_h2_workspace_bytes(int B);
int ocv_mha_few_keys_h2_fwd(const float* q_src, const float* k_src, const float* v_src, const uint8_t* key_padding_mask,
                            const void* in_proj_h2, const float* in_proj_b, const void* out_proj_h2, const float* out_b,
                            float* out, int B, int Sq, int Sk, int kv_limit, int E, int H, void* workspace,
                            size_t workspace_bytes, ocv_stream_t stream);

/* One post-norm nn.TransformerEncoderLayer(E=128, H=4, FF, relu, eps) in eval mode on x [B,S,E] (dense):
 *   x1 = LN1(x + MHA(x, x, x, mask));  out = LN2(x1 + W2 relu(W1 x1 + b1) + b2)
 * (modules/ObjCAViT.py:155-161,169,188; modules/layers.py:8-9,23).  `out` may alias `x`.
 * zero_padded_rows != 0: rows flagged in key_padding_mask are written as 0.0 in `out` (last layer of a masked
 * encoder). */
typedef struct {
  /* = sizeof(ocv_encoder_layer_params) AS THE CALLER WAS COMPILED.  The library reads only the first struct_size bytes
   * and treats every field beyond them as NULL, so the struct can grow at its end without breaking callers built
   * against an earlier header (ABI 1 had no such field and grew by the four *_p3 pointers: a silent break).  In an
   * array of layers (ocv_encoder_stack_fwd) consecutive elements lie struct_size bytes apart. */
  size_t struct_size;
  const float *in_proj_w, *in_proj_b, *out_proj_w, *out_proj_b;
  const float *norm1_w, *norm1_b, *linear1_w, *linear1_b, *linear2_w, *linear2_b, *norm2_w, *norm2_b;
  /* optional: the four weight matrices packed by ocv_pack_split3_fwd (all four or none).  With them the projections and
   * the feed-forward block run as the three-term bf16 split (fp32-faithful, 2.7x the matrix rate); NULL = exact fp32. */
  const void *in_proj_p3, *out_proj_p3, *linear1_p3, *linear2_p3;
  /* optional (round 3; callers built before it pass a shorter struct_size and these read as NULL): the same four matrices
   * packed by ocv_pack_split_h2_fwd (all four or none).  With them ocv_encoder_stack_fwd runs every layer's token-local tail
   * (output projection, both LayerNorms, feed-forward block, the next layer's q | k | v projection) as two-term fp16 splits
   * (ocv_layer_tail_h2_fwd: half the matrix operations and two thirds of the weight stream of the three-term form; fp16's range). */
  const void *in_proj_h2, *out_proj_h2, *linear1_h2, *linear2_h2;
} ocv_encoder_layer_params;
size_t ocv_encoder_layer_workspace_bytes(int B, int S, int E, int FF);
int ocv_encoder_layer_fwd(const float* x, const ocv_encoder_layer_params* p, const uint8_t* key_padding_mask,
                          int zero_padded_rows, float* out, int B, int S, int E, int H, int FF, float eps,
                          void* workspace, size_t workspace_bytes, ocv_stream_t stream);

/* Three-term bf16 split ("split3") forms of the token-path linear layers: every operand v = h + m + l (h = bf16(v),
 * m = bf16(v - h), l = bf16(v - h - m): 24 significant bits), every product as the six terms whose magnitude exceeds
 * 2^-24 of it, on v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- results equal to the exact-fp32 kernels' to fp32
 * rounding at 2.7x their matrix rate (and ~1/2.7 of their matrix-core energy).  The static weight W [N][K] (nn.Linear
 * layout, row stride ldw) is split and packed once per weight version by ocv_pack_split3_fwd into MFMA B-operand
 * fragments, ocv_split3_packed_elems(N, K) bf16 elements:
 *   packed[((jt * ceil(K/16) + s) * 3 + part) * 512 + lane * 8 + e] = part of W[32 jt + (lane & 31)][16 s + 8 (lane >> 5) + e]
 * (zero beyond N / K).  K a multiple of 8.  Same reference lines as ocv_linear_fwd /
 * ocv_linear_residual_layernorm_fwd / ocv_ffn_residual_layernorm_fwd. */
size_t ocv_split3_packed_elems(int N, int K);
int ocv_pack_split3_fwd(const float* W, int ldw, int N, int K, void* packed, ocv_stream_t stream);
int ocv_linear_split3_fwd(const float* A, int lda, const void* w_packed, const float* bias, float* out, int ldo, int M, int N,
                          int K, int act, ocv_stream_t stream);
int ocv_linear_residual_layernorm_split3_fwd(const float* A, int lda, const void* w_packed, const float* bias,
                                             const float* residual, int ldres, const float* gamma, const float* beta,
                                             float eps, const uint8_t* zero_row_mask, float* out, int ldo, int M, int N,
                                             int K, ocv_stream_t stream);
size_t ocv_ffn_split3_workspace_bytes(int M, int FF);
int ocv_ffn_residual_layernorm_split3_fwd(const float* x, const void* w1_packed, const float* b1, const void* w2_packed,
                                          const float* b2, const float* gamma, const float* beta, float eps,
                                          const uint8_t* zero_row_mask, float* out, int M, int E, int FF, void* workspace,
                                          size_t workspace_bytes, ocv_stream_t stream);

/* Everything of a post-norm transformer layer that is local to a token, one launch per 32-token tile (split3 arithmetic):
 *   x1 = LayerNorm1(x + ctx Wo^T + bo);  out = LayerNorm2(x1 + W2 relu(W1 x1 + b1) + b2)   (rows with zero_row_mask != 0: 0)
 *   qkv_next[M][3E] = out . Wqkv_next^T + bqkv_next      when next_in_proj_p3 / next_in_proj_b / qkv_next are given
 * ctx = the attention output of the layer, x its input; p: the layer's parameters with the packed *_p3 weights set. */
int ocv_layer_tail_split3_fwd(const float* ctx, const float* x, const ocv_encoder_layer_params* p, const void* next_in_proj_p3,
                              const float* next_in_proj_b, float eps, const uint8_t* zero_row_mask, float* out,
                              float* qkv_next, int M, int E, int FF, ocv_stream_t stream);
/* The same on the two-term fp16 weights (params->out_proj_h2 / linear1_h2 / linear2_h2, next_in_proj_h2 = the NEXT layer's packed
 * in_proj or NULL); csrc/token_h2.hip. */
int ocv_layer_tail_h2_fwd(const float* ctx, const float* x, const ocv_encoder_layer_params* p, const void* next_in_proj_h2,
                              const float* next_in_proj_b, float eps, const uint8_t* zero_row_mask, float* out,
                              float* qkv_next, int M, int E, int FF, ocv_stream_t stream);
/* The same with the feed-forward chunks of every 32-token row block shared out over G workgroups where the launch would leave
 * most of the chip idle (few tokens: the reference's own batch of 1 - 2 images); the row block's last workgroup to arrive adds the
 * partial sums in a fixed order and finishes the layer (one launch, nobody waits).  ocv_layer_tail_h2_groups: G for (M, FF) -- 8 up
 * to 24 row blocks, 4 up to 56, else 1.  workspace: ocv_layer_tail_h2_workspace_bytes (0 when G = 1) = the partial sums
 * followed by one arrival ticket per row block; THE TICKETS MUST BE ZERO WHEN THE CALL STARTS and are zero again when it has
 * finished (ocv_encoder_stack_fwd, which uses this form for batches of up to 4 sequences, clears them itself).  workspace NULL / too small: one workgroup per row block, as above.
 * Results are bit-reproducible for a given (M, FF); they differ from the G = 1 form in the last bits (summation order). */
int ocv_layer_tail_h2_groups(int M, int FF);
size_t ocv_layer_tail_h2_workspace_bytes(int M, int FF);
int ocv_layer_tail_h2_ws_fwd(const float* ctx, const float* x, const ocv_encoder_layer_params* p, const void* next_in_proj_h2,
                             const float* next_in_proj_b, float eps, const uint8_t* zero_row_mask, float* out,
                             float* qkv_next, int M, int E, int FF, void* workspace, size_t workspace_bytes, ocv_stream_t stream);
/* nn.TransformerEncoder(layer, n_layers) forward (eval, post-norm, batch-first [B,S,E]) in 1 + 2 n_layers launches: the
 * packed projection of layer 0, then per layer ocv_attention_fwd and ocv_layer_tail_split3_fwd.  Every layer needs its
 * packed split3 weights.  Semantics of key_padding_mask / zero_padded_rows as in ocv_encoder_layer_fwd (the zeros are
 * written by the last layer).  Replaces the nn.TransformerEncoder calls at modules/ObjCAViT.py:169,188 and
 * modules/layers.py:23. */
size_t ocv_encoder_stack_workspace_bytes(int B, int S, int E);
int ocv_encoder_stack_fwd(const float* x, const ocv_encoder_layer_params* layers, int n_layers, const uint8_t* key_padding_mask,
                          int zero_padded_rows, float* out, int B, int S, int E, int H, int FF, float eps, void* workspace,
                          size_t workspace_bytes, ocv_stream_t stream);

/* Patch embedding: Conv2d(C -> E, kernel = stride = 16, no padding) on fmap [B,C,h,w] (NCHW), flattened to tokens,
 * plus bias and positional embedding, written token-major:
 *   out[b][s][e] = bias[e] + pos[b*pos_bs + s*E + e] + sum_{c,i,j} W[e][c][i][j] * fmap[b][c][16*ph+i][16*pw+j]
 * with s = ph*(w/16) + pw.  pos_bs = 0 shares one [S,E] table across the batch; pos may be NULL.
 * channels_last != 0: fmap is stored [B,h,w,C] (torch channels_last) and W in its channels_last storage order
 * [E,16,16,C]; otherwise fmap is [B,C,h,w] and W [E,C,16,16].  (The same flag on the pixel-dot / bin-head entry
 * points selects [B,P,C] vs [B,C,P] for feat; outputs are always NCHW.)
 * Replaces image_embedding_convPxP + flatten + pos add + permute (modules/ObjCAViT.py:287-288,333,362-364;
 * modules/layers.py:11-12,17-22).  E == 128. */
size_t ocv_patch_embed_workspace_bytes(int B, int C, int h, int w, int E);
int ocv_patch_embed_fwd(const float* fmap, int channels_last, const float* W, const float* bias, const float* pos,
                        long pos_bs, float* out, int B, int C, int h, int w, int E, void* workspace,
                        size_t workspace_bytes, ocv_stream_t stream);

/* The same patch embedding on a feature map that is already stored in the "hl32" split-bf16 layout (below: what the
 * decoder's last convolution leaves beside its fp32 result), as 16 two-term-split GEMMs -- one per patch row ky -- in ONE launch
 * of the LDS-DMA convolution kernel (since round 4 their K steps are one axis cut in a batch-dependent number of pieces: the raw
 * partial results, 1 ... 64 of them, go to the workspace), summed in a fixed order with bias and positional embedding.  In that layout the 16 x C
 * values of one patch row are contiguous and the patches of an image row follow each other, so the map is read in place.
 * Products hi*hi + hi*lo + lo*hi (error <= 2^-17 per product, as in every convolution of the path), fp32 accumulation.
 *   x_hl  [B][h][w][2 C] bf16 (C a multiple of 32);   w_hi / w_lo [16 (ky)][E][16 C] bf16, column kx*C + c =
 *   bf16 split of W[e][c][ky][kx];   pos / pos_bs / out as above;   E a multiple of 8;   h a multiple of 16 when B > 1.
 *   workspace: ocv_patch_embed_split_workspace_bytes (the raw partial results).
 *   f16 / oscale: the element type of x_hl, w_hi, w_lo (0 = bf16 pairs, 1 = fp16 pairs) and the nullable per-output-channel
 *   factor [E] on the raw sums, as ocv_conv_nhwc_split_x_fwd (ABI 3). */
size_t ocv_patch_embed_split_workspace_bytes(int B, int C, int h, int w, int E);
int ocv_patch_embed_split_fwd(const void* x_hl, int C, const void* w_hi, const void* w_lo, const float* oscale, int f16,
                              const float* bias, const float* pos, long pos_bs, float* out, int B, int h, int w, int E,
                              void* workspace, size_t workspace_bytes, ocv_stream_t stream);

/* PixelWiseDotProduct (modules/layers.py:31-36): ram[b][q][p] = sum_c feat[b][c][p] * queries[b][q][c].
 * feat [B,C,P] (NCHW with P = h*w), queries addressed as ptr + b*q_bs + q*q_ld + c, ram [B,Q,P].  C == Q == 128. */
int ocv_pixel_dot_fwd(const float* feat, int channels_last, const float* queries, long q_bs, int q_ld, float* ram,
                      int B, int C, int Q, int P, ocv_stream_t stream);

/* Fused bin head (modules/GraphBins.py:109-119 == modules/AdaBins.py:77-87 together with modules/layers.py:31-36):
 *   logits[b][k][p] = bout[k] + sum_q Wout[k][q] * ( sum_c feat[b][c][p] * queries[b][q][c] )
 *   depth[b][p]     = sum_k softmax_k(logits[b][:, p]) * centers[b][k]
 * One pass over feat; the 128-channel range-attention maps and the 256-bin logits / probabilities never reach
 * HBM.  The two contractions are associated as (Wout . queries[b]) . feat (workspace holds the folded
 * [B,256,128] matrix).  C == Q == 128, n_bins == 256. */
size_t ocv_bin_head_workspace_bytes(int B, int n_bins, int C);
/* The two stages of ocv_bin_head_fwd as separate calls (same arithmetic, lets a caller time the main kernel):
 *   ocv_bin_head_fold_fwd   Wf[b] = Wout (n_bins x Q) . queries[b] (Q x C)            -> Wf [B, n_bins, C]
 *   ocv_bin_head_folded_fwd depth[b][p] = sum_k softmax_k(bout + Wf[b] . feat[b][:, p]) * centers[b][k]
 * channels_last selects layout AND arithmetic: 0 = feat is NCHW, 1 = NHWC (both exact fp32 MFMA); 3 = NHWC with the logits as a
 * TWO-term fp16 split with a scaled low term (v = hi + 2^-11 lo', hi = fp16(v), lo' = fp16((v - hi) 2^11): three
 * v_mfma_f32_32x32x16_f16 per product block, 22-bit products = the error of an fp32 FMA chain; both parts of Wf[b] fit the LDS, so
 * one workgroup walks all 256 bins and the map is read once -- the fastest faithful form; fp16's range: a map value or folded
 * weight beyond +-65504 turns the pixel's depth inf / NaN); 4 = 3 with TWO-LEVEL logits: every bin coarsely first (the hi hi
 * product alone), the three-product logits and the softmax arithmetic only for the 32-bin tiles in which some pixel of the
 * wavefront's 32 has a bin within T = 24 (+ twice the coarse product's proven error bound) of its largest coarse logit -- the
 * bins left out weigh <= 224 e^-24 of a pixel's softmax; cost 64 + 24 n MFMAs per 32 pixels, n = tiles kept, against 192;
 * 2 = NHWC, three-term bf16 (fp32's range): ocv_bin_head_folded_ws_fwd only.  (Rounds 2 - 4 also carried a two-term bf16 form under code 2 -- 3x the depth error under near-one-hot softmaxes, opt-in,
 * never a default: removed in round 5.) */
int ocv_bin_head_fold_fwd(const float* queries, long q_bs, int q_ld, const float* Wout, float* Wf, int B, int C, int Q,
                          int n_bins, ocv_stream_t stream);
int ocv_bin_head_folded_fwd(const float* feat, int channels_last, const float* Wf, const float* bout,
                            const float* centers, float* depth, int B, int C, int n_bins, int P, ocv_stream_t stream);
/* The same with scratch for the per-half softmax states: channels_last == 2 = an NHWC map whose logits run
 * as a THREE-term bf16 split (v = h + m + l, six matrix-core products per product, dropped terms <= 2^-24: fp32-faithful)
 * at 2.7x the matrix rate of the exact fp32 kernel -- 256 bins in two halves of 128 (three parts of a half's folded
 * matrix fill 96 KB of LDS), merged per pixel by a second launch.  partials: ocv_bin_head_partials_bytes(B, P) bytes,
 * 16-byte aligned.  Any other channels_last code is handed to ocv_bin_head_folded_fwd. */
size_t ocv_bin_head_partials_bytes(int B, int P);
int ocv_bin_head_folded_ws_fwd(const float* feat, int channels_last, const float* Wf, const float* bout, const float* centers,
                               float* depth, int B, int C, int n_bins, int P, void* partials, size_t partials_bytes,
                               ocv_stream_t stream);
int ocv_bin_head_fwd(const float* feat, int channels_last, const float* queries, long q_bs, int q_ld, const float* Wout,
                     const float* bout, const float* centers, float* depth, int B, int C, int Q, int n_bins, int P,
                     void* workspace, size_t workspace_bytes, ocv_stream_t stream);

/* 3 x 3 convolution, stride 1, zero padding 1, of an image with 1..4 channels into Cout channels (multiple of 4): exact fp32
 * FMA, raw result (no bias, no activation) as NHWC fp32 y [B][H][W][Cout].  x is addressed by element strides (batch,
 * channel, row, column), so NCHW and channels_last images are read in place; w_taps [9][C][Cout] fp32 (tap t = 3 ky + kx
 * major, 16-byte aligned) = W[co][c][ky][kx] transposed.  The skip part of the last decoder stage of do_final_upscale models
 * (reference modules/DenseFeatureExtractor.py:99-101,116-117: the skip tensor is the input image); ocv_tap_interp_combine_fwd
 * adds it to the low-resolution half. */
int ocv_conv3x3_few_channels_fwd(const float* x, long stride_b, long stride_c, long stride_y, long stride_x, const float* w_taps,
                                 float* y, int B, int C, int H, int W, int Cout, ocv_stream_t stream);

/* Depthwise k x k convolution (k in {3,5}, stride in {1,2}) with explicit top/left zero padding (bottom/right
 * padding is implied by Ho/Wo -- covers TensorFlow "SAME"), per-channel bias (folded BatchNorm) and optional SiLU:
 *   out[b][c][y][x] = act( bias[c] + sum_{i,j} w[c][i][j] * in[b][c][y*stride - pad_t + i][x*stride - pad_l + j] )
 * in [B,C,H,W], w [C,k,k], out [B,C,Ho,Wo], NCHW fp32.  Replaces conv_dw + bn + act of the EfficientNet MBConv
 * blocks that the reference runs through its hub backbone (modules/DenseFeatureExtractor.py:18-27,149). */
int ocv_depthwise_conv_fwd(const float* in, const float* w, const float* bias, float* out, int B, int C, int H, int W,
                           int k, int stride, int pad_t, int pad_l, int Ho, int Wo, int act, ocv_stream_t stream);

/* NHWC building blocks of the EfficientNet MBConv stages (exact fp32):
 * pointwise (1x1) convolution as a row GEMM with everything around it fused:
 *   y[m][co] = act( bias[co] + sum_ci x[m][ci] * gate[m / rows_per_image][ci] * W[co][ci] ) + residual[m][co]
 * x [M, Cin] (= [B,H,W,Cin]), W [Cout, Cin], gate (nullable) [B, Cin] = squeeze-excite gate applied to the INPUT,
 * residual (nullable) [M, Cout]; Cin a multiple of 8; act any OCV_ACT_*.  Replaces conv_pw / conv_pwl / conv_head
 * + BatchNorm + SiLU + the SE multiply + the skip add of the hub backbone's blocks
 * (modules/DenseFeatureExtractor.py:18-27,149) and the SE fully-connected layers (with M = B). */
int ocv_pointwise_conv_nhwc_fwd(const float* x, const float* gate, int rows_per_image, const float* W,
                                const float* bias, const float* residual, float* y, long M, int Cin, int Cout, int act,
                                ocv_stream_t stream);

/* The same contraction on the bf16 matrix cores at fp32-level accuracy (split-bf16: every product is formed as
 * hi*hi + hi*lo + lo*hi with fp32 accumulation, relative error of a product <= 2^-17): identical contract, except that
 * the (static) weights arrive pre-split AND packed in matrix-core operand order, so that a wavefront's weight load is
 * one contiguous 1 KB run.  With w_hi = bf16(W), w_lo = bf16(W - w_hi), both zero-padded to [ceil32(Cout)][Kp],
 * Kp = ceil16(Cin):
 *   w_packed[(((jt * (Kp/16) + s) * 2 + part) * 64 + lane) * 8 + e] = w_part[32 jt + (lane & 31)][16 s + 8 (lane >> 5) + e]
 * (part 0 = hi, 1 = lo; bf16; ocv_pointwise_packed_weight_elems() elements).  The exact-fp32 entry point above is
 * MFMA-bound from stage 4 of the encoder on (47 TFLOP/s of 157); this one is the encoder's default. */
size_t ocv_pointwise_packed_weight_elems(int Cin, int Cout);
/* Diagnostics / tests: pin the kernel family ocv_pointwise_conv_nhwc_split_fwd dispatches to, for every later call of
 * this process: 0 = automatic (default), 1 = rows, 2 = stream, 3 = 32-row tile (a = wavefronts across channels 2|4 or 0,
 * b = K groups 1|2 or 0).  A family that cannot run a shape falls back to the automatic choice for that call.  Results
 * do not depend on it beyond fp32 summation order. */
int ocv_pointwise_split_set_dispatch(int family, int a, int b);
int ocv_pointwise_conv_nhwc_split_fwd(const float* x, const float* gate, int rows_per_image, const void* w_packed,
                                      const float* bias, const float* residual, float* y, long M, int Cin, int Cout,
                                      int act, ocv_stream_t stream);
/* ... that also leaves y_hl (nullable): the output in the hl32 split layout ([M][2 ceil32(Cout)] bf16, pad channels zero),
 * for a consumer that reads its rows by LDS-DMA (ocv_pointwise_hl_fwd).  Cout % 8 == 0 when y_hl is given. */
int ocv_pointwise_conv_nhwc_split_hl_fwd(const float* x, const float* gate, int rows_per_image, const void* w_packed,
                                         const float* bias, const float* residual, float* y, void* y_hl, long M, int Cin,
                                         int Cout, int act, ocv_stream_t stream);
/* The same with a caller-provided scratch of ocv_pointwise_split_workspace_bytes(M, Cin, Cout) bytes (0 for most shapes): where the
 * launch would be a handful of 32-row tiles walking a long K on an otherwise idle chip (a batch of 1 - 2 in the late encoder stages:
 * fewer than 128 tiles, K >= 1024) the K slabs of a tile are shared out over up to 8 workgroups that write raw partial tiles to the
 * scratch, and a second launch adds them in ascending order + bias + activation + residual (bitwise reproducible).  Without scratch
 * (or with y_hl) it is ocv_pointwise_conv_nhwc_split_hl_fwd. */
size_t ocv_pointwise_split_workspace_bytes(long M, int Cin, int Cout);
int ocv_pointwise_conv_nhwc_split_ws_fwd(const float* x, const float* gate, int rows_per_image, const void* w_packed,
                                         const float* bias, const float* residual, float* y, void* y_hl, long M, int Cin, int Cout,
                                         int act, void* workspace, size_t workspace_bytes, ocv_stream_t stream);
/* The same contraction on a row operand that is ALREADY split, in the "hl32" layout (below: per pixel and 32-channel block,
 * 32 hi then 32 lo bf16 values, pad channels zero) -- what the depthwise / project epilogues of the late encoder stages
 * write -- read by LDS-DMA with no conversion work, and with the squeeze-excite gate folded into PER-IMAGE weights
 * (ocv_se_gate_weights_fwd) instead of multiplied into the rows:
 *   y[m][co] = act( bias[co] + sum_ci x[m][ci] * Wimg[co][ci] ) + residual[m][co],  img = m / rows_per_image
 *   x_hl [M][2 ceil32(Cin)] bf16;  w_packed as above; w_image_elems = 0: one matrix for all rows, else the packed matrix
 *   of image i starts at w_packed + i * w_image_elems (>= ocv_pointwise_packed_weight_elems) and no tile spans two images
 *   (M a multiple of rows_per_image);  y (fp32 [M][Cout]) and / or y_hl (hl32 [M][2 ceil32(Cout)], pad channels written as
 *   zero) -- at least one;  Cin % 8 == 0, Cout % 4 == 0 (% 8 for y_hl).  Same reference lines as above.
 * ocv_pointwise_hl_set_dispatch(rt, tn): pin the wavefront tile (rt x tn blocks of 32 x 32; rt in {1,2,4}, tn in {1,2});
 * (0, 0) = automatic.  Diagnostics / tests only. */
int ocv_pointwise_hl_set_dispatch(int rt, int tn);
int ocv_pointwise_hl_fwd(const void* x_hl, int Cin, const void* w_packed, long w_image_elems, int rows_per_image,
                         const float* bias, const float* residual, float* y, void* y_hl, long M, int Cout, int act,
                         ocv_stream_t stream);
/* Stem convolution: dense 3x3 (Cin * 9 <= 32, Cout <= 64), any stride, explicit top/left zero padding (bottom/right
 * implied by Ho/Wo: TensorFlow "SAME"), + bias (folded BatchNorm) + activation; reads the NCHW image x [B,Cin,H,W] and
 * writes the NHWC activation y [B,Ho,Wo,Cout]; w [Cout][Cin*3*3] (PyTorch's weight, flattened).  Exact fp32.  Replaces
 * conv_stem + bn1 + act1 of the hub backbone (modules/DenseFeatureExtractor.py:18-27). */
int ocv_stem_conv_fwd(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int H, int W, int Cout,
                      int k, int stride, int pad_t, int pad_l, int Ho, int Wo, int act, ocv_stream_t stream);
/* depthwise k x k (k in {3,5}, stride in {1,2}) on NHWC: in [B,H,W,C], w [k*k][C], out [B,Ho,Wo,C]; C % 4 == 0.
 * Padding / bias / act as ocv_depthwise_conv_fwd. */
int ocv_depthwise_conv_nhwc_fwd(const float* in, const float* w, const float* bias, float* out, int B, int C, int H,
                                int W, int k, int stride, int pad_t, int pad_l, int Ho, int Wo, int act,
                                ocv_stream_t stream);
/* Depthwise convolution + bias + SiLU as ocv_depthwise_conv_nhwc_fwd, that ALSO emits the squeeze-excite pooling sums
 * of its own output, so the activation is not read a second time: part [B][tiles][C] holds one partial sum per
 * workgroup of the launch (tiles = ocv_depthwise_sum_tiles(...), fixed summation order -> reproducible), and
 * ocv_se_gate_partials_fwd turns the partials into the gate (hidden_ws: B * R floats of scratch, R <= 256):
 *   gate[b][c] = sigmoid( b2[c] + sum_r w2t[r][c] * silu( b1[r] + sum_c' w1[r][c'] * mean[b][c'] ) ),
 *   mean[b][c] = (sum_tile part[b][tile][c]) / pixels_per_image.
 * w1 [R][C], w2t [R][C] (= conv_expand's weight transposed).  Replaces conv_dw + bn + act + se.conv_reduce / act /
 * conv_expand / sigmoid of the hub backbone's blocks (modules/DenseFeatureExtractor.py:18-27,149). */
int ocv_depthwise_sum_tiles(int B, int C, int Ho, int Wo, int k, int stride);
int ocv_depthwise_conv_nhwc_sum_fwd(const float* in, const float* w, const float* bias, float* out, float* part, int B,
                                    int C, int H, int W, int k, int stride, int pad_t, int pad_l, int Ho, int Wo,
                                    ocv_stream_t stream);
int ocv_se_gate_partials_fwd(const float* part, int tiles, long pixels_per_image, const float* w1, const float* b1,
                             const float* w2t, const float* b2, float* gate, float* hidden_ws, int B, int C, int R,
                             ocv_stream_t stream);
/* ocv_depthwise_conv_nhwc_sum_fwd that writes its output (also, or only: out nullable) in the hl32 split layout: out_hl
 * [B][Ho][Wo][2 C] bf16, C a multiple of 32.  The values are split where they are produced; ocv_pointwise_hl_fwd reads them. */
int ocv_depthwise_conv_nhwc_sum_hl_fwd(const float* in, const float* w, const float* bias, float* out, void* out_hl, float* part,
                                       int B, int C, int H, int W, int k, int stride, int pad_t, int pad_l, int Ho, int Wo,
                                       ocv_stream_t stream);
/* ocv_se_gate_partials_fwd that FOLDS THE GATE INTO THE PROJECT WEIGHTS: for every image b
 *   w_packed + b * w_image_elems  <-  pack(split(W[n][k] * gate[b][k]))      (packed order of ocv_pointwise_conv_nhwc_split_fwd)
 * W fp32 [N][C] (the BN-folded conv_pwl weight), w_image_elems >= ocv_pointwise_packed_weight_elems(C, N) and a multiple of 8;
 * gate (nullable) also receives the [B][C] gate itself.  C a multiple of 8.  Two launches (hidden layer; gate + weights). */
int ocv_se_gate_weights_fwd(const float* part, int tiles, long pixels_per_image, const float* w1, const float* b1,
                            const float* w2t, const float* b2, const float* W, void* w_packed, long w_image_elems, float* gate,
                            float* hidden_ws, int B, int C, int R, int N, ocv_stream_t stream);


/* squeeze: out[b][c] = mean over the P = H*W pixels of x [B,P,C]; two-stage, fixed summation order. */
size_t ocv_channel_mean_workspace_bytes(int B, int C, long P);
int ocv_channel_mean_nhwc_fwd(const float* x, float* out, int B, int C, long P, void* workspace, size_t workspace_bytes,
                              ocv_stream_t stream);

/* squeeze-excite gate: gate[b][c] = sigmoid( b2[c] + sum_r w2t[r][c] * silu( b1[r] + sum_c' w1[r][c'] * mean[b][c'] ) );
 * mean / gate [B, C], w1 [R, C] (conv_reduce), w2t [R, C] (conv_expand weight TRANSPOSED), R <= 256;
 * hidden_ws: caller scratch of B*R floats. */
int ocv_se_gate_fwd(const float* mean, const float* w1, const float* b1, const float* w2t, const float* b2, float* gate,
                    float* hidden_ws, int B, int C, int R, ocv_stream_t stream);

/* Convolution k x k (k in {1,3}), stride 1, zero "same" padding, on NHWC fp32 activations, computed as an implicit
 * GEMM on the bf16 matrix cores with split-bf16 operands (x = hi + lo, 3 MFMAs per product, fp32 accumulate;
 * ~1e-6 relative error, see csrc/conv_igemm.hip):
 *   y[b][h][w][co] = act( bias[co] + sum_{ky,kx,ci} xcat[b][h+ky-k/2][w+kx-k/2][ci] * Wt[ky*k+kx][co][ci] ) (+ residual)
 * xcat is x1 [B,H,W,C1] followed along channels by the optional x2 [B,H,W,C2] (virtual concat: the skip connection of
 * UpSampleWithSkip, modules/DenseFeatureExtractor.py:44-47).  w_hi / w_lo: bf16 [k*k][Cout][Cp], Cp = C1+C2 rounded up
 * to 32, zero padded, w_hi = bf16(W), w_lo = bf16(W - w_hi) (prepared once on the host).  C1, C2 multiples of 4; C1 a
 * multiple of 32 when x2 is given.  residual (nullable) and y are [B,H,W,Cout].  act: OCV_ACT_*.
 * Replaces the conv3x3 (+ folded BN + LeakyReLU) stages of the UNet decoder (modules/DenseFeatureExtractor.py:37-42,97)
 * and ObjCAViT.conv3x3 / mViT.conv3x3 (modules/ObjCAViT.py:298,374; modules/miniViT.py:15,25). */
int ocv_conv_nhwc_fwd(const float* x1, int C1, const float* x2, int C2, const void* w_hi, const void* w_lo,
                      const float* bias, const float* residual, float* y, int B, int H, int W, int Cout, int ksize,
                      int act, ocv_stream_t stream);
/* The same convolution in EXACT fp32 (v_mfma_f32_32x32x2_f32: bit-for-bit a k-ordered fp32 fma chain, no split
 * operands), any odd k <= 7, any channel counts: the hand-written exact route for A/B numerics and for shapes the
 * split-bf16 kernels do not take (5x slower by construction; not on any default path).  w_tap_major: fp32
 * [k*k][Cout][C1+C2].  Replaces the same reference lines as ocv_conv_nhwc_fwd. */
int ocv_conv_nhwc_exact_fwd(const float* x1, int C1, const float* x2, int C2, const float* w_tap_major, const float* bias,
                            const float* residual, float* y, int B, int H, int W, int Cout, int ksize, int act,
                            ocv_stream_t stream);

/* Expand 1x1 convolution (+ bias = folded BN, SiLU) and the depthwise k x k convolution (+ bias, SiLU) behind it, fused:
 * the expanded tensor is never written to memory.  x [B,H,W,Cin] NHWC fp32, Cin a multiple of 8 in [24, 64];
 * w_packed = the expand weight [mid][Cin] in the packed split-bf16 layout of ocv_pointwise_conv_nhwc_split_fwd;
 * w_dw [k*k][mid], bias_dw [mid]; y [B,Ho,Wo,mid]; part [B][tiles][mid] = squeeze-excite pooling partials for
 * ocv_se_gate_partials_fwd with tiles = ocv_mbconv_expand_dw_tiles(Ho, Wo, k, stride).  k in {3,5}, stride in {1,2},
 * padding as ocv_depthwise_conv_fwd (out-of-image taps of the EXPANDED tensor are zero).  Numerics: expand as
 * ocv_pointwise_conv_nhwc_split_fwd (split-bf16, fp32 accumulate), depthwise exact fp32.  Replaces conv_pw + bn1 + act1 +
 * conv_dw + bn2 + act2 (+ the pooling of se) of the hub backbone's InvertedResidual blocks
 * (modules/DenseFeatureExtractor.py:18-27). */
int ocv_mbconv_expand_dw_tiles(int Ho, int Wo, int k, int stride);
int ocv_mbconv_expand_dw_fwd(const float* x, const void* w_packed, const float* bias_expand, const float* w_dw,
                             const float* bias_dw, float* y, float* part, int B, int H, int W, int Cin, int mid, int k,
                             int stride, int pad_t, int pad_l, int Ho, int Wo, ocv_stream_t stream);

/* Split-bf16 activation layout "hl32" shared by ocv_upsample_concat_split_fwd and ocv_conv_nhwc_split_fwd: for a
 * logical NHWC activation [B,H,W,C] one bf16 buffer [B*H*W][2*Cp], Cp = C rounded up to 32, holding per pixel and per
 * block of 32 channels the 32 hi values (hi = bf16(v)) followed by the 32 lo values (lo = bf16(v - hi)); element
 * (m, c, part) sits at  m * 2*Cp + (c / 32) * 64 + part * 32 + (c % 32).  Channels C..Cp-1 are stored as zeros.  One
 * (pixel, 32-channel) K step of the convolution is therefore one 128-byte line.  Returns the number of bf16 elements
 * of such a buffer (B*H*W*2*Cp), or 0 on bad sizes. */
size_t ocv_split_act_elems(int B, int H, int W, int C);

/* The same convolution on an input that is ALREADY split: x_hl in the hl32 layout above for Cin channels (128-byte
 * aligned), as produced by ocv_upsample_concat_split_fwd or by a previous convolution's y_hl.  Outputs: y (fp32
 * [B,H,W,Cout], nullable) and / or y_hl (hl32 layout for Cout channels, nullable; needs Cout a multiple of 32).  No
 * fp32->bf16 work per tap in the kernel. */
int ocv_conv_nhwc_split_fwd(const void* x_hl, int Cin, const void* w_hi, const void* w_lo, const float* bias,
                            const float* residual, float* y, void* y_hl, int B, int H, int W, int Cout, int ksize,
                            int act, ocv_stream_t stream);
/* The same with a caller-provided workspace of ocv_conv_nhwc_split_workspace_bytes(...) bytes (0 for most shapes): when
 * the tile count would leave the last round of workgroups mostly empty (the 30 x 40 stages of the decoder at B = 16:
 * 600 tiles on 256 CUs) the channel chunks are halved between two workgroups per tile, which write fp32 partial sums
 * to the workspace, and a second pass adds them in a fixed order and applies bias / activation / residual / split.
 * Without a workspace (or with one that is too small) it is ocv_conv_nhwc_split_fwd. */
size_t ocv_conv_nhwc_split_workspace_bytes(int B, int H, int W, int Cin, int Cout, int ksize);
int ocv_conv_nhwc_split_ws_fwd(const void* x_hl, int Cin, const void* w_hi, const void* w_lo, const float* bias,
                               const float* residual, float* y, void* y_hl, int B, int H, int W, int Cout, int ksize,
                               int act, void* workspace, size_t workspace_bytes, ocv_stream_t stream);
/* Round 4: the same convolution with the element type of the split operands as a parameter.
 *   f16 = 0: x_hl, w_hi, w_lo, y_hl hold bf16 (hi, lo) pairs -- products good to 2^-17, fp32's range (ocv_conv_nhwc_split_ws_fwd).
 *   f16 = 1: they hold FP16 pairs (hi = fp16(v), lo = fp16(v - hi), the low term unscaled): the same three matrix-core products
 *     per block on v_mfma_f32_16x16x32_f16, good to 2^-22 -- the stress-case margin of the depth map (4.5e-4 of the 1e-3 bar
 *     with bf16 pairs) was these products' alone.  fp16's range is handled explicitly: WEIGHTS are scaled per output channel by
 *     a power of two that puts the row's largest entry near 2^8 (out of the subnormals; hip_ops.prep_conv_weight) and
 *     oscale [Cout] (nullable = ones) = the inverse powers, applied to the raw accumulators in front of bias / activation (exact);
 *     ACTIVATIONS are stored unscaled: beyond +-65504 they become inf (the output is non-finite: loud), below ~0.1 their low
 *     term is subnormal, an ABSOLUTE error floor of 2^-25 per value that only matters for a tensor whose every entry is tiny
 *     (objcavit_amd.hip_ops.fp16_range_report checks a model's activations against both ends).
 * The producers of hl32 tensors take the same flag: ocv_upsample_concat_split_x_fwd, ocv_tap_interp_combine_x_fwd,
 * ocv_conv3x3_winograd43_split_fwd (hl_f16: its input AND its split output), ocv_patch_embed_split_fwd (reads). */
int ocv_conv_nhwc_split_x_fwd(const void* x_hl, int Cin, const void* w_hi, const void* w_lo, const float* oscale, int f16,
                              const float* bias, const float* residual, float* y, void* y_hl, int B, int H, int W, int Cout,
                              int ksize, int act, void* workspace, size_t workspace_bytes, ocv_stream_t stream);

/* The same 3 x 3 convolution (stride 1, zero padding 1) with PACKED TAPS (round 6): the K axis of the implicit GEMM is the nine taps'
 * REAL 8-channel granules laid end to end instead of nine chunks of ceil32(Cin) channels, so the matrix cores do not multiply the
 * zero pad channels of every tap -- 24 input channels: 7 K steps of 32 instead of 9; 40: 12 instead of 18 (the skip parts of the
 * decoder's first convolutions, modules/DenseFeatureExtractor.py:44-47 on the encoder's 24- and 40-channel features).  x_hl, outputs,
 * oscale, f16, bias, residual, act: as ocv_conv_nhwc_split_x_fwd (Cin % 8 == 0).  w_hi / w_lo: ONE [Cout][Kp] matrix each,
 * Kp = ocv_conv3x3_packed_taps_k(Cin) = ceil(9 (Cin / 8) / 4) * 32, element (co, 8 ((Cin / 8) t + g) + e) = weight[co][8 g + e] of
 * tap t = 3 ky + kx, zeros beyond the ninth tap.  Same products, same fp32 accumulation; the K order differs from the tap-major form,
 * so results agree to rounding, not bit for bit. */
int ocv_conv3x3_packed_taps_k(int Cin);
int ocv_conv3x3_split_packed_taps_fwd(const void* x_hl, int Cin, const void* w_hi, const void* w_lo, const float* oscale, int f16,
                                      const float* bias, const float* residual, float* y, void* y_hl, int B, int H, int W, int Cout,
                                      int act, ocv_stream_t stream);

/* The same 3 x 3 convolution (stride 1, zero padding 1) in Winograd F(4x4, 3x3) form on TWO-TERM FP16 splits (round 3), for shapes
 * where the arithmetic dominates the transforms' traffic: 4x fewer matrix-core operations than the direct form, a transformed input
 * of 2.25x the activation.  Input and output are the hl32 split / fp32 tensors of ocv_conv_nhwc_split_x_fwd; inside, the transformed input and filter are fp16 (hi, lo) pairs
 * (22-bit products: the transforms' ~100x error amplification stays at 2 - 3.5e-6 of max |y|, where two bf16 terms give 1e-4).
 *   u_hi, u_lo [36][Cout][Cp] fp16: U'[6 i + j][n][c] = (G g G^T)[i][j][n][c] * 2^k[6 i + j] * 2^-a[c] (fp64 transform; the
 *   power of two per POSITION puts the position's largest entry near 2^8, out of fp16's subnormals; the power of two per INPUT
 *   CHANNEL equalises the filters' columns), hi = fp16(U'), lo = fp16(U' - hi); fscale [36] fp32 = 2^-k; cscale [Cp] fp32 = 2^a
 *   (1 for pad channels; NULL = all ones), applied to the activations by the input transform.
 * Interpolation points 0, 1, -1, 2, -1/2, inf (G rows [1, a, a^2] / prod_{j != i}(a_i - a_j), last row [0 0 1]).  The
 * transformed input is up to 49x the activation (7x per 1-D pass) with an unscaled low term, so the input transform scales every
 * TILE by a power of two taken from that tile's largest input (49 amax 2^s in [2^14, 2^15); a GEMM row: undone exactly by the
 * output transform): no overflow and no subnormal low terms at ANY activation magnitude fp32 holds; inf / NaN inputs give
 * non-finite outputs.  What remains of fp16's range is the spread INSIDE one tile and channel: values 2^-13 below the tile's
 * largest lose low-term bits gradually (their products are that much smaller than the tile's result, too). */
size_t ocv_conv3x3_winograd43_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int ocv_conv3x3_winograd43_split_fwd(const void* x_hl, int Cin, const void* u_hi, const void* u_lo, const float* fscale,
                                     const float* cscale, const float* bias, float* y, void* y_hl, int B, int H, int W, int Cout,
                                     int act, int hl_f16, void* workspace, size_t workspace_bytes, ocv_stream_t stream);

/* Second half of "3 x 3 convolution of an up-sampled tensor, computed at the low resolution" (first convolution of every
 * UpSampleWithSkip stage: F.interpolate(bilinear, align_corners=True) + torch.cat + Conv2d(k=3) + BatchNorm + LeakyReLU,
 * modules/DenseFeatureExtractor.py:44-47,37-39).  Bilinear up-sampling is linear and per channel, the convolution mixes
 * channels per tap, so conv_{Wa}(up(x))[p] = sum_t sum_{4 nb} coef(p + t, nb) (Wa_t x[nb]): the caller forms the nine tap
 * products once per LOW-resolution pixel -- z [B,h,w,9 Cout] = ocv_conv_nhwc_split_fwd(ksize 1) of x with the weight rows
 * stacked tap-major, column t Cout + co -- and the skip part s [B,H,W,Cout] = conv3x3 over the skip channels (raw, nullable);
 * this entry point computes
 *   y[b][Y][X][co] = act( bias[co] + s[b][Y][X][co] + sum_t [ (Y,X) + t inside H x W ] bilinear(z[..][t Cout + co]; (Y,X) + t) )
 * with ATen's align_corners=True coefficients, writing fp32 y and / or the hl32 split y_hl.  Exact re-association of the
 * reference's arithmetic; ~4x fewer matrix-core operations for the up-sampled channels.  ocv_tap_interp_supported tells
 * whether (h,w) -> (H,W) is an up-sampling the staging buffer covers (ratios of ~2 and more).
 * zpad = 1 (Decoder.conv2, the 1 x 1 convolution with padding 1 in front of the first stage, :57,:105): the resize source is
 * the h x w grid whose one-pixel border ring holds one constant vector per column (zborder [9 Cout]: the tap products of
 * conv2's bias) and z stores only the (h-2) x (w-2) interior; zpad = 0: zborder NULL, z stores h x w. */
int ocv_tap_interp_supported(int h, int w, int H, int W, int Cout);
int ocv_tap_interp_combine_fwd(const float* z, int h, int w, int zpad, const float* zborder, const float* s, const float* bias,
                               float* y, void* y_hl, int B, int H, int W, int Cout, int act, ocv_stream_t stream);
/* the same with the element type of y_hl as a parameter (0 = bf16 pairs, 1 = fp16 pairs: ocv_conv_nhwc_split_x_fwd) */
int ocv_tap_interp_combine_x_fwd(const float* z, int h, int w, int zpad, const float* zborder, const float* s, const float* bias,
                                 float* y, void* y_hl, int hl_f16, int B, int H, int W, int Cout, int act, ocv_stream_t stream);


/* Bilinear resize of x [B,h,w,C1] (NHWC fp32) to H x W with align_corners = True, concatenated along channels with
 * skip [B,H,W,C2] (nullable, then C2 = 0), written in the hl32 split layout for C1+C2 channels (out_hl,
 * ocv_split_act_elems(B,H,W,C1+C2) bf16 elements, 16-byte aligned; pad channels zeroed).  C1, C2 multiples of 4.
 * Replaces F.interpolate + torch.cat of UpSampleWithSkip.forward (modules/DenseFeatureExtractor.py:44-47) and feeds
 * ocv_conv_nhwc_split_fwd. */
int ocv_upsample_concat_split_fwd(const float* x, int h, int w, int C1, const float* skip, int C2, void* out_hl, int B,
                                  int H, int W, ocv_stream_t stream);
/* the same with the element type of out_hl as a parameter (0 = bf16 pairs, 1 = fp16 pairs: ocv_conv_nhwc_split_x_fwd) */
int ocv_upsample_concat_split_x_fwd(const float* x, int h, int w, int C1, const float* skip, int C2, void* out_hl, int f16, int B,
                                    int H, int W, ocv_stream_t stream);

/* Validation-step arithmetic in one pass (next row N2: modules/GraphBinsLM.py:154-212, metrics/MetricsPreprocess.py:14-45,
 * metrics/AbsRel.py:44-52, SqRel.py:45-52, RMSE.py:48-55, RMSELog.py:45-52, Log10.py:52-61, AccThresh.py:59-66).
 * pred [B,1,h,w] = model output; pred_mirror (nullable) = model output for the horizontally flipped image (NOT flipped
 * back): with it the flip-TTA average 0.5 (clamp(pred) + clamp(flip(pred_mirror))) is evaluated, without it clamp(pred).
 * That map is resized to the ground truth's H x W (bilinear, align_corners = True), nan -> min_depth, +-inf ->
 * max_depth; valid pixels: min_depth < gt <= max_depth inside the crop box [crop_y0, crop_y1) x [crop_x0, crop_x1)
 * (pass 0, H, 0, W for none).  records [B][10] per image: abs_rel, sq_rel, rmse, rmse_log, log10, delta1, delta2,
 * delta3 (means over the image's valid pixels; the two RMSEs square-rooted), n_valid, first_image_id + b.
 * Deterministic (fixed-order double-precision reduction); workspace from ocv_depth_metrics_workspace_bytes. */
size_t ocv_depth_metrics_workspace_bytes(int B, int H, int W);
int ocv_depth_metrics_fwd(const float* pred, const float* pred_mirror, int h, int w, const float* gt, int H, int W,
                          float min_depth, float max_depth, int crop_y0, int crop_y1, int crop_x0, int crop_x1,
                          long first_image_id, float* records, int B, void* workspace, size_t workspace_bytes,
                          ocv_stream_t stream);

/* Tail of mViT / ObjCAViT.forward + glue of AdaBins / GraphBins.forward in one launch (modules/miniViT.py:33-42, modules/AdaBins.py:79-83):
 *   y = raw [B][n_bins] (the regressor's last Linear) -> OCV_BINNORM_LINEAR: relu(y) + 0.1 | OCV_BINNORM_SIGMOID: sigmoid(y) |
 *   OCV_BINNORM_NONE: y (already normalised, e.g. a softmax);  widths_normed = y / sum_row(y) (NONE: y);
 *   edges [B][n_bins + 1] = cumsum([min_depth, (max_depth - min_depth) * widths_normed]);  centers [B][n_bins] = edge midpoints. */
#define OCV_BINNORM_LINEAR 0
#define OCV_BINNORM_SIGMOID 1
#define OCV_BINNORM_NONE 2
int ocv_bin_edges_fwd(const float* raw, int mode, float min_depth, float max_depth, float* widths_normed, float* edges,
                      float* centers, int B, int n_bins, ocv_stream_t stream);
/* The bin regressor and ocv_bin_edges_fwd in ONE launch (modules/miniViT.py:33-42 == modules/ObjCAViT.py:373-388): per image,
 * row x + b * x_stride (E floats: token 0) -> Linear(E, H1) + LeakyReLU(leaky_slope) -> Linear(H1, H2) + LeakyReLU -> Linear(H2, n_bins) ->
 * normalisation `mode` -> widths, edges, centres as above.  Weights row-major [out][in], 16-byte aligned; E, H1, H2 multiples of 4, <= 1024.
 * Plain fp32 FMA chains in k order (the three GEMM launches it replaces summed in MFMA order: results agree to fp32 rounding). */
int ocv_regressor_bins_fwd(const float* x, long x_stride, const float* w1, const float* b1, const float* w2, const float* b2,
                           const float* w3, const float* b3, int E, int H1, int H2, int n_bins, float leaky_slope, int mode,
                           float min_depth, float max_depth, float* widths_normed, float* edges, float* centers, int B,
                           ocv_stream_t stream);

/* Ragged object lists with the per-image COUNT in device memory (shape-static: a captured graph serves any object set up to its
 * capacity; replaces pad_sequence / F.pad / mask building of SelfAttnCrossAttn.forward, modules/ObjCAViT.py:180-183,192-194).
 *   ocv_object_tokens_pad_fwd: tokens [B][capacity][E] (rows >= counts[b] arbitrary) -> out = rows < count kept, the rest
 *     pad_value (1e-4); mask[b][j] = (j >= counts[b]).  out may alias tokens.
 *   ocv_object_front_pad_fwd: objects [B][capacity][E] -> out [B][S][E] = [ pad_value x (S - Nmax) | rows 0 .. Nmax-1 ] (rows
 *     padded at the FRONT), key_padding_mask [B][S] = (j >= counts[b]) (mask padded at the BACK).  Nmax = nmax
 *     when nmax > 0 (the global batch's longest list, data-parallel shards), else the largest count within the image's group
 *     of `group` consecutive images (one group = one call of the reference: its Nmax is that call's longest list, Q3).
 * counts[b] is read as min(max(counts[b], 1), capacity): the reference never has an image without a row (an image without
 * detections carries ONE <UNK> row, :311-315), and a count of 0 would mask every key (softmax over nothing = NaN). */
int ocv_object_tokens_pad_fwd(const float* tokens, const int* counts, float pad_value, float* out, uint8_t* mask, int B,
                              int capacity, int E, ocv_stream_t stream);
int ocv_object_front_pad_fwd(const float* objects, const int* counts, int group, int nmax, float pad_value, float* out,
                             uint8_t* key_padding_mask, int B, int capacity, int S, int E, ocv_stream_t stream);

/* Positional-embedding samplers of GridRandomPositionalEmbeddings.forward (modules/ObjCAViT.py:50-147) on the learnable
 * table `table` [>= gh*gw][E], viewed as a gh x gw grid of E-vectors (row y*gw + x; reference :82-83).  One output row
 * per coordinate row: out[r][0..E) = sample (+ addend[r][0..E) when addend is not NULL -- the object embedding it is
 * added to at :330).  coords [n_rows][coord_ld] fp32.
 *   OCV_POS_CENTRE_OBJ  coords = (x, y, ...) in full-resolution pixels; F.grid_sample(bilinear, zeros,
 *                       align_corners=False) at (x / p0 * 2 - 1, y / p1 * 2 - 1); the reference passes
 *                       p0 = image HEIGHT, p1 = image WIDTH (:104-105).                       (:102-110)
 *   OCV_POS_CENTRE_IMG  coords = (x, y, ...) of the image tokens, rows_per_image = S rows per image; row s = r % S:
 *                       s == 0 -> both components / p0 * 2 - 1 (p0 = gh), s == 1 -> both / p1 * 2 - 1 (p1 = gw),
 *                       s >= 2 -> raw coordinates (the reference indexes dim 1 of a B x S x 2 tensor; Q6).   (:93-100)
 *   OCV_POS_ROI         coords = (cx, cy, w, h): x1y1x2y2 = centre -+ half size clamped at 0 from below, then
 *                       torchvision.ops.ps_roi_align(output_size=[1,1], spatial_scale=p0, sampling_ratio=-1):
 *                       corners * p0 - 0.5, ceil(roi_h) x ceil(roi_w) bilinear samples averaged (samples outside
 *                       [-1, gh] x [-1, gw] count as 0; a box without extent gives 0/0 = NaN).            (:111-145)
 * The sample loops are bounded by the grid size whatever the box; no host synchronisation (hipGraph-capturable). */
#define OCV_POS_CENTRE_OBJ 0
#define OCV_POS_CENTRE_IMG 1
#define OCV_POS_ROI 2
int ocv_pos_grid_sample_fwd(const float* table, int gh, int gw, int E, const float* coords, int coord_ld, int n_rows,
                            int mode, float p0, float p1, int rows_per_image, const float* addend, float* out,
                            ocv_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* OBJCAVIT_HIP_H */
