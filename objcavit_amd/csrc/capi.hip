// Composite entry points of the C ABI (multi-head attention module and one
// transformer encoder layer) plus error reporting.  Each composite only chains
// the kernels of linear.hip / attention.hip on the caller's stream using the
// caller's workspace: no allocation, no synchronisation.
#include <stdarg.h>
#include <stdio.h>
#include <math.h>

#include <stdlib.h>
#include <string.h>
#include <stddef.h>

#include "common.hpp"
#include "../../include/objcavit_hip.h"

namespace {
thread_local char g_err[512] = "";
inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }
}  // namespace

void ocv_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

bool ocv_layer_params_view(const void* caller_params, int index, void* lib_params) {
  ocv_encoder_layer_params* out = (ocv_encoder_layer_params*)lib_params;
  memset(out, 0, sizeof(*out));
  const size_t sz = ((const ocv_encoder_layer_params*)caller_params)->struct_size;
  if (sz < offsetof(ocv_encoder_layer_params, in_proj_p3) || sz > 4096 || (sz & (sizeof(void*) - 1)) != 0) return false;
  const char* src = (const char*)caller_params + (size_t)index * sz;
  memcpy(out, src, sz < sizeof(*out) ? sz : sizeof(*out));
  out->struct_size = sizeof(*out);
  return true;
}

// ---------------------------------------------------------------------------
// range guard of the fp16 pairs (common.hpp: ocv_range_note)
// ---------------------------------------------------------------------------
namespace {
thread_local unsigned* g_range_flag = nullptr;

__global__ void range_flag_take_kernel(unsigned* __restrict__ flag, unsigned* __restrict__ out) {
  if (threadIdx.x == 0) {
    *out = *flag;
    *flag = 0u;
  }
}
}  // namespace

unsigned* ocv_range_flag_current() { return g_range_flag; }

extern "C" int ocv_range_flag_set(unsigned* flag) {
  OCV_CHECK_ARG((reinterpret_cast<uintptr_t>(flag) & 3) == 0, "ocv_range_flag_set: the word must be 4-byte aligned");
  g_range_flag = flag;
  return 0;
}

extern "C" int ocv_range_flag_take_fwd(unsigned* flag, unsigned* out, ocv_stream_t stream) {
  OCV_CHECK_ARG(flag && out && flag != out, "ocv_range_flag_take_fwd: null pointer (or out == flag)");
  OCV_CHECK_ARG(((reinterpret_cast<uintptr_t>(flag) | reinterpret_cast<uintptr_t>(out)) & 3) == 0, "ocv_range_flag_take_fwd: 4-byte alignment");
  hipLaunchKernelGGL(range_flag_take_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, flag, out);
  OCV_CHECK_LAUNCH("ocv_range_flag_take_fwd");
  return 0;
}

extern "C" int ocv_abi_version(void) { return OCV_ABI_VERSION; }
extern "C" const char* ocv_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------
// nn.MultiheadAttention forward (distinct q / k / v sources allowed)
// workspace: qp [B*Sq, E] | kp [B*Sk, E] | vp [B*Sk, E] | ctx [B*Sq, E]
// ---------------------------------------------------------------------------
extern "C" size_t ocv_mha_workspace_bytes(int B, int Sq, int Sk, int E) {
  if (B < 1 || Sq < 1 || Sk < 1 || E < 1) return 0;
  const size_t q = align_up((size_t)B * Sq * E * sizeof(float)), k = align_up((size_t)B * Sk * E * sizeof(float));
  const size_t kv = align_up((size_t)B * 2 * 32 * E * sizeof(float));      // ocv_mha_split3_fwd: the per-image K / V record
  return 2 * q + (2 * k > kv ? 2 * k : kv);
}

extern "C" int ocv_mha_fwd(const float* q_src, const float* k_src, const float* v_src, const uint8_t* key_padding_mask,
                           const float* in_proj_w, const float* in_proj_b, const float* out_w, const float* out_b,
                           float* out, int B, int Sq, int Sk, int kv_limit, int E, int H, void* workspace,
                           size_t workspace_bytes, ocv_stream_t stream) {
  OCV_CHECK_ARG(q_src && k_src && v_src && in_proj_w && in_proj_b && out_w && out_b && out && workspace,
                "ocv_mha_fwd: null pointer");
  OCV_CHECK_ARG(H >= 1 && E == H * 32, "ocv_mha_fwd: head dim must be 32 (E=%d, H=%d)", E, H);
  OCV_CHECK_ARG(workspace_bytes >= ocv_mha_workspace_bytes(B, Sq, Sk, E), "ocv_mha_fwd: workspace too small");
  const size_t qb = align_up((size_t)B * Sq * E * sizeof(float)), kb = align_up((size_t)B * Sk * E * sizeof(float));
  char* ws = (char*)workspace;
  float* qp = (float*)ws;
  float* kp = (float*)(ws + qb);
  float* vp = (float*)(ws + qb + kb);
  float* ctx = (float*)(ws + qb + 2 * kb);
  int rc;
  OCV_CHECK_ARG(kv_limit >= 0, "ocv_mha_fwd: negative kv_limit");
  OCV_CHECK_ARG(kv_limit == 0 || key_padding_mask != nullptr, "ocv_mha_fwd: kv_limit needs a key_padding_mask");
  // keys >= kv_limit are all masked (caller's promise): project and score only the first Se keys of every batch row
  const int Se = (kv_limit > 0 && kv_limit < Sk) ? kv_limit : Sk;
  {
    // at most 32 live keys (the image <- object cross-attention): everything in one launch
    rc = ocv_cross_attn_fused_launch(q_src, k_src, v_src, key_padding_mask, Sk, in_proj_w, in_proj_b, out_w, out_b, out, B,
                                     Sq, Sk, Se, E, H, (hipStream_t)stream);
    if (rc != 1) return rc;
  }
  if ((rc = ocv_linear_fwd(q_src, E, 0, in_proj_w, E, 0, 0, in_proj_b, qp, E, 0, 1, B * Sq, E, E, OCV_ACT_NONE, stream))) return rc;
  if ((rc = ocv_linear_fwd(k_src, E, (long)Sk * E, in_proj_w + (size_t)E * E, E, 0, 0, in_proj_b + E, kp, E, (long)Se * E, B, Se, E, E, OCV_ACT_NONE, stream))) return rc;
  if ((rc = ocv_linear_fwd(v_src, E, (long)Sk * E, in_proj_w + (size_t)2 * E * E, E, 0, 0, in_proj_b + 2 * E, vp, E, (long)Se * E, B, Se, E, E, OCV_ACT_NONE, stream))) return rc;
  if ((rc = ocv_attention_launch(qp, (long)Sq * E, E, kp, (long)Se * E, E, vp, (long)Se * E, E, key_padding_mask, Sk, ctx,
                                 (long)Sq * E, E, B, H, Sq, Se, 1.0f / sqrtf(32.0f), (hipStream_t)stream))) return rc;
  return ocv_linear_fwd(ctx, E, 0, out_w, E, 0, 0, out_b, out, E, 0, 1, B * Sq, E, E, OCV_ACT_NONE, stream);
}

// nn.MultiheadAttention forward on packed three-term-split projection weights (ocv_pack_split3_fwd of in_proj_weight
// [3E, E] and out_proj.weight [E, E]); QK^T / PV stay on exact fp32 MFMA.  <= 32 live keys: K / V projected once per image
// + one fused launch per 32-query tile; otherwise the split3 linears around the attention kernel.
extern "C" int ocv_mha_split3_fwd(const float* q_src, const float* k_src, const float* v_src, const uint8_t* key_padding_mask,
                                  const void* in_proj_p3, const float* in_proj_b, const void* out_proj_p3, const float* out_b,
                                  float* out, int B, int Sq, int Sk, int kv_limit, int E, int H, void* workspace,
                                  size_t workspace_bytes, ocv_stream_t stream) {
  OCV_CHECK_ARG(q_src && k_src && v_src && in_proj_p3 && in_proj_b && out_proj_p3 && out_b && out && workspace,
                "ocv_mha_split3_fwd: null pointer");
  OCV_CHECK_ARG(H == 4 && E == 128, "ocv_mha_split3_fwd: built for E = 128, H = 4 (got %d, %d)", E, H);
  OCV_CHECK_ARG(B >= 1 && Sq >= 1 && Sk >= 1, "ocv_mha_split3_fwd: bad sizes");
  OCV_CHECK_ARG(workspace_bytes >= ocv_mha_workspace_bytes(B, Sq, Sk, E), "ocv_mha_split3_fwd: workspace too small");
  OCV_CHECK_ARG(kv_limit >= 0, "ocv_mha_split3_fwd: negative kv_limit");
  OCV_CHECK_ARG(kv_limit == 0 || key_padding_mask != nullptr, "ocv_mha_split3_fwd: kv_limit needs a key_padding_mask");
  const int Se = (kv_limit > 0 && kv_limit < Sk) ? kv_limit : Sk;
  const size_t qb = align_up((size_t)B * Sq * E * sizeof(float)), kb = align_up((size_t)B * Sk * E * sizeof(float));
  char* ws = (char*)workspace;
  int rc;
  {
    rc = ocv_cross_attn_split3_launch(q_src, k_src, v_src, key_padding_mask, Sk, in_proj_p3, in_proj_b, out_proj_p3, out_b, out,
                                      (float*)(ws + 2 * qb), B, Sq, Sk, Se, E, H, (hipStream_t)stream);
    if (rc != 1) return rc;
  }
  float* qp = (float*)ws;
  float* ctx = (float*)(ws + qb);
  float* kp = (float*)(ws + 2 * qb);
  float* vp = (float*)(ws + 2 * qb + kb);
  const size_t tile = (size_t)4 * (E / 16) * 3 * 512;                       // packed elements of 128 weight rows
  const __bf16* wp = (const __bf16*)in_proj_p3;
  // all Sk key rows are projected (contiguous rows; keys >= Se are simply not scored)
  if ((rc = ocv_linear_split3_fwd(q_src, E, wp, in_proj_b, qp, E, B * Sq, E, E, OCV_ACT_NONE, stream))) return rc;
  if ((rc = ocv_linear_split3_fwd(k_src, E, wp + tile, in_proj_b + E, kp, E, B * Sk, E, E, OCV_ACT_NONE, stream))) return rc;
  if ((rc = ocv_linear_split3_fwd(v_src, E, wp + 2 * tile, in_proj_b + 2 * E, vp, E, B * Sk, E, E, OCV_ACT_NONE, stream))) return rc;
  if ((rc = ocv_attention_launch(qp, (long)Sq * E, E, kp, (long)Sk * E, E, vp, (long)Sk * E, E, key_padding_mask, Sk, ctx,
                                 (long)Sq * E, E, B, H, Sq, Se, 1.0f / sqrtf(32.0f), (hipStream_t)stream))) return rc;
  return ocv_linear_split3_fwd(ctx, E, out_proj_p3, out_b, out, E, B * Sq, E, E, OCV_ACT_NONE, stream);
}

// ---------------------------------------------------------------------------
// one post-norm transformer encoder layer
// workspace: qkv [M, 3E] | ctx [M, E] | x1 [M, E] | hid [M, FF]
// ---------------------------------------------------------------------------
extern "C" size_t ocv_encoder_layer_workspace_bytes(int B, int S, int E, int FF) {
  if (B < 1 || S < 1 || E < 1 || FF < 1) return 0;
  const size_t M = (size_t)B * S;
  return align_up(M * 3 * E * sizeof(float)) + 2 * align_up(M * E * sizeof(float)) + align_up(M * FF * sizeof(float));
}

extern "C" int ocv_encoder_layer_fwd(const float* x, const ocv_encoder_layer_params* p_caller,
                                     const uint8_t* key_padding_mask, int zero_padded_rows, float* out, int B, int S,
                                     int E, int H, int FF, float eps, void* workspace, size_t workspace_bytes,
                                     ocv_stream_t stream) {
  OCV_CHECK_ARG(x && p_caller && out && workspace, "ocv_encoder_layer_fwd: null pointer");
  ocv_encoder_layer_params pv;
  OCV_CHECK_ARG(ocv_layer_params_view(p_caller, 0, &pv), "ocv_encoder_layer_fwd: params->struct_size (%zu) is not a valid ocv_encoder_layer_params size", p_caller->struct_size);
  const ocv_encoder_layer_params* p = &pv;
  OCV_CHECK_ARG(E == 128 && H == 4, "ocv_encoder_layer_fwd: built for E = 128, H = 4 (got %d, %d)", E, H);
  OCV_CHECK_ARG(workspace_bytes >= ocv_encoder_layer_workspace_bytes(B, S, E, FF), "ocv_encoder_layer_fwd: workspace too small");
  const int M = B * S;
  char* ws = (char*)workspace;
  float* qkv = (float*)ws;
  ws += align_up((size_t)M * 3 * E * sizeof(float));
  float* ctx = (float*)ws;
  ws += align_up((size_t)M * E * sizeof(float));
  float* x1 = (float*)ws;
  ws += align_up((size_t)M * E * sizeof(float));
  float* hid = (float*)ws;
  int rc;
  const uint8_t* zmask = (zero_padded_rows && key_padding_mask) ? key_padding_mask : nullptr;
  const bool any3 = p->in_proj_p3 || p->out_proj_p3 || p->linear1_p3 || p->linear2_p3;
  const bool all3 = p->in_proj_p3 && p->out_proj_p3 && p->linear1_p3 && p->linear2_p3;
  OCV_CHECK_ARG(all3 || !any3, "ocv_encoder_layer_fwd: give all four packed split3 weights or none");
  const bool s3 = all3 && FF % 128 == 0;
  // packed QKV projection
  if (s3) rc = ocv_linear_split3_fwd(x, E, p->in_proj_p3, p->in_proj_b, qkv, 3 * E, M, 3 * E, E, OCV_ACT_NONE, stream);
  else rc = ocv_linear_fwd(x, E, 0, p->in_proj_w, E, 0, 0, p->in_proj_b, qkv, 3 * E, 0, 1, M, 3 * E, E, OCV_ACT_NONE, stream);
  if (rc) return rc;
  if ((rc = ocv_attention_fwd(qkv, (long)S * 3 * E, 3 * E, qkv + E, (long)S * 3 * E, 3 * E, qkv + 2 * E,
                              (long)S * 3 * E, 3 * E, key_padding_mask, ctx, (long)S * E, E, B, H, S, S,
                              1.0f / sqrtf(32.0f), stream))) return rc;
  // x1 = LN1(x + ctx Wo^T + bo)
  if (s3) {
    if ((rc = ocv_linear_residual_layernorm_split3_fwd(ctx, E, p->out_proj_p3, p->out_proj_b, x, E, p->norm1_w, p->norm1_b, eps,
                                                       nullptr, x1, E, M, E, E, stream))) return rc;
    // out = LN2(x1 + W2 relu(W1 x1 + b1) + b2); split partials live in the (otherwise unused) hid region
    return ocv_ffn_residual_layernorm_split3_fwd(x1, p->linear1_p3, p->linear1_b, p->linear2_p3, p->linear2_b, p->norm2_w,
                                                 p->norm2_b, eps, zmask, out, M, E, FF, hid, (size_t)M * FF * sizeof(float), stream);
  }
  if ((rc = ocv_linear_residual_layernorm_fwd(ctx, E, p->out_proj_w, E, p->out_proj_b, x, E, p->norm1_w, p->norm1_b,
                                              eps, nullptr, x1, E, M, E, E, stream))) return rc;
  if (FF % 128 == 0) {
    // out = LN2(x1 + W2 relu(W1 x1 + b1) + b2), hidden activations stay in LDS
    const int ns = ocv_ffn_split_count(M, FF);       // partials live in the (otherwise unused) hid region: ns * 128 <= FF
    if (ns > 1)
      return ocv_ffn_split_launch(x1, p->linear1_w, p->linear1_b, p->linear2_w, p->linear2_b, p->norm2_w, p->norm2_b, eps,
                                  zmask, out, M, FF, hid, ns, (hipStream_t)stream);
    return ocv_ffn_residual_layernorm_fwd(x1, p->linear1_w, p->linear1_b, p->linear2_w, p->linear2_b, p->norm2_w,
                                          p->norm2_b, eps, zmask, out, M, E, FF, stream);
  }
  // generic hidden size: hid = relu(x1 W1^T + b1); out = LN2(x1 + hid W2^T + b2)
  if ((rc = ocv_linear_fwd(x1, E, 0, p->linear1_w, E, 0, 0, p->linear1_b, hid, FF, 0, 1, M, FF, E, OCV_ACT_RELU, stream))) return rc;
  return ocv_linear_residual_layernorm_fwd(hid, FF, p->linear2_w, FF, p->linear2_b, x1, E, p->norm2_w, p->norm2_b, eps,
                                           zmask, out, E, M, E, FF, stream);
}


// ---------------------------------------------------------------------------
// a whole nn.TransformerEncoder (L post-norm layers) in 1 + 2 L launches: packed projection of layer 0, then per layer
// the attention kernel and ocv_layer_tail_split3_fwd / ocv_layer_tail_h2_fwd (everything token-local + the next layer's projection).
// workspace: qkv [M, 3E] | ctx [M, E] | xa [M, E] | xb [M, E]
// ---------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void zero_u32_kernel(unsigned* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0u;
}
}  // namespace

int ocv_zero_async(void* p, size_t nbytes, hipStream_t stream) {
  if (nbytes == 0) return 0;
  if (p == nullptr || (reinterpret_cast<uintptr_t>(p) & 3) != 0 || (nbytes & 3) != 0) {
    ocv_set_error("ocv_zero_async: needs a 4-byte aligned buffer of a multiple of 4 bytes");
    return -1;
  }
  const size_t n = nbytes / 4;
  size_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(zero_u32_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (unsigned*)p, n);
  OCV_CHECK_LAUNCH("ocv_zero_async");
  return 0;
}

extern "C" size_t ocv_encoder_stack_workspace_bytes(int B, int S, int E) {
  if (B < 1 || S < 1 || E < 1) return 0;
  const size_t M = (size_t)B * S;
  // + the feed-forward partial sums and arrival tickets of the tails' few-token form (ocv_layer_tail_h2_ws_fwd; 0 for large M)
  return align_up(M * 3 * E * sizeof(float)) + 3 * align_up(M * E * sizeof(float)) +
         align_up(B <= 4 ? ocv_layer_tail_h2_workspace_bytes((int)M, 1024) : 0);
}

extern "C" int ocv_encoder_stack_fwd(const float* x, const ocv_encoder_layer_params* layers_caller, int n_layers,
                                     const uint8_t* key_padding_mask, int zero_padded_rows, float* out, int B, int S, int E,
                                     int H, int FF, float eps, void* workspace, size_t workspace_bytes, ocv_stream_t stream) {
  OCV_CHECK_ARG(x && layers_caller && out && workspace && n_layers >= 1 && n_layers <= 64, "ocv_encoder_stack_fwd: null pointer / layer count not in 1..64");
  ocv_encoder_layer_params layers[64];
  for (int l = 0; l < n_layers; ++l)
    OCV_CHECK_ARG(ocv_layer_params_view(layers_caller, l, &layers[l]), "ocv_encoder_stack_fwd: layers[0].struct_size (%zu) is not a valid ocv_encoder_layer_params size", layers_caller->struct_size);
  OCV_CHECK_ARG(E == 128 && H == 4 && FF >= 128 && FF % 128 == 0, "ocv_encoder_stack_fwd: built for E = 128, H = 4, FF a multiple of 128 (got %d, %d, %d)", E, H, FF);
  OCV_CHECK_ARG(workspace_bytes >= ocv_encoder_stack_workspace_bytes(B, S, E), "ocv_encoder_stack_fwd: workspace too small");
  for (int l = 0; l < n_layers; ++l)
    OCV_CHECK_ARG(layers[l].in_proj_p3 && layers[l].out_proj_p3 && layers[l].linear1_p3 && layers[l].linear2_p3,
                  "ocv_encoder_stack_fwd: layer %d lacks its packed split3 weights (ocv_pack_split3_fwd)", l);
  // every layer carries the two-term fp16 weights too: the token-local tails run on them (csrc/token_h2.hip); layer 0's
  // projection stays on its three-term weights (one launch)
  bool tails_h2 = true;
  for (int l = 0; l < n_layers; ++l)
    tails_h2 = tails_h2 && layers[l].in_proj_h2 && layers[l].out_proj_h2 && layers[l].linear1_h2 && layers[l].linear2_h2;
  const int M = B * S;
  char* ws = (char*)workspace;
  float* qkv = (float*)ws;
  ws += align_up((size_t)M * 3 * E * sizeof(float));
  float* ctx = (float*)ws;
  ws += align_up((size_t)M * E * sizeof(float));
  float* xa = (float*)ws;
  ws += align_up((size_t)M * E * sizeof(float));
  float* xb = (float*)ws;
  ws += align_up((size_t)M * E * sizeof(float));
  // few tokens: the tails share their feed-forward chunks out over workgroups; their arrival tickets are cleared ONCE per call
  // (every tail leaves them zero); FF other than the 1024 the workspace was sized for: one workgroup per row block
  void* tail_ws = nullptr;
  // (only for batches of up to 4 images: at bs 16 the OBJECT tokens' stack is few tokens too -- 16 row blocks -- but the chip has the
  //  other launches of the forward / the other batches in flight to run beside it, and the groups' repeated output projection is
  //  then pure extra work: 1073 -> 1067 img/s with three batches in flight, 997 -> 995 one at a time)
  const size_t tail_bytes = (FF == 1024 && B <= 4) ? ocv_layer_tail_h2_workspace_bytes(M, FF) : 0;
  if (tails_h2 && tail_bytes != 0) {
    tail_ws = ws;
    const size_t nblk = (size_t)((M + 31) / 32), G = (size_t)ocv_layer_tail_h2_groups(M, FF);
    // (ocv_zero_async: a launch, not a memset node -- see common.hpp)
    const int zrc = ocv_zero_async((char*)tail_ws + nblk * G * 32 * 128 * sizeof(float), nblk * sizeof(unsigned), (hipStream_t)stream);
    if (zrc != 0) return zrc;
  }
  int rc;
  if ((rc = ocv_linear_split3_fwd(x, E, layers[0].in_proj_p3, layers[0].in_proj_b, qkv, 3 * E, M, 3 * E, E, OCV_ACT_NONE, stream))) return rc;
  const float* cur = x;
  for (int l = 0; l < n_layers; ++l) {
    const bool last = l + 1 == n_layers;
    if ((rc = ocv_attention_fwd(qkv, (long)S * 3 * E, 3 * E, qkv + E, (long)S * 3 * E, 3 * E, qkv + 2 * E, (long)S * 3 * E, 3 * E,
                                key_padding_mask, ctx, (long)S * E, E, B, H, S, S, 1.0f / sqrtf(32.0f), stream))) return rc;
    float* dst = last ? out : ((l & 1) ? xb : xa);
    const uint8_t* zmask = (last && zero_padded_rows && key_padding_mask) ? key_padding_mask : nullptr;
    // the tail of layer l writes the next layer's q | k | v into the buffer this layer's attention has just consumed:
    // launches on one stream run in order, so the attention above has finished reading it
    if (tails_h2)
      rc = ocv_layer_tail_h2_ws_fwd(ctx, cur, &layers[l], last ? nullptr : layers[l + 1].in_proj_h2, last ? nullptr : layers[l + 1].in_proj_b,
                                    eps, zmask, dst, last ? nullptr : qkv, M, E, FF, tail_ws, tail_bytes, stream);
    else
      rc = ocv_layer_tail_split3_fwd(ctx, cur, &layers[l], last ? nullptr : layers[l + 1].in_proj_p3,
                                     last ? nullptr : layers[l + 1].in_proj_b, eps, zmask, dst, last ? nullptr : qkv, M, E, FF, stream);
    if (rc) return rc;
    cur = dst;
  }
  return 0;
}
