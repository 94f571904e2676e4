// Shared device helpers for the gfx950 (CDNA4 / MI355X) kernels.
//
// All matrix work uses the exact-fp32 matrix instruction
// v_mfma_f32_32x32x2_f32 (64-lane wavefront, 32x32 output tile, K = 2 per
// issue, 64 FLOP/clk/SIMD).  Operand and result lane maps (guide section 3):
//   A operand: lane l supplies A[i = l & 31][k = l >> 5]        (one VGPR)
//   B operand: lane l supplies B[k = l >> 5][j = l & 31]        (one VGPR)
//   C/D      : 16 VGPRs; register r of lane l is
//              D[row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][col = l & 31]
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define OCV_WAVE 64

__device__ __forceinline__ f32x16 mfma_32x32x2(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// row of accumulator register r for lane-half hh (= lane >> 5)
__device__ __forceinline__ constexpr int acc_row(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float xor32_max(float v) { return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float xor32_sum(float v) { return v + __shfl_xor(v, 32, 64); }

// exp via the hardware exp2 (v_exp_f32): ~1 ulp, exp(-inf) == 0
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

// SiLU / sigmoid with the hardware reciprocal (v_rcp_f32, 1 ulp) instead of an IEEE division (a ~12-instruction
// sequence): the expand convolutions apply SiLU to every output and were VALU-bound on it.
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + fast_exp(-x)); }
__device__ __forceinline__ float fast_silu(float x) { return x * fast_sigmoid(x); }

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// Two-term split of an fp32 value into the 2-byte (hi, lo) bit patterns of the "hl32" activation layout (csrc/conv_igemm.hip):
// F16 = fp16 terms (hi = fp16(v), lo = fp16(v - hi): 22 bits down to an absolute floor of 2^-25, range +-65504 -- beyond it inf,
// loud), else bf16 terms (16 bits, fp32's range).  Which one a tensor holds is the producing call's ``f16`` flag; consumers are
// told the same flag.
template <bool F16>
__device__ __forceinline__ void ocv_split1(float v, unsigned short& hi, unsigned short& lo) {
  if constexpr (F16) {
    const _Float16 h = (_Float16)v;
    hi = __builtin_bit_cast(unsigned short, h);
    lo = __builtin_bit_cast(unsigned short, (_Float16)(v - (float)h));
  } else {
    const __bf16 h = (__bf16)v;
    hi = __builtin_bit_cast(unsigned short, h);
    lo = __builtin_bit_cast(unsigned short, (__bf16)(v - (float)h));
  }
}

// Range guard of the fp16 pairs (round 5).  The reference's convolutions are fp32 for any input (modules/DenseFeatureExtractor.py:
// 37-47,104-118); fp16 pairs end at +-65504.  Whether a model's activations fit is calibrated on its first batch -- and a LATER batch,
// or a replay of a captured graph, can exceed it.  So every kernel that writes fp16 pairs keeps the largest magnitude it converts
// (one v_max per element pair) and, on the rare true branch only, ORs 1 into ONE device word (the guard the host armed for this
// thread: ocv_range_flag_set; nullptr = not armed).  The word is sticky; the host reads it where it reads results and re-runs the
// batch on bf16 pairs (objcavit_amd/hip_ops/_core.py RangeGuard).  The limit is the calibration's own, 65504 / 16 (hip_ops.fp16_range_report):
// a batch that leaves the envelope its model was calibrated for is re-run, long before a value can turn into inf.  NaN inputs stay
// NaN (loud) without tripping -- bf16 pairs would not cure them.
constexpr float OCV_F16_GUARD = 4094.f;
__device__ __forceinline__ void ocv_range_note(unsigned* flag, float amax) {
  if (flag != nullptr && amax > OCV_F16_GUARD) atomicOr(flag, 1u);
}
__device__ __forceinline__ float ocv_amax4(float m, const f32x4 v) {
  return fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
}


// In-launch hand-off between workgroups (csrc/token_h2.hip: the layer tail's feed-forward chunks shared out over workgroups, the row
// block's last arriver finishes the layer).  Per-XCD L2s are not coherent and a CU's L1 is never refreshed by other CUs' stores, so a
// producer's partial goes out WRITE-THROUGH (sc1: an agent-scope atomic store), the storing wave drains it (s_waitcnt vmcnt(0)), and
// ONE lane draws a ticket with an agent-scope atomic add; the last arriver issues one agent-scope acquire and reads with ordinary loads.
typedef __attribute__((address_space(1))) unsigned ocv_gu32;
__device__ __forceinline__ void ocv_store_sc1(float* p, float v) {
  __hip_atomic_store((ocv_gu32*)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
void ocv_set_error(const char* fmt, ...);

// the calling thread's armed range-guard word (device memory), or nullptr (csrc/capi.hip)
unsigned* ocv_range_flag_current();

#define OCV_CHECK_ARG(cond, ...)        \
  do {                                  \
    if (!(cond)) {                      \
      ocv_set_error(__VA_ARGS__);       \
      return -1;                        \
    }                                   \
  } while (0)

#define OCV_CHECK_LAUNCH(name)                                              \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      ocv_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return (int)e__;                                                      \
    }                                                                       \
  } while (0)

// zero-fill of a device buffer (4-byte aligned, a multiple of 4 bytes) by a LAUNCH: a hipMemsetAsync issued inside a stream capture
// becomes a memset node, and a forward replayed with such a node came back with wrong values (csrc/capi.hip, round 4) while eager
// dispatch of the same calls was right.  0 / hipError_t.
int ocv_zero_async(void* p, size_t nbytes, hipStream_t stream);

// attention launch with an explicit row stride for the key-padding mask (mask[b * mask_ld + key])
int ocv_attention_launch(const float* q, long q_bs, int q_ss, const float* k, long k_bs, int k_ss, const float* v,
                         long v_bs, int v_ss, const uint8_t* key_padding_mask, int mask_ld, float* ctx, long o_bs,
                         int o_ss, int B, int H, int Sq, int Sk, float scale, hipStream_t stream);

// fused few-key multi-head attention (csrc/linear.hip); 0 = launched, 1 = shape not covered, other = launch error
int ocv_cross_attn_fused_launch(const float* q_src, const float* k_src, const float* v_src, const uint8_t* mask, int mask_ld,
                                const float* in_w, const float* in_b, const float* out_w, const float* out_b, float* out, int B,
                                int Sq, int Sk, int Se, int E, int H, hipStream_t st);

// the same on packed three-term-split weights, K / V projected once per image into kv_ws [B][2][32][128] (csrc/token_split3.hip)
int ocv_cross_attn_split3_launch(const float* q_src, const float* k_src, const float* v_src, const uint8_t* mask, int mask_ld,
                                 const void* in_p3, const float* in_b, const void* out_p3, const float* out_b, float* out,
                                 float* kv_ws, int B, int Sq, int Sk, int Se, int E, int H, hipStream_t st);

// FFN with a row tile's hidden units shared out over several workgroups + finish pass (csrc/linear.hip)
int ocv_ffn_split_count(int M, int FF);
int ocv_ffn_split_launch(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                         const float* gamma, const float* beta, float eps, const uint8_t* zero_row_mask, float* out, int M,
                         int FF, float* part, int nsplit, hipStream_t st);

// A caller's ocv_encoder_layer_params (element ``index`` of an array whose stride is the caller's struct_size) copied into
// the library's own struct: fields beyond the caller's struct_size read as NULL.  false = struct_size too small to hold
// the twelve fp32 parameter pointers.
bool ocv_layer_params_view(const void* caller_params, int index, void* lib_params);

static inline bool ocv_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline int ocv_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
