// Multi-head attention core for gfx950: softmax(q k^T * scale + key mask) v,
// head dim 32, fp32, v_mfma_f32_32x32x2_f32 for both contractions.
//
// Work split: grid = (ceil(Sq / 128), H, B); a workgroup = 4 wavefronts, each
// wavefront owns one 32-query tile of one (batch, head).  The K and V rows of
// the head (Sk x 32 each) are staged once per workgroup in LDS in chunks of up
// to KC keys (S = 300 / 418 fit in one chunk), and every wavefront walks the
// chunk in 32-key tiles with an online softmax:
//
//   S^T tile = K_tile (32 keys x 32) . Q^T (32 x 32 queries)      16 MFMAs
//     -> lane (query = lane & 31) holds 16 of the tile's 32 keys; the other 16
//        sit in lane ^ 32, so the row max / row sum are 15 in-register ops and
//        ONE cross-lane exchange (wavefront shuffle xor 32).
//   O^T     += V_tile^T (32 x 32 keys) . P^T (32 keys x 32 queries)  16 MFMAs
//     -> the probability registers are consumed DIRECTLY as the B operand of
//        the second contraction (accumulator register r of lane-half hh is key
//        row (r&3)+8(r>>2)+4hh = exactly the two k-slots of one 32x32x2 issue),
//        so P never goes through LDS.
//
// LDS images: K rows padded to 33 floats (the A operand reads 32 consecutive
// keys at one column: 33 makes that conflict-free), V rows dense (the A operand
// of P.V reads one key row across 32 consecutive columns).
#include "common.hpp"
#include "../../include/objcavit_hip.h"

namespace {

constexpr int HD = 32;          // head dim
constexpr int KC_MAX = 512;     // keys per LDS chunk
constexpr int KLD = HD + 1;

struct AttnArgs {
  const float *q, *k, *v;
  const uint8_t* mask;
  float* ctx;
  long q_bs, k_bs, v_bs, o_bs;
  int q_ss, k_ss, v_ss, o_ss;
  int Sq, Sk, kc;   // kc = chunk capacity (multiple of 32)
  int mask_ld;
  float scale;
};

template <bool VEC>
__global__ __launch_bounds__(256) void attention_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Vs = lds;                      // [kc][32]
  float* Ks = Vs + p.kc * HD;           // [kc][33]
  float* Ms = Ks + p.kc * KLD;          // [kc] additive mask (0 / -inf)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int qbase = (blockIdx.x * 4 + wave) * 32;
  const float NEG_INF = -__builtin_inff();

  const float* qp = p.q + (long)b * p.q_bs + h * HD;
  const float* kp = p.k + (long)b * p.k_bs + h * HD;
  const float* vp = p.v + (long)b * p.v_bs + h * HD;

  // Q^T fragment (B operand): lane supplies Q[qbase + l31][2s + hh] * scale
  float qf[16];
  {
    const int qi = qbase + l31;
    const bool ok = qi < p.Sq;
    const float* src = qp + (long)(ok ? qi : 0) * p.q_ss + hh;
#pragma unroll
    for (int s = 0; s < 16; ++s) qf[s] = ok ? src[2 * s] * p.scale : 0.f;
  }

  float m_run = NEG_INF, l_half = 0.f;
  f32x16 o = {0};

  for (int c0 = 0; c0 < p.Sk; c0 += p.kc) {
    const int nk = min(p.kc, ((p.Sk - c0 + 31) / 32) * 32);   // keys in this chunk, padded to a tile
    __syncthreads();
    // ---- stage K / V / mask chunk: thread -> (row, 4 consecutive d)
    for (int r = tid >> 3; r < nk; r += 32) {
      const int d4 = (tid & 7) * 4, key = c0 + r;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (key < p.Sk) {
        const float* ks = kp + (long)key * p.k_ss + d4;
        const float* vs = vp + (long)key * p.v_ss + d4;
        if (VEC) {
          kv = ld4(ks);
          vv = ld4(vs);
        } else {
          kv = make_float4(ks[0], ks[1], ks[2], ks[3]);
          vv = make_float4(vs[0], vs[1], vs[2], vs[3]);
        }
      }
      float* kd = Ks + r * KLD + d4;
      kd[0] = kv.x; kd[1] = kv.y; kd[2] = kv.z; kd[3] = kv.w;
      *reinterpret_cast<float4*>(Vs + r * HD + d4) = vv;
    }
    for (int r = tid; r < nk; r += 256) {
      const int key = c0 + r;
      const bool dead = key >= p.Sk || (p.mask != nullptr && p.mask[(long)b * p.mask_ld + key] != 0);
      Ms[r] = dead ? NEG_INF : 0.f;
    }
    __syncthreads();

    for (int kt = 0; kt < nk; kt += 32) {
      // ---- scores^T tile: rows = keys, cols = queries
      f32x16 s = {0};
      const float* krow = Ks + (kt + l31) * KLD + hh;
#pragma unroll
      for (int st = 0; st < 16; ++st) s = mfma_32x32x2(krow[2 * st], qf[st], s);

      float tmax = NEG_INF;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] += Ms[kt + acc_row(r, hh)];
        tmax = fmaxf(tmax, s[r]);
      }
      tmax = xor32_max(tmax);
      const float m_new = fmaxf(m_run, tmax);
      const bool none = m_new == NEG_INF;                 // every key so far masked for this query
      const float alpha = none ? 1.f : fast_exp(m_run - m_new);
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pr = none ? 0.f : fast_exp(s[r] - m_new);
        s[r] = pr;
        psum += pr;
      }
      l_half = l_half * alpha + psum;
      m_run = m_new;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] *= alpha;

      // ---- O^T += V^T . P^T ; k-slot hh of MFMA r is key kt + acc_row(r, hh)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float a = Vs[(kt + acc_row(r, hh)) * HD + l31];
        o = mfma_32x32x2(a, s[r], o);
      }
    }
  }

  const float l = xor32_sum(l_half);
  const float inv = 1.0f / l;           // l == 0 (all keys masked) -> inf * 0 = NaN, as torch
  const int qi = qbase + l31;
  if (qi < p.Sq) {
    float* dst = p.ctx + (long)b * p.o_bs + (long)qi * p.o_ss + h * HD;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      // registers 4g..4g+3 are d = 8g + 4hh + 0..3 for this query
      float4 t = make_float4(o[4 * g + 0] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv);
      float* d = dst + 8 * g + 4 * hh;
      if (VEC) {
        *reinterpret_cast<float4*>(d) = t;
      } else {
        d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
      }
    }
  }
}

}  // namespace

extern "C" int ocv_attention_fwd(const float* q, long q_bs, int q_ss, const float* k, long k_bs, int k_ss,
                                 const float* v, long v_bs, int v_ss, const uint8_t* key_padding_mask, float* ctx,
                                 long o_bs, int o_ss, int B, int H, int Sq, int Sk, float scale, ocv_stream_t stream) {
  return ocv_attention_launch(q, q_bs, q_ss, k, k_bs, k_ss, v, v_bs, v_ss, key_padding_mask, Sk, ctx, o_bs, o_ss, B, H,
                              Sq, Sk, scale, (hipStream_t)stream);
}

int ocv_attention_launch(const float* q, long q_bs, int q_ss, const float* k, long k_bs, int k_ss, const float* v,
                         long v_bs, int v_ss, const uint8_t* key_padding_mask, int mask_ld, float* ctx, long o_bs,
                         int o_ss, int B, int H, int Sq, int Sk, float scale, hipStream_t stream) {
  OCV_CHECK_ARG(q && k && v && ctx, "ocv_attention_fwd: null pointer");
  OCV_CHECK_ARG(B >= 1 && H >= 1 && Sq >= 1 && Sk >= 1, "ocv_attention_fwd: bad sizes B=%d H=%d Sq=%d Sk=%d", B, H, Sq, Sk);
  OCV_CHECK_ARG(B <= 65535 && H <= 65535, "ocv_attention_fwd: grid too large");
  AttnArgs a{q, k, v, key_padding_mask, ctx, q_bs, k_bs, v_bs, o_bs, q_ss, k_ss, v_ss, o_ss, Sq, Sk, 0, mask_ld, scale};
  a.kc = ((Sk + 31) / 32) * 32;
  if (a.kc > KC_MAX) a.kc = KC_MAX;
  const size_t lds = (size_t)a.kc * (HD + KLD + 1) * sizeof(float);
  const bool vec = ocv_aligned16(q) && ocv_aligned16(k) && ocv_aligned16(v) && ocv_aligned16(ctx) &&
                   (k_bs % 4 == 0) && (v_bs % 4 == 0) && (o_bs % 4 == 0) && (k_ss % 4 == 0) && (v_ss % 4 == 0) &&
                   (o_ss % 4 == 0);
  dim3 grid(ocv_cdiv(Sq, 128), H, B), block(256);
  hipStream_t st = stream;
  if (vec) {
    static bool attr_v = false;
    if (!attr_v) {
      (void)hipFuncSetAttribute((const void*)attention_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_v = true;
    }
    hipLaunchKernelGGL((attention_kernel<true>), grid, block, lds, st, a);
  } else {
    static bool attr_s = false;
    if (!attr_s) {
      (void)hipFuncSetAttribute((const void*)attention_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_s = true;
    }
    hipLaunchKernelGGL((attention_kernel<false>), grid, block, lds, st, a);
  }
  OCV_CHECK_LAUNCH("ocv_attention_fwd");
  return 0;
}
