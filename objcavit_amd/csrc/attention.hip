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
#include <stdlib.h>
#include <string.h>

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

// ---------------------------------------------------------------------------
// The same attention with BOTH contractions as two-term fp16 splits (round 3; the arithmetic of csrc/xattn_h2.hip:
// v = hi + 2^-11 lo', hi = fp16(v), lo' = fp16((v - hi) 2^11), a product block = three v_mfma_f32_32x32x16_f16 into an
// accumulator pair, 22-bit products = the error of an fp32 FMA chain).  A 32-key tile costs 6 + 6 MFMAs of 32 cycles
// instead of 16 + 16 exact-fp32 ones of 64: 384 matrix-pipe cycles against 2048 -- the exact kernel is bound by them at
// S = 1200 (do_final_upscale models: 124 us per launch, 8 launches per forward).
// K and V are split ONCE per workgroup while they are staged (4 query tiles share a chunk) and parked in LDS in operand
// order, so the key loop reads them with one conflict-free ds_read_b128 per fragment:
//   K: lane (key, half) of step t holds K[key][16 t + 8 half + e]       (the queries' B operand uses the same d order)
//   V: lane (d, half) of step t holds V[kt + acc_row(8 t + e, half)][d] (= the register order of the probabilities,
//      which are packed from their accumulator straight into the B operand, as before)
// The scores are formed in the log2 domain (Q scaled by log2(e) / sqrt(d)): the softmax is a bare v_exp_f32; the running
// output pair is rescaled only in the tiles where some query's running maximum moved (wavefront-uniform test).
// fp16's range applies to q / sqrt(d), k and v (beyond +-65504 the affected rows turn inf / NaN); ocv_attention_set_dispatch(1)
// selects the exact kernel above.
// ---------------------------------------------------------------------------
typedef _Float16 at_h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 at_h16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void at_split(float x, _Float16& hi, _Float16& lo) {
  hi = (_Float16)x;
  lo = (_Float16)((x - (float)hi) * 2048.0f);
}

__device__ __forceinline__ void at_mfma3(const at_h16x8 ah, const at_h16x8 al, const at_h16x8 bh, const at_h16x8 bl, f32x16& a1,
                                         f32x16& a2) {
  a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, a2, 0, 0, 0);
  a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, a2, 0, 0, 0);
  a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, a1, 0, 0, 0);
}

template <bool VEC>
__global__ __launch_bounds__(256) void attention_h2_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  _Float16* Kf = reinterpret_cast<_Float16*>(lds);          // [kc / 32][2 steps][hi, lo'][64 lanes][8]
  _Float16* Vf = Kf + (long)p.kc * 64;                       // same shape
  float* Ms = reinterpret_cast<float*>(Vf + (long)p.kc * 64);   // [kc] additive mask (0 / -inf)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int qbase = (blockIdx.x * 4 + wave) * 32;
  const float NEG_INF = -__builtin_inff();
  constexpr float LO_DOWN = 1.0f / 2048.0f;

  const float* qp = p.q + (long)b * p.q_bs + h * HD;
  const float* kp = p.k + (long)b * p.k_bs + h * HD;
  const float* vp = p.v + (long)b * p.v_bs + h * HD;

  // Q^T fragments (B operand): lane (query l31, half hh) of step t holds Q[query][16 t + 8 hh + e] * scale * log2(e)
  at_h16x8 qh[2], ql[2];
  {
    const int qi = qbase + l31;
    const bool ok = qi < p.Sq;
    const float* src = qp + (long)(ok ? qi : 0) * p.q_ss + 8 * hh;
    const float sc = p.scale * 1.44269504088896340736f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        _Float16 a, c;
        at_split(ok ? src[16 * t + e] * sc : 0.f, a, c);
        qh[t][e] = a;
        ql[t][e] = c;
      }
    }
  }

  float m_run = NEG_INF, l_half = 0.f;
  f32x16 o1 = {0}, o2 = {0};

  for (int c0 = 0; c0 < p.Sk; c0 += p.kc) {
    const int nk = min(p.kc, ((p.Sk - c0 + 31) / 32) * 32);   // keys in this chunk, padded to a tile
    __syncthreads();
    // ---- stage + split the K / V chunk: thread -> (row, 4 consecutive d)
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
      const int tile = r >> 5, kk = r & 31;
      {
        const float f[4] = {kv.x, kv.y, kv.z, kv.w};
        at_h16x4 a, c;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          _Float16 x, y;
          at_split(f[j], x, y);
          a[j] = x;
          c[j] = y;
        }
        _Float16* d = Kf + ((long)((tile * 2 + (d4 >> 4)) * 2) * 64 + ((d4 >> 3) & 1) * 32 + kk) * 8 + (d4 & 7);
        *reinterpret_cast<at_h16x4*>(d) = a;
        *reinterpret_cast<at_h16x4*>(d + 512) = c;
      }
      {
        const float f[4] = {vv.x, vv.y, vv.z, vv.w};
        // key kk of the tile is k-slot e of step t for lane-half vh:  kk = (e & 3) + 16 t + 8 (e >> 2) + 4 vh
        const int t = kk >> 4, vh = (kk >> 2) & 1, e = (kk & 3) + 4 * ((kk >> 3) & 1);
        _Float16* d = Vf + ((long)((tile * 2 + t) * 2) * 64 + vh * 32 + d4) * 8 + e;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          _Float16 x, y;
          at_split(f[j], x, y);
          d[8 * j] = x;
          d[8 * j + 512] = y;
        }
      }
    }
    for (int r = tid; r < nk; r += 256) {
      const int key = c0 + r;
      const bool dead = key >= p.Sk || (p.mask != nullptr && p.mask[(long)b * p.mask_ld + key] != 0);
      Ms[r] = dead ? NEG_INF : 0.f;
    }
    __syncthreads();

    for (int kt = 0; kt < nk; kt += 32) {
      const _Float16* kf = Kf + (long)(kt >> 5) * 2048 + lane * 8;
      const _Float16* vf = Vf + (long)(kt >> 5) * 2048 + lane * 8;
      // ---- scores^T tile (log2 domain): rows = keys, cols = queries
      f32x16 s1 = {0}, s2 = {0}, s;
#pragma unroll
      for (int t = 0; t < 2; ++t)
        at_mfma3(*reinterpret_cast<const at_h16x8*>(kf + t * 1024), *reinterpret_cast<const at_h16x8*>(kf + t * 1024 + 512), qh[t], ql[t], s1, s2);
      float tmax = NEG_INF;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = s1[r] + s2[r] * LO_DOWN + Ms[kt + acc_row(r, hh)];
        tmax = fmaxf(tmax, s[r]);
      }
      tmax = xor32_max(tmax);
      const float m_new = fmaxf(m_run, tmax);
      const bool none = m_new == NEG_INF;                 // every key so far masked for this query
      const float alpha = none ? 1.f : __builtin_amdgcn_exp2f(m_run - m_new);
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pr = none ? 0.f : __builtin_amdgcn_exp2f(s[r] - m_new);
        s[r] = pr;
        psum += pr;
      }
      l_half = l_half * alpha + psum;
      m_run = m_new;
      if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {   // some query's running maximum moved: rescale the output pair
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          o1[r] *= alpha;
          o2[r] *= alpha;
        }
      }
      // ---- O^T += V^T . P^T: the probabilities' register order is the k-slot order of the B operand
      at_h16x8 ph[2], pl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          _Float16 a, c;
          at_split(s[8 * t + e], a, c);
          ph[t][e] = a;
          pl[t][e] = c;
        }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
        at_mfma3(*reinterpret_cast<const at_h16x8*>(vf + t * 1024), *reinterpret_cast<const at_h16x8*>(vf + t * 1024 + 512), ph[t], pl[t], o1, o2);
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
      float4 t = make_float4((o1[4 * g + 0] + o2[4 * g + 0] * LO_DOWN) * inv, (o1[4 * g + 1] + o2[4 * g + 1] * LO_DOWN) * inv,
                             (o1[4 * g + 2] + o2[4 * g + 2] * LO_DOWN) * inv, (o1[4 * g + 3] + o2[4 * g + 3] * LO_DOWN) * inv);
      float* d = dst + 8 * g + 4 * hh;
      if (VEC) {
        *reinterpret_cast<float4*>(d) = t;
      } else {
        d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
      }
    }
  }
}

// != 0: this thread's attention cores take the exact-fp32 form whatever ocv_attention_set_dispatch says (ocv_attention_set_fp32_range: the
// fp16 range guard's fallback route, hip_ops.bf16_pairs -- the two-term fp16 core ends at +-65504 like the pairs it stands beside)
thread_local int g_attn_fp32_range = 0;
// 0 = the two-term fp16 core (default), 1 = exact fp32: ocv_attention_set_dispatch (process-wide; the A/B numerics route of tests, tools
// and bench.py's exact-fp32 leg -- set from OCV_ATTN_FORM by the Python side, objcavit_amd/hip_ops/tokens.py attention_form_sync)
int g_attn_form = 0;
}  // namespace

extern "C" int ocv_attention_fwd(const float* q, long q_bs, int q_ss, const float* k, long k_bs, int k_ss,
                                 const float* v, long v_bs, int v_ss, const uint8_t* key_padding_mask, float* ctx,
                                 long o_bs, int o_ss, int B, int H, int Sq, int Sk, float scale, ocv_stream_t stream) {
  return ocv_attention_launch(q, q_bs, q_ss, k, k_bs, k_ss, v, v_bs, v_ss, key_padding_mask, Sk, ctx, o_bs, o_ss, B, H,
                              Sq, Sk, scale, (hipStream_t)stream);
}

extern "C" int ocv_attention_set_fp32_range(int on) {
  g_attn_fp32_range = on != 0;
  return 0;
}

extern "C" int ocv_attention_set_dispatch(int form) {
  OCV_CHECK_ARG(form == 0 || form == 1, "ocv_attention_set_dispatch: form must be 0 (two-term fp16 core) or 1 (exact fp32 core), got %d", form);
  g_attn_form = form;
  return 0;
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
  if (g_attn_fp32_range == 0 && g_attn_form == 0) {
    static bool attr_h = false;
    if (!attr_h) {
      (void)hipFuncSetAttribute((const void*)attention_h2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)attention_h2_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_h = true;
    }
    // smaller key chunks than the exact kernel: 33 KB of LDS per 128 keys = four workgroups per CU, one's softmax / split
    // arithmetic under another's MFMAs (512-key chunks -- one workgroup per CU -- measured 170 us at S = 1200, B = 16)
    constexpr int kc_h2 = 128;                                 // keys per LDS chunk (64 / 256 measured no faster: profiles/r03_attention_h2.txt)
    if (a.kc > kc_h2 && kc_h2 >= 32 && kc_h2 % 32 == 0) a.kc = kc_h2;
    const size_t ldsh = (size_t)a.kc * (2 * 64 * sizeof(_Float16) + sizeof(float));
    if (vec)
      hipLaunchKernelGGL((attention_h2_kernel<true>), grid, block, ldsh, st, a);
    else
      hipLaunchKernelGGL((attention_h2_kernel<false>), grid, block, ldsh, st, a);
    OCV_CHECK_LAUNCH("ocv_attention_fwd(h2)");
    return 0;
  }
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
