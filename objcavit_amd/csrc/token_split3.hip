// Token-path linear layers on the bf16 matrix cores at fp32 accuracy: THREE-term split operands.
//
//   v = h + m + l,  h = bf16(v), m = bf16(v - h), l = bf16(v - h - m)      (24 significant bits; both subtractions exact)
//   a b ~= ah bh + ah bm + am bh + am bm + ah bl + al bh                     (dropped terms <= 2^-24 of the product)
//
// Six v_mfma_f32_32x32x16_bf16 (192 matrix-pipe cycles per 32 x 32 x 16 block) against eight v_mfma_f32_32x32x2_f32
// (512): the same results to fp32 rounding at 2.7x the matrix rate and ~1/2.7 of the matrix-core energy -- the forward
// runs at the package power cap once batches are pipelined, so joules per image are what is left
// to save.  The two-term split of the convolutions (2^-17 per product) is NOT used here: the token path feeds the
// queries of a near-one-hot bin softmax.
//
// Operands: activation rows are split once per workgroup while they are staged into LDS (three bf16 planes, rows padded
// to 272 bytes: conflict-free ds_read_b128 of 16 rows x one K octet); the static weights are split and packed ONCE on the
// device by ocv_pack_split3_fwd into MFMA B-operand fragments,
//   Wp[((jt * (K/16) + s) * 3 + part) * 512 + lane * 8 + e] = part of W[32 jt + (lane & 31)][16 s + 8 (lane >> 5) + e],
// so a wavefront's weight load is one contiguous 1 KB run per (channel tile, K step, part), straight from L2 into VGPRs
// (every weight element is used by exactly one wavefront of a workgroup).
//   lin3_kernel<NT, EPI>   out = epi(A W^T + b): 32 rows x (128 NT) columns per workgroup; epi = none | residual + LayerNorm
//   ffn3_kernel            out = LayerNorm(x + W2 relu(W1 x + b1) + b2), hidden units 128 at a time through LDS
// Replace nn.Linear / nn.MultiheadAttention projections / linear1 + linear2 of nn.TransformerEncoderLayer at
// modules/ObjCAViT.py:155-161,169,188 and modules/layers.py:8-9,23 (same lines as the fp32 kernels of csrc/linear.hip).
#include <math.h>
#include <stdlib.h>

#include "common.hpp"
#include "../../include/objcavit_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int TM = 32;               // rows per workgroup
constexpr int KC = 128;              // K chunk staged at a time
constexpr int PROW = KC + 8;         // bf16 per plane row (272 bytes)
constexpr int PLANE = TM * PROW;     // bf16 per plane
constexpr int E128 = 128;

__device__ __forceinline__ void split8x3(const float4 u, const float4 v, bf16x8& h, bf16x8& m, bf16x8& l) {
  const float f[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 a = (__bf16)f[i];
    const float r1 = f[i] - (float)a;
    const __bf16 b = (__bf16)r1;
    h[i] = a;
    m[i] = b;
    l[i] = (__bf16)(r1 - (float)b);
  }
}

// 256 threads stage rows [m0, m0 + 32) x columns [k0, k0 + kc) of a row-major fp32 matrix as three bf16 planes
__device__ __forceinline__ void stage_rows3(__bf16* planes, const float* __restrict__ src, int ld, int m0, int M, int k0, int kc,
                                            int tid) {
  const int row = tid >> 3;
  const bool ok = m0 + row < M;
  const float* s = src + (long)(ok ? m0 + row : 0) * ld + k0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int o = (tid & 7) + 8 * i;                     // K octet of the chunk
    float4 u = make_float4(0.f, 0.f, 0.f, 0.f), v = u;
    if (ok && 8 * o < kc) {
      u = ld4(s + 8 * o);
      v = ld4(s + 8 * o + 4);
    }
    bf16x8 h, m, l;
    split8x3(u, v, h, m, l);
    __bf16* d = planes + row * PROW + 8 * o;
    *reinterpret_cast<bf16x8*>(d) = h;
    *reinterpret_cast<bf16x8*>(d + PLANE) = m;
    *reinterpret_cast<bf16x8*>(d + 2 * PLANE) = l;
  }
}

__device__ __forceinline__ f32x16 mfma6(const bf16x8 ah, const bf16x8 am, const bf16x8 al, const bf16x8 bh, const bf16x8 bm,
                                        const bf16x8 bl, f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);      // small terms first
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
  return acc;
}

// The weight fragments of one channel tile over a whole K chunk, resident in VGPRs (8 K steps x 3 parts x 16 bytes per
// lane = 96 registers): loaded a phase ahead of their use, so the L2 latency of the stream sits under the previous
// phase's MFMAs (loaded just in time, each group of six MFMAs waited for its own three loads: the first version of
// these kernels ran SLOWER than the exact-fp32 ones, 160 us against 122 us per encoder layer).
struct WFrag3 { bf16x8 w[KC / 16][3]; };

__device__ __forceinline__ void load_w3(WFrag3& f, const __bf16* wp, int nsteps) {
#pragma unroll
  for (int s = 0; s < KC / 16; ++s) {
    if (s < nsteps) {
#pragma unroll
      for (int p = 0; p < 3; ++p) f.w[s][p] = *reinterpret_cast<const bf16x8*>(wp + (long)s * 1536 + p * 512);
    }
  }
}

// acc += planes[32 x (16 nsteps)] . (resident fragments)^T
__device__ __forceinline__ f32x16 chunk_mfma3(f32x16 acc, const __bf16* planes, const WFrag3& f, int nsteps, int l31, int hh) {
  const __bf16* pa = planes + l31 * PROW + 8 * hh;
#pragma unroll
  for (int s = 0; s < KC / 16; ++s) {
    if (s < nsteps) {
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(pa + 16 * s);
      const bf16x8 am = *reinterpret_cast<const bf16x8*>(pa + 16 * s + PLANE);
      const bf16x8 al = *reinterpret_cast<const bf16x8*>(pa + 16 * s + 2 * PLANE);
      acc = mfma6(ah, am, al, f.w[s][0], f.w[s][1], f.w[s][2], acc);
    }
  }
  return acc;
}

// residual + LayerNorm over the 128 columns of a 32-row tile held in Cs (one wavefront per 8 rows)
__device__ __forceinline__ void ln_rows3(const float (*Cs)[E128 + 1], const float* gamma, const float* beta, float eps,
                                         const uint8_t* zero_mask, float* out, int ldo, int m0, int M, int lane, int wave) {
  const float g0 = gamma[lane], g1 = gamma[lane + 64];
  const float b0 = beta[lane], b1 = beta[lane + 64];
#pragma unroll
  for (int i = 0; i < TM / 4; ++i) {
    const int row = wave * (TM / 4) + i, m = m0 + row;
    const float x0 = Cs[row][lane], x1 = Cs[row][lane + 64];
    const float mean = wave_sum(x0 + x1) * (1.0f / E128);
    const float d0 = x0 - mean, d1 = x1 - mean;
    const float var = wave_sum(d0 * d0 + d1 * d1) * (1.0f / E128);
    const float rstd = 1.0f / sqrtf(var + eps);
    if (m < M) {
      const bool z = zero_mask != nullptr && zero_mask[m] != 0;
      out[(long)m * ldo + lane] = z ? 0.f : d0 * rstd * g0 + b0;
      out[(long)m * ldo + lane + 64] = z ? 0.f : d1 * rstd * g1 + b1;
    }
  }
}

enum { L3_NONE = 0, L3_RELU = 1, L3_LEAKY = 2, L3_RES_LN = 3 };

struct L3Args {
  const float* A; int lda;
  const __bf16* Wp;
  const float* bias;
  float* out; int ldo;
  int M, N, K;
  const float* res; int ldres;
  const float *gamma, *beta; float eps;
  const uint8_t* zero_mask;
};

template <int NT, int EPI>
__global__ __launch_bounds__(256) void lin3_kernel(L3Args p) {
  __shared__ __attribute__((aligned(16))) __bf16 planes[3 * PLANE];
  __shared__ float Cs[EPI == L3_RES_LN ? TM : 1][E128 + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int m0 = blockIdx.x * TM;
  const int nsteps_all = (p.K + 15) >> 4, ntl = (p.N + 31) >> 5;
  int jt[NT];
  const __bf16* wp[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    jt[t] = (blockIdx.y * 4 + wave) * NT + t;
    wp[t] = p.Wp + ((long)min(jt[t], ntl - 1) * nsteps_all * 3) * 512 + lane * 8;
  }
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x16{0};
  WFrag3 fa, fb;                                           // ping-pong: tile t multiplies from one while t + 1 loads into the other
  for (int k0 = 0; k0 < p.K; k0 += KC) {
    const int kc = min(KC, p.K - k0), ns = (kc + 15) >> 4;
    load_w3(fa, wp[0] + (long)(k0 >> 4) * 1536, ns);
    __syncthreads();
    stage_rows3(planes, p.A, p.lda, m0, p.M, k0, kc, tid);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      WFrag3& cur = (t & 1) ? fb : fa;
      WFrag3& nxt = (t & 1) ? fa : fb;
      if (t + 1 < NT) load_w3(nxt, wp[t + 1] + (long)(k0 >> 4) * 1536, ns);
      acc[t] = chunk_mfma3(acc[t], planes, cur, ns, l31, hh);
    }
  }
  if (EPI != L3_RES_LN) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = jt[t] * 32 + l31;
      if (n >= p.N) continue;
      const float bn = p.bias != nullptr ? p.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + acc_row(r, hh);
        float v = acc[t][r] + bn;
        if (EPI == L3_RELU) v = fmaxf(v, 0.f);
        if (EPI == L3_LEAKY) v = v > 0.f ? v : 0.01f * v;
        if (m < p.M) p.out[(long)m * p.ldo + n] = v;
      }
    }
  } else {
    const int n = wave * 32 + l31;                       // N == 128, NT == 1, gridDim.y == 1
    const float bn = p.bias != nullptr ? p.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = acc_row(r, hh), m = m0 + row;
      float v = acc[0][r] + bn;
      if (m < p.M) v += p.res[(long)m * p.ldres + n];
      Cs[row][n] = v;
    }
    __syncthreads();
    ln_rows3(Cs, p.gamma, p.beta, p.eps, p.zero_mask, p.out, p.ldo, m0, p.M, lane, wave);
  }
}

struct F3Args {
  const float* x;
  const __bf16 *w1p, *w2p;
  const float *b1, *b2, *gamma, *beta;
  float eps;
  const uint8_t* zero_mask;
  float* out;
  int M, FF;
  float* part;           // nsplit > 1: raw partial outputs [nsplit][M][128]
  int nsplit;
};

__global__ __launch_bounds__(256) void ffn3_kernel(F3Args p) {
  __shared__ __attribute__((aligned(16))) __bf16 xp[3 * PLANE];
  __shared__ __attribute__((aligned(16))) __bf16 hp[3 * PLANE];
  __shared__ float Cs[TM][E128 + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int m0 = blockIdx.x * TM;
  const int col = wave * 32 + l31;
  const int nchunk_all = p.FF / KC;
  const int cbeg = nchunk_all * (int)blockIdx.y / p.nsplit, cend = nchunk_all * ((int)blockIdx.y + 1) / p.nsplit;
  const int ksteps2 = p.FF >> 4;                         // K steps of W2 (K = FF)

  stage_rows3(xp, p.x, E128, m0, p.M, 0, E128, tid);
  f32x16 acc = {0};
  WFrag3 f1, f2;
  load_w3(f1, p.w1p + ((long)(cbeg * 4 + wave) * (E128 / 16) * 3) * 512 + lane * 8, E128 / 16);
  __syncthreads();
  for (int c = cbeg; c < cend; ++c) {
    // this chunk's W2 fragments: in flight during phase 1
    load_w3(f2, p.w2p + (((long)wave * ksteps2 + c * (KC / 16)) * 3) * 512 + lane * 8, KC / 16);
    // phase 1: hidden units [128 c + 32 wave, + 32) of the tile's rows
    f32x16 h = {0};
    h = chunk_mfma3(h, xp, f1, E128 / 16, l31, hh);
    const float b1 = p.b1[c * KC + col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = fmaxf(h[r] + b1, 0.f);
      const __bf16 a = (__bf16)v;
      const float r1 = v - (float)a;
      const __bf16 b = (__bf16)r1;
      __bf16* d = hp + acc_row(r, hh) * PROW + col;
      d[0] = a;
      d[PLANE] = b;
      d[2 * PLANE] = (__bf16)(r1 - (float)b);
    }
    // the next chunk's W1 fragments: in flight during phase 2
    if (c + 1 < cend) load_w3(f1, p.w1p + ((long)((c + 1) * 4 + wave) * (E128 / 16) * 3) * 512 + lane * 8, E128 / 16);
    __syncthreads();
    // phase 2: out[:, 32 wave ..] += H_chunk . W2[:, 128 c ..]^T
    acc = chunk_mfma3(acc, hp, f2, KC / 16, l31, hh);
    __syncthreads();
  }
  if (p.nsplit > 1) {
    float* dst = p.part + ((long)blockIdx.y * p.M + m0) * E128 + col;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = acc_row(r, hh);
      if (m0 + row < p.M) dst[(long)row * E128] = acc[r];
    }
    return;
  }
  const float b2 = p.b2[col];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = acc_row(r, hh), m = m0 + row;
    Cs[row][col] = acc[r] + b2 + (m < p.M ? p.x[(long)m * E128 + col] : 0.f);
  }
  __syncthreads();
  ln_rows3(Cs, p.gamma, p.beta, p.eps, p.zero_mask, p.out, E128, m0, p.M, lane, wave);
}

// out = LayerNorm(x + sum_s part[s] + b2), partials added in split order (second pass of a split FFN)
__global__ __launch_bounds__(256) void ffn3_finish_kernel(F3Args p) {
  __shared__ float Cs[TM][E128 + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * TM;
  for (int i = tid; i < TM * E128; i += 256) {
    const int row = i / E128, col = i % E128;
    const long m = m0 + row;
    float v = 0.f;
    if (m < p.M) {
      v = p.part[m * E128 + col];
      for (int sidx = 1; sidx < p.nsplit; ++sidx) v += p.part[((long)sidx * p.M + m) * E128 + col];
      v += p.b2[col] + p.x[m * E128 + col];
    }
    Cs[row][col] = v;
  }
  __syncthreads();
  ln_rows3(Cs, p.gamma, p.beta, p.eps, p.zero_mask, p.out, E128, m0, p.M, lane, wave);
}

// ---------------------------------------------------------------------------
// Everything of a post-norm transformer layer that is LOCAL to a token, in one launch per 32-token tile:
//   x1 = LayerNorm1(x + ctx Wo^T + bo)            (ctx = the attention kernel's output for these tokens)
//   x2 = LayerNorm2(x1 + W2 relu(W1 x1 + b1) + b2)          -> out
//   qkv_next = x2 Wqkv'^T + bqkv'                           -> the NEXT layer's packed q | k | v rows (if there is one)
// so a layer is two launches -- attention, this -- instead of four, and the next layer's projection rides along.  The
// four separate launches cost 13 - 17 us each for ~5 us of arithmetic (a cold start per launch: row staging, first
// weight fragments, tail), 112 us per layer.  x1 and x2 never leave the workgroup between the phases (fp32 copy in LDS
// for the residuals, three-term planes for the next contraction).  Same arithmetic, same order as lin3 + ffn3 + lin3.
// ---------------------------------------------------------------------------
struct TailArgs {
  const float *ctx, *x;                 // [M][128]
  const __bf16 *wo_p, *w1_p, *w2_p, *wqkv_p;     // wqkv_p: the next layer's packed in_proj (nullable)
  const float *bo, *g1, *be1, *b1, *b2, *g2, *be2, *bqkv;
  float eps;
  const uint8_t* zero_mask;             // rows written as 0 in `out` (last layer of a masked stack), nullable
  float* out;                           // [M][128]
  float* qkv;                           // [M][384] (nullable with wqkv_p)
  int M, FF;
};

// Row tiles per workgroup.  RT = 2 (64 tokens: every streamed weight fragment serves two MFMA tiles, half the L2 traffic)
// was measured SLOWER, 92 us against 60 us per launch: the launch is not bound by L2 bandwidth but by the chain of 2 x 8
// phases per workgroup, each waiting ~2 - 3 us for weight fragments requested one phase (48 MFMAs, 0.75 us) earlier, and
// doubling the rows doubles every phase's MFMA time without shortening any wait.  Kept as a parameter.
constexpr int RT = 1;
constexpr int TR = TM * RT;              // rows per workgroup

// LayerNorm of the TR x 128 tile in Cs, IN PLACE (one wavefront per TR / 4 rows)
__device__ __forceinline__ void ln_tile_inplace(float (*Cs)[E128 + 1], const float* gamma, const float* beta, float eps, int lane,
                                                int wave) {
  const float g0 = gamma[lane], g1 = gamma[lane + 64];
  const float b0 = beta[lane], b1 = beta[lane + 64];
#pragma unroll 4
  for (int i = 0; i < TR / 4; ++i) {
    const int row = wave * (TR / 4) + i;
    const float x0 = Cs[row][lane], x1 = Cs[row][lane + 64];
    const float mean = wave_sum(x0 + x1) * (1.0f / E128);
    const float d0 = x0 - mean, d1 = x1 - mean;
    const float var = wave_sum(d0 * d0 + d1 * d1) * (1.0f / E128);
    const float rstd = 1.0f / sqrtf(var + eps);
    Cs[row][lane] = d0 * rstd * g0 + b0;
    Cs[row][lane + 64] = d1 * rstd * g1 + b1;
  }
}

// the TR x 128 fp32 tile in Cs -> three bf16 planes per row tile (thread: row tid / 8 of each tile, two K octets)
__device__ __forceinline__ void tile_to_planes(__bf16* planes, const float (*Cs)[E128 + 1], int tid) {
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int row = tid >> 3;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int o = (tid & 7) + 8 * i;
      const float* c = &Cs[rt * TM + row][8 * o];
      bf16x8 h, m, l;
      split8x3(make_float4(c[0], c[1], c[2], c[3]), make_float4(c[4], c[5], c[6], c[7]), h, m, l);
      __bf16* d = planes + rt * 3 * PLANE + row * PROW + 8 * o;
      *reinterpret_cast<bf16x8*>(d) = h;
      *reinterpret_cast<bf16x8*>(d + PLANE) = m;
      *reinterpret_cast<bf16x8*>(d + 2 * PLANE) = l;
    }
  }
}

__global__ __launch_bounds__(256) void layer_tail3_kernel(TailArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tail_lds[];
  __bf16* xp = reinterpret_cast<__bf16*>(tail_lds);                   // [RT][3 planes]: ctx, then x1, then x2
  __bf16* hp = xp + RT * 3 * PLANE;                                   // [RT][3 planes]: hidden chunk
  float (*Cs)[E128 + 1] = reinterpret_cast<float (*)[E128 + 1]>(hp + RT * 3 * PLANE);   // [TR][129] fp32: sums, x1, x2
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int m0 = blockIdx.x * TR;
  const int col = wave * 32 + l31;
  const int ksteps2 = p.FF >> 4, nchunk = p.FF / KC;

  // ---- x1 = LN1(x + ctx Wo^T + bo)
  WFrag3 f1, f2;
  load_w3(f1, p.wo_p + ((long)wave * (E128 / 16) * 3) * 512 + lane * 8, E128 / 16);
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) stage_rows3(xp + rt * 3 * PLANE, p.ctx, E128, m0 + rt * TM, p.M, 0, E128, tid);
  __syncthreads();
  {
    f32x16 a[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) a[rt] = chunk_mfma3(f32x16{0}, xp + rt * 3 * PLANE, f1, E128 / 16, l31, hh);
    load_w3(f1, p.w1_p + ((long)wave * (E128 / 16) * 3) * 512 + lane * 8, E128 / 16);      // W1, first chunk: under LN1
    const float bo = p.bo[col];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rt * TM + acc_row(r, hh), m = m0 + row;
        Cs[row][col] = a[rt][r] + bo + (m < p.M ? p.x[(long)m * E128 + col] : 0.f);
      }
  }
  __syncthreads();
  ln_tile_inplace(Cs, p.g1, p.be1, p.eps, lane, wave);
  __syncthreads();
  tile_to_planes(xp, Cs, tid);                        // every wavefront is past its reads of the ctx planes
  __syncthreads();

  // ---- x2 = LN2(x1 + W2 relu(W1 x1 + b1) + b2)
  f32x16 acc[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x16{0};
  for (int c = 0; c < nchunk; ++c) {
    load_w3(f2, p.w2_p + (((long)wave * ksteps2 + c * (KC / 16)) * 3) * 512 + lane * 8, KC / 16);
    const float b1 = p.b1[c * KC + col];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const f32x16 h = chunk_mfma3(f32x16{0}, xp + rt * 3 * PLANE, f1, E128 / 16, l31, hh);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = fmaxf(h[r] + b1, 0.f);
        const __bf16 a = (__bf16)v;
        const float r1 = v - (float)a;
        const __bf16 b = (__bf16)r1;
        __bf16* d = hp + rt * 3 * PLANE + acc_row(r, hh) * PROW + col;
        d[0] = a;
        d[PLANE] = b;
        d[2 * PLANE] = (__bf16)(r1 - (float)b);
      }
    }
    if (c + 1 < nchunk) load_w3(f1, p.w1_p + ((long)((c + 1) * 4 + wave) * (E128 / 16) * 3) * 512 + lane * 8, E128 / 16);
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = chunk_mfma3(acc[rt], hp + rt * 3 * PLANE, f2, KC / 16, l31, hh);
    __syncthreads();
  }
  const bool next = p.wqkv_p != nullptr;
  if (next) load_w3(f1, p.wqkv_p + ((long)(wave * 3) * (E128 / 16) * 3) * 512 + lane * 8, E128 / 16);     // under LN2
  {
    const float b2 = p.b2[col];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rt * TM + acc_row(r, hh);
        Cs[row][col] = acc[rt][r] + b2 + Cs[row][col];   // residual = x1 (each element read and written by one lane)
      }
  }
  __syncthreads();
  ln_tile_inplace(Cs, p.g2, p.be2, p.eps, lane, wave);
  __syncthreads();
  for (int i = tid; i < TR * (E128 / 4); i += 256) {     // x2 -> out, 16-byte stores
    const int row = i / (E128 / 4), c4 = (i % (E128 / 4)) * 4, m = m0 + row;
    if (m < p.M) {
      const bool z = p.zero_mask != nullptr && p.zero_mask[m] != 0;
      const float4 v = z ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(Cs[row][c4], Cs[row][c4 + 1], Cs[row][c4 + 2], Cs[row][c4 + 3]);
      *reinterpret_cast<float4*>(p.out + (long)m * E128 + c4) = v;
    }
  }
  if (!next) return;

  // ---- the next layer's packed projection: wavefront w -> channel tiles 3 w .. 3 w + 2 of the 12
  tile_to_planes(xp, Cs, tid);
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    WFrag3& cur = (t & 1) ? f2 : f1;
    WFrag3& nxt = (t & 1) ? f1 : f2;
    if (t + 1 < 3) load_w3(nxt, p.wqkv_p + ((long)(wave * 3 + t + 1) * (E128 / 16) * 3) * 512 + lane * 8, E128 / 16);
    const int n = (wave * 3 + t) * 32 + l31;
    const float bn = p.bqkv[n];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const f32x16 q = chunk_mfma3(f32x16{0}, xp + rt * 3 * PLANE, cur, E128 / 16, l31, hh);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + rt * TM + acc_row(r, hh);
        if (m < p.M) p.qkv[(long)m * 3 * E128 + n] = q[r] + bn;
      }
    }
  }
}

// W [N][K] fp32 (row stride ldw) -> packed three-term fragments; one thread per (jt, s, lane)
__global__ __launch_bounds__(256) void pack3_kernel(const float* __restrict__ W, int ldw, int N, int K, __bf16* __restrict__ out,
                                                    long items) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= items) return;
  const int lane = (int)(i & 63);
  const long js = i >> 6;
  const int nsteps = (K + 15) >> 4;
  const int s = (int)(js % nsteps), jt = (int)(js / nsteps);
  const int n = jt * 32 + (lane & 31), k0 = 16 * s + 8 * (lane >> 5);
  float f[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = (n < N && k0 + e < K) ? W[(long)n * ldw + k0 + e] : 0.f;
  bf16x8 h, m, l;
  split8x3(make_float4(f[0], f[1], f[2], f[3]), make_float4(f[4], f[5], f[6], f[7]), h, m, l);
  __bf16* d = out + (js * 3) * 512 + lane * 8;
  *reinterpret_cast<bf16x8*>(d) = h;
  *reinterpret_cast<bf16x8*>(d + 512) = m;
  *reinterpret_cast<bf16x8*>(d + 1024) = l;
}

template <int NT>
int launch_lin3(const L3Args& a, int act, hipStream_t st) {
  dim3 grid(ocv_cdiv(a.M, TM), ocv_cdiv(a.N, 128 * NT)), block(256);
  switch (act) {
    case OCV_ACT_RELU: hipLaunchKernelGGL((lin3_kernel<NT, L3_RELU>), grid, block, 0, st, a); break;
    case OCV_ACT_LEAKY_RELU: hipLaunchKernelGGL((lin3_kernel<NT, L3_LEAKY>), grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL((lin3_kernel<NT, L3_NONE>), grid, block, 0, st, a); break;
  }
  OCV_CHECK_LAUNCH("ocv_linear_split3_fwd");
  return 0;
}

// ---------------------------------------------------------------------------
// Few-key multi-head attention, the image <- object cross-attention (modules/ObjCAViT.py:192-201): at most 32 live keys
// per image, E = 128, 4 heads of 32.  Round 2's single launch (csrc/linear.hip, cross_attn_fused_kernel) spent 128 of its
// 288 exact-fp32 MFMAs per wavefront RE-PROJECTING K and V for every 32-query tile and was bound by that dependent
// matrix chain (4 % of the HBM roofline at bs = 16, 7.6 % at bs >= 512: profiles/r02_cross_attention_roofline.txt).  Here
//   xattn_kv3_kernel    K = Xk Wk^T + bk, V = Xv Wv^T + bv ONCE per image (32 rows, a 32 KB record the query tiles of the
//                       image then read from L2), and
//   xattn_main3_kernel  per 32-query tile: wavefront h projects head h of Q TRANSPOSED (weights as the MFMA A operand,
//                       query rows as B), so its accumulator IS the B operand of the score MFMAs -- Q never touches LDS
//                       --, scores / softmax / context exactly as before on exact fp32 MFMA (both operands are
//                       activations), then 32 columns of the output projection.
// The four projections are three-term-split contractions (six bf16 MFMAs per product block, dropped terms <= 2^-24) on
// weights packed once by ocv_pack_split3_fwd: 48 + 48 bf16 MFMAs (32 cycles) + 32 fp32 MFMAs (64 cycles) per wavefront =
// 5.1 K matrix-pipe cycles against 18.4 K.
// ---------------------------------------------------------------------------
struct XKV3Args {
  const float *k_src, *v_src;          // [B][Sk][128]
  const __bf16* in_p3;                 // packed in_proj_weight [384][128]
  const float* in_b;                   // [384]
  float* kv;                           // [B][2][4 heads][64 lanes][16]: the fragments xattn_main3_kernel's lanes consume
  int Sk, Se;
};

// The record is written in the ORDER THE QUERY TILES' LANES READ IT (round 3b): K projected transposed (weights as the A
// operand, like Q below), so lane (key l31, half hh) of wavefront h holds K[key][32 h + acc_row(r, hh)], r = 0..15 -- the 16
// A-operand values of its 16 score MFMAs; V projected straight, so lane (d l31, half hh) holds V[acc_row(r, hh)][32 h + d] --
// the A operands of its 16 context MFMAs.  Each lane stores 64 contiguous bytes, the query tiles load them back with four
// 16-byte loads per operand and K / V never pass through LDS (the tiled form parked the record at a 129-float row pitch:
// 33 KB of LDS and 32 four-way-conflicting scalar stores per thread and tile, which held the kernel at two workgroups per CU).
__global__ __launch_bounds__(256) void xattn_kv3_kernel(XKV3Args p) {
  __shared__ __attribute__((aligned(16))) __bf16 planes[3 * PLANE];
  const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const long b = blockIdx.x;
  const bool is_v = blockIdx.y != 0;                  // one workgroup per image and operand: two short chains instead of one long
  constexpr int NS = E128 / 16;
  WFrag3 f;
  load_w3(f, p.in_p3 + ((long)((is_v ? 8 : 4) + h) * NS * 3) * 512 + lane * 8, NS);   // Wk / Wv rows of head h
  stage_rows3(planes, (is_v ? p.v_src : p.k_src) + b * p.Sk * E128, E128, 0, p.Se, 0, E128, tid);
  __syncthreads();
  float* d = p.kv + (((b * 2 + (is_v ? 1 : 0)) * 4 + h) * 64 + lane) * 16;
  f32x16 a = {0};
  if (!is_v) {
    // K^T of head h: D[d][key] = sum_k Wk[32 h + d][k] X[key][k]
    const __bf16* pa = planes + l31 * PROW + 8 * hh;
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      const bf16x8 xh = *reinterpret_cast<const bf16x8*>(pa + 16 * st);
      const bf16x8 xm = *reinterpret_cast<const bf16x8*>(pa + 16 * st + PLANE);
      const bf16x8 xl = *reinterpret_cast<const bf16x8*>(pa + 16 * st + 2 * PLANE);
      a = mfma6(f.w[st][0], f.w[st][1], f.w[st][2], xh, xm, xl, a);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] += p.in_b[E128 + h * 32 + acc_row(r, hh)];
  } else {
    const __bf16* pa = planes + l31 * PROW + 8 * hh;
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(pa + 16 * st);
      const bf16x8 am = *reinterpret_cast<const bf16x8*>(pa + 16 * st + PLANE);
      const bf16x8 al = *reinterpret_cast<const bf16x8*>(pa + 16 * st + 2 * PLANE);
      a = mfma6(ah, am, al, f.w[st][0], f.w[st][1], f.w[st][2], a);
    }
    const float bv = p.in_b[2 * E128 + h * 32 + l31];
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] += bv;
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(d + 4 * g) = make_float4(a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]);
}

struct XA3Args {
  const float* q_src;                  // [B][Sq][128]
  const uint8_t* mask;                 // [B][mask_ld] (nullable)
  const float* kv;                     // [B][2][4][64][16] from xattn_kv3_kernel
  const __bf16 *in_p3, *out_p3;
  const float *in_b, *out_b;
  float* out;                          // [B][Sq][128]
  int Sq, Se, mask_ld;
  float scale;
};

// One workgroup = NSUB sub-tiles of 32 queries on ONE load of the weight fragments.  What a tile costs (ablation builds at bs
// 512, profiles/r03_cross_attention_roofline.txt): the two 24 KB fragment sets per wavefront are 192 KB per workgroup through
// the CU's 64 B/clk vector-memory path = 3.1 K cycles, beside 5.1 K of matrix-pipe time -- NSUB = 2 halves the former per query.
// Both projections run TRANSPOSED (weights = A operand): Q^T's accumulator is the score MFMAs' B operand, and the output's
// accumulator holds four consecutive columns of one query row per register quad -> 16-byte stores (sixteen 4-byte stores per
// lane cost 1.9 K cycles of store issue per tile).
template <int NSUB>
__global__ __launch_bounds__(256, 3) void xattn_main3_kernel(XA3Args p) {
  __shared__ __attribute__((aligned(16))) __bf16 planes[NSUB * 3 * PLANE];       // query rows, later the context rows (3 planes each)
  __shared__ float Ms[32];
  const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const long b = blockIdx.y;
  const int q0 = blockIdx.x * TM * NSUB;
  constexpr float NEG_INF = -__builtin_inff();
  constexpr int NS = E128 / 16;

  WFrag3 fq;
  load_w3(fq, p.in_p3 + ((long)h * NS * 3) * 512 + lane * 8, NS);                // Wq rows of head h
#pragma unroll
  for (int u = 0; u < NSUB; ++u) stage_rows3(planes + u * 3 * PLANE, p.q_src + b * p.Sq * E128, E128, q0 + u * TM, p.Sq, 0, E128, tid);
  if (tid < 32) Ms[tid] = (tid >= p.Se || (p.mask != nullptr && p.mask[b * p.mask_ld + tid] != 0)) ? NEG_INF : 0.f;
  const float* kf = p.kv + ((b * 2 * 4 + h) * 64 + lane) * 16;                  // this lane's 16 K and 16 V operands (L2)
  float4 kr[4], vr[4];
  if (NSUB == 1) {
#pragma unroll
    for (int g = 0; g < 4; ++g) kr[g] = ld4(kf + 4 * g);                         // in flight under the Q projection
  }
  __syncthreads();

  // Q^T of head h: D[d][query] = sum_k Wq[32 h + d][k] X[query][k]  (weight fragment = A operand, row fragment = B operand)
  f32x16 q[NSUB];
#pragma unroll
  for (int u = 0; u < NSUB; ++u) {
    q[u] = f32x16{0};
    const __bf16* pa = planes + u * 3 * PLANE + l31 * PROW + 8 * hh;
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      const bf16x8 xh = *reinterpret_cast<const bf16x8*>(pa + 16 * st);
      const bf16x8 xm = *reinterpret_cast<const bf16x8*>(pa + 16 * st + PLANE);
      const bf16x8 xl = *reinterpret_cast<const bf16x8*>(pa + 16 * st + 2 * PLANE);
      q[u] = mfma6(fq.w[st][0], fq.w[st][1], fq.w[st][2], xh, xm, xl, q[u]);
    }
  }
  if (NSUB > 1) {                                                                 // two query accumulators live: K after them
#pragma unroll
    for (int g = 0; g < 4; ++g) kr[g] = ld4(kf + 4 * g);
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) vr[g] = ld4(kf + 4 * 64 * 16 + 4 * g);            // V operands: under the score MFMAs
  float4 bq[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bq[g] = ld4(p.in_b + h * 32 + 8 * g + 4 * hh);    // bias of rows acc_row(4 g .. 4 g + 3, hh)
  __syncthreads();                                   // every wavefront has read the query planes: they may take the context
#pragma unroll
  for (int u = 0; u < NSUB; ++u) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      q[u][4 * g] = (q[u][4 * g] + bq[g].x) * p.scale;
      q[u][4 * g + 1] = (q[u][4 * g + 1] + bq[g].y) * p.scale;
      q[u][4 * g + 2] = (q[u][4 * g + 2] + bq[g].z) * p.scale;
      q[u][4 * g + 3] = (q[u][4 * g + 3] + bq[g].w) * p.scale;
    }
  }
  WFrag3 fo;
  const float kop[16] = {kr[0].x, kr[0].y, kr[0].z, kr[0].w, kr[1].x, kr[1].y, kr[1].z, kr[1].w,
                         kr[2].x, kr[2].y, kr[2].z, kr[2].w, kr[3].x, kr[3].y, kr[3].z, kr[3].w};
  const float vop[16] = {vr[0].x, vr[0].y, vr[0].z, vr[0].w, vr[1].x, vr[1].y, vr[1].z, vr[1].w,
                         vr[2].x, vr[2].y, vr[2].z, vr[2].w, vr[3].x, vr[3].y, vr[3].z, vr[3].w};
#pragma unroll
  for (int u = 0; u < NSUB; ++u) {
    // scores^T (rows = keys, columns = queries): step st contracts d = acc_row(st, 0 | 1) -- register st of q IS the B operand
    f32x16 s = {0};
#pragma unroll
    for (int st = 0; st < 16; ++st) s = mfma_32x32x2(kop[st], q[u][st], s);
    // Wo rows 32 h ..: requested once no query accumulator is live any more, in flight under the last softmax / context
    if (u == NSUB - 1) load_w3(fo, p.out_p3 + ((long)h * NS * 3) * 512 + lane * 8, NS);
    float tmax = NEG_INF;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] += Ms[acc_row(r, hh)];
      tmax = fmaxf(tmax, s[r]);
    }
    tmax = xor32_max(tmax);
    const bool none = tmax == NEG_INF;               // every key masked for this image: 0 / 0 = NaN, as torch
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pr = none ? 0.f : fast_exp(s[r] - tmax);
      s[r] = pr;
      psum += pr;
    }
    const float inv = 1.0f / xor32_sum(psum);
    f32x16 o = {0};
#pragma unroll
    for (int r = 0; r < 16; ++r) o = mfma_32x32x2(vop[r], s[r], o);
    // o: register r = context[query l31][d = acc_row(r, hh)] of head h -> three-term planes, row = query, column 32 h + d
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = o[r] * inv;
      const __bf16 a = (__bf16)v;
      const float r1 = v - (float)a;
      const __bf16 m = (__bf16)r1;
      __bf16* d = planes + u * 3 * PLANE + l31 * PROW + h * 32 + acc_row(r, hh);
      d[0] = a;
      d[PLANE] = m;
      d[2 * PLANE] = (__bf16)(r1 - (float)m);
    }
  }
  float4 bo[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bo[g] = ld4(p.out_b + h * 32 + 8 * g + 4 * hh);
  __syncthreads();

  // output projection TRANSPOSED, 32 columns per wavefront: register 4 g + j = out[query l31][32 h + 8 g + 4 hh + j]
#pragma unroll
  for (int u = 0; u < NSUB; ++u) {
    f32x16 acc = {0};
    const __bf16* pa = planes + u * 3 * PLANE + l31 * PROW + 8 * hh;
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      const bf16x8 ch = *reinterpret_cast<const bf16x8*>(pa + 16 * st);
      const bf16x8 cm = *reinterpret_cast<const bf16x8*>(pa + 16 * st + PLANE);
      const bf16x8 cl = *reinterpret_cast<const bf16x8*>(pa + 16 * st + 2 * PLANE);
      acc = mfma6(fo.w[st][0], fo.w[st][1], fo.w[st][2], ch, cm, cl, acc);
    }
    const int qi = q0 + u * TM + l31;
    if (qi < p.Sq) {
      float* od = p.out + (b * p.Sq + qi) * E128 + h * 32 + 4 * hh;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(od + 8 * g) = make_float4(acc[4 * g] + bo[g].x, acc[4 * g + 1] + bo[g].y, acc[4 * g + 2] + bo[g].z, acc[4 * g + 3] + bo[g].w);
    }
  }
}

}  // namespace

// fused few-key attention on packed three-term-split weights; returns 1 when the shape is not covered
int ocv_cross_attn_split3_launch(const float* q_src, const float* k_src, const float* v_src, const uint8_t* mask, int mask_ld,
                                 const void* in_p3, const float* in_b, const void* out_p3, const float* out_b, float* out,
                                 float* kv_ws, int B, int Sq, int Sk, int Se, int E, int H, hipStream_t st) {
  if (E != E128 || H != 4 || Se < 1 || Se > 32 || B > 65535) return 1;
  if (!(ocv_aligned16(q_src) && ocv_aligned16(k_src) && ocv_aligned16(v_src) && ocv_aligned16(in_p3) && ocv_aligned16(out_p3) && ocv_aligned16(kv_ws) &&
        ocv_aligned16(in_b) && ocv_aligned16(out_b) && ocv_aligned16(out))) return 1;
  XKV3Args ka{k_src, v_src, (const __bf16*)in_p3, in_b, kv_ws, Sk, Se};
  hipLaunchKernelGGL(xattn_kv3_kernel, dim3(B, 2), dim3(256), 0, st, ka);
  OCV_CHECK_LAUNCH("ocv_mha_split3_fwd(K / V projection)");
  XA3Args a{q_src, mask, kv_ws, (const __bf16*)in_p3, (const __bf16*)out_p3, in_b, out_b, out, Sq, Se, mask_ld, 1.0f / sqrtf(32.0f)};
  // two sub-tiles per workgroup once the 64-query workgroups alone fill the chip several times over (256 CUs x 3 resident):
  // 110 against 117 us at bs 512, S = 300; slower below (bs 128: 44.9 against 41.8 us)
  const long wg64 = (long)ocv_cdiv(Sq, 2 * TM) * B;
  const int nsub = wg64 >= 2048 ? 2 : 1;
  if (nsub == 2)
    hipLaunchKernelGGL(xattn_main3_kernel<2>, dim3(ocv_cdiv(Sq, 2 * TM), B), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(xattn_main3_kernel<1>, dim3(ocv_cdiv(Sq, TM), B), dim3(256), 0, st, a);
  OCV_CHECK_LAUNCH("ocv_mha_split3_fwd(fused)");
  return 0;
}

extern "C" size_t ocv_split3_packed_elems(int N, int K) {
  if (N < 1 || K < 1) return 0;
  return (size_t)((N + 31) / 32) * ((K + 15) / 16) * 3 * 512;
}

extern "C" int ocv_pack_split3_fwd(const float* W, int ldw, int N, int K, void* packed, ocv_stream_t stream) {
  OCV_CHECK_ARG(W && packed, "ocv_pack_split3_fwd: null pointer");
  OCV_CHECK_ARG(N >= 1 && K >= 1 && ldw >= K, "ocv_pack_split3_fwd: bad sizes N=%d K=%d ldw=%d", N, K, ldw);
  OCV_CHECK_ARG(ocv_aligned16(packed), "ocv_pack_split3_fwd: packed must be 16-byte aligned");
  const long items = (long)((N + 31) / 32) * ((K + 15) / 16) * 64;
  hipLaunchKernelGGL(pack3_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, ldw, N, K,
                     (__bf16*)packed, items);
  OCV_CHECK_LAUNCH("ocv_pack_split3_fwd");
  return 0;
}

extern "C" int ocv_linear_split3_fwd(const float* A, int lda, const void* w_packed, const float* bias, float* out, int ldo,
                                     int M, int N, int K, int act, ocv_stream_t stream) {
  OCV_CHECK_ARG(A && w_packed && out, "ocv_linear_split3_fwd: null pointer");
  OCV_CHECK_ARG(M >= 0 && N >= 1 && K >= 8 && K % 8 == 0 && lda >= K && ldo >= N && lda % 4 == 0,
                "ocv_linear_split3_fwd: bad sizes (K and lda must be multiples of 8 / 4; M=%d N=%d K=%d)", M, N, K);
  OCV_CHECK_ARG(act >= 0 && act <= 2, "ocv_linear_split3_fwd: unknown activation %d", act);
  OCV_CHECK_ARG(ocv_aligned16(A) && ocv_aligned16(w_packed), "ocv_linear_split3_fwd: A / w_packed must be 16-byte aligned");
  if (M == 0) return 0;
  L3Args a{A, lda, (const __bf16*)w_packed, bias, out, ldo, M, N, K, nullptr, 0, nullptr, nullptr, 0.f, nullptr};
  hipStream_t st = (hipStream_t)stream;
  const int ntl = (N + 31) / 32;
  if (ntl % 12 == 0 || ntl > 8) return launch_lin3<3>(a, act, st);       // 384 columns per workgroup (packed QKV)
  if (ntl > 4) return launch_lin3<2>(a, act, st);
  return launch_lin3<1>(a, act, st);
}

extern "C" int ocv_linear_residual_layernorm_split3_fwd(const float* A, int lda, const void* w_packed, const float* bias,
                                                        const float* residual, int ldres, const float* gamma,
                                                        const float* beta, float eps, const uint8_t* zero_row_mask,
                                                        float* out, int ldo, int M, int N, int K, ocv_stream_t stream) {
  OCV_CHECK_ARG(A && w_packed && residual && gamma && beta && out, "ocv_linear_residual_layernorm_split3_fwd: null pointer");
  OCV_CHECK_ARG(N == E128, "ocv_linear_residual_layernorm_split3_fwd: N must be %d (got %d)", E128, N);
  OCV_CHECK_ARG(M >= 0 && K >= 8 && K % 8 == 0 && lda >= K && lda % 4 == 0 && ldo >= N && ldres >= N,
                "ocv_linear_residual_layernorm_split3_fwd: bad sizes");
  OCV_CHECK_ARG(ocv_aligned16(A) && ocv_aligned16(w_packed), "ocv_linear_residual_layernorm_split3_fwd: A / w_packed must be 16-byte aligned");
  if (M == 0) return 0;
  L3Args a{A, lda, (const __bf16*)w_packed, bias, out, ldo, M, N, K, residual, ldres, gamma, beta, eps, zero_row_mask};
  hipLaunchKernelGGL((lin3_kernel<1, L3_RES_LN>), dim3(ocv_cdiv(M, TM), 1), dim3(256), 0, (hipStream_t)stream, a);
  OCV_CHECK_LAUNCH("ocv_linear_residual_layernorm_split3_fwd");
  return 0;
}

extern "C" int ocv_layer_tail_split3_fwd(const float* ctx, const float* x, const ocv_encoder_layer_params* p_caller,
                                         const void* next_in_proj_p3, const float* next_in_proj_b, float eps,
                                         const uint8_t* zero_row_mask, float* out, float* qkv_next, int M, int E, int FF,
                                         ocv_stream_t stream) {
  OCV_CHECK_ARG(ctx && x && p_caller && out, "ocv_layer_tail_split3_fwd: null pointer");
  ocv_encoder_layer_params pv;
  OCV_CHECK_ARG(ocv_layer_params_view(p_caller, 0, &pv), "ocv_layer_tail_split3_fwd: params->struct_size (%zu) is not a valid ocv_encoder_layer_params size", p_caller->struct_size);
  const ocv_encoder_layer_params* p = &pv;
  OCV_CHECK_ARG(p->out_proj_p3 && p->linear1_p3 && p->linear2_p3, "ocv_layer_tail_split3_fwd: needs the packed split3 weights");
  OCV_CHECK_ARG(E == E128 && FF >= KC && FF % KC == 0, "ocv_layer_tail_split3_fwd: needs E = %d and FF a multiple of %d (got %d, %d)", E128, KC, E, FF);
  OCV_CHECK_ARG((next_in_proj_p3 == nullptr) == (qkv_next == nullptr) && (next_in_proj_p3 == nullptr || next_in_proj_b != nullptr),
                "ocv_layer_tail_split3_fwd: next_in_proj_p3 / next_in_proj_b / qkv_next go together");
  OCV_CHECK_ARG(M >= 0 && ocv_aligned16(ctx) && ocv_aligned16(x) && ocv_aligned16(out) && ocv_aligned16(qkv_next),
                "ocv_layer_tail_split3_fwd: bad M / alignment");
  if (M == 0) return 0;
  TailArgs a{ctx, x, (const __bf16*)p->out_proj_p3, (const __bf16*)p->linear1_p3, (const __bf16*)p->linear2_p3,
             (const __bf16*)next_in_proj_p3, p->out_proj_b, p->norm1_w, p->norm1_b, p->linear1_b, p->linear2_b, p->norm2_w,
             p->norm2_b, next_in_proj_b, eps, zero_row_mask, out, qkv_next, M, FF};
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)layer_tail3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  const size_t lds = (size_t)2 * RT * 3 * PLANE * sizeof(__bf16) + (size_t)TR * (E128 + 1) * sizeof(float);
  hipLaunchKernelGGL(layer_tail3_kernel, dim3(ocv_cdiv(M, TR)), dim3(256), lds, (hipStream_t)stream, a);
  OCV_CHECK_LAUNCH("ocv_layer_tail_split3_fwd");
  return 0;
}

extern "C" size_t ocv_ffn_split3_workspace_bytes(int M, int FF) {
  if (M < 1 || FF < KC || FF % KC != 0) return 0;
  return (size_t)ocv_ffn_split_count(M, FF) * M * E128 * sizeof(float);
}

extern "C" int ocv_ffn_residual_layernorm_split3_fwd(const float* x, const void* w1_packed, const float* b1,
                                                     const void* w2_packed, const float* b2, const float* gamma,
                                                     const float* beta, float eps, const uint8_t* zero_row_mask, float* out,
                                                     int M, int E, int FF, void* workspace, size_t workspace_bytes,
                                                     ocv_stream_t stream) {
  OCV_CHECK_ARG(x && w1_packed && b1 && w2_packed && b2 && gamma && beta && out, "ocv_ffn_residual_layernorm_split3_fwd: null pointer");
  OCV_CHECK_ARG(E == E128 && FF >= KC && FF % KC == 0, "ocv_ffn_residual_layernorm_split3_fwd: needs E = %d and FF a multiple of %d (got %d, %d)", E128, KC, E, FF);
  OCV_CHECK_ARG(M >= 0 && ocv_aligned16(x) && ocv_aligned16(w1_packed) && ocv_aligned16(w2_packed), "ocv_ffn_residual_layernorm_split3_fwd: bad M / alignment");
  if (M == 0) return 0;
  int ns = ocv_ffn_split_count(M, FF);
  if (ns > 1 && (workspace == nullptr || workspace_bytes < (size_t)ns * M * E128 * sizeof(float))) ns = 1;
  F3Args a{x, (const __bf16*)w1_packed, (const __bf16*)w2_packed, b1, b2, gamma, beta, eps, zero_row_mask, out, M, FF,
           (float*)workspace, ns};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ffn3_kernel, dim3(ocv_cdiv(M, TM), ns), dim3(256), 0, st, a);
  OCV_CHECK_LAUNCH("ocv_ffn_residual_layernorm_split3_fwd");
  if (ns > 1) {
    hipLaunchKernelGGL(ffn3_finish_kernel, dim3(ocv_cdiv(M, TM)), dim3(256), 0, st, a);
    OCV_CHECK_LAUNCH("ocv_ffn_residual_layernorm_split3_fwd(finish)");
  }
  return 0;
}
