// Linear layers (+ fused residual/LayerNorm epilogue) and stand-alone LayerNorm
// for gfx950.  fp32 in, fp32 accumulate, v_mfma_f32_32x32x2_f32.
//
// Tiling: one workgroup = 4 wavefronts = a 32 (rows) x 128 (columns) output
// tile; wave w owns columns [32w, 32w+32).  K is walked in steps of 32 through
// LDS (A tile 32x32, W tile 128x32, rows padded to 33 floats so that the
// per-lane ds_read_b32 of an MFMA operand -- 32 consecutive rows, one column --
// hits 32 distinct banks).  fp32 MFMA issues once per 64 cycles per SIMD, so
// two 4-byte LDS reads per MFMA are far from the LDS limit; the kernel is
// bound by the matrix pipe (157 TFLOP/s chip peak) or, for the tiny object-side
// layers, by launch latency.
#include <stdlib.h>

#include "common.hpp"
#include "../../include/objcavit_hip.h"

namespace {

constexpr int BM = 32, BN = 128, BK = 32, LDT = BK + 1;

enum { EPI_NONE = 0, EPI_RELU = 1, EPI_LEAKY = 2, EPI_RES_LN = 3 };

struct LinearArgs {
  const float* A; int lda; long sA;
  const float* W; int ldw; long sW;
  const float* bias;
  float* out; int ldo; long sO;
  int M, N, K;
  // EPI_RES_LN only
  const float* res; int ldres;
  const float* gamma; const float* beta; float eps;
  const uint8_t* zero_mask;
};

template <int EPI, bool W_KN, bool VEC>
__global__ __launch_bounds__(256) void linear_kernel(LinearArgs p) {
  __shared__ float As[BM][LDT];
  __shared__ float Ws[BN][LDT];
  __shared__ float Cs[EPI == EPI_RES_LN ? BM : 1][EPI == EPI_RES_LN ? BN + 1 : 1];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const float* __restrict__ A = p.A + (long)blockIdx.z * p.sA;
  const float* __restrict__ W = p.W + (long)blockIdx.z * p.sW;
  float* __restrict__ out = p.out + (long)blockIdx.z * p.sO;
  const int M = p.M, N = p.N, K = p.K;

  f32x16 acc = {0};

  for (int k0 = 0; k0 < K; k0 += BK) {
    // ---- A tile: thread -> (row tid/8, 4 consecutive k)
    {
      const int r = tid >> 3, kk = (tid & 7) * 4;
      const int gm = m0 + r, gk = k0 + kk;
      float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
      if (gm < M) {
        const float* src = A + (long)gm * p.lda + gk;
        if (VEC && gk + 3 < K) {
          float4 t = ld4(src);
          v0 = t.x; v1 = t.y; v2 = t.z; v3 = t.w;
        } else {
          if (gk + 0 < K) v0 = src[0];
          if (gk + 1 < K) v1 = src[1];
          if (gk + 2 < K) v2 = src[2];
          if (gk + 3 < K) v3 = src[3];
        }
      }
      As[r][kk + 0] = v0; As[r][kk + 1] = v1; As[r][kk + 2] = v2; As[r][kk + 3] = v3;
    }
    // ---- W tile (128 x 32)
    if (!W_KN) {
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int r = pass * 32 + (tid >> 3), kk = (tid & 7) * 4;
        const int gn = n0 + r, gk = k0 + kk;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
        if (gn < N) {
          const float* src = W + (long)gn * p.ldw + gk;
          if (VEC && gk + 3 < K) {
            float4 t = ld4(src);
            v0 = t.x; v1 = t.y; v2 = t.z; v3 = t.w;
          } else {
            if (gk + 0 < K) v0 = src[0];
            if (gk + 1 < K) v1 = src[1];
            if (gk + 2 < K) v2 = src[2];
            if (gk + 3 < K) v3 = src[3];
          }
        }
        Ws[r][kk + 0] = v0; Ws[r][kk + 1] = v1; Ws[r][kk + 2] = v2; Ws[r][kk + 3] = v3;
      }
    } else {
      // W stored [K, N]: thread -> (k = pass*8 + tid/32, 4 consecutive n)
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int kk = pass * 8 + (tid >> 5), nn = (tid & 31) * 4;
        const int gk = k0 + kk, gn = n0 + nn;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
        if (gk < K) {
          const float* src = W + (long)gk * p.ldw + gn;
          if (VEC && gn + 3 < N) {
            float4 t = ld4(src);
            v0 = t.x; v1 = t.y; v2 = t.z; v3 = t.w;
          } else {
            if (gn + 0 < N) v0 = src[0];
            if (gn + 1 < N) v1 = src[1];
            if (gn + 2 < N) v2 = src[2];
            if (gn + 3 < N) v3 = src[3];
          }
        }
        Ws[nn + 0][kk] = v0; Ws[nn + 1][kk] = v1; Ws[nn + 2][kk] = v2; Ws[nn + 3][kk] = v3;
      }
    }
    __syncthreads();
    const float* arow = &As[l31][hh];
    const float* wrow = &Ws[wave * 32 + l31][hh];
#pragma unroll
    for (int s = 0; s < BK / 2; ++s) acc = mfma_32x32x2(arow[2 * s], wrow[2 * s], acc);
    __syncthreads();
  }

  const int n = n0 + wave * 32 + l31;
  const float bn = (p.bias != nullptr && n < N) ? p.bias[n] : 0.f;

  if (EPI != EPI_RES_LN) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + acc_row(r, hh);
      float v = acc[r] + bn;
      if (EPI == EPI_RELU) v = fmaxf(v, 0.f);
      if (EPI == EPI_LEAKY) v = v > 0.f ? v : 0.01f * v;
      if (m < M && n < N) out[(long)m * p.ldo + n] = v;
    }
  } else {
    // residual add, then LayerNorm over the 128 columns held by this workgroup
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = acc_row(r, hh), m = m0 + row;
      float v = acc[r] + bn;
      if (m < M) v += p.res[(long)m * p.ldres + n];
      Cs[row][wave * 32 + l31] = v;
    }
    __syncthreads();
    const float g0 = p.gamma[lane], g1 = p.gamma[lane + 64];
    const float b0 = p.beta[lane], b1 = p.beta[lane + 64];
#pragma unroll
    for (int i = 0; i < BM / 4; ++i) {
      const int row = wave * (BM / 4) + i, m = m0 + row;
      const float x0 = Cs[row][lane], x1 = Cs[row][lane + 64];
      const float mean = wave_sum(x0 + x1) * (1.0f / BN);
      const float d0 = x0 - mean, d1 = x1 - mean;
      const float var = wave_sum(d0 * d0 + d1 * d1) * (1.0f / BN);
      const float rstd = 1.0f / sqrtf(var + p.eps);
      if (m < M) {
        const bool z = p.zero_mask != nullptr && p.zero_mask[m] != 0;
        out[(long)m * p.ldo + lane] = z ? 0.f : d0 * rstd * g0 + b0;
        out[(long)m * p.ldo + lane + 64] = z ? 0.f : d1 * rstd * g1 + b1;
      }
    }
  }
}

// one wavefront per row
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        float* __restrict__ out, int rows, int E) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (long)row * E;
  const float* rr = res ? res + (long)row * E : nullptr;
  float s = 0.f;
  for (int c = lane; c < E; c += 64) s += xr[c] + (rr ? rr[c] : 0.f);
  const float mean = wave_sum(s) / (float)E;
  float q = 0.f;
  for (int c = lane; c < E; c += 64) {
    const float d = xr[c] + (rr ? rr[c] : 0.f) - mean;
    q += d * d;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)E + eps);
  for (int c = lane; c < E; c += 64) {
    const float d = xr[c] + (rr ? rr[c] : 0.f) - mean;
    out[(long)row * E + c] = d * rstd * gamma[c] + beta[c];
  }
}

// ---------------------------------------------------------------------------
// "streamed-weight" linear: the layer sizes of the transformer stack (K = 128,
// 512, 1024; weights in nn.Linear [N, K] layout, everything 16-byte aligned).
//
// In out = A W^T every weight element is consumed by exactly ONE wavefront (the
// one that owns its output column), so the weight operand is loaded straight
// from L2 into VGPRs as float4 -- no LDS round trip; at 64 cycles per fp32 MFMA
// one 16-byte load feeds 4 issues.  Only the activation rows, which all four
// wavefronts share, are staged in LDS (32 x 128-float chunks, rows padded to
// 132 floats so the ds_read_b128 of the A operand is conflict-free).
// The K order inside a chunk is permuted so that both operands are 16-byte
// vectors: MFMA step (t, e), k-slot hh  <->  k = 8 t + 4 hh + e.
// ---------------------------------------------------------------------------
constexpr int FM = 32, FKC = 128, FLD = FKC + 4;

__device__ __forceinline__ void stage_rows(float (*Xs)[FLD], const float* __restrict__ src, int ld, int m0, int M,
                                           int k0, int kc, int tid) {
  const int row = tid >> 3;
  const bool ok = m0 + row < M;
#pragma unroll
  for (int pass = 0; pass < FKC / 32; ++pass) {
    const int c4 = (tid & 7) * 4 + 32 * pass;
    if (c4 < kc) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) t = ld4(src + (long)(m0 + row) * ld + k0 + c4);
      *reinterpret_cast<float4*>(&Xs[row][c4]) = t;
    }
  }
}

// 64 MFMAs (or fewer for a short chunk): acc += Xs[32 x kc] . Wregs^T
__device__ __forceinline__ f32x16 chunk_mfma(f32x16 acc, const float (*Xs)[FLD], const float4 (&w)[FKC / 8], int kc,
                                             int l31, int hh) {
#pragma unroll
  for (int t = 0; t < FKC / 8; ++t) {
    if (8 * t < kc) {
      const float4 a = *reinterpret_cast<const float4*>(&Xs[l31][8 * t + 4 * hh]);
      acc = mfma_32x32x2(a.x, w[t].x, acc);
      acc = mfma_32x32x2(a.y, w[t].y, acc);
      acc = mfma_32x32x2(a.z, w[t].z, acc);
      acc = mfma_32x32x2(a.w, w[t].w, acc);
    }
  }
  return acc;
}

__device__ __forceinline__ void load_wregs(float4 (&w)[FKC / 8], const float* __restrict__ wrow, int kc) {
#pragma unroll
  for (int t = 0; t < FKC / 8; ++t)
    w[t] = (8 * t < kc) ? ld4(wrow + 8 * t) : make_float4(0.f, 0.f, 0.f, 0.f);
}

// residual + LayerNorm over the 128 columns of a 32-row tile held in Cs (one wavefront per 8 rows)
__device__ __forceinline__ void ln_rows(const float (*Cs)[BN + 1], const float* gamma, const float* beta, float eps,
                                        const uint8_t* zero_mask, float* out, int ldo, int m0, int M, int lane, int wave) {
  const float g0 = gamma[lane], g1 = gamma[lane + 64];
  const float b0 = beta[lane], b1 = beta[lane + 64];
#pragma unroll
  for (int i = 0; i < FM / 4; ++i) {
    const int row = wave * (FM / 4) + i, m = m0 + row;
    const float x0 = Cs[row][lane], x1 = Cs[row][lane + 64];
    const float mean = wave_sum(x0 + x1) * (1.0f / BN);
    const float d0 = x0 - mean, d1 = x1 - mean;
    const float var = wave_sum(d0 * d0 + d1 * d1) * (1.0f / BN);
    const float rstd = 1.0f / sqrtf(var + eps);
    if (m < M) {
      const bool z = zero_mask != nullptr && zero_mask[m] != 0;
      out[(long)m * ldo + lane] = z ? 0.f : d0 * rstd * g0 + b0;
      out[(long)m * ldo + lane + 64] = z ? 0.f : d1 * rstd * g1 + b1;
    }
  }
}

template <int EPI>
__global__ __launch_bounds__(256) void linear_stream_kernel(LinearArgs p) {
  __shared__ __attribute__((aligned(16))) float Xs[FM][FLD];
  __shared__ float Cs[EPI == EPI_RES_LN ? FM : 1][BN + 1];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int m0 = blockIdx.x * FM, n0 = blockIdx.y * BN;
  const float* __restrict__ A = p.A + (long)blockIdx.z * p.sA;
  const float* __restrict__ W = p.W + (long)blockIdx.z * p.sW;
  float* __restrict__ out = p.out + (long)blockIdx.z * p.sO;
  const int M = p.M, N = p.N, K = p.K;
  const int n = n0 + wave * 32 + l31;
  const float* wrow = W + (long)(n < N ? n : N - 1) * p.ldw + 4 * hh;

  f32x16 acc = {0};
  for (int k0 = 0; k0 < K; k0 += FKC) {
    const int kc = min(FKC, K - k0);
    float4 w[FKC / 8];
    load_wregs(w, wrow + k0, kc);
    __syncthreads();
    stage_rows(Xs, A, p.lda, m0, M, k0, kc, tid);
    __syncthreads();
    acc = chunk_mfma(acc, Xs, w, kc, l31, hh);
  }

  const float bn = (p.bias != nullptr && n < N) ? p.bias[n] : 0.f;
  if (EPI != EPI_RES_LN) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + acc_row(r, hh);
      float v = acc[r] + bn;
      if (EPI == EPI_RELU) v = fmaxf(v, 0.f);
      if (EPI == EPI_LEAKY) v = v > 0.f ? v : 0.01f * v;
      if (m < M && n < N) out[(long)m * p.ldo + n] = v;
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = acc_row(r, hh), m = m0 + row;
      float v = acc[r] + bn;
      if (m < M) v += p.res[(long)m * p.ldres + n];
      Cs[row][wave * 32 + l31] = v;
    }
    __syncthreads();
    ln_rows(Cs, p.gamma, p.beta, p.eps, p.zero_mask, out, p.ldo, m0, M, lane, wave);
  }
}

// ---------------------------------------------------------------------------
// Fused feed-forward block of a post-norm transformer layer:
//   out = LayerNorm( x + W2 relu(W1 x + b1) + b2 )        x: [M, 128], FF hidden units
// One workgroup = 32 rows.  The hidden activations never leave the chip: they
// are produced 128 units at a time into LDS (phase 1, 64 MFMAs per wavefront)
// and immediately contracted with the matching 128-column slice of W2 into the
// output accumulator (phase 2, 64 MFMAs).  W1 / W2 stream from L2 into VGPRs
// (see above); the W2 slice is fetched while phase 1 multiplies.
// Saves the [M, FF] round trip through HBM (19.7 MB at bs = 16) and two launches.
// ---------------------------------------------------------------------------
struct FFNArgs {
  const float *x, *w1, *b1, *w2, *b2, *gamma, *beta;
  float eps;
  const uint8_t* zero_mask;
  float* out;
  int M, FF;
  float* part;           // nsplit > 1: raw partial outputs [nsplit][M][128] (ffn_finish_kernel adds them)
  int nsplit;            // workgroups per row tile, each over FF / nsplit hidden units
};

__global__ __launch_bounds__(256) void ffn_fused_kernel(FFNArgs p) {
  __shared__ __attribute__((aligned(16))) float Xs[FM][FLD];
  __shared__ __attribute__((aligned(16))) float Hs[FM][FLD];
  __shared__ float Cs[FM][BN + 1];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int m0 = blockIdx.x * FM;
  const int col = wave * 32 + l31;                 // hidden unit within a chunk (phase 1) / output column (phase 2)

  stage_rows(Xs, p.x, BN, m0, p.M, 0, BN, tid);
  f32x16 acc = {0};
  float4 w1r[FKC / 8], w2r[FKC / 8];
  // A row tile's hidden units may be shared out over nsplit workgroups (blockIdx.y): with 150 row tiles (4800 image
  // tokens) or 16 (512 object tokens) on 256 CUs the chain of FF / 128 dependent chunk rounds per workgroup, not the
  // matrix pipe, set the launch time.
  const int nchunk_all = p.FF / FKC;
  const int cbeg = nchunk_all * (int)blockIdx.y / p.nsplit, nchunk = nchunk_all * ((int)blockIdx.y + 1) / p.nsplit;
  load_wregs(w1r, p.w1 + (long)(cbeg * FKC + col) * BN + 4 * hh, BN);
  __syncthreads();

  for (int c = cbeg; c < nchunk; ++c) {
    load_wregs(w2r, p.w2 + (long)col * p.FF + c * FKC + 4 * hh, FKC);        // in flight during phase 1
    f32x16 h = {0};
    h = chunk_mfma(h, Xs, w1r, BN, l31, hh);
    const float b1 = p.b1[c * FKC + col];
#pragma unroll
    for (int r = 0; r < 16; ++r) Hs[acc_row(r, hh)][col] = fmaxf(h[r] + b1, 0.f);
    if (c + 1 < nchunk) load_wregs(w1r, p.w1 + (long)((c + 1) * FKC + col) * BN + 4 * hh, BN);   // next chunk's W1
    __syncthreads();
    acc = chunk_mfma(acc, Hs, w2r, FKC, l31, hh);
    __syncthreads();
  }

  if (p.nsplit > 1) {
    float* dst = p.part + ((long)blockIdx.y * p.M + m0) * BN + col;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = acc_row(r, hh);
      if (m0 + row < p.M) dst[(long)row * BN] = acc[r];
    }
    return;
  }
  const float b2 = p.b2[col];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = acc_row(r, hh);
    Cs[row][col] = acc[r] + b2 + Xs[row][col];      // residual = the layer input rows (zero beyond M)
  }
  __syncthreads();
  ln_rows(Cs, p.gamma, p.beta, p.eps, p.zero_mask, p.out, BN, m0, p.M, lane, wave);
}

// second pass of a split FFN: out = LayerNorm(x + sum_s part[s] + b2), partials added in split order
__global__ __launch_bounds__(256) void ffn_finish_kernel(FFNArgs p) {
  __shared__ float Cs[FM][BN + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * FM;
  for (int i = tid; i < FM * BN; i += 256) {
    const int row = i / BN, col = i % BN;
    const long m = m0 + row;
    float v = 0.f;
    if (m < p.M) {
      v = p.part[m * BN + col];
      for (int sidx = 1; sidx < p.nsplit; ++sidx) v += p.part[((long)sidx * p.M + m) * BN + col];
      v += p.b2[col] + p.x[m * BN + col];
    }
    Cs[row][col] = v;
  }
  __syncthreads();
  ln_rows(Cs, p.gamma, p.beta, p.eps, p.zero_mask, p.out, BN, m0, p.M, lane, wave);
}

// ---------------------------------------------------------------------------
// Fused multi-head attention for FEW keys (at most 32 live keys per batch row: the image <- object cross-attention,
// where only the first N_max object tokens are unmasked; E = 128, 4 heads of 32):
//   out = ( softmax_keys( (X_q Wq^T + bq)(X_k Wk^T + bk)^T / sqrt(32) + mask ) (X_v Wv^T + bv) ) Wo^T + bo
// ONE launch instead of five (three projections, attention, output projection): a workgroup owns 32 query rows of one
// batch row, stages them and the <= 32 key / value source rows in LDS, and wavefront h does head h end to end --
// its 32 columns of Q, K and V (weights streamed from L2 as in linear_stream_kernel), the 32 x 32 score tile in the
// K Q^T orientation (softmax statistics in registers + one xor-32 shuffle, probabilities consumed in place as the MFMA
// B operand, as in csrc/attention.hip), its 32 context columns -- then, behind one barrier, 32 columns of the
// output projection.  Recomputing K and V per 32-query tile costs 128 MFMAs of the 288 per wavefront; in exchange
// nothing but the inputs and the output touches memory and four launch latencies disappear (46 -> ~12 us at bs = 16).
// ---------------------------------------------------------------------------
struct XAArgs {
  const float *q_src, *k_src, *v_src;
  const uint8_t* mask;
  const float *in_w, *in_b, *out_w, *out_b;
  float* out;
  int Sq, Sk, Se, mask_ld;
  float scale;
};

__global__ __launch_bounds__(256) void cross_attn_fused_kernel(XAArgs p) {
  extern __shared__ __attribute__((aligned(16))) float xa_lds[];
  float (*Xq)[FLD] = reinterpret_cast<float (*)[FLD]>(xa_lds);                       // query rows, later context rows
  float (*Xk)[FLD] = Xq + FM;
  float (*Xv)[FLD] = Xk + FM;
  float (*Qs)[BN + 1] = reinterpret_cast<float (*)[BN + 1]>(xa_lds + 3 * FM * FLD);  // projected rows, all heads
  float (*Ks)[BN + 1] = Qs + FM;
  float (*Vs)[BN + 1] = Ks + FM;
  float* Ms = xa_lds + 3 * FM * FLD + 3 * FM * (BN + 1);                             // additive key mask (0 / -inf)

  const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const long b = blockIdx.y;
  const int q0 = blockIdx.x * FM;
  const int col = h * 32 + l31;
  constexpr float NEG_INF = -__builtin_inff();

  stage_rows(Xq, p.q_src + b * p.Sq * BN, BN, q0, p.Sq, 0, BN, tid);
  stage_rows(Xk, p.k_src + b * p.Sk * BN, BN, 0, p.Se, 0, BN, tid);
  stage_rows(Xv, p.v_src + b * p.Sk * BN, BN, 0, p.Se, 0, BN, tid);
  if (tid < FM) Ms[tid] = (tid >= p.Se || (p.mask != nullptr && p.mask[b * p.mask_ld + tid] != 0)) ? NEG_INF : 0.f;
  float4 wa[FKC / 8], wb[FKC / 8];
  load_wregs(wa, p.in_w + (long)col * BN + 4 * hh, BN);                              // Wq rows of this head
  load_wregs(wb, p.in_w + (long)(BN + col) * BN + 4 * hh, BN);                       // Wk
  __syncthreads();

  {
    f32x16 acc = {0};
    acc = chunk_mfma(acc, Xq, wa, BN, l31, hh);
    const float bq = p.in_b[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) Qs[acc_row(r, hh)][col] = (acc[r] + bq) * p.scale;
    load_wregs(wa, p.in_w + (long)(2 * BN + col) * BN + 4 * hh, BN);                 // Wv, in flight during the K projection
    f32x16 ak = {0};
    ak = chunk_mfma(ak, Xk, wb, BN, l31, hh);
    const float bk = p.in_b[BN + col];
#pragma unroll
    for (int r = 0; r < 16; ++r) Ks[acc_row(r, hh)][col] = ak[r] + bk;
    load_wregs(wb, p.out_w + (long)col * BN + 4 * hh, BN);                           // Wo, in flight from here on
    f32x16 av = {0};
    av = chunk_mfma(av, Xv, wa, BN, l31, hh);
    const float bvv = p.in_b[2 * BN + col];
#pragma unroll
    for (int r = 0; r < 16; ++r) Vs[acc_row(r, hh)][col] = av[r] + bvv;
  }
  __syncthreads();                                   // Xq is free (every wavefront has read it); Q / K / V columns visible

  // ---- head h: scores^T (rows = keys, columns = queries), softmax over the key rows, context^T
  f32x16 s = {0};
#pragma unroll
  for (int st = 0; st < 16; ++st) s = mfma_32x32x2(Ks[l31][h * 32 + 2 * st + hh], Qs[l31][h * 32 + 2 * st + hh], s);
  float tmax = NEG_INF;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    s[r] += Ms[acc_row(r, hh)];
    tmax = fmaxf(tmax, s[r]);
  }
  tmax = xor32_max(tmax);
  const bool none = tmax == NEG_INF;                 // every key masked for this batch row: 0 / 0 = NaN, as torch
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
  for (int r = 0; r < 16; ++r) o = mfma_32x32x2(Vs[acc_row(r, hh)][h * 32 + l31], s[r], o);
  // o: register r = context[query l31][d = acc_row(r, hh)] of head h -> context rows (the old query-row buffer)
#pragma unroll
  for (int r = 0; r < 16; ++r) Xq[l31][h * 32 + acc_row(r, hh)] = o[r] * inv;
  __syncthreads();

  // ---- output projection, 32 columns per wavefront
  f32x16 acc = {0};
  acc = chunk_mfma(acc, Xq, wb, BN, l31, hh);
  const float bo = p.out_b[col];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int q = q0 + acc_row(r, hh);
    if (q < p.Sq) p.out[(b * p.Sq + q) * BN + col] = acc[r] + bo;
  }
}

template <int EPI>
int launch_linear(const LinearArgs& a, int batch, bool w_kn, hipStream_t st) {
  dim3 grid(ocv_cdiv(a.M, BM), ocv_cdiv(a.N, BN), batch), block(256);
  const bool vecA = (a.lda % 4 == 0) && (a.sA % 4 == 0) && ocv_aligned16(a.A);
  const bool vecW = (a.ldw % 4 == 0) && (a.sW % 4 == 0) && ocv_aligned16(a.W);
  const bool vec = vecA && vecW && (w_kn || a.K % 4 == 0);
  if (!w_kn && vec && a.K % 8 == 0) {
    hipLaunchKernelGGL((linear_stream_kernel<EPI>), grid, block, 0, st, a);
    OCV_CHECK_LAUNCH("ocv_linear(stream)");
    return 0;
  }
  if (w_kn) {
    if (vec) hipLaunchKernelGGL((linear_kernel<EPI, true, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((linear_kernel<EPI, true, false>), grid, block, 0, st, a);
  } else {
    if (vec) hipLaunchKernelGGL((linear_kernel<EPI, false, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((linear_kernel<EPI, false, false>), grid, block, 0, st, a);
  }
  OCV_CHECK_LAUNCH("ocv_linear");
  return 0;
}

}  // namespace

extern "C" int ocv_linear_fwd(const float* A, int lda, long strideA, const float* W, int ldw, long strideW, int w_kn,
                              const float* bias, float* out, int ldo, long strideO, int batch, int M, int N, int K,
                              int act, ocv_stream_t stream) {
  OCV_CHECK_ARG(A && W && out, "ocv_linear_fwd: null pointer");
  OCV_CHECK_ARG(batch >= 1 && M >= 0 && N >= 1 && K >= 1, "ocv_linear_fwd: bad sizes batch=%d M=%d N=%d K=%d", batch, M, N, K);
  OCV_CHECK_ARG(lda >= K && ldo >= N && ldw >= (w_kn ? N : K), "ocv_linear_fwd: leading dimension too small");
  OCV_CHECK_ARG(act >= 0 && act <= 2, "ocv_linear_fwd: unknown activation %d", act);
  if (M == 0) return 0;
  LinearArgs a{A, lda, strideA, W, ldw, strideW, bias, out, ldo, strideO, M, N, K, nullptr, 0, nullptr, nullptr, 0.f, nullptr};
  hipStream_t st = (hipStream_t)stream;
  switch (act) {
    case OCV_ACT_RELU: return launch_linear<EPI_RELU>(a, batch, w_kn != 0, st);
    case OCV_ACT_LEAKY_RELU: return launch_linear<EPI_LEAKY>(a, batch, w_kn != 0, st);
    default: return launch_linear<EPI_NONE>(a, batch, w_kn != 0, st);
  }
}

extern "C" int ocv_linear_residual_layernorm_fwd(const float* A, int lda, const float* W, int ldw, const float* bias,
                                                 const float* residual, int ldres, const float* gamma,
                                                 const float* beta, float eps, const uint8_t* zero_row_mask,
                                                 float* out, int ldo, int M, int N, int K, ocv_stream_t stream) {
  OCV_CHECK_ARG(A && W && residual && gamma && beta && out, "ocv_linear_residual_layernorm_fwd: null pointer");
  OCV_CHECK_ARG(N == BN, "ocv_linear_residual_layernorm_fwd: N must be %d (got %d)", BN, N);
  OCV_CHECK_ARG(M >= 0 && K >= 1 && lda >= K && ldw >= K && ldo >= N && ldres >= N,
                "ocv_linear_residual_layernorm_fwd: bad sizes");
  if (M == 0) return 0;
  LinearArgs a{A, lda, 0, W, ldw, 0, bias, out, ldo, 0, M, N, K, residual, ldres, gamma, beta, eps, zero_row_mask};
  return launch_linear<EPI_RES_LN>(a, 1, false, (hipStream_t)stream);
}

extern "C" int ocv_layernorm_residual_fwd(const float* x, const float* residual, const float* gamma,
                                          const float* beta, float eps, float* out, int rows, int E,
                                          ocv_stream_t stream) {
  OCV_CHECK_ARG(x && gamma && beta && out, "ocv_layernorm_residual_fwd: null pointer");
  OCV_CHECK_ARG(rows >= 0 && E >= 1, "ocv_layernorm_residual_fwd: bad sizes");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(layernorm_kernel, dim3(ocv_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, residual, gamma,
                     beta, eps, out, rows, E);
  OCV_CHECK_LAUNCH("ocv_layernorm_residual_fwd");
  return 0;
}

extern "C" int ocv_ffn_residual_layernorm_fwd(const float* x, const float* w1, const float* b1, const float* w2,
                                              const float* b2, const float* gamma, const float* beta, float eps,
                                              const uint8_t* zero_row_mask, float* out, int M, int E, int FF,
                                              ocv_stream_t stream) {
  OCV_CHECK_ARG(x && w1 && b1 && w2 && b2 && gamma && beta && out, "ocv_ffn_residual_layernorm_fwd: null pointer");
  OCV_CHECK_ARG(E == BN && FF >= FKC && FF % FKC == 0, "ocv_ffn_residual_layernorm_fwd: needs E = %d and FF a multiple of %d (got %d, %d)", BN, FKC, E, FF);
  OCV_CHECK_ARG(M >= 0, "ocv_ffn_residual_layernorm_fwd: bad M");
  OCV_CHECK_ARG(ocv_aligned16(x) && ocv_aligned16(w1) && ocv_aligned16(w2), "ocv_ffn_residual_layernorm_fwd: x / w1 / w2 must be 16-byte aligned");
  if (M == 0) return 0;
  FFNArgs a{x, w1, b1, w2, b2, gamma, beta, eps, zero_row_mask, out, M, FF, nullptr, 1};
  hipLaunchKernelGGL(ffn_fused_kernel, dim3(ocv_cdiv(M, FM)), dim3(256), 0, (hipStream_t)stream, a);
  OCV_CHECK_LAUNCH("ocv_ffn_residual_layernorm_fwd");
  return 0;
}

// FFN with the hidden units of a row tile shared out over several workgroups (see ffn_fused_kernel); part: scratch of
// ocv_ffn_split_count(M, FF) * M * 128 floats.  Same contract as ocv_ffn_residual_layernorm_fwd otherwise.
int ocv_ffn_split_count(int M, int FF) {
  const int nchunk = FF / FKC, tiles = ocv_cdiv(M, FM);
  int ns = 1;
  while (2 * ns <= nchunk && nchunk % (2 * ns) == 0 && tiles * 2 * ns <= 256) ns *= 2;      // measured: 150 tiles x 2 lost 6 % (0.48 -> 0.51 ms for 4 layers), 16 tiles x 8 won 27 % (0.37 -> 0.27)
  return ns;
}

int ocv_ffn_split_launch(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                         const float* gamma, const float* beta, float eps, const uint8_t* zero_row_mask, float* out, int M,
                         int FF, float* part, int nsplit, hipStream_t st) {
  FFNArgs a{x, w1, b1, w2, b2, gamma, beta, eps, zero_row_mask, out, M, FF, part, nsplit};
  hipLaunchKernelGGL(ffn_fused_kernel, dim3(ocv_cdiv(M, FM), nsplit), dim3(256), 0, st, a);
  OCV_CHECK_LAUNCH("ocv_ffn_split_launch");
  hipLaunchKernelGGL(ffn_finish_kernel, dim3(ocv_cdiv(M, FM)), dim3(256), 0, st, a);
  OCV_CHECK_LAUNCH("ocv_ffn_split_launch(finish)");
  return 0;
}

// fused few-key attention (see cross_attn_fused_kernel); returns 1 when the shape is not covered (caller falls back)
int ocv_cross_attn_fused_launch(const float* q_src, const float* k_src, const float* v_src, const uint8_t* mask, int mask_ld,
                                const float* in_w, const float* in_b, const float* out_w, const float* out_b, float* out, int B,
                                int Sq, int Sk, int Se, int E, int H, hipStream_t st) {
  if (E != BN || H != 4 || Se < 1 || Se > FM || B > 65535) return 1;
  if (!(ocv_aligned16(q_src) && ocv_aligned16(k_src) && ocv_aligned16(v_src) && ocv_aligned16(in_w) && ocv_aligned16(out_w))) return 1;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)cross_attn_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  XAArgs a{q_src, k_src, v_src, mask, in_w, in_b, out_w, out_b, out, Sq, Sk, Se, mask_ld, 1.0f / sqrtf(32.0f)};
  const size_t lds = (size_t)(3 * FM * FLD + 3 * FM * (BN + 1) + FM) * sizeof(float);
  hipLaunchKernelGGL(cross_attn_fused_kernel, dim3(ocv_cdiv(Sq, FM), B), dim3(256), lds, st, a);
  OCV_CHECK_LAUNCH("ocv_mha_fwd(fused)");
  return 0;
}
