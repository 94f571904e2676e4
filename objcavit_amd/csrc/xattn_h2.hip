// Few-key multi-head attention on TWO-TERM HALF-PRECISION SPLITS -- the image <- object cross-attention of
// SelfAttnCrossAttn.forward (reference modules/ObjCAViT.py:192-201: nn.MultiheadAttention, E = 128, 4 heads of 32, at most
// 32 live keys per image behind a key-padding mask).
//
// Why another split.  csrc/token_split3.hip contracts fp32 operands as three bf16 terms (8 bits each): six MFMAs per
// product block for a 2^-24 product.  An fp16 term carries 11 bits, so TWO terms carry 22: x = hi + lo' 2^-11 with
// hi = fp16(x) and lo' = fp16((x - hi) 2^11) -- the residual is exact in fp32 and, scaled by 2^11, sits in the same binade
// as hi, so it never falls into fp16's subnormals whatever the magnitude of x.  A product block is then THREE MFMAs,
//     acc1 += hi_a hi_b          acc2 += hi_a lo'_b + lo'_a hi_b          result = acc1 + 2^-11 acc2,
// the dropped lo lo term and the two roundings of lo' each 2^-22 of the product: measured 3e-7 of max |result| on
// 128-long contractions, the same as an fp32 FMA chain (bf16 x 3: 1.5e-7, bf16 x 2: 5e-6; tests/test_hip_kernels.py).
// With both operands of the score and context products split the same way the whole tile runs on
// v_mfma_f32_32x32x16_f16: 24 + 6 + 6 + 24 = 60 MFMAs of 32 cycles per wavefront and 32 queries = 1.9 K matrix-pipe cycles
// against 5.1 K for the split3 form (96 bf16 MFMAs + 32 exact-fp32 ones of 64 cycles), two weight-fragment parts
// instead of three (128 KB per workgroup through the vector-memory path instead of 192 KB) and half the split arithmetic.
// (profiles/r03_cross_attention_roofline.txt holds the ablation of the split3 tile that led here.)
//
// Small magnitudes: the HIGH term of an operand below 6.1e-5 is an fp16 subnormal; v_mfma_f32_32x32x16_f16 honours subnormal
// inputs on gfx950 (tools/diag/mfma_f16_denorm.hip, measured), and whatever hi loses the scaled low term carries.
// Range: fp16 tops out at 65504.  An ACTIVATION beyond that (token, projected query / key / value) converts to hi = inf and
// the output row turns inf / NaN -- loud, not silently saturated; tokens, projected keys / values and softmax-weighted
// contexts of this model are O(1) .. O(100).  The weight packer saturates at +-65504 (a static matrix: checked once, not
// per element and launch).  OCV_TOKENS=split3 (csrc/token_split3.hip) is the route with fp32's range.
//
//   xattn_kv_h2_kernel    K and V projected ONCE per image, one workgroup per image and operand, written pre-split in the
//                         order the query tiles' lanes consume them: [B][2 operands][4 heads][64 lanes][2 steps][2 parts][8]
//                         fp16 = 32 KB per image.  K is projected transposed (weights = A operand), so lane (key, half)
//                         holds K[key][32 h + acc_row(r, half)], r = 0..15 = the A operands of its two score steps; V
//                         straight, so lane (d, half) holds V[acc_row(r, half)][32 h + d] = the A operands of its context.
//   xattn_main_h2_kernel  per 32-query tile (two tiles per workgroup on one load of the weight fragments in large
//                         launches): wavefront h projects head h of Q transposed -- the accumulator's register r of lane
//                         (query, half) is d = acc_row(r, half), exactly the k-slot order of the B operand of the score
//                         MFMAs once registers 8 t .. 8 t + 7 are packed as step t --, scores^T = K Q^T, softmax over the
//                         keys (rows: in-lane + one xor-32 shuffle), context^T = V^T P^T, context rows to LDS as two fp16
//                         planes, output projection transposed (a register quad = 16 contiguous bytes of an output row).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>

#include "common.hpp"
#include "../../include/objcavit_hip.h"

namespace {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

constexpr int TM = 32;               // queries per sub-tile
constexpr int E128 = 128;
constexpr int NS = E128 / 16;        // K steps of a 128-long contraction
constexpr int PROW = E128 + 8;       // fp16 per plane row (272 bytes: b128 reads of 32 rows are conflict-free)
constexpr int PLANE = TM * PROW;
constexpr float LO_UP = 2048.0f, LO_DOWN = 1.0f / 2048.0f;
constexpr float H_MAX = 65504.0f;

__device__ __forceinline__ void split_h2(float x, _Float16& hi, _Float16& lo) {
  hi = (_Float16)x;
  lo = (_Float16)((x - (float)hi) * LO_UP);
}

__device__ __forceinline__ void split8_h2(const float4 u, const float4 v, h16x8& h, h16x8& l) {
  const float f[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    _Float16 a, c;
    split_h2(f[i], a, c);
    h[i] = a;
    l[i] = c;
  }
}

// 256 threads stage rows [m0, m0 + 32) of a row-major [*, 128] fp32 matrix as two fp16 planes (rows >= M: zeros)
__device__ __forceinline__ void stage_rows_h2(_Float16* planes, const float* __restrict__ src, int m0, int M, int tid) {
  const int row = tid >> 3;
  const bool ok = m0 + row < M;
  const float* s = src + (long)(ok ? m0 + row : 0) * E128;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int o = (tid & 7) + 8 * i;                     // octet of the row
    float4 u = make_float4(0.f, 0.f, 0.f, 0.f), v = u;
    if (ok) {
      u = ld4(s + 8 * o);
      v = ld4(s + 8 * o + 4);
    }
    h16x8 h, l;
    split8_h2(u, v, h, l);
    _Float16* d = planes + row * PROW + 8 * o;
    *reinterpret_cast<h16x8*>(d) = h;
    *reinterpret_cast<h16x8*>(d + PLANE) = l;
  }
}

// one product block step: small terms into acc2 (carries 2^-11), the hi hi term into acc1
__device__ __forceinline__ void mfma3(const h16x8 ah, const h16x8 al, const h16x8 bh, const h16x8 bl, f32x16& acc1, f32x16& acc2) {
  acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc2, 0, 0, 0);
  acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc2, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc1, 0, 0, 0);
}

// the weight fragments of one 32-row tile over the whole 128-long contraction: 8 steps x 2 parts x 16 bytes per lane = 64 VGPRs
struct WFragH { h16x8 w[NS][2]; };

__device__ __forceinline__ void load_wh(WFragH& f, const _Float16* wp) {
#pragma unroll
  for (int s = 0; s < NS; ++s) {
#pragma unroll
    for (int p = 0; p < 2; ++p) f.w[s][p] = *reinterpret_cast<const h16x8*>(wp + (long)s * 1024 + p * 512);
  }
}

// D^T[n][m] = sum_k W[n][k] X[m][k]: weight fragments = A operand, plane rows = B operand.  FENCE: a scheduling barrier every
// two steps keeps the compiler from hoisting all sixteen LDS reads (64 registers) above the MFMAs (the 128-VGPR pre-pass).
template <bool FENCE = false>
__device__ __forceinline__ void project_t(const WFragH& f, const _Float16* planes, int l31, int hh, f32x16& a1, f32x16& a2) {
  const _Float16* pb = planes + l31 * PROW + 8 * hh;
#pragma unroll
  for (int st = 0; st < NS; ++st) {
    const h16x8 xh = *reinterpret_cast<const h16x8*>(pb + 16 * st);
    const h16x8 xl = *reinterpret_cast<const h16x8*>(pb + 16 * st + PLANE);
    mfma3(f.w[st][0], f.w[st][1], xh, xl, a1, a2);
    if (FENCE && (st & 1)) __builtin_amdgcn_sched_barrier(0);
  }
}

// registers 8 t .. 8 t + 7 of an accumulator-shaped value -> the two fp16 parts of k-step t of an MFMA operand
__device__ __forceinline__ void pack_steps(const f32x16 v, h16x8 (&hi)[2], h16x8 (&lo)[2]) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      _Float16 a, c;
      split_h2(v[8 * t + e], a, c);
      hi[t][e] = a;
      lo[t][e] = c;
    }
  }
}

// W [N][K] fp32 (row stride ldw) -> packed two-term fp16 fragments; one thread per (jt, s, lane)
__global__ __launch_bounds__(256) void pack_h2_kernel(const float* __restrict__ W, int ldw, int N, int K, _Float16* __restrict__ out,
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
  for (int e = 0; e < 8; ++e) {
    const float w = (n < N && k0 + e < K) ? W[(long)n * ldw + k0 + e] : 0.f;
    f[e] = fabsf(w) > H_MAX ? copysignf(H_MAX, w) : w;                 // saturate; a NaN fails the comparison and stays a NaN
  }
  h16x8 h, l;
  split8_h2(make_float4(f[0], f[1], f[2], f[3]), make_float4(f[4], f[5], f[6], f[7]), h, l);
  _Float16* d = out + (js * 2) * 512 + lane * 8;
  *reinterpret_cast<h16x8*>(d) = h;
  *reinterpret_cast<h16x8*>(d + 512) = l;
}

struct XKVArgs {
  const float *k_src, *v_src;          // [B][Sk][128]
  const _Float16* in_h2;               // packed in_proj_weight [384][128]
  const float* in_b;                   // [384]
  _Float16* kv;                        // [B][2][4][64][2][2][8]
  int Sk, Se;
};

// the two halves of stage_rows_h2: global loads into registers (issued early), split + LDS stores (after the previous
// image's MFMAs have read the planes)
struct RowRegs { float4 u[2], v[2]; };

__device__ __forceinline__ void rows_load(RowRegs& r, const float* __restrict__ src, int M, int tid) {
  const int row = tid >> 3;
  const bool ok = row < M;
  const float* s = src + (long)(ok ? row : 0) * E128;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int o = (tid & 7) + 8 * i;
    r.u[i] = r.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) {
      r.u[i] = ld4(s + 8 * o);
      r.v[i] = ld4(s + 8 * o + 4);
    }
  }
}

__device__ __forceinline__ void rows_store(_Float16* planes, const RowRegs& r, int tid) {
  const int row = tid >> 3;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int o = (tid & 7) + 8 * i;
    h16x8 h, l;
    split8_h2(r.u[i], r.v[i], h, l);
    _Float16* d = planes + row * PROW + 8 * o;
    *reinterpret_cast<h16x8*>(d) = h;
    *reinterpret_cast<h16x8*>(d + PLANE) = l;
  }
}

// One workgroup per operand (blockIdx.y) walks images blockIdx.x, + gridDim.x, ... on ONE load of its weight fragments: a
// workgroup per image re-read 64 KB of fragments for 16 KB of rows and the launch ran at the L2's pace (13 us at bs 512, 45 us
// at bs 2048); the next image's rows are requested before the current image's MFMAs.
__global__ __launch_bounds__(256, 3) void xattn_kv_h2_kernel(XKVArgs p, int B) {
  __shared__ __attribute__((aligned(16))) _Float16 planes[2 * PLANE];
  const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const bool is_v = blockIdx.y != 0;
  const float* src = is_v ? p.v_src : p.k_src;
  WFragH f;
  load_wh(f, p.in_h2 + ((long)((is_v ? 8 : 4) + h) * NS * 2) * 512 + lane * 8);      // Wk / Wv rows of head h
  float bias[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bias[r] = is_v ? p.in_b[2 * E128 + h * 32 + l31] : p.in_b[E128 + h * 32 + acc_row(r, hh)];
  RowRegs rr;
  long b = blockIdx.x;
  rows_load(rr, src + b * p.Sk * E128, p.Se, tid);
  rows_store(planes, rr, tid);
  __syncthreads();
  for (; b < B; b += gridDim.x) {
    const long nb = b + gridDim.x;
    if (nb < B) rows_load(rr, src + nb * p.Sk * E128, p.Se, tid);
    f32x16 a1 = {0}, a2 = {0}, a;
    if (!is_v) {
      project_t<true>(f, planes, l31, hh, a1, a2);                                      // K^T: rows d, columns keys
    } else {
      const _Float16* pa = planes + l31 * PROW + 8 * hh;
#pragma unroll
      for (int st = 0; st < NS; ++st) {
        const h16x8 xh = *reinterpret_cast<const h16x8*>(pa + 16 * st);
        const h16x8 xl = *reinterpret_cast<const h16x8*>(pa + 16 * st + PLANE);
        mfma3(xh, xl, f.w[st][0], f.w[st][1], a1, a2);                                  // V: rows keys, columns d
        if (st & 1) __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = a1[r] + a2[r] * LO_DOWN + bias[r];
    h16x8 hi[2], lo[2];
    pack_steps(a, hi, lo);
    _Float16* d = p.kv + (((b * 2 + (is_v ? 1 : 0)) * 4 + h) * 64 + lane) * 32;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      *reinterpret_cast<h16x8*>(d + 16 * t) = hi[t];
      *reinterpret_cast<h16x8*>(d + 16 * t + 8) = lo[t];
    }
    if (nb < B) {                                      // uniform over the workgroup
      __syncthreads();                                 // every wavefront has read this image's planes
      rows_store(planes, rr, tid);
      __syncthreads();
    }
  }
}

struct XAArgs {
  const float* q_src;                  // [B][Sq][128]
  const uint8_t* mask;                 // [B][mask_ld] (nullable)
  const _Float16* kv;                  // from xattn_kv_h2_kernel (the two-launch form)
  const float *k_src, *v_src;          // [B][mask_ld][128]: the one-launch form projects them itself
  const _Float16 *in_h2, *out_h2;
  const float *in_b, *out_b;
  float* out;                          // [B][Sq][128]
  int Sq, Se, mask_ld;
  float scale;
  int B, tiles;                        // images, query tiles (of 32 NSUB queries) per image
};

// OWNKV: the ONE-LAUNCH form of small calls (every workgroup resident at once, the call a latency chain): each query tile
// projects its image's K and V itself -- the same MFMAs on the same operands in the same order as xattn_kv_h2_kernel, the packed
// accumulators kept in the registers the record would have been loaded into, so the two forms agree bit for bit --, which
// trades the K / V launch and its record's round trip through L2 for two more projections per workgroup.
template <int NSUB, bool OWNKV = false>
__global__ __launch_bounds__(256, OWNKV ? 2 : 3) void xattn_main_h2_kernel(XAArgs p) {
  __shared__ __attribute__((aligned(16))) _Float16 planes[(NSUB + (OWNKV ? 2 : 0)) * 2 * PLANE];   // query rows, later the context rows (+ key and value rows)
  __shared__ float Ms[32];
  const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  // XCD-aware: workgroup w runs on XCD w mod 8 (round-robin dispatch); all tiles of an image go to ONE XCD, so its K / V record
  // is fetched into that L2 once instead of once per tile (84 of the call's 279 MB at bs 512: profiles/r03_cross_attention_roofline.txt)
  const int tiles = p.tiles;
  int b_, t_;
  {
    const long w = blockIdx.x, nimg8 = p.B >> 3;              // images in whole groups of eight
    const long full = nimg8 * 8 * tiles;
    if (w < full) {
      const long xcd = w & 7, idx = w >> 3;
      b_ = (int)((idx / tiles) * 8 + xcd);
      t_ = (int)(idx % tiles);
    } else {                                                   // the last B mod 8 images: plain order
      const long r = w - full;
      b_ = (int)(nimg8 * 8 + r / tiles);
      t_ = (int)(r % tiles);
    }
  }
  const long b = b_;
  const int q0 = t_ * TM * NSUB;
  constexpr float NEG_INF = -__builtin_inff();

  WFragH fq;
  if (OWNKV) load_wh(fq, p.in_h2 + ((long)(4 + h) * NS * 2) * 512 + lane * 8);     // Wk rows of head h first
  else load_wh(fq, p.in_h2 + ((long)h * NS * 2) * 512 + lane * 8);                 // Wq rows of head h
  if (OWNKV) {
    stage_rows_h2(planes + NSUB * 2 * PLANE, p.k_src + b * p.mask_ld * E128, 0, p.Se, tid);
    stage_rows_h2(planes + (NSUB + 1) * 2 * PLANE, p.v_src + b * p.mask_ld * E128, 0, p.Se, tid);
  }
#pragma unroll
  for (int u = 0; u < NSUB; ++u) stage_rows_h2(planes + u * 2 * PLANE, p.q_src + b * p.Sq * E128, q0 + u * TM, p.Sq, tid);
  if (tid < 32) Ms[tid] = (tid >= p.Se || (p.mask != nullptr && p.mask[b * p.mask_ld + tid] != 0)) ? NEG_INF : 0.f;
  // this lane's K and V operands of the image's record (L2): [step][part][8] fp16 each, in flight under the Q projection
  const _Float16* kf = OWNKV ? nullptr : p.kv + ((b * 2 * 4 + h) * 64 + lane) * 32;
  h16x8 kh[2], kl[2], vh[2], vl[2];
  if (!OWNKV) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      kh[t] = *reinterpret_cast<const h16x8*>(kf + 16 * t);
      kl[t] = *reinterpret_cast<const h16x8*>(kf + 16 * t + 8);
    }
  }
  // Q is scaled by log2(e) / sqrt(32): the scores come out in the log2 domain and the softmax is a bare v_exp_f32 (exp2)
  const float qs1 = p.scale * 1.44269504088896340736f, qs2 = qs1 * LO_DOWN;
  float bq[16];
#pragma unroll
  for (int g = 0; g < 4; ++g) {                                                    // bias of rows acc_row(4 g .. 4 g + 3, hh)
    const float4 t = ld4(p.in_b + h * 32 + 8 * g + 4 * hh);
    bq[4 * g] = t.x * qs1; bq[4 * g + 1] = t.y * qs1; bq[4 * g + 2] = t.z * qs1; bq[4 * g + 3] = t.w * qs1;
  }
  __syncthreads();

  if (OWNKV) {
    // K^T (rows d, columns keys), then V (rows keys, columns d): xattn_kv_h2_kernel's arithmetic, the results kept in registers
    {
      f32x16 a1 = {0}, a2 = {0}, a;
      project_t<true>(fq, planes + NSUB * 2 * PLANE, l31, hh, a1, a2);
#pragma unroll
      for (int r = 0; r < 16; ++r) a[r] = a1[r] + a2[r] * LO_DOWN + p.in_b[E128 + h * 32 + acc_row(r, hh)];
      pack_steps(a, kh, kl);
    }
    load_wh(fq, p.in_h2 + ((long)(8 + h) * NS * 2) * 512 + lane * 8);              // Wv rows of head h
    {
      f32x16 a1 = {0}, a2 = {0}, a;
      const _Float16* pa = planes + (NSUB + 1) * 2 * PLANE + l31 * PROW + 8 * hh;
#pragma unroll
      for (int st = 0; st < NS; ++st) {
        const h16x8 xh = *reinterpret_cast<const h16x8*>(pa + 16 * st);
        const h16x8 xl = *reinterpret_cast<const h16x8*>(pa + 16 * st + PLANE);
        mfma3(xh, xl, fq.w[st][0], fq.w[st][1], a1, a2);
        if (st & 1) __builtin_amdgcn_sched_barrier(0);
      }
      const float bv = p.in_b[2 * E128 + h * 32 + l31];
#pragma unroll
      for (int r = 0; r < 16; ++r) a[r] = a1[r] + a2[r] * LO_DOWN + bv;
      pack_steps(a, vh, vl);
    }
    load_wh(fq, p.in_h2 + ((long)h * NS * 2) * 512 + lane * 8);                    // Wq rows of head h
  }

  // Q^T of head h, straight into the B-operand form of the score MFMAs
  h16x8 qh[NSUB][2], ql[NSUB][2];
#pragma unroll
  for (int u = 0; u < NSUB; ++u) {
    f32x16 q1 = {0}, q2 = {0}, q;
    project_t(fq, planes + u * 2 * PLANE, l31, hh, q1, q2);
#pragma unroll
    for (int r = 0; r < 16; ++r) q[r] = fmaf(q2[r], qs2, fmaf(q1[r], qs1, bq[r]));
    pack_steps(q, qh[u], ql[u]);
  }
  WFragH fo;
  if (NSUB == 1) load_wh(fo, p.out_h2 + ((long)h * NS * 2) * 512 + lane * 8);      // Wo rows 32 h ..: in flight from here on
  if (!OWNKV) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      vh[t] = *reinterpret_cast<const h16x8*>(kf + 4 * 64 * 32 + 16 * t);
      vl[t] = *reinterpret_cast<const h16x8*>(kf + 4 * 64 * 32 + 16 * t + 8);
    }
  }
  __syncthreads();                                   // every wavefront has read the query planes: they may take the context

#pragma unroll
  for (int u = 0; u < NSUB; ++u) {
    // scores^T: rows = keys, columns = queries, contraction over the head's 32 channels in two steps
    f32x16 s1 = {0}, s2 = {0}, s;
#pragma unroll
    for (int t = 0; t < 2; ++t) mfma3(kh[t], kl[t], qh[u][t], ql[u][t], s1, s2);
    // two sub-tiles: Wo requested once the first sub-tile's registers are free, in flight under the last softmax / context
    if (NSUB > 1 && u == NSUB - 1) load_wh(fo, p.out_h2 + ((long)h * NS * 2) * 512 + lane * 8);
    float tmax = NEG_INF;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = s1[r] + s2[r] * LO_DOWN + Ms[acc_row(r, hh)];
      tmax = fmaxf(tmax, s[r]);
    }
    tmax = xor32_max(tmax);
    const bool none = tmax == NEG_INF;               // every key masked for this image: 0 / 0 = NaN, as torch
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pr = none ? 0.f : __builtin_amdgcn_exp2f(s[r] - tmax);
      s[r] = pr;
      psum += pr;
    }
    const float inv = 1.0f / xor32_sum(psum);
    // context^T: rows = d, columns = queries, contraction over the 32 keys: register r of s is key acc_row(r, hh) = slot order
    h16x8 ph[2], pl[2];
    pack_steps(s, ph, pl);
    f32x16 o1 = {0}, o2 = {0};
#pragma unroll
    for (int t = 0; t < 2; ++t) mfma3(vh[t], vl[t], ph[t], pl[t], o1, o2);
    // register r = context[query l31][d = acc_row(r, hh)] of head h -> two fp16 planes, row = query, column 32 h + d
    // (register quad 4 g .. 4 g + 3 = four consecutive columns: one 8-byte LDS store per plane)
    _Float16* d = planes + u * 2 * PLANE + l31 * PROW + h * 32 + 4 * hh;
    const float inv2 = inv * LO_DOWN;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      h16x4 a, c;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        _Float16 x, y;
        split_h2(fmaf(o2[4 * g + j], inv2, o1[4 * g + j] * inv), x, y);
        a[j] = x;
        c[j] = y;
      }
      *reinterpret_cast<h16x4*>(d + 8 * g) = a;
      *reinterpret_cast<h16x4*>(d + 8 * g + PLANE) = c;
    }
  }
  float4 bo[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bo[g] = ld4(p.out_b + h * 32 + 8 * g + 4 * hh);
  __syncthreads();

  // output projection transposed, 32 columns per wavefront: register 4 g + j = out[query l31][32 h + 8 g + 4 hh + j]
#pragma unroll
  for (int u = 0; u < NSUB; ++u) {
    f32x16 a1 = {0}, a2 = {0};
    project_t(fo, planes + u * 2 * PLANE, l31, hh, a1, a2);
    const int qi = q0 + u * TM + l31;
    if (qi < p.Sq) {
      float* od = p.out + (b * p.Sq + qi) * E128 + h * 32 + 4 * hh;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(od + 8 * g) =
            make_float4(a1[4 * g] + a2[4 * g] * LO_DOWN + bo[g].x, a1[4 * g + 1] + a2[4 * g + 1] * LO_DOWN + bo[g].y,
                        a1[4 * g + 2] + a2[4 * g + 2] * LO_DOWN + bo[g].z, a1[4 * g + 3] + a2[4 * g + 3] * LO_DOWN + bo[g].w);
    }
  }
}

int g_one_launch_wgs = 320;          // one launch up to here (measured: 14.9 / 18.2 us against 15.8 / 18.7 at bs 16 / 32 x 300 tokens, behind at bs 64: 24.0 / 22.5)

}  // namespace

extern "C" int ocv_mha_few_keys_h2_set_dispatch(int one_launch_max_workgroups) {
  g_one_launch_wgs = one_launch_max_workgroups < 0 ? 320 : one_launch_max_workgroups;
  return 0;
}

extern "C" size_t ocv_split_h2_packed_elems(int N, int K) {
  if (N < 1 || K < 1) return 0;
  return (size_t)((N + 31) / 32) * ((K + 15) / 16) * 2 * 512;
}

extern "C" int ocv_pack_split_h2_fwd(const float* W, int ldw, int N, int K, void* packed, ocv_stream_t stream) {
  OCV_CHECK_ARG(W && packed, "ocv_pack_split_h2_fwd: null pointer");
  OCV_CHECK_ARG(N >= 1 && K >= 1 && ldw >= K, "ocv_pack_split_h2_fwd: bad sizes N=%d K=%d ldw=%d", N, K, ldw);
  OCV_CHECK_ARG(ocv_aligned16(packed), "ocv_pack_split_h2_fwd: packed must be 16-byte aligned");
  const long items = (long)((N + 31) / 32) * ((K + 15) / 16) * 64;
  hipLaunchKernelGGL(pack_h2_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, ldw, N, K,
                     (_Float16*)packed, items);
  OCV_CHECK_LAUNCH("ocv_pack_split_h2_fwd");
  return 0;
}

extern "C" size_t ocv_mha_few_keys_h2_workspace_bytes(int B) { return B < 1 ? 0 : (size_t)B * 2 * 4 * 64 * 32 * sizeof(_Float16); }

extern "C" int ocv_mha_few_keys_h2_fwd(const float* q_src, const float* k_src, const float* v_src, const uint8_t* key_padding_mask,
                                       const void* in_proj_h2, const float* in_proj_b, const void* out_proj_h2, const float* out_b,
                                       float* out, int B, int Sq, int Sk, int kv_limit, int E, int H, void* workspace,
                                       size_t workspace_bytes, ocv_stream_t stream) {
  OCV_CHECK_ARG(q_src && k_src && v_src && in_proj_h2 && in_proj_b && out_proj_h2 && out_b && out && workspace,
                "ocv_mha_few_keys_h2_fwd: null pointer");
  OCV_CHECK_ARG(H == 4 && E == E128, "ocv_mha_few_keys_h2_fwd: built for E = 128, H = 4 (got %d, %d)", E, H);
  OCV_CHECK_ARG(B >= 1 && B <= 65535 && Sq >= 1 && Sk >= 1, "ocv_mha_few_keys_h2_fwd: bad sizes (B=%d Sq=%d Sk=%d)", B, Sq, Sk);
  OCV_CHECK_ARG(kv_limit >= 0, "ocv_mha_few_keys_h2_fwd: negative kv_limit");
  OCV_CHECK_ARG(kv_limit == 0 || key_padding_mask != nullptr, "ocv_mha_few_keys_h2_fwd: kv_limit needs a key_padding_mask");
  const int Se = (kv_limit > 0 && kv_limit < Sk) ? kv_limit : Sk;
  OCV_CHECK_ARG(Se <= 32, "ocv_mha_few_keys_h2_fwd: at most 32 live keys (got %d); use ocv_mha_split3_fwd", Se);
  OCV_CHECK_ARG(workspace_bytes >= ocv_mha_few_keys_h2_workspace_bytes(B), "ocv_mha_few_keys_h2_fwd: workspace too small");
  OCV_CHECK_ARG(ocv_aligned16(q_src) && ocv_aligned16(k_src) && ocv_aligned16(v_src) && ocv_aligned16(in_proj_h2) &&
                    ocv_aligned16(out_proj_h2) && ocv_aligned16(workspace) && ocv_aligned16(in_proj_b) && ocv_aligned16(out_b) &&
                    ocv_aligned16(out),
                "ocv_mha_few_keys_h2_fwd: every pointer must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  XAArgs a{q_src, key_padding_mask, (const _Float16*)workspace, k_src, v_src, (const _Float16*)in_proj_h2,
           (const _Float16*)out_proj_h2, in_proj_b, out_b, out, Sq, Se, Sk, 1.0f / sqrtf(32.0f), B, 0};
  if ((long)ocv_cdiv(Sq, TM) * B <= g_one_launch_wgs) {      // small call: ONE launch, K / V projected by every query tile
    a.tiles = ocv_cdiv(Sq, TM);
    hipLaunchKernelGGL((xattn_main_h2_kernel<1, true>), dim3((unsigned)((long)a.tiles * B)), dim3(256), 0, st, a);
    OCV_CHECK_LAUNCH("ocv_mha_few_keys_h2_fwd(one launch)");
    return 0;
  }
  XKVArgs ka{k_src, v_src, (const _Float16*)in_proj_h2, in_proj_b, (_Float16*)workspace, Sk, Se};
  // one image per workgroup while they are all resident at once, several per workgroup beyond
  const int per_wg = ocv_cdiv(B, 256);               // images per workgroup, evenly: bs 300 -> 150 workgroups x 2 (384 per operand measured no faster)
  hipLaunchKernelGGL(xattn_kv_h2_kernel, dim3(ocv_cdiv(B, per_wg), 2), dim3(256), 0, st, ka, B);
  OCV_CHECK_LAUNCH("ocv_mha_few_keys_h2_fwd(K / V projection)");
  // two sub-tiles per workgroup once the 64-query workgroups alone fill the chip several times over
  const long wg64 = (long)ocv_cdiv(Sq, 2 * TM) * B;
  const int nsub = wg64 >= 2048 ? 2 : 1;
  a.tiles = ocv_cdiv(Sq, nsub * TM);
  OCV_CHECK_ARG((long)a.tiles * B < (1L << 31), "ocv_mha_few_keys_h2_fwd: grid too large");
  if (nsub == 2)
    hipLaunchKernelGGL(xattn_main_h2_kernel<2>, dim3((unsigned)((long)a.tiles * B)), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(xattn_main_h2_kernel<1>, dim3((unsigned)((long)a.tiles * B)), dim3(256), 0, st, a);
  OCV_CHECK_LAUNCH("ocv_mha_few_keys_h2_fwd(fused)");
  return 0;
}
