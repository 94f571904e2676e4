// The token-local tail of a post-norm transformer encoder layer on TWO-TERM fp16 SPLITS (round 3) -- the arithmetic of
// csrc/xattn_h2.hip (x = hi + 2^-11 lo', three v_mfma_f32_32x32x16_f16 per product block into an accumulator pair, 22-bit
// products = the error of an fp32 FMA chain) applied to layer_tail3_kernel of csrc/token_split3.hip:
//   x1 = LayerNorm1(x + ctx Wo^T + bo)
//   x2 = LayerNorm2(x1 + W2 relu(W1 x1 + b1) + b2)          -> out
//   qkv_next = x2 Wqkv'^T + bqkv'                           -> the NEXT layer's packed q | k | v rows (if there is one)
// (nn.TransformerEncoderLayer, eval mode: reference modules/ObjCAViT.py:155-161,169,188 and modules/layers.py:8-9,23.)
// Half the MFMAs of the three-term bf16 form (480 instead of 960 per wavefront and 32 tokens) and two weight-fragment
// parts instead of three: a workgroup streams 1.27 MB of fragments instead of 1.9 MB -- and the launch is bound by exactly
// that stream (twenty phases, each waiting for fragments requested one phase earlier: 60 us per launch whatever the
// number of workgroups).  LDS 51 KB instead of 69.
// fp16's range applies to the tokens, the hidden activations and the weights (packer saturates); OCV_TOKENS=split3 keeps
// the three-term bf16 kernels, OCV_TOKENS=fp32 the exact ones.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "common.hpp"
#include "../../include/objcavit_hip.h"

namespace {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

constexpr int TM = 32;               // tokens per workgroup
constexpr int E128 = 128;
constexpr int KC = 128;              // hidden units per feed-forward chunk
constexpr int NS = E128 / 16;        // K steps of a 128-long contraction
constexpr int PROW = E128 + 8;       // fp16 per plane row (272 bytes)
constexpr int PLANE = TM * PROW;
constexpr float LO_UP = 2048.0f, LO_DOWN = 1.0f / 2048.0f;

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

// rows [m0, m0 + 32) of a row-major [M][128] fp32 matrix -> two fp16 planes (rows >= M: zeros)
__device__ __forceinline__ void stage_rows(_Float16* planes, const float* __restrict__ src, int m0, int M, int tid) {
  const int row = tid >> 3;
  const bool ok = m0 + row < M;
  const float* s = src + (long)(ok ? m0 + row : 0) * E128;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int o = (tid & 7) + 8 * i;
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

// the 32 x 128 fp32 tile in Cs -> two fp16 planes
__device__ __forceinline__ void tile_to_planes(_Float16* planes, const float (*Cs)[E128 + 1], int tid) {
  const int row = tid >> 3;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int o = (tid & 7) + 8 * i;
    const float* c = &Cs[row][8 * o];
    h16x8 h, l;
    split8_h2(make_float4(c[0], c[1], c[2], c[3]), make_float4(c[4], c[5], c[6], c[7]), h, l);
    _Float16* d = planes + row * PROW + 8 * o;
    *reinterpret_cast<h16x8*>(d) = h;
    *reinterpret_cast<h16x8*>(d + PLANE) = l;
  }
}

// eight K steps of one 32-row weight tile: 8 x 2 parts x 16 bytes per lane = 64 VGPRs
struct WFragH { h16x8 w[NS][2]; };

__device__ __forceinline__ void load_wh(WFragH& f, const _Float16* wp) {
#pragma unroll
  for (int s = 0; s < NS; ++s) {
#pragma unroll
    for (int p = 0; p < 2; ++p) f.w[s][p] = *reinterpret_cast<const h16x8*>(wp + (long)s * 1024 + p * 512);
  }
}

// (a1, a2) += planes[32 x 128] . (fragments)^T : rows = tokens, columns = the tile's 32 outputs
__device__ __forceinline__ void chunk_h2(f32x16& a1, f32x16& a2, const _Float16* planes, const WFragH& f, int l31, int hh) {
  const _Float16* pa = planes + l31 * PROW + 8 * hh;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const h16x8 ah = *reinterpret_cast<const h16x8*>(pa + 16 * s);
    const h16x8 al = *reinterpret_cast<const h16x8*>(pa + 16 * s + PLANE);
    a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, f.w[s][0], a2, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, f.w[s][1], a2, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, f.w[s][0], a1, 0, 0, 0);
  }
}

// LayerNorm of the 32 x 128 tile in Cs, in place (one wavefront per 8 rows)
__device__ __forceinline__ void ln_tile_inplace(float (*Cs)[E128 + 1], const float* gamma, const float* beta, float eps, int lane,
                                                int wave) {
  const float g0 = gamma[lane], g1 = gamma[lane + 64];
  const float b0 = beta[lane], b1 = beta[lane + 64];
#pragma unroll 4
  for (int i = 0; i < TM / 4; ++i) {
    const int row = wave * (TM / 4) + i;
    const float x0 = Cs[row][lane], x1 = Cs[row][lane + 64];
    const float mean = wave_sum(x0 + x1) * (1.0f / E128);
    const float d0 = x0 - mean, d1 = x1 - mean;
    const float var = wave_sum(d0 * d0 + d1 * d1) * (1.0f / E128);
    const float rstd = 1.0f / sqrtf(var + eps);
    Cs[row][lane] = d0 * rstd * g0 + b0;
    Cs[row][lane + 64] = d1 * rstd * g1 + b1;
  }
}

struct TailArgs {
  const float *ctx, *x;                 // [M][128]
  const _Float16 *wo_p, *w1_p, *w2_p, *wqkv_p;     // wqkv_p: the next layer's packed in_proj (nullable)
  const float *bo, *g1, *be1, *b1, *b2, *g2, *be2, *bqkv;
  float eps;
  const uint8_t* zero_mask;             // rows written as 0 in `out` (last layer of a masked stack), nullable
  float* out;                           // [M][128]
  float* qkv;                           // [M][384] (nullable with wqkv_p)
  int M, FF;
  float* part;                          // G > 1: [row blocks][G][32][128] raw feed-forward partial sums
  unsigned* cnt;                        // G > 1: [row blocks] arrival tickets, zero at launch, zero again afterwards
};

// Three fragment sets rotate (64 registers each): while a phase multiplies from one, the next TWO phases' fragments are
// in flight -- the three-term kernel, with 96-register sets, could hold two and waited 2 - 3 us per phase for fragments
// requested one 0.75 us phase earlier.
//
// G > 1 (round 4, few tokens -- the reference's own batch of 1 - 2 images, 10 - 19 row blocks on 256 CUs): the launch is bound by
// the 20-phase fragment stream of ONE workgroup (~50 us whatever the batch), so the eight feed-forward chunks of a row block go to
// G workgroups (each repeats the output projection and LN1, walks nchunk / G chunks and leaves its raw partial sums write-through),
// and the row block's LAST workgroup to arrive -- an agent-scope ticket, nobody waits (common.hpp) -- adds the partials in the
// fixed order 0 .. G - 1, does LN2 and the next layer's projection: 3 + 4 phases instead of 20.  G = 1 is the kernel of round 3,
// instruction for instruction.
template <int G>
__global__ __launch_bounds__(256, 1) void layer_tail_h2_kernel(TailArgs p) {
  __shared__ __attribute__((aligned(16))) _Float16 xp[2 * PLANE];   // ctx, then x1, then x2 (two planes)
  __shared__ __attribute__((aligned(16))) _Float16 hp[2 * PLANE];   // hidden chunk
  __shared__ float Cs[TM][E128 + 1];                                 // fp32: sums, x1, x2
  __shared__ unsigned last_flag;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int blk = G == 1 ? (int)blockIdx.x : (int)blockIdx.x / G, g = G == 1 ? 0 : (int)blockIdx.x % G;
  const int m0 = blk * TM;
  const int col = wave * 32 + l31;
  const int ksteps2 = p.FF >> 4, nchunk_all = p.FF / KC;
  const int c0 = nchunk_all * g / G, nchunk = nchunk_all * (g + 1) / G - c0;        // this workgroup's chunks [c0, c0 + nchunk)
  auto w1_at = [&](int c) { return p.w1_p + ((long)((c0 + c) * 4 + wave) * NS * 2) * 512 + lane * 8; };
  auto w2_at = [&](int c) { return p.w2_p + (((long)wave * ksteps2 + (c0 + c) * (KC / 16)) * 2) * 512 + lane * 8; };

  // ---- x1 = LN1(x + ctx Wo^T + bo)
  WFragH fa, fb, fc;                    // roles rotate: see the feed-forward loop
  load_wh(fa, p.wo_p + ((long)wave * NS * 2) * 512 + lane * 8);
  load_wh(fb, w1_at(0));                // W1 chunk 0 and W2 chunk 0: in flight under the output projection and LN1
  load_wh(fc, w2_at(0));
  stage_rows(xp, p.ctx, m0, p.M, tid);
  __syncthreads();
  {
    f32x16 a1 = {0}, a2 = {0};
    chunk_h2(a1, a2, xp, fa, l31, hh);
    if (nchunk > 1) load_wh(fa, w1_at(1));
    const float bo = p.bo[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = acc_row(r, hh), m = m0 + row;
      Cs[row][col] = a1[r] + a2[r] * LO_DOWN + bo + (m < p.M ? p.x[(long)m * E128 + col] : 0.f);
    }
  }
  __syncthreads();
  ln_tile_inplace(Cs, p.g1, p.be1, p.eps, lane, wave);
  __syncthreads();
  tile_to_planes(xp, Cs, tid);                        // every wavefront is past its reads of the ctx planes
  __syncthreads();

  // ---- x2 = LN2(x1 + W2 relu(W1 x1 + b1) + b2).  Chunk c: phase 1 multiplies by W1(c), phase 2 by W2(c).
  // Entering chunk c the sets hold  s1 = W1(c), s2 = W2(c), s3 = W1(c + 1) (requested a whole chunk ago);  after phase 1 the
  // freed set takes W2(c + 1), after phase 2 the next freed one takes W1(c + 2): every fragment is requested two phases ahead.
  f32x16 acc1 = {0}, acc2 = {0};
  auto ffn_chunk = [&](int c, WFragH& s1, WFragH& s2, WFragH& s3) {
    const float b1 = p.b1[(c0 + c) * KC + col];
    f32x16 h1 = {0}, h2 = {0};
    chunk_h2(h1, h2, xp, s1, l31, hh);
    if (c + 1 < nchunk) load_wh(s1, w2_at(c + 1));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      _Float16 a, b;
      split_h2(fmaxf(h1[r] + h2[r] * LO_DOWN + b1, 0.f), a, b);
      _Float16* d = hp + acc_row(r, hh) * PROW + col;
      d[0] = a;
      d[PLANE] = b;
    }
    __syncthreads();
    chunk_h2(acc1, acc2, hp, s2, l31, hh);
    if (c + 2 < nchunk) load_wh(s2, w1_at(c + 2));
    __syncthreads();
    (void)s3;
  };
  // rotation of (W1(c), W2(c), W1(c + 1)) over (fb, fc, fa): period three chunks
  int c = 0;
  for (; c + 3 <= nchunk; c += 3) {
    ffn_chunk(c, fb, fc, fa);          // frees fb -> W2(c + 1), fc -> W1(c + 2)
    ffn_chunk(c + 1, fa, fb, fc);      // W1(c + 1) in fa, W2(c + 1) in fb, W1(c + 2) in fc
    ffn_chunk(c + 2, fc, fa, fb);      // W1(c + 2) in fc, W2(c + 2) in fa, W1(c + 3) in fb
  }
  if (c < nchunk) {
    ffn_chunk(c, fb, fc, fa);
    if (c + 1 < nchunk) ffn_chunk(c + 1, fa, fb, fc);
  }
  if constexpr (G > 1) {
    // raw partial sums of this workgroup's chunks, write-through; then the ticket
    float* mine = p.part + ((long)(blk * G + g) * TM) * E128 + col;
#pragma unroll
    for (int r = 0; r < 16; ++r) ocv_store_sc1(mine + acc_row(r, hh) * E128, acc1[r] + acc2[r] * LO_DOWN);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const unsigned old = __hip_atomic_fetch_add((ocv_gu32*)(p.cnt + blk), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool last = old == (unsigned)(G - 1);
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store((ocv_gu32*)(p.cnt + blk), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // nobody else touches it any more
      }
      last_flag = last ? 1u : 0u;
    }
    __syncthreads();
    if (last_flag == 0u) return;
    // the row block's last workgroup: partials of all G workgroups, fixed order (its own comes back from memory like the others)
    const float* all = p.part + ((long)blk * G * TM) * E128 + col;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = acc_row(r, hh);
      float v[G];
#pragma unroll
      for (int gg = 0; gg < G; ++gg) v[gg] = all[((long)gg * TM + row) * E128];
      float sacc = v[0];
#pragma unroll
      for (int gg = 1; gg < G; ++gg) sacc += v[gg];
      acc1[r] = sacc;
      acc2[r] = 0.f;
    }
  }
  const bool next = p.wqkv_p != nullptr;
  WFragH& q0 = fa;
  WFragH& q1 = fb;
  WFragH& q2 = fc;
  if (next) {                                                      // the three projection tiles of this wavefront: under LN2
    load_wh(q0, p.wqkv_p + ((long)(wave * 3 + 0) * NS * 2) * 512 + lane * 8);
    load_wh(q1, p.wqkv_p + ((long)(wave * 3 + 1) * NS * 2) * 512 + lane * 8);
    load_wh(q2, p.wqkv_p + ((long)(wave * 3 + 2) * NS * 2) * 512 + lane * 8);
  }
  {
    const float b2 = p.b2[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = acc_row(r, hh);
      Cs[row][col] = acc1[r] + acc2[r] * LO_DOWN + b2 + Cs[row][col];   // residual = x1 (each element read and written by one lane)
    }
  }
  __syncthreads();
  ln_tile_inplace(Cs, p.g2, p.be2, p.eps, lane, wave);
  __syncthreads();
  for (int i = tid; i < TM * (E128 / 4); i += 256) {     // x2 -> out, 16-byte stores
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
  auto qkv_tile = [&](int t, const WFragH& f) {
    const int n = (wave * 3 + t) * 32 + l31;
    const float bn = p.bqkv[n];
    f32x16 a1 = {0}, a2 = {0};
    chunk_h2(a1, a2, xp, f, l31, hh);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + acc_row(r, hh);
      if (m < p.M) p.qkv[(long)m * 3 * E128 + n] = a1[r] + a2[r] * LO_DOWN + bn;
    }
  };
  qkv_tile(0, q0);
  qkv_tile(1, q1);
  qkv_tile(2, q2);
}

}  // namespace

extern "C" int ocv_layer_tail_h2_fwd(const float* ctx, const float* x, const ocv_encoder_layer_params* p_caller,
                                     const void* next_in_proj_h2, const float* next_in_proj_b, float eps,
                                     const uint8_t* zero_row_mask, float* out, float* qkv_next, int M, int E, int FF,
                                     ocv_stream_t stream) {
  return ocv_layer_tail_h2_ws_fwd(ctx, x, p_caller, next_in_proj_h2, next_in_proj_b, eps, zero_row_mask, out, qkv_next, M, E, FF, nullptr, 0,
                                  stream);
}

// Feed-forward chunks per row block over G workgroups: only where the launch leaves most of the chip idle (see the kernel)
extern "C" int ocv_layer_tail_h2_groups(int M, int FF) {
  const int nblk = ocv_cdiv(M, TM), nchunk = FF / KC;
  int G = nblk <= 24 ? 8 : nblk <= 56 ? 4 : 1;          // (2 groups at 75 - 105 row blocks measured neutral: 870 vs 870, 641 vs 640 img/s)
  while (G > 1 && nchunk % G != 0) G >>= 1;
  return G;
}

extern "C" size_t ocv_layer_tail_h2_workspace_bytes(int M, int FF) {
  if (M < 1 || FF < KC) return 0;
  const int G = ocv_layer_tail_h2_groups(M, FF);
  if (G == 1) return 0;
  const size_t nblk = (size_t)ocv_cdiv(M, TM);
  return nblk * G * TM * E128 * sizeof(float) + ((nblk * sizeof(unsigned) + 255) / 256) * 256;
}

// workspace (ocv_layer_tail_h2_workspace_bytes; may be null / 0 = one workgroup per row block): partial sums, then the row blocks'
// arrival tickets, which must be ZERO when the call starts and are zero again when it has finished.
extern "C" int ocv_layer_tail_h2_ws_fwd(const float* ctx, const float* x, const ocv_encoder_layer_params* p_caller,
                                        const void* next_in_proj_h2, const float* next_in_proj_b, float eps,
                                        const uint8_t* zero_row_mask, float* out, float* qkv_next, int M, int E, int FF,
                                        void* workspace, size_t workspace_bytes, ocv_stream_t stream) {
  OCV_CHECK_ARG(ctx && x && p_caller && out, "ocv_layer_tail_h2_fwd: null pointer");
  ocv_encoder_layer_params pv;
  OCV_CHECK_ARG(ocv_layer_params_view(p_caller, 0, &pv), "ocv_layer_tail_h2_fwd: params->struct_size (%zu) is not a valid ocv_encoder_layer_params size", p_caller->struct_size);
  const ocv_encoder_layer_params* p = &pv;
  OCV_CHECK_ARG(p->out_proj_h2 && p->linear1_h2 && p->linear2_h2, "ocv_layer_tail_h2_fwd: needs the packed two-term fp16 weights (ocv_pack_split_h2_fwd)");
  OCV_CHECK_ARG(E == E128 && FF >= KC && FF % KC == 0, "ocv_layer_tail_h2_fwd: needs E = %d and FF a multiple of %d (got %d, %d)", E128, KC, E, FF);
  OCV_CHECK_ARG((next_in_proj_h2 == nullptr) == (qkv_next == nullptr) && (next_in_proj_h2 == nullptr || next_in_proj_b != nullptr),
                "ocv_layer_tail_h2_fwd: next_in_proj_h2 / next_in_proj_b / qkv_next go together");
  OCV_CHECK_ARG(M >= 0 && ocv_aligned16(ctx) && ocv_aligned16(x) && ocv_aligned16(out) && ocv_aligned16(qkv_next) &&
                    ocv_aligned16(p->out_proj_h2) && ocv_aligned16(p->linear1_h2) && ocv_aligned16(p->linear2_h2) && ocv_aligned16(next_in_proj_h2),
                "ocv_layer_tail_h2_fwd: bad M / alignment");
  if (M == 0) return 0;
  TailArgs a{ctx, x, (const _Float16*)p->out_proj_h2, (const _Float16*)p->linear1_h2, (const _Float16*)p->linear2_h2,
             (const _Float16*)next_in_proj_h2, p->out_proj_b, p->norm1_w, p->norm1_b, p->linear1_b, p->linear2_b, p->norm2_w,
             p->norm2_b, next_in_proj_b, eps, zero_row_mask, out, qkv_next, M, FF, nullptr, nullptr};
  const int nblk = ocv_cdiv(M, TM);
  const size_t need = ocv_layer_tail_h2_workspace_bytes(M, FF);
  const int G = (workspace != nullptr && need != 0 && workspace_bytes >= need && ocv_aligned16(workspace)) ? ocv_layer_tail_h2_groups(M, FF) : 1;
  if (G > 1) {
    a.part = (float*)workspace;
    a.cnt = (unsigned*)((char*)workspace + (size_t)nblk * G * TM * E128 * sizeof(float));
  }
  switch (G) {
    case 8: hipLaunchKernelGGL(layer_tail_h2_kernel<8>, dim3(nblk * 8), dim3(256), 0, (hipStream_t)stream, a); break;
    case 4: hipLaunchKernelGGL(layer_tail_h2_kernel<4>, dim3(nblk * 4), dim3(256), 0, (hipStream_t)stream, a); break;
    case 2: hipLaunchKernelGGL(layer_tail_h2_kernel<2>, dim3(nblk * 2), dim3(256), 0, (hipStream_t)stream, a); break;
    default: hipLaunchKernelGGL(layer_tail_h2_kernel<1>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, a); break;
  }
  OCV_CHECK_LAUNCH("ocv_layer_tail_h2_fwd");
  return 0;
}
