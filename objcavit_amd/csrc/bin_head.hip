// Fused AdaBins-style bin head and pixel-wise dot product for gfx950.
//
// bin head, per pixel p of image b (C = 128 channels, 256 bins):
//   logits[k] = bout[k] + sum_c Wf[b][k][c] * feat[b][c][p],   Wf[b] = Wout . queries[b]   (256 x 128)
//   depth     = sum_k softmax_k(logits) * centers[b][k]
// The feature map is streamed from HBM exactly once (4 B x 128 per pixel in,
// 4 B out); the 128-channel range-attention maps and the 256-bin logits and
// probabilities (433 MB/img in the reference's un-fused form) never exist.
//
// Mapping: a workgroup = 4 wavefronts, persistent over 128-pixel tiles of one
// image; Wf[b] (128 KiB) is staged ONCE per workgroup into LDS with rows padded
// to 129 floats.  A wavefront owns 32 consecutive pixels: its B operand
// (feat[c][p], lane = pixel, 128 B coalesced per half-wave per channel) is
// loaded straight into 64 VGPRs -- every element is used by exactly one
// wavefront, so LDS staging would buy nothing -- while the next tile's 64
// loads are already in flight.  The 256 bins are walked in 8 tiles of 32:
// 64 x v_mfma_f32_32x32x2_f32 per tile (A operand = one ds_read_b32 per
// issue), then an online softmax update in registers.  A lane holds 16 bins of
// the tile for its pixel, lane^32 the other 16: max / sum / weighted sum need
// one wavefront shuffle (xor 32) per tile.
#include <stdlib.h>
#include <string.h>

#include "common.hpp"
#include "../../include/objcavit_hip.h"

namespace {

constexpr int CH = 128;        // channels == queries
constexpr int NB = 256;        // bins
constexpr int WLD = CH + 1;
constexpr int TP = 128;        // pixels per workgroup tile

// B operand of one 32-pixel wavefront tile: 64 VGPRs, lane = (pixel l31, k-slot hh).
// The K order of an MFMA chain is free as long as A and B agree, so it is chosen per layout:
//   NCHW  feat[c][p]: step s, slot hh <-> channel 2s + hh   (128-B coalesced run per half-wave per channel)
//   NHWC  feat[p][c]: step s, slot hh <-> channel 64hh + s  (each lane reads its pixel's 256 contiguous bytes)
template <bool NHWC>
__device__ __forceinline__ void load_pixels(float (&bf)[CH / 2], const float* __restrict__ fb, long P, long pix, int hh,
                                            bool ok) {
  if (!NHWC) {
    const float* src = fb + (long)hh * P + (ok ? pix : 0);
#pragma unroll
    for (int s = 0; s < CH / 2; ++s) bf[s] = ok ? src[(long)(2 * s) * P] : 0.f;
  } else {
    const float* src = fb + (ok ? pix : 0) * CH + 64 * hh;
#pragma unroll
    for (int s = 0; s < CH / 8; ++s) {
      const float4 t = ok ? ld4(src + 4 * s) : make_float4(0.f, 0.f, 0.f, 0.f);
      bf[4 * s + 0] = t.x; bf[4 * s + 1] = t.y; bf[4 * s + 2] = t.z; bf[4 * s + 3] = t.w;
    }
  }
}
// LDS column of the A operand for (step s, slot hh) under the same K order
template <bool NHWC>
__device__ __forceinline__ constexpr int a_col(int s, int hh) { return NHWC ? 64 * hh + s : 2 * s + hh; }

template <bool NHWC>
__global__ __launch_bounds__(256) void bin_head_kernel(const float* __restrict__ feat, const float* __restrict__ Wf,
                                                       const float* __restrict__ bout,
                                                       const float* __restrict__ centers, float* __restrict__ depth,
                                                       long P, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Wl = lds;                  // [256][129]
  float* bl = Wl + NB * WLD;        // [256]
  float* cl = bl + NB;              // [256]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int b = blockIdx.y;
  const float* fb = feat + (long)b * CH * P;
  const float* wb = Wf + (long)b * NB * CH;

  for (int i = tid; i < NB * CH / 4; i += 256) {
    const float4 t = ld4(wb + (long)i * 4);
    const int k = (i * 4) / CH, c = (i * 4) % CH;
    float* d = Wl + k * WLD + c;
    d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
  }
  bl[tid] = bout[tid];
  cl[tid] = centers[(long)b * NB + tid];
  __syncthreads();

  float cur[CH / 2], nxt[CH / 2];
  int tile = blockIdx.x;
  if (tile < ntiles) {
    const long pix = (long)tile * TP + wave * 32 + l31;
    load_pixels<NHWC>(cur, fb, P, pix, hh, pix < P);
  }
  for (; tile < ntiles; tile += gridDim.x) {
    const long pix = (long)tile * TP + wave * 32 + l31;
    const int tn = tile + gridDim.x;
    if (tn < ntiles) {
      const long pn = (long)tn * TP + wave * 32 + l31;
      load_pixels<NHWC>(nxt, fb, P, pn, hh, pn < P);
    }

    float m_run = -__builtin_inff(), l_half = 0.f, d_half = 0.f;
#pragma unroll 1
    for (int t = 0; t < NB / 32; ++t) {
      f32x16 acc = {0};
      const float* wrow = Wl + (t * 32 + l31) * WLD + a_col<NHWC>(0, hh);
#pragma unroll
      for (int s = 0; s < CH / 2; ++s) acc = mfma_32x32x2(wrow[a_col<NHWC>(s, 0)], cur[s], acc);

      float tmax = -__builtin_inff();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[r] += bl[t * 32 + acc_row(r, hh)];
        tmax = fmaxf(tmax, acc[r]);
      }
      tmax = xor32_max(tmax);
      const float m_new = fmaxf(m_run, tmax);
      const float alpha = fast_exp(m_run - m_new);
      float ps = 0.f, ds = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pr = fast_exp(acc[r] - m_new);
        ps += pr;
        ds += pr * cl[t * 32 + acc_row(r, hh)];
      }
      l_half = l_half * alpha + ps;
      d_half = d_half * alpha + ds;
      m_run = m_new;
    }
    const float l = xor32_sum(l_half), d = xor32_sum(d_half);
    if (hh == 0 && pix < P) depth[(long)b * P + pix] = d / l;

    if (tn < ntiles) {
#pragma unroll
      for (int s = 0; s < CH / 2; ++s) cur[s] = nxt[s];
    }
  }
}

typedef __bf16 bh_bf16x8 __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------------------
// THREE-term split bin head (NHWC map): fp32-faithful logits at the bf16 matrix rate.  Every operand is written as
// v = h + m + l with h = bf16(v), m = bf16(v - h), l = bf16(v - h - m) (24 significant bits: the two subtractions are
// exact in fp32), every product as the six terms h h + h m + m h + m m + h l + l h on v_mfma_f32_32x32x16_bf16 with
// fp32 accumulation -- the dropped terms are <= 2^-24 of the product, i.e. below fp32's own rounding of it -- at
// 6 x 32 = 192 matrix-pipe cycles per 32 x 32 x 16 block against 8 x 64 = 512 for the exact v_mfma_f32_32x32x2_f32
// kernel above (which runs at 70 % of the fp32 matrix peak and is bound by it).  The two-term kernel below cannot
// replace it: a near-one-hot bin softmax passes logit error straight into depth.
// Wf[b] in three parts is 192 KB, more than a CU's LDS, so the 256 bins are cut in two HALVES: a workgroup (8 wavefronts,
// two per SIMD: one's softmax arithmetic runs under the other's MFMAs) stages the three parts of its 128 bins once
// (96 KB, A-operand fragment order) and walks pixel tiles like the kernels above; it leaves the online-softmax state of
// its half per pixel -- (running max, sum of exponentials, sum of exponentials x centre) -- and
// bin_head_combine_kernel merges the two halves (fixed order) into depth.  The map is read twice (the second time
// mostly from L2 / Infinity Cache when the two halves of a tile run together).
// ---------------------------------------------------------------------------
constexpr int HB = NB / 2;         // bins per half
constexpr int TP3 = 256;           // pixels per workgroup tile (8 wavefronts x 32)

__device__ __forceinline__ void bh_split8x3(const float4 u, const float4 v, bh_bf16x8& h, bh_bf16x8& m, bh_bf16x8& l) {
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

__global__ __launch_bounds__(512, 2) void bin_head_split3_kernel(const float* __restrict__ feat, const float* __restrict__ Wf,
                                                                 const float* __restrict__ bout,
                                                                 const float* __restrict__ centers, float* __restrict__ part,
                                                                 long P, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __bf16* wfrag = reinterpret_cast<__bf16*>(lds);           // [4 bin tiles][8 K steps][h, m, l][64 lanes][8]
  float* bl = lds + (HB * CH * 3 * 2) / 4;                   // [128]
  float* cl = bl + HB;                                       // [128]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int b = blockIdx.z, half = blockIdx.y;
  const float* fb = feat + (long)b * CH * P;
  const float* wb = Wf + ((long)b * NB + half * HB) * CH;

  for (int idx = tid; idx < HB * CH / 8; idx += 512) {
    const int k = idx >> 4, o = idx & 15;                    // bin of the half, K octet
    bh_bf16x8 h, m, l;
    bh_split8x3(ld4(wb + (long)k * CH + 8 * o), ld4(wb + (long)k * CH + 8 * o + 4), h, m, l);
    __bf16* d = wfrag + ((((k >> 5) * 8 + (o >> 1)) * 3) * 64 + (o & 1) * 32 + (k & 31)) * 8;
    *reinterpret_cast<bh_bf16x8*>(d) = h;
    *reinterpret_cast<bh_bf16x8*>(d + 512) = m;
    *reinterpret_cast<bh_bf16x8*>(d + 1024) = l;
  }
  if (tid < HB) {
    bl[tid] = bout[half * HB + tid];
    cl[tid] = centers[(long)b * NB + half * HB + tid];
  }
  __syncthreads();

  float4 nxt[16];
  auto load_px = [&](float4 (&dst)[16], long pix) {
    const float* src = fb + (pix < P ? pix : 0) * CH + 8 * hh;        // rows past P re-read pixel 0 (never stored)
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      dst[2 * s] = ld4(src + 16 * s);
      dst[2 * s + 1] = ld4(src + 16 * s + 4);
    }
  };
  int tile = blockIdx.x;
  if (tile < ntiles) load_px(nxt, (long)tile * TP3 + wave * 32 + l31);
  for (; tile < ntiles; tile += gridDim.x) {
    const long pix = (long)tile * TP3 + wave * 32 + l31;
    bh_bf16x8 ph[8], pm[8], pl[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) bh_split8x3(nxt[2 * s], nxt[2 * s + 1], ph[s], pm[s], pl[s]);
    const int tn = tile + gridDim.x;
    if (tn < ntiles) load_px(nxt, (long)tn * TP3 + wave * 32 + l31);   // in flight under this tile's 192 MFMAs

    float m_run = -__builtin_inff(), l_half = 0.f, d_half = 0.f;
#pragma unroll 1
    for (int t = 0; t < HB / 32; ++t) {
      f32x16 acc = {0};
      const __bf16* wf = wfrag + (t * 8 * 3) * 512 + lane * 8;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const bh_bf16x8 ah = *reinterpret_cast<const bh_bf16x8*>(wf + s * 1536);
        const bh_bf16x8 am = *reinterpret_cast<const bh_bf16x8*>(wf + s * 1536 + 512);
        const bh_bf16x8 al = *reinterpret_cast<const bh_bf16x8*>(wf + s * 1536 + 1024);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, ph[s], acc, 0, 0, 0);      // small terms first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, pl[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, pm[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, ph[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, pm[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ph[s], acc, 0, 0, 0);
      }
      float tmax = -__builtin_inff();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[r] += bl[t * 32 + acc_row(r, hh)];
        tmax = fmaxf(tmax, acc[r]);
      }
      tmax = xor32_max(tmax);
      const float m_new = fmaxf(m_run, tmax);
      const float alpha = fast_exp(m_run - m_new);
      float ps = 0.f, ds = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pr = fast_exp(acc[r] - m_new);
        ps += pr;
        ds += pr * cl[t * 32 + acc_row(r, hh)];
      }
      l_half = l_half * alpha + ps;
      d_half = d_half * alpha + ds;
      m_run = m_new;
    }
    const float l = xor32_sum(l_half), d = xor32_sum(d_half);
    if (hh == 0 && pix < P) {
      float* o = part + (((long)b * 2 + half) * P + pix) * 4;
      *reinterpret_cast<float4*>(o) = make_float4(m_run, l, d, 0.f);
    }
  }
}

// ---------------------------------------------------------------------------
// TWO-term fp16 split bin head (NHWC map) -- round 3, the default.  v = hi + 2^-11 lo' with hi = fp16(v) and
// lo' = fp16((v - hi) 2^11): the residual is exact in fp32 and, scaled, lives in hi's binade (never in fp16's subnormals), so
// the pair carries 22 bits; a product block is THREE v_mfma_f32_32x32x16_f16 -- hi hi into acc1, hi lo' + lo' hi into acc2,
// logits = acc1 + 2^-11 acc2 -- with the error of an fp32 FMA chain (3e-7 of max |logit| on 128-long contractions; the
// three-term bf16 split: 1.5e-7 with six MFMAs; csrc/xattn_h2.hip has the derivation).  Two parts of Wf[b] are 128 KB: they
// FIT a CU's LDS, so one workgroup (8 wavefronts, two per SIMD: one's softmax arithmetic under the other's MFMAs) walks all
// 256 bins of its pixel tiles -- the map is read ONCE (the three-term form reads it once per bin half and merges the halves
// in a second launch) and every bin tile costs 24 MFMAs instead of 48.  fp16's range: a map value or folded weight beyond
// +-65504 turns the pixel's depth inf / NaN; OCV_BINHEAD=split3 / exact keep fp32's range.
// ---------------------------------------------------------------------------
typedef _Float16 bh_h16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void bh_split8_h2(const float4 u, const float4 v, bh_h16x8& hi, bh_h16x8& lo) {
  const float f[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const _Float16 h = (_Float16)f[i];
    hi[i] = h;
    lo[i] = (_Float16)((f[i] - (float)h) * 2048.0f);
  }
}

// Schedule (round 4).  A 32-bin tile is 24 MFMAs of 32 cycles (the wavefront is held at the matrix pipe's door for 768 cycles) and
// its softmax arithmetic.  The round-3 kernel ran the two one after the other, and its two wavefronts per SIMD IN STEP -- they leave
// the staging barrier together and do identical work -- so the SQ counters showed matrix pipe busy 46 % + VALU busy 53 % = the SUM
// of the two, not their overlap.  A wavefront issues in order: its vector work runs under its own MFMAs only when it sits BETWEEN
// them in program order.  So the loop over the eight tiles is software-pipelined with two accumulator sets: the 24 MFMAs of tile
// t + 1 are interleaved one by one with the softmax instructions of tile t (__builtin_amdgcn_sched_group_barrier pins 1 MFMA :
// 1 LDS read : 4 VALU per slot), the weight fragments are fetched two K steps ahead through three rotating register sets (a
// fourth set spills; the first two fetches of a tile are its own, covered by the interleaved softmax), and the VALU side is made
// to fit the slots: the logits carry a factor log2(e) (folded into the split weights and the bias when they are staged: the
// softmax is exp2 of differences), the bias is the hi*hi accumulator's INITIAL value (fetched into the accumulator as soon as the
// previous tile's values have been read), and the arithmetic is two-wide packed fp32 -- per tile 8 v_pk_fma + 16 v_pk_add + 8
// v_pk_fma, 16 v_exp_f32 and 8 v_max3 instead of 112 scalar operations + 16 v_exp_f32.  Result (profiles/r04_binhead_sq.txt):
// vector instructions 63 M -> 45 M, wave cycles -29 %, matrix pipe busy 46 -> 65 %, 0.287 -> 0.244 ms in the step (the launch is
// power-bound: the clock sinks as the pipes fill).  Measured on the way and dropped: two wavefront groups held half a tile apart
// by raw s_barriers, weights one step ahead (0.346 ms: with ONE wavefront per SIMD in its MFMA phase the LDS latency is exposed
// in every K step); the same stagger by one s_sleep with the weights two steps ahead (0.255 ms; MFMA / VALU co-execution 7 M
// cycles against 71 M before: the groups fall back in step).
//
// TWO-LEVEL logits (round 5, TWO_LEVEL = true, the default).  A bin whose logit lies T below the pixel's largest carries e^-T of the
// softmax: it needs no 22-bit logit -- it needs no exponential at all.  So the eight bin tiles are first formed COARSELY, the hi hi
// product alone (8 MFMAs a tile instead of 24, an 11-bit product: |coarse - exact| <= 2^-10 |w_k| |f_p| by Cauchy-Schwarz, with
// |w_k| <= the largest row norm of Wf[b], found while the weights are staged, and |f_p| the pixel's own norm, one dot product per
// lane), each lane keeps the largest coarse logit of its 16 bins per tile, and a tile gets its full three-product logits and its
// softmax arithmetic only if SOME lane of the wavefront has a bin within T_p = T + 2 eps_p of its pixel's coarse maximum (eps_p the
// bound above): then every skipped bin is provably >= T below the exact maximum, the skipped bins of a pixel together weigh
// <= 224 e^-T of its largest bin (T = 24: 8.5e-9), and the tile that holds a pixel's maximum is always kept.  A non-finite pixel
// or weight makes the test true for every tile (the comparison is written to fail on NaN): inf / NaN reach the output as in the
// one-level kernel.  Matrix work: 64 + 24 n MFMAs per 32 pixels, n = tiles kept (1 .. 8), against 192: on the benchmark's maps
// n = 1 (tools/exp_binhead_skip.py: one tile of eight per 32 pixels at T = 6 .. 14 alike) -- 46 %; every tile kept (a flat
// distribution over the bins): 133 %.  The coarse pass is as much LDS as matrix work (one 1 KB fragment read per MFMA = the LDS
// pipe's 128 B / clk when four SIMDs do it); it is not software-pipelined by hand: the compiler has two accumulator sets to
// alternate and only 8 v_max3 per tile to place.  OCV_BINHEAD=h2dense keeps the one-level kernel.
typedef float bh_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 bh_h16x2 __attribute__((ext_vector_type(2)));
constexpr float BH_SKIP_T = 24.0f * 1.44269504088896340736f;     // T in the kernel's base-2 logits

template <bool TWO_LEVEL>
__global__ __launch_bounds__(512, 2) void bin_head_h2_kernel(const float* __restrict__ feat, const float* __restrict__ Wf,
                                                             const float* __restrict__ bout,
                                                             const float* __restrict__ centers, float* __restrict__ depth,
                                                             long P, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  _Float16* wfrag = reinterpret_cast<_Float16*>(lds);       // [8 bin tiles][8 K steps][hi, lo'][64 lanes][8]
  float* bl = lds + (NB * CH * 2 * 2) / 4;                   // [256]  bias * log2(e)
  float* cl = bl + NB;                                       // [256]  bin centres
  constexpr float LOG2E = 1.44269504088896340736f;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int b = blockIdx.y;
  const float* fb = feat + (long)b * CH * P;
  const float* wb = Wf + (long)b * NB * CH;

  unsigned* wn2 = reinterpret_cast<unsigned*>(cl + NB);     // [1] the largest squared row norm of Wf[b] log2(e), as bits (TWO_LEVEL)
  if (TWO_LEVEL) {
    if (tid == 0) *wn2 = 0u;
    __syncthreads();
  }
  for (int idx = tid; idx < NB * CH / 8; idx += 512) {
    const int k = idx >> 4, o = idx & 15;                    // bin, K octet
    bh_h16x8 hi, lo;
    float4 u = ld4(wb + (long)k * CH + 8 * o), v = ld4(wb + (long)k * CH + 8 * o + 4);
    u.x *= LOG2E; u.y *= LOG2E; u.z *= LOG2E; u.w *= LOG2E;
    v.x *= LOG2E; v.y *= LOG2E; v.z *= LOG2E; v.w *= LOG2E;
    if (TWO_LEVEL) {                                         // 16 consecutive lanes hold one row: its squared norm, then the maximum
      float q = u.x * u.x + u.y * u.y + u.z * u.z + u.w * u.w + v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
#pragma unroll
      for (int d_ = 8; d_ > 0; d_ >>= 1) q += __shfl_xor(q, d_, 64);
      if (o == 0) atomicMax(wn2, __float_as_uint(q));        // (non-negative floats order as their bits; a NaN sits above them all)
    }
    bh_split8_h2(u, v, hi, lo);
    _Float16* d = wfrag + ((((k >> 5) * 8 + (o >> 1)) * 2) * 64 + (o & 1) * 32 + (k & 31)) * 8;
    *reinterpret_cast<bh_h16x8*>(d) = hi;
    *reinterpret_cast<bh_h16x8*>(d + 512) = lo;
  }
  if (tid < NB) {
    bl[tid] = bout[tid] * LOG2E;
    cl[tid] = centers[(long)b * NB + tid];
  }
  __syncthreads();

  float4 cur[16], nxt[16];
  auto load_px = [&](float4 (&dst)[16], long pix) {
    const float* src = fb + (pix < P ? pix : 0) * CH + 8 * hh;   // (a pixel beyond the map computes pixel 0 again and stores nothing)
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      dst[2 * s] = ld4(src + 16 * s);
      dst[2 * s + 1] = ld4(src + 16 * s + 4);
    }
  };
  int tile = blockIdx.x;
  if (tile < ntiles) load_px(cur, (long)tile * TP3 + wave * 32 + l31);
  for (; tile < ntiles; tile += gridDim.x) {
    const long pix = (long)tile * TP3 + wave * 32 + l31;
    const int tn = tile + gridDim.x;
    bh_h16x8 ph[8], pl[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) bh_split8_h2(cur[2 * s], cur[2 * s + 1], ph[s], pl[s]);
    __builtin_amdgcn_sched_barrier(0);                       // (the raw values are dead before the next tile's are asked for)
    if (tn < ntiles) load_px(nxt, (long)tn * TP3 + wave * 32 + l31);

    // weight fragments of (bin tile t, K step s): hi at wfrag + ((t 8 + s) 2) 512 + lane 8, lo' 512 halves further
    bh_h16x8 wh[3], wl[3];                                   // three rotating sets: K step s lives in set s % 3
    auto fetch_w = [&](int t, int s_) {
      const _Float16* wf = wfrag + ((t * 8 + s_) * 2) * 512 + lane * 8;
      wh[s_ % 3] = *reinterpret_cast<const bh_h16x8*>(wf);
      wl[s_ % 3] = *reinterpret_cast<const bh_h16x8*>(wf + 512);
    };
    auto fetch_bias = [&](int t, f32x16& a1) {               // accumulator register r = bin acc_row(r, hh): four runs of four
      const float* bp = bl + t * 32 + 4 * hh;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bp + 8 * g);
        a1[4 * g] = b4[0]; a1[4 * g + 1] = b4[1]; a1[4 * g + 2] = b4[2]; a1[4 * g + 3] = b4[3];
      }
    };
    // the 24 MFMAs of bin tile t: a1 (preloaded with the bias) += hi hi, a2 = 2^11 (lo hi + hi lo); weights two K steps ahead
    // (the first two steps' fetch is this tile's own: the softmax instructions interleaved with it cover their latency)
    auto mma_tile = [&](int t, f32x16& a1, f32x16& a2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) a2[r] = 0.f;
      fetch_w(t, 0); fetch_w(t, 1);
#pragma unroll
      for (int s_ = 0; s_ < 8; ++s_) {
        if (s_ + 2 < 8) fetch_w(t, s_ + 2);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[s_ % 3], ph[s_], a2, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s_ % 3], ph[s_], a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s_ % 3], pl[s_], a2, 0, 0, 0);
      }
    };
    float m_run = -__builtin_inff();                         // online softmax over this lane's 16 bins per tile, in base 2
    bh_f32x2 l2 = {0.f, 0.f}, d2 = {0.f, 0.f};               // (its pixel's other 16 bins sit in lane ^ 32)
    // softmax arithmetic of bin tile t; afterwards a1 holds the bias of tile tb (the accumulator's next initial value)
    auto softmax_tile = [&](int t, f32x16& a1, const f32x16& a2, int tb) {
      bh_f32x2 lg[8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        lg[i] = bh_f32x2{a2[2 * i], a2[2 * i + 1]} * (1.0f / 2048.0f) + bh_f32x2{a1[2 * i], a1[2 * i + 1]};
      fetch_bias(tb, a1);
      float tmax = fmaxf(lg[0].x, lg[0].y);
#pragma unroll
      for (int i = 1; i < 8; ++i) tmax = fmaxf(fmaxf(tmax, lg[i].x), lg[i].y);
      tmax = xor32_max(tmax);
      const float m_new = fmaxf(m_run, tmax);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      const float* cp = cl + t * 32 + 4 * hh;
      bh_f32x2 ps = {0.f, 0.f}, ds = {0.f, 0.f};
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 c4 = *reinterpret_cast<const f32x4*>(cp + 8 * g);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bh_f32x2 e = lg[2 * g + j] - bh_f32x2{m_new, m_new};
          const bh_f32x2 pr = {__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
          ps += pr;
          ds += pr * bh_f32x2{c4[2 * j], c4[2 * j + 1]};
        }
      }
      l2 = l2 * alpha + ps;
      d2 = d2 * alpha + ds;
      m_run = m_new;
    };
    // one MFMA, one LDS read, four VALU slots -- 24 times: the program order of a fused half step
#define BH_INTERLEAVE()                                                   \
  _Pragma("unroll") for (int q_ = 0; q_ < 24; ++q_) {                     \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                    \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                    \
    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                    \
  }
    f32x16 A1, A2, B1, B2;
    if constexpr (TWO_LEVEL) {
      // the pixel's squared norm (this lane's 64 channels, the other 64 in lane ^ 32) from the hi parts: inf / NaN if any is
      float fn2 = 0.f;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bh_h16x2 h2v = {ph[s][2 * e], ph[s][2 * e + 1]};
          fn2 = __builtin_amdgcn_fdot2(h2v, h2v, fn2, false);
        }
      }
      fn2 += __shfl_xor(fn2, 32, 64);
      // coarse logits: a1 (bias) += hi hi, the lane's largest of each tile is all that is kept.  The 64 MFMAs of the eight tiles are
      // ONE stream: their hi fragments come SIX steps ahead through six rotating registers (the one-level loop's hi and lo' sets)
      // -- one 1 KB LDS read per MFMA saturates the LDS pipe, and its queueing latency is what a wavefront would wait for
      auto lane_max = [&](const f32x16& a1) {
        float m = fmaxf(a1[0], a1[1]);
#pragma unroll
        for (int r = 2; r < 16; r += 2) m = fmaxf(fmaxf(m, a1[r]), a1[r + 1]);
        return m;
      };
      bh_h16x8 wq[6];
      auto fetch_q = [&](int j) { wq[j % 6] = *reinterpret_cast<const bh_h16x8*>(wfrag + (j * 2) * 512 + lane * 8); };
      float tmx[NB / 32];
#pragma unroll
      for (int j = 0; j < 6; ++j) fetch_q(j);
      fetch_bias(0, A1);
#pragma unroll
      for (int t = 0; t < NB / 32; ++t) {
        f32x16& acc = (t & 1) ? B1 : A1;
        f32x16& oth = (t & 1) ? A1 : B1;
        if (t + 1 < NB / 32) fetch_bias(t + 1, oth);
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) {
          const int j = t * 8 + s_;
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[j % 6], ph[s_], acc, 0, 0, 0);
          if (j + 6 < NB / 32 * 8) fetch_q(j + 6);
        }
        tmx[t] = lane_max(acc);
      }
      float pmax = tmx[0];
#pragma unroll
      for (int t = 1; t < NB / 32; ++t) pmax = fmaxf(pmax, tmx[t]);
      pmax = xor32_max(pmax);
      // T_p = T + 2 eps_p, eps_p = 2^-10 |w|max |f_p| (+ 1 % and 2^-6 for the roundings of the bound itself and of the accumulation)
      const float wn = __uint_as_float(*wn2);
      const float eps = __builtin_sqrtf(wn * fn2) * (1.01f / 1024.0f) + 0.015625f;
      const float floor_ = pmax - (BH_SKIP_T + 2.0f * eps);
      unsigned keep = 0;
#pragma unroll
      for (int t = 0; t < NB / 32; ++t)
        // (a lane maximum of -inf: fmaxf dropped NaN logits -- an inf feature times a zero weight beside -inf ones -- and such a tile
        //  must reach the exact pass like in the one-level kernel, where the NaN poisons the pixel's sums: keep it)
        keep |= (__builtin_amdgcn_ballot_w64(!(tmx[t] <= floor_) || tmx[t] == -__builtin_inff()) != 0ull ? 1u : 0u) << t;
      keep = __builtin_amdgcn_readfirstlane(keep);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
      for (int t = 0; t < NB / 32; ++t) {
        if (!((keep >> t) & 1u)) continue;
        fetch_bias(t, A1);
        mma_tile(t, A1, A2);
        softmax_tile(t, A1, A2, t);
      }
    } else {
    fetch_bias(0, A1);
    fetch_bias(1, B1);
    mma_tile(0, A1, A2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int t = 0; t < NB / 32 - 2; t += 2) {
      mma_tile(t + 1, B1, B2);
      softmax_tile(t, A1, A2, t + 2);
      BH_INTERLEAVE();
      __builtin_amdgcn_sched_barrier(0);
      mma_tile(t + 2, A1, A2);
      softmax_tile(t + 1, B1, B2, t + 3);
      BH_INTERLEAVE();
      __builtin_amdgcn_sched_barrier(0);
    }
    mma_tile(NB / 32 - 1, B1, B2);
    softmax_tile(NB / 32 - 2, A1, A2, 0);
    BH_INTERLEAVE();
    __builtin_amdgcn_sched_barrier(0);
    softmax_tile(NB / 32 - 1, B1, B2, 1);
    __builtin_amdgcn_sched_barrier(0);
    }
#undef BH_INTERLEAVE
    const float l = xor32_sum(l2.x + l2.y), d = xor32_sum(d2.x + d2.y);
    if (hh == 0 && pix < P) depth[(long)b * P + pix] = d / l;

    if (tn < ntiles) {
#pragma unroll
      for (int s = 0; s < 16; ++s) cur[s] = nxt[s];
    }
  }
}

// depth = (d0 e0 + d1 e1) / (l0 e0 + l1 e1),  e_i = exp(m_i - max(m0, m1)): the online-softmax merge of the two halves
__global__ __launch_bounds__(256) void bin_head_combine_kernel(const float* __restrict__ part, float* __restrict__ depth, long P,
                                                               long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;     // b * P + pixel
  if (i >= total) return;
  const long b = i / P, px = i - b * P;
  const float4 a = *reinterpret_cast<const float4*>(part + ((b * 2 + 0) * P + px) * 4);
  const float4 c = *reinterpret_cast<const float4*>(part + ((b * 2 + 1) * P + px) * 4);
  const float m = fmaxf(a.x, c.x);
  const float ea = fast_exp(a.x - m), ec = fast_exp(c.x - m);
  depth[i] = (a.z * ea + c.z * ec) / (a.y * ea + c.y * ec);
}

// ram[b][q][p] = sum_c queries[b][q][c] * feat[b][c][p]
template <bool NHWC>
__global__ __launch_bounds__(256) void pixel_dot_kernel(const float* __restrict__ feat, const float* __restrict__ qm,
                                                        long q_bs, int q_ld, float* __restrict__ ram, long P,
                                                        int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Ql = lds;                  // [128][129]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int b = blockIdx.y;
  const float* fb = feat + (long)b * CH * P;
  const float* qb = qm + (long)b * q_bs;
  float* rb = ram + (long)b * CH * P;

  for (int i = tid; i < CH * CH; i += 256) {
    const int q = i / CH, c = i % CH;
    Ql[q * WLD + c] = qb[(long)q * q_ld + c];
  }
  __syncthreads();

  float cur[CH / 2];
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long pix = (long)tile * TP + wave * 32 + l31;
    load_pixels<NHWC>(cur, fb, P, pix, hh, pix < P);
#pragma unroll 1
    for (int t = 0; t < CH / 32; ++t) {
      f32x16 acc = {0};
      const float* qrow = Ql + (t * 32 + l31) * WLD + a_col<NHWC>(0, hh);
#pragma unroll
      for (int s = 0; s < CH / 2; ++s) acc = mfma_32x32x2(qrow[a_col<NHWC>(s, 0)], cur[s], acc);
      if (pix < P) {
#pragma unroll
        for (int r = 0; r < 16; ++r) rb[(long)(t * 32 + acc_row(r, hh)) * P + pix] = acc[r];
      }
    }
  }
}

int blocks_per_image(int B, int ntiles) {
  // persistent grid: about one workgroup per CU (256), at least 1 and at most one per tile
  int per = (256 + B - 1) / B;
  if (per > ntiles) per = ntiles;
  if (per < 1) per = 1;
  return per;
}

}  // namespace

extern "C" size_t ocv_bin_head_workspace_bytes(int B, int n_bins, int C) {
  if (B < 1 || n_bins != NB || C != CH) return 0;
  return (size_t)B * NB * CH * sizeof(float);
}

extern "C" size_t ocv_bin_head_partials_bytes(int B, int P) {
  if (B < 1 || P < 1) return 0;
  return (size_t)B * 2 * P * 4 * sizeof(float);
}

extern "C" int ocv_pixel_dot_fwd(const float* feat, int channels_last, const float* queries, long q_bs, int q_ld,
                                 float* ram, int B, int C, int Q, int P, ocv_stream_t stream) {
  OCV_CHECK_ARG(feat && queries && ram, "ocv_pixel_dot_fwd: null pointer");
  OCV_CHECK_ARG(C == CH && Q == CH, "ocv_pixel_dot_fwd: C and Q must be %d (got %d, %d)", CH, C, Q);
  OCV_CHECK_ARG(B >= 1 && B <= 65535 && P >= 1 && q_ld >= C, "ocv_pixel_dot_fwd: bad sizes");
  OCV_CHECK_ARG(!channels_last || ocv_aligned16(feat), "ocv_pixel_dot_fwd: channels_last map must be 16-byte aligned");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)pixel_dot_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)pixel_dot_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  const int ntiles = ocv_cdiv(P, TP);
  int per = blocks_per_image(B, ntiles) * 2;      // 66 KiB LDS -> two workgroups per CU
  if (per > ntiles) per = ntiles;
  const size_t lds = (size_t)CH * WLD * sizeof(float);
  if (channels_last)
    hipLaunchKernelGGL(pixel_dot_kernel<true>, dim3(per, B), dim3(256), lds, (hipStream_t)stream, feat, queries, q_bs,
                       q_ld, ram, (long)P, ntiles);
  else
    hipLaunchKernelGGL(pixel_dot_kernel<false>, dim3(per, B), dim3(256), lds, (hipStream_t)stream, feat, queries, q_bs,
                       q_ld, ram, (long)P, ntiles);
  OCV_CHECK_LAUNCH("ocv_pixel_dot_fwd");
  return 0;
}

extern "C" int ocv_bin_head_fold_fwd(const float* queries, long q_bs, int q_ld, const float* Wout, float* Wf, int B,
                                     int C, int Q, int n_bins, ocv_stream_t stream) {
  OCV_CHECK_ARG(queries && Wout && Wf, "ocv_bin_head_fold_fwd: null pointer");
  OCV_CHECK_ARG(C == CH && Q == CH && n_bins == NB, "ocv_bin_head_fold_fwd: needs C = Q = %d, n_bins = %d", CH, NB);
  OCV_CHECK_ARG(B >= 1 && B <= 65535 && q_ld >= C, "ocv_bin_head_fold_fwd: bad sizes");
  // Wf[b] (256 x 128) = Wout (256 x 128q) . queries[b] (128q x 128c): W operand given in [K, N] layout
  return ocv_linear_fwd(Wout, Q, 0, queries, q_ld, q_bs, 1, nullptr, Wf, CH, (long)NB * CH, B, NB, CH, Q, OCV_ACT_NONE,
                        stream);
}

extern "C" int ocv_bin_head_folded_fwd(const float* feat, int channels_last, const float* Wf, const float* bout,
                                       const float* centers, float* depth, int B, int C, int n_bins, int P,
                                       ocv_stream_t stream) {
  OCV_CHECK_ARG(feat && Wf && bout && centers && depth, "ocv_bin_head_folded_fwd: null pointer");
  OCV_CHECK_ARG(C == CH && n_bins == NB, "ocv_bin_head_folded_fwd: needs C = %d, n_bins = %d", CH, NB);
  OCV_CHECK_ARG(B >= 1 && B <= 65535 && P >= 1 && ocv_aligned16(Wf), "ocv_bin_head_folded_fwd: bad sizes / alignment");
  OCV_CHECK_ARG(channels_last == 0 || channels_last == 1 || channels_last == 3 || channels_last == 4, "ocv_bin_head_folded_fwd: channels_last must be 0 (NCHW, exact fp32), 1 (NHWC, exact fp32), 3 (NHWC, two-term fp16) or 4 (the same with two-level logits); 2 (three-term bf16) needs ocv_bin_head_folded_ws_fwd and its partials buffer");
  OCV_CHECK_ARG(!channels_last || ocv_aligned16(feat), "ocv_bin_head_folded_fwd: channels_last map must be 16-byte aligned");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)bin_head_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)bin_head_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  if (channels_last == 3 || channels_last == 4) {            // two-term fp16 split, all 256 bins per workgroup; 4 = two-level logits (the default route)
    static bool attr3 = false;
    if (!attr3) {
      (void)hipFuncSetAttribute((const void*)bin_head_h2_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)bin_head_h2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr3 = true;
    }
    const int nt = ocv_cdiv(P, TP3);
    int perh = (256 + B - 1) / B;                            // persistent grid: about one workgroup per CU over the images
    if (perh > nt) perh = nt;
    if (perh < 1) perh = 1;
    const size_t ldsh = (size_t)NB * CH * 2 * 2 + 2 * NB * sizeof(float) + 16;
    if (channels_last == 4)
      hipLaunchKernelGGL(bin_head_h2_kernel<true>, dim3(perh, B), dim3(512), ldsh, (hipStream_t)stream, feat, Wf, bout, centers,
                         depth, (long)P, nt);
    else
      hipLaunchKernelGGL(bin_head_h2_kernel<false>, dim3(perh, B), dim3(512), ldsh, (hipStream_t)stream, feat, Wf, bout, centers,
                         depth, (long)P, nt);
    OCV_CHECK_LAUNCH("ocv_bin_head_folded_fwd(h2)");
    return 0;
  }
  const int ntiles = ocv_cdiv(P, TP);
  const int per = blocks_per_image(B, ntiles);
  const size_t lds = (size_t)(NB * WLD + 2 * NB) * sizeof(float);
  // NHWC maps: channels_last == 1 -> exact fp32 MFMA; NCHW maps (0): the same kernel reading planes
  if (channels_last)
    hipLaunchKernelGGL(bin_head_kernel<true>, dim3(per, B), dim3(256), lds, (hipStream_t)stream, feat, Wf, bout,
                       centers, depth, (long)P, ntiles);
  else
    hipLaunchKernelGGL(bin_head_kernel<false>, dim3(per, B), dim3(256), lds, (hipStream_t)stream, feat, Wf, bout,
                       centers, depth, (long)P, ntiles);
  OCV_CHECK_LAUNCH("ocv_bin_head_folded_fwd");
  return 0;
}

extern "C" int ocv_bin_head_folded_ws_fwd(const float* feat, int channels_last, const float* Wf, const float* bout,
                                          const float* centers, float* depth, int B, int C, int n_bins, int P, void* partials,
                                          size_t partials_bytes, ocv_stream_t stream) {
  if (channels_last != 2 || partials == nullptr)
    return ocv_bin_head_folded_fwd(feat, channels_last, Wf, bout, centers, depth, B, C, n_bins, P, stream);
  OCV_CHECK_ARG(feat && Wf && bout && centers && depth, "ocv_bin_head_folded_ws_fwd: null pointer");
  OCV_CHECK_ARG(C == CH && n_bins == NB, "ocv_bin_head_folded_ws_fwd: needs C = %d, n_bins = %d", CH, NB);
  OCV_CHECK_ARG(B >= 1 && B <= 65535 && P >= 1 && ocv_aligned16(Wf) && ocv_aligned16(feat) && ocv_aligned16(partials),
                "ocv_bin_head_folded_ws_fwd: bad sizes / alignment");
  OCV_CHECK_ARG(partials_bytes >= ocv_bin_head_partials_bytes(B, P), "ocv_bin_head_folded_ws_fwd: partials buffer too small");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)bin_head_split3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  const int ntiles = ocv_cdiv(P, TP3);
  int per = (256 + 2 * B - 1) / (2 * B);          // persistent grid: about one workgroup per CU over (halves x images)
  if (per > ntiles) per = ntiles;
  if (per < 1) per = 1;
  const size_t lds3 = (size_t)HB * CH * 3 * 2 + 2 * HB * sizeof(float);
  hipLaunchKernelGGL(bin_head_split3_kernel, dim3(per, 2, B), dim3(512), lds3, (hipStream_t)stream, feat, Wf, bout, centers,
                     (float*)partials, (long)P, ntiles);
  OCV_CHECK_LAUNCH("ocv_bin_head_folded_ws_fwd(halves)");
  const long total = (long)B * P;
  hipLaunchKernelGGL(bin_head_combine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)partials, depth, (long)P, total);
  OCV_CHECK_LAUNCH("ocv_bin_head_folded_ws_fwd(combine)");
  return 0;
}

extern "C" int ocv_bin_head_fwd(const float* feat, int channels_last, const float* queries, long q_bs, int q_ld, const float* Wout,
                                const float* bout, const float* centers, float* depth, int B, int C, int Q, int n_bins,
                                int P, void* workspace, size_t workspace_bytes, ocv_stream_t stream) {
  OCV_CHECK_ARG(feat && queries && Wout && bout && centers && depth && workspace, "ocv_bin_head_fwd: null pointer");
  OCV_CHECK_ARG(workspace_bytes >= ocv_bin_head_workspace_bytes(B, n_bins, C) && workspace_bytes > 0 &&
                    ocv_aligned16(workspace),
                "ocv_bin_head_fwd: workspace too small or misaligned (or unsupported C / n_bins)");
  float* Wf = (float*)workspace;
  int rc = ocv_bin_head_fold_fwd(queries, q_bs, q_ld, Wout, Wf, B, C, Q, n_bins, stream);
  if (rc != 0) return rc;
  return ocv_bin_head_folded_fwd(feat, channels_last, Wf, bout, centers, depth, B, C, n_bins, P, stream);
}
