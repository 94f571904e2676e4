// Pointwise (1x1) convolution of the EfficientNet-B5 MBConv stages on the bf16 matrix cores with fp32-level accuracy
// (row N1 of SURVEY.md section 8; the reference runs conv_pw / conv_pwl / conv_head through its hub backbone,
// modules/DenseFeatureExtractor.py:18-27,149).
//
//   y[m][co] = act( bias[co] + sum_ci x[m][ci] * gate[m / rows_per_image][ci] * W[co][ci] ) + residual[m][co]
//
// Same contract as ocv_pointwise_conv_nhwc_fwd (csrc/encoder_nhwc.hip) except that the weights arrive pre-split:
// W = w_hi + w_lo with w_hi = bf16(W), w_lo = bf16(W - w_hi), rows zero-padded to Kp = ceil16(Cin).  The activation
// rows are split the same way on the fly and every product is formed as hi*hi + hi*lo + lo*hi on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation (the dropped lo*lo term is 2^-18 relative): 3 x 32 cycles per
// 32 x 32 x 16 block against 8 x 64 cycles for v_mfma_f32_32x32x2_f32.  That is what the exact-fp32 kernels were
// bound by from stage 4 of the encoder on (19200 x 1056 -> 176: 47 TFLOP/s = 30 % of the fp32 MFMA peak, 7 GFLOP
// against 98 MB of traffic); with the split form every layer is back under its HBM / latency bound.
//
// Three kernels (dispatch and the measurements behind it: ocv_pointwise_conv_nhwc_split_fwd at the end of this file):
//   pw_rows_kernel<KS>       Cin <= 128, very many rows: a wavefront keeps its 32 rows (x gate), already split, in
//                            VGPRs in A-operand order and walks the output-channel tiles.  No LDS, no barrier; rows
//                            are read from HBM exactly once.
//   pw_stream_kernel<NTL,U>  <= 32 output channels, very many rows: a wavefront streams K for its 32 rows, U steps of
//                            loads in flight; nothing shared between wavefronts.
//   pw_tile_kernel<WN,WK,RT> everything else: workgroup tile (32 RT) rows x (32 WN) channels, WK wavefront groups
//                            split every 64 WK-wide K slab between them (summed through LDS in a fixed order at the
//                            end).  Rows are staged through LDS as split bf16 (converted ONCE per workgroup, coalesced
//                            256-byte row segments), double buffered, one barrier per slab; the next slab's global
//                            loads (rows and weights) are issued before the current slab's MFMAs.
// In all three the weights are read straight from L2 into the B operand as contiguous 1 KB fragments (w_frag below).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "common.hpp"
#include "../../include/objcavit_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct PSArgs {
  const float *x, *gate, *bias, *res;
  const __bf16* wp;                 // packed B-operand fragments, see w_frag()
  float* y;
  long M;
  int K, Kp, N, rows_per_image, act;
  __bf16* yhl;                      // nullable: hl32 split copy of the output (pad channels written as zero), for a consumer that
  int Cpo;                          // reads its rows by LDS-DMA (csrc/pointwise_hl.hip); Cpo = ceil32(N)
  int nbx, nby, col_major;          // tile kernel: row blocks, channel blocks, traversal order
  int ksplit;                       // tile kernel: > 1 = the K slabs are shared out over that many workgroups per tile (round 4: the
                                    // 1x1 layers of a batch of 1 - 2 are 10 - 40 tiles walking K = 1824 ... 3072 on an idle chip);
                                    // slice z writes its RAW partial tile to y + z * M * N (bias / act / residual-free epilogue:
                                    // pw_splitk_finish_kernel adds the slices in a fixed order)
};

__device__ __forceinline__ float act_fn(float v, int act) {
  switch (act) {
    case OCV_ACT_RELU: return fmaxf(v, 0.f);
    case OCV_ACT_LEAKY_RELU: return v > 0.f ? v : 0.01f * v;
    case OCV_ACT_SILU: return fast_silu(v);
    case OCV_ACT_SIGMOID: return fast_sigmoid(v);
    default: return v;
  }
}

// 8 consecutive floats (two 16-byte vectors) -> bf16 hi / lo octets
__device__ __forceinline__ void split8(const float4 u, const float4 v, bf16x8& hi, bf16x8& lo) {
  const float f[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 h = (__bf16)f[i];
    hi[i] = h;
    lo[i] = (__bf16)(f[i] - (float)h);
  }
}

__device__ __forceinline__ float4 mul4(float4 a, const float4 g) {
  a.x *= g.x; a.y *= g.y; a.z *= g.z; a.w *= g.w;
  return a;
}

__device__ __forceinline__ bf16x8 ldb8(const __bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ float4 lda4(const float* p) { return ld4(p); }
__device__ __forceinline__ float4 ldg4(const float* p) { return ld4(p); }

// Weight layout ("packed", built once on the host side of the C ABI): the B operand of v_mfma_f32_32x32x16_bf16 for
// channel tile jt (32 channels) and K step s (16 inputs) is ONE contiguous 1 KB fragment -- lane l = 32 hh + l31
// holds W[32 jt + l31][16 s + 8 hh .. + 7] -- hi fragment first, lo fragment right behind it:
//   wp[((jt * nsteps + s) * 2 + part) * 512 + lane * 8 + e]
// so a wavefront's weight load touches 8 consecutive cache lines instead of 32 scattered ones (with row-major [N][K]
// weights the per-line tag work of those loads, not bandwidth, was the largest single cost of the tile kernel).
__device__ __forceinline__ const __bf16* w_frag(const PSArgs& p, int jt, int s, int lane) {
  return p.wp + (((long)jt * (p.Kp >> 4) + s) * 2) * 512 + lane * 8;
}

__device__ __forceinline__ f32x16 mfma3(const bf16x8 ah, const bf16x8 al, const bf16x8 bh, const bf16x8 bl, f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
  return acc;
}

// bias + activation + residual + store of one 32 x 32 accumulator tile whose first row is m_base
__device__ __forceinline__ void store_tile(const PSArgs& p, const f32x16& acc, long m_base, int n, int hh) {
  if (n >= p.N) return;
  const float bv = p.bias != nullptr ? p.bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const long m = m_base + acc_row(r, hh);
    if (m < p.M) {
      float v = act_fn(acc[r] + bv, p.act);
      if (p.res != nullptr) v += p.res[m * p.N + n];
      p.y[m * p.N + n] = v;
    }
  }
}

// Coalesced epilogue.  Storing a 32 x 32 accumulator tile straight from its MFMA layout takes 16 four-byte store
// instructions of two 128-byte runs each and is store-issue-bound (the expand layers spent as long in their stores as
// in everything else: 24 -> 144 at 240 x 320, 269 us with and 135 us without them).  The tile is transposed through
// 4 KB of LDS owned by the wavefront and leaves as four 16-byte-per-lane stores of 8 rows x 128 B.  Lane l owns columns
// 4 (l & 7) .. +3 of rows (l >> 3) + 8 it.  Rows are UNPADDED (32 floats): the ds_write_b32 of a 32-lane half covers one
// row = 32 consecutive banks, and each 16-lane group of the ds_read_b128 ({0-3, 12-15, 20-27}, ...) covers whole 256-byte
// bank rows -- one LDS cycle per group; the 40-float rows of round 2 cost three (MI355X_MICROARCH.md, LDS table).
constexpr int TS = 32;                          // floats per LDS row of the transpose scratch
constexpr int TSCRATCH = 32 * TS;               // floats per wavefront

// The residual (skip connection) of a tile in that same lane order, fetched BEFORE the K loop: read in the epilogue
// its latency sat fully exposed at the end of every wavefront (240 -> 40 at 120 x 160: 51 of 166 us for 49 MB).
__device__ __forceinline__ void load_res(const PSArgs& p, float4 (&rq)[4], long m_base, int n0, int lane) {
  const int ncol = n0 + 4 * (lane & 7);
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const long m = m_base + (lane >> 3) + 8 * it;
    rq[it] = (p.res != nullptr && ncol < p.N && m < p.M) ? ld4(p.res + m * p.N + ncol) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// n0 = first column of the tile; rq = residual prefetched by load_res (or nullptr: fetched here, coalesced)
__device__ __forceinline__ void store_tile_lds(const PSArgs& p, const f32x16& acc, float* scratch, const float4* rq,
                                               long m_base, int n0, int lane) {
  const int l31 = lane & 31, hh = lane >> 5;
  if ((p.N & 3) != 0) {                          // ragged rows cannot take 16-byte stores
    store_tile(p, acc, m_base, n0 + l31, hh);
    return;
  }
  const int n = n0 + l31;
  const float bv = (p.bias != nullptr && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) scratch[acc_row(r, hh) * TS + l31] = act_fn(acc[r] + bv, p.act);
  const int c4 = lane & 7, ncol = n0 + 4 * c4;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = (lane >> 3) + 8 * it;
    const long m = m_base + row;
    float4 v = *reinterpret_cast<const float4*>(scratch + row * TS + 4 * c4);
    if (ncol < p.N && m < p.M) {
      if (rq != nullptr) {
        v.x += rq[it].x; v.y += rq[it].y; v.z += rq[it].z; v.w += rq[it].w;
      } else if (p.res != nullptr) {
        const float4 q = ld4(p.res + m * p.N + ncol);
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
      }
      *reinterpret_cast<float4*>(p.y + m * p.N + ncol) = v;
    } else {
      v = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (p.yhl != nullptr && ncol < p.Cpo && m < p.M) {
      typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
      const float f[4] = {v.x, v.y, v.z, v.w};
      bf16x4_t h, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const __bf16 hb = (__bf16)f[e];
        h[e] = hb;
        l[e] = (__bf16)(f[e] - (float)hb);
      }
      __bf16* d = p.yhl + m * 2 * p.Cpo + (ncol >> 5) * 64 + (ncol & 31);
      *reinterpret_cast<bf16x4_t*>(d) = h;
      *reinterpret_cast<bf16x4_t*>(d + 32) = l;
    }
  }
}

// ---------------------------------------------------------------------------
// Cin <= 128: rows resident in registers.  KS = 16-wide K steps held (2 / 4 / 8).
// ---------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(256) void pw_rows_kernel(PSArgs p) {
  __shared__ __attribute__((aligned(16))) float tscratch[4 * TSCRATCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* scratch = tscratch + wave * TSCRATCH;
  const int l31 = lane & 31, hh = lane >> 5;
  const long m_base = (long)blockIdx.x * 128 + wave * 32;
  const int K = p.K;
  bf16x8 ahi[KS], alo[KS];
  {
    const long m = m_base + l31;
    const bool ok = m < p.M;
    const float* src = p.x + (ok ? m : 0) * K + 8 * hh;
    const float* gsrc = p.gate != nullptr ? p.gate + ((ok ? m : 0) / p.rows_per_image) * K + 8 * hh : nullptr;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      float4 u = make_float4(0.f, 0.f, 0.f, 0.f), v = u;
      if (ok && 16 * s + 8 * hh < K) {
        u = lda4(src + 16 * s);
        v = lda4(src + 16 * s + 4);
        if (gsrc != nullptr) {
          u = mul4(u, ldg4(gsrc + 16 * s));
          v = mul4(v, ldg4(gsrc + 16 * s + 4));
        }
      }
      split8(u, v, ahi[s], alo[s]);
    }
  }
  // blockIdx.y splits the channel tiles when there are too few 128-row workgroups to fill the chip
  const int ntiles_all = (p.N + 31) >> 5;
  const int per_y = (ntiles_all + gridDim.y - 1) / gridDim.y;
  const int nt_lo = blockIdx.y * per_y, nt_hi = min(ntiles_all, nt_lo + per_y);
  for (int nt = nt_lo; nt < nt_hi; ++nt) {
    const int n = nt * 32 + l31;
    const __bf16* wf = w_frag(p, nt, 0, lane);
    f32x16 acc = {0};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (16 * s < p.Kp) acc = mfma3(ahi[s], alo[s], ldb8(wf + s * 1024), ldb8(wf + s * 1024 + 512), acc);
    }
    store_tile_lds(p, acc, scratch, nullptr, m_base, nt * 32, lane);
  }
}

// ---------------------------------------------------------------------------
// general tile kernel
// ---------------------------------------------------------------------------
constexpr int LROW = 144;            // bytes per LDS row: 64 bf16 + 16 pad (conflict-free ds_read_b128 across rows)

template <int WN, int WK, int RT>
__global__ __launch_bounds__(64 * WN * WK) void pw_tile_kernel(PSArgs p) {
  constexpr int NT = 64 * WN * WK;             // threads
  constexpr int ROWS = 32 * RT;
  constexpr int KI = 64 * WK;                  // K slab per iteration
  constexpr int OCT = (ROWS * 8 * WK) / NT;    // float octets staged per thread and slab (= 4 RT / WN)
  constexpr int PART = ROWS * LROW;            // one (group, hi|lo) plane
  constexpr int BUF = WK * 2 * PART;
  static_assert(OCT >= 1 && OCT * NT == ROWS * 8 * WK, "staging map");
  extern __shared__ __attribute__((aligned(16))) char lds[];    // 2 x BUF

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int wn = wave % WN, g = wave / WN;
  // XCD-aware, bijective workgroup -> tile map: hardware deals consecutive workgroup ids round-robin to the 8 XCDs
  // (each with its own 4 MB L2); give every XCD a CONTIGUOUS run of tiles in the order the host chose (channel-block
  // major when the whole weight matrix would not fit an L2, so an XCD keeps one weight slice resident and streams
  // rows; row-block major otherwise, so its rows are read once and all of W stays resident).
  int bx, by, kz = 0;
  {
    const int nwg = gridDim.x;
    int wg = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    if (p.ksplit > 1) {                                     // K slice fastest: the slices of a tile run side by side
      kz = wg % p.ksplit;
      wg /= p.ksplit;
    }
    if (p.col_major) { by = wg / p.nbx; bx = wg % p.nbx; }
    else { bx = wg / p.nby; by = wg % p.nby; }
  }
  if (p.ksplit > 1) p.y += (long)kz * p.M * p.N;
  const long m0 = (long)bx * ROWS;
  const int n = by * (32 * WN) + wn * 32 + l31;
  const int ntl_all = (p.N + 31) >> 5;
  const __bf16* wf = w_frag(p, min(by * WN + wn, ntl_all - 1), 4 * g, lane);      // + (it * KI / 16 + s) * 1024
  const int K = p.K, Kp = p.Kp;
  const int nit_all = (Kp + KI - 1) / KI;
  const int it0 = p.ksplit > 1 ? nit_all * kz / p.ksplit : 0;           // this workgroup's K slabs [it0, nit)
  const int nit = p.ksplit > 1 ? nit_all * (kz + 1) / p.ksplit : nit_all;

  // staging map: octet o -> (row, group, j): 8 floats at k = it * KI + 64 group + 8 j of row m0 + row
  const float* asrc[OCT];
  const float* gsrc[OCT];
  int akoff[OCT], ldst[OCT];
#pragma unroll
  for (int i = 0; i < OCT; ++i) {
    const int o = tid + i * NT;
    const int row = o / (8 * WK), oc = o % (8 * WK);
    const long m = m0 + row;
    const bool ok = m < p.M;
    akoff[i] = ok ? 8 * oc : (1 << 30);                        // rows past M never pass the k < K test
    asrc[i] = p.x + (ok ? m : 0) * K + 8 * oc;
    gsrc[i] = p.gate != nullptr ? p.gate + ((ok ? m : 0) / p.rows_per_image) * K + 8 * oc : nullptr;
    ldst[i] = ((oc >> 3) * 2) * PART + row * LROW + (oc & 7) * 16;
  }
  float4 ra[OCT][2];
  bf16x8 wh[4], wl[4], whn[4], wln[4];
  auto load_a = [&](int it) {
#pragma unroll
    for (int i = 0; i < OCT; ++i) {
      const int k = it * KI + akoff[i];
      ra[i][0] = ra[i][1] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < K && k >= 0) {
        ra[i][0] = lda4(asrc[i] + it * KI);
        ra[i][1] = lda4(asrc[i] + it * KI + 4);
        if (gsrc[i] != nullptr) {
          ra[i][0] = mul4(ra[i][0], ldg4(gsrc[i] + it * KI));
          ra[i][1] = mul4(ra[i][1], ldg4(gsrc[i] + it * KI + 4));
        }
      }
    }
  };
  auto store_a = [&](int buf) {
#pragma unroll
    for (int i = 0; i < OCT; ++i) {
      bf16x8 h, l;
      split8(ra[i][0], ra[i][1], h, l);
      char* d = lds + buf * BUF + ldst[i];
      *reinterpret_cast<bf16x8*>(d) = h;
      *reinterpret_cast<bf16x8*>(d + PART) = l;
    }
  };
  auto load_w = [&](int it, bf16x8 (&h)[4], bf16x8 (&l)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int k = it * KI + 64 * g + 16 * s;
      if (k < Kp) {
        h[s] = ldb8(wf + (long)(it * (KI / 16) + s) * 1024);
        l[s] = ldb8(wf + (long)(it * (KI / 16) + s) * 1024 + 512);
      }
    }
  };

  f32x16 acc[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x16{0};
  // single-K-group tiles fetch the residual up front (see load_res); with two K groups that measured slower
  // (1824 -> 304: 53 vs 45 us) and the epilogue loads it
  constexpr bool RES_EARLY = WK == 1;
  const int ntile0 = by * (32 * WN) + wn * 32;              // first column of this wavefront's tile
  float4 resq[RES_EARLY ? RT : 1][4];

  auto multiply = [&](int it) {
    const char* base = lds + (it & 1) * BUF + (g * 2) * PART + l31 * LROW + hh * 16;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (it * KI + 64 * g + 16 * s < Kp) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const bf16x8 ah = *reinterpret_cast<const bf16x8*>(base + rt * 32 * LROW + s * 32);
          const bf16x8 al = *reinterpret_cast<const bf16x8*>(base + PART + rt * 32 * LROW + s * 32);
          acc[rt] = mfma3(ah, al, wh[s], wl[s], acc[rt]);
        }
      }
    }
  };
  load_a(it0);
  load_w(it0, wh, wl);
  store_a(it0 & 1);
  __syncthreads();
  for (int it = it0; it + 1 < nit; ++it) {
    load_a(it + 1);
    load_w(it + 1, whn, wln);
    multiply(it);
    store_a((it + 1) & 1);
#pragma unroll
    for (int s = 0; s < 4; ++s) { wh[s] = whn[s]; wl[s] = wln[s]; }
    __syncthreads();
  }
  // last slab, peeled: no prefetch registers are live any more, so the residual (see load_res) is fetched here, in
  // front of the last MFMAs, without raising the register count of the loop (it cost one wavefront per SIMD there)
  if (RES_EARLY) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) load_res(p, resq[rt], m0 + rt * 32, ntile0, lane);
  }
  multiply(nit - 1);
  __syncthreads();
  // K groups 1.. hand their partial tiles to group 0 through LDS ([group - 1][wn][rt][reg][lane], fixed order)
  if (WK > 1) {
    float* red = reinterpret_cast<float*>(lds);
    if (g > 0) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((((g - 1) * WN + wn) * RT + rt) * 16 + r) * 64 + lane] = acc[rt][r];
    }
    __syncthreads();
    if (g > 0) return;
#pragma unroll
    for (int gg = 1; gg < WK; ++gg)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rt][r] += red[((((gg - 1) * WN + wn) * RT + rt) * 16 + r) * 64 + lane];
  }
  // transpose scratch: the staging buffers are idle now (every wavefront is past the last barrier of the K loop); with
  // K groups it sits behind the reduction slabs, which the other wavefronts of group 0 may still be reading
  constexpr int RED_FLOATS = (WK - 1) * WN * RT * 16 * 64;
  float* scratch = reinterpret_cast<float*>(lds) + RED_FLOATS + wn * TSCRATCH;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
    store_tile_lds(p, acc[rt], scratch, RES_EARLY ? resq[rt] : nullptr, m0 + rt * 32, ntile0, lane);
}

template <int WN, int WK, int RT>
int launch_tile(const PSArgs& a, hipStream_t st) {
  constexpr int ROWS = 32 * RT;
  constexpr size_t STAGE = (size_t)2 * WK * 2 * ROWS * LROW;
  constexpr size_t TRANS = ((size_t)(WK - 1) * WN * RT * 16 * 64 + (size_t)WN * TSCRATCH) * sizeof(float);   // reduction slabs + scratch
  constexpr size_t LDS = STAGE > TRANS ? STAGE : TRANS;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)pw_tile_kernel<WN, WK, RT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  PSArgs b = a;
  b.nbx = (int)((a.M + ROWS - 1) / ROWS);
  b.nby = ocv_cdiv(a.N, 32 * WN);
  b.col_major = (long)a.N * a.Kp * 4 > (3L << 20);
  const int ks = a.ksplit > 1 ? a.ksplit : 1;
  hipLaunchKernelGGL((pw_tile_kernel<WN, WK, RT>), dim3((unsigned)((long)b.nbx * b.nby * ks)), dim3(64 * WN * WK), LDS, st, b);
  OCV_CHECK_LAUNCH("ocv_pointwise_conv_nhwc_split_fwd");
  return 0;
}

template <int WN>
int launch_tile_wn(const PSArgs& a, int wk, hipStream_t st) {
  if (wk >= 4) return launch_tile<WN, 4, 1>(a, st);
  return wk >= 2 ? launch_tile<WN, 2, 1>(a, st) : launch_tile<WN, 1, 1>(a, st);
}

// Second pass of a split-K 1x1 layer: y = act(sum_z part[z] + bias) + residual, slices added in ascending order.
struct PwFinArgs {
  const float *part, *bias, *res;
  float* y;
  long M, items;         // items = M * N / 4
  int N, act, ksplit;
};

__global__ __launch_bounds__(256) void pw_splitk_finish_kernel(PwFinArgs p) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.items) return;
  const long o = 4 * i;
  const int n = (int)(o % p.N);
  float4 a = ld4(p.part + o);
  for (int z = 1; z < p.ksplit; ++z) {
    const float4 u = ld4(p.part + (long)z * p.M * p.N + o);
    a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
  }
  const float4 bv = p.bias != nullptr ? ld4(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  a.x = act_fn(a.x + bv.x, p.act); a.y = act_fn(a.y + bv.y, p.act); a.z = act_fn(a.z + bv.z, p.act); a.w = act_fn(a.w + bv.w, p.act);
  if (p.res != nullptr) {
    const float4 q = ld4(p.res + o);
    a.x += q.x; a.y += q.y; a.z += q.z; a.w += q.w;
  }
  *reinterpret_cast<float4*>(p.y + o) = a;
}

// K slices per tile for a tile-kernel launch of `tiles` workgroups over Kp inputs: only where the launch would leave most of the
// chip idle (fewer than 128 tiles on 256 CUs) and the chain is long (a batch of 1 - 2 in B5's stages 5 - 7).  1 = no split.
int pw_ksplit(long tiles, int Kp, int Cout) {
  if ((Cout & 3) != 0) return 1;
  const int slabs = Kp / 128;                                  // the split launch runs two K groups per workgroup: 128-wide slabs
  if (tiles >= 128 || Kp < 1024) return 1;
  int ks = (int)(256 / tiles);
  if (ks > 8) ks = 8;
  if (ks > slabs / 2) ks = slabs / 2;                          // at least two slabs per slice
  return ks > 1 ? ks : 1;
}

// ---------------------------------------------------------------------------
// few output channels (<= 128), many rows: K-streaming rows kernel.  A wavefront owns 32 rows and ALL output channels
// (NTL accumulator tiles), reads its rows straight from HBM in A-operand order (lane = (row, 8 consecutive floats) --
// every 32-byte sector is fetched exactly once), splits them in registers and streams the matching weight columns from
// L1/L2.  No LDS, no barrier, nothing shared between wavefronts: the memory system sees one long independent stream
// per wavefront, which is what the HBM-bound project layers (240 -> 40 at 120 x 160: 393 MB, 6 GFLOP) want.
// ---------------------------------------------------------------------------
template <int NTL, int U, bool RES>
__global__ __launch_bounds__(256) void pw_stream_kernel(PSArgs p) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const long m_base = (long)blockIdx.x * 128 + wave * 32;
  const int K = p.K, nsteps = p.Kp >> 4;
  const long m = m_base + l31;
  const bool ok = m < p.M;
  const float* src = p.x + (ok ? m : 0) * K + 8 * hh;
  const float* gsrc = p.gate != nullptr ? p.gate + ((ok ? m : 0) / p.rows_per_image) * K + 8 * hh : nullptr;
  const int klim = ok ? K - 8 * hh : 0;        // this lane's octet of step s is in range iff 16 s < klim
  const int ntl_all = (p.N + 31) >> 5;
  const __bf16* wf[NTL];
#pragma unroll
  for (int j = 0; j < NTL; ++j) wf[j] = w_frag(p, min(j, ntl_all - 1), 0, lane);
  f32x16 acc[NTL];
#pragma unroll
  for (int j = 0; j < NTL; ++j) acc[j] = f32x16{0};
  __shared__ __attribute__((aligned(16))) float tscratch[4 * TSCRATCH];
  float* scratch = tscratch + wave * TSCRATCH;
  float4 resq[RES ? NTL : 1][4];
  if (RES) {
#pragma unroll
    for (int j = 0; j < NTL; ++j) load_res(p, resq[j], m_base, 32 * j, lane);
  }

  for (int s0 = 0; s0 < nsteps; s0 += U) {
    float4 ra[U][2];
    bf16x8 wh[U][NTL], wl[U][NTL];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int s = s0 + u;
      ra[u][0] = ra[u][1] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (16 * s < klim) {
        ra[u][0] = lda4(src + 16 * s);
        ra[u][1] = lda4(src + 16 * s + 4);
      }
      if (s < nsteps) {
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
          wh[u][j] = ldb8(wf[j] + (long)s * 1024);
          wl[u][j] = ldb8(wf[j] + (long)s * 1024 + 512);
        }
      }
    }
    if (gsrc != nullptr) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int s = s0 + u;
        if (16 * s < klim) {
          ra[u][0] = mul4(ra[u][0], ldg4(gsrc + 16 * s));
          ra[u][1] = mul4(ra[u][1], ldg4(gsrc + 16 * s + 4));
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (s0 + u < nsteps) {
        bf16x8 ah, al;
        split8(ra[u][0], ra[u][1], ah, al);
#pragma unroll
        for (int j = 0; j < NTL; ++j) acc[j] = mfma3(ah, al, wh[u][j], wl[u][j], acc[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NTL; ++j) store_tile_lds(p, acc[j], scratch, RES ? resq[j] : nullptr, m_base, 32 * j, lane);
}

// diagnostic override of the dispatch (ocv_pointwise_split_set_dispatch: tests, tools): family 1 = rows, 2 = stream, 3 = tile (with wn, wk)
struct PwCfg { int wn = 0, wk = 0, family = 0; };
PwCfg& pw_cfg() {
  static PwCfg c;
  return c;
}

}  // namespace

extern "C" int ocv_pointwise_split_set_dispatch(int family, int a, int b) {
  OCV_CHECK_ARG(family >= 0 && family <= 3, "ocv_pointwise_split_set_dispatch: family must be 0 (automatic) .. 3");
  PwCfg& c = pw_cfg();
  c.family = family;
  c.wn = a;
  c.wk = b;
  return 0;
}

extern "C" size_t ocv_pointwise_packed_weight_elems(int Cin, int Cout) {
  if (Cin < 1 || Cout < 1) return 0;
  return (size_t)((Cout + 31) / 32) * ((Cin + 15) / 16) * 2 * 512;
}

extern "C" int ocv_pointwise_conv_nhwc_split_fwd(const float* x, const float* gate, int rows_per_image,
                                                 const void* w_packed, const float* bias, const float* residual,
                                                 float* y, long M, int Cin, int Cout, int act, ocv_stream_t stream) {
  return ocv_pointwise_conv_nhwc_split_hl_fwd(x, gate, rows_per_image, w_packed, bias, residual, y, nullptr, M, Cin, Cout, act, stream);
}

extern "C" int ocv_pointwise_conv_nhwc_split_hl_fwd(const float* x, const float* gate, int rows_per_image,
                                                    const void* w_packed, const float* bias, const float* residual,
                                                    float* y, void* y_hl, long M, int Cin, int Cout, int act, ocv_stream_t stream) {
  return ocv_pointwise_conv_nhwc_split_ws_fwd(x, gate, rows_per_image, w_packed, bias, residual, y, y_hl, M, Cin, Cout, act, nullptr, 0,
                                              stream);
}

// tiles of the 32-row kernel for (M, Cout) and the K slices the dispatcher would use
static int pw_plan_ksplit(long M, int Cin, int Cout) {
  const int Kp = (Cin + 15) / 16 * 16;
  const int wn = Cout > 64 ? 4 : 2;
  const long tiles = ((M + 31) / 32) * ocv_cdiv(Cout, 32 * wn);
  return pw_ksplit(tiles, Kp, Cout);
}

extern "C" size_t ocv_pointwise_split_workspace_bytes(long M, int Cin, int Cout) {
  if (M < 1 || Cin < 8 || Cout < 1) return 0;
  const PwCfg& cfg = pw_cfg();
  if (cfg.family != 0 && cfg.family != 3) return 0;
  if ((Cout <= 32 && M >= 65536) || (Cin <= 128 && M >= 200000)) return 0;       // rows / stream kernels: no K split
  const int ks = pw_plan_ksplit(M, Cin, Cout);
  return ks > 1 ? (size_t)ks * M * Cout * sizeof(float) : 0;
}

extern "C" int ocv_pointwise_conv_nhwc_split_ws_fwd(const float* x, const float* gate, int rows_per_image,
                                                    const void* w_packed, const float* bias, const float* residual,
                                                    float* y, void* y_hl, long M, int Cin, int Cout, int act, void* workspace,
                                                    size_t workspace_bytes, ocv_stream_t stream) {
  const int Kp = (Cin + 15) / 16 * 16;
  OCV_CHECK_ARG(y_hl == nullptr || (Cout % 8 == 0 && ocv_aligned16(y_hl)), "ocv_pointwise_conv_nhwc_split_hl_fwd: the split output needs Cout to be a multiple of 8 (got %d) and 16-byte alignment", Cout);
  OCV_CHECK_ARG(x && w_packed && y, "ocv_pointwise_conv_nhwc_split_fwd: null pointer");
  OCV_CHECK_ARG(M >= 0 && Cin >= 8 && Cin % 8 == 0 && Cout >= 1, "ocv_pointwise_conv_nhwc_split_fwd: Cin must be a positive multiple of 8 (got M=%ld Cin=%d Cout=%d)", M, Cin, Cout);
  OCV_CHECK_ARG(gate == nullptr || rows_per_image >= 1, "ocv_pointwise_conv_nhwc_split_fwd: gate needs rows_per_image");
  OCV_CHECK_ARG(act >= 0 && act <= OCV_ACT_SIGMOID, "ocv_pointwise_conv_nhwc_split_fwd: unknown activation %d", act);
  OCV_CHECK_ARG(ocv_aligned16(x) && ocv_aligned16(w_packed) && ocv_aligned16(gate) && ocv_aligned16(y) && ocv_aligned16(residual),
                "ocv_pointwise_conv_nhwc_split_fwd: x / w_packed / gate / y / residual must be 16-byte aligned");
  if (M == 0) return 0;
  PSArgs a{x, gate, bias, residual, (const __bf16*)w_packed, y, M, Cin, Kp, Cout,
           rows_per_image > 0 ? rows_per_image : 1, act, (__bf16*)y_hl, (Cout + 31) / 32 * 32};
  hipStream_t st = (hipStream_t)stream;
  // Dispatch (measured on MI355X, bs = 16 encoder shapes, tools/history/run_pw.py):
  //   rows   : Cin <= 128 and >= 2 10^5 rows (the stage 1-2 expand layers: one pass over the rows, channels walked)
  //   stream : <= 32 output channels and many rows (stage-1 project layers: pure row stream)
  //   tile   : everything else; 32 rows x 128 (64) channels per workgroup -- the smallest tile won on every late-stage
  //            shape because these launches are latency-bound and it keeps the most workgroups in flight -- with two
  //            K groups when the launch would otherwise have fewer than 512 workgroups
  const PwCfg& cfg = pw_cfg();
  int family = cfg.family;
  if (family == 0) family = (Cout <= 32 && M >= 65536) ? 2 : ((Cin <= 128 && M >= 200000) ? 1 : 3);
  if (family == 1 && Cin > 128) family = 3;
  if (family == 2 && Cout > 32) family = 3;
  if (family == 1) {
    const long mblocks = (M + 127) / 128;
    int ysplit = 1;
    const int ntl = (Cout + 31) / 32;
    while (mblocks * ysplit < 768 && ysplit * 2 <= ntl) ysplit *= 2;
    const dim3 grid((unsigned)mblocks, ysplit);
    if (Cin <= 32) hipLaunchKernelGGL(pw_rows_kernel<2>, grid, dim3(256), 0, st, a);
    else if (Cin <= 64) hipLaunchKernelGGL(pw_rows_kernel<4>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(pw_rows_kernel<8>, grid, dim3(256), 0, st, a);
    OCV_CHECK_LAUNCH("ocv_pointwise_conv_nhwc_split_fwd(rows)");
    return 0;
  }
  if (family == 2) {
    const dim3 grid((unsigned)((M + 127) / 128));
    if (residual != nullptr) hipLaunchKernelGGL((pw_stream_kernel<1, 4, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((pw_stream_kernel<1, 4, false>), grid, dim3(256), 0, st, a);
    OCV_CHECK_LAUNCH("ocv_pointwise_conv_nhwc_split_fwd(stream)");
    return 0;
  }
  int wn = Cout > 64 ? 4 : 2;
  if (cfg.wn == 2 || cfg.wn == 4) wn = cfg.wn;
  const long wgs = ((M + 31) / 32) * ocv_cdiv(Cout, 32 * wn);
  // a batch of 1 - 2 in the late stages: a handful of tiles, each a long K chain on an idle chip -> the K slabs of a tile go to
  // several workgroups (raw partial tiles into the caller's workspace) and a second launch adds them in a fixed order
  const int ks = (cfg.wn == 0 && cfg.wk == 0 && y_hl == nullptr && workspace != nullptr) ? pw_plan_ksplit(M, Cin, Cout) : 1;
  if (ks > 1 && workspace_bytes >= (size_t)ks * M * Cout * sizeof(float) && ocv_aligned16(workspace)) {
    PSArgs h = a;
    h.bias = nullptr; h.res = nullptr; h.act = OCV_ACT_NONE; h.yhl = nullptr; h.y = (float*)workspace; h.ksplit = ks;
    const int rc = wn == 4 ? launch_tile_wn<4>(h, 2, st) : launch_tile_wn<2>(h, 2, st);
    if (rc != 0) return rc;
    PwFinArgs f{(const float*)workspace, bias, residual, y, M, M * Cout / 4, Cout, act, ks};
    hipLaunchKernelGGL(pw_splitk_finish_kernel, dim3((unsigned)((f.items + 255) / 256)), dim3(256), 0, st, f);
    OCV_CHECK_LAUNCH("ocv_pointwise_conv_nhwc_split_fwd(finish)");
    return 0;
  }
  // (round 4, the reference's own batch of 1 - 2: a stage 6 - 7 layer is then 10 - 40 workgroups walking K = 1824 ... 3072 in a
  // chain of 14 - 24 slabs, ~1 us each, on a chip that is otherwise idle: FOUR K groups per workgroup halve the chain)
  int wk = (wgs < 512 && Kp >= 512) ? ((wgs < 128 && Kp >= 1024) ? 4 : 2) : 1;
  if (cfg.wk) wk = cfg.wk;
  return wn == 4 ? launch_tile_wn<4>(a, wk, st) : launch_tile_wn<2>(a, wk, st);
}
