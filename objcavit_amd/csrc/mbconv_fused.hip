// Expand 1x1 (+BN+SiLU) and depthwise k x k (+BN+SiLU, + squeeze-excite pooling partials) of an EfficientNet MBConv
// block in ONE kernel, NHWC fp32, for the early encoder stages (row N1 of SURVEY.md section 8; conv_pw/bn1/act1 ->
// conv_dw/bn2/act2 of the InvertedResidual blocks the reference runs through its hub backbone,
// modules/DenseFeatureExtractor.py:18-27).
//
// Why: the 6x-expanded tensor is the encoder's dominant HBM traffic -- as two launches it is written by the expand layer
// and read back by the depthwise layer (stage 2 at 120 x 160, 40 -> 240 channels, B = 16: 344 MB + 590 MB moved in
// 123 + 150 us).  Fused, the expanded values only ever exist in LDS: the block reads x (49 MB) and writes the depthwise
// output (295 MB).
//
// Work item = (spatial tile of TH x TW outputs, chunk of 32 expanded channels); all chunks of a tile run on one XCD,
// back to back, so the x tile they share comes from HBM once.
//   phase A  the input pixels under the tile (IH x IW halo, (TH-1) S + K by (TW-1) S + K) are a GEMM
//            [P pixels x Cin] x [Cin x 32] on v_mfma_f32_32x32x16_bf16, split-bf16 (hi*hi + hi*lo + lo*hi, fp32
//            accumulate: the numerics of csrc/pointwise_split.hip, same packed weights).  A lane loads 8 consecutive
//            input channels of its pixel straight into A-operand order; the chunk's weight fragments stay in VGPRs for
//            all M tiles.  bias + SiLU, pixels outside the image forced to ZERO (the depthwise convolution pads the
//            EXPANDED tensor), result to LDS as [pixel][32 channels] (128 B per pixel: two pixels fill the 64 banks).
//   phase B  thread (channel quad, output column, row group) slides down the halo rows: K ds_read_b128 per input row,
//            K x K x 4 FMAs per row into the NR running outputs; depthwise weights of the quad in VGPRs (fetched before
//            phase A).  bias + SiLU, 16-byte stores (128 B per pixel and chunk), and the tile's per-channel sum for the
//            squeeze-excite mean through LDS in a fixed order (no atomics).
// For stride 2 the halo columns are stored even-columns-first so that neighbouring output columns read neighbouring
// LDS slots (otherwise every read would be a 2-way bank conflict).
// Halo recompute of the GEMM: 340 / 256 pixels (k = 3), 432 / 256 (k = 5) -- the GEMM is a few percent of the block's
// time at Cin <= 64, which is what this kernel is limited to (A fragments of a pixel tile fully in VGPRs).
#include "common.hpp"
#include "../../include/objcavit_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct FXArgs {
  const float *x, *be, *wd, *bd;
  const __bf16* wp;                 // packed expand weights (csrc/pointwise_split.hip, w_frag)
  float *y, *part;
  int Cin, mid, H, W, Ho, Wo, pad_t, pad_l;
  int tiles_x, tiles_per_image, nchunks;
};

template <int K, int S>
struct FXGeom {
  static constexpr int TH = 8, TW = S == 1 ? 32 : 16;
  static constexpr int RG = 256 / (8 * TW);            // thread row groups
  static constexpr int NR = TH / RG;                   // output rows per thread
  static constexpr int IH = (TH - 1) * S + K, IW = (TW - 1) * S + K;
  static constexpr int P = IH * IW, MT = (P + 31) / 32;
  static constexpr int HALF = (IW + 1) / 2;
  static constexpr int NLI = (NR - 1) * S + K;         // halo rows a thread walks
  static constexpr int LDS_BYTES = MT * 32 * 32 * 4;
  // LDS slot of halo pixel (py, px)
  __device__ static __forceinline__ int slot(int py, int px) {
    return S == 1 ? py * IW + px : py * IW + (px >> 1) + (px & 1) * HALF;
  }
};

__device__ __forceinline__ void fx_split8(const float4 u, const float4 v, bf16x8& hi, bf16x8& lo) {
  const float f[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 h = (__bf16)f[i];
    hi[i] = h;
    lo[i] = (__bf16)(f[i] - (float)h);
  }
}

__device__ __forceinline__ float4 fx_fma4(const float4 w, const float4 v, float4 a) {
  a.x = fmaf(w.x, v.x, a.x); a.y = fmaf(w.y, v.y, a.y); a.z = fmaf(w.z, v.z, a.z); a.w = fmaf(w.w, v.w, a.w);
  return a;
}

template <int K, int S, int KS>
__global__ __launch_bounds__(256, K == 3 ? (S == 2 ? 2 : 3) : 1) void mbconv_expand_dw_kernel(FXArgs p) {
  using G = FXGeom<K, S>;
  extern __shared__ __attribute__((aligned(16))) float e[];        // [MT * 32 pixels][32 channels]
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // scalar: what depends on it branches on the scalar unit

  int wg = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile = wg / p.nchunks, chunk = wg - tile * p.nchunks;
  const int b = tile / p.tiles_per_image, t2 = tile - b * p.tiles_per_image;
  const int ty = t2 / p.tiles_x, tx = t2 - ty * p.tiles_x;
  const int oy0 = ty * G::TH, ox0 = tx * G::TW;
  const int iy0 = oy0 * S - p.pad_t, ix0 = ox0 * S - p.pad_l;
  const int n0 = chunk * 32;

  // phase-B role; its depthwise weights are fetched before phase A when the registers allow (k = 3), behind it otherwise
  const int q = tid & 7, col = (tid >> 3) % G::TW, rg = (tid >> 3) / G::TW;
  const int cq = n0 + 4 * q;
  const bool cok = cq < p.mid;
  float4 wdw[K * K];                              // (fetched behind phase A: held through it they cost a wavefront of occupancy)
  const float4 bdw = (cok && p.bd != nullptr) ? ld4(p.bd + cq) : make_float4(0.f, 0.f, 0.f, 0.f);

  // ---------------- phase A: expand GEMM of the halo pixels into LDS
  // Halo pixel number pp = py * IW + j IS its LDS slot; column px = j (stride 1) or the even columns first (stride 2).
  bf16x8 bh[KS], bl[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const __bf16* f = p.wp + (((long)chunk * KS + s) * 2) * 512 + lane * 8;
    bh[s] = *reinterpret_cast<const bf16x8*>(f);
    bl[s] = *reinterpret_cast<const bf16x8*>(f + 512);
  }
  const bool nok = n0 + l31 < p.mid;
  const float bev = (nok && p.be != nullptr) ? p.be[n0 + l31] : 0.f;
  const float* xb = p.x + (long)b * p.H * p.W * p.Cin;
  const bool interior = iy0 >= 0 && ix0 >= 0 && iy0 + G::IH <= p.H && ix0 + G::IW <= p.W;      // wave-uniform
  constexpr int NMT = (G::MT + 3) / 4;          // M tiles (32 halo pixels) per wavefront
  constexpr int GRP = S == 1 ? 2 : 3;           // M tiles whose loads are in flight together.  Stride 1 (stage 2's four blocks): two, at
                                                // three wavefronts per SIMD (194 -> 179 us); stride 2: three at two per SIMD (the other way
                                                // round they lose 7 - 14 %: tools/exp_mbconv.py)
#pragma unroll
  for (int g0 = 0; g0 < NMT; g0 += GRP) {
    float4 raw[GRP][KS][2];
#pragma unroll
    for (int i = 0; i < GRP; ++i) {
      const int mt = wave + 4 * (g0 + i);
      if (g0 + i < NMT && mt < G::MT) {
        // UNCONDITIONAL loads from a clamped address: rows beyond the halo (the last M tile) and pixels outside the image read a
        // real pixel instead -- the former land in LDS slots nobody reads, the latter are forced to zero at the store below --
        // and the K padding (Cin = 40: octet 40 .. 47) re-reads the pixel's last octet against zero weights (hip_ops.SplitWeight
        // pads K with zeros).  A predicated load is a branch around every load here: ~200 in this kernel before.
        const int pp = min(mt * 32 + l31, G::P - 1);
        const int py = pp / G::IW, j = pp - py * G::IW;
        const int px = S == 1 ? j : (j < G::HALF ? 2 * j : 2 * (j - G::HALF) + 1);
        const int iy = min(max(iy0 + py, 0), p.H - 1), ix = min(max(ix0 + px, 0), p.W - 1);
        const float* row = xb + (unsigned)((iy * p.W + ix) * p.Cin);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const int k = min(16 * s + 8 * hh, p.Cin - 8);
          raw[i][s][0] = ld4(row + k);
          raw[i][s][1] = ld4(row + k + 4);
        }
      }
    }
    f32x16 acc[GRP];                              // start from the bias: column l31 of every accumulator row
#pragma unroll
    for (int i = 0; i < GRP; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = bev;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int i = 0; i < GRP; ++i) {
        const int mt = wave + 4 * (g0 + i);
        if (g0 + i < NMT && mt < G::MT) {
          bf16x8 ah, al;
          fx_split8(raw[i][s][0], raw[i][s][1], ah, al);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[s], acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[s], acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[s], acc[i], 0, 0, 0);
        }
      }
#pragma unroll
    for (int i = 0; i < GRP; ++i) {
      const int mt = wave + 4 * (g0 + i);
      if (g0 + i < NMT && mt < G::MT) {
        float* dst = e + (mt * 32 + 4 * hh) * 32 + l31;           // accumulator row r -> pixel mt * 32 + acc_row(r, hh)
        if (interior) {
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[acc_row(r, 0) * 32] = fast_silu(acc[i][r]);   // channels >= mid: weights and bias are 0 -> silu(0) = 0
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int pr = mt * 32 + acc_row(r, hh);
            const int pry = pr / G::IW, jr = pr - pry * G::IW;
            const int prx = S == 1 ? jr : (jr < G::HALF ? 2 * jr : 2 * (jr - G::HALF) + 1);
            const bool in = nok && (unsigned)(iy0 + pry) < (unsigned)p.H && (unsigned)(ix0 + prx) < (unsigned)p.W;
            dst[acc_row(r, 0) * 32] = in ? fast_silu(acc[i][r]) : 0.f;
          }
        }
      }
    }
  }
  {
    const int cqs = cok ? cq : 0;                 // (channels beyond mid: any quad -- their outputs are never stored)
#pragma unroll
    for (int t = 0; t < K * K; ++t) wdw[t] = ld4(p.wd + (unsigned)(t * p.mid + cqs));
  }
  __syncthreads();

  // ---------------- phase B: depthwise over the LDS tile
  float4 acc[G::NR];
#pragma unroll
  for (int o = 0; o < G::NR; ++o) acc[o] = bdw;
  const int prow0 = rg * G::NR * S;
#pragma unroll
  for (int li = 0; li < G::NLI; ++li) {
    float4 v[K];
#pragma unroll
    for (int kx = 0; kx < K; ++kx) v[kx] = *reinterpret_cast<const float4*>(e + G::slot(prow0 + li, col * S + kx) * 32 + 4 * q);
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
      const int d = li - ky;
      if (d >= 0 && d % S == 0 && d / S < G::NR) {
#pragma unroll
        for (int kx = 0; kx < K; ++kx) acc[d / S] = fx_fma4(wdw[ky * K + kx], v[kx], acc[d / S]);
      }
    }
  }
  float4 psum = make_float4(0.f, 0.f, 0.f, 0.f);
  const int ox = ox0 + col;
  float* orow = p.y + (((long)b * p.Ho + oy0 + rg * G::NR) * p.Wo + ox) * p.mid + cq;
  const long ostep = (long)p.Wo * p.mid;
#pragma unroll
  for (int o = 0; o < G::NR; ++o, orow += ostep) {
    const int oy = oy0 + rg * G::NR + o;
    if (cok && oy < p.Ho && ox < p.Wo) {
      float4 r = acc[o];
      r.x = fast_silu(r.x); r.y = fast_silu(r.y); r.z = fast_silu(r.z); r.w = fast_silu(r.w);
      *reinterpret_cast<float4*>(orow) = r;
      psum.x += r.x; psum.y += r.y; psum.z += r.z; psum.w += r.w;
    }
  }
  // per-channel sum of the tile (squeeze-excite pooling partial), fixed order
  __syncthreads();
  *reinterpret_cast<float4*>(e + (tid >> 3) * 32 + 4 * q) = psum;
  __syncthreads();
  if (tid < 32 && n0 + tid < p.mid) {
    float s = 0.f;
#pragma unroll 8
    for (int j = 0; j < 32; ++j) s += e[j * 32 + tid];
    p.part[((long)b * p.tiles_per_image + t2) * p.mid + n0 + tid] = s;
  }
}

template <int K, int S, int KS>
int fx_launch(const FXArgs& a, int B, hipStream_t st) {
  using G = FXGeom<K, S>;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)mbconv_expand_dw_kernel<K, S, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    attr = true;
  }
  const long grid = (long)B * a.tiles_per_image * a.nchunks;
  hipLaunchKernelGGL((mbconv_expand_dw_kernel<K, S, KS>), dim3((unsigned)grid), dim3(256), G::LDS_BYTES, st, a);
  OCV_CHECK_LAUNCH("ocv_mbconv_expand_dw_fwd");
  return 0;
}

template <int K, int S>
int fx_dispatch_ks(const FXArgs& a, int B, int ks, hipStream_t st) {
  if (ks == 2) return fx_launch<K, S, 2>(a, B, st);
  if (ks == 3) return fx_launch<K, S, 3>(a, B, st);
  return fx_launch<K, S, 4>(a, B, st);
}

}  // namespace

extern "C" int ocv_mbconv_expand_dw_tiles(int Ho, int Wo, int k, int stride) {
  if (Ho < 1 || Wo < 1 || (k != 3 && k != 5) || (stride != 1 && stride != 2)) return 0;
  const int tw = stride == 1 ? 32 : 16;
  return ((Ho + 7) / 8) * ((Wo + tw - 1) / tw);
}

namespace {
int fx_run(const float* x, const void* w_packed, const float* bias_expand, const float* w_dw, const float* bias_dw, float* y,
           float* part, int B, int H, int W, int Cin, int mid, int k, int stride, int pad_t, int pad_l,
           int Ho, int Wo, ocv_stream_t stream) {
  OCV_CHECK_ARG(x && w_packed && w_dw && y && part, "ocv_mbconv_expand_dw_fwd: null pointer");
  OCV_CHECK_ARG(B >= 1 && H >= 1 && W >= 1 && Ho >= 1 && Wo >= 1, "ocv_mbconv_expand_dw_fwd: bad sizes");
  OCV_CHECK_ARG(Cin >= 24 && Cin % 8 == 0 && Cin <= 64, "ocv_mbconv_expand_dw_fwd: Cin must be a multiple of 8 in [24, 64] (got %d)", Cin);
  OCV_CHECK_ARG(mid >= 4 && mid % 4 == 0, "ocv_mbconv_expand_dw_fwd: expanded channels must be a multiple of 4 (got %d)", mid);
  OCV_CHECK_ARG((k == 3 || k == 5) && (stride == 1 || stride == 2), "ocv_mbconv_expand_dw_fwd: k must be 3 or 5 and stride 1 or 2 (got k=%d s=%d)", k, stride);
  OCV_CHECK_ARG(pad_t >= 0 && pad_l >= 0 && pad_t < k && pad_l < k, "ocv_mbconv_expand_dw_fwd: bad padding");
  OCV_CHECK_ARG((Ho - 1) * stride - pad_t < H && (Wo - 1) * stride - pad_l < W, "ocv_mbconv_expand_dw_fwd: output larger than the padded input allows");
  OCV_CHECK_ARG(ocv_aligned16(x) && ocv_aligned16(w_packed) && ocv_aligned16(w_dw) && ocv_aligned16(bias_dw) && ocv_aligned16(y),
                "ocv_mbconv_expand_dw_fwd: operands must be 16-byte aligned");
  FXArgs a{};
  a.x = x; a.be = bias_expand; a.wd = w_dw; a.bd = bias_dw; a.wp = (const __bf16*)w_packed; a.y = y; a.part = part;
  a.Cin = Cin; a.mid = mid; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.pad_t = pad_t; a.pad_l = pad_l;
  const int tw = stride == 1 ? 32 : 16;
  a.tiles_x = (Wo + tw - 1) / tw;
  a.tiles_per_image = ((Ho + 7) / 8) * a.tiles_x;
  a.nchunks = (mid + 31) / 32;
  OCV_CHECK_ARG((long)B * a.tiles_per_image * a.nchunks < (1L << 31), "ocv_mbconv_expand_dw_fwd: too many work items");
  const int ks = (Cin + 15) / 16;
  hipStream_t st = (hipStream_t)stream;
  if (k == 3 && stride == 1) return fx_dispatch_ks<3, 1>(a, B, ks, st);
  if (k == 3 && stride == 2) return fx_dispatch_ks<3, 2>(a, B, ks, st);
  if (k == 5 && stride == 1) return fx_dispatch_ks<5, 1>(a, B, ks, st);
  return fx_dispatch_ks<5, 2>(a, B, ks, st);
}
}  // namespace

extern "C" int ocv_mbconv_expand_dw_fwd(const float* x, const void* w_packed, const float* bias_expand, const float* w_dw,
                                        const float* bias_dw, float* y, float* part, int B, int H, int W, int Cin, int mid,
                                        int k, int stride, int pad_t, int pad_l, int Ho, int Wo, ocv_stream_t stream) {
  return fx_run(x, w_packed, bias_expand, w_dw, bias_dw, y, part, B, H, W, Cin, mid, k, stride, pad_t, pad_l, Ho, Wo, stream);
}
