// Stem convolution of the EfficientNet encoder: dense 3x3, stride 2, TensorFlow "SAME" padding, 3 -> 48 channels,
// + folded BatchNorm + SiLU, reading the NCHW image and writing the NHWC activation the MBConv kernels consume
// (row N1 of SURVEY.md section 8; conv_stem / bn1 / act1 of the hub backbone the reference walks in
// modules/DenseFeatureExtractor.py:18-27).
//
// K = Cin k k = 27 is far too short for an implicit-GEMM pipeline (MIOpen's fp32 NHWC igemm needs 0.75 ms for it, plus
// separate BatchNorm and SiLU passes over the 236 MB output); the layer is a pure output-write stream.  A wavefront
// owns 32 consecutive output pixels: lane (pixel l & 31, tap parity l >> 5) gathers its ceil(K / 2) input taps straight
// from the three image planes (neighbouring lanes read neighbouring pixels, stride-2 floats), which is exactly the A
// operand of v_mfma_f32_32x32x2_f32; the whole weight matrix (Cout x K, 5 KB) sits in VGPRs as the B operand for the
// lifetime of the wavefront, which walks pixel tiles grid-stride.  Exact fp32.  Output rows are 128-byte (+ 64-byte)
// contiguous runs per pixel.
#include "common.hpp"
#include "../../include/objcavit_hip.h"

namespace {

struct StemArgs {
  const float *x, *w, *bias;
  float* y;
  int Cin, H, W, Cout, stride, pad_t, pad_l, Ho, Wo, act;
  long M;          // B * Ho * Wo
  long tiles;      // ceil(M / 32)
};

constexpr int STEM_KT = 16;      // K <= 32

template <int KS, int NT>
__global__ __launch_bounds__(256) void stem_conv_kernel(StemArgs p) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, hh = lane >> 5;
  const int K = p.Cin * KS * KS;
  float wreg[NT][STEM_KT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = 32 * j + l31;
#pragma unroll
    for (int t = 0; t < STEM_KT; ++t) {
      const int k = 2 * t + hh;
      wreg[j][t] = (n < p.Cout && k < K) ? p.w[(long)n * K + k] : 0.f;
    }
  }
  float bv[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) bv[j] = (p.bias != nullptr && 32 * j + l31 < p.Cout) ? p.bias[32 * j + l31] : 0.f;

  const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
  const long plane = (long)p.H * p.W;
  for (long tile = wave0; tile < p.tiles; tile += nwaves) {
    const long m = tile * 32 + l31;
    const bool ok = m < p.M;
    const long mm = ok ? m : 0;
    const int ox = (int)(mm % p.Wo);
    const long t1 = mm / p.Wo;
    const int oy = (int)(t1 % p.Ho);
    const long b = t1 / p.Ho;
    const int iy0 = oy * p.stride - p.pad_t, ix0 = ox * p.stride - p.pad_l;
    const float* img = p.x + b * p.Cin * plane;
    float a[STEM_KT];
#pragma unroll
    for (int t = 0; t < STEM_KT; ++t) {
      // taps k = 2t (hh = 0) and 2t + 1 (hh = 1): channel / row / column are compile-time for each, selected by hh
      constexpr int dummy = 0; (void)dummy;
      const int k0 = 2 * t, k1 = 2 * t + 1;
      const int c = hh ? k1 / (KS * KS) : k0 / (KS * KS);
      const int ky = hh ? (k1 % (KS * KS)) / KS : (k0 % (KS * KS)) / KS;
      const int kx = hh ? k1 % KS : k0 % KS;
      const int iy = iy0 + ky, ix = ix0 + kx;
      const bool in = ok && (2 * t + hh) < K && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      a[t] = in ? img[c * plane + (long)iy * p.W + ix] : 0.f;
    }
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = f32x16{0};
#pragma unroll
    for (int t = 0; t < STEM_KT; ++t) {
      if (2 * t < K) {
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = mfma_32x32x2(a[t], wreg[j][t], acc[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = 32 * j + l31;
      if (n < p.Cout) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long mo = tile * 32 + acc_row(r, hh);
          if (mo < p.M) {
            float v = acc[j][r] + bv[j];
            if (p.act == OCV_ACT_SILU) v = fast_silu(v);
            else if (p.act == OCV_ACT_RELU) v = fmaxf(v, 0.f);
            else if (p.act == OCV_ACT_LEAKY_RELU) v = v > 0.f ? v : 0.01f * v;
            p.y[mo * p.Cout + n] = v;
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Round 4: the kernel above, re-done for the shapes the encoder has (Cout a multiple of 8).
// Index arithmetic is 32-bit throughout (the launcher checks that the image and the output have fewer than 2^31 elements): the first
// version did three 64-bit divisions per lane and tile plus a 64-bit multiply per load and store -- 1629 vector instructions per
// 32-pixel tile (SQ counters, profiles/r04g_sq.json: VALU busy 121 us of the launch's 174), i.e. it was bound by its own address
// arithmetic, not by the 295 MB it moves.  The activation is a template parameter; only the last tile checks its rows.
template <int KS, int NT, int ACT, int KT>        // KT = K steps of two taps: 14 for the encoder's 3 channels x 9 taps, 16 in general
__global__ __launch_bounds__(256, 2) void stem_conv_wide_kernel(StemArgs p) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, hh = lane >> 5;
  const int K = p.Cin * KS * KS;
  float wreg[NT][KT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = 32 * j + l31;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      const int k = 2 * t + hh;
      wreg[j][t] = (n < p.Cout && k < K) ? p.w[n * K + k] : 0.f;          // (taps beyond K multiply by zero: no branch in the loop)
    }
  }
  // The MFMA's A operand is the WEIGHT pair of channel l31, its B operand the tap pair of pixel l31: the accumulator's rows are
  // channels (register r = channel acc_row(r, hh): four runs of four consecutive channels), its column is this lane's pixel -- a lane
  // stores its pixel's channels as 16-byte pieces (6 wide stores per tile at 48 channels, where the pixel-major accumulator took 48
  // four-byte ones).  For Cout % 8 == 0 (the launcher keeps the kernel above for other widths).
  float bv[NT][16];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = 32 * j + acc_row(r, hh);
      bv[j][r] = (p.bias != nullptr && n < p.Cout) ? p.bias[n] : 0.f;
    }

  // taps k = 2 t (lanes 0 - 31) and 2 t + 1 (lanes 32 - 63): plane / row / column offsets are wave-uniform per parity (scalar registers),
  // a lane selects its own with one v_cndmask -- held per lane they cost 48 VGPRs and a wavefront of occupancy
  const unsigned plane = (unsigned)(p.H * p.W);
  auto tap_ky = [&](int t) { const int k0 = (2 * t) % (KS * KS), k1 = (2 * t + 1) % (KS * KS); return hh ? k1 / KS : k0 / KS; };
  auto tap_kx = [&](int t) { const int k0 = (2 * t) % (KS * KS), k1 = (2 * t + 1) % (KS * KS); return hh ? k1 % KS : k0 % KS; };
  auto tap_off = [&](int t) {
    const int k0 = 2 * t, k1 = 2 * t + 1;
    const int o0 = k0 < K ? (int)((k0 / (KS * KS)) * plane) + ((k0 % (KS * KS)) / KS) * p.W + (k0 % KS) : -1;
    const int o1 = k1 < K ? (int)((k1 / (KS * KS)) * plane) + ((k1 % (KS * KS)) / KS) * p.W + (k1 % KS) : -1;
    return hh ? o1 : o0;
  };
  const unsigned M = (unsigned)p.M, Wo = (unsigned)p.Wo, Ho = (unsigned)p.Ho;
  const unsigned img_elems = (unsigned)p.Cin * plane;
  const unsigned lane_off = (unsigned)(l31 * p.Cout + 4 * hh);               // output offset of pixel l31, channel 4 hh
  const unsigned wave0 = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = gridDim.x * 4;
  // the taps of a tile: KT gathers per lane, issued one tile AHEAD (their ~2 us of latency was a third of a wavefront's time per tile)
  // (the loads are UNCONDITIONAL, from a clamped index; which taps lie outside the image is a bit mask applied when the values are
  //  used, one tile later -- a select right behind the load would make the wavefront wait for it there)
  auto gather = [&](unsigned tile, float (&a)[KT]) -> unsigned {
    const unsigned m = tile * 32 + l31;
    const bool ok = m < M;
    const unsigned mm = ok ? m : 0u;
    const unsigned row = mm / Wo, ox = mm - row * Wo;
    const unsigned b = row / Ho, oy = row - b * Ho;
    const int iy0 = (int)oy * p.stride - p.pad_t, ix0 = (int)ox * p.stride - p.pad_l;
    const int base = (int)(b * img_elems) + iy0 * p.W + ix0;                 // 32-bit element index from p.x (scalar base + lane offset)
    const bool inner = ok && iy0 >= 0 && ix0 >= 0 && iy0 + KS <= p.H && ix0 + KS <= p.W;
    unsigned mask = 0;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      const int iy = iy0 + tap_ky(t), ix = ix0 + tap_kx(t), off = tap_off(t);
      const bool in = off >= 0 && (inner || (ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W));
      mask |= in ? 1u << t : 0u;
      a[t] = p.x[in ? (unsigned)(base + off) : 0u];
    }
    return mask;
  };
  float a[KT], an[KT];
  unsigned amask = 0, anmask = 0;
  if (wave0 < (unsigned)p.tiles) amask = gather(wave0, a);
  for (unsigned tile = wave0; tile < (unsigned)p.tiles; tile += nwaves) {    // (scalar: the tile is the wavefront's)
    if (tile + nwaves < (unsigned)p.tiles) anmask = gather(tile + nwaves, an);
#pragma unroll
    for (int t = 0; t < KT; ++t) a[t] = (amask >> t) & 1u ? a[t] : 0.f;
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = bv[j][r];
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = mfma_32x32x2(wreg[j][t], a[t], acc[j]);
    const unsigned yt = tile * 32 * (unsigned)p.Cout + lane_off;             // 32-bit element index from p.y
    const bool pix_ok = tile * 32 + l31 < M;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (32 * j + 8 * g < p.Cout) {                                        // wave-uniform (Cout % 8 == 0: both halves of the run are channels)
          float4 v;
          float* vv = reinterpret_cast<float*>(&v);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float x = acc[j][4 * g + e];
            if (ACT == OCV_ACT_SILU) x = fast_silu(x);
            else if (ACT == OCV_ACT_RELU) x = fmaxf(x, 0.f);
            else if (ACT == OCV_ACT_LEAKY_RELU) x = x > 0.f ? x : 0.01f * x;
            vv[e] = x;
          }
          if (pix_ok) *reinterpret_cast<float4*>(p.y + (yt + 32 * j + 8 * g)) = v;
        }
      }
#pragma unroll
    for (int t = 0; t < KT; ++t) a[t] = an[t];
    amask = anmask;
  }
}

template <int NT, int KT>
void stem_wide_launch_kt(const StemArgs& a, unsigned blocks, hipStream_t st) {
  switch (a.act) {
    case OCV_ACT_SILU: hipLaunchKernelGGL((stem_conv_wide_kernel<3, NT, OCV_ACT_SILU, KT>), dim3(blocks), dim3(256), 0, st, a); break;
    case OCV_ACT_RELU: hipLaunchKernelGGL((stem_conv_wide_kernel<3, NT, OCV_ACT_RELU, KT>), dim3(blocks), dim3(256), 0, st, a); break;
    case OCV_ACT_LEAKY_RELU: hipLaunchKernelGGL((stem_conv_wide_kernel<3, NT, OCV_ACT_LEAKY_RELU, KT>), dim3(blocks), dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((stem_conv_wide_kernel<3, NT, OCV_ACT_NONE, KT>), dim3(blocks), dim3(256), 0, st, a); break;
  }
}
template <int NT>
void stem_wide_launch(const StemArgs& a, unsigned blocks, hipStream_t st) {
  if (a.Cin * 9 <= 28) stem_wide_launch_kt<NT, 14>(a, blocks, st);
  else stem_wide_launch_kt<NT, STEM_KT>(a, blocks, st);
}

}  // namespace

extern "C" int ocv_stem_conv_fwd(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int H, int W,
                                 int Cout, int k, int stride, int pad_t, int pad_l, int Ho, int Wo, int act,
                                 ocv_stream_t stream) {
  OCV_CHECK_ARG(x && w && y, "ocv_stem_conv_fwd: null pointer");
  OCV_CHECK_ARG(B >= 1 && Cin >= 1 && H >= 1 && W >= 1 && Ho >= 1 && Wo >= 1 && Cout >= 1, "ocv_stem_conv_fwd: bad sizes");
  OCV_CHECK_ARG(k == 3 && Cin * k * k <= 2 * STEM_KT, "ocv_stem_conv_fwd: kernel must be 3x3 with Cin * 9 <= %d (got k=%d Cin=%d)", 2 * STEM_KT, k, Cin);
  OCV_CHECK_ARG(Cout <= 64, "ocv_stem_conv_fwd: at most 64 output channels (got %d)", Cout);
  OCV_CHECK_ARG(stride >= 1 && pad_t >= 0 && pad_l >= 0 && pad_t < k && pad_l < k, "ocv_stem_conv_fwd: bad stride / padding");
  OCV_CHECK_ARG((Ho - 1) * stride - pad_t < H && (Wo - 1) * stride - pad_l < W, "ocv_stem_conv_fwd: output larger than the padded input allows");
  OCV_CHECK_ARG(act >= OCV_ACT_NONE && act <= OCV_ACT_SILU, "ocv_stem_conv_fwd: activation must be none / ReLU / LeakyReLU / SiLU");
  StemArgs a{x, w, bias, y, Cin, H, W, Cout, stride, pad_t, pad_l, Ho, Wo, act, (long)B * Ho * Wo, 0};
  a.tiles = (a.M + 31) / 32;
  long blocks = (a.tiles + 3) / 4;
  if (blocks > 256L * 8) blocks = 256L * 8;
  hipStream_t st = (hipStream_t)stream;
  if (Cout % 8 == 0 && ocv_aligned16(y) && (long)B * Cin * H * W < (1L << 31) && (long)B * Ho * Wo * Cout < (1L << 31)) {
    if (Cout <= 32) stem_wide_launch<1>(a, (unsigned)blocks, st);           // 32-bit index arithmetic, 16-byte stores
    else stem_wide_launch<2>(a, (unsigned)blocks, st);
  } else if (Cout <= 32) hipLaunchKernelGGL((stem_conv_kernel<3, 1>), dim3((unsigned)blocks), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((stem_conv_kernel<3, 2>), dim3((unsigned)blocks), dim3(256), 0, st, a);
  OCV_CHECK_LAUNCH("ocv_stem_conv_fwd");
  return 0;
}
