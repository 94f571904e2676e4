// 3 x 3 convolution of a bilinearly UP-SAMPLED tensor, computed at the LOW resolution (UNet decoder, first convolution of
// every UpSampleWithSkip stage: modules/DenseFeatureExtractor.py:44-47 then :37-39).
//
//   y = act( bias + conv3x3_{Wa}( up(x) ) + conv3x3_{Ws}( skip ) )          Wa / Ws: the input-channel halves of the weight
//
// up() (F.interpolate, bilinear, align_corners=True) is linear and acts per channel; the convolution mixes channels per
// tap.  So conv_{Wa}(up(x))[p] = sum_t Wa_t . up(x)[p + t] = sum_t sum_{4 nb} coef(p + t, nb) . (Wa_t . x[nb]):
// the nine tap products z_t = Wa_t . x are formed ONCE PER LOW-RESOLUTION PIXEL (one 1 x 1 GEMM with 9 Cout output
// columns on h x w pixels instead of nine taps on H x W: ~4x fewer matrix-core operations for the up-sampled channels,
// which are 90 % of the stage's input) and this kernel interpolates them to the high resolution per tap, with the zero
// padding of the convolution applied to the TAP position, adds the high-resolution skip part and the bias, applies the
// activation and writes fp32 and / or the hl32 split layout.  Exact re-association of the reference's arithmetic: the
// interpolation uses ATen's coefficients (scale = (in - 1) / (out - 1), src = scale * dst, lambda = src - floor(src)).
//
// Work item: 4 output channels of one output pixel; a workgroup owns an 8 x 16 pixel tile x 32 channels and walks the nine
// taps: per tap the 32-channel slab of the low-resolution rows under the tile (<= FQ pixels) is staged in LDS (double
// buffered, 128 bytes per pixel and tap) and every item reads its 4 neighbours from there.
#include "common.hpp"
#include "../../include/objcavit_hip.h"

namespace {

typedef __bf16 ti_bf16x4 __attribute__((ext_vector_type(4)));

constexpr int TY = 8, TX = 16, CB = 32;          // output tile and channel block
constexpr int FQ = 192;                          // low-resolution pixels staged per tap (footprint capacity)

struct TIArgs {
  const float* z;        // [B][h - 2 zpad][w - 2 zpad][9 Cout]: column t Cout + co = (Wa_t . x)[co]
  const float* zborder;  // [9 Cout]: the value of z on the zpad-wide border ring of the logical h x w grid (zpad = 1)
  const float* s;        // [B][H][W][Cout] skip-part convolution (raw), nullable
  const float* bias;     // nullable
  float* y;              // [B][H][W][Cout] fp32, nullable
  __bf16* yhl;           // hl32 split, nullable
  int h, w, H, W, Cout, Cpo, act, zpad;
  float sh, sw;
  int tiles_x, tiles_y;
};

__device__ __forceinline__ float ti_act(float v, int act) {
  if (act == OCV_ACT_LEAKY_RELU) return v > 0.f ? v : 0.01f * v;
  if (act == OCV_ACT_SILU) return fast_silu(v);
  if (act == OCV_ACT_RELU) return fmaxf(v, 0.f);
  return v;
}

__global__ __launch_bounds__(256) void tap_interp_kernel(TIArgs p) {
  __shared__ __attribute__((aligned(16))) float zs[2][FQ][CB];
  const int tid = threadIdx.x;
  int wg = blockIdx.x;
  const int tx = wg % p.tiles_x;
  wg /= p.tiles_x;
  const int ty = wg % p.tiles_y, b = wg / p.tiles_y;
  const int cb0 = blockIdx.y * CB;
  const int Y0 = ty * TY, X0 = tx * TX;
  // low-resolution footprint of the tile with its one-pixel tap halo
  const int ya = max(Y0 - 1, 0), yb = min(Y0 + TY, p.H - 1), xa = max(X0 - 1, 0), xb = min(X0 + TX, p.W - 1);
  const int qy0 = (int)(p.sh * ya), qx0 = (int)(p.sw * xa);
  const int qy1 = min((int)(p.sh * yb) + 1, p.h - 1), qx1 = min((int)(p.sw * xb) + 1, p.w - 1);
  const int fw = qx1 - qx0 + 1, fq = (qy1 - qy0 + 1) * fw;           // <= FQ (checked on the host)
  const int hp = p.h - 2 * p.zpad, wp = p.w - 2 * p.zpad;             // the stored grid
  const float* zb = p.z + (long)b * hp * wp * 9 * p.Cout + cb0;

  auto stage = [&](int t, int buf) {
    for (int i = tid; i < fq * (CB / 4); i += 256) {
      const int q = i >> 3, c4 = (i & 7) * 4;
      const int qy = qy0 + q / fw - p.zpad, qx = qx0 + q % fw - p.zpad;
      const float* src = (unsigned)qy < (unsigned)hp && (unsigned)qx < (unsigned)wp
                             ? zb + ((long)qy * wp + qx) * 9 * p.Cout + (long)t * p.Cout + c4
                             : p.zborder + (long)t * p.Cout + cb0 + c4;
      *reinterpret_cast<float4*>(&zs[buf][q][c4]) = ld4(src);
    }
  };

  // items: channel group cg (4 channels), pixels px = (tid >> 3) + 32 i of the 128-pixel tile
  const int cg = (tid & 7) * 4;
  float4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);

  stage(0, 0);
  __syncthreads();
  for (int t = 0; t < 9; ++t) {
    if (t + 1 < 9) stage(t + 1, (t + 1) & 1);
    const int dy = t / 3 - 1, dx = t % 3 - 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int px = (tid >> 3) + 32 * i;
      const int Y = Y0 + px / TX + dy, X = X0 + px % TX + dx;
      if ((unsigned)Y < (unsigned)p.H && (unsigned)X < (unsigned)p.W) {      // zero padding of the convolution
        const float sy = p.sh * Y, sx = p.sw * X;
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < p.h - 1 ? 1 : 0), x1 = x0 + (x0 < p.w - 1 ? 1 : 0);
        const float h1 = sy - (float)y0, h0 = 1.0f - h1, w1 = sx - (float)x0, w0 = 1.0f - w1;
        const int i0 = (y0 - qy0) * fw - qx0, i1 = (y1 - qy0) * fw - qx0;
        const float4 v00 = *reinterpret_cast<const float4*>(&zs[t & 1][i0 + x0][cg]);
        const float4 v01 = *reinterpret_cast<const float4*>(&zs[t & 1][i0 + x1][cg]);
        const float4 v10 = *reinterpret_cast<const float4*>(&zs[t & 1][i1 + x0][cg]);
        const float4 v11 = *reinterpret_cast<const float4*>(&zs[t & 1][i1 + x1][cg]);
        acc[i].x += h0 * (w0 * v00.x + w1 * v01.x) + h1 * (w0 * v10.x + w1 * v11.x);
        acc[i].y += h0 * (w0 * v00.y + w1 * v01.y) + h1 * (w0 * v10.y + w1 * v11.y);
        acc[i].z += h0 * (w0 * v00.z + w1 * v01.z) + h1 * (w0 * v10.z + w1 * v11.z);
        acc[i].w += h0 * (w0 * v00.w + w1 * v01.w) + h1 * (w0 * v10.w + w1 * v11.w);
      }
    }
    __syncthreads();
  }

  const int n = cb0 + cg;
  if (n >= p.Cout) return;
  const float4 bv = p.bias != nullptr ? ld4(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int px = (tid >> 3) + 32 * i;
    const int Y = Y0 + px / TX, X = X0 + px % TX;
    if (Y >= p.H || X >= p.W) continue;
    const long pix = ((long)b * p.H + Y) * p.W + X;
    float4 v = acc[i];
    if (p.s != nullptr) {
      const float4 sv = ld4(p.s + pix * p.Cout + n);
      v.x += sv.x; v.y += sv.y; v.z += sv.z; v.w += sv.w;
    }
    v.x = ti_act(v.x + bv.x, p.act); v.y = ti_act(v.y + bv.y, p.act);
    v.z = ti_act(v.z + bv.z, p.act); v.w = ti_act(v.w + bv.w, p.act);
    if (p.y != nullptr) *reinterpret_cast<float4*>(p.y + pix * p.Cout + n) = v;
    if (p.yhl != nullptr) {
      const float f[4] = {v.x, v.y, v.z, v.w};
      ti_bf16x4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const __bf16 hb = (__bf16)f[e];
        hi[e] = hb;
        lo[e] = (__bf16)(f[e] - (float)hb);
      }
      __bf16* d = p.yhl + pix * 2 * p.Cpo + (n >> 5) * 64 + (n & 31);
      *reinterpret_cast<ti_bf16x4*>(d) = hi;
      *reinterpret_cast<ti_bf16x4*>(d + 32) = lo;
    }
  }
}

// largest low-resolution footprint of an 8 x 16 tile with halo, over all tile positions (monotone maps: check every tile row / column)
int ti_footprint(int h, int w, int H, int W) {
  const float sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
  int fh = 0, fwid = 0;
  for (int Y0 = 0; Y0 < H; Y0 += TY) {
    const int ya = Y0 - 1 > 0 ? Y0 - 1 : 0, yb = Y0 + TY < H - 1 ? Y0 + TY : H - 1;
    int q1 = (int)(sh * yb) + 1;
    if (q1 > h - 1) q1 = h - 1;
    const int f = q1 - (int)(sh * ya) + 1;
    if (f > fh) fh = f;
  }
  for (int X0 = 0; X0 < W; X0 += TX) {
    const int xa = X0 - 1 > 0 ? X0 - 1 : 0, xb = X0 + TX < W - 1 ? X0 + TX : W - 1;
    int q1 = (int)(sw * xb) + 1;
    if (q1 > w - 1) q1 = w - 1;
    const int f = q1 - (int)(sw * xa) + 1;
    if (f > fwid) fwid = f;
  }
  return fh * fwid;
}

}  // namespace

extern "C" int ocv_tap_interp_supported(int h, int w, int H, int W, int Cout) {
  if (h < 1 || w < 1 || H < 1 || W < 1 || Cout < 4 || Cout % 4 != 0) return 0;
  return ti_footprint(h, w, H, W) <= FQ ? 1 : 0;
}

extern "C" int ocv_tap_interp_combine_fwd(const float* z, int h, int w, int zpad, const float* zborder, const float* s,
                                          const float* bias, float* y, void* y_hl, int B, int H, int W, int Cout, int act,
                                          ocv_stream_t stream) {
  OCV_CHECK_ARG(z && (y || y_hl), "ocv_tap_interp_combine_fwd: null pointer");
  OCV_CHECK_ARG((zpad == 0 && zborder == nullptr) || (zpad == 1 && zborder != nullptr && h >= 3 && w >= 3 && ocv_aligned16(zborder)),
                "ocv_tap_interp_combine_fwd: zpad must be 0 (no border vector) or 1 (with a 16-byte aligned border vector, h, w >= 3)");
  OCV_CHECK_ARG(B >= 1 && h >= 1 && w >= 1 && H >= 1 && W >= 1 && Cout >= 4 && Cout % 4 == 0,
                "ocv_tap_interp_combine_fwd: bad sizes (Cout must be a multiple of 4, got %d)", Cout);
  OCV_CHECK_ARG(act >= 0 && act <= 3, "ocv_tap_interp_combine_fwd: unknown activation %d", act);
  OCV_CHECK_ARG(ocv_aligned16(z) && ocv_aligned16(s) && ocv_aligned16(bias) && ocv_aligned16(y) && ocv_aligned16(y_hl),
                "ocv_tap_interp_combine_fwd: operands must be 16-byte aligned");
  OCV_CHECK_ARG(ocv_tap_interp_supported(h, w, H, W, Cout), "ocv_tap_interp_combine_fwd: the low-resolution footprint of an output tile "
                "exceeds the staging buffer (h=%d w=%d H=%d W=%d): not an up-sampling by ~2 or more", h, w, H, W);
  TIArgs a{z, zborder, s, bias, y, (__bf16*)y_hl, h, w, H, W, Cout, (Cout + 31) / 32 * 32, act, zpad,
           H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f,
           ocv_cdiv(W, TX), ocv_cdiv(H, TY)};
  const long nwg = (long)a.tiles_x * a.tiles_y * B;
  OCV_CHECK_ARG(nwg < (1L << 31) && ocv_cdiv(Cout, CB) <= 65535, "ocv_tap_interp_combine_fwd: grid too large");
  if (y_hl != nullptr && Cout % 32 != 0) {
    const hipError_t e = hipMemsetAsync(y_hl, 0, ocv_split_act_elems(B, H, W, Cout) * sizeof(__bf16), (hipStream_t)stream);
    OCV_CHECK_ARG(e == hipSuccess, "ocv_tap_interp_combine_fwd: hipMemsetAsync failed: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(tap_interp_kernel, dim3((unsigned)nwg, ocv_cdiv(Cout, CB)), dim3(256), 0, (hipStream_t)stream, a);
  OCV_CHECK_LAUNCH("ocv_tap_interp_combine_fwd");
  return 0;
}
