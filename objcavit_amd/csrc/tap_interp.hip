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
// Work item: 4 output channels of 4 output pixels (rows Y, Y+2, Y+4, Y+6 of one column); a workgroup owns an 8 x 16 pixel
// tile x 32 channels and walks the nine taps.  Per tap the 32-channel slab of the low-resolution pixels under the tile (its
// footprint, < FQ pixels, 128 bytes each) is staged in LDS buffer (tap mod 3) by a FIFTH, producer wavefront with LDS-DMA
// (global_load_lds_dwordx4: no registers, no ds_write), two taps ahead of the four consumer wavefronts.  The kernel is bound
// by the latency of those ~10 KB pieces and by how much of it other workgroups on the CU can cover, not by HBM bandwidth
// (ablations, tools/diag/tap_interp_diag.patch: interpolation alone 0.43 ms, staging + epilogue alone 0.55 ms, together
// 0.80 ms when every wavefront did both through registers at two workgroups per CU): the producer / consumer split keeps the
// consumers at 128 registers = 4 wavefronts per SIMD = THREE five-wavefront workgroups per CU (LDS: ~37 KB each for a 2x up-sampling; at
// __launch_bounds__(320, 5) -- four workgroups -- the compiler spills 27 - 37 registers and the launch takes 1.9x as long, and removing the
// kernel's LDS bank conflicts made it 9 % slower: profiles/r03_tap_interp_swizzle.txt).  The x interpolation
// coefficients and LDS offsets depend on (pixel, dx) only and are formed once; the y ones per tap row.  The right-hand x
// neighbour is always read at +128 bytes: where ATen clamps it (last column) its weight is exactly 0 and the slot read holds
// staged (finite) data.
//
// (Round 4 also formed the skip part INSIDE this launch, on the matrix cores: correct, tested, 8 % slower end to end -- a
// wavefront-sized MFMA tile fed from L1 is what a GEMM with LDS tiles exists to avoid (profiles/r04_tap_skip.txt).  Removed in round
// 5; source: tools/diag/tap_skip.patch.txt.)
#include "common.hpp"
#include "../../include/objcavit_hip.h"

namespace {

typedef __bf16 ti_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 ti_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 ti_h16x8 __attribute__((ext_vector_type(8)));

constexpr int TY = 8, TX = 16, CB = 32;          // output tile and channel block
constexpr int FQ = 192;                          // low-resolution pixels staged per tap (footprint capacity)
constexpr int NBUF = 3;                          // LDS buffers: tap t lives in buffer t mod 3 (= dx)

struct TIArgs {
  const float* z;        // [B][h - 2 zpad][w - 2 zpad][9 Cout]: column t Cout + co = (Wa_t . x)[co]
  const float* zborder;  // [9 Cout]: the value of z on the zpad-wide border ring of the logical h x w grid (zpad = 1)
  const float* s;        // [B][H][W][Cout] skip-part convolution (raw), nullable
  const float* bias;     // nullable
  float* y;              // [B][H][W][Cout] fp32, nullable
  __bf16* yhl;           // hl32 split, nullable (bf16 pairs; fp16 pairs with f16)
  int h, w, H, W, Cout, Cpo, act, zpad;
  float sh, sw;
  int tiles_x, tiles_y;
  int fq_cap;            // pixels per staging buffer (multiple of 32: whole 256-lane rounds)
  int f16;               // element type of yhl
  unsigned* range_flag;  // nullable: armed range-guard word, noted only where yhl holds fp16 pairs (common.hpp ocv_range_note)
};

template <int ACT>
__device__ __forceinline__ float ti_act(float v) {
  if constexpr (ACT == OCV_ACT_LEAKY_RELU) return v > 0.f ? v : 0.01f * v;
  if constexpr (ACT == OCV_ACT_SILU) return fast_silu(v);
  if constexpr (ACT == OCV_ACT_RELU) return fmaxf(v, 0.f);
  return v;
}

struct TIArgs;
template <int ACT>
__device__ __forceinline__ void ti_store(const TIArgs& p, const float4 (&acc)[4], const float4 (&sv)[4], int b, int Y0, int X, int n);

typedef __attribute__((address_space(1))) const void* ti_gptr;
typedef __attribute__((address_space(3))) void* ti_lptr;

// s_waitcnt vmcnt(N) alone (gfx9 encoding: vmcnt in bits 3:0 and 15:14; expcnt / lgkmcnt fields at "no wait"), and a
// compiler-level memory barrier with it
template <int N>
__device__ __forceinline__ void ti_wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
  asm volatile("" ::: "memory");
}

// NJ = 256-chunk rounds (16 bytes per chunk) per tap: ceil((footprint + 1) x 8 / 256)
template <int NJ>
__global__ __launch_bounds__(320, 4) void tap_interp_kernel(TIArgs p) {
  extern __shared__ __attribute__((aligned(16))) float zs[];          // [NBUF][fq_cap][CB]
  const int tid = threadIdx.x;
  int wg = blockIdx.x;
  {                                                    // XCD-aware: the eight L2s each take a contiguous run of tiles (shared halos)
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tx = wg % p.tiles_x;
  wg /= p.tiles_x;
  const int ty = wg % p.tiles_y, b = wg / p.tiles_y;
  const int cb0 = blockIdx.y * CB;
  const int Y0 = ty * TY, X0 = tx * TX;
  // low-resolution footprint of the tile with its one-pixel tap halo
  const int ya = max(Y0 - 1, 0), yb = min(Y0 + TY, p.H - 1), xa = max(X0 - 1, 0), xb = min(X0 + TX, p.W - 1);
  const int qy0 = (int)(p.sh * ya), qx0 = (int)(p.sw * xa);
  const int qy1 = min((int)(p.sh * yb) + 1, p.h - 1), qx1 = min((int)(p.sw * xb) + 1, p.w - 1);
  const int fw = qx1 - qx0 + 1, fq = (qy1 - qy0 + 1) * fw;           // < NJ * 32 = fq_cap (checked on the host)
  const int bufstride = p.fq_cap * CB;

  if (tid >= 256) {
    // =========================== PRODUCER wavefront: LDS-DMA issuer ===========================
    // One instruction moves 64 chunks = 8 footprint pixels x 128 bytes to wave-uniform base + 16 lane; 4 NJ instructions
    // per tap.  Three buffers, two taps ahead: in interval t (consumers on buffer t mod 3) tap t+2 is issued into the
    // buffer read in interval t-1 and only tap t+1 -- issued a whole interval earlier -- is waited for (counted vmcnt).
    const int lane = tid & 63;
    const int hp = p.h - 2 * p.zpad, wp = p.w - 2 * p.zpad;           // the stored grid
    const int c4 = (lane & 7) * 4;
    const int c4s = cb0 + c4 < p.Cout ? c4 : 0;                       // channel tail: lanes past Cout re-fetch the block's first group
    const float* zb = p.z + (long)b * hp * wp * 9 * p.Cout + cb0 + c4s;
    const float* zbr = p.zborder + cb0 + c4s;                         // only dereferenced for border-ring pixels
    int soff[4 * NJ];                                                 // element offset into the image's z (tap 0), -1 = border ring
#pragma unroll
    for (int j = 0; j < 4 * NJ; ++j) {
      int q = (lane >> 3) + 8 * j;
      q = q < fq ? q : 0;                                             // slots past the footprint hold its first pixel (finite data)
      const int qy = qy0 + q / fw - p.zpad, qx = qx0 + q % fw - p.zpad;
      soff[j] = (unsigned)qy < (unsigned)hp && (unsigned)qx < (unsigned)wp ? (qy * wp + qx) * 9 * p.Cout : -1;
    }
    auto issue = [&](int t) {
      __attribute__((address_space(3))) float* dst = (__attribute__((address_space(3))) float*)zs + (t % NBUF) * bufstride;
#pragma unroll
      for (int j = 0; j < 4 * NJ; ++j) {
        const float* src = soff[j] >= 0 ? zb + soff[j] + t * p.Cout : zbr + t * p.Cout;
        __builtin_amdgcn_global_load_lds((ti_gptr)src, (ti_lptr)(dst + j * 256), 16, 0, 0);
      }
    };
    issue(0);
    issue(1);
    ti_wait_vm<4 * NJ>();
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (t + 2 < 9) {
        issue(t + 2);
        ti_wait_vm<4 * NJ>();                           // tap t+1 has landed, tap t+2 may stay in flight
      } else {
        ti_wait_vm<0>();
      }
      __builtin_amdgcn_s_barrier();
    }
    return;
  }

  // =========================== CONSUMERS ===========================
  // items: channel group cg (4 channels) of pixels (Y0 + (tid >> 7) + 2 i, X0 + ((tid >> 3) & 15)), i = 0..3.
  // Coefficients are zero where the tap falls outside the image (zero padding of the convolution).
  const int cg = (tid & 7) * 4;
  int xo[3];                                                          // LDS float offset: buffer dx, column x0, channel group
  float wx0[3], wx1[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int X = X0 + ((tid >> 3) & 15) + d - 1;
    const bool ok = (unsigned)X < (unsigned)p.W;
    const float sx = p.sw * (ok ? X : 0);
    const int x0 = (int)sx;
    const float w1 = x0 < p.w - 1 ? sx - (float)x0 : 0.f;             // last column: ATen's x1 = x0, i.e. no right-hand share
    xo[d] = d * bufstride + (ok ? (x0 - qx0) * CB : 0) + cg;
    wx0[d] = ok ? 1.0f - w1 : 0.f;
    wx1[d] = ok ? w1 : 0.f;
  }

  float4 acc[4], sv[4];                                // sv: the skip part of this item's pixels, requested before the taps
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int Y = Y0 + (tid >> 7) + 2 * i, X = X0 + ((tid >> 3) & 15);
    const bool ok = p.s != nullptr && Y < p.H && X < p.W && cb0 + cg < p.Cout;
    sv[i] = ok ? ld4(p.s + (((long)b * p.H + Y) * p.W + X) * p.Cout + cb0 + cg) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);

  __builtin_amdgcn_s_barrier();                        // tap 0 is in LDS
  asm volatile("" ::: "memory");
#pragma unroll 1
  for (int dy = 0; dy < 3; ++dy) {
    int iy0[4], iy1[4];
    float hy0[4], hy1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int Y = Y0 + (tid >> 7) + 2 * i + dy - 1;
      const bool ok = (unsigned)Y < (unsigned)p.H;
      const float sy = p.sh * (ok ? Y : 0);
      const int y0 = (int)sy, y1 = y0 + (y0 < p.h - 1 ? 1 : 0);
      const float h1 = sy - (float)y0;
      iy0[i] = ok ? (y0 - qy0) * fw * CB : 0;
      iy1[i] = ok ? (y1 - qy0) * fw * CB : 0;
      hy0[i] = ok ? 1.0f - h1 : 0.f;
      hy1[i] = ok ? h1 : 0.f;
    }
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {                   // tap 3 dy + dx lives in buffer dx
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float* r0p = zs + iy0[i] + xo[dx];
        const float* r1p = zs + iy1[i] + xo[dx];
        const float4 v00 = *reinterpret_cast<const float4*>(r0p);
        const float4 v01 = *reinterpret_cast<const float4*>(r0p + CB);
        const float4 v10 = *reinterpret_cast<const float4*>(r1p);
        const float4 v11 = *reinterpret_cast<const float4*>(r1p + CB);
        // four fused multiply-adds per channel INTO the accumulator (two channels per v_pk_fma_f32): the sum-then-add form cost
        // a v_pk_mul + a v_pk_add more per channel pair, a fifth of the loop's vector instructions in a kernel whose consumers are
        // bound by them (profiles/r05_sq.json: 1024 per wavefront item, vector pipe ~50 % busy beside HBM at ~55 %)
        const float a = hy0[i] * wx0[dx], bq = hy0[i] * wx1[dx], c = hy1[i] * wx0[dx], d = hy1[i] * wx1[dx];
        acc[i].x = fmaf(d, v11.x, fmaf(c, v10.x, fmaf(bq, v01.x, fmaf(a, v00.x, acc[i].x))));
        acc[i].y = fmaf(d, v11.y, fmaf(c, v10.y, fmaf(bq, v01.y, fmaf(a, v00.y, acc[i].y))));
        acc[i].z = fmaf(d, v11.z, fmaf(c, v10.z, fmaf(bq, v01.z, fmaf(a, v00.z, acc[i].z))));
        acc[i].w = fmaf(d, v11.w, fmaf(c, v10.w, fmaf(bq, v01.w, fmaf(a, v00.w, acc[i].w))));
      }
      __builtin_amdgcn_sched_barrier(0);               // keep each tap's arithmetic with its LDS reads
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this tap's LDS reads are done before the buffer is handed back
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  }

  const int n = cb0 + cg;
  if (n >= p.Cout) return;
  const int Yt = Y0 + (tid >> 7), X = X0 + ((tid >> 3) & 15);
  switch (p.act) {                                                    // uniform
    case OCV_ACT_LEAKY_RELU: ti_store<OCV_ACT_LEAKY_RELU>(p, acc, sv, b, Yt, X, n); break;
    case OCV_ACT_SILU: ti_store<OCV_ACT_SILU>(p, acc, sv, b, Yt, X, n); break;
    case OCV_ACT_RELU: ti_store<OCV_ACT_RELU>(p, acc, sv, b, Yt, X, n); break;
    default: ti_store<OCV_ACT_NONE>(p, acc, sv, b, Yt, X, n); break;
  }
}

// + skip part + bias, activation, fp32 and / or hl32 split store of one item's four pixels (rows Y, Y+2, Y+4, Y+6)
template <int ACT>
__device__ __forceinline__ void ti_store(const TIArgs& p, const float4 (&acc)[4], const float4 (&sv)[4], int b, int Y, int X, int n) {
  if (X >= p.W) return;
  const float4 bv = p.bias != nullptr ? ld4(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (Y + 2 * i >= p.H) break;
    const long pix = ((long)b * p.H + Y + 2 * i) * p.W + X;
    float4 v = acc[i];
    v.x += sv[i].x; v.y += sv[i].y; v.z += sv[i].z; v.w += sv[i].w;
    v.x = ti_act<ACT>(v.x + bv.x); v.y = ti_act<ACT>(v.y + bv.y);
    v.z = ti_act<ACT>(v.z + bv.z); v.w = ti_act<ACT>(v.w + bv.w);
    if (p.y != nullptr) *reinterpret_cast<float4*>(p.y + pix * p.Cout + n) = v;
    if (p.yhl != nullptr) {
      const float f[4] = {v.x, v.y, v.z, v.w};
      ti_bf16x4 hi, lo;
      unsigned short hb_[4], lb_[4];
      if (p.f16) {                                                 // (uniform: one scalar branch per store)
        ocv_range_note(p.range_flag, fmaxf(fmaxf(fabsf(f[0]), fabsf(f[1])), fmaxf(fabsf(f[2]), fabsf(f[3]))));
#pragma unroll
        for (int e = 0; e < 4; ++e) ocv_split1<true>(f[e], hb_[e], lb_[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) ocv_split1<false>(f[e], hb_[e], lb_[e]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        hi[e] = __builtin_bit_cast(__bf16, hb_[e]);
        lo[e] = __builtin_bit_cast(__bf16, lb_[e]);
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
  return ti_footprint(h, w, H, W) < FQ ? 1 : 0;                     // + the one spare slot the right-hand neighbour may touch
}

extern "C" int ocv_tap_interp_combine_fwd(const float* z, int h, int w, int zpad, const float* zborder, const float* s,
                                          const float* bias, float* y, void* y_hl, int B, int H, int W, int Cout, int act,
                                          ocv_stream_t stream) {
  return ocv_tap_interp_combine_x_fwd(z, h, w, zpad, zborder, s, bias, y, y_hl, 0, B, H, W, Cout, act, stream);
}

namespace {
int ti_launch(const char* who, const float* z, int h, int w, int zpad, const float* zborder, const float* s, const float* bias, float* y,
              void* y_hl, int hl_f16, int B, int H, int W, int Cout, int act, ocv_stream_t stream) {
  OCV_CHECK_ARG(z && (y || y_hl), "%s: null pointer", who);
  OCV_CHECK_ARG(hl_f16 == 0 || hl_f16 == 1, "%s: hl_f16 must be 0 (bf16 pairs) or 1 (fp16 pairs)", who);
  OCV_CHECK_ARG((zpad == 0 && zborder == nullptr) || (zpad == 1 && zborder != nullptr && h >= 3 && w >= 3 && ocv_aligned16(zborder)),
                "%s: zpad must be 0 (no border vector) or 1 (with a 16-byte aligned border vector, h, w >= 3)", who);
  OCV_CHECK_ARG(B >= 1 && h >= 1 && w >= 1 && H >= 1 && W >= 1 && Cout >= 4 && Cout % 4 == 0,
                "%s: bad sizes (Cout must be a multiple of 4, got %d)", who, Cout);
  OCV_CHECK_ARG(act >= 0 && act <= 3, "%s: unknown activation %d", who, act);
  OCV_CHECK_ARG((long)h * w * 9 * Cout < (1L << 31), "%s: one image's tap products must stay below 2^31 elements "
                "(32-bit offsets inside the kernel)", who);
  OCV_CHECK_ARG(ocv_aligned16(z) && ocv_aligned16(s) && ocv_aligned16(bias) && ocv_aligned16(y) && ocv_aligned16(y_hl),
                "%s: operands must be 16-byte aligned", who);
  OCV_CHECK_ARG(ocv_tap_interp_supported(h, w, H, W, Cout), "%s: the low-resolution footprint of an output tile "
                "exceeds the staging buffer (h=%d w=%d H=%d W=%d): not an up-sampling by ~2 or more", who, h, w, H, W);
  const int nj = ocv_cdiv(ti_footprint(h, w, H, W) + 1, 32);       // 1..6 staging rounds per tap (footprint + one spare slot)
  TIArgs a{z, zborder, s, bias, y, (__bf16*)y_hl, h, w, H, W, Cout, (Cout + 31) / 32 * 32, act, zpad,
           H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f,
           ocv_cdiv(W, TX), ocv_cdiv(H, TY), nj * 32, hl_f16,
           (hl_f16 && y_hl != nullptr) ? ocv_range_flag_current() : nullptr};
  const long nwg = (long)a.tiles_x * a.tiles_y * B;
  OCV_CHECK_ARG(nwg < (1L << 31) && ocv_cdiv(Cout, CB) <= 65535, "%s: grid too large", who);
  if (y_hl != nullptr && Cout % 32 != 0) {
    const int zrc = ocv_zero_async(y_hl, ocv_split_act_elems(B, H, W, Cout) * sizeof(__bf16), (hipStream_t)stream);   // (a launch, not a memset node: common.hpp)
    if (zrc != 0) return zrc;
  }
  const dim3 grid((unsigned)nwg, ocv_cdiv(Cout, CB));
  const size_t buf = (size_t)a.fq_cap * CB * sizeof(float);
  const size_t lds = NBUF * buf;
  const void* fn = nullptr;
  switch (nj) {
#define OCV_TI_CASE(NJ) case NJ: fn = reinterpret_cast<const void*>(&tap_interp_kernel<NJ>); \
    hipLaunchKernelGGL((tap_interp_kernel<NJ>), grid, dim3(320), lds, (hipStream_t)stream, a); break;
    OCV_TI_CASE(1) OCV_TI_CASE(2) OCV_TI_CASE(3) OCV_TI_CASE(4) OCV_TI_CASE(5)
#undef OCV_TI_CASE
    default: {                                                      // nj == 6: 72 KB of dynamic LDS, above the 64 KB default limit
      fn = reinterpret_cast<const void*>(&tap_interp_kernel<6>);
      const hipError_t attr = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      OCV_CHECK_ARG(attr == hipSuccess, "%s: hipFuncSetAttribute failed: %s", who, hipGetErrorString(attr));
      hipLaunchKernelGGL((tap_interp_kernel<6>), grid, dim3(320), lds, (hipStream_t)stream, a);
    }
  }
  (void)fn;
  OCV_CHECK_LAUNCH(who);
  return 0;
}
}  // namespace

extern "C" int ocv_tap_interp_combine_x_fwd(const float* z, int h, int w, int zpad, const float* zborder, const float* s,
                                            const float* bias, float* y, void* y_hl, int hl_f16, int B, int H, int W, int Cout,
                                            int act, ocv_stream_t stream) {
  return ti_launch("ocv_tap_interp_combine_fwd", z, h, w, zpad, zborder, s, bias, y, y_hl, hl_f16, B, H, W, Cout, act, stream);
}
