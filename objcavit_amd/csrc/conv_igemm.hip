// Implicit-GEMM convolution (k = 1 or 3, stride 1, "same" zero padding) on
// NHWC fp32 activations for gfx950, computed on the bf16 matrix cores with
// SPLIT-bf16 operands and fp32 accumulation:
//
//     a = a_hi + a_lo,  b = b_hi + b_lo   (each part bf16, a_lo = bf16(a - a_hi))
//     a * b  ~=  a_hi b_hi + a_hi b_lo + a_lo b_hi            (3 MFMAs, fp32 accumulate)
//
// hi + lo carries 16 significant bits, the dropped a_lo b_lo term is 2^-18
// relative, so a product is good to ~1e-5 relative and a K-long sum to ~1e-6 of
// its magnitude -- two orders inside the 1e-3 depth tolerance of the path and
// measured end to end in the parity tests -- while the three
// v_mfma_f32_32x32x16_bf16 issues cost 96 cycles per 32x32x16 block against
// 512 cycles for the eight exact v_mfma_f32_32x32x2_f32 issues: 5.3x the fp32
// matrix rate (gfx950 has no xf32/TF32 path).  This is row N1 of SURVEY.md section 8:
// the UNet decoder's 3x3 convolutions are 83 % of the forward's FLOPs
// (modules/DenseFeatureExtractor.py:37-42,97,104-116) and MIOpen's fp32
// implicit GEMM already sits at 85 % of the fp32 matrix peak there.
//
// GEMM view: M = B*H*W output pixels, N = Cout, K = taps x Cin.  The A operand
// is gathered on the fly (tap offset, zero outside the image) from one or two
// NHWC tensors -- a second tensor acts as a virtual channel concat, which is how
// UpSampleWithSkip feeds [upsampled, skip] without materialising the cat --
// and split into hi/lo in registers on its way to LDS.  Weights are split once
// on the host into two bf16 arrays laid out [tap][Cout][Cin rounded up to 32].
//
// Tile: 256 pixels x 128 channels per workgroup of 8 wavefronts (4 x 2, each
// 64 x 64 = 2 x 2 accumulators of 32x32); K advances 32 channels of one tap per
// step through a double-buffered LDS image (A_hi, A_lo, B_hi, B_lo; rows padded
// to 80 bytes so the 16-byte fragment reads of 16 consecutive rows land in 16
// different bank slots).  Per step and wavefront: 16 ds_read_b128, 24 MFMAs;
// the next step's global loads are issued before the MFMAs and converted /
// written to the other buffer after them; one barrier per step.
// Workgroup ids are remapped so that the N-tiles of one pixel tile run on the
// same XCD and share its A rows in that XCD's L2.
#include "common.hpp"
#include "../../include/objcavit_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int CBM = 256, CBN = 128, CBK = 32;
constexpr int ROWB = 80;                                  // bytes per LDS row (64 data + 16 pad)
constexpr int A_BYTES = CBM * ROWB, B_BYTES = CBN * ROWB;
constexpr int BUF_BYTES = 2 * A_BYTES + 2 * B_BYTES;      // 61440

struct ConvArgs {
  const float* x1; const float* x2;     // NHWC; x2 (nullable) is concatenated after x1's channels
  const __bf16* whi; const __bf16* wlo; // [taps][Cout][Cp]
  const float* bias; const float* res; float* y;
  int C1, C2, Cin, Cp, Cout, H, W, ks, act;
  long M;
  int mtiles, ntiles;
};

// Global loads whose completion is counted BY HAND.  hipcc's own s_waitcnt insertion cannot express "wait for the
// older of two in-flight register stages" across the loop back-edge (it emitted vmcnt(5..0), draining the younger
// stage too), so the loads are issued from inline asm -- invisible to that pass -- and each stage is retired by an
// explicit counted s_waitcnt whose asm statement takes the stage's registers as in/out operands: every consumer
// is data-dependent on the wait and cannot be scheduled above it.
__device__ __forceinline__ f32x4 gload16_async(const void* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}

__device__ __forceinline__ void split4(const f32x4 v, __bf16* hi, __bf16* lo) {
  const float f[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const __bf16 h = (__bf16)f[i];
    hi[i] = h;
    lo[i] = (__bf16)(f[i] - (float)h);
  }
}

__global__ __launch_bounds__(512) void conv_igemm_kernel(ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- XCD-aware, bijective workgroup -> tile map: consecutive tiles (N fastest) share an XCD
  const int nwg = p.mtiles * p.ntiles;
  int wg = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int mt = wg / p.ntiles, nt = wg - mt * p.ntiles;
  const long m0 = (long)mt * CBM;
  const int n0 = nt * CBN;

  // ---- A-gather role: thread -> (row = tid / 2, 16 channels at (tid & 1) * 16)
  const int arow = tid >> 1, acol = (tid & 1) * 16;
  const long am = m0 + arow;
  const bool avalid = am < p.M;
  int ay = 0, ax = 0;
  long apix = 0;
  if (avalid) {
    const long hw = (long)p.H * p.W;
    const long b = am / hw, rem = am - b * hw;
    ay = (int)(rem / p.W);
    ax = (int)(rem - (long)ay * p.W);
    apix = am;                            // NHWC pixel index == m
  }
  const int pad = p.ks >> 1;
  // ---- B role: thread -> (n = tid / 4, 8 channels at (tid & 3) * 8)
  const int brow = tid >> 2, bcol = (tid & 3) * 8;
  const int bn = min(n0 + brow, p.Cout - 1);
  const int taps = p.ks * p.ks;
  const int cchunks = p.Cp / CBK;
  const int nsteps = taps * cchunks;

  // Two register stages: the global loads of K step s+2 are issued while step s is multiplied and step s+1's loads
  // are still in flight, so every load has two full steps (>= 2 x 1536 matrix-pipe cycles) to land before it is
  // converted and written to LDS -- one step of cover was not enough under load (27 % -> see DESIGN.md).
  struct Stage { f32x4 ra[4]; f32x4 rbh, rbl; unsigned ok; };

  auto issue_loads = [&](int step, Stage& st) {
    const int tap = step / cchunks, c0 = (step - tap * cchunks) * CBK;
    const int ky = tap / p.ks, kx = tap - ky * p.ks;
    const int iy = ay + ky - pad, ix = ax + kx - pad;
    const bool inb = avalid && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    const long pix = apix + (long)(ky - pad) * p.W + (kx - pad);
    const float* src;
    int cend;
    if (c0 < p.C1) { src = p.x1 + pix * p.C1 + c0; cend = p.C1 - c0; }
    else { src = p.x2 + pix * p.C2 + (c0 - p.C1); cend = p.Cin - c0; }
    // Unconditional loads (out-of-image taps / channel tails read a safe address and are zeroed by a select):
    // a load inside an exec-masked branch makes hipcc fall back to s_waitcnt vmcnt(0), which would drain the
    // younger stage's loads as well and undo the two-step prefetch.
    // The zeroing select is deferred to write_lds (stage.ok): touching the loaded registers here would make the
    // compiler wait for the load at once.
    st.ok = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = acol + 4 * e;
      const bool ok = inb && c + 4 <= cend;
      st.ra[e] = gload16_async(ok ? src + c : p.x1);
      st.ok |= ok ? (1u << e) : 0u;
    }
    const long woff = ((long)tap * p.Cout + bn) * p.Cp + c0 + bcol;
    st.rbh = gload16_async(p.whi + woff);
    st.rbl = gload16_async(p.wlo + woff);
  };
  // retire a stage: N = number of YOUNGER loads that may stay in flight (6 = the other stage, 0 = none)
#define OCV_RETIRE(st, N)                                                                                              \
  asm volatile("s_waitcnt vmcnt(" #N ")"                                                                               \
               : "+v"(st.ra[0]), "+v"(st.ra[1]), "+v"(st.ra[2]), "+v"(st.ra[3]), "+v"(st.rbh), "+v"(st.rbl)           \
               :                                                                                                       \
               : "memory")

  auto write_lds = [&](int buf, Stage& st) {
    unsigned char* base = lds + buf * BUF_BYTES;
    __bf16 hi[16], lo[16];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool ok = (st.ok >> e) & 1u;
      const f32x4 t = st.ra[e];
      const f32x4 z = {ok ? t[0] : 0.f, ok ? t[1] : 0.f, ok ? t[2] : 0.f, ok ? t[3] : 0.f};
      split4(z, hi + 4 * e, lo + 4 * e);
    }
    unsigned char* ah = base + arow * ROWB + acol * 2;
    unsigned char* al = ah + A_BYTES;
    *reinterpret_cast<bf16x8*>(ah) = *reinterpret_cast<bf16x8*>(hi);
    *reinterpret_cast<bf16x8*>(ah + 16) = *reinterpret_cast<bf16x8*>(hi + 8);
    *reinterpret_cast<bf16x8*>(al) = *reinterpret_cast<bf16x8*>(lo);
    *reinterpret_cast<bf16x8*>(al + 16) = *reinterpret_cast<bf16x8*>(lo + 8);
    unsigned char* bh = base + 2 * A_BYTES + brow * ROWB + bcol * 2;
    *reinterpret_cast<f32x4*>(bh) = st.rbh;
    *reinterpret_cast<f32x4*>(bh + B_BYTES) = st.rbl;
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x16{0};

  auto compute = [&](int buf) {
    const unsigned char* base = lds + buf * BUF_BYTES;
    const unsigned char* pa = base + (wm * 64 + l31) * ROWB + hh * 16;
    const unsigned char* pb = base + 2 * A_BYTES + (wn * 64 + l31) * ROWB + hh * 16;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = *reinterpret_cast<const bf16x8*>(pa + i * 32 * ROWB + kk * 32);
        al[i] = *reinterpret_cast<const bf16x8*>(pa + A_BYTES + i * 32 * ROWB + kk * 32);
        bh[i] = *reinterpret_cast<const bf16x8*>(pb + i * 32 * ROWB + kk * 32);
        bl[i] = *reinterpret_cast<const bf16x8*>(pb + B_BYTES + i * 32 * ROWB + kk * 32);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
  };

  Stage s0, s1;
  issue_loads(0, s0);
  OCV_RETIRE(s0, 0);
  write_lds(0, s0);
  __syncthreads();
  if (nsteps > 1) issue_loads(1, s1);

  for (int step = 0; step < nsteps; step += 2) {
    // even step: multiply buffer 0; step+1 is in flight in s1; fetch step+2 into s0
    const bool more2 = step + 2 < nsteps;
    if (more2) issue_loads(step + 2, s0);
    compute(0);
    if (step + 1 < nsteps) {
      if (more2) OCV_RETIRE(s1, 6); else OCV_RETIRE(s1, 0);
      write_lds(1, s1);
    }
    __syncthreads();
    if (step + 1 >= nsteps) break;
    // odd step: multiply buffer 1; step+2 is in flight in s0; fetch step+3 into s1
    const bool more3 = step + 3 < nsteps;
    if (more3) issue_loads(step + 3, s1);
    compute(1);
    if (more2) {
      if (more3) OCV_RETIRE(s0, 6); else OCV_RETIRE(s0, 0);
      write_lds(0, s0);
    }
    __syncthreads();
  }
#undef OCV_RETIRE

  // ---- epilogue: bias, activation, optional residual, NHWC store (128-byte runs per half-wave)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + l31;
    const bool nok = n < p.Cout;
    const float bv = (p.bias != nullptr && nok) ? p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long m = m0 + wm * 64 + i * 32 + acc_row(r, hh);
        if (nok && m < p.M) {
          float v = acc[i][j][r] + bv;
          if (p.act == OCV_ACT_LEAKY_RELU) v = v > 0.f ? v : 0.01f * v;
          else if (p.act == OCV_ACT_SILU) v = v / (1.0f + fast_exp(-v));
          else if (p.act == OCV_ACT_RELU) v = fmaxf(v, 0.f);
          if (p.res != nullptr) v += p.res[m * p.Cout + n];
          p.y[m * p.Cout + n] = v;
        }
      }
  }
}

}  // namespace

extern "C" int ocv_conv_nhwc_fwd(const float* x1, int C1, const float* x2, int C2, const void* w_hi, const void* w_lo,
                                 const float* bias, const float* residual, float* y, int B, int H, int W, int Cout,
                                 int ksize, int act, ocv_stream_t stream) {
  OCV_CHECK_ARG(x1 && w_hi && w_lo && y, "ocv_conv_nhwc_fwd: null pointer");
  OCV_CHECK_ARG(ksize == 1 || ksize == 3, "ocv_conv_nhwc_fwd: kernel size must be 1 or 3 (got %d)", ksize);
  OCV_CHECK_ARG(B >= 1 && H >= 1 && W >= 1 && Cout >= 1 && C1 >= 4, "ocv_conv_nhwc_fwd: bad sizes");
  OCV_CHECK_ARG(C1 % 4 == 0 && (x2 == nullptr || (C2 >= 4 && C2 % 4 == 0 && C1 % CBK == 0)),
                "ocv_conv_nhwc_fwd: channel counts must be multiples of 4 (and C1 a multiple of %d when a second tensor is concatenated)", CBK);
  OCV_CHECK_ARG(act >= 0 && act <= 3, "ocv_conv_nhwc_fwd: unknown activation %d", act);
  OCV_CHECK_ARG(ocv_aligned16(x1) && ocv_aligned16(x2) && ocv_aligned16(w_hi) && ocv_aligned16(w_lo),
                "ocv_conv_nhwc_fwd: operands must be 16-byte aligned");
  ConvArgs a;
  a.x1 = x1; a.x2 = x2; a.whi = (const __bf16*)w_hi; a.wlo = (const __bf16*)w_lo;
  a.bias = bias; a.res = residual; a.y = y;
  a.C1 = C1; a.C2 = x2 ? C2 : 0; a.Cin = a.C1 + a.C2; a.Cp = (a.Cin + CBK - 1) / CBK * CBK;
  a.Cout = Cout; a.H = H; a.W = W; a.ks = ksize; a.act = act;
  a.M = (long)B * H * W;
  a.mtiles = ocv_cdiv(a.M, CBM); a.ntiles = ocv_cdiv(Cout, CBN);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv_igemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(conv_igemm_kernel, dim3(a.mtiles * a.ntiles), dim3(512), 2 * BUF_BYTES, (hipStream_t)stream, a);
  OCV_CHECK_LAUNCH("ocv_conv_nhwc_fwd");
  return 0;
}
