// 16x16 / stride-16 patch-embedding convolution for gfx950, written as an
// im2row-free split-K GEMM:  tokens[B*S, E] = patches[B*S, C*256] . W[E, C*256]^T
// with the patch rows gathered straight from the feature map (NCHW or NHWC).
//
// Memory behaviour (the reason for the tiling):
//  * NCHW: for a fixed (b, c, y) the row fmap[b][c][y][0..w) is contiguous and
//    feeds w/16 consecutive patches with 16 floats (64 B) each; a K step is one
//    channel x 4 patch rows.
//  * NHWC (channels_last, what MIOpen's fast fp32 igemm decoder convolutions
//    produce): a pixel's 128 channels are contiguous (512 B); a K step is one
//    pixel (i, j) x 64 channels = 256 contiguous bytes per patch, and the weight
//    is consumed in its channels_last storage order [E][16][16][C].
//  Either way every byte of the 39 MB/img feature map is read exactly once with
//  16-byte-per-lane loads.
//
// Tiling: workgroup = 4 wavefronts = 64 patches x 128 outputs (2x2 waves, each
// 32 x 64 = two 32x32 fp32 MFMA accumulators); K is split over blockIdx.y in
// whole 64-wide K steps so that 64-row x ksplit tiles cover the chip.  The next
// K step's global loads are issued before the current step's 64 MFMAs and
// written to LDS afterwards (register prefetch), so HBM latency overlaps the
// matrix pipe.  Partial sums go to one workspace slab per K-slice; a second
// tiny kernel adds the slabs in fixed order (bitwise reproducible, no float
// atomics) together with bias and the positional embedding -> tokens [B, S, E].
#include "common.hpp"
#include "../../include/objcavit_hip.h"

namespace {

constexpr int PS = 16;                 // patch size
constexpr int BM = 64, BN = 128, BK = 64, LDT = BK + 1;

struct PEArgs {
  const float* fmap; const float* W;
  float* part;                 // [ksplit][M][E]
  int B, C, h, w, gh, gw, M;   // M = B * gh * gw
  int kt_per;                  // 64-wide K steps per K-slice
};

// A-operand source of K step kt for this thread's patch
//   NCHW: kt -> (c = kt / 4, rows i0 = 4 * (kt % 4)); the thread loads 4 rows x float4
//   NHWC: kt -> (pixel q = kt / (C/64), channel block (kt % (C/64)) * 64); 4 float4 of one 256-B run
template <bool NHWC, bool VEC>
__device__ __forceinline__ void load_a(const PEArgs& p, int kt, long abase, bool pok, int tid, float4 (&ra)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!pok) return;
  if (!NHWC) {
    const int c = kt >> 2, i0 = (kt & 3) * 4;
    const float* src = p.fmap + abase + (long)c * p.h * p.w + (long)i0 * p.w;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float* s = src + (long)i * p.w;
      ra[i] = VEC ? ld4(s) : make_float4(s[0], s[1], s[2], s[3]);
    }
  } else {
    const int cb = p.C / 64;
    const int q = kt / cb, c0 = (kt - q * cb) * 64;
    const int i = q >> 4, j = q & 15;
    const float* src = p.fmap + abase + ((long)i * p.w + j) * p.C + c0;
#pragma unroll
    for (int e = 0; e < 4; ++e) ra[e] = ld4(src + e * 16);          // columns (tid&3)*4 + 16 e
  }
}

template <bool NHWC, bool VEC>
__global__ __launch_bounds__(256) void patch_embed_partial_kernel(PEArgs p) {
  __shared__ float As[BM][LDT];
  __shared__ float Ws[BN][LDT];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * BM;
  const int S = p.gh * p.gw;
  const long Ktot = (long)p.C * PS * PS;

  // this thread's patch for the A-tile loads: patch = tid / 4, float4 column (tid % 4) * 4
  const int lp = tid >> 2, j4 = (tid & 3) * 4;
  const int gm = m0 + lp;
  const bool pok = gm < p.M;
  long abase = 0;
  if (pok) {
    const int b = gm / S, s = gm - b * S;
    const int ph = s / p.gw, pw = s - ph * p.gw;
    if (!NHWC) abase = (long)b * p.C * p.h * p.w + (long)(ph * PS) * p.w + pw * PS + j4;
    else abase = (((long)b * p.h + ph * PS) * p.w + pw * PS) * p.C + j4;
  }
  // W tile: thread -> (n = pass*16 + tid/16, float4 column (tid%16)*4)
  const float* wsrc = p.W + (long)(tid >> 4) * Ktot + (tid & 15) * 4;

  f32x16 acc0 = {0}, acc1 = {0};
  const int kt_lo = blockIdx.y * p.kt_per, kt_hi = kt_lo + p.kt_per;

  float4 ra[4], rw[8];
  load_a<NHWC, VEC>(p, kt_lo, abase, pok, tid, ra);
#pragma unroll
  for (int pass = 0; pass < 8; ++pass) rw[pass] = ld4(wsrc + (long)pass * 16 * Ktot + (long)kt_lo * BK);

  for (int kt = kt_lo; kt < kt_hi; ++kt) {
    // ---- registers -> LDS
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float* d = NHWC ? &As[lp][j4 + 16 * i] : &As[lp][i * PS + j4];
      d[0] = ra[i].x; d[1] = ra[i].y; d[2] = ra[i].z; d[3] = ra[i].w;
    }
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      float* d = &Ws[pass * 16 + (tid >> 4)][(tid & 15) * 4];
      d[0] = rw[pass].x; d[1] = rw[pass].y; d[2] = rw[pass].z; d[3] = rw[pass].w;
    }
    __syncthreads();
    // ---- prefetch the next K step while this one is multiplied
    if (kt + 1 < kt_hi) {
      load_a<NHWC, VEC>(p, kt + 1, abase, pok, tid, ra);
#pragma unroll
      for (int pass = 0; pass < 8; ++pass) rw[pass] = ld4(wsrc + (long)pass * 16 * Ktot + (long)(kt + 1) * BK);
    }
    const float* arow = &As[wm * 32 + l31][hh];
    const float* w0 = &Ws[wn * 64 + l31][hh];
    const float* w1 = &Ws[wn * 64 + 32 + l31][hh];
#pragma unroll
    for (int s = 0; s < BK / 2; ++s) {
      const float a = arow[2 * s];
      acc0 = mfma_32x32x2(a, w0[2 * s], acc0);
      acc1 = mfma_32x32x2(a, w1[2 * s], acc1);
    }
    __syncthreads();
  }

  float* dst = p.part + (long)blockIdx.y * p.M * BN;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 32 + acc_row(r, hh);
    if (m < p.M) {
      dst[(long)m * BN + wn * 64 + l31] = acc0[r];
      dst[(long)m * BN + wn * 64 + 32 + l31] = acc1[r];
    }
  }
}

// out[m][e] = bias[e] + pos[...] + sum_ks part[ks][m][e]   (float4 per thread)
__global__ __launch_bounds__(256) void patch_embed_reduce_kernel(const float* __restrict__ part, int ksplit, long M,
                                                                 int S, const float* __restrict__ bias,
                                                                 const float* __restrict__ pos, long pos_bs,
                                                                 float* __restrict__ out) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;     // float4 index
  if (idx >= M * (BN / 4)) return;
  const long m = idx / (BN / 4);
  const int e = (int)(idx - m * (BN / 4)) * 4;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int ks = 0; ks < ksplit; ++ks) {
    const float4 t = ld4(part + ((long)ks * M + m) * BN + e);
    a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
  }
  if (bias) { a.x += bias[e]; a.y += bias[e + 1]; a.z += bias[e + 2]; a.w += bias[e + 3]; }
  if (pos) {
    const long b = m / S, s = m - b * S;
    const float* pp = pos + b * pos_bs + s * BN + e;
    a.x += pp[0]; a.y += pp[1]; a.z += pp[2]; a.w += pp[3];
  }
  *reinterpret_cast<float4*>(out + m * BN + e) = a;
}

int pick_ksplit(int M, int C) {
  // enough 64-row x ksplit workgroups for ~2 per CU; ksplit is a power of two dividing the C*4 K steps
  const int mt = ocv_cdiv(M, BM), kts = C * 4;
  int ks = 1;
  while (ks < kts && mt * ks < 1024 && kts % (ks * 2) == 0) ks *= 2;
  return ks;
}

}  // namespace

extern "C" size_t ocv_patch_embed_workspace_bytes(int B, int C, int h, int w, int E) {
  if (B < 1 || C < 1 || h < PS || w < PS || E != BN) return 0;
  const long M = (long)B * (h / PS) * (w / PS);
  return (size_t)pick_ksplit((int)M, C) * M * BN * sizeof(float);
}

extern "C" int ocv_patch_embed_fwd(const float* fmap, int channels_last, const float* W, const float* bias,
                                   const float* pos, long pos_bs, float* out, int B, int C, int h, int w, int E,
                                   void* workspace, size_t workspace_bytes, ocv_stream_t stream) {
  OCV_CHECK_ARG(fmap && W && out && workspace, "ocv_patch_embed_fwd: null pointer");
  OCV_CHECK_ARG(E == BN, "ocv_patch_embed_fwd: E must be %d (got %d)", BN, E);
  OCV_CHECK_ARG(B >= 1 && C >= 1 && h >= PS && w >= PS, "ocv_patch_embed_fwd: bad sizes B=%d C=%d h=%d w=%d", B, C, h, w);
  OCV_CHECK_ARG(ocv_aligned16(W) && ocv_aligned16(workspace) && ocv_aligned16(out), "ocv_patch_embed_fwd: W / workspace / out must be 16-byte aligned");
  OCV_CHECK_ARG(!channels_last || (C % 64 == 0 && ocv_aligned16(fmap)), "ocv_patch_embed_fwd: channels_last needs C %% 64 == 0 and a 16-byte aligned map");
  const size_t need = ocv_patch_embed_workspace_bytes(B, C, h, w, E);
  OCV_CHECK_ARG(workspace_bytes >= need, "ocv_patch_embed_fwd: workspace too small (%zu < %zu)", workspace_bytes, need);
  PEArgs a;
  a.fmap = fmap; a.W = W; a.part = (float*)workspace;
  a.B = B; a.C = C; a.h = h; a.w = w; a.gh = h / PS; a.gw = w / PS;
  a.M = B * a.gh * a.gw;
  const int ks = pick_ksplit(a.M, C);
  a.kt_per = C * 4 / ks;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(ocv_cdiv(a.M, BM), ks), block(256);
  if (channels_last) {
    hipLaunchKernelGGL((patch_embed_partial_kernel<true, true>), grid, block, 0, st, a);
  } else {
    const bool vec = (w % 4 == 0) && ocv_aligned16(fmap);
    if (vec) hipLaunchKernelGGL((patch_embed_partial_kernel<false, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((patch_embed_partial_kernel<false, false>), grid, block, 0, st, a);
  }
  OCV_CHECK_LAUNCH("ocv_patch_embed_fwd(partial)");
  const long n4 = (long)a.M * (BN / 4);
  hipLaunchKernelGGL(patch_embed_reduce_kernel, dim3(ocv_cdiv(n4, 256)), dim3(256), 0, st, a.part, ks, (long)a.M,
                     a.gh * a.gw, bias, pos, pos_bs, out);
  OCV_CHECK_LAUNCH("ocv_patch_embed_fwd(reduce)");
  return 0;
}
