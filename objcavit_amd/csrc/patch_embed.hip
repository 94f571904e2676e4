// 16x16 / stride-16 patch-embedding convolution for gfx950, written as an
// im2row-free split-K GEMM:  tokens[B*S, E] = patches[B*S, C*256] . W[E, C*256]^T
// with the patch rows gathered straight from the NCHW feature map.
//
// Memory behaviour (the reason for the layout): for a fixed (b, c, y) the
// feature-map row fmap[b][c][y][0..w) is contiguous and feeds w/16 consecutive
// patches with 16 floats (64 B) each, so a wavefront that walks consecutive
// patches of one patch row issues 16-byte-per-lane loads over one contiguous
// 1-2 KiB run: the 39 MB/img feature map is read exactly once, fully coalesced.
//
// Tiling: workgroup = 4 wavefronts = 64 patches x 128 outputs; K is split over
// blockIdx.y in whole channels (K-slice = C/ksplit channels = 256*C/ksplit
// values) so that 64-row x ksplit tiles cover the chip; each K step is one
// channel x 4 patch rows (64 values).  The partial sums go to a workspace slab
// per K-slice and a second tiny kernel adds the slabs in fixed order (bitwise
// reproducible, no float atomics) together with bias and the positional
// embedding, writing tokens as [B, S, E].
#include "common.hpp"
#include "../../include/objcavit_hip.h"

namespace {

constexpr int PS = 16;                 // patch size
constexpr int BM = 64, BN = 128, BK = 64, LDT = BK + 1;

struct PEArgs {
  const float* fmap; const float* W;
  float* part;                 // [ksplit][M][E]
  int B, C, h, w, gh, gw, M;   // M = B * gh * gw
  int cper;                    // channels per K-slice
};

template <bool VEC>
__global__ __launch_bounds__(256) void patch_embed_partial_kernel(PEArgs p) {
  __shared__ float As[BM][LDT];
  __shared__ float Ws[BN][LDT];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * BM;
  const int S = p.gh * p.gw;
  const long plane = (long)p.h * p.w;
  const int Ktot = p.C * PS * PS;

  // this thread's patch for the A-tile loads: patch = tid / 4, 4 floats at column (tid % 4) * 4
  const int lp = tid >> 2, j4 = (tid & 3) * 4;
  const int gm = m0 + lp;
  const bool pok = gm < p.M;
  long abase = 0;
  if (pok) {
    const int b = gm / S, s = gm - b * S;
    const int ph = s / p.gw, pw = s - ph * p.gw;
    abase = (long)b * p.C * plane + (long)(ph * PS) * p.w + pw * PS + j4;
  }

  f32x16 acc0 = {0}, acc1 = {0};
  const int c_lo = blockIdx.y * p.cper, c_hi = c_lo + p.cper;

  for (int c = c_lo; c < c_hi; ++c) {
    for (int i0 = 0; i0 < PS; i0 += 4) {
      // ---- A tile: 64 patches x (4 rows x 16 cols)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pok) {
          const float* src = p.fmap + abase + (long)c * plane + (long)(i0 + i) * p.w;
          if (VEC) t = ld4(src);
          else t = make_float4(src[0], src[1], src[2], src[3]);
        }
        float* d = &As[lp][i * PS + j4];
        d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
      }
      // ---- W tile: 128 outputs x 64 consecutive k
      const int kbase = c * PS * PS + i0 * PS;
#pragma unroll
      for (int pass = 0; pass < 8; ++pass) {
        const int n = pass * 16 + (tid >> 4), kk = (tid & 15) * 4;
        float4 t = ld4(p.W + (long)n * Ktot + kbase + kk);
        float* d = &Ws[n][kk];
        d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
      }
      __syncthreads();
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
  // enough 64-row x ksplit workgroups for ~2 per CU; ksplit must divide C
  const int mt = ocv_cdiv(M, BM);
  int ks = 1;
  while (ks < C && mt * ks < 512 && C % (ks * 2) == 0) ks *= 2;
  return ks;
}

}  // namespace

extern "C" size_t ocv_patch_embed_workspace_bytes(int B, int C, int h, int w, int E) {
  if (B < 1 || C < 1 || h < PS || w < PS || E != BN) return 0;
  const long M = (long)B * (h / PS) * (w / PS);
  return (size_t)pick_ksplit((int)M, C) * M * BN * sizeof(float);
}

extern "C" int ocv_patch_embed_fwd(const float* fmap, const float* W, const float* bias, const float* pos, long pos_bs,
                                   float* out, int B, int C, int h, int w, int E, void* workspace,
                                   size_t workspace_bytes, ocv_stream_t stream) {
  OCV_CHECK_ARG(fmap && W && out && workspace, "ocv_patch_embed_fwd: null pointer");
  OCV_CHECK_ARG(E == BN, "ocv_patch_embed_fwd: E must be %d (got %d)", BN, E);
  OCV_CHECK_ARG(B >= 1 && C >= 1 && h >= PS && w >= PS, "ocv_patch_embed_fwd: bad sizes B=%d C=%d h=%d w=%d", B, C, h, w);
  OCV_CHECK_ARG(ocv_aligned16(W) && ocv_aligned16(workspace) && ocv_aligned16(out), "ocv_patch_embed_fwd: W / workspace / out must be 16-byte aligned");
  const size_t need = ocv_patch_embed_workspace_bytes(B, C, h, w, E);
  OCV_CHECK_ARG(workspace_bytes >= need, "ocv_patch_embed_fwd: workspace too small (%zu < %zu)", workspace_bytes, need);
  PEArgs a;
  a.fmap = fmap; a.W = W; a.part = (float*)workspace;
  a.B = B; a.C = C; a.h = h; a.w = w; a.gh = h / PS; a.gw = w / PS;
  a.M = B * a.gh * a.gw;
  const int ks = pick_ksplit(a.M, C);
  a.cper = C / ks;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(ocv_cdiv(a.M, BM), ks), block(256);
  const bool vec = (w % 4 == 0) && ocv_aligned16(fmap);
  if (vec) hipLaunchKernelGGL((patch_embed_partial_kernel<true>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((patch_embed_partial_kernel<false>), grid, block, 0, st, a);
  OCV_CHECK_LAUNCH("ocv_patch_embed_fwd(partial)");
  const long n4 = (long)a.M * (BN / 4);
  hipLaunchKernelGGL(patch_embed_reduce_kernel, dim3(ocv_cdiv(n4, 256)), dim3(256), 0, st, a.part, ks, (long)a.M,
                     a.gh * a.gw, bias, pos, pos_bs, out);
  OCV_CHECK_LAUNCH("ocv_patch_embed_fwd(reduce)");
  return 0;
}
