// Ragged object lists on the device, shape-static: what SelfAttnCrossAttn.forward does with its list of N_i x E object tokens
// (modules/ObjCAViT.py:180-183 pad_sequence(..., 1e-4) + mask, :192-194 the BACK-padded mask and FRONT-padded rows in front of the
// first cross-attention) from a per-image COUNT that lives in device memory -- so that a captured hipGraph
// serves any ragged object set up to its capacity (BASELINE configs[4]: detector output changes per image), and the eager path
// needs no per-count-tuple mask cache and none of the ~10 ATen pad / cat / copy launches.
//
//   ocv_object_tokens_pad_fwd   tokens [B][cap][E] (every row embedded, rows >= count are garbage) -> rows >= count replaced by the
//                               pad value, mask[b][j] = (j >= count[b])                                   (:180-183)
//   ocv_object_front_pad_fwd    encoded objects [B][cap][E] -> keys [B][S][E] = [ pad x (S - Nmax) | rows 0 .. Nmax - 1 ], mask
//                               [B][S] = (j >= count[b]); Nmax = the largest count of the image's GROUP of `group` consecutive
//                               images (the reference pads to the longest list of the batch it was called with: one group = one
//                               reference call, e.g. image batch + mirrored batch in one launch), or a caller-given value
//                               (a data-parallel shard passing the global batch's Nmax)                    (:192-194)
// Both are pure data movement (fp32, 16-byte accesses), one thread per channel quad.
#include "common.hpp"
#include "../../include/objcavit_hip.h"

namespace {

__global__ __launch_bounds__(256) void object_tokens_pad_kernel(const float* __restrict__ tok, const int* __restrict__ counts,
                                                                float pad, float* __restrict__ out, uint8_t* __restrict__ mask,
                                                                int B, int cap, int E4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * cap * E4) return;
  const long row = i / E4;
  const int b = (int)(row / cap), j = (int)(row - (long)b * cap);
  const int cnt = counts[b];
  const bool live = j < min(max(cnt, 1), cap);                  // counts are device data nobody has checked: [1, cap] by force, so an
                                                               // image always keeps one live key (a count of 0 = fully masked softmax = NaN)
  // ... and a count below 1 makes that one key the PAD row, not whatever the caller's buffer holds in row 0: defined and
  // deterministic (the reference's own padded rows, modules/ObjCAViT.py:180-183), never a plausible-looking stray object
  const float4 v = (live && cnt >= 1) ? ld4(tok + 4 * i) : make_float4(pad, pad, pad, pad);
  *reinterpret_cast<float4*>(out + 4 * i) = v;
  if (i - row * E4 == 0) mask[row] = live ? 0 : 1;
}

__global__ __launch_bounds__(256) void object_front_pad_kernel(const float* __restrict__ obj, const int* __restrict__ counts,
                                                               int group, int nmax_given, float pad, float* __restrict__ out,
                                                               uint8_t* __restrict__ kpm, int B, int cap, int S, int E4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * S * E4) return;
  const long row = i / E4;
  const int b = (int)(row / S), j = (int)(row - (long)b * S);
  int nmax = nmax_given;
  if (nmax <= 0) {                                             // the longest list of this image's group (uniform per image; <= 2 B loads)
    const int g0 = (b / group) * group, g1 = min(B, g0 + group);
    for (int g = g0; g < g1; ++g) nmax = max(nmax, counts[g]);
  }
  nmax = min(max(nmax, 1), cap);
  const int src = j - (S - nmax);                              // rows are padded at the FRONT (reference :194)
  const float4 v = src >= 0 ? ld4(obj + (((long)b * cap + src) * E4 + (i - row * E4)) * 4) : make_float4(pad, pad, pad, pad);
  *reinterpret_cast<float4*>(out + 4 * i) = v;
  if (i - row * E4 == 0) kpm[row] = j >= min(max(counts[b], 1), cap) ? 1 : 0;    // the mask at the BACK (reference :193); count in [1, cap] by force
}

}  // namespace

extern "C" int ocv_object_tokens_pad_fwd(const float* tokens, const int* counts, float pad_value, float* out, uint8_t* mask, int B,
                                         int capacity, int E, ocv_stream_t stream) {
  OCV_CHECK_ARG(tokens && counts && out && mask, "ocv_object_tokens_pad_fwd: null pointer");
  OCV_CHECK_ARG(B >= 1 && capacity >= 1 && E >= 4 && E % 4 == 0, "ocv_object_tokens_pad_fwd: bad sizes (E must be a multiple of 4)");
  OCV_CHECK_ARG(ocv_aligned16(tokens) && ocv_aligned16(out), "ocv_object_tokens_pad_fwd: tokens / out must be 16-byte aligned");
  const long n = (long)B * capacity * (E / 4);
  hipLaunchKernelGGL(object_tokens_pad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, tokens, counts,
                     pad_value, out, mask, B, capacity, E / 4);
  OCV_CHECK_LAUNCH("ocv_object_tokens_pad_fwd");
  return 0;
}

extern "C" int ocv_object_front_pad_fwd(const float* objects, const int* counts, int group, int nmax, float pad_value, float* out,
                                        uint8_t* key_padding_mask, int B, int capacity, int S, int E, ocv_stream_t stream) {
  OCV_CHECK_ARG(objects && counts && out && key_padding_mask, "ocv_object_front_pad_fwd: null pointer");
  OCV_CHECK_ARG(B >= 1 && capacity >= 1 && S >= capacity && E >= 4 && E % 4 == 0,
                "ocv_object_front_pad_fwd: bad sizes (more object rows (%d) than image tokens (%d), or E not a multiple of 4)", capacity, S);
  OCV_CHECK_ARG(group >= 1 && nmax >= 0 && nmax <= capacity, "ocv_object_front_pad_fwd: group must be >= 1 and 0 <= nmax <= capacity");
  OCV_CHECK_ARG(ocv_aligned16(objects) && ocv_aligned16(out), "ocv_object_front_pad_fwd: objects / out must be 16-byte aligned");
  const long n = (long)B * S * (E / 4);
  hipLaunchKernelGGL(object_front_pad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, objects, counts,
                     group, nmax, pad_value, out, key_padding_mask, B, capacity, S, E / 4);
  OCV_CHECK_LAUNCH("ocv_object_front_pad_fwd");
  return 0;
}
