// Validation pre-processing on the device: resize to the longest side + letter-box pad + /255 + HWC -> CHW, one launch
// per batch, reading u8 source images from the pool in HBM.
//
// Reference (per sample, DataLoader workers): SampleReader.__call__(letter_box=True)
//   kod/data/sample_reader.py:16-40   A.LongestMaxSize(S, cv2.INTER_LINEAR) + A.PadIfNeeded(S, S, BORDER_CONSTANT, 114)
//   kod/data/sample_reader.py:102-136 the call
//   kod/data/augmentations/albu.py:91-119  ValidationSampleAugmentor: A.ToFloat(255) + ToTensorV2 (HWC -> CHW)
// cv2.resize(INTER_LINEAR) on 8-bit images is OpenCV's fixed-point path (resize.cpp): 11-bit horizontal and vertical
// weights, intermediate rows in int32, dst = ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2.  The geometry (new
// size, pads, the two double-precision scales) is computed on the host exactly as the libraries do
// (data/device_pipeline.py) and passed per sample.
#include "kodhip_common.h"

namespace {

struct ValDesc {
  long off;                 // byte offset of the source image in the pool (HWC u8)
  int h, w;                 // source size
  int nh, nw;               // size after LongestMaxSize
  int top, left;            // PadIfNeeded offsets
  double scale_x, scale_y;  // source / destination (1 / inv_scale as OpenCV computes it)
};

__device__ __forceinline__ int sat16(float v) {
  int r = __float2int_rn(v);
  return r < -32768 ? -32768 : (r > 32767 ? 32767 : r);
}

// left source index + 11-bit weights of destination index d (resize.cpp, INTER_LINEAR)
__device__ __forceinline__ void lin_coeff(int d, double scale, int n_src, bool clamp_edges, int& s, int& w0, int& w1) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  s = (int)floorf(f);
  f -= (float)s;
  if (clamp_edges) {        // horizontal: x < 0 and x >= width-1 collapse to one tap; vertical clamps the two rows instead
    if (s < 0) { s = 0; f = 0.f; }
    if (s >= n_src - 1) { s = n_src - 1; f = 0.f; }
  }
  w0 = sat16((1.f - f) * 2048.f);
  w1 = sat16(f * 2048.f);
}

// grid: (ceil(S*S/256), B)
__global__ __launch_bounds__(256) void val_prep_kernel(const unsigned char* pool, const ValDesc* descs, float* out_f32,
                                                       bf16_t* out_pairs, int S) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= S * S) return;
  const int y = p / S, x = p - y * S;
  const ValDesc d = descs[b];
  int px[3] = {114, 114, 114};
  const int dy = y - d.top, dx = x - d.left;
  if (dy >= 0 && dy < d.nh && dx >= 0 && dx < d.nw) {
    const unsigned char* src = pool + d.off;
    if (d.nh == d.h && d.nw == d.w) {
#pragma unroll
      for (int c = 0; c < 3; ++c) px[c] = src[((long)dy * d.w + dx) * 3 + c];
    } else {
      int sx, a0, a1, sy, b0, b1;
      lin_coeff(dx, d.scale_x, d.w, true, sx, a0, a1);
      lin_coeff(dy, d.scale_y, d.h, false, sy, b0, b1);
      const int sx1 = min(sx + 1, d.w - 1);
      const int r0 = min(max(sy, 0), d.h - 1), r1 = min(max(sy + 1, 0), d.h - 1);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int S0 = src[((long)r0 * d.w + sx) * 3 + c] * a0 + src[((long)r0 * d.w + sx1) * 3 + c] * a1;
        const int S1 = src[((long)r1 * d.w + sx) * 3 + c] * a0 + src[((long)r1 * d.w + sx1) * 3 + c] * a1;
        const int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
        px[c] = min(max(v, 0), 255);
      }
    }
  }
  float v[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) v[c] = (float)px[c] / 255.f;
  if (out_f32) {
#pragma unroll
    for (int c = 0; c < 3; ++c) out_f32[((size_t)(b * 3 + c) * S + y) * S + x] = v[c];
  }
  if (out_pairs) {   // network input layout [B][S][S/2][8] = pixel pairs x 4 channels
    bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)0.f};
    *reinterpret_cast<bf16x4*>(out_pairs + ((size_t)(b * S + y) * S + x) * 4) = o;
  }
}

}  // namespace

extern "C" {

int kodhip_val_prep_desc_bytes(void) { return (int)sizeof(ValDesc); }

// pool: u8 source images (HWC); descs: device [B] ValDesc; out_f32 [B,3,S,S] and/or out_pairs (bf16 [B,S,S/2,8]).
int kodhip_val_prep_batch(const void* pool, const void* descs, float* out_f32, void* out_pairs, int B, int S,
                          hipStream_t stream) {
  KOD_CHECK_ARG(pool && descs && (out_f32 || out_pairs) && B > 0 && S > 0 && S % 2 == 0, "val_prep_batch: bad args");
  hipLaunchKernelGGL(val_prep_kernel, dim3(cdiv((long)S * S, 256), B), dim3(256), 0, stream, (const unsigned char*)pool,
                     (const ValDesc*)descs, out_f32, (bf16_t*)out_pairs, S);
  KOD_LAUNCH_CHECK("val_prep_batch");
  return KOD_OK;
}

}  // extern "C"
