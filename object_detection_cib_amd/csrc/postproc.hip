// Evaluation post-process on the device: prediction decode + YOLOv5-style batched NMS.
//
//   decode : kod/lightning/experiments/yv5_baseline/layers.py:55-63,74-76,87-89,143-153 (get_detections,
//            exp.py:70-102): xy = (sigmoid*2 + grid - .5)*stride, wh = (sigmoid*2)^2*anchor_px, cxcywh->xyxy,
//            sigmoid(obj), sigmoid(cls), rows ordered (level, anchor, y, x)
//   nms    : kod/core/nms.py:9-75 + torchvision.ops.nms: obj > conf, cls *= obj, multi-label candidates in
//            nonzero (row, class) order, top-30000 by score, boxes offset by class*4096 (in fp32, as the
//            reference does - the offset rounding is part of the result), greedy IoU > thr, first 300.
//
// Candidate compaction by ballot prefix sums (one block per image), a chip-wide bitonic sort of (score desc,
// candidate index asc) 64-bit keys, then a wave-batched greedy pass that keeps the <=300 survivors in LDS.  Everything is integer / comparison logic on fp32 values => results are bit-identical to the reference.
#include "kodhip_common.h"

namespace {

struct DecodeLevel { const float* raw; int h, w, stride, row0; float aw[3], ah[3]; };
struct DecodeArgs { DecodeLevel lv[3]; float* det; int B, A, nc, rows; };

__global__ void decode_kernel(DecodeArgs a) {
  const int P = 5 + a.nc;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (b, row)
  if (i >= (long)a.B * a.rows) return;
  int b = (int)(i / a.rows), row = (int)(i - (long)b * a.rows);
  int l = row >= a.lv[2].row0 ? 2 : (row >= a.lv[1].row0 ? 1 : 0);
  const DecodeLevel& L = a.lv[l];
  int r = row - L.row0;
  int hw = L.h * L.w;
  int an = r / hw, cell = r - an * hw;
  int gy = cell / L.w, gx = cell - gy * L.w;
  const float* p = L.raw + ((size_t)(b * a.A + an) * hw + cell) * P;
  float* o = a.det + (size_t)i * P;
  float sx = 1.f / (1.f + expf(-p[0])), sy = 1.f / (1.f + expf(-p[1]));
  float sw = 1.f / (1.f + expf(-p[2])), sh = 1.f / (1.f + expf(-p[3]));
  float cx = (sx * 2.f + (float)gx - 0.5f) * (float)L.stride;
  float cy = (sy * 2.f + (float)gy - 0.5f) * (float)L.stride;
  float w = (sw * 2.f) * (sw * 2.f) * L.aw[an], h = (sh * 2.f) * (sh * 2.f) * L.ah[an];
  o[0] = cx - 0.5f * w; o[1] = cy - 0.5f * h; o[2] = cx + 0.5f * w; o[3] = cy + 0.5f * h;
  for (int k = 4; k < P; ++k) o[k] = 1.f / (1.f + expf(-p[k]));
}

// ------------------------------------------------------------------------------------------------ NMS
struct NmsArgs {
  const float* det;      // [B][rows][5+nc]
  unsigned long long* keys;   // [B][kcap]  (power of two capacity)
  int* ncand;            // [B]
  float* out;            // [B][max_det][6]
  int* nout;             // [B]
  int B, rows, nc, kcap, max_det, max_nms;
  float conf, iou_thr, max_wh;
};

__device__ __forceinline__ unsigned int sortable_desc(float s) {
  // positive floats: bit pattern is monotonic; invert for descending order under an ascending sort
  return 0xFFFFFFFFu - __float_as_uint(s);
}

__global__ __launch_bounds__(1024) void nms_candidates_kernel(NmsArgs a) {
  __shared__ int wtot[16];
  __shared__ int base_s;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int P = 5 + a.nc;
  const float* D = a.det + (size_t)b * a.rows * P;
  unsigned long long* K = a.keys + (size_t)b * a.kcap;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int r0 = 0; r0 < a.rows; r0 += 1024) {
    int row = r0 + tid;
    int cnt = 0;
    float obj = 0.f;
    if (row < a.rows) {
      obj = D[(size_t)row * P + 4];
      if (obj > a.conf) {
        if (a.nc > 1) {
          for (int c = 0; c < a.nc; ++c) cnt += (D[(size_t)row * P + 5 + c] * obj > a.conf) ? 1 : 0;
        } else {
          cnt = (D[(size_t)row * P + 5] * obj > a.conf) ? 1 : 0;      // best-class path, one class
        }
      }
    }
    // exclusive scan of cnt over the block (wave scan + wave totals)
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int v = __shfl_up(incl, o, 64);
      if (lane >= o) incl += v;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; ++w) before += wtot[w];
    int total = 0;
    for (int w = 0; w < 16; ++w) total += wtot[w];
    int pos = base_s + before + incl - cnt;
    if (cnt) {
      for (int c = 0; c < a.nc; ++c) {
        float s = D[(size_t)row * P + 5 + c] * obj;
        if (s > a.conf) {
          if (pos < a.kcap)
            K[pos] = ((unsigned long long)sortable_desc(s) << 32) | (unsigned int)(row * a.nc + c);
          ++pos;
        }
      }
    }
    __syncthreads();
    if (tid == 0) base_s += total;
    __syncthreads();
  }
  if (tid == 0) a.ncand[b] = base_s < a.kcap ? base_s : a.kcap;
}

// ---- sort of the candidate keys (ascending = best score first, ties by candidate index) ------------------------
// Bitonic network over npad_b = next power of two >= ncand[b] keys per image, spread over the whole chip instead of
// one block per image: the passes with partner distance < SORT_CHUNK run inside LDS (one block per 4096-key chunk),
// the few passes with a longer distance are one launch each.  An image whose candidate count is small leaves every
// block beyond its padded length (and every pass beyond its size) immediately, so a trained network (hundreds of
// candidates) costs one LDS sort per image, while the random-init worst case (252 000 candidates) uses all CUs.
constexpr int SORT_CHUNK = 4096;          // keys per block (32 KB of LDS), 512 threads x 8 keys

__device__ __forceinline__ int npad_of(int n) {
  int npad = 64;
  while (npad < n) npad <<= 1;
  return npad;
}

// pad [n, npad) with +inf keys, then sort every SORT_CHUNK-sized chunk completely (k = 2 .. SORT_CHUNK), directions
// taken from the GLOBAL element index so that the chunks form the bitonic sequences the later merge passes expect
__global__ __launch_bounds__(512) void nms_sort_local_kernel(NmsArgs a) {
  __shared__ unsigned long long sk[SORT_CHUNK];
  const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  const int n = a.ncand[b];
  const int npad = npad_of(n);
  const int base = chunk * SORT_CHUNK;
  if (base >= npad) return;
  unsigned long long* K = a.keys + (size_t)b * a.kcap;
  const int len = npad - base < SORT_CHUNK ? npad - base : SORT_CHUNK;      // power of two
  for (int i = tid; i < len; i += 512) sk[i] = (base + i < n) ? K[base + i] : ~0ull;
  __syncthreads();
  for (int k = 2; k <= len; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < (len >> 1); t += 512) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));      // lower index of pair t
        const int p = i | j;
        unsigned long long x = sk[i], y = sk[p];
        const bool up = ((base + i) & k) == 0;
        if ((x > y) == up) { sk[i] = y; sk[p] = x; }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < len; i += 512) K[base + i] = sk[i];
}

// one pass (k, j) with j >= SORT_CHUNK: partner pairs straight in global memory (L2-resident), one pair per thread
__global__ __launch_bounds__(256) void nms_sort_global_kernel(NmsArgs a, int k, int j) {
  const int b = blockIdx.y;
  const int npad = npad_of(a.ncand[b]);
  if (k > npad) return;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= (npad >> 1)) return;
  unsigned long long* K = a.keys + (size_t)b * a.kcap;
  const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
  const int p = i | j;
  unsigned long long x = K[i], y = K[p];
  const bool up = (i & k) == 0;
  if ((x > y) == up) { K[i] = y; K[p] = x; }
}

// the passes j = SORT_CHUNK/2 .. 1 of merge size k (k > SORT_CHUNK) inside LDS
__global__ __launch_bounds__(512) void nms_sort_merge_kernel(NmsArgs a, int k) {
  __shared__ unsigned long long sk[SORT_CHUNK];
  const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  const int npad = npad_of(a.ncand[b]);
  const int base = chunk * SORT_CHUNK;
  if (k > npad || base >= npad) return;
  unsigned long long* K = a.keys + (size_t)b * a.kcap;
  for (int i = tid; i < SORT_CHUNK; i += 512) sk[i] = K[base + i];
  __syncthreads();
  const bool up = (base & k) == 0;                 // k > SORT_CHUNK: one direction per chunk
  for (int j = SORT_CHUNK >> 1; j > 0; j >>= 1) {
    for (int t = tid; t < (SORT_CHUNK >> 1); t += 512) {
      const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
      const int p = i | j;
      unsigned long long x = sk[i], y = sk[p];
      if ((x > y) == up) { sk[i] = y; sk[p] = x; }
    }
    __syncthreads();
  }
  for (int i = tid; i < SORT_CHUNK; i += 512) K[base + i] = sk[i];
}

__device__ __forceinline__ bool iou_gt(const float* A, float aa, const float* Bx, float ab, float thr) {
  float xx1 = fmaxf(A[0], Bx[0]), yy1 = fmaxf(A[1], Bx[1]);
  float xx2 = fminf(A[2], Bx[2]), yy2 = fminf(A[3], Bx[3]);
  float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
  float inter = w * h;
  return inter / (aa + ab - inter) > thr;
}

// one wave per image
__global__ __launch_bounds__(64) void nms_greedy_kernel(NmsArgs a) {
  __shared__ float kb[304][4];
  __shared__ float ka[304];
  const int b = blockIdx.x, lane = threadIdx.x;
  const int P = 5 + a.nc;
  const float* D = a.det + (size_t)b * a.rows * P;
  const unsigned long long* K = a.keys + (size_t)b * a.kcap;
  int n = a.ncand[b];
  if (n > a.max_nms) n = a.max_nms;
  float* O = a.out + (size_t)b * a.max_det * 6;
  int nk = 0;
  for (int base = 0; base < n && nk < a.max_det; base += 64) {
    int i = base + lane;
    bool alive = i < n;
    float box[4] = {0, 0, 0, 0}, raw[4] = {0, 0, 0, 0}, area = 0.f, score = 0.f;
    int cls = 0;
    if (alive) {
      unsigned long long key = K[i];
      unsigned int id = (unsigned int)key;
      int row = id / a.nc;
      cls = id - row * a.nc;
      const float* r = D + (size_t)row * P;
      score = r[5 + cls] * r[4];
      float off = (float)cls * a.max_wh;
#pragma unroll
      for (int k = 0; k < 4; ++k) { raw[k] = r[k]; box[k] = r[k] + off; }
      area = (box[2] - box[0]) * (box[3] - box[1]);
      for (int k = 0; k < nk && alive; ++k)
        if (iou_gt(kb[k], ka[k], box, area, a.iou_thr)) alive = false;
    }
    unsigned long long mask = __ballot(alive);
    while (mask && nk < a.max_det) {
      int j = __ffsll((long long)mask) - 1;          // earliest surviving candidate of the batch: kept
      float jb[4], jarea;
#pragma unroll
      for (int k = 0; k < 4; ++k) jb[k] = __shfl(box[k], j, 64);
      jarea = __shfl(area, j, 64);
      if (lane == j) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { kb[nk][k] = box[k]; O[nk * 6 + k] = raw[k]; }
        ka[nk] = area;
        O[nk * 6 + 4] = score;
        O[nk * 6 + 5] = (float)cls;
        alive = false;
      }
      if (alive && lane > j && iou_gt(jb, jarea, box, area, a.iou_thr)) alive = false;
      ++nk;
      mask = __ballot(alive);
      __syncthreads();
    }
    __syncthreads();
  }
  if (lane == 0) a.nout[b] = nk;
}

}  // namespace

extern "C" {

struct KodDecodeLevel { const float* raw; int h, w, stride; float anchor_w[3], anchor_h[3]; };

// det [B][sum A*h*w][5+nc] fp32 from the three head tensors [B][A][h][w][5+nc] (anchors in pixels)
int kodhip_decode(const KodDecodeLevel* levels /* host[3] */, float* det, int B, int A, int nc, hipStream_t stream) {
  KOD_CHECK_ARG(levels && det && B > 0 && A == 3 && nc > 0, "decode: bad args (3 anchors per cell)");
  DecodeArgs a = {};
  int row0 = 0;
  for (int l = 0; l < 3; ++l) {
    KOD_CHECK_ARG(levels[l].raw, "decode: null level");
    a.lv[l].raw = levels[l].raw; a.lv[l].h = levels[l].h; a.lv[l].w = levels[l].w; a.lv[l].stride = levels[l].stride;
    a.lv[l].row0 = row0;
    for (int k = 0; k < 3; ++k) { a.lv[l].aw[k] = levels[l].anchor_w[k]; a.lv[l].ah[k] = levels[l].anchor_h[k]; }
    row0 += A * levels[l].h * levels[l].w;
  }
  a.det = det; a.B = B; a.A = A; a.nc = nc; a.rows = row0;
  hipLaunchKernelGGL(decode_kernel, dim3(cdiv((long)B * row0, 256)), dim3(256), 0, stream, a);
  KOD_LAUNCH_CHECK("decode");
  return KOD_OK;
}

// keys: B * key_cap u64 workspace (key_cap = power of two >= rows*nc, or smaller to cap candidates);
// out [B][max_det][6] (x1,y1,x2,y2,score,cls), nout [B], ncand [B].
int kodhip_nms(const float* det, void* keys, int key_cap, int* ncand, float* out, int* nout,
               int B, int rows, int nc, float conf_thres, float nms_thres, int max_det, int max_nms, float max_wh,
               hipStream_t stream) {
  KOD_CHECK_ARG(det && keys && ncand && out && nout && B > 0 && rows > 0 && nc > 0, "nms: bad args");
  KOD_CHECK_ARG(key_cap >= 64 && (key_cap & (key_cap - 1)) == 0, "nms: key_cap must be a power of two >= 64");
  KOD_CHECK_ARG(max_det > 0 && max_det <= 300 && max_nms > 0, "nms: max_det must be in 1..300");
  KOD_CHECK_ARG((long)rows * nc < (1l << 31), "nms: too many candidates");
  NmsArgs a = {};
  a.det = det; a.keys = (unsigned long long*)keys; a.ncand = ncand; a.out = out; a.nout = nout;
  a.B = B; a.rows = rows; a.nc = nc; a.kcap = key_cap; a.max_det = max_det; a.max_nms = max_nms;
  a.conf = conf_thres; a.iou_thr = nms_thres; a.max_wh = max_wh;
  hipLaunchKernelGGL(nms_candidates_kernel, dim3(B), dim3(1024), 0, stream, a);
  KOD_LAUNCH_CHECK("nms_candidates");
  const int chunks = key_cap > SORT_CHUNK ? key_cap / SORT_CHUNK : 1;
  hipLaunchKernelGGL(nms_sort_local_kernel, dim3(chunks, B), dim3(512), 0, stream, a);
  for (int k = 2 * SORT_CHUNK; k <= key_cap; k <<= 1) {
    for (int j = k >> 1; j >= SORT_CHUNK; j >>= 1)
      hipLaunchKernelGGL(nms_sort_global_kernel, dim3(cdiv(key_cap / 2, 256), B), dim3(256), 0, stream, a, k, j);
    hipLaunchKernelGGL(nms_sort_merge_kernel, dim3(chunks, B), dim3(512), 0, stream, a, k);
  }
  KOD_LAUNCH_CHECK("nms_sort");
  hipLaunchKernelGGL(nms_greedy_kernel, dim3(B), dim3(64), 0, stream, a);
  KOD_LAUNCH_CHECK("nms_greedy");
  return KOD_OK;
}

}  // extern "C"
