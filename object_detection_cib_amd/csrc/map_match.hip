// Detection <-> ground-truth matching for COCO-style mAP, on the device (the Python per-box loops of
// kod/lightning/callbacks/pycoco_map_eval.py:50-106 plus pycocotools' COCOeval.evaluateImg).
//
// One block per image, one lane per (class, IoU threshold) pair: each lane walks the image's detections in
// descending-score order (NMS output order), keeps the first maxDets of its class and greedily matches each to
// the best still-free ground truth of that class with IoU >= threshold (fp64 IoU, no +1), exactly the
// sequential semantics of COCOeval.  Outputs per-detection TP flags per threshold; the precision/recall
// accumulation over the whole validation set stays on the host (a few thousand rows).
#include "kodhip_common.h"

namespace {

struct MapArgs {
  const float* det;      // [B][max_det][6]
  const int* ndet;       // [B]
  const double* gt;      // [n][4] xyxy
  const long* gt_label;  // [n]
  const int* gt_start;   // [B+1] offsets of each image's ground truths (sorted by image)
  unsigned char* tp;     // [B][max_det][T]
  unsigned char* counted;// [B][max_det]  1 when the detection is within its class's maxDets
  int B, max_det, nc, T, max_per_class;
  double thr[8];
};

__global__ void map_match_kernel(MapArgs a) {
  const int b = blockIdx.x;
  const int nd = min(a.ndet[b], a.max_det);
  const int g0 = a.gt_start[b], g1 = a.gt_start[b + 1];
  const float* D = a.det + (size_t)b * a.max_det * 6;
  for (int pair = threadIdx.x; pair < a.nc * a.T; pair += blockDim.x) {
    const int c = pair / a.T, t = pair - c * a.T;
    const double thr = a.thr[t];
    unsigned long long used[4] = {0, 0, 0, 0};                 // up to 256 ground truths per image
    int seen = 0;
    for (int d = 0; d < nd; ++d) {
      if ((int)D[d * 6 + 5] != c) continue;
      const bool in_budget = seen < a.max_per_class;
      ++seen;
      if (t == 0) a.counted[(size_t)b * a.max_det + d] = in_budget ? 1 : 0;
      unsigned char hit = 0;
      if (in_budget) {
        double x1 = D[d * 6], y1 = D[d * 6 + 1], x2 = D[d * 6 + 2], y2 = D[d * 6 + 3];
        double ad = (x2 - x1) * (y2 - y1);
        double best = thr < 1.0 - 1e-10 ? thr : 1.0 - 1e-10;
        int m = -1;
        for (int g = g0; g < g1; ++g) {
          if (a.gt_label[g] != c) continue;
          int k = g - g0;
          if (k >= 256) break;       // beyond the bitmap: never matched (the host wrapper rejects such images)
          if ((used[k >> 6] >> (k & 63)) & 1ull) continue;
          const double* G = a.gt + (size_t)g * 4;
          double w = fmin(x2, G[2]) - fmax(x1, G[0]);
          double h = fmin(y2, G[3]) - fmax(y1, G[1]);
          w = w > 0 ? w : 0; h = h > 0 ? h : 0;
          double inter = w * h;
          double iou = inter / (ad + (G[2] - G[0]) * (G[3] - G[1]) - inter);
          if (iou < best) continue;
          best = iou; m = k;
        }
        if (m >= 0) { used[m >> 6] |= 1ull << (m & 63); hit = 1; }
      }
      a.tp[((size_t)b * a.max_det + d) * a.T + t] = hit;
    }
  }
}

}  // namespace

extern "C" {

int kodhip_map_match(const float* det, const int* ndet, const double* gt_boxes, const long* gt_labels,
                     const int* gt_start, void* tp, void* counted, int B, int max_det, int nc,
                     const double* iou_thresholds /* host */, int T, int max_per_class, hipStream_t stream) {
  KOD_CHECK_ARG(det && ndet && gt_start && tp && counted && iou_thresholds && B > 0 && max_det > 0 && nc > 0,
                "map_match: bad args");
  KOD_CHECK_ARG(T > 0 && T <= 8, "map_match: at most 8 IoU thresholds");
  MapArgs a = {};
  a.det = det; a.ndet = ndet; a.gt = gt_boxes; a.gt_label = gt_labels; a.gt_start = gt_start;
  a.tp = (unsigned char*)tp; a.counted = (unsigned char*)counted;
  a.B = B; a.max_det = max_det; a.nc = nc; a.T = T; a.max_per_class = max_per_class;
  for (int i = 0; i < T; ++i) a.thr[i] = iou_thresholds[i];
  hipLaunchKernelGGL(map_match_kernel, dim3(B), dim3(64), 0, stream, a);
  KOD_LAUNCH_CHECK("map_match");
  return KOD_OK;
}

}  // extern "C"
