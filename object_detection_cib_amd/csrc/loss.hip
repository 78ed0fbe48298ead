// YOLOv5 target assignment + loss, forward and analytic backward, for gfx950.
//
//   assigner : kod/core/label_assignment/yv5.py:45-319  (one block per pyramid level, ballot prefix sums
//              reproduce the reference's row order: [all] [left] [top] [right] [bottom], anchor-major)
//   loss     : kod/lightning/experiments/yv5_baseline/loss.py:65-248 with kod/core/bbox/iou.py:200-246 (CIoU)
//              including the reference quirks: objectness target = clamp(iou, 0) NOT detached, duplicate
//              cells = last row wins, empty level => NaN box/cls loss, per-level means.
//
// Everything is fp32 like the reference (fp64 where the reference is fp64: box -> grid conversion).
// Duplicate-cell gradient accumulation is done in row order by the cell's last writer (no float atomics),
// loss sums go through fixed-order partial slabs => bitwise run-to-run reproducible.
#include "kodhip_common.h"
#include "kodhip_iou.h"
#include <limits.h>

namespace {

struct AssignLevel {
  int* idx;        // [4][cap] sample, anchor, gy, gx
  int* label;      // [cap]
  float* gt;       // [cap][4]
  float* anc;      // [cap][2]
  int* count;      // [1]
  float aw[3], ah[3];
  float fmap_w, fmap_h;
  int fw, fh;
  double inv_stride;
};
struct AssignArgs {
  AssignLevel lv[3];
  const double* boxes;    // [n][4]
  const long* labels;     // [n]
  const int* samples;     // [n]
  int n, cap;
  float threshold;
};

__global__ __launch_bounds__(1024) void assign_kernel(AssignArgs a) {
  const AssignLevel L = a.lv[blockIdx.x];
  __shared__ int wtot[5][16];
  __shared__ int base[5];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int items = 3 * a.n;

  auto flags_of = [&](int it, float* g4) -> int {
    if (it >= items) return 0;
    int an = it / a.n, i = it - an * a.n;
    const double* b = a.boxes + (size_t)i * 4;
    float cx = (float)(((b[0] + b[2]) / 2.0) * L.inv_stride);
    float cy = (float)(((b[1] + b[3]) / 2.0) * L.inv_stride);
    float w = (float)((b[2] - b[0]) * L.inv_stride);
    float h = (float)((b[3] - b[1]) * L.inv_stride);
    g4[0] = cx; g4[1] = cy; g4[2] = w; g4[3] = h;
    float rw = w / L.aw[an], rh = h / L.ah[an];
    float mw = fmaxf(rw, 1.0f / rw), mh = fmaxf(rh, 1.0f / rh);
    if (!(fmaxf(mw, mh) < a.threshold)) return 0;
    int f = 1;
    float gix = L.fmap_w - cx, giy = L.fmap_h - cy;
    if ((cx - floorf(cx) < 0.5f) && cx > 1.f) f |= 2;
    if ((cy - floorf(cy) < 0.5f) && cy > 1.f) f |= 4;
    if ((gix - floorf(gix) < 0.5f) && gix > 1.f) f |= 8;
    if ((giy - floorf(giy) < 0.5f) && giy > 1.f) f |= 16;
    return f;
  };

  // pass 1: totals per offset block
  int tot[5] = {0, 0, 0, 0, 0};
  for (int c0 = 0; c0 < items; c0 += 1024) {
    float g4[4];
    int f = flags_of(c0 + tid, g4);
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      unsigned long long m = __ballot((f >> k) & 1);
      if (lane == 0) wtot[k][wave] = __popcll(m);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 5; ++k)
      for (int w = 0; w < 16; ++w) tot[k] += wtot[k][w];
    __syncthreads();
  }
  if (tid == 0) {
    int s = 0;
    for (int k = 0; k < 5; ++k) { base[k] = s; s += tot[k]; }
    *L.count = s;
  }
  __syncthreads();
  // pass 2: write rows
  const float offx[5] = {0.f, 0.5f, 0.f, -0.5f, 0.f};
  const float offy[5] = {0.f, 0.f, 0.5f, 0.f, -0.5f};
  int run[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) run[k] = base[k];
  for (int c0 = 0; c0 < items; c0 += 1024) {
    float g4[4];
    int it = c0 + tid;
    int f = flags_of(it, g4);
    int pre[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      unsigned long long m = __ballot((f >> k) & 1);
      pre[k] = __popcll(m & ((1ull << lane) - 1ull));
      if (lane == 0) wtot[k][wave] = __popcll(m);
    }
    __syncthreads();
    int ctot[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      int before = 0, all = 0;
      for (int w = 0; w < 16; ++w) { int v = wtot[k][w]; all += v; if (w < wave) before += v; }
      pre[k] += before;
      ctot[k] = all;
    }
    if (f & 1) {
      int an = it / a.n, i = it - an * a.n;
      int smp = a.samples[i];
      int lab = (int)a.labels[i];
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        if ((f >> k) & 1) {
          int pos = run[k] + pre[k];
          if (pos < a.cap) {
            long gx = (long)(g4[0] - offx[k]);
            long gy = (long)(g4[1] - offy[k]);
            int cxi = gx < 0 ? 0 : (gx > L.fw - 1 ? L.fw - 1 : (int)gx);
            int cyi = gy < 0 ? 0 : (gy > L.fh - 1 ? L.fh - 1 : (int)gy);
            L.idx[0 * a.cap + pos] = smp;
            L.idx[1 * a.cap + pos] = an;
            L.idx[2 * a.cap + pos] = cyi;
            L.idx[3 * a.cap + pos] = cxi;
            L.label[pos] = lab;
            L.gt[pos * 4 + 0] = g4[0] - (float)gx;
            L.gt[pos * 4 + 1] = g4[1] - (float)gy;
            L.gt[pos * 4 + 2] = g4[2];
            L.gt[pos * 4 + 3] = g4[3];
            L.anc[pos * 2 + 0] = L.aw[an];
            L.anc[pos * 2 + 1] = L.ah[an];
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) run[k] += ctot[k];
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------ loss
struct LossLevel {
  const float* logits;  // [B][A][HW][P]
  float* grad;          // same shape (written when compute_grad)
  const int* idx;       // [4][cap]
  const int* label;
  const float* gt;
  const float* anc;
  const int* count;
  int* last;            // [B*A*HW] init -1: highest row index on the cell (the surviving writer)
  int* first;           // init -1: head of the per-cell row chain
  int* cnt;             // unused scratch
  int* prev;            // [cap] next row of the same cell (chain), -1 terminates
  float* G;             // [cap][4+nc] per-row logit gradients
  float* tobj;          // [cap] clamp(iou,0)
  int fh, fw;
  float balance;
};
struct LossArgs {
  LossLevel lv[3];
  int B, A, nc, P, cap;
  float lam_box, lam_obj, lam_cls;     // already include (W/640)^2 and nc/80
  const float* pos_weight;             // [nc] or null
  const float* upstream;               // 3 floats (d total / d box, obj, cls) or null (=1)
  float* partials;                     // [3 levels][3 kinds][nblk]
  int nblk;                            // slots per (level, kind)
  float* out;                          // [3] losses + [9] per-level raw means
  int compute_grad;
  uint32_t magic_p;                    // magic of P (loss_cells: element index -> cell within a 64-cell chunk)
  int iou_kind;                        // 0 iou | 1 giou | 2 diou | 3 ciou (IoUCalculator.iou_type, kod/core/bbox/iou.py:9-14)
  float iou_eps;
  int iou_generic;                     // anything but the fused ciou / 1e-7 form: the dual-number row of kodhip_iou.h
};

__device__ __forceinline__ int cell_of(const LossLevel& L, int A, int cap, int r) {
  int s = L.idx[r], an = L.idx[cap + r], gy = L.idx[2 * cap + r], gx = L.idx[3 * cap + r];
  return ((s * A + an) * L.fh + gy) * L.fw + gx;
}

__global__ void loss_mark_kernel(LossArgs a) {
  const LossLevel& L = a.lv[blockIdx.y];
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= *L.count || r >= a.cap) return;
  int c = cell_of(L, a.A, a.cap, r);
  atomicMax(&L.last[c], r);
  // per-cell chain of rows (order of insertion is arbitrary; consumers walk it in row-index order)
  L.prev[r] = atomicExch(&L.first[c], r);
}

__device__ __forceinline__ float tie_lt(float a, float b) { return a < b ? 1.f : (a == b ? 0.5f : 0.f); }

__device__ float block_sum_256(float v, float* sm) {
  v = wave_sum(v);
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sm[wave] = v;
  __syncthreads();
  float s = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += sm[w];
  return s;
}

__global__ __launch_bounds__(256) void loss_rows_kernel(LossArgs a) {
  __shared__ float sm[4];
  const int lvl = blockIdx.y;
  const LossLevel& L = a.lv[lvl];
  const int m = min(*L.count, a.cap);
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  float lbox = 0.f, lcls = 0.f;
  if (r < m) {
    const int cell = cell_of(L, a.A, a.cap, r);
    const float* p = L.logits + (size_t)cell * a.P;
    const float eps = 1e-7f;
    float s0 = 1.f / (1.f + expf(-p[0])), s1 = 1.f / (1.f + expf(-p[1]));
    float s2 = 1.f / (1.f + expf(-p[2])), s3 = 1.f / (1.f + expf(-p[3]));
    float aw = L.anc[r * 2], ah = L.anc[r * 2 + 1];
    float px = s0 * 2.f - 0.5f, py = s1 * 2.f - 0.5f;
    float pw = (s2 * 2.f) * (s2 * 2.f) * aw, ph = (s3 * 2.f) * (s3 * 2.f) * ah;
    float x1 = px - 0.5f * pw, y1 = py - 0.5f * ph, x2 = px + 0.5f * pw, y2 = py + 0.5f * ph;
    float gcx = L.gt[r * 4], gcy = L.gt[r * 4 + 1], gw = L.gt[r * 4 + 2], gh = L.gt[r * 4 + 3];
    float x1g = gcx - 0.5f * gw, y1g = gcy - 0.5f * gh, x2g = gcx + 0.5f * gw, y2g = gcy + 0.5f * gh;
    float iw = fminf(x2, x2g) - fmaxf(x1, x1g), ih = fminf(y2, y2g) - fmaxf(y1, y1g);
    float iwc = fmaxf(iw, 0.f), ihc = fmaxf(ih, 0.f);
    float inter = iwc * ihc;
    float w1 = x2 - x1, h1 = y2 - y1, w2 = x2g - x1g, h2 = y2g - y1g;
    float uni = w1 * h1 + w2 * h2 - inter;
    float ue = uni + eps;
    float iou = inter / ue;
    float cw = fmaxf(x2, x2g) - fminf(x1, x1g), chh = fmaxf(y2, y2g) - fminf(y1, y1g);
    float diag = cw * cw + chh * chh;
    float ddx = (x1 + x2) / 2.f - (x1g + x2g) / 2.f, ddy = (y1 + y2) / 2.f - (y1g + y2g) / 2.f;
    float cdist = ddx * ddx + ddy * ddy;
    float de = diag + eps;
    float D = cdist / de;
    const float c4 = 4.f / (3.14159265358979323846f * 3.14159265358979323846f);
    float q1 = w1 / (h1 + eps);
    float dat = atanf(w2 / (h2 + eps)) - atanf(q1);
    float v = c4 * dat * dat;
    float alpha = v / ((1.f - iou) + v + eps);
    float ciou = iou - D - alpha * v;
    // any other IoUCalculator (iou / giou / diou, or another eps; loss.py:46-63 takes whichever it is given): the value and
    // its four derivatives w.r.t. the predicted box from the dual-number row that also backs the IoUCalculator op
    float dgen[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.iou_generic) {
      const float b1[4] = {x1, y1, x2, y2}, b2[4] = {x1g, y1g, x2g, y2g};
      if (a.compute_grad) {
        const Dual<8> rr = iou_row<8>(b1, b2, a.iou_kind, a.iou_eps);
        ciou = rr.v;
#pragma unroll
        for (int k = 0; k < 4; ++k) dgen[k] = rr.d[k];
      } else {
        ciou = iou_row<0>(b1, b2, a.iou_kind, a.iou_eps).v;
      }
    }
    lbox = 1.f - ciou;
    L.tobj[r] = fmaxf(ciou, 0.f);

    // classification BCE over nc logits (loss.py:128-164)
    const int lab = L.label[r];
    const float* pc = p + 5;
    float* G = L.G + (size_t)r * (4 + a.nc);
    const float inv_m = 1.f / (float)m;
    const float u_box = a.upstream ? a.upstream[0] : 1.f;
    const float u_obj = a.upstream ? a.upstream[1] : 1.f;
    const float u_cls = a.upstream ? a.upstream[2] : 1.f;
    const float kcls = a.lam_cls * u_cls * inv_m / (float)a.nc;
    for (int c = 0; c < a.nc; ++c) {
      float x = pc[c];
      float t = (c == lab) ? 1.f : 0.f;
      float lw = a.pos_weight ? (a.pos_weight[c] - 1.f) * t + 1.f : 1.f;
      float sp = log1pf(expf(-fabsf(x))) + fmaxf(-x, 0.f);      // = -log_sigmoid(x)
      lcls += (1.f - t) * x + lw * sp;
      if (a.compute_grad) {
        float sg = 1.f / (1.f + expf(-x));
        G[4 + c] = kcls * ((1.f - t) - lw * (1.f - sg));
      }
    }

    if (a.compute_grad) {
      // d(ciou)/d(x1,y1,x2,y2)
      float miw = iw >= 0.f ? 1.f : 0.f, mih = ih >= 0.f ? 1.f : 0.f;
      float diw_x2 = tie_lt(x2, x2g), diw_x1 = -tie_lt(x1g, x1);
      float dih_y2 = tie_lt(y2, y2g), dih_y1 = -tie_lt(y1g, y1);
      float dI[4] = {ihc * miw * diw_x1, iwc * mih * dih_y1, ihc * miw * diw_x2, iwc * mih * dih_y2};
      float dA1[4] = {-h1, -w1, h1, w1};
      float dcw_x2 = tie_lt(x2g, x2), dcw_x1 = -tie_lt(x1, x1g);
      float dch_y2 = tie_lt(y2g, y2), dch_y1 = -tie_lt(y1, y1g);
      float dDiag[4] = {2.f * cw * dcw_x1, 2.f * chh * dch_y1, 2.f * cw * dcw_x2, 2.f * chh * dch_y2};
      float dCd[4] = {ddx, ddy, ddx, ddy};
      float hq = h1 + eps;
      float dv_w1 = -2.f * c4 * dat / (1.f + q1 * q1) / hq;
      float dv_h1 = 2.f * c4 * dat / (1.f + q1 * q1) * (w1 / (hq * hq));
      float dV[4] = {-dv_w1, -dv_h1, dv_w1, dv_h1};
      float dC[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float dU = dA1[k] - dI[k];
        float diou = (dI[k] * ue - inter * dU) / (ue * ue);
        float dD = (dCd[k] * de - cdist * dDiag[k]) / (de * de);
        dC[k] = a.iou_generic ? dgen[k] : diou - dD - alpha * dV[k];
      }
      float d_px = dC[0] + dC[2], d_py = dC[1] + dC[3];
      float d_pw = 0.5f * (dC[2] - dC[0]), d_ph = 0.5f * (dC[3] - dC[1]);
      // upstream on ciou: box mean + objectness-target path.  torch's index_put_ backward hands the cell's
      // gradient to EVERY row that wrote the cell (overwritten duplicates included), not only the survivor.
      float up = -a.lam_box * u_box * inv_m;
      if (ciou >= 0.f) {
        float ncells = (float)a.B * a.A * L.fh * L.fw;
        float xo = p[4];
        up += -xo * a.lam_obj * u_obj * L.balance / ncells;
      }
      G[0] = up * d_px * 2.f * s0 * (1.f - s0);
      G[1] = up * d_py * 2.f * s1 * (1.f - s1);
      G[2] = up * d_pw * 8.f * aw * s2 * s2 * (1.f - s2);
      G[3] = up * d_ph * 8.f * ah * s3 * s3 * (1.f - s3);
    }
  }
  float sb = block_sum_256(lbox, sm);
  float sc = block_sum_256(lcls, sm);
  if (threadIdx.x == 0) {
    a.partials[((size_t)(lvl * 3 + 0)) * a.nblk + blockIdx.x] = sb;
    a.partials[((size_t)(lvl * 3 + 2)) * a.nblk + blockIdx.x] = sc;
  }
}

// dense objectness pass: one lane per CELL computes the BCE term and its gradient (one exp / log1p per cell, no
// divergence), then the wave writes the 64 cells' P gradient slots as ONE contiguous run - lane-linear 4-byte stores,
// slot 4 fetched from the owning lane by a shuffle, zeros elsewhere (round 1 ran one lane per element: every wave paid
// the transcendental branch for its 4 active lanes, 15 x the arithmetic).
__global__ __launch_bounds__(256) void loss_cells_kernel(LossArgs a) {
  __shared__ float sm[4];
  const int lvl = blockIdx.y;
  const LossLevel& L = a.lv[lvl];
  const long ncells = (long)a.B * a.A * L.fh * L.fw;
  const float u_obj = a.upstream ? a.upstream[1] : 1.f;
  const float kobj = a.lam_obj * u_obj * L.balance / (float)ncells;
  const int lane = threadIdx.x & 63;
  const long nchunks = (ncells + 63) / 64;
  const long wave0 = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * (blockDim.x >> 6);
  float acc = 0.f;
  for (long c = wave0; c < nchunks; c += nwaves) {
    const long cell = c * 64 + lane;
    float g = 0.f;
    if (cell < ncells) {
      const float x = L.logits[cell * a.P + 4];
      const int lr = L.last[cell];
      const float t = lr >= 0 ? L.tobj[lr] : 0.f;
      acc += (1.f - t) * x + log1pf(expf(-fabsf(x))) + fmaxf(-x, 0.f);
      g = kobj * (1.f / (1.f + expf(-x)) - t);
    }
    if (a.compute_grad) {
      const long rest = ncells - c * 64;
      const int nelem = (int)(rest < 64 ? rest : 64) * a.P;
      float* out = L.grad + c * 64 * a.P;
      for (int k = 0; k < a.P; ++k) {
        const int idx = k * 64 + lane;
        const int cl = (int)__umulhi((uint32_t)idx, a.magic_p);       // idx / P (idx < 2^13: exact)
        const int slot = idx - cl * a.P;
        const float v = __shfl(g, cl & 63, 64);
        if (idx < nelem) out[idx] = slot == 4 ? v : 0.f;
      }
    }
  }
  float s = block_sum_256(acc, sm);
  if (threadIdx.x == 0) a.partials[((size_t)(lvl * 3 + 1)) * a.nblk + blockIdx.x] = s;
}

// last writer of each matched cell sums the row gradients of all rows on that cell in ascending row order
// (walks the cell's chain repeatedly, picking the next larger row index: chains are 1-3 rows long)
__global__ void loss_scatter_kernel(LossArgs a) {
  const LossLevel& L = a.lv[blockIdx.y];
  const int m = min(*L.count, a.cap);
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= m) return;
  const int cell = cell_of(L, a.A, a.cap, r);
  if (L.last[cell] != r) return;
  const int W = 4 + a.nc;
  float* g = L.grad + (size_t)cell * a.P;
  int rows[8];
  int cnt = 0, done = -1;
  for (;;) {
    int nxt = INT_MAX;
    for (int q = L.first[cell]; q >= 0; q = L.prev[q])
      if (q > done && q < nxt) nxt = q;
    if (nxt == INT_MAX) break;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i == cnt) rows[i] = nxt;
    ++cnt;
    done = nxt;
  }
  if (cnt <= 8) {
    for (int k = 0; k < W; ++k) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (i < cnt) s += L.G[(size_t)rows[i] * W + k];
      g[k < 4 ? k : k + 1] = s;
    }
  } else {                                   // very crowded cell: re-walk per component
    for (int k = 0; k < W; ++k) {
      float s = 0.f;
      int d2 = -1;
      for (;;) {
        int nxt = INT_MAX;
        for (int q = L.first[cell]; q >= 0; q = L.prev[q])
          if (q > d2 && q < nxt) nxt = q;
        if (nxt == INT_MAX) break;
        s += L.G[(size_t)nxt * W + k];
        d2 = nxt;
      }
      g[k < 4 ? k : k + 1] = s;
    }
  }
}

__global__ void loss_finalize_kernel(LossArgs a, int nblk_rows, int nblk_cells) {
  const int lane = threadIdx.x;          // one wave
  double box = 0, obj = 0, cls = 0;
  for (int l = 0; l < 3; ++l) {
    const LossLevel& L = a.lv[l];
    double sb = 0, so = 0, sc = 0;
    for (int i = lane; i < nblk_rows; i += 64) {
      sb += (double)a.partials[(size_t)(l * 3 + 0) * a.nblk + i];
      sc += (double)a.partials[(size_t)(l * 3 + 2) * a.nblk + i];
    }
    for (int i = lane; i < nblk_cells; i += 64) so += (double)a.partials[(size_t)(l * 3 + 1) * a.nblk + i];
    sb = wave_sum_d(sb); sc = wave_sum_d(sc); so = wave_sum_d(so);
    int m = min(*L.count, a.cap);
    double ncells = (double)a.B * a.A * L.fh * L.fw;
    float lb = (float)(sb / (double)m);                 // 0/0 = NaN on an empty level, as the reference
    float lc = (float)(sc / ((double)m * a.nc));
    float lo = L.balance * (float)(so / ncells);
    if (lane == 0) { a.out[3 + l * 3 + 0] = lb; a.out[3 + l * 3 + 1] = lo; a.out[3 + l * 3 + 2] = lc; }
    box += lb; obj += lo; cls += lc;
  }
  if (lane == 0) {
    const float o0 = a.lam_box * (float)box, o1 = a.lam_obj * (float)obj, o2 = a.lam_cls * (float)cls;
    a.out[0] = o0;
    a.out[1] = o1;
    a.out[2] = o2;
    // the training-step scalar scale * ((localization + classification) + objectness) (exp.py:104-121) in the order and
    // precision torch evaluates it, for callers whose three upstream gradients are the same scale
    if (a.upstream) a.out[12] = a.upstream[0] * ((o0 + o2) + o1);
  }
}

}  // namespace

extern "C" {

// Host-visible flat descriptors (all pointers are device pointers).
struct KodAssignLevel {
  int* idx; int* label; float* gt; float* anc; int* count;
  float anchor_w[3], anchor_h[3];     // anchors / stride (fp32, as the reference builds them)
  int stride;
};

int kodhip_assign_targets(const double* boxes, const long* labels, const int* samples, int n, int cap,
                          int img_w, int img_h, float threshold, const KodAssignLevel* levels /*[3], host*/,
                          hipStream_t stream) {
  KOD_CHECK_ARG(levels && cap > 0 && n >= 0, "assign_targets: bad args");
  KOD_CHECK_ARG(n == 0 || (boxes && labels && samples), "assign_targets: null inputs");
  AssignArgs a = {};
  for (int l = 0; l < 3; ++l) {
    const KodAssignLevel& s = levels[l];
    KOD_CHECK_ARG(s.idx && s.label && s.gt && s.anc && s.count && s.stride > 0, "assign_targets: bad level %d", l);
    AssignLevel& d = a.lv[l];
    d.idx = s.idx; d.label = s.label; d.gt = s.gt; d.anc = s.anc; d.count = s.count;
    for (int k = 0; k < 3; ++k) { d.aw[k] = s.anchor_w[k]; d.ah[k] = s.anchor_h[k]; }
    d.fmap_w = (float)((double)img_w / s.stride); d.fmap_h = (float)((double)img_h / s.stride);
    d.fw = img_w / s.stride; d.fh = img_h / s.stride;
    d.inv_stride = 1.0 / (double)s.stride;
  }
  a.boxes = boxes; a.labels = labels; a.samples = samples; a.n = n; a.cap = cap; a.threshold = threshold;
  hipLaunchKernelGGL(assign_kernel, dim3(3), dim3(1024), 0, stream, a);
  KOD_LAUNCH_CHECK("assign_targets");
  return KOD_OK;
}

struct KodLossLevel {
  const float* logits; float* grad;
  const int* idx; const int* label; const float* gt; const float* anc; const int* count;
  int* cellmaps;       // 3 * ncells ints: last | chain head | scratch
  int* rowprev;        // cap ints: per-row chain links
  float* rowgrad;      // cap * (4+nc)
  float* tobj;         // cap
  int fh, fw;
  float balance;
};

// partials: 9 * nslots floats with nslots >= max(ceil(cap/256), 1024) (need not be initialised); out: 16 floats
// ([0..2] losses, [3..11] per-level raw means, [12] upstream[0] * ((loc + cls) + obj) when upstream is given).
int kodhip_yolo_loss_iou(const KodLossLevel* levels, int B, int A, int nc, int cap, float lam_box, float lam_obj, float lam_cls,
                         const float* pos_weight, const float* upstream, float* partials, int nslots, float* out,
                         int compute_grad, int iou_kind, float iou_eps, hipStream_t stream);

int kodhip_yolo_loss(const KodLossLevel* levels /*[3], host*/, int B, int A, int nc, int cap,
                     float lam_box, float lam_obj, float lam_cls, const float* pos_weight,
                     const float* upstream, float* partials, int nslots, float* out, int compute_grad,
                     hipStream_t stream) {
  return kodhip_yolo_loss_iou(levels, B, A, nc, cap, lam_box, lam_obj, lam_cls, pos_weight, upstream, partials, nslots, out,
                              compute_grad, 3, 1e-7f, stream);
}

// The same with the IoUCalculator the loss was constructed with (kod/lightning/experiments/yv5_baseline/loss.py:46-63, 96:
// `iou = self.iou_calculator(pred_xyxy, gt_xyxy)`): iou_kind 0 iou | 1 giou | 2 diou | 3 ciou, its eps.  (3, 1e-7) - the
// reference's configuration, kod/configs/nn/losses/yv5.yaml:13-16 - runs the closed-form CIoU and its analytic gradient.
int kodhip_yolo_loss_iou(const KodLossLevel* levels /*[3], host*/, int B, int A, int nc, int cap,
                         float lam_box, float lam_obj, float lam_cls, const float* pos_weight,
                         const float* upstream, float* partials, int nslots, float* out, int compute_grad,
                         int iou_kind, float iou_eps, hipStream_t stream) {
  KOD_CHECK_ARG(levels && partials && out && B > 0 && A > 0 && nc > 0 && cap > 0, "yolo_loss: bad args");
  KOD_CHECK_ARG(iou_kind >= 0 && iou_kind <= 3 && iou_eps >= 0.f, "yolo_loss: bad IoU kind %d / eps", iou_kind);
  LossArgs a = {};
  a.iou_kind = iou_kind; a.iou_eps = iou_eps; a.iou_generic = !(iou_kind == 3 && iou_eps == 1e-7f);
  long max_cells = 0;
  for (int l = 0; l < 3; ++l) {
    const KodLossLevel& s = levels[l];
    KOD_CHECK_ARG(s.logits && s.idx && s.label && s.gt && s.anc && s.count && s.cellmaps && s.rowprev && s.rowgrad && s.tobj,
                  "yolo_loss: null pointer in level %d", l);
    KOD_CHECK_ARG(!compute_grad || s.grad, "yolo_loss: grad buffer missing");
    long ncells = (long)B * A * s.fh * s.fw;
    KOD_CHECK_ARG(ncells * (5 + nc) < (1l << 31), "yolo_loss: level too large");
    LossLevel& d = a.lv[l];
    d.logits = s.logits; d.grad = s.grad; d.idx = s.idx; d.label = s.label; d.gt = s.gt; d.anc = s.anc;
    d.count = s.count; d.last = s.cellmaps; d.first = s.cellmaps + ncells; d.cnt = s.cellmaps + 2 * ncells;
    d.G = s.rowgrad; d.tobj = s.tobj; d.prev = s.rowprev; d.fh = s.fh; d.fw = s.fw; d.balance = s.balance;
    if (ncells > max_cells) max_cells = ncells;
  }
  int nblk_rows = cdiv(cap, 256);
  int nblk_cells = 1024;
  KOD_CHECK_ARG(nslots >= nblk_rows && nslots >= nblk_cells, "yolo_loss: partials too small (%d)", nslots);
  a.B = B; a.A = A; a.nc = nc; a.P = 5 + nc; a.cap = cap;
  a.lam_box = lam_box; a.lam_obj = lam_obj; a.lam_cls = lam_cls;
  a.pos_weight = pos_weight; a.upstream = upstream; a.partials = partials; a.nblk = nslots; a.out = out;
  a.compute_grad = compute_grad;
  KOD_CHECK_ARG(5 + nc <= 128, "yolo_loss: at most 123 classes");
  a.magic_p = magic_u32((uint32_t)(5 + nc));
  // cell maps: last = -1, first (chain head) = -1 - adjacent halves of cellmaps, one fill per level
  for (int l = 0; l < 3; ++l) {
    long ncells = (long)B * A * a.lv[l].fh * a.lv[l].fw;
    hipError_t e1 = hipMemsetAsync(a.lv[l].last, 0xFF, 2 * ncells * sizeof(int), stream);
    if (e1 != hipSuccess) { kodhip_set_error("yolo_loss: memset failed"); return 1; }
  }
  hipLaunchKernelGGL(loss_mark_kernel, dim3(nblk_rows, 3), dim3(256), 0, stream, a);
  KOD_LAUNCH_CHECK("loss_mark");
  hipLaunchKernelGGL(loss_rows_kernel, dim3(nblk_rows, 3), dim3(256), 0, stream, a);
  KOD_LAUNCH_CHECK("loss_rows");
  hipLaunchKernelGGL(loss_cells_kernel, dim3(nblk_cells, 3), dim3(256), 0, stream, a);
  KOD_LAUNCH_CHECK("loss_cells");
  if (compute_grad) {
    hipLaunchKernelGGL(loss_scatter_kernel, dim3(nblk_rows, 3), dim3(256), 0, stream, a);
    KOD_LAUNCH_CHECK("loss_scatter");
  }
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, stream, a, nblk_rows, nblk_cells);
  KOD_LAUNCH_CHECK("loss_finalize");
  return KOD_OK;
}

}  // extern "C"
