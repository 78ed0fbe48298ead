/* libkodhip - C ABI of the MI355X-native YOLOv5 training hot path for `kod`
 * (craston/object_detection_cib).
 *
 * The reference has no FFI: its seam is Hydra `_target_` instantiation of Python classes
 * (kod/configs/nn/networks/yv5.yaml:1, kod/configs/nn/losses/yv5.yaml:4, kod/configs/assigners/yv5.yaml:6)
 * whose arithmetic is reached through aten / torchvision ops.  Each entry point below replaces the aten
 * op(s) a reference call site asks for; INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - plain C types only; every pointer is a DEVICE pointer owned by the caller unless marked "host";
 *   - the library never allocates, frees or retains device memory and never synchronises: work is
 *     enqueued on `stream` (pass torch.cuda.current_stream().cuda_stream);
 *   - return 0 = OK, <0 = argument error (nothing launched), >0 = hipError_t; kodhip_last_error() gives
 *     the message (thread local);
 *   - activations are channels-last bf16 [B][H][W][ld]; a tensor may be a channel slice (ld, coff) of a
 *     wider concat buffer - that is how torch.cat (csp.py:109, sppf.py:76, yolov5_pafpn.py:186,199)
 *     disappears; all channel counts / offsets are multiples of 8 (16-byte accesses);
 *   - packed MFMA weight operands (kodhip_pack_weights) have a tap-major K axis with every tap padded to a
 *     multiple of 32 channels: w_packed [N][Kp], k = tap * round_up(Cin, 32) + ci, Kp = KH*KW*round_up(Cin, 32);
 *     w_dgrad [Cin][Kdp], k = tap * round_up(N, 32) + n.  (kodhip_conv_wgrad's Kp is the K extent of its fp32
 *     gradient slabs, k = tap * Cin + ci, Kp = round_up(KH*KW*Cin, 32).)
 */
#ifndef KODHIP_H
#define KODHIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* kodStream_t;

const char* kodhip_last_error(void);
int kodhip_version(void);
int kodhip_device_count(void);

/* ---- layout / packing ------------------------------------------------------------------------- */
/* Lightning batch_to_device + first conv input: NCHW fp32 image -> [B][H][W][4] bf16 (C<=4, zero pad). */
int kodhip_nchw_to_nhwc4(const float* x, void* y, int B, int C, int H, int W, kodStream_t stream);
/* fp32 master weights (state_dict layout [Cout][Cin][KH][KW]) -> bf16 MFMA packs; descs = device array of
 * int64[13] rows {w_off,f_off,d_off,N,Cin,KH,KW,Kp,Kdp,Ntot,n_off,stem,blk_begin}. */
int kodhip_pack_weights(const float* master, void* fpack, void* dpack, const void* descs, int nlayers,
                        long total_blocks, kodStream_t stream);
int kodhip_pack_desc_bytes(void);

/* ---- convolution (aten::convolution / convolution_backward; call sites kod/nn/layers/csp.py:30-46,
 *      kod/nn/layers/sppf.py:29-36,61-67, kod/nn/backbones/yolov5.py:44-52,102-110,
 *      kod/nn/necks/yolov5_pafpn.py:61-73,110-122,152-166) ------------------------------------------ */
int kodhip_conv_stats_slots(long M, int N);
int kodhip_conv_fwd_raw(const void* x, const void* w_packed, void* y, float* stats,
                        int B, int H, int W, int ldx, int xcoff, int Cin,
                        int N, int KH, int KW, int SH, int SW, int PH, int PW, int Kp,
                        int ldy, int ycoff, kodStream_t stream);
/* three biased 1x1 head convs of one level fused (kod/nn/heads/yolov5.py:12-136), out [B][A][H*W][5+nc] fp32 */
int kodhip_conv_fwd_head(const void* x, const void* w_packed, const float* bias, float* out,
                         int B, int H, int W, int ldx, int xcoff, int Cin, int A, int nc, int Kp,
                         kodStream_t stream);
/* Data gradients (aten::convolution_backward dX).  `accumulate`: bit 0 = add to the bf16 partial already in dx (read-
 * modify-write: rounds the partial and the sum); bits 8.. = fp32 accumulation of an activation gradient with several
 * producers (autograd sums those in fp32 before anything is rounded): 1<<8 first producer - also store the fp32 values
 * to dx_f32, the fp32 shadow of dx (same indexing, row stride ldx floats); 2<<8 later producer - add the shadow's sum
 * in the MFMA accumulators before the single bf16 rounding, store the sum back; 3<<8 last producer - the same without
 * the store; 4<<8 - the partial is dx's own bf16 content left by exact producers (a residual pass-through), added in
 * fp32 before the rounding.  dx_f32 may be NULL for modes 0 and 4. */
int kodhip_conv_dgrad(const void* dy, const void* w_dgrad, void* dx,
                      int B, int H, int W, int ldx, int xcoff, int Cin,
                      int N, int KH, int KW, int SH, int SW, int PH, int PW, int Kp,
                      int ldy, int ycoff, int accumulate, void* dx_f32, kodStream_t stream);
/* dX of a 3x3/s2/p1 conv by output-pixel parity classes (9 instead of 36 taps of MFMA work) */
int kodhip_conv_dgrad_s2(const void* dy, const void* w_dgrad_s2, void* dx,
                         int B, int H, int W, int ldx, int xcoff, int Cin, int N,
                         int ldy, int ycoff, int accumulate, void* dx_f32, kodStream_t stream);
/* Data gradient that also produces the BatchNorm-backward reduction of the conv units whose output gradient it
 * completes (it must be the LAST writer of those channel ranges of dx): fuses aten::convolution_backward (dX) with
 * the reduction half of native_batch_norm_backward + silu_backward of the producing Conv2dNormActivation
 * (kod/nn/layers/csp.py:30-46).  Per segment: partials[2][ch_count][slots] = per-block sums of dz and dz*y. */
typedef struct KodBnRedSeg {
  int ch_begin, ch_count;       /* channel range of dx (relative to xcoff) owned by one producing unit */
  const void* raw; int ldr;     /* that unit's pre-BN output y, bf16 [pixels of dx][ldr] */
  const float* aff;             /* scale[ch_count] | shift[ch_count] */
  float* partials;
} KodBnRedSeg;
int kodhip_conv_dgrad_bnred_slots(int B, int H, int W, int Cin, int N, int KH, int KW, int SH, int SW, int PH, int PW,
                                  int ldy, int stride2 /* 1: the kodhip_conv_dgrad_s2 form */);  /* 0: cannot be fused */
int kodhip_conv_dgrad_bnred(const void* dy, const void* w_dgrad, void* dx,
                            int B, int H, int W, int ldx, int xcoff, int Cin,
                            int N, int KH, int KW, int SH, int SW, int PH, int PW, int Kp,
                            int ldy, int ycoff, int accumulate, void* dx_f32, const void* segments /* host KodBnRedSeg[nseg] */,
                            int nseg, int slots, kodStream_t stream);
int kodhip_conv_dgrad_s2_bnred(const void* dy, const void* w_dgrad_s2, void* dx,
                               int B, int H, int W, int ldx, int xcoff, int Cin, int N,
                               int ldy, int ycoff, int accumulate, void* dx_f32, const void* segments, int nseg, int slots,
                               kodStream_t stream);
/* The same data gradient "folded": one stride-1 gather over the 2x2 dY neighbourhood of each 2x2 output-pixel block,
 * 4 parity classes x Cin output columns, depth-to-space epilogue (16 tap-class products instead of 9, but dY is staged
 * once and both x parities of a pixel pair are stored together: faster for the shallow, staging-bound layers).
 * w_fold: [4*Cin][4*round_up(N,32)] (pack mode 3).  kodhip_conv_dgrad_s2_folded: which form to pack / call (1 = folded). */
int kodhip_conv_dgrad_s2_folded(int Cin, int N);
int kodhip_conv_dgrad_s2f(const void* dy, const void* w_fold, void* dx,
                          int B, int H, int W, int ldx, int xcoff, int Cin, int N,
                          int ldy, int ycoff, int accumulate, void* dx_f32, kodStream_t stream);
int kodhip_conv_dgrad_s2f_bnred_slots(int B, int H, int W, int Cin, int N, int ldy);
int kodhip_conv_dgrad_s2f_bnred(const void* dy, const void* w_fold, void* dx,
                                int B, int H, int W, int ldx, int xcoff, int Cin, int N,
                                int ldy, int ycoff, int accumulate, void* dx_f32, const void* segments, int nseg, int slots,
                                kodStream_t stream);
/* dX of TWO pointwise (1x1/s1/p0) convs that read the same tensor - a CSP layer's main_conv and short_conv
 * (kod/nn/layers/csp.py:96-111) - as one launch over the concatenated reduction: dx is written once instead of written
 * and then accumulated into by a second launch.  dy1 / dy2: [B*H*W][ldy] (+ycoff, N channels each), w1 / w2: their
 * dgrad packs [Cin][Kp], Kp = round_up(N, 32). */
int kodhip_conv_dgrad_dual(const void* dy1, const void* w1, const void* dy2, const void* w2, void* dx,
                           int B, int H, int W, int ldx, int xcoff, int Cin, int N, int Kp, int ldy, int ycoff,
                           int accumulate, void* dx_f32, kodStream_t stream);
int kodhip_conv_dgrad_dual_bnred_slots(int B, int H, int W, int Cin, int N, int ldy);
int kodhip_conv_dgrad_dual_bnred(const void* dy1, const void* w1, const void* dy2, const void* w2, void* dx,
                                 int B, int H, int W, int ldx, int xcoff, int Cin, int N, int Kp, int ldy, int ycoff,
                                 int accumulate, void* dx_f32, const void* segments, int nseg, int slots, kodStream_t stream);
int kodhip_conv_wgrad_splits(long M, int N, int Kp);     /* generic split-K kernel */
/* split count of the kernel kodhip_conv_wgrad[_partial] picks for this geometry (3x3 / stride 1 / pad 1 layers with whole
 * 32-channel chunks take a form that stages dY once per block and every input row once per kernel row): size the slab
 * region with this one.  H, W: input dims; ldx / ldy: row strides of x / dy in elements. */
int kodhip_conv_wgrad_splits_geo(int B, int H, int W, int ldx, int Cin, int N, int KH, int KW, int SH, int SW, int PH, int PW,
                                 int Kp, int ldy);
int kodhip_conv_wgrad(const void* x, const void* dy, float* partials, float* grad,
                      int B, int H, int W, int ldx, int xcoff, int Cin,
                      int N, int KH, int KW, int SH, int SW, int PH, int PW, int Kp,
                      int ldy, int ycoff, int n_valid, int stem, float scale, kodStream_t stream);
/* The two halves of kodhip_conv_wgrad apart: the split-K kernel alone (fp32 slabs partials[splits][N][Kp], a region of
 * its own per layer), and ONE launch that sums the slabs of many layers into their gradients (same fixed-order
 * arithmetic as the per-layer form: bit-identical) - a training step reduces a whole gradient bucket at a time instead
 * of launching ~60 small reductions. */
int kodhip_conv_wgrad_partial(const void* x, const void* dy, float* partials,
                              int B, int H, int W, int ldx, int xcoff, int Cin,
                              int N, int KH, int KW, int SH, int SW, int PH, int PW, int Kp,
                              int ldy, int ycoff, kodStream_t stream);
typedef struct KodWgradReduceDesc {
  long part_off, grad_off;      /* floats, relative to the `partials` / `grads` arguments */
  int splits, Nfull, N, K, Kp, Cin /* stem: 8 */, KK /* KH*KW */, stem;
  float scale;
  int block_start;              /* first block of the layer; it owns kodhip_wgrad_reduce_blocks(N, K) blocks */
} KodWgradReduceDesc;
int kodhip_wgrad_reduce_desc_bytes(void);
int kodhip_wgrad_reduce_blocks(int n_valid, int K);
int kodhip_wgrad_reduce_batched(const float* partials, float* grads, const void* descs /* device KodWgradReduceDesc[n] */,
                                int n_desc, int total_blocks, kodStream_t stream);

/* Weight gradients of TWO pointwise layers with the same input (a CSP layer's main_conv and short_conv,
 * kod/nn/layers/csp.py:85-99) in one launch + one reduction: the shared input is streamed from HBM once.  dy1 / dy2:
 * [B*H*W][ldy] (+ycoff, N channels each); partials: kodhip_conv_wgrad_dual_splits(...) * 2N * Kp floats; grad1 / grad2: fp32
 * [N][Cin].  kodhip_conv_wgrad_dual_splits returns 0 where the form does not apply (then: two kodhip_conv_wgrad launches). */
int kodhip_conv_wgrad_dual_splits(int B, int H, int W, int ldx, int Cin, int N, int Kp, int ldy);
int kodhip_conv_wgrad_dual(const void* x, const void* dy1, const void* dy2, float* partials, float* grad1, float* grad2,
                           int B, int H, int W, int ldx, int xcoff, int Cin, int N, int Kp, int ldy, int ycoff,
                           float scale, kodStream_t stream);
/* The stem's whole backward in one kernel + the slab reduction: dY = k1 * dA * silu'(y * scale + shift) + k2 * y + k3 is
 * formed on the fly and multiplied into dW; dY is never written (the stem - kod/nn/backbones/yolov5.py:44-52, 6x6 / stride 2
 * / pad 2 on the image - has no data gradient, so the weight gradient is dY's only reader).  Replaces
 * kodhip_bn_silu_bwd_apply + kodhip_conv_wgrad(stem = 1) for that unit.  x: pixel pairs [B][H][Wp][8] bf16 (Wp = width / 2);
 * dA: [B * H/2 * Wp][lda] (+dacoff), y: [..][ldy] pre-BatchNorm output (left untouched); coef = k1[N] | k2[N] | k3[N] from
 * kodhip_bn_bwd_coeffs*; partials: kodhip_stem_bwd_fused_blocks(B, H, Wp, N) * (N <= 32 ? 32 : 64) * 160 floats; grad: fp32
 * [N][3][6][6].  N <= 64 (two 32-channel tiles per block above 32). */
int kodhip_stem_bwd_fused_blocks(int B, int H, int Wp, int N);
int kodhip_stem_bwd_fused(const void* x, const void* dA, int lda, int dacoff, const void* y, int ldy,
                          const float* scale, const float* shift, const float* coef, float* partials, float* grad,
                          int B, int H, int Wp, int N, float gscale, kodStream_t stream);

/* ---- BatchNorm2d(eps 1e-3, momentum .03) + SiLU (kod/nn/networks/yolov5.py:24,
 *      kod/nn/layers/activations.py:7; aten::native_batch_norm(+backward), silu(+backward)) ---------- */
int kodhip_bn_reduce_partials(const float* partials, double* sums, int C, int T, kodStream_t stream);
int kodhip_bn_finalize(const double* sums, double count, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, float momentum, float eps,
                       float* scale, float* shift, float* mean, float* rstd, int C, int update_running,
                       kodStream_t stream);
/* single-GPU forms: partial slabs -> constants in one launch (no cross-rank all-reduce in between) */
int kodhip_bn_finalize_partials(const float* partials, int T, double count, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, float momentum, float eps,
                                float* scale, float* shift, float* mean, float* rstd, int C, int update_running,
                                kodStream_t stream);
/* two units whose convolution ran as ONE launch with N = 2 * Ch columns - a CSP layer's main_conv + short_conv
 * (kod/nn/layers/csp.py:87-88: same input): partials [2][2 * Ch][T], unit h owns channels [h * Ch, (h + 1) * Ch); aff* =
 * scale | shift | mean | rstd.  view != NULL: SyncBN over the peer buffers (slot0 / slot1, count = pixels of all ranks) */
int kodhip_bn_finalize_partials_pair(const float* partials, int T, double count, int Ch, float momentum, float eps,
                                     int update_running,
                                     const float* gamma0, const float* beta0, float* running_mean0, float* running_var0, float* aff0,
                                     const float* gamma1, const float* beta1, float* running_mean1, float* running_var1, float* aff1,
                                     const void* view /* host KodPeerView or NULL */, unsigned int slot0, unsigned int slot1,
                                     kodStream_t stream);
/* SyncBN forms of the two single launches: the rank's sums are exchanged through the peer buffers (kodhip_peer_*, below)
 * inside the kernel.  count = pixels of ALL ranks; view = host KodPeerView (copied into the launch); slot = first granule
 * of this exchange (it uses 4 * C).  Parameter gradients (dgamma, dbeta) keep the rank's own sums. */
int kodhip_bn_finalize_partials_peer(const float* partials, int T, double count, const float* gamma, const float* beta,
                                     float* running_mean, float* running_var, float momentum, float eps,
                                     float* scale, float* shift, float* mean, float* rstd, int C, int update_running,
                                     const void* view, unsigned int slot, kodStream_t stream);
int kodhip_bn_bwd_coeffs_partials_peer(const float* partials, int T, double count, const float* gamma, const float* mean,
                                       const float* rstd, float* dgamma, float* dbeta, float* coef, int C,
                                       int raw_moment, const void* view, unsigned int slot, kodStream_t stream);
/* kodhip_bn_bwd_coeffs_partials for two units in one launch (a CSP layer's short_conv + main_conv) */
int kodhip_bn_bwd_coeffs_partials2(const float* partials0, int T0, double count0, const float* gamma0, const float* mean0,
                                   const float* rstd0, float* dgamma0, float* dbeta0, float* coef0, int C0, int raw_moment0,
                                   const float* partials1, int T1, double count1, const float* gamma1, const float* mean1,
                                   const float* rstd1, float* dgamma1, float* dbeta1, float* coef1, int C1, int raw_moment1,
                                   kodStream_t stream);
int kodhip_bn_bwd_coeffs_partials(const float* partials, int T, double count, const float* gamma, const float* mean,
                                  const float* rstd, float* dgamma, float* dbeta, float* coef, int C,
                                  int raw_moment /* 1: partials[1] = sum dz*y (kodhip_conv_dgrad_bnred) */,
                                  kodStream_t stream);
int kodhip_bn_silu_apply(const void* y, int ldy /* row stride of y (>= C: y may be a channel slice) */, const float* scale, const float* shift,
                         const void* residual, int ldr, int rcoff,
                         void* out, int ldo, int ocoff, long M, int C, kodStream_t stream);
/* the apply passes of two units whose pre-BN outputs are the channel halves of one tensor y[m][2 * Ch] (one convolution
 * launch for a CSP layer's main_conv + short_conv, csp.py:87-88), each half to its own destination slice */
int kodhip_bn_silu_apply_pair(const void* y, int ldy, int Ch,
                              const float* scale0, const float* shift0, void* out0, int ldo0, int ocoff0,
                              const float* scale1, const float* shift1, void* out1, int ldo1, int ocoff1,
                              long M, kodStream_t stream);
int kodhip_bn_bwd_slots(long M, int C);
int kodhip_bn_silu_bwd_reduce(const void* dA, int lda, int dacoff, const void* y, int ldy, const float* scale,
                              const float* shift, const float* mean, const float* rstd, float* partials,
                              long M, int C, kodStream_t stream);
int kodhip_bn_bwd_coeffs(const double* sums_local, const double* sums_global, double count, const float* gamma,
                         const float* mean, const float* rstd, float* dgamma, float* dbeta, float* coef, int C,
                         int raw_moment, kodStream_t stream);
int kodhip_bn_silu_bwd_apply(const void* dA, int lda, int dacoff, void* y_inout, int ldy, const float* scale,
                             const float* shift, const float* coef, void* dI, int ldi, int dicoff, int di_accum,
                             long M, int C, kodStream_t stream);

/* The same three elementwise passes for the other activations the reference's layer signatures admit
 * (kod/nn/layers/csp.py:16-46: `activation_layer: Callable[..., nn.Module]`; its configs use SiLUInplace only):
 * act = 0 SiLU (dispatches to the entries above), 1 ReLU, 2 LeakyReLU(slope), 3 Hardswish, 4 identity (activation_layer=None);
 * torch's conventions at the kinks.  A network built with one of them runs the BatchNorm-backward reduction as its own pass. */
int kodhip_bn_act_apply(const void* y, int ldy, const float* scale, const float* shift, const void* residual, int ldr, int rcoff,
                        void* out, int ldo, int ocoff, long M, int C, int act, float slope, kodStream_t stream);
int kodhip_bn_act_bwd_apply(const void* dA, int lda, int dacoff, void* y_inout, int ldy, const float* scale, const float* shift,
                            const float* coef, void* dI, int ldi, int dicoff, int di_accum, long M, int C, int act, float slope,
                            kodStream_t stream);
int kodhip_bn_act_bwd_reduce(const void* dA, int lda, int dacoff, const void* y, int ldy, const float* scale, const float* shift,
                             const float* mean, const float* rstd, float* partials, long M, int C, int act, float slope,
                             kodStream_t stream);

/* ---- SPPF max-pool, nearest upsample (kod/nn/layers/sppf.py:46-50,73-76,
 *      kod/nn/necks/yolov5_pafpn.py:144-146,182-184) ------------------------------------------------- */
/* idx: [B][H][W][C] bytes, the winning tap of every output (an opaque code: written by _fwd, read by _bwd of the same
 * library); argmax as torch's scan: first maximum in row-major window order, a NaN beats every number */
int kodhip_maxpool5_fwd(const void* x, int ldx, int xcoff, void* y, int ldy, int ycoff, void* idx,
                        int B, int H, int W, int C, kodStream_t stream);
/* dx_f32 (NULL = none): fp32 shadow of dx holding the earlier producers' partial sum (see kodhip_conv_dgrad) */
int kodhip_maxpool5_bwd(const void* dy, int ldy, int ycoff, const void* idx, void* dx, int ldx, int xcoff,
                        int B, int H, int W, int C, const float* dx_f32, kodStream_t stream);
/* any odd window K <= 15 (SPPFBottleneck's kernel_sizes, sppf.py:27-67): K = 5 dispatches to the kernels above */
int kodhip_maxpool_fwd(const void* x, int ldx, int xcoff, void* y, int ldy, int ycoff, void* idx,
                       int B, int H, int W, int C, int K, kodStream_t stream);
int kodhip_maxpool_bwd(const void* dy, int ldy, int ycoff, const void* idx, void* dx, int ldx, int xcoff,
                       int B, int H, int W, int C, int K, const float* dx_f32, kodStream_t stream);
int kodhip_upsample2x_fwd(const void* x, int ldx, int xcoff, void* y, int ldy, int ycoff,
                          int B, int H, int W, int C, kodStream_t stream);
int kodhip_upsample2x_bwd(const void* dy, int ldy, int ycoff, void* dx, int ldx, int xcoff, int accumulate,
                          int B, int H, int W, int C, const float* dx_f32, kodStream_t stream);
/* workspace: 2048 * Npad floats (per-block bias partials) */
int kodhip_head_bwd_prep(const float* g, void* dy, float* workspace, float* db_box, float* db_obj, float* db_cls,
                         int B, int HW, int A, int nc, int Npad, kodStream_t stream);

/* ---- optimizer (torch.optim.SGD as grouped by kod/nn/optim/smart.py:36-58) ---------------------- */
int kodhip_sgd_nesterov(float* params, const float* grads, float* momentum_buf, const void* group_ids,
                        long n, const float* hyper /* device, 12 floats: lr[3] momentum[3] wd[3] grad_scale,
                                                      flags = nesterov (smart_sgd.yaml: 1) + 2 maximize + 4 first step
                                                      (only read when dampening != 0), dampening */,
                        kodStream_t stream);
int kodhip_fill_u32(void* p, uint32_t value, long n, kodStream_t stream);
/* dst (device) <- src (PINNED host memory), bytes % 16 == 0, both 16-byte aligned: a kernel pulling the bytes through the
 * host memory's device mapping - the small per-step tables of the data path (kod/data/detection.py's per-sample results)
 * without an async-copy hand-over on the step's stream; hipMemcpyAsync when the memory has no device mapping */
int kodhip_pull_from_host(void* dst, const void* src_pinned, long bytes, kodStream_t stream);
/* debug: *dst = the device's constant-rate wall clock (100 MHz), stream-ordered - time stamps inside a replayed hipGraph */
int kodhip_debug_stamp(unsigned long long* dst, kodStream_t stream);

/* ---- target assignment + loss (kod/core/label_assignment/yv5.py:45-319,
 *      kod/lightning/experiments/yv5_baseline/loss.py:65-248, kod/core/bbox/iou.py:200-246) ---------- */
typedef struct KodAssignLevel {
  int* idx; int* label; float* gt; float* anc; int* count;
  float anchor_w[3], anchor_h[3];
  int stride;
} KodAssignLevel;
int kodhip_assign_targets(const double* boxes, const long* labels, const int* samples, int n, int cap,
                          int img_w, int img_h, float threshold, const KodAssignLevel* levels /* host[3] */,
                          kodStream_t stream);
typedef struct KodLossLevel {
  const float* logits; float* grad;
  const int* idx; const int* label; const float* gt; const float* anc; const int* count;
  int* cellmaps; int* rowprev; float* rowgrad; float* tobj;
  int fh, fw;
  float balance;
} KodLossLevel;
/* out: 16 floats - [0..2] localization / objectness / classification, [3..11] per-level raw means, [12] (with upstream)
 * upstream[0] * ((loc + cls) + obj): the training-step scalar of exp.py:104-121 when the three upstreams are one scale */
int kodhip_yolo_loss(const KodLossLevel* levels /* host[3] */, int B, int A, int nc, int cap,
                     float lam_box, float lam_obj, float lam_cls, const float* pos_weight,
                     const float* upstream, float* partials, int nslots, float* out, int compute_grad,
                     kodStream_t stream);

/* the same with the loss's IoUCalculator (loss.py:46-63,96): iou_kind 0 iou | 1 giou | 2 diou | 3 ciou (iou.py:9-14) and its eps;
 * (3, 1e-7f) is what kodhip_yolo_loss runs */
int kodhip_yolo_loss_iou(const KodLossLevel* levels /* host[3] */, int B, int A, int nc, int cap,
                         float lam_box, float lam_obj, float lam_cls, const float* pos_weight,
                         const float* upstream, float* partials, int nslots, float* out, int compute_grad,
                         int iou_kind, float iou_eps, kodStream_t stream);

/* Aligned IoU family behind kod.core.bbox.iou.IoUCalculator.__call__ (kod/core/bbox/iou.py:77-95,142-268):
 * boxes [m][4] xyxy fp32 -> out [m]; kind 0 iou | 1 giou | 2 diou | 3 ciou (IoUType order, iou.py:9-14).
 * The backward follows autograd's conventions (max/min ties split evenly, clamp(0) passes at 0, CIoU alpha constant). */
int kodhip_iou_fwd(const float* boxes1, const float* boxes2, float* out, long m, int kind, float eps,
                   kodStream_t stream);
int kodhip_iou_bwd(const float* boxes1, const float* boxes2, const float* grad_out, float* grad_boxes1 /* or NULL */,
                   float* grad_boxes2 /* or NULL */, long m, int kind, float eps, kodStream_t stream);

/* ---- device data path: mosaic + warpAffine + HSV + flip + /255 (+ mixup) in one gather kernel
 *      (kod/data/mosaic.py:58-132, kod/data/augmentations/default.py:279-320,354-408,433-438) ------- */
int kodhip_compose_desc_bytes(void);
int kodhip_compose_batch(const void* pool, const void* descs, const float* mix, const void* bilinear_tab,
                         float* out_f32, void* out_pairs, int B, int S, kodStream_t stream);
/* image_color_transforms (kod/data/augmentations/default.py:420-432,460-461; the reference's shipped default,
 * kod/configs/data/augmentations/aug_params.yaml:15): the albumentations stage between warp and HSV for ONE sample whose
 * gate fired - descriptor `entry` (2 * sample + slot of descs [B][2]) is warped into out_u8 [S][S][3] and the fired
 * transforms run on it in Compose order: ops bit 0 Blur(blur_k in 3/5/7), 1 MedianBlur(median_k), 2 ToGray, 3 CLAHE(clahe_clip
 * in [1, 4], 8 x 8 tiles, on the L channel of an 8-bit Lab image).  tmp_u8: S*S*3 bytes, luts_u8: 64*256 bytes, color_tab: the
 * Lab tables (data/device_pipeline.color_table()).  The descriptor's `pre` field must hold out_u8 when descs is uploaded:
 * kodhip_compose_batch (same stream, afterwards) then takes that sample's warped pixels from there. */
int kodhip_compose_color(const void* pool, const void* descs, const void* bilinear_tab, const void* color_tab, int entry,
                         int ops, int blur_k, int median_k, double clahe_clip, void* out_u8, void* tmp_u8, void* luts_u8,
                         int S, kodStream_t stream);

/* ---- validation pre-processing: SampleReader (LongestMaxSize + PadIfNeeded(114), kod/data/sample_reader.py:16-40,
 *      102-136) + ValidationSampleAugmentor (ToFloat(255) + CHW, kod/data/augmentations/albu.py:91-119) ------------ */
int kodhip_val_prep_desc_bytes(void);
int kodhip_val_prep_batch(const void* pool, const void* descs /* device [B] */, float* out_f32, void* out_pairs,
                          int B, int S, kodStream_t stream);

/* ---- evaluation post-process (kod/lightning/experiments/yv5_baseline/layers.py:55-155, exp.py:70-102;
 *      kod/core/nms.py:9-75 + torchvision.ops.nms) -------------------------------------------------- */
typedef struct KodDecodeLevel { const float* raw; int h, w, stride; float anchor_w[3], anchor_h[3]; } KodDecodeLevel;
int kodhip_decode(const KodDecodeLevel* levels /* host[3] */, float* det, int B, int A, int nc, kodStream_t stream);
int kodhip_nms(const float* det, void* keys, int key_cap, int* ncand, float* out, int* nout,
               int B, int rows, int nc, float conf_thres, float nms_thres, int max_det, int max_nms, float max_wh,
               kodStream_t stream);

/* detection <-> ground-truth matching of COCO-style mAP (kod/lightning/callbacks/pycoco_map_eval.py:50-125 ->
 * vision_evaluation -> pycocotools COCOeval.evaluateImg); tp [B][max_det][T] u8, counted [B][max_det] u8.
 * At most 256 ground truths per image take part in the matching (later ones are never matched). */
int kodhip_map_match(const float* det, const int* ndet, const double* gt_boxes, const long* gt_labels,
                     const int* gt_start, void* tp, void* counted, int B, int max_det, int nc,
                     const double* iou_thresholds /* host */, int T, int max_per_class, kodStream_t stream);

/* ---- data-parallel collectives (Lightning strategy=ddp + sync_batchnorm=True, kod/configs/trainer/ddp.yaml:4-9:
 *      torch DistributedDataParallel's bucketed gradient all-reduce, SyncBatchNorm's statistic exchange and the
 *      initial parameter/buffer broadcast).  RCCL enqueued on the caller's stream; hipGraph-capturable. ------ */
int kodhip_comm_load(const char* rccl_path /* NULL: the copy already loaded in the process */);
int kodhip_comm_unique_id(void* id128 /* host, 128 bytes out */);
int kodhip_comm_init(void** comm, const void* id128, int rank, int world);
int kodhip_comm_destroy(void* comm);
int kodhip_comm_allreduce_sum(void* comm, void* buf, long count, int elem_bytes /* 4: fp32, 8: fp64 */, kodStream_t stream);
int kodhip_comm_allreduce_sum_to(void* comm, const void* send, void* recv, long count, int elem_bytes, kodStream_t stream);
int kodhip_comm_broadcast(void* comm, void* buf, long bytes, int root, kodStream_t stream);
/* collectives enqueued between the two calls are launched as one fused operation (ncclGroupStart / ncclGroupEnd) */
int kodhip_comm_group_start(void);
int kodhip_comm_group_end(void);

/* ---- SyncBN statistic exchange over peer buffers (sync_batchnorm: True, kod/configs/trainer/ddp.yaml:9; replaces the
 *      114 small all-reduces torch SyncBatchNorm issues per step).  Every rank of the node owns one small exchange
 *      buffer - the ONE device allocation this library makes (fine-grained memory), freed by kodhip_peer_destroy - and
 *      maps the others' through HIP IPC; the BatchNorm finalize / coefficient kernels publish their two fp64 sums per
 *      channel there and read the other ranks' directly over xGMI: no collective launch, no communicator order. ----- */
#define KODHIP_PEER_MAX 8
typedef struct KodPeerView {                       /* passed BY VALUE to the *_peer kernels */
  unsigned long long* peers[KODHIP_PEER_MAX];
  int world, rank;
  const unsigned int* seq;
  int* timeout_flag;                               /* device: 1 = a poll gave up, 2 = the ranks' step counters diverged */
  int* host_flag;                                  /* the same verdict mirrored into pinned host memory */
  long max_spins;                                  /* polls before a wait gives up and raises the flag */
} KodPeerView;
/* fails (no coarse-grained fallback) when fine-grained device memory cannot be had: the caller keeps the RCCL exchanges */
int kodhip_peer_create(void** peer, int rank, int world, long granules /* 8-byte granules: 4 per channel per exchange site */);
int kodhip_peer_export(void* peer, void* handle64 /* host, 64 bytes out: hipIpcMemHandle_t */);
int kodhip_peer_connect(void* peer, const void* handles /* host, world x 64 bytes in rank order */);
/* the same for ranks living in ONE process (no IPC): peers = the world's kodhip_peer_create handles in rank order */
int kodhip_peer_connect_local(void* peer, void* const* peers);
int kodhip_peer_view_bytes(void);
int kodhip_peer_view(void* peer, void* view_out /* host KodPeerView */);
int kodhip_peer_step_begin(void* peer, kodStream_t stream);      /* once per step, before its first exchange */
int kodhip_peer_allreduce_f64(void* peer, const double* in, double* out, int n, unsigned int slot, kodStream_t stream);
/* all ranks of a one-process group (kodhip_peer_connect_local) in ONE dispatch (the ranks' blocks wait for each other, so they
 * must be co-resident: separate launches on separate streams may share a hardware queue); ins / outs in rank order */
int kodhip_peer_allreduce_f64_multi(void* const* peers, int world, const double* const* ins, double* const* outs, int n,
                                    unsigned int slot, kodStream_t stream);
int kodhip_peer_timed_out(void* peer, int* flag /* host out; synchronises, resets the flag */);
/* the verdict without a device synchronisation (pinned host mirror): 0 ok, 1 a poll gave up, 2 step counters diverged.
 * A failed exchange never folds a stale payload into the statistics: its sums become NaN. */
int kodhip_peer_status(void* peer, int* flag /* host out */);
int kodhip_peer_destroy(void* peer);

#ifdef __cplusplus
}
#endif
#endif /* KODHIP_H */
