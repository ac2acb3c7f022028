/*
 * ecm_hip.h -- C ABI of libecm_hip.so: the MI355X (gfx950) hot path of the
 * Explicit-Context-Mapping stereo network (reference: cmf/models/cmfsm.py).
 *
 * Conventions (all entry points):
 *   - plain pointers and ints only; every pointer is a DEVICE pointer that the
 *     caller owns (borrowed; kernels never allocate, free or synchronise);
 *   - tensors are contiguous fp32, NCHW / NCDHW exactly as the reference holds them;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); work is
 *     enqueued asynchronously on it;
 *   - return 0 on success, a negative ECM_E* code on a bad argument, or the positive
 *     hipError_t of a failed launch; no exceptions cross the ABI;
 *   - re-entrant; the only process-wide state is the sticky asynchronous-error word (ecm_async_status) and the
 *     GroupNorm cluster mode (ecm_gn3d_cluster_mode), both documented below.
 *
 * Each function cites the reference interface (file:line under the reference repo)
 * it replaces; INTEGRATION.md shows the ctypes binding a maintainer would add.
 */
#ifndef ECM_HIP_H
#define ECM_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define ECM_EINVAL   (-1)   /* bad shape / null pointer */
#define ECM_EUNSUP   (-2)   /* shape outside what the kernels are built for */
#define ECM_ESCRATCH (-3)   /* caller-provided scratch too small */
#define ECM_EASYNC   (-4)   /* an EARLIER asynchronous launch failed on the device (sticky; see ecm_async_status) */

/* Library / build info. */
int         ecm_abi_version(void);            /* bumps on any signature change */
const char* ecm_error_string(int code);       /* for both ECM_E* and hipError_t codes */

/* ---- a1: cost volume (cmfsm.py:667-682; == cat_d matchshifted, cmfsm.py:88-108) ------------
 * cost[b, c,   d, y, x] = L[b,c,y,x]     if x >= d else 0
 * cost[b, C+c, d, y, x] = R[b,c,y,x-d]   if x >= d else 0
 * L,R: [B,C,h,w]; cost: [B,2C,D,h,w].  fwd writes every element (no pre-zeroing needed). */
int ecm_costvol_concat_fwd(const float* L, const float* R, float* cost,
                           int B, int C, int h, int w, int D, void* stream);
/* gL[b,c,y,x] = sum_{d<=x} gcost[b,c,d,y,x];  gR[b,c,y,x'] = sum_{d, x'+d<w} gcost[b,C+c,d,y,x'+d] */
int ecm_costvol_concat_bwd(const float* gcost, float* gL, float* gR,
                           int B, int C, int h, int w, int D, void* stream);

/* ---- a8: soft-argmin over D (cmfsm.py:703-706 + disparityregression 111-123) -----------------
 * Fused three-head form: logits_k = sum_{j<=k} c_j (cmfsm.py:725,748), disp[k] = sum_d softmax_d(logits_k)*d.
 * c: nheads pointers' worth of [B,D,h*w] raw classifier outputs packed as c[head] = c0 + head*head_stride.
 * disp: [nheads,B,h*w].  nheads in 1..3. */
int ecm_softargmin_heads_fwd(const float* c0, long long head_stride, float* disp,
                             int nheads, int B, int D, int hw, void* stream);
/* gc[head] (same packing as c) from gdisp [nheads,B,hw]; recomputes the softmax. */
int ecm_softargmin_heads_bwd(const float* c0, long long head_stride, const float* gdisp, float* gc0,
                             int nheads, int B, int D, int hw, void* stream);
/* disparityregression.forward alone (cmfsm.py:120-123): out[b,p] = sum_d x[b,d,p]*d */
int ecm_disparity_regression_fwd(const float* x, float* out, int B, int D, int hw, void* stream);

/* ---- a9: NN-upsample x scale + 9-neighbour weighted sum (cmfsm.py:709-723, 730-744, 755-769) --
 * out[k,b,Y,X] = sum_n w9[b,n,Y,X] * s * d[k,b,Y/s+dy_n,X/s+dx_n]   (0 outside), n order: c,l,r,t,b,lt,rt,lb,rb
 * d: [nheads,B,h,w]; w9: [B,9,H,W] (H=h*s, W=w*s); out: [nheads,B,H,W]. */
int ecm_aggregate9_fwd(const float* d, const float* w9, float* out,
                       int nheads, int B, int h, int w, int s, void* stream);
int ecm_aggregate9_bwd(const float* d, const float* w9, const float* gout, float* gd, float* gw9,
                       int nheads, int B, int h, int w, int s, void* stream);

/* ---- a3: eight-related context-mapping weights (cmfsm.py:431-593, 304-358, 391-428) ----------
 * lr: [B,32,h,w]; hr: [B,32,H,W]; W0 [32,66], W1 [16,32], W2 [8,16], W3 [1,8] (1x1 conv weights, no bias);
 * w9: [B,9,H,W] softmax planes in the reference's return order.
 * scratch: >= ecm_weights9_scratch_bytes(B,h,w) bytes of device memory. */
long long ecm_weights9_scratch_bytes(int B, int h, int w);
int ecm_weights9_fwd(const float* lr, const float* hr, const float* W0, const float* W1, const float* W2,
                     const float* W3, float* w9, void* scratch, long long scratch_bytes,
                     int B, int h, int w, int s, void* stream);
/* Gradients of all inputs from gw9 [B,9,H,W].  gW = [gW0(2112) | gW1(512) | gW2(128) | gW3(8)] floats. */
long long ecm_weights9_bwd_scratch_bytes(int B, int h, int w, int s);
int ecm_weights9_bwd(const float* lr, const float* hr, const float* W0, const float* W1, const float* W2,
                     const float* W3, const float* w9, const float* gw9,
                     float* glr, float* ghr, float* gW, void* scratch, long long scratch_bytes,
                     int B, int h, int w, int s, void* stream);

/* ---- a4: six-related context mapping (cmfsm_sub_8.py:440-572; same text in cmfsm_sub_16.py, cm_sub_4/8/16.py) and the
 * general form of a3.  variant 0: eight_related (9 planes, softmax; == ecm_weights9_fwd); variant 1: six_related on the
 * reference image (5 planes c,r,l,t,b; zero padding; extra LeakyReLU; output softmax*logit); variant 2: six_related on
 * the target image (3 planes c,r,l).  out: [B,N,H,W], N = 9/5/3.  Any even scale s. */
int ecm_context_weights_fwd(const float* lr, const float* hr, const float* W0, const float* W1, const float* W2,
                            const float* W3, float* out, void* scratch, long long scratch_bytes,
                            int B, int h, int w, int s, int variant, void* stream);

/* Backward of ecm_context_weights_fwd for any variant and any scale with s % 4 == 0.  out_saved = the forward output
 * (used by variant 0; variants 1/2 recompute their logits).  gW = [gW0(2112) | gW1(512) | gW2(128) | gW3(8)] floats. */
long long ecm_context_weights_bwd_scratch_bytes(int B, int h, int w, int s, int variant);
int ecm_context_weights_bwd(const float* lr, const float* hr, const float* W0, const float* W1, const float* W2,
                            const float* W3, const float* out_saved, const float* gout,
                            float* glr, float* ghr, float* gW, void* scratch, long long scratch_bytes,
                            int B, int h, int w, int s, int variant, void* stream);

/* ---- a10: volume mapping head (cmfsm_sub_16.py:767-801 [+804-848], cm_sub_8.py:765-800), fused: NN-upsample of the LR
 * logits in D,H,W, 5-neighbour spatial fuse with m5 [B,5,H,W] (c,r,l,t,b), three target-weight volumes built from
 * mt3 [B,3,H,W] (c,r,l) shifted by the disparity, +-s fuse along D, softmax over D = Dl*s, regression.
 * c: raw classifier outputs [nheads][B,Dl,h,w] (head k uses c_0+...+c_k); disp: [nheads,B,H,W]. */
int ecm_volume_mapping_fwd(const float* c0, long long head_stride, const float* m5, const float* mt3, float* disp,
                           int nheads, int B, int Dl, int h, int w, int s, void* stream);
/* Gradients of c (same packing), m5 and mt3 from gdisp [nheads,B,H,W]; s must be a power of two <= 64.  Deterministic:
 * every sum is taken in a fixed order (per-row partial sums in `scratch`, gathered by a second kernel; no float atomics),
 * every output element is written exactly once. */
long long ecm_volume_mapping_bwd_scratch_bytes(int nheads, int B, int Dl, int h, int w, int s);
int ecm_volume_mapping_bwd(const float* c0, long long head_stride, const float* m5, const float* mt3, const float* gdisp,
                           float* gc0, float* gm5, float* gmt3, void* scratch, long long scratch_bytes,
                           int nheads, int B, int Dl, int h, int w, int s, void* stream);

/* ---- a11: trilinear head (bilinear_cmf.py:447-471), fused: F.interpolate(trilinear, align_corners=False) of the LR
 * logits to [Do,H,W], softmax over Do, regression.  c as above (cumulative over heads); disp: [nheads,B,H,W]. */
int ecm_trilinear_softargmin_fwd(const float* c0, long long head_stride, float* disp,
                                 int nheads, int B, int Dl, int h, int w, int Do, int H, int W, void* stream);
/* Deterministic (per-pixel plane gradients in `scratch`, then separable fixed-order reductions along x and y). */
long long ecm_trilinear_softargmin_bwd_scratch_bytes(int nheads, int B, int Dl, int h, int w, int H, int W);
int ecm_trilinear_softargmin_bwd(const float* c0, long long head_stride, const float* gdisp, float* gc0,
                                 void* scratch, long long scratch_bytes,
                                 int nheads, int B, int Dl, int h, int w, int Do, int H, int W, void* stream);

/* ---- a5-a7: 3-D aggregation (cmfsm.py:49-58 convbn_3d, 240-303 hourglass, 604-634) ----------
 * Weights are taken in the reference (checkpoint) layouts and repacked on device by the *_pack calls. */

/* Repack Conv3d weight [Co,Ci,3,3,3] -> kernel layout [27][Ci][CoP] (CoP = Co rounded up to 32).
 * flip_transpose != 0 packs the data-gradient operator of a stride-1 conv instead (taps reversed, Ci/Co
 * swapped: a conv with Cin'=Co, Cout'=Ci), so dgrad runs on ecm_conv3d_k3_fwd too. */
long long ecm_conv3d_packed_floats(int Ci, int Co);
int ecm_conv3d_pack_weight(const float* w, float* packed, int Co, int Ci, int flip_transpose, void* stream);

/* y[b,co,od,oh,ow] = sum w[co,ci,kd,kh,kw] x[b,ci,od*st+kd-1,oh*st+kh-1,ow*st+kw-1]; k=3, pad=1, st in {1,2}.
 * x: [B,Ci,D,H,W]; y: [B,Co,Do,Ho,Wo] with Do=(D-1)/st+1 etc.  Ci % 4 == 0; 1 <= Co <= 64. */
int ecm_conv3d_k3_fwd(const float* x, const float* wpacked, float* y,
                      int B, int Ci, int Co, int D, int H, int W, int stride, void* stream);

/* The same stride-1 convolution by Winograd F(2x2,3x3) in the (h,w) plane, direct along depth (conv_wino.hip): 2.25x fewer
 * multiplies, fp32 throughout (transforms add/subtract/halve only; results differ from the direct kernel by fp32 rounding).
 * kd = 3: Conv3d 3x3x3, x [B,Ci,D,H,W] -> y [B,Co,D,H,W];  kd = 1: Conv2d 3x3 applied to each of the D planes (D = 1 for
 * a plain image; D = d*d phase planes of a dilation-d layer, see models.feature_extraction).  Any Ci, Co (32 output
 * channels per workgroup; more go to a second grid dimension).  Weights: reference layout [Co,Ci,kd,3,3], transformed and
 * packed by ecm_conv_wino_pack_weight (flip_transpose != 0: the data-gradient operator, Cin' = Co, Cout' = Ci; size query
 * with the swapped counts). */
long long ecm_conv_wino_packed_floats(int Ci, int Co, int kd);
int ecm_conv_wino_pack_weight(const float* w, float* packed, int Co, int Ci, int kd, int flip_transpose, void* stream);
/* Both layouts in one launch: packed[0 .. packed_floats(Ci,Co,kd)) = the forward layout, followed by the data-gradient layout
 * (packed_floats(Co,Ci,kd) floats) -- what a training step needs of every layer (forward now, backward later). */
int ecm_conv_wino_pack_weight2(const float* w, float* packed, int Co, int Ci, int kd, void* stream);
int ecm_conv_wino_fwd(const float* x, const float* upacked, float* y, int B, int Ci, int Co, int D, int H, int W, int kd,
                      void* stream);
/* out = ((a + b) + c) + d elementwise over n floats, c and d optional (NULL): the gradient accumulation of a tensor with
 * several consumers in one pass (cmfsm.py:686-693: cost0 feeds the first hourglass and three residual adds) instead of
 * autograd's chain of binary adds.  16-byte aligned pointers; out may alias an input. */
int ecm_sum_n(const float* a, const float* b, const float* c, const float* d, float* out, long long n, void* stream);
/* out [planes,H,W] = small [planes,Hs,Ws] on the even positions, 0 elsewhere: the data gradient of a 1x1 / stride-2 projection
 * (the encoder's downsample layers) once W^T gy exists on the coarse grid. */
int ecm_zero_insert2d(const float* small, float* out, long long planes, int H, int W, int Hs, int Ws, void* stream);

/* y = conv(x) + addend, addend shaped like y and added in the kernel's epilogue: with the data-gradient weights this is
 * autograd's accumulation at a skip connection (gx = dgrad(gy) + g_skip; BasicBlock cmfsm.py:76-85, dres1 612-613, the
 * hourglass / classifier inputs 636-660) without the separate elementwise pass.  y may not alias addend. */
int ecm_conv_wino_fwd_add(const float* x, const float* upacked, const float* addend, float* y, int B, int Ci, int Co, int D,
                          int H, int W, int kd, void* stream);

/* Weight gradient of the same stride-1 convolutions in Winograd form: gw = G^T [ sum_tiles (A gy A^T) (.) (B^T x B) ] G, the
 * per-lane operand transforms done on the fly from the raw LDS tiles of ecm_conv3d_k3_wgrad's kernel (conv3d_wgrad.hip).
 * kd = 3: x [B,Ci,D,H,W], gy [B,Co,D,H,W] -> gw [Co,Ci,3,3,3];  kd = 1 (D independent planes): gw [Co,Ci,3,3].  Deterministic. */
long long ecm_conv_wino_wgrad_scratch_bytes(int B, int Ci, int Co, int D, int H, int W, int kd);
int ecm_conv_wino_wgrad(const float* x, const float* gy, float* gw, void* scratch, long long scratch_bytes, int B, int Ci,
                        int Co, int D, int H, int W, int kd, void* stream);

/* The classifier's last layer Conv3d(Ci<=32 -> 1) (cmfsm.py:624,629,634) on its own kernels: w is the reference weight
 * [1,Ci,3,3,3] (no packing); y: [B,1,D,H,W].  wgrad: gw [1,Ci,27] from x [B,Ci,D,H,W] and gy [B,1,D,H,W]. */
int ecm_conv3d_c1_fwd(const float* x, const float* w, float* y, int B, int Ci, int D, int H, int W, void* stream);
/* data gradient of the same layer: gx[B,Ci,D,H,W] from gy[B,1,D,H,W] and w[1,Ci,27] (any Ci). */
int ecm_conv3d_c1_dgrad(const float* gy, const float* w, float* gx, int B, int Ci, int D, int H, int W, void* stream);
long long ecm_conv3d_c1_wgrad_scratch_bytes(int B, int Ci, int D, int H, int W);
int ecm_conv3d_c1_wgrad(const float* x, const float* gy, float* gw, void* scratch, long long scratch_bytes,
                        int B, int Ci, int D, int H, int W, void* stream);
/* ABI 4: the whole tail of a classifier, GroupNorm(32) + ReLU + Conv3d(32 -> 1) (cmfsm.py:621-634: classifN[0][1], [1], [2]),
 * with the normalisation applied while x is staged: x is the RAW output of classifN[0][0] and mean_rstd [B,32,2] its group
 * statistics from ecm_gn3d_stats; h = relu(gn(x)) is never written.  Ci must be 32 (one channel per group).  fwd: y [B,1,D,H,W];
 * wgrad: gw [1,32,27] = sum gy (x) h.  The data gradient w.r.t. h is ecm_conv3d_c1_dgrad, the GroupNorm backward
 * ecm_gn3d_bwd with y == NULL (ReLU mask recomputed from x). */
int ecm_conv3d_c1_gn_fwd(const float* x, const float* mean_rstd, const float* gamma, const float* beta, const float* w,
                         float* y, int B, int Ci, int D, int H, int W, void* stream);
int ecm_conv3d_c1_gn_wgrad(const float* x, const float* mean_rstd, const float* gamma, const float* beta, const float* gy,
                           float* gw, void* scratch, long long scratch_bytes, int B, int Ci, int D, int H, int W, void* stream);

/* ConvTranspose3d k=3, stride 2, pad 1, output_padding 1 (cmfsm.py:262-281): x [B,Ci,D,H,W] -> y [B,Co,Do,Ho,Wo],
 * Do = 2D (or 2D-1 when used as the data gradient of a stride-2 conv over an odd extent).
 * Weight in the reference layout [Ci,Co,3,3,3]; a Conv3d weight [Co_f,Ci_f,27] is the same memory layout
 * for its own data gradient (Ci := Co_f, Co := Ci_f).  packed: 27*Ci*CoP floats. */
int ecm_deconv3d_pack_weight(const float* w, float* packed, int Ci, int Co, void* stream);
int ecm_deconv3d_k3s2_fwd(const float* x, const float* wpacked, float* y,
                          int B, int Ci, int Co, int D, int H, int W, int Do, int Ho, int Wo, void* stream);

/* gw[co,ci,27] (reference Conv3d layout) = sum_{b,o} gy[b,co,o] * x[b,ci,o*st+k-1].
 * x: [B,Ci,D,H,W], gy: [B,Co,Do,Ho,Wo]; scratch >= ecm_conv3d_wgrad_scratch_bytes(...) */
long long ecm_conv3d_wgrad_scratch_bytes(int B, int Ci, int Co, int D, int H, int W, int stride);
int ecm_conv3d_k3_wgrad(const float* x, const float* gy, float* gw, void* scratch, long long scratch_bytes,
                        int B, int Ci, int Co, int D, int H, int W, int stride, void* stream);

/* General 2-D convolution family (conv2d.hip), same implicit-GEMM kernel: the encoder's Conv2d layers (feature_extraction,
 * cmfsm.py:126-236: 3x3 with stride 1|2 and dilation 1|2|4, the 3-channel stem, 64/128/320-channel stages, 1x1 projections),
 * the class-indexed convolutions of the collapsed cost volume (3x3 32->480, sheared 3x5 32->192; cmfsm.py:667-684) and all
 * stride-1 data gradients (flip_transpose packing).  (kh,kw,stride,dil) in {(3,3,1,1|2|4), (3,3,2,1), (3,5,1,1), (1,1,1|2,1)}.
 *   y[b,co,oh,ow] = sum w[co,ci,i,j] x[b,ci,oh*stride - pad_top + i*dil, ow*stride - pad_left + j*dil]   (0 outside x)
 * x: [B,Ci,H,W]; y: [B,Co,Ho,Wo] -- Ho, Wo and the top/left padding are given explicitly (the bottom/right padding is
 * whatever Ho, Wo imply), which also covers asymmetric padding.  packed: ecm_conv2d_packed_floats_ex floats, filled by
 * ecm_conv2d_pack_weight_ex from the reference layout [Co,Ci,kh,kw] (flip_transpose: the data-gradient operator with
 * Cin' = Co, Cout' = Ci; call the size query with the swapped channel counts). */
long long ecm_conv2d_packed_floats_ex(int Ci, int Co, int kh, int kw);
int ecm_conv2d_pack_weight_ex(const float* w, float* packed, int Co, int Ci, int kh, int kw, int flip_transpose, void* stream);
int ecm_conv2d_fwd_ex(const float* x, const float* wpacked, float* y, int B, int Ci, int Co, int H, int W, int kh, int kw,
                      int stride, int dil, int pad_top, int pad_left, int Ho, int Wo, void* stream);
/* gw[co,ci,kh,kw] = sum_{b,oh,ow} gy[b,co,oh,ow] x[b,ci,oh*stride - pad_top + i*dil, ow*stride - pad_left + j*dil] for the
 * same family; gy: [B,Co,Ho,Wo].  Deterministic (per-workgroup partials, fixed-order sum). */
long long ecm_conv2d_wgrad_ex_scratch_bytes(int B, int Ci, int Co, int Ho, int Wo, int kh, int kw, int stride);
int ecm_conv2d_wgrad_ex(const float* x, const float* gy, float* gw, void* scratch, long long scratch_bytes, int B, int Ci,
                        int Co, int H, int W, int kh, int kw, int stride, int dil, int pad_top, int pad_left, int Ho, int Wo,
                        void* stream);
/* Data gradient of a stride-2 3x3 Conv2d (pad 1) = ConvTranspose2d(k 3, stride 2, pad 1): x [B,Ci,H,W] -> y [B,Co,Ho,Wo],
 * Ho in {2H-1, 2H}.  w: a Conv2d weight [Co_f = Ci here, Ci_f = Co here, 3, 3] as it stands.  Co <= 64, Ci % 4 == 0. */
int ecm_deconv2d_pack_weight(const float* w, float* packed, int Ci, int Co, void* stream);
int ecm_deconv2d_k3s2_fwd(const float* x, const float* wpacked, float* y, int B, int Ci, int Co, int H, int W, int Ho, int Wo,
                          void* stream);

/* GroupNorm(32 groups, eps) over [B,C,S] (S = D*H*W), cmfsm.py:58; deterministic fixed-order reductions.
 * scratch for every call: >= ecm_gn3d_scratch_bytes(B,C,S).
 * fwd: y = relu?( (x-mean)*rstd*gamma[c]+beta[c] (+ skip) ) (skip may be NULL) and mean_rstd [B,32,2] in ONE pass over x
 *      (a cluster of co-resident workgroups keeps the span in registers across the reduction); shapes that do not fit
 *      that kernel run stats + apply.  stats / apply are also exported on their own. */
long long ecm_gn3d_scratch_bytes(int B, int C, long long S);
int ecm_gn3d_fwd(const float* x, const float* gamma, const float* beta, const float* skip, float* y,
                 float* mean_rstd, void* scratch, long long scratch_bytes, int B, int C, long long S,
                 int relu, float eps, void* stream);
int ecm_gn3d_stats(const float* x, float* mean_rstd, void* scratch, long long scratch_bytes,
                   int B, int C, long long S, float eps, void* stream);
int ecm_gn3d_apply(const float* x, const float* mean_rstd, const float* gamma, const float* beta,
                   const float* skip, float* y, int B, int C, long long S, int relu, void* stream);
/* The one-pass forms of ecm_gn3d_fwd / ecm_gn3d_bwd make workgroups of one launch wait for each other (a cluster of
 * <= 128 workgroups per (sample, group) span exchanges partial sums).  Members are assigned by tickets drawn at run
 * time, so progress needs only one cluster's worth of this launch's workgroups running, whatever else shares the
 * device; every wait is bounded in wall time.  If a bound expires (device shared so heavily that a cluster never became
 * resident within the poll time), the affected outputs are NaN AND a sticky, process-wide error word is set from the
 * device: every later ecm_gn3d_* call returns ECM_EASYNC, and ecm_async_status(clear) reports (and optionally clears) it
 * -- call it after synchronising the stream at the end of a step.  Never a hang, never a silent success.
 *   ecm_gn3d_cluster_mode(mode): 1 = cluster kernels (default), 0 = two-stage kernels only (no inter-workgroup waits:
 *     the safe choice when many processes share one device), 2 / 3 = diagnostics (static member ids / undersized grid that
 *     forces the timeout path), 4 = as 1, and ALSO on a stream that is being captured into a HIP graph (by default a capturing
 *     stream gets the two-stage kernels: two graphs with cluster launches replayed concurrently would starve each other, and
 *     cluster launches of one process are otherwise ordered across streams by the library -- mode 4 is the caller's promise
 *     that such graphs replay one at a time); any other value only queries.  Returns the previous mode.  Env
 *     ECM_GN_CLUSTER_MODE presets it.
 *   ecm_gn3d_poll_ms(ms): wait bound in milliseconds (default 2000; ms <= 0 only queries).  Returns the previous bound. */
int ecm_gn3d_cluster_mode(int mode);
int ecm_gn3d_poll_ms(int ms);
int ecm_async_status(int clear);

/* Backward of y = relu?(gn(x) + skip): writes gx, gskip (NULL to skip it), ggamma[C] and gbeta[C].
 * ReLU mask (relu != 0): from the forward OUTPUT y when y != NULL; with y == NULL it is recomputed from x, which needs
 * beta and is only valid for a forward WITHOUT skip (saves one tensor read per pass).  beta may be NULL otherwise. */
int ecm_gn3d_bwd(const float* x, const float* mean_rstd, const float* gamma, const float* beta, const float* y,
                 const float* gy, float* gx, float* gskip, float* ggamma, float* gbeta, void* scratch,
                 long long scratch_bytes, int B, int C, long long S, int relu, void* stream);

/* The class-indexed 2-D kernels of that collapse from dres0[0][0].weight w [Co,2C,3,3,3] (reference layout):
 * wP [15,Co,C,3,3] (one 3x3 kernel per (wedge class, depth-edge class) of the reference-image half) and wQ [6,Co,C,3,5]
 * (sheared 3x5 kernels of the target-image half); bwd: gw from (gwP, gwQ), every element written. */
int ecm_costvol_class_weights_fwd(const float* w, float* wP, float* wQ, int Co, int C, void* stream);
int ecm_costvol_class_weights_bwd(const float* gwP, const float* gwQ, float* gw, int Co, int C, void* stream);

/* The same two calls for a caller that KEEPS the one-pass kernels' exchange memory across calls: `cluster` is a device
 * buffer of >= ecm_gn3d_cluster_bytes(B) bytes that was filled once by ecm_gn3d_cluster_preset (all-ones) and is used by one
 * stream at a time.  The kernels leave it in the preset state when they finish (the last member of a cluster restores the
 * cluster's slots, the last ticket draw the ticket counter), so no memset is issued per launch (ecm_gn3d_fwd / _bwd issue one:
 * 172 per cmfsm training step).  After an asynchronous time-out (ECM_EASYNC) the buffer must be preset again.  `scratch` as
 * for ecm_gn3d_fwd / _bwd (two-stage partials and the per-channel sums).
 * The one-pass kernels keep 64 KB of dynamic LDS per workgroup (two workgroups per CU): part of a slab's output waits there and
 * is stored under the NEXT slab's exchange (csrc/gn3d.hip: STASH_V4).  y / gx must not alias any input of the call. */
long long ecm_gn3d_cluster_bytes(int B);
int ecm_gn3d_cluster_preset(void* cluster, long long cluster_bytes, void* stream);
int ecm_gn3d_fwd_p(const float* x, const float* gamma, const float* beta, const float* skip, float* y,
                   float* mean_rstd, void* scratch, long long scratch_bytes, void* cluster, long long cluster_bytes,
                   int B, int C, long long S, int relu, float eps, void* stream);
int ecm_gn3d_bwd_p(const float* x, const float* mean_rstd, const float* gamma, const float* beta, const float* y,
                   const float* gy, float* gx, float* gskip, float* ggamma, float* gbeta, void* scratch,
                   long long scratch_bytes, void* cluster, long long cluster_bytes, int B, int C, long long S,
                   int relu, void* stream);

/* Cost volume + dres0's first Conv3d (cmfsm.py:667-684) without the 4-D volume: both halves of the concat volume are
 * constant along a line in (d,x), so the 3x3x3 convolution collapses to 2-D convolutions of the feature maps (host side):
 *   P[b, classP, co, y, x]     classP(d,x) = (clamp(d-x,-2,2)+2)*3 + edge(d)        15 classes, reference-image half
 *   Qp[b, classQ, co, y, u+2]  classQ(d,x) = edge(d)*2 + (x == w-1), u = x-d         6 classes, target-image half (3x5)
 *   edge(d) = 0 / 1 / 2 for d == 0 / interior / d == D-1
 * fwd: y[B,Co,D,h,w] = P[classP(d,x)][x] + Qp[classQ(d,x)][x-d+2] where d-x < 3, else 0.
 * bwd: gP [B,15,Co,h,w], gQp [B,6,Co,h,w+2] = sums of gy over the d of each class (adjoint of fwd).  D >= 2. */
int ecm_costvol_conv_assemble_fwd(const float* P, const float* Qp, float* y, int B, int Co, int D, int h, int w, void* stream);
int ecm_costvol_conv_assemble_bwd(const float* gy, float* gP, float* gQp, int B, int Co, int D, int h, int w, void* stream);

/* Input side of the harness (cmf/loader/Flying3d.py:49-99): a batch of resident float32 frames [B,H,W,7] = (left RGB,
 * right RGB, disparity) -> left, right [B,3,th,tw] = ((v/255) - mean[c]) / std[c], disp [B,th,tw], and optionally
 * image [B,3,th,tw] = the raw left image in CHW (NULL to skip).  Output row r < split reads frame row crop_y0[b]+r, row
 * r >= split reads frame row H-tail+(r-split) (eval: split 540, tail 36 -> 576 rows; train: split == th); columns
 * crop_x0[b] .. +tw.  crop_y0 / crop_x0 / mean3 / std3 are HOST arrays (B ints / 3 floats).  Bit-identical to the loader. */
int ecm_frame_prep(const float* frames, float* left, float* right, float* disp, float* image, int B, int H, int W,
                   const int* crop_y0, const int* crop_x0, int th, int tw, int split, int tail,
                   const float* mean3, const float* std3, void* stream);

/* KITTI evaluation frames (cmf/loader/KITTI.py:98-108): pad [B,H,W,7] frames to th x tw (384 x 1248) at the TOP and LEFT
 * by repeating the first th-H rows / tw-W columns; disparity is 0 wherever the source row < th-H or the source column <
 * tw-W (the loader zeroes it through numpy views, which also hits the un-padded frame).  Needs th-H <= H, tw-W <= W. */
int ecm_frame_prep_kitti_eval(const float* frames, float* left, float* right, float* disp, float* image, int B,
                              int H, int W, int th, int tw, const float* mean3, const float* std3, void* stream);

/* Packed shards (frame contents per flying3ddata.py:34-39): rgb6 uint8 [B,H,W,6] (left RGB, right RGB) and
 * disp_in [B,H,W] as fp16 (disp_is_half != 0) or fp32.  Same outputs and arithmetic as ecm_frame_prep; mode 0 = the
 * window / split-tail form with crop_y0/crop_x0/split/tail as there, mode 1 = the KITTI top-left padding (crop arrays,
 * split and tail ignored).  Colour outputs are bit-identical to the float32-frame path on the same pixel values. */
int ecm_frame_prep_packed(const unsigned char* rgb6, const void* disp_in, int disp_is_half, float* left, float* right,
                          float* disp, float* image, int B, int H, int W, const int* crop_y0, const int* crop_x0,
                          int th, int tw, int split, int tail, const float* mean3, const float* std3, int mode, void* stream);

/* SceneFlow evaluation (test.py:69-94): pred [B,Hp,Wp] (= output3 squeezed) and gt [B,Hg,Wg], both cropped to
 * [:crop_h, :crop_w] (540 x 960).  out6 (device) = [epe, epe_non, epe_true, n, n_non, n_true]: mean |pred - gt| under
 *   mask = 0 <= gt < maxdisp;  mask_non = mask and x - gt >= 0;  mask_true = 0 < gt < maxdisp and x - gt >= 0
 * (x = column index).  An empty mask gives NaN like the reference's mean of an empty selection. */
long long ecm_eval_epe_scratch_bytes(long long n);
int ecm_eval_epe(const float* pred, const float* gt, float* out6, void* scratch, long long scratch_bytes, int B,
                 int Hp, int Wp, int Hg, int Wg, int crop_h, int crop_w, float maxdisp, void* stream);

/* KITTI submission image (test_kitti.py:163-168): out[b,y,x] = (uint16)(pred[b, Hp-h[b]+y, Wp-w[b]+x] * scale) for
 * y < h[b], x < w[b] (the loader padded at the top and left), 0 elsewhere; scale = 256.  The cast is numpy's on the
 * reference's host: truncation toward zero, low 16 bits of the 32-bit integer, 0 for NaN / out-of-int32-range values.
 * pred: [B,Hp,Wp]; out: uint16 [B,Ho,Wo]; h, w: HOST arrays of B ints.  Integer output: bit-exact. */
int ecm_disp_to_u16(const float* pred, unsigned short* out, int B, int Hp, int Wp, const int* h, const int* w,
                    int Ho, int Wo, float scale, void* stream);

/* Harness loss + metrics (train.py:162,172-174; train_kitti.py:205-216) over n = B*H*W pixels; mask = 0 < gt < maxdisp.
 * out8 (device): [loss, #mask, epe(p3), err3(p3) in %, mean smooth-L1 of p1, p2, p3, 0];
 * loss = w1*m1 + w2*m2 + w3*m3 (reference weights 0.5 / 0.7 / 1.0).  Empty mask -> NaN (as the reference's empty mean).
 * bwd: g_k = gloss[0] * w_k / #mask * clamp(p_k - gt, -1, 1) inside the mask, 0 outside (gloss: device scalar). */
long long ecm_stereo_loss_scratch_bytes(long long n);
int ecm_stereo_loss_fwd(const float* p1, const float* p2, const float* p3, const float* gt, float* out8,
                        void* scratch, long long scratch_bytes, long long n, float maxdisp,
                        float w1, float w2, float w3, void* stream);
int ecm_stereo_loss_bwd(const float* p1, const float* p2, const float* p3, const float* gt, const float* out8,
                        const float* gloss, float* g1, float* g2, float* g3, long long n, float maxdisp,
                        float w1, float w2, float w3, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ECM_HIP_H */
