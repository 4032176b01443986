/* udapose.h - C ABI of libudapose_hip.so: the MI355X (gfx950) kernels behind the mean-teacher UDA pose-estimation
 * hot path of VisionLearningGroup/UDA_PoseEstimation.
 *
 * The reference has no FFI layer: its hot path is reached through Python modules (lib.models.pose_resnet*,
 * lib.models.loss, lib.models.Style_net, lib.keypoint_detection, utils).  Each entry point below names the reference
 * interface (file:line under the reference tree) whose device work it replaces; the Python shells in
 * uda_poseestimation_amd/ bind them with ctypes.
 *
 * Conventions: extern "C"; plain pointers and sizes; every pointer is a DEVICE pointer unless its name starts with
 * h_ (host) or the comment says "host array"; `stream` is a hipStream_t passed as void*; return 0 on success,
 * negative UDAPOSE_ERR_* otherwise.
 *
 * State and re-entrancy.  No entry point reads an environment variable or a mutable global to decide what it runs: the
 * dispatch policy is an explicit value (udapose_policy: per network plan, or named by a convolution descriptor; a
 * default-initialised one is the measured production policy).  Device-side tables are built only by the explicit
 * preparation calls - udapose_net_create / udapose_net_bind / udapose_net_bind_grads for a network plan,
 * udapose_conv_prepare for a single geometry - which allocate and copy synchronously and therefore must run outside stream
 * capture; every compute call (udapose_conv2d_*, udapose_net_forward / backward / pack_weights / apply_running, all the
 * element-wise and loss kernels) then neither allocates nor synchronises, and returns UDAPOSE_ERR_NOT_PREPARED instead of
 * building a missing table (a per-op convolution call on an unprepared geometry outside capture prepares it itself, once per
 * device, under a lock).  Distinct network plans may be driven concurrently from different host threads on different
 * streams; ONE plan is not re-entrant (its scratch workspace belongs to one call at a time).  Kernel attributes are set
 * once per device, so one process may drive several GPUs.
 * Activations are NHWC bf16 (fp16 in the fp16 build) unless stated; heat-maps are NCHW fp32.
 */
#ifndef UDAPOSE_H
#define UDAPOSE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define UDAPOSE_OK 0
#define UDAPOSE_ERR_ARG (-1)
#define UDAPOSE_ERR_LAUNCH (-2)
#define UDAPOSE_ERR_UNSUPPORTED (-3)
#define UDAPOSE_ERR_NOT_PREPARED (-4)   /* a device table this call needs was not built (udapose_net_bind*, udapose_conv_prepare) */

int udapose_version(void);
/* element type this build stores and multiplies: 0 = bf16 (libudapose_hip.so), 1 = fp16 (libudapose_hip_f16.so: the same
 * sources with -DUDAPOSE_ELEM_F16; every `void*` activation / packed-weight pointer below is then fp16) */
int udapose_elem_kind(void);

/* ---------------------------------------------------------------- convolution family
 * Replaces torch.nn.Conv2d / ConvTranspose2d as used by torchvision Bottleneck (lib/models/resnet.py:8-10,25-40),
 * Upsampling (lib/models/pose_resnet.py:33-43), the head (pose_resnet.py:74) and the VGG encoder / decoder
 * (lib/models/Style_net.py:32-118). */
/* Explicit dispatch policy (tests force a code path with it, bench.py's tuning flags run A/B comparisons through it).
 * udapose_policy_default fills in the production policy; fields: igemm_tile / wgrad_tile / wgrad_ksplit -1 = heuristics;
 * igemm_h3: run-staged 3x3 form 0 off, 1 measured per-shape policy, 2 / 3 force the 64- / 128-row form; wgrad_group: one grouped weight-gradient launch per tile class in udapose_net_backward (0: layer by layer),
 * wgrad_stages: 64-pixel stages a work-group reduces before a layer's pixel range is split; bn_bwd_fused: dgrad epilogues
 * mask for the consumer BatchNorm and reduce its backward sums; bn_fwd_chunked / bn_bwd_chunked: finalize + apply of the wide,
 * small-spatial BatchNorm layers in one launch (0 off, 1 on, > 1: on with that target work-group count instead of 1024);
 * wgrad_fastgeo: loader of the weight-gradient kernels on power-of-two maps (0 general, 1 bit-field coordinates, 2 buffer loads
 * with out-of-range zero fill and an unrolled ring: the production form); wgrad_row3: weight gradients of 3x3 stride-1 convolutions with one work-group per (64x64 tile, filter row) - the row's three taps
 * share one staged dy tile and one x window (3x the FLOPs per byte filled into LDS, which is what bounds these kernels);
 * igemm_wg_min: 128x64 tiles as soon as they give that many work-groups, else 64x64; bn3_mask: block outputs save a ReLU bit mask
 * that the masking data gradients read instead of z; stem_fused: 1 = the stem's BN + ReLU + max-pool in one sweep, 2 = also the max-pool
 * backward gathered inside the BN backward; timeline: device buffer ([work-groups][8] uint64) for
 * per-work-group s_memrealtime stamps, or NULL. */
typedef struct {
    int igemm_tile, igemm_h3, igemm_lean, igemm_short_lds, igemm_tap0;
    int wgrad_tile, wgrad_ksplit, wgrad_fastgeo;
    int wgrad_group, wgrad_stages, wgrad_group_stem;
    int bn_bwd_fused, bn_fwd_chunked, bn_bwd_chunked, bn_bwd_pre_legacy;
    int igemm_wg_min;
    int wgrad_row3;
    int bn3_mask;
    int stem_fused;
    int debug_sync;
    int igemm_big_min;    /* > 0: 128x128 tiles (2-stage ring) for single-class launches with Co % 128 == 0 whose 128x64 grid has at least this
                           * many work-groups - the style network's large maps, run on one stream (+13-18 % there); 0 (default): never */
    int patch_conv;       /* reflection-padded 3x3 stride-1 convolutions (the style network) through the patch-staged kernels (input patch staged once,
                           * not once per tap): 0 never (the implicit GEMM for every layer), 1 the 64 -> 3 and 3 -> 64 end layers, 2 (default)
                           * the trunk layers too, 3 = 2 with 128 output channels per work-group in the 16-bit form */
    int eval_fold;        /* 1 (default): eval-mode network forwards (validate(), train_human.py:461-500) apply BatchNorm's running-statistics scale /
                           * shift, the residual and the ReLU in the convolution's epilogue: no BN-apply launch, no pre-BN tensor; 0: conv + apply */
    int bn_xcd_rows;      /* 1 (default): the BatchNorm apply kernels give XCD k the k-th eighth of the pixel rows - what the implicit GEMMs' work-groups on
                           * XCD k wrote and will read - so activations cross the conv <-> BatchNorm kernel boundaries through one L2; 0: interleaved */
    int wgrad_det;        /* 1 (default, round 6): split weight-gradient reductions of the grouped launches store per-split partial tiles into the pass's
                           * workspace and ONE launch adds them in split order (bit-reproducible gradients; the `loss.backward()` of
                           * train_human.py:436 run twice gives the same bits); 0: fp32 atomics into cleared tensors, arrival order */
    int igemm_ns3_k;      /* 64x64 implicit-GEMM tiles take the 3-stage LDS ring from this reduction length on (K = taps x Ci), the 2-stage ring
                           * below it; 0 = the default, 2048 */
    void* timeline;
} udapose_policy;
void udapose_policy_default(udapose_policy* p);

typedef struct {
    int N, Hi, Wi, Ci;   /* input NHWC (Ci multiple of 32, or exactly 8 for 3-channel images padded to 8) */
    int Co, KH, KW, stride, pad;
    int transposed;      /* 1: ConvTranspose2d(k, stride, pad), output_padding 0 */
    int reflect;         /* 1: ReflectionPad2d(pad) instead of zero padding */
    int upsample;        /* 1: input is read through nn.Upsample(scale_factor=2, mode='nearest') */
    const udapose_policy* policy;   /* host pointer, NULL = production policy */
} udapose_conv_desc;
/* builds (once per device) the small tap tables the three directions of this geometry use: the only allocation a convolution
 * ever needs; call it before capturing a stream that will run udapose_conv2d_* on the geometry */
int udapose_conv_prepare(const udapose_conv_desc* d);

#define UDAPOSE_EPI_RELU 1
#define UDAPOSE_EPI_OUT_F32 2
#define UDAPOSE_EPI_F32 4   /* x, w_fwd ([Co][taps][Ci] fp32), res and y are fp32: exact fp32 MFMA path (Ci multiple of 32 or 8) */
/* x, w_fwd, res - and y unless UDAPOSE_EPI_OUT_F32 - are "f16x2" split tensors: the FAST fp32-grade mode (forward only).  A split
 * tensor has the byte footprint and addressing of the fp32 tensor of the same shape; every group of 8 consecutive channels
 * (32 bytes) holds [8 x h fp16][8 x l fp16] with value = h + l * 2^-11, |value| <= 65504 (udapose_f32_to_split /
 * udapose_split_to_f32 convert).  The kernel multiplies with three v_mfma_f32_16x16x32_f16 per 32-deep K step (h.h, h.l, l.h; fp32
 * accumulation): products carry ~2^-22 relative error against the fp32 the reference computes the teacher, validate() and the
 * style network in (train_human.py:346-358,461-500), at 3/16 of the exact-fp32 MFMA's cost. */
#define UDAPOSE_EPI_SPLIT 8
void udapose_conv_out_hw(const udapose_conv_desc* d, int* Ho, int* Wo);
int udapose_conv_stat_rows(const udapose_conv_desc* d);
/* y[N,Ho,Wo,Co] = conv(x, w_fwd) (+bias[Co]) (+res) (ReLU); stats (optional): [stat_rows][2][Co] fp32 partial
 * (sum, sum of squares) of the fp32 result before bias/res - the BatchNorm batch statistics.  w_fwd: bf16 [Co][KH*KWp][Ci]. */
int udapose_conv2d_fwd(void* stream, const udapose_conv_desc* d, const void* x, const void* w_fwd, void* y, const void* res,
                       const float* bias, float* stats, int epilogue_flags);
/* dx[N,Hi,Wi,Ci] = conv^T(dy, w_bwd) (+res);  w_bwd: bf16 [Ci][KH*KW][Co]; out_f32: dx stored as fp32 (gradients that
 * feed a BatchNorm backward close to the loss, where the BN projection cancels most of the gradient) */
int udapose_conv2d_bwd_data(void* stream, const udapose_conv_desc* d, const void* dy, const void* w_bwd, void* dx, const void* res,
                            int out_f32);
/* The same dgrad when dx is the gradient dz entering a training-mode BatchNorm (+ReLU) whose input was bn_y [N,Hi,Wi,Ci] bf16:
 * the epilogue writes g = dz * mask to dx (mask: bn_z > 0 when bn_z is given - BN + residual + ReLU -, else
 * bn_y*gamma*invstd + (beta - mean*gamma*invstd) > 0, the forward's own expression) and one partial row per m-tile of
 * (sum g, sum g * (bn_y - mean) * invstd) to slab[rows][2][Ci] fp32, rows = the return value of udapose_conv_bwd_stat_rows.
 * That BatchNorm's backward then needs no reduction pass (reference: torch autograd of lib/models/resnet.py's
 * conv-bn-relu chains, train_human.py:338-444). */
int udapose_conv_bwd_stat_rows(const udapose_conv_desc* d);
int udapose_conv2d_bwd_data_bn(void* stream, const udapose_conv_desc* d, const void* dy, const void* w_bwd, void* dx, const void* res,
                               int out_f32, const void* bn_y, const void* bn_z, const float* bn_mean, const float* bn_invstd,
                               const float* bn_gamma, const float* bn_beta, float* slab);
/* dw fp32 [Co][KH*KWp][Ci] (transposed: [Ci][KH*KW][Co]) = (accumulate ? dw : 0) + sum_pixels dy * x.  This per-layer call splits long pixel
 * reductions over work-groups that add with fp32 atomics (arrival order: last-bit differences between runs; policy wgrad_ksplit = 1 forbids the
 * split); the network plans' grouped launches (udapose_net_backward*, udapose_net_wgrad_pair) are bit-reproducible (udapose_policy.wgrad_det). */
int udapose_conv2d_bwd_weight(void* stream, const udapose_conv_desc* d, const void* dy, const void* x, float* dw, int accumulate);
/* weight packing from fp32: cast (n % 8 == 0); per-tap transpose [A][T][B] -> [B][T][A]; strided gather with zero padding */
int udapose_cast_f32_bf16(void* stream, const float* src, void* dst, size_t n);
int udapose_transpose_cast(void* stream, const float* src, void* dst, int A, int T, int B);
int udapose_pack_strided(void* stream, const float* src, void* dst, int A, int KH, int KWp, int KW, int Bp, int B, long sa, long skh,
                         long skw, long sb);

/* ---------------------------------------------------------------- layout conversion at the NCHW fp32 boundary */
int udapose_nchw_f32_to_nhwc_bf16(void* stream, const float* src, void* dst, int N, int C, int HW, int Cpad);
/* the same into fp32 NHWC (style path at the reference's precision: it runs outside autocast, train_human.py:347-356) */
int udapose_nchw_f32_to_nhwc_f32(void* stream, const float* src, float* dst, int N, int C, int HW, int Cpad);
/* the same into an f16x2 split NHWC tensor (UDAPOSE_EPI_SPLIT), and the element-wise conversions fp32 <-> split (n % 8 == 0;
 * udapose_f32_to_split may run in place) */
int udapose_nchw_f32_to_nhwc_split(void* stream, const float* src, void* dst, int N, int C, int HW, int Cpad);
int udapose_f32_to_split(void* stream, const float* src, void* dst, size_t n);
int udapose_split_to_f32(void* stream, const void* src, float* dst, size_t n);
/* optional per-channel clamp lo/hi[C] = the "recover" clamp of train_human.py:32-33,276,351,356 */
/* src_is_f32: 0 = the library's 16-bit element type, 1 = fp32, 2 = f16x2 split */
int udapose_nhwc_to_nchw_f32(void* stream, const void* src, int src_is_f32, float* dst, int N, int C, int HW, int Cstride,
                             const float* lo, const float* hi);

/* ---------------------------------------------------------------- BatchNorm2d, training mode (torch.nn.BatchNorm2d in
 * .train(): 107 layers of the pose net, train_human.py:320-321) */
int udapose_bn_finalize(void* stream, const float* stats, int stat_rows, int C, double count, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float eps,
                        float* scale, float* shift, float* save_mean, float* save_invstd);
int udapose_bn_eval_coeff(void* stream, int C, const float* gamma, const float* beta, const float* running_mean,
                          const float* running_var, float eps, float* scale, float* shift);
int udapose_bn_apply(void* stream, const void* y, const void* res, void* z, size_t numel, int C, const float* scale, const float* shift,
                     int relu);
int udapose_bn_bwd_rows(size_t npix);
/* dz (bf16, or fp32 when dz_is_f32) -> dy (+ masked g); slab: [bn_bwd_rows][2][C] fp32 scratch, coef: [3][C] fp32 scratch.
 * relu: 0 = none; 1 = ReLU mask from the saved output z (z > 0); 2 = mask recomputed from y as y*gamma*invstd +
 * (beta - mean*gamma*invstd) > 0, the forward's own expression: z is not read (valid when the BN output had no residual
 * added before its ReLU); beta = the BN bias parameter, needed for relu == 2 only. */
int udapose_bn_bwd(void* stream, const void* dz, int dz_is_f32, const void* z, const void* y, void* dy, void* gout, size_t npix, int C,
                   const float* gamma, const float* save_mean, const float* save_invstd, int relu, float* slab, float* coef,
                   float* dgamma, float* dbeta, float beta_acc, const float* beta);
/* The same backward when the dgrad that produced the gradient already applied the ReLU mask and reduced it
 * (udapose_conv2d_bwd_data_bn): g (bf16, or fp32 when g_is_f32) and slab[rows][2][C] -> dgamma, dbeta (beta_acc*old + new) and
 * dy = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)); coef: [3][C] fp32 scratch.  No reduction pass over the activations. */
int udapose_bn_bwd_pre(void* stream, const void* g, int g_is_f32, const void* y, void* dy, size_t npix, int C, const float* gamma,
                       const float* save_mean, const float* save_invstd, const float* slab, int rows, float* coef, float* dgamma,
                       float* dbeta, float beta_acc);

/* ---------------------------------------------------------------- pooling (ResNet stem maxpool 3x3 s2 p1, resnet.py:30;
 * VGG MaxPool2d(2,2,ceil_mode=True), Style_net.py:72) */
int udapose_maxpool3x3s2_fwd(void* stream, const void* x, void* y, unsigned char* idx, int N, int H, int W, int C);
int udapose_maxpool3x3s2_bwd(void* stream, const void* dy, const unsigned char* idx, void* dx, int N, int H, int W, int C);
int udapose_maxpool2x2_ceil(void* stream, const void* x, void* y, int N, int H, int W, int C);
int udapose_maxpool2x2_ceil_f32(void* stream, const float* x, float* y, int N, int H, int W, int C);   /* fp32 NHWC (Style_net.py:72) */
int udapose_maxpool2x2_ceil_split(void* stream, const void* x, void* y, int N, int H, int W, int C);   /* f16x2 split NHWC */

/* ---------------------------------------------------------------- whole pose network (lib/models/pose_resnet.py:59-126:
 * PoseResNet.forward = head(upsampling(backbone(x)))), parameters by index in .parameters() order (host arrays of
 * device pointers), buffers in .buffers() order.  4-D weights are fp32 in channels_last physical layout. */
typedef void* udapose_net_t;
/* fp32 == 1: fp32 activations and exact fp32 MFMA, FORWARD ONLY (the reference runs the teacher and validate() in fp32,
 * train_human.py:347-358,461-500); fp32 == 2: the fast fp32-grade form of the same (f16x2 split activations and weight packs,
 * UDAPOSE_EPI_SPLIT; pre-BatchNorm conv outputs and statistics in fp32), FORWARD ONLY; fp32 == 0: the library's 16-bit
 * element type with fp32 accumulation, forward and backward. */
/* bit 8 of `fp32` (value | 0x100): the three deconvolutions carry a bias parameter (`deconv_with_bias=True`, lib/models/pose_resnet.py:
 * 15,41,96) - parameter order weight, bias, then the BatchNorm's, as in the reference's Upsampling. */
/* bit 9 of `fp32` (value | 0x200): a FORWARD-ONLY plan (the teacher's forwards under torch.no_grad(), train_human.py:346-372, and
 * validate(), :461-500): nothing is kept for a backward, so the pre- and post-BatchNorm tensors of all layers rotate through five
 * scratch buffers laid out by liveness (udapose_net_act_bytes: ~0.3 GB instead of 2.8 GB at N = 32, 256x256) and stay resident
 * in the L2s / the Infinity Cache; udapose_net_backward* and udapose_net_bind_grads return UDAPOSE_ERR_UNSUPPORTED on such a plan. */
int udapose_net_create(const int layers[4], int num_keypoints, int N, int H, int W, int fp32, udapose_net_t* out);
void udapose_net_destroy(udapose_net_t net);
int udapose_net_num_params(udapose_net_t net);
int udapose_net_num_buffers(udapose_net_t net);
long long udapose_net_param_numel(udapose_net_t net, int i);
size_t udapose_net_wpack_bytes(udapose_net_t net);
size_t udapose_net_act_bytes(udapose_net_t net);
size_t udapose_net_ws_bytes(udapose_net_t net);
void udapose_net_out_shape(udapose_net_t net, int shape[4]);
/* the plan's dispatch policy (set it before udapose_net_bind_grads: the grouped weight-gradient tables depend on it) */
int udapose_net_set_policy(udapose_net_t net, const udapose_policy* p);
int udapose_net_get_policy(udapose_net_t net, udapose_policy* p);
/* Preparation (allocates + copies synchronously; outside stream capture; repeat when a pointer changes):
 *   bind:        tap tables of every layer geometry, the weight-packing job tables for (h_params, wpack), the
 *                running-statistics job table for h_buffers (may be NULL for a plan that never defers them);
 *   bind_grads:  the grouped weight-gradient tables for this placement of the gradient tensors (they hold offsets relative
 *                to h_grads[0]: any other set of buffers with the same relative placement reuses them).
 * pack_weights / forward / apply_running / backward return UDAPOSE_ERR_NOT_PREPARED if what they are given was not bound. */
int udapose_net_bind(udapose_net_t net, const void* const* h_params, void* const* h_buffers, void* wpack);
int udapose_net_bind_grads(udapose_net_t net, void* const* h_grads);
int udapose_net_pack_weights(udapose_net_t net, void* stream, const void* const* h_params, void* wpack, int with_bwd);
int udapose_net_forward(udapose_net_t net, void* stream, const float* x_nchw, const void* const* h_params, void* const* h_buffers,
                        const void* wpack, void* act, void* ws, float* out_nchw, int training, float momentum);
/* training: bit 0 = batch statistics (train mode), bit 1 = do NOT update the running statistics in this call; apply them
 * later, in program order, with udapose_net_apply_running (two forwards of one module running on different streams). */
int udapose_net_apply_running(udapose_net_t net, void* stream, const void* act, void* const* h_buffers, float momentum);
/* y += x over n fp32 values (y, x 16-byte aligned): sum of per-pass gradient buffers */
int udapose_axpy_f32(void* stream, float* y, const float* x, size_t n);
int udapose_net_backward(udapose_net_t net, void* stream, const float* dout_nchw, const void* const* h_params, const void* wpack,
                         void* act, void* ws, void* const* h_grads, float beta);
/* The same backward in two calls, cut after the first block of layer3, for a data-parallel step that overlaps the gradient
 * all-reduce with the backward (replaces nn.DataParallel's reduce, train_human.py:145-148,436): part 1 = head, deconvs,
 * layer4, layer3 and the weight gradients of those layers - a contiguous suffix of .parameters() starting at
 * udapose_net_grad_split_param(), final when part 1 has run; part 2 = layer2, layer1, stem and theirs, continuing from the
 * gradient part 1 left in `ws` (same act / ws / grads / beta as part 1; dout is ignored). */
int udapose_net_backward_part(udapose_net_t net, void* stream, const float* dout_nchw, const void* const* h_params, const void* wpack,
                              void* act, void* ws, void* const* h_grads, float beta, int part);
/* The same with the two halves of a part separable: phase 0 = the gradient chain of `part` followed by its grouped weight-gradient
 * launches (= udapose_net_backward_part; part 0 = the whole backward); phase 1 = the chain only; phase 2 = the grouped
 * weight-gradient launches of `part` only, on ANY stream that has waited for the chain (every layer owns its dy buffer in `ws`):
 * one device: the weight gradients of part 1 run under the gradient chain of part 2. */
int udapose_net_backward_phase(udapose_net_t net, void* stream, const float* dout_nchw, const void* const* h_params, const void* wpack,
                               void* act, void* ws, void* const* h_grads, float beta, int part, int phase);
/* The grouped weight-gradient launches (phase 2) of TWO passes of one plan whose gradient chains (phase 1) have run - each with its
 * own act / ws arenas, gradient tensors and beta - as ONE launch per tile class: the two student passes of a mean-teacher step end
 * together and their weight gradients are exposed there; one grid of twice the size has half the tail.
 * Determinism (round 6, udapose_policy.wgrad_det = 1, the default): a layer whose pixel range is split over several work-groups (layer1 / layer2,
 * the last deconvolution, the head, the stem) has every split store its partial tile into `ws`; one launch then adds the splits in split order
 * into the gradient tensor.  Two runs of a backward on the same inputs give the same bits (rounds 1-5 accumulated the splits with fp32
 * atomics in arrival order).  `ws` must have the size udapose_net_ws_bytes returns AFTER udapose_net_set_policy. */
int udapose_net_wgrad_pair(udapose_net_t net, void* stream, const void* act_a, void* ws_a, void* const* h_grads_a, float beta_a,
                           const void* act_b, void* ws_b, void* const* h_grads_b, float beta_b, int part);
long long udapose_net_grad_split_param(udapose_net_t net);

/* ---------------------------------------------------------------- heat-map losses and decode (fp32 NCHW rows [R=B*K][HW]) */
/* JointsMSELoss (lib/models/loss.py:39-49): rows[r] = 0.5*w[r]*mean_hw((p-g)^2); mean_out = mean_r rows (reduction='mean') */
int udapose_joints_mse_fwd(void* stream, const float* pred, const float* gt, const float* w, int R, int HW, float* rows, float* mean_out);
/* d pred = gscale[0] * w[r] * (p-g) / (R*HW) */
int udapose_joints_mse_bwd(void* stream, const float* pred, const float* gt, const float* w, const float* gscale, int R, int HW,
                           float* dpred);
/* ConsLoss (lib/models/loss.py:124-132): mean_out = sum(mask*(s-t)^2)/(R*HW) */
int udapose_cons_loss_fwd(void* stream, const float* stu, const float* tea, const unsigned char* mask, int R, int HW, float* rows,
                          float* mean_out);
int udapose_cons_loss_bwd(void* stream, const float* stu, const float* tea, const unsigned char* mask, const float* gscale, int R,
                          int HW, float* dstu);
/* ConsLoss(valid_mask=) (lib/models/loss.py:129-130: loss_map[valid_mask].mean()): `valid` [R/K][HW] selects (b,h,w) positions of
 * loss_map = mean over the K channels; valid_count = number of selected positions, on the device (udapose_mask_count):
 * mean_out = sum over selected of mask*(s-t)^2 / (K * valid_count) */
int udapose_mask_count(void* stream, const unsigned char* mask, size_t n, float* count);
int udapose_cons_loss_valid_fwd(void* stream, const float* stu, const float* tea, const unsigned char* mask, const unsigned char* valid,
                                const float* valid_count, int R, int K, int HW, float* rows, float* mean_out);
int udapose_cons_loss_valid_bwd(void* stream, const float* stu, const float* tea, const unsigned char* mask, const unsigned char* valid,
                                const float* valid_count, const float* gscale, int R, int K, int HW, float* dstu);
/* get_max_preds(_torch) (lib/keypoint_detection.py:9-37, utils.py:54-75) and rectify (utils.py:77-109): any output may
 * be NULL.  patch: [(2*rad+1)^2] fp32 Gaussian table built by the caller exactly as utils.py:93-98 does. */
int udapose_heatmap_argmax(void* stream, const float* hm, int R, int H, int W, float* maxvals, int* flat_idx, float* preds_xy,
                           float* rectified, const float* patch, int rad);
/* confidence mask (train_human.py:427-430): thr = k-th smallest of act[n]; mask[i] = (tea_mask[i]*act_local[i]) > thr */
int udapose_kth_mask(void* stream, const float* act, const float* tea_mask, int n, int k, float* thr_out, unsigned char* mask,
                     const float* act_local, int n_local);
/* PCK (lib/keypoint_detection.py:40-94) from decoded coordinates [B,K,2]; acc[K], avg_cnt[2] = (avg_acc, cnt) */
int udapose_pck(void* stream, const float* pred_xy, const float* gt_xy, int B, int K, float norm_x, float norm_y, float thr, float* acc,
                float* avg_cnt);

/* ---------------------------------------------------------------- optimizer sweeps over many tensors (device tables) */
int udapose_multi_chunk(void);
/* OldWeightEMA.step (utils.py:21-25): t = fl(fl(t*alpha) + fl(s*one_minus_alpha)), bit-exact two-rounding form */
int udapose_ema_multi(void* stream, const long long* tgt_ptrs, const long long* src_ptrs, const long long* sizes, const int* blk_tensor,
                      const long long* blk_off, int nblocks, float alpha, float one_minus_alpha);
/* torch.optim.Adam.step (train_human.py:139,286).  dev_state (optional, 8 floats: [step, 1-b1^step, sqrt(1-b2^step), lr,
 * grad_scale, -, -, -]): the step counter and bias corrections are advanced by the call itself on the device, and lr /
 * grad_scale are READ from it instead of the by-value arguments, so that a captured call follows an lr scheduler
 * (MultiStepLR, train_human.py:143,202) through an 8-byte copy; if NULL, `step`, `lr`, `grad_scale` are the host's. */
int udapose_adam_multi(void* stream, const long long* p, const long long* g, const long long* m, const long long* v,
                       const long long* sizes, const int* blk_tensor, const long long* blk_off, int nblocks, float lr, float beta1,
                       float beta2, float eps, float weight_decay, int step, float grad_scale, float* dev_state);
/* torch.optim.SGD(momentum, nesterov) (train_human.py:137); dev_state as for Adam ([0] = step counter, [3] = lr,
 * [4] = grad_scale; first_step is then `step == 1` on the device) */
int udapose_sgd_multi(void* stream, const long long* p, const long long* g, const long long* buf, const long long* sizes,
                      const int* blk_tensor, const long long* blk_off, int nblocks, float lr, float momentum, float weight_decay,
                      int nesterov, int first_step, float grad_scale, float* dev_state);

/* The whole tail of a mean-teacher step in ONE sweep over the parameters: torch.optim.Adam.step on the student
 * (train_human.py:437), OldWeightEMA.step into the teacher (utils.py:21-25, train_human.py:438) and the element-type weight
 * packs that the next forwards of both networks' plans need (what udapose_net_pack_weights would re-read the masters for):
 * 46 bytes per parameter instead of 58, four launches fewer.  Arithmetic identical to udapose_adam_multi followed by
 * udapose_ema_multi (bit for bit); parameters by index in .parameters() order (host arrays of device pointers); exp_avg /
 * exp_avg_sq entries are NULL for parameters without gradient (backbone.fc: EMA only).  bind_update builds the device job
 * table (allocates: outside capture; again when a pointer changes); dev_state as for udapose_adam_multi; do_adam = 0: EMA
 * and packs only. */
int udapose_net_bind_update(udapose_net_t student, udapose_net_t teacher, void* const* h_params_s, void* const* h_grads,
                            void* const* h_exp_avg, void* const* h_exp_avg_sq, void* const* h_params_t, void* wpack_s, void* wpack_t);
int udapose_net_fused_update(udapose_net_t student, udapose_net_t teacher, void* stream, void* const* h_params_s, void* const* h_grads,
                             void* const* h_exp_avg, void* const* h_params_t, void* wpack_s, void* wpack_t, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int step, float grad_scale, float* dev_state, float alpha,
                             float one_minus_alpha, int do_adam, long long grad2_delta_bytes);
/* grad2_delta_bytes != 0: the gradient is h_grads[i] + the tensor grad2_delta_bytes behind it (the second per-pass gradient buffer
 * of a step whose two backward passes ran on different streams; a multiple of 16): the sum udapose_axpy_f32 would have written
 * first, taken in the same sweep.  h_grads itself is left holding the first pass's share. */

/* Dynamic loss scaling = torch.cuda.amp.GradScaler (train_human.py:260,285-287,324,436-440) on the device, for the fp16 build.
 * dev_state is the optimizer's 8-float state: [5] = found_inf, [6] = loss scale S, [7] = growth tracker, [4] = 1/S.
 * check: raises found_inf if any gradient is inf / nan (then adam_multi / sgd_multi skip the step, counter included);
 * update: S *= backoff on found_inf, S *= growth after `interval` clean steps; clears found_inf, refreshes 1/S. */
int udapose_grad_scaler_check(void* stream, const long long* g, const long long* sizes, const int* blk_tensor, const long long* blk_off,
                              int nblocks, float* dev_state);
/* the same check over g + g2, g2 = the second per-pass gradient buffer at grad2_delta_bytes from the first (the sum the fused optimizer tail
 * forms itself, udapose_net_fused_update's grad2_delta_bytes): no separate axpy in front of the check (round 4) */
int udapose_grad_scaler_check2(void* stream, const long long* g, const long long* sizes, const int* blk_tensor, const long long* blk_off,
                               int nblocks, float* dev_state, long long grad2_delta_bytes);
int udapose_grad_scaler_update(void* stream, float* dev_state, float growth, float backoff, int interval);

/* ---------------------------------------------------------------- gradient all-reduce in bf16 on the wire (data parallel,
 * the 212 MB fp32 student-gradient buffer): pack = one rounding of the fp32 bucket to bf16 (zero padded to n_padded, a
 * multiple of the world size); after an all-to-all every rank holds the W ranks' copies of its shard: shard_mean adds them in fp32
 * (rank order), multiplies by 1/W and ROUNDS THE MEAN to bf16 for the all-gather - an averaged element therefore carries two bf16
 * roundings, one per contribution and one of the mean (all-gathering the mean in fp32 would cost 6 instead of 4 bytes per element and rank
 * on the wire); unpack widens the bucket back to fp32.  tests/test_gpu_hotpath.py emulates W = 2, 3, 8 ranks on one GPU. */
int udapose_comm_pack_bf16(void* stream, const float* src, long long n, void* dst_bf16, long long n_padded);
int udapose_comm_shard_mean(void* stream, const void* shards_bf16, int world, long long m, void* out_bf16);
int udapose_comm_unpack_bf16(void* stream, const void* src_bf16, float* dst, long long n);

/* ---------------------------------------------------------------- AdaIN (lib/models/Style_net.py:4-29,167-168), NHWC bf16
 * out = alpha*adain(content, style) + (1-alpha)*content; stats_out (optional) [N][C][4] = (mean_c, std_c, mean_s, std_s) */
int udapose_adain(void* stream, const void* content, const void* style, void* out, int N, int HWc, int HWs, int C, float eps,
                  float alpha, float* stats_out);
/* the same on fp32 NHWC features (the reference's precision); out may be NULL: statistics only (calc_mean_std) */
int udapose_adain_f32(void* stream, const float* content, const float* style, float* out, int N, int HWc, int HWs, int C, float eps,
                      float alpha, float* stats_out);
/* the same on f16x2 split NHWC features (UDAPOSE_EPI_SPLIT): statistics and blend in fp32, out (may be NULL) split;
 * alpha_dev != NULL: the blend factor is read from device memory at run time */
int udapose_adain_split(void* stream, const void* content, const void* style, void* out, int N, int HWc, int HWs, int C, float eps,
                        float alpha, const float* alpha_dev, float* stats_out);
/* the same with the blend factor read from device memory at run time (one float): a launch captured in a hipGraph then follows
 * the step's draw of alpha.  is_f32 selects the fp32 form (content / style / out are float*), else the library's element type. */
int udapose_adain_alpha_dev(void* stream, const void* content, const void* style, void* out, int N, int HWc, int HWs, int C, float eps,
                            const float* alpha_dev, float* stats_out, int is_f32);

/* ---------------------------------------------------------------- batched nearest inverse-affine re-warp
 * (torchvision.transforms.functional.affine x3 per sample, train_human.py:366-368,388-390,412,421-423): NCHW fp32;
 * theta [N][nstage][6] = the inverse affine matrices in application order; backward != 0: src = d(out), dst = d(in). */
int udapose_affine_nearest(void* stream, const float* src, float* dst, const float* theta, int N, int C, int H, int W, int nstage,
                           int backward);
/* f16x2 (UDAPOSE_EPI_SPLIT / split tensors) range check: the number of split STORES, since the last reset, of a value outside fp16's range
 * (|v| > 65504, which the format saturates, or NaN) by any kernel of THIS library on the current device.  Synchronous (reads device
 * counters): call it at an evaluation boundary - engine.validate() does and warns - never inside a stream capture. */
int udapose_split_saturations(int reset, unsigned long long* count);
/* mean of k re-warped teacher heat-map tensors (train_human.py:361-372 with `--k` > 1: `torch.mean(recons, dim=0)` per sample): h_views = HOST
 * array of k (1..8) device pointers to fp32 tensors of n elements; dst[i] = (v0[i] + v1[i] + ...) / k, added in view order in fp32. */
int udapose_mean_views(void* stream, const float* const* h_views, int k, float* dst, size_t n);
/* the loop's matrices on the device, in double precision, from the collated aug_param of lib/transforms/keypoint_detection.py:139:
 * params[n] = (angle, tx, ty, shear_x, shear_y, scale), angles in degrees (6 doubles per sample).  theta_fwd [N][3][6] (may be NULL):
 * translate by (tx, ty) / ratio | rotate by angle and scale | shear - the three warps of train_human.py:366-368,421-423;
 * theta_back [N][1][6] (may be NULL): the occlusion path's warp back (train_human.py:412).  A captured step computes its matrices
 * with this launch from a static parameter buffer: the host only copies 48 bytes per sample. */
int udapose_recon_thetas(void* stream, const double* params, int N, double ratio, float* theta_fwd, float* theta_back);

/* occlusion paste (train_human.py:399-409): for image i of img[n][C][H][W] (fp32) and boxes[i] = (r0,r1,c0,c1,rs,cs):
 * img[i][:, r0:r1, c0:c1] = img[i][:, rs:rs+(r1-r0), cs:cs+(c1-c0)] (source read completely before the write). */
int udapose_patch_paste(void* stream, float* img, const int* boxes, int n, int C, int H, int W, int max_patch_elems);
/* The occlusion DECISIONS (train_human.py:374-410) on the device, so that the step needs no read-back and can be captured:
 * conf / flat_idx [N][K] from udapose_heatmap_argmax of the teacher's re-warped heat-maps, u [N][4] uniform [0,1) draws (rate
 * test, key-point choice, patch row / column origin) -> boxes [N][6] for udapose_patch_paste (zero area when not selected) and
 * apply [N]; udapose_select_rows then keeps the occluded image for the selected samples only: dst[n] = apply[n] ? a[n] : b[n]. */
int udapose_occlusion_pick(void* stream, const float* conf, const int* flat_idx, const float* u, int N, int K, int w, double ratio,
                           int image_size, float rate, float thresh, int occlude_size, int* boxes, unsigned char* apply);
int udapose_select_rows(void* stream, float* dst, const float* a, const float* b, const unsigned char* flag, int N, size_t row_elems);

/* ---------------------------------------------------------------- per-launch timing of the MFMA kernels (bench.py roofline)
 * HIP events are recorded on the launch stream around every convolution launch between begin and end.
 * h_out9 (host): for kind in (fprop, dgrad, wgrad): launches, total milliseconds, total algorithmic FLOPs. */
/* ---------------------------------------------------------------- device-side data pipeline of the target views (the work
 * of the reference's DataLoader workers for the `_mt` datasets, lib/datasets/human36m_mt.py:76-161): images are uint8 NHWC
 * [N][H][W][3] as PIL arrays are; results are bit-exact with PIL's own arithmetic.
 * aug_affine_u8: torchvision F.affine on a PIL image = Image.transform(AFFINE, NEAREST) (lib/transforms/keypoint_detection.py:138):
 *   coef [N][6] int64 = PIL's 16.16 fixed-point coefficients (FIX(a0), FIX(a1), FIX(a2 + a0/2 + a1/2), FIX(a3), FIX(a4),
 *   FIX(a5 + a3/2 + a4/2)) of the inverse affine matrix (a0..a5), prepared on the host in double.
 * aug_color_op: one PIL.ImageEnhance step per image, in place (ColorJitter, train_human.py:68): op[n] 0 none, 1 brightness,
 *   2 contrast, 3 saturation; factor[n]; mean_scratch [N] int (the contrast step's rounded L mean).
 * aug_to_tensor: ToTensor + Normalize -> NCHW fp32.
 * gaussian_labels: generate_target (lib/datasets/util.py:12-70): kp [R][2] double pixels, vis [R] -> target [R][Hh][Wh],
 *   weight [R]; patch = the (2*rad+1)^2 Gaussian built by the caller as the reference builds it. */
int udapose_aug_affine_u8(void* stream, const unsigned char* src, unsigned char* dst, const long long* coef, int N, int H, int W);
int udapose_aug_color_op(void* stream, unsigned char* img, const int* op, const float* factor, int* mean_scratch, int N, int HW);
/* PIL.ImageFilter.GaussianBlur (T.GaussianBlur, lib/transforms/keypoint_detection.py:216-225) in place on img [N][H][W][3] uint8 through
 * the scratch buffer tmp (same size): three box-blur passes per direction in PIL's 8.24 fixed point, bit-exact.  prm[n] = (r, ww, fw)
 * as uint32 (box radius integer part, centre and far weights: data_gpu.pil_box_blur_params), r = 0xffffffff: sample n is left as is. */
int udapose_aug_gaussian_blur_u8(void* stream, unsigned char* img, unsigned char* tmp, const unsigned int* prm, int N, int H, int W);
/* T.RandomResizedCrop's image side (lib/transforms/keypoint_detection.py:456-521 -> resized_crop :66-88 = F.crop + F.resize(BILINEAR) on a
 * PIL image): src [N][Hs][Ws][3] uint8 -> dst [N][S][S][3] uint8, sample n's crop box[n] = (top, left, h, w) resampled to S x S with PIL's
 * two-pass 8-bit resampler (libImaging Resample.c: horizontal pass into the uint8 intermediate tmp [N][Hs][S][3], then vertical; 22-bit
 * fixed-point coefficients), bit-exact.  bounds [N][2][S][2] int32 = (first source index, tap count) per output column (axis 0) / row
 * (axis 1), coef [N][2][S][ksize] int32: prepared on the host in double as PIL does (data_gpu.pil_resample_coeffs). */
int udapose_aug_resized_crop_u8(void* stream, const unsigned char* src, unsigned char* dst, unsigned char* tmp, const int* box, const int* bounds,
                                const int* coef, int N, int Hs, int Ws, int S, int ksize);
int udapose_aug_to_tensor(void* stream, const unsigned char* img, float* out, int N, int HW, const float* mean3, const float* std3);
int udapose_gaussian_labels(void* stream, const double* kp, const float* vis, float* target, float* weight, int R, int Hh, int Wh,
                            double stride_x, double stride_y, const float* patch, int rad);
/* draw_labelmap_ori (lib/datasets/util.py:326-363), the animal pipelines' label generator as their datasets call it
 * (lib/datasets/real_animal_all_mt.py:274-283, animal_pose_mt.py:169-177,200-205; BASELINE.json configs[4]): pt [R][2] float32 = the 0-based
 * centres the datasets pass (`tpts[i] - 1`), truncated to int32 inside; vis [R] = pts[:, 2]; gate [R] uint8 = the datasets' `tpts[i, 1] > 0`
 * test (0: the row keeps vis and an empty map) -> target [R][Hh][Wh] fp32, weight [R] = vis * (whole stamp inside the map) where the
 * gate is open.  r3 = float32(3 * sigma); patch = the reference's (6 sigma + 1)^2 float64 stamp, 'Gaussian' or 'Cauchy', rounded to
 * float32 by the caller (psize = its side). */
int udapose_draw_labelmap_ori(void* stream, const float* pt, const float* vis, const unsigned char* gate, float* target, float* weight, int R,
                              int Hh, int Wh, float r3, const float* patch, int psize);

/* measurement: HIP events around every conv launch between begin and end (process-wide recorder, mutex-guarded) */
void udapose_prof_begin(void);
int udapose_prof_end(double* h_out9);

#ifdef __cplusplus
}
#endif
#endif
