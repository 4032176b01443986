// extern "C" surface of libudapose_hip.so (declared in include/udapose.h).
#include <cstdio>
#include "conv_plan.h"
#include "../../include/udapose.h"

int pw_nchw_f32_to_nhwc_bf16(hipStream_t, const float*, elem_t*, int, int, int, int);
int pw_nhwc_to_nchw_f32(hipStream_t, const void*, int, float*, int, int, int, int, const float*, const float*);
int pw_cast_f32_bf16(hipStream_t, const float*, elem_t*, size_t);
int pw_transpose_cast(hipStream_t, const float*, elem_t*, int, int, int);
int pw_pack_strided(hipStream_t, const float*, elem_t*, int, int, int, int, int, int, long, long, long, long);
int pw_bn_finalize(hipStream_t, const float*, int, int, double, const float*, const float*, float*, float*, long long*, float, float, float*, float*,
                   float*, float*, const float*);
int pw_bn_eval_coeff(hipStream_t, int, const float*, const float*, const float*, const float*, float, float*, float*);
int pw_bn_apply(hipStream_t, const elem_t*, const elem_t*, elem_t*, size_t, int, const float*, const float*, int, unsigned char*, int);
int pw_bn_bwd_rows(size_t);
int pw_bn_bwd_pre(hipStream_t, const void*, int, const elem_t*, elem_t*, size_t, int, const float*, const float*, const float*, const float*, int, float*,
                  float*, float*, float, int, int);
int pw_bn_bwd(hipStream_t, const void*, int, const elem_t*, const elem_t*, elem_t*, elem_t*, size_t, int, const float*, const float*, const float*, int,
              float*, float*, float*, float*, float, const float*, int);
int pw_maxpool3x3s2_fwd(hipStream_t, const elem_t*, elem_t*, unsigned char*, int, int, int, int);
int pw_maxpool3x3s2_bwd(hipStream_t, const elem_t*, const unsigned char*, elem_t*, int, int, int, int);
int pw_maxpool2x2_ceil(hipStream_t, const elem_t*, elem_t*, int, int, int, int);
int hm_sqdiff_rows(hipStream_t, const float*, const float*, const float*, const unsigned char*, int, int, float, float*, float*,
                   const unsigned char*, const float*, int);
int hm_sqdiff_bwd(hipStream_t, const float*, const float*, const float*, const unsigned char*, const float*, float, int, int, float*,
                  const unsigned char*, const float*, int);
int hm_mask_count(hipStream_t, const unsigned char*, size_t, float*);
int pw_maxpool2x2_ceil_f32(hipStream_t, const float*, float*, int, int, int, int);
int pw_nchw_f32_to_nhwc_f32(hipStream_t, const float*, float*, int, int, int, int);
int adain_launch_f32(hipStream_t, const float*, const float*, float*, int, int, int, int, float, float, const float*, float*);
int adain_launch_split(hipStream_t, const void*, const void*, void*, int, int, int, int, float, float, const float*, float*);
int pw_nchw_f32_to_nhwc_split(hipStream_t, const float*, void*, int, int, int, int);
int pw_f32_to_split(hipStream_t, const float*, void*, size_t);
int pw_split_to_f32(hipStream_t, const void*, float*, size_t);
int pw_maxpool2x2_ceil_split(hipStream_t, const void*, void*, int, int, int, int);
int hm_argmax_rectify(hipStream_t, const float*, int, int, int, float*, int*, float*, float*, const float*, int);
int hm_kth_mask(hipStream_t, const float*, const float*, int, int, float*, unsigned char*, const float*, int);
int hm_pck(hipStream_t, const float*, const float*, int, int, float, float, float, float*, float*);
int opt_chunk();
int opt_ema(hipStream_t, const long long*, const long long*, const long long*, const int*, const long long*, int, float, float);
int opt_adam(hipStream_t, const long long*, const long long*, const long long*, const long long*, const long long*, const int*, const long long*, int,
             float, float, float, float, float, int, float, float*);
int opt_sgd(hipStream_t, const long long*, const long long*, const long long*, const long long*, const int*, const long long*, int, float, float, float,
            int, int, float, float*);
int opt_grad_check(hipStream_t, const long long*, const long long*, const int*, const long long*, int, float*, long long);
int opt_scaler_update(hipStream_t, float*, float, float, int);
int comm_pack_bf16(hipStream_t, const float*, long long, void*, long long);
int comm_shard_mean(hipStream_t, const void*, int, long long, void*);
int comm_unpack_bf16(hipStream_t, const void*, float*, long long);
int adain_launch(hipStream_t, const elem_t*, const elem_t*, elem_t*, int, int, int, int, float, float, const float*, float*);
int aug_affine_u8(hipStream_t, const unsigned char*, unsigned char*, const long long*, int, int, int);
int aug_color_op(hipStream_t, unsigned char*, const int*, const float*, int*, int, int);
int aug_gaussian_blur_u8(hipStream_t, unsigned char*, unsigned char*, const unsigned int*, int, int, int);
int aug_resized_crop_u8(hipStream_t, const unsigned char*, unsigned char*, unsigned char*, const int*, const int*, const int*, int, int, int, int, int);
int aug_to_tensor(hipStream_t, const unsigned char*, float*, int, int, const float*, const float*);
int aug_gaussian_labels(hipStream_t, const double*, const float*, float*, float*, int, int, int, double, double, const float*, int);
int aug_draw_labelmap_ori(hipStream_t, const float*, const float*, const unsigned char*, float*, float*, int, int, int, float, const float*, int);
int affine_warp_chain(hipStream_t, const float*, float*, const float*, int, int, int, int, int, int);
int affine_recon_thetas(hipStream_t, const double*, int, double, float*, float*);
int affine_mean_views(hipStream_t, const float* const*, int, float*, size_t);
unsigned long long sp_sat_read_adain(int);
unsigned long long sp_sat_read_igemm(int);
unsigned long long sp_sat_read_patchconv(int);
unsigned long long sp_sat_read_pointwise(int);
int patch_paste(hipStream_t, float*, const int*, int, int, int, int, int);
int occlusion_pick(hipStream_t, const float*, const int*, const float*, int, int, int, double, int, float, float, int, int*, unsigned char*);
int select_rows(hipStream_t, float*, const float*, const float*, const unsigned char*, int, size_t);
int net_apply_running(void*, hipStream_t, const void*, void* const*, float);
int pw_axpy(hipStream_t, float*, const float*, size_t);
void prof_begin();
int prof_end(double*);
void* net_create(const int layers[4], int K, int N, int H, int W, int mode);
void net_destroy(void*);
void net_set_policy(void*, const Policy&);
const Policy& net_get_policy(void*);
int net_bind(void*, const void* const*, void* const*, void*);
int net_bind_grads(void*, void* const*);
long long net_grad_split_param(void*);
int net_bind_update(void*, void*, void* const*, void* const*, void* const*, void* const*, void* const*, void*, void*);
int net_fused_update(void*, void*, hipStream_t, void* const*, void* const*, void* const*, void* const*, void*, void*, float, float, float, float, float,
                     int, float, float*, float, float, int, long long);
int net_num_params(void*);
int net_num_buffers(void*);
long long net_param_numel(void*, int);
size_t net_wpack_bytes(void*);
size_t net_act_bytes(void*);
size_t net_ws_bytes(void*);
void net_out_shape(void*, int*);
int net_pack_weights(void*, hipStream_t, const void* const*, void*, int);
int net_forward(void*, hipStream_t, const float*, const void* const*, void* const*, const void*, void*, void*, float*, int, float);
int net_backward(void*, hipStream_t, const float*, const void* const*, const void*, void*, void*, void* const*, float, int, int);
int net_wgrad_pair(void*, hipStream_t, const void*, void*, void* const*, float, const void*, void*, void* const*, float, int);

static Policy from_c(const udapose_policy& c) {
    Policy p;
    p.igemm_tile = c.igemm_tile; p.igemm_h3 = c.igemm_h3; p.igemm_lean = c.igemm_lean; p.igemm_short_lds = c.igemm_short_lds;
    p.igemm_tap0 = c.igemm_tap0; p.wgrad_tile = c.wgrad_tile; p.wgrad_ksplit = c.wgrad_ksplit; p.wgrad_fastgeo = c.wgrad_fastgeo;
    p.wgrad_group = c.wgrad_group; p.wgrad_stages = c.wgrad_stages > 0 ? c.wgrad_stages : 128; p.wgrad_group_stem = c.wgrad_group_stem;
    p.bn_bwd_fused = c.bn_bwd_fused; p.bn_fwd_chunked = c.bn_fwd_chunked; p.bn_bwd_chunked = c.bn_bwd_chunked;
    p.bn_bwd_pre_legacy = c.bn_bwd_pre_legacy; p.igemm_wg_min = c.igemm_wg_min; p.wgrad_row3 = c.wgrad_row3; p.bn3_mask = c.bn3_mask; p.stem_fused = c.stem_fused; p.debug_sync = c.debug_sync; p.igemm_big_min = c.igemm_big_min; p.patch_conv = c.patch_conv; p.eval_fold = c.eval_fold; p.bn_xcd_rows = c.bn_xcd_rows; p.igemm_ns3_k = c.igemm_ns3_k; p.wgrad_det = c.wgrad_det; p.timeline = (unsigned long long*)c.timeline;
    return p;
}
static void to_c(const Policy& p, udapose_policy* c) {
    c->igemm_tile = p.igemm_tile; c->igemm_h3 = p.igemm_h3; c->igemm_lean = p.igemm_lean; c->igemm_short_lds = p.igemm_short_lds;
    c->igemm_tap0 = p.igemm_tap0; c->wgrad_tile = p.wgrad_tile; c->wgrad_ksplit = p.wgrad_ksplit; c->wgrad_fastgeo = p.wgrad_fastgeo;
    c->wgrad_group = p.wgrad_group; c->wgrad_stages = p.wgrad_stages; c->wgrad_group_stem = p.wgrad_group_stem;
    c->bn_bwd_fused = p.bn_bwd_fused; c->bn_fwd_chunked = p.bn_fwd_chunked; c->bn_bwd_chunked = p.bn_bwd_chunked;
    c->bn_bwd_pre_legacy = p.bn_bwd_pre_legacy; c->igemm_wg_min = p.igemm_wg_min; c->wgrad_row3 = p.wgrad_row3; c->bn3_mask = p.bn3_mask; c->stem_fused = p.stem_fused; c->debug_sync = p.debug_sync; c->igemm_big_min = p.igemm_big_min; c->patch_conv = p.patch_conv; c->eval_fold = p.eval_fold; c->bn_xcd_rows = p.bn_xcd_rows; c->igemm_ns3_k = p.igemm_ns3_k; c->wgrad_det = p.wgrad_det; c->timeline = p.timeline;
}
// a convolution descriptor and the policy it names, as the host-side geometry (the policy lives as long as this object)
struct Geom {
    Policy pol;
    ConvGeom g;
    explicit Geom(const udapose_conv_desc* d)
        : g{d->N, d->Hi, d->Wi, d->Ci, d->Co, d->KH, d->KW, d->stride, d->pad, d->transposed, d->reflect, d->upsample} {
        if (d->policy) { pol = from_c(*d->policy); g.pol = &pol; }
    }
};
#define S(x) ((hipStream_t)(x))
#define B16(x) ((elem_t*)(x))
#define CB16(x) ((const elem_t*)(x))

extern "C" {

int udapose_version(void) { return 200; }
int udapose_elem_kind(void) { return UDAPOSE_ELEM_KIND; }      // 0: this build stores / multiplies bf16, 1: fp16

void udapose_policy_default(udapose_policy* p) { if (p) to_c(default_policy(), p); }
void udapose_conv_out_hw(const udapose_conv_desc* d, int* Ho, int* Wo) { Geom G(d); *Ho = G.g.Ho(); *Wo = G.g.Wo(); }
int udapose_conv_stat_rows(const udapose_conv_desc* d) { Geom G(d); return conv_stat_rows(G.g); }
int udapose_conv_prepare(const udapose_conv_desc* d) { if (!d) return UDAPOSE_ERR_ARG; Geom G(d); return conv_prepare(G.g); }
int udapose_conv2d_fwd(void* stream, const udapose_conv_desc* d, const void* x, const void* w_fwd, void* y, const void* res, const float* bias,
                       float* stats, int flags) {
    if (!d || !x || !w_fwd || !y) return UDAPOSE_ERR_ARG;
    ConvEpilogue e;
    e.res = CB16(res); e.bias = bias; e.stats = stats; e.relu = (flags & UDAPOSE_EPI_RELU) != 0; e.out_f32 = (flags & UDAPOSE_EPI_OUT_F32) != 0; e.f32 = (flags & UDAPOSE_EPI_F32) != 0;
    e.split = (flags & UDAPOSE_EPI_SPLIT) != 0;
    if (e.split && e.f32) return UDAPOSE_ERR_ARG;
    Geom G(d);
    return conv_fprop(S(stream), G.g, CB16(x), CB16(w_fwd), y, e);
}
int udapose_conv2d_bwd_data(void* stream, const udapose_conv_desc* d, const void* dy, const void* w_bwd, void* dx, const void* res, int out_f32) {
    if (!d || !dy || !w_bwd || !dx) return UDAPOSE_ERR_ARG;
    Geom G(d);
    return conv_dgrad(S(stream), G.g, CB16(dy), CB16(w_bwd), dx, CB16(res), out_f32);
}
int udapose_conv_bwd_stat_rows(const udapose_conv_desc* d) { if (!d) return UDAPOSE_ERR_ARG; Geom G(d); return conv_dgrad_stat_rows(G.g); }
int udapose_conv2d_bwd_data_bn(void* stream, const udapose_conv_desc* d, const void* dy, const void* w_bwd, void* dx, const void* res, int out_f32,
                               const void* bn_y, const void* bn_z, const float* bn_mean, const float* bn_invstd, const float* bn_gamma,
                               const float* bn_beta, float* slab) {
    if (!d || !dy || !w_bwd || !dx || !bn_y || !bn_mean || !bn_invstd || !slab) return UDAPOSE_ERR_ARG;
    DgradBnStat st;
    st.y = CB16(bn_y); st.z = CB16(bn_z); st.mean = bn_mean; st.invstd = bn_invstd; st.gamma = bn_gamma; st.beta = bn_beta; st.slab = slab;
    Geom G(d);
    return conv_dgrad(S(stream), G.g, CB16(dy), CB16(w_bwd), dx, CB16(res), out_f32, &st);
}
int udapose_conv2d_bwd_weight(void* stream, const udapose_conv_desc* d, const void* dy, const void* x, float* dw, int accumulate) {
    if (!d || !dy || !x || !dw) return UDAPOSE_ERR_ARG;
    Geom G(d);
    return conv_wgrad(S(stream), G.g, CB16(dy), CB16(x), dw, accumulate, -1);
}
int udapose_cast_f32_bf16(void* stream, const float* src, void* dst, size_t n) { return pw_cast_f32_bf16(S(stream), src, B16(dst), n); }
int udapose_transpose_cast(void* stream, const float* src, void* dst, int A, int T, int B) { return pw_transpose_cast(S(stream), src, B16(dst), A, T, B); }
int udapose_pack_strided(void* stream, const float* src, void* dst, int A, int KH, int KWp, int KW, int Bp, int B, long sa, long skh, long skw, long sb) {
    return pw_pack_strided(S(stream), src, B16(dst), A, KH, KWp, KW, Bp, B, sa, skh, skw, sb);
}
int udapose_nchw_f32_to_nhwc_bf16(void* stream, const float* src, void* dst, int N, int C, int HW, int Cpad) {
    return pw_nchw_f32_to_nhwc_bf16(S(stream), src, B16(dst), N, C, HW, Cpad);
}
int udapose_nhwc_to_nchw_f32(void* stream, const void* src, int src_is_f32, float* dst, int N, int C, int HW, int Cs, const float* lo, const float* hi) {
    return pw_nhwc_to_nchw_f32(S(stream), src, src_is_f32, dst, N, C, HW, Cs, lo, hi);
}
int udapose_bn_finalize(void* stream, const float* stats, int rows, int C, double count, const float* gamma, const float* beta, float* rm, float* rv,
                        long long* nbt, float momentum, float eps, float* scale, float* shift, float* save_mean, float* save_invstd) {
    return pw_bn_finalize(S(stream), stats, rows, C, count, gamma, beta, rm, rv, nbt, momentum, eps, scale, shift, save_mean, save_invstd, nullptr);
}
int udapose_bn_eval_coeff(void* stream, int C, const float* gamma, const float* beta, const float* rm, const float* rv, float eps, float* scale,
                          float* shift) {
    return pw_bn_eval_coeff(S(stream), C, gamma, beta, rm, rv, eps, scale, shift);
}
int udapose_bn_apply(void* stream, const void* y, const void* res, void* z, size_t numel, int C, const float* scale, const float* shift, int relu) {
    return pw_bn_apply(S(stream), CB16(y), CB16(res), B16(z), numel, C, scale, shift, relu, nullptr, default_policy().bn_xcd_rows >= 2);
}
int udapose_bn_bwd_rows(size_t npix) { return pw_bn_bwd_rows(npix); }
int udapose_bn_bwd(void* stream, const void* dz, int dz_is_f32, const void* z, const void* y, void* dy, void* gout, size_t npix, int C,
                   const float* gamma, const float* mean, const float* invstd, int relu, float* slab, float* coef, float* dgamma, float* dbeta,
                   float beta_acc, const float* beta) {
    return pw_bn_bwd(S(stream), dz, dz_is_f32, CB16(z), CB16(y), B16(dy), B16(gout), npix, C, gamma, mean, invstd, relu, slab, coef, dgamma, dbeta,
                     beta_acc, beta, default_policy().bn_bwd_chunked);
}
int udapose_bn_bwd_pre(void* stream, const void* g, int g_is_f32, const void* y, void* dy, size_t npix, int C, const float* gamma, const float* mean,
                       const float* invstd, const float* slab, int rows, float* coef, float* dgamma, float* dbeta, float beta_acc) {
    if (!g || !y || !dy || !gamma || !mean || !invstd || !slab || !coef) return UDAPOSE_ERR_ARG;
    return pw_bn_bwd_pre(S(stream), g, g_is_f32, CB16(y), B16(dy), npix, C, gamma, mean, invstd, slab, rows, coef, dgamma, dbeta, beta_acc,
                         default_policy().bn_bwd_chunked, default_policy().bn_bwd_pre_legacy);
}
int udapose_maxpool3x3s2_fwd(void* stream, const void* x, void* y, unsigned char* idx, int N, int H, int W, int C) {
    return pw_maxpool3x3s2_fwd(S(stream), CB16(x), B16(y), idx, N, H, W, C);
}
int udapose_maxpool3x3s2_bwd(void* stream, const void* dy, const unsigned char* idx, void* dx, int N, int H, int W, int C) {
    return pw_maxpool3x3s2_bwd(S(stream), CB16(dy), idx, B16(dx), N, H, W, C);
}
int udapose_maxpool2x2_ceil(void* stream, const void* x, void* y, int N, int H, int W, int C) {
    return pw_maxpool2x2_ceil(S(stream), CB16(x), B16(y), N, H, W, C);
}
int udapose_maxpool2x2_ceil_f32(void* stream, const float* x, float* y, int N, int H, int W, int C) {
    return pw_maxpool2x2_ceil_f32(S(stream), x, y, N, H, W, C);
}
int udapose_nchw_f32_to_nhwc_f32(void* stream, const float* src, float* dst, int N, int C, int HW, int Cpad) {
    return pw_nchw_f32_to_nhwc_f32(S(stream), src, dst, N, C, HW, Cpad);
}
int udapose_nchw_f32_to_nhwc_split(void* stream, const float* src, void* dst, int N, int C, int HW, int Cpad) {
    if (!src || !dst) return UDAPOSE_ERR_ARG;
    return pw_nchw_f32_to_nhwc_split(S(stream), src, dst, N, C, HW, Cpad);
}
int udapose_f32_to_split(void* stream, const float* src, void* dst, size_t n) {
    if (!src || !dst) return UDAPOSE_ERR_ARG;
    return pw_f32_to_split(S(stream), src, dst, n);
}
int udapose_split_to_f32(void* stream, const void* src, float* dst, size_t n) {
    if (!src || !dst) return UDAPOSE_ERR_ARG;
    return pw_split_to_f32(S(stream), src, dst, n);
}
int udapose_maxpool2x2_ceil_split(void* stream, const void* x, void* y, int N, int H, int W, int C) {
    if (!x || !y) return UDAPOSE_ERR_ARG;
    return pw_maxpool2x2_ceil_split(S(stream), x, y, N, H, W, C);
}

int udapose_net_create(const int layers[4], int K, int N, int H, int W, int fp32, udapose_net_t* out) {
    if (!out) return UDAPOSE_ERR_ARG;
    *out = net_create(layers, K, N, H, W, fp32);
    return *out ? UDAPOSE_OK : UDAPOSE_ERR_ARG;
}
void udapose_net_destroy(udapose_net_t n) { net_destroy(n); }
int udapose_net_set_policy(udapose_net_t n, const udapose_policy* p) {
    if (!n || !p) return UDAPOSE_ERR_ARG;
    net_set_policy(n, from_c(*p));
    return UDAPOSE_OK;
}
int udapose_net_get_policy(udapose_net_t n, udapose_policy* p) {
    if (!n || !p) return UDAPOSE_ERR_ARG;
    to_c(net_get_policy(n), p);
    return UDAPOSE_OK;
}
int udapose_net_bind(udapose_net_t n, const void* const* params, void* const* buffers, void* wpack) {
    if (!n || !params || !wpack) return UDAPOSE_ERR_ARG;
    return net_bind(n, params, buffers, wpack);
}
int udapose_net_bind_grads(udapose_net_t n, void* const* grads) {
    if (!n || !grads) return UDAPOSE_ERR_ARG;
    return net_bind_grads(n, grads);
}
int udapose_net_num_params(udapose_net_t n) { return net_num_params(n); }
int udapose_net_num_buffers(udapose_net_t n) { return net_num_buffers(n); }
long long udapose_net_param_numel(udapose_net_t n, int i) { return net_param_numel(n, i); }
size_t udapose_net_wpack_bytes(udapose_net_t n) { return net_wpack_bytes(n); }
size_t udapose_net_act_bytes(udapose_net_t n) { return net_act_bytes(n); }
size_t udapose_net_ws_bytes(udapose_net_t n) { return net_ws_bytes(n); }
void udapose_net_out_shape(udapose_net_t n, int shape[4]) { net_out_shape(n, shape); }
int udapose_net_pack_weights(udapose_net_t n, void* stream, const void* const* params, void* wpack, int with_bwd) {
    return net_pack_weights(n, S(stream), params, wpack, with_bwd);
}
int udapose_net_forward(udapose_net_t n, void* stream, const float* x, const void* const* params, void* const* buffers, const void* wpack, void* act,
                        void* ws, float* out, int training, float momentum) {
    return net_forward(n, S(stream), x, params, buffers, wpack, act, ws, out, training, momentum);
}
int udapose_net_apply_running(udapose_net_t n, void* stream, const void* act, void* const* buffers, float momentum) {
    return net_apply_running(n, S(stream), act, buffers, momentum);
}
int udapose_axpy_f32(void* stream, float* y, const float* x, size_t n) { return pw_axpy(S(stream), y, x, n); }
int udapose_net_backward(udapose_net_t n, void* stream, const float* dout, const void* const* params, const void* wpack, void* act, void* ws,
                         void* const* grads, float beta) {
    return net_backward(n, S(stream), dout, params, wpack, act, ws, grads, beta, 0, 0);
}
int udapose_net_backward_part(udapose_net_t n, void* stream, const float* dout, const void* const* params, const void* wpack, void* act,
                              void* ws, void* const* grads, float beta, int part) {
    if (part != 1 && part != 2) return UDAPOSE_ERR_ARG;
    return net_backward(n, S(stream), dout, params, wpack, act, ws, grads, beta, part, 0);
}
int udapose_net_wgrad_pair(udapose_net_t n, void* stream, const void* act_a, void* ws_a, void* const* grads_a, float beta_a, const void* act_b,
                           void* ws_b, void* const* grads_b, float beta_b, int part) {
    if (!n || !act_a || !ws_a || !grads_a || !act_b || !ws_b || !grads_b) return UDAPOSE_ERR_ARG;
    return net_wgrad_pair(n, S(stream), act_a, ws_a, grads_a, beta_a, act_b, ws_b, grads_b, beta_b, part);
}
int udapose_net_backward_phase(udapose_net_t n, void* stream, const float* dout, const void* const* params, const void* wpack, void* act,
                               void* ws, void* const* grads, float beta, int part, int phase) {
    if (!n) return UDAPOSE_ERR_ARG;
    return net_backward(n, S(stream), dout, params, wpack, act, ws, grads, beta, part, phase);
}
long long udapose_net_grad_split_param(udapose_net_t n) { return net_grad_split_param(n); }
int udapose_net_bind_update(udapose_net_t student, udapose_net_t teacher, void* const* params_s, void* const* grads, void* const* exp_avg,
                            void* const* exp_avg_sq, void* const* params_t, void* wpack_s, void* wpack_t) {
    if (!student || !teacher || !params_s || !grads || !exp_avg || !exp_avg_sq || !params_t || !wpack_s || !wpack_t) return UDAPOSE_ERR_ARG;
    return net_bind_update(student, teacher, params_s, grads, exp_avg, exp_avg_sq, params_t, wpack_s, wpack_t);
}
int udapose_net_fused_update(udapose_net_t student, udapose_net_t teacher, void* stream, void* const* params_s, void* const* grads,
                             void* const* exp_avg, void* const* params_t, void* wpack_s, void* wpack_t, float lr, float beta1, float beta2,
                             float eps, float weight_decay, int step, float grad_scale, float* dev_state, float alpha, float one_minus_alpha,
                             int do_adam, long long grad2_delta_bytes) {
    if (!student || !teacher) return UDAPOSE_ERR_ARG;
    return net_fused_update(student, teacher, S(stream), params_s, grads, exp_avg, params_t, wpack_s, wpack_t, lr, beta1, beta2, eps, weight_decay,
                            step, grad_scale, dev_state, alpha, one_minus_alpha, do_adam, grad2_delta_bytes);
}

int udapose_joints_mse_fwd(void* stream, const float* pred, const float* gt, const float* w, int R, int HW, float* rows, float* mean_out) {
    return hm_sqdiff_rows(S(stream), pred, gt, w, nullptr, R, HW, 0.5f, rows, mean_out, nullptr, nullptr, 1);
}
int udapose_joints_mse_bwd(void* stream, const float* pred, const float* gt, const float* w, const float* gscale, int R, int HW, float* dpred) {
    return hm_sqdiff_bwd(S(stream), pred, gt, w, nullptr, gscale, (float)(1.0 / ((double)R * HW)), R, HW, dpred, nullptr, nullptr, 1);
}
int udapose_cons_loss_fwd(void* stream, const float* stu, const float* tea, const unsigned char* mask, int R, int HW, float* rows, float* mean_out) {
    return hm_sqdiff_rows(S(stream), stu, tea, nullptr, mask, R, HW, 1.0f, rows, mean_out, nullptr, nullptr, 1);
}
int udapose_cons_loss_bwd(void* stream, const float* stu, const float* tea, const unsigned char* mask, const float* gscale, int R, int HW, float* dstu) {
    return hm_sqdiff_bwd(S(stream), stu, tea, nullptr, mask, gscale, (float)(2.0 / ((double)R * HW)), R, HW, dstu, nullptr, nullptr, 1);
}
int udapose_mask_count(void* stream, const unsigned char* mask, size_t n, float* count) {
    if (!mask || !count) return UDAPOSE_ERR_ARG;
    return hm_mask_count(S(stream), mask, n, count);
}
int udapose_cons_loss_valid_fwd(void* stream, const float* stu, const float* tea, const unsigned char* mask, const unsigned char* valid,
                                const float* valid_count, int R, int K, int HW, float* rows, float* mean_out) {
    if (!valid || !valid_count) return UDAPOSE_ERR_ARG;
    return hm_sqdiff_rows(S(stream), stu, tea, nullptr, mask, R, HW, 1.0f, rows, mean_out, valid, valid_count, K);
}
int udapose_cons_loss_valid_bwd(void* stream, const float* stu, const float* tea, const unsigned char* mask, const unsigned char* valid,
                                const float* valid_count, const float* gscale, int R, int K, int HW, float* dstu) {
    if (!valid || !valid_count) return UDAPOSE_ERR_ARG;
    return hm_sqdiff_bwd(S(stream), stu, tea, nullptr, mask, gscale, (float)(2.0 / ((double)R * HW)), R, HW, dstu, valid, valid_count, K);
}
int udapose_heatmap_argmax(void* stream, const float* hm, int R, int H, int W, float* maxvals, int* flat_idx, float* preds, float* rect,
                           const float* patch, int rad) {
    if (rect && !patch) return UDAPOSE_ERR_ARG;
    return hm_argmax_rectify(S(stream), hm, R, H, W, maxvals, flat_idx, preds, rect, patch, rad);
}
int udapose_kth_mask(void* stream, const float* act, const float* tm, int n, int k, float* thr_out, unsigned char* mask, const float* act_local,
                     int n_local) {
    return hm_kth_mask(S(stream), act, tm, n, k, thr_out, mask, act_local, n_local);
}
int udapose_pck(void* stream, const float* pred, const float* gt, int B, int K, float nx, float ny, float thr, float* acc, float* avg_cnt) {
    return hm_pck(S(stream), pred, gt, B, K, nx, ny, thr, acc, avg_cnt);
}
int udapose_multi_chunk(void) { return opt_chunk(); }
int udapose_ema_multi(void* stream, const long long* t, const long long* s, const long long* sizes, const int* bt, const long long* bo, int nb,
                      float alpha, float oma) {
    return opt_ema(S(stream), t, s, sizes, bt, bo, nb, alpha, oma);
}
int udapose_adam_multi(void* stream, const long long* p, const long long* g, const long long* m, const long long* v, const long long* sizes,
                       const int* bt, const long long* bo, int nb, float lr, float b1, float b2, float eps, float wd, int step, float gscale,
                       float* dev_state) {
    return opt_adam(S(stream), p, g, m, v, sizes, bt, bo, nb, lr, b1, b2, eps, wd, step, gscale, dev_state);
}
int udapose_sgd_multi(void* stream, const long long* p, const long long* g, const long long* buf, const long long* sizes, const int* bt,
                      const long long* bo, int nb, float lr, float mom, float wd, int nesterov, int first, float gscale, float* dev_state) {
    return opt_sgd(S(stream), p, g, buf, sizes, bt, bo, nb, lr, mom, wd, nesterov, first, gscale, dev_state);
}
int udapose_grad_scaler_check(void* stream, const long long* g, const long long* sizes, const int* bt, const long long* bo, int nb, float* dev_state) {
    return opt_grad_check(S(stream), g, sizes, bt, bo, nb, dev_state, 0);
}
int udapose_grad_scaler_check2(void* stream, const long long* g, const long long* sizes, const int* bt, const long long* bo, int nb, float* dev_state,
                               long long grad2_delta_bytes) {
    return opt_grad_check(S(stream), g, sizes, bt, bo, nb, dev_state, grad2_delta_bytes);
}
int udapose_grad_scaler_update(void* stream, float* dev_state, float growth, float backoff, int interval) {
    return opt_scaler_update(S(stream), dev_state, growth, backoff, interval);
}
int udapose_comm_pack_bf16(void* stream, const float* src, long long n, void* dst_bf16, long long n_padded) {
    if (!src || !dst_bf16) return UDAPOSE_ERR_ARG;
    return comm_pack_bf16(S(stream), src, n, dst_bf16, n_padded);
}
int udapose_comm_shard_mean(void* stream, const void* shards_bf16, int world, long long m, void* out_bf16) {
    if (!shards_bf16 || !out_bf16) return UDAPOSE_ERR_ARG;
    return comm_shard_mean(S(stream), shards_bf16, world, m, out_bf16);
}
int udapose_comm_unpack_bf16(void* stream, const void* src_bf16, float* dst, long long n) {
    if (!src_bf16 || !dst) return UDAPOSE_ERR_ARG;
    return comm_unpack_bf16(S(stream), src_bf16, dst, n);
}
int udapose_adain(void* stream, const void* c, const void* s, void* out, int N, int HWc, int HWs, int C, float eps, float alpha, float* stats_out) {
    return adain_launch(S(stream), CB16(c), CB16(s), B16(out), N, HWc, HWs, C, eps, alpha, nullptr, stats_out);
}
int udapose_adain_alpha_dev(void* stream, const void* c, const void* s, void* out, int N, int HWc, int HWs, int C, float eps, const float* alpha_dev,
                            float* stats_out, int is_f32) {
    if (!alpha_dev) return UDAPOSE_ERR_ARG;
    if (is_f32) return adain_launch_f32(S(stream), (const float*)c, (const float*)s, (float*)out, N, HWc, HWs, C, eps, 1.f, alpha_dev, stats_out);
    return adain_launch(S(stream), CB16(c), CB16(s), B16(out), N, HWc, HWs, C, eps, 1.f, alpha_dev, stats_out);
}
int udapose_adain_f32(void* stream, const float* c, const float* s, float* out, int N, int HWc, int HWs, int C, float eps, float alpha,
                      float* stats_out) {
    return adain_launch_f32(S(stream), c, s, out, N, HWc, HWs, C, eps, alpha, nullptr, stats_out);
}
int udapose_adain_split(void* stream, const void* c, const void* s, void* out, int N, int HWc, int HWs, int C, float eps, float alpha,
                        const float* alpha_dev, float* stats_out) {
    if (!c || !s) return UDAPOSE_ERR_ARG;
    return adain_launch_split(S(stream), c, s, out, N, HWc, HWs, C, eps, alpha, alpha_dev, stats_out);
}

int udapose_patch_paste(void* stream, float* img, const int* boxes, int n, int C, int H, int W, int max_patch_elems) {
    return patch_paste(S(stream), img, boxes, n, C, H, W, max_patch_elems);
}
int udapose_occlusion_pick(void* stream, const float* conf, const int* flat_idx, const float* u, int N, int K, int w, double ratio, int image_size,
                           float rate, float thresh, int occlude_size, int* boxes, unsigned char* apply) {
    if (!conf || !flat_idx || !u || !boxes || !apply) return UDAPOSE_ERR_ARG;
    return occlusion_pick(S(stream), conf, flat_idx, u, N, K, w, ratio, image_size, rate, thresh, occlude_size, boxes, apply);
}
int udapose_select_rows(void* stream, float* dst, const float* a, const float* b, const unsigned char* flag, int N, size_t row_elems) {
    if (!dst || !a || !b || !flag) return UDAPOSE_ERR_ARG;
    return select_rows(S(stream), dst, a, b, flag, N, row_elems);
}
int udapose_recon_thetas(void* stream, const double* params, int N, double ratio, float* theta_fwd, float* theta_back) {
    if (!params || (!theta_fwd && !theta_back)) return UDAPOSE_ERR_ARG;
    return affine_recon_thetas(S(stream), params, N, ratio, theta_fwd, theta_back);
}
int udapose_split_saturations(int reset, unsigned long long* count) {
    if (!count) return UDAPOSE_ERR_ARG;
    unsigned long long t = 0;
    for (unsigned long long v : {sp_sat_read_adain(reset), sp_sat_read_igemm(reset), sp_sat_read_patchconv(reset), sp_sat_read_pointwise(reset)}) {
        if (v == ~0ull) return UDAPOSE_ERR_LAUNCH;
        t += v;
    }
    *count = t;
    return UDAPOSE_OK;
}
int udapose_mean_views(void* stream, const float* const* h_views, int k, float* dst, size_t n) {
    return affine_mean_views(S(stream), h_views, k, dst, n);
}
int udapose_affine_nearest(void* stream, const float* src, float* dst, const float* theta, int N, int C, int H, int W, int nstage, int backward) {
    return affine_warp_chain(S(stream), src, dst, theta, N, C, H, W, nstage, backward);
}

int udapose_aug_affine_u8(void* stream, const unsigned char* src, unsigned char* dst, const long long* coef, int N, int H, int W) {
    if (!src || !dst || !coef) return UDAPOSE_ERR_ARG;
    return aug_affine_u8(S(stream), src, dst, coef, N, H, W);
}
int udapose_aug_color_op(void* stream, unsigned char* img, const int* op, const float* factor, int* mean_scratch, int N, int HW) {
    if (!img || !op || !factor || !mean_scratch) return UDAPOSE_ERR_ARG;
    return aug_color_op(S(stream), img, op, factor, mean_scratch, N, HW);
}
int udapose_aug_gaussian_blur_u8(void* stream, unsigned char* img, unsigned char* tmp, const unsigned int* prm, int N, int H, int W) {
    if (!img || !tmp || !prm) return UDAPOSE_ERR_ARG;
    return aug_gaussian_blur_u8(S(stream), img, tmp, prm, N, H, W);
}
int udapose_aug_resized_crop_u8(void* stream, const unsigned char* src, unsigned char* dst, unsigned char* tmp, const int* box, const int* bounds,
                                const int* coef, int N, int Hs, int Ws, int S, int ksize) {
    if (!src || !dst || !tmp || !box || !bounds || !coef) return UDAPOSE_ERR_ARG;
    return aug_resized_crop_u8(S(stream), src, dst, tmp, box, bounds, coef, N, Hs, Ws, S, ksize);
}
int udapose_aug_to_tensor(void* stream, const unsigned char* img, float* out, int N, int HW, const float* mean3, const float* std3) {
    if (!img || !out || !mean3 || !std3) return UDAPOSE_ERR_ARG;
    return aug_to_tensor(S(stream), img, out, N, HW, mean3, std3);
}
int udapose_gaussian_labels(void* stream, const double* kp, const float* vis, float* target, float* weight, int R, int Hh, int Wh,
                            double stride_x, double stride_y, const float* patch, int rad) {
    if (!kp || !vis || !target || !weight) return UDAPOSE_ERR_ARG;
    return aug_gaussian_labels(S(stream), kp, vis, target, weight, R, Hh, Wh, stride_x, stride_y, patch, rad);
}
int udapose_draw_labelmap_ori(void* stream, const float* pt, const float* vis, const unsigned char* gate, float* target, float* weight, int R,
                              int Hh, int Wh, float r3, const float* patch, int psize) {
    if (!target || !weight) return UDAPOSE_ERR_ARG;
    return aug_draw_labelmap_ori(S(stream), pt, vis, gate, target, weight, R, Hh, Wh, r3, patch, psize);
}
void udapose_prof_begin(void) { prof_begin(); }
int udapose_prof_end(double* h_out9) { return prof_end(h_out9); }

}  // extern "C"
