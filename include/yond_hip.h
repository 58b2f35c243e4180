/* yond_hip.h -- C ABI of libyond_hip.so: the MI355X (gfx950) kernels behind YOND's per-image hot path.
 *
 * The reference (fenghansen/YOND_public) has no FFI: its hot path is Python over NumPy / cv2 / SciPy /
 * torch.nn.  The boundary a maintainer binds is therefore this C ABI, called through ctypes from the
 * Python functions that keep the reference's names.  Every entry point
 *   - takes plain device pointers (from torch.Tensor.data_ptr()), sizes and a hipStream_t (as void*),
 *   - is asynchronous on that stream, allocates nothing, keeps no state and owns none of its arguments,
 *   - returns 0 on success, a negative YOND_E* code for a rejected argument (nothing launched) or the
 *     positive hipError_t of a failed launch.
 * All images are float32.  "packed" means the 4-channel half-resolution view of a Bayer frame,
 * channel = 2*dy+dx  (utils/isp_ops.py:57-63).  Each function cites the reference lines it replaces
 * (paths relative to the reference root).
 */
#ifndef YOND_HIP_H
#define YOND_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YOND_OK 0
#define YOND_EINVAL (-1)      /* bad pointer / size / flag */
#define YOND_EUNSUPPORTED (-2) /* valid request the kernels do not cover (e.g. channel count) */

/* Library / device probe.  Returns the ABI version (this header: YOND_ABI_VERSION; the loader refuses a mismatch). */
#define YOND_ABI_VERSION 8
int yond_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * K1  pack + VST + bias + normalise + reflect-pad + clamp            (memory bound, 8 B / Bayer px)
 * Replaces YOND_SIDD.py:251-269 (pack, *scale, bias LUT, VST, normalise), :281-282 (reflect pad to a
 * multiple of 32), :286 (clamp to [0,1]), utils/isp_algos.py:5-14 (VST) and the per-pixel evaluation
 * of the interp1d object returned by utils/isp_algos.py:128.  Also produces the per-image maximum that
 * archs/modules.py:15-21 (data_normalize) needs.
 *   bayer      [H][W]                      input frame (device)
 *   out        [Hp][Wp][4]                 Hp = h+pad_t+pad_b, Wp = w+pad_l+pad_r, h=H/2, w=W/2
 *   mode       0: identity (Simple_Denoiser, YOND_SIDD.py:238-243: pack, pad, clamp only)
 *              1: u = (VST(x*scale) - bias(max(x*scale,0)) - lo) / (hi - lo)
 *   lut_x/lut_y  knots of the bias LUT (float64 abscissae, float32 ordinates, device); lut_n = 0
 *              disables the bias correction (bias_corr=None).
 *   img_max    optional device float[1]: max over the clamped output (must be zeroed by the caller
 *              -- the function issues the memset itself on `stream`).
 * Arithmetic: float32 x*scale, then float64 in NumPy's staging with three shortcuts that stay below 1e-12 relative --
 * sqrt as a float32 estimate + one float64 Newton step, the LUT as per-interval coefficients a + b x, (v - lo) times the
 * reciprocal span -- and ONE rounding to float32: the result differs from the staged float64 evaluation by at most one
 * float32 ulp, and only where that evaluation lies within 1e-12 of a rounding boundary. */
int yond_pack_vst_norm_f32(const float* bayer, int H, int W, float* out, int pad_l, int pad_r, int pad_t,
                           int pad_b, int mode, double scale, double gain, double sigma, double lo,
                           double hi, const double* lut_x, const float* lut_y, int lut_n, float* img_max,
                           void* stream);

/* K1 with the 2-D bias LUT (YOND_SIDD.py:258-259 -> utils/isp_algos.py:162-231 BiasLUT.get_lut): lut_x = the table's x
 * knots in DN (x_lut * gain, float64), lut_y = the table row merged for sigma / gain (float64; the host interpolates the
 * two neighbouring sigma rows, :188-194).  Beyond the last knot: the last ordinate for one more interval, then
 * get_bias_points' closed form (:226-230).  Otherwise as yond_pack_vst_norm_f32 with mode 1. */
int yond_pack_vst_norm_biaslut_f32(const float* bayer, int H, int W, float* out, int pad_l, int pad_r, int pad_t,
                                   int pad_b, double scale, double gain, double sigma, double lo, double hi,
                                   const double* lut_x, const double* lut_y, int lut_n, float* img_max, void* stream);

/* The bias LUT alone on a flat float32 array (the callable get_bias returns, utils/isp_algos.py:128, or
 * BiasLUT.get_lut(x), :196-231): y_is_f64 selects float64 ordinates, biaslut the 2-D LUT's behaviour beyond the knots.
 * out: float64[n]. */
int yond_bias_eval_f32(const float* x, size_t n, const double* lut_x, const void* lut_y, int lut_n, int y_is_f64, int biaslut,
                       double gain, double sigma, double* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K4  clamp + crop + de-normalise + inverse VST + unpack (+ /scale, clip)          (8 B / Bayer px)
 * Replaces YOND_SIDD.py:286 (output clamp), :289-299 and utils/isp_algos.py:17-33 (inverse_VST).
 *   net_out  [Hp][Wp][4];  bayer_out [2h][2w];  crop origin (pad_t, pad_l)
 *   mode 0: identity (Simple_Denoiser :244-248), 1: algebraic inverse, 2: closed-form exact inverse
 *   clip01: apply the caller's .clip(0,1) (YOND_SIDD.py:389,407). */
int yond_denorm_ivst_unpack_f32(const float* net_out, int Hp, int Wp, int pad_t, int pad_l, int h, int w,
                                float* bayer_out, int mode, double scale, double gain, double sigma,
                                double lo, double hi, int clip01, void* stream);

/* K1 / K4 for B equally sized frames that share every constant and the bias LUT -- the 32 blocks of a SIDD image, which the
 * reference sends through VST_Denoiser one by one with the same p and bias_func (YOND_SIDD.py:392-407): ONE launch each.
 *   bayer [B][H][W] -> out [B][Hp][Wp][4], img_max [B];   net_out [B][Hp][Wp][4] -> bayer_out [B][2h][2w].
 *   biaslut != 0: lut_y is float64, the merged row of the 2-D table (as yond_pack_vst_norm_biaslut_f32). */
int yond_pack_vst_norm_batch_f32(const float* bayer, int B, int H, int W, float* out, int pad_l, int pad_r, int pad_t, int pad_b,
                                 double scale, double gain, double sigma, double lo, double hi, const double* lut_x,
                                 const void* lut_y, int lut_n, int biaslut, float* img_max, void* stream);
int yond_denorm_ivst_unpack_batch_f32(const float* net_out, int B, int Hp, int Wp, int pad_t, int pad_l, int h, int w,
                                      float* bayer_out, int mode, double scale, double gain, double sigma,
                                      double lo, double hi, int clip01, void* stream);

/* Stand-alone elementwise VST / inverse VST on flat arrays (utils/isp_algos.py:5-14, 17-33) for the function
 * seam; the hot path uses the fused K1 / K4 above.  float32 -> float64 and float64 -> float64 as NumPy stages
 * them with np.float64 noise parameters. */
int yond_vst_elem_f32(const float* x, size_t n, double sigma, double mu, double gain, double* out, void* stream);
int yond_ivst_elem_f64(const double* z, size_t n, double sigma, double gain, int exact, double* out, void* stream);

/* Bayer <-> packed planar/NHWC4 copies (utils/isp_ops.py:57-63), bit exact. */
int yond_bayer2rggb_f32(const float* bayer, int H, int W, float* rggb /*[H/2][W/2][4]*/, void* stream);
int yond_rggb2bayer_f32(const float* rggb, int h, int w, float* bayer /*[2h][2w]*/, void* stream);

/* np.rot90(x, k, axes=(-2, -1)) on a stack [N][H][W] -> [N][H'][W'] (rot_bayer: utils/sidd_utils.py:198-213, used around
 * the denoiser when the runfile sets rot_cfa, YOND_SIDD.py:402-404, 462-464), bit exact. */
int yond_rot90_f32(const float* src, int N, int H, int W, int k, float* dst, void* stream);

/* Layout helpers for the archs plugin surface (NCHW tensors in, NCHW out; C == 4). */
int yond_nchw4_to_nhwc4_f32(const float* src, float* dst, int N, int H, int W, void* stream);
int yond_nhwc4_to_nchw4_f32(const float* src, float* dst, int N, int H, int W, void* stream);

/* Per-image maximum (archs/modules.py:18-19).  x: [N][elems]; partial: workspace float[N*256];
 * out: float[N].  Two launches, deterministic. */
int yond_image_max_f32(const float* x, int N, size_t elems, float* partial, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K2  convolutions of the denoiser as MFMA implicit GEMMs on NHWC float32 activations with float32 results.  `algo`
 * selects the arithmetic: 3 / 5 (the default path of engine.py) fp32-accurate split-operand products on the fp16 matrix
 * cores (v_mfma_f32_32x32x16_f16 / 32x32x8_f16, three per fp32 product block), 0 / 1 the fp32-input MFMA
 * (v_mfma_f32_32x32x2_f32 direct, v_mfma_f32_16x16x4_f32 Winograd), 2 / 4 fp16 operands (BASELINE cfg 5).
 * One descriptor covers: 3x3 stride 1 / stride 2 and 1x1 convolutions,
 * an input that is the channel concatenation of two tensors (torch.cat([up, skip], 1) is never
 * materialised), 2x2 stride-2 transposed convolution (as a 1x1 GEMM with a pixel-shuffle store), an
 * optional SiLU on the staged input, and the epilogue  v = acc*escale + eshift ; act(v) ; v += res.
 * Replaces nn.Conv2d / nn.ConvTranspose2d / SiLU / FiLM / residual at archs/modules.py:117-125,
 * 186-196, 221-233 and archs/Unet.py:55-104, 332-378, 424-470.
 * Weights must be packed by yond_pack_conv_weight_f32 (host side) for the same (taps, TN, KC). */
typedef struct YondConvDesc {
    const float* src0;    /* [N][H][W][C0] */
    const float* src1;    /* [N][H][W][C1] or NULL; with shuffle: the skip tensor at the OUTPUT resolution [N][2H][2W][C1],
                             read at the sub-position each channel block stores to (fused convT + cat + 1x1 shortcut) */
    int C0, C1;           /* Cin = C0 + C1; each a multiple of kc */
    int N, H, W;          /* input extent */
    int Ho, Wo;           /* GEMM-M extent: H/stride, W/stride (convT: H, W) */
    int Cout;             /* GEMM-N extent (convT: 4 * real Cout), multiple of 32 */
    int ksize;            /* 1 or 3 */
    int stride;           /* 1 or 2 (2 only with ksize 3) */
    int shuffle;          /* 1: convT 2x2 s2 store, dst is [N][2Ho][2Wo][Cout/4], GEMM column = sub-position * Cout/4 + channel;
                             2 (algo 3, split-plane inputs, Cout/4 = Cr a multiple of 32): the same with TWO sub-positions (dy, 0),
                             (dy, 1) per 64-wide channel tile -- GEMM columns ordered [dy][channel block of 32][dx][32]; the K range
                             of src1 is [Cr channels read at dx = 0 | the same channels read at dx = 1 | zero-weight channels up to
                             (C0 + C1) % 48 == 0], and the weight rows of a sub-position are zero in the other one's range:
                             src0 is staged once for both */
    int pre_act;          /* 0 none, 1 SiLU applied to the staged input */
    int post_act;         /* 0 none, 1 SiLU (algo 3 / 4, 3x3 only: the stored tensor is the consumer's SiLU input), 2 LeakyReLU(slope) */
    float slope;
    const float* wpk;     /* packed weights */
    const float* escale;  /* [ebatch ? N : 1][Cout'] or NULL (=1) ; Cout' = real Cout */
    const float* eshift;  /* [ebatch ? N : 1][Cout'] or NULL (=0) */
    int ebatch;
    const float* res;     /* residual, same shape as dst, or NULL */
    float* dst;           /* [N][Ho][Wo][Cout]  (shuffle: see above) */
    int tn;               /* channel-tile width the weights were packed for (32 or 64, from yond_conv_config) */
    int kc;               /* channel chunk the weights were packed for (8 or 16, from yond_conv_config; 0 = default) */
    int algo;             /* 0 direct implicit GEMM, fp32 MFMA; 1 Winograd F(2x2,3x3), fp32 MFMA (3x3 stride 1 only; wpk from
                             yond_pack_conv_wino_weight_f32, tn = 64, see yond_conv_wino_supported); 2 direct implicit GEMM
                             on the fp16 MFMA (operands rounded to half at the matrix core, fp32 accumulate, fp32 tensors:
                             BASELINE cfg 5; same packed weights as algo 0); 3 direct implicit GEMM on the fp16 MFMA with
                             fp32-accurate SPLIT operands (a = h + l 2^-11, three fp16 products per fp32 product, fp32
                             accumulate; 3x3 only, wpk from yond_pack_conv_split_weight_f32 with parts = 2, tn from
                             yond_conv_split_supported); 4 the same kernel with h only = plain fp16 MFMA (parts = 1); 5 the split-operand
                             arithmetic of algo 3 in the generic kernel of algo 0 for the 1x1 / transposed layers (wpk from
                             yond_pack_conv_weight_split_f32, tn / kc from yond_conv_config) */
    /* Fused output projection (archs/Unet.py:466-470 `conv10` + residual + data_inv_normalize, i.e. what yond_conv_out_f32
       does as a kernel of its own): when out4_dst is set the convolution must produce ONE 32-channel tile (Cout = tn = 32,
       algo 3, stride 1); its result v is not stored (dst may be NULL) and out4_dst[n][y][x][c] =
       ((sum_k out4_w[c][k] v[k]) + out4_b[c] + out4_x / ub) * ub is written instead (same operation order as
       yond_conv_out_f32: bit-identical). */
    const float* out4_w;  /* [4][Cout] */
    const float* out4_b;  /* [4] or NULL */
    const float* out4_x;  /* NHWC4 network input [N][Ho][Wo][4] (global residual) or NULL */
    const float* out4_ub; /* [N] per-image maximum (data_normalize) or NULL */
    float* out4_dst;      /* [N][Ho][Wo][4] or NULL (no fused projection) */
    /* Range guard of the half-precision operand paths (algo 2..5): a device word (or NULL) into which a kernel ORs
       YOND_STATUS_HALF_OVERFLOW when a staged activation does not fit fp16 (|a| > 65504: its h half would be +-inf and the
       result silently wrong).  The caller zeroes it, reads it after the forward and falls back to the fp32-input MFMA
       kernels (algo 0 / 1).  Weights are checked by the packing functions (YOND_EUNSUPPORTED). */
    unsigned int* status;
    /* Tensor formats (algo 3 and 4 only; 0 = [N][H][W][C] float32, the default everywhere).
       1 = SPLIT PLANES: what the consumer's LDS staging holds, stored once by the producer -- every value a (after the
       consumer's pre-activation, which the PRODUCER applies: post_act) as the half pair h = fp16(a), l = fp16((a - h) 2^11),
       laid out [N][C/16][channel half 0..1][part h, l][YOND_SP_PLANE_UNITS(H, W)] in units of 16 bytes = 8 consecutive
       channels of one pixel, pixel index y*W + x; the units behind H*W of every plane are a zero pad that the CALLER
       zeroes once (producers never write it; consumers read conv zero padding from it).  A consumer (in_fmt 1: src0 and
       src1, pre_act must be 0) stages its input by LDS-DMA alone; a producer (out_fmt 1: dst, 3x3 stride 1, no res,
       no fused projection) stores from the accumulator layout without an LDS transpose.  Bit-identical to staging the
       float32 tensor (the same split of the same float32 value).
       With algo 4 (h-only operands: the fp16 path, BASELINE cfg 5) format 1 means H-ONLY PLANES: the h halves alone,
       [N][C/16][channel half 0..1][YOND_SP_PLANE_UNITS(H, W)] units -- 2 bytes per element, the value every algo-4 consumer would have
       rounded to when staging the float32 tensor; the same roles (3x3 stride-1 producers / consumers, the stride-2 layers, the decoder
       GEMM with shuffle 1 or 2, dst2, the fused projection with split-plane input), the same zero pad, the same residual rule. */
    int in_fmt, out_fmt;
    /* 2 = PLANES OF 4 CHANNELS, float32 [N][C/4][H*W][4]: the format of the tensors that are read as RESIDUALS by a
       split-plane store (res_fmt 2 is required with out_fmt 1 and a residual, and only there): in the accumulator layout
       a lane reads / writes 16 bytes of its own pixel and a wave 512 contiguous bytes.  Producers: the stride-2 and
       transposed layers (out_fmt 2), yond_conv_in_f32; consumers: the register-staged input of a 3x3 layer (in_fmt 2:
       src0 and src1) and `res`. */
    int res_fmt;
    /* Measurement aid (algo 3 / 4; NULL = off): workgroup 0 adds its elapsed shader cycles (s_memtime) to clk[0] and the
       elapsed 100 MHz reference ticks (s_memrealtime) to clk[1]: over many launches clk[0] / clk[1] * 100 MHz is the clock the
       chip really held inside these kernels (bench.py `gfx_clock.in_kernel_mhz`).  Two counter reads and two atomics per launch. */
    unsigned long long* clk;
    /* Order in which the persistent workgroups of algo 3 / 4 walk the pixel tiles: 0 first row of tiles to last, 1 last to
       first.  A chain of memory-bound layers alternates it: the consumer then starts with the rows its producer wrote (and read)
       LAST, which are still in the 256 MB Infinity Cache, instead of the rows that were evicted first.  Results do not depend on it. */
    int tile_order;
    /* Second output of a stride-2 layer (algo 3 at split precision, split-plane input, out_fmt 2; NULL = none): SiLU of the
       stored value in SPLIT PLANES -- the tensor the next residual block's conv1 would otherwise build in its own staging
       (SiLU + split of every input pixel, once per output-channel tile: 4 / 8 times at 256 / 512 channels); with it that conv1
       stages by LDS-DMA alone (in_fmt 1, pre_act 0).  Same bits as the consumer-side staging.  Zero pad as for every
       split-plane tensor. */
    float* dst2;
} YondConvDesc;
#define YOND_STATUS_HALF_OVERFLOW 1u
#define YOND_FMT_NHWC_F32 0
#define YOND_FMT_SPLIT_PLANES 1
#define YOND_FMT_PLANES4 2
/* 16-byte units per plane of a split-plane tensor: H*W pixels + at least one zero unit, rounded to 128 bytes */
#define YOND_SP_PLANE_UNITS(H, W) ((((H) * (W)) + 8) / 8 * 8)
/* bytes of a split-plane tensor of C channels (C a multiple of 16); of an h-only tensor (algo 4): half of it */
#define YOND_SP_BYTES(N, C, H, W) ((size_t)(N) * ((C) / 16) * 4 * (size_t)YOND_SP_PLANE_UNITS(H, W) * 16)
#define YOND_HP_BYTES(N, C, H, W) ((size_t)(N) * ((C) / 16) * 2 * (size_t)YOND_SP_PLANE_UNITS(H, W) * 16)

/* Tile configuration for a convolution (needed to pack weights): kc = channel chunk, tn = channel-tile width.
 * N, Ho, Wo (GEMM-M extent; 0 = unknown) let the library pick the tn that fills its persistent grid best. */
int yond_conv_config(int ksize, int stride, int cin, int cout, int shuffle, int N, int Ho, int Wo, int* tn, int* kc);
/* Host-side weight packing: w is OIHW [cout][cin][k][k] (Conv2d) -- for a transposed conv pass the
 * already re-indexed [4*cout][cin][1][1] matrix.  dst has cout*cin*k*k floats. */
int yond_pack_conv_weight_f32(const float* w, int cout, int cin, int ksize, int tn, int kc, float* dst);
/* as yond_pack_conv_weight_f32, every weight stored as the half pair {fp16(w), fp16((w - fp16(w)) * 2^11)} (algo 5) */
int yond_pack_conv_weight_split_f32(const float* w, int cout, int cin, int ksize, int tn, int kc, float* dst);
int yond_conv2d_f32(const YondConvDesc* desc, void* stream);

/* Winograd F(2x2,3x3) variant of the 3x3 stride-1 convolution (same descriptor, algo = 1): 16 instead of 36
 * multiplications per output patch, fp32 throughout; the result differs from the direct kernel by rounding order
 * only.  yond_conv_wino_supported: the channel-tile width the kernel would use for (cin, cout) -- 64
 * (cout % 64 == 0: 8 x 32 pixel tiles), 32 (cout % 32 == 0: 16 x 32 pixel tiles) or 0 (not supported; cin % 8 != 0); put it in desc.tn and pass it to the packing function.
 * Weights: w OIHW [cout][cin][3][3] -> dst, 16*cout*cin floats (U = G g G^T in float64, rounded once). */
int yond_conv_wino_supported(int cin, int cout);
int yond_pack_conv_wino_weight_f32(const float* w, int cout, int cin, int tn, float* dst);

/* Split-operand fp16-MFMA form of the 3x3 convolutions (same descriptor, algo = 3 or 4; archs/modules.py:117-125,
 * 163-233 are the layers it serves).  Each fp32 operand is split when it is staged into LDS: h = fp16(a),
 * l = fp16((a - h) * 2^11); a*w = h_a h_w + 2^-11 (h_a l_w + l_a h_w) on v_mfma_f32_32x32x16_f16 with fp32
 * accumulation -- 22-23 significant bits per operand, error vs a float64 convolution no larger than the fp32 kernels'.
 * Precondition: |activations|, |weights| <= 65504 -- enforced: the packing functions return YOND_EUNSUPPORTED for a weight
 * outside that range and the kernels report an activation outside it through YondConvDesc.status.  yond_conv_split_supported: channel-tile width (64, 32) or 0; h-only weights (parts = 1) of a 3x3 stride-1 layer whose output channels divide by 128 may also be packed for tn = 128 (YondConvDesc.tn = 128, algo 4, plain tensors).
 * Weights: w OIHW [cout][cin][3][3] -> dst, cout*cin*9*parts/2 floats (packed halves).
 * ksize 1 (descriptor: ksize 1, shuffle 1, algo 3): the decoder's pixel-shuffle GEMM -- ConvTranspose2d 2x2 (src0, C0, low
 * resolution) + the skip tensor (src1, C1, at the OUTPUT resolution) + 1x1 shortcut folded into one weight matrix
 * [4*cout][C0+C1] (archs/Unet.py:445-463, modules.py:163-196) -- in the same kernel, for (C0+C1) % 48 == 0 and cout % 64 == 0. */
int yond_conv_split_supported(int ksize, int stride, int cin, int cout);
int yond_pack_conv_split_weight_f32(const float* w, int cout, int cin, int ksize, int tn, int parts, float* dst);
/* The same packing on the device (w and dst are device pointers; asynchronous on `stream`): the training step re-packs the
 * weights it has just updated without a host round trip.  A weight outside fp16's range sets bit 0 of *status (device int,
 * may be NULL) instead of failing the call. */
int yond_pack_conv_split_weight_dev_f32(const float* w, int cout, int cin, int ksize, int tn, int parts, float* dst, int* status,
                                        void* stream);
/* The same for several layers in one launch (a training step re-packs every layer): desc (device) holds 7 int64 per layer -- source
 * offset in src (floats), cout, cin, taps (ksize^2), tn, destination offset in dst (floats), index of the layer's first 16-byte
 * group among all layers' groups (cout * cin * taps / 4 groups per layer at parts = 2); ngroups = their total. */
int yond_pack_conv_split_weights_batch_dev_f32(const float* src, const long long* desc, int nlayers, float* dst, size_t ngroups,
                                               int* status, void* stream);

/* First layer: 3x3, Cin=4 -> Cout=32k, input NHWC4, optional division by the per-image maximum
 * (data_normalize, archs/Unet.py:427-431) and LeakyReLU(slope).  wpk from yond_pack_conv_in_weight_f32. */
int yond_conv_in_f32(const float* x, const float* ub /*[N] or NULL*/, int N, int H, int W, int Cout,
                     const float* wpk, const float* bias, float slope, float* dst,
                     int out_fmt /* YOND_FMT_NHWC_F32: [N][H][W][Cout]; YOND_FMT_PLANES4: [N][Cout/4][H*W][4] */, void* stream);
int yond_pack_conv_in_weight_f32(const float* w /*[cout][4][3][3]*/, int cout, float* dst /*cout*40*/);

/* Last layer: 1x1 Cin -> 4, + x/ub residual, * ub (archs/Unet.py:463-468). */
int yond_conv_out_f32(const float* feat /*[N][H][W][Cin]*/, int Cin, const float* w /*[4][Cin]*/,
                      const float* bias /*[4]*/, const float* x /*[N][H][W][4] or NULL*/,
                      const float* ub /*[N] or NULL*/, int N, int H, int W, float* dst /*[N][H][W][4]*/,
                      void* stream);

/* 2x2 max pooling, NHWC (UNetSeeInDark, archs/Unet.py:60-72). */
int yond_maxpool2_f32(const float* src, int N, int H, int W, int C, float* dst, void* stream);

/* sigma-conditioning vectors for one GuidedResidualBlock / SNR_Block (archs/modules.py:170-178,190-193
 * / 205-214, 225-231), folded with the conv biases into the epilogue (scale, shift) pairs of conv1 and
 * conv2.  t: [N] (already divided by ub when ub != NULL is NOT assumed: the kernel divides).
 *   kind 0 GuidedResidualBlock: m1 = gamma (w_a0,b_a0,w_a2,b_a2), m2 = beta (w_b,b_b)
 *   kind 1 SNR_Block:           m1 = sfm1, m2 = sfm2 (w_b0,b_b0 used as second first-layer)
 * Outputs s1,t1,s2,t2: [N][ld], only the first C entries of a row are written (channel padding stays 0).
 * `descs` is a DEVICE array of nblocks descriptors. */
typedef struct YondFilmDesc {
    int kind, C, ld;                           /* C <= 1024 real channels; ld = row stride of the outputs */
    const float *w_a0, *b_a0, *w_a2, *b_a2;   /* first MLP: 1->C, C->C */
    const float *w_b0, *b_b0;                  /* SNR only: second MLP first layer 1->C */
    const float *w_b, *b_b;                    /* guided: beta C->C ; SNR: sfm2 second layer C->C */
    const float *cb1, *cb2;                    /* conv1 / conv2 biases [C] */
    float *s1, *t1, *s2, *t2;                  /* [N][ld] */
} YondFilmDesc;
int yond_film_f32(const YondFilmDesc* descs, int nblocks, const float* t /*[N]*/, const float* ub /*[N] or NULL*/,
                  int N, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K5  local statistics for the noise-level estimator (box means in float64, cv2.blur semantics:
 * normalised k x k window, BORDER_REFLECT_101, result rounded to float32).
 * Replaces utils/isp_algos.py:234-242 (stdfilt) and the cv2.blur calls at YOND_SIDD.py:67-71, 95-98.
 * All maps are planar packed [4][h][w]; `tile_w` > 0 makes every tile_w columns an independent image
 * (the SIDD_256 re-tiling of YOND_SIDD.py:64-65, 91-93), 0 means the whole width.
 *   stage 1 (self):   from Bayer: mean = B_k(x), var = stdfilt(x,k)^2, blur2 = B_k2(x)
 *   stage 2 (self):   from blur2: lap = stdfilt(blur2, k)
 *   collab:           from noisy & denoised Bayer: var = stdfilt(lr)^2 - stdfilt(hr)^2,
 *                     mean = B_k(hr), lap = stdfilt(hr) */
int yond_box_stats_self1_f32(const float* bayer, int H, int W, int k, int k2, int tile_w, float* mean,
                             float* var, float* blur2, void* stream);
int yond_box_stats_self2_f32(const float* blur2, int h, int w, int k, int tile_w, float* lap, void* stream);
int yond_box_stats_collab_f32(const float* bayer_lr, const float* bayer_hr, int H, int W, int k, int tile_w,
                              float* mean, float* var, float* lap, void* stream);

/* (K5' -- the same maps in ONE pass, the B19 map never leaving the chip -- measured slower than K5 + sweep 1 and is built only
 * into experiment libraries: include/yond_hip_experiments.h, `python -m yond_public_amd.build --experiments`.) */

/* K5+ the hot-path producers of the estimator, one call per frame: the streaming kernels of K5 (stage 1 of the self mode also
 * collects the frame maximum into the workspace head) followed by sweep 1 of the threshold selection (yond_nle_stats_f32
 * below) on the same workspace, which this call resets first.  Same results and same workspace contract as K5';
 * `blur2` [4][H/2][W/2] is scratch (the B19 map).  The default of SimpleNLF: fewer instructions per pixel than K5' at
 * 32 B/px more HBM traffic, which the estimator (issue-bound, not HBM-bound) does not notice.  Any k that K5 takes. */
int yond_box_stats_self_stats_f32(const float* bayer, int H, int W, int k, int k2, int tile_w, float* mean, float* var,
                                  float* blur2, float* lap, const double* q_host, int nq, void* ws, void* stream);
int yond_box_stats_collab_stats_f32(const float* bayer_lr, const float* bayer_hr, int H, int W, int k, int tile_w,
                                    float* mean, float* var, float* lap, const double* q_host, int nq, void* ws, void* stream);

/* K6  exact order statistics of a non-negative float32 map (np.percentile's two neighbours).
 * ranks: nr 0-based ranks (device int64); out: nr float values (device).  ws: workspace of
 * yond_select_ws_bytes(nr) bytes.  Replaces the sort inside np.percentile at YOND_SIDD.py:26,80. */
size_t yond_select_ws_bytes(int nr);
int yond_select_ranks_f32(const float* data, size_t n, const int64_t* ranks_host, int nr, float* out, void* ws,
                          void* stream);
/* np.percentile(data, q, method='linear') (YOND_SIDD.py:26,80): q_host[nq] in percent (host array, nq <= 32),
 * out: nq float64 values (device).  Same workspace as yond_select_ranks_f32. */
int yond_percentiles_f32(const float* data, size_t n, const double* q_host, int nq, double* out, void* ws,
                         void* stream);

/* K6'/K7' the threshold selection of the estimator in two sweeps (YOND_SIDD.py:22-49: np.percentile at :26, the masked
 * bincounts at :37-43, score and argmin at :45-47), replacing yond_percentiles_f32 + yond_nlf_occupancy_f32 +
 * yond_nlf_score3_f64 on the hot path.  `ws`: device workspace of yond_nle_ws_bytes(n) bytes, 16-byte aligned.
 *   yond_nle_stats_f32      sweep 1 over (lap, mean) (n elements as rows of `width`): level-1 histogram of lap and, per
 *                           1/1000 mean bin, the smallest lap (a bin is occupied among lap <= T iff its smallest lap <= T);
 *                           resets the workspace first.  (The experimental one-pass box kernel does the same
 *                           while it produces the maps.)
 *   yond_nle_threshold_f32  sweep 2 over lap + finish: the exact order statistics, np.percentile(lap, q, 'linear') ->
 *                           ths; with want_score: npeaks[i], score = ths / (q * npeaks), i* = argmin(score[1:]) + 1 -> sel.
 *                           Needs sweep 1's state in `ws`; may be repeated on it (its own counters are reset when it ends).
 *   Results stay in the head of the workspace: yond_nle_state_layout gives the byte offsets of
 *   {ths double[32], sel double[4] = {i*, ths[i*], q[i*], score[i*]}, mom double[10], npeaks int32[32], frame_max_key};
 *   yond_nlf_moments_f32 can take th = ws + off[1] + 8 and mom = ws + off[2] directly. */
size_t yond_nle_ws_bytes(size_t n);
int yond_nle_stats_f32(const float* lap, const float* mean, size_t n, int width, const double* q_host, int nq, void* ws,
                       void* stream);
int yond_nle_threshold_f32(const float* lap, size_t n, const double* q_host, int nq, int want_score, void* ws, void* stream);
int yond_nle_state_layout(int* off /*[5]*/);
/* K7b on that workspace: the moment sums below sel[1] are ADDED to the workspace's mom, which the reset of sweep 1 zeroed (no
 * memset launch of its own): ONE call per sweep 1 -- a second call on the same state would double the sums (use
 * yond_nlf_moments_f32, which zeroes its destination, for anything else). */
int yond_nle_moments_f32(const float* lap, const float* mean, const float* var, size_t n, void* ws, void* stream);

/* K7a occupancy: one pass over (lap, mean), n elements laid out as rows of `width` (n % width == 0; pass the
 *   image row length so that a lane can walk down a column of the smooth maps; any width is correct):
 *   for thresholds ths[0..nt) (ascending, float64, device)
 *   occ  [nt][32] uint32 bitmap: bit (bin & 31) of word bin >> 5 set when bin floor(clip(mean,0,1)*1000) is
 *        occupied among {ths[i-1] < lap <= ths[i]}   (YOND_SIDD.py:37-43; prefix-OR over i gives npeaks) */
int yond_nlf_occupancy_f32(const float* lap, const float* mean, size_t n, int width, const double* ths, int nt,
                           uint32_t* occ, void* stream);

/* K7s score3 on the device (YOND_SIDD.py:37-47): npeaks[i] = number of bins occupied among lap <= ths[i],
 *   score = ths / (quants * npeaks) in float64, i* = argmin(score[1:]) + 1.
 *   quants_host: nt float64 on the host.  sel (device, 4 float64): {i*, ths[i*], quants[i*], score[i*]};
 *   npeaks (device, nt int32). */
int yond_nlf_score3_f64(const uint32_t* occ, const double* ths, const double* quants_host, int nt, double* sel,
                        int32_t* npeaks, void* stream);

/* K7b moments: one pass over (lap, mean, var): mom [2][5] float64 = {n, sum m, sum v, sum m^2, sum m v} over
 *   {lap < *th}, [0] all pixels, [1] only 1e-4 < mean < 0.8 (YOND_SIDD.py:77, 105; utils/isp_algos.py:348-350,
 *   352-364 as moment sums).  th: one float64 on the device (e.g. sel + 1 of yond_nlf_score3_f64). */
int yond_nlf_moments_f32(const float* lap, const float* mean, const float* var, size_t n, const double* th,
                         double* mom, void* stream);

/* Row H  bias LUT construction on the device (utils/isp_algos.py:49-140): for every knot lam <= th the
 * Poisson (*) Gaussian expectation of the VST, above th Foi's closed form.  lams: float64[n] (device),
 * bias: float32[n] (device out). */
int yond_bias_lut_f64(const double* lams, int n, double gain, double sigma, float* bias, void* stream);

/* N1  per-block PSNR / SSIM partial sums (YOND_SIDD.py:649-652, 679-697).  dn, hr: Bayer [H][W]; blocks of
 * bh x bw (row-major block order); out: [nblocks][ntiles][2] float64 = {sum of squared error, sum of SSIM
 * over the 'valid' map} per 32x32 tile, ntiles = yond_block_metrics_tiles(bh, bw); the host adds the tiles:
 * psnr = 10 log10(1 / (sum_se / (bh*bw))), ssim = sum_ssim / ((bh-10)*(bw-10)). */
int yond_block_metrics_tiles(int bh, int bw);
int yond_block_metrics_f32(const float* dn, const float* hr, int H, int W, int bh, int bw, double* out,
                           void* stream);

/* ------------------------------------------------------------------------------------------------
 * The per-frame parameter chain on the device: no host round trip between the estimator and the network.
 * Replaces the host arithmetic of YOND_SIDD.py:341-356 (beta -> K, sigma), :438-447 (round-2 guards), :263-264, 284-285
 * (lower = VST(0), upper = VST(scale), t = 1.03 / (upper - lower)), utils/isp_algos.py:101-108 (knot grid of get_bias)
 * and :345-365 (polyfit on the moment sums).
 * yond_frame_params_f64: reads the estimator's workspace (after yond_nle_moments_f32) and writes
 *   prm[YOND_PRM_N]   the parameter block below (float64),
 *   t_out             the network's noise level as float32 (or NULL),
 *   lut_x[lut_cap]    the knot grid of get_bias for the frame maximum (max_dev: a float32 device scalar, or NULL: the
 *                     maximum the estimator's first kernel collected).
 * mode 0: round 1 (:356), 1: round 2 (:438-447).  Branches that need more work than this chain does are FLAGGED in
 * prm[YOND_PRM_FLAGS] and left to the caller (host path): no flat area, knot capacity, K <= 0.
 * The *_dev_* entry points below take gain / sigma / lower / upper / the LUT size from such a block. */
#define YOND_PRM_BETA1 0
#define YOND_PRM_BETA2 1
#define YOND_PRM_GAIN 2
#define YOND_PRM_SIGMA 3
#define YOND_PRM_LO 4
#define YOND_PRM_HI 5
#define YOND_PRM_NSR 6
#define YOND_PRM_T 7          /* float32 value of the network's t */
#define YOND_PRM_FLAGS 8      /* YOND_PRM_FLAG_* as a number */
#define YOND_PRM_LUT_N 9
#define YOND_PRM_NSEL 10      /* pixels below the selected threshold */
#define YOND_PRM_TH 11
#define YOND_PRM_PCT 12
#define YOND_PRM_FRAME_MAX 13
#define YOND_PRM_N 16
#define YOND_PRM_FLAG_NO_FLAT_AREA 1   /* YOND_SIDD.py:79-84: the 25 % fallback needs the host path */
#define YOND_PRM_FLAG_LUT_CAPACITY 2   /* more knots than lut_cap (or more LDS than the LUT kernel has) */
#define YOND_PRM_FLAG_ROUND_ABORTED 4  /* round 2: beta1 < 0 (:445-447): the caller drops this round's output */
#define YOND_PRM_FLAG_BAD_ESTIMATE 8   /* K <= 0 or NaN: nothing downstream is defined */
int yond_frame_params_f64(const void* nle_ws, const float* max_dev, int mode,
                          double scale_est /* wp - bl: beta -> (K, sigma) in DN, :356 */, double scale /* p['scale'] = (wp - bl) / ratio: the VST's DN scale, :251 */,
                          double tfac, int lut_cap, double* prm, float* t_out, double* lut_x, void* stream);
/* The frame chain as ONE launch: yond_frame_params_f64 + yond_bias_lut_dev_f64 + yond_lut_table_f64 (same block, knots, ordinates,
 * table and flags).  Every workgroup derives the parameters itself, integrates one knot, and the last to arrive writes the table.
 * nle_ws: the estimator's workspace after yond_nle_moments_f32 (its arrival counter, word 2 of the state's tickets, is left at zero);
 * lut_y [lut_cap] float32, lut_ws yond_lut_ws_bytes(lut_cap) bytes. */
int yond_frame_chain_f64(void* nle_ws, const float* max_dev, int mode, double scale_est, double scale, double tfac, int lut_cap,
                         double* prm, float* t_out, double* lut_x, float* lut_y, void* lut_ws, void* stream);
/* Row H for large K * sigma (14-bit sensors at a digital gain: the Gaussian table of the integration exceeds the LDS and
 * yond_bias_lut_f64 returns YOND_EUNSUPPORTED): the same integration with the table in a caller-provided scratch buffer of
 * yond_bias_lut_big_scratch(gain, sigma, nwg) doubles, nwg workgroups striding over the knots.  Seconds, not microseconds. */
size_t yond_bias_lut_big_scratch(double gain, double sigma, int nwg);
/* get_bias_points (utils/isp_algos.py:142-160; BiasLUT.get_lut's pointwise branch :210-212 calls it with pho_min = 100,
 * close_form = True): the integration at ARBITRARY abscissae lams[n], sampling rate max(int(sqrt K), pho_min), Foi's closed
 * form above th only with close_form (else every point is integrated; pass lam_max = max(lams)).  Output float32 (bias32)
 * or float64 (bias64), scratch of yond_bias_points_scratch(...) doubles. */
size_t yond_bias_points_scratch(double gain, double sigma, int pho_min, int close_form, double lam_max, int nwg);
int yond_bias_points_f64(const double* lams, int n, double gain, double sigma, int pho_min, int close_form, double lam_max,
                         float* bias32, double* bias64, double* scratch, size_t scratch_doubles, int nwg, void* stream);
int yond_bias_lut_big_f64(const double* lams, int n, double gain, double sigma, float* bias, double* scratch,
                          size_t scratch_doubles, int nwg, void* stream);
/* Row H with (n, K, sigma) read from the block: launches lut_cap workgroups, those beyond prm[LUT_N] exit. */
int yond_bias_lut_dev_f64(const double* lams, int lut_cap, double* prm, float* bias, void* stream);
/* The LUT in the form K1 evaluates (per-interval coefficients + run table), prepared ONCE per frame instead of once per
 * workgroup of K1: lut_ws must hold yond_lut_ws_bytes(lut_cap) bytes.  n < 0: the size comes from prm. */
size_t yond_lut_ws_bytes(int lut_cap);
int yond_lut_table_f64(const double* lut_x, const float* lut_y, int n, const double* prm /* n < 0: also receives a flag */, void* lut_ws, void* stream);
/* K1 / K4 with the frame's constants in a parameter block and the LUT as a prepared table. */
int yond_pack_vst_norm_dev_f32(const float* bayer, int H, int W, float* out, int pad_l, int pad_r, int pad_t, int pad_b,
                               double scale, const double* prm, const void* lut_ws /* NULL: no bias correction */,
                               int lut_cap /* the capacity lut_ws was prepared for */, float* img_max, void* stream);
/* K1 of the device chain proper: as yond_pack_vst_norm_dev_f32 for the table yond_frame_chain_f64 / yond_lut_table_f64 prepared from
 * get_bias' knot grid (<= 3 evenly spaced runs of non-zero width), with the affine tail of the normalisation folded into the table's coefficients
 * (6 float64 operations per element instead of 11; the float32 result differs from yond_pack_vst_norm_dev_f32's for about one element
 * in 1e8, by one ulp).  prm is also written: a table of another shape sets YOND_PRM_FLAG_LUT_CAPACITY and nothing is computed. */
int yond_pack_vst_norm_chain_f32(const float* bayer, int H, int W, float* out, int pad_l, int pad_r, int pad_t, int pad_b,
                                 double scale, double* prm, const void* lut_ws, int lut_cap, float* img_max, void* stream);
int yond_denorm_ivst_unpack_dev_f32(const float* net_out, int Hp, int Wp, int pad_t, int pad_l, int h, int w, float* bayer,
                                    int mode, double scale, const double* prm, int clip01, void* stream);
/* K1 / K4 of the device chain for B equally sized frames that share the parameter block and the table (the 32 blocks of a SIDD image,
 * YOND_SIDD.py:392-407): bayer [B][H][W] -> out [B][Hp][Wp][4], img_max [B];  net_out [B][Hp][Wp][4] -> bayer_out [B][2h][2w]. */
int yond_pack_vst_norm_batch_dev_f32(const float* bayer, int B, int H, int W, float* out, int pad_l, int pad_r, int pad_t, int pad_b,
                                     double scale, const double* prm, const void* lut_ws, int lut_cap, float* img_max, void* stream);
int yond_denorm_ivst_unpack_batch_dev_f32(const float* net_out, int B, int Hp, int Wp, int pad_t, int pad_l, int h, int w,
                                          float* bayer_out, int mode, double scale, const double* prm, int clip01, void* stream);

/* ------------------------------------------------------------------------------------------------
 * N4, first slice: what one training step needs beyond the forward kernels (trainer_AWGN.py:101-117, losses/base_loss.py:81-113,
 * torch.optim.Adam).  Data gradients run on yond_conv2d_f32 with re-indexed weights (yond_public_amd/train.py).
 * yond_conv_wgrad_f32: dw[tap][Cout][Cin] = sum_p dy[p_out][Cout] x[p_in(p_out, tap)][Cin] on the fp32 MFMA (Cin, Cout
 *   multiples of 32, NHWC); mode 0: 3x3 pad 1 (stride 1 / 2: Ho = ceil(H / stride)), 1: ConvTranspose2d 2x2 stride 2
 *   (Ho = 2H; taps dy*2+dx), 2: 1x1.
 * yond_colsum_f32: db[c] = sum_p dy[p][c].   yond_l1_loss_f32: loss_sum = sum |pred - target|, grad = sign(.) / n.
 * yond_adam_step_f32: torch.optim.Adam's single-tensor update (no weight decay / amsgrad), step = 1, 2, ... */
int yond_conv_wgrad_f32(const float* x, const float* dy, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int mode,
                        int stride, float* dw, void* stream);
/* The same with a workspace: ws (device, yond_conv_wgrad_ws_bytes(...) bytes) takes the workgroups' partial sums, which a second
 * kernel adds up -- the K axis (pixels) is split over thousands of waves, and float atomics onto the few addresses of a
 * low-channel layer serialise at the memory side.  ws NULL: atomics, as yond_conv_wgrad_f32. */
size_t yond_conv_wgrad_ws_bytes(int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int mode, int stride);
int yond_conv_wgrad_ws_f32(const float* x, const float* dy, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int mode, int stride,
                           float* dw, float* ws, size_t ws_bytes, void* stream);
/* The weight gradient of a 3x3 stride-1 pad-1 layer on the fp16 matrix cores with fp32-accurate split operands (csrc/wgrad_split.hip;
 * the pixel axis is the MFMA's K axis, both tensors staged as fp16 halves and read transposed from LDS).  dw as above
 * ([9][Cout][Cin]); needs |dy| < 32 (bit 0 of *status is set otherwise: TrainStep's loss scale keeps it there).
 * yond_conv_wgrad_split_ws_bytes returns 0 for a layer the kernel does not take (the caller keeps yond_conv_wgrad_ws_f32). */
size_t yond_conv_wgrad_split_ws_bytes(int N, int H, int W, int Cin, int Cout);
int yond_conv_wgrad_split_f32(const float* x, const float* dy, int N, int H, int W, int Cin, int Cout, float* dw,
                              int with_bias /* bit 0: dw has 9 Cout Cin + Cout floats, the last Cout = db[co] = sum_p dy[p][co];
                                               bit 1: the weight gradient in OIHW order, dw[co][ci][tap] */,
                              float* ws, size_t ws_bytes, int* status, void* stream);
int yond_colsum_f32(const float* dy, size_t npix, int C, float* db, void* stream);
/* The guided block's middle (archs/modules.py:186-196) for training: out = SiLU(z * tk[n][c] + tb[n][c]) over z [N][P][C] with
 * per-image vectors tk, tb [N][C], and its backward in one pass: dz, dtk[n][c] = sum_p g z, dtb[n][c] = sum_p g with
 * g = dout * SiLU'(z tk + tb).  C in {32, 64, 128, 256, 512, 1024} (yond_film_silu_supported); other widths stay with the caller. */
int yond_film_silu_supported(int C);
int yond_film_silu_f32(const float* z, const float* tk, const float* tb, float* out, int N, size_t P, int C, void* stream);
int yond_film_silu_bwd_f32(const float* z, const float* tk, const float* tb, const float* dout, float* dz, float* dtk, float* dtb, int N,
                           size_t P, int C, void* stream);
/* The sigma-conditioning MLPs of a guided block for training (archs/modules.py:170-178): a = t w1 + b1, h = SiLU(a), tk = W2 h + b2,
 * tb = W3 SiLU(tk) + b3 for t [B]; w1, b1, b2, b3 [C]; W2, W3 [C][C] (Conv2d 1x1 weights).  fwd writes tk, tb at row stride ld >= C
 * (zero beyond C).  bwd: from the gradients dtk, dtb (row stride ld) the parameter gradients; scratch
 * 2 B C floats.  yond_silu_bwd_add_f32: dx = dres + dz SiLU'(x) (a residual block's input gradient in one pass). */
int yond_film_mlp_fwd_f32(const float* t, const float* w1, const float* b1, const float* W2, const float* b2, const float* W3,
                          const float* b3, int B, int C, int ld, float* tk, float* tb, void* stream);
int yond_film_mlp_bwd_f32(const float* t, const float* w1, const float* b1, const float* W2, const float* W3, const float* tk,
                          const float* dtk, const float* dtb, int B, int C, int ld, float* scratch, float* dw1, float* db1,
                          float* dW2, float* db2, float* dW3, float* db3, void* stream);
/* The same for ALL guided blocks of a network at once (n <= 12; the MLPs depend on sigma and the weights only): 2 launches forward,
 * 5 backward, instead of that many per block.  d: HOST array.  Forward reads t, w1 .. b3 and writes tk, tb ([B][ld]; columns beyond C
 * must already be zero); backward reads t, w1, b1, W2, W3, tk, dtk, dtb, uses scratch (2 * B * C floats) and writes dw1 .. db3. */
/* The plain GEMMs of a training step on the fp16 matrix cores with fp32-accurate split operands (csrc/gemm_split.hip): 1x1 convolutions
 * over one or two sources, their data gradients, ConvTranspose2d 2x2 (pixel-shuffle store) and its data gradient
 * (archs/Unet.py:447-461, archs/modules.py:142-147; backward of trainer_AWGN.py:116):
 *     y[p][n] = sum_s sum_k x_s[p][k] * B_s(k, n) (+ bias[n % nblk]),
 *     B_s(k, n) = w_s[(k % kblk) sk_lo + (k / kblk) sk_hi + (n % nblk) sn_lo + (n / nblk) sn_hi], zero where k % kblk >= k_real or n % nblk >= n_real
 * -- the float32 parameter itself, read through strides and split when staged (no packing pass).  x_s: [P][ld] float32, k (a multiple
 * of 32) channels taken; y: [P][ldy], n_p (a multiple of 32) channels written; shuffle != 0: y is [N][2H][2W][ldy], P = N H W,
 * n = j (n_p / 4) + co goes to pixel (2 yy + j / 2, 2 xx + j % 2), channel co (nblk must be n_p / 4).
 * status (may be NULL): TWO words -- bit 0 of status[0] is set when a staged x is NaN or beyond fp16's range (|x| > 65504), bit 0 of status[1]
 * when a weight is NaN or |w| >= 32 (the 2^11 w part leaves fp16's range).  (ABI 7; version 6 took one word, the weights'.) */
typedef struct YondGemmSrc {
    const float* x;
    const float* w;
    long long sk_lo, sk_hi;
    int ld, k, kblk, k_real;
} YondGemmSrc;
int yond_gemm_split_f32(const YondGemmSrc* src /* host array */, int nsrc /* 1 or 2 */, size_t P, int n_p, int n_real, long long sn_lo, long long sn_hi,
                        int nblk, const float* bias /* or NULL */, float* y, int ldy, int shuffle, int H, int W, int* status, void* stream);
typedef struct YondFilmMlpDesc {
    const float *t, *w1, *b1, *W2, *b2, *W3, *b3;
    float *tk, *tb;
    const float *dtk, *dtb;
    float* scratch;
    float *dw1, *db1, *dW2, *db2, *dW3, *db3;
    int B, C, ld, pad_;
} YondFilmMlpDesc;
int yond_film_mlp_fwd_multi_f32(const YondFilmMlpDesc* d, int n, void* stream);
int yond_film_mlp_bwd_multi_f32(const YondFilmMlpDesc* d, int n, void* stream);
int yond_silu_bwd_add_f32(const float* x, const float* dz, const float* dres, float* dx, size_t n, void* stream);
/* y = SiLU(x), n (a multiple of 4) floats. */
int yond_silu_f32(const float* x, float* y, size_t n, void* stream);
/* g [N][H][W][C] = dy [N][ceil(H/2)][ceil(W/2)][C] at the even pixels, 0 elsewhere (the stride-2 layers' data gradient runs as a
 * stride-1 convolution over it). */
int yond_zero_interleave_f32(const float* dy, int N, int Ho, int Wo, int C, int H, int W, float* g, void* stream);
int yond_l1_loss_f32(const float* pred, const float* target, size_t n, double* loss_sum, float* grad /* or NULL */, void* stream);
/* L1_Charbonnier_loss (losses/base_loss.py:69-79; Unet_Loss(charbonnier=True), :82-85): loss_sum = sum sqrt(diff^2 + eps),
 * grad = diff / sqrt(diff^2 + eps) / n in the float32 steps of torch's backward */
int yond_charbonnier_loss_f32(const float* pred, const float* target, size_t n, double eps, double* loss_sum, float* grad /* or NULL */,
                              void* stream);
int yond_adam_step_f32(float* p, const float* g, float* m, float* v, size_t n, double lr, double beta1, double beta2, double eps,
                       int step, void* stream);
/* The same with the step's scalars on the device (hyp[0] = lr / (1 - beta1^step), hyp[1] = 1 / sqrt(1 - beta2^step) as float32), for a
 * step captured in a hipGraph; status (three int words, optional): no update when bit 0 of any is set (the kernels' two range words and one
 * of the caller's, e.g. a non-finite loss). */
int yond_adam_step_dev_f32(float* p, const float* g, float* m, float* v, size_t n, double beta1, double beta2, double eps,
                           const float* hyp, const int* status, void* stream);

/* Measurement aid (bench.py; not on the reference's path): one wave sleeps for `us` microseconds (<= 5 s) of wall time and
 * writes out[0] = elapsed shader cycles (s_memtime), out[1] = elapsed 100 MHz reference ticks (s_memrealtime): the clock
 * the chip holds under the load running beside it = out[0] / out[1] * 100 MHz. */
int yond_clock_probe(double us, unsigned long long* out /* [2], device */, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* YOND_HIP_H */
