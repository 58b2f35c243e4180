// K2s dispatcher, weight packing and shape support of the split-operand kernel (conv_split_kernel.h); the kernel's
// instantiations live in conv_split_*.hip (one group per translation unit, compiled in parallel).
#include "conv_split_kernel.h"

SPLIT_GROUP_K1S2(SPLIT_EXTERN)
SPLIT_GROUP_S1_64(SPLIT_EXTERN)
SPLIT_GROUP_S1_32(SPLIT_EXTERN)
SPLIT_GROUP_HALF(SPLIT_EXTERN)
SPLIT_GROUP_HALF128(SPLIT_EXTERN)
SPLIT_GROUP_HALF_TALL(SPLIT_EXTERN)
SPLIT_GROUP_FOLD(SPLIT_EXTERN)
SPLIT_GROUP_FOLD_S2(SPLIT_EXTERN)
SPLIT_GROUP_FOLD_K1(SPLIT_EXTERN)
SPLIT_GROUP_FOLD_NHWC(SPLIT_EXTERN)
SPLIT_GROUP_K1_D2(SPLIT_EXTERN)
SPLIT_GROUP_OSP(SPLIT_EXTERN)
SPLIT_GROUP_ISP(SPLIT_EXTERN)
SPLIT_GROUP_ISP_OSP(SPLIT_EXTERN)
SPLIT_GROUP_ISP_K1S2(SPLIT_EXTERN)
SPLIT_GROUP_K1_SUB2(SPLIT_EXTERN)
SPLIT_GROUP_WRES(SPLIT_EXTERN)
SPLIT_GROUP_D2(SPLIT_EXTERN)
SPLIT_GROUP_H_OSP(SPLIT_EXTERN)
SPLIT_GROUP_H_ISP_OSP(SPLIT_EXTERN)
SPLIT_GROUP_H_ISP_O4(SPLIT_EXTERN)
SPLIT_GROUP_H_K1S2(SPLIT_EXTERN)
SPLIT_GROUP_H_D2(SPLIT_EXTERN)
SPLIT_GROUP_H_SUB2(SPLIT_EXTERN)
SPLIT_GROUP_H128_OSP(SPLIT_EXTERN)
SPLIT_GROUP_H128_ISP_OSP(SPLIT_EXTERN)
SPLIT_GROUP_H_S2_TALL(SPLIT_EXTERN)
SPLIT_GROUP_H_K1_TALL(SPLIT_EXTERN)
SPLIT_GROUP_H_TALL4(SPLIT_EXTERN)
#ifdef YOND_EXPERIMENTS         // (round 6, measured no-gos: profiles/r06_experiments/README.md section 1b)
SPLIT_GROUP_S2_W4(SPLIT_EXTERN)
SPLIT_GROUP_S2_ROLES(SPLIT_EXTERN)
SPLIT_GROUP_WRES_W4(SPLIT_EXTERN)
#endif

// the channel-tile width the split kernel uses for a layer (0: not supported)
// ksize 1: the decoder's pixel-shuffle GEMM (cout = 4 sub-positions x channels of an output pixel; the descriptor has
// shuffle = 1): 48-channel steps, 64-wide tiles that must not straddle two sub-positions
extern "C" int yond_conv_split_supported(int ksize, int stride, int cin, int cout) {
    if (ksize == 1) {
        if (!(stride == 1 && cin > 0 && cin % 48 == 0 && cout > 0 && cout % 4 == 0)) return 0;
        return (cout / 4) % 64 == 0 ? 64 : ((cout / 4) % 32 == 0 ? 32 : 0);       // a tile may not straddle two sub-positions
    }
    if (ksize != 3 || (stride != 1 && stride != 2) || cin <= 0 || cout <= 0 || cin % 16 != 0 || cout % 32 != 0) return 0;
    if (stride == 2) return cout % 64 == 0 ? 64 : 0;
    return cout % 64 == 0 ? 64 : 32;
}

// OIHW fp32 weights -> the kernel's LDS image order [cout tile][cin chunk of 16][tap][channel half][part][tn][8 halves];
// parts = 2: h = fp16(w), l = fp16((w - h) * 2^11); parts = 1: h only.  dst: cout*cin*k*k * parts/2 floats.
// ksize 1 (w = the re-indexed [4*cout][cin][1][1] matrix of a transposed convolution): [cout tile][step of 48 channels]
// [chunk of 16][channel half][part][tn][8 halves].
extern "C" int yond_pack_conv_split_weight_f32(const float* w, int cout, int cin, int ksize, int tn, int parts, float* dst) {
    if (!w || !dst || (ksize != 3 && ksize != 1) || (tn != 32 && tn != 64 && !(tn == 128 && parts == 1)) || cout % tn != 0 || cin % 16 != 0 || (parts != 1 && parts != 2)) return YOND_EINVAL;
    _Float16* o = (_Float16*)dst;
    for (size_t i = 0, n = (size_t)cout * cin * ksize * ksize; i < n; ++i)
        if (!(fabsf(w[i]) <= 65504.0f)) return YOND_EUNSUPPORTED;          // its h half would be +-inf
    if (ksize == 1) {
        if (cin % 48 != 0) return YOND_EINVAL;
        for (int ct = 0; ct < cout / tn; ++ct)
            for (int ch = 0; ch < cin / 16; ++ch)              // (step, chunk) in order = chunk index
                for (int hh = 0; hh < 2; ++hh)
                    for (int p = 0; p < parts; ++p)
                        for (int j = 0; j < tn; ++j)
                            for (int e = 0; e < 8; ++e) {
                                const float v = w[(size_t)(ct * tn + j) * cin + ch * 16 + hh * 8 + e];
                                const _Float16 h = (_Float16)v;
                                *o++ = p == 0 ? h : (_Float16)((v - (float)h) * 2048.0f);
                            }
        return YOND_OK;
    }
    const int taps = 9;
    for (int ct = 0; ct < cout / tn; ++ct)
        for (int ch = 0; ch < cin / 16; ++ch)
            for (int tap = 0; tap < taps; ++tap)
                for (int hh = 0; hh < 2; ++hh)
                    for (int p = 0; p < parts; ++p)
                        for (int j = 0; j < tn; ++j)
                            for (int e = 0; e < 8; ++e) {
                                const int co = ct * tn + j, ci = ch * 16 + hh * 8 + e;
                                const float v = w[((size_t)co * cin + ci) * taps + tap];
                                const _Float16 h = (_Float16)v;
                                *o++ = p == 0 ? h : (_Float16)((v - (float)h) * 2048.0f);
                            }
    return YOND_OK;
}

// The same packing on the device (training: the weights change every step and never leave HBM).  One thread per 8 halves of the
// destination; a weight outside fp16's range sets bit 0 of *status (when given) instead of failing the call.
__global__ __launch_bounds__(256) void pack_split_weight_kernel(const float* __restrict__ w, int cout, int cin, int taps, int tn, int parts,
                                                                uint4* __restrict__ dst, size_t ngroups, int* __restrict__ status) {
    const size_t gi = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gi >= ngroups) return;
    size_t g = gi;
    const int j = (int)(g % tn); g /= tn;
    const int p = (int)(g % parts); g /= parts;
    const int hh = (int)(g % 2); g /= 2;
    const int tap = (int)(g % taps); g /= taps;
    const int nch = cin / 16;
    const int ch = (int)(g % nch), ct = (int)(g / nch);
    const int co = ct * tn + j;
    union { _Float16 h[8]; uint4 v; } u;
    bool over = false;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ci = ch * 16 + hh * 8 + e;
        const float v = w[((size_t)co * cin + ci) * taps + tap];
        over |= !(fabsf(v) <= 65504.0f);
        const _Float16 h = (_Float16)v;
        u.h[e] = p == 0 ? h : (_Float16)((v - (float)h) * 2048.0f);
    }
    dst[gi] = u.v;
    if (over && status) atomicOr(status, 1);
}

extern "C" int yond_pack_conv_split_weight_dev_f32(const float* w, int cout, int cin, int ksize, int tn, int parts, float* dst, int* status,
                                                   void* stream) {
    if (!w || !dst || (ksize != 3 && ksize != 1) || (tn != 32 && tn != 64 && !(tn == 128 && parts == 1)) || cout % tn != 0 || cin % 16 != 0 || (parts != 1 && parts != 2)) return YOND_EINVAL;
    if (ksize == 1 && cin % 48 != 0) return YOND_EINVAL;
    const int taps = ksize * ksize;
    const size_t ngroups = (size_t)cout * cin * taps * parts / 8;
    hipLaunchKernelGGL(pack_split_weight_kernel, dim3((unsigned)((ngroups + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, cout, cin, taps, tn,
                       parts, (uint4*)dst, ngroups, status);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

// All of a training step's layers in ONE launch (45 launches of ~7 us before): desc[l] = {source offset (floats) in src, cout, cin,
// taps, tn, destination offset (floats) in dst, first 16-byte group of the layer}; a thread finds its layer by bisection.
__global__ __launch_bounds__(256) void pack_split_weight_batch_kernel(const float* __restrict__ src, const long long* __restrict__ desc, int nlayers,
                                                                      float* __restrict__ dst, size_t ngroups, int* __restrict__ status) {
    const size_t gi = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gi >= ngroups) return;
    int lo = 0, hi = nlayers - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((size_t)desc[mid * 7 + 6] <= gi) lo = mid; else hi = mid - 1;
    }
    const long long* dl = desc + lo * 7;
    const float* w = src + dl[0];
    const int cin = (int)dl[2], taps = (int)dl[3], tn = (int)dl[4];
    size_t g = gi - (size_t)dl[6];
    const size_t gl = g;
    const int j = (int)(g % tn); g /= tn;
    const int p = (int)(g % 2); g /= 2;
    const int hh = (int)(g % 2); g /= 2;
    const int tap = (int)(g % taps); g /= taps;
    const int nch = cin / 16;
    const int ch = (int)(g % nch), ct = (int)(g / nch);
    const int co = ct * tn + j;
    union { _Float16 h[8]; uint4 v; } u;
    bool over = false;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ci = ch * 16 + hh * 8 + e;
        const float v = w[((size_t)co * cin + ci) * taps + tap];
        over |= !(fabsf(v) <= 65504.0f);
        const _Float16 h = (_Float16)v;
        u.h[e] = p == 0 ? h : (_Float16)((v - (float)h) * 2048.0f);
    }
    ((uint4*)(dst + dl[5]))[gl] = u.v;
    if (over && status) atomicOr(status, 1);
}

extern "C" int yond_pack_conv_split_weights_batch_dev_f32(const float* src, const long long* desc, int nlayers, float* dst, size_t ngroups,
                                                          int* status, void* stream) {
    if (!src || !desc || !dst || nlayers <= 0 || ngroups == 0) return YOND_EINVAL;
    hipLaunchKernelGGL(pack_split_weight_batch_kernel, dim3((unsigned)((ngroups + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, desc, nlayers,
                       dst, ngroups, status);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

int yond_conv_split_dispatch(const YondConvDesc& d, hipStream_t st) {
    const int parts = d.algo == 3 ? 2 : 1;
    if (d.shuffle == 2) {
        // the decoder GEMM with two sub-positions per 64-wide tile (conv_split_kernel.h, S2): Cr = Cout / 4 output channels,
        // GEMM columns ordered [dy][channel block of 32][dx][32]; src1's K range = [Cr channels at dx = 0 | the same at dx = 1 |
        // zero-weight channels up to a multiple of 48], split-plane inputs
        const int Cr = d.Cout / 4;
        const int pad = d.C1 - 2 * Cr;
        if (d.ksize != 1 || d.stride != 1 || d.Cout % 128 || d.tn != 64 || d.C0 % 16 || pad < 0 || pad >= 48 || pad % 16 ||
            (d.C0 + d.C1) % 48 || d.in_fmt != YOND_FMT_SPLIT_PLANES || d.out_fmt == YOND_FMT_SPLIT_PLANES || d.res_fmt || d.pre_act || d.res ||
            d.out4_dst || d.post_act == 1 || d.Ho != d.H || d.Wo != d.W || !d.src1)
            return YOND_EUNSUPPORTED;
        const long long e0 = (long long)d.N * (d.C0 / 16) * 4 * YOND_SP_PLANE_UNITS(d.H, d.W) * 4;
        const long long e1 = (long long)d.N * (Cr / 16) * 4 * YOND_SP_PLANE_UNITS(2 * d.H, 2 * d.W) * 4;
        if (e0 >= 0x7fffffffLL || e1 >= 0x7fffffffLL) return YOND_EUNSUPPORTED;
        if (parts == 1) return launch_split<1, 8, 64, 2, 1, 3, false, false, true, 2, false, true>(d, st);
        return launch_split<1, 8, 64, 2, 2, 3, false, false, true, 2, false, true>(d, st);
    }
    const int tn = yond_conv_split_supported(d.ksize, d.stride, d.C0 + d.C1, d.Cout);
    if (!tn || d.C0 % 16 != 0 || d.C1 % 16 != 0 || (d.shuffle != 0) != (d.ksize == 1)) return YOND_EUNSUPPORTED;
    // h-only operands (algo 4) may be packed for 128-channel tiles: 3x3 stride-1 layers with plain tensors (conv_split_kernel.h, HALF128)
    const bool half128 = parts == 1 && d.tn == 128 && tn == 64 && d.ksize == 3 && d.stride == 1 && d.Cout % 128 == 0 && !d.out4_dst && !d.dst2 &&
                         (d.out_fmt == YOND_FMT_SPLIT_PLANES || (!d.in_fmt && !d.out_fmt && !d.res_fmt));     // plain tensors, or a producer of the h-only flow
    // ... and the flow's stride-2 layers on 128-channel tiles (h-only planes in, planes of 4 channels out)
    const bool half128s2 = parts == 1 && d.tn == 128 && tn == 64 && d.ksize == 3 && d.stride == 2 && d.Cout % 128 == 0 && d.in_fmt == YOND_FMT_SPLIT_PLANES &&
                           d.out_fmt == YOND_FMT_PLANES4;
    // ... and its decoder GEMMs whose output pixels have >= 128 channels (a 128-column tile may not straddle two sub-positions)
    const bool half128k1 = parts == 1 && d.tn == 128 && tn == 64 && d.ksize == 1 && d.shuffle == 1 && (d.Cout / 4) % 128 == 0 && d.in_fmt == YOND_FMT_SPLIT_PLANES && !d.dst2;
    if (d.tn != tn && !half128 && !half128s2 && !half128k1) return YOND_EINVAL;             // the layout the weights were packed for
    if (d.post_act < 0 || d.post_act > 2) return YOND_EUNSUPPORTED;
    // tensor formats (include/yond_hip.h): split planes in (LDS-DMA staging) / out (stored from the accumulator layout), planes
    // of 4 channels for the float32 tensors that are read as residuals
    const bool isp = d.in_fmt == YOND_FMT_SPLIT_PLANES, osp = d.out_fmt == YOND_FMT_SPLIT_PLANES;
    const bool ip4 = d.in_fmt == YOND_FMT_PLANES4, op4 = d.out_fmt == YOND_FMT_PLANES4, rp4 = d.res_fmt == YOND_FMT_PLANES4;
    // (parts 1 -- the fp16 path: the same formats with H-ONLY planes, [n][C/16][channel half][units]: 2 bytes per element)
    if (d.res_fmt != YOND_FMT_NHWC_F32 && !rp4) return YOND_EINVAL;
    if (isp && d.pre_act) return YOND_EUNSUPPORTED;             // the producer applied the activation
    // second output (SiLU in split planes): the stride-2 layers with split-plane input and planes-of-4 output
    if (d.dst2 && !(isp && op4 && ((d.ksize == 3 && d.stride == 2 && d.Cout % 16 == 0) || (d.ksize == 1 && d.shuffle == 1 && tn == 64 && (d.Cout / 4) % 64 == 0)))) return YOND_EUNSUPPORTED;
    if (rp4 != (osp && d.res != nullptr)) return YOND_EUNSUPPORTED;   // a split-plane store reads its residual in planes of 4, nothing else does
    if (osp && d.res && !isp) return YOND_EUNSUPPORTED;              // ... and only conv2 of a block has one: split-plane input
    if (ip4 && (long long)d.N * (d.C0 > d.C1 ? d.C0 : d.C1) * d.H * d.W * (d.ksize == 1 ? 4 : 1) >= 0x7fffffffLL) return YOND_EUNSUPPORTED;   // 32-bit element offsets
    if (isp && (d.ksize == 1 || d.stride == 2)) {
        // (these stage split planes through registers: 32-bit element offsets over the whole tensor)
        const long long e0 = (long long)d.N * (d.C0 / 16) * 4 * YOND_SP_PLANE_UNITS(d.H, d.W) * 4;
        const long long e1 = d.ksize == 1 ? (long long)d.N * (d.C1 / 16) * 4 * YOND_SP_PLANE_UNITS(2 * d.H, 2 * d.W) * 4 : 0;
        if (e0 >= 0x7fffffffLL || e1 >= 0x7fffffffLL) return YOND_EUNSUPPORTED;
    }
    if (isp) {
        // 32-bit unit offsets inside a plane, 32-bit plane arithmetic
        const long long ps = (long long)YOND_SP_PLANE_UNITS(d.H, d.W) * (d.ksize == 1 ? 4 : 1);
        if (ps * 16 * 4 >= 0x7fffffffLL) return YOND_EUNSUPPORTED;
    }
    if (d.ksize == 1) {
        // the decoder GEMM: low-resolution input (C0) + skip tensor at the output resolution (C1), pixel-shuffle store
        if (d.pre_act || d.res || d.out4_dst || d.post_act == 1 || d.Ho != d.H || d.Wo != d.W || osp || ip4) return YOND_EUNSUPPORTED;
        if (parts == 1) {
            // h-only operands: the decoder GEMM of the split-plane flow only (h-only planes in, planes of 4 channels out)
            if (!isp || !d.src1) return YOND_EUNSUPPORTED;
            if (half128k1) return launch_split<1, 8, 128, 2, 1, 3, false, false, true, 2>(d, st);
            if (d.dst2) return launch_split<1, 8, 64, 2, 1, 3, false, false, true, 2, false, false, true>(d, st);
            // 64 columns: 16-row tiles (four rows per wave share every weight fragment: 1.25 instead of 1.5 KiB of LDS fragments per MFMA) where they fill the workgroups
            if (tn == 64 && op4 && (long long)(d.Cout / 64) * ((d.Wo + 31) / 32) * ((d.Ho + 15) / 16) * d.N >= 256 && yond_exp_long("YOND_SPLIT_K1_TALL", 1) != 0)
                return launch_split<1, 16, 64, 4, 1, 2, false, false, true, 2>(d, st);
            return tn == 32 ? launch_split<1, 8, 32, 1, 1, 3, false, false, true, 2>(d, st) : launch_split<1, 8, 64, 2, 1, 3, false, false, true, 2>(d, st);
        }
        if (isp && tn == 64 && d.Wo <= 16 && d.src1 && !d.dst2 && yond_exp_long("YOND_SPLIT_FOLD", 1) != 0) {
            // input at most 16 pixels wide: 2 / 4 sub-tiles per MFMA row (conv_split_kernel.h, FOLD) when that saves a round of 256 workgroups
            const int f = d.Wo <= 8 ? 4 : 2;
            const long long subs = (long long)d.N * ((d.Ho + 7) / 8) * ((d.Wo + 32 / f - 1) / (32 / f));
            const long long tiles_f = (long long)(d.Cout / 64) * ((subs + f - 1) / f), tiles = (long long)(d.Cout / 64) * ((d.Ho + 7) / 8) * d.N;
            const bool fits = (long long)d.H * d.W * (d.C0 > 4 * d.C1 ? d.C0 : 4 * d.C1) * 4 * f < (1LL << 31);
            if (fits && (tiles_f + 255) / 256 < (tiles + 255) / 256)
                return f == 2 ? launch_split<1, 8, 64, 2, 2, 3, false, false, true, 2, false, false, false, 2>(d, st) : launch_split<1, 8, 64, 2, 2, 3, false, false, true, 2, false, false, false, 4>(d, st);
        }
        if (isp && d.dst2) return launch_split<1, 8, 64, 2, 2, 3, false, false, true, 2, false, false, true>(d, st);
        if (isp) return tn == 32 ? launch_split<1, 8, 32, 1, 2, 3, false, false, true, 2>(d, st) : launch_split<1, 8, 64, 2, 2, 3, false, false, true, 2>(d, st);
        if (tn == 32) return launch_split<1, 8, 32, 1, 2, 3, false, false, true>(d, st);    // level 1 -> 0: 32-channel output pixels
        return launch_split<1, 8, 64, 2, 2, 3, false, false, true>(d, st);
    }
    if (d.stride == 2) {
        if (d.Ho != (d.H + 1) / 2 || d.Wo != (d.W + 1) / 2 || d.pre_act) return YOND_EINVAL;
        if (osp || ip4 || (op4 && d.res)) return YOND_EUNSUPPORTED;
        // stride 2: 4 x 32 output pixels read 9 x 65 input pixels -- two weight buffers fit beside the two input images
        if (isp && parts == 2 && d.Wo <= 16 && !d.src1 && !d.res && yond_exp_long("YOND_SPLIT_FOLD", 1) != 0) {
            // output at most 16 pixels wide: 2 / 4 sub-tiles per MFMA row (conv_split_kernel.h, FOLD) when that saves a round of 256 workgroups
            const int f = d.Wo <= 8 ? 4 : 2;
            const long long subs = (long long)d.N * ((d.Ho + 3) / 4) * ((d.Wo + 32 / f - 1) / (32 / f));
            const long long tiles_f = (long long)(d.Cout / 64) * ((subs + f - 1) / f), tiles = (long long)(d.Cout / 64) * ((d.Ho + 3) / 4) * d.N;
            const bool fits = (long long)d.H * d.W * d.C0 * 4 * f < (1LL << 31);
            if (fits && (tiles_f + 255) / 256 < (tiles + 255) / 256) {
                if (f == 2) return d.dst2 ? launch_split<2, 4, 64, 1, 2, 2, false, false, false, 2, false, false, true, 2>(d, st) : launch_split<2, 4, 64, 1, 2, 2, false, false, false, 2, false, false, false, 2>(d, st);
                return d.dst2 ? launch_split<2, 4, 64, 1, 2, 2, false, false, false, 2, false, false, true, 4>(d, st) : launch_split<2, 4, 64, 1, 2, 2, false, false, false, 2, false, false, false, 4>(d, st);
            }
        }
        if (isp && parts == 1) {
            if (!op4) return YOND_EUNSUPPORTED;
            // 8-row tiles (two output rows per wave) where they fill the persistent workgroups: the 4-row form's steps are bound by LDS reads
            // (2.0 KiB of fragments per MFMA with h-only operands; conv_split_kernel.h, SPLIT_GROUP_H_S2_TALL)
            const long long tiles8 = (long long)(d.Cout / d.tn) * ((d.Wo + 31) / 32) * ((d.Ho + 7) / 8) * d.N;
            const bool tall = tiles8 >= 256 && yond_exp_long("YOND_SPLIT_S2_TALL", 1) != 0;
            if (half128s2) {
                if (!tall) return YOND_EUNSUPPORTED;        // (the caller packs 64-channel tiles for small images)
                return d.dst2 ? YOND_EUNSUPPORTED : launch_split<2, 8, 128, 2, 1, 2, false, false, false, 2>(d, st);
            }
            if (tall) return d.dst2 ? launch_split<2, 8, 64, 2, 1, 2, false, false, false, 2, false, false, true>(d, st) : launch_split<2, 8, 64, 2, 1, 2, false, false, false, 2>(d, st);
            return d.dst2 ? launch_split<2, 4, 64, 1, 1, 2, false, false, false, 2, false, false, true>(d, st) : launch_split<2, 4, 64, 1, 1, 2, false, false, false, 2>(d, st);
        }
#ifdef YOND_EXPERIMENTS
        // round 6: two output rows per wave inside the same 4 x 32 x 64 tile (0.89 instead of 1.33 KiB of LDS fragments per MFMA) -- as eight waves in two roles
        // (four multiply, four move the data) or as four waves, one per SIMD.  Both bit-identical, both 12-36 us SLOWER per launch: one multiplying wave per SIMD
        // does not hide its own LDS latencies (profiles/r06_experiments/s2_*_ab.txt)
        if (isp && parts == 2 && yond_exp_long("YOND_SPLIT_S2_ROLES", 0) != 0)
            return d.dst2 ? launch_split<2, 4, 64, 2, 2, 2, false, false, false, 2, false, false, true, 0, 8, 1>(d, st) : launch_split<2, 4, 64, 2, 2, 2, false, false, false, 2, false, false, false, 0, 8, 1>(d, st);
        if (isp && parts == 2 && yond_exp_long("YOND_SPLIT_S2_W4", 0) != 0)
            return d.dst2 ? launch_split<2, 4, 64, 2, 2, 2, false, false, false, 2, false, false, true, 0, 4>(d, st) : launch_split<2, 4, 64, 2, 2, 2, false, false, false, 2, false, false, false, 0, 4>(d, st);
#endif
        if (isp && d.dst2) return launch_split<2, 4, 64, 1, 2, 2, false, false, false, 2, false, false, true>(d, st);
        if (isp) return launch_split<2, 4, 64, 1, 2, 2, false, false, false, 2>(d, st);
        if (!isp && parts == 2 && d.Wo <= 16 && !d.src1 && !d.res && !d.in_fmt && !d.out_fmt && yond_exp_long("YOND_SPLIT_FOLD", 1) != 0) {
            // the [N][H][W][C] form of the same (training's forward)
            const int f = d.Wo <= 8 ? 4 : 2;
            const long long subs = (long long)d.N * ((d.Ho + 3) / 4) * ((d.Wo + 32 / f - 1) / (32 / f));
            const long long tiles_f = (long long)(d.Cout / 64) * ((subs + f - 1) / f), tiles = (long long)(d.Cout / 64) * ((d.Ho + 3) / 4) * d.N;
            if ((tiles_f + 255) / 256 < (tiles + 255) / 256)
                return f == 2 ? launch_split<2, 4, 64, 1, 2, 2, false, false, false, 0, false, false, false, 2>(d, st) : launch_split<2, 4, 64, 1, 2, 2, false, false, false, 0, false, false, false, 4>(d, st);
        }
        return parts == 2 ? launch_split<2, 4, 64, 1, 2, 2, false>(d, st) : launch_split<2, 4, 64, 1, 1, 2, false>(d, st);
    }
    if (d.Ho != d.H || d.Wo != d.W) return YOND_EINVAL;
    if (d.out4_dst && (tn != 32 || (parts != 2 && !isp))) return YOND_EUNSUPPORTED;
    if (op4) return YOND_EUNSUPPORTED;                          // (3x3 stride-1 layers store [N][H][W][C] or split planes)
    // 12-row or 8-row tiles (64-channel kernels): 256 persistent workgroups walk the tiles in rounds, so a launch costs
    // rounds x rows per tile; a row of a 12-row tile is ~10 % cheaper (0.59 instead of 0.78 KiB of LDS fragments per MFMA, 1.5x the MFMA
    // work per barrier).  The full frames' levels (94 / 188 / 376 / 752 rows) fill whole rounds of 12-row tiles; batches of small
    // images do not: 8- and 16-row images are padded 1.5x by 12-row tiles and are exact in 8-row tiles, 32-row images of a batch of 64
    // take 1.5 rounds of 12-row tiles.  8-row tiles when they are >= 7 % cheaper by that count; fewer than 256 tiles: 8-row, as before.
    long long tiles12 = (long long)(d.Cout / 64) * ((d.Wo + 31) / 32) * ((d.Ho + 11) / 12) * d.N;
    {
        const long long tiles8 = (long long)(d.Cout / 64) * ((d.Wo + 31) / 32) * ((d.Ho + 7) / 8) * d.N;
        const long long r12 = (tiles12 + 255) / 256, r8 = (tiles8 + 255) / 256;
        if (8000 * r8 < 10044 * r12) tiles12 = 0;                 // cost8 = 8 r8 < 0.93 x cost12 = 0.93 x 10.8 r12
    }
    // 32 -> 32 channels: two weight slices in all -- on two buffers they stay resident in LDS (conv_split_kernel.h, wres)
    const bool wres = parts == 2 && tn == 32 && d.Cout == 32 && d.C0 + d.C1 == 32 && yond_exp_long("YOND_SPLIT_WRES", 1) != 0;
#ifdef YOND_EXPERIMENTS
    // round 6, measured no-go: level 0 as half-size workgroups (four waves, 8-row tiles, 80 KB of LDS), two per CU -- meant to put one workgroup's epilogue and first
    // loads under the other's MFMAs; same-box A/B: conv1 +2-7 %, conv2 +-0, the last convolution -4 %, the frame 2 % slower (profiles/r06_experiments/README.md, section 6)
    const bool w4 = wres && (long long)((d.Wo + 31) / 32) * ((d.Ho + 7) / 8) * d.N >= 2048 && yond_exp_long("YOND_SPLIT_L0_W4", 0) != 0;
    if (w4 && isp && osp && !d.out4_dst) return launch_split<1, 8, 32, 2, 2, 2, false, false, false, true, true, false, false, 0, 4>(d, st);
    if (w4 && isp && !osp && d.out4_dst) return launch_split<1, 8, 32, 2, 2, 2, false, true, false, true, false, false, false, 0, 4>(d, st);
    if (w4 && !isp && osp && d.pre_act && !d.out4_dst) return launch_split<1, 8, 32, 2, 2, 2, true, false, false, false, true, false, false, 0, 4>(d, st);
#endif
    if (wres && isp && osp && !d.out4_dst) return launch_split<1, 16, 32, 2, 2, 2, false, false, false, true, true>(d, st);
    if (wres && isp && !osp && d.out4_dst) return launch_split<1, 16, 32, 2, 2, 2, false, true, false, true, false>(d, st);
    if (wres && !isp && osp && d.pre_act && !d.out4_dst) return launch_split<1, 16, 32, 2, 2, 2, true, false, false, false, true>(d, st);
    if ((isp || osp) && parts == 1) {
        // the flow on h-only planes: conv1 (float32 planes of 4 in, SiLU staged, h-only store), conv2 / deep conv1 (h-only in by LDS-DMA), the
        // last convolution with the fused output projection
        if (osp && d.out4_dst) return YOND_EUNSUPPORTED;
        if (!isp && !(osp && d.pre_act)) return YOND_EUNSUPPORTED;
        if (isp && !osp) return (d.out4_dst && tn == 32) ? launch_split<1, 16, 32, 2, 1, 3, false, true, false, true>(d, st) : YOND_EUNSUPPORTED;
        if (half128) return isp ? launch_split<1, 8, 128, 2, 1, 3, false, false, false, true, true>(d, st) : launch_split<1, 8, 128, 2, 1, 3, true, false, false, false, true>(d, st);
        // four rows per wave where such tiles fill the workgroups (0.75 KiB of LDS fragments per MFMA instead of 0.89 / 1.17)
        const bool t4_64 = tn == 64 && (long long)(d.Cout / 64) * ((d.Wo + 31) / 32) * ((d.Ho + 15) / 16) * d.N >= 256;
        const bool t4_32 = tn == 32 && (long long)((d.Wo + 31) / 32) * ((d.Ho + 31) / 32) * d.N >= 256;
        if (isp) {
            if (t4_64) return launch_split<1, 16, 64, 4, 1, 2, false, false, false, true, true>(d, st);
            if (t4_32) return launch_split<1, 32, 32, 4, 1, 2, false, false, false, true, true>(d, st);
            if (tn == 64 && tiles12 >= 256) return launch_split<1, 12, 64, 3, 1, 2, false, false, false, true, true>(d, st);
            if (tn == 64) return launch_split<1, 8, 64, 2, 1, 3, false, false, false, true, true>(d, st);
            return launch_split<1, 16, 32, 2, 1, 3, false, false, false, true, true>(d, st);
        }
        if (t4_64) return launch_split<1, 16, 64, 4, 1, 2, true, false, false, false, true>(d, st);
        if (tn == 64 && tiles12 >= 256) return launch_split<1, 12, 64, 3, 1, 2, true, false, false, false, true>(d, st);
        if (tn == 64) return launch_split<1, 8, 64, 2, 1, 3, true, false, false, false, true>(d, st);
        return launch_split<1, 16, 32, 2, 1, 3, true, false, false, false, true>(d, st);
    }
    if (isp || osp) {
        if (osp && d.out4_dst) return YOND_EUNSUPPORTED;
        // images at most 16 pixels wide (the deepest level of a batch of small blocks): a 32-pixel MFMA row would be half padding -- two
        // 16-column sub-tiles (of one image or of two) share it instead (conv_split_kernel.h, FOLD)
        // Folded kernels: 4-row tiles, one row per wave (half the MFMAs per step and workgroup of the 8-row kernel: ~0.67x its step time, the
        // fragments of a single row are not shared between kernel rows); taken when the rounds of 256 workgroups say it is cheaper.
        int fold = 0;
        if (osp && (isp || d.pre_act) && tn == 64 && parts == 2 && d.Wo <= 16 && d.Wo == d.W && d.Ho == d.H && !d.src1 && yond_exp_long("YOND_SPLIT_FOLD", 1) != 0) {
            const int f = d.Wo <= 8 ? 4 : 2;
            const long long subs = (long long)d.N * ((d.Ho + 3) / 4) * ((d.Wo + 32 / f - 1) / (32 / f));
            const long long tiles_f = (long long)(d.Cout / 64) * ((subs + f - 1) / f), tiles8 = (long long)(d.Cout / 64) * ((d.Ho + 7) / 8) * d.N;
            // (a sub-tile of another image is addressed from sub-tile 0's plane base: at most f - 1 images further, in 32 bits)
            const bool fits = (long long)d.H * d.W * (d.C0 > d.Cout ? d.C0 : d.Cout) * 4 * f < (1LL << 31);
            if (fits && 2 * ((tiles_f + 255) / 256) < 3 * ((tiles8 + 255) / 256)) fold = f;
        }
        if (isp && osp) {
            if (fold == 2) return launch_split<1, 4, 64, 1, 2, 3, false, false, false, true, true, false, false, 2>(d, st);
            if (fold == 4) return launch_split<1, 4, 64, 1, 2, 3, false, false, false, true, true, false, false, 4>(d, st);
            if (tn == 64 && tiles12 >= 256) return launch_split<1, 12, 64, 3, 2, 2, false, false, false, true, true>(d, st);
            if (tn == 64) return launch_split<1, 8, 64, 2, 2, 3, false, false, false, true, true>(d, st);
            return launch_split<1, 16, 32, 2, 2, 3, false, false, false, true, true>(d, st);
        }
        if (osp) {
            if (fold == 2) return launch_split<1, 4, 64, 1, 2, 3, true, false, false, false, true, false, false, 2>(d, st);
            if (fold == 4) return launch_split<1, 4, 64, 1, 2, 3, true, false, false, false, true, false, false, 4>(d, st);
            if (tn == 64 && tiles12 >= 256) return d.pre_act ? launch_split<1, 12, 64, 3, 2, 2, true, false, false, false, true>(d, st) : launch_split<1, 12, 64, 3, 2, 2, false, false, false, false, true>(d, st);
            if (tn == 64) return d.pre_act ? launch_split<1, 8, 64, 2, 2, 3, true, false, false, false, true>(d, st) : launch_split<1, 8, 64, 2, 2, 3, false, false, false, false, true>(d, st);
            return d.pre_act ? launch_split<1, 16, 32, 2, 2, 3, true, false, false, false, true>(d, st) : launch_split<1, 16, 32, 2, 2, 3, false, false, false, false, true>(d, st);
        }
        if (tn == 64 && tiles12 >= 256) return launch_split<1, 12, 64, 3, 2, 2, false, false, false, true>(d, st);
        if (tn == 64) return launch_split<1, 8, 64, 2, 2, 3, false, false, false, true>(d, st);
        if (d.out4_dst) return launch_split<1, 16, 32, 2, 2, 3, false, true, false, true>(d, st);
        return launch_split<1, 16, 32, 2, 2, 3, false, false, false, true>(d, st);
    }
    if (half128) {
        if (d.Ho != d.H || d.Wo != d.W) return YOND_EINVAL;
        long long t12 = (long long)(d.Cout / 128) * ((d.Wo + 31) / 32) * ((d.Ho + 11) / 12) * d.N;
        const long long t8 = (long long)(d.Cout / 128) * ((d.Wo + 31) / 32) * ((d.Ho + 7) / 8) * d.N;
        if (8000 * ((t8 + 255) / 256) < 10044 * ((t12 + 255) / 256)) t12 = 0;
        if (t12 >= 256) return d.pre_act ? launch_split<1, 12, 128, 3, 1, 2, true>(d, st) : launch_split<1, 12, 128, 3, 1, 2, false>(d, st);
        return d.pre_act ? launch_split<1, 8, 128, 2, 1, 3, true>(d, st) : launch_split<1, 8, 128, 2, 1, 3, false>(d, st);
    }
    if (tn == 64) {
        // 12 x 32-pixel tiles, three rows per wave: 0.59 instead of 0.78 KiB of LDS fragments per MFMA, 1.5x the MFMA work per
        // step (and per barrier), and 94 / 188 / 376 / 752 rows fill 256 workgroups in whole rounds (1 / 2 / 4 / 8)
        const bool th12 = tiles12 >= 256;
        if (parts == 2 && !d.pre_act && !d.src1 && !d.in_fmt && !d.out_fmt && !d.res_fmt && !d.out4_dst && !(d.ebatch && (d.escale || d.eshift)) && d.Wo <= 16 &&
            yond_exp_long("YOND_SPLIT_FOLD", 1) != 0) {
            // the plain [N][H][W][C] layer on images at most 16 pixels wide (training patches' deep levels): folded 4-row tiles, as in the split-plane flow above
            const int f = d.Wo <= 8 ? 4 : 2;
            const long long subs = (long long)d.N * ((d.Ho + 3) / 4) * ((d.Wo + 32 / f - 1) / (32 / f));
            const long long tiles_f = (long long)(d.Cout / 64) * ((subs + f - 1) / f), tiles8 = (long long)(d.Cout / 64) * ((d.Ho + 7) / 8) * d.N;
            const double cost_plain = th12 ? 1.35 * (double)((tiles12 + 255) / 256) : (double)((tiles8 + 255) / 256);
            const bool fits = (long long)d.N * d.H * d.W * (d.C0 > d.Cout ? d.C0 : d.Cout) < (1LL << 31);
            if (fits && 0.667 * (double)((tiles_f + 255) / 256) < cost_plain)
                return f == 2 ? launch_split<1, 4, 64, 1, 2, 3, false, false, false, 0, false, false, false, 2>(d, st) : launch_split<1, 4, 64, 1, 2, 3, false, false, false, 0, false, false, false, 4>(d, st);
        }
        if (parts == 2 && th12) return d.pre_act ? launch_split<1, 12, 64, 3, 2, 2, true>(d, st) : launch_split<1, 12, 64, 3, 2, 2, false>(d, st);
        if (parts == 2) return d.pre_act ? launch_split<1, 8, 64, 2, 2, 3, true>(d, st) : launch_split<1, 8, 64, 2, 2, 3, false>(d, st);
        if (th12) return d.pre_act ? launch_split<1, 12, 64, 3, 1, 2, true>(d, st) : launch_split<1, 12, 64, 3, 1, 2, false>(d, st);
        return d.pre_act ? launch_split<1, 8, 64, 2, 1, 3, true>(d, st) : launch_split<1, 8, 64, 2, 1, 3, false>(d, st);
    }
    if (parts == 2 && d.out4_dst)      // the last convolution of the network with the 1x1 output projection in its epilogue
        return d.pre_act ? launch_split<1, 16, 32, 2, 2, 3, true, true>(d, st) : launch_split<1, 16, 32, 2, 2, 3, false, true>(d, st);
    if (parts == 2) return d.pre_act ? launch_split<1, 16, 32, 2, 2, 3, true>(d, st) : launch_split<1, 16, 32, 2, 2, 3, false>(d, st);
    // h only, 32 channels: 32-row tiles where they fill the 256 persistent workgroups
    if ((long long)((d.Wo + 31) / 32) * ((d.Ho + 31) / 32) * d.N >= 256)
        return d.pre_act ? launch_split<1, 32, 32, 4, 1, 2, true>(d, st) : launch_split<1, 32, 32, 4, 1, 2, false>(d, st);
    return d.pre_act ? launch_split<1, 16, 32, 2, 1, 3, true>(d, st) : launch_split<1, 16, 32, 2, 1, 3, false>(d, st);
}
