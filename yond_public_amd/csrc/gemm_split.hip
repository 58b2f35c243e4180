// N4: the plain GEMMs of a training step -- 1x1 convolutions over one or two sources (the decoder's short cuts, the output projection),
// their data gradients, the transposed 2x2 layers (a GEMM with a pixel-shuffle store) and their data gradients -- on the fp16 matrix
// cores with fp32-accurate SPLIT operands (trainer_AWGN.py:111-116; archs/Unet.py:447-461, archs/modules.py:142-147).
//
//   Y[p][n] = sum over sources s, channels k of X_s[p][k] * B_s(k, n)  (+ bias[n]),      p < P pixels, n < Nout
//
// They ran on the fp32-input MFMA (157 TF/s peak: compute bound at every level but the first, 1.7-1.8 ms of a 12 ms step).  Here:
// v_mfma_f32_32x32x16_f16 with A = 32 output channels x 16 input channels of the WEIGHTS, B = 16 input channels x 32 pixels; both
// operands are K-contiguous in memory ([pixel][channel] activations, the weight matrix read through strides), so a fragment is one
// 16-byte LDS read per lane.  Split operands as in conv_split_kernel.h / wgrad_split.hip: a = h + l 2^-11; the weights are staged in
// THREE parts (P = fp16(2^11 w), h, l) so that the three products P_w h_x + h_w l_x + l_w h_x of an fp32-accurate product add into
// ONE accumulator carrying the factor 2^11 (needs |w| < 32: bit 0 of *status otherwise).
// The weight matrix is whatever the layer's parameter is: element (k, n) of source s sits at
//     w_s + (k % kblk) sk_lo + (k / kblk) sk_hi + (n % nblk) sn_lo + (n / nblk) sn_hi        (zero where k % kblk >= k_real or n % nblk >= n_real)
// -- Conv2d 1x1 forward: sk_lo = 1, sn_lo = Cin; its data gradient: sk_lo = Cin, sn_lo = 1; ConvTranspose2d 2x2 [ci][co][2][2] with
// n = j cop + co: sk_lo = 4 cout, nblk = cop, sn_lo = 4, sn_hi = 1 -- so no packing pass exists: the arena's float32 weights are split
// when they are staged.  Workgroup = 4 waves = 128 pixels x (32 TN) output channels, 32-channel K steps, two LDS buffers, one barrier
// per step; the loads of step s + 1 are issued before the MFMAs of step s.
#include <type_traits>
#include "common.h"

typedef _Float16 gs_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 gs_f16x4 __attribute__((ext_vector_type(4)));

#define GS_MAXSRC 2
struct GsArgs {
    YondGemmSrc src[GS_MAXSRC];
    int nsrc;
    long long P;
    int n_p, n_real;
    long long sn_lo, sn_hi;
    int nblk;
    const float* bias;
    float* y;
    int ldy;
    int shuffle, H, W;             // shuffle: y is [N][2H][2W][ldy], n = j * (n_p / 4) + co -> pixel (2 yy + j / 2, 2 xx + j % 2), channel co
    int* status;
};

template <int TN>
__global__ __launch_bounds__(256) void gemm_split_kernel(const GsArgs a) {
    constexpr int NR = TN * 32;                                 // output channels (rows of the weight tile) per workgroup
    // (rows padded by 16 bytes: a fragment read is 16 bytes per lane at a stride of one row, and rows of 128 / 192 bytes put the 32 lanes
    // of a read on 2 / 4 bank groups -- 16- / 8-way conflicts, 3.6 us per K step; 144 / 208 bytes spread them over all 64 banks)
    __shared__ __attribute__((aligned(16))) char s_w[2][NR][3 * 64 + 16];   // [buffer][row][part P, h, l][32 k] halves
    __shared__ __attribute__((aligned(16))) char s_x[2][128][2 * 64 + 16];  // [buffer][pixel][part h, l][32 k] halves
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long p0 = (long long)blockIdx.x * 128;
    const int n0 = blockIdx.y * NR;
    int nst[GS_MAXSRC];
    int nsteps = 0;
    for (int s = 0; s < a.nsrc; ++s) { nst[s] = a.src[s].k / 32; nsteps += nst[s]; }
    float wmax = 0.0f, xmax = 0.0f;
    bool wbad = false, xbad = false;

    // ---- staging ----
    // pixels: 128 x 32 channels = 1024 items of 16 bytes, 4 per thread: item i -> pixel i / 8, channels 4 (i % 8) ..
    // weights: NR x 32 = NR * 8 groups of 4 consecutive k, TN per thread: group g -> row g / 8, k 4 (g % 8) ..
    f32x4 vx[2][4], vw[2][TN];                                  // two steps of loads in flight
    // No division and no 64-bit multiply in the loop: per thread the element offsets of its four pixel items and of its TN x 4 weight
    // elements are formed when a SOURCE starts (32-bit: the host entry checks the extents); the steps are loaded in order, so the
    // position inside the source (k0, and the block / offset pair of a two-level k index) advances by additions.
    long long prow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        long long p = p0 + ((tid + j * 256) >> 3);
        prow[j] = p < a.P ? p : a.P - 1;                        // (a valid address; the pixel's outputs are not stored)
    }
    long long noff[TN];
    bool nok[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + ((tid + j * 256) >> 3);
        const int nlo = n % a.nblk, nhi = n / a.nblk;          // (n_real: the real extent of the LOW output index, n % nblk)
        nok[j] = nlo < a.n_real;
        noff[j] = (long long)nlo * a.sn_lo + (long long)nhi * a.sn_hi;
    }
    const int kq = 4 * (tid & 7);                               // the thread's first channel inside a 32-channel step
    int cur_s = -1, left = 0, k0 = 0, klo0 = 0;                 // uniform: source, steps left in it, its channel position
    long long kbase = 0;                                        // uniform: klo0 sk_lo + khi sk_hi
    const float *xb = nullptr, *wb = nullptr;
    int xoff[4], toff[TN][4];
    auto load_step = [&](int rs) __attribute__((always_inline)) {
        if (left == 0) {                                        // (uniform) the next source starts
            ++cur_s;
            const YondGemmSrc& q = a.src[cur_s];
            left = nst[cur_s]; k0 = 0; klo0 = 0; kbase = 0;
            xb = q.x; wb = q.w;
#pragma unroll
            for (int j = 0; j < 4; ++j) xoff[j] = (int)(prow[j] * q.ld) + kq;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) toff[j][e] = (int)(noff[j] + (long long)(kq + e) * q.sk_lo);
        }
        const YondGemmSrc& q = a.src[cur_s];
#pragma unroll
        for (int j = 0; j < 4; ++j) vx[rs][j] = *(const f32x4*)(xb + xoff[j] + k0);
        const float* ws = wb + kbase;
        const int kleft = q.k_real - klo0 - kq;                 // element e of the thread's group is real while e < kleft
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = nok[j] & (e < kleft);
                const float t = ws[ok ? toff[j][e] : 0];
                v[e] = ok ? t : 0.0f;
            }
            vw[rs][j] = v;
        }
        k0 += 32; klo0 += 32; kbase += 32 * q.sk_lo; --left;
        if (klo0 >= q.kblk) { kbase += q.sk_hi - (long long)klo0 * q.sk_lo; klo0 = 0; }   // (uniform) the next block of a two-level k index
    };
    auto stage_step = [&](int buf, int rs) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = tid + j * 256;
            const f32x4 v = vx[rs][j];
            // (range guard: fmaxf drops a NaN operand, so a NaN is caught by its own test -- v != v -- and counted as out of range)
            xbad |= (v[0] != v[0]) | (v[1] != v[1]) | (v[2] != v[2]) | (v[3] != v[3]);
            xmax = fmaxf(xmax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
            const gs_f16x4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            const gs_f16x4 l = {(_Float16)((v[0] - (float)h[0]) * 2048.0f), (_Float16)((v[1] - (float)h[1]) * 2048.0f),
                                (_Float16)((v[2] - (float)h[2]) * 2048.0f), (_Float16)((v[3] - (float)h[3]) * 2048.0f)};
            *(gs_f16x4*)&s_x[buf][i >> 3][0 * 64 + 8 * (i & 7)] = h;
            *(gs_f16x4*)&s_x[buf][i >> 3][1 * 64 + 8 * (i & 7)] = l;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int gi = tid + j * 256;
            const f32x4 v = vw[rs][j];
            wbad |= (v[0] != v[0]) | (v[1] != v[1]) | (v[2] != v[2]) | (v[3] != v[3]);
            wmax = fmaxf(wmax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
            const gs_f16x4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            const gs_f16x4 P = {(_Float16)(v[0] * 2048.0f), (_Float16)(v[1] * 2048.0f), (_Float16)(v[2] * 2048.0f), (_Float16)(v[3] * 2048.0f)};
            const gs_f16x4 l = {(_Float16)((v[0] - (float)h[0]) * 2048.0f), (_Float16)((v[1] - (float)h[1]) * 2048.0f),
                                (_Float16)((v[2] - (float)h[2]) * 2048.0f), (_Float16)((v[3] - (float)h[3]) * 2048.0f)};
            *(gs_f16x4*)&s_w[buf][gi >> 3][0 * 64 + 8 * (gi & 7)] = P;
            *(gs_f16x4*)&s_w[buf][gi >> 3][1 * 64 + 8 * (gi & 7)] = h;
            *(gs_f16x4*)&s_w[buf][gi >> 3][2 * 64 + 8 * (gi & 7)] = l;
        }
    };

    f32x16 acc[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    load_step(0);
    if (nsteps > 1) load_step(1);
    stage_step(0, 0);
    __syncthreads();
    const int fr = lane & 31, fk = 16 * (lane >> 5);            // fragment row (channel / pixel), byte offset of its 8 k values
    // step st: the products of buffer st & 1; the loads of step st + 2 are issued (two steps of latency cover), the registers of
    // step st + 1 -- loaded during step st - 1 -- are split into the other buffer behind the MFMAs
    auto body = [&](int st, auto rsc) {
        constexpr int rs = decltype(rsc)::value;                // register set of step st + 2 (= that of step st)
        const int buf = st & 1;
        if (st + 2 < nsteps) load_step(rs);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const gs_f16x8 xh = *(const gs_f16x8*)&s_x[buf][wave * 32 + fr][0 * 64 + 32 * kk + fk];
            const gs_f16x8 xl = *(const gs_f16x8*)&s_x[buf][wave * 32 + fr][1 * 64 + 32 * kk + fk];
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                const gs_f16x8 wP = *(const gs_f16x8*)&s_w[buf][t * 32 + fr][0 * 64 + 32 * kk + fk];
                const gs_f16x8 wh = *(const gs_f16x8*)&s_w[buf][t * 32 + fr][1 * 64 + 32 * kk + fk];
                const gs_f16x8 wl = *(const gs_f16x8*)&s_w[buf][t * 32 + fr][2 * 64 + 32 * kk + fk];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wP, xh, acc[t], 0, 0, 0);      // 2^11 h_w h_x
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, acc[t], 0, 0, 0);      // h_w (2^11 l_x)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, acc[t], 0, 0, 0);      // (2^11 l_w) h_x
            }
        }
        if (st + 1 < nsteps) stage_step(buf ^ 1, rs ^ 1);
        __syncthreads();
    };
    for (int st = 0; st < nsteps; st += 2) {
        body(st, std::integral_constant<int, 0>{});
        if (st + 1 < nsteps) body(st + 1, std::integral_constant<int, 1>{});
    }
    // status[0] bit 0: a staged x part left fp16's range (|x| > 65504) or is NaN -- the word the training step's retry / the graph's Adam
    // gate read (an activation, or a loss-scaled gradient in the data-gradient uses); status[1] bit 0: a P part (2^11 w) did (|w| >= 32)
    if (a.status && (xbad || !(xmax <= 65504.0f))) atomicOr(a.status, 1);
    if (a.status && (wbad || !(wmax < 31.9f))) atomicOr(a.status + 1, 1);

    // ---- epilogue: accumulator layout: lane l holds pixel (column) l % 32, output channels (rows) (r & 3) + 8 (r >> 2) + 4 (l / 32) ----
    const long long p = p0 + wave * 32 + fr;
    if (p >= a.P) return;
    float* yrow;
    int cop4 = 0;
    if (a.shuffle) {
        const long long hw = (long long)a.H * a.W;
        const long long n = p / hw, rem = p - n * hw;
        const int yy = (int)(rem / a.W), xx = (int)(rem - (long long)yy * a.W);
        yrow = a.y + (((size_t)n * (2 * a.H) + 2 * yy) * (size_t)(2 * a.W) + 2 * xx) * a.ldy;
        cop4 = a.n_p / 4;
    } else {
        yrow = a.y + (size_t)p * a.ldy;
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int n = n0 + t * 32 + 8 * qd + 4 * (lane >> 5);
            if (n >= a.n_p) continue;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[t][4 * qd + e] * (1.0f / 2048.0f);
                if (a.bias) {
                    const int nb = (n + e) % a.nblk;
                    const bool ok = nb < a.n_real;
                    v += ok ? a.bias[ok ? nb : 0] : 0.0f;
                }
                o[e] = v;
            }
            if (a.shuffle) {
                const int j = n / cop4, co = n - j * cop4;        // (4 consecutive n share j: cop4 is a multiple of 4)
                *(f32x4*)(yrow + ((size_t)(j >> 1) * (size_t)(2 * a.W) + (j & 1)) * a.ldy + co) = o;
            } else {
                *(f32x4*)(yrow + n) = o;
            }
        }
    }
}

extern "C" int yond_gemm_split_f32(const YondGemmSrc* src, int nsrc, size_t P, int n_p, int n_real, long long sn_lo, long long sn_hi, int nblk,
                                   const float* bias, float* y, int ldy, int shuffle, int H, int W, int* status, void* stream) {
    if (!src || nsrc < 1 || nsrc > GS_MAXSRC || P == 0 || !y || n_p <= 0 || n_p % 32 || n_real <= 0 || n_real > n_p || nblk <= 0 || n_real > nblk) return YOND_EINVAL;
    if (P > 0x7fffffffull * 64) return YOND_EUNSUPPORTED;
    GsArgs a;
    for (int s = 0; s < nsrc; ++s) {
        const YondGemmSrc& q = src[s];
        if (!q.x || !q.w || q.k <= 0 || q.k % 32 || q.ld < q.k || q.k_real <= 0 || q.k_real > q.k || q.kblk <= 0 || q.k_real > q.kblk) return YOND_EINVAL;
        if (q.kblk < q.k && q.kblk % 32) return YOND_EUNSUPPORTED;        // (a 32-channel step must not straddle two k blocks)
        if ((double)P * q.ld >= 2.0e9) return YOND_EUNSUPPORTED;           // (32-bit element offsets inside the kernel)
        if (((uintptr_t)q.x & 15) || (q.ld & 3)) return YOND_EINVAL;
        a.src[s] = q;
    }
    if (shuffle) {
        if (H <= 0 || W <= 0 || P % ((size_t)H * W) || n_p % 128 || nblk != n_p / 4 || ldy < n_p / 4) return YOND_EINVAL;
    } else if (ldy < n_p) return YOND_EINVAL;
    if (((uintptr_t)y & 15) || (ldy & 3)) return YOND_EINVAL;
    a.nsrc = nsrc; a.P = (long long)P; a.n_p = n_p; a.n_real = n_real; a.sn_lo = sn_lo; a.sn_hi = sn_hi; a.nblk = nblk;
    a.bias = bias; a.y = y; a.ldy = ldy; a.shuffle = shuffle; a.H = H; a.W = W; a.status = status;
    const unsigned gx = (unsigned)((P + 127) / 128);
    if (n_p % 64 == 0) hipLaunchKernelGGL(gemm_split_kernel<2>, dim3(gx, (unsigned)(n_p / 64)), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(gemm_split_kernel<1>, dim3(gx, (unsigned)(n_p / 32)), dim3(256), 0, (hipStream_t)stream, a);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}
