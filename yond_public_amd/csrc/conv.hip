// K2: the denoiser's convolutions as fp32 MFMA implicit GEMM on gfx950 (v_mfma_f32_32x32x2_f32).
//
// GEMM view per launch:  M = output pixels (32 consecutive x per MFMA tile), N = output channels,
// K = taps * Cin, walked as (channel chunk of KC) x (tap) x (group of 8 channels) x (4 k-steps).
//
// Structure (v2): PERSISTENT workgroups, one per CU (256 threads = one wave per SIMD), each walking a
// flat list of (tile, chunk) steps; a tile is TH x 32 output pixels x TN output channels, wave w owns rows
// [w*MW, (w+1)*MW) x TN channels = MW x NW accumulators of 32x32.  Per step, in ONE instruction stream:
//     issue the global loads of step s+1 (registers)  ->  MFMAs of step s on LDS buffer s&1, with the
//     (SiLU +) ds_write of step s+1 into buffer (s+1)&1 placed in their middle  ->  ONE barrier.
// So HBM/L2 latency, the activation prologue and the LDS writes hide under the wave's own MFMAs and do not
// depend on what a co-resident workgroup happens to be doing (v1: two single-buffered workgroups per CU ran
// in lockstep -- rocprof showed the MFMA pipe 46 % busy on the 32-channel layers).  The loads of the next
// tile's first chunk are issued during the current tile's last chunk, the residual is prefetched into
// registers during that chunk, and the epilogue only issues stores.
//
// LDS images (double-buffered, one dynamic array):
//   input  [IH][TWP] pixels x (KC+4) floats -- the +4 pad makes the pixel stride an odd multiple of 16 B,
//          so the 16 lanes of a ds_read_b128 group (consecutive pixels, same 4 channels) hit 16 different
//          16-byte bank groups; tap offsets are compile-time immediates.  For stride 2 the even and odd
//          input columns are stored in separate halves of each row so a tap still reads consecutive pixels.
//   weights [tap][KC/8][half][TN][4] -- lane (j = lane&31, half = lane>>5) reads its 4 k-steps with one
//          conflict-free ds_read_b128; produced in this order on the host by yond_pack_conv_weight_f32.
// MFMA operand map (cdna_hip_programming.md section 3): lane l supplies A[i=l&31][k=l>>5] and B[k=l>>5][j=l&31];
// for k-step t of group q the half h = l>>5 contributes channel q*8 + h*4 + t.  A = weights (i = output channel),
// B = pixels (j = pixel), so D: col = l&31 (pixel), row = (r&3) + 8*(r>>2) + 4*(l>>5) (output channel).
#include "common.h"
#include <cstdlib>

#ifndef YOND_B32_WGS
#define YOND_B32_WGS 2      // workgroups per CU of the (tn 32, kc 8) 3x3 configuration
#endif
#ifndef YOND_ABL
#define YOND_ABL 0      // timing-only ablations (bit 0: no weight DMA, bit 1: no input staging, bit 2: no epilogue stores)
#endif

template <int KS, int STRIDE, int TH, int TN, int KC>
struct ConvCfg {
    static constexpr int TW = 32;
    static constexpr int TAPS = KS * KS;
    static constexpr int PADK = KS / 2;
    static constexpr int IH = (TH - 1) * STRIDE + KS;
    static constexpr int IW = (TW - 1) * STRIDE + KS;
    static constexpr int HALF = (IW + 1) / 2;
    static constexpr int TWP = (STRIDE == 2) ? 2 * HALF : IW;
    static constexpr int PS = KC + 4;
    static constexpr int IN_FLOATS = IH * TWP * PS;
    static constexpr int Q = KC / 8;
    static constexpr int W_FLOATS = TAPS * Q * 2 * TN * 4;
    static constexpr int BUF_FLOATS = IN_FLOATS + W_FLOATS;
    static constexpr int NW = TN / 32;
    static constexpr int MW = TH / 4;
    static constexpr int EP_FLOATS = 4 * TN;              // epilogue scale / shift vectors, double-buffered by tile parity
    static constexpr int SMEM_BYTES = (2 * BUF_FLOATS + EP_FLOATS) * 4;
    static constexpr int SL = KC / 4;
    static constexpr int NITEM = IH * IW * SL;
    static constexpr int NIN = (NITEM + 255) / 256;
    static constexpr int NWV = W_FLOATS / 4;
    static constexpr int NWT = (NWV + 255) / 256;
};

// x * sigmoid(x) with the hardware exp2 / rcp (about 1 ulp each; relative error < 4e-7)
__device__ __forceinline__ float silu_fast(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f));
}

// 3x3: one workgroup per CU (its own MFMAs hide its memory traffic); 1x1 / transposed convolutions are memory
// bound (K = Cin only), so they run two workgroups per CU for more loads in flight.
template <int KS, int STRIDE, int KC, int TN>
constexpr int conv_wgs_per_cu() {
    if (KS == 3 && STRIDE == 1 && KC == 8) return TN == 32 ? YOND_B32_WGS : 2;
    if (KS == 3 && STRIDE == 2 && TN == 32) return 2;            // instantiated with 4-row tiles: two images fit the LDS
    // 1x1 / transposed layers: one tap per step, so a step is short and latency bound (loads, LDS round trip, barrier): the
    // more workgroups share a CU the better; the 32-wide shape needs < 170 registers and 45 KB of LDS: three fit
    return KS == 1 ? (TN == 32 ? 3 : 2) : 1;
}

// F16 (descriptor algo 2, the "fp16 MFMA conv path" of BASELINE cfg 5): the same kernel with the four fp32 k-steps of
// a group (8 channels) replaced by ONE v_mfma_f32_32x32x8_f16 -- the fragments are read from the fp32 LDS images as
// before and rounded to half (v_cvt_pk_f16_f32, round-to-nearest) on their way into the matrix core; accumulation,
// epilogue and every tensor in HBM stay fp32.  4x the matrix rate of the fp32 form, and the fp16 MFMA leaves the
// vector ALU free (the fp32 form does not).
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f16x4 to_half4(const f32x4 v) {
    f16x4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    return h;
}

// SPLIT mode (descriptor algo 5; see conv_split.hip for the arithmetic): an fp32 operand a is kept in its 4-byte LDS slot
// as the pair {h = fp16(a), l = fp16((a - h) * 2^11)}; a fragment of four slots is de-interleaved into the h and the l
// operand of v_mfma_f32_32x32x8_f16 with two v_perm_b32 each, and a product block costs three MFMAs
// (h_w h_x -> acc, h_w l_x and l_w h_x -> acc2, folded in as acc + acc2 * 2^-11 in the epilogue).
__device__ __forceinline__ unsigned split_pair(float x) {
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)((x - (float)h) * 2048.0f);
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    const f16x2 p = {h, l};
    return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ f32x4 split_pack4(const f32x4 v) {
    const float x0 = v[0], x1 = v[1], x2 = v[2], x3 = v[3];
    f32x4 o = {__uint_as_float(split_pair(x0)), __uint_as_float(split_pair(x1)), __uint_as_float(split_pair(x2)),
               __uint_as_float(split_pair(x3))};
    return o;
}
__device__ __forceinline__ void split_unpack4(const f32x4 v, f16x4& h, f16x4& l) {
    const float x0 = v[0], x1 = v[1], x2 = v[2], x3 = v[3];
    const unsigned d0 = __float_as_uint(x0), d1 = __float_as_uint(x1), d2 = __float_as_uint(x2), d3 = __float_as_uint(x3);
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 hh = {__builtin_amdgcn_perm(d1, d0, 0x05040100u), __builtin_amdgcn_perm(d3, d2, 0x05040100u)};
    const u32x2 ll = {__builtin_amdgcn_perm(d1, d0, 0x07060302u), __builtin_amdgcn_perm(d3, d2, 0x07060302u)};
    h = __builtin_bit_cast(f16x4, hh);
    l = __builtin_bit_cast(f16x4, ll);
}

template <int KS, int STRIDE, int TH, int TN, int KC, bool PRE, int MODE>
__global__ __launch_bounds__(256, (conv_wgs_per_cu<KS, STRIDE, KC, TN>())) void conv_mfma_kernel(const YondConvDesc d) {
    constexpr bool F16 = MODE == 1;          // operands rounded to half on their way into the matrix core (descriptor algo 2)
    constexpr bool SPL = MODE == 2;          // fp32-accurate split operands, staged as {h, l} half pairs (descriptor algo 5)
    constexpr bool CAN_DEFER = conv_wgs_per_cu<KS, STRIDE, KC, TN>() == 1;     // two workgroups per CU cover each other's epilogues
    using C = ConvCfg<KS, STRIDE, TH, TN, KC>;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 31;
    const int lh = lane >> 5;

    const int nct = d.Cout / TN;
    const int ntx = (d.Wo + 31) / 32;
    const int nty = (d.Ho + TH - 1) / TH;
    const int tiles_per_img = nct * ntx * nty;
    const int total = tiles_per_img * d.N;
    const int G = gridDim.x;
    // round r handles logical tiles [r*G, (r+1)*G); inside a round the workgroups of one XCD (blockIdx % 8
    // equal) take a contiguous run of logical tiles so halos / weight slices are shared in that XCD's L2.
    const int lslot = (G % 8 == 0) ? (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;

    const int Cin = d.C0 + d.C1;
    const int nchunk = Cin / KC;
    const int Cr = d.shuffle ? d.Cout / 4 : d.Cout;
    const int my_sl = (tid % C::SL) * 4;       // 256 % SL == 0: a thread keeps the same 16-byte slot in every item

    int in_lds[C::NIN];                        // LDS float offset of each staged slot, tile independent; items past the
                                               // end of the tile go to pixel 0's pad slot (never read): no branch
#pragma unroll
    for (int k = 0; k < C::NIN; ++k) {
        const int it = tid + k * 256;
        const int pix = it / C::SL, sl = it % C::SL;
        const int py = pix / C::IW, px = pix % C::IW;
        const int lp = (STRIDE == 2) ? py * C::TWP + (px & 1) * C::HALF + (px >> 1) : py * C::TWP + px;
        in_lds[k] = it < C::NITEM ? lp * C::PS + sl * 4 : KC;
    }

    // shuffle (transposed conv as GEMM) with a SECOND source: src1 is the skip tensor at the OUTPUT resolution
    // [N][2H][2W][C1], read at the sub-position (dy, dx) the tile's channel block stores to -- the fused
    // "ConvTranspose2d -> cat(up, skip) -> 1x1 shortcut" of the decoder blocks (engine.py), K = C0 + C1
    const bool up_skip = d.shuffle && d.C1 > 0;
    struct Tile {
        int ct, n, ox0, oy0;
        int goff[C::NIN];                      // pixel offset into the NHWC source (-1: outside the image -> zeros)
        int goff1[C::NIN];                     // up_skip: pixel offset into src1 (output resolution)
    };
    auto decode = [&](int t, Tile& T) {
        const int n = t / tiles_per_img;
        int b = t - n * tiles_per_img;
        T.n = n;
        T.ct = b % nct;
        b /= nct;
        const int tx = b % ntx, ty = b / ntx;
        T.ox0 = tx * 32;
        T.oy0 = ty * TH;
        const int ix0 = T.ox0 * STRIDE - C::PADK, iy0 = T.oy0 * STRIDE - C::PADK;
#pragma unroll
        for (int k = 0; k < C::NIN; ++k) {
            const int it = tid + k * 256;
            const int pix = it / C::SL;
            const int py = pix / C::IW, px = pix % C::IW;
            const int gy = iy0 + py, gx = ix0 + px;
            const bool ok = it < C::NITEM && gy >= 0 && gy < d.H && gx >= 0 && gx < d.W;
            T.goff[k] = ok ? ((n * d.H + gy) * d.W + gx) : -1;
            if (up_skip) {
                const int sp = (T.ct * TN) / Cr;               // a channel tile never straddles two sub-positions
                T.goff1[k] = ok ? ((n * 2 * d.H + 2 * gy + (sp >> 1)) * (2 * d.W) + 2 * gx + (sp & 1)) : -1;
            } else {
                T.goff1[k] = 0;
            }
        }
    };

    f32x4 vin[C::NIN];
    unsigned vin_ok = 0;                       // bit k: staged item k lies inside the image
    auto issue_loads = [&](const Tile& T, int ch, float* obuf) {
        const int c0 = ch * KC;
        const float* src;
        int Cs, cc;
        const bool second = c0 >= d.C0;
        if (!second) { src = d.src0; Cs = d.C0; cc = c0; }
        else { src = d.src1; Cs = d.C1; cc = c0 - d.C0; }
        const bool hires = second && up_skip;
        vin_ok = 0;
#pragma unroll
        for (int k = 0; k < C::NIN; ++k) {
            // outside the image: read pixel 0 (valid memory); the value is zeroed when it is written to LDS, so
            // nothing touches the loaded registers (and waits for them) before the middle of the MFMA stream
            const bool ok = T.goff[k] >= 0;
            const int po = hires ? T.goff1[k] : T.goff[k];
            if ((YOND_ABL & 2) == 0) vin[k] = *(const f32x4*)(src + (size_t)(ok ? po : 0) * Cs + cc + my_sl);
            vin_ok |= (ok ? 1u : 0u) << k;
        }
        // weight slice: already in LDS order on the host side -> linear copy by LDS-DMA (no VGPRs, no ds_write);
        // one wave-instruction moves 64 lanes x 16 B = 1 KiB to (wave-uniform base + lane*16)
        const float* wsrc = d.wpk + ((size_t)T.ct * nchunk + ch) * C::W_FLOATS;
#pragma unroll
        for (int k = 0; k < C::NWT; ++k) {
            const int it = tid + k * 256;
            if ((YOND_ABL & 1) == 0 && (C::NWV % 256 == 0 || it < C::NWV))
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + it * 4),
                                                 (__attribute__((address_space(3))) void*)(obuf + C::IN_FLOATS + (it - lane) * 4),
                                                 16, 0, 0);
        }
    };
    float amax = 0.0f;                                         // largest |activation| staged for a half-precision operand path
    auto write_item = [&](float* buf, int k) {
        f32x4 v = vin[k];
        if (PRE) { v[0] = silu_fast(v[0]); v[1] = silu_fast(v[1]); v[2] = silu_fast(v[2]); v[3] = silu_fast(v[3]); }
        const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
        v = ((vin_ok >> k) & 1u) ? v : z;                                                         // conv zero padding
        if (MODE != 0) amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));   // range guard
        if (SPL) v = split_pack4(v);
        if ((YOND_ABL & 2) == 0) *(f32x4*)(buf + in_lds[k]) = v;
    };
    auto write_lds = [&](float* buf) {
#pragma unroll
        for (int k = 0; k < C::NIN; ++k) write_item(buf, k);
    };

    f32x16 acc[C::MW][C::NW];
    f32x16 acc2[SPL ? C::MW : 1][SPL ? C::NW : 1];             // SPLIT: the two cross terms (carry the 2^11 scale)
    auto zero_acc = [&]() {
#pragma unroll
        for (int m = 0; m < C::MW; ++m)
#pragma unroll
            for (int nn = 0; nn < C::NW; ++nn)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[m][nn][r] = 0.0f;
                    if constexpr (SPL) acc2[m][nn][r] = 0.0f;
                }
    };
    // SPLIT: fold the cross terms into the main accumulator (before the epilogue / before parking a finished tile)
    auto fold_acc = [&]() {
        if constexpr (SPL) {
#pragma unroll
            for (int m = 0; m < C::MW; ++m)
#pragma unroll
                for (int nn = 0; nn < C::NW; ++nn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][nn][r] = fmaf(acc2[m][nn][r], 1.0f / 2048.0f, acc[m][nn][r]);
        }
    };
    // ---------------------------------------------------------------------------------------------------
    // Output side.  The MFMAs are issued with the WEIGHT fragment as the A operand and the pixel fragment as
    // B, so D = W . X^T: a lane owns ONE pixel (column = lane&31) and, in its 16 registers, the channels
    // (r&3) + 8*(r>>2) + 4*(lane>>5) of the 32-channel tile -- four runs of 4 consecutive channels.  The epilogue
    // therefore moves 16 bytes per lane and instruction (4 stores per 32x32 tile instead of 16), and the FiLM
    // vectors / the residual are read as float4 too.
    // Address of run g of element block (m, nn): dst[ubase(m, nn) + 8*g + lane_off]; ubase is wave-uniform.
    // The epilogue of a finished tile is DEFERRED into the first step of the next tile (accumulators parked in
    // `eacc`): its residual loads, FiLM arithmetic and stores then run in the shadow of that step's MFMAs
    // (rocprof / ablation: an exposed epilogue cost 17-29 % on the 32- and 64-channel layers).
    // ---------------------------------------------------------------------------------------------------
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int pstride = d.shuffle ? 2 * Cr : d.Cout;            // elements between horizontally adjacent pixels
    const int lane_off = li * pstride + 4 * lh;
    const float slope_eff = d.post_act == 2 ? d.slope : 1.0f;   // LeakyReLU(slope); slope 1 = identity
    auto out_ubase = [&](const Tile& T, int m, int nn) -> long long {
        const int oy = T.oy0 + wave_u * C::MW + m;
        const int cu = T.ct * TN + nn * 32;                     // a channel tile never straddles two sub-positions
        if (d.shuffle) {
            const int sp = cu / Cr, pcu = cu % Cr;
            return ((long long)(T.n * 2 * d.Ho + 2 * oy + (sp >> 1)) * (2 * d.Wo) + 2 * T.ox0 + (sp & 1)) * Cr + pcu;
        }
        return ((long long)(T.n * d.Ho + oy) * d.Wo + T.ox0) * d.Cout + cu;
    };
    f32x4 rres[C::MW][C::NW][4];                                // residual of the pending tile (zeros when there is none)
#pragma unroll
    for (int m = 0; m < C::MW; ++m)
#pragma unroll
        for (int nn = 0; nn < C::NW; ++nn)
#pragma unroll
            for (int g = 0; g < 4; ++g) { const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f}; rres[m][nn][g] = z; }
    auto load_res = [&](const Tile& T) {
        const bool col_ok = T.ox0 + li < d.Wo;
#pragma unroll
        for (int nn = 0; nn < C::NW; ++nn)
#pragma unroll
            for (int m = 0; m < C::MW; ++m) {
                const bool ok = col_ok && (T.oy0 + wave_u * C::MW + m < d.Ho);
                const long long base = ok ? out_ubase(T, m, nn) + lane_off : 0;      // invalid lanes read element 0..27
#pragma unroll
                for (int g = 0; g < 4; ++g) rres[m][nn][g] = *(const f32x4*)(d.res + base + (ok ? 8 * g : 0));
            }
    };
    // FiLM / bias vectors of a tile -> LDS (issued at the start of the tile's last step; the global latency hides
    // under that step's MFMAs, the step-end barrier publishes them).  Double-buffered by tile parity because the
    // deferred epilogue of tile i reads them while tile i+1 may already stage its own.
    float* s_ep = smem + 2 * C::BUF_FLOATS;
    auto stage_ep = [&](const Tile& T, int par) {
        if (tid < 2 * TN) {
            const int c = tid < TN ? tid : tid - TN;
            const int cu = T.ct * TN + c;
            const int eoff = (d.ebatch ? T.n * Cr : 0) + (d.shuffle ? cu % Cr : cu);
            float v;
            if (tid < TN) v = d.escale ? d.escale[eoff] : 1.0f;
            else v = d.eshift ? d.eshift[eoff] : 0.0f;
            s_ep[par * 2 * TN + tid] = v;
        }
    };
    // v = A*escale + eshift ; LeakyReLU ; + residual ; 16-byte stores.  Branch-free apart from the store mask.
    auto epilogue_from = [&](const Tile& T, f32x16 (&A)[C::MW][C::NW], int par) {
        const bool col_ok = T.ox0 + li < d.Wo;
        const float* ep = s_ep + par * 2 * TN;
#pragma unroll
        for (int nn = 0; nn < C::NW; ++nn) {
            f32x4 es[4], et[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                es[g] = *(const f32x4*)(ep + nn * 32 + 8 * g + 4 * lh);
                et[g] = *(const f32x4*)(ep + TN + nn * 32 + 8 * g + 4 * lh);
            }
#pragma unroll
            for (int m = 0; m < C::MW; ++m) {
                const bool ok = col_ok && (T.oy0 + wave_u * C::MW + m < d.Ho);
                float* op = d.dst + out_ubase(T, m, nn) + lane_off;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = fmaf(A[m][nn][4 * g + e], es[g][e], et[g][e]);
                        x = x > 0.0f ? x : x * slope_eff;
                        v[e] = x + rres[m][nn][g][e];
                    }
                    if (ok && ((YOND_ABL & 16) == 0 || d.N < 0)) *(f32x4*)(op + 8 * g) = v;
                }
            }
        }
    };

    // One step: MFMA groups g = (tap, q) of 4*MW*NW instructions each.  The fragments of group g+1 are read
    // from LDS before the MFMAs of group g are issued (one wave per SIMD: nothing else hides the LDS latency),
    // and the staged items of the NEXT step are converted / written to the other buffer a few groups in.
    constexpr int NG = C::TAPS * C::Q;
    // first group that carries a staged item: late enough for the global loads (issued after group 0, latency
    // 2-4 us with every CU streaming) to have landed
    constexpr int G0 = (NG * 5) / 9 > 0 ? (NG * 5) / 9 : 0;
    constexpr int IPG = (C::NIN + (NG - G0) - 1) / (NG - G0);             // items per group
    constexpr int GE = NG >= 9 ? 1 : NG - 1;                              // group that carries a pending epilogue (early:
                                                                          // its stores must retire before the step-end barrier)
    f32x16 eacc[C::MW][C::NW];                                            // accumulators of the pending tile
    auto mfma_step = [&](auto epi_tag, const float* buf, float* obuf, const Tile& Tl, int lch, const Tile& Tp, int ppar) {
        constexpr bool EPI = decltype(epi_tag)::value != 0;
        const float* a_base = buf + ((wave * C::MW * STRIDE) * C::TWP + li) * C::PS + lh * 4;
        const float* b_base = buf + C::IN_FLOATS + (lh * TN + li) * 4;
        f32x4 a[2][C::MW], bb[2][C::NW];
        auto load_frag = [&](int g, f32x4* af, f32x4* bf) {
            const int tap = g / C::Q, q = g % C::Q;
            const int dy = tap / KS, dx = tap % KS;
            const int xo = (STRIDE == 2) ? (dx & 1) * C::HALF + (dx >> 1) : dx;
#pragma unroll
            for (int m = 0; m < C::MW; ++m) af[m] = *(const f32x4*)(a_base + ((m * STRIDE + dy) * C::TWP + xo) * C::PS + q * 8);
#pragma unroll
            for (int nn = 0; nn < C::NW; ++nn) bf[nn] = *(const f32x4*)(b_base + (g * 2 * TN + nn * 32) * 4);
        };
        load_frag(0, a[0], bb[0]);
        static_for<0, NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            constexpr int k0 = (g - G0) * IPG;
            constexpr int nitem = (g < G0 || k0 >= C::NIN) ? 0 : (C::NIN - k0 < IPG ? C::NIN - k0 : IPG);
            if constexpr (g + 1 < NG) load_frag(g + 1, a[(g + 1) & 1], bb[(g + 1) & 1]);
            if constexpr (F16) {
                f16x4 ah[C::MW], bh[C::NW];
#pragma unroll
                for (int m = 0; m < C::MW; ++m) ah[m] = to_half4(a[g & 1][m]);
#pragma unroll
                for (int nn = 0; nn < C::NW; ++nn) bh[nn] = to_half4(bb[g & 1][nn]);
#pragma unroll
                for (int m = 0; m < C::MW; ++m)
#pragma unroll
                    for (int nn = 0; nn < C::NW; ++nn)
                        acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x8f16(bh[nn], ah[m], acc[m][nn], 0, 0, 0);                 // D = W . X^T
            } else if constexpr (SPL) {
                f16x4 ah[C::MW], al[C::MW], bh[C::NW], bl[C::NW];
#pragma unroll
                for (int m = 0; m < C::MW; ++m) split_unpack4(a[g & 1][m], ah[m], al[m]);
#pragma unroll
                for (int nn = 0; nn < C::NW; ++nn) split_unpack4(bb[g & 1][nn], bh[nn], bl[nn]);
#pragma unroll
                for (int m = 0; m < C::MW; ++m)
#pragma unroll
                    for (int nn = 0; nn < C::NW; ++nn)
                        acc2[m][nn] = __builtin_amdgcn_mfma_f32_32x32x8f16(bh[nn], al[m], acc2[m][nn], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < C::MW; ++m)
#pragma unroll
                    for (int nn = 0; nn < C::NW; ++nn)
                        acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x8f16(bh[nn], ah[m], acc[m][nn], 0, 0, 0);                 // D = W . X^T
#pragma unroll
                for (int m = 0; m < C::MW; ++m)
#pragma unroll
                    for (int nn = 0; nn < C::NW; ++nn)
                        acc2[m][nn] = __builtin_amdgcn_mfma_f32_32x32x8f16(bl[nn], ah[m], acc2[m][nn], 0, 0, 0);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int m = 0; m < C::MW; ++m)
#pragma unroll
                        for (int nn = 0; nn < C::NW; ++nn)
                            acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb[g & 1][nn][t], a[g & 1][m][t], acc[m][nn], 0, 0, 0);   // D = W . X^T
            }
            // the address arithmetic and the issue of the next step's global loads (and of the pending tile's
            // residual) ride in the shadow of group 0
            if constexpr (g == 0) {
                issue_loads(Tl, lch, obuf);
            }
            if constexpr (EPI && g == GE) epilogue_from(Tp, eacc, ppar);
#pragma unroll
            for (int j = 0; j < nitem; ++j) write_item(obuf, k0 + j);
            // Scheduling pipeline of this group (hipcc otherwise sinks the LDS reads to one MFMA before their use
            // and reuses the fragment registers, exposing the LDS latency on every group): first the reads of
            // group g+1, then the MFMAs of group g with the staging VALU work in their shadow, then the ds_writes.
            if constexpr (g + 1 < NG) __builtin_amdgcn_sched_group_barrier(0x100, C::MW + C::NW, 0);
#pragma unroll
            for (int i = 0; i < (F16 ? 1 : SPL ? 3 : 4) * C::MW * C::NW; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if constexpr (nitem > 0 || SPL || (EPI && g >= GE && g < GE + 4)) __builtin_amdgcn_sched_group_barrier(0x002, PRE ? 4 : 2, 0);
            }
            if constexpr (nitem > 0) __builtin_amdgcn_sched_group_barrier(0x200, nitem, 0);
        });
    };

    int tile = lslot;                            // logical tile of round 0 (the launch guarantees tile < total)
    if (tile >= total) return;
    Tile cur, ld, prev;                          // tile being computed / being loaded / awaiting its epilogue
    decode(tile, cur);
    ld = cur;
    prev = cur;
    int ch = 0, pb = 0, par = 0, ppar = 0;
    bool pend = false;
    // single-step tiles finish their epilogue at once; the memory-bound 1x1 kernels (two workgroups per CU, half the
    // register budget) never defer
    const bool defer = CAN_DEFER && nchunk >= 2;
    zero_acc();
    issue_loads(cur, 0, smem);
    write_lds(smem);
    __syncthreads();                             // (the barrier's fence also drains the LDS-DMA of the weights)
    while (true) {
        const bool last_ch = (ch + 1 == nchunk);
        const int ntile = last_ch ? tile + G : tile;
        const int nch = last_ch ? 0 : ch + 1;
        const bool has_next = ntile < total;
        float* buf = smem + pb * C::BUF_FLOATS;
        float* obuf = smem + (pb ^ 1) * C::BUF_FLOATS;      // all waves left obuf at the previous barrier
        // The loads of the next step are always issued (on the very last step they harmlessly re-read the first
        // chunk of the current tile), so the step body has no data-dependent branch.
        if (last_ch) {
            if (has_next) decode(ntile, ld);
            stage_ep(cur, par);
            if (CAN_DEFER && defer && d.res) load_res(cur);      // residual of this tile: needed by its (deferred) epilogue
        }
        if constexpr (CAN_DEFER) {
            if (pend) mfma_step(IntC<1>{}, buf, obuf, ld, nch, prev, ppar);
            else mfma_step(IntC<0>{}, buf, obuf, ld, nch, prev, ppar);
        } else {
            mfma_step(IntC<0>{}, buf, obuf, ld, nch, prev, ppar);
        }
        pend = false;
        __syncthreads();
        if (last_ch) {
            fold_acc();
            if (CAN_DEFER && defer) {
                if constexpr (CAN_DEFER) {
#pragma unroll
                    for (int m = 0; m < C::MW; ++m)
#pragma unroll
                        for (int nn = 0; nn < C::NW; ++nn) eacc[m][nn] = acc[m][nn];
                }
                prev = cur;
                ppar = par;
                pend = true;
            } else {
                if (d.res) load_res(cur);
                if ((YOND_ABL & 4) == 0) epilogue_from(cur, acc, par);
            }
            par ^= 1;
            zero_acc();
            cur = ld;
        }
        if (!has_next) break;
        tile = ntile;
        ch = nch;
        pb ^= 1;
    }
    if constexpr (CAN_DEFER) {
        if (pend) {                              // the last tile's epilogue has no next step to hide in
            if ((YOND_ABL & 4) == 0) epilogue_from(prev, eacc, ppar);
        }
    }
    if (MODE != 0 && d.status && !(amax <= 65504.0f)) atomicOr(d.status, YOND_STATUS_HALF_OVERFLOW);
}

template <int KS, int STRIDE, int TH, int TN, int KC, bool PRE, int MODE>
static int launch_conv_p(const YondConvDesc& d, hipStream_t st) {
    using C = ConvCfg<KS, STRIDE, TH, TN, KC>;
    static bool attr_set = false;
    auto kern = conv_mfma_kernel<KS, STRIDE, TH, TN, KC, PRE, MODE>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const long long total = (long long)(d.Cout / TN) * ((d.Wo + 31) / 32) * ((d.Ho + TH - 1) / TH) * d.N;
    if (total > 0x7fffffffLL) return YOND_EUNSUPPORTED;
    const int slots = 256 * conv_wgs_per_cu<KS, STRIDE, KC, TN>();   // persistent workgroups per CU: see conv_wgs_per_cu
    const int grid = total < slots ? (int)total : slots;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), C::SMEM_BYTES, st, d);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

template <int KS, int STRIDE, int TH, int TN, int KC, bool PRE>
static int launch_conv(const YondConvDesc& d, hipStream_t st) {
    if (d.algo == 5) {
        if constexpr (KS == 1) return launch_conv_p<KS, STRIDE, TH, TN, KC, PRE, 2>(d, st);      // split operands: the 1x1 / transposed layers
        else return YOND_EUNSUPPORTED;                                                          // (3x3: conv_split.hip, algo 3)
    }
    return d.algo == 2 ? launch_conv_p<KS, STRIDE, TH, TN, KC, PRE, 1>(d, st) : launch_conv_p<KS, STRIDE, TH, TN, KC, PRE, 0>(d, st);
}

// Tile configuration.  3x3 stride-1 layers choose between
//   A: 16-channel chunks, one persistent workgroup per CU, deferred epilogue   (long K loops)
//   B:  8-channel chunks, two workgroups per CU covering each other's bubbles   (measured +8 % where it fits)
// and between 32- / 64-wide channel tiles, by how well the tile count fills the last round of the persistent
// grid (256 or 512 slots).  The factors are measured relative throughputs on MI355X (round 1).
extern "C" int yond_conv_config(int ksize, int stride, int cin, int cout, int shuffle, int N, int Ho, int Wo, int* tn,
                                int* kc) {
    if (!((ksize == 3 && (stride == 1 || stride == 2)) || (ksize == 1 && stride == 1))) return YOND_EUNSUPPORTED;
    int k = (ksize == 3 && stride == 2) ? 8 : 16;
    if (cout % 32 != 0 || cin % 16 != 0 || cin <= 0 || cout <= 0) return YOND_EUNSUPPORTED;
    if (shuffle && (ksize != 1 || cout % 128 != 0)) return YOND_EUNSUPPORTED;
    const int ntile = shuffle ? cout / 4 : cout;      // a channel tile must not straddle two sub-positions
    int t = (ntile % 64 == 0) ? 64 : 32;
    if (N > 0 && Ho > 0 && Wo > 0) {
        const long long px = (long long)N * ((Ho + 7) / 8) * ((Wo + 31) / 32);
        auto fill = [&](int tw, int slots) {
            const long long tiles = px * (cout / tw);
            return (double)tiles / (double)(((tiles + slots - 1) / slots) * slots);
        };
        if (ksize == 3 && stride == 1) {
            double best = -1.0;
            const int tws[2] = {64, 32};
            for (int i = 0; i < 2; ++i) {
                if (ntile % tws[i] != 0) continue;
                const double sa = fill(tws[i], 256) * (tws[i] == 64 ? 1.00 : 0.97);
                const double sb = fill(tws[i], 512) * (tws[i] == 64 ? 1.08 : 1.05);
                if (sa > best) { best = sa; t = tws[i]; k = 16; }
                if (sb > best) { best = sb; t = tws[i]; k = 8; }
            }
        } else if (ksize == 3 && stride == 2) {
            t = 32;       // 4-row tiles, two workgroups per CU: measured >= the 64-wide single-workgroup form on every level
        } else if (t == 64 && ksize != 1) {
            if (fill(64, 256) < 0.85 && fill(32, 256) > fill(64, 256) + 0.1) t = 32;
        }
        // (1x1 / transposed layers: the 64-wide tile wherever the channel count allows -- measured faster than the 32-wide one
        // even where that fills the grid better: 2168 vs 2146 MP/s end to end)
    }
#ifdef YOND_EXPERIMENTS
    {
        const long e11 = yond_exp_long("YOND_1X1_TN", 0), es2 = yond_exp_long("YOND_S2_TN", 0), ekc = yond_exp_long("YOND_CONV_KC", 0),
                   etn = yond_exp_long("YOND_CONV_TN", 0);
        if (e11 && ksize == 1) t = e11 == 32 ? 32 : (ntile % 64 == 0 ? 64 : 32);
        if (es2 && ksize == 3 && stride == 2) t = es2 == 32 ? 32 : (ntile % 64 == 0 ? 64 : 32);
        if (ekc && ksize == 3 && stride == 1) k = ekc == 8 ? 8 : 16;
        if (etn && ksize == 3 && stride == 1 && ntile % 64 == 0) t = etn == 32 ? 32 : 64;
    }
#endif
    if (tn) *tn = t;
    if (kc) *kc = k;
    return YOND_OK;
}

extern "C" int yond_pack_conv_weight_f32(const float* w, int cout, int cin, int ksize, int tn, int kc, float* dst) {
    if (!w || !dst || (tn != 32 && tn != 64) || cout % tn != 0 || cin % kc != 0 || kc % 8 != 0) return YOND_EINVAL;
    const int taps = ksize * ksize, Q = kc / 8;
    size_t o = 0;
    for (int ct = 0; ct < cout / tn; ++ct)
        for (int ch = 0; ch < cin / kc; ++ch)
            for (int tap = 0; tap < taps; ++tap)
                for (int q = 0; q < Q; ++q)
                    for (int h = 0; h < 2; ++h)
                        for (int j = 0; j < tn; ++j)
                            for (int e = 0; e < 4; ++e) {
                                const int co = ct * tn + j, ci = ch * kc + q * 8 + h * 4 + e;
                                dst[o++] = w[((size_t)co * cin + ci) * taps + tap];
                            }
    return YOND_OK;
}

// the same image with every weight stored as the half pair {h, l} in its 4-byte slot (descriptor algo 5)
extern "C" int yond_pack_conv_weight_split_f32(const float* w, int cout, int cin, int ksize, int tn, int kc, float* dst) {
    const int rc = yond_pack_conv_weight_f32(w, cout, cin, ksize, tn, kc, dst);
    if (rc != YOND_OK) return rc;
    const size_t n = (size_t)cout * cin * ksize * ksize;
    for (size_t i = 0; i < n; ++i)
        if (!(fabsf(w[i]) <= 65504.0f)) return YOND_EUNSUPPORTED;          // its h half would be +-inf
    for (size_t i = 0; i < n; ++i) {
        const float v = dst[i];
        const _Float16 h = (_Float16)v;
        const _Float16 l = (_Float16)((v - (float)h) * 2048.0f);
        unsigned short hb, lb;
        __builtin_memcpy(&hb, &h, 2);
        __builtin_memcpy(&lb, &l, 2);
        const unsigned u = (unsigned)hb | ((unsigned)lb << 16);
        __builtin_memcpy(&dst[i], &u, 4);
    }
    return YOND_OK;
}

int yond_conv_wino_dispatch(const YondConvDesc& d, hipStream_t st);      // conv_wino.hip
int yond_conv_split_dispatch(const YondConvDesc& d, hipStream_t st);     // conv_split.hip

extern "C" int yond_conv2d_f32(const YondConvDesc* dp, void* stream) {
    if (!dp) return YOND_EINVAL;
    const YondConvDesc& d = *dp;
    hipStream_t st = (hipStream_t)stream;
    if (!d.src0 || !(d.dst || d.out4_dst) || !d.wpk || d.N <= 0 || d.H <= 0 || d.W <= 0 || d.Ho <= 0 || d.Wo <= 0) return YOND_EINVAL;
    if (d.out4_dst && ((d.algo != 3 && d.algo != 4) || !d.out4_w || d.Cout != 32 || d.stride != 1)) return YOND_EUNSUPPORTED;   // fused projection: conv_split.hip only
    if (d.C1 > 0 && !d.src1) return YOND_EINVAL;
    if (d.shuffle < 0 || d.shuffle > 2) return YOND_EINVAL;
    if (d.shuffle == 2 && d.algo != 3 && d.algo != 4) return YOND_EUNSUPPORTED;   // two sub-positions per tile: the split-operand kernel only
    if ((long long)d.N * d.H * d.W > 0x7fffffffLL) return YOND_EUNSUPPORTED;    // 32-bit pixel offsets
    if (d.pre_act != 0 && d.pre_act != 1) return YOND_EINVAL;
    if (d.in_fmt < 0 || d.in_fmt > 2 || d.out_fmt < 0 || d.out_fmt > 2 || (d.res_fmt != 0 && d.res_fmt != 2)) return YOND_EINVAL;
    if ((d.in_fmt || d.out_fmt || d.res_fmt) && d.algo != 3 && d.algo != 4) return YOND_EUNSUPPORTED;       // other formats: the split-operand kernel only (algo 4: h-only planes)
    if (d.algo == 1) return yond_conv_wino_dispatch(d, st);
    if (d.algo == 3 || d.algo == 4) return yond_conv_split_dispatch(d, st);
    if (d.algo != 0 && d.algo != 2 && d.algo != 5) return YOND_EINVAL;
    if (d.tn != 32 && d.tn != 64) return YOND_EINVAL;
    int tn, kc;
    const int rc = yond_conv_config(d.ksize, d.stride, d.C0 + d.C1, d.Cout, d.shuffle, 0, 0, 0, &tn, &kc);
    if (rc != YOND_OK) return rc;
    tn = d.tn;                                           // the layout the weights were packed for
    if (d.kc != 0) kc = d.kc;
    if (kc != 8 && kc != 16) return YOND_EINVAL;
    if (kc == 8 && d.ksize != 3) return YOND_EINVAL;
    if (d.ksize == 3 && d.stride == 2 && kc != 8) return YOND_EINVAL;
    if (d.C0 % kc != 0 || d.C1 % kc != 0 || d.Cout % tn != 0) return YOND_EUNSUPPORTED;
    if (d.shuffle && (d.ksize != 1 || (d.Cout / 4) % tn != 0)) return YOND_EUNSUPPORTED;
    if ((long long)d.N * d.H * d.W > 0x7fffffffLL) return YOND_EUNSUPPORTED;    // 32-bit pixel offsets
    if (d.stride == 2) {
        if (d.Ho != (d.H + 1) / 2 || d.Wo != (d.W + 1) / 2) return YOND_EINVAL;
    } else if (d.Ho != d.H || d.Wo != d.W) return YOND_EINVAL;
    if (d.pre_act != 0 && d.pre_act != 1) return YOND_EINVAL;
    if (d.post_act != 0 && d.post_act != 2) return YOND_EUNSUPPORTED;      // SiLU runs in the consumer's prologue (pre_act)
    if (d.pre_act == 1 && !(d.ksize == 3 && d.stride == 1)) return YOND_EUNSUPPORTED;
    if (d.ksize == 3 && d.stride == 1 && kc == 8) {
        if (d.pre_act) return tn == 64 ? launch_conv<3, 1, 8, 64, 8, true>(d, st) : launch_conv<3, 1, 8, 32, 8, true>(d, st);
        return tn == 64 ? launch_conv<3, 1, 8, 64, 8, false>(d, st) : launch_conv<3, 1, 8, 32, 8, false>(d, st);
    }
    if (d.ksize == 3 && d.stride == 1) {
        if (d.pre_act) return tn == 64 ? launch_conv<3, 1, 8, 64, 16, true>(d, st) : launch_conv<3, 1, 8, 32, 16, true>(d, st);
        return tn == 64 ? launch_conv<3, 1, 8, 64, 16, false>(d, st) : launch_conv<3, 1, 8, 32, 16, false>(d, st);
    }
    if (d.ksize == 3 && d.stride == 2)
        return tn == 64 ? launch_conv<3, 2, 8, 64, 8, false>(d, st) : launch_conv<3, 2, 4, 32, 8, false>(d, st);
    return tn == 64 ? launch_conv<1, 1, 8, 64, 16, false>(d, st) : launch_conv<1, 1, 8, 32, 16, false>(d, st);
}
