// K2s: 3x3 convolutions on the fp16 matrix cores with fp32-accurate SPLIT operands (descriptor algo 3).
//
// Why: on gfx950 the fp32-input MFMA runs at 1/16 of the fp16 rate and blocks the vector ALU while it runs
// ("fp32 MFMA and the vector ALU do not overlap").  Every fp32 operand a is therefore split ONCE, when it
// is staged into LDS (activations) or packed on the host (weights), into two halves
//        h = fp16(a)                 (round to nearest: 11 significant bits)
//        l = fp16((a - h) * 2^11)    (a - h is exact in fp32; the scale keeps l in fp16's normal range)
// so that a = h + l * 2^-11 to 22-23 significant bits, and a product a*w is evaluated as
//        h_a*h_w  +  2^-11 * (h_a*l_w + l_a*h_w)          (the l_a*l_w term, 2^-22 relative, is dropped)
// by THREE v_mfma_f32_32x32x16_f16 -- fp16 x fp16 products are exact in the fp32 accumulator; the two cross terms share
// a second accumulator that is folded in with one FMA in the epilogue.  Measured against a float64 convolution
// (tests/test_hip_conv.py::test_conv3x3_split_accuracy_beside_fp32_kernels) the error is that of the fp32 direct kernel.
// Limits: |a| < 65504 (fp16 max); an operand below fp16's normal range (|a| < 6.1e-5) keeps an absolute error of 2^-36
// instead of a relative one.  Activations and weights of the denoisers are O(1).
// With PARTS = 1 (descriptor algo 4) only the h halves are staged and multiplied: the plain fp16 MFMA path of
// BASELINE cfg 5 (fp32 tensors in HBM, fp32 accumulate), without the per-fragment conversions of conv.hip's mode 1.
//
// Structure: persistent 512-thread workgroups (two waves per SIMD), tile = TH x 32 output pixels x TN output
// channels, walked in 16-channel steps with ONE barrier per step.  Shapes (STRIDE, TH, TN, rows per wave MW):
// (1,12,64,3) and (1,8,64,2) for >= 64 channels, (1,16,32,2) for 32 channels, (2,4,64,1) for stride 2; K1: the decoder's
// 1x1 pixel-shuffle GEMM with three 16-channel chunks per step in place of the three tap columns.  LDS:
//   input   two images of planes [channel half 0..1][part h, l][IH x TWP pixels] x 16 bytes (8 halves): the 32 lanes of
//           a fragment read consecutive pixels = consecutive 16-byte units (conflict-free, no padding); stride 2 keeps
//           even and odd columns in separate halves of a row so a tap still reads consecutive pixels
//   weights two or three buffers [tap][channel half][part][TN] x 16 bytes, produced in this order by
//           yond_pack_conv_split_weight_f32 and copied by LDS-DMA one or two steps ahead.
// MFMA operand map (cdna_hip_programming.md section 3, 32x32x16): lane l (r = l&31, hh = l>>5) supplies
// A[row r][k = 8 hh + j] and B[k = 8 hh + j][col r], j = 0..7 -- one ds_read_b128 each.  A = weights (row = output
// channel), B = pixels, so a lane of D owns ONE pixel and channels (reg&3) + 8 (reg>>2) + 4 hh.
// Pipeline: all waves run the same program.  Between the MFMAs of step s the compiler is handed, group by group
// (sched_group_barrier), the step's memory instructions (loads of a later input into a free register set, LDS-DMA of
// later weights) and the staging of input(s+1) -- SiLU, zero padding, split, ds_write -- as single-element tasks: the fp16
// MFMA co-executes with the vector ALU.  The end-of-step barrier waits with a counted vmcnt.  See DESIGN.md (K2s) for the
// measurements behind each of these choices; the kernel runs at the board's power limit.
#pragma once
#include "common.h"
#include <cstdlib>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#ifndef SPLIT_DBG
#define SPLIT_DBG 0      // 1: timestamps of workgroup 0, waves 0 and 4 -> g_split_dbg (read with yond_split_debug_read)
#endif
#if SPLIT_DBG
static __device__ unsigned long long g_split_dbg[2][64][12];       // (per translation unit: the reader below sees the kernels of ITS unit)
#ifdef SPLIT_DBG_READER                                            // (each translation unit names its own reader)
extern "C" int SPLIT_DBG_READER(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_split_dbg), sizeof(g_split_dbg));
}
#endif
#define SDBG(slot)                                                                                  \
    do {                                                                                            \
        if (blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0 && dbg_step < 64)              \
            g_split_dbg[wave >> 2][dbg_step][slot] = __builtin_readcyclecounter();                  \
    } while (0)
#else
#define SDBG(slot) do {} while (0)
#endif

__host__ __device__ constexpr int yond_sp_plane_units(int H, int W) { return YOND_SP_PLANE_UNITS(H, W); }

template <int STRIDE, int TH, int TN, int MW, int PARTS, int NWB, bool K1 = false, int FOLD = 0, int NWAVE = 8, int ROLES = 0>
struct SplitCfg {
    // NWAVE 8: 512-thread workgroups, two waves per SIMD (every shipped kernel but one); NWAVE 4 (round 6, the stride-2 layer of the split path): ONE wave per
    // SIMD with twice the registers, so that a wave can own two output rows -- the only way to fewer LDS fragment reads per MFMA that fits the CU's 160 KB
    static constexpr int NT = 64 * NWAVE;
    // ROLES 1 (round 6, experiment): the first half of the waves (one per SIMD) only multiply -- they own the accumulators, two output rows each --, the second half
    // only moves data: global loads, weight DMA, the staging writes.  The multiplying wave's instruction stream is then fragment reads + MFMAs alone
    static constexpr int NWC = ROLES ? NWAVE / 2 : NWAVE;            // waves that hold accumulators
    static constexpr int NTS = ROLES ? NT / 2 : NT;                  // threads that stage and issue the memory operations
    static constexpr int KC = 16;
    // K1: the decoder's 1x1 GEMM (ConvTranspose2d 2x2 + cat + 1x1 shortcut folded, engine.py) -- one tap, so a step takes
    // THREE consecutive 16-channel chunks as pseudo-taps (a short step is all overhead); no halo
    static constexpr int NPT = K1 ? 3 : 1;                           // 16-channel chunks per step
    static constexpr int KSTEP = KC * NPT;
    static constexpr int TAPS = K1 ? NPT : 9;
    static constexpr int IH = K1 ? TH : (TH - 1) * STRIDE + 3;
    // FOLD = 2 / 4 (images at most 16 / 8 pixels wide -- the deep levels of a batch of small blocks): the MFMA's 32 pixels are FOLD sub-tiles of 32 / FOLD
    // columns side by side, each with its own halo columns: an LDS row = [1 + 16 + 1 | 1 + 16 + 1] units (stride 2: a sub-tile's 2 w + 1 input
    // columns and one unused column -- an even width keeps the even / odd column halves of the row aligned between sub-tiles)
    static constexpr int FSUBW = FOLD ? (STRIDE == 2 ? 2 * (32 / FOLD) + 2 : 32 / FOLD + 2) : 0;       // input columns of a sub-tile's share of the LDS row
    static constexpr int IW = K1 ? 32 : (FOLD ? FOLD * FSUBW : 31 * STRIDE + 3);
    static constexpr int HALF = (IW + 1) / 2;
    static constexpr int TWP = STRIDE == 2 ? 2 * HALF : IW;
    static constexpr int PLANE = IH * TWP * 4 + 4;                  // floats; the last 16 bytes take the staging items past the tile
    static constexpr int IN_FLOATS = NPT * 2 * PARTS * PLANE;       // planes [pseudo-tap][channel half][part]
    static constexpr int PIX_ITEMS = IH * IW * 4;                    // staging items of one 16-channel chunk
    static constexpr int W_FLOATS = TAPS * 2 * PARTS * TN * 4;
    static constexpr int RG = TH / MW;                              // row groups of waves
    static constexpr int NCW = NWC / RG;                            // channel groups of waves
    static constexpr int NW = TN / 32 / NCW;                        // 32-channel blocks per wave
    static constexpr int NITEM = NPT * IH * IW * 4;                 // 16-byte (4-channel) staging items per step
    static constexpr int NIN = (NITEM + NT - 1) / NT;
    static constexpr int NWV = W_FLOATS / 4;
    static constexpr int NWT = (NWV + NTS - 1) / NTS;
    static constexpr int NWT_MIN = NWV / NTS;                       // LDS-DMA instructions every wave issues per step
    static constexpr int NOPS = NWT + NIN;                          // vector-memory instructions per thread and step
    static constexpr int WAHEAD = NWB - 1;                          // the LDS-DMA of step s fetches weights(s + WAHEAD)
    static constexpr bool LOADS_FIRST = WAHEAD == 2;                // order of a step's memory operations (see the step pipeline)
    static constexpr int KEEP = WAHEAD == 2 ? NIN + NWT_MIN : NIN;  // memory operations that may stay in flight across a barrier
    static constexpr int W4_OFF = 2 * IN_FLOATS + NWB * W_FLOATS;   // floats: 4 x 32 weights of the fused output projection
    // (the projection's weights -- O4 -- and the FiLM vectors -- OSP -- share one region: no kernel has both)
    static constexpr int FILM_OFF = W4_OFF;                         // floats: per wave and 32-channel block {scale[32], shift[32]} (split-plane epilogue)
    static constexpr int FILM_FLOATS_ = NWC * NW * 64 * (FOLD && STRIDE == 1 ? FOLD : 1);     // (FOLD: every sub-tile may belong to another image; stride 2 never stores split planes)
    static constexpr int FILM_FLOATS = (TN == 32 && FILM_FLOATS_ < 128) ? 128 : FILM_FLOATS_;
    // h-only operands (PARTS 1) leave the transposed epilogue no consumed buffer large enough for its scratch (8 waves x 32 pixels x 36 floats):
    // their kernels use little LDS, so the scratch gets a region of its own behind everything else
    static constexpr int EP_FLOATS_C = NWC * 32 * 36;
    static constexpr bool EP_OWN = PARTS == 1 && STRIDE == 1 && EP_FLOATS_C > ((TN >= 64 && !K1) ? W_FLOATS : IN_FLOATS);
    static constexpr int EP_OFF = FILM_OFF + FILM_FLOATS;
    static constexpr int SMEM_BYTES = (EP_OFF + (EP_OWN ? EP_FLOATS_C : 0)) * 4;  // (W4: the O4 instantiation only; FILM: the split-plane epilogue only)
    // input in SPLIT PLANES (YondConvDesc.in_fmt 1): a step's 2 x PARTS x NPT planes arrive by LDS-DMA alone, one 16-byte unit
    // per lane, a wave-instruction per 64 consecutive units of a plane's LDS image
    static constexpr int UPP = IH * TWP;                            // units of a plane's LDS image
    static constexpr int WPP = (UPP + 63) / 64;                     // wave-instructions per plane
    static constexpr int NPL = NPT * 2 * PARTS;                     // planes per step
    static constexpr int NDS = NPL * WPP;                           // DMA wave-slots per step
    static constexpr int NDI = (NDS + NWAVE - 1) / NWAVE;           // ... per wave
    static_assert(NWB == 2 || NWB == 3, "two or three weight buffers");
    static_assert(!K1 || (STRIDE == 1 && PIX_ITEMS % NT == 0), "1x1 mode: whole chunks per pass of the staging threads");
    static_assert((NWAVE == 8 || NWAVE == 4) && RG * NCW == NWC && NW >= 1 && NW * NCW * 32 == TN, "wave grid does not cover the tile");
    static_assert(!ROLES || (NWAVE == 8 && !K1 && FOLD == 0), "role split: eight waves, the 3x3 forms");
    static_assert(FOLD == 0 || FOLD == 2 || FOLD == 4, "folded tiles: two or four sub-tiles");
};

// output rows m of a wave that read input row r (taps dy = r - m*stride in 0..2)
constexpr int split_gcd(int a, int b) { return b == 0 ? a : split_gcd(b, a % b); }
constexpr int split_pairs(int mw, int stride, int r) {
    int n = 0;
    for (int m = 0; m < mw; ++m) n += (r - m * stride >= 0 && r - m * stride <= 2) ? 1 : 0;
    return n;
}
__device__ __forceinline__ float split_silu(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f));
}
// ... wait for all but the N most recent vector-memory operations and for this wave's LDS traffic, then the barrier
template <int N>
__device__ __forceinline__ void split_barrier_keep_loads() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory"); }

// ISP: the input tensor(s) are SPLIT PLANES [n][C/16][channel half][part][H*W (+ zero pad)] of 16-byte units (8 halves) --
// what the staging below would have written into LDS, stored once by the PRODUCER's epilogue (OSP) with the consumer's
// pre-activation already applied; the consumer's step then has no vector work at all for its input: 2 x PARTS LDS-DMAs per
// 16-channel chunk (out-of-image units come from the zero unit behind every plane).  Same bits as staging the fp32 tensor.
// S2 (K1 with 32-channel output pixels, split-plane input): a 64-wide channel tile = the TWO sub-positions (dy, 0), (dy, 1) of a
// low-resolution pixel.  The skip tensor's K range is laid out [channels at dx = 0 | the same channels at dx = 1 | one zero chunk]
// (descriptor shuffle 2; the weights of a sub-position are zero in the other one's range), so the low-resolution input is
// staged once for both sub-positions: at 32 channels a tile takes 3 steps where two 32-wide tiles took 4, at 64 ... 256 channels
// (64-wide tile = 32 channels x two sub-positions) 6 / 11 / 22 steps where two tiles took 8 / 16 / 32.
// D2 (stride 2): the layer also stores SiLU(value) in split planes (YondConvDesc.dst2), from the same epilogue loop -- an
// instantiation of its own, so that the kernels without it keep their code and register allocation.
template <int STRIDE, int TH, int TN, int MW, int PARTS, int NWB, bool PRE, bool O4 = false, bool K1 = false, int ISPM = 0, bool OSP = false, bool S2 = false, bool D2 = false, int FOLD = 0, int NWAVE = 8, int ROLES = 0>
__global__ __launch_bounds__(64 * NWAVE, (NWAVE == 4 && STRIDE == 1) ? 2 : 1) void conv_split_kernel(const YondConvDesc d) {    // (four waves at stride 1: two workgroups per CU, 256 registers per wave)
    using C = SplitCfg<STRIDE, TH, TN, MW, PARTS, NWB, K1, FOLD, NWAVE, ROLES>;
    static_assert(!ROLES || (ISPM == 2 && !OSP && !O4 && !PRE), "role split: the register-staged split-plane input (stride 2)");
    static_assert(!FOLD || (!O4 && !S2 && ((!K1 && STRIDE == 1 && ISPM != 2 && OSP) || (STRIDE == 2 && (ISPM == 2 || (ISPM == 0 && !PRE && !D2))) || (K1 && ISPM == 2) || (!K1 && STRIDE == 1 && ISPM == 0 && !OSP && !PRE))),
                  "folded tiles: the split-plane data flow's kernels (3x3 stride 1: LDS-DMA or register-staged input, split-plane store; stride 2 and the decoder GEMM: register-staged split planes) "
                  "and the plain [N][H][W][C] 3x3 layer (training, UNetSeeInDark)");
    // split-plane input: ISPM 1 (ISP) by LDS-DMA, one step ahead -- the layers whose steps are long enough to cover the DMA's
    // latency; ISPM 2 (ISR) through the three register sets of the staging pipeline (a load has two steps to arrive), written to LDS
    // as whole 16-byte units with no arithmetic -- the stride-2 layers and the decoder GEMMs, whose steps hold 9-27 MFMAs per wave
    constexpr bool ISP = ISPM == 1, ISR = ISPM == 2;
    // (PARTS 1 -- the fp16 path, BASELINE cfg 5 -- runs the same data flow on H-ONLY planes: [n][C/16][channel half][H*W (+ zero pad)] units
    // of 8 halves, i.e. 2 bytes per element; the operands are the halves the plain-tensor h-only kernels would have rounded when staging)
    static_assert(ISPM == 0 || !PRE, "split-plane input: the producer applied the activation");
    static_assert(!OSP || (!O4 && !K1 && STRIDE == 1), "split-plane output: 3x3 stride-1 layers");
    static_assert(!S2 || (K1 && ISPM == 2 && TN == 64), "two sub-positions per tile: the decoder GEMM with register-staged split planes");
    static_assert(!D2 || (!OSP && !O4 && ((!K1 && STRIDE == 2) || (K1 && ISPM == 2 && !S2 && TN == 64))), "second output: the stride-2 layers and the decoder GEMM at split precision");
    constexpr int NACC = PARTS;                              // [0] h_w h_x ; [1] the two cross terms (carry the 2^11 scale)
    // register-staged 16-byte items: 4 per pixel and 16-channel chunk from float32 tensors (4 channels each); from split planes (ISR) one per
    // plane = 2 PARTS per pixel (a whole unit of 8 halves)
    constexpr int IPP = ISR ? 2 * PARTS : 4;
    constexpr int PIX_ITEMS = C::IH * C::IW * IPP;            // staging items of one 16-channel chunk
    constexpr int NITEM = C::NPT * PIX_ITEMS;                 // ... of a step
    static_assert(!K1 || PIX_ITEMS % C::NTS == 0, "1x1 mode: whole chunks per pass of the staging threads");
    constexpr int NIN = ISP ? 0 : (NITEM + C::NTS - 1) / C::NTS;   // register-staged 16-byte items per (staging) thread and step
    constexpr int NINA = NIN > 0 ? NIN : 1;                  // (array extents)
    constexpr int NDI = ISP ? C::NDI : 0;                    // input LDS-DMAs per wave and step
    // (K1: a thread's items k and k + PIX_ITEMS / NT are the SAME pixel of different 16-channel chunks -- one offset serves both)
    constexpr int KD = K1 ? PIX_ITEMS / C::NTS : NIN;      // distinct pixels among a thread's register-staged items
    // (LDS-DMA input: a slot's unit offset inside its plane depends on the slot's position in the plane only -- slots k and k + DPER
    // of a wave address the same units of different planes)
    constexpr int DPER = C::WPP / split_gcd(C::WPP, NWAVE);
    constexpr int NG = ISP ? (NDI < DPER ? NDI : DPER) : KD;  // per-tile offsets a thread keeps
    constexpr int NOPS = C::NWT + NIN + NDI;                 // vector-memory instructions per thread and step
    constexpr int KEEP = (ISP || ROLES) ? (C::WAHEAD == 2 ? C::NWT_MIN : 0) : (C::WAHEAD == 2 ? NIN + C::NWT_MIN : NIN);   // memory operations that may stay in flight across a barrier
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ROLES: the staging / memory work belongs to the second half of the workgroup (stid = its thread index there; the multiplying half never uses what it derives from it)
    const int stid = ROLES ? tid - C::NTS : tid;
    const bool consumer = !ROLES || __builtin_amdgcn_readfirstlane(wave) < C::NWC;
    const int li = lane & 31, lh = lane >> 5;
    // FOLD: lane li of a fragment = column li % FSW of sub-tile li / FSW; in the LDS row every sub-tile has its own two halo columns
    constexpr int FSW = FOLD ? 32 / FOLD : 32;
    const int fsub = FOLD ? li / FSW : 0, fcol = FOLD ? li % FSW : li, flds = (FOLD && !K1) ? li + (STRIDE == 2 ? 1 : 2) * fsub : li;     // (stride 2: position in the row's even-column half; the decoder GEMM has no halo)
    const int rg = wave % C::RG, cg = wave / C::RG;

    const int nct = d.Cout / TN;
    const bool computes = d.Cout > 0;                        // always true; opaque to the compiler (keeps the MFMA stretch a block of its own)
    // FOLD: the sub-tiles (image n, row band of TH rows, column band of FSW) are numbered through the whole batch and taken FOLD at a time: the
    // cursor's tx digit is the pair, its ty and n digits have one value each
    const int fnsx = (d.Wo + FSW - 1) / FSW, fnsy = (d.Ho + TH - 1) / TH;
    const int fper = fnsy * fnsx, fsubs = d.N * fper;
    struct FGeo { int n, oy0, ox0; };
    auto fold_geo = [&](int u) {                // sub-tile u: image, first row (Ho: past the last sub-tile -- nothing stored, zeros staged), first column
        FGeo g;
        const bool ok = u < fsubs;
        const int uu = ok ? u : fsubs - 1;
        g.n = uu / fper;
        const int r = uu - g.n * fper, ry = r / fnsx;
        g.oy0 = ok ? ry * TH : d.Ho;
        g.ox0 = (r - ry * fnsx) * FSW;
        return g;
    };
    const int ntx = FOLD ? (fsubs + FOLD - 1) / FOLD : (d.Wo + 31) / 32, nty = FOLD ? 1 : (d.Ho + TH - 1) / TH;
    const int Neff = FOLD ? 1 : d.N;
    const int tiles_per_img = nct * ntx * nty;
    const int total = tiles_per_img * Neff;
    const int G = gridDim.x;
    const int lslot = (G % 8 == 0) ? (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;   // XCD-contiguous runs
    const int Cin = d.C0 + d.C1;
    const int nchunk = Cin / C::KSTEP;                       // steps per tile
    const int Cr = K1 ? d.Cout / 4 : d.Cout;                 // K1: channels of an OUTPUT pixel (GEMM N = 4 sub-positions x Cr)
    // K1: GEMM column cu of channel tile ct -> (sub-position, channel of the output pixel).  Plain: cu = sp * Cr + channel.
    // S2: a 64-wide tile holds 32 channels (block cblk) of the sub-positions (dy, 0) and (dy, 1): ct = dy * (Cr / 32) + cblk,
    // column inside the tile = dx * 32 + channel % 32.
    const int cpb = S2 ? Cr / 32 : 1;
    auto k1_sp = [&](int ct, int cu) -> int {
        if constexpr (S2) return 2 * (ct / cpb) + ((cu >> 5) & 1);
        return cu / Cr;
    };
    auto k1_cb = [&](int ct, int cu) -> int {
        if constexpr (S2) return (ct % cpb) * 32 + (cu & 31);
        return cu % Cr;
    };
    const int my_sl = stid & (IPP - 1);                       // the thread's 4-channel slot of a pixel (512 % 4 == 0); ISR: its PLANE (channel half x part)
    const int my_plane = (my_sl >> 1) * PARTS * C::PLANE + (my_sl & 1) * 2;

    int in_lds[NINA];
#pragma unroll
    for (int k = 0; k < NIN; ++k) {
        const int it = stid + k * C::NTS;
        const int pt = it / PIX_ITEMS;                      // pseudo-tap (K1; 0 otherwise)
        const int pix = (it % PIX_ITEMS) / IPP;
        const int py = pix / C::IW, px = pix % C::IW;
        const int lp = (STRIDE == 2) ? py * C::TWP + (px & 1) * C::HALF + (px >> 1) : py * C::TWP + px;
        in_lds[k] = my_plane + (it < NITEM ? pt * 2 * PARTS * C::PLANE + lp * 4 : C::IH * C::TWP * 4);
        // ISR: the thread's slot is a PLANE (index my_sl = channel half x PARTS + part) and the item a whole 16-byte unit
        if constexpr (ISR) in_lds[k] = my_sl * C::PLANE + (it < NITEM ? pt * 2 * PARTS * C::PLANE + lp * 4 : C::IH * C::TWP * 4);
    }

    struct Tile {
        int ct, n, ox0, oy0;
        int fu0;                               // FOLD: the group's first sub-tile (n, ox0, oy0 are its geometry; the others': fold_geo)
        int goff[NG];                          // pixel offset into the NHWC source (-1: outside the image -> zeros)
        int goff1[K1 ? NG : 1];                // K1: pixel offset into src1, the skip tensor at the OUTPUT resolution, read at
    };                                         // the sub-position (dy, dx) this tile's channel block stores to
    // (ISP: goff / goff1 are the BYTE offsets of the thread's unit inside a plane for each of its DMA slots -- the zero unit
    // behind the plane for a pixel outside the image, -1 for a lane without a unit)
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int PS0 = yond_sp_plane_units(d.H, d.W);                                   // units per plane of src0 (and of a same-size src1)
    const int PS1 = K1 ? yond_sp_plane_units(2 * d.H, 2 * d.W) : PS0;                // K1: the skip tensor at twice the resolution
    // A cursor walks the tiles lslot, lslot + G, ...: its position is kept as the digits (n, ty, tx, ct) of the tile index
    // and advanced by adding the digits of G with carries -- scalar compares instead of four integer divisions per tile
    // (they sat in the middle of the MFMA stretch of every tile's last step).
    struct Cur { int tile, ch, ct, tx, ty, n; };
    int g_ct, g_tx, g_ty, g_n;
    {
        int b = G;
        g_ct = b % nct; b /= nct;
        g_tx = b % ntx; b /= ntx;
        g_ty = b % nty; g_n = b / nty;
    }
    auto cursor_at = [&](int t) {                // once per workgroup
        Cur c;
        c.tile = t; c.ch = 0;
        int b = t;
        c.ct = b % nct; b /= nct;
        c.tx = b % ntx; b /= ntx;
        c.ty = b % nty; c.n = b / nty;
        return c;
    };
    // (YondConvDesc.tile_order 1: the spatial digits are complemented -- the same tiles, last row of tiles first)
    const bool rev = d.tile_order != 0;
    auto tile_origin = [&](const Cur& c, Tile& T) {
        T.n = rev ? Neff - 1 - c.n : c.n;
        T.ct = c.ct;
        T.ox0 = (rev ? ntx - 1 - c.tx : c.tx) * 32;
        T.oy0 = (rev ? nty - 1 - c.ty : c.ty) * TH;
        T.fu0 = 0;
        if constexpr (FOLD != 0) {
            T.fu0 = FOLD * (rev ? ntx - 1 - c.tx : c.tx);
            const FGeo g = fold_geo(T.fu0);
            T.n = g.n; T.oy0 = g.oy0; T.ox0 = g.ox0;
        }
    };
    auto decode = [&](const Cur& c, Tile& T) {
        tile_origin(c, T);
        const int n = T.n;
        if constexpr (ISP) {
#pragma unroll
            for (int k = 0; k < NG; ++k) {
                const int sid = k * NWAVE + wave_s;             // DMA wave-slot: plane sid / WPP, units (sid % WPP) * 64 ...
                const int q = (sid % C::WPP) * 64 + lane;       // the lane's unit of the plane's LDS image
                const int py = q / C::TWP, rem = q % C::TWP;
                const int px = STRIDE == 2 ? 2 * (rem % C::HALF) + rem / C::HALF : rem;      // stride 2: even columns first
                const bool valid = q < C::UPP && px < C::IW;    // (slots past the step's last plane are refused at issue)
                int gy = K1 ? T.oy0 + py : T.oy0 * STRIDE - 1 + py, gx = K1 ? T.ox0 + px : T.ox0 * STRIDE - 1 + px;
                int nshift = 0;                                 // FOLD: byte distance of the sub-tile's image from sub-tile 0's (whose plane bases the DMA uses)
                if constexpr (FOLD != 0) {
                    const int sb = px / C::FSUBW, lx = px - C::FSUBW * sb;
                    const FGeo g = fold_geo(T.fu0 + sb);
                    gy = g.oy0 >= d.Ho ? -1 : g.oy0 - 1 + py;     // (past the last sub-tile: zeros)
                    gx = g.ox0 - 1 + lx;
                    // (a column beyond the sub-tile's that belongs to the NEXT column band is a real neighbour; rows / columns outside the image: zero)
                    nshift = (g.n - T.n) * ((d.C0 / 16) * (2 * PARTS) * PS0 * 16);
                }
                const bool in = gy >= 0 && gy < d.H && gx >= 0 && gx < d.W;
                T.goff[k] = !valid ? -1 : (in ? (gy * d.W + gx) * 16 + nshift : (d.H * d.W) * 16);
                if constexpr (K1) {
                    const int sp = k1_sp(T.ct, T.ct * TN);
                    T.goff1[k] = !valid ? -1 : (in ? (2 * gy + (sp >> 1)) * (2 * d.W) + 2 * gx + (sp & 1) : 4 * d.H * d.W) * 16;
                }
            }
        } else {
        // ([N][H][W][C]-path instantiations -- training, UNetSeeInDark, the unfused first / last layers: the per-item pixel geometry is a
        // loop invariant that the compiler hoists out of the tile loop and then SPILLS (8-31 registers, reloaded at every tile); formed
        // from an opaque copy of the thread index it is recomputed per tile instead: a few integer operations per item)
        int tid_g = stid;
        if constexpr (ISPM == 0 && !OSP && !K1) asm volatile("" : "+v"(tid_g));
#pragma unroll
        for (int k = 0; k < KD; ++k) {
            const int it = tid_g + k * C::NTS;
            const int pix = (it % PIX_ITEMS) / IPP;
            const int py = pix / C::IW, px = pix % C::IW;
            int gy = K1 ? T.oy0 + py : T.oy0 * STRIDE - 1 + py, gx = K1 ? T.ox0 + px : T.ox0 * STRIDE - 1 + px;
            int nb = n * d.H * d.W;                                // the image's first pixel
            int ushift = 0, ushift1 = 0;                           // ISR + FOLD: units from sub-tile 0's image's planes to the sub-tile's image's (src0, src1)
            if constexpr (FOLD != 0) {
                constexpr int SUBW = K1 ? FSW : C::FSUBW;
                const int sb = px / SUBW, lx = px - SUBW * sb;
                const FGeo g = fold_geo(T.fu0 + sb);
                gy = g.oy0 >= d.Ho ? -1 : (K1 ? g.oy0 + py : g.oy0 * STRIDE - 1 + py);     // (past the last sub-tile: zeros)
                gx = K1 ? g.ox0 + lx : g.ox0 * STRIDE - 1 + lx;
                ushift1 = (g.n - n) * (d.C1 / 16) * (2 * PARTS) * PS1;
                // the sub-tile's image: [N][H][W][C]: its pixels; planes of 4 channels: the plane base (load_src) is sub-tile 0's image's, C0/4 planes per image
                nb = d.in_fmt == YOND_FMT_PLANES4 ? nb + (g.n - n) * (d.C0 / 4) * d.H * d.W : g.n * d.H * d.W;
                ushift = (g.n - n) * (d.C0 / 16) * (2 * PARTS) * PS0;
            }
            const bool ok = it < NITEM && gy >= 0 && gy < d.H && gx >= 0 && gx < d.W;
            T.goff[k] = ok ? (nb + gy * d.W + gx) : -1;
            if constexpr (ISR) T.goff[k] = ok ? gy * d.W + gx + ushift : d.H * d.W;      // unit inside a plane (n is in the plane base); outside: the zero unit
            if constexpr (K1) {
                const int sp = k1_sp(T.ct, T.ct * TN);         // a channel tile never straddles two sub-positions (S2: dx = 0 here, + 1 unit for dx = 1)
                T.goff1[k] = ok ? ((n * 2 * d.H + 2 * gy + (sp >> 1)) * (2 * d.W) + 2 * gx + (sp & 1)) : -1;
                if constexpr (ISR) T.goff1[k] = ok ? (2 * gy + (sp >> 1)) * (2 * d.W) + 2 * gx + (sp & 1) + ushift1 : 4 * d.H * d.W;
            }
        }
        }
    };

    // Three register sets: set s % 3 receives the loads of input(s+3) during step s and is split / written out as
    // input(s+3) during step s+2, so a load has two steps to arrive.
    // (three rows per wave: 96 accumulator registers leave room for two sets; likewise the h-only stride-2 layer on 128-channel tiles -- 64 accumulator
    // registers, two weight-fragment sets of two blocks: with three sets it spills five registers)
    // (ROLES: ONE set -- the moving half loads input(s+1) and stages it within step s: it has nothing else to do while the other half multiplies, and a second set
    // would not fit beside the accumulators: both roles' registers are allocated together)
    constexpr int NSET = ROLES ? 1 : (MW >= 3 || (STRIDE == 2 && PARTS == 1 && TN == 128)) ? 2 : 3;
    static_assert(NSET == 3 || C::WAHEAD == 1, "the two-set pipeline goes with two weight buffers");
    f32x4 vin[NSET][NINA];
    unsigned vin_ok[NSET] = {};
    float amax = 0.0f;                                         // largest |activation| this thread has staged (range guard)
    // one 16-byte load of a set (item k); the source of the chunk is selected once per step (LoadSrc)
    // source of a step's 16-channel chunks (K1: three of them, each from the low-resolution input or from the skip tensor)
    // (address of an item = src + pixel offset * A + B: [N][H][W][C] float32: A = C, B = chunk channel + 4 slot; planes of 4
    // channels, YOND_FMT_PLANES4 [N][C/4][H*W][4]: A = 4, B = 4 (plane index * H*W - n H*W): the pixel offset carries n H*W)
    struct LoadSrc { const float* src[C::NPT]; int A[C::NPT], B[C::NPT]; bool hi[C::NPT]; int dx[C::NPT]; };     // (B: 32 bits -- the dispatcher checks the extent)
    const bool in_p4 = d.in_fmt == YOND_FMT_PLANES4;
    auto load_src = [&](int ch, int n) {
        LoadSrc L;
#pragma unroll
        for (int t = 0; t < C::NPT; ++t) {
            const int c0 = (ch * C::NPT + t) * C::KC;
            const bool second = c0 >= d.C0;
            int Cs = second ? d.C1 : d.C0, cc = second ? c0 - d.C0 : c0;
            L.src[t] = second ? d.src1 : d.src0;
            L.hi[t] = second && K1;
            L.dx[t] = 0;
            if constexpr (S2) {
                if (second) {
                    // [real channels at dx = 0 | the same at dx = 1 | zero-weight chunks up to a multiple of 48 (any finite data: dx = 0, channel 0)]
                    const int creal = Cr;                      // (the skip tensor has the output pixels' channel count)
                    const int q = cc / creal;
                    L.dx[t] = q == 1 ? 1 : 0;
                    cc = q < 2 ? cc - q * creal : 0;
                    Cs = creal;
                }
            }
            if constexpr (ISR) {
                // split planes: unit index inside the plane * 4 floats + the base of plane (n, chunk, channel half, part = the slot)
                L.A[t] = 4;
                L.B[t] = ((n * (Cs / 16) + cc / 16) * (2 * PARTS) + my_sl) * (L.hi[t] ? PS1 : PS0) * 4;
            } else if (in_p4) {
                const int hw = d.H * d.W * (L.hi[t] ? 4 : 1);
                L.A[t] = 4;
                L.B[t] = 4 * ((n * (Cs / 4) + cc / 4 + my_sl) * hw - n * hw);
            } else {
                L.A[t] = Cs;
                L.B[t] = cc + my_sl * 4;
            }
        }
        return L;
    };
    auto issue_load = [&](auto pc, auto kc, const Tile& T, const LoadSrc& L) {
        constexpr int P = decltype(pc)::value, k = decltype(kc)::value;
        constexpr int t = K1 ? (k * C::NTS) / PIX_ITEMS : 0;  // the item's chunk (K1: PIX_ITEMS is a multiple of the thread count)
        constexpr int kd = k % KD;                             // (K1: the item's pixel)
        const bool ok = T.goff[kd] >= 0;                       // outside the image: read pixel 0, zeroed at the LDS write
        int po = T.goff[kd];
        if constexpr (K1) po = L.hi[t] ? T.goff1[kd] : po;
        if constexpr (S2) {
            // the neighbouring output pixel (dx = 1) is the next unit; the zero unit of an outside pixel stays where it is
            if (L.hi[t] && L.dx[t] && T.goff1[kd] != 4 * d.H * d.W) po += 1;
        }
        vin[P][k] = *(const f32x4*)(L.src[t] + (long long)(ok ? po : 0) * L.A[t] + L.B[t]);
        if (k == 0) vin_ok[P] = 0;
        vin_ok[P] |= (ok ? 1u : 0u) << k;
    };
    // weight slice global -> LDS by LDS-DMA as inline assembly (a builtin DMA makes the compiler order every later LDS
    // access behind s_waitcnt vmcnt(0)); one instruction (k) moves 512 x 16 bytes; completion is awaited by a barrier
    // Resident weights: a layer with two weight slices in all (32 -> 32 channels: two 16-channel chunks, one channel tile) on the
    // two-buffer rotation finds slice s % 2 in buffer s % 2 for ever -- after the first step no weight DMA is issued at all (they
    // were a third of the bytes such a layer's steps fetch into LDS).
    // (not where the transposed epilogue uses the consumed weight buffer as its scratch: 64-wide stride-1 tiles storing [N][H][W][C])
    constexpr bool W_IS_SCRATCH = TN >= 64 && !K1 && STRIDE == 1 && !OSP;
    const bool wres = NWB == 2 && !W_IS_SCRATCH && nchunk == 2 && nct == 1;
    bool wskip = false;                                      // true once both slices have been fetched (uniform)
    auto issue_dma = [&](auto kc, const float* wsrc, float* wbuf) {
        constexpr int k = decltype(kc)::value;
        if (wskip) return;
        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)wbuf;
        const int it = stid + k * C::NTS;
        const int it_wave = __builtin_amdgcn_readfirstlane(it - lane);
        const unsigned lds_wave = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)it_wave * 16u);
        const unsigned voff = (unsigned)it * 16u;
        if (C::NWV % C::NTS == 0 || it_wave < C::NWV)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_wave), "v"(voff), "s"(wsrc) : "memory");
    };
    // ISP: the planes of a step's chunk(s), wave-uniform: [n][chunk][half][part][PS units]; DMA slot k of this wave moves the
    // 64 units (sid % WPP) * 64 ... of plane sid / WPP, sid = 8 k + wave
    struct InSrc { const char* base[C::NPT]; unsigned hi; };
    auto in_src = [&](int ch, int n) {
        InSrc I;
        I.hi = 0;
#pragma unroll
        for (int t = 0; t < C::NPT; ++t) {
            const int c0 = (ch * C::NPT + t) * C::KC;
            const bool second = c0 >= d.C0;
            const bool hi = K1 && second;
            const int c16 = (second ? c0 - d.C0 : c0) / 16, nc16 = (second ? d.C1 : d.C0) / 16;
            const char* src = (const char*)(second ? d.src1 : d.src0);
            const unsigned long long a = (unsigned long long)(uintptr_t)(src + (size_t)(n * nc16 + c16) * (2 * PARTS) * (size_t)(hi ? PS1 : PS0) * 16);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi32 = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
            I.base[t] = (const char*)(uintptr_t)(((unsigned long long)hi32 << 32) | lo);
            I.hi |= (hi ? 1u : 0u) << t;
        }
        return I;
    };
    auto issue_in_dma = [&](auto kc, const Tile& T, const InSrc& I, float* ob) {
        constexpr int k = decltype(kc)::value;
        const int sid = k * NWAVE + wave_s;
        const int plane = sid / C::WPP, wv = sid - plane * C::WPP;
        const int pt = plane / (2 * PARTS), pl = plane - pt * (2 * PARTS);
        const bool hi = K1 && ((I.hi >> pt) & 1u);
        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)ob;
        const unsigned lds_wave = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(plane * C::PLANE * 4 + wv * 1024));
        const char* b0 = I.base[0];
        if constexpr (C::NPT > 1) b0 = pt == 1 ? I.base[1] : (pt == 2 ? I.base[C::NPT - 1] : b0);
        const unsigned long long ga = (unsigned long long)(uintptr_t)(b0 + (size_t)pl * (size_t)(hi ? PS1 : PS0) * 16);
        // (wave-uniform: the DMA takes it in SGPRs.  readfirstlane returns int: through unsigned, or the low half sign-extends)
        const unsigned ga_lo = __builtin_amdgcn_readfirstlane((unsigned)ga), ga_hi = __builtin_amdgcn_readfirstlane((unsigned)(ga >> 32));
        const char* gb = (const char*)(uintptr_t)(((unsigned long long)ga_hi << 32) | ga_lo);
        int voff = T.goff[k % NG];
        if constexpr (K1) voff = hi ? T.goff1[k % NG] : voff;
        if (sid < C::NDS && voff >= 0)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_wave), "v"(voff), "s"(gb) : "memory");
    };
    auto weight_src = [&](int ct, int ch) {                  // wave-uniform: handed to the DMA in scalar registers
        const unsigned long long a = (unsigned long long)(uintptr_t)(d.wpk + ((size_t)ct * nchunk + ch) * C::W_FLOATS);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        return (const float*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
    };
    // Staging of one register set into an input image, as NE = 4 NIN element tasks so that the step can spread them
    // between its MFMAs: task e = (item k, element j) applies SiLU / zero padding in place; the item's last task splits
    // the four values into the h and l halves and writes them (two ds_write_b64).
    constexpr int NE = 4 * NIN;
    auto stage_task = [&](auto pc, auto ec, float* ob) {
        constexpr int P = decltype(pc)::value;
        constexpr int e = decltype(ec)::value, k = e / 4, j = e % 4;
        if constexpr (ISR) {
            // the unit is already what the LDS image holds (zero unit outside the image): one 16-byte store per item
            if constexpr (j == 3) *(f32x4*)(ob + in_lds[k]) = vin[P][k];
            return;
        }
        float x = vin[P][k][j];
        if (PRE) x = split_silu(x);
        vin[P][k][j] = ((vin_ok[P] >> k) & 1u) ? x : 0.0f;                              // conv zero padding
        if constexpr (j == 3) {
            const f32x4 v = vin[P][k];
            amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));   // two v_max3
            const f16x4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            *(f16x4*)(ob + in_lds[k]) = h;
            if constexpr (PARTS == 2) {
                const f16x4 l = {(_Float16)((v[0] - (float)h[0]) * 2048.0f), (_Float16)((v[1] - (float)h[1]) * 2048.0f),
                                 (_Float16)((v[2] - (float)h[2]) * 2048.0f), (_Float16)((v[3] - (float)h[3]) * 2048.0f)};
                *(f16x4*)(ob + in_lds[k] + C::PLANE) = l;
            }
        }
    };
    auto write_in = [&](auto pc, float* ob) { static_for<0, NE>([&](auto ec) { stage_task(pc, ec, ob); }); };

    f32x16 acc[NACC][MW][C::NW];
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < NACC; ++a)
#pragma unroll
            for (int m = 0; m < MW; ++m)
#pragma unroll
                for (int nn = 0; nn < C::NW; ++nn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][m][nn][r] = 0.0f;
    };

    // ---- the multiplications of one step ----
    // The wave's MW output rows read input rows r = 0 .. (MW-1) stride + 2; fragment X(r, dx) serves every (m, dy) with
    // m stride + dy = r, so it is read ONCE per step (3 (MW+2) pixel fragments per part instead of 9 MW): the kernel is
    // LDS-bandwidth bound otherwise (1 KiB of fragments per 32-cycle MFMA and wave).  Loop: dx outermost, the three
    // weight fragments (dy) of a column held in registers and fetched one column ahead, pixel fragments two ahead.
    // MFMA order inside a group: h_w l_x, then h_w h_x, then l_w h_x -- the two that share an accumulator are never
    // back to back (the dependent-issue latency of v_mfma_f32_32x32x16_f16 exceeds its 32 cycles).
    const int x_off = (lh * PARTS) * C::PLANE + ((rg * MW * (K1 ? 1 : STRIDE)) * C::TWP + flds) * 4;
    const int w_off = ((lh * PARTS) * TN + (cg * C::NW) * 32 + li) * 4;
    // (the step's cursor arithmetic -- `prep`, which sets wsrc_s / ls_s and may decode the next tile -- runs after the
    // first groups of MFMAs have been issued; memory instructions and staging start at group Q0)
    const float* wsrc_s = nullptr;
    const bool wave_hi = NWAVE == 8 && __builtin_amdgcn_readfirstlane(wave) >= 4;       // (one wave per SIMD: no partner to arbitrate with)
    LoadSrc ls_s = {};
    InSrc is_s = {};
    int dbg_step = 0;
    (void)dbg_step;
    auto mfma_stage = [&](auto pc, auto fc, const float* buf, const float* wbuf, float* ob, float* wnext, const Tile& lt, auto&& prep)
        __attribute__((always_inline)) {
        typedef const __attribute__((address_space(3))) f16x8* lds_h8;
        const __attribute__((address_space(3))) float* xb = (const __attribute__((address_space(3))) float*)(buf + x_off);
        const __attribute__((address_space(3))) float* wb = (const __attribute__((address_space(3))) float*)(wbuf + w_off);
        // K1: the three 'columns' are the step's three 16-channel chunks, a row r serves output row m = r only
        constexpr int R = K1 ? MW : (MW - 1) * STRIDE + 3, NQ = 3 * R, XD = 3, Q0 = 2, NQW = NQ - Q0;
        // weight fragments of a column: two sets (the next column is fetched during the current one), or -- three rows per
        // wave, register budget -- ONE set, each dy refilled for the next column right behind its last use (two groups ahead
        // of its next use)
        constexpr bool WINPLACE = MW >= 3 || ROLES != 0;      // (ROLES: the register budget holds both roles)
        constexpr int WS = WINPLACE ? 1 : 2;
        f16x8 xr[XD][PARTS], wt[WS][3][C::NW][PARTS];
        auto loadX = [&](auto qc) {
            constexpr int q = decltype(qc)::value;
            constexpr int dx = q / R, r = q % R;
            constexpr int xo = K1 ? 0 : (STRIDE == 2) ? (dx & 1) * C::HALF + (dx >> 1) : dx;
            constexpr int po = K1 ? dx * 2 * PARTS * C::PLANE : 0;         // K1: the chunk's planes
#pragma unroll
            for (int p = 0; p < PARTS; ++p) xr[q % XD][p] = *(lds_h8)(xb + po + p * C::PLANE + (r * C::TWP + xo) * 4);
        };
        auto loadW1 = [&](auto dc, auto yc) {
            constexpr int dx = decltype(dc)::value, dy = decltype(yc)::value;
            constexpr int tap = K1 ? dx : dy * 3 + dx;
#pragma unroll
            for (int nn = 0; nn < C::NW; ++nn)
#pragma unroll
                for (int p = 0; p < PARTS; ++p)
                    wt[dx % WS][dy][nn][p] = *(lds_h8)(wb + ((tap * 2 * PARTS + p) * TN + nn * 32) * 4);
        };
        auto loadW = [&](auto dc) { static_for<0, (K1 ? 1 : 3)>([&](auto yc) { loadW1(dc, yc); }); };
        loadW(IntC<0>{});
        loadX(IntC<0>{});
        loadX(IntC<1>{});
        static_for<0, NQ>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            constexpr int dx = q / R, r = q % R;
            constexpr bool wpre = !WINPLACE && (r == (R >= 4 ? R - 3 : 0) && dx < 2);
            constexpr int wdy = r - (MW - 1) * STRIDE;                      // WINPLACE: the tap row whose last use is this group
            constexpr bool wrep = WINPLACE && dx < 2 && wdy >= 0 && wdy <= 2;
            if constexpr (q + 2 < NQ) loadX(IntC<q + 2>{});
            if constexpr (wpre) loadW(IntC<dx + 1>{});
            constexpr int nmf = (PARTS == 2 ? 3 : 1) * C::NW * (K1 ? 1 : split_pairs(MW, STRIDE, r));
#pragma unroll
            for (int a = 0; a < 3; ++a) {                  // 0: h_w l_x   1: h_w h_x   2: l_w h_x
                if (PARTS == 1 && a != 1) continue;
#pragma unroll
                for (int m = 0; m < MW; ++m) {
                    const int dy = K1 ? (m == r ? 0 : -1) : r - m * STRIDE;
                    if (dy < 0 || dy > 2) continue;
#pragma unroll
                    for (int nn = 0; nn < C::NW; ++nn)
                        acc[a == 1 ? 0 : 1][m][nn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                            wt[dx % WS][dy][nn][a == 2 ? PARTS - 1 : 0], xr[q % XD][a == 0 ? PARTS - 1 : 0], acc[a == 1 ? 0 : 1][m][nn], 0, 0, 0);   // D = W . X^T
                }
            }
            if constexpr (wrep) loadW1(IntC<dx + 1>{}, IntC<wdy>{});          // behind this group's MFMAs (its last readers)
            // the older wave of a SIMD wins the issue arbitration and would finish its MFMAs ~900 cycles before the younger
            // one, which then runs alone and exposes its LDS latencies: the younger half leads for the first half instead
            if constexpr (q == 0) { if (wave_hi) __builtin_amdgcn_s_setprio(1); }
            if constexpr (q == NQ / 2) { if (wave_hi) __builtin_amdgcn_s_setprio(0); }
            if constexpr (q == Q0 - 1 && !ROLES) prep();
            if constexpr (q == NQ / 4) SDBG(8);
            if constexpr (q == NQ / 2) SDBG(9);
            if constexpr (q == (3 * NQ) / 4) SDBG(10);
            // this group's share of the step's vector-memory instructions (loads of input(s+3) / LDS-DMA of the weights) --
            // spread over the step so that no wave ever queues behind the CU's 64 B/clk memory pipe
            constexpr int qq = q >= Q0 ? q - Q0 : 0;
            // (split-plane input with two weight buffers: the end-of-step barrier waits for EVERYTHING this step issued, so all of
            // it goes into the first half of the step)
            constexpr int NQO = (ISP && C::WAHEAD == 1) ? (NQW + 1) / 2 : NQW;
            constexpr int qo = qq < NQO ? qq : NQO;
            // (ROLES: the multiplying waves issue no memory operation and stage nothing)
            constexpr int o_lo = (q >= Q0 && !ROLES) ? (qo * NOPS + NQO - 1) / NQO : 0, o_hi = (q >= Q0 && !ROLES) ? ((qo + 1 < NQO ? qo + 1 : NQO) * NOPS + NQO - 1) / NQO : 0;
            static_for<o_lo, o_hi>([&](auto oc) {
                constexpr int o = decltype(oc)::value;
                if constexpr (ISP) {
                    // input(s+1) first (it has the whole step to land), then the weights
                    if constexpr (o < NDI) issue_in_dma(IntC<o>{}, lt, is_s, ob);
                    else issue_dma(IntC<o - NDI>{}, wsrc_s, wnext);
                } else if constexpr (C::LOADS_FIRST) {
                    if constexpr (o < NIN) issue_load(fc, IntC<o>{}, lt, ls_s);
                    else issue_dma(IntC<o - NIN>{}, wsrc_s, wnext);
                } else {
                    if constexpr (o < C::NWT) issue_dma(IntC<o>{}, wsrc_s, wnext);
                    else issue_load(fc, IntC<o - C::NWT>{}, lt, ls_s);
                }
            });
            // this group's share of the staging work (element tasks e_lo .. e_hi of the set loaded during the previous step)
            constexpr int e_lo = (q >= Q0 && !ROLES) ? qq * NE / NQW : 0, e_hi = (q >= Q0 && !ROLES) ? (qq + 1) * NE / NQW : 0;
            static_for<e_lo, e_hi>([&](auto ec) { stage_task(pc, ec, ob); });
            constexpr int nfin = (e_hi + 0) / 4 - (e_lo + 0) / 4;                     // items completed in this group
            // Issue order of the group: its LDS reads first (they are two groups / one column ahead of their use), then
            // each MFMA followed by a few of the vector instructions, then the LDS writes of a completed item.
            constexpr int nrd = (q + 2 < NQ ? PARTS : 0) + (wpre ? (K1 ? 1 : 3) * C::NW * PARTS : 0);
            if constexpr (nrd > 0) __builtin_amdgcn_sched_group_barrier(0x100, nrd, 0);
            // The scheduler's VALU class (0x002) does not hold the transcendental instructions: v_exp_f32 / v_rcp_f32 of SiLU are class 0x400 and used to land
            // where they might -- up to nine in one MFMA gap, 79 VALU + 7 of them behind the last MFMA of a level-0 step.  Each gap takes its share of them
            // explicitly (round 6: the two level-0 conv1 launches -3 %, everything else unchanged, bit-identical; profiles/r06_experiments/sched_trans_ab.txt)
            constexpr int ntr = (!ISR && PRE) ? (e_hi - e_lo) * 2 : 0;
            constexpr int nvalu = ISR ? 0 : (e_hi - e_lo) * (PRE ? 5 : 2) + nfin * (PARTS == 2 ? 22 : 6);
            constexpr int vpm = (nvalu + nmf - 1) / nmf;
#pragma unroll
            for (int i = 0; i < nmf; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if ((i * ntr) / nmf != ((i + 1) * ntr) / nmf) __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);
                if constexpr (vpm > 0) __builtin_amdgcn_sched_group_barrier(0x002, vpm, 0);
            }
            if constexpr (wrep) __builtin_amdgcn_sched_group_barrier(0x100, C::NW * PARTS, 0);
            if constexpr (nfin > 0) __builtin_amdgcn_sched_group_barrier(0x200, ISR ? nfin : nfin * PARTS, 0);
        });
    };

    // ---- output side ----
    // In the accumulators a lane owns ONE pixel (li) and, per 32-channel block, channels (r&3) + 8 (r>>2) + 4 lh: stored
    // from there, every instruction would touch 32 cache lines with 32 bytes each (and the residual loads likewise) --
    // measured, that request rate made the epilogue 4-6 thousand cycles per tile.  So each wave transposes its
    // 32-pixel x 32-channel blocks through a private LDS scratch (pixel stride 36 floats: conflict-free both ways) and
    // reads them back with 8 lanes per pixel: a global instruction then covers 8 whole 128-byte lines, FiLM vectors are
    // one float4 per lane, and nothing is shared between waves (no barrier inside).
    const float slope_eff = d.post_act == 2 ? d.slope : 1.0f;
    // post_act 1: SiLU of the stored value -- the block-internal tensor of a residual block has ONE consumer, whose staging
    // would otherwise apply the same SiLU once per output-channel tile (Cout / 64 times at the deeper levels)
    const bool silu_out = d.post_act == 1;
    constexpr int EPS = 36;                                   // floats per pixel of the scratch
    constexpr int EP_FLOATS = NWAVE * 32 * EPS;               // 36,864 bytes: one weight buffer (TN 64) or part of an input image
    // (stride 2: short steps, no residual -- measured slower with the transpose and its extra barrier: 211 vs 175 us at level 0)
    constexpr bool EP_IN_W = TN >= 64 && !K1;                 // scratch = the weight buffer just consumed; else the input image
    constexpr bool EP_FIT = STRIDE == 1 && (C::EP_OWN || EP_FLOATS <= (EP_IN_W ? C::W_FLOATS : C::IN_FLOATS));
    // Straight-line on purpose: a branch around a load (`res ? load : 0`) makes the compiler lose count of the outstanding
    // memory operations and wait with vmcnt(0) before EVERY store -- i.e. for the previous store (measured: 3.4-4.9
    // thousand cycles per tile).  So the variant (residual / scale / shift present) is chosen by ONE uniform switch
    // outside, and all loads precede the first store.
    f32x4 pes[C::NW], pet[C::NW];
    auto epi_prefetch = [&](const Tile& T) __attribute__((always_inline)) {
        const int u = lane & 7;
#pragma unroll
        for (int nn = 0; nn < C::NW; ++nn) {
            const int cu = T.ct * TN + (cg * C::NW + nn) * 32 + 4 * u;
            int eoff = (d.ebatch ? T.n * Cr : 0) + (K1 ? k1_cb(T.ct, cu) : cu);
            if constexpr (FOLD != 0) eoff = (d.ebatch ? fold_geo(T.fu0 + (lane >> 4)).n * Cr : 0) + cu;     // lanes 16 s ... 16 s + 15: sub-tile s's image
            // (no branch around the loads: an absent vector is read from the weights and never used)
            if constexpr (OSP) {
                // the split-plane epilogue redistributes the vectors through LDS: lanes 0-7 fetch the scale, lanes 8-15 the shift,
                // ONE register quad per block instead of two
                const float* src = (lane & 8) ? d.eshift : d.escale;
                pes[nn] = *(const f32x4*)(src ? src + eoff : d.wpk);
            } else {
                pes[nn] = *(const f32x4*)(d.escale ? d.escale + eoff : d.wpk);
                pet[nn] = *(const f32x4*)(d.eshift ? d.eshift + eoff : d.wpk);
            }
        }
    };
    // Residual prefetch (split-plane input kernels: their steps need no staging registers).  In-kernel stamps showed the
    // transposed epilogue waiting 10-14 thousand cycles per tile for its residual loads (two dependent round trips to HBM per
    // tile, every wave of the CU at once): the loads of the first two rows are issued at the START of the tile's last step
    // and land under its MFMAs; a third row is requested when the epilogue begins, ahead of the first rows' arithmetic.
    constexpr bool RPF = ISP && !OSP && EP_FIT && !K1 && C::NW == 1;
    constexpr int PFR = MW < 2 ? MW : (C::NW >= 2 ? 1 : 2);    // (two blocks per wave -- the h-only flow's 128-channel tiles: one prefetched row is what the registers hold)
    f32x4 prr[RPF ? PFR : 1][4];
    float pxq[RPF && O4 ? PFR : 1][4];
    auto res_addr = [&](const Tile& T, int m, int j) -> const float* {      // residual of pixel pj + 8 j of the wave's row m, channels 4 u ..
        const int pj = lane >> 3, u = lane & 7;
        const int oy = T.oy0 + rg * MW + m;
        const bool ok = d.res && oy < d.Ho && T.ox0 + pj + 8 * j < d.Wo;
        const long long off = ((long long)(T.n * d.Ho + oy) * d.Wo + T.ox0 + pj + 8 * j) * d.Cout + T.ct * TN + cg * C::NW * 32 + 4 * u;
        return (d.res ? d.res : d.wpk) + (ok ? off : 0);                     // (no branch around the load: see the epilogue)
    };
    auto xq_addr = [&](const Tile& T, int m, int j) -> const float* {       // O4: component u & 3 of the pixel's network input
        const int pj = lane >> 3, u = lane & 7;
        const int oy = T.oy0 + rg * MW + m;
        const bool ok = d.out4_x && oy < d.Ho && T.ox0 + pj + 8 * j < d.Wo;
        const long long gp = (long long)(T.n * d.Ho + oy) * d.Wo + T.ox0 + pj + 8 * j;
        return (d.out4_x ? d.out4_x : d.wpk) + (ok ? gp * 4 + (u & 3) : 0);
    };
    auto res_prefetch = [&](const Tile& T) __attribute__((always_inline)) {
        if constexpr (RPF) {
#pragma unroll
            for (int mm = 0; mm < PFR; ++mm)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    prr[mm][j] = *(const f32x4*)res_addr(T, mm, j);
                    if constexpr (O4) pxq[mm][j] = *xq_addr(T, mm, j);
                }
        }
    };
    // (ACT: the activation as a compile-time constant for the combinations the networks use -- 0 none, 1 SiLU, 2 LeakyReLU --
    // or -1: decided by the descriptor at run time, which costs the arithmetic of BOTH activations plus two selects per element)
    auto epilogue = [&](auto hr, auto actc, const Tile& T, float* scratch) __attribute__((always_inline)) {
        constexpr bool HAS_RES = (decltype(hr)::value & 1) != 0, HAS_SCALE = (decltype(hr)::value & 2) != 0, HAS_SHIFT = (decltype(hr)::value & 4) != 0;
        constexpr int ACT = decltype(actc)::value;
        constexpr bool OUT4 = O4;                                            // fused 1x1 output projection (see YondConvDesc)
        f32x4 lrr[RPF && MW > PFR ? MW - PFR : 1][4];                        // RPF: the rows beyond the prefetched ones, requested first
        float lxq[RPF && O4 && MW > PFR ? MW - PFR : 1][4];
        if constexpr (RPF && MW > PFR) {
#pragma unroll
            for (int mm = PFR; mm < MW; ++mm)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    lrr[mm - PFR][j] = *(const f32x4*)res_addr(T, mm, j);
                    if constexpr (O4) lxq[mm - PFR][j] = *xq_addr(T, mm, j);
                }
        }
        float* sw = scratch + wave * (32 * EPS);
        const int pj = lane >> 3, u = lane & 7;               // read-back: pixel pj + 8 j, channels 4 u .. 4 u + 3
        FGeo jg[FOLD ? 4 : 1];                                // FOLD: (image, first row, column) of the read-back pixel pj + 8 j -- its sub-tile's geometry
        if constexpr (FOLD != 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                jg[j] = fold_geo(T.fu0 + (pj + 8 * j) / FSW);
                jg[j].ox0 += (pj + 8 * j) % FSW;
            }
        }
        float ubv = 1.0f, bq = 0.0f;                          // OUT4: per-image maximum, bias of output component u & 3
        if constexpr (OUT4) {
            if (d.out4_ub) ubv = d.out4_ub[T.n];
            if (d.out4_b) bq = d.out4_b[u & 3];
        }
        static_for<0, C::NW>([&](auto nc) __attribute__((always_inline)) {
            constexpr int nn = decltype(nc)::value;
            const int cu = T.ct * TN + (cg * C::NW + nn) * 32 + 4 * u;
            // K1: GEMM channel cu = sub-position sp x Cr + channel; the pixel goes to (2 y + sp/2, 2 x + sp%2) of the output
            const int sp = K1 ? k1_sp(T.ct, cu) : 0, cb = K1 ? k1_cb(T.ct, cu) : cu;
            const int pstep = K1 ? 2 * Cr : d.Cout;              // elements between horizontally adjacent pixels of the tile
            const int eoff = (d.ebatch ? T.n * Cr : 0) + cb;
            const f32x4 one = {1.0f, 1.0f, 1.0f, 1.0f}, zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
            (void)eoff;
            f32x4 es = one, et = zero4;                        // FiLM / bias vectors: requested at the start of the tile's last step
            if constexpr (HAS_SCALE) es = pes[nn];
            if constexpr (HAS_SHIFT) et = pet[nn];
            // two rows at a time: their residual loads first, then transposes / arithmetic / stores
            constexpr int MG = MW < 2 ? MW : 2;
            static_for<0, (MW + MG - 1) / MG>([&](auto gcx) __attribute__((always_inline)) {
                constexpr int m0 = decltype(gcx)::value * MG;
                constexpr int MGN = MW - m0 < MG ? MW - m0 : MG;           // rows of this group
                f32x4 rr[MG][4];
                float xq[OUT4 ? MG : 1][OUT4 ? 4 : 1];                      // OUT4: component u & 3 of the pixel's network input (global residual)
                long long rowoff[MG];
#pragma unroll
                for (int mm = 0; mm < MGN; ++mm) {
                    const int oy = T.oy0 + rg * MW + m0 + mm;
                    if constexpr (K1) rowoff[mm] = ((long long)(T.n * 2 * d.Ho + 2 * oy + (sp >> 1)) * (2 * d.Wo) + 2 * T.ox0 + (sp & 1)) * Cr + cb;
                    else rowoff[mm] = ((long long)(T.n * d.Ho + oy) * d.Wo + T.ox0) * d.Cout + cb;
                    if constexpr (RPF) {
                        // (prefetched at the start of the tile's last step / requested at the top of the epilogue)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if constexpr (m0 == 0) {
                                rr[mm][j] = HAS_RES ? prr[mm][j] : zero4;
                                if constexpr (OUT4) xq[mm][j] = pxq[mm][j];
                            } else {
                                rr[mm][j] = HAS_RES ? lrr[m0 - PFR + mm][j] : zero4;
                                if constexpr (OUT4) xq[mm][j] = lxq[m0 - PFR + mm][j];
                            }
                        }
                        continue;
                    }
                    if constexpr (OUT4) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const bool ok = oy < d.Ho && T.ox0 + pj + 8 * j < d.Wo && d.out4_x;
                            const long long gp = (long long)(T.n * d.Ho + oy) * d.Wo + T.ox0 + pj + 8 * j;
                            xq[mm][j] = *((d.out4_x ? d.out4_x : d.wpk) + (ok ? gp * 4 + (u & 3) : 0));
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        bool ok = oy < d.Ho && T.ox0 + pj + 8 * j < d.Wo;      // masked lanes read element 0..
                        long long off = rowoff[mm] + (long long)(pj + 8 * j) * pstep;
                        if constexpr (FOLD != 0) {
                            const int oyj = jg[j].oy0 + rg * MW + m0 + mm;
                            ok = oyj < d.Ho && jg[j].ox0 < d.Wo;
                            off = ((long long)(jg[j].n * d.Ho + oyj) * d.Wo + jg[j].ox0) * d.Cout + cb;
                        }
                        if constexpr (HAS_RES) rr[mm][j] = *(const f32x4*)(d.res + (ok ? off : 0));
                        else rr[mm][j] = zero4;
                    }
                }
#pragma unroll
                for (int mm = 0; mm < MGN; ++mm) {
                    const int m = m0 + mm;
                    const int oy = T.oy0 + rg * MW + m;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = acc[0][m][nn][4 * g + e];
                            if constexpr (PARTS == 2) v[e] = fmaf(acc[1][m][nn][4 * g + e], 1.0f / 2048.0f, v[e]);
                        }
                        *(f32x4*)(sw + li * EPS + 8 * g + 4 * lh) = v;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        bool ok = oy < d.Ho && T.ox0 + pj + 8 * j < d.Wo;
                        long long soff = rowoff[mm] + (long long)(pj + 8 * j) * pstep;
                        if constexpr (FOLD != 0) {
                            const int oyj = jg[j].oy0 + rg * MW + m;
                            ok = oyj < d.Ho && jg[j].ox0 < d.Wo;
                            soff = ((long long)(jg[j].n * d.Ho + oyj) * d.Wo + jg[j].ox0) * d.Cout + cb;
                        }
                        const f32x4 x = *(const f32x4*)(sw + (pj + 8 * j) * EPS + 4 * u);
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float y = fmaf(x[e], es[e], et[e]);
                            if constexpr (ACT == 2 || ACT < 0) y = y > 0.0f ? y : y * slope_eff;
                            if constexpr (ACT == 1) y = split_silu(y);
                            if constexpr (ACT < 0) { if (silu_out) y = split_silu(y); }
                            v[e] = y + rr[mm][j][e];
                        }
                        if constexpr (OUT4) {
                            // 1x1 projection 32 -> 4 in yond_conv_out_f32's operation order: per lane four FMAs per output over its
                            // channels 4u..4u+3, xor-shuffle tree over the pixel's 8 lanes, then bias, residual, de-normalisation
                            const float* w4 = smem + C::W4_OFF + 4 * u;
                            float o[4];
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const f32x4 wv = *(const f32x4*)(w4 + 32 * c);
                                float t = 0.0f;
#pragma unroll
                                for (int e = 0; e < 4; ++e) t = fmaf(v[e], wv[e], t);
                                o[c] = t;
                            }
                            // xor 1, xor 2 (quad permutes), then the other half of the 8 lanes (half-row mirror: every lane of a
                            // quad holds the same sum by then) -- DPP instead of 12 dependent ds_bpermute round trips
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                o[c] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, o[c]), 0xB1, 0xF, 0xF, true));
                                o[c] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, o[c]), 0x4E, 0xF, 0xF, true));
                                o[c] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, o[c]), 0x141, 0xF, 0xF, true));
                            }
                            // every lane of the pixel now holds all four sums: lane u < 4 finishes and stores component u
                            const int cu = u & 3;
                            float t = cu == 0 ? o[0] : cu == 1 ? o[1] : cu == 2 ? o[2] : o[3];
                            if (d.out4_b) t += bq;
                            if (d.out4_x) t += d.out4_ub ? xq[mm][j] / ubv : xq[mm][j];
                            if (d.out4_ub) t *= ubv;
                            const long long gp = (long long)(T.n * d.Ho + oy) * d.Wo + T.ox0 + pj + 8 * j;
                            if (ok && u < 4) d.out4_dst[gp * 4 + cu] = t;
                        } else {
                            if (ok) *(f32x4*)(d.dst + soff) = v;
                        }
                    }
                }
            });
        });
    };

    // ---- split-plane output (OSP): the stored tensor is what its ONE consumer would have staged -- act(FiLM(conv)) split into
    // (h, l) halves, planes [n][C/16][channel half][part][Ho*Wo (+ zero pad)] of 16-byte units.  The accumulator layout is
    // already the store layout: a lane owns 4 consecutive channels of one pixel = half a unit, the 32 pixels of a fragment
    // are 32 consecutive units, so a wave-instruction writes 512 contiguous bytes and nothing goes through LDS but the FiLM
    // vectors (prefetched 8 lanes per block as for the transposed epilogue, redistributed through 256 private bytes).
    // The residual of a split-plane store is read in PLANES OF 4 CHANNELS (YOND_FMT_PLANES4): in the accumulator layout a lane
    // then loads 16 bytes of its own pixel and the 32 pixels of a fragment are 512 contiguous bytes -- no transpose either.
    // With split-plane input (no staging registers) the first two rows' residual is requested at the start of the tile's last
    // step, like the FiLM vectors.
    constexpr bool ORP = OSP && ISP;
    f32x4 qrr[ORP ? PFR : 1][C::NW][4];
    auto lane_geo = [&](const Tile& T) {          // the lane's pixel column: image, first row of the (sub-)tile, column
        FGeo g;
        g.n = T.n; g.oy0 = T.oy0; g.ox0 = T.ox0 + li;
        if constexpr (FOLD != 0) {
            g = fold_geo(T.fu0 + fsub);
            g.ox0 += fcol;
        }
        return g;
    };
    auto res4_addr = [&](const Tile& T, const FGeo& lg, int nn, int m, int g) -> const float* {
        const int oy = lg.oy0 + rg * MW + m, ox = lg.ox0;
        const bool ok = d.res && oy < d.Ho && ox < d.Wo;
        const int c4 = (T.ct * TN + (cg * C::NW + nn) * 32 + 8 * g) / 4 + lh;
        const long long off = (((long long)lg.n * (d.Cout / 4) + c4) * d.Ho * d.Wo + (long long)oy * d.Wo + ox) * 4;
        return (d.res ? d.res : d.wpk) + (ok ? off : 0);
    };
    auto res4_prefetch = [&](const Tile& T) __attribute__((always_inline)) {
        if constexpr (ORP) {
            const FGeo lg = lane_geo(T);
#pragma unroll
            for (int mm = 0; mm < PFR; ++mm)
#pragma unroll
                for (int nn = 0; nn < C::NW; ++nn)
#pragma unroll
                    for (int g = 0; g < 4; ++g) qrr[mm][nn][g] = *(const f32x4*)res4_addr(T, lg, nn, mm, g);
        }
    };
    auto epilogue_sp = [&](auto hr, auto actc, const Tile& T) __attribute__((always_inline)) {
        constexpr bool HAS_RES = (decltype(hr)::value & 1) != 0, HAS_SCALE = (decltype(hr)::value & 2) != 0, HAS_SHIFT = (decltype(hr)::value & 4) != 0;
        constexpr int ACT = decltype(actc)::value;
        // residual rows that were not prefetched: requested first, ahead of all arithmetic
        constexpr int LR0 = ORP ? PFR : 0;
        f32x4 lrr[HAS_RES && MW > LR0 ? MW - LR0 : 1][C::NW][4];
        const FGeo lg = lane_geo(T);
        if constexpr (HAS_RES && MW > LR0) {
#pragma unroll
            for (int mm = LR0; mm < MW; ++mm)
#pragma unroll
                for (int nn = 0; nn < C::NW; ++nn)
#pragma unroll
                    for (int g = 0; g < 4; ++g) lrr[mm - LR0][nn][g] = *(const f32x4*)res4_addr(T, lg, nn, mm, g);
        }
        float* fw = smem + C::FILM_OFF + wave * (C::NW * 64 * (FOLD ? FOLD : 1));
        if (lane < (FOLD ? 16 * FOLD : 16)) {
#pragma unroll
            for (int nn = 0; nn < C::NW; ++nn) *(f32x4*)(fw + (FOLD ? (lane >> 4) * (C::NW * 64) : 0) + nn * 64 + ((lane & 8) ? 32 : 0) + 4 * (lane & 7)) = pes[nn];
        }
        if constexpr (FOLD != 0) fw += fsub * (C::NW * 64);        // the lane's sub-tile reads its own image's vectors
        const int PSo = yond_sp_plane_units(d.Ho, d.Wo);
        const int nc16o = d.Cout / 16;
        const int ox = lg.ox0;
        const bool col_ok = ox < d.Wo;
        const int e_n = lg.n, e_oy0 = lg.oy0;
#pragma unroll
        for (int nn = 0; nn < C::NW; ++nn) {
            const int cb = T.ct * TN + (cg * C::NW + nn) * 32;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 one = {1.0f, 1.0f, 1.0f, 1.0f}, zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
                f32x4 es = one, et = zero4;
                if constexpr (HAS_SCALE) es = *(const f32x4*)(fw + nn * 64 + 8 * g + 4 * lh);
                if constexpr (HAS_SHIFT) et = *(const f32x4*)(fw + nn * 64 + 32 + 8 * g + 4 * lh);
                const int c16 = (cb + 8 * g) >> 4, hh = g & 1;
                char* pb = (char*)d.dst + (size_t)(((e_n * nc16o + c16) * 2 + hh) * PARTS) * (size_t)PSo * 16 + lh * 8;
#pragma unroll
                for (int m = 0; m < MW; ++m) {
                    const int oy = e_oy0 + rg * MW + m;
                    const bool ok = col_ok && oy < d.Ho;
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float y = acc[0][m][nn][4 * g + e];
                        if constexpr (PARTS == 2) y = fmaf(acc[1][m][nn][4 * g + e], 1.0f / 2048.0f, y);
                        y = fmaf(y, es[e], et[e]);
                        if constexpr (ACT == 2 || ACT < 0) y = y > 0.0f ? y : y * slope_eff;
                        if constexpr (ACT == 1) y = split_silu(y);
                        if constexpr (ACT < 0) { if (silu_out) y = split_silu(y); }
                        if constexpr (HAS_RES) y += (m < LR0 ? qrr[m < LR0 ? m : 0][nn][g][e] : lrr[m >= LR0 ? m - LR0 : 0][nn][g][e]);
                        v[e] = y;
                    }
                    amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                    const f16x4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                    char* pp = pb + (size_t)(oy * d.Wo + ox) * 16;
                    if (ok) *(f16x4*)pp = h;
                    if constexpr (PARTS == 2) {
                        const f16x4 l = {(_Float16)((v[0] - (float)h[0]) * 2048.0f), (_Float16)((v[1] - (float)h[1]) * 2048.0f),
                                         (_Float16)((v[2] - (float)h[2]) * 2048.0f), (_Float16)((v[3] - (float)h[3]) * 2048.0f)};
                        if (ok) *(f16x4*)(pp + (size_t)PSo * 16) = l;
                    }
                }
            }
        }
    };

    // (plain form, lane = pixel: kept for the shapes whose free buffer is smaller than the scratch -- the h-only fp16 path)
    auto epilogue_direct = [&](const Tile& T) __attribute__((always_inline)) {      // (left to the inliner, the two call sites of a two-set kernel became real calls in one instantiation: the descriptor went to scratch memory, 13x slower)
        // lane = pixel: NHWC stores from here touch 32 lines with 32 bytes each; into PLANES OF 4 CHANNELS (out_fmt 2) the 32
        // pixels of a fragment are 512 contiguous bytes (K1: every other 16-byte unit of the pixel-shuffled row)
        const FGeo lg = lane_geo(T);
        const int ox = lg.ox0;
        const bool col_ok = ox < d.Wo;
        const bool out_p4 = d.out_fmt == YOND_FMT_PLANES4;
        const int Hout = K1 ? 2 * d.Ho : d.Ho, Wout = K1 ? 2 * d.Wo : d.Wo;
#pragma unroll
        for (int nn = 0; nn < C::NW; ++nn) {
            const int cfull = T.ct * TN + (cg * C::NW + nn) * 32 + 4 * lh;
            const int sp = K1 ? k1_sp(T.ct, cfull) : 0, cbase = K1 ? k1_cb(T.ct, cfull) : cfull;     // K1: sub-position of the tile's channel block
            const int eoff = (d.ebatch ? lg.n * Cr : 0) + cbase;
            f32x4 es[4], et[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 one = {1.0f, 1.0f, 1.0f, 1.0f}, zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
                es[g] = d.escale ? *(const f32x4*)(d.escale + eoff + 8 * g) : one;
                et[g] = d.eshift ? *(const f32x4*)(d.eshift + eoff + 8 * g) : zero4;
            }
            f32x4 rr[MW][4];
#pragma unroll
            for (int m = 0; m < MW; ++m) {
                const int oy = lg.oy0 + rg * MW + m;
                const bool ok = col_ok && oy < d.Ho;
                const long long off = ok ? ((long long)(lg.n * d.Ho + oy) * d.Wo + ox) * d.Cout + cbase : 0;   // masked lanes read element 0..
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
                    rr[m][g] = (d.res && !K1) ? *(const f32x4*)(d.res + off + (ok ? 8 * g : 0)) : z;
                }
            }
#pragma unroll
            for (int m = 0; m < MW; ++m) {
                const int oy = lg.oy0 + rg * MW + m;
                const bool ok = col_ok && oy < d.Ho;
                const long long pixo = (long long)(K1 ? 2 * oy + (sp >> 1) : oy) * Wout + (K1 ? 2 * ox + (sp & 1) : ox);
                float* op = out_p4 ? d.dst + (((long long)lg.n * (Cr / 4) + cbase / 4) * Hout * Wout + pixo) * 4
                                   : d.dst + ((long long)lg.n * Hout * Wout + pixo) * Cr + cbase;
                const long long gstep = out_p4 ? 8LL * Hout * Wout : 8;              // elements from one 8-channel group to the next
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = acc[0][m][nn][4 * g + e];
                        if constexpr (PARTS == 2) x = fmaf(acc[1][m][nn][4 * g + e], 1.0f / 2048.0f, x);
                        x = fmaf(x, es[g][e], et[g][e]);
                        x = x > 0.0f ? x : x * slope_eff;
                        if (silu_out) x = split_silu(x);
                        v[e] = x + rr[m][g][e];
                    }
                    if (ok) *(f32x4*)(op + g * gstep) = v;
                    if constexpr (D2) {
                        // second output: SiLU(v) split into (h, l); a lane holds HALF a 16-byte unit (4 of a pixel's 8 consecutive
                        // channels), as in epilogue_sp
                        f32x4 a;
#pragma unroll
                        for (int e = 0; e < 4; ++e) a[e] = split_silu(v[e]);
                        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(a[0]), fabsf(a[1])), fmaxf(fabsf(a[2]), fabsf(a[3]))));
                        const f16x4 h = {(_Float16)a[0], (_Float16)a[1], (_Float16)a[2], (_Float16)a[3]};
                        const int c8 = cbase - 4 * lh + 8 * g;                          // first channel of the lane's unit
                        const size_t PS2 = (size_t)yond_sp_plane_units(Hout, Wout);
                        char* pp = (char*)d.dst2 + ((size_t)(((lg.n * (Cr / 16) + (c8 >> 4)) * 2 + ((c8 >> 3) & 1)) * PARTS) * PS2 + (size_t)pixo) * 16 + lh * 8;
                        if (ok) *(f16x4*)pp = h;
                        if constexpr (PARTS == 2) {
                            const f16x4 l = {(_Float16)((a[0] - (float)h[0]) * 2048.0f), (_Float16)((a[1] - (float)h[1]) * 2048.0f),
                                             (_Float16)((a[2] - (float)h[2]) * 2048.0f), (_Float16)((a[3] - (float)h[3]) * 2048.0f)};
                            if (ok) *(f16x4*)(pp + PS2 * 16) = l;
                        }
                    }
                }
            }
        }
    };

    // ---- the step pipeline: ONE barrier per step ----
    auto adv = [&](Cur c) {
        Cur n = c;
        const bool last = c.ch + 1 == nchunk;
        n.ch = last ? 0 : c.ch + 1;
        if (last) {
            n.tile = c.tile + G;
            int v = c.ct + g_ct, cy = v >= nct ? 1 : 0;
            n.ct = v - (cy ? nct : 0);
            v = c.tx + g_tx + cy; cy = v >= ntx ? 1 : 0;
            n.tx = v - (cy ? ntx : 0);
            v = c.ty + g_ty + cy; cy = v >= nty ? 1 : 0;
            n.ty = v - (cy ? nty : 0);
            n.n = c.n + g_n + cy;
        }
        return n;
    };
    if (lslot >= total) return;
    const bool clk_on = d.clk != nullptr && blockIdx.x == 0 && tid == 0;       // measurement aid (YondConvDesc.clk)
    unsigned long long clk_c0 = 0, clk_r0 = 0;
    if (clk_on) { clk_c0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
    if constexpr (O4) {
        if (tid < 128) smem[C::W4_OFF + tid] = d.out4_w[tid];                    // visible behind the prologue's barrier
    }
    Cur cs = cursor_at(lslot);                 // step being computed
    Tile cur;                                   // its tile (epilogue)
    Tile lt;                                    // tile of the load cursor
    int lt_tile = -1;
    decode(cs, cur);
    zero_acc();
    float* ibuf = smem;                         // input(s)
    float* obuf = smem + C::IN_FLOATS;          // receives input(s+1)
    float* w0 = smem + 2 * C::IN_FLOATS;        // weights(s)
    float* w1 = w0 + C::W_FLOATS;               // weights(s+1)  (two buffers: receives them)
    float* w2 = w1 + (NWB == 3 ? C::W_FLOATS : 0);   // three buffers: receives weights(s+2)
    auto ct_of = [&](Cur c, int fallback) { return c.tile < total ? c.ct : fallback; };
    auto tile_for = [&](Cur c) {                // steps past the end re-read the last decoded tile (harmless)
        if (c.tile < total && c.tile != lt_tile) { decode(c, lt); lt_tile = c.tile; }
    };
    auto load_all = [&](auto pc, Cur c) {
        tile_for(c);
        const LoadSrc L = load_src(c.ch, lt.n);
        static_for<0, NIN>([&](auto kc) { issue_load(pc, kc, lt, L); });
    };
    auto dma_in_all = [&](Cur c, float* ob) {      // ISP: a whole input image by LDS-DMA
        tile_for(c);
        const InSrc I = in_src(c.ch, lt.n);
        static_for<0, NDI>([&](auto kc) { issue_in_dma(kc, lt, I, ob); });
    };
    auto dma_all = [&](Cur c, float* wb) {
        const float* ws = weight_src(ct_of(c, cur.ct), c.ch);
        static_for<0, C::NWT>([&](auto kc) { issue_dma(kc, ws, wb); });
    };
    // prologue, in the steady-state order of the memory operations (DMA of a step before its loads):
    //   loads input(0) | DMA weights(0), loads input(1) | stage input(0) | DMA weights(1), loads input(2)
    const Cur c1 = adv(cs), c2 = adv(c1);
    if constexpr (!ISP) { if (!ROLES || !consumer) load_all(IntC<0>{}, cs); }
    Cur cl, cw;                                 // cursors of the step's loads / weight DMA
    if constexpr (ISP) {
        // split-plane input: input(0), weights(0) (, weights(1)); step s issues the DMA of input(s+1) and of weights(s + WAHEAD)
        dma_in_all(cs, ibuf);
        dma_all(cs, w0);
        if constexpr (C::WAHEAD == 2) dma_all(c1, w1);
        cl = c1;
        cw = C::WAHEAD == 2 ? c2 : c1;
        split_barrier_keep_loads<C::WAHEAD == 2 ? C::NWT_MIN : 0>();
    } else if constexpr (ROLES != 0) {
        if (!consumer) {
            dma_all(cs, w0);
            write_in(IntC<0>{}, ibuf);
        }
        cl = c1;
        cw = c1;
        split_barrier_keep_loads<0>();
    } else if constexpr (NSET == 2) {
        // two register sets (and two weight buffers): set s % 2 receives input(s+2) during step s, staged in step s+1
        dma_all(cs, w0);
        write_in(IntC<0>{}, ibuf);
        load_all(IntC<1>{}, c1);
        cl = c2;
        cw = c1;
        split_barrier_keep_loads<NIN>();
    } else {
        if constexpr (C::LOADS_FIRST) {
            load_all(IntC<1>{}, c1);
            dma_all(cs, w0);
            write_in(IntC<0>{}, ibuf);
            load_all(IntC<2>{}, c2);
            dma_all(c1, w1);
        } else {
            dma_all(cs, w0);
            load_all(IntC<1>{}, c1);
            write_in(IntC<0>{}, ibuf);
            if constexpr (C::WAHEAD == 2) dma_all(c1, w1);
            load_all(IntC<2>{}, c2);
        }
        cl = adv(c2);                           // loads of step s: input(s+3)
        cw = C::WAHEAD == 2 ? c2 : c1;          // DMA of step s: weights(s + WAHEAD)
        split_barrier_keep_loads<C::WAHEAD == 2 ? NIN + C::NWT_MIN : 2 * NIN>();
    }
    // Step s (all waves alike, S = s % NSET; described for three register sets).  In program order: the MFMAs of step s with, between them, (a) the loads of
    // input(s+3) into register set S and the LDS-DMA of weights(s+2) (three weight buffers; with two: DMA of weights(s+1)
    // first, then the loads), (b) the staging of set (s+1) % 3 = input(s+1), loaded two steps ago; then ONE barrier that
    // awaits weights(s+1) only -- vmcnt counts in order: loads(s-1), DMA(s-1), loads(s), DMA(s): the last two stay in
    // flight.  Loads before DMA because the compiler (which does not see the DMA) makes the next step wait for every
    // outstanding operation before it touches the staged set: the youngest are then L2-resident weight slices.
    auto step = [&](auto sc) __attribute__((always_inline)) -> bool {
        constexpr int S = decltype(sc)::value;
        SDBG(0);
        const bool last_ch = (cs.ch + 1 == nchunk);
        const Cur cn = adv(cs);
        if constexpr (EP_FIT || OSP) {
            if (last_ch) {
                epi_prefetch(cur);
                res_prefetch(cur);
                res4_prefetch(cur);
            }
        }
        auto prep = [&]() __attribute__((always_inline)) {
            tile_for(cl);
            if constexpr (ISP) is_s = in_src(cl.ch, lt.n);
            else ls_s = load_src(cl.ch, lt.n);
            wsrc_s = weight_src(ct_of(cw, cur.ct), cw.ch);
        };
        SDBG(1);
        if constexpr (ROLES != 0) {
            if (consumer) {
                if (computes) mfma_stage(IntC<0>{}, IntC<0>{}, ibuf, w0, obuf, w1, lt, prep);
            } else {
                // the moving half: the weight DMA of step s+1 and the loads of input(s+1), which it stages as soon as they arrive -- all of it under the other
                // half's MFMAs of step s
                prep();
                static_for<0, C::NWT>([&](auto kc) { issue_dma(kc, wsrc_s, w1); });
                static_for<0, NIN>([&](auto kc) { issue_load(IntC<0>{}, kc, lt, ls_s); });
                write_in(IntC<0>{}, obuf);
            }
        } else if (computes) mfma_stage(IntC<(S + 1) % NSET>{}, IntC<S>{}, ibuf, w0, obuf, C::WAHEAD == 2 ? w2 : w1, lt, prep);
        SDBG(2);
        SDBG(3);
        split_barrier_keep_loads<KEEP>();                       // weights(s+1) have landed, input(s+1) is written
        SDBG(4);
        if (last_ch) {
            // scratch: weights(s) / input(s), which no wave reads any more; the barrier behind the epilogue keeps the next
            // step's DMA and staging writes (of OTHER waves) out of it until every wave has read its block back
            if (computes && consumer) {
                if constexpr (OSP) {
                    // straight-line variants: the residual blocks' conv1 (FiLM + SiLU) and conv2 (FiLM + residual), a plain layer
                    // with LeakyReLU; the rest generic
                    const int flags = (d.res ? 1 : 0) | (d.escale ? 2 : 0) | (d.eshift ? 4 : 0);
                    if (flags == 6 && d.post_act == 1) epilogue_sp(IntC<6>{}, IntC<1>{}, cur);
                    else if (ORP && flags == 7 && d.post_act == 0) {
                        if constexpr (ORP) epilogue_sp(IntC<7>{}, IntC<0>{}, cur);
                    } else if (flags == 4 && d.post_act == 2) epilogue_sp(IntC<4>{}, IntC<2>{}, cur);
                    else
                        static_for<0, 8>([&](auto fcx) __attribute__((always_inline)) {
                            // (a residual only with split-plane input -- conv2 of a block: the dispatcher refuses the rest)
                            if constexpr (ORP || (decltype(fcx)::value & 1) == 0) {
                                if (flags == decltype(fcx)::value) epilogue_sp(fcx, IntC<-1>{}, cur);
                            }
                        });
                } else if (K1 && (FOLD != 0 || D2 || d.out_fmt == YOND_FMT_PLANES4)) {
                    epilogue_direct(cur);                          // (planes of 4 channels: stored from the accumulator layout)
                } else if constexpr (EP_FIT) {
                    float* scr = C::EP_OWN ? smem + C::EP_OFF : (EP_IN_W ? w0 : ibuf);
                    const int flags = (d.res ? 1 : 0) | (d.escale ? 2 : 0) | (d.eshift ? 4 : 0);
                    // straight-line variants for what the networks launch: conv2 of a residual block (FiLM + residual), conv1
                    // with its SiLU, a plain layer with bias (+ LeakyReLU); every other combination: generic
                    if (flags == 7 && d.post_act == 0) epilogue(IntC<7>{}, IntC<0>{}, cur, scr);
                    else if (flags == 6 && d.post_act == 1) epilogue(IntC<6>{}, IntC<1>{}, cur, scr);
                    else if (flags == 4 && d.post_act == 2) epilogue(IntC<4>{}, IntC<2>{}, cur, scr);
                    else if (flags == 4 && d.post_act == 0) epilogue(IntC<4>{}, IntC<0>{}, cur, scr);
                    else
                        static_for<0, 8>([&](auto fcx) __attribute__((always_inline)) {
                            if (flags == decltype(fcx)::value) epilogue(fcx, IntC<-1>{}, cur, scr);
                        });
                } else {
                    epilogue_direct(cur);
                }
            }
            SDBG(6);
            if constexpr (EP_FIT && !OSP) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (OSP: nothing shared was touched)
            SDBG(7);
            zero_acc();
            if (cn.tile < total) {
                tile_origin(cn, cur);
            }
        }
        SDBG(5);
        ++dbg_step;
        wskip = wres;
        if (cn.tile >= total) return false;
        cs = cn;
        cl = adv(cl);
        cw = adv(cw);
        float* t = ibuf;
        ibuf = obuf;
        obuf = t;
        t = w0;
        w0 = w1;
        w1 = NWB == 3 ? w2 : t;
        w2 = t;
        return true;
    };
    while (true) {
        if (!step(IntC<0>{})) break;
        if (!step(IntC<1>{})) break;
        if constexpr (NSET == 3) {
            if (!step(IntC<2>{})) break;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the look-ahead loads / DMA of the steps past the end
    if (d.status && !(amax <= 65504.0f)) atomicOr(d.status, YOND_STATUS_HALF_OVERFLOW);   // an h half became +-inf
    if (clk_on) {
        atomicAdd(d.clk, __builtin_amdgcn_s_memtime() - clk_c0);
        atomicAdd(d.clk + 1, __builtin_amdgcn_s_memrealtime() - clk_r0);
    }
}

template <int STRIDE, int TH, int TN, int MW, int PARTS, int NWB, bool PRE, bool O4 = false, bool K1 = false, int ISP = 0, bool OSP = false, bool S2 = false, bool D2 = false, int FOLD = 0, int NWAVE = 8, int ROLES = 0>
int launch_split(const YondConvDesc& d, hipStream_t st) {
    using C = SplitCfg<STRIDE, TH, TN, MW, PARTS, NWB, K1, FOLD, NWAVE, ROLES>;
    static_assert(C::SMEM_BYTES <= 160 * 1024, "LDS budget");
    static bool attr_set = false;
    auto kern = conv_split_kernel<STRIDE, TH, TN, MW, PARTS, NWB, PRE, O4, K1, ISP, OSP, S2, D2, FOLD, NWAVE, ROLES>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    constexpr int FSWL = FOLD ? 32 / FOLD : 32;
    const long long total = FOLD ? (long long)(d.Cout / TN) * (((long long)d.N * ((d.Ho + TH - 1) / TH) * ((d.Wo + FSWL - 1) / FSWL) + FOLD - 1) / (FOLD ? FOLD : 1))
                                 : (long long)(d.Cout / TN) * ((d.Wo + 31) / 32) * ((d.Ho + TH - 1) / TH) * d.N;
    if (total > 0x7fffffffLL) return YOND_EUNSUPPORTED;
    // (four-wave workgroups whose LDS fits twice into a CU: two persistent workgroups per CU -- one's epilogue and prologue under the other's MFMAs)
    constexpr int PER_CU = (NWAVE == 4 && C::SMEM_BYTES <= 80 * 1024) ? 2 : 1;
    const int gmax = (int)yond_exp_long("YOND_SPLIT_GRID", 256) * PER_CU;  // (experiment builds: fewer workgroups than CUs, leaving CUs to a side stream's kernels)
    const int grid = total < gmax ? (int)total : gmax;            // one persistent workgroup per CU
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), C::SMEM_BYTES, st, d);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}


// ---- the instantiations, grouped into translation units that compile in parallel (conv_split_*.hip); the dispatcher
// (conv_split.hip) sees them as extern templates.  X(STRIDE, TH, TN, MW, PARTS, NWB, PRE, O4, K1, ISP, OSP)
#define SPLIT_GROUP_K1S2(X)                                                                             \
    X(1, 8, 32, 1, 2, 3, false, false, true, false, false) X(1, 8, 64, 2, 2, 3, false, false, true, false, false) \
    X(2, 4, 64, 1, 2, 2, false, false, false, false, false) X(2, 4, 64, 1, 1, 2, false, false, false, false, false)
#define SPLIT_GROUP_S1_64(X)                                                                            \
    X(1, 12, 64, 3, 2, 2, true, false, false, false, false) X(1, 12, 64, 3, 2, 2, false, false, false, false, false) \
    X(1, 8, 64, 2, 2, 3, true, false, false, false, false) X(1, 8, 64, 2, 2, 3, false, false, false, false, false)
#define SPLIT_GROUP_S1_32(X)                                                                            \
    X(1, 16, 32, 2, 2, 3, true, true, false, false, false) X(1, 16, 32, 2, 2, 3, false, true, false, false, false) \
    X(1, 16, 32, 2, 2, 3, true, false, false, false, false) X(1, 16, 32, 2, 2, 3, false, false, false, false, false)
#define SPLIT_GROUP_HALF(X)                                                                             \
    X(1, 8, 64, 2, 1, 3, true, false, false, false, false) X(1, 8, 64, 2, 1, 3, false, false, false, false, false) \
    X(1, 16, 32, 2, 1, 3, true, false, false, false, false) X(1, 16, 32, 2, 1, 3, false, false, false, false, false)
// h-only operands (the fp16 path, BASELINE cfg 5) leave ONE accumulator per block: a wave can own two 32-channel blocks in the registers
// the split form spends on its second accumulator -- 128-channel tiles: every pixel fragment read from LDS serves two blocks, twice the
// MFMAs per staged input, per weight DMA and per barrier
#define SPLIT_GROUP_HALF128(X)                                                                          \
    X(1, 12, 128, 3, 1, 2, true, false, false, false, false) X(1, 12, 128, 3, 1, 2, false, false, false, false, false) \
    X(1, 8, 128, 2, 1, 3, true, false, false, false, false) X(1, 8, 128, 2, 1, 3, false, false, false, false, false)
// ... and, for the layers with 64 / 32 output channels, taller tiles than the split form's registers allow: 12 rows x 64 channels (three
// rows per wave) and 32 rows x 32 channels (four rows per wave; its input image is also the first of this path large enough to serve as
// the transposed epilogue's scratch -- the 16-row form stores from the accumulator layout, 32 cache lines per instruction)
#define SPLIT_GROUP_HALF_TALL(X)                                                                        \
    X(1, 12, 64, 3, 1, 2, true, false, false, false, false) X(1, 12, 64, 3, 1, 2, false, false, false, false, false) \
    X(1, 32, 32, 4, 1, 2, true, false, false, false, false) X(1, 32, 32, 4, 1, 2, false, false, false, false, false)
#define SPLIT_GROUP_OSP(X)                                                                              \
    X(1, 12, 64, 3, 2, 2, true, false, false, false, true) X(1, 12, 64, 3, 2, 2, false, false, false, false, true) \
    X(1, 8, 64, 2, 2, 3, true, false, false, false, true) X(1, 8, 64, 2, 2, 3, false, false, false, false, true) \
    X(1, 16, 32, 2, 2, 3, true, false, false, false, true) X(1, 16, 32, 2, 2, 3, false, false, false, false, true)
#define SPLIT_GROUP_ISP(X)                                                                              \
    X(1, 12, 64, 3, 2, 2, false, false, false, true, false) X(1, 8, 64, 2, 2, 3, false, false, false, true, false) \
    X(1, 16, 32, 2, 2, 3, false, true, false, true, false) X(1, 16, 32, 2, 2, 3, false, false, false, true, false)
#define SPLIT_GROUP_ISP_OSP(X)                                                                          \
    X(1, 12, 64, 3, 2, 2, false, false, false, true, true) X(1, 8, 64, 2, 2, 3, false, false, false, true, true) \
    X(1, 16, 32, 2, 2, 3, false, false, false, true, true)
#define SPLIT_GROUP_ISP_K1S2(X)                                                                         \
    X(1, 8, 32, 1, 2, 3, false, false, true, 2, false) X(1, 8, 64, 2, 2, 3, false, false, true, 2, false) \
    X(2, 4, 64, 1, 2, 2, false, false, false, 2, false)
#define SPLIT_GROUP_K1_SUB2(X) X(1, 8, 64, 2, 2, 3, false, false, true, 2, false, true)
// folded tiles (images at most 16 pixels wide): the 8-row kernel of the split-plane data flow with two 16-column sub-tiles per MFMA row
#define SPLIT_GROUP_FOLD(X)                                                                             \
    X(1, 4, 64, 1, 2, 3, false, false, false, true, true, false, false, 2) X(1, 4, 64, 1, 2, 3, true, false, false, false, true, false, false, 2)         \
    X(1, 4, 64, 1, 2, 3, false, false, false, true, true, false, false, 4) X(1, 4, 64, 1, 2, 3, true, false, false, false, true, false, false, 4)
// ... the decoder GEMMs (no halo: the sub-tiles lie side by side as the columns of one tile do)
// ... the plain [N][H][W][C] layer (training's forward and data-gradient convolutions, UNetSeeInDark's deep stages)
#define SPLIT_GROUP_FOLD_NHWC(X) X(1, 4, 64, 1, 2, 3, false, false, false, 0, false, false, false, 2) X(1, 4, 64, 1, 2, 3, false, false, false, 0, false, false, false, 4)
// the decoder GEMM that also stores SiLU(value) in split planes (YondConvDesc.dst2: the next block's conv1 then stages by LDS-DMA alone)
#define SPLIT_GROUP_K1_D2(X) X(1, 8, 64, 2, 2, 3, false, false, true, 2, false, false, true)
#define SPLIT_GROUP_FOLD_K1(X) X(1, 8, 64, 2, 2, 3, false, false, true, 2, false, false, false, 2) X(1, 8, 64, 2, 2, 3, false, false, true, 2, false, false, false, 4)
// ... and the stride-2 layers of the flow (split planes in, planes of 4 channels out, with and without the second, split-plane output)
#define SPLIT_GROUP_FOLD_S2(X)                                                                          \
    X(2, 4, 64, 1, 2, 2, false, false, false, 2, false, false, true, 2) X(2, 4, 64, 1, 2, 2, false, false, false, 2, false, false, false, 2)            \
    X(2, 4, 64, 1, 2, 2, false, false, false, 2, false, false, true, 4) X(2, 4, 64, 1, 2, 2, false, false, false, 2, false, false, false, 4)            \
    X(2, 4, 64, 1, 2, 2, false, false, false, 0, false, false, false, 2) X(2, 4, 64, 1, 2, 2, false, false, false, 0, false, false, false, 4)
#define SPLIT_GROUP_D2(X) X(2, 4, 64, 1, 2, 2, false, false, false, 2, false, false, true)
#define SPLIT_GROUP_WRES(X)                                                                             \
    X(1, 16, 32, 2, 2, 2, true, false, false, false, true) X(1, 16, 32, 2, 2, 2, false, false, false, true, true) \
    X(1, 16, 32, 2, 2, 2, false, true, false, true, false)
// ---- the split-plane data flow on H-ONLY planes (PARTS 1: the fp16 path, BASELINE cfg 5): block-internal and block-output tensors travel as
// 2-byte halves [n][C/16][channel half][H*W (+ zero pad)] x 16-byte units
// conv1 from the float32 block input (planes of 4 channels), SiLU in its staging, h-only store
#define SPLIT_GROUP_H_OSP(X)                                                                            \
    X(1, 12, 64, 3, 1, 2, true, false, false, false, true) X(1, 8, 64, 2, 1, 3, true, false, false, false, true) \
    X(1, 16, 32, 2, 1, 3, true, false, false, false, true)
// conv2 (h-only planes in by LDS-DMA, FiLM + float32 residual in planes of 4, h-only store) and conv1 of the levels whose producer stored SiLU(x)
#define SPLIT_GROUP_H_ISP_OSP(X)                                                                        \
    X(1, 12, 64, 3, 1, 2, false, false, false, true, true) X(1, 8, 64, 2, 1, 3, false, false, false, true, true) \
    X(1, 16, 32, 2, 1, 3, false, false, false, true, true)
// the last convolution with the fused output projection
#define SPLIT_GROUP_H_ISP_O4(X) X(1, 16, 32, 2, 1, 3, false, true, false, true, false)
// stride 2 and the decoder GEMMs: h-only planes through the register sets, planes of 4 channels (float32) out; with the second (h-only) output
#define SPLIT_GROUP_H_K1S2(X)                                                                           \
    X(1, 8, 32, 1, 1, 3, false, false, true, 2, false) X(1, 8, 64, 2, 1, 3, false, false, true, 2, false) \
    X(2, 4, 64, 1, 1, 2, false, false, false, 2, false)
#define SPLIT_GROUP_H_D2(X) X(2, 4, 64, 1, 1, 2, false, false, false, 2, false, false, true) X(1, 8, 64, 2, 1, 3, false, false, true, 2, false, false, true)
#define SPLIT_GROUP_H_SUB2(X) X(1, 8, 64, 2, 1, 3, false, false, true, 2, false, true)
// ... and 128-channel tiles for the flow's 3x3 layers with >= 128 output channels (one accumulator per block: a wave owns two blocks, as in HALF128)
// (8-row tiles: the 12-row forms of these two spill 33 / 45 registers)
#define SPLIT_GROUP_H128_OSP(X) X(1, 8, 128, 2, 1, 3, true, false, false, false, true)
#define SPLIT_GROUP_H128_ISP_OSP(X) X(1, 8, 128, 2, 1, 3, false, false, false, true, true)
// (resident weights for the 32 -> 32 layers, as SPLIT_GROUP_WRES, measured on this path in round 6: no gain -- its level 0 is bound by HBM bytes and the
// SiLU arithmetic, not by the weight DMA -- and not built)
// ... and the flow's stride-2 layers on 8-row tiles, two output rows per wave, 64 or 128 channels: with h-only operands a fragment read from LDS feeds
// ONE MFMA (not 1.5), and the 4-row form reads 2.0 KiB of fragments per MFMA where the CU's LDS delivers 1 KiB per MFMA slot -- its steps are
// LDS-read bound (in-kernel stamps, profiles/r06_experiments).  Two rows per wave share every weight fragment and 1 of 6 pixel-row fragments
// (1.33 KiB per MFMA), two channel blocks per wave share every pixel fragment (0.92); the h-only planes leave the LDS for the 17-row input image
// (the 128-channel form with the second output spills 42 registers: not built -- those layers take the 64-channel form)
#define SPLIT_GROUP_H_S2_TALL(X)                                                                        \
    X(2, 8, 64, 2, 1, 2, false, false, false, 2, false) X(2, 8, 128, 2, 1, 2, false, false, false, 2, false) \
    X(2, 8, 64, 2, 1, 2, false, false, false, 2, false, false, true)
// ... and its decoder GEMMs: 16-row tiles (four rows per wave share every weight fragment), 128 GEMM columns where an output pixel has >= 128 channels
// (two column blocks per wave share every pixel fragment): 1.5 KiB of LDS fragments per MFMA (8 rows x 64 columns) -> 1.25 (16 x 64) / 1.0 (8 x 128)
// (16 rows x 128 columns spills 52 registers: not built)
#define SPLIT_GROUP_H_K1_TALL(X) X(1, 16, 64, 4, 1, 2, false, false, true, 2, false) X(1, 8, 128, 2, 1, 3, false, false, true, 2, false)
// ... and four rows per wave for its 32- and 64-channel 3x3 layers (every weight fragment serves four rows: 1.17 -> 0.75 and 0.89 -> 0.75 KiB of LDS
// fragments per MFMA; the 32-row forms with register-staged input / the output projection spill 39 / 8 registers: not built)
#define SPLIT_GROUP_H_TALL4(X)                                                                          \
    X(1, 32, 32, 4, 1, 2, false, false, false, true, true)                                               \
    X(1, 16, 64, 4, 1, 2, true, false, false, false, true) X(1, 16, 64, 4, 1, 2, false, false, false, true, true)
// round 6, the split path's stride-2 layers with ONE wave per SIMD (256 threads): two output rows per wave (0.89 instead of 1.33 KiB of LDS fragments per MFMA) in the
// same 4 x 32 x 64 tile and the same LDS
#define SPLIT_GROUP_S2_W4(X)                                                                            \
    X(2, 4, 64, 2, 2, 2, false, false, false, 2, false, false, false, 0, 4) X(2, 4, 64, 2, 2, 2, false, false, false, 2, false, false, true, 0, 4)
// ... and with eight waves in two roles: four multiply (two output rows each), four move the data
#define SPLIT_GROUP_S2_ROLES(X)                                                                         \
    X(2, 4, 64, 2, 2, 2, false, false, false, 2, false, false, false, 0, 8, 1) X(2, 4, 64, 2, 2, 2, false, false, false, 2, false, false, true, 0, 8, 1)
// round 6, level 0 (32 -> 32 channels, resident weights): HALF-SIZE workgroups -- four waves (one per SIMD) on 8 x 32-pixel tiles, 80 KB of LDS, TWO per CU (512
// persistent workgroups): a level-0 tile has two 16-channel steps and a full epilogue, and with one workgroup per CU nothing runs under its epilogue, its first
// loads and its barriers (stamps: ~19 k cycles per tile for ~7 k of matrix pipe); two independent workgroups were meant to drift apart and fill each other's gaps.
// Measured (same box): conv1 2-7 % SLOWER, conv2 equal, the last convolution 4 % faster, the frame 2 % slower -- experiment builds only
#define SPLIT_GROUP_WRES_W4(X)                                                                          \
    X(1, 8, 32, 2, 2, 2, true, false, false, false, true, false, false, 0, 4) X(1, 8, 32, 2, 2, 2, false, false, false, true, true, false, false, 0, 4) \
    X(1, 8, 32, 2, 2, 2, false, true, false, true, false, false, false, 0, 4)
#define SPLIT_INSTANTIATE(...) template int launch_split<__VA_ARGS__>(const YondConvDesc&, hipStream_t);
#define SPLIT_EXTERN(...) extern template int launch_split<__VA_ARGS__>(const YondConvDesc&, hipStream_t);
