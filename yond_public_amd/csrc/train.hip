// N4, first slice: the kernels a training step of the AWGN denoiser needs beyond the forward pass
// (trainer_AWGN.py:101-117: pred = net(lr, sigma); loss = F.l1_loss(pred, hr); loss.backward(); Adam.step()).
//   wgrad   dW[tap][co][ci] = sum over pixels of dY[p_out][co] * X[p_in(p_out, tap)][ci] for a 3x3 (stride 1 / 2, pad 1), a
//           2x2 stride-2 transposed or a 1x1 convolution, NHWC float32, on the fp32 matrix cores (v_mfma_f32_32x32x2_f32:
//           one MFMA = the outer product of two pixels' channel vectors; exact fp32 products, fp32 accumulation)
//   colsum  db[c] = sum over pixels of dY[p][c]
//   l1      loss = mean |pred - target|, dpred = sign(pred - target) / n            (losses/base_loss.py:81-113: F.l1_loss)
//   adam    torch.optim.Adam's update (defaults betas 0.9 / 0.999, eps 1e-8, no weight decay, no amsgrad)
// The data gradients (dgrad) need no kernels of their own: they are convolutions with re-indexed weights and run on the
// forward kernels (yond_public_amd/train.py).  Correctness first: operands stream from global memory / L2 straight into
// the MFMA (one 4-byte load per operand and lane), partial sums are combined with float atomics.
#include "common.h"

// Zero-fill on the stream as a KERNEL, not hipMemsetAsync: the training step is replayed as a captured hipGraph, and the memset NODES of such a
// graph did not reliably precede the kernels that accumulate (atomicAdd) onto the cleared buffers -- after an idle spell a replay now and then
// summed onto whatever the buffer held (bias / FiLM gradients of 1e35, then NaN parameters: found with tools/probe/poison_run.py, which fills
// every torch.empty with NaN).  Kernel nodes of one stream keep their order.
__global__ __launch_bounds__(256) void zero_words_kernel(unsigned int* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0u;
}
static hipError_t zero_async(void* p, size_t bytes, hipStream_t st) {
    const size_t n = bytes / 4;                                  // (every caller clears float / double arrays)
    if (!n) return hipSuccess;
    size_t nb = (n + 255) / 256;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)nb), dim3(256), 0, st, (unsigned int*)p, n);
    return hipGetLastError();
}

struct WgradGeom {
    int N, H, W, Cin;          // X: [N][H][W][Cin]
    int Ho, Wo, Cout;          // dY: [N][Ho][Wo][Cout]
    int mode;                  // 0: 3x3 pad 1 (stride s), 1: 2x2 stride-2 transposed (Ho = 2H, Wo = 2W), 2: 1x1
    int stride, taps;
    int chunk;                 // K-space rows per wave
};

// One wave per (NCO 32-wide co tiles, one 32-wide ci tile, chunk of K-space rows), ALL taps at once.  MFMA 32x32x2: lane l supplies
// A[row l&31][k = l>>5] and B[k = l>>5][col l&31]: A = dY (row = output channel), B = X (column = input channel), k = two
// neighbouring pixels of a row -- lanes 0..31 read 32 consecutive channels of one pixel, lanes 32..63 of the next (two 128-byte
// segments).  Per pixel pair a 3x3 layer loads NCO values of dY and nine of X (the 3x3 neighbourhood; rows outside the image are
// skipped wave-uniformly, columns outside it read as zero) and issues 9 NCO MFMAs into 9 NCO accumulators (AGPRs), so a loaded
// operand feeds 9 (dY) or NCO (X) products instead of one; the transposed 2x2 layer is the mirror image (one X value, 4 NCO of dY).
// The next pair's operands are fetched before the current pair's MFMAs are issued (one wave per SIMD: nothing else hides the
// L2 latency), rows follow one another without draining that pipeline.  K-space rows: dY rows, or X rows for the transposed layer.
// A buffer descriptor (base, stride 0, size in bytes, raw 32-bit format) whose words are PROVABLY wave-uniform: readfirstlane on the
// pointer halves and the size -- without it the loop-carried row cursor counts as divergent and every buffer load is wrapped in a
// waterfall loop.
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 uniform_srd(const void* p, int bytes) {
    const unsigned long long a = (unsigned long long)p;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));  // stride 0
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000;
    return r;
}

template <int MODE, int NCO>
__global__ __launch_bounds__(256) void wgrad_rows_kernel(const float* __restrict__ x, const float* __restrict__ dy, WgradGeom g,
                                                         float* __restrict__ dw /* [taps][Cout][Cin], zeroed by the caller */,
                                                         float* __restrict__ ws /* partial sums per workgroup, or null: atomics */) {
    // taps per wave: a 3x3 layer's kernel rows go to three different waves (grid x = ky-major): 3 NCO accumulators = 96 registers
    // at NCO 2, two waves per SIMD -- nine taps in one wave (288 registers) left one wave per SIMD and nothing to hide latency with
    // With ONE output-channel tile (the 32-channel level) a wave takes all three kernel rows instead: nine accumulators (144
    // registers), one dY and nine X values per pixel pair for nine MFMAs -- the per-pair cursor, descriptor and offset work is
    // shared by three times as many MFMAs (three MFMAs per pair left the launch bound by exactly that work).
    constexpr bool ALLKY = MODE == 0 && NCO == 1;
    constexpr int TAPS = MODE == 0 ? (ALLKY ? 9 : 3) : (MODE == 1 ? 4 : 1);
    constexpr int NA = MODE == 1 ? 4 * NCO : NCO;
    // (the wave index through readfirstlane: the compiler cannot see that threadIdx.x >> 6 is wave-uniform, and everything derived
    // from it -- the row cursor, the 64-bit row bases -- would be computed on the vector ALU, whose time ADDS to the fp32 MFMAs')
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, lk = lane >> 5;
    const int nci = g.Cin / 32, ntile = nci * (g.Cout / (32 * NCO));
    const int ky = (MODE == 0 && !ALLKY) ? blockIdx.x / ntile : 0;
    const int tile = (MODE == 0 && !ALLKY) ? blockIdx.x % ntile : blockIdx.x;
    const int cit = tile % nci, cog = tile / nci;
    const int Hk = MODE == 1 ? g.H : g.Ho, Wk = MODE == 1 ? g.W : g.Wo;
    // (32-bit cursors: rows, positions and row sizes are checked on the host to stay below 2^31 -- 64-bit counters put multiplies,
    // compares and carries of the cursor on the vector ALU)
    const int rows = g.N * Hk;
    const int r0 = ((int)blockIdx.y * 4 + wave) * g.chunk;
    const bool idle = r0 >= rows;
    if (idle && !ws) return;                                        // (with a workspace every wave takes part in the workgroup's sum)
    const int r1 = idle ? r0 : (r0 + g.chunk < rows ? r0 + g.chunk : rows);
    const int rsafe = idle ? 0 : r0;                          // a row every wave may read (an idle wave's r0 lies past the tensor)
    const int nsafe = rsafe / Hk, ysafe = rsafe - nsafe * Hk;      // (divisions: once, not per stage)
    const int s = g.stride;

    f32x16 acc[TAPS][NCO];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int c = 0; c < NCO; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][c][r] = 0.0f;
    // (The bias gradient = the column sums of dY was tried as a by-product of these loads -- one add per loaded dY value in the
    // waves of one input-channel tile: the launch slowed down by 40-50 %; it stays a kernel of its own, colsum_kernel.)

    if constexpr (MODE == 0) {
        // The 3x3 pipeline is STRAIGHT-LINE code: D pixel pairs in flight, stage d refilled right behind its MFMAs, and no branch
        // anywhere in the loop body -- a uniform `if` around a stage (a row outside the image, the ends of a row, the end of the
        // chunk) makes the compiler drain every outstanding load at the join (s_waitcnt vmcnt(0)) and the prefetch depth is gone:
        // measured +40-50 % per launch for each such branch.  So every load is issued unconditionally and what must not count reads
        // as zero (below); a kernel row outside the image still issues its MFMAs, on zeros.
        constexpr int NKY = ALLKY ? 3 : 1, NBV = 3 * NKY;              // kernel rows per wave, X values per pixel pair
        constexpr int D = 6;                                        // (D (NCO + NBV) loads in flight: 30 / 60 of the counter's 63)
        const int ppr = (Wk + 1) / 2;                               // positions (pixel pairs) per row
        const int total = (r1 - r0) * ppr;
        int frow = r0;                                              // the fetch cursor (all scalar)
        int fn = nsafe, fy = ysafe, fxo = 0;                         // (an idle wave fetches nothing that counts)
        int fetched = 0;
        float pa[D][NCO], pb[D][NBV];
        const unsigned rowA = (unsigned)(Wk * g.Cout), rowB = (unsigned)(g.W * g.Cin);     // elements per image row
        const unsigned cout4 = (unsigned)g.Cout * 4u, cin4 = (unsigned)g.Cin * 4u;
        const unsigned lane4 = (unsigned)li * 4u;
        // Operands come through BUFFER loads: a descriptor per image row (built on the scalar unit: base = the row's first byte
        // of this wave's channel tile, size = the bytes left in the row) and a 32-bit per-lane byte offset.  The range check of the
        // buffer path returns 0 for every byte outside the row -- column -1 (the offset wraps), column W, the second pixel of an
        // odd row's last pair -- and a size of 0 silences a whole stage (a kernel row outside the image, a pair past the end of
        // the chunk): no clamps, no selects, no 64-bit per-lane addresses; what is left on the vector ALU per pair is six
        // instructions of offset arithmetic.  (fp32 MFMA time and vector-ALU time add on gfx950.)
        auto fetch0 = [&](float (&a)[NCO], float (&b)[NBV]) {
            const bool live = fetched < total;
            const int rowc = live ? frow : rsafe;                   // (past the end of the chunk: any valid row)
            const int nc = live ? fn : nsafe;
            const int yi0 = fy * s + ky - 1;                         // the (first) kernel row's input row
            const bool aon = ALLKY ? live : (live && yi0 >= 0 && yi0 < g.H);
            const float* dyr = dy + (unsigned long long)(unsigned)rowc * rowA + cog * (32 * NCO);          // (32 x 32 -> 64 bit: two scalar multiplies)
            const i32x4 ra = uniform_srd(dyr, aon ? (int)(Wk * cout4 - cog * (128u * NCO)) : 0);
            const unsigned px = (unsigned)(fxo + lk);
            const unsigned aoff = __umul24(px, cout4) + lane4;
            // (inline asm: the compiler's own wait placement drains ALL loads at the top of the loop -- vmcnt(0) -- whatever the
            // order of the stages; these loads are invisible to it and waited for by the counted s_waitcnt in front of their stage)
            // s_nop 4 in front of every load: a descriptor word may have been written by the vector ALU just before (readfirstlane,
            // the readlane of a spilled scalar) and "VALU writes SGPR -> VMEM reads it" needs five wait states; the compiler's hazard
            // recognizer does not look into inline asm.  (Found with a deeper pipeline whose scalar spills made it bite: 25 % error.)
            asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, 0 offen" : "=v"(a[0]) : "v"(aoff), "s"(ra) : "memory");
            if constexpr (NCO == 2) asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, 0 offen offset:128" : "=v"(a[1]) : "v"(aoff), "s"(ra) : "memory");
            const unsigned boff = __umul24(px * (unsigned)s, cin4) + lane4 - cin4;          // column px s - 1 (wraps below column 0)
#pragma unroll
            for (int r = 0; r < NKY; ++r) {
                const int yi = yi0 + r;
                const bool on = live && yi >= 0 && yi < g.H;       // (a kernel row outside the image: an empty descriptor)
                const float* xr = x + (unsigned long long)(unsigned)(nc * g.H + (on ? yi : 0)) * rowB + cit * 32;
                const i32x4 rb = uniform_srd(xr, on ? (int)(g.W * cin4 - cit * 128u) : 0);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const unsigned bo = boff + cin4 * kx;
                    asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, 0 offen" : "=v"(b[r * 3 + kx]) : "v"(bo), "s"(rb) : "memory");
                }
            }
            // advance (selects, no branches)
            const bool wrap = fxo + 2 >= Wk;
            fxo = wrap ? 0 : fxo + 2;
            const bool ywrap = wrap && fy + 1 == Hk;
            frow += wrap ? 1 : 0;
            fy = wrap ? (ywrap ? 0 : fy + 1) : fy;
            fn += ywrap ? 1 : 0;
            ++fetched;
        };
#pragma unroll
        for (int d = 0; d < D; ++d) fetch0(pa[d], pb[d]);
        for (int i = 0; i < total; i += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                // this stage's NCO + 3 loads are the oldest of the D (NCO + 3) in flight; the operands pass through the wait so that
                // the MFMAs cannot be scheduled in front of it
                if constexpr (NCO == 2)
                    asm volatile("s_waitcnt vmcnt(%5)" : "+v"(pa[d][0]), "+v"(pa[d][1]), "+v"(pb[d][0]), "+v"(pb[d][1]), "+v"(pb[d][2]) : "n"((D - 1) * 5));
                else
                    asm volatile("s_waitcnt vmcnt(%10)" : "+v"(pa[d][0]), "+v"(pb[d][0]), "+v"(pb[d][1]), "+v"(pb[d][2]), "+v"(pb[d][3]), "+v"(pb[d][4]),
                                 "+v"(pb[d][5]), "+v"(pb[d][6]), "+v"(pb[d][7]), "+v"(pb[d][8]) : "n"((D - 1) * 10));
#pragma unroll
                for (int t = 0; t < NBV; ++t)
#pragma unroll
                    for (int c = 0; c < NCO; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[d][c], pb[d][t], acc[t][c], 0, 0, 0);
                fetch0(pa[d], pb[d]);
                __builtin_amdgcn_sched_barrier(0);      // (left alone the scheduler issues all 36 MFMAs, then all 30 loads)
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the D stages past the end (empty descriptors) still write their registers

    } else {
        // 1x1 and transposed 2x2 layers: the same straight-line pipeline; one X value per pair, NA values of dY (the transposed layer:
        // its four taps)
        constexpr int D = MODE == 1 ? 4 : 8;
        const int ppr = (Wk + 1) / 2;
        const int total = (r1 - r0) * ppr;
        int frow = r0;
        int fn = nsafe, fy = ysafe, fxo = 0;                         // (an idle wave fetches nothing that counts)
        int fetched = 0;
        float pa[D][NA], pb[D];
        const unsigned rowA = (unsigned)((MODE == 1 ? g.Wo : Wk) * g.Cout), rowB = (unsigned)(Wk * g.Cin);
        const unsigned cout4 = (unsigned)g.Cout * 4u, cin4 = (unsigned)g.Cin * 4u;
        const unsigned lane4 = (unsigned)li * 4u;
        // (buffer loads through row descriptors, as above: a pixel past the row's end reads as zero, an empty descriptor silences a
        // pair past the end of the chunk)
        auto bload = [](float& dst, unsigned off, const i32x4& r) {
            asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, 0 offen" : "=v"(dst) : "v"(off), "s"(r) : "memory");
        };
        auto bload128 = [](float& dst, unsigned off, const i32x4& r) {
            asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, 0 offen offset:128" : "=v"(dst) : "v"(off), "s"(r) : "memory");
        };
        auto fetch12 = [&](float (&a)[NA], float& b) {
            const bool live = fetched < total;
            const int rowc = live ? frow : rsafe;
            const int nc = live ? fn : nsafe, yc = live ? fy : ysafe;
            const unsigned px = (unsigned)(fxo + lk);
            const i32x4 rb = uniform_srd(x + (unsigned long long)(unsigned)rowc * rowB + cit * 32, live ? (int)(Wk * cin4 - cit * 128u) : 0);
            bload(b, __umul24(px, cin4) + lane4, rb);
            if constexpr (MODE == 1) {
#pragma unroll
                for (int ty = 0; ty < 2; ++ty) {
                    const i32x4 ra = uniform_srd(dy + (unsigned long long)(unsigned)(nc * g.Ho + 2 * yc + ty) * rowA + cog * (32 * NCO),
                                                 live ? (int)(g.Wo * cout4 - cog * (128u * NCO)) : 0);
#pragma unroll
                    for (int tx = 0; tx < 2; ++tx) {
                        const unsigned ao = __umul24(2u * px + tx, cout4) + lane4;
                        bload(a[(ty * 2 + tx) * NCO], ao, ra);
                        if constexpr (NCO == 2) bload128(a[(ty * 2 + tx) * NCO + 1], ao, ra);
                    }
                }
            } else {
                const i32x4 ra = uniform_srd(dy + (unsigned long long)(unsigned)rowc * rowA + cog * (32 * NCO), live ? (int)(Wk * cout4 - cog * (128u * NCO)) : 0);
                const unsigned ao = __umul24(px, cout4) + lane4;
                bload(a[0], ao, ra);
                if constexpr (NCO == 2) bload128(a[1], ao, ra);
            }
            const bool wrap = fxo + 2 >= Wk;
            fxo = wrap ? 0 : fxo + 2;
            const bool ywrap = wrap && fy + 1 == Hk;
            frow += wrap ? 1 : 0;
            fy = wrap ? (ywrap ? 0 : fy + 1) : fy;
            fn += ywrap ? 1 : 0;
            ++fetched;
        };
#pragma unroll
        for (int d = 0; d < D; ++d) fetch12(pa[d], pb[d]);
        for (int i = 0; i < total; i += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                // (NA + 1 loads per stage; the operands pass through the wait: no MFMA above it)
                if constexpr (NA == 1) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(pa[d][0]), "+v"(pb[d]) : "n"((D - 1) * 2));
                else if constexpr (NA == 2) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(pa[d][0]), "+v"(pa[d][1]), "+v"(pb[d]) : "n"((D - 1) * 3));
                else {
                    static_assert(NA == 4, "the transposed layer runs with one output-channel tile per wave");
                    asm volatile("s_waitcnt vmcnt(%5)" : "+v"(pa[d][0]), "+v"(pa[d][1]), "+v"(pa[d][2]), "+v"(pa[d][3]), "+v"(pb[d]) : "n"((D - 1) * 5));
                }
#pragma unroll
                for (int t = 0; t < TAPS; ++t)
#pragma unroll
                    for (int c = 0; c < NCO; ++c)
                        acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(MODE == 1 ? pa[d][t * NCO + c] : pa[d][c], pb[d], acc[t][c], 0, 0, 0);
                fetch12(pa[d], pb[d]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // D rows (output channel) (r&3) + 8 (r>>2) + 4 lk, column (input channel) li
    if (ws) {
        // The K axis is split over thousands of waves; their partial sums meet in few addresses (a 32 -> 32 channel layer has
        // 9,216 weights), and memory-side float atomics onto the same address serialise: 1,368 of them in a row cost more than
        // the MFMAs.  So: the four waves of a workgroup add up through LDS, the workgroup STORES its partial tile
        // [workgroup chunk][tap][Cout][Cin], wgrad_reduce64_kernel adds the chunks.
        __shared__ float red[4][16][64];
        const size_t per = (size_t)g.taps * g.Cout * g.Cin;                     // floats per workgroup chunk
        float* wsc = ws + (size_t)blockIdx.y * per;
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int c = 0; c < NCO; ++c) {
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[t][c][r];
                __syncthreads();
                float* out = wsc + ((size_t)((MODE == 0 && !ALLKY) ? ky * 3 + t : t) * g.Cout + cog * (32 * NCO) + c * 32) * g.Cin + cit * 32 + li;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int r = wave * 4 + rr;
                    const float v = (red[0][r][lane] + red[1][r][lane]) + (red[2][r][lane] + red[3][r][lane]);
                    out[(size_t)((r & 3) + 8 * (r >> 2) + 4 * lk) * g.Cin] = v;
                }
            }
        return;
    }
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int c = 0; c < NCO; ++c) {
            float* out = dw + ((size_t)((MODE == 0 && !ALLKY) ? ky * 3 + t : t) * g.Cout + cog * (32 * NCO) + c * 32) * g.Cin + cit * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) atomicAdd(out + (size_t)((r & 3) + 8 * (r >> 2) + 4 * lk) * g.Cin, acc[t][c][r]);
        }
}

// (wgrad_split.hip: the split-operand weight-gradient kernel stores slices of the same form)
// dw[i] = sum over the slices of ws[slice][i], no atomics and no zeroed output: a workgroup owns 64 consecutive elements, its four
// waves take every fourth slice and meet in LDS.  taps > 0: the first taps * cc elements ([tap][co][ci]) are written as
// [co][ci][tap] -- the OIHW order of an nn.Conv2d weight, so that the caller needs no permute / copy pass -- the rest (the bias
// gradient behind them) where it is.
__global__ __launch_bounds__(256) void wgrad_reduce64_kernel(const float* __restrict__ ws, int nchunk, size_t n, float* __restrict__ dw, int taps,
                                                             size_t cc) {
    __shared__ float part[4][64];
    const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * 64 + e;
    float a0 = 0.0f, a1 = 0.0f;
    if (i < n) {
        int j = q;
        for (; j + 4 < nchunk; j += 8) {
            a0 += ws[(size_t)j * n + i];
            a1 += ws[(size_t)(j + 4) * n + i];
        }
        for (; j < nchunk; j += 4) a0 += ws[(size_t)j * n + i];
    }
    part[q][e] = a0 + a1;
    __syncthreads();
    if (q == 0 && i < n) {
        const float v = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
        size_t o = i;
        if (taps > 0 && i < (size_t)taps * cc) o = (i % cc) * taps + i / cc;
        dw[o] = v;
    }
}
int yond_wgrad_reduce_launch(const float* ws, int nchunk, size_t n, float* dw, hipStream_t st, int oihw_taps, size_t cc) {
    hipLaunchKernelGGL(wgrad_reduce64_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, ws, nchunk, n, dw, oihw_taps, cc);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

// chunk (K-space rows per wave) and workgroup chunks of a launch
template <int MODE, int NCO>
static void wgrad_split(const WgradGeom& g, long long& chunk, long long& wgchunks) {
    const int Hk = MODE == 1 ? g.H : g.Ho, Wk = MODE == 1 ? g.W : g.Wo;
    const long long rows = (long long)g.N * Hk;
    const long long tiles = (long long)(g.Cout / (32 * NCO)) * (g.Cin / 32) * ((MODE == 0 && NCO == 2) ? 3 : 1);
    // about one round of waves at the kernel's occupancy (256 CUs x 4 SIMDs x waves per SIMD), at least ~128 pixels per wave
    const long long target = 1024 * (MODE == 0 ? (NCO == 2 ? 3 : 2) : 4);
    long long chunks = (target + tiles - 1) / tiles;
    const long long min_rows = (128 + Wk - 1) / Wk;
    chunk = (rows + chunks - 1) / chunks;
    if (chunk < min_rows) chunk = min_rows;
    chunks = (rows + chunk - 1) / chunk;
    wgchunks = (chunks + 3) / 4;
}

template <int MODE, int NCO>
static void launch_wgrad(const float* x, const float* dy, WgradGeom g, float* dw, float* ws, hipStream_t stream) {
    const long long tiles = (long long)(g.Cout / (32 * NCO)) * (g.Cin / 32) * ((MODE == 0 && NCO == 2) ? 3 : 1);
    long long chunk, wgchunks;
    wgrad_split<MODE, NCO>(g, chunk, wgchunks);
    g.chunk = (int)chunk;
    hipLaunchKernelGGL((wgrad_rows_kernel<MODE, NCO>), dim3((unsigned)tiles, (unsigned)wgchunks), dim3(256), 0, stream, x, dy, g, dw, ws);
    if (ws) yond_wgrad_reduce_launch(ws, (int)wgchunks, (size_t)g.taps * g.Cout * g.Cin, dw, stream, 0, 0);
}

static bool wgrad_check(const float* x, const float* dy, const float* dw, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int mode, int stride) {
    if (!x || !dy || !dw || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % 32 || Cout % 32) return false;
    // the kernel's 32-bit cursors and row offsets: rows, row sizes in bytes and pixel pairs per wave chunk below 2^31
    const long long hk = mode == 1 ? H : Ho, big = 0x7fffffffLL;
    if ((long long)N * (hk > Ho ? hk : Ho) >= big / 4 || (long long)Wo * Cout * 4 >= big || (long long)W * Cin * 4 >= big || (long long)N * H * W >= big) return false;
    if (mode < 0 || mode > 2 || (mode == 0 && stride != 1 && stride != 2)) return false;
    if (mode == 0 && (Ho != (H + stride - 1) / stride || Wo != (W + stride - 1) / stride)) return false;
    if (mode == 1 && (Ho != 2 * H || Wo != 2 * W)) return false;
    if (mode == 2 && (Ho != H || Wo != W)) return false;
    return true;
}

// bytes of workspace yond_conv_wgrad_ws_f32 wants for this layer (0: invalid arguments)
extern "C" size_t yond_conv_wgrad_ws_bytes(int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int mode, int stride) {
    if (!wgrad_check((const float*)1, (const float*)1, (const float*)1, N, H, W, Cin, Ho, Wo, Cout, mode, stride)) return 0;
    WgradGeom g{N, H, W, Cin, Ho, Wo, Cout, mode, stride, mode == 0 ? 9 : (mode == 1 ? 4 : 1), 0};
    long long chunk, wgchunks;
    const bool two = Cout % 64 == 0;
    if (mode == 0) two ? wgrad_split<0, 2>(g, chunk, wgchunks) : wgrad_split<0, 1>(g, chunk, wgchunks);
    else if (mode == 1) wgrad_split<1, 1>(g, chunk, wgchunks);
    else two ? wgrad_split<2, 2>(g, chunk, wgchunks) : wgrad_split<2, 1>(g, chunk, wgchunks);
    return (size_t)wgchunks * (size_t)g.taps * Cout * Cin * sizeof(float);
}

extern "C" int yond_conv_wgrad_ws_f32(const float* x, const float* dy, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int mode,
                                      int stride, float* dw, float* ws, size_t ws_bytes, void* stream) {
    if (!wgrad_check(x, dy, dw, N, H, W, Cin, Ho, Wo, Cout, mode, stride)) return YOND_EINVAL;
    if (ws && ws_bytes < yond_conv_wgrad_ws_bytes(N, H, W, Cin, Ho, Wo, Cout, mode, stride)) return YOND_EINVAL;
    WgradGeom g{N, H, W, Cin, Ho, Wo, Cout, mode, stride, mode == 0 ? 9 : (mode == 1 ? 4 : 1), 0};
    hipStream_t st = (hipStream_t)stream;
    if (!ws) {                                                       // the atomics path sums into a zeroed output
        hipError_t e = zero_async(dw, (size_t)g.taps * Cout * Cin * sizeof(float), st);
        if (e != hipSuccess) return (int)e;
    }
    const bool two = Cout % 64 == 0;
    if (mode == 0) two ? launch_wgrad<0, 2>(x, dy, g, dw, ws, st) : launch_wgrad<0, 1>(x, dy, g, dw, ws, st);
    else if (mode == 1) launch_wgrad<1, 1>(x, dy, g, dw, ws, st);           // (two tiles per wave: 309 registers, one wave per SIMD)
    else two ? launch_wgrad<2, 2>(x, dy, g, dw, ws, st) : launch_wgrad<2, 1>(x, dy, g, dw, ws, st);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

extern "C" int yond_conv_wgrad_f32(const float* x, const float* dy, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int mode,
                                   int stride, float* dw, void* stream) {
    return yond_conv_wgrad_ws_f32(x, dy, N, H, W, Cin, Ho, Wo, Cout, mode, stride, dw, nullptr, 0, stream);
}

// db[c] = sum over pixels of dY[p][c]  (C a multiple of 32).  A thread owns four channels (one 16-byte load per pixel), a workgroup
// strides over the pixels with 256 / (C / 4) of them per pass (C <= 1024) -- float64 partial sums per thread, added up through LDS,
// one float atomic per channel and workgroup.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ dy, long long npix, int C, float* __restrict__ db) {
    __shared__ double s[256][4];
    const int c4 = C / 4;                                   // threads per pixel
    const int ppw = 256 / c4;                               // pixels per workgroup pass (c4 divides 256 for C = 32 ... 1024: see the launcher)
    const int cl = threadIdx.x % c4, pr = threadIdx.x / c4;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    const long long stride = (long long)gridDim.x * ppw;
    long long p = (long long)blockIdx.x * ppw + pr;
    for (; p + stride < npix; p += 2 * stride) {            // two loads in flight
        const float4 u = *(const float4*)(dy + p * C + cl * 4);
        const float4 v = *(const float4*)(dy + (p + stride) * C + cl * 4);
        a0 += (double)u.x + (double)v.x; a1 += (double)u.y + (double)v.y; a2 += (double)u.z + (double)v.z; a3 += (double)u.w + (double)v.w;
    }
    if (p < npix) {
        const float4 u = *(const float4*)(dy + p * C + cl * 4);
        a0 += (double)u.x; a1 += (double)u.y; a2 += (double)u.z; a3 += (double)u.w;
    }
    s[threadIdx.x][0] = a0; s[threadIdx.x][1] = a1; s[threadIdx.x][2] = a2; s[threadIdx.x][3] = a3;
    __syncthreads();
    if (threadIdx.x < C) {                                  // thread = channel: add the ppw pixel rows
        const int t = threadIdx.x >> 2, e = threadIdx.x & 3;
        double acc = 0.0;
        for (int r = 0; r < ppw; ++r) acc += s[r * c4 + t][e];
        atomicAdd(db + threadIdx.x, (float)acc);
    }
}

// the general form (any C that is a multiple of 32): 32 channels x 8 pixel rows per workgroup
__global__ __launch_bounds__(256) void colsum_wide_kernel(const float* __restrict__ dy, long long npix, int C, float* __restrict__ db) {
    __shared__ double s[8][32];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31), row = threadIdx.x >> 5;
    double acc = 0.0;
    for (long long p = (long long)blockIdx.y * 8 + row; p < npix; p += (long long)gridDim.y * 8) acc += (double)dy[p * C + c];
    s[row][threadIdx.x & 31] = acc;
    __syncthreads();
    if (row == 0) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += s[r][threadIdx.x & 31];
        atomicAdd(db + c, (float)t);
    }
}

extern "C" int yond_colsum_f32(const float* dy, size_t npix, int C, float* db, void* stream) {
    if (!dy || !db || npix == 0 || C <= 0 || C % 32) return YOND_EINVAL;
    hipError_t e = zero_async(db, (size_t)C * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    const int c4 = C / 4;
    if (C <= 256 && 256 % c4 == 0) {                         // C = 32, 64, 128, 256: the 16-byte form (thread = channel needs C <= 256)
        const int ppw = 256 / c4;
        size_t nb = (npix + (size_t)ppw * 8 - 1) / ((size_t)ppw * 8);      // >= 8 pixels per thread
        if (nb > 1024) nb = 1024;
        if (nb < 1) nb = 1;
        hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, dy, (long long)npix, C, db);
    } else {
        size_t gy = (npix + 8 * 64 - 1) / (8 * 64);
        if (gy > 256) gy = 256;
        hipLaunchKernelGGL(colsum_wide_kernel, dim3(C / 32, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, dy, (long long)npix, C, db);
    }
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

// The guided block's middle (archs/modules.py:186-196): out = SiLU(z * tk[n][c] + tb[n][c]) over [N][P pixels][C] with per-image
// (scale, shift) vectors, and its backward in ONE pass over the tensors: with u = z tk + tb, s = sigmoid(u), g = dout * s (1 + u (1 - s)):
// dz = g tk, dtk[n][c] = sum_p g z, dtb[n][c] = sum_p g.  A thread owns four channels (16-byte accesses), a workgroup strides over
// the pixels of one image; the two per-image sums go through LDS and one float atomic per channel and workgroup.
__global__ __launch_bounds__(256) void film_silu_fwd_kernel(const float* __restrict__ z, const float* __restrict__ tk, const float* __restrict__ tb,
                                                            float* __restrict__ out, long long P, int C) {
    const int c4 = C / 4, n = blockIdx.y;
    const long long groups = P * c4;                                   // float4 groups of this image
    const float4* zi = (const float4*)(z + (size_t)n * P * C);
    float4* oi = (float4*)(out + (size_t)n * P * C);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < groups; i += (long long)gridDim.x * 256) {
        const int cl = (int)(i % c4);
        const float4 k = *(const float4*)(tk + (size_t)n * C + cl * 4), b = *(const float4*)(tb + (size_t)n * C + cl * 4);
        const float4 v = zi[i];
        float4 o;
        o.x = silu_f(v.x * k.x + b.x); o.y = silu_f(v.y * k.y + b.y); o.z = silu_f(v.z * k.z + b.z); o.w = silu_f(v.w * k.w + b.w);
        oi[i] = o;
    }
}

__global__ __launch_bounds__(256) void film_silu_bwd_kernel(const float* __restrict__ z, const float* __restrict__ tk, const float* __restrict__ tb,
                                                            const float* __restrict__ dout, float* __restrict__ dz, float* __restrict__ dtk,
                                                            float* __restrict__ dtb, long long P, int C) {
    __shared__ float sk[256][4], sb[256][4];
    const int c4 = C / 4, n = blockIdx.y;
    const int ppw = 256 / c4;                                          // pixels per workgroup pass (c4 divides 256: see the launcher)
    const int cl = threadIdx.x % c4, pr = threadIdx.x / c4;
    const float4 k = *(const float4*)(tk + (size_t)n * C + cl * 4), b = *(const float4*)(tb + (size_t)n * C + cl * 4);
    const float* zi = z + (size_t)n * P * C;
    const float* gi = dout + (size_t)n * P * C;
    float* di = dz + (size_t)n * P * C;
    float ak[4] = {0.0f, 0.0f, 0.0f, 0.0f}, ab[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (long long p = (long long)blockIdx.x * ppw + pr; p < P; p += (long long)gridDim.x * ppw) {
        const float4 v = *(const float4*)(zi + p * C + cl * 4), go = *(const float4*)(gi + p * C + cl * 4);
        const float vv[4] = {v.x, v.y, v.z, v.w}, gg[4] = {go.x, go.y, go.z, go.w}, kk[4] = {k.x, k.y, k.z, k.w}, bb[4] = {b.x, b.y, b.z, b.w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float u = vv[e] * kk[e] + bb[e];
            const float s = 1.0f / (1.0f + expf(-u));
            const float g = gg[e] * (s * (1.0f + u * (1.0f - s)));
            o[e] = g * kk[e];
            ak[e] += g * vv[e];
            ab[e] += g;
        }
        *(float4*)(di + p * C + cl * 4) = make_float4(o[0], o[1], o[2], o[3]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { sk[threadIdx.x][e] = ak[e]; sb[threadIdx.x][e] = ab[e]; }
    __syncthreads();
    for (int ch = threadIdx.x; ch < C; ch += 256) {                    // thread = channel: add the ppw pixel rows
        const int t = ch >> 2, e = ch & 3;
        float a = 0.0f, c = 0.0f;
        for (int r = 0; r < ppw; ++r) { a += sk[r * c4 + t][e]; c += sb[r * c4 + t][e]; }
        atomicAdd(dtk + (size_t)n * C + ch, a);
        atomicAdd(dtb + (size_t)n * C + ch, c);
    }
}

static bool film_silu_shape_ok(int N, long long P, int C) { return N > 0 && P > 0 && C >= 32 && C <= 1024 && C % 32 == 0 && 256 % (C / 4) == 0; }

// 1 when the fused kernels take this channel count (32, 64, 128, 256, 512, 1024), else 0 (the caller keeps its elementwise form)
extern "C" int yond_film_silu_supported(int C) { return film_silu_shape_ok(1, 1, C) ? 1 : 0; }

extern "C" int yond_film_silu_f32(const float* z, const float* tk, const float* tb, float* out, int N, size_t P, int C, void* stream) {
    if (!z || !tk || !tb || !out || !film_silu_shape_ok(N, (long long)P, C)) return YOND_EINVAL;
    size_t nb = (P * (size_t)(C / 4) + 256 * 8 - 1) / (256 * 8);
    if (nb > 2048 / (size_t)N + 1) nb = 2048 / (size_t)N + 1;
    hipLaunchKernelGGL(film_silu_fwd_kernel, dim3((unsigned)nb, (unsigned)N), dim3(256), 0, (hipStream_t)stream, z, tk, tb, out, (long long)P, C);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

extern "C" int yond_film_silu_bwd_f32(const float* z, const float* tk, const float* tb, const float* dout, float* dz, float* dtk, float* dtb,
                                      int N, size_t P, int C, void* stream) {
    if (!z || !tk || !tb || !dout || !dz || !dtk || !dtb || !film_silu_shape_ok(N, (long long)P, C)) return YOND_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = zero_async(dtk, (size_t)N * C * sizeof(float), st);
    if (e == hipSuccess) e = zero_async(dtb, (size_t)N * C * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    const int ppw = 256 / (C / 4);
    size_t nb = (P + (size_t)ppw * 8 - 1) / ((size_t)ppw * 8);
    if (nb > 2048 / (size_t)N + 1) nb = 2048 / (size_t)N + 1;
    hipLaunchKernelGGL(film_silu_bwd_kernel, dim3((unsigned)nb, (unsigned)N), dim3(256), 0, st, z, tk, tb, dout, dz, dtk, dtb, (long long)P, C);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

// ---- the sigma-conditioning of a guided block for training (archs/modules.py:170-178, 186-196) ------------------------------------
//   gamma: Conv2d(1, C, 1) -> SiLU -> Conv2d(C, C, 1);  beta: SiLU -> Conv2d(C, C, 1), applied to t [B][1][1][1]:
//   a = t w1 + b1, h = SiLU(a), tk = W2 h + b2, s = SiLU(tk), tb = W3 s + b3          (three tiny linear layers per block)
// torch.autograd ran them as ~8 broadcast / reduce kernels forward and ~16 backward per block over [B][C][C] temporaries (67 MB at
// C = 512): 1.2 ms of a 16 ms step.  Here they are six small matrix products (M, N <= 512, K <= 512) on one LDS-tiled kernel with
// the element functions (SiLU of the operand, SiLU' in the epilogue) folded in: two launches forward, five backward.  (A first
// version with a workgroup per batch item and a wave per output row was latency bound: 430 us per call.)  tk / tb are written at
// row stride ld >= C, zero beyond C (channel padding).
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float dsilu_f(float u) {
    const float s = 1.0f / (1.0f + expf(-u));
    return s * (1.0f + u * (1.0f - s));
}

// One tile kernel for the six small products of the MLPs: out[m][n] = sum_k A(m, k) B(k, n), 16 x 16 outputs per workgroup
// (thread = one output), K in chunks of 64 through LDS; the operands' element functions and the epilogue are chosen by MODE:
//   0  tk[b][j]      = sum_k SiLU(t_b w1_k + b1_k) W2[j][k] + b2[j]                     (M = B, N = C, K = C)
//   1  tb[b][j]      = sum_k SiLU(tk[b][k])        W3[j][k] + b3[j]
//   2  dtk_tot[b][k] = dtk[b][k] + SiLU'(tk[b][k]) sum_j dtb[b][j] W3[j][k]
//   3  da[b][k]      = SiLU'(t_b w1_k + b1_k)      sum_j dtk_tot[b][j] W2[j][k]
//   4  dW3[j][k]     = sum_b dtb[b][j] SiLU(tk[b][k])                                    (M = C, N = C, K = B)
//   5  dW2[j][k]     = sum_b dtk_tot[b][j] SiLU(t_b w1_k + b1_k)
struct FilmMlpArgs {
    const float *t, *w1, *b1, *W2, *b2, *W3, *b3;     // parameters ([C], [C][C])
    const float *tk, *dtk, *dtb;                        // [B][ld]
    const float *dtk_tot;                               // [B][C]
    float* out;
    int B, C, ld;
};
template <int MODE>
__device__ __forceinline__ void film_mlp_tile_body(const FilmMlpArgs& a, int bx, int by) {
    __shared__ float As[16][65], Bs[64][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int M = MODE < 4 ? a.B : a.C, N = a.C, K = MODE < 4 ? a.C : a.B;
    const int m0 = by * 16, n0 = bx * 16;
    auto hval = [&](int b, int k) { return silu_f(a.t[b] * a.w1[k] + a.b1[k]); };
    auto A_at = [&](int m, int k) -> float {
        if (m >= M || k >= K) return 0.0f;
        if constexpr (MODE == 0) return hval(m, k);
        if constexpr (MODE == 1) return silu_f(a.tk[(size_t)m * a.ld + k]);
        if constexpr (MODE == 2) return a.dtb[(size_t)m * a.ld + k];
        if constexpr (MODE == 3) return a.dtk_tot[(size_t)m * a.C + k];
        if constexpr (MODE == 4) return a.dtb[(size_t)k * a.ld + m];
        return a.dtk_tot[(size_t)k * a.C + m];
    };
    auto B_at = [&](int k, int n) -> float {
        if (n >= N || k >= K) return 0.0f;
        if constexpr (MODE == 0) return a.W2[(size_t)n * a.C + k];
        if constexpr (MODE == 1) return a.W3[(size_t)n * a.C + k];
        if constexpr (MODE == 2) return a.W3[(size_t)k * a.C + n];
        if constexpr (MODE == 3) return a.W2[(size_t)k * a.C + n];
        if constexpr (MODE == 4) return silu_f(a.tk[(size_t)k * a.ld + n]);
        return hval(k, n);
    };
    float acc = 0.0f;
    for (int k0 = 0; k0 < K; k0 += 64) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = threadIdx.x + e * 256;
            // A chunk [16 m][64 k]: consecutive threads along k for the row-major operands (modes 0-3), along m for the batch-major ones
            if constexpr (MODE < 4) As[i >> 6][i & 63] = A_at(m0 + (i >> 6), k0 + (i & 63));
            else As[i & 15][i >> 4] = A_at(m0 + (i & 15), k0 + (i >> 4));
            // B chunk [64 k][16 n]: W[n][k] (modes 0, 1) is contiguous along k, the others along n
            if constexpr (MODE < 2) Bs[i & 63][i >> 6] = B_at(k0 + (i & 63), n0 + (i >> 6));
            else Bs[i >> 4][i & 15] = B_at(k0 + (i >> 4), n0 + (i & 15));
        }
        __syncthreads();
#pragma unroll 16
        for (int k = 0; k < 64; ++k) acc += As[ty][k] * Bs[k][tx];
        __syncthreads();
    }
    const int m = m0 + ty, n = n0 + tx;
    if (m >= M || n >= N) return;
    if constexpr (MODE == 0) a.out[(size_t)m * a.ld + n] = acc + a.b2[n];
    else if constexpr (MODE == 1) a.out[(size_t)m * a.ld + n] = acc + a.b3[n];
    else if constexpr (MODE == 2) a.out[(size_t)m * a.C + n] = a.dtk[(size_t)m * a.ld + n] + acc * dsilu_f(a.tk[(size_t)m * a.ld + n]);
    else if constexpr (MODE == 3) a.out[(size_t)m * a.C + n] = acc * dsilu_f(a.t[m] * a.w1[n] + a.b1[n]);
    else a.out[(size_t)m * a.C + n] = acc;
}
template <int MODE>
__global__ __launch_bounds__(256) void film_mlp_tile_kernel(const FilmMlpArgs a) {
    film_mlp_tile_body<MODE>(a, blockIdx.x, blockIdx.y);
}
// ... and for ALL guided blocks of a network in one launch (their MLPs depend on sigma and the weights only: every block's forward
// can run before the first convolution, every block's backward after the last): nine blocks x (2 + 5) launches of ~12 us each -- latency,
// not work -- become 2 + 5 launches as long as the largest block's.  The descriptors travel in the kernel arguments.
#define FILM_MULTI_MAX 12
struct FilmMlpMulti {
    FilmMlpArgs a[FILM_MULTI_MAX];
    int tile0[FILM_MULTI_MAX + 1];                      // first workgroup of entry e (prefix sums of the entries' tile counts)
    int n;
};
__device__ __forceinline__ int film_multi_entry(const FilmMlpMulti& mm, int wg) {
    int e = 0;
    for (int i = 1; i < mm.n; ++i) e += (wg >= mm.tile0[i]) ? 1 : 0;
    return e;
}
template <int MODE>
__global__ __launch_bounds__(256) void film_mlp_tile_multi_kernel(const FilmMlpMulti mm) {
    const int e = film_multi_entry(mm, blockIdx.x);
    const int local = blockIdx.x - mm.tile0[e];
    const int nx = (mm.a[e].C + 15) / 16;
    film_mlp_tile_body<MODE>(mm.a[e], local % nx, local / nx);
}

// the five parameter vectors' gradients (sums over the batch) and the zero padding of tk / tb rows
// (a workgroup = 64 channels x 4 quarters of the batch, joined in LDS: one thread per channel over the whole batch was latency bound)
__device__ __forceinline__ void film_mlp_vec_body(const float* __restrict__ t, const float* __restrict__ dtb, const float* __restrict__ dtk_tot,
                                                  const float* __restrict__ da, int B, int C, int ld, float* __restrict__ db3,
                                                  float* __restrict__ db2, float* __restrict__ dw1, float* __restrict__ db1, int bx) {
    __shared__ float part[4][4][64];
    const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = bx * 64 + e;
    float s3 = 0.0f, s2 = 0.0f, sw = 0.0f, sb = 0.0f;
    if (i < C)
        for (int b = q; b < B; b += 4) {
            s3 += dtb[(size_t)b * ld + i];
            s2 += dtk_tot[(size_t)b * C + i];
            const float d = da[(size_t)b * C + i];
            sw += d * t[b];
            sb += d;
        }
    part[q][0][e] = s3; part[q][1][e] = s2; part[q][2][e] = sw; part[q][3][e] = sb;
    __syncthreads();
    if (i < C && q < 4) {
        const float v = (part[0][q][e] + part[1][q][e]) + (part[2][q][e] + part[3][q][e]);
        (q == 0 ? db3 : q == 1 ? db2 : q == 2 ? dw1 : db1)[i] = v;
    }
}
__global__ __launch_bounds__(256) void film_mlp_vec_kernel(const float* __restrict__ t, const float* __restrict__ dtb, const float* __restrict__ dtk_tot,
                                                           const float* __restrict__ da, int B, int C, int ld, float* __restrict__ db3,
                                                           float* __restrict__ db2, float* __restrict__ dw1, float* __restrict__ db1) {
    film_mlp_vec_body(t, dtb, dtk_tot, da, B, C, ld, db3, db2, dw1, db1, blockIdx.x);
}
struct FilmVecMulti {
    const float *t[FILM_MULTI_MAX], *dtb[FILM_MULTI_MAX], *dtk_tot[FILM_MULTI_MAX], *da[FILM_MULTI_MAX];
    float *db3[FILM_MULTI_MAX], *db2[FILM_MULTI_MAX], *dw1[FILM_MULTI_MAX], *db1[FILM_MULTI_MAX];
    int B[FILM_MULTI_MAX], C[FILM_MULTI_MAX], ld[FILM_MULTI_MAX];
    int tile0[FILM_MULTI_MAX + 1];
    int n;
};
__global__ __launch_bounds__(256) void film_mlp_vec_multi_kernel(const FilmVecMulti vm) {
    int e = 0;
    for (int i = 1; i < vm.n; ++i) e += ((int)blockIdx.x >= vm.tile0[i]) ? 1 : 0;
    film_mlp_vec_body(vm.t[e], vm.dtb[e], vm.dtk_tot[e], vm.da[e], vm.B[e], vm.C[e], vm.ld[e], vm.db3[e], vm.db2[e], vm.dw1[e], vm.db1[e],
                      (int)blockIdx.x - vm.tile0[e]);
}
__global__ __launch_bounds__(256) void film_mlp_pad_kernel(float* __restrict__ tk, float* __restrict__ tb, int B, int C, int ld) {
    const int i = blockIdx.x * 256 + threadIdx.x, w = ld - C;
    if (w <= 0 || i >= B * w) return;
    const int b = i / w, j = C + i % w;
    tk[(size_t)b * ld + j] = 0.0f;
    tb[(size_t)b * ld + j] = 0.0f;
}

template <int MODE>
static void film_mlp_launch(const FilmMlpArgs& a, hipStream_t st) {
    const int M = MODE < 4 ? a.B : a.C;
    hipLaunchKernelGGL(film_mlp_tile_kernel<MODE>, dim3((unsigned)((a.C + 15) / 16), (unsigned)((M + 15) / 16)), dim3(256), 0, st, a);
}

extern "C" int yond_film_mlp_fwd_f32(const float* t, const float* w1, const float* b1, const float* W2, const float* b2, const float* W3,
                                     const float* b3, int B, int C, int ld, float* tk, float* tb, void* stream) {
    if (!t || !w1 || !b1 || !W2 || !b2 || !W3 || !b3 || !tk || !tb || B <= 0 || C <= 0 || ld < C) return YOND_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    FilmMlpArgs a{t, w1, b1, W2, b2, W3, b3, tk, nullptr, nullptr, nullptr, tk, B, C, ld};
    if (ld > C) hipLaunchKernelGGL(film_mlp_pad_kernel, dim3((unsigned)((B * (ld - C) + 255) / 256)), dim3(256), 0, st, tk, tb, B, C, ld);
    film_mlp_launch<0>(a, st);
    a.out = tb;
    film_mlp_launch<1>(a, st);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

// scratch: 2 * B * C floats (dtk_tot, da)
extern "C" int yond_film_mlp_bwd_f32(const float* t, const float* w1, const float* b1, const float* W2, const float* W3, const float* tk,
                                     const float* dtk, const float* dtb, int B, int C, int ld, float* scratch, float* dw1, float* db1,
                                     float* dW2, float* db2, float* dW3, float* db3, void* stream) {
    if (!t || !w1 || !b1 || !W2 || !W3 || !tk || !dtk || !dtb || !scratch || !dw1 || !db1 || !dW2 || !db2 || !dW3 || !db3 || B <= 0 || C <= 0 ||
        ld < C)
        return YOND_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    float* dtk_tot = scratch;
    float* da = scratch + (size_t)B * C;
    FilmMlpArgs a{t, w1, b1, W2, nullptr, W3, nullptr, tk, dtk, dtb, dtk_tot, dtk_tot, B, C, ld};
    film_mlp_launch<2>(a, st);
    a.out = da;
    film_mlp_launch<3>(a, st);
    a.out = dW3;
    film_mlp_launch<4>(a, st);
    a.out = dW2;
    film_mlp_launch<5>(a, st);
    hipLaunchKernelGGL(film_mlp_vec_kernel, dim3((unsigned)((C + 63) / 64)), dim3(256), 0, st, t, dtb, dtk_tot, da, B, C, ld, db3, db2, dw1, db1);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

// The same for n <= 12 blocks at once (host array of descriptors; tk / tb rows beyond C must be zero already: the caller clears them).
template <int MODE>
static void film_mlp_launch_multi(const YondFilmMlpDesc* d, int n, hipStream_t st) {
    FilmMlpMulti mm;
    int tiles = 0;
    for (int e = 0; e < n; ++e) {
        const YondFilmMlpDesc& q = d[e];
        float* dtk_tot = q.scratch;
        float* da = q.scratch ? q.scratch + (size_t)q.B * q.C : nullptr;
        float* out = MODE == 0 ? q.tk : MODE == 1 ? q.tb : MODE == 2 ? dtk_tot : MODE == 3 ? da : MODE == 4 ? q.dW3 : q.dW2;
        mm.a[e] = FilmMlpArgs{q.t, q.w1, q.b1, q.W2, q.b2, q.W3, q.b3, q.tk, q.dtk, q.dtb, dtk_tot, out, q.B, q.C, q.ld};
        mm.tile0[e] = tiles;
        const int M = MODE < 4 ? q.B : q.C;
        tiles += ((q.C + 15) / 16) * ((M + 15) / 16);
    }
    mm.tile0[n] = tiles;
    mm.n = n;
    hipLaunchKernelGGL(film_mlp_tile_multi_kernel<MODE>, dim3((unsigned)tiles), dim3(256), 0, st, mm);
}
static int film_multi_check(const YondFilmMlpDesc* d, int n, bool bwd) {
    if (!d || n < 1 || n > FILM_MULTI_MAX) return YOND_EINVAL;
    for (int e = 0; e < n; ++e) {
        const YondFilmMlpDesc& q = d[e];
        if (!q.t || !q.w1 || !q.b1 || !q.W2 || !q.W3 || !q.tk || q.B <= 0 || q.C <= 0 || q.ld < q.C) return YOND_EINVAL;
        if (!bwd && (!q.b2 || !q.b3 || !q.tb)) return YOND_EINVAL;
        if (bwd && (!q.dtk || !q.dtb || !q.scratch || !q.dw1 || !q.db1 || !q.dW2 || !q.db2 || !q.dW3 || !q.db3)) return YOND_EINVAL;
    }
    return YOND_OK;
}
extern "C" int yond_film_mlp_fwd_multi_f32(const YondFilmMlpDesc* d, int n, void* stream) {
    if (int rc = film_multi_check(d, n, false)) return rc;
    film_mlp_launch_multi<0>(d, n, (hipStream_t)stream);
    film_mlp_launch_multi<1>(d, n, (hipStream_t)stream);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}
extern "C" int yond_film_mlp_bwd_multi_f32(const YondFilmMlpDesc* d, int n, void* stream) {
    if (int rc = film_multi_check(d, n, true)) return rc;
    hipStream_t st = (hipStream_t)stream;
    film_mlp_launch_multi<2>(d, n, st);
    film_mlp_launch_multi<3>(d, n, st);
    film_mlp_launch_multi<4>(d, n, st);
    film_mlp_launch_multi<5>(d, n, st);
    FilmVecMulti vm;
    int tiles = 0;
    for (int e = 0; e < n; ++e) {
        const YondFilmMlpDesc& q = d[e];
        vm.t[e] = q.t; vm.dtb[e] = q.dtb; vm.dtk_tot[e] = q.scratch; vm.da[e] = q.scratch + (size_t)q.B * q.C;
        vm.db3[e] = q.db3; vm.db2[e] = q.db2; vm.dw1[e] = q.dw1; vm.db1[e] = q.db1;
        vm.B[e] = q.B; vm.C[e] = q.C; vm.ld[e] = q.ld;
        vm.tile0[e] = tiles;
        tiles += (q.C + 63) / 64;
    }
    vm.tile0[n] = tiles;
    vm.n = n;
    hipLaunchKernelGGL(film_mlp_vec_multi_kernel, dim3((unsigned)tiles), dim3(256), 0, st, vm);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

// y = SiLU(x) (the operand of a guided block's first convolution, archs/modules.py:186-188)
__global__ __launch_bounds__(256) void silu_kernel(const float4* __restrict__ x, float4* __restrict__ y, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = x[i];
        y[i] = make_float4(silu_f(v.x), silu_f(v.y), silu_f(v.z), silu_f(v.w));
    }
}
extern "C" int yond_silu_f32(const float* x, float* y, size_t n, void* stream) {
    if (!x || !y || n == 0 || n % 4) return YOND_EINVAL;
    size_t nb = (n / 4 + 256 * 4 - 1) / (256 * 4);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(silu_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const float4*)x, (float4*)y, n / 4);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

// A residual block's input side in backward: dx = dres + dz SiLU'(x) in one pass (the block computes conv1(SiLU(x)) and adds x to its
// output: autograd ran silu_backward and the accumulation of the two gradients as two kernels over three / three tensors)
__global__ __launch_bounds__(256) void silu_bwd_add_kernel(const float4* __restrict__ x, const float4* __restrict__ dz, const float4* __restrict__ dres,
                                                           float4* __restrict__ dx, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 xv = x[i], g = dz[i], r = dres[i];
        float4 o;
        o.x = r.x + g.x * dsilu_f(xv.x); o.y = r.y + g.y * dsilu_f(xv.y); o.z = r.z + g.z * dsilu_f(xv.z); o.w = r.w + g.w * dsilu_f(xv.w);
        dx[i] = o;
    }
}
extern "C" int yond_silu_bwd_add_f32(const float* x, const float* dz, const float* dres, float* dx, size_t n, void* stream) {
    if (!x || !dz || !dres || !dx || n == 0 || n % 4) return YOND_EINVAL;
    size_t nb = (n / 4 + 256 * 4 - 1) / (256 * 4);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(silu_bwd_add_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const float4*)x, (const float4*)dz,
                       (const float4*)dres, (float4*)dx, n / 4);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

// The data gradient of a stride-2 layer is a stride-1 convolution over the gradient with zeros between its pixels (train.py): g[n][y][x] =
// dy[n][y / 2][x / 2] at even (y, x), 0 elsewhere -- ONE pass (torch: a zero fill of g plus a strided copy).
__global__ __launch_bounds__(256) void zero_interleave_kernel(const float4* __restrict__ dy, int Ho, int Wo, int H, int W, int c4, size_t n4,
                                                              float4* __restrict__ g) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % c4);
        size_t p = i / c4;
        const int x = (int)(p % W); p /= W;
        const int y = (int)(p % H);
        const size_t n = p / H;
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (!(x & 1) && !(y & 1)) v = dy[((n * Ho + (y >> 1)) * Wo + (x >> 1)) * c4 + c];
        g[i] = v;
    }
}
extern "C" int yond_zero_interleave_f32(const float* dy, int N, int Ho, int Wo, int C, int H, int W, float* g, void* stream) {
    if (!dy || !g || N <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || C % 4 || Ho != (H + 1) / 2 || Wo != (W + 1) / 2) return YOND_EINVAL;
    const size_t n4 = (size_t)N * H * W * (C / 4);
    size_t nb = (n4 + 256 * 4 - 1) / (256 * 4);
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(zero_interleave_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const float4*)dy, Ho, Wo, H, W, C / 4, n4,
                       (float4*)g);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

// F.l1_loss (mean reduction) and its gradient: loss_sum += sum |pred - target| (float64), grad = sign(pred - target) * gscale
__global__ __launch_bounds__(256) void l1_kernel(const float* __restrict__ pred, const float* __restrict__ target, size_t n, float gscale,
                                                 double* __restrict__ loss_sum, float* __restrict__ grad) {
    __shared__ double s_red[4];
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float d = pred[i] - target[i];
        acc += (double)fabsf(d);
        if (grad) grad[i] = d > 0.0f ? gscale : (d < 0.0f ? -gscale : 0.0f);          // torch: sign(0) = 0
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss_sum, s_red[0] + s_red[1] + s_red[2] + s_red[3]);
}

extern "C" int yond_l1_loss_f32(const float* pred, const float* target, size_t n, double* loss_sum, float* grad, void* stream) {
    if (!pred || !target || !loss_sum || n == 0) return YOND_EINVAL;
    hipError_t e = zero_async(loss_sum, sizeof(double), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    size_t nb = (n + 255) / 256;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(l1_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, pred, target, n, 1.0f / (float)n, loss_sum, grad);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

// L1_Charbonnier_loss (losses/base_loss.py:69-79): error = sqrt(diff * diff + eps), loss = mean(error); the gradient in the
// float32 steps of torch's backward: grad_u = (1 / n) / (2 error) through the sqrt, then grad_u * diff from EACH operand of
// diff * diff, summed
__global__ __launch_bounds__(256) void charbonnier_kernel(const float* __restrict__ pred, const float* __restrict__ target, size_t n, float eps,
                                                          float gscale, double* __restrict__ loss_sum, float* __restrict__ grad) {
    __shared__ double s_red[4];
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float d = __fsub_rn(pred[i], target[i]);
        const float e = __fsqrt_rn(__fadd_rn(__fmul_rn(d, d), eps));
        acc += (double)e;
        if (grad) {
            const float gu = __fdiv_rn(gscale, __fmul_rn(2.0f, e));
            const float t = __fmul_rn(gu, d);
            grad[i] = __fadd_rn(t, t);
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss_sum, s_red[0] + s_red[1] + s_red[2] + s_red[3]);
}

extern "C" int yond_charbonnier_loss_f32(const float* pred, const float* target, size_t n, double eps, double* loss_sum, float* grad,
                                         void* stream) {
    if (!pred || !target || !loss_sum || n == 0 || !(eps > 0.0)) return YOND_EINVAL;
    hipError_t e = zero_async(loss_sum, sizeof(double), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    size_t nb = (n + 255) / 256;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(charbonnier_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, pred, target, n, (float)eps,
                       1.0f / (float)n, loss_sum, grad);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

// torch.optim.Adam.step (single tensor, no weight decay, no amsgrad), float32 state as torch keeps it:
//   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g g;  p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, size_t n, float b1, float b2, float step_size, float inv_bc2_sqrt,
                                                   float eps) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float gi = g[i];
        // torch: exp_avg.lerp_(grad, 1 - beta1) = m + (1 - b1) (g - m); exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
        const float mi = __fadd_rn(m[i], __fmul_rn(1.0f - b1, __fsub_rn(gi, m[i])));
        const float vi = __fadd_rn(__fmul_rn(v[i], b2), __fmul_rn(__fmul_rn(gi, gi), 1.0f - b2));
        m[i] = mi;
        v[i] = vi;
        const float denom = __fadd_rn(__fmul_rn(__fsqrt_rn(vi), inv_bc2_sqrt), eps);
        p[i] = __fsub_rn(p[i], __fmul_rn(step_size, __fdiv_rn(mi, denom)));
    }
}

// The same update with the step's two scalars read from the device -- hyp[0] = lr / (1 - beta1^t), hyp[1] = 1 / sqrt(1 - beta2^t), computed
// by the host in float64 exactly as below and rounded to float32 -- so that a CAPTURED step (hipGraph) can be replayed with another
// learning rate and step count; status (optional, three words): the update is skipped when bit 0 of any is set (a gradient, a weight left
// fp16's range in the split-operand kernels; the caller's own word, e.g. a non-finite loss: the host lowers the loss scale and redoes the step).
__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, size_t n, float b1, float b2, const float* __restrict__ hyp, float eps,
                                                       const int* __restrict__ status) {
    if (status && ((status[0] | status[1] | status[2]) & 1)) return;
    const float step_size = hyp[0], inv_bc2_sqrt = hyp[1];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float gi = g[i];
        const float mi = __fadd_rn(m[i], __fmul_rn(1.0f - b1, __fsub_rn(gi, m[i])));
        const float vi = __fadd_rn(__fmul_rn(v[i], b2), __fmul_rn(__fmul_rn(gi, gi), 1.0f - b2));
        m[i] = mi;
        v[i] = vi;
        const float denom = __fadd_rn(__fmul_rn(__fsqrt_rn(vi), inv_bc2_sqrt), eps);
        p[i] = __fsub_rn(p[i], __fmul_rn(step_size, __fdiv_rn(mi, denom)));
    }
}
extern "C" int yond_adam_step_dev_f32(float* p, const float* g, float* m, float* v, size_t n, double beta1, double beta2, double eps,
                                      const float* hyp, const int* status, void* stream) {
    if (!p || !g || !m || !v || !hyp || n == 0) return YOND_EINVAL;
    size_t nb = (n + 255) / 256;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(adam_dev_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, (float)beta1, (float)beta2, hyp,
                       (float)eps, status);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

extern "C" int yond_adam_step_f32(float* p, const float* g, float* m, float* v, size_t n, double lr, double beta1, double beta2, double eps,
                                  int step, void* stream) {
    if (!p || !g || !m || !v || n == 0 || step < 1) return YOND_EINVAL;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    size_t nb = (n + 255) / 256;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, (float)beta1, (float)beta2,
                       (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), (float)eps);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}
