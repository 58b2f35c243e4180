// The bias LUT as a prepared table (shared by K1 in vst.hip and the frame chain's kernel in bias_lut.hip).
#pragma once
#include "common.h"

#define LUT_MAX 4096
#define LUT_MAXSEG 8
#define LUT_BYTES_PER_KNOT 24

// The bias LUT (utils/isp_algos.py:103-108, 128: interp1d over knots that are runs of evenly spaced values, step 0.1 /
// 1 / 10) is a continuous piecewise-linear function, evaluated per pixel as a + b * x with the interval's coefficients
//   b_i = (y_i - y_{i-1}) / (x_i - x_{i-1})   (the float32 difference of the float32 ordinates, as interp1d forms it),
//   a_i = y_{i-1} - b_i x_{i-1}                (float64)
// from a 16-byte LDS entry.  The interval index comes from the run's spacing with one multiply; because the function is
// continuous, landing in the neighbouring interval when x sits within rounding distance of a knot changes the value by
// < 1e-12, far below the float32 rounding of K1's output -- so no search / repair against the knots is needed (the first
// version spent most of K1's time there, in float64 sqrt and in the float64 divide).  Knots that are not <= 8 evenly
// spaced runs fall back to bisection.
struct LutLds {
    double* x;                   // [n]  knots (bisection fallback)
    double2* ab;                 // [n]  coefficients of the interval that ENDS at knot i (i >= 1)
    float seg_x[LUT_MAXSEG], seg_inv[LUT_MAXSEG];
    int seg_i[LUT_MAXSEG + 1];
    int nseg;                    // 0: bisection
    int nbreak;
};

// flags: bit 0 -- ordinates are float64 (the 2-D BiasLUT's merged row) instead of float32 (get_bias' interp1d knots);
//        bit 1 -- BiasLUT semantics beyond the last knot (utils/isp_algos.py:188-194, 226-230): the last ordinate up to one
//                 more interval, Foi's closed form (float32-rounded, as the reference stores it) further out
#define LUT_Y64 1
#define LUT_BIASLUT 2
__device__ __forceinline__ void lut_prepare(LutLds& L, const double* __restrict__ lut_x, const void* __restrict__ lut_yv, int n,
                                            int flags) {
    const int tid = threadIdx.x;
    const float* lut_y = (const float*)lut_yv;
    const double* lut_y64 = (const double*)lut_yv;
    for (int i = tid; i < n; i += 256) L.x[i] = lut_x[i];
    if (tid == 0) { L.nbreak = 0; L.nseg = 0; }
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        if (i >= 1) {
            double dy, y0;
            if (flags & LUT_Y64) { dy = lut_y64[i] - lut_y64[i - 1]; y0 = lut_y64[i - 1]; }
            else { dy = (double)(lut_y[i] - lut_y[i - 1]); y0 = (double)lut_y[i - 1]; }   // float32 difference, as interp1d forms it
            double dx = L.x[i] - L.x[i - 1];
            if (dx == 0.0 && i >= 2) {
                // a repeated knot (get_bias concatenates its runs, so 50 and 500 appear twice): searchsorted('left') never
                // selects the empty interval -- a query equal to the knot belongs to the interval that ENDS at its first copy
                dx = L.x[i - 1] - L.x[i - 2];
                if (flags & LUT_Y64) { dy = lut_y64[i - 1] - lut_y64[i - 2]; y0 = lut_y64[i - 2]; }
                else { dy = (double)(lut_y[i - 1] - lut_y[i - 2]); y0 = (double)lut_y[i - 2]; }
                const double b = dy / dx;
                L.ab[i] = make_double2(y0 - b * L.x[i - 2], b);
            } else {
                const double b = dy / dx;
                L.ab[i] = make_double2(y0 - b * L.x[i - 1], b);
            }
        }
        if (i >= 1 && i + 1 < n) {
            const double d0 = L.x[i] - L.x[i - 1], d1 = L.x[i + 1] - L.x[i];
            if (fabs(d1 - d0) > 1e-3 * fabs(d0)) {                 // the spacing changes at knot i
                const int slot = atomicAdd(&L.nbreak, 1);
                if (slot < LUT_MAXSEG - 1) L.seg_i[slot + 1] = i;
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        const int nb = L.nbreak;
        if (nb <= LUT_MAXSEG - 1 && n >= 2) {
            L.seg_i[0] = 0;
            for (int a = 2; a <= nb; ++a) {                        // insertion sort of <= 7 break indices
                const int v = L.seg_i[a];
                int j = a - 1;
                while (j >= 1 && L.seg_i[j] > v) { L.seg_i[j + 1] = L.seg_i[j]; --j; }
                L.seg_i[j + 1] = v;
            }
            L.seg_i[nb + 1] = n - 1;
            bool ok = true;
            for (int sgm = 0; sgm <= nb; ++sgm) {
                const int i0 = L.seg_i[sgm];
                L.seg_x[sgm] = (float)L.x[i0];
                L.seg_inv[sgm] = (float)(1.0 / (L.x[i0 + 1] - L.x[i0]));
                ok = ok && ((double)L.seg_x[sgm] == L.x[i0]);      // run starts must be float32 values (0, 50, 500 are)
            }
            L.nseg = ok ? nb + 1 : 0;
        }
    }
    __syncthreads();
    // The index guess of lut_eval trusts that every run is EVENLY spaced.  The break detection above compares neighbouring
    // intervals with a relative tolerance, so a slowly drifting grid (a log grid of ratio < 1.001) would pass as one run:
    // check every knot against its run's ideal position x[i0] + (i - i0) * step and fall back to the bisection (exact for
    // any spacing) if one is off by more than 1e-3 of a step.  (float64 np.linspace knots are exact to ~1e-13 of a step;
    // the float32 ones NumPy 2 produces for a float32 maximum -- 500 ... ub in steps of ~9.8 -- to 6e-6 of a step; a query
    // that close to a knot may be evaluated on the neighbouring interval, which differs from the right one by the change of
    // slope times that distance: < 1e-9 on these tables.  A drifting grid is off by whole steps after a few hundred knots.)
    const int nseg = L.nseg;
    if (nseg > 0) {
        bool bad = false;
        for (int i = tid; i < n; i += 256) {
            int sgm = 0;
            for (int a = 1; a < nseg; ++a) sgm += (i > L.seg_i[a]) ? 1 : 0;
            const int i0 = L.seg_i[sgm];
            const double step = L.x[i0 + 1] - L.x[i0];
            // (a repeated knot -- get_bias' 50 and 500 -- opens a run: i0 is its second copy, the first copy closes the previous run)
            const double dev = fabs(L.x[i] - (L.x[i0] + (double)(i - i0) * step));
            bad = bad || !(dev <= 1e-3 * fabs(step));          // (the zero-width run between the two copies: dev = 0)
        }
        if (__syncthreads_or(bad ? 1 : 0)) {
            if (tid == 0) L.nseg = 0;
        }
        __syncthreads();
    }
}

// ---- the prepared table in global memory (yond_lut_table_f64): header, coefficients, knots ----
struct LutHeader {
    int n, nseg, nbreak, pad;
    float seg_x[LUT_MAXSEG], seg_inv[LUT_MAXSEG];
    int seg_i[LUT_MAXSEG + 1];
    int pad2[7];
};
static_assert(sizeof(LutHeader) == 144, "LutHeader layout");
__device__ __forceinline__ void lut_load(LutLds& L, const void* __restrict__ ws, int& n_out) {
    // plain copy of the image yond_lut_table_f64 stored: no arithmetic per workgroup
    const LutHeader* hd = (const LutHeader*)ws;
    const int n = hd->n;
    const double2* ab = (const double2*)((const char*)ws + sizeof(LutHeader));
    const double* x = (const double*)(ab + LUT_MAX);
    (void)x;                                          // (the knots themselves stay in global memory: the table's runs are even,
    for (int i = threadIdx.x; i < n; i += blockDim.x) L.ab[i] = ab[i];      //  yond_lut_table_f64 flags a grid that is not)
    if (threadIdx.x < LUT_MAXSEG) { L.seg_x[threadIdx.x] = hd->seg_x[threadIdx.x]; L.seg_inv[threadIdx.x] = hd->seg_inv[threadIdx.x]; }
    if (threadIdx.x <= LUT_MAXSEG) L.seg_i[threadIdx.x] = hd->seg_i[threadIdx.x];
    if (threadIdx.x == 0) { L.nseg = hd->nseg; L.nbreak = hd->nbreak; }
    n_out = n;
    __syncthreads();
}


// the image of a prepared table in global memory: what yond_lut_table_f64 / the frame chain store and lut_load copies
__device__ __forceinline__ void lut_table_store(const LutLds& L, void* __restrict__ ws, int n) {
    LutHeader* hd = (LutHeader*)ws;
    double2* ab = (double2*)((char*)ws + sizeof(LutHeader));
    double* x = (double*)(ab + LUT_MAX);
    for (int i = threadIdx.x; i < n; i += 256) { ab[i] = i >= 1 ? L.ab[i] : make_double2(0.0, 0.0); x[i] = L.x[i]; }
    if (threadIdx.x < LUT_MAXSEG) { hd->seg_x[threadIdx.x] = L.seg_x[threadIdx.x]; hd->seg_inv[threadIdx.x] = L.seg_inv[threadIdx.x]; }
    if (threadIdx.x <= LUT_MAXSEG) hd->seg_i[threadIdx.x] = L.seg_i[threadIdx.x];
    if (threadIdx.x == 0) { hd->n = n; hd->nseg = L.nseg; hd->nbreak = L.nbreak; }
}
