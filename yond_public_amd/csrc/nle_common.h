// Shared pieces of the noise-level estimator's kernels (nle.hip, nle_select.hip, nle_fast.hip).
#pragma once
#include <stddef.h>
#include <math.h>
#include "common.h"

// order-preserving 32-bit key of a float (total order of floats as unsigned) and its inverse
__device__ __forceinline__ unsigned int f2key(float f) {
    const unsigned int b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned int k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

__device__ __forceinline__ unsigned long long wave_incl_scan_u64(unsigned long long v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long up = __shfl_up(v, o);
        if (lane >= o) v += up;
    }
    return v;
}

// ---- state of one threshold estimate (yond_nle_*; nle_fast.hip, written to by the fused box kernel of nle.hip) ----
#define NF_MAXT 64            // order statistics per call (two per quantile)
#define NF_MAXQ 32            // quantiles per call
#define NF_BINS 1024          // mean bins: 1001 used (np.bincount(minlength=nbins+1), YOND_SIDD.py:40)
#define NF_L1 65536           // level-1 histogram: key >> 16
#define NF_WIN_LO 0x8000u     // key >> 16 of +0.0f
#define NF_WIN_N 16384        // keys 0x8000 .. 0xBFFF: 0 <= x < 2 (standard deviations of a [0,1] image) live in LDS

struct NleState {
    // results: the host reads this head (NLE_HEAD_BYTES) in one copy
    double ths[NF_MAXQ];             // np.percentile(lap, quants, 'linear')
    double sel[4];                   // i*, ths[i*], quants[i*], score[i*]      (YOND_SIDD.py:45-47)
    double mom[10];                  // {n, Sm, Sv, Smm, Smv} x {all, 1e-4 < mean < 0.8} over lap < sel[1]
    int npeaks[NF_MAXQ];             // occupied mean bins among lap <= ths[i]   (YOND_SIDD.py:37-43)
    float vals[NF_MAXT];             // the order statistics behind ths
    unsigned int frame_max_key;      // f2key of the frame maximum (0 = not collected)
    int nslots, nt, nq;
    // select state
    unsigned int ticket[4];
    long long tgt_rank[NF_MAXT];     // remaining rank inside the target's level-1 bin
    unsigned int tgt_prefix[NF_MAXT];
    int tgt_slot[NF_MAXT];
    unsigned int slot_prefix[NF_MAXT];   // sorted distinct level-1 bins that hold a rank
    unsigned int slot_off[NF_MAXT], slot_cnt[NF_MAXT], slot_fill[NF_MAXT];   // candidate ranges of sweep 2
    unsigned int maxinv[NF_BINS];    // per mean bin: ~key of the SMALLEST lap seen (0 = bin empty): the bin is occupied
                                     // among lap <= T  <=>  its smallest lap <= T
    unsigned int hist1[NF_L1];
    unsigned long long dbg[64];      // cycle stamps of diagnostic builds (-DBF_STAMPS)
};
#define NLE_HEAD_DOUBLES (NF_MAXQ + 4 + 10)

struct NfArgs {
    long long ranks[NF_MAXT];
    double lerp_t[NF_MAXQ];
    double quants[NF_MAXQ];
    int nt, nq;
};

// ------------------------------------------------------------------------------------------------------------
// level-1 resolve, run by ONE workgroup (the last to arrive) after every level-1 count has been added to st->hist1
// ------------------------------------------------------------------------------------------------------------
__device__ inline void nf_resolve1(NleState* st, const NfArgs& a, unsigned int* s_scratch /* >= 64 * 68 + 1024 words of LDS */) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x, nw = nthr >> 6;
    const int nt = a.nt;
    unsigned long long* s_wtot = (unsigned long long*)s_scratch;                 // [16]
    unsigned long long* s_before = s_wtot + 16;                                 // [64]
    int* s_owner = (int*)(s_before + NF_MAXT);                                  // [64]
    unsigned int* s_newp = (unsigned int*)(s_owner + NF_MAXT);                  // [64]
    int* s_first = (int*)(s_newp + NF_MAXT);                                    // [64]
    unsigned int* s_slotp = (unsigned int*)(s_first + NF_MAXT);                 // [64]
    int* s_tslot = (int*)(s_slotp + NF_MAXT);                                   // [64]
    unsigned int* s_cnt = (unsigned int*)(s_tslot + NF_MAXT);                   // [64]
    int* s_ns = (int*)(s_cnt + NF_MAXT);                                        // [1]
    const int per = NF_L1 / nthr;                                               // bins per thread (multiple of 4)
    const uint4* h4 = (const uint4*)(st->hist1 + (size_t)tid * per);
    unsigned long long mine = 0;
    for (int i = 0; i < per / 4; ++i) {
        const uint4 q = h4[i];
        mine += (unsigned long long)q.x + q.y + q.z + q.w;
    }
    unsigned long long incl = wave_incl_scan_u64(mine, lane);
    if (lane == 63) s_wtot[wave] = incl;
    __syncthreads();
    for (int w = 0; w < wave; ++w) incl += s_wtot[w];
    const unsigned long long excl = incl - mine;
    for (int t = 0; t < nt; ++t) {
        const unsigned long long rank = (unsigned long long)a.ranks[t];
        if (rank >= excl && rank < incl) { s_owner[t] = tid; s_before[t] = excl; }
    }
    __syncthreads();
    // one wave per target: the owner's `per` bins across the lanes (per <= 256: up to 4 per lane)
    for (int t = wave; t < nt; t += nw) {
        const unsigned long long rank = (unsigned long long)a.ranks[t];
        const unsigned int* hb = st->hist1 + (size_t)s_owner[t] * per;
        const int pl = (per + 63) / 64;                                         // bins per lane
        unsigned long long c[4] = {0, 0, 0, 0}, tot = 0;
        for (int j = 0; j < pl; ++j) {
            const int b = lane * pl + j;
            c[j] = b < per ? hb[b] : 0u;
            tot += c[j];
        }
        const unsigned long long bi = wave_incl_scan_u64(tot, lane) + s_before[t];
        const unsigned long long m = __ballot(bi > rank);
        const int src = m ? (__ffsll((long long)m) - 1) : 63;
        unsigned long long cum = bi - tot;                                      // elements before this lane's bins
        int d = 0;
        for (int j = 0; j < pl - 1; ++j) {
            if (rank >= cum + c[j] && d == j) { cum += c[j]; d = j + 1; }
        }
        d = __shfl(d, src);
        cum = __shfl(cum, src);
        if (lane == 0) {
            const unsigned int np = (unsigned int)(s_owner[t] * per + src * pl + d);
            st->tgt_rank[t] = (long long)(rank - cum);
            st->tgt_prefix[t] = np;
            s_newp[t] = np;
        }
    }
    __syncthreads();
    // slots = sorted distinct prefixes
    if (tid == 0) *s_ns = 0;
    if (tid < nt) {
        bool first = true;
        for (int u = 0; u < tid; ++u) first = first && (s_newp[u] != s_newp[tid]);
        s_first[tid] = first ? 1 : 0;
    }
    __syncthreads();
    if (tid < nt) {
        const unsigned int p = s_newp[tid];
        int pos = 0;
        for (int u = 0; u < nt; ++u) pos += (s_first[u] && s_newp[u] < p) ? 1 : 0;
        s_tslot[tid] = pos;
        if (s_first[tid]) { s_slotp[pos] = p; s_cnt[pos] = st->hist1[p]; atomicAdd(s_ns, 1); }
    }
    __syncthreads();
    const int ns = *s_ns;
    if (tid < nt) st->tgt_slot[tid] = s_tslot[tid];
    if (tid < ns) {
        unsigned int off = 0;
        for (int u = 0; u < tid; ++u) off += (s_cnt[u] + 7u) & ~7u;             // 16-byte aligned ranges of 2-byte entries
        st->slot_prefix[tid] = s_slotp[tid];
        st->slot_cnt[tid] = s_cnt[tid];
        st->slot_off[tid] = off;
    }
    if (tid == 0) { st->nslots = ns; st->nt = nt; st->nq = a.nq; }
}

// every workgroup calls this after its last global atomic of the sweep; true in exactly one workgroup: the last
__device__ inline bool nf_arrive_last(unsigned int* ticket, unsigned int nblocks) {
    __shared__ int s_last;
    __threadfence();                                       // this workgroup's atomics are performed before the ticket
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(ticket, 1u) == nblocks - 1) ? 1 : 0;
    __syncthreads();
    if (s_last) __threadfence();                           // acquire: nothing of the other workgroups is read from a stale L1
    return s_last != 0;
}


// host helpers (nle_fast.hip)
int nf_make_args(size_t n, const double* q_host, int nq, NfArgs* a);
static inline size_t nf_state_bytes() { return (sizeof(NleState) + 255) & ~(size_t)255; }
