// sweep2_dev.h -- device helpers shared by the second-generation sweep kernels (sweep2.hip: large batches and the
// gallery's rank-1 refresh; small2.hip: DIRECT's batches): buffer-resource loads, the table-based exp, the k*
// value of one exponent, the acquisition epilogue and the per-tile candidate staging.
#pragma once
#include "ibo_common.h"

typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t s2_rsrc(const void *p, size_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (int)(bytes > 0x7fffffffu ? 0x7fffffffu : bytes), 0x00020000);
}

__device__ __forceinline__ double s2_ld_f64(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    v2u_t v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return __hiloint2double((int)v.y, (int)v.x);
}

__device__ __forceinline__ double s2_lo(const v4u_t &v) { return __hiloint2double((int)v.y, (int)v.x); }
__device__ __forceinline__ double s2_hi(const v4u_t &v) { return __hiloint2double((int)v.w, (int)v.z); }

// exp(y) for the k* generation in 11 VALU instructions (exp_fast in ibo_common.h takes 17, the library ~30): every
// VALU instruction here is paid in MFMA issue slots.  A 2048-entry table of 2^(j/2048) in LDS (reads are not VALU
// instructions) shortens the polynomial to degree 3:
//   t = y 2048/ln2 + 1.5 2^52 puts n = rint(y 2048/ln2) in the low mantissa bits of t; r = y - n ln2/2048, |r| <=
//   1.7e-4 (r^4/24 < 4e-17); j = n mod 2048 indexes the table and v_ldexp_f64 applies n div 2048, flushing to zero
//   what underflows.  One FMA forms r: the rounding of ln2/2048 costs |n| 2.7e-20 relative -- 2.4e-15 at y = -30
//   where k* is already 1e-13, 5.6e-14 at the underflow threshold.  Relative error < 5e-16 for |y| < 10.
//   Needs |y| < 7e5 (n must fit 32 bits): the kernel's prologue bounds the scaled candidates so that it holds.
__device__ __forceinline__ double s2_exp(double y, const double *tab)
{
    const double magic = 6755399441055744.0;               // 1.5 * 2^52
    const double t = fma(y, 2954.6394437405970584, magic);              // 2048 / ln 2
    const double n = t - magic;
    const double r = fma(n, -3.384507717577858e-04, y);                 // ln2 / 2048
    const int ti = __double2loint(t);
    const double T = *(const double *)((const char *)tab + ((ti << 3) & (2047 << 3)));
    double p = fma(r, 1.0 / 6.0, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return __builtin_amdgcn_ldexp(T * p, ti >> 11);
}

// acquisition epilogue of one candidate; coordinates are read from global memory where needed (prior,
// exclusion balls) so that no per-lane coordinate array exists (dynamic indexing would put it in scratch)
// SYS: the candidate's coordinates lie in host memory that a resident kernel sees change (server.hip): system-scope loads
template <bool SYS = false>
__device__ __forceinline__ double s2_finish(const SweepArgs &a, const double *xp, double q, double muY, double mu1,
                                            int64_t li, bool valid, bool &excluded)
{
    const int D = a.kp.D;
    auto X = [&](int j) { return SYS ? __hip_atomic_load(xp + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : xp[j]; };
    double m = 0.0;
    if (a.prior.nb > 0) {
        for (int i = 0; i < a.prior.nb; i++) {
            double d = 0.0;
            for (int j = 0; j < D; j++) {
                double t = (X(j) - a.prior.lowerb[j]) / a.prior.width[j] - a.prior.means[(size_t)i * D + j];
                d += t * t;
            }
            m += a.prior.beta[i] * exp(-a.prior.theta * d);
        }
    }
    const double mu = (a.prior.nb > 0) ? (m + muY - m * mu1) : muY;
    double s2 = 1.0 + a.noise - q;
    if (s2 < a.clamp_lo) s2 = a.clamp_lo;
    else if (s2 > 10.0) s2 = 10.0;
    const double val = (a.acq == 3) ? mu : acq_value_dev(a.acq, a.erf_mode, mu, sqrt(s2), a.ymax, a.parm);
    excluded = false;
    for (int e = 0; e < a.n_excl; e++) {
        double d2 = 0.0;
        for (int j = 0; j < D; j++) { double t = X(j) - a.excl[(size_t)e * D + j]; d2 += t * t; }
        if (!(sqrt(d2) > a.excl_radius)) excluded = true;
    }
    if (valid) {
        if (SYS) {                                      // (a resident kernel has no launch boundary to flush its results to host memory)
            if (a.out_mu) __hip_atomic_store(a.out_mu + li, mu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (a.out_s2) __hip_atomic_store(a.out_s2 + li, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (a.out_acq) __hip_atomic_store(a.out_acq + li, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
            if (a.out_mu) a.out_mu[li] = mu;
            if (a.out_s2) a.out_s2[li] = s2;
            if (a.out_acq) a.out_acq[li] = val;
        }
    }
    return val;
}


// k* from the exponent y = a_k + b_c + x~.c~ (b_c carries log sf2 for the squared exponential)
template <int FAM>
__device__ __forceinline__ double s2_kstar(double y, double sf2, const double *tab)
{
    if (FAM == FAM_SE) return s2_exp(y, tab);
    const double z = fmax(-2.0 * y, 0.0);                                           // z = |x~ - c~|^2 = -2y
    const double rr = sqrt_fast((FAM == FAM_M3 ? 3.0 : 5.0) * z);
    const double poly = FAM == FAM_M3 ? 1.0 + rr : fma(rr, fma(rr, 1.0 / 3.0, 1.0), 1.0);
    return sf2 * poly * s2_exp(-rr, tab);
}

// TCAND candidates of a tile -> lds_c[cand][KA (+1: odd row stride, conflict-free fragment reads)] = [c~ (D) | 1 | b_c | 0..]  (c~ = c sqrt(w)).  A candidate more than 775
// length scales from the origin (hence > 450 from every observation: |x~| <= 316 where the dot form is in use) has k* = 0
// exactly; it is pulled in to that radius, where k* is still 0, so that the exponent stays within what s2_exp's integer
// arithmetic covers (|y| < 7e5).  Called by the whole workgroup; ends with a barrier.
template <int FAM, int TCAND, int KA, int NT, bool SYS = false>
__device__ __forceinline__ void s2_stage_candidates(const SweepArgs &a, int64_t tile0, double *lds_c, const double *cand = nullptr, int64_t Mo = -1)
{
    const int tid = threadIdx.x, D = a.kp.D;
    const int64_t Mtot = Mo >= 0 ? Mo : a.M;          // (a resident kernel's batches differ in size: the caller says)
    if (!cand) cand = a.cand;
    for (int e = tid; e < TCAND * KA; e += NT) {
        const int c = e / KA, col = e - c * KA;
        int64_t gi = tile0 + c;
        if (gi > Mtot - 1) gi = Mtot - 1;
        lds_c[c * (KA + 1) + col] = (col < D) ? (SYS ? __hip_atomic_load(cand + gi * D + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : cand[gi * D + col]) * a.kp.sw[col]
                                              : (col == D ? 1.0 : 0.0);
    }
    __syncthreads();
    if (tid < TCAND) {
        double n2 = 0.0;
        for (int d = 0; d < D; d++) { const double v = lds_c[tid * (KA + 1) + d]; n2 = fma(v, v, n2); }
        if (n2 > 6e5) {
            const double sc = sqrt(6e5 / n2);
            for (int d = 0; d < D; d++) lds_c[tid * (KA + 1) + d] *= sc;
            n2 = 6e5;
        }
        lds_c[tid * (KA + 1) + D + 1] = fma(-0.5, n2, FAM == FAM_SE ? a.log_sf2 : 0.0);
    }
    __syncthreads();
}
