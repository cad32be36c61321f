// Per-row arithmetic of the deprojection pre-pass, shared by every kernel that streams the visibility table
// (deproject_kernel of bin_gram.hip; uv_hist_kernel and deproject_scatter_kernel of bin_prepass.hip) so that a row's
// baseline, bucket and scaled visibility are the same bits wherever they are formed.
//
// geometry.py:69-79 (inverse phase shift, NumPy's Smith complex division), :111-131 (deproject),
// statistical_models.py:166 (hypot).  Every product / sum rounds separately, as the NumPy expressions do.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"

struct VisRow {
    double u, v, Vre, Vim, w;
};

// one row of the table (fp32 tables are widened here; everything after it is the fp64 path)
__device__ __forceinline__ VisRow fh_load_row(const BinParams &p, int64_t g) {
    VisRow r;
    if (p.u32) {
        r.u = (double)p.u32[g];
        r.v = (double)p.v32[g];
        r.Vre = (double)p.Vre32[g];
        r.Vim = p.Vim32 ? (double)p.Vim32[g] : 0.0;
        r.w = (double)p.w32[p.w_scalar ? 0 : g];
    } else {
        // (streamed: non-temporal, so that the pass does not push the fit loops' working sets out of L2)
        r.u = __builtin_nontemporal_load(&p.u[g]);
        r.v = __builtin_nontemporal_load(&p.v[g]);
        r.Vre = __builtin_nontemporal_load(&p.Vre[g]);
        r.Vim = p.Vim ? __builtin_nontemporal_load(&p.Vim[g]) : 0.0;
        r.w = p.w_scalar ? p.w[0] : __builtin_nontemporal_load(&p.w[g]);
    }
    return r;
}
__device__ __forceinline__ void fh_load_uv(const BinParams &p, int64_t g, double &u, double &v) {
    if (p.u32) {
        u = (double)p.u32[g];
        v = (double)p.v32[g];
    } else {
        u = __builtin_nontemporal_load(&p.u[g]);
        v = __builtin_nontemporal_load(&p.v[g]);
    }
}

// deprojected baseline length q = hypot(u', v') (geometry.py:111-131, statistical_models.py:166)
__device__ __forceinline__ double fh_deproject_q(const BinParams &p, double u, double v) {
#pragma clang fp contract(off)
    double up = u * p.cos_t - v * p.sin_t;
    const double vp = u * p.sin_t + v * p.cos_t;
    up = up * p.cos_i;
    return hypot(up, vp);
}

// Re of V / exp(i phi), phi = u dRA + v dDec (geometry.py:69-79; NumPy's complex division is Smith's algorithm)
__device__ __forceinline__ double fh_phase_centre_re(const BinParams &p, double u, double v, double Vre, double Vim) {
#pragma clang fp contract(off)
    const double phi = u * p.dRA + v * p.dDec;
    double sn, cs;
    sincos(phi, &sn, &cs);
    double re;
    if (fabs(cs) >= fabs(sn)) {
        const double rat = sn / cs, scl = 1.0 / (cs + sn * rat);
        re = (Vre + Vim * rat) * scl;
    } else {
        const double rat = cs / sn, scl = 1.0 / (sn + cs * rat);
        re = (Vre * rat + Vim) * scl;
    }
    return re;
}

// vertical uv-distance squared of the 3-D deprojection (debris model), wp = u' sin(inc) (geometry.py:128)
__device__ __forceinline__ double fh_deproject_kz2(const BinParams &p, double u, double v) {
#pragma clang fp contract(off)
    const double wz = (u * p.cos_t - v * p.sin_t) * p.sin_i;
    return wz * wz;
}

// bucket of s = q / Qmax (j0_buckets.h): the last bucket takes whatever lies beyond
__device__ __forceinline__ int fh_bucket_of(double s, double inv_delta, int nb) {
    int b = (int)(s * inv_delta);  // s >= 0
    return b < nb - 1 ? b : nb - 1;
}
// offset inside the bucket, in [-1, 1]: the argument of the bucket's Taylor table (fh_k1_bucket_centre)
__device__ __forceinline__ double fh_bucket_tau(double s, int b, double delta, double inv_half) {
#pragma clang fp contract(off)
    return (s - ((double)b + 0.5) * delta) * inv_half;
}

// ---- the same quantities with fewer instructions (bin_prepass.hip) ---------------------------------------------------------
// The functions above follow NumPy operation by operation (Smith's complex division, hypot, the library's sincos and log) and
// cost ~480 fp64 vector instructions per 64 rows; on MI355X a 64-wide fp64 instruction takes four cycles and the pre-pass of
// the moments path was half bound by them.  These differ from the above by an ulp or two per row (the rows path keeps the
// NumPy-faithful forms and is the cross-check: tests/test_gpu_configs.py::test_moment_path_equals_row_path).

// q = sqrt(u'^2 + v'^2): baselines are 1e2 .. 1e9 wavelengths, nothing to rescale
__device__ __forceinline__ double fh_deproject_q_fast(const BinParams &p, double u, double v) {
    double up, vp;
    {
#pragma clang fp contract(off)
        up = (u * p.cos_t - v * p.sin_t) * p.cos_i;  // (u', v' as NumPy rounds them)
        vp = u * p.sin_t + v * p.cos_t;
    }
    return sqrt(fma(up, up, vp * vp));
}

// sin and cos of a phase |x| < 1e5: two-term Cody-Waite reduction by pi/2 (exact products through fma), then the fdlibm
// kernels on [-pi/4, pi/4] (|error| <= 1.1e-16, checked against NumPy on 2e6 points)
__device__ __forceinline__ void fh_sincos_small(double x, double &sn, double &cs) {
    const double k = rint(x * 0.63661977236758134308);
    double r = fma(-k, 1.5707963267948966, x);
    r = fma(-k, 6.123233995736766e-17, r);
    const double z = r * r;
    double ps = 1.58969099521155010221e-10;
    ps = fma(ps, z, -2.50507602534068634195e-08);
    ps = fma(ps, z, 2.75573137070700676789e-06);
    ps = fma(ps, z, -1.98412698298579493134e-04);
    ps = fma(ps, z, 8.33333333332248946124e-03);
    ps = fma(ps, z, -1.66666666666666324348e-01);
    const double s = fma(r * z, ps, r);
    double pc = -1.13596475577881948265e-11;
    pc = fma(pc, z, 2.08757232129817482790e-09);
    pc = fma(pc, z, -2.75573143513906633035e-07);
    pc = fma(pc, z, 2.48015872894767294178e-05);
    pc = fma(pc, z, -1.38888888888741095749e-03);
    pc = fma(pc, z, 4.16666666666666019037e-02);
    const double c = fma(z * z, pc, fma(-0.5, z, 1.0));
    const int q = (int)k;
    const double a = (q & 1) ? c : s, b = (q & 1) ? s : c;  // sin x = [s, c, -s, -c][q mod 4], cos x = [c, -s, -c, s][q mod 4]
    sn = (q & 2) ? -a : a;
    cs = ((q + 1) & 2) ? -b : b;
}

// Re of V exp(-i phi) = Re V cos(phi) + Im V sin(phi): what the complex division of geometry.py:76-79 amounts to
__device__ __forceinline__ double fh_phase_centre_re_fast(const BinParams &p, double u, double v, double Vre, double Vim) {
    double phi;
    {
#pragma clang fp contract(off)
        phi = u * p.dRA + v * p.dDec;  // (the phase as NumPy rounds it)
    }
    double sn, cs;
    fh_sincos_small(phi, sn, cs);  // (the caller has bounded |phi| by 1e5: bin_visibilities_v4)
    return fma(Vre, cs, Vim * sn);
}
