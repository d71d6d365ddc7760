// covfun.h -- device-side covariance functions of GeoBO's kernel library (reference: geobo/kernels.py).
//
// One CovParams struct is prepared on the host (make_cov) and passed by value to every kernel that
// evaluates covariances: the materialising k_block / k_eval kernels and the generator stage of the
// fused fp64-MFMA product (gemm_f64.hip).  All arithmetic is IEEE fp64.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

enum CovId {
  COV_D2 = 0, COV_EXP = 1, COV_EXP_X = 2, COV_MATERN32 = 3, COV_MATERN32_X = 4, COV_SPARSE = 5, COV_SPARSE_X = 6
};

struct CovParams {
  int id;
  double scale;   // w * amp, applied last (kernels.py:183-195: w * k(...), inversion.py:92: gp_amp * create_cov)
  double l1, l2;  // lengths as the reference sees them (after the sparse equal-length offset)
  double c[8];    // family-specific constants, see make_cov
};

// ---- host: precompute constants with the reference's own operation order where it matters -------------
static inline CovParams make_cov(int id, double l1, double l2, double w, double amp) {
  CovParams p;
  p.id = id; p.scale = w * amp; p.l1 = l1; p.l2 = l2;
  for (int i = 0; i < 8; ++i) p.c[i] = 0.0;
  const double PI = 3.141592653589793;
  switch (id) {
    case COV_EXP:  // exp(-0.5*D2/gamma**2)                                  kernels.py:88
      p.c[0] = 1.0 / (l1 * l1);
      break;
    case COV_EXP_X:  // sqrt(2 l1 l2/(l1^2+l2^2)) * exp(-D2/(l1^2+l2^2))     kernels.py:99
      p.c[0] = 1.0 / (l1 * l1 + l2 * l2);
      p.c[1] = sqrt(2. * l1 * l2 / (l1 * l1 + l2 * l2));
      break;
    case COV_MATERN32:  // nu = sqrt(3)*sqrt(D2)/gamma; (1+nu) exp(-nu)      kernels.py:145-146
      p.c[0] = 1.0 / l1;
      p.c[1] = sqrt(3.0);
      break;
    case COV_MATERN32_X:  // norm*(l1 exp(-sqrt(3 D2)/l1) - l2 exp(-sqrt(3 D2)/l2))   kernels.py:153-156
      p.c[0] = 1.0 / l1;
      p.c[1] = 1.0 / l2;
      p.c[2] = 2 * sqrt(l1 * l2) / (l1 * l1 - l2 * l2);  // inf at l1 == l2, like the reference
      break;
    case COV_SPARSE:  // kernels.py:109-113
      p.c[0] = 1.0 / l1;
      p.c[1] = 2 * PI;
      p.c[2] = 1 / (2. * PI);
      break;
    case COV_SPARSE_X: {  // kernels.py:121-137
      if (l1 == l2) { l2 += 1e-3 * l2; p.l2 = l2; }
      const double lmean = (l1 + l2) / 2.0;  // np.mean([l1,l2])
      const double lmin = l1 < l2 ? l1 : l2, lmax = l1 < l2 ? l2 : l1;
      p.c[0] = fabs(l2 - l1) / 2.;                    // branch A upper limit
      p.c[1] = (l1 + l2) / 2.;                        // branch B upper limit
      p.c[2] = 2. / (3 * sqrt(l1 * l2));              // common prefactor
      p.c[3] = lmin;
      p.c[4] = 1 / PI * (lmax * lmax * lmax) / (lmax * lmax - lmin * lmin);
      p.c[5] = PI * lmin / lmax;
      p.c[6] = lmax;
      p.c[7] = lmean;
      break;
    }
    default: break;
  }
  return p;
}

#ifdef __HIPCC__
// exp(x) for x <= ~0 (covariances only ever need decaying exponentials): Cody-Waite reduction by ln2,
// degree-13 Taylor/Horner on |r| <= ln2/2 (truncation 4e-18), v_ldexp_f64.  < 1.5 ulp, no special cases
// beyond a clamp that maps very negative arguments to 0 through ldexp's gradual underflow.
__device__ __forceinline__ double exp_decay(double x) {
  x = fmax(x, -800.0);
  const double n = __builtin_rint(x * 1.4426950408889634);
  double r = __builtin_fma(-n, 6.93147180369123816490e-01, x);
  r = __builtin_fma(-n, 1.90821492927058770002e-10, r);
  double p = 1.6059043836821613e-10;               // 1/13!
  p = __builtin_fma(p, r, 2.08767569878681e-09);   // 1/12!
  p = __builtin_fma(p, r, 2.505210838544172e-08);  // 1/11!
  p = __builtin_fma(p, r, 2.755731922398589e-07);  // 1/10!
  p = __builtin_fma(p, r, 2.7557319223985893e-06); // 1/9!
  p = __builtin_fma(p, r, 2.48015873015873e-05);   // 1/8!
  p = __builtin_fma(p, r, 1.984126984126984e-04);  // 1/7!
  p = __builtin_fma(p, r, 1.388888888888889e-03);  // 1/6!
  p = __builtin_fma(p, r, 8.333333333333333e-03);  // 1/5!
  p = __builtin_fma(p, r, 4.1666666666666664e-02); // 1/4!
  p = __builtin_fma(p, r, 1.6666666666666666e-01); // 1/3!
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return ldexp(p, (int)n);
}

// squared distance with the reference's rounding: 0 + dx^2 + dy^2 + dz^2, no FMA contraction (kernels.py:46,51-58)
__device__ __forceinline__ double sqdist3(double px, double py, double pz, double qx, double qy, double qz) {
#pragma clang fp contract(off)
  const double dx = qx - px, dy = qy - py, dz = qz - pz;
  return (dx * dx + dy * dy) + dz * dz;
}

// l1*e1 - l2*e2 with BOTH products rounded (no FMA contraction): at l1 == l2 the reference gets exactly 0 here and
// inf*0 = NaN everywhere (kernels.py:153-156); a contracted fma would leave the rounding residual and give +-inf.
__device__ __forceinline__ double matern_x_diff(double l1, double e1, double l2, double e2) {
#pragma clang fp contract(off)
  const double a = l1 * e1, b = l2 * e2;
  return a - b;
}

// k(d2) for family ID (compile-time) -- without the w*amp scale
template <int ID>
__device__ __forceinline__ double cov_eval(const CovParams& p, double d2) {
  if constexpr (ID == COV_D2) {
    return d2;
  } else if constexpr (ID == COV_EXP) {
    return exp_decay((-0.5 * d2) * p.c[0]);
  } else if constexpr (ID == COV_EXP_X) {
    return p.c[1] * exp_decay(-d2 * p.c[0]);
  } else if constexpr (ID == COV_MATERN32) {
    const double nu = (p.c[1] * sqrt(d2)) * p.c[0];
    return (1 + nu) * exp_decay(-nu);
  } else if constexpr (ID == COV_MATERN32_X) {
    const double r = sqrt(3 * d2);
    return p.c[2] * matern_x_diff(p.l1, exp_decay(-r * p.c[0]), p.l2, exp_decay(-r * p.c[1]));
  } else if constexpr (ID == COV_SPARSE) {
    const double d = sqrt(d2);
    double res = 0.0;
    if (d < p.l1) {
      const double t = (p.c[1] * d) / p.l1;  // 2*pi*d/gamma (true divisions: sparse is not the hot family)
      res = (2 + cos(t)) / 3. * (1 - d / p.l1) + p.c[2] * sin(t);
      if (res < 0.) res = 0.;
    }
    return res;
  } else {  // COV_SPARSE_X
    const double d = sqrt(d2);
    double res = 0.0;
    const double l1 = p.l1, l2 = p.l2;
    if (d >= p.c[0] && d <= p.c[1]) {  // branch B (assigned last in the reference, wins at equality)
      const double den = 2 * 3.141592653589793 * (l1 * l1 - l2 * l2);
      res = p.c[2] * (p.c[7] - d + (l1 * l1 * l1) * sin(3.141592653589793 * (l2 - 2. * d) / l1) / den -
                      (l2 * l2 * l2) * sin(3.141592653589793 * (l1 - 2. * d) / l2) / den);
    } else if (d <= p.c[0]) {  // branch A: cos sits inside the sin argument, as coded in kernels.py:133
      res = p.c[2] * (p.c[3] + p.c[4] * sin(p.c[5] * cos(2 * 3.141592653589793 * d / p.c[6])));
    }
    if (res < 0.) res = 0.;
    return res;
  }
}

// runtime-dispatch wrapper for the non-fused kernels
#define COV_DISPATCH(id, F)                     \
  switch (id) {                                 \
    case COV_D2: F(COV_D2); break;              \
    case COV_EXP: F(COV_EXP); break;            \
    case COV_EXP_X: F(COV_EXP_X); break;        \
    case COV_MATERN32: F(COV_MATERN32); break;  \
    case COV_MATERN32_X: F(COV_MATERN32_X); break; \
    case COV_SPARSE: F(COV_SPARSE); break;      \
    case COV_SPARSE_X: F(COV_SPARSE_X); break;  \
    default: return -1;                         \
  }
#endif  // __HIPCC__
