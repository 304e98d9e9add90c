// flux_math.hpp -- device arithmetic of the compressible-Euler face flux (gfx950).
//
// Two formulations of the same flux:
//   * `*_ref`  : the reference's operation sequence (examples/compressible_euler/kernels.cu:24-133,
//                220-279; examples/subgrid/kernels.inl:132-332), kept for the reference-dataflow
//                ("compat") kernels. The eigenvector matrix is applied in its sparse form; dropping
//                exact zeros and ones leaves every rounding step of the dense loops unchanged.
//   * `Prim` + `kepes_prim` : per-ELEMENT quantities (velocity, p, beta, logs, entropy
//                variable) are computed once per element and stage, so a face costs no log and a
//                third of the divisions. Used by the fused tile kernels.
#pragma once

#include <cstdlib>
#include <hip/hip_runtime.h>

#include "log_table.hpp"

namespace t8gpu_hip {

#define T8_DEV __device__ __forceinline__

template <class T>
struct rk3c;
template <>
struct rk3c<float> {
  static constexpr float c21 = 0.75f, c22 = 0.25f, c23 = 0.25f;
  static constexpr float c31 = 0.33333333333333f, c32 = 0.66666666666666f, c33 = 0.66666666666666f;
};
template <>
struct rk3c<double> {
  static constexpr double c21 = 0.75, c22 = 0.25, c23 = 0.25;
  static constexpr double c31 = 0.33333333333333, c32 = 0.66666666666666, c33 = 0.66666666666666;
};

T8_DEV float  t8_log(float x) { return logf(x); }
T8_DEV double t8_log(double x) { return log(x); }
T8_DEV float  t8_sqrt(float x) { return sqrtf(x); }
T8_DEV double t8_sqrt(double x) { return sqrt(x); }
T8_DEV float  t8_abs(float x) { return fabsf(x); }
T8_DEV double t8_abs(double x) { return fabs(x); }
T8_DEV float  t8_cbrt(float x) { return cbrtf(x); }
T8_DEV double t8_cbrt(double x) { return cbrt(x); }
T8_DEV float  t8_min(float a, float b) { return fminf(a, b); }
T8_DEV double t8_min(double a, double b) { return fmin(a, b); }
T8_DEV float  t8_max(float a, float b) { return fmaxf(a, b); }
T8_DEV double t8_max(double a, double b) { return fmax(a, b); }

// logarithmic mean, kernels.cu:24-36
template <class T>
T8_DEV T ln_mean_ref(T aL, T aR) {
  const T xi = aR / aL;
  const T u  = (xi * (xi - T(2.0)) + T(1.0)) / (xi * (xi + T(2.0)) + T(1.0));
  if (u < T(1.0e-4)) return (aL + aR) * T(52.50) / (T(105.0) + u * (T(35.0) + u * (T(21.0) + u * T(15.0))));
  return (aR - aL) / t8_log(xi);
}

// Total KEPES flux in the face frame: EC flux (kernels.cu:38-93) minus half the matrix
// dissipation R |L| R^T [v] (kernels.cu:95-133, 224-279). speed = |uHat| + aHat (kernels.cu:222).
template <class T>
T8_DEV void kepes_ref(const T uL[5], const T uR[5], T F[5], T& speed) {
  const T one = T(1), half = T(0.5);
  const T kappa = T(1.4);
  const T km1   = kappa - one;
  const T skm1  = one / km1;

  const T irL = one / uL[0];
  const T vxL = irL * uL[1], vyL = irL * uL[2], vzL = irL * uL[3];
  const T irR = one / uR[0];
  const T vxR = irR * uR[1], vyR = irR * uR[2], vzR = irR * uR[3];
  const T qL  = half * (vxL * vxL + vyL * vyL + vzL * vzL);
  const T qR  = half * (vxR * vxR + vyR * vyR + vzR * vzR);
  const T pL  = km1 * (uL[4] - uL[0] * qL);
  const T pR  = km1 * (uR[4] - uR[0] * qR);
  const T bL  = half * uL[0] / pL;
  const T bR  = half * uR[0] / pR;

  const T rho_mean  = half * (uL[0] + uR[0]);
  const T rho       = ln_mean_ref<T>(uL[0], uR[0]);
  const T beta_mean = half * (bL + bR);
  const T beta_hat  = ln_mean_ref<T>(bL, bR);

  const T u  = half * (vxL + vxR);
  const T v  = half * (vyL + vyR);
  const T w  = half * (vzL + vzR);
  const T a  = t8_sqrt(kappa * half * (pL + pR) / rho);
  const T h  = kappa / (T(2.0f) * km1 * beta_hat) + half * (vxL * vxR + vyL * vyR + vzL * vzR);
  const T p1 = half * rho_mean / beta_mean;
  const T q2 = qL + qR;

  T Fs[5];
  Fs[0] = rho * u;
  Fs[1] = Fs[0] * u + p1;
  Fs[2] = Fs[0] * v;
  Fs[3] = Fs[0] * w;
  Fs[4] = Fs[0] * half * (skm1 / beta_hat - q2) + u * Fs[1] + v * Fs[2] + w * Fs[3];

  speed = t8_abs(u) + a;

  const T D0 = half * t8_abs(u - a) * rho / kappa;
  const T D1 = t8_abs(u) * (km1 / kappa) * rho;
  const T D2 = t8_abs(u) * p1;
  const T D4 = half * t8_abs(u + a) * rho / kappa;

  // entropy variables, kernels.cu:227-262 (pressure recomputed the way the kernel does)
  const T VL[3] = {uL[1] * irL, uL[2] * irL, uL[3] * irL};
  const T VR[3] = {uR[1] * irR, uR[2] * irR, uR[3] * irR};
  const T pL2 = km1 * (uL[4] - half * (uL[1] * VL[0] + uL[2] * VL[1] + uL[3] * VL[2]));
  const T pR2 = km1 * (uR[4] - half * (uR[1] * VR[0] + uR[2] * VR[1] + uR[3] * VR[2]));
  const T sL  = t8_log(pL2) - kappa * t8_log(uL[0]);
  const T sR  = t8_log(pR2) - kappa * t8_log(uR[0]);
  const T rpL = uL[0] / pL2, rpR = uR[0] / pR2;
  const T v0L = (kappa - sL) / (km1)-half * rpL * (VL[0] * VL[0] + VL[1] * VL[1] + VL[2] * VL[2]);
  const T v0R = (kappa - sR) / (km1)-half * rpR * (VR[0] * VR[0] + VR[1] * VR[1] + VR[2] * VR[2]);
  const T J0 = v0R - v0L;
  const T J1 = rpR * VR[0] - rpL * VL[0];
  const T J2 = rpR * VR[1] - rpL * VL[1];
  const T J3 = rpR * VR[2] - rpL * VL[2];
  const T J4 = (-rpR) - (-rpL);

  const T hm = h - u * a, hp = h + u * a, k2 = static_cast<T>(0.5) * (u * u + v * v + w * w);
  // d = D o (R^T J), column by column (zeros/ones of R dropped, order of the sums kept)
  const T d0 = D0 * (J0 + (u - a) * J1 + v * J2 + w * J3 + hm * J4);
  const T d1 = D1 * (J0 + u * J1 + v * J2 + w * J3 + k2 * J4);
  const T d2 = D2 * (J2 + v * J4);
  const T d3 = D2 * (J3 + w * J4);
  const T d4 = D4 * (J0 + (u + a) * J1 + v * J2 + w * J3 + hp * J4);
  // R d, row by row
  F[0] = Fs[0] - half * (d0 + d1 + d4);
  F[1] = Fs[1] - half * ((u - a) * d0 + u * d1 + (u + a) * d4);
  F[2] = Fs[2] - half * (v * d0 + v * d1 + d2 + v * d4);
  F[3] = Fs[3] - half * (w * d0 + w * d1 + d3 + w * d4);
  F[4] = Fs[4] - half * (hm * d0 + k2 * d1 + v * d2 + w * d3 + hp * d4);
}

// HLL (reference dead code), kernels.inl:263-332
template <class T>
T8_DEV void hll_ref(const T uL[5], const T uR[5], T F[5], T& speed) {
  const T zero = T(0), one = T(1), half = T(0.5);
  const T g = T(1.4);
  const T v1l = uL[1] / uL[0], v2l = uL[2] / uL[0], v3l = uL[3] / uL[0];
  const T pl  = (g - 1) * (uL[4] - half * uL[0] * (v1l * v1l + v2l * v2l + v3l * v3l));
  const T Hl  = (uL[4] + pl) / uL[0];
  const T cl  = t8_sqrt((g - 1) * (Hl - half * (v1l * v1l + v2l * v2l + v3l * v3l)));
  const T v1r = uR[1] / uR[0], v2r = uR[2] / uR[0], v3r = uR[3] / uR[0];
  const T pr  = (g - one) * (uR[4] - half * uR[0] * (v1r * v1r + v2r * v2r + v3r * v3r));
  const T Hr  = (uR[4] + pr) / uR[0];
  const T cr  = t8_sqrt((g - one) * (Hr - half * (v1r * v1r + v2r * v2r + v3r * v3r)));
  const T wl = t8_sqrt(uL[0]), wr = t8_sqrt(uR[0]);
  const T ws = wl + wr;
  const T v1 = (wl * v1l + wr * v1r) / ws;
  const T v2 = (wl * v2l + wr * v2r) / ws;
  const T v3 = (wl * v3l + wr * v3r) / ws;
  const T H  = (wl * Hl + wr * Hr) / ws;
  const T c  = t8_sqrt((g - one) * (H - half * (v1 * v1 + v2 * v2 + v3 * v3)));
  const T Sl = t8_min(v1 - c, v1l - cl);
  const T Sr = t8_max(v1 + c, v1r + cr);
  speed = t8_max(t8_abs(Sl), t8_abs(Sr));   // signal speed for the CFL step (not in the reference; oracle.hpp: hll_total_flux)
  const T Fl[5] = {uL[1], uL[1] * uL[1] / uL[0] + pl, uL[1] * v2l, uL[1] * v3l, uL[1] * Hl};
  const T Fr[5] = {uR[1], uR[1] * uR[1] / uR[0] + pr, uR[1] * v2r, uR[1] * v3r, uR[1] * Hr};
  const T sl = t8_min(Sl, zero);
  const T sr = t8_max(Sr, zero);
#pragma unroll
  for (int k = 0; k < 5; k++) F[k] = ((sr * Fl[k] - sl * Fr[k]) + sr * sl * (uR[k] - uL[k])) / (sr - sl);
}

// HLLC (not in the reference; see oracle.hpp): the HLL above with the contact wave restored, same wave speeds.
template <class T>
T8_DEV void hllc_ref(const T uL[5], const T uR[5], T F[5], T& speed) {
  const T zero = T(0), one = T(1), half = T(0.5);
  const T g = T(1.4);
  const T v1l = uL[1] / uL[0], v2l = uL[2] / uL[0], v3l = uL[3] / uL[0];
  const T kl  = half * (v1l * v1l + v2l * v2l + v3l * v3l);
  const T pl  = (g - one) * (uL[4] - uL[0] * kl);
  const T Hl  = (uL[4] + pl) / uL[0];
  const T cl  = t8_sqrt((g - one) * (Hl - kl));
  const T v1r = uR[1] / uR[0], v2r = uR[2] / uR[0], v3r = uR[3] / uR[0];
  const T kr  = half * (v1r * v1r + v2r * v2r + v3r * v3r);
  const T pr  = (g - one) * (uR[4] - uR[0] * kr);
  const T Hr  = (uR[4] + pr) / uR[0];
  const T cr  = t8_sqrt((g - one) * (Hr - kr));
  const T wl = t8_sqrt(uL[0]), wr = t8_sqrt(uR[0]);
  const T ws = wl + wr;
  const T v1 = (wl * v1l + wr * v1r) / ws, v2 = (wl * v2l + wr * v2r) / ws, v3 = (wl * v3l + wr * v3r) / ws;
  const T H  = (wl * Hl + wr * Hr) / ws;
  const T c  = t8_sqrt((g - one) * (H - half * (v1 * v1 + v2 * v2 + v3 * v3)));
  const T Sl = t8_min(v1 - c, v1l - cl), Sr = t8_max(v1 + c, v1r + cr);
  speed = t8_max(t8_abs(Sl), t8_abs(Sr));
  const T ml = uL[0] * (Sl - v1l), mr = uR[0] * (Sr - v1r);
  const T Ss = ((pr - pl) + (uL[1] * (Sl - v1l) - uR[1] * (Sr - v1r))) / (ml - mr);
  const bool left = Ss >= zero;
  const T    S  = left ? t8_min(Sl, zero) : t8_max(Sr, zero);
  const T    SK = left ? Sl : Sr, vn = left ? v1l : v1r, vt1 = left ? v2l : v2r, vt2 = left ? v3l : v3r;
  const T    p = left ? pl : pr, Hk = left ? Hl : Hr, m = left ? ml : mr;
  T          u[5];
#pragma unroll
  for (int k = 0; k < 5; k++) u[k] = left ? uL[k] : uR[k];
  const T fac = m / (SK - Ss);
  const T Us[5] = {fac, fac * Ss, fac * vt1, fac * vt2, fac * (u[4] / u[0] + (Ss - vn) * (Ss + p / m))};
  const T Fk[5] = {u[1], u[1] * vn + p, u[1] * vt1, u[1] * vt2, u[1] * Hk};
#pragma unroll
  for (int k = 0; k < 5; k++) F[k] = Fk[k] + S * (Us[k] - u[k]);
}

// frame of an axis-aligned face: exactly what face_basis() returns for n = +-e_axis
template <class T>
T8_DEV void axis_basis(int axis, bool positive, T n[3], T t1[3], T t2[3]) {
  const T s = positive ? T(1) : T(-1);
  n[0] = n[1] = n[2] = t1[0] = t1[1] = t1[2] = t2[0] = t2[1] = t2[2] = T(0);
  n[axis] = s;
  if (axis == 0) {
    t1[2] = -s;
    t2[1] = T(1);
  } else if (axis == 1) {
    t1[0] = s;
    t2[2] = T(-1);
  } else {
    t1[1] = s;
    t2[0] = T(-1);
  }
}

// face frame (n, t1, t2): kernels.cu:174-193 == kernels.inl:133-156
// (No contraction: tile_plan.cpp computes the same frame on the host for the geometry dictionary, with separately rounded
// products; a tile with a dictionary and one without must see the same bits.)
template <class T>
T8_DEV void face_basis(const T n[3], T t1[3], T t2[3]) {
#pragma clang fp contract(off)
  t1[0] = n[1];
  t1[1] = n[2];
  t1[2] = -n[0];
  const T dp = n[0] * t1[0] + n[1] * t1[1] + n[2] * t1[2];
  t1[0] -= dp * n[0];
  t1[1] -= dp * n[1];
  t1[2] -= dp * n[2];
  const T nrm = t8_sqrt(t1[0] * t1[0] + t1[1] * t1[1] + t1[2] * t1[2]);
  t1[0] /= nrm;
  t1[1] /= nrm;
  t1[2] /= nrm;
  t2[0] = n[1] * t1[2] - n[2] * t1[1];
  t2[1] = n[2] * t1[0] - n[0] * t1[2];
  t2[2] = n[0] * t1[1] - n[1] * t1[0];
}

template <class T>
T8_DEV void to_face_frame(const T n[3], const T t1[3], const T t2[3], const T s[5], T r[5], bool mirror) {
  r[0]       = s[0];
  const T mn = s[1] * n[0] + s[2] * n[1] + s[3] * n[2];
  r[1]       = mirror ? -(mn) : mn;
  r[2]       = s[1] * t1[0] + s[2] * t1[1] + s[3] * t1[2];
  r[3]       = s[1] * t2[0] + s[2] * t2[1] + s[3] * t2[2];
  r[4]       = s[4];
}

// Face-frame flux of the pair (sL, sR) given in xyz: rotate both states, evaluate the flux.
// KIND: 0 KEPES, 1 HLL. mirror => right state is the wall reflection of sL
// (kernels.cu:371-375, kernels.inl:169-176). The caller rotates back with `from_face_frame`;
// plain kernels scale by the area BEFORE rotating back (kernels.cu:281-290), subgrid kernels
// AFTER (kernels.inl:395-401).
template <class T, int KIND>
T8_DEV void face_frame_flux_ref(const T n[3], const T t1[3], const T t2[3], const T sL[5], const T sR[5],
                                bool mirror, T Ff[5], T& speed) {
  T a[5], b[5];
  to_face_frame<T>(n, t1, t2, sL, a, false);
  to_face_frame<T>(n, t1, t2, mirror ? sL : sR, b, mirror);
  if (KIND == 1) {
    hll_ref<T>(a, b, Ff, speed);
  } else if (KIND == 2) {
    hllc_ref<T>(a, b, Ff, speed);
  } else {
    kepes_ref<T>(a, b, Ff, speed);
  }
}

template <class T>
T8_DEV void from_face_frame(const T n[3], const T t1[3], const T t2[3], const T Ff[5], T g[5]) {
  g[0] = Ff[0];
  g[1] = Ff[1] * n[0] + Ff[2] * t1[0] + Ff[3] * t2[0];
  g[2] = Ff[1] * n[1] + Ff[2] * t1[1] + Ff[3] * t2[1];
  g[3] = Ff[1] * n[2] + Ff[2] * t1[2] + Ff[3] * t2[2];
  g[4] = Ff[4];
}

// ------------------------------------------------------------------------------------------------
// Per-element formulation (fused tile kernels).
//
// Everything in the KEPES flux that depends on ONE cell only is evaluated once per element and
// stage: velocity, pressure, beta = rho/(2p), log(rho), log(2 beta) and the first entropy variable.
// A face then needs no logarithm (log(aR/aL) = log aR - log aL; the series branch of ln_mean takes
// over before that difference loses accuracy: |log| >= 0.02 outside it) and 7 divisions instead of
// ~20. Frame-invariant pieces (|v|^2, the scalar entropy variable) are not rotated at all.
// Same flux as kernels.cu:38-133,220-279 up to rounding (a few ulp; parity tolerance in tests/).
// ------------------------------------------------------------------------------------------------
// ROUNDING IS PART OF THE CONTRACT of this tier: the fused kernels promise results that do not depend on the tiling,
// the partition or the kernel variant, bit for bit. Left to the compiler, which product of `a*b - c*d` is fused into
// an FMA depends on the code around the expression (measured: one kernel that routed the face velocities through
// branches differed from the others by one ulp in a few hundred values). Every function below therefore switches
// contraction off and spells its FMAs out; kernels use rk_stage_update() for the RK stage for the same reason.
T8_DEV float  t8_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
T8_DEV double t8_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// one SSP-RK3 stage value (ssp_runge_kutta.inl:30-99) from the previous-step value, the stage's source value and
// sa = dt / volume * (sum of the element's face fluxes)
template <class T, int STAGE>
T8_DEV T rk_stage_update(T pv, T s0, T scale, T acc) {
#pragma clang fp contract(off)
  if (STAGE == 1) return t8_fma(scale, acc, s0);
  if (STAGE == 2) return t8_fma(rk3c<T>::c23 * scale, acc, t8_fma(rk3c<T>::c22, s0, rk3c<T>::c21 * pv));
  return t8_fma(rk3c<T>::c33 * scale, acc, t8_fma(rk3c<T>::c32, s0, rk3c<T>::c31 * pv));
}

// Division in the fast tier. The reference-dataflow kernels keep IEEE division (v_div_scale / v_div_fmas /
// v_div_fixup: 10 instructions in fp32, 13 in fp64); here operands are O(1) physical quantities, never
// denormal or huge, so a reciprocal plus Newton steps is enough: fp32 v_rcp_f32 (1 ulp) and one
// multiply, fp64 v_rcp_f64 + two Newton steps + one residual correction (< 1 ulp, checked in tests).
T8_DEV float t8_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
T8_DEV double t8_rcp(double x) {   // (a bare reciprocal has no residual step behind it: two Newton steps; one leaves > 2 ulp)
  double r = __builtin_amdgcn_rcp(x);
  r        = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  r        = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  return r;
}
// sqrt for the fast tier: fp64 v_rsq_f64 + Goldschmidt step + residual (x > 0, normal range)
T8_DEV float  t8_sqrt_fast(float x) { return __builtin_amdgcn_sqrtf(x); }
T8_DEV double t8_sqrt_fast(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double       g = x * y, h = 0.5 * y;
  double       r = __builtin_fma(-h, g, 0.5);
  g              = __builtin_fma(g, r, g);
  h              = __builtin_fma(h, r, h);
  const double d = __builtin_fma(-g, g, x);
  return __builtin_fma(d, h, g);
}
T8_DEV float  t8_div(float a, float b) { return a * t8_rcp(b); }
T8_DEV double t8_div(double a, double b) {
  // v_rcp_f64 (~27 bits) + ONE Newton step (~52 bits) is enough for the reciprocal here: the quotient gets its
  // last bits from the residual correction below (pinned at < 2 ulp in tests/test_gpu_fastmath.py)
  double r = __builtin_amdgcn_rcp(b);
  r        = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
  const double q = a * r;
  return __builtin_fma(__builtin_fma(-b, q, a), r, q);
}

// A reciprocal that serves SEVERAL quotients with the same denominator, and the quotient from it (fp64: residual step as
// in t8_div; one v_rcp_f64 = 16 issue cycles, four times a DP multiply -- scripts/microbench/valu_rate.hip).
T8_DEV float  t8_rcp_shared(float b) { return __builtin_amdgcn_rcpf(b); }
T8_DEV double t8_rcp_shared(double b) {
  const double r = __builtin_amdgcn_rcp(b);
  return __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
}
T8_DEV float  t8_div_by(float a, float /*b*/, float rb) { return a * rb; }
T8_DEV double t8_div_by(double a, double b, double rb) {
  const double q = a * rb;
  return __builtin_fma(__builtin_fma(-b, q, a), rb, q);
}
// sqrt(y / x) = y * rsqrt(y * x) for x, y > 0 (normal range): one v_rsq instead of a division and a square root
T8_DEV float t8_sqrt_ratio(float y, float x) { return y * __builtin_amdgcn_rsqf(y * x); }
T8_DEV double t8_sqrt_ratio(double y, double x) {
  const double z  = y * x;
  const double r0 = __builtin_amdgcn_rsq(z);
  const double g  = z * r0;                              // ~ sqrt(z)
  double       h  = 0.5 * r0;                            // ~ 1 / (2 sqrt(z)); one Goldschmidt step -> ~2^-51
  h               = __builtin_fma(h, __builtin_fma(-h, g, 0.5), h);
  const double a  = (y + y) * h;                         // y / sqrt(y x)
  const double d  = __builtin_fma(-a, a * x, y);         // residual y - a^2 x; 1 / (2 a x) = h
  return __builtin_fma(d, h, a);
}

// dt / vol of the RK update. Where vol is a power of two -- every Cartesian mesh: element volumes are 2^-k of the unit domain --
// its reciprocal is one integer subtraction on the exponent field, exact, and dt * (1 / vol) IS dt / vol bit for bit: five
// instructions with the test instead of the eleven (one of them quarter-rate) of an IEEE division. Anything else takes the division.
// Used where the volume is WAVE-UNIFORM (2D patches of uniform volume: the test runs on scalar registers; c4 fp32 +2 %, c2 +1.4 %,
// c4 fp64 unchanged); with a per-lane volume the test costs what it saves (Subgrid fp32 -2 %), and the 3D patch kernels have no
// register to spare for it (14 - 21 spills): those keep the division.
// (Normal range only: volumes of 2^-1021 .. 2^1022.)
T8_DEV double rk_scale(double dt, double vol) {
  const int hi = __double2hiint(vol), lo = __double2loint(vol);
  // mantissa bits all zero, exponent field in [2, 0x7FC]: a normal power of two whose reciprocal is normal too
  if ((lo | (hi & 0x000FFFFF)) == 0 && static_cast<unsigned>(hi - 0x00200000) < 0x7FB00000u) return dt * __hiloint2double(0x7FE00000 - hi, 0);
  return dt / vol;
}
T8_DEV float rk_scale(float dt, float vol) {
  const int b = __float_as_int(vol);
  if ((b & 0x007FFFFF) == 0 && static_cast<unsigned>(b - 0x01000000) < 0x7D800000u) return dt * __int_as_float(0x7F000000 - b);
  return dt / vol;
}

// Stage results and the previous step's state are touched ONCE per stage kernel. Where the planes of a stage (previous, source,
// result: 15 of them) are much larger than the 256 MB Infinity Cache, caching them only evicts what is about to be re-read (the
// source states of neighbouring tiles): non-temporal accesses there (NT instantiations of the patch and family kernels: c4 fp64
// +2.6 %, fp32 +4.9 %, c3 +2 - 3 %, c5 +2.5 %) and ordinary ones where the working set stays resident (c2, 1.03 M elements: -8 % with
// non-temporal stores; c5u -1.5 %). A compile-time choice: as a wave-uniform run-time branch the two arms have to be kept apart
// with scheduling barriers (the optimiser merges them and drops the hint otherwise), which cost 2 - 5 % -- more than the gain.
template <bool NT, class T>
T8_DEV T stream_load(const T* p) {
  if constexpr (NT)
    return __builtin_nontemporal_load(p);
  else
    return *p;
}
template <bool NT, class T>
T8_DEV void stream_store(T* p, T v) {
  if constexpr (NT)
    __builtin_nontemporal_store(v, p);
  else
    *p = v;
}
// The launchers' test: T8GPU_STREAM_MB = the threshold in MB (default 384 = 1.5 x the cache), 0 = never.
inline bool stream_hint(long long cells, size_t float_size) {
  static const long long mb = [] {
    const char* env = std::getenv("T8GPU_STREAM_MB");
    return env ? std::atoll(env) : 384ll;
  }();
  return mb > 0 && cells > 0 && 15ll * cells * static_cast<long long>(float_size) > mb * (1ll << 20);
}

// 1 / sqrt(x) for the fast tier (x > 0, normal range): the hardware seed (v_rsq_f64: ~2^-23) and two Newton steps, ~1 ulp;
// fp32: v_rsq_f32 is 1 ulp as it is. 1 for x = 1 exactly (seed 1, residual 0).
T8_DEV float  t8_rsqrt_fast(float x) { return __builtin_amdgcn_rsqf(x); }
T8_DEV double t8_rsqrt_fast(double x) {
  double       r = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  r              = __builtin_fma(r, __builtin_fma(-(h * r), r, 0.5), r);
  r              = __builtin_fma(r, __builtin_fma(-(h * r), r, 0.5), r);
  return r;
}

// face_basis() for the fused kernels that build the frame per face and stage (meshes without a geometry dictionary: every
// face of a curved mesh has its own normal): the same construction with the normalisation as ONE reciprocal square root and
// three products instead of an IEEE square root and three IEEE divisions -- 25 VALU instructions instead of 75 in fp64, a tenth
// of such a face's work. The tangents differ from face_basis()'s in the last bit or two (the flux does not depend on their
// choice; the dictionary's frames, computed on the host, differ from the device's by as much already:
// tests/test_gpu_unstructured.py::test_fused_variants_agree_and_conserve); for n = +-e_axis both give the exact frame.
template <class T>
T8_DEV void face_basis_fast(const T n[3], T t1[3], T t2[3]) {
#pragma clang fp contract(off)
  t1[0] = n[1];
  t1[1] = n[2];
  t1[2] = -n[0];
  const T dp = n[0] * t1[0] + n[1] * t1[1] + n[2] * t1[2];
  t1[0] -= dp * n[0];
  t1[1] -= dp * n[1];
  t1[2] -= dp * n[2];
  const T inv = t8_rsqrt_fast(t1[0] * t1[0] + t1[1] * t1[1] + t1[2] * t1[2]);
  t1[0] *= inv;
  t1[1] *= inv;
  t1[2] *= inv;
  t2[0] = n[1] * t1[2] - n[2] * t1[1];
  t2[1] = n[2] * t1[0] - n[0] * t1[2];
  t2[2] = n[0] * t1[1] - n[1] * t1[0];
}

// log for the fast tier (x > 0, normal range). fp64: the library routine is ~95 VALU instructions (special
// cases, double-double reduction); per element and stage two of them dominated the per-cell work. This
// is the classic reduction x = m 2^e, m in [sqrt(1/2), sqrt(2)), log m = 2 atanh(s), s = (m-1)/(m+1), with
// the degree-7 minimax polynomial in s^2 of Sun's fdlibm e_log.c (error < 1 ulp; pinned against the
// host libm in tests/test_gpu_fastmath.py): ~35 instructions.
// fp32: the hardware log2 (v_log_f32, ~1 ulp) times ln 2 in two pieces instead of the library logf (which wraps the same
// instruction in scaling for denormals and a correction sequence: ~10 instructions, and a cell needs two logs, a face two
// more). Inputs here are densities, pressures and their ratios: O(1), never denormal (pinned in tests/test_gpu_fastmath.py).
T8_DEV float  t8_log_fast(float x) {
  const float l2 = __builtin_amdgcn_logf(x);
  return __builtin_fmaf(l2, 9.0580015e-06f, l2 * 6.9313812256e-01f);   // ln 2 = 0.69313812 + 9.0580015e-06 (fdlibm's split)
}
T8_DEV double t8_log_fast(double x) {
#pragma clang fp contract(off)
  double     m  = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
  int        e  = __builtin_amdgcn_frexp_exp(x);
  const bool lo = m < 0.70710678118654752440;
  m             = lo ? m + m : m;
  e             = lo ? e - 1 : e;
  const double f = m - 1.0;
  const double s = t8_div(f, 2.0 + f);
  const double z = s * s, w = z * z;
  const double t1 = w * __builtin_fma(w, __builtin_fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
  const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01),
                                                         2.857142874366239149e-01), 6.666666666666735130e-01);
  const double R    = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double dk   = static_cast<double>(e);
  return __builtin_fma(dk, 6.93147180369123816490e-01, -((hfsq - __builtin_fma(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f));
}

// Table-driven fp64 logarithm for the plain tile kernels (x > 0, normal range): x = m 2^e with m in [1, 2); the top 7
// mantissa bits pick an interval whose midpoint c has rc = 1/c and lc = -log(rc) tabulated (log_table.hpp: 128 x 16 B);
// r = m rc - 1 is exact to an FMA and |r| < 2^-8, so log1p(r) needs a degree-7 Taylor polynomial, and
// log x = e ln2 + lc + log1p(r). ~14 DP instructions and one 16-byte table read instead of the ~35 DP instructions of
// t8_log_fast (one division, a degree-7 polynomial in s^2): a cell needs two logarithms, which were 3/4 of its
// per-element work. `tab` points at a copy of kLogTab in LDS (the tile kernels copy it once per workgroup).
// Accuracy ~1 ulp, pinned against the host libm in tests/test_gpu_fastmath.py.
T8_DEV double t8_log_tab(double x, const double* __restrict__ tab) {
#pragma clang fp contract(off)
  const double m  = __builtin_amdgcn_frexp_mant(x) * 2.0;              // [1, 2)
  const int    e  = __builtin_amdgcn_frexp_exp(x) - 1;
  const int    i  = (__double2hiint(m) >> 13) & 127;                    // top 7 bits of the mantissa
  const double2 t = reinterpret_cast<const double2*>(tab)[i];           // {rc, lc}
  const double r  = __builtin_fma(m, t.x, -1.0);
  const double q  = __builtin_fma(r, __builtin_fma(r, __builtin_fma(r, __builtin_fma(r, __builtin_fma(r, __builtin_fma(r,
                        1.0 / 7.0, -1.0 / 6.0), 0.2), -0.25), 1.0 / 3.0), -0.5), 1.0);
  const double dk = static_cast<double>(e);
  return __builtin_fma(dk, 6.93147180369123816490e-01, __builtin_fma(dk, 1.90821492927058770002e-10, __builtin_fma(r, q, t.y)));
}
T8_DEV float t8_log_tab(float x, const double*) { return t8_log_fast(x); }

template <class T>
struct Prim {
  T rho, vx, vy, vz, p, beta, lrho, lbeta, v0;
};
constexpr int kPrimWords = 9;

// logtab: LDS copy of kLogTab (plain tile kernels in fp64: table-driven logarithms), or null (polynomial logarithm)
template <class T, bool TAB = false>
T8_DEV Prim<T> prim_from_state(const T s[5], const double* logtab = nullptr) {
#pragma clang fp contract(off)
  const T one = T(1), half = T(0.5), kappa = T(1.4);
  const T km1 = kappa - one;
  Prim<T> q;
  const T ir = t8_rcp(s[0]);
  q.rho      = s[0];
  q.vx       = s[1] * ir;
  q.vy       = s[2] * ir;
  q.vz       = s[3] * ir;
  const T ke = half * t8_fma(q.vx, q.vx, t8_fma(q.vy, q.vy, q.vz * q.vz));
  q.p        = km1 * t8_fma(-s[0], ke, s[4]);
  const T rp = t8_div(s[0], q.p);
  q.beta     = half * rp;
  q.lrho     = TAB ? t8_log_tab(s[0], logtab) : t8_log_fast(s[0]);
  const T lp = TAB ? t8_log_tab(q.p, logtab) : t8_log_fast(q.p);
  q.lbeta    = q.lrho - lp;
  q.v0       = t8_fma(-rp, ke, (kappa - t8_fma(-kappa, q.lrho, lp)) * (one / km1));
  return q;
}

// logarithmic mean from the two values and log(aR) - log(aL)
// (f = d / s only feeds u = f^2: the branch test and the series 105 + 35 u + ..., where a relative error e of u becomes one of
//  3e-5 e of the result -- a reciprocal with one Newton step (2^-50) times d is plenty; the full quotient cost 2 more instructions)
T8_DEV double ln_mean_dlog(double aL, double aR, double dlog) {
#pragma clang fp contract(off)
  const double d = aR - aL, s = aR + aL;
  const double f = d * t8_rcp_shared(s);
  const double u = f * f;
  const bool   small = u < 1.0e-4;
  const double num = small ? s * 52.50 : d;
  const double den = small ? t8_fma(u, t8_fma(u, t8_fma(u, 15.0, 21.0), 35.0), 105.0) : dlog;
  return t8_div(num, den);
}
// the same with s = aR + aL and its (shared) reciprocal handed in
T8_DEV double ln_mean_dlog_rs(double aL, double aR, double dlog, double s, double rs) {
#pragma clang fp contract(off)
  const double d = aR - aL;
  const double f = d * rs;
  const double u = f * f;
  const bool   small = u < 1.0e-4;
  const double num = small ? s * 52.50 : d;
  const double den = small ? t8_fma(u, t8_fma(u, t8_fma(u, 15.0, 21.0), 35.0), 105.0) : dlog;
  return t8_div(num, den);
}
// 1 / (logarithmic mean), same arguments: den / num in ONE division where 1 / (num / den) takes a division and a reciprocal (the
// flux needs the mean of beta only through its reciprocal: kernels.inl:60-75, `1 / beta_hat`). num is never 0: it is the
// difference of two values at least 2 % apart, or 52.5 times their sum.
T8_DEV double ln_mean_inv_dlog_rs(double aL, double aR, double dlog, double s, double rs) {
#pragma clang fp contract(off)
  const double d = aR - aL;
  const double f = d * rs;
  const double u = f * f;
  const bool   small = u < 1.0e-4;
  const double num = small ? s * 52.50 : d;
  const double den = small ? t8_fma(u, t8_fma(u, t8_fma(u, 15.0, 21.0), 35.0), 105.0) : dlog;
  return t8_div(den, num);
}
// fp32: a difference of stored logs would cost accuracy near the branch switch; v_log_f32 is cheap.
T8_DEV float ln_mean_dlog(float aL, float aR, float /*dlog*/) {
#pragma clang fp contract(off)
  const float d = aR - aL, s = aR + aL;
  const float f = t8_div(d, s);
  const float u = f * f;
  const bool  small = u < 1.0e-4f;
  const float num = small ? s * 52.50f : d;
  const float den = small ? t8_fma(u, t8_fma(u, t8_fma(u, 15.0f, 21.0f), 35.0f), 105.0f) : t8_log_fast(t8_div(aR, aL));
  return t8_div(num, den);
}

T8_DEV float ln_mean_dlog_rs(float aL, float aR, float /*dlog*/, float s, float rs) {
#pragma clang fp contract(off)
  const float d = aR - aL;
  const float f = d * rs;
  const float u = f * f;
  const bool  small = u < 1.0e-4f;
  const float num = small ? s * 52.50f : d;
  const float den = small ? t8_fma(u, t8_fma(u, t8_fma(u, 15.0f, 21.0f), 35.0f), 105.0f) : t8_log_fast(t8_div(aR, aL));
  return t8_div(num, den);
}
T8_DEV float ln_mean_inv_dlog_rs(float aL, float aR, float /*dlog*/, float s, float rs) {
#pragma clang fp contract(off)
  const float d = aR - aL;
  const float f = d * rs;
  const float u = f * f;
  const bool  small = u < 1.0e-4f;
  const float num = small ? s * 52.50f : d;
  const float den = small ? t8_fma(u, t8_fma(u, t8_fma(u, 15.0f, 21.0f), 35.0f), 105.0f) : t8_log_fast(t8_div(aR, aL));
  return t8_div(den, num);
}

// KEPES flux through a face with unit normal n (basis n, t1, t2), scaled by `area`, in xyz.
// mirror => the right state is the wall reflection of L (R is ignored).
// core: velocities already in the face frame; f = area-scaled flux in the face frame
template <class T>
T8_DEV void kepes_core(const Prim<T>& L, const Prim<T>& R, T uL, T vL, T wL, T uR, T vR, T wR, T area, T f[5], T& speed) {
#pragma clang fp contract(off)
  const T one = T(1), half = T(0.5), kappa = T(1.4);
  const T km1 = kappa - one, skm1 = one / km1, ikappa = one / kappa;
  // (|vL|^2 + |vR|^2) / 2 as ONE scaling of the sum: halving is exact, so this is the sum of the two halves bit for bit
  const T q2 = half * (t8_fma(uL, uL, t8_fma(vL, vL, wL * wL)) + t8_fma(uR, uR, t8_fma(vR, vR, wR * wR)));

  const T bsum = L.beta + R.beta, rbs = t8_rcp_shared(bsum);   // serves the log mean of beta and the pressure mean
  const T rho  = ln_mean_dlog(L.rho, R.rho, R.lrho - L.lrho);
  const T ib   = ln_mean_inv_dlog_rs(L.beta, R.beta, R.lbeta - L.lbeta, bsum, rbs);   // 1 / (logarithmic mean of beta)
  const T rho_mean = half * (L.rho + R.rho);
  const T u = half * (uL + uR), v = half * (vL + vR), w = half * (wL + wR);
  const T a  = t8_sqrt_ratio(kappa * half * (L.p + R.p), rho);
  const T dot = t8_fma(uL, uR, t8_fma(vL, vR, wL * wR));   // vL . vR: in the enthalpy mean and in |v_mean|^2 below
  const T h   = t8_fma(kappa / (T(2) * km1), ib, half * dot);
  const T p1 = t8_div_by(rho_mean, bsum, rbs);

  // the area rides on the two quantities every term of the flux is linear in (density mean, pressure mean): two products
  // instead of five at the end (an exact rescaling where the area is a power of two: Cartesian meshes keep their bits)
  const T rho_a = rho * area, p1_a = p1 * area;
  const T Fs0 = rho_a * u;
  const T Fs1 = t8_fma(Fs0, u, p1_a);
  const T Fs2 = Fs0 * v;
  const T Fs3 = Fs0 * w;
  const T Fs4 = t8_fma(Fs0 * half, t8_fma(skm1, ib, -q2), t8_fma(u, Fs1, t8_fma(v, Fs2, w * Fs3)));

  speed = t8_abs(u) + a;

  const T au = t8_abs(u);
  const T ra = rho_a * (half * ikappa), re = rho_a * (km1 * ikappa);   // |lambda| rho / (2 gamma), |lambda| rho (gamma - 1) / gamma
  const T D0 = t8_abs(u - a) * ra;
  const T D1 = au * re;
  const T D2 = au * p1_a;
  const T D4 = t8_abs(u + a) * ra;

  const T rpL = L.beta + L.beta, rpR = R.beta + R.beta;
  const T J0 = R.v0 - L.v0;
  const T J1 = t8_fma(rpR, uR, -(rpL * uL));
  const T J2 = t8_fma(rpR, vR, -(rpL * vL));
  const T J3 = t8_fma(rpR, wR, -(rpL * wL));
  const T J4 = rpL - rpR;

  const T ua = u * a;
  // |v_mean|^2 / 2 with v_mean = (vL + vR) / 2: (|vL|^2 + |vR|^2 + 2 vL . vR) / 8 = (q2 + vL . vR) / 4 -- both terms are at hand
  const T hm = h - ua, hp = h + ua, k2 = T(0.25) * (q2 + dot);
  const T c  = t8_fma(w, J3, t8_fma(v, J2, J0));  // common part of the three acoustic/entropy columns
  const T d0 = D0 * t8_fma(hm, J4, t8_fma(u - a, J1, c));
  const T d1 = D1 * t8_fma(k2, J4, t8_fma(u, J1, c));
  const T d2 = D2 * t8_fma(v, J4, J2);
  const T d3 = D2 * t8_fma(w, J4, J3);
  const T d4 = D4 * t8_fma(hp, J4, t8_fma(u + a, J1, c));
  const T s014 = d0 + d1 + d4;
  f[0] = t8_fma(-half, s014, Fs0);
  f[1] = t8_fma(-half, t8_fma(u + a, d4, t8_fma(u, d1, (u - a) * d0)), Fs1);
  f[2] = t8_fma(-half, t8_fma(v, s014, d2), Fs2);
  f[3] = t8_fma(-half, t8_fma(w, s014, d3), Fs3);
  f[4] = t8_fma(-half, t8_fma(hp, d4, t8_fma(w, d3, t8_fma(v, d2, t8_fma(k2, d1, hm * d0)))), Fs4);
}

// The rotation into the face frame and back, spelled out once (same reason: one rounding sequence everywhere).
template <class T>
T8_DEV T dot3(T x, T y, T z, const T b[3]) {
#pragma clang fp contract(off)
  return t8_fma(z, b[2], t8_fma(y, b[1], x * b[0]));
}

template <class T>
T8_DEV void kepes_prim(const Prim<T>& L, const Prim<T>& R, bool mirror, const T n[3], const T t1[3], const T t2[3],
                       T area, T g[5], T& speed) {
#pragma clang fp contract(off)
  const T uL = dot3<T>(L.vx, L.vy, L.vz, n);
  const T vL = dot3<T>(L.vx, L.vy, L.vz, t1);
  const T wL = dot3<T>(L.vx, L.vy, L.vz, t2);
  T       uR = dot3<T>(R.vx, R.vy, R.vz, n);
  T       vR = dot3<T>(R.vx, R.vy, R.vz, t1);
  T       wR = dot3<T>(R.vx, R.vy, R.vz, t2);
  if (mirror) {
    uR = -uL;
    vR = vL;
    wR = wL;
  }
  T f[5];
  kepes_core<T>(L, R, uL, vL, wL, uR, vR, wR, area, f, speed);
  g[0] = f[0];
  g[1] = t8_fma(f[3], t2[0], t8_fma(f[2], t1[0], f[1] * n[0]));
  g[2] = t8_fma(f[3], t2[1], t8_fma(f[2], t1[1], f[1] * n[1]));
  g[3] = t8_fma(f[3], t2[2], t8_fma(f[2], t1[2], f[1] * n[2]));
  g[4] = f[4];
}

// Axis-aligned face, normal +-e_axis (Subgrid blocks: kernels.inl:717-750 requires it). The frame is
// (s e_axis, e_axis+1, e_axis+2): component selection and one sign instead of 27 multiplications by the
// zeros and ones of a general basis. The flux does not depend on the choice of the two tangents.
template <class T>
T8_DEV T axis_pick(T x, T y, T z, int a) { return a == 0 ? x : (a == 1 ? y : z); }

template <class T>
T8_DEV void kepes_axis(const Prim<T>& L, const Prim<T>& R, bool mirror, int axis, bool positive, T area, T g[5], T& speed) {
#pragma clang fp contract(off)
  const int a1 = axis == 2 ? 0 : axis + 1, a2 = axis == 0 ? 2 : axis - 1;
  const T   uLp = axis_pick(L.vx, L.vy, L.vz, axis), uRp = axis_pick(R.vx, R.vy, R.vz, axis);
  const T   uL = positive ? uLp : -uLp;
  const T   vL = axis_pick(L.vx, L.vy, L.vz, a1), wL = axis_pick(L.vx, L.vy, L.vz, a2);
  T         uR = positive ? uRp : -uRp;
  T         vR = axis_pick(R.vx, R.vy, R.vz, a1), wR = axis_pick(R.vx, R.vy, R.vz, a2);
  if (mirror) {
    uR = -uL;
    vR = vL;
    wR = wL;
  }
  T f[5];
  kepes_core<T>(L, R, uL, vL, wL, uR, vR, wR, area, f, speed);
  const T fn = positive ? f[1] : -f[1];
  g[0] = f[0];
  g[1] = axis == 0 ? fn : (axis == 1 ? f[3] : f[2]);   // x is a2 of axis 1 and a1 of axis 2
  g[2] = axis == 1 ? fn : (axis == 2 ? f[3] : f[2]);
  g[3] = axis == 2 ? fn : (axis == 0 ? f[3] : f[2]);
  g[4] = f[4];
}

// Face with normal s * e_AXIS (s = +-1) in exactly the frame face_basis() returns for that normal:
//   AXIS 0: t1 = (0, 0, -s), t2 = (0, 1, 0);   AXIS 1: t1 = (s, 0, 0), t2 = (0, 0, -1);   AXIS 2: t1 = (0, s, 0), t2 = (-1, 0, 0)
// Every product of the general rotation (kepes_prim) is then a product with 0 or +-1, so selecting components and
// signs gives the SAME values (at most the sign of a zero differs): a face may be evaluated by either form, e.g. by
// this one where a whole wavefront shares the direction and by the general one in a mixed wavefront of another tile.
template <class T, int AXIS>
T8_DEV void kepes_axis_fixed(const Prim<T>& L, const Prim<T>& R, bool mirror, T s, T area, T g[5], T& speed) {
#pragma clang fp contract(off)
  T uL, vL, wL, uR, vR, wR;
  if (AXIS == 0) {
    uL = s * L.vx; vL = -(s * L.vz); wL = L.vy;
    uR = s * R.vx; vR = -(s * R.vz); wR = R.vy;
  } else if (AXIS == 1) {
    uL = s * L.vy; vL = s * L.vx; wL = -L.vz;
    uR = s * R.vy; vR = s * R.vx; wR = -R.vz;
  } else {
    uL = s * L.vz; vL = s * L.vy; wL = -L.vx;
    uR = s * R.vz; vR = s * R.vy; wR = -R.vx;
  }
  if (mirror) {
    uR = -uL;
    vR = vL;
    wR = wL;
  }
  T f[5];
  kepes_core<T>(L, R, uL, vL, wL, uR, vR, wR, area, f, speed);
  g[0] = f[0];
  g[4] = f[4];
  if (AXIS == 0) {
    g[1] = s * f[1]; g[2] = f[3]; g[3] = -(s * f[2]);
  } else if (AXIS == 1) {
    g[1] = s * f[2]; g[2] = s * f[1]; g[3] = -f[3];
  } else {
    g[1] = -f[3]; g[2] = s * f[2]; g[3] = s * f[1];
  }
}

// HLL for the fast tier: the formulas of hll_ref (examples/subgrid/kernels.inl:263-332) with shared
// reciprocals and the fast division above; takes the states already rotated into the face frame.
template <class T>
T8_DEV void hll_fast(const T uL[5], const T uR[5], T F[5], T& speed) {
#pragma clang fp contract(off)
  const T zero = T(0), one = T(1), half = T(0.5);
  const T gm1 = T(1.4) - one;
  const T irl = t8_rcp(uL[0]), irr = t8_rcp(uR[0]);
  const T v1l = uL[1] * irl, v2l = uL[2] * irl, v3l = uL[3] * irl;
  const T v1r = uR[1] * irr, v2r = uR[2] * irr, v3r = uR[3] * irr;
  const T kl = half * t8_fma(v1l, v1l, t8_fma(v2l, v2l, v3l * v3l)), kr = half * t8_fma(v1r, v1r, t8_fma(v2r, v2r, v3r * v3r));
  const T pl = gm1 * t8_fma(-uL[0], kl, uL[4]), pr = gm1 * t8_fma(-uR[0], kr, uR[4]);
  const T Hl = (uL[4] + pl) * irl, Hr = (uR[4] + pr) * irr;
  const T cl = t8_sqrt_fast(gm1 * (Hl - kl)), cr = t8_sqrt_fast(gm1 * (Hr - kr));
  const T wl = t8_sqrt_fast(uL[0]), wr = t8_sqrt_fast(uR[0]);
  const T iw = t8_rcp(wl + wr);
  const T v1 = t8_fma(wl, v1l, wr * v1r) * iw, v2 = t8_fma(wl, v2l, wr * v2r) * iw, v3 = t8_fma(wl, v3l, wr * v3r) * iw;
  const T H  = t8_fma(wl, Hl, wr * Hr) * iw;
  const T c  = t8_sqrt_fast(gm1 * t8_fma(-half, t8_fma(v1, v1, t8_fma(v2, v2, v3 * v3)), H));
  const T Sl = t8_min(v1 - c, v1l - cl), Sr = t8_max(v1 + c, v1r + cr);
  speed      = t8_max(t8_abs(Sl), t8_abs(Sr));
  const T sl = t8_min(Sl, zero);
  const T sr = t8_max(Sr, zero);
  const T Fl[5] = {uL[1], t8_fma(uL[1], v1l, pl), uL[1] * v2l, uL[1] * v3l, uL[1] * Hl};
  const T Fr[5] = {uR[1], t8_fma(uR[1], v1r, pr), uR[1] * v2r, uR[1] * v3r, uR[1] * Hr};
  const T id = t8_rcp(sr - sl);
#pragma unroll
  for (int k = 0; k < 5; k++) F[k] = t8_fma(sr * sl, uR[k] - uL[k], t8_fma(sr, Fl[k], -(sl * Fr[k]))) * id;
}

// HLLC for the fast tier: hllc_ref with shared reciprocals and the fast division / sqrt
template <class T>
T8_DEV void hllc_fast(const T uL[5], const T uR[5], T F[5], T& speed) {
#pragma clang fp contract(off)
  const T zero = T(0), one = T(1), half = T(0.5);
  const T gm1 = T(1.4) - one;
  const T irl = t8_rcp(uL[0]), irr = t8_rcp(uR[0]);
  const T v1l = uL[1] * irl, v2l = uL[2] * irl, v3l = uL[3] * irl;
  const T v1r = uR[1] * irr, v2r = uR[2] * irr, v3r = uR[3] * irr;
  const T kl = half * t8_fma(v1l, v1l, t8_fma(v2l, v2l, v3l * v3l)), kr = half * t8_fma(v1r, v1r, t8_fma(v2r, v2r, v3r * v3r));
  const T pl = gm1 * t8_fma(-uL[0], kl, uL[4]), pr = gm1 * t8_fma(-uR[0], kr, uR[4]);
  const T Hl = (uL[4] + pl) * irl, Hr = (uR[4] + pr) * irr;
  const T cl = t8_sqrt_fast(gm1 * (Hl - kl)), cr = t8_sqrt_fast(gm1 * (Hr - kr));
  const T wl = t8_sqrt_fast(uL[0]), wr = t8_sqrt_fast(uR[0]);
  const T iw = t8_rcp(wl + wr);
  const T v1 = t8_fma(wl, v1l, wr * v1r) * iw, v2 = t8_fma(wl, v2l, wr * v2r) * iw, v3 = t8_fma(wl, v3l, wr * v3r) * iw;
  const T H  = t8_fma(wl, Hl, wr * Hr) * iw;
  const T c  = t8_sqrt_fast(gm1 * t8_fma(-half, t8_fma(v1, v1, t8_fma(v2, v2, v3 * v3)), H));
  const T Sl = t8_min(v1 - c, v1l - cl), Sr = t8_max(v1 + c, v1r + cr);
  speed = t8_max(t8_abs(Sl), t8_abs(Sr));
  const T ml = uL[0] * (Sl - v1l), mr = uR[0] * (Sr - v1r);
  const T Ss = t8_div((pr - pl) + t8_fma(uL[1], Sl - v1l, -(uR[1] * (Sr - v1r))), ml - mr);
  const bool left = Ss >= zero;
  const T    S  = left ? t8_min(Sl, zero) : t8_max(Sr, zero);
  const T    SK = left ? Sl : Sr, vn = left ? v1l : v1r, vt1 = left ? v2l : v2r, vt2 = left ? v3l : v3r;
  const T    p = left ? pl : pr, Hk = left ? Hl : Hr, m = left ? ml : mr, ir = left ? irl : irr;
  T          u[5];
#pragma unroll
  for (int k = 0; k < 5; k++) u[k] = left ? uL[k] : uR[k];
  const T fac = t8_div(m, SK - Ss);
  const T Us[5] = {fac, fac * Ss, fac * vt1, fac * vt2, fac * t8_fma(Ss - vn, Ss + t8_div(p, m), u[4] * ir)};
  const T Fk[5] = {u[1], t8_fma(u[1], vn, p), u[1] * vt1, u[1] * vt2, u[1] * Hk};
#pragma unroll
  for (int k = 0; k < 5; k++) F[k] = t8_fma(S, Us[k] - u[k], Fk[k]);
}

// face-frame HLL (hllc = false) or HLLC flux of an xyz state pair, scaled by `area`, rotated back to xyz (fast tier)
template <class T>
T8_DEV void hll_face(const T sL[5], const T sR[5], bool mirror, const T n[3], const T t1[3], const T t2[3], T area, T g[5],
                     T& speed, bool hllc = false) {
#pragma clang fp contract(off)
  T a[5], b[5], Ff[5];
  a[0] = sL[0]; a[1] = dot3<T>(sL[1], sL[2], sL[3], n); a[2] = dot3<T>(sL[1], sL[2], sL[3], t1); a[3] = dot3<T>(sL[1], sL[2], sL[3], t2); a[4] = sL[4];
  if (mirror) {   // reflective wall: mirror image of the left state (kernels.inl:169-176)
    b[0] = a[0]; b[1] = -a[1]; b[2] = a[2]; b[3] = a[3]; b[4] = a[4];
  } else {
    b[0] = sR[0]; b[1] = dot3<T>(sR[1], sR[2], sR[3], n); b[2] = dot3<T>(sR[1], sR[2], sR[3], t1); b[3] = dot3<T>(sR[1], sR[2], sR[3], t2); b[4] = sR[4];
  }
  if (hllc)
    hllc_fast<T>(a, b, Ff, speed);
  else
    hll_fast<T>(a, b, Ff, speed);
#pragma unroll
  for (int k = 0; k < 5; k++) Ff[k] = area * Ff[k];
  g[0] = Ff[0];
  g[1] = t8_fma(Ff[3], t2[0], t8_fma(Ff[2], t1[0], Ff[1] * n[0]));
  g[2] = t8_fma(Ff[3], t2[1], t8_fma(Ff[2], t1[1], Ff[1] * n[1]));
  g[3] = t8_fma(Ff[3], t2[2], t8_fma(Ff[2], t1[2], Ff[1] * n[2]));
  g[4] = Ff[4];
}

}  // namespace t8gpu_hip
