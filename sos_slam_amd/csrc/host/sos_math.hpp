// sos_math.hpp -- fp64 math floor of the host facade: SE3 with the semantics of the reference's
// vendored Sophus 0.9a (thirdparty/Sophus/sophus/se3.hpp:131-139 Adj, :168-173 inverse, :407-428 exp;
// so3.hpp:343-368 expAndTheta) and a symmetric-pivoting LDL^T solve in place of Eigen's
// `.ldlt().solve` (OB/EnergyFunctional.cpp:1148).  Eigen is not available in this image.
#pragma once

#include <immintrin.h>

#include <algorithm>
#include <cmath>
#include <memory>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <ctime>
#include <vector>

namespace sos {

struct SE3 {
  double R[9];  // row-major
  double t[3];

  SE3() {
    std::memset(R, 0, sizeof(R));
    R[0] = R[4] = R[8] = 1;
    t[0] = t[1] = t[2] = 0;
  }
  static SE3 from12(const double *p) {
    SE3 T;
    std::memcpy(T.R, p, 9 * sizeof(double));
    std::memcpy(T.t, p + 9, 3 * sizeof(double));
    return T;
  }
  void to12(double *p) const {
    std::memcpy(p, R, 9 * sizeof(double));
    std::memcpy(p + 9, t, 3 * sizeof(double));
  }
  SE3 operator*(const SE3 &o) const {
    SE3 c;
    for (int i = 0; i < 3; i++) {
      for (int j = 0; j < 3; j++) c.R[3 * i + j] = R[3 * i] * o.R[j] + R[3 * i + 1] * o.R[3 + j] + R[3 * i + 2] * o.R[6 + j];
      c.t[i] = t[i] + (R[3 * i] * o.t[0] + R[3 * i + 1] * o.t[1] + R[3 * i + 2] * o.t[2]);
    }
    return c;
  }
  SE3 inverse() const {
    SE3 c;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) c.R[3 * i + j] = R[3 * j + i];
    for (int i = 0; i < 3; i++) c.t[i] = -(c.R[3 * i] * t[0] + c.R[3 * i + 1] * t[1] + c.R[3 * i + 2] * t[2]);
    return c;
  }
  // 6x6 row-major [R, hat(t) R; 0, R]
  void Adj(double *Ad) const {
    const double H[9] = {0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0};
    std::memset(Ad, 0, 36 * sizeof(double));
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        Ad[6 * i + j] = R[3 * i + j];
        Ad[6 * (i + 3) + j + 3] = R[3 * i + j];
        Ad[6 * i + j + 3] = H[3 * i] * R[j] + H[3 * i + 1] * R[3 + j] + H[3 * i + 2] * R[6 + j];
      }
  }
  // tangent of this transform (Sophus se3.hpp:437-470 with so3.hpp:491-526 on the matrix -> quaternion conversion)
  void log(double *a) const {
    double q[4];  // w x y z
    const double tr = R[0] + R[4] + R[8];
    if (tr > 0) {
      double s = std::sqrt(tr + 1.0);
      q[0] = 0.5 * s;
      s = 0.5 / s;
      q[1] = (R[7] - R[5]) * s;
      q[2] = (R[2] - R[6]) * s;
      q[3] = (R[3] - R[1]) * s;
    } else {
      int i = 0;
      if (R[4] > R[0]) i = 1;
      if (R[8] > R[4 * i]) i = 2;
      const int j = (i + 1) % 3, k = (j + 1) % 3;
      double s = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
      q[1 + i] = 0.5 * s;
      s = 0.5 / s;
      q[0] = (R[3 * k + j] - R[3 * j + k]) * s;
      q[1 + j] = (R[3 * j + i] + R[3 * i + j]) * s;
      q[1 + k] = (R[3 * k + i] + R[3 * i + k]) * s;
    }
    const double sq = q[1] * q[1] + q[2] * q[2] + q[3] * q[3], nrm = std::sqrt(sq), w = q[0];
    double f;
    if (nrm < 1e-10) f = 2.0 / w - 2.0 * sq / (w * w * w);
    else if (std::fabs(w) < 1e-10) f = (w > 0 ? M_PI : -M_PI) / nrm;
    else f = 2.0 * std::atan(nrm / w) / nrm;
    const double om[3] = {f * q[1], f * q[2], f * q[3]};
    const double theta = f * nrm;
    const double Om[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
    double Om2[9], Vi[9];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) Om2[3 * r + c] = Om[3 * r] * Om[c] + Om[3 * r + 1] * Om[3 + c] + Om[3 * r + 2] * Om[6 + c];
    double c2;
    if (std::fabs(theta) < 1e-10) c2 = 1.0 / 12.0;
    else c2 = (1.0 - theta / (2.0 * std::tan(0.5 * theta))) / (theta * theta);
    for (int i = 0; i < 9; i++) Vi[i] = ((i % 4 == 0) ? 1.0 : 0.0) - 0.5 * Om[i] + c2 * Om2[i];
    for (int i = 0; i < 3; i++) {
      a[i] = Vi[3 * i] * t[0] + Vi[3 * i + 1] * t[1] + Vi[3 * i + 2] * t[2];
      a[3 + i] = om[i];
    }
  }
  // tangent = [upsilon(3), omega(3)]
  static SE3 exp(const double *a) {
    const double eps = 1e-10;
    const double *om = a + 3;
    const double theta_sq = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
    const double theta = std::sqrt(theta_sq);
    double imag, real;
    if (theta < eps) {
      const double po4 = theta_sq * theta_sq;
      imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * po4;
      real = 1.0 - 0.5 * theta_sq + (1.0 / 384.0) * po4;
    } else {
      imag = std::sin(0.5 * theta) / theta;
      real = std::cos(0.5 * theta);
    }
    double qw = real, qx = imag * om[0], qy = imag * om[1], qz = imag * om[2];
    const double nrm = std::sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
    qw /= nrm; qx /= nrm; qy /= nrm; qz /= nrm;
    SE3 T;
    const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
    const double twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx,
                 tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    T.R[0] = 1 - (tyy + tzz); T.R[1] = txy - twz; T.R[2] = txz + twy;
    T.R[3] = txy + twz; T.R[4] = 1 - (txx + tzz); T.R[5] = tyz - twx;
    T.R[6] = txz - twy; T.R[7] = tyz + twx; T.R[8] = 1 - (txx + tyy);
    const double Om[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
    double Om2[9], V[9];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) Om2[3 * i + j] = Om[3 * i] * Om[j] + Om[3 * i + 1] * Om[3 + j] + Om[3 * i + 2] * Om[6 + j];
    if (theta < eps) {
      std::memcpy(V, T.R, sizeof(V));
    } else {
      const double c1 = (1.0 - std::cos(theta)) / theta_sq, c2 = (theta - std::sin(theta)) / (theta_sq * theta);
      for (int i = 0; i < 9; i++) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + c1 * Om[i] + c2 * Om2[i];
    }
    for (int i = 0; i < 3; i++) T.t[i] = V[3 * i] * a[0] + V[3 * i + 1] * a[1] + V[3 * i + 2] * a[2];
    return T;
  }
};

// x = A^-1 b for symmetric A (n x n row-major) by LDL^T with symmetric pivoting on the largest |diagonal|;
// exact-zero pivots contribute nothing (Eigen LDLT::solve semantics, OB/EnergyFunctional.cpp:1148).
// Reference implementation: unblocked right-looking elimination on the lower triangle.
inline void ldlt_solve_ref(const std::vector<double> &A, const std::vector<double> &b, std::vector<double> &x, int n) {
  const size_t N = (size_t)n;
  std::vector<double> M(N * N), D(N), y(N), ck(N);
  std::vector<int> perm(N);
  for (int i = 0; i < n; i++) {
    perm[i] = i;
    for (int j = 0; j <= i; j++) M[i * N + j] = A[i * N + j];  // lower triangle
  }
  auto at = [&](int i, int j) -> double & { return i >= j ? M[(size_t)i * N + j] : M[(size_t)j * N + i]; };
  for (int k = 0; k < n; k++) {
    int p = k;
    double best = std::fabs(M[k * N + k]);
    for (int i = k + 1; i < n; i++)
      if (std::fabs(M[i * N + i]) > best) { best = std::fabs(M[i * N + i]); p = i; }
    if (p != k) {  // symmetric swap of rows/columns k and p within the lower triangle (L part included)
      for (int j = 0; j < k; j++) std::swap(M[k * N + j], M[p * N + j]);
      std::swap(M[k * N + k], M[p * N + p]);
      for (int j = k + 1; j < p; j++) std::swap(at(j, k), at(p, j));
      for (int j = p + 1; j < n; j++) std::swap(at(j, k), at(j, p));
      std::swap(perm[k], perm[p]);
    }
    const double d = M[k * N + k];
    D[k] = d;
    if (!(std::fabs(d) > 2.2250738585072014e-308)) {
      for (int i = k + 1; i < n; i++) M[i * N + k] = 0.0;
      continue;
    }
    for (int i = k + 1; i < n; i++) ck[i] = M[i * N + k];
    for (int i = k + 1; i < n; i++) {
      const double lik = ck[i] / d;
      double *__restrict row = &M[i * N];
      const double *__restrict c = ck.data();
      for (int j = k + 1; j <= i; j++) row[j] -= lik * c[j];
      row[k] = lik;  // L(i,k)
    }
  }
  for (int i = 0; i < n; i++) y[i] = b[perm[i]];
  for (int i = 0; i < n; i++) {
    double s = y[i];
    const double *row = &M[i * N];
    for (int j = 0; j < i; j++) s -= row[j] * y[j];
    y[i] = s;
  }
  for (int i = 0; i < n; i++) y[i] = (std::fabs(D[i]) > 2.2250738585072014e-308) ? y[i] / D[i] : 0.0;
  for (int i = n - 1; i >= 0; i--) {
    const double yi = y[i];
    const double *row = &M[i * N];
    for (int j = 0; j < i; j++) y[j] -= row[j] * yi;
  }
  x.assign(N, 0.0);
  for (int i = 0; i < n; i++) x[perm[i]] = y[i];
}

// ---- production LDL^T (the (4+8n)-dim solve sits on the critical path of every Gauss-Newton iteration) ----------
// Diagonal pivoting like ldlt_solve_ref (largest |diagonal|), relaxed to threshold pivoting (see below), and the same
// zero-pivot semantics, organised for speed: the matrix is held as the UPPER triangle row-major (U[j][i], j < i), so that the sub-diagonal part of
// column k of L is the contiguous row k of U; the trailing matrix is updated once per panel of LDLT_NB pivots
// (rank-NB update from the contiguous panel copies WT = L*D and LT = L); inside a panel the candidate diagonal is
// kept up to date separately and a column is brought up to date only when it becomes the pivot column.
// AVX2 + FMA: this is fp64 host arithmetic feeding a linear solve; the no-contraction rule of the fp32 residual
// path does not apply here.
#define LDLT_NB 8
__attribute__((target("avx2,fma"))) inline int ldlt_argmax_abs(const double *v, int lo, int hi) {
  const __m256d sign = _mm256_set1_pd(-0.0);
  __m256d m = _mm256_setzero_pd();
  int i = lo;
  double best = 0.0;
  for (; i + 4 <= hi; i += 4) m = _mm256_max_pd(m, _mm256_andnot_pd(sign, _mm256_loadu_pd(v + i)));
  double t[4];
  _mm256_storeu_pd(t, m);
  best = std::max(std::max(t[0], t[1]), std::max(t[2], t[3]));
  for (; i < hi; i++) best = std::max(best, std::fabs(v[i]));
  const __m256d vb = _mm256_set1_pd(best);
  for (i = lo; i + 4 <= hi; i += 4) {
    const int m = _mm256_movemask_pd(_mm256_cmp_pd(_mm256_andnot_pd(sign, _mm256_loadu_pd(v + i)), vb, _CMP_EQ_OQ));
    if (m) return i + __builtin_ctz(m);
  }
  for (; i < hi; i++)
    if (std::fabs(v[i]) == best) return i;
  return lo;  // only reached with NaNs in v: keep the current pivot
}
// U[j][i] -= sum_c L(j,c) * W(i,c) for k1 <= j < i < n.  Three rows j at a time and the panel in two halves of four
// columns: 12 broadcast registers + 3 accumulators + 1 operand fill the 16 ymm registers, so the inner loop runs on
// the FMA ports (12 FMAs per 4 + 3 loads) instead of on the load ports.
__attribute__((target("avx2,fma"))) inline void ldlt_trailing_update(double *U, const double *WT, const double *LT, int n, int k1, int kb) {
  const size_t N = (size_t)n;
  int j = k1;
  for (; j + 2 < n; j += 3) {
    double *r0 = U + (size_t)j * N, *r1 = r0 + N, *r2 = r1 + N;
    // leading entries that are not common to the three rows: (j, j+1), (j, j+2), (j+1, j+2)
    {
      double s01 = 0, s02 = 0, s12 = 0;
      for (int c = 0; c < kb; c++) {
        const double l0 = LT[c * N + j], l1 = LT[c * N + j + 1];
        s01 += l0 * WT[c * N + j + 1];
        s02 += l0 * WT[c * N + j + 2];
        s12 += l1 * WT[c * N + j + 2];
      }
      r0[j + 1] -= s01;
      r0[j + 2] -= s02;
      r1[j + 2] -= s12;
    }
    for (int c0 = 0; c0 < kb; c0 += 4) {
      const int cb = std::min(4, kb - c0);
      __m256d l0[4], l1[4], l2[4];
      for (int c = 0; c < 4; c++) {
        const bool on = c < cb;
        l0[c] = _mm256_set1_pd(on ? LT[(c0 + c) * N + j] : 0.0);
        l1[c] = _mm256_set1_pd(on ? LT[(c0 + c) * N + j + 1] : 0.0);
        l2[c] = _mm256_set1_pd(on ? LT[(c0 + c) * N + j + 2] : 0.0);
      }
      const double *w0 = WT + (size_t)(c0 + 0) * N, *w1 = WT + (size_t)(c0 + (cb > 1 ? 1 : 0)) * N,
                   *w2 = WT + (size_t)(c0 + (cb > 2 ? 2 : 0)) * N, *w3 = WT + (size_t)(c0 + (cb > 3 ? 3 : 0)) * N;
      int i = j + 3;
      for (; i + 4 <= n; i += 4) {
        __m256d a0 = _mm256_loadu_pd(r0 + i), a1 = _mm256_loadu_pd(r1 + i), a2 = _mm256_loadu_pd(r2 + i);
        __m256d w = _mm256_loadu_pd(w0 + i);
        a0 = _mm256_fnmadd_pd(l0[0], w, a0); a1 = _mm256_fnmadd_pd(l1[0], w, a1); a2 = _mm256_fnmadd_pd(l2[0], w, a2);
        w = _mm256_loadu_pd(w1 + i);
        a0 = _mm256_fnmadd_pd(l0[1], w, a0); a1 = _mm256_fnmadd_pd(l1[1], w, a1); a2 = _mm256_fnmadd_pd(l2[1], w, a2);
        w = _mm256_loadu_pd(w2 + i);
        a0 = _mm256_fnmadd_pd(l0[2], w, a0); a1 = _mm256_fnmadd_pd(l1[2], w, a1); a2 = _mm256_fnmadd_pd(l2[2], w, a2);
        w = _mm256_loadu_pd(w3 + i);
        a0 = _mm256_fnmadd_pd(l0[3], w, a0); a1 = _mm256_fnmadd_pd(l1[3], w, a1); a2 = _mm256_fnmadd_pd(l2[3], w, a2);
        _mm256_storeu_pd(r0 + i, a0);
        _mm256_storeu_pd(r1 + i, a1);
        _mm256_storeu_pd(r2 + i, a2);
      }
      for (; i < n; i++) {
        double s0 = 0, s1 = 0, s2 = 0;
        for (int c = 0; c < cb; c++) {
          const double w = WT[(c0 + c) * N + i];
          s0 += LT[(c0 + c) * N + j] * w; s1 += LT[(c0 + c) * N + j + 1] * w; s2 += LT[(c0 + c) * N + j + 2] * w;
        }
        r0[i] -= s0; r1[i] -= s1; r2[i] -= s2;
      }
    }
  }
  for (; j + 1 < n; j++) {  // at most two remaining rows (the last one has no entries right of the diagonal)
    double *r0 = U + (size_t)j * N;
    for (int i = j + 1; i < n; i++) {
      double s0 = 0;
      for (int c = 0; c < kb; c++) s0 += LT[c * N + j] * WT[c * N + i];
      r0[i] -= s0;
    }
  }
}
// The same update with 512-bit vectors where the host has them (the bench host is a Zen 5 EPYC): the whole panel of eight columns
// in one pass -- 24 broadcast registers + 3 accumulators + 1 operand of the 32 zmm registers -- and masked tails instead of scalar
// remainders.  Per element the operations are the ones of the 256-bit version's vector body in the same order (a chain of fnmadd
// over the panel columns, ascending); that version's scalar remainders sum the products first, so the two paths agree to rounding,
// not bit for bit.
__attribute__((target("avx512f,fma"))) inline void ldlt_trailing_update_512(double *U, const double *WT, const double *LT, int n, int k1, int kb) {
  const size_t N = (size_t)n;
  int j = k1;
  for (; j + 2 < n; j += 3) {
    double *r0 = U + (size_t)j * N, *r1 = r0 + N, *r2 = r1 + N;
    __m512d l0[8], l1[8], l2[8];
    const double *w[8];
    for (int c = 0; c < 8; c++) {
      const bool on = c < kb;
      l0[c] = _mm512_set1_pd(on ? LT[c * N + j] : 0.0);
      l1[c] = _mm512_set1_pd(on ? LT[c * N + j + 1] : 0.0);
      l2[c] = _mm512_set1_pd(on ? LT[c * N + j + 2] : 0.0);
      w[c] = WT + (size_t)(on ? c : 0) * N;
    }
    // the columns start at j + 1: the entries between the rows of the group -- (j, j+1), (j, j+2), (j+1, j+2) -- ride in the first
    // vector, whose store masks leave out what is not right of each row's diagonal (they were a scalar prologue of 3 kb strided
    // multiply-adds per group, a third of a group's time at dimension 101)
    for (int i = j + 1; i < n; i += 8) {
      const __mmask8 m = (n - i >= 8) ? (__mmask8)0xff : (__mmask8)((1u << (n - i)) - 1u);
      const __mmask8 m1 = (i == j + 1) ? (__mmask8)(m & 0xfe) : m, m2 = (i == j + 1) ? (__mmask8)(m & 0xfc) : m;
      __m512d a0 = _mm512_maskz_loadu_pd(m, r0 + i), a1 = _mm512_maskz_loadu_pd(m1, r1 + i), a2 = _mm512_maskz_loadu_pd(m2, r2 + i);
#pragma GCC unroll 8
      for (int c = 0; c < 8; c++) {
        const __m512d wv = _mm512_maskz_loadu_pd(m, w[c] + i);
        a0 = _mm512_fnmadd_pd(l0[c], wv, a0);
        a1 = _mm512_fnmadd_pd(l1[c], wv, a1);
        a2 = _mm512_fnmadd_pd(l2[c], wv, a2);
      }
      _mm512_mask_storeu_pd(r0 + i, m, a0);
      _mm512_mask_storeu_pd(r1 + i, m1, a1);
      _mm512_mask_storeu_pd(r2 + i, m2, a2);
    }
  }
  for (; j + 1 < n; j++) {
    double *r0 = U + (size_t)j * N;
    for (int i = j + 1; i < n; i++) {
      double s0 = 0;
      for (int c = 0; c < kb; c++) s0 += LT[c * N + j] * WT[c * N + i];
      r0[i] -= s0;
    }
  }
}
// one pivot of a panel in ONE pass over rows k+1..n (512-bit): column k brought up to date with the q earlier pivots of the panel,
// L(i,k) = a / d into both copies, the candidate diagonal updated -- the operations of the three loops of the 256-bit path, per
// element in the same order -- and the largest |diagonal| that is left, so that the next pivot's threshold test needs no search
__attribute__((target("avx512f,fma"))) inline double ldlt_pivot_512(double *uk, const double *WT, const double *LT, double *wt, double *lt, double *diag,
                                                                    int n, int k, int q, double dinv) {
  const size_t N = (size_t)n;
  __m512d lk[LDLT_NB];
  for (int c = 0; c < q; c++) lk[c] = _mm512_set1_pd(LT[c * N + k]);
  const __m512d dv = _mm512_set1_pd(dinv);
  __m512d mx = _mm512_setzero_pd();
  for (int i = k + 1; i < n; i += 8) {
    const __mmask8 m = (n - i >= 8) ? (__mmask8)0xff : (__mmask8)((1u << (n - i)) - 1u);
    __m512d a = _mm512_maskz_loadu_pd(m, uk + i);
    for (int c = 0; c < q; c++) a = _mm512_fnmadd_pd(_mm512_maskz_loadu_pd(m, &WT[c * N + i]), lk[c], a);
    const __m512d l = _mm512_mul_pd(a, dv);
    _mm512_mask_storeu_pd(wt + i, m, a);
    _mm512_mask_storeu_pd(lt + i, m, l);
    _mm512_mask_storeu_pd(uk + i, m, l);
    const __m512d dg = _mm512_sub_pd(_mm512_maskz_loadu_pd(m, diag + i), _mm512_mul_pd(a, l));
    _mm512_mask_storeu_pd(diag + i, m, dg);
    mx = _mm512_max_pd(mx, _mm512_abs_pd(dg));
  }
  return _mm512_reduce_max_pd(mx);
}
inline bool ldlt_have_avx512() {
  static const bool have = __builtin_cpu_supports("avx512f") && getenv("SOS_NO_AVX512") == nullptr;
  return have;
}

// The factorisation proper: eliminates the pivots [0, stop) of the n x n symmetric matrix held as the strict upper triangle of U
// (row-major) with its diagonal in diag[]; pivot candidates are [k, pend).  stop == pend == n is the full factorisation of
// ldlt_solve; stop < n leaves the Schur complement of the eliminated block in the rows >= stop of U (strict upper part) and in
// diag[stop..n) (ldlt_partial_* below).  D[k], perm[k] for k < stop; row k of U = the sub-diagonal part of column k of L.
__attribute__((target("avx2,fma"))) inline void ldlt_factor(double *U, int n, int stop, int pend, double *D, int *perm, double *diag, double *WTp, double *LTp) {
  const size_t N = (size_t)n;
  const bool wide = ldlt_have_avx512();
  const bool whole = pend == n;  // the 512-bit pivot pass reports the largest remaining |diagonal| over [k + 1, n): usable only then
  double restMax = -1.0;  // max |diag[k..n)| when the previous pivot's pass has left it (512-bit path), else < 0
  for (int k0 = 0; k0 < stop; k0 += LDLT_NB) {
    const int kb = std::min(LDLT_NB, stop - k0), k1 = k0 + kb;
    for (int k = k0; k < k1; k++) {
      const int q = k - k0;  // pivots of this panel already eliminated
      // threshold pivoting: the natural pivot is kept while it is within a factor 10 of the largest candidate (element
      // growth stays bounded by that factor); interchanges -- strided row/column swaps -- happen only when they buy
      // stability.  Jacobi-scaled normal matrices (diagonal ~ 1) hardly ever need one.  The 512-bit path knows the largest
      // candidate from the previous pivot's pass and searches for its position only when the test fails.
      int p = k;
      if (!(whole && restMax >= 0.0 && std::fabs(diag[k]) >= 0.1 * restMax)) {
        p = ldlt_argmax_abs(diag, k, pend);
        if (std::fabs(diag[k]) >= 0.1 * std::fabs(diag[p])) p = k;
      }
      restMax = -1.0;
      perm[k] = p;  // interchange k (LAPACK ipiv style)
      if (p != k) {  // symmetric swap k <-> p (k < p) of the not yet eliminated part.  Finished L columns keep the row
                     // order they were computed in; the substitutions below replay the interchanges one by one instead
        for (int j = k + 1; j < p; j++) std::swap(U[k * N + j], U[j * N + p]);
        double *rk = &U[k * N], *rp = &U[p * N];
        for (int i = p + 1; i < n; i++) std::swap(rk[i], rp[i]);
        std::swap(diag[k], diag[p]);
        for (int c = 0; c < q; c++) { std::swap(WTp[c * N + k], WTp[c * N + p]); std::swap(LTp[c * N + k], LTp[c * N + p]); }
      }
      const double d = diag[k];
      D[k] = d;
      double *wt = &WTp[(size_t)q * N], *lt = &LTp[(size_t)q * N];  // column k of L*D and of L
      double *uk = &U[k * N];
      if (!(std::fabs(d) > 2.2250738585072014e-308)) {
        for (int i = k + 1; i < n; i++) { uk[i] = 0.0; wt[i] = 0.0; lt[i] = 0.0; }
        continue;
      }
      if (wide) {
        restMax = ldlt_pivot_512(uk, WTp, LTp, wt, lt, diag, n, k, q, 1.0 / d);
        continue;
      }
      // bring column k (rows below the diagonal) up to date with the q earlier pivots of the panel
      {
        int i = k + 1;
        __m256d lk[LDLT_NB];
        for (int c = 0; c < q; c++) lk[c] = _mm256_set1_pd(LTp[c * N + k]);
        for (; i + 4 <= n; i += 4) {
          __m256d a = _mm256_loadu_pd(uk + i);
          for (int c = 0; c < q; c++) a = _mm256_fnmadd_pd(_mm256_loadu_pd(&WTp[c * N + i]), lk[c], a);
          _mm256_storeu_pd(wt + i, a);
        }
        for (; i < n; i++) {
          double a = uk[i];
          for (int c = 0; c < q; c++) a -= WTp[c * N + i] * LTp[c * N + k];
          wt[i] = a;
        }
      }
      const double dinv = 1.0 / d;
      for (int i = k + 1; i < n; i++) {
        const double a = wt[i], l = a * dinv;
        lt[i] = l;
        uk[i] = l;  // L(i,k)
        diag[i] -= a * l;
      }
    }
    if (k1 < n) {
      if (wide) ldlt_trailing_update_512(U, WTp, LTp, n, k1, kb);
      else ldlt_trailing_update(U, WTp, LTp, n, k1, kb);
    }
  }
}
// Both substitutions of ldlt_solve in blocks of eight pivots (512-bit): one step of the plain loops is a store -> load round trip on y
// (the next pivot's entry has just been written) plus a loop start-up, 2 n of them in a row -- 5 of the 20 us of a 101-dimensional
// solve.  A block does its 8 x 8 triangle in scalars and the rows beyond it as ONE rank-8 update (forward) / eight simultaneous dot
// products (backward), whose vectors are independent of each other.  A block with an interchange in it (rare on Jacobi-scaled
// systems: the interchanges are replayed one by one, in order, and may reach beyond the block) takes the plain steps.
__attribute__((target("avx512f,fma"))) inline void ldlt_substitute_512(const double *U, const double *D, const int *perm, double *y, int n) {
  const size_t N = (size_t)n;
  auto plain_block = [&](int k0, int k1) {
    for (int k = k0; k < k1; k++) {
      if (perm[k] != k) return false;
    }
    return true;
  };
  for (int k0 = 0; k0 < n; k0 += 8) {  // L z = P b
    const int k1 = std::min(n, k0 + 8);
    if (k1 - k0 < 8 || !plain_block(k0, k1)) {
      for (int k = k0; k < k1; k++) {
        std::swap(y[k], y[perm[k]]);
        const double yk = y[k];
        const double *uk = &U[k * N];
        for (int i = k + 1; i < n; i++) y[i] -= uk[i] * yk;
      }
      continue;
    }
    __m512d yk[8];
    for (int c = 0; c < 8; c++) {
      const double v = y[k0 + c];
      const double *uk = &U[(size_t)(k0 + c) * N];
      for (int c2 = c + 1; c2 < 8; c2++) y[k0 + c2] -= uk[k0 + c2] * v;
      yk[c] = _mm512_set1_pd(v);
    }
    const double *u0 = &U[(size_t)k0 * N];
    for (int i = k1; i < n; i += 8) {
      const __mmask8 mk = (n - i >= 8) ? (__mmask8)0xff : (__mmask8)((1u << (n - i)) - 1u);
      __m512d a = _mm512_maskz_loadu_pd(mk, y + i);
#pragma GCC unroll 8
      for (int c = 0; c < 8; c++) a = _mm512_fnmadd_pd(_mm512_maskz_loadu_pd(mk, u0 + (size_t)c * N + i), yk[c], a);
      _mm512_mask_storeu_pd(y + i, mk, a);
    }
  }
  for (int i = 0; i < n; i++) y[i] = (std::fabs(D[i]) > 2.2250738585072014e-308) ? y[i] / D[i] : 0.0;
  const int full = n / 8 * 8;  // blocks [0, 8), ..., [full - 8, full); the ragged block [full, n) comes first and takes plain steps
  for (int kb = n; kb > 0;) {  // L^T w = z
    const int k0 = (kb > full) ? full : kb - 8, k1 = kb;
    kb = k0;
    if (k1 - k0 < 8 || !plain_block(k0, k1)) {
      for (int k = k1 - 1; k >= k0; k--) {
        const double *uk = &U[k * N];
        double dot = 0;
        for (int i = k + 1; i < n; i++) dot += uk[i] * y[i];
        y[k] -= dot;
        std::swap(y[k], y[perm[k]]);
      }
      continue;
    }
    __m512d acc[8];
    for (int c = 0; c < 8; c++) acc[c] = _mm512_setzero_pd();
    const double *u0 = &U[(size_t)k0 * N];
    for (int i = k1; i < n; i += 8) {
      const __mmask8 mk = (n - i >= 8) ? (__mmask8)0xff : (__mmask8)((1u << (n - i)) - 1u);
      const __m512d yv = _mm512_maskz_loadu_pd(mk, y + i);
#pragma GCC unroll 8
      for (int c = 0; c < 8; c++) acc[c] = _mm512_fmadd_pd(_mm512_maskz_loadu_pd(mk, u0 + (size_t)c * N + i), yv, acc[c]);
    }
    for (int c = 7; c >= 0; c--) {
      const double *uk = &U[(size_t)(k0 + c) * N];
      double dot = _mm512_reduce_add_pd(acc[c]);
      for (int c2 = c + 1; c2 < 8; c2++) dot += uk[k0 + c2] * y[k0 + c2];
      y[k0 + c] -= dot;
    }
  }
}
// `inplace` != nullptr: the caller's matrix (== A.data(), upper triangle filled) is factorised where it lies and is destroyed -- the
// large systems are built for this one solve, copying them first is 1.3 MB of traffic at dimension 401
__attribute__((target("avx2,fma"))) inline void ldlt_solve(const std::vector<double> &A, const std::vector<double> &b, std::vector<double> &x, int n,
                                                           double *inplace = nullptr) {
  const size_t N = (size_t)n;
  static thread_local std::vector<double> Uown, D, y, diag, WT, LT;
  static thread_local std::vector<int> perm;
  if (!inplace) Uown.resize(N * N);
  double *const U = inplace ? inplace : Uown.data();
  D.resize(N); y.resize(N); diag.resize(N); perm.resize(N);
  WT.resize((size_t)LDLT_NB * N); LT.resize((size_t)LDLT_NB * N);
  if (inplace) {
    for (int j = 0; j < n; j++) diag[j] = U[j * N + j];
  } else {
    for (int j = 0; j < n; j++) {
      const double *src = &A[j * N];  // A is symmetric: row j right of the diagonal == column j below it
      double *dst = &U[j * N];
      for (int i = j + 1; i < n; i++) dst[i] = src[i];
      diag[j] = src[j];
    }
  }
  ldlt_factor(U, n, n, n, D.data(), perm.data(), diag.data(), WT.data(), LT.data());
  for (int i = 0; i < n; i++) y[i] = b[i];
  if (ldlt_have_avx512()) {
    ldlt_substitute_512(U, D.data(), perm.data(), y.data(), n);
    x.assign(y.begin(), y.begin() + n);
    return;
  }
  for (int k = 0; k < n; k++) {  // L z = P b, column-oriented: L(i,k) = U[k][i]
    std::swap(y[k], y[perm[k]]);
    const double yk = y[k];
    const double *uk = &U[k * N];
    for (int i = k + 1; i < n; i++) y[i] -= uk[i] * yk;
  }
  for (int i = 0; i < n; i++) y[i] = (std::fabs(D[i]) > 2.2250738585072014e-308) ? y[i] / D[i] : 0.0;
  for (int k = n - 1; k >= 0; k--) {  // L^T w = z: dot product with four independent vector accumulators
    const double *uk = &U[k * N];
    __m256d a0 = _mm256_setzero_pd(), a1 = _mm256_setzero_pd();
    int i = k + 1;
    for (; i + 8 <= n; i += 8) {
      a0 = _mm256_fmadd_pd(_mm256_loadu_pd(uk + i), _mm256_loadu_pd(&y[i]), a0);
      a1 = _mm256_fmadd_pd(_mm256_loadu_pd(uk + i + 4), _mm256_loadu_pd(&y[i + 4]), a1);
    }
    double t[4];
    _mm256_storeu_pd(t, _mm256_add_pd(a0, a1));
    double dot = (t[0] + t[1]) + (t[2] + t[3]);
    for (; i < n; i++) dot += uk[i] * y[i];
    y[k] -= dot;
    std::swap(y[k], y[perm[k]]);
  }
  x.assign(y.begin(), y.begin() + n);
}

// ---- partial factorisation: K = [A B^T; B C] with the leading m x m block A eliminated once and reused for many right-hand sides
// and many trailing blocks C (the visual-inertial KKT system: everything except the visual block of the border is constant over the
// iterations of one optimize() once the scale is trapped, sos_imu.cpp).  Pivots are taken from the leading block only.
//
// The leading block may carry STRUCTURE (round 5): pivot blocks (per keyframe: its bias / spline states, then the multipliers of its
// constraints) that couple only to a few following blocks and to the border.  blk_end[k] = end of the pivot block of interior index k
// (pivot candidates stay inside the block), env_end[k] = end of the interior columns a row of that block can be nonzero in (a block end,
// the same for every k of a block, monotone in k).  The elimination then never leaves [k, env_end) + the border: per pivot
// (env - k + border)^2 / 2 updates instead of (n - k)^2 / 2 -- at W12 (12 blocks of <= 27 interior unknowns, border 101) a quarter
// of the dense count.  What lies outside the envelope is never read and need not be initialised.  Empty vectors: one block, full envelope.
struct LdltPartial {
  int n = 0, m = 0;
  std::vector<double> U;     // n x n: rows < m hold L (row k = column k of L right of the diagonal, inside its envelope and over the
                             // border); rows >= m the strict upper triangle of the Schur complement C - B A^-1 B^T
  std::vector<double> D;     // m pivots
  std::vector<double> diag;  // n: [m, n) = the diagonal of the Schur complement
  std::vector<int> perm;     // m interchanges (within the pivot blocks)
  std::vector<int> blk_end, env_end;
  int be(int k) const { return blk_end.empty() ? m : blk_end[k]; }
  int ee(int k) const { return env_end.empty() ? m : env_end[k]; }
};
// U[j][i] -= sum_c L(j,c) W(i,c) for R consecutive rows j .. j + R - 1 and the columns i in [max(i0, j + 1), i1): the register tile
// of the dense update (ldlt_rows_512) on a column range
// (a second column range [i2, i3) shares the set-up of the row group: the interior rows update their envelope AND the border)
// (KB = the pivots of the panel, a template parameter: a pivot block of 25 unknowns ends with a panel of one pivot, which costs an
// eighth of a full one this way instead of all of it)
template <int R, int KB>
__attribute__((target("avx512f,fma"))) inline void ldlt_range_rows_512(double *U, const double *WT, const double *LT, size_t N, int j, int i0, int i1,
                                                                       int i2 = 0, int i3 = 0) {
  double *r[R];
  for (int a = 0; a < R; a++) r[a] = U + (size_t)(j + a) * N;
  __m512d l[R][KB];
  for (int c = 0; c < KB; c++)
    for (int a = 0; a < R; a++) l[a][c] = _mm512_set1_pd(LT[c * N + j + a]);
  for (int part = 0; part < 2; part++) {
    const int e1 = part == 0 ? i1 : i3;
    // the first range starts right of the group's FIRST row: the entries between the rows of the group ride in the first vector,
    // whose per-row masks leave out what is not right of that row's diagonal (no scalar prologue)
    for (int i = part == 0 ? std::max(i0, j + 1) : i2; i < e1; i += 8) {
      const __mmask8 m = (e1 - i >= 8) ? (__mmask8)0xff : (__mmask8)((1u << (e1 - i)) - 1u);
      __mmask8 mr[R];
      for (int a = 0; a < R; a++) {
        const int lo = part == 0 ? j + a + 1 - i : 0;  // lanes below lo are at or left of row j + a's diagonal
        mr[a] = lo > 0 ? (__mmask8)(m & ~((1u << std::min(lo, 8)) - 1u)) : m;
      }
      __m512d acc[R];
      for (int a = 0; a < R; a++) acc[a] = _mm512_maskz_loadu_pd(mr[a], r[a] + i);
#pragma GCC unroll 8
      for (int c = 0; c < KB; c++) {
        const __m512d wv = _mm512_maskz_loadu_pd(m, WT + (size_t)c * N + i);
        for (int a = 0; a < R; a++) acc[a] = _mm512_fnmadd_pd(l[a][c], wv, acc[a]);
      }
      for (int a = 0; a < R; a++) _mm512_mask_storeu_pd(r[a] + i, mr[a], acc[a]);
    }
  }
}
template <int KB>
inline void ldlt_range_update_512(double *U, const double *WT, const double *LT, size_t N, int j0, int j1, int i0, int i1, int i2, int i3) {
  int j = j0;
  for (; j + 3 <= j1; j += 3) ldlt_range_rows_512<3, KB>(U, WT, LT, N, j, i0, i1, i2, i3);
  if (j1 - j == 2) ldlt_range_rows_512<2, KB>(U, WT, LT, N, j, i0, i1, i2, i3);
  else if (j1 - j == 1) ldlt_range_rows_512<1, KB>(U, WT, LT, N, j, i0, i1, i2, i3);
}
template <int R>
__attribute__((target("avx2,fma"))) inline void ldlt_range_rows_256(double *U, const double *WT, const double *LT, size_t N, int j, int kb, int i0, int i1) {
  double *r[R];
  for (int a = 0; a < R; a++) r[a] = U + (size_t)(j + a) * N;
  for (int a = 0; a < R; a++)
    for (int b = a + 1; b < R; b++) {
      if (j + b < i0 || j + b >= i1) continue;
      double sv = 0;
      for (int c = 0; c < kb; c++) sv += LT[c * N + j + a] * WT[c * N + j + b];
      r[a][j + b] -= sv;
    }
  const int is = std::max(i0, j + R);
  for (int c0 = 0; c0 < kb; c0 += 4) {
    const int cb = std::min(4, kb - c0);
    __m256d l[R][4];
    const double *w[4];
    for (int c = 0; c < 4; c++) {
      const bool on = c < cb;
      for (int a = 0; a < R; a++) l[a][c] = _mm256_set1_pd(on ? LT[(c0 + c) * N + j + a] : 0.0);
      w[c] = WT + (size_t)(c0 + (on ? c : 0)) * N;
    }
    int i = is;
    for (; i + 4 <= i1; i += 4) {
      __m256d acc[R];
      for (int a = 0; a < R; a++) acc[a] = _mm256_loadu_pd(r[a] + i);
      for (int c = 0; c < 4; c++) {
        const __m256d wv = _mm256_loadu_pd(w[c] + i);
        for (int a = 0; a < R; a++) acc[a] = _mm256_fnmadd_pd(l[a][c], wv, acc[a]);
      }
      for (int a = 0; a < R; a++) _mm256_storeu_pd(r[a] + i, acc[a]);
    }
    for (; i < i1; i++)
      for (int a = 0; a < R; a++) {
        double sv = 0;
        for (int c = 0; c < cb; c++) sv += LT[(c0 + c) * N + j + a] * WT[(c0 + c) * N + i];
        r[a][i] -= sv;
      }
  }
}
// one pivot of a panel over a column range: column k brought up to date with the q earlier pivots of the panel, L = a / d into both
// panel copies and the row of U, the candidate diagonal updated (the three loops of ldlt_factor's 256-bit path, fused)
__attribute__((target("avx2,fma"))) inline void ldlt_pivot_range(double *uk, const double *WT, const double *LT, double *wt, double *lt, double *diag, size_t N,
                                                               int k, int q, int a0, int a1, double dinv, bool zero) {
  __m256d lk[LDLT_NB];
  for (int c = 0; c < q; c++) lk[c] = _mm256_set1_pd(LT[c * N + k]);
  const __m256d dv = _mm256_set1_pd(dinv);
  int i = a0;
  if (!zero)
    for (; i + 4 <= a1; i += 4) {
      __m256d a = _mm256_loadu_pd(uk + i);
      for (int c = 0; c < q; c++) a = _mm256_fnmadd_pd(_mm256_loadu_pd(&WT[c * N + i]), lk[c], a);
      const __m256d l = _mm256_mul_pd(a, dv);
      _mm256_storeu_pd(wt + i, a);
      _mm256_storeu_pd(lt + i, l);
      _mm256_storeu_pd(uk + i, l);
      _mm256_storeu_pd(diag + i, _mm256_fnmadd_pd(a, l, _mm256_loadu_pd(diag + i)));
    }
  for (; i < a1; i++) {
    double a = uk[i];
    for (int c = 0; c < q; c++) a -= WT[c * N + i] * LT[c * N + k];
    if (zero) a = 0.0;  // an exact-zero pivot contributes nothing (Eigen LDLT::solve semantics)
    const double l = a * dinv;
    wt[i] = a; lt[i] = l; uk[i] = l;
    diag[i] -= a * l;
  }
}
__attribute__((target("avx512f,fma"))) inline void ldlt_pivot_range_512(double *uk, const double *WT, const double *LT, double *wt, double *lt, double *diag, size_t N,
                                                                      int k, int q, int a0, int a1, double dinv, bool zero) {
  if (zero) return ldlt_pivot_range(uk, WT, LT, wt, lt, diag, N, k, q, a0, a1, dinv, zero);
  __m512d lk[LDLT_NB];
  for (int c = 0; c < q; c++) lk[c] = _mm512_set1_pd(LT[c * N + k]);
  const __m512d dv = _mm512_set1_pd(dinv);
  for (int i = a0; i < a1; i += 8) {
    const __mmask8 mk = (a1 - i >= 8) ? (__mmask8)0xff : (__mmask8)((1u << (a1 - i)) - 1u);
    __m512d a = _mm512_maskz_loadu_pd(mk, uk + i);
    for (int c = 0; c < q; c++) a = _mm512_fnmadd_pd(_mm512_maskz_loadu_pd(mk, &WT[c * N + i]), lk[c], a);
    const __m512d l = _mm512_mul_pd(a, dv);
    _mm512_mask_storeu_pd(wt + i, mk, a);
    _mm512_mask_storeu_pd(lt + i, mk, l);
    _mm512_mask_storeu_pd(uk + i, mk, l);
    _mm512_mask_storeu_pd(diag + i, mk, _mm512_fnmadd_pd(a, l, _mm512_maskz_loadu_pd(mk, diag + i)));
  }
}
// rows [j0, j1) x columns [max(i0, row + 1), i1) and [i2, i3)
inline void ldlt_range_update(double *U, const double *WT, const double *LT, size_t N, int j0, int j1, int i0, int i1, int kb, bool wide, int i2 = 0, int i3 = 0) {
  int j = j0;
  if (wide) {
    switch (kb) {
      case 1: return ldlt_range_update_512<1>(U, WT, LT, N, j0, j1, i0, i1, i2, i3);
      case 2: return ldlt_range_update_512<2>(U, WT, LT, N, j0, j1, i0, i1, i2, i3);
      case 3: return ldlt_range_update_512<3>(U, WT, LT, N, j0, j1, i0, i1, i2, i3);
      case 4: return ldlt_range_update_512<4>(U, WT, LT, N, j0, j1, i0, i1, i2, i3);
      case 5: return ldlt_range_update_512<5>(U, WT, LT, N, j0, j1, i0, i1, i2, i3);
      case 6: return ldlt_range_update_512<6>(U, WT, LT, N, j0, j1, i0, i1, i2, i3);
      case 7: return ldlt_range_update_512<7>(U, WT, LT, N, j0, j1, i0, i1, i2, i3);
      default: return ldlt_range_update_512<8>(U, WT, LT, N, j0, j1, i0, i1, i2, i3);
    }
  } else {
    if (i3 > i2) ldlt_range_update(U, WT, LT, N, j0, j1, i2, i3, kb, false);
    for (; j + 3 <= j1; j += 3) ldlt_range_rows_256<3>(U, WT, LT, N, j, kb, i0, i1);
    if (j1 - j == 2) ldlt_range_rows_256<2>(U, WT, LT, N, j, kb, i0, i1);
    else if (j1 - j == 1) ldlt_range_rows_256<1>(U, WT, LT, N, j, kb, i0, i1);
  }
}
// The trailing block's share of the elimination, applied ONCE behind the last pivot instead of panel by panel:
//   U[m + j][m + i] -= sum_k W[k][j] * G[k][i]   (j < i),   G[k] = row k of L over the border columns, W[k] = the same row times D[k].
// Nothing reads the trailing block before the factorisation ends, and in this form the sum over the pivots runs in registers (6 x 16
// tile: 8 loads per 12 multiply-adds).  Both operands are PACKED while the pivots are produced -- G tile-major ([column tile of 16][pivot]
// [16], zero-padded), W row-major with a short stride -- and the pivots go in blocks of LDLT_BK so that a column tile's G block stays in
// L1 over all row groups: read from the rows of U (stride n doubles = a new page per pivot, no prefetcher follows that) the same loop
// ran at 18 % of the FMA rate, one L2 latency per pivot and tile (131 of the 290 us of a W12 factorisation; 40 us now).
#define LDLT_BK 128
struct LdltBorderPack {
  // column tiles RIGHT-aligned: tile t holds the border columns [16 t - off, 16 t - off + 16), off = 16 T - nb, so that the ragged tile is
  // the first one -- which only the first row group needs -- instead of the last one, which every row group needs
  int m = 0, nb = 0, T = 0, ldw = 0, off = 0;
  std::vector<double> G, W;
  void shape(int m_, int nb_) {
    m = m_; nb = nb_; T = (nb + 15) / 16; off = 16 * T - nb;
    ldw = (nb + 5) / 6 * 6 + 2;  // row groups of 6 read W[k][j .. j + 5]
    G.resize((size_t)T * m * 16);
    W.resize((size_t)m * ldw);
  }
  double g(int k, int i) const { return G[((size_t)((i + off) >> 4) * m + k) * 16 + ((i + off) & 15)]; }
  void put(int k, const double *l, const double *w) {  // pivot k: its L and L D over the border columns
    for (int z = 0; z < off; z++) G[(size_t)k * 16 + z] = 0.0;
    for (int i = 0; i < nb; i++) G[((size_t)((i + off) >> 4) * m + k) * 16 + ((i + off) & 15)] = l[i];
    double *dw = &W[(size_t)k * ldw];
    std::memcpy(dw, w, sizeof(double) * nb);
    for (int z = nb; z < ldw; z++) dw[z] = 0.0;
  }
  __attribute__((target("avx512f"))) void put_512(int k, const double *l, const double *w) {
    double *dw = &W[(size_t)k * ldw];
    {  // first tile: `off` zeros, then the first 16 - off columns (masked loads do not touch what lies in front of l)
      const unsigned keep = 0xffffu & ~((1u << off) - 1u);
      double *dst = &G[(size_t)k * 16];
      _mm512_storeu_pd(dst, _mm512_maskz_loadu_pd((__mmask8)(keep & 0xffu), l - off));
      _mm512_storeu_pd(dst + 8, _mm512_maskz_loadu_pd((__mmask8)(keep >> 8), l - off + 8));
    }
    for (int t = 1; t < T; t++) {
      double *dst = &G[((size_t)t * m + k) * 16];
      const double *src = l + 16 * t - off;
      _mm512_storeu_pd(dst, _mm512_loadu_pd(src)); _mm512_storeu_pd(dst + 8, _mm512_loadu_pd(src + 8));
    }
    int i = 0;
    for (; i + 8 <= nb; i += 8) _mm512_storeu_pd(dw + i, _mm512_loadu_pd(w + i));
    if (i < nb) { const __mmask8 mk = (__mmask8)((1u << (nb - i)) - 1u); _mm512_mask_storeu_pd(dw + i, mk, _mm512_maskz_loadu_pd(mk, w + i)); }
    for (int z = nb; z < ldw; z++) dw[z] = 0.0;  // the last row group reads up to five entries past nb
  }
};
__attribute__((target("avx512f,fma"))) inline void ldlt_border_update_512(double *U, const LdltBorderPack &B, size_t N, int m, int n) {
  const int nb = n - m, T = B.T, ldw = B.ldw;
  for (int k0 = 0; k0 < m; k0 += LDLT_BK) {
    const int k1 = std::min(m, k0 + LDLT_BK);
    for (int t = 0; t < T; t++) {
      const double *g0 = &B.G[((size_t)t * m + k0) * 16];
      const int c0 = 16 * t - B.off, cend = c0 + 16;       // the tile's columns (c0 < 0 for the first tile: padding in front)
      for (int j = 0; j + 1 < cend; j += 6) {              // row groups with a column right of their first row
        const double *g = g0, *w = &B.W[(size_t)k0 * ldw + j];
        __m512d acc[6][2];
        for (int a = 0; a < 6; a++) acc[a][0] = acc[a][1] = _mm512_setzero_pd();
        for (int k = k0; k < k1; k++, g += 16, w += ldw) {
          const __m512d ga = _mm512_loadu_pd(g), gb = _mm512_loadu_pd(g + 8);
#pragma GCC unroll 6
          for (int a = 0; a < 6; a++) {
            const __m512d wa = _mm512_set1_pd(w[a]);
            acc[a][0] = _mm512_fmadd_pd(wa, ga, acc[a][0]);
            acc[a][1] = _mm512_fmadd_pd(wa, gb, acc[a][1]);
          }
        }
        for (int a = 0; a < 6 && j + a < nb; a++) {  // only the columns right of the row's diagonal
          const int lo = std::max(j + a + 1, std::max(c0, 0)) - c0;  // local columns [lo, 16)
          if (lo >= 16) continue;
          const unsigned full = 0xffffu & ~((1u << lo) - 1u);
          const __mmask8 s0 = (__mmask8)(full & 0xffu), s1 = (__mmask8)(full >> 8);
          double *row = U + (ptrdiff_t)((size_t)(m + j + a) * N + m) + c0;
          _mm512_mask_storeu_pd(row, s0, _mm512_sub_pd(_mm512_maskz_loadu_pd(s0, row), acc[a][0]));
          _mm512_mask_storeu_pd(row + 8, s1, _mm512_sub_pd(_mm512_maskz_loadu_pd(s1, row + 8), acc[a][1]));
        }
      }
    }
  }
}
inline void ldlt_border_update(double *U, const LdltBorderPack &B, size_t N, int m, int n, bool wide) {
  if (wide) return ldlt_border_update_512(U, B, N, m, n);
  const int nb = n - m;
  static thread_local std::vector<double> gk;
  gk.resize((size_t)16 * B.T);
  for (int k = 0; k < m; k++) {
    for (int t = 0; t < B.T; t++) std::memcpy(&gk[16 * t], &B.G[((size_t)t * m + k) * 16], sizeof(double) * 16);
    const double *w = &B.W[(size_t)k * B.ldw], *g = gk.data() + B.off;
    for (int j = 0; j < nb; j++) {
      const double wj = w[j];
      if (wj == 0.0) continue;
      double *row = U + (size_t)(m + j) * N + m;
      for (int i = j + 1; i < nb; i++) row[i] -= wj * g[i];
    }
  }
}
// F.U holds the upper triangle INCLUDING the diagonal on entry (inside the envelope and over the border)
inline void ldlt_partial_factor(LdltPartial &F) {
  const int n = F.n, m = F.m;
  const size_t N = (size_t)n;
  F.D.assign(m, 0.0);
  F.diag.resize(N);
  F.perm.assign(m, 0);
  static thread_local std::vector<double> WTv, LTv;
  static thread_local LdltBorderPack BP;  // L and L D over the border columns, pivot by pivot, for the trailing block's update at the end
  WTv.resize((size_t)LDLT_NB * N); LTv.resize((size_t)LDLT_NB * N);
  BP.shape(m, n - m);
  double *U = F.U.data(), *diag = F.diag.data(), *WTp = WTv.data(), *LTp = LTv.data();
  for (int j = 0; j < n; j++) diag[j] = U[j * N + j];
  const bool wide = ldlt_have_avx512();
  static const bool TM = getenv("SOS_TIMING_IMU") != nullptr;  // phase split on stderr, with the other [imu_cached] lines
  auto NOW = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; };
  double tp = 0, tu = 0, tbd = 0, t_ = 0;
  for (int k0 = 0; k0 < m;) {
    if (TM) t_ = NOW();
    const int bend = F.be(k0), eend = F.ee(k0);     // pivot block and interior envelope of this panel
    const int kb = std::min(LDLT_NB, bend - k0), k1 = k0 + kb;
    for (int k = k0; k < k1; k++) {
      const int q = k - k0;
      // threshold pivoting inside the pivot block (see ldlt_factor)
      int p = k;
      {
        double best = std::fabs(diag[k]);
        int pb = k;
        for (int i = k + 1; i < bend; i++)
          if (std::fabs(diag[i]) > best) { best = std::fabs(diag[i]); pb = i; }
        if (!(std::fabs(diag[k]) >= 0.1 * best)) p = pb;
      }
      F.perm[k] = p;
      if (p != k) {  // symmetric interchange k <-> p of the not yet eliminated part (both inside one block: same envelope)
        for (int j = k + 1; j < p; j++) std::swap(U[k * N + j], U[j * N + p]);
        double *rk = &U[k * N], *rp = &U[p * N];
        for (int i = p + 1; i < eend; i++) std::swap(rk[i], rp[i]);
        for (int i = m; i < n; i++) std::swap(rk[i], rp[i]);
        std::swap(diag[k], diag[p]);
        for (int c = 0; c < q; c++) { std::swap(WTp[c * N + k], WTp[c * N + p]); std::swap(LTp[c * N + k], LTp[c * N + p]); }
      }
      const double d = diag[k];
      F.D[k] = d;
      double *wt = &WTp[(size_t)q * N], *lt = &LTp[(size_t)q * N], *uk = &U[k * N];
      const bool zero = !(std::fabs(d) > 2.2250738585072014e-308);
      const double dinv = zero ? 0.0 : 1.0 / d;
      // column k brought up to date with the q earlier pivots of the panel, L = a / d, candidate diagonal -- over its two ranges
      for (int part = 0; part < 2; part++)
        if (wide) ldlt_pivot_range_512(uk, WTp, LTp, wt, lt, diag, N, k, q, part == 0 ? k + 1 : m, part == 0 ? eend : n, dinv, zero);
        else ldlt_pivot_range(uk, WTp, LTp, wt, lt, diag, N, k, q, part == 0 ? k + 1 : m, part == 0 ? eend : n, dinv, zero);
      if (n > m) { if (wide) BP.put_512(k, lt + m, wt + m); else BP.put(k, lt + m, wt + m); }
    }
    // trailing update: (interior rest of the envelope) x (itself + border); border x border once, behind the last pivot
    if (TM) { const double t2 = NOW(); tp += t2 - t_; t_ = t2; }
    ldlt_range_update(U, WTp, LTp, N, k1, eend, k1, eend, kb, wide, m, n);
    if (TM) { const double t2 = NOW(); tu += t2 - t_; t_ = t2; }
    k0 = k1;
  }
  if (TM) t_ = NOW();
  if (n > m) ldlt_border_update(U, BP, N, m, n, wide);
  if (TM) { tbd = NOW() - t_; fprintf(stderr, "[ldlt_partial] pivots %.0f us, range update %.0f us, border update %.0f us\n", tp, tu, tbd); }
  // the factorisation leaves the finished columns of L in the row order they were computed in; bring them to the final order (the
  // later interchanges applied to the earlier columns whose envelope holds them), so that the substitutions can apply all
  // interchanges to the vector first
  for (int k = 0; k < m; k++) {
    const int p = F.perm[k];
    if (p != k)
      for (int c = 0; c < k; c++)
        if (F.ee(c) > k) std::swap(U[c * N + k], U[c * N + p]);
  }
}
// y (n): right-hand side in, [L^-1 P a ; c - B A^-1 a] out (the leading part still to be divided by D: done by the backward pass).
// L is in its final row order (ldlt_partial_factor), so the interchanges are applied to y up front and the pass is a plain
// triangular solve over each row's envelope and the border.
__attribute__((target("avx2,fma"))) inline void ldlt_partial_forward(const LdltPartial &F, double *__restrict y) {
  const int n = F.n, m = F.m;
  const size_t N = (size_t)n;
  for (int k = 0; k < m; k++) std::swap(y[k], y[F.perm[k]]);
  for (int k = 0; k < m; k++) {
    const double yk = y[k];
    if (yk == 0.0) continue;
    const double *__restrict uk = &F.U[k * N];
    const __m256d v = _mm256_set1_pd(yk);
    for (int part = 0; part < 2; part++) {
      int i = part == 0 ? k + 1 : m;
      const int a1 = part == 0 ? F.ee(k) : n;
      for (; i + 4 <= a1; i += 4) _mm256_storeu_pd(y + i, _mm256_fnmadd_pd(_mm256_loadu_pd(uk + i), v, _mm256_loadu_pd(y + i)));
      for (; i < a1; i++) y[i] -= uk[i] * yk;
    }
  }
}
// y (n): leading part as ldlt_partial_forward left it, trailing part = the solution of the trailing block; out: the whole solution
__attribute__((target("avx2,fma"))) inline void ldlt_partial_backward(const LdltPartial &F, double *__restrict y) {
  const int n = F.n, m = F.m;
  const size_t N = (size_t)n;
  for (int i = 0; i < m; i++) y[i] = (std::fabs(F.D[i]) > 2.2250738585072014e-308) ? y[i] / F.D[i] : 0.0;
  for (int k = m - 1; k >= 0; k--) {
    const double *__restrict uk = &F.U[k * N];
    __m256d a0 = _mm256_setzero_pd(), a1 = a0;
    double dot = 0;
    for (int part = 0; part < 2; part++) {
      int i = part == 0 ? k + 1 : m;
      const int e1 = part == 0 ? F.ee(k) : n;
      for (; i + 8 <= e1; i += 8) {
        a0 = _mm256_fmadd_pd(_mm256_loadu_pd(uk + i), _mm256_loadu_pd(y + i), a0);
        a1 = _mm256_fmadd_pd(_mm256_loadu_pd(uk + i + 4), _mm256_loadu_pd(y + i + 4), a1);
      }
      for (; i < e1; i++) dot += uk[i] * y[i];
    }
    double t[4];
    _mm256_storeu_pd(t, _mm256_add_pd(a0, a1));
    y[k] -= ((t[0] + t[1]) + (t[2] + t[3])) + dot;
  }
  for (int q = m - 1; q >= 0; q--) std::swap(y[q], y[F.perm[q]]);
}

// dense inverse (Gauss-Jordan, partial pivoting) in place of Eigen `.inverse()` (OB/EnergyFunctional.cpp:841)
inline bool mat_inverse(const std::vector<double> &A, std::vector<double> &Ai, int n) {
  const size_t N = (size_t)n;
  std::vector<double> M(N * 2 * N, 0.0);
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < n; j++) M[i * 2 * N + j] = A[i * N + j];
    M[i * 2 * N + N + i] = 1.0;
  }
  for (int k = 0; k < n; k++) {
    int p = k;
    for (int i = k + 1; i < n; i++)
      if (std::fabs(M[i * 2 * N + k]) > std::fabs(M[p * 2 * N + k])) p = i;
    if (p != k)
      for (int j = 0; j < 2 * n; j++) std::swap(M[k * 2 * N + j], M[p * 2 * N + j]);
    const double d = M[k * 2 * N + k];
    if (d == 0.0) return false;
    for (int j = 0; j < 2 * n; j++) M[k * 2 * N + j] /= d;
    for (int i = 0; i < n; i++)
      if (i != k) {
        const double f = M[i * 2 * N + k];
        if (f != 0.0)
          for (int j = 0; j < 2 * n; j++) M[i * 2 * N + j] -= f * M[k * 2 * N + j];
      }
  }
  Ai.assign(N * N, 0.0);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) Ai[i * N + j] = M[i * 2 * N + N + j];
  return true;
}

}  // namespace sos
