// sos_imu.cpp -- IMU / spline factor assembly on the backend boundary, host side, fp64:
//   FrameHessian::getImuHi + spline accessors   FS/HessianBlocks.cpp:178-225, FS/HessianBlocks.h:352-412
//   EnergyFunctional::getImuHessian{,CurrentFrame}, expandHbtoFitImu   OB/EnergyFunctional.cpp:256-494
//   the IMU branch of EnergyFunctional::solveSystemF                   OB/EnergyFunctional.cpp:1053-1171
// This is the block the reference runs between the accumulation (device, sos_ba_gn_accumulate) and the solve; it works
// on the stitched H / b the device hands over and on per-keyframe IMU records the caller keeps (sosf_imu_frame).
#include <cmath>
#include <cstring>
#include <vector>

#include "../../../include/sos_slam_host.h"
#include "sos_math.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
static inline double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
namespace {
using sos::SE3;
constexpr int CP = 4;
constexpr double kScale = 200.0, kSlRot = 100.0, kSqTrans = 1000.0, kSqRot = 1000.0, kScTrans = 1000.0, kScRot = 1000.0, kBa = 100.0,
                 kBg = 1.0, kXiRot = 1.0, kXiTrans = 0.5;  // FS/HessianBlocks.h:53-79

struct V3 {
  double v[3];
  double &operator[](int i) { return v[i]; }
  double operator[](int i) const { return v[i]; }
};
struct M3 {
  double m[9];
  double &operator()(int r, int c) { return m[3 * r + c]; }
  double operator()(int r, int c) const { return m[3 * r + c]; }
  M3 T() const {
    M3 o;
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) o(r, c) = (*this)(c, r);
    return o;
  }
  M3 operator*(const M3 &b) const {
    M3 o;
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) o(r, c) = (*this)(r, 0) * b(0, c) + (*this)(r, 1) * b(1, c) + (*this)(r, 2) * b(2, c);
    return o;
  }
  V3 operator*(const V3 &x) const {
    V3 o;
    for (int r = 0; r < 3; r++) o[r] = (*this)(r, 0) * x[0] + (*this)(r, 1) * x[1] + (*this)(r, 2) * x[2];
    return o;
  }
  static M3 from(const double *p) {
    M3 o;
    std::memcpy(o.m, p, sizeof(o.m));
    return o;
  }
  static M3 hat(const V3 &w) { return M3{{0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0}}; }
};
M3 so3_exp(const V3 &w) {  // Sophus SO3::exp through the unit quaternion, as SE3::exp does
  const double a[6] = {0, 0, 0, w[0], w[1], w[2]};
  return M3::from(SE3::exp(a).R);
}
V3 so3_log(const M3 &R) {  // Sophus SO3::log (so3.hpp:491-526) on Eigen's matrix -> quaternion conversion
  double q[4];             // w x y z
  const double tr = R(0, 0) + R(1, 1) + R(2, 2);
  if (tr > 0) {
    double s = std::sqrt(tr + 1.0);
    q[0] = 0.5 * s;
    s = 0.5 / s;
    q[1] = (R(2, 1) - R(1, 2)) * s;
    q[2] = (R(0, 2) - R(2, 0)) * s;
    q[3] = (R(1, 0) - R(0, 1)) * s;
  } else {
    int i = 0;
    if (R(1, 1) > R(0, 0)) i = 1;
    if (R(2, 2) > R(i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double s = std::sqrt(R(i, i) - R(j, j) - R(k, k) + 1.0);
    q[1 + i] = 0.5 * s;
    s = 0.5 / s;
    q[0] = (R(k, j) - R(j, k)) * s;
    q[1 + j] = (R(j, i) + R(i, j)) * s;
    q[1 + k] = (R(k, i) + R(i, k)) * s;
  }
  const double sq = q[1] * q[1] + q[2] * q[2] + q[3] * q[3], nrm = std::sqrt(sq), w = q[0];
  double f;
  if (nrm < 1e-10) f = 2.0 / w - 2.0 * sq / (w * w * w);
  else if (std::fabs(w) < 1e-10) f = (w > 0 ? M_PI : -M_PI) / nrm;
  else f = 2.0 * std::atan(nrm / w) / nrm;
  return V3{{f * q[1], f * q[2], f * q[3]}};
}

// one keyframe's IMU state as the reference's Eigen::Ref views see it
struct ImuView {
  const sosf_imu_frame &f;
  double sc[21];
  explicit ImuView(const sosf_imu_frame &fr) : f(fr) {
    const double k[7] = {kBa, kBg, kSlRot, kSqTrans, kSqRot, kScTrans, kScRot};
    for (int s = 0; s < 7; s++)
      for (int i = 0; i < 3; i++) sc[3 * s + i] = k[s] * f.state_imu[3 * s + i];
  }
  double bias(int i) const { return sc[i]; }
  V3 acc(double t, bool zero) const {  // getSplineAcc
    V3 a;
    for (int i = 0; i < 3; i++)
      a[i] = zero ? 2 * kSqTrans * f.state_imu_zero[9 + i] + 6 * t * kScTrans * f.state_imu_zero[15 + i] : 2 * sc[9 + i] + 6 * t * sc[15 + i];
    return a;
  }
  V3 gyro(double t) const {  // getSplineGryo
    V3 g;
    for (int i = 0; i < 3; i++) g[i] = sc[6 + i] + (2 * t * sc[12 + i] + 3 * t * t * sc[18 + i]);
    return g;
  }
  M3 R_c_t(double t, bool zero) const {  // getSplineR_c_t
    const double t2 = t * t;
    V3 w;
    for (int i = 0; i < 3; i++)
      w[i] = zero ? t * kSlRot * f.state_imu_zero[6 + i] + t2 * kSqRot * f.state_imu_zero[12 + i] + t * t2 * kScRot * f.state_imu_zero[18 + i]
                  : t * sc[6 + i] + (t2 * sc[12 + i] + t * t2 * sc[18 + i]);
    return so3_exp(w);
  }
};

struct HiOut {
  double JsTW[6], JfTW[29 * 6], Hss, Hff[29 * 29], Hfs[29];
};
// The products of getImuHi (J^T W, J^T W J, J^T W J_s for the 6 x 29 frame Jacobian of one sample) with the six rows of J held in
// 24 zmm registers: 29 x 29 x 6 multiply-adds per sample are what the assembly of a visual-inertial window spends its time on (0.8 us
// of scalar code per sample, 8 - 50 samples per keyframe).  Sums over k ascending as the scalar statement, fused multiply-adds: equal
// to rounding, not bit for bit (the kept J^T W of a sample and its Hessian come from the same call either way).
__attribute__((target("avx512f,fma"))) void hi_products_512(const double *Jf, const double *Js, const double *W, HiOut &o) {
  const __mmask8 tail = 0x1f;  // 29 = 3 x 8 + 5
  __m512d Jv[6][4];
  for (int k = 0; k < 6; k++) {
    for (int v = 0; v < 3; v++) Jv[k][v] = _mm512_loadu_pd(Jf + 29 * k + 8 * v);
    Jv[k][3] = _mm512_maskz_loadu_pd(tail, Jf + 29 * k + 24);
  }
  alignas(64) double T[6][32];  // T[c][r] = (J^T W)(r, c)
  for (int c = 0; c < 6; c++) {
    __m512d t[4] = {_mm512_setzero_pd(), _mm512_setzero_pd(), _mm512_setzero_pd(), _mm512_setzero_pd()};
#pragma GCC unroll 6
    for (int k = 0; k < 6; k++) {
      const __m512d w = _mm512_set1_pd(W[6 * k + c]);
      for (int v = 0; v < 4; v++) t[v] = _mm512_fmadd_pd(w, Jv[k][v], t[v]);
    }
    for (int v = 0; v < 4; v++) _mm512_store_pd(&T[c][8 * v], t[v]);
  }
  for (int r = 0; r < 29; r++)
    for (int c = 0; c < 6; c++) o.JfTW[6 * r + c] = T[c][r];
  for (int r = 0; r < 29; r++) {
    __m512d h[4] = {_mm512_setzero_pd(), _mm512_setzero_pd(), _mm512_setzero_pd(), _mm512_setzero_pd()};
#pragma GCC unroll 6
    for (int k = 0; k < 6; k++) {
      const __m512d t = _mm512_set1_pd(T[k][r]);
      for (int v = 0; v < 4; v++) h[v] = _mm512_fmadd_pd(t, Jv[k][v], h[v]);
    }
    double *dst = &o.Hff[29 * r];
    for (int v = 0; v < 3; v++) _mm512_storeu_pd(dst + 8 * v, h[v]);
    _mm512_mask_storeu_pd(dst + 24, tail, h[3]);
  }
  {
    __m512d sv[4] = {_mm512_setzero_pd(), _mm512_setzero_pd(), _mm512_setzero_pd(), _mm512_setzero_pd()};
    for (int k = 0; k < 6; k++) {
      const __m512d js = _mm512_set1_pd(Js[k]);
      for (int v = 0; v < 4; v++) sv[v] = _mm512_fmadd_pd(js, _mm512_load_pd(&T[k][8 * v]), sv[v]);
    }
    for (int v = 0; v < 3; v++) _mm512_storeu_pd(o.Hfs + 8 * v, sv[v]);
    _mm512_mask_storeu_pd(o.Hfs + 24, tail, sv[3]);
  }
}
void get_Hi(const sosf_imu_settings &S, const sosf_imu_calib &C, const sosf_imu_frame &fr, double tt, HiOut &o) {
  const ImuView f(fr);
  const bool trapped = C.scale_trapped != 0;
  const double tt2 = tt * tt, scale_scaled = (trapped ? C.scale_zero : C.scale) * kScale;
  const V3 a = f.acc(tt, trapped);
  V3 acc_w;
  for (int i = 0; i < 3; i++) acc_w[i] = scale_scaled * a[i] + S.gravity[i];
  const M3 Ric = M3::from(S.rot_imu_cam);
  const M3 rot_t_w = f.R_c_t(tt, trapped).T() * M3::from(fr.evalPT_R).T();
  const M3 rot_i_w = Ric * rot_t_w;
  const M3 R_acc_t_hat = Ric * M3::hat(rot_t_w * acc_w);
  double Js[6] = {0, 0, 0, 0, 0, 0};
  double Jf[6 * 29];
  std::memset(Jf, 0, sizeof(Jf));
  auto J = [&](int r, int c) -> double & { return Jf[29 * r + c]; };
  const V3 ra = rot_i_w * a;
  for (int i = 0; i < 3; i++) Js[i] = kScale * ra[i];
  if (trapped) {  // the pose columns take part only once the scale is trapped
    const M3 d = rot_i_w * M3::hat(acc_w);
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) J(r, 3 + c) = kXiRot * d(r, c);
  }
  for (int r = 0; r < 3; r++) {
    J(r, 8 + r) = kBa;
    J(3 + r, 11 + r) = kBg;
    for (int c = 0; c < 3; c++) {
      J(r, 14 + c) = kSlRot * R_acc_t_hat(r, c) * tt;
      J(r, 20 + c) = kSqRot * R_acc_t_hat(r, c) * tt2;
      J(r, 26 + c) = kScRot * R_acc_t_hat(r, c) * tt * tt2;
      J(r, 17 + c) = kSqTrans * rot_i_w(r, c) * 2 * scale_scaled;
      J(r, 23 + c) = kScTrans * rot_i_w(r, c) * 6 * tt * scale_scaled;
      J(3 + r, 14 + c) = kSlRot * Ric(r, c);
      J(3 + r, 20 + c) = kSqRot * Ric(r, c) * 2 * tt;
      J(3 + r, 26 + c) = kScRot * Ric(r, c) * 3 * tt2;
    }
  }
  for (int c = 0; c < 6; c++) {
    double s = 0;
    for (int k = 0; k < 6; k++) s += Js[k] * S.weight_imu[6 * k + c];
    o.JsTW[c] = s;
  }
  o.Hss = 0;
  for (int k = 0; k < 6; k++) o.Hss += o.JsTW[k] * Js[k];
  if (sos::ldlt_have_avx512()) return hi_products_512(Jf, Js, S.weight_imu, o);
  for (int r = 0; r < 29; r++)
    for (int c = 0; c < 6; c++) {
      double s = 0;
      for (int k = 0; k < 6; k++) s += J(k, r) * S.weight_imu[6 * k + c];
      o.JfTW[6 * r + c] = s;
    }
  for (int r = 0; r < 29; r++) {
    for (int c = 0; c < 29; c++) {
      double s = 0;
      for (int k = 0; k < 6; k++) s += o.JfTW[6 * r + k] * J(k, c);
      o.Hff[29 * r + c] = s;
    }
    double s = 0;
    for (int k = 0; k < 6; k++) s += o.JfTW[6 * r + k] * Js[k];
    o.Hfs[r] = s;
  }
}

// dense row-major matrix with the handful of block operations the assembly needs
struct Dense {
  int rows, cols;
  std::vector<double> a;
  Dense(int r, int c) : rows(r), cols(c), a((size_t)r * c, 0.0) {}
  double &operator()(int r, int c) { return a[(size_t)r * cols + c]; }
  double operator()(int r, int c) const { return a[(size_t)r * cols + c]; }
};

// the constraint rows, contiguous (rows x dim) and kept between uses: a vector per row was 48 allocations + zero fills per assembly at
// W12 and a copy to get them side by side
struct RowStore {
  int dim = 0, rows = 0;
  std::vector<double> a;
  size_t size() const { return (size_t)rows; }
  double *add_row() {
    if (a.size() < (size_t)(rows + 1) * dim) a.resize((size_t)(rows + 8) * dim);
    double *p = &a[(size_t)rows * dim];
    std::memset(p, 0, sizeof(double) * dim);
    rows++;
    return p;
  }
  double *operator[](size_t k) { return &a[k * dim]; }
  const double *operator[](size_t k) const { return &a[k * dim]; }
  void clear() { rows = 0; }
};
struct Assembly {
  Dense H;
  std::vector<double> b;
  RowStore Jrows;  // constraint rows
  std::vector<double> r;
  std::vector<int> spline_valid;
  Assembly(int dim, int n) : H(dim, dim), b(dim, 0.0), spline_valid(n, 0) { Jrows.dim = dim; }
};

// per-sample J^T W of the first-estimate case (scale trapped), kept by the cached solve: the Jacobians of getImuHi are then taken at
// state_imu_zero / scale_zero / evalPT and do not move with the iterations
struct HiStore {
  std::vector<double> JsTW, JfTW;  // 6 / 29 x 6 per sample, samples of frame 1, 2, ... in order (valid splines only)
};
void add_frame(const sosf_imu_settings &S, const sosf_imu_calib &C, int n, const sosf_imu_frame *F, int fi, Assembly &A, HiStore *store = nullptr) {
  const sosf_imu_frame &cur = F[fi], &prv = F[fi - 1];
  const ImuView vc(cur), vp(prv);
  const double tpf = prv.timestamp - cur.timestamp, tpf2 = tpf * tpf;
  const int ci = CP + 1 + 29 * fi, pi = CP + 1 + 29 * (fi - 1);
  // bias random walk between consecutive keyframes
  double W[36], rb[6], tb[6];
  for (int k = 0; k < 36; k++) W[k] = S.weight_imu_bias[k] / -tpf;
  for (int k = 0; k < 6; k++) rb[k] = vc.bias(k) - vp.bias(k);
  for (int r = 0; r < 6; r++) {
    double s = 0;
    for (int k = 0; k < 6; k++) s += W[6 * r + k] * rb[k];
    tb[r] = s * (r < 3 ? kBa : kBg);
  }
  for (int r = 0; r < 6; r++)
    for (int c = 0; c < 6; c++) {
      double w = W[6 * r + c];
      if (r < 3 && c < 3) w *= (kBa * kBa);
      if (r >= 3 && c >= 3) w *= (kBg * kBg);
      A.H(pi + 8 + r, pi + 8 + c) += w;
      A.H(ci + 8 + r, ci + 8 + c) += w;
      A.H(pi + 8 + r, ci + 8 + c) += -w;
      A.H(ci + 8 + r, pi + 8 + c) += -w;
    }
  for (int r = 0; r < 6; r++) {
    A.b[pi + 8 + r] += -tb[r];
    A.b[ci + 8 + r] += tb[r];
  }
  const bool valid = cur.trackingRefIsPrev && (-tpf < S.maxImuInterval);
  A.spline_valid[fi] = valid ? 1 : 0;
  if (!valid) return;
  const bool vel_valid = fi < n - 1;
  const int rows = vel_valid ? 6 : 3, row0 = (int)A.Jrows.size();
  for (int k = 0; k < rows; k++) {
    A.Jrows.add_row();
    A.r.push_back(0.0);
  }
  // spline rotation against the relative rotation of the two keyframes
  const M3 Rc = M3::from(cur.camToWorld), Rp = M3::from(prv.camToWorld);
  const M3 meas = Rc.T() * Rp, pred = vc.R_c_t(tpf, false);
  const V3 rr = so3_log(meas.T() * pred);
  const M3 Rpe = M3::from(prv.evalPT_R).T();
  for (int r = 0; r < 3; r++) {
    A.r[row0 + r] = rr[r];
    double *J = A.Jrows[row0 + r];
    for (int c = 0; c < 3; c++) {
      J[pi + 3 + c] = -kXiRot * Rpe(r, c);
      J[ci + 3 + c] = kXiRot * Rpe(r, c);
    }
    J[ci + 14 + r] = kSlRot * tpf;
    J[ci + 20 + r] = kSqRot * tpf2;
    J[ci + 26 + r] = kScRot * tpf * tpf2;
  }
  if (vel_valid) {  // velocity continuity with the following keyframe
    const sosf_imu_frame &nxt = F[fi + 1];
    const ImuView vn(nxt);
    const double tnf = cur.timestamp - nxt.timestamp;
    if (nxt.trackingRefIsPrev && (-tnf < S.maxImuInterval)) {
      const int ni = CP + 1 + 29 * (fi + 1);
      const double tnf2 = tnf * tnf;
      for (int r = 0; r < 3; r++) {
        const double dso = (1 / tpf) * (prv.camToWorld[9 + r] - cur.camToWorld[9 + r]) - (1 / tnf) * (cur.camToWorld[9 + r] - nxt.camToWorld[9 + r]);
        const double imu = tpf * vc.sc[9 + r] + tpf2 * vc.sc[15 + r] + tnf * vn.sc[9 + r] + 2 * tnf2 * vn.sc[15 + r];
        A.r[row0 + 3 + r] = imu - dso;
        double *J = A.Jrows[row0 + 3 + r];
        J[pi + r] = -kXiTrans / tpf;
        J[ci + r] = kXiTrans * (1 / tpf + 1 / tnf);
        J[ni + r] = -kXiTrans / tnf;
        J[ci + 17 + r] = kSqTrans * tpf;
        J[ci + 23 + r] = kScTrans * tpf2;
        J[ni + 17 + r] = kSqTrans * tnf;
        J[ni + 23 + r] = kScTrans * 2 * tnf2;
      }
    }
  }
  // IMU samples against the spline
  auto add_H = [&](double Hss, const double *Hff, const double *Hfs) {
    A.H(CP, CP) += Hss;
    for (int r = 0; r < 29; r++) {
      A.H(ci + r, CP) += Hfs[r];
      A.H(CP, ci + r) += Hfs[r];
      for (int c = 0; c < 29; c++) A.H(ci + r, ci + c) += Hff[29 * r + c];
    }
  };
  HiOut hi;
  // first-estimate Jacobians (scale trapped): the per-sample Hessians are summed (setImuStateZero) and added whole -- the sums are
  // formed in the sample loop below, from the same get_Hi call that serves the right-hand side
  double sHss = 0, sHff[29 * 29], sHfs[29];
  if (C.scale_trapped) {
    std::memset(sHff, 0, sizeof(sHff));
    std::memset(sHfs, 0, sizeof(sHfs));
  }
  const M3 Ric = M3::from(S.rot_imu_cam), Rwc = Rc.T();
  const double scale_scaled = C.scale * kScale;
  for (int j = 0; j < cur.n_imu; j++) {
    const double tt = cur.imu[7 * j] - cur.timestamp;
    const V3 a = vc.acc(tt, false);
    V3 aw;
    for (int i = 0; i < 3; i++) aw[i] = scale_scaled * a[i] + S.gravity[i];
    const V3 pa = ((Ric * vc.R_c_t(tt, false).T()) * Rwc) * aw, pg = Ric * vc.gyro(tt);
    double res[6];
    for (int i = 0; i < 3; i++) {
      res[i] = (pa[i] + vc.bias(i)) - cur.imu[7 * j + 1 + i];
      res[3 + i] = (pg[i] + vc.bias(3 + i)) - cur.imu[7 * j + 4 + i];
    }
    get_Hi(S, C, cur, tt, hi);
    if (store) {
      store->JsTW.insert(store->JsTW.end(), hi.JsTW, hi.JsTW + 6);
      store->JfTW.insert(store->JfTW.end(), hi.JfTW, hi.JfTW + 29 * 6);
    }
    if (!C.scale_trapped) add_H(hi.Hss, hi.Hff, hi.Hfs);
    else {
      sHss += hi.Hss;
      for (int k = 0; k < 29 * 29; k++) sHff[k] += hi.Hff[k];
      for (int k = 0; k < 29; k++) sHfs[k] += hi.Hfs[k];
    }
    double s = 0;
    for (int k = 0; k < 6; k++) s += hi.JsTW[k] * res[k];
    A.b[CP] += s;
    for (int r = 0; r < 29; r++) {
      double t = 0;
      for (int k = 0; k < 6; k++) t += hi.JfTW[6 * r + k] * res[k];
      A.b[ci + r] += t;
    }
  }
  if (C.scale_trapped && cur.n_imu > 0) add_H(sHss, sHff, sHfs);
}

Assembly assemble(const sosf_imu_settings &S, const sosf_imu_calib &C, int n, const sosf_imu_frame *F) {
  Assembly A(SOSF_IMU_DIM(n), n);
  for (int i = 1; i < n; i++) add_frame(S, C, n, F, i, A);
  return A;
}

// the blocks of the persistent H_imu holder that add_frame can touch, back to zero
void clear_imu_blocks(Assembly &A, int n) {
  const int dimI = A.H.rows;
  double *Hh = A.H.a.data();
  std::memset(Hh + (size_t)CP * dimI, 0, sizeof(double) * dimI);
  for (int i = 0; i < n; i++) {
    const int bi = CP + 1 + 29 * i;
    for (int r = 0; r < 29; r++) {
      double *row = Hh + (size_t)(bi + r) * dimI;
      row[CP] = 0.0;
      std::memset(row + bi, 0, sizeof(double) * 29);
      if (r >= 8 && r < 14) {
        if (i > 0) std::memset(row + bi - 29 + 8, 0, sizeof(double) * 6);
        if (i < n - 1) std::memset(row + bi + 29 + 8, 0, sizeof(double) * 6);
      }
    }
  }
}

void expand(int n, const double *H, const double *b, Dense &He, std::vector<double> &be) {
  const int d0 = CP + 8 * n;
  auto src = [&](int r, int c) { return H[(size_t)r * d0 + c]; };
  auto map = [&](int k) { return k < CP ? k : CP + 1 + 29 * ((k - CP) / 8) + (k - CP) % 8; };  // dso index -> expanded index
  for (int r = 0; r < d0; r++) {
    be[map(r)] += b[r];
    for (int c = 0; c < d0; c++) He(map(r), map(c)) += src(r, c);
  }
}
}  // namespace

extern "C" int sosf_imu_get_Hi(const sosf_imu_settings *S, const sosf_imu_calib *C, const sosf_imu_frame *f, double tt, double *JsTW,
                               double *JfTW, double *Hss, double *Hff, double *Hfs) {
  if (!S || !C || !f || !JsTW || !JfTW || !Hss || !Hff || !Hfs) return SOS_ERR_ARG;
  HiOut o;
  get_Hi(*S, *C, *f, tt, o);
  std::memcpy(JsTW, o.JsTW, sizeof(o.JsTW));
  std::memcpy(JfTW, o.JfTW, sizeof(o.JfTW));
  *Hss = o.Hss;
  std::memcpy(Hff, o.Hff, sizeof(o.Hff));
  std::memcpy(Hfs, o.Hfs, sizeof(o.Hfs));
  return SOS_OK;
}

extern "C" int sosf_imu_hessian(const sosf_imu_settings *S, const sosf_imu_calib *C, int n, const sosf_imu_frame *frames, double *H,
                                double *b, double *J_cst, double *r_cst, int32_t *n_cst, int32_t *spline_valid) {
  if (!S || !C || n < 1 || !frames || !H || !b || !J_cst || !r_cst || !n_cst || !spline_valid) return SOS_ERR_ARG;
  const int dim = SOSF_IMU_DIM(n);
  const Assembly A = assemble(*S, *C, n, frames);
  std::memcpy(H, A.H.a.data(), sizeof(double) * (size_t)dim * dim);
  std::memcpy(b, A.b.data(), sizeof(double) * dim);
  for (size_t k = 0; k < A.Jrows.size(); k++) {
    std::memcpy(J_cst + k * dim, A.Jrows[k], sizeof(double) * dim);
    r_cst[k] = A.r[k];
  }
  *n_cst = (int32_t)A.Jrows.size();
  for (int i = 0; i < n; i++) spline_valid[i] = A.spline_valid[i];
  return SOS_OK;
}

extern "C" int sosf_imu_expand(int n, const double *H, const double *b, double *He, double *be) {
  if (n < 1 || !H || !b || !He || !be) return SOS_ERR_ARG;
  const int dim = SOSF_IMU_DIM(n);
  Dense E(dim, dim);
  std::vector<double> e(dim, 0.0);
  expand(n, H, b, E, e);
  std::memcpy(He, E.a.data(), sizeof(double) * (size_t)dim * dim);
  std::memcpy(be, e.data(), sizeof(double) * dim);
  return SOS_OK;
}

// the persistent holder of H_imu (all-zero H between calls), with b / constraints reset
static Assembly &imu_holder(int dimI, int n) {
  static thread_local std::unique_ptr<Assembly> PA;
  if (!PA || PA->H.rows != dimI || (int)PA->spline_valid.size() != n) PA.reset(new Assembly(dimI, n));
  Assembly &A = *PA;
  std::fill(A.b.begin(), A.b.end(), 0.0);
  A.Jrows.clear();
  A.r.clear();
  std::fill(A.spline_valid.begin(), A.spline_valid.end(), 0);
  return A;
}

// The literal form: the whole KKT system built and factorised by every call.
static int imu_solve_dense(const sosf_imu_settings *S, const sosf_imu_calib *C, int n, const sosf_imu_frame *F, const double *H_top,
                           const double *b_top, const double *H_sc, const double *b_sc, const double *HM, const double *bM,
                           const double *delta, double lambda, double *x, double *scale_step, double *step_imu) {
  const int dimI = SOSF_IMU_DIM(n);
  static const bool tmg = getenv("SOS_TIMING_IMU") != nullptr;
  const double tA0 = tmg ? now_us() : 0;
  // H_imu, b_imu, constraints.  H_imu is block-sparse (a 29 x 29 block per keyframe, the 6 x 6 bias blocks between neighbours, the
  // scale row / column); its dense dimI x dimI holder is kept between calls and cleared block by block after use instead of being
  // allocated and zeroed (1 MB) per solve
  Assembly &A = imu_holder(dimI, n);
  for (int i = 1; i < n; i++) add_frame(*S, *C, n, F, i, A);
  const double tA1 = tmg ? now_us() : 0;
  // The KKT system of OB/EnergyFunctional.cpp:1062-1140 formed in ONE pass over the kept states, already Jacobi-scaled:
  //   K = [(H_imu + expand(H_top) + HM) with the diagonal times (1 + lambda)  -  expand(H_sc) / (1 + lambda)   J^T ; J  0]
  //   rhs = [expand(b_top) + b_imu + bM + HM d2 - expand(b_sc) ; r_cst]
  // restricted to the constrained states (kept indices in order), without the dimI x dimI temporaries of the literal form.
  const int d0 = CP + 8 * n;
  auto gidx = [&](int a) { return a < CP ? a : CP + 1 + 29 * ((a - CP) / 8) + (a - CP) % 8; };  // dso index -> expanded index
  // marginalisation prior around the expanded delta
  static thread_local std::vector<double> d2, bf, diagK;
  d2.assign(dimI, 0.0);
  for (int i = 0; i < CP; i++) d2[i] = delta[i];
  if (C->scale_trapped) d2[CP] = C->scale - C->scale_zero;
  for (int i = 0; i < n; i++) {
    for (int k = 0; k < 8; k++) d2[CP + 1 + 29 * i + k] = delta[CP + 8 * i + k];
    if (C->scale_trapped)
      for (int k = 0; k < 21; k++) d2[CP + 1 + 29 * i + 8 + k] = F[i].state_imu[k] - F[i].state_imu_zero[k];
  }
  const int cdim = (int)A.Jrows.size();
  static thread_local std::vector<int> keep, pos;
  keep.clear();
  for (int i = 0; i < CP; i++) keep.push_back(i);
  if (!S->enable_scale_opt) keep.push_back(CP);
  for (int i = 0; i < n; i++)
    for (int k = 0; k < (A.spline_valid[i] ? 29 : 14); k++) keep.push_back(CP + 1 + 29 * i + k);
  const int ms = (int)keep.size(), m = ms + cdim;
  pos.assign(dimI, -1);
  for (int r = 0; r < ms; r++) pos[keep[r]] = r;
  const double f = 1.0f / (1 + lambda);  // a DOUBLE quotient (float literal over a double sum, :1098); as a float quotient it would be 1.3e-8
                                         // off, and the solve amplifies that to 2e-4 of the step (tests/test_host_on_chain_data.py)
  // right-hand side of the kept states (HM d2 runs over ALL expanded columns) and the unscaled diagonal
  const double tB0 = tmg ? now_us() : 0;
  bf.assign(m, 0.0);  // here: the visual part b_top - b_sc; the prior's H_M d2 joins in the fill loop below, which has the row in cache
  diagK.assign(m, 0.0);
  for (int r = 0; r < ms; r++) {
    const int g = keep[r];
    diagK[r] = A.H(g, g) + HM[(size_t)g * dimI + g];
  }
  for (int a = 0; a < d0; a++) {
    const int r = pos[gidx(a)];
    if (r < 0) continue;
    bf[r] = b_top[a] - b_sc[a];
    diagK[r] += H_top[(size_t)a * d0 + a];
  }
  const double tB1 = tmg ? now_us() : 0;
  static thread_local std::vector<double> K, rhs, sI, sol;
  K.resize((size_t)m * m);  // the state block is written whole below; the multiplier rows / columns are cleared here
  for (int r = 0; r < ms; r++) std::memset(&K[(size_t)r * m + ms], 0, sizeof(double) * cdim);
  for (int k = 0; k < cdim; k++) std::memset(&K[(size_t)(ms + k) * m + ms + k], 0, sizeof(double) * (cdim - k));  // (upper triangle)
  rhs.assign(m, 0.0);
  sI.assign(m, 0.0);
  // the kept states are a few runs of consecutive expanded indices: (first expanded index, first kept position, length)
  struct Run { int g0, r0, len; };
  static thread_local std::vector<Run> runs;
  runs.clear();
  for (int r = 0; r < ms; r++) {
    if (!runs.empty() && runs.back().g0 + runs.back().len == keep[r]) runs.back().len++;
    else runs.push_back(Run{keep[r], r, 1});
  }
  const double tB2 = tmg ? now_us() : 0;
  for (int r = 0; r < ms; r++) diagK[r] *= (1 + lambda);
  for (int a = 0; a < d0; a++) {
    const int r = pos[gidx(a)];
    if (r >= 0) diagK[r] -= H_sc[(size_t)a * d0 + a] * f;
  }
  for (int i = 0; i < m; i++) sI[i] = 1.0 / std::sqrt(diagK[i] + 10);  // multipliers: zero diagonal
  for (int r = 0; r < ms; r++) {
    const int g = keep[r];
    const double *hm = HM + (size_t)g * dimI, *hi = &A.H.a[(size_t)g * dimI];
    double *kr = &K[(size_t)r * m];
    const double sr = sI[r];
    {  // H_M d2 over ALL expanded columns (a single running sum is a chain of dimI dependent additions, 4 cycles each: four vector
       // accumulators, combined at the end)
      __m256d a0 = _mm256_setzero_pd(), a1 = a0, a2 = a0, a3 = a0;
      int c = 0;
      for (; c + 16 <= dimI; c += 16) {
        a0 = _mm256_add_pd(a0, _mm256_mul_pd(_mm256_loadu_pd(hm + c), _mm256_loadu_pd(&d2[c])));
        a1 = _mm256_add_pd(a1, _mm256_mul_pd(_mm256_loadu_pd(hm + c + 4), _mm256_loadu_pd(&d2[c + 4])));
        a2 = _mm256_add_pd(a2, _mm256_mul_pd(_mm256_loadu_pd(hm + c + 8), _mm256_loadu_pd(&d2[c + 8])));
        a3 = _mm256_add_pd(a3, _mm256_mul_pd(_mm256_loadu_pd(hm + c + 12), _mm256_loadu_pd(&d2[c + 12])));
      }
      double t4[4];
      _mm256_storeu_pd(t4, _mm256_add_pd(_mm256_add_pd(a0, a1), _mm256_add_pd(a2, a3)));
      double dot = (t4[0] + t4[1]) + (t4[2] + t4[3]);
      for (; c < dimI; c++) dot += hm[c] * d2[c];
      bf[r] = ((bM[g] + A.b[g]) + dot) + bf[r];
    }
    // (only the upper triangle, c >= r: it is all the factorisation reads)
    if (g == CP) {  // the scale row of H_imu is dense
      for (const Run &u : runs) {
        const double *h1 = hi + u.g0, *h2 = hm + u.g0, *sc = &sI[u.r0];
        double *ko = kr + u.r0;
        for (int i = std::max(0, r - u.r0); i < u.len; i++) ko[i] = (h1[i] + h2[i]) * (sr * sc[i]);
      }
    } else {
      for (const Run &u : runs) {
        const double *h2 = hm + u.g0, *sc = &sI[u.r0];
        double *ko = kr + u.r0;
        for (int i = std::max(0, r - u.r0); i < u.len; i++) ko[i] = h2[i] * (sr * sc[i]);  // == (0 + hm) * scale of the dense form
      }
      if (g > CP) {  // ... and the entries of the row where H_imu has blocks, in the dense form's arithmetic
        const int fi = (g - CP - 1) / 29, k = (g - CP - 1) % 29;
        auto span = [&](int g0, int len) {
          for (int gc = g0; gc < g0 + len; gc++) {
            const int c = pos[gc];
            if (c >= r) kr[c] = (hi[gc] + hm[gc]) * (sr * sI[c]);
          }
        };
        span(CP, 1);
        span(CP + 1 + 29 * fi, 29);
        if (k >= 8 && k < 14) {
          if (fi > 0) span(CP + 1 + 29 * (fi - 1) + 8, 6);
          if (fi < n - 1) span(CP + 1 + 29 * (fi + 1) + 8, 6);
        }
      }
    }
    rhs[r] = bf[r] * sr;
  }
  const double tB3 = tmg ? now_us() : 0;
  for (int a = 0; a < d0; a++) {  // the visual system: dense over calibration and poses
    const int r = pos[gidx(a)];
    if (r < 0) continue;
    double *kr = &K[(size_t)r * m];
    const double *ht = H_top + (size_t)a * d0, *hs = H_sc + (size_t)a * d0;
    const double sr = sI[r];
    for (int b2 = 0; b2 < d0; b2++) {
      const int c = pos[gidx(b2)];
      if (c >= r) kr[c] += (ht[b2] - hs[b2] * f) * (sr * sI[c]);
    }
  }
  const double tB4 = tmg ? now_us() : 0;
  for (int r = 0; r < ms; r++) K[(size_t)r * m + r] = diagK[r] * sI[r] * sI[r];  // (1 + lambda) on the whole diagonal
  for (int k = 0; k < cdim; k++) {
    const double *J = A.Jrows[k];
    const double sk = sI[ms + k];
    for (int r = 0; r < ms; r++) {
      const double v = J[keep[r]];
      if (v != 0.0) K[(size_t)r * m + ms + k] = v * (sk * sI[r]);
    }
    rhs[ms + k] = A.r[k] * sk;
  }
  clear_imu_blocks(A, n);  // H_imu back to zero for the next call
  const double tA2 = tmg ? now_us() : 0;
  // blocked LDL^T with threshold pivoting on the diagonal (sos_math.hpp): the KKT matrix is indefinite, but quasi-definite in the order
  // states first, multipliers last -- the multipliers only become pivots once the states they constrain are eliminated
  sos::ldlt_solve(K, rhs, sol, m, K.data());  // in place: K is rebuilt by every call
  if (tmg) fprintf(stderr, "[imu_build] pre %.0f rhs %.0f zero %.0f fill %.0f visual %.0f cst %.0f\n", tB0 - tA1, tB1 - tB0, tB2 - tB1, tB3 - tB2, tB4 - tB3, tA2 - tB4);
  if (tmg) fprintf(stderr, "[imu_solve] assemble %.0f us, dense build (dim %d, kkt %d) %.0f us, ldlt %.0f us\n", tA1 - tA0, dimI, m, tA2 - tA1, now_us() - tA2);
  for (int i = 0; i < m; i++) sol[i] *= sI[i];
  // split into the dso increment, the scale step and the IMU steps
  std::memset(x, 0, sizeof(double) * (CP + 8 * n));
  std::memset(step_imu, 0, sizeof(double) * 21 * n);
  *scale_step = 0;
  for (int r = 0; r < ms; r++) {
    const int g = keep[r];
    if (g < CP) x[g] = sol[r];
    else if (g == CP) *scale_step = -sol[r];
    else {
      const int i = (g - CP - 1) / 29, k = (g - CP - 1) % 29;
      if (k < 8) x[CP + 8 * i + k] = sol[r];
      else step_imu[21 * i + (k - 8)] = -sol[r];
    }
  }
  return SOS_OK;
}

// ================================================================================================
// The IMU branch with first-estimate Jacobians (scale trapped), cached.  Once the scale is trapped, getImuHi evaluates its Jacobians at
// state_imu_zero / scale_zero / camToWorld_evalPT (FS/HessianBlocks.cpp:178-225), the spline constraints are linear with constant
// rows, HM and lambda are fixed: of the whole KKT matrix of OB/EnergyFunctional.cpp:1062-1140 only the visual block H_top - H_sc / (1 +
// lambda) (calibration + 8 pose / affine states per keyframe) changes between the iterations of one optimize().  So the unknowns are
// ordered  [ bias + spline states of every keyframe | constraint multipliers | calibration, scale, pose / affine states ]  and the
// leading part -- 3/4 of the unknowns, 85 % of the factorisation -- is eliminated ONCE; an iteration then costs the right-hand sides
// (sample residuals against the kept J^T W), one forward pass, the (4 + 1 + 8 n)-dimensional border solve and one backward pass.
// None of it needs the device's H / b until the border: sosf_imu_solve_prepare runs while the accumulation is in flight.
// What "constant" rests on is compared value by value on every call (settings, linearisation points, timestamps, the whole of HM):
// any difference rebuilds the factor; three rebuilds in a row send the calls to the literal form until the inputs repeat.
// ================================================================================================
namespace {
struct ImuCache {
  bool valid = false;
  int n = 0, dimI = 0, nIs = 0, cdim = 0, mI = 0, nb = 0, nt = 0;
  std::vector<double> sig, HMcopy, HMdiag;
  uint64_t prior_id = 0;            // != 0: the caller's name for the values of HM (compared instead of the values; the diagonal still is)
  const double *HMptr = nullptr;
  std::vector<int> gI, gB;          // expanded index of interior state u (u < nIs) / of border unknown j
  std::vector<int> pI, pC;          // position of interior state u / of constraint multiplier k in the interior's elimination order:
                                    // keyframe by keyframe [bias + spline states | multipliers of the constraints the keyframe owns]
  std::vector<double> bI0, rc0;     // b_imu / constraint residuals of the assembly the factor was built from (the untrapped solve's)
  std::vector<int> aB;              // dso index (4 + 8 n system) of border unknown j, -1 for the scale
  std::vector<int> spline_valid, rows_of;  // per frame: spline valid, constraint rows it owns
  std::vector<double> scI;          // Jacobi scale of the interior unknowns, by position
  std::vector<double> HcDiagB;      // (H_imu + HM) diagonal at the border unknowns, unscaled
  HiStore hi;
  sos::LdltPartial F;
  int misses = 0;                   // rebuilds in a row
  bool unusable = false;            // the leading block of THESE inputs (sig) is singular beyond its all-zero constraint rows: literal form
  // the call in flight (prepare -> finish)
  std::vector<double> y;            // nt: forward-substituted right-hand side
  std::vector<double> rhsB;         // nb: constant part of the border right-hand side (prior + IMU), before the forward pass
  std::vector<double> HMd;          // nIs: (HM d2) at the interior states, left by the build that scanned those rows for THIS call's d2
  bool HMd_fresh = false;
};
struct Prepared {
  bool active = false, cached = false;
  const sosf_imu_settings *S = nullptr;
  const sosf_imu_calib *C = nullptr;
  int n = 0;
  const sosf_imu_frame *F = nullptr;
  const double *HM = nullptr, *bM = nullptr, *delta = nullptr;
  double lambda = 0;
  uint64_t prior_id = 0;
};
thread_local ImuCache g_cache;
thread_local Prepared g_prep;
int g_mode = 1;                     // 0 literal form; 1 cached (sosf_imu_solve_mode switches)
thread_local int g_stats[3] = {0, 0, 0};  // solves on a kept factor / rebuilds / literal-form solves

void make_signature(const sosf_imu_settings &S, const sosf_imu_calib &C, int n, const sosf_imu_frame *F, double lambda, std::vector<double> &sig) {
  sig.clear();
  sig.push_back(n);
  sig.push_back(lambda);
  sig.push_back(S.enable_scale_opt);
  sig.push_back(S.maxImuInterval);
  sig.insert(sig.end(), S.weight_imu, S.weight_imu + 36);
  sig.insert(sig.end(), S.weight_imu_bias, S.weight_imu_bias + 36);
  sig.insert(sig.end(), S.gravity, S.gravity + 3);
  sig.insert(sig.end(), S.rot_imu_cam, S.rot_imu_cam + 9);
  sig.push_back(C.scale_zero);
  for (int i = 0; i < n; i++) {
    const sosf_imu_frame &f = F[i];
    sig.push_back(f.timestamp);
    sig.push_back(f.trackingRefIsPrev);
    sig.push_back(f.n_imu);
    sig.insert(sig.end(), f.evalPT_R, f.evalPT_R + 9);
    sig.insert(sig.end(), f.state_imu_zero, f.state_imu_zero + 21);
    for (int j = 0; j < f.n_imu; j++) sig.push_back(f.imu[7 * j]);
  }
}

__attribute__((target("avx2,fma"))) double dot4(const double *a, const double *b, int len) {
  __m256d a0 = _mm256_setzero_pd(), a1 = a0, a2 = a0, a3 = a0;
  int c = 0;
  for (; c + 16 <= len; c += 16) {
    a0 = _mm256_fmadd_pd(_mm256_loadu_pd(a + c), _mm256_loadu_pd(b + c), a0);
    a1 = _mm256_fmadd_pd(_mm256_loadu_pd(a + c + 4), _mm256_loadu_pd(b + c + 4), a1);
    a2 = _mm256_fmadd_pd(_mm256_loadu_pd(a + c + 8), _mm256_loadu_pd(b + c + 8), a2);
    a3 = _mm256_fmadd_pd(_mm256_loadu_pd(a + c + 12), _mm256_loadu_pd(b + c + 12), a3);
  }
  double t4[4];
  _mm256_storeu_pd(t4, _mm256_add_pd(_mm256_add_pd(a0, a1), _mm256_add_pd(a2, a3)));
  double dot = (t4[0] + t4[1]) + (t4[2] + t4[3]);
  for (; c < len; c++) dot += a[c] * b[c];
  return dot;
}

// the constant part of the KKT matrix in the cache's ordering, and its partial factorisation
void build_cache(ImuCache &Q, const sosf_imu_settings &S, const sosf_imu_calib &C, int n, const sosf_imu_frame *F, const double *HM, double lambda,
                 uint64_t prior_id, bool keep = true, const double *d2 = nullptr) {
  const int dimI = SOSF_IMU_DIM(n);
  static const bool tmgb = getenv("SOS_TIMING_IMU") != nullptr;
  const double tq0 = tmgb ? now_us() : 0;
  Assembly &A = imu_holder(dimI, n);
  Q.hi.JsTW.clear();
  Q.hi.JfTW.clear();
  Q.rows_of.assign(n, 0);
  for (int i = 1; i < n; i++) {
    const size_t r0 = A.Jrows.size();
    add_frame(S, C, n, F, i, A, &Q.hi);
    Q.rows_of[i] = (int)(A.Jrows.size() - r0);
  }
  const double tq1 = tmgb ? now_us() : 0;
  Q.n = n;
  Q.dimI = dimI;
  Q.spline_valid = A.spline_valid;
  Q.cdim = (int)A.Jrows.size();
  Q.gI.clear();
  Q.gB.clear();
  Q.aB.clear();
  for (int i = 0; i < CP; i++) { Q.gB.push_back(i); Q.aB.push_back(i); }
  if (!S.enable_scale_opt) { Q.gB.push_back(CP); Q.aB.push_back(-1); }
  for (int i = 0; i < n; i++) {
    for (int k = 0; k < 8; k++) { Q.gB.push_back(CP + 1 + 29 * i + k); Q.aB.push_back(CP + 8 * i + k); }
    for (int k = 8; k < (A.spline_valid[i] ? 29 : 14); k++) Q.gI.push_back(CP + 1 + 29 * i + k);
  }
  Q.nIs = (int)Q.gI.size();
  Q.mI = Q.nIs + Q.cdim;
  Q.nb = (int)Q.gB.size();
  Q.nt = Q.mI + Q.nb;
  const int nt = Q.nt, nIs = Q.nIs, mI = Q.mI, nb = Q.nb;
  // ---- elimination order of the interior: one pivot block per keyframe, [its bias + spline states | the multipliers of the
  // constraint rows it owns].  H_imu couples a block to its neighbours only (bias random walk; velocity continuity reaches the next
  // keyframe's spline states) and to the border (poses, calibration, scale); the prior a running chain leaves behind has the same
  // shape.  What the data really couples is read off below (reach), so that an unusual prior widens the envelope instead of being
  // assumed away.
  std::vector<int> bstart(n + 1, 0), blockOfState(nIs, 0);
  Q.pI.assign(nIs, 0);
  Q.pC.assign(Q.cdim, 0);
  {
    int pos = 0, u = 0, k = 0;
    for (int i = 0; i < n; i++) {
      bstart[i] = pos;
      for (int q = 8; q < (A.spline_valid[i] ? 29 : 14); q++, u++) { Q.pI[u] = pos++; blockOfState[u] = i; }
      for (int q = 0; q < Q.rows_of[i]; q++, k++) Q.pC[k] = pos++;
    }
    bstart[n] = pos;
  }
  std::vector<int> reach(n);
  for (int i = 0; i < n; i++) reach[i] = std::min(i + 1, n - 1);
  {
    std::vector<int> u0(n + 1, 0);  // first interior state of keyframe i
    for (int u = 0; u < nIs; u++) u0[blockOfState[u] + 1] = u + 1;
    for (int i = 1; i <= n; i++) u0[i] = std::max(u0[i], u0[i - 1]);
    // (the interior states of a keyframe are consecutive expanded indices: a block of the prior is scanned as contiguous row segments,
    // a constraint row as one contiguous pass with the keyframe of each expanded index looked up.  The scan of a prior without
    // far couplings touches 3.5 k cache lines of HM, 33 us at W12 -- the price of not assuming the prior's shape)
    // the prior beyond the neighbour, row by row; with the caller's expanded delta at hand the row's share of HM d2 is taken in the same
    // visit (the right-hand side needs it, and the scan alone cost as much as that product: 3.5 k cold lines of HM at W12)
    Q.HMd_fresh = d2 != nullptr;
    if (d2) Q.HMd.resize(nIs);
    for (int i = 0; i < n; i++)
      for (int a = u0[i]; a < u0[i + 1]; a++) {
        const double *row = HM + (size_t)Q.gI[a] * dimI;
        if (d2) Q.HMd[a] = dot4(row, d2, dimI);
        for (int j = n - 1; j > reach[i]; j--) {
          const int len = u0[j + 1] - u0[j];
          if (len <= 0) continue;
          const double *seg = row + Q.gI[u0[j]];
          int nz = 0;
          for (int c = 0; c < len; c++) nz |= seg[c] != 0.0;
          if (nz) { reach[i] = j; break; }
        }
      }
    std::vector<int> blockOfG(dimI, -1);  // keyframe of an interior state's expanded index, -1: border
    for (int u = 0; u < nIs; u++) blockOfG[Q.gI[u]] = blockOfState[u];
    int k = 0;
    for (int i = 0; i < n; i++)       // constraint rows against the interior states of other keyframes
      for (int q = 0; q < Q.rows_of[i]; q++, k++) {
        const double *J = A.Jrows[k];
        for (int g = 0; g < dimI; g++)
          if (J[g] != 0.0 && blockOfG[g] >= 0) {
            const int f = blockOfG[g], lo = std::min(f, i), hi = std::max(f, i);
            reach[lo] = std::max(reach[lo], hi);
          }
      }
    for (int i = 1; i < n; i++) reach[i] = std::max(reach[i], reach[i - 1]);  // fill stays inside a monotone envelope
  }
  sos::LdltPartial &P = Q.F;
  P.n = nt;
  P.m = mI;
  P.blk_end.assign(mI, mI);
  P.env_end.assign(mI, mI);
  for (int i = 0; i < n; i++)
    for (int p = bstart[i]; p < bstart[i + 1]; p++) { P.blk_end[p] = bstart[i + 1]; P.env_end[p] = bstart[reach[i] + 1]; }
  // Jacobi scale of the interior: sqrt(diagonal + 10)^-1 as the reference scales the whole system (:1143-1146); the border is left
  // unscaled here (no pivot is taken from it) and scaled at the border solve, where its diagonal is complete
  Q.scI.assign(mI, 1.0 / std::sqrt(10.0));
  for (int u = 0; u < nIs; u++) {
    const int g = Q.gI[u];
    Q.scI[Q.pI[u]] = 1.0 / std::sqrt((A.H(g, g) + HM[(size_t)g * dimI + g]) * (1 + lambda) + 10);
  }
  Q.HcDiagB.resize(nb);
  for (int j = 0; j < nb; j++) {
    const int g = Q.gB[j];
    Q.HcDiagB[j] = A.H(g, g) + HM[(size_t)g * dimI + g];
  }
  P.U.resize((size_t)nt * nt);  // only the envelope of the upper triangle and the border columns are ever read
  double *U = P.U.data();
  // what sits at an interior position: the expanded index of a state (>= 0) or -1 - k for multiplier k
  std::vector<int> what(mI);
  for (int u = 0; u < nIs; u++) what[Q.pI[u]] = Q.gI[u];
  for (int k = 0; k < Q.cdim; k++) what[Q.pC[k]] = -1 - k;
  struct Run { int g0, u0, len; };
  std::vector<Run> runsB;
  for (int j = 0; j < nb; j++) {
    if (!runsB.empty() && runsB.back().g0 + runsB.back().len == Q.gB[j]) runsB.back().len++;
    else runsB.push_back(Run{Q.gB[j], j, 1});
  }
  const double *Jcol = A.Jrows.a.data();  // the constraint rows, contiguous
  for (int p = 0; p < mI; p++) {
    double *row = U + (size_t)p * nt;
    const double sp = Q.scI[p];
    const int ee = P.env_end[p];
    if (what[p] >= 0) {  // a state row: states (H_imu + HM), multipliers (the constraint's entry), border
      const int g = what[p];
      const double *hi = &A.H.a[(size_t)g * dimI], *hm = HM + (size_t)g * dimI;
      row[p] = (hi[g] + hm[g]) * (1 + lambda) * (sp * sp);
      for (int c = p + 1; c < ee; c++) {
        const int w = what[c];
        row[c] = w >= 0 ? (hi[w] + hm[w]) * (sp * Q.scI[c]) : Jcol[(size_t)(-1 - w) * dimI + g] * (sp * Q.scI[c]);
      }
      // H_imu couples a state to its own keyframe's columns, the neighbours' biases and the scale (add_frame): over the border it is
      // zero outside the run with the keyframe's pose and the run with the scale column, and its row need not be read there
      const int ci = CP + 1 + 29 * ((g - CP - 1) / 29);
      for (const Run &r : runsB) {
        const double *h1 = hi + r.g0, *h2 = hm + r.g0;
        double *o = row + mI + r.u0;
        const bool imu = (r.g0 < ci + 8 && r.g0 + r.len > ci) || (r.g0 <= CP && r.g0 + r.len > CP);
        if (imu) for (int i = 0; i < r.len; i++) o[i] = (h1[i] + h2[i]) * sp;
        else for (int i = 0; i < r.len; i++) o[i] = h2[i] * sp;
      }
    } else {             // a multiplier row: zero against the other multipliers, the constraint's entries elsewhere
      const double *J = &Jcol[(size_t)(-1 - what[p]) * dimI];
      row[p] = 0.0;
      for (int c = p + 1; c < ee; c++) {
        const int w = what[c];
        row[c] = w >= 0 ? J[w] * (sp * Q.scI[c]) : 0.0;
      }
      for (int j = 0; j < nb; j++) row[mI + j] = J[Q.gB[j]] * sp;
    }
  }
  for (int j = 0; j < nb; j++) {  // border rows: H_imu + HM, the diagonal times (1 + lambda); the visual block joins per iteration
    const int g = Q.gB[j];
    const double *hi = &A.H.a[(size_t)g * dimI], *hm = HM + (size_t)g * dimI;
    double *row = U + (size_t)(mI + j) * nt + mI;
    for (const Run &r : runsB) {
      const int i0 = std::max(0, j + 1 - r.u0);
      const double *h1 = hi + r.g0, *h2 = hm + r.g0;
      double *o = row + r.u0;
      for (int i = i0; i < r.len; i++) o[i] = h1[i] + h2[i];
    }
    row[j] = (hi[g] + hm[g]) * (1 + lambda);
  }
  Q.bI0 = A.b;
  Q.rc0 = A.r;
  clear_imu_blocks(A, n);
  const double tf0 = tmgb ? now_us() : 0;
  sos::ldlt_partial_factor(P);
  const double tf1 = tmgb ? now_us() : 0;
  // A zero pivot is expected for every all-zero constraint row (velocity rows whose successor has no valid spline, :389-391) and
  // nowhere else.  Any other one means states without information (a valid spline with no samples and no prior, say): the literal
  // form's pivoting over the WHOLE system then decides what the step is, so those inputs are left to it.
  {
    int zeroRows = 0, zeroPiv = 0;
    for (int k = 0; k < Q.cdim; k++) {
      bool any = false;
      for (int g = 0; g < dimI && !any; g++) any = A.Jrows[k][g] != 0.0;
      if (!any) zeroRows++;
    }
    for (int k = 0; k < mI; k++)
      if (!(std::fabs(P.D[k]) > 2.2250738585072014e-308)) zeroPiv++;
    Q.unusable = zeroPiv != zeroRows;
  }
  Q.prior_id = prior_id;
  Q.HMptr = HM;
  if (keep) {  // what a later call is compared with
    Q.HMdiag.resize(dimI);
    for (int g = 0; g < dimI; g++) Q.HMdiag[g] = HM[(size_t)g * dimI + g];
    if (prior_id == 0) Q.HMcopy.assign(HM, HM + (size_t)dimI * dimI);
  }
  Q.valid = true;
  if (tmgb) {
    long env = 0;
    for (int p = 0; p < mI; p++) env += P.env_end[p] - p;
    fprintf(stderr, "[imu_cached] build: assemble %.0f us, fill %.0f us, partial factorisation (%d of %d, mean envelope %.1f) %.0f us, copy %.0f us\n", tq1 - tq0, tf0 - tq1, mI, nt,
            mI ? (double)env / mI : 0.0, tf1 - tf0, now_us() - tf1);
  }
}

// b_imu (expanded dimension, kept between calls zeroed by the caller) and the constraint residuals at the current states, from the kept
// per-sample J^T W: the right-hand-side half of getImuHessianCurrentFrame (OB/EnergyFunctional.cpp:296-494)
void imu_rhs(const ImuCache &Q, const sosf_imu_settings &S, const sosf_imu_calib &C, int n, const sosf_imu_frame *F, double *b, double *r) {
  size_t smp = 0;
  int row0 = 0;
  const M3 Ric = M3::from(S.rot_imu_cam);
  const double scale_scaled = C.scale * kScale;
  for (int fi = 1; fi < n; fi++) {
    const sosf_imu_frame &cur = F[fi], &prv = F[fi - 1];
    const ImuView vc(cur), vp(prv);
    const double tpf = prv.timestamp - cur.timestamp, tpf2 = tpf * tpf;
    const int ci = CP + 1 + 29 * fi, pi = CP + 1 + 29 * (fi - 1);
    double rb[6];
    for (int k = 0; k < 6; k++) rb[k] = vc.bias(k) - vp.bias(k);
    for (int q = 0; q < 6; q++) {
      double sacc = 0;
      for (int k = 0; k < 6; k++) sacc += (S.weight_imu_bias[6 * q + k] / -tpf) * rb[k];
      const double tb = sacc * (q < 3 ? kBa : kBg);
      b[pi + 8 + q] += -tb;
      b[ci + 8 + q] += tb;
    }
    if (!Q.spline_valid[fi]) continue;
    const M3 Rc = M3::from(cur.camToWorld), Rp = M3::from(prv.camToWorld);
    const V3 rr = so3_log((Rc.T() * Rp).T() * vc.R_c_t(tpf, false));
    for (int q = 0; q < 3; q++) r[row0 + q] = rr[q];
    if (Q.rows_of[fi] == 6) {
      for (int q = 0; q < 3; q++) r[row0 + 3 + q] = 0.0;
      const sosf_imu_frame &nxt = F[fi + 1];
      const double tnf = cur.timestamp - nxt.timestamp;
      if (nxt.trackingRefIsPrev && (-tnf < S.maxImuInterval)) {
        const ImuView vn(nxt);
        const double tnf2 = tnf * tnf;
        for (int q = 0; q < 3; q++) {
          const double dso = (1 / tpf) * (prv.camToWorld[9 + q] - cur.camToWorld[9 + q]) - (1 / tnf) * (cur.camToWorld[9 + q] - nxt.camToWorld[9 + q]);
          const double imu = tpf * vc.sc[9 + q] + tpf2 * vc.sc[15 + q] + tnf * vn.sc[9 + q] + 2 * tnf2 * vn.sc[15 + q];
          r[row0 + 3 + q] = imu - dso;
        }
      }
    }
    row0 += Q.rows_of[fi];
    const M3 Rwc = Rc.T();
    for (int j = 0; j < cur.n_imu; j++, smp++) {
      const double tt = cur.imu[7 * j] - cur.timestamp;
      const V3 a = vc.acc(tt, false);
      V3 aw;
      for (int i = 0; i < 3; i++) aw[i] = scale_scaled * a[i] + S.gravity[i];
      const V3 pa = ((Ric * vc.R_c_t(tt, false).T()) * Rwc) * aw, pg = Ric * vc.gyro(tt);
      double res[6];
      for (int i = 0; i < 3; i++) {
        res[i] = (pa[i] + vc.bias(i)) - cur.imu[7 * j + 1 + i];
        res[3 + i] = (pg[i] + vc.bias(3 + i)) - cur.imu[7 * j + 4 + i];
      }
      const double *js = &Q.hi.JsTW[6 * smp], *jf = &Q.hi.JfTW[174 * smp];
      double sacc = 0;
      for (int k = 0; k < 6; k++) sacc += js[k] * res[k];
      b[CP] += sacc;
      for (int q = 0; q < 29; q++) {
        double t = 0;
        for (int k = 0; k < 6; k++) t += jf[6 * q + k] * res[k];
        b[ci + q] += t;
      }
    }
  }
}

// trapped == false (the scale not yet trapped: Jacobians at the current states, nothing to keep): the same elimination, built and used once
int cached_prepare(ImuCache &Q, const Prepared &P, bool trapped) {
  static const bool tmg = getenv("SOS_TIMING_IMU") != nullptr;
  const sosf_imu_settings &S = *P.S;
  const sosf_imu_calib &C = *P.C;
  const int n = P.n, dimI = SOSF_IMU_DIM(n);
  const double t0 = tmg ? now_us() : 0;
  double tb = t0;
  static thread_local std::vector<double> sig;
  if (trapped) make_signature(S, C, n, P.F, P.lambda, sig);
  bool same = trapped && Q.valid && sig == Q.sig && Q.prior_id == P.prior_id;
  if (same && P.prior_id != 0) {  // a named prior: same name, same place, same diagonal
    same = Q.HMptr == P.HM;
    for (int g = 0; same && g < dimI; g++) same = Q.HMdiag[g] == P.HM[(size_t)g * dimI + g];
  } else if (same) {
    same = std::memcmp(Q.HMcopy.data(), P.HM, sizeof(double) * (size_t)dimI * dimI) == 0;
  }
  tb = tmg ? now_us() : 0;
  static thread_local std::vector<double> d2;  // the expanded delta the prior is taken around (:1081-1098)
  d2.assign(dimI, 0.0);
  for (int i = 0; i < CP; i++) d2[i] = P.delta[i];
  if (C.scale_trapped) d2[CP] = C.scale - C.scale_zero;
  for (int i = 0; i < n; i++) {
    for (int k = 0; k < 8; k++) d2[CP + 1 + 29 * i + k] = P.delta[CP + 8 * i + k];
    if (C.scale_trapped)
      for (int k = 0; k < 21; k++) d2[CP + 1 + 29 * i + 8 + k] = P.F[i].state_imu[k] - P.F[i].state_imu_zero[k];
  }
  Q.HMd_fresh = false;
  bool once = !trapped;  // the elimination is built for this call alone (counted as a literal-form solve)
  if (!trapped) {
    build_cache(Q, S, C, n, P.F, P.HM, P.lambda, P.prior_id, false, d2.data());
    Q.valid = false;  // (nothing of it is kept: the next call's Jacobians are taken elsewhere)
    Q.sig.clear();
    tb = tmg ? now_us() : 0;
    if (Q.unusable) return 1;
  } else if (!same) {
    const bool repeats = !Q.sig.empty() && sig == Q.sig;  // same inputs as the previous call: a prior that moved, or a cache given up on
    Q.valid = false;
    Q.sig = sig;
    if (Q.misses >= 3 && !repeats) {  // the inputs move with every call: nothing is kept, the elimination is built and used once
      build_cache(Q, S, C, n, P.F, P.HM, P.lambda, P.prior_id, false, d2.data());
      Q.valid = false;
      tb = tmg ? now_us() : 0;
      if (Q.unusable) return 1;
      once = true;
    } else {
      Q.misses++;
      g_stats[1]++;
      build_cache(Q, S, C, n, P.F, P.HM, P.lambda, P.prior_id, true, d2.data());
      tb = tmg ? now_us() : 0;
      if (Q.unusable) return 1;
    }
  } else if (Q.unusable) {
    return 1;
  } else {
    Q.misses = 0;
    g_stats[0]++;
  }
  // right-hand side: prior around the expanded delta (bM + HM d2, :1081-1098), b_imu, constraint residuals
  static thread_local std::vector<double> bI, rc;
  if (C.scale_trapped) {
    bI.assign(dimI, 0.0);
    rc.assign(Q.cdim, 0.0);
    imu_rhs(Q, S, C, n, P.F, bI.data(), rc.data());
  } else {  // the assembly the factor was just built from holds b_imu and the constraint residuals of these very states
    bI = Q.bI0;
    rc = Q.rc0;
  }
  const double t1 = tmg ? now_us() : 0;
  Q.y.assign(Q.nt, 0.0);
  Q.rhsB.assign(Q.nb, 0.0);
  for (int u = 0; u < Q.nIs; u++) {
    const int g = Q.gI[u], p = Q.pI[u];
    Q.y[p] = ((P.bM[g] + bI[g]) + (Q.HMd_fresh ? Q.HMd[u] : dot4(P.HM + (size_t)g * dimI, d2.data(), dimI))) * Q.scI[p];
  }
  for (int k = 0; k < Q.cdim; k++) Q.y[Q.pC[k]] = rc[k] * Q.scI[Q.pC[k]];
  for (int j = 0; j < Q.nb; j++) {
    const int g = Q.gB[j];
    Q.y[Q.mI + j] = (P.bM[g] + bI[g]) + dot4(P.HM + (size_t)g * dimI, d2.data(), dimI);
  }
  const double t2 = tmg ? now_us() : 0;
  sos::ldlt_partial_forward(Q.F, Q.y.data());  // the visual part of the border's right-hand side is added after it: the pass is linear
  if (tmg)
    fprintf(stderr, "[imu_cached] %s: compare/build %.0f us, residuals %.0f us, prior rhs %.0f us, forward %.0f us (interior %d, border %d)\n",
            same ? "hit" : "REBUILD", tb - t0, t1 - tb, t2 - t1, now_us() - t2, Q.mI, Q.nb);
  return once ? 2 : 0;
}

void cached_finish(ImuCache &Q, const Prepared &P, const double *H_top, const double *b_top, const double *H_sc, const double *b_sc, double *x,
                   double *scale_step, double *step_imu) {
  static const bool tmg = getenv("SOS_TIMING_IMU") != nullptr;
  const double t0 = tmg ? now_us() : 0;
  const int n = P.n, d0 = CP + 8 * n, nb = Q.nb, nt = Q.nt, mI = Q.mI;
  const double lambda = P.lambda, f = 1.0f / (1 + lambda);
  static thread_local std::vector<double> B, rb, sB, xb;
  B.resize((size_t)nb * nb);
  rb.resize(nb);
  sB.resize(nb);
  // Jacobi scale of the border from the diagonal of the WHOLE system as the reference forms it (:1143)
  for (int j = 0; j < nb; j++) {
    const int a = Q.aB[j];
    double dg = Q.HcDiagB[j];
    if (a >= 0) dg += H_top[(size_t)a * d0 + a];
    dg *= (1 + lambda);
    if (a >= 0) dg -= H_sc[(size_t)a * d0 + a] * f;
    sB[j] = 1.0 / std::sqrt(dg + 10);
  }
  for (int j = 0; j < nb; j++) {  // Schur complement of the constant part + the visual block, upper triangle
    const int a = Q.aB[j];
    const double *sr = &Q.F.U[(size_t)(mI + j) * nt + mI];
    double *br = &B[(size_t)j * nb];
    const double sj = sB[j];
    double dg = Q.F.diag[mI + j];
    if (a >= 0) dg += H_top[(size_t)a * d0 + a] * (1 + lambda) - H_sc[(size_t)a * d0 + a] * f;
    br[j] = dg * (sj * sj);
    if (a < 0) {
      for (int c = j + 1; c < nb; c++) br[c] = sr[c] * (sj * sB[c]);
    } else {
      const double *ht = H_top + (size_t)a * d0, *hs = H_sc + (size_t)a * d0;
      for (int c = j + 1; c < nb; c++) {
        const int a2 = Q.aB[c];
        br[c] = (a2 >= 0 ? sr[c] + (ht[a2] - hs[a2] * f) : sr[c]) * (sj * sB[c]);
      }
    }
    rb[j] = (Q.y[mI + j] + (a >= 0 ? b_top[a] - b_sc[a] : 0.0)) * sj;
  }
  // (the scale sits between calibration and poses in the border: for a < 0 rows the visual block has no entry; for a >= 0 rows the
  //  column of the scale has none either -- handled by the a2 test above)
  const double t1 = tmg ? now_us() : 0;
  sos::ldlt_solve(B, rb, xb, nb, B.data());
  const double t2 = tmg ? now_us() : 0;
  for (int j = 0; j < nb; j++) Q.y[mI + j] = xb[j] * sB[j];
  sos::ldlt_partial_backward(Q.F, Q.y.data());
  std::memset(x, 0, sizeof(double) * d0);
  std::memset(step_imu, 0, sizeof(double) * 21 * n);
  *scale_step = 0;
  for (int j = 0; j < nb; j++) {
    if (Q.aB[j] >= 0) x[Q.aB[j]] = Q.y[mI + j];
    else *scale_step = -Q.y[mI + j];
  }
  for (int u = 0; u < Q.nIs; u++) {
    const int g = Q.gI[u], i = (g - CP - 1) / 29, k = (g - CP - 1) % 29;
    step_imu[21 * i + (k - 8)] = -(Q.y[Q.pI[u]] * Q.scI[Q.pI[u]]);
  }
  if (tmg) fprintf(stderr, "[imu_cached] border build %.0f us, border solve (dim %d) %.0f us, backward %.0f us\n", t1 - t0, nb, t2 - t1, now_us() - t2);
}
}  // namespace

extern "C" int sosf_imu_solve_prepare(const sosf_imu_settings *S, const sosf_imu_calib *C, int n, const sosf_imu_frame *F, const double *HM,
                                      const double *bM, const double *delta, double lambda, uint64_t prior_id) {
  g_prep.active = false;
  if (!S || !C || n < 1 || !F || !HM || !bM || !delta) return SOS_ERR_ARG;
  g_prep = Prepared{true, false, S, C, n, F, HM, bM, delta, lambda, prior_id};
  int form = 1;  // 0 kept factor (or its rebuild), 2 the structured elimination built for this call alone, 1 the dense KKT system
  if (g_mode == 1) form = cached_prepare(g_cache, g_prep, C->scale_trapped != 0);
  g_prep.cached = form != 1;
  if (form != 0) g_stats[2]++;  // (literal = the whole system built and factorised by this call, in either form)
  return SOS_OK;
}

extern "C" int sosf_imu_solve_prepared_form(void) { return !g_prep.active ? -1 : g_prep.cached ? 1 : 0; }

extern "C" int sosf_imu_solve_mode(int mode) {
  const int before = g_mode;
  if (mode == 0 || mode == 1) g_mode = mode;
  return before;
}

extern "C" int sosf_imu_solve_stats(int32_t *kept, int32_t *rebuilt, int32_t *literal, int reset) {
  if (kept) *kept = g_stats[0];
  if (rebuilt) *rebuilt = g_stats[1];
  if (literal) *literal = g_stats[2];
  if (reset) g_stats[0] = g_stats[1] = g_stats[2] = 0;
  return SOS_OK;
}

extern "C" int sosf_imu_solve_finish(const double *H_top, const double *b_top, const double *H_sc, const double *b_sc, double *x, double *scale_step,
                                     double *step_imu) {
  if (!g_prep.active) return SOS_ERR_STATE;
  g_prep.active = false;
  if (!H_top || !b_top || !H_sc || !b_sc || !x || !scale_step || !step_imu) return SOS_ERR_ARG;
  const Prepared &P = g_prep;
  if (!P.cached) return imu_solve_dense(P.S, P.C, P.n, P.F, H_top, b_top, H_sc, b_sc, P.HM, P.bM, P.delta, P.lambda, x, scale_step, step_imu);
  cached_finish(g_cache, P, H_top, b_top, H_sc, b_sc, x, scale_step, step_imu);
  return SOS_OK;
}

extern "C" int sosf_imu_solve(const sosf_imu_settings *S, const sosf_imu_calib *C, int n, const sosf_imu_frame *F, const double *H_top,
                              const double *b_top, const double *H_sc, const double *b_sc, const double *HM, const double *bM,
                              const double *delta, double lambda, double *x, double *scale_step, double *step_imu) {
  if (!S || !C || n < 1 || !F || !H_top || !b_top || !H_sc || !b_sc || !HM || !bM || !delta || !x || !scale_step || !step_imu) return SOS_ERR_ARG;
  const int rc = sosf_imu_solve_prepare(S, C, n, F, HM, bM, delta, lambda, 0);
  if (rc != SOS_OK) return rc;
  return sosf_imu_solve_finish(H_top, b_top, H_sc, b_sc, x, scale_step, step_imu);
}

namespace {
// inverse of a small dense matrix: LU with partial pivoting, then the columns of the identity
std::vector<double> inverse(std::vector<double> A, int n) {
  std::vector<int> piv(n);
  for (int k = 0; k < n; k++) {
    int p = k;
    for (int r = k + 1; r < n; r++)
      if (std::fabs(A[(size_t)r * n + k]) > std::fabs(A[(size_t)p * n + k])) p = r;
    piv[k] = p;
    if (p != k)
      for (int c = 0; c < n; c++) std::swap(A[(size_t)k * n + c], A[(size_t)p * n + c]);
    const double d = A[(size_t)k * n + k];
    for (int r = k + 1; r < n; r++) {
      const double f = A[(size_t)r * n + k] / d;
      A[(size_t)r * n + k] = f;
      for (int c = k + 1; c < n; c++) A[(size_t)r * n + c] -= f * A[(size_t)k * n + c];
    }
  }
  std::vector<double> X((size_t)n * n, 0.0), col(n);
  for (int j = 0; j < n; j++) {
    for (int r = 0; r < n; r++) col[r] = r == j ? 1.0 : 0.0;
    for (int k = 0; k < n; k++) std::swap(col[k], col[piv[k]]);
    for (int r = 1; r < n; r++)
      for (int c = 0; c < r; c++) col[r] -= A[(size_t)r * n + c] * col[c];
    for (int r = n - 1; r >= 0; r--) {
      for (int c = r + 1; c < n; c++) col[r] -= A[(size_t)r * n + c] * col[c];
      col[r] /= A[(size_t)r * n + r];
    }
    for (int r = 0; r < n; r++) X[(size_t)r * n + j] = col[r];
  }
  return X;
}
}  // namespace

namespace {
// C[r][c] -= sum_k L[r][k] * B[k][c] for r, c < nd, with C = Hs (row stride ld), B = the rows nd .. nd + st of Hs, L = bli (nd x st)
__attribute__((target("avx2,fma"))) void schur_sub(double *Hs, int ld, int nd, const double *L, int st) {
  const double *B = Hs + (size_t)nd * ld;
  int r = 0;
  for (; r + 4 <= nd; r += 4) {
    double *c0 = Hs + (size_t)r * ld, *c1 = c0 + ld, *c2 = c1 + ld, *c3 = c2 + ld;
    const double *l0 = L + (size_t)r * st, *l1 = l0 + st, *l2 = l1 + st, *l3 = l2 + st;
    int c = 0;
    for (; c + 8 <= nd; c += 8) {
      __m256d a00 = _mm256_loadu_pd(c0 + c), a01 = _mm256_loadu_pd(c0 + c + 4), a10 = _mm256_loadu_pd(c1 + c), a11 = _mm256_loadu_pd(c1 + c + 4),
              a20 = _mm256_loadu_pd(c2 + c), a21 = _mm256_loadu_pd(c2 + c + 4), a30 = _mm256_loadu_pd(c3 + c), a31 = _mm256_loadu_pd(c3 + c + 4);
      for (int k = 0; k < st; k++) {
        const double *bk = B + (size_t)k * ld + c;
        const __m256d b0 = _mm256_loadu_pd(bk), b1 = _mm256_loadu_pd(bk + 4);
        __m256d l = _mm256_set1_pd(l0[k]);
        a00 = _mm256_fnmadd_pd(l, b0, a00); a01 = _mm256_fnmadd_pd(l, b1, a01);
        l = _mm256_set1_pd(l1[k]);
        a10 = _mm256_fnmadd_pd(l, b0, a10); a11 = _mm256_fnmadd_pd(l, b1, a11);
        l = _mm256_set1_pd(l2[k]);
        a20 = _mm256_fnmadd_pd(l, b0, a20); a21 = _mm256_fnmadd_pd(l, b1, a21);
        l = _mm256_set1_pd(l3[k]);
        a30 = _mm256_fnmadd_pd(l, b0, a30); a31 = _mm256_fnmadd_pd(l, b1, a31);
      }
      _mm256_storeu_pd(c0 + c, a00); _mm256_storeu_pd(c0 + c + 4, a01); _mm256_storeu_pd(c1 + c, a10); _mm256_storeu_pd(c1 + c + 4, a11);
      _mm256_storeu_pd(c2 + c, a20); _mm256_storeu_pd(c2 + c + 4, a21); _mm256_storeu_pd(c3 + c, a30); _mm256_storeu_pd(c3 + c + 4, a31);
    }
    for (; c < nd; c++) {
      double s0 = c0[c], s1 = c1[c], s2 = c2[c], s3 = c3[c];
      for (int k = 0; k < st; k++) {
        const double b = B[(size_t)k * ld + c];
        s0 -= l0[k] * b; s1 -= l1[k] * b; s2 -= l2[k] * b; s3 -= l3[k] * b;
      }
      c0[c] = s0; c1[c] = s1; c2[c] = s2; c3[c] = s3;
    }
  }
  for (; r < nd; r++) {
    double *cr = Hs + (size_t)r * ld;
    const double *lr = L + (size_t)r * st;
    for (int k = 0; k < st; k++) {
      const double l = lr[k];
      const double *bk = B + (size_t)k * ld;
      for (int c = 0; c < nd; c++) cr[c] -= l * bk[c];
    }
  }
}
}  // namespace

extern "C" int sosf_imu_marginalize_frame(const sosf_imu_settings *S, const sosf_imu_calib *C, int n, const sosf_imu_frame *F, int idx,
                                          const double *delta, const double *prior8, const double *delta_prior8, double margWeightFac,
                                          const double *HM_in, const double *bM_in, double *HM_out, double *bM_out) {
  if (!S || !C || !F || n < 2 || idx < 0 || idx >= n - 1 || !delta || !prior8 || !delta_prior8 || !HM_in || !bM_in || !HM_out || !bM_out)
    return SOS_ERR_ARG;
  const int dim = SOSF_IMU_DIM(n);
  static const bool tmg = getenv("SOS_TIMING_IMU") != nullptr;
  const double t0 = tmg ? now_us() : 0;
  // the IMU factors that tie the keyframe to its neighbours, linearised at the current delta, go into the prior.  HM_change lives in
  // the persistent holder of H_imu: it has entries only in the scale row / column and in the blocks of keyframes idx - 1 .. idx + 1
  Assembly &A = imu_holder(dim, n);
  add_frame(*S, *C, n, F, idx + 1, A);
  if (idx > 0) add_frame(*S, *C, n, F, idx, A);
  static thread_local std::vector<double> d2, bM, Hs, bs, sc, isc, blk, bli;
  d2.assign(dim, 0.0);
  for (int i = 0; i < CP; i++) d2[i] = delta[i];
  if (C->scale_trapped) d2[CP] = C->scale - C->scale_zero;
  for (int nb : {idx + 1, idx - 1}) {
    if (nb < 0) continue;
    for (int k = 0; k < 8; k++) d2[CP + 1 + 29 * nb + k] = delta[CP + 8 * nb + k];
    if (C->scale_trapped)
      for (int k = 0; k < 21; k++) d2[CP + 1 + 29 * nb + 8 + k] = F[nb].state_imu[k] - F[nb].state_imu_zero[k];
  }
  // bM + w (b_change - HM_change d2): the product runs over the columns where HM_change can have entries
  const int f0 = std::max(0, idx - 1), f1 = std::min(n - 1, idx + 1);
  const int c0 = CP + 1 + 29 * f0, c1 = CP + 1 + 29 * (f1 + 1);
  bM.resize(dim);
  for (int r = 0; r < dim; r++) {
    double hd = 0;
    if (r == CP || (r >= c0 && r < c1)) {
      const double *hr = &A.H.a[(size_t)r * dim];
      for (int c = 0; c <= CP; c++) hd += hr[c] * d2[c];
      for (int c = c0; c < c1; c++) hd += hr[c] * d2[c];
    }
    bM[r] = bM_in[r] + margWeightFac * (A.b[r] - hd);
  }
  // order of the states with the keyframe's block last; without a valid spline its 15 spline states are not eliminated
  // but simply dropped.  The order is three runs of consecutive indices: [0, io), [io + 29, dim), [io, io + step)
  const int args = CP + 1, io = args + 29 * idx, ndim = dim - 29;
  const bool constrained = idx > 0 && A.spline_valid[idx];
  const int step = constrained ? 29 : 14, cur = ndim + step;
  struct Run { int src, dst, len; };
  const Run runs[3] = {{0, 0, io}, {io + 29, io, dim - io - 29}, {io, ndim, step}};
  auto srcOf = [&](int r) { return r < io ? r : r < ndim ? r + 29 : io + (r - ndim); };
  Hs.resize((size_t)cur * cur);
  bs.resize(cur); sc.resize(cur); isc.resize(cur);
  for (int r = 0; r < cur; r++) {
    const int g = srcOf(r);
    double dg = HM_in[(size_t)g * dim + g] + margWeightFac * A.H(g, g);
    if (r >= ndim && r < ndim + 8) dg += prior8[r - ndim];  // the keyframe's pose prior joins here (marginalizeFrame :812-813)
    sc[r] = std::sqrt(std::fabs(dg) + 10);
    isc[r] = 1.0 / sc[r];
    double bv = bM[g];
    if (r >= ndim && r < ndim + 8) bv += prior8[r - ndim] * delta_prior8[r - ndim];
    bs[r] = isc[r] * bv;
  }
  for (int r = 0; r < cur; r++) {  // HM + w HM_change, permuted and Jacobi-scaled (:825-836)
    const int g = srcOf(r);
    const double *hm = HM_in + (size_t)g * dim, *hc = &A.H.a[(size_t)g * dim];
    double *dst = &Hs[(size_t)r * cur];
    const double ir = isc[r];
    const bool touched = g == CP || (g >= c0 && g < c1);  // rows where HM_change has entries anywhere
    for (const Run &u : runs) {
      const double *m = hm + u.src, *h = hc + u.src, *ic = &isc[u.dst];
      double *o = dst + u.dst;
      if (touched) {
        for (int i = 0; i < u.len; i++) o[i] = ir * (m[i] + margWeightFac * h[i]) * ic[i];
      } else {  // HM_change has only its scale column here (and that only if the row is in a touched block -- it is not): plain HM
        for (int i = 0; i < u.len; i++) o[i] = ir * m[i] * ic[i];
        if (CP >= u.src && CP < u.src + u.len) o[CP - u.src] = ir * (m[CP - u.src] + margWeightFac * h[CP - u.src]) * ic[CP - u.src];
      }
    }
    if (r >= ndim && r < ndim + 8) dst[r] = ir * (hm[g] + margWeightFac * hc[g] + prior8[r - ndim]) * ir;
  }
  clear_imu_blocks(A, n);
  const double t1 = tmg ? now_us() : 0;
  blk.resize((size_t)step * step);
  for (int r = 0; r < step; r++)
    for (int c = 0; c < step; c++) blk[(size_t)r * step + c] = Hs[(size_t)(ndim + r) * cur + ndim + c];
  const std::vector<double> hpi = inverse(blk, step);
  // bli = B^T hpi with B = the keyframe's rows of the scaled matrix (step x ndim, contiguous rows)
  bli.assign((size_t)ndim * step, 0.0);
  for (int k = 0; k < step; k++) {
    const double *bk = &Hs[(size_t)(ndim + k) * cur], *hk = &hpi[(size_t)k * step];
    for (int r = 0; r < ndim; r++) {
      const double v = bk[r];
      double *o = &bli[(size_t)r * step];
      for (int c = 0; c < step; c++) o[c] += v * hk[c];
    }
  }
  // Schur complement (:846-849): Hs[0..ndim)[0..ndim) -= bli * B, a 4 x 8 register tile over all eliminated states (fp64 prior
  // algebra feeding the next solves: fused multiply-adds, like the LDL^T)
  schur_sub(Hs.data(), cur, ndim, bli.data(), step);
  for (int r = 0; r < ndim; r++) {
    const double *lr = &bli[(size_t)r * step];
    double sb = 0;
    for (int k = 0; k < step; k++) sb += lr[k] * bs[ndim + k];
    bs[r] -= sb;
  }
  const double t2 = tmg ? now_us() : 0;
  // unscale and symmetrise (:852-857), tile by tile so that the transposed reads stay in cache
  const int TB = 32;
  for (int r0 = 0; r0 < ndim; r0 += TB)
    for (int q0 = 0; q0 < ndim; q0 += TB) {
      const int r1 = std::min(ndim, r0 + TB), q1 = std::min(ndim, q0 + TB);
      for (int r = r0; r < r1; r++)
        for (int c = q0; c < q1; c++)
          HM_out[(size_t)r * ndim + c] = 0.5 * (sc[r] * Hs[(size_t)r * cur + c] * sc[c] + sc[c] * Hs[(size_t)c * cur + r] * sc[r]);
    }
  for (int r = 0; r < ndim; r++) bM_out[r] = sc[r] * bs[r];
  if (tmg) fprintf(stderr, "[imu_marg] factors + permuted scaled copy %.0f us, inverse + Schur %.0f us, unscale %.0f us\n", t1 - t0, t2 - t1, now_us() - t2);
  return SOS_OK;
}

// ================================================================================================
// VIO front-end around the assembly (FS/HessianBlocks.cpp:225-429): the spline / bias / scale state of a frame before it
// enters the window.  Host fp64, called once per frame (propagateImuState, updateVel), once per run (initializeImu) or
// once per optimisation (tryTrapScale).
// ================================================================================================
namespace {
// FS/HessianBlocks.h:84-89, 93: the *_INVERSE constants are float quotients widened to double
const double kInvState[7] = {(double)(1.0f / 100.0f), (double)(1.0f / 1.0f), (double)(1.0f / 100.0f), (double)(1.0f / 1000.0f),
                             (double)(1.0f / 1000.0f), (double)(1.0f / 1000.0f), (double)(1.0f / 1000.0f)};
const double kScaleInv = (double)(1.0f / 200.0f);

// setImuStateScaled(scaled) + setImuStateZero: state_imu = INVERSE * scaled, state_imu_zero = state_imu (the cached
// per-sample Jacobians of setImuStateZero are recomputed by sosf_imu_hessian when needed)
void store_scaled_state(sosf_imu_frame &f, const double *scaled) {
  for (int s = 0; s < 7; s++)
    for (int i = 0; i < 3; i++) f.state_imu[3 * s + i] = kInvState[s] * scaled[3 * s + i];
  for (int i = 0; i < 21; i++) f.state_imu_zero[i] = f.state_imu[i];
}
void load_scaled_state(const sosf_imu_frame &f, double *scaled) {
  const double k[7] = {kBa, kBg, kSlRot, kSqTrans, kSqRot, kScTrans, kScRot};
  for (int s = 0; s < 7; s++)
    for (int i = 0; i < 3; i++) scaled[3 * s + i] = k[s] * f.state_imu[3 * s + i];
}

// Eigen's inverse() of a DYNAMIC-size expression: PartialPivLU (unblocked for this size) and a solve against the identity.
// A zero pivot is not divided by during the factorisation (Eigen/src/LU/PartialPivLU.h, partial_lu_impl::unblocked_lu)
// but IS divided by in the back-substitution -- which is what makes propagateImuState work: the first column of its
// accelerometer design matrix is identically zero, the 3 x 3 normal matrix is singular, and the rows of the "inverse"
// that the function goes on to use (1 and 2) come out as the inverse of the regular 2 x 2 block, with inf / NaN
// confined to row 0.
void lu_inverse(const double *Ain, int n, double *inv) {
  std::vector<double> lu(Ain, Ain + n * n);
  std::vector<int> perm(n);
  for (int i = 0; i < n; i++) perm[i] = i;
  for (int k = 0; k < n; k++) {
    int p = k;
    double best = std::fabs(lu[k * n + k]);
    for (int i = k + 1; i < n; i++)
      if (std::fabs(lu[i * n + k]) > best) { best = std::fabs(lu[i * n + k]); p = i; }
    if (best != 0.0) {
      if (p != k) {
        for (int j = 0; j < n; j++) std::swap(lu[k * n + j], lu[p * n + j]);
        std::swap(perm[k], perm[p]);
      }
      for (int i = k + 1; i < n; i++) lu[i * n + k] /= lu[k * n + k];
    }
    for (int i = k + 1; i < n; i++)
      for (int j = k + 1; j < n; j++) lu[i * n + j] -= lu[i * n + k] * lu[k * n + j];
  }
  for (int c = 0; c < n; c++) {
    std::vector<double> y(n);
    for (int i = 0; i < n; i++) y[i] = perm[i] == c ? 1.0 : 0.0;
    for (int i = 0; i < n; i++)
      for (int j = 0; j < i; j++) y[i] -= lu[i * n + j] * y[j];
    for (int i = n - 1; i >= 0; i--) {
      for (int j = i + 1; j < n; j++) y[i] -= lu[i * n + j] * y[j];
      y[i] /= lu[i * n + i];
    }
    for (int i = 0; i < n; i++) inv[i * n + c] = y[i];
  }
}
// x (3 x 3) = inverse(A^T A) * (A^T B) for an m x 3 design matrix A and m x 3 right-hand sides B
void normal_solve3(const std::vector<double> &A, const std::vector<double> &B, int m, double *x) {
  double M[9] = {0}, R[9] = {0}, Mi[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      double s = 0, q = 0;
      for (int i = 0; i < m; i++) {
        s += A[3 * i + r] * A[3 * i + c];
        q += A[3 * i + r] * B[3 * i + c];
      }
      M[3 * r + c] = s;
      R[3 * r + c] = q;
    }
  lu_inverse(M, 3, Mi);
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) x[3 * r + c] = Mi[3 * r] * R[c] + Mi[3 * r + 1] * R[3 + c] + Mi[3 * r + 2] * R[6 + c];
}
// fixed-size Mat33::inverse(): cofactors and one reciprocal of the determinant (Eigen/src/LU/InverseImpl.h, size 3)
void inverse3_cofactor(const double *m, double *r) {
  const double c00 = m[4] * m[8] - m[5] * m[7], c10 = m[5] * m[6] - m[3] * m[8], c20 = m[3] * m[7] - m[4] * m[6];
  const double det = c00 * m[0] + c10 * m[1] + c20 * m[2];
  const double id = 1.0 / det;
  r[0] = c00 * id; r[3] = c10 * id; r[6] = c20 * id;
  r[1] = (m[2] * m[7] - m[1] * m[8]) * id; r[4] = (m[0] * m[8] - m[2] * m[6]) * id; r[7] = (m[1] * m[6] - m[0] * m[7]) * id;
  r[2] = (m[1] * m[5] - m[2] * m[4]) * id; r[5] = (m[2] * m[3] - m[0] * m[5]) * id; r[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}
}  // namespace

extern "C" int sosf_imu_propagate_state(const sosf_imu_settings *S, const sosf_imu_calib *C, sosf_imu_frame *f, sosf_imu_shell *shell,
                                        const sosf_imu_shell *last_shell, const double *last_imu_bias6) {
  if (!S || !C || !f || !shell || !last_shell || !last_imu_bias6 || (f->n_imu > 0 && !f->imu)) return SOS_ERR_ARG;
  double sc[21];
  load_scaled_state(*f, sc);
  for (int i = 0; i < 6; i++) sc[i] = last_imu_bias6[i];  // imu_bias = last_imu_bias
  const int m = f->n_imu;
  double imu_ts = last_shell->timestamp;
  M3 R = M3::from(last_shell->camToWorld);
  const M3 RicT = M3::from(S->rot_imu_cam).T();
  const double scale_scaled = C->scale * kScale;
  std::vector<double> Aa(3 * (size_t)m, 0.0), ba(3 * (size_t)m), Ag(3 * (size_t)m), bg(3 * (size_t)m);
  for (int i = 0; i < m; i++) {
    const double *d = f->imu + 7 * (size_t)i;
    const double dt = d[0] - imu_ts;
    if (!(dt >= 0)) return SOS_ERR_ARG;  // assert(dt >= 0)
    imu_ts = d[0];
    const double t = d[0] - shell->timestamp;
    const V3 ua{{d[1] - sc[0], d[2] - sc[1], d[3] - sc[2]}}, ug{{d[4] - sc[3], d[5] - sc[4], d[6] - sc[5]}};
    R = R * so3_exp(V3{{ug[0] * dt, ug[1] * dt, ug[2] * dt}});  // integrate the gyroscope
    Aa[3 * i] = 0; Aa[3 * i + 1] = 2 * scale_scaled; Aa[3 * i + 2] = 6 * t * scale_scaled;
    const V3 aw = (R * RicT) * ua;
    for (int k = 0; k < 3; k++) ba[3 * i + k] = aw[k] - S->gravity[k];
    Ag[3 * i] = 1; Ag[3 * i + 1] = 2 * t; Ag[3 * i + 2] = 3 * t * t;
    const V3 gw = RicT * ug;
    for (int k = 0; k < 3; k++) bg[3 * i + k] = gw[k];
  }
  double xa[9], xg[9];
  normal_solve3(Aa, ba, m, xa);
  normal_solve3(Ag, bg, m, xg);
  for (int k = 0; k < 3; k++) {
    sc[9 + k] = xa[3 + k];   // spline_q.head(3) = xa.row(1)
    sc[15 + k] = xa[6 + k];  // spline_c.head(3) = xa.row(2)
    sc[6 + k] = xg[k];       // spline_l_rot = xg.row(0)
    sc[12 + k] = xg[3 + k];  // spline_q.tail(3)
    sc[18 + k] = xg[6 + k];  // spline_c.tail(3)
  }
  store_scaled_state(*f, sc);
  const double t = last_shell->timestamp - shell->timestamp;
  for (int k = 0; k < 3; k++) shell->velInWorld[k] = last_shell->velInWorld[k] - (2 * t * sc[9 + k] + 3 * t * t * sc[15 + k]);
  return SOS_OK;
}

extern "C" int sosf_imu_update_vel(const sosf_imu_frame *f, sosf_imu_shell *shell, const sosf_imu_shell *last_shell) {
  if (!f || !shell || !last_shell) return SOS_ERR_ARG;
  const double t = last_shell->timestamp - shell->timestamp;
  for (int k = 0; k < 3; k++) {
    const double q = kSqTrans * f->state_imu[9 + k];
    shell->velInWorld[k] = (last_shell->camToWorld[9 + k] - shell->camToWorld[9 + k]) / t - t * q - t * t * q;  // (sic) spline_q twice, :411
  }
  return SOS_OK;
}

extern "C" int sosf_imu_initialize(const sosf_imu_settings *S, sosf_imu_calib *C, sosf_imu_frame *F, sosf_imu_shell *sh, int *ok) {
  if (!S || !C || !F || !sh || !ok) return SOS_ERR_ARG;
  *ok = 0;
  const int B = 4;  // base_frame = frame_hessians.back()
  double A[9], b[18];
  const SE3 baseInv = SE3::from12(sh[B].camToWorld).inverse();
  for (int i = 0; i < 3; i++) {
    const int cur = i + 1;
    A[3 * i] = sh[cur].timestamp - sh[B].timestamp;
    A[3 * i + 1] = A[3 * i] * A[3 * i];
    A[3 * i + 2] = A[3 * i + 1] * A[3 * i];
    double lg[6];
    (baseInv * SE3::from12(sh[cur].camToWorld)).log(lg);
    for (int k = 0; k < 3; k++) {
      b[6 * i + k] = sh[cur].camToWorld[9 + k] - sh[B].camToWorld[9 + k];
      b[6 * i + 3 + k] = lg[3 + k];
    }
  }
  double Ai[9], x[18];
  inverse3_cofactor(A, Ai);
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 6; c++) x[6 * r + c] = Ai[3 * r] * b[c] + Ai[3 * r + 1] * b[6 + c] + Ai[3 * r + 2] * b[12 + c];
  const double *l0 = x, *q0 = x + 6, *c0 = x + 12;
  double sc[5][21];
  for (int fidx = 0; fidx < 5; fidx++) {
    load_scaled_state(F[fidx], sc[fidx]);
    const double t0 = sh[fidx].timestamp - sh[B].timestamp;
    for (int k = 0; k < 6; k++) {
      const double vel = l0[k] + 2 * q0[k] * t0 + 3 * c0[k] * t0 * t0;
      if (k < 3) sh[fidx].velInWorld[k] = vel;
      else sc[fidx][6 + (k - 3)] = vel;                        // spline_l_rot
      sc[fidx][9 + k] = q0[k] + 3 * c0[k] * t0;                 // spline_q
      sc[fidx][15 + k] = c0[k];                                 // spline_c
    }
  }
  size_t total = 0;
  for (int i = 2; i < 5; i++) total += (size_t)F[i].n_imu;
  // gyro bias: mean of measurement - prediction over the samples of frames 2..4
  const M3 Ric = M3::from(S->rot_imu_cam);
  const double *bs = sc[B];
  double gb[3] = {0, 0, 0};
  for (int i = 2; i < 5; i++)
    for (int j = 0; j < F[i].n_imu; j++) {
      const double *d = F[i].imu + 7 * (size_t)j;
      const double t = d[0] - sh[B].timestamp;
      V3 g;
      for (int k = 0; k < 3; k++) g[k] = bs[6 + k] + (2 * t * bs[12 + k] + 3 * t * t * bs[18 + k]);
      const V3 pred = Ric * g;
      for (int k = 0; k < 3; k++) gb[k] += d[4 + k] - pred[k];
    }
  for (int k = 0; k < 3; k++) gb[k] /= (double)total;
  for (int fidx = 0; fidx < 5; fidx++)
    for (int k = 0; k < 3; k++) sc[fidx][3 + k] = gb[k];
  // scale (accelerometer bias stays zero): one-parameter least squares of measured against predicted acceleration
  double scale = C->scale * kScale;
  if (!S->enable_scale_opt) {
    const M3 Rw2c = M3::from(F[B].camToWorld).T();  // PRE_worldToCam.rotationMatrix()
    double ab = 0, aa = 0;
    for (int i = 2; i < 5; i++)
      for (int j = 0; j < F[i].n_imu; j++) {
        const double *d = F[i].imu + 7 * (size_t)j;
        const double t = d[0] - sh[B].timestamp, t2 = t * t;
        V3 w, acc;
        for (int k = 0; k < 3; k++) {
          w[k] = t * bs[6 + k] + (t2 * bs[12 + k] + t * t2 * bs[18 + k]);
          acc[k] = 2 * bs[9 + k] + 6 * t * bs[15 + k];
        }
        const M3 rot_ti_w = (Ric * so3_exp(w).T()) * Rw2c;
        const V3 pred = rot_ti_w * acc, gr = rot_ti_w * V3{{S->gravity[0], S->gravity[1], S->gravity[2]}};
        for (int k = 0; k < 3; k++) {
          ab += pred[k] * (d[1 + k] - gr[k]);
          aa += pred[k] * pred[k];
        }
      }
    scale = ab / aa;
    C->scale = kScaleInv * scale;  // setScaleScaledZero
    C->scale_zero = C->scale;
  }
  if (scale < 0) return SOS_OK;  // "IMU initialization failed": the spline / velocity fields stay as computed, as in the reference
  for (int fidx = 0; fidx < 5; fidx++) {
    for (int k = 0; k < 3; k++) sc[fidx][k] = 0.0;
    store_scaled_state(F[fidx], sc[fidx]);
  }
  C->imu_initialized = 1;
  *ok = 1;
  return SOS_OK;
}

extern "C" int sosf_imu_try_trap_scale(sosf_imu_calib *C, double *scale_queue10, int32_t *scale_queue_i, double thres) {
  if (!C || !scale_queue10 || !scale_queue_i || *scale_queue_i < 0 || *scale_queue_i > 9) return SOS_ERR_ARG;
  C->scale_zero = C->scale;
  scale_queue10[*scale_queue_i] = C->scale;
  *scale_queue_i = (*scale_queue_i + 1) % 10;
  double mean = 0;
  for (int i = 0; i < 10; i++) mean += scale_queue10[i];
  mean /= 10.0;
  double sq = 0;
  for (int i = 0; i < 10; i++) sq += (scale_queue10[i] - mean) * (scale_queue10[i] - mean);
  const double var = 1.0 / 9.0 * kScale * kScale * sq;
  if (var < thres) {
    C->scale_trapped = 1;
    C->scale_zero = mean;
  }
  return SOS_OK;
}
