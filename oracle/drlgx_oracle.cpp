// drlgx CPU ORACLE — TEST INFRASTRUCTURE ONLY.
//
// A dependency-free, double-precision, single-threaded C++17 restatement of the reference's
// exploration-environment belief step (simulate -> SLAM belief -> occupancy/virtual-map rebuild ->
// utility / look-ahead reward -> graph export).  It exists to CHECK the HIP path: only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
// (drl_graph_exploration_amd/) never links, imports or calls anything in oracle/.
//
// PARITY STATUS: "parity weakly pinned".  The reference's own hot path cannot be built here
// (gtsam fork / Eigen / boost absent), and the reference ships no unit tests.  The
// oracle is pinned by (i) closed-form known answers derived from the reference source (occupancy
// ladder, covariance-intersection identities, line-planner lists, mt19937 / normal_distribution
// streams, start poses) and (ii) the per-step triples of data/test_result/40_DQN_GCN.csv replayed
// end-to-end (tests/test_oracle_csv_pin.py; fixtures in tests/golden/).  iSAM2 is restated as a
// dense Gauss-Newton step with iSAM2's linearisation-point policy; the
// differences (wildfire threshold 1e-3 on back-substitution, elimination order) are documented in
// DESIGN.md.
//
// Every function cites the reference file:line it follows (paths relative to /root/reference).
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <random>
#include <unordered_map>
#include <vector>

namespace orc {

// ----------------------------------------------------------------------------------------------
// Geometry: gtsam Pose2/Rot2/Point2 semantics (gtsam 4.0 geometry/Pose2.cpp,
// Rot2.cpp — third-party, absent; restated from the published algorithm).
// A Pose2 stores (x, y, cos, sin); theta() = atan2(sin, cos).
// ----------------------------------------------------------------------------------------------
struct P2 {
  double x = 0, y = 0;
};
struct Pose {
  double x = 0, y = 0, c = 1, s = 0;
};

static inline void rot_from_cos_sin(double c, double s, double &oc, double &os) {
  // Rot2::fromCosSin: renormalise only when off by more than 1e-9
  if (std::fabs(c * c + s * s - 1.0) > 1e-9) {
    double n = std::sqrt(c * c + s * s);
    c /= n;
    s /= n;
  }
  oc = c;
  os = s;
}
static inline Pose make_pose(double x, double y, double th) { return Pose{x, y, std::cos(th), std::sin(th)}; }
static inline double theta_of(const Pose &p) { return std::atan2(p.s, p.c); }
static inline double wrap_theta(double th) { return std::atan2(std::sin(th), std::cos(th)); }  // Rot2(th).theta()

static inline Pose compose(const Pose &a, const Pose &b) {  // Pose2::operator*
  Pose r;
  rot_from_cos_sin(a.c * b.c - a.s * b.s, a.s * b.c + a.c * b.s, r.c, r.s);
  r.x = a.x + (a.c * b.x - a.s * b.y);
  r.y = a.y + (a.s * b.x + a.c * b.y);
  return r;
}
// between(p1,p2) = p1^-1 * p2 with H1 = -Ad(between^-1) (Pose2::between inlined form), H2 = I
static inline Pose between(const Pose &p1, const Pose &p2, double *H1 /*3x3 row-major or null*/) {
  double c = p1.c * p2.c + p1.s * p2.s, s = -p1.s * p2.c + p1.c * p2.s;
  Pose r;
  rot_from_cos_sin(c, s, r.c, r.s);
  double dx = p2.x - p1.x, dy = p2.y - p1.y;
  r.x = p1.c * dx + p1.s * dy;
  r.y = -p1.s * dx + p1.c * dy;
  if (H1) {
    double dt1 = -p2.s * dx + p2.c * dy;
    double dt2 = -p2.c * dx - p2.s * dy;
    H1[0] = -r.c; H1[1] = -r.s; H1[2] = dt1;
    H1[3] = r.s;  H1[4] = -r.c; H1[5] = dt2;
    H1[6] = 0;    H1[7] = 0;    H1[8] = -1;
  }
  return r;
}
static inline P2 transform_to(const Pose &p, const P2 &pt) {  // R^T (pt - t)
  double dx = pt.x - p.x, dy = pt.y - p.y;
  return P2{p.c * dx + p.s * dy, -p.s * dx + p.c * dy};
}
static inline P2 transform_from(const Pose &p, const P2 &q) {  // R q + t
  return P2{p.c * q.x - p.s * q.y + p.x, p.s * q.x + p.c * q.y + p.y};
}
// Pose2::bearing(point) -> Rot2::relativeBearing(d).theta(); Jacobians as in App. A.1
static inline double bearing_of(const Pose &p, const P2 &pt, double *Hx /*1x3*/, double *Hl /*1x2*/) {
  P2 d = transform_to(p, pt);
  double d2 = d.x * d.x + d.y * d.y, n = std::sqrt(d2);
  if (std::fabs(n) > 1e-5) {
    if (Hx) {
      double a = -d.y / d2, b = d.x / d2;  // D_result_d
      // D1 = [[-1,0,d.y],[0,-1,-d.x]]
      Hx[0] = a * -1.0;
      Hx[1] = b * -1.0;
      Hx[2] = a * d.y + b * -d.x;
      // D2 = R^T = [[c,s],[-s,c]]
      Hl[0] = a * p.c + b * -p.s;
      Hl[1] = a * p.s + b * p.c;
    }
    double c, s;
    rot_from_cos_sin(d.x / n, d.y / n, c, s);
    return std::atan2(s, c);
  }
  if (Hx) {
    Hx[0] = Hx[1] = Hx[2] = 0;
    Hl[0] = Hl[1] = 0;
  }
  return 0.0;
}
static inline double range_of(const Pose &p, const P2 &pt, double *Hx /*1x3*/, double *Hl /*1x2*/) {
  double dx = pt.x - p.x, dy = pt.y - p.y;
  double r = std::sqrt(dx * dx + dy * dy);
  if (Hx) {
    double ux = dx / r, uy = dy / r;
    // D_d_pose = [[-c, s, 0], [-s, -c, 0]]
    Hx[0] = ux * -p.c + uy * -p.s;
    Hx[1] = ux * p.s + uy * -p.c;
    Hx[2] = 0;
    Hl[0] = ux;
    Hl[1] = uy;
  }
  return r;
}

// ----------------------------------------------------------------------------------------------
// Small dense SPD helpers (Eigen LLT / inverse semantics, include/em_exploration/Utils.h:29-33)
// ----------------------------------------------------------------------------------------------
static inline double det2(const double *m) { return m[0] * m[3] - m[1] * m[2]; }
static inline void inv2_closed(const double *m, double *o) {  // Eigen 2x2 .inverse()
  double id = 1.0 / det2(m);
  o[0] = m[3] * id; o[1] = -m[1] * id; o[2] = -m[2] * id; o[3] = m[0] * id;
}
static inline void inv2_llt(const double *m, double *o) {  // inverse<2>: m.llt().solve(I)
  double l00 = std::sqrt(m[0]), l10 = m[2] / l00, l11 = std::sqrt(m[3] - l10 * l10);
  for (int col = 0; col < 2; ++col) {
    double b0 = col == 0 ? 1.0 : 0.0, b1 = col == 1 ? 1.0 : 0.0;
    double y0 = b0 / l00, y1 = (b1 - l10 * y0) / l11;
    double x1 = y1 / l11, x0 = (y0 - l10 * x1) / l00;
    o[0 * 2 + col] = x0;
    o[1 * 2 + col] = x1;
  }
}
static inline double det3(const double *m) {
  return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}
struct LLT3 {
  double l00, l10, l11, l20, l21, l22;
  explicit LLT3(const double *m) {
    l00 = std::sqrt(m[0]);
    l10 = m[3] / l00;
    l20 = m[6] / l00;
    l11 = std::sqrt(m[4] - l10 * l10);
    l21 = (m[7] - l20 * l10) / l11;
    l22 = std::sqrt(m[8] - l20 * l20 - l21 * l21);
  }
  void solve(const double *b, double *x) const {
    double y0 = b[0] / l00, y1 = (b[1] - l10 * y0) / l11, y2 = (b[2] - l20 * y0 - l21 * y1) / l22;
    x[2] = y2 / l22;
    x[1] = (y1 - l21 * x[2]) / l11;
    x[0] = (y0 - l10 * x[1] - l20 * x[2]) / l00;
  }
};
static inline void inv3_llt(const double *m, double *o) {
  LLT3 llt(m);
  for (int col = 0; col < 3; ++col) {
    double b[3] = {0, 0, 0}, x[3];
    b[col] = 1.0;
    llt.solve(b, x);
    for (int r = 0; r < 3; ++r) o[r * 3 + col] = x[r];
  }
}

// ----------------------------------------------------------------------------------------------
// RNG (include/em_exploration/RNG.h:47-126): mt19937 + uniform_real_distribution<>(0,1) +
// ONE normal_distribution<>(0,1) object (its cached second variate is shared by all normal()).
// libstdc++'s own types are used so the streams are the reference's by construction.
// ----------------------------------------------------------------------------------------------
struct RNG {
  std::mt19937 gen;
  std::uniform_real_distribution<> uni{0.0, 1.0};
  std::normal_distribution<> nrm{0.0, 1.0};
  explicit RNG(uint32_t seed = 0) : gen(seed) {}
  double uniformReal(double lo, double hi) { return (hi - lo) * uni(gen) + lo; }  // RNG.h:68-71
  double normal(double m, double sd) { return nrm(gen) * sd + m; }                // RNG.h:87-96
};

// ----------------------------------------------------------------------------------------------
// Parameters (scripts/envs/exploration_env.ini; pyss2d.py:10-55; pyplanner2d.py:24-54)
// ----------------------------------------------------------------------------------------------
extern "C" struct orc_config {
  // sensor (radians / metres) — setters wrap angles through Rot2(x).theta() (Simulation2D.h:52-55)
  double bearing_noise, range_noise, min_bearing, max_bearing, min_range, max_range;
  // control (Simulation2D.h:149-152)
  double translation_noise, rotation_noise;
  // environment box (unpadded) and map box (padded by ext=20, pyss2d.py:48-55)
  double env_min_x, env_max_x, env_min_y, env_max_y, safe_distance;
  double map_min_x, map_max_x, map_min_y, map_max_y;
  // virtual map
  double resolution, sigma0;
  int num_samples;
  // simulator
  double sigma_x0, sigma_y0, sigma_theta0;  // sigma_theta0 in radians
  int num_landmarks;
  // planner
  double angle_weight, distance_weight0, distance_weight1, occupancy_threshold, max_edge_length;
  int algorithm;  // 0 = EM_AOPT, 1 = EM_DOPT (Planner2D.h OptimizationAlgorithm)
};

struct Measurement {
  unsigned key;
  double bearing, range;
};

// ----------------------------------------------------------------------------------------------
// Simulator2D (src/em_exploration/Simulator2D.cpp:421-527): ground-truth world + noisy
// move/measure.  Three RNGs seeded with the same seed (Simulator2D.cpp:436-443).
// ----------------------------------------------------------------------------------------------
struct Simulator {
  orc_config cfg;
  RNG sensor_rng, control_rng, rng;
  Pose vehicle;
  std::unordered_map<unsigned, P2> landmarks;  // Environment::landmarks_ (Simulation2D.h:350)
  std::vector<Pose> trajectory;

  Simulator(const orc_config &c, uint32_t seed) : cfg(c), sensor_rng(seed), control_rng(seed), rng(seed) {}

  void initializeVehicle(const Pose &p) {  // Simulator2D.cpp:477-480
    vehicle = p;
    trajectory.push_back(p);
  }
  // Simulator2D::addLandmarks (Simulator2D.cpp:445-464): the listed landmarks take the keys 0 .. n_fixed - 1, the random ones follow
  void addLandmarks(unsigned num, const double *fixed_xy = nullptr, unsigned n_fixed = 0) {
    landmarks.clear();  // environment_ = Environment(params)  -- NOTE also clears the trajectory
    trajectory.clear();
    for (unsigned i = 0; i < n_fixed; ++i) landmarks.emplace(i, P2{fixed_xy[2 * i], fixed_xy[2 * i + 1]});
    for (unsigned i = n_fixed; i < num;) {
      double x = rng.uniformReal(cfg.env_min_x, cfg.env_max_x);
      double y = rng.uniformReal(cfg.env_min_y, cfg.env_max_y);
      double dx = x - vehicle.x, dy = y - vehicle.y;
      if (std::sqrt(dx * dx + dy * dy) < 2.0) continue;
      landmarks.emplace(i, P2{x, y});
      i++;
    }
  }
  // SimpleControlModel::evolve (Simulator2D.cpp:161-182) + Simulator2D::move (:491-503)
  void move(const Pose &odom) {
    double xn = control_rng.normal(0.0, cfg.translation_noise);
    double yn = control_rng.normal(0.0, cfg.translation_noise);
    double tn = control_rng.normal(0.0, cfg.rotation_noise);
    Pose n = make_pose(xn, yn, tn);
    Pose np = compose(vehicle, odom);
    vehicle = compose(np, n);
    trajectory.push_back(vehicle);
  }
  // Simulator2D::measure (Simulator2D.cpp:505-527): neighbours in unordered_map iteration order
  // (buildLandmarkKDTree :331-344, queryRadiusNeighbors Distance.cpp:78-97); noise drawn BEFORE
  // the validity check (BearingRangeSensorModel::measure :113-132, check :100-105).
  std::vector<Measurement> measure() {
    std::vector<Measurement> ms;
    for (const auto &it : landmarks) {
      double dx = vehicle.x - it.second.x, dy = vehicle.y - it.second.y;
      if (!(std::sqrt(dx * dx + dy * dy) < cfg.max_range)) continue;
      double bn = sensor_rng.normal(0.0, cfg.bearing_noise);
      double rn = sensor_rng.normal(0.0, cfg.range_noise);
      double b = bearing_of(vehicle, it.second, nullptr, nullptr) + bn;
      double r = range_of(vehicle, it.second, nullptr, nullptr) + rn;
      if (b < cfg.max_bearing && b > cfg.min_bearing && r < cfg.max_range && r > cfg.min_range)
        ms.push_back(Measurement{it.first, b, r});
    }
    return ms;
  }
};

// ----------------------------------------------------------------------------------------------
// iSAM2 restated densely (gtsam nonlinear/ISAM2.cpp — third-party, absent).
// State: linearisation point theta, delta, all factors, update counter.  One update() = add
// factors/variables, (every relinearizeSkip-th call) fold delta into theta for variables whose
// |delta|_inf >= relinearizeThreshold, linearise every factor at theta, solve the normal
// equations ONCE for delta.  estimate = theta (+) delta.  Marginals = blocks of (J^T J)^-1 at theta.
// ----------------------------------------------------------------------------------------------
struct OdoFactor {
  Pose measured;
  double sig[3];
};
struct MeasFactor {
  int pose;
  int lm;  // landmark slot
  double bearing, range;
};

// development knobs (scripts/dbg_*.py): relinearisation threshold / skip tried against the reference CSV
static double g_relin_thr = 0.1;
static int g_relin_skip = 10;
static int g_relin_mode = 0;  // 0: variables with |delta|_inf >= threshold; 1: all variables if any one crosses it

struct Isam {
  int count = 0;
  std::vector<Pose> th_pose;
  std::vector<double> d_pose;  // 3 per pose
  std::vector<P2> th_lm;
  std::vector<double> d_lm;  // 2 per landmark
  // factors
  bool has_prior = false;
  Pose prior_pose;
  double prior_info[9];
  std::vector<OdoFactor> odo;  // odo[i] links pose i and i+1
  std::vector<MeasFactor> meas;
  double sig_b = 0, sig_r = 0;
  // outputs of the last solve
  std::vector<double> cov;   // dense (2L+3P)^2, order [landmarks, poses]; only the diagonal blocks are filled
  std::vector<double> Linv;  // inverse of the Cholesky factor of the last solve: Sigma = Linv^T Linv
  int n = 0;

  int P() const { return (int)th_pose.size(); }
  int L() const { return (int)th_lm.size(); }

  Pose est_pose(int i) const {  // Pose2 retract: p * Pose2(v) (fast chart)
    return compose(th_pose[i], make_pose(d_pose[3 * i], d_pose[3 * i + 1], d_pose[3 * i + 2]));
  }
  P2 est_lm(int j) const { return P2{th_lm[j].x + d_lm[2 * j], th_lm[j].y + d_lm[2 * j + 1]}; }

  // new variables/factors are appended by the caller before update()
  void update(int n_old_pose, int n_old_lm) {
    count++;
    const bool relin = (count % g_relin_skip == 0);  // relinearizeSkip = 10 (offsets 1,2 and thresholds
    // 0.05/0.01/0.001 were tried against the reference CSV pins and track it worse — DESIGN.md)
    double relin_thr = g_relin_thr;  // relinearizeThreshold = 0.1
    if (relin && g_relin_mode == 1) {
      double mx = 0;
      for (int i = 0; i < 3 * n_old_pose; ++i) mx = std::max(mx, std::fabs(d_pose[i]));
      for (int j = 0; j < 2 * n_old_lm; ++j) mx = std::max(mx, std::fabs(d_lm[j]));
      if (mx >= relin_thr) relin_thr = -1.0;
    }
    if (relin) {
      for (int i = 0; i < n_old_pose; ++i) {
        double m = std::max({std::fabs(d_pose[3 * i]), std::fabs(d_pose[3 * i + 1]), std::fabs(d_pose[3 * i + 2])});
        if (m >= relin_thr) {  // relinearizeThreshold = 0.1
          th_pose[i] = est_pose(i);
          d_pose[3 * i] = d_pose[3 * i + 1] = d_pose[3 * i + 2] = 0;
        }
      }
      for (int j = 0; j < n_old_lm; ++j) {
        double m = std::max(std::fabs(d_lm[2 * j]), std::fabs(d_lm[2 * j + 1]));
        if (m >= relin_thr) {
          th_lm[j] = est_lm(j);
          d_lm[2 * j] = d_lm[2 * j + 1] = 0;
        }
      }
    }
    solve();
  }

  // Linearise all factors at theta; Lambda delta = eta; cov = Lambda^-1.
  void solve() {
    const int Pn = P(), Ln = L();
    n = 2 * Ln + 3 * Pn;
    std::vector<double> A((size_t)n * n, 0.0), b(n, 0.0);
    auto pidx = [&](int i) { return 2 * Ln + 3 * i; };
    auto lidx = [&](int j) { return 2 * j; };
    auto addBlock = [&](int r0, int rn, int c0, int cn, const double *M) {
      for (int r = 0; r < rn; ++r)
        for (int c = 0; c < cn; ++c) A[(size_t)(r0 + r) * n + (c0 + c)] += M[r * cn + c];
    };
    // prior (SLAM2D.cpp:44-57): e = Local(prior, x0), J = Hlocal = diag(R_h^T, 1), W = information
    if (has_prior) {
      Pose h = between(prior_pose, th_pose[0], nullptr);
      double e[3] = {h.x, h.y, theta_of(h)};
      double J[9] = {h.c, h.s, 0, -h.s, h.c, 0, 0, 0, 1};
      double WJ[9], JtWJ[9], We[3], g[3];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
          double s = 0;
          for (int k = 0; k < 3; ++k) s += prior_info[r * 3 + k] * J[k * 3 + c];
          WJ[r * 3 + c] = s;
        }
      for (int r = 0; r < 3; ++r) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += prior_info[r * 3 + k] * e[k];
        We[r] = s;
      }
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) {
          double s = 0;
          for (int k = 0; k < 3; ++k) s += J[k * 3 + r] * WJ[k * 3 + c];
          JtWJ[r * 3 + c] = s;
        }
        double s = 0;
        for (int k = 0; k < 3; ++k) s += J[k * 3 + r] * We[k];
        g[r] = s;
      }
      addBlock(pidx(0), 3, pidx(0), 3, JtWJ);
      for (int r = 0; r < 3; ++r) b[pidx(0) + r] -= g[r];
    }
    // odometry (SLAM2D.cpp:59-89): e = Local(measured, between(x1,x2)); J2 = Hlocal, J1 = Hlocal*H1
    for (int i = 0; i < (int)odo.size(); ++i) {
      double H1[9];
      Pose hx = between(th_pose[i], th_pose[i + 1], H1);
      Pose h = between(odo[i].measured, hx, nullptr);
      double e[3] = {h.x, h.y, theta_of(h)};
      double Hl[9] = {h.c, h.s, 0, -h.s, h.c, 0, 0, 0, 1};
      double J1[9], J2[9];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
          double s = 0;
          for (int k = 0; k < 3; ++k) s += Hl[r * 3 + k] * H1[k * 3 + c];
          J1[r * 3 + c] = s;
          J2[r * 3 + c] = Hl[r * 3 + c];
        }
      double w[3] = {1.0 / (odo[i].sig[0] * odo[i].sig[0]), 1.0 / (odo[i].sig[1] * odo[i].sig[1]),
                     1.0 / (odo[i].sig[2] * odo[i].sig[2])};
      double B11[9], B12[9], B21[9], B22[9], g1[3], g2[3];
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) {
          double s11 = 0, s12 = 0, s22 = 0;
          for (int k = 0; k < 3; ++k) {
            s11 += J1[k * 3 + r] * w[k] * J1[k * 3 + c];
            s12 += J1[k * 3 + r] * w[k] * J2[k * 3 + c];
            s22 += J2[k * 3 + r] * w[k] * J2[k * 3 + c];
          }
          B11[r * 3 + c] = s11;
          B12[r * 3 + c] = s12;
          B21[c * 3 + r] = s12;
          B22[r * 3 + c] = s22;
        }
        double s1 = 0, s2 = 0;
        for (int k = 0; k < 3; ++k) {
          s1 += J1[k * 3 + r] * w[k] * e[k];
          s2 += J2[k * 3 + r] * w[k] * e[k];
        }
        g1[r] = s1;
        g2[r] = s2;
      }
      addBlock(pidx(i), 3, pidx(i), 3, B11);
      addBlock(pidx(i), 3, pidx(i + 1), 3, B12);
      addBlock(pidx(i + 1), 3, pidx(i), 3, B21);
      addBlock(pidx(i + 1), 3, pidx(i + 1), 3, B22);
      for (int r = 0; r < 3; ++r) {
        b[pidx(i) + r] -= g1[r];
        b[pidx(i + 1) + r] -= g2[r];
      }
    }
    // bearing-range (SLAM2D.cpp:91-124): e = [wrap(b_pred - b_meas), r_pred - r_meas]
    const double wb = 1.0 / (sig_b * sig_b), wr = 1.0 / (sig_r * sig_r);
    for (const MeasFactor &f : meas) {
      double Hbx[3], Hbl[2], Hrx[3], Hrl[2];
      const Pose &p = th_pose[f.pose];
      const P2 &l = th_lm[f.lm];
      double bp = bearing_of(p, l, Hbx, Hbl);
      double rp = range_of(p, l, Hrx, Hrl);
      // Rot2 Local(measured, predicted) = theta of (measured^-1 * predicted)
      double cm = std::cos(f.bearing), sm = std::sin(f.bearing), cp = std::cos(bp), sp = std::sin(bp);
      double eb = std::atan2(-sm * cp + cm * sp, cm * cp + sm * sp);
      double er = rp - f.range;
      double Jx[6] = {Hbx[0], Hbx[1], Hbx[2], Hrx[0], Hrx[1], Hrx[2]};
      double Jl[4] = {Hbl[0], Hbl[1], Hrl[0], Hrl[1]};
      double w[2] = {wb, wr}, e[2] = {eb, er};
      double Bxx[9], Bxl[6], Blx[6], Bll[4], gx[3], gl[2];
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) Bxx[r * 3 + c] = Jx[r] * w[0] * Jx[c] + Jx[3 + r] * w[1] * Jx[3 + c];
        for (int c = 0; c < 2; ++c) {
          double v = Jx[r] * w[0] * Jl[c] + Jx[3 + r] * w[1] * Jl[2 + c];
          Bxl[r * 2 + c] = v;
          Blx[c * 3 + r] = v;
        }
        gx[r] = Jx[r] * w[0] * e[0] + Jx[3 + r] * w[1] * e[1];
      }
      for (int r = 0; r < 2; ++r) {
        for (int c = 0; c < 2; ++c) Bll[r * 2 + c] = Jl[r] * w[0] * Jl[c] + Jl[2 + r] * w[1] * Jl[2 + c];
        gl[r] = Jl[r] * w[0] * e[0] + Jl[2 + r] * w[1] * e[1];
      }
      addBlock(pidx(f.pose), 3, pidx(f.pose), 3, Bxx);
      addBlock(pidx(f.pose), 3, lidx(f.lm), 2, Bxl);
      addBlock(lidx(f.lm), 2, pidx(f.pose), 3, Blx);
      addBlock(lidx(f.lm), 2, lidx(f.lm), 2, Bll);
      for (int r = 0; r < 3; ++r) b[pidx(f.pose) + r] -= gx[r];
      for (int r = 0; r < 2; ++r) b[lidx(f.lm) + r] -= gl[r];
    }
    // dense Cholesky A = L L^T (lower, in place)
    std::vector<double> Lm = A;
    for (int j = 0; j < n; ++j) {
      double d = Lm[(size_t)j * n + j];
      for (int k = 0; k < j; ++k) d -= Lm[(size_t)j * n + k] * Lm[(size_t)j * n + k];
      d = std::sqrt(d);
      Lm[(size_t)j * n + j] = d;
      for (int i = j + 1; i < n; ++i) {
        double s = Lm[(size_t)i * n + j];
        const double *ri = &Lm[(size_t)i * n], *rj = &Lm[(size_t)j * n];
        for (int k = 0; k < j; ++k) s -= ri[k] * rj[k];
        Lm[(size_t)i * n + j] = s / d;
      }
    }
    // delta = A^-1 b
    std::vector<double> y(n), x(n);
    for (int i = 0; i < n; ++i) {
      double s = b[i];
      for (int k = 0; k < i; ++k) s -= Lm[(size_t)i * n + k] * y[k];
      y[i] = s / Lm[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
      double s = y[i];
      for (int k = i + 1; k < n; ++k) s -= Lm[(size_t)k * n + i] * x[k];
      x[i] = s / Lm[(size_t)i * n + i];
    }
    for (int j = 0; j < Ln; ++j) {
      d_lm[2 * j] = x[lidx(j)];
      d_lm[2 * j + 1] = x[lidx(j) + 1];
    }
    for (int i = 0; i < Pn; ++i)
      for (int r = 0; r < 3; ++r) d_pose[3 * i + r] = x[pidx(i) + r];
    // cov = A^-1 via Linv: Linv lower, cov = Linv^T Linv
    std::vector<double> &Li = Linv;
    Li.assign((size_t)n * n, 0.0);
    for (int c = 0; c < n; ++c) {
      Li[(size_t)c * n + c] = 1.0 / Lm[(size_t)c * n + c];
      for (int i = c + 1; i < n; ++i) {
        double s = 0;
        for (int k = c; k < i; ++k) s -= Lm[(size_t)i * n + k] * Li[(size_t)k * n + c];
        Li[(size_t)i * n + c] = s / Lm[(size_t)i * n + i];
      }
    }
    cov.assign((size_t)n * n, 0.0);
    // only the block diagonal (3x3 pose blocks, 2x2 landmark blocks) plus full matrix for tests of
    // small systems is needed; compute the full symmetric product for n <= 400, blocks otherwise.
    auto dot_cols = [&](int a, int c) {
      double s = 0;
      for (int k = std::max(a, c); k < n; ++k) s += Li[(size_t)k * n + a] * Li[(size_t)k * n + c];
      return s;
    };
    for (int j = 0; j < Ln; ++j)
      for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 2; ++c) cov[(size_t)(lidx(j) + r) * n + lidx(j) + c] = dot_cols(lidx(j) + r, lidx(j) + c);
    for (int i = 0; i < Pn; ++i)
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) cov[(size_t)(pidx(i) + r) * n + pidx(i) + c] = dot_cols(pidx(i) + r, pidx(i) + c);
  }
  void pose_cov(int i, double *o) const {
    int b = 2 * L() + 3 * i;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) o[r * 3 + c] = cov[(size_t)(b + r) * n + b + c];
  }
  void lm_cov(int j, double *o) const {
    int b = 2 * j;
    for (int r = 0; r < 2; ++r)
      for (int c = 0; c < 2; ++c) o[r * 2 + c] = cov[(size_t)(b + r) * n + b + c];
  }
};

// ----------------------------------------------------------------------------------------------
// SLAM2D (src/em_exploration/SLAM2D.cpp): factor bookkeeping + Map (estimates + information).
// ----------------------------------------------------------------------------------------------
struct MapPose {
  Pose pose;
  double info[9];
  bool core;
};
struct MapLm {
  unsigned key;
  P2 point;
  double info[4];
};

struct Slam {
  orc_config cfg;
  Isam isam;       // isam_
  Isam copy_isam;  // copy_isam_ (look-ahead)
  bool use_copy = false;
  unsigned step = 0;
  std::map<unsigned, int> lm_slot;  // GT key -> landmark slot (order of first sighting)
  std::vector<unsigned> slot_key;
  // result_ (estimate after the last optimise)
  std::vector<Pose> res_pose;
  std::vector<P2> res_lm;
  // pending (graph_ / initial_estimate_)
  int pend_old_pose = 0, pend_old_lm = 0;
  bool pending = false;
  // Map
  std::vector<MapPose> traj;
  std::vector<MapLm> lms;  // in slot order (iteration-order-independent uses only)
  // marginal covariance traces cached for adjacency_degree_get
  std::vector<double> pose_cov_tr, lm_cov_tr;

  Isam &cur() { return use_copy ? copy_isam : isam; }

  void begin_pending() {
    if (!pending) {
      pend_old_pose = cur().P();
      pend_old_lm = cur().L();
      pending = true;
    }
  }
  // SLAM2D::addPrior(VehicleBeliefState) (SLAM2D.cpp:44-57)
  void addPrior(const Pose &pose, const double *info) {
    begin_pending();
    Isam &I = cur();
    I.has_prior = true;
    I.prior_pose = pose;
    std::memcpy(I.prior_info, info, sizeof(double) * 9);
    I.sig_b = cfg.bearing_noise;
    I.sig_r = cfg.range_noise;
    I.th_pose.push_back(pose);
    I.d_pose.insert(I.d_pose.end(), {0, 0, 0});
    step++;
  }
  // SLAM2D::addOdometry (SLAM2D.cpp:70-89): initial guess p1*odom with p1 from result_ if present
  void addOdometry(const Pose &odom, const double *sig) {
    begin_pending();
    Isam &I = cur();
    OdoFactor f;
    f.measured = odom;
    f.sig[0] = sig[0]; f.sig[1] = sig[1]; f.sig[2] = sig[2];
    I.odo.push_back(f);
    unsigned x1 = step - 1;
    Pose p1 = (x1 < res_pose.size()) ? res_pose[x1] : I.th_pose[x1];
    I.th_pose.push_back(compose(p1, odom));
    I.d_pose.insert(I.d_pose.end(), {0, 0, 0});
    step++;
  }
  // SLAM2D::addMeasurement (SLAM2D.cpp:103-124)
  void addMeasurement(unsigned key, double bearing, double range) {
    begin_pending();
    Isam &I = cur();
    unsigned x = step - 1;
    auto it = lm_slot.find(key);
    int slot;
    if (it == lm_slot.end()) {
      Pose origin = (x < res_pose.size()) ? res_pose[x] : I.th_pose[x];
      P2 g = transform_from(origin, P2{range * std::cos(bearing), range * std::sin(bearing)});  // Simulator2D.cpp:95-98
      slot = (int)slot_key.size();
      lm_slot[key] = slot;
      slot_key.push_back(key);
      I.th_lm.push_back(g);
      I.d_lm.insert(I.d_lm.end(), {0, 0});
      MapLm m;
      m.key = key;
      m.point = g;
      m.info[0] = 1; m.info[1] = 0; m.info[2] = 0; m.info[3] = 1;
      lms.push_back(m);
    } else
      slot = it->second;
    I.meas.push_back(MeasFactor{(int)x, slot, bearing, range});
  }
  // SLAM2D::optimize / copy_optimize (SLAM2D.cpp:374-430 / 432-488)
  void optimize() {
    if (!pending) return;
    Isam &I = cur();
    I.update(pend_old_pose, pend_old_lm);
    pending = false;
    res_pose.resize(I.P());
    res_lm.resize(I.L());
    pose_cov_tr.resize(I.P());
    lm_cov_tr.resize(I.L());
    for (int i = 0; i < I.P(); ++i) {
      res_pose[i] = I.est_pose(i);
      double cov[9];
      I.pose_cov(i, cov);
      pose_cov_tr[i] = cov[0] + cov[4] + cov[8];
      MapPose mp;
      mp.pose = res_pose[i];
      inv3_llt(cov, mp.info);  // VehicleBeliefState(pose, inverse(covariance))
      if (i < (int)traj.size()) {
        mp.core = traj[i].core;
        traj[i] = mp;
      } else {
        mp.core = (i == (int)step - 1);
        traj.push_back(mp);
      }
    }
    for (int j = 0; j < I.L(); ++j) {
      res_lm[j] = I.est_lm(j);
      double cov[4];
      I.lm_cov(j, cov);
      lm_cov_tr[j] = cov[0] + cov[3];
      lms[j].point = res_lm[j];
      inv2_closed(cov, lms[j].info);  // marginalCovariance(l).inverse()
    }
  }
  // SLAM2D::set_copy_isam (SLAM2D.cpp:490-497): fresh ISAM2, batch update at calculateBestEstimate()
  void set_copy_isam() {
    copy_isam = isam;
    for (int i = 0; i < isam.P(); ++i) {
      copy_isam.th_pose[i] = isam.est_pose(i);
      copy_isam.d_pose[3 * i] = copy_isam.d_pose[3 * i + 1] = copy_isam.d_pose[3 * i + 2] = 0;
    }
    for (int j = 0; j < isam.L(); ++j) {
      copy_isam.th_lm[j] = isam.est_lm(j);
      copy_isam.d_lm[2 * j] = copy_isam.d_lm[2 * j + 1] = 0;
    }
    copy_isam.count = 0;
    copy_isam.update(0, 0);  // update #1 of the new ISAM2: one GN step at the best estimate
    use_copy = true;
  }
  int key_size() const { return (int)(traj.size() + lms.size()); }
};

// ----------------------------------------------------------------------------------------------
// OccupancyMap (include/em_exploration/OccupancyMap.h:10-19, src/em_exploration/OccupancyMap.cpp)
// ----------------------------------------------------------------------------------------------
static inline double prob2logodds(double p) { return std::log(p / (1.0 - p)); }
static inline double logodds2prob(double l) { return std::exp(l) / (1.0 + std::exp(l)); }

struct Occupancy {
  int rows, cols;
  double min_x, min_y, res;
  std::vector<double> map;
  double LO_FREE, LO_OCC, LO_MIN, LO_MAX, OCC_THRESH;
  Occupancy(const orc_config &c) {
    res = c.resolution;
    min_x = c.map_min_x;
    min_y = c.map_min_y;
    cols = (int)std::ceil((c.map_max_x - c.map_min_x) / res);  // OccupancyMap.cpp:13-14
    rows = (int)std::ceil((c.map_max_y - c.map_min_y) / res);
    map.assign((size_t)rows * cols, 0.0);
    LO_FREE = prob2logodds(0.3);
    LO_OCC = prob2logodds(0.7);
    LO_MIN = prob2logodds(0.05);
    LO_MAX = logodds2prob(0.95);  // sic (App. C.1): MAX_LOGODDS = LOGODDS2PROB(0.95)
    OCC_THRESH = prob2logodds(0.5);
  }
  void update_cell(int row, int col, bool free_) {  // OccupancyMap.cpp:55-62
    if (row >= rows || row < 0 || col >= cols || col < 0) return;
    double l = map[(size_t)row * cols + col] + (free_ ? LO_FREE : LO_OCC);
    l = std::min(LO_MAX, std::max(LO_MIN, l));
    map[(size_t)row * cols + col] = l;
  }
  void update_pose(const Pose &pose, const orc_config &c) {  // OccupancyMap.cpp:64-120
    int origin_row = (int)std::floor((pose.y - min_y) / res);
    int origin_col = (int)std::floor((pose.x - min_x) / res);
    int min_row, max_row, min_col, max_col;
    min_col = max_col = std::min(std::max(0, origin_col), cols - 1);
    min_row = max_row = std::min(std::max(0, origin_row), rows - 1);
    double theta0 = theta_of(pose);
    for (double b = c.min_bearing; b < c.max_bearing + 1e-5; b += 3 * 0.01745329251994329575) {
      double x = pose.x + c.max_range * std::cos(theta0 + b);
      double y = pose.y + c.max_range * std::sin(theta0 + b);
      int row = std::min(std::max(0, (int)std::floor((y - min_y) / res)), rows - 1);
      int col = std::min(std::max(0, (int)std::floor((x - min_x) / res)), cols - 1);
      min_row = std::min(min_row, row);
      max_row = std::max(max_row, row);
      min_col = std::min(min_col, col);
      max_col = std::max(max_col, col);
    }
    for (int row = min_row; row <= max_row; ++row)
      for (int col = min_col; col <= max_col; ++col) {
        double &l = map[(size_t)row * cols + col];
        if (std::fabs(l - LO_MIN) < 1e-5) continue;
        P2 pt{min_x + res * (col + 0.5), min_y + res * (row + 0.5)};
        double bearing = bearing_of(pose, pt, nullptr, nullptr);
        double range = range_of(pose, pt, nullptr, nullptr);
        if (!(bearing < c.max_bearing && bearing > c.min_bearing && range < c.max_range)) continue;
        if (l > OCC_THRESH + 1e-8)
          update_cell(row, col, false);
        else
          update_cell(row, col, true);
      }
  }
  void update_map(const Slam &slam, const orc_config &c) {  // OccupancyMap.cpp:122-138
    std::fill(map.begin(), map.end(), 0.0);
    for (const MapLm &l : slam.lms) {
      int r = (int)std::floor((l.point.y - min_y) / res);
      int col = (int)std::floor((l.point.x - min_x) / res);
      update_cell(r, col, false);
    }
    for (const MapPose &p : slam.traj)
      if (p.core) update_pose(p.pose, c);
  }
};

// ----------------------------------------------------------------------------------------------
// VirtualMap (src/em_exploration/VirtualMap.cpp)
// ----------------------------------------------------------------------------------------------
struct VirtualMap {
  int rows = 0, cols = 0, count_explored = 0;
  std::vector<double> prob;  // V
  std::vector<double> info;  // V x 4 (row-major 2x2)
  std::vector<uint8_t> updated;
  std::vector<P2> point;
  void initialize(const orc_config &c) {  // VirtualMap.cpp:318-362
    cols = (int)std::floor((c.map_max_x - c.map_min_x) / c.resolution);
    rows = (int)std::floor((c.map_max_y - c.map_min_y) / c.resolution);
    int extg = 20;
    for (int row = 0; row < rows; ++row)
      for (int col = 0; col < cols; ++col) {
        point.push_back(P2{(col + 0.5) * c.resolution + c.map_min_x, (row + 0.5) * c.resolution + c.map_min_y});
        prob.push_back(0.5);
        double i0 = 1.0 / std::pow(c.sigma0, 2);
        info.insert(info.end(), {i0, 0, 0, i0});
        updated.push_back(0);
      }
    count_explored = (rows - extg * 2 / (int)c.resolution) * (cols - extg * 2 / (int)c.resolution);
  }
  // VirtualMap::updateProbability(slam, sensor) (VirtualMap.cpp:61-84)
  void updateProbability(const Slam &slam, const orc_config &c) {
    std::fill(prob.begin(), prob.end(), 0.0);
    Occupancy occ(c);
    for (int s = 0; s < c.num_samples; ++s) {
      occ.update_map(slam, c);
      for (size_t i = 0; i < prob.size(); ++i) prob[i] += logodds2prob(occ.map[i]) / c.num_samples;
    }
  }
  // VirtualMap::covarianceIntersection2D (VirtualMap.cpp:364-378)
  static void covarianceIntersection2D(const double *m1, const double *m2, double *o) {
    double a = det2(m1), b = det2(m2);
    // m1.llt().solve(m2).trace()
    double l00 = std::sqrt(m1[0]), l10 = m1[2] / l00, l11 = std::sqrt(m1[3] - l10 * l10);
    double tr = 0;
    for (int col = 0; col < 2; ++col) {
      double b0 = m2[0 * 2 + col], b1 = m2[1 * 2 + col];
      double y0 = b0 / l00, y1 = (b1 - l10 * y0) / l11;
      double x1 = y1 / l11, x0 = (y0 - l10 * x1) / l00;
      tr += (col == 0 ? x0 : x1);
    }
    double cc = a * tr;
    double d = a + b - cc;
    double w = 0.5 * (2 * b - cc) / d;
    if ((w < 0 && d < 0) || (w > 1 && d > 0))
      w = 0.0;
    else if ((w < 0 && d > 0) || (w > 1 && d < 0))
      w = 1.0;
    for (int k = 0; k < 4; ++k) o[k] = w * m1[k] + (1.0 - w) * m2[k];
  }
  // VirtualMap::predictVirtualLandmark (VirtualMap.cpp:213-229)
  static bool predictVirtualLandmark(const MapPose &st, const P2 &pt, const orc_config &c, double *info_out) {
    double Hbx[3], Hbl[2], Hrx[3], Hrl[2];
    double bearing = bearing_of(st.pose, pt, Hbx, Hbl);
    double range = range_of(st.pose, pt, Hrx, Hrl);
    if (!(bearing < c.max_bearing && bearing > c.min_bearing && range < c.max_range && range > c.min_range))
      return false;
    double R[4] = {c.bearing_noise * c.bearing_noise, 0, 0, c.range_noise * c.range_noise};
    double Hx[6] = {Hbx[0], Hbx[1], Hbx[2], Hrx[0], Hrx[1], Hrx[2]};
    double Hl[4] = {Hbl[0], Hbl[1], Hrl[0], Hrl[1]};
    // Hl <- (Hl^T Hl)^-1 Hl^T
    double HtH[4] = {Hl[0] * Hl[0] + Hl[2] * Hl[2], Hl[0] * Hl[1] + Hl[2] * Hl[3],
                     Hl[1] * Hl[0] + Hl[3] * Hl[2], Hl[1] * Hl[1] + Hl[3] * Hl[3]};
    double HtHi[4];
    inv2_closed(HtH, HtHi);
    double Hp[4];  // HtHi * Hl^T
    Hp[0] = HtHi[0] * Hl[0] + HtHi[1] * Hl[1];
    Hp[1] = HtHi[0] * Hl[2] + HtHi[1] * Hl[3];
    Hp[2] = HtHi[2] * Hl[0] + HtHi[3] * Hl[1];
    Hp[3] = HtHi[2] * Hl[2] + HtHi[3] * Hl[3];
    // S = R + Hx * info.llt().solve(Hx^T)
    LLT3 llt(st.info);
    double X[6];  // 3x2 = info^-1 Hx^T
    for (int col = 0; col < 2; ++col) {
      double bcol[3] = {Hx[col * 3 + 0], Hx[col * 3 + 1], Hx[col * 3 + 2]}, x[3];
      llt.solve(bcol, x);
      for (int r = 0; r < 3; ++r) X[r * 2 + col] = x[r];
    }
    double S[4];
    for (int r = 0; r < 2; ++r)
      for (int col = 0; col < 2; ++col) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += Hx[r * 3 + k] * X[k * 2 + col];
        S[r * 2 + col] = R[r * 2 + col] + s;
      }
    // cov = Hp * S * Hp^T
    double T[4], cov[4];
    for (int r = 0; r < 2; ++r)
      for (int col = 0; col < 2; ++col) T[r * 2 + col] = Hp[r * 2 + 0] * S[0 * 2 + col] + Hp[r * 2 + 1] * S[1 * 2 + col];
    for (int r = 0; r < 2; ++r)
      for (int col = 0; col < 2; ++col) cov[r * 2 + col] = T[r * 2 + 0] * Hp[col * 2 + 0] + T[r * 2 + 1] * Hp[col * 2 + 1];
    inv2_llt(cov, info_out);
    return true;
  }
  // VirtualMap::updateInformation(map) / (state) (VirtualMap.cpp:256-271, 290-316)
  void updateInformation(const Slam &slam, const orc_config &c) {
    double i0 = 1.0 / std::pow(c.sigma0, 2);
    for (size_t i = 0; i < prob.size(); ++i) {
      updated[i] = 0;
      info[4 * i] = i0; info[4 * i + 1] = 0; info[4 * i + 2] = 0; info[4 * i + 3] = i0;
    }
    for (const MapPose &st : slam.traj) {
      if (!st.core) continue;
      if (det3(st.info) < 1e-10) continue;
      for (size_t i = 0; i < prob.size(); ++i) {
        double dx = st.pose.x - point[i].x, dy = st.pose.y - point[i].y;
        if (!(std::sqrt(dx * dx + dy * dy) < c.max_range)) continue;  // queryRadiusNeighbors
        double ninfo[4];
        if (!predictVirtualLandmark(st, point[i], c, ninfo)) continue;
        if (updated[i]) {
          double o[4];
          covarianceIntersection2D(&info[4 * i], ninfo, o);
          std::memcpy(&info[4 * i], o, sizeof(o));
        } else {
          std::memcpy(&info[4 * i], ninfo, sizeof(ninfo));
          updated[i] = 1;
        }
      }
    }
  }
  double explored(const orc_config &c) const {  // VirtualMap.cpp:47-59
    int count = 0, extg = 20;
    for (size_t i = 0; i < prob.size(); ++i)
      if ((prob[i] < 0.49 || prob[i] > 0.6) && c.map_min_x + extg <= point[i].x && point[i].x <= c.map_max_x - extg &&
          c.map_min_y + extg <= point[i].y && point[i].y <= c.map_max_y - extg)
        count++;
    return (double)count / count_explored;
  }
  double cov_trace(size_t i) const {
    double cv[4];
    inv2_llt(&info[4 * i], cv);
    return cv[0] + cv[3];
  }
};

// EMPlanner2D::calculateUncertainty / calculateUtility (Planner2D.cpp:343-366) and the
// D-optimality variant of calculateUncertainty_EM (Planner2D.cpp:321-341).
static double calculate_utility(const VirtualMap &vm, double distance, const orc_config &c) {
  int known = 0;
  for (double p : vm.prob)
    if (p < c.occupancy_threshold) known++;
  double pk = (double)known / (double)vm.prob.size();
  double dw = c.distance_weight0 - (c.distance_weight0 - c.distance_weight1) * pk;
  double u = 0;
  for (size_t i = 0; i < vm.prob.size(); ++i) u += 1.0 * vm.cov_trace(i);
  return u + distance * dw;
}
static double calculate_uncertainty_em(const VirtualMap &vm, int algorithm) {
  double u = 0;
  for (size_t i = 0; i < vm.prob.size(); ++i) {
    double w = vm.prob[i] > 0.49 ? 1.0 : 0.0;
    if (algorithm == 1)
      u += w / det2(&vm.info[4 * i]);
    else
      u += w * vm.cov_trace(i);
  }
  return u;
}

// ----------------------------------------------------------------------------------------------
// SS2D life-cycle (scripts/envs/pyss2d.py:59-206) + the planner calls the DRL loop uses.
// ----------------------------------------------------------------------------------------------
struct Env {
  orc_config cfg;
  Simulator sim;
  Slam slam;
  VirtualMap vm;
  int step = 0;
  bool cleared = true;

  // fixed_xy / n_fixed: the ini file's optional [Landmarks] list (pyss2d.py:107-115); cfg.num_landmarks counts them too
  // prior_info: the 3 x 3 information of SLAM2D::addPrior's VehicleBeliefState (row major; null: the ini file's diagonal)
  // prior_xyth: the pose of SLAM2D::addPrior's VehicleBeliefState when it is not the simulator's initial vehicle pose (SLAM2D.cpp:44-57:
  // the prior factor's pose AND the initial estimate of x0; null: the vehicle's, as pyss2d.py:124-135 passes it)
  Env(const orc_config &c, uint32_t seed, double x0, double y0, double th0, const double *fixed_xy = nullptr, int n_fixed = 0,
      const double *prior_info = nullptr, const double *prior_xyth = nullptr) : cfg(c), sim(c, seed) {
    slam.cfg = c;
    vm.initialize(c);
    // pyss2d.py:102-138
    sim.initializeVehicle(make_pose(x0, y0, th0));
    sim.addLandmarks((unsigned)c.num_landmarks, fixed_xy, (unsigned)n_fixed);
    double info[9] = {1.0 / (c.sigma_x0 * c.sigma_x0), 0, 0, 0, 1.0 / (c.sigma_y0 * c.sigma_y0), 0, 0, 0,
                      1.0 / (c.sigma_theta0 * c.sigma_theta0)};
    if (prior_info) std::memcpy(info, prior_info, sizeof(info));
    slam.addPrior(prior_xyth ? make_pose(prior_xyth[0], prior_xyth[1], prior_xyth[2]) : sim.vehicle, info);
    for (const Measurement &m : sim.measure()) slam.addMeasurement(m.key, m.bearing, m.range);
    slam.optimize();
    step = 1;
  }
  // SS2D.simulate(odom, core=True) (pyss2d.py:171-206)
  int simulate(double ox, double oy, double oth) {
    if (!(cfg.map_min_x < ox && ox < cfg.map_max_x) || !(cfg.map_min_y < oy && oy < cfg.map_max_y)) return 1;
    Pose odom = make_pose(ox, oy, oth);
    sim.move(odom);
    double sig[3] = {cfg.translation_noise, cfg.translation_noise, cfg.rotation_noise};
    slam.addOdometry(odom, sig);
    bool obstacle = false;
    std::vector<Measurement> ms = sim.measure();  // first (discarded) noisy measure, pyss2d.py:182
    for (const Measurement &m : ms) {
      if (cleared) {
        if (m.range < cfg.safe_distance) { obstacle = true; cleared = false; break; }
      } else {
        if (slam.lm_slot.find(m.key) == slam.lm_slot.end() && m.range < cfg.safe_distance) {
          obstacle = true; cleared = false; break;
        }
      }
    }
    if (!obstacle) cleared = true;
    step++;
    for (const Measurement &m : sim.measure()) slam.addMeasurement(m.key, m.bearing, m.range);
    slam.optimize();
    vm.updateProbability(slam, cfg);
    vm.updateInformation(slam, cfg);
    return obstacle ? 2 : 0;
  }
  // EMPlanner2D::line_planner (Planner2D.cpp:937-1041), frontier-goal branch
  int line_plan(double gx, double gy, double *out, int max_actions) const {
    std::vector<Pose> actions;
    const Pose &cur = slam.traj.back().pose;
    double rx = cur.x, ry = cur.y, rth = theta_of(cur);
    double gth = std::atan2(gy - ry, gx - rx);
    if (rth < 0) rth = M_PI * 2 + rth;
    if (gth < 0) gth = M_PI * 2 + gth;
    double dr = 180 * M_PI / 180;
    double diff = gth - rth;
    auto push_rot = [&](double d, double sign) {
      int q = (int)(d / dr);
      double rem = d - dr * q;
      for (int i = 0; i < q; ++i) actions.push_back(make_pose(0, 0, sign * dr));
      actions.push_back(make_pose(0, 0, sign * rem));
    };
    if (diff > M_PI) {
      push_rot(2 * M_PI - diff, -1.0);
    } else if (diff > -M_PI && diff < 0) {
      push_rot(std::abs(diff), -1.0);
    } else if (diff <= -M_PI) {
      push_rot(2 * M_PI - std::abs(diff), 1.0);
    } else {
      push_rot(diff, 1.0);
    }
    double dist = std::sqrt(std::pow(rx - gx, 2) + std::pow(ry - gy, 2));
    int dq = (int)(dist / cfg.max_edge_length);
    double drem = dist - dq * cfg.max_edge_length;
    for (int i = 0; i < dq; ++i) actions.push_back(make_pose(cfg.max_edge_length, 0, 0));
    actions.push_back(make_pose(drem, 0, 0));
    int n = (int)actions.size();
    for (int i = 0; i < n && i < max_actions; ++i) {
      out[3 * i] = actions[i].x;
      out[3 * i + 1] = actions[i].y;
      out[3 * i + 2] = theta_of(actions[i]);
    }
    return n;
  }
  // EMPlanner2D::simulations_reward (Planner2D.cpp:1416-1468)
  double simulations_reward(const double *actions, int n) const {
    Slam ts = slam;
    ts.set_copy_isam();
    VirtualMap tvm = vm;
    Simulator tsim = sim;  // copies the three RNG states
    double dist = 0;
    double u0 = calculate_utility(tvm, 0, cfg);
    for (int i = 0; i < n; ++i) {
      double sx = actions[3 * i], sy = actions[3 * i + 1], sth = actions[3 * i + 2];
      // the Pose2 passed from Python has theta() = atan2(sin, cos) of the python float
      Pose odom = make_pose(sx, sy, sth);
      double oth = theta_of(odom);
      dist = dist + std::sqrt(std::pow(sx, 2) + std::pow(sy, 2) + cfg.angle_weight * std::pow(oth, 2));
      tsim.move(odom);
      double sig[3] = {cfg.translation_noise, cfg.translation_noise, cfg.rotation_noise};
      ts.addOdometry(odom, sig);
      for (const Measurement &m : tsim.measure()) ts.addMeasurement(m.key, m.bearing, m.range);
      ts.optimize();
      tvm.updateProbability(ts, cfg);
      tvm.updateInformation(ts, cfg);
    }
    double u1 = calculate_utility(tvm, dist, cfg);
    return u0 - u1;
  }
  // SLAM2D::adjacency_degree_get (SLAM2D.cpp:198-273): node order = landmarks by GT key, then poses
  void adjacency(std::vector<double> &A, std::vector<double> &X) const {
    int Ln = (int)slam.lms.size(), Pn = (int)slam.traj.size(), N = Ln + Pn;
    A.assign((size_t)N * N, 0.0);
    X.assign(N, 0.0);
    std::vector<int> node_of_slot(Ln);
    {
      int k = 0;
      for (const auto &kv : slam.lm_slot) node_of_slot[kv.second] = k++;
    }
    const Isam &I = slam.isam;
    for (size_t i = 0; i < I.odo.size(); ++i) {
      double d = std::sqrt(std::pow(I.odo[i].measured.x, 2) + std::pow(I.odo[i].measured.y, 2)) + 0.001;
      int a = Ln + (int)i, b = Ln + (int)i + 1;
      A[(size_t)a * N + b] = d;
      A[(size_t)b * N + a] = d;
      X[a] = slam.pose_cov_tr[i];
      X[b] = slam.pose_cov_tr[i + 1];
    }
    for (const MeasFactor &f : I.meas) {  // factor order: last factor wins
      int a = Ln + f.pose, b = node_of_slot[f.lm];
      A[(size_t)a * N + b] = f.range;
      A[(size_t)b * N + a] = f.range;
      X[a] = slam.pose_cov_tr[f.pose];
      X[b] = slam.lm_cov_tr[f.lm];
    }
  }
};

}  // namespace orc

// ================================================================================================
// C interface for ctypes (tests / smoke / cpu_baseline only)
// ================================================================================================
using namespace orc;
extern "C" {

void *orc_create(const orc_config *cfg, uint32_t seed, double x0, double y0, double th0) {
  return new Env(*cfg, seed, x0, y0, th0);
}
void *orc_create_fixed(const orc_config *cfg, uint32_t seed, double x0, double y0, double th0, const double *fixed_xy, int n_fixed) {
  if (n_fixed < 0 || n_fixed > cfg->num_landmarks) return nullptr;
  return new Env(*cfg, seed, x0, y0, th0, fixed_xy, n_fixed);
}
void *orc_create_prior_pose(const orc_config *cfg, uint32_t seed, double x0, double y0, double th0, const double *prior_info9,
                            const double *prior_xyth) {
  return new Env(*cfg, seed, x0, y0, th0, nullptr, 0, prior_info9, prior_xyth);
}
void *orc_create_prior(const orc_config *cfg, uint32_t seed, double x0, double y0, double th0, const double *prior_info9) {
  return new Env(*cfg, seed, x0, y0, th0, nullptr, 0, prior_info9);
}
void orc_destroy(void *h) { delete (Env *)h; }
void *orc_clone(void *h) { return new Env(*(Env *)h); }
int orc_simulate(void *h, double ox, double oy, double oth) { return ((Env *)h)->simulate(ox, oy, oth); }
double orc_utility(void *h, double dist) { return calculate_utility(((Env *)h)->vm, dist, ((Env *)h)->cfg); }
double orc_uncertainty_em(void *h, int algorithm) { return calculate_uncertainty_em(((Env *)h)->vm, algorithm); }
double orc_explored(void *h) { return ((Env *)h)->vm.explored(((Env *)h)->cfg); }
int orc_line_plan(void *h, double gx, double gy, double *out, int max_actions) {
  return ((Env *)h)->line_plan(gx, gy, out, max_actions);
}
double orc_simulations_reward(void *h, const double *actions, int n) { return ((Env *)h)->simulations_reward(actions, n); }
int orc_step_count(void *h) { return ((Env *)h)->step; }
int orc_num_poses(void *h) { return (int)((Env *)h)->slam.traj.size(); }
int orc_num_landmarks(void *h) { return (int)((Env *)h)->slam.lms.size(); }
int orc_num_factors(void *h) { return (int)((Env *)h)->slam.isam.meas.size(); }
int orc_vm_rows(void *h) { return ((Env *)h)->vm.rows; }
int orc_vm_cols(void *h) { return ((Env *)h)->vm.cols; }
// estimated poses (x, y, theta) and information (9) per pose
void orc_get_poses(void *h, double *xyt, double *info) {
  Env *e = (Env *)h;
  for (size_t i = 0; i < e->slam.traj.size(); ++i) {
    const MapPose &p = e->slam.traj[i];
    xyt[3 * i] = p.pose.x; xyt[3 * i + 1] = p.pose.y; xyt[3 * i + 2] = theta_of(p.pose);
    if (info) std::memcpy(info + 9 * i, p.info, sizeof(double) * 9);
  }
}
// landmarks sorted by GT key (node order): key, x, y, info(4)
void orc_get_landmarks(void *h, int *keys, double *xy, double *info) {
  Env *e = (Env *)h;
  int k = 0;
  for (const auto &kv : e->slam.lm_slot) {
    const MapLm &l = e->slam.lms[kv.second];
    keys[k] = (int)l.key;
    xy[2 * k] = l.point.x; xy[2 * k + 1] = l.point.y;
    if (info) std::memcpy(info + 4 * k, l.info, sizeof(double) * 4);
    k++;
  }
}
void orc_get_cov_traces(void *h, double *lm_tr /*sorted by key*/, double *pose_tr) {
  Env *e = (Env *)h;
  int k = 0;
  for (const auto &kv : e->slam.lm_slot) lm_tr[k++] = e->slam.lm_cov_tr[kv.second];
  for (size_t i = 0; i < e->slam.pose_cov_tr.size(); ++i) pose_tr[i] = e->slam.pose_cov_tr[i];
}
void orc_get_virtual_map(void *h, double *prob, double *info /*V*4*/, double *cov_trace, uint8_t *updated) {
  Env *e = (Env *)h;
  size_t V = e->vm.prob.size();
  if (prob) std::memcpy(prob, e->vm.prob.data(), V * sizeof(double));
  if (info) std::memcpy(info, e->vm.info.data(), V * 4 * sizeof(double));
  if (cov_trace)
    for (size_t i = 0; i < V; ++i) cov_trace[i] = e->vm.cov_trace(i);
  if (updated) std::memcpy(updated, e->vm.updated.data(), V);
}
void orc_get_gt(void *h, double *vehicle_xyt, double *landmarks_xy /*num*2 by key*/, int *iter_order) {
  Env *e = (Env *)h;
  vehicle_xyt[0] = e->sim.vehicle.x; vehicle_xyt[1] = e->sim.vehicle.y; vehicle_xyt[2] = theta_of(e->sim.vehicle);
  int k = 0;
  for (const auto &kv : e->sim.landmarks) {
    landmarks_xy[2 * kv.first] = kv.second.x;
    landmarks_xy[2 * kv.first + 1] = kv.second.y;
    if (iter_order) iter_order[k] = (int)kv.first;
    k++;
  }
}
// dense adjacency ((L+P)^2) and trace feature (L+P), node order landmarks-by-key then poses
void orc_get_adjacency(void *h, double *A, double *X) {
  std::vector<double> a, x;
  ((Env *)h)->adjacency(a, x);
  std::memcpy(A, a.data(), a.size() * sizeof(double));
  std::memcpy(X, x.data(), x.size() * sizeof(double));
}
// measurement factor list (pose idx, GT key, bearing, range)
void orc_get_factors(void *h, int *pose, int *key, double *bearing, double *range) {
  Env *e = (Env *)h;
  const Isam &I = e->slam.isam;
  for (size_t i = 0; i < I.meas.size(); ++i) {
    pose[i] = I.meas[i].pose;
    key[i] = (int)e->slam.slot_key[I.meas[i].lm];
    bearing[i] = I.meas[i].bearing;
    range[i] = I.meas[i].range;
  }
}
// full state of the linearisation point (theta, delta) for parity checks of the SLAM kernel
// the full covariance of the last solve (n x n, n = 2 L + 3 P, order [landmarks by slot, poses]): the joint marginals that
// FastMarginals::recover provides to FastMarginals2 (FastMarginals.cpp:130-186)
int orc_get_full_cov(void *h, double *out) {
  const Isam &I = ((Env *)h)->slam.isam;
  const int n = I.n;
  if (out)
    for (int a = 0; a < n; ++a)
      for (int c = a; c < n; ++c) {
        double s = 0;
        for (int k = c; k < n; ++k) s += I.Linv[(size_t)k * n + a] * I.Linv[(size_t)k * n + c];
        out[(size_t)a * n + c] = out[(size_t)c * n + a] = s;
      }
  return n;
}
void orc_get_slot_keys(void *h, int *out) {
  const Slam &S = ((Env *)h)->slam;
  for (size_t j = 0; j < S.slot_key.size(); ++j) out[j] = (int)S.slot_key[j];
}
void orc_dev_set_relin(double thr, int skip, int mode) {
  g_relin_thr = thr;
  g_relin_skip = skip;
  g_relin_mode = mode;
}
void orc_get_isam(void *h, double *th_pose /*P*3 x,y,theta*/, double *d_pose, double *th_lm /*slot order*/, double *d_lm,
                  int *count) {
  Env *e = (Env *)h;
  const Isam &I = e->slam.isam;
  for (int i = 0; i < I.P(); ++i) {
    th_pose[3 * i] = I.th_pose[i].x; th_pose[3 * i + 1] = I.th_pose[i].y; th_pose[3 * i + 2] = theta_of(I.th_pose[i]);
    for (int r = 0; r < 3; ++r) d_pose[3 * i + r] = I.d_pose[3 * i + r];
  }
  for (int j = 0; j < I.L(); ++j) {
    th_lm[2 * j] = I.th_lm[j].x; th_lm[2 * j + 1] = I.th_lm[j].y;
    d_lm[2 * j] = I.d_lm[2 * j]; d_lm[2 * j + 1] = I.d_lm[2 * j + 1];
  }
  *count = I.count;
}

// ---- stand-alone known-answer entry points ---------------------------------------------------
void orc_kat_rng(uint32_t seed, int n_uniform, int n_normal, double *out) {
  RNG r(seed);
  for (int i = 0; i < n_uniform; ++i) out[i] = r.uniformReal(0.0, 1.0);
  for (int i = 0; i < n_normal; ++i) out[n_uniform + i] = r.normal(0.0, 1.0);
}
void orc_kat_ci(const double *m1, const double *m2, double *out) { VirtualMap::covarianceIntersection2D(m1, m2, out); }
// occupancy ladder: apply a sequence of observations to one cell; obs[i]=1 => "in view"
void orc_kat_occupancy_ladder(int n, double *probs_out, int landmark_first) {
  orc_config c{};
  c.resolution = 2; c.map_min_x = 0; c.map_max_x = 2; c.map_min_y = 0; c.map_max_y = 2;
  Occupancy o(c);
  if (landmark_first) o.update_cell(0, 0, false);
  for (int i = 0; i < n; ++i) {
    probs_out[i] = logodds2prob(o.map[0]);
    if (std::fabs(o.map[0] - o.LO_MIN) < 1e-5) continue;
    if (o.map[0] > o.OCC_THRESH + 1e-8) o.update_cell(0, 0, false);
    else o.update_cell(0, 0, true);
  }
}
int orc_kat_predict(const double *pose_xyt, const double *info9, double px, double py, const orc_config *c, double *info_out) {
  MapPose st;
  st.pose = make_pose(pose_xyt[0], pose_xyt[1], pose_xyt[2]);
  std::memcpy(st.info, info9, sizeof(double) * 9);
  st.core = true;
  return VirtualMap::predictVirtualLandmark(st, P2{px, py}, *c, info_out) ? 1 : 0;
}
double orc_wrap_theta(double th) { return wrap_theta(th); }
int orc_landmark_iteration_order(int num, int *order) {
  std::unordered_map<unsigned, int> m;
  for (int i = 0; i < num; ++i) m.emplace((unsigned)i, i);
  int k = 0;
  for (const auto &kv : m) order[k++] = (int)kv.first;
  return k;
}
}
