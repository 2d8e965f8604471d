// sfw_kernels.hip — gfx950 (MI355X / CDNA4) kernels of the DWA rollout +
// social-force scoring path.  wave = 64 lanes everywhere.
//
//   K1  sfw_rollout_kernel   one THREAD per (v,w) sample: accel-limited unicycle
//       sfw_footprint_kernel rollout (K1a); one thread per (step, sample):
//       sfw_costmap_scan_..  footprint legality + cost (K1b); in-order scan (K1c)
//                            (reference src/sfw_planner.cpp:540-588, :643-667;
//                            world_model.hpp:45-75; src/costmap_model.cpp:21-121;
//                            line_iterator.hpp:39-97).  The robot's motion does
//                            not depend on the pedestrians, so it is rolled out
//                            once here; the post-step robot states go to HBM as
//                            a [step][sample] table that K2 streams back.
//   K2  sfw_social_kernel    one WAVE per G samples: the pedestrians of each
//                            sample are integrated under the social-force model
//                            (lightsfm computeForces/updatePosition, reference
//                            call sites :592,:594,:697) with all agent state in
//                            LDS; social work accumulated per lane and reduced
//                            per sample (reference :613-629, :678-705).
//   K3  sfw_argmin_*         block-wide + grid argmin under the reference's
//                            selection order (reference :394-414).
//
// The pair interaction is the hot loop (>= 90 % of the work for N >= 10).  It
// is a nonlinear function of a 2-vector pair (2 exp, atan2, 2 rsqrt), not a
// contraction: vector ALU work, no MFMA.

#include "sfw_device.h"

#include <math.h>

namespace {

constexpr int WAVE = 64;

// ===========================================================================
// K1: robot rollout + costmap
// ===========================================================================
// Contraction is switched off in K1 so that the pose arithmetic is the same
// IEEE sequence the CPU reference executes (cell-index truncation makes the
// last ulp matter, SURVEY.md §7 "discrete thresholds").
#pragma clang fp contract(off)

__device__ __forceinline__ bool world_to_map(const sfw_launch &L, double wx, double wy, unsigned &mx,
                                             unsigned &my) {
  // nav2_costmap_2d::Costmap2D::worldToMap (Foxy), SURVEY.md Appendix C
  if (wx < L.origin_x || wy < L.origin_y) return false;
  mx = static_cast<unsigned>((wx - L.origin_x) / L.resolution);
  my = static_cast<unsigned>((wy - L.origin_y) / L.resolution);
  return mx < L.size_x && my < L.size_y;
}

// reference src/costmap_model.cpp:112-121
__device__ __forceinline__ int point_code(const sfw_launch &L, int x, int y) {
  return L.cells[static_cast<size_t>(y) * L.size_x + static_cast<size_t>(x)];
}

// Bresenham walk of one footprint edge (reference line_iterator.hpp:39-97,
// src/costmap_model.cpp:95-110).  Returns the max cell cost, or -1 / -2 at the
// first lethal / unknown cell.
__device__ double line_cost(const sfw_launch &L, int x0, int y0, int x1, int y1) {
  const int adx = abs(x1 - x0), ady = abs(y1 - y0);
  const int sx = (x1 >= x0) ? 1 : -1, sy = (y1 >= y0) ? 1 : -1;
  const bool x_major = adx >= ady;
  const int den = x_major ? adx : ady;
  const int add = x_major ? ady : adx;
  int num = den / 2, x = x0, y = y0, best = 0;
  for (int n = 0; n <= den; ++n) {
    const int c = point_code(L, x, y);
    if (c == 255) return -2.0;
    if (c == 254) return -1.0;
    best = max(best, c);
    num += add;
    if (num >= den) {
      num -= den;
      if (x_major) y += sy; else x += sx;
    }
    if (x_major) x += sx; else y += sy;
  }
  return static_cast<double>(best);
}

// reference world_model.hpp:45-75 + src/costmap_model.cpp:21-92; c,s = cos/sin(theta)
__device__ double footprint_cost(const sfw_launch &L, double x, double y, double c, double s) {
  unsigned cx, cy;
  if (!world_to_map(L, x, y, cx, cy)) return -3.0;
  const int K = L.K;
  if (K < 3) {
    const int code = point_code(L, (int)cx, (int)cy);
    if (code == 255) return -2.0;
    if (code == 254 || code == 253) return -1.0;
    return static_cast<double>(code);
  }
  double fc = 0.0;
  // first vertex of edge 0
  double qx = L.footprint[0], qy = L.footprint[1];
  double ax = x + (qx * c - qy * s), ay = y + (qx * s + qy * c);
  const double fx0 = ax, fy0 = ay;
  for (int e = 0; e < K; ++e) {
    double bx, by;
    double sx0 = ax, sy0 = ay;
    if (e + 1 < K) {
      qx = L.footprint[2 * (e + 1)];
      qy = L.footprint[2 * (e + 1) + 1];
      bx = x + (qx * c - qy * s);
      by = y + (qx * s + qy * c);
    } else {  // closing edge: back() -> front()
      bx = fx0;
      by = fy0;
    }
    unsigned x0, y0, x1, y1;
    if (!world_to_map(L, sx0, sy0, x0, y0)) return -3.0;
    if (!world_to_map(L, bx, by, x1, y1)) return -3.0;
    const double lc = line_cost(L, (int)x0, (int)y0, (int)x1, (int)y1);
    fc = fmax(lc, fc);
    if (lc < 0) return lc;
    ax = bx;
    ay = by;
  }
  return fc;
}

// reference sfw_planner.hpp:457-463
__device__ __forceinline__ double new_velocity(double vg, double vi, double a_max, double dt) {
  if ((vg - vi) >= 0) return fmin(vg, vi + a_max * dt);
  return fmax(vg, vi - a_max * dt);
}
// reference sfw_planner.hpp:399-407 (float in, float out)
__device__ __forceinline__ float normalize_angle_f(float val, float mn, float mx) {
  if (val >= mn) return mn + fmodf(val - mn, mx - mn);
  return mx - fmodf(mn - val, mx - mn);
}

// K1a: sequential pose integration, one thread per sample.  Writes the
// pre-step footprint frame (x_i, y_i, cos th_i, sin th_i) and the post-step robot
// agent state for every step, plus the pedestrian-independent cost terms.
__global__ void __launch_bounds__(64) sfw_rollout_kernel(const sfw_launch L) {
  const int64_t local = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (local >= L.chunk_count) return;
  const int64_t t = L.chunk_begin + local;
  const int iv = static_cast<int>(t / L.nw), iw = static_cast<int>(t % L.nw);
  const double vx_samp = L.linvels[iv], vth_samp = L.angvels[iw], vy_samp = L.vy_samp;
  if (L.skip_zero_sample && vx_samp == 0.0 && vth_samp == 0.0) {  // ref :349-352
    L.status[t] = SFW_ST_SKIPPED;
    L.costs[t] = SFW_COST_SKIPPED;
    return;
  }
  L.status[t] = SFW_ST_VALID;  // K1c downgrades it if a step is illegal
  double x_i = L.rs.x, y_i = L.rs.y, th_i = L.rs.theta;
  double vx_i = L.rs.vx, vy_i = L.rs.vy, vth_i = L.rs.vtheta;
  const int S = L.S;
  const double dt = L.dt;
  for (int i = 0; i < S; ++i) {
    double s, c;
    sincos(th_i, &s, &c);
    sfw_pose_frame f;
    f.x = x_i; f.y = y_i; f.c = c; f.s = s;
    L.frame[static_cast<int64_t>(i) * L.rstep_stride + local] = f;
    if (L.points) {                                       // ref :578
      L.points[3 * i] = x_i;
      L.points[3 * i + 1] = y_i;
      L.points[3 * i + 2] = th_i;
    }
    vx_i = new_velocity(vx_samp, vx_i, L.ga.acc_x, dt);   // ref :581-583
    vy_i = new_velocity(vy_samp, vy_i, L.ga.acc_y, dt);
    vth_i = new_velocity(vth_samp, vth_i, L.ga.acc_theta, dt);
    double c2 = 0.0, s2 = 0.0;
    if (vy_i != 0.0) sincos(M_PI_2 + th_i, &s2, &c2);     // holonomic term, 0 for the grid
    const double xn = x_i + (vx_i * c + vy_i * c2) * dt;  // ref :586-588 (old theta)
    const double yn = y_i + (vx_i * s + vy_i * s2) * dt;
    x_i = xn;
    y_i = yn;
    th_i = th_i + vth_i * dt;
    sfw_robot_step r;
    r.x = x_i; r.y = y_i; r.vx = vx_i; r.vy = vy_i;
    L.rstep[static_cast<int64_t>(i) * L.rstep_stride + local] = r;
  }
  // ref :643-666 without the costmap and social terms (left-to-right sum order kept)
  const double dx = L.ga.wpx - x_i, dy = L.ga.wpy - y_i;
  const double d = dx * dx + dy * dy;
  double ang = atan2(dy, dx) - th_i;
  ang = normalize_angle_f(static_cast<float>(ang), static_cast<float>(-M_PI), static_cast<float>(M_PI));
  ang = fabs(ang) / M_PI;
  const double vel = fabs(L.p.max_vel_x - vx_i) / L.p.max_vel_x;
  L.base_cost[t] = L.p.vel_weight * vel + L.p.distance_weight * d + L.p.angle_weight * ang;
}

// K1b: footprint legality/cost of one pose, one thread per (step, sample).
// Independent across steps once the poses are known, so the S*T checks run in
// parallel instead of serially inside the rollout.
__global__ void __launch_bounds__(256) sfw_footprint_kernel(const sfw_launch L) {
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t n = L.chunk_count;
  if (idx >= n * L.S) return;
  const int64_t step = idx / n, local = idx - step * n;
  if (L.status[L.chunk_begin + local] == SFW_ST_SKIPPED) return;
  const sfw_pose_frame f = L.frame[step * L.rstep_stride + local];
  const double fc = footprint_cost(L, f.x, f.y, f.c, f.s);  // includes the ref :545 map check
  L.fcode[step * L.rstep_stride + local] = static_cast<int16_t>(fc);
}

// K1c: in-order scan of the per-step footprint costs, one thread per sample:
// first illegal step rejects the trajectory (ref :555-573), otherwise
// costmap_cost accumulates in step order (ref :575, :656).
__global__ void __launch_bounds__(256) sfw_costmap_scan_kernel(const sfw_launch L) {
  const int64_t local = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (local >= L.chunk_count) return;
  const int64_t t = L.chunk_begin + local;
  if (L.status[t] == SFW_ST_SKIPPED) return;
  const int S = L.S;
  double cm = 0.0;
  int n_ok = 0;
  for (; n_ok < S; ++n_ok) {
    const double fc = static_cast<double>(L.fcode[static_cast<int64_t>(n_ok) * L.rstep_stride + local]);
    if (fc >= 254.0 || fc < 0) break;
    cm += fc / 255.0;
  }
  if (L.n_points) *L.n_points = n_ok;
  if (n_ok < S) {
    L.status[t] = SFW_ST_INVALID;
    L.costs[t] = SFW_COST_INVALID;
    return;
  }
  cm = cm / S;
  const double base = L.base_cost[t] + L.p.costmap_weight * cm;
  L.base_cost[t] = base;
  // No agent vector at all: social work is identically 0 and K2 is not launched.
  if (L.A == 0) L.costs[t] = base + L.p.social_weight * 0.0;
}

#pragma clang fp contract(fast)

// ===========================================================================
// K2: social-force integration, one wave per G samples
// ===========================================================================
template <typename R> struct vec2;
template <> struct vec2<double> { using type = double2; };
template <> struct vec2<float> { using type = float2; };

template <typename R> __device__ __forceinline__ R m_sqrt(R x);
template <> __device__ __forceinline__ double m_sqrt<double>(double x) { return sqrt(x); }
template <> __device__ __forceinline__ float m_sqrt<float>(float x) { return sqrtf(x); }
template <typename R> __device__ __forceinline__ R m_rsqrt(R x);
template <> __device__ __forceinline__ double m_rsqrt<double>(double x) { return rsqrt(x); }
template <> __device__ __forceinline__ float m_rsqrt<float>(float x) { return rsqrtf(x); }
template <typename R> __device__ __forceinline__ R m_exp(R x);
template <> __device__ __forceinline__ double m_exp<double>(double x) { return exp(x); }
template <> __device__ __forceinline__ float m_exp<float>(float x) { return expf(x); }
template <typename R> __device__ __forceinline__ R m_atan2(R y, R x);
template <> __device__ __forceinline__ double m_atan2<double>(double y, double x) { return atan2(y, x); }
template <> __device__ __forceinline__ float m_atan2<float>(float y, float x) { return atan2f(y, x); }

// Per-launch social-force constants in the kernel's real type.
template <typename R> struct sfm_consts {
  R lambda, gamma, inv_gamma, n, n_prime, f_social;
  R f_desired, inv_tau, f_obstacle, inv_sigma;
  R dt, rr;
};

// Force exerted ON agent i BY agent j (one term of lightsfm's
// computeSocialForce).  The term is antisymmetric under
// i<->j when every agent carries the same sfm::Parameters (the reference never
// overrides them), so the caller applies -f to j and evaluates each unordered
// pair once.
//   diff = pj - pi, dhat = diff/|diff|, w = vi - vj, I = lambda*w + dhat,
//   theta = angle(dhat) - angle(I) = atan2(I x diff, I . diff)   in (-pi, pi]
//   B = gamma*|I|
//   f = Fs * ( -exp(-|diff|/B - (n' B theta)^2) * Ihat
//              - sign(theta) * exp(-|diff|/B - (n B theta)^2) * leftNormal(Ihat) )
template <typename R>
__device__ __forceinline__ void pair_force(const sfm_consts<R> &k, R pix, R piy, R vix, R viy, R pjx,
                                           R pjy, R vjx, R vjy, R &fx, R &fy) {
  const R dx = pjx - pix, dy = pjy - piy;
  const R d2 = dx * dx + dy * dy;
  const R rd = d2 > R(0) ? m_rsqrt<R>(d2) : R(0);  // zero vector stays zero (normalized())
  const R dn = d2 * rd;
  const R ix = k.lambda * (vix - vjx) + dx * rd;
  const R iy = k.lambda * (viy - vjy) + dy * rd;
  const R l2 = ix * ix + iy * iy;
  const R rl = m_rsqrt<R>(l2);
  const R il = l2 * rl;
  const R ihx = ix * rl, ihy = iy * rl;
  const R cr = ix * dy - iy * dx;   // |I||diff| sin(theta)
  const R dt = ix * dx + iy * dy;   // |I||diff| cos(theta)
  const R theta = m_atan2<R>(cr, dt);
  const R B = k.gamma * il;
  const R a = -dn * rl * k.inv_gamma;  // -|diff| / B
  const R bt = B * theta;
  const R sv = k.n_prime * bt, sa = k.n * bt;
  const R ev = m_exp<R>(a - sv * sv);
  R ea = m_exp<R>(a - sa * sa);
  ea = theta > R(0) ? ea : (theta < R(0) ? -ea : R(0));  // sign(theta) * exp(...)
  // f = Fs * (-ev * Ihat - ea * leftNormal(Ihat)),  leftNormal(x,y) = (-y, x)
  fx = k.f_social * (ea * ihy - ev * ihx);
  fy = -k.f_social * (ev * ihy + ea * ihx);
}

// desiredForce of one person (lightsfm computeDesiredForce).
template <typename R>
__device__ __forceinline__ void desired_force(const sfm_consts<R> &k, R px, R py, R vx, R vy, bool has_goal,
                                              R gx, R gy, R gr, R dv, R &fx, R &fy) {
  const R ex = gx - px, ey = gy - py;
  const R e2 = ex * ex + ey * ey;
  const R en = m_sqrt<R>(e2);
  if (has_goal && en > gr) {
    const R inv = en > R(0) ? R(1) / en : R(0);
    fx = k.f_desired * (ex * inv * dv - vx) * k.inv_tau;
    fy = k.f_desired * (ey * inv * dv - vy) * k.inv_tau;
  } else {
    fx = -vx * k.inv_tau;
    fy = -vy * k.inv_tau;
  }
}

// obstacleForce of one agent: mean over the shared laser points (lightsfm
// computeObstacleForce).  obs lives in LDS; every lane reads the same address
// (broadcast).
template <typename R, typename R2>
__device__ __forceinline__ void obstacle_force(const sfm_consts<R> &k, const R2 *obs, int O, R inv_O, R px,
                                               R py, R radius, R &fx, R &fy) {
  R ax = R(0), ay = R(0);
  for (int o = 0; o < O; ++o) {
    const R2 q = obs[o];
    const R mx = px - q.x, my = py - q.y;
    const R m2 = mx * mx + my * my;
    const R rm = m2 > R(0) ? m_rsqrt<R>(m2) : R(0);
    const R dist = m2 * rm - radius;
    const R e = k.f_obstacle * m_exp<R>(-dist * k.inv_sigma);
    ax += e * mx * rm;
    ay += e * my * rm;
  }
  fx = ax * inv_O;
  fy = ay * inv_O;
}

template <typename R> struct lds_layout {
  using R2 = typename vec2<R>::type;
  R2 *pos, *vel, *frc, *frj, *goal, *obs;
  R *gr, *dv, *rad;
  double *swp;
  int *id, *hasgoal, *dead;
  __device__ lds_layout(char *base, int A, int GA, int G, int O) {
    auto take = [&](size_t bytes) {
      char *p = base;
      base += (bytes + 15) & ~size_t(15);
      return p;
    };
    pos = reinterpret_cast<R2 *>(take(sizeof(R2) * GA));
    vel = reinterpret_cast<R2 *>(take(sizeof(R2) * GA));
    frc = reinterpret_cast<R2 *>(take(sizeof(R2) * GA));
    frj = reinterpret_cast<R2 *>(take(sizeof(R2) * GA));
    goal = reinterpret_cast<R2 *>(take(sizeof(R2) * A));
    obs = reinterpret_cast<R2 *>(take(sizeof(R2) * (O > 0 ? O : 1)));
    swp = reinterpret_cast<double *>(take(sizeof(double) * GA));
    gr = reinterpret_cast<R *>(take(sizeof(R) * A));
    dv = reinterpret_cast<R *>(take(sizeof(R) * A));
    rad = reinterpret_cast<R *>(take(sizeof(R) * A));
    id = reinterpret_cast<int *>(take(sizeof(int) * A));
    hasgoal = reinterpret_cast<int *>(take(sizeof(int) * GA));
    dead = reinterpret_cast<int *>(take(sizeof(int) * G));
  }
};

template <typename R>
__global__ void __launch_bounds__(WAVE) sfw_social_kernel(const sfw_launch L, const int G) {
  using R2 = typename vec2<R>::type;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  const int A = L.A, O = L.O, S = L.S;
  const int GA = G * A;
  lds_layout<R> s(smem, A, GA, G, O);

  const int64_t first_local = static_cast<int64_t>(blockIdx.x) * G;  // first sample of this wave
  // number of real samples in this wave
  const int64_t remain = L.chunk_count - first_local;
  const int Gn = remain < G ? static_cast<int>(remain) : G;

  sfm_consts<R> k;
  k.lambda = R(L.p.sfm_lambda);
  k.gamma = R(L.p.sfm_gamma);
  k.inv_gamma = R(1.0 / L.p.sfm_gamma);
  k.n = R(L.p.sfm_n);
  k.n_prime = R(L.p.sfm_n_prime);
  k.f_social = R(L.p.sfm_force_factor_social);
  k.f_desired = R(L.p.sfm_force_factor_desired);
  k.inv_tau = R(1.0 / L.p.sfm_relaxation_time);
  k.f_obstacle = R(L.p.sfm_force_factor_obstacle);
  k.inv_sigma = R(1.0 / L.p.sfm_force_sigma_obstacle);
  k.dt = R(L.dt);
  k.rr = R(static_cast<double>(L.p.robot_radius * L.p.robot_radius));  // float product, ref :617
  const R inv_O = O > 0 ? R(1.0 / O) : R(0);

  // ---- stage constants + initial state ----------------------------------
  for (int i = lane; i < A; i += WAVE) {
    const sfw_agent_const c = L.agent_c[i];
    s.goal[i] = R2{R(c.goal_x), R(c.goal_y)};
    s.gr[i] = R(c.goal_radius);
    s.dv[i] = R(c.desired_velocity);
    s.rad[i] = R(c.radius);
    s.id[i] = c.id;
  }
  for (int o = lane; o < O; o += WAVE) s.obs[o] = R2{R(L.obstacles[2 * o]), R(L.obstacles[2 * o + 1])};
  if (lane < G) {
    int dead = 1;
    if (lane < Gn) dead = (L.status[L.chunk_begin + first_local + lane] != SFW_ST_VALID);
    s.dead[lane] = dead;
  }
  __syncthreads();
  {
    bool any_live = false;
    for (int g = 0; g < G; ++g) any_live |= (s.dead[g] == 0);
    if (!any_live) return;  // every sample of this wave was rejected by K1
  }
  for (int sl = lane; sl < GA; sl += WAVE) {
    const int g = (G == 1) ? 0 : sl / A;
    const int i = sl - g * A;
    const R px = R(L.agent_pos[2 * i]), py = R(L.agent_pos[2 * i + 1]);
    const R vx = R(L.agent_vel[2 * i]), vy = R(L.agent_vel[2 * i + 1]);
    const int hg = L.agent_c[i].has_goal;
    s.pos[sl] = R2{px, py};
    s.vel[sl] = R2{vx, vy};
    s.hasgoal[sl] = hg;
    R fx = R(0), fy = R(0);
    if (i != 0) {
      const R2 gl = s.goal[i];
      desired_force<R>(k, px, py, vx, vy, hg != 0, gl.x, gl.y, s.gr[i], s.dv[i], fx, fy);
      if (O > 0) {
        R ox, oy;
        obstacle_force<R, R2>(k, s.obs, O, inv_O, px, py, s.rad[i], ox, oy);
        fx += ox;
        fy += oy;
      }
    }
    s.frc[sl] = R2{fx, fy};
    s.frj[sl] = R2{R(0), R(0)};
  }
  __syncthreads();

  const int P = A * (A - 1) / 2;  // unordered pairs per sample
  const int GP = G * P;
  const float invA = 1.0f / static_cast<float>(A);
  const float invP = P > 0 ? 1.0f / static_cast<float>(P) : 0.0f;
  const int robot_id = s.id[0];
  double sw_acc = 0.0;  // this lane's share of the social work (one fixed sample per lane)

  for (int step = 0; step < S; ++step) {
    // ---- pair pass: social forces at the pre-step state ------------------
    // Items are the unordered pairs of every sample, flattened over the lanes:
    // u in [0,P) -> row = u / A, i = u % A, j = (i + row + 1) % A  (half ring).
    for (int t = lane; t < GP; t += WAVE) {
      int g = 0, u = t;
      if (G > 1) {
        g = static_cast<int>(static_cast<float>(t) * invP);
        u = t - g * P;
        if (u < 0) { --g; u += P; }
        if (u >= P) { ++g; u -= P; }
      }
      int row = static_cast<int>(static_cast<float>(u) * invA);
      int i = u - row * A;
      if (i < 0) { --row; i += A; }
      if (i >= A) { ++row; i -= A; }
      int j = i + row + 1;
      if (j >= A) j -= A;
      const int si = g * A + i, sj = g * A + j;
      const R2 pi = s.pos[si], pj = s.pos[sj], vi = s.vel[si], vj = s.vel[sj];
      R fx, fy;
      pair_force<R>(k, pi.x, pi.y, vi.x, vi.y, pj.x, pj.y, vj.x, vj.y, fx, fy);
      // Two accumulators per agent: what it receives as the pair's `i` and as its
      // `j`.  Each one then sums its contributions in item order whatever the
      // sample's position inside the wave, so a sample's result does not depend
      // on how samples are packed into waves (or sharded over GPUs).
      atomicAdd(&s.frc[si].x, fx);
      atomicAdd(&s.frc[si].y, fy);
      atomicAdd(&s.frj[sj].x, -fx);
      atomicAdd(&s.frj[sj].y, -fy);
    }
    __syncthreads();

    // ---- per-agent pass: integrate, collide, social work, next forces ----
    for (int sl = lane; sl < GA; sl += WAVE) {
      const int g = (G == 1) ? 0 : sl / A;
      const int i = sl - g * A;
      if (s.dead[g]) continue;
      const sfw_robot_step rs = L.rstep[static_cast<int64_t>(step) * L.rstep_stride + first_local + g];
      const R rx = R(rs.x), ry = R(rs.y), rvx = R(rs.vx), rvy = R(rs.vy);
      R2 F = s.frc[sl];
      {
        const R2 Fj = s.frj[sl];
        F.x += Fj.x;
        F.y += Fj.y;
        s.frj[sl] = R2{R(0), R(0)};
      }
      if (i == 0) {
        // Wr (ref :681-682): robot's social + obstacle force norms at the pre-step state
        R wr = m_sqrt<R>(F.x * F.x + F.y * F.y);
        if (O > 0) {
          const R2 p0 = s.pos[sl];
          R ox, oy;
          obstacle_force<R, R2>(k, s.obs, O, inv_O, p0.x, p0.y, s.rad[0], ox, oy);
          wr += m_sqrt<R>(ox * ox + oy * oy);
        }
        sw_acc += static_cast<double>(wr);
        s.pos[sl] = R2{rx, ry};      // ref :600
        s.vel[sl] = R2{rvx, rvy};    // ref :604 (robot-local twist)
        s.frc[sl] = R2{R(0), R(0)};
      } else {
        // lightsfm updatePosition, non-teleoperated branch
        const R2 p0 = s.pos[sl], v0 = s.vel[sl];
        R vx = v0.x + F.x * k.dt, vy = v0.y + F.y * k.dt;
        const R dv = s.dv[i];
        const R sp = m_sqrt<R>(vx * vx + vy * vy);
        if (sp > dv) {
          const R sc = dv / sp;  // normalize() then *= desiredVelocity
          vx *= sc;
          vy *= sc;
        }
        const R px = p0.x + vx * k.dt, py = p0.y + vy * k.dt;
        const R2 gl = s.goal[i];
        const R grad = s.gr[i];
        int hg = s.hasgoal[sl];
        if (hg) {
          const R ex = gl.x - px, ey = gl.y - py;
          if (m_sqrt<R>(ex * ex + ey * ey) <= grad) hg = 0;  // goal reached: pop
        }
        // dynamic collision with the robot's post-step pose (ref :613-627)
        const R cx = rx - px, cy = ry - py;
        if (cx * cx + cy * cy <= k.rr) s.dead[g] = 2;
        // Wp (ref :692-699): force the post-step robot alone exerts on this person
        if (s.id[i] != robot_id) {
          R qx, qy;
          pair_force<R>(k, px, py, vx, vy, rx, ry, rvx, rvy, qx, qy);
          sw_acc += static_cast<double>(m_sqrt<R>(qx * qx + qy * qy));
        }
        // desired + obstacle force at the new state = next step's starting force
        R fx, fy;
        desired_force<R>(k, px, py, vx, vy, hg != 0, gl.x, gl.y, grad, dv, fx, fy);
        if (O > 0) {
          R ox, oy;
          obstacle_force<R, R2>(k, s.obs, O, inv_O, px, py, s.rad[i], ox, oy);
          fx += ox;
          fy += oy;
        }
        s.pos[sl] = R2{px, py};
        s.vel[sl] = R2{vx, vy};
        s.frc[sl] = R2{fx, fy};
        s.hasgoal[sl] = hg;
      }
    }
    __syncthreads();
    bool any_live = false;
    for (int g = 0; g < G; ++g) any_live |= (s.dead[g] == 0);
    if (!any_live) break;
  }

  // ---- per-sample reduction of the social work, in lane order -----------
  // A lane owns slots lane, lane+64, ... which all belong to one sample when
  // G == 1, and exactly one slot when G > 1 (GA <= 64).
  if (G == 1) {
    double v = sw_acc;
    for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
    if (lane == 0) {
      const int64_t t = L.chunk_begin + first_local;
      const int d = s.dead[0];
      if (d == 0) L.costs[t] = L.base_cost[t] + L.p.social_weight * v;  // ref :663-667
      else if (d == 2) { L.costs[t] = SFW_COST_INVALID; L.status[t] = SFW_ST_INVALID; }
    }
  } else {
    if (lane < GA) s.swp[lane] = sw_acc;
    __syncthreads();
    if (lane < Gn) {
      double v = 0.0;
      for (int i = 0; i < A; ++i) v += s.swp[lane * A + i];
      const int64_t t = L.chunk_begin + first_local + lane;
      const int d = s.dead[lane];
      if (d == 0) L.costs[t] = L.base_cost[t] + L.p.social_weight * v;
      else if (d == 2) { L.costs[t] = SFW_COST_INVALID; L.status[t] = SFW_ST_INVALID; }
    }
  }
}

// ===========================================================================
// K3: argmin under the reference's selection order
// ===========================================================================
// Selectable (ref :394-404 with best_cost initialised to 10000.0 and best_traj
// to xv_=0, thetav_=0): cost >= 0 and (cost < 1e4, or cost == 1e4 and
// (linvel > 0 or (linvel == 0 and angvel == 0))).
// Order: cost up, linvel down, |angvel| up, iteration index down.
__device__ __forceinline__ bool sel_less(const sfw_sel &a, const sfw_sel &b) {
  if (a.cost != b.cost) return a.cost < b.cost;
  if (a.neg_linvel != b.neg_linvel) return a.neg_linvel < b.neg_linvel;
  if (a.abs_angvel != b.abs_angvel) return a.abs_angvel < b.abs_angvel;
  return a.neg_index < b.neg_index;
}
__device__ __forceinline__ sfw_sel sel_empty() {
  sfw_sel e;
  e.cost = INFINITY;
  e.neg_linvel = INFINITY;
  e.abs_angvel = INFINITY;
  e.neg_index = 0x7fffffffffffffffLL;
  e.n_valid = 0;
  return e;
}
__device__ __forceinline__ sfw_sel sel_merge(const sfw_sel &a, const sfw_sel &b) {
  sfw_sel r = sel_less(b, a) ? b : a;
  r.n_valid = a.n_valid + b.n_valid;
  return r;
}
__device__ __forceinline__ sfw_sel sel_shfl_down(const sfw_sel &a, int off) {
  sfw_sel r;
  r.cost = __shfl_down(a.cost, off, WAVE);
  r.neg_linvel = __shfl_down(a.neg_linvel, off, WAVE);
  r.abs_angvel = __shfl_down(a.abs_angvel, off, WAVE);
  r.neg_index = __shfl_down(a.neg_index, off, WAVE);
  r.n_valid = __shfl_down(a.n_valid, off, WAVE);
  return r;
}
constexpr int ARGMIN_BLOCK = 256;
__device__ __forceinline__ sfw_sel block_reduce(sfw_sel v) {
  __shared__ sfw_sel wave_best[ARGMIN_BLOCK / WAVE];
  for (int off = WAVE / 2; off > 0; off >>= 1) v = sel_merge(v, sel_shfl_down(v, off));
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
  if (lane == 0) wave_best[wid] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < ARGMIN_BLOCK / WAVE; ++w) v = sel_merge(v, wave_best[w]);
  }
  return v;  // valid in thread 0
}

__global__ void __launch_bounds__(ARGMIN_BLOCK)
sfw_argmin_stage1(const double *costs, const double *linvels, const double *angvels, int nw, int64_t T,
                  int64_t index_base, sfw_sel *partials) {
  sfw_sel best = sel_empty();
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < T;
       t += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const double c = costs[t];
    if (!(c >= 0.0)) continue;
    best.n_valid += 1;
    const double lin = linvels[t / nw], ang = angvels[t % nw];
    const bool selectable = c < 10000.0 || (c == 10000.0 && (lin > 0.0 || (lin == 0.0 && ang == 0.0)));
    if (!selectable) continue;
    sfw_sel cand;
    cand.cost = c;
    cand.neg_linvel = -lin;
    cand.abs_angvel = fabs(ang);
    cand.neg_index = -(index_base + t);
    cand.n_valid = 0;
    const long long nv = best.n_valid;
    if (sel_less(cand, best)) best = cand;
    best.n_valid = nv;
  }
  best = block_reduce(best);
  if (threadIdx.x == 0) partials[blockIdx.x] = best;
}
__global__ void __launch_bounds__(ARGMIN_BLOCK)
sfw_argmin_stage2(const sfw_sel *partials, int n, sfw_sel *out) {
  sfw_sel best = sel_empty();
  for (int i = threadIdx.x; i < n; i += blockDim.x) best = sel_merge(best, partials[i]);
  best = block_reduce(best);
  if (threadIdx.x == 0) *out = best;
}

}  // namespace

// ===========================================================================
// launchers
// ===========================================================================
int sfw_samples_per_wave(int A) {
  if (A <= 0) return 1;
  int g = WAVE / A;
  return g < 1 ? 1 : g;
}

size_t sfw_social_lds_bytes(int A, int O, int precision) {
  const size_t r = precision == SFW_PRECISION_F32 ? 4 : 8;
  const int G = sfw_samples_per_wave(A);
  const size_t GA = static_cast<size_t>(G) * A;
  auto up = [](size_t b) { return (b + 15) & ~size_t(15); };
  size_t n = 0;
  n += 4 * up(2 * r * GA);                 // pos, vel, frc, frj
  n += up(2 * r * A);                      // goal
  n += up(2 * r * (O > 0 ? O : 1));        // obs
  n += up(8 * GA);                         // swp
  n += 3 * up(r * A);                      // gr, dv, rad
  n += up(4 * A) + up(4 * GA) + up(4 * G); // id, hasgoal, dead
  return n;
}

hipError_t sfw_launch_rollout(const sfw_launch &L, hipStream_t stream) {
  if (L.chunk_count <= 0) return hipSuccess;
  {
    const int block = 64;  // latency-bound serial rollout: spread the waves over all CUs
    const unsigned grid = static_cast<unsigned>((L.chunk_count + block - 1) / block);
    hipLaunchKernelGGL(sfw_rollout_kernel, dim3(grid), dim3(block), 0, stream, L);
  }
  {
    const int block = 256;
    const int64_t n = L.chunk_count * L.S;
    const unsigned grid = static_cast<unsigned>((n + block - 1) / block);
    hipLaunchKernelGGL(sfw_footprint_kernel, dim3(grid), dim3(block), 0, stream, L);
  }
  {
    const int block = 256;
    const unsigned grid = static_cast<unsigned>((L.chunk_count + block - 1) / block);
    hipLaunchKernelGGL(sfw_costmap_scan_kernel, dim3(grid), dim3(block), 0, stream, L);
  }
  return hipGetLastError();
}

hipError_t sfw_launch_social(const sfw_launch &L, hipStream_t stream) {
  if (L.chunk_count <= 0 || L.A <= 0) return hipSuccess;
  const int G = sfw_samples_per_wave(L.A);
  const unsigned grid = static_cast<unsigned>((L.chunk_count + G - 1) / G);
  const size_t lds = sfw_social_lds_bytes(L.A, L.O, L.p.precision);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  if (L.p.precision == SFW_PRECISION_F32) {
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(sfw_social_kernel<float>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(sfw_social_kernel<float>, dim3(grid), dim3(WAVE), lds, stream, L, G);
  } else {
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(sfw_social_kernel<double>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(sfw_social_kernel<double>, dim3(grid), dim3(WAVE), lds, stream, L, G);
  }
  return hipGetLastError();
}

int64_t sfw_argmin_partials(int64_t T) {
  int64_t blocks = (T + ARGMIN_BLOCK - 1) / ARGMIN_BLOCK;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  return blocks;
}

hipError_t sfw_launch_argmin(const double *costs, const double *linvels, const double *angvels, int32_t nw,
                             int64_t T, int64_t index_base, sfw_sel *partials, sfw_sel *out,
                             hipStream_t stream) {
  const int blocks = static_cast<int>(sfw_argmin_partials(T));
  hipLaunchKernelGGL(sfw_argmin_stage1, dim3(blocks), dim3(ARGMIN_BLOCK), 0, stream, costs, linvels,
                     angvels, nw, T, index_base, partials);
  hipLaunchKernelGGL(sfw_argmin_stage2, dim3(1), dim3(ARGMIN_BLOCK), 0, stream, partials, blocks, out);
  return hipGetLastError();
}
