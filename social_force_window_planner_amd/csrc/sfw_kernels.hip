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
//       sfw_social_kernel_.. sample are integrated under the social-force model
//                            (lightsfm computeForces/updatePosition, reference
//                            call sites :592,:594,:697).  Two organisations,
//                            picked per agent count by plan_for(): register-
//                            resident agent slots walking the pairs as a half
//                            ring, or all pairs flattened over the lanes with
//                            the state in LDS.  Social work accumulated per
//                            lane and reduced per sample (reference :613-629,
//                            :678-705).  Group forces only in the GROUPS=true
//                            instantiations; the flat form exists with and
//                            without the laser-point pass (OBS); the whole file
//                            is compiled a second time with longer polynomials
//                            for SFW_PRECISION_F64_STRICT (sfw_kernels_strict.hip).
//   K3  sfw_argmin_*         block-wide + grid argmin under the reference's
//                            selection order (reference :394-414).
//
// The pair interaction is the hot loop (>= 90 % of the work for N >= 10).  It
// is a nonlinear function of a 2-vector pair (2 exp, atan2, 2 rsqrt), not a
// contraction: vector ALU work, no MFMA.

#include "sfw_device.h"
#include "sfw_math.h"

#include <type_traits>

#include <math.h>
#include <stdlib.h>
#include <cmath>

namespace {

constexpr int WAVE = 64;

// ===========================================================================
// K1: robot rollout + costmap
// ===========================================================================
// Contraction is switched off in K1 so that the pose arithmetic is the same
// IEEE sequence the CPU reference executes (cell-index truncation makes the
// last ulp matter, SURVEY.md §7 "discrete thresholds").
#pragma clang fp contract(off)

// (unsigned)(a / res) for a >= 0, bit-for-bit the IEEE division + truncation of
// nav2_costmap_2d::Costmap2D::worldToMap (Foxy), without paying for the
// ~14-instruction division 34 times per pose: a * (1/res) is within a few ulp of the quotient, so the
// truncations can only differ when the product lies within 1e-9 of an integer — then, and only then,
// the lane takes the exact division (grid-aligned coordinates; rare, and still exact).
__device__ __forceinline__ unsigned cell_index(double a, double res, double inv_res) {
  const double q = a * inv_res;
  const double m = __builtin_trunc(q);
  const double d = q - m;
  if (d < 1e-9 || d > 1.0 - 1e-9) return static_cast<unsigned>(a / res);
  return static_cast<unsigned>(m);
}

__device__ __forceinline__ bool world_to_map(const sfw_launch &L, double inv_res, double wx, double wy, unsigned &mx,
                                             unsigned &my) {
  if (wx < L.origin_x || wy < L.origin_y) return false;
  mx = cell_index(wx - L.origin_x, L.resolution, inv_res);
  my = cell_index(wy - L.origin_y, L.resolution, inv_res);
  return mx < L.size_x && my < L.size_y;
}

// reference src/costmap_model.cpp:112-121
__device__ __forceinline__ int point_code(const sfw_launch &L, int x, int y) {
  return L.cells[static_cast<size_t>(y) * L.size_x + static_cast<size_t>(x)];
}

// Bresenham walk of one footprint edge (reference line_iterator.hpp:39-97, src/costmap_model.cpp:95-110):
// the largest cell value on the edge.  The reference stops at the first lethal (254 -> -1) or unknown
// (255 -> -2) cell; a pose with such a cell on an edge is illegal whichever code it gets (ref :555-573
// reject footprint_cost >= 254 and < 0 alike), so the walk is branch-free and the caller classifies the
// maximum.  One linear cell index, major/minor steps in index units.
__device__ __forceinline__ int line_max(const sfw_launch &L, int x0, int y0, int x1, int y1) {
  const int dx = x1 - x0, dy = y1 - y0;
  const int adx = abs(dx), ady = abs(dy);
  const int stride = static_cast<int>(L.size_x);
  const int sx = (dx >= 0) ? 1 : -1, sy = (dy >= 0) ? stride : -stride;
  const bool x_major = adx >= ady;
  const int den = x_major ? adx : ady, add = x_major ? ady : adx;
  const int dmaj = x_major ? sx : sy, dmin = x_major ? sy : sx;
  int num = den / 2;
  unsigned idx = static_cast<unsigned>(y0 * stride + x0);
  int best = 0;
  for (int n = 0; n <= den; ++n) {
    best = max(best, static_cast<int>(L.cells[idx]));
    num += add;
    const bool wrap = num >= den;
    num -= wrap ? den : 0;
    idx += static_cast<unsigned>(dmaj + (wrap ? dmin : 0));
  }
  return best;
}

// reference world_model.hpp:45-75 + src/costmap_model.cpp:21-92; c,s = cos/sin(theta)
__device__ double footprint_cost(const sfw_launch &L, double x, double y, double c, double s) {
  const double inv_res = 1.0 / L.resolution;
  unsigned cx, cy;
  if (!world_to_map(L, inv_res, x, y, cx, cy)) return -3.0;
  const int K = L.K;
  if (K < 3) {
    const int code = point_code(L, (int)cx, (int)cy);
    if (code == 255) return -2.0;
    if (code == 254 || code == 253) return -1.0;
    return static_cast<double>(code);
  }
  // Every footprint vertex is the end of one edge and the start of the next, so
  // its cell is computed once (the reference converts it twice, same result).
  // Any vertex off the map makes the pose illegal (-3); which illegal code wins
  // when several apply does not matter to scoreTrajectory (all map to -1.0).
  int fc = 0;
  double qx = L.footprint[0], qy = L.footprint[1];
  unsigned fx0, fy0;
  if (!world_to_map(L, inv_res, x + (qx * c - qy * s), y + (qx * s + qy * c), fx0, fy0)) return -3.0;
  unsigned x0 = fx0, y0 = fy0;
  for (int e = 0; e < K; ++e) {
    unsigned x1, y1;
    if (e + 1 < K) {
      qx = L.footprint[2 * (e + 1)];
      qy = L.footprint[2 * (e + 1) + 1];
      if (!world_to_map(L, inv_res, x + (qx * c - qy * s), y + (qx * s + qy * c), x1, y1)) return -3.0;
    } else {  // closing edge: back() -> front()
      x1 = fx0;
      y1 = fy0;
    }
    fc = max(fc, line_max(L, (int)x0, (int)y0, (int)x1, (int)y1));
    x0 = x1;
    y0 = y1;
  }
  return static_cast<double>(fc);  // >= 254 (lethal / unknown on an edge): illegal, like the reference's -1 / -2
}

// reference sfw_planner.hpp:457-463: if ((vg - vi) >= 0) return min(vg, vi + a_max dt); return max(vg, vi - a_max dt);
// Both candidates and a select instead of the branch: the recurrence is the serial chain at the head of every launch (one
// lane walks S steps), and a divergent branch costs such a lane ~100 cycles per step where the select costs two
// instructions (profiles/r05_k1small_ablation.txt).  Same operations on the same values: bit-identical.
__device__ __forceinline__ double new_velocity(double vg, double vi, double a_max, double dt) {
  const double step = a_max * dt;
  const double up = fmin(vg, vi + step), dn = fmax(vg, vi - step);
  return (vg - vi) >= 0 ? up : dn;
}
// reference sfw_planner.hpp:399-407 (float in, float out)
__device__ __forceinline__ float normalize_angle_f(float val, float mn, float mx) {
  if (val >= mn) return mn + fmodf(val - mn, mx - mn);
  return mx - fmodf(mn - val, mx - mn);
}

// ---- the K1 tables (sfw_device.h sfw_unit) ---------------------------------------------------------------------------
// grid row of a chunk-local sample, counted from the chunk's first row (the velocity unit of its ptab row)
__device__ __forceinline__ int64_t vel_row_of(const sfw_launch &L, int64_t local) {
  return (L.chunk_begin + local) / L.nw - L.chunk_begin / L.nw;
}
// One step of one sample as the pose rollout leaves it: the post-step position always; the step's velocity by the first
// sample of every grid row inside the chunk, cos / sin of the pre-step heading by the chunk's first nw samples (one per
// column) — the same values whoever writes them.
__device__ __forceinline__ void put_robot_step(const sfw_launch &L, int64_t local, int i, double xn, double yn, double vx,
                                               double vy, double c, double s) {
  sfw_unit *const row = L.ptab + static_cast<int64_t>(i) * L.row_units;
  row[local] = sfw_unit{xn, yn};
  const int64_t t = L.chunk_begin + local;
  if (local == 0 || t % L.nw == 0) row[L.rstep_stride + vel_row_of(L, local)] = sfw_unit{vx, vy};
  if (local < L.nw) L.cs_tab[static_cast<int64_t>(i) * L.nw + t % L.nw] = sfw_unit{c, s};
}

// Sequential pose integration of one sample (reference :527-:611 without the costmap and the
// pedestrians): writes the K1 tables (put_robot_step) and the pedestrian-independent cost
// terms.  Returns false for the never-scored (0,0) sample (whose records are written all the same).
__device__ __forceinline__ bool rollout_sample(const sfw_launch &L, int64_t local) {
  const int64_t t = L.chunk_begin + local;
  const int iv = static_cast<int>(t / L.nw), iw = static_cast<int>(t % L.nw);
  const double vx_samp = L.linvels[iv], vth_samp = L.angvels[iw], vy_samp = L.vy_samp;
  // The never-scored (0,0) sample (ref :349-352) still gets its robot-step records: it can be the
  // representative of a shared-prefix class (sfw_cls_agent) whose other members are scored.
  const bool scored = !(L.skip_zero_sample && vx_samp == 0.0 && vth_samp == 0.0);
  if (!scored) {
    L.status[t] = SFW_ST_SKIPPED;
    L.costs[t] = SFW_COST_SKIPPED;
  } else {
    L.status[t] = SFW_ST_VALID;  // the costmap scan downgrades it if a step is illegal
  }
  if (L.coll_step) L.coll_step[t] = -1;
  double x_i = L.rs.x, y_i = L.rs.y, th_i = L.rs.theta;
  double vx_i = L.rs.vx, vy_i = L.rs.vy, vth_i = L.rs.vtheta;
  const int S = L.S;
  const double dt = L.dt;
  for (int i = 0; i < S; ++i) {
    double s, c;
    sincos(th_i, &s, &c);
    if (L.points) {                                       // ref :578
      double *pt = L.points + (local * S + i) * 3;
      pt[0] = x_i;
      pt[1] = y_i;
      pt[2] = th_i;
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
    put_robot_step(L, local, i, x_i, y_i, vx_i, vy_i, c, s);
  }
  // ref :643-666 without the costmap and social terms (left-to-right sum order kept)
  const double dx = L.ga.wpx - x_i, dy = L.ga.wpy - y_i;
  const double d = dx * dx + dy * dy;
  double ang = atan2(dy, dx) - th_i;
  ang = normalize_angle_f(static_cast<float>(ang), static_cast<float>(-M_PI), static_cast<float>(M_PI));
  ang = fabs(ang) / M_PI;
  const double vel = fabs(L.p.max_vel_x - vx_i) / L.p.max_vel_x;
  L.base_cost[t] = L.p.vel_weight * vel + L.p.distance_weight * d + L.p.angle_weight * ang;
  return scored;
}

// In-order consumption of a sample's per-step footprint costs (n_ok legal steps so far, running
// sum cm): first illegal step rejects the trajectory (ref :555-573), otherwise costmap_cost
// accumulates in step order (ref :575, :656).
__device__ __forceinline__ bool scan_code(double fc, double &cm, int &n_ok) {
  if (fc >= 254.0 || fc < 0) return false;
  cm += fc / 255.0;
  ++n_ok;
  return true;
}
// Returns whether every pose was legal; *base_out (nullable) receives the pedestrian-free cost of a legal trajectory.
__device__ __forceinline__ bool scan_finish(const sfw_launch &L, int64_t t, int64_t local, double cm, int n_ok,
                                            double *base_out = nullptr, const double *base_in = nullptr) {
  const int S = L.S;
  if (L.n_points) L.n_points[local] = n_ok;
  if (n_ok < S) {
    L.status[t] = SFW_ST_INVALID;
    L.costs[t] = SFW_COST_INVALID;
    return false;
  }
  cm = cm / S;
  const double base = (base_in ? *base_in : L.base_cost[t]) + L.p.costmap_weight * cm;
  L.base_cost[t] = base;
  if (base_out) *base_out = base;
  // No agent vector at all: social work is identically 0 and K2 is not launched.
  if (L.A == 0) L.costs[t] = base + L.p.social_weight * 0.0;
  return true;
}

// K1a: one thread per sample.
// (a timed launch's clock probe is cleared here, by the kernel every K2 dispatch of the launch is ordered behind: a memset of
// its own in front of the rollout was a 4 us operation with its 6 us gap at the head of every timed step)
__device__ __forceinline__ void clear_clock_probe(const sfw_launch &L) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && L.clock_probe) L.clock_probe[0] = L.clock_probe[1] = L.clock_probe[2] = L.clock_probe[3] = 0ull;
}
// The stage's arena from the host's pinned memory to its place on the device, by the kernel that runs first in a GPU-filling
// launch (sfw_launch.arena_host: the stage enqueued no H2D copy): 16 bytes per thread, the whole grid's threads side by side —
// one round for a target-sized arena.  This kernel itself reads its sample vectors through the host addresses it was given
// (L.linvels / L.angvels point into the pinned arena for this launch); every later kernel reads the device copy.
__device__ __forceinline__ void fetch_arena(const sfw_launch &L) {
  if (!L.arena_host) return;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 *const src = reinterpret_cast<const u32x4 *>(L.arena_host);
  u32x4 *const dst = reinterpret_cast<u32x4 *>(L.arena_dev);
  const uint32_t n = L.arena_bytes / 16, stride = gridDim.x * blockDim.x;
  for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < n; u += stride) dst[u] = __builtin_nontemporal_load(src + u);
}
__global__ void __launch_bounds__(64) sfw_rollout_kernel(const sfw_launch L) {
  clear_clock_probe(L);
  fetch_arena(L);
  const int64_t local = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (local >= L.chunk_count) return;
  rollout_sample(L, local);
}

// K1a again, for GPU-filling grids: TEAMS of eight lanes per sample, eight samples per wave.  One thread per sample leaves a
// lone wave per SIMD (cfg2: 256 waves on 1024 SIMDs) walking S serial steps, each with a library sincos — 19 us at the head
// of cfg2's blocking call, 31 us at the target configuration's (profiles/r05_step_timeline_cfg2.txt).  As in the small-grid
// kernel below only the true recurrences stay serial; the steps are taken in rounds of eight, one step per lane of the team:
//   (1) lanes 0, 1, 2 of a team: the round's eight steps of the three velocity recurrences (computeNewVelocity, ref :581-583),
//       lane 2 also the heading sum (ref :588) — ONE loop for the three lanes;
//   (2) lane q: sincos of step q's heading and the position increments (vx cos + vy cos(pi/2 + th)) dt (ref :586-587);
//   (3) lanes 0, 1: the two position sums x += dx_q, y += dy_q over the round;
//   (4) lane q: step q's units of the K1 tables (put_robot_step), the Trajectory point.
// Eight times the waves (two to eight per SIMD), the sincos of eight steps side by side.  The same operations on the same
// values in the same order as rollout_sample (no contraction here either): bit-identical tables and cost terms
// (tests/test_prefix_sharing_gpu.py::test_team_rollout_equals_the_thread_rollout).
constexpr int K1A_TEAM = 8;
__global__ void __launch_bounds__(64) sfw_rollout_team_kernel(const sfw_launch L) {
  constexpr int TEAMS = WAVE / K1A_TEAM;
  clear_clock_probe(L);
  fetch_arena(L);
  __shared__ double r_th[TEAMS][K1A_TEAM], r_vx[TEAMS][K1A_TEAM], r_vy[TEAMS][K1A_TEAM];  // [team][step of the round]
  __shared__ double2 r_inc[TEAMS][K1A_TEAM];
  __shared__ double r_px[TEAMS][K1A_TEAM + 1], r_py[TEAMS][K1A_TEAM + 1];                  // pose before step q; [nst]: after the round
  __shared__ double r_end[TEAMS][4];
  const int lane = threadIdx.x, g = lane / K1A_TEAM, j = lane % K1A_TEAM;
  const int64_t local_raw = static_cast<int64_t>(blockIdx.x) * TEAMS + g;
  const bool live = local_raw < L.chunk_count;
  const int64_t local = live ? local_raw : L.chunk_count - 1;  // a team past the end repeats the last sample and stores nothing
  const int64_t t = L.chunk_begin + local;
  const int iv = static_cast<int>(t / L.nw), iw = static_cast<int>(t % L.nw);
  const double vx_samp = L.linvels[iv], vth_samp = L.angvels[iw], vy_samp = L.vy_samp;
  const bool scored = !(L.skip_zero_sample && vx_samp == 0.0 && vth_samp == 0.0);
  if (live && j == 0) {
    if (!scored) {
      L.status[t] = SFW_ST_SKIPPED;
      L.costs[t] = SFW_COST_SKIPPED;
    } else {
      L.status[t] = SFW_ST_VALID;  // the costmap scan downgrades it if a step is illegal
    }
    if (L.coll_step) L.coll_step[t] = -1;
  }
  const int S = L.S;
  const double dt = L.dt;
  // lanes 0, 1, 2: target, limit and running velocity of "their" recurrence; lane 2 the heading; lanes 0, 1 the position
  const double target = j == 0 ? vx_samp : j == 1 ? vy_samp : vth_samp;  // ref :581-583
  const double a_max = j == 0 ? L.ga.acc_x : j == 1 ? L.ga.acc_y : L.ga.acc_theta;
  double v = j == 0 ? L.rs.vx : j == 1 ? L.rs.vy : L.rs.vtheta;
  double th_i = L.rs.theta;
  double p = j == 0 ? L.rs.x : L.rs.y;
  double *const vel_out = j == 0 ? r_vx[g] : j == 1 ? r_vy[g] : r_th[g];
  double *const pos_out = j == 0 ? r_px[g] : r_py[g];
  for (int base = 0; base < S; base += K1A_TEAM) {
    const int nst = min(K1A_TEAM, S - base);
    if (j < 3) {
      for (int q = 0; q < nst; ++q) {
        v = new_velocity(target, v, a_max, dt);
        vel_out[q] = j == 2 ? th_i : v;  // lane 2: the heading BEFORE this step's update (ref :586-588 integrate with the old theta)
        th_i = th_i + v * dt;            // (meaningful on lane 2 only)
      }
    }
    __syncthreads();
    double s = 0.0, c = 0.0, vxq = 0.0, vyq = 0.0, thq = 0.0;
    if (j < nst) {
      thq = r_th[g][j];
      vxq = r_vx[g][j];
      vyq = r_vy[g][j];
      double c2 = 0.0, s2 = 0.0;
      sincos(thq, &s, &c);
      if (vyq != 0.0) sincos(M_PI_2 + thq, &s2, &c2);  // holonomic term, 0 for the grid
      r_inc[g][j] = double2{(vxq * c + vyq * c2) * dt, (vxq * s + vyq * s2) * dt};  // ref :586-587 (old theta)
    }
    __syncthreads();
    if (j < 2) {
      const double *const inc = reinterpret_cast<const double *>(r_inc[g]) + j;  // .x or .y of every increment
      pos_out[0] = p;
      double d[K1A_TEAM];  // (all the round's increments first, then the dependent additions)
#pragma unroll
      for (int q = 0; q < K1A_TEAM; ++q) d[q] = inc[2 * q];
#pragma unroll
      for (int q = 0; q < K1A_TEAM; ++q)
        if (q < nst) pos_out[q + 1] = p = p + d[q];
    }
    __syncthreads();
    if (j < nst && live) {
      const int i = base + j;
      if (L.points) {                                       // ref :578
        double *pt = L.points + (local * S + i) * 3;
        pt[0] = r_px[g][j];
        pt[1] = r_py[g][j];
        pt[2] = thq;
      }
      put_robot_step(L, local, i, r_px[g][j + 1], r_py[g][j + 1], vxq, vyq, c, s);
    }
    // (no barrier here: the next round's phase (1) writes r_th / r_vx / r_vy, which this round read in front of its second
    // barrier, and r_px / r_py are rewritten behind two more)
  }
  if (j < 3) r_end[g][j] = j == 2 ? th_i : p;  // x_S, y_S, th_S
  if (j == 0) r_end[g][3] = v;                  // vx of the last step
  __syncthreads();
  if (j == 0 && live) {
    // ref :643-666 without the costmap and social terms (left-to-right sum order kept)
    const double dx = L.ga.wpx - r_end[g][0], dy = L.ga.wpy - r_end[g][1];
    const double d = dx * dx + dy * dy;
    double ang = atan2(dy, dx) - r_end[g][2];
    ang = normalize_angle_f(static_cast<float>(ang), static_cast<float>(-M_PI), static_cast<float>(M_PI));
    ang = fabs(ang) / M_PI;
    const double vel = fabs(L.p.max_vel_x - r_end[g][3]) / L.p.max_vel_x;
    L.base_cost[t] = L.p.vel_weight * vel + L.p.distance_weight * d + L.p.angle_weight * ang;
  }
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
  // the pose before the step: where the step before it ended (the handed-over pose for step 0), heading by column
  const sfw_unit pos = step == 0 ? sfw_unit{L.rs.x, L.rs.y} : L.ptab[(step - 1) * L.row_units + local];
  const sfw_unit cs = L.cs_tab[step * L.nw + (L.chunk_begin + local) % L.nw];
  const double fc = footprint_cost(L, pos.a, pos.b, cs.a, cs.b);  // includes the ref :545 map check
  L.fcode[step * L.rstep_stride + local] = static_cast<int16_t>(fc);
}

// K1c: in-order scan of the per-step footprint costs, one thread per sample.
__global__ void __launch_bounds__(256) sfw_costmap_scan_kernel(const sfw_launch L) {
  const int64_t local = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (local >= L.chunk_count) return;
  const int64_t t = L.chunk_begin + local;
  if (L.status[t] == SFW_ST_SKIPPED) {
    if (L.n_points) L.n_points[local] = 0;
    return;
  }
  const int S = L.S;
  double cm = 0.0;
  int n_ok = 0;
  bool stopped = false;
  // codes are fetched 8 at a time (independent loads in flight) and consumed in step order
  for (int base = 0; base < S && !stopped; base += 8) {
    int16_t v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      v[j] = (base + j < S) ? L.fcode[static_cast<int64_t>(base + j) * L.rstep_stride + local] : int16_t(-1);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (!stopped && base + j < S) stopped = !scan_code(static_cast<double>(v[j]), cm, n_ok);
  }
  scan_finish(L, t, local, cm, n_ok);
}

// One footprint edge of a pose: the largest cell value on edge e (vertex e -> vertex e+1, the last edge closes the
// polygon), or FOOT_OFFMAP when one of its vertices is off the map.  The same expressions as footprint_cost.
constexpr int FOOT_OFFMAP = 300;
__device__ __forceinline__ int footprint_edge(const sfw_launch &L, double x, double y, double c, double s, int e) {
  const double inv_res = 1.0 / L.resolution;
  const int e1 = (e + 1 < L.K) ? e + 1 : 0;
  const double ax = L.footprint[2 * e], ay = L.footprint[2 * e + 1], bx = L.footprint[2 * e1], by = L.footprint[2 * e1 + 1];
  unsigned x0, y0, x1, y1;
  if (!world_to_map(L, inv_res, x + (ax * c - ay * s), y + (ax * s + ay * c), x0, y0)) return FOOT_OFFMAP;
  if (!world_to_map(L, inv_res, x + (bx * c - by * s), y + (bx * s + by * c), x1, y1)) return FOOT_OFFMAP;
  return line_max(L, (int)x0, (int)y0, (int)x1, (int)y1);
}

// K1 for small grids (a control cycle of the reference's own 5 x 9 samples): the three stages in one launch, one
// 256-thread block per sample, two kernel boundaries fewer on the latency path.  A single wave is latency-bound, so
// everything that is not a true recurrence runs across the lanes:
//   (1) the three velocity recurrences (computeNewVelocity, ref :581-583) on lanes 0, 1, 2, the heading sum with the
//       angular one (ref :588);
//   (2) one step per lane: sincos of the heading, the position increments (vx cos + vy cos(pi/2 + th)) dt (ref :586-587);
//   (3) the two position sums x += dx_i, y += dy_i on lanes 0 and 1 — 40 dependent adds each instead of 40 whole steps;
//   (4) one step per lane: robot-step records, Trajectory points; lane 0: the pedestrian-free cost terms;
//   (5) one (pose, footprint edge) per lane, combined per pose with an LDS max; (6) lane 0 scans the codes in step order.
// The same operations on the same values in the same order as the one-thread-per-sample K1a + K1b + K1c (no
// contraction here either): bit-identical poses, cells, codes and costs.
// The phases are functions of (thread index, thread count) over one sample's LDS arrays (k1s_lds): sfw_rollout_small_kernel
// runs them with its 256 threads and a block barrier between them; sfw_cycle_kernel (below, behind K2) spreads them over
// its waves — the recurrences on one wave while another stages the pedestrians, the footprint tasks beside the pedestrian
// rollout.
constexpr int K1_SMALL_MAX_STEPS = 512;
constexpr int K1_SMALL_BLOCK = 256;  // four waves per sample: the (pose, edge) footprint tasks are the bulk of the work
struct k1s_lds {
  double *th, *vxs, *vys;  // [S] heading before step i, velocities after it
  double2 *cs, *dxy;       // [S] cos / sin of the heading, position increments
  double *xs, *ys;         // [S + 1] pose before step i; [S]: the final pose
  int *code;               // [S] footprint code of pose i (maximum over its edges)
  double *quot;            // [S] code / 255.0 (ref :575), formed across the lanes
  double *th_end;          // [1]
  // carved out of `base` (16-byte aligned) for S steps; base = nullptr sizes the allocation
  __host__ __device__ k1s_lds(char *base, int S, size_t *bytes = nullptr) {
    char *const base0 = base;
    auto take = [&](size_t n) {
      char *p = base;
      base += (n + 15) & ~size_t(15);
      return p;
    };
    const size_t n = static_cast<size_t>(S);
    th = reinterpret_cast<double *>(take(8 * n));
    vxs = reinterpret_cast<double *>(take(8 * n));
    vys = reinterpret_cast<double *>(take(8 * n));
    cs = reinterpret_cast<double2 *>(take(16 * n));
    dxy = reinterpret_cast<double2 *>(take(16 * n));
    xs = reinterpret_cast<double *>(take(8 * (n + 1)));
    ys = reinterpret_cast<double *>(take(8 * (n + 1)));
    code = reinterpret_cast<int *>(take(4 * n));
    quot = reinterpret_cast<double *>(take(8 * n));
    th_end = reinterpret_cast<double *>(take(8));
    if (bytes) *bytes = static_cast<size_t>(base - base0);
  }
};
struct k1s_sample {
  int64_t local, t;
  double vx_samp, vy_samp, vth_samp;
  bool scored;
};
__device__ __forceinline__ k1s_sample k1s_sample_of(const sfw_launch &L, int64_t local) {
  k1s_sample q;
  q.local = local;
  q.t = L.chunk_begin + local;
  const int iv = static_cast<int>(q.t / L.nw), iw = static_cast<int>(q.t % L.nw);
  q.vx_samp = L.linvels[iv];
  q.vth_samp = L.angvels[iw];
  q.vy_samp = L.vy_samp;
  q.scored = !(L.skip_zero_sample && q.vx_samp == 0.0 && q.vth_samp == 0.0);
  return q;
}
// the sample's status words (one thread)
__device__ __forceinline__ void k1s_head(const sfw_launch &L, const k1s_sample &q) {
  if (!q.scored) {
    L.status[q.t] = SFW_ST_SKIPPED;
    L.costs[q.t] = SFW_COST_SKIPPED;
  } else {
    L.status[q.t] = SFW_ST_VALID;  // the costmap scan downgrades it if a step is illegal
  }
  if (L.coll_step) L.coll_step[q.t] = -1;
}
// The three recurrences' running values of one thread (velocity / heading on threads 0-2, position on threads 0-1), so that
// the phases can be run over a RANGE of steps and resumed (the cycle kernel hands the pedestrians' wave the first steps early)
struct k1s_state {
  double v, th, p;
};
__device__ __forceinline__ k1s_state k1s_begin(const sfw_launch &L, int tid) {
  k1s_state st;
  st.v = tid == 0 ? L.rs.vx : tid == 1 ? L.rs.vy : L.rs.vtheta;
  st.th = L.rs.theta;
  st.p = tid == 0 ? L.rs.x : L.rs.y;
  return st;
}
// (1) velocities and headings of steps [i0, i1): threads 0, 1, 2 walk ONE loop, each with its own target, velocity and limit
// (as three branches of an if the wave ran the three recurrences one after the other); thread 2 also sums the heading
__device__ __forceinline__ void k1s_velocities(const sfw_launch &L, const k1s_lds &a, const k1s_sample &q, int i0, int i1, int S,
                                               int tid, k1s_state &st) {
  if (tid >= 3) return;
  const double dt = L.dt;
  const double target = tid == 0 ? q.vx_samp : tid == 1 ? q.vy_samp : q.vth_samp;  // ref :581-583
  const double a_max = tid == 0 ? L.ga.acc_x : tid == 1 ? L.ga.acc_y : L.ga.acc_theta;
  double v = st.v, th_i = st.th;
  double *const out = tid == 0 ? a.vxs : tid == 1 ? a.vys : a.th;
  for (int i = i0; i < i1; ++i) {
    v = new_velocity(target, v, a_max, dt);
    // threads 0, 1: the new velocity; thread 2: the heading BEFORE this step's update (ref :586-588 integrate with the old theta)
    out[i] = tid == 2 ? th_i : v;
    th_i = th_i + v * dt;  // (meaningful on thread 2 only)
  }
  st.v = v;
  st.th = th_i;
  if (tid == 2 && i1 == S) *a.th_end = th_i;
}
// (2) sines and position increments of steps [i0, i1), one step per thread
__device__ __forceinline__ void k1s_increments(const sfw_launch &L, const k1s_lds &a, int i0, int i1, int tid, int nthr) {
  const double dt = L.dt;
  for (int i = i0 + tid; i < i1; i += nthr) {
    double s, c, c2 = 0.0, s2 = 0.0;
    sincos(a.th[i], &s, &c);
    if (a.vys[i] != 0.0) sincos(M_PI_2 + a.th[i], &s2, &c2);  // holonomic term, 0 for the grid
    a.cs[i] = double2{c, s};
    a.dxy[i] = double2{(a.vxs[i] * c + a.vys[i] * c2) * dt, (a.vxs[i] * s + a.vys[i] * s2) * dt};  // ref :586-587
    a.code[i] = 0;
  }
}
// (3) positions: xs[i] = pose before step i, xs[S] = final pose (threads 0 and 1 in ONE loop, as above), steps [i0, i1)
__device__ __forceinline__ void k1s_positions(const sfw_launch &L, const k1s_lds &a, int i0, int i1, int tid, k1s_state &st) {
  if (tid >= 2) return;
  double p = st.p;
  double *const out = tid == 0 ? a.xs : a.ys;
  const double *const inc = reinterpret_cast<const double *>(a.dxy) + tid;  // .x or .y of every increment
  if (i0 == 0) out[0] = p;
  // eight increments at a time: all eight LDS reads first, then the eight dependent additions (left as one read per addition
  // the loop was an LDS round trip per step: 40 x ~130 cycles on the path every K2 wave of a control cycle waits for)
  for (int base = i0; base < i1; base += 8) {
    double d[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] = inc[2 * min(base + j, i1 - 1)];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (base + j < i1) out[base + j + 1] = p = p + d[j];
  }
  st.p = p;
}
// (4) records, one step per thread: Trajectory points (ref :578) and — for a K2 that reads them from memory — the robot steps
template <bool TABLE>
__device__ __forceinline__ void k1s_records(const sfw_launch &L, const k1s_lds &a, const k1s_sample &q, int S, int tid, int nthr) {
  for (int i = tid; i < S; i += nthr) {
    if (L.points) {
      double *pt = L.points + (q.local * S + i) * 3;
      pt[0] = a.xs[i];
      pt[1] = a.ys[i];
      pt[2] = a.th[i];
    }
    if constexpr (TABLE) put_robot_step(L, q.local, i, a.xs[i + 1], a.ys[i + 1], a.vxs[i], a.vys[i], a.cs[i].x, a.cs[i].y);
  }
}
// ... and the pedestrian-free cost terms (one thread): ref :643-666 without the costmap and social terms (left-to-right sum order kept)
__device__ __forceinline__ double k1s_base_cost_value(const sfw_launch &L, const k1s_lds &a, int S) {
  const double dx = L.ga.wpx - a.xs[S], dy = L.ga.wpy - a.ys[S];
  const double d = dx * dx + dy * dy;
  double ang = atan2(dy, dx) - *a.th_end;
  ang = normalize_angle_f(static_cast<float>(ang), static_cast<float>(-M_PI), static_cast<float>(M_PI));
  ang = fabs(ang) / M_PI;
  const double vel = fabs(L.p.max_vel_x - a.vxs[S - 1]) / L.p.max_vel_x;
  return L.p.vel_weight * vel + L.p.distance_weight * d + L.p.angle_weight * ang;
}
__device__ __forceinline__ void k1s_base_cost(const sfw_launch &L, const k1s_lds &a, const k1s_sample &q, int S) {
  L.base_cost[q.t] = k1s_base_cost_value(L, a, S);
}
// (5) footprint: the pose centre must be on the map (ref :545, src/costmap_model.cpp:36-37); K < 3: centre cell
// only; else every (pose, edge) is a task of its own and a pose's code is the maximum over its tasks
__device__ __forceinline__ void k1s_footprint(const sfw_launch &L, const k1s_lds &a, int S, int tid, int nthr) {
  const int K = L.K;
  if (K < 3) {
    for (int i = tid; i < S; i += nthr) {
      const double2 cs = a.cs[i];
      const double fc = footprint_cost(L, a.xs[i], a.ys[i], cs.x, cs.y);
      a.code[i] = fc < 0 ? FOOT_OFFMAP : static_cast<int>(fc);
    }
    return;
  }
  const double inv_res = 1.0 / L.resolution;
  for (int task = tid; task < S * K; task += nthr) {
    const int i = task / K, e = task - i * K;
    const double2 cs = a.cs[i];
    int v = footprint_edge(L, a.xs[i], a.ys[i], cs.x, cs.y, e);
    if (e == 0) {
      unsigned cx, cy;
      if (!world_to_map(L, inv_res, a.xs[i], a.ys[i], cx, cy)) v = FOOT_OFFMAP;
    }
    atomicMax(&a.code[i], v);
  }
}
// (6) in-order scan (scan_code, K1c): the first illegal step rejects, else the sum of code / 255.0 in step order.  The
// divisions — ~15 dependent instructions each — are formed one step per thread; one thread only adds, eight quotients at a time
__device__ __forceinline__ void k1s_quotients(const k1s_lds &a, int S, int tid, int nthr) {
  for (int i = tid; i < S; i += nthr) a.quot[i] = static_cast<double>(a.code[i]) / 255.0;
}
// base_in (nullable): the pedestrian-free cost terms when the caller holds them (else L.base_cost[t], as K1c reads them)
__device__ __forceinline__ bool k1s_scan(const sfw_launch &L, const k1s_lds &a, const k1s_sample &q, int S, double *base_out = nullptr,
                                         const double *base_in = nullptr) {
  double cm = 0.0;
  int n_ok = 0;
  bool stopped = false;
  for (int base = 0; base < S && !stopped; base += 8) {
    int c[8];
    double qt[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      c[j] = a.code[min(base + j, S - 1)];
      qt[j] = a.quot[min(base + j, S - 1)];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (!stopped && base + j < S) {
        if (c[j] >= 254 || c[j] < 0) {
          stopped = true;
        } else {
          cm += qt[j];
          ++n_ok;
        }
      }
  }
  return scan_finish(L, q.t, q.local, cm, n_ok, base_out, base_in);
}

__global__ void __launch_bounds__(K1_SMALL_BLOCK) sfw_rollout_small_kernel(const sfw_launch L) {
  __shared__ __attribute__((aligned(16))) char k1s_area[8 * K1_SMALL_MAX_STEPS * 3 + 16 * K1_SMALL_MAX_STEPS * 2 +
                                                        8 * (K1_SMALL_MAX_STEPS + 2) * 2 + 4 * K1_SMALL_MAX_STEPS +
                                                        8 * K1_SMALL_MAX_STEPS + 16];
  const int S = L.S;
  const k1s_lds a(k1s_area, K1_SMALL_MAX_STEPS);
  const k1s_sample q = k1s_sample_of(L, blockIdx.x);
  const int tid = threadIdx.x, nthr = blockDim.x;
  if (tid == 0) k1s_head(L, q);
  k1s_state st = k1s_begin(L, tid);
  k1s_velocities(L, a, q, 0, S, S, tid, st);
  __syncthreads();
  k1s_increments(L, a, 0, S, tid, nthr);
  __syncthreads();
  k1s_positions(L, a, 0, S, tid, st);
  __syncthreads();
  k1s_records<true>(L, a, q, S, tid, nthr);
  if (tid == 0) k1s_base_cost(L, a, q, S);
  if (!q.scored) {
    if (tid == 0 && L.n_points) L.n_points[q.local] = 0;
    return;
  }
  k1s_footprint(L, a, S, tid, nthr);
  __syncthreads();
  k1s_quotients(a, S, tid, nthr);
  __syncthreads();
  if (tid == 0) k1s_scan(L, a, q, S);
}

#pragma clang fp contract(fast)

// ===========================================================================
// K2: social-force integration, one wave per G samples
// ===========================================================================
// Agent STATE (position, velocity), its integration and every discrete
// threshold (contact, goal pop, speed clamp) are always double.  The template
// type R is the type the FORCES are evaluated in: double (parity mode) or float
// (fast mode: differences are formed in double, then rounded to float).
template <typename R> struct tiny_of;
template <> struct tiny_of<double> { static constexpr double v = 1e-300; };
template <> struct tiny_of<float> { static constexpr float v = 1e-30f; };

// XCD-aware work assignment (MI355X: 8 XCDs with private L2s, block b is observed
// to run on XCD b % 8 — used for locality only).  Consecutive sample groups share
// 128-byte lines of the K1->K2 robot-step table (4 records per line); mapping
// XCD x to the contiguous range [x*q + min(x,r), ...) of groups makes the blocks
// that share a line hit the same L2.  Bijective for any grid size and any XCD count
// (nx = sfw_launch.n_xcd: 8 on a whole MI355X, 1 on a CPX partition — the identity).
__device__ __forceinline__ unsigned xcd_contiguous(unsigned b, unsigned n, unsigned nx) {
  nx = nx ? nx : 1u;
  const unsigned x = b % nx, k = b / nx;
  const unsigned q = n / nx, r = n % nx;
  return x * q + (x < r ? x : r) + k;
}

// The kernel arguments again, through the kernarg segment pointer made opaque at the point of use: a field read
// through late_args() is loaded (s_load, scalar cache) where it is consumed instead of at kernel entry.  The pair
// loop needs ~60 SGPRs for its polynomial coefficients and math constants; with every sfw_launch field the
// per-agent pass and the epilogue use ALSO held in SGPRs across it, the allocator spills coefficients to VGPR lanes
// and reloads them with v_readlane inside the loop (14 VALU issues per pair in the first version of this layout).
typedef const __attribute__((address_space(4))) sfw_launch *late_launch;
__device__ __forceinline__ late_launch late_args() {
#ifdef SFW_DBG_NO_LAUNDER
  return (late_launch)__builtin_amdgcn_kernarg_segment_ptr();
#endif
  const __attribute__((address_space(4))) void *p =
      (const __attribute__((address_space(4))) void *)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return static_cast<late_launch>(p);  // sfw_launch is the first kernel argument: offset 0 of the segment
}

// Measurement aid (sfw_set_timing): the middle wave of a K2 launch (wave 0 is the never-scored (0,0) sample) records the shader-clock counter (s_memtime) and the
// constant-rate counter (s_memrealtime) when it starts and when it ends; the host turns the two differences into the
// clock the kernel sustained (boxes differ by ~10 %: a kernel time means little without it).  Written straight to
// memory: nothing is held in registers across the rollout.
__device__ __forceinline__ void clock_probe(int which) {
  if (blockIdx.x != gridDim.x / 2 || threadIdx.x != 0) return;
  unsigned long long *const p = late_args()->clock_probe;
  if (!p) return;
  p[2 * which] = __builtin_readcyclecounter();
  p[2 * which + 1] = wall_clock64();
}

// Social-force constants of the PAIR term in the force type (host-derived, see sfw_derived).
template <typename R> struct sfm_consts {
  sfwm::poly_consts pc;
  R lambda, neg_l2e_inv_gamma, l2_f_social, c_vel, c_ang;
};
// Constants of the per-agent pass (desired / obstacle / group forces, integration, contact test): read late.
struct agent_consts {
  double f_desired, inv_tau, dt, rr, inv_O, l2_f_obstacle, l2e_inv_sigma;
  double f_gaze, f_coherence, f_repulsion;
  const double *obstacles;  // (read with the rest: a lone wave waits for every scalar load it issues on its own)
  int O, robot_id, obs_tasks;
};
__device__ __forceinline__ agent_consts load_agent_consts(late_launch La, bool f32) {
  agent_consts c;
  c.f_desired = La->k.f_desired;
  c.inv_tau = La->k.inv_tau;
  c.dt = La->dt;
  c.rr = La->k.rr;
  c.inv_O = La->k.inv_O;
  c.l2_f_obstacle = f32 ? static_cast<double>(La->k.f.l2_f_obstacle) : La->k.d.l2_f_obstacle;
  c.l2e_inv_sigma = f32 ? static_cast<double>(La->k.f.l2e_inv_sigma) : La->k.d.l2e_inv_sigma;
  c.f_gaze = La->p.sfm_force_factor_group_gaze;
  c.f_coherence = La->p.sfm_force_factor_group_coherence;
  c.f_repulsion = La->p.sfm_force_factor_group_repulsion;
  c.O = La->O;
  c.obstacles = La->obstacles;
  c.obs_tasks = La->k.obs_tasks;
  c.robot_id = La->agent_c[0].id;
  return c;
}

// Force exerted ON agent i BY agent j (one term of lightsfm's
// computeSocialForce), from diff = pj - pi and
// w = vi - vj.  Antisymmetric under i<->j when every agent carries the same
// sfm::Parameters (the reference never overrides them), so the caller applies -f
// to j and evaluates each unordered pair once.
//   dhat = diff/|diff|, I = lambda*w + dhat,
//   theta = angle(dhat) - angle(I) = atan2(I x dhat, I . dhat)   in (-pi, pi]
//   sign(theta) = sign(w x diff)   (I x dhat = lambda * w x dhat exactly)
//   B = gamma*|I|
//   f = Fs * ( -exp(-|diff|/B - (n' B theta)^2) * Ihat
//              - sign(theta) * exp(-|diff|/B - (n B theta)^2) * leftNormal(Ihat) )
// cw = w x diff evaluated in double by the caller (exact sign in both modes).
__device__ __forceinline__ double copysign_from(double mag, double sgn) { return __builtin_copysign(mag, sgn); }
__device__ __forceinline__ float copysign_from(float mag, double sgn) {
  return __builtin_copysignf(mag, static_cast<float>(sgn));  // the conversion keeps the sign bit, of zeros and of underflows too
}

// NORM_ONLY: fx receives |f| = sqrt(ev^2 + ea^2) (Ihat and its left normal are orthonormal) and fy nothing — what
// computeSocialWork's robot-on-person term needs (ref :692-699): no direction, no sign.
template <typename R, bool NORM_ONLY = false>
__device__ __forceinline__ void pair_force(const sfm_consts<R> &k, R dx, R dy, R wx, R wy, double cw, R &fx,
                                           R &fy) {
  using namespace sfwm;
  const R tiny = tiny_of<R>::v;
  R rd, dn, rl, il;
  // + tiny instead of a clamp: far below an ulp of any d2 that matters, and coincident agents (d2 = 0) still
  // get dhat -> 0 instead of a NaN
  const R d2 = fma(dx, dx, fma(dy, dy, tiny));
  rsqrt_sqrt(d2, rd, dn);
  const R ux = dx * rd, uy = dy * rd;
  const R ix = fma(k.lambda, wx, ux), iy = fma(k.lambda, wy, uy);
  const R l2 = fma(ix, ix, fma(iy, iy, tiny));
  rsqrt_sqrt(l2, rl, il);
  const R sn = fma(ix, uy, -(iy * ux));    // |I| sin(theta)
  const R ncs = fma(-ix, ux, -(iy * uy));  // -|I| cos(theta)
  const R theta = angle_abs(k.pc, fabs(sn), ncs, rl, il);  // |(sn,cs)| = |I| since dhat is unit
  // log2 Fs - log2(e) |diff| / B: the force factor rides in the exponent (Fs exp(x) = 2^(x log2 e + log2 Fs)), clamped so
  // that exp2_fast's integer exponent stays in range for any input (2^-1100 is 0 in double and in float; the clamp never
  // changes a result)
  const R a = fmax(fma(dn * rl, k.neg_l2e_inv_gamma, k.l2_f_social), R(-1100));
  const R t2 = l2 * (theta * theta);      // (B theta)^2 = gamma^2 |I|^2 theta^2; gamma^2 sits in c_vel / c_ang
  R ev, ea;  // Fs exp(-|diff|/B - (n' B theta)^2), Fs exp(-|diff|/B - (n B theta)^2)
#if SFW_SIGN_OF_ZERO
  exp2_fast2(k.pc, fma(k.c_vel, t2, a), fma(k.c_ang, t2, a), ev, ea);
#else
  // sign(theta) = sign(w x diff) is 0 for a pair whose w x diff is exactly 0 (relative rest, motion exactly along the
  // connecting line — which PERSISTS over the steps for a robot driving straight at a person on its axis): the angular
  // term is then exactly 0, as lightsfm's is for theta == 0.  The zero enters through the exponent of the term's 2^k
  // scaling (a compare and ONE select on an integer; as a select on the f64 result it was a compare and two).
  exp2_fast2_gated(k.pc, fma(k.c_vel, t2, a), fma(k.c_ang, t2, a), cw, ev, ea);
#endif
  if constexpr (NORM_ONLY) {
    const R q = fma(ev, ev, ea * ea);
    R rq, nq;
    rsqrt_sqrt(fma(R(1), q, tiny), rq, nq);
    fx = q * rq;  // exactly 0 for q = 0
    fy = R(0);
    return;
  }
  // sign(theta) * exp(...): the sign BIT of cw, one v_bfi_b32 (ea is exactly 0 when cw is: its sign is then immaterial).
  // SFW_SIGN_OF_ZERO=1 (round 3's default, kept for A/B) skips the gate above: the sign of a zero then decides, and the host
  // takes that term back out for the pairs of the handed-over state (rest_forces, sfw_capi.hip) — but not for an alignment
  // that persists past it.
  ea = copysign_from(ea, cw);
  const R gx = ix * rl, gy = iy * rl;    // Ihat
  // f = -ev * Ihat - ea * leftNormal(Ihat),  leftNormal(x,y) = (-y, x)
  fx = fma(ea, gy, -(ev * gx));
  fy = fma(-ev, gy, -(ea * gx));
}
template <typename R, bool NORM_ONLY = false>
__device__ __forceinline__ void pair_force_state(const sfm_consts<R> &k, double pix, double piy, double vix,
                                                 double viy, double pjx, double pjy, double vjx, double vjy,
                                                 R &fx, R &fy) {
  const double dx = pjx - pix, dy = pjy - piy, wx = vix - vjx, wy = viy - vjy;
  const double cw = fma(wx, dy, -(wy * dx));
  pair_force<R, NORM_ONLY>(k, R(dx), R(dy), R(wx), R(wy), cw, fx, fy);
}

// |(x, y)|, exactly 0 for the zero vector (the clamp only keeps the reciprocal root finite)
__device__ __forceinline__ double fast_norm(double x, double y) {
  const double q = fma(x, x, y * y);
  double rs, sq;
  sfwm::rsqrt_sqrt(fmax(q, 1e-300), rs, sq);
  return q * rs;
}

// desiredForce of one person (lightsfm computeDesiredForce), double.
__device__ __forceinline__ void desired_force(const agent_consts &k, double px, double py, double vx, double vy,
                                              bool has_goal, double gx, double gy, double gr, double dv, double &fx,
                                              double &fy) {
  const double ex = gx - px, ey = gy - py;
  double inv, en;
  sfwm::rsqrt_sqrt(fmax(fma(ex, ex, ey * ey), 1e-300), inv, en);
  if (has_goal && en > gr) {
    fx = k.f_desired * (ex * inv * dv - vx) * k.inv_tau;
    fy = k.f_desired * (ey * inv * dv - vy) * k.inv_tau;
  } else {
    fx = -vx * k.inv_tau;
    fy = -vy * k.inv_tau;
  }
}

// obstacleForce of one agent: mean over the shared laser points (lightsfm computeObstacleForce; SURVEY.md
// Appendix A): (1/O) sum_o k exp(-(|p - o| - radius)/sigma) (p - o)/|p - o|.  Evaluated as
//   [k exp(radius/sigma) / O] * sum_o 2^(-|p - o| log2(e)/sigma) / |p - o| * (p - o):
// the agent's constant factor multiplies the finished sum (obstacle_scale), so the exponent inside the loop is a plain
// product with the distance and is formed inside the two fma of the exponential's range reduction (exp2_scaled).
// 24 VALU instructions per (agent, point), one of them the four-slot v_rsq_f64: 27 issue slots (round 3: 29).
//
// Summation order (the same in both kernel organisations, so that they stay bit-identical): the points are cut
// into OBS_SEG = 16 consecutive segments of L = ceil(O / 16) points; a segment's terms are added in point order
// starting from 0, the segment sums are added in segment order.  (Round 5 measured a two-level order — groups of four segments,
// then the groups: a chain of seven dependent additions instead of sixteen — and took it back out: the lanes' reduction gained
// ~80 cycles per phase, the register form's and the lane-per-agent pass's nested segment loops lost 14-22 us per 40-step control
// cycle with 50 people, profiles/r05_cycle_k2.txt.)
constexpr int OBS_SEG = 16;
constexpr int OBS_AGENT_LANES = WAVE / OBS_SEG;  // flat form: 4 lanes per segment, each with its own agents
#ifndef SFW_OBS_KA
#define SFW_OBS_KA 4
#endif
constexpr int OBS_AGENTS_PER_LANE = SFW_OBS_KA;  // ... up to four per round of tasks: 16 agents x 16 segments per round
// The points a WAVE-UNIFORM loop reads come straight from global memory through the scalar cache (obs_global: the array
// as a constant-address-space pointer, so that the loads are s_load and cost neither LDS space nor VALU/VMEM issue);
// the flat form's task loop, whose lanes walk 16 different segments, reads them with per-lane vector loads (the scan is a
// few KB: L1 hits).  Staged in LDS (16 B per point and wave, round 2), a 720-point laser scan took the flat form of the
// target crowd from five waves per SIMD to two and the register form of cfg2 from six to three.
typedef const __attribute__((address_space(4))) double *obs_global_ptr;
__device__ __forceinline__ obs_global_ptr obs_global(const double *obstacles) {
  return (obs_global_ptr)(const __attribute__((address_space(1))) double *)obstacles;
}
__device__ __forceinline__ double2 obs_point(const double2 *obs, int o) { return obs[o]; }
__device__ __forceinline__ double2 obs_point(obs_global_ptr obs, int o) { return double2{obs[2 * o], obs[2 * o + 1]}; }
// The hand-pipelined scalar loads of obstacle_sums (obs_group_issue / _wait: the next four points' s_load in flight while four
// are evaluated, across the segment boundaries) — what VERDICT r3 asked for — measured and OFF: their scalar bookkeeping per
// group of four costs more than the latency they hide, with several waves per SIMD (register form, cfg2 + 240 points: K2 3.33
// -> 3.56 ms) and for a lone wave alike (flat form's lane-per-agent pass, control cycle with 50 people and 60 points: K2 474
// -> 499 us, 16 points: 385 -> 409).  -DSFW_FLAT_PIPELINED=true builds them into the flat form's lane-per-agent pass.
#ifndef SFW_FLAT_PIPELINED
#define SFW_FLAT_PIPELINED false
#endif
#ifndef SFW_OBS_UNIFORM
#define SFW_OBS_UNIFORM 1  // the task loop of a GPU-filling launch with wave-uniform trip counts (obstacle_segment_multi_uniform)
#endif
#ifndef SFW_OBS_UNIFORM_LDS
#define SFW_OBS_UNIFORM_LDS 1  // ... and of an under-filled one (LDS copy of the points)
#endif
#ifndef SFW_SKIP_EMPTY_K2
#define SFW_SKIP_EMPTY_K2 1  // a robot alone without laser points: sfw_no_social_kernel instead of K2
#endif
#ifndef SFW_OBS_UNROLL_SCALAR
#define SFW_OBS_UNROLL_SCALAR 4  // points per s_load group of a wave-uniform loop (4: one s_load_dwordx16)
#endif
// one (agent, point) term: a += exp(-|p - q| / sigma) / |p - q| * (p - q)
template <typename R>
__device__ __forceinline__ void obstacle_term(const sfm_consts<R> &k, double2 q, double px, double py, R nis, R &ax, R &ay) {
  using namespace sfwm;
  const R mx = R(px - q.x), my = R(py - q.y);
  R rm, mn;
  rsqrt_sqrt(fma(mx, mx, fma(my, my, tiny_of<R>::v)), rm, mn);
  const R e = exp2_scaled(k.pc, mn, nis) * rm;
  ax = fma(e, mx, ax);
  ay = fma(e, my, ay);
}
// nis: -log2(e) / sigma in a VGPR (vgpr_const; the range reduction's first fma also reads the shift constant, and a VOP3
// instruction takes one scalar operand)
template <typename R, typename ObsPtr>
__device__ __forceinline__ void obstacle_segment(const sfm_consts<R> &k, ObsPtr obs, int o_begin, int o_end,
                                                 double px, double py, R nis, R &ax, R &ay) {
  ax = R(0);
  ay = R(0);
#pragma unroll SFW_OBS_UNROLL_SCALAR
  for (int o = o_begin; o < o_end; ++o) obstacle_term<R>(k, obs_point(obs, o), px, py, nis, ax, ay);
}
// The same segment for NJ agents at once (the flat form's task loop): a point is loaded ONCE per lane — a per-lane vector
// load, the lanes of a wave walk 16 different segments — and meets the lane's NJ agents.  With one agent per lane the
// vector memory pipe, not the VALU, set the pace of a 720-point scan (one 16-byte load per 27 issue slots and lane).
// Every (agent, segment) sum is formed in point order as in obstacle_segment: bit-identical.
// This version walks the lane's own bounds [o_begin, o_end) — a divergent loop; the kernels run obstacle_segment_multi_uniform
// below (SFW_OBS_UNIFORM / SFW_OBS_UNIFORM_LDS = 0 build this one back in: profiles/r04_ab_uniform*.txt).
typedef const __attribute__((address_space(3))) double *obs_lds_ptr;
__device__ __forceinline__ double2 obs_point(obs_lds_ptr obs, int o) { return double2{obs[2 * o], obs[2 * o + 1]}; }
template <typename R, int NJ, typename ObsPtr>
__device__ __forceinline__ void obstacle_segment_multi(const sfm_consts<R> &k, ObsPtr obs, int o_begin, int o_end,
                                                       const double *px, const double *py, R neg_l2e_inv_sigma, R *ax, R *ay) {
#pragma unroll
  for (int j = 0; j < NJ; ++j) ax[j] = ay[j] = R(0);
  const R nis = neg_l2e_inv_sigma;
  auto terms = [&](const double2 q) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) obstacle_term<R>(k, q, px[j], py[j], nis, ax[j], ay[j]);
  };
  // One point ahead, two register sets in turn: the load of the next point is in flight while a point meets the lane's
  // agents (a lone wave — a control cycle's — otherwise sits out a round trip per point).  The loads run up to two points
  // past the segment — into the next segment, or into the padding behind the last point (64 bytes in global memory,
  // sfw_set_agents; two points in the LDS copy, lds_layout): loaded, never evaluated, and no index to clamp.
  if (o_begin < o_end) {
    double2 qa = obs_point(obs, o_begin);
    if constexpr (NJ <= 2) {
      // one or two agents per lane (small crowds: a control cycle's lone waves): two points per iteration in ONE basic block,
      // so that the 2 NJ independent chains are interleaved — a lone wave issues a dependent instruction every ~8.5 cycles,
      // an independent one every ~5 (control cycle with 5 people and 240 points: K2 218 -> 209 us; a robot alone: 123 -> 111)
      int o = o_begin;
#pragma unroll 1
      for (; o + 2 <= o_end; o += 2) {
        const double2 qb = obs_point(obs, o + 1);
        const double2 qn = obs_point(obs, o + 2);
        terms(qa);
        terms(qb);
        qa = qn;
      }
      if (o < o_end) terms(qa);
    } else {
#pragma unroll 1
      for (int o = o_begin; o < o_end; o += 2) {
        const double2 qb = obs_point(obs, o + 1);
        terms(qa);
        qa = obs_point(obs, o + 2);
        if (o + 1 < o_end) terms(qb);
      }
    }
  }
}
// The task loop of a GPU-FILLING launch (points in global memory): the same sums with wave-uniform trip counts.  A lane's
// segment holds Lseg points, or the `partial` rest of the scan, or none: written with the lane's own bounds (above) the loop
// is divergent — index, compare, address and the copies between the two register sets cost 8 vector instructions per point
// next to the 4 x 24 of the terms (PMC: 26.6 per evaluation).  Here the wave runs `partial` iterations with every lane that
// has points and Lseg - partial more with the lanes of the full segments: scalar counter, scalar base address + the lane's
// fixed byte offset (global_load saddr form), loads written as asm one point ahead into two register sets that the
// unrolled body alternates without copying — the waits are explicit, the compiler does not count an asm load.  Reads up to
// two points past the range like the loop above.
typedef double obs_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ obs_d2 obs_load_ahead(const double2 *base, uint32_t lane_bytes) {
  obs_d2 q;
  // (s_nop 4: the base may have just come back from an SGPR spill lane — v_readlane — and a VMEM instruction reading such an
  // SGPR needs 5 wait states, which the hazard recogniser does not insert into an asm block: load_pair_entries)
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(q) : "v"(lane_bytes), "s"(base) : "memory");
  return q;
}
// the oldest of the point loads has arrived (YOUNGER: point loads issued after it; loads return in order, and whatever
// else is in flight — the next step's robot record — is older than both)
template <int YOUNGER> __device__ __forceinline__ void obs_load_wait(obs_d2 &q) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(q) : "n"(YOUNGER));
}
// (LDS copy of the points — an under-filled launch —: the same two loops with the compiler's own loads)
template <typename R, int NJ>
__device__ __forceinline__ void obstacle_segment_multi_uniform(const sfm_consts<R> &k, obs_lds_ptr obs, int O, int Lseg, int seg,
                                                               const double *px, const double *py, R neg_l2e_inv_sigma, R *ax,
                                                               R *ay) {
#pragma unroll
  for (int j = 0; j < NJ; ++j) ax[j] = ay[j] = R(0);
  const R nis = neg_l2e_inv_sigma;
  auto terms = [&](const double2 q) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) obstacle_term<R>(k, q, px[j], py[j], nis, ax[j], ay[j]);
  };
  const int n_full = O / Lseg, partial = O - n_full * Lseg;
  const obs_lds_ptr mine = obs + 2 * (seg * Lseg);
  auto run = [&](obs_lds_ptr base, int n) {
    double2 qa = obs_point(base, 0);
    int i = 0;
#pragma unroll 1
    for (; i + 2 <= n; i += 2) {
      const double2 qb = obs_point(base, i + 1);
      const double2 qn = obs_point(base, i + 2);
      terms(qa);
      terms(qb);
      qa = qn;
    }
    if (i < n) terms(qa);
  };
  const int first = partial > 0 ? partial : Lseg;
  if (seg < n_full + (partial > 0 ? 1 : 0)) run(mine, first);
  if (first < Lseg && seg < n_full) run(mine + 2 * first, Lseg - first);
}
template <typename R, int NJ>
__device__ __forceinline__ void obstacle_segment_multi_uniform(const sfm_consts<R> &k, const double2 *obs, int O, int Lseg, int seg,
                                                               const double *px, const double *py, R neg_l2e_inv_sigma, R *ax,
                                                               R *ay) {
#pragma unroll
  for (int j = 0; j < NJ; ++j) ax[j] = ay[j] = R(0);
  const R nis = neg_l2e_inv_sigma;
  auto terms = [&](const obs_d2 q) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) obstacle_term<R>(k, double2{q.x, q.y}, px[j], py[j], nis, ax[j], ay[j]);
  };
  const int n_full = O / Lseg, partial = O - n_full * Lseg;  // wave-uniform: full segments, points of the one behind them
  const uint32_t lane_bytes = static_cast<uint32_t>(seg * Lseg) * 16u;
  auto run = [&](const double2 *base, int n) {  // n > 0 points from base[lane's first point] on, every active lane alike
    obs_d2 qa = obs_load_ahead(base, lane_bytes);
    int i = 0;
#pragma unroll 1
    for (; i + 2 <= n; i += 2) {
      obs_d2 qb = obs_load_ahead(base + i + 1, lane_bytes);
      obs_load_wait<1>(qa);
      terms(qa);
      qa = obs_load_ahead(base + i + 2, lane_bytes);
      obs_load_wait<1>(qb);
      terms(qb);
    }
    obs_load_wait<0>(qa);  // (also when it is not evaluated: no load of this loop is left in flight)
    if (i < n) terms(qa);
  };
  const int first = partial > 0 ? partial : Lseg;  // points every non-empty segment has
  if (seg < n_full + (partial > 0 ? 1 : 0)) run(obs, first);
  if (first < Lseg && seg < n_full) run(obs + first, Lseg - first);
}
// what an agent's sum over the points is multiplied with: k exp(radius / sigma) / O
template <typename R>
__device__ __forceinline__ double obstacle_scale(const sfm_consts<R> &k, const agent_consts &c, double radius) {
  return static_cast<double>(sfwm::exp2_fast(k.pc, static_cast<R>(fma(radius, c.l2e_inv_sigma, c.l2_f_obstacle)))) * c.inv_O;
}
// Four points = one s_load_dwordx16 into 16 SGPRs, issued and awaited by hand: the loads of a wave-uniform pass are
// software-pipelined ACROSS the segments — while a group of four points is evaluated (4 x 27 issue slots) the next group's
// load is in flight, also when that group opens the next segment.  Left to the compiler every s_load was followed at once
// by s_waitcnt lgkmcnt(0) (scalar loads return out of order: any use needs the counter at zero), so a lone wave — a control
// cycle, a coarse shared-prefix level — sat out the scalar-cache latency once per four points, sixteen times per agent and
// step for a 64-point scan.  Two register sets alternate; a set is only read after the wait that follows its load (the asm
// operands say so: the wait "modifies" the set).
typedef double obs_group __attribute__((ext_vector_type(8)));  // x0 y0 x1 y1 x2 y2 x3 y3
__device__ __forceinline__ void obs_group_issue(obs_group &g, obs_global_ptr p) {
  asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(g) : "s"(p));
}
__device__ __forceinline__ void obs_group_wait(obs_group &g) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(g)); }

// all sixteen segments on one lane: the agent's raw sums (tx, ty) and the factor they are to be multiplied with.  The
// callers form the force as ONE explicit fma per component, acc = fma(t, scale, acc), in both organisations: written as a
// product followed by +=, whether the two are contracted into an fma is the compiler's choice per call site, and the
// organisations stopped being bit-identical the day it chose differently.
// The points array is readable 48 bytes past its last point (sfw_set_agents pads it): the last group of a segment is
// loaded whole and evaluated up to the segment's end.
// PIPELINED = false (every caller's default, see SFW_FLAT_PIPELINED): the compiler's loop (s_load, wait, four interleaved
// terms).  Same sums in the same order either way.
template <typename R, bool PIPELINED>
__device__ __forceinline__ void obstacle_sums(const sfm_consts<R> &k, const agent_consts &c, obs_global_ptr obs, double px,
                                              double py, double radius, double &tx_out, double &ty_out, double &scale) {
  const int O = c.O, L = (O + OBS_SEG - 1) / OBS_SEG;
  const R nis = sfwm::vgpr_const(static_cast<R>(-c.l2e_inv_sigma));
  R tx = R(0), ty = R(0);  // the segment sums are added to +0 in segment order (the flat form's reduction does the same)
  R ax = R(0), ay = R(0);
  if constexpr (!PIPELINED) {
    for (int b = 0; b < O; b += L) {  // ceil(O / L) <= 16 segments
      obstacle_segment<R, obs_global_ptr>(k, obs, b, min(b + L, O), px, py, nis, ax, ay);
      tx += ax;
      ty += ay;
    }
    tx_out = static_cast<double>(tx);
    ty_out = static_cast<double>(ty);
    scale = obstacle_scale<R>(k, c, radius);
    return;
  }
  // one group: `cnt` points of set g, then — if the segment ends with it — the segment's sum joins the total
  auto eval = [&](const obs_group &g, int cnt, bool seg_done) {
    if (cnt == 4) {
      obstacle_term<R>(k, double2{g.s0, g.s1}, px, py, nis, ax, ay);
      obstacle_term<R>(k, double2{g.s2, g.s3}, px, py, nis, ax, ay);
      obstacle_term<R>(k, double2{g.s4, g.s5}, px, py, nis, ax, ay);
      obstacle_term<R>(k, double2{g.s6, g.s7}, px, py, nis, ax, ay);
    } else {
      asm volatile("" ::: "memory");  // not the same code as above: the compiler must not merge the first term of the two paths
      obstacle_term<R>(k, double2{g.s0, g.s1}, px, py, nis, ax, ay);  // (the four terms of a full group are to be interleaved)
      if (cnt > 1) obstacle_term<R>(k, double2{g.s2, g.s3}, px, py, nis, ax, ay);
      if (cnt > 2) obstacle_term<R>(k, double2{g.s4, g.s5}, px, py, nis, ax, ay);
    }
    if (seg_done) {
      tx += ax;
      ty += ay;
      ax = R(0);
      ay = R(0);
    }
  };
  // the group after the one at o of the segment ending at e: (o, e) advance, false when there is none
  auto advance = [&](int &o, int &e) {
    o += 4;
    if (o >= e) {  // next segment
      o = e;
      e = min(e + L, O);
    }
    return o < O;
  };
  if (O > 0) {
    int o = 0, e = min(L, O);
    obs_group ga, gb;
    obs_group_issue(ga, obs);
    for (;;) {
      int o2 = o, e2 = e;
      const bool more = advance(o2, e2);
      obs_group_wait(ga);
      if (more) obs_group_issue(gb, obs + 2 * o2);
      eval(ga, min(4, e - o), o + 4 >= e);
      if (!more) break;
      o = o2;
      e = e2;
      const bool more2 = advance(o2, e2);
      obs_group_wait(gb);
      if (more2) obs_group_issue(ga, obs + 2 * o2);
      eval(gb, min(4, e - o), o + 4 >= e);
      if (!more2) break;
      o = o2;
      e = e2;
    }
  }
  tx_out = static_cast<double>(tx);
  ty_out = static_cast<double>(ty);
  scale = obstacle_scale<R>(k, c, radius);
}

// The value of the neighbouring lane (lane ^ 1), as a DPP quad permutation: two vector instructions where __shfl_xor goes
// through the LDS crossbar (ds_bpermute_b32 x 2: ~130 cycles on the chain of a wave with the GPU to itself)
__device__ __forceinline__ double swap_pair(double v) {
  const uint64_t u = __builtin_bit_cast(uint64_t, v);
  constexpr int QP_1032 = 0xB1;  // quad_perm:[1,0,3,2]
  const uint32_t lo = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(static_cast<uint32_t>(u)), QP_1032, 0xf, 0xf, true));
  const uint32_t hi = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(static_cast<uint32_t>(u >> 32)), QP_1032, 0xf, 0xf, true));
  return __builtin_bit_cast(double, (static_cast<uint64_t>(hi) << 32) | lo);
}

// LDS map of one wave.  Agent state lives in PLANES of `cap` doubles each — px, py, vx, vy, the force
// accumulators fjx, fjy (received "as j") and, in the flat kernel, fcx, fcy (received "as i") — `cap` doubles
// apart, so (1) a wave's ds_read_b64 / ds_add_f64 of consecutive agents hit consecutive 8-byte words (all 32
// banks; with 16-byte (x, y) records every 8-byte access used half the banks twice: 30 % of the LDS-active cycles
// were bank conflicts, profiles/r02a) and (2) with a compile-time cap the plane distances fold into the DS
// instructions' offset fields: one address VGPR per agent reaches all eight.  The per-agent launch constants are
// staged only when `consts` is set (the flat kernel without groups reads them from global memory instead: 40 B/agent
// less LDS, which is what bounds its occupancy for large crowds).  Built on the host with base = nullptr to size
// the allocation.
struct lds_layout {
  double *px, *py, *vx, *vy, *fjx, *fjy, *fcx, *fcy;
  double2 *gcen;
  double2 *obs;         // flat form, launches that leave the GPU under-filled (sfw_launch.k.obs_lds): the wave's copy of the laser
                        // points.  A lone wave per SIMD waits out every load it issues, and LDS answers in a quarter of the time
                        // of the vector L1; in a GPU-filling launch the copy (16 B per point and wave) would cost occupancy instead
  sfw_robot_step *rsb;  // robot records: one per sample of the wave (register form), two (flat form: this step's
                        // and the prefetched next step's)
  sfw_agent_const *ac;  // the per-agent launch constants as they sit in global memory (48 B records: one address
                        // register per agent reaches every field through the DS offset field)
  double *swp;
  double2 *opart;       // flat form with laser points: the 64 lanes' segment sums of two of their agents ([2][64]; of all four
                        // — [4][64] — in the launches that leave the GPU under-filled, obs_in_lds: LDS is plentiful there)
  double *wr;           // flat form with laser points: the robot's social-work term waiting for its obstacle part [0] and
                        // the two components of that part [1], [2]
  double *oscale;       // flat form with laser points: obstacle_scale of every agent, formed once per launch (the reduction
                        // of a round of tasks would otherwise wait for the agent's radius from global memory every time)
  int *hasgoal, *dead, *grp, *goff, *gmem;  // hasgoal: register form (one int per 8-byte cell)
  unsigned char *hasgoal8;  // flat form: one byte per agent (crowds of 129..255 agents are LDS-bound in occupancy)
  int hg_stride;        // ints between two slots' hasgoal words (see hg())
  size_t bytes;
  // Register form (with_frc = false): everything a slot reads and writes every step sits at a COMPILE-TIME distance from
  // its px word — the six state planes, then a plane of 8-byte cells {hasgoal, Wp switch} and the social-work plane, then
  // the samples' contact flags and robot records — so one address register (8 * slot) and one (4 * sample) serve the
  // whole per-agent pass; only the constants (ac) and the arrays of the laser-point / group passes have run-time bases.
  // Round 2's layout had eight run-time array bases, i.e. eight per-lane address registers held across the rollout, four
  // of which the allocator spilled to scratch and reloaded one after the other in every step.
  static constexpr int REG_DEAD_CAP = 32;  // samples per register-form wave (plan_for)
  __host__ __device__ lds_layout(char *base, int A, int cap, int GA, int G, int O, int NG, int NM, bool consts,
                                 bool with_frc, bool obs_in_lds = false) {
    char *const base0 = base;
    auto take = [&](size_t n) {
      char *p = base;
      base += (n + 15) & ~size_t(15);
      return p;
    };
    const size_t plane = sizeof(double) * static_cast<size_t>(cap);  // cap is even: planes stay 16-byte aligned
    px = reinterpret_cast<double *>(take(plane));
    py = reinterpret_cast<double *>(take(plane));
    vx = reinterpret_cast<double *>(take(plane));
    vy = reinterpret_cast<double *>(take(plane));
    fjx = reinterpret_cast<double *>(take(plane));
    fjy = reinterpret_cast<double *>(take(plane));
    if (with_frc) {  // flat kernel
      fcx = reinterpret_cast<double *>(take(plane));
      fcy = reinterpret_cast<double *>(take(plane));
      rsb = reinterpret_cast<sfw_robot_step *>(take(sizeof(sfw_robot_step) * 2));
      swp = reinterpret_cast<double *>(take(sizeof(double) * GA));
      opart = reinterpret_cast<double2 *>(take(sizeof(double2) * (O > 0 ? (obs_in_lds ? 4 : 2) * 64 : 0)));
      wr = reinterpret_cast<double *>(take(sizeof(double) * (O > 0 ? 4 : 0)));
      oscale = reinterpret_cast<double *>(take(sizeof(double) * (O > 0 ? A : 0)));
      obs = reinterpret_cast<double2 *>(take(sizeof(double2) * (obs_in_lds ? O + 2 : 0)));  // + 2: read ahead, never evaluated
      if (!obs_in_lds) obs = nullptr;
      hasgoal8 = reinterpret_cast<unsigned char *>(take(static_cast<size_t>(GA)));
      hasgoal = nullptr;
      hg_stride = 1;
      dead = reinterpret_cast<int *>(take(sizeof(int) * G));
    } else {         // register form: fixed distances first (reg_off below restates them)
      fcx = fcy = nullptr;
      opart = nullptr;
      wr = nullptr;
      oscale = nullptr;
      obs = nullptr;
      hasgoal = reinterpret_cast<int *>(take(plane));
      hasgoal8 = nullptr;
      hg_stride = 2;
      swp = reinterpret_cast<double *>(take(plane));
      dead = reinterpret_cast<int *>(take(sizeof(int) * REG_DEAD_CAP));
      rsb = reinterpret_cast<sfw_robot_step *>(take(sizeof(sfw_robot_step) * G));
    }
    ac = reinterpret_cast<sfw_agent_const *>(take(sizeof(sfw_agent_const) * (consts ? A : 0)));
    // group arrays only when an agent carries a group id (the GROUPS kernels): nothing reads them otherwise
    gcen = reinterpret_cast<double2 *>(take(sizeof(double2) * (NG > 0 ? G * NG : 0)));
    grp = reinterpret_cast<int *>(take(sizeof(int) * (NG > 0 ? A : 0)));
    goff = reinterpret_cast<int *>(take(sizeof(int) * (NG > 0 ? NG + 1 : 0)));
    gmem = reinterpret_cast<int *>(take(sizeof(int) * (NG > 0 && NM > 0 ? NM : 0)));
    bytes = static_cast<size_t>(base - base0);
  }
  __device__ __forceinline__ int hg(int sl) const { return hg_stride == 1 ? static_cast<int>(hasgoal8[sl]) : hasgoal[sl * hg_stride]; }
};
// Byte offsets of the register form's fixed part from a slot's px word (CAP = 64 * NS doubles per plane).
template <int CAP> struct reg_off {
  static constexpr int PY = 8 * CAP, VX = 16 * CAP, VY = 24 * CAP, FJX = 32 * CAP, FJY = 40 * CAP, HG = 48 * CAP, SW = 56 * CAP,
                       DEAD = 64 * CAP, RSB = 64 * CAP + 4 * lds_layout::REG_DEAD_CAP;
};

// Per-agent launch constants as agent_step consumes them.
struct agent_k {
  double gx, gy, gr, dv, rad;
  int id;
};
__device__ __forceinline__ agent_k agent_k_of(const sfw_agent_const &c) {
  return agent_k{c.goal_x, c.goal_y, c.goal_radius, c.desired_velocity, c.radius, c.id};
}
__device__ __forceinline__ agent_k agent_k_lds(const lds_layout &s, int i) { return agent_k_of(s.ac[i]); }
__device__ __forceinline__ agent_k agent_k_global(const sfw_agent_const *agent_c, int i) { return agent_k_of(agent_c[i]); }

// lightsfm computeGroupForce (non-_PAPER_VERSION_ branch)
// for agent i of sample g at position (px,py): gaze + coherence + repulsion.
// Group centres (sums of member positions) must already be in s.gcen.  Double in
// both precision modes, written like the pair term — no library calls:
//   * gaze fires when acos(dir . rel / (|dir||rel|)) > 90 deg, i.e. when dir . rel < 0 (no acos; the two
//     tests differ only for a cosine in (-1.6e-16, 0), where acos rounds to pi/2);
//   * (tanh(x) + 1) / 2 = 1 / (1 + exp(-2x));
//   * |d| < ra + rb is tested as |d|^2 < (ra + rb)^2.
template <typename R>
__device__ double2 group_force(const sfm_consts<R> &k, const agent_consts &c, const lds_layout &s, int NG, int g, int A,
                               int i, int sl, double px, double py) {
  const sfw_agent_const ci = s.ac[i];
  const int q = s.grp[i];
  if (q < 0) return double2{0.0, 0.0};
  const int m0 = s.goff[q], m1 = s.goff[q + 1];
  if (m1 - m0 < 2) return double2{0.0, 0.0};
  const double n = static_cast<double>(m1 - m0), inv_n = sfwm::rcp_nr(n);
  const double2 csum = s.gcen[g * NG + q];
  const double cx = csum.x * inv_n, cy = csum.y * inv_n;
  // desired direction at this state (zero when the agent has no goal to walk to)
  const double ex = ci.goal_x - px, ey = ci.goal_y - py;
  double inv_en, en;
  sfwm::rsqrt_sqrt(fmax(fma(ex, ex, ey * ey), 1e-300), inv_en, en);
  const bool has_dir = s.hg(sl) && en > ci.goal_radius;
  const double ddx = has_dir ? ex * inv_en : 0.0, ddy = has_dir ? ey * inv_en : 0.0;
  double fx = 0.0, fy = 0.0;
  {  // gaze: pulls along the desired direction when the rest of the group is behind
    const double w = sfwm::rcp_nr(n - 1.0);
    const double rx = fma(w, fma(n, cx, -px), -px), ry = fma(w, fma(n, cy, -py), -py);
    const double ep = fma(ddx, rx, ddy * ry);
    if (ep < 0.0) {  // has_dir is implied: ep == 0 without a direction
      const double sc = c.f_gaze * ep;  // (ep / |dir|^2) dir with |dir| = 1
      fx = sc * ddx;
      fy = sc * ddy;
    }
  }
  {  // coherence
    const double rx = cx - px, ry = cy - py;
    const double dist = fast_norm(rx, ry);
    const double x2 = fmin(2.0 * ((n - 1.0) * 0.5 - dist), 700.0);  // -2 (dist - maxd)
    const double soft = c.f_coherence * sfwm::rcp_nr(1.0 + sfwm::exp2_fast(k.pc, x2 * 1.4426950408889634074));
    fx = fma(rx, soft, fx);
    fy = fma(ry, soft, fy);
  }
  double rx = 0.0, ry = 0.0;  // repulsion between overlapping members
  const double ra = ci.radius;
  for (int m = m0; m < m1; ++m) {
    const int b = s.gmem[m];
    if (b == i) continue;
    const double dx = px - s.px[g * A + b], dy = py - s.py[g * A + b], rr = ra + s.ac[b].radius;
    if (fma(dx, dx, dy * dy) < rr * rr) { rx += dx; ry += dy; }
  }
  fx = fma(rx, c.f_repulsion, fx);
  fy = fma(ry, c.f_repulsion, fy);
  return double2{fx, fy};
}

template <typename R> __device__ __forceinline__ const sfw_force_k<R> &force_k(const sfw_launch &L);
template <> __device__ __forceinline__ const sfw_force_k<double> &force_k<double>(const sfw_launch &L) { return L.k.d; }
template <> __device__ __forceinline__ const sfw_force_k<float> &force_k<float>(const sfw_launch &L) { return L.k.f; }

template <typename R, bool PIN_ALL> __device__ __forceinline__ sfm_consts<R> make_consts(const sfw_launch &L) {
  const sfw_force_k<R> &f = force_k<R>(L);
  sfm_consts<R> k;
  // The pair loop reads ~30 distinct FP64 constants (19 polynomial coefficients, the range-reduction and fold
  // constants, these five): more SGPRs than the allocator has next to the loop's addresses and counters, so
  // these stay in vector registers (all five in the flat kernel, three in the register-resident one, which has to
  // stay within 80 VGPRs for six waves per SIMD).
  k.lambda = sfwm::vgpr_const(f.lambda);
  k.neg_l2e_inv_gamma = sfwm::vgpr_const(f.neg_l2e_inv_gamma);
  k.l2_f_social = PIN_ALL ? sfwm::vgpr_const(f.l2_f_social) : f.l2_f_social;
  k.c_vel = sfwm::vgpr_const(f.c_vel);
  k.c_ang = PIN_ALL ? sfwm::vgpr_const(f.c_ang) : f.c_ang;
  return k;
}

// The same from the late-read kernel arguments (see late_args): for a kernel that builds the constants once per step.
template <typename R, bool PIN_ALL> __device__ __forceinline__ sfm_consts<R> make_consts(late_launch La) {
  constexpr bool F32 = sizeof(R) == 4;
  sfm_consts<R> k;
  const R lambda = F32 ? R(La->k.f.lambda) : R(La->k.d.lambda), nig = F32 ? R(La->k.f.neg_l2e_inv_gamma) : R(La->k.d.neg_l2e_inv_gamma);
  const R lfs = F32 ? R(La->k.f.l2_f_social) : R(La->k.d.l2_f_social), cv = F32 ? R(La->k.f.c_vel) : R(La->k.d.c_vel);
  const R ca = F32 ? R(La->k.f.c_ang) : R(La->k.d.c_ang);
  k.lambda = sfwm::vgpr_const(lambda);
  k.neg_l2e_inv_gamma = sfwm::vgpr_const(nig);
  k.l2_f_social = PIN_ALL ? sfwm::vgpr_const(lfs) : lfs;
  k.c_vel = sfwm::vgpr_const(cv);
  k.c_ang = PIN_ALL ? sfwm::vgpr_const(ca) : ca;
  return k;
}

// A person that can never move next to a robot that stands still for the whole rollout (sfw_launch.pin_rest, built by
// sfw_capi.hip pinned_rest_table; null in every launch without such a person): exact relative rest at every step, where the
// pair loop's angular term is exactly 0 and lightsfm's the rounding noise of two atan2.  The host has evaluated the
// reference's term for the one geometry such a pair can have.  It enters in a pass of its own behind the per-agent pass —
// a wave-uniform branch no other launch takes, kept out of the loops whose register budgets are pinned
// (tests/test_kernel_resources.py) — at a step whose post-step robot record is the table's position at velocity 0:
//   * the robot's starting force of the NEXT step (its social force there is evaluated at this record) gets the lateral
//     parts pin[2..3]; the pair pass adds the rest;
//   * person i's social work gets pin[4 + i] = (the reference's Wp) - (the Wp the kernels have just counted: the same
//     pair with the angular part 0), which is 0 for a person that can move.
__device__ __forceinline__ bool pin_at_rest(const double *pin, const sfw_robot_step &rs) {
  return rs.vx == 0.0 && rs.vy == 0.0 && rs.x == pin[0] && rs.y == pin[1];
}

// One agent slot after the pair pass of a step: integrate the person (the robot's
// overwrite is the caller's), contact test, social-work terms, next step's
// desired+obstacle force.  F = total force on the agent at the pre-step state
// (for the robot: its social force only).  Returns this slot's social work.
// The laser-point term is the caller's: `work` lacks the robot's obstacle part and (nfx, nfy) a person's obstacle force.
template <typename R>
__device__ __forceinline__ double agent_step(const sfm_consts<R> &k, const agent_consts &c, const sfw_robot_step &rs,
                                             const agent_k &ak, bool robot, bool wp_on, int &hg, bool &contact,
                                             double &px, double &py, double &vx, double &vy, double Fx, double Fy,
                                             double &nfx, double &nfy) {
  double work = 0.0;
  contact = false;
  nfx = 0.0;
  nfy = 0.0;
  if (robot) {
    work = fast_norm(Fx, Fy);  // Wr, social part (ref :681-682): the robot's social force at the pre-step state
  } else {
    // lightsfm updatePosition, non-teleoperated branch
    vx = fma(Fx, c.dt, vx);
    vy = fma(Fy, c.dt, vy);
    double rsp, sp;
    sfwm::rsqrt_sqrt(fmax(fma(vx, vx, vy * vy), 1e-300), rsp, sp);
    if (sp > ak.dv) {
      const double sc = ak.dv * rsp;  // normalize() then *= desiredVelocity
      vx *= sc;
      vy *= sc;
    }
    px = fma(vx, c.dt, px);
    py = fma(vy, c.dt, py);
    if (hg) {
      const double ex = ak.gx - px, ey = ak.gy - py;
      if (fast_norm(ex, ey) <= ak.gr) hg = 0;  // goal reached: pop
    }
    // dynamic collision with the robot's post-step pose (ref :613-627)
    const double cx = rs.x - px, cy = rs.y - py;
    contact = cx * cx + cy * cy <= c.rr;  // the caller marks the sample rejected at this step
    // Wp (ref :692-699): force the post-step robot alone exerts on this person
    if (wp_on) {  // a person whose id is not the robot's (ref :692)
      R qn, unused;
      pair_force_state<R, true>(k, px, py, vx, vy, rs.x, rs.y, rs.vx, rs.vy, qn, unused);
      work = static_cast<double>(qn);
    }
    // desired force at the new state: with the obstacle term below, the next step's starting force
    desired_force(c, px, py, vx, vy, hg != 0, ak.gx, ak.gy, ak.gr, ak.dv, nfx, nfy);
  }
  // The robot's own state is overwritten with its post-step record (ref :600-604) by the caller,
  // after this function: here it would keep the record live across the obstacle loop.
  return work;
}

// Work items of a K2 launch: samples of the chunk (whole rollout, suffix phase) or classes of one
// level (prefix phases, see sfw_cls_agent).
__device__ __forceinline__ int64_t item_count(const sfw_launch &L) {
  return L.phase == SFW_PHASE_PREFIX ? static_cast<int64_t>(L.n_cls) : L.chunk_count;
}
// (row, column) coordinates of an item: class coordinates for a PREFIX item, grid coordinates of a sample
__device__ __forceinline__ void item_coords(const sfw_launch &L, int64_t item, int64_t &r, int64_t &c) {
  const int64_t width = L.phase == SFW_PHASE_PREFIX ? static_cast<int64_t>(L.n_col_cls) : static_cast<int64_t>(L.nw);
  r = item / width;  // chunks are whole rows when classes exist
  c = item - r * width;
}
// record of in_state an item resumes from
__device__ __forceinline__ int64_t source_class_of_item(const sfw_launch &L, int64_t item) {
  int64_t r, c;
  item_coords(L, item, r, c);
  return static_cast<int64_t>(L.row_src[r]) * L.n_col_src + L.col_src[c];
}
// chunk-local sample whose K1 robot-step records an item reads
__device__ __forceinline__ int64_t robot_sample_of_item(const sfw_launch &L, int64_t item) {
  if (L.phase != SFW_PHASE_PREFIX) return item;
  int64_t r, c;
  item_coords(L, item, r, c);
  return static_cast<int64_t>(L.row_rep[r]) * L.nw + L.col_rep[c];
}

// Stage the per-launch constants and the initial dead flags into LDS; returns false when no item of
// this wave is live (rejected by K1, or by a pedestrian contact inside the shared prefix).
template <bool GROUPS, bool CONSTS>
__device__ __forceinline__ void stage_consts(const sfw_launch &L, const lds_layout &s, int lane) {
  const int A = L.A;
  // lds_at() takes offsets into the wave's allocation for LDS addresses: true only while the K2 kernels have no
  // static __shared__ data in front of the dynamic allocation (s.px is the allocation's first byte)
  if (static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char *)(s.px))) != 0u) __builtin_trap();
  if constexpr (CONSTS) {
    for (int i = lane; i < A; i += WAVE) {
      s.ac[i] = L.agent_c[i];
    }
  }
  if (s.obs)
    for (int o = lane; o < L.O + 2; o += WAVE) s.obs[o] = o < L.O ? double2{L.obstacles[2 * o], L.obstacles[2 * o + 1]} : double2{0.0, 0.0};
  if constexpr (GROUPS) {
    for (int i = lane; i < A; i += WAVE) s.grp[i] = L.agent_grp[i];
    for (int q = lane; q <= L.NG; q += WAVE) s.goff[q] = L.grp_off[q];
    for (int m = lane; m < L.n_grp_mem; m += WAVE) s.gmem[m] = L.grp_mem[m];
  }
}
template <bool GROUPS, bool CONSTS>
__device__ __forceinline__ bool stage_wave(const sfw_launch &L, const lds_layout &s, int lane, int G, int Gn,
                                           int64_t first_local) {
  stage_consts<GROUPS, CONSTS>(L, s, lane);
  if (lane < G) {
    int dead = 1;
    if (lane < Gn) {
      if (L.phase == SFW_PHASE_PREFIX) {
        // a class is simulated whatever K1 said about its members, unless its parent already ended in a contact
        dead = L.resume ? L.in_dead[source_class_of_item(L, first_local + lane)] : 0;
      } else {
        const int st = L.status[L.chunk_begin + first_local + lane];
        // force_alive (Trajectory-point dumps): a sample the costmap rejected at pose a is integrated all the same, so
        // that a pedestrian contact at an earlier step b < a is found (the reference returns there, ref :613-627)
        dead = L.force_alive ? (st == SFW_ST_SKIPPED) : (st != SFW_ST_VALID);
        if (L.resume && dead == 0) dead = L.in_dead[source_class_of_item(L, first_local + lane)];
      }
    }
    s.dead[lane] = dead;
  }
  __syncthreads();
  bool any_live = false;
  for (int g = 0; g < G; ++g) any_live |= (s.dead[g] == 0);
  return any_live;
}

// Write the per-sample results: cost = base + w_s * social_work (ref :663-667).  The output pointers are read
// through late_args(): they are needed once, after the rollout.
__device__ __forceinline__ void finish_wave(const lds_layout &s, int lane, int G, int Gn, int64_t first_local,
                                            double sw_acc) {
  const late_launch La = late_args();
  const int A = La->A;
  const double social_weight = La->p.social_weight;
  double *const costs = La->costs;
  const double *const base_cost = La->base_cost;
  int32_t *const status = La->status, *const coll_step = La->coll_step;
  const int64_t t0 = La->chunk_begin + first_local;
  auto put = [&](int g, double v) {
    const int64_t t = t0 + g;
    const int d = s.dead[g];
    if (d == 0) {
      if (status[t] == SFW_ST_VALID) costs[t] = base_cost[t] + social_weight * v;  // not VALID: force_alive run of a costmap-rejected sample
    }
    else if (d >= 2) {
      costs[t] = SFW_COST_INVALID;
      status[t] = SFW_ST_INVALID;
      if (coll_step) coll_step[t] = d - 2;
    }
  };
  if (G == 1) {
    double v = sw_acc;
    for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
    if (lane == 0) put(0, v);
  } else {
    // G > 1 (GA <= 64): per-slot sums were written to swp[]; every sample is reduced
    // with the SAME 64-lane shuffle tree as the G == 1 case (lanes >= A contribute 0),
    // so a sample's cost does not depend on how the wave was organised.
    __syncthreads();
    for (int g = 0; g < Gn; ++g) {
      double v = (lane < A) ? s.swp[g * A + lane] : 0.0;
      for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
      if (lane == 0) put(g, v);
    }
  }
}

// LDS word at a byte offset of the wave's allocation.  The K2 kernels have no static __shared__ data, so their
// dynamic allocation starts at LDS address 0 and an offset IS the address: forming it from `smem` instead costs a
// v_add with the (link-time zero) base per address register in the pair loop.
template <typename T> __device__ __forceinline__ T &lds_at(char *, unsigned byte_off) {
  return *(T *)reinterpret_cast<__attribute__((address_space(3))) T *>(static_cast<uintptr_t>(byte_off));
}

// State (px, py, vx, vy) of the two agents of a pair from the LDS planes: EIGHT ds_read_b64 with immediate plane offsets
// and one wait.  Left to the compiler the loads are merged into four ds_read2st64_b64, which the LDS serves at half the
// rate (8 cycles per wave-instruction for 16 bytes per lane against 2 x 2 for two ds_read_b64, MI355X_MICROARCH.md
// §LDS) — on the LDS pipe that the flat form keeps 84 % busy.  PY/VX/VY: byte distances of the planes (compile-time).
template <int PY, int VX, int VY>
__device__ __forceinline__ void lds_pair_state(uint32_t io, uint32_t jo, double &pix, double &piy, double &vix, double &viy,
                                               double &pjx, double &pjy, double &vjx, double &vjy) {
  asm volatile(
      "ds_read_b64 %0, %8\n\t"
      "ds_read_b64 %4, %9\n\t"
      "ds_read_b64 %1, %8 offset:%10\n\t"
      "ds_read_b64 %5, %9 offset:%10\n\t"
      "ds_read_b64 %2, %8 offset:%11\n\t"
      "ds_read_b64 %6, %9 offset:%11\n\t"
      "ds_read_b64 %3, %8 offset:%12\n\t"
      "ds_read_b64 %7, %9 offset:%12\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(pix), "=&v"(piy), "=&v"(vix), "=&v"(viy), "=&v"(pjx), "=&v"(pjy), "=&v"(vjx), "=&v"(vjy)
      : "v"(io), "v"(jo), "n"(PY), "n"(VX), "n"(VY)
      : "memory");
}

// Two ds_add_f64 (no return) into planes FX / FY of the slot at byte offset `o`, written as asm.  As compiler-generated
// atomics they are DS accesses the wait-count pass has to order against the LDS-direct load of the step's robot records
// (global_load_lds writes LDS behind the compiler's back): it put an s_waitcnt vmcnt(0) in front of them, i.e. the first
// row of every step waited for the load that was issued to be in flight during the whole pair pass.  The LDS executes a
// wave's operations in issue order, so the agent pass reads the sums without further ado (its first s_waitcnt lgkmcnt
// covers them: the counter is decremented in order).
template <int FX, int FY> __device__ __forceinline__ void lds_add_pair(uint32_t o, double x, double y) {
  asm volatile("ds_add_f64 %0, %1 offset:%3\n\tds_add_f64 %0, %2 offset:%4" : : "v"(o), "v"(x), "v"(y), "n"(FX), "n"(FY) : "memory");
}

// ---------------------------------------------------------------------------
// K2, register-resident form: every lane owns NS agent slots (slot = r*64+lane,
// slot -> (sample g, agent i)) whose force accumulator and social-work sum stay in
// VGPRs for the whole rollout; LDS holds the state the partners read.  The unordered
// pairs are walked as a half ring: in row k every agent i meets
// j = (i + k + 1) mod A, so within a row each agent is `i` once and `j` once:
// the i-side force accumulates in registers, the j-side goes through one LDS
// atomic whose addresses are all distinct within the instruction.
// ---------------------------------------------------------------------------
// What a register-form wave does with its G samples, as the body of two kernels: sfw_social_kernel (every block) and
// sfw_social_kernel_mixed (its first blocks; the rest run the flat form's body).  bid / nblk: the wave's block index and the
// block count among the waves of ITS form.
template <typename R, int NS, bool GROUPS>
__device__ __forceinline__ void social_reg_wave(const sfw_launch &L, const int G, char *const smem, const unsigned bid, const unsigned nblk) {
  constexpr int CAP = WAVE * NS;  // GA <= CAP: state planes at compile-time distances
  using off = reg_off<CAP>;       // byte offsets from a slot's px word
  constexpr int PY = off::PY, VX = off::VX, VY = off::VY, FJX = off::FJX, FJY = off::FJY;
  const int lane = threadIdx.x;
  const int A = L.A, O = L.O;
  const int GA = G * A;
  const int NG = GROUPS ? L.NG : 0;  // GROUPS=false instantiation: no group code, lean register budget
  const lds_layout s(smem, A, CAP, GA, G, O, NG, GROUPS ? L.n_grp_mem : 0, true, false);
  const int64_t first_local = static_cast<int64_t>(xcd_contiguous(bid, nblk, static_cast<unsigned>(L.n_xcd))) * G;
  const int64_t remain = item_count(L) - first_local;
  const int Gn = remain < G ? static_cast<int>(remain) : G;
  const int step_begin = L.step_begin, step_end = L.step_end;
  const sfm_consts<R> k0 = make_consts<R, false>(L);  // the prologue's; every step builds its own (below)
  constexpr bool F32 = sizeof(R) == 4;
  if (!stage_wave<GROUPS, true>(L, s, lane, G, Gn, first_local)) {
    // nothing to integrate; items that inherit a contact still pass the verdict on
    if (L.phase == SFW_PHASE_SUFFIX) finish_wave(s, lane, G, Gn, first_local, 0.0);
    if (L.phase == SFW_PHASE_PREFIX && lane < Gn) L.out_dead[first_local + lane] = s.dead[lane];
    return;
  }
  clock_probe(0);

  // ---- this lane's slots --------------------------------------------------
  // A slot is addressed through three registers for the whole rollout: io = 8 * slot (the planes, its {hasgoal, 2 i + Wp}
  // cell and its social-work word at the fixed distances of reg_off), ci = the LDS address of its agent's constants,
  // g4 = 4 * sample (contact flag at off::DEAD, robot record at off::RSB + 8 * g4).
  int sl_[NS], g_[NS], i_[NS];
  uint32_t io_[NS], ci_[NS], g4_[NS];
  bool ok_[NS];
  // Registers hold only the force accumulator and the social-work sum of a slot; position and
  // velocity are re-read from LDS where needed (LDS has the headroom): with them in VGPRs the
  // NS = 1 kernel needs 87 registers, i.e. scratch spills under the 80 that six waves per SIMD allow.
  double fx[NS], fy[NS], sw[NS];
  const uint32_t ac_off = static_cast<uint32_t>(reinterpret_cast<char *>(s.ac) - smem);
  {
    const agent_consts c0 = load_agent_consts(late_args(), F32);
#pragma unroll
    for (int r = 0; r < NS; ++r) {
      const int sl = r * WAVE + lane;
      ok_[r] = sl < GA;
      const int slc = ok_[r] ? sl : 0;
      g_[r] = (G == 1) ? 0 : slc / A;
      i_[r] = slc - g_[r] * A;
      sl_[r] = slc;
      io_[r] = 8u * static_cast<uint32_t>(slc);
      ci_[r] = ac_off + static_cast<uint32_t>(sizeof(sfw_agent_const)) * static_cast<uint32_t>(i_[r]);
      g4_[r] = 4u * static_cast<uint32_t>(g_[r]);
      asm("" : "+v"(g4_[r]));  // opaque: the epilogue's g4 >> 2 must not resurrect g_ (kept in scratch across the rollout otherwise)
      const int i = i_[r];
      fx[r] = fy[r] = sw[r] = 0.0;
      // the slot's Wp switch: a person whose id is not the robot's receives the robot-on-person term (ref :692)
      const int wp_on = (i << 1) | ((i != 0 && s.ac[i].id != c0.robot_id) ? 1 : 0);  // + the agent index (epilogue)
      if (L.resume) {  // resume from the record of the item's (parent) class
        if (ok_[r] && g_[r] < Gn) {
          const sfw_cls_agent c = L.in_state[source_class_of_item(L, first_local + g_[r]) * A + i];
          s.px[sl] = c.px;
          s.py[sl] = c.py;
          s.vx[sl] = c.vx;
          s.vy[sl] = c.vy;
          s.fjx[sl] = s.fjy[sl] = 0.0;
          lds_at<int2>(smem, io_[r] + off::HG) = int2{c.hasgoal, wp_on};
          fx[r] = c.fx;
          fy[r] = c.fy;
          sw[r] = c.sw;
        } else if (ok_[r]) {
          s.px[sl] = s.py[sl] = s.vx[sl] = s.vy[sl] = s.fjx[sl] = s.fjy[sl] = 0.0;
          lds_at<int2>(smem, io_[r] + off::HG) = int2{0, wp_on};
        }
        continue;
      }
      if (ok_[r]) {
        const double px = L.agent_pos[2 * i], py = L.agent_pos[2 * i + 1];
        const double vx = L.agent_vel[2 * i], vy = L.agent_vel[2 * i + 1];
        const int hg = L.agent_c[i].has_goal;
        s.px[sl] = px;
        s.py[sl] = py;
        s.vx[sl] = vx;
        s.vy[sl] = vy;
        s.fjx[sl] = s.fjy[sl] = 0.0;
        lds_at<int2>(smem, io_[r] + off::HG) = int2{hg, wp_on};
        if (i != 0) {
          const agent_k ak = agent_k_lds(s, i);
          desired_force(c0, px, py, vx, vy, hg != 0, ak.gx, ak.gy, ak.gr, ak.dv, fx[r], fy[r]);
          if (O > 0) {
            double tx, ty, sc;
            obstacle_sums<R, false>(k0, c0, obs_global(L.obstacles), px, py, ak.rad, tx, ty, sc);
            fx[r] = fma(tx, sc, fx[r]);
            fy[r] = fma(ty, sc, fy[r]);
          }
        }
        if (L.agent_rest) {  // pairs at exact relative rest in the handed-over state (robot included)
          fx[r] += L.agent_rest[2 * i];
          fy[r] += L.agent_rest[2 * i + 1];
        }
      }
    }
  }
  __syncthreads();
  // Group forces belong to the force at the CURRENT state, so they are added to
  // the starting force after every state update (and once here for step 0).
  auto add_group_forces = [&](const sfm_consts<R> &k, const agent_consts &c) {
    for (int q = lane; q < G * NG; q += WAVE) s.gcen[q] = double2{0.0, 0.0};
    __syncthreads();
#pragma unroll
    for (int r = 0; r < NS; ++r)
      if (ok_[r] && s.grp[i_[r]] >= 0) {
        atomicAdd(&s.gcen[g_[r] * NG + s.grp[i_[r]]].x, s.px[sl_[r]]);
        atomicAdd(&s.gcen[g_[r] * NG + s.grp[i_[r]]].y, s.py[sl_[r]]);
      }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < NS; ++r)
      if (ok_[r] && i_[r] != 0) {
        const double2 gf = group_force<R>(k, c, s, NG, g_[r], A, i_[r], sl_[r], s.px[sl_[r]], s.py[sl_[r]]);
        fx[r] += gf.x;
        fy[r] += gf.y;
      }
  };
  if constexpr (GROUPS) {
    if (!L.resume) add_group_forces(k0, load_agent_consts(late_args(), F32));  // a class record's force already has them
  }

  const int rows = A / 2;            // half ring; for even A the last row is half length
  const bool even = (A & 1) == 0;
  // byte offset (within a plane) of the partner each slot meets next: the partner walks i+1, i+2, ... inside the slot's own
  // sample, i.e. from the word behind the sample's last agent (hi_) back to its first (lo_)
  uint32_t jo_[NS], hi_[NS], lo_[NS];
#pragma unroll
  for (int r = 0; r < NS; ++r) {
    lo_[r] = 8u * static_cast<uint32_t>(sl_[r] - i_[r]);
    hi_[r] = lo_[r] + 8u * static_cast<uint32_t>(A);
  }
  // byte offset of this lane's half record inside a row of the robot-step table (a chunk's row is < 4 GB: one VGPR)
  // (even lane: the position unit of its sample; odd lane: the velocity unit of the sample's grid row, behind the positions)
  uint32_t rs_off = 0;
  if (lane < 2 * Gn) {
    const int64_t rsmp = robot_sample_of_item(L, first_local + (lane >> 1));
    rs_off = static_cast<uint32_t>(((lane & 1) ? L.rstep_stride + vel_row_of(L, rsmp) : rsmp) * static_cast<int64_t>(sizeof(sfw_unit)));
  }

  // Every load of the prologue has landed before the first step, and every scratch reload of a step before the next (the
  // builtin, unlike an asm string, is seen by the compiler's wait-count pass): otherwise the pass, merging the loop's entry
  // and back edges, puts an s_waitcnt vmcnt(0) in front of the first use of such a register inside the row loop, where it
  // also waits for the LDS-direct load of the robot records that is meant to stay in flight during the pair pass.
  constexpr int WAIT_VMCNT0 = 0x0F70;  // gfx9 s_waitcnt encoding: vmcnt(0), expcnt and lgkmcnt left alone
  __builtin_amdgcn_s_waitcnt(WAIT_VMCNT0);
  for (int step = step_begin; step < step_end; ++step) {
    // The pair term's constants are (re)built per step: the ones pinned to VGPRs are opaque to the compiler, which can only
    // spill what it cannot rematerialise — held across the rollout, the asin polynomial's leading coefficient went to scratch
    // around the laser-point loop and came back with a wait inside the row loop.
    const sfm_consts<R> k = make_consts<R, false>(late_args());
    // This step's robot records (32 B per sample, contiguous) go from the K1 table straight to
    // LDS (global_load_lds_dwordx4: no VGPRs), in flight during the pair pass.  The waves of a
    // launch start together and run the same instruction stream, so a load issued where it is
    // consumed stalls every resident wave of the SIMD at once (14 % of wave time in s_waitcnt
    // at cfg2 before this, profiles/r01e).  Lane l < 2 Gn brings half (l & 1) of sample (l >> 1)'s record from
    // the step's table row (a scalar base) plus its own byte offset, fixed for the rollout (G <= 32, plan_for; forming the address from the item
    // tables every step cost ~90 and, at 80 VGPRs, the scratch spills around it).  The pair pass below holds no
    // compiler-visible DS or VMEM instruction (lds_pair_state, lds_add_pair), so nothing in it waits for this load.
#if defined(SFW_ABL_NOROBOT)
    if (step == step_begin)
#endif
    if (lane < 2 * Gn)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(L.ptab + static_cast<int64_t>(step) * L.row_units) + rs_off),
          (__attribute__((address_space(3))) void *)(reinterpret_cast<char *>(s.rsb)), 16, 0, 0);
    // ---- pair pass: social forces at the pre-step state -------------------
    // one pair of slot r with the partner at plane offset jo
    auto pair_with = [&](int r, uint32_t jo) {
      R qx, qy;
      double pix, piy, vix, viy, pjx, pjy, vjx, vjy;
#if defined(SFW_ABL_NOREAD)
      pix = fx[r]; piy = fy[r]; vix = sw[r] + 1.0; viy = 0.5; pjx = fx[r] + static_cast<double>(jo); pjy = fy[r] - 3.0; vjx = 0.25; vjy = static_cast<double>(jo) * 0.001;
#else
      lds_pair_state<PY, VX, VY>(io_[r], jo, pix, piy, vix, viy, pjx, pjy, vjx, vjy);
#endif
#if defined(SFW_ABL_NOMATH)
      qx = R(pix + pjy - vjx); qy = R(piy - pjx + vix * viy + vjy);
#else
      pair_force_state<R>(k, pix, piy, vix, viy, pjx, pjy, vjx, vjy, qx, qy);
#endif
      fx[r] += static_cast<double>(qx);
      fy[r] += static_cast<double>(qy);
      // the partner receives -q: accumulated with the opposite sign, subtracted in the agent pass
#if !defined(SFW_ABL_NOATOM)
      lds_add_pair<FJX, FJY>(jo, static_cast<double>(qx), static_cast<double>(qy));
#endif
    };
    // the partner offset advances by one word per row, so it can only MEET the bound
    auto next_partner = [&](int r) {
      uint32_t jo = jo_[r] + 8u;
      jo = (jo == hi_[r]) ? lo_[r] : jo;
      jo_[r] = jo;
      return jo;
    };
#pragma unroll
    for (int r = 0; r < NS; ++r) jo_[r] = io_[r];
    if constexpr (NS == 1) {
      // One slot per lane: the lane mask of the full rows is the same for every row (the lanes that own a slot), so it is
      // set ONCE around the loop; the half row of an even agent count is peeled off with its own mask.  Written with the
      // mask inside the loop, every row paid 14 scalar instructions of mask arithmetic and a branch.
      if (ok_[0]) {
        const int full_rows = even ? rows - 1 : rows;
        for (int row = 0; row < full_rows; ++row) pair_with(0, next_partner(0));
        if (even && rows > 0) {
          const uint32_t jo = next_partner(0);
          if (i_[0] < rows) pair_with(0, jo);
        }
      }
    } else {
      for (int row = 0; row < rows; ++row) {
        const bool half = even && (row == rows - 1);
#pragma unroll
        for (int r = 0; r < NS; ++r) {
          const uint32_t jo = next_partner(r);
          if (ok_[r] && !(half && i_[r] >= rows)) pair_with(r, jo);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the LDS-direct load above has landed, the row loop's atomics are done
    __syncthreads();

    // ---- per-agent pass ---------------------------------------------------
    const agent_consts c = load_agent_consts(late_args(), F32);
    // (1) integration, contact test, Wp, desired force; a person's new state goes to LDS, the robot keeps its
    // pre-step position there until the laser-point term (2) has been evaluated at it
    const bool with_obs = c.O > 0;
#pragma unroll
    for (int r = 0; r < NS; ++r) {
#if defined(SFW_ABL_NOAGENT)
      if (false) {
#else
      if (ok_[r] && lds_at<int>(smem, off::DEAD + g4_[r]) == 0) {
#endif
        const uint32_t io = io_[r], ci = ci_[r];
        const bool robot = i_[r] == 0;
        const sfw_robot_step rs = lds_at<sfw_robot_step>(smem, off::RSB + 8u * g4_[r]);
        double px = lds_at<double>(smem, io), py = lds_at<double>(smem, io + PY), vx = lds_at<double>(smem, io + VX),
               vy = lds_at<double>(smem, io + VY);
        const double2 goal = lds_at<double2>(smem, ci), grdv = lds_at<double2>(smem, ci + 16u);
        const agent_k ak{goal.x, goal.y, grdv.x, grdv.y, 0.0, 0};
        const int2 cell = lds_at<int2>(smem, io + off::HG);
        int hg = cell.x;
        bool contact;
        double nfx, nfy;
        const double w = agent_step<R>(k, c, rs, ak, robot, (cell.y & 1) != 0, hg, contact, px, py, vx, vy,
                                       fx[r] - lds_at<double>(smem, io + FJX), fy[r] - lds_at<double>(smem, io + FJY), nfx, nfy);
        if (hg != cell.x) lds_at<int>(smem, io + off::HG) = hg;  // goal reached: popped
#if !defined(SFW_ABL_NOATOM) && !defined(SFW_ABL_NOREAD) && !defined(SFW_ABL_NOMATH) && !defined(SFW_ABL_KEEPALIVE)
        if (contact) lds_at<int>(smem, off::DEAD + g4_[r]) = 2 + step;  // >= 2: rejected by contact at `step`
#endif
        fx[r] = nfx;
        fy[r] = nfy;
        if (robot && with_obs) {
          lds_at<double>(smem, io + off::SW) = w;  // Wr = social part + obstacle part, summed in (2) before it joins the social work
        } else {
          sw[r] += w;
        }
        if (robot) {  // the robot does not move by SFM: post-step record (ref :600, :604 robot-local twist)
          px = rs.x;
          py = rs.y;
          vx = rs.vx;
          vy = rs.vy;
        }
        if (!(robot && with_obs)) {
          lds_at<double>(smem, io) = px;
          lds_at<double>(smem, io + PY) = py;
          lds_at<double>(smem, io + VX) = vx;
          lds_at<double>(smem, io + VY) = vy;
        }
        lds_at<double>(smem, io + FJX) = 0.0;
        lds_at<double>(smem, io + FJY) = 0.0;
      }
    }
    // (2) obstacle term, ONE loop over the laser points for every lane of the wave: the robot needs it at its
    // pre-step position (Wr's obstacle part), a person at its new position (next starting force)
    if (with_obs) {
#pragma unroll
      for (int r = 0; r < NS; ++r)
        if (ok_[r] && lds_at<int>(smem, off::DEAD + g4_[r]) == 0) {
          const uint32_t io = io_[r];
          double tx, ty, sc;
          obstacle_sums<R, false>(k, c, obs_global(c.obstacles), lds_at<double>(smem, io), lds_at<double>(smem, io + PY),
                           lds_at<double>(smem, ci_[r] + 32u), tx, ty, sc);
          if (i_[r] == 0) {
            sw[r] += lds_at<double>(smem, io + off::SW) + fast_norm(tx * sc, ty * sc);
            const sfw_robot_step r2 = lds_at<sfw_robot_step>(smem, off::RSB + 8u * g4_[r]);
            lds_at<double>(smem, io) = r2.x;
            lds_at<double>(smem, io + PY) = r2.y;
            lds_at<double>(smem, io + VX) = r2.vx;
            lds_at<double>(smem, io + VY) = r2.vy;
          } else {
            fx[r] = fma(tx, sc, fx[r]);
            fy[r] = fma(ty, sc, fy[r]);
          }
        }
    }
    if (const double *const pin = late_args()->pin_rest) {  // (pin_at_rest: no launch of a BASELINE workload comes here)
#pragma unroll
      for (int r = 0; r < NS; ++r)
        if (ok_[r] && lds_at<int>(smem, off::DEAD + g4_[r]) == 0 &&
            pin_at_rest(pin, lds_at<sfw_robot_step>(smem, off::RSB + 8u * g4_[r]))) {
          const int i = lds_at<int2>(smem, io_[r] + off::HG).y >> 1;
          if (i == 0) {
            fx[r] += pin[2];
            fy[r] += pin[3];
          } else {
            sw[r] += pin[4 + i];
          }
        }
    }
    __syncthreads();
    bool any_live = false;
    for (int g = 0; g < G; ++g) any_live |= (s.dead[g] == 0);
    if (!any_live) break;
    if constexpr (GROUPS) add_group_forces(k, c);
      }

  const late_launch Le = late_args();
  if (Le->phase == SFW_PHASE_PREFIX) {  // leave the class records
    sfw_cls_agent *const out_state = Le->out_state;
    // The record index is formed HERE from values the rollout keeps anyway (4 * sample, the late-read A, the agent index
    // parked in the slot's cell): computed from L.A and i_ it is the same expression as the prologue's resume index, and
    // the compiler kept that 64-bit value in scratch across the whole rollout.
    const int64_t A_late = Le->A;
#pragma unroll
    for (int r = 0; r < NS; ++r)
      if (ok_[r] && static_cast<int>(g4_[r] >> 2) < Gn) {
        const uint32_t io = io_[r];
        sfw_cls_agent c;
        c.px = lds_at<double>(smem, io); c.py = lds_at<double>(smem, io + PY);
        c.vx = lds_at<double>(smem, io + VX); c.vy = lds_at<double>(smem, io + VY);
        c.fx = fx[r]; c.fy = fy[r]; c.sw = sw[r];
        const int2 cell = lds_at<int2>(smem, io + off::HG);
        c.hasgoal = cell.x;
        c.pad = 0;
        out_state[(first_local + static_cast<int64_t>(g4_[r] >> 2)) * A_late + (cell.y >> 1)] = c;
      }
    if (lane < Gn) Le->out_dead[first_local + lane] = s.dead[lane];
    clock_probe(1);
    return;
  }
  double sw_acc = 0.0;
#pragma unroll
  for (int r = 0; r < NS; ++r) {
    if (G == 1) sw_acc += ok_[r] ? sw[r] : 0.0;
    else if (ok_[r]) lds_at<double>(smem, io_[r] + off::SW) = sw[r];
  }
  finish_wave(s, lane, G, Gn, first_local, sw_acc);
  clock_probe(1);
}
template <typename R, int NS, bool GROUPS>
__global__ void __launch_bounds__(WAVE, (NS == 1 && !GROUPS) ? 6 : 1) sfw_social_kernel(const sfw_launch L, const int G) {
  sfwm::fp_mode_for_omod();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  social_reg_wave<R, NS, GROUPS>(L, G, smem, blockIdx.x, gridDim.x);
}

// ---------------------------------------------------------------------------
// K2, flat form (one sample per wave): the A(A-1)/2 unordered pairs are
// flattened over the 64 lanes, u -> row = u / A, i = u % A, j = (i + row + 1) % A,
// so lane utilisation is ~100 % for any A (the register-resident form idles
// 64*NS - A lanes).  All agent state lives in LDS; both sides of a pair go through
// LDS atomics, into two accumulators per agent (received "as i" / "as j") so that
// each accumulator's summation order is a function of u only.
//
// The u -> (i, j) map is the same for every step of every sample, so it is not
// recomputed: sfw_pair_table_kernel writes it once per agent set as two uint16
// arrays of plane byte offsets (8*i, 8*j), padded to a multiple of 64 with the
// DUMMY slot 8*A (a record behind the last agent: its pair evaluates on zeros and
// lands in accumulators nobody reads, so the loop needs no lane predicate).  A
// lane's entries arrive as two zero-extending global_load_ushort from a
// wave-uniform base (scalar address arithmetic, no unpacking): the pair loop
// spends no VALU issue on indices.  The loads are written as asm, one iteration
// ahead, into two alternating register pairs (loop unrolled by two: no copies).
// ---------------------------------------------------------------------------
// n_dummy: free slots behind the last agent (plane capacity - A >= 1).  The padding lanes of the last iteration are spread
// over them: with ONE dummy slot all of them added into the same four accumulators — up to 63 same-address ds_add_f64 per
// instruction, every step (a control cycle with 5 people: 49 of 64 lanes; A = 24 on a full grid: K2 +20 %).
__global__ void __launch_bounds__(256) sfw_pair_table_kernel(uint16_t *tab, int A, int n_entries, int n_dummy) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n_entries) return;
  const int P = A * (A - 1) / 2;
  int i = A + (u >= P ? (u - P) % n_dummy : 0), j = i;  // a dummy slot
  if (u < P) {
    const int row = u / A;
    i = u - row * A;
    j = i + row + 1;
    j = (j >= A) ? j - A : j;
  }
  tab[u] = static_cast<uint16_t>(8 * i);
  tab[n_entries + u] = static_cast<uint16_t>(8 * j);
}

__device__ __forceinline__ void load_pair_entries(const uint16_t *ti, const uint16_t *tj, uint32_t lane_off, uint32_t &io,
                                                  uint32_t &jo) {
#ifdef SFW_DBG_C_LOADS
  io = ti[lane_off / 2];
  jo = tj[lane_off / 2];
  return;
#endif
  // s_nop 4: the table addresses may have just been reloaded from a spill lane (v_readlane: VALU writes an SGPR), and a
  // VMEM instruction reading such an SGPR needs 5 wait states — the compiler's hazard recogniser does not look inside
  // an asm block (found as a memory fault that only showed with many waves per SIMD)
  asm volatile("s_nop 4\n\tglobal_load_ushort %0, %2, %3\n\tglobal_load_ushort %1, %2, %4"
               : "=&v"(io), "=&v"(jo)
               : "v"(lane_off), "s"(ti), "s"(tj));
}
__device__ __forceinline__ void wait_pair_entries(uint32_t &io, uint32_t &jo) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(io), "+v"(jo));
}

#ifndef SFW_FIRST_PAIRS_IN_REGS
#define SFW_FIRST_PAIRS_IN_REGS 1  // flat form: the first 64 pairs' table entries stay in registers across the rollout
#endif
#ifndef SFW_FLAT_WAVES
#define SFW_FLAT_WAVES 5  // waves per SIMD the flat kernel is compiled for (<= 96 VGPRs; tuning knob, csrc/Makefile EXTRA)
#endif
#ifndef SFW_FLAT_WAVES_NOOBS
#define SFW_FLAT_WAVES_NOOBS 6  // ... the kernel without the laser-point pass (77 VGPRs: six waves per SIMD)
#endif
// Synchronisation of ONE wave with itself: the LDS executes a wave's operations in issue order, so all that is needed is
// that the compiler keeps the order.  (What __syncthreads() is for a block of one wave — the flat kernel's — after the
// backend has dropped its s_barrier; the cycle kernel runs the same body as one wave of a larger block, where a block
// barrier would wait for waves that are doing something else.)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// What the flat form's wave does with its sample, as the body of two kernels:
//   CYCLE = false  sfw_social_kernel_flat: one wave per block, robot records from the K1->K2 table, results by finish_wave;
//   CYCLE = true   sfw_cycle_kernel (small grids: a control cycle in ONE launch): wave 0 of a block whose other waves roll the
//                  robot out and check its footprint meanwhile.  The robot's records come from that rollout's LDS arrays
//                  (`k1`) as they come to stand (cycle_wait_ready), and the wave hands back its social-work sum and contact
//                  verdict instead of writing the cost.
// Same statements either way: costs are bit-identical.
struct cycle_result {
  double social_work;  // lane 0
  int dead;            // 0, or 2 + the step of a pedestrian contact
  int ready;           // steps whose robot records stand in the block's LDS arrays (written by the wave that rolls the robot out)
  int fdone;           // waves that have finished their footprint tasks
  int legal;           // the costmap's verdict (the last of those waves scans the codes)
  double base0, base;  // pedestrian-free cost terms without / with the costmap term
};
// Wait until the robot's records of the first `need` steps stand (sfw_cycle_kernel): a poll of one LDS word, wave-uniform.
__device__ __forceinline__ int cycle_wait_ready(const cycle_result *res, int need) {
  int r;
  while ((r = *const_cast<const volatile int *>(&res->ready)) < need) __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return r;
}
#define K2_SYNC() do { if constexpr (CYCLE) wave_sync(); else __syncthreads(); } while (0)
template <typename R, bool GROUPS, int CAP, bool OBS, bool CYCLE>
__device__ __forceinline__ void social_flat_wave(const sfw_launch &L, char *const smem, const k1s_lds *const k1, cycle_result *const res,
                                                 const unsigned bid, const unsigned nblk, const int64_t item_base) {
  const int lane = threadIdx.x;
  const int A = L.A, O = L.O;
  const int NG = GROUPS ? L.NG : 0;
  const int cap = CAP > 0 ? CAP : ((A + 2) & ~1);  // > A: the dummy slot of the padded pair table
  const int PY = 8 * cap, VX = 16 * cap, VY = 24 * cap, FJX = 32 * cap, FJY = 40 * cap, FCX = 48 * cap,
            FCY = 56 * cap;  // byte offsets from px[] (immediates when CAP > 0)
  (void)FCX;
  (void)FCY;
  // (the cycle kernel's wave keeps the agents' constants in LDS like the GROUPS kernels: it has the CU's LDS to itself, and the
  // 48-byte-per-agent load from memory at the top of every agent pass is latency nothing hides on a lone wave)
  constexpr bool CONSTS_IN_LDS = GROUPS || CYCLE;
  const lds_layout s(smem, A, cap, A, 1, O, NG, GROUPS ? L.n_grp_mem : 0, CONSTS_IN_LDS, true, L.k.obs_lds != 0);
  const int64_t first_local = CYCLE ? static_cast<int64_t>(bid) : item_base + xcd_contiguous(bid, nblk, static_cast<unsigned>(L.n_xcd));
  const sfm_consts<R> k0 = make_consts<R, true>(L);  // the prologue's; every step builds its own (below)
  // the five force constants stay in scalar registers for the rollout (a step's copy into VGPRs is five v_mov; read from the
  // kernel arguments every step, a lone wave waited for the scalar loads at the top of each)
  const sfw_force_k<R> &fk = force_k<R>(L);
  const R s_lambda = sfwm::sgpr_const(fk.lambda), s_nig = sfwm::sgpr_const(fk.neg_l2e_inv_gamma), s_lfs = sfwm::sgpr_const(fk.l2_f_social),
          s_cvel = sfwm::sgpr_const(fk.c_vel), s_cang = sfwm::sgpr_const(fk.c_ang);
  constexpr bool F32 = sizeof(R) == 4;
  const int step_begin = L.step_begin, step_end = L.step_end;
  if constexpr (CYCLE) {  // (the block runs this wave for a scored sample only; the costmap's verdict is formed beside it)
    stage_consts<GROUPS, CONSTS_IN_LDS>(L, s, lane);
    if (lane == 0) s.dead[0] = 0;
    wave_sync();
  } else {
    if (!stage_wave<GROUPS, GROUPS>(L, s, lane, 1, 1, first_local)) {
      if (L.phase == SFW_PHASE_SUFFIX) finish_wave(s, lane, 1, 1, first_local, 0.0);  // inherited contact
      if (L.phase == SFW_PHASE_PREFIX && lane == 0) L.out_dead[first_local] = s.dead[0];
      return;
    }
  }
  clock_probe(0);
  // wave-uniform, but formed from table loads: made scalar explicitly (as a VGPR pair it is held — in scratch, once the
  // laser-point pass needs the registers — across the whole rollout for the two lanes that fetch the robot records)
  const int64_t rsample_v = CYCLE ? 0 : robot_sample_of_item(L, first_local);
  const int64_t rsample = static_cast<int64_t>(
      (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(rsample_v >> 32)))) << 32) |
      static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(rsample_v))));

  if (!CYCLE && L.resume) {  // resume from the record of the item's (parent) class
    const sfw_cls_agent *rec = L.in_state + source_class_of_item(L, first_local) * A;
    for (int sl = lane; sl < A; sl += WAVE) {
      const sfw_cls_agent c = rec[sl];
      s.px[sl] = c.px;
      s.py[sl] = c.py;
      s.vx[sl] = c.vx;
      s.vy[sl] = c.vy;
      s.hasgoal8[sl] = static_cast<unsigned char>(c.hasgoal != 0);
      s.swp[sl] = c.sw;
      s.fcx[sl] = c.fx;
      s.fcy[sl] = c.fy;
      s.fjx[sl] = s.fjy[sl] = 0.0;
    }
  } else {
    const agent_consts c0 = load_agent_consts(late_args(), F32);
    for (int sl = lane; sl < A; sl += WAVE) {
      const double px = L.agent_pos[2 * sl], py = L.agent_pos[2 * sl + 1];
      const double vx = L.agent_vel[2 * sl], vy = L.agent_vel[2 * sl + 1];
      const sfw_agent_const c = L.agent_c[sl];
      s.px[sl] = px;
      s.py[sl] = py;
      s.vx[sl] = vx;
      s.vy[sl] = vy;
      s.hasgoal8[sl] = static_cast<unsigned char>(c.has_goal != 0);
      s.swp[sl] = 0.0;
      double fx = 0.0, fy = 0.0;
      if (sl != 0) {
        desired_force(c0, px, py, vx, vy, c.has_goal != 0, c.goal_x, c.goal_y, c.goal_radius, c.desired_velocity, fx, fy);
        if (OBS && O > 0) {
          double tx, ty, sc;
          obstacle_sums<R, SFW_FLAT_PIPELINED>(k0, c0, obs_global(L.obstacles), px, py, c.radius, tx, ty, sc);
          fx = fma(tx, sc, fx);
          fy = fma(ty, sc, fy);
        }
      }
      if (L.agent_rest) {  // pairs at exact relative rest in the handed-over state (robot included)
        fx += L.agent_rest[2 * sl];
        fy += L.agent_rest[2 * sl + 1];
      }
      s.fcx[sl] = fx;
      s.fcy[sl] = fy;
      s.fjx[sl] = s.fjy[sl] = 0.0;
    }
  }
  if (OBS && O > 0) {
    const agent_consts c0 = load_agent_consts(late_args(), F32);
    for (int sl = lane; sl < A; sl += WAVE) s.oscale[sl] = obstacle_scale<R>(k0, c0, L.agent_c[sl].radius);
  }
  for (int sl = A + lane; sl < cap; sl += WAVE) {  // the dummy slots: finite state, accumulators nobody reads
    s.px[sl] = s.py[sl] = s.vx[sl] = s.vy[sl] = 0.0;
    s.fcx[sl] = s.fcy[sl] = s.fjx[sl] = s.fjy[sl] = 0.0;
  }
  K2_SYNC();
  auto add_group_forces = [&](const sfm_consts<R> &k, const agent_consts &c) {
    for (int q = lane; q < NG; q += WAVE) s.gcen[q] = double2{0.0, 0.0};
    K2_SYNC();
    for (int sl = lane; sl < A; sl += WAVE)
      if (s.grp[sl] >= 0) {
        atomicAdd(&s.gcen[s.grp[sl]].x, s.px[sl]);
        atomicAdd(&s.gcen[s.grp[sl]].y, s.py[sl]);
      }
    K2_SYNC();
    for (int sl = lane; sl < A; sl += WAVE)
      if (sl != 0) {
        const double2 gf = group_force<R>(k, c, s, NG, 0, A, sl, sl, s.px[sl], s.py[sl]);
        s.fcx[sl] += gf.x;
        s.fcy[sl] += gf.y;
      }
    K2_SYNC();
  };
  if constexpr (GROUPS) {
    if (!L.resume) add_group_forces(k0, load_agent_consts(late_args(), F32));  // a class record's force already has them
  }

  const int P = A * (A - 1) / 2;  // unordered pairs
  const int n_it = (P + WAVE - 1) / WAVE;
  const uint16_t *const tab_i = L.pair_tab, *const tab_j = L.pair_tab + static_cast<size_t>(n_it) * WAVE;
  const uint32_t lane_off = 2u * static_cast<uint32_t>(lane);

  // robot record of a step: lanes 0 and 1 bring 16 bytes each from the K1 table straight to LDS
  // (lane 0: the position unit of the sample; lane 1: the velocity unit of its grid row, behind the row's positions — two
  // wave-uniform unit indices, the lane picks one)
  const int64_t unit_vel_v = CYCLE ? 0 : L.rstep_stride + vel_row_of(L, rsample);  // (an integer division: VALU work, made scalar)
  const int64_t unit_pos = rsample,
                unit_vel = static_cast<int64_t>(
                    (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(unit_vel_v >> 32)))) << 32) |
                    static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(unit_vel_v))));
  auto fetch_robot = [&](const sfw_unit *ptab, int64_t row_units, int st, int buf) {
    if constexpr (CYCLE) return;  // (the records are in the block's LDS: read where they are used)
    int l2 = lane;
    asm volatile("" : "+v"(l2));  // opaque: the lane's choice is formed here, not held across the rollout
    if (l2 < 2) {
      const char *src = reinterpret_cast<const char *>(ptab + static_cast<int64_t>(st) * row_units + (l2 ? unit_vel : unit_pos));
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)(reinterpret_cast<char *>(s.rsb + buf)), 16,
                                       0, 0);
    }
  };
  if (step_begin < step_end) fetch_robot(L.ptab, L.row_units, step_begin, step_begin & 1);
  // (the 64-double planes only — crowds of up to 63 agents, every control cycle —: the larger capacities' kernels have no
  // registers to spare, tests/test_kernel_resources.py)
  constexpr bool FIRST_IN_REGS = SFW_FIRST_PAIRS_IN_REGS && CAP == 64;
  uint32_t i0 = 0, j0 = 0;  // the first iteration's pair-table entries (see the pair loop)
  if constexpr (FIRST_IN_REGS) {
    if (n_it > 0) {
      load_pair_entries(tab_i, tab_j, lane_off, i0, j0);
      wait_pair_entries(i0, j0);
    }
  }

#if defined(SFW_ABL_HALF_LDS)
  double abl_ix = s.px[lane < A ? lane : 0], abl_iy = s.py[lane < A ? lane : 0], abl_fx = 0.0, abl_fy = 0.0;
#endif
  // one pair per lane: both agents from LDS, the force into both agents' accumulators
  auto pair_at = [&](const sfm_consts<R> &k, uint32_t io, uint32_t jo) {
    R qx, qy;
    if constexpr (CAP > 0) {
      double pix, piy, vix, viy, pjx, pjy, vjx, vjy;
#if defined(SFW_ABL_NOREAD)
      pix = static_cast<double>(io); piy = 1.0; vix = 0.3; viy = 0.5; pjx = static_cast<double>(jo) * 0.37; pjy = -3.0; vjx = 0.25; vjy = static_cast<double>(jo) * 0.001;
#elif defined(SFW_ABL_HALF_LDS)
      // upper bound of a lane-stationary organisation: the i side from registers, only the partner's state from LDS
      pix = abl_ix; piy = abl_iy; vix = 0.3; viy = 0.5;
      asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:%5\n\tds_read_b64 %2, %4 offset:%6\n\tds_read_b64 %3, %4 offset:%7\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(pjx), "=&v"(pjy), "=&v"(vjx), "=&v"(vjy) : "v"(jo), "n"(8 * CAP), "n"(16 * CAP), "n"(24 * CAP) : "memory");
#else
      lds_pair_state<8 * CAP, 16 * CAP, 24 * CAP>(io, jo, pix, piy, vix, viy, pjx, pjy, vjx, vjy);
#endif
#if defined(SFW_ABL_NOMATH)
      qx = R(pix + pjy - vjx); qy = R(piy - pjx + vix * viy + vjy);
#else
      pair_force_state<R>(k, pix, piy, vix, viy, pjx, pjy, vjx, vjy, qx, qy);
#endif
    } else {
      pair_force_state<R>(k, lds_at<double>(smem, io), lds_at<double>(smem, io + PY), lds_at<double>(smem, io + VX),
                          lds_at<double>(smem, io + VY), lds_at<double>(smem, jo), lds_at<double>(smem, jo + PY),
                          lds_at<double>(smem, jo + VX), lds_at<double>(smem, jo + VY), qx, qy);
    }
#if defined(SFW_ABL_NOATOM)
    asm volatile("" :: "v"(qx), "v"(qy));
#elif defined(SFW_ABL_HALF_LDS)
    abl_fx += static_cast<double>(qx);
    abl_fy += static_cast<double>(qy);
    atomicAdd(&lds_at<double>(smem, jo + FJX), static_cast<double>(qx));
    atomicAdd(&lds_at<double>(smem, jo + FJY), static_cast<double>(qy));
#else
    atomicAdd(&lds_at<double>(smem, io + FCX), static_cast<double>(qx));
    atomicAdd(&lds_at<double>(smem, io + FCY), static_cast<double>(qy));
    // j receives -q: accumulated with the opposite sign, subtracted in the agent pass
    atomicAdd(&lds_at<double>(smem, jo + FJX), static_cast<double>(qx));
    atomicAdd(&lds_at<double>(smem, jo + FJY), static_cast<double>(qy));
#endif
  };

  // The kernel exists twice: with and without the laser-point pass (OBS).  With it, the pair term's constants pinned to VGPRs
  // (five force constants, two leading polynomial coefficients: 14 registers) are refreshed per step — they are opaque to the
  // compiler, so held across the rollout they also sit through that pass, whose four-agents-per-lane loop needs the
  // registers.  Without it they are built once: the refresh is ~30 vector instructions per step (the v_mov and the reloads of
  // the scalar constants it pushes out), 1.5 % of a step at the target crowd and 5 % of a lone wave's step with 5 people.
  // (Two kernels, not two loops in one: with both in one function the backend fails — "illegal VGPR to SGPR copy".)
  [[maybe_unused]] int cyc_ready = 0;  // (CYCLE: steps whose robot records are known to stand)
  for (int step = step_begin; step < step_end; ++step) {
    sfm_consts<R> k = k0;  // the scalar part (15 polynomial coefficients) as built in front of the rollout ...
    if constexpr (OBS) {
      k.pc.leading_here();   // ... the vector part afresh: two leading coefficients and the five force constants
      k.lambda = sfwm::vgpr_copy_here(s_lambda);
      k.neg_l2e_inv_gamma = sfwm::vgpr_copy_here(s_nig);
      k.l2_f_social = sfwm::vgpr_copy_here(s_lfs);
      k.c_vel = sfwm::vgpr_copy_here(s_cvel);
      k.c_ang = sfwm::vgpr_copy_here(s_cang);
    }
    if constexpr (FIRST_IN_REGS) {
    if (n_it > 0) {
      // The first 64 pairs' table entries stay in registers for the whole rollout (i0, j0: loaded in front of it): a wave with
      // the GPU to itself — a control cycle's, a coarse shared-prefix level's — otherwise opens every step with a round trip
      // to the L1 that nothing hides (crowds of up to 11 agents have no other pairs).
      uint32_t ia, ja, ib, jb;
      const uint16_t *ti = tab_i + WAVE, *tj = tab_j + WAVE;  // the second iteration's entries
      int left = n_it - 1;  // iterations after the first: a plain scalar countdown
      if (left >= 1) load_pair_entries(ti, tj, lane_off, ia, ja);
      pair_at(k, i0, j0);
      ti += WAVE;
      tj += WAVE;
      for (; left >= 2; left -= 2, ti += 2 * WAVE, tj += 2 * WAVE) {
        wait_pair_entries(ia, ja);  // also covers the robot record issued a step ago
        load_pair_entries(ti, tj, lane_off, ib, jb);
        pair_at(k, ia, ja);
        wait_pair_entries(ib, jb);
        if (left > 2) load_pair_entries(ti + WAVE, tj + WAVE, lane_off, ia, ja);
        pair_at(k, ib, jb);
      }
      if (left == 1) {
        wait_pair_entries(ia, ja);
        pair_at(k, ia, ja);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the robot record issued a step ago (landed long since)
    } else {
    if (n_it > 0) {
      uint32_t ia, ja, ib, jb;
      load_pair_entries(tab_i, tab_j, lane_off, ia, ja);
      int left = n_it;  // iterations still to run: a plain scalar countdown (the it < n_it form kept its state in a VGPR lane)
      const uint16_t *ti = tab_i + WAVE, *tj = tab_j + WAVE;
      for (; left >= 2; left -= 2, ti += 2 * WAVE, tj += 2 * WAVE) {
        wait_pair_entries(ia, ja);  // also covers the robot record issued a step ago
        load_pair_entries(ti, tj, lane_off, ib, jb);
        pair_at(k, ia, ja);
        wait_pair_entries(ib, jb);
        if (left > 2) load_pair_entries(ti + WAVE, tj + WAVE, lane_off, ia, ja);
        pair_at(k, ib, jb);
      }
      if (left == 1) {
        wait_pair_entries(ia, ja);
        pair_at(k, ia, ja);
      }
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    }
    K2_SYNC();
#if defined(SFW_ABL_HALF_LDS)
    if (lane < A) { s.fcx[lane] += abl_fx; s.fcy[lane] += abl_fy; abl_ix = s.px[lane] + 1e-3; abl_iy = s.py[lane]; abl_fx = abl_fy = 0.0; }
#endif
    // ---- per-agent pass: its parameters are read here, not held across the pair loop -----------
    const late_launch La = late_args();
    const agent_consts c = load_agent_consts(La, F32);
    const sfw_agent_const *const agent_c = La->agent_c;
    // the cycle kernel's hand-over, behind the pair pass (which needs the pre-step state only): the wave that rolls the robot
    // out publishes how many steps' records stand in LDS — the first few early, then all; a wait is a poll of that word
    if constexpr (CYCLE) {
      if (step >= cyc_ready) cyc_ready = cycle_wait_ready(res, step + 1);
    }
    const sfw_robot_step rs = CYCLE ? sfw_robot_step{k1->xs[step + 1], k1->ys[step + 1], k1->vxs[step], k1->vys[step]} : s.rsb[step & 1];
    // (fetched ONE step ahead.  Two steps ahead into a third slot — so that no barrier of a step finds the fetch still in
    // flight — was measured on the control cycle and is slower: +3 % without laser points, +5 % with them, 32 bytes of scratch)
    if (step + 1 < step_end) fetch_robot(La->ptab, La->row_units, step + 1, (step + 1) & 1);
    const bool with_obs = OBS && c.O > 0;  // (the GROUPS kernels exist with OBS only and serve both cases)
    // the lane index, opaque once per step: the 64-bit byte offset of the lane's agent constants (48 * lane) is then formed
    // here (two instructions) instead of being held — in scratch, for the 104- and 208-double capacities — across the pair loop
    int lane_s = lane;
    asm volatile("" : "+v"(lane_s));
#if defined(SFW_ABL_NOAGENT)
    for (int sl = lane_s; sl < 0; sl += WAVE) {
#else
    for (int sl = lane_s; sl < A; sl += WAVE) {
#endif
      const agent_k ak = CONSTS_IN_LDS ? agent_k_lds(s, sl) : agent_k_global(agent_c, sl);
      double px = s.px[sl], py = s.py[sl], vx = s.vx[sl], vy = s.vy[sl];
      double nfx, nfy;
      int hg = s.hasgoal8[sl];
      const int hg0 = hg;
      bool contact;
      const double w = agent_step<R>(k, c, rs, ak, sl == 0, ak.id != c.robot_id, hg, contact, px, py, vx, vy,
                                     s.fcx[sl] - s.fjx[sl], s.fcy[sl] - s.fjy[sl], nfx, nfy);
      if (sl != 0 && hg0) s.hasgoal8[sl] = static_cast<unsigned char>(hg);
#if !defined(SFW_ABL_NOATOM) && !defined(SFW_ABL_NOREAD) && !defined(SFW_ABL_NOMATH) && !defined(SFW_ABL_KEEPALIVE) && !defined(SFW_ABL_HALF_LDS)
      if (contact) s.dead[0] = 2 + step;  // >= 2: rejected by contact at `step`
#endif
      if (sl == 0 && with_obs) {
        s.wr[0] = w;  // Wr = social part + obstacle part: summed below, then added to the robot's social work
      } else {
        s.swp[sl] += w;
      }
      if (sl == 0) {  // the robot does not move by SFM: post-step record (ref :600, :604 robot-local twist)
        px = rs.x;
        py = rs.y;
        vx = rs.vx;
        vy = rs.vy;
      }
      if (!(sl == 0 && with_obs)) {  // with laser points the robot keeps its pre-step position until their pass is done
        s.px[sl] = px;
        s.py[sl] = py;
        s.vx[sl] = vx;
        s.vy[sl] = vy;
      }
      s.fcx[sl] = nfx;
      s.fcy[sl] = nfy;
      s.fjx[sl] = 0.0;
      s.fjy[sl] = 0.0;
    }
    if (with_obs) {
      // Obstacle term: the robot at its pre-step position (Wr's obstacle part), a person at its new position.
      K2_SYNC();
      if (c.obs_tasks) {
        // Every (agent, segment) pair is a task; the wave walks them 256 at a time, 16 agents x 16 segments per round (same
        // sums in the same order as obstacle_sums).  Lane l: segment l / 4 of the agents a0 + l % 4 + {0, 4, 8, 12} — four
        // neighbouring lanes read the same point — the points through per-lane loads from global memory (L1), one load per
        // point and lane for up to four agents (obstacle_segment_multi).
        const int Lseg = (c.O + OBS_SEG - 1) / OBS_SEG, seg = lane >> 2, sub = lane & (OBS_AGENT_LANES - 1);
        const int ob = min(seg * Lseg, c.O), oe = min(ob + Lseg, c.O);
        const R nis = sfwm::vgpr_const(static_cast<R>(-c.l2e_inv_sigma));
        // The wave's LDS copy in an under-filled launch, else global memory — as two instantiations of the loop, not one
        // over a generic pointer: a flat load counts on vmcnt too, and waiting for it a lone wave also waited, in front of
        // its first point of every step, for the robot record of the next step that was fetched to be in flight until then.
        const double2 *const pts_g = reinterpret_cast<const double2 *>(c.obstacles);
        const obs_lds_ptr pts_l = (obs_lds_ptr)s.obs;
        const bool in_lds = s.obs != nullptr;
        const double *const part = reinterpret_cast<const double *>(s.opart);
        constexpr int KA = OBS_AGENTS_PER_LANE;
        for (int a0 = 0; a0 < A; a0 += OBS_AGENT_LANES * KA) {
          // lane (seg, sub) takes agents a0 + sub + 4 j, j < nj (wave-uniform)
          const int nj = min(KA, (A - a0 + OBS_AGENT_LANES - 1) / OBS_AGENT_LANES);
          double pxj[KA], pyj[KA];
          R axj[KA + 1], ayj[KA + 1];  // (+1: the reduction below takes the slots in twos)
#pragma unroll
          for (int j = 0; j < KA; ++j) {
            pxj[j] = pyj[j] = 0.0;
            if (j < nj) {  // (wave-uniform; a lane past the last agent evaluates the last agent's position, unused)
              const int a = min(a0 + sub + OBS_AGENT_LANES * j, A - 1);
              pxj[j] = s.px[a];
              pyj[j] = s.py[a];
            }
          }
#pragma unroll
          for (int j = 0; j <= KA; ++j) axj[j] = ayj[j] = R(0);
          auto run = [&](auto pts) {
            switch (nj) {
              case 1: obstacle_segment_multi<R, 1>(k, pts, ob, oe, pxj, pyj, nis, axj, ayj); break;
              case 2: obstacle_segment_multi<R, 2>(k, pts, ob, oe, pxj, pyj, nis, axj, ayj); break;
              case (KA > 3 ? 3 : -1): obstacle_segment_multi<R, 3>(k, pts, ob, oe, pxj, pyj, nis, axj, ayj); break;
              default: obstacle_segment_multi<R, KA>(k, pts, ob, oe, pxj, pyj, nis, axj, ayj); break;
            }
          };
          // (a GPU-filling launch — points in global memory — runs the loop with wave-uniform trip counts)
          auto run_uniform = [&](auto pts) {
            switch (nj) {
              case 1: obstacle_segment_multi_uniform<R, 1>(k, pts, c.O, Lseg, seg, pxj, pyj, nis, axj, ayj); break;
              case 2: obstacle_segment_multi_uniform<R, 2>(k, pts, c.O, Lseg, seg, pxj, pyj, nis, axj, ayj); break;
              case (KA > 3 ? 3 : -1): obstacle_segment_multi_uniform<R, 3>(k, pts, c.O, Lseg, seg, pxj, pyj, nis, axj, ayj); break;
              default: obstacle_segment_multi_uniform<R, KA>(k, pts, c.O, Lseg, seg, pxj, pyj, nis, axj, ayj); break;
            }
          };
#if !defined(SFW_ABL_NOOBSLOOP)  // (ablation builds: time only, results wrong by construction)
          if (in_lds) {
            if (SFW_OBS_UNIFORM_LDS) run_uniform(pts_l);
            else run(pts_l);
          } else if (SFW_OBS_UNIFORM) run_uniform(pts_g);
          else run(pts_g);
#endif
          // The sixteen segment sums of an agent slot are added in segment order by one lane per (slot, component) out of LDS.  SL slots go through LDS at a time: two in a GPU-filling launch
          // (2 KB: what the wave can spare without losing a wave per SIMD at the target crowd), all four in a launch that leaves
          // the GPU under-filled (the ones that also keep the points in LDS) when the round has more than two — there a phase is
          // a latency chain (write, wait, sixteen loads, sixteen additions, the agent's update: 570 cycles, a seventh of such a
          // wave's step, profiles/r05_cycle_ablation.txt) and one phase does where two did.  Lane l < 8 SL: component l & 1 of agent slot
          // j0 + (l >> 1) % SL of lane group l / (2 SL).
          auto reduce = [&](auto sl_tag) {
            constexpr int SL = decltype(sl_tag)::value;
            // (the slots of a round go through LDS SL at a time: axj / opart are sized for KA <= 4 and a whole number of phases)
            static_assert(KA <= 4 && (SL == 2 || KA % SL == 0), "SFW_OBS_KA: at most four agent slots per lane");
#pragma unroll
            for (int j0 = 0; j0 < KA; j0 += SL) {
#if defined(SFW_ABL_NOREDUCE)
              asm volatile("" :: "v"(axj[j0]), "v"(ayj[j0]), "v"(axj[j0 + 1]), "v"(ayj[j0 + 1]));
              if (false) {
#else
              if (j0 < nj) {
#endif
#pragma unroll
                for (int u = 0; u < SL; ++u)
                  s.opart[u * WAVE + lane] = double2{static_cast<double>(axj[j0 + u]), static_cast<double>(ayj[j0 + u])};
                K2_SYNC();
                const int comp = lane & 1, js = (lane >> 1) & (SL - 1), gsub = lane / (2 * SL);  // gsub < 4 for lane < 8 SL
                const int a = a0 + gsub + OBS_AGENT_LANES * (j0 + js);
                if (lane < 8 * SL && a < A) {
                  const double *const col = part + 2 * (WAVE * js + gsub) + comp;  // segment q of that agent: + 2 * 4 * q
                  // all sixteen loads first, then the additions (left to the compiler: load two, wait, add two, eight LDS
                  // round trips one after the other)
                  double v[OBS_SEG];
#pragma unroll
                  for (int q = 0; q < OBS_SEG; ++q) v[q] = col[2 * OBS_AGENT_LANES * q];
                  double *const acc = (comp == 0 ? s.fcx : s.fcy) + a;  // (the robot's: not used)
                  const double sc = s.oscale[a], acc0 = *acc, wr0 = s.wr[0];
                  asm volatile("" ::: "memory");
                  R t = R(0);  // + 0 first, as obstacle_sums does; empty segments of a short scan add +0
#pragma unroll
                  for (int q = 0; q < OBS_SEG; ++q) t += static_cast<R>(v[q]);
                  const double f = static_cast<double>(t) * sc;
                  // the robot's Wr needs both components: lane 1 (y) hands its to lane 0 (x) — a swap inside the quad
                  const double f_other = swap_pair(f);
                  if (a != 0) *acc = fma(static_cast<double>(t), sc, acc0);
                  else if (comp == 0) s.swp[0] += wr0 + fast_norm(f, f_other);
                }
                K2_SYNC();
              }
            }
          };
          if (in_lds && nj > 2) reduce(std::integral_constant<int, (KA >= 4 ? 4 : 2)>{});
          else reduce(std::integral_constant<int, 2>{});
        }
      } else {
        // A short scan: the agent's lane runs the sixteen segments itself, the points through the scalar cache (the rounds
        // of the task loop cost more than they spread; sfw_derive prices both)
        for (int a = lane; a < A; a += WAVE) {
          const double rad = CONSTS_IN_LDS ? s.ac[a].radius : agent_c[a].radius;
          double tx, ty, sc;
          obstacle_sums<R, SFW_FLAT_PIPELINED>(k, c, obs_global(c.obstacles), s.px[a], s.py[a], rad, tx, ty, sc);
          if (a == 0) {
            s.swp[0] += s.wr[0] + fast_norm(tx * sc, ty * sc);
          } else {
            s.fcx[a] = fma(tx, sc, s.fcx[a]);
            s.fcy[a] = fma(ty, sc, s.fcy[a]);
          }
        }
        K2_SYNC();
      }
      if (lane == 0) {
        s.px[0] = rs.x;
        s.py[0] = rs.y;
        s.vx[0] = rs.vx;
        s.vy[0] = rs.vy;
      }
    }
    if (const double *const pin = La->pin_rest) {  // (pin_at_rest: no launch of a BASELINE workload comes here)
      if (pin_at_rest(pin, rs))
        for (int sl = lane_s; sl < A; sl += WAVE) {
          if (sl == 0) {
            s.fcx[0] += pin[2];
            s.fcy[0] += pin[3];
          } else {
            s.swp[sl] += pin[4 + sl];
          }
        }
    }
    K2_SYNC();
    if (s.dead[0] != 0) break;
    if constexpr (GROUPS) add_group_forces(k, c);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // a prefetched robot record may still be in flight
  const late_launch Le = late_args();
  if (Le->phase == SFW_PHASE_PREFIX) {  // leave the class record
    sfw_cls_agent *rec = Le->out_state + first_local * A;
    for (int sl = lane; sl < A; sl += WAVE) {
      sfw_cls_agent c;
      c.px = s.px[sl]; c.py = s.py[sl]; c.vx = s.vx[sl]; c.vy = s.vy[sl];
      c.fx = s.fcx[sl]; c.fy = s.fcy[sl]; c.sw = s.swp[sl];
      c.hasgoal = s.hasgoal8[sl];
      c.pad = 0;
      rec[sl] = c;
    }
    if (lane == 0) Le->out_dead[first_local] = s.dead[0];
    clock_probe(1);
    return;
  }
  double sw_acc = 0.0;
  for (int sl = lane; sl < A; sl += WAVE) sw_acc += s.swp[sl];
  if constexpr (CYCLE) {
    // finish_wave's reduction (the same 64-lane shuffle tree); the cost is formed by the wave that holds the costmap's verdict
    for (int off = WAVE / 2; off > 0; off >>= 1) sw_acc += __shfl_down(sw_acc, off, WAVE);
    if (lane == 0) {
      res->social_work = sw_acc;
      res->dead = s.dead[0];
    }
  } else {
    finish_wave(s, lane, 1, 1, first_local, sw_acc);
  }
  clock_probe(1);
}
#undef K2_SYNC
template <typename R, bool GROUPS, int CAP, bool OBS>
__global__ void __launch_bounds__(WAVE, (GROUPS || CAP == 0) ? 1 : OBS ? SFW_FLAT_WAVES : SFW_FLAT_WAVES_NOOBS) sfw_social_kernel_flat(const sfw_launch L, const int G_unused) {
  sfwm::fp_mode_for_omod();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  (void)G_unused;
  social_flat_wave<R, GROUPS, CAP, OBS, false>(L, smem, nullptr, nullptr, blockIdx.x, gridDim.x, L.item_base);
}
// Both forms in ONE launch (split_point: a register-form launch whose waves do not divide evenly over the SIMDs hands its last
// items to flat-form waves): the first n_reg blocks run the register form's body on the first `keep` items, the others the flat
// form's on the rest.  As two launches on two streams (rounds 4-5, still the way with laser points: the flat form with its
// laser-point pass needs 96 VGPRs, five waves per SIMD, and could not sit beside five register-form waves) the fork and the join
// cost ~12 us each on this runtime (profiles/r06_step_timeline_cfg2.txt) — of a 600 us step.  Same bodies: bit-identical.
template <typename R>
__global__ void __launch_bounds__(WAVE, 6) sfw_social_kernel_mixed(const sfw_launch L, const int G, const int n_reg, const int keep) {
  sfwm::fp_mode_for_omod();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (blockIdx.x < static_cast<unsigned>(n_reg)) {
    // (the register part scores items [0, keep): its item count is read from the launch, so it gets a copy that says so)
    sfw_launch Lr = L;
    if (Lr.phase == SFW_PHASE_PREFIX) Lr.n_cls = keep;
    else Lr.chunk_count = keep;
    social_reg_wave<R, 1, false>(Lr, G, smem, blockIdx.x, static_cast<unsigned>(n_reg));
  } else {
    social_flat_wave<R, false, 64, false, false>(L, smem, nullptr, nullptr, blockIdx.x - static_cast<unsigned>(n_reg),
                                                 gridDim.x - static_cast<unsigned>(n_reg), keep);
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
// one sample's cost against the running best of a thread (ref :394-404; see sel_less)
__device__ __forceinline__ void sel_consider(sfw_sel &best, double c, const double *linvels, const double *angvels, int nw,
                                             int64_t t, int64_t index_base) {
  if (!(c >= 0.0)) return;
  best.n_valid += 1;
  const double lin = linvels[t / nw], ang = angvels[t % nw];
  const bool selectable = c < 10000.0 || (c == 10000.0 && (lin > 0.0 || (lin == 0.0 && ang == 0.0)));
  if (!selectable) return;
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

// The cost vector's way to the host (sfw_host_mirror): stage 1 reads every cost exactly once, so it also writes it to the
// caller-visible pinned buffer `costs_host` (nullable), and the kernel that forms the final record leaves a copy at
// `sel_host`.  The blocking call then ends with the stream's completion instead of a device-to-host copy behind it (a blit
// kernel and two dependency gaps: 15 us of cfg2's step, 9 us of a control cycle; profiles/r06_step_timeline_cfg2.txt).
__global__ void __launch_bounds__(ARGMIN_BLOCK)
sfw_argmin_stage1(const double *costs, const double *linvels, const double *angvels, int nw, int64_t T,
                  int64_t index_base, sfw_sel *partials, double *costs_host, sfw_sel *sel_host) {
  sfw_sel best = sel_empty();
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < T;
       t += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const double c = costs[t];
    if (costs_host) costs_host[t] = c;
    sel_consider(best, c, linvels, angvels, nw, t, index_base);
  }
  best = block_reduce(best);
  if (threadIdx.x == 0) {
    partials[blockIdx.x] = best;
    if (sel_host) *sel_host = best;  // (single-block launch: the partial is the result)
  }
}
__global__ void __launch_bounds__(ARGMIN_BLOCK)
sfw_argmin_stage2(const sfw_sel *partials, int n, sfw_sel *out, sfw_sel *sel_host) {
  sfw_sel best = sel_empty();
  for (int i = threadIdx.x; i < n; i += blockDim.x) best = sel_merge(best, partials[i]);
  best = block_reduce(best);
  if (threadIdx.x == 0) {
    *out = best;
    if (sel_host) *sel_host = best;
  }
}

// Multi-device exchange record (sfw_multi_*): row `r` of an [R,5] table = this rank's selection key
// (cost, -linvel, |angvel|, -index) and its count of valid samples; every other row +inf, so that an
// element-wise all-reduce(min) over the ranks assembles the table of all local keys.
__global__ void __launch_bounds__(64) sfw_key_table_kernel(const sfw_sel *sel, double *table, int r, int R) {
  for (int e = threadIdx.x; e < 5 * R; e += blockDim.x) {
    double v = INFINITY;
    if (e / 5 == r) {
      const sfw_sel s = sel ? *sel : sel_empty();  // null: a rank without rows (+inf key, 0 valid samples)
      const int c = e % 5;
      if (c == 4) v = static_cast<double>(s.n_valid);
      else if (isfinite(s.cost)) v = c == 0 ? s.cost : c == 1 ? s.neg_linvel : c == 2 ? s.abs_angvel : static_cast<double>(s.neg_index);
    }
    table[e] = v;
  }
}

// ===========================================================================
// One launch per control cycle (grids of up to CYCLE_MAX_SAMPLES samples: the reference's own 5 x 9, :64-85)
// ===========================================================================
// K1 + K2 + K3 of such a grid were three launches back to back (14 + 56 + 5 us at 5 people and 40 steps,
// profiles/r05_cycle_timeline.txt), each a lone wave per sample waiting for the one before.  Here a sample is one block of
// four waves, each with a job of its own:
//   wave 0   the pedestrians (social_flat_wave<.., CYCLE>): stages the agents and runs the pair pass of the first step — none
//            of it needs the robot's trajectory — then the rollout, the robot's post-step records read from the LDS arrays
//            wave 1 fills (no K1->K2 table, no load in flight across a step).  It waits for a step's record by polling ONE
//            LDS word (cycle_result.ready), which wave 1 moves as the records come to stand;
//   wave 1   the robot's recurrences (k1s_velocities / _increments / _positions, a wave's worth of lanes): the first
//            CYCLE_HEAD steps first — published after ~1 us, so that wave 0 never stands still for the whole rollout — then
//            the rest;
//   1, 2, 3  once every pose stands: Trajectory points, the pedestrian-free cost terms, the footprint tasks (k1s_footprint)
//            — beside the pedestrian rollout;            | the block's one barrier behind all of it |
//   wave 1   (default FP mode) the in-order costmap scan, the cost = base + w_s x social work, and the selection: the last
//            block to finish — one atomic counter — reduces the cost vector under the reference's order (sel_consider) and
//            leaves vector and record in the host's pinned mirror too.
// A sample the costmap rejects is integrated all the same (its verdict comes from the waves beside it) and discarded.  The
// three-kernel path remains for everything else and as this kernel's checker (tests/test_cycle_kernel_gpu.py: bit-identical).
constexpr int CYCLE_BLOCK = 4 * WAVE;
constexpr int CYCLE_MAX_SAMPLES = 1024;
constexpr int CYCLE_HEAD = 8;  // robot steps handed to the pedestrians' wave ahead of the rest
template <typename R, bool GROUPS, bool OBS>
__global__ void __launch_bounds__(CYCLE_BLOCK) sfw_cycle_kernel(const sfw_launch L, const int k2_bytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // K2 wave's area (from LDS address 0) | k1s_lds | cycle_result
  const int tid = threadIdx.x, wave = tid / WAVE, lane = tid % WAVE;
  const int S = L.S;
  size_t k1_bytes;
  const k1s_lds a(smem + k2_bytes, S, &k1_bytes);
  cycle_result *const res = reinterpret_cast<cycle_result *>(smem + k2_bytes + k1_bytes);
  // the stage's arena straight from the host's pinned memory (sfw_launch.arena_host: no H2D copy was enqueued for it): 16
  // bytes per thread and round, one PCIe round trip for a control cycle's ~1 KB; every block writes the same bytes, and reads
  // them back behind the barrier below
  if (L.arena_host) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 *const src = reinterpret_cast<const u32x4 *>(L.arena_host);
    u32x4 *const dst = reinterpret_cast<u32x4 *>(L.arena_dev);
    for (uint32_t u = tid; u < L.arena_bytes / 16; u += CYCLE_BLOCK) dst[u] = __builtin_nontemporal_load(src + u);
    __threadfence();
  }
  // (the hand-over word starts at 0: written by thread 0, and a block barrier before anybody polls or publishes)
  if (tid == 0) res->ready = res->fdone = 0;
  __syncthreads();
  const k1s_sample q = k1s_sample_of(L, blockIdx.x);
  // pedestrians to integrate?  (no agents at all, or a robot alone without a laser point: social work identically 0)
  const bool social = q.scored && L.A > 0 && !(L.A == 1 && L.O == 0 && L.NG == 0);
  if (wave == 0 && social) {
    sfwm::fp_mode_for_omod();
    social_flat_wave<R, GROUPS, 64, OBS, true>(L, smem, &a, res, blockIdx.x, gridDim.x, 0);
  } else {
    if (wave == 1) {
      // the first CYCLE_HEAD steps first, published to the pedestrians' wave (res->ready) as soon as they stand; then the rest
      if (lane == 0) k1s_head(L, q);
      k1s_state st = k1s_begin(L, lane);
      for (int i0 = 0; i0 < S;) {
        const int i1 = i0 == 0 ? min(S, CYCLE_HEAD) : S;
        k1s_velocities(L, a, q, i0, i1, S, lane, st);
        wave_sync();
        k1s_increments(L, a, i0, i1, lane, WAVE);
        wave_sync();
        k1s_positions(L, a, i0, i1, lane, st);
        wave_sync();
        if (lane == 0) *const_cast<volatile int *>(&res->ready) = i1;  // (the LDS executes this wave's stores in order)
        i0 = i1;
      }
    } else {
      cycle_wait_ready(res, S);  // the footprint tasks need every pose
    }
    const int ftid = social ? tid - WAVE : tid, fn = social ? CYCLE_BLOCK - WAVE : CYCLE_BLOCK;
    k1s_records<false>(L, a, q, S, ftid, fn);
    if (ftid == fn - 1) res->base0 = k1s_base_cost_value(L, a, S);
    if (q.scored) {
      k1s_footprint(L, a, S, ftid, fn);
      // the LAST of these waves to finish scans the codes in step order (K1c) — beside the pedestrian rollout, not behind it
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      int before = 0;
      if (lane == 0) before = atomicAdd(&res->fdone, 1);
      before = __shfl(before, 0, WAVE);
      if (before == fn / WAVE - 1) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        k1s_quotients(a, S, lane, WAVE);
        wave_sync();
        if (lane == 0) {
          double base = 0.0;
          res->legal = k1s_scan(L, a, q, S, &base, &res->base0) ? 1 : 0;
          res->base = base;
        }
      }
    }
  }
  __syncthreads();  // every wave's part is done
  if (wave != 1) return;
  const int64_t t = q.t;
  if (lane == 0) {
    if (!q.scored) {
      if (L.n_points) L.n_points[q.local] = 0;
    } else {
      const double base = res->base;
      const bool legal = res->legal != 0;
      const int d = social ? res->dead : 0;
      if (legal && L.A > 0) {  // (no agent vector at all: scan_finish has written the cost)
        if (d == 0) {
          L.costs[t] = base + L.p.social_weight * (social ? res->social_work : 0.0);  // finish_wave / sfw_no_social_kernel
        } else {
          L.costs[t] = SFW_COST_INVALID;
          L.status[t] = SFW_ST_INVALID;
          if (L.coll_step) L.coll_step[t] = d - 2;
        }
      } else if (!legal && L.force_alive && d >= 2 && L.coll_step) {
        L.coll_step[t] = d - 2;  // point dumps: the contact in front of the illegal pose (finish_wave, force_alive)
      }
    }
  }
  // ---- selection by the last block to get here (a launch without a selection record — sfw_score_one's one sample — ends here)
  if (!L.sel_out) return;
  __threadfence();
  unsigned prev = 0;
  if (lane == 0) prev = atomicAdd(L.cycle_counter, 1u);
  prev = __shfl(prev, 0, WAVE);
  if (prev != gridDim.x - 1) return;
  __threadfence();
  sfw_sel best = sel_empty();
  for (int64_t i = lane; i < L.chunk_count; i += WAVE) {
    const int64_t ti = L.chunk_begin + i;
    const double c = __builtin_nontemporal_load(L.costs + ti);
    if (L.costs_host) L.costs_host[ti] = c;
    sel_consider(best, c, L.linvels, L.angvels, L.nw, ti, L.index_base);
  }
  for (int off = WAVE / 2; off > 0; off >>= 1) best = sel_merge(best, sel_shfl_down(best, off));
  if (lane == 0) {
    *L.sel_out = best;
    if (L.sel_host) *L.sel_host = best;
    *L.cycle_counter = 0u;  // for the next launch (stream order)
  }
}

}  // namespace

// ===========================================================================
// launchers
// ===========================================================================


// How one wave is organised for A agents.  Both organisations run at the vector-ALU issue rate on a GPU-filling grid, so
// the plan compares the VALU instructions they issue per sample and step (tools/k2_isa.sh: 90 per row and 225 in the
// per-agent pass of the register form, 85 per 64-pair iteration and 215 per 64-agent pass of the flat form):
//   reg  NS=1 : G = floor(64/A) samples per wave: (rows * 90 + 225) / G with rows = floor(A/2) half-ring rows
//   reg  NS=2 : one sample, two slots per lane (64 < A <= 128): rows * 180 + 450, times 0.9 (measured: it only wins at
//               A = 128, where the flat form's planes jump to the 208-double capacity)
//   flat      : one sample: (ceil(P/64) * 85 + ceil(A/64) * 215) * 1.03 with P = A (A-1) / 2 pairs (its four atomics per pair)
// Measured over the crowd size on a 128 x 128 grid (tools/form_crossover.py, profiles/r03_form_crossover.txt): the model
// names the faster form for every A tested: register form for A <= 21 (three or more samples
// per wave: 20-60 % faster), either for 22 <= A <= 32 (two samples per wave, within +-6 %), flat from A = 33 on (15-30 %
// faster than one sample on 52-100 % of the lanes).  Round 2's rule (lanes used >= 0.82 x the flat form's fill) chose
// the register form for 55 <= A <= 64 and 111 <= A <= 127, where it is 2-11 % slower.
//   few items (T <= 4096; 3072 for crowds of up to 12 agents, 1536 for up to 8): flat, one sample per wave — the GPU is
//   not full, so the shorter per-step critical path (P/64 iterations instead of A/2 rows) wins (A = 21: K2 0.10 ms vs
//   0.15 ms at 512 items).  Measured over the item count for A = 6, 11, 21, 31 with and without the shared prefix
//   (tools/form_vs_items.py, profiles/r03_form_crossover.txt): the flat form wins by 25-35 % up to 1024 items; small crowds
//   (many samples per register-form wave) turn early — A = 6 from 2048 items on (11 %, 41 % at 4096), A = 11 from 4096
//   (19 %) — while 21 and 31 agents stay within +-9 % of each other between 3072 and 5120 items.
// All organisations produce bit-identical costs (tools/kernel_equiv.py), so the choice
// never shows in the results.
struct wave_plan { int G; int ns; bool flat; };
// form: SFW_K2_AUTO, or SFW_K2_REGISTER / SFW_K2_FLAT forced by the caller (sfw_set_k2_form, SFW_FORCE_FLAT in the
// environment of sfw_create) — honoured wherever the form exists for A (register: 1 <= A <= 128; flat: A >= 2 or O > 0).
static wave_plan plan_for(int A, int64_t T, int O, int form, int cus) {
  if (A <= 0) return wave_plan{1, 1, false};
  const int P = A * (A - 1) / 2;
  const int rows = A / 2;
  const wave_plan flat{1, 0, true};
  const wave_plan reg = (A <= WAVE) ? wave_plan{WAVE / A < 32 ? WAVE / A : 32, 1, false} : wave_plan{1, 2, false};
  wave_plan best = flat;
  if (P == 0) {
    best = reg;  // a robot alone: nothing to flatten
  } else if (A <= 2 * WAVE) {
    const double c_flat = 1.03 * (85.0 * ((P + WAVE - 1) / WAVE) + 215.0 * ((A + WAVE - 1) / WAVE));
    const double c_reg = (A <= WAVE) ? (90.0 * rows + 225.0) / reg.G : 0.9 * (180.0 * rows + 450.0);
    if (c_reg < c_flat) best = reg;
  }
  // ... and a robot alone among laser points: the flat form spreads the points' sixteen segments over sixteen lanes
  // (the item thresholds were measured on 256 compute units and scale with the device: 6 / 12 / 16 items per CU)
#ifndef SFW_FLAT_ITEMS_PER_CU
#define SFW_FLAT_ITEMS_PER_CU 16
#endif
  if (T <= static_cast<int64_t>(A <= 8 ? 6 : A <= 12 ? 12 : SFW_FLAT_ITEMS_PER_CU) * cus && (A >= 2 || O > 0)) best = flat;
  if (form == SFW_K2_FLAT && (A >= 2 || O > 0)) best = flat;
  if (form == SFW_K2_REGISTER && A <= 2 * WAVE) best = reg;
  return best;
}

int sfw_samples_per_wave(int A, int64_t T, int form, int cus) { return plan_for(A, T, 0, form, cus).G; }

// SFW_ORG_* of the launch plan_for() picks for T items
int sfw_social_organisation(int A, int64_t T, int O, int form, int cus) {
  if (A <= 0) return SFW_ORG_NONE;
  const wave_plan pl = plan_for(A, T, O, form, cus);
  return pl.flat ? SFW_ORG_FLAT : pl.ns == 2 ? SFW_ORG_REGISTER_2 : SFW_ORG_REGISTER_1;
}

void sfw_derive(sfw_launch &L) {
  const sfw_params &p = L.p;
  sfw_force_k<double> &d = L.k.d;
  d.lambda = p.sfm_lambda;
  const double l2e = 1.4426950408889634074;  // log2(e): every exponent argument is handed over in log2 units (sfw_math.h)
  d.neg_l2e_inv_gamma = -l2e / p.sfm_gamma;
  d.l2_f_social = std::log2(p.sfm_force_factor_social);  // -inf for Fs = 0: the clamp at -1100 makes the force 0
  d.c_vel = -(p.sfm_n_prime * p.sfm_n_prime) * (p.sfm_gamma * p.sfm_gamma) * l2e;
  d.c_ang = -(p.sfm_n * p.sfm_n) * (p.sfm_gamma * p.sfm_gamma) * l2e;
  // log2 of the obstacle force factor (it multiplies the agent's sum as 2^(radius log2(e)/sigma + log2 k)); a factor of 0
  // becomes 2^-1000 = 1e-301
  d.l2_f_obstacle = p.sfm_force_factor_obstacle > 0 ? std::log2(p.sfm_force_factor_obstacle) : -1000.0;
  d.l2e_inv_sigma = l2e / p.sfm_force_sigma_obstacle;
  sfw_force_k<float> &f = L.k.f;
  f.lambda = static_cast<float>(d.lambda);
  f.neg_l2e_inv_gamma = static_cast<float>(d.neg_l2e_inv_gamma);
  f.l2_f_social = static_cast<float>(d.l2_f_social);
  f.c_vel = static_cast<float>(d.c_vel);
  f.c_ang = static_cast<float>(d.c_ang);
  f.l2_f_obstacle = static_cast<float>(d.l2_f_obstacle);
  f.l2e_inv_sigma = static_cast<float>(d.l2e_inv_sigma);
  L.k.f_desired = p.sfm_force_factor_desired;
  L.k.inv_tau = 1.0 / p.sfm_relaxation_time;
  L.k.rr = static_cast<double>(static_cast<float>(p.robot_radius) * static_cast<float>(p.robot_radius));  // ref :617
  L.k.inv_O = L.O > 0 ? 1.0 / L.O : 0.0;
  // Flat form, laser-point pass: (agent, segment) tasks over all 64 lanes, or one lane per agent?  Issue slots per step
  // (27 per evaluation; a round of the task loop also costs its two reduction phases, ~40 slots each):
  {
    const int A = L.A, O = L.O, Lseg = (O + OBS_SEG - 1) / OBS_SEG;
    const int full = A / (OBS_AGENT_LANES * OBS_AGENTS_PER_LANE), rest = (A % (OBS_AGENT_LANES * OBS_AGENTS_PER_LANE) + OBS_AGENT_LANES - 1) / OBS_AGENT_LANES;
    const double red = 40.0;
    const double c_tasks = full * (27.0 * Lseg * OBS_AGENTS_PER_LANE + 2 * red) + (rest ? 27.0 * Lseg * rest + ((rest + 1) / 2) * red : 0.0);
    const double c_lane = 27.0 * O * ((A + WAVE - 1) / WAVE);
    L.k.obs_tasks = (O > 0 && c_tasks < c_lane) ? 1 : 0;
  }
}

// Capacity (doubles per LDS plane) of the flat kernel for A agents: compile-time 64 / 128 / 256 when A fits with
// one record to spare (the dummy slot of the padded pair table), otherwise 0 = run-time (A + 1 rounded up to even).
// 208: crowds of 128..207 agents (BASELINE cfg4: 201) keep their plane distances as immediates AND fit ten waves per CU
// (8 planes x 208 x 8 B + 2.4 KB < 15 KB; with 256-double planes 19 KB: eight)
static int flat_cap(int A) { return A < 64 ? 64 : A < 104 ? 104 : A < 128 ? 128 : A < 208 ? 208 : A < 256 ? 256 : 0; }
static int flat_cap_runtime(int A) { return (A + 2) & ~1; }

// Flat form: does a launch of `items` waves keep its own LDS copy of the laser points?  Yes while it leaves the GPU
// under-filled (at most two waves per SIMD: LDS is plentiful and every load's latency is exposed); a GPU-filling launch
// reads the points through the L1 instead (the copy would cost it occupancy).
static size_t lds_bytes_for(const wave_plan &pl, int A, int O, int NG, int n_grp_mem, bool obs_lds);
// (... and only while the wave's whole allocation stays within 64 KB: the copy is an optimisation and must never be what
// makes an agent set exceed the LDS of a compute unit)
static bool obs_in_lds(const wave_plan &pl, int A, int O, int NG, int n_grp_mem, int64_t items, int cus) {
  return pl.flat && O > 0 && items <= static_cast<int64_t>(8) * cus && lds_bytes_for(pl, A, O, NG, n_grp_mem, true) <= 64 * 1024;
}
static size_t lds_bytes_for(const wave_plan &pl, int A, int O, int NG, int n_grp_mem, bool obs_lds) {
  if (pl.flat) {
    const int c = flat_cap(A);
    return lds_layout(nullptr, A, c > 0 ? c : flat_cap_runtime(A), A, 1, O, NG, n_grp_mem, NG > 0, true, obs_lds).bytes;
  }
  (void)obs_lds;
  return lds_layout(nullptr, A, WAVE * pl.ns, pl.G * A, pl.G, O, NG, n_grp_mem, true, false).bytes;
}

// Largest LDS allocation any launch of a chunk of T samples may ask for (the prefix phase of the
// shared-prefix rollout may pick the flat organisation where the chunk itself uses the other).
size_t sfw_social_lds_bytes(int A, int O, int NG, int n_grp_mem, int64_t T, int form, int cus) {
  const wave_plan pl = plan_for(A, T, O, form, cus);
  const size_t a = lds_bytes_for(pl, A, O, NG, n_grp_mem, obs_in_lds(pl, A, O, NG, n_grp_mem, T, cus));
  // (a prefix level may hold any number of classes up to T: the flat form — with the points' copy where that is allowed — is
  // the largest it can ask for)
  const wave_plan fl{1, 0, true};
  const size_t b = (A >= 2 || O > 0) ? lds_bytes_for(fl, A, O, NG, n_grp_mem, obs_in_lds(fl, A, O, NG, n_grp_mem, 1, cus)) : 0;
  return a > b ? a : b;
}

// uint16 entries of the pair table: two arrays (i offsets, j offsets) of ceil(P / 64) * 64 entries each
int64_t sfw_pair_table_entries(int A) {
  const int64_t P = static_cast<int64_t>(A) * (A - 1) / 2;
  const int64_t n = (P + WAVE - 1) / WAVE * WAVE;
  return 2 * (n > 0 ? n : WAVE);
}

hipError_t sfw_launch_pair_table(uint16_t *tab, int A, hipStream_t stream) {
  const int n = static_cast<int>(sfw_pair_table_entries(A) / 2);
  const int cap = flat_cap(A) > 0 ? flat_cap(A) : flat_cap_runtime(A);  // the capacity the flat kernel will run with
  hipLaunchKernelGGL(sfw_pair_table_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, tab, A, n, cap - A);
  return hipGetLastError();
}

// SFW_K1A_THREADS=1 in the environment (read at every launch: tests flip it): K1a as one thread per sample
static bool sfw_k1a_threads() {
  const char *e = std::getenv("SFW_K1A_THREADS");
  return e && e[0] == '1';
}

bool sfw_rollout_is_fused(const sfw_launch &L) { return L.chunk_count <= 2048 && L.S <= K1_SMALL_MAX_STEPS; }

// K1a alone (pose integration + robot-step table), or the fused small-grid K1.
hipError_t sfw_launch_rollout_poses(const sfw_launch &L, hipStream_t stream) {
  if (L.chunk_count <= 0) return hipSuccess;
  if (sfw_rollout_is_fused(L)) {  // latency path: one launch does all of K1
    hipLaunchKernelGGL(sfw_rollout_small_kernel, dim3(static_cast<unsigned>(L.chunk_count)), dim3(K1_SMALL_BLOCK), 0, stream, L);
    return hipGetLastError();
  }
  const int block = 64;  // latency-bound serial rollout: spread the waves over all CUs
  if (sfw_k1a_threads()) {  // SFW_K1A_THREADS=1: round 1-5's one thread per sample (A/B and the equality test)
    const unsigned grid = static_cast<unsigned>((L.chunk_count + block - 1) / block);
    hipLaunchKernelGGL(sfw_rollout_kernel, dim3(grid), dim3(block), 0, stream, L);
    return hipGetLastError();
  }
  constexpr int per_wave = WAVE / K1A_TEAM;
  const unsigned grid = static_cast<unsigned>((L.chunk_count + per_wave - 1) / per_wave);
  hipLaunchKernelGGL(sfw_rollout_team_kernel, dim3(grid), dim3(block), 0, stream, L);
  return hipGetLastError();
}

// K1b + K1c (footprint checks, costmap scan); nothing to do after the fused K1.
hipError_t sfw_launch_rollout_costmap(const sfw_launch &L, hipStream_t stream) {
  if (L.chunk_count <= 0 || sfw_rollout_is_fused(L)) return hipSuccess;
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

hipError_t sfw_launch_rollout(const sfw_launch &L, hipStream_t stream) {
  hipError_t e = sfw_launch_rollout_poses(L, stream);
  return e != hipSuccess ? e : sfw_launch_rollout_costmap(L, stream);
}

template <typename K> static hipError_t launch_social_as(K kernel, const sfw_launch &L, int G, unsigned grid,
                                                          size_t lds, hipStream_t stream) {
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(WAVE), lds, stream, L, G);
  return hipGetLastError();
}

// reg_off<CAP> restates where lds_layout puts the register form's fixed part: checked once per process.
template <int CAP> static bool reg_layout_matches() {
  const lds_layout s(nullptr, 5, CAP, 10, 2, 3, 0, 0, true, false);
  auto at = [&](const void *p) { return static_cast<int>(reinterpret_cast<const char *>(p) - reinterpret_cast<const char *>(s.px)); };
  using off = reg_off<CAP>;
  return at(s.py) == off::PY && at(s.vx) == off::VX && at(s.vy) == off::VY && at(s.fjx) == off::FJX && at(s.fjy) == off::FJY &&
         at(s.hasgoal) == off::HG && at(s.swp) == off::SW && at(s.dead) == off::DEAD && at(s.rsb) == off::RSB && s.hg_stride == 2;
}

// A register-form launch of W waves on S SIMDs keeps all its waves resident at once (up to six per SIMD), q = floor(W / S) on
// every SIMD and one more on W mod S of them — and those set the duration: BASELINE cfg2's 5462 waves are 5.33 per SIMD and
// take the time of 6 (VALU-active 75 %).  The organisations are bit-identical, so the launch may hand its last items to
// flat-form waves — one sample each, about half the work of a register-form wave of three — that run beside exactly q
// register-form waves per SIMD: cfg2's 16 384 samples = 5 x 1024 x 3 in the register form + 1024 flat waves, one per SIMD.
// Returns how many of the `items` stay in the register form (all of them when a split would not pay; the costs are plan_for's
// instruction counts per sample and step, plus 24 per laser-point evaluation).
#ifndef SFW_SPLIT_FORMS
#define SFW_SPLIT_FORMS 1
#endif
#ifndef SFW_MIXED_LAUNCH
#define SFW_MIXED_LAUNCH 1  // 0: the split launch as two launches on two streams whatever the scan (rounds 4-5; A/B)
#endif
static int64_t split_point(const wave_plan &pl, int A, int O, int NG, int form, int64_t items, int cus) {
  if (!SFW_SPLIT_FORMS || form != SFW_K2_AUTO || pl.flat || pl.ns != 1 || A < 2 || A > WAVE || NG > 0) return items;
  const int64_t S = 4LL * cus, W = (items + pl.G - 1) / pl.G, q = W / S;
  if (q < 1 || q > 5 || W % S == 0) return items;  // (six waves per SIMD are resident at once; beyond that the waves queue)
  const int64_t reg_items = q * S * pl.G, rest = items - reg_items;
  const int P = A * (A - 1) / 2, Lseg = (O + OBS_SEG - 1) / OBS_SEG;
  const double c_reg_wave = 90.0 * (A / 2) + 225.0 + 24.0 * O;
  const double c_flat_wave = 1.03 * (85.0 * ((P + WAVE - 1) / WAVE) + 215.0) + 24.7 * ((A + OBS_AGENT_LANES - 1) / OBS_AGENT_LANES) * Lseg;
  const double t_now = static_cast<double>(q + 1), t_split = q + static_cast<double>((rest + S - 1) / S) * c_flat_wave / c_reg_wave;
  return t_split < 0.97 * t_now ? reg_items : items;
}

// how many of the T samples of a launch the register form hands to flat-form waves (0: none; sfw_grid_plan_info)
int64_t sfw_social_flat_items(int A, int O, int NG, int64_t T, int form, int cus) {
  if (A <= 0 || T <= 0) return 0;
  return T - split_point(plan_for(A, T, O, form, cus), A, O, NG, form, T, cus);
}

template <typename R> static hipError_t launch_social_typed(const sfw_launch &L_in, hipStream_t stream, const sfw_split_streams *sp = nullptr) {
  // The organisations are bit-identical and the class records of the shared-prefix rollout are
  // organisation-neutral, so every launch picks its own by its item count (measured: forcing the
  // flat form on an under-filled prefix phase at cfg2 — 2024 register-form waves — changes nothing).
  const int64_t all_items = L_in.phase == SFW_PHASE_PREFIX ? static_cast<int64_t>(L_in.n_cls) : L_in.chunk_count;
  const int64_t item_base = L_in.item_base, items = all_items - item_base;  // (item_base > 0: the flat part of a split launch)
  const int cus = L_in.n_cu > 0 ? L_in.n_cu : SFW_DEFAULT_CUS;
  const wave_plan pl = plan_for(L_in.A, items, L_in.O, L_in.k2_form, cus);
  if (sp && sp->side && item_base == 0) {
    const int64_t keep = L_in.pair_tab ? split_point(pl, L_in.A, L_in.O, L_in.NG, L_in.k2_form, items, cus) : items;
    if (keep < items && L_in.O == 0 && flat_cap(L_in.A) == 64 && SFW_MIXED_LAUNCH) {
      // both forms in one launch (sfw_social_kernel_mixed): no second stream, no fork, no join
      const unsigned n_reg = static_cast<unsigned>((keep + pl.G - 1) / pl.G), n_flat = static_cast<unsigned>(items - keep);
      const wave_plan fl{1, 0, true};
      const size_t lds = std::max(lds_bytes_for(pl, L_in.A, 0, 0, 0, false), lds_bytes_for(fl, L_in.A, 0, 0, 0, false));
      sfw_launch L = L_in;
      L.k.obs_lds = 0;
      hipLaunchKernelGGL((sfw_social_kernel_mixed<R>), dim3(n_reg + n_flat), dim3(WAVE), lds, stream, L, pl.G, static_cast<int>(n_reg),
                         static_cast<int>(keep));
      return hipGetLastError();
    }
    if (keep < items) {
      // the register-form part first (its waves take their q places per SIMD), the flat part beside it on the other stream
      // (the register-form kernel is left as it is — a bound of its own cost its strict build 36 bytes of scratch —: its part
      // is a launch over the first `keep` items; the flat kernel takes the first item of its part from item_base)
      sfw_launch La = L_in, Lb = L_in;
      if (La.phase == SFW_PHASE_PREFIX) La.n_cls = static_cast<int32_t>(keep);
      else La.chunk_count = keep;
      La.k2_form = SFW_K2_REGISTER;
      Lb.item_base = static_cast<int32_t>(keep);
      Lb.k2_form = SFW_K2_FLAT;
      Lb.clock_probe = nullptr;  // (one launch writes the clock probe)
      hipError_t e = hipEventRecord(sp->fork, stream);
      if (e == hipSuccess) e = hipStreamWaitEvent(sp->side, sp->fork, 0);
      if (e == hipSuccess) e = launch_social_typed<R>(La, stream);
      if (e == hipSuccess) e = launch_social_typed<R>(Lb, sp->side);
      if (e == hipSuccess) e = hipEventRecord(sp->join, sp->side);
      if (e == hipSuccess) e = hipStreamWaitEvent(stream, sp->join, 0);
      return e;
    }
  }
  const unsigned grid = static_cast<unsigned>((items + pl.G - 1) / pl.G);
  sfw_launch L = L_in;
  // (the flat part of a split launch runs beside a full GPU: no LDS copy of the points, whatever its own item count)
  L.k.obs_lds = (item_base == 0 && obs_in_lds(pl, L.A, L.O, L.NG, L.n_grp_mem, items, cus)) ? 1 : 0;
  const size_t lds = lds_bytes_for(pl, L.A, L.O, L.NG, L.n_grp_mem, L.k.obs_lds != 0);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  static const bool layout_ok = reg_layout_matches<WAVE>() && reg_layout_matches<2 * WAVE>();
  if (!layout_ok) return hipErrorInvalidValue;
  const bool groups = L.NG > 0;  // at least one agent carries a group id: kernels with the group pass
  // the register form brings a step's robot records in with ONE lane < 2 G load and keeps REG_DEAD_CAP contact flags
  if (!pl.flat && pl.G > lds_layout::REG_DEAD_CAP) return hipErrorInvalidValue;
  if (pl.flat) {
    if (8 * (static_cast<int64_t>(L.A) + 1) > 65535 || !L.pair_tab) return hipErrorInvalidValue;  // 16-bit plane offsets
    const bool obs = L.O > 0;  // the kernels with the laser-point pass (sfw_social_kernel_flat<.., OBS>)
    switch (flat_cap(L.A)) {
      case 64:
        return groups ? launch_social_as(sfw_social_kernel_flat<R, true, 64, true>, L, 1, grid, lds, stream)
               : obs  ? launch_social_as(sfw_social_kernel_flat<R, false, 64, true>, L, 1, grid, lds, stream)
                      : launch_social_as(sfw_social_kernel_flat<R, false, 64, false>, L, 1, grid, lds, stream);
      case 104:
        return groups ? launch_social_as(sfw_social_kernel_flat<R, true, 104, true>, L, 1, grid, lds, stream)
               : obs  ? launch_social_as(sfw_social_kernel_flat<R, false, 104, true>, L, 1, grid, lds, stream)
                      : launch_social_as(sfw_social_kernel_flat<R, false, 104, false>, L, 1, grid, lds, stream);
      case 128:
        return groups ? launch_social_as(sfw_social_kernel_flat<R, true, 128, true>, L, 1, grid, lds, stream)
               : obs  ? launch_social_as(sfw_social_kernel_flat<R, false, 128, true>, L, 1, grid, lds, stream)
                      : launch_social_as(sfw_social_kernel_flat<R, false, 128, false>, L, 1, grid, lds, stream);
      case 208:
        return groups ? launch_social_as(sfw_social_kernel_flat<R, true, 208, true>, L, 1, grid, lds, stream)
               : obs  ? launch_social_as(sfw_social_kernel_flat<R, false, 208, true>, L, 1, grid, lds, stream)
                      : launch_social_as(sfw_social_kernel_flat<R, false, 208, false>, L, 1, grid, lds, stream);
      case 256:
        return groups ? launch_social_as(sfw_social_kernel_flat<R, true, 256, true>, L, 1, grid, lds, stream)
               : obs  ? launch_social_as(sfw_social_kernel_flat<R, false, 256, true>, L, 1, grid, lds, stream)
                      : launch_social_as(sfw_social_kernel_flat<R, false, 256, false>, L, 1, grid, lds, stream);
      default:
        return groups ? launch_social_as(sfw_social_kernel_flat<R, true, 0, true>, L, 1, grid, lds, stream)
               : obs  ? launch_social_as(sfw_social_kernel_flat<R, false, 0, true>, L, 1, grid, lds, stream)
                      : launch_social_as(sfw_social_kernel_flat<R, false, 0, false>, L, 1, grid, lds, stream);
    }
  }
  if (groups) {
    if (pl.ns == 1) return launch_social_as(sfw_social_kernel<R, 1, true>, L, pl.G, grid, lds, stream);
    return launch_social_as(sfw_social_kernel<R, 2, true>, L, pl.G, grid, lds, stream);
  }
  if (pl.ns == 1) return launch_social_as(sfw_social_kernel<R, 1, false>, L, pl.G, grid, lds, stream);
  return launch_social_as(sfw_social_kernel<R, 2, false>, L, pl.G, grid, lds, stream);
}

// A robot alone without a laser point has no social work to integrate: no pair, no obstacle force, so Wr = |0| and Wp = 0
// at every step, nobody to run into, and K2 would walk its S dependent steps to add social_weight x 0 to the base cost
// (65 us of a 118 us control cycle; BASELINE cfg1).  This writes what finish_wave would have written.
namespace {
__global__ void __launch_bounds__(256) sfw_no_social_kernel(const sfw_launch L) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= L.chunk_count) return;
  const int64_t t = L.chunk_begin + i;
  if (L.status[t] == SFW_ST_VALID) L.costs[t] = L.base_cost[t] + L.p.social_weight * 0.0;
}
}  // namespace

hipError_t sfw_launch_social(const sfw_launch &L, hipStream_t stream, const sfw_split_streams *sp) {
  if (L.chunk_count <= 0 || L.A <= 0) return hipSuccess;
  if (SFW_SKIP_EMPTY_K2 && L.A == 1 && L.O == 0 && L.NG == 0 && L.phase != SFW_PHASE_PREFIX && !L.out_state) {
    hipLaunchKernelGGL(sfw_no_social_kernel, dim3(static_cast<unsigned>((L.chunk_count + 255) / 256)), dim3(256), 0, stream, L);
    return hipGetLastError();
  }
#ifndef SFW_STRICT_BUILD  // (the strict build of this file holds the f64 kernels only)
  if (L.p.precision == SFW_PRECISION_F32) return launch_social_typed<float>(L, stream, sp);
#endif
  return launch_social_typed<double>(L, stream, sp);
}

hipError_t sfw_launch_key_table(const sfw_sel *sel, double *table, int r, int R, hipStream_t stream) {
  hipLaunchKernelGGL(sfw_key_table_kernel, dim3(1), dim3(64), 0, stream, sel, table, r, R);
  return hipGetLastError();
}

int64_t sfw_argmin_partials(int64_t T) {
  if (T <= 16 * ARGMIN_BLOCK) return 1;  // one block, one launch (latency path)
  int64_t blocks = (T + ARGMIN_BLOCK - 1) / ARGMIN_BLOCK;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  return blocks;
}

hipError_t sfw_launch_argmin(const double *costs, const double *linvels, const double *angvels, int32_t nw,
                             int64_t T, int64_t index_base, sfw_sel *partials, sfw_sel *out,
                             hipStream_t stream, double *costs_host, sfw_sel *sel_host) {
  const int blocks = static_cast<int>(sfw_argmin_partials(T));
  if (blocks == 1) {  // the single block's partial is the result
    hipLaunchKernelGGL(sfw_argmin_stage1, dim3(1), dim3(ARGMIN_BLOCK), 0, stream, costs, linvels, angvels, nw, T,
                       index_base, out, costs_host, sel_host);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(sfw_argmin_stage1, dim3(blocks), dim3(ARGMIN_BLOCK), 0, stream, costs, linvels,
                     angvels, nw, T, index_base, partials, costs_host, static_cast<sfw_sel *>(nullptr));
  hipLaunchKernelGGL(sfw_argmin_stage2, dim3(1), dim3(ARGMIN_BLOCK), 0, stream, partials, blocks, out, sel_host);
  return hipGetLastError();
}

// ---- one launch per control cycle ------------------------------------------------------------------------------------
static size_t cycle_k2_bytes(const sfw_launch &L, bool obs_lds) {
  const bool social = L.A > 0 && !(L.A == 1 && L.O == 0 && L.NG == 0);
  if (!social) return 0;
  // the flat form's layout on 64-double planes, with the agents' constants staged (social_flat_wave<.., CYCLE>)
  return (lds_layout(nullptr, L.A, 64, L.A, 1, L.O, L.NG, L.n_grp_mem, true, true, obs_lds).bytes + 15) & ~size_t(15);
}
// Does sfw_launch_cycle take this launch?  A whole (unshared, single-chunk) rollout of a small grid whose K2 would run the
// flat form on 64-double planes, or has no pedestrians to integrate.  SFW_CYCLE_FUSED=0 in the environment (read at every
// launch: tests flip it): never.
bool sfw_cycle_applies(const sfw_launch &L) {
  if (const char *e = std::getenv("SFW_CYCLE_FUSED"))
    if (e[0] == '0') return false;
  if (L.phase != SFW_PHASE_WHOLE || L.resume || L.chunk_begin != 0 || L.chunk_count <= 0 || L.chunk_count > CYCLE_MAX_SAMPLES) return false;
  if (L.S > K1_SMALL_MAX_STEPS || (L.sel_out && !L.cycle_counter)) return false;
  const bool social = L.A > 0 && !(L.A == 1 && L.O == 0 && L.NG == 0);
  if (social) {
    const int cus = L.n_cu > 0 ? L.n_cu : SFW_DEFAULT_CUS;
    if (flat_cap(L.A) != 64 || !L.pair_tab || !plan_for(L.A, L.chunk_count, L.O, L.k2_form, cus).flat) return false;
  }
  const wave_plan fl{1, 0, true};
  const bool obs_lds = social && obs_in_lds(fl, L.A, L.O, L.NG, L.n_grp_mem, L.chunk_count, L.n_cu > 0 ? L.n_cu : SFW_DEFAULT_CUS);
  size_t k1_bytes = 0;
  (void)k1s_lds(nullptr, L.S, &k1_bytes);
  return cycle_k2_bytes(L, obs_lds) + k1_bytes + ((sizeof(cycle_result) + 15) & ~size_t(15)) <= 64 * 1024;
}
template <typename R> static hipError_t launch_cycle_typed(const sfw_launch &L_in, hipStream_t stream) {
  sfw_launch L = L_in;
  const bool social = L.A > 0 && !(L.A == 1 && L.O == 0 && L.NG == 0);
  const wave_plan fl{1, 0, true};
  L.k.obs_lds = (social && obs_in_lds(fl, L.A, L.O, L.NG, L.n_grp_mem, L.chunk_count, L.n_cu > 0 ? L.n_cu : SFW_DEFAULT_CUS)) ? 1 : 0;
  const size_t k2 = cycle_k2_bytes(L, L.k.obs_lds != 0);
  size_t k1_bytes = 0;
  (void)k1s_lds(nullptr, L.S, &k1_bytes);
  const size_t lds = k2 + k1_bytes + ((sizeof(cycle_result) + 15) & ~size_t(15));
  const dim3 grid(static_cast<unsigned>(L.chunk_count)), block(CYCLE_BLOCK);
  const int k2i = static_cast<int>(k2);
  if (L.NG > 0) hipLaunchKernelGGL((sfw_cycle_kernel<R, true, true>), grid, block, lds, stream, L, k2i);
  else if (L.O > 0) hipLaunchKernelGGL((sfw_cycle_kernel<R, false, true>), grid, block, lds, stream, L, k2i);
  else hipLaunchKernelGGL((sfw_cycle_kernel<R, false, false>), grid, block, lds, stream, L, k2i);
  return hipGetLastError();
}
hipError_t sfw_launch_cycle(const sfw_launch &L, hipStream_t stream) {
#ifndef SFW_STRICT_BUILD
  if (L.p.precision == SFW_PRECISION_F32) return launch_cycle_typed<float>(L, stream);
#endif
  return launch_cycle_typed<double>(L, stream);
}
