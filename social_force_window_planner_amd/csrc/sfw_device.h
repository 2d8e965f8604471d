// sfw_device.h — internal contract between the C-ABI host code (sfw_capi.hip)
// and the gfx950 kernels (sfw_kernels.hip).  Not part of the public ABI.
#ifndef SFW_DEVICE_H_
#define SFW_DEVICE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sfw_hip.h"

// 0 (default): the pair term's angular part is exactly 0 for a pair whose w x diff is exactly 0 — sign(theta) = 0, as in
// lightsfm for theta == 0 — at every step (a compare and one integer select per pair evaluation, exp_fast2_gated); the host
// evaluates the reference's rounding-noise term for such pairs of the handed-over state (rest_forces).
// 1: round 3's form, kept for A/B: the sign BIT of w x diff decides also for a zero (one v_bfi_b32, 2 issue slots less per
// pair) and rest_forces takes the kernels' own term back out — right for the handed-over state only: an alignment that
// persists (a robot driving straight at a person on its axis) then gets a full-magnitude lateral force from step 1 on.
#ifndef SFW_SIGN_OF_ZERO
#define SFW_SIGN_OF_ZERO 0
#endif

// Ablation builds (make EXTRA=-DSFW_ABL_...: a part of a kernel taken out to price it) produce results that are wrong by
// construction.  Any of those macros marks the whole library: sfw_create refuses to make a handle unless SFW_ALLOW_ABLATION=1
// is in the environment (the timing scripts under tools/ set it; no test and no bench run does).
#if defined(SFW_ABL_NOROBOT) || defined(SFW_ABL_NOREAD) || defined(SFW_ABL_NOMATH) || defined(SFW_ABL_NOATOM) ||        \
    defined(SFW_ABL_NOAGENT) || defined(SFW_ABL_KEEPALIVE) || defined(SFW_ABL_HALF_LDS) || defined(SFW_ABL_NOOBSLOOP) || \
    defined(SFW_ABL_NOREDUCE)
#define SFW_ABLATION_BUILD 1
#endif

// Device shape the launch heuristics default to (a whole MI355X: 256 compute units in 8 XCDs of 32); a handle reads its
// device's own (hipDeviceProp_t.multiProcessorCount) and carries it into every launch (sfw_launch.n_cu / n_xcd).
#define SFW_DEFAULT_CUS 256
#define SFW_CUS_PER_XCD 32

// Per-sample status written by the rollout kernel.
enum : int32_t { SFW_ST_VALID = 0, SFW_ST_INVALID = 1, SFW_ST_SKIPPED = 2 };

// Post-step robot agent state for one (step, sample): what the reference
// writes into myagents[0] at src/sfw_planner.cpp:600-604 (position and the
// robot-LOCAL velocity).  32 bytes: the form a K2 wave holds it in (LDS).
struct __attribute__((aligned(32))) sfw_robot_step {
  double x, y, vx, vy;
};

// Pre-step footprint frame of one (step, sample): position and cos/sin(theta).
struct __attribute__((aligned(32))) sfw_pose_frame {
  double x, y, c, s;
};

// The K1 -> K2 / K1b tables in memory (round 6; rounds 1-5 stored the two 32-byte records above per (step, sample): 64 B).
// Of a robot step only the POSITION depends on both sample axes: the velocity sequence is the linear sample's alone
// (computeNewVelocity, ref :581-583), the heading — hence cos / sin — the angular sample's (ref :588), and the pose BEFORE
// step i is the position AFTER step i - 1.  So per step one row of 16-byte units
//     [ position after the step, one unit (x, y) per sample of the chunk | velocity (vx, vy) of the step, one unit per grid row ]
// (sfw_launch.ptab, row_units units per step) and one row (cos, sin) per grid column (sfw_launch.cs_tab): 16 B per (step,
// sample) written by the pose rollout and read back by K1b and K2 — a quarter of the traffic that bounded the rollout
// (cfg2: 42 MB in 16 us) — and a K2 lane still fetches its half of a record by ONE 16-byte load from a row base plus an
// offset that is fixed for the rollout.
struct __attribute__((aligned(16))) sfw_unit {
  double a, b;
};

// Person constants shared by every sample (device copy, SoA-friendly AoS).
struct __attribute__((aligned(16))) sfw_agent_const {
  double goal_x, goal_y;
  double goal_radius, desired_velocity;
  double radius;
  int32_t id, has_goal;
};

// Social-force constants derived from sfw_params on the host (sfw_derive), in the type the
// forces are evaluated in.  They reach the kernels as kernel arguments, i.e. in SGPRs: a value
// derived on the device (a division, a product) is a VALU result and would sit in a VGPR pair
// of every lane for the whole rollout.
template <typename R> struct sfw_force_k {
  // Everything that ends up in an exponent is in LOG2 units (the device evaluates 2^x, sfw_math.h): -log2(e)/gamma,
  // log2(Fs), c_vel = -(n' gamma)^2 log2(e), c_ang = -(n gamma)^2 log2(e), log2(k_obstacle), log2(e)/sigma
  R lambda, neg_l2e_inv_gamma, l2_f_social, c_vel, c_ang, l2_f_obstacle, l2e_inv_sigma;
};
struct sfw_derived {
  sfw_force_k<double> d;
  sfw_force_k<float> f;
  double f_desired, inv_tau, rr, inv_O;
  int32_t obs_tasks;  // flat K2, laser-point pass: 1 = (agent, segment) tasks over all lanes, 0 = one lane per agent
  int32_t obs_lds;    // flat K2: 1 = every wave keeps a copy of the laser points in LDS (set per launch by sfw_launch_social:
                      // launches that leave the GPU under-filled)
};

// Shared-prefix rollout (K2).  Under the acceleration limits the robot's first P steps are
// bit-identical for every sample of a CLASS (same clipped linear-velocity sequence x same clipped
// angular-velocity sequence), hence so is the whole pedestrian simulation of those steps.  Classes
// refine as P grows, so the shared steps form a tree: level l simulates steps [P(l-1), P(l)) once per
// class of level l, each class resuming from the record its parent class (level l-1) left and leaving
// its own 64-byte record per agent; the suffix phase resumes every sample from its class of the last
// level for steps [P(last), S).  Same arithmetic on the same values in the same order: costs are
// bit-identical to the unshared rollout.
struct sfw_cls_agent {
  double px, py, vx, vy;  // state after the level's last step
  double fx, fy;          // desired + obstacle (+ group) force at that state = the next step's starting force
  double sw;              // the agent slot's social-work sum over the steps so far
  int32_t hasgoal, pad;
};
enum { SFW_PHASE_WHOLE = 0, SFW_PHASE_PREFIX = 1, SFW_PHASE_SUFFIX = 2 };

// Selection record (see sfw_best / sfw_best_key in the public header).
struct sfw_sel {
  double cost;       // +inf when nothing selectable
  double neg_linvel;
  double abs_angvel;
  long long neg_index;  // -(global iteration index)
  long long n_valid;
};

// Everything the kernels need that is uniform over a launch.
struct sfw_launch {
  // scoring parameters
  sfw_params p;
  sfw_derived k;    // filled by sfw_derive
  int32_t S;        // num_steps
  double dt;        // sim_time / S
  // robot + goal
  sfw_robot_state rs;
  sfw_goal_args ga;
  double vy_samp;   // 0.0 for the grid loop
  int32_t skip_zero_sample;  // 1 for the grid loop (:349-352), 0 for score_one
  // samples: sample t = chunk_begin + local index; iv = t / nw, iw = t % nw
  const double *linvels;
  const double *angvels;
  int32_t nv, nw;
  int64_t chunk_begin;  // first sample of this chunk
  int64_t chunk_count;  // samples in this chunk
  // costmap snapshot
  const uint8_t *cells;
  uint32_t size_x, size_y;
  double origin_x, origin_y, resolution;
  const double *footprint;  // K x (x,y)
  int32_t K;
  // agents (index 0 = robot) and shared obstacle points
  const double *agent_pos;         // A x (x,y)
  const double *agent_vel;         // A x (vx,vy)
  const sfw_agent_const *agent_c;  // A
  const double *agent_rest;        // A x (fx, fy) or null: angular terms of the pairs at exact relative rest in the
                                   // handed-over state, evaluated on the host (sfw_capi.hip rest_forces)
  const double *pin_rest;          // 4 + A doubles or null: a person that can never move next to a robot that stands still for
                                   // the whole rollout is at exact relative rest at EVERY step; {robot x, y, lateral force on
                                   // the robot (x, y), Wp correction per person — 0 when the person is not pinned —} evaluated
                                   // on the host (sfw_capi.hip pinned_rest_table) for the steps whose robot record is (x, y)
                                   // at velocity 0; the kernels add entry 4 + i to person i's social work unconditionally
  int32_t A;
  const double *obstacles;         // O x (x,y)
  int32_t O;
  // groups of >= 1 agents with group_id >= 0 (dense index q, CSR member lists)
  const int32_t *agent_grp;        // A : dense group index or -1
  const int32_t *grp_off;          // NG + 1
  const int32_t *grp_mem;          // grp_off[NG] member agent indices, ascending per group
  int32_t NG;
  int32_t n_grp_mem;
  // shared-prefix rollout (see sfw_cls_agent); chunks are whole grid rows when phase != WHOLE.
  // Items of a PREFIX launch are the classes of one level (row class x column class), items of a
  // SUFFIX / WHOLE launch are the chunk's samples.
  int32_t phase;                 // SFW_PHASE_*
  int32_t step_begin, step_end;  // steps this launch integrates
  int32_t resume;                // 1: start from in_state records instead of the initial agents
  int32_t force_alive;           // 1: K2 also integrates samples K1 rejected on the costmap (point dumps only)
  int32_t k2_form;               // SFW_K2_AUTO / _REGISTER / _FLAT: which organisation of a K2 wave (sfw_set_k2_form)
  int32_t n_cu, n_xcd;           // compute units / XCDs of the device (organisation thresholds, XCD-contiguous block order)
  int32_t n_cls, n_col_cls;      // PREFIX: classes of this level, its column classes
  // flat K2 only: the launch covers the items from item_base on (0: all of them) — a register-form launch may hand its last
  // items to flat-form waves (sfw_launch_social, sfw_split_streams)
  int32_t item_base;
  const int32_t *row_rep;        // PREFIX [row classes]  chunk-local row whose robot records represent the class
  const int32_t *col_rep;        // PREFIX [n_col_cls]    column likewise
  // where an item resumes from: record row_src[r] * n_col_src + col_src[c] of in_state, with (r, c) =
  // (row class, column class) of a PREFIX item, (chunk-local row, column) of a sample
  int32_t n_col_src;
  const int32_t *row_src, *col_src;
  const sfw_cls_agent *in_state;  // [source classes][A]
  const int32_t *in_dead;         // [source classes] 0, or 2 + step of a pedestrian contact so far
  sfw_cls_agent *out_state;       // PREFIX [n_cls][A]
  int32_t *out_dead;              // PREFIX [n_cls]
  // flat social kernel: pair u -> LDS plane byte offsets 8*i (first array) and 8*j (second array), padded with the
  // dummy slot 8*A; see sfw_launch_pair_table
  const uint16_t *pair_tab;
  // per-sample outputs, indexed by GLOBAL sample index
  int32_t *status;       // T
  double *base_cost;     // T : vel + distance + angle + costmap terms (ref :663-666)
  double *costs;         // T : final cost or sentinel
  int32_t *coll_step;    // T : step at which a pedestrian touched the robot (ref :613-627), -1 if none; nullable
  // per-chunk tables (see sfw_unit)
  sfw_unit *ptab;        // [step][row_units]: positions of the chunk's samples, then velocities of its grid rows
  sfw_unit *cs_tab;      // [step][nw]: cos / sin of the heading before the step, per grid column
  int64_t row_units;     // units per step row of ptab (>= rstep_stride + grid rows the chunk touches)
  int16_t *fcode;        // footprint cost per (step, sample): -3,-2,-1 or 0..253, [step][rstep_stride]
  int64_t rstep_stride;  // samples per step row (position units in a ptab row, codes in an fcode row)
  // optional Trajectory-points dump (x,y,theta per pre-step pose) of the chunk's samples
  double *points;        // nullable, chunk_count x S x 3 doubles
  int32_t *n_points;     // nullable, chunk_count ints
  // measurement only (sfw_set_timing): the middle block of a K2 launch leaves {s_memtime, s_memrealtime} at its start in [0..1]
  // and at its end in [2..3]; their ratio is the shader clock the kernel really ran at.  Nullable.
  unsigned long long *clock_probe;
  // one launch per control cycle (sfw_cycle_kernel): the selection's inputs and outputs ride with the scoring launch
  int64_t index_base;       // global iteration index of sample 0 (sfw_grid_stage)
  unsigned *cycle_counter;  // one word, zero between launches: blocks that have written their cost
  sfw_sel *sel_out;         // the selection record (device)
  double *costs_host;       // nullable: pinned mirror of the cost vector ...
  sfw_sel *sel_host;        // ... and of the record
  // sfw_cycle_kernel only, nullable: the stage's arena (footprint | agents | sample vectors | ...) has NOT been copied to the
  // device; every block copies it from the host's pinned memory to arena_dev itself, in front of everything else (all blocks
  // write the same bytes).  A control cycle whose costmap has not changed is then ONE kernel and no copy at all.
  const char *arena_host;
  char *arena_dev;
  uint32_t arena_bytes;
};


// Fills L.k from L.p and L.O (host).
void sfw_derive(sfw_launch &L);

// Launchers (sfw_kernels.hip).  All enqueue on `stream` and return hipError_t.
hipError_t sfw_launch_rollout(const sfw_launch &L, hipStream_t stream);  // = poses, then costmap
// The two halves of K1: the social kernel's prefix phase needs only the first (the robot-step
// table), so the second can run beside it on another stream.
hipError_t sfw_launch_rollout_poses(const sfw_launch &L, hipStream_t stream);
hipError_t sfw_launch_rollout_costmap(const sfw_launch &L, hipStream_t stream);
// sp (optional): a second stream and two events, with which a register-form launch whose waves do not divide evenly over
// the SIMDs hands its last items to concurrent flat-form waves (bit-identical organisations; sfw_kernels.hip split_point)
struct sfw_split_streams {
  hipStream_t side;
  hipEvent_t fork, join;
};
hipError_t sfw_launch_social(const sfw_launch &L, hipStream_t stream, const sfw_split_streams *sp = nullptr);
// K1 + K2 + K3 of a small grid in ONE launch (sfw_kernels.hip sfw_cycle_kernel); sfw_cycle_applies says whether a launch
// qualifies (L.cycle_counter / sel_out / index_base set; costs_host / sel_host optional)
bool sfw_cycle_applies(const sfw_launch &L);
hipError_t sfw_launch_cycle(const sfw_launch &L, hipStream_t stream);
#ifndef SFW_STRICT_BUILD
// the same launcher over the K2 kernels compiled with the longer polynomials (sfw_kernels_strict.hip): SFW_PRECISION_F64_STRICT
// samples of a launch over T samples that the register form hands to flat-form waves (0: none)
int64_t sfw_social_flat_items(int A, int O, int NG, int64_t T, int form, int cus);
hipError_t sfw_launch_social_strict(const sfw_launch &L, hipStream_t stream, const sfw_split_streams *sp = nullptr);
hipError_t sfw_launch_cycle_strict(const sfw_launch &L, hipStream_t stream);
#endif
// true when sfw_launch_rollout_poses runs all of K1 in one launch (small grids): the only form that writes L.points
bool sfw_rollout_is_fused(const sfw_launch &L);
// Reduces costs[0..T) to one sfw_sel at *out (device memory).  partials must
// hold >= sfw_argmin_partials(T) records.
int64_t sfw_argmin_partials(int64_t T);
// costs_host / sel_host (nullable): host-visible pinned memory that receives a copy of the cost vector and of the record
hipError_t sfw_launch_argmin(const double *costs, const double *linvels, const double *angvels,
                             int32_t nw, int64_t T, int64_t index_base, sfw_sel *partials,
                             sfw_sel *out, hipStream_t stream, double *costs_host = nullptr,
                             sfw_sel *sel_host = nullptr);
// Row r of the [R,5] multi-device exchange table from a selection record (+inf in every other row).
hipError_t sfw_launch_key_table(const sfw_sel *sel, double *table, int r, int R, hipStream_t stream);
// Pair table of the flat social kernel for A agents: sfw_pair_table_entries(A) uint16 entries.
int64_t sfw_pair_table_entries(int A);
hipError_t sfw_launch_pair_table(uint16_t *tab, int A, hipStream_t stream);
// Samples handled by one wave of the social kernel for A agents (form: SFW_K2_*).
// (cus: compute units of the device, sfw_launch.n_cu)
int sfw_samples_per_wave(int A, int64_t T, int form, int cus);
size_t sfw_social_lds_bytes(int A, int O, int NG, int n_grp_mem, int64_t T, int form, int cus);
// SFW_ORG_* of a K2 launch over T items
int sfw_social_organisation(int A, int64_t T, int O, int form, int cus);

#endif  // SFW_DEVICE_H_
