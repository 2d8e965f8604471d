/*
 * sfw_hip.h — C ABI of the MI355X-native DWA rollout + social-force scorer.
 *
 * This is the drop-in boundary for ONE hot path of
 * robotics-upo/social_force_window_planner: the (v,w) sample loop of
 * SFWPlanner::findBestAction (reference src/sfw_planner.cpp:338-417) and the
 * two single-sample calls of SFWPlanner::scoreTrajectory
 * (src/sfw_planner.cpp:204-206 and :299-301; signature
 * include/social_force_window_planner/sfw_planner.hpp:309-314).
 *
 * Plain C: POD structs, plain pointers and sizes, int status codes, no C++
 * types and no exceptions across the boundary.  All floating point inputs are
 * double, as in the reference; the two places where the reference computes in
 * float (robot_radius_, normalizeAngle) are float here too.
 *
 * Ownership: the caller owns every buffer it passes in; the library copies
 * what it needs into its own pinned/device buffers during the call.  Output
 * buffers are caller-allocated.
 *
 * Threading: one handle = one planner = one caller thread at a time (the
 * reference holds configuration_mutex_ for the whole findBestAction,
 * src/sfw_planner.cpp:123-454).  Each handle owns one HIP stream.
 *
 * Error convention: every function returns SFW_OK (0) or a negative
 * sfw_status.  An invalid trajectory is DATA, not an error: its cost is
 * exactly -1.0 (src/sfw_planner.cpp:549,561,572,625), never NaN.  Non-finite
 * inputs (agents, laser points, robot state, goal arguments, sample
 * velocities) are refused with SFW_ERR_INVALID_ARG, so no NaN reaches a cost.
 */
#ifndef SFW_HIP_H_
#define SFW_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SFW_ABI_VERSION 2 /* 2 (round 6): sfw_plan_info.one_launch, sfw_plan_axis_classes */

typedef enum sfw_status {
  SFW_OK = 0,
  SFW_ERR_INVALID_ARG = -1,  /* null pointer, negative size, ...            */
  SFW_ERR_NO_DEVICE = -2,    /* no HIP device / kernels cannot be launched  */
  SFW_ERR_HIP = -3,          /* a HIP runtime call failed (see last_error)  */
  SFW_ERR_STATE = -4,        /* call order violated (e.g. no costmap set)   */
  SFW_ERR_UNSUPPORTED = -5   /* input does not fit the device (e.g. LDS)     */
} sfw_status;

/* Cost sentinels written into the per-sample cost vector. */
#define SFW_COST_INVALID (-1.0) /* reference "return -1.0"                    */
#define SFW_COST_SKIPPED (-2.0) /* the (0,0) sample the grid loop never scores
                                   (src/sfw_planner.cpp:349-352)             */

/* Robot agent id for callers that have none to give (the reference never sets
 * agents_[0].id, src/sensor_interface.cpp:31-37): no people_msgs tag parses to
 * it, so no person's robot-induced social work is skipped by an id collision. */
#define SFW_ROBOT_ID_NONE INT32_MIN

/* Arithmetic mode of the social-force kernel. */
#define SFW_PRECISION_F64 0 /* parity mode: everything in double            */
#define SFW_PRECISION_F32 1 /* fast mode: agent state, integration and every
                               threshold stay double; only the pair/obstacle
                               FORCES are evaluated in float  */
#define SFW_PRECISION_F64_STRICT 2 /* everything in double as SFW_PRECISION_F64,
                               with the angle's polynomial one degree longer
                               (asin 8 / exp 9 against the default's 7 / 9 —
                               7 / 8 until round 4 —: the pair term at ~1e-14
                               relative instead of ~5e-14; K2 +1 %).  Kept for
                               callers that selected it; since the default
                               evaluates the exponential at degree 9 the two
                               are within a digit of each other */

/*
 * Scoring parameters = the subset of ControllerParams
 * (sfw_planner.hpp:55-66,186-226) that scoreTrajectory reads, plus lightsfm's
 * sfm::Parameters (never overridden by the reference).
 * Fill with sfw_params_default() and then override.
 */
typedef struct sfw_params {
  double max_vel_x;        /* max_trans_vel, 0.7  (:58, used at :654)       */
  double sim_time;         /* 1.0   (:61, :519)                             */
  double sim_granularity;  /* 0.025 (:61, :519)                             */
  float robot_radius;      /* 0.35, FLOAT on purpose (:62,:208, :617)       */
  float reserved0;
  double social_weight;    /* 1.2 (:65)                                     */
  double costmap_weight;   /* 2.0                                           */
  double angle_weight;     /* 0.7                                           */
  double distance_weight;  /* 1.0 (:66)                                     */
  double vel_weight;       /* 1.0                                           */
  double robot_goal_radius; /* 0.20, the per-step robot goal (:608)         */
  /* lightsfm sfm::Parameters defaults */
  double sfm_force_factor_desired;  /* 2.0  */
  double sfm_force_factor_obstacle; /* 10.0 */
  double sfm_force_sigma_obstacle;  /* 0.2  */
  double sfm_force_factor_social;   /* 2.1  */
  double sfm_lambda;                /* 2.0  */
  double sfm_gamma;                 /* 0.35 */
  double sfm_n;                     /* 2.0  */
  double sfm_n_prime;               /* 3.0  */
  double sfm_relaxation_time;       /* 0.5  */
  double sfm_force_factor_group_gaze;      /* 3.0 */
  double sfm_force_factor_group_coherence; /* 2.0 */
  double sfm_force_factor_group_repulsion; /* 1.0 */
  int32_t precision;                /* SFW_PRECISION_*                      */
  int32_t reserved1;
} sfw_params;

/*
 * One social agent = the fields of sfm::Agent the hot path consumes
 * (built by SFMSensorInterface, src/sensor_interface.cpp:31-37,442-527,
 * 552-580).  Index 0 of the array handed to sfw_set_agents is the robot
 * (src/sfw_planner.cpp:155).
 */
typedef struct sfw_agent {
  double x, y;             /* position                                      */
  double vx, vy;           /* velocity; for the robot this is the ROBOT-LOCAL
                              twist, as in sensor_interface.cpp:566-575     */
  double goal_x, goal_y;   /* goals.front().center (ignored if !has_goal)   */
  double goal_radius;      /* goals.front().radius                          */
  double desired_velocity; /* Agent::desiredVelocity.  <= 0 is accepted, as
                              the reference accepts people_velocity_ = 0
                              (sensor_interface.cpp:503): the speed clamp of
                              updatePosition pins a person with 0 where it
                              stands.  Next to a robot without twist it is
                              at exact relative rest at EVERY step, where
                              lightsfm's sign(theta) is the rounding noise of
                              two atan2 (-1, 0 or +1): reproduced for the
                              handed-over state and for a robot that stands
                              still from the start (the linvel = 0 samples),
                              NOT for one that brakes to a stop during the
                              rollout — the angular term of that pair is 0
                              from there on                  */
  double radius;           /* Agent::radius                                 */
  int32_t has_goal;        /* goals non-empty (people: 1, robot at t0: 0)   */
  int32_t id;              /* Agent::id — the robot-on-person force skips a
                              person whose id equals the robot's            */
  int32_t group_id;        /* Agent::groupId (people_msgs tags[1]); < 0 = no
                              group.  Members of a group of >= 2 agents feel
                              lightsfm's gaze/coherence/repulsion forces    */
  int32_t reserved;
} sfw_agent;

/* Robot pose and velocity as findBestAction hands them to scoreTrajectory.
 * The reference truncates all six to float first (src/sfw_planner.cpp:145-152)
 * — the host mirror does that; the ABI takes whatever it is given. */
typedef struct sfw_robot_state {
  double x, y, theta;
  double vx, vy, vtheta;
} sfw_robot_state;

/* Per-call sample-independent arguments of scoreTrajectory. */
typedef struct sfw_goal_args {
  double acc_x, acc_y, acc_theta; /* max_trans_acc_, 0.0, max_rot_acc_      */
  double wpx, wpy;                /* current way-point                      */
} sfw_goal_args;

/* Result of the selection rule (src/sfw_planner.cpp:394-414, :426-468). */
typedef struct sfw_best {
  int64_t index;    /* iv*nw+iw of the winner, -1 if no sample is selectable */
  double cost;      /* winner's cost, -1.0 if none                           */
  double vx, vy, vtheta; /* cmd_vel (0,0,0 if none)                          */
  int64_t n_valid;  /* samples with cost >= 0                                */
} sfw_best;

/* 4-double key used for the multi-GPU exchange: lexicographic minimum over
 * ranks reproduces the reference's selection order exactly
 * (cost up, linvel down, |angvel| up, iteration index down). */
typedef struct sfw_best_key {
  double cost;        /* +inf if the rank holds no selectable sample         */
  double neg_linvel;  /* -linvel                                             */
  double abs_angvel;  /* |angvel|                                            */
  double neg_index;   /* -(global iteration index)                           */
} sfw_best_key;

typedef struct sfw_planner_s *sfw_handle;

/* ---- lifecycle --------------------------------------------------------- */
void sfw_params_default(sfw_params *p);
int sfw_abi_version(void);
/* device: HIP device ordinal.  Fails with SFW_ERR_NO_DEVICE when no GPU is
 * visible — there is no CPU fallback in this library. */
int sfw_create(const sfw_params *params, int device, sfw_handle *out);
int sfw_destroy(sfw_handle h);
/* The reference re-reads its parameters on every cycle (:125). */
int sfw_set_params(sfw_handle h, const sfw_params *params);
const char *sfw_last_error(sfw_handle h);

/* ---- world state (replaces const Costmap2D& / footprint_spec_ / getAgents) */
/* The three calls snapshot their arguments on the host (the caller's buffers may change on return); what has changed
 * reaches the device with the NEXT sfw_grid_stage / sfw_score_* — a launch of a grid staged before the call still
 * scores the old state.
 * cells: row-major, y outer, cells[my*size_x+mx] == Costmap2D::getCost(mx,my) (the reference reads nav2's live
 * costmap, sfw_planner.hpp:362).  A snapshot whose cells and geometry equal the last one's is recognised (a memcmp)
 * and not sent again: hand the live map over every cycle. */
int sfw_set_costmap(sfw_handle h, const uint8_t *cells, uint32_t size_x,
                    uint32_t size_y, double origin_x, double origin_y,
                    double resolution);
/* sfw_set_footprint / sfw_set_agents keep a host copy; the next stage
 * (sfw_grid_stage, sfw_score_grid, sfw_score_one) uploads footprint, agents,
 * laser points and the sample vectors in ONE copy.  A sfw_grid_launch without
 * a new stage keeps using what the last stage uploaded. */
/* xy: K points (x0,y0,x1,y1,...) in the robot frame = footprint_spec_
 * (src/sfw_planner.cpp:32).  K < 3 => centre-cell check only
 * (src/costmap_model.cpp:41-48). */
int sfw_set_footprint(sfw_handle h, const double *xy, int32_t K);
/* agents[0] = robot.  obstacles_xy: O laser points shared by every agent's
 * obstacles1 (src/sensor_interface.cpp:513-524).  At most 8190 agents
 * (SFW_ERR_UNSUPPORTED beyond; a set that does not fit one wave's 160 KiB of
 * LDS — roughly 2000 agents — is refused by the scoring call the same way). */
int sfw_set_agents(sfw_handle h, const sfw_agent *agents, int32_t A,
                   const double *obstacles_xy, int32_t O);

/* ---- scoring ----------------------------------------------------------- */
/*
 * The grid loop of findBestAction (src/sfw_planner.cpp:345-417): for every
 * (linvels[iv], angvels[iw]), iv outer, cost = scoreTrajectory(rs, linvel,
 * 0.0, angvel, args...); costs_out[iv*nw+iw] receives it (SFW_COST_SKIPPED
 * for the (0,0) sample).  best_out receives the reference's selection.
 * Blocking: returns after the D2H copy.  costs_out / best_out may be NULL.
 */
int sfw_score_grid(sfw_handle h, const sfw_robot_state *rs,
                   const double *linvels, int32_t nv, const double *angvels,
                   int32_t nw, const sfw_goal_args *args, double *costs_out,
                   sfw_best *best_out);

/*
 * One scoreTrajectory call (the two scalar call sites :204-206, :299-301).
 * points_xyth (nullable) receives up to points_cap (x,y,theta) triples = the
 * Trajectory points (src/sfw_planner.cpp:578); *n_points their count.
 */
int sfw_score_one(sfw_handle h, const sfw_robot_state *rs, double vx_samp,
                  double vy_samp, double vtheta_samp, const sfw_goal_args *args,
                  double *cost_out, double *points_xyth, int32_t points_cap,
                  int32_t *n_points);

/* ---- device-resident pipeline (what sfw_score_grid is made of) --------- */
/* Stage one grid call: one H2D copy of footprint, agents, laser points and
 * the sample vectors; robot state and goal arguments are recorded.  index_base
 * is the global iteration index of this rank's first sample (multi-GPU
 * sharding by linvel rows); 0 on a single GPU. */
int sfw_grid_stage(sfw_handle h, const sfw_robot_state *rs,
                   const double *linvels, int32_t nv, const double *angvels,
                   int32_t nw, const sfw_goal_args *args, int64_t index_base);
/* Enqueue rollout + social-force + argmin kernels on the handle's stream.
 * Everything stays in HBM; no host sync. */
int sfw_grid_launch(sfw_handle h);
int sfw_grid_sync(sfw_handle h);
/* Wait for the launch; the cost vector (nullable) and the local selection (nullable).  Since round 6 the selection kernels
 * themselves leave vector and record in a pinned host buffer of the handle (grids of up to SFW_MIRROR_MAX_MB, default
 * 64 MB, of costs): the call is a wait on the stream — polling it for up to SFW_SPIN_US microseconds, default 20000,
 * before it blocks — plus one memcpy into costs_out; larger grids are copied device-to-host here. */
int sfw_grid_fetch(sfw_handle h, double *costs_out, sfw_best *best_out,
                   sfw_best_key *key_out);
/* The cost vector of the last fetched launch where the handle holds it on the host (nv * nw doubles, sample order;
 * valid until the next sfw_grid_launch / sfw_score_* on this handle), or NULL when this launch was not mirrored (then
 * pass costs_out).  A caller that only reads the vector — a marker publisher, the tests — saves the memcpy:
 * sfw_grid_fetch(h, NULL, &best, NULL) and this. */
const double *sfw_grid_costs_view(sfw_handle h);
/* How the staged grid will be launched.  levels > 0: the shared-prefix
 * rollout is in use — under the acceleration limits (sfw_planner.hpp:457-463)
 * the robot's first steps are bit-identical for all samples of a class (same
 * clipped linear x angular velocity sequences), and classes refine step by
 * step, so the first split_step steps are simulated along a tree of `levels`
 * levels of classes (class_steps class-steps in all instead of
 * samples * split_step sample-steps) and every sample resumes from its class
 * of the last level (`classes` of them).  Costs are bit-identical to the
 * plain rollout; SFW_PREFIX=0 in the environment of sfw_create switches it
 * off, SFW_PREFIX=3,7,12 forces the levels' end steps. */
#define SFW_ORG_NONE 0       /* no agents: the social-force kernel is not launched             */
#define SFW_ORG_REGISTER_1 1 /* register-resident, one agent slot per lane (A <= 64, floor(64/A) samples per wave) */
#define SFW_ORG_REGISTER_2 2 /* register-resident, two agent slots per lane (64 < A <= 128)      */
#define SFW_ORG_FLAT 3       /* all unordered pairs flattened over the lanes, one sample per wave */
typedef struct sfw_plan_info {
  int32_t split_step;  /* last shared step + 1; 0: plain rollout             */
  int32_t levels;
  int32_t chunks;      /* launches of the K1->K2 table (SFW_TABLE_BUDGET_MB) */
  int32_t organisation; /* SFW_ORG_*: how a wave of the launch over the SAMPLES is organised (the suffix
                           launch of the shared-prefix rollout, or the whole rollout); the prefix levels
                           pick theirs by their class counts.  Costs do not depend on it.  */
  int64_t classes;     /* classes of the last level, summed over chunks      */
  int64_t class_steps; /* sum over levels of classes x steps of the level    */
  int64_t samples;     /* nv * nw                                            */
  int64_t flat_samples; /* organisation = SFW_ORG_REGISTER_1 only: samples (of the first chunk's launch) that the launch
                           hands to flat-form waves running beside the register-form ones, so that every SIMD holds
                           the same number of those (0: none).  Costs do not depend on it.                      */
  int64_t one_launch;   /* 1: a control cycle's grid (<= 1024 samples, fewer than 64 agents, flat form) — rollout,
                           footprint checks, pedestrian simulation and selection run as ONE kernel launch.  Costs,
                           sentinels and selection do not depend on it (SFW_CYCLE_FUSED=0 in the environment: never). */
  int64_t rest_noise_unreproduced; /* 1: this stage holds the one configuration whose reference result is NOT reproduced
                           (see sfw_agent.desired_velocity): a person that can never move (desired_velocity == 0) AND a
                           robot that moves now but BRAKES TO A STOP inside the rollout of some sample (a linvel sample of
                           0 whose deceleration ends before the horizon).  From the step the robot stands, the pair is at
                           exact relative rest at a position the device's own pose rollout produced; lightsfm's lateral
                           term there is the rounding noise of two atan2 on the host's libm (0 for most geometries, a
                           full-magnitude +-1 for the others) and the kernels' is 0.  Affects those samples' social work
                           only; 0 for every other stage, incl. the robot that stands still from the start (reproduced). */
} sfw_plan_info;
int sfw_grid_plan_info(sfw_handle h, sfw_plan_info *out);
/* The plan a single-chunk stage of this grid would choose ON A WHOLE MI355X
 * (256 compute units; SFW_DEVICE_CUS in the environment pretends another
 * count, as it does for a handle — a handle on a partition reads its own
 * device and may plan differently: sfw_grid_plan_info says what it chose),
 * computed on the host alone (no handle, no device): end steps of the levels and their class
 * counts (row classes x column classes), up to cap entries; *n_levels = 0
 * when sharing does not pay (fewer than 4096 samples, fewer than two agents,
 * or nothing to share).  vx0 / vtheta0: the robot's current velocities,
 * acc_*: sfw_goal_args, num_steps as scoreTrajectory derives it (:519-525). */
int sfw_plan_shared_prefix(const double *linvels, int32_t nv,
                           const double *angvels, int32_t nw, double vx0,
                           double vtheta0, double acc_x, double acc_theta,
                           double sim_time, int32_t num_steps,
                           int32_t n_agents, int32_t *level_ends,
                           int64_t *level_classes, int32_t cap,
                           int32_t *n_levels);
/* One axis of that plan (host only; diagnostics and tests): the classes of `n` target velocities whose first p clipped
 * velocities (computeNewVelocity, sfw_planner.hpp:457-463, from the current velocity v0 under a_max, dt) are bit-equal,
 * for p = 1 .. *n_levels <= max_p (the walk stops once every target is its own class).  counts[p-1] = classes of level
 * p; classes (nullable) receives [level][n] class ids, numbered in target order.  form 0: the planner's own choice (the
 * closed form over the two "not yet reached" groups where its premises hold — *closed_form says so —, else the generic
 * walk of every target's recurrence); form 1: the generic walk.  Both give the same classes. */
int sfw_plan_axis_classes(const double *targets, int32_t n, double v0, double a_max, double dt, int32_t max_p,
                          int32_t form, int32_t *counts, int32_t *classes, int32_t *n_levels, int32_t *closed_form);
/* Contiguous blocks of linvel rows of about equal PLANNED work for R ranks (host only): row0[0..R], rank r takes rows
 * [row0[r], row0[r+1]).  The share of steps the shared-prefix tree saves differs along the row axis (BASELINE cfg5 cut
 * into 8 equal blocks integrates 64..76 % of its steps per block), so equal row counts are unequal work; the cuts follow
 * the planned class-steps instead.  Equal row counts when nothing is shared (fewer than 4096 samples per rank, fewer
 * than two agents).  sfw_multi_score_grid cuts this way; multi-process callers use it to agree on the same cut. */
int sfw_plan_row_blocks(const double *linvels, int32_t nv, const double *angvels, int32_t nw, double vx0,
                        double vtheta0, double acc_x, double acc_theta, double sim_time, int32_t num_steps,
                        int32_t n_agents, int32_t R, int32_t *row0);
/* Tuning / test knob: which organisation the social-force kernel's waves use.  SFW_K2_AUTO (default) picks
 * per launch by agent and item count; SFW_K2_REGISTER / SFW_K2_FLAT force one wherever it exists for the
 * agent count (register: A <= 128; flat: A >= 2, or laser points).  The organisations are bit-identical in
 * their results.  Takes effect at the next stage.  SFW_FORCE_FLAT=0|1 in the environment of sfw_create sets
 * the handle's initial value. */
#define SFW_K2_AUTO (-1)
#define SFW_K2_REGISTER 0
#define SFW_K2_FLAT 1
int sfw_set_k2_form(sfw_handle h, int32_t form);
/* Per-kernel HIP events around the kernels of sfw_grid_launch, off by default
 * (a control cycle is latency-bound; four event records cost as much as a
 * kernel).  Measurement tooling (bench.py) switches them on. */
int sfw_set_timing(sfw_handle h, int32_t enabled);
/* HIP-event time (ms) of the most recent sfw_grid_launch (SFW_ERR_STATE unless
 * timing was on): which = 0 whole launch, 1 rollout kernels, 2 social-force
 * kernel, 3 argmin. */
int sfw_last_launch_ms(sfw_handle h, int32_t which, float *ms_out);
/* Shader clock (GHz) the social-force kernel of the most recent timed sfw_grid_launch really ran at: the middle wave of the
 * launch over the samples reads the core-clock and the constant-rate counters when it starts and when it ends.
 * 0.0 when there was nothing to sample (no agents).  Boxes and thermal states differ by ~10 %: a kernel time is
 * only comparable across runs next to this number.  SFW_ERR_STATE unless timing was on. */
int sfw_last_clock_ghz(sfw_handle h, double *ghz_out);
/* Optional dump of the per-step robot poses of sample `index` of the last
 * launch (Trajectory points for RViz markers, :366-374). */
int sfw_grid_points(sfw_handle h, int64_t index, double *points_xyth,
                    int32_t points_cap, int32_t *n_points);
/* The same for `count` consecutive samples starting at `first` in one call (the
 * reference fills one RViz marker per sample every cycle, :347-386).
 * points_xyth: count x steps x 3 doubles (steps = num_steps of the params),
 * n_points: count ints = poses the reference's Trajectory would hold: all of them
 * for a valid sample, those before the first illegal footprint pose, or 0..i for a
 * sample rejected by pedestrian contact at step i (0 for the (0,0) sample).
 * A sample that is illegal on the costmap at pose j AND touches a pedestrian at an earlier step i < j
 * reports i + 1 poses, as the reference does (it returns at the contact, :613-627): when the range holds
 * costmap-rejected samples their pedestrians are integrated in an extra pass of this call. */
int sfw_grid_points_batch(sfw_handle h, int64_t first, int64_t count, double *points_xyth, int32_t *n_points);
/* Marker support without a second rollout: with capture on, a scoring launch over a grid small enough for the
 * one-launch rollout (at most 2048 samples and 512 steps: a control cycle's 5 x 9 samples) also leaves every
 * sample's Trajectory points, point count and contact step on the device, and sfw_grid_points(_batch) of that
 * launch is ONE device-to-host copy instead of re-running the rollout (and, when a sample was rejected on the
 * costmap, the pedestrian integration).  Costs and selection are unaffected.  Larger grids ignore it.  Off by
 * default; takes effect at the next sfw_grid_launch / sfw_score_grid. */
int sfw_set_points_capture(sfw_handle h, int32_t enabled);
/* Raw HIP stream (hipStream_t) the handle launches on, for callers that want
 * to record their own events. */
void *sfw_stream(sfw_handle h);

/* ---- one process, several devices ---------------------------------------
 * The reference plugin is ONE process (sfw_plugin.xml:1-9; computeVelocityCommands,
 * src/sfw_planner_node.cpp:220-331), so a host that wants the (v,w) grid on several
 * MI355X drives them from there: one sfw_handle per listed device, the linvel rows
 * (the OUTER loop of src/sfw_planner.cpp:345) split into R contiguous blocks of equal
 * planned work (sfw_plan_row_blocks; [r*nv/R, (r+1)*nv/R) when the shared-prefix rollout
 * has nothing to share), world state replicated by each handle's own upload, and the
 * winner picked by ONE ncclAllReduce(min) over xGMI of an [R,5] double table in which
 * rank r fills its own row (sfw_best_key + n_valid) and +inf elsewhere — the
 * lexicographic row minimum is the reference's selection order (:394-414).  RCCL is
 * loaded on first use (dlopen librccl.so): single-device callers never touch it. */
typedef struct sfw_multi_s *sfw_multi_handle;
#define SFW_MULTI_RCCL 0        /* devices must be distinct; exchange = ncclAllReduce(min)            */
#define SFW_MULTI_HOST_REDUCE 1 /* exchange on the host from each rank's 40-byte row: no RCCL needed, a
                                   device may be listed more than once (tests on a one-GPU box)       */
int sfw_multi_create(const sfw_params *params, const int *devices, int32_t R, int32_t exchange,
                     sfw_multi_handle *out);
int sfw_multi_destroy(sfw_multi_handle m);
const char *sfw_multi_last_error(sfw_multi_handle m);
int32_t sfw_multi_ranks(sfw_multi_handle m);
/* What the handle really runs on (diagnostics; a scaling record must say what it measured): the devices as listed, the
 * exchange, and — SFW_MULTI_RCCL — how many communicators ncclCommInitAll returned, the size communicator 0 reports
 * (ncclCommCount), the device every communicator reports (ncclCommCuDevice) and RCCL's version code (ncclGetVersion);
 * -1 / 0 where the library does not export the query.  At most the first 64 ranks are listed. */
typedef struct sfw_multi_desc {
  int32_t ranks, exchange, communicators, comm_size, rccl_version;
  int32_t devices[64], comm_devices[64];
  /* SFW_MULTI_RCCL: the file ncclAllReduce was resolved from (dladdr) and how it was found — "SFW_RCCL_LIB" (that path in
   * the environment, honoured first), "already mapped" (an RCCL this process had loaded — e.g. the copy torch bundles — is
   * reused rather than a second one opened beside it) or the name the loader was given (librccl.so, librccl.so.1,
   * /opt/rocm/lib/librccl.so).  Empty strings for SFW_MULTI_HOST_REDUCE. */
  char rccl_path[512], rccl_found[64];
} sfw_multi_desc;
int sfw_multi_describe(sfw_multi_handle m, sfw_multi_desc *out);
/* Rank r's handle (owned by m): for the single-sample calls (sfw_score_one on rank 0) and diagnostics. */
sfw_handle sfw_multi_rank_handle(sfw_multi_handle m, int32_t r);
/* World state and parameters, replicated to every rank. */
int sfw_multi_set_params(sfw_multi_handle m, const sfw_params *params);
int sfw_multi_set_costmap(sfw_multi_handle m, const uint8_t *cells, uint32_t size_x, uint32_t size_y,
                          double origin_x, double origin_y, double resolution);
int sfw_multi_set_footprint(sfw_multi_handle m, const double *xy, int32_t K);
int sfw_multi_set_agents(sfw_multi_handle m, const sfw_agent *agents, int32_t A, const double *obstacles_xy,
                         int32_t O);
/* sfw_score_grid over all ranks: same arguments, same results (costs bit-identical: a sample's cost does
 * not depend on how the grid is cut).  costs_out / best_out may be NULL. */
int sfw_multi_score_grid(sfw_multi_handle m, const sfw_robot_state *rs, const double *linvels, int32_t nv,
                         const double *angvels, int32_t nw, const sfw_goal_args *args, double *costs_out,
                         sfw_best *best_out);
/* The block of rows rank r scored in the last sfw_multi_score_grid. */
int sfw_multi_rank_rows(sfw_multi_handle m, int32_t r, int32_t *first_row, int32_t *n_rows);
/* Host wall-clock of the last call's phases, microseconds: which = 0 stage+launch of all ranks (enqueue),
 * 1 exchange (enqueue of the all-reduce + fetch of the table, i.e. until every rank's kernels are done),
 * 2 cost-vector fetches. */
int sfw_multi_last_us(sfw_multi_handle m, int32_t which, double *us_out);
/* Trajectory points of sample `index` of the last sfw_multi_score_grid (as sfw_grid_points). */
int sfw_multi_grid_points(sfw_multi_handle m, int64_t index, double *points_xyth, int32_t points_cap,
                          int32_t *n_points);

#ifdef __cplusplus
}
#endif
#endif /* SFW_HIP_H_ */
