// sfw_capi.hip — host side of the C ABI declared in include/sfw_hip.h.
// Owns the device buffers, the handle's HIP stream and events, and sequences
// the three kernels of sfw_kernels.hip.  There is NO CPU fallback: without a
// HIP device sfw_create fails with SFW_ERR_NO_DEVICE.

#include "sfw_device.h"

#include <dlfcn.h>
#include <link.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <vector>

namespace {

template <typename T> struct dev_buf {
  T *p = nullptr;
  size_t cap = 0;  // elements
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = n + n / 8 + 16;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), want * sizeof(T));
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

// Pinned host staging area for one logical upload.  The caller's buffer is copied
// here synchronously (so it may be freed or reused on return) and the H2D copy is
// enqueued on the handle's stream; `done` guards the area against being rewritten
// while that copy is still pending.  No stream synchronisation on the hot path.
struct pinned_buf {
  char *p = nullptr;
  size_t cap = 0;
  hipEvent_t done = nullptr;
  bool pending = false;
  hipError_t wait() {
    if (!pending) return hipSuccess;
    pending = false;
    return hipEventSynchronize(done);
  }
  hipError_t reserve(size_t bytes) {
    hipError_t e = wait();
    if (e != hipSuccess) return e;
    if (!done && (e = hipEventCreateWithFlags(&done, hipEventDisableTiming)) != hipSuccess) return e;
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = bytes + bytes / 4 + 256;
    e = hipHostMalloc(reinterpret_cast<void **>(&p), want, hipHostMallocDefault);
    if (e == hipSuccess) cap = want;
    return e;
  }
  hipError_t mark(hipStream_t s) {
    pending = true;
    return hipEventRecord(done, s);
  }
  void release() {
    (void)wait();
    if (p) (void)hipHostFree(p);
    if (done) (void)hipEventDestroy(done);
    p = nullptr;
    cap = 0;
    done = nullptr;
  }
};

}  // namespace

struct sfw_planner_s {
  sfw_params params;
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t side = nullptr;  // K1b + K1c beside the shared-prefix phase of K2
  hipEvent_t ev_poses = nullptr, ev_side = nullptr;
  // the samples' K2 launch may hand its last items to flat-form waves on `side` (sfw_launch_social, sfw_split_streams)
  sfw_split_streams split{nullptr, nullptr, nullptr};
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  std::vector<hipEvent_t> chunk_ev;  // 3 per chunk of a multi-chunk launch: K1 start, K1 end/K2 start, K2 end
  int n_chunks = 1;
  std::string err;

  // world state.  The costmap has its own snapshot buffer (uploaded by sfw_set_costmap).  Footprint,
  // agents, laser points, groups and the sample vectors are small: sfw_set_* keep them on the host
  // and every stage packs them into ONE pinned arena and issues ONE H2D copy into `world`
  // (device pointers below are offsets into it) — a control cycle is latency-bound and each
  // separate copy costs as much as a kernel launch.
  dev_buf<uint8_t> cells;               // the snapshot on the device when it is too large to ride in `world`
  uint32_t size_x = 0, size_y = 0;      // ... as the last stage uploaded it
  double origin_x = 0, origin_y = 0, resolution = 0;
  struct map_geometry {
    uint32_t size_x, size_y;
    double origin_x, origin_y, resolution;
  } map_new{0, 0, 0, 0, 0};             // what sfw_set_costmap handed over last (pin_map holds its cells)
  bool arena_pending = false;           // the last stage left its arena in pin_world only (see stage_common / flush_arena)
  size_t arena_from = 0, arena_bytes = 0;
  bool arena_direct_on = true;          // SFW_ARENA_DIRECT=0 in the environment of sfw_create: always copy at stage time
  bool cells_dirty = false;             // ... and the device has not seen yet
  bool cells_in_world = false;          // the device copy is the head of `world` (small maps), else `cells`
  size_t world_cells_bytes = 0;         // bytes of that head
  const uint8_t *d_cells = nullptr;
  bool have_costmap = false;
  std::vector<double> h_footprint;  // 2K
  int K = 0;
  std::vector<char> h_agents;       // pos | vel | const | obstacles | grp | off | mem, 16-byte aligned parts
  size_t ao_vel = 0, ao_cst = 0, ao_obs = 0, ao_grp = 0, ao_off = 0, ao_mem = 0;
  int A = 0, O = 0, NG = 0, n_grp_mem = 0;
  // ordered pairs (i, j) of agents with w x diff == 0 at hand-over (standing people, collinear walkers): see rest_forces
  std::vector<std::pair<int32_t, int32_t>> rest_pairs;
  const double *d_agent_rest = nullptr;  // A x (fx, fy) in `world`, or null when there is no such pair
  const double *d_pin_rest = nullptr;  // see pinned_rest_table: the stage's table on the device, or null
  int obs_tasks_force = -1;            // SFW_OBS_TASKS=0|1 in the environment of sfw_create: the flat form's laser-point pass as
                                       // one lane per agent / as (agent, segment) tasks, whatever sfw_derive prices (tuning aid)
  bool clock_cleared = false;          // the stage cleared the clock probe in front of the pose rollout it started
  bool pin_rest_on = true;             // SFW_PIN_REST=0 in the environment of sfw_create: no such table (tests: what it changes)
  size_t st_pin_doubles = 0;           // its length (4 + A) when the last stage built one
  std::vector<double> st_pin_pos;      // ... and what it was built from (sfw_set_params between stage and launch rebuilds it)
  std::vector<sfw_agent_const> st_pin_cst;
  std::vector<std::pair<int32_t, int32_t>> st_rest_pairs;  // what the last stage evaluated them from: sfw_set_params
  std::vector<double> st_rest_pv;                           // between stage and launch re-evaluates (pos | vel, 4A doubles)
  // what the last stage uploaded (sfw_set_* after a stage take effect at the next stage; a launch
  // in between must keep describing the device copy)
  int st_K = 0, st_A = 0, st_O = 0, st_NG = 0, st_n_grp_mem = 0;
  dev_buf<char> world;
  const double *d_footprint = nullptr, *d_agent_pos = nullptr, *d_agent_vel = nullptr, *d_obstacles = nullptr;
  const sfw_agent_const *d_agent_c = nullptr;
  const int32_t *d_agent_grp = nullptr, *d_grp_off = nullptr, *d_grp_mem = nullptr;
  const double *d_linvels = nullptr, *d_angvels = nullptr;
  dev_buf<uint16_t> pair_tab;  // rebuilt when the agent count changes
  int pair_tab_A = -1;

  // staged grid
  std::vector<double> h_lin, h_ang;
  int nv = 0, nw = 0;
  sfw_robot_state rs{};
  sfw_goal_args ga{};
  double vy_samp = 0.0;
  int skip_zero = 1;
  int64_t index_base = 0;
  bool staged = false, launched = false, launched_timed = false;
  bool launched_cycle = false;        // the last launch was the one-kernel control cycle (sfw_grid_plan_info)
  dev_buf<unsigned> cycle_counter;    // its "blocks done" word (zero between launches)
  bool fetched = false;               // ... and a fetch has waited for it (sfw_grid_costs_view)
  bool mirrored = false;              // the last launch's selection kernels left costs + record in pin_mirror (sfw_launch_argmin)
  size_t mirror_max_bytes = size_t(64) << 20;  // SFW_MIRROR_MAX_MB in the environment of sfw_create; 0: always copy
  long spin_us = 20000;               // SFW_SPIN_US in the environment of sfw_create: how long a fetch polls the stream before
                                      // it blocks in hipStreamSynchronize (0: block at once)
  bool timing = false;  // sfw_set_timing: record the per-kernel events sfw_last_launch_ms reads
  int k2_form = SFW_K2_AUTO;     // sfw_set_k2_form / SFW_FORCE_FLAT
  // shape of the device the launch heuristics are scaled to: compute units (hipDeviceProp_t.multiProcessorCount: 256 on a
  // whole MI355X, 32 on a CPX partition) and XCDs (32 CUs each); SFW_DEVICE_CUS / SFW_DEVICE_XCDS in the environment of
  // sfw_create override them (a test pretends 32 CUs: the plan changes, the results may not)
  int n_cu = SFW_DEFAULT_CUS, n_xcd = SFW_DEFAULT_CUS / SFW_CUS_PER_XCD;
  // sfw_set_points_capture: a grid small enough for the fused K1 (a control cycle's samples) leaves its Trajectory points,
  // point counts and contact steps during the scoring launch itself, in one buffer -> the dump is one D2H copy
  // K1a of a single-chunk grid is enqueued by the STAGE, before the shared-prefix planning (25..45 us of host work that
  // the pose rollout does not depend on): early_poses says the staged grid's robot-step table is already in the stream
  bool early_poses = false;
  bool early_poses_timed = false;  // ... and the stage recorded ev[0] in front of it (timing was on at stage time)
  bool capture_points = false, captured = false;
  int cap_S = 0;                 // step count the capturing launch ran with (the dump's layout)
  pinned_buf pin_cap;            // points (24 S T bytes) | n_points (4 T) | contact steps (4 T), written by the capturing launch

  // shared-prefix plan of the staged grid (sfw_device.h: sfw_cls_agent); no levels: not used
  struct level_tables {               // offsets (ints) into d_cls
    int32_t n_row = 0, n_col = 0;     // classes of the level: rows in this chunk x columns
    size_t o_row_rep = 0, o_col_rep = 0;  // representative chunk-local row / column of a class
    size_t o_row_src = 0, o_col_src = 0;  // class of the previous level a class resumes from (level > 0)
  };
  struct chunk_plan {
    std::vector<level_tables> lv;
    size_t o_row_cls = 0;             // chunk-local row -> row class of the last level
  };
  std::vector<int> prefix_steps;      // P(0) < P(1) < ...: level l integrates steps [P(l-1), P(l))
  std::vector<chunk_plan> prefix_chunks;
  size_t prefix_o_col_cls = 0;        // column -> column class of the last level
  int prefix_S = 0;
  int64_t prefix_chunk = 0;
  int64_t prefix_class_steps = 0, prefix_last_classes = 0;
  std::vector<int32_t> cls_ints;        // the plan's class tables as planned on the host (level_tables offsets index it)
  int64_t cls_max = 0;                  // largest class count of a level (sizes cls_state / cls_dead)
  bool cls_two = false;                 // more than one level: both ping-pong buffers
  const int32_t *d_cls_tab = nullptr;   // ... on the device: inside `world` (staged with the arena) or d_cls (re-planned at launch)
  dev_buf<int32_t> d_cls, cls_dead[2];
  dev_buf<sfw_cls_agent> cls_state[2];  // ping-pong between levels
  std::vector<int> prefix_env;          // SFW_PREFIX: empty automatic, {0} off, else forced split steps
  const void *shared_cols = nullptr;    // axis_classes of the angular targets lent by sfw_multi_score_grid for the stage in
                                        // progress (every rank has the same column axis), else null

  // per-sample outputs + per-chunk table
  dev_buf<int32_t> status, coll_step;
  dev_buf<double> base_cost, costs;  // costs: T doubles followed by the sfw_sel record (one D2H fetches both)
  dev_buf<sfw_unit> ptab, cs_tab;   // the K1 tables (sfw_device.h sfw_unit), sized for table_chunk samples per step row
  int64_t table_chunk = 0;          // samples per chunk the tables of the staged grid hold
  dev_buf<int16_t> fcode;
  dev_buf<sfw_sel> partials;
  sfw_sel *d_sel = nullptr;
  dev_buf<double> points;
  dev_buf<int32_t> n_points;
  dev_buf<unsigned long long> clock;  // sfw_launch.clock_probe (timing only)
  dev_buf<char> one_out;  // sfw_score_one: cost | n_points | coll_step | points, contiguous -> one D2H
  // scratch outputs of sfw_grid_points_batch's K1 re-run (kept across calls: hipMalloc/hipFree synchronise the device)
  dev_buf<int32_t> pts_status, pts_coll;
  dev_buf<double> pts_base, pts_costs;
  dev_buf<sfw_unit> pts_ptab, pts_cs;
  dev_buf<int16_t> pts_fcode;
  // sfw_set_params bumps params_epoch; the table sizes and the shared-prefix plan carry the epoch they were made for
  uint64_t params_epoch = 1, plan_epoch = 0;
  size_t table_budget_bytes = size_t(8) << 30;  // K1->K2 per-step tables per chunk (of 288 GB): BASELINE cfg4 runs in one chunk
  pinned_buf pin_map, pin_world, pin_out, pin_cls;
  pinned_buf pin_one;     // sfw_score_one's outputs where the one-launch kernel writes them (cost | n_points | coll_step | points)
  pinned_buf pin_mirror;  // costs + selection record as the selection kernels of the last launch leave them (device writes)
};

namespace {

int fail(sfw_handle h, int code, const std::string &msg) {
  if (h) h->err = msg;
  return code;
}
int hip_fail(sfw_handle h, hipError_t e, const char *what) {
  return fail(h, SFW_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define SFW_HIP(h, call)                                   \
  do {                                                     \
    hipError_t e_ = (call);                                \
    if (e_ != hipSuccess) return hip_fail((h), e_, #call); \
  } while (0)

// K2 through the kernels the precision mode names (SFW_PRECISION_F64_STRICT: the build with the longer polynomials)
hipError_t launch_social_of(const sfw_launch &L, hipStream_t stream, const sfw_split_streams *sp = nullptr) {
  return L.p.precision == SFW_PRECISION_F64_STRICT ? sfw_launch_social_strict(L, stream, sp) : sfw_launch_social(L, stream, sp);
}

// SFW_DEVICE_CUS in the environment: pretend a device of that many compute units (planning heuristics only)
int env_device_cus(int fallback) {
  if (const char *b = std::getenv("SFW_DEVICE_CUS")) {
    const long v = std::atol(b);
    if (v >= 1 && v <= 4096) return static_cast<int>(v);
  }
  return fallback;
}

bool all_finite(const double *v, size_t n) {
  for (size_t i = 0; i < n; ++i)
    if (!std::isfinite(v[i])) return false;
  return true;
}

// Wait for the handle's stream.  hipStreamSynchronize parks the thread on an interrupt once its short active wait is over,
// and waking up costs 10-20 us — of a 100 us control cycle or a 600 us grid; the blocking call polls the stream instead
// (one core busy for the duration of the launch, as a caller asking hipDeviceScheduleSpin would have it) for up to spin_us,
// then blocks.
hipError_t wait_stream(sfw_handle h) {
  if (h->spin_us > 0) {
    const auto t0 = std::chrono::steady_clock::now();
    bool polled = false;
    for (;;) {
      const hipError_t e = hipStreamQuery(h->stream);
      if (e != hipErrorNotReady) {
        // ("not ready" is an answer, not an error: a runtime that records it as the thread's last error must not hand it to the
        // next launch's hipGetLastError)
        if (polled && e == hipSuccess) (void)hipGetLastError();
        return e;
      }
      polled = true;
      if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > h->spin_us) break;
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
    (void)hipGetLastError();
  }
  return hipStreamSynchronize(h->stream);
}
// ... after which no upload enqueued on it is still reading its pinned staging area
void stream_is_idle(sfw_handle h) { h->pin_map.pending = h->pin_world.pending = h->pin_cls.pending = false; }

// the register-form K2 addresses a unit inside a step row of the position / velocity table by a 32-bit byte offset: a row of
// `chunk` positions and at most chunk + 2 velocities (a one-column grid) must stay below 4 GB
constexpr int64_t kRowLimit = static_cast<int64_t>((uint64_t(1) << 32) / (2 * sizeof(sfw_unit))) - 2;

int num_steps_of(const sfw_params &p) {
  int n = static_cast<int>(p.sim_time / p.sim_granularity + 0.5);  // ref :519
  return n == 0 ? 1 : n;                                           // ref :523-525
}

int check_params(sfw_handle h, const sfw_params *p) {
  if (!p) return fail(h, SFW_ERR_INVALID_ARG, "params is NULL");
  if (!(p->sim_granularity > 0) || !(p->sim_time >= 0))
    return fail(h, SFW_ERR_INVALID_ARG, "sim_time/sim_granularity out of range");
  if (p->precision != SFW_PRECISION_F64 && p->precision != SFW_PRECISION_F32 && p->precision != SFW_PRECISION_F64_STRICT)
    return fail(h, SFW_ERR_INVALID_ARG, "unknown precision");
  if (!(p->sfm_gamma > 0) || !(p->sfm_relaxation_time > 0) || !(p->sfm_force_sigma_obstacle > 0))
    return fail(h, SFW_ERR_INVALID_ARG, "sfm gamma/relaxation_time/sigma must be > 0");
  if (!(p->sfm_force_factor_social >= 0) || !(p->sfm_force_factor_obstacle >= 0))
    return fail(h, SFW_ERR_INVALID_ARG, "sfm_force_factor_social / _obstacle must be >= 0");
  return SFW_OK;
}

// Units per step row of the position / velocity table for chunks of `stride` samples of an nw-column grid: the positions,
// then one velocity unit per grid row a chunk can touch (it may begin and end inside a row)
int64_t table_row_units(int64_t stride, int nw) { return stride + stride / (nw > 0 ? nw : 1) + 2; }
// bytes of the K1 tables per sample and step (position unit + footprint code; the per-row / per-column units are noise)
constexpr size_t kTableBytesPerSampleStep = sizeof(sfw_unit) + sizeof(int16_t);

// Fill the launch descriptor for samples [begin, begin+count).
void fill_launch(sfw_handle h, sfw_launch &L, int64_t begin, int64_t count, int64_t stride) {
  std::memset(&L, 0, sizeof(L));
  L.p = h->params;
  L.S = num_steps_of(h->params);
  L.dt = h->params.sim_time / L.S;  // ref :527
  L.rs = h->rs;
  L.ga = h->ga;
  L.vy_samp = h->vy_samp;
  L.skip_zero_sample = h->skip_zero;
  L.linvels = h->d_linvels;
  L.angvels = h->d_angvels;
  L.nv = h->nv;
  L.nw = h->nw;
  L.chunk_begin = begin;
  L.chunk_count = count;
  L.cells = h->d_cells;
  L.size_x = h->size_x;
  L.size_y = h->size_y;
  L.origin_x = h->origin_x;
  L.origin_y = h->origin_y;
  L.resolution = h->resolution;
  L.footprint = h->d_footprint;
  L.K = h->st_K;
  L.agent_pos = h->d_agent_pos;
  L.agent_vel = h->d_agent_vel;
  L.agent_c = h->d_agent_c;
  L.agent_rest = h->d_agent_rest;
  L.pin_rest = h->d_pin_rest;
  L.A = h->st_A;
  L.obstacles = h->d_obstacles;
  L.O = h->st_O;
  L.agent_grp = h->d_agent_grp;
  L.grp_off = h->d_grp_off;
  L.grp_mem = h->d_grp_mem;
  L.NG = h->st_NG;
  L.n_grp_mem = h->st_n_grp_mem;
  L.n_cu = h->n_cu;
  L.n_xcd = h->n_xcd;
  sfw_derive(L);
  if (h->obs_tasks_force >= 0 && L.O > 0) L.k.obs_tasks = h->obs_tasks_force;
  L.pair_tab = h->pair_tab.p;
  L.status = h->status.p;
  L.base_cost = h->base_cost.p;
  L.costs = h->costs.p;
  L.coll_step = h->coll_step.p;
  L.ptab = h->ptab.p;
  L.cs_tab = h->cs_tab.p;
  L.row_units = table_row_units(stride, h->nw);
  L.fcode = h->fcode.p;
  L.rstep_stride = stride;
  L.points = nullptr;
  L.n_points = nullptr;
  L.phase = SFW_PHASE_WHOLE;
  L.step_begin = 0;
  L.step_end = L.S;
  L.resume = 0;
  L.k2_form = h->k2_form;
  L.index_base = h->index_base;
  L.cycle_counter = h->cycle_counter.p;
  L.sel_out = h->d_sel;
  L.costs_host = nullptr;
  L.sel_host = nullptr;
  L.arena_host = nullptr;
  L.arena_dev = nullptr;
  L.arena_bytes = 0;
}

// ---- shared-prefix plan ---------------------------------------------------
// reference sfw_planner.hpp:457-463, the recurrence K1a runs on the device (same IEEE operations)
double new_velocity_host(double vg, double vi, double a_max, double dt) {
  if ((vg - vi) >= 0) return std::fmin(vg, vi + a_max * dt);
  return std::fmax(vg, vi - a_max * dt);
}

// Classes of the samples' velocity sequences after 1..max_p steps: cls[p-1][i] is the class of
// sample value i when the first p velocities are compared, counts[p-1] the number of classes.
// Stops early once every value is its own class (nothing left to share from there on): cls/counts
// may hold fewer than max_p levels.  new_velocity is monotone in the target (and in the previous
// velocity), hence so is every v_p, and the samples that share a sequence are CONTIGUOUS when ordered
// by target — one sort up front, then a run-length pass per level.  (Were they not, the result would
// only be more classes, never a wrong merge: two samples share a class only if all their compared
// velocities are bit-equal.)  This is the GENERIC form — every sample walks the recurrence —, the
// definition the closed form below (axis_classes::build_fast) is tested against
// (tests/test_prefix_plan.py) and the fallback for the inputs that form does not take.
void velocity_classes(const std::vector<double> &targets, double v0, double a_max, double dt, int max_p,
                      std::vector<std::vector<int32_t>> &cls, std::vector<int32_t> &counts) {
  const size_t n = targets.size();
  std::vector<int32_t> order(n);
  for (size_t i = 0; i < n; ++i) order[i] = static_cast<int32_t>(i);
  std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
    return targets[static_cast<size_t>(a)] != targets[static_cast<size_t>(b)] ? targets[static_cast<size_t>(a)] < targets[static_cast<size_t>(b)]
                                                                                : a < b;
  });
  std::vector<double> t(n), v(n, v0);
  for (size_t k = 0; k < n; ++k) t[k] = targets[static_cast<size_t>(order[k])];
  std::vector<int32_t> cur(n, 0), nxt(n);
  cls.clear();
  counts.clear();
  for (int p = 0; p < max_p; ++p) {
    int32_t id = -1;
    uint64_t prev_bits = 0;
    for (size_t k = 0; k < n; ++k) {
      v[k] = new_velocity_host(t[k], v[k], a_max, dt);
      uint64_t bits;
      std::memcpy(&bits, &v[k], sizeof(bits));
      if (k == 0 || cur[k] != cur[k - 1] || bits != prev_bits) ++id;
      nxt[k] = id;
      prev_bits = bits;
    }
    cur.swap(nxt);
    cls.emplace_back(n);
    std::vector<int32_t> &out = cls.back();
    for (size_t k = 0; k < n; ++k) out[static_cast<size_t>(order[k])] = cur[k];
    counts.push_back(id + 1);
    if (static_cast<size_t>(id + 1) == n) break;
  }
}

// Per-step cost of a launch of `items` work items, in units of one step of a SIMD-filling launch's wave
// round: below one wave per issue slot of the GPU the FP64 dependency chains are exposed and a step takes
// about 0.45 of a full round whatever the item count (cfg2: 10 us against 22 us), above it time is
// proportional to the waves.
double step_cost(double items, double items_per_wave, int cus) {
  const double capacity = 4.0 * cus * 5.5;  // waves resident at once: 4 SIMDs per CU x 5-6 waves (1024 SIMDs on a whole MI355X)
  const double x = items / items_per_wave / capacity;
  return x <= 1.0 ? 0.45 + 0.55 * x : x;
}

// Classes of ONE axis of the grid (velocity_classes of the linear or of the angular targets).
//
// The planning sits on the blocking call's critical path (the GPU has nothing but the pose rollout to do until the class
// tables arrive: profiles/r05_step_timeline_cfg2.txt), and the generic form above walks n x max_p recurrence steps and
// allocates a class vector per step count — 26 us for cfg2's 128 + 128 targets, 52 us for the target configuration's.
// Closed form (build_fast): with s = a_max dt > 0 every sample that has not reached its target yet carries the SAME
// velocity — u_p = fl(u_{p-1} + s) above the start velocity, d_p = fl(d_{p-1} - s) below it — and a sample that has
// reached its target keeps it:  v_p(t) = t - v0 >= 0 ? fmin(t, u_p) : fmax(t, d_p)  (induction over new_velocity_host's
// two branches; the reached sample takes the first branch with vg - vi == 0 and fmin(t, fl(t + s)) = t).  In target
// order the classes of level p are therefore
//     [0, lo_p) all t <= d_p : ONE class | [lo_p, hi_p) reached: one class per distinct target | [hi_p, n) all t >= u_p : ONE class
// with lo_p falling and hi_p rising in p: two pointers, O(n + max_p) for all the class COUNTS (what the level choice needs),
// and the class array of a level is written only when a chosen level asks for it (level()).  Taken only when the model's
// premises hold — s > 0, u and d really move at every step (no stall at an ulp), no -0.0 among the targets (the generic form
// compares bits) —, otherwise the generic form runs.  Same classes, same numbering (in target order) as the generic form.
struct axis_classes {
  std::vector<int32_t> n;               // class counts per number of compared steps (index p - 1)
  // what they were derived from (a shared copy is only reused for exactly these inputs)
  std::vector<double> targets;
  double v0 = 0, a_max = 0, dt = 0;
  int max_p = 0;
  bool made_for(const std::vector<double> &t, double v0_, double a_, double dt_, int mp) const {
    return v0 == v0_ && a_max == a_ && dt == dt_ && max_p == mp && targets == t;
  }
  // class of every value when the first p velocities are compared (p >= 1; past the last computed level: that level, where
  // every value is its own class)
  const std::vector<int32_t> &level(int p) const {
    const size_t l = std::min<size_t>(static_cast<size_t>(p), n.size()) - 1;
    if (!fast) return c[l];
    if (c[l].empty()) write_level(l);
    return c[l];
  }
  // (a copy lent to several threads — sfw_multi_score_grid's column axis — is completed first: level() then only reads)
  void complete() const {
    if (fast)
      for (size_t l = 0; l < n.size(); ++l)
        if (c[l].empty()) write_level(l);
  }
  bool fast = false;

  void build(bool allow_fast) {
    fast = allow_fast && build_fast();
    if (!fast) {
      c.clear();
      velocity_classes(targets, v0, a_max, dt, max_p, c, n);
    }
  }

 private:
  mutable std::vector<std::vector<int32_t>> c;  // generic: every level; fast: the levels asked for so far
  std::vector<int32_t> order;                   // fast: target order -> value index
  std::vector<int32_t> run;                     // fast: run (distinct bit pattern) of every position in target order, from 0
  std::vector<int32_t> lo, hi;                  // fast: per level
  static uint64_t bits_of(double x) {
    uint64_t b;
    std::memcpy(&b, &x, sizeof(b));
    return b;
  }
  bool build_fast() {
    const size_t cnt = targets.size();
    n.clear();
    c.clear();
    lo.clear();
    hi.clear();
    if (cnt == 0 || max_p < 1) return false;
    const double s = a_max * dt;
    if (!(s > 0.0) || !std::isfinite(s) || !std::isfinite(v0)) return false;
    for (double t : targets)
      if (!std::isfinite(t) || bits_of(t) == (uint64_t(1) << 63)) return false;
    order.resize(cnt);
    bool sorted = true;
    for (size_t i = 0; i < cnt; ++i) {
      order[i] = static_cast<int32_t>(i);
      if (i && targets[i] < targets[i - 1]) sorted = false;
    }
    if (!sorted)
      std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
        return targets[static_cast<size_t>(a)] != targets[static_cast<size_t>(b)] ? targets[static_cast<size_t>(a)] < targets[static_cast<size_t>(b)]
                                                                                    : a < b;
      });
    auto t_at = [&](size_t k) { return targets[static_cast<size_t>(order[k])]; };
    run.resize(cnt);
    int32_t r = 0;
    for (size_t k = 0; k < cnt; ++k) {
      if (k && bits_of(t_at(k)) != bits_of(t_at(k - 1))) ++r;
      run[k] = r;
    }
    size_t m = 0;  // first position of the side that accelerates (vg - vi >= 0 at the start velocity)
    while (m < cnt && !((t_at(m) - v0) >= 0)) ++m;
    double u = v0, d = v0;
    size_t l = m, h = m;  // lo_p / hi_p: the reached zone [l, h) grows from the start velocity outwards
    for (int p = 0; p < max_p; ++p) {
      // the recurrence's own expressions (new_velocity_host): vi + a_max * dt, vi - a_max * dt
      const double un = u + a_max * dt, dn = d - a_max * dt;
      if (!(un > u) || !(dn < d) || !std::isfinite(un) || !std::isfinite(dn)) return false;  // stalled at an ulp: generic form
      u = un;
      d = dn;
      while (h < cnt && t_at(h) < u) ++h;       // t < u_p: reached (fmin(t, u_p) = t); t >= u_p: carries u_p
      while (l > 0 && t_at(l - 1) > d) --l;     // t > d_p: reached; t <= d_p: carries d_p
      lo.push_back(static_cast<int32_t>(l));
      hi.push_back(static_cast<int32_t>(h));
      // classes: the group below (if any) + the distinct targets in [l, h) + the group above (if any).  A target equal to
      // u_p (d_p) carries the group's value: it sits in the group (h stops in front of it)
      int32_t classes = (l > 0 ? 1 : 0) + (h < cnt ? 1 : 0);
      if (h > l) classes += run[h - 1] - run[l] + 1;
      n.push_back(classes);
      if (static_cast<size_t>(classes) == cnt) break;
    }
    c.assign(n.size(), std::vector<int32_t>());
    return true;
  }
  void write_level(size_t lv) const {
    const size_t cnt = targets.size();
    const size_t l = static_cast<size_t>(lo[lv]), h = static_cast<size_t>(hi[lv]);
    std::vector<int32_t> &out = c[lv];
    out.resize(cnt);
    const int32_t below = l > 0 ? 1 : 0;
    const int32_t mid = h > l ? run[h - 1] - run[l] + 1 : 0;
    for (size_t k = 0; k < cnt; ++k) {
      int32_t id;
      if (k < l) id = 0;
      else if (k < h) id = below + (run[k] - run[l]);
      else id = below + mid;
      out[static_cast<size_t>(order[k])] = id;
    }
  }
};
// SFW_PLAN_GENERIC=1 in the environment: always the generic form (A/B and tests)
bool plan_fast_allowed() {
  static const bool on = [] {
    const char *e = std::getenv("SFW_PLAN_GENERIC");
    return !(e && e[0] == '1');
  }();
  return on;
}
axis_classes classes_of_axis(const std::vector<double> &targets, double v0, double a_max, double dt, int max_p,
                             bool allow_fast = plan_fast_allowed()) {
  axis_classes a;
  a.targets = targets;
  a.v0 = v0;
  a.a_max = a_max;
  a.dt = dt;
  a.max_p = max_p;
  a.build(allow_fast);
  return a;
}
int prefix_max_p(int S) { return std::min(S - 1, 48); }

// Everything plan_prefix derives from the staged sample vectors alone (no device involved).  The column axis may be
// borrowed (`shared_cols`): every rank of a multi-device grid has the same angular targets.
struct prefix_classes {
  axis_classes rows, own_cols;
  const axis_classes *cols = nullptr;
  int max_p = 0;
  // a level past the last computed one has every value in its own class
  int32_t n_rows_at(int p) const { return rows.n[std::min<size_t>(static_cast<size_t>(p), rows.n.size()) - 1]; }
  int32_t n_cols_at(int p) const { return cols->n[std::min<size_t>(static_cast<size_t>(p), cols->n.size()) - 1]; }
};

void classes_of_grid(prefix_classes &pc, const std::vector<double> &lin, const std::vector<double> &ang, double vx0,
                     double vth0, double acc_x, double acc_theta, double dt, int S, const axis_classes *shared_cols = nullptr) {
  pc.max_p = prefix_max_p(S);
  pc.rows = classes_of_axis(lin, vx0, acc_x, dt, pc.max_p);
  if (shared_cols && shared_cols->made_for(ang, vth0, acc_theta, dt, pc.max_p)) {
    pc.cols = shared_cols;
  } else {
    pc.own_cols = classes_of_axis(ang, vth0, acc_theta, dt, pc.max_p);
    pc.cols = &pc.own_cols;
  }
}

// The levels' end steps.  A level ending at step p costs (p - q) steps over classes(p) items plus a
// launch and the class records (0.4 to 1.0 of a round); the suffix costs (S - p) steps over all
// samples.  Dynamic programme over the end step of the last level; empty when sharing does not pay.
std::vector<int> choose_levels(const prefix_classes &pc, int64_t T, int S, double samples_per_wave, int cus) {
  std::vector<int> steps;
  const double full = step_cost(static_cast<double>(T), samples_per_wave, cus);
  const int last_p = std::min<int>(pc.max_p, static_cast<int>(std::max(pc.rows.n.size(), pc.cols->n.size())));
  std::vector<double> best(static_cast<size_t>(last_p) + 1, 0.0);
  std::vector<int> from(static_cast<size_t>(last_p) + 1, 0);
  double best_total = S * full;
  int best_end = 0;
  for (int p = 1; p <= last_p; ++p) {
    const double c = step_cost(static_cast<double>(pc.n_rows_at(p)) * pc.n_cols_at(p), samples_per_wave, cus);
    // per extra launch: dispatch + class records for an under-filled level; a level that fills the GPU
    // also pays its ramp-up and tail (measured: 0.4 / 1.0 pick the fastest plans at cfg2 / target)
    const double launch_cost = c < 1.0 ? 0.4 : 1.0;
    best[p] = 1e300;
    for (int q = 0; q < p; ++q) {
      const double v = best[q] + (p - q) * c + launch_cost;
      if (v < best[p]) { best[p] = v; from[p] = q; }
    }
    const double total = best[p] + (S - p) * full;
    if (total < best_total) { best_total = total; best_end = p; }
  }
  if (best_total > 0.97 * S * full) return steps;  // not worth the extra launches
  for (int p = best_end; p > 0; p = from[p]) steps.push_back(p);
  std::reverse(steps.begin(), steps.end());
  return steps;
}

// Row blocks of (about) equal PLANNED work for R ranks.  A block's work is its class-steps through the shared-prefix
// levels plus its samples' remaining steps, and the share of steps the tree saves differs along the row axis (rows are
// ordered by target velocity: cfg5 cut into 8 equal blocks integrates 64 %..76 % of its steps per block).  With the
// levels of the whole grid's plan every row gets a weight — for each level, (steps of the level) x (column classes) if
// the row opens a new row class at that level, plus nw x (S - split) for its samples' own steps — and the cuts go where
// the running sum passes r/R of the total.  Equal row counts when nothing is shared.  Each rank still makes its own plan
// for its block; costs do not depend on the cut.
void plan_row_blocks(const std::vector<double> &lin, const std::vector<double> &ang, double vx0, double vth0, double acc_x,
                     double acc_theta, double dt, int S, int A, int R, int form, int cus, std::vector<int32_t> &row0) {
  const int64_t nv = static_cast<int64_t>(lin.size()), nw = static_cast<int64_t>(ang.size());
  row0.assign(static_cast<size_t>(R) + 1, 0);
  for (int r = 0; r <= R; ++r) row0[static_cast<size_t>(r)] = static_cast<int32_t>((static_cast<int64_t>(r) * nv) / R);
  if (R <= 1 || A < 2 || S < 2 || (nv / R) * nw < 4096) return;
  prefix_classes pc;
  classes_of_grid(pc, lin, ang, vx0, vth0, acc_x, acc_theta, dt, S);
  // the levels of the WHOLE grid's plan (its class counts are the whole grid's): the blocks' own plans differ in detail,
  // the relative weights along the row axis do not
  const std::vector<int> steps = choose_levels(pc, nv * nw, S, static_cast<double>(sfw_samples_per_wave(A, (nv / R) * nw, form, cus)), cus);
  if (steps.empty()) return;
  std::vector<double> w(static_cast<size_t>(nv), static_cast<double>(nw) * (S - steps.back()));
  int prev = 0;
  for (int p : steps) {
    const std::vector<int32_t> &rc = pc.rows.level(p);
    const double per_class = static_cast<double>(pc.n_cols_at(p)) * (p - prev);
    for (int64_t i = 0; i < nv; ++i)
      if (i == 0 || rc[static_cast<size_t>(i)] != rc[static_cast<size_t>(i - 1)]) w[static_cast<size_t>(i)] += per_class;
    prev = p;
  }
  double total = 0.0;
  for (double v : w) total += v;
  double run = 0.0;
  int r = 1;
  for (int64_t i = 0; i < nv && r < R; ++i) {
    run += w[static_cast<size_t>(i)];
    while (r < R && run >= total * r / R) row0[static_cast<size_t>(r++)] = static_cast<int32_t>(i + 1);
  }
  for (; r < R; ++r) row0[static_cast<size_t>(r)] = static_cast<int32_t>(nv);
}

// Decide the levels of the staged grid's shared-prefix tree and lay the class tables out per chunk of
// whole rows.
int plan_prefix(sfw_handle h, int64_t chunk, int S) {
  h->prefix_steps.clear();
  h->prefix_chunks.clear();
  h->cls_ints.clear();
  h->cls_max = 0;
  h->prefix_class_steps = h->prefix_last_classes = 0;
  const int64_t T = static_cast<int64_t>(h->nv) * h->nw;
  const bool forced = !h->prefix_env.empty();
  if ((forced && h->prefix_env[0] == 0) || h->st_A < 2 || S < 2 || h->vy_samp != 0.0) return SFW_OK;
  if (!forced && T < 4096) return SFW_OK;  // the GPU is not full: extra launches cost more than they save
  const int64_t rows_per_chunk = chunk / h->nw;
  if (rows_per_chunk < 1) return SFW_OK;  // a single row does not fit the table budget: no sharing
  prefix_classes pc;
  classes_of_grid(pc, h->h_lin, h->h_ang, h->rs.vx, h->rs.vtheta, h->ga.acc_x, h->ga.acc_theta, h->params.sim_time / S, S,
                  static_cast<const axis_classes *>(h->shared_cols));
  const axis_classes &rc = pc.rows, &cc = *pc.cols;
  const std::vector<int32_t> &nr = pc.rows.n, &nc = pc.cols->n;
  auto level = [](const axis_classes &c, int p) -> const std::vector<int32_t> & { return c.level(p); };
  auto count_at = [](const std::vector<int32_t> &n, int p) { return n[std::min<size_t>(static_cast<size_t>(p), n.size()) - 1]; };
  std::vector<int> steps;
  if (forced) {
    for (int p : h->prefix_env)
      if (p >= 1 && p <= pc.max_p) steps.push_back(p);
  } else {
    steps = choose_levels(pc, T, S, static_cast<double>(sfw_samples_per_wave(h->st_A, std::min<int64_t>(T, rows_per_chunk * h->nw), h->k2_form, h->n_cu)), h->n_cu);
  }
  if (steps.empty()) return SFW_OK;
  const size_t n_lv = steps.size();

  // ---- tables: columns (chunk-independent), then rows per chunk
  std::vector<int32_t> ints;
  ints.swap(h->cls_ints);  // (reuse the last stage's allocation)
  ints.clear();
  ints.reserve(static_cast<size_t>(2 * n_lv + 2) * static_cast<size_t>(h->nv + h->nw));
  auto append = [&](const std::vector<int32_t> &v) {
    const size_t o = ints.size();
    ints.insert(ints.end(), v.begin(), v.end());
    return o;
  };
  // representatives (first member) of the classes of `cls` over the index range [i0, i1); ids are
  // relabelled in order of first appearance; `local` receives the relabelled class of every index
  std::vector<int32_t> ids;  // class id -> relabelled id of the current range (ids are < the axis length)
  auto relabel = [&ids](const std::vector<int32_t> &cls, int64_t i0, int64_t i1, std::vector<int32_t> &local,
                        std::vector<int32_t> &rep) {
    ids.assign(cls.size(), -1);
    local.clear();
    rep.clear();
    local.reserve(static_cast<size_t>(i1 - i0));  // (no growth by doubling: this runs on the blocking call's critical path)
    rep.reserve(static_cast<size_t>(i1 - i0));
    for (int64_t i = i0; i < i1; ++i) {
      int32_t &id = ids[static_cast<size_t>(cls[static_cast<size_t>(i)])];
      if (id < 0) {
        id = static_cast<int32_t>(rep.size());
        rep.push_back(static_cast<int32_t>(i - i0));
      }
      local.push_back(id);
    }
  };
  struct axis_level { std::vector<int32_t> local, rep, src; };
  // levels of one axis over [i0, i1): local classes, representatives, parent classes
  auto axis_levels = [&](const axis_classes &cls, int64_t i0, int64_t i1) {
    std::vector<axis_level> out(n_lv);
    for (size_t l = 0; l < n_lv; ++l) {
      relabel(level(cls, steps[l]), i0, i1, out[l].local, out[l].rep);
      if (l > 0) {  // classes refine: the parent of a class is the previous level's class of its representative
        out[l].src.reserve(out[l].rep.size());
        for (int32_t r : out[l].rep) out[l].src.push_back(out[l - 1].local[static_cast<size_t>(r)]);
      }
    }
    return out;
  };
  const std::vector<axis_level> cols = axis_levels(cc, 0, h->nw);
  std::vector<size_t> o_col_rep(n_lv), o_col_src(n_lv);
  for (size_t l = 0; l < n_lv; ++l) {
    o_col_rep[l] = append(cols[l].rep);
    o_col_src[l] = append(cols[l].src);
  }
  h->prefix_o_col_cls = append(cols[n_lv - 1].local);
  int64_t max_cls = 0;
  for (int64_t r0 = 0; r0 < h->nv; r0 += rows_per_chunk) {
    const int64_t r1 = std::min<int64_t>(h->nv, r0 + rows_per_chunk);
    const std::vector<axis_level> rows = axis_levels(rc, r0, r1);
    sfw_planner_s::chunk_plan cp;
    for (size_t l = 0; l < n_lv; ++l) {
      sfw_planner_s::level_tables t;
      t.n_row = static_cast<int32_t>(rows[l].rep.size());
      t.n_col = static_cast<int32_t>(cols[l].rep.size());
      t.o_row_rep = append(rows[l].rep);
      t.o_row_src = append(rows[l].src);
      t.o_col_rep = o_col_rep[l];
      t.o_col_src = o_col_src[l];
      cp.lv.push_back(t);
      const int64_t n = static_cast<int64_t>(t.n_row) * t.n_col;
      max_cls = std::max(max_cls, n);
      h->prefix_class_steps += n * (steps[l] - (l ? steps[l - 1] : 0));
      if (l + 1 == n_lv) h->prefix_last_classes += n;
    }
    cp.o_row_cls = append(rows[n_lv - 1].local);
    h->prefix_chunks.push_back(cp);
  }
  // (host only: the tables reach the device with the stage's one arena copy — stage_common — or, when a launch re-plans
  // after sfw_set_params, by a copy of their own: upload_class_tables)
  h->cls_ints.swap(ints);
  h->cls_max = max_cls;
  h->cls_two = n_lv > 1;
  if (std::getenv("SFW_DEBUG_PLAN")) {
    std::fprintf(stderr, "[sfw] shared prefix, %d x %d samples, S=%d, %zu chunk(s):", h->nv, h->nw, S, h->prefix_chunks.size());
    for (size_t l = 0; l < n_lv; ++l)
      std::fprintf(stderr, " [%d,%d) %d x %d;", l ? steps[l - 1] : 0, steps[l], count_at(nr, steps[l]), count_at(nc, steps[l]));
    std::fprintf(stderr, " [%d,%d) samples\n", steps.back(), S);
  }
  h->prefix_steps = steps;
  h->prefix_S = S;
  h->prefix_chunk = rows_per_chunk * h->nw;
  return SFW_OK;
}

// Pairs with w x diff == 0 exactly, w = v_i - v_j: relative rest (two standing people, a stopped robot next to a
// standing person) and motion exactly along the connecting line (two people on one grid-aligned line walking towards
// or away from each other).  There the model's interaction angle theta is mathematically 0 or +-pi and its angular
// term -sign(theta) exp(-d/B - (n B theta)^2) leftNormal(Ihat) is discontinuous.  The kernels take sign(theta) from
// w x diff, exactly 0 here: the sign of a zero (their exact zero in the SFW_SIGN_OF_ZERO=0 build).  lightsfm instead forms theta as the difference of two
// atan2 — of vectors equal up to rounding when I = lambda w + dhat points along dhat (sign(theta) = -1, 0 or +1 by the
// rounding of the HOST's libm: a full-magnitude lateral force on an ordinary scene), of opposite vectors when the pair
// separates faster than 1/lambda (theta = +-pi by the sign of a zero).  Such a configuration can only come from the state
// the caller hands over (from step 1 on every agent has been pushed by a different force — the reference's own lateral
// term breaks the alignment), and that state is the same for every sample, so this term is evaluated here, once per
// stage, with the same expression sequence and the same libm the reference would run on this host (SURVEY.md Appendix
// A: computeSocialForce), and added to the agents' starting forces.  Bug-compatibility with the restated text of an
// unpinned dependency, not a CPU path: no sample is scored on the host.  out: A x (fx, fy).
void rest_forces(const sfw_params &p, const std::vector<std::pair<int32_t, int32_t>> &pairs, const double *pos, const double *vel,
                 int A, double *out) {
  for (int i = 0; i < 2 * A; ++i) out[i] = 0.0;
  for (const auto &pr : pairs) {
    const int i = pr.first, j = pr.second;
    const double dx = pos[2 * j] - pos[2 * i], dy = pos[2 * j + 1] - pos[2 * i + 1];  // diff = other - me
    const double dn = std::sqrt(dx * dx + dy * dy);
    if (!(dn > 0.0)) continue;
    const double ux = dx / dn, uy = dy / dn;                                          // diffDirection
    const double wx = vel[2 * i] - vel[2 * j], wy = vel[2 * i + 1] - vel[2 * j + 1];  // velDiff (= 0)
    const double ix = p.sfm_lambda * wx + ux, iy = p.sfm_lambda * wy + uy;            // interactionVector
    const double il = std::sqrt(ix * ix + iy * iy);
    if (!(il > 0.0)) continue;  // I = 0 (apart at exactly 1/lambda along the line): the reference divides by zero there; the kernels' term is 0
    const double ex = ix / il, ey = iy / il;                                          // interactionDirection
    double theta = std::atan2(uy, ux) - std::atan2(ey, ex);                           // angleTo, kept in (-pi, pi]
    while (theta <= -M_PI) theta += 2.0 * M_PI;
    while (theta > M_PI) theta -= 2.0 * M_PI;
    const double B = p.sfm_gamma * il;
    if (theta != 0.0) {
      const double sq = p.sfm_n * B * theta;
      const double fa = -(theta > 0.0 ? 1.0 : -1.0) * std::exp(-dn / B - sq * sq);
      out[2 * i] += p.sfm_force_factor_social * fa * -ey;                             // leftNormal = (-y, x)
      out[2 * i + 1] += p.sfm_force_factor_social * fa * ex;
    }
#if SFW_SIGN_OF_ZERO
    // ... minus what the kernels themselves apply to such a pair: their |theta| is 0 or pi by the side I points to, their
    // exponential a 1e-12 polynomial (the difference to this libm one stays as a 1e-12 share of the term), their sign the
    // sign bit of w x diff = fma(wx, dy, -(wy dx)) — of a ZERO here, which depends on the order the pair is taken in
    // (v - v is +0 either way while diff changes sign).  Both K2 organisations evaluate an unordered pair once, as (a, b)
    // with b = (a + row + 1) mod A, row < A / 2 (the last row of an even A only for a < A / 2), and give b the negative.
    const int fwd = ((j - i) % A + A) % A, rows = A / 2;
    const bool i_first = fwd < rows || (fwd == rows && ((A & 1) ? true : i < rows));
    auto rev = [](double x) { return -x + 0.0; };  // b - a from a - b: the exact negative, a zero difference stays +0
    const double cw_k = i_first ? std::fma(wx, dy, -(wy * dx)) : std::fma(rev(wx), rev(dy), -(rev(wy) * rev(dx)));
    const double tk = (ix * ux + iy * uy) < 0.0 ? M_PI : 0.0, sk = p.sfm_n * B * tk;
    const double s_k = std::signbit(cw_k) ? -1.0 : 1.0;
    const double fk = -s_k * std::exp(-dn / B - sk * sk);
    out[2 * i] -= p.sfm_force_factor_social * fk * -ey;
    out[2 * i + 1] -= p.sfm_force_factor_social * fk * ex;
#endif
  }
}

// A person that can never move (desired_velocity == 0: lightsfm's speed clamp sets its velocity to 0 in every updatePosition)
// next to a robot that stands still for the whole rollout (twist 0 at hand-over, sample (0, w): it may turn, its agent
// velocity — the robot-LOCAL twist, ref :600-604 — and its position stay what they are) is at exact relative rest at EVERY
// step, not only in the handed-over state: lightsfm's sign(theta) is the rounding noise of two atan2 there (0 for most
// geometries, +-1 — a full-magnitude lateral term in Wr and Wp — for the others), the kernels' is 0.  The geometry
// of such a pair is the same at every step of every such sample — robot at (rs.x, rs.y), person where it stands — so, like
// rest_forces for the handed-over state, the host evaluates the reference's expression once per stage:
//   out[0..1] = the robot position the table holds for (the kernels compare a step's robot record with it, bit for bit)
//   out[2..3] = sum over the pinned people k of the LATERAL part of the force k exerts on the robot (joins the robot's social
//               force of the step AFTER every step whose post-step robot record is that position at velocity 0)
//   out[4+i]  = |force the robot exerts on person i| with the lateral part minus the same without it — what the kernels' Wp
//               of person i lacks at every such step; 0 for a person that can move or that carries the robot's id (ref :692).
// Returns false when no person is pinned (no table).  A robot that BRAKES to a stop during the rollout stands at a position
// the device's own pose rollout produced; the rounding noise at that position is not reproduced.
bool pinned_rest_table(const sfw_params &p, const sfw_robot_state &rs, const double *pos, const sfw_agent_const *cst, int A,
                       double *out) {
#if SFW_SIGN_OF_ZERO
  // (round 3's A/B build: its kernels apply a sign-of-zero lateral term of their own to every pair at exact rest, at every
  // step, which this table does not take back out — ADVICE r5.  No table there: that build is for timing comparisons.)
  (void)p; (void)rs; (void)pos; (void)cst; (void)A; (void)out;
  return false;
#endif
  bool any = false;
  for (int i = 1; i < A; ++i) any |= (cst[i].desired_velocity == 0.0);
  if (!any) return false;
  out[0] = rs.x;
  out[1] = rs.y;
  out[2] = out[3] = 0.0;
  for (int i = 0; i < A; ++i) out[4 + i] = 0.0;
  // force exerted on `me` by `other`, both at velocity 0 (lightsfm computeSocialForce, one term):
  // the whole term in (fx, fy), its angular part in (lx, ly)
  auto pair_at_rest = [&](double mx, double my, double ox, double oy, double &fx, double &fy, double &lx, double &ly) {
    fx = fy = lx = ly = 0.0;
    const double dx = ox - mx, dy = oy - my;
    const double dn = std::sqrt(dx * dx + dy * dy);
    if (!(dn > 0.0)) return;
    const double ux = dx / dn, uy = dy / dn;                  // diffDirection
    const double ix = p.sfm_lambda * 0.0 + ux, iy = p.sfm_lambda * 0.0 + uy;  // interactionVector, velDiff = 0
    const double il = std::sqrt(ix * ix + iy * iy);
    if (!(il > 0.0)) return;
    const double ex = ix / il, ey = iy / il;                  // interactionDirection
    double theta = std::atan2(uy, ux) - std::atan2(ey, ex);   // angleTo, kept in (-pi, pi]
    while (theta <= -M_PI) theta += 2.0 * M_PI;
    while (theta > M_PI) theta -= 2.0 * M_PI;
    const double B = p.sfm_gamma * il;
    const double sv = p.sfm_n_prime * B * theta, sa = p.sfm_n * B * theta;
    const double fv = -std::exp(-dn / B - sv * sv);
    const double sg = theta > 0.0 ? 1.0 : theta < 0.0 ? -1.0 : 0.0;
    const double fa = -sg * std::exp(-dn / B - sa * sa);
    lx = p.sfm_force_factor_social * (fa * -ey);              // leftNormal = (-y, x)
    ly = p.sfm_force_factor_social * (fa * ex);
    fx = p.sfm_force_factor_social * (fv * ex + fa * -ey);
    fy = p.sfm_force_factor_social * (fv * ey + fa * ex);
  };
  for (int i = 1; i < A; ++i) {
    if (cst[i].desired_velocity != 0.0) continue;
    double fx, fy, lx, ly;
    pair_at_rest(rs.x, rs.y, pos[2 * i], pos[2 * i + 1], fx, fy, lx, ly);  // on the robot, by person i
    out[2] += lx;
    out[3] += ly;
    pair_at_rest(pos[2 * i], pos[2 * i + 1], rs.x, rs.y, fx, fy, lx, ly);  // on person i, by the robot
    if (cst[i].id != cst[0].id) out[4 + i] = std::sqrt(fx * fx + fy * fy) - std::sqrt((fx - lx) * (fx - lx) + (fy - ly) * (fy - ly));
  }
  return true;
}

// Everything of a stage that depends on sfw_params: the K1->K2 tables ([S][chunk] records, chunk bounded
// by the table budget) and the shared-prefix plan (classes of the velocity sequences under dt = sim_time/S).
// The one configuration whose reference result the kernels do not reproduce (sfw_plan_info.rest_noise_unreproduced, DESIGN.md
// §10): a person that can never move next to a robot that moves now and brakes to a stop inside some sample's rollout.
// Decided from what the stage uploaded, with the device's own velocity recurrence (plain IEEE operations).
bool rest_noise_unreproduced(sfw_handle h) {
  if (!h->staged || h->st_A < 2 || (h->rs.vx == 0.0 && h->rs.vy == 0.0)) return false;  // (standing from the start: reproduced)
  const sfw_agent_const *cst = reinterpret_cast<const sfw_agent_const *>(h->h_agents.data() + h->ao_cst);
  bool pinned = false;
  for (int i = 1; i < h->st_A && i < h->A; ++i) pinned |= (cst[i].desired_velocity == 0.0);
  if (!pinned || h->vy_samp != 0.0) return false;
  bool zero_row = false;
  for (double v : h->h_lin) zero_row |= (v == 0.0);
  if (!zero_row) return false;
  const int S = num_steps_of(h->params);
  const double dt = h->params.sim_time / S;
  double vx = h->rs.vx, vy = h->rs.vy;
  for (int i = 0; i + 1 < S; ++i) {  // at rest AFTER step i < S - 1: at least one step is integrated from the rest pose
    vx = new_velocity_host(0.0, vx, h->ga.acc_x, dt);
    vy = new_velocity_host(0.0, vy, h->ga.acc_y, dt);
    if (vx == 0.0 && vy == 0.0) return true;
  }
  return false;
}

// Host half: chunk size under the table budget and the shared-prefix plan (class tables in h->cls_ints).  No device call.
int64_t plan_tables_host(sfw_handle h, int *err) {
  const int64_t T = static_cast<int64_t>(h->nv) * h->nw;
  const int S = num_steps_of(h->params);
  h->early_poses = false;
  int64_t chunk = static_cast<int64_t>(
      h->table_budget_bytes / (kTableBytesPerSampleStep * S));
  if (chunk < 1024) chunk = 1024;
  if (chunk > kRowLimit) chunk = kRowLimit;
  if (chunk > T) chunk = T;
  *err = plan_prefix(h, chunk, S);
  h->plan_epoch = h->params_epoch;
  return chunk;
}
// Device half: the K1->K2 tables and the class-record buffers; with may_start_poses the pose rollout of a single-chunk grid
// is enqueued here, by the STAGE (the robot's poses depend on the sample vectors alone).
int plan_tables_device(sfw_handle h, int64_t chunk, bool may_start_poses) {
  const int64_t T = static_cast<int64_t>(h->nv) * h->nw;
  const int S = num_steps_of(h->params);
  SFW_HIP(h, h->ptab.reserve(static_cast<size_t>(table_row_units(chunk, h->nw)) * S));
  SFW_HIP(h, h->cs_tab.reserve(static_cast<size_t>(h->nw) * S));
  SFW_HIP(h, h->fcode.reserve(static_cast<size_t>(chunk) * S));
  h->table_chunk = chunk;
  if (!h->prefix_steps.empty())
    for (size_t b = 0; b < (h->cls_two ? 2u : 1u); ++b) {
      SFW_HIP(h, h->cls_dead[b].reserve(static_cast<size_t>(h->cls_max)));
      SFW_HIP(h, h->cls_state[b].reserve(static_cast<size_t>(h->cls_max) * h->st_A));
    }
  if (may_start_poses && chunk >= T) {
    sfw_launch L;
    fill_launch(h, L, 0, T, T);
    if (!sfw_rollout_is_fused(L)) {  // (the fused small-grid K1 is one launch with the costmap part, and may capture points)
      h->clock_cleared = false;
      if (h->timing) {  // the clock probe of a timed launch is cleared by the pose rollout itself (clear_clock_probe)
        SFW_HIP(h, h->clock.reserve(4));
        L.clock_probe = h->clock.p;
        h->clock_cleared = true;
      }
      if (h->timing) SFW_HIP(h, hipEventRecord(h->ev[0], h->stream));
      h->early_poses_timed = h->timing;
      if (h->arena_pending) {  // no copy was enqueued: this kernel fetches the arena, and reads its sample vectors at their host addresses
        L.arena_host = h->pin_world.p + h->arena_from;
        L.arena_dev = h->world.p + h->arena_from;
        L.arena_bytes = static_cast<uint32_t>(h->arena_bytes);
        L.linvels = reinterpret_cast<const double *>(h->pin_world.p + (reinterpret_cast<const char *>(h->d_linvels) - h->world.p));
        L.angvels = reinterpret_cast<const double *>(h->pin_world.p + (reinterpret_cast<const char *>(h->d_angvels) - h->world.p));
      }
      SFW_HIP(h, sfw_launch_rollout_poses(L, h->stream));
      if (h->arena_pending) {
        h->arena_pending = false;
        SFW_HIP(h, h->pin_world.mark(h->stream));
      }
      h->early_poses = true;
    }
  }
  return SFW_OK;
}
// The class tables by a copy of their own (a launch that re-plans after sfw_set_params: the stage's arena is on the device already)
int upload_class_tables(sfw_handle h) {
  h->d_cls_tab = nullptr;
  if (h->cls_ints.empty()) return SFW_OK;
  const size_t bytes = sizeof(int32_t) * h->cls_ints.size();
  SFW_HIP(h, h->d_cls.reserve(h->cls_ints.size()));
  SFW_HIP(h, h->pin_cls.reserve(bytes));
  std::memcpy(h->pin_cls.p, h->cls_ints.data(), bytes);
  SFW_HIP(h, hipMemcpyAsync(h->d_cls.p, h->pin_cls.p, bytes, hipMemcpyHostToDevice, h->stream));
  SFW_HIP(h, h->pin_cls.mark(h->stream));
  h->d_cls_tab = h->d_cls.p;
  return SFW_OK;
}
// Everything of a stage that depends on sfw_params, redone by a launch when sfw_set_params came in between (the reference
// re-reads its parameters every cycle, :125): a plan made for another dt would merge samples whose robot trajectories now differ.
int replan_at_launch(sfw_handle h) {
  int err = SFW_OK;
  const int64_t chunk = plan_tables_host(h, &err);
  if (err) return err;
  if (int e = plan_tables_device(h, chunk, false)) return e;
  return upload_class_tables(h);
}

// The agent / laser-point set must fit one wave's LDS allocation (160 KiB per CU).
int check_lds(sfw_handle h, int64_t items) {
  const size_t lds = sfw_social_lds_bytes(h->st_A, h->st_O, h->st_NG, h->st_n_grp_mem, items, h->k2_form, h->n_cu);
  if (h->st_A > 0 && lds > 160 * 1024)
    return fail(h, SFW_ERR_UNSUPPORTED, "agent/obstacle set does not fit the 160 KiB LDS of one CU");
  return SFW_OK;
}

constexpr size_t kMapMergeMax = size_t(64) << 10;  // costmaps up to this many cells ride in the stage's arena copy

// SFW_DEBUG_STAGE=1 in the environment: every stage / launch prints where its host time went (stderr; tuning aid)
struct host_phases {
  bool on;
  std::chrono::steady_clock::time_point t;
  std::string out;
  explicit host_phases(const char *what) {
    static const bool enabled = std::getenv("SFW_DEBUG_STAGE") != nullptr;
    on = enabled;
    if (on) {
      out = what;
      t = std::chrono::steady_clock::now();
    }
  }
  void mark(const char *name) {
    if (!on) return;
    const auto n = std::chrono::steady_clock::now();
    char b[64];
    std::snprintf(b, sizeof(b), " %s %.1f", name, std::chrono::duration<double, std::micro>(n - t).count());
    out += b;
    t = n;
  }
  ~host_phases() {
    if (on) std::fprintf(stderr, "[sfw] %s us\n", out.c_str());
  }
};

int flush_arena(sfw_handle h);

int stage_common(sfw_handle h, const sfw_robot_state *rs, const double *lin, int32_t nv, const double *ang,
                 int32_t nw, const sfw_goal_args *args, double vy_samp, int skip_zero, int64_t index_base,
                 bool grid = false) {
  if (!h) return SFW_ERR_INVALID_ARG;
  if (!rs || !lin || !ang || !args || nv <= 0 || nw <= 0)
    return fail(h, SFW_ERR_INVALID_ARG, "grid_stage: null pointer or non-positive sample count");
  if (!h->have_costmap) return fail(h, SFW_ERR_STATE, "grid_stage: no costmap set (sfw_set_costmap)");
  // A NaN would come back as a NaN cost (the header promises sentinels, never NaN) and, in a sample vector, break the
  // ordering the shared-prefix planner sorts by: O(nv + nw) checks
  if (!all_finite(&rs->x, 6) || !all_finite(&args->acc_x, 5) || !std::isfinite(vy_samp))
    return fail(h, SFW_ERR_INVALID_ARG, "grid_stage: non-finite robot state, goal argument or sample velocity");
  if (!all_finite(lin, static_cast<size_t>(nv)) || !all_finite(ang, static_cast<size_t>(nw)))
    return fail(h, SFW_ERR_INVALID_ARG, "grid_stage: non-finite sample velocity");
  host_phases ph("stage:");
  SFW_HIP(h, hipSetDevice(h->device));
  // a large costmap that has changed goes out first, by a copy of its own that runs while the host plans and packs
  if (h->cells_dirty && static_cast<size_t>(h->map_new.size_x) * h->map_new.size_y > kMapMergeMax) {
    const size_t n_cells = static_cast<size_t>(h->map_new.size_x) * h->map_new.size_y;
    SFW_HIP(h, h->cells.reserve(n_cells));
    SFW_HIP(h, hipMemcpyAsync(h->cells.p, h->pin_map.p, n_cells, hipMemcpyHostToDevice, h->stream));
    SFW_HIP(h, h->pin_map.mark(h->stream));
    h->cells_dirty = false;
  }
  ph.mark("check+setdev");
  h->h_lin.assign(lin, lin + nv);
  h->h_ang.assign(ang, ang + nw);
  h->st_K = h->K;
  h->st_A = h->A;
  h->st_O = h->O;
  h->st_NG = h->NG;
  h->st_n_grp_mem = h->n_grp_mem;
  h->nv = nv;
  h->nw = nw;
  h->rs = *rs;
  h->ga = *args;
  h->vy_samp = vy_samp;
  h->skip_zero = skip_zero;
  h->index_base = index_base;
  // The shared-prefix plan FIRST, on the host (a few microseconds since round 6: axis_classes): its class tables ride in the
  // stage's one arena copy and the levels can be enqueued right behind the pose rollout.  (Rounds 1-5 planned behind the
  // arena copy and the pose rollout, 25-50 us, and sent the tables in a second copy the first level then waited for.)
  int plan_err = SFW_OK;
  const int64_t chunk = plan_tables_host(h, &plan_err);
  if (plan_err) return plan_err;
  ph.mark("plan");
  {  // one arena, one copy: footprint | agents blob | linvels | angvels | relative-rest terms | pinned-rest table | class tables
    auto up16 = [](size_t b) { return (b + 15) & ~size_t(15); };
    const bool rest = !h->rest_pairs.empty();
    // a small costmap (<= 64 KB: a control cycle's local costmap of 200 x 200 cells) is the head of the arena and sent with
    // it when it has changed; a larger one has gone out by a copy of its own at the top of this stage (above)
    const size_t n_cells = static_cast<size_t>(h->map_new.size_x) * h->map_new.size_y;
    const bool merge = n_cells <= kMapMergeMax;
    const size_t cells_head = merge ? up16(n_cells) : 0;
    const size_t o_fp = cells_head, o_ag = o_fp + up16(sizeof(double) * (h->h_footprint.empty() ? 2 : h->h_footprint.size())),
                 o_lin = o_ag + up16(h->h_agents.size()), o_ang = o_lin + up16(sizeof(double) * nv),
                 o_rest = o_ang + up16(sizeof(double) * nw),
                 o_pin = o_rest + (rest ? up16(sizeof(double) * 2 * static_cast<size_t>(h->A)) : 0),
                 pin_doubles = 4 + static_cast<size_t>(h->A > 0 ? h->A : 0),
                 // (room for the pinned-rest table whenever the robot stands still at hand-over: whether one is needed is known
                 // once the agents' constants have been looked at, below)
                 o_cls = o_pin + ((h->pin_rest_on && rs->vx == 0.0 && rs->vy == 0.0 && h->A > 1) ? up16(sizeof(double) * pin_doubles) : 0),
                 total = o_cls + up16(sizeof(int32_t) * h->cls_ints.size());
    SFW_HIP(h, h->pin_world.reserve(total));
    // (a `world` that moves, or whose head changes size, has lost its copy of the cells)
    bool send_cells = h->cells_dirty || (merge && (!h->cells_in_world || h->world_cells_bytes != cells_head || total > h->world.cap));
    SFW_HIP(h, h->world.reserve(total));
    char *pb = h->pin_world.p;
    if (!merge) send_cells = false;  // (sent above)
    if (merge && send_cells) std::memcpy(pb, h->pin_map.p, n_cells);
    h->cells_dirty = false;
    h->cells_in_world = merge;
    h->world_cells_bytes = cells_head;
    h->size_x = h->map_new.size_x;
    h->size_y = h->map_new.size_y;
    h->origin_x = h->map_new.origin_x;
    h->origin_y = h->map_new.origin_y;
    h->resolution = h->map_new.resolution;
    if (!h->h_footprint.empty()) std::memcpy(pb + o_fp, h->h_footprint.data(), sizeof(double) * h->h_footprint.size());
    if (!h->h_agents.empty()) std::memcpy(pb + o_ag, h->h_agents.data(), h->h_agents.size());
    std::memcpy(pb + o_lin, lin, sizeof(double) * nv);
    std::memcpy(pb + o_ang, ang, sizeof(double) * nw);
    if (!h->cls_ints.empty()) std::memcpy(pb + o_cls, h->cls_ints.data(), sizeof(int32_t) * h->cls_ints.size());
    h->st_rest_pairs.clear();
    if (rest) {
      const double *pos = reinterpret_cast<const double *>(h->h_agents.data());
      const double *vel = reinterpret_cast<const double *>(h->h_agents.data() + h->ao_vel);
      h->st_rest_pairs = h->rest_pairs;
      h->st_rest_pv.assign(pos, pos + 2 * h->A);
      h->st_rest_pv.insert(h->st_rest_pv.end(), vel, vel + 2 * h->A);
      rest_forces(h->params, h->rest_pairs, pos, vel, h->A, reinterpret_cast<double *>(pb + o_rest));
    }
    bool pinned = false;
    if (o_cls > o_pin)
      pinned = pinned_rest_table(h->params, *rs, reinterpret_cast<const double *>(h->h_agents.data()),
                                 reinterpret_cast<const sfw_agent_const *>(h->h_agents.data() + h->ao_cst), h->A,
                                 reinterpret_cast<double *>(pb + o_pin));
    ph.mark("pack");
    {
      const size_t from = send_cells ? 0 : cells_head;
      // A control cycle's grid whose costmap has not changed: no copy here.  If the launch turns out to be the one-launch kernel
      // its blocks read the (~1 KB) arena from this pinned memory themselves; else the launch enqueues the copy (flush_arena).
      h->arena_pending = false;
      // ... and a GPU-filling single-chunk grid: the pose rollout this stage enqueues below (plan_tables_device) fetches the
      // arena — with a small costmap that has changed, if any — itself, its thousands of threads 16 bytes each
      const int64_t T_ = static_cast<int64_t>(nv) * nw;
      const bool small = T_ <= 1024 && from == cells_head && total - from <= (size_t(16) << 10);
      const bool early = T_ > 2048 && chunk >= T_ && total - from <= (size_t(1) << 20);
      if (grid && h->arena_direct_on && (small || early) && (total - from) % 16 == 0 && from % 16 == 0) {
        h->arena_pending = true;
        h->arena_from = from;
        h->arena_bytes = total - from;
      } else {
        SFW_HIP(h, hipMemcpyAsync(h->world.p + from, pb + from, total - from, hipMemcpyHostToDevice, h->stream));
      }
    }
    h->d_cells = merge ? reinterpret_cast<const uint8_t *>(h->world.p) : h->cells.p;
    ph.mark("h2d");
    if (!h->arena_pending) SFW_HIP(h, h->pin_world.mark(h->stream));
    ph.mark("evrec");
    const char *db = h->world.p;
    h->d_cls_tab = h->cls_ints.empty() ? nullptr : reinterpret_cast<const int32_t *>(db + o_cls);
    h->d_pin_rest = pinned ? reinterpret_cast<const double *>(db + o_pin) : nullptr;
    h->st_pin_doubles = pinned ? pin_doubles : 0;
    if (pinned) {
      const double *pos = reinterpret_cast<const double *>(h->h_agents.data());
      const sfw_agent_const *cst = reinterpret_cast<const sfw_agent_const *>(h->h_agents.data() + h->ao_cst);
      h->st_pin_pos.assign(pos, pos + 2 * h->A);
      h->st_pin_cst.assign(cst, cst + h->A);
    }
    h->d_footprint = reinterpret_cast<const double *>(db + o_fp);
    h->d_agent_pos = reinterpret_cast<const double *>(db + o_ag);
    h->d_agent_vel = reinterpret_cast<const double *>(db + o_ag + h->ao_vel);
    h->d_agent_c = reinterpret_cast<const sfw_agent_const *>(db + o_ag + h->ao_cst);
    h->d_obstacles = reinterpret_cast<const double *>(db + o_ag + h->ao_obs);
    h->d_agent_grp = reinterpret_cast<const int32_t *>(db + o_ag + h->ao_grp);
    h->d_grp_off = reinterpret_cast<const int32_t *>(db + o_ag + h->ao_off);
    h->d_grp_mem = reinterpret_cast<const int32_t *>(db + o_ag + h->ao_mem);
    h->d_linvels = reinterpret_cast<const double *>(db + o_lin);
    h->d_angvels = reinterpret_cast<const double *>(db + o_ang);
    h->d_agent_rest = rest ? reinterpret_cast<const double *>(db + o_rest) : nullptr;
  }
  if (h->pair_tab_A != h->A) {
    SFW_HIP(h, h->pair_tab.reserve(static_cast<size_t>(sfw_pair_table_entries(h->A))));
    SFW_HIP(h, sfw_launch_pair_table(h->pair_tab.p, h->A, h->stream));
    h->pair_tab_A = h->A;
  }
  const int64_t T = static_cast<int64_t>(nv) * nw;
  SFW_HIP(h, h->status.reserve(T));
  SFW_HIP(h, h->coll_step.reserve(T));
  SFW_HIP(h, h->base_cost.reserve(T));
  SFW_HIP(h, h->costs.reserve(T + (sizeof(sfw_sel) + sizeof(double) - 1) / sizeof(double)));
  h->d_sel = reinterpret_cast<sfw_sel *>(h->costs.p + T);
  SFW_HIP(h, h->partials.reserve(sfw_argmin_partials(T)));
  ph.mark("reserve");
  if (int e = plan_tables_device(h, chunk, grid)) return e;
  if (T > 1024)
    if (int e = flush_arena(h)) return e;  // (no pose rollout was enqueued after all: the copy it would have stood in for)
  ph.mark("tables+K1a");
  h->staged = true;
  h->launched = false;
  return SFW_OK;
}

// The arena a stage left in pinned memory only (arena_pending) goes to the device by a copy after all: the launch is not the
// one-launch kernel, or something is about to patch the device copy.
int flush_arena(sfw_handle h) {
  if (!h->arena_pending) return SFW_OK;
  h->arena_pending = false;
  SFW_HIP(h, hipMemcpyAsync(h->world.p + h->arena_from, h->pin_world.p + h->arena_from, h->arena_bytes, hipMemcpyHostToDevice, h->stream));
  SFW_HIP(h, h->pin_world.mark(h->stream));
  return SFW_OK;
}

int launch_common(sfw_handle h) {
  if (!h) return SFW_ERR_INVALID_ARG;
  if (!h->staged) return fail(h, SFW_ERR_STATE, "grid_launch before grid_stage");
  SFW_HIP(h, hipSetDevice(h->device));
  const int64_t T = static_cast<int64_t>(h->nv) * h->nw;
  if (h->plan_epoch != h->params_epoch)
    if (int e = flush_arena(h)) return e;  // (the re-plan below patches the device copy of the rest terms)
  // sfw_set_params since the stage: tables and shared-prefix plan are redone for the live parameters
  if (h->plan_epoch != h->params_epoch) {
    if (int e = replan_at_launch(h)) return e;  // (also drops the poses the stage started: they were rolled out under the old parameters)
    if (h->d_agent_rest && !h->st_rest_pairs.empty()) {  // the relative-rest terms depend on the sfm parameters too
      const size_t bytes = sizeof(double) * 2 * static_cast<size_t>(h->st_A);
      SFW_HIP(h, h->pin_cls.wait());
      SFW_HIP(h, h->pin_out.reserve(bytes));
      rest_forces(h->params, h->st_rest_pairs, h->st_rest_pv.data(), h->st_rest_pv.data() + 2 * h->st_A, h->st_A,
                  reinterpret_cast<double *>(h->pin_out.p));
      SFW_HIP(h, hipMemcpyAsync(const_cast<double *>(h->d_agent_rest), h->pin_out.p, bytes, hipMemcpyHostToDevice, h->stream));
      SFW_HIP(h, hipStreamSynchronize(h->stream));  // pin_out is the fetch buffer too: rare path, keep it simple
    }
    if (h->d_pin_rest && h->st_pin_doubles) {  // ... and so does the pinned-rest table
      const size_t bytes = sizeof(double) * h->st_pin_doubles;
      SFW_HIP(h, h->pin_cls.wait());
      SFW_HIP(h, h->pin_out.reserve(bytes));
      pinned_rest_table(h->params, h->rs, h->st_pin_pos.data(), h->st_pin_cst.data(), h->st_A, reinterpret_cast<double *>(h->pin_out.p));
      SFW_HIP(h, hipMemcpyAsync(const_cast<double *>(h->d_pin_rest), h->pin_out.p, bytes, hipMemcpyHostToDevice, h->stream));
      SFW_HIP(h, hipStreamSynchronize(h->stream));
    }
  }
  const int S = num_steps_of(h->params);
  int64_t chunk = h->table_chunk;  // (plan_tables_device: what the tables were sized for, under the live step count)
  if (chunk > kRowLimit) chunk = kRowLimit;
  if (chunk > T) chunk = T;
  if (chunk < 1) return fail(h, SFW_ERR_STATE, "robot-step table too small");
  if (int e = check_lds(h, chunk)) return e;
  // the shared-prefix plan was laid out for the staged step count and for chunks of whole rows
  const bool prefix = !h->prefix_steps.empty() && h->prefix_S == S && h->prefix_chunk <= chunk;
  if (prefix) chunk = h->prefix_chunk;
  const bool timing = h->timing;
  // the stage enqueued K1a of this (single-chunk) grid already — unless sfw_set_timing(1) arrived since: the events of a
  // timed launch bracket its own kernels, so the pose rollout runs again behind a freshly recorded ev[0]
  const bool poses_done = h->early_poses && chunk >= T && (!h->timing || h->early_poses_timed);
  h->early_poses = false;                                 // ... once: a second launch of the same stage rolls out again
  if (timing) {
    SFW_HIP(h, h->clock.reserve(4));
    // (cleared by the stage already when it started the pose rollout: in front of it, where the GPU waits for the host's
    // planning anyway, instead of between the rollout and the first K2 dispatch)
    // ... or will be by the pose rollout this launch enqueues; only the small-grid kernels (a block per sample, no kernel in
    // front of their K2 part) need the memset
    sfw_launch probe0;
    fill_launch(h, probe0, 0, std::min<int64_t>(T, chunk), chunk);
    if (!(poses_done && h->clock_cleared) && sfw_rollout_is_fused(probe0))
      SFW_HIP(h, hipMemsetAsync(h->clock.p, 0, 4 * sizeof(unsigned long long), h->stream));
    if (!poses_done) SFW_HIP(h, hipEventRecord(h->ev[0], h->stream));
  }
  h->clock_cleared = false;
  const bool single = chunk >= T;
  h->n_chunks = static_cast<int>((T + chunk - 1) / chunk);
  if (!single && timing) {
    while (h->chunk_ev.size() < static_cast<size_t>(3 * h->n_chunks)) {
      hipEvent_t e = nullptr;
      SFW_HIP(h, hipEventCreate(&e));
      h->chunk_ev.push_back(e);
    }
  }
  // Trajectory points on request (sfw_set_points_capture), when the whole grid goes through the fused K1
  h->captured = false;
  char *cap_pts = nullptr, *cap_n = nullptr, *cap_coll = nullptr;
  if (h->capture_points && single) {
    sfw_launch probe;
    fill_launch(h, probe, 0, T, chunk);
    if (sfw_rollout_is_fused(probe)) {
      const size_t pts_bytes = sizeof(double) * 3 * static_cast<size_t>(S) * static_cast<size_t>(T);
      // (in pinned HOST memory: the kernels write points, counts and contact steps over PCIe as they go — 43 KB for 45 samples
      // of 40 steps — and the dump is a memcpy; until round 5 a device buffer and a D2H copy per dump, 12 us)
      if (pts_bytes + 8 * static_cast<size_t>(T) > h->pin_cap.cap) SFW_HIP(h, hipStreamSynchronize(h->stream));
      SFW_HIP(h, h->pin_cap.reserve(pts_bytes + 8 * static_cast<size_t>(T)));
      cap_pts = h->pin_cap.p;
      cap_n = cap_pts + pts_bytes;
      cap_coll = cap_n + 4 * static_cast<size_t>(T);
      h->captured = true;
      h->cap_S = S;
    }
  }
  // the cost vector and the selection record reach the host through the selection kernels themselves (pinned mirror): the
  // fetch then only waits for the stream.  Grids whose vector is larger than SFW_MIRROR_MAX_MB (default 64) keep the copy.
  h->mirrored = false;
  h->fetched = false;
  double *costs_host = nullptr;
  sfw_sel *sel_host = nullptr;
  if (h->mirror_max_bytes > 0 && sizeof(double) * static_cast<size_t>(T) <= h->mirror_max_bytes) {
    const size_t cost_bytes = sizeof(double) * static_cast<size_t>(T);
    // (growing it frees the old area: no launch still in the stream may be writing there)
    if (cost_bytes + sizeof(sfw_sel) > h->pin_mirror.cap) SFW_HIP(h, hipStreamSynchronize(h->stream));
    SFW_HIP(h, h->pin_mirror.reserve(cost_bytes + sizeof(sfw_sel)));
    costs_host = reinterpret_cast<double *>(h->pin_mirror.p);
    sel_host = reinterpret_cast<sfw_sel *>(h->pin_mirror.p + cost_bytes);
    h->mirrored = true;
  }
  if (single && !prefix && !poses_done) {
    // A control cycle's grid: K1 + K2 + K3 in ONE launch (sfw_cycle_kernel) when the launch qualifies
    sfw_launch L;
    fill_launch(h, L, 0, T, chunk);
    if (h->captured) {
      L.points = reinterpret_cast<double *>(cap_pts);
      L.n_points = reinterpret_cast<int32_t *>(cap_n);
      L.coll_step = reinterpret_cast<int32_t *>(cap_coll);
      L.force_alive = 1;  // (as below)
    }
    L.clock_probe = timing ? h->clock.p : nullptr;
    L.costs_host = costs_host;
    L.sel_host = sel_host;
    if (sfw_cycle_applies(L)) {
      if (timing) SFW_HIP(h, hipEventRecord(h->ev[1], h->stream));  // (no K1 of its own: K1 time 0, the launch is "K2")
      if (h->arena_pending) {  // the kernel's blocks fetch the arena from pinned memory themselves: no copy at all this cycle
        L.arena_host = h->pin_world.p + h->arena_from;
        L.arena_dev = h->world.p + h->arena_from;
        L.arena_bytes = static_cast<uint32_t>(h->arena_bytes);
      }
      SFW_HIP(h, h->params.precision == SFW_PRECISION_F64_STRICT ? sfw_launch_cycle_strict(L, h->stream) : sfw_launch_cycle(L, h->stream));
      if (h->arena_pending) {
        h->arena_pending = false;
        SFW_HIP(h, h->pin_world.mark(h->stream));
      }
      if (timing) {
        SFW_HIP(h, hipEventRecord(h->ev[2], h->stream));
        SFW_HIP(h, hipEventRecord(h->ev[3], h->stream));
      }
      h->launched = true;
      h->launched_timed = timing;
      h->launched_cycle = true;
      return SFW_OK;
    }
  }
  h->launched_cycle = false;
  if (int e = flush_arena(h)) return e;
  int c = 0;
  for (int64_t b = 0; b < T; b += chunk, ++c) {
    const int64_t n = (T - b < chunk) ? (T - b) : chunk;
    sfw_launch L;
    fill_launch(h, L, b, n, chunk);
    if (h->captured) {
      L.points = reinterpret_cast<double *>(cap_pts);
      L.n_points = reinterpret_cast<int32_t *>(cap_n);
      L.coll_step = reinterpret_cast<int32_t *>(cap_coll);
      // a sample the costmap rejects at pose a may touch a pedestrian at an earlier step b < a, where the reference's
      // Trajectory ends (ref :613-627): integrated all the same, its cost stays -1 (finish_wave)
      L.force_alive = 1;
    }
    L.clock_probe = timing ? h->clock.p : nullptr;  // every K2 dispatch writes it; the last one (the launch over the samples) stays
    if (!single && timing) SFW_HIP(h, hipEventRecord(h->chunk_ev[3 * c], h->stream));
    if (prefix) {
      // K1a -> { K2 prefix phase on the main stream  ||  K1b + K1c on the side stream } -> K2 suffix phase.
      // The prefix phase needs the robot-step table only and under-fills the GPU (one item per class).
      // The levels are the critical path and are enqueued FIRST; the footprint checks only have to be done when the suffix
      // launch starts, so their stream gets its work afterwards (enqueued in front of the levels — round 4 — the four
      // side-stream calls delayed the first level by the host time they take: ~15 us of cfg2's 630 us step).
      if (!poses_done) SFW_HIP(h, sfw_launch_rollout_poses(L, h->stream));
      SFW_HIP(h, hipEventRecord(h->ev_poses, h->stream));
      if (timing) SFW_HIP(h, hipEventRecord(single ? h->ev[1] : h->chunk_ev[3 * c + 1], h->stream));
      const sfw_planner_s::chunk_plan &cp = h->prefix_chunks[static_cast<size_t>(c)];
      const int32_t *tab = h->d_cls_tab;
      const size_t n_lv = h->prefix_steps.size();
      for (size_t l = 0; l < n_lv; ++l) {  // the tree of shared steps, coarsest classes first
        const sfw_planner_s::level_tables &t = cp.lv[l];
        L.phase = SFW_PHASE_PREFIX;
        L.step_begin = l ? h->prefix_steps[l - 1] : 0;
        L.step_end = h->prefix_steps[l];
        L.n_cls = t.n_row * t.n_col;
        L.n_col_cls = t.n_col;
        L.row_rep = tab + t.o_row_rep;
        L.col_rep = tab + t.o_col_rep;
        L.resume = l > 0;
        if (l > 0) {
          L.n_col_src = cp.lv[l - 1].n_col;
          L.row_src = tab + t.o_row_src;
          L.col_src = tab + t.o_col_src;
          L.in_state = h->cls_state[(l - 1) & 1].p;
          L.in_dead = h->cls_dead[(l - 1) & 1].p;
        }
        L.out_state = h->cls_state[l & 1].p;
        L.out_dead = h->cls_dead[l & 1].p;
        SFW_HIP(h, launch_social_of(L, h->stream));
      }
      {  // K1b + K1c beside the levels (they wait for the pose rollout, not for the levels)
        sfw_launch Lc = L;
        Lc.phase = SFW_PHASE_WHOLE;
        SFW_HIP(h, hipStreamWaitEvent(h->side, h->ev_poses, 0));
        SFW_HIP(h, sfw_launch_rollout_costmap(Lc, h->side));
        SFW_HIP(h, hipEventRecord(h->ev_side, h->side));
      }
      SFW_HIP(h, hipStreamWaitEvent(h->stream, h->ev_side, 0));
      L.phase = SFW_PHASE_SUFFIX;
      L.step_begin = h->prefix_steps.back();
      L.step_end = S;
      L.resume = 1;
      L.n_col_src = cp.lv[n_lv - 1].n_col;
      L.row_src = tab + cp.o_row_cls;
      L.col_src = tab + h->prefix_o_col_cls;
      L.in_state = h->cls_state[(n_lv - 1) & 1].p;
      L.in_dead = h->cls_dead[(n_lv - 1) & 1].p;
      SFW_HIP(h, launch_social_of(L, h->stream, &h->split));  // (`side` is idle: the stream has just waited for it)
    } else {
      if (!poses_done) SFW_HIP(h, sfw_launch_rollout_poses(L, h->stream));
      SFW_HIP(h, sfw_launch_rollout_costmap(L, h->stream));
      if (timing) SFW_HIP(h, hipEventRecord(single ? h->ev[1] : h->chunk_ev[3 * c + 1], h->stream));
      SFW_HIP(h, launch_social_of(L, h->stream, &h->split));
    }
    if (!single && timing) SFW_HIP(h, hipEventRecord(h->chunk_ev[3 * c + 2], h->stream));
  }
  if (timing) SFW_HIP(h, hipEventRecord(h->ev[2], h->stream));
  SFW_HIP(h, sfw_launch_argmin(h->costs.p, h->d_linvels, h->d_angvels, h->nw, T, h->index_base,
                               h->partials.p, h->d_sel, h->stream, costs_host, sel_host));
  if (timing) SFW_HIP(h, hipEventRecord(h->ev[3], h->stream));
  h->launched = true;
  h->launched_timed = timing;
  return SFW_OK;
}

void sel_to_best(sfw_handle h, const sfw_sel &s, sfw_best *best, sfw_best_key *key) {
  const bool found = std::isfinite(s.cost);
  if (best) {
    best->n_valid = s.n_valid;
    if (found) {
      const int64_t local = -s.neg_index - h->index_base;
      best->index = local;
      best->cost = s.cost;
      best->vx = h->h_lin[static_cast<size_t>(local / h->nw)];
      best->vy = 0.0;
      best->vtheta = h->h_ang[static_cast<size_t>(local % h->nw)];
    } else {  // ref :456-468: stop the robot
      best->index = -1;
      best->cost = -1.0;
      best->vx = best->vy = best->vtheta = 0.0;
    }
  }
  if (key) {
    key->cost = found ? s.cost : std::numeric_limits<double>::infinity();
    key->neg_linvel = found ? s.neg_linvel : std::numeric_limits<double>::infinity();
    key->abs_angvel = found ? s.abs_angvel : std::numeric_limits<double>::infinity();
    key->neg_index = found ? static_cast<double>(s.neg_index) : std::numeric_limits<double>::infinity();
  }
}

}  // namespace

extern "C" {

void sfw_params_default(sfw_params *p) {
  if (!p) return;
  std::memset(p, 0, sizeof(*p));
  // ControllerParams defaults, reference sfw_planner.hpp:56-66
  p->max_vel_x = 0.7;
  p->sim_time = 1.0;
  p->sim_granularity = 0.025;
  p->robot_radius = 0.35f;
  p->social_weight = 1.2;
  p->costmap_weight = 2.0;
  p->angle_weight = 0.7;
  p->distance_weight = 1.0;
  p->vel_weight = 1.0;
  p->robot_goal_radius = 0.20;  // reference src/sfw_planner.cpp:608
  // lightsfm sfm::Parameters defaults
  p->sfm_force_factor_desired = 2.0;
  p->sfm_force_factor_obstacle = 10.0;
  p->sfm_force_sigma_obstacle = 0.2;
  p->sfm_force_factor_social = 2.1;
  p->sfm_lambda = 2.0;
  p->sfm_gamma = 0.35;
  p->sfm_n = 2.0;
  p->sfm_n_prime = 3.0;
  p->sfm_relaxation_time = 0.5;
  p->sfm_force_factor_group_gaze = 3.0;
  p->sfm_force_factor_group_coherence = 2.0;
  p->sfm_force_factor_group_repulsion = 1.0;
  p->precision = SFW_PRECISION_F64;
}

int sfw_abi_version(void) { return SFW_ABI_VERSION; }

int sfw_create(const sfw_params *params, int device, sfw_handle *out) {
  if (!out) return SFW_ERR_INVALID_ARG;
  *out = nullptr;
  if (int e = check_params(nullptr, params)) return e;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return SFW_ERR_NO_DEVICE;
  if (device < 0 || device >= count) return SFW_ERR_INVALID_ARG;
#ifdef SFW_ABLATION_BUILD
  {  // results wrong by construction (sfw_device.h): only a timing script that says so gets a handle
    const char *ok = std::getenv("SFW_ALLOW_ABLATION");
    if (!ok || ok[0] != '1') {
      std::fprintf(stderr, "[sfw] this libsfw_hip.so is an ABLATION build (results wrong by construction); set SFW_ALLOW_ABLATION=1 to time it\n");
      return SFW_ERR_UNSUPPORTED;
    }
  }
#endif
  sfw_handle h = new (std::nothrow) sfw_planner_s();
  if (!h) return SFW_ERR_HIP;
  h->params = *params;
  h->device = device;
  if (const char *b = std::getenv("SFW_PREFIX")) {  // "0" off, "10" one split, "3,7,12" several levels
    for (const char *q = b; *q;) {
      char *end = nullptr;
      const long v = std::strtol(q, &end, 10);
      if (end == q) break;
      if (v >= 0) h->prefix_env.push_back(static_cast<int>(v));
      q = (*end == ',') ? end + 1 : end;
    }
    std::sort(h->prefix_env.begin(), h->prefix_env.end());
    h->prefix_env.erase(std::unique(h->prefix_env.begin(), h->prefix_env.end()), h->prefix_env.end());
  }
  if (const char *b = std::getenv("SFW_FORCE_FLAT")) h->k2_form = std::atoi(b) == 1 ? SFW_K2_FLAT : std::atoi(b) == 0 ? SFW_K2_REGISTER : SFW_K2_AUTO;
  if (const char *b = std::getenv("SFW_PIN_REST")) h->pin_rest_on = std::atoi(b) != 0;
  if (const char *b = std::getenv("SFW_SPIN_US")) h->spin_us = std::max(0L, std::atol(b));
  if (const char *b = std::getenv("SFW_ARENA_DIRECT")) h->arena_direct_on = std::atoi(b) != 0;
  if (const char *b = std::getenv("SFW_MIRROR_MAX_MB")) h->mirror_max_bytes = static_cast<size_t>(std::max(0L, std::atol(b))) << 20;
  if (const char *b = std::getenv("SFW_OBS_TASKS")) h->obs_tasks_force = std::atoi(b) != 0 ? 1 : 0;
  if (const char *b = std::getenv("SFW_TABLE_BUDGET_MB")) {
    long mb = std::atol(b);
    if (mb > 0) h->table_budget_bytes = static_cast<size_t>(mb) << 20;
  }
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) {
    // the launch heuristics (shared-prefix cost model, organisation thresholds, XCD-contiguous block order) scale with the
    // device: a CPX partition of an MI355X shows 32 compute units on one XCD where the whole chip shows 256 on eight
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) h->n_cu = prop.multiProcessorCount;
    h->n_cu = env_device_cus(h->n_cu);
    // XCDs: what the device says (hipDeviceAttributeNumberOfXccs; 8 on a whole MI355X, 1 on a CPX partition) unless the CU count
    // is pretended (SFW_DEVICE_CUS) or the attribute is not answered — then CUs / 32, right for gfx950's 32-CU XCDs only
    int xccs = 0;
    if (!std::getenv("SFW_DEVICE_CUS") && hipDeviceGetAttribute(&xccs, hipDeviceAttributeNumberOfXccs, device) == hipSuccess && xccs >= 1)
      h->n_xcd = xccs;
    else
      h->n_xcd = std::max(1, h->n_cu / SFW_CUS_PER_XCD);
    (void)hipGetLastError();  // (an unanswered attribute is not an error of this call)
    if (const char *b = std::getenv("SFW_DEVICE_XCDS")) {
      const long v = std::atol(b);
      if (v >= 1 && v <= 64) h->n_xcd = static_cast<int>(v);
    }
  }
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_poses, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_side, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->split.fork, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->split.join, hipEventDisableTiming);
  if (e == hipSuccess) h->split.side = h->side;
  for (int i = 0; i < 5 && e == hipSuccess; ++i) e = hipEventCreate(&h->ev[i]);
  if (e == hipSuccess) e = h->cycle_counter.reserve(1);
  if (e == hipSuccess) e = hipMemsetAsync(h->cycle_counter.p, 0, sizeof(unsigned), h->stream);
  if (e != hipSuccess) {
    sfw_destroy(h);
    return SFW_ERR_HIP;
  }
  *out = h;
  return SFW_OK;
}

int sfw_destroy(sfw_handle h) {
  if (!h) return SFW_OK;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->side) (void)hipStreamSynchronize(h->side);
  h->cells.release();
  h->world.release();
  h->pair_tab.release();
  h->status.release();
  h->coll_step.release();
  h->base_cost.release();
  h->costs.release();
  h->ptab.release();
  h->cs_tab.release();
  h->fcode.release();
  h->partials.release();
  h->clock.release();
  h->cycle_counter.release();
  h->pin_cap.release();
  h->points.release();
  h->n_points.release();
  h->one_out.release();
  h->pts_status.release();
  h->pts_coll.release();
  h->pts_base.release();
  h->pts_costs.release();
  h->pts_ptab.release();
  h->pts_cs.release();
  h->pts_fcode.release();
  h->pin_map.release();
  h->pin_world.release();
  h->pin_out.release();
  h->pin_mirror.release();
  h->pin_one.release();
  h->pin_cls.release();
  h->d_cls.release();
  for (auto &b : h->cls_dead) b.release();
  for (auto &b : h->cls_state) b.release();
  for (auto &e : h->ev)
    if (e) (void)hipEventDestroy(e);
  for (auto &e : h->chunk_ev)
    if (e) (void)hipEventDestroy(e);
  if (h->ev_poses) (void)hipEventDestroy(h->ev_poses);
  if (h->ev_side) (void)hipEventDestroy(h->ev_side);
  if (h->split.fork) (void)hipEventDestroy(h->split.fork);
  if (h->split.join) (void)hipEventDestroy(h->split.join);
  if (h->side) (void)hipStreamDestroy(h->side);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return SFW_OK;
}

int sfw_set_params(sfw_handle h, const sfw_params *params) {
  if (!h) return SFW_ERR_INVALID_ARG;
  if (int e = check_params(h, params)) return e;
  if (std::memcmp(&h->params, params, sizeof(sfw_params)) != 0) ++h->params_epoch;
  h->params = *params;
  return SFW_OK;
}

const char *sfw_last_error(sfw_handle h) { return h ? h->err.c_str() : "null handle"; }

int sfw_set_costmap(sfw_handle h, const uint8_t *cells, uint32_t size_x, uint32_t size_y, double origin_x,
                    double origin_y, double resolution) {
  if (!h) return SFW_ERR_INVALID_ARG;
  if (!cells || size_x == 0 || size_y == 0 || !(resolution > 0))
    return fail(h, SFW_ERR_INVALID_ARG, "set_costmap: null cells, zero size or non-positive resolution");
  SFW_HIP(h, hipSetDevice(h->device));
  const size_t n = static_cast<size_t>(size_x) * size_y;
  // The snapshot is taken here (the caller's buffer may change on return) and reaches the device with the NEXT stage — like
  // footprint and agents — as part of that stage's one arena copy when it is small (a control cycle's local costmap: one
  // copy per cycle instead of two with the host's call-to-call latency between them), by a copy of its own otherwise.
  // (an unchanged snapshot — a controller hands its local costmap over every cycle, the map changes at the costmap's own
  // rate — is recognised here, ~1 us for 40 KB, and not sent again)
  if (h->have_costmap && h->pin_map.p && n <= h->pin_map.cap && h->map_new.size_x == size_x && h->map_new.size_y == size_y &&
      h->map_new.origin_x == origin_x && h->map_new.origin_y == origin_y && h->map_new.resolution == resolution &&
      std::memcmp(h->pin_map.p, cells, n) == 0)
    return SFW_OK;
  SFW_HIP(h, h->pin_map.reserve(n));  // (waits for a copy out of it that is still pending)
  std::memcpy(h->pin_map.p, cells, n);
  h->map_new = {size_x, size_y, origin_x, origin_y, resolution};
  h->cells_dirty = true;
  h->have_costmap = true;
  return SFW_OK;
}

int sfw_set_footprint(sfw_handle h, const double *xy, int32_t K) {
  if (!h) return SFW_ERR_INVALID_ARG;
  if (K < 0 || (K > 0 && !xy)) return fail(h, SFW_ERR_INVALID_ARG, "set_footprint: bad arguments");
  h->h_footprint.assign(xy, xy + 2 * static_cast<size_t>(K));  // uploaded with the next stage
  h->K = K;
  return SFW_OK;
}

int sfw_set_agents(sfw_handle h, const sfw_agent *agents, int32_t A, const double *obstacles_xy, int32_t O) {
  if (!h) return SFW_ERR_INVALID_ARG;
  if (A < 0 || O < 0 || (A > 0 && !agents) || (O > 0 && !obstacles_xy))
    return fail(h, SFW_ERR_INVALID_ARG, "set_agents: bad arguments");
  for (int i = 0; i < A; ++i) {
    const sfw_agent &a = agents[i];
    if (!all_finite(&a.x, 4) || !std::isfinite(a.desired_velocity) || !std::isfinite(a.radius) ||
        (a.has_goal && !(all_finite(&a.goal_x, 2) && std::isfinite(a.goal_radius))))
      return fail(h, SFW_ERR_INVALID_ARG, "set_agents: non-finite field in agent " + std::to_string(i));
    // desired_velocity <= 0 is accepted, as the reference accepts people_velocity_ = 0 (sensor_interface.cpp:503): the speed
    // clamp of lightsfm's updatePosition pins such a person where it stands (include/sfw_hip.h, sfw_agent)
  }
  if (O > 0 && !all_finite(obstacles_xy, 2 * static_cast<size_t>(O)))
    return fail(h, SFW_ERR_INVALID_ARG, "set_agents: non-finite laser point");
  // The flat K2 addresses an agent's LDS words through 16-bit byte offsets (pair table) and the whole set has to fit one
  // wave's LDS allocation long before that: refused here, not as a failed launch later
  if (A > 8190) return fail(h, SFW_ERR_UNSUPPORTED, "set_agents: more than 8190 agents");
  const size_t An = static_cast<size_t>(A > 0 ? A : 1), On = static_cast<size_t>(O > 0 ? O : 1);
  // groups: dense index in order of first appearance, CSR member lists in agent order
  std::vector<int32_t> grp(An, -1), ids, off(1, 0), mem;
  for (int i = 0; i < A; ++i) {
    if (agents[i].group_id < 0) continue;
    size_t q = 0;
    while (q < ids.size() && ids[q] != agents[i].group_id) ++q;
    if (q == ids.size()) ids.push_back(agents[i].group_id);
    grp[i] = static_cast<int32_t>(q);
  }
  for (size_t q = 0; q < ids.size(); ++q) {
    for (int i = 0; i < A; ++i)
      if (grp[i] == static_cast<int32_t>(q)) mem.push_back(i);
    off.push_back(static_cast<int32_t>(mem.size()));
  }
  const int n_mem = off.back();
  // the set has to fit one wave's LDS allocation whatever the grid (the flat form of one sample is the smallest): refused
  // here, before the all-pairs scan below and long before a launch
  if (A > 0 && sfw_social_lds_bytes(A, O, static_cast<int>(ids.size()), n_mem, 1, SFW_K2_FLAT, h->n_cu) > 160 * 1024)
    return fail(h, SFW_ERR_UNSUPPORTED, "set_agents: agent/obstacle set does not fit the 160 KiB LDS of one CU");
  if (mem.empty()) mem.push_back(0);
  // host blob (uploaded with the next stage): pos | vel | const | obstacles | grp | off | mem
  auto up16 = [](size_t b) { return (b + 15) & ~size_t(15); };
  const size_t o_pos = 0, o_vel = o_pos + up16(16 * An), o_cst = o_vel + up16(16 * An),
               o_obs = o_cst + up16(sizeof(sfw_agent_const) * An),
               // + 64 behind the points: the kernels read them in groups of four (s_load_dwordx16) and, in the flat form's task
               // loop, up to two points ahead of the one being evaluated, without clamping the index (loaded, never used)
               o_grp = o_obs + up16(16 * On) + 64,
               o_off = o_grp + up16(4 * An), o_mem = o_off + up16(4 * off.size()),
               total = o_mem + up16(4 * mem.size());
  h->h_agents.assign(total, 0);
  char *base = h->h_agents.data();
  double *pos = reinterpret_cast<double *>(base + o_pos), *vel = reinterpret_cast<double *>(base + o_vel);
  sfw_agent_const *cst = reinterpret_cast<sfw_agent_const *>(base + o_cst);
  for (int i = 0; i < A; ++i) {
    pos[2 * i] = agents[i].x;
    pos[2 * i + 1] = agents[i].y;
    vel[2 * i] = agents[i].vx;
    vel[2 * i + 1] = agents[i].vy;
    sfw_agent_const &c = cst[i];
    c.goal_x = agents[i].goal_x;
    c.goal_y = agents[i].goal_y;
    c.goal_radius = agents[i].goal_radius;
    c.desired_velocity = agents[i].desired_velocity;
    c.radius = agents[i].radius;
    c.id = agents[i].id;
    c.has_goal = agents[i].has_goal ? 1 : 0;
  }
  if (O > 0) std::memcpy(base + o_obs, obstacles_xy, sizeof(double) * 2 * O);
  std::memcpy(base + o_grp, grp.data(), 4 * grp.size());
  std::memcpy(base + o_off, off.data(), 4 * off.size());
  std::memcpy(base + o_mem, mem.data(), 4 * mem.size());
  h->ao_vel = o_vel;
  h->ao_cst = o_cst;
  h->ao_obs = o_obs;
  h->ao_grp = o_grp;
  h->ao_off = o_off;
  h->ao_mem = o_mem;
  h->NG = static_cast<int>(ids.size());
  h->n_grp_mem = n_mem;
  h->A = A;
  h->O = O;
  // Pairs whose w x diff is exactly 0 in the handed-over state — relative rest (w = 0) and motion exactly along the
  // connecting line — are the pairs for which the kernels' sign(theta) = sign(w x diff) is 0 while lightsfm's theta is
  // rounding noise around 0 or exactly +-pi: their angular terms are evaluated on the host (rest_forces).  The test is
  // the kernels' own expression (pair_force_state), IEEE fma on both sides.  All pairs: ~1 ns each, 1275 at 50
  // pedestrians; the scoring that follows costs A^2 per sample-step.
  h->rest_pairs.clear();
  for (int i = 0; i < A; ++i)
    for (int j = i + 1; j < A; ++j) {
      const double dx = agents[j].x - agents[i].x, dy = agents[j].y - agents[i].y;
      const double wx = agents[i].vx - agents[j].vx, wy = agents[i].vy - agents[j].vy;
      if (std::fma(wx, dy, -(wy * dx)) == 0.0 && !(dx == 0.0 && dy == 0.0)) {
        h->rest_pairs.emplace_back(i, j);
        h->rest_pairs.emplace_back(j, i);
      }
    }
  return SFW_OK;
}

int sfw_grid_stage(sfw_handle h, const sfw_robot_state *rs, const double *linvels, int32_t nv,
                   const double *angvels, int32_t nw, const sfw_goal_args *args, int64_t index_base) {
  return stage_common(h, rs, linvels, nv, angvels, nw, args, 0.0, 1, index_base, true);
}

int sfw_grid_launch(sfw_handle h) { return launch_common(h); }

int sfw_grid_sync(sfw_handle h) {
  if (!h) return SFW_ERR_INVALID_ARG;
  SFW_HIP(h, hipSetDevice(h->device));
  SFW_HIP(h, hipStreamSynchronize(h->stream));
  return SFW_OK;
}

int sfw_grid_fetch(sfw_handle h, double *costs_out, sfw_best *best_out, sfw_best_key *key_out) {
  if (!h) return SFW_ERR_INVALID_ARG;
  if (!h->launched) return fail(h, SFW_ERR_STATE, "grid_fetch before grid_launch");
  SFW_HIP(h, hipSetDevice(h->device));
  const int64_t T = static_cast<int64_t>(h->nv) * h->nw;
  // costs and the selection record are contiguous on the device: one copy fetches both
  sfw_sel s;
  if (h->mirrored) {  // the launch's selection kernels wrote both to pin_mirror: nothing to copy, only to wait for
    const size_t all_bytes = sizeof(double) * static_cast<size_t>(T);
    SFW_HIP(h, wait_stream(h));
    stream_is_idle(h);
    h->fetched = true;
    if (costs_out) std::memcpy(costs_out, h->pin_mirror.p, all_bytes);
    std::memcpy(&s, h->pin_mirror.p + all_bytes, sizeof(s));
  } else {
    const size_t cost_bytes = costs_out ? sizeof(double) * static_cast<size_t>(T) : 0;
    SFW_HIP(h, h->pin_out.reserve(cost_bytes + sizeof(sfw_sel)));
    const char *src = reinterpret_cast<const char *>(h->d_sel) - cost_bytes;
    SFW_HIP(h, hipMemcpyAsync(h->pin_out.p, src, cost_bytes + sizeof(sfw_sel), hipMemcpyDeviceToHost, h->stream));
    SFW_HIP(h, wait_stream(h));
    stream_is_idle(h);
    h->fetched = true;
    if (costs_out) std::memcpy(costs_out, h->pin_out.p, cost_bytes);
    std::memcpy(&s, h->pin_out.p + cost_bytes, sizeof(s));
  }
  sel_to_best(h, s, best_out, key_out);
  return SFW_OK;
}

const double *sfw_grid_costs_view(sfw_handle h) {
  if (!h || !h->launched || !h->mirrored || !h->fetched) return nullptr;
  return reinterpret_cast<const double *>(h->pin_mirror.p);
}

int sfw_score_grid(sfw_handle h, const sfw_robot_state *rs, const double *linvels, int32_t nv,
                   const double *angvels, int32_t nw, const sfw_goal_args *args, double *costs_out,
                   sfw_best *best_out) {
  if (int e = sfw_grid_stage(h, rs, linvels, nv, angvels, nw, args, 0)) return e;
  if (int e = sfw_grid_launch(h)) return e;
  return sfw_grid_fetch(h, costs_out, best_out, nullptr);
}

int sfw_score_one(sfw_handle h, const sfw_robot_state *rs, double vx_samp, double vy_samp, double vtheta_samp,
                  const sfw_goal_args *args, double *cost_out, double *points_xyth, int32_t points_cap,
                  int32_t *n_points) {
  if (!h) return SFW_ERR_INVALID_ARG;
  if (!cost_out) return fail(h, SFW_ERR_INVALID_ARG, "score_one: cost_out is NULL");
  if (int e = stage_common(h, rs, &vx_samp, 1, &vtheta_samp, 1, args, vy_samp, 0, 0, true)) return e;
  if (int e = check_lds(h, 1)) return e;
  const int S = num_steps_of(h->params);
  {
    // The latency path (called on every approach / rotate cycle, ref :204-206, :299-301) as ONE launch and no copy: the
    // one-launch kernel with its outputs — cost, point count, contact step, Trajectory points — in pinned host memory, the
    // stage's arena fetched by the kernel (stage_common left it pending), no selection.
    const size_t head = 16, pts_bytes = sizeof(double) * 3 * static_cast<size_t>(S);
    sfw_launch L;
    fill_launch(h, L, 0, 1, 1);
    L.sel_out = nullptr;
    L.cycle_counter = nullptr;
    const bool want_pts = (points_xyth && points_cap > 0) || n_points;
    L.force_alive = want_pts ? 1 : 0;
    if (sfw_cycle_applies(L)) {
      if (head + pts_bytes > h->pin_one.cap) SFW_HIP(h, hipStreamSynchronize(h->stream));  // (growing it frees the old area)
      SFW_HIP(h, h->pin_one.reserve(head + pts_bytes));
      L.costs = reinterpret_cast<double *>(h->pin_one.p);
      L.n_points = reinterpret_cast<int32_t *>(h->pin_one.p + 8);
      L.coll_step = reinterpret_cast<int32_t *>(h->pin_one.p + 12);
      L.points = reinterpret_cast<double *>(h->pin_one.p + head);
      if (h->arena_pending) {
        L.arena_host = h->pin_world.p + h->arena_from;
        L.arena_dev = h->world.p + h->arena_from;
        L.arena_bytes = static_cast<uint32_t>(h->arena_bytes);
      }
      SFW_HIP(h, h->params.precision == SFW_PRECISION_F64_STRICT ? sfw_launch_cycle_strict(L, h->stream) : sfw_launch_cycle(L, h->stream));
      if (h->arena_pending) {
        h->arena_pending = false;
        SFW_HIP(h, h->pin_world.mark(h->stream));
      }
      SFW_HIP(h, wait_stream(h));
      stream_is_idle(h);
      int32_t n = 0, coll = -1;
      std::memcpy(cost_out, h->pin_one.p, sizeof(double));
      std::memcpy(&n, h->pin_one.p + 8, sizeof(n));
      std::memcpy(&coll, h->pin_one.p + 12, sizeof(coll));
      if (coll >= 0 && coll + 1 < n) n = coll + 1;  // rejected by contact at step `coll`: poses 0..coll were added
      if (n_points) *n_points = n;
      if (points_xyth && points_cap > 0 && n > 0) {
        const int m = n < points_cap ? n : points_cap;
        std::memcpy(points_xyth, h->pin_one.p + head, sizeof(double) * 3 * static_cast<size_t>(m));
      }
      h->staged = false;  // score_one clobbers the staged grid
      h->launched = false;
      return SFW_OK;
    }
    if (int e = flush_arena(h)) return e;  // the three-kernel path below reads the device copy
  }
  // cost (8) | n_points (4) | coll_step (4) | points (24 S): contiguous on the device, so the latency path
  // (called on every approach / rotate cycle, ref :204-206, :299-301) pays one pinned D2H, like sfw_grid_fetch
  const size_t head = 16, pts_bytes = sizeof(double) * 3 * static_cast<size_t>(S);
  SFW_HIP(h, h->one_out.reserve(head + pts_bytes));
  sfw_launch L;
  fill_launch(h, L, 0, 1, 1);
  L.costs = reinterpret_cast<double *>(h->one_out.p);
  L.n_points = reinterpret_cast<int32_t *>(h->one_out.p + 8);
  L.coll_step = reinterpret_cast<int32_t *>(h->one_out.p + 12);
  L.points = reinterpret_cast<double *>(h->one_out.p + head);
  const bool want_pts = (points_xyth && points_cap > 0) || n_points;
  // With points wanted a costmap-rejected sample is integrated too: a pedestrian contact at an earlier step ends the
  // reference's Trajectory there (ref :613-627 returns before the later illegal pose is ever reached)
  L.force_alive = want_pts ? 1 : 0;
  SFW_HIP(h, sfw_launch_rollout(L, h->stream));
  SFW_HIP(h, launch_social_of(L, h->stream));
  const size_t fetch = head + ((points_xyth && points_cap > 0) ? pts_bytes : 0);
  SFW_HIP(h, h->pin_out.reserve(fetch));
  SFW_HIP(h, hipMemcpyAsync(h->pin_out.p, h->one_out.p, fetch, hipMemcpyDeviceToHost, h->stream));
  SFW_HIP(h, hipStreamSynchronize(h->stream));
  int32_t n = 0, coll = -1;
  std::memcpy(cost_out, h->pin_out.p, sizeof(double));
  std::memcpy(&n, h->pin_out.p + 8, sizeof(n));
  std::memcpy(&coll, h->pin_out.p + 12, sizeof(coll));
  if (coll >= 0 && coll + 1 < n) n = coll + 1;  // rejected by contact at step `coll`: poses 0..coll were added
  if (n_points) *n_points = n;
  if (points_xyth && points_cap > 0 && n > 0) {
    const int m = n < points_cap ? n : points_cap;
    std::memcpy(points_xyth, h->pin_out.p + head, sizeof(double) * 3 * static_cast<size_t>(m));
  }
  h->staged = false;  // score_one clobbers the staged grid
  h->launched = false;
  return SFW_OK;
}

int sfw_plan_shared_prefix(const double *linvels, int32_t nv, const double *angvels, int32_t nw, double vx0, double vtheta0,
                           double acc_x, double acc_theta, double sim_time, int32_t num_steps, int32_t n_agents,
                           int32_t *level_ends, int64_t *level_classes, int32_t cap, int32_t *n_levels) {
  if (!linvels || !angvels || nv <= 0 || nw <= 0 || num_steps < 1 || !n_levels || cap < 0 ||
      (cap > 0 && (!level_ends || !level_classes)))
    return SFW_ERR_INVALID_ARG;
  *n_levels = 0;
  const int64_t T = static_cast<int64_t>(nv) * nw;
  if (n_agents < 2 || num_steps < 2 || T < 4096) return SFW_OK;
  const std::vector<double> lin(linvels, linvels + nv), ang(angvels, angvels + nw);
  prefix_classes pc;
  classes_of_grid(pc, lin, ang, vx0, vtheta0, acc_x, acc_theta, sim_time / num_steps, num_steps);
  const int cus = env_device_cus(SFW_DEFAULT_CUS);  // host only: no device to ask
  const std::vector<int> steps = choose_levels(pc, T, num_steps, static_cast<double>(sfw_samples_per_wave(n_agents, T, SFW_K2_AUTO, cus)), cus);
  *n_levels = static_cast<int32_t>(steps.size());
  for (size_t l = 0; l < steps.size() && l < static_cast<size_t>(cap); ++l) {
    level_ends[l] = steps[l];
    level_classes[l] = static_cast<int64_t>(pc.n_rows_at(steps[l])) * pc.n_cols_at(steps[l]);
  }
  return SFW_OK;
}

int sfw_plan_axis_classes(const double *targets, int32_t n, double v0, double a_max, double dt, int32_t max_p, int32_t form,
                          int32_t *counts, int32_t *classes, int32_t *n_levels, int32_t *closed_form) {
  if (!targets || n <= 0 || max_p < 1 || !counts || !n_levels || (form != 0 && form != 1)) return SFW_ERR_INVALID_ARG;
  if (!all_finite(targets, static_cast<size_t>(n)) || !std::isfinite(v0) || !std::isfinite(a_max) || !std::isfinite(dt))
    return SFW_ERR_INVALID_ARG;
  const axis_classes a = classes_of_axis(std::vector<double>(targets, targets + n), v0, a_max, dt, max_p, form == 0);
  *n_levels = static_cast<int32_t>(a.n.size());
  if (closed_form) *closed_form = a.fast ? 1 : 0;
  for (size_t l = 0; l < a.n.size(); ++l) {
    counts[l] = a.n[l];
    if (classes) std::memcpy(classes + l * static_cast<size_t>(n), a.level(static_cast<int>(l) + 1).data(), sizeof(int32_t) * static_cast<size_t>(n));
  }
  return SFW_OK;
}

int sfw_plan_row_blocks(const double *linvels, int32_t nv, const double *angvels, int32_t nw, double vx0, double vtheta0,
                        double acc_x, double acc_theta, double sim_time, int32_t num_steps, int32_t n_agents, int32_t R,
                        int32_t *row0) {
  if (!linvels || !angvels || nv <= 0 || nw <= 0 || num_steps < 1 || R < 1 || !row0) return SFW_ERR_INVALID_ARG;
  if (!all_finite(linvels, static_cast<size_t>(nv)) || !all_finite(angvels, static_cast<size_t>(nw))) return SFW_ERR_INVALID_ARG;
  const std::vector<double> lin(linvels, linvels + nv), ang(angvels, angvels + nw);
  std::vector<int32_t> cuts;
  plan_row_blocks(lin, ang, vx0, vtheta0, acc_x, acc_theta, sim_time / num_steps, num_steps, n_agents, R, SFW_K2_AUTO,
                  env_device_cus(SFW_DEFAULT_CUS), cuts);
  std::memcpy(row0, cuts.data(), sizeof(int32_t) * (static_cast<size_t>(R) + 1));
  return SFW_OK;
}

int sfw_grid_plan_info(sfw_handle h, sfw_plan_info *out) {
  if (!h || !out) return SFW_ERR_INVALID_ARG;
  if (!h->staged) return fail(h, SFW_ERR_STATE, "grid_plan_info before grid_stage");
  const int64_t T = static_cast<int64_t>(h->nv) * h->nw;
  out->split_step = h->prefix_steps.empty() ? 0 : h->prefix_steps.back();
  out->levels = static_cast<int32_t>(h->prefix_steps.size());
  out->samples = T;
  out->classes = h->prefix_last_classes;
  out->class_steps = h->prefix_class_steps;
  if (!h->prefix_steps.empty()) {
    out->chunks = static_cast<int32_t>(h->prefix_chunks.size());
  } else {
    int64_t chunk = h->table_chunk;
    if (chunk > T) chunk = T;
    out->chunks = chunk > 0 ? static_cast<int32_t>((T + chunk - 1) / chunk) : 0;
  }
  out->organisation = sfw_social_organisation(h->st_A, out->chunks > 0 ? (T + out->chunks - 1) / out->chunks : T, h->st_O, h->k2_form, h->n_cu);
  out->flat_samples = sfw_social_flat_items(h->st_A, h->st_O, h->st_NG, out->chunks > 0 ? (T + out->chunks - 1) / out->chunks : T, h->k2_form, h->n_cu);
  out->rest_noise_unreproduced = rest_noise_unreproduced(h) ? 1 : 0;
  out->one_launch = 0;
  if (out->chunks == 1 && h->prefix_steps.empty()) {
    sfw_launch L;
    fill_launch(h, L, 0, T, T);
    out->one_launch = sfw_cycle_applies(L) ? 1 : 0;
  }
  return SFW_OK;
}

int sfw_set_k2_form(sfw_handle h, int32_t form) {
  if (!h) return SFW_ERR_INVALID_ARG;
  if (form != SFW_K2_AUTO && form != SFW_K2_REGISTER && form != SFW_K2_FLAT)
    return fail(h, SFW_ERR_INVALID_ARG, "set_k2_form: form must be SFW_K2_AUTO, SFW_K2_REGISTER or SFW_K2_FLAT");
  h->k2_form = form;
  return SFW_OK;
}

int sfw_set_timing(sfw_handle h, int32_t enabled) {
  if (!h) return SFW_ERR_INVALID_ARG;
  h->timing = enabled != 0;
  return SFW_OK;
}

int sfw_last_launch_ms(sfw_handle h, int32_t which, float *ms_out) {
  if (!h || !ms_out) return SFW_ERR_INVALID_ARG;
  if (!h->launched) return fail(h, SFW_ERR_STATE, "last_launch_ms before grid_launch");
  if (!h->launched_timed) return fail(h, SFW_ERR_STATE, "last_launch_ms: timing was off for the last launch (sfw_set_timing)");
  SFW_HIP(h, hipSetDevice(h->device));
  SFW_HIP(h, hipEventSynchronize(h->ev[3]));
  if (h->n_chunks > 1 && (which == 1 || which == 2)) {  // sum over the chunks of a multi-chunk launch
    float total = 0.0f;
    for (int c = 0; c < h->n_chunks; ++c) {
      float ms = 0.0f;
      const int a = 3 * c + (which == 1 ? 0 : 1);
      SFW_HIP(h, hipEventElapsedTime(&ms, h->chunk_ev[a], h->chunk_ev[a + 1]));
      total += ms;
    }
    *ms_out = total;
    return SFW_OK;
  }
  int a = 0, b = 3;
  switch (which) {
    case 0: a = 0; b = 3; break;
    case 1: a = 0; b = 1; break;
    case 2: a = 1; b = 2; break;
    case 3: a = 2; b = 3; break;
    default: return fail(h, SFW_ERR_INVALID_ARG, "last_launch_ms: which must be 0..3");
  }
  SFW_HIP(h, hipEventElapsedTime(ms_out, h->ev[a], h->ev[b]));
  return SFW_OK;
}

int sfw_set_points_capture(sfw_handle h, int32_t enabled) {
  if (!h) return SFW_ERR_INVALID_ARG;
  h->capture_points = enabled != 0;
  return SFW_OK;
}

int sfw_last_clock_ghz(sfw_handle h, double *ghz_out) {
  if (!h || !ghz_out) return SFW_ERR_INVALID_ARG;
  if (!h->launched) return fail(h, SFW_ERR_STATE, "last_clock_ghz before grid_launch");
  if (!h->launched_timed) return fail(h, SFW_ERR_STATE, "last_clock_ghz: timing was off for the last launch (sfw_set_timing)");
  SFW_HIP(h, hipSetDevice(h->device));
  SFW_HIP(h, hipEventSynchronize(h->ev[3]));
  unsigned long long v[4] = {0, 0, 0, 0};
  SFW_HIP(h, hipMemcpy(v, h->clock.p, sizeof(v), hipMemcpyDeviceToHost));
  int wall_khz = 0;
  SFW_HIP(h, hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, h->device));
  *ghz_out = 0.0;  // no sample: no agents, or wave 0 had nothing to integrate
  if (v[2] > v[0] && v[3] > v[1] && wall_khz > 0)
    *ghz_out = static_cast<double>(v[2] - v[0]) / static_cast<double>(v[3] - v[1]) * wall_khz * 1e-6;
  return SFW_OK;
}

int sfw_grid_points_batch(sfw_handle h, int64_t first, int64_t count, double *points_xyth, int32_t *n_points) {
  if (!h) return SFW_ERR_INVALID_ARG;
  if (!h->staged) return fail(h, SFW_ERR_STATE, "grid_points before grid_stage");
  const int64_t T = static_cast<int64_t>(h->nv) * h->nw;
  if (first < 0 || count <= 0 || first + count > T || !points_xyth || !n_points)
    return fail(h, SFW_ERR_INVALID_ARG, "grid_points: bad range or buffer");
  SFW_HIP(h, hipSetDevice(h->device));
  if (int e = flush_arena(h)) return e;  // (a dump between stage and launch: the re-run below reads the device copy of the arena)
  const int S = num_steps_of(h->params);
  const size_t n = static_cast<size_t>(count);
  if (h->launched && h->captured && h->cap_S == S) {  // (sfw_set_params since the launch changed the step count: re-run below)
    // the scoring launch left everything in pinned memory (sfw_set_points_capture): points | counts | contact steps, no
    // kernel, no copy — only the wait for the launch, if no fetch has waited for it yet
    const size_t pts_bytes = sizeof(double) * 3 * static_cast<size_t>(S) * static_cast<size_t>(T);
    if (!h->fetched) {
      SFW_HIP(h, wait_stream(h));
      stream_is_idle(h);
    }
    const int32_t *np = reinterpret_cast<const int32_t *>(h->pin_cap.p + pts_bytes), *coll = np + T;
    std::memcpy(points_xyth, h->pin_cap.p + sizeof(double) * 3 * static_cast<size_t>(S) * static_cast<size_t>(first), sizeof(double) * 3 * S * n);
    for (size_t i = 0; i < n; ++i) {
      const int32_t c = coll[first + static_cast<int64_t>(i)];
      int32_t k = np[first + static_cast<int64_t>(i)];
      if (c >= 0 && c + 1 < k) k = c + 1;  // rejected by contact at step c: poses 0..c were added (ref :578, :613-627)
      n_points[i] = k;
    }
    return SFW_OK;
  }
  SFW_HIP(h, h->points.reserve(3 * static_cast<size_t>(S) * n));
  SFW_HIP(h, h->n_points.reserve(n));
  // Re-run K1 for those samples into scratch outputs so the grid results stay intact.  The scratch is
  // `count` records long and lives in the handle; the kernels index per-sample outputs by the global
  // sample index, so the pointers are biased by -first.
  hipError_t e = h->pts_status.reserve(n);
  if (e == hipSuccess) e = h->pts_base.reserve(n);
  if (e == hipSuccess) e = h->pts_costs.reserve(n);
  if (e == hipSuccess) e = h->pts_ptab.reserve(static_cast<size_t>(S) * static_cast<size_t>(table_row_units(count, h->nw)));
  if (e == hipSuccess) e = h->pts_cs.reserve(static_cast<size_t>(S) * h->nw);
  if (e == hipSuccess) e = h->pts_fcode.reserve(static_cast<size_t>(S) * n);
  if (e == hipSuccess) {
    sfw_launch L;
    fill_launch(h, L, first, count, count);
    L.status = h->pts_status.p - first;
    L.base_cost = h->pts_base.p - first;
    L.costs = h->pts_costs.p - first;
    L.coll_step = nullptr;  // keep the launch's contact steps
    L.ptab = h->pts_ptab.p;
    L.cs_tab = h->pts_cs.p;
    L.fcode = h->pts_fcode.p;
    L.points = h->points.p;
    L.n_points = h->n_points.p;
    e = sfw_launch_rollout(L, h->stream);
  }
  std::vector<int32_t> coll(n, -1);
  // a capturing launch wrote its contact steps into the capture buffer, not into coll_step (this path is then only reached
  // when the step count changed since: cap_S != S): the contact steps come from the re-integration below instead
  const bool coll_elsewhere = h->launched && h->captured;
  if (e == hipSuccess && h->launched && !coll_elsewhere)
    e = hipMemcpyAsync(coll.data(), h->coll_step.p + first, sizeof(int32_t) * n, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess)
    e = hipMemcpyAsync(n_points, h->n_points.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess)
    e = hipMemcpyAsync(points_xyth, h->points.p, sizeof(double) * 3 * S * n, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) return hip_fail(h, e, "grid_points");
  // A sample whose footprint turns illegal at pose a was never integrated by the grid launch.  If a pedestrian
  // touches the robot at an earlier step b < a the reference's Trajectory ends there with b + 1 points (ref :613-627
  // return before pose a is reached): integrate the range once more with those samples kept alive and take their
  // contact steps from that run.
  bool rejected = false;
  for (size_t i = 0; i < n && !rejected; ++i) rejected = n_points[i] < S;
  if ((rejected || coll_elsewhere) && h->st_A > 1) {
    if (int rc = check_lds(h, count)) return rc;
    SFW_HIP(h, h->pts_coll.reserve(n));
    sfw_launch L;
    fill_launch(h, L, first, count, count);
    L.status = h->pts_status.p - first;
    L.base_cost = h->pts_base.p - first;
    L.costs = h->pts_costs.p - first;
    L.coll_step = h->pts_coll.p - first;
    L.ptab = h->pts_ptab.p;
    L.cs_tab = h->pts_cs.p;
    L.force_alive = 1;
    SFW_HIP(h, hipMemsetAsync(h->pts_coll.p, 0xff, sizeof(int32_t) * n, h->stream));  // -1: no contact
    SFW_HIP(h, launch_social_of(L, h->stream));
    SFW_HIP(h, hipMemcpyAsync(coll.data(), h->pts_coll.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, h->stream));
    SFW_HIP(h, hipStreamSynchronize(h->stream));
  }
  // a trajectory rejected by contact at step i holds the poses 0..i (addPoint precedes the test, ref :578, :613-627)
  for (size_t i = 0; i < n; ++i)
    if (coll[i] >= 0 && coll[i] + 1 < n_points[i]) n_points[i] = coll[i] + 1;
  return SFW_OK;
}

int sfw_grid_points(sfw_handle h, int64_t index, double *points_xyth, int32_t points_cap, int32_t *n_points) {
  if (!h) return SFW_ERR_INVALID_ARG;
  if (!points_xyth || points_cap <= 0) return fail(h, SFW_ERR_INVALID_ARG, "grid_points: bad index or buffer");
  const int S = num_steps_of(h->params);
  std::vector<double> tmp(static_cast<size_t>(3) * S);
  int32_t n = 0;
  if (int rc = sfw_grid_points_batch(h, index, 1, tmp.data(), &n)) return rc;
  const int m = n < points_cap ? n : points_cap;
  std::memcpy(points_xyth, tmp.data(), sizeof(double) * 3 * static_cast<size_t>(m));
  if (n_points) *n_points = n;
  return SFW_OK;
}

void *sfw_stream(sfw_handle h) { return h ? static_cast<void *>(h->stream) : nullptr; }

}  // extern "C"

// ===========================================================================
// one process, several devices (sfw_multi_*)
// ===========================================================================
namespace {
// The six RCCL entry points this file uses, declared here (values as in <rccl/rccl.h> of ROCm 7.2, NCCL 2.x ABI) so
// that the library builds on a ROCm install without the rccl development package and a single-device caller never
// needs librccl at all: it is resolved with dlopen on the first sfw_multi_create(SFW_MULTI_RCCL).
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;
constexpr ncclResult_t ncclSuccess = 0;
constexpr int ncclDouble = 8;  // ncclDataType_t: ncclFloat64
constexpr int ncclMin = 3;     // ncclRedOp_t
struct rccl_api {
  void *lib = nullptr;
  std::string error;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  // optional (sfw_multi_describe): what the communicators say about themselves
  ncclResult_t (*GetVersion)(int *) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommCuDevice)(const ncclComm_t, int *) = nullptr;
  std::string path;      // the file ncclAllReduce really came from (dladdr)
  std::string how;       // "SFW_RCCL_LIB", "already mapped", or the name dlopen was given
  // Which RCCL?  (1) SFW_RCCL_LIB in the environment: that file, or an error — never a silent fall-through; (2) an RCCL
  // this process has mapped already (a Python process that imported torch holds torch's bundled copy: a second RCCL beside
  // it would mean two sets of communicator state in one process) — found by walking the loaded objects, reopened with
  // RTLD_NOLOAD; (3) the loader's search: librccl.so, librccl.so.1, /opt/rocm/lib/librccl.so.
  void load_once() {
    if (const char *forced = std::getenv("SFW_RCCL_LIB")) {
      if (forced[0]) {
        lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        if (!lib) {
          error = std::string("SFW_RCCL_LIB=") + forced + ": " + dlerror();
          return;
        }
        how = "SFW_RCCL_LIB";
      }
    }
    if (!lib) {
      std::string mapped;
      dl_iterate_phdr(
          [](struct dl_phdr_info *info, size_t, void *out) {
            if (info->dlpi_name && std::strstr(info->dlpi_name, "librccl.so")) {
              *static_cast<std::string *>(out) = info->dlpi_name;
              return 1;
            }
            return 0;
          },
          &mapped);
      if (!mapped.empty()) {
        lib = dlopen(mapped.c_str(), RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
        if (lib) how = "already mapped";
      }
    }
    if (!lib)
      for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (lib) {
          how = name;
          break;
        }
      }
    if (!lib) {
      error = std::string("cannot load librccl.so: ") + dlerror();
      return;
    }
    auto sym = [&](const char *n) { return dlsym(lib, n); };
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
    AllReduce = reinterpret_cast<decltype(AllReduce)>(sym("ncclAllReduce"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
    GetVersion = reinterpret_cast<decltype(GetVersion)>(sym("ncclGetVersion"));
    CommCount = reinterpret_cast<decltype(CommCount)>(sym("ncclCommCount"));
    CommCuDevice = reinterpret_cast<decltype(CommCuDevice)>(sym("ncclCommCuDevice"));
    if (!CommInitAll || !CommDestroy || !AllReduce || !GroupStart || !GroupEnd || !GetErrorString) {
      dlclose(lib);
      lib = nullptr;
      error = "librccl.so lacks ncclCommInitAll/ncclAllReduce/ncclGroupStart/...";
      return;
    }
    Dl_info di;
    if (dladdr(reinterpret_cast<void *>(AllReduce), &di) && di.dli_fname) path = di.dli_fname;
  }
  // "" or why RCCL is not available; safe from several threads (two planners created concurrently)
  const std::string &load() {
    std::call_once(once, [this] { load_once(); });
    return error;
  }
  std::once_flag once;
};
rccl_api g_rccl;
thread_local std::string g_multi_create_error;

// One worker thread per rank of a multi handle, alive for the handle's lifetime: a rank's stage + launch is ~100 us
// of host work (shared-prefix planning, one H2D copy, a dozen kernel launches) and the ranks are independent, so a
// grid cut R ways is enqueued in the time of one rank instead of R.  run() hands every worker its job and returns
// when all are done.
struct rank_workers {
  struct slot {
    std::thread th;
    std::function<int()> job;
    bool has_job = false, quit = false;
    int rc = 0;
  };
  std::mutex mu;
  std::condition_variable cv_job, cv_done;
  std::vector<slot> w;
  int pending = 0;
  void start(int R) {
    w.resize(static_cast<size_t>(R));
    for (int r = 0; r < R; ++r) w[static_cast<size_t>(r)].th = std::thread([this, r] { loop(r); });
  }
  void loop(int r) {
    slot &me = w[static_cast<size_t>(r)];
    for (;;) {
      std::function<int()> job;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_job.wait(lk, [&] { return me.has_job || me.quit; });
        if (me.quit) return;
        job = std::move(me.job);
        me.has_job = false;
      }
      const int rc = job();
      {
        std::lock_guard<std::mutex> lk(mu);
        me.rc = rc;
        if (--pending == 0) cv_done.notify_all();
      }
    }
  }
  // jobs[r] runs on worker r; returns the first non-zero result (by rank), or 0
  int run(std::vector<std::function<int()>> &jobs, int *failed_rank) {
    {
      std::lock_guard<std::mutex> lk(mu);
      pending = static_cast<int>(jobs.size());
      for (size_t r = 0; r < jobs.size(); ++r) {
        w[r].job = std::move(jobs[r]);
        w[r].has_job = true;
      }
    }
    cv_job.notify_all();
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [&] { return pending == 0; });
    for (size_t r = 0; r < w.size(); ++r)
      if (w[r].rc != 0) {
        if (failed_rank) *failed_rank = static_cast<int>(r);
        return w[r].rc;
      }
    return 0;
  }
  void stop() {
    {
      std::lock_guard<std::mutex> lk(mu);
      for (slot &s : w) s.quit = true;
    }
    cv_job.notify_all();
    for (slot &s : w)
      if (s.th.joinable()) s.th.join();
    w.clear();
  }
};
}  // namespace

struct sfw_multi_s {
  int R = 0;
  int exchange = SFW_MULTI_RCCL;
  std::vector<sfw_handle> h;
  std::vector<int> dev;
  std::vector<ncclComm_t> comm;       // RCCL exchange only
  std::vector<double *> d_table;      // per rank: [R,5] doubles on its device
  double *pin_table = nullptr;        // pinned host copy: the reduced table (RCCL) or the ranks' own rows (host reduce)
  std::vector<int32_t> row0;          // R + 1 row offsets of the last grid
  std::vector<double> lin, ang;
  int32_t nv = 0, nw = 0;
  bool scored = false;
  double us[3] = {0, 0, 0};
  std::string err;
  rank_workers workers;               // R > 1 only
  std::vector<std::string> rank_err;  // what went wrong on a rank's worker (its handle's message, or a HIP error)
};

#define SFW_MHIP(m, h, call)                                                                  \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      (m)->err = std::string(#call) + ": " + hipGetErrorString(e_);                           \
      return hip_fail((h), e_, #call);                                                        \
    }                                                                                         \
  } while (0)

namespace {
int mfail(sfw_multi_handle m, int code, const std::string &msg) {
  if (m) m->err = msg;
  return code;
}
int mrank_fail(sfw_multi_handle m, int r, int code, const char *what) {
  return mfail(m, code, std::string(what) + " on rank " + std::to_string(r) + ": " + sfw_last_error(m->h[static_cast<size_t>(r)]));
}
double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

extern "C" {

int sfw_multi_create(const sfw_params *params, const int *devices, int32_t R, int32_t exchange, sfw_multi_handle *out) {
  if (!out) return SFW_ERR_INVALID_ARG;
  *out = nullptr;
  if (!params || !devices || R <= 0 || R > 64 || (exchange != SFW_MULTI_RCCL && exchange != SFW_MULTI_HOST_REDUCE))
    return SFW_ERR_INVALID_ARG;
  if (exchange == SFW_MULTI_RCCL)
    for (int a = 0; a < R; ++a)
      for (int b = a + 1; b < R; ++b)
        if (devices[a] == devices[b]) return SFW_ERR_INVALID_ARG;  // one RCCL rank per device
  sfw_multi_handle m = new (std::nothrow) sfw_multi_s();
  if (!m) return SFW_ERR_HIP;
  m->R = R;
  m->exchange = exchange;
  m->dev.assign(devices, devices + R);
  m->d_table.assign(static_cast<size_t>(R), nullptr);
  m->rank_err.assign(static_cast<size_t>(R), "");
  int rc = SFW_OK;
  for (int r = 0; r < R && rc == SFW_OK; ++r) {
    sfw_handle h = nullptr;
    rc = sfw_create(params, devices[r], &h);
    if (rc == SFW_OK) {
      m->h.push_back(h);
      if (hipSetDevice(devices[r]) != hipSuccess ||
          hipMalloc(reinterpret_cast<void **>(&m->d_table[static_cast<size_t>(r)]), sizeof(double) * 5 * R) != hipSuccess)
        rc = SFW_ERR_HIP;
    }
  }
  if (rc == SFW_OK && hipHostMalloc(reinterpret_cast<void **>(&m->pin_table), sizeof(double) * 5 * R, hipHostMallocDefault) != hipSuccess)
    rc = SFW_ERR_HIP;
  if (rc == SFW_OK && exchange == SFW_MULTI_RCCL) {
    if (!g_rccl.load().empty()) {
      rc = SFW_ERR_UNSUPPORTED;
      g_multi_create_error = g_rccl.load();
    } else {
      m->comm.assign(static_cast<size_t>(R), nullptr);
      const ncclResult_t nr = g_rccl.CommInitAll(m->comm.data(), R, devices);
      if (nr != ncclSuccess) {
        m->comm.clear();
        rc = SFW_ERR_HIP;
        g_multi_create_error = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(nr) + " (" + g_rccl.path + ")";
      }
    }
  }
  if (rc != SFW_OK) {
    sfw_multi_destroy(m);
    return rc;
  }
  if (R > 1) m->workers.start(R);
  *out = m;
  return SFW_OK;
}

int sfw_multi_destroy(sfw_multi_handle m) {
  if (!m) return SFW_OK;
  m->workers.stop();
  for (size_t r = 0; r < m->h.size(); ++r) (void)sfw_grid_sync(m->h[r]);
  for (ncclComm_t c : m->comm)
    if (c) (void)g_rccl.CommDestroy(c);
  for (size_t r = 0; r < m->d_table.size(); ++r)
    if (m->d_table[r]) {
      (void)hipSetDevice(m->dev[r]);
      (void)hipFree(m->d_table[r]);
    }
  if (m->pin_table) (void)hipHostFree(m->pin_table);
  for (sfw_handle h : m->h) (void)sfw_destroy(h);
  delete m;
  return SFW_OK;
}

// (a NULL handle: why this thread's last sfw_multi_create failed, when it said why — RCCL could not be resolved or initialised)
const char *sfw_multi_last_error(sfw_multi_handle m) {
  if (m) return m->err.c_str();
  return g_multi_create_error.empty() ? "null handle" : g_multi_create_error.c_str();
}
int32_t sfw_multi_ranks(sfw_multi_handle m) { return m ? m->R : 0; }
int sfw_multi_describe(sfw_multi_handle m, sfw_multi_desc *out) {
  if (!m || !out) return SFW_ERR_INVALID_ARG;
  std::memset(out, 0, sizeof(*out));
  out->ranks = m->R;
  out->exchange = m->exchange;
  for (int r = 0; r < m->R && r < 64; ++r) out->devices[r] = m->dev[static_cast<size_t>(r)];
  for (int r = 0; r < 64; ++r) out->comm_devices[r] = -1;
  out->comm_size = -1;
  if (m->exchange == SFW_MULTI_RCCL) {
    if (g_rccl.GetVersion) (void)g_rccl.GetVersion(&out->rccl_version);
    std::snprintf(out->rccl_path, sizeof(out->rccl_path), "%s", g_rccl.path.c_str());
    std::snprintf(out->rccl_found, sizeof(out->rccl_found), "%s", g_rccl.how.c_str());
    for (size_t r = 0; r < m->comm.size() && r < 64; ++r)
      if (m->comm[r]) {
        out->communicators += 1;
        if (g_rccl.CommCuDevice) (void)g_rccl.CommCuDevice(m->comm[r], &out->comm_devices[r]);
        if (r == 0 && g_rccl.CommCount) (void)g_rccl.CommCount(m->comm[r], &out->comm_size);
      }
  }
  return SFW_OK;
}
sfw_handle sfw_multi_rank_handle(sfw_multi_handle m, int32_t r) {
  return (m && r >= 0 && r < m->R) ? m->h[static_cast<size_t>(r)] : nullptr;
}

int sfw_multi_set_params(sfw_multi_handle m, const sfw_params *params) {
  if (!m) return SFW_ERR_INVALID_ARG;
  for (int r = 0; r < m->R; ++r)
    if (int e = sfw_set_params(m->h[static_cast<size_t>(r)], params)) return mrank_fail(m, r, e, "sfw_set_params");
  return SFW_OK;
}
int sfw_multi_set_costmap(sfw_multi_handle m, const uint8_t *cells, uint32_t size_x, uint32_t size_y, double origin_x,
                          double origin_y, double resolution) {
  if (!m) return SFW_ERR_INVALID_ARG;
  for (int r = 0; r < m->R; ++r)
    if (int e = sfw_set_costmap(m->h[static_cast<size_t>(r)], cells, size_x, size_y, origin_x, origin_y, resolution))
      return mrank_fail(m, r, e, "sfw_set_costmap");
  return SFW_OK;
}
int sfw_multi_set_footprint(sfw_multi_handle m, const double *xy, int32_t K) {
  if (!m) return SFW_ERR_INVALID_ARG;
  for (int r = 0; r < m->R; ++r)
    if (int e = sfw_set_footprint(m->h[static_cast<size_t>(r)], xy, K)) return mrank_fail(m, r, e, "sfw_set_footprint");
  return SFW_OK;
}
int sfw_multi_set_agents(sfw_multi_handle m, const sfw_agent *agents, int32_t A, const double *obstacles_xy, int32_t O) {
  if (!m) return SFW_ERR_INVALID_ARG;
  // validated, packed and scanned for w x diff == 0 pairs ONCE (the scan is all pairs); the other ranks take rank 0's
  // host-side copy — the upload happens per rank with its next stage
  sfw_handle h0 = m->h[0];
  if (int e = sfw_set_agents(h0, agents, A, obstacles_xy, O)) return mrank_fail(m, 0, e, "sfw_set_agents");
  for (int r = 1; r < m->R; ++r) {
    sfw_handle h = m->h[static_cast<size_t>(r)];
    h->h_agents = h0->h_agents;
    h->ao_vel = h0->ao_vel; h->ao_cst = h0->ao_cst; h->ao_obs = h0->ao_obs; h->ao_grp = h0->ao_grp; h->ao_off = h0->ao_off; h->ao_mem = h0->ao_mem;
    h->A = h0->A; h->O = h0->O; h->NG = h0->NG; h->n_grp_mem = h0->n_grp_mem;
    h->rest_pairs = h0->rest_pairs;
  }
  return SFW_OK;
}

int sfw_multi_score_grid(sfw_multi_handle m, const sfw_robot_state *rs, const double *linvels, int32_t nv,
                         const double *angvels, int32_t nw, const sfw_goal_args *args, double *costs_out,
                         sfw_best *best_out) {
  if (!m) return SFW_ERR_INVALID_ARG;
  if (!rs || !linvels || !angvels || !args || nv <= 0 || nw <= 0)
    return mfail(m, SFW_ERR_INVALID_ARG, "multi_score_grid: null pointer or non-positive sample count");
  const int R = m->R;
  m->scored = false;
  m->lin.assign(linvels, linvels + nv);
  m->ang.assign(angvels, angvels + nw);
  m->nv = nv;
  m->nw = nw;
  const double t0 = now_us();
  sfw_handle h0 = m->h[0];
  if (!all_finite(linvels, static_cast<size_t>(nv)) || !all_finite(angvels, static_cast<size_t>(nw)) || !all_finite(&rs->x, 6) ||
      !all_finite(&args->acc_x, 5))
    return mfail(m, SFW_ERR_INVALID_ARG, "multi_score_grid: non-finite robot state, goal argument or sample velocity");
  {  // ref :345: rows are the outer axis — contiguous blocks of equal planned work (equal row counts when nothing is shared)
    const int S = num_steps_of(h0->params);
    const bool sharing = h0->prefix_env.empty() || h0->prefix_env[0] != 0;
    plan_row_blocks(m->lin, m->ang, rs->vx, rs->vtheta, args->acc_x, args->acc_theta, h0->params.sim_time / S, S, sharing ? h0->A : 0, R,
                    h0->k2_form, h0->n_cu, m->row0);
  }
  // (0) what is the same for every rank, once: the agent set must fit a wave's LDS for every rank's item count
  // before anything is launched anywhere, and the classes of the column axis (every rank scores all nw columns)
  for (int r = 0; r < R; ++r) {
    const int64_t items = static_cast<int64_t>(m->row0[static_cast<size_t>(r) + 1] - m->row0[static_cast<size_t>(r)]) * nw;
    if (items > 0 && h0->A > 0 && sfw_social_lds_bytes(h0->A, h0->O, h0->NG, h0->n_grp_mem, items, h0->k2_form, h0->n_cu) > 160 * 1024)
      return mfail(m, SFW_ERR_UNSUPPORTED, "agent/obstacle set does not fit the 160 KiB LDS of one CU");
  }
  axis_classes cols;
  const axis_classes *shared = nullptr;
  {
    const int S = num_steps_of(h0->params);
    const int64_t rows_max = (nv + R - 1) / R;
    if (R > 1 && h0->A >= 2 && S >= 2 && rows_max * nw >= 4096 && (h0->prefix_env.empty() || h0->prefix_env[0] != 0)) {
      cols = classes_of_axis(m->ang, rs->vtheta, args->acc_theta, h0->params.sim_time / S, prefix_max_p(S));
      cols.complete();  // (read by every rank's thread: no level is written lazily from there)
      shared = &cols;
    }
  }
  // (1) every rank: stage its block of rows, enqueue the kernels and its row of the exchange table
  auto rank_job = [&](int r) -> int {
    sfw_handle h = m->h[static_cast<size_t>(r)];
    std::string &err = m->rank_err[static_cast<size_t>(r)];
    err.clear();
    const int32_t lo = m->row0[static_cast<size_t>(r)], n = m->row0[static_cast<size_t>(r) + 1] - lo;
    hipError_t e = hipSetDevice(m->dev[static_cast<size_t>(r)]);
    if (e != hipSuccess) {
      err = std::string("hipSetDevice: ") + hipGetErrorString(e);
      return SFW_ERR_HIP;
    }
    if (n > 0) {
      h->shared_cols = shared;
      const int rc_stage = sfw_grid_stage(h, rs, linvels + lo, n, angvels, nw, args, static_cast<int64_t>(lo) * nw);
      h->shared_cols = nullptr;
      if (rc_stage) {
        err = std::string("sfw_grid_stage: ") + sfw_last_error(h);
        return rc_stage;
      }
      if (int rc = sfw_grid_launch(h)) {
        err = std::string("sfw_grid_launch: ") + sfw_last_error(h);
        return rc;
      }
    }
    // fewer rows than ranks: this rank holds nothing, its row is +inf / 0 valid (null selection record)
    e = sfw_launch_key_table(n > 0 ? h->d_sel : nullptr, m->d_table[static_cast<size_t>(r)], r, R, h->stream);
    if (e != hipSuccess) {
      err = std::string("key table launch: ") + hipGetErrorString(e);
      return SFW_ERR_HIP;
    }
    return SFW_OK;
  };
  {
    int failed = 0, rc = SFW_OK;
    if (R == 1) {
      rc = rank_job(0);
    } else {
      std::vector<std::function<int()>> jobs;
      for (int r = 0; r < R; ++r) jobs.emplace_back([&rank_job, r] { return rank_job(r); });
      rc = m->workers.run(jobs, &failed);
    }
    if (rc != SFW_OK) return mfail(m, rc, m->rank_err[static_cast<size_t>(failed)] + " (rank " + std::to_string(failed) + ")");
  }
  const double t1 = now_us();
  // (2) the exchange
  if (m->exchange == SFW_MULTI_RCCL) {
    ncclResult_t nr = g_rccl.GroupStart();
    for (int r = 0; r < R && nr == ncclSuccess; ++r) {
      (void)hipSetDevice(m->dev[static_cast<size_t>(r)]);
      double *t = m->d_table[static_cast<size_t>(r)];
      nr = g_rccl.AllReduce(t, t, static_cast<size_t>(5) * R, ncclDouble, ncclMin, m->comm[static_cast<size_t>(r)],
                            m->h[static_cast<size_t>(r)]->stream);
    }
    const ncclResult_t ne = g_rccl.GroupEnd();
    if (nr == ncclSuccess) nr = ne;
    if (nr != ncclSuccess) return mfail(m, SFW_ERR_HIP, std::string("ncclAllReduce: ") + g_rccl.GetErrorString(nr));
    SFW_MHIP(m, h0, hipSetDevice(m->dev[0]));
    SFW_MHIP(m, h0, hipMemcpyAsync(m->pin_table, m->d_table[0], sizeof(double) * 5 * R, hipMemcpyDeviceToHost, h0->stream));
    SFW_MHIP(m, h0, hipStreamSynchronize(h0->stream));
  } else {  // host reduce: each rank's own row
    for (int r = 0; r < R; ++r) {
      sfw_handle h = m->h[static_cast<size_t>(r)];
      SFW_MHIP(m, h, hipSetDevice(m->dev[static_cast<size_t>(r)]));
      SFW_MHIP(m, h, hipMemcpyAsync(m->pin_table + 5 * r, m->d_table[static_cast<size_t>(r)] + 5 * r, sizeof(double) * 5,
                                hipMemcpyDeviceToHost, h->stream));
    }
    for (int r = 0; r < R; ++r) {
      sfw_handle h = m->h[static_cast<size_t>(r)];
      SFW_MHIP(m, h, hipSetDevice(m->dev[static_cast<size_t>(r)]));
      SFW_MHIP(m, h, hipStreamSynchronize(h->stream));
    }
  }
  const double t2 = now_us();
  // (3) cost slices
  if (costs_out)
    for (int r = 0; r < R; ++r) {
      const int32_t lo = m->row0[static_cast<size_t>(r)], n = m->row0[static_cast<size_t>(r) + 1] - lo;
      if (n > 0)
        if (int e = sfw_grid_fetch(m->h[static_cast<size_t>(r)], costs_out + static_cast<int64_t>(lo) * nw, nullptr, nullptr))
          return mrank_fail(m, r, e, "sfw_grid_fetch");
    }
  const double t3 = now_us();
  m->us[0] = t1 - t0;
  m->us[1] = t2 - t1;
  m->us[2] = t3 - t2;
  // (4) lexicographic minimum over the rows = the reference's selection order
  if (best_out) {
    int win = -1;
    int64_t n_valid = 0;
    for (int r = 0; r < R; ++r) {
      const double *k = m->pin_table + 5 * r;
      n_valid += static_cast<int64_t>(k[4]);
      if (!std::isfinite(k[0])) continue;
      if (win < 0 || std::lexicographical_compare(k, k + 4, m->pin_table + 5 * win, m->pin_table + 5 * win + 4)) win = r;
    }
    best_out->n_valid = n_valid;
    if (win >= 0) {
      const double *k = m->pin_table + 5 * win;
      const int64_t idx = static_cast<int64_t>(-k[3]);
      best_out->index = idx;
      best_out->cost = k[0];
      best_out->vx = linvels[idx / nw];
      best_out->vy = 0.0;
      best_out->vtheta = angvels[idx % nw];
    } else {  // ref :456-468
      best_out->index = -1;
      best_out->cost = -1.0;
      best_out->vx = best_out->vy = best_out->vtheta = 0.0;
    }
  }
  m->scored = true;
  return SFW_OK;
}

int sfw_multi_rank_rows(sfw_multi_handle m, int32_t r, int32_t *first_row, int32_t *n_rows) {
  if (!m || r < 0 || r >= m->R || !first_row || !n_rows) return SFW_ERR_INVALID_ARG;
  if (!m->scored) return mfail(m, SFW_ERR_STATE, "multi_rank_rows before multi_score_grid");
  *first_row = m->row0[static_cast<size_t>(r)];
  *n_rows = m->row0[static_cast<size_t>(r) + 1] - m->row0[static_cast<size_t>(r)];
  return SFW_OK;
}

int sfw_multi_last_us(sfw_multi_handle m, int32_t which, double *us_out) {
  if (!m || !us_out || which < 0 || which > 2) return SFW_ERR_INVALID_ARG;
  if (!m->scored) return mfail(m, SFW_ERR_STATE, "multi_last_us before multi_score_grid");
  *us_out = m->us[which];
  return SFW_OK;
}

int sfw_multi_grid_points(sfw_multi_handle m, int64_t index, double *points_xyth, int32_t points_cap, int32_t *n_points) {
  if (!m) return SFW_ERR_INVALID_ARG;
  if (!m->scored) return mfail(m, SFW_ERR_STATE, "multi_grid_points before multi_score_grid");
  if (index < 0 || index >= static_cast<int64_t>(m->nv) * m->nw) return mfail(m, SFW_ERR_INVALID_ARG, "multi_grid_points: bad index");
  const int32_t row = static_cast<int32_t>(index / m->nw);
  int r = 0;
  while (r + 1 < m->R && row >= m->row0[static_cast<size_t>(r) + 1]) ++r;
  const int64_t local = index - static_cast<int64_t>(m->row0[static_cast<size_t>(r)]) * m->nw;
  if (int e = sfw_grid_points(m->h[static_cast<size_t>(r)], local, points_xyth, points_cap, n_points))
    return mrank_fail(m, r, e, "sfw_grid_points");
  return SFW_OK;
}

}  // extern "C"
