/* minimal_c.c — the C ABI from a plain C99 caller: one control cycle of the reference's own 5 x 9 grid
 * (src/sfw_planner.cpp:64-85) against an empty 10 m x 10 m costmap with three pedestrians.
 *
 *   build: make -C social_force_window_planner_amd/csrc cdemo
 *   run:   build/minimal_c
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/sfw_hip.h"

int main(void) {
  enum { N = 200, NV = 5, NW = 9 };
  const double pi = 3.14159265358979323846;  /* strict C99: no M_PI */
  static uint8_t cells[N * N];
  double footprint[32], linvels[NV], angvels[NW], costs[NV * NW];
  sfw_agent agents[4];
  sfw_params params;
  sfw_handle h = NULL;
  sfw_best best;
  sfw_robot_state rs = {0.0, 0.0, 0.0, 0.3, 0.0, 0.0};
  sfw_goal_args ga = {1.0, 0.0, 1.0, 2.0, 0.5};
  int i, rc;

  sfw_params_default(&params);
  if (sfw_create(&params, 0, &h) != SFW_OK) {
    fprintf(stderr, "sfw_create failed: no HIP device (there is no CPU fallback)\n");
    return 2;
  }
  memset(cells, 0, sizeof(cells));
  for (i = 0; i < 16; ++i) {
    footprint[2 * i] = 0.35 * cos(i * pi / 8.0);
    footprint[2 * i + 1] = 0.35 * sin(i * pi / 8.0);
  }
  for (i = 0; i < NV; ++i) linvels[i] = i * params.max_vel_x / 4.0;
  angvels[0] = 0.0;
  for (i = 1; i <= 4; ++i) {
    angvels[2 * i - 1] = i * 0.125;
    angvels[2 * i] = -i * 0.125;
  }
  memset(agents, 0, sizeof(agents));
  agents[0].vx = 0.3;  /* the robot: local-frame twist, no goal (src/sensor_interface.cpp:552-580) */
  agents[0].desired_velocity = 0.7;
  agents[0].radius = 0.35;
  agents[0].group_id = -1;
  for (i = 1; i < 4; ++i) {
    agents[i].x = 2.0;
    agents[i].y = -1.5 + i;
    agents[i].vx = -0.8;
    agents[i].vy = 0.1 * (i - 2);
    agents[i].goal_x = agents[i].x + 2.0 * agents[i].vx;
    agents[i].goal_y = agents[i].y + 2.0 * agents[i].vy;
    agents[i].goal_radius = 0.35;
    agents[i].desired_velocity = 1.0;
    agents[i].radius = 0.35;
    agents[i].has_goal = 1;
    agents[i].id = i;
    agents[i].group_id = -1;
  }
  rc = sfw_set_costmap(h, cells, N, N, -5.0, -5.0, 0.05);
  if (rc == SFW_OK) rc = sfw_set_footprint(h, footprint, 16);
  if (rc == SFW_OK) rc = sfw_set_agents(h, agents, 4, NULL, 0);
  if (rc == SFW_OK) rc = sfw_score_grid(h, &rs, linvels, NV, angvels, NW, &ga, costs, &best);
  if (rc != SFW_OK) {
    fprintf(stderr, "error %d: %s\n", rc, sfw_last_error(h));
    sfw_destroy(h);
    return 1;
  }
  for (i = 0; i < NV * NW; ++i) printf("%s%8.3f", (i % NW) ? " " : "\n", costs[i]);
  printf("\nRESULT index=%lld vx=%.3f vtheta=%.3f cost=%.6f valid=%lld\n", (long long)best.index, best.vx, best.vtheta,
         best.cost, (long long)best.n_valid);
  sfw_destroy(h);
  return best.index >= 0 ? 0 : 1;
}
