// ref_driver.cpp — thin C shim over the TWO reference translation units that
// compile standalone in this image: the header-only
// LineIterator and src/trajectory.cpp.  The reference sources are compiled
// where they lie (-I$(REF)/include, $(REF)/src/trajectory.cpp); nothing is
// copied into this repo.  Output goes to oracle/_ref/ (git-ignored).
// Used only by tests/test_oracle_kat.py and tests/golden/make_golden.py to pin
// the oracle's Bresenham restatement against the real reference code.
#include <social_force_window_planner/line_iterator.hpp>
#include <social_force_window_planner/trajectory.hpp>

extern "C" {
// cells visited by LineIterator(x0,y0,x1,y1); returns the count
int ref_line_cells(int x0, int y0, int x1, int y1, int *xy_out, int cap) {
  int n = 0;
  for (social_force_window_planner::LineIterator it(x0, y0, x1, y1); it.isValid(); it.advance()) {
    if (n < cap) { xy_out[2 * n] = it.getX(); xy_out[2 * n + 1] = it.getY(); }
    ++n;
  }
  return n;
}
// Trajectory container round trip: add n points, read them back + endpoint.
int ref_trajectory_roundtrip(const double *xyth, int n, double *out_xyth, double *endpoint, double *cost0) {
  social_force_window_planner::Trajectory t;
  *cost0 = t.cost_;
  t.resetPoints();
  for (int i = 0; i < n; ++i) t.addPoint(xyth[3 * i], xyth[3 * i + 1], xyth[3 * i + 2]);
  for (unsigned i = 0; i < t.getPointsSize(); ++i)
    t.getPoint(i, out_xyth[3 * i], out_xyth[3 * i + 1], out_xyth[3 * i + 2]);
  if (n > 0) t.getEndpoint(endpoint[0], endpoint[1], endpoint[2]);
  return (int)t.getPointsSize();
}
}
