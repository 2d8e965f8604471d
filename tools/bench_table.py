#!/usr/bin/env python3
"""Markdown table of one default `python bench.py` line (headline + every entry of `extra`): what DESIGN.md §8 and BASELINE.md §4
quote.  usage: bench_table.py <bench.json>"""
import json
import sys

d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
ex = d.get("extra", {})


def vtxt(v):
    if not v:
        return "—"
    cov = "whole grid" if v["coverage"].startswith("the whole grid") else v["coverage"].split(" sub-grid")[0] + " + 32 lowest"
    return f"{v['max_rel_err']:.1e} over {v['samples']} samples ({cov}), sentinels {'=' if v['invalid_set_equal'] else 'DIFFER'}, cmd_vel {'match' if v['cmd_vel_match'] else 'MISMATCH'}"


rows = [("**target** 256×256, 50 ped, S = 40 (headline)", d["value"], d["ms_per_step"], d["kernel_ms"]["social"], d["sustained_clock_ghz"],
         d["roofline"]["frac"], d["roofline"]["executed_frac"], vtxt(d.get("verify")))]
names = {"cfg2": "cfg2 128×128, 20 ped, S = 40", "cfg2_o64": "cfg2 + 64 laser points", "cfg2_o240": "cfg2 + 240 laser points",
         "target_o720": "target + 720 laser points", "cfg3": "cfg3 256×256, 50 ped, S = 80", "cfg4": "cfg4 1024×1024, 200 ped (from 2.1 m)",
         "cfg4_spec": "cfg4 with §8d's crowd (from 0.8 m: every sample ends in a contact)"}
for k, label in names.items():
    e = ex.get(k)
    if e:
        rows.append((label, e["value"], e["ms_per_step"], e["kernel_ms"]["social"], e["sustained_clock_ghz"], e.get("roofline_frac"),
                     e.get("roofline_executed_frac"), vtxt(e.get("verify"))))
c5 = ex.get("cfg5_strong")
if c5:
    rf = c5.get("roofline_frac_rank0", {})
    rows.append((f"cfg5 4096×4096, 100 ped: the whole grid on ONE GPU ({c5.get('chunks_per_gpu', '?')} table chunk(s), {c5.get('steps', '?')} timed steps)", c5["value"], c5["ms_per_step"],
                 c5["per_rank"]["social_kernel_ms"][0], None, rf.get("frac"), rf.get("executed_frac"), "(suite: strided sub-grids)"))
for k, label in (("f64_strict_mode", "target, `SFW_PRECISION_F64_STRICT`"), ("f32_forces_mode", "target, `SFW_PRECISION_F32`")):
    e = ex.get(k)
    if e:
        rows.append((label, e["value"], None, e["kernel_ms"]["social"], None, None, None, vtxt(e.get("verify"))))
print("| Workload | trajectories/s (blocking call) | step ms | K2 ms | GHz | `roofline.frac` / executed | `verify` vs oracle |")
print("|---|---|---|---|---|---|---|")
f = lambda x, fmt: "—" if x is None else format(x, fmt)
for label, val, ms, k2, ghz, fr, xf, v in rows:
    print(f"| {label} | {val:.3g} | {f(ms, '.3f')} | {f(k2, '.3f')} | {f(ghz, '.2f')} | {f(fr, '.2f')} / {f(xf, '.2f')} | {v} |")
cb = d.get("cpu_baseline")
if cb:
    print(f"\nCPU oracle, same box ({cb.get('cpu_model', '?')}): {cb['value']:.0f} trajectories/s on 1 thread ({cb['sample']}), "
          f"{cb['all_cores']['value']:.0f} on {cb['all_cores']['cores']} threads.")
im = ex.get("inproc_multi")
if im:
    print("`sfw_multi_score_grid`, R host-reduce ranks on the one device (ms per call / enqueue µs / levels): " +
          ", ".join(f"R = {r}: {im[f'R{r}']['ms_per_call']:.2f} / {im[f'R{r}']['enqueue_us']:.0f} / {im[f'R{r}']['levels_rank0']}" for r in (1, 2, 4, 8)))
print(f"world upload {ex.get('world_upload_ms', 0):.3f} ms per cycle; value incl. it {d.get('value_incl_world_upload', 0):.3g}")
