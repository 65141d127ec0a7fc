#!/usr/bin/env python3
"""Condense rocprofv3 output into a short table (kernel names shortened) for profiles/.

  summarize_rocprof.py <*_kernel_stats.csv> [top]                       whole-process per-kernel table
  summarize_rocprof.py --graph-replays-only <*_kernel_trace.csv> --steps K [--json out.json] [top]
        STEADY STATE ONLY: the last K decode steps of the process (bench.py's timed region: hipGraph replays).  A step ends
        with its sampler launch (`skv_sample_topk_kernel`, once per token); the window runs from the end of the sampler that
        precedes the first of the last K steps to the end of the last sampler, so state building, warm-up, capture and the
        measurement legs that run before / after do not enter the averages or the percentages.  --json writes the in-step
        average of the dominant HBM kernel (`skv_score_tile_kernel`) for bench.py's `roofline.us_per_launch_in_step`.
"""
import csv
import json
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    if name.startswith("Cijk_"):
        m = re.search(r"MT(\d+x\d+x\d+)", name)
        return "hipBLASLt GEMV/GEMM Cijk_..." + (f"MT{m.group(1)}" if m else "")
    m = re.match(r"([\w:<>, ]+?)\(", name)
    base = m.group(1) if m else name
    if base.startswith("at::native::"):
        inner = re.search(r"at::native::(?:\(anonymous namespace\)::)?(\w+)", name)
        fn = re.findall(r"(\w+_kernel_cuda|\w+Functor\w*|\w+Ops\b|silu_kernel|normal_kernel|gatherTopK|cunn_SoftMaxForward\w*|vectorized_layer_norm_kernel)", name)
        return "torch " + (inner.group(1) if inner else "") + (":" + fn[0] if fn else "")
    return base[:90]


def main(path, top=40):
    rows = list(csv.DictReader(open(path)))
    tot = sum(int(r["TotalDurationNs"]) for r in rows)
    print(f"# source: {path}\n# total kernel time {tot/1e6:.2f} ms over {sum(int(r['Calls']) for r in rows)} launches")
    print(f"{'kernel':<92} {'calls':>6} {'avg_us':>9} {'min_us':>8} {'max_us':>8} {'total_ms':>9} {'pct':>6}")
    for r in rows[:top]:
        print(f"{short(r['Name']):<92} {int(r['Calls']):>6} {float(r['AverageNs'])/1e3:>9.2f} {int(r['MinNs'])/1e3:>8.2f} "
              f"{int(r['MaxNs'])/1e3:>8.2f} {int(r['TotalDurationNs'])/1e6:>9.2f} {float(r['Percentage']):>6.2f}")


def steady_state(path, steps, top=40, json_out=None, step_marker="skv_sample_topk", roof_kernel="skv_score_tile_kernel"):
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if step_marker in r[2]]
    if len(marks) < steps + 1:
        raise SystemExit(f"{path}: {len(marks)} '{step_marker}' launches, need {steps + 1} (K steps + the one that precedes them)")
    # (profile bench.py with --no-extras --no-cpu-baseline: the legs it runs after the timed loop launch no sampler, so the
    # last K sampler launches of the process are the timed loop's)
    t_open = rows[marks[-steps - 1]][1]
    t_close = rows[marks[-1]][1]
    win = [r for r in rows if r[0] >= t_open and r[1] <= t_close]
    agg = {}
    for s0, s1, name in win:
        a = agg.setdefault(short(name), [0, 0, 10 ** 18, 0])
        d = s1 - s0
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    wall = t_close - t_open
    print(f"# source: {path}\n# steady state only: the last {steps} decode steps ({len(win)} launches between two sampler launches "
          f"{steps} steps apart)\n# wall {wall / 1e6:.3f} ms = {wall / steps / 1e3:.1f} us per step; kernel time {tot / 1e6:.3f} ms = "
          f"{tot / steps / 1e3:.1f} us per step ({100.0 * tot / wall:.1f} % of the wall time: the rest is launch boundaries)")
    print(f"{'kernel':<92} {'calls':>6} {'per_step':>8} {'avg_us':>9} {'min_us':>8} {'max_us':>8} {'us/step':>9} {'pct':>6}")
    order = sorted(agg.items(), key=lambda kv: -kv[1][1])
    for name, (n, t, lo, hi) in order[:top]:
        print(f"{name:<92} {n:>6} {n / steps:>8.1f} {t / n / 1e3:>9.2f} {lo / 1e3:>8.2f} {hi / 1e3:>8.2f} {t / steps / 1e3:>9.2f} "
              f"{100.0 * t / tot:>6.2f}")
    if json_out:
        hit = [(name, v) for name, v in order if roof_kernel in name]
        if hit:
            name, (n, t, lo, hi) = hit[0]
            with open(json_out, "w") as f:
                json.dump({"kernel": name, "us_per_launch_in_step": round(t / n / 1e3, 3), "launches": n, "steps": steps,
                           "min_us": round(lo / 1e3, 3), "max_us": round(hi / 1e3, 3),
                           "us_per_step_all_kernels": round(tot / steps / 1e3, 1), "us_per_step_wall": round(wall / steps / 1e3, 1),
                           "source": "rocprofv3 --kernel-trace of bench.py, steady-state window (tools/summarize_rocprof.py "
                                     "--graph-replays-only)"}, f, indent=1)
                f.write("\n")


if __name__ == "__main__":
    a = sys.argv[1:]
    if a and a[0] == "--graph-replays-only":
        steps = int(a[a.index("--steps") + 1])
        jo = a[a.index("--json") + 1] if "--json" in a else None
        rest = [x for i, x in enumerate(a[1:]) if x not in ("--steps", "--json") and a[i] not in ("--steps", "--json")]
        steady_state(rest[0], steps, int(rest[1]) if len(rest) > 1 else 40, jo)
    else:
        main(a[0], int(a[1]) if len(a) > 1 else 40)
