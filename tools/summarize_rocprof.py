#!/usr/bin/env python3
"""Condense a rocprofv3 `*_kernel_stats.csv` into a short table (kernel names shortened) for profiles/."""
import csv
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


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
