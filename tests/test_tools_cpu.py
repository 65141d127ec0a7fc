"""CPU: the measurement tooling behind the numbers in the bench line (no GPU)."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_steady_state_summary_takes_exactly_the_last_k_steps(tmp_path):
    """tools/summarize_rocprof.py --graph-replays-only: the window runs from the end of the sampler launch that precedes the last
    K steps to the end of the last sampler; state building and warm-up before it and the measurement legs behind it stay out
    (VERDICT r4 weak #8).  A synthetic rocprofv3 kernel trace: 5 'build' launches, 6 steps of [scan x 2 layers, sampler],
    then 7 back-to-back scans of a measurement leg."""
    rows, t = [], 1000
    def launch(name, dur):
        nonlocal t
        rows.append({"Kind": "KERNEL_DISPATCH", "Kernel_Name": name, "Start_Timestamp": t, "End_Timestamp": t + dur})
        t += dur + 100
    for _ in range(5):
        launch("void at::native::build_kernel(int)", 50000)
    for step in range(6):
        for layer in range(2):
            launch("void skv_score_tile_kernel<4, 0, 16, 64>(unsigned short const*)", 9000 if step >= 3 else 20000)
        launch("void skv_sample_topk_kernel<16>(float*)", 17000)
    for _ in range(7):
        launch("void skv_score_tile_kernel<4, 0, 16, 64>(unsigned short const*)", 8000)
    trace = tmp_path / "x_kernel_trace.csv"
    with open(trace, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)
    out_json = tmp_path / "in_step.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_rocprof.py"), "--graph-replays-only", str(trace),
                        "--steps", "3", "--json", str(out_json)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    z = json.load(open(out_json))
    assert z["launches"] == 6 and z["steps"] == 3                      # 3 steps x 2 layers: the 20 us warm-up scans and the 8 us leg are out
    assert z["us_per_launch_in_step"] == 9.0 and "skv_score_tile_kernel" in z["kernel"]
    assert "build_kernel" not in r.stdout and "the last 3 decode steps (9 launches" in r.stdout
    # too few steps in the trace: refused, not silently shortened
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_rocprof.py"), "--graph-replays-only", str(trace),
                        "--steps", "6"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0
