#!/usr/bin/env python3
"""Prints the main numbers of a bench.py JSON line."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "roofline frac", d["roofline"]["frac"], "us/launch", d["roofline"]["us_per_launch"])
print("early_fetch", json.dumps(d.get("early_fetch")))
print("secondary", [(s["workload"][:24], s["value"], s.get("scan_roofline", {}).get("frac")) for s in d.get("secondary", [])])
print("batched", [(b["batch"], b["value"], b.get("fetch_launch", {}).get("pcie_gbs")) for b in d.get("batched", [])])
for k in ("value_call_order", "value_eager", "value_reference_layout", "value_resident_512", "fetch_launch"):
    v = d.get(k)
    print(k, None if v is None else {kk: v[kk] for kk in v if kk in ("value", "ms_per_step", "us_per_layer", "pcie_gbs", "fraction_of_eager_fused")})
print("pair", json.dumps(d.get("speedup_vs_full_attention"))[:400])
print("cpu_baseline", json.dumps(d.get("cpu_baseline"))[:300])
