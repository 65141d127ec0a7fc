"""Child entry for tests/test_dist_cpu.py::test_plain_bench_gpus_2_launches_its_own_replicas: bench.main() with the stub seams
(no GPU work, gloo instead of RCCL) - what bench.launch_replicas starts instead of bench.py itself (bench.CHILD_ENTRY)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import bench                    # noqa: E402
import test_dist_cpu as T       # noqa: E402

if __name__ == "__main__":
    T._stub_bench(bench, int(os.environ["RANK"]))
    if os.environ.get("SKV_TEST_FAIL_RANK") == os.environ["RANK"]:
        sys.exit(7)
    bench.main(sys.argv[1:])
