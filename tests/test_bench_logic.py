"""CPU: the host-side arithmetic of bench.py (no GPU): the CPU count behind the blocking-wait decision, the
configurations named after BASELINE.json."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_host_cpus_is_positive_and_bounded():
    n = bench.host_cpus()
    assert 1 <= n <= len(os.sched_getaffinity(0))


def test_configs_name_baseline_shapes():
    assert bench.MAIN["nvar"] + 1 == 128 and bench.MAIN["ni"] == 64 and bench.MAIN["batch"] == 10000
    keys = [c["key"] for c in bench.OTHERS]
    assert keys == ["configs[1]", "configs[4]"]
    assert bench.OTHERS[1]["ebits"] == 128 and bench.OTHERS[0]["integer"] is False
