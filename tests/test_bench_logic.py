"""CPU: the scheduling arithmetic of bench.py (no GPU): lanes that divide --steps, the CPU count behind the
blocking-wait decision, the strong-scaling shard sizes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_lane_count_divides_steps():
    assert bench.lane_count(12, 96) == 12
    assert bench.lane_count(12, 20) == 10      # the driver's --steps 20: ten lanes, two steps each
    assert bench.lane_count(24, 20) == 20
    assert bench.lane_count(12, 7) == 7
    assert bench.lane_count(12, 1) == 1
    for pipeline in (1, 4, 12, 24):
        for steps in range(1, 200):
            d = bench.lane_count(pipeline, steps)
            assert 1 <= d <= min(pipeline, steps)
            if steps % d:  # no divisor between depth/2 and depth: the full depth is kept
                assert d == min(pipeline, steps)


def test_host_cpus_is_positive_and_bounded():
    n = bench.host_cpus()
    assert 1 <= n <= len(os.sched_getaffinity(0))


def test_configs_name_baseline_shapes():
    assert bench.MAIN["nvar"] + 1 == 128 and bench.MAIN["ni"] == 64 and bench.MAIN["batch"] == 10000
    keys = [c["key"] for c in bench.OTHERS]
    assert keys == ["configs[1]", "configs[4]"]
    assert bench.OTHERS[1]["ebits"] == 128 and bench.OTHERS[0]["integer"] is False
