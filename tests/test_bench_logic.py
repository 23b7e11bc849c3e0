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


def test_pick_fuse_targets_ten_thousand_tableaux_in_whole_passes():
    # a whole batch: no fusing; shards of a strong-scaling run: about 10,000 tableaux per launch sequence
    assert bench.pick_fuse(10000, 20) == 1
    assert bench.pick_fuse(5000, 20) == 2 and bench.pick_fuse(2500, 20) == 4
    assert bench.pick_fuse(1250, 192) == 8 and bench.pick_fuse(1250, 96) == 8
    # the driver's 20 steps at the N = 8 shard size: the nearest divisor of the step count below 8 (four passes of five)
    assert bench.pick_fuse(1250, 20) == 5
    for shard in (1, 7, 300, 1250, 3334, 9999, 10000, 50000):
        for steps in (1, 5, 7, 20, 96):
            f = bench.pick_fuse(shard, steps)
            assert 1 <= f <= 16 and (f * shard <= 2 * 10000 or f == 1)
