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


def test_gpus_n_spawns_one_rank_per_gpu_on_localhost():
    cmd = bench.spawn_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], 29555)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    assert cmd[-7].endswith("bench.py") and cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert 1024 < bench.free_port() < 65536


def test_screen_list_comes_from_the_oracle_and_names_the_known_tableaux():
    """tests/golden/bench_screen.json (CPU oracle, ORACLE_MAX_CUTS=448): 128 batches of the headline workload; the three
    slow-converging tableaux round 3 found with the engine are in it, found without it"""
    rec = bench.screen_records("bench_screen")
    assert len(rec) == 128 and all(set(r) >= {"seed", "slow", "pivots_screened", "pivots_finishing"} for r in rec.values())
    assert rec["0"]["slow"] == [] and rec["0"]["pivots_screened"] == 772044 and rec["0"]["seed"] == 1000
    assert rec["4"]["slow"] == [893] and rec["7"]["slow"] == [6225] and rec["11"]["slow"] == [4572]
    assert all(r["seed"] == 1000 + 7919 * int(g) for g, r in rec.items())
    w = bench.screen_records("wide128")
    assert set(w) == {"0"} and w["0"]["pivots_screened"] > 0


def test_replacement_is_the_next_tableau_that_is_not_slow():
    assert bench.replacement(893, {893}, 10000) == 894
    assert bench.replacement(9999, {9999, 0}, 10000) == 1
    assert bench.replacement(5, {5, 6, 7}, 10) == 8


def test_batch_rows_are_the_seeded_batches():
    import numpy as np
    from piplib_amd import synth
    cfg = dict(bench.MAIN, batch=50)
    assert np.array_equal(bench.batch_rows(cfg, 3), synth.lexmin_batch(1000 + 7919 * 3, 50, 127, 64))
    w = bench.batch_rows(bench.OTHERS[1], 0)
    assert w.shape == (1000, 128, 256) and np.array_equal(w, bench.batch_rows(bench.OTHERS[1], 5))
