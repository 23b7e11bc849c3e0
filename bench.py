#!/usr/bin/env python3
"""bench.py -- pivots/s of the HIP PIP engine on BASELINE.json's workloads.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Headline workload (BASELINE.json configs[2], the one the metric is quoted on): per GPU a batch of
10,000 synthetic 64x128 int64 tableaux (nvar = 127 unknowns, 64 inequality rows, constant
column; no parameters), integer solve with Gomory cuts, one workgroup per tableau.
A "step" = tab_get-style load of the batch into the HBM row store + the whole traiter()
pivot loop for every tableau + solution() (pipamd_batch_load, pipamd_batch_solve, pipamd_batch_results: status,
pivot and cut counts and the solutions into the caller's arrays); inputs are resident in HBM before the timed region.
Steps are pipelined: up to --pipeline (default 14) batches are in flight on separate HIP streams,
each lane with its own engine and workspace (each step is a complete load + solve + results of its
batch), all driven by ONE host thread through the asynchronous C ABI (pipamd_batch_solve_async /
pipamd_batch_poll; `--threads` = round 2's one-host-thread-per-lane driver).  The timed region is run three times; ms_per_step is
the median region / steps and all three are printed (`regions_ms`).  `pipeline1_value` is the same
workload with one batch at a time.  `other_configs` carries BASELINE configs[1] and configs[4]
measured the same way (shorter runs).
Multi-GPU (--gpus N > 1): BASELINE configs[3] -- every 10k-tableau batch is sharded over the ranks
(strong scaling), no data-path collective; a rank fuses its shards of several batches in flight into
one workspace (pipamd_batch_load_part) so that one launch sequence serves ~10,000 tableaux whatever N (pick_fuse);
RCCL only sums the totals here (the results gather is piplib_amd.dist.gather_results).  The
weak-scaling figure (10k tableaux per GPU and batch) is measured as well and reported as
`other_scaling`.  `python bench.py --gpus N` on its own starts the N ranks itself (a child
`python -m torch.distributed.run`, before this process touches the GPU); under a launcher the world size must be N.

What is timed is checked: which tableaux are left out of a batch (slow-converging cuts) comes from a committed list
made by the CPU oracle (tests/golden/bench_screen.json), the engine must agree with it, the pivots of the pre-pass
must be the oracle's, and after every timed region the status / pivot arrays the lanes fetched are reduced on the
device and compared with the pre-pass (`regions_checked`).  A mismatch voids the line (exit code 3).

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
STATUS_NAMES = {0: "run", 1: "solution", 2: "nil", 3: "need_compa", 4: "need_parmcut", 5: "overflow", 6: "capacity",
                7: "range", 8: "internal", 9: "maxcol"}

# name, tableaux per batch, unknowns, rows, integer solve?, entry bits, generator keywords
# batch g of a family is synth.lexmin_batch(seed_base + 7919 * g, ...) (see batch_rows)
MAIN = dict(key="configs[2]", workload="10k-batch synthetic 64x128 tableaux, int64, integer solve with Gomory cuts",
            batch=10000, nvar=127, ni=64, integer=True, ebits=64, gen={}, seed_base=1000, screen="bench_screen")
OTHERS = [
    dict(key="configs[1]", workload="1k-batch synthetic 32x64 tableaux, int64, rational (non-integer) solve",
         batch=1000, nvar=63, ni=32, integer=False, ebits=64, gen={}, seed_base=2000, screen=None),
    # The batch tests/test_gpu_parity.py::test_full_size_int128_config holds, tableau by tableau, against the outputs of
    # the reference's own GMP build (tests/golden/gmp/wide128.json): up to 16 non-zeros of magnitude <= 30 per row, 587 of
    # the 1,000 tableaux form entries beyond 2^63 (the int64 build stops with "Integer overflow" or wraps on them).
    # Every lane solves this one batch.
    dict(key="configs[4]", workload="1k-batch synthetic 128x256 tableaux, 128-bit Entier, integer solve: the `wide128` "
                                    "family pinned by tests/golden/gmp/wide128.json (587 of 1,000 tableaux leave 64 bits)",
         batch=1000, nvar=255, ni=128, integer=True, ebits=128, gen=dict(nnz=16, cmax=30), family="wide128",
         screen="wide128", screen_cuts=1024, max_rows=128 + 1280),  # every one of them finishes within 1,024 cuts
]
SCREEN_CUTS = 448  # a tableau on which integrer() asks for more constant cuts than this (cfg["screen_cuts"]) is left out


def screen_records(name):
    """tests/golden/bench_screen.json, made by the CPU oracle (tests/golden/make_bench_screen.py): batch index ->
    {slow, unfinished, pivots_screened}; {} when the file did not travel"""
    path = os.path.join(ROOT, "tests", "golden", "bench_screen.json")
    try:
        doc = json.load(open(path))
    except (OSError, ValueError):
        return {}
    return doc.get("batches" if name == "bench_screen" else name, {})


def batch_rows(cfg, g):
    """the whole batch g of a workload as a host array (batch, ni, nvar + 1), before anything is screened"""
    from piplib_amd import synth
    if cfg.get("family") == "wide128":
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import make_bigint_fixtures as mk
        return mk.rows_full("wide128")
    return synth.lexmin_batch(cfg["seed_base"] + 7919 * g, cfg["batch"], cfg["nvar"], cfg["ni"], **cfg["gen"])


def replacement(b, slow, n):
    """the tableau that stands in for slow tableau b: the next one (cyclically) that is not slow itself"""
    return next(x % n for x in range(b + 1, b + n) if x % n not in slow)


class VoidLine(Exception):
    """the workload that was timed is not the workload the line names: the line is void"""


def cpu_baseline(rows, nvar, ni):
    """The reference CPU path on a bounded sample of the headline workload, one process per host
    core: oracle/_ref/refpip_fast (the reference's five sources + our driver in one -O3 executable,
    no pivot counter in the timed path); pivots are counted in a second, untimed pass of the
    counting build.  Falls back to the CPU restatement ("port") if the reference build did not
    travel."""
    import concurrent.futures as cf
    import numpy as np
    import pipbatch as pb
    from piplib_amd import synth
    fast = pb.REFPIP + "_fast"
    if os.access(fast, os.X_OK) and pb.have_ref():
        exe, counter, kind = fast, pb.REFPIP, "reference"
    elif pb.have_ref():
        exe, counter, kind = pb.REFPIP, None, "reference"
    elif os.access(pb.ORACLEPIP, os.X_OK):
        exe, counter, kind = pb.ORACLEPIP, None, "port"
    else:
        return None
    cores = max(1, len(os.sched_getaffinity(0)))
    quota = None
    try:  # a container's CPU share (cgroup v2): more processes than that only take turns
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(int(q) / int(per)))
            cores = min(cores, quota)
    except (OSError, ValueError):
        pass
    # ~25k pivots per process (about 0.3-0.7 s each); all of the batch at most
    per = max(1, min(320, rows.shape[0] // cores))
    # on a many-core host a process's share of the batch is small: solve it `reps` times over, so
    # that every process runs for about a second (about 10-30 s of CPU work in all)
    reps = max(1, min(8, 1280 // per))
    chunks = [np.concatenate([rows[c * per:(c + 1) * per]] * reps) for c in range(cores)]

    def run_with(x):
        def run(chunk):
            probs = [synth.Problem(nvar, 0, ni, 0, -1, 1, chunk[b], np.zeros((0, 1), np.int64))
                     for b in range(chunk.shape[0])]
            return pb.run_batch(x, probs, pb.F_NOSIMPLIFY | pb.F_NOTEXT)
        with cf.ThreadPoolExecutor(cores) as ex:
            return list(ex.map(run, chunks))

    run_with(exe)  # warm the page cache / CPU clocks
    t0 = time.time()
    outs = run_with(exe)
    wall = time.time() - t0
    piv = sum(o.total_pivots for o in (run_with(counter) if counter else outs))
    tmax = max(o.solve_seconds for o in outs)
    model = "?"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": piv / tmax, "unit": "pivots/s", "cores": cores, "kind": kind, "cpu": model,
            "cpus_in_affinity_mask": len(os.sched_getaffinity(0)), "cgroup_cpu_quota": quota,
            "build": "gcc -O3 -fomit-frame-pointer, one executable (oracle/Makefile refpip_fast)" if exe == fast else "see oracle/Makefile",
            "sample": f"first {per * cores} tableaux of rank 0's first batch, {cores} processes x {per} tableaux x "
                      f"{reps} repeats ({piv} pivots), slowest process {tmax:.2f} s of traiter() time "
                      f"(wall {wall:.1f} s incl. I/O)",
            "per_core": piv / sum(o.solve_seconds for o in outs)}


def parametric_leg(local, no_cpu):
    """Parametric problems (BASELINE names none; reported beside the headline): the screened set of
    tests/manual/forest_good.json -- 1,327 random problems of 16 unknowns, 3 parameters, 20 inequalities,
    3 context rows -- through pipamd_solve_tableaux_lockstep (the device-resident traiter() of
    csrc/pip_quast.hip serves them; the time is the C call's), once as it is and once eight-fold for a
    throughput figure, beside the reference on one host core."""
    import pipbatch as pb
    from piplib_amd import engine as eng
    from piplib_amd import synth
    cfg = json.load(open(os.path.join(ROOT, "tests", "manual", "forest_good.json")))
    allp = synth.random_problems(cfg["seed"], cfg["count"], *cfg["shape"], 1, cmax=cfg["cmax"], bmax=cfg["bmax"])
    probs = [allp[i] for i in cfg["good"]]
    e = eng.Engine(local)
    out = {"workload": "%d random parametric problems, (unknowns, parameters, inequalities, context rows) = %s, integer solve"
                       % (len(probs), tuple(cfg["shape"]))}
    for label, ps in (("set", probs), ("set_x8", probs * 8)):
        best, piv = None, 0
        for _ in range(4):
            prep = eng.PreparedProblems(ps)
            t0 = time.perf_counter()
            eng.solve_prepared(e, prep, lockstep=True)
            dt = time.perf_counter() - t0
            res = prep.results()
            piv = sum(r[3] for r in res)
            best = dt if best is None else min(best, dt)
        served, back = e.last_device_tree()
        out[label] = {"problems": len(ps), "ms": best * 1e3, "problems_per_sec": len(ps) / best, "pivots_per_sec": piv / best,
                      "served_on_device": served, "handed_back": back, "failed": sum(r[1] != 0 for r in res)}
    if not no_cpu:
        fast = pb.REFPIP + "_fast"
        exe = fast if os.access(fast, os.X_OK) else (pb.REFPIP if pb.have_ref() else (pb.ORACLEPIP if pb.have_oracle() else None))
        if exe:
            pb.run_batch(exe, probs, pb.F_NOTEXT)
            o = pb.run_batch(exe, probs, pb.F_NOTEXT)
            out["cpu_one_core"] = {"problems_per_sec": len(probs) / o.solve_seconds, "ms": o.solve_seconds * 1e3,
                                   "kind": "reference" if exe != pb.ORACLEPIP else "port"}
            out["set"]["vs_one_cpu_core"] = out["set"]["problems_per_sec"] / out["cpu_one_core"]["problems_per_sec"]
            out["set_x8"]["vs_one_cpu_core"] = out["set_x8"]["problems_per_sec"] / out["cpu_one_core"]["problems_per_sec"]
    return out


_STREAMS = {}  # device -> the lanes' HIP streams, kept for the whole process


def lane_stream(torch, dev, i):
    """torch.cuda.Stream() deals out a pool of 32 streams per device in turn, and the HIP runtime binds a
    stream to one of its GPU_MAX_HW_QUEUES hardware queues when the stream is first used (tools/dbg_queues.py:
    streams 0..14 get a queue each, 15.. share with 14, 13, ...).  A second set of 12 lanes on the next 12
    pool streams therefore shared queues (configs[4] measured after the headline: 29 M -> 22 M pivots/s).
    Lane i keeps its stream for the process, and the streams are first used in order before any lane runs."""
    pool = _STREAMS.setdefault(str(dev), [])
    if len(pool) <= i:
        fresh = [torch.cuda.Stream(dev) for _ in range(max(i + 1, 24) - len(pool))]
        for st in fresh:
            with torch.cuda.stream(st):
                torch.zeros(1, device=dev)
        torch.cuda.synchronize(dev)
        pool.extend(fresh)
    return pool[i]


class Lanes:
    """`depth` batches in flight: `depth` resident batches (own seed each) and as many lanes, a lane being an engine,
    a workspace and a HIP stream.  Step k solves batch k mod depth on whichever lane is free (a complete
    pipamd_batch_load + pipamd_batch_solve + pipamd_batch_results of that batch): while one batch's last stragglers finish (a latency-bound
    tail that leaves most CUs idle) the other batches' bulk launches run.  ONE host thread drives all lanes through
    pipamd_batch_solve_async / pipamd_batch_poll -- it starts a batch on every lane and goes round polling; a lane
    that is done gets the next batch (threads=True: a host thread per lane calling the synchronous
    pipamd_batch_solve, round 2's driver, the threads drawing steps from a shared counter).
    fuse = G > 1: a batch of the lanes is G batches of the workload (this rank's shards of G batches in flight, or G
    small batches), loaded part by part with pipamd_batch_load_part into one workspace: one launch sequence per G
    steps; such a pass counts as G steps."""

    def __init__(self, cfg, depth, dev, local, gids, args, shard=None, threads=False, fuse=1, passes=None):
        """gids: per lane the `fuse` batch indices of the workload its workspace holds (batch_rows(cfg, g));
        shard = (lo, hi): this rank's slice of every batch (strong scaling)"""
        import numpy as np
        import torch
        from piplib_amd import engine as eng
        self.torch, self.eng, self.cfg, self.dev, self.depth = torch, eng, cfg, dev, depth
        self.threads, self.fuse = threads, fuse
        self.lanes, self.batches = [], []
        self.screened = {}      # batch index -> indices of the tableaux replaced (slow-converging cuts)
        self.screen_source = {}  # batch index -> who said so
        self.want = []          # per lane batch: what the oracle says about it (None: no record)
        self.next_step = 0  # steps handed out so far (the batches take turns across timed regions)
        self.last_of_lane = {}  # lane -> the batch its arrays hold (set by the runners)
        self.checked_passes = 0
        # Few launch sequences in flight, or a run so short that every sequence has a lane to itself: nothing runs beside a
        # sequence's middle launch for long, so the engines run in their lone-batches mode (pipamd_engine_set_lone_batches).
        # One MI355X, 96 steps of 10k tableaux: 2 lanes 234 -> 281 M pivots/s, 4 lanes 404 -> 393 M, 6 lanes 496 -> 432 M;
        # 20 steps of 1,250-tableau shards, five per sequence on 4 lanes: 265 -> 330 M.
        self.lone = depth <= 3 or (passes is not None and passes <= depth <= 5) or getattr(args, "lone", -1) == 1
        # the input rows stay resident and untouched in HBM for the whole run: T_ROWS_STAY lets the first pivot
        # launch read them where they are instead of a copy pass (--copy-rows switches that off)
        stay = 0 if getattr(args, "copy_rows", False) else eng.T_ROWS_STAY
        tf = (eng.T_INT if cfg["integer"] else 0) | stay
        screen_engine = eng.Engine(local)
        records = screen_records(cfg["screen"]) if cfg.get("screen") else {}
        full_batch = cfg["batch"]
        lo, hi = shard if shard else (0, full_batch)
        cache = {}

        def resident(g):
            """This rank's part of batch g in HBM.  Integer workloads: without the tableaux on which Gomory's cuts do not
            converge (about 3 in 100,000: integrer() asks for more than SCREEN_CUTS constant cuts; the reference itself
            does not finish them within minutes), replaced by their neighbours.  Which ones is the CPU oracle's verdict
            (tests/golden/bench_screen.json); the engine solves the unscreened part under the same row budget and
            must leave exactly those at PIPAMD_ST_CAPACITY, else the line is void.  A batch the file has no record of
            (another --batch) is screened by the engine alone and the line says so."""
            if g in cache:
                return cache[g]
            rows = batch_rows(cfg, g)
            assert rows.shape[0] == full_batch
            rec = records.get(str(g)) if rows.shape[0] == full_batch else None
            want = None
            if cfg["integer"] and rows.shape[0] > 1:
                mine = torch.as_tensor(rows[lo:hi], dtype=torch.int64).to(dev).contiguous()
                ncuts = cfg.get("screen_cuts", SCREEN_CUTS)
                bad = [b_ + lo for b_ in eng.slow_converging(screen_engine, mine, cfg["nvar"], cut_rows=ncuts,
                                                            entier_bits=cfg["ebits"])]
                del mine
                if rec is not None:
                    slow = sorted(rec["slow"])
                    if bad != [b_ for b_ in slow if lo <= b_ < hi]:
                        raise VoidLine(f"{cfg['key']} batch {g}: the engine leaves tableaux {bad} at PIPAMD_ST_CAPACITY under a budget "
                                       f"of {ncuts} cuts, the oracle's list (tests/golden/bench_screen.json) says {slow}")
                    src = "tests/golden/bench_screen.json (CPU oracle)"
                    want = rec
                    rows = rows.copy()
                    for b_ in slow:
                        rows[b_] = rows[replacement(b_, set(slow), full_batch)]
                    rows = rows[lo:hi]
                else:
                    src = "the engine itself (no oracle record for this batch)"
                    slow = bad
                    rows = rows[lo:hi].copy()
                    for b_ in slow:
                        rows[b_ - lo] = rows[replacement(b_ - lo, {x - lo for x in slow}, hi - lo)]
                if slow:
                    self.screened[g] = slow
                self.screen_source[g] = src
            else:
                rows = rows[lo:hi]
            cache[g] = (torch.as_tensor(np.ascontiguousarray(rows), dtype=torch.int64).to(dev).contiguous(), want)
            return cache[g]

        for i in range(depth):
            # G row arrays (the shards of G different batches, or G small batches) per batch of the lanes
            got = [resident(g) for g in gids[i]]
            self.batches.append([t for t, _ in got])
            self.want.append([w for _, w in got])
        self.gids = gids
        shape = (sum(p_.shape[0] for p_ in self.batches[0]),) + tuple(self.batches[0][0].shape[1:])
        for i in range(depth):
            e = eng.Engine(local)
            if args.waves:
                e.set_waves_per_job(args.waves)
            if args.round:
                e.set_round_pivots(args.round)
            if args.round_rows:
                e.set_round_rows(args.round_rows)
            # a lone batch is latency-bound in its tail: eight waves per tableau there (+8 % for one batch at a time, +3 %
            # with two in flight, -8 % with five, -3 % with 12, where the waves of a tail crowd out other batches' bulk)
            tw = getattr(args, "tail_waves", 0) or (8 if depth <= 3 and cfg["ebits"] == 64 and cfg["nvar"] + 1 <= 128
                                                    and shape[0] >= 2048 else 0)
            if tw:
                e.set_tail_waves(tw)
            if self.lone and getattr(args, "lone", -1) != 0:
                e.set_lone_batches(True)
            # safety net: a tableau that escaped the screening ends PIPAMD_ST_CAPACITY (and voids the line) instead of
            # growing for minutes
            e.set_max_rows(cfg.get("max_rows") or cfg["ni"] + 1024)
            e.set_timing(False)  # no HIP events in the timed region (kernel_ms_of switches them on)
            bw = getattr(args, "blocking_wait", -1)
            if bw > 0 or (bw < 0 and threads and depth > host_cpus()):  # more polling threads than CPUs only take turns
                e.set_blocking_wait(True)
            bulk_min = args.bulk_min if args.bulk_min > 0 else (256 if depth > 1 else 0)  # depth = this Lanes' lane count
            if bulk_min:
                e.set_bulk_min(bulk_min)
            b = eng.Batch(e, None, cfg["nvar"], 0, tflags=tf, entier_bits=cfg["ebits"], shape=shape)
            self.lanes.append((e, b, lane_stream(torch, dev, i)))
        # what one solve of each batch does (pivots, cuts, rows rewritten, tableaux, finished tableaux) and how it ends:
        # every batch once through lane 0 -- the pre-pass.  Its per-tableau arrays are reduced on the device: `sums` =
        # (sum of the pivot counts, sum of the cut counts, histogram of the statuses) is what every later solve of the
        # batch must reproduce (check_lanes), and the pivots of the tableaux that finished must be the oracle's.
        self.per_batch, self.hist, self.sums, self.fin_piv, self.want_piv = [], {}, [], [], []
        self.oracle_checked = 0
        e0, b0, _ = self.lanes[0]
        for k, parts in enumerate(self.batches):
            b0.load_parts(parts)
            b0.solve()
            c = b0.counters()
            self.per_batch.append((c["pivots"], c["cuts"], c["rows_rewritten"], shape[0], c["finished"]))
            b0.fetch()
            self.sums.append(self._reduce(b0))
            h = self.sums[-1][2]
            for st_, v in enumerate(h):
                if v:
                    self.hist[st_] = self.hist.get(st_, 0) + v
            if self.sums[-1][0] != c["pivots"]:
                raise VoidLine(f"{cfg['key']}: pipamd_batch_counters and the fetched pivot array disagree on batch {gids[k]}")
            # the oracle's word on the batch that is timed: which tableaux do not finish, and the pivots of the others
            done = (b0.status == eng.ST_SOLUTION) | (b0.status == eng.ST_NIL)
            fin_piv = int((b0.pivots.to(torch.int64) * done).sum().item())
            unfinished = torch.nonzero(~done).flatten().cpu().tolist()
            self.fin_piv.append(fin_piv)
            if all(w is not None for w in self.want[k]):
                off, want_unf = 0, []
                for w, part in zip(self.want[k], parts):
                    want_unf += [off + x - lo for x in w.get("unfinished", []) if lo <= x < hi]
                    off += part.shape[0]
                want_piv = sum(w["pivots_screened"] for w in self.want[k])
                self.want_piv.append(want_piv)
                # (a rank that holds a shard of every batch can only check its share of the list; the pivots of the shards
                # are summed over the ranks by the caller: oracle_pivot_check)
                if unfinished != want_unf or (shard is None and fin_piv != want_piv):
                    raise VoidLine(f"{cfg['key']} batches {gids[k]}: the engine finishes all but {unfinished[:8]} with {fin_piv} pivots, "
                                   f"the oracle (tests/golden/bench_screen.json) all but {want_unf[:8]} with {want_piv}")
                self.oracle_checked += 1
            else:
                self.want_piv.append(None)
        self.workers = None

    def _reduce(self, b):
        """(sum of pivots, sum of cuts, status histogram) of the arrays a lane's pipamd_batch_results filled, reduced on
        the device"""
        torch = self.torch
        h = torch.bincount(b.status.to(torch.int64), minlength=10)
        v = torch.cat([b.pivots.to(torch.int64).sum().reshape(1), b.cuts.to(torch.int64).sum().reshape(1), h]).cpu().tolist()
        return v[0], v[1], v[2:]

    def check_lanes(self):
        """Outside the timed bracket: what every lane's last pass left in its status / pivot / cut arrays against the
        pre-pass of the same batch.  Returns the number of passes checked; raises VoidLine on a difference."""
        self.torch.cuda.synchronize(self.dev)
        n = 0
        for i, k in sorted(self.last_of_lane.items()):
            got = self._reduce(self.lanes[i][1])
            if got != self.sums[k]:
                raise VoidLine(f"{self.cfg['key']}: lane {i} solved batch {self.gids[k]} in the timed region with (pivots, cuts, statuses) = "
                               f"{got}, the pre-pass with {self.sums[k]}")
            n += 1
        self.checked_passes += n
        return n

    def _take(self, npass):
        """the batches of the next `npass` steps"""
        order = [(self.next_step + k) % self.depth for k in range(npass)]
        self.next_step += npass
        return order

    def _run_async(self, order):
        """one host thread over all lanes; returns the batches solved, in completion order"""
        torch, it, active, done = self.torch, iter(order), {}, []

        def start(i):
            k = next(it, None)
            if k is None:
                return
            _, bi, st = self.lanes[i]
            bi.load_parts(self.batches[k], st.cuda_stream)
            bi.solve_async(st.cuda_stream)
            active[i] = k

        with torch.cuda.device(self.dev):
            for i in range(len(self.lanes)):
                start(i)
            while active:
                for i in list(active):
                    if self.lanes[i][1].poll():
                        # solution(): status, pivot and cut counts and the solutions into the caller's arrays
                        self.lanes[i][1].fetch(self.lanes[i][2].cuda_stream)
                        self.last_of_lane[i] = active[i]
                        done.append(active.pop(i))
                        start(i)
        return done

    def _run_threads(self, order):
        """a host thread per lane, each calling the synchronous pipamd_batch_solve; steps drawn from a shared list"""
        lock, it, done, failed = threading.Lock(), iter(order), [], []

        def worker(i):
            _, bi, st = self.lanes[i]
            try:
                with self.torch.cuda.device(self.dev):
                    while True:
                        with lock:
                            k = next(it, None)
                        if k is None:
                            break
                        bi.load_parts(self.batches[k], st.cuda_stream)
                        bi.solve(st.cuda_stream)
                        bi.fetch(st.cuda_stream)
                        with lock:
                            self.last_of_lane[i] = k
                            done.append(k)
            except BaseException as ex:  # surfaced below
                failed.append(ex)
        ts = [threading.Thread(target=worker, args=(i,), daemon=True) for i in range(len(self.lanes))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if failed:
            raise failed[0]
        return done

    def run(self, nsteps):
        """`nsteps` steps (a pass over a batch of `fuse` fused parts counts as `fuse` steps; nsteps is rounded up to
        whole passes); returns the batches solved"""
        order = self._take((nsteps + self.fuse - 1) // self.fuse)
        return self._run_threads(order) if self.threads else self._run_async(order)

    def close(self):
        pass

    def status_histogram(self):
        """PIPAMD_ST_* -> tableaux, over one solve of every batch"""
        return dict(self.hist)

    def totals(self, solved):
        """pivots, cuts, rows rewritten, tableaux, finished tableaux of the batches in `solved`"""
        tot = [0, 0, 0, 0, 0]
        for k in solved:
            for j in range(5):
                tot[j] += self.per_batch[k][j]
        return tot


def progress(msg):
    """a line on stderr per leg: a long run shows where it is (and a hung one, where it stopped)"""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_cpus():
    """CPUs this process may use: the affinity mask, capped by the container's cgroup quota"""
    n = max(1, len(os.sched_getaffinity(0)))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return n


def timed(lanes, steps, warmup, barrier, stagger_arg=0):
    lanes.run(max(warmup, lanes.depth))
    barrier()
    t0 = time.perf_counter()
    solved = lanes.run(steps)
    barrier()
    dt = time.perf_counter() - t0
    lanes.check_lanes()  # outside the bracket: the arrays the lanes fetched against the pre-pass
    return dt, solved


def timed_regions(lanes, steps, warmup, barrier, stagger_arg, n=3):
    """`n` timed regions of `steps` steps each behind one warm-up; sorted by duration: [(seconds, share)]"""
    lanes.run(max(warmup, lanes.depth))
    out = []
    for _ in range(n):
        lanes.next_step = 0   # every region times the same steps: batch 0, 1, ... in turn
        barrier()
        t0 = time.perf_counter()
        share = lanes.run(steps)
        barrier()
        out.append((time.perf_counter() - t0, share))
        lanes.check_lanes()  # outside the bracket: the arrays the lanes fetched against the pre-pass
    return sorted(out, key=lambda r: r[0])


def roofline_of(b, e, k_ms, cfg, extra=None, split=None):
    """Algorithmic HBM bytes of the pivots of THIS algorithm: read the pivot row, write the row
    that replaces the entering unit row, read+write every row that actually changes (counted by the kernel) -- rows
    with a zero multiplier keep their bits -- at the reference's 8 bytes per entry, over the launch durations (HIP
    events on the kernel's stream, un-pipelined).  With `split` (launch_split) the object is about the dominant
    kernel -- the launch that does most of the step's pivots -- and `step` holds the same figures for all pivot
    launches of the step together; without it the object is about the whole step."""
    c = b.counters()
    eb = 16.0 if cfg["ebits"] == 128 else 8.0
    ncol = cfg["nvar"] + 1
    algo = eb * ncol * (2.0 * c["rows_rewritten"] + 2.0 * c["pivots"])
    ach = algo / (k_ms * 1e-3) / 1e9
    step = {"achieved": ach, "frac": ach / HBM_PEAK_GBS, "kernel_ms": k_ms, "launches_per_step": e.last_solve_launches(),
            "avg_launch_ms": k_ms / max(1, e.last_solve_launches()), "algorithmic_bytes_per_step": algo,
            "rows_rewritten_per_pivot": c["rows_rewritten"] / max(1, c["pivots"])}
    dom = max(split, key=lambda l: l["pivots"]) if isinstance(split, list) and split else None
    if dom:
        r = {"bound": "hbm", "achieved": dom["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["frac"], "traffic": None,
             "kernel": dom["launch"], "kernel_ms": dom["ms"], "algorithmic_bytes_per_launch": dom["algorithmic_bytes"],
             "pivots_per_launch": dom["pivots"], "share_of_the_steps_pivots": dom["pivots"] / max(1, c["pivots"]),
             "rows_rewritten_per_pivot": dom["rows_rewritten_per_pivot"], "step": step}
    else:
        r = dict({"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None,
                  "kernel": "the pivot launches of a step (pip_lean_kernel where the shape has one, pip_advance_kernel)"}, **step)
    if extra:
        r.update(extra)
    if dom and r.get("traffic"):
        # the same launch on the bytes the counters saw (int rows: fewer than the 8-byte convention of `achieved`)
        r["achieved_counter_bytes"] = r["traffic"] / (dom["ms"] * 1e-3) / 1e9
        r["frac_counter_bytes"] = r["achieved_counter_bytes"] / HBM_PEAK_GBS
    return r


def launch_split(b, e, cfg, parts):
    """Roofline per launch of one un-pipelined solve.  The launches do very different work per pivot: the lean bulk
    launch (csrc/pip_lean.h, one wave per tableau, int rows: the headline shape only) takes every tableau as far as
    its entries stay ints and its pivot budget lasts, a second one-wave launch the ones it left, the four-wave tail launches the long
    tableaux (a late pivot of a long tableau rewrites 15-25 rows, an early one 1-3).  Durations: HIP events around each
    launch; pivots and rows per launch: the batch solved again with the solve stopped after the lean launch and after
    the bulk launches (pipamd_debug_single_launch), the tail's = the whole solve's minus those."""
    e.set_timing(True)
    b.load_parts(parts)
    b.solve()
    n = e.last_solve_launches()
    ms = [e.last_launch_ms(i) for i in range(n)]
    whole = b.counters()
    if n < 2:
        return None

    def stopped(level):
        e.debug_single_launch(level)
        try:
            b.load_parts(parts)
            b.solve()
            return b.counters(), e.last_solve_launches()
        finally:
            e.debug_single_launch(0)
    bulk, nb = stopped(1)     # after the one-wave launches (lean + general, or the general one alone)
    lean, nl = stopped(2)     # after the lean launch alone (the same as `bulk` when the batch has no lean launch)
    b.load_parts(parts)   # leave the batch solved
    b.solve()
    eb = 16.0 if cfg["ebits"] == 128 else 8.0
    ncol = cfg["nvar"] + 1
    legs = []
    if nb == 2:
        legs.append(("lean bulk (pip_lean_kernel, one wave per tableau, int rows)", ms[0], lean["pivots"], lean["rows_rewritten"],
                     lean["finished"]))
        legs.append(("second bulk launch over what the first left (pip_lean_kernel again where the shape has one: it resumes its "
                     "paused tableaux, 160 pivots more; else pip_advance_kernel, one wave per tableau)", ms[1],
                     bulk["pivots"] - lean["pivots"], bulk["rows_rewritten"] - lean["rows_rewritten"], bulk["finished"] - lean["finished"]))
    elif cfg["ebits"] == 64 and cfg["nvar"] <= 127 and cfg["ni"] + (48 if cfg["integer"] else 0) <= 160:
        # (an engine in lone-batches mode: the lean launch is the whole bulk stage)
        legs.append(("lean bulk (pip_lean_kernel, one wave per tableau, int rows)", ms[0], bulk["pivots"], bulk["rows_rewritten"],
                     bulk["finished"]))
    else:
        legs.append(("bulk (pip_advance_kernel, one wave per tableau)", ms[0], bulk["pivots"], bulk["rows_rewritten"], bulk["finished"]))
    legs.append(("tail (pip_advance_kernel, four waves per tableau -- sixteen for a short list of 128-bit tableaux --, %d launch%s)"
                 % (n - nb, "" if n - nb == 1 else "es"), sum(ms[nb:]),
                 whole["pivots"] - bulk["pivots"], whole["rows_rewritten"] - bulk["rows_rewritten"], whole["finished"] - bulk["finished"]))
    out = []
    for kind, t, piv, rows, fin in legs:
        by = eb * ncol * (2.0 * rows + 2.0 * piv)
        out.append({"launch": kind, "ms": t, "pivots": piv, "tableaux_finished": fin, "rows_rewritten_per_pivot": rows / max(1, piv),
                    "algorithmic_bytes": by, "achieved": by / (t * 1e-3) / 1e9, "frac": by / (t * 1e-3) / 1e9 / HBM_PEAK_GBS})
    return out


def profile_json(*names):
    """the first of profiles/<name> that exists, parsed; (None, None) otherwise"""
    for nm in names:
        path = os.path.join(ROOT, "profiles", nm)
        if os.path.exists(path):
            try:
                return json.load(open(path)), "profiles/" + nm
            except Exception:
                pass
    return None, None


def pick_fuse(shard, steps, target=10000):
    """How many batches (or shards) of `shard` tableaux share one workspace and launch sequence: about `target`
    tableaux per sequence (one MI355X, 1,250-tableau shards, 192 steps: 4 per sequence 454 M pivots/s, 8: 502 M,
    20: 501 M), as a divisor of the step count where one is near -- the nearest one below, else the nearest above --
    so that a short run is whole passes with a lane each (the driver's 20 steps at 1,250, lanes in lone-batches mode:
    4 per sequence 268 M, 5: 330 M, 10: 315 M; 8 -- three passes, 24 steps -- 215 M)."""
    f0 = max(1, min(16, target // max(1, shard)))
    if f0 == 1 or steps % f0 == 0:
        return f0
    below = [d for d in range(max(1, (f0 + 1) // 2), f0) if steps % d == 0]
    above = [d for d in range(f0 + 1, min(16, 2 * f0) + 1) if steps % d == 0]
    return max(below) if below else (min(above) if above else f0)


def kernel_ms_of(b, parts=None, reps=2):
    b.e.set_timing(True)
    ms = []
    for _ in range(reps):
        if parts is None:
            b.load()
        else:
            b.load_parts(parts)
        b.solve()
        ms.append(b.last_solve_ms())
    return sum(ms) / len(ms)


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_command(ngpus, argv, port):
    """the launcher `bench.py --gpus N` starts when nobody launched it: one rank per GPU of this node"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def spawn_ranks(ngpus, argv):
    """`python bench.py --gpus N` without a launcher: N ranks as a CHILD process (this process has not touched the GPU and
    never does; no exec), rank 0's JSON line passed through, the child's exit code returned"""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(spawn_command(ngpus, argv, free_port()), env=env)
    try:
        return proc.wait()
    except KeyboardInterrupt:
        proc.terminate()
        return proc.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=96)
    ap.add_argument("--warmup", type=int, default=24)
    ap.add_argument("--batch", type=int, default=10000, help="tableaux per GPU")
    ap.add_argument("--waves", type=int, default=0, help="waves per tableau (0 = engine default)")
    ap.add_argument("--round", type=int, default=0, help="pivot budget per tableau in the bulk launch (0 = engine default)")
    ap.add_argument("--round-rows", type=int, default=0, help="spare rows in the bulk launch's LDS image (0 = engine default)")
    ap.add_argument("--tail-waves", type=int, default=0, help="waves per tableau in the tail launch (0 = engine default)")
    ap.add_argument("--bulk-min", type=int, default=-1,
                    help="smallest batch that starts with the one-wave bulk launch (-1 = 256 when several batches are "
                         "in flight, the engine default of 2048 otherwise)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="strong (default for --gpus > 1, BASELINE configs[3]): every --batch-tableau batch is sharded "
                         "over the ranks; weak (default for one GPU): --batch tableaux per GPU and batch.  The other "
                         "mode is measured too and reported as `other_scaling`.")
    ap.add_argument("--pipeline", type=int, default=16, help="batches in flight (streams)")
    ap.add_argument("--threads", action="store_true",
                    help="one host thread per lane calling the synchronous pipamd_batch_solve (round 2's driver) instead of "
                         "one thread over pipamd_batch_solve_async / pipamd_batch_wait")
    ap.add_argument("--fuse", type=int, default=0,
                    help="strong scaling: shards of this many batches share a workspace and a launch sequence (0 = enough "
                         "for about 10,000 tableaux per launch sequence, pick_fuse)")
    ap.add_argument("--stagger", type=float, default=0.0, help="(ignored; kept for old command lines)")
    ap.add_argument("--blocking-wait", type=int, default=-1,
                    help="1: host threads sleep while the device works, 0: they poll; -1: sleep when there are more lanes than CPUs")
    ap.add_argument("--lone", type=int, default=-1,
                    help="0: a lone batch (--pipeline 1) keeps the launch sequence of the pipelined run (profiles); "
                         "-1: it runs with pipamd_engine_set_lone_batches; 1: every lane does, whatever their number (experiments)")
    ap.add_argument("--copy-rows", action="store_true", help="load copies the input rows into the job blocks (no PIPAMD_T_ROWS_STAY)")
    ap.add_argument("--no-dense", action="store_true", help="skip the row-skipping-off measurement")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-others", action="store_true", help="skip other_configs and pipeline1_value")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} under a launcher with WORLD_SIZE={os.environ.get('WORLD_SIZE')}: "
                         "the line would be labelled with the wrong GPU count")

    # Batches in flight run on separate HIP streams; the runtime multiplexes streams onto
    # GPU_MAX_HW_QUEUES hardware queues (default 4) and kernels that share a queue serialise, so
    # give every lane a queue of its own (a HIP runtime setting, read when the runtime starts).
    # 16 queues suit 14 lanes best (round 3, one MI355X, 20 / 96 steps: 14 lanes on 16 queues 383 / 394 M pivots/s,
    # 16 lanes 378 / 394, 12 lanes 364 / 384; 24 queues 355 / 373, 32 queues 311 / 373).
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    import numpy as np
    import torch
    from piplib_amd import engine as eng
    from piplib_amd import synth

    from piplib_amd import dist as pdist
    rank, world, local = pdist.env_rank()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # PIPAMD_BENCH_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than
    # ranks (ranks then share GPUs); the driver's runs use nccl (== RCCL on ROCm), one GPU per rank
    backend = os.environ.get("PIPAMD_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    pdist.init(backend, dev)

    def barrier():
        torch.cuda.synchronize(dev)
        pdist.barrier()
        torch.cuda.synchronize(dev)

    if args.scaling is None:
        args.scaling = "strong" if world > 1 else "weak"

    def build_lanes(scaling, threads=False):
        """the lanes of one measurement: (cfg, lanes, depth)"""
        cfg = dict(MAIN)
        cfg["batch"] = args.batch
        if scaling == "strong":
            # BASELINE configs[3]: every 10k-tableau batch is sharded over the GPUs: a GPU holds 1/world of every
            # batch in flight.  The shards are independent tableaux, so a rank loads its shards of `fuse` batches
            # into ONE workspace (pipamd_batch_load_part) and one launch sequence serves them: about 10,000
            # tableaux per launch sequence whatever the world size, instead of 24 launch sequences of 1,250 tableaux
            # (round 2: 262 M pivots/s per GPU at 1,250, 304 M at 2,500, 350 M at 5,000, 362 M at 10,000).
            shard = max(1, (args.batch + world - 1) // world)
            fuse = args.fuse if args.fuse > 0 else pick_fuse(shard, args.steps)
            depth = max(1, min(args.pipeline, (args.steps + fuse - 1) // fuse))
            span = pdist.shard_range(args.batch, rank, world)
            # lane i holds this rank's shards of batches i * fuse .. i * fuse + fuse - 1 (the same on every rank)
            gids = [[i * fuse + k for k in range(fuse)] for i in range(depth)]
        else:
            # small batches (--batch below 2,500 per GPU) are fused the same way: a lane's workspace holds `fuse` of them
            fuse = args.fuse if args.fuse > 0 else pick_fuse(args.batch, args.steps)
            depth = max(1, min(args.pipeline, (args.steps + fuse - 1) // fuse))
            span = None
            # every rank has batches of its own: rank r takes batches r * depth * fuse ...
            gids = [[(rank * depth + i) * fuse + k for k in range(fuse)] for i in range(depth)]
        return cfg, Lanes(cfg, depth, dev, local, gids, args, shard=span, threads=threads, fuse=fuse,
                          passes=(args.steps + fuse - 1) // fuse), depth

    def oracle_pivot_check(ln):
        """strong scaling: the pivots of the shards of a batch, summed over the ranks, against the oracle's count for the
        batch; returns the number of lane batches checked (every rank calls this)"""
        if not ln.want_piv or any(w is None for w in ln.want_piv):
            return 0
        got, _ = pdist.gather_totals(ln.fin_piv, 0.0, dev)
        if rank == 0 and [int(x) for x in got] != ln.want_piv:
            raise VoidLine(f"pivots of the sharded batches over all ranks {[int(x) for x in got]}, the oracle's {ln.want_piv}")
        return len(got)

    progress("building the lanes")
    try:
        cfg, lanes, depth = build_lanes(args.scaling, args.threads)
        oracle_batches = lanes.oracle_checked if args.scaling == "weak" else oracle_pivot_check(lanes)
    except VoidLine as ex:
        print(json.dumps({"error": "void: " + str(ex), "metric": "pivots/sec", "value": None}), flush=True)
        raise SystemExit(3)
    progress(f"{depth} lanes ready; timing")
    my_batch = cfg["batch"]
    fuse = lanes.fuse
    e, b, _ = lanes.lanes[0]

    # three timed regions of --steps steps each; the median is the line's value, all three are printed
    try:
        regions = timed_regions(lanes, args.steps, args.warmup, barrier, args.stagger)
    except VoidLine as ex:
        print(json.dumps({"error": "void: " + str(ex), "metric": "pivots/sec", "value": None}), flush=True)
        raise SystemExit(3)
    regions_checked_passes = lanes.checked_passes
    dt, share = regions[1]
    steps_done = len(share) * fuse   # == --steps unless shards are fused (whole passes of `fuse` steps)
    progress(f"regions {[round(r[0] * 1e3, 2) for r in regions]} ms")
    tot = lanes.totals(share)
    hist = lanes.status_histogram()
    screened0 = dict(lanes.screened)
    screen_src0 = sorted(set(lanes.screen_source.values()))
    gids0 = lanes.gids
    # the same lanes over a longer run: the driver's 20 steps are one round of 16 batches that start together plus four
    # stragglers (pipeline fill and drain); 96 steps show the rate of a steady stream of batches
    steady = None
    if args.steps < 96:
        try:
            sdt, sshare = timed(lanes, 96, 0, barrier)
            sgt, sdt_max = pdist.gather_totals(lanes.totals(sshare), sdt, dev)
            steady = {"value": sgt[0] / sdt_max, "steps": len(sshare) * fuse, "ms_per_step": sdt_max / (len(sshare) * fuse) * 1e3}
        except VoidLine as ex:
            steady = {"error": str(ex)}

    # the advance kernel's own launch durations (HIP events on its stream), un-overlapped
    parts0 = lanes.batches[0]
    k_ms = kernel_ms_of(b, parts0)
    torch.cuda.synchronize(dev)
    gt, dt_max = pdist.gather_totals(tot, dt, dev)

    other = None
    if world > 1:  # the other scaling mode, same steps (every rank takes part)
        mode2 = "weak" if args.scaling == "strong" else "strong"
        lanes.close()
        del lanes
        torch.cuda.empty_cache()
        try:
            cfg2, lanes2, depth2 = build_lanes(mode2, args.threads)
            if mode2 == "strong":
                oracle_pivot_check(lanes2)
            dt2, share2 = timed(lanes2, args.steps, args.warmup, barrier, args.stagger)
        except VoidLine as ex:
            print(json.dumps({"error": "void (other_scaling): " + str(ex), "metric": "pivots/sec", "value": None}), flush=True)
            raise SystemExit(3)
        gt2, dt2_max = pdist.gather_totals(lanes2.totals(share2), dt2, dev)
        other = {"scaling": mode2, "value": gt2[0] / dt2_max, "unit": "pivots/s",
                 "ms_per_step": dt2_max / (len(share2) * lanes2.fuse) * 1e3,
                 "batch_per_gpu": cfg2["batch"], "pipeline_depth": depth2, "fused_batches_per_launch_sequence": lanes2.fuse,
                 "problems_per_sec": gt2[3] / dt2_max}
        lanes2.close()
        del lanes2
        torch.cuda.empty_cache()
        lanes = None

    if rank != 0:
        # ranks other than 0 only take part in the headline measurement
        pdist.finish()
        return

    ms_step = dt_max / steps_done * 1e3
    out = {
        "metric": "pivots/sec (batched 64x128 int64 tableaux, integer solve with Gomory cuts)",
        "value": gt[0] / dt_max,
        "unit": "pivots/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_step,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "int64",
        "data": "synthetic",
        "config": {"workload": MAIN["workload"] + (" (BASELINE configs[2])" if args.scaling == "weak" else
                                                     " sharded over the ranks (BASELINE configs[3])"),
                   "batch_per_gpu": my_batch, "nvar": cfg["nvar"], "nparm": 0, "ni": cfg["ni"],
                   "parallelism": f"{world} x independent batches (one workgroup per tableau)",
                   "pipeline_depth": depth,
                   "batches": "batch g = synth.lexmin_batch(1000 + 7919 * g, ...); weak scaling: lane i of rank r holds batch "
                              "r * lanes + i; strong: lane i holds every rank's shard of batches i * fuse .. i * fuse + fuse - 1",
                   "rank0_batches": gids0,
                   "host_threads": depth if args.threads else 1,
                   "driver": "one host thread per lane, pipamd_batch_solve" if args.threads else
                             "one host thread, pipamd_batch_solve_async / pipamd_batch_wait",
                   "fused_batches_per_launch_sequence": fuse,
                   "screened_count": sum(len(v) for v in screened0.values()),
                   "screened_by": "; ".join(screen_src0),
                   "screened_out": {"what": "tableaux on which integrer() asks for more than 448 constant cuts (Gomory's cuts do "
                                            "not converge: the reference does not finish them within minutes either), replaced "
                                            "by their neighbours before anything is timed; batch -> indices (rank 0's lanes)",
                                    "rank0": {str(k): v for k, v in sorted(screened0.items())}}},
        "rccl_ranks": pdist.world_size(), "backend": backend if world > 1 else None,
        "regions_checked": regions_checked_passes > 0, "regions_checked_passes": regions_checked_passes,
        "regions_checked_how": "after every timed region the status / pivot / cut arrays each lane's last pipamd_batch_results "
                               "filled are reduced on the device (sums, status histogram) and compared with the pre-pass of "
                               "the same batch; a difference voids the line",
        "oracle_checked_batches": oracle_batches,
        "oracle_checked_how": "pre-pass: the tableaux the engine leaves at PIPAMD_ST_CAPACITY under a 448-cut budget must be the CPU "
                              "oracle's list and the pivots of each batch the oracle's count (tests/golden/bench_screen.json)",
        "steps_timed": steps_done,
        "regions_ms": [round(r[0] * 1e3, 3) for r in regions],
        "problems_per_sec": gt[3] / dt_max,
        "pivots_per_step": gt[0] / steps_done,
        "cuts_per_step": gt[1] / steps_done,
        "finished_fraction": gt[4] / max(1.0, gt[3]),
        "status_histogram_rank0": {STATUS_NAMES.get(k, str(k)): v for k, v in sorted(hist.items())},
        "rows_rewritten_per_pivot": gt[2] / max(1.0, gt[0]),
    }
    if out["finished_fraction"] < 1.0:
        # a tableau without a final status of the reference's (solution / nil) voids the line
        out["error"] = "finished_fraction < 1: the workload was not solved completely"
        print(json.dumps(out), flush=True)
        pdist.finish()
        raise SystemExit(3)
    if other:
        out["other_scaling"] = other
    if steady:
        out["steady_state_value"] = steady.get("value")
        out["steady_state"] = steady
    traffic, traffic_src = None, None
    t, src = profile_json("r04_pmc_hbm.json", "r03_pmc_hbm.json", "r02_pmc_hbm.json")
    step_traffic = None
    if t and t.get("batch_per_gpu") == my_batch:
        traffic, traffic_src = t["hbm_bytes_per_step"], src
        step_traffic = traffic
    c0 = b.counters()
    issue, issue_src = profile_json("r04_pmc_issue.json", "r03_pmc_issue.json")
    split = None
    if fuse == 1:
        try:
            split = launch_split(b, e, cfg, parts0)
        except Exception as ex:
            split = {"error": repr(ex)}
    # `roofline` is about the dominant launch: its own counter figure where the profile has one (the lean launch)
    if (isinstance(split, list) and split and "pip_lean_kernel" in max(split, key=lambda l: l["pivots"])["launch"]
            and t and t.get("batch_per_gpu") == my_batch and t.get("lean_launch")):
        traffic = t["lean_launch"]["hbm_bytes"]
    out["roofline"] = roofline_of(b, e, k_ms, cfg, split=split, extra={
        "traffic": traffic, "traffic_of_all_pivot_launches_of_a_step": step_traffic,
        "traffic_source": (traffic_src + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command with --pipeline 1; "
                           "read from the file, not measured in this run)") if traffic_src else None,
        "launches": split,
        "issue": dict(issue, source=issue_src + " (rocprofv3 --pmc SQ_INSTS_* / SQ_ACTIVE_INST_* passes; read from the file, "
                                                "not measured in this run)") if issue else None,
        "measured": "HIP events around each launch, 2 un-pipelined steps after the timed region "
                    "(= `bench.py --pipeline 1`, the command of profiles/r02_kernel_stats.csv)",
        "timed_region_GBps_per_gpu": 8.0 * (cfg["nvar"] + 1) * (2.0 * gt[2] + 2.0 * gt[0]) / world / dt_max / 1e9,
        # the regime that is timed (16 batches in flight), per GPU: algorithmic bytes at the reference's 8 bytes per entry,
        # and the HBM bytes the counters saw for one step (the file's figure: int rows) over the measured time per step
        "timed_region": {
            "hbm_GBps_algorithmic": 8.0 * (cfg["nvar"] + 1) * (2.0 * gt[2] + 2.0 * gt[0]) / world / dt_max / 1e9,
            "frac_algorithmic": 8.0 * (cfg["nvar"] + 1) * (2.0 * gt[2] + 2.0 * gt[0]) / world / dt_max / 1e9 / HBM_PEAK_GBS,
            "hbm_GBps_counter_bytes": (step_traffic / (ms_step * 1e-3) / 1e9) if step_traffic else None,
            "frac": (step_traffic / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS) if step_traffic else None,
            "counter_bytes_per_step": step_traffic, "counter_bytes_source": traffic_src},
        "dense_equivalent_GBps": b.pivot_bytes() * c0["pivots"] / (k_ms * 1e-3) / 1e9,
        "note": "sparse workload: ~2.7 of ~80 rows change per pivot, so the pivot loop is latency/issue-bound, "
                "not HBM-bound; see roofline_dense_mode for the HBM-bound regime of the same kernel"})

    # Same workload once more with row skipping off: every real row is read and written on
    # every pivot, which is the reference's access pattern (traiter.c:467-502) and the regime
    # in which the row-update path is HBM-bound.  Reported beside the main number.
    progress("headline done: %.1f M pivots/s" % (out["value"] / 1e6))
    if not args.no_dense:
        try:  # an extra leg never costs the headline line
            ed = eng.Engine(local)
            ed.set_waves_per_job(4)  # streaming regime: four waves share a tableau's rows
            bd = eng.Batch(ed, parts0[0], cfg["nvar"], 0, tflags=eng.T_INT | eng.T_NOSKIP)
            bd.load()
            bd.solve()
            dk = kernel_ms_of(bd)
            cd = bd.counters()
            # every real row (cut rows included, counted by the kernel) is read and written once per
            # pivot, plus the pivot-row read and the write of the row that replaces the unit row
            dbytes = 8.0 * (cfg["nvar"] + 1) * (2.0 * cd["rows_rewritten"] + 2.0 * cd["pivots"])
            out["roofline_dense_mode"] = {
                "bound": "hbm", "achieved": dbytes / (dk * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": dbytes / (dk * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel_ms": dk, "pivots": cd["pivots"],
                "rows_rewritten_per_pivot": cd["rows_rewritten"] / max(1, cd["pivots"]),
                "algorithmic_bytes_per_step": dbytes,
                "note": "same batch with row skipping disabled (PIPAMD_T_NOSKIP, 4 waves per tableau): every real row "
                        "is read and written on every pivot, the reference's access pattern"}
            del bd, ed
        except Exception as ex:
            out["roofline_dense_mode_error"] = repr(ex)

    if not args.no_others and world == 1:
        try:  # an extra leg never costs the headline line
            # one batch at a time (what a caller gets from a single pipamd_batch_load + pipamd_batch_solve)
            if lanes is not None:
                lanes.close()
            del lanes
            torch.cuda.empty_cache()
            progress("one batch at a time")
            one = Lanes(cfg, 1, dev, local, [[0]], args)  # a fresh engine with its defaults for a lone batch
            n1 = max(8, min(24, args.steps))
            dt1, sh1 = timed(one, n1, 2, barrier, 0)
            t1 = one.totals(sh1)
            out["pipeline1_value"] = t1[0] / dt1
            out["pipeline1_ms_per_step"] = dt1 / n1 * 1e3
            one.close()
            del one
            torch.cuda.empty_cache()
            others = []
            for oc in OTHERS:
                progress(oc["key"])
                # (128-bit tableaux of 256 columns: 12 lanes -- 40 M pivots/s against 26 M with 16, whose working sets crowd the L2)
                od = args.pipeline if oc["ebits"] == 64 else min(args.pipeline, 12)
                # batches of a thousand small tableaux are a tenth of a millisecond of GPU work each: ten of them share a
                # workspace and a launch sequence (pipamd_batch_load_part), as the shards of a strong-scaling run do
                # (5 per sequence 617 M pivots/s, 10: 651 M, 16: 662 M)
                ofuse = max(1, min(16, 10000 // oc["batch"])) if oc["ebits"] == 64 else 1
                # (configs[4]: every lane holds the one pinned batch)
                ogids = [[0] for _ in range(od)] if oc.get("family") else [[i * ofuse + k for k in range(ofuse)] for i in range(od)]
                try:
                    ol = Lanes(oc, od, dev, local, ogids, args, fuse=ofuse)
                except VoidLine as ex:
                    others.append({"config": oc["key"], "workload": oc["workload"], "error": "void: " + str(ex), "value": None})
                    continue
                # (the pinned configs[4] batch is seconds of GPU time a pass -- its longest tableau needs 4,513 pivots on
                # some 900 rows --: two passes per lane and region, and two lone steps)
                heavy = bool(oc.get("family"))
                osteps = (2 if heavy else 16) * od * ofuse
                # The median of three timed regions: a region here is 35 ms to 1.4 s long, and the first region of a
                # fresh set of lanes came out 2-3x slower than every later one on configs[1] (cause not found).
                regions = sorted((timed(ol, osteps, od, barrier) for _ in range(3)), key=lambda r: r[0])
                odt, osh = regions[1]
                osteps = len(osh) * ofuse
                ot = ol.totals(osh)
                oe, ob, _ = ol.lanes[0]
                okm = kernel_ms_of(ob, ol.batches[0])
                o1 = Lanes(oc, 1, dev, local, [[0]], args)
                n_lone = 2 if heavy else 16
                odt1, osh1 = timed(o1, n_lone, 1 if heavy else 2, barrier, 0)
                ot1 = o1.totals(osh1)
                others.append({
                    "config": oc["key"], "workload": oc["workload"], "dtype": "int128" if oc["ebits"] == 128 else "int64",
                    "value": ot[0] / odt, "unit": "pivots/s", "ms_per_step": odt / osteps * 1e3, "steps": osteps,
                    "pipeline_depth": od, "fused_batches_per_launch_sequence": ofuse, "host_threads": 1,
                    "regions_ms": [round(r[0] * 1e3, 3) for r in regions], "problems_per_sec": ot[3] / odt, "pivots_per_step": ot[0] / osteps,
                    "finished_fraction": ot[4] / max(1, ot[3]),
                    "status_histogram": {STATUS_NAMES.get(k, str(k)): v for k, v in sorted(ol.status_histogram().items())},
                    "regions_checked": ol.checked_passes > 0, "oracle_checked_batches": ol.oracle_checked,
                    "screened_count": sum(len(v) for v in ol.screened.values()),
                    "pipeline1_value": ot1[0] / odt1, "pipeline1_ms_per_step": odt1 / n_lone * 1e3,
                    "roofline": roofline_of(ob, oe, okm, oc)})
                ol.close()
                o1.close()
                del ol, o1
                torch.cuda.empty_cache()
            out["other_configs"] = others
            try:
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                progress("parametric leg")
                out["parametric"] = parametric_leg(local, args.no_cpu)
            except Exception as ex:  # the leg is an extra: never lose the headline line over it
                out["parametric"] = {"error": repr(ex)}
        except Exception as ex:
            out["other_configs_error"] = repr(ex)

    if not args.no_cpu and world == 1:  # the CPU baseline belongs to the one-GPU line
        progress("cpu baseline")
        rows_h = batch_rows(cfg, 0)  # (batch 0 has no slow-converging tableau)
        cb = cpu_baseline(rows_h, cfg["nvar"], cfg["ni"])
        if cb:
            out["cpu_baseline"] = cb
            out["speedup_vs_cpu_baseline"] = out["value"] / cb["value"]
    print(json.dumps(out), flush=True)
    pdist.finish()


if __name__ == "__main__":
    main()
