#!/usr/bin/env python3
"""bench.py -- pivots/s of the HIP PIP engine on BASELINE.json's workloads.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Headline workload (BASELINE.json configs[2], the one the metric is quoted on): per GPU a batch of
10,000 synthetic 64x128 int64 tableaux (nvar = 127 unknowns, 64 inequality rows, constant
column; no parameters), integer solve with Gomory cuts, one workgroup per tableau.
A "step" = tab_get-style load of the batch into the HBM row store + the whole traiter()
pivot loop for every tableau (inputs are resident in HBM before the timed region).
Steps are pipelined: up to --pipeline (default 12) batches are in flight on separate HIP streams,
each lane with its own batch (own seed), engine and workspace (each step is a complete load +
solve of its batch), all driven by ONE host thread through the asynchronous C ABI
(pipamd_batch_solve_async / pipamd_batch_wait; `--threads` = round 2's one-host-thread-per-lane
driver, measured beside it as `threaded_value`).  The timed region is run three times; ms_per_step is
the median region / steps and all three are printed (`regions_ms`).  `pipeline1_value` is the same
workload with one batch at a time.  `other_configs` carries BASELINE configs[1] and configs[4]
measured the same way (shorter runs).
Multi-GPU (--gpus N > 1): BASELINE configs[3] -- every 10k-tableau batch is sharded over the ranks
(strong scaling), no data-path collective; a rank fuses its shards of several batches in flight into
one workspace (pipamd_batch_load_part) so that one launch sequence serves ~5,000 tableaux whatever N;
RCCL only sums the totals here (the results gather is piplib_amd.dist.gather_results).  The
weak-scaling figure (10k tableaux per GPU and batch) is measured as well and reported as
`other_scaling`.

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
MAIN = dict(key="configs[2]", workload="10k-batch synthetic 64x128 tableaux, int64, integer solve with Gomory cuts",
            batch=10000, nvar=127, ni=64, integer=True, ebits=64, gen={})
OTHERS = [
    dict(key="configs[1]", workload="1k-batch synthetic 32x64 tableaux, int64, rational (non-integer) solve",
         batch=1000, nvar=63, ni=32, integer=False, ebits=64, gen={}),
    # coefficients up to 30 in up to 6 columns per row: the determinant limbs of the int64 build overflow
    # on these ("Integer overflow", traiter.c:424,442), the 128-bit Entier build solves them
    dict(key="configs[4]", workload="1k-batch synthetic 128x256 tableaux, 128-bit Entier, integer solve "
                                    "(inputs on which the int64 build stops with 'Integer overflow')",
         batch=1000, nvar=255, ni=128, integer=True, ebits=128, gen=dict(nnz=6, cmax=30),
         # Gomory cuts converge slowly on about 8 tableaux per 10,000 of this family (tens of thousands of pivots over
         # thousands of cut rows; the reference on the CPU takes seconds to minutes for each): the leg gives a tableau
         # ni + 192 rows (pipamd_engine_set_max_rows) and reports the few that want more as `capacity`
         max_rows=128 + 192),
]


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
    """`depth` batches in flight, each with its own engine, workspace, input rows (own seed) and HIP
    stream: while one batch's last stragglers finish (a latency-bound tail that leaves most CUs idle)
    the other batches' bulk launches run.  Every step is a full load + solve.  One host thread drives
    all lanes through pipamd_batch_solve_async / pipamd_batch_wait (threads=True: a host thread per
    lane calling the synchronous pipamd_batch_solve, round 2's driver).
    fuse = G > 1: a lane's workspace holds G batches' worth of tableaux (this rank's shards of G
    batches in flight, loaded part by part with pipamd_batch_load_part): one launch sequence per G
    steps; a lane pass counts as G steps."""

    def __init__(self, cfg, depth, dev, local, seeds, args, gen=None, threads=False, fuse=1):
        import torch
        from piplib_amd import engine as eng
        from piplib_amd import synth
        self.torch, self.eng, self.cfg, self.dev, self.depth = torch, eng, cfg, dev, depth
        self.threads, self.fuse = threads, fuse
        self.lanes = []
        gen = gen or (lambda seed: synth.lexmin_batch(seed, cfg["batch"], cfg["nvar"], cfg["ni"], **cfg["gen"]))
        for i in range(depth):
            e = eng.Engine(local)
            if args.waves:
                e.set_waves_per_job(args.waves)
            if args.round:
                e.set_round_pivots(args.round)
            if args.round_rows:
                e.set_round_rows(args.round_rows)
            # a lone batch is latency-bound in its tail: eight waves per tableau there (+8 % for one batch
            # at a time, -3 % with 12 in flight, where the waves of a tail crowd out other batches' bulk)
            tw = getattr(args, "tail_waves", 0) or (8 if depth == 1 and cfg["ebits"] == 64 and cfg["nvar"] + 1 <= 128
                                                    and cfg["batch"] >= 2048 else 0)
            if tw:
                e.set_tail_waves(tw)
            if cfg.get("max_rows"):
                e.set_max_rows(cfg["max_rows"])
            e.set_timing(False)  # no HIP events in the timed region (kernel_ms_of switches them on)
            bw = getattr(args, "blocking_wait", -1)
            if bw > 0 or (bw < 0 and threads and depth > host_cpus()):  # more polling threads than CPUs only take turns
                e.set_blocking_wait(True)
            bulk_min = args.bulk_min if args.bulk_min > 0 else (256 if depth > 1 else 0)  # depth = this Lanes' lane count
            if bulk_min:
                e.set_bulk_min(bulk_min)
            # the input rows stay resident and untouched in HBM for the whole run: T_ROWS_STAY lets the first pivot
            # launch read them where they are instead of a copy pass (--copy-rows switches that off)
            stay = 0 if getattr(args, "copy_rows", False) else eng.T_ROWS_STAY
            tf = (eng.T_INT if cfg["integer"] else 0) | stay
            if fuse > 1:
                # G row arrays (the shards of G different batches), one workspace of G x shard tableaux
                parts = [torch.as_tensor(gen(seeds[i] + 104729 * k), dtype=torch.int64).to(dev) for k in range(fuse)]
                b = eng.Batch(e, torch.cat(parts), cfg["nvar"], 0, tflags=tf, entier_bits=cfg["ebits"])
                b.parts = parts
            else:
                b = eng.Batch(e, torch.as_tensor(gen(seeds[i]), dtype=torch.int64).to(dev), cfg["nvar"], 0,
                              tflags=tf, entier_bits=cfg["ebits"])
                b.parts = None
            self.lanes.append((e, b, lane_stream(torch, dev, i)))
        self.stagger = 0.0
        self.done = [0] * depth
        self.workers, self.share, self.failed = None, None, None

    @staticmethod
    def _load(bi):
        if bi.parts is None:
            bi.load()
        else:  # the shards arrive from different batches: one load per part, one solve for all
            off = 0
            for part in bi.parts:
                bi.load_part(part, off)
                off += part.shape[0]

    def _run_async(self, share):
        """one host thread: round-robin over the lanes, `share[i]` passes on lane i"""
        left, pending = list(share), [False] * self.depth
        with self.torch.cuda.device(self.dev):
            while any(left):
                for i, (_, bi, st) in enumerate(self.lanes):
                    if not left[i]:
                        continue
                    if pending[i]:
                        bi.wait()
                    with self.torch.cuda.stream(st):
                        self._load(bi)
                        bi.solve_async()
                    pending[i] = True
                    left[i] -= 1
            for i, (_, bi, _) in enumerate(self.lanes):
                if pending[i]:
                    bi.wait()
        for i, n in enumerate(share):
            self.done[i] += n

    def _lane_steps(self, i, nsteps):
        _, bi, st = self.lanes[i]
        # lanes start a fraction of a step apart, so that one batch's under-filled last launch
        # coincides with another batch's bulk launch instead of with its last launch
        if self.stagger > 0 and i:
            time.sleep(i * self.stagger)
        with self.torch.cuda.stream(st):
            for _ in range(nsteps):
                self._load(bi)
                bi.solve()
            st.synchronize()
        self.done[i] += nsteps

    def _worker(self, i):
        """lane i's host thread, alive for the life of the Lanes: starting a Python thread costs ~0.1 ms, which a
        20-step timed region of a few tens of milliseconds would see 10-20 times over"""
        while True:
            self.go.wait()
            if self.share is None:
                return
            try:
                if self.share[i]:
                    self._lane_steps(i, self.share[i])
            except BaseException as ex:  # surfaced by run()
                self.failed = ex
            self.fin.wait()

    def run(self, nsteps):
        """`nsteps` steps (a pass of a lane with fused parts counts as `fuse` steps; nsteps is rounded up to whole
        passes); returns the passes per lane"""
        d = self.depth
        npass = (nsteps + self.fuse - 1) // self.fuse
        if not self.threads:
            share = [npass // d + (1 if i < npass % d else 0) for i in range(d)]
            self._run_async(share)
            return share
        nsteps = npass
        if self.workers is None:
            self.go, self.fin = threading.Barrier(d + 1), threading.Barrier(d + 1)
            self.workers = [threading.Thread(target=self._worker, args=(i,), daemon=True) for i in range(d)]
            for t in self.workers:
                t.start()
        self.share = [nsteps // d + (1 if i < nsteps % d else 0) for i in range(d)]
        self.go.wait()   # releases every lane at once
        self.fin.wait()  # every lane has synchronised its stream
        if self.failed is not None:
            raise self.failed
        return self.share

    def close(self):
        if self.workers is not None:
            self.share = None
            self.go.wait()
            for t in self.workers:
                t.join()
            self.workers = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def status_histogram(self):
        """PIPAMD_ST_* -> tableaux, over the lanes' last solves"""
        hist = {}
        for _, b, st in self.lanes:
            with self.torch.cuda.stream(st):
                b.fetch()
                h = self.torch.bincount(b.status.to(self.torch.int64), minlength=10).cpu().tolist()
            for k, v in enumerate(h):
                if v:
                    hist[k] = hist.get(k, 0) + v
        return hist

    def totals(self, share):
        """pivots, cuts, rows rewritten, tableaux, finished tableaux of `share[i]` passes of lane i"""
        tot = [0, 0, 0, 0, 0]
        for (e, b, _), n in zip(self.lanes, share):
            if not n:
                continue
            c = b.counters()
            tot[0] += n * c["pivots"]
            tot[1] += n * c["cuts"]
            tot[2] += n * c["rows_rewritten"]
            tot[3] += n * b.desc.batch
            tot[4] += n * c["finished"]
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


def lane_count(pipeline, steps):
    depth = max(1, min(pipeline, steps))
    # every lane should time the same number of steps: prefer a lane count that divides --steps
    for d in range(depth, max(1, depth // 2) - 1, -1):
        if steps % d == 0:
            return d
    return depth


def timed(lanes, steps, warmup, barrier, stagger_arg):
    lanes.run(max(warmup, lanes.depth))
    barrier()
    if lanes.depth > 1 and stagger_arg != 0:
        if stagger_arg > 0:
            lanes.stagger = stagger_arg * 1e-3
        else:  # one lane's own step latency with every lane busy, spread evenly over the lanes
            best = 1e9  # the quickest of three tries: one slow try (a late thread) would hold the last lane back for long
            for _ in range(3):
                lanes.stagger = 0.0
                tw = time.perf_counter()
                lanes.run(lanes.depth)
                lanes.torch.cuda.synchronize(lanes.dev)
                best = min(best, (time.perf_counter() - tw) / lanes.depth)
            lanes.stagger = best
        barrier()
    t0 = time.perf_counter()
    share = lanes.run(steps)
    barrier()
    return time.perf_counter() - t0, share


def timed_regions(lanes, steps, warmup, barrier, stagger_arg, n=3):
    """`n` timed regions of `steps` steps each behind one warm-up; sorted by duration: [(seconds, share)]"""
    out = [timed(lanes, steps, warmup, barrier, stagger_arg)]
    for _ in range(n - 1):
        barrier()
        t0 = time.perf_counter()
        share = lanes.run(steps)
        barrier()
        out.append((time.perf_counter() - t0, share))
    return sorted(out, key=lambda r: r[0])


def roofline_of(b, e, k_ms, cfg, extra=None):
    """Algorithmic HBM bytes of the pivots of one step of THIS algorithm: read
    the pivot row, write the row that replaces the entering unit row, read+write every row that
    actually changes (counted by the kernel) -- rows with a zero multiplier keep their bits --
    over the kernel's own launch durations (HIP events on its stream, un-pipelined)."""
    c = b.counters()
    eb = 16.0 if cfg["ebits"] == 128 else 8.0
    ncol = cfg["nvar"] + 1
    algo = eb * ncol * (2.0 * c["rows_rewritten"] + 2.0 * c["pivots"])
    ach = algo / (k_ms * 1e-3) / 1e9
    r = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
         "traffic": None, "kernel": "pip_advance_kernel", "kernel_ms": k_ms,
         "launches_per_step": e.last_solve_launches(),
         "avg_launch_ms": k_ms / max(1, e.last_solve_launches()),
         "algorithmic_bytes_per_step": algo,
         "rows_rewritten_per_pivot": c["rows_rewritten"] / max(1, c["pivots"])}
    if extra:
        r.update(extra)
    return r


def kernel_ms_of(b, reps=2):
    b.e.set_timing(True)
    ms = []
    for _ in range(reps):
        b.load()
        b.solve()
        ms.append(b.last_solve_ms())
    return sum(ms) / len(ms)


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
    ap.add_argument("--pipeline", type=int, default=12, help="batches in flight (streams)")
    ap.add_argument("--threads", action="store_true",
                    help="one host thread per lane calling the synchronous pipamd_batch_solve (round 2's driver) instead of "
                         "one thread over pipamd_batch_solve_async / pipamd_batch_wait")
    ap.add_argument("--fuse", type=int, default=0,
                    help="strong scaling: shards of this many batches share a workspace and a launch sequence (0 = enough "
                         "for about 5,000 tableaux per launch sequence)")
    ap.add_argument("--stagger", type=float, default=0.0,
                    help="ms between the lanes' starts (0 = none, <0 = step latency / lanes, measured in warm-up)")
    ap.add_argument("--blocking-wait", type=int, default=-1,
                    help="1: host threads sleep while the device works, 0: they poll; -1: sleep when there are more lanes than CPUs")
    ap.add_argument("--copy-rows", action="store_true", help="load copies the input rows into the job blocks (no PIPAMD_T_ROWS_STAY)")
    ap.add_argument("--no-dense", action="store_true", help="skip the row-skipping-off measurement")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-others", action="store_true", help="skip other_configs and pipeline1_value")
    args = ap.parse_args()

    # Batches in flight run on separate HIP streams; the runtime multiplexes streams onto
    # GPU_MAX_HW_QUEUES hardware queues (default 4) and kernels that share a queue serialise, so
    # give every lane a queue of its own (a HIP runtime setting, read when the runtime starts).
    # 16 queues suit 12 lanes of 10k-tableau batches best (24 or 32 queues: -3..4 %); the 24 lanes of the small
    # shards of an 8-way strong-scaling run gain 9 % from 32 (two lanes on a queue wait for each other's tails).
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    strong = args.scaling == "strong" or (args.scaling is None and world_env > 1)
    small_shards = strong and (args.batch + world_env - 1) // world_env < 2000
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "32" if small_shards else "16")
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
        """the lanes of one measurement: (cfg, lanes, depth, seeds, gen)"""
        cfg = dict(MAIN)
        cfg["batch"] = args.batch
        fuse = 1
        if scaling == "strong":
            # BASELINE configs[3]: every 10k-tableau batch is sharded over the GPUs: a GPU holds 1/world of every
            # batch in flight.  The shards are independent tableaux, so a rank loads its shards of `fuse` batches
            # into ONE workspace (pipamd_batch_load_part) and one bulk + tail launch pair serves them: about 5,000
            # tableaux per launch sequence whatever the world size, instead of 24 launch sequences of 1,250 tableaux
            # (round 2: 262 M pivots/s per GPU at 1,250, 304 M at 2,500, 350 M at 5,000, 362 M at 10,000).
            shard = max(1, (args.batch + world - 1) // world)
            fuse = args.fuse if args.fuse > 0 else max(1, min(16, 5000 // shard))
            depth = lane_count(args.pipeline, (args.steps + fuse - 1) // fuse)
            lo, hi = pdist.shard_range(args.batch, rank, world)
            cfg["batch"] = hi - lo
            seeds = [1000 + 7919 * i for i in range(depth)]

            def gen(seed):
                return synth.lexmin_batch(seed, args.batch, cfg["nvar"], cfg["ni"])[lo:hi]
        else:
            depth = lane_count(args.pipeline, args.steps)
            # lane i of rank r draws its own batch: seed 1000 + r + 7919 * i
            seeds = [pdist.shard_seed(1000, rank) + 7919 * i for i in range(depth)]
            gen = None
        return cfg, Lanes(cfg, depth, dev, local, seeds, args, gen, threads=threads, fuse=fuse), depth, seeds, gen

    progress("building the lanes")
    cfg, lanes, depth, seeds, gen = build_lanes(args.scaling, args.threads)
    progress(f"{depth} lanes ready; timing")
    my_batch = cfg["batch"]
    fuse = lanes.fuse
    e, b, _ = lanes.lanes[0]

    # three timed regions of --steps steps each; the median is the line's value, all three are printed
    regions = timed_regions(lanes, args.steps, args.warmup, barrier, args.stagger)
    dt, share = regions[1]
    steps_done = sum(share) * fuse   # == --steps unless shards are fused (whole passes of `fuse` steps)
    progress(f"regions {[round(r[0] * 1e3, 2) for r in regions]} ms")
    tot = lanes.totals(share)
    hist = lanes.status_histogram()

    # the advance kernel's own launch durations (HIP events on its stream), un-overlapped
    k_ms = kernel_ms_of(b)
    torch.cuda.synchronize(dev)
    gt, dt_max = pdist.gather_totals(tot, dt, dev)

    other = None
    if world > 1:  # the other scaling mode, same steps (every rank takes part)
        mode2 = "weak" if args.scaling == "strong" else "strong"
        rows_keep = b.rows
        lanes.close()
        del lanes
        torch.cuda.empty_cache()
        cfg2, lanes2, depth2, _, _ = build_lanes(mode2, args.threads)
        dt2, share2 = timed(lanes2, args.steps, args.warmup, barrier, args.stagger)
        gt2, dt2_max = pdist.gather_totals(lanes2.totals(share2), dt2, dev)
        other = {"scaling": mode2, "value": gt2[0] / dt2_max, "unit": "pivots/s",
                 "ms_per_step": dt2_max / (sum(share2) * lanes2.fuse) * 1e3,
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
                   "pipeline_depth": depth, "lane_seeds": "1000 + rank + 7919 * lane (strong: 1000 + 7919 * lane)",
                   "host_threads": depth if args.threads else 1,
                   "driver": "one host thread per lane, pipamd_batch_solve" if args.threads else
                             "one host thread, pipamd_batch_solve_async / pipamd_batch_wait",
                   "fused_batches_per_launch_sequence": fuse},
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
    traffic = None
    for tp in ("r02_pmc_hbm.json", "r01_pmc_hbm.json"):
        tp = os.path.join(ROOT, "profiles", tp)
        if os.path.exists(tp):
            try:
                t = json.load(open(tp))
                if t.get("batch_per_gpu") == my_batch:
                    traffic = t["hbm_bytes_per_step"]
                    break
            except Exception:
                pass
    c0 = b.counters()
    out["roofline"] = roofline_of(b, e, k_ms, cfg, {
        "traffic": traffic,
        "measured": "HIP events around each launch, 2 un-pipelined steps after the timed region "
                    "(= `bench.py --pipeline 1`, the command of profiles/r02_kernel_stats.csv)",
        "timed_region_GBps_per_gpu": 8.0 * (cfg["nvar"] + 1) * (2.0 * gt[2] + 2.0 * gt[0]) / world / dt_max / 1e9,
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
            bd = eng.Batch(ed, b.rows, cfg["nvar"], 0, tflags=eng.T_INT | eng.T_NOSKIP)
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
            one = Lanes(cfg, 1, dev, local, seeds[:1], args, gen)  # a fresh engine with its defaults for a lone batch
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
                # a 1k batch of small tableaux is a tenth of a millisecond of GPU work: twice the lanes keep the GPU fed
                od = args.pipeline * (2 if oc["batch"] < 4096 and oc["nvar"] < 128 else 1)
                ol = Lanes(oc, od, dev, local, [2000 + 7919 * i for i in range(od)], args)
                osteps = 16 * od
                # 16 steps per lane: long enough for the lanes to start a fraction of a step apart (their tails
                # then fall into other lanes' bulk phases; it costs the short headline runs more than it gives)
                # The median of three timed regions: a region here is 35 ms to 1.4 s long, and the first region of a
                # fresh set of lanes came out 2-3x slower than every later one on configs[1] (cause not found).
                regions = sorted((timed(ol, osteps, od, barrier, args.stagger if args.stagger != 0 else -1.0) for _ in range(3)),
                                 key=lambda r: r[0])
                odt, osh = regions[1]
                ot = ol.totals(osh)
                oe, ob, _ = ol.lanes[0]
                okm = kernel_ms_of(ob)
                o1 = Lanes(oc, 1, dev, local, [2000], args)
                odt1, osh1 = timed(o1, 16, 2, barrier, 0)
                ot1 = o1.totals(osh1)
                others.append({
                    "config": oc["key"], "workload": oc["workload"], "dtype": "int128" if oc["ebits"] == 128 else "int64",
                    "value": ot[0] / odt, "unit": "pivots/s", "ms_per_step": odt / osteps * 1e3, "steps": osteps,
                    "pipeline_depth": od, "regions_ms": [round(r[0] * 1e3, 3) for r in regions], "problems_per_sec": ot[3] / odt, "pivots_per_step": ot[0] / osteps,
                    "finished_fraction": ot[4] / max(1, ot[3]),
                    "pipeline1_value": ot1[0] / odt1, "pipeline1_ms_per_step": odt1 / 16 * 1e3,
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
        rows_h = gen(seeds[0]) if gen else synth.lexmin_batch(seeds[0], args.batch, cfg["nvar"], cfg["ni"])
        cb = cpu_baseline(rows_h, cfg["nvar"], cfg["ni"])
        if cb:
            out["cpu_baseline"] = cb
            out["speedup_vs_cpu_baseline"] = out["value"] / cb["value"]
    print(json.dumps(out), flush=True)
    pdist.finish()


if __name__ == "__main__":
    main()
