"""ctypes binding of the C ABI (include/piplib_amd.h) for tests and bench.py.

torch is used only for device memory and streams.  There is NO CPU fallback: if
libpipamd.so is missing or no GPU is visible, construction raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# PIPAMD_LIB: tuning experiments with an alternative build of the same library (tools/ only)
LIB_PATH = os.environ.get("PIPAMD_LIB") or os.path.join(HERE, "libpipamd.so")

ST_RUN, ST_SOLUTION, ST_NIL, ST_NEED_COMPA, ST_NEED_PARMCUT, ST_OVERFLOW, ST_CAPACITY, ST_RANGE, ST_INTERNAL, ST_MAXCOL = range(10)
T_INT, T_DUAL = 1, 2
T_NOSKIP = 2048
T_ROWS_STAY = 8192  # the rows of Batch.load stay valid until the next solve: no copy pass (include/piplib_amd.h)


class BatchDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("batch", "nvar", "nparm", "ni", "bigparm", "tflags", "cap_cuts", "cap_newparm", "entier_bits")]


ABI_VERSION = 400  # include/piplib_amd.h PIPAMD_VERSION
_lib = None


def lib():
    """Load libpipamd.so (built by piplib_amd.build); raises if it is not there."""
    global _lib
    if _lib is None:
        # torch first: its wheel bundles the HIP runtime; loading ours before it would put a
        # second libamdhip64 in the process and torch would then see no GPU.
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} missing: run `python -m piplib_amd.build` (HIP extension is mandatory)")
        L = C.CDLL(LIB_PATH)
        if L.pipamd_version() != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} is interface version {L.pipamd_version()}, this binding is written for "
                               f"{ABI_VERSION} (include/piplib_amd.h PIPAMD_VERSION): rebuild with `python -m piplib_amd.build`")
        L.pipamd_last_error.restype = C.c_char_p
        L.pipamd_batch_workspace_bytes.restype = C.c_size_t
        L.pipamd_batch_workspace_bytes.argtypes = [C.POINTER(BatchDesc)]
        L.pipamd_dense_pivot_bytes.restype = C.c_size_t
        L.pipamd_dense_pivot_bytes.argtypes = [C.POINTER(BatchDesc)]
        L.pipamd_engine_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        L.pipamd_engine_destroy.argtypes = [C.c_void_p]
        L.pipamd_engine_set_iter_limit.argtypes = [C.c_void_p, C.c_int]
        L.pipamd_engine_set_waves_per_job.argtypes = [C.c_void_p, C.c_int]
        L.pipamd_engine_set_round_pivots.argtypes = [C.c_void_p, C.c_int]
        if hasattr(L, "pipamd_engine_set_round_rows"):
            L.pipamd_engine_set_round_rows.argtypes = [C.c_void_p, C.c_int]
        L.pipamd_last_solve_launches.argtypes = [C.c_void_p]
        L.pipamd_batch_load.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(BatchDesc), C.c_void_p, C.c_void_p]
        L.pipamd_batch_solve.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(BatchDesc), C.c_void_p]
        L.pipamd_batch_solve_async.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(BatchDesc), C.c_void_p]
        L.pipamd_batch_wait.argtypes = [C.c_void_p]
        L.pipamd_batch_poll.argtypes = [C.c_void_p]
        L.pipamd_batch_load_part.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(BatchDesc), C.c_void_p, C.c_int, C.c_int,
                                             C.c_void_p]
        L.pipamd_batch_results.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(BatchDesc)] + [C.c_void_p] * 6
        L.pipamd_batch_counters.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(BatchDesc), C.c_void_p, C.c_void_p]
        L.pipamd_last_solve_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.pipamd_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise RuntimeError(f"piplib_amd error {rc}: {lib().pipamd_last_error().decode()}")


class Engine:
    def __init__(self, device=0):
        self._h = C.c_void_p()
        _check(lib().pipamd_engine_create(C.byref(self._h), int(device)))
        self.device = int(device)

    def close(self):
        if self._h:
            lib().pipamd_engine_destroy(self._h)
            self._h = C.c_void_p()

    def set_iter_limit(self, n):
        _check(lib().pipamd_engine_set_iter_limit(self._h, int(n)))

    def set_round_pivots(self, n):
        _check(lib().pipamd_engine_set_round_pivots(self._h, int(n)))

    def set_bulk_min(self, n):
        lib().pipamd_engine_set_bulk_min.argtypes = [C.c_void_p, C.c_int]
        _check(lib().pipamd_engine_set_bulk_min(self._h, int(n)))

    def set_device_tree(self, on):
        """pipamd_solve_tableaux_lockstep: try the device-resident traiter() first (default on)"""
        lib().pipamd_engine_set_device_tree.argtypes = [C.c_void_p, C.c_int]
        _check(lib().pipamd_engine_set_device_tree(self._h, int(bool(on))))

    def last_device_tree(self):
        """(problems the device tree finished, problems it handed back) in the last lock-step call"""
        a, b = C.c_int(0), C.c_int(0)
        lib().pipamd_last_device_tree.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        _check(lib().pipamd_last_device_tree(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_max_rows(self, rows):
        lib().pipamd_engine_set_max_rows.argtypes = [C.c_void_p, C.c_int]
        _check(lib().pipamd_engine_set_max_rows(self._h, int(rows)))

    def set_blocking_wait(self, on):
        lib().pipamd_engine_set_blocking_wait.argtypes = [C.c_void_p, C.c_int]
        _check(lib().pipamd_engine_set_blocking_wait(self._h, int(bool(on))))

    def set_lone_batches(self, on):
        """this engine's batches run one at a time (pipamd_engine_set_lone_batches)"""
        lib().pipamd_engine_set_lone_batches.argtypes = [C.c_void_p, C.c_int]
        _check(lib().pipamd_engine_set_lone_batches(self._h, int(bool(on))))

    def set_lean64(self, on):
        """128-bit batches of 129 ... 256 columns start with the lean kernel of csrc/pip_lean64.h (default off)"""
        lib().pipamd_engine_set_lean64.argtypes = [C.c_void_p, C.c_int]
        _check(lib().pipamd_engine_set_lean64(self._h, int(bool(on))))

    def set_tail_waves(self, n):
        lib().pipamd_engine_set_tail_waves.argtypes = [C.c_void_p, C.c_int]
        _check(lib().pipamd_engine_set_tail_waves(self._h, int(n)))

    def set_timing(self, on):
        lib().pipamd_engine_set_timing.argtypes = [C.c_void_p, C.c_int]
        _check(lib().pipamd_engine_set_timing(self._h, int(bool(on))))

    def set_round_rows(self, n):
        _check(lib().pipamd_engine_set_round_rows(self._h, int(n)))

    def last_launch_ms(self, i):
        """duration of launch i of the last solve (HIP events on its stream; timing must be on)"""
        ms = C.c_float()
        lib().pipamd_last_launch_ms.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float)]
        _check(lib().pipamd_last_launch_ms(self._h, int(i), C.byref(ms)))
        return float(ms.value)

    def debug_single_launch(self, on):
        """measurement aid: the next solves stop after their bulk launches (True / 1: the one-wave launches of a large
        batch; 2: after the lean launch alone, when the batch has one) -- unfinished tableaux stay PIPAMD_ST_RUN"""
        lib().pipamd_debug_single_launch.argtypes = [C.c_void_p, C.c_int]
        _check(lib().pipamd_debug_single_launch(self._h, int(on)))

    def debug_lean(self, on):
        """measurement / testing aid: bulk launches with (default) or without the lean kernel (csrc/pip_lean.h)"""
        lib().pipamd_debug_lean.argtypes = [C.c_void_p, C.c_int]
        _check(lib().pipamd_debug_lean(self._h, int(bool(on))))

    def last_solve_launches(self):
        return int(lib().pipamd_last_solve_launches(self._h))

    def set_waves_per_job(self, n):
        _check(lib().pipamd_engine_set_waves_per_job(self._h, int(n)))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def wide_to_int(a):
    """numpy (..., 2) int64 (low, high) pairs -> object array of Python ints (two's complement)."""
    import numpy as np
    lo = a[..., 0].astype(object) & ((1 << 64) - 1)
    hi = a[..., 1].astype(object)
    return hi * (1 << 64) + lo


class Batch:
    """A uniform batch of tableaux resident in HBM (layer 1 of the C ABI)."""

    def __init__(self, engine, rows, nvar, nparm, bigparm=-1, tflags=T_INT, cap_cuts=None, cap_newparm=0,
                 entier_bits=64, shape=None):
        """rows: (batch, ni, ncol) int64, host or device; or None with shape=(batch, ni, ncol) for a workspace whose
        tableaux come from load_parts()"""
        import torch
        self.torch = torch
        self.e = engine
        B, ni, ncol = rows.shape if rows is not None else shape
        assert ncol == nvar + nparm + 1
        if cap_cuts is None:
            cap_cuts = max(0, min(ni + 64, 2048 - ni)) if (tflags & T_INT) else 0
        self.desc = BatchDesc(B, nvar, nparm, ni, bigparm, tflags, cap_cuts, cap_newparm, entier_bits)
        self.entier_bits = entier_bits
        ew = 2 if entier_bits == 128 else 1
        self.dev = torch.device("cuda", engine.device)
        if rows is None:
            self.rows = None
        else:
            self.rows = rows if (torch.is_tensor(rows) and rows.is_cuda) else torch.as_tensor(rows, dtype=torch.int64).to(self.dev)
            self.rows = self.rows.contiguous()
        nbytes = lib().pipamd_batch_workspace_bytes(C.byref(self.desc))
        if nbytes == 0:
            raise RuntimeError(lib().pipamd_last_error().decode())
        self.ws = torch.empty(nbytes // 8 + 1, dtype=torch.int64, device=self.dev)
        self.status = torch.empty(B, dtype=torch.int32, device=self.dev)
        self.pivots = torch.empty(B, dtype=torch.int32, device=self.dev)
        self.cuts = torch.empty(B, dtype=torch.int32, device=self.dev)
        # 128-bit values come back as (low, high) int64 pairs: see wide_to_int()
        self.sol_num = torch.empty((B, nvar, nparm + 1) + ((2,) if ew == 2 else ()), dtype=torch.int64, device=self.dev)
        self.sol_den = torch.empty((B, nvar) + ((2,) if ew == 2 else ()), dtype=torch.int64, device=self.dev)

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.dev).cuda_stream)

    def load(self):
        _check(lib().pipamd_batch_load(self.e._h, C.c_void_p(self.ws.data_ptr()), C.byref(self.desc),
                                       C.c_void_p(self.rows.data_ptr()), self._stream()))

    def load_part(self, rows, first, stream=None):
        """tableaux first .. first + len(rows) - 1 of the batch from a resident row array (pipamd_batch_load_part)"""
        assert rows.is_cuda and rows.is_contiguous()
        st = C.c_void_p(stream) if stream is not None else self._stream()
        _check(lib().pipamd_batch_load_part(self.e._h, C.c_void_p(self.ws.data_ptr()), C.byref(self.desc),
                                            C.c_void_p(rows.data_ptr()), int(first), int(rows.shape[0]), st))

    def load_parts(self, parts, stream=None):
        """the whole batch from a list of resident row arrays (one pipamd_batch_load_part each)"""
        off = 0
        for part in parts:
            self.load_part(part, off, stream)
            off += part.shape[0]
        assert off == self.desc.batch

    def solve(self, stream=None):
        st = C.c_void_p(stream) if stream is not None else self._stream()
        _check(lib().pipamd_batch_solve(self.e._h, C.c_void_p(self.ws.data_ptr()), C.byref(self.desc), st))

    def solve_async(self, stream=None):
        """pipamd_batch_solve_async on `stream` (a raw hipStream_t handle; default: torch's current stream); one solve
        in flight per engine -- wait() before the next one, and before fetch()."""
        st = C.c_void_p(stream) if stream is not None else self._stream()
        _check(lib().pipamd_batch_solve_async(self.e._h, C.c_void_p(self.ws.data_ptr()), C.byref(self.desc), st))

    def wait(self):
        _check(lib().pipamd_batch_wait(self.e._h))

    def poll(self):
        """pipamd_batch_poll: 1 = the batch is done, 0 = still running (never blocks)"""
        rc = int(lib().pipamd_batch_poll(self.e._h))
        if rc < 0:
            _check(rc)
        return rc

    def fetch(self, stream=None):
        st = C.c_void_p(stream) if stream is not None else self._stream()
        _check(lib().pipamd_batch_results(self.e._h, C.c_void_p(self.ws.data_ptr()), C.byref(self.desc),
                                          C.c_void_p(self.status.data_ptr()), C.c_void_p(self.pivots.data_ptr()),
                                          C.c_void_p(self.cuts.data_ptr()), C.c_void_p(self.sol_num.data_ptr()),
                                          C.c_void_p(self.sol_den.data_ptr()), st))

    def counters(self):
        """dict of batch totals (pivots, cuts, rows_rewritten, finished) -- synchronises."""
        out = self.torch.zeros(4, dtype=self.torch.int64, device=self.dev)
        _check(lib().pipamd_batch_counters(self.e._h, C.c_void_p(self.ws.data_ptr()), C.byref(self.desc),
                                           C.c_void_p(out.data_ptr()), self._stream()))
        v = out.cpu().tolist()
        return {"pivots": v[0], "cuts": v[1], "rows_rewritten": v[2], "finished": v[3]}

    def last_solve_ms(self):
        ms = C.c_float()
        _check(lib().pipamd_last_solve_ms(self.e._h, C.byref(ms)))
        return float(ms.value)

    def pivot_bytes(self):
        return int(lib().pipamd_dense_pivot_bytes(C.byref(self.desc)))


def slow_converging(engine, rows, nvar, cut_rows=448, entier_bits=64):
    """Indices of the tableaux of an integer batch (nparm = 0) on which Gomory's cuts have not converged within
    `cut_rows` cut rows: the batch is solved once with that row budget (pipamd_engine_set_max_rows) and the tableaux left
    at PIPAMD_ST_CAPACITY are returned.  A property of the tableau and of the reference's algorithm (the number of
    cuts integrer() adds is the same on any correct implementation), not of this engine: on the synthetic families of
    piplib_amd/synth.py about 3 tableaux in 100,000 are of this kind, and on them the reference's own loop
    (traiter.c:665-789, unbounded expanser) does not end within minutes on a CPU either.  bench.py replaces them by
    their neighbours before it times anything (a benchmark workload must be one the reference finishes)."""
    import torch
    e2 = Engine(engine.device)
    ni = int(rows.shape[1])
    e2.set_max_rows(ni + cut_rows)
    b = Batch(e2, rows, nvar, 0, tflags=T_INT, entier_bits=entier_bits)
    b.load()
    b.solve()
    b.fetch()
    torch.cuda.synchronize(b.dev)
    idx = torch.nonzero(b.status == ST_CAPACITY).flatten().cpu().tolist()
    del b
    e2.close()
    return idx


class SolCell(C.Structure):
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("param1", C.c_int64), ("param2", C.c_int64)]


SOL_NIL, SOL_IF, SOL_LIST, SOL_FORM, SOL_NEW, SOL_DIV, SOL_VAL = range(1, 8)


def _frac(n, d):
    import math
    g = math.gcd(n, d)
    return f" {n // g}" if g == abs(d) and d > 0 else (f" {n // g}/{d // g}" if g else f" {n}/{d}")


def tape_text(cells):
    """Test/bench harness only: the text the reference's sol_edit (sol.c:291-422) prints for a tape,
    cells = [(kind, param1, param2), ...].  An empty tape is the front ends' "void"."""
    if not cells:
        return "void\n"
    out = []

    def item(i):
        while cells[i][0] == SOL_NEW:
            out.append("(newparm %d " % cells[i][1])
            i = item(i + 1)
            out.append(")\n")
        k, a, b = cells[i]
        if k == SOL_NIL:
            out.append("()\n")
            return i + 1
        if k == SOL_IF:
            out.append("(if ")
            i = item(item(item(i + 1)))
            out.append(")\n")
            return i
        if k == SOL_LIST:
            out.append("(list ")
            i += 1
            for _ in range(a):
                i = item(i)
            out.append(")\n")
            return i
        if k == SOL_FORM:
            out.append("#[")
            for j in range(a):
                out.append(_frac(cells[i + 1 + j][1], cells[i + 1 + j][2]))
            out.append("]\n")
            return i + 1 + a
        if k == SOL_DIV:
            out.append("(div ")
            i = item(item(i + 1))
            out.append(")\n")
            return i
        if k == SOL_VAL:
            out.append(_frac(a, b))
            return i + 1
        raise ValueError(f"unknown tape cell kind {k}")

    i = 0
    while i < len(cells):
        i = item(i)
    return "".join(out)


def _take_cells(ptr, n):
    if not ptr or not n:
        return []
    arr = C.cast(ptr, C.POINTER(SolCell * n)).contents
    cells = [(c.kind, c.param1, c.param2) for c in arr]
    lib().pipamd_free(ptr)
    return cells


def _rows(ineq, ctx, ni, nc, nvar, nparm):
    import numpy as np
    a = np.ascontiguousarray(ineq, dtype=np.int64).reshape(ni, nvar + nparm + 1)
    c = np.ascontiguousarray(ctx, dtype=np.int64).reshape(nc, nparm + 1)
    return a, c


class SolCell128(C.Structure):
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("p1lo", C.c_int64), ("p1hi", C.c_int64),
                ("p2lo", C.c_int64), ("p2hi", C.c_int64)]


def _take_cells128(ptr, n):
    if not ptr or not n:
        return []
    arr = C.cast(ptr, C.POINTER(SolCell128 * n)).contents
    m = (1 << 64) - 1
    cells = [(c.kind, (c.p1hi << 64) | (c.p1lo & m), (c.p2hi << 64) | (c.p2lo & m)) for c in arr]
    lib().pipamd_free(ptr)
    return cells


def solve_tableau_cells(engine, nvar, nparm, ni, nc, bigparm, nq, ineq, ctx, simplify=True, deepest_cut=False, bits=64):
    """Layer 3: one problem in .dat form (host arrays) -> (tape cells, pivots); bits=128 runs the
    overflow-safe flavour (128-bit entries on the device and in the host tree).
    Raises SolverError(status) where the reference would have exit()ed."""
    a, c = _rows(ineq, ctx, ni, nc, nvar, nparm)
    L = lib()
    fn = L.pipamd_solve_tableau128 if bits == 128 else L.pipamd_solve_tableau
    fn.argtypes = [C.c_void_p] + [C.c_int] * 6 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                                  C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                                  C.POINTER(C.c_int), C.POINTER(C.c_int64)]
    cells, n = C.c_void_p(), C.c_size_t(0)
    status, piv = C.c_int(0), C.c_int64(0)
    rc = fn(engine._h, nvar, nparm, ni, nc, bigparm, nq, C.c_void_p(a.ctypes.data),
            C.c_void_p(c.ctypes.data), int(bool(simplify)), int(bool(deepest_cut)),
            C.byref(cells), C.byref(n), C.byref(status), C.byref(piv))
    if rc == -5:
        raise SolverError(status.value, piv.value)
    _check(rc)
    return (_take_cells128 if bits == 128 else _take_cells)(cells, n.value), piv.value


def solve_tableau(engine, nvar, nparm, ni, nc, bigparm, nq, ineq, ctx, simplify=True, deepest_cut=False, bits=64):
    """solve_tableau_cells, with the tape printed as sol_edit would (tests compare texts)."""
    cells, piv = solve_tableau_cells(engine, nvar, nparm, ni, nc, bigparm, nq, ineq, ctx, simplify, deepest_cut, bits)
    return tape_text(cells), piv


def traiter(engine, nvar, nparm, ni, nc, bigparm, flags, tableau, context, deepest_cut=False, bits=64):
    """pipamd_traiter / pipamd_traiter128: one traiter() call -> (tape cells, pivots)."""
    a, c = _rows(tableau, context, ni, nc, nvar, nparm)
    L = lib()
    fn = L.pipamd_traiter128 if bits == 128 else L.pipamd_traiter
    fn.argtypes = [C.c_void_p] + [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p),
                                                  C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_int64)]
    cells, n = C.c_void_p(), C.c_size_t(0)
    status, piv = C.c_int(0), C.c_int64(0)
    rc = fn(engine._h, nvar, nparm, ni, nc, bigparm, int(flags), int(bool(deepest_cut)),
            C.c_void_p(a.ctypes.data), C.c_void_p(c.ctypes.data), C.byref(cells), C.byref(n),
            C.byref(status), C.byref(piv))
    if rc == -5:
        raise SolverError(status.value, piv.value)
    _check(rc)
    return (_take_cells128 if bits == 128 else _take_cells)(cells, n.value), piv.value


class SolverError(RuntimeError):
    def __init__(self, status, pivots=0):
        super().__init__(f"solver stopped with PIPAMD_ST status {status}")
        self.status = status
        self.pivots = pivots


class PipProblem(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("nvar", "nparm", "ni", "nc", "bigparm", "nq")] + \
               [("ineq", C.c_void_p), ("ctx", C.c_void_p)]


class PreparedProblems:
    """ctypes view of a list of problems (objects with nvar, nparm, ni, nc, bigparm, nq, ineq, ctx) and the
    result arrays of one pipamd_solve_tableaux* call"""

    def __init__(self, problems):
        import numpy as np
        self.n = n = len(problems)
        self.arr = (PipProblem * max(1, n))()
        self.keep = []
        for i, p in enumerate(problems):
            a = np.ascontiguousarray(p.ineq, dtype=np.int64).reshape(p.ni, p.nvar + p.nparm + 1)
            c = np.ascontiguousarray(p.ctx, dtype=np.int64).reshape(p.nc, p.nparm + 1)
            self.keep += [a, c]
            self.arr[i] = PipProblem(p.nvar, p.nparm, p.ni, p.nc, p.bigparm, p.nq, a.ctypes.data, c.ctypes.data)
        self.cells = (C.c_void_p * max(1, n))()
        self.ncell = (C.c_size_t * max(1, n))()
        self.rcs = (C.c_int * max(1, n))()
        self.sts = (C.c_int * max(1, n))()
        self.piv = (C.c_int64 * max(1, n))()

    def results(self):
        """[(text | None, rc, status, pivots)]; frees the cells"""
        out = []
        for i in range(self.n):
            t = tape_text(_take_cells(self.cells[i], self.ncell[i])) if self.rcs[i] == 0 else None
            out.append((t, self.rcs[i], self.sts[i], self.piv[i]))
        return out


def solve_tableaux128(engine, problems, simplify=True, deepest_cut=False, nthreads=8):
    """Many problems through pipamd_solve_tableaux128 (128-bit entries, a host thread and a TreeT<__int128> each).
    Returns a list of (text | None, rc, status, pivots)."""
    prep = PreparedProblems(problems)
    L = lib()
    L.pipamd_solve_tableaux128.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _check(L.pipamd_solve_tableaux128(engine._h, prep.n, prep.arr, int(bool(simplify)), int(bool(deepest_cut)), int(nthreads),
                                      prep.cells, prep.ncell, prep.rcs, prep.sts, prep.piv))
    out = []
    for i in range(prep.n):
        t = tape_text(_take_cells128(prep.cells[i], prep.ncell[i])) if prep.rcs[i] == 0 else None
        out.append((t, prep.rcs[i], prep.sts[i], prep.piv[i]))
    return out


def solve_tableaux_lockstep128(engine, problems, simplify=True, deepest_cut=False):
    """Many problems through pipamd_solve_tableaux_lockstep128 (128-bit entries, the lock-step scheduler).
    Returns a list of (text | None, rc, status, pivots)."""
    prep = PreparedProblems(problems)
    L = lib()
    L.pipamd_solve_tableaux_lockstep128.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _check(L.pipamd_solve_tableaux_lockstep128(engine._h, prep.n, prep.arr, int(bool(simplify)), int(bool(deepest_cut)),
                                               prep.cells, prep.ncell, prep.rcs, prep.sts, prep.piv))
    out = []
    for i in range(prep.n):
        t = tape_text(_take_cells128(prep.cells[i], prep.ncell[i])) if prep.rcs[i] == 0 else None
        out.append((t, prep.rcs[i], prep.sts[i], prep.piv[i]))
    return out


def solve_prepared(engine, prep, simplify=True, deepest_cut=False, nthreads=8, lockstep=False):
    """the bare C call on prepared problems (what a C caller pays); prep.results() turns the cells into text"""
    L = lib()
    L.pipamd_solve_tableaux.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    if lockstep:
        L.pipamd_solve_tableaux_lockstep.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _check(L.pipamd_solve_tableaux_lockstep(engine._h, prep.n, prep.arr, int(bool(simplify)), int(bool(deepest_cut)),
                                                prep.cells, prep.ncell, prep.rcs, prep.sts, prep.piv))
    else:
        _check(L.pipamd_solve_tableaux(engine._h, prep.n, prep.arr, int(bool(simplify)), int(bool(deepest_cut)),
                                       int(nthreads), prep.cells, prep.ncell, prep.rcs, prep.sts, prep.piv))


def solve_tableaux(engine, problems, simplify=True, deepest_cut=False, nthreads=8, lockstep=False):
    """Many problems (objects with nvar, nparm, ni, nc, bigparm, nq, ineq, ctx) through
    pipamd_solve_tableaux / pipamd_solve_tableaux_lockstep.  Returns a list of (text | None, rc, status, pivots)."""
    prep = PreparedProblems(problems)
    solve_prepared(engine, prep, simplify, deepest_cut, nthreads, lockstep)
    return prep.results()
