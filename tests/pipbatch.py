"""Test helper: binary batch exchange with the two CPU checkers.

* ``oracle/_ref/refpip``   -- the real reference (int64 build) behind our driver
* ``oracle/oraclepip``     -- the CPU restatement (oracle/pip_oracle.c)

Format: oracle/batchfmt.h.  TEST INFRASTRUCTURE ONLY.
"""
import os
import struct
import subprocess
import tempfile
from dataclasses import dataclass
from typing import List, Optional

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFPIP = os.path.join(ROOT, "oracle", "_ref", "refpip")
REFPIP_GPU = os.path.join(ROOT, "oracle", "_ref", "refpip_gpu")  # the reference's front ends over the GPU traiter hook
REFPIP_GMP = os.path.join(ROOT, "oracle", "_ref", "refpip_gmp")  # the reference's arbitrary-precision flavour (build container only)
ORACLEPIP = os.path.join(ROOT, "oracle", "oraclepip")
ORACLEPIP128 = os.path.join(ROOT, "oracle", "oraclepip128")
MAGIC = 0x50495042

F_NOTEXT, F_NOSIMPLIFY, F_DEEPEST = 1, 2, 4
ST_OK, ST_VOID, ST_ABORT = 0, 1, 2


@dataclass
class Problem:
    nvar: int
    nparm: int
    ni: int
    nc: int
    bigparm: int
    nq: int
    ineq: np.ndarray  # (ni, nvar+nparm+1) int64
    ctx: np.ndarray   # (nc, nparm+1) int64


@dataclass
class Result:
    status: int
    abort_code: int
    pivots: int
    text: str
    entry_bits: int = 0  # refpip_gmp only: widest value the reference formed outside the determinant ...
    det_bits: int = 0    # ... and the widest determinant (oracle/ref_driver.c, width tracker)


@dataclass
class BatchOut:
    results: List[Result]
    solve_seconds: float
    total_pivots: int


def have_ref() -> bool:
    return os.access(REFPIP, os.X_OK)


def have_ref_gpu() -> bool:
    return os.access(REFPIP_GPU, os.X_OK)


def have_oracle() -> bool:
    return os.access(ORACLEPIP, os.X_OK)


def write_batch(path: str, probs: List[Problem], flags: int = 0) -> None:
    with open(path, "wb") as f:
        f.write(struct.pack("<IIII", MAGIC, len(probs), flags, 0))
        for p in probs:
            f.write(struct.pack("<6i", p.nvar, p.nparm, p.ni, p.nc, p.bigparm, p.nq))
            a = np.ascontiguousarray(p.ineq, dtype="<i8").reshape(p.ni, p.nvar + p.nparm + 1)
            c = np.ascontiguousarray(p.ctx, dtype="<i8").reshape(p.nc, p.nparm + 1)
            f.write(a.tobytes())
            f.write(c.tobytes())


def read_batch_out(path: str) -> BatchOut:
    with open(path, "rb") as f:
        data = f.read()
    magic, count, secs, total = struct.unpack_from("<IIdq", data, 0)
    assert magic == MAGIC
    off = struct.calcsize("<IIdq")
    res = []
    for _ in range(count):
        status, code, piv, tlen, r_ = struct.unpack_from("<iiqII", data, off)
        off += struct.calcsize("<iiqII")
        res.append(Result(status, code, piv, data[off:off + tlen].decode(), r_ & 0xffff, r_ >> 16))
        off += tlen
    return BatchOut(res, secs, total)


def run_batch(exe: str, probs: List[Problem], flags: int = 0, timeout: Optional[float] = 600) -> BatchOut:
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        write_batch(fin, probs, flags)
        p = subprocess.run([exe, "batch", fin, fout], capture_output=True, timeout=timeout)
        if p.returncode != 0:
            raise RuntimeError(f"{exe} batch failed rc={p.returncode}: {p.stderr.decode()[:500]}")
        return read_batch_out(fout)


def squash(s: str) -> str:
    """The reference's tests compare with ``diff -w``: drop all whitespace."""
    return "".join(s.split())
