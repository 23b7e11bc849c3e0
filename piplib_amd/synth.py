"""Synthetic PIP tableaux (deterministic, numpy only).

Column order is PIP's own (reference tab.c:222-248 / maind.c:190):
    unknowns (nvar) | constant | parameters (nparm)
A row ``r`` means  sum_j r[j]*x_j + r[nvar] + sum_k r[nvar+1+k]*p_k >= 0.
"""
from dataclasses import dataclass

import numpy as np


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


def lexmin_rows(rng, nvar, ni, nnz=4, cmax=5, x0max=9, slackmax=3, pneg=0.25):
    """One sparse 'polyhedral-like' system A x >= b around a hidden integer point x0.

    Each constraint has 2..nnz non-zeros of magnitude <= cmax; b = A x0 - slack keeps x0
    feasible, so lexmin exists; non-unimodular coefficients make the rational optimum
    fractional, which is what exercises the Gomory-cut generator.
    """
    A = np.zeros((ni, nvar), dtype=np.int64)
    for i in range(ni):
        k = int(rng.integers(2, nnz + 1))
        cols = rng.choice(nvar, size=min(k, nvar), replace=False)
        vals = rng.integers(1, cmax + 1, size=len(cols))
        vals = np.where(rng.random(len(cols)) < pneg, -vals, vals)
        A[i, cols] = vals
    x0 = rng.integers(0, x0max + 1, size=nvar)
    slack = rng.integers(0, slackmax + 1, size=ni)
    b = A @ x0 - slack
    return np.concatenate([A, -b[:, None]], axis=1)


def lexmin_batch(seed, batch, nvar, ni, nnz=4, cmax=5, x0max=9, slackmax=3, pneg=0.25):
    """(batch, ni, nvar+1) int64: non-parametric lexmin problems (nparm = 0), vectorised.

    Same family as lexmin_rows (sparse rows, hidden feasible integer point); duplicate
    column draws inside a row simply collapse, so a row has 1..nnz non-zeros."""
    rng = np.random.default_rng(seed)
    k = rng.integers(2, nnz + 1, size=(batch, ni))
    cols = rng.integers(0, nvar, size=(batch, ni, nnz))
    vals = rng.integers(1, cmax + 1, size=(batch, ni, nnz))
    vals = np.where(rng.random((batch, ni, nnz)) < pneg, -vals, vals)
    live = np.arange(nnz)[None, None, :] < k[..., None]
    T = np.zeros((batch, ni, nvar + 1), dtype=np.int64)
    A = T[:, :, :nvar]
    bi, ii, _ = np.nonzero(live)
    A[bi, ii, cols[live]] = vals[live]
    x0 = rng.integers(0, x0max + 1, size=(batch, nvar))
    slack = rng.integers(0, slackmax + 1, size=(batch, ni))
    T[:, :, nvar] = slack - np.einsum("bij,bj->bi", A, x0)
    return T


def random_problems(seed, count, nvar, nparm, ni, nc, nq, cmax=4, bmax=12):
    """Small dense-ish random problems, parametric when nparm > 0 (many are infeasible
    or split several times: good coverage of Nil / if / newparm paths)."""
    rng = np.random.default_rng(seed)
    out = []
    ncol = nvar + nparm + 1
    for _ in range(count):
        T = rng.integers(-cmax, cmax + 1, size=(ni, ncol)).astype(np.int64)
        T[rng.random((ni, ncol)) < 0.45] = 0
        T[:, nvar] = rng.integers(-bmax, bmax + 1, size=ni)
        C = rng.integers(-cmax, cmax + 1, size=(nc, nparm + 1)).astype(np.int64)
        if nc:
            C[:, nparm] = rng.integers(0, bmax + 1, size=nc)
        out.append(Problem(nvar, nparm, ni, nc, -1, nq, T, C))
    return out


def dense_batch(seed, batch, nvar, ni, cmax=40, x0max=9, pzero=0.3):
    """(batch, ni, nvar+1) int64: dense lexmin problems with large coefficients.  Their
    determinants outgrow 64 bits quickly -- the reference's int64 build stops with "Integer
    overflow" on many of them -- which is what the 128-bit Entier variant is for."""
    rng = np.random.default_rng(seed)
    A = rng.integers(-cmax, cmax + 1, size=(batch, ni, nvar)).astype(np.int64)
    A[rng.random((batch, ni, nvar)) < pzero] = 0
    x0 = rng.integers(0, x0max + 1, size=(batch, nvar))
    slack = rng.integers(0, 4, size=(batch, ni))
    T = np.zeros((batch, ni, nvar + 1), dtype=np.int64)
    T[:, :, :nvar] = A
    T[:, :, nvar] = slack - np.einsum("bij,bj->bi", A, x0)
    return T
