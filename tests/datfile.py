"""Reader for PIP's native .dat input (what the reference's maind.c:146-196 + tab.c:222-248
read): "( comment ... ) nvar nparm ni nc bigparm nq ( #[row] ... ) ( #[row] ... )" groups.
Test helper only: the product takes arrays (pipamd_solve_tableau)."""
import numpy as np


class _Rd:
    def __init__(self, text):
        self.t, self.i = text, 0

    def getc(self):
        if self.i >= len(self.t):
            return ""
        c = self.t[self.i]
        self.i += 1
        return c

    def until(self, ch):
        while True:
            c = self.getc()
            if c == "" or c == ch:
                return c

    def integer(self):
        while self.i < len(self.t) and self.t[self.i] in " \n\t\r":
            self.i += 1
        j = self.i
        if j < len(self.t) and self.t[j] in "+-":
            j += 1
        k = j
        while k < len(self.t) and self.t[k].isdigit():
            k += 1
        if k == j:
            return None
        v = int(self.t[self.i:k])
        self.i = k
        return v


def _tableau(rd, h, w):
    rd.until("(")
    m = np.zeros((h, w), dtype=object)
    for i in range(h):
        rd.until("[")
        for j in range(w):
            v = rd.integer()
            if v is None:
                return None
            m[i, j] = v
    rd.until("]")
    return m


def read_dat(path):
    """Yields dicts: comment, nvar, nparm, ni, nc, bigparm, nq, ineq, ctx for each problem."""
    text = open(path, encoding="latin-1").read()
    rd = _Rd(text)
    out = []
    while True:
        c = rd.getc()
        if c == "":
            break
        if c != "(":
            continue
        level, com = 0, []
        while True:
            c = rd.getc()
            if c == "":
                break
            if c == "(":
                level += 1
            elif c == ")":
                level -= 1
                if level == 0:
                    break
            com.append(c)
        hdr = [rd.integer() for _ in range(6)]
        if any(h is None for h in hdr):
            break
        nvar, nparm, ni, nc, bigparm, nq = hdr
        ineq = _tableau(rd, ni, nvar + nparm + 1)
        ctx = _tableau(rd, nc, nparm + 1)
        if ineq is None or ctx is None:
            break
        out.append(dict(comment="".join(com), nvar=nvar, nparm=nparm, ni=ni, nc=nc, bigparm=bigparm, nq=nq,
                        ineq=ineq.astype(np.int64), ctx=ctx.astype(np.int64)))
    return out


def read_pip(path):
    """The stdin protocol of the reference's example/example.c:72-108: context matrix, bignum,
    domain matrix (PolyLib format, '#' comments), then option keywords."""
    lines = open(path, encoding="latin-1").read().split("\n")
    pos = 0

    def matrix():
        nonlocal pos
        while True:
            s = lines[pos]
            pos += 1
            t = s.split()
            if s.startswith("#") or not t:
                continue
            if len(t) >= 2 and t[0].isdigit() and t[1].isdigit():
                nr, nc = int(t[0]), int(t[1])
                break
        m = np.zeros((nr, nc), dtype=np.int64)
        i = 0
        while i < nr:
            s = lines[pos].strip()
            pos += 1
            if not s or s.startswith("#"):
                continue
            m[i] = [int(x) for x in s.split()[:nc]]
            i += 1
        return m

    context = matrix()
    while not lines[pos].strip():
        pos += 1
    bignum = int(lines[pos].split()[0])
    pos += 1
    domain = matrix()
    opts = {}
    for s in lines[pos:]:
        low = s.lower()
        if low.startswith("maximize"):
            opts["Maximize"] = 1
        if low.startswith("urs_parms"):
            opts["Urs_parms"] = 1
        if low.startswith("urs_unknowns"):
            opts["Urs_unknowns"] = 1
        if low.startswith("rational"):
            opts["Nq"] = 0
        if low.startswith("dual"):
            opts["Compute_dual"] = 1
    return context, bignum, domain, opts


def matrix_text(m):
    """pip_matrix_print (piplib.c:176-190)"""
    out = [f"{m.shape[0]} {m.shape[1]}\n"]
    for r in m:
        out.append("".join(f" {int(v)}" for v in r) + "\n")
    return "".join(out)
