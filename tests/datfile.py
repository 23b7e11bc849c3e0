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
