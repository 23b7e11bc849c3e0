"""-m gpu: the HIP path (through the C ABI in libpipamd.so) against the CPU oracle,
bit-exact: status, pivot count and every numerator/denominator of the solution."""
import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]


@pytest.mark.parametrize("seed,batch,nvar,ni,nq,kw", [
    (11, 64, 6, 8, 1, dict(nnz=3, cmax=3, x0max=5)),
    (12, 64, 6, 8, 0, dict(nnz=3, cmax=3, x0max=5)),
    (13, 64, 20, 16, 1, dict()),
    (14, 48, 63, 32, 0, dict()),      # BASELINE configs[1] shape: 32x64, rational
    (15, 48, 63, 32, 1, dict()),
    (16, 32, 127, 64, 1, dict()),     # BASELINE configs[2] shape: 64x128, integer + cuts
    (17, 16, 200, 90, 1, dict()),     # two 128-column chunks per row
])
def test_lexmin_batch_vs_oracle(seed, batch, nvar, ni, nq, kw):
    from gpu_common import compare
    from piplib_amd import synth
    rows = synth.lexmin_batch(seed, batch, nvar, ni, **kw)
    n, piv = compare(rows, nvar, 0, nq)
    assert n == batch and piv > 0
