"""Test helper: the committed outputs of the reference's own arbitrary-precision build (tests/golden/gmp/*.json,
made by tests/golden/make_gmp_fixtures.py from oracle/_ref/refpip_gmp = the reference's five sources compiled with
-DPIPLIB_INT_GMP).  TEST INFRASTRUCTURE ONLY."""
import json
import os
import sys

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gmp_fixture(name):
    """(problems, batch flags, records, sha) of one family; the inputs are regenerated from the seed, the records
    carry status / pivots / sha256 of the squashed sol_edit text / widths (see make_gmp_fixtures.py)."""
    sys.path.insert(0, G)
    import make_gmp_fixtures as mg
    probs, flags = mg.problems_of(name)
    recs = json.load(open(os.path.join(G, "gmp", name + ".json")))["problems"]
    assert len(recs) == len(probs)
    return probs, flags, recs, mg.sha
