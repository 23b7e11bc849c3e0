#!/usr/bin/env python3
"""Regenerate tests/golden/ref_dp/ from the reference itself (int64 "dp" build).

Runs oracle/_ref/refpip (the reference library compiled by oracle/Makefile from
/root/reference/source, driven by oracle/ref_driver.c) on the inputs of the
reference's test-suite that have no usable .ll of their own:
  * test/{boulet,bouleti,dirk}.dat  (listed as "pbs with" in test/Makefile.am:17-21;
    their .ll files were produced by another integer flavour)
  * test/challenges/*               (no .ll shipped)
  * example/*.dat                   (only the .pip variants have a .ll)
and stores stdout plus (exit code, first stderr line) in manifest.json.
Only needed when the fixtures change; needs /root/reference (not on the GPU box).
"""
import json, os, subprocess, sys

HERE = os.path.dirname(os.path.abspath(__file__))
REFPIP = os.path.join(HERE, "..", "..", "oracle", "_ref", "refpip")

INPUTS = (
    [("test", n + ".dat") for n in ("boulet", "bouleti", "dirk")]
    + [("challenges", n) for n in ("pipFile_0", "system_sysmo_for_pipMP", "vivien32.dat")]
    + [("example", n + ".dat") for n in
       ("big", "cg1", "esced", "ex", "ex2", "fimmel", "max", "small", "square", "square_max", "sven")]
)

def main():
    out_dir = os.path.join(HERE, "ref_dp")
    os.makedirs(out_dir, exist_ok=True)
    manifest = {}
    for sub, name in INPUTS:
        src = os.path.join(HERE, sub, name)
        p = subprocess.run([REFPIP, "dat", src], capture_output=True, timeout=60)
        key = f"{sub}/{name}"
        out_name = key.replace("/", "__") + ".ll"
        with open(os.path.join(out_dir, out_name), "wb") as f:
            f.write(p.stdout)
        err = p.stderr.decode(errors="replace").strip().splitlines()
        manifest[key] = {"rc": p.returncode, "stderr": err[0] if err else "", "ll": out_name}
        print(key, manifest[key])
    # Compute_dual (rational solve + dual variables): non-parametric examples only -- with
    # parameters the reference reads uninitialised memory in tab_sort_rows (traiter.c:576-589
    # skips unit rows before ineq[i] is set) and crashes or prints garbage.
    for name in ("small", "square", "max", "big", "cg1", "sven"):
        src = open(os.path.join(HERE, "example", name + ".pip"), "rb").read() + b"\nRational\nDual\n"
        with open(os.path.join(out_dir, f"dual__{name}.pip"), "wb") as f:
            f.write(src)
        p = subprocess.run([REFPIP, "pip"], input=src, capture_output=True, timeout=60)
        assert p.returncode == 0, name
        with open(os.path.join(out_dir, f"dual__{name}.ll"), "wb") as f:
            f.write(p.stdout)
        print("dual", name, len(p.stdout))
    with open(os.path.join(out_dir, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)

if __name__ == "__main__":
    sys.exit(main())
