"""Build the HIP extension in-tree: piplib_amd/libpipamd.so (gfx950 only).

    python -m piplib_amd.build            # build if sources are newer
    python -m piplib_amd.build --force
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libpipamd.so")
SOURCES = ["pip_kernels.hip", "pip_quast.hip", "pip_host.cpp", "pip_tree.cpp"]
HEADERS = ["pip_job.h", "pip_host.h", "pip_quast.h", os.path.join("..", "..", "include", "piplib_amd.h")]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=True, profile=False):
    """profile=True builds the diagnostic variant libpipamd_prof.so (-DPIP_PROFILE: per-phase
    cycle stamps in the kernel; never used for timing or shipped results)."""
    if profile == "events":
        return _compile(os.path.join(HERE, "libpipamd_prof_events.so"), ["-DPIP_PROFILE", "-DPIP_PROFILE_EVENTS"], verbose)
    if profile:
        return _compile(os.path.join(HERE, "libpipamd_prof.so"), ["-DPIP_PROFILE"], verbose)
    if os.environ.get("PIP_MINWAVES"):  # tuning experiments only
        return _compile(os.path.join(HERE, "libpipamd_mw%s.so" % os.environ["PIP_MINWAVES"]),
                        ["-DPIP_MINWAVES=" + os.environ["PIP_MINWAVES"]], verbose)
    if os.environ.get("PIP_VARIANT"):  # tuning experiments: PIP_VARIANT=name PIP_DEFS="-DPIP_PF=2 ..."
        return _compile(os.path.join(HERE, "libpipamd_%s.so" % os.environ["PIP_VARIANT"]),
                        os.environ.get("PIP_DEFS", "").split(), verbose)
    if os.environ.get("PIP_DUP"):  # instruction-count experiments only (tools/pmc_dup.sh)
        return _compile(os.path.join(HERE, "libpipamd_dup%s.so" % os.environ["PIP_DUP"]),
                        ["-DPIP_DUP=" + os.environ["PIP_DUP"]], verbose)
    if not force and not needs_build():
        return OUT
    return _compile(OUT, [], verbose)


def _compile(out, extra, verbose):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fgpu-rdc" if False else "-DNDEBUG",
           "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-x", "hip"] + extra
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv,
          profile="events" if "--profile-events" in sys.argv else ("--profile" in sys.argv))
