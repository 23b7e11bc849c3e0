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
SOURCES = ["pip_adv_d.hip", "pip_adv_c.hip", "pip_adv_b.hip", "pip_adv_a.hip", "pip_adv_e.hip", "pip_adv_f.hip", "pip_kernels.hip", "pip_quast.hip", "pip_host.cpp",
           "pip_tree.cpp"]
HEADERS = ["pip_job.h", "pip_host.h", "pip_quast.h", "pip_advance.h", "pip_lean.h", "pip_lean64.h", "pip_adv_inst.h", os.path.join("..", "..", "include", "piplib_amd.h")]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=True, profile=False):
    """profile=True builds the diagnostic variant libpipamd_prof.so (-DPIP_PROFILE: per-phase
    cycle stamps in the kernel; never used for timing or shipped results)."""
    # (the cycle stamps take 32 more VGPRs: without -DPIP_MINWAVES=1 the one-wave kernels' 80-register bound would push
    # them into scratch and the stamps would measure the spills)
    if profile == "events":
        return _compile(os.path.join(HERE, "libpipamd_prof_events.so"), ["-DPIP_PROFILE", "-DPIP_PROFILE_EVENTS", "-DPIP_MINWAVES=1"], verbose)
    if profile:
        return _compile(os.path.join(HERE, "libpipamd_prof.so"), ["-DPIP_PROFILE", "-DPIP_MINWAVES=1", "-DPIP_LEAN_WAVES=4"], verbose)
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
    """every source to its own object (side by side: the four take 20-50 s each), then one link; objects are kept under
    piplib_amd/build/<variant>/ and reused while their source and the headers are older"""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DNDEBUG", "-Wall", "-Wno-unused-function",
             "-Wno-unused-value"] + extra
    tag = os.path.splitext(os.path.basename(out))[0] + ("_" + "_".join(x.strip("-").replace("=", "") for x in extra) if extra else "")
    odir = os.path.join(HERE, "build", tag)
    os.makedirs(odir, exist_ok=True)
    hdr_t = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    jobs, objs = [], []
    for src in SOURCES:
        sp, obj = os.path.join(CSRC, src), os.path.join(odir, src + ".o")
        objs.append(obj)
        if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(sp), hdr_t):
            continue
        cmd = [hipcc] + flags + ["-x", "hip", "-c", sp, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in jobs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv,
          profile="events" if "--profile-events" in sys.argv else ("--profile" in sys.argv))
