"""Build libmcamd.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m modelcompression_amd.build        # or __graft_entry__.build()

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with the
repo snapshot to the GPU box.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libmcamd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
COMMON = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-variable",
          "-fhip-fp32-correctly-rounded-divide-sqrt"]
SOURCES = {
    "api.hip": [],
    "conv_igemm.hip": [],
    "conv_igemm_pp.hip": [],
    "conv_stem.hip": [],
    "conv_stem_block.hip": [],
    "conv_stem_f32.hip": [],
    "conv_small.hip": [],
    "conv_win.hip": [],
    "conv_wres.hip": [],
    "conv_wgrad.hip": [],
    "conv_wgrad_stem.hip": [],
    "bn_act.hip": [],
    "fold.hip": [],
    "plan.hip": [],
    "region_loss.hip": [],
    "prune.hip": ["-ffp-contract=off"],   # pinned fp32 arithmetic: no FMA contraction
}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "mcamd.h"))
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for src, extra in SOURCES.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            jobs.append([HIPCC] + COMMON + extra + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        for err in ex.map(run, jobs):
            if verbose and err.strip():
                print(err)
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(OUT, objs):
        run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", OUT] + objs)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
