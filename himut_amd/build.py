"""In-tree native builds: HIP kernels + C-ABI, host ingest, synthetic generator.

Everything is compiled with explicit compiler invocations (no JIT cache), so
the resulting ``.so`` files sit next to the sources and travel with the repo
snapshot to the GPU box.  ``build_all`` is what ``__graft_entry__.build`` calls.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")

HIP_LIB = os.path.join(HERE, "libhimut_hip.so")
HOST_LIB = os.path.join(HERE, "libhimut_host.so")
SYNTH_LIB = os.path.join(HERE, "libhimut_synth.so")


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _run(cmd):
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        sys.stderr.write(proc.stdout)
        raise RuntimeError("build failed: " + " ".join(cmd))
    return proc.stdout


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


def build_hip(force=False, verbose=False):
    """Compile the gfx950 kernels and the C-ABI into libhimut_hip.so."""
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps.append(os.path.join(INCLUDE, "himut_hip.h"))
    if not force and _newer(HIP_LIB, deps):
        return HIP_LIB
    hipcc = hipcc_path()
    if hipcc is None:
        raise RuntimeError("hipcc not found; cannot build libhimut_hip.so")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-result",
           "-I", INCLUDE, "-I", CSRC, "-o", HIP_LIB] + srcs
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    out = _run(cmd)
    if verbose:
        print(out)
    return HIP_LIB


def build_host(force=False):
    """Compile the host-side ingest library (BGZF/BAM reader + writer)."""
    srcs = [os.path.join(CSRC, "bam_ingest.cpp")]
    if not os.path.exists(srcs[0]):
        return None
    if not force and _newer(HOST_LIB, srcs + [os.path.join(INCLUDE, "himut_hip.h")]):
        return HOST_LIB
    _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-I", INCLUDE, "-I", CSRC,
          "-o", HOST_LIB] + srcs + ["-lz", "-ldl"])
    return HOST_LIB


def build_synth(force=False):
    src = os.path.join(CSRC, "synth.cpp")
    if not force and _newer(SYNTH_LIB, [src]):
        return SYNTH_LIB
    _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", SYNTH_LIB, src])
    return SYNTH_LIB


def build_all(force=False):
    build_synth(force)
    build_host(force)
    build_hip(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
    print("built:", HIP_LIB, HOST_LIB, SYNTH_LIB)
