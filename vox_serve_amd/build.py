"""Build libvoxhip.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvoxhip.so")
SOURCES = ["kernels_lm.hip", "sampler.hip", "engine.hip", "codec.hip"]
# -ffp-contract=off: the numeric contract spells every fused multiply-add as an explicit fmaf
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-value"] + \
    os.environ.get("VOX_EXTRA_FLAGS", "").split()


def _hipcc():
    for c in ("/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.isabs(c) and os.path.exists(c):
            return c
    return "hipcc"


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "voxhip.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s in srcs:
        o = os.path.join(HERE, "build", os.path.basename(s) + ".o")
        objs.append(o)
        procs.append((s, subprocess.Popen([_hipcc(), *FLAGS, "-c", s, "-o", o], stdout=subprocess.PIPE,
                                          stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f"hipcc failed on {s}")
        if verbose and out:
            sys.stderr.write(out.decode())
    subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
