"""Scan the device ISA of the library's kernels for requests that are waited for one by one: a `global_load*` / `buffer_load*` followed
within four instructions by `s_waitcnt vmcnt(0)` and no other request in between.  A few per kernel are legitimate (a poll loop, a
dependent index); a row of them is the pattern "zero default + conditional load": the compiler joins the two with a copy behind a wait
of its own, so N such loads are N dependent memory round trips.  No GPU needed.
python tools/isa_wait_scan.py [kernels_lm sampler engine codec] [--min N] [--show KERNEL_SUBSTRING]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-value", "-S", "--cuda-device-only"]


def asm_of(name):
    out = os.path.join(tempfile.gettempdir(), f"voxscan_{name}.s")
    src = os.path.join(ROOT, "vox_serve_amd", "csrc", name + ".hip")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, src, "-o", out], check=True, stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def kernels(lines):
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^(_Z\w+|k_\w+):\s", l)]
    starts.append((len(lines), None))
    for (a, name), (b, _) in zip(starts, starts[1:]):
        yield name, lines[a:b]


def is_load(l):
    return "global_load" in l or "buffer_load" in l


def scan(body):
    hits = []
    for j, l in enumerate(body):
        if not is_load(l):
            continue
        for k in range(1, 5):
            if j + k >= len(body) or is_load(body[j + k]):
                break
            if "s_waitcnt" in body[j + k] and "vmcnt(0)" in body[j + k]:
                hits.append(j)
                break
    return hits, sum(1 for l in body if is_load(l))


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    mn = int(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 3
    show = sys.argv[sys.argv.index("--show") + 1] if "--show" in sys.argv else None
    args = [a for a in args if a != str(mn) and a != show]
    for f in args or ["kernels_lm", "sampler", "engine", "codec"]:
        rows = []
        for name, body in kernels(asm_of(f)):
            hits, nl = scan(body)
            if show and show in name:
                print(f"== {name}")
                for i, l in enumerate(body):
                    if is_load(l) or "vmcnt" in l or "s_barrier" in l or "global_store" in l:
                        print(f"{i:6d} {l.strip()[:110]}")
            if len(hits) >= mn:
                rows.append((len(hits), nl, name))
        for n, nl, name in sorted(rows, reverse=True):
            print(f"{f:12s} {n:3d} of {nl:3d} requests waited for alone   {name[:100]}")
