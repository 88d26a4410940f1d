"""Build libafm_hip.so (gfx950) in-tree with hipcc.  `python -m multimodalanalytical_amd.csrc.build`.

The library is the product's compute path; there is no CPU fallback.  hipcc cross-compiles
without a GPU, so this runs in the build container and the .so travels to the GPU box.
"""
import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
LIB = os.path.join(PKG, "libafm_hip.so")
OBJ = os.path.join(HERE, "build")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-I" + os.path.join(ROOT, "include"),
         "-I" + HERE, "-Wno-unused-result", "-Wno-unused-value"]
FLAGS += os.environ.get("AFM_EXTRA_FLAGS", "").split()   # e.g. -DAFM_GEMM_ABLATIONS for tools/bench_gemm.py --ablate


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libafm_hip.so cannot be built")


def have_hipcc() -> bool:
    try:
        _hipcc()
        return True
    except RuntimeError:
        return False


def _sources():
    return sorted(f for f in os.listdir(HERE) if f.endswith(".hip"))


def _digest(path, extra=b""):
    h = hashlib.sha1(extra)
    for f in [path] + [os.path.join(HERE, x) for x in sorted(os.listdir(HERE)) if x.endswith(".h")] + \
            [os.path.join(ROOT, "include", "afm_hip.h")]:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    jobs, objs = [], []
    for src in _sources():
        sp = os.path.join(HERE, src)
        op = os.path.join(OBJ, src[:-4] + ".o")
        stamp = op + ".sha1"
        # flags enter the digest with the checkout path normalised: the same tree at another path (the GPU box)
        # must not look stale
        dig = _digest(sp, " ".join(FLAGS).replace(ROOT, "$ROOT").encode())
        objs.append(op)
        if not force and os.path.exists(op) and os.path.exists(stamp) and open(stamp).read() == dig:
            continue
        jobs.append((sp, op, stamp, dig))

    def run(job):
        sp, op, stamp, dig = job
        cmd = [hipcc] + FLAGS + ["-c", sp, "-o", op]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {sp}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        with open(stamp, "w") as fh:
            fh.write(dig)
        return sp

    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for done in ex.map(run, jobs):
                if verbose:
                    print("compiled", os.path.basename(done))
    if jobs or not os.path.exists(LIB):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
