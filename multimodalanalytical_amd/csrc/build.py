"""Build libafm_hip.so (gfx950) in-tree with hipcc.  `python -m multimodalanalytical_amd.csrc.build`.

The library is the product's compute path; there is no CPU fallback.  hipcc cross-compiles
without a GPU, so this runs in the build container and the .so travels to the GPU box.
"""
import concurrent.futures as cf
import fcntl
import hashlib
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
# AFM_BUILD_VARIANT=name: an experiment build (usually with AFM_EXTRA_FLAGS) beside the product library, with its own object
# directory; load it with AFM_LIB_OVERRIDE=<printed path>.  Never set in product runs.
_VARIANT = os.environ.get("AFM_BUILD_VARIANT", "")
LIB = os.path.join(PKG, "libafm_hip.so") if not _VARIANT else os.path.join(ROOT, "tools", "experiments", "_abl", f"libafm_{_VARIANT}.so")
OBJ = os.path.join(HERE, "build" if not _VARIANT else f"build/{_VARIANT}")
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


_INC = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)


def _closure(path, seen=None):
    """`path` and every header it includes with quotes, transitively (csrc/ and include/ only)."""
    seen = [] if seen is None else seen
    if path in seen:
        return seen
    seen.append(path)
    with open(path, "r") as fh:
        text = fh.read()
    for name in _INC.findall(text):
        for d in (HERE, os.path.join(ROOT, "include")):
            cand = os.path.join(d, name)
            if os.path.exists(cand):
                _closure(cand, seen)
                break
    return seen


def _digest(path, extra=b""):
    """sha1 of a source, the headers it (transitively) includes and the flags: an object is stale exactly when one of
    them changed (the single-pass MFMA kernels live in *_impl.h files that two .hip files instantiate)."""
    h = hashlib.sha1(extra)
    for f in _closure(path):
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build(force=False, verbose=False):
    """Compile what is stale and link.  Safe under `torch.distributed.run`: an exclusive lock file serialises the ranks (the
    first one builds, the others find everything fresh), the library is linked to a temporary name and renamed into place,
    and a link stamp (digest of all object digests) forces a relink when an earlier run died between compile and link."""
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    with open(os.path.join(OBJ, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose):
    hipcc = _hipcc()
    jobs, objs, digs = [], [], []
    for src in _sources():
        sp = os.path.join(HERE, src)
        op = os.path.join(OBJ, src[:-4] + ".o")
        stamp = op + ".sha1"
        # flags enter the digest with the checkout path normalised: the same tree at another path (the GPU box)
        # must not look stale
        dig = _digest(sp, " ".join(FLAGS).replace(ROOT, "$ROOT").encode())
        objs.append(op)
        digs.append(dig)
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
        with cf.ThreadPoolExecutor(max_workers=min(int(os.environ.get("AFM_BUILD_JOBS", "6")), len(jobs))) as ex:
            for done in ex.map(run, jobs):
                if verbose:
                    print("compiled", os.path.basename(done))
    link_dig = hashlib.sha1("\n".join(digs).encode()).hexdigest()
    link_stamp = os.path.join(OBJ, "libafm_hip.so.sha1")
    stale = not os.path.exists(LIB) or not os.path.exists(link_stamp) or open(link_stamp).read() != link_dig
    if jobs or stale:
        tmp = LIB + f".tmp{os.getpid()}"
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", tmp] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        os.replace(tmp, LIB)
        with open(link_stamp, "w") as fh:
            fh.write(link_dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
