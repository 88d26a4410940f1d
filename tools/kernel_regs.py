"""Register / occupancy table of every kernel in one .hip file: `python tools/kernel_regs.py afm_gemm_mfma_f16.hip [filter]`.
Compiles the file for gfx950 with -Rpass-analysis=kernel-resource-usage (no GPU needed) and prints one line per kernel."""
import os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "multimodalanalytical_amd", "csrc")


def table(src, extra=()):
    path = src if os.path.exists(src) else os.path.join(CSRC, src)
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
           "-I" + CSRC, "-Rpass-analysis=kernel-resource-usage", "-c", path, "-o", "/dev/null", *extra]
    cmd += os.environ.get("AFM_EXTRA_FLAGS", "").split()
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr)
        raise SystemExit(r.returncode)
    rows, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+(?:\[[^\]]*\])?): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return rows


def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip() or n
    except OSError:
        return n


if __name__ == "__main__":
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    print(f"{'VGPR':>5s} {'AGPR':>5s} {'spill':>5s} {'SGPR':>5s} {'sspill':>6s} {'scratch':>7s} {'occ':>3s}  kernel")
    for r in table(sys.argv[1]):
        name = demangle(r["name"])
        if flt and flt not in name:
            continue
        name = re.sub(r"\(.*\)$", "", name)
        print(f"{r.get('VGPRs', -1):5d} {r.get('AGPRs', -1):5d} {r.get('VGPRs Spill', -1):5d} {r.get('TotalSGPRs', -1):5d} "
              f"{r.get('SGPRs Spill', -1):6d} {r.get('ScratchSize [bytes/lane]', -1):7d} {r.get('Occupancy [waves/SIMD]', -1):3d}  {name}")
