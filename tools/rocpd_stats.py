"""Per-kernel statistics from a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace`):
name, calls, total / average / min / max duration, share of the total -- the table `--stats` prints, as CSV.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [out.csv] [--from KERNEL_SUBSTRING ...]

`--from a b`: only dispatches that start at or after the first dispatch of a kernel whose name contains `a` (else `b`, ...): cuts a
traced training command's set-up (parameter initialisation, synthetic-set generation) off the step statistics.
`--by-grid`: one row per (kernel, grid size, workgroup size, LDS bytes) instead of per kernel name: the encoder- and decoder-shape
launches of one kernel (attention, grouped weight gradients, LayerNorm) get their own averages.
`--cluster [ratio]`: additionally split a row whose durations fall into separate bands (sorted neighbours more than `ratio` apart, default
1.3) -- the persistent GEMM kernels launch 256 workgroups whatever the shape, so only their durations tell the shapes apart.
"""
import csv
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name)            # drop the argument list of demangled names
    return name.replace("void ", "").strip()


def _clusters(durs, ratio):
    """Sorted durations cut where a neighbour is more than `ratio` times the one before it."""
    ds = sorted(durs)
    out, cur = [], [ds[0]]
    for v in ds[1:]:
        if v > cur[-1] * ratio:
            out.append(cur); cur = []
        cur.append(v)
    out.append(cur)
    return out


def stats(db_path, start_at=(), by_grid=False, cluster=0.0):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    extra = ""
    if by_grid:
        cols = {r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)").fetchall()}
        want = [c for c in ("grid_size_x", "grid_size_y", "grid_size_z", "workgroup_size_x", "group_segment_size", "lds_block_size") if c in cols]
        extra = "".join(f", d.{c}" for c in want)
    rows = cur.execute(
        f"select s.kernel_name, d.end - d.start, d.start{extra} from rocpd_kernel_dispatch d "
        "join rocpd_info_kernel_symbol s on d.kernel_id = s.id").fetchall()
    if by_grid:
        def label(r):
            dims = dict(zip(want, r[3:]))
            g = dims.get("grid_size_x", 0) * max(1, dims.get("grid_size_y", 1) or 1) * max(1, dims.get("grid_size_z", 1) or 1)
            wgs = dims.get("workgroup_size_x", 0) or 1
            lds = dims.get("group_segment_size", dims.get("lds_block_size", 0))
            return f"{short(r[0])} [grid {g // wgs if g % wgs == 0 else g} x {wgs}, lds {lds}]"
        rows = [(label(r), r[1], r[2]) for r in rows]
        if cluster:
            per = {}
            for name, dur, st in rows:
                per.setdefault(name, []).append(dur)
            bands = {}
            for name, durs in per.items():
                cl = _clusters(durs, cluster)
                if len(cl) > 1:
                    bands[name] = [c[-1] for c in cl]          # upper edge of every band
            def banded(name, dur):
                if name not in bands:
                    return name
                k = next(i for i, hi in enumerate(bands[name]) if dur <= hi)
                return f"{name} #band{k}"
            rows = [(banded(n, d), d, st) for n, d, st in rows]
    first = 0
    for key in start_at:
        hits = [st for name, _, st in rows if key in name]
        if hits:
            first = min(hits)
            break
    agg = {}
    for name, dur, st in rows:
        if st < first:
            continue
        a = agg.setdefault(name if by_grid else short(name), [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
    total = sum(a[1] for a in agg.values())
    out = [(n, a[0], a[1], a[1] / a[0], a[2], a[3], 100.0 * a[1] / total) for n, a in agg.items()]
    out.sort(key=lambda r: -r[2])
    return out, total


def main():
    argv = sys.argv[1:]
    start_at = ()
    if "--from" in argv:
        i = argv.index("--from")
        start_at, argv = tuple(argv[i + 1:]), argv[:i]
    by_grid = "--by-grid" in argv
    cluster = 0.0
    if "--cluster" in argv:
        i = argv.index("--cluster")
        nxt = argv[i + 1] if i + 1 < len(argv) else ""
        try:
            cluster = float(nxt); del argv[i + 1]
        except ValueError:
            cluster = 1.3
        argv.remove("--cluster")
    if by_grid:
        argv.remove("--by-grid")
    out, total = stats(argv[0], start_at, by_grid, cluster)
    w = csv.writer(open(argv[1], "w", newline="") if len(argv) > 1 else sys.stdout)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
    for r in out:
        w.writerow([r[0], r[1], r[2], f"{r[3]:.1f}", r[4], r[5], f"{r[6]:.2f}"])
    print(f"# total kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in out)} dispatches", file=sys.stderr)


if __name__ == "__main__":
    main()
