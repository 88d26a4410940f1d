"""Per-kernel statistics from a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace`):
name, calls, total / average / min / max duration, share of the total -- the table `--stats` prints, as CSV.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [out.csv] [--from KERNEL_SUBSTRING ...]

`--from a b`: only dispatches that start at or after the first dispatch of a kernel whose name contains `a` (else `b`, ...): cuts a
traced training command's set-up (parameter initialisation, synthetic-set generation) off the step statistics.
"""
import csv
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name)            # drop the argument list of demangled names
    return name.replace("void ", "").strip()


def stats(db_path, start_at=()):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    rows = cur.execute(
        "select s.kernel_name, d.end - d.start, d.start from rocpd_kernel_dispatch d "
        "join rocpd_info_kernel_symbol s on d.kernel_id = s.id").fetchall()
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
        a = agg.setdefault(short(name), [0, 0, 1 << 62, 0])
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
    out, total = stats(argv[0], start_at)
    w = csv.writer(open(argv[1], "w", newline="") if len(argv) > 1 else sys.stdout)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
    for r in out:
        w.writerow([r[0], r[1], r[2], f"{r[3]:.1f}", r[4], r[5], f"{r[6]:.2f}"])
    print(f"# total kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in out)} dispatches", file=sys.stderr)


if __name__ == "__main__":
    main()
