"""Per-kernel statistics from a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace`):
name, calls, total / average / min / max duration, share of the total -- the table `--stats` prints, as CSV.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [out.csv]
"""
import csv
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name)            # drop the argument list of demangled names
    return name.replace("void ", "").strip()


def stats(db_path):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    rows = cur.execute(
        "select s.kernel_name, d.end - d.start from rocpd_kernel_dispatch d "
        "join rocpd_info_kernel_symbol s on d.kernel_id = s.id").fetchall()
    agg = {}
    for name, dur in rows:
        a = agg.setdefault(short(name), [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
    total = sum(a[1] for a in agg.values())
    out = [(n, a[0], a[1], a[1] / a[0], a[2], a[3], 100.0 * a[1] / total) for n, a in agg.items()]
    out.sort(key=lambda r: -r[2])
    return out, total


def main():
    out, total = stats(sys.argv[1])
    w = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
    for r in out:
        w.writerow([r[0], r[1], r[2], f"{r[3]:.1f}", r[4], r[5], f"{r[6]:.2f}"])
    print(f"# total kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in out)} dispatches", file=sys.stderr)


if __name__ == "__main__":
    main()
