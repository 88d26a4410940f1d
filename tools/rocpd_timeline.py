"""Dispatch timeline of a rocprofv3 rocpd database: for the last N dispatches (one micro-batch, say) the kernel, its duration and
the idle gap since the previous kernel ended; then the totals per kernel of (duration, gap in front).

    python tools/rocpd_timeline.py results.db [N] [--list]
"""
import collections
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name).replace("void ", "").strip()
    m = re.match(r"_ZN7afm_(?:f16|bf16)\d+([a-z_0-9]+?)I", name) or re.match(r"_Z\d+([a-z_0-9A-Z]+?)(?:I|E|P)", name)
    return (m.group(1) if m else name)[:40]


def main():
    db = sqlite3.connect(sys.argv[1])
    n = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 400
    rows = db.execute("select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                      "on d.kernel_id = s.id order by d.start").fetchall()
    rows = rows[-n:]
    agg = collections.defaultdict(lambda: [0, 0, 0])
    prev_end, busy, idle = None, 0, 0
    for name, st, en in rows:
        gap = 0 if prev_end is None else max(0, st - prev_end)
        k = short(name)
        a = agg[k]; a[0] += 1; a[1] += en - st; a[2] += gap
        busy += en - st; idle += gap
        if "--list" in sys.argv:
            print(f"{k:40s} {(en - st) / 1e3:9.1f} us   gap {gap / 1e3:7.1f} us")
        prev_end = en if prev_end is None else max(prev_end, en)
    print(f"# last {len(rows)} dispatches: {busy / 1e6:.3f} ms in kernels, {idle / 1e6:.3f} ms idle between them "
          f"({(rows[-1][2] - rows[0][1]) / 1e6:.3f} ms wall)")
    for k, a in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:40]:
        print(f"{k:40s} x{a[0]:4d}  {a[1] / 1e3:9.1f} us  + gaps {a[2] / 1e3:8.1f} us  ({a[2] / a[0] / 1e3:5.1f} us each)")


if __name__ == "__main__":
    main()
