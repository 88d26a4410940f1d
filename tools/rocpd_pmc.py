"""Per-kernel averages of the hardware counters in a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --pmc ...`).

    python tools/rocpd_pmc.py out_results.db [name-substring]      -> JSON {kernel: {counter: mean per dispatch, "dispatches": n, "avg_ns": t}}
"""
import json
import re
import sqlite3
import sys


def short(name):
    return re.sub(r"\(.*$", "", name).replace("void ", "").strip()


def collect(path, want=""):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute(
        "select s.kernel_name, d.id, d.end - d.start, p.name, e.value from rocpd_kernel_dispatch d "
        "join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
        "left join rocpd_pmc_event e on e.event_id = d.event_id "
        "left join rocpd_info_pmc p on e.pmc_id = p.id").fetchall()
    out = {}
    seen = {}
    for kname, did, dur, cname, val in rows:
        k = short(kname)
        if want and want not in k:
            continue
        o = out.setdefault(k, {"_n": 0, "_t": 0})
        if (k, did) not in seen:
            seen[(k, did)] = 1
            o["_n"] += 1
            o["_t"] += dur
        if cname is not None:
            o[cname] = o.get(cname, 0.0) + float(val)
    res = {}
    for k, o in out.items():
        n = max(1, o["_n"])
        res[k] = {"dispatches": o["_n"], "avg_ns": o["_t"] / n}
        for c, v in o.items():
            if not c.startswith("_"):
                res[k][c] = v / n
    return res


if __name__ == "__main__":
    print(json.dumps(collect(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""), indent=1))
