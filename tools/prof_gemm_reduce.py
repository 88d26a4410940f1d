"""Merge the PMC passes of tools/prof_gemm_pmc.sh: per configuration (the i-th run of R consecutive k_gemm dispatches of each pass),
means per launch (the first 2 launches of a run are left out: cold clocks / caches) and the derived figures DESIGN.md quotes."""
import json, os, sqlite3, subprocess, sys

here = os.path.dirname(os.path.abspath(__file__))


def passes(path, reps):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select d.id, s.kernel_name, d.start, d.end - d.start from rocpd_kernel_dispatch d "
        "join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
    ev = {}
    for did, cname, val in db.execute(
            "select d.id, p.name, e.value from rocpd_kernel_dispatch d join rocpd_pmc_event e on e.event_id = d.event_id "
            "join rocpd_info_pmc p on e.pmc_id = p.id"):
        ev.setdefault(did, {})
        ev[did][cname] = ev[did].get(cname, 0.0) + float(val)
    gemm = [(did, name, dur) for did, name, st, dur in rows if "k_gemm" in name]
    out = []
    for i in range(0, len(gemm) - reps + 1, reps):
        run = gemm[i:i + reps][2:]
        agg = {"kernel": run[0][1].split("(")[0][:110], "avg_ns": sum(r[2] for r in run) / len(run)}
        for did, _, _ in run:
            for c, v in ev.get(did, {}).items():
                agg[c] = agg.get(c, 0.0) + v / len(run)
        out.append(agg)
    return out


def main():
    O, tag, reps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    which = sys.argv[4] if len(sys.argv) > 4 else "c2"
    names = subprocess.run([sys.executable, os.path.join(here, "prof_gemm_run.py"), "--list", "--set", which], capture_output=True, text=True).stdout.split()
    merged = {n: {} for n in names}
    for i in range(1, 6):
        p = os.path.join(O, f"{tag}_gemm_pass{i}.db")
        if not os.path.exists(p):
            continue
        for n, agg in zip(names, passes(p, reps)):
            t = merged[n].setdefault("avg_ns_by_pass", [])
            t.append(round(agg.pop("avg_ns")))
            merged[n].update(agg)
    for n, m in merged.items():
        if "GRBM_GUI_ACTIVE" in m and m.get("avg_ns_by_pass"):
            cyc = m["GRBM_GUI_ACTIVE"] / 8.0
            m["clock_ghz"] = round(cyc / m["avg_ns_by_pass"][1], 3) if len(m["avg_ns_by_pass"]) > 1 else None
            if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
                m["mfma_busy_frac"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc, 4)
        if "SQ_INSTS_MFMA" in m and m.get("SQ_INSTS_MFMA"):
            m["valu_per_mfma"] = round(m.get("SQ_INSTS_VALU", 0) / m["SQ_INSTS_MFMA"], 3)
        if "SQ_WAVE_CYCLES" in m:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU"):
                if c in m:
                    m[c.lower() + "_frac"] = round(m[c] / m["SQ_WAVE_CYCLES"], 4)
        if "FETCH_SIZE" in m or "WRITE_SIZE" in m:
            # KB counters; gfx950 tallies 128-byte read requests at 64 bytes (MI355X_MICROARCH.md, HBM): FETCH doubled
            m["hbm_read_bytes"] = round(m.get("FETCH_SIZE", 0) * 1024 * 2)
            m["hbm_write_bytes"] = round(m.get("WRITE_SIZE", 0) * 1024)
            m["hbm_bytes_per_launch"] = m["hbm_read_bytes"] + m["hbm_write_bytes"]
        if "TCC_HIT_sum" in m:
            m["l2_hit_rate"] = round(m["TCC_HIT_sum"] / max(1.0, m["TCC_HIT_sum"] + m.get("TCC_MISS_sum", 0)), 4)
    print(json.dumps(merged, indent=1))


if __name__ == "__main__":
    main()
