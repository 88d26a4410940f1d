"""Reduce the passes of tools/prof_step_pmc.sh: counters summed over EVERY kernel dispatch of the traced command and divided by the
number of optimiser steps it ran; set-up (parameter initialisation, the synthetic set's generation) is cut off at the first dispatch of
the device input path.
  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)   (share of the kernels' cycles the matrix pipes are busy)
  hbm bytes      = FETCH_SIZE KB x 1024 x 2 (gfx950 tallies 128-byte read requests at 64 bytes) + WRITE_SIZE KB x 1024
  hbm_gbs        = bytes / kernel time of the same pass."""
import json, os, re, sqlite3, sys


def load(path):
    db = sqlite3.connect(path)
    rows = db.execute("select d.id, s.kernel_name, d.end - d.start, d.start from rocpd_kernel_dispatch d "
                      "join rocpd_info_kernel_symbol s on d.kernel_id = s.id").fetchall()
    # the optimiser steps begin with the first collated micro-batch: everything before the first dispatch of the device input path
    # (afm_patch_preprocess: k_patch_values / k_patch_mask; k_gather_rows where a workload has no patched modality) is set-up -- parameter
    # initialisation, the synthetic set's generation: ~200 elementwise launches -- and is left out of the step's counters
    first = min((st for _, name, _, st in rows if "k_patch_" in name), default=None)
    if first is None:
        first = min((st for _, name, _, st in rows if "k_gather_rows" in name), default=0)
    k = {did: (name, dur) for did, name, dur, st in rows if st >= first}
    ev = {}
    for did, cname, val in db.execute(
            "select d.id, p.name, e.value from rocpd_kernel_dispatch d join rocpd_pmc_event e on e.event_id = d.event_id "
            "join rocpd_info_pmc p on e.pmc_id = p.id"):
        if did not in k:
            continue
        ev.setdefault(did, {})
        ev[did][cname] = ev[did].get(cname, 0.0) + float(val)
    return k, ev


def short(n):
    n = re.sub(r"\(.*$", "", n).replace("void ", "")
    n = re.sub(r"<.*", "", n)
    return n.split("::")[-1][:60]


def main():
    O, tag, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    tot, by = {}, {}
    t_pass = {}
    for i in (1, 2, 3):
        p = os.path.join(O, f"{tag}_step_pass{i}.db")
        if not os.path.exists(p):
            continue
        k, ev = load(p)
        t_pass[i] = sum(d for _, d in k.values())
        for did, (name, dur) in k.items():
            b = by.setdefault(short(name), {})
            if i == 1:
                b["ns"] = b.get("ns", 0) + dur
                b["launches"] = b.get("launches", 0) + 1
            for c, v in ev.get(did, {}).items():
                tot[c] = tot.get(c, 0.0) + v
                b[c] = b.get(c, 0.0) + v
    wl = sys.argv[4] if len(sys.argv) > 4 else "c2"
    out = {"command": f"python3 bench.py --workload {wl} --dtype fp16 --steps 2 --warmup 1 --other-modes '' --extra-workloads '' --no-roofline --no-cpu-baseline --no-input-compare --no-eval",
           "optimiser_steps_in_trace": steps, "kernel_ms_per_step": round(t_pass.get(1, 0) / steps / 1e6, 3)}
    if "GRBM_GUI_ACTIVE" in tot:
        cyc = tot["GRBM_GUI_ACTIVE"] / 8.0
        out["clock_ghz"] = round(cyc / max(1, t_pass[1]), 3)
        out["mfma_busy_frac"] = round(tot.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0 / cyc, 4)
        out["valu_per_mfma"] = round(tot.get("SQ_INSTS_VALU", 0) / max(1.0, tot.get("SQ_INSTS_MFMA", 0)), 3)
    rd, wr = tot.get("FETCH_SIZE", 0) * 1024 * 2, tot.get("WRITE_SIZE", 0) * 1024
    out["hbm_read_bytes_per_step"] = round(rd / steps)
    out["hbm_write_bytes_per_step"] = round(wr / steps)
    out["hbm_bytes_per_step"] = round((rd + wr) / steps)
    if t_pass.get(2) and t_pass.get(3):
        out["hbm_gbs"] = round(rd / t_pass[2] + wr / t_pass[3], 1)        # bytes per ns = GB/s, each from its own pass
    top = sorted(by.items(), key=lambda kv: -kv[1].get("ns", 0))[:14]
    out["by_kernel"] = {}
    for n, b in top:
        e = {"ms_per_step": round(b.get("ns", 0) / steps / 1e6, 3), "launches_per_step": round(b.get("launches", 0) / steps, 1)}
        if b.get("GRBM_GUI_ACTIVE"):
            e["mfma_busy_frac"] = round(b.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0 / (b["GRBM_GUI_ACTIVE"] / 8.0), 4)
        e["hbm_gb_per_step"] = round((b.get("FETCH_SIZE", 0) * 2048 + b.get("WRITE_SIZE", 0) * 1024) / steps / 1e9, 3)
        out["by_kernel"][n] = e
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
