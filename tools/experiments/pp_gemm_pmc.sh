#!/bin/bash
# cycles (GRBM_GUI_ACTIVE / 8) and clock of the ping-pong GEMM's timing ablations (AFM_GEMM_ABLATIONS build): 0 full,
# 2 no LDS-DMA, 4 no epilogue, 6 LDS reads + MFMAs only -- at N 512 / K 2048 (launches 150 .. 299 of each kernel) and N 1536 / K 512 (450 .. 599)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export AFM_LIB_OVERRIDE=$R/tools/experiments/_abl/libafm_gemmabl.so
timeout 150 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES -d $O/pmc_gabl -o pmc -- python3 $R/tools/experiments/pp_gemm_pmc.py > $O/pp_gemm_pmc.log 2>&1
python3 - <<PY
import sqlite3, re, glob, json
db = sqlite3.connect(glob.glob("$O/pmc_gabl/**/*.db", recursive=True)[0])
rows = db.execute("select s.kernel_name, d.id, d.end - d.start, p.name, e.value from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
                  "left join rocpd_pmc_event e on e.event_id = d.event_id left join rocpd_info_pmc p on e.pmc_id = p.id order by d.id").fetchall()
disp = {}
for k, did, dur, cn, v in rows:
    if "gemm_nt_pp" not in k: continue
    e = disp.setdefault(did, {"k": k, "ns": dur})
    if cn: e[cn] = e.get(cn, 0.0) + float(v)
byk = {}
for did in sorted(disp): byk.setdefault(disp[did]["k"], []).append(disp[did])
out = {}
for k, lst in byk.items():
    m = re.search(r"gemm_nt_ppILi\d+ELi(\d+)E", k)
    abl = int(m.group(1)) if m else -1
    for shape, part in (("N512_K2048", lst[150:300]), ("N1536_K512", lst[450:600])):
        if not part: continue
        n = len(part); ns = sum(x["ns"] for x in part) / n; cyc = sum(x.get("GRBM_GUI_ACTIVE", 0) for x in part) / n / 8
        out[f"{shape}_abl{abl}"] = {"us": ns / 1e3, "kcycles": cyc / 1e3, "clock_ghz": cyc / ns,
                                    "mfma_busy": sum(x.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for x in part) / n / (cyc * 1024),
                                    "valu_busy": sum(x.get("SQ_ACTIVE_INST_VALU", 0) for x in part) / n * 4 / (cyc * 1024),
                                    "lds_busy": sum(x.get("SQ_ACTIVE_INST_LDS", 0) for x in part) / n * 4 / (cyc * 1024), "launches": n}
json.dump(out, open("$O/pp_gemm_abl_pmc.json", "w"), indent=1)
for k in sorted(out): print(k, {a: round(b, 3) for a, b in out[k].items()})
PY
rm -rf $O/pmc_gabl
