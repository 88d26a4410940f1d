"""gpurun_out/prof_r05/* (tools/prof_r05.sh) -> profiles/r05_*: step statistics of c2 and c4 in the timed mode, the merged PMC table of
the attention kernels, the dQ kernel in both MFMA shapes, the GEMM PMC tables at the c2 and c4 shapes, the step-level counters of both
workloads, and `r05_c2_fp16_pmc.json` / `r05_c4_fp16_pmc.json` = {bench.py roofline `what`: HBM bytes per launch} (bench.py fills `traffic`
of `roofline*` and of `workloads.c4.roofline_kernels` from them)."""
import json
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import merge_r02_profiles as M2  # noqa: E402
import merge_r03_profiles as M3  # noqa: E402

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(R, "gpurun_out", "prof_r05")
DST = os.path.join(R, "profiles")
GEMM_WHAT = {
    "c2": {"qkv_fwd_default": "QKV projection forward",
           "ffn1_fwd_gelu_drop_sg_default": "FFN up-projection forward (bias + GELU + dropout fused, keep*scale*GELU' stored)",
           "ffn2_dgrad_xsaved_default": "FFN down-projection data gradient (x stored keep*scale*GELU')",
           "ffn1_wgrad_default": "FFN up-projection weight gradient (bias gradient fused)",
           "ffn2_fwd_default": "FFN down-projection forward", "qkv_dgrad_default": "QKV projection data gradient",
           "enc_layer_wgrad_group_default": "weight gradients of one encoder layer, grouped launch (bias gradients fused)"},
    "c4": {"c4_qkv_fwd_default": "QKV projection forward", "c4_ffn2_fwd_default": "FFN down-projection forward",
           "c4_qkv_dgrad_default": "QKV projection data gradient", "c4_ffn1_dgrad_2f_default": "gated FFN up-projection data gradient",
           "c4_glu_fwd_sg_default": "gated FFN up-projection forward (bias + gelu(u) v + dropout fused, keep*scale*[gelu'(u) v | gelu(u)] stored)",
           "c4_glu_dgrad_default": "gated FFN down-projection data gradient (x stored factors -> [du | dv])",
           "c4_enc_layer_wgrad_group_default": "weight gradients of one encoder layer, grouped launch (bias gradients fused)"}}


def main():
    M2.SRC = SRC
    M2.pretty = M3.pretty
    M2.NAMES.update({"k_attn_bwd_dkv_pipe": "attention backward, dK/dV kernel", "k_attn_bwd_dq_m16": "attention backward, dQ kernel on v_mfma_f32_16x16x32 (A/B form)",
                     "k_attn_bwd_dkv_pipe16": "attention backward, dK/dV kernel"})
    for wl in ("c2", "c4"):
        src = os.path.join(SRC, f"step_{wl}_fp16_kernel_stats.csv")
        if os.path.exists(src):
            dst = os.path.join(DST, f"r05_{wl}_fp16_kernel_stats.csv")
            shutil.copy(src, dst)
            tot = os.path.join(SRC, f"step_{wl}_fp16_total.txt")
            if os.path.exists(tot):
                with open(dst, "a") as fh:
                    fh.write(open(tot).read().strip() + "\n")
        sp = os.path.join(SRC, f"r05_{wl}_fp16_step_pmc.json")
        if os.path.exists(sp):
            shutil.copy(sp, os.path.join(DST, f"r05_{wl}_fp16_step_pmc.json"))
    if os.path.exists(os.path.join(SRC, "attn_fp16_kernel_stats.csv")):
        shutil.copy(os.path.join(SRC, "attn_fp16_kernel_stats.csv"), os.path.join(DST, "r05_attn_fp16_kernel_stats.csv"))
    tables = {"c2": {}, "c4": {}}
    if os.path.exists(os.path.join(SRC, "attn_fp16_pmc_1.json")):
        a = M2.merge("attn_fp16")
        json.dump(a, open(os.path.join(DST, "r05_attn_fp16_pmc.json"), "w"), indent=1)
        tables["c2"].update({e["what"]: {"hbm_bytes_per_launch": e.get("hbm_bytes_per_launch"), "kernel": k} for k, e in a.items()})
        for k, e in a.items():
            print(k, {c: round(e[c], 3) for c in ("mfma_busy_frac", "valu_per_mfma", "l2_hit_rate") if c in e}, e.get("hbm_bytes_per_launch"), round(e["avg_ns"] / 1e3, 1), "us")
    # the dQ kernel in both MFMA shapes: two passes (issue counters; LDS + GRBM)
    if os.path.exists(os.path.join(SRC, "dqshape_pmc_1.json")):
        out = {}
        for i in (1, 2):
            pth = os.path.join(SRC, f"dqshape_pmc_{i}.json")
            if not os.path.exists(pth):
                continue
            for k, e in json.load(open(pth)).items():
                o = out.setdefault(k, {})
                o.setdefault("avg_ns_by_pass", []).append(round(e.pop("avg_ns")))
                o["dispatches"] = e.pop("dispatches")
                o.update(e)
        for k, m in out.items():
            if "GRBM_GUI_ACTIVE" in m:
                cyc = m["GRBM_GUI_ACTIVE"] / 8.0
                m["kcycles"] = round(cyc / 1e3, 1)
                m["clock_ghz"] = round(cyc / m["avg_ns_by_pass"][-1], 3)
                if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
                    m["mfma_busy_frac"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc, 4)
            if m.get("SQ_INSTS_MFMA"):
                m["valu_per_mfma"] = round(m.get("SQ_INSTS_VALU", 0) / m["SQ_INSTS_MFMA"], 2)
                m["salu_per_mfma"] = round(m.get("SQ_INSTS_SALU", 0) / m["SQ_INSTS_MFMA"], 2)
            print(k[:70], {c: m.get(c) for c in ("avg_ns_by_pass", "kcycles", "clock_ghz", "mfma_busy_frac", "valu_per_mfma", "salu_per_mfma")})
        out["_note"] = ("tools/experiments/attn_m16.py --time-only under two rocprofv3 counter passes: means over all launches of a kernel name; the template "
                        "arguments are the dropout path (0 none, 1 hash, 2 keep-bit tensor) and the build; k_attn_fwd_mfma / k_attn_bwd_dq_mfma / k_attn_bwd_dkv_pipe / "
                        "k_attn_bwd_dkv_mfma run v_mfma_f32_32x32x16, k_attn_fwd_m16 / k_attn_bwd_dq_m16 / k_attn_bwd_dkv_pipe16 / k_attn_bwd_dkv_m16 run "
                        "v_mfma_f32_16x16x32 at the same tile per wave; cycles = GRBM_GUI_ACTIVE / 8, clock = cycles / time")
        json.dump(out, open(os.path.join(DST, "r05_attn_shape_pmc.json"), "w"), indent=1)
    for wl, fn in (("c2", "r05_gemm_fp16_pmc.json"), ("c4", "r05_c4_gemm_fp16_pmc.json")):
        gp = os.path.join(SRC, fn)
        if not os.path.exists(gp):
            continue
        g = json.load(open(gp))
        json.dump(g, open(os.path.join(DST, fn), "w"), indent=1)
        for name, what in GEMM_WHAT[wl].items():
            if name in g and "hbm_bytes_per_launch" in g[name]:
                tables[wl][what] = {"hbm_bytes_per_launch": g[name]["hbm_bytes_per_launch"], "kernel": g[name].get("kernel"), "config": name}
        for n, e in g.items():
            print(n, {c: e.get(c) for c in ("mfma_busy_frac", "clock_ghz", "valu_per_mfma", "l2_hit_rate", "hbm_bytes_per_launch")}, (e.get("avg_ns_by_pass") or [0])[0] / 1e3, "us")
    for wl in ("c2", "c4"):
        if tables[wl]:
            json.dump(tables[wl], open(os.path.join(DST, f"r05_{wl}_fp16_pmc.json"), "w"), indent=1)
    pr = os.path.join(R, "gpurun_out", "parity_records.jsonl")
    if os.path.exists(pr):
        shutil.copy(pr, os.path.join(DST, "r05_parity_records.jsonl"))


if __name__ == "__main__":
    main()
