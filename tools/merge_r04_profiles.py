"""gpurun_out/prof_r04/* (tools/prof_r04.sh) -> profiles/r04_*: step statistics of the timed mode, the merged PMC table of the attention
kernels, the GEMM PMC table, the step-level counters, and `r04_fp16_pmc.json` = {bench.py roofline `what`: HBM bytes per launch} for
the attention AND the GEMM entries (bench.py fills `traffic` from it)."""
import json
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import merge_r02_profiles as M2  # noqa: E402
import merge_r03_profiles as M3  # noqa: E402

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(R, "gpurun_out", "prof_r04")
DST = os.path.join(R, "profiles")
GEMM_WHAT = {"qkv_fwd_default": "QKV projection forward",
             "ffn1_fwd_gelu_drop_sg_default": "FFN up-projection forward (bias + GELU + dropout fused, keep*scale*GELU' stored)",
             "ffn2_dgrad_xsaved_default": "FFN down-projection data gradient (x stored keep*scale*GELU')",
             "ffn1_wgrad_default": "FFN up-projection weight gradient (bias gradient fused)"}


def main():
    M2.SRC = SRC
    M2.pretty = M3.pretty
    M2.NAMES.update({"k_attn_bwd_dkv_pipe": "attention backward, dK/dV kernel", "k_attn_fwd_st": "attention forward, 8-wave staggered form (opt-in)",
                     "k_attn_bwd_dq_st": "attention backward, dQ kernel, 8-wave staggered form (opt-in)"})
    src = os.path.join(SRC, "step_fp16_kernel_stats.csv")
    if os.path.exists(src):
        dst = os.path.join(DST, "r04_c2_fp16_kernel_stats.csv")
        shutil.copy(src, dst)
        with open(dst, "a") as fh:
            fh.write(open(os.path.join(SRC, "step_fp16_total.txt")).read().strip() + "\n")
    if os.path.exists(os.path.join(SRC, "attn_fp16_kernel_stats.csv")):
        shutil.copy(os.path.join(SRC, "attn_fp16_kernel_stats.csv"), os.path.join(DST, "r04_attn_fp16_kernel_stats.csv"))
    table = {}
    if os.path.exists(os.path.join(SRC, "attn_fp16_pmc_1.json")):
        a = M2.merge("attn_fp16")
        json.dump(a, open(os.path.join(DST, "r04_attn_fp16_pmc.json"), "w"), indent=1)
        table.update({e["what"]: {"hbm_bytes_per_launch": e.get("hbm_bytes_per_launch"), "kernel": k} for k, e in a.items()})
        for k, e in a.items():
            print(k, {c: round(e[c], 3) for c in ("mfma_busy_frac", "valu_per_mfma", "l2_hit_rate") if c in e}, e.get("hbm_bytes_per_launch"), round(e["avg_ns"] / 1e3, 1), "us")
    gp = os.path.join(SRC, "r04_gemm_fp16_pmc.json")
    if os.path.exists(gp):
        g = json.load(open(gp))
        json.dump(g, open(os.path.join(DST, "r04_gemm_fp16_pmc.json"), "w"), indent=1)
        for name, what in GEMM_WHAT.items():
            if name in g and "hbm_bytes_per_launch" in g[name]:
                table[what] = {"hbm_bytes_per_launch": g[name]["hbm_bytes_per_launch"], "kernel": g[name].get("kernel"), "config": name}
        for n, e in g.items():
            print(n, {c: e.get(c) for c in ("mfma_busy_frac", "clock_ghz", "valu_per_mfma", "l2_hit_rate", "hbm_bytes_per_launch")}, e.get("avg_ns_by_pass", [0])[0] / 1e3, "us")
    json.dump(table, open(os.path.join(DST, "r04_fp16_pmc.json"), "w"), indent=1)
    sp = os.path.join(SRC, "r04_c2_fp16_step_pmc.json")
    if os.path.exists(sp):
        shutil.copy(sp, os.path.join(DST, "r04_c2_fp16_step_pmc.json"))
    pr = os.path.join(R, "gpurun_out", "parity_records.jsonl")
    if os.path.exists(pr):
        shutil.copy(pr, os.path.join(DST, "r04_parity_records.jsonl"))


if __name__ == "__main__":
    main()
