"""gpurun_out/prof_r03/* (tools/prof_r03.sh) -> profiles/r03_*: step statistics of the timed mode and of bf16x3-mixed, and the merged
PMC table of the single-pass attention kernels in fp16 with the derived quantities (see tools/merge_r02_profiles.py)."""
import json
import os
import re
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import merge_r02_profiles as M2  # noqa: E402

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(R, "gpurun_out", "prof_r03")
DST = os.path.join(R, "profiles")


def pretty(k):
    m = re.match(r"_ZN\d+afm_(?:f16|bf16)(\d+)", k)      # kernels of the per-format namespaces: _ZN7afm_f16<len><name>...
    if m:
        return k[m.end():m.end() + int(m.group(1))]
    m = re.match(r"_Z(\d+)", k)
    return k[m.end():m.end() + int(m.group(1))] if m else k


def main():
    M2.SRC = SRC
    M2.pretty = pretty
    M2.NAMES.update({"k_attn_fwd_st": "attention forward, 8-wave staggered form (opt-in)", "k_attn_bwd_dq_st": "attention backward, dQ kernel, 8-wave staggered form (opt-in)"})
    for m in ("fp16", "bf16x3-mixed"):
        src = os.path.join(SRC, f"step_{m}_kernel_stats.csv")
        if os.path.exists(src):
            dst = os.path.join(DST, f"r03_c2_{m}_kernel_stats.csv")
            shutil.copy(src, dst)
            with open(dst, "a") as fh:
                fh.write(open(os.path.join(SRC, f"step_{m}_total.txt")).read().strip() + "\n")
    if os.path.exists(os.path.join(SRC, "attn_fp16_kernel_stats.csv")):
        shutil.copy(os.path.join(SRC, "attn_fp16_kernel_stats.csv"), os.path.join(DST, "r03_attn_fp16_kernel_stats.csv"))
    a = M2.merge("attn_fp16")
    json.dump(a, open(os.path.join(DST, "r03_attn_fp16_pmc.json"), "w"), indent=1)
    table = {e["what"]: {"hbm_bytes_per_launch": e.get("hbm_bytes_per_launch"), "kernel": k} for k, e in a.items()}
    json.dump(table, open(os.path.join(DST, "r03_fp16_pmc.json"), "w"), indent=1)
    for k, e in a.items():
        print(k, {c: round(e[c], 3) for c in ("mfma_busy_frac", "valu_per_mfma", "l2_hit_rate") if c in e}, e.get("hbm_bytes_per_launch"), round(e["avg_ns"] / 1e3, 1), "us")


if __name__ == "__main__":
    main()
