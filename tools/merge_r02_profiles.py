"""gpurun_out/prof_r02/* (tools/prof_r02.sh) -> profiles/r02_*: per-kernel step statistics of both precision modes and
merged PMC tables (one JSON per kernel family) with the derived quantities the guide prescribes
(MI355X_MICROARCH.md, HBM section: FETCH_SIZE in KB of 64-B requests -> x 1024 x 2 for wide coalesced reads; WRITE_SIZE
KB exact; SQ_VALU_MFMA_BUSY_CYCLES = 32 x MFMAs of 32x32x16 / 16 x MFMAs of 16x16x32 per SIMD; SQ_WAVE_CYCLES etc. in quad-cycles)."""
import glob
import json
import os
import re
import shutil

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(R, "gpurun_out", "prof_r02")
DST = os.path.join(R, "profiles")

NAMES = {"k_attn_fwd_x3": "attention forward (encoder self-attention)", "k_attn_bwd_dq_x3": "attention backward, dQ kernel",
         "k_attn_bwd_dkv_x3": "attention backward, dK/dV kernel", "k_attn_fwd_mfma": "attention forward (encoder self-attention)",
         "k_attn_bwd_dq_mfma": "attention backward, dQ kernel", "k_attn_bwd_dkv_mfma": "attention backward, dK/dV kernel",
         "k_x3_nt256": "FFN up-projection forward (plain epilogue in the PMC run)", "k_x3_tn256": "FFN up-projection weight gradient (bias gradient fused)",
         "k_x3_tn": "FFN up-projection weight gradient, 256 x 128 form"}


def pretty(k):
    m = re.match(r"_Z(\d+)", k)           # Itanium mangling: _Z<len><name>...
    return k[m.end():m.end() + int(m.group(1))] if m else k


def merge(prefix):
    out = {}
    for f in sorted(glob.glob(os.path.join(SRC, prefix + "_pmc_*.json"))):
        for k, v in json.load(open(f)).items():
            o = out.setdefault(k, {})
            for c, x in v.items():
                if c in ("dispatches", "avg_ns"):
                    o.setdefault(c, x)
                else:
                    o[c] = x
    res = {}
    for k, o in out.items():
        e = dict(o)
        if "FETCH_SIZE" in e:
            e["hbm_read_bytes_corrected"] = e["FETCH_SIZE"] * 1024 * 2
        if "WRITE_SIZE" in e:
            e["hbm_write_bytes"] = e["WRITE_SIZE"] * 1024
        if "hbm_read_bytes_corrected" in e and "hbm_write_bytes" in e:
            e["hbm_bytes_per_launch"] = e["hbm_read_bytes_corrected"] + e["hbm_write_bytes"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "GRBM_GUI_ACTIVE" in e:
            # busy cycles summed over 1024 SIMDs / (GUI active cycles summed over 8 XCDs / 8)
            e["mfma_busy_frac"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (e["GRBM_GUI_ACTIVE"] / 8.0)
        if "SQ_INSTS_VALU" in e and "SQ_INSTS_MFMA" in e:
            e["valu_per_mfma"] = e["SQ_INSTS_VALU"] / max(1.0, e["SQ_INSTS_MFMA"])
        if "TCC_HIT_sum" in e:
            e["l2_hit_rate"] = e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"])
        e["what"] = NAMES.get(pretty(k), pretty(k))
        tag = "<dropout, re-hash>" if "ILi1E" in k else "<dropout, keep bits>" if "ILi2E" in k else ""
        res[pretty(k) + tag] = e
    return res


def main():
    os.makedirs(DST, exist_ok=True)
    for m in ("bf16x3-mixed", "bf16x3", "bf16"):
        if not os.path.exists(os.path.join(SRC, f"step_{m}_kernel_stats.csv")):
            continue
        shutil.copy(os.path.join(SRC, f"step_{m}_kernel_stats.csv"), os.path.join(DST, f"r02_c2_{m}_kernel_stats.csv"))
        tot = open(os.path.join(SRC, f"step_{m}_total.txt")).read().strip()
        with open(os.path.join(DST, f"r02_c2_{m}_kernel_stats.csv"), "a") as fh:
            fh.write(tot + "\n")
    table = {}
    for m in ("bf16x3", "bf16"):
        a = merge(f"attn_{m}")
        json.dump(a, open(os.path.join(DST, f"r02_attn_{m}_pmc.json"), "w"), indent=1)
        hsh = merge(f"attn_{m}_hash")
        if hsh:
            json.dump(hsh, open(os.path.join(DST, f"r02_attn_{m}_rehash_pmc.json"), "w"), indent=1)
        table[m] = {e["what"]: {"hbm_bytes_per_launch": e.get("hbm_bytes_per_launch"), "kernel": k} for k, e in a.items()}
    g = merge("gemm_bf16x3")
    json.dump(g, open(os.path.join(DST, "r02_gemm_bf16x3_pmc.json"), "w"), indent=1)
    for k, e in g.items():
        if k.startswith("k_x3_tn256"):
            table["bf16x3"][e["what"]] = {"hbm_bytes_per_launch": e.get("hbm_bytes_per_launch"), "kernel": k}
    for m in table:
        json.dump(table[m], open(os.path.join(DST, f"r02_{m}_pmc.json"), "w"), indent=1)
    print(json.dumps({m: {k: v["hbm_bytes_per_launch"] for k, v in t.items()} for m, t in table.items()}, indent=1))


if __name__ == "__main__":
    main()
