"""Instruction mix per kernel (and per innermost loop) of a gfx950 assembly file produced by
`hipcc -S --cuda-device-only`: counts of MFMA / VALU / transcendental / LDS / VMEM / scratch / SALU instructions.

    python tools/isa_mix.py /tmp/kernel.s [name-substring]
"""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")): return "trans"
    if op.startswith("v_pk_"): return "valu_pk"
    if op.startswith("v_cvt"): return "cvt"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_")): return "vmem"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_"): return "salu"
    return "other"


def main():
    text = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    parts = re.split(r"\n(_Z\w+):[^\n]*\n", text)
    for i in range(1, len(parts), 2):
        name, body = parts[i], parts[i + 1].split(".Lfunc_end")[0]
        if want not in name:
            continue
        lines = body.split("\n")
        # basic blocks by label; a loop body = blocks between a label and a backward branch to it
        labels, ops = {}, []
        for ln in lines:
            m = re.match(r"^(\.LBB\d+_\d+):", ln)
            if m:
                labels[m.group(1)] = len(ops)
                continue
            t = ln.strip()
            if not ln.startswith("\t") or t.startswith((".", ";")) or not t:
                continue
            ops.append(t)
        total = Counter(classify(o.split()[0]) for o in ops)
        print(f"== {name[:70]}\n   whole kernel: {dict(total)}")
        loops = []
        for idx, o in enumerate(ops):
            m = re.match(r"s_cbranch\w*\s+(\.LBB\d+_\d+)", o)
            if m and m.group(1) in labels and labels[m.group(1)] <= idx:
                loops.append((labels[m.group(1)], idx))
        for a, b in sorted(loops, key=lambda r: r[0] - r[1])[:3]:
            c = Counter(classify(o.split()[0]) for o in ops[a:b + 1])
            print(f"   loop [{a}:{b}] {b - a + 1} instrs: {dict(c)}")


if __name__ == "__main__":
    main()
