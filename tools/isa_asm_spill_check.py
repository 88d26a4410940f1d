"""Hazard check for kernels that issue LDS reads through inline asm (ds_read_b64_tr_b16 without a wait: afm_attn_tiles.h): the compiler
does not know those reads are asynchronous, so a SPILL of such a register right behind the read would store it before the data is there.
`python tools/isa_asm_spill_check.py file.s [kernel-filter]`: for every kernel, every scratch_store whose source registers were written by
an asm LDS read (any `ds_read*` inside an asm statement) since the last s_waitcnt lgkmcnt(0) is reported.  (Compile with hipcc --cuda-device-only -S.)"""
import re, sys


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def check(path, flt=""):
    s = open(path).read()
    bad = 0
    for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)\.Lfunc_end", s, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if flt not in name:
            continue
        inflight, spills, in_asm = set(), 0, False
        for ln in body.split("\n"):
            t = ln.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
            elif t.startswith(";;#ASMEND"):
                in_asm = False
            if t.startswith("ds_read_b64_tr_b16") or (in_asm and t.startswith("ds_read")):      # (compiler-issued reads are tracked by the compiler)
                inflight |= regs(t.split()[1].rstrip(","))
            elif re.match(r"s_waitcnt.*lgkmcnt\(0\)", t) or (t.startswith("s_waitcnt") and "lgkmcnt" not in t and "vmcnt" not in t and "expcnt" not in t):
                inflight.clear()
            elif t.startswith("scratch_store"):
                spills += 1
                src = regs(t.split()[2].rstrip(","))
                if src & inflight:
                    bad += 1
                    print(f"HAZARD {name}: {t}")
        print(f"{name}: {spills} spill stores checked")
    return bad


if __name__ == "__main__":
    sys.exit(1 if check(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "") else 0)
