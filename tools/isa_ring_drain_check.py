"""Scan a kernel file's ISA for what drains an LDS-DMA ring: inside loops of kernels that issue `global_load_lds`, list scratch reloads
(each comes with an `s_waitcnt vmcnt(0)` when its value is used -- all pieces in flight are waited for) and every `s_waitcnt vmcnt(0)`.
`python tools/isa_ring_drain_check.py file.s [kernel-filter]`   (hipcc --cuda-device-only -S)"""
import re, sys


def scan(path, flt=""):
    s = open(path).read()
    out = []
    for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)\.Lfunc_end", s, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if flt not in name or "global_load_lds" not in body:
            continue
        lines = body.split("\n")
        # loop bodies: from a label marked "Loop Header" to the last line that says "in Loop: Header=<that label>"
        hdrs = [(i, re.match(r"^(\.LBB\w+):", l).group(1)) for i, l in enumerate(lines) if "Loop Header" in l and l.startswith(".LBB")]
        nsl = nv0 = ndma = 0
        for i0, lab in hdrs:
            tag = "Header=" + lab[2:]
            last = max([i for i, l in enumerate(lines) if tag in l] + [i0])
            # extend to the backward branch
            for j in range(last, min(last + 400, len(lines))):
                if re.search(r"s_cbranch\w+\s+" + re.escape(lab) + r"\b", lines[j]):
                    last = j
                    break
            seg = lines[i0:last + 1]
            if not any("global_load_lds" in l for l in seg):
                continue
            nsl += sum("scratch_load" in l for l in seg)
            nv0 += sum(bool(re.search(r"s_waitcnt.*vmcnt\(0\)", l)) for l in seg)
            ndma += sum("global_load_lds" in l for l in seg)
        out.append((name, ndma, nsl, nv0))
    return out


if __name__ == "__main__":
    for name, ndma, nsl, nv0 in scan(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""):
        if ndma:
            print(f"{nsl:3d} scratch reloads, {nv0:3d} vmcnt(0), {ndma:3d} LDS-DMA pieces in ring loops  {name[:110]}")
