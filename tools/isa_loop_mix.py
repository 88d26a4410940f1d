"""Instruction mix of a kernel's main loop (first .. last MFMA) from hipcc -S output:  python tools/isa_loop_mix.py file.s <mangled-name-substring>"""
import collections, sys
txt = open(sys.argv[1]).read()
key = sys.argv[2]
start = txt.index("\n" + [l for l in txt.split("\n") if l.startswith("_Z") and key in l and l.rstrip().endswith(tuple(":")) or (l.startswith("_Z") and key in l and ": " in l)][0].split(":")[0] + ":")
end = txt.index("s_endpgm", start)
lines = txt[start:end].split("\n")
idx = [i for i, l in enumerate(lines) if l.strip().startswith("v_mfma")]
body = lines[idx[0]: idx[-1] + 1]
c = collections.Counter()
for l in body:
    l = l.strip()
    if not l or l.startswith((".", ";", "//")) or l.endswith(":"):
        continue
    c[l.split()[0]] += 1
nm = sum(n for o, n in c.items() if o.startswith("v_mfma"))
nv = sum(n for o, n in c.items() if o.startswith("v_") and not o.startswith("v_mfma"))
print(f"lines {len(body)}  mfma {nm}  valu {nv}  ds {sum(n for o, n in c.items() if o.startswith('ds_'))}  salu/smem {sum(n for o, n in c.items() if o.startswith('s_'))}  valu/mfma {nv / max(nm, 1):.1f}")
for o, n in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 25):
    print(f"  {o:28s} {n}")
