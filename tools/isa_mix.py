"""Instruction mix of a kernel between its s_barrier's, from hipcc -S output:  python tools/isa_mix.py file.s <mangled-name-substring>"""
import collections
import sys

txt = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = [i for i, l in enumerate(txt) if key in l and l.startswith("_Z") and ":" in l][0]
end = [i for i, l in enumerate(txt) if i > start and "s_endpgm" in l][0]
seg, segs = collections.Counter(), []
for l in txt[start:end]:
    l = l.strip()
    if not l or l.startswith((".", ";", "//")):
        continue
    op = l.split()[0]
    if op.endswith(":"):
        continue
    if op == "s_barrier":
        segs.append(seg)
        seg = collections.Counter()
        continue
    seg[op] += 1
segs.append(seg)
for i, s in enumerate(segs):
    tot = sum(s.values())
    mf = sum(v for k, v in s.items() if "mfma" in k)
    valu = sum(v for k, v in s.items() if k.startswith("v_") and "mfma" not in k)
    ds = sum(v for k, v in s.items() if k.startswith("ds_"))
    gl = sum(v for k, v in s.items() if k.startswith(("global_", "buffer_")))
    if tot > 60:
        print(i, "total", tot, "mfma", mf, "valu", valu, "ds", ds, "global", gl, "waitcnt", s.get("s_waitcnt", 0), "nop", s.get("s_nop", 0),
              [(k, v) for k, v in s.most_common(9) if k.startswith("v_") and "mfma" not in k])
