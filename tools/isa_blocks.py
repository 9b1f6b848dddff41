#!/usr/bin/env python3
"""Per-basic-block instruction counts of k_extend (default layout) from `hipcc -S --cuda-device-only` output: which blocks of the walk's loop
cost what (usage: isa_blocks.py file.s).  The mov column is how the loop-structure problem of DESIGN.md 7 (round 2) was found."""
import re,sys
s=open(sys.argv[1]).read()
name=re.findall(r"^(_ZN3rt38k_extendILb0ELi2E\w+):", s, re.M)[0]
code=s[s.index("\n"+name+":"):s.index(".amdhsa_kernel "+name)]
blocks=[]; cur=None
for l in code.split("\n"):
    if re.match(r"^\.LBB\d+_\d+:", l):
        cur=[l.split(":")[0], []]; blocks.append(cur)
    elif cur is not None and l.strip() and not l.strip().startswith(";") and not l.strip().startswith("."):
        cur[1].append(l.strip())
tot=0
for lab,ins in blocks:
    v=sum(1 for i in ins if i.startswith("v_")); mv=sum(1 for i in ins if i.startswith("v_mov")); sa=sum(1 for i in ins if i.startswith("s_")); ld=sum(1 for i in ins if "load" in i); ds=sum(1 for i in ins if i.startswith("ds_"))
    tag = "cvt" if any("cvt_f32_ubyte" in i for i in ins) else ("div" if any("div_scale" in i for i in ins) else "")
    if v>=5: print(f"{lab:10s} n={len(ins):4d} valu={v:4d} mov={mv:3d} salu={sa:3d} loads={ld:2d} ds={ds:2d} {tag}")
