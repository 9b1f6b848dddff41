#!/usr/bin/env python3
"""Instruction statistics of the traversal kernels from `hipcc -S --cuda-device-only` output (usage: isa_stats.py file.s)."""
import re
import sys

s = open(sys.argv[1]).read()
for kern in ("k_extend", "k_shadow"):
    for lay in (0, 1, 2, 3):
        name = re.findall(r"^(_ZN3rt3\d+%sILb0ELi%dE\w+):" % (kern, lay), s, re.M)
        if not name:
            continue
        name = name[0]
        code = s[s.index("\n" + name + ":"):s.index(".amdhsa_kernel " + name)]
        vg = re.search(re.escape(name) + r"\.num_vgpr, (\d+)", s).group(1)
        print(kern, "layout", lay, "dwordx4 loads", code.count("global_load_dwordx4"), "flat", code.count("flat_load"), "scratch", code.count("scratch_"),
              "valu", len(re.findall(r"^\s+v_", code, re.M)), "salu", len(re.findall(r"^\s+s_", code, re.M)), "fma", code.count("v_fma_f32"),
              "cvt_ubyte", code.count("v_cvt_f32_ubyte"), "vgpr", vg)
