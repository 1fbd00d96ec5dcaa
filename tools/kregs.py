#!/usr/bin/env python3
"""VGPR / SGPR / LDS / scratch of every kernel in a gfx950 assembly file (hipcc --save-temps): tools/kregs.py file.s [filter]"""
import re, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name")
    if flt in name:
        print(f"{name[:90]:90s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size')}")
