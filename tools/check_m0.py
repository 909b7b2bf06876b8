#!/usr/bin/env python3
"""Build-time check: the fused MLP kernels leave M0 to the LDS-DMA inline asm (mlp_common.h::dma_piece*).
Scans hipcc's -save-temps ISA for any M0 reference outside ;;#ASMSTART/;;#ASMEND blocks."""
import re
import sys


def check(path):
    bad, inside = [], False
    for i, line in enumerate(open(path), 1):
        if "#ASMSTART" in line:
            inside = True
        elif "#ASMEND" in line:
            inside = False
        elif not inside and re.search(r"\bm0\b", line.split(";")[0]) and not line.lstrip().startswith("."):
            bad.append((i, line.rstrip()))
    return bad


if __name__ == "__main__":
    rc = 0
    for p in sys.argv[1:]:
        bad = check(p)
        if bad:
            rc = 1
            print(f"{p}: compiler-generated code touches M0:")
            for i, l in bad[:10]:
                print(f"  {i}: {l}")
        else:
            print(f"{p}: M0 untouched outside the DMA asm")
    sys.exit(rc)
