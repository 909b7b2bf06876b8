#!/usr/bin/env python3
"""Instruction mix of every kernel in a -save-temps ISA file: MFMAs, VALU per MFMA, stores, compares.
Usage: python tools/isa_mix.py build/csrc/<unit>-hip-amdgcn-amd-amdhsa-gfx950.s [name substring]"""
import collections, re, sys
text = open(sys.argv[1]).read()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_Z\w+):\s*; @\1\n(.*?)^\s*s_endpgm", text, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if sub not in name:
        continue
    ops = [l.split()[0] for l in body.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    c = collections.Counter(ops)
    grp = lambda p: sum(v for k, v in c.items() if k.startswith(p))
    mf = grp("v_mfma")
    valu = grp("v_") - mf
    print(f"{name[:72]}\n  instrs {len(ops)}  mfma {mf}  valu {valu} ({valu / max(mf, 1):.2f}/mfma)  global_store {grp('global_store')}  "
          f"global_load {grp('global_load')}  ds_read {grp('ds_read')}  v_cmp {grp('v_cmp')}  v_cndmask {grp('v_cndmask')}  "
          f"v_or/lshl_or {grp('v_or') + grp('v_lshl_or')}  v_cvt {grp('v_cvt')}  v_accvgpr {grp('v_accvgpr')}  v_mov {grp('v_mov')}  s_waitcnt {grp('s_waitcnt')}")
    print("  top VALU:", ", ".join(f"{k} {v}" for k, v in c.most_common(40) if k.startswith("v_") and not k.startswith("v_mfma"))[:600])
