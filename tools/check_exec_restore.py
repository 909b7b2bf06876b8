#!/usr/bin/env python3
"""ISA lint for a hipcc (ROCm 7.2) defect met in mlp_f16_2t.hip: a VGPR saved around a nested divergent region is restored
by a v_mov placed at the join of the INNER region, i.e. between a label that `s_cbranch_execz` jumps to (EXEC may be the
inner, possibly empty mask there) and the `s_or_b64 exec, exec, ...` that re-opens the outer mask -- lanes that took the
outer branch keep the clobbered value.  Flags vector instructions in that window.  Usage: check_exec_restore.py file.s ..."""
import re
import sys


def check(path):
    lines = open(path).read().split("\n")
    execz_targets = set()
    for l in lines:
        m = re.match(r"\s*s_cbranch_execz\s+(\S+)", l)
        if m:
            execz_targets.add(m.group(1))
    hits = []
    for i, l in enumerate(lines):
        m = re.match(r"(\.LBB\S+):", l)
        if not (m and m.group(1) in execz_targets):
            continue
        window = []
        for j in range(i + 1, min(i + 12, len(lines))):
            t = lines[j].strip()
            if not t or t.startswith(";"):
                continue
            if re.match(r"s_or_b64\s+exec,\s*exec", t):
                for k, w in window:
                    if re.match(r"v_|ds_|global_|buffer_|flat_|scratch_", w):
                        hits.append((k + 1, w))
                break
            if re.match(r"(\.LBB\S+):", t):
                continue               # fall-through labels belong to the same join sequence
            if t.startswith("s_cbranch") or t.startswith("s_branch") or t.startswith("s_endpgm"):
                break
            if re.search(r"saveexec|\bexec\b", t.split(",")[0]) or re.match(r"s_\w+\s+exec", t):
                break                  # EXEC is set afresh (an else-branch, a new region): what follows runs under a defined mask
            window.append((j, t))
    return hits


if __name__ == "__main__":
    rc = 0
    for p in sys.argv[1:]:
        hits = check(p)
        if hits:
            rc = 1
            print(f"{p}: vector instructions between an execz join label and the exec restore:")
            for k, w in hits[:10]:
                print(f"  {k}: {w}")
        else:
            print(f"{p}: no vector work between execz joins and their exec restores")
    sys.exit(rc)
