"""Prints, for kernels whose mangled name contains a pattern, the s_waitcnt vmcnt values, scratch use and register
counts found in a -save-temps ISA file.  Usage: python tools/isa_waits.py file.s pattern"""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
meta = {m.group(3): (m.group(1), m.group(2), m.group(4), m.group(5)) for m in re.finditer(
    r'\.agpr_count:\s*(\d+).*?\.group_segment_fixed_size:\s*(\d+).*?\.name:\s*(\S+).*?\.private_segment_fixed_size:\s*(\d+).*?\.vgpr_count:\s*(\d+)', s, re.S)}
for m in re.finditer(r'^(\S+):\s*; @\1', s, re.M):
    nm = m.group(1)
    if pat not in nm:
        continue
    j = s.index('.end_amdhsa_kernel', m.end()) if '.end_amdhsa_kernel' in s[m.end():] else len(s)
    k = s[m.end():j]
    k = k[:k.index('s_endpgm') + 8] if 's_endpgm' in k else k
    print(nm)
    print('  vmcnt waits:', ' '.join(re.findall(r's_waitcnt vmcnt\((\d+)\)', k)))
    print('  scratch ops', len(re.findall(r'scratch_', k)), ' barriers', k.count('s_barrier'), ' mfma', k.count('v_mfma'),
          ' agpr/lds/scratch/vgpr', meta.get(nm))
