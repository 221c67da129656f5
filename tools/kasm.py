"""Instruction histogram and resource lines of one kernel in a hipcc -S listing: python tools/kasm.py file.s kernel_substring"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
m = re.search(r'^(\S*%s\S*):' % re.escape(sys.argv[2]), s, re.M)
i = m.start()
j = s.index('.end_amdhsa_kernel', i) if '.end_amdhsa_kernel' in s[i:] else len(s)
body = s[i:s.index('s_endpgm', i)] if 's_endpgm' in s[i:j] else s[i:j]
# whole function incl. all s_endpgm: up to the .section / .rodata that follows
k = s.find('.section', i)
body = s[i:k if k > 0 else j]
ins = []
for l in body.split('\n'):
    t = l.strip()
    if not l.startswith('\t') or not t or t[0] in '.;': continue
    ins.append(t.split()[0])
c = Counter(ins)
print(m.group(1), len(ins), "instructions")
for name, n in c.most_common(30): print("  %-28s %d" % (name, n))
for key in ('NumVgprs', 'NumAgprs', 'ScratchSize', 'Occupancy', 'LDSByteSize', 'NumSgprs'):
    r = re.search(r';\s*%s:\s*(\d+)' % key, s[i:])
    if r: print("  %s = %s" % (key, r.group(1)))
