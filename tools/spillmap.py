"""Where a kernel's register spills sit: scratch loads / stores of one function of a gfx950 assembly listing built with
-gline-tables-only, counted per source file and ten-line block.  python tools/spillmap.py listing.s mangled_function_name"""
import collections
import re
import sys

files, cur, fn = {}, None, None
cnt = collections.Counter()
for l in open(sys.argv[1]):
    m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2))
    m = re.match(r'^(_Z\w+):', l)
    if m:
        fn = m.group(1)
    m = re.match(r'\s+\.loc\s+(\d+)\s+(\d+)', l)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
    if fn == sys.argv[2] and re.search(r'\bscratch_(load|store)', l) and cur:
        cnt[(files.get(cur[0], '?').split('/')[-1], cur[1] // 10 * 10, 'ld' if 'scratch_load' in l else 'st')] += 1
for (f, ln, k), c in sorted(cnt.items(), key=lambda x: -x[1])[:50]:
    print("%-22s %5d %s %4d" % (f, ln, k, c))
