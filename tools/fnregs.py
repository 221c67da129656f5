"""Per-function code size / VGPR / SGPR / scratch of a gfx950 assembly listing (hipcc --cuda-device-only -S)."""
import re
import subprocess
import sys

d = {}
name = None
for l in open(sys.argv[1]):
    m = re.match(r'\s+\.type\s+(\S+),@function', l)
    if m:
        name = m.group(1)
    m = re.match(r'; (codeLenInByte|NumVgprs|ScratchSize|NumSgprs|Occupancy)\s*[:=]\s*(\d+)', l)
    if m and name:
        d.setdefault(name, {})[m.group(1)] = m.group(2)
names = list(d)
dem = subprocess.run(['c++filt'], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for n, dn in zip(names, dem):
    v = d[n]
    print("%-72s code=%6s vgpr=%4s sgpr=%4s scratch=%4s" % (dn[:72], v.get('codeLenInByte'), v.get('NumVgprs'), v.get('NumSgprs', '?'), v.get('ScratchSize')))
