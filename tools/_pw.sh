python bench.py --steps 150 --warmup 3 --no-cpu-baseline > /tmp/b.log 2>&1 &
BP=$!
while kill -0 $BP 2>/dev/null; do rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -E "Package Power|sclk|junction" | sed 's/.*: //' | tr '\n' ' '; echo; done > /tmp/pw.log
wait $BP
grep -v "(9[0-9]Mhz)\|(1[0-9][0-9]Mhz)" /tmp/pw.log | head -40
tail -1 /tmp/b.log | cut -c1-200
