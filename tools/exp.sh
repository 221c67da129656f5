cd $GRAFT_REPO_ROOT
python tools/fuzz_parity.py 1500 31001 2>&1 | tail -3
python tools/fuzz_parity.py --submit 1000 31002 2>&1 | tail -3
python tools/fuzz_parity.py --a1 800 31003 2>&1 | tail -3
python tools/fuzz_mixed.py 60 31004 2>&1 | tail -3
