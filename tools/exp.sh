cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -8
echo "== a1 fuzz"; python tools/fuzz_parity.py --a1 400 9001 2>&1 | tail -5
