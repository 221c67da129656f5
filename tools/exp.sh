cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -4
python tools/fuzz_parity.py 600 41001 2>&1 | tail -3
python tools/fuzz_parity.py --submit 400 41002 2>&1 | tail -3
