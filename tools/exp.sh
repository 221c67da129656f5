cd $GRAFT_REPO_ROOT
python tools/fuzz_cli.py --batch 60 11 2>&1 | tail -12
python tools/fuzz_cli.py 300 12 2>&1 | tail -12
