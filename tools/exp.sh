cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -q -x -k "cli_whole_file or reference_cli_source" 2>&1 | tail -4
python tools/fuzz_cli.py 400 2 2>&1 | tail -30
