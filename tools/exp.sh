cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -q -x -k "batch_bitstream or every_stage or stress or random_conf or mpeg2_batch or mono_batch or golden" 2>&1 | tail -3
bash tools/kprof.sh cur new 2>&1 | grep -E "==|k_alloc|k_spec|k_prep|k_poly|k_pack "
AB_ROUNDS=2 AB_VERIFY=4 bash tools/ab.sh cur new
AB_ROUNDS=1 AB_ARGS="--config 3 --steps 6 --warmup 1" bash tools/ab.sh cur new
