cd $GRAFT_REPO_ROOT
AB_ROUNDS=2 AB_VERIFY=4 bash tools/ab.sh cur f6 g1 g2 g3
bash tools/kprof.sh cur g1 2>&1 | grep -E "==|k_alloc|k_spec|k_prep|k_poly|k_pack "
