cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lognorm or mirror or scaled or pca or arpack or fit_matches" 2>&1 | tail -3
timeout 100 python profiles/tools/mirror_time.py 2>&1 | tail -1
