#!/bin/bash
out=gpurun_out; mkdir -p $out
for o in "knn_emit_rt=2" "knn_emit_rt=4" "knn_emit_rt=2" "knn_emit_rt=4"; do
  echo "== $o"; DDX_OPTIONS=$o python profiles/tools/spmm_time.py $o 2>&1 | tail -1 | python -c "
import sys,ast
l=sys.stdin.read().strip(); i=l.index('{'); d=ast.literal_eval(l[i:]); print({k:v for k,v in d.items() if k.startswith('knn')})"
done 2>&1 | tee $out/r06l_knn_emit_rt.txt
DDX_OPTIONS=knn_emit_rt=4 timeout 900 python -m pytest tests/test_gpu_knn_fullsize.py tests/test_gpu_knn_adversarial.py -x -q 2>&1 | tail -3
# configs[3]-like size: 625 k points
python - <<'PY' 2>&1 | tee -a gpurun_out/r06l_knn_emit_rt.txt
import numpy as np
from doubletdetection_amd import _lib
M = 625000
rng = np.random.default_rng(0)
centers = rng.normal(size=(40, 30)) * 6
emb = (centers[rng.integers(0, 40, M)] + rng.normal(size=(M, 30))).astype(np.float32)
for rt in ("2", "4", "2", "4"):
    _lib.OPTIONS["knn_emit_rt"] = rt
    ctx = _lib.Context(0)
    ctx.timing_enable(True)
    ctx.set_embedding(emb)
    for rep in range(2):
        ctx.timing_reset(); ctx.knn(30, False); ctx.synchronize()
    t = ctx.timings()
    print("625k points, knn_emit_rt", rt, {k: round(v[1], 3) for k, v in t.items() if k in ("knn_emit", "knn_bound", "knn_select", "knn_lists")})
    ctx.close()
PY
