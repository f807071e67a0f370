"""N>1 path with the real HIP engine: two processes (gloo rendezvous, both on GPU 0 -- the test box has one GPU,
and RCCL refuses two ranks on one device) shard the boosting iterations and all-gather the result rows; every rank
must end up with exactly what a single process computes.  The nccl/RCCL flavour of the same code runs in
``bench.py --gpus N`` (one process per GPU)."""
import os
import socket
import sys
import warnings

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _counts():
    from doubletdetection_amd._synthetic import make_counts

    return make_counts(1500, 900, density=0.12, n_types=5, seed=77)


_KW = dict(n_iters=5, n_top_var_genes=700, random_state=3, device=0)


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from doubletdetection_amd import BoostClassifier

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            clf = BoostClassifier(**_KW).fit(_counts())
            labels = clf.predict()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), scores=clf.all_scores_, logp=clf.all_log_p_values_,
                 comm=clf.communities_, synth=clf.synth_communities_, parents=np.asarray(clf.parents_), labels=labels)
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_the_hip_engine_equal_one_process(tmp_path):
    import torch.multiprocessing as mp

    from doubletdetection_amd import BoostClassifier

    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        single = BoostClassifier(**_KW).fit(_counts())
        single_labels = single.predict()
    for rank in (0, 1):
        r = np.load(tmp_path / f"rank{rank}.npz")
        np.testing.assert_array_equal(r["parents"], np.asarray(single.parents_))
        np.testing.assert_array_equal(r["comm"], single.communities_)
        np.testing.assert_array_equal(r["synth"], single.synth_communities_)
        np.testing.assert_array_equal(r["scores"], single.all_scores_)
        np.testing.assert_array_equal(r["logp"], single.all_log_p_values_)
        np.testing.assert_array_equal(r["labels"], single_labels)
