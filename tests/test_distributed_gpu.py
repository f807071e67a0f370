"""N>1 path with the real HIP engine.

* two processes (gloo rendezvous, both on GPU 0 -- the test box has one GPU, and RCCL refuses two ranks on one device)
  shard the boosting iterations and all-gather the result rows; every rank must end up with exactly what a single
  process computes;
* the RCCL flavour of the very same code (backend "nccl": device tensors, ``all_gather_into_tensor`` over RCCL) with
  as many ranks as the box has GPUs -- one rank on a one-GPU box, which still drives libddx's HIP runtime, torch's
  HIP runtime and RCCL in one process -- and with >= 2 ranks whenever >= 2 GPUs are visible;
* one process driving several GPUs / several streams per GPU (``devices=``, ``streams_per_device=``)."""
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


def _worker(rank, world, port, out_dir, backend="gloo"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from doubletdetection_amd import BoostClassifier

    kw = dict(_KW)
    if backend == "nccl":
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{rank}"))
        kw["device"] = rank
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            clf = BoostClassifier(**kw).fit(_counts())
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


def _compare_with_single(tmp_path, ranks):
    from doubletdetection_amd import BoostClassifier

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        single = BoostClassifier(**_KW).fit(_counts())
        single_labels = single.predict()
    for rank in ranks:
        r = np.load(tmp_path / f"rank{rank}.npz")
        np.testing.assert_array_equal(r["parents"], np.asarray(single.parents_))
        np.testing.assert_array_equal(r["comm"], single.communities_)
        np.testing.assert_array_equal(r["synth"], single.synth_communities_)
        np.testing.assert_array_equal(r["scores"], single.all_scores_)
        np.testing.assert_array_equal(r["logp"], single.all_log_p_values_)
        np.testing.assert_array_equal(r["labels"], single_labels)


def test_rccl_backend_one_rank_per_visible_gpu(tmp_path):
    """backend="nccl" (= RCCL): the result rows travel as device tensors through all_gather_into_tensor.  World size =
    number of visible GPUs (capped at 4), so a one-GPU box still runs the RCCL code path end to end."""
    import torch
    import torch.multiprocessing as mp

    world = max(1, min(4, torch.cuda.device_count()))
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), "nccl"), nprocs=world, join=True)
    _compare_with_single(tmp_path, range(world))


def test_streams_and_devices_of_one_process_equal_single_lane():
    """devices= / streams_per_device=: iterations dealt out over device contexts driven by host threads of ONE process;
    followers take the prologue's result by device-to-device copy.  Results are those of the single-lane run."""
    import torch

    from doubletdetection_amd import BoostClassifier

    counts = _counts()
    kw = {k: v for k, v in _KW.items() if k != "device"}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        base = BoostClassifier(device=0, streams_per_device=1, **kw).fit(counts)
        layouts = [dict(device=0, streams_per_device=2), dict(device=0, streams_per_device=4)]
        if torch.cuda.device_count() >= 2:
            layouts.append(dict(devices=list(range(min(4, torch.cuda.device_count()))), streams_per_device=2))
        for layout in layouts:
            clf = BoostClassifier(**layout, **kw).fit(counts)
            assert clf._lanes_used == min(kw["n_iters"], len(layout.get("devices", [0])) * layout["streams_per_device"])
            for name in ("all_log_p_values_", "all_scores_", "communities_", "synth_communities_"):
                np.testing.assert_array_equal(getattr(clf, name), getattr(base, name))


def test_bench_two_ranks_control_flow(tmp_path):
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one process per rank), with the collectives on
    gloo so that two ranks can share this box's GPU(s): warm-up, timed steps, the instrumented fits that follow on every
    rank, the max-over-ranks reduction and the single JSON line of rank 0 -- no rank may wait for a fit the others skip."""
    import json
    import subprocess
    import sys

    from conftest import ROOT

    env = dict(os.environ, DDX_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--cells", "6000",
           "--genes", "3000", "--density", "0.05", "--iters", "3", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["value"] > 0 and line["scaling"] == "strong"
    assert line["roofline"] is not None and line["roofline"]["frac"] > 0


def test_sleeping_host_waits_give_the_same_fit_for_less_cpu_time(monkeypatch):
    """Option host_wait=block (the library polls an event with sleeps instead of the runtime's spinning hipStreamSynchronize; what the
    classifier selects by itself when several ranks share a node): the same arrays, no more CPU time and no slower.  (How much CPU time
    the sleeping waits save depends on how long the lanes wait: profiles/r06_host_wait_cpu.txt has the headline-size figures.)"""
    import time

    from doubletdetection_amd import BoostClassifier, _lib
    from doubletdetection_amd._synthetic import make_counts

    counts = make_counts(12000, 4000, density=0.08, n_types=6, seed=5)
    kw = dict(n_iters=8, n_top_var_genes=3000, random_state=2, device=0, streams_per_device=4)
    out = {}
    for mode in ("spin", "block"):
        monkeypatch.setitem(_lib.OPTIONS, "host_wait", mode)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            BoostClassifier(**kw).fit(counts)                      # (warm: contexts, pinned buffers)
            c0, t0 = time.process_time(), time.perf_counter()
            clf = BoostClassifier(**kw).fit(counts)
            out[mode] = (clf, time.process_time() - c0, time.perf_counter() - t0)
    print({m: (round(c, 3), round(t, 3)) for m, (_, c, t) in out.items()})
    for name in ("all_log_p_values_", "all_scores_", "communities_", "synth_communities_"):
        np.testing.assert_array_equal(getattr(out["block"][0], name), getattr(out["spin"][0], name))
    assert out["block"][1] < 1.15 * out["spin"][1] + 0.02, out
    assert out["block"][2] < 1.5 * out["spin"][2], out


def _share_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    import time

    import torch.distributed as dist

    from doubletdetection_amd import BoostClassifier
    from doubletdetection_amd._synthetic import make_counts

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        counts = make_counts(12000, 3000, density=0.06, n_types=6, seed=31)
        forms, cpu = [], []
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for _ in range(3):
                c0 = time.process_time()
                clf = BoostClassifier(n_iters=4, n_top_var_genes=2000, random_state=1, device=0).fit(counts)
                cpu.append(time.process_time() - c0)
                forms.append(clf._upload_form_used)
        np.savez(os.path.join(out_dir, f"share{rank}.npz"), scores=clf.all_scores_, comm=clf.communities_, forms=np.asarray(forms), cpu=np.asarray(cpu))
    finally:
        dist.destroy_process_group()


def test_ranks_of_a_node_share_one_packing(tmp_path):
    """One process per GPU (dd.py:149-160 x ranks): local rank 0 packs the matrix once into a POSIX shared-memory image, the other ranks of
    the node send its chunks to their own GPU (ddx_get_upload_form = 2) -- here two gloo ranks on the one GPU of the test box, three fits
    in a row (the segment is reused generation by generation).  Results are those of a single process."""
    import torch.multiprocessing as mp

    from doubletdetection_amd import BoostClassifier
    from doubletdetection_amd._synthetic import make_counts

    mp.spawn(_share_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "share0.npz"), np.load(tmp_path / "share1.npz")
    print("upload forms: rank 0", r0["forms"], "rank 1", r1["forms"], " cpu s per fit:", r0["cpu"].round(2), r1["cpu"].round(2))
    assert list(r0["forms"]) == [1, 1, 1], r0["forms"]
    assert list(r1["forms"]) == [2, 2, 2], r1["forms"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        single = BoostClassifier(n_iters=4, n_top_var_genes=2000, random_state=1, device=0).fit(make_counts(12000, 3000, density=0.06, n_types=6, seed=31))
    for r in (r0, r1):
        np.testing.assert_array_equal(r["comm"], single.communities_)
        np.testing.assert_array_equal(r["scores"], single.all_scores_)
