"""bench.py's launch contract, checked without a GPU: `--gpus N` outside torchrun starts N ranks itself; a rank count
that does not match --gpus, or too few visible devices, stops the run instead of benchmarking one rank silently."""
import importlib.util
import os
import sys

import pytest

from conftest import ROOT


def _load_bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_self_spawn_builds_the_torchrun_command(monkeypatch):
    import subprocess

    bench = _load_bench()
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    assert bench.spawn_ranks(4) == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 0 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


@pytest.mark.parametrize("world,gpus", [("1", "2"), ("4", "1")])
def test_rank_count_must_match_gpus(monkeypatch, world, gpus):
    bench = _load_bench()
    monkeypatch.setenv("WORLD_SIZE", world)
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("LOCAL_RANK", "0")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", gpus])
    with pytest.raises(SystemExit, match="every GPU needs its own rank"):
        bench.main()


def test_too_few_devices_stop_the_run(monkeypatch):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    bench = _load_bench()
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "1"])
    with pytest.raises(SystemExit, match="device"):
        bench.main()
