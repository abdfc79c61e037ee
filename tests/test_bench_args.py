"""CPU: bench.py's argument handling (`--gpus N` must never silently report a 1-GPU run as N GPUs)."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def bench(monkeypatch):
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_world_mismatch_is_an_error(bench):
    bench.check_world(1, 1)
    bench.check_world(8, 8)
    for gpus, world in ((2, 1), (1, 2), (8, 4)):
        with pytest.raises(SystemExit) as e:
            bench.check_world(gpus, world)
        assert str(gpus) in str(e.value) and "WORLD_SIZE" in str(e.value)


def test_gpus_without_launcher_starts_torch_distributed_run(bench, monkeypatch):
    """`python bench.py --gpus 4` with no WORLD_SIZE: the launcher is started as a CHILD process (no exec, nothing has
    touched the GPU yet) with one rank per GPU on 127.0.0.1, the flags are passed through and its exit code is relayed."""
    seen = {}

    class R:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return R()

    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" or "HSA_ENABLE_IPC_MODE_LEGACY" in os.environ


def test_config5_refuses_multiple_gpus(bench, monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--config", "5"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "replicas only" in str(e.value)


def test_traffic_is_quoted_only_for_the_measured_build(bench, tmp_path, monkeypatch):
    import json
    from ishara_amd.build import source_hash
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    val, why = bench.committed_traffic("k", "cfg2")
    assert val is None and "no committed" in why
    json.dump(dict(source_hash="deadbeef", kernels={"k": {"hbm_bytes_per_launch": 5.0}}), open(tmp_path / "profiles" / "r2_traffic_cfg2.json", "w"))
    val, why = bench.committed_traffic("k", "cfg2")
    assert val is None and "another build" in why
    json.dump(dict(source_hash=source_hash(), kernels={"k": {"hbm_bytes_per_launch": 5.0}}), open(tmp_path / "profiles" / "r2_traffic_cfg2.json", "w"))
    val, why = bench.committed_traffic("k", "cfg2")
    assert val == 5.0 and "not live" in why
    assert bench.committed_traffic("other", "cfg2")[0] is None
