"""CPU checks of bench.py's plumbing: the self-launch of the N > 1 ranks (no GPU, nothing is started: the
child job is intercepted) and the workload table."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launch_ranks_builds_a_child_torchrun_job(monkeypatch):
    sys.path.insert(0, ROOT)
    import subprocess
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"])
    assert bench.launch_ranks(4) == 7                      # the job's exit code is handed back
    cmd = seen["cmd"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"]      # arguments passed through
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_main_self_launches_only_without_a_launcher(monkeypatch):
    """--gpus N > 1 with WORLD_SIZE unset -> launch_ranks (before torch is imported); with WORLD_SIZE set the
    process is a rank and must not spawn anything"""
    sys.path.insert(0, ROOT)
    import pytest
    import bench
    called = []
    monkeypatch.setattr(bench, "launch_ranks", lambda n: called.append(n) or 0)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and called == [2]


def test_workload_table_matches_baseline_configs():
    sys.path.insert(0, ROOT)
    import bench
    w = bench.WORKLOADS
    assert (w["c2"][1], w["c2"][2]) == (360, 720) and (w["c3"][1], w["c3"][2]) == (2048, 4096)
    assert (w["c4"][1], w["c4"][2], w["c4"][3]) == (720, 1440, 24)
    assert (w["c5_phys"][1], w["c5_phys"][2], w["c5_phys"][3]) == (1440, 2880, 40)
    assert set(bench.ALSO) >= {"c2", "c3", "c4", "c4_f32", "c5_phys"}
    assert w["c3"][6] == 80.0 and abs(w["c4"][6] - (64 + 16 / 24)) < 1e-12     # algorithmic bytes per cell-update
