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
    # the launcher chooses and holds the rendezvous port itself (no probe-then-bind window), on 127.0.0.1
    assert "--rdzv-backend=c10d" in cmd and "--rdzv-endpoint=127.0.0.1:0" in cmd
    assert cmd[cmd.index("--local-addr") + 1] == "127.0.0.1" and not any(c.startswith("--master-port") for c in cmd)
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


# ---------------------------------------------------------------- bring_up_direct_rccl: the three failure branches
# A stand-in for gcmiipy_amd.rccl.RcclP2P over a world-2 gloo group on the CPU.  Each rank runs bench.py's own
# bring_up_direct_rccl + agree_stuck; what must hold: both ranks reach the same verdict (nobody is left in a
# collective the other never enters), the reason is reported, and a stuck bring-up thread makes the job exit non-zero.
class _FakeRccl:
    mode = "ok"

    @staticmethod
    def new_unique_id():
        if _FakeRccl.mode == "nolib":
            raise OSError("librccl.so: cannot open shared object file")
        return bytes(range(128))

    def __init__(self, bootstrap, rank, world, uid_bytes=None):
        import time
        assert uid_bytes == bytes(range(128))              # the id arrived on every rank, NUL byte included
        self.rank, self.world = rank, world
        if _FakeRccl.mode == "init_raises" and rank == 1:
            raise RuntimeError("ncclCommInitRank: unhandled system error")
        if _FakeRccl.mode == "init_hangs" and rank == 1:
            time.sleep(3600)

    def self_check(self):
        pass

    def count(self):
        return self.world


def _bring_up_worker(rank, world, mode, outdir):
    import json
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import bench
    dist.init_process_group("gloo", init_method="file://" + os.path.join(outdir, "rdzv"), rank=rank, world_size=world)
    cx = bench.Ctx()
    cx.rank, cx.world, cx.local, cx.backend = rank, world, 0, "gloo"
    cx.ring, cx.exchange, cx.stuck, cx.exchange_fallback = dist, "torch.distributed", False, None
    cx.cpu_group = dist.new_group(backend="gloo")
    _FakeRccl.mode = mode
    if mode == "nolib" and rank == 0:
        pass                                               # (rank 0 is the one that asks for the id)
    bench.bring_up_direct_rccl(cx, torch, dist, rccl_cls=_FakeRccl, device="cpu", timeout_s=1.5)
    bench.agree_stuck(cx, torch, dist)
    json.dump({"stuck": cx.stuck, "fallback": cx.exchange_fallback, "direct": isinstance(cx.ring, _FakeRccl),
               "exchange": cx.exchange}, open(os.path.join(outdir, "r%d.json" % rank), "w"))
    dist.barrier()
    if cx.stuck:
        bench.finish(cx, {"exchange_fallback": cx.exchange_fallback})     # os._exit(4) from this very process
    dist.destroy_process_group()


def _run_bring_up(tmp_path, mode):
    """two fresh children (spawn: nothing exec'ed over a live process); -> (exit codes, per-rank verdicts)"""
    import json
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_bring_up_worker, args=(r, 2, mode, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert not p.is_alive(), "a rank was left waiting in a collective"
    return [p.exitcode for p in procs], [json.load(open(os.path.join(str(tmp_path), "r%d.json" % r))) for r in range(2)]


def test_bring_up_ok(tmp_path):
    codes, v = _run_bring_up(tmp_path, "ok")
    assert codes == [0, 0]
    assert all(x["direct"] and x["fallback"] is None and not x["stuck"] for x in v)


def test_bring_up_without_librccl(tmp_path):
    codes, v = _run_bring_up(tmp_path, "nolib")
    assert codes == [0, 0]                                 # a clean fallback: the run goes on over torch.distributed
    assert all((not x["direct"]) and not x["stuck"] for x in v)
    assert "librccl" in v[0]["fallback"] and v[1]["fallback"]        # the failing rank names the cause, the other knows


def test_bring_up_init_raises_on_one_rank(tmp_path):
    codes, v = _run_bring_up(tmp_path, "init_raises")
    assert codes == [0, 0]
    assert all((not x["direct"]) and not x["stuck"] for x in v)
    assert "unhandled system error" in v[1]["fallback"] and "another rank" in v[0]["fallback"]


def test_bring_up_init_never_returns(tmp_path):
    """the stuck branch: the watchdog gives up after its timeout, BOTH ranks learn of it (all-reduced), the JSON line
    carries the reason and every rank exits non-zero -- from the rank process itself, no re-exec"""
    codes, v = _run_bring_up(tmp_path, "init_hangs")
    assert codes == [4, 4], codes
    assert all(x["stuck"] and not x["direct"] for x in v)
    assert "timed out" in v[1]["fallback"]
