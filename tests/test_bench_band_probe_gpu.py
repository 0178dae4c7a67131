"""bench.py's multi-rank path on the one GPU of the test box: run_workload for a 2-way split of the
C3 grid with the loopback exchange standing in for the RCCL ring and a one-rank stand-in for
torch.distributed -- the band runs through gcm_band_run (deep halo, k = 4), both exchange sequences
are timed (gcm_set_band_overlap) and one is kept, exactly the code the driver's --gpus N run executes."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _OneRank:
    """the collectives bench.py issues, for a world of one process"""
    class ReduceOp:
        MAX = "max"

    def barrier(self):
        pass

    def all_reduce(self, t, op=None):
        return t

    def all_gather(self, out, t):
        for o in out:
            o.copy_(t)


def test_run_workload_band_path_and_overlap_probe(monkeypatch):
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from gcmiipy_amd.bands import LoopbackExchange
    monkeypatch.setattr(bench, "SETTLE_S", 0.01)
    monkeypatch.setattr(bench, "MIN_TIMED_S", 0.02)
    cx = bench.Ctx()
    cx.torch, cx.dist = torch, _OneRank()
    cx.rank, cx.world, cx.local, cx.backend = 0, 2, 0, "nccl"
    cx.ring, cx.exchange, cx.stuck, cx.exchange_fallback = LoopbackExchange(), "loopback (test)", False, None
    torch.cuda.set_device(0)
    with torch.cuda.stream(torch.cuda.Stream()):
        res = bench.run_workload(cx, "c3", 8, 4, want_kernel=False)
    assert res["n_gpus"] == 2 and res["ms_per_step"] > 0 and np.isfinite(res["value"])
    probe = res["band_overlap_probe_ms_per_step"]
    assert probe["chosen"] in ("plain", "overlap") and probe["plain"] > 0 and probe["overlap"] > 0
    assert "diagnostics" in res and res["diagnostics"]["band_ms_per_step_local_exchange"] > 0
    # what makes the N > 1 line readable on its own
    assert res["roofline"]["peak"] == 2 * 8000.0 and 0 < res["roofline"]["frac"] < 1
    assert len(res["band_ms_per_step"]["per_rank"]) == 2 and res["band_ms_per_step"]["max"] > 0
    assert res["exchange_bytes"]["message_bytes"] == 5 * 8 * 4096 * 8 and res["exchange_bytes"]["exchanges_per_step"] == 0.25
    assert "rccl_ranks" in res


def test_run_workload_pe25d_band_path(monkeypatch):
    """the same for the 2.5-D workload: an 8-way split's band (90 rows: 3-row K4 groups, edge rows in
    level segments, exchange behind the pack on the second stream) through bench.run_workload"""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from gcmiipy_amd.bands import LoopbackExchange
    monkeypatch.setattr(bench, "SETTLE_S", 0.01)
    monkeypatch.setattr(bench, "MIN_TIMED_S", 0.02)
    cx = bench.Ctx()
    cx.torch, cx.dist = torch, _OneRank()
    cx.rank, cx.world, cx.local, cx.backend = 3, 8, 0, "nccl"
    cx.ring, cx.exchange, cx.stuck, cx.exchange_fallback = LoopbackExchange(), "loopback (test)", False, None
    torch.cuda.set_device(0)
    with torch.cuda.stream(torch.cuda.Stream()):
        res = bench.run_workload(cx, "c4", 4, 2, want_kernel=False)
    assert res["n_gpus"] == 8 and res["ms_per_step"] > 0 and np.isfinite(res["value"])
    probe = res["band_overlap_probe_ms_per_step"]           # GCM_PE25D: the edge rows' update dispatched first, or not
    assert probe["chosen"] in ("plain", "overlap") and probe["plain"] > 0 and probe["overlap"] > 0


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` as the driver types it, with no launcher and no WORLD_SIZE in the
    environment: the parent starts the two ranks itself (a child torch.distributed.run job; gloo here, since
    RCCL refuses two ranks on one device), rank 0 prints ONE JSON line with n_gpus == 2, the exit code is the
    job's"""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(GCM_BENCH_BACKEND="gloo", GCM_BENCH_SETTLE_S="0.01", GCM_BENCH_MIN_TIMED_S="0.02")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "4",
                        "--only", "--no-cpu"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "strong"
    assert "exchange" in d and "exchange_fallback" in d
    assert d["diagnostics"]["host_queue_ms_per_step_idle_device"] > 0
