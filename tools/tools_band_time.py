#!/usr/bin/env python3
"""Compute-side strong-scaling bound, measured on ONE GPU.

For N in --splits, builds the band a middle rank of an N-way latitude split would own
(bench.py's workloads and synthetic state), and steps it with exactly the launches the band
engine issues (gcmiipy_amd.bands.HipBandEngine: edge-first phases for GCM_PE25D, deep-halo
windows for the 2-D models), with the ghost-row exchange replaced by a device-local copy
of the band's own edge rows (send buffer -> receive buffer on the second stream).  The numbers
say how far the kernels alone let the step shrink when the grid is split N ways -- the exchange
over xGMI then has to hide behind them.  The 8-GPU runs themselves are the driver's.

  python tools/tools_band_time.py [--workload c4|c3] [--splits 1,2,4,8] [--steps 20]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c4")
    ap.add_argument("--splits", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--out", default=None)
    ap.add_argument("--exchange", default="local", choices=["local", "rccl", "torch-nccl"],
                    help="local: device copy in place of the exchange; rccl: the band sends to itself through "
                         "gcmiipy_amd.rccl (RCCL called directly: the production path, no xGMI); torch-nccl: the "
                         "same through torch.distributed's batch_isend_irecv")
    a = ap.parse_args()
    import torch
    import bench
    import gcmiipy_amd as g
    from gcmiipy_amd import _lib, geometry
    from gcmiipy_amd.bands import BandRunner, HipBandEngine, LoopbackExchange as LoopbackDist, split_rows
    torch.cuda.set_device(0)
    tdist = None
    if a.exchange == "torch-nccl":
        import torch.distributed as tdist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        tdist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    elif a.exchange == "rccl":
        from gcmiipy_amd.rccl import RcclP2P
        tdist = RcclP2P(None, 0, 1)
    desc, H, W, L, model, tracer, bpc, dt = bench.WORKLOADS[a.workload]
    # a stream of its own, as bench.py's bands use
    torch.cuda.set_stream(torch.cuda.Stream())
    geom = geometry.gen_geometry(H, W, L, sig_func=geometry.manabe_sig) if model == "PE25D" else None
    full = bench.synth(a.workload, H, W, L, geom=geom)
    res = []
    for n in [int(x) for x in a.splits.split(",")]:
        rank = n // 2
        row0, nrows = split_rows(H, n)[rank]
        k = bench.halo_steps_for(a.workload, n)
        core = g.Core(getattr(_lib, model), W, nrows, L, dx=bench.DX, geom=geom,
                      tracer={None: _lib.TRACER_NONE, "van_leer": _lib.TRACER_VANLEER}[tracer],
                      nranks=n, rank=rank, global_height=H, row0=row0,
                      stream=torch.cuda.current_stream().cuda_stream, halo_steps=k,
                      dtype="f32" if a.workload.endswith("_f32") else "f64")
        sl = slice(row0, row0 + nrows)
        core.set_state(**{f: (x[sl] if x.ndim == 2 else x[:, sl]) for f, x in full.items()})
        if model != "PE25D":
            core.snapshot()
        eng = HipBandEngine(core, torch) if n > 1 else None
        if "phys" in a.workload:
            # BASELINE configs[4]: dynamics + solar_timestep every step (gcm_set_physics); on a band the ghost rows are
            # radiated locally inside gcm_band_run
            import numpy as np
            core.set_ground(np.full((nrows, W), 288.0))
            (eng if eng is not None else core).set_physics(geom, bench.PHYS_UTC0)
        # the exchange goes to the band itself (one-rank communicator / device-local copy): neighbours = rank 0
        runner = BandRunner(eng, rank, n, (tdist if tdist is not None else LoopbackDist()) if n > 1 else None,
                            north=0 if n > 1 else None, south=0 if n > 1 else None)
        if n == 1:
            runner.e = type("E", (), {"step_all": staticmethod(lambda dt_: core.step(1, dt_))})()
        import time
        runner.run(max(2 * k, 4), dt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        runner.run(8 * k, dt)
        torch.cuda.synchronize()
        per = (time.perf_counter() - t0) / (8 * k)
        # short bands finish a few steps in milliseconds: run long enough (0.3 s untimed, then at
        # least 0.3 s timed) for the clocks to settle, or the numbers depend on what ran before
        pre = int(0.3 / per) // k * k
        steps = max(a.steps, int(0.3 / per))
        steps = (steps + k - 1) // k * k
        if model != "PE25D":
            # the 2-D noise state only lives for a few hundred steps: pre-run in pieces that go back
            # to the initial state (gcm_restore), and time at most 320 steps
            steps = min(steps, 320 // k * k)
            core.restore()
            runner.count = 0
            done = 0
            while done < pre:
                runner.run(320 // k * k, dt)
                core.restore()
                done += 320 // k * k
        else:
            runner.run(pre, dt)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        t0 = time.perf_counter()
        runner.run(steps, dt)
        host_ms = (time.perf_counter() - t0) * 1e3 / steps      # time the host needs to queue a step
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        assert core.diag(_lib.DIAG_ANY_NAN) == 0.0, "state went NaN: timing invalid"
        # What the HOST needs to queue a step, measured where no call can block on a full queue: the steps
        # are queued behind one long spin kernel on the compute stream (torch.cuda._sleep, ~40 ms), i.e. into
        # a device that executes none of them before the host is done.  `host_ms_per_step` above is taken
        # on a busy device and also counts the time the host is held back by the queue (it tracks the band's
        # GPU time); this figure does not depend on the band's size.
        nq = max(k, 4) // k * k
        torch.cuda.synchronize()
        torch.cuda._sleep(int(40e-3 * 2.0e9))
        t0 = time.perf_counter()
        runner.run(nq, dt)
        host_idle_ms = (time.perf_counter() - t0) * 1e3 / nq
        torch.cuda.synchronize()
        res.append({"split": n, "band_rows": nrows, "halo_steps": k, "ms_per_step": ms, "host_ms_per_step": host_ms,
                    "host_queue_ms_per_step_idle_device": host_idle_ms,
                    "one_library_call_per_run": bool(getattr(runner, "native", False))})
        print("N=%d  band of %4d rows  %.4f ms/step  (host queues a step in %.4f ms behind a busy queue, %.4f ms into an idle one)"
              % (n, nrows, ms, host_ms, host_idle_ms), flush=True)
        core.close()
    base = res[0]["ms_per_step"] if res and res[0]["split"] == 1 else None
    for r in res:
        if base:
            r["compute_bound_speedup"] = base / r["ms_per_step"]
    doc = {"workload": desc, "exchange": a.exchange,
           "note": "one band of an N-way split stepped on one GPU, exchange replaced by a device-local copy "
                   "(local) or sent to itself through RCCL (rccl): the kernels' own strong-scaling bound", "bands": res}
    print(json.dumps(doc))
    if a.out:
        with open(os.path.join(ROOT, a.out), "w") as f:
            f.write(json.dumps(doc, indent=1) + "\n")


if __name__ == "__main__":
    main()
