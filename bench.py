#!/usr/bin/env python3
"""Headline benchmark: cell-updates/s of the C-grid Matsuno step + fraction of the
HBM roofline (BASELINE.json).  One "step" = one full Matsuno step (predictor +
corrector, every prognostic field read once and written once) over the whole grid.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2]

N > 1 is launched by torch.distributed.run, one rank per GPU; the grid is split
into latitude bands (strong scaling: the global grid is fixed).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (description, H, W, model, tracer, bytes per cell-update = 2 * fields * 8)
    "c3": ("2-D shallow water + theta + viscosity + van-Leer tracer, 4096x2048 fp64 "
           "(BASELINE configs[2])", 2048, 4096, "SW2D_TEMP", "van_leer", 80.0),
    "c2": ("2-D shallow water Matsuno C-grid, 720x360 fp64 (BASELINE configs[1])",
           360, 720, "SW2D", None, 48.0),
    # cells are (k, j, i) points; 4 3-D fields + p: (64 + 16/L) bytes per cell-update
    "c4": ("2.5-D sigma-level primitive equations, 1440x720x24 fp64 (BASELINE configs[3])",
           720, 1440, "PE25D", None, 64.0 + 16.0 / 24),
}
LAYERS = {"c4": 24}


def synth(name, H, W, row0=0, nrows=None, geom=None):
    """SURVEY.md 8d synthetic inputs, seed default_rng(0); rows [row0,row0+nrows) only."""
    rng = np.random.default_rng(0)
    nrows = H if nrows is None else nrows
    if name == "c4":
        L = LAYERS[name]
        sl = slice(row0, row0 + nrows)
        p = 1e5 + 10 * rng.standard_normal((H, W))
        u = rng.standard_normal((L, H, W))
        v = rng.standard_normal((L, H, W))
        v[:, -1, :] = 0
        tt = 300 + rng.standard_normal((L, H, W))
        tp = p * np.asarray(geom.sig) + geom.ptop
        t = tt * ((1e5 / tp) ** (287.0 / 1004.0))          # to_potential_temp, temperature.py:15-19
        q = 3e-6 * (1 + 0.1 * rng.random((L, H, W)))
        return dict(p=p[sl], u=u[:, sl], v=v[:, sl], t=t[:, sl], q=q[:, sl])
    u = rng.standard_normal((H, W))
    v = rng.standard_normal((H, W))
    sl = slice(row0, row0 + nrows)
    if name == "c2":
        p = 8000 + rng.standard_normal((H, W))
        return dict(u=u[sl], v=v[sl], p=p[sl])
    p = 101325 + rng.standard_normal((H, W))
    t = 273.16 + rng.standard_normal((H, W))
    q = rng.random((H, W))
    return dict(u=u[sl], v=v[sl], p=p[sl], t=t[sl], q=q[sl])


def cpu_baseline(name, H, W):
    """the oracle (NumPy restatement, bit-identical to the reference) on the host, 1 core"""
    from oracle import sw2d, sw2d_temp, tracer
    if name == "c4":
        # bounded sample: the same recipe on a 360x180x24 grid (1/16 of the cells), 2 steps
        from oracle import dynamics, geometry as ogeo
        h, w, L = 180, 360, LAYERS[name]
        og = ogeo.gen_geometry(h, w, L, sig_func=ogeo.manabe_sig)
        s = synth(name, h, w, geom=og)
        st = (s["p"], s["u"], s["v"], s["t"], s["q"])
        t0 = time.perf_counter()
        for _ in range(2):
            st = dynamics.matsuno_timestep(*st, 1.0, og)
        el = time.perf_counter() - t0
        return {"value": h * w * L * 2 / el, "unit": "cell-updates/s", "cores": 1, "kind": "port",
                "sample": "2 steps of a 360x180x24 grid (1/16 of the cells, same recipe) with the NumPy "
                          "oracle, %.1f s; host has %d cores" % (el, os.cpu_count())}
    s = synth(name, H, W)
    dx, dt = 300e3, 300.0
    t0 = time.perf_counter()
    if name == "c2":
        st = (s["u"], s["v"], s["p"])
        nst = 20
        for _ in range(nst):
            st = sw2d.matsumo_scheme(*st, dx, dt)
    else:
        nst = 1
        V = np.stack([s["v"], s["u"]])
        tracer.limited_advection(dt, (dx, dx), V, s["q"])
        sw2d_temp.matsumo_scheme(s["u"], s["v"], s["p"], s["t"], dx, dt)
    el = time.perf_counter() - t0
    return {"value": H * W * nst / el, "unit": "cell-updates/s", "cores": 1, "kind": "port",
            "sample": "%d full step(s) of the %dx%d grid with the NumPy oracle, %.1f s; host has %d cores"
                      % (nst, W, H, el, os.cpu_count())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--variant", default="fused", choices=["fused", "staged"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    a = ap.parse_args()

    import torch
    import gcmiipy_amd as g
    from gcmiipy_amd import _lib
    from gcmiipy_amd.bands import BandRunner, HipBandEngine, split_rows

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)"
                         % (a.gpus, world))
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    desc, H, W, model, tracer, bpc = WORKLOADS[a.workload]
    row0, nrows = split_rows(H, world)[rank]
    dx, dt = 300e3, 300.0
    geom, L = None, 1
    if model == "PE25D":
        from gcmiipy_amd import geometry
        L, dt = LAYERS[a.workload], 1.0
        geom = geometry.gen_geometry(H, W, L, sig_func=geometry.manabe_sig)
    core = g.Core(getattr(_lib, model), W, nrows, L, dx=dx, geom=geom,
                  filter=not os.environ.get("GCM_BENCH_NOFILTER"),   # diagnostic only
                  tracer={None: _lib.TRACER_NONE, "van_leer": _lib.TRACER_VANLEER}[tracer],
                  variant=_lib.VARIANT_FUSED if a.variant == "fused" else _lib.VARIANT_STAGED,
                  nranks=world, rank=rank, global_height=H, row0=row0, device=local,
                  stream=torch.cuda.current_stream().cuda_stream)
    core.set_state(**synth(a.workload, H, W, row0, nrows, geom))
    runner = BandRunner(HipBandEngine(core, torch) if world > 1 else None, rank, world, dist)

    region = {}

    def run(n, timed=False):
        if world == 1:
            if timed:   # same launches, bracketed by HIP events on the launch stream
                region["ms"], _ = core.time_steps(n, dt, per_kernel=False)
            else:
                core.step(n, dt)
        else:
            for _ in range(n):
                runner.step(dt)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    run(a.warmup)
    fence()
    t0 = time.perf_counter()
    run(a.steps, timed=True)
    fence()
    el = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([el], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    assert core.diag(_lib.DIAG_ANY_NAN) == 0.0, "state went NaN during the timed run"

    out = None
    if rank == 0:
        cells = H * W * L
        value = cells * a.steps / el
        out = {
            "metric": "cell-updates/s (C-grid Matsuno step)", "value": value,
            "unit": "cell-updates/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": el / a.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc, "grid": [W, H], "variant": a.variant,
                       "decomposition": "%d latitude band(s)" % world,
                       "bytes_per_cell_update": bpc},
            "hbm_roofline_frac_whole_job": value * bpc / (world * HBM_PEAK_GBS * 1e9),
        }
    if world == 1:
        # dominant kernel.  fused: one launch per step, so its average duration over the timed
        # region is (HIP-event time of the region on the launch stream) / launches; the
        # back-to-back launches leave no gap (rocprofv3 trace: next start == previous end).
        # "kernel_ms_isolated" is a second pass with an event pair around every launch
        # (idle gaps between launches let the chip clock higher, so it reads lower).
        _, kiso = core.time_steps(min(a.steps, 50), dt)
        launches = 1
        if model == "PE25D":
            # dominant kernel = pe_update_kernel, launched twice per step (once per Euler stage);
            # its algorithmic bytes per launch are half of the step's
            kname, kms, launches = "pe_update_kernel", kiso, 2
        elif a.variant == "fused":
            kname, kms = "sw2d_fused_kernel", region["ms"] / a.steps
        else:
            kname, kms = "sw2d_stage_kernel (corrector stage)", kiso
        ach = H * W * L * bpc / launches / (kms * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": ach / HBM_PEAK_GBS, "traffic": None, "kernel": kname,
                           "kernel_ms": kms, "kernel_ms_isolated": kiso,
                           "algorithmic_bytes_per_launch": H * W * L * bpc / launches}
        out["cpu_baseline"] = None if a.no_cpu else cpu_baseline(a.workload, H, W)
    core.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
