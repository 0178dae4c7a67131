#!/usr/bin/env python3
"""Headline benchmark: cell-updates/s of the C-grid Matsuno step + fraction of the
HBM roofline (BASELINE.json).  One "step" = one full Matsuno step (predictor +
corrector, every prognostic field read once and written once) over the whole grid.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c4|c5|c5_phys...] [--only]

The timed region is a block of EXACTLY K steps between barrier + synchronize fences; the block is
repeated until at least MIN_TIMED_S seconds have been timed and the MEDIAN block is reported
(`ms_per_step`, `value`), with the fastest and slowest block beside it (`ms_per_step_min/max`).

The headline workload is c3 (BASELINE configs[2], 4096x2048 shallow water + theta +
viscosity + van-Leer tracer, fp64).  N > 1 is launched by torch.distributed.run, one rank
per GPU; the FIXED global grid is split into latitude bands (strong scaling).  Unless --only
is given the JSON line also carries, under "also", short runs of the other workloads -- at
N > 1 that includes the 1440x720x24 primitive-equation workload c4 with a same-run 1-GPU
reference on rank 0, the strong-scaling pair BASELINE.json's north_star names.
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
L2_PEAK_GBS = 34500.0     # aggregate L2 read rate, 8 XCDs x 4 MiB (MI355X_MICROARCH.md, L2 section)
L2_BYTES = 32 * 2 ** 20
SETTLE_S = float(os.environ.get("GCM_BENCH_SETTLE_S", "0.15"))   # untimed pre-conditioning before warm-up (see run_workload)
MIN_TIMED_S = float(os.environ.get("GCM_BENCH_MIN_TIMED_S", "0.5"))   # the K-step block is repeated until this much is timed
MAX_BLOCKS = 400
PROFILE_ROUND = "r04"     # profiles/<round>/traffic.json: PMC passes of this same command (tools/tools_prof.sh)

WORKLOADS = {
    # name: (description, H, W, L, model, tracer, bytes per cell-update = 2 * fields * 8, dt)
    "c3": ("2-D shallow water + theta + viscosity + van-Leer tracer, 4096x2048 fp64 "
           "(BASELINE configs[2])", 2048, 4096, 1, "SW2D_TEMP", "van_leer", 80.0, 300.0),
    "c2": ("2-D shallow water Matsuno C-grid, 720x360 fp64 (BASELINE configs[1])",
           360, 720, 1, "SW2D", None, 48.0, 300.0),
    # cells are (k, j, i) points; 4 3-D fields + p: (64 + 16/L) bytes per cell-update
    "c4": ("2.5-D sigma-level primitive equations, 1440x720x24 fp64 (BASELINE configs[3])",
           720, 1440, 24, "PE25D", None, 64.0 + 16.0 / 24, 1.0),
    # the same workload with arithmetic and storage in fp32 (BASELINE configs[4]: fp32 vs fp64 sweep)
    "c4_f32": ("2.5-D sigma-level primitive equations, 1440x720x24 fp32",
               720, 1440, 24, "PE25D", None, 32.0 + 8.0 / 24, 1.0),
    # BASELINE configs[4] grid (dynamics only; the radiation + fp32-vs-fp64 sweep is tools/tools_c5_sweep.py);
    # not part of the default run: --workload c5 / c5_f32
    "c5": ("2.5-D sigma-level primitive equations, 2880x1440x40 fp64 (BASELINE configs[4] grid)",
           1440, 2880, 40, "PE25D", None, 64.0 + 16.0 / 40, 1.0),
    "c5_f32": ("2.5-D sigma-level primitive equations, 2880x1440x40 fp32 (BASELINE configs[4] grid)",
               1440, 2880, 40, "PE25D", None, 32.0 + 8.0 / 40, 1.0),
    # BASELINE configs[4] as a workload: every step = the dynamics (with the humidity tracer q) + the grey
    # radiation / solar_timestep column physics (grey_solar.py:358-563, no_limits_2_5d.py:66-75).  The
    # physics pass reads and writes theta once more (+16 B per cell) and the 2-D ground temperature.
    "c5_phys": ("2.5-D primitive equations + grey_solar radiation + humidity tracer, 2880x1440x40 fp64 "
                "(BASELINE configs[4])", 1440, 2880, 40, "PE25D", None, 80.0 + 32.0 / 40, 1.0),
    "c5_phys_f32": ("2.5-D primitive equations + grey_solar radiation + humidity tracer, 2880x1440x40 fp32 "
                    "(BASELINE configs[4])", 1440, 2880, 40, "PE25D", None, 40.0 + 16.0 / 40, 1.0),
}
PHYS_UTC0 = 6 * 3600.0
ALSO = ("c2", "c3", "c4", "c4_f32", "c5_phys")     # secondary workloads of the default run (N > 1: c3, c4, c5_phys)
DX = 300e3


def synth(name, H, W, L=1, row0=0, nrows=None, geom=None):
    """SURVEY.md 8d synthetic inputs, seed default_rng(0); rows [row0,row0+nrows) only."""
    rng = np.random.default_rng(0)
    nrows = H if nrows is None else nrows
    sl = slice(row0, row0 + nrows)
    if WORKLOADS[name][4] == "PE25D":
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
    if name == "c2":
        p = 8000 + rng.standard_normal((H, W))
        return dict(u=u[sl], v=v[sl], p=p[sl])
    p = 101325 + rng.standard_normal((H, W))
    t = 273.16 + rng.standard_normal((H, W))
    q = rng.random((H, W))
    return dict(u=u[sl], v=v[sl], p=p[sl], t=t[sl], q=q[sl])


def cpu_baseline(name):
    """the oracle (NumPy restatement, bit-identical to the reference) on the host, 1 core"""
    from oracle import sw2d, sw2d_temp, tracer
    _, H, W, L, _, _, _, dt = WORKLOADS[name]
    if WORKLOADS[name][4] == "PE25D":
        from oracle import dynamics, geometry as ogeo, physics
        import copy
        og = ogeo.gen_geometry(H, W, L, sig_func=ogeo.manabe_sig)
        phys = "phys" in name
        if H * W * L <= 30e6:
            # the configured grid itself (c4: 2 steps, ~10 s each on one core)
            rows, nst, what = slice(0, H), 1, "%d full step of the %dx%dx%d grid" % (1, W, H, L)
            sg = og
        else:
            # bounded sample of the configured grid: a latitude strip of FULL width and depth (the cost of a
            # step is per row: FFT rows of W, column scans of L), 90 mid-latitude rows, same recipe
            rows, nst = slice(H // 4, H // 4 + 90), 2
            what = "%d steps of a 90-row latitude strip of the %dx%dx%d grid (full width and depth)" % (nst, W, H, L)
            sg = copy.copy(og)
            sg.height = 90
            sg.dx_j, sg.dx_h = og.dx_j[:, rows, :], og.dx_h[:, rows, :]
            sg.lat, sg.heightmap = og.lat[rows], og.heightmap[rows]
        s = synth(name, H, W, L, geom=og)
        st = (s["p"][rows], s["u"][:, rows], s["v"][:, rows], s["t"][:, rows], s["q"][:, rows])
        gt = np.full(st[0].shape, 288.0)
        del s
        cells = st[1].size
        t0 = time.perf_counter()
        for n in range(nst):
            st = dynamics.matsuno_timestep(*st, dt, sg)
            if phys:
                t_n, gt = physics.solar_timestep(st[3], st[0], gt, dt, PHYS_UTC0 + n * dt, sg)
                st = (st[0], st[1], st[2], t_n, st[4])
        el = time.perf_counter() - t0
        return {"value": cells * nst / el, "unit": "cell-updates/s", "cores": 1, "kind": "port",
                "sample": "%s with the NumPy oracle%s, %.1f s; host has %d cores"
                          % (what, " (dynamics + solar_timestep)" if phys else "", el, os.cpu_count())}
    s = synth(name, H, W)
    t0 = time.perf_counter()
    if name == "c2":
        st = (s["u"], s["v"], s["p"])
        nst = 20
        for _ in range(nst):
            st = sw2d.matsumo_scheme(*st, DX, dt)
    else:
        nst = 5
        V = np.stack([s["v"], s["u"]])
        q, st = s["q"], (s["u"], s["v"], s["p"], s["t"])
        for _ in range(nst):
            q = tracer.limited_advection(dt, (DX, DX), V, q)
            st = sw2d_temp.matsumo_scheme(*st, DX, dt)
    el = time.perf_counter() - t0
    return {"value": H * W * nst / el, "unit": "cell-updates/s", "cores": 1, "kind": "port",
            "sample": "%d full step(s) of the %dx%d grid with the NumPy oracle, %.1f s; host has %d cores"
                      % (nst, W, H, el, os.cpu_count())}


class Ctx:
    """process-wide handles: torch, dist, ranks"""


def halo_steps_for(name, world):
    if os.environ.get("GCM_HALO_STEPS"):
        return int(os.environ["GCM_HALO_STEPS"])
    if WORKLOADS[name][4] == "PE25D" or world == 1:
        return 1
    # 2-D bands are small (C3 at 8 GPUs: 21 us of compute per step): exchange every k steps
    return 4 if world == 2 else 8


def run_workload(cx, name, steps, warmup, variant="fused", world=None, want_kernel=True):
    """-> dict(value, ms_per_step, [roofline]) measured on `world` ranks (default: all)"""
    import gcmiipy_amd as g
    from gcmiipy_amd import _lib, geometry
    from gcmiipy_amd.bands import BandRunner, HipBandEngine, LoopbackExchange, split_rows
    torch, dist = cx.torch, cx.dist
    world = cx.world if world is None else world
    solo = world == 1 and cx.world > 1                # 1-GPU reference inside an N-rank job
    active = (not solo) or cx.rank == 0
    desc, H, W, L, model, tracer, bpc, dt = WORKLOADS[name]
    res = None
    if active:
        rank = 0 if solo else cx.rank
        row0, nrows = split_rows(H, world)[rank]
        geom = geometry.gen_geometry(H, W, L, sig_func=geometry.manabe_sig) if model == "PE25D" else None
        k = halo_steps_for(name, world)
        core = g.Core(getattr(_lib, model), W, nrows, L, dx=DX, geom=geom,
                      tracer={None: _lib.TRACER_NONE, "van_leer": _lib.TRACER_VANLEER}[tracer],
                      variant=_lib.VARIANT_FUSED if variant == "fused" else _lib.VARIANT_STAGED,
                      filter=not os.environ.get("GCM_BENCH_NOFILTER"),   # diagnostic only
                      nranks=world, rank=rank, global_height=H, row0=row0, device=cx.local,
                      stream=torch.cuda.current_stream().cuda_stream, halo_steps=k,
                      dtype="f32" if name.endswith("_f32") else "f64")
        core.set_state(**synth(name, H, W, L, row0, nrows, geom))
        eng = HipBandEngine(core, torch, stream_aware=cx.backend == "nccl") if world > 1 else None
        runner = BandRunner(eng, rank, world, cx.ring if world > 1 else dist)
        region = {}

        phys = "phys" in name
        if phys:
            # configs[4]: every step = the dynamics step + solar_timestep at the handle's clock (gcm_set_physics), on one
            # GPU inside gcm_step, on a latitude band inside gcm_band_run (ghost rows radiated locally: no third exchange)
            core.set_ground(np.full((nrows, W), 288.0))
            if eng is not None:
                eng.set_physics(geom, PHYS_UTC0)
            else:
                core.set_physics(geom, PHYS_UTC0)

        def run_chunk(n, timed):
            if world == 1:
                if timed:   # same launches, bracketed by HIP events on the launch stream
                    region["ms"] = region.get("ms", 0.0) + core.time_steps(n, dt, per_kernel=False)[0]
                else:
                    core.step(n, dt)
            else:
                runner.run(n, dt)

        # The SURVEY's noise initial state of the 2-D workloads is not a balanced flow: it goes
        # non-finite after ~1100 (c3) / ~1300 (c2) steps, in the reference as here.  A run longer
        # than `life` steps goes back to the initial state (a device-side copy, gcm_restore, inside
        # the timed region: one 0.1 ms copy per `life` steps), at an exchange boundary on bands.
        life = {"c3": 400, "c2": 800}.get(name)
        if life is not None:
            life = life // k * k
            core.snapshot()
        since = [0]

        def run(n, timed=False):
            while n > 0:
                m = n if life is None else min(n, life - since[0])
                run_chunk(m, timed)
                since[0] += m
                n -= m
                if life is not None and since[0] == life:
                    core.restore()
                    since[0] = 0
    else:
        def run(n, timed=False):
            pass

    def fence():
        if dist is not None and not solo:
            dist.barrier()
        torch.cuda.synchronize()

    if dist is not None:
        dist.barrier()
    # Untimed pre-conditioning: the chip needs some tens of milliseconds of sustained load before it
    # runs at its sustained clock (the same 400 steps of c3 take 0.153 ms each from idle and 0.146
    # after 60 ms of load).  SETTLE_S seconds of the same steps, then back to the initial state
    # (2-D workloads), then the W warm-up steps and the K timed steps of the contract.
    if active:
        t0 = time.perf_counter()
        run(2 * k)
        torch.cuda.synchronize()
        per = (time.perf_counter() - t0) / (2 * k)
    else:
        per = 0.0
    if dist is not None and not solo:
        tt = torch.tensor([per], dtype=torch.float64, device="cuda" if cx.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        per = float(tt.item())
    n_settle = 0
    if active:
        n_settle = min(20000, int(SETTLE_S / max(per, 1e-6))) // k * k
        run(n_settle)
        if life is not None and since[0]:
            core.restore()
            since[0] = 0
    # Deep-halo bands whose exchange the library posts itself: the exchange can be hidden behind the
    # interior rows of the steps around it at the price of four more launches per window
    # (gcm_set_band_overlap).  GCM_PE25D bands: the interior rows' update can be held back until the edge rows'
    # update has been dispatched, which gets the edge rows -- and the exchange -- out 45 us earlier per stage and
    # costs the interior rows ~10 us (profiles/r04/band_matrix_edges_first.txt: it pays from ~30 us per exchange).
    # Whether either pays depends on what the exchange costs between THESE devices, so both sequences are
    # timed here (max over ranks) and the faster one is kept.
    overlap_probe = None
    if active and world > 1 and not solo and (k > 1 or model == "PE25D") and getattr(runner, "native", False):
        overlap_probe = {}
        npr = 6 * k if model != "PE25D" else 24
        for flag in (0, 1):
            core.set_band_overlap(flag)
            run(2 * k)
            fence()
            t0 = time.perf_counter()
            run(npr)
            fence()
            tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64,
                              device="cuda" if cx.backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            overlap_probe["overlap" if flag else "plain"] = float(tt.item()) / npr * 1e3
        overlap_probe["chosen"] = "overlap" if overlap_probe["overlap"] < overlap_probe["plain"] else "plain"
        core.set_band_overlap(overlap_probe["chosen"] == "overlap")
        if life is not None and since[0]:
            core.restore()
            since[0] = 0
    run(warmup)
    # The timed region: a block of exactly `steps` steps between two fences (barrier + synchronize),
    # repeated until MIN_TIMED_S seconds have been timed (every rank sees the same all-reduced block
    # times, so all stop together); the median block is the result.
    blocks, own_blocks, t_queued = [], [], 0.0
    while True:
        fence()
        t0 = time.perf_counter()
        run(steps, timed=True)
        t_queued += time.perf_counter() - t0          # the host has queued all K steps
        torch.cuda.synchronize()
        own_blocks.append(time.perf_counter() - t0)   # this rank's own band, before it waits for the others
        fence()
        el = time.perf_counter() - t0
        if dist is not None and not solo:
            tt = torch.tensor([el], dtype=torch.float64, device="cuda" if cx.backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        blocks.append(el)
        if sum(blocks) >= MIN_TIMED_S or len(blocks) >= MAX_BLOCKS:
            break
    el = float(np.median(blocks))
    t_queued /= len(blocks)
    # N > 1: every rank's own median (its band done, before the closing barrier), gathered so that the line shows the
    # slowest and the fastest band
    per_rank_ms = None
    if dist is not None and not solo and world > 1:
        tt = torch.tensor([float(np.median(own_blocks)) / steps * 1e3], dtype=torch.float64,
                          device="cuda" if cx.backend == "nccl" else "cpu")
        allv = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(allv, tt)
        per_rank_ms = [float(x.item()) for x in allv]
    if active:
        assert core.diag(_lib.DIAG_ANY_NAN) == 0.0, "state went NaN during the timed run"
        cells = H * W * L
        value = cells * steps / el
        res = {"workload": desc, "grid": [W, H] + ([L] if L > 1 else []), "n_gpus": world, "steps": steps,
               "warmup": warmup, "value": value, "ms_per_step": el / steps * 1e3,
               "ms_per_step_min": min(blocks) / steps * 1e3, "ms_per_step_max": max(blocks) / steps * 1e3,
               "timed_blocks": len(blocks), "timed_seconds": sum(blocks), "settle_steps": 2 * k + n_settle,
               "dtype": "f32" if name.endswith("_f32") else "f64",
               "bytes_per_cell_update": bpc,
               "hbm_roofline_frac_whole_job": value * bpc / (world * HBM_PEAK_GBS * 1e9),
               "exchange": cx.exchange if world > 1 else None,
               "band_overlap_probe_ms_per_step": overlap_probe,
               "decomposition": "%d latitude band(s)%s" % (
                   world, ", ghost rows exchanged every %d steps" % k if world > 1 and k > 1 else "")}
        if world > 1:
            # what makes the N > 1 line readable on its own: whole-job roofline, every band's own time, the bytes the
            # ghost-row exchange moves, and how many ranks the RCCL communicator really has
            per_step = 2.0 if model == "PE25D" else 1.0 / k                      # exchanges per step
            hb = core.halo_bytes()
            res["roofline"] = {"bound": "hbm", "achieved": value * bpc / 1e9, "peak": world * HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": value * bpc / (world * HBM_PEAK_GBS * 1e9), "traffic": None,
                               "kernel": "whole job: %d latitude bands, algorithmic bytes of the global grid over the "
                                         "slowest rank's wall time, against %d x %.0f GB/s" % (world, world, HBM_PEAK_GBS)}
            res["band_ms_per_step"] = {"max": max(per_rank_ms), "min": min(per_rank_ms), "per_rank": per_rank_ms}
            res["exchange_bytes"] = {"message_bytes": hb, "messages_per_exchange_per_rank": 2,
                                     "exchanges_per_step": per_step,
                                     "sent_per_rank_per_step": 2 * hb * per_step,
                                     "whole_job_per_step": world * 2 * hb * per_step}
            res["rccl_ranks"] = cx.ring.count() if hasattr(cx.ring, "count") else None
            # diagnostics for the multi-GPU line: how long the host needs to queue a step, and what
            # the same band costs with the exchange replaced by a device-local copy (same launches
            # and stream dependencies, no xGMI traffic) -- the difference to ms_per_step is what the
            # exchange adds.  Run AFTER the timed region; its ghost rows are not a model state.
            lb = BandRunner(eng, rank, world, LoopbackExchange())
            lb.primed, lb.count = True, 0
            nlb = max(k, 8) * 4
            lb.run(nlb, dt)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            lb.run(nlb, dt)
            e1.record()
            torch.cuda.synchronize()
            # what the HOST needs to queue a step when no call can block on a full queue: a few steps queued
            # behind one long spin kernel (torch.cuda._sleep, ~40 ms).  "host_queue_ms_per_step" above it is
            # taken on a busy device and also counts the time the queue holds the host back.
            nq = max(k, 4) // k * k
            host_idle = None
            try:
                torch.cuda.synchronize()
                torch.cuda._sleep(int(40e-3 * 2.0e9))
                tq = time.perf_counter()
                lb.run(nq, dt)
                host_idle = (time.perf_counter() - tq) * 1e3 / nq
            except AttributeError:              # (a torch without the spin kernel: the figure is simply absent)
                pass
            torch.cuda.synchronize()
            res["diagnostics"] = {"host_queue_ms_per_step": t_queued / steps * 1e3,
                                  "host_queue_ms_per_step_idle_device": host_idle,
                                  "band_ms_per_step_local_exchange": e0.elapsed_time(e1) / nlb,
                                  "rank": rank}
        if world == 1 and want_kernel:
            # dominant kernel.  fused 2-D: one launch per step, so its average duration over the
            # timed region is (HIP-event time of the region on the launch stream) / launches; the
            # back-to-back launches leave no gap (rocprofv3 trace: next start == previous end).
            # "kernel_ms_isolated" is a second pass with an event pair around every launch
            # (idle gaps between launches let the chip clock higher, so it reads lower).
            launches_timed = steps * len(blocks)
            kiso = None
            launches = 1
            if model == "PE25D":
                # one Euler stage = four launches (spu filter with pit in its launch, geopot, pgf filter, update) on two
                # streams; the roofline entry is the WHOLE stage: half of the step's algorithmic bytes over
                # half of the step's device time (HIP events around the timed blocks)
                kname, kms, launches = ("pe25d stage: pe_spu_filter_loop (+ pit) + pe_geopot + pe_pgf_filter + pe_update_rows"
                                        + (" (+ half of pe_radiation)" if phys else "")), region["ms"] / launches_timed / 2, 2
            elif variant == "fused" and model == "SW2D":
                # plain shallow water steps in pairs (one launch = two steps): per-step figures
                kname, kms = "sw2d_fused2_kernel (two steps per launch; per step)", region["ms"] / launches_timed
                _, kiso = core.time_steps(min(steps, 50), dt)
            elif variant == "fused":
                kname, kms = "sw2d_fused_kernel", region["ms"] / launches_timed
                _, kiso = core.time_steps(min(steps, 50), dt)
            else:
                _, kiso = core.time_steps(min(steps, 50), dt)
                kname, kms = "sw2d_stage_kernel (corrector stage)", kiso
            ach = cells * bpc / launches / (kms * 1e-3) / 1e9
            # HBM bytes per launch from the rocprofv3 PMC passes of this same command (separate
            # FETCH_SIZE / WRITE_SIZE runs, gfx950 correction applied): profiles/<round>/traffic.json,
            # written by tools_prof.sh + tools_traffic.py; null if that file is absent
            traffic = None
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", PROFILE_ROUND, "traffic.json")))[name]
                if model == "PE25D":                   # sum over the stage's kernels (each once per stage)
                    traffic = sum(v["hbm_bytes_per_launch"] for kk, v in tj.items()
                                  if kk.startswith("gcm::pe_") and "to_device" not in kk and "to_host" not in kk
                                  and "radiation" not in kk and "energy" not in kk
                                  and "colsum" not in kk)          # (colsum: once after set_state, not per stage)
                    if phys:
                        traffic += tj["gcm::pe_radiation_kernel"]["hbm_bytes_per_launch"] / 2.0
                else:
                    traffic = tj[("gcm::" + kname.split(" ")[0])]["hbm_bytes_per_launch"]   # fp64 runs only
                    if "two steps per launch" in kname:
                        traffic /= 2.0                      # per step, as kernel_ms
            except Exception:
                pass
            # context: what a plain device-to-device copy of the same bytes (state in, state out)
            # reaches on this box, timed the same way
            nb = int(cells * bpc / 2)
            src = torch.empty(nb // 8, dtype=torch.float64, device="cuda").normal_()
            dst = torch.empty_like(src)
            for _ in range(3):
                dst.copy_(src)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                dst.copy_(src)
            e1.record()
            torch.cuda.synchronize()
            copy_gbs = 2.0 * nb / (e0.elapsed_time(e1) / 20 * 1e-3) / 1e9
            del src, dst
            res["device_copy_same_bytes"] = {"GB/s": copy_gbs, "kernel_traffic_over_copy_rate":
                                             (traffic or cells * bpc / launches) / (kms * 1e-3) / 1e9 / copy_gbs}
            res["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "kernel": kname,
                               "kernel_ms": kms, "kernel_ms_isolated": kiso, "launches_timed": launches_timed * launches,
                               "algorithmic_bytes_per_launch": cells * bpc / launches}
            state_bytes = cells * bpc / 2.0
            if state_bytes <= L2_BYTES:
                # a state that stays in the L2s between steps (c2: 6.2 MB): HBM is not what bounds it.
                # Against the cache's own rate the step is bounded by launch + latency.
                res["roofline"]["cache_ceiling"] = {"level": "L2 (8 x 4 MiB)", "peak": L2_PEAK_GBS, "unit": "GB/s",
                                                    "frac": ach / L2_PEAK_GBS, "state_bytes": state_bytes,
                                                    "floor_us_per_step_at_peak": cells * bpc / (L2_PEAK_GBS * 1e9) * 1e6,
                                                    "dependent_launch_boundary_us": [1.5, 1.9]}
        core.close()
    if dist is not None:
        dist.barrier()
    return res


def bring_up_direct_rccl(cx, torch, dist, rccl_cls=None, device="cuda", timeout_s=None):
    """(rccl_cls / device / timeout_s: what the CPU tests replace -- a stand-in for gcmiipy_amd.rccl.RcclP2P over a gloo
    group, tests/test_bench_cpu.py -- the control flow is the product's.)
    The ghost rows go over RCCL called directly (gcmiipy_amd.rccl; the library posts the exchange
    itself, gcm_band_run).  Bringing that communicator up between devices cannot be rehearsed on the
    one-GPU development box, so every step is agreed on by ALL ranks through torch.distributed
    collectives (a gloo group on the CPU, created before the bring-up) issued from the main thread in
    the same order everywhere -- a rank that fails early
    cannot leave the others parked in a different collective:
      1. every rank loads librccl, rank 0 makes the unique id         -> all_reduce(MIN) of 'ok'
      2. the id travels as a 128-byte CUDA tensor                      -> broadcast
      3. ncclCommInitRank + a one-hop ring self-check on a watchdog thread (the only calls that can
         hang), joined with a timeout                                  -> all_reduce(MIN) of 'ok'
    Any failure: cx.exchange_fallback = <reason> (reported in the JSON line), torch.distributed
    carries the exchange, and if a bring-up thread is still inside RCCL the process exits NON-ZERO
    after printing (cx.stuck)."""
    import threading

    def agree(flag):
        # over the CPU (gloo) group: a bring-up thread that timed out may still sit inside
        # ncclCommInitRank on this device, and a device collective beside it could deadlock
        t = torch.tensor([1 if flag else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=cx.cpu_group)
        return int(t.item()) == 1

    err, uid = None, None
    try:
        if rccl_cls is None:
            from gcmiipy_amd.rccl import RcclP2P
        else:
            RcclP2P = rccl_cls
        if cx.rank == 0:
            uid = RcclP2P.new_unique_id()
    except Exception as e:          # noqa: BLE001
        err = "librccl / unique id: %r" % (e,)
    if not agree(err is None):
        cx.exchange_fallback = err or "librccl / unique id failed on another rank"
        print("bench.py: direct RCCL exchange unavailable (%s); using torch.distributed" % cx.exchange_fallback, file=sys.stderr)
        return
    box_t = torch.zeros(128, dtype=torch.uint8, device=device)
    if cx.rank == 0:
        box_t.copy_(torch.frombuffer(bytearray(uid), dtype=torch.uint8))
    dist.broadcast(box_t, src=0)
    uid = bytes(box_t.cpu().numpy().tobytes())
    box = {}

    def init():
        try:
            if device == "cuda":
                torch.cuda.set_device(cx.local)       # the HIP device is per thread
            ring = RcclP2P(None, cx.rank, cx.world, uid_bytes=uid)
            ring.self_check()
            box["ring"] = ring
        except Exception as e:      # noqa: BLE001
            box["err"] = e

    th = threading.Thread(target=init, daemon=True)
    th.start()
    th.join(float(os.environ.get("GCM_BENCH_RCCL_TIMEOUT_S", "120")) if timeout_s is None else timeout_s)
    cx.stuck = th.is_alive()
    if agree("ring" in box):
        cx.ring, cx.exchange = box["ring"], "RCCL ncclSend/ncclRecv groups posted by the library (gcm_band_run)"
        return
    cx.exchange_fallback = ("ncclCommInitRank / self-check timed out" if cx.stuck
                            else repr(box.get("err", "failed on another rank")))
    print("bench.py: direct RCCL exchange unavailable (%s); using torch.distributed" % cx.exchange_fallback, file=sys.stderr)


def agree_stuck(cx, torch, dist):
    """one verdict for the whole job: if a bring-up thread is stuck on ANY rank, every rank reports the fallback, keeps
    off the device collectives' path of that thread and exits non-zero (finish)"""
    t = torch.tensor([1 if cx.stuck else 0], dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=cx.cpu_group)
    cx.stuck = int(t.item()) == 1
    return cx.stuck


def finish(cx, out):
    """rank 0 prints the ONE JSON line; a job with a bring-up thread still inside RCCL has said so in that line
    (exchange_fallback) and now says so with its exit code too -- non-zero, from THIS process, without waiting for
    that thread and without starting or exec'ing anything"""
    if cx.rank == 0:
        print(json.dumps(out), flush=True)
    if cx.stuck:
        sys.stdout.flush()
        os._exit(4)


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD job
    (torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1) from a parent that has not
    imported torch nor touched the GPU, pass the arguments through, let rank 0's JSON line go to the
    inherited stdout and return the job's exit code.  Nothing is re-exec'ed."""
    import subprocess
    import uuid
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    # the launcher picks AND HOLDS its rendezvous port itself (c10d store on 127.0.0.1:0): nothing is probed here
    # and bound again later, so two benches on one node cannot collide on a port
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--rdzv-backend=c10d", "--rdzv-endpoint=127.0.0.1:0", "--rdzv-id=" + uuid.uuid4().hex,
           "--local-addr", "127.0.0.1", os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--variant", default="fused", choices=["fused", "staged"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--only", action="store_true", help="skip the secondary workloads under 'also'")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a.gpus))                 # before torch is imported or the GPU touched

    import torch
    cx = Ctx()
    cx.torch = torch
    cx.rank = int(os.environ.get("RANK", "0"))
    cx.world = int(os.environ.get("WORLD_SIZE", "1"))
    cx.local = int(os.environ.get("LOCAL_RANK", "0"))
    if cx.world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)"
                         % (a.gpus, cx.world))
    if os.environ.get("GCM_BENCH_BACKEND", "nccl") != "nccl":
        cx.local = 0
    # --gpus N against the devices this node shows, BEFORE any collective: a rank that cannot have a device of its own
    # says so and the job ends non-zero (device_count() does not initialise the GPU)
    if cx.local >= torch.cuda.device_count():
        if cx.rank == 0:
            print("bench.py: --gpus %d but this node shows %d GPU(s) (GCM_BENCH_BACKEND=gloo rehearses the ranks on one)"
                  % (a.gpus, torch.cuda.device_count()), file=sys.stderr)
        raise SystemExit(3)
    torch.cuda.set_device(cx.local)
    cx.dist = None
    cx.ring, cx.exchange, cx.stuck, cx.exchange_fallback = None, None, False, None
    cx.backend = "nccl"
    if cx.world > 1:
        import torch.distributed as dist
        # GCM_BENCH_BACKEND=gloo: rehearsal of the multi-rank path on ONE GPU (RCCL refuses two
        # ranks per device); the measured configuration is nccl (= RCCL over xGMI)
        cx.backend = os.environ.get("GCM_BENCH_BACKEND", "nccl")
        if cx.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", cx.local))
        else:
            cx.local = 0
            dist.init_process_group(cx.backend)
        cx.dist = dist
        # the ghost rows go over RCCL called directly (gcmiipy_amd.rccl: the exchange kernel on the
        # library's comm stream, a few ctypes calls per exchange); torch.distributed bootstraps it and
        # does the barriers.  GCM_BENCH_EXCHANGE=torch keeps batch_isend_irecv.
        cx.ring, cx.exchange = dist, "torch.distributed batch_isend_irecv (%s)" % cx.backend
        cx.exchange_fallback = None
        cx.cpu_group = dist.new_group(backend="gloo") if cx.backend == "nccl" else None
        if cx.backend == "nccl" and os.environ.get("GCM_BENCH_EXCHANGE", "rccl") == "rccl":
            bring_up_direct_rccl(cx, torch, dist)
            agree_stuck(cx, torch, dist)

    main_res = run_workload(cx, a.workload, a.steps, a.warmup, a.variant)
    also, cpu_cache = {}, {}
    if not a.only:
        for name in ALSO:
            if name == a.workload:
                continue
            if cx.world > 1 and name == "c2":
                continue                                   # 360 rows: not a multi-GPU workload
            if cx.world > 1 and name == "c4_f32":
                continue
            # c2: the noise IC goes unstable (in the reference too) near step 1300
            st, wu = {"c2": (600, 50), "c3": (100, 10), "c4": (16, 3), "c4_f32": (16, 3), "c5_phys": (6, 2)}[name]
            if cx.world > 1 and name == "c4":
                st, wu = 60, 10                            # a band's step is a fraction of a millisecond
            if cx.world > 1 and name == "c5_phys":
                st, wu = 12, 3
            r = run_workload(cx, name, st, wu, want_kernel=cx.world == 1)
            if cx.world > 1:                               # same-run single-GPU reference (rank 0 alone)
                r1 = run_workload(cx, name, max(st // 2, 4), 2, world=1, want_kernel=False)
                if cx.rank == 0:
                    r["one_gpu_same_run"] = {"value": r1["value"], "ms_per_step": r1["ms_per_step"]}
                    r["speedup_vs_one_gpu"] = r["value"] / r1["value"]
            if cx.world == 1 and not a.no_cpu:
                # the oracle is float64 whatever the handle's storage type: c4_f32 is set beside c4's figure
                base = name[:-4] if name.endswith("_f32") else name
                if base not in cpu_cache:
                    cpu_cache[base] = cpu_baseline(base)
                r["cpu_baseline"] = cpu_cache[base]
            if cx.rank == 0:
                also[name] = r

    if cx.rank == 0:
        out = {
            "metric": "cell-updates/s (C-grid Matsuno step)", "value": main_res["value"],
            "unit": "cell-updates/s", "n_gpus": cx.world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": main_res["ms_per_step"], "ms_per_step_min": main_res["ms_per_step_min"],
            "ms_per_step_max": main_res["ms_per_step_max"], "timed_blocks": main_res["timed_blocks"],
            "timed_seconds": main_res["timed_seconds"], "settle_steps": main_res["settle_steps"],
            "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": main_res["dtype"], "data": "synthetic",
            "config": {"workload": main_res["workload"], "grid": main_res["grid"], "variant": a.variant,
                       "decomposition": main_res["decomposition"],
                       "bytes_per_cell_update": main_res["bytes_per_cell_update"]},
            "hbm_roofline_frac_whole_job": main_res["hbm_roofline_frac_whole_job"],
        }
        if "roofline" in main_res:
            out["roofline"] = main_res["roofline"]
        if cx.world > 1:
            out["exchange"] = cx.exchange
            out["exchange_fallback"] = cx.exchange_fallback     # null: the direct RCCL ring came up on every rank
            for kk in ("rccl_ranks", "band_ms_per_step", "exchange_bytes"):
                out[kk] = main_res.get(kk)
        if "diagnostics" in main_res:
            out["diagnostics"] = main_res["diagnostics"]
        if "device_copy_same_bytes" in main_res:
            out["device_copy_same_bytes"] = main_res["device_copy_same_bytes"]
        if cx.world == 1:
            out["cpu_baseline"] = None if a.no_cpu else cpu_cache.get(a.workload) or cpu_baseline(a.workload)
        if also:
            out["also"] = also
    if cx.dist is not None:
        cx.dist.barrier()
    finish(cx, out if cx.rank == 0 else None)
    if cx.dist is not None:
        cx.dist.destroy_process_group()


if __name__ == "__main__":
    main()
