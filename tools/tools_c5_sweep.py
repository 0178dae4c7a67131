#!/usr/bin/env python3
"""BASELINE configs[4]: 2.5-D primitive equations + grey radiation + humidity tracer, fp32 vs fp64.

Runs the same synthetic initial state (SURVEY.md 8d recipe, bench.synth) through two resident
GCM_PE25D handles, one float64 and one float32, each step = gcm_step (Matsuno) + gcm_solar_step
(grey_solar.basic_grey_radiation + no_limits_2_5d.solar_timestep), and reports the relative error
of the fp32 state against the fp64 state (L-inf over max|field|) after 1, 10 and 100 steps, with
the step times of both.  One GPU; writes one JSON document.

  python tools/tools_c5_sweep.py [--grid c5|c4|small] [--out profiles/r01/c5_fp32_sweep.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

GRIDS = {"c5": (1440, 2880, 40), "c4": (720, 1440, 24), "small": (90, 180, 12)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", default="c5", choices=sorted(GRIDS))
    ap.add_argument("--out", default=None)
    ap.add_argument("--dt", type=float, default=1.0)
    a = ap.parse_args()
    import torch
    import bench
    import gcmiipy_amd as g
    from gcmiipy_amd import _lib, geometry
    torch.cuda.set_device(0)
    H, W, L = GRIDS[a.grid]
    geom = geometry.gen_geometry(H, W, L, sig_func=geometry.manabe_sig)
    bench.WORKLOADS["_sweep"] = ("sweep", H, W, L, "PE25D", None, 0.0, a.dt)
    st = bench.synth("_sweep", H, W, L, geom=geom)
    gt = np.full((H, W), 300.0)
    cores = {}
    for dt_name in ("f64", "f32"):
        c = g.Core(_lib.PE25D, W, H, L, geom=geom, dtype=dt_name)
        c.set_state(**st)
        c.set_ground(gt)
        cores[dt_name] = c
    del st
    marks, out, utc, done = (1, 10, 100), [], 0.0, 0
    ms = {"f64": 0.0, "f32": 0.0}
    for mark in marks:
        for name, c in cores.items():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            u = utc
            for _ in range(mark - done):
                c.step(1, a.dt)
                c.solar_step(geom, a.dt, u)
                u += a.dt
            torch.cuda.synchronize()
            ms[name] += (time.perf_counter() - t0) * 1e3
        utc += a.dt * (mark - done)
        done = mark
        ref = cores["f64"].get_state()
        got = cores["f32"].get_state()
        err = {}
        for f, x, y in zip("puvtq", ref, got):
            assert np.isfinite(x).all() and np.isfinite(y).all(), "non-finite state in field " + f
            err[f] = float(np.abs(y - x).max() / np.abs(x).max())
        eg = float(np.abs(cores["f32"].get_ground() - cores["f64"].get_ground()).max() / 300.0)
        err["ground"] = eg
        out.append({"steps": mark, "rel_err_fp32_vs_fp64": err})
        print("after %3d steps:" % mark, " ".join("%s=%.2e" % kv for kv in err.items()), flush=True)
        del ref, got
    doc = {"workload": "2.5-D primitive equations + grey radiation + humidity tracer, %dx%dx%d" % (W, H, L),
           "dt": a.dt, "step": "gcm_step + gcm_solar_step", "error_norm": "max|fp32 - fp64| / max|fp64| per field",
           "sweep": out,
           "ms_per_step_incl_radiation": {k: v / marks[-1] for k, v in ms.items()},
           "note": "fp32 handle: state, intermediates and dynamics arithmetic in float32; the column physics "
                   "computes in float64 and stores float32"}
    txt = json.dumps(doc, indent=1)
    print(txt)
    if a.out:
        with open(os.path.join(ROOT, a.out), "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    main()
