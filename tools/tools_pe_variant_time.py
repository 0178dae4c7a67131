#!/usr/bin/env python3
"""Diagnostic: step the C4 grid (1440x720x24 fp64 primitive equations) with or without the polar
filter and print the time per step; run it under `rocprofv3 --kernel-trace --stats` with
GCM_PE_SINGLE_STREAM=1 for per-kernel durations (filter off: what K1 / K3 cost without their
transforms, i.e. their load / thermodynamics / store phases alone).

  python3 tools/tools_pe_variant_time.py [--no-filter] [--steps 30]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-filter", action="store_true")
    ap.add_argument("--steps", type=int, default=30)
    a = ap.parse_args()
    import torch
    import bench
    import gcmiipy_amd as g
    from gcmiipy_amd import _lib, geometry
    desc, H, W, L, model, tracer, bpc, dt = bench.WORKLOADS["c4"]
    geom = geometry.gen_geometry(H, W, L, sig_func=geometry.manabe_sig)
    st = bench.synth("c4", H, W, L, geom=geom)
    core = g.Core(_lib.PE25D, W, H, L, geom=geom, filter=not a.no_filter)
    core.set_state(**st)
    core.step(5, dt if not a.no_filter else dt / 20)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    core.step(a.steps, dt if not a.no_filter else dt / 20)
    e1.record()
    torch.cuda.synchronize()
    print("filter=%s  %.4f ms/step" % (not a.no_filter, e0.elapsed_time(e1) / a.steps))
    core.close()


if __name__ == "__main__":
    main()
