"""State I/O: the reference keeps its state in RAM only; long device-resident
runs want a restart file.  One `.npz` per handle holds the prognostic tuple, the ground
temperature of the column physics (when set), the model tag, every option the handle was created
with (dx, tracer, kernel variant, filter, Coriolis, dtype, band placement) and the geometry tables:
`restore()` rebuilds an equivalent handle and the run resumes bit for bit.  A latitude band writes
ITS rows (one file per rank; `row0` / `global_height` are in the file)."""
import numpy as np

from . import _lib
from .core import Core
from .geometry import Geom

_GEOM_KEYS = ("sige", "sigt", "sigb", "dsig", "sig", "dsigv", "dx_j", "dx_h", "dy", "ptop",
              "heightmap", "area", "lat", "long")
_MODEL_NAMES = {_lib.SW2D: "SW2D", _lib.SW2D_TEMP: "SW2D_TEMP", _lib.PE2D: "PE2D", _lib.PE25D: "PE25D"}


def save(path, core, step=0, time=0.0, geom=None, **extra):
    """write the core's current state (gathered from HBM) to `path` (.npz)"""
    p, u, v, t, q = core.get_state()
    out = {"model": _MODEL_NAMES[core.model], "step": step, "time": time,
           "shape": np.asarray([core.L, core.H, core.W])}
    for name, val in core.options.items():
        out["opt_" + name] = np.asarray(val)
    if core.has_ground:
        out["ground"] = core.get_ground()
    for k, a in zip("puvtq", (p, u, v, t, q)):
        if a is not None:
            out["state_" + k] = a
    if geom is not None:
        for k in _GEOM_KEYS:
            if hasattr(geom, k):
                out["geom_" + k] = np.asarray(getattr(geom, k))
    out.update({"extra_" + k: np.asarray(v) for k, v in extra.items()})
    np.savez(path, **out)


def load(path):
    """-> dict(model, step, time, state={p,u,v,t,q}, geom or None, extra)"""
    d = np.load(path, allow_pickle=False)
    L, H, W = (int(x) for x in d["shape"])
    state = {k: d["state_" + k] for k in "puvtq" if "state_" + k in d.files}
    geom = None
    if any(f.startswith("geom_") for f in d.files):
        geom = Geom(H, W, L)
        for k in _GEOM_KEYS:
            if "geom_" + k in d.files:
                a = d["geom_" + k]
                setattr(geom, k, float(a) if a.ndim == 0 else a)
    extra = {f[6:]: d[f] for f in d.files if f.startswith("extra_")}
    opts = {}
    for f in d.files:
        if f.startswith("opt_"):
            a = d[f]
            opts[f[4:]] = str(a) if a.dtype.kind in "US" else (bool(a) if a.dtype.kind == "b" else
                                                               (float(a) if a.dtype.kind == "f" else int(a)))
    return dict(model=str(d["model"]), step=int(d["step"]), time=float(d["time"]), state=state,
                geom=geom, extra=extra, shape=(L, H, W), options=opts,
                ground=d["ground"] if "ground" in d.files else None)


def restore(path, **core_kwargs):
    """-> (Core with the saved state resident, checkpoint dict)"""
    ck = load(path)
    L, H, W = ck["shape"]
    model = {v: k for k, v in _MODEL_NAMES.items()}[ck["model"]]
    kw = dict(ck["options"])            # the saved handle's options; keyword arguments override them
    kw.update(core_kwargs)
    if model == _lib.SW2D_TEMP and "q" in ck["state"] and not kw.get("tracer"):
        kw["tracer"] = _lib.TRACER_VANLEER      # files written before the options were stored
    core = Core(model, W, H, L, geom=ck["geom"], **kw)
    core.set_state(**ck["state"])
    if ck["ground"] is not None:
        core.set_ground(ck["ground"])
    return core, ck
