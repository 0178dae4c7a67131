"""State I/O: the reference keeps its state in RAM only; long device-resident
runs want a restart file.  One `.npz` holds the prognostic tuple, the model tag and the
geometry tables, so a run can be resumed bit-for-bit on the same or another GPU count."""
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
    return dict(model=str(d["model"]), step=int(d["step"]), time=float(d["time"]), state=state,
                geom=geom, extra=extra, shape=(L, H, W))


def restore(path, **core_kwargs):
    """-> (Core with the saved state resident, checkpoint dict)"""
    ck = load(path)
    L, H, W = ck["shape"]
    model = {v: k for k, v in _MODEL_NAMES.items()}[ck["model"]]
    if model == _lib.SW2D_TEMP and "q" in ck["state"] and "tracer" not in core_kwargs:
        core_kwargs["tracer"] = _lib.TRACER_VANLEER
    core = Core(model, W, H, L, geom=ck["geom"], **core_kwargs)
    core.set_state(**ck["state"])
    return core, ck
