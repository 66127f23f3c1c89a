"""Collects the measured HIP-vs-reference / HIP-vs-oracle errors of the -m gpu parity tests and writes them as one JSON
artifact at session end (gpurun_out/r06_parity.json on the GPU box; copied to profiles/ and committed): per case and
chain row the error, the reference's own sensitivity `sens` to a relative 1e-6 UNet perturbation (stored in the golden
fixtures by tools/make_golden.py), their ratio and the bound that was asserted."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_RECORDS = []
_TRACKED = {}


def track(name, value, total=None, floor=None, note=None):
    """A headline parity NUMBER kept at the top of the artifact, e.g. how many trajectories match the reference end
    to end within 1e-3; `floor` is what the test asserted."""
    _TRACKED[name] = {"value": value, "total": total, "floor_asserted": floor, "note": note}


def record(test, case, row, err, sens=None, bound=None, note=None, **extra):
    r = {"test": test, "case": case, "row": None if row is None else int(row), "err": float(err)}
    r.update({k: (float(v) if isinstance(v, (int, float)) else v) for k, v in extra.items()})
    if sens is not None:
        r["sens"] = float(sens)
        r["err_over_sens"] = float(err) / float(sens) if float(sens) > 0 else None
    if bound is not None:
        r["bound"] = float(bound)
    if note:
        r["note"] = note
    _RECORDS.append(r)


def flush():
    if not _RECORDS:
        return None
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, "r06_parity.json")
    summary = {}
    for r in _RECORDS:
        s = summary.setdefault(r["test"], {"n": 0, "max_err": 0.0, "max_err_over_sens": None})
        s["n"] += 1
        if r.get("flip"):
            s["branch_flips"] = s.get("branch_flips", 0) + 1
        s["max_err"] = max(s["max_err"], r["err"])
        if r.get("err_over_sens") is not None:
            s["max_err_over_sens"] = max(s["max_err_over_sens"] or 0.0, r["err_over_sens"])
    with open(path, "w") as f:
        json.dump({"tracked": _TRACKED, "summary": summary, "records": _RECORDS}, f, indent=1)
    return path
