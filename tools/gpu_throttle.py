#!/usr/bin/env python
"""Why is the shader clock below 2.4 GHz?  Throttler residencies of the GPU from the driver's gpu_metrics accumulators, read
through the amdsmi python binding that ships with ROCm (/opt/rocm/share/amd_smi): the firmware counts, per sampling tick, whether
the clock was held down by the package power tracker (PPT), the socket / VR / HBM thermal limits or PROCHOT.  Two snapshots
bracket a region; residency of a cause = delta(acc_<cause>) / delta(acc_counter).

    python tools/gpu_throttle.py snapshot            one snapshot as JSON (accumulators, active flags, power cap, power, clock)
    python tools/gpu_throttle.py watch SECONDS       residencies over a window (while something else runs on the GPU)
    python tools/gpu_throttle.py setcap WATTS        try to lower / restore the socket power cap (root; may be refused)

bench.py imports snapshot() / residency() for `roofline.power.throttle` (best effort: None when the binding is missing).
"""
import json
import sys
import time

_CAUSES = ("ppt_pwr", "socket_thrm", "prochot_thrm", "vr_thrm", "hbm_thrm", "gfx_clk_below_host_limit")
_state = {}


def _smi():
    if "smi" not in _state:
        if "/opt/rocm/share/amd_smi" not in sys.path:
            sys.path.insert(0, "/opt/rocm/share/amd_smi")
        import amdsmi
        amdsmi.amdsmi_init()
        handles = amdsmi.amdsmi_get_processor_handles()
        if not handles:
            raise RuntimeError("amdsmi: no GPU")
        _state["smi"], _state["handles"] = amdsmi, handles
    return _state["smi"], _state["handles"]


def _num(v):
    return v if isinstance(v, (int, float)) and not isinstance(v, bool) else None


def snapshot(index=0):
    """Accumulators + instantaneous flags of GPU `index`; raises if the binding or the driver does not provide them."""
    smi, handles = _smi()
    h = handles[index]
    v = smi.amdsmi_get_violation_status(h)
    out = {"t": time.time(), "acc_counter": _num(v.get("acc_counter"))}
    for c in _CAUSES:
        out["acc_" + c] = _num(v.get("acc_" + c))
        out["active_" + c] = v.get("active_" + c)
    for key in ("acc_gfx_clk_below_host_limit_pwr", "acc_gfx_clk_below_host_limit_thm", "acc_low_utilization"):
        rows = v.get(key)
        if rows:                                   # [partition][XCD]: keep the first partition's XCDs that report numbers
            out[key] = [x for x in rows[0] if _num(x) is not None]
    try:
        cap = smi.amdsmi_get_power_cap_info(h)
        out["power_cap"] = {k: cap.get(k) for k in ("power_cap", "default_power_cap", "min_power_cap", "max_power_cap")}
    except Exception as e:      # noqa: BLE001
        out["power_cap"] = repr(e)
    try:
        p = smi.amdsmi_get_power_info(h)
        out["power"] = {k: p.get(k) for k in ("socket_power", "current_socket_power", "average_socket_power", "power_limit")}
    except Exception as e:      # noqa: BLE001
        out["power"] = repr(e)
    try:
        c = smi.amdsmi_get_clock_info(h, smi.AmdSmiClkType.GFX)
        out["gfx_clock"] = {k: c.get(k) for k in ("clk", "min_clk", "max_clk", "clk_locked")}
    except Exception as e:      # noqa: BLE001
        out["gfx_clock"] = repr(e)
    return out


def residency(a, b):
    """Fraction of the firmware's sampling ticks between snapshots a and b in which each throttler held the clock down."""
    if not a or not b or a.get("acc_counter") is None or b.get("acc_counter") is None:
        return None
    ticks = b["acc_counter"] - a["acc_counter"]
    if ticks <= 0:
        return None
    out = {"ticks": ticks, "seconds": b["t"] - a["t"]}
    for c in _CAUSES:
        x, y = a.get("acc_" + c), b.get("acc_" + c)
        out[c] = None if x is None or y is None else (y - x) / ticks
    for key in ("acc_gfx_clk_below_host_limit_pwr", "acc_gfx_clk_below_host_limit_thm", "acc_low_utilization"):
        x, y = a.get(key), b.get(key)
        if x and y and len(x) == len(y):
            out[key[4:] + "_per_xcd"] = [(q - p) / ticks for p, q in zip(x, y)]
    return out


def set_cap(watts, index=0):
    smi, handles = _smi()
    smi.amdsmi_set_power_cap(handles[index], 0, int(watts * 1e6))


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "snapshot"
    if cmd == "snapshot":
        print(json.dumps(snapshot()))
    elif cmd == "watch":
        a = snapshot()
        time.sleep(float(sys.argv[2]))
        b = snapshot()
        print(json.dumps({"residency": residency(a, b), "start": a, "end": b}))
    elif cmd == "setcap":
        before = snapshot().get("power_cap")
        try:
            set_cap(float(sys.argv[2]))
            err = None
        except Exception as e:      # noqa: BLE001
            err = repr(e)
        print(json.dumps({"before": before, "after": snapshot().get("power_cap"), "error": err}))
    else:
        raise SystemExit(__doc__)
