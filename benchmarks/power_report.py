#!/usr/bin/env python3
"""Socket power / shader clock / firmware throttle accumulators while a workload runs (diagnosis tooling of rounds 4 - 5, moved out of
bench.py in round 6: it is not part of the driver's measurement).  ``bench.py --power`` puts its result into bench_detail.json.

    power_and_clock(one_pass, passes=16) -> dict | None
"""
import json
import time

import torch


def _smi(*args):
    import subprocess
    try:
        return subprocess.run(["rocm-smi", *args], capture_output=True, text=True, timeout=10).stdout
    except Exception:
        return ""


def power_limits():
    """What the board says about its own limits, read once (idle): the power cap the firmware enforces (rocm-smi --showmaxpower), the
    performance level policy and the clock range of the shader domain.  None-valued keys = the tool did not print that field here."""
    import re
    cap = re.search(r"Max Graphics Package Power \(W\): ([0-9.]+)", _smi("--showmaxpower"))
    perf = re.search(r"Performance Level: (\S+)", _smi("--showperflevel"))
    levels = [int(m) for m in re.findall(r"\b\d+: (\d+)Mhz", _smi("--showclkfrq").split("sclk")[-1].split("Supported")[0])] if "sclk" in _smi("--showclkfrq") else []
    return {"cap_w": float(cap.group(1)) if cap else None, "perf_level": perf.group(1) if perf else None,
            "sclk_levels_mhz": levels or None}


def throttle_accumulators():
    """The firmware's own throttle accounting (amd-smi metric --violation, MI300 and newer): a free-running accumulation counter and, per
    limiter, how many of its ticks were spent limited -- package power tracking (PPT), PROCHOT, socket / VR / HBM thermal.  None where the
    tool or a field is missing."""
    import subprocess
    try:
        txt = subprocess.run(["amd-smi", "metric", "--violation", "--json"], capture_output=True, text=True, timeout=15).stdout
        data = json.loads(txt)
    except Exception:
        return None
    found = {}
    def walk(node):
        if isinstance(node, dict):
            for k, v in node.items():
                kl = str(k).lower()
                if kl in ("accumulation_counter", "ppt_accumulated", "prochot_accumulated", "socket_thermal_accumulated", "vr_thermal_accumulated",
                          "hbm_thermal_accumulated") and kl not in found:
                    val = v.get("value") if isinstance(v, dict) else v
                    if isinstance(val, (int, float)):
                        found[kl] = int(val)
                walk(v)
        elif isinstance(node, list):
            for v in node:
                walk(v)
    walk(data)
    return found if "accumulation_counter" in found else None


def power_and_clock(one_pass, passes=16):
    """Socket power and shader clock while the path runs: `passes` more untimed passes with rocm-smi sampled from a side thread (the timed
    region is not touched), next to the board's power cap.  None if rocm-smi is unavailable."""
    import re
    import threading
    limits = power_limits()
    acc0 = throttle_accumulators()
    stop, out = threading.Event(), []

    def sample():
        while not stop.is_set():
            txt = _smi("--showpower", "--showclocks")
            if not txt:
                return
            p = re.search(r"Socket Graphics Package Power \(W\): ([0-9.]+)", txt)
            c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", txt)
            if p and c:
                out.append((float(p.group(1)), int(c.group(1))))
            time.sleep(0.1)
    th = threading.Thread(target=sample, daemon=True)
    th.start()
    for _ in range(passes):
        one_pass()
    torch.cuda.synchronize()
    stop.set()
    th.join(timeout=10)
    acc1 = throttle_accumulators()
    throttle = None
    if acc0 and acc1 and acc1["accumulation_counter"] > acc0["accumulation_counter"]:
        ticks = acc1["accumulation_counter"] - acc0["accumulation_counter"]
        throttle = {"ticks": ticks, "source": "amd-smi metric --violation, accumulator deltas over the sampled passes (share = ticks spent limited / ticks)"}
        for k in ("ppt", "prochot", "socket_thermal", "vr_thermal", "hbm_thermal"):
            if k + "_accumulated" in acc0 and k + "_accumulated" in acc1:
                throttle[k + "_share"] = round((acc1[k + "_accumulated"] - acc0[k + "_accumulated"]) / ticks, 4)
    body = out[len(out) // 3:-1] if len(out) > 5 else out          # (the first second still rides the boost after an idle gap: the steady state follows)
    if not body:
        return None
    pw, ck = [p for p, _ in body], [c for _, c in body]
    mean_w = sum(pw) / len(pw)
    cap = limits["cap_w"]
    head = None if cap is None else round(cap - mean_w, 1)
    if throttle and throttle.get("ppt_share") is not None:
        other = 100 * max(throttle.get(k + "_share") or 0 for k in ("prochot", "socket_thermal", "vr_thermal", "hbm_thermal"))
        facts = (f"the firmware's own accounting (amd-smi throttle accumulators over the sampled passes): the package-power limiter (PPT) was throttling on "
                 f"{100 * throttle['ppt_share']:.0f} % of the ticks, the thermal and PROCHOT limiters on {other:.0f} %, while rocm-smi's averaged socket power read "
                 f"{mean_w:.0f} W of a {cap if cap is not None else float('nan'):.0f} W cap and the shader clock sat at {round(sum(ck) / len(ck))} of 2400 MHz (no clock locked, "
                 f"performance level {limits['perf_level']})")
        if throttle["ppt_share"] >= 0.2:
            verdict = facts + (": power management is the one active limiter of this path on this box -- it acts on a faster power estimate than the averaged "
                               "reading, and a controller that holds the part at its limit reports a violation only on the ticks where the estimate exceeds it")
        elif throttle["ppt_share"] >= 0.05:
            verdict = facts + (": the power limiter engages intermittently and is the only limiter that reports any activity; boxes of the pool sit at 2 - 30 % PPT "
                               "ticks and 1.98 - 2.19 GHz under this same path (DESIGN.md section 5 A')")
        else:
            verdict = facts + (": NO limiter reports activity worth the name on this box, yet the clock stays below its top level -- what holds it there is not visible "
                               "in these counters (DESIGN.md section 5 A')")
    elif head is None:
        verdict = "no power cap reported by rocm-smi on this box: the clock figure stands alone"
    elif head <= 100:
        verdict = (f"average draw within {head:.0f} W of the {cap:.0f} W cap (peaks at or above it) with the shader clock below its top level: "
                   "the firmware's power management sets the pace of this path on this box")
    else:
        verdict = (f"average draw {head:.0f} W under the {cap:.0f} W cap: on THIS box the power cap is not what holds the clock at "
                   f"{round(sum(ck) / len(ck))} MHz -- see DESIGN.md (power section) for what the samples do and do not show")
    return {"socket_power_w": round(mean_w, 1), "socket_power_w_min": min(pw), "socket_power_w_max": max(pw),
            "shader_clock_mhz": round(sum(ck) / len(ck)), "shader_clock_mhz_min": min(ck), "shader_clock_mhz_max": max(ck),
            "max_shader_clock_mhz": max(limits["sclk_levels_mhz"]) if limits["sclk_levels_mhz"] else 2400, "samples": len(body),
            "cap_w": cap, "headroom_w": head, "perf_level": limits["perf_level"], "sclk_levels_mhz": limits["sclk_levels_mhz"],
            "throttle": throttle, "reading": verdict,
            "note": "rocm-smi sampled every ~0.15 s (a ~1 ms-averaged register, not an energy counter) during extra untimed passes of this mode right "
                    "after its timed region; cap from rocm-smi --showmaxpower"}
