#!/usr/bin/env python3
"""Records the PUBLIC CALL SURFACE of the reference's hot-path modules -- names, parameter names, kinds and defaults of every public
function, class and public method of k_diffusion/{sampling,layers,external,config}.py -- into tests/golden/signatures.json, so that
"drop-in" is a checked statement (tests/test_host_cpu.py::test_public_signatures_match_the_reference) wherever the suite runs.
Run in the build container (the reference is imported through oracle/ref_import's stubs):  python oracle/make_golden_signatures.py"""
import inspect
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import ref_import  # noqa: E402

MODULES = ("sampling", "layers", "external", "config")


def params(f):
    out = []
    for p in inspect.signature(f).parameters.values():
        d = p.default
        if d is inspect.Parameter.empty:
            dv = "<required>"
        elif callable(d):
            dv = "<callable>"
        else:
            dv = repr(d)
        out.append([p.name, p.kind.name, dv])
    return out


def surface(module):
    res = {}
    for name, obj in vars(module).items():
        if name.startswith("_") or getattr(obj, "__module__", None) != module.__name__:
            continue
        if inspect.isfunction(obj):
            res[name] = {"kind": "function", "params": params(obj)}
        elif inspect.isclass(obj):
            methods = {m: params(f) for m, f in vars(obj).items()
                       if inspect.isfunction(f) and (not m.startswith("_") or m in ("__init__", "__call__"))}
            res[name] = {"kind": "class", "methods": methods}
    return res


def main():
    R = ref_import.load()
    out = {m: surface(getattr(R, m)) for m in MODULES}
    path = os.path.join(REPO, "tests", "golden", "signatures.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(path, {m: len(v) for m, v in out.items()})


if __name__ == "__main__":
    main()
