#!/usr/bin/env python
"""Write tests/golden/signatures.json: the parameter names, in order, of every function and
Module (constructor and ``forward``) of the LIVE reference that this package mirrors.

Run in the build container only (the reference never travels to the GPU box):

    PYTHONPATH=/root/reference/src python tests/golden/make_signatures.py

The fixture is DATA (names and whether a default exists); tests/test_host_logic.py checks that
a call written against the reference -- positional or by keyword -- binds here too.
"""
import inspect
import json
import os
import sys

REF = os.environ.get("PDT_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(REF, "src"))
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "pydrobert-pytorch_amd"))

import pydrobert.torch.functional as RF  # noqa: E402
import pydrobert.torch.modules as RM  # noqa: E402

import pydrobert_amd.functional as F  # noqa: E402
import pydrobert_amd.modules as M  # noqa: E402


def params(fn):
    fn = getattr(fn, "__wrapped__", fn)
    if isinstance(fn, (getattr(__import__("torch").jit, "ScriptFunction", ()),)):
        raise TypeError
    out = []
    for p in inspect.signature(fn).parameters.values():
        if p.name == "self":
            continue
        out.append([p.name, p.default is not inspect.Parameter.empty, p.kind.name])
    return out


sig = {"functional": {}, "modules": {}}
for name in F.__all__:
    if hasattr(RF, name):  # (ctc_prefix_search is this package's own entry point)
        sig["functional"][name] = params(getattr(RF, name))
for name in M.__all__:
    cls = getattr(RM, name)
    sig["modules"][name] = {"__init__": params(cls.__init__), "forward": params(cls.forward)}
with open(os.path.join(HERE, "signatures.json"), "w") as f:
    json.dump(sig, f, indent=1, sort_keys=True)
print("wrote signatures.json:", len(sig["functional"]), "functions,", len(sig["modules"]), "modules")
