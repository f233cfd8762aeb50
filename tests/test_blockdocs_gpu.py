"""Every block description (|PothosDoc markup, tests/test_blockdocs_cpu.py) against the CONSTRUCTED blocks: what a topology saved by
the Pothos GUI does when it is loaded -- make(path, factory properties), then every |initializer and |setter with its property,
here with the descriptions' own defaults."""
import ast

import numpy as np
import pytest

from tests.test_blockdocs_cpu import our_docs

pytestmark = pytest.mark.gpu


def literal(text):
    """a |default / |option value as Python: "ADD" -> 'ADD', 1e6 -> 1000000.0, [] -> [], false -> False"""
    t = text.strip().replace("\\[", "[").replace("\\]", "]")
    if t in ("true", "false"):
        return t == "true"
    return ast.literal_eval(t)


def default_of(p):
    if p["default"] is not None:
        return literal(p["default"])
    assert p["options"], "a parameter without |default must have |option lines"
    return literal(p["options"][0])


def test_every_description_instantiates_and_applies_its_defaults():
    from pothoscomms_amd import blocks as B
    docs = our_docs()
    assert len(docs) >= 13
    for path, d in sorted(docs.items()):
        fargs = [default_of(d["params"][k]) for k in d["factory"][1]]
        for p in [path] + d["alias"]:
            blk = B.make(p, *fargs) if fargs else B.make(p)
            calls = blk.calls()
            for kind, fn, keys in d["calls"]:
                assert calls.get(fn) == 1, "%s: |%s %s: the block registers %r" % (p, kind, fn, calls.get(fn))
                v = default_of(d["params"][keys[0]])
                if isinstance(v, list) and fn in ("setTaps", "setWindowArgs"):
                    v = np.asarray(v, dtype=np.float64)
                if isinstance(v, (int, float)) and not isinstance(v, bool) and fn in (
                        "setPhase", "setFactor", "setSampleRate", "setFrequencyLower", "setFrequencyUpper", "setBandwidthTrans", "setAlpha",
                        "setStopDB", "setPassDB", "setGain"):
                    v = float(v)
                blk.call(fn, v)
            if "getDevice" in calls:
                assert blk.call("getDevice") == 0 and blk.call("getPortSlabBytes") == 64 << 20
            blk.close()


def test_fir_description_options_are_all_accepted():
    """every |option of the FIR's enumerated parameters is a value the named call takes"""
    from pothoscomms_amd import blocks as B
    d = our_docs()["/comms/fir_filter"]
    for taps_type in [literal(o) for o in d["params"]["tapsType"]["options"]]:
        blk = B.make("/comms/fir_filter", "complex_float32", taps_type)
        for o in d["params"]["kernel"]["options"]:
            blk.call("setKernel", literal(o))
        for o in d["params"]["waitTaps"]["options"]:
            blk.call("setWaitTaps", literal(o))
        blk.close()
    d = our_docs()["/comms/arithmetic"]
    for o in d["params"]["operation"]["options"]:
        B.make("/comms/arithmetic", "float32", literal(o)).close()
    d = our_docs()["/comms/fir_designer"]
    blk = B.make("/comms/fir_designer")
    for key, fn in (("type", "setFilterType"), ("band", "setBandType"), ("window", "setWindowType")):
        for o in d["params"][key]["options"]:
            blk.call(fn, literal(o))
    blk.close()
