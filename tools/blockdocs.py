"""The block descriptions of the module as JSON, one object per registered path -- the information PothosUtil's doc parser extracts from
the |PothosDoc markup when the module is built with ENABLE_DOCS (keys named as in Pothos's block-description JSON: path, name,
categories, keywords, aliases, params [key, name, desc, default, options, widgetType, widgetKwargs, preview, tab, units], args, calls
[type, name, args]).  For a maintainer to diff against `PothosUtil --doc-parse` output, and for the tests.  CPU only.
    python tools/blockdocs.py [path ...]        e.g.  python tools/blockdocs.py /comms/fir_filter
"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_blockdocs_cpu import our_docs      # the one parser of the markup in this repository


def widget(text):
    """'SpinBox(minimum=1)' -> ('SpinBox', {'minimum': '1'})"""
    if not text:
        return None, {}
    m = re.match(r"(\w+)\((.*)\)$", text.strip())
    if not m:
        return text.strip(), {}
    kw = {}
    for part in filter(None, (p.strip() for p in m.group(2).split(","))):
        k, _, v = part.partition("=")
        kw[k.strip()] = v.strip()
    return m.group(1), kw


def as_json(d):
    params = []
    for key in d["order"]:
        p = d["params"][key]
        wt, wk = widget(p["widget"])
        params.append({"key": key, "name": p["name"] or key, "desc": [l for l in p["desc"] if l.strip()], "default": p["default"],
                       "options": [{"value": o} for o in p["options"]], "widgetType": wt, "widgetKwargs": wk,
                       "preview": p["preview"], "tab": p["tab"], "units": p["units"]})
    return {"path": d["factory"][0], "name": d["title"], "categories": d["category"], "keywords": d["keywords"].split(), "aliases": d["alias"],
            "docs": [l for l in d["prose"] if l.strip()], "params": params, "args": d["factory"][1],
            "calls": [{"type": kind, "name": fn, "args": keys} for kind, fn, keys in d["calls"]]}


if __name__ == "__main__":
    docs = our_docs()
    want = sys.argv[1:] or sorted(docs)
    print(json.dumps([as_json(docs[p]) for p in want], indent=1))
