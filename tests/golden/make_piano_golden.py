#!/usr/bin/env python
"""Runs the REFERENCE's own piano builder and dumps what it builds -> tests/golden/piano_layout.json.

`robopianist/models/piano/piano_mjcf.py:build()` is pure Python over a dm_control `mjcf` element
tree; dm_control is not installable here, so the module is imported BY FILE PATH against a small
recording stand-in of `dm_control.mjcf` (attribute get/set, `.add(tag, **kw)`), with
`piano_constants.py` loaded by path as well.  The dump is the resolved per-key layout (default
classes applied) and every constant of piano_constants.py and music/constants.py.  Nothing from the
reference is copied into the repo: this script only READS /root/reference (or $RP_REFERENCE) at
generation time; tests/test_model.py compares the in-tree piano model with the committed JSON, and
re-runs this script when the reference is present to check that the JSON is current.

Usage:  python tests/golden/make_piano_golden.py [--check]
"""
import importlib.util
import json
import os
import sys
import types

REF = os.environ.get("RP_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "piano_layout.json")


class Element:
    """Recording stand-in of a dm_control mjcf element."""

    def __init__(self, tag, **attrs):
        object.__setattr__(self, "tag", tag)
        object.__setattr__(self, "attrs", dict(attrs))
        object.__setattr__(self, "children", [])
        object.__setattr__(self, "singletons", {})

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        if name in self.attrs:
            return self.attrs[name]
        if name not in self.singletons:          # nested singleton element (root.default.geom, ...)
            self.singletons[name] = Element(name)
        return self.singletons[name]

    def __setattr__(self, name, value):
        self.attrs[name] = list(value) if isinstance(value, (list, tuple)) else value

    def add(self, tag, **attrs):
        e = Element(tag, **{k: (list(v) if isinstance(v, (list, tuple)) else v) for k, v in attrs.items()})
        self.children.append(e)
        return e


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _public(mod):
    out = {}
    for k, v in vars(mod).items():
        if k.isupper() and isinstance(v, (int, float, list, tuple, str)):
            out[k] = list(v) if isinstance(v, tuple) else v
    return out


def generate():
    consts = _load("rp_ref_piano_constants", os.path.join(REF, "robopianist/models/piano/piano_constants.py"))
    music_consts = _load("rp_ref_music_constants", os.path.join(REF, "robopianist/music/constants.py"))
    # stand-ins for the imports of piano_mjcf.py
    dm = types.ModuleType("dm_control"); mj = types.ModuleType("dm_control.mjcf")
    mj.RootElement = lambda: Element("mujoco")
    dm.mjcf = mj
    mu = types.ModuleType("mujoco_utils"); mt = types.ModuleType("mujoco_utils.types")
    mt.MjcfRootElement = Element
    mu.types = mt
    pkgs = {"dm_control": dm, "dm_control.mjcf": mj, "mujoco_utils": mu, "mujoco_utils.types": mt}
    for n in ("robopianist", "robopianist.models", "robopianist.models.piano"):
        pkgs[n] = types.ModuleType(n)
    pkgs["robopianist.models.piano"].piano_constants = consts
    pkgs["robopianist.models.piano.piano_constants"] = consts
    saved = {k: sys.modules.get(k) for k in pkgs}
    sys.modules.update(pkgs)
    try:
        pm = _load("rp_ref_piano_mjcf", os.path.join(REF, "robopianist/models/piano/piano_mjcf.py"))
        roots = {False: pm.build(add_actuators=False), True: pm.build(add_actuators=True)}
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v

    def resolved(root, add_actuators):
        dflt = root.singletons["default"]
        classes = {c.attrs["dclass"]: c for c in dflt.children}

        def attrs_of(kind, elem):
            a = dict(dflt.singletons[kind].attrs) if kind in dflt.singletons else {}
            dc = elem.attrs.get("dclass")
            if dc and kind in classes[dc].singletons:
                a.update(classes[dc].singletons[kind].attrs)
            a.update({k: v for k, v in elem.attrs.items()})
            for cosmetic in ("rgba", "material", "group", "dclass"):
                a.pop(cosmetic, None)
            return a
        bodies = []
        for b in root.singletons["worldbody"].children:
            rec = {"name": b.attrs["name"], "pos": b.attrs["pos"]}
            for ch in b.children:
                rec[ch.tag] = attrs_of(ch.tag, ch)
            bodies.append(rec)
        acts = [attrs_of("general", a) for a in root.singletons["actuator"].children] if add_actuators else []
        return {"compiler": dict(root.singletons["compiler"].attrs), "bodies": bodies, "actuators": acts}

    return {
        "source": "generated by tests/golden/make_piano_golden.py from robopianist/models/piano/piano_mjcf.py, "
                  "piano_constants.py and music/constants.py of the reference checkout",
        "piano_constants": _public(consts),
        "music_constants": {k: v for k, v in _public(music_consts).items()},
        "music_notes": list(music_consts.NOTES),
        "piano": resolved(roots[False], False),
        "piano_with_actuators": resolved(roots[True], True),
    }


if __name__ == "__main__":
    data = generate()
    text = json.dumps(data, sort_keys=True).replace('{"geom"', '\n{"geom"').replace('{"biasprm"', '\n{"biasprm"') + "\n"
    if "--check" in sys.argv:
        ok = open(OUT).read() == text
        print("piano_layout.json is", "current" if ok else "STALE")
        sys.exit(0 if ok else 1)
    open(OUT, "w").write(text)
    print("wrote", OUT, len(text), "bytes;", len(data["piano"]["bodies"]), "bodies")
