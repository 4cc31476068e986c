"""Shared helpers for the CPU (oracle) and GPU (product) parity tests."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def H(s):
    return int(s, 16)


def tup(e):
    return tuple(tup(x) if isinstance(x, list) else x for x in e)


def evalh_case(zo):
    """Rebuilds the evalh.json circuit: returns (domain, kwargs-for-EvalhPack.build, expected h ints).
    Cosets are computed with the C oracle's coeff_to_extended (itself pinned by domain.json)."""
    import halo2_zkcert_amd.evaluator as ev

    g = load("evalh.json")
    dom = zo.Domain(g["degree"], g["k"])
    gates = ev.build_custom_gates([tup(x) for x in g["gates"]])
    lookups = [ev.build_lookup([tup(e) for e in i], [tup(e) for e in t]) for i, t in g["lookups"]]
    cos = {}
    for kk, v in g["polys"].items():
        cos[kk] = [dom.coeff_to_extended(zo.fr_arr_from_ints([H(x) for x in p])) for p in v]
    l0, ll, la = dom.l_cosets(g["blinding_factors"])
    zeta, delta = zo.fr_constants()
    ch = g["challenges"]
    kw = dict(k=g["k"], extended_k=dom.extended_k, cs_degree=g["degree"], blinding_factors=g["blinding_factors"],
              extended_omega=dom.extended_omega, g_coset=dom.g_coset, delta=delta,
              beta=zo.fr_from_int(H(ch["beta"])), gamma=zo.fr_from_int(H(ch["gamma"])),
              theta=zo.fr_from_int(H(ch["theta"])), y=zo.fr_from_int(H(ch["y"])),
              fixed=cos["fixed"], advice=cos["advice"], instance=cos["instance"],
              challenges=zo.fr_arr_from_ints([H(x) for x in ch["challenges"]]),
              l0=l0, l_last=ll, l_active=la, gates_graph=gates,
              perm_columns=[tuple(c) for c in g["perm_columns"]], sigma=cos["sigma"], perm_z=cos["perm_z"],
              lookup_graphs=lookups, lookup_z=cos["lookup_z"], lookup_a=cos["lookup_a"], lookup_s=cos["lookup_s"],
              to_mont=lambda xs: zo.fr_arr_from_ints(xs))
    return dom, kw, [H(x) for x in g["h"]]
