"""Helpers for the -m gpu tests: fixtures -> device tensors -> plans."""
import numpy as np
import torch

from tests.util import Golden

RENAME = {
    "sqrtG_new": "sqrtG", "h_contra_new": "h_contra", "christoffel": "christoffel", "inv_dzdeta_new": "inv_dzdeta",
    "sqrtG_itf_i_new": "sqrtG_itf_i", "sqrtG_itf_j_new": "sqrtG_itf_j", "sqrtG_itf_k_new": "sqrtG_itf_k",
    "h_contra_itf_i_new": "h_contra_itf_i", "h_contra_itf_j_new": "h_contra_itf_j",
    "h_contra_itf_k_new": "h_contra_itf_k", "damp_coef": "damp_coef", "damp_uref": "damp_uref",
}


def device_metric(g: Golden, p: int, dev="cuda:0"):
    m = {}
    for k, v in g.metric(p).items():
        if k in RENAME:
            m[RENAME[k]] = torch.from_numpy(np.ascontiguousarray(v)).to(dev)
    m["boundary_sn"] = torch.from_numpy(np.ascontiguousarray(g[f"p{p}/geom/boundary_sn_new"][:, 0, :]).reshape(-1)).to(dev)
    m["boundary_we"] = torch.from_numpy(np.ascontiguousarray(g[f"p{p}/geom/boundary_we_new"][:, 0, :]).reshape(-1)).to(dev)
    return m


def make_plan(g: Golden, p: int, dtype=torch.float64, dev="cuda:0"):
    from wxfactory_amd.rhs_euler3d import Euler3DPlan

    return Euler3DPlan(g.n, g.H, g.V, g.case, p, g.ops, device_metric(g, p, dev), dtype=dtype)


def to_dev(a, dev="cuda:0"):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
