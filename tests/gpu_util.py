"""Device-side plumbing for the GPU parity tests (torch = memory only)."""
import numpy as np
import torch


def to_dev(case, dev):
    d = {}
    for k, v in case.items():
        if isinstance(v, np.ndarray) and k != "table":
            d[k] = torch.from_numpy(v.copy()).to(dev)
        else:
            d[k] = v
    if "table" in case:
        base = d["pool"].data_ptr()
        t = case["table"]
        ptrs = np.where(t >= 0, base + 4 * t, 0).astype(np.int64)
        d["page_table"] = torch.from_numpy(ptrs).to(dev)
    return d


def host(t):
    torch.cuda.synchronize()
    return t.detach().cpu().numpy()
