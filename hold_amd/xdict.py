"""``xdict``: the keyed output container HOLD's callers expect from ``HOLDNet.forward``.

The reference's training / rendering loops call container methods on the model output, not only ``[]``:
``model_outputs.search("index_off_surface")`` (code/src/hold/loss.py:46,55,60), ``self.model(s).detach().to("cpu")``
and ``out.search("fg_rgb.vis")`` (code/src/hold/hold.py:192-199), ``my_out_dict.prefix(...)`` / ``.merge(...)``
(code/src/hold/hold_net.py:84-89).  This is an independent implementation of that contract
(/root/reference/common/xdict.py:26-333): same method names, same semantics (duplicate-key assertions included).

``output_class()`` returns the reference's own ``common.xdict.xdict`` when the reference tree is importable (so
``isinstance`` checks in user code keep working after ``hold_amd.install()``), this class otherwise.
"""
from __future__ import annotations

import os

import numpy as np
import torch


def _map(v, fn):
    """apply fn to tensors nested in lists / tuples / dicts; everything else passes through."""
    if isinstance(v, torch.Tensor):
        return fn(v)
    if isinstance(v, list):
        return [_map(x, fn) for x in v]
    if isinstance(v, tuple):
        return tuple(_map(x, fn) for x in v)
    if isinstance(v, dict):
        return {k: _map(x, fn) for k, x in v.items()}
    return v


class xdict(dict):
    def __init__(self, mydict=None):
        super().__init__()
        if mydict is not None:
            for k, v in mydict.items():
                dict.__setitem__(self, k, v)

    # -- assignment contract: plain [] never overwrites (xdict.py:50-55,87-92)
    def __setitem__(self, key, val):
        assert key not in self, f"Key already exists {key}"
        dict.__setitem__(self, key, val)

    def overwrite(self, k, v):
        assert k in self, f"Key does not exist {k}"
        dict.__setitem__(self, k, v)

    def merge(self, dict2):
        assert isinstance(dict2, dict)
        dup = set(self.keys()) & set(dict2.keys())
        assert not dup, f"Merge failed: duplicate keys ({dup})"
        self.update(dict2)

    # -- key-space views
    def subset(self, keys):
        return type(self)({k: self[k] for k in keys})

    def search(self, keyword, replace_to=None):
        hit = {}
        for k, v in self.items():
            if keyword in k:
                hit[k if replace_to is None else k.replace(keyword, replace_to)] = v
        return type(self)(hit)

    def fuzzy_get(self, keyword):
        for k, v in self.items():
            if keyword in k:
                return v
        return None

    def rm(self, keyword, keep_list=(), verbose=False):
        kept = {}
        for k, v in self.items():
            if keyword not in k or k in keep_list:
                kept[k] = v
            elif verbose:
                print(f"Removing: {k}")
        return type(self)(kept)

    def prefix(self, text):
        return type(self)({text + k: v for k, v in self.items()})

    def postfix(self, text):
        return type(self)({k + text: v for k, v in self.items()})

    def replace_keys(self, str_src, str_tar):
        return type(self)({k.replace(str_src, str_tar): v for k, v in self.items()})

    def sorted_keys(self):
        return sorted(self.keys())

    # -- value transforms
    def mul(self, scalar):
        scalar = float(scalar) if isinstance(scalar, int) else scalar
        assert isinstance(scalar, float)
        return type(self)({k: ([x * scalar for x in v] if isinstance(v, list) else v * scalar) for k, v in self.items()})

    def apply(self, operation, criterion=None):
        return type(self)({k: operation(v) for k, v in self.items() if criterion is None or criterion(k, v)})

    def to(self, dev):
        if dev is None:
            return self

        def move(v):
            if hasattr(v, "to"):
                return v.to(dev)
            if isinstance(v, list):
                return [move(x) for x in v]
            if isinstance(v, tuple):
                return tuple(move(x) for x in v)
            if isinstance(v, dict):
                return {k: move(x) for k, x in v.items()}
            return v

        return type(self)({k: move(v) for k, v in self.items()})

    def detach(self):
        """tensors -> detached CPU copies (thing.detach_thing), the per-chunk D2H of hold.py:192."""
        return type(self)({k: _map(v, lambda t: t.cpu().detach()) for k, v in self.items()})

    def to_torch(self):
        def conv(v):
            if isinstance(v, np.ndarray):
                return torch.from_numpy(v)
            if isinstance(v, list):
                return torch.as_tensor(np.array(v)) if len(v) and not isinstance(v[0], (str, dict)) else v
            if isinstance(v, dict):
                return {k: conv(x) for k, x in v.items()}
            return v
        return type(self)({k: conv(v) for k, v in self.items()})

    def to_np(self):
        def conv(v):
            if isinstance(v, torch.Tensor):
                return v.detach().cpu().numpy()
            if isinstance(v, list):
                return np.array(v)
            if isinstance(v, dict):
                return {k: conv(x) for k, x in v.items()}
            return v
        return type(self)({k: conv(v) for k, v in self.items()})

    def tolist(self):
        def conv(v):
            if isinstance(v, (torch.Tensor, np.ndarray)):
                return v.tolist()
            if isinstance(v, dict):
                return {k: conv(x) for k, x in v.items()}
            return v
        return type(self)({k: conv(v) for k, v in self.items()})

    def has_invalid(self):
        for k, v in self.items():
            if isinstance(v, torch.Tensor) and v.is_floating_point():
                if torch.isnan(v).any():
                    print(f"{k} contains nan values")
                    return True
                if torch.isinf(v).any():
                    print(f"{k} contains inf values")
                    return True
        return False

    def print_stat(self):
        for k, v in self.items():
            if isinstance(v, (torch.Tensor, np.ndarray)):
                print(f"{k:<20}: {str(tuple(v.shape)):<30}\t{type(v)}\t{getattr(v, 'device', '')}")
            elif isinstance(v, (list, tuple)):
                print(f"{k:<20}: {len(v):<30}\t{type(v)}")
            else:
                print(f"{k:<20}: {type(v)}")

    def save(self, path, dev=None, verbose=True):
        if verbose:
            print(f"Saving to {path}")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        torch.save(self.to(dev), path)

    def to_16_bits(self):
        for k, v in list(self.items()):
            if isinstance(v, dict):
                self.overwrite(k, xdict(v).to_16_bits())
            elif isinstance(v, torch.Tensor):
                if v.dtype in (torch.float32, torch.float64):
                    self.overwrite(k, v.to(torch.float16))
                elif v.dtype == torch.int64:
                    self.overwrite(k, v.to(torch.int16))
        return self

    def to_std_precision(self):
        for k, v in list(self.items()):
            if isinstance(v, torch.Tensor):
                if v.dtype in (torch.float16, torch.float64):
                    self.overwrite(k, v.to(torch.float32))
                elif v.dtype in (torch.int16, torch.int32):
                    self.overwrite(k, v.to(torch.int64))
        return self


_OUT_CLS = None


def output_class():
    """the mapping type HOLDNet.forward returns: common.xdict.xdict if the reference tree is on sys.path."""
    global _OUT_CLS
    if _OUT_CLS is None:
        try:
            from common.xdict import xdict as ref_xdict  # type: ignore
            _OUT_CLS = ref_xdict
        except Exception:
            _OUT_CLS = xdict
    return _OUT_CLS
