"""Host-side helpers shared by the choosers: method-arg parsing and the
univariate slice sampler.  Semantics (including the order in which the global
numpy RNG is consumed) follow spearmint/spearmint/util.py:26-93 so that a
seeded run proposes the same hyper-parameters as the reference."""
from __future__ import absolute_import, print_function

import re

import numpy as np
import numpy.random as npr

_SPLIT_ITEMS = re.compile(r"\s*,\s*")
_SPLIT_KV = re.compile(r"\s*=\s*")


def unpack_args(arg_string):
    """'k=v,k=v' -> {k: v} with string values (util.py:26-32).  Strings of
    length <= 1 mean "no arguments", as in the reference."""
    if len(arg_string) <= 1:
        return {}
    out = {}
    for item in _SPLIT_ITEMS.split(arg_string):
        kv = _SPLIT_KV.split(item)
        out[kv[0]] = kv[1]
    return out


class SliceSamplerError(Exception):
    pass


def _slice_along(direction, x0, logprob, sigma, step_out, max_steps_out):
    """One slice-sampling move along `direction` through x0 (util.py:35-76)."""
    def f(z):
        return logprob(direction * z + x0)

    hi = sigma * npr.rand()
    lo = hi - sigma
    level = np.log(npr.rand()) + f(0.0)
    if step_out:
        n = 0
        while f(lo) > level and n < max_steps_out:
            n += 1
            lo -= sigma
        n = 0
        while f(hi) > level and n < max_steps_out:
            n += 1
            hi += sigma
    while True:
        z = (hi - lo) * npr.rand() + lo
        lp = f(z)
        if np.isnan(lp):
            raise SliceSamplerError("Slice sampler got a NaN")
        if lp > level:
            return z * direction + x0
        if z < 0:
            lo = z
        elif z > 0:
            hi = z
        else:
            raise SliceSamplerError("Slice sampler shrank to zero!")


def slice_sample(init_x, logprob, sigma=1.0, step_out=True, max_steps_out=1000, compwise=False):
    """util.py:34-93.  compwise: one move per coordinate in a shuffled order;
    otherwise one move along a random unit direction."""
    x = np.asarray(init_x, dtype=float)
    if not x.shape:
        x = np.array([float(x)])
    dims = x.shape[0]
    if compwise:
        order = list(range(dims))
        npr.shuffle(order)
        cur = x.copy()
        for d in order:
            e = np.zeros(dims)
            e[d] = 1.0
            cur = _slice_along(e, cur, logprob, sigma, step_out, max_steps_out)
        return cur
    direction = npr.randn(dims)
    direction = direction / np.sqrt(np.sum(direction ** 2))
    return _slice_along(direction, x, logprob, sigma, step_out, max_steps_out)
