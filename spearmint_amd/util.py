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


# ---------------------------------------------------------------------------
# Speculative (batched) slice sampling.
#
# The sampler is inherently sequential, but which points it WOULD evaluate next
# does not depend on the log-probabilities themselves as long as the answers are
# "keep going": stepping out visits lower - k*sigma / upper + k*sigma, and a
# rejected shrink proposal moves the bracket by the SIGN of the proposal only.
# So a batch of the next few points can be evaluated in one call (on the GPU a
# batch of factorisations costs the latency of one) and then consumed in the
# reference's order.  Random numbers are drawn ahead of their use (_Uniforms) and the
# global numpy RNG is left in exactly the state the reference would have reached
# (util.py:40-57: one rand() per shrink proposal), so a seeded run is bit-for-bit
# the same Markov chain.
# ---------------------------------------------------------------------------
class _Uniforms(object):
    """The stream of npr.rand() with look-ahead.  peek(n) shows the next n numbers without consuming them, take()
    consumes one; close() leaves the global generator exactly where as many plain npr.rand() calls as were taken
    would have left it.  One state copy per slice_sample_batched call instead of four per move (get_state /
    set_state copy the 2.5 KB Mersenne-Twister state: 20-45 us each, which was most of a move's host time)."""

    def __init__(self):
        self.state0 = npr.get_state()
        self.buf = []        # drawn from the generator, not yet taken
        self.taken = 0

    def peek(self, n):
        while len(self.buf) < n:
            self.buf.append(npr.rand())
        return self.buf[:n]

    def take(self):
        self.taken += 1
        return self.buf.pop(0) if self.buf else npr.rand()

    def close(self):
        if self.buf:         # the generator is ahead of the consumer: back to the start, forward by what was taken
            npr.set_state(self.state0)
            for _ in range(self.taken):
                npr.rand()
            self.buf = []


# brackets whose shrink proposals ride in a move's first batch: the two most probable ones, the second only from a 15 % chance
# (three or four cut the calls by another 6 % and fill the 32-row call: measured, not taken; csrc/spx_sampler.hip plans the same way)
_MAX_SCEN = 2
_MIN_SCEN_P = 0.15


class _Plan(object):
    """What the first batch of a move evaluates: zs = [0, lo, hi] + the two step-out ladders + the shrink proposals of
    the (at most two) most probable brackets; spec = [(bracket, offset into zs, uniforms used, proposals)]."""
    __slots__ = ("zs", "lo", "hi", "lo_wide", "hi_wide", "spec", "n_lad")


def _ladder(start, step, first, count):
    # positions by repeated addition, exactly like the reference's `lower -= sigma` / `upper += sigma`
    out, p = [], start
    for _ in range(first):
        p = p + step
    for _ in range(count):
        out.append(p)
        p = p + step
    return out


def _propose(l, h, us):
    # shrink proposals under the assumption that each one is rejected (bracket update by sign only)
    zlist = []
    for u in us:
        z = (h - l) * u + l
        zlist.append(z)
        if z < 0:
            l = z
        elif z > 0:
            h = z
        else:
            break   # the reference raises on z == 0 if this proposal is reached and rejected
    return zlist


def _plan_move(direction, x0, u_hi, peek, sigma, step_out, lookahead, adm, hist):
    """Plan the first batch of the move along `direction` through x0 whose upper edge is drawn with `u_hi`;
    `peek(n)` shows the n uniforms that follow the move's slice-level uniform (its shrink proposals' numbers).
    Pure: draws nothing, evaluates nothing -- so it also plans a move that has not started yet (`follow`)."""
    pl = _Plan()
    hi = sigma * u_hi
    lo = hi - sigma
    pl.lo, pl.hi = lo, hi
    # f(0), f(lo), f(hi) are always evaluated by the reference; add the next step-outs speculatively.
    zs = [0.0, lo, hi]
    lo_lad = _ladder(lo, -sigma, 1, lookahead) if step_out else []
    hi_lad = _ladder(hi, sigma, 1, lookahead) if step_out else []
    pl.n_lad = lookahead if step_out else 0
    zs += lo_lad + hi_lad
    # Stepping out draws no random numbers, so the uniforms of the first shrink proposals are already determined;
    # only the bracket they are scaled to is not.  Each end of the bracket has two likely fates: it stays where it
    # was drawn (its log-probability is below the slice level), or it steps out until the first point outside the
    # priors' support (-inf a priori: the walk stops there whatever the data say) -- e.g. a length scale the data push
    # against the upper end of its top-hat prior.  The proposals of the (at most two) most probable brackets, by the
    # frequencies of earlier moves, ride in this same batch and are used only if the walk below confirms the bracket;
    # any other outcome costs a second batch, as stepping out always did before.
    def options(first, lad, key):
        """[(position, probability)] for one end of the bracket."""
        if adm is None or not step_out:
            return [(first, 1.0)], None
        if not adm(direction * first + x0):
            return [(first, 1.0)], None                      # stops here a priori
        wide = None
        for z in lad:
            if not adm(direction * z + x0):
                wide = z
                break
        if wide is None:
            return [(first, 1.0)], None
        n_stay, n_wide, n_other = hist.setdefault(key, [1.0, 1.0, 0.0]) if hist is not None else (1.0, 1.0, 0.0)
        tot = n_stay + n_wide + n_other
        return [(first, n_stay / tot), (wide, n_wide / tot)], wide

    lo_opts, pl.lo_wide = options(lo, lo_lad, "lo")
    hi_opts, pl.hi_wide = options(hi, hi_lad, "hi")
    combos = sorted(((p_l * p_h, (zl, zh)) for zl, p_l in lo_opts for zh, p_h in hi_opts), key=lambda c: -c[0])
    scen = [combos[0][1]]
    for c in combos[1:_MAX_SCEN]:
        if c[0] >= _MIN_SCEN_P and lookahead >= 2:
            scen.append(c[1])
    if len(scen) == 1:
        counts = [lookahead]
    elif len(scen) == 2:
        if combos[0][0] > 2.0 * combos[1][0]:
            counts = [lookahead - max(1, lookahead // 3), max(1, lookahead // 3)]
        else:
            counts = [lookahead - lookahead // 2, lookahead // 2]
    else:
        counts = [lookahead - lookahead // 2] + [max(1, lookahead // 3)] * (len(scen) - 1)
    us = peek(max(counts))
    pl.spec = []             # (bracket, offset into zs, number of uniforms, number of proposals)
    for br, cnt in zip(scen, counts):
        zl = _propose(br[0], br[1], us[:cnt])
        pl.spec.append((br, len(zs), cnt, len(zl)))
        zs += zl
    pl.zs = zs
    return pl


def _slice_along_batched(direction, x0, logprob_many, sigma, step_out, max_steps_out, lookahead, dim=None, rng=None,
                         follow=None):
    own_rng = rng is None
    if own_rng:
        rng = _Uniforms()
    try:
        return _slice_along_batched_impl(direction, x0, logprob_many, sigma, step_out, max_steps_out, lookahead, dim, rng,
                                         follow)
    finally:
        if own_rng:
            rng.close()


def _slice_along_batched_impl(direction, x0, logprob_many, sigma, step_out, max_steps_out, lookahead, dim, rng, follow=None):
    # Optional attributes of logprob_many that let the sampler plan its speculative batch:
    #   admissible(x)  False where the log-probability is -inf a priori (outside the priors' support) -- known
    #                  without evaluating the GP;
    #   history        a dict the caller keeps across moves: how often each end of the bracket stayed where it was
    #                  drawn / stepped out all the way to the edge of the support (one record per kind of move:
    #                  keeping one per coordinate `dim` learns too slowly -- 1007 instead of 968 calls per next());
    #   submit(points, extras) -> _LazyValues   the lazy form: nothing is evaluated until the sampler asks for a value
    #                  the evaluator does not already hold; then every missing point of the batch AND the `extras`
    #                  (points the NEXT move would evaluate if this move ends as guessed) go out in one call.
    adm = getattr(logprob_many, "admissible", None)
    hist = getattr(logprob_many, "history", None)
    submit = getattr(logprob_many, "submit", None)

    def many(zs, extras=()):
        pts = [direction * z + x0 for z in zs]
        if submit is not None:
            return submit(pts, extras)
        return logprob_many(pts)

    def follow_points(zlist):
        """Cross-move speculation: if proposal j of `zlist` is accepted, this move has consumed j + 1 of the uniforms
        ahead and the next move (direction follow[0]) starts at that point with the numbers after them -- so its edges,
        the ladder points inside the support and its first proposals are already determined.  Returned as extra
        points for the same call; the next move finds them in the evaluator's memo or evaluates them itself."""
        if follow is None or submit is None:
            return []
        ndir, la2, nhyp = follow
        out = []
        for j in range(min(nhyp, len(zlist))):
            x_new = zlist[j] * direction + x0           # what this move returns if proposal j is accepted
            if adm is not None and not adm(x_new):
                continue                                 # -inf: cannot be accepted
            off = j + 1
            u_hi2 = rng.peek(off + 1)[off]
            p2 = _plan_move(ndir, x_new, u_hi2, lambda n, o=off + 2: rng.peek(o + n)[o:o + n], sigma, step_out, la2, adm, hist)
            out.extend(ndir * z + x_new for z in p2.zs[1:])
        return out

    u_hi = rng.take()
    u_level = rng.take()
    plan = _plan_move(direction, x0, u_hi, rng.peek, sigma, step_out, lookahead, adm, hist)
    zs, lo, hi, spec = plan.zs, plan.lo, plan.hi, plan.spec
    extras = follow_points(zs[spec[0][1]:spec[0][1] + spec[0][3]]) if spec else []
    vals = many(zs, extras)
    level = np.log(u_level) + vals.get(0)
    lo_init, hi_init = lo, hi
    if step_out:
        lo_cache = {0: (lo, vals.get(1))}
        hi_cache = {0: (hi, vals.get(2))}
        for k in range(lookahead):
            lo_cache[k + 1] = (zs[3 + k], vals.peek(3 + k))
            hi_cache[k + 1] = (zs[3 + lookahead + k], vals.peek(3 + lookahead + k))

        def walk(start, step, cache):
            n = 0
            while True:
                if n not in cache:   # beyond the speculation window: fetch the next window
                    pos = _ladder(cache[n - 1][0], step, 1, lookahead)
                    more = many(pos)
                    for j in range(lookahead):
                        cache[n + j] = (pos[j], more.peek(j))
                z, v = cache[n]
                v = v() if callable(v) else v
                if not (v > level and n < max_steps_out):
                    return z
                n += 1
        lo = walk(lo, -sigma, lo_cache)
        hi = walk(hi, sigma, hi_cache)
    if hist is not None:
        for key, wide, z0, z1 in (("lo", plan.lo_wide, lo_init, lo), ("hi", plan.hi_wide, hi_init, hi)):
            if wide is not None:
                hist[key][0 if z1 == z0 else (1 if z1 == wide else 2)] += 1.0
    hit = None
    for br, at, nu, cnt in spec:
        if (lo, hi) == br:
            hit = (at, nu, cnt)      # the speculated proposals for this bracket are the real ones
    while True:
        if hit is not None:
            base, nu, cnt = hit
            zlist = _propose(lo, hi, rng.peek(nu))    # same numbers, same bracket: already evaluated
            batch = vals
            hit = None
        else:
            zlist = _propose(lo, hi, rng.peek(lookahead))
            batch, base = many(zlist, follow_points(zlist)), 0
        for k, z in enumerate(zlist):
            rng.take()                          # the reference draws one number per proposal it reaches (util.py:40-57)
            lp = batch.get(base + k)
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
        # every proposal of this batch rejected and consumed; continue with the shrunk bracket


class _LazyValues(object):
    """Results of a speculative batch.  get(k) is an evaluation the reference performs (errors
    such as a non-PD covariance surface there, in order); peek(k) defers that decision.
    With `missing` / `fill` the batch itself is deferred: the first get() of an entry in `missing` calls
    fill(self), which evaluates every missing entry (one GPU call) and stores values / errors in place."""

    def __init__(self, values, errors, missing=None, fill=None):
        self.values, self.errors = values, errors
        self.missing = missing if missing else ()
        self.fill = fill

    def get(self, k):
        if self.missing and k in self.missing:
            self.fill(self)
        if self.errors[k] is not None:
            raise self.errors[k]
        return self.values[k]

    def peek(self, k):
        return lambda: self.get(k)


def slice_sample_batched(init_x, logprob_many, sigma=1.0, step_out=True, max_steps_out=1000,
                         compwise=False, lookahead=4, follow=(0, 0)):
    """slice_sample with speculative batches.  `logprob_many(list of x) -> _LazyValues`.
    follow = (proposals planned for the next coordinate's move, acceptance hypotheses): cross-move speculation between
    the coordinate moves of a component-wise sweep, for evaluators with the lazy `submit` form (see _slice_along_batched_impl)."""
    x = np.asarray(init_x, dtype=float)
    if not x.shape:
        x = np.array([float(x)])
    dims = x.shape[0]
    if compwise:
        order = list(range(dims))
        npr.shuffle(order)
        cur = x.copy()
        rng = _Uniforms()        # (after the shuffle: from here on every draw of the moves is a plain rand())
        try:
            for i, d in enumerate(order):
                e = np.zeros(dims)
                e[d] = 1.0
                fol = None
                if follow[0] > 0 and follow[1] > 0 and i + 1 < dims:
                    e2 = np.zeros(dims)
                    e2[order[i + 1]] = 1.0
                    fol = (e2, follow[0], follow[1])
                cur = _slice_along_batched(e, cur, logprob_many, sigma, step_out, max_steps_out, lookahead, dim=d, rng=rng,
                                           follow=fol)
        finally:
            rng.close()
        return cur
    direction = npr.randn(dims)
    direction = direction / np.sqrt(np.sum(direction ** 2))
    return _slice_along_batched(direction, x, logprob_many, sigma, step_out, max_steps_out, lookahead)
