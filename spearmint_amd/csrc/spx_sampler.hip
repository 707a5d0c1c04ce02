// spx_sampler.hip -- HOST code: the slice sampler of the GP hyper-parameters inside the library.
//
// What it replaces (S = spearmint/spearmint/): `sample_hypers` of the GP-EI choosers -- S/chooser/GPEIChooser.py:268-346
// (`_sample_noisy` / `_sample_noiseless` / `_sample_ls` and their logprob closures), S/chooser/GPEIOptChooser.py:621-706,
// S/chooser/GPEIperSecChooser.py:558-700 -- on S/util.py:34-93 (`slice_sample`), whose every log-probability is a covariance
// build + Cholesky + solve: spx_gp_logprob.  Round 5 kept the sampler's control flow in Python (spearmint_amd/util.py) and
// spent more wall time between two GPU calls than inside them at Spearmint's operating sizes (N = 256: 15 ms of interpreter
// for 24 ms of calls per next()).  This file is that control flow in C++:
//
//   * numpy's legacy random stream (RandomState on MT19937: rand = 53-bit doubles from two 32-bit draws, randn = polar
//     Box-Muller with the cached second value, shuffle = masked-rejection interval per swap) -- state in, state out;
//   * util.slice_sample's moves in the reference's order (random-direction move over [mean, amp2, noise]; shuffled
//     component-wise sweep over the length scales), the priors and a-priori rejections of the closures;
//   * the speculative batching of spearmint_amd/util.py (bracket, step-out ladders, shrink proposals of the most probable
//     brackets in ONE call) and, new in round 6, cross-move speculation: the call also carries what the NEXT coordinate's
//     move will ask for if this one accepts one of its first proposals, so that move often needs no call at all.
//
// Which points are accepted, and how many random numbers are consumed, does not depend on the batching: it is the
// reference's Markov chain (tests/test_sampler_native.py: the reference's golden trace, util.slice_sample, the Python
// batched sampler -- values and generator state).  Floating point: every expression is written as numpy evaluates it
// (elementwise direction * z + x0 with two roundings: contraction is off in this file).
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <chrono>
#include <deque>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "spx_internal.h"

#pragma STDC FP_CONTRACT OFF
#pragma clang fp contract(off)

namespace {

// ------------------------------------------------------------------------------------------------------------------
// numpy.random.RandomState (legacy) on MT19937: numpy/random/src/mt19937/mt19937.c, src/legacy/legacy-distributions.c,
// src/distributions/distributions.c (random_interval), _mt19937.pyx / mtrand.pyx (shuffle, untyped path)
// ------------------------------------------------------------------------------------------------------------------
struct Rng {
    uint32_t key[624];
    int pos;
    int has_gauss;
    double gauss;

    void load(const spx_rng_state* s) { memcpy(key, s->key, sizeof key); pos = s->pos; has_gauss = s->has_gauss; gauss = s->gauss; }
    void store(spx_rng_state* s) const { memcpy(s->key, key, sizeof key); s->pos = pos; s->has_gauss = has_gauss; s->gauss = gauss; }

    void gen()
    {
        const int N = 624, M = 397;
        const uint32_t A = 0x9908b0dfu, UP = 0x80000000u, LO = 0x7fffffffu;
        uint32_t y;
        int i;
        for (i = 0; i < N - M; i++) {
            y = (key[i] & UP) | (key[i + 1] & LO);
            key[i] = key[i + M] ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1)) & A);
        }
        for (; i < N - 1; i++) {
            y = (key[i] & UP) | (key[i + 1] & LO);
            key[i] = key[i + (M - N)] ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1)) & A);
        }
        y = (key[N - 1] & UP) | (key[0] & LO);
        key[N - 1] = key[M - 1] ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1)) & A);
        pos = 0;
    }
    uint32_t u32()
    {
        if (pos == 624) gen();
        uint32_t y = key[pos++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    double rand()           // npr.rand(): mt19937_next_double
    {
        const int32_t a = (int32_t)(u32() >> 5), b = (int32_t)(u32() >> 6);
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
    double randn()          // npr.randn(): legacy_gauss
    {
        if (has_gauss) {
            const double t = gauss;
            has_gauss = 0;
            gauss = 0.0;
            return t;
        }
        double f, x1, x2, r2;
        do {
            x1 = 2.0 * rand() - 1.0;
            x2 = 2.0 * rand() - 1.0;
            r2 = x1 * x1 + x2 * x2;
        } while (r2 >= 1.0 || r2 == 0.0);
        f = sqrt(-2.0 * log(r2) / r2);
        gauss = f * x1;
        has_gauss = 1;
        return f * x2;
    }
    uint32_t interval(uint32_t max)   // random_interval for max <= 0xffffffff
    {
        if (max == 0) return 0;
        uint32_t mask = max, v;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        while ((v = (u32() & mask)) > max) {}
        return v;
    }
    void shuffle(std::vector<int>& x)  // npr.shuffle(list): for i in reversed(range(1, n)): j = random_interval(i); swap
    {
        for (int i = (int)x.size() - 1; i >= 1; --i) {
            const int j = (int)interval((uint32_t)i);
            const int t = x[i]; x[i] = x[j]; x[j] = t;
        }
    }
};

// The stream of rand() with look-ahead (spearmint_amd/util.py: _Uniforms): `real` is advanced only by numbers the sampler
// consumes, `ahead` is a copy that runs in front of it for peek().
struct Uniforms {
    Rng* real;
    Rng ahead;
    std::deque<double> buf;
    explicit Uniforms(Rng* r) : real(r), ahead(*r) {}
    double peek_at(size_t i)
    {
        while (buf.size() <= i) buf.push_back(ahead.rand());
        return buf[i];
    }
    double take()
    {
        const double v = real->rand();
        if (!buf.empty()) buf.pop_front();
        else (void)ahead.rand();
        return v;
    }
};

struct SliceError { int code; };

typedef std::vector<double> Vec;

// what the reference's closures capture: which coordinates move, the a-priori rejections, the priors
struct Model {
    int kind;                 // 0: [mean, amp2, noise] (joint move)   1: length scales (component-wise sweep)
    int D;
    const spx_sampler_cfg* cfg;
    double mean, amp2, noise; // kind 1: fixed
    Vec ls;                   // kind 0: fixed
    bool check_mean;

    bool admissible(const Vec& x) const
    {
        if (kind == 0) {      // GPEIChooser.py:289-295 / :330-337
            const double m = x[0], a = x[1], n = cfg->noiseless ? 1e-3 : x[2];
            if (check_mean && (m > cfg->vals_max || m < cfg->vals_min)) return false;
            if (a < 0 || n < 0) return false;
            return true;
        }
        for (int i = 0; i < D; ++i)       // GPEIChooser.py:278-279
            if (x[i] < 0 || x[i] > cfg->max_ls) return false;
        return true;
    }
    void row(const Vec& x, double* r) const   // [mean, noise, amp2, ls...]
    {
        if (kind == 0) {
            r[0] = x[0]; r[1] = cfg->noiseless ? 1e-3 : x[2]; r[2] = x[1];
            for (int i = 0; i < D; ++i) r[3 + i] = ls[i];
        } else {
            r[0] = mean; r[1] = noise; r[2] = amp2;
            for (int i = 0; i < D; ++i) r[3 + i] = x[i];
        }
    }
    double finish(const Vec& x, double lp) const
    {
        if (kind == 1) return lp;
        const double a2 = x[1], n = cfg->noiseless ? 1e-3 : x[2];
        if (!cfg->noiseless)                                         // horseshoe (GPEIChooser.py:309)
            lp += log(log(1 + pow(cfg->noise_scale / n, 2.0)));
        const double a = cfg->amp2_prior_on_sqrt ? sqrt(a2) : a2;    // log-normal (:312 / GPEIOptChooser.py:668)
        lp -= 0.5 * pow(log(a) / cfg->amp2_scale, 2.0);
        return lp;
    }
};

struct Evaluator;

// results of a speculative batch (util._LazyValues): nothing is evaluated until a missing entry is asked for
struct Batch {
    Evaluator* ev;
    const Model* m;
    std::vector<Vec> xs, extras;
    std::vector<double> values;
    std::vector<char> bad, missing;
    int n_missing = 0;
    double get(int k);
};

struct Evaluator {
    spx_logprob_fn fn;
    void* ctx;
    int D, max_rows;
    std::unordered_map<std::string, std::pair<double, bool> > memo[2];   // this batch's rows and the batch before
    int64_t calls = 0, rows = 0;
    int64_t by_rows[34] = {0};   // calls by number of hyper rows (33: more than 32)
    int64_t ns_in_calls = 0;     // wall time inside the log-likelihood callback
    int rc = 0;

    static std::string key_of(const double* r, int n) { return std::string((const char*)r, (size_t)n * 8); }
    const std::pair<double, bool>* find(const std::string& k) const
    {
        auto it = memo[0].find(k);
        if (it != memo[0].end()) return &it->second;
        it = memo[1].find(k);
        if (it != memo[1].end()) return &it->second;
        return nullptr;
    }
    std::unique_ptr<Batch> submit(const Model* m, std::vector<Vec>&& xs, std::vector<Vec>&& extras)
    {
        std::unique_ptr<Batch> b(new Batch);
        b->ev = this; b->m = m;
        const size_t n = xs.size();
        b->values.assign(n, -INFINITY); b->bad.assign(n, 0); b->missing.assign(n, 0);
        Vec r((size_t)3 + D);
        for (size_t k = 0; k < n; ++k) {
            if (!m->admissible(xs[k])) continue;           // -inf a priori, as the reference's closures return it
            m->row(xs[k], r.data());
            const std::pair<double, bool>* got = find(key_of(r.data(), 3 + D));
            if (got) {
                if (got->second) b->bad[k] = 1;
                else b->values[k] = m->finish(xs[k], got->first);
            } else {
                b->missing[k] = 1;
                b->n_missing += 1;
            }
        }
        b->xs = std::move(xs);
        if (b->n_missing) b->extras = std::move(extras);
        return b;
    }
    void fill(Batch* b)
    {
        const int L = 3 + D;
        std::vector<double> R;
        std::unordered_map<std::string, int> index;
        std::vector<int> slot(b->xs.size(), -1);
        Vec r((size_t)L);
        for (size_t k = 0; k < b->xs.size(); ++k) {
            if (!b->missing[k]) continue;
            b->m->row(b->xs[k], r.data());
            const std::string key = key_of(r.data(), L);
            auto it = index.find(key);
            if (it == index.end()) {
                it = index.emplace(key, (int)(R.size() / L)).first;
                R.insert(R.end(), r.begin(), r.end());
            }
            slot[k] = it->second;
        }
        int room = max_rows - (int)(R.size() / L);
        for (size_t e = 0; e < b->extras.size() && room > 0; ++e) {
            if (!b->m->admissible(b->extras[e])) continue;
            b->m->row(b->extras[e], r.data());
            const std::string key = key_of(r.data(), L);
            if (index.count(key) || find(key)) continue;
            index.emplace(key, (int)(R.size() / L));
            R.insert(R.end(), r.begin(), r.end());
            room -= 1;
        }
        const int n = (int)(R.size() / L);
        std::vector<double> lp((size_t)n);
        const auto t0 = std::chrono::steady_clock::now();
        rc = fn(ctx, R.data(), n, lp.data());
        ns_in_calls += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        if (rc) throw SliceError{rc};
        calls += 1; rows += n;
        by_rows[n > 32 ? 33 : n] += 1;
        memo[1].swap(memo[0]);
        memo[0].clear();
        for (auto& kv : index) memo[0][kv.first] = std::make_pair(lp[kv.second], (bool)(isinf(lp[kv.second]) && lp[kv.second] < 0));
        for (size_t k = 0; k < b->xs.size(); ++k) {
            if (!b->missing[k]) continue;
            const double v = lp[slot[k]];
            if (isinf(v) && v < 0) b->bad[k] = 1;    // spla.cholesky would raise here -- only if the sampler really gets to it
            else b->values[k] = b->m->finish(b->xs[k], v);
            b->missing[k] = 0;
        }
        b->n_missing = 0;
        b->extras.clear();
    }
};

double Batch::get(int k)
{
    if (n_missing && missing[k]) ev->fill(this);
    if (bad[k]) throw SliceError{SPX_ERR_NOT_PD};
    return values[k];
}

struct Ref { Batch* b; int k; };
struct Rung { double z; Ref ref; };

std::vector<double> ladder(double start, double step, int first, int count)
{   // positions by repeated addition, exactly like the reference's `lower -= sigma` / `upper += sigma` (util.py:47-52)
    std::vector<double> out;
    double p = start;
    for (int i = 0; i < first; ++i) p = p + step;
    for (int i = 0; i < count; ++i) { out.push_back(p); p = p + step; }
    return out;
}

std::vector<double> propose(double l, double h, Uniforms& u, size_t off, int count)
{   // shrink proposals under the assumption that each one is rejected (bracket update by sign only; util.py:56-69)
    std::vector<double> z;
    for (int i = 0; i < count; ++i) {
        const double v = (h - l) * u.peek_at(off + (size_t)i) + l;
        z.push_back(v);
        if (v < 0) l = v;
        else if (v > 0) h = v;
        else break;
    }
    return z;
}

struct Spec { double lo, hi; int at, nu, cnt; };
struct Plan {
    std::vector<double> zs;
    double lo, hi, lo_wide, hi_wide;
    bool has_lo_wide, has_hi_wide;
    std::vector<Spec> spec;
};

struct Mover {
    Evaluator* ev;
    const Model* m;
    double* hist;             // [lo: stay, wide, other][hi: ...] of this kind of move
    double sigma = 1.0;
    int max_steps_out = 1000;
    int lookahead;
    int64_t moves = 0, free_moves = 0;

    static Vec at(const Vec& dir, double z, const Vec& x0)
    {
        Vec p(x0.size());
        for (size_t i = 0; i < x0.size(); ++i) p[i] = dir[i] * z + x0[i];
        return p;
    }
    // one end of the bracket: stays where it was drawn, or steps out to the first point outside the priors' support
    void options(const Vec& dir, const Vec& x0, double first, const std::vector<double>& lad, int end,
                 double* pos, double* prob, int* n, double* wide, bool* has_wide) const
    {
        *has_wide = false;
        pos[0] = first; prob[0] = 1.0; *n = 1;
        if (!m->admissible(at(dir, first, x0))) return;        // stops here a priori
        bool found = false;
        double w = 0.0;
        for (double z : lad)
            if (!m->admissible(at(dir, z, x0))) { w = z; found = true; break; }
        if (!found) return;
        double* hh = hist + 3 * end;
        if (hh[0] + hh[1] + hh[2] == 0.0) { hh[0] = 1.0; hh[1] = 1.0; hh[2] = 0.0; }
        const double tot = hh[0] + hh[1] + hh[2];
        pos[0] = first; prob[0] = hh[0] / tot;
        pos[1] = w; prob[1] = hh[1] / tot;
        *n = 2; *wide = w; *has_wide = true;
    }
    // the first batch of the move along `dir` through x0 whose upper edge is drawn with u_hi; its shrink proposals use the
    // uniforms from stream offset `off` on.  Pure: draws nothing, evaluates nothing -- also plans a move not yet started.
    Plan plan(const Vec& dir, const Vec& x0, double u_hi, Uniforms& u, size_t off, int la) const
    {
        Plan pl;
        pl.hi = sigma * u_hi;
        pl.lo = pl.hi - sigma;
        pl.zs = {0.0, pl.lo, pl.hi};
        std::vector<double> lo_lad = ladder(pl.lo, -sigma, 1, la), hi_lad = ladder(pl.hi, sigma, 1, la);
        pl.zs.insert(pl.zs.end(), lo_lad.begin(), lo_lad.end());
        pl.zs.insert(pl.zs.end(), hi_lad.begin(), hi_lad.end());
        double lp[2], lq[2], hp[2], hq[2];
        int nl, nh;
        options(dir, x0, pl.lo, lo_lad, 0, lp, lq, &nl, &pl.lo_wide, &pl.has_lo_wide);
        options(dir, x0, pl.hi, hi_lad, 1, hp, hq, &nh, &pl.hi_wide, &pl.has_hi_wide);
        // brackets by probability; sorted(..., key=-p) is stable: ties keep the order (lo option, hi option)
        struct Combo { double p, lo, hi; };
        std::vector<Combo> combos;
        for (int i = 0; i < nl; ++i)
            for (int j = 0; j < nh; ++j) combos.push_back({lq[i] * hq[j], lp[i], hp[j]});
        for (size_t i = 1; i < combos.size(); ++i)          // stable insertion sort, descending
            for (size_t j = i; j > 0 && combos[j].p > combos[j - 1].p; --j) std::swap(combos[j], combos[j - 1]);
        int counts[2] = {la, 0};
        int nscen = 1;
        if (combos.size() > 1 && combos[1].p >= 0.15 && la >= 2) {
            nscen = 2;
            if (combos[0].p > 2.0 * combos[1].p) {
                const int c1 = la / 3 > 1 ? la / 3 : 1;
                counts[0] = la - c1; counts[1] = c1;
            } else {
                counts[0] = la - la / 2; counts[1] = la / 2;
            }
        }
        for (int s = 0; s < nscen; ++s) {
            std::vector<double> zl = propose(combos[s].lo, combos[s].hi, u, off, counts[s]);
            pl.spec.push_back({combos[s].lo, combos[s].hi, (int)pl.zs.size(), counts[s], (int)zl.size()});
            pl.zs.insert(pl.zs.end(), zl.begin(), zl.end());
        }
        return pl;
    }

    // one slice move (util.py:35-76) along `dir` through x0; `fdir` != nullptr: the direction of the NEXT move of the sweep
    Vec move(const Vec& dir, const Vec& x0, Uniforms& u, const Vec* fdir, int fprops, int fhyps)
    {
        std::vector<std::unique_ptr<Batch> > keep;
        const int64_t calls_before = ev->calls;
        auto many = [&](const std::vector<double>& zs, std::vector<Vec>&& extras) -> Batch* {
            std::vector<Vec> pts;
            pts.reserve(zs.size());
            for (double z : zs) pts.push_back(at(dir, z, x0));
            keep.push_back(ev->submit(m, std::move(pts), std::move(extras)));
            return keep.back().get();
        };
        // Cross-move speculation: if proposal j of `zl` is accepted this move has consumed j + 1 of the uniforms ahead, and
        // the next move starts at that point with the numbers after them -- its edges, ladder and first proposals are
        // already determined.  They ride in the same call; the next move finds them in the memo or asks for them itself.
        auto follow = [&](const std::vector<double>& zl) -> std::vector<Vec> {
            std::vector<Vec> out;
            if (!fdir || fprops <= 0) return out;
            for (int j = 0; j < fhyps && j < (int)zl.size(); ++j) {
                Vec xn(x0.size());                                 // what this move returns if proposal j is accepted
                for (size_t i = 0; i < xn.size(); ++i) xn[i] = zl[(size_t)j] * dir[i] + x0[i];
                if (!m->admissible(xn)) continue;
                const size_t off = (size_t)j + 1;
                Plan p2 = plan(*fdir, xn, u.peek_at(off), u, off + 2, fprops);
                for (size_t i = 1; i < p2.zs.size(); ++i) out.push_back(at(*fdir, p2.zs[i], xn));
            }
            return out;
        };

        const double u_hi = u.take();
        const double u_level = u.take();
        Plan pl = plan(dir, x0, u_hi, u, 0, lookahead);
        std::vector<double> first_props(pl.zs.begin() + pl.spec[0].at, pl.zs.begin() + pl.spec[0].at + pl.spec[0].cnt);
        Batch* vals = many(pl.zs, follow(first_props));
        const double level = log(u_level) + vals->get(0);
        double lo = pl.lo, hi = pl.hi;
        {
            std::vector<Rung> lo_c, hi_c;
            (void)vals->get(1);
            (void)vals->get(2);          // f(lo), f(hi): the reference always evaluates both (errors surface here, in its order)
            lo_c.push_back({pl.lo, {vals, 1}});
            hi_c.push_back({pl.hi, {vals, 2}});
            for (int k = 0; k < lookahead; ++k) {
                lo_c.push_back({pl.zs[(size_t)(3 + k)], {vals, 3 + k}});
                hi_c.push_back({pl.zs[(size_t)(3 + lookahead + k)], {vals, 3 + lookahead + k}});
            }
            auto walk = [&](double step, std::vector<Rung>& c) -> double {
                size_t n = 0;
                while (true) {
                    if (n >= c.size()) {   // beyond the speculation window: fetch the next window
                        std::vector<double> pos = ladder(c[n - 1].z, step, 1, lookahead);
                        Batch* more = many(pos, std::vector<Vec>());
                        for (int j = 0; j < lookahead; ++j) c.push_back({pos[(size_t)j], {more, j}});
                    }
                    const double v = c[n].ref.b->get(c[n].ref.k);
                    if (!(v > level && (int)n < max_steps_out)) return c[n].z;
                    n += 1;
                }
            };
            // (the reference evaluates f(lo) before f(hi): a not-PD error at lo surfaces first)
            lo = walk(-sigma, lo_c);
            hi = walk(sigma, hi_c);
        }
        if (pl.has_lo_wide) hist[lo == pl.lo ? 0 : (lo == pl.lo_wide ? 1 : 2)] += 1.0;
        if (pl.has_hi_wide) hist[3 + (hi == pl.hi ? 0 : (hi == pl.hi_wide ? 1 : 2))] += 1.0;
        int hit = -1;
        for (size_t s = 0; s < pl.spec.size(); ++s)
            if (lo == pl.spec[s].lo && hi == pl.spec[s].hi) hit = (int)s;
        while (true) {
            std::vector<double> zl;
            Batch* batch;
            int base;
            if (hit >= 0) {               // the speculated proposals for this bracket are the real ones
                zl = propose(lo, hi, u, 0, pl.spec[(size_t)hit].nu);
                batch = vals;
                base = pl.spec[(size_t)hit].at;
                hit = -1;
            } else {
                zl = propose(lo, hi, u, 0, lookahead);
                batch = many(zl, follow(zl));
                base = 0;
            }
            for (size_t k = 0; k < zl.size(); ++k) {
                (void)u.take();           // the reference draws one number per proposal it reaches (util.py:57)
                const double z = zl[k];
                const double lp = batch->get(base + (int)k);
                if (lp != lp) throw SliceError{SPX_ERR_SLICE_NAN};
                if (lp > level) {
                    moves += 1;
                    if (ev->calls == calls_before) free_moves += 1;
                    Vec out(x0.size());
                    for (size_t i = 0; i < x0.size(); ++i) out[i] = z * dir[i] + x0[i];
                    return out;
                }
                if (z < 0) lo = z;
                else if (z > 0) hi = z;
                else throw SliceError{SPX_ERR_SLICE_ZERO};
            }
        }
    }
};

int run_sampler(spx_logprob_fn fn, void* ctx, const spx_sampler_cfg* cfg, spx_rng_state* rng_io, double* hyper_io,
                double* rows_out, double* hist_io, int64_t* stats_out)
{
    if (!fn || !cfg || !rng_io || !hyper_io || !hist_io)
        return fail(SPX_ERR_ARG, "spx_sample_hypers: null argument");
    const int D = cfg->D;
    if (D < 1 || cfg->n_iter < 0 || cfg->lookahead < 1 || cfg->max_rows < 1 || rng_io->pos < 0 || rng_io->pos > 624)
        return fail(SPX_ERR_ARG, "spx_sample_hypers: bad configuration (D=%d, n_iter=%d, lookahead=%d, max_rows=%d, pos=%d)",
                    D, cfg->n_iter, cfg->lookahead, cfg->max_rows, rng_io->pos);
    Rng rng;
    rng.load(rng_io);
    double mean = hyper_io[0], noise = hyper_io[1], amp2 = hyper_io[2];
    Vec ls(hyper_io + 3, hyper_io + 3 + D);
    int64_t calls = 0, rows = 0, moves = 0, free_moves = 0, done = 0;
    int64_t by_rows[34] = {0};
    int64_t ns_calls = 0;
    const auto t_begin = std::chrono::steady_clock::now();
    int rc = SPX_OK;
    try {
        for (int it = 0; it < cfg->n_iter; ++it) {
            // ---- [mean, amp2, noise]: one move along a random direction (GPEIChooser.py:268-271, util.py:88-93)
            if (cfg->noiseless) noise = 1e-3;
            {
                Model m;
                m.kind = 0; m.D = D; m.cfg = cfg; m.ls = ls;
                m.check_mean = cfg->check_mean != 0;
                Evaluator ev;
                ev.fn = fn; ev.ctx = ctx; ev.D = D; ev.max_rows = cfg->max_rows;
                Mover mv;
                mv.ev = &ev; mv.m = &m; mv.hist = hist_io; mv.lookahead = cfg->lookahead;
                Vec dir(3);
                for (int i = 0; i < 3; ++i) dir[(size_t)i] = rng.randn();
                double ss = 0.0;
                for (int i = 0; i < 3; ++i) ss += dir[(size_t)i] * dir[(size_t)i];
                const double nrm = sqrt(ss);
                for (int i = 0; i < 3; ++i) dir[(size_t)i] = dir[(size_t)i] / nrm;
                Uniforms u(&rng);
                Vec x0 = {mean, amp2, noise};
                struct Acc { int64_t &c, &r, &mv_, &f; int64_t* hb; int64_t& ns; Evaluator& e; Mover& m_; ~Acc() { c += e.calls; r += e.rows; mv_ += m_.moves; f += m_.free_moves; ns += e.ns_in_calls; for (int q = 0; q < 34; ++q) hb[q] += e.by_rows[q]; } }
                    acc{calls, rows, moves, free_moves, by_rows, ns_calls, ev, mv};
                Vec nx = mv.move(dir, x0, u, nullptr, 0, 0);
                mean = nx[0]; amp2 = nx[1]; noise = cfg->noiseless ? 1e-3 : nx[2];
            }
            // ---- length scales: one move per coordinate in a shuffled order (GPEIChooser.py:274, util.py:80-87)
            {
                Model m;
                m.kind = 1; m.D = D; m.cfg = cfg; m.mean = mean; m.amp2 = amp2; m.noise = noise;
                m.check_mean = false;
                Evaluator ev;
                ev.fn = fn; ev.ctx = ctx; ev.D = D; ev.max_rows = cfg->max_rows;
                Mover mv;
                mv.ev = &ev; mv.m = &m; mv.hist = hist_io + 6; mv.lookahead = cfg->lookahead;
                std::vector<int> order((size_t)D);
                for (int i = 0; i < D; ++i) order[(size_t)i] = i;
                rng.shuffle(order);
                Uniforms u(&rng);
                Vec cur = ls;
                struct Acc { int64_t &c, &r, &mv_, &f; int64_t* hb; int64_t& ns; Evaluator& e; Mover& m_; ~Acc() { c += e.calls; r += e.rows; mv_ += m_.moves; f += m_.free_moves; ns += e.ns_in_calls; for (int q = 0; q < 34; ++q) hb[q] += e.by_rows[q]; } }
                    acc{calls, rows, moves, free_moves, by_rows, ns_calls, ev, mv};
                for (int i = 0; i < D; ++i) {
                    Vec e((size_t)D, 0.0), e2;
                    e[(size_t)order[(size_t)i]] = 1.0;
                    const bool fol = cfg->follow_props > 0 && cfg->follow_hyps > 0 && i + 1 < D;
                    if (fol) { e2.assign((size_t)D, 0.0); e2[(size_t)order[(size_t)i + 1]] = 1.0; }
                    cur = mv.move(e, cur, u, fol ? &e2 : nullptr, cfg->follow_props, cfg->follow_hyps);
                }
                ls = cur;
            }
            if (rows_out) {
                double* r = rows_out + (size_t)it * (3 + D);
                r[0] = mean; r[1] = noise; r[2] = amp2;
                for (int i = 0; i < D; ++i) r[3 + i] = ls[(size_t)i];
            }
            done += 1;
        }
    } catch (const SliceError& e) {
        rc = e.code;
    } catch (const std::exception& e) {
        rc = fail(SPX_ERR_ARG, "spx_sample_hypers: %s", e.what());
    }
    rng.store(rng_io);
    hyper_io[0] = mean; hyper_io[1] = noise; hyper_io[2] = amp2;
    for (int i = 0; i < D; ++i) hyper_io[3 + i] = ls[(size_t)i];
    if (stats_out) {
        stats_out[0] = calls; stats_out[1] = rows; stats_out[2] = moves; stats_out[3] = free_moves; stats_out[4] = done;
        for (int q = 0; q < 34; ++q) stats_out[5 + q] = by_rows[q];
        stats_out[39] = ns_calls;
        stats_out[40] = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_begin).count();
    }
    if (rc == SPX_ERR_NOT_PD) return fail(rc, "slice sampler: covariance not positive definite at a point the sampler evaluated");
    if (rc == SPX_ERR_SLICE_NAN) return fail(rc, "Slice sampler got a NaN");
    if (rc == SPX_ERR_SLICE_ZERO) return fail(rc, "Slice sampler shrank to zero!");
    return rc;
}

int gpu_logprob(void* ctx, const double* rows, int32_t n_rows, double* lp_out)
{
    spx_handle* h = (spx_handle*)ctx;
    int rc = spx_set_hypers(h, rows, n_rows);
    if (rc) return rc;
    return spx_gp_logprob(h, lp_out);
}

}  // namespace

extern "C" {

int spx_sample_hypers(spx_handle* h, const spx_sampler_cfg* cfg, spx_rng_state* rng, double* hyper_io, double* rows_out,
                      double* hist_io, int64_t* stats_out)
{
    if (!h || !cfg) return fail(SPX_ERR_ARG, "spx_sample_hypers: null handle / configuration");
    int64_t d = 0;      // (the rows the sampler hands to spx_set_hypers are 3 + cfg->D long: they must be the handle's)
    int rc = spx_get_stat(h, "obs_dims", &d);
    if (rc) return rc;
    if (d == 0) return fail(SPX_ERR_ARG, "spx_sample_hypers: call spx_set_observations first");
    if (d != cfg->D) return fail(SPX_ERR_ARG, "spx_sample_hypers: cfg->D = %d but the observations have D = %lld", cfg->D, (long long)d);
    return run_sampler(gpu_logprob, h, cfg, rng, hyper_io, rows_out, hist_io, stats_out);
}

int spx_sample_hypers_with(spx_logprob_fn fn, void* ctx, const spx_sampler_cfg* cfg, spx_rng_state* rng, double* hyper_io,
                           double* rows_out, double* hist_io, int64_t* stats_out)
{
    return run_sampler(fn, ctx, cfg, rng, hyper_io, rows_out, hist_io, stats_out);
}

int spx_rng_draw(spx_rng_state* rng, int32_t n_rand, double* rand_out, int32_t n_randn, double* randn_out,
                 int32_t n_shuffle, int32_t* shuffle_out)
{
    if (!rng || rng->pos < 0 || rng->pos > 624 || (n_rand > 0 && !rand_out) || (n_randn > 0 && !randn_out)
        || (n_shuffle > 0 && !shuffle_out))
        return fail(SPX_ERR_ARG, "spx_rng_draw: bad argument");
    Rng r;
    r.load(rng);
    for (int i = 0; i < n_rand; ++i) rand_out[i] = r.rand();
    for (int i = 0; i < n_randn; ++i) randn_out[i] = r.randn();
    if (n_shuffle > 0) {
        std::vector<int> x((size_t)n_shuffle);
        for (int i = 0; i < n_shuffle; ++i) x[(size_t)i] = i;
        r.shuffle(x);
        for (int i = 0; i < n_shuffle; ++i) shuffle_out[i] = x[(size_t)i];
    }
    r.store(rng);
    return SPX_OK;
}

}  // extern "C"
