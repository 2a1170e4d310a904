"""Random mixed factor graphs for parity tests: boolean + categorical variables,
sparse domains with truthiness, every factor function, arities 1-4, duplicate
variables inside a factor, duplicate factors, non-f32 feature values."""
import numpy as np

from sampler_amd.rawgraph import RawGraph

BOOL_FUNCS = [0, 1, 2, 3, 4, 7, 8, 9, 13]


def random_graph(seed, V=60, F=200, W=12, p_cat=0.4, with_domains=True, truthy=False,
                 max_arity=4, exact_fvals=False):
    rng = np.random.default_rng(seed)
    dtype = (rng.random(V) < p_cat).astype(np.uint16)
    card = np.where(dtype == 1, rng.integers(1, 6, V), 2).astype(np.uint64)
    role = (rng.random(V) < 0.45).astype(np.uint8)
    dom_vid, dom_off, dom_val, dom_tr = [], [0], [], []
    domain = {}
    for v in range(V):
        if dtype[v] == 1 and with_domains and rng.random() < 0.6:
            vals = rng.choice(50, size=int(card[v]), replace=False)
            domain[v] = vals
            dom_vid.append(v)
            dom_val.extend(vals.tolist())
            if truthy:
                t = rng.random(int(card[v])) * (rng.random(int(card[v])) < 0.7)
                t = t / max(t.sum(), 1.0)
            else:
                t = np.zeros(int(card[v]))
            dom_tr.extend(t.tolist())
            dom_off.append(len(dom_val))
        elif dtype[v] == 1:
            domain[v] = np.arange(int(card[v]))
        else:
            domain[v] = np.array([0, 1])
    init = np.zeros(V, np.uint64)
    for v in range(V):
        if role[v]:
            init[v] = rng.choice(domain[v])
    func, off, wid, fval, evid, eeq = [], [0], [], [], [], []
    cats = np.flatnonzero(dtype == 1)
    bools = np.flatnonzero(dtype == 0)
    for f in range(F):
        ar = int(rng.integers(1, max_arity + 1))
        if len(cats) and (rng.random() < 0.4 or not len(bools)):
            fn = 12
            vs = rng.choice(cats, size=ar, replace=True)
        else:
            fn = int(rng.choice(BOOL_FUNCS))
            vs = rng.choice(bools, size=ar, replace=True)
        func.append(fn)
        for v in vs:
            evid.append(int(v))
            eeq.append(int(rng.choice(domain[int(v)])))
        off.append(len(evid))
        wid.append(int(rng.integers(0, W)))
        fval.append(float(rng.choice([1.0, 1.0, -1.5, 0.25, 2.0, 3.0] if exact_fvals else
                                     [1.0, 1.0, -1.5, 0.1, 2.0, 0.3333333333333333])))
        if rng.random() < 0.05 and f + 1 < F:   # exact duplicate factor
            func.append(fn)
            evid.extend(evid[off[-2]:off[-1]]); eeq.extend(eeq[off[-2]:off[-1]])
            off.append(len(evid)); wid.append(wid[-1]); fval.append(fval[-1])
    w_init = rng.normal(0, 0.7, W)
    w_fixed = (rng.random(W) < 0.3).astype(np.uint8)
    return RawGraph(role, init, dtype, card, np.array(func, np.uint16), np.array(off, np.uint64),
                    np.array(wid, np.uint64), np.array(fval), np.array(evid, np.uint64),
                    np.array(eeq, np.uint64), w_init, w_fixed,
                    np.array(dom_vid, np.uint64), np.array(dom_off, np.uint64),
                    np.array(dom_val, np.uint64), np.array(dom_tr, np.float64))


def hub_graph(seed, V=400, hub_degree=5000, W=50):
    """Power-law-ish: variables 0 and 1 (one evidence, one query) each sit in `hub_degree`
    factors (unary ISTRUE and binary EQUAL/OR/IMPLY to random others); everybody else has a
    few unary factors.  The hubs exceed any LDS tile and take the block-cooperative path."""
    rng = np.random.default_rng(seed)
    func, off, wid, fval, evid, eeq = [], [0], [], [], [], []

    def add(fn, vs, eqs):
        func.append(fn); evid.extend(vs); eeq.extend(eqs); off.append(len(evid))
        wid.append(int(rng.integers(0, W))); fval.append(float(rng.choice([1.0, -1.0, 0.5, 2.0])))

    for hub in (0, 1):
        for _ in range(hub_degree):
            if rng.random() < 0.5:
                add(4, [hub], [int(rng.integers(0, 2))])
            else:
                o = int(rng.integers(2, V))
                pair = [hub, o] if rng.random() < 0.5 else [o, hub]
                add(int(rng.choice([3, 1, 0, 2])), pair, [int(rng.integers(0, 2)), int(rng.integers(0, 2))])
    for v in range(2, V):
        for _ in range(3):
            add(4, [v], [1])
    role = (rng.random(V) < 0.4).astype(np.uint8)
    role[0], role[1] = 1, 0
    init = (rng.random(V) < 0.6).astype(np.uint64) * role
    return RawGraph(role, init, np.zeros(V, np.uint16), np.full(V, 2, np.uint64),
                    np.array(func, np.uint16), np.array(off, np.uint64), np.array(wid, np.uint64),
                    np.array(fval), np.array(evid, np.uint64), np.array(eeq, np.uint64),
                    rng.normal(0, 0.02, W), np.zeros(W, np.uint8))
