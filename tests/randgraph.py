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


def degree_graph(seed, n_low=3000, n_high=120, max_degree=20_000, W=300, p_cat=0.25, card=4):
    """Skewed degrees (the power-law shape of real DeepDive graphs): n_low variables with 1-8
    factors and n_high variables whose degrees are log-uniform in [16, max_degree] -- tens,
    hundreds, thousands of factors: the lane-per-variable tiles, the wave-per-variable bin
    (TILE_WIDE) and the workgroup-per-variable kernel (TILE_GIANT) all get work.  Factors of a
    high-degree variable: unary (any boolean function / AND_CATEGORICAL for the categorical
    ones) or pairwise to a random low-degree boolean variable.  Feature values f32-exact."""
    rng = np.random.default_rng(seed)
    V = n_low + n_high
    dtype = np.zeros(V, np.uint16)
    dtype[rng.random(V) < p_cat] = 1
    cardv = np.where(dtype == 1, card, 2).astype(np.uint64)
    role = (rng.random(V) < 0.45).astype(np.uint8)
    init = np.array([int(rng.integers(0, cardv[v])) if role[v] else 0 for v in range(V)], np.uint64)
    deg = np.concatenate([rng.integers(1, 9, n_low),
                          np.exp(rng.uniform(np.log(16), np.log(max_degree), n_high)).astype(np.int64)])
    rng.shuffle(deg)
    low_bools = np.flatnonzero((dtype == 0) & (deg <= 8))
    func, off, wid, fval, evid, eeq = [], [0], [], [], [], []
    fvals = [1.0, -1.0, 0.5, 2.0, 0.25]
    for v in range(V):
        for _ in range(int(deg[v])):
            if dtype[v] == 1:
                func.append(12); evid.append(v); eeq.append(int(rng.integers(0, card)))
            elif rng.random() < 0.6 or not len(low_bools):
                func.append(int(rng.choice(BOOL_FUNCS))); evid.append(v); eeq.append(int(rng.integers(0, 2)))
            else:
                o = int(rng.choice(low_bools))
                pair = [v, o] if rng.random() < 0.5 else [o, v]
                func.append(int(rng.choice([3, 1, 0, 2, 13])))
                evid.extend(pair); eeq.extend([int(rng.integers(0, 2)), int(rng.integers(0, 2))])
            off.append(len(evid)); wid.append(int(rng.integers(0, W))); fval.append(float(rng.choice(fvals)))
    return RawGraph(role, init, dtype, cardv, np.array(func, np.uint16), np.array(off, np.uint64),
                    np.array(wid, np.uint64), np.array(fval), np.array(evid, np.uint64), np.array(eeq, np.uint64),
                    rng.normal(0, 0.05, W), (rng.random(W) < 0.1).astype(np.uint8))


def degree_graph_fast(seed, n_low=200_000, n_high=3000, max_degree=100_000, W=5000):
    """degree_graph's shape at sizes a Python loop per factor cannot build: boolean variables
    only, degrees 1-8 for n_low of them and log-uniform in [16, max_degree] for n_high, 60 %
    unary ISTRUE / 40 % pairwise (EQUAL, OR, IMPLY, AND) to a random low-degree variable;
    numpy-vectorised."""
    rng = np.random.default_rng(seed)
    V = n_low + n_high
    deg = np.concatenate([rng.integers(1, 9, n_low),
                          np.exp(rng.uniform(np.log(16), np.log(max_degree), n_high)).astype(np.int64)])
    rng.shuffle(deg)
    low = np.flatnonzero(deg <= 8)
    owner = np.repeat(np.arange(V, dtype=np.uint64), deg)
    F = len(owner)
    is_bin = rng.random(F) < 0.4
    other = low[rng.integers(0, len(low), F)].astype(np.uint64)
    arity = 1 + is_bin.astype(np.uint64)
    off = np.zeros(F + 1, np.uint64)
    np.cumsum(arity, out=off[1:])
    E = int(off[-1])
    evid = np.empty(E, np.uint64)
    first = off[:-1].astype(np.int64)
    swap = is_bin & (rng.random(F) < 0.5)
    evid[first] = np.where(swap, other, owner)
    evid[first[is_bin] + 1] = np.where(swap, owner, other)[is_bin]
    func = np.where(is_bin, rng.choice(np.array([3, 1, 0, 2], np.uint16), F), np.uint16(4)).astype(np.uint16)
    role = (rng.random(V) < 0.45).astype(np.uint8)
    init = ((rng.random(V) < 0.6) & (role == 1)).astype(np.uint64)
    return RawGraph(role, init, np.zeros(V, np.uint16), np.full(V, 2, np.uint64), func, off,
                    rng.integers(0, W, F).astype(np.uint64), rng.choice(np.array([1.0, -1.0, 0.5, 2.0]), F),
                    evid, rng.integers(0, 2, E).astype(np.uint64), rng.normal(0, 0.02, W), np.zeros(W, np.uint8))
