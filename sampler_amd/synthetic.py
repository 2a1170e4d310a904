"""Synthetic factor graphs of BASELINE.json's configs.  All seeds
fixed; generators are numpy-vectorised so the 10M-variable graph builds in seconds.

`var_offset`/`total_variables` generate one contiguous variable block of a larger
graph (the variable-block shard one GPU owns, config 5): ids are local to the block,
weights are global.
"""
import numpy as np

from .rawgraph import (RawGraph, FUNC_ISTRUE, FUNC_EQUAL, FUNC_AND_CATEGORICAL,
                       DTYPE_BOOLEAN, DTYPE_CATEGORICAL)


def _rng(seed, shard=0):
    return np.random.Generator(np.random.PCG64([seed, shard]))


def _unary_block(V, k, W, rng):
    """k unary factors per variable, emitted variable-major."""
    F = V * k
    edge_vid = np.repeat(np.arange(V, dtype=np.uint64), k)
    wid = rng.integers(0, W, size=F, dtype=np.uint64)
    return edge_vid, wid


def cfg2(V=1_000_000, k=10, n_weights=None, seed=1234, shard=0):
    """Config 2: V boolean query variables x k unary ISTRUE factors, W fixed weights
    ~ N(0, 0.5^2); closed form P(x=1) = sigmoid(2 * sum_j w_j)."""
    W = n_weights or max(1, V // 10)
    rng = _rng(seed, shard)
    w = _rng(seed, 10_000).normal(0.0, 0.5, W)
    edge_vid, wid = _unary_block(V, k, W, rng)
    F = V * k
    return RawGraph(
        var_role=np.zeros(V, np.uint8), var_init_value=np.zeros(V, np.uint64),
        var_dtype=np.full(V, DTYPE_BOOLEAN, np.uint16), var_cardinality=np.full(V, 2, np.uint64),
        fac_func=np.full(F, FUNC_ISTRUE, np.uint16),
        fac_edge_offset=np.arange(F + 1, dtype=np.uint64),
        fac_weight_id=wid, fac_feature_value=np.ones(F),
        edge_vid=edge_vid, edge_equal_to=np.ones(F, np.uint64),
        w_initial_value=w, w_is_fixed=np.ones(W, np.uint8))


def cfg2_closed_form(g: RawGraph):
    """Exact marginals of a unary ISTRUE/AND graph: sigmoid(2 * sum w*f)."""
    V = g.num_variables
    contrib = g.w_initial_value[g.fac_weight_id.astype(np.int64)] * g.fac_feature_value
    s = np.zeros(V)
    np.add.at(s, g.edge_vid.astype(np.int64), contrib)
    return 1.0 / (1.0 + np.exp(-2.0 * s))


def cfg3(V=10_000_000, k=10, n_weights=None, seed=1234, shard=0):
    """Config 3: as cfg2 but learnable weights (init 0), 50 % evidence variables with
    value ~ Bernoulli(0.7)."""
    W = n_weights or max(1, V // 10)
    rng = _rng(seed, shard)
    edge_vid, wid = _unary_block(V, k, W, rng)
    is_evid = rng.random(V) < 0.5
    val = (rng.random(V) < 0.7) & is_evid
    F = V * k
    return RawGraph(
        var_role=is_evid.astype(np.uint8), var_init_value=val.astype(np.uint64),
        var_dtype=np.full(V, DTYPE_BOOLEAN, np.uint16), var_cardinality=np.full(V, 2, np.uint64),
        fac_func=np.full(F, FUNC_ISTRUE, np.uint16),
        fac_edge_offset=np.arange(F + 1, dtype=np.uint64),
        fac_weight_id=wid, fac_feature_value=np.ones(F),
        edge_vid=edge_vid, edge_equal_to=np.ones(F, np.uint64),
        w_initial_value=np.zeros(W), w_is_fixed=np.zeros(W, np.uint8))


def cfg3b(V=10_000_000, n_weights=None, seed=1234, offsets=None):
    """Config 3b: per variable 6 unary ISTRUE + 4 binary EQUAL(v, (v+o) mod V),
    o in {1, 7, 101, V/8+3}: F = 10 V, E = 14 V.  Exercises the colouring."""
    W = n_weights or max(1, V // 10)
    rng = _rng(seed, 0)
    offsets = offsets or [1, 7, 101, V // 8 + 3]
    nu, nb = 6, len(offsets)
    k = nu + nb
    F = V * k
    arity = np.tile(np.array([1] * nu + [2] * nb, np.uint64), V)
    off = np.zeros(F + 1, np.uint64)
    np.cumsum(arity, out=off[1:])
    E = int(off[-1])
    func = np.tile(np.array([FUNC_ISTRUE] * nu + [FUNC_EQUAL] * nb, np.uint16), V)
    edge_vid = np.empty(E, np.uint64)
    v = np.arange(V, dtype=np.uint64)
    per = nu + 2 * nb
    ev = edge_vid.reshape(V, per)
    for j in range(nu):
        ev[:, j] = v
    for j, o in enumerate(offsets):
        ev[:, nu + 2 * j] = v
        ev[:, nu + 2 * j + 1] = (v + np.uint64(o)) % np.uint64(V)
    is_evid = rng.random(V) < 0.5
    val = (rng.random(V) < 0.7) & is_evid
    return RawGraph(
        var_role=is_evid.astype(np.uint8), var_init_value=val.astype(np.uint64),
        var_dtype=np.full(V, DTYPE_BOOLEAN, np.uint16), var_cardinality=np.full(V, 2, np.uint64),
        fac_func=func, fac_edge_offset=off,
        fac_weight_id=rng.integers(0, W, size=F, dtype=np.uint64),
        fac_feature_value=np.ones(F), edge_vid=edge_vid, edge_equal_to=np.ones(E, np.uint64),
        w_initial_value=np.zeros(W), w_is_fixed=np.zeros(W, np.uint8))


def cfg3c(V=10_000_000, n_weights=None, seed=1234):
    """Config 3c (not in BASELINE.json; times the generic path): per variable 6 unary ISTRUE +
    2 ternary IMPLY_NATURAL factors (v+1 and v+7 imply v; v+101 and v+211 imply v):
    F = 8 V, E = 12 V; every variable sits in 6 ternary factors."""
    from .rawgraph import FUNC_IMPLY_NATURAL
    W = n_weights or max(1, V // 10)
    rng = _rng(seed, 0)
    bodies = [(1, 7), (101, 211)]
    nu, nt = 6, len(bodies)
    k = nu + nt
    F = V * k
    arity = np.tile(np.array([1] * nu + [3] * nt, np.uint64), V)
    off = np.zeros(F + 1, np.uint64)
    np.cumsum(arity, out=off[1:])
    E = int(off[-1])
    func = np.tile(np.array([FUNC_ISTRUE] * nu + [FUNC_IMPLY_NATURAL] * nt, np.uint16), V)
    edge_vid = np.empty(E, np.uint64)
    v = np.arange(V, dtype=np.uint64)
    ev = edge_vid.reshape(V, nu + 3 * nt)
    for j in range(nu):
        ev[:, j] = v
    for j, (a, b) in enumerate(bodies):
        ev[:, nu + 3 * j] = (v + np.uint64(a)) % np.uint64(V)
        ev[:, nu + 3 * j + 1] = (v + np.uint64(b)) % np.uint64(V)
        ev[:, nu + 3 * j + 2] = v          # head
    is_evid = rng.random(V) < 0.5
    val = (rng.random(V) < 0.7) & is_evid
    return RawGraph(
        var_role=is_evid.astype(np.uint8), var_init_value=val.astype(np.uint64),
        var_dtype=np.full(V, DTYPE_BOOLEAN, np.uint16), var_cardinality=np.full(V, 2, np.uint64),
        fac_func=func, fac_edge_offset=off,
        fac_weight_id=rng.integers(0, W, size=F, dtype=np.uint64),
        fac_feature_value=np.ones(F), edge_vid=edge_vid, edge_equal_to=np.ones(E, np.uint64),
        w_initial_value=np.zeros(W), w_is_fixed=np.zeros(W, np.uint8))


def tied(n_evid=100_000, n_query=1_000, n_weights=1, seed=7, p_one=(0.7, 0.3, 0.9, 0.5)):
    """Heavily tied weights, the usual DeepDive shape (one weight per rule, many groundings):
    boolean variables with ONE unary ISTRUE factor each on weight (v mod n_weights); the first
    n_evid are evidence with value ~ Bernoulli(p_one[weight]), the rest query.  The maximum-
    likelihood weight of rule j is logit(p_one[j]) / 2 (P(x = 1) = sigmoid(2 w))."""
    V = n_evid + n_query
    rng = _rng(seed, 0)
    wid = (np.arange(V) % n_weights).astype(np.uint64)
    p = np.asarray(p_one, float)[wid.astype(np.int64) % len(p_one)]
    role = np.zeros(V, np.uint8); role[:n_evid] = 1
    val = ((rng.random(V) < p) & (role == 1)).astype(np.uint64)
    return RawGraph(
        var_role=role, var_init_value=val,
        var_dtype=np.full(V, DTYPE_BOOLEAN, np.uint16), var_cardinality=np.full(V, 2, np.uint64),
        fac_func=np.full(V, FUNC_ISTRUE, np.uint16), fac_edge_offset=np.arange(V + 1, dtype=np.uint64),
        fac_weight_id=wid, fac_feature_value=np.ones(V),
        edge_vid=np.arange(V, dtype=np.uint64), edge_equal_to=np.ones(V, np.uint64),
        w_initial_value=np.zeros(n_weights), w_is_fixed=np.zeros(n_weights, np.uint8))


def chain(n_chains=100_000, p_observed=0.5, seed=9, agree=(0.8, 0.65)):
    """Heavily tied weights on PAIRWISE factors (the test/partial_observation shape at scale):
    n_chains chains a - b - c of boolean variables, EQUAL(a, b) on weight 0 and EQUAL(b, c) on
    weight 1 (func 3, predicates 1, 1): two weights x n_chains factors.  a and c are evidence,
    b is observed with probability p_observed (else a query variable both chains sample);
    the hidden truth has b = a with probability agree[0] and c = b with probability agree[1]."""
    rng = _rng(seed, 0)
    n = n_chains
    V = 3 * n
    a = rng.random(n) < 0.5
    b = np.where(rng.random(n) < agree[0], a, ~a)
    c = np.where(rng.random(n) < agree[1], b, ~b)
    obs = rng.random(n) < p_observed
    role = np.ones(V, np.uint8)
    role[1::3] = obs
    val = np.zeros(V, np.uint64)
    val[0::3] = a; val[1::3] = b & obs; val[2::3] = c
    F = 2 * n
    i = np.arange(n, dtype=np.uint64)
    edge_vid = np.empty(2 * F, np.uint64)
    ev = edge_vid.reshape(n, 4)
    ev[:, 0] = 3 * i; ev[:, 1] = 3 * i + 1; ev[:, 2] = 3 * i + 1; ev[:, 3] = 3 * i + 2
    return RawGraph(
        var_role=role, var_init_value=val,
        var_dtype=np.full(V, DTYPE_BOOLEAN, np.uint16), var_cardinality=np.full(V, 2, np.uint64),
        fac_func=np.full(F, FUNC_EQUAL, np.uint16), fac_edge_offset=2 * np.arange(F + 1, dtype=np.uint64),
        fac_weight_id=np.tile(np.array([0, 1], np.uint64), n), fac_feature_value=np.ones(F),
        edge_vid=edge_vid, edge_equal_to=np.ones(2 * F, np.uint64),
        w_initial_value=np.zeros(2), w_is_fixed=np.zeros(2, np.uint8))


def cfg4(V=5_000_000, card=8, seed=1234, learn=False, shard=0):
    """Config 4: V categorical variables of cardinality `card` (implicit dense domain),
    one unary AND_CATEGORICAL factor per (v, d) with weight id d (the
    biased_coin_with_multinomial shape).  Infer-only: weights fixed ~ N(0,1), all
    query.  Learn: weights init 0, first half evidence ~ Categorical(softmax(w*))."""
    rng = _rng(seed, shard)
    wstar = _rng(seed, 10_000).normal(0.0, 1.0, card)
    F = V * card
    role = np.zeros(V, np.uint8)
    init = np.zeros(V, np.uint64)
    if learn:
        p = np.exp(wstar - wstar.max()); p /= p.sum()
        h = V // 2
        role[:h] = 1
        init[:h] = rng.choice(card, size=h, p=p).astype(np.uint64)
        w0, fixed = np.zeros(card), np.zeros(card, np.uint8)
    else:
        w0, fixed = wstar, np.ones(card, np.uint8)
    d = np.tile(np.arange(card, dtype=np.uint64), V)
    return RawGraph(
        var_role=role, var_init_value=init,
        var_dtype=np.full(V, DTYPE_CATEGORICAL, np.uint16),
        var_cardinality=np.full(V, card, np.uint64),
        fac_func=np.full(F, FUNC_AND_CATEGORICAL, np.uint16),
        fac_edge_offset=np.arange(F + 1, dtype=np.uint64),
        fac_weight_id=d.copy(), fac_feature_value=np.ones(F),
        edge_vid=np.repeat(np.arange(V, dtype=np.uint64), card), edge_equal_to=d,
        w_initial_value=w0, w_is_fixed=fixed)


def cfg4b(V=2_000_000, card=8, n_weights=None, seed=1234, learn=True):
    """Config 4b (not in BASELINE.json; times categorical tiles with non-unary factors, the
    linear-chain shape): V categorical variables of cardinality `card`; per (v, d) one unary
    AND_CATEGORICAL factor and one binary agreement factor AND(v == d, v+1 == d).  Every value
    row holds three records (one unary, two memberships): F = 2 V card, E = 3 V card.  Weights:
    2 * card tied ones (None) or `n_weights` at random."""
    rng = _rng(seed, 0)
    F = 2 * V * card
    arity = np.tile(np.array([1] * card + [2] * card, np.uint64), V)
    off = np.zeros(F + 1, np.uint64)
    np.cumsum(arity, out=off[1:])
    E = int(off[-1])
    v = np.arange(V, dtype=np.uint64)
    d = np.arange(card, dtype=np.uint64)
    per = 3 * card
    ev = np.empty((V, per), np.uint64)
    eq = np.empty((V, per), np.uint64)
    ev[:, :card] = v[:, None]
    eq[:, :card] = d[None, :]
    ev[:, card::2] = v[:, None]
    ev[:, card + 1::2] = ((v + np.uint64(1)) % np.uint64(V))[:, None]
    eq[:, card::2] = d[None, :]
    eq[:, card + 1::2] = d[None, :]
    if n_weights is None:
        W = 2 * card
        wid = np.tile(np.arange(2 * card, dtype=np.uint64), V)
    else:
        W = n_weights
        wid = rng.integers(0, W, size=F, dtype=np.uint64)
    role = np.zeros(V, np.uint8)
    init = np.zeros(V, np.uint64)
    if learn:
        role[: V // 2] = 1
        init[: V // 2] = rng.integers(0, card, size=V // 2, dtype=np.uint64)
        w0, fixed = np.zeros(W), np.zeros(W, np.uint8)
    else:
        w0, fixed = rng.normal(0.0, 0.5, W), np.ones(W, np.uint8)
    return RawGraph(
        var_role=role, var_init_value=init,
        var_dtype=np.full(V, DTYPE_CATEGORICAL, np.uint16),
        var_cardinality=np.full(V, card, np.uint64),
        fac_func=np.full(F, FUNC_AND_CATEGORICAL, np.uint16), fac_edge_offset=off,
        fac_weight_id=wid, fac_feature_value=np.ones(F),
        edge_vid=ev.reshape(-1), edge_equal_to=eq.reshape(-1),
        w_initial_value=w0, w_is_fixed=fixed)


# ---- config 5b: the 3b mix over variable-block shards, generated shard by shard ----------
# Every attribute is a pure function of a GLOBAL id (splitmix64 of (seed, kind, id)), so a rank
# can build its own block -- and the factors other blocks own that touch it -- without anybody
# ever holding the 10^8-variable graph.

def _mix64(x, salt):
    """splitmix64 finaliser over uint64 arrays (wraps modulo 2^64 by construction)."""
    with np.errstate(over="ignore"):
        z = x.astype(np.uint64) + np.uint64((0x9E3779B97F4A7C15 * (salt + 1)) & 0xFFFFFFFFFFFFFFFF)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _unit(x, salt):
    return (_mix64(x, salt) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def cfg5b_offsets(V_total):
    return [1, 7, 101, V_total // 8 + 3]


def cfg5b_shard(V_total, begin, end, n_weights, seed=1234, offsets=None):
    """The block [begin, end) of the config-5b graph: V_total boolean variables, 50 % evidence
    ~ Bernoulli(0.7), per variable v 6 unary ISTRUE factors and 4 binary EQUAL(v, (v + o) mod
    V_total), o in {1, 7, 101, V_total/8 + 3} (global factor id = 10 v + j; weight id = hash of
    the factor id).  Returns (local RawGraph, ghost global ids): the owned variables (local id =
    global - begin), then the ghosts -- remote endpoints of local factors -- in ascending global
    id; local factors = the 10 factors of every owned variable, then (ascending global factor
    id) the binary factors OTHER blocks own whose second endpoint is owned here: a factor
    spanning two shards lives on both, as sampler_amd.shard.make_shard does for a graph in
    memory.  With 8 equal blocks the last offset makes every variable read a neighbour in the
    next block: a dense halo; the small offsets give thin ones."""
    offsets = offsets or cfg5b_offsets(V_total)
    W = n_weights
    n = end - begin
    nu, nb = 6, len(offsets)
    k = nu + nb
    own = np.arange(begin, end, dtype=np.uint64)
    Vt = np.uint64(V_total)
    # own factors, variable-major
    per = nu + 2 * nb
    ev = np.empty((n, per), np.uint64)
    for j in range(nu):
        ev[:, j] = own
    for j, o in enumerate(offsets):
        ev[:, nu + 2 * j] = own
        ev[:, nu + 2 * j + 1] = (own + np.uint64(o)) % Vt
    fid_own = (own[:, None] * np.uint64(k) + np.arange(k, dtype=np.uint64)[None, :]).ravel()
    # incoming: factor (v', u), v' = (u - o) mod V_total outside the block, u owned
    inc_fid, inc_src, inc_dst = [], [], []
    for j, o in enumerate(offsets):
        src = (own + Vt - np.uint64(o % V_total)) % Vt
        outside = (src < np.uint64(begin)) | (src >= np.uint64(end))
        inc_src.append(src[outside]); inc_dst.append(own[outside])
        inc_fid.append(src[outside] * np.uint64(k) + np.uint64(nu + j))
    inc_fid = np.concatenate(inc_fid); inc_src = np.concatenate(inc_src); inc_dst = np.concatenate(inc_dst)
    order = np.argsort(inc_fid, kind="stable")
    inc_fid, inc_src, inc_dst = inc_fid[order], inc_src[order], inc_dst[order]
    n_inc = len(inc_fid)
    edge_global = np.concatenate([ev.ravel(), np.stack([inc_src, inc_dst], 1).ravel()])
    remote = (edge_global < np.uint64(begin)) | (edge_global >= np.uint64(end))
    ghosts = np.unique(edge_global[remote])
    local = np.where(remote, np.uint64(n) + np.searchsorted(ghosts, edge_global).astype(np.uint64),
                     edge_global - np.uint64(begin))
    arity = np.concatenate([np.tile(np.array([1] * nu + [2] * nb, np.uint64), n), np.full(n_inc, 2, np.uint64)])
    off = np.zeros(len(arity) + 1, np.uint64)
    np.cumsum(arity, out=off[1:])
    func = np.concatenate([np.tile(np.array([FUNC_ISTRUE] * nu + [FUNC_EQUAL] * nb, np.uint16), n),
                           np.full(n_inc, FUNC_EQUAL, np.uint16)])
    fid = np.concatenate([fid_own, inc_fid])
    wid = _mix64(fid, 3 * seed + 2) % np.uint64(W)
    ids = np.concatenate([own, ghosts])
    is_evid = _unit(ids, 3 * seed) < 0.5
    val = (_unit(ids, 3 * seed + 1) < 0.7) & is_evid
    nv = len(ids)
    g = RawGraph(
        var_role=is_evid.astype(np.uint8), var_init_value=val.astype(np.uint64),
        var_dtype=np.full(nv, DTYPE_BOOLEAN, np.uint16), var_cardinality=np.full(nv, 2, np.uint64),
        fac_func=func, fac_edge_offset=off, fac_weight_id=wid, fac_feature_value=np.ones(len(fid)),
        edge_vid=local, edge_equal_to=np.ones(len(local), np.uint64),
        w_initial_value=np.zeros(W), w_is_fixed=np.zeros(W, np.uint8), num_ghost_variables=len(ghosts))
    return g, ghosts


def cfg5b(V_total, n_weights, seed=1234):
    """The WHOLE config-5b graph (small sizes: tests): one block, no ghosts."""
    return cfg5b_shard(V_total, 0, V_total, n_weights, seed)[0]


def cfg4_closed_form(g: RawGraph, card):
    w = g.w_initial_value[:card]
    p = np.exp(w - w.max())
    return p / p.sum()
