"""Reader/writer for the reference's big-endian binary factor-graph files
(/root/reference/doc/binary_format.md; readers in src/binary_format.cc:23-226,
writers in src/text2bin.cc:19-243).  numpy structured dtypes do the byte swapping.

The C++ `dw` binary has its own native loader (sampler_amd/csrc/binary_format.cc);
this module is the Python-side mirror used by tests, bench.py and the multi-GPU
harness.
"""
import os

import numpy as np

from .rawgraph import RawGraph

VAR_DT = np.dtype([("vid", ">u8"), ("role", "u1"), ("init", ">u8"),
                   ("dtype", ">u2"), ("card", ">u8")])          # 27 B
WEIGHT_DT = np.dtype([("wid", ">u8"), ("fixed", "u1"), ("value", ">f8")])  # 17 B


def read_meta(path):
    """numWeights,numVariables,numFactors,numEdges (src/binary_format.cc:23-36);
    trailing fields and a missing newline are tolerated."""
    txt = open(path).read()
    parts = txt.split(",")
    nw, nv, nf, ne = (int(p.strip() or 0) for p in parts[:4])
    return dict(num_weights=nw, num_variables=nv, num_factors=nf, num_edges=ne)


def _read_all(paths):
    if isinstance(paths, (str, os.PathLike)):
        paths = [paths]
    return b"".join(open(p, "rb").read() for p in paths)


def read_graph(meta, variables, weights, factors, domains=None):
    """Load a graph into columnar form. Each of variables/weights/factors/domains is
    a path or a list of paths (several files of one kind are concatenated, which is
    what the reference's per-file loader threads amount to for a single file list)."""
    m = read_meta(meta)
    V, W = m["num_variables"], m["num_weights"]
    vb = np.frombuffer(_read_all(variables), dtype=VAR_DT)
    assert len(vb) == V, "variable count != meta"
    role = np.zeros(V, np.uint8); init = np.zeros(V, np.uint64)
    dt = np.zeros(V, np.uint16); card = np.zeros(V, np.uint64)
    vid = vb["vid"].astype(np.int64)
    role[vid] = vb["role"]; init[vid] = vb["init"]; dt[vid] = vb["dtype"]; card[vid] = vb["card"]
    wb = np.frombuffer(_read_all(weights), dtype=WEIGHT_DT)
    assert len(wb) == W, "weight count != meta"
    wv = np.zeros(W, np.float64); wf = np.zeros(W, np.uint8)
    wid = wb["wid"].astype(np.int64)
    wv[wid] = wb["value"]; wf[wid] = wb["fixed"]
    # factors: variable-length records
    fb = _read_all(factors)
    F, E = m["num_factors"], m["num_edges"]
    func = np.zeros(F, np.uint16); off = np.zeros(F + 1, np.uint64)
    fw = np.zeros(F, np.uint64); fv = np.zeros(F, np.float64)
    ev = np.zeros(E, np.uint64); eq = np.zeros(E, np.uint64)
    pos = 0; e = 0; f = 0
    u2 = np.dtype(">u2"); u8 = np.dtype(">u8"); f8 = np.dtype(">f8")
    n = len(fb)
    # fast path: every factor has the same arity (checked, not assumed)
    if F and E % F == 0 and n == F * (26 + 16 * (E // F)):
        a = E // F
        dtf = np.dtype([("func", ">u2"), ("arity", ">u8"), ("pairs", ">u8", (a, 2)),
                        ("wid", ">u8"), ("val", ">f8")])
        rec = np.frombuffer(fb, dtf)
        if (rec["arity"] == a).all():
            func[:] = rec["func"]; fw[:] = rec["wid"]; fv[:] = rec["val"]
            ev[:] = rec["pairs"][:, :, 0].reshape(-1); eq[:] = rec["pairs"][:, :, 1].reshape(-1)
            off[:] = np.arange(F + 1, dtype=np.uint64) * np.uint64(a)
            pos = n; f = F; e = E
    while pos < n:
        func[f] = np.frombuffer(fb, u2, 1, pos)[0]; pos += 2
        ar = int(np.frombuffer(fb, u8, 1, pos)[0]); pos += 8
        pairs = np.frombuffer(fb, u8, 2 * ar, pos); pos += 16 * ar
        ev[e:e + ar] = pairs[0::2]; eq[e:e + ar] = pairs[1::2]
        e += ar
        fw[f] = np.frombuffer(fb, u8, 1, pos)[0]; pos += 8
        fv[f] = np.frombuffer(fb, f8, 1, pos)[0]; pos += 8
        f += 1
        off[f] = e
    assert f == F and e == E, "factor/edge count != meta (%d/%d vs %d/%d)" % (f, e, F, E)
    dom_vid, dom_off, dom_val, dom_tr = [], [0], [], []
    if domains:
        db = _read_all(domains)
        pos = 0
        while pos < len(db):
            v = int(np.frombuffer(db, u8, 1, pos)[0]); pos += 8
            c = int(np.frombuffer(db, u8, 1, pos)[0]); pos += 8
            rec = np.frombuffer(db, np.dtype([("v", ">u8"), ("t", ">f8")]), c, pos); pos += 16 * c
            dom_vid.append(v); dom_val.extend(rec["v"].tolist()); dom_tr.extend(rec["t"].tolist())
            dom_off.append(len(dom_val))
    return RawGraph(role, init, dt, card, func, off, fw, fv, ev, eq, wv, wf,
                    np.array(dom_vid, np.uint64), np.array(dom_off, np.uint64),
                    np.array(dom_val, np.uint64), np.array(dom_tr, np.float64))


def read_graph_dir(d):
    p = lambda n: os.path.join(d, n)
    dom = p("graph.domains") if os.path.exists(p("graph.domains")) else None
    return read_graph(p("graph.meta"), p("graph.variables"), p("graph.weights"),
                      p("graph.factors"), dom)


def write_graph(g: RawGraph, d):
    """Write graph.{meta,variables,weights,factors[,domains]} into directory d in the
    reference's binary format (vectorised for uniform-arity graphs)."""
    os.makedirs(d, exist_ok=True)
    V, F, E, W = g.num_variables, g.num_factors, g.num_edges, g.num_weights
    with open(os.path.join(d, "graph.meta"), "w") as f:
        f.write("%d,%d,%d,%d\n" % (W, V, F, E))
    vb = np.zeros(V, VAR_DT)
    vb["vid"] = np.arange(V); vb["role"] = g.var_role; vb["init"] = g.var_init_value
    vb["dtype"] = g.var_dtype; vb["card"] = g.var_cardinality
    vb.tofile(os.path.join(d, "graph.variables"))
    wb = np.zeros(W, WEIGHT_DT)
    wb["wid"] = np.arange(W); wb["fixed"] = g.w_is_fixed; wb["value"] = g.w_initial_value
    wb.tofile(os.path.join(d, "graph.weights"))
    arity = np.diff(g.fac_edge_offset.astype(np.int64))
    # the arity pattern's period (uniform graphs: 1; the synthetic mixes repeat per variable, e.g. six
    # unary + four binary factors: 10): ONE structured record of `period` factors, written by slices
    period = 0
    for P in range(1, 65):
        if F and F % P == 0 and (arity.reshape(-1, P) == arity[:P]).all():
            period = P
            break
    with open(os.path.join(d, "graph.factors"), "wb") as f:
        if period:
            pat = [int(x) for x in arity[:period]]
            fields = []
            for j, a in enumerate(pat):
                fields += [("func%d" % j, ">u2"), ("arity%d" % j, ">u8"), ("pairs%d" % j, ">u8", (a, 2)),
                           ("wid%d" % j, ">u8"), ("val%d" % j, ">f8")]
            dt = np.dtype(fields)
            epat = np.concatenate(([0], np.cumsum(pat)))      # edge offsets inside one period
            EP = int(epat[-1])
            step = max(1, 8_000_000 // period)          # (slices: the 10^8 factors of config 3 are 4.2 GB of records)
            G = F // period
            for lo in range(0, G, step):
                hi = min(G, lo + step)
                rec = np.zeros(hi - lo, dt)
                ev = g.edge_vid[lo * EP:hi * EP].reshape(hi - lo, EP)
                eq = g.edge_equal_to[lo * EP:hi * EP].reshape(hi - lo, EP)
                for j, a in enumerate(pat):
                    sl = slice(lo * period + j, hi * period, period)
                    rec["func%d" % j] = g.fac_func[sl]; rec["arity%d" % j] = a
                    rec["pairs%d" % j][:, :, 0] = ev[:, epat[j]:epat[j + 1]]
                    rec["pairs%d" % j][:, :, 1] = eq[:, epat[j]:epat[j + 1]]
                    rec["wid%d" % j] = g.fac_weight_id[sl]; rec["val%d" % j] = g.fac_feature_value[sl]
                rec.tofile(f)
        else:
            # mixed arities: records of 26 + 16 * arity bytes, scattered field by field into a byte
            # buffer per slice of factors (numpy; the per-factor loop took 37 s per 10^7 factors)
            off = g.fac_edge_offset.astype(np.int64)
            step = 4_000_000
            k8 = np.arange(8, dtype=np.int64)

            def put(buf, at, values, dt, width):
                b = np.ascontiguousarray(values).astype(dt).view(np.uint8).reshape(-1, width)
                buf[at[:, None] + np.arange(width, dtype=np.int64)] = b
            for lo in range(0, F, step):
                hi = min(F, lo + step)
                ar = arity[lo:hi]
                size = 26 + 16 * ar
                start = np.concatenate(([0], np.cumsum(size)))[:-1]
                buf = np.zeros(int(size.sum()), np.uint8)
                put(buf, start, g.fac_func[lo:hi], ">u2", 2)
                put(buf, start + 2, ar, ">u8", 8)
                e0, e1 = int(off[lo]), int(off[hi])
                # byte position of every edge: its factor's start + 10 + 16 * (index inside the factor)
                fac_of = np.repeat(np.arange(hi - lo, dtype=np.int64), ar)
                inside = np.arange(e1 - e0, dtype=np.int64) - np.repeat(off[lo:hi] - e0, ar)
                epos = start[fac_of] + 10 + 16 * inside
                put(buf, epos, g.edge_vid[e0:e1], ">u8", 8)
                put(buf, epos + 8, g.edge_equal_to[e0:e1], ">u8", 8)
                tail = start + 10 + 16 * ar
                put(buf, tail, g.fac_weight_id[lo:hi], ">u8", 8)
                put(buf, tail + 8, g.fac_feature_value[lo:hi], ">f8", 8)
                buf.tofile(f)
            del k8
    if len(g.dom_vid):
        with open(os.path.join(d, "graph.domains"), "wb") as f:
            for b in range(len(g.dom_vid)):
                lo, hi = int(g.dom_offset[b]), int(g.dom_offset[b + 1])
                f.write(np.array(g.dom_vid[b], ">u8").tobytes())
                f.write(np.array(hi - lo, ">u8").tobytes())
                rec = np.zeros(hi - lo, np.dtype([("v", ">u8"), ("t", ">f8")]))
                rec["v"] = g.dom_value[lo:hi]; rec["t"] = g.dom_truthiness[lo:hi]
                f.write(rec.tobytes())
