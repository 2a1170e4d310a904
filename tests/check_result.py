"""The acceptance checks of the reference's end-to-end fixtures
(/root/reference/test/<fixture>/check_result, awk scripts) restated as data +
one checker over the two output files' text."""
import math


def _parse(txt):
    return [line.split() for line in txt.strip().splitlines() if line.strip()]


def check(fixture, weights_text, marginals_text):
    """Raises AssertionError with the offending line, like the awk scripts exit(1)."""
    W = [(int(a), float(b)) for a, b in _parse(weights_text)]
    M = [(int(a), int(b), float(c)) for a, b, c in _parse(marginals_text)]

    def near(x, exp, eps, what):
        assert exp - eps <= x <= exp + eps, "%s: %g not within %g of %g" % (what, x, eps, exp)

    if fixture == "biased_coin":
        for i, w in W: near(w, 1.0, 0.1, "weight %d" % i)
        for v, e, p in M: near(p, 0.89, 0.03, "var %d" % v)
    elif fixture == "biased_coin_continuous":
        for i, w in W: near(w, 0.0, 0.1, "weight %d" % i)
        for v, e, p in M: near(p, 0.5, 0.03, "var %d" % v)
    elif fixture == "biased_coin_with_multinomial":
        for i, w in W: near(w, -1.05 if i == 0 else 1.05, 0.1, "weight %d" % i)
        for v, e, p in M:
            if e == 1: near(p, 0.89, 0.03, "var %d" % v)
    elif fixture == "biased_coin_truthiness":
        for i, w in W: near(w, -0.7 if i == 0 else 0.7, 0.1, "weight %d" % i)
        for v, e, p in M:
            if e == 1: near(p, 0.80, 0.03, "var %d" % v)
    elif fixture == "partial_observation":
        for i, w in W: assert w > 0, "weight %d <= 0" % i
        for v, e, p in M: assert p > 0.9, "var %d has prob %g <= 0.9" % (v, p)
    elif fixture == "sparse_domains":
        for i, w in W:
            if i < 4: near(w, math.log(i + 1), 0.2, "weight %d" % i)
        e2, e3 = math.exp(2), math.exp(3)
        special = {
            15: {1: 1.0}, 17: {1: 1.0},
            14: {0: e2 / (e2 + e3 + 1), 1: e3 / (e2 + e3 + 1), 3: 1 / (e2 + e3 + 1)},
            16: {1: e3 / (e2 + e3), 3: e2 / (e2 + e3)},
            18: {0: 1 / 7.0, 1: 2 / 7.0, 3: 4 / 7.0},
            19: {1: 1 / 3.0, 3: 2 / 3.0},
        }
        for v, e, p in M:
            exp = (e + 1) / 10.0
            if v in special:
                assert e in special[v], "var %d has a value outside its domain" % v
                exp = special[v][e]
            near(p, exp, 0.04, "var %d category %d" % (v, e))
    elif fixture == "sparse_multinomial2":
        table = {0: {0: 0, 1: 1, 2: 0}, 1: {1: 1, 2: 0, 3: 0}, 2: {1: 0.05, 2: 0, 3: 0.95}}
        for v, e, p in M:
            assert e in table[v], "var %d has a value outside its domain" % v
            near(p, table[v][e], 0.04, "var %d category %d" % (v, e))
    else:
        raise KeyError(fixture)
