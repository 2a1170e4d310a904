"""The reference's replica arithmetic (-c N, n_datacopy: src/dimmwitted.cc:97-119, 162-216, 264-282)
emulated with the oracle's REFERENCE mode (byte-pinned to the real binary, oracle/dw_oracle.cc): N
copies of the state, every learning round each copy runs one sequential sample_sgd sweep, the
weights are averaged and copied back; ceil(l / N) rounds, ceil(i / N) inference rounds, tallies summed.
Several such runs with different erand48 seeds give the spread a fixture's result has under -c N --
the yardstick for the product's -c N where the fixture's own tolerance was written for -c 1
(biased_coin_continuous: half as many rounds leave the step at 7e-4 instead of 4e-6).  The real
binary refuses -c 2 on a box with one NUMA node ("n_datacopy must be a divisor of the number of NUMA
nodes"), so the emulation is what can run everywhere."""
import numpy as np

from oracle import binding as orc


def reference_replica_runs(raw, n_copies, n_learn, n_infer, stepsize, decay, n_runs=8, **flags):
    """-> (weights [n_runs, W], marginals [n_runs, num_values])"""
    W, P = [], []
    for run in range(n_runs):
        copies = [orc.Oracle(raw, **flags) for _ in range(n_copies)]
        for k, o in enumerate(copies):
            o.set_workers(1)
            o.set_seed(0, 1 + 7 * run + k, 11 + 3 * run, 101 + k)
        cur = stepsize
        for _ in range((n_learn + n_copies - 1) // n_copies):
            for o in copies:
                o.sample_sgd(cur)
            mean = sum(np.array(o.weights) for o in copies) / n_copies
            for o in copies:
                o.weights[:] = mean
            cur *= decay
        for o in copies:
            o.clear_tallies()
        for _ in range((n_infer + n_copies - 1) // n_copies):
            for o in copies:
                o.sample()
        tallies = sum(np.array(o.tallies, np.float64) for o in copies)
        ns = sum(np.array(o.nsamples, np.float64) for o in copies)
        base = np.array(copies[0].var_val_base, np.int64)
        per_value = np.repeat(ns, np.diff(np.append(base, len(tallies))))
        W.append(np.array(copies[0].weights))
        P.append(tallies / np.maximum(per_value, 1))
    return np.array(W), np.array(P)
