"""Statistical comparators for Monte-Carlo parity."""
import numpy as np
from scipy import stats


def z_scores_vs_exact(p_hat, p_exact, n):
    """z_v = (p_hat - p) / sqrt(p (1-p) / n); for i.i.d. draws ~ N(0,1)."""
    var = np.clip(p_exact * (1 - p_exact), 1e-12, None) / n
    return (p_hat - p_exact) / np.sqrt(var)


def z_scores_two_sample(p1, n1, p2, n2):
    pbar = (p1 * n1 + p2 * n2) / (n1 + n2)
    var = np.clip(pbar * (1 - pbar), 1e-12, None) * (1.0 / n1 + 1.0 / n2)
    return (p1 - p2) / np.sqrt(var)


def ks_normal(z):
    """one-sample KS of z against N(0,1): returns p-value."""
    return stats.kstest(z, "norm").pvalue


def ks_two_sample(a, b):
    return stats.ks_2samp(a, b).pvalue
