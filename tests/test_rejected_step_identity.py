"""The arithmetic identity behind mcmc_chain_tab's branch-free rejected step: a certainly rejected proposal is
`flip` + flip back (pseudo_mcmc.rs:739,746) on the two touched clusters -- (s - x) + x on the one the read leaves, (s + x) - x on the
one it would join -- and the kernel computes, for EVERY cluster, (s + m) - m with m = x * f, f in {-1, +1, 0}.  IEEE-754 doubles in
numpy are what the device's v_mul_f64 / v_add_f64 compute: the bits must agree, including the sign of zero."""
import numpy as np


def bits(a):
    return np.asarray(a, dtype=np.float64).view(np.uint64)


def test_branch_free_rejected_step_is_the_reference_arithmetic():
    rng = np.random.default_rng(5)
    s = np.concatenate([rng.normal(0, 50, 200000), rng.normal(0, 1e-6, 20000), np.zeros(2000), rng.normal(0, 1e6, 20000)])
    x = np.concatenate([rng.normal(0, 5, 200000), rng.normal(0, 50, 20000), rng.normal(0, 5, 2000), np.zeros(20000)])
    x[rng.random(len(x)) < 0.05] = 0.0          # compress_small_gains leaves exact zeros
    with np.errstate(all="ignore"):
        leave = (s - x) + x                      # the cluster the read leaves
        join = (s + x) - x                       # the cluster it would join
        m = x * -1.0
        assert np.array_equal(bits((s + m) - m), bits(leave))
        m = x * 1.0
        assert np.array_equal(bits((s + m) - m), bits(join))
        m = x * 0.0                              # (+0 or -0, by the sign of x)
        assert np.array_equal(bits((s + m) - m), bits(s))   # s is never -0: sums grow from +0 by additions
    # the one value the third identity does not hold for, and why it cannot occur
    neg_zero = np.float64(-0.0)
    assert bits((neg_zero + 0.0) - 0.0) != bits(neg_zero)
    acc = np.float64(0.0)
    for v in (-0.0, 3.5, -3.5, -0.0):
        acc = acc + np.float64(v)
    assert bits(acc) == bits(np.float64(0.0))    # +0: an accumulator that starts at +0 never becomes -0
