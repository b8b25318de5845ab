"""
snmpc.py -- scenario (sigma-point / Monte-Carlo) fan-out around the batched solver.

Host-side restatement of the one-off PCE set-up of the reference's stochastic controller
(Model_Predictive_Controller/Stochastic_NMPC/stochastic_mpc_utils.py:17-91, run once at construction,
SNMPC_class.py:39-177) plus the batch restatement of its per-step work: the sigma points become the
batch axis (independent 8-state OCPs sharing yref), the PCE mean / variance become a reduction over
each scenario group (SNMPC_acados_settings.py:116-133). chaospy (Hammersley sampling) is not available;
the sampling recipe below reproduces the sigma points stored in the reference's acados_ocp_SNMPC.json
(tests/golden/pce.npz) to 1e-14.
"""
import itertools
import math

import numpy as np
from scipy.special import ndtri

from . import config as _config
from .solver import BatchedOcpSolver


def hermite(x, n):
    """stochastic_mpc_utils.py:17-25 (note: divides by sqrt(n!) at EVERY recursion level, reproduced literally)."""
    if n == 0:
        return (1.0 + 0.0 * x) / math.sqrt(math.factorial(0))
    if n == 1:
        return x / math.sqrt(math.factorial(1))
    return (x * hermite(x, n - 1) - (n - 1) * hermite(x, n - 2)) / math.sqrt(math.factorial(n))


def alpha_generation(n_rand, degree):
    """Multi-indices with |alpha| <= degree, ascending total degree, ties in itertools.product order (:27-38)."""
    al = [a for a in itertools.product(range(degree + 1), repeat=n_rand) if sum(a) <= degree]
    al.sort(key=sum)                       # stable
    return np.array(al, dtype=int)


def pce_basis(w, alphas):
    """Phi(w) for one sample w (n_rand,) -> (L,) (:40-54, gaussian)."""
    return np.array([np.prod([hermite(w[j], int(a[j])) for j in range(len(a))]) for a in alphas])


def _vdc(i, base):
    v, f = 0.0, 1.0 / base
    while i > 0:
        v += (i % base) * f
        i //= base
        f /= base
    return v


_PRIMES = (2, 3, 5, 7, 11, 13, 17, 19)


def hammersley_normal(n_samples, n_rand):
    """chaospy.J(Normal(0,1)^n).sample(n, rule='hammersley') (:59-63): van-der-Corput in the first n-1 prime
    bases (burn-in = largest base used), last dimension equispaced, mapped through the normal inverse CDF.
    Returns (n_rand, n_samples)."""
    bases = _PRIMES[:max(n_rand - 1, 0)]
    burn = max(bases) if bases else 0
    u = np.zeros((n_rand, n_samples))
    for k in range(n_samples):
        for d, bse in enumerate(bases):
            u[d, k] = _vdc(burn + 1 + k, bse)
        u[n_rand - 1, k] = (k + 1) / (n_samples + 1)
    return ndtri(u)


def pce_matrix(w_samples, alphas):
    """A = inv(Phi' Phi) Phi' (L x n_samples), (:66-74)."""
    Phi = np.array([pce_basis(w_samples[:, i], alphas) for i in range(w_samples.shape[1])])
    return np.linalg.inv(Phi.T @ Phi) @ Phi.T


def x0_offsets(w_samples, stds):
    """(n_samples, 8) offsets stds (.) w on the states with non-zero std (compute_x0dist, :78-91)."""
    stds = np.asarray(stds, dtype=float)
    act = np.nonzero(stds)[0]
    off = np.zeros((w_samples.shape[1], 8))
    for s in range(w_samples.shape[1]):
        off[s, act] = stds[act] * w_samples[:, s]
    return off


def compute_x0dist(x0, w_samples, stds):
    """(n_samples+1, 8): row 0 = x0, row s = x0 + offsets[s-1]."""
    return np.vstack([np.asarray(x0, float)[None], np.asarray(x0, float)[None] + x0_offsets(w_samples, stds)])


class ScenarioSNMPC:
    """Batch restatement of the SNMPC step: P poses x (1 nominal + S scenarios) independent nominal OCPs."""

    def __init__(self, n_poses, n_samples=None, stds=None, degree=None, N=40, dt=0.08, nsub=3, device=0,
                 w_samples=None, cfg=None):
        cfg = cfg or _config.default_config()
        m = cfg["mpc"]
        self.stds = np.asarray(m["stds"] if stds is None else stds, dtype=float)
        self.n_rand = int(np.count_nonzero(self.stds))
        self.degree = m["expansion_degree"] if degree is None else degree
        self.alphas = alpha_generation(self.n_rand, self.degree)
        if w_samples is None:
            n_samples = m["n_samples"] if n_samples is None else n_samples
            w_samples = hammersley_normal(n_samples, self.n_rand)
        self.w = np.asarray(w_samples, dtype=float)
        self.S = self.w.shape[1]
        self.P = int(n_poses)
        self.A = pce_matrix(self.w, self.alphas) if self.S >= len(self.alphas) else None
        self.offsets = x0_offsets(self.w, self.stds)
        self.gamma = m["gamma"]
        self.kappa = math.sqrt((1 - self.gamma) / self.gamma)
        self.solver = BatchedOcpSolver(N=N, dt=dt, nsub=nsub, batch=self.P * (self.S + 1), device=device, cfg=cfg)
        self.solver.install_reference_ocp()
        self.N = N

    def solve(self, x0_poses, yref_poses, cold=True):
        """x0_poses (P,8), yref_poses (P,N+1,6). Returns status, nominal u0 (P,2), PCE mean / var of x_1 (P,8)."""
        s = self.solver
        s.set_x0_fanout(x0_poses, self.offsets)
        s.set_yref_all(np.repeat(np.asarray(yref_poses, float), self.S + 1, axis=0))
        if cold:
            s.cold_start()
        st = s.solve()
        X, U = s.get_iterate()
        mean = var = None
        if self.A is not None:
            mean, var = s.pce_moments("x", 1, self.A)
        return st, U[::self.S + 1, 0], mean, var
