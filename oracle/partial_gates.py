"""TEST INFRASTRUCTURE ONLY - a host restatement of the sweep DECISIONS of the opt-in partial ("selective") re-orthogonalisation loops.

The reference has no selective re-orthogonalisation (only the full sweep, Python/Regular/Lanczos.py:233-251; commented-out
alternatives at :240-245), so there is nothing in it to mirror: what is pinned here is the build's own device logic, against an
independent restatement in plain Python floats (IEEE double, same expression order, no FMA - the kernels are compiled with
-ffp-contract=off):

* ``simon_gates``     - Simon's omega-recurrence (H. D. Simon 1984) exactly as ``k_omega`` (lanczos_amd/csrc/lz_reorth.hip) advances it:
                        engine 7, and the host-decided loop of knob 18 = 1;
* ``lookahead_gates`` - the one-step look-ahead of ``k_partial_onered_post``: engine 8 (one all-reduce per step).

Inputs are the coefficients a run delivered - ``alpha[0..n)`` and ``hb[0..n)``, ``hb[k]`` = the norm that formed basis vector k
(``hb[0]``: the warm-up residual's norm ``||A v0 - (v0 . A v0) v0||``, which the reference parks in ``beta[-1]`` and later overwrites;
``hb[k] = beta[k-1]`` for k >= 1) - so a replay follows the very run it is compared with: the tests fetch the device's own per-step
record (``lz_last_sweep_log``) and demand equality.
"""
import math

EPS = 2.220446049250313e-16
THRESH = 1.4901161193847656e-08  # sqrt(eps)


def _row(n):
    return [0.0] * (n + 1)


def _advance(W, alpha, hb, j, a_last, hbj, normA, n):
    """omega_{j,:} from omega_{j-1,:} and omega_{j-2,:} (k_omega's loop, expression for expression); returns (row, worst)"""
    cur, prev = W[(j + 2) % 3], W[(j + 1) % 3]
    hb_prev = hb[j - 1]
    nw = _row(n)
    worst = 0.0
    for k in range(n + 1):
        v = 0.0
        if k == j:
            v = 1.0
        elif k == j - 1:
            v = EPS
        elif k + 2 <= j:
            t = hb[k + 1] * cur[k + 1] + (alpha[k] - a_last) * cur[k] - hb_prev * prev[k]
            if k > 0:
                t += hb[k] * cur[k - 1]
            t += (-1.0 if t < 0 else 1.0) * 2.0 * EPS * normA
            v = t / hbj
            worst = max(worst, abs(v))
        nw[k] = v
    return nw, worst


def simon_gates(alpha, hb):
    """-> list of n booleans: step j runs the sweep (engine 7 / the host-decided loop)"""
    n = len(alpha)
    W = [_row(n), _row(n), _row(n)]
    W[0][0] = 1.0  # omega_{0,0} = v_0 . v_0
    gates = [True]
    normA, force = 0.0, False
    for j in range(1, n):
        a_last, hbj = alpha[j - 1], hb[j]
        normA = max(normA, abs(a_last) + hb[j - 1] + hbj)
        nw, worst = _advance(W, alpha, hb, j, a_last, hbj, normA, n)
        due = worst > THRESH
        sweep = due or force  # a due sweep also covers the next vector
        force = due
        if sweep:
            for k in range(j):
                nw[k] = EPS
        W[j % 3] = nw
        gates.append(sweep)
    return gates


class LookaheadGate:
    """engine 8's decision, one step at a time (what `k_partial_onered_post` does right after the all-reduce of step j): feed
    ``step(j, a, hbj)`` with a = alpha_{j-1} and hbj = the norm that forms v_j; read ``gates[j + 1]``.  ``alpha`` / ``hb`` are the
    coefficient lists so far (the instance appends to them)."""

    def __init__(self, n, kappa=4.0):
        self.n, self.kappa = n, kappa
        self.W = [_row(n), _row(n), _row(n)]
        self.W[0][0] = 1.0
        self.gates = [True] + [False] * (n - 1)
        self.normA, self.st1, self.misses = 0.0, False, 0
        self.alpha, self.hb = [0.0] * n, [0.0] * n

    def step(self, j, a, hbj):
        n = self.n
        self.hb[j] = hbj
        if j == 0:
            return
        self.alpha[j - 1] = a
        g = self.gates[j]
        self.normA = max(self.normA, abs(a) + self.hb[j - 1] + hbj)
        nw, worst = _advance(self.W, self.alpha, self.hb, j, a, hbj, self.normA, n)
        if g:
            for k in range(j):
                nw[k] = EPS
        miss = (not g) and worst > THRESH
        self.misses += miss
        cur = self.W[(j + 2) % 3]
        wp = 0.0
        if j + 1 < n:
            for k in range(0, j):
                ak = a if k == j - 1 else self.alpha[k]
                hk1 = hbj if k + 1 == j else self.hb[k + 1]
                t = hk1 * nw[k + 1] + (ak - a) * nw[k] - hbj * cur[k]
                if k > 0:
                    t += self.hb[k] * nw[k - 1]
                t += (-1.0 if t < 0 else 1.0) * 2.0 * EPS * self.normA
                wp = max(wp, abs(t / hbj))
        self.W[j % 3] = nw
        due = self.kappa * wp > THRESH or miss
        gn = due or self.st1
        self.st1 = due and not g
        if j + 1 < n:
            self.gates[j + 1] = gn


def lookahead_gates(alpha, hb, kappa=4.0):
    """-> (gates, misses): engine 8's decisions - the gate of step j + 1 is taken right after the all-reduce of step j from the exact
    row j and a PREDICTED row j + 1 (unknown alpha_j ~ alpha_{j-1}, beta_{j+1} ~ beta_j); a vector whose exact omega exceeds sqrt(eps)
    although it was not swept is a miss (and forces the next two)."""
    n = len(alpha)
    lg = LookaheadGate(n, kappa)
    lg.step(0, 0.0, hb[0])
    for j in range(1, n):
        lg.step(j, alpha[j - 1], hb[j])
    return lg.gates, lg.misses


def warmup_norm(H, v0):
    """hb[0]: ||A v0 - (v0 . A v0) v0|| for the NORMALISED start vector (Lanczos.py:108-110)"""
    r = H @ v0
    a0 = float(v0 @ r)
    r = r - a0 * v0
    return math.sqrt(float(r @ r))
