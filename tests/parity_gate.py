"""The parity gate on P (the mixture parameters the network predicts), shared by the -m gpu network tests.

north_star: "CDF/logits within 1e-5 fp32"; SURVEY.md section 8c P2: "within 1e-5 (abs, fp32)".  P's channels come in parameter
groups (index = p * C * K + c * K + k, logistic_mixture.py:228-232): logit pi, mu, log sigma and -- on the RGB scale -- lambda.
Every group is gated on its own:

    max |P - P_ref|  <  1e-5 * max(1, max |P_ref| / 4)

i.e. the ABSOLUTE 1e-5 for values up to 4 and 2.5e-6 RELATIVE above -- about 21 ulp of the group's largest value in every binade.
Why not a flat absolute 1e-5 for everything below 16 (the round-3 verdict's proposal): it is 10.5 ulp of a log sigma near 9, and
the MI355X path measures 10 ulp there at 768x512 (9.5e-6) and 12 ulp (1.14e-5) on the 64x96 case of tests/test_gpu_net.py -- two
correct fp32 evaluations of a 192-term dot product (the classifier's last 1x1 layer) in different summation orders differ by
that much; a gate AT the noise floor fails on the next image.  This form still trips on a 2x regression of any group (round 3
gated the whole tensor at 1e-5 x its largest |value|: 2.8e-3 absolute for the log-sigma channels next to a 280-valued mean -- a
100x regression would have passed; that form is gone from every network test and from bench.py's parity leg).

Round 5: THE TRUTH GATE.  The gate above compares two fp32 evaluations with each other and cannot say which of them is off.
tools/parity_truth.py (-> profiles/r05_parity_truth.json) runs the oracle's own decoder chain in fp64 on the same weights and
bottlenecks: at 768x512 on the calibrated checkpoint the MI355X path is within 7.2e-6 of fp64 in every group whose values stay
below 16 (log sigma: 3.3e-6) while the torch-CPU fp32 oracle is 9.7e-6 off on log sigma -- the 9.5e-6 between the two that sat on
the old budget is the ORACLE's rounding (it sums the classifier's 192-term dot products in one fp32 chain; the MFMA kernel in
blocked partial sums).  So the accuracy claim is gated against fp64 and FLAT, as north_star states it:

    max |P - P_fp64|  <  1e-5                       for every group whose largest |value| is <= 16
                      <  1e-6 x largest |value|     above (the RGB means, up to 265: 8 ulp; an absolute 1e-5 is below one ulp there)

`truth_chain` computes P_fp64 (a few seconds at 768x512 on the GPU box's host), `assert_P_truth` applies the gate; the fp32-vs-fp32
form above stays as the consistency check against the reference-generated fixtures (whose P is the reference's fp32)."""
import numpy as np

TOL_ABS = 1e-5
ABS_UP_TO = 4.0
GROUPS = ('logit_pi', 'mu', 'log_sigma', 'lambda')


def group_errs(P, ref, num_params):
    """P, ref: (N, Kp, ...) arrays / tensors -> {group: {'max_abs', 'max_rel', 'max_value'}}"""
    P = np.asarray(P.cpu() if hasattr(P, 'cpu') else P, dtype=np.float64)
    ref = np.asarray(ref.cpu() if hasattr(ref, 'cpu') else ref, dtype=np.float64)
    assert P.shape == ref.shape and P.shape[1] % num_params == 0, (P.shape, ref.shape, num_params)
    n = P.shape[1] // num_params
    out = {}
    for p, name in enumerate(GROUPS[:num_params]):
        d = np.abs(P[:, p * n:(p + 1) * n] - ref[:, p * n:(p + 1) * n]).max()
        m = np.abs(ref[:, p * n:(p + 1) * n]).max()
        out[name] = {'max_abs': float(d), 'max_rel': float(d / m) if m > 0 else float(d), 'max_value': float(m)}
    return out


def group_tol(max_value):
    return TOL_ABS * max(1.0, max_value / ABS_UP_TO)


def group_ok(g):
    return g['max_abs'] < group_tol(g['max_value'])


def assert_P(P, ref, scale, what=''):
    """scale 0 is the RGB scale (4 parameter groups), the bottleneck scales have 3."""
    groups = group_errs(P, ref, 4 if scale == 0 else 3)
    for name, g in groups.items():
        assert group_ok(g), (what, scale, name, g)
    return groups


def feature_tol(ref):
    """features F (no parameter groups; |F| stays below ~3 on both checkpoints): absolute 1e-5, relative above magnitude 1"""
    return TOL_ABS * max(float(np.abs(np.asarray(ref.cpu() if hasattr(ref, 'cpu') else ref)).max()), 1.0)


# ---- the truth gate (round 5) --------------------------------------------------------------------------------------------------

TRUTH_FLAT_UP_TO = 16.0
TRUTH_REL_ABOVE = 1e-6


def truth_chain(bn_q, sd):
    """P of every scale in fp64: the oracle's get_P (oracle/net.py; reference multiscale_network.py:308-322) coarse to fine on the
    given bottlenecks bn_q[1..3] (exact quantiser levels in any precision) with the weights promoted to double.  -> [P0, P1, P2]"""
    import torch
    from oracle import net as onet
    sd64 = {k: v.double() for k, v in sd.items()}
    P, f = [None] * 3, None
    with torch.no_grad():
        for s in (2, 1, 0):
            b = bn_q[s + 1]
            b = b if hasattr(b, 'double') else torch.from_numpy(np.asarray(b))
            P[s], f = onet.get_P(s, b.double().cpu(), f, sd64)
    return P


def truth_tol(max_value):
    return TOL_ABS if max_value <= TRUTH_FLAT_UP_TO else TRUTH_REL_ABOVE * max_value


def truth_errs(P, P64, scale):
    groups = group_errs(P, P64, 4 if scale == 0 else 3)
    for g in groups.values():
        g['tolerance'] = truth_tol(g['max_value'])
        g['ok'] = bool(g['max_abs'] < g['tolerance'])
    return groups


def assert_P_truth(P, P64, scale, what=''):
    groups = truth_errs(P, P64, scale)
    for name, g in groups.items():
        assert g['ok'], ('vs fp64', what, scale, name, g)
    return groups
