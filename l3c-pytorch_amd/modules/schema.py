"""State-dict schema of `MultiscaleNetwork` (key -> shape, in the reference's registration order).

The released checkpoints of the reference are `torch.save({'net': net.state_dict(), ...})` (helpers/saver.py:168);
this module enumerates exactly the keys that `MultiscaleNetwork(config_ms).state_dict()` of the reference produces
(modules/multiscale_network.py:168-218, modules/net.py:89-184, modules/edsr.py, modules/head.py, modules/prob_clf.py),
so that `load_state_dict(strict=True)` semantics (multiscale_tester.py:199) can be enforced without an nn.Module tree.
Note the quantiser levels are registered twice (`enc.levels` and `enc.q.levels`, net.py:125-127).
"""
from collections import OrderedDict


def non_shared_get_Kp(K, C):
    """Number of channels the probability classifier predicts (criterion/logistic_mixture.py:68-73)."""
    return (4 if C == 3 else 3) * C * K


def _conv(d, key, cout, cin, k):
    d[key + '.weight'] = (cout, cin, k, k)
    d[key + '.bias'] = (cout,)


def _body(d, prefix, Cf, k, num_blocks):
    for i in range(num_blocks):
        _conv(d, '{}.{}.body.0'.format(prefix, i), Cf, Cf, k)
        _conv(d, '{}.{}.body.2'.format(prefix, i), Cf, Cf, k)
    _conv(d, '{}.{}'.format(prefix, num_blocks), Cf, Cf, k)


def param_schema(config_ms):
    Cf, k = config_ms.Cf, config_ms.kernel_size
    C, L, K = config_ms.q.C, config_ms.q.L, config_ms.prob.K
    rgb = config_ms.rgb_bicubic_baseline
    d = OrderedDict()
    _conv(d, 'sub_rgb_mean', 3, 3, 1)
    if not rgb:
        _conv(d, 'heads.0.head.0', 3, 3, 1)
        _conv(d, 'heads.0.head.1.head', Cf, 3, k)
        cin = Cf if config_ms.enc.feed_F else C
        for s in range(1, config_ms.num_scales):
            _conv(d, 'heads.{}.head'.format(s), Cf, cin, k)
    for s in range(config_ms.num_scales):
        p = 'nets.{}'.format(s)
        if config_ms.enc.cls == 'EDSRLikeEnc':
            d[p + '.enc.levels'] = (L,)
            _conv(d, p + '.enc.down', Cf, Cf, 5)
            _body(d, p + '.enc.body', Cf, k, config_ms.enc.num_blocks)
            _conv(d, p + '.enc.to_q.0', C, Cf, 1)
            d[p + '.enc.q.levels'] = (L,)
        _conv(d, p + '.dec.head', Cf, C, 1)
        _body(d, p + '.dec.body', Cf, k, config_ms.dec.num_blocks)
        _conv(d, p + '.dec.tail.0', 4 * Cf, Cf, 3)
    for s in range(config_ms.num_scales):
        p = 'prob_clfs.{}.atrous'.format(s)
        for i in range(3):
            _conv(d, '{}.atrous.{}'.format(p, i), Cf, Cf, k)
        _conv(d, p + '.lin', non_shared_get_Kp(K, 3 if (rgb or s == 0) else C), 3 * Cf, 1)
    return d


def check_state_dict(state_dict, config_ms):
    """strict=True check: raises RuntimeError listing missing / unexpected keys and shape mismatches."""
    schema = param_schema(config_ms)
    missing = [k for k in schema if k not in state_dict]
    unexpected = [k for k in state_dict if k not in schema]
    wrong = ['{}: {} != {}'.format(k, tuple(state_dict[k].shape), schema[k]) for k in schema
             if k in state_dict and tuple(state_dict[k].shape) != schema[k]]
    if missing or unexpected or wrong:
        raise RuntimeError('Error(s) in loading state_dict for MultiscaleNetwork:\n'
                           '  Missing key(s): {}\n  Unexpected key(s): {}\n  Size mismatch: {}'.format(
                               missing, unexpected, wrong))
    return schema
