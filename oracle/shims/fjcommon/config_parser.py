"""Stand-in for fjcommon.config_parser: `key = <python expr>`, dotted keys -> nested attrs,
`use <file>` include (relative to the including file), `#` comments, `constrain k :: a, b`."""
import os


class _Config(object):
    def __init__(self):
        self._keys = []

    def set_attr(self, key, value):
        parts = key.split('.')
        obj = self
        for p in parts[:-1]:
            if not hasattr(obj, p):
                setattr(obj, p, _Config())
                obj._keys.append(p) if p not in obj._keys else None
            obj = getattr(obj, p)
        setattr(obj, parts[-1], value)
        if parts[-1] not in obj._keys:
            obj._keys.append(parts[-1])

    def all_params_and_values(self, prefix=''):
        for k in self._keys:
            v = getattr(self, k)
            if isinstance(v, _Config):
                yield from v.all_params_and_values(prefix + k + '.')
            else:
                yield prefix + k, v

    def __str__(self):
        return '\n'.join('{} = {!r}'.format(k, v) for k, v in self.all_params_and_values())


def _parse_into(config, path):
    with open(path) as f:
        for line in f:
            line = line.split('#', 1)[0].strip()
            if not line:
                continue
            if line.startswith('use '):
                _parse_into(config, os.path.join(os.path.dirname(path), line[4:].strip()))
                continue
            if line.startswith('constrain '):
                continue
            key, expr = line.split('=', 1)
            config.set_attr(key.strip(), eval(expr.strip(), {}, {}))


def parse(path):
    config = _Config()
    _parse_into(config, path)
    return config, os.path.basename(path)
