"""Parser for the reference's `.cf` experiment configs.

The reference delegates to the third-party `fjcommon.config_parser` (pip_requirements.txt:4, pinned 0.2.10; not
vendored in the reference).  Call sites that define the needed behaviour: multiscale_tester.py:183
(`config_parser.parse(path) -> (config, rel_path)`), global_config.py:83-89 (`all_params_and_values()`,
`set_attr(k, v)`), and the files themselves (configs/ms/cr.cf, cr_rgb_shared.cf, cr_rgb.cf):

    key = <python literal / expression>        dotted keys build nested namespaces (`q.L = 25` -> cfg.q.L)
    use <other.cf>                              include, path relative to the including file; later keys override
    # comment                                  (also trailing)
"""
import ast
import os


class Config(object):
    """Attribute namespace with the small API the reference uses on fjcommon configs."""

    def __init__(self):
        object.__setattr__(self, '_order', [])

    def set_attr(self, key, value):
        head, _, rest = key.partition('.')
        if rest:
            child = self.__dict__.get(head)
            if not isinstance(child, Config):
                child = Config()
                self._put(head, child)
            child.set_attr(rest, value)
        else:
            self._put(head, value)

    def _put(self, k, v):
        if k not in self._order:
            self._order.append(k)
        self.__dict__[k] = v

    def __setattr__(self, k, v):
        self._put(k, v)

    def all_params_and_values(self, _prefix=''):
        for k in self._order:
            v = self.__dict__[k]
            if isinstance(v, Config):
                for kv in v.all_params_and_values(_prefix + k + '.'):
                    yield kv
            else:
                yield _prefix + k, v

    def get(self, key, default=None):
        node = self
        for part in key.split('.'):
            if not isinstance(node, Config) or part not in node.__dict__:
                return default
            node = node.__dict__[part]
        return node

    def __repr__(self):
        return 'Config({})'.format(', '.join('{}={!r}'.format(k, v) for k, v in self.all_params_and_values()))


def _eval_value(text):
    try:
        return ast.literal_eval(text)
    except (ValueError, SyntaxError):
        return eval(text, {'__builtins__': {}}, {})


def _parse_file(path, config, seen):
    path = os.path.abspath(path)
    if path in seen:
        raise ValueError('circular `use` in config: {}'.format(path))
    seen = seen | {path}
    with open(path) as f:
        for lineno, raw in enumerate(f, 1):
            line = raw.split('#', 1)[0].strip()
            if not line:
                continue
            if line.startswith('use '):
                _parse_file(os.path.join(os.path.dirname(path), line[4:].strip()), config, seen)
                continue
            if '=' not in line:
                raise ValueError('{}:{}: expected `key = value`, got {!r}'.format(path, lineno, raw.rstrip()))
            key, value = line.split('=', 1)
            config.set_attr(key.strip(), _eval_value(value.strip()))


def parse(path):
    """-> (Config, basename) like fjcommon.config_parser.parse."""
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    config = Config()
    _parse_file(path, config, frozenset())
    return config, os.path.basename(path)


CONFIG_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'configs')


def parse_builtin(kind, name):
    """Parse one of the configs shipped with this package, e.g. parse_builtin('ms', 'cr')."""
    return parse(os.path.join(CONFIG_DIR, kind, name if name.endswith('.cf') else name + '.cf'))[0]
