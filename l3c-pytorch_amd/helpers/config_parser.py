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


_ARITH = (ast.Add, ast.Sub, ast.Mult, ast.Div, ast.FloorDiv, ast.Pow, ast.Mod, ast.USub, ast.UAdd)


def _arith(node):
    """Numbers combined with + - * / // % ** and parentheses (and tuples / lists of such): what the reference's .cf files use
    beyond plain literals (e.g. `2 ** 16`, `(-1, 1)`).  Anything else -- names, calls, attribute access -- is refused."""
    if isinstance(node, ast.Constant):
        return node.value
    if isinstance(node, ast.Tuple):
        return tuple(_arith(e) for e in node.elts)
    if isinstance(node, ast.List):
        return [_arith(e) for e in node.elts]
    if isinstance(node, ast.UnaryOp) and isinstance(node.op, _ARITH):
        v = _arith(node.operand)
        return -v if isinstance(node.op, ast.USub) else +v
    if isinstance(node, ast.BinOp) and isinstance(node.op, _ARITH):
        a, b = _arith(node.left), _arith(node.right)
        if not all(isinstance(x, (int, float)) and not isinstance(x, bool) for x in (a, b)):
            raise ValueError('arithmetic on non-numbers')
        if isinstance(node.op, ast.Pow) and abs(b) > 64:
            raise ValueError('exponent too large')
        return {ast.Add: lambda: a + b, ast.Sub: lambda: a - b, ast.Mult: lambda: a * b, ast.Div: lambda: a / b,
                ast.FloorDiv: lambda: a // b, ast.Mod: lambda: a % b, ast.Pow: lambda: a ** b}[type(node.op)]()
    raise ValueError('unsupported expression')


def _eval_value(text):
    """Literal (ast.literal_eval) or plain arithmetic on literals; never `eval` -- the tester also feeds `key=value` tokens taken
    from the experiment directory's NAME through this function."""
    try:
        return ast.literal_eval(text)
    except (ValueError, SyntaxError):
        pass
    try:
        return _arith(ast.parse(text, mode='eval').body)
    except (ValueError, SyntaxError, TypeError, ZeroDivisionError, KeyError) as e:
        raise ValueError('config value {!r}: only literals and arithmetic on literals are allowed ({})'.format(text, e))


def _strip_comment(line):
    """Cut at the first '#' that is not inside a quoted string."""
    quote = None
    for i, ch in enumerate(line):
        if quote:
            if ch == quote and line[i - 1] != '\\':
                quote = None
        elif ch in '\'"':
            quote = ch
        elif ch == '#':
            return line[:i]
    return line


def _parse_file(path, config, seen):
    path = os.path.abspath(path)
    if path in seen:
        raise ValueError('circular `use` in config: {}'.format(path))
    seen = seen | {path}
    with open(path) as f:
        for lineno, raw in enumerate(f, 1):
            line = _strip_comment(raw).strip()
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
