"""Stand-in for fjcommon.functools_ext (API surface used by the reference only)."""
import functools
import itertools


def return_list(f):
    @functools.wraps(f)
    def wrapper(*a, **kw):
        return list(f(*a, **kw))
    return wrapper


def return_tuple(f):
    @functools.wraps(f)
    def wrapper(*a, **kw):
        return tuple(f(*a, **kw))
    return wrapper


def lconcat(it):
    return list(itertools.chain.from_iterable(it))


def unzip(it):
    return zip(*it)


def identity(x):
    return x


def compose(*fs):
    def composed(x):
        for f in reversed(fs):
            x = f(x)
        return x
    return composed
