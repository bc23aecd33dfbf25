"""`.partN` file naming for auto-cropped images (reference bitcoding/part_suffix_helper.py:10-35)."""
import glob
import os
import re

_SUFFIX = re.compile(r'\.part(\d+)$')


def make_part_suffix(i):
    assert i >= 0, i
    return '.part{}'.format(i)


def contains_part_suffix(p):
    return _SUFFIX.search(p) is not None


def index_of_part_suffix(p):
    return int(_SUFFIX.search(p).group(1))


def iter_part_suffixes(pin):
    """All sibling part files of `pin`, sorted by part index."""
    assert os.path.isfile(pin) and contains_part_suffix(pin)
    base = pin[:_SUFFIX.search(pin).start()] + '.part'
    matches = [m for m in glob.glob(glob.escape(base) + '*') if contains_part_suffix(m)]
    return sorted(matches, key=index_of_part_suffix)
