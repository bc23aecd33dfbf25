"""Experiment-directory conventions of the reference (helpers/paths.py:39-62, helpers/logdir_helpers.py:71-108,
helpers/saver.py:33-110, :188-210), inference side only.

An experiment lives in  LOG_DIR/"<MMDD_HHMM> <ms-config> <dl-config> [r@...] [postfix ...]"/ckpts/ckpt_XXXXXXXXXX.pt  and a
checkpoint is `torch.save({'net': state_dict, 'optim': ...})`.  The config names in the directory name select `.cf` files
under the config dir ('@' stands for a path separator)."""
import glob
import os
import re
from collections import namedtuple

import torch

CKPTS_DIR_NAME = 'ckpts'
_LOG_DATE = re.compile(r'^\d{4}_\d{4}$')
_CKPT = re.compile(r'ckpt_(\d+)\.pt(\.tmp)?$')

LogDirComps = namedtuple('LogDirComps', ['config_paths', 'postfix'])


def is_log_date(s):
    return _LOG_DATE.match(s) is not None


def get_experiment_dir(log_dir, experiment_spec):
    """`experiment_spec` is a log date (unique prefix of a sub-directory of log_dir) or a sub-directory name."""
    if is_log_date(experiment_spec):
        if log_dir is None:
            raise ValueError('Can only infer experiment_dir from log_date if log_dir is not None')
        matches = glob.glob(os.path.join(glob.escape(log_dir), experiment_spec + '*'))
        if len(matches) != 1:
            raise ValueError('Expected one match for {}*, got {}'.format(os.path.join(log_dir, experiment_spec), matches))
        experiment_dir = matches[0]
    else:
        experiment_dir = os.path.join(log_dir, experiment_spec)
    experiment_dir = experiment_dir.rstrip(os.path.sep)
    if not os.path.isdir(experiment_dir):
        raise ValueError('Invalid experiment_dir: {}'.format(experiment_dir))
    return experiment_dir


def get_ckpts_dir(experiment_dir):
    p = os.path.join(experiment_dir, CKPTS_DIR_NAME)
    if not os.path.isdir(p):
        raise ValueError('Not found: {}'.format(p))
    return p


def parse_log_dir(log_dir, configs_dir, base_dirs=('ms', 'dl'), append_ext='.cf'):
    """-> LogDirComps(paths of the configs named by the directory, postfix tuple or None)."""
    name = os.path.basename(log_dir.strip(os.path.sep))
    comps = name.split(' ')
    if not is_log_date(comps[0]):
        raise ValueError('Invalid log_dir: {}'.format(log_dir))
    if len(comps) <= len(base_dirs):
        raise ValueError('Expected a config for each of {}, got {}'.format(base_dirs, comps))
    names = comps[1:1 + len(base_dirs)]
    has_restore = any('r@' in c for c in comps)
    postfix = comps[1 + len(base_dirs) + has_restore:]
    paths = []
    for base, n in zip(base_dirs, names):
        p = os.path.join(configs_dir, base, n.replace('@', os.path.sep)) + append_ext
        if not os.path.isfile(p):
            raise ValueError('Cannot find config on disk: {}'.format(p))
        paths.append(p)
    return LogDirComps(tuple(paths), tuple(postfix) if postfix else None)


def list_checkpoints(ckpts_dir):
    """[(iteration, path)] sorted by iteration (persistent `.pt` and temporary `.pt.tmp` alike, saver.py:52-76)."""
    out = []
    for p in glob.glob(os.path.join(glob.escape(ckpts_dir), 'ckpt_*')):
        m = _CKPT.search(os.path.basename(p))
        if m:
            out.append((int(m.group(1)), p))
    return sorted(out)


def get_ckpt_for_itr(ckpts_dir, itr):
    """itr == -1: latest.  Otherwise the first checkpoint with iteration >= itr (saver.py:78-95)."""
    ckpts = list_checkpoints(ckpts_dir)
    if not ckpts:
        raise ValueError('No checkpoints in {}'.format(ckpts_dir))
    if itr == -1:
        return ckpts[-1]
    for i, p in ckpts:
        if i >= itr:
            return i, p
    raise ValueError('No checkpoint with iteration >= {} in {}'.format(itr, ckpts_dir))


def restore(modules, ckpt_p, strict=True):
    """Restorer.restore (saver.py:188-210): load `{'net': state_dict, ...}` into `modules['net']`, strictly."""
    print('Restoring {}... (strict={})'.format(ckpt_p, strict))
    state_dicts = torch.load(ckpt_p, map_location='cpu')
    for key, m in modules.items():
        m.load_state_dict(state_dicts[key], strict=strict)
    return int(_CKPT.search(os.path.basename(ckpt_p)).group(1))
