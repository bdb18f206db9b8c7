"""Small host utilities with the reference's names (utils.py:12-49, :126-151) so that scripts
written against the reference find them here; the graph builders are thin aliases of graph.py."""
import os
import random
import sys

import numpy as np
import torch

from .graph import adjacency_host, graph_rank_nodes  # noqa: F401  (re-exported)
from .trainer import AverageMeter  # noqa: F401


def set_seed(seed=0):
    """utils.py:12-20."""
    random.seed(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


class Unbuffered:
    """File-like wrapper that flushes after every write, so a redirected log shows progress live
    (the reference's utils.Unbuffered, utils.py:138-151)."""

    def __init__(self, stream):
        self._target = stream

    def write(self, text):
        n = self._target.write(text)
        self._target.flush()
        return n

    def writelines(self, lines):
        self._target.writelines(lines)
        self._target.flush()

    def __getattr__(self, name):          # everything else (fileno, close, ...) goes to the wrapped stream
        return getattr(self._target, name)


def init_run(log_path, seed):
    """utils.py:23-29: seed everything and redirect stdout / stderr to <log_path>/log.txt."""
    set_seed(seed)
    os.makedirs(log_path, exist_ok=True)
    f = Unbuffered(open(os.path.join(log_path, 'log.txt'), 'w'))
    sys.stderr = f
    sys.stdout = f


def generate_daj_mat(dataset):
    """utils.py:41-49 as host CSR arrays (rowptr, col, val)."""
    return adjacency_host(dataset.train_array, dataset.n_users, dataset.n_items)
