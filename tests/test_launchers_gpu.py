"""The launchers README advertises, each run as a FRESH child process on the GPU box: run/run.py (the five steps of
the reference's run/run.py:10-26) and run/inductive.py (run/dropui/igcn_dropui.py:26-35, run/dropit/igcn_dropit.py:26-37)."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SECTION = re.compile(r'(Precision|Recall|NDCG): ((?:[^%\s]+%@\d+, )+)')         # "Recall: 2.345%@20, " (trainer.py:170-176)
ENTRY = re.compile(r'([^%\s]+)%@(\d+), ')


def _run(script, *args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'run', script)] + list(args), cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    return p.stdout.decode()


def _metrics(line):
    vals = {(m, int(k)): float(v) for m, body in SECTION.findall(line) for v, k in ENTRY.findall(body)}
    assert {m for m, _ in vals} == {'Precision', 'Recall', 'NDCG'}, line
    for v in vals.values():
        assert v == v and 0.0 <= v <= 100.0, line                                 # finite, a percentage
    return vals


@pytest.mark.parametrize('index,name', [(1, 'LightGCN'), (0, 'MF')])
def test_run_py_trains_and_tests_on_the_synthetic_gowalla_split(index, name):
    out = _run('run.py', '--synthetic', '--dataset', 'gowalla', '--index', str(index), '--epochs', '1')
    lines = [l for l in out.splitlines() if l.startswith('Test result.')]
    assert len(lines) == 1, out[-2000:]
    vals = _metrics(lines[0])
    assert ('Recall', 20) in vals and ('NDCG', 20) in vals
    assert vals[('Recall', 20)] > 0.0                                             # one epoch of BPR on a popularity-skewed split learns something
    assert 'Epoch' in out or 'epoch' in out                                       # the training loop reported its epoch


def test_inductive_py_dropui_prints_the_six_masked_evaluations():
    out = _run('inductive.py', '--scenario', 'dropui', '--dataset', 'gowalla', '--index', '2', '--epochs', '1')
    assert 'Inductive results.' in out
    after = out[out.index('Inductive results.'):]
    lines = [l for l in after.splitlines() if SECTION.search(l)]
    assert len(lines) == 6, after[-3000:]                                          # trainer.py:179-219: six masked eval('test') calls
    for l in lines:
        _metrics(l)


def test_inductive_py_dropit_evaluates_before_and_after_the_update():
    out = _run('inductive.py', '--scenario', 'dropit', '--dataset', 'gowalla', '--index', '2', '--epochs', '1')
    prev = [l for l in out.splitlines() if l.startswith('Previous interactions test result.')]
    upd = [l for l in out.splitlines() if l.startswith('Updated interactions test result.')]
    assert len(prev) == 1 and len(upd) == 1, out[-2000:]
    a, b = _metrics(prev[0]), _metrics(upd[0])
    assert a.keys() == b.keys()
