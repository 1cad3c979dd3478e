"""Fixed-seed slices of the scripts that are otherwise run by hand for thousands of trials (tests/fused_random.py, rank_random.py,
adversarial_probe.py, big_list.py) — so that each of them is exercised, small, by every `-m gpu` run.  Same pattern as
test_gpu_fim_random.py: the script's own main() with a short argv, in this process (one GPU process, no child runners)."""
import importlib.util
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(name, argv, monkeypatch):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tests", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(sys, "argv", [name + ".py"] + argv)
    return mod.main()


def test_fused_random_slice(monkeypatch, capsys):
    _run("fused_random", ["40", "11"], monkeypatch)
    assert "40 trials passed" in capsys.readouterr().out


def test_rank_random_slice(monkeypatch, capsys):
    _run("rank_random", ["150", "11"], monkeypatch)
    assert "150 trials passed" in capsys.readouterr().out


def test_adversarial_probe(monkeypatch, capsys):
    assert _run("adversarial_probe", [], monkeypatch) == 0
    assert "problems: []" in capsys.readouterr().out
