"""Randomised parity of fs_score_fim against the oracle: clouds from one landmark to 400 k, dense clumps (crowded voxels,
multi-pass and HBM-tier poses), ranges from 1 m to 400 m, every cone mode, tables with and without holes — a fixed-seed slice
of tests/fim_random.py (which runs hundreds of trials by hand).  Integers exactly, FI to 1e-4 (observed <= 2e-7)."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_fim_random_configurations_match_the_oracle(monkeypatch, capsys):
    spec = importlib.util.spec_from_file_location("fim_random", os.path.join(ROOT, "tests", "fim_random.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(sys, "argv", ["fim_random.py", "30", "5"])
    mod.main()                                              # asserts inside; prints one line per trial
    out = capsys.readouterr().out
    assert "30 trials passed" in out
