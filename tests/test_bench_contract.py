"""bench.py's bookkeeping, checked without a GPU: SURVEY section 8(d)'s per-node byte counts
(the figures DESIGN.md section 3 tabulates) and the element counts of the weak-scaling family."""
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def test_algorithmic_bytes_per_node_held_suarez_and_config1():
    hs = types.SimpleNamespace(ns=5, naux=17, ngradflux=9, ngradlap=4, nhyper=12)
    assert bench.algorithmic_bytes_per_node(hs, "GRADIENTS") == 483
    assert bench.algorithmic_bytes_per_node(hs, "DIVGRAD") == 283
    assert bench.algorithmic_bytes_per_node(hs, "GRADLAP") == 443
    assert bench.algorithmic_bytes_per_node(hs, "TENDENCY") == 619
    ad = types.SimpleNamespace(ns=1, naux=15, ngradflux=3, ngradlap=0, nhyper=0)
    assert bench.algorithmic_bytes_per_node(ad, "GRADIENTS") == 291
    assert bench.algorithmic_bytes_per_node(ad, "TENDENCY") == 331
    # N = 6: the face tables cost 336 / 7 = 48 B per node
    mo = types.SimpleNamespace(ns=6, naux=19, ngradflux=13, ngradlap=0, nhyper=0)
    assert bench.algorithmic_bytes_per_node(mo, "TENDENCY", 7) == 584


def test_weak_family_keeps_the_baseline_sphere_per_gpu():
    per_gpu = {n: 6 * bench.hs_nhorz("weak", n) ** 2 * 8 / n for n in (1, 2, 4, 8)}
    assert per_gpu[1] == 43200
    assert all(abs(v - 43200) / 43200 < 0.025 for v in per_gpu.values()), per_gpu
    assert bench.hs_nhorz("strong", 8) == 30
    assert 6 * bench.hs_nhorz("weak-small", 8) ** 2 * 8 / 8 == 5400
