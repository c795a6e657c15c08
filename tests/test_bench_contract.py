"""bench.py's bookkeeping, checked without a GPU: SURVEY section 8(d)'s per-node byte counts
(the figures DESIGN.md section 3 tabulates), the bytes the shipped instantiations need (what
``roofline.frac`` is computed from), the element counts of the scaling families, and the
launcher that starts one rank per GPU."""
import json
import os
import subprocess
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


def _hs_info(**kw):
    info = {"ns": 5, "naux": 17, "ngf": 9, "ngl": 4, "nhyp": 12, "Nq": 5, "Nqv": 5,
            "direction": 0, "diffusion_direction": 1, "gf_live": False, "law_gf": False,
            "nder": 2, "nupd_fused": 2, "q_read": [5, 0, 0, 5], "aux_read": [4, 0, 1, 6]}
    info.update(kw)
    return info


def test_needed_bytes_follow_the_instantiation():
    """Held-Suarez has zero viscosity: the nine gradient-flux columns are neither formed by
    k_gradients nor read by k_tendency<..., USE_GF = false>, so they are not counted; the face
    tables are the digested 36 B per face node (43.2 B per node at N = 4), not 67."""
    hs = _hs_info()
    F = 36.0 * 150 / 125
    # (of the 17 auxiliary columns the tendency pass reads 6: Phi, grad Phi, ref rho, ref p; the
    # gradient pass 4; the last hyperdiffusion pass only Delta and no state)
    assert abs(bench.needed_bytes_per_node(hs, "TENDENCY") - (8 * (5 + 6 + 12 + 2 + 9 + 2 + 15) + F)) < 1e-9
    assert abs(bench.needed_bytes_per_node(hs, "GRADIENTS") - (8 * (5 + 4 + 6 + 1 + 2 + 12) + F)) < 1e-9
    assert abs(bench.needed_bytes_per_node(hs, "DIVGRAD") - (8 * (12 + 2 + 6 + 4) + F)) < 1e-9
    assert abs(bench.needed_bytes_per_node(hs, "GRADLAP") - (8 * (4 + 0 + 1 + 6 + 1 + 12) + F)) < 1e-9
    # a law that declares nothing is charged every column
    full = _hs_info(q_read=None, aux_read=None, naux=17)
    assert abs(bench.needed_bytes_per_node(full, "TENDENCY") - bench.needed_bytes_per_node(hs, "TENDENCY") - 8 * 11) < 1e-9
    # the needed bytes never exceed SURVEY's generic count for this law, and keeping the gradient
    # flux alive (CMDG_OPT_KEEP_GRADFLUX, or a viscous law) puts the 72 B back
    law = types.SimpleNamespace(ns=5, naux=17, ngradflux=9, ngradlap=4, nhyper=12)
    for k in ("GRADIENTS", "GRADLAP", "TENDENCY"):
        assert bench.needed_bytes_per_node(hs, k) < bench.algorithmic_bytes_per_node(law, k)
    live = _hs_info(gf_live=True, law_gf=True)
    assert abs(bench.needed_bytes_per_node(live, "TENDENCY") - bench.needed_bytes_per_node(hs, "TENDENCY") - 72) < 1e-9
    assert abs(bench.needed_bytes_per_node(live, "GRADIENTS") - bench.needed_bytes_per_node(hs, "GRADIENTS") - 72) < 1e-9
    # the frac of round 2's profile on these bytes: no kernel above 1 (k_gradients read 1.01 on
    # the stale model), the headline kernel near 0.5
    nodes = 43200 * 125
    for k, us in (("TENDENCY", 709.8), ("GRADIENTS", 322.5)):
        frac = bench.needed_bytes_per_node(hs, k) * nodes / (us * 1e-6) / 8e12
        assert 0.3 < frac < 0.9, (k, frac)
    for k, us in (("DIVGRAD", 315.5), ("GRADLAP", 239.0)):          # profiles/r02_ab_*: us per launch
        frac = bench.needed_bytes_per_node(hs, k) * nodes / (us * 1e-6) / 8e12
        assert 0.3 < frac < 0.9, (k, frac)


def test_strong_family_is_the_baseline_sphere():
    assert bench.hs_nhorz("strong", 1) == bench.hs_nhorz("strong", 8) == 30
    assert bench.hs_nhorz("both", 8) == 30


def _bench(*argv, env=None):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(os.path.dirname(bench.__file__), "bench.py")]
                          + list(argv), env=e, capture_output=True, text=True, timeout=120)


def test_launcher_starts_one_rank_per_gpu_without_touching_the_gpu():
    """``python bench.py --gpus 3`` with no RANK in the environment starts three children with
    the environment torch.distributed.run would give them and relays rank 0's one line."""
    r = _bench("--gpus", "3", "--dry-launch")
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1                                  # ONE JSON line on stdout
    rec = json.loads(lines[0])["dry_launch"]
    assert rec["rank"] == 0 and rec["world"] == 3 and rec["gpus"] == 3
    assert rec["master"].startswith("127.0.0.1:") and not rec["torch_imported"]
    others = [json.loads(ln.split("] ", 1)[1]) for ln in r.stderr.splitlines() if ln.startswith("[dry-launch]")]
    assert sorted(o["rank"] for o in others) == [1, 2]
    assert all(o["master"] == rec["master"] and o["local_rank"] == o["rank"] for o in others)


def test_launcher_fails_when_a_rank_fails():
    r = _bench("--gpus", "2", "--dry-launch", env={"BENCH_DRY_FAIL_RANK": "1"})
    assert r.returncode == 3
    # launched as a rank already (torch.distributed.run): no second generation of children
    r = _bench("--gpus", "2", "--dry-launch", env={"RANK": "1", "WORLD_SIZE": "2", "LOCAL_RANK": "1",
                                                   "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "1"})
    assert r.returncode == 0 and r.stdout.strip() == "" and "[dry-launch]" in r.stderr


def test_result_line_survives_native_output_on_stdout():
    """RCCL prints warnings to file descriptor 1; a rank keeps the real stdout for its one JSON
    line and sends everything else written to descriptor 1 to stderr (bench.claim_stdout)."""
    code = ("import os, sys; sys.path.insert(0, %r); import bench; bench.claim_stdout(); "
            "os.write(1, b'NCCL WARN something native\\n'); print('python chatter'); "
            "bench.emit({'metric': 'x', 'value': 1})" % os.path.dirname(bench.__file__))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip().splitlines() == ['{"metric": "x", "value": 1}']
    assert "NCCL WARN something native" in r.stderr and "python chatter" in r.stderr



def test_cpu_baseline_reports_its_threads_and_memory_ceiling(monkeypatch):
    """The all-core figure says how many threads it ran on (never more than the affinity mask or
    the cgroup quota allow), how they were placed, and the streaming bandwidth they reach together
    (SURVEY 8(d)); its arrays are first-touched by the threads that work on them."""
    from cmdg_loader import cm
    info = bench.host_cpu_info()
    assert info["affinity_threads"] >= 1
    assert bench.baseline_threads({"affinity_threads": 128, "cgroup_cpu_quota": 16.0}, 128) == 16
    assert bench.baseline_threads({"affinity_threads": 8, "cgroup_cpu_quota": None}, 128) == 8
    args = types.SimpleNamespace(workload="advdiff-brick", bomex_ne=16, nvert=8)
    law, grid, direction, dt, _ = bench.build_workload(cm, "advdiff-brick", 0, 1, 3, args)
    out = bench.cpu_baseline(cm, law, grid, direction, dt, 0.2, args)
    assert out["kind"] == "port" and out["value"] > 0
    assert out["cores"] == out["threads"] == bench.baseline_threads(info, 10 ** 6) or out["threads"] <= info["affinity_threads"]
    assert out["omp"] == {"OMP_PROC_BIND": os.environ.get("OMP_PROC_BIND"), "OMP_PLACES": os.environ.get("OMP_PLACES")}
    assert out["omp"]["OMP_PROC_BIND"] is not None      # bench.py sets them before libgomp loads
    assert out["stream_triad_GBs"] > 0 and "first_touch" in out and out["host"] == info
