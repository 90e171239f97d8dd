"""bench.py's output contract, checked without a GPU (missing #4):
 * the ONE JSON line with every BASELINE row present stays under the driver's 8 KB window, and each row keeps
   workload / value / ms_per_step / dtype / frac / kernel_ms / config.workload / cpu_baseline.value / cold;
 * `--gpus N` that disagrees with WORLD_SIZE is refused instead of printing a 1-GPU line;
 * `--gpus N` without a launcher starts N ranks itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* for each)."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _fake_row(name, bench):
    return {"workload": name, "label": "x" * 120, "value": 12345.678901234, "unit": "M samples/s", "dtype": "f32", "ms_per_step": 1.3680123456,
            "steps": 20, "warmup": 5, "preroll_steps": 44, "scaling": "weak",
            "config": {"workload": "RationalQuadraticSpline K=16 fwd+inv+logabsdetjac (BASELINE configs[2])", "dim": 32, "batch_per_gpu": 4194304},
            "roofline": {"bound": "hbm", "achieved": 6358.123456789, "peak": bench.HBM_PEAK_GBS, "unit": "GB/s", "frac": 0.79476543210, "traffic": 8598323200.0,
                         "traffic_source": "y" * 150, "kernel": "planar_reg2_kernel (forward + inverse launch)", "kernel_ms": 1.35134567,
                         "kernel_launches_per_step": 2.0, "stream_region_ms_per_step": 1.3612345, "algorithmic_bytes_per_launch": 8589934592,
                         "frac_of_measured_copy_ceiling_6290": 1.0108},
            "sum_logabsdetjac": -744261117.123, "cold": {"value": 11111.123456, "ms_per_step": 1.51234567, "kernel_ms": 1.4987654},
            "cpu_baseline": {"value": 12.3456789, "unit": "M samples/s", "cores": 1, "kind": "port", "sample": "z" * 200}, "us_per_call": 17.6123}


def test_the_line_fits_the_drivers_window_with_every_row():
    import bench

    a = argparse.Namespace(steps=20, warmup=5, scaling="weak", collective="torch", no_cache_params=False)
    names = ["c1", "c3", "c3_uncached", "c4", "c5a", "c5b", "c2_f64", "c4_f64"]
    head = _fake_row("c2", bench)
    cpu = dict(head["cpu_baseline"], variants={"fused_single_pass_1_core": {"value": 210.123456, "cores": 1}, "fused_single_pass_threads": {"value": 5210.123456, "cores": 64}})
    graph = [{"workload": w, "log2_batch_per_gpu": lb, "label": "q" * 100, "steps": 50, "eager_ms_per_step": 0.0176123, "graph_ms_per_step": 0.0140123,
              "speedup": 1.2567, "M_samples_per_s_graph": 1234.5, "sum_logabsdetjac_eager": 1.0, "sum_logabsdetjac_graph": 1.0} for w, lb in (("c1", None), ("c2", 20), ("c2", 16))]
    strong = [_fake_row("c2", bench), _fake_row("c4", bench)]
    line = bench.build_line(a, 8, head, [_fake_row(n, bench) for n in names], graph, strong, cpu)
    line["detail"] = "gpurun_out/bench_detail.json"
    text = json.dumps(line, separators=(",", ":"))
    assert len(text.encode()) < 7000, len(text.encode())        # the driver keeps an 8 KB tail
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "cold"):
        assert k in line, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    assert [r["workload"] for r in line["rows"]] == names
    for r in line["rows"]:
        for k in ("workload", "value", "ms_per_step", "dtype", "frac", "kernel_ms", "cold"):
            assert k in r, (r["workload"], k)
        assert r["config"]["workload"] and r["cpu_baseline"]["value"] is not None


def test_gpus_flag_that_disagrees_with_the_launcher_is_refused():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and p.stdout.strip() == "" and "WORLD_SIZE=1" in p.stderr


def test_gpus_n_without_a_launcher_starts_n_ranks(tmp_path, monkeypatch):
    """spawn_ranks: every rank gets its RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, rank 0's stdout is relayed, a failing rank fails the run."""
    import bench

    seen = []

    class FakeP:
        def __init__(self, argv, env=None, stdout=None):
            seen.append((argv, {k: env[k] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}, stdout))
            self.rank = int(env["RANK"])

        def communicate(self):
            return (b'{"n_gpus": 4}\n', None)

        def wait(self):
            return 0

    monkeypatch.setattr(subprocess, "Popen", FakeP)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    out = []
    monkeypatch.setattr(sys.stdout, "write", lambda t: out.append(t))
    bench.spawn_ranks(argparse.Namespace(gpus=4))
    assert [e["RANK"] for _, e, _ in seen] == ["0", "1", "2", "3"] and all(e["WORLD_SIZE"] == "4" and e["MASTER_ADDR"] == "127.0.0.1" for _, e, _ in seen)
    assert len({e["MASTER_PORT"] for _, e, _ in seen}) == 1 and [e["LOCAL_RANK"] for _, e, _ in seen] == ["0", "1", "2", "3"]
    assert all(argv[2:] == ["--gpus", "4", "--steps", "3"] for argv, _, _ in seen)
    assert "".join(out) == '{"n_gpus": 4}\n'
