"""bench.py end to end on the GPU box: one JSON line with the fields the driver reads, for the plain run and for the
exchange loop (one-rank RCCL group), at a small row count so that both finish in seconds."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _run(*flags):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "6", "--warmup", "2", "--rows", "4096", *flags],
                       capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_json_line():
    d = _run()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["vs_baseline"] is None
    assert d["value"] > 0 and d["higher_is_better"] is True and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert r["alone"]["kernel_ms"] > 0 and r["step"]["frac"] > 0
    assert r["kernel_ms"] == r["alone"]["kernel_ms"], "frac must come from the one-stream leg, not from overlapped launches"
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0
    assert d["cpu_baseline_all_cores"]["cores"] >= 1 and d["cpu_baseline_all_cores"]["value"] > 0
    assert d["stress"]["uniform_text"]["value"] > 0 and d["stress"]["no_memo"]["ms_per_step"] > 0
    assert d["stress"]["fixed_memo_only"]["ms_per_step"] > 0 and d["stress"]["mixed_script_text"]["value"] > 0
    memo = d["config"]["piece_memo"]
    # (memo_learn = 0, the library's default: the first level learns up to max(cache_capacity, the store's capacity) pieces)
    assert memo["fixed"] > 1000 and memo["cache_capacity"] == 20000 and memo["memo_learn"] == 0
    assert 0 < memo["learned"] <= max(memo["cache_capacity"], memo["store"]["capacity"])
    assert d["stress"]["reference_cache_count"]["value"] > 0
    assert d["end_to_end"]["value"] > 0 and "8 distinct batches" in d["config"]["workload"]
    assert d["parity_prefix_bit_exact"] is True


@pytest.mark.gpu
@pytest.mark.parametrize("config", ["1", "5", "r2d", "vocab_encoder", "pipeline"])
def test_bench_other_configs(config):
    d = _run("--config", config, "--no-extras")
    assert d["value"] > 0 and d["parity_prefix_bit_exact"] is True and 0 < d["roofline"]["frac"] < 1


@pytest.mark.gpu
@pytest.mark.parametrize("config", ["2", "3"])
def test_bench_exchange_loop(config):
    d = _run("--force-exchange", "--no-cpu-baseline", "--config", config)
    assert d["value"] > 0 and "all-gather" in d["config"]["exchange"] and "shard_unpack" in d["kernel_ms"]
    if config == "2":   # the fused BPE encode writes the send wire itself
        assert "shard_pack" not in d["kernel_ms"] and "written by the encode itself" in d["config"]["exchange"]
    else:
        assert "shard_pack" in d["kernel_ms"]


@pytest.mark.gpu
def test_bench_exchange_loop_packed():
    d = _run("--force-exchange", "--no-cpu-baseline", "--wire", "0", "--exchange", "p2p")
    assert d["value"] > 0 and "shard_pack" in d["kernel_ms"] and "grouped direct" in d["config"]["exchange"]


def test_relaunch_argv():
    """`python bench.py --gpus N` without WORLD_SIZE replaces itself with N ranks under torch.distributed.run (CPU: the
    command only)."""
    sys.path.insert(0, str(ROOT))
    import bench
    argv = bench.relaunch_argv(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], port=29611)
    assert argv[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in argv and "--nproc-per-node=8" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1" and argv[argv.index("--master-port") + 1] == "29611"
    k = argv.index(str(ROOT / "bench.py"))
    assert argv[k + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]


def test_gpus_must_match_world_size():
    """Launched as 2 ranks but told --gpus 1 (or the reverse): refuse before touching a GPU."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0 and "must agree" in p.stderr


@pytest.mark.gpu
def test_bench_through_the_relaunch():
    """--spawn: the path `--gpus N` takes (re-exec under torch.distributed.run), with one rank and its RCCL group."""
    d = _run("--spawn", "--force-exchange", "--no-cpu-baseline", "--no-extras")
    assert d["n_gpus"] == 1 and d["value"] > 0 and "all-gather" in d["config"]["exchange"]
    assert d["config"]["world"] == {"ranks": 1, "launched_by": "torch.distributed.run"}
