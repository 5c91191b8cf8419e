# SPDX-License-Identifier: Apache-2.0
"""The process-per-GPU leg of bench.py exactly as the driver launches it at N > 1 -- `python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N` -- on the one GPU of the test box: ASTC_BENCH_SHARE_GPU=1 puts both ranks on device 0
and swaps RCCL (which refuses two ranks on one device) for gloo in the timing barrier and the max-over-ranks reduction;
everything else (rank environment, per-rank context and image, the JSON line of rank 0) is the code an 8-GPU node runs."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
def test_two_ranks_print_one_complete_line(built):
    env = dict(os.environ, ASTC_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-extra"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                 # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 1 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["unit"] == "Mtexels/s" and line["higher_is_better"] is True and line["vs_baseline"] is None
    # whole-job value: two images' texels over the slowest rank's time
    texels = 2 * 8192 * 8192
    assert abs(line["value"] - texels / (line["ms_per_step"] * 1e-3) / 1e6) / line["value"] < 1e-3
    for key in ("roofline", "cpu_baseline", "quality"):
        assert key in line, sorted(line)
    assert line["roofline"]["kernel"] == "astcd::astc_compress_blocks_ldr_6x6m"
    assert line["cpu_baseline"]["blocks_mismatching_gpu"] == 0 and line["cpu_baseline"]["value"] > 0


@pytest.mark.gpu
def test_rccl_calls_of_the_multi_rank_launch_with_one_rank(built):
    """The RCCL leg itself -- init_process_group(backend="nccl", device_id=...), the barriers around the timed region, the
    MAX all-reduce of the elapsed time on a device tensor, destroy_process_group -- cannot run with two ranks on one GPU; with
    ONE rank under torch.distributed.run (ASTC_BENCH_FORCE_DIST=1) the same calls execute against the real RCCL."""
    env = dict(os.environ, ASTC_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1")
    env.pop("ASTC_BENCH_SHARE_GPU", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--no-extra",
           "--no-cpu-baseline", "--no-host-api"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["value"] > 0 and abs(line["value"] - 8192 * 8192 / (line["ms_per_step"] * 1e-3) / 1e6) / line["value"] < 1e-3
