"""-m gpu: two ranks of the block-column multi-GPU driver on ONE MI355X (real kernels, real streams, gloo collectives).
The 8-GPU RCCL run itself only exists on the driver's node; this covers everything but the transport."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_two_ranks_share_one_gpu():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "mp_gpu_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "MP_GPU_OK" in out.stdout


def test_bench_contract_two_ranks_one_gpu():
    # bench.py exactly as the driver launches it for N > 1 (torch.distributed.run, RANK/LOCAL_RANK/WORLD_SIZE from the env), with
    # both ranks on the one GPU of the test box and gloo instead of RCCL: one JSON line, last on stdout, sane contents
    import json

    with socket.socket() as s2:
        s2.bind(("127.0.0.1", 0))
        port = s2.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RFLU_BENCH_ONE_GPU="1", RFLU_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--size", "3072",
           "--steps", "2", "--warmup", "1"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    line = out.stdout.strip().splitlines()[-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "GFLOP/s" and d["value"] > 0
    assert d["check"]["info"] == 0 and d["check"]["residual_matvec"] < 1e-12
    assert d["roofline"]["bound"] == "mfma" and d["cpu_baseline"] is None


def test_bench_contract_c_entry_two_ranks_one_gpu():
    # the same launch with the multi-GPU C entry (rflu_getrf_f64_mgpu) driven by rank 0: on this one-GPU box both logical
    # devices name GPU 0 (fake multi-GPU: copies instead of ncclBroadcast); rank 1 only takes part in the barriers
    import json

    with socket.socket() as s2:
        s2.bind(("127.0.0.1", 0))
        port = s2.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RFLU_BENCH_ONE_GPU="1", RFLU_BENCH_BACKEND="gloo", RFLU_BENCH_MGPU="c")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--size", "3072",
           "--steps", "2", "--warmup", "1"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["unit"] == "GFLOP/s" and d["value"] > 0
    assert "rflu_getrf_*_mgpu" in d["config"]["layout"]
    assert d["check"]["info"] == 0 and d["check"]["residual_matvec"] < 1e-12
    assert d["roofline"]["bound"] == "mfma"
