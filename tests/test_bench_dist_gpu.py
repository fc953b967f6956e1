"""bench.py's N > 1 code path through the REAL collective backend on a one-GPU box: `torch.distributed.run` with one rank,
backend "nccl" (= RCCL on ROCm), the row-parallel step with its all-reduce inside the timed region, hipGraph capture or
the eager fallback.  No 8-GPU node has been available to the driver in four rounds; this is what can be verified here so
that the first scaling run is boring."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.gpu
def test_bench_world1_through_rccl():
    env = dict(os.environ, VPTQ_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", VPTQ_BENCH_CONDITIONING_S="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
           "--regions", "1", "--tp-layers", "1", "--no-extras"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # exactly ONE JSON line on stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["config"]["mode"] == "tp_row" and out["scaling"] == "strong"
    assert out["process"] == {"rank": 0, "local_rank": 0, "world": 1, "device": 0, "collective_backend": "nccl"}
    assert out["value"] > 0 and out["roofline"]["frac"] > 0
    assert out["tp_row"]["parity_rel_err_vs_cpu_oracle"] <= 1e-3     # the reduced outputs against the C oracle
    print(f"\n[bench world 1 / nccl] hipgraph={out['config']['hipgraph']} value={out['value']:.0f} GB/s")
