"""The N > 1 code path of bench.py on the one GPU of the test box — needs an MI355X.

`bench.py --force-dist` creates the RCCL process group (backend "nccl"), runs the step with the all_gather of the
96-byte records on RCCL's stream (double-buffered, overlapped with the next step), the barrier / max-over-ranks
timing and the whole-table check every rank makes (per-rank CRC exchanged with a second all_gather) — with
world_size 1, which is what one GPU allows. The scaling curve itself (N = 1, 2, 4, 8) is the driver's to measure;
the reference analogue of the gather is the Rayon reduce of per-thread maps (obs_dataset_api.rs:191-205).
The child is a fresh interpreter: nothing has touched the GPU before the process group comes up.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra, port="29541"):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    # bench.py itself puts HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC, what RCCL needs on this pool) into its environment
    # before the first HIP call: the child starts WITHOUT the variable, as under a launcher that does not export it
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-side", "--no-lsq", "--no-modes", *extra],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-1500:]          # ONE JSON line on stdout (RCCL's banner must not land there)
    return checked_line(p.stdout, lines[0])


def checked_line(stdout, text):
    """The printed line is the LAST thing on stdout and short enough for the driver's 8 KB tail; the per-rank tables
    live in the detail file it names (returned under "_detail")."""
    assert stdout.rstrip().splitlines()[-1] == text and len(text) < 4096, len(text)
    line = json.loads(text)
    line["_detail"] = json.load(open(os.path.join(ROOT, line["detail"])))
    return line


def test_rccl_gather_path_with_one_rank():
    line = run_bench("--force-dist", "--objects", "20000")
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["scaling"] == "weak"
    assert line["math_mode"] == "glibc"
    assert line["config"]["orbits_found_rank0"] > 19000
    gc = line["gather_check"]
    assert gc["ranks_verified_by_every_rank"] == 1 and gc["records"] == 20000
    assert len(line["_detail"]["gather_check"]["crc32_per_rank"]) == 1
    assert line["roofline"]["kernel"].startswith("k_") and line["value"] > 1e5


def test_strong_scaling_shape_with_one_rank():
    """--scaling strong: one global dataset of 25 000-object blocks split into contiguous ranges (configs[3] is
    10^6 objects over 8 GPUs; here 50 000 over the one GPU, through the same code)."""
    line = run_bench("--force-dist", "--scaling", "strong", "--total-objects", "50000", port="29542")
    assert line["scaling"] == "strong" and line["config"]["objects_per_gpu"] == [50000]
    assert line["gather_check"]["records"] == 50000
    assert abs(line["value"] - 50000 * 2 / (line["ms_per_step"] * 2e-3)) < 1e-6 * line["value"]


def test_world_size_must_match_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], env=env, capture_output=True,
                       text=True, timeout=300, cwd=ROOT)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


def test_two_ranks_share_the_gpu_through_gloo_unequal_shards():
    """World size 2 on the one GPU (rehearsal mode: gloo gathers host copies, both ranks use GPU 0): the barrier /
    max-over-ranks timing, the padded gather of UNEQUAL shards (75 000 objects as 3 blocks over 2 ranks: 25 000 +
    50 000) and the whole-table check on every rank — each rank verifies both shards against the CRCs exchanged."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29543", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "2", "--warmup", "1", "--backend", "gloo", "--single-device",
                        "--scaling", "strong", "--total-objects", "75000", "--no-cpu-baseline", "--no-lsq"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-1500:]
    line = checked_line(p.stdout, lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["objects_per_gpu"] == [25000, 50000]
    gc = line["gather_check"]
    assert gc["ranks_verified_by_every_rank"] == 2 and gc["records"] == 75000 and len(gc["ms_per_step_per_rank"]) == 2
    assert gc["ipc_mode"] == "HSA_ENABLE_IPC_MODE_LEGACY=0"
    gc = line["_detail"]["gather_check"]
    assert len(set(gc["crc32_per_rank"])) == 2
    # per-rank timing of the timed region, and the IPC mode the ranks ran with (set by bench.py itself)
    pr = gc["per_rank"]
    assert all(len(pr[k]) == 2 for k in ("ms_per_step", "gather_wait_ms_per_step", "host_enqueue_ms_per_step",
                                         "main_stream_kernel_ms_per_step", "setup_s_before_timed_region"))
    assert max(pr["ms_per_step"]) <= line["ms_per_step"] * (1 + 1e-9) and min(pr["ms_per_step"]) > 0
    assert gc["ipc_mode"] == "HSA_ENABLE_IPC_MODE_LEGACY=0"
    assert abs(line["value"] - 75000 * 2 / (line["ms_per_step"] * 2e-3)) < 1e-6 * line["value"]
