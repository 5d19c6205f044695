"""The N > 1 code paths on ONE device — needs an MI355X (a multi-GPU node is not available to these tests; the
driver's scaling run is the first execution on several devices).

  * two gloo ranks, both on GPU 0, each with a REAL resident session (`ShardedIOD`): partition, per-rank kernels on an
    explicit stream, gather, global order; the shard is reused across runs; the corrector through the same path;
  * the in-process path of `outfit_hip_fit_full_iod(n_devices > 1)` — one host thread, one session and one slice of
    the output per shard — with two and three shards mapped to device 0 (OUTFIT_SHARD_DEVICES).
Reference: trajectories are independent tasks (obs_dataset_api.rs:277-296); the Rayon fold/reduce (:198-205) is the
only exchange."""
import os
import socket
import sys

import numpy as np
import pytest

import outfit_amd
from outfit_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _case():
    eph_d = synth.make_ephemeris()
    ds = synth.make_dataset(3001, seed=4242, ephem=eph_d, topocentric=True)
    return outfit_amd.ObsBatch.from_dict(ds), outfit_amd.Ephemeris.from_dict(eph_d), synth.bench_params(outfit_amd.IODParams())


def _worker(rank, world, port, out_dir):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import torch
    import torch.distributed as dist
    from outfit_amd import dist as odist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    batch, ephem, params = _case()
    sh = odist.ShardedIOD(batch, ephem, params, dist, device_index=0)
    a = sh.run()
    b = sh.run()                      # the resident shard is reused
    c = odist.fit_full_iod_distributed(batch, ephem, params, dist, 0, sharded=sh)
    sh.close()
    d = odist.fit_full_iod_distributed(batch, ephem, params, dist, 0)   # builds and closes its own shard
    assert a.tobytes() == b.tobytes() == c.tobytes() == d.tobytes()
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), a)
    cfg = outfit_amd.DifferentialCorrectionConfig(rms_divergence_ratio=10.0)
    lsq = odist.fit_lsq_distributed(batch, ephem, params, cfg, dist, device_index=0, initial_orbits=a)
    np.save(os.path.join(out_dir, f"lsq_rank{rank}.npy"), lsq)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_gloo_ranks_with_real_sessions_on_gpu0(tmp_path):
    import torch.multiprocessing as mp
    world = 2
    mp.start_processes(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    batch, ephem, params = _case()
    ref = outfit_amd.fit_full_iod_records(batch, ephem, params, n_devices=1)
    assert (ref["status"] == 0).sum() > 2900
    for rank in range(world):
        got = np.load(os.path.join(tmp_path, f"rank{rank}.npy"))
        assert got.tobytes() == ref.tobytes(), f"rank {rank}: gathered records differ from the single-device run"
    cfg = outfit_amd.DifferentialCorrectionConfig(rms_divergence_ratio=10.0)
    lref, _ = outfit_amd.fit_lsq_records(batch, ephem, params, cfg, ref, n_devices=1)
    for rank in range(world):
        got = np.load(os.path.join(tmp_path, f"lsq_rank{rank}.npy"))
        assert got.tobytes() == lref.tobytes(), f"rank {rank}: gathered LSQ records differ"


@pytest.mark.parametrize("shards", ["0,0", "0,0,0"])
def test_per_device_thread_path_with_logical_shards(shards, monkeypatch):
    batch, ephem, params = _case()
    ref = outfit_amd.fit_full_iod_records(batch, ephem, params, n_devices=1)
    cfg = outfit_amd.DifferentialCorrectionConfig(rms_divergence_ratio=10.0)
    lref, oref = outfit_amd.fit_lsq_records(batch, ephem, params, cfg, None, n_devices=1)
    monkeypatch.setenv("OUTFIT_SHARD_DEVICES", shards)
    got = outfit_amd.fit_full_iod_records(batch, ephem, params, n_devices=0)
    assert got.tobytes() == ref.tobytes()
    lgot, ogot = outfit_amd.fit_lsq_records(batch, ephem, params, cfg, None, n_devices=0)
    assert lgot.tobytes() == lref.tobytes() and ogot.tobytes() == oref.tobytes()
    # a whole-call failure in ONE shard fails the call with that shard's message
    bad = outfit_amd.ObsBatch.from_dict(synth.make_dataset(3001, seed=4242, ephem=synth.make_ephemeris(), topocentric=True))
    bad.mjd_tt[-5:] += 1.0e5     # last shard only: outside the ephemeris table
    with pytest.raises(outfit_amd.OutfitError) as e:
        outfit_amd.fit_full_iod_records(bad, ephem, params, n_devices=0)
    assert "ephemeris" in str(e.value)
