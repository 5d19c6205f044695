"""The N > 1 path on CPU: world_size-2 gloo run of the sharding + gather logic
(outfit_amd/dist.py). What is under test is the partition, the per-rank slicing incl. noise
(`shard_of`), the padded all_gather (`all_gather_records`) and the global ordering of the
96- and 480-byte records. The product's entry points have ONE code path, the HIP session; the
stand-in for it on a box without a GPU — the CPU oracle — lives HERE, in the tests
(`_distributed_with`), built from the product's own partition and gather functions."""
import os
import socket
import sys

import numpy as np
import pytest

import oracle_lib as O
import outfit_amd
from outfit_amd import dist as odist
from outfit_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ragged_batch():
    ds = synth.make_dataset(23, seed=9, n_nights=6, n_exp=4)  # 24 obs each
    n = 24
    counts = [24, 10, 24, 3, 0, 17, 24, 24, 5, 24, 12, 24, 24, 2, 24, 20, 24, 7, 24, 24, 9, 24, 24]
    cols = {k: [] for k in ("mjd_tt", "ra", "dec", "sigma_ra", "sigma_dec")}
    helio, geo, off = [], [], [0]
    for t, kc in enumerate(counts):
        for k in cols:
            cols[k].append(ds[k][t * n:t * n + kc])
        helio.append(ds["obs_helio_equ"][3 * t * n:3 * (t * n + kc)])
        geo.append(ds["obs_geo_ecl"][3 * t * n:3 * (t * n + kc)])
        off.append(off[-1] + kc)
    params = outfit_amd.IODParams(max_triplets=12, n_noise_realizations=2)
    noise = synth.SplitMix64(4).normal(len(counts) * 12 * 2 * 6)
    batch = outfit_amd.ObsBatch(np.array(off, dtype=np.uint64), *[np.concatenate(cols[k]) for k in cols],
                                np.concatenate(helio), np.concatenate(geo), noise)
    return batch, outfit_amd.Ephemeris.from_dict(ds["ephem"]), params


def _oracle_compute(ephem, params):
    def compute(sub):
        rc, res = O.fit_full_iod(params.as_c(), sub.as_c(), ephem.as_c(), 2, 1)
        assert rc == 0
        return res
    return compute


def _oracle_lsq_compute(ephem, params, cfg):
    def compute(sub, sub_init):
        rc, res, _ = O.fit_lsq(params.as_c(), cfg.as_c(), sub.as_c(), ephem.as_c(), sub_init, 2)
        assert rc == 0
        return res
    return compute


def _distributed_with(compute, batch, params, dist, width, dtype, initial=None):
    """fit_*_distributed with `compute` in the place of the rank's HIP session: same shard, same gather."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    sub, counts, t0, t1 = odist.shard_of(batch, params, rank, world)
    args = (sub,) if initial is None else (sub, np.ascontiguousarray(initial[t0:t1]))
    recs = compute(*args) if counts[rank] else np.zeros(0, dtype=dtype)
    local = torch.from_numpy(np.frombuffer(recs.tobytes(), dtype=np.uint8).reshape(-1, width).copy())
    out = odist.all_gather_records(local, counts, dist)
    return np.frombuffer(out.numpy().tobytes(), dtype=dtype)


def _worker(rank, world, port, out_dir):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    batch, ephem, params = _ragged_batch()
    recs = _distributed_with(_oracle_compute(ephem, params), batch, params, dist, odist.RECORD_BYTES,
                             outfit_amd._abi.RESULT_DTYPE)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), recs)
    cfg = outfit_amd.DifferentialCorrectionConfig(rms_divergence_ratio=10.0)
    seeds = recs.copy()
    seeds["status"][::3] = 2          # a third of the seeds "missing": the IOD is rerun for them
    lsq = _distributed_with(_oracle_lsq_compute(ephem, params, cfg), batch, params, dist, odist.LSQ_RECORD_BYTES,
                            outfit_amd._abi.LSQ_DTYPE, initial=seeds)
    np.save(os.path.join(out_dir, f"lsq_rank{rank}.npy"), lsq)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_of_slices_the_normals_with_the_trajectories():
    batch, _, params = _ragged_batch()
    stride = params.max_triplets * params.n_noise_realizations * 6
    seen = 0
    for rank in range(3):
        sub, counts, t0, t1 = odist.shard_of(batch, params, rank, 3)
        assert sub.n_traj == counts[rank] == t1 - t0 and t0 == seen
        assert np.array_equal(sub.noise_z, batch.noise_z[t0 * stride:t1 * stride])
        assert np.array_equal(sub.mjd_tt, batch.mjd_tt[int(batch.traj_offset[t0]):int(batch.traj_offset[t1])])
        sub.validate(params)
        seen = t1
    assert seen == batch.n_traj


def test_partition_is_contiguous_and_balanced():
    off = np.array([0, 20, 40, 45, 45, 100, 120, 140, 160, 200], dtype=np.uint64)
    for world in (1, 2, 3, 4, 8, 16):
        cuts = odist.partition_by_observations(off, world)
        assert cuts[0] == 0 and cuts[-1] == 9 and len(cuts) == world + 1
        assert all(b >= a for a, b in zip(cuts, cuts[1:]))
    cuts = odist.partition_by_observations(np.arange(0, 2001, 20, dtype=np.uint64), 4)
    assert cuts == [0, 25, 50, 75, 100]


@pytest.mark.timeout(300)
def test_two_rank_gloo_gather_equals_single_process(tmp_path):
    import torch.multiprocessing as mp
    world = 2
    port = _free_port()
    mp.start_processes(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    batch, ephem, params = _ragged_batch()
    ref = _oracle_compute(ephem, params)(batch)
    for rank in range(world):
        got = np.load(os.path.join(tmp_path, f"rank{rank}.npy"))
        assert got.dtype == ref.dtype and got.shape == ref.shape
        assert got.tobytes() == ref.tobytes(), f"rank {rank} gathered records differ from the single-process run"
    assert (ref["status"] == 1).sum() >= 2 and (ref["status"] == 0).sum() >= 10
    # the differential-correction records (480 B) through the same partition + padded gather
    cfg = outfit_amd.DifferentialCorrectionConfig(rms_divergence_ratio=10.0)
    seeds = ref.copy()
    seeds["status"][::3] = 2
    lsq_ref = _oracle_lsq_compute(ephem, params, cfg)(batch, seeds)
    for rank in range(world):
        got = np.load(os.path.join(tmp_path, f"lsq_rank{rank}.npy"))
        assert got.dtype == lsq_ref.dtype and got.shape == lsq_ref.shape
        assert got.tobytes() == lsq_ref.tobytes(), f"rank {rank}: gathered LSQ records differ"
    assert (lsq_ref["status"] == 0).sum() >= 5
