"""Multi-GPU sharding of the IOD path: one process per GPU, torch.distributed (RCCL on ROCm).

Trajectories are independent tasks (reference: `process_traj`,
src/initial_orbit_determination/obs_dataset_api.rs:277-296; the Rayon
`fold`/`reduce` of per-thread maps at :198-205 is the only "collective"). Here each
rank takes a contiguous range of trajectories balanced by observation count, runs
the kernels on its own GPU with no data-path exchange, and the fixed-size result
records are gathered once at the end (all_gather over xGMI; 96 B per trajectory).

Environment: RCCL between processes shares device memory through IPC handles; on hosts whose driver supports only
dmabuf IPC (this MI355X pool) `HSA_ENABLE_IPC_MODE_LEGACY=0` must be in the environment BEFORE the process makes its
first HIP call, otherwise communicator set-up fails with `hipIpcGetMemHandle: invalid argument`. Importing this module
sets it when the variable is absent (an explicit value is respected); bench.py does the same at its first line.
"""
import os

import numpy as np

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from ._abi import LSQ_DTYPE, RESULT_DTYPE

RECORD_BYTES = 96
LSQ_RECORD_BYTES = 480


def partition_by_observations(traj_offset, world):
    """Cut points c[0..world] so that rank r owns trajectories [c[r], c[r+1]) and the ranges
    carry ~equal numbers of observations (same rule as outfit_hip_fit_full_iod in api.hip)."""
    traj_offset = np.asarray(traj_offset, dtype=np.uint64)
    n_traj = len(traj_offset) - 1
    o0 = int(traj_offset[0])
    total = int(traj_offset[-1]) - o0
    cuts = [0]
    for d in range(1, world):
        target = o0 + total * d // world
        lo = int(np.searchsorted(traj_offset[:n_traj], np.uint64(target), side="left"))
        cuts.append(max(cuts[-1], lo))
    cuts.append(n_traj)
    return cuts


def all_gather_records(local, counts, dist):
    """Gather per-rank record tensors (uint8 [n_r, 96]) into one [sum n_r, 96] tensor in rank order.
    `counts` are the per-rank trajectory counts (known to every rank from the partition)."""
    import torch
    world = dist.get_world_size()
    n_max = max(counts)
    width = local.shape[1]  # 96 (outfit_gauss_result_t) or 480 (outfit_lsq_result_t)
    if all(c == n_max for c in counts):
        out = torch.empty((world * n_max, width), dtype=torch.uint8, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    pad = torch.zeros((n_max, width), dtype=torch.uint8, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * n_max, width), dtype=torch.uint8, device=local.device)
    dist.all_gather_into_tensor(out, pad)
    return torch.cat([out[r * n_max: r * n_max + counts[r]] for r in range(world)], dim=0)


def shard_of(batch, params, rank, world):
    """(this rank's sub-batch, per-rank trajectory counts, t0, t1): the contiguous range the partition gives `rank`,
    with its slice of the host-drawn normals (per trajectory max_triplets * n_noise_realizations * 6 of them)."""
    cuts = partition_by_observations(batch.traj_offset, world)
    counts = [cuts[r + 1] - cuts[r] for r in range(world)]
    t0, t1 = cuts[rank], cuts[rank + 1]
    sub = batch.slice(t0, t1)
    if batch.noise_z is not None:
        stride = params.max_triplets * params.n_noise_realizations * 6
        sub.noise_z = batch.noise_z[t0 * stride: t1 * stride]
    return sub, counts, t0, t1


def records_from_tensor(t):
    """uint8 [n, 96] tensor -> numpy structured array of outfit_gauss_result_t."""
    return np.frombuffer(t.cpu().numpy().tobytes(), dtype=RESULT_DTYPE)


class ShardedIOD:
    """This rank's shard of a batch, RESIDENT on its GPU, plus the gather of everybody's records.

    Build once, `run()` as often as needed (each call = the rank's kernels + one all_gather of the 96-byte records):
    the session, its result buffer and the gather buffers live as long as the object. Ordering is by stream, not by a
    device-wide synchronize: the kernels go to an explicit torch stream, the collective is issued under that stream
    (RCCL orders a collective after torch's CURRENT stream), the host waits once for the gathered table.
    With a gloo group (rehearsal on one GPU / CPU tensors) the records are copied to the host on that stream first."""

    def __init__(self, batch, ephem, params, dist, device_index):
        import torch
        from .api import IODSession
        self.dist, self.torch = dist, torch
        rank, world = dist.get_rank(), dist.get_world_size()
        sub, self.counts, self.t0, self.t1 = shard_of(batch, params, rank, world)
        self.rank = rank
        self.dev = torch.device("cuda", device_index)
        torch.cuda.current_stream(self.dev)                      # torch's HIP runtime first (see _lib.lib)
        self.stream = torch.cuda.Stream(self.dev)                # explicit: handle 0 would mean "the session's own"
        self.local = torch.empty((self.counts[rank], RECORD_BYTES), dtype=torch.uint8, device=self.dev)
        self.session = IODSession(sub, ephem, params, device=device_index) if self.counts[rank] else None
        self.nccl = dist.get_backend() == "nccl"

    def run(self):
        """-> structured records of ALL trajectories, in trajectory order."""
        torch = self.torch
        with torch.cuda.stream(self.stream):
            if self.session is not None:
                self.session.run(stream=self.stream.cuda_stream, d_out=self.local.data_ptr())
            if self.nccl:
                out = all_gather_records(self.local, self.counts, self.dist)
            else:
                host = self.local.to("cpu", non_blocking=False)  # waits for the kernels on this stream
                out = all_gather_records(host, self.counts, self.dist)
        if self.session is not None:
            self.session.check()   # an epoch outside the ephemeris table raises here, as on one GPU
        self.stream.synchronize()
        return records_from_tensor(out)

    def close(self):
        if self.session is not None:
            self.session.close()
            self.session = None


def fit_full_iod_distributed(batch, ephem, params, dist, device_index, sharded=None):
    """Strong-scaling entry: every rank passes the SAME batch, computes its own range on its GPU
    and returns the records of ALL trajectories (gathered). Pass `sharded` (a ShardedIOD built from the same
    arguments) to reuse the resident shard across calls; otherwise one is built and closed here.
    There is one code path: the HIP session (no device, no result)."""
    own = sharded is None
    sh = sharded or ShardedIOD(batch, ephem, params, dist, device_index)
    try:
        return sh.run()
    finally:
        if own:
            sh.close()


def fit_lsq_distributed(batch, ephem, iod_params, diff_cor_config, dist, device_index, initial_orbits=None):
    """`FitLSQ::fit_lsq` sharded like the IOD (the reference loops over trajectories one by one,
    differential_orbit_correction/obs_dataset_api.rs:76-96: no exchange between them). Every rank passes
    the SAME batch (and the same optional seeds) and gets the 480-byte records of ALL trajectories."""
    import torch
    from .api import IODSession
    rank, world = dist.get_rank(), dist.get_world_size()
    sub, counts, t0, t1 = shard_of(batch, iod_params, rank, world)
    sub_init = None if initial_orbits is None else np.ascontiguousarray(initial_orbits[t0:t1])
    if not counts[rank]:
        recs = np.zeros(0, dtype=LSQ_DTYPE)
    else:
        dev = torch.device("cuda", device_index)
        torch.cuda.current_stream(dev)  # torch's HIP runtime first (see _lib.lib)
        s = IODSession(sub, ephem, iod_params, device=device_index)
        stream = torch.cuda.current_stream(dev).cuda_stream
        if sub_init is None or (sub_init["status"] != 0).any():
            s.run(stream=stream)
        s.run_lsq(diff_cor_config, sub_init, stream=stream)
        recs, _ = s.lsq_results()
        s.close()
    local = torch.from_numpy(np.frombuffer(recs.tobytes(), dtype=np.uint8).reshape(-1, LSQ_RECORD_BYTES).copy())
    if dist.get_backend() == "nccl":
        local = local.to(torch.device("cuda", device_index))
    out = all_gather_records(local, counts, dist)
    return np.frombuffer(out.cpu().numpy().tobytes(), dtype=LSQ_DTYPE)
