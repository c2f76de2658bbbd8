"""Host-side rendezvous of the data-parallel ranks of ONE node -- no torch, no sockets.

All that has to travel between the processes before RCCL takes over is the 128-byte communicator id (and, for the
training loop, one shuffle seed): rank 0 writes them to a file that only the ranks of THIS launch look for, the others
poll for it.  Every later exchange (barriers, max-over-ranks of a time) runs over RCCL itself
(EdgeGAN.comm_reduce_scalars).  The file name is derived from the launcher's process id -- the ranks of one
`python -m torch.distributed.run` (or any launcher that forks them) share their parent -- plus MASTER_PORT, so a stale
file of an earlier launch is never picked up; EG_RDZV_DIR / EG_RDZV_KEY override directory and key (tests, other
launchers)."""
from __future__ import annotations

import os
import tempfile
import time


def _path(tag: str) -> str:
    d = os.environ.get('EG_RDZV_DIR') or tempfile.gettempdir()
    key = os.environ.get('EG_RDZV_KEY') or f"{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}"
    return os.path.join(d, f'edgegan_rdzv_{key}_{tag}')


def broadcast_bytes(tag: str, payload: bytes | None, rank: int, timeout: float = 300.0) -> bytes:
    """Rank 0 publishes `payload` under `tag`; every rank returns it.  One use per tag and launch."""
    path = _path(tag)
    if rank == 0:
        assert payload is not None
        tmp = f'{path}.tmp{os.getpid()}'
        with open(tmp, 'wb') as f:
            f.write(payload)
        os.replace(tmp, path)          # atomic: a reader sees nothing or everything
        return payload
    t0 = time.time()
    while True:
        try:
            with open(path, 'rb') as f:
                return f.read()
        except FileNotFoundError:
            if time.time() - t0 > timeout:
                raise TimeoutError(f'rank {rank}: no rendezvous file {path} after {timeout:.0f} s')
            time.sleep(0.005)


def cleanup(tags, rank: int) -> None:
    """Rank 0 removes its files (call after every rank has read them, i.e. after comm_init returned everywhere)."""
    if rank != 0:
        return
    for tag in tags:
        try:
            os.remove(_path(tag))
        except OSError:
            pass


def init_data_parallel(net, rank: int, world: int) -> None:
    """The whole DP bring-up: RCCL unique id from rank 0 to everyone, communicator init, first barrier."""
    uid = broadcast_bytes('uid', net.comm_unique_id() if rank == 0 else None, rank)
    net.comm_init(uid, rank, world)
    net.comm_barrier()                 # every rank holds the id and a live communicator
    cleanup(['uid'], rank)
