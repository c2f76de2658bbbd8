"""Host-side rendezvous of the data-parallel ranks of ONE node -- no torch, no sockets.

All that has to travel between the processes before RCCL takes over is the 128-byte communicator id (and, for the
training loop, one shuffle seed): rank 0 writes them to a file that only the ranks of THIS launch look for, the others
poll for it.  Every later exchange (barriers, max-over-ranks of a time) runs over RCCL itself
(EdgeGAN.comm_reduce_scalars).  The file lives in a private (0700) per-user directory, is created with O_EXCL |
O_NOFOLLOW, and is named after the launch: the launcher's process id -- the ranks of one `python -m
torch.distributed.run` (or any launcher that forks them) share their parent -- plus the elastic run id and MASTER_PORT;
rank 0 removes it once every rank holds a live communicator.  EG_RDZV_KEY names the launch explicitly for launchers
whose ranks do NOT share a parent (the ranks would otherwise wait out the 300 s timeout); EG_RDZV_DIR overrides the
directory (tests)."""
from __future__ import annotations

import os
import stat
import tempfile
import time


def _dir() -> str:
    """A per-user directory with mode 0700 (created on demand, ownership and mode checked): other local users can
    neither pre-create the rendezvous file nor plant a symlink under its name.  EG_RDZV_DIR overrides it (tests)."""
    d = os.environ.get('EG_RDZV_DIR')
    if d:
        return d
    d = os.path.join(tempfile.gettempdir(), f'edgegan_rdzv_{os.getuid()}')
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise RuntimeError(f'rendezvous directory {d} is not a private directory of this user')
    return d


def _key() -> str:
    """What the ranks of ONE launch share and no other launch does.  EG_RDZV_KEY if given (launchers whose ranks do not
    share a parent process: each rank started through its own wrapper shell, srun, mpirun ...); else the elastic run id
    that torch.distributed.run exports to every worker (TORCHELASTIC_RUN_ID) when it is a real one; else the launcher's
    process id.  MASTER_PORT is always part of it, so two launches from one shell cannot collide either."""
    port = os.environ.get('MASTER_PORT', '0')
    if os.environ.get('EG_RDZV_KEY'):
        return os.environ['EG_RDZV_KEY']
    run_id = os.environ.get('TORCHELASTIC_RUN_ID', '')
    if run_id and run_id != 'none':
        return f'run{run_id}_{os.getppid()}_{port}'
    return f'{os.getppid()}_{port}'


def _path(tag: str) -> str:
    return os.path.join(_dir(), f'edgegan_rdzv_{_key()}_{tag}')


def broadcast_bytes(tag: str, payload: bytes | None, rank: int, timeout: float = 300.0) -> bytes:
    """Rank 0 publishes `payload` under `tag`; every rank returns it.  One use per tag and launch."""
    path = _path(tag)
    if rank == 0:
        assert payload is not None
        tmp = f'{path}.tmp{os.getpid()}'
        for stale in (tmp,):
            try:
                os.remove(stale)
            except OSError:
                pass
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, 'O_NOFOLLOW', 0), 0o600)
        with os.fdopen(fd, 'wb') as f:
            f.write(payload)
        os.replace(tmp, path)          # atomic: a reader sees nothing or everything
        return payload
    t0 = time.time()
    while True:
        try:
            fd = os.open(path, os.O_RDONLY | getattr(os, 'O_NOFOLLOW', 0))
            with os.fdopen(fd, 'rb') as f:
                return f.read()
        except FileNotFoundError:
            if time.time() - t0 > timeout:
                raise TimeoutError(f'rank {rank}: no rendezvous file {path} after {timeout:.0f} s -- do the ranks of this '
                                   f'launch share a parent process?  If not, export the same EG_RDZV_KEY to all of them')
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
