"""`python bench.py --gpus N` (N > 1) without a launcher: start the N ranks ourselves.

The driver launches multi-GPU runs as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N`
and every rank finds RANK / WORLD_SIZE in its environment.  A bare `python bench.py --gpus 8` used to run ONE rank and label
the line n_gpus = 1 (honest, but not what was asked for).  Now: before anything initialises a GPU, the process checks that N
devices are visible and re-runs itself under torch.distributed.run as a CHILD process (never exec: a process that touched the
GPU must not be replaced), relays the child's stdout (the one JSON line) and exits with its code; with fewer than N devices it
exits non-zero with a message instead of measuring something else.
"""
import os
import socket
import subprocess
import sys


def free_port() -> int:
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def needs_spawn(gpus: int, env=os.environ) -> bool:
    return gpus > 1 and 'WORLD_SIZE' not in env and 'RANK' not in env


def spawn_ranks(gpus: int, script: str, argv) -> int:
    """run `script argv` as `gpus` ranks of one node; returns the launcher's exit code"""
    import torch  # (device_count() does not initialise the GPU on this image)
    share = bool(os.environ.get('SKGS_SHARE_GPU'))
    have = torch.cuda.device_count()
    if have < gpus and not share:
        sys.stderr.write(f'bench.py --gpus {gpus}: only {have} device(s) visible; refusing to run fewer ranks than asked for '
                         f'(SKGS_SHARE_GPU=1 SKGS_DIST_BACKEND=gloo lets ranks share a device, for tests)\n')
        return 2
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), script] + list(argv)
    sys.stderr.write('bench.py: no launcher in the environment, starting the ranks: ' + ' '.join(cmd) + '\n')
    return subprocess.run(cmd, env=env).returncode


def cluster_info(dist, torch, local_rank: int):
    """what proves N ranks on N devices: backend, world size, every rank's device (name, uuid / bus id) and the RCCL version"""
    info = dict(backend=None, world=1, devices=None, rccl_version=None, torch=torch.__version__)
    p = torch.cuda.get_device_properties(local_rank)
    me = dict(rank=int(os.environ.get('RANK', '0')), local_rank=local_rank, name=p.name,
              uuid=str(getattr(p, 'uuid', '')) or None, pci_bus_id=getattr(p, 'pci_bus_id', None),
              cus=p.multi_processor_count, hbm_gb=round(p.total_memory / 2 ** 30, 1))
    try:
        v = torch.cuda.nccl.version()
        info['rccl_version'] = '.'.join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception:
        pass
    if dist.is_initialized():
        info['backend'], info['world'] = dist.get_backend(), dist.get_world_size()
        every = [None] * info['world']
        dist.all_gather_object(every, me)
        info['devices'] = every
        ids = [d.get('uuid') or d.get('pci_bus_id') or d['local_rank'] for d in every]
        info['distinct_devices'] = len(set(map(str, ids)))
    else:
        info['devices'] = [me]
        info['distinct_devices'] = 1
    return info
