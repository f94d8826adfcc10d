"""One process per GPU, started from a plain shell (environments shard across ranks; the reference's
call shape is one driver process, `main.py:4-7`).

`spawn_ranks` starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N <script> <argv>` as a CHILD
process and relays its output, so that `python bench.py --gpus N` works without a launcher.  It is called before
anything in the parent has initialised the GPU (this module imports neither torch nor the native library): the
parent never becomes a rank, it only waits for the child and exits with its code.  Rendezvous on 127.0.0.1 (the
container's host name may not resolve).
"""
import os
import socket
import subprocess
import sys


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_command(nproc, script, argv, port=None):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(nproc)}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), str(script), *[str(a) for a in argv]]


def spawn_ranks(nproc, script, argv, env=None, out=None, timeout=None):
    """Runs `script argv` on `nproc` ranks of this node; the child's stdout (rank 0's JSON line among it) is
    relayed line by line to `out` (default: this process's stdout), its stderr goes to ours.  Returns the
    child's exit code (non-zero if any rank failed)."""
    if nproc < 1:
        raise ValueError("nproc must be >= 1")
    e = dict(os.environ if env is None else env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)  # the launcher sets them per rank
    out = sys.stdout if out is None else out
    p = subprocess.Popen(rank_command(nproc, script, argv), stdout=subprocess.PIPE, env=e, text=True)
    try:
        for line in p.stdout:
            out.write(line)
            out.flush()
        return p.wait(timeout=timeout)
    except BaseException:
        p.kill()
        p.wait()
        raise
