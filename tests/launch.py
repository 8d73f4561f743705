"""Launching `torch.distributed.run` from a test: the launcher needs a TCP port, and "bind port 0, close it, hand the number on"
can lose the port to another process before the launcher listens (EADDRINUSE — seen once in six full GPU-suite runs in round 6,
where `-x` turned it into the end of the run).  A launch that failed for THAT reason is repeated with a fresh port."""
import socket
import subprocess


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def run_with_port(make_cmd, attempts=3, **kw):
    """make_cmd(port) -> argv; subprocess.run(**kw) with capture_output / text forced on.  -> CompletedProcess"""
    kw.update(capture_output=True, text=True)
    r = None
    for _ in range(attempts):
        r = subprocess.run(make_cmd(free_port()), **kw)
        if r.returncode == 0 or "EADDRINUSE" not in (r.stderr or "") + (r.stdout or ""):
            break
    return r
