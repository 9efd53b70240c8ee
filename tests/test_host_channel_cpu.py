"""The host channel of a multi-rank run (pengk_comm_host_*, peng-motif_amd/csrc/comm.hip): a star of TCP connections to
rank 0 that carries what the sharded FASTA ingest has to agree on.  CPU only: the library loads without a GPU and the
channel never touches the device.  Covered: the collectives, the deadline on a missing rank, a stray connection that
is not admitted, and a rank that dies."""
import os
import socket
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, %r)
import peng_motif_amd as pk
L = pk.lib()
rc = L.pengk_comm_host_init_env()
if rc:
    print("INIT_FAILED", rc, L.pengk_last_error().decode()); sys.exit(3)
r, w = C.c_int(), C.c_int()
L.pengk_comm_host_info(C.byref(r), C.byref(w))
rank, world = r.value, w.value
mode = sys.argv[1]
if mode == "die" and rank == 1:
    os._exit(0)
send = np.arange(5, dtype=np.int64) + 100 * rank
recv = np.zeros(5 * world, np.int64)
rc = L.pengk_comm_host_allgather(send.ctypes.data, recv.ctypes.data, send.nbytes)
if rc:
    print("GATHER_FAILED", rc, L.pengk_last_error().decode()); sys.exit(4)
want = np.concatenate([np.arange(5) + 100 * q for q in range(world)])
assert np.array_equal(recv, want), recv
big = np.full(300000, rank + 1, np.uint64)
big[7] = 2 ** 40 + rank
assert L.pengk_comm_host_allreduce_u64(big.ctypes.data, big.size) == 0
assert big[0] == world * (world + 1) // 2 and big[7] == world * 2 ** 40 + world * (world - 1) // 2
assert L.pengk_comm_host_allreduce_u64(big.ctypes.data, 0) == 0
print("OK", rank, world)
L.pengk_comm_host_shutdown()
""" % ROOT


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def start(rank, world, port, mode="run", timeout_s=30, token=None):
    env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               PENGK_COMM_PORT=str(port), PENGK_COMM_TIMEOUT=str(timeout_s))
    if token:
        env["PENGK_COMM_TOKEN"] = token
    return subprocess.Popen([sys.executable, "-c", WORKER, mode], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)


def finish(procs, limit=120):
    out = []
    for p in procs:
        so, se = p.communicate(timeout=limit)
        out.append((p.returncode, so.decode(), se.decode()))
    return out


@pytest.mark.parametrize("world", [1, 2, 4])
def test_collectives(world):
    port = free_port()
    res = finish([start(r, world, port) for r in range(world)])
    for r, (rc, so, se) in enumerate(res):
        assert rc == 0 and so.strip() == "OK %d %d" % (r, world), (so, se[-400:])


def test_a_missing_rank_is_an_error_after_the_deadline_not_a_hang():
    port = free_port()
    t0 = time.time()
    (rc, so, se), = finish([start(0, 2, port, timeout_s=2)])
    assert rc == 3 and "INIT_FAILED" in so and "only 1 of 2 ranks" in so, (so, se[-300:])
    assert time.time() - t0 < 60
    # and a rank whose rank 0 never appears
    (rc, so, se), = finish([start(1, 2, free_port(), timeout_s=2)])
    assert rc == 3 and "not reachable" in so


def test_a_stray_connection_is_not_a_rank():
    """Something else connects to rank 0's port first -- silent, or talking nonsense: it is dropped, receives nothing,
    and the real ranks still find each other.  A rank of ANOTHER job (different token) is not admitted either."""
    port = free_port()
    p0 = start(0, 2, port, timeout_s=40)
    strays = []
    deadline = time.time() + 30
    while time.time() < deadline and len(strays) < 2:
        try:
            s = socket.create_connection(("127.0.0.1", port), timeout=1)
            strays.append(s)
        except OSError:
            time.sleep(0.05)
    assert len(strays) == 2
    strays[1].sendall(b"GET / HTTP/1.0\r\n\r\n" + b"x" * 64)
    other = start(1, 2, port, timeout_s=3, token="another job")
    (rc, so, se), = finish([other])
    assert rc == 3 and "did not admit" in so, so
    p1 = start(1, 2, port, timeout_s=40)
    res = finish([p0, p1])
    assert [r[0] for r in res] == [0, 0], res
    for s in strays:
        s.settimeout(2)
        try:
            assert s.recv(256) == b""  # closed without a byte
        except (ConnectionResetError, socket.timeout):
            pass
        s.close()


def test_a_rank_that_dies_fails_its_peers():
    port = free_port()
    t0 = time.time()
    res = finish([start(r, 2, port, mode="die", timeout_s=20) for r in range(2)])
    assert res[1][0] == 0
    assert res[0][0] == 4 and "GATHER_FAILED" in res[0][1] and "did not answer" in res[0][1], res[0]
    assert time.time() - t0 < 60


def test_silent_connections_do_not_hold_up_the_ranks():
    """Introductions are read without blocking: six connections that never say a word (each used to cost the accept loop a
    5 s blocking read, one after the other) leave the rendezvous of the real ranks where it was."""
    port = free_port()
    p0 = start(0, 2, port, timeout_s=60)
    strays = []
    deadline = time.time() + 30
    while time.time() < deadline and len(strays) < 6:
        try:
            strays.append(socket.create_connection(("127.0.0.1", port), timeout=1))
        except OSError:
            time.sleep(0.05)
    assert len(strays) == 6
    t0 = time.time()
    p1 = start(1, 2, port, timeout_s=60)
    res = finish([p0, p1])
    took = time.time() - t0
    assert [r[0] for r in res] == [0, 0], res
    assert took < 15, took  # process start + collectives; 30 s and more with serial blocking reads
    for s in strays:
        s.close()


def test_rank0_listens_on_all_interfaces_only_with_a_job_secret():
    """MASTER_ADDR that is not an address of this host (192.0.2.1, TEST-NET-1): without PENGK_COMM_TOKEN rank 0 refuses
    (the derived token is computable from public launcher values and authenticates nobody); with a secret it falls
    back to all interfaces and then waits for its peers as usual."""
    def rank0(token):
        env = dict(os.environ, RANK="0", WORLD_SIZE="2", MASTER_ADDR="192.0.2.1", MASTER_PORT="29999", PENGK_COMM_PORT=str(free_port()),
                   PENGK_COMM_TIMEOUT="2")
        env.pop("PENGK_COMM_TOKEN", None)
        if token:
            env["PENGK_COMM_TOKEN"] = token
        p = subprocess.Popen([sys.executable, "-c", WORKER, "run"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        (rc, so, se), = finish([p])
        return rc, so
    rc, so = rank0(None)
    assert rc == 3 and "not an address of this host" in so, so
    rc, so = rank0("a job secret")
    assert rc == 3 and "only 1 of 2 ranks" in so, so
