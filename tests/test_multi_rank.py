"""N>1 control flow on CPU (world_size 2, gloo): pictures shard by picture with no collective on
the data path; only the barrier and the MAX-over-ranks timing use torch.distributed."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_picture_shard_is_a_partition():
    sys.path.insert(0, ROOT)
    import bench
    for world in (1, 2, 4, 8):
        for n in (1, 7, 16, 33):
            shards = [bench.picture_shard(n, r, world) for r in range(world)]
            flat = sorted(k for s in shards for k in s)
            assert flat == list(range(n))
            assert all(k % world == r for r, s in enumerate(shards) for k in s)


def test_bench_two_ranks_gloo_dry_run():
    env = dict(os.environ, VC2_BENCH_DRYRUN="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "3"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout          # exactly one JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == 2 and d["pictures"] == 6
    assert d["ms_per_step"] >= 19.0             # MAX over ranks: rank 1 sleeps 20 ms
    # the ranks coded real pictures (through the oracle): the same six pictures coded by ONE rank give the same streams
    # and the same decoded pictures (an order-free sum of per-picture digests)
    env1 = dict(os.environ, VC2_BENCH_DRYRUN="1")
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "6"], env=env1, capture_output=True,
                         text=True, timeout=300)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])
    assert d1["pictures"] == 6 and d1["check"] == d["check"]


def test_oracle_frames_are_independent(oracle):
    """The property the sharding relies on: encoding frames separately and concatenating the picture
    data units equals encoding them in one run (only parse offsets / picture numbers chain)."""
    from synth import synth
    from vc2lib import make_params
    w, h = 128, 64
    raw = synth(w, h, "422", 10, 3, frames=2)
    p = make_params(w, h, "422", 10, "LeGall", 2, 2, 4, q=7)
    both = oracle.encode_stream(p, raw, 2)
    half = len(raw) // 2
    s0 = oracle.encode_stream(p, raw[:half], 1)
    s1 = oracle.encode_stream(p, raw[half:], 1)

    def picture_payload(stream):   # bytes after the 13-byte parse info + 4-byte picture number
        i = stream.find(b"BBCD\xe8")
        nxt = int.from_bytes(stream[i + 5:i + 9], "big")
        return stream[i + 17:i + nxt]

    i0 = both.find(b"BBCD\xe8")
    n0 = int.from_bytes(both[i0 + 5:i0 + 9], "big")
    assert both[i0 + 17:i0 + n0] == picture_payload(s0)
    i1 = i0 + n0
    n1 = int.from_bytes(both[i1 + 5:i1 + 9], "big")
    assert both[i1 + 17:i1 + n1] == picture_payload(s1)
