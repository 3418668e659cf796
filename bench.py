#!/usr/bin/env python3
"""bench.py -- VC-2 HQ encode+decode throughput of the HIP hot path on MI355X.

A "step" = one pass of the hot path (encode_batch_dev then decode_batch_dev) over one batch of
synthetic pictures that are already resident in HBM.  Workload at every N: BASELINE.json config 2
(UHD-1 3840x2160 4:2:2 10-bit, HQ_ConstQ, DD97, 4 levels, slices -u 1 -a 2, q 16, scalar 2);
each rank/GPU owns its own batch (frames are independent: weak scaling, no collective on the data
path).  Prints ONE JSON line on rank 0.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--no-cpu-baseline]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "vc2-reference_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)


def picture_shard(n_pictures, rank, world):
    """Pictures are independent: picture k belongs to rank k mod world."""
    return list(range(rank, n_pictures, world))


def dry_run(args, rank, world):
    """CPU-only exercise of the multi-rank control flow (tests/test_multi_rank.py): rendezvous, barrier,
    MAX-over-ranks of the step time, picture sharding and the rank-0 JSON line.  No codec work is done
    and the line is marked dry_run: it is not a measurement."""
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    mine = picture_shard(args.batch * world, rank, world)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))       # ranks finish at different times: MAX must pick the slowest
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    counts = torch.tensor([len(mine)], dtype=torch.int64)
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "pictures": int(counts.item()),
                          "ms_per_step": dt.item() * 1e3, "scaling": "weak"}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="pictures per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=4, help="pictures in the CPU-baseline sample")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams the library cuts each batch over (vc2hip_set_streams); 1 = one launch per kernel and "
                         "batch, which is what the roofline figures describe")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N")
    if os.environ.get("VC2_BENCH_DRYRUN") == "1":
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import vc2hip_py
    from synth import synth

    W, H, CFMT, BITS = 3840, 2160, "422", 10
    KERNEL, DEPTH, U, A, Q, SCALAR = "DD97", 4, 1, 2, 16, 2
    hip = vc2hip_py.Vc2Hip(local_rank)
    if args.streams > 1:
        hip.set_streams(args.streams)
    fmt = vc2hip_py.picture_format(W, H, CFMT, BITS)
    cp = vc2hip_py.coding_params(hip.lib, fmt, KERNEL, DEPTH, U, A, q=Q, scalar=SCALAR)
    B = args.batch
    rb = hip.raw_picture_bytes(fmt)
    stride = (hip.max_payload_bytes(fmt, cp) + 255) // 256 * 256

    # synthetic pictures: SURVEY Appendix-B generator (seed 1234); 2 distinct frames tiled over the batch
    distinct = 2
    raw = synth(W, H, CFMT, BITS, 1234, frames=distinct)
    host = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
    d_raw = torch.empty(B * rb, dtype=torch.uint8, device=dev)
    for k in range(B):
        d_raw[k * rb:(k + 1) * rb] = host[(k % distinct) * rb:((k % distinct) + 1) * rb].to(dev)
    d_pay = torch.zeros(B * stride, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(B, dtype=torch.int64, device=dev)
    d_out = torch.zeros(B * rb, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    def step():
        hip.encode_batch_dev(d_raw.data_ptr(), B, fmt, cp, d_pay.data_ptr(), stride, d_len.data_ptr())
        hip.decode_batch_dev(d_pay.data_ptr(), stride, d_len.data_ptr(), B, fmt, cp, d_out.data_ptr())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    hip.sync()
    coded = int(d_len[0].item())

    # ---- timed region: exactly K steps, HIP events around every kernel on the library's stream
    hip.profile_reset()
    hip.profile_enable(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    hip.sync()  # collects the event pairs; raises on any device-side error flag
    hip.profile_enable(False)
    prof = hip.profile()

    # ---- same region without the per-kernel events (reported beside, not as `value`)
    barrier()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt_noev = time.perf_counter() - t1
    hip.sync()

    tmax = torch.tensor([dt, dt_noev], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt, dt_noev = tmax.tolist()

    # ---- parity spot check of what was timed (pictures 0 and 1: digests of reference output)
    ok = None
    if rank == 0:
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_digests.json")))["cfg2"]
        dec = d_out[:distinct * rb].cpu().numpy().tobytes()
        ok = hashlib.sha256(dec).hexdigest() == gold["decoded"]["sha256"]
        if not ok:
            raise SystemExit("decoded pictures differ from the reference digest: refusing to report a number")

    pixels = W * H
    total_px = pixels * B * world * args.steps
    value = total_px / dt / 1e6
    out = None
    if rank == 0:
        # algorithmic bytes: encode w*S + C, decode C + w*S per picture
        samples = W * H * 2  # 4:2:2
        alg_dir = 2 * samples + coded
        kern = {k: v for k, v in prof.items() if v[0] > 0 and k != "fill"}
        total_ms = sum(v[1] for v in kern.values())
        dom = max(kern, key=lambda k: kern[k][1])
        dom_launches, dom_ms = kern[dom]
        dom_avg_s = dom_ms / dom_launches / 1e3
        per_launch = B / max(1, min(args.streams, B))   # pictures one launch processes
        achieved = alg_dir * per_launch / dom_avg_s / 1e9
        path_achieved = 2 * alg_dir * B * args.steps / (total_ms / 1e3) / 1e9
        # HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and
        # WRITE_SIZE in separate runs, gfx950 correction applied; see profiles/r01_pmc_traffic.json)
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            if pmc.get("pictures_per_launch") == per_launch and dom in pmc["kernels"]:
                traffic = pmc["kernels"][dom]["hbm_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            traffic = None
        out = {
            "metric": "Mpixels/s encode+decode, UHD-1 10-bit HQ_ConstQ",
            "value": round(value, 1),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic (SURVEY Appendix-B generator, seed 1234; 2 distinct pictures tiled over the batch)",
            "config": {"workload": "BASELINE cfg2: UHD-1 3840x2160 4:2:2 10-bit HQ_ConstQ DD97 depth 4, -u 1 -a 2 -q 16 -S 2",
                       "pictures_per_gpu_per_step": B, "coded_bytes_per_picture": coded, "streams": args.streams,
                       "parallelism": f"frame-parallel x{world}, no collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": dom, "kernel_avg_ms": round(dom_ms / dom_launches, 4),
                         "algorithmic_bytes_per_launch": int(alg_dir * per_launch),
                         "path_achieved_GBs": round(path_achieved, 1),
                         "path_frac": round(path_achieved / HBM_PEAK_GBS, 4),
                         "kernel_ms_per_step": {k: round(v[1] / args.steps, 4) for k, v in sorted(kern.items())}},
            "value_without_kernel_events": round(total_px / dt_noev / 1e6, 1),
            "parity_checked": ok,
        }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the oracle (port of the reference algorithm, 1 thread) on a bounded sample of the same workload
        from vc2lib import load_oracle, make_params
        oracle = load_oracle()
        p = make_params(W, H, CFMT, BITS, KERNEL, DEPTH, U, A, q=Q, scalar=SCALAR)
        nfr = max(1, args.cpu_frames)
        sample = (raw * ((nfr + distinct - 1) // distinct))[:nfr * rb]
        c0 = time.perf_counter()
        stream = oracle.encode_stream(p, sample, nfr)
        dec, n = oracle.decode_stream(p, stream, nfr)
        cdt = time.perf_counter() - c0
        assert n == nfr and dec[:distinct * rb] == d_out[:min(nfr, distinct) * rb].cpu().numpy().tobytes()[:len(dec[:distinct * rb])]
        out["cpu_baseline"] = {"value": round(pixels * nfr / cdt / 1e6, 3), "unit": "Mpixels/s", "cores": 1,
                               "kind": "port",
                               "sample": f"{nfr} UHD-1 pictures of the same workload, encode+decode through oracle/ (1 thread), {cdt:.1f} s"}
    elif rank == 0:
        out["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
