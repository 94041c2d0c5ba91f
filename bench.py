#!/usr/bin/env python3
"""Headline benchmark: BASELINE.json configs[1]+[2] -- a batch of 1024 synthetic 1024x1024 RGBA
images (SURVEY.md Appendix C class "mixed", seed 1) encoded to .sqoa streams and decoded back,
device-resident, on N MI355X (one process per GPU, images sharded, no collective in the data path).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--images 1024] [--size 1024]

One step = encode the rank's whole batch, then decode the streams just produced.  `value` is
whole-job Mpixels/s over the K timed steps (pixels that went through encode AND decode, counted
once).  The JSON line also carries the roofline of the dominant kernel (encode, HIP events on the
kernel's own stream) and the reference seqoia.h timed on the host cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def cpu_baseline(width, height, channels, budget_s=12.0):
    """Reference codec (oracle/_ref when present, else the restatement) on the host cores, on a
    bounded sample of the same workload: as many images as fit ~budget_s, every core busy."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor

    from oracle import pyoracle as po

    codec = po.best()
    cores = len(os.sched_getaffinity(0))
    per_thread = 8  # 8 x (encode + decode) of a 1 Mpixel image ~ 0.15 s per thread-iteration: ~20-40 s of CPU work in total
    imgs = [po.synth(width, height, channels, "mixed", 1, i) for i in range(min(cores, 16))]

    def work(t):
        d = po.Desc(width, height, channels, 0, 0)
        n = C.c_int(0)
        dd = po.Desc()
        px = imgs[t % len(imgs)]
        enc = codec.encode(px, width, height, channels, 0)
        t_enc = t_dec = 0.0
        for _ in range(per_thread):
            a = time.perf_counter()
            codec.encode_raw(px.ctypes.data, C.byref(d), C.byref(n))  # malloc+free inside, sqoabench.c:527-538
            b = time.perf_counter()
            codec.decode_raw(enc, len(enc), C.byref(dd), 4)           # sqoabench.c:490-494
            c = time.perf_counter()
            t_enc += b - a
            t_dec += c - b
        return t_enc, t_dec

    work(0)  # warm-up run, discarded (sqoabench.c:394-406)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        parts = list(ex.map(work, range(cores)))
    wall = time.perf_counter() - t0
    px_total = cores * per_thread * width * height
    return {
        "value": round(px_total / wall / 1e6, 1),
        "unit": "Mpixels/s",
        "cores": cores,
        "kind": codec.kind,
        "sample": f"{cores * per_thread} images {width}x{height}x{channels} (class mixed), encode+decode each, {cores} threads",
        "encode_mpps_per_core": round(per_thread * width * height / (sum(p[0] for p in parts) / cores) / 1e6, 1),
        "decode_mpps_per_core": round(per_thread * width * height / (sum(p[1] for p in parts) / cores) / 1e6, 1),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=20, help="untimed steps; the first ~50 ms after an idle period run at lower clocks (tools/time_warm.py)")
    ap.add_argument("--images", type=int, default=1024, help="images per GPU")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--graph", action="store_true", help="capture the step once into a HIP graph and replay it (same time: launch overhead is not what bounds the step)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for the timing barrier/MAX (nccl = RCCL; gloo for rehearsals)")
    args = ap.parse_args()

    import torch

    import seqoia_amd as sq

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device: the product has no CPU path")
    local = local % max(torch.cuda.device_count(), 1)  # rehearsal on a 1-GPU box: ranks share the device
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)

    n, s, ch = args.images, args.size, 4
    dev = torch.device("cuda", local)
    px = torch.empty((n, s, s, ch), dtype=torch.uint8, device=dev)
    sq.synth(px, "mixed", seed=1, img0=rank * n)  # every rank owns a disjoint slice of the image list
    enc = sq.EncodePlan(px, qoi_compat=0)
    enc.launch()
    res = enc.results()
    assert (res["status"] == 0).all(), "encode failed"
    enc_bytes = int(res["len"].sum())
    dec = None
    try:
        dec = sq.DecodePlan(enc.out, res["len"], s, s, 4)
        dec.launch()
        dres = dec.results()
        assert (dres["status"] == 0).all(), "decode failed"
        assert torch.equal(dec.out, px), "round trip mismatch"  # sqoabench.c:448-455 on the whole batch
    except sq.SqoaError as e:
        if rank == 0:
            print(f"[bench] decode leg unavailable: {e}", file=sys.stderr)
        dec = None

    work_stream = torch.cuda.Stream(device=dev)  # both legs in order on one stream
    stream = work_stream.cuda_stream

    def step():
        enc.launch(stream)
        if dec is not None:
            dec.launch(stream)

    # plan launches allocate nothing and never touch the host, so the step is captured once into a HIP graph and replayed
    graph = None
    if args.graph:
        try:
            step()
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=work_stream):
                step()
            run = graph.replay
        except Exception as e:  # noqa: BLE001 -- the direct path is always available
            if rank == 0:
                print(f"[bench] HIP graph capture unavailable ({e}); launching directly", file=sys.stderr)
            graph = None
    if graph is None:
        run = step

    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # per-kernel durations with HIP events on the stream the kernels run on (the plan's own)
    k = max(args.steps, 5)
    enc_ms = enc.time(k) / k
    dec_ms = dec.time(k) / k if dec is not None else None

    pixels = n * s * s
    total_px = pixels * world
    read_bytes = pixels * ch
    out = {
        "metric": "Mpixels/s encode+decode, 1024-image RGBA batch; % HBM roofline",
        "value": round(total_px * args.steps / dt / 1e6, 1),
        "unit": "Mpixels/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {
            "workload": f"{n} x {s}x{s} RGBA (class mixed, seed 1) per GPU: sqoa_encode qoi_compat=0 then sqoa_decode to 4 channels, device-resident",
            "images_per_gpu": n,
            "encoded_bytes_per_gpu": enc_bytes,
            "rate": round(enc_bytes / (pixels * ch), 4),
            "decode_leg": dec is not None,
            "launch": "hip graph replay" if graph is not None else "direct",
            "encode_mpps_per_gpu": round(pixels / enc_ms / 1e3, 1),
            "decode_mpps_per_gpu": round(pixels / dec_ms / 1e3, 1) if dec_ms else None,
        },
        "roofline": {
            "kernel": "sqoa::encode_kernel",
            "bound": "hbm",
            "achieved": round(read_bytes / enc_ms / 1e6, 1),   # algorithmic READ bytes (4 B/px) / kernel time
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(read_bytes / enc_ms / 1e6 / HBM_PEAK_GBS, 4),
            "traffic": None,
            "achieved_read_plus_write": round((read_bytes + enc_bytes) / enc_ms / 1e6, 1),
            "kernel_ms": round(enc_ms, 4),
        },
    }
    pmc = os.path.join(ROOT, "profiles", "pmc_hbm_latest.json")
    if os.path.exists(pmc) and n == 1024 and s == 1024:
        # HBM bytes per launch measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on this
        # same workload, corrected per MI355X_MICROARCH.md (gfx950 FETCH_SIZE counts half of streamed reads)
        k_ = json.load(open(pmc))["kernels"].get("encode")
        if k_:
            out["roofline"]["traffic"] = int((2 * k_["FETCH_SIZE_KB"] + k_["WRITE_SIZE_KB"]) * 1024)
            out["roofline"]["traffic_source"] = "profiles/pmc_hbm_latest.json"
    if dec_ms:
        out["roofline"]["decode"] = {
            "kernel": "sqoa::decode_kernel",
            "achieved": round((enc_bytes + pixels * 4) / dec_ms / 1e6, 1),
            "frac": round((enc_bytes + pixels * 4) / dec_ms / 1e6 / HBM_PEAK_GBS, 4),
            "kernel_ms": round(dec_ms, 4),
        }
    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(s, s, ch)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
