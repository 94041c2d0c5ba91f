#!/usr/bin/env python3
"""Headline benchmark: BASELINE.json configs[1]+[2] -- a batch of 1024 synthetic 1024x1024 RGBA images (SURVEY.md
Appendix C class "mixed", seed 1) encoded to .sqoa streams and decoded back, device-resident, on N MI355X: one process
per GPU, the image list sharded over the ranks, no collective in the data path (images are independent work items; the
per-image loop this replaces is sqoabench.c:575-628).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong] [--config headline|c5]

* ``--gpus N`` (N > 1) without a launcher: this process touches no GPU API; it starts N rank processes (RANK /
  LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment), relays rank 0's JSON line and
  exits non-zero if any rank failed.  Under ``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`` the
  environment is already there and the process is a rank.
* ``--scaling weak`` (default): 1024 images per GPU.  ``strong``: 1024 images in total, contiguous blocks per rank.
* ``--config c5``: BASELINE.json configs[4], the qoi-suite-shaped set of 65 536 mixed-size images
  (seqoia_amd/suite.py), longest-processing-time-first over the ranks on w*h; strong scaling by construction.

One step = encode the rank's whole shard, then decode the streams just produced.  `value` is whole-job Mpixels/s over
the K timed steps (pixels that went through encode AND decode, counted once; MAX over ranks of the time).  The JSON line
also carries the roofline of both kernels (HIP events on the stream they run on), the secondary configurations
(QOI-compat on the same batch, RGB icon sets) and the reference seqoia.h timed on the host cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
HBM_COPY_GBS = 6290.0  # the same guide's measured streaming copy rate: what read + write traffic can reach in practice
METRIC = "Mpixels/s encode+decode, 1024-image RGBA batch; % HBM roofline"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=20, help="untimed steps; the first ~50 ms after an idle period run at lower clocks (tools/time_warm.py)")
    ap.add_argument("--images", type=int, default=None, help="headline: images per GPU (weak) / in total (strong), default 1024; c5: images in total, default 65536")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--config", choices=("headline", "c5"), default="headline")
    ap.add_argument("--no-cpu", action="store_true", help="skip the host-core reference leg")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary configurations (QOI-compat, icon sets)")
    ap.add_argument("--graph", action="store_true", help="capture the step once into a HIP graph and replay it (same time: launch overhead is not what bounds the step)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for the timing barrier/MAX (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--rehearsal", action="store_true", help="allow more ranks than visible devices (ranks share GPUs); the JSON line is marked as a rehearsal")
    ap.add_argument("--dry-run", action="store_true", help="ranks only rendezvous and report their shard (no GPU): launcher / sharding rehearsal")
    return ap.parse_args(argv)


# ---- launcher: --gpus N without torchrun -----------------------------------------------------------------------------


def launch_ranks(n, args, argv, limit_s=1500.0):
    """Start n rank processes of this script and relay rank 0's stdout.  Nothing here may touch a GPU API: a process
    that has initialised HIP must not fork/exec the ranks (and this parent never imports torch).  The parent runs the
    reference leg on the host cores once, BEFORE the ranks exist (they would compete for the cores, and rank 0 hosts the
    rendezvous store the others wait for), and hands it to rank 0 in a file.  All children are polled: when one fails or the
    time limit passes, the others are terminated (they would sit in the rendezvous or a barrier otherwise)."""
    import tempfile

    cpu_file = None
    if not (args.no_cpu or args.dry_run):
        cpu = cpu_baseline(args.size, args.size, 4)
        fd, cpu_file = tempfile.mkstemp(prefix="sqoa_bench_cpu_", suffix=".json")
        with os.fdopen(fd, "w") as f:
            json.dump(cpu, f)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs, out0 = [], tempfile.TemporaryFile()
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        if cpu_file:
            env["SQOA_BENCH_CPU_JSON"] = cpu_file
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    t_end, bad = time.time() + limit_s, []
    try:
        while True:
            rcs = [p.poll() for p in procs]
            bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
            if bad or all(rc == 0 for rc in rcs):
                break
            if time.time() > t_end:
                bad = [(r, "time limit") for r, rc in enumerate(rcs) if rc is None]
                break
            time.sleep(0.05)
    finally:
        live = [p for p in procs if p.poll() is None]
        for p in live:
            p.terminate()
        t_kill = time.time() + 5.0
        for p in live:
            try:
                p.wait(timeout=max(0.1, t_kill - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        if cpu_file:
            os.unlink(cpu_file)
    out0.seek(0)
    sys.stdout.write(out0.read().decode())
    sys.stdout.flush()
    if bad:
        sys.exit(f"bench.py: ranks failed (rank, exit code): {bad}; the other ranks were stopped")


# ---- the reference on the host cores -----------------------------------------------------------------------------------


def cpu_quota():
    """CPUs' worth of run time the container may use (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def physical_cores():
    """Physical cores this process may run on: distinct (package, core) pairs among the allowed logical CPUs."""
    allowed = os.sched_getaffinity(0)
    seen = set()
    for cpu in allowed:
        try:
            base = f"/sys/devices/system/cpu/cpu{cpu}/topology/"
            seen.add((open(base + "physical_package_id").read().strip(), open(base + "core_id").read().strip()))
        except OSError:
            seen.add(("?", cpu))
    return max(len(seen), 1), len(allowed)


def cpu_worker(arg):
    """`bench.py --cpu-worker t,per,width,height,channels,start` -- one host core of the reference leg: synthesise `per`
    images of its own, wait for the common start time, then encode + free and decode + free each (sqoabench.c:527-538,
    :490-494); prints one JSON line.  A process of its own: see cpu_baseline."""
    import ctypes as C

    from oracle import pyoracle as po

    t, per, width, height, channels = (int(x) for x in arg.split(",")[:5])
    start = float(arg.split(",")[5])
    rounds = int(arg.split(",")[6]) if len(arg.split(",")) > 6 else 1
    codec = po.best()
    dl, nl, ddl = po.Desc(width, height, channels, 0, 0), C.c_int(0), po.Desc()
    imgs = [po.synth(width, height, channels, "mixed", 1, t * per + k) for k in range(per)]
    encs = [codec.encode(p, width, height, channels, 0) for p in imgs]
    te = td = 0.0
    while time.time() < start:  # every core enters the timed loop together
        time.sleep(0.001)
    t0 = time.time()
    for _ in range(rounds):
        for p, e in zip(imgs, encs):
            a = time.perf_counter()
            codec.encode_raw(p.ctypes.data, C.byref(dl), C.byref(nl))
            b = time.perf_counter()
            codec.decode_raw(e, len(e), C.byref(ddl), 4)
            c = time.perf_counter()
            te += b - a
            td += c - b
    print(json.dumps([te, td, t0, time.time()]))


def cpu_baseline(width, height, channels):
    """Reference codec (oracle/_ref when present, else the restatement) on the host cores, on a bounded sample of the same
    workload: (i) one thread, sqoabench protocol -- one discarded warm-up run, mean of N runs,
    malloc + free inside the timed region (sqoabench.c:394-406, :490-494, :527-538); (ii) T workers, T = physical cores,
    every worker encoding + decoding images of its own.  Runs BEFORE this process touches the GPU (it starts processes)."""
    import ctypes as C

    from oracle import pyoracle as po

    codec = po.best()
    cores, logical = physical_cores()
    quota = cpu_quota()
    if quota is not None:  # more workers than the container's CPU share only time-slice each other
        cores = max(1, min(cores, int(quota)))
    px_img = width * height

    # (i) one thread, the protocol of BENCHMARK_FN (run before and after the multi-process leg, the faster of the two
    # kept: the first measurement of a fresh process can fall into the container's CPU throttling)
    runs = 5
    px0 = po.synth(width, height, channels, "mixed", 1, 0)
    enc0 = codec.encode(px0, width, height, channels, 0)

    def one_thread():
        d, n, dd = po.Desc(width, height, channels, 0, 0), C.c_int(0), po.Desc()
        t_enc = t_dec = 0.0
        for i in range(-1, runs):  # run -1 is the warm-up
            a = time.perf_counter()
            codec.decode_raw(enc0, len(enc0), C.byref(dd), 4)
            b = time.perf_counter()
            codec.encode_raw(px0.ctypes.data, C.byref(d), C.byref(n))
            c = time.perf_counter()
            if i >= 0:
                t_dec += b - a
                t_enc += c - b
        return {"encode_mpps": round(px_img * runs / t_enc / 1e6, 1), "decode_mpps": round(px_img * runs / t_dec / 1e6, 1),
                "runs": runs, "protocol": "sqoabench BENCHMARK_FN: 1 warm-up discarded, mean of runs, malloc+free timed"}

    single = one_thread()

    # (ii) all physical cores, distinct images per worker.  Workers are PROCESSES: the reference mallocs its worst-case
    # buffer in every call (seqoia.h:492, :710), and threads of one process serialise on the address-space lock for those
    # mmap / munmap / page-fault cycles (128 threads: 36 / 44 Mpixels/s per core against 332 / 388 alone).  Plain child
    # processes with a common start time, each waited for with a time-out: nothing here can hang the benchmark.
    # images per worker and passes over them: encode + decode of a 1 Mpixel image ~6 ms alone, so ~0.8 s of timed work per worker,
    # ~12 core-seconds on 16 cores (8 images once, 50 ms, was at the mercy of the workers' start skew: 1 590 .. 2 700 Mpixels/s)
    per_thread, rounds = 8, max(1, int(16 * (1 << 20) / px_img))
    start = time.time() + 2.0 + 0.03 * cores  # interpreter start + synthesis of every worker's images fit in front of it
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", f"{t},{per_thread},{width},{height},{channels},{start},{rounds}"],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL) for t in range(cores)]
    parts = []
    for pr in procs:
        try:
            o = pr.communicate(timeout=120)[0]
            if pr.returncode == 0:
                parts.append(json.loads(o.decode().strip().splitlines()[-1]))
        except Exception:  # noqa: BLE001 -- a worker that hangs or dies is dropped from the sample, never waited for
            pr.kill()
    if not parts:
        return {"value": None, "unit": "Mpixels/s", "cores": 0, "kind": codec.kind, "sample": "reference workers failed", "single_thread": single}
    again = one_thread()
    if 1.0 / again["encode_mpps"] + 1.0 / again["decode_mpps"] < 1.0 / single["encode_mpps"] + 1.0 / single["decode_mpps"]:
        single = again
    late = sum(1 for p in parts if p[2] > start + 0.05)
    cores_used = len(parts)
    busy = max(p[3] for p in parts) - min(p[2] for p in parts)  # from the common start to the last worker's last call (time.time: one clock for all)
    px_total = cores_used * per_thread * rounds * px_img
    return {
        "value": round(px_total / busy / 1e6, 1),
        "unit": "Mpixels/s",
        "cores": cores_used,
        "workers_started_late": late,
        "cpu_quota": quota,
        "logical_cpus": logical,
        "kind": codec.kind,
        "sample": f"{cores_used * per_thread} distinct images {width}x{height}x{channels} (class mixed, seed 1), encode+decode each, {rounds} passes, {cores_used} worker processes (one per physical core)",
        "encode_mpps_per_core_under_load": round(per_thread * rounds * px_img / (sum(p[0] for p in parts) / cores_used) / 1e6, 1),
        "decode_mpps_per_core_under_load": round(per_thread * rounds * px_img / (sum(p[1] for p in parts) / cores_used) / 1e6, 1),
        "single_thread": single,
    }


# ---- workloads -------------------------------------------------------------------------------------------------------


class Uniform:
    """n same-shaped images in one tensor: EncodePlan -> DecodePlan."""

    def __init__(self, sq, torch, dev, n, w, h, ch, cls, img0, qoi=0, check=True, blend=None):
        self.n, self.pixels = n, n * w * h
        self.read_bytes = self.pixels * ch
        self.px = torch.empty((n, h, w, ch), dtype=torch.uint8, device=dev)
        sq.synth(self.px, cls, seed=1, img0=img0)
        if blend:   # (the first `blend[0]` images of another class: a list of mixed content)
            sq.synth(self.px[: blend[0]], blend[1], seed=1, img0=img0)
        self.enc = sq.EncodePlan(self.px, qoi_compat=qoi)
        self.enc.launch()
        res = self.enc.results()
        assert (res["status"] == 0).all(), "encode failed"
        self.lens = res["len"].astype("int64")
        self.enc_bytes = int(self.lens.sum())
        self.dec = sq.DecodePlan(self.enc.out, res["len"], w, h, ch)
        self.dec.launch()
        dres = self.dec.results()
        assert (dres["status"] == 0).all(), "decode failed"
        if check:
            assert torch.equal(self.dec.out, self.px), "round trip mismatch"  # sqoabench.c:448-455 on the whole batch
        self.write_bytes = self.pixels * ch

    def step(self, stream):
        self.enc.launch(stream)
        self.dec.launch(stream)

    def decoded_ok(self, torch):
        return bool(torch.equal(self.dec.out, self.px))

    def close(self):
        self.dec.close()
        self.enc.close()


class Suite:
    """A rank's shard of the suite-shaped image list: EncodeList -> DecodeList (BASELINE.json configs[4])."""

    def __init__(self, sq, torch, dev, shard):
        self.n = len(shard)
        self.idx = [im.index for im in shard]
        sizes = [im.pixels * im.channels for im in shard]
        offs = [0]
        for s in sizes:
            offs.append(offs[-1] + ((s + 255) & ~255))
        self.px = torch.empty((max(offs[-1], 1),), dtype=torch.uint8, device=dev)
        images = []
        for im, o, s in zip(shard, offs, sizes):
            t = self.px[o : o + s]
            sq.synth(t.view(1, im.height, im.width, im.channels), im.cls, seed=1, img0=im.index)
            images.append((t, sq.Desc(im.width, im.height, im.channels, 0, 0)))
        self.pixels = sum(im.pixels for im in shard)
        self.read_bytes = sum(sizes)
        self.write_bytes = self.read_bytes
        self.enc = sq.EncodeList(images)
        self.enc.launch()
        res = self.enc.results()
        assert (res["status"] == 0).all(), "encode failed"
        self.lens = res["len"].astype("int64")
        self.enc_bytes = int(self.lens.sum())
        base = self.enc.out.data_ptr()
        self.dec = sq.DecodeList([(base + o, int(l), im.width, im.height, im.channels)
                                  for im, o, l in zip(shard, self.enc.offsets, res["len"])], device=dev.index or 0)
        self.dec.launch()
        dres = self.dec.results()
        assert (dres["status"] == 0).all(), "decode failed"
        self._offs, self._sizes = offs, sizes
        for i, (o, s) in enumerate(zip(offs, sizes)):  # round trip of every image (sqoabench.c:448-455)
            assert torch.equal(self.dec.image(i), self.px[o : o + s]), f"round trip mismatch, image {self.idx[i]}"

    def decoded_ok(self, torch):
        return all(bool(torch.equal(self.dec.image(i), self.px[o : o + s])) for i, (o, s) in enumerate(zip(self._offs, self._sizes)))

    def step(self, stream):
        self.enc.launch(stream)
        self.dec.launch(stream)

    def close(self):
        self.dec.close()
        self.enc.close()


def board_power(pci=None):
    """(watts drawn, the cap, shader clock in MHz) from sysfs hwmon of the card at PCI address `pci` ("0000:c5:00.0": this rank's own device) --
    or, where that cannot be told, of the busiest card the host shows -- or None.  Both kernels run AT the board's power cap on MI355X
    (1400 W: profiles/r04_power.txt).  One read, a few microseconds; called while untimed steps run on the device."""
    import glob

    best = None
    for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        try:
            w = int(open(os.path.join(h, "power1_input")).read()) / 1e6
            cap = int(open(os.path.join(h, "power1_cap")).read()) / 1e6
            mhz = int(open(os.path.join(h, "freq1_input")).read()) / 1e6
        except (OSError, ValueError):
            continue
        row = (round(w, 1), round(cap, 1), round(mhz))
        if pci and os.path.basename(os.path.realpath(os.path.join(h, "..", ".."))).lower() == pci.lower():
            return row + ("own",)
        if best is None or w > best[0]:
            best = row
    return best + ("busiest",) if best else None


def in_step_times(torch, wl, work_stream, k):
    """(encode ms, decode ms): each plan's launch INSIDE the encode -> decode step, medians over k steps, events on the stream the step runs on.
    (A kernel behind the other one is not the kernel in a loop of its own: the board's power budget is shared -- profiles/r04_a_step_parts.txt.)"""
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * k + 1)]
    stream = work_stream.cuda_stream
    ev[0].record(work_stream)
    for i in range(k):
        wl.enc.launch(stream)
        ev[2 * i + 1].record(work_stream)
        wl.dec.launch(stream)
        ev[2 * i + 2].record(work_stream)
    torch.cuda.synchronize()
    e = sorted(ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(k))
    d = sorted(ev[2 * i + 1].elapsed_time(ev[2 * i + 2]) for i in range(k))
    return e[k // 2], d[k // 2]


def kernel_times(wl, k):
    """(encode ms, decode ms) per launch, HIP events on the plans' own stream."""
    return wl.enc.time(k) / k, wl.dec.time(k) / k


def secondary(sq, torch, dev, headline, k=5):
    """Configurations BASELINE.json names besides the headline, timed outside ms_per_step with HIP events over k launches
    after a warm-up launch each: QOI-compat (qoi_compat=1, the 64-entry running colour hash: seqoia.h:563-582, :753-760,
    :785-787) on the headline batch, and configs[3], RUN-heavy 3-channel icon sets, in both wire formats.  Read-roofline
    fraction = channels B/px / encode time / 8 TB/s; decode fraction = (stream + pixel bytes) / time / 8 TB/s."""
    out = {}

    def warm(plan, ms=60.0):
        # (the first ~50 ms of launches after the chip has idled run at lower clocks -- tools/time_warm.py --, and between two rows it
        # idles while the next workload is synthesised: two warm-up launches left these rows 8 % behind what tools/time_*.py and the
        # headline's spin-up steps measure for the same plans)
        t = 0.0
        while t < ms:
            t += plan.time(3)

    def row(name, wl, four=False):
        warm(wl.enc)
        e = wl.enc.time(k) / k      # (each kernel behind its OWN warm-up, like the headline's plan_launch_ms: behind the other kernel's it is
        warm(wl.dec)                #  that kernel's clocks and temperature that are measured -- the QOI-compat encode 2.71 against 2.47 ms)
        # (what the timed launches leave is looked at BEFORE results() is asked: a launch that decoded nothing -- round 5's lazy skip did that to
        # the speculative QOI-compat route, and results() repaired it out of sight -- must not yield a time)
        wl.dec.out.zero_()
        torch.cuda.synchronize()
        d = wl.dec.time(k) / k
        torch.cuda.synchronize()
        assert wl.decoded_ok(torch), f"{name}: the timed decode launches did not reproduce the pixels"
        assert (wl.dec.results()["status"] == 0).all(), name
        out[name] = {
            "encode_ms": round(e, 4), "decode_ms": round(d, 4),
            "encode_mpps": round(wl.pixels / e / 1e3, 1), "decode_mpps": round(wl.pixels / d / 1e3, 1),
            "rate": round(wl.enc_bytes / wl.read_bytes, 4),
            "encode_read_roofline_frac": round(wl.read_bytes / e / 1e6 / HBM_PEAK_GBS, 4),
            "decode_roofline_frac": round((wl.enc_bytes + wl.write_bytes) / d / 1e6 / HBM_PEAK_GBS, 4),
        }
        if four:
            # sqoabench's TIMED decode asks for 4 channels whatever the file holds (sqoabench.c:490-494; its verify decode, :448-455,
            # asks for the file's own count: `decode_ms` above): the same streams to RGBA, checked against the pixels they came from
            nn, hh, ww, cc = wl.px.shape
            d4 = sq.DecodePlan(wl.enc.out, wl.lens, ww, hh, 4)
            d4.launch()
            assert (d4.results()["status"] == 0).all() and torch.equal(d4.out[..., :cc], wl.px) and bool((d4.out[..., 3] == 255).all()), "4-channel decode mismatch"
            warm(d4)
            d4.out.zero_()
            torch.cuda.synchronize()
            t4 = d4.time(k) / k
            torch.cuda.synchronize()
            assert torch.equal(d4.out[..., :cc], wl.px) and (d4.results()["status"] == 0).all(), f"{name}: the timed 4-channel decode launches did not reproduce the pixels"
            out[name].update({"decode_to_4_channels_ms": round(t4, 4), "decode_to_4_channels_mpps": round(wl.pixels / t4 / 1e3, 1),
                              "decode_to_4_channels_roofline_frac": round((wl.enc_bytes + wl.pixels * 4) / t4 / 1e6 / HBM_PEAK_GBS, 4)})
            d4.close()
        wl.close()

    h = headline
    n, hh, ww, ch = h.px.shape
    wl = Uniform.__new__(Uniform)  # QOI-compat on the very same pixels
    wl.n, wl.pixels, wl.read_bytes, wl.write_bytes, wl.px = h.n, h.pixels, h.read_bytes, h.write_bytes, h.px
    wl.enc = sq.EncodePlan(h.px, qoi_compat=1)
    wl.enc.launch()
    res = wl.enc.results()
    assert (res["status"] == 0).all()
    wl.enc_bytes = int(res["len"].astype("int64").sum())
    wl.dec = sq.DecodePlan(wl.enc.out, res["len"], ww, hh, ch)
    wl.dec.launch()
    assert (wl.dec.results()["status"] == 0).all() and torch.equal(wl.dec.out, h.px), "QOI-compat round trip mismatch"
    row(f"qoi_compat=1, {n} x {ww}x{hh} RGBA (the headline batch)", wl)
    # class photo (LUMA ops throughout: the content the balanced decoder's table look-ups like least), as RGBA and -- what a
    # photograph usually is -- as RGB decoded to its own three channels and, as sqoabench's timed decode asks (sqoabench.c:490-494), to four
    row(f"class photo {n} x {ww}x{hh} RGBA, qoi_compat=0", Uniform(sq, torch, dev, n, ww, hh, 4, "photo", 0))
    row(f"class photo {n} x {ww}x{hh} RGB, qoi_compat=0", Uniform(sq, torch, dev, n, ww, hh, 3, "photo", 0), four=True)
    # a list of BOTH kinds (a quarter photographs): the plan splits the photographs off into a launch of their own beside the balanced kernel's
    row(f"{n // 4} class photo among {n - n // 4} class mixed, {ww}x{hh} RGBA, qoi_compat=0", Uniform(sq, torch, dev, n, ww, hh, 4, "mixed", 0, blend=(n // 4, "photo")))
    for size in (512, 64):
        for qoi in (0, 1):
            row(f"C4 icons 4096 x {size}x{size} RGB, qoi_compat={qoi}", Uniform(sq, torch, dev, 4096, size, size, 3, "icon", 0, qoi=qoi), four=True)
    # few large images: the plan cuts them into strips on the device (encode) / decodes them in three chip-wide passes
    # (DESIGN.md 3.1b, 3.2b); one workgroup per image would be one compute unit's speed (0.57 / 2.0 ms per Mpixel)
    row("one 3840x2160 RGBA image alone, qoi_compat=0", Uniform(sq, torch, dev, 1, 3840, 2160, 4, "mixed", 0))
    row("16 x 1024x1024 RGBA, qoi_compat=0", Uniform(sq, torch, dev, 16, 1024, 1024, 4, "mixed", 0))
    # what `sqoaconv x.png y.qoi` makes (sqoaconv.c:84-90): the encoder cuts it into strips and hands the colour index from strip to
    # strip (sqoa_strips.hip); the decoder's INDEX ops make every pixel depend on the values of all earlier ones: one workgroup
    row("one 3840x2160 RGBA image alone, qoi_compat=1", Uniform(sq, torch, dev, 1, 3840, 2160, 4, "mixed", 0, qoi=1))
    # BASELINE.json configs[4] as rank 0 of 8 sees it: its LPT shard of the 65 536 suite-shaped images in one plan per direction
    from seqoia_amd import suite

    row("C5 rank 0 of 8: LPT shard of the 65536 suite-shaped images (8192 images, 64x64 .. 3840x2160, 3 and 4 channels), qoi_compat=0",
        Suite(sq, torch, dev, suite.shard(suite.C5_IMAGES, 8, 0)))
    out["legacy sqoa_encode / sqoa_decode on one 1024x1024 RGBA host image"] = legacy_calls(sq, headline)
    return out


def legacy_calls(sq, headline, runs=10):
    """The "sqoa:" row a relinked sqoabench prints: the four-symbol API on HOST buffers, one image per call, protocol of
    BENCHMARK_FN (sqoabench.c:394-406: one discarded warm-up run, mean of `runs`; malloc by the library and free by the caller
    inside the timed region, sqoabench.c:490-494 decode to 4 channels, :527-538 encode).  PCIe copies included by nature."""
    import ctypes as C

    from seqoia_amd import _lib

    lib = _lib.load()
    px = headline.px[0].cpu().numpy().copy()
    h, w, ch = px.shape
    d, n, dd = sq.Desc(w, h, ch, 0, 0), C.c_int(0), sq.Desc()
    p = lib.sqoa_encode(px.ctypes.data, C.byref(d), C.byref(n))
    assert p, "legacy sqoa_encode failed"
    enc = C.string_at(p, n.value)
    _lib.c_free(p)
    t_enc = t_dec = 0.0
    for i in range(-1, runs):
        a = time.perf_counter()
        q = lib.sqoa_decode(enc, len(enc), C.byref(dd), 4)
        _lib.c_free(q)
        b = time.perf_counter()
        q = lib.sqoa_encode(px.ctypes.data, C.byref(d), C.byref(n))
        _lib.c_free(q)
        c = time.perf_counter()
        if i >= 0:
            t_dec += b - a
            t_enc += c - b
    return {"encode_ms": round(t_enc / runs * 1e3, 4), "decode_ms": round(t_dec / runs * 1e3, 4),
            "encode_mpps": round(w * h * runs / t_enc / 1e6, 1), "decode_mpps": round(w * h * runs / t_dec / 1e6, 1),
            "protocol": "sqoabench BENCHMARK_FN: 1 warm-up discarded, mean of 10 runs, malloc+free timed, host buffers (H2D + kernels + D2H per call)"}


def counters_from_profile(kernel, images, size):
    """What the committed rocprofv3 PMC summary says about a headline kernel -- HBM bytes per launch, instructions retired per 64
    pixels by kind, LDS bank-conflict cycles per active LDS cycle, the clock of the counter pass (SQ_BUSY_CYCLES / shader engines /
    the kernel's duration in that same pass: an estimate of the timed run's), the kernel's durations by phase of the kernel-trace pass -- only when it was taken from THIS kernel source (the summary
    records the hash of seqoia_amd/csrc) on the same image shape; the bytes are scaled to this rank's number of images (the
    images are independent and equal in size: traffic is per image).  {} otherwise."""
    import hashlib

    pmc = os.path.join(ROOT, "profiles", "pmc_hbm_latest.json")
    if not os.path.exists(pmc) or images <= 0:
        return {}
    d = json.load(open(pmc))
    hsh = hashlib.sha256()
    src = os.path.join(ROOT, "seqoia_amd", "csrc")
    for f in sorted(os.listdir(src)):
        hsh.update(open(os.path.join(src, f), "rb").read())
    if d.get("csrc_sha256") != hsh.hexdigest() or d.get("size", 1024) != size:
        return {}
    k = d["kernels"].get(kernel)
    if not k:
        return {}
    # gfx950: FETCH_SIZE counts half of a wide streaming read (MI355X_MICROARCH.md, HBM); KB
    per_launch = (2 * k["FETCH_SIZE_KB"] + k["WRITE_SIZE_KB"]) * 1024
    out = {"traffic": int(per_launch * images / d.get("images", 1024)), "source": "profiles/pmc_hbm_latest.json"}
    for f in ("insts_per_64px", "lds_conflict_cycles_per_active_cycle", "clock_ghz_pmc_pass", "rocprof_avg_ms", "rocprof_min_ms", "rocprof_in_step_ms", "rocprof_alone_ms", "kernel"):
        if f in k:
            out[f] = k[f]
    return out


# ---- one rank --------------------------------------------------------------------------------------------------------


def run_rank(args):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: using {world} ranks", file=sys.stderr)

    from seqoia_amd import shard as sh
    from seqoia_amd import suite

    c5 = args.config == "c5"
    scaling = "strong" if c5 else args.scaling
    n_total = args.images if args.images is not None else (suite.C5_IMAGES if c5 else 1024)
    if c5:
        mine = suite.shard(n_total, world, rank)
    elif scaling == "strong":
        mine = sh.contiguous(n_total, world, rank)
    else:
        mine = range(rank * n_total, (rank + 1) * n_total)

    if args.dry_run:  # rendezvous + shard report only: rehearses the launcher and the partition without a GPU
        import torch.distributed as dist

        if os.environ.get("SQOA_BENCH_FAULT_RANK") == str(rank):  # fault injection for the launcher's tests: this rank dies before the rendezvous
            sys.exit(3)
        if world > 1:
            dist.init_process_group("gloo")
        cnt = len(mine)
        allv = sh.gather_lengths([im.index for im in mine] if c5 else list(mine), [1] * cnt, n_total if (c5 or scaling == "strong") else n_total * world)
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "scaling": scaling, "config": args.config, "images_rank0": cnt, "images_total": int(sum(allv))}))
        if world > 1:
            dist.destroy_process_group()
        return

    # The reference on the host cores, in the same run as every point of the 1..N curve (north-star).  Rank 0, before this
    # process touches any GPU API (it starts worker processes) and before the rendezvous, during which the other ranks only
    # wait; under bench.py's own launcher the parent has already run it (launch_ranks) and left it in a file.
    cpu = None
    if rank == 0 and not args.no_cpu:
        handed = os.environ.get("SQOA_BENCH_CPU_JSON")
        if handed and os.path.exists(handed):
            cpu = json.load(open(handed))
        else:
            cpu = cpu_baseline(args.size, args.size, 4)

    import torch

    import seqoia_amd as sq

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device: the product has no CPU path")
    ndev = torch.cuda.device_count()
    if world > ndev and not args.rehearsal:
        sys.exit(f"bench.py: {world} ranks but {ndev} visible device(s): a scaling number from ranks that share a GPU means nothing "
                 f"(--rehearsal runs it anyway and marks the line)")
    local = local % max(ndev, 1)  # --rehearsal on a smaller box: ranks share devices, the line says so
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)

    dev = torch.device("cuda", local)
    s, ch = args.size, 4
    if c5:
        # pixels + worst-case encode slots (channels + 1 bytes per pixel) + decoded pixels: the whole suite is 345 GB, i.e. it
        # needs two GPUs of 288 GB (BASELINE.json quotes it on eight)
        need = sum(im.pixels * (3 * im.channels + 1) + 1024 for im in mine)
        have = torch.cuda.mem_get_info(dev)[0]
        if need > 0.95 * have:
            raise SystemExit(f"bench.py --config c5: rank {rank} of {world} would need {need / 2**30:.0f} GiB of device memory for {len(mine)} images, "
                             f"{have / 2**30:.0f} GiB are free: use more GPUs (--gpus N) or fewer images (--images M)")
        wl = Suite(sq, torch, dev, mine)
    else:
        wl = Uniform(sq, torch, dev, len(mine), s, s, ch, "mixed", mine[0] if len(mine) else 0)

    work_stream = torch.cuda.Stream(device=dev)  # both legs in order on one stream
    stream = work_stream.cuda_stream
    run = lambda: wl.step(stream)  # noqa: E731
    graph = None
    if args.graph:  # plan launches allocate nothing and never touch the host: the step can be captured once and replayed
        try:
            run()
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=work_stream):
                wl.step(stream)
            run = graph.replay
        except Exception as e:  # noqa: BLE001 -- the direct path is always available
            if rank == 0:
                print(f"[bench] HIP graph capture unavailable ({e}); launching directly", file=sys.stderr)
            graph = None

    # The device has been idle while the CPU baseline and the setup ran, and the first ~50 ms of work after an idle period run at
    # lower clocks (tools/time_warm.py): untimed steps for at least 1 s, then the W warm-up steps the caller asked for.
    # A cold box can take longer (profiles/r03_d_bench.json: 28 spin-up steps at 5.4 ms, then a timed region at 5.27 ms while the
    # kernels measured right after it ran at their usual 1.54 + 3.46 ms): go on in groups of 8 until a group is no faster than the
    # one before it (within 1.5 %), for at least 1 s (profiles/r03_f_bench.json: the same again after 0.17 s of spin-up -- whatever
    # it is that a cold box ramps, it ramps slowly) and 3 s at most.
    spin_up, t_spin, prev_group, spin_ms = 0, time.perf_counter(), None, []
    while True:
        t_g = time.perf_counter()
        for _ in range(8):
            run()
        torch.cuda.synchronize()
        spin_up += 8
        group = time.perf_counter() - t_g
        spin_ms.append(round(group / 8 * 1e3, 4))
        gone = time.perf_counter() - t_spin
        if gone > 3.0 or (gone >= 1.0 and prev_group is not None and group > 0.985 * prev_group):
            break
        prev_group = group
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
    # board power / shader clock of this rank's card, read while a few more (untimed) steps run: outside the timed region
    for _ in range(max(4, min(args.steps, 16))):
        run()
    try:
        pr = torch.cuda.get_device_properties(dev)
        pci = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
    except Exception:  # noqa: BLE001 -- the property names vary with the torch build
        pci = None
    power = board_power(pci)
    torch.cuda.synchronize()
    totals = [float(wl.pixels), float(wl.enc_bytes), float(wl.read_bytes)]
    if dist is not None:
        on = dev if args.backend == "nccl" else "cpu"
        t = torch.tensor([dt], device=on, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tt = torch.tensor(totals, device=on, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        totals = tt.tolist()
    total_px, total_enc, total_read = totals

    # per-image stream lengths in global order: the only cross-rank step, a few bytes per image, outside the timed region
    n_global = n_total if (c5 or scaling == "strong") else n_total * world
    all_lens = sh.gather_lengths(wl.idx if c5 else list(mine), wl.lens.tolist(), n_global)

    # per-kernel durations with HIP events on the stream the kernels run on (the plan's own)
    k = max(args.steps, 5)
    enc_step_ms, dec_step_ms = in_step_times(torch, wl, work_stream, k)
    enc_ms, dec_ms = kernel_times(wl, k)

    if c5:
        workload = (f"C5: {n_total} qoi-suite-shaped images (seqoia_amd/suite.py: 64x64 .. 3840x2160, 3 and 4 channels), LPT-sharded on w*h over "
                    f"{world} GPU(s): sqoa_encode qoi_compat=0 then sqoa_decode, device-resident")
    else:
        workload = (f"{n_total} x {s}x{s} RGBA (class mixed, seed 1) {'per GPU' if scaling == 'weak' else 'in total, contiguous blocks per rank'}: "
                    f"sqoa_encode qoi_compat=0 then sqoa_decode to 4 channels, device-resident")
    out = {
        "metric": METRIC,
        "value": round(total_px * args.steps / dt / 1e6, 1),
        "unit": "Mpixels/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {
            "workload": workload,
            "images_rank0": wl.n,
            "images_total": n_global,
            "devices_used": min(world, ndev),
            "pixels_total": int(total_px),
            "encoded_bytes_total": int(total_enc),
            "encoded_bytes_check": int(sum(all_lens)),  # from the gathered per-image lengths: equals encoded_bytes_total
            "rate": round(total_enc / total_read, 4),
            "launch": "hip graph replay" if graph is not None else "direct",
            "spin_up_ms_per_step": [spin_ms[0], min(spin_ms), spin_ms[-1]],  # first group of 8, fastest, last
            "spin_up_steps": spin_up,  # untimed, in front of the W warm-up steps: the clocks ramp up after the idle setup phase
            # board power / its cap / shader clock from sysfs hwmon, read while untimed steps ran behind the timed region: of this rank's own card
            # ("own": matched by PCI address) or of the busiest card the host shows; None where hwmon is not readable
            "board_power_w": power[0] if power else None, "board_power_cap_w": power[1] if power else None, "sclk_mhz_hwmon": power[2] if power else None,
            "board_power_card": power[3] if power else None,
            "encode_mpps_rank0": round(wl.pixels / enc_ms / 1e3, 1),
            "decode_mpps_rank0": round(wl.pixels / dec_ms / 1e3, 1),
            # which form of the QOI-compat kernels this device gets (1: the ones that rest on the LDS's lane order -- the device passed the library's
            # probe of it, sqoa_selftest.hip; 0: the order-free ones)
            "lds_lane_order_probe_passed": int(sq.load().sqoa_hip_lds_order(local)),
            "roofline_frac_of_n_gpus": round(total_px * args.steps / dt / (world * HBM_PEAK_GBS * 1e9 / 4), 4),
        },
    }
    # Both kernels are HBM-bound by their algorithmic bytes; the decoder is the dominant one by time, the
    # encoder is the one the north-star's target (>= 50 % of the HBM-read roofline) is stated on.
    enc_pmc = {} if c5 else counters_from_profile("encode", wl.n, s)
    dec_pmc = {} if c5 else counters_from_profile("decode", wl.n, s)
    src = enc_pmc.get("source") or dec_pmc.get("source")
    dominant = "decode" if dec_step_ms >= enc_step_ms else "encode"

    def pmc_fields(row, pmc):
        # from the committed PMC summary of this very source (None / absent otherwise): HBM bytes per launch, instructions retired
        # per 64 pixels by kind, LDS conflict ratio, the clock the chip held under this kernel, the kernel ALONE as rocprofv3 timed it
        row["traffic"] = pmc.get("traffic")
        for f in ("insts_per_64px", "lds_conflict_cycles_per_active_cycle", "clock_ghz_pmc_pass"):
            if f in pmc:
                row[f] = pmc[f]
        # the kernel as rocprofv3 timed it in the committed profile round, by phase: inside the timed steps / in a loop of its own (median, p10:
        # steady state, the spin-up left out) -- `frac` recomputed from `alone.median` must reproduce the line's within the box-to-box spread
        if "rocprof_in_step_ms" in pmc:
            row["kernel_ms_rocprof"] = {"in_step": pmc["rocprof_in_step_ms"], "alone": pmc["rocprof_alone_ms"]}
        elif "rocprof_avg_ms" in pmc:
            row["kernel_ms_rocprof"] = {"avg": pmc["rocprof_avg_ms"], "min": pmc["rocprof_min_ms"]}
        return row

    # `plan_launch_ms`: HIP events around whole plan launches (the kernel named here plus the plan's stand-by launches behind it:
    # two for a decode plan, ~10 us in all), k launches back to back on the plan's stream, in a loop of its own behind the timed region
    # `frac` / `achieved`: from the launch INSIDE the encode -> decode step (`in_step_ms`: what `value` is made of -- a kernel behind the other one
    # shares the board's power budget with it and is not the kernel in a loop of its own); `frac_alone` / `achieved_alone`: from `plan_launch_ms`.
    # `cycles_per_64px_per_cu`: in-step time x shader clock (hwmon, read under the step) x compute units / (pixels / 64) -- the figure that
    # survives a change of box: the boxes of the pool differ by 5 % in the clock they hold under the 1 400 W cap.
    try:
        n_cus = int(torch.cuda.get_device_properties(dev).multi_processor_count)
    except Exception:  # noqa: BLE001
        n_cus = 256
    sclk = power[2] if power else None

    def cyc(ms):
        return round(ms * 1e-3 * sclk * 1e6 * n_cus / (wl.pixels / 64.0), 1) if sclk else None

    roof_enc = pmc_fields({
        "kernel": enc_pmc.get("kernel", "sqoa::encode_kernel<8, false>"), "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
        "achieved": round(wl.read_bytes / enc_step_ms / 1e6, 1),  # algorithmic READ bytes (channels B/px) / the launch's time in the step
        "frac": round(wl.read_bytes / enc_step_ms / 1e6 / HBM_PEAK_GBS, 4),
        "achieved_alone": round(wl.read_bytes / enc_ms / 1e6, 1),
        "frac_alone": round(wl.read_bytes / enc_ms / 1e6 / HBM_PEAK_GBS, 4),
        "cycles_per_64px_per_cu": cyc(enc_step_ms), "compute_units": n_cus,
        "achieved_read_plus_write": round((wl.read_bytes + wl.enc_bytes) / enc_step_ms / 1e6, 1),
        # the encoder also writes enc_bytes / read_bytes bytes per byte read: at the measured copy rate of the chip (6.29 TB/s of read +
        # write traffic) the READ rate cannot exceed this fraction of the 8 TB/s peak on this content
        "practical_ceiling_frac": round(HBM_COPY_GBS / HBM_PEAK_GBS / (1.0 + wl.enc_bytes / wl.read_bytes), 4),
        "plan_launch_ms": round(enc_ms, 4),
        "in_step_ms": round(enc_step_ms, 4),   # the same launch inside the encode -> decode step (median): what `value` is made of
    }, enc_pmc)
    roof_dec = pmc_fields({
        "kernel": dec_pmc.get("kernel", "sqoa::bal::decode_bal_kernel<8, 16, false>"), "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
        "achieved": round((wl.enc_bytes + wl.write_bytes) / dec_step_ms / 1e6, 1),  # stream bytes read + pixel bytes written / the launch's time in the step
        "frac": round((wl.enc_bytes + wl.write_bytes) / dec_step_ms / 1e6 / HBM_PEAK_GBS, 4),
        "achieved_alone": round((wl.enc_bytes + wl.write_bytes) / dec_ms / 1e6, 1),
        "frac_alone": round((wl.enc_bytes + wl.write_bytes) / dec_ms / 1e6 / HBM_PEAK_GBS, 4),
        "cycles_per_64px_per_cu": cyc(dec_step_ms), "compute_units": n_cus,
        "practical_ceiling_frac": round(HBM_COPY_GBS / HBM_PEAK_GBS, 4),   # all of its bytes are streaming traffic
        "plan_launch_ms": round(dec_ms, 4),
        "in_step_ms": round(dec_step_ms, 4),
    }, dec_pmc)
    out["roofline"] = dict(roof_dec if dominant == "decode" else roof_enc)
    out["roofline"]["dominant_by_time"] = dominant
    out["roofline"]["share_of_step"] = round(max(enc_step_ms, dec_step_ms) / (enc_step_ms + dec_step_ms), 3)
    if src:
        out["roofline"]["traffic_source"] = src
    out["roofline"]["encode"] = roof_enc
    out["roofline"]["decode"] = roof_dec
    if world > ndev:
        out["rehearsal"] = f"{world} ranks on {ndev} device(s): not a scaling measurement"
    if rank == 0 and world == 1 and not c5 and not args.no_secondary:
        out["config"]["secondary"] = secondary(sq, torch, dev, wl)
    if cpu is not None:
        out["cpu_baseline"] = cpu
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.cpu_worker:
        cpu_worker(args.cpu_worker)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus, args, sys.argv[1:])
        return
    run_rank(args)


if __name__ == "__main__":
    main()
