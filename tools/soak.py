#!/usr/bin/env python3
"""Differential soak on the GPU box: plans of random image lists (sizes 1x1 .. 900x700, 3 / 4 channels, the three synthetic
classes plus low-entropy noise with long runs, both wire formats) through EncodeList / DecodeList under every forcing of the
routes (SQOA_HIP_STRIPS, SQOA_HIP_MULTIPASS, and the wave decoder's shapes: SQOA_HIP_DEC_WIDE, SQOA_HIP_DEC_SWEEP, SQOA_HIP_DEC_BAL), every
stream and every decoded image against the reference (oracle/_ref).
usage: soak.py [rounds] [seed] [seconds]      (rounds or seconds, whichever ends first; tests/test_gpu_soak.py runs 40 rounds of it in the -m gpu suite)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, seqoia_amd as sq
from oracle import pyoracle as po

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
t_start = time.time()
o = po.best()
assert o.kind == "reference", "the soak compares with the compiled reference (oracle/_ref)"
checked = rounds_done = 0
shapes = {"balanced": 0, "changed kernel": 0, "split": 0, "lazy": 0}   # what the default-knob decode plans took (sqoa_hip_plan_shape: bit 3 when filled -- and gone behind results() --, bits 9 / 8 behind results())
for r in range(rounds):
    if time.time() - t_start > budget:
        break
    n = int(rng.choice([1, 2, 5, 17, 40, 700]))   # (700: enough small images for the one-wave encode shape)
    # every eighth round: a list the plan's OWN choice sends to the balanced decoder -- 300 four-channel images of 30 KB or so (more streams than compute units:
    # one workgroup each, not the chip-wide passes), photographs among screenshots --, so that
    # the content feedback of sqoa_hip_plan_results (the whole list changes kernel, or the photographs are split off into a launch of their own) and the
    # relaunch behind it are part of the soak
    blend = r % 8 == 7
    if blend:
        n = 300
    imgs = []
    for k in range(n):
        big = not blend and rng.random() < (0.01 if n > 100 else 0.25)
        w = int(rng.integers(200, 900)) if big else int(rng.integers(90, 160)) if blend else int(rng.integers(1, 120))
        h = int(rng.integers(150, 700)) if big else int(rng.integers(90, 140)) if blend else int(rng.integers(1, 90))
        ch = 4 if blend else int(rng.choice([3, 4]))
        kind = (0 if rng.random() < 0.6 else 2) if blend else int(rng.integers(0, 5))
        if kind < 3:
            raw = po.synth(w, h, ch, ["mixed", "icon", "photo"][kind], seed=100 + r, img=k)
        elif kind == 3:   # few colours, long runs, repeats: INDEX / RUN heavy
            pal = rng.integers(0, 256, size=(int(rng.integers(2, 9)), ch), dtype=np.uint8)
            idx = np.repeat(rng.integers(0, len(pal), size=w * h // 7 + 2), rng.integers(1, 40, size=w * h // 7 + 2))[: w * h]
            idx = np.concatenate([idx, np.zeros(w * h - len(idx), dtype=idx.dtype)])
            raw = pal[idx].reshape(-1)
        else:
            raw = rng.integers(0, 256, size=w * h * ch, dtype=np.uint8)
            a = int(rng.integers(0, max(1, w * h - 1)))
            raw.reshape(-1, ch)[a : a + int(rng.integers(1, 70000))] = raw.reshape(-1, ch)[a]
        imgs.append((w, h, ch, np.ascontiguousarray(raw)))
    for mode in (0, 1):
        wants = [o.encode(raw, w, h, ch, mode) for (w, h, ch, raw) in imgs]
        # (the plan's own choices; the chip-wide routes for everything; one workgroup per image with the wave decoder at 8 bytes per
        # lane, at 12, at 12 with the position-major store path, and the balanced kernel wherever a plan's jobs all go to four channels)
        # (round 6: one of the forcings runs on the order-free QOI-compat kernels, SQOA_HIP_LDS_ORDER=0 -- sqoa_selftest.hip; and every decode plan of up to
        # 40 images (and of the blend rounds) is launched a second time BEHIND results() -- the lazy skip of the stand-by kernels, the content-driven change of kernel -- with the
        # pixels looked at before results() is asked again)
        for strips, passes, wide, sweep, bal, lds in (("", "", "", "", "", ""), ("1", "1", "", "", "", ""), ("0", "0", "0", "0", "0", "0"), ("0", "0", "1", "0", "0", ""),
                                                      ("0", "0", "1", "1", "0", ""), ("0", "0", "1", "0", "1", ""), ("", "", "", "", "", "0")):
            if lds: os.environ["SQOA_HIP_LDS_ORDER"] = lds
            else: os.environ.pop("SQOA_HIP_LDS_ORDER", None)
            for knob, v in (("STRIPS", strips), ("MULTIPASS", passes), ("DEC_WIDE", wide), ("DEC_SWEEP", sweep), ("DEC_BAL", bal)):
                if v: os.environ["SQOA_HIP_" + knob] = v
                else: os.environ.pop("SQOA_HIP_" + knob, None)   # (unset, not empty: a SQOA_HIP_DEC_BAL that is set at all keeps the plan from changing kernel)
            images = [(torch.from_numpy(raw).cuda(), sq.Desc(w, h, ch, 0, mode)) for (w, h, ch, raw) in imgs]
            enc = sq.EncodeList(images)
            enc.launch()
            res = enc.results()
            got = enc.streams()
            for k, wnt in enumerate(wants):
                assert res["status"][k] == 0 and got[k] == wnt, ("encode", r, mode, strips, k, imgs[k][:3])
            base = enc.out.data_ptr()
            for oc in (0, 4, 3):
                items = [(base + enc.offsets[k], len(wants[k]), w, h, oc if oc else ch) for k, (w, h, ch, _) in enumerate(imgs)]
                dec = sq.DecodeList(items)
                sh0 = dec.shape()
                dec.launch()
                dres = dec.results()
                for k, (w, h, ch, raw) in enumerate(imgs):
                    occ = oc if oc else ch
                    want = o.decode(wants[k], occ)[0]
                    have = dec.out[dec.offsets[k] : dec.offsets[k] + w * h * occ].cpu().numpy()
                    assert dres["status"][k] == 0 and have.tobytes() == want.tobytes(), ("decode", r, mode, passes, wide, sweep, bal, oc, k, imgs[k][:3])
                    checked += 1
                if n <= 40 or blend:
                    first = dec.out.clone()
                    dec.out.zero_()
                    torch.cuda.synchronize()
                    dec.launch()
                    torch.cuda.synchronize()
                    for k, (w, h, ch, _) in enumerate(imgs):   # (image by image: the few bytes between two images are nobody's)
                        a, b = dec.offsets[k], dec.offsets[k] + w * h * (oc if oc else ch)
                        assert torch.equal(dec.out[a:b], first[a:b]), ("relaunch behind results()", r, mode, passes, wide, sweep, bal, oc, k)
                    assert (dec.results()["status"] == 0).all(), ("relaunch", r, mode, oc)
                    if not (strips or passes or wide or sweep or bal or lds):
                        sh = dec.shape()
                        shapes["balanced"] += sh0 >> 3 & 1
                        shapes["changed kernel"] += (sh0 >> 3 & 1) & ~(sh >> 3 & 1)
                        shapes["split"] += sh >> 9 & 1
                        shapes["lazy"] += sh >> 8 & 1
                dec.close()
            enc.close()
    print(f"round {r}: {n} images ok", flush=True)
    rounds_done = r + 1
print(f"default-knob decode plans that were launched again behind results(): {shapes}")
print(f"soak ok: {checked} decoded images and their streams identical to the reference ({rounds_done} rounds, {time.time() - t_start:.0f} s)")
