#!/usr/bin/env python3
"""Board power and shader clock (sysfs hwmon / rocm-smi) while a loop of the headline kernels runs: encode only, decode only, the
encode -> decode step.  Is the step at the board's power cap?"""
import glob, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, seqoia_amd as sq
n = 1024
px = torch.empty((n, 1024, 1024, 4), dtype=torch.uint8, device="cuda"); sq.synth(px, "mixed", 1, 0)
enc = sq.EncodePlan(px); enc.launch(); res = enc.results()
dec = sq.DecodePlan(enc.out, res["len"], 1024, 1024, 4); dec.launch(); dec.results()
st = torch.cuda.Stream(); s = st.cuda_stream
hw = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")
def rd(p):
    try: return open(p).read().strip()
    except OSError: return None
def sample():
    """The busiest card of the host's eight (the one this process runs on, unless a neighbour is busier): max over hwmon dirs."""
    best = {}
    for h in hw:
        out = {}
        for f in ("power1_average", "power1_input", "power1_cap", "freq1_input", "temp1_input"):
            v = rd(os.path.join(h, f))
            if v is not None: out[f] = v
        v = rd(os.path.join(os.path.dirname(os.path.dirname(h)), "pp_dpm_sclk"))
        if v: out["sclk"] = [l for l in v.splitlines() if "*" in l]
        pw = int(out.get("power1_average") or out.get("power1_input") or 0)
        if pw >= int(best.get("power1_average") or best.get("power1_input") or -1): best = out
    try:
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
        best["smi"] = [l.split(":", 1)[1].strip() for l in r.splitlines() if "Power (W)" in l or "sclk clock" in l]
    except Exception as e: best["smi"] = str(e)
    return best
print("hwmon dirs:", hw, "idle:", sample(), flush=True)
def loop(fn, secs):
    stop = [False]; cnt = [0]
    def run():
        while not stop[0]:
            for _ in range(8): fn()
            torch.cuda.synchronize(); cnt[0] += 8
    t = threading.Thread(target=run); t0 = time.perf_counter(); t.start()
    samples = []
    while time.perf_counter() - t0 < secs:
        time.sleep(0.1); samples.append(sample())
    stop[0] = True; t.join(); dt = time.perf_counter() - t0
    return dt / cnt[0] * 1e3, samples
for name, fn in (("encode", lambda: enc.launch(s)), ("decode", lambda: dec.launch(s)), ("step", lambda: (enc.launch(s), dec.launch(s))), ("decode", lambda: dec.launch(s)), ("step", lambda: (enc.launch(s), dec.launch(s)))):
    ms, sm = loop(fn, 4.0)
    print(f"{name}: {ms:.3f} ms per iteration;", "hwmon power (W):", [int(x.get("power1_average") or x.get("power1_input") or 0) // 1000000 for x in sm][-5:], "sclk:", sm[-1].get("sclk"), "rocm-smi:", [x.get("smi") for x in sm][-3:], flush=True)
try:
    print(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True, timeout=20).stdout[-1500:])
except Exception as e: print("rocm-smi:", e)
