#!/usr/bin/env python3
"""C5 (suite-shaped mixed sizes): rank 0-of-8's shard on one GPU, encode / decode kernel times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, seqoia_amd as sq, bench
from seqoia_amd import suite
shard = suite.shard(suite.C5_IMAGES, 8, 0)
wl = bench.Suite(sq, torch, torch.device("cuda", 0), shard)
wl.enc.time(2); wl.dec.time(2)
e, d = bench.kernel_times(wl, 5)
print(f"{os.environ.get('SQOA_LIB','default')} C5 rank 0/8: {len(shard)} images, {wl.pixels/1e9:.2f} Gpx: encode {e:.2f} ms ({wl.pixels/e/1e6:.0f} Gpx/s), decode {d:.2f} ms ({wl.pixels/d/1e6:.0f} Gpx/s)")
