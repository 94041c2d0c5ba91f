"""The qoi-benchmark-suite-shaped workload of BASELINE.json configs[4] ("C5": 65 536 mixed-size images sharded over
8 MI355X), as a size table.

The reference's recorded sqoabench runs over the (external, un-vendored) qoi image suite give, per directory, the image
count (total size / average size) and the average pixel count (mpps x ms): /root/reference/bench1-noaverage.txt:118-123,
/root/reference/bench10.txt:44-123; SURVEY.md Appendix C lists the derived table.  One replica = 2 871 images, 1.32 Gpx
(the reference's suite: about 2 850 images, 1.32 Gpx).  C5 walks the replica cyclically until 65 536 images (22.8
replicas); image k has content class by directory kind and generator index k (SURVEY.md Appendix C generator, seed 1).

The per-image loop this replaces is sqoabench.c:575-628 (one result per image, totals per directory).
"""
from __future__ import annotations

from typing import List, NamedTuple

# directory, images, width, height, channels, content class
SUITE = [
    ("icon_64", 226, 64, 64, 4, "icon"),
    ("icon_512", 214, 512, 512, 4, "icon"),
    ("textures_pk", 1010, 192, 192, 4, "mixed"),
    ("textures_pk01", 113, 368, 368, 4, "mixed"),
    ("textures_pk02", 235, 640, 480, 4, "mixed"),
    ("screenshot_game", 619, 1024, 625, 4, "mixed"),
    ("pngimg", 187, 1600, 1125, 4, "mixed"),
    ("textures_plants", 60, 1024, 1024, 4, "mixed"),
    ("screenshot_web", 14, 3840, 2160, 4, "mixed"),
    ("photo_kodak", 24, 768, 512, 3, "photo"),
    ("photo_tecnick", 100, 1200, 1200, 3, "photo"),
    ("photo_wikipedia", 49, 1200, 900, 3, "photo"),
    ("textures_photo", 20, 1024, 1024, 3, "photo"),
]

C5_IMAGES = 65536


class SuiteImage(NamedTuple):
    index: int      # global image number = generator index
    directory: str
    width: int
    height: int
    channels: int
    cls: str

    @property
    def pixels(self) -> int:
        return self.width * self.height


def replica() -> List[tuple]:
    out = []
    for name, count, w, h, ch, cls in SUITE:
        out += [(name, w, h, ch, cls)] * count
    return out


def images(n: int = C5_IMAGES) -> List[SuiteImage]:
    """The first n images of the cyclically repeated suite replica."""
    rep = replica()
    return [SuiteImage(k, *rep[k % len(rep)]) for k in range(n)]


def shard(n: int, world: int, rank: int) -> List[SuiteImage]:
    """Rank `rank`'s images: longest-processing-time-first on the pixel count."""
    from .shard import lpt

    imgs = images(n)
    return [imgs[i] for i in lpt([im.pixels for im in imgs], world)[rank]]
