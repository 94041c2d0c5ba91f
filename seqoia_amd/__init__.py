"""seqoia_amd -- MI355X-native batch SQOA/QOI codec behind jido/seqoia's C API.

The product is ``seqoia_amd/lib/libsqoa_mi355x.so`` (sources in ``seqoia_amd/csrc``,
C headers in ``include/``).  This package is the thin host-side mirror of the
reference interface over that C-ABI; it contains no codec logic of its own.
"""
from ._lib import Desc, Job, Result, SqoaError, LIB_PATH, load  # noqa: F401
from .api import (  # noqa: F401
    SQOA_LINEAR,
    SQOA_SRGB,
    DecodeList,
    DecodePlan,
    EncodeList,
    EncodePlan,
    Plan,
    decode,
    encode,
    encode_bound,
    fnv1a64,
    read,
    stride_channels,
    synth,
    write,
)
