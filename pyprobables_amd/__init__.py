"""pyprobables_amd -- MI355X-native engine for the bulk insert / lookup path of pyprobables.

Drop-in names for the accelerated path (reference ``probables/__init__.py:3-53``):
``BloomFilter``, ``CountingBloomFilter``, ``CountMinSketch`` (+ ``CountMeanSketch`` /
``CountMeanMinSketch``), ``ExpandingBloomFilter`` / ``RotatingBloomFilter``, their exceptions and the ``hash_function`` helpers.  Tables live in GPU HBM,
the work is done by hand-written gfx950 HIP kernels behind the C ABI in ``include/psk.h``.
"""

from .bloom import BloomFilter
from .countingbloom import CountingBloomFilter
from .countminsketch import CountMeanMinSketch, CountMeanSketch, CountMinSketch
from .expandingbloom import ExpandingBloomFilter, RotatingBloomFilter
from .exceptions import (
    CountMinSketchError,
    InitializationError,
    NativeLibraryError,
    NotSupportedError,
    ProbablesBaseException,
    RotatingBloomFilterError,
    SimilarityError,
)
from .hashes import default_fnv_1a, default_md5, default_sha256, fnv_1a, hash_with_depth_bytes, hash_with_depth_int

__version__ = "0.1.0"

__all__ = [
    "BloomFilter",
    "CountingBloomFilter",
    "CountMinSketch",
    "CountMeanSketch",
    "CountMeanMinSketch",
    "ExpandingBloomFilter",
    "RotatingBloomFilter",
    "RotatingBloomFilterError",
    "InitializationError",
    "NotSupportedError",
    "ProbablesBaseException",
    "SimilarityError",
    "CountMinSketchError",
    "NativeLibraryError",
    "default_fnv_1a",
    "fnv_1a",
    "default_md5",
    "default_sha256",
    "hash_with_depth_bytes",
    "hash_with_depth_int",
]
