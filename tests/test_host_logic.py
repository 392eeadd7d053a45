"""Host-side logic of the drop-in surface (CPU only): sizing, validation messages, the hash plugin
helpers, key packing.  Expected values come from the real reference (tests/golden/golden.json) and
from the reference's own tests (cited)."""

import numpy as np
import pytest

import pyprobables_amd as pa
from pyprobables_amd import _native as N
from pyprobables_amd import hashes as H
from pyprobables_amd.keys import pack_hashes, pack_keys

from _util import as_key


def test_sizing_matches_reference(golden):
    for s in golden["sizing"]:
        if "error" in s:
            with pytest.raises(pa.InitializationError) as ei:
                pa.BloomFilter._get_optimized_params(s["n"], s["p"])
            assert str(ei.value) == s["error"] == ei.value.message
        elif "raises" in s:  # (10, 0.0): math.log(0) -> ValueError in the reference too
            with pytest.raises(ValueError):
                pa.BloomFilter._get_optimized_params(s["n"], s["p"])
        else:
            assert pa.BloomFilter._get_optimized_params(s["n"], s["p"]) == (s["fpr"], s["k"], s["m"]), s
    # BASELINE parameter sets: exact powers of two
    assert pa.BloomFilter._get_optimized_params(28005615, 0.01)[1:] == (7, 2**28)
    assert pa.BloomFilter._get_optimized_params(224044920, 0.01)[1:] == (7, 2**31)


def test_constructor_validation_messages():
    # reference tests/bloom_test.py:395-473, countminsketch_test.py init tests
    for cls, msg in ((pa.BloomFilter, "Insufecient parameters to set up the Bloom Filter"),
                     (pa.CountingBloomFilter, "Insufecient parameters to set up the Counting Bloom Filter")):
        with pytest.raises(pa.InitializationError) as ei:
            cls()
        assert str(ei.value) == msg
        with pytest.raises(pa.InitializationError):
            cls(est_elements=10)
        with pytest.raises(pa.InitializationError) as ei:
            cls(est_elements=0, false_positive_rate=0.1)
        assert str(ei.value) == "Bloom: estimated elements must be greater than 0"
        with pytest.raises(pa.InitializationError) as ei:
            cls(est_elements=10, false_positive_rate=1.5)
        assert str(ei.value) == "Bloom: false positive rate must be between 0.0 and 1.0"
        with pytest.raises(pa.InitializationError) as ei:
            cls(est_elements=50, false_positive_rate=0.99)
        assert str(ei.value) == "Bloom: Number hashes is zero; unusable parameters provided"
        with pytest.raises(pa.InitializationError):  # invalid hex -> falls through to params (bloom_test.py:318-321)
            cls(hex_string="85f240623b6d9459000000000000000a000000000000000a3d4ccccQ")
    with pytest.raises(pa.InitializationError) as ei:
        pa.CountMinSketch(width=0, depth=5)
    assert str(ei.value) == "CountMinSketch: width and depth must be greater than 0"
    with pytest.raises(pa.InitializationError):
        pa.CountMinSketch(confidence=-1, error_rate=0.1)
    with pytest.raises(pa.InitializationError) as ei:
        pa.CountMinSketch()
    assert str(ei.value).startswith("Must provide one of the following to initialize the Count-Min Sketch:")
    assert issubclass(pa.InitializationError, pa.ProbablesBaseException)


def test_host_hash_helpers_match_reference(golden):
    for case in golden["hashes"]:
        assert H.default_fnv_1a(as_key(case), case["depth"]) == case["hashes"]
    for case in golden["fnv_1a_seeded"]:
        assert H.fnv_1a(case["key"], case["seed"]) == case["hash"]
    # reference tests/hashes_test.py:57-62, :64-146
    assert H.fnv_1a_32("this is a test", 0) == 2139996864
    assert H.fnv_1a_32("this is also a test", 0) == 1462718619
    assert H.default_md5("this is a test", 5) == [
        12174049463882854484, 10455450501617390806, 3838261292881602234, 12102952520950148619, 12126605867972429202]
    assert H.default_sha256("this is a test", 5)[0] == 10244166640140130606
    assert H.default_md5(b"this is a test", 5) == H.default_md5("this is a test", 5)

    @H.hash_with_depth_int
    def my_hash(key, depth=1, encoding="utf-8"):
        import hashlib
        return int(hashlib.sha512(key.encode(encoding)).hexdigest(), 16)

    assert len(my_hash("this is a test", 5)) == 5
    assert my_hash("this is a test", 3) == my_hash("this is a test", 5)[:3]


def test_pack_keys_layouts():
    b = pack_keys([b"abcd", b"efgh"])
    assert (b.layout, b.n, b.key_len, b.where) == (N.KEYS_FIXED, 2, 4, N.HOST)
    b = pack_keys("a single key")
    assert (b.layout, b.n, b.key_len) == (N.KEYS_FIXED, 1, 12)
    b = pack_keys([b"a", b"bcd", b""])
    assert (b.layout, b.n) == (N.KEYS_VARLEN8, 3)
    offs = b.keep[1]
    assert offs.tolist() == [0, 1, 4, 4] and bytes(b.keep[0]) == b"abcd"
    b = pack_keys(["caf\xe9", "na\xefve"])  # latin-1 range: one byte per code point (NOT utf-8)
    assert b.layout == N.KEYS_VARLEN8 and bytes(b.keep[0]) == "caf\xe9na\xefve".encode("latin-1")
    b = pack_keys(["€1", b"ab", "x"])  # a code point > 255 widens the whole batch
    assert b.layout == N.KEYS_VARLEN32
    assert b.keep[0].tolist() == [0x20AC, ord("1"), ord("a"), ord("b"), ord("x")] and b.keep[1].tolist() == [0, 2, 4, 5]
    a = np.arange(48, dtype=np.uint8).reshape(3, 16)
    b = pack_keys(a)
    assert (b.layout, b.n, b.key_len) == (N.KEYS_FIXED, 3, 16)
    b = pack_keys(np.array([b"abc", b"de"], dtype="S3"))
    assert (b.n, b.key_len) == (2, 3) and bytes(b.keep[0]) == b"abcde\x00"
    b = pack_keys([])
    assert b.n == 0
    with pytest.raises(TypeError):
        pack_keys([1, 2, 3])
    with pytest.raises(TypeError):
        pack_keys(np.zeros((3, 4), dtype=np.int32))


def test_pack_hashes():
    b = pack_hashes([1, 2, 3, 2**64 - 1], 4)
    assert (b.layout, b.n, b.key_len) == (N.KEYS_HASHES, 1, 4) and b.keep[0].tolist() == [[1, 2, 3, 2**64 - 1]]
    b = pack_hashes([[1, 2], [3, 4], [5, 6]], 2)
    assert (b.n, b.key_len) == (3, 2)
    b = pack_hashes(np.arange(12, dtype=np.uint64).reshape(4, 3), 3)
    assert (b.n, b.key_len) == (4, 3)
    with pytest.raises(ValueError):
        pack_hashes([1, 2], 4)
    with pytest.raises(ValueError):
        pack_hashes([[1, 2], [3]], 1)


def test_public_surface_names():
    for name in ("BloomFilter", "CountingBloomFilter", "CountMinSketch", "CountMeanSketch", "CountMeanMinSketch",
                 "InitializationError", "NotSupportedError", "ProbablesBaseException", "SimilarityError"):
        assert hasattr(pa, name)
    for meth in ("add", "check", "add_alt", "check_alt", "hashes", "export", "export_hex", "frombytes", "union",
                 "intersection", "jaccard_index", "estimate_elements", "clear", "add_many", "check_many"):
        assert hasattr(pa.BloomFilter, meth)
    for meth in ("add", "remove", "check", "add_alt", "remove_alt", "check_alt", "join", "export", "frombytes",
                 "add_many", "remove_many", "check_many"):
        assert hasattr(pa.CountMinSketch, meth)
    for meth in ("remove", "remove_alt", "remove_many"):
        assert hasattr(pa.CountingBloomFilter, meth)


def test_vectorised_key_packing_equals_the_per_key_path(monkeypatch):
    """homogeneous lists are packed with one join / one encode; the result must be the buffer the careful per-key loop
    builds (layout, bytes, offsets), for latin-1 and wide strings, ragged / fixed / empty byte keys"""
    import numpy as np

    from pyprobables_amd import _native as N
    from pyprobables_amd import keys as K

    def image(b):
        import ctypes as C

        if b.layout == N.KEYS_FIXED:
            size = b.n * b.key_len
            data = bytes((C.c_uint8 * size).from_address(b.data)) if size else b""
            return (b.layout, b.n, b.key_len, data, None)
        offs = np.ctypeslib.as_array((C.c_uint64 * (b.n + 1)).from_address(b.offsets)).copy()
        width = 4 if b.layout == N.KEYS_VARLEN32 else 1
        size = int(offs[-1]) * width
        data = bytes((C.c_uint8 * size).from_address(b.data)) if size else b""
        return (b.layout, b.n, 0, data, offs.tolist())

    rng = np.random.default_rng(3)
    cases = [
        [f"key-{i}" for i in range(300)],                      # ragged latin-1 str
        [f"{i:08d}" for i in range(300)],                      # fixed-length str
        ["é", "", "abc", "\xff\x00"],                          # latin-1 edge values, empties
        ["日本語", "a", "", "😀 emoji"],                          # wide: code points, never UTF-8
        [bytes(rng.integers(0, 256, size=int(n), dtype=np.uint8)) for n in rng.integers(0, 20, size=200)],
        [bytes(rng.integers(0, 256, size=16, dtype=np.uint8)) for _ in range(100)],
        [b"", b""],
        [bytearray(b"ab"), b"cd", memoryview(b"efg")],
    ]
    cases += [
        ["ab", b"cd", bytearray(b"ef")],                          # mixed str / bytes of one length
        ["a", b"bcd", "\xe9\xff", b""],                           # mixed, ragged
        ["x", b"\xfe\xff", "\u0100", "\U0001F600\ud800"],        # mixed with wide strings and a lone surrogate: everything widens
        [("k%d" % i) * (i % 5) for i in range(500)] + ["\u20ac"],  # one wide key at the end widens 500 narrow ones
    ]
    if K._pylist is None:  # a tree whose HIP engine was built elsewhere: the packer is one gcc call (build.py does it with the engine)
        import importlib

        from pyprobables_amd import build as B

        B.build_pylist(verbose=False)
        monkeypatch.setattr(K, "_pylist", importlib.import_module("pyprobables_amd._pylist"))
    for keys in cases:
        c_path = image(K.pack_keys(keys))                         # csrc/psk_pylist.c
        with monkeypatch.context() as m:
            m.setattr(K, "_pylist", None)
            fast = image(K.pack_keys(keys))                       # one join / one encode
            m.setattr(K, "_pack_homogeneous", lambda keys, n: None)
            slow = image(K.pack_keys(keys))                       # the per-key loop
        assert fast == slow, keys[:3]
        assert c_path == slow, keys[:3]
    with pytest.raises(TypeError):
        K.pack_keys(["a", 3])
    mixed = K.pack_keys(["ab", b"cd"])   # mixed str / bytes still works (careful path)
    assert mixed.n == 2


def _foreign(name, impl):
    """a function that looks like the reference's own `probables.hashes.<name>` (module + qualified name)"""
    def fn(key, depth=1):
        return impl(key, depth)
    fn.__module__, fn.__name__, fn.__qualname__ = "probables.hashes", name, name
    return fn


def test_reference_hash_functions_are_recognised_as_the_fused_families():
    """the literal drop-in scenario: a user passes the REFERENCE's default_fnv_1a (bloom.py:496-499 installs that object
    by default); it must take the fused kernels, not the per-key host route.  Accepted by name, then probed."""
    assert H.is_fused_fnv(None) and H.is_fused_fnv(H.default_fnv_1a)
    assert H.is_fused_fnv(_foreign("default_fnv_1a", H.default_fnv_1a))
    assert H.device_digest(_foreign("default_md5", H.default_md5)) == 0
    assert H.device_digest(_foreign("default_sha256", H.default_sha256)) == 1
    # a look-alike that hashes differently stays a plugin (host route), whatever it is called
    assert not H.is_fused_fnv(_foreign("default_fnv_1a", lambda k, d: [1] * d))
    assert not H.is_fused_fnv(_foreign("default_fnv_1a", H.default_md5))
    # same results but an unrelated function object (a user wrapper) is NOT taken over: only identity or the reference's name
    assert not H.is_fused_fnv(lambda k, d: H.default_fnv_1a(k, d))
    assert H.device_digest(H.default_fnv_1a) is None and H.device_digest(_foreign("default_fnv_1a", H.default_fnv_1a)) is None
    # a function that raises on the probe is simply a plugin
    def boom(key, depth=1):
        raise RuntimeError("no")
    boom.__module__, boom.__qualname__, boom.__name__ = "probables.hashes", "default_fnv_1a", "default_fnv_1a"
    assert not H.is_fused_fnv(boom)


def test_homogeneous_packing_rejects_non_key_elements():
    """every element of a bytes list is type-checked (b''.join accepts any buffer: a numpy array in the middle used to be
    hashed as its raw bytes, while the per-key path and add() raise TypeError for it)"""
    good = [b"abc", b"defg", b"hi"]
    assert pack_keys(good).n == 3
    bad = [b"abc", np.arange(4, dtype=np.uint8), b"hi"]
    with pytest.raises(TypeError):
        pack_keys(bad)
    import array as _array
    with pytest.raises(TypeError):
        pack_keys([b"abc", _array.array("B", [1, 2, 3]), b"hi"])
    assert pack_keys([b"abc", bytearray(b"xy"), memoryview(b"z")]).n == 3


def test_pack_keys_ragged_pairs_on_the_host():
    """(blob, offsets) pairs: layout by element width, offsets rebased to the first key, validation of shapes / dtypes / order"""
    blob = np.arange(20, dtype=np.uint8)
    b = pack_keys((blob, np.array([3, 5, 5, 12], dtype=np.int64)))
    assert (b.layout, b.n, b.where) == (N.KEYS_VARLEN8, 3, N.HOST)
    assert b.keep[1].tolist() == [0, 2, 2, 9] and bytes(b.keep[0][:9]) == bytes(range(3, 12))
    cps = np.array([0x20AC, 65, 66, 0x1F600], dtype=np.uint32)
    b = pack_keys((cps, np.array([0, 1, 4], dtype=np.uint64)))
    assert (b.layout, b.n) == (N.KEYS_VARLEN32, 2)
    b = pack_keys((np.zeros(0, dtype=np.uint8), np.zeros(3, dtype=np.int64)))  # all keys empty
    assert (b.layout, b.n) == (N.KEYS_VARLEN8, 2) and b.data != 0
    with pytest.raises(TypeError):
        pack_keys((blob, np.array([0, 3], dtype=np.int32)))
    with pytest.raises(TypeError):
        pack_keys((blob.astype(np.float32), np.array([0, 3], dtype=np.int64)))
    with pytest.raises(ValueError):
        pack_keys((blob, np.array([0, 30], dtype=np.int64)))
    with pytest.raises(ValueError):
        pack_keys((blob, np.array([0, 5, 4], dtype=np.int64)))


def test_single_key_bytes_are_the_packers_bytes():
    """`one_key_bytes` (the value-returning per-key calls) hands the engine the same bytes `pack_keys` would, or defers to it: a str by code
    point (hashes.py:98) -- so only code points <= 255 fit one byte each -- bytes as they are, everything else through the general packer"""
    from pyprobables_amd._base import OneKey
    from pyprobables_amd.keys import one_key_bytes

    for key in ("plain ascii", "café ÿ", "", b"", b"raw \x00 bytes \xff", "x" * 300):
        raw = one_key_bytes(key)
        b = pack_keys([key])
        assert b.layout == N.KEYS_FIXED and b.n == 1 and b.key_len == len(raw)
        assert raw == (b.keep[0].tobytes() if b.key_len else b"")
    for key in ("wide € \U0001f600", bytearray(b"ba"), memoryview(b"mv"), 17, None, ["k"]):
        assert one_key_bytes(key) is None      # (pack_keys packs the first three and raises TypeError for the rest)
    with pytest.raises(TypeError):
        pack_keys([17])
    one = OneKey()                             # preallocated words: the views alias ONE buffer, the addresses are its
    one.o[0] = -1
    assert one.o_addr == one.o.ctypes.data and one.w_addr == one.w.ctypes.data
    assert int(one.o_u8[0]) == 255 and int(one.o_u32[1]) == 0xFFFFFFFF and int(one.o_i32[0]) == -1
