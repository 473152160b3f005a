"""The C-ABI shared library builds for gfx950, loads on a GPU-less host, and exports every symbol
that include/fact_hip.h declares (no compute calls here; those are the -m gpu tests)."""
import ctypes as C
import os
import re

import pytest

from mint_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "fact_hip.h")              # the drop-in boundary
DEBUG_HEADER = os.path.join(ROOT, "include", "fact_hip_debug.h")  # test / bench surface, not for production hosts


def _declared(path=HEADER):
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fact_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(L.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return L.lib()


def test_header_symbols_are_exported_and_bound(lib):
    public, debug = _declared(), _declared(DEBUG_HEADER)
    assert len(public) >= 20 and len(debug) >= 20 and not set(public) & set(debug)
    for n in public + debug:
        assert hasattr(lib, n), "libfact_hip_dbg.so does not export %s" % n
        assert n in L.SIGNATURES, "mint_amd/_lib.py has no ctypes signature for %s" % n
    for n in L.SIGNATURES:
        assert n in public or n in debug, "%s bound in _lib.py but declared in neither header" % n
    assert set(debug) == set(L.DEBUG_SYMBOLS)


def _exported(path):
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    # functions (T / W) and data objects; hipcc's per-translation-unit id objects (__hip_cuid_*) are not an interface
    return sorted(l.split()[-1] for l in out.splitlines() if len(l.split()) >= 3 and l.split()[-2] in ("T", "W", "D", "B")
                  and not l.split()[-1].startswith("__hip_cuid_"))


def test_production_library_exports_exactly_the_public_header(lib):
    """libfact_hip.so (what a host binds) is built with -fvisibility=hidden: its dynamic symbol table holds the entry points
    of include/fact_hip.h and NOTHING else - no fact_debug_* / fact_op_* / fact_probe_* / fact_kprof* (the timing-only
    `skip` mask is not one dlsym away), no internal C++ launcher.  libfact_hip_dbg.so (tests, bench.py) adds the debug
    header and still no internals."""
    public, debug = _declared(), _declared(DEBUG_HEADER)
    assert os.path.exists(L.PROD_LIB_PATH) and os.path.exists(L.DEBUG_LIB_PATH)
    assert _exported(L.PROD_LIB_PATH) == public
    assert _exported(L.DEBUG_LIB_PATH) == sorted(public + debug)
    prod = C.CDLL(L.PROD_LIB_PATH)
    assert prod.fact_abi_version() == 3
    for n in debug:
        assert not hasattr(prod, n), n
    assert L.DEBUG_ABI and L.LIB_PATH == L.DEBUG_LIB_PATH  # tests/conftest.py selected the test / bench build


def test_public_header_carries_no_lab_bench():
    """The header a maintainer binds is the SURVEY 8(b) surface only: no single-op / probe / recorder entry points, no
    fact_debug_* switch, and no option key that selects kernels or - like the timing-only ablation mask - changes results."""
    text = open(HEADER).read()
    for n in _declared():
        assert not n.startswith(("fact_debug", "fact_op_", "fact_probe", "fact_kprof")), n
    block = text[text.index("/* Engine options"):text.index("int fact_set_option")]
    keys = set(re.findall(r'^ \*   "([a-z0-9_]+)"', block, flags=re.M))
    assert keys == set(L.PUBLIC_OPTIONS), keys
    assert '"skip"' not in text and "attn_variant" not in text and "tn_loop" not in text


def test_abi_version_and_struct_layout(lib):
    assert lib.fact_abi_version() == 3
    assert C.sizeof(L.FactStackCfg) == 24 and C.sizeof(L.FactConfig) == 3 * 24 + 8
    assert C.sizeof(L.FactParamDesc) == 96 + 8 + 12 + 4  # name, offset, rows/cols/kind, padding
    assert C.sizeof(L.FactArenas) == 32


def test_config_validation_without_gpu(lib):
    def cfg(hm=800, ha=800, heads=10, ff=3072):
        st = lambda seq, feat, h, layers=2: L.FactStackCfg(seq, feat, h, layers, heads, ff)
        return L.FactConfig(st(120, 225, hm), st(240, 35, ha), st(0, 0, hm, 12), 225, 1e-5)
    n, t = C.c_size_t(0), C.c_int(0)
    assert lib.fact_arena_size(C.byref(cfg()), C.byref(n), C.byref(t)) == 0
    assert t.value == 184 and n.value >= 120406977  # Keras tensor count; arena is padded per tensor
    # hidden-size mismatch between the modalities (base_models.py:184-189 ValueError)
    assert lib.fact_arena_size(C.byref(cfg(ha=640)), C.byref(n), C.byref(t)) == -2
    assert b"hidden size" in lib.fact_last_error()
    # unsupported head dim is a clean error, not a crash
    assert lib.fact_arena_size(C.byref(cfg(heads=16)), C.byref(n), C.byref(t)) == -3


def test_product_path_has_no_cpu_fallback():
    import mint_amd.fact_model as fm
    src = open(fm.__file__).read() + open(L.__file__).read()
    assert "oracle" not in src.replace("The oracle under /oracle is test infrastructure", "")
    for mod in ("trainer", "model_builder", "configs", "config_util", "learning_schedules", "protos"):
        path = os.path.join(ROOT, "mint_amd", mod + ".py")
        assert "from oracle" not in open(path).read() and "import oracle" not in open(path).read()


def _accepted(engine, fn):
    body = engine[engine.index("int %s(FactHandle* h, const char* key, int value) {" % fn):]
    body = body[:body.index("\n}\n")]
    return set(re.findall(r'!strcmp\(key, "([a-z0-9_]+)"\)', body))


def test_documented_options_exist_in_the_engine():
    """Every option key a header documents is one that entry point accepts, and every key the engine accepts is
    documented in the header of ITS entry point: the comment blocks are the only specification of the two string-keyed
    interfaces, and the production one accepts nothing but the production keys."""
    engine = open(os.path.join(ROOT, "mint_amd", "csrc", "engine.hip")).read()
    text = open(HEADER).read()
    block = text[text.index("/* Engine options"):text.index("int fact_set_option")]
    assert set(re.findall(r'^ \*   "([a-z0-9_]+)"', block, flags=re.M)) == _accepted(engine, "fact_set_option") == set(L.PUBLIC_OPTIONS)
    dtext = open(DEBUG_HEADER).read()
    dblock = dtext[dtext.index("/* Test / bench knobs"):dtext.index("int fact_debug_set_option")]
    documented = set(re.findall(r'"([a-z0-9_]+)"', dblock))
    accepted = _accepted(engine, "fact_debug_set_option")
    assert documented and documented == accepted, (sorted(documented - accepted), sorted(accepted - documented))
