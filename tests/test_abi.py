"""CPU checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports exactly the
entry points include/lako_hip.h declares (no compute calls — there is no GPU here); the Python binding table
covers every one of them; argument validation returns error codes instead of crashing."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "lako_hip.h")
LIB = os.path.join(ROOT, "lako_amd", "liblako_hip.so")


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\bint\s+(lako_\w+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        subprocess.run(["bash", os.path.join(ROOT, "lako_amd", "csrc", "build.sh")], check=True)
    return ctypes.CDLL(LIB)


def test_header_declares_the_expected_surface():
    fns = header_functions()
    for must in ("lako_gemm_nt", "lako_gemm_tn", "lako_attn_fwd", "lako_attn_bwd", "lako_rmsnorm_fwd",
                 "lako_rmsnorm_bwd", "lako_embed_fwd", "lako_embed_bwd", "lako_ce_fwd_bwd", "lako_adamw_step",
                 "lako_sumsq", "lako_greedy_step", "lako_relpos_expand", "lako_relpos_reduce", "lako_last_error"):
        assert must in fns


def test_library_exports_every_declared_symbol(lib):
    out = subprocess.run(["nm", "-D", "--defined-only", LIB], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\bT\s+(lako_\w+)", out))
    declared = set(header_functions())
    assert declared <= exported, f"declared but not exported: {sorted(declared - exported)}"
    assert exported <= declared, f"exported but not declared in the header: {sorted(exported - declared)}"
    for name in declared:
        getattr(lib, name)


def test_binding_table_matches_header():
    from lako_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_functions()


def test_struct_layouts_match_header_order():
    """ctypes field order must follow the C structs (same names, same order)."""
    from lako_amd import _lib
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for cname, py in (("lako_gemm_nt_t", _lib.GemmNT), ("lako_attn_fwd_t", _lib.AttnFwd), ("lako_attn_bwd_t", _lib.AttnBwd),
                      ("lako_dropout_t", _lib.Dropout), ("lako_headbatch_t", _lib.HeadBatch),
                      ("lako_gemm_tn_item_t", _lib.GemmTNItem)):
        body = re.search(r"typedef struct \{([^{}]*)\}\s*" + cname, src, re.S).group(1)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            decl = re.sub(r"^(const\s+)?(void|float|int64_t|int32_t|int|uint8_t|uint32_t|lako_dropout_t)\b", "", decl)
            names += [n.strip(" *") for n in decl.split(",")]
        assert names == [f[0] for f in py._fields_], (cname, names)


def test_bad_arguments_return_error_codes(lib):
    """Validation happens on the host before any launch, so it can be exercised without a GPU."""
    from lako_amd import _lib
    _lib._lib = None
    L = _lib.load()
    assert L.lako_version() == 1
    p = _lib.GemmNT()               # all zeros: M = N = K = 0
    rc = L.lako_gemm_nt(ctypes.byref(p), None)
    assert rc == -1
    buf = ctypes.create_string_buffer(256)
    n = L.lako_last_error(buf, 256)
    assert n > 0 and b"lako_gemm_nt" in buf.value
    a = _lib.AttnFwd()
    a.Bn, a.H, a.Lq, a.Lk, a.d_head, a.dtype = 1, 1, 4, 4, 48, 1      # unsupported head dim
    assert L.lako_attn_fwd(ctypes.byref(a), None) == -4
    assert L.lako_rmsnorm_fwd(None, None, None, None, 4, 12, 1e-6, 1, _lib.NO_DROP, None) == -1   # d % 8 != 0
    assert L.lako_set_tuning(b"no_such_knob", 1) == -1
    # the encoder-space cross-attention entry points (round 2)
    hb = _lib.HeadBatch()
    assert L.lako_headbatch_nt(ctypes.byref(hb), None) == -1 and L.lako_headbatch_tn(ctypes.byref(hb), None) == -1
    assert L.lako_xattn_scores(None, 0, 0, None, 0, None, None, 256, None, 256, 96, 768, 16, None) == -1      # null operands
    x = ctypes.c_void_p(4096)         # never dereferenced: the shape checks come first
    assert L.lako_xattn_scores(x, 8, 8, x, 800, x, x, 256, x, 256, 96, 800, 16, None) == -1                    # d_model % 128
    assert L.lako_xattn_context(x, 256, x, 768, x, x, x, 0, 0, 768, 96, 768, 16, 0, None) == -1                # key_splits < 1
    assert L.lako_xattn_decode(x, 8, 8, x, 768, x, x, x, 17, 768, 16, 16, None) == -1                          # R > 16
    assert L.lako_xattn_decode_combine(x, x, x, 768, x, 768, 12, 640, 16, 16, None) == -1                      # D not in {512, 768, 1024}
    assert L.lako_xattn_softmax_fwd(x, 255, x, x, 256, x, x, 16, 8, 12, 4000, _lib.NO_DROP, None) == -1        # s_ld % 4


def test_product_fails_loudly_without_library(monkeypatch, tmp_path):
    from lako_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(_lib.LakoError, match="no CPU fallback"):
        _lib.load(str(tmp_path / "missing.so"))


def test_release_library_rejects_timing_experiments(lib):
    """The knobs that switch off DMA / stores / atomics (wrong results) or change the stores' cache policy exist only in
    the -DLAKO_EXPERIMENTS build: the shipped library refuses them, so a stray LAKO_TUNING cannot corrupt training."""
    from lako_amd import _lib
    _lib._lib = None
    L = _lib.load()
    for key, val in ((b"gemm_nt_debug", 1), (b"gemm_nt_debug", 16), (b"gemm_nt_store_aux", 2), (b"gemm_tn_big", 2)):
        assert L.lako_set_tuning(key, val) == -4, key
        buf = ctypes.create_string_buffer(256)
        L.lako_last_error(buf, 256)
        assert b"LAKO_EXPERIMENTS" in buf.value
    for key, val in ((b"gemm_nt_variant", -1), (b"gemm_nt_group_m", 8), (b"gemm_tn_big", 1), (b"gemm_nt_tail_split", 1)):
        assert L.lako_set_tuning(key, val) == 0, key       # result-preserving kernel selection stays available
    src = open(os.path.join(ROOT, "lako_amd", "csrc", "gemm.hip")).read() + open(os.path.join(ROOT, "lako_amd", "csrc", "attn.hip")).read()
    assert "gemm_nt_pipe_kernel" not in src
    # no un-gated use of the experiment bits is left in the sources
    depth = 0
    for line in src.splitlines():
        t = line.strip()
        if t.startswith("#ifdef LAKO_EXPERIMENTS"):
            depth += 1
        elif t.startswith("#else") or t.startswith("#endif"):
            depth = max(depth - 1, 0) if t.startswith("#endif") or depth else depth
        if depth == 0 and re.search(r"\ba\.debug\s*&|dbg_flags\s*&|g_nt_debug\s*>>", line):
            assert False, f"ungated experiment test: {t}"


def test_build_force_recompiles_every_source():
    """__graft_entry__.build() must exercise the compiler, not just relink stale objects (build.sh --force)."""
    script = open(os.path.join(ROOT, "lako_amd", "csrc", "build.sh")).read()
    assert "--force" in script and "FORCE = 1" in script.replace("$FORCE", "FORCE")
    entry = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert '"--force"' in entry
