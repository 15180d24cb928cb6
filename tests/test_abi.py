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
    return sorted(set(re.findall(r"\b(?:int|int64_t)\s+(lako_\w+)\s*\(", src)))


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
            decl = re.sub(r"^(const\s+)?(void|float|int64_t|int32_t|int|uint8_t|uint32_t|lako_dropout_t|lako_tuning_t)\b", "", decl)
            names += [n.strip(" *") for n in decl.split(",")]
        assert names == [f[0] for f in py._fields_], (cname, names)


def test_bad_arguments_return_error_codes(lib):
    """Validation happens on the host before any launch, so it can be exercised without a GPU."""
    from lako_amd import _lib
    _lib._lib = None
    L = _lib.load()
    assert L.lako_version() == 4 == _lib.ABI_VERSION
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
    t = _lib.Tuning()
    assert L.lako_tuning_init(ctypes.byref(t)) == 0 and L.lako_tuning_set(ctypes.byref(t), b"no_such_knob", 1) == -1
    assert L.lako_tuning_init(None) == -1 and L.lako_tuning_set(None, b"gemm_nt_variant", 1) == -1
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
    # the product-quantised index and the retriever's backward (round 3)
    assert L.lako_pq_assign(x, 100, 256, x, 16, 256, 5, x, None, None, None, None) == -1       # sub-vector length 5: no instantiation
    assert L.lako_pq_assign(x, 100, 256, x, 16, 256, 16, None, None, None, None, None) == -1   # neither codes nor sums
    assert L.lako_pq_assign(x, 100, 256, x, 16, 256, 16, x, x, None, None, None) == -1         # sums without counts
    assert L.lako_pq_assign(x, 100, 128, x, 16, 256, 16, x, None, None, None, None) == -1      # row stride < M * dsub
    assert L.lako_pq_lut(x, 8, 256, x, 16, 256, 128, x, None) == -1                            # sub-vector longer than 64
    assert L.lako_pq_scan(x, x, 1000, 8, 256, 256, x, 1000, None) == -1                        # one query's table > 128 KiB of LDS
    assert L.lako_pq_scan(x, x, 1000, 8, 16, 256, x, 999, None) == -1                          # ld < n
    assert L.lako_kldiv_bwd(x, x, x, None, 0, 8, None) == -1
    assert L.lako_layernorm_bwd(x, x, None, None, x, x, x, x, None, 16, 2000, 1e-5, 1, None) == -1   # d > the kernel's register budget


def test_comm_and_workspace_entry_points_validate_their_arguments(lib):
    """(round 5) lako_comm_* / lako_allreduce / lako_workspace_bytes exist behind the C-ABI; on a host without a GPU only
    their argument validation runs: null handles and bad ranks are LAKO_E_BADARG (-1), an op without scratch wants 0 bytes."""
    lib.lako_workspace_bytes.restype = ctypes.c_int64
    lib.lako_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_void_p]
    assert lib.lako_workspace_bytes(0, None) == 0 and lib.lako_workspace_bytes(12345, None) == 0
    assert lib.lako_workspace_bytes(1, None) == -1                      # LAKO_WS_GEMM_TN_GROUPED without its arguments
    lib.lako_comm_init.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    h = ctypes.c_void_p()
    idb = (ctypes.c_uint8 * 128)()
    assert lib.lako_comm_init(None, 0, 1, idb) == -1
    assert lib.lako_comm_init(ctypes.byref(h), 0, 1, None) == -1
    assert lib.lako_comm_init(ctypes.byref(h), 2, 2, idb) == -1 and lib.lako_comm_init(ctypes.byref(h), 0, 0, idb) == -1 and not h.value
    lib.lako_allreduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
    assert lib.lako_allreduce(None, None, 4, 0, None) == -1
    lib.lako_comm_destroy.argtypes = [ctypes.c_void_p]
    assert lib.lako_comm_destroy(None) == 0                             # destroying nothing is fine
    lib.lako_comm_world_size.argtypes = [ctypes.c_void_p]
    assert lib.lako_comm_world_size(None) == -1
    assert lib.lako_comm_unique_id(None) == -1


def test_exclusive_weight_gradient_modes_refuse_overlapping_outputs(lib):
    """lako_gemm_tn_grouped with split_k < 0 promises ONE contributor per element of C (plain read-modify-write, overwrite, the hybrid
    schedule): two items whose output ranges overlap would race or overwrite each other silently — refused before anything is
    planned or launched."""
    from lako_amd import _lib
    base = 0x7F0000000000
    def items(c1):
        arr = (_lib.GemmTNItem * 2)()
        for it, c in zip(arr, (base, c1)):
            it.a, it.b, it.c = base + (1 << 30), base + (2 << 30), c
            it.M, it.N, it.lda, it.ldb, it.ldc, it.alpha, it.rows_out = 256, 256, 256, 256, 256, 1.0, 0
        return arr
    lib.lako_gemm_tn_grouped.restype = ctypes.c_int
    call = lambda arr, sk: lib.lako_gemm_tn_grouped(arr, 2, 1024, 0, sk, None, None, 0, None)      # noqa: E731
    for sk in (-1, -2, -3):
        assert call(items(base), sk) == -1                                # the same C twice
        assert call(items(base + 255 * 256 * 4 + 16), sk) == -1           # second C starts inside the first's last row
        buf = ctypes.create_string_buffer(512)
        lib.lako_last_error.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
        lib.lako_last_error(buf, 512)
        assert b"overlap" in buf.value, buf.value
    # (disjoint outputs — every grouped launch of the training step — run in the GPU suite; nothing is launched from this host)


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
    t = _lib.Tuning()
    assert L.lako_tuning_init(ctypes.byref(t)) == 0
    for key, val in ((b"gemm_nt_debug", 1), (b"gemm_nt_debug", 16), (b"gemm_nt_store_aux", 2), (b"gemm_tn_big", 2)):
        assert L.lako_tuning_set(ctypes.byref(t), key, val) == -4, key
        buf = ctypes.create_string_buffer(256)
        L.lako_last_error(buf, 256)
        assert b"LAKO_EXPERIMENTS" in buf.value
    assert (t.nt_debug, t.nt_store_aux, t.tn_big) == (0, 0, 1)                  # a refused key leaves the struct untouched
    for key, val in ((b"gemm_nt_variant", -1), (b"gemm_nt_group_m", 8), (b"gemm_tn_big", 1), (b"gemm_nt_tail_split", 1)):
        assert L.lako_tuning_set(ctypes.byref(t), key, val) == 0, key       # result-preserving kernel selection stays available
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
        if depth == 0 and re.search(r"\ba\.debug\s*&|dbg_flags\s*&|nt_debug\s*>>", line):
            assert False, f"ungated experiment test: {t}"


def test_tuning_is_caller_owned_and_the_library_keeps_no_mutable_knobs(lib, monkeypatch):
    """re-entrant, no mutable globals except once-initialised caches (fifteen process-global
    tuning ints).  The knobs now live in a caller-owned lako_tuning_t: two structs are independent, the defaults are the documented
    ones, LAKO_TUNING is parsed into the caller's struct (bad strings are an error, experiment keys refused), the ctypes layout
    matches the header, and the shared object exports / defines no g_nt_* / g_tn_* data symbol any more."""
    import subprocess
    from lako_amd import _lib
    _lib._lib = None
    L = _lib.load()
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    body = re.search(r"typedef struct lako_tuning \{([^{}]*)\}\s*lako_tuning_t", src, re.S).group(1)
    names = [re.sub(r"^int32_t\s+", "", d.strip()) for d in body.split(";") if d.strip()]
    assert names == [f[0] for f in _lib.Tuning._fields_[:-1]] + ["reserved[10]"]
    assert ctypes.sizeof(_lib.Tuning) == 32 * 4
    monkeypatch.delenv("LAKO_TUNING", raising=False)
    a, b = _lib.Tuning(), _lib.Tuning()
    assert L.lako_tuning_init(ctypes.byref(a)) == 0 and L.lako_tuning_init(ctypes.byref(b)) == 0
    assert (a.nt_variant, a.nt_tail_split, a.nt_ring, a.nt_skinny, a.nt_side_lds, a.nt_wide_epi, a.nt_group_m, a.nt_persistent,
            a.nt_stagger, a.nt_dephase, a.nt_dephase_n, a.tn_big, a.tn_split, a.nt_debug, a.nt_store_aux, a.nt_tile192, a.nt_queue, a.nt_pp, a.nt_glds, a.nt_tile288, a.nt_four, a.tn_four) == \
        (-1, 1, 1, 1, 1, 1, 8, 1, 1, 100, 2, 1, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1)
    assert L.lako_tuning_set(ctypes.byref(a), b"gemm_nt_variant", 2) == 0 and L.lako_tuning_set(ctypes.byref(a), b"gemm_nt_dephase_n", 0) == 0
    assert (a.nt_variant, a.nt_dephase_n, b.nt_variant, b.nt_dephase_n) == (2, 2, -1, 2)
    monkeypatch.setenv("LAKO_TUNING", "gemm_nt_variant=0, gemm_tn_split = 3")
    c = _lib.Tuning()
    assert L.lako_tuning_init(ctypes.byref(c)) == 0 and (c.nt_variant, c.tn_split, c.nt_group_m) == (0, 3, 8)
    for bad, rc in (("gemm_nt_variant", -1), ("nonsense=1", -1), ("gemm_nt_debug=1", -4)):
        monkeypatch.setenv("LAKO_TUNING", bad)
        assert L.lako_tuning_init(ctypes.byref(_lib.Tuning())) == rc, bad
    out = subprocess.run(["nm", "-C", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert not re.search(r"\bg_(nt|tn)_\w+", out)
    assert "lako_set_tuning" not in out


def test_build_force_recompiles_every_source():
    """__graft_entry__.build() must exercise the compiler, not just relink stale objects (build.sh --force)."""
    script = open(os.path.join(ROOT, "lako_amd", "csrc", "build.sh")).read()
    assert "--force" in script and "FORCE = 1" in script.replace("$FORCE", "FORCE")
    entry = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert '"--force"' in entry


def test_host_side_validation_under_address_sanitizer():
    """SURVEY.md §5.2 / VERDICT round 2: the library's host code — argument validation, error plumbing, the tuning parser — under
    AddressSanitizer (`LAKO_ASAN=1 csrc/build.sh` → liblako_hip_asan.so; device code is not instrumented: GPU ASan is not available
    on this pool).  The CPU-side ABI checks of this file run in a child process against that build with the ASan runtime preloaded;
    any heap / stack / global overflow or use-after-free in the host paths they exercise aborts the child."""
    import glob
    import shutil
    import subprocess
    import sys
    if not shutil.which("hipcc") and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    rt = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    if not rt:
        pytest.skip("no ASan runtime in the ROCm toolchain")
    r = subprocess.run(["bash", os.path.join(ROOT, "lako_amd", "csrc", "build.sh")], env=dict(os.environ, LAKO_ASAN="1"), capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lib = os.path.join(ROOT, "lako_amd", "liblako_hip_asan.so")
    env = dict(os.environ, LD_PRELOAD=rt[0], ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0:abort_on_error=1", LAKO_LIB=lib)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-p", "no:cacheprovider", "-k",
                        "bad_arguments or rejects_timing or caller_owned"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0 and "AddressSanitizer" not in r.stderr and "AddressSanitizer" not in r.stdout, r.stdout[-2500:] + r.stderr[-2500:]
    assert "3 passed" in r.stdout, r.stdout[-500:]


def test_documents_state_the_current_number_of_entry_points():
    """README.md / DESIGN.md / INTEGRATION.md quote the size of the C-ABI; keep the quote equal to what include/lako_hip.h declares."""
    n = len(header_functions())
    for doc in ("README.md", "DESIGN.md", "INTEGRATION.md"):
        text = open(os.path.join(ROOT, doc)).read()
        quoted = {int(m) for m in re.findall(r"(\d+)\s+(?:`extern \"C\"`\s+|extern \"C\"\s+)?(?:entry points|functions taking raw device pointers)", text)}
        assert quoted and quoted == {n}, (doc, quoted, n)
