"""libcmdg.so loads on a machine without a GPU and exports every symbol include/cmdg.h
declares (no compute calls here)."""
import ctypes
import os
import re

from cmdg_loader import cm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "cmdg.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(cmdg_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound():
    names = declared_symbols()
    assert len(names) >= 20
    L = ctypes.CDLL(cm._lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), "libcmdg.so lacks %s" % n
    bound = {s[0] for s in cm._lib.SYMBOLS}
    assert set(names) == bound, set(names) ^ bound


def test_queries_work_without_a_gpu():
    L = cm._lib.lib()
    assert b"gfx950" in L.cmdg_version()
    ip = (ctypes.c_int32 * 16)(1, 1, 1, 0)
    out = (ctypes.c_int32 * 6)()
    assert L.cmdg_physics_counts(1, ctypes.cast(ip, ctypes.c_void_p),
                                 ctypes.cast(out, ctypes.c_void_p)) == 0
    assert list(out) == [1, 15, 1, 3, 0, 0]
    assert L.cmdg_physics_counts(99, ctypes.cast(ip, ctypes.c_void_p),
                                 ctypes.cast(out, ctypes.c_void_p)) == -5
    assert L.cmdg_status_string(-5) == b"unsupported physics / polynomial order"


def _header_fields(struct):
    txt = open(os.path.join(ROOT, "include", "cmdg.h")).read()
    body = txt[txt.index("typedef struct %s {" % struct):txt.index("} %s;" % struct)]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    body = body.replace("typedef struct %s {" % struct, "")
    fields = []
    for stmt in body.split(";"):
        stmt = " ".join(stmt.split())
        if not stmt:
            continue
        decl = stmt
        names = re.sub(r"^(const\s+)?\w+\s+", "", decl.strip())
        for nm in names.split(","):
            nm = re.sub(r"\[.*\]", "", nm).replace("*", "").strip()
            if nm:
                fields.append(nm)
    return fields


def test_desc_struct_matches_header_field_order():
    fields = _header_fields("cmdg_desc")
    assert fields == [f[0] for f in cm._lib.CmdgDesc._fields_], fields


def test_filter_desc_struct_matches_header_field_order():
    fields = _header_fields("cmdg_filter_desc")
    F = cm.mesh.filters
    assert fields == [f[0] for f in F.CmdgFilterDesc._fields_], fields
    txt = open(os.path.join(ROOT, "include", "cmdg.h")).read()
    assert "#define CMDG_MAX_FILTER_STATES %d" % F.MAX_FILTER_STATES in txt


def test_stack_integral_desc_struct_matches_header_field_order():
    fields = _header_fields("cmdg_stack_integral_desc")
    assert fields == [f[0] for f in cm._lib.CmdgStackIntegralDesc._fields_], fields
    txt = open(os.path.join(ROOT, "include", "cmdg.h")).read()
    assert "#define CMDG_STACK_MAXOUT %d" % cm._lib.STACK_MAXOUT in txt


def test_rhs_hooks_struct_matches_header_field_order():
    fields = _header_fields("cmdg_rhs_hooks")
    assert fields == [f[0] for f in cm._lib.CmdgRhsHooks._fields_], fields
    txt = open(os.path.join(ROOT, "include", "cmdg.h")).read()
    assert "#define CMDG_MAX_HOOK_OPS %d" % cm._lib.MAX_HOOK_OPS in txt


def test_engine_plugin_builds_and_loads():
    """climatemachine.jl_amd/plugins.py writes a plug-in translation unit, hipcc cross-compiles it
    for gfx950 against libcmdg.so, it exports cmdg_plugin_make_engine and cmdg_load_plugin takes
    it; something that is not a plug-in is refused with the reason.  (Its engine runs on the GPU:
    tests/test_gpu_plugins.py.)"""
    import ctypes as C
    import subprocess
    from cmdg_loader import cm
    so = cm.plugins.build_dry_atmos(orient=True, ref_state=False, hyperdiffusion=True, N=4)
    names = subprocess.check_output(["nm", "-D", so]).decode()
    assert " T cmdg_plugin_make_engine" in names
    L = cm._lib.lib()
    assert L.cmdg_load_plugin(so.encode()) == 0
    assert L.cmdg_load_plugin(so.encode()) == 0          # idempotent
    L.cmdg_last_error.restype = C.c_char_p
    assert L.cmdg_load_plugin(b"/nonexistent/plugin.so") != 0
    assert b"cannot load plug-in" in L.cmdg_last_error(None)


def test_plugin_built_against_other_headers_is_refused(tmp_path):
    """A plug-in shares the C++ layout of the engine base class with the library (cmdg_plugin_abi);
    one whose stamp differs -- or that has none -- is refused at load with the reason."""
    import ctypes as C
    import subprocess
    from cmdg_loader import cm
    L = cm._lib.lib()
    L.cmdg_last_error.restype = C.c_char_p
    for name, body, reason in (
            ("stale", "unsigned long cmdg_plugin_abi(void) { return 1; }", b"another libcmdg"),
            ("unstamped", "", b"does not export cmdg_plugin_abi")):
        src = tmp_path / (name + ".c")
        src.write_text("void *cmdg_plugin_make_engine(const void *d, char *e, int n) { return 0; }\n" + body + "\n")
        so = tmp_path / (name + ".so")
        subprocess.check_call(["gcc", "-shared", "-fPIC", "-o", str(so), str(src)])
        assert L.cmdg_load_plugin(str(so).encode()) != 0
        assert reason in L.cmdg_last_error(None), L.cmdg_last_error(None)
