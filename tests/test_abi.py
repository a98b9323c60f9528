"""The C-ABI library loads and exports every symbol include/sdfhip.h declares;
struct layouts match the reference's; errors are status codes + messages."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

from conftest import REPO


def declared_symbols(header="sdfhip.h"):
    text = open(os.path.join(REPO, "include", header)).read()
    return sorted(set(re.findall(r"SDFHIP_API[^;]*?\b(sdfhip_\w+)\s*\(", text)))


def exported_symbols(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted({l.split()[-1] for l in out.splitlines() if l.split() and l.split()[-1].startswith("sdfhip_")})


def test_the_product_exports_exactly_its_header(sb):
    # libsdfhip.so == include/sdfhip.h, symbol for symbol: the laboratory (A/B kernel forms, superseded gather formats, test
    # hooks) is a library of its own
    names = declared_symbols()
    assert len(names) >= 40
    assert exported_symbols(sb._lib.LIB_PATH) == names
    assert os.path.basename(sb._lib.LIB_PATH) == "libsdfhip.so" and not sb._lib.EXPERIMENTS
    # and the Python binding covers exactly the declared set
    assert sorted(sb._lib.EXPORTED_SYMBOLS) == names
    for n in names:
        assert hasattr(sb._lib.lib, n)


def test_the_experiments_build_exports_both_headers(sb):
    lab = declared_symbols("sdfhip_experimental.h")
    assert 8 <= len(lab) <= 14 and not set(lab) & set(declared_symbols())
    assert exported_symbols(sb._lib.LAB_LIB_PATH) == sorted(declared_symbols() + lab)
    assert sorted(sb._lib.EXPERIMENTAL_SYMBOLS) == lab
    import sdfbox_amd.lab
    sbx = sdfbox_amd.lab.load()
    assert sbx._lib.EXPERIMENTS and sbx.Info is sb.Info and sbx._lib.lib is not sb._lib.lib      # one set of ctypes classes, two libraries


def test_flag_constants_match_the_headers(sb):
    # every SDFHIP_FLAG_* / SDFHIP_TUNE_* / SDFHIP_KERNEL_* enumerator of the two headers that the Python mirror names has the
    # header's value, no two render flags share a bit, and the product's header carries no A/B knob
    def enums(header):
        text = open(os.path.join(REPO, "include", header)).read()
        return {m.group(1): int(m.group(2), 0) for m in re.finditer(r"\bSDFHIP_((?:FLAG|TUNE|KERNEL)_\w+)\s*=\s*(0x[0-9A-Fa-f]+|\d+)", text)}
    product, lab = enums("sdfhip.h"), enums("sdfhip_experimental.h")
    assert {"FLAG_COMPACT", "FLAG_COUNT", "FLAG_DISPLAY", "FLAG_DISPLAY_DEBUG", "FLAG_TILE_ORDER", "KERNEL_AUTO", "KERNEL_GENERIC", "KERNEL_STACK", "KERNEL_MASK"} == set(product)
    assert {"FLAG_WIRE", "TUNE_ONE_KERNEL", "TUNE_SHADOW_QUEUE", "TUNE_LDS_TOP", "TUNE_BYTE_CELLS", "TUNE_PERSISTENT_WAVES", "TUNE_ORDER_SHIFT", "TUNE_BLOCK_SHIFT"} == set(lab)
    header = dict(product, **lab)
    L = sb._lib
    seen = 0
    for name, value in header.items():
        if hasattr(L, name):
            assert getattr(L, name) == value, name
            seen += 1
    assert seen >= 14
    bits = [v for k, v in header.items() if k.startswith(("FLAG_", "TUNE_")) and not k.endswith("_SHIFT")]
    assert len(bits) == len(set(bits)) and all(v & (v - 1) == 0 for v in bits)      # single, distinct bits


def test_info_layout_matches_logic_cs(sb):
    # Logic.cs:407-420 (Pack 16, Size 112) / Compute.hlsl:70-81
    I = sb.Info
    assert ctypes.sizeof(I) == 112
    off = {f: getattr(I, f).offset for f, _ in I._fields_}
    assert off["heading"] == 0 and off["position"] == 48 and off["margin"] == 60
    assert off["screen_size"] == 64 and off["buffer_size"] == 72 and off["limit"] == 76
    assert off["light"] == 80 and off["strength"] == 92 and off["fov"] == 96 and off["hidef"] == 100


def test_errors_are_codes_not_crashes(sb, tmp_path):
    L = sb._lib
    raw = L.COctData()
    rc = L.lib.sdfhip_asdf_load(os.fsencode(str(tmp_path / "nope.asdf")), ctypes.byref(raw))
    assert rc == L.ERR_IO and b"could not open" in L.lib.sdfhip_last_error()
    assert L.lib.sdfhip_asdf_load(None, ctypes.byref(raw)) == L.ERR_ARG
    with pytest.raises(sb.SdfHipError) as e:
        sb.OctData.Generate(7, [0.0] * 4, 3)
    assert e.value.code == L.ERR_ARG
    with pytest.raises(sb.SdfHipError):
        sb.OctData.Generate(L.SHAPE_SPHERE, [0.5, 0.5, 0.5], 3)       # wrong param count
    with pytest.raises(sb.SdfHipError):
        sb.OctData.Generate(L.SHAPE_SPHERE, [0.5, 0.5, 0.5, 0.3], 13)  # deeper than the shader can descend


def test_validate_rejects_out_of_range_links(sb):
    od = sb.sphere_d4()
    s = od.Structs.copy()
    s[5, 1] = od.Length - 3          # children block would run past the end
    with pytest.raises(sb.SdfHipError) as e:
        sb.OctData(s, od.Values).validate()
    assert e.value.code == sb._lib.ERR_BAD_TREE
    s = od.Structs.copy()
    s[9, 0] = od.Length + 10         # parent out of range
    with pytest.raises(sb.SdfHipError):
        sb.OctData(s, od.Values).validate()
    # parent cycles and endless parent chains would hang the shader's ascend loop: rejected
    s = od.Structs.copy()
    s[9, 0] = 17; s[17, 0] = 9
    with pytest.raises(sb.SdfHipError) as e:
        sb.OctData(s, od.Values).validate()
    assert e.value.code == sb._lib.ERR_BAD_TREE and "chain" in str(e.value)
    s = od.Structs.copy()
    s[1:200, 0] = np.arange(0, 199)            # node i's parent is node i-1: a 199-link chain
    with pytest.raises(sb.SdfHipError):
        sb.OctData(s, od.Values).validate()
    s = od.Structs.copy()
    s[0, 0] = 0                                # the root is its own parent
    with pytest.raises(sb.SdfHipError):
        sb.OctData(s, od.Values).validate()
    # in range but inconsistent: legal for the generic kernel, not for the stack kernel
    s = od.Structs.copy()
    s[9, 0] = 3
    depth, consistent = sb.OctData(s, od.Values).validate()
    assert not consistent
    assert od.validate() == (4, True)


def test_product_fails_loudly_without_a_gpu(sb):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(sb.SdfHipError) as e:
        sb.Scene(sb.sphere_d4())
    assert e.value.code == sb._lib.ERR_DEVICE


def test_product_never_imports_the_oracle():
    # the oracle is test infrastructure: no file of the package may import, include,
    # link or load it (comments may cite it)
    pkg = os.path.join(REPO, "sdfbox_amd")
    pat = re.compile(r"^\s*(import\s+oracle|from\s+oracle\b)|#\s*include\s*[<\"][^>\"]*oracle|liboracle|-loracle", re.M)
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")) or f == "Makefile":
                text = open(os.path.join(root, f), errors="replace").read()
                assert not pat.search(text), os.path.join(root, f)


def _c_prototypes():
    """name -> (return type, [parameter types]) of every SDFHIP_API declaration of include/sdfhip.h"""
    text = re.sub(r"/\*.*?\*/", " ", open(os.path.join(REPO, "include", "sdfhip.h")).read(), flags=re.S)
    protos = {}
    for m in re.finditer(r"SDFHIP_API\s+([^;(]+?)\b(sdfhip_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        params = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        protos[name] = (ret, params)
    return protos


def _c_class(t):
    """pointer / the scalar's width class of a C parameter (its name stripped)"""
    if "*" in t:
        return "ptr"
    base = re.sub(r"\b(const|unsigned)\b", "", t).split()
    base = base[0] if base else ""
    return {"uint32_t": "u32", "int32_t": "i32", "int": "i32", "uint64_t": "u64", "float": "f32"}.get(base, base)


def _cs_class(t):
    t = re.sub(r"^\s*\[[A-Za-z][^\]]*\]\s*", "", t).strip()      # a leading attribute: [Out]
    words = t.split()
    if words[0] in ("ref", "out") or words[-2].endswith("[]") or words[0] in ("string", "IntPtr", "System.Text.StringBuilder"):
        return "ptr"
    return {"uint": "u32", "int": "i32", "ulong": "u64", "float": "f32"}.get(words[0], words[0])


def test_csharp_stub_matches_the_header():
    """INTEGRATION.md's [DllImport] declarations cannot be compiled here (no .NET in the image): at least they must name
    entry points the header declares, with the same number of parameters, each a pointer where C has a pointer and a scalar
    of the same width where C has a scalar, and the same kind of return value."""
    protos = _c_prototypes()
    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    decls = re.findall(r"\[DllImport\(Lib(?:,\s*EntryPoint\s*=\s*\"(\w+)\")?\)\]\s*(?:public\s+)?static\s+extern\s+(\w+)\s+(\w+)\s*\(([^;]*)\)\s*;", text)
    assert len(decls) >= 25, len(decls)
    for entry, ret, name, args in decls:
        cname = entry or name
        assert cname in protos, f"the C# stub imports {cname}, which include/sdfhip.h does not declare"
        cret, cparams = protos[cname]
        params = [a.strip() for a in re.sub(r"\[MarshalAs\([^)]*\)\]", "", args).split(",")] if args.strip() else []
        assert len(params) == len(cparams), f"{cname}: {len(params)} parameters in C#, {len(cparams)} in C"
        for a, c in zip(params, cparams):
            assert _cs_class(a) == _c_class(c), f"{cname}: C# `{a}` against C `{c}`"
        want = "IntPtr" if "*" in cret else {"int": "int", "void": "void", "float": "float"}[cret.split()[-1]]
        assert ret == want, f"{cname}: returns {ret} in C#, {cret} in C"


def test_a_failed_load_of_the_laboratory_package_leaves_nothing_behind(monkeypatch):
    # sdfbox_amd.lab.load() imports the package a second time against libsdfhip_lab.so; when that import fails half-way the package
    # and the submodules it had already imported must leave sys.modules, or a retry meets a half-initialised package
    import importlib.util
    import sdfbox_amd.lab as lab
    saved = {k: v for k, v in sys.modules.items() if k == lab._NAME or k.startswith(lab._NAME + ".")}
    for k in saved:
        del sys.modules[k]
    real = importlib.util.module_from_spec

    def broken(spec):
        mod = real(spec)
        sys.modules[lab._NAME + "._lib"] = object()            # "a submodule was imported before the failure"

        class Loader:
            def exec_module(self, m):
                raise RuntimeError("import failed half-way")
        spec.loader = Loader()
        return mod
    monkeypatch.setattr(importlib.util, "module_from_spec", broken)
    try:
        with pytest.raises(RuntimeError):
            lab.load()
        assert not [k for k in sys.modules if k == lab._NAME or k.startswith(lab._NAME + ".")]
    finally:
        monkeypatch.undo()
        sys.modules.update(saved)
    assert lab.load()._lib.EXPERIMENTS                         # and a load after the failure works


def test_every_entry_point_is_behind_the_exception_firewall():
    """the C ABI never throws or aborts across the boundary.  Every `extern "C"` DEFINITION under csrc/ (both flavours of
    the library) is a function-try-block closed by one of csrc/abi_guard.h's macros naming that entry point -- so a std::bad_alloc,
    a std::system_error or anything else a callee throws comes back as a status code -- and every thread the library starts has a
    noexcept body or catches everything.  The one exception is sdfhip_last_error (it returns a thread-local array).  What the
    guard DOES under failure is tests/test_host_sanitizers.py::test_injected_failures_come_back_as_status_codes."""
    src = os.path.join(REPO, "sdfbox_amd", "csrc")
    guarded, unguarded = set(), []
    for f in sorted(os.listdir(src)):
        if not f.endswith((".cpp", ".hip", ".h")):
            continue
        text = open(os.path.join(src, f)).read()
        for m in re.finditer(r'^extern "C"[^;{]*?\b(sdfhip_\w+)\s*\([^;{]*?\)\s*(try\s*)?\{', text, re.M | re.S):
            name, is_try = m.group(1), bool(m.group(2))
            if name == "sdfhip_last_error":
                continue
            # the block's end: the first line that is exactly "}" after the opening, then the macro with this very name
            end = re.compile(r"^\}\n(SDFHIP_ABI_CATCH(?:_VOID|_AS)?)\((\w+)", re.M).search(text, m.end())
            if not is_try or not end or end.group(2) != name:
                unguarded.append(f"{f}: {name}")
            guarded.add(name)
        # threads: std::thread bodies must not let an exception out
        for m in re.finditer(r"std::thread\(\[[^\]]*\]\s*\{([^}]*)\}", text):
            body = m.group(1)
            assert "run()" in body, f"{f}: a thread body the firewall test does not know: {body!r}"
        if "void run()" in text:
            assert "void run() noexcept" in text, f"{f}: Worker::run must be noexcept with a catch-all inside"
    assert not unguarded, "entry points outside the firewall: " + ", ".join(unguarded)
    declared = set(declared_symbols()) | set(declared_symbols("sdfhip_experimental.h"))
    assert declared - {"sdfhip_last_error"} <= guarded, sorted(declared - guarded)
    # the scene builder's pool: its threads' bodies catch everything themselves
    gen = open(os.path.join(src, "scene_gen.cpp")).read()
    assert "catch (...) { err[s] = 1; }" in gen


def test_loading_the_library_exports_the_hardware_queue_count_unless_the_host_decided(sb):
    # a host that keeps frames in flight is 20 % slower on the runtime's four hardware queues; the runtime reads
    # GPU_MAX_HW_QUEUES at the first HIP call, so the library exports 8 when it is loaded -- never over the host's own value, and
    # not at all with SDFHIP_KEEP_ENV (csrc/errors.cpp)
    import subprocess
    prog = ("import ctypes, sys; ctypes.CDLL(sys.argv[1]); g = ctypes.CDLL(None).getenv; g.restype = ctypes.c_char_p; "
            "print(g(b'GPU_MAX_HW_QUEUES'))")
    base = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "SDFHIP_KEEP_ENV")}
    for path in (sb._lib.LIB_PATH, sb._lib.LAB_LIB_PATH):
        for extra, want in (({}, "b'8'"), ({"GPU_MAX_HW_QUEUES": "2"}, "b'2'"), ({"SDFHIP_KEEP_ENV": "1"}, "None")):
            out = subprocess.run([sys.executable, "-c", prog, path], env=dict(base, **extra), capture_output=True, text=True, timeout=120)
            assert out.returncode == 0 and out.stdout.strip() == want, (path, extra, out.stdout, out.stderr[-500:])


def test_the_binaries_carry_the_firewall(sb):
    # ... and the same from the BINARIES ("nm shows no entry point without the guard"): every exported entry
    # point's machine code -- its body or its .cold part -- calls sdfhip::abi_caught from its catch handler, except the few whose
    # handler the compiler removed because nothing in them can throw (arithmetic on the caller's struct, free(), a thread-local
    # array): that list is closed and spelled out here.
    import subprocess
    nothrow = {"sdfhip_last_error", "sdfhip_camera_mouse_wheel", "sdfhip_info_set_heading", "sdfhip_info_set_position", "sdfhip_octdata_free",
               "sdfhip_points_free", "sdfhip_sparse2_bytes", "sdfhip_sparse2_floats_offset", "sdfhip_upload_options_default",
               # laboratory only
               "sdfhip_wire_sparse_bytes", "sdfhip_wire_sparse_head_offset", "sdfhip_debug_fail_host_allocations"}
    for path in (sb._lib.LIB_PATH, sb._lib.LAB_LIB_PATH):
        out = subprocess.run(["objdump", "-d", "--no-show-raw-insn", path], capture_output=True, text=True, check=True).stdout
        cur, guarded = None, set()
        for line in out.splitlines():
            m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
            if m:
                cur = re.sub(r"\.cold.*|\[clone.*", "", m.group(1)).strip()
            elif cur and "abi_caught" in line and "call" in line:
                guarded.add(cur)
        exported = set(exported_symbols(path))
        assert exported - guarded <= nothrow, (os.path.basename(path), sorted(exported - guarded - nothrow))
        assert len(exported & guarded) >= 45
