"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/mli_kernels.h (and include/mli_engine.h) declares, and the Python binding table matches the header.
No compute call is made here (there is no GPU in the build container)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.findall(r"\b(mli_[a-z0-9_]+)\s*\(", text)


def test_library_exports_every_declared_symbol(mli):
    names = _declared("mli_kernels.h") + _declared("mli_engine.h") + _declared("mli_shard.h")
    assert len(set(names)) >= 19
    for n in set(names):
        assert hasattr(mli, n), f"libmli_hip.so does not export {n}"


def test_binding_table_matches_header():
    from min_llm_inference_amd import _lib
    declared = set(_declared("mli_kernels.h"))
    bound = {n for n in _lib.SIGNATURES if not n.startswith(("mli_engine", "mli_shard"))}
    assert declared == bound, (declared ^ bound)


def test_arity_matches_header():
    """Each binding passes exactly as many arguments as the C prototype takes."""
    from min_llm_inference_amd import _lib
    text = open(os.path.join(ROOT, "include", "mli_kernels.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for name, args in re.findall(r"\b(mli_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        n = 0 if args.strip() in ("", "void") else args.count(",") + 1
        assert len(_lib.SIGNATURES[name]) == n, (name, n, len(_lib.SIGNATURES[name]))


def test_abi_version_and_workspace_query(mli):
    assert mli.mli_abi_version() == 4
    # [row arrival counters: 64 KiB, fixed][chunk statistics, padded to 256 bytes][partial sums]
    assert mli.mli_attention_workspace_bytes(4, 64, 64) == 65536 + 256   # single chunk: no partial sums
    assert mli.mli_attention_workspace_bytes(1024, 4096, 512) == 65536 + 1024 * 64 * 8 + 1024 * 64 * 512 * 4
    assert mli.mli_decoder_scratch_bytes(1024, 1024) == 1024 * 32 * 8  # a (value, index) pair per row and 32-column tile
    assert mli.mli_decoder_scratch_bytes(3, 65) == 3 * 3 * 8
    assert mli.mli_attention_workspace_bytes(0, 4096, 512) == 0


def test_missing_library_is_a_hard_error(monkeypatch, tmp_path):
    from min_llm_inference_amd import _lib
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "library_path", lambda: str(tmp_path / "nope.so"))
    try:
        _lib.load_library()
    except _lib.MliError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("load_library() must raise when the HIP library is absent")


def test_cpu_tensors_are_rejected():
    import torch
    from min_llm_inference_amd import ops, MliError
    t = torch.zeros(4, 8)
    try:
        ops._p(t)
    except MliError:
        pass
    else:
        raise AssertionError("host tensors must not cross the C ABI")


def test_public_headers_compile_as_c99_and_cxx17(tmp_path):
    """include/*.h are the drop-in boundary for hosts in any language: plain C (cgo, ctypes, a C host) must be able to include
    them -- no C++ types, no torch types -- and so must the C++ host mirror."""
    import shutil
    import subprocess
    src = tmp_path / "headers.c"
    src.write_text('#include "mli_kernels.h"\n#include "mli_engine.h"\n#include "mli_shard.h"\n'
                   "int main(void) { mli_engine_config c; mli_shard_stats s; (void)c; (void)s; return mli_abi_version() > 0 ? 0 : 1; }\n")
    inc = os.path.join(ROOT, "include")
    assert shutil.which("gcc") and shutil.which("g++")
    for cmd in (["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, str(src)],
                ["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I", inc, "-x", "c++", str(src)]):
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_a_plain_c_host_links_and_calls_the_library(tmp_path):
    """A C program (no C++ runtime of its own, no Python) links libmli_hip.so and calls two entry points that need no GPU:
    what a cgo / FFI host does first."""
    import subprocess
    libdir = os.path.join(ROOT, "min_llm_inference_amd", "lib")
    src = tmp_path / "host.c"
    src.write_text('#include <stdio.h>\n#include "mli_kernels.h"\n'
                   "int main(void) {\n"
                   '    printf("%d %zu\\n", mli_abi_version(), mli_attention_workspace_bytes(1024, 4096, 512));\n'
                   "    return 0;\n}\n")
    exe = tmp_path / "host"
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                        "-L", libdir, "-lmli_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    version, ws = out.stdout.split()
    assert int(version) >= 4 and int(ws) == 65536 + 1024 * 64 * 8 + 1024 * 64 * 512 * 4
