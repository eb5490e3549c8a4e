"""INTEGRATION.md is code, not prose: every ```cpp block of the document is extracted, concatenated in document order into ONE translation
unit and compiled (g++ -fsyntax-only -Wall -Wextra -Werror) against include/gmmiv.h and tests/integration/alize_stub.h -- a STUB that
declares only the ALIZE / LIA_SpkTools signatures LIA_RAL's own call sites imply and pins nothing about ALIZE.  What this
catches: drift between the documented call-site branches (AccumulateStat.cpp:131-140, AccumulateTVStat.cpp:268-278, :2103-2111,
PldaTools.cpp:4175-4183, ...) and the C ABI -- argument order, counts, types, renamed or removed entry points."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cpp_blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    return re.findall(r"```cpp\n(.*?)```", text, flags=re.S)


def compile_tu(src, tmp_path, extra=()):
    f = tmp_path / "integration_doc.cpp"
    f.write_text(src)
    cmd = ["g++", "-std=c++11", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-Wno-unused-function", "-Wno-unused-parameter",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "integration"), *extra, str(f)]
    return subprocess.run(cmd, capture_output=True, text=True)


def test_every_cpp_block_of_integration_md_compiles_against_the_c_abi(tmp_path):
    blocks = cpp_blocks()
    assert len(blocks) >= 18
    src = '#include "alize_stub.h"\n' + "\n".join("// ---- INTEGRATION.md block %d\n%s" % (i, b) for i, b in enumerate(blocks))
    r = compile_tu(src, tmp_path)
    assert r.returncode == 0, r.stderr[-4000:]
    # the blocks are not comment tables: they call the ABI
    called = set(re.findall(r"\b(gmmiv_[a-z0-9_]+)\s*\(", src))
    declared = set(re.findall(r"\b(gmmiv_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "include", "gmmiv.h")).read()))
    assert called <= declared, sorted(called - declared)
    assert len(called) >= 70, len(called)


def test_the_check_catches_a_drifted_call(tmp_path):
    """swap two arguments of one documented call: the compile must fail (the test above is not vacuous)"""
    blocks = cpp_blocks()
    src = '#include "alize_stub.h"\n' + "\n".join(blocks)
    good = "gmmiv_tv_update_t(_gpu, (int)_n_distrib, (int)_vectSize, (int)_rankT, &_aPacked[0], _Cmx.getArray(), _T.getArray())"
    assert good in src
    bad = src.replace(good, "gmmiv_tv_update_t(_gpu, (int)_n_distrib, (int)_vectSize, &_aPacked[0], (int)_rankT, _Cmx.getArray(), _T.getArray())")
    assert compile_tu(bad, tmp_path).returncode != 0
    gone = src.replace("gmmiv_score_cosine(", "gmmiv_score_cosine_v2(")
    assert compile_tu(gone, tmp_path).returncode != 0


def test_gmmiv_h_is_c99(tmp_path):
    f = tmp_path / "h.c"
    f.write_text('#include "gmmiv.h"\nint main(void) { return 0; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(f)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_c99_example_compiles_and_links_against_the_abi_alone(tmp_path):
    """examples/computetest_llr.c: a plain C99 caller (no HIP headers, no C++) builds with -pedantic -Werror and links against libgmmiv.so
    only -- the boundary is plain pointers and sizes (it is RUN, on the ComputeTest golden, by tests/test_gpu_kat5.py)."""
    exe = tmp_path / "computetest_llr"
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "computetest_llr.c"),
                        "-L", os.path.join(ROOT, "lia_ral_amd", "csrc"), "-lgmmiv", "-Wl,-rpath," + os.path.join(ROOT, "lia_ral_amd", "csrc"),
                        "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run(["ldd", str(exe)], capture_output=True, text=True).stdout
    assert "libgmmiv.so" in out and "libtorch" not in out and "python" not in out
