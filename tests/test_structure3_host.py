"""Static structure tables of the N = 19 / 25 QP kernels (csrc/structure3.hpp): host code, checked on the CPU by a small
C++ program (tests/host/structure3_check.cpp) — permutation, canonical slot order of the sparse K_JC (what the loop
kernel's "base + immediate" addressing relies on), completeness of the term lists, the sizes the ELL table is built from."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_structure3_tables(tmp_path):
    exe = str(tmp_path / "structure3_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-DMPCMP_HD=", "-o", exe, os.path.join(ROOT, "tests", "host", "structure3_check.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ok 6" in out.stdout and "ok 8" in out.stdout
