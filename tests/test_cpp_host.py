"""The C++ facade mirror (include/hip_renderer.hpp) compiles against the C ABI and behaves: scene preparation works on
any host; creating a renderer either succeeds (GPU present) or throws the library's device error (no GPU)."""
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def test_cpp_facade_builds_and_runs(tmp_path, rt):
    exe = tmp_path / "host_smoke"
    libdir = ROOT / "ray-tracer_amd"
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", str(ROOT / "include"), str(ROOT / "tests" / "cpp" / "host_smoke.cpp"),
                    "-L", str(libdir), "-l:librt_hip.so", f"-Wl,-rpath,{libdir}", "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "gpu render ok" in out.stdout or "no gpu" in out.stdout
