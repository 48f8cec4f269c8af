"""The static tile schedule of the whole-step decode kernels (zonos_amd/csrc/zn_step_sched.h: which slot reads which tile from which LDS
park slot or register buffer, which request each slot raises and when) replayed on the CPU: tests/step_schedule_check.cpp walks a block
the way a compute wave does and checks that every slot reads the tile it should, that no request overwrites an unconsumed buffer, that
every tile is requested once and consumed once (op 0's twice), and that the deferral / early-request rules hold - for the shipped
instantiation and for every variant DESIGN.md section 4.1 reports a measurement of.  The schedule is compile-time arithmetic shared
verbatim with the kernels (the header has no HIP in it), so this is the kernels' schedule, not a model of it."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_static_schedule_replays_without_hazards(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    exe = tmp_path / "step_schedule_check"
    subprocess.run([gxx, "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "zonos_amd", "csrc"),
                    os.path.join(ROOT, "tests", "step_schedule_check.cpp"), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "FAIL" not in r.stdout and r.stdout.count("\n") >= 13
