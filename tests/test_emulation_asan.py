"""CPU tier: the emulated kernels under AddressSanitizer on exact-size heap buffers.  The GPU reports a stray access as
"Memory access fault" without a location; here it is a stack trace.  (Found this way: the LZ4 round driver took the
position of a lane that could not start inside the staged window as the new read position.)"""
import os
import shutil
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_emulated_kernels_are_asan_clean(tmp_path):
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not os.path.isabs(asan) or shutil.which("g++") is None:
        pytest.skip("no AddressSanitizer runtime")
    lib = str(tmp_path / "libswc_emu_asan.so")
    subprocess.run(["g++", "-O1", "-g", "-fsanitize=address", "-fno-omit-frame-pointer", "-std=c++17", "-DSWC_HOST_EMULATION",
                    "-fPIC", "-shared", "-Wno-unknown-pragmas", "-pthread", "-o", lib, os.path.join(HERE, "host_emu", "emu.cpp")], check=True)
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0")
    p = subprocess.run([sys.executable, os.path.join(HERE, "_asan_child.py"), lib], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0 and "asan-clean" in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])
