"""The host side of libdlpm_amd (host.cpp: schedules, MT19937 parity streams, LIM tables, error channel; png.cpp: the PNG
encoder) under AddressSanitizer + UndefinedBehaviorSanitizer, on the CPU build (GPU sanitizers are not available on the pool)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.skipif(shutil.which('g++') is None or not os.path.isdir('/opt/rocm/include'), reason='needs g++ and the ROCm headers')
def test_host_library_clean_under_asan_ubsan(tmp_path):
    src = os.path.join(ROOT, 'dlpm_amd', 'csrc')
    flags = ['-O1', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=undefined', '-fno-omit-frame-pointer',
             '-D__HIP_PLATFORM_AMD__', '-I/opt/rocm/include', '-I' + os.path.join(ROOT, 'include')]
    objs = []
    for f in ('host.cpp', 'png.cpp'):
        o = str(tmp_path / (f + '.o'))
        subprocess.check_call(['g++'] + flags + ['-c', os.path.join(src, f), '-o', o])
        objs.append(o)
    d = str(tmp_path / 'driver.o')
    subprocess.check_call(['gcc'] + flags + ['-c', os.path.join(ROOT, 'tests', 'native', 'host_sanitizer_driver.c'), '-o', d])
    exe = str(tmp_path / 'drv')
    subprocess.check_call(['g++', '-fsanitize=address,undefined'] + objs + [d, '-L/opt/rocm/lib', '-lamdhip64', '-lz', '-lpthread',
                                                                          '-Wl,-rpath,/opt/rocm/lib', '-o', exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith('ok'), (r.returncode, r.stdout[-500:], r.stderr[-3000:])
    assert 'runtime error' not in r.stderr and 'AddressSanitizer' not in r.stderr, r.stderr[-3000:]
