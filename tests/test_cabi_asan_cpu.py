"""SURVEY.md 5 (sanitizers): the host side of every C-ABI entry point -- argument validation, workspace arithmetic, launch
set-up -- under AddressSanitizer.  `build.py --asan` compiles the library with the host code instrumented (device code as
usual); the C-ABI tests of tests/test_cabi_cpu.py then run in a child process with the sanitizer runtime preloaded.  No GPU
is needed: those tests never launch a kernel.  GPU AddressSanitizer is not available on this pool (xnack)."""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_module():
    spec = importlib.util.spec_from_file_location('vpn_build', os.path.join(ROOT, 'volumetric-primitives-net_amd', 'build.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_cabi_host_code_under_address_sanitizer():
    b = _build_module()
    try:
        b.hipcc()
        runtime = b.asan_runtime()
    except RuntimeError as e:
        pytest.skip(str(e))
    lib = b.build(verbose=False, asan=True)
    env = dict(os.environ, VPN_HIP_LIB=lib, LD_PRELOAD=runtime, PYTHONPATH=ROOT,
               ASAN_OPTIONS='detect_leaks=0:abort_on_error=0:exitcode=97')       # the interpreter's own leaks are not ours
    out = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_cabi_cpu.py'), os.path.join(ROOT, 'tests', 'test_emd.py'),
                          '-x', '-q', '-m', 'not gpu', '-p', 'no:cacheprovider'], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = (out.stdout + out.stderr)[-3000:]
    assert 'AddressSanitizer' not in out.stdout + out.stderr, tail
    assert out.returncode == 0, tail
    assert ' passed' in out.stdout, tail
