"""Build libvpn_hip.so in-tree with hipcc for gfx950 (no torch extension machinery, no hipify).

    python volumetric-primitives-net_amd/build.py [--force]

One translation unit per .hip file so each can carry its own floating-point flags
(chamfer.hip must keep correctly rounded sqrt and no FMA contraction)."""
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libvpn_hip.so')
SOURCES = ['vpn_api.hip', 'sampler.hip', 'chamfer.hip', 'raster.hip', 'emd.hip', 'head.hip', 'mesh.hip', 'trainstep.hip']
COMMON = ['-O3', '--offload-arch=gfx950', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function']
COMMON += os.environ.get('VPN_EXTRA_FLAGS', '').split()      # experiments only (e.g. -DVPN_CHAMFER_DEBUG)
RASTER_EXTRA = os.environ.get('VPN_RASTER_FLAGS', '').split()  # experiments only: extra flags for raster.hip alone
PER_FILE = {
    # index-exact argmin: correctly rounded sqrt (hipcc default) and no contraction (also a pragma in the file)
    # -amdgpu-mfma-vgpr-form: MFMA results straight into VGPRs (no v_accvgpr_read copies before the min-tree)
    'chamfer.hip': ['-ffp-contract=off', '-mllvm', '-amdgpu-mfma-vgpr-form'],
    # bit-equal to the oracle's auction: same rounding rules as the Chamfer scan
    'emd.hip': ['-ffp-contract=off', '-fno-slp-vectorize'],     # packed fp32 is half rate: keep the scan scalar
    'trainstep.hip': ['-ffp-contract=off'],                     # the VP-diversity neighbours follow the Chamfer arithmetic
    # the raster is compared with a 1e-4 tolerance: 1-ulp v_rcp/v_sqrt instead of the IEEE sequences
    # -fgpu-flush-denormals-to-zero: no denormal-safe scaling around v_rcp / v_sqrt / v_exp
    # -fno-slp-vectorize: packed fp32 is half rate here and the packing costs v_mov shuffles and 25 VGPRs
    'raster.hip': ['-fno-hip-fp32-correctly-rounded-divide-sqrt', '-fgpu-flush-denormals-to-zero', '-fno-slp-vectorize'] + RASTER_EXTRA,
}


def hipcc():
    for c in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if c and os.path.exists(c):
            return c
    raise RuntimeError('hipcc not found: libvpn_hip.so cannot be built')


def digest():
    h = hashlib.sha256()
    for root in (CSRC, os.path.join(HERE, '..', 'include')):
        for f in sorted(os.listdir(root)):
            if f.endswith(('.hip', '.h')):
                h.update(open(os.path.join(root, f), 'rb').read())
    h.update(repr((COMMON, PER_FILE)).encode())
    return h.hexdigest()


ASAN_LIB = os.path.join(HERE, 'libvpn_hip_asan.so')
# host-side AddressSanitizer build: the launch / validation code of every entry point instrumented, the
# device code compiled as usual (-fno-gpu-sanitize: GPU ASan needs xnack+, which this pool does not offer).  Loaded by
# tests/test_cabi_asan_cpu.py in a child process with the sanitizer runtime preloaded.
ASAN_FLAGS = ['-g', '-fsanitize=address', '-fno-gpu-sanitize']


def asan_runtime():
    base = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc()))), 'lib', 'llvm', 'lib', 'clang')
    for root, _dirs, files in os.walk(base):
        if 'libclang_rt.asan-x86_64.so' in files:
            return os.path.join(root, 'libclang_rt.asan-x86_64.so')
    raise RuntimeError('libclang_rt.asan-x86_64.so not found under ' + base)


def build(force=False, verbose=True, asan=False):
    lib = ASAN_LIB if asan else LIB
    stamp = lib + '.stamp'
    d = digest() + ('asan' if asan else '')
    if not force and os.path.exists(lib) and os.path.exists(stamp) and open(stamp).read() == d:
        return lib
    cc = hipcc()
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(CSRC, s.replace('.hip', '.asan.o' if asan else '.o'))
        cmd = [cc] + COMMON + (ASAN_FLAGS if asan else []) + PER_FILE.get(s, []) + ['-c', os.path.join(CSRC, s), '-o', o]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd)))
        objs.append(o)
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed on ' + s)
    cmd = [cc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + (['-fsanitize=address', '-fno-gpu-sanitize', '-shared-libasan'] if asan else []) + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    open(stamp, 'w').write(d)
    return lib


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, asan='--asan' in sys.argv))
