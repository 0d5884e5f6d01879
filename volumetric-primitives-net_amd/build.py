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


def build(force=False, verbose=True):
    stamp = LIB + '.stamp'
    d = digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == d:
        return LIB
    cc = hipcc()
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(CSRC, s.replace('.hip', '.o'))
        cmd = [cc] + COMMON + PER_FILE.get(s, []) + ['-c', os.path.join(CSRC, s), '-o', o]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd)))
        objs.append(o)
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed on ' + s)
    cmd = [cc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    open(stamp, 'w').write(d)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
