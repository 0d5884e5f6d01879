"""Static VALU issue-class mix of the raster kernels, from the device assembly of raster.hip:

    python tools/isa_mix.py <tag>      ->  profiles/<tag>_isa_mix.json (+ .txt)

gfx950 has no SQ_INSTS_VALU_TRANS-style counter, so the executed-work roofline of the raster prices the PMC count
SQ_INSTS_VALU with the class mix of the kernel's ISA and the issue costs measured by tools/ubench/valu_rates2.hip
(plain 2.5 cycles per wave-instruction per SIMD, slow class 4.3, transcendental 8.3).  The mix is
static (all instructions of the kernel, the loop bodies dominate); bench.py reads the JSON."""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'volumetric-primitives-net_amd'))
import build as vbuild  # noqa: E402

COST = {'plain': 2.5, 'slow': 4.3, 'trans': 8.3}      # cycles per wave-instruction per SIMD at 2.4 GHz (measured)


def classify(ins):
    op = ins.split()[0]
    if not op.startswith('v_'):
        return None
    base = re.sub(r'_(e32|e64|dpp|sdwa)$', '', op)
    if re.match(r'v_(exp|rcp|sqrt|rsq|log|sin|cos)_', base):
        return 'trans'
    if re.match(r'v_(min|max|med3|cmp|cmpx|cndmask|minimum|maximum|readlane|readfirstlane|writelane)', base):
        return 'slow'
    rest = ins[len(op):]
    if 'dpp' in ins or 'row_' in rest or 'quad_perm' in rest:
        return 'slow'
    if re.search(r'(?<![a-z_])s\d+|s\[\d+:\d+\]|vcc|exec', rest):     # an SGPR source operand makes it slow-class
        return 'slow'
    return 'plain'


def main(tag):
    src = os.path.join(vbuild.CSRC, 'raster.hip')
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, 'raster.s')
        subprocess.check_call([vbuild.hipcc()] + vbuild.COMMON + vbuild.PER_FILE['raster.hip'] +
                              ['--cuda-device-only', '-S', '-o', asm, src], stderr=subprocess.DEVNULL)
        text = open(asm).read().split('\n')
    funcs, cur = {}, None
    for line in text:
        m = re.match(r'^(_Z\w+):', line)
        if m:
            cur = m.group(1)
            funcs[cur] = []
        elif line.startswith('.Lfunc_end'):
            cur = None
        elif cur and line.startswith('\t') and not line.startswith(('\t.', '\t;')):
            funcs[cur].append(line.strip())
    # raster_total_kernel<true> is the training step's instantiation (tile entries) = the name the launch profile uses;
    # <false> (module path, C2) is listed beside it (the two mixes differ in the third digit)
    names = {'raster_total_kernelILb1': 'raster_total_kernel', 'raster_total_kernelILb0': 'raster_total_kernel<common>',
             'raster_fwd_kernelILi0': 'raster_fwd_kernel<0>',
             'raster_fwd_kernelILi1': 'raster_fwd_kernel<1>', 'raster_bwd_kernelILi0': 'raster_bwd_kernel<0>',
             'raster_bwd_kernelILi1': 'raster_bwd_kernel<1>'}
    out, lines = {}, ['# static VALU issue-class mix of raster.hip (tools/isa_mix.py); cycles per class: %s' % COST]
    for mangled, ins in funcs.items():
        key = next((v for k, v in names.items() if k in mangled), None)
        if key is None:
            continue
        c = collections.Counter(filter(None, (classify(i) for i in ins)))
        valu = sum(c.values())
        mix = {k: c[k] / valu for k in COST}
        out[key] = {'valu_static': valu, 'mix': mix, 'cycles_per_valu': sum(mix[k] * COST[k] for k in COST)}
        lines.append('%-24s VALU %5d  plain %.3f  slow %.3f  trans %.3f  -> %.2f cycles per wave-instruction' %
                     (key, valu, mix['plain'], mix['slow'], mix['trans'], out[key]['cycles_per_valu']))
    out['_cost_cycles'] = COST
    json.dump(out, open(os.path.join(ROOT, 'profiles', tag + '_isa_mix.json'), 'w'), indent=1)
    open(os.path.join(ROOT, 'profiles', tag + '_isa_mix.txt'), 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'r02')
