"""`python bench.py --gpus N` (N>1, no launcher in the environment) must start its N ranks itself, as CHILD processes,
before anything in the parent has touched the GPU or loaded the HIP library."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROBE = r'''
import json, os, subprocess, sys
sys.argv = ['bench.py', '--gpus', '4', '--steps', '5', '--warmup', '2', '--rehearse']
os.environ.pop('WORLD_SIZE', None)
seen = {}
def fake_run(cmd, env=None, **kw):
    import torch
    seen['cmd'] = cmd
    seen['vpn_loaded'] = any(m == 'vpn_amd' or m.startswith('vpn_amd.') for m in sys.modules)
    seen['cuda_initialised'] = torch.cuda.is_initialized()
    seen['ipc'] = (env or {}).get('HSA_ENABLE_IPC_MODE_LEGACY')
    class R: returncode = 7; stdout = 'noise\n{"metric": "x"}\n'
    return R()
subprocess.run = fake_run
import bench
try:
    bench.main()
except SystemExit as e:
    seen['rc'] = e.code
print('PROBE ' + json.dumps(seen))
'''


def test_gpus_n_spawns_children_before_touching_the_gpu():
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop('WORLD_SIZE', None)
    out = subprocess.run([sys.executable, '-c', PROBE], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    line = [l for l in out.stdout.splitlines() if l.startswith('PROBE ')][-1]
    seen = json.loads(line[6:])
    cmd = seen['cmd']
    assert cmd[0] == sys.executable and cmd[1:3] == ['-m', 'torch.distributed.run']
    assert cmd[cmd.index('--nproc-per-node') + 1] == '4'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert os.path.basename(cmd[cmd.index('--master-port') + 2]) == 'bench.py'
    assert cmd[-6:] == ['--gpus', '4', '--steps', '5', '--warmup', '2'] or cmd[-7:] == ['--gpus', '4', '--steps', '5', '--warmup', '2', '--rehearse']
    assert seen['vpn_loaded'] is False and seen['cuda_initialised'] is False
    assert seen['ipc'] == '0'
    assert seen['rc'] == 7            # the children's return code is the parent's
    # only rank 0's JSON line reaches stdout
    assert [l for l in out.stdout.splitlines() if not l.startswith('PROBE ')] == ['{"metric": "x"}']


def test_under_a_launcher_no_respawn():
    """With WORLD_SIZE in the environment (torch.distributed.run started us) the launcher branch is not taken."""
    src = open(os.path.join(ROOT, 'bench.py')).read()
    assert "'WORLD_SIZE' not in os.environ and args.gpus > 1" in src
    assert 'os.exec' not in src


def test_multi_rank_line_is_self_checking():
    """the N>1 line carries `dist` = what the process group itself reports (backend, world size,
    RCCL version, every rank's device) and hipEvent-timed compute / collective microseconds.  The code path needs a GPU to
    run (tests/test_dist_gpu.py rehearses it); here: the record is built from the group's own queries, not from arguments."""
    src = open(os.path.join(ROOT, 'bench.py')).read()
    blk = src[src.index('dist_info = {'):src.index('dist_info = {') + 600]
    for key in ("'backend': dist.get_backend()", "'world_size': dist.get_world_size()", "'rccl_version'", "'device_names'",
                "'compute_us'", "'collective_us'"):
        assert key in blk, key
    assert "out['dist'] = dist_info" in src
    assert 'dist.all_gather_object(names' in src and 'torch.cuda.nccl.version()' in src
