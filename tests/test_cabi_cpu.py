"""CPU-only checks of the drop-in boundary: the C-ABI library builds, loads and exports every
symbol include/vpn_hip.h declares, argument validation works without a GPU, and the host-side
mirror keeps the reference's names and signatures.  No compute call is made here."""
import ctypes
import inspect
import os
import re

import pytest
import torch

from conftest import ROOT


def _header_symbols():
    src = open(os.path.join(ROOT, 'include', 'vpn_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(vpn_[a-z0-9_]+)\s*\(', src)))


def test_library_builds_and_exports_header_symbols():
    import importlib.util
    spec = importlib.util.spec_from_file_location('vpn_build', os.path.join(ROOT, 'volumetric-primitives-net_amd', 'build.py'))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    path = b.build(verbose=False)
    L = ctypes.CDLL(path)
    syms = _header_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(L, s), 'libvpn_hip.so does not export ' + s
    import vpn_amd._lib as lib
    assert sorted(lib.SIGNATURES) == syms, 'ctypes binding and header disagree'
    assert lib.lib().vpn_abi_version() == lib.ABI_VERSION


def test_argument_validation_needs_no_gpu():
    import vpn_amd._lib as lib
    L = lib.lib()
    assert L.vpn_chamfer_fwd(None, None, 1, 1, 1, None, None, None, None, None) == -1
    assert L.vpn_sample_fwd(None, None, None, 0, None, 0, 1, 1, 1, None, None) == -1
    assert L.vpn_raster_fwd(None, None, None, 1, 1, 8, 8, 0.1, 0.1, 2.0, None, None, None, None, None) == -1
    assert L.vpn_raster_loss_fwd(None, None, None, 1, 1, 8, 8, 0.1, 0.1, 2.0, None, None, 0, None, None, None, None, None) == -1
    assert L.vpn_raster_total_fwd(None, None, None, 1, 1, 8, 8, 0.1, 0.1, 2.0, None, None, 0, 1.0, 1.0, None, None, None, 0, None) == -1
    assert L.vpn_loss_finalize(None, 1, 8, 8, None, None, 0, 0, 1.0, 1.0, 1.0, 1.0, 1.0, None, None, None) == -1
    assert L.vpn_raster_total_bwd(None, None, 1, 1, 8, 8, None, None, None, None, 0, None) == -1
    assert L.vpn_camera_transform_fwd(None, None, None, None, None, 1, 8, 1, None, None) == -1
    assert L.vpn_raster_records_size(2, 3, 32, 32) == 2 * 3 * 16 * 16 + 2 * 4 * 8         # records + one mask word per tile
    assert L.vpn_raster_records_size(1, 65, 16, 16) == 65 * 16 * 16 + 2 * 8
    assert L.vpn_raster_bwd_workspace(2, 3, 32, 32) == 2 * 4 * 3 * 12 * 4
    assert L.vpn_raster_bwd_workspace(0, 3, 32, 32) == 0
    assert L.vpn_raster_loss_workspace(2, 32, 32) == 16 + 2 * 16 + 2 * 4 * 2 * 4
    # round-3 entry points: the fused raster + finalisation and the triangle-mesh path
    assert L.vpn_raster_total_fwd_fin(None, None, None, 1, 1, 8, 8, 0.1, 0.1, 2.0, None, None, 0, 1.0, 1.0, None, None, None, 0,
                                      None, 0, 0, 0, 1.0, 1.0, 1.0, None, None, None, None, None) == -1
    assert L.vpn_mesh_raster_fwd(None, None, None, 1, 3, 1, 8, 8, 1e-4, None, None, None) == -1
    assert L.vpn_mesh_raster_bwd(None, None, None, 1, 3, 1, 8, 8, 1e-4, None, None, None, None, None) == -1
    assert L.vpn_mesh_sample_fwd(None, None, None, 0, 0, 1, 3, 1, 8, None, None, None, None, None) == -1
    assert L.vpn_mesh_sample_bwd(None, None, None, None, 1, 3, 1, 8, None, None) == -1
    assert L.vpn_mesh_raster_workspace(2, 10) == 2 * 10 * (16 + 8) and L.vpn_mesh_raster_workspace(0, 10) == 0
    assert L.vpn_emd_workspace(2, 100) == 2 * 100 * 10 * 4 + 16
    assert b'null pointer' in L.vpn_error_string(-1)
    with pytest.raises(RuntimeError):
        lib.check(-2)


def test_no_cpu_fallback():
    """The product path must fail loudly on CPU tensors (no oracle / eager fallback)."""
    import vpn_amd
    p = torch.rand(1, 4, 3)
    with pytest.raises(RuntimeError, match='GPU only'):
        vpn_amd.ChamferDistanceLoss()(p, p)
    with pytest.raises(RuntimeError, match='GPU only'):
        vpn_amd.Sampling.sphere_sampling(torch.rand(1, 3), torch.rand(1, 4), torch.rand(1, 3), 8)
    src = ''
    pkg = os.path.join(ROOT, 'volumetric-primitives-net_amd')
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src += open(os.path.join(root, f)).read()
    assert not re.search(r'^\s*(from|import)\s+\S*oracle', src, flags=re.M), 'the package must not import the oracle'
    assert 'vpn_oracle.' not in re.sub(r'#.*', '', re.sub(r'"""(.|\n)*?"""', '', src))


def test_reference_surface_is_mirrored():
    """Names / argument order of the reference call sites."""
    import vpn_amd
    from vpn_amd.modules import sampling, loss, render, transform, meshing
    sig = inspect.signature
    # kaolin's TriangleMesh as train_sphere.py:50-78 uses it
    for name in ('from_obj', 'cuda', 'sample', 'vertices', 'faces'):
        assert hasattr(meshing.TriangleMesh(torch.zeros(3, 3), torch.zeros(1, 3, dtype=torch.int64)), name) or hasattr(meshing.TriangleMesh, name)
    assert list(sig(meshing.TriangleMesh.sample).parameters)[:2] == ['self', 'num_samples']
    assert list(sig(sampling.Sampling.sphere_sampling).parameters)[:4] == ['v', 'q', 't', 'num_points']
    assert list(sig(sampling.Sampling.cuboid_sampling).parameters)[:4] == ['v', 'q', 't', 'num_points']
    assert sig(sampling.Sampling.sphere_sampling).parameters['num_points'].default == 1000
    assert sampling.Sampling.cone_sampling(None, None, None) is None
    assert list(sig(loss.ChamferDistanceLoss.forward).parameters) == ['self', 'points1', 'points2', 'each_batch', 'w1', 'w2']
    assert list(sig(loss.VPDiverseLoss.forward).parameters) == ['self', 'translates', 'gt_points']
    assert list(sig(loss.SilhouetteLoss.forward).parameters) == ['self', 'predict_meshes', 'gt_silhouettes', 'dists', 'elevs', 'azims']
    assert list(sig(render.VertexRenderer.render).parameters)[:5] == ['mesh', 'dist', 'elev', 'azim', 'colors']
    assert list(sig(transform.transform_points).parameters) == ['points', 'q', 't']
    assert list(sig(transform.view_to_obj_points).parameters) == ['points', 'dists', 'elevs', 'azims', 'angles']
    # shape assertions like the reference
    with pytest.raises(AssertionError):
        vpn_amd.Sampling.check_parameters(torch.rand(2, 3), torch.rand(2, 3), torch.rand(2, 3))
    with pytest.raises(AssertionError):
        vpn_amd.ChamferDistanceLoss.check_parameters(torch.rand(2, 3))
    with pytest.raises(ValueError):
        vpn_amd.kinds_from_counts(0, 1, cone_num=1)
    with pytest.raises(ValueError):
        vpn_amd.kinds_tensor([0, 2], torch.device('cpu'))
    assert vpn_amd.kinds_from_counts(1, 2) == [1, 0, 0]          # cuboids first (train.py:112-116)
    v = [torch.rand(2, 3) for _ in range(3)]
    q = [torch.rand(2, 4) for _ in range(3)]
    t = [torch.rand(2, 3) for _ in range(3)]
    p = vpn_amd.pack_primitives(v, q, t)
    assert p.shape == (2, 3, 10) and torch.equal(p[:, 1, 3:7], q[1])
