"""ctypes binding of libvpn_hip.so (C ABI declared in include/vpn_hip.h).

There is no CPU or PyTorch fallback: if the library is missing or a call fails the
binding raises.  PyTorch is used only for device memory and the current HIP stream."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('VPN_HIP_LIB') or os.path.join(_HERE, 'libvpn_hip.so')     # VPN_HIP_LIB: the sanitizer build of the tests
ABI_VERSION = 5

_c_f = ctypes.c_void_p      # device pointers travel as void*
_i, _f, _u64, _sz = ctypes.c_int, ctypes.c_float, ctypes.c_uint64, ctypes.c_size_t

SIGNATURES = {
    'vpn_abi_version': (ctypes.c_int, []),
    'vpn_error_string': (ctypes.c_char_p, [_i]),
    'vpn_profile_enable': (_i, [_i]),
    'vpn_profile_read': (_i, [ctypes.c_char_p, _i, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int), _i]),
    'vpn_sample_fwd': (_i, [_c_f, _c_f, _c_f, _u64, _c_f, _u64, _i, _i, _i, _c_f, _c_f]),
    'vpn_sample_bwd': (_i, [_c_f, _c_f, _c_f, _u64, _c_f, _u64, _i, _i, _i, _c_f, _c_f, _c_f]),
    'vpn_sample_chamfer_bwd': (_i, [_c_f, _c_f, _c_f, _u64, _c_f, _u64, _i, _i, _i, _c_f, _c_f, _i, _c_f, _c_f, _c_f, _c_f, _c_f,
                                    _f, _f, _c_f, _c_f]),
    'vpn_transform_fwd': (_i, [_c_f, _c_f, _c_f, _i, _i, _c_f, _c_f]),
    'vpn_transform_bwd': (_i, [_c_f, _c_f, _c_f, _i, _i, _c_f, _c_f, _c_f, _c_f]),
    'vpn_chamfer_fwd': (_i, [_c_f, _c_f, _i, _i, _i, _c_f, _c_f, _c_f, _c_f, _c_f]),
    'vpn_chamfer_nn': (_i, [_c_f, _c_f, _i, _i, _i, _c_f, _c_f, _c_f]),
    'vpn_chamfer_workspace': (_sz, [_i, _i, _i]),
    'vpn_chamfer_fwd_ws': (_i, [_c_f, _c_f, _i, _i, _i, _c_f, _c_f, _c_f, _c_f, _c_f, _sz, _i, _c_f]),
    'vpn_chamfer_loss': (_i, [_c_f, _c_f, _i, _i, _i, _f, _f, _c_f, _c_f]),
    'vpn_chamfer_bwd': (_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _i, _i, _i, _f, _f, _c_f, _c_f, _c_f]),
    'vpn_raster_records_size': (_sz, [_i, _i, _i, _i]),
    'vpn_raster_fwd': (_i, [_c_f, _c_f, _c_f, _i, _i, _i, _i, _f, _f, _f, _c_f, _c_f, _c_f, _c_f, _c_f]),
    'vpn_raster_bwd_workspace': (_sz, [_i, _i, _i, _i]),
    'vpn_raster_bwd': (_i, [_c_f, _c_f, _c_f, _i, _i, _i, _i, _f, _f, _f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f]),
    'vpn_raster_loss_workspace': (_sz, [_i, _i, _i]),
    'vpn_raster_loss_fwd': (_i, [_c_f, _c_f, _c_f, _i, _i, _i, _i, _f, _f, _f, _c_f, _c_f, _i, _c_f, _c_f, _c_f, _c_f, _c_f]),
    'vpn_raster_total_fwd': (_i, [_c_f, _c_f, _c_f, _i, _i, _i, _i, _f, _f, _f, _c_f, _c_f, _i, _f, _f, _c_f, _c_f, _c_f, _i, _c_f]),
    'vpn_raster_total_fwd_fin': (_i, [_c_f, _c_f, _c_f, _i, _i, _i, _i, _f, _f, _f, _c_f, _c_f, _i, _f, _f, _c_f, _c_f, _c_f, _i,
                                      _c_f, _sz, _i, _i, _f, _f, _f, _c_f, _c_f, _c_f, _c_f, _c_f]),
    'vpn_hotpath_sample_fwd': (_i, [_c_f, _c_f, _c_f, _u64, _c_f, _u64, _i, _i, _i, _c_f, _c_f, _i, _i, _f, _c_f, _c_f,
                                    _c_f, _i, _c_f, _sz, _c_f]),
    'vpn_hotpath_fused_features': (_i, [_i, _i, _i, _i]),
    'vpn_raster_order_size': (_sz, [_i, _i, _i]),
    'vpn_hotpath_chamfer_fwd': (_i, [_c_f, _c_f, _i, _i, _i, _c_f, _c_f, _c_f, _c_f, _c_f, _sz, _i, _c_f, _i, _i, _i, _c_f, _c_f]),
    'vpn_loss_finalize': (_i, [_c_f, _i, _i, _i, _c_f, _c_f, _i, _i, _f, _f, _f, _f, _f, _c_f, _c_f, _c_f]),
    'vpn_raster_total_bwd': (_i, [_c_f, _c_f, _i, _i, _i, _i, _c_f, _c_f, _c_f, _c_f, _i, _c_f]),
    'vpn_hotpath_bwd': (_i, [_c_f, _c_f, _c_f, _u64, _c_f, _u64, _i, _i, _i, _c_f, _c_f, _i, _c_f, _c_f, _c_f, _c_f, _c_f,
                             _f, _f, _c_f, _i, _i, _c_f, _c_f, _c_f, _c_f, _c_f]),
    'vpn_raster_loss_bwd': (_i, [_c_f, _c_f, _c_f, _i, _i, _i, _i, _f, _f, _f, _c_f, _c_f, _c_f, _c_f, _i, _c_f, _c_f, _c_f,
                                 _i, _c_f]),
    'vpn_camera_transform_fwd': (_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _i, _i, _i, _c_f, _c_f]),
    'vpn_camera_transform_bwd': (_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _i, _i, _i, _c_f, _c_f]),
    'vpn_mesh_fwd': (_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _i, _i, _i, _c_f, _c_f]),
    'vpn_mesh_bwd': (_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _i, _i, _i, _c_f, _c_f, _c_f]),
    'vpn_mesh_raster_workspace': (_sz, [_i, _i]),
    'vpn_mesh_raster_fwd': (_i, [_c_f, _c_f, _c_f, _i, _i, _i, _i, _i, _f, _c_f, _c_f, _c_f]),
    'vpn_mesh_raster_bwd': (_i, [_c_f, _c_f, _c_f, _i, _i, _i, _i, _i, _f, _c_f, _c_f, _c_f, _c_f, _c_f]),
    'vpn_mesh_sample_fwd': (_i, [_c_f, _c_f, _c_f, _u64, _u64, _i, _i, _i, _i, _c_f, _c_f, _c_f, _c_f, _c_f]),
    'vpn_mesh_sample_bwd': (_i, [_c_f, _c_f, _c_f, _c_f, _i, _i, _i, _i, _c_f, _c_f]),
    'vpn_head_pack_fwd': (_i, [_c_f, _c_f, _c_f, _i, _i, _i, _f, _f, _f, _f, _f, _c_f, _c_f]),
    'vpn_head_pack_bwd': (_i, [_c_f, _c_f, _c_f, _c_f, _i, _i, _i, _f, _f, _f, _f, _f, _c_f, _c_f, _c_f, _c_f]),
    'vpn_vpdiv_workspace': (_sz, [_i, _i]),
    'vpn_vpdiv_fwd': (_i, [_c_f, _c_f, _i, _i, _i, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f]),
    'vpn_trainstep_workspace': (_sz, [_i]),
    'vpn_camera_matrix': (_i, [_c_f, _c_f, _c_f, _c_f, _i, _i, _c_f, _c_f]),
    'vpn_trainstep_finalize': (_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _i, _i, _i, _i, _i, _f, _f, _f, _f, _f, _f, _f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f]),
    'vpn_trainstep_bwd': (_i, [_c_f, _c_f, _u64, _c_f, _u64, _i, _i, _i, _c_f, _c_f, _i, _c_f, _c_f, _c_f, _c_f, _f, _f, _c_f, _i, _i,
                               _c_f, _c_f, _c_f, _c_f, _c_f, _f, _c_f, _c_f, _c_f, _c_f, _f, _f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f,
                               _f, _f, _i, _c_f, _c_f]),
    'vpn_emd_workspace': (_sz, [_i, _i]),
    'vpn_emd_fwd': (_i, [_c_f, _c_f, _i, _i, _f, _i, _c_f, _c_f, _c_f, _i, _c_f]),
    'vpn_emd_bwd': (_i, [_c_f, _c_f, _c_f, _c_f, _i, _i, _c_f, _c_f]),
}

_lib = None


def lib():
    """Load libvpn_hip.so once.  Raises if it has not been built (no fallback path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('libvpn_hip.so is missing: build it with '
                               '`python volumetric-primitives-net_amd/build.py` (needs hipcc, gfx950)')
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)        # AttributeError if the .so does not export the symbol
            fn.restype, fn.argtypes = res, args
        if L.vpn_abi_version() != ABI_VERSION:
            raise RuntimeError('libvpn_hip.so ABI %d != binding ABI %d: rebuild' % (L.vpn_abi_version(), ABI_VERSION))
        _lib = L
    return _lib


class KernelTimer:
    """Brackets every C-ABI call with a pair of HIP events on the stream the kernels are
    launched on (torch's current stream) so bench.py can report per-entry-point device time.
    Off by default; the timed throughput region of bench.py runs without it."""

    def __init__(self):
        self.records = []

    def __enter__(self):
        global _timer
        _timer = self
        return self

    def __exit__(self, *exc):
        global _timer
        _timer = None

    def summary(self):
        """name -> (calls, mean milliseconds).  Synchronises."""
        torch.cuda.synchronize()
        out = {}
        for name, a, b in self.records:
            n, t = out.get(name, (0, 0.0))
            out[name] = (n + 1, t + a.elapsed_time(b))
        return {k: (n, t / n) for k, (n, t) in out.items()}


_timer = None


class KernelProfile:
    """Per-KERNEL device times measured by the library itself (HIP events around every launch on the
    launch stream): `with KernelProfile() as kp: ...; kp.summary()` -> {kernel: (calls, mean ms)}."""

    def __enter__(self):
        lib().vpn_profile_enable(1)
        return self

    def __exit__(self, *exc):
        self._summary = self._read()
        lib().vpn_profile_enable(0)

    def _read(self):
        names = ctypes.create_string_buffer(8192)
        ms = (ctypes.c_float * 64)()
        calls = (ctypes.c_int * 64)()
        n = lib().vpn_profile_read(names, 8192, ms, calls, 64)
        if n < 0:
            check(n)
        keys = names.value.decode().split('\n')[:n]
        return {k: (calls[i], ms[i]) for i, k in enumerate(keys)}

    def summary(self):
        return getattr(self, '_summary', None) or self._read()


def call(name, *args):
    """Invoke entry point `name`, raise on a non-zero return code."""
    fn = getattr(lib(), name)
    if _timer is None:
        check(fn(*args))
        return
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    rc = fn(*args)
    b.record()
    _timer.records.append((name, a, b))
    check(rc)


def check(rc):
    if rc != 0:
        raise RuntimeError('libvpn_hip: %s (code %d)' % (lib().vpn_error_string(rc).decode(), rc))


def ptr(t):
    """Device pointer of a contiguous CUDA(HIP) tensor, or NULL for None."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('vpn_amd operators run on the GPU only (got a %s tensor); there is no CPU path'
                           % t.device.type)
    if not t.is_contiguous():
        raise RuntimeError('vpn_amd: tensor must be contiguous')
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
