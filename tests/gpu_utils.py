import ctypes as C

import torch

from counterfactualworldmodels_amd import _lib


def dev():
    return torch.device("cuda:0")


def stream():
    return _lib.current_stream_handle(dev())


_DEFAULT = None  # None: the production library


def _lib_of(lib):
    return lib if lib is not None else (_DEFAULT or _lib.get_lib())


class Ops:
    """The stand-alone entry points below bound to ONE build of the library: `Ops(_lib.get_dev_lib())` for the tests that flip development
    switches (`cwm_debug_set` exists in libcwm_hip_dev.so only and acts on that shared object's thread-local options)."""

    def __init__(self, lib):
        self.lib = lib

    def set(self, key, value):
        _lib.check(self.lib.cwm_debug_set(key.encode() if isinstance(key, str) else key, int(value)), self.lib)

    def __getattr__(self, name):
        fn = globals()[name]
        return lambda *a, **k: fn(*a, lib=self.lib, **k)


def linear(a, w, bias=None, resid=None, gelu=False, mode="parity", lib=None):
    """through the C ABI: cwm_linear"""
    lib = _lib_of(lib)
    M, K = a.shape
    N = w.shape[0]
    a_d, w_d = a.to(dev()).contiguous(), w.to(dev()).contiguous()
    b_d = bias.to(dev()).contiguous() if bias is not None else None
    r_d = resid.to(dev()).contiguous() if resid is not None else None
    out = torch.empty(M, N, device=dev(), dtype=torch.float32)
    _lib.check(lib.cwm_linear(a_d.data_ptr(), w_d.data_ptr(), _lib.ptr(b_d), _lib.ptr(r_d), out.data_ptr(), M, N, K, int(gelu),
                              _lib.mode_id(mode), stream()), lib)
    return out.cpu()


def attention(qkv, H, mode="parity", lib=None):
    lib = _lib_of(lib)
    B, N, _ = qkv.shape
    q_d = qkv.to(dev()).contiguous()
    out = torch.empty(B, N, H * 64, device=dev(), dtype=torch.float32)
    _lib.check(lib.cwm_attention(q_d.data_ptr(), out.data_ptr(), B, N, H, _lib.mode_id(mode), stream()), lib)
    return out.cpu()


def layernorm(x, g, b, eps=1e-6, lib=None):
    lib = _lib_of(lib)
    rows, D = x.shape
    x_d, g_d, b_d = x.to(dev()).contiguous(), g.to(dev()), b.to(dev())
    out = torch.empty_like(x_d)
    _lib.check(lib.cwm_layernorm(x_d.data_ptr(), g_d.data_ptr(), b_d.data_ptr(), out.data_ptr(), rows, D, eps, stream()), lib)
    return out.cpu()


def mask_to_perm(mask, n_vis, lib=None):
    lib = _lib_of(lib)
    B, Nt = mask.shape
    m_d = mask.to(dev()).contiguous()
    perm = torch.empty(B, Nt, device=dev(), dtype=torch.int32)
    _lib.check(lib.cwm_mask_to_perm(m_d.data_ptr(), B, Nt, n_vis, perm.data_ptr(), stream()), lib)
    return perm.cpu()
