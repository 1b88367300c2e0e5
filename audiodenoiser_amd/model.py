"""Drop-in for the reference's ``model.py`` whose eval forward runs on hand-written HIP kernels (gfx950).

Mirrors the public surface of ``/root/reference/code/model.py``: ``UNet(in_channels=1, num_classes=1)`` is an
``nn.Module`` with the reference's exact parameter tree (136 ``state_dict`` entries, strict-loadable from a
reference checkpoint, ``test.py:65``), and ``forward(x: (N,C,F,T) float32) -> (N,K,F,T) float32``
(``model.py:70-94``).  The arithmetic does NOT go through ATen: ``forward`` hands raw device pointers and the
current PyTorch-ROCm stream to ``libadn.so`` (``include/adn.h``).

Scope: the inference forward — eval mode, no autograd.  The reference's own caller keeps the model
and its input on the CPU (``test.py:63-66,100,112-113``); such a call is served by the same HIP path: the input is
staged onto the current ROCm device, the packed weights are uploaded from wherever the parameters live, and the
result comes back on the input's device.  Train-mode forward (batch-statistics BatchNorm + backward, reference
``train.py:62-71``) raises, and so does any call on a machine without a ROCm device; nothing falls back to CPU
arithmetic.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .weights import DOUBLE_CONVS, UP_CONVTS

__all__ = ["UNet", "DoubleConvLayer", "DownSampleLayer", "UpSampleLayer"]


class _ParamsOnly(nn.Module):
    """The sub-blocks only hold parameters here; the fused forward lives in :meth:`UNet.forward`."""

    def forward(self, *args, **kwargs):  # pragma: no cover - guard
        raise NotImplementedError(
            f"{type(self).__name__} is a parameter container in audiodenoiser_amd; call UNet.forward "
            "(the whole network runs as one fused HIP launch sequence)")


class DoubleConvLayer(_ParamsOnly):
    """Parameters of conv3x3-BN-ReLU-conv3x3-BN-ReLU (reference model.py:7-17): ``double_conv.{0,1,3,4}``."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        seq = nn.Sequential()
        widths = (in_channels, out_channels)
        for slot, cin in zip((0, 3), widths):
            seq.add_module(str(slot), nn.Conv2d(cin, out_channels, kernel_size=3, padding=1))
            seq.add_module(str(slot + 1), nn.BatchNorm2d(out_channels))
            seq.add_module(str(slot + 2), nn.ReLU(inplace=True))
        self.double_conv = seq


class DownSampleLayer(_ParamsOnly):
    """Parameters of reference model.py:23-28 (``conv``; the max-pool has no state)."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.pool = nn.MaxPool2d(2)
        self.conv = DoubleConvLayer(in_channels, out_channels)


class UpSampleLayer(_ParamsOnly):
    """Parameters of reference model.py:35-39 (``up`` = ConvTranspose2d k2 s2, ``conv`` on 2*C channels)."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.up = nn.ConvTranspose2d(in_channels, out_channels, kernel_size=2, stride=2)
        self.conv = DoubleConvLayer(in_channels, out_channels)


class UNet(nn.Module):
    def __init__(self, in_channels: int = 1, num_classes: int = 1):
        super().__init__()
        if not (1 <= in_channels <= 64 and 1 <= num_classes <= 64):
            raise ValueError("UNet(in_channels, num_classes): the MI355X path takes 1..64 input planes and 1..64 classes "
                             "(reference configuration: UNet(1, 1), test.py:63)")
        self.in_channels, self.num_classes = in_channels, num_classes
        for (name, cin, cout) in DOUBLE_CONVS:
            if name == "downconv1.conv":
                cin = in_channels
            top = name.split(".")[0]
            if top.startswith("downconv"):
                setattr(self, top, DownSampleLayer(cin, cout))
            elif top == "bottleneck":
                self.bottleneck = DoubleConvLayer(cin, cout)
        for (name, cin, cout) in UP_CONVTS:
            setattr(self, name.split(".")[0], UpSampleLayer(cin, cout))
        self.out = nn.Conv2d(in_channels=64, out_channels=num_classes, kernel_size=1)
        self._handle = None
        self._handle_key = None
        self._workspace = None
        self._compute_dtype = "f32"
        self._batch_invariant = None                  # None: the library's default (adn.h, adn_unet_set_batch_invariant)

    # ------------------------------------------------------------------ kernel choice
    def set_batch_invariant(self, on: bool = True):
        """Default (off): the library picks the kernel of each layer by the launch's grid (small grids: finer tiles, K loop cut over
        several workgroups) -- fastest at every batch size, but the same clip computed alone (e.g. the short last batch of a
        dataset, ``test.py``'s 5-clip set) and inside a large batch then differs in the last bits (fp32: <= 2e-5 of max|y|, both within
        1e-4 of the reference; fp16: <= 5e-3, both within 1e-2).  ``set_batch_invariant(True)``
        pins one kernel per layer by geometry alone: a clip's output is bit-identical whatever batch it is computed in -- use it for
        evaluation / regression runs that compare outputs across batch sizes (costs up to 2x at batch 1-4)."""
        self._batch_invariant = bool(on)
        if self._handle is not None:
            _lib.check(_lib.load().adn_unet_set_batch_invariant(self._handle, int(self._batch_invariant)), "adn_unet_set_batch_invariant")
        return self

    # ------------------------------------------------------------------ arithmetic type
    def set_compute_dtype(self, dtype: str):
        """"f32" (default): fp32 arithmetic, parity 1e-4 with the reference -- the 3x3 layers on the exact-fp32 matrix cores
        (Winograd), the four transposed convolutions on the bf16 matrix cores through a three-term split of both fp32 operands
        (six products, fp32 accumulation: fp32-level accuracy for every finite value; ``ADN_CONVT_SPLIT=0`` when the handle is
        created keeps them on the exact-fp32 MFMA; non-finite activations give non-finite results in both forms, NaN where the
        exact form may give inf).  "f16": fp16 storage + fp16 MFMA
        with fp32 accumulation inside the library (BASELINE configs[4]); inputs/outputs stay float32 tensors and the
        result is within 1e-2 of the fp32 path."""
        if dtype not in ("f32", "f16"):
            raise ValueError("compute dtype must be 'f32' or 'f16'")
        if dtype != self._compute_dtype:
            self._compute_dtype = dtype
            self._release()
        return self

    # ------------------------------------------------------------------ handle management
    def _float_tensors(self):
        return [v for k, v in self.state_dict(keep_vars=True).items() if not k.endswith("num_batches_tracked")]

    def _weights_key(self, device):
        ts = self._float_tensors()
        return (device.index, self._compute_dtype, tuple(t._version for t in ts), tuple(t.data_ptr() for t in ts))

    def _release(self):
        if self._handle is not None:
            try:
                _lib.load().adn_unet_destroy(self._handle)
            finally:
                self._handle = None
                self._handle_key = None

    def __del__(self):
        try:
            self._release()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def refresh_weights(self):
        """Force re-packing (BatchNorm fold + re-layout + upload) at the next forward."""
        self._release()

    def _ensure_handle(self, device):
        key = self._weights_key(device)
        if self._handle is not None and key == self._handle_key:
            return self._handle
        self._release()
        L = _lib.load()
        host = [t.detach().to("cpu", torch.float32).contiguous() for t in self._float_tensors()]
        if len(host) != 118:
            raise _lib.AdnError(f"unexpected parameter tree: {len(host)} float tensors, expected 118")
        table = (_lib.c_float_p * len(host))(*[ctypes.cast(t.data_ptr(), _lib.c_float_p) for t in host])
        handle = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(L.adn_unet_create_general(ctypes.byref(handle), device.index, table, len(host),
                                                 1 if self._compute_dtype == "f16" else 0, self.in_channels, self.num_classes),
                       "adn_unet_create_general")
        self._handle = handle
        self._handle_key = key
        if self._batch_invariant is not None:
            _lib.check(L.adn_unet_set_batch_invariant(handle, int(self._batch_invariant)), "adn_unet_set_batch_invariant")
        return handle

    def _workspace_for(self, n, f, t, device):
        L = _lib.load()
        need = ctypes.c_size_t()
        _lib.check(L.adn_unet_workspace_bytes(self._handle, n, f, t, ctypes.byref(need)), "adn_unet_workspace_bytes")
        ws = self._workspace
        if ws is None or ws.device != device or ws.numel() < need.value:
            self._workspace = None   # drop the old one first
            ws = torch.empty(need.value, dtype=torch.uint8, device=device)
            self._workspace = ws
        return ws

    # ------------------------------------------------------------------ forward
    def _check_input(self, x):
        if self.training:
            raise RuntimeError("audiodenoiser_amd.UNet: train-mode forward (batch-statistics BatchNorm + autograd, "
                               "reference train.py:62-71) is outside the MI355X inference path; call .eval()")
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise RuntimeError("audiodenoiser_amd.UNet: the HIP forward records no autograd graph; wrap the call in "
                               "torch.no_grad() as the reference's test.py:112 / train.py:80 do")
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise ValueError(f"expected input (N, {self.in_channels}, F, T), got {tuple(x.shape)}")
        if x.shape[2] < 16 or x.shape[3] < 16:
            raise ValueError("F and T must be >= 16 (four 2x poolings)")
        if x.dtype != torch.float32:
            raise TypeError("expected float32 input")

    def forward(self, x: torch.Tensor, return_taps: bool = False):
        """``(N, C, F, T) float32 -> (N, K, F, T) float32`` (reference ``model.py:70-94``), eval-mode semantics.  Non-finite input
        values travel as through the reference's ``nn.ReLU`` / ``nn.MaxPool2d`` (NaN-propagating).  Results are within 1e-4 of
        the reference; they are bit-identical across batch sizes only after :meth:`set_batch_invariant`."""
        self._check_input(x)
        home = x.device
        if not x.is_cuda:
            # the reference's test.py call shape: CPU tensor into a model that was never moved.  Stage on the current
            # ROCm device (raises when there is none) -- still the HIP path, no CPU arithmetic.
            x = x.to(_lib.staging_device())
        x = x.contiguous()
        n, _, f, t = x.shape
        dev = x.device
        handle = self._ensure_handle(dev)
        ws = self._workspace_for(n, f, t, dev)
        y = torch.empty((n, self.num_classes, f, t), dtype=torch.float32, device=dev)
        L = _lib.load()
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            if not return_taps:
                _lib.check(L.adn_unet_forward(handle, x.data_ptr(), y.data_ptr(), n, f, t, ws.data_ptr(), ws.numel(),
                                              stream), "adn_unet_forward")
                return y if home == dev else y.to(home)
            names = ("down1", "down2", "down3", "down4", "bottleneck", "up1", "up2", "up3", "up4", "out")
            ch = (64, 128, 256, 512, 1024)
            hs, wsz = [f], [t]
            for _ in range(4):
                hs.append(hs[-1] // 2)
                wsz.append(wsz[-1] // 2)
            shapes = [(n, ch[l], hs[l], wsz[l]) for l in range(4)] + [(n, 1024, hs[4], wsz[4])]
            shapes += [(n, ch[l], hs[l], wsz[l]) for l in (3, 2, 1, 0)] + [(n, self.num_classes, f, t)]
            taps = [torch.empty(s, dtype=torch.float32, device=dev) for s in shapes]
            arr = (ctypes.c_void_p * 10)(*[tp.data_ptr() for tp in taps])
            _lib.check(L.adn_unet_forward_taps(handle, x.data_ptr(), y.data_ptr(), n, f, t, ws.data_ptr(), ws.numel(),
                                               arr, stream), "adn_unet_forward_taps")
            if home != dev:
                return y.to(home), {k: v.to(home) for k, v in zip(names, taps)}
            return y, dict(zip(names, taps))

    # ------------------------------------------------------------------ host-resident batches
    def forward_host_batches(self, batches, copy: bool = True):
        """Generator over an iterable of CPU ``(N, C, F, T) float32`` batches (the ``noisy`` half of the reference's evaluation loop
        over its ``DataLoader``, ``train.py:78-88``, whose batches are pinned, ``train.py:119``; pageable tensors are staged through
        pinned buffers): yields each batch's output as a CPU tensor, in order, with the two PCIe copies hidden under the
        neighbouring batches' forwards -- ``for out in model.forward_host_batches(noisy for noisy, _ in loader)`` runs at the rate of
        resident inputs (33.5 against 42.6 ms per 64 x 513x256 batch for ``model(noisy)``).

        Two pinned staging buffers and two device buffers per direction; the copies run on streams of their own, and batch i+1's copy
        in is submitted before batch i's copy out (the copy queue is served in submission order).  Batches may differ in N (a short
        last batch, none larger than the first); F, T and C are fixed by the first.  ``copy=False`` yields views of the two pinned
        output buffers, each valid until the generator is advanced again."""
        dev = _lib.staging_device()
        it = iter(batches)
        comp = torch.cuda.current_stream(dev)
        s_in, s_out = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        ev_in = [torch.cuda.Event() for _ in range(2)]
        ev_comp = [torch.cuda.Event() for _ in range(2)]
        ev_out = [torch.cuda.Event() for _ in range(2)]
        bufs = {"cap": 0}

        def grow(n, c, f, t):
            torch.cuda.synchronize(dev)                   # nothing in flight may still use the old buffers
            bufs.update(cap=n, shape=(c, f, t), xs=None,
                        xd=[torch.empty((n, c, f, t), dtype=torch.float32, device=dev) for _ in range(2)],
                        ys=[torch.empty((n, self.num_classes, f, t), dtype=torch.float32).pin_memory() for _ in range(2)])

        def copy_in(i, x):
            self._check_input(x)
            if x.is_cuda:
                raise ValueError("forward_host_batches takes CPU tensors; call the model on device tensors directly")
            n, c, f, t = x.shape
            if bufs["cap"] and (c, f, t) != bufs["shape"]:
                raise ValueError(f"every batch must be (N, {bufs['shape'][0]}, {bufs['shape'][1]}, {bufs['shape'][2]}), got {tuple(x.shape)}")
            if n > bufs["cap"]:
                grow(n, c, f, t)
            k = i & 1
            src = x
            if not x.is_pinned():
                if bufs["xs"] is None:
                    bufs["xs"] = [torch.empty((bufs["cap"],) + bufs["shape"], dtype=torch.float32).pin_memory() for _ in range(2)]
                ev_in[k].synchronize()                   # batch i-2's copy in has read this staging buffer
                src = bufs["xs"][k][:n]
                np.copyto(src.numpy(), x.detach().numpy())    # one plain memcpy (ATen's threaded copy is erratic on a shared host)
            with torch.cuda.stream(s_in):
                s_in.wait_event(ev_comp[k])               # batch i-2's forward has read this device buffer
                bufs["xd"][k][:n].copy_(src, non_blocking=True)
                ev_in[k].record(s_in)
            return n

        def result(j, n):
            ev_out[j & 1].synchronize()
            y = bufs["ys"][j & 1][:n]
            return torch.from_numpy(y.numpy().copy()) if copy else y

        nxt = next(it, None)
        if nxt is None:
            return
        sizes = {0: copy_in(0, nxt)}
        i = 0
        while True:
            k = i & 1
            comp.wait_event(ev_in[k])
            y = self.forward(bufs["xd"][k][:sizes[i]])
            ev_comp[k].record(comp)
            nxt = next(it, None)
            if nxt is not None:
                if nxt.dim() == 4 and nxt.shape[0] > bufs["cap"]:
                    raise ValueError("forward_host_batches: no batch may be larger than the first (it sizes the staging buffers)")
                sizes[i + 1] = copy_in(i + 1, nxt)
            with torch.cuda.stream(s_out):
                s_out.wait_event(ev_comp[k])
                bufs["ys"][k][:sizes[i]].copy_(y, non_blocking=True)
                ev_out[k].record(s_out)
                y.record_stream(s_out)
            del y
            if i >= 1:
                yield result(i - 1, sizes.pop(i - 1))
            if nxt is None:
                yield result(i, sizes.pop(i))
                return
            i += 1
