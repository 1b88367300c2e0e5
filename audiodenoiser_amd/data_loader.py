"""Drop-in for the reference's ``data_loader.py`` (``/root/reference/code/data_loader.py:1-72``).

``SpectrogramDataset(data_dir, target_size=(256, 64))`` keeps the reference contract: it pairs the sorted
``clean*.npy`` / ``noisy*.npy`` files of one folder, and ``__getitem__`` returns ``(noisy, clean)`` as
``(1, H, W)`` float32 tensors whose values went through float16 and were cropped / zero padded (bottom, right)
to ``target_size``.  That per-item host path is file I/O plus two numpy calls and stays on the host, as in the
reference (DataLoader worker processes cannot share a GPU context).

``load_batch_to_device`` is the MI355X ingest for the step right before the forward: raw fp32 spectrograms
are uploaded once and quantised + cropped/padded by one HIP kernel (``adn_quantize_pad``), producing the
``(B, 1, H, W)`` batch directly in HBM.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from . import _lib
from .wav import read_wav


def _list(data_dir: str, prefix: str, suffix: str = ".npy"):
    return sorted(os.path.join(data_dir, f) for f in os.listdir(data_dir)
                  if f.startswith(prefix) and f.endswith(suffix))


def fit_to(data: np.ndarray, target_size) -> np.ndarray:
    """Crop or zero-pad (bottom / right) a 2-D array to ``target_size`` (reference ``_pad_or_truncate``)."""
    th, tw = target_size
    out = np.zeros((th, tw), dtype=data.dtype)
    h, w = min(th, data.shape[0]), min(tw, data.shape[1])
    out[:h, :w] = data[:h, :w]
    return out


class SpectrogramDataset(Dataset):
    def __init__(self, data_dir, target_size=(256, 64)):
        self.target_size = tuple(target_size)
        clean = _list(data_dir, "clean")
        noisy = _list(data_dir, "noisy")
        print(f"Found {len(clean)} clean files and {len(noisy)} noisy files in {data_dir}")
        assert len(clean) == len(noisy), f"Mismatch in {data_dir}"
        self.pairs = list(zip(noisy, clean))
        print(f"Total pairs loaded: {len(self.pairs)}")

    def __len__(self):
        return len(self.pairs)

    def _load(self, path):
        with np.errstate(over="ignore"):
            spec = np.load(path).astype(np.float16)       # honours the header's fortran_order flag
        return torch.from_numpy(fit_to(spec, self.target_size).astype(np.float32)).unsqueeze(0)

    def __getitem__(self, idx):
        noisy_path, clean_path = self.pairs[idx]
        return self._load(noisy_path), self._load(clean_path)

    # ---- MI355X ingest -------------------------------------------------------------------------------
    def load_batch_to_device(self, indices, device="cuda"):
        """(noisy, clean) batches ``(B, 1, H, W)`` float32 on ``device`` for same-shaped source files."""
        noisy = np.stack([np.ascontiguousarray(np.load(self.pairs[i][0]), dtype=np.float32) for i in indices])
        clean = np.stack([np.ascontiguousarray(np.load(self.pairs[i][1]), dtype=np.float32) for i in indices])
        return (quantize_pad_on_device(torch.from_numpy(noisy).to(device), self.target_size),
                quantize_pad_on_device(torch.from_numpy(clean).to(device), self.target_size))


def quantize_pad_on_device(spec: torch.Tensor, target_size) -> torch.Tensor:
    """``spec`` (B, h, w) float32 on a ROCm device -> (B, 1, H, W) = fp32(fp16(spec)) cropped / zero padded."""
    if not spec.is_cuda or spec.dtype != torch.float32 or spec.dim() != 3:
        raise ValueError("quantize_pad_on_device: expected a (B, h, w) float32 tensor on a ROCm device")
    spec = spec.contiguous()
    b, h, w = spec.shape
    H, W = target_size
    out = torch.empty((b, 1, H, W), dtype=torch.float32, device=spec.device)
    stream = torch.cuda.current_stream(spec.device).cuda_stream
    with torch.cuda.device(spec.device):
        _lib.check(_lib.load().adn_quantize_pad(spec.data_ptr(), b, h, w, out.data_ptr(), H, W, stream),
                   "adn_quantize_pad")
    return out


class WavToSpecDataset(Dataset):
    """On-the-fly wav -> magnitude-spectrogram pairs: the dataset ``train.py`` imports
    (``/root/reference/code/train.py:16,106-109``: ``WavToSpecDataset(data_dir=..., subset_fraction=...)``) but whose
    module (``new_unet_data_loader``) is not part of the reference tree.  The contract is therefore assembled from
    the reference pieces on either side of it:

    * pairing as in ``SpectrogramDataset`` (``data_loader.py:17-31``): sorted ``clean*`` / ``noisy*`` files of one
      folder, equal counts asserted, here with the ``.wav`` suffix;
    * audio -> spectrogram as in ``audio_to_spectrogram`` (``create_test_dataset.py:35-41``): centred STFT,
      n_fft 512 / hop 128 by default, magnitude;
    * item format as in ``SpectrogramDataset.__getitem__`` (``data_loader.py:37-52``): fp32(fp16(.)), crop or
      bottom/right zero-pad to ``target_size``, ``(noisy, clean)`` each ``(1, H, W)`` float32.

    ``subset_fraction`` keeps the first ``max(1, int(n * fraction))`` pairs of the sorted list (deterministic).
    The STFT, the quantisation and the crop/pad run on the device (``adn_stft_mag`` + ``adn_quantize_pad``); there is
    no host STFT.  That fixes how the dataset is fed to a ``DataLoader`` (reference ``train.py:118-119`` uses
    ``num_workers=4, pin_memory=True``):

    * ``ds[i]`` (main process, ``num_workers=0``): device STFT, items come back as host tensors like the reference's
      datasets.  Inside a DataLoader worker that was FORKED from a parent that has already initialised HIP
      (``model.to(DEVICE)``, ``train.py:122``, then the default fork start method) it raises: such a child cannot use
      the GPU.  Workers started with ``multiprocessing_context="spawn"`` (or forked before anything touched the GPU)
      own a HIP context of their own and work -- at the price of one context per worker.
    * ``ds.loader(clip_samples, batch_size=16, num_workers=4, pin_memory=True, shuffle=True)``: a ``DataLoader`` over
      :meth:`audio_view` whose workers only decode wav files (host I/O) and collate fixed-length audio; the MAIN
      process then runs ONE batched device STFT per batch and yields ``(noisy, clean)`` batches ``(B, 1, H, W)``
      already resident in HBM -- the MI355X-shaped feed (a ``collate_fn`` cannot do this: it runs in the worker).
      ``subset=`` takes the ``Subset`` objects of ``random_split(ds, ...)`` (``train.py:111-114``) or a list of indices.
    * ``load_batch_to_device(indices)``: the same without a DataLoader.

    ``ds[i]`` transforms the WHOLE file and crops / zero-pads the spectrogram; the loader crops / zero-pads the AUDIO to
    ``clip_samples`` first.  Both give the same item when ``clip_samples >= min_clip_samples()`` =
    ``(W - 1) * hop + n_fft // 2``: then every frame inside ``target_size`` sees the same samples, and for a file SHORTER
    than ``clip_samples`` the frames that do not exist in its own STFT (index > L // hop; they would overlap the file's tail
    in the padded audio) are zeroed on the device from the true length the view hands along -- the loader rule's right
    zero-padding.  Below that bound a file longer than ``clip_samples`` loses samples the last frames would have seen;
    :meth:`audio_view` / :meth:`loader` refuse it unless ``allow_cut_frames=True``.

    No resampling: ``sample_rate`` (if given) is checked against each file.
    """

    def __init__(self, data_dir, subset_fraction: float = 1.0, target_size=(256, 64), n_fft: int = 512,
                 hop_length: int = 128, sample_rate=None, device="cuda"):
        if not 0.0 < subset_fraction <= 1.0:
            raise ValueError("subset_fraction must be in (0, 1]")
        self.target_size = tuple(target_size)
        self.n_fft, self.hop_length, self.sample_rate, self.device = n_fft, hop_length, sample_rate, device
        clean = _list(data_dir, "clean", ".wav")
        noisy = _list(data_dir, "noisy", ".wav")
        print(f"Found {len(clean)} clean files and {len(noisy)} noisy files in {data_dir}")
        assert len(clean) == len(noisy), f"Mismatch in {data_dir}"
        pairs = list(zip(noisy, clean))
        keep = max(1, int(len(pairs) * subset_fraction)) if pairs else 0
        self.pairs = pairs[:keep]
        print(f"Total pairs loaded: {len(self.pairs)}")

    def __len__(self):
        return len(self.pairs)

    def _audio(self, path):
        audio, rate = read_wav(path, mono=True)
        if self.sample_rate is not None and rate != self.sample_rate:
            raise ValueError(f"{path}: sample rate {rate} != expected {self.sample_rate} (no resampler in this build)")
        return audio

    def _spec_batch(self, audios):
        """(B, L) float32 audio (array, list of equally long arrays, or tensor) -> (B, 1, H, W) float32 on the device."""
        from .stft import stft_magnitude_fit
        a = audios if isinstance(audios, torch.Tensor) else torch.from_numpy(np.stack(audios))
        a = a.to(self.device, non_blocking=True)
        # STFT + fp16 round trip + crop/pad in ONE kernel: only the frames inside target_size are computed
        return stft_magnitude_fit(a, self.target_size, self.n_fft, self.hop_length, True)

    def __getitem__(self, idx):
        if torch.utils.data.get_worker_info() is not None and _forked_from_gpu_parent():
            raise RuntimeError(
                "WavToSpecDataset computes its spectrograms on the GPU (there is no host STFT) and a DataLoader worker "
                "process forked from a GPU-initialised parent cannot use HIP.  Use num_workers=0, "
                "multiprocessing_context='spawn', or let the workers decode audio only and transform in the main "
                "process: ds.loader(clip_samples, batch_size=..., num_workers=4)")
        noisy_path, clean_path = self.pairs[idx]
        noisy, clean = self._audio(noisy_path), self._audio(clean_path)
        if len(noisy) == len(clean):
            both = self._spec_batch([noisy, clean]).cpu()
            return both[0], both[1]
        return self._spec_batch([noisy]).cpu()[0], self._spec_batch([clean]).cpu()[0]

    def load_batch_to_device(self, indices):
        """(noisy, clean) batches ``(B, 1, H, W)`` on the device; clips of one call must have one length."""
        noisy = [self._audio(self.pairs[i][0]) for i in indices]
        clean = [self._audio(self.pairs[i][1]) for i in indices]
        return self._spec_batch(noisy), self._spec_batch(clean)

    # ---- DataLoader feed: host-only items in the workers, device transform in the main process ---------------
    def min_clip_samples(self) -> int:
        """Samples the frames inside ``target_size`` reach: the last one (index W - 1, centred STFT) ends at
        ``(W - 1) * hop + n_fft // 2``.  An audio crop at least this long leaves every item equal to ``ds[i]``."""
        return (self.target_size[1] - 1) * self.hop_length + self.n_fft // 2

    def audio_view(self, clip_samples: int, allow_cut_frames: bool = False):
        """Host-only ``Dataset`` of ``(noisy_audio, clean_audio)`` float32 tensors cropped / zero-padded at the end to
        ``clip_samples`` -- safe in DataLoader worker processes (wav decoding only, no GPU).  ``clip_samples`` below
        :meth:`min_clip_samples` would cut samples off frames inside ``target_size`` for longer files (items would
        differ from ``ds[i]``): refused unless ``allow_cut_frames=True``."""
        clip_samples = int(clip_samples)
        if clip_samples < self.min_clip_samples() and not allow_cut_frames:
            raise ValueError(
                f"clip_samples={clip_samples} is shorter than the {self.min_clip_samples()} samples the {self.target_size[1]} frames "
                f"of target_size reach (n_fft {self.n_fft}, hop {self.hop_length}): files longer than clip_samples would give "
                "other items than ds[i].  Pass a longer clip or allow_cut_frames=True")
        return _WavAudioView(self, clip_samples)

    def to_device_batch(self, host_batch):
        """``(noisy_audio (B, L), clean_audio (B, L), noisy_len (B,), clean_len (B,))`` host tensors (what a DataLoader
        over :meth:`audio_view` yields; the lengths are the files' true sample counts) -> ``(noisy, clean)`` each
        ``(B, 1, H, W)`` float32 on the device: one batched STFT + quantise + crop/pad per side, then the frames a short
        file's own STFT does not have (index > len // hop) are zeroed.  Must run in the process that owns the GPU context
        (the DataLoader's consumer, not its workers)."""
        noisy, clean = host_batch[0], host_batch[1]
        out = [self._spec_batch(noisy), self._spec_batch(clean)]
        if len(host_batch) >= 4:
            frames = torch.arange(self.target_size[1], device=out[0].device)
            for k in range(2):
                nfr = 1 + torch.as_tensor(host_batch[2 + k]).to(out[k].device).clamp(max=noisy.shape[1]) // self.hop_length
                keep = frames[None, :] < nfr[:, None]                       # (B, W)
                out[k] = torch.where(keep[:, None, None, :], out[k], torch.zeros((), device=out[k].device))
        return out[0], out[1]

    def loader(self, clip_samples: int, subset=None, allow_cut_frames: bool = False, **dataloader_kwargs):
        """Iterable with the ``DataLoader`` call shape of ``train.py:118-119`` (``batch_size``, ``shuffle``,
        ``num_workers``, ``pin_memory``, ...) that yields device-resident spectrogram batches.  ``subset``: a
        ``torch.utils.data.Subset`` of this dataset (what ``random_split`` returns, ``train.py:111-114``) or a sequence of
        indices -- the feed then covers those items only."""
        from torch.utils.data import DataLoader, Subset
        if "collate_fn" in dataloader_kwargs:
            raise ValueError("loader(): the collate function is fixed (stacked fixed-length audio)")
        view = self.audio_view(clip_samples, allow_cut_frames)
        if subset is not None:
            if isinstance(subset, Subset):
                if subset.dataset is not self:
                    raise ValueError("loader(subset=...): the Subset must be a split of this dataset")
                subset = subset.indices
            view = Subset(view, [int(i) for i in subset])
        return _DeviceSpecLoader(self, DataLoader(view, **dataloader_kwargs))


def _forked_from_gpu_parent() -> bool:
    """True in a process forked from a parent that had already initialised the GPU through torch: HIP cannot be used
    there.  False in spawned workers and in workers forked before anything touched the GPU (both can open a context)."""
    probe = getattr(torch.cuda, "_is_in_bad_fork", None)
    return bool(probe()) if probe is not None else True


class _DeviceSpecLoader:
    def __init__(self, parent: "WavToSpecDataset", host_loader):
        self.parent, self.host_loader = parent, host_loader

    def __len__(self):
        return len(self.host_loader)

    def __iter__(self):
        for host_batch in self.host_loader:
            yield self.parent.to_device_batch(host_batch)


class _WavAudioView(Dataset):
    def __init__(self, parent: "WavToSpecDataset", clip_samples: int):
        if clip_samples < parent.n_fft // 2 + 1:
            raise ValueError("clip_samples too short for one centred STFT frame")
        self.parent, self.clip_samples = parent, clip_samples

    def __len__(self):
        return len(self.parent)

    def _fit(self, audio):
        out = np.zeros(self.clip_samples, dtype=np.float32)
        n = min(self.clip_samples, len(audio))
        out[:n] = audio[:n]
        return torch.from_numpy(out)

    def __getitem__(self, idx):
        noisy_path, clean_path = self.parent.pairs[idx]
        noisy, clean = self.parent._audio(noisy_path), self.parent._audio(clean_path)
        return self._fit(noisy), self._fit(clean), len(noisy), len(clean)
