"""CPU-side checks: the C-ABI library loads and exports every declared symbol, host-only entry points,
the loader mirror against reference goldens, weight generator invariants, and the 2-rank gloo path."""
import ctypes
import os
import re
import socket
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from audiodenoiser_amd import _lib
    L = _lib.load()
    header = open(os.path.join(ROOT, "include", "adn.h")).read()
    declared = set(re.findall(r"\b(adn_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(L, name), f"libadn.so does not export {name}"
    assert declared == set(_lib.EXPORTED_SYMBOLS)
    assert L.adn_version() >= 1
    # ... and NOTHING else: the dynamic symbol table holds exactly the header's functions (no internal launchers, no weak
    # instantiations of C++ runtime templates) -- -fvisibility=hidden + csrc/libadn.map
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", L._name], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1].split("@")[0] for ln in out.splitlines() if ln.split() and ln.split()[-2] in "TWVBDR"}
    assert exported == declared, sorted(exported ^ declared)


def test_host_only_entry_points_and_errors():
    from audiodenoiser_amd import _lib
    L = _lib.load()
    need = ctypes.c_size_t()
    assert L.adn_unet_workspace_bytes(None, 1, 513, 256, ctypes.byref(need)) == 0
    one = need.value
    assert L.adn_unet_workspace_bytes(None, 4, 513, 256, ctypes.byref(need)) == 0
    # activations scale with N; the split-K scratch of the deep layers (small batches only) does not
    assert 3.3 * one <= need.value <= 4 * one and one > 513 * 256 * 64 * 4 * 2
    assert L.adn_unet_workspace_bytes(None, 1, 8, 256, ctypes.byref(need)) == 1      # ADN_ERR_INVALID
    assert b"F,T>=16" in L.adn_last_error()
    nfr = ctypes.c_long()
    for (length, n_fft, hop, center, expect) in ((16000, 512, 128, 0, 122), (24000, 512, 128, 1, 188),
                                                 (132300, 1024, 256, 1, 517), (100, 512, 128, 0, 0)):
        assert L.adn_stft_n_frames(length, n_fft, hop, center, ctypes.byref(nfr)) == 0
        assert nfr.value == expect
    assert L.adn_stft_mag(None, 1, 1000, 512, 128, 0, None, None) == 1
    buf = (ctypes.c_float * 16)()
    pbuf = ctypes.cast(buf, ctypes.c_void_p)
    assert L.adn_stft_mag_fit(None, 1, 1000, 512, 128, 1, None, 256, 64, None) == 1
    assert L.adn_stft_mag_fit(pbuf, 1, 1000, 500, 128, 1, pbuf, 256, 64, None) == 1        # n_fft not a power of two
    assert L.adn_stft_mag_fit(pbuf, 1, 100, 512, 128, 0, pbuf, 256, 64, None) == 1         # shorter than n_fft
    assert L.adn_stft_mag_fit(pbuf, 1, 1000, 512, 128, 1, pbuf, 0, 64, None) == 1
    assert L.adn_unet_forward(None, None, None, 1, 16, 16, None, 0, None) == 1


def test_model_parameter_tree_and_guards(weights_np):
    from audiodenoiser_amd.model import UNet
    from audiodenoiser_amd.weights import state_dict_schema
    m = UNet(1, 1)
    sd = m.state_dict()
    schema = state_dict_schema()
    assert list(sd.keys()) == list(schema.keys()) and len(sd) == 136
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(schema[k]), k
    assert sum(p.numel() for p in m.parameters()) == 31042369
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in weights_np.items()}, strict=True)
    with pytest.raises(RuntimeError, match="train-mode"):
        m(torch.zeros(1, 1, 16, 16))
    m.eval()
    with pytest.raises(RuntimeError, match="no_grad"):
        m(torch.zeros(1, 1, 16, 16))
    with torch.no_grad(), pytest.raises(RuntimeError, match="ROCm device"):
        m(torch.zeros(1, 1, 16, 16))
    with pytest.raises(ValueError):
        UNet(65, 1)                                       # 1..64 input planes, 1..64 classes
    with pytest.raises(ValueError):
        UNet(1, 0)
    # UNet(in_channels, num_classes) as the reference declares it (model.py:54,56,68): same 136 keys, two shapes differ
    m23 = UNet(2, 3)
    sd23 = m23.state_dict()
    schema23 = state_dict_schema(2, 3)
    assert list(sd23.keys()) == list(schema23.keys())
    for k, v in sd23.items():
        assert tuple(v.shape) == tuple(schema23[k]), k
    assert tuple(sd23["downconv1.conv.double_conv.0.weight"].shape) == (64, 2, 3, 3) and tuple(sd23["out.weight"].shape) == (3, 64, 1, 1)


def test_dataset_mirror_matches_reference_golden(tmp_path, golden_dir, capsys):
    from audiodenoiser_amd.data_loader import SpectrogramDataset
    from audiodenoiser_amd.weights import hash_uniform
    g = np.load(os.path.join(golden_dir, "loader_cases.npz"))
    for ci in range(4):
        shape = tuple(g[f"case{ci}_in_shape"])
        target = tuple(g[f"case{ci}_target"])
        u = hash_uniform(5, f"loader{ci}", 2 * shape[0] * shape[1]).reshape(2, *shape)
        noisy = (u[0] * np.float32(8.0)).astype(np.float32)
        clean = (u[1] * np.float32(8.0)).astype(np.float32)
        noisy[0, 0], noisy[0, 1], noisy[0, 2], noisy[1, 0] = 70000.0, 1e-8, 3e-6, 65504.0
        d = tmp_path / f"case{ci}"
        d.mkdir()
        np.save(d / "noisy_a_chunk_0.npy", noisy)
        np.save(d / "clean_a_chunk_0.npy", np.asfortranarray(clean))   # fortran_order header flag
        (d / "other.txt").write_text("ignored")
        ds = SpectrogramDataset(str(d), target_size=target)
        assert len(ds) == 1
        n_t, c_t = ds[0]
        assert n_t.dtype == torch.float32 and tuple(n_t.shape) == (1,) + target
        assert np.array_equal(n_t.numpy(), g[f"case{ci}_noisy"])
        assert np.array_equal(c_t.numpy(), g[f"case{ci}_clean"])
    out = capsys.readouterr().out
    assert "Found 1 clean files and 1 noisy files" in out and "Total pairs loaded: 1" in out


def test_dataset_pairing_and_mismatch(tmp_path):
    from audiodenoiser_amd.data_loader import SpectrogramDataset
    for k in (2, 10, 1):
        np.save(tmp_path / f"noisy_white_chunk_{k}.npy", np.full((4, 4), k, np.float32))
        np.save(tmp_path / f"clean_white_chunk_{k}.npy", np.full((4, 4), -k, np.float32))
    ds = SpectrogramDataset(str(tmp_path), target_size=(4, 4))
    # lexicographic order (chunk_1, chunk_10, chunk_2) on both lists keeps pairs consistent
    for i in range(3):
        n, c = ds[i]
        assert float(n[0, 0, 0]) == -float(c[0, 0, 0])
    np.save(tmp_path / "noisy_extra.npy", np.zeros((4, 4), np.float32))
    with pytest.raises(AssertionError):
        SpectrogramDataset(str(tmp_path))


def test_weight_generator_is_deterministic_and_windowed():
    from audiodenoiser_amd.weights import hash_uniform, make_state_dict
    a = hash_uniform(1, "k", 1000)
    b = hash_uniform(1, "k", 500, offset=500)
    assert np.array_equal(a[500:], b) and a.min() >= 0 and a.max() < 1
    assert not np.array_equal(a, hash_uniform(2, "k", 1000))
    sd = make_state_dict(1234)
    assert abs(float(sd["downconv2.conv.double_conv.0.weight"].std()) - np.sqrt(2.0 / (64 * 9))) < 1e-3
    assert float(sd["bottleneck.double_conv.1.running_var"].min()) >= 0.75


def test_shard_range():
    from audiodenoiser_amd.distributed import shard_range
    assert shard_range(2048, 3, 8) == (768, 1024)
    assert [shard_range(8, r, 2) for r in range(2)] == [(0, 4), (4, 8)]
    with pytest.raises(ValueError):
        shard_range(10, 0, 4)
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gloo_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import torch as th
    import oracle
    from audiodenoiser_amd import distributed as D
    from audiodenoiser_amd.weights import make_input, make_state_dict
    th.set_num_threads(2)
    D.init_from_env("gloo")
    total = 4
    lo, hi = D.shard_range(total, rank, world)
    x = make_input(11, total, 16, 16)
    tgt = make_input(12, total, 16, 16)
    y = oracle.unet_forward(make_state_dict(1234), x[lo:hi])          # stand-in compute for the CPU test
    local = th.from_numpy(np.abs(y - tgt[lo:hi]).reshape(hi - lo, -1).mean(axis=1).astype(np.float32))
    D.barrier()
    allv = D.gather_per_clip(local)
    tmax = D.max_over_ranks(float(rank + 1), "cpu")
    q.put((rank, allv.numpy().copy(), tmax))
    th.distributed.destroy_process_group()


def test_two_rank_gloo_gather(weights_np):
    """World-size-2 run of the multi-GPU host path on CPU (gloo): shard, per-clip value, all-gather, max-reduce."""
    import torch.multiprocessing as mp
    import oracle
    from audiodenoiser_amd.weights import make_input
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    y = oracle.unet_forward(weights_np, make_input(11, 4, 16, 16))
    ref = np.abs(y - make_input(12, 4, 16, 16)).reshape(4, -1).mean(axis=1).astype(np.float32)
    for rank, allv, tmax in res:
        assert allv.shape == (4,)
        assert np.array_equal(allv, ref)      # same arithmetic per clip -> identical, in rank order
        assert tmax == 2.0


def _bench_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import torch as th
    import bench
    from audiodenoiser_amd import distributed as D
    th.set_num_threads(1)
    D.init_from_env("gloo")
    b = bench.default_batch_per_gpu(world)
    calls = [0]

    def step():                                             # stub of forward + per-clip loss: 4 floats per clip, = rank
        calls[0] += 1
        return D.gather_per_clip(th.full((4 * b,), float(rank), dtype=th.float32))

    started = []
    tr = bench.timed_region(step, 3, 2, lambda: None, "cpu", on_timed_start=lambda: started.append(calls[0]),
                            gather_probe=lambda: D.gather_per_clip(th.zeros(4 * b)))
    line = bench.assemble_line(tr["elapsed_s"], 3, 2, world, b, "f32", tr["ranks"]) if rank == 0 else None
    q.put((rank, calls[0], started[0], tr["last"].numpy().copy(), tr["elapsed_s"], line))
    th.distributed.destroy_process_group()


def test_bench_timed_region_and_line_world8_gloo():
    """bench.py's timed region and JSON assembly at world size 8 (gloo, stub step): the default batch is BASELINE
    configs[3]'s 256 clips per GPU (2048 in all), the line reports how many ranks the gathered tensor really held, every
    rank runs exactly warmup + steps steps, and all ranks agree on the max-over-ranks time."""
    import torch.multiprocessing as mp
    import bench
    assert bench.default_batch_per_gpu(1) == 64 and bench.default_batch_per_gpu(2) == 256 and bench.default_batch_per_gpu(8) == 256
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    elapsed = {round(r[4], 9) for r in res}
    assert len(elapsed) == 1                                 # MAX over ranks, identical everywhere
    for rank, calls, started, last, _, line in res:
        assert calls == 5 and started == 2                   # 2 warm-up steps, then exactly 3 timed ones
        assert last.shape == (world * 4 * 256,)
        assert np.array_equal(last.reshape(world, -1), np.repeat(np.arange(world, dtype=np.float32)[:, None], 1024, axis=1))
    line = res[0][5]
    assert line["n_gpus"] == 8 and line["steps"] == 3 and line["warmup"] == 2 and line["scaling"] == "weak"
    assert line["config"]["batch_per_gpu"] == 256 and line["config"]["global_batch"] == 2048
    assert "configs[3]" in line["config"]["workload"] and "2048" in line["config"]["workload"]
    assert line["ranks"]["ranks_seen"] == 8 and line["ranks"]["rank_ids"] == list(range(8)) and line["ranks"]["world_size"] == 8
    assert line["ranks"]["allgather_ms"] is not None and line["ranks"]["allgather_ms"] > 0
    assert line["ranks"]["rank_ms_per_step_min"] <= line["ranks"]["rank_ms_per_step_max"] <= line["ms_per_step"] * 1.5
    assert abs(line["value"] - 2048 * 256 * 3 / res[0][4]) < 1.0
    assert line["metric"].startswith("spectrogram frames/sec") and line["unit"] == "frames/s" and line["vs_baseline"] is None
    one = bench.assemble_line(1.0, 10, 2, 1, 64, "f32", {"ranks_seen": 1})
    assert "configs[1]" in one["config"]["workload"] and one["value"] == 64 * 256 * 10


def test_wav_reader_formats(tmp_path):
    """PCM16 (checked against the stdlib reader), float32, stereo mix-down, extensible header, odd chunk padding."""
    import struct
    import wave
    from audiodenoiser_amd.wav import read_wav, write_wav
    rng = np.random.default_rng(3)
    a = (rng.uniform(-1, 1, 1001) * 0.9).astype(np.float32)
    p16 = str(tmp_path / "a16.wav")
    write_wav(p16, a, 8000, "PCM_16")
    with wave.open(p16, "rb") as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (1, 2, 8000, 1001)
        ref = np.frombuffer(w.readframes(1001), dtype="<i2").astype(np.float32) / 32768.0
    got, rate = read_wav(p16)
    assert rate == 8000 and got.dtype == np.float32 and np.array_equal(got, ref)
    assert np.max(np.abs(got - a)) <= 0.5 / 32768 + 1e-7
    pf = str(tmp_path / "af.wav")
    write_wav(pf, a, 44100, "FLOAT")
    got, rate = read_wav(pf)
    assert rate == 44100 and np.array_equal(got, a)
    st = np.stack([a, -0.5 * a], axis=1)
    ps = str(tmp_path / "st.wav")
    write_wav(ps, st, 8000, "FLOAT")
    mono, _ = read_wav(ps)
    assert np.allclose(mono, 0.25 * a, atol=1e-7)
    both, _ = read_wav(ps, mono=False)
    assert both.shape == (1001, 2) and np.array_equal(both, st)
    # WAVE_FORMAT_EXTENSIBLE header + a LIST chunk of odd size before the data + 8-bit PCM
    pcm8 = np.array([0, 64, 128, 192, 255], dtype=np.uint8)
    fmt = struct.pack("<HHIIHH", 0xFFFE, 1, 8000, 8000, 1, 8) + struct.pack("<HHI", 22, 8, 4) + \
        struct.pack("<H", 1) + b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71"
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"LIST" + struct.pack("<I", 3) + b"abc\x00" + \
        b"data" + struct.pack("<I", 5) + pcm8.tobytes() + b"\x00"
    pe = tmp_path / "ext.wav"
    pe.write_bytes(b"RIFF" + struct.pack("<I", len(body)) + body)
    got, rate = read_wav(str(pe))
    assert rate == 8000 and np.array_equal(got, (pcm8.astype(np.float32) - 128) / 128)
    bad = tmp_path / "bad.wav"
    bad.write_bytes(b"RIFX" + b"\x00" * 40)
    with pytest.raises(ValueError):
        read_wav(str(bad))


def test_wav_dataset_discovery_and_subset(tmp_path, capsys):
    from audiodenoiser_amd.data_loader import WavToSpecDataset
    from audiodenoiser_amd.wav import write_wav
    for i in range(5):
        write_wav(str(tmp_path / f"clean_{i}.wav"), np.zeros(800, np.float32), 8000)
        write_wav(str(tmp_path / f"noisy_{i}.wav"), np.zeros(800, np.float32), 8000)
    (tmp_path / "clean_notes.txt").write_text("ignored")
    ds = WavToSpecDataset(str(tmp_path), subset_fraction=0.5)
    out = capsys.readouterr().out
    assert "Found 5 clean files and 5 noisy files" in out and "Total pairs loaded: 2" in out
    assert len(ds) == 2
    assert [os.path.basename(n) for n, _ in ds.pairs] == ["noisy_0.wav", "noisy_1.wav"]
    assert [os.path.basename(c) for _, c in ds.pairs] == ["clean_0.wav", "clean_1.wav"]
    assert len(WavToSpecDataset(str(tmp_path))) == 5
    assert len(WavToSpecDataset(str(tmp_path), subset_fraction=0.01)) == 1
    with pytest.raises(ValueError):
        WavToSpecDataset(str(tmp_path), subset_fraction=0.0)
    os.remove(tmp_path / "noisy_4.wav")
    with pytest.raises(AssertionError):
        WavToSpecDataset(str(tmp_path))


def test_wav_dataset_worker_guard_and_audio_view(tmp_path):
    """train.py:118-119 wraps the dataset in DataLoader(num_workers=4): __getitem__ needs the GPU, so inside a worker
    it must fail with a clear message (not hang on a forked HIP context); the audio view is host-only and works in
    workers; ds.loader() runs the device STFT on their batches in the main process."""
    from torch.utils.data import DataLoader
    from audiodenoiser_amd.data_loader import WavToSpecDataset
    from audiodenoiser_amd.wav import write_wav
    rng = np.random.default_rng(2)
    for i, n in enumerate((900, 1200, 700, 1000)):
        write_wav(str(tmp_path / f"clean_{i}.wav"), rng.uniform(-1, 1, n).astype(np.float32), 8000, "FLOAT")
        write_wav(str(tmp_path / f"noisy_{i}.wav"), rng.uniform(-1, 1, n).astype(np.float32), 8000, "FLOAT")
    ds = WavToSpecDataset(str(tmp_path), sample_rate=8000)
    if torch.cuda.is_available():                          # (forked from a GPU-initialised parent: covered by the -m gpu test)
        torch.zeros(1, device="cuda")
        with pytest.raises(RuntimeError, match="DataLoader worker"):
            next(iter(DataLoader(ds, batch_size=2, num_workers=1)))
    # ds[i] transforms the whole file and crops the spectrogram, the loader crops the audio first: the two agree when the crop
    # keeps every sample the frames inside target_size reach -- (64 - 1) * 128 + 256 = 8320 for the defaults
    assert ds.min_clip_samples() == 8320
    with pytest.raises(ValueError, match="8320"):
        ds.audio_view(1000)
    with pytest.raises(ValueError, match="allow_cut_frames"):
        ds.loader(8000, batch_size=2)
    assert len(ds.audio_view(8320)) == 4
    view = ds.audio_view(1000, allow_cut_frames=True)
    assert len(view) == 4
    batches = list(DataLoader(view, batch_size=2, num_workers=2))
    assert len(batches) == 2
    for noisy, clean, n_len, c_len in batches:
        assert noisy.shape == clean.shape == (2, 1000) and noisy.dtype == torch.float32
        assert n_len.shape == c_len.shape == (2,)
    assert [int(v) for b in batches for v in b[2]] == [900, 1200, 700, 1000]      # the files' true lengths travel along
    n0 = view[0][0]                                       # 900 samples -> zero padded at the end
    assert float(n0[900:].abs().max()) == 0.0 and float(n0[:900].abs().max()) > 0.0
    n1 = view[1][0]                                       # 1200 samples -> cropped
    from audiodenoiser_amd.wav import read_wav
    assert np.array_equal(n1.numpy(), read_wav(str(tmp_path / "noisy_1.wav"))[0][:1000])
    with pytest.raises(ValueError):
        ds.audio_view(100, allow_cut_frames=True)
    with pytest.raises(ValueError):
        ds.loader(1000, allow_cut_frames=True, collate_fn=lambda b: b)
    assert len(ds.loader(1000, allow_cut_frames=True, batch_size=3, num_workers=2)) == 2
    # train.py:111-114 splits the dataset with random_split before it builds the loaders: the Subset objects feed loader()
    from torch.utils.data import random_split
    train_ds, val_ds = random_split(ds, [3, 1], generator=torch.Generator().manual_seed(0))
    lt, lv = ds.loader(8320, subset=train_ds, batch_size=2), ds.loader(8320, subset=val_ds, batch_size=2)
    assert len(lt) == 2 and len(lv) == 1
    got = [int(i) for i in lt.host_loader.dataset.indices] + [int(i) for i in lv.host_loader.dataset.indices]
    assert sorted(got) == [0, 1, 2, 3] and got[:3] == list(train_ds.indices)
    assert len(ds.loader(8320, subset=[2, 0], batch_size=1)) == 2
    other = WavToSpecDataset(str(tmp_path), sample_rate=8000)
    with pytest.raises(ValueError, match="split of this dataset"):
        other.loader(8320, subset=train_ds)


def test_cpu_tensors_without_a_device_raise_instead_of_computing(weights_np):
    """test.py hands CPU tensors to a CPU-resident model; the mirror stages them on the current ROCm device.  On a
    machine with no device that must raise -- there is no CPU arithmetic anywhere in the product path."""
    if torch.cuda.is_available():
        pytest.skip("this box has a ROCm device; covered by the -m gpu drop-in test")
    from audiodenoiser_amd._lib import AdnError
    from audiodenoiser_amd.loss import CombinedPerceptualLoss
    from audiodenoiser_amd.model import UNet
    m = UNet(1, 1).eval()
    with torch.no_grad(), pytest.raises(AdnError, match="no ROCm device"):
        m(torch.zeros(1, 1, 16, 16))
    with torch.no_grad(), pytest.raises(AdnError, match="no ROCm device"):
        CombinedPerceptualLoss()(torch.zeros(1, 1, 16, 64), torch.zeros(1, 1, 16, 64))
    with torch.no_grad(), pytest.raises(AdnError, match="no ROCm device"):
        next(m.forward_host_batches([torch.zeros(1, 1, 16, 16)]))


def test_perceptual_loss_frame_limit_is_reported_before_any_launch():
    from audiodenoiser_amd import _lib
    L = _lib.load()
    tlds = 6784                                   # up to here a clip's series and mel frames live in one CU's LDS (adn.h)
    need = ctypes.c_size_t()
    assert L.adn_perceptual_loss_workspace_bytes(2, 513, tlds, ctypes.byref(need)) == 0
    assert need.value == 2 * 17 * (2 * tlds + 1) * 4
    # longer clips (round 5: no limit, as in the reference): the two series of every clip join the workspace
    assert L.adn_perceptual_loss_workspace_bytes(2, 513, tlds + 1, ctypes.byref(need)) == 0
    assert need.value == (2 * 17 * (2 * (tlds + 1) + 1) + 2 * 2 * (tlds + 1)) * 4
    assert L.adn_perceptual_loss_workspace_bytes(1, 513, 1 << 25, ctypes.byref(need)) == 1
    buf = (ctypes.c_float * 16)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert L.adn_perceptual_loss(p, p, 1, 64, 1 << 25, p, 1 << 40, p, None) == 1     # rejected on the host
    assert b"2^24" in L.adn_last_error()
    assert L.adn_perceptual_loss(p, p, 1, 64, 31, p, 1 << 40, p, None) == 1           # reflect pad of 31 needs T >= 32 (loss.py:39-41)
    assert b"32 <= T" in L.adn_last_error()
    assert L.adn_perceptual_loss_workspace_bytes(1, 64, 31, ctypes.byref(need)) == 1
    assert L.adn_perceptual_loss_workspace_bytes(1, 64, 32, ctypes.byref(need)) == 0 and need.value == 2 * (2 * 32 + 1) * 4


def test_compat_modules_resolve_reference_import_names():
    """PYTHONPATH=compat: the reference's `from model import UNet` etc. pick up the MI355X mirror."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    want = {"model": ["UNet"], "data_loader": ["SpectrogramDataset"], "loss": ["CombinedPerceptualLoss"],
            "new_unet_data_loader": ["WavToSpecDataset"]}
    for mod, names in want.items():
        spec = importlib.util.spec_from_file_location(f"_compat_{mod}", os.path.join(root, "compat", f"{mod}.py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        for n in names:
            assert getattr(m, n).__module__.startswith("audiodenoiser_amd.")


def test_griffin_lim_entry_points_validate_arguments_without_a_device():
    """Argument checks of the inverse-STFT / Griffin-Lim C ABI run before any HIP call (status 1 = ADN_ERR_INVALID,
    3 = ADN_ERR_WORKSPACE), so they are testable on the CPU-only box."""
    from audiodenoiser_amd import _lib
    L = _lib.load()
    n = ctypes.c_long()
    assert L.adn_istft_length(188, 128, ctypes.byref(n)) == 0 and n.value == 128 * 187
    assert L.adn_istft_length(0, 128, ctypes.byref(n)) == 1
    need = ctypes.c_size_t()
    assert L.adn_griffin_lim_workspace_bytes(2, 257, 188, ctypes.byref(need)) == 0
    assert need.value == 2 * 188 * (257 * 2 + 512) * 4
    assert L.adn_griffin_lim_workspace_bytes(0, 257, 188, ctypes.byref(need)) == 1
    assert L.adn_istft_workspace_bytes(2, 188, 512, ctypes.byref(need)) == 0 and need.value == 2 * 188 * 512 * 4
    assert L.adn_istft_workspace_bytes(2, 188, 500, ctypes.byref(need)) == 1
    buf = (ctypes.c_float * 16)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert L.adn_griffin_lim(None, None, 1, 257, 188, 512, 128, 50, None, 0, None, None) == 1
    assert L.adn_griffin_lim(p, p, 1, 257, 188, 500, 128, 50, p, 64, p, None) == 1            # n_fft not a power of two
    assert L.adn_griffin_lim(p, p, 1, 200, 188, 512, 128, 50, p, 64, p, None) == 1            # n_bins != n_fft/2+1
    assert b"n_bins" in L.adn_last_error()
    assert L.adn_griffin_lim(p, p, 1, 257, 188, 512, 128, 50, p, 64, p, None) == 3            # workspace too small
    assert L.adn_istft(p, 1, 188, 512, 1024, p, 1 << 30, p, None) == 1                         # hop > n_fft
    assert L.adn_stft_complex(None, 1, 1000, 512, 128, None, None) == 1


def test_forward_shape_limits_are_reported():
    from audiodenoiser_amd import _lib
    L = _lib.load()
    need = ctypes.c_size_t()
    assert L.adn_unet_workspace_bytes(None, 1, 64, 4094, ctypes.byref(need)) == 0
    assert L.adn_unet_workspace_bytes(None, 1, 64, 65536, ctypes.byref(need)) == 0           # no limit on T by itself (round 5)
    assert L.adn_unet_workspace_bytes(None, 1, 8192, 4000, ctypes.byref(need)) == 0 and need.value > 8192 * 4000 * 64 * 4 * 2
    assert L.adn_unet_workspace_bytes(None, 1, 513, 261000, ctypes.byref(need)) == 0         # F*T just below 2^27
    assert L.adn_unet_workspace_bytes(None, 1, 8192, 16384, ctypes.byref(need)) == 1         # F*T = 2^27
    assert b"F*T<2^27" in L.adn_last_error()


def test_bf16_three_term_split_is_fp32_accurate():
    """The arithmetic behind conv_dma<..., SPLIT> (fp32 transposed convolutions on the bf16 matrix cores): x = hi + mid + lo with
    three round-to-nearest bf16 terms reconstructs an fp32 value to 2^-24, and the six products of order <= 2 summed in fp32 match
    an fp32 dot product's accuracy (numpy restatement of pack_convt_split / split3_bf16; the GPU form is checked against the
    exact-fp32 MFMA form in tests/test_gpu_parity.py::test_convt_split_bf16_matches_exact_fp32_form)."""
    def bf16(x):                                             # round to nearest even, as v_cvt_pk_bf16_f32 / bf16_rne
        u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
        u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16 << 16
        return u.astype(np.uint32).view(np.float32)

    def split(x):
        hi = bf16(x)
        r1 = (x - hi).astype(np.float32)
        mid = bf16(r1)
        lo = bf16((r1 - mid).astype(np.float32))
        return hi, mid, lo

    rng = np.random.default_rng(0)
    x = (rng.standard_normal(4096) * np.exp(rng.uniform(-20, 20, 4096))).astype(np.float32)      # 17 decades of magnitude
    hi, mid, lo = split(x)
    assert np.all(np.abs((hi.astype(np.float64) + mid + lo) - x) <= np.abs(x) * 2.0 ** -24)
    a = np.maximum(rng.standard_normal((64, 1024)), 0).astype(np.float32) * 3
    w = (rng.standard_normal((1024, 32)) * 0.02).astype(np.float32)
    ref = a.astype(np.float64) @ w.astype(np.float64)
    (ah, am, al), (wh, wm, wl) = split(a), split(w)
    six = (al @ wh + ah @ wl + am @ wm + am @ wh + ah @ wm + ah @ wh).astype(np.float32)
    m = np.abs(ref).max()
    e_six, e_f32, e_one = np.abs(six - ref).max() / m, np.abs(a @ w - ref).max() / m, np.abs(ah @ wh - ref).max() / m
    assert e_six <= 2e-6 and e_six <= 4 * e_f32 + 1e-7 and e_one > 1e-3        # one bf16 product alone is nowhere near


def test_roofline_accounting():
    """bench.py's roofline inputs: SURVEY 8d totals, and executed matrix-core FLOPs (padded tiles counted)."""
    from audiodenoiser_amd.roofline import executed_mfma_flops, totals, unet_launches
    flops, act, wts = totals(513, 256)
    assert abs(flops / 1e9 - 192.443) < 1e-3 and abs(act / 1e6 - 670.63) < 0.1 and abs(wts / 1e6 - 124.12) < 0.05
    ls = unet_launches(513, 256)
    c3 = [l for l in ls if l["kind"] == "conv3x3"]
    assert len(c3) == 17
    from audiodenoiser_amd.roofline import winograd_tile
    for l in c3:
        w2, d = executed_mfma_flops(l, "winograd", "2"), executed_mfma_flops(l, "direct")
        assert w2 <= l["flops"] / 2.25 * 1.03 and w2 >= l["flops"] / 2.25        # only 513 -> 528 row padding on top
        assert l["flops"] <= d <= l["flops"] * 1.03
        # F(4x4,3x3): a quarter of the direct count over the 16x16-pixel blocks that touch the image (513 -> 528 rows: the
        # blocks of rows 528-543 of the 32x32 tiles do no arithmetic); the 32x16 bottleneck runs in pair mode (two clips
        # per tile), exact fit (the rule of wino4_applicable in csrc/wino4_kernels.hip)
        w = executed_mfma_flops(l, "winograd")
        assert winograd_tile(l) == 4 and l["flops"] / 4 <= w <= l["flops"] / 4 * 1.07
        if l["h"] == 513:      # pooling / fused-1x1 variants skip blocks outside the image (528 rows), the plain variant does not (544)
            assert w == 4.5 * l["cin"] * l["cout"] * (544 if l["name"] == "up4.conv1(cat)" else 528) * 256
        if l["name"].startswith("bottleneck"):
            assert w * 4 == l["flops"]
        assert executed_mfma_flops(l, "winograd", "4") >= l["flops"] / 4
    assert executed_mfma_flops(ls[0], "winograd") == 0.0 and executed_mfma_flops(ls[-1], "direct") == 0.0
    even = unet_launches(512, 256)
    for l in even:
        if l["kind"] == "conv3x3":
            assert executed_mfma_flops(l, "winograd", "2") * 2.25 == l["flops"]
            if winograd_tile(l) == 4:
                assert executed_mfma_flops(l, "winograd") * 4 == l["flops"]
        if l["kind"] == "convt":
            assert executed_mfma_flops(l, "direct") == l["flops"]


def test_bench_roofline_objects_from_synthetic_timings():
    """bench.py's roofline / forward objects (no GPU needed): frac = executed matrix-core FLOPs / peak <= 1 for timings at
    the measured scale, the direct-convolution rate is reported separately, traffic only with a matching library digest."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from audiodenoiser_amd.roofline import PEAK_MFMA_F16_TFLOPS, PEAK_MFMA_F32_TFLOPS, unet_launches
    launches = unet_launches(513, 256)
    ms = np.array([0.4 if l["kind"] == "first" else 0.45 if l["kind"] == "out" else 1.2 if l["kind"] == "convt"
                   else 2.0 * l["flops"] / 9.68e9 for l in launches], dtype=np.float32)
    r = bench.conv_roofline(ms, 64, "winograd", PEAK_MFMA_F32_TFLOPS, "wino4", "wino4_conv_f32")
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and 0.3 < r["frac"] <= 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "17 launches" in r["kernel"] and "other_3x3_kernel" not in r
    assert r["algorithmic"]["tflops"] > r["achieved"] * 3.7            # F(4x4,3x3) executes 1/4 of the direct count
    r2 = bench.conv_roofline(ms, 64, "winograd", PEAK_MFMA_F32_TFLOPS, "wino", "wino_conv_dma_f32", "2")
    assert "17 launches" in r2["kernel"] and "other_3x3_kernel" not in r2 and r2["achieved"] > r["achieved"] * 1.6
    # a small image: the 16x16 tiles of F(2x2,3x3) fit where the 32x32 tiles would overhang
    from audiodenoiser_amd.roofline import winograd_tile
    assert winograd_tile({"kind": "conv3x3", "h": 33, "w": 47}) == 2 and winograd_tile({"kind": "conv3x3", "h": 64, "w": 80}) == 4
    assert winograd_tile({"kind": "conv3x3", "h": 32, "w": 16}) == 4 and winograd_tile({"kind": "conv3x3", "h": 20, "w": 9}) == 2
    assert r["traffic"] is None or isinstance(r["traffic"], int)
    f = bench.forward_summary(ms, 64, "winograd", PEAK_MFMA_F32_TFLOPS)
    assert set(f["per_launch_ms"]) == {l["name"] for l in launches} and f["frac_mfma_peak_executed"] <= 1.0
    r16 = bench.conv_roofline(ms * 0.9, 256, "direct_f16", PEAK_MFMA_F16_TFLOPS, "conv_dma", "conv_mfma_f16")
    assert 0.0 < r16["frac"] <= 1.0 and r16["algorithmic"]["tflops"] <= r16["achieved"] * 1.02
    # a PMC file of another build must not be reported
    t, why = bench.tracked_traffic("no_such_kernel")
    assert t is None and why


def test_profile_summary_demangles_float16_kernel_names():
    import importlib.util
    spec = importlib.util.spec_from_file_location("_pmc", os.path.join(ROOT, "tools", "pmc_summary.py"))
    pmc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pmc)
    n = pmc.short("_ZN3adn12_GLOBAL__N_18conv_dmaIDF16_Li32ELi64ELi8ELi1ELi9ELi1ELi0ELi4EEEvNS_8ConvArgsE")
    assert n == "conv_dma<_Float16, 32, 64, 8, 1, 9, 1, 0, 4>"
    assert pmc.FAMILIES["conv_mfma_f16"](n) and not pmc.FAMILIES["convt_f16"](n)
    assert pmc.FAMILIES["convt_f32"]("conv_dma<float, 8, 128, 2, 2, 1, 4, 2, 2>")
    assert pmc.FAMILIES["stft_wave_kernel"]("stft_wave_kernel<512, 4, 16, 3, false>")
    assert pmc.FAMILIES["stft_wave_kernel_fit"]("stft_wave_kernel<512, 4, 16, 3, true>")
    assert pmc.short("void adn::(anonymous namespace)::wino_conv_dma_f32<0, 4, 0, 0>(adn::ConvArgs)") == "wino_conv_dma_f32<0, 4, 0, 0>"


def test_library_digest_identifies_code_not_comments(tmp_path, monkeypatch):
    """profiles/pmc_traffic.json is tied to a build by the digest of its sources; comments and whitespace do not count."""
    from audiodenoiser_amd import build as B
    src = 'int a = 1; // note\n/* block\n comment */ const char *s = "// kept /* kept */"; char c = \'"\';  // tail \\\ncontinued\nint b = 2;\n'
    assert B._strip_comments(src) == 'int a = 1; const char *s = "// kept /* kept */"; char c = \'"\'; int b = 2; '
    # spacing inside a literal is code, and so is the newline that ends a preprocessor directive
    assert B._strip_comments('f("a  b");') != B._strip_comments('f("a b");')
    assert B._strip_comments("#define X 1\nint y;\n") != B._strip_comments("#define X 1 int y;\n")
    assert B._strip_comments("#define X(a) \\\n    ((a) + 1)\nint y;\n") == "#define X(a) ((a) + 1)\nint y; "
    assert B._strip_comments("int   a ;\n\n  int b;") == B._strip_comments("int a ; int b;")
    csrc = tmp_path / "csrc"
    inc = tmp_path / "include"
    csrc.mkdir()
    inc.mkdir()
    (inc / "adn.h").write_text("int adn_version(void);\n")
    (csrc / "k.hip").write_text("// v1\n__global__ void k(float *p) { p[0] = 1.f; }\n")
    monkeypatch.setattr(B, "CSRC", str(csrc))
    monkeypatch.setattr(B, "INCLUDE", str(inc))
    d0 = B._digest()
    (csrc / "k.hip").write_text("// reworded comment\n\n__global__ void k(float *p)   { p[0] = 1.f; }   /* same code */\n")
    assert B._digest() == d0
    r0 = B._raw_digest()
    (csrc / "k.hip").write_text("// reworded again\n\n__global__ void k(float *p)   { p[0] = 1.f; }   /* same code */\n")
    assert B._digest() == d0 and B._raw_digest() != r0        # any edit rebuilds; only code edits orphan the PMC profile
    (csrc / "k.hip").write_text("__global__ void k(float *p) { p[0] = 2.f; }\n")
    assert B._digest() != d0


def test_library_sources_carry_no_experiment_switches():
    """Source hygiene: the production sources contain no timing-experiment code and read the environment only
    where a U-Net handle is created -- the switches listed in the table of include/adn.h, nothing else."""
    import glob
    csrc = os.path.join(ROOT, "audiodenoiser_amd", "csrc")
    header = open(os.path.join(ROOT, "include", "adn.h")).read()
    documented = set(re.findall(r"\b(ADN_[A-Z0-9_]+)=", header[header.index("environment switches"):]))
    read = {}
    for path in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))):
        text = open(path).read()
        assert "ADN_EXPERIMENTS" not in text, path
        for name in re.findall(r'getenv\("([A-Z0-9_]+)"\)', text):
            read.setdefault(name, set()).add(os.path.basename(path))
    assert set().union(*read.values()) == {"adn_api.hip"}, read
    assert set(read) == documented, sorted(set(read) ^ documented)
