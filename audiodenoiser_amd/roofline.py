"""Algorithmic work of the forward, per kernel launch, in libadn's launch order (include/adn.h, ADN_N_LAUNCHES).

FLOPs: 2*Cin*Cout*k*k*Hout*Wout per convolution (direct-convolution count, no Winograd discount).
Bytes: every op reads its input once and writes its output once; BatchNorm/bias/ReLU are fused, the max-pool
is written by its producer, concat and pad are virtual (192.443 GFLOP and 670.63 MB of
fp32 activation traffic per 513x256 sample, + 124.12 MB of weights once per launch sequence).
"""
from __future__ import annotations

PEAK_MFMA_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: 256 CU x 2.4 GHz x 256 FLOP/clk, v_mfma_f32_32x32x2_f32
PEAK_MFMA_F16_TFLOPS = 2516.6  # dense fp16 MFMA (16x the fp32 rate), MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0          # HBM3E spec
ACHIEVABLE_HBM_GBS = 6300.0    # measured float4 copy rate (MI355X_MICROARCH.md: 6.29 TB/s, 79 % of the spec)


def _ceil_to(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def winograd_tile(launch: dict, mode: str = "auto") -> int:
    """Which Winograd form libadn's fp32 path runs a 3x3 launch with (mirrors ``wino4_applicable`` in
    csrc/wino4_kernels.hip): 4 = F(4x4,3x3) on 32x32-pixel workgroup tiles (two clips side by side for images at most 16
    pixels wide) where they cover the layer with at most a quarter of the tiled area outside the image, else 2 =
    F(2x2,3x3) on 16x16-pixel tiles.  ``mode``: the handle's
    ADN_WINO_TILE setting ("auto", "2" or "4")."""
    if launch["kind"] != "conv3x3" or mode == "2":
        return 2
    if mode == "4":
        return 4
    h, w = launch["h"], launch["w"]
    return 4 if _wino4_tiled_area(h, w) * 3 <= h * w * 4 else 2


def _wino4_tiled_area(h: int, w: int) -> int:
    """Pixels per clip the F(4x4,3x3) workgroup tiles cover: 32x32 tiles, or -- images at most 16 pixels wide, "pair mode":
    a tile holds the same 32 rows of two clips side by side -- 32 rows x 16 columns.  (What ``wino4_applicable`` weighs.)"""
    return _ceil_to(h, 32) * 16 if w <= 16 else _ceil_to(h, 32) * _ceil_to(w, 32)


def _wino4_computed_area(h: int, w: int, epi: str = "pool") -> int:
    """Pixels per clip the F(4x4,3x3) kernel runs MFMAs for.  A workgroup tile is 2x2 blocks of 16x16 pixels.  In the pooling
    and fused-1x1 variants (``epi`` "pool" / "dot": the kernel's LEAN form) a block that lies wholly outside the image does no
    arithmetic (``if (LEAN && !active)`` in csrc/wino4_kernels.hip), so the image is padded to multiples of 16 -- 528 rows for
    H = 513; the plain variant ("plain") has no such branch and computes whole 32x32 tiles -- 544 rows."""
    m = 32 if epi == "plain" else 16
    return _ceil_to(h, m) * 16 if w <= 16 else _ceil_to(h, m) * _ceil_to(w, m)


def executed_mfma_flops(launch: dict, algo: str, wino_mode: str = "auto") -> float:
    """Matrix-core FLOPs one launch EXECUTES per sample, padded tiles counted (what `roofline.frac` is made of).

    ``algo``: "winograd" (fp32 default: wino4_conv_f32 = F(4x4,3x3), 36 multiply-adds per 4x4 output tile and (cin, cout)
    pair = 4.5 FLOP per computed output pixel -- ``_wino4_computed_area``: by epilogue variant the 16x16-pixel blocks that
    touch the image or whole 32x32 tiles --, where
    ``winograd_tile`` says so; wino_conv_dma_f32 =
    F(2x2,3x3), 16 multiply-adds per 2x2 tile = 8 FLOP per padded pixel on 16x16 tiles, elsewhere), "direct"
    (conv_mfma<float>: TH x 16 tiles with TH = 16 for the 64-channel layers and 8 otherwise, 18 FLOP per pixel) or
    "direct_f16" (conv16_f16 / conv_dma<_Float16>: 32 x 16 tiles for every layer).  Transposed convolutions: K = Cin, 4*Cout
    GEMM columns; fp32: conv_dma<float, 8, 128, ...> on 8 x 16 tiles of input pixels, fp16: convt16_f16 on 16 x 16 tiles.  The
    first (Cin = 1) and last (1x1, Cout = 1) layers do not use the matrix cores: 0."""
    kind = launch["kind"]
    if kind in ("first", "out"):
        return 0.0
    cin, cout, h, w = launch["cin"], launch["cout"], launch["h"], launch["w"]
    if kind == "convt":
        return 2.0 * cin * 4 * cout * _ceil_to(h, 16 if algo == "direct_f16" else 8) * _ceil_to(w, 16)
    if algo == "winograd":
        if winograd_tile(launch, wino_mode) == 4:
            return 4.5 * cin * cout * _wino4_computed_area(h, w, launch.get("epi", "plain"))
        return 8.0 * cin * cout * _ceil_to(h, 16) * _ceil_to(w, 16)
    th = 32 if algo == "direct_f16" else (16 if cout == 64 else 8)
    return 18.0 * cin * cout * _ceil_to(h, th) * _ceil_to(w, 16)


def unet_launches(f: int, t: int):
    """List of dicts (name, kind, flops, act_bytes, weight_bytes) per sample, in launch order."""
    ch = (64, 128, 256, 512, 1024)
    hs, ws = [f], [t]
    for _ in range(4):
        hs.append(hs[-1] // 2)
        ws.append(ws[-1] // 2)
    out = []

    def conv(name, kind, cin, cout, h, w, k, pool=False, hout=None, wout=None):
        hout = h if hout is None else hout
        wout = w if wout is None else wout
        flops = 2.0 * cin * cout * k * k * (hout * wout if kind != "convt" else h * w)
        if kind == "convt":
            flops = 2.0 * cin * cout * 4 * h * w
        act = 4.0 * (cin * h * w + cout * hout * wout)
        if pool:
            act += 4.0 * cout * (h // 2) * (w // 2)
        wb = 4.0 * (cin * cout * k * k + cout)
        # epilogue variant of the 3x3 kernels: pooling, plain, or (the network's last 3x3 layer) fused with the 1x1 output conv
        epi = "pool" if pool else "dot" if name == "up4.conv2" else "plain"
        out.append(dict(name=name, kind=kind, flops=flops, act_bytes=act, weight_bytes=wb, cin=cin, cout=cout, h=h, w=w, epi=epi))

    conv("down1.conv1", "first", 1, 64, hs[0], ws[0], 3)
    conv("down1.conv2+pool", "conv3x3", 64, 64, hs[0], ws[0], 3, pool=True)
    for l in range(1, 4):
        conv(f"down{l + 1}.conv1", "conv3x3", ch[l - 1], ch[l], hs[l], ws[l], 3)
        conv(f"down{l + 1}.conv2+pool", "conv3x3", ch[l], ch[l], hs[l], ws[l], 3, pool=True)
    conv("bottleneck.conv1", "conv3x3", 512, 1024, hs[4], ws[4], 3)
    conv("bottleneck.conv2", "conv3x3", 1024, 1024, hs[4], ws[4], 3)
    uh, uw, upc = hs[4], ws[4], 1024
    for i, l in enumerate((3, 2, 1, 0)):
        co = ch[l]
        conv(f"up{i + 1}.convT", "convt", upc, co, uh, uw, 2, hout=2 * uh, wout=2 * uw)
        conv(f"up{i + 1}.conv1(cat)", "conv3x3", 2 * co, co, hs[l], ws[l], 3)
        # the concat input is read as skip (co channels at hs[l] x ws[l]) + upsampled (co channels at 2uh x 2uw)
        out[-1]["act_bytes"] = 4.0 * (co * hs[l] * ws[l] + co * 4 * uh * uw + co * hs[l] * ws[l])
        conv(f"up{i + 1}.conv2", "conv3x3", co, co, hs[l], ws[l], 3)
        uh, uw, upc = hs[l], ws[l], co
    conv("out.conv1x1", "out", 64, 1, hs[0], ws[0], 1)
    assert len(out) == 23
    return out


def totals(f: int, t: int):
    ls = unet_launches(f, t)
    return (sum(l["flops"] for l in ls), sum(l["act_bytes"] for l in ls), sum(l["weight_bytes"] for l in ls))
