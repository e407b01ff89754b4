"""VAE decoder parity: implicit-GEMM conv3d kernel and the full decoder (libltxhip.so) vs the oracle.

Tolerance: the HIP path feeds bf16 activations x bf16 weights to the MFMA (f32 accumulate, f32 residual stream);
the oracle keeps f32 activations. Single conv on the same bf16-rounded inputs: accumulation order only (exact on
integer data). Whole decoder (42 convs): rel-L2 <= 3e-2 on the raw output, <= 2e-2 abs on [0,1] frames.
"""
import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def relayout(w):
    """(O,I,kT,kH,kW) -> [O][27][I] (the ABI's conv weight layout)."""
    o, i = w.shape[:2]
    return np.ascontiguousarray(w.reshape(o, i, 27).transpose(0, 2, 1))


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("F,H,W,Cin,Cout", [(3, 5, 6, 64, 96), (1, 2, 2, 128, 48), (4, 7, 3, 192, 132)])
def test_conv3d_integer_exact(gpu_ctx, oracle, F, H, W, Cin, Cout, causal):
    rng = np.random.default_rng(F * 100 + H * 10 + W + Cin)
    x = rng.integers(-2, 3, (1, Cin, F, H, W)).astype(np.float32)
    w = rng.integers(-2, 3, (Cout, Cin, 3, 3, 3)).astype(np.float32)
    b = rng.integers(-4, 5, (Cout,)).astype(np.float32)
    ref = oracle.conv3d_full(x, w, b, causal=causal)[0].transpose(1, 2, 3, 0)  # (F,H,W,Cout)
    xd = torch.from_numpy(np.ascontiguousarray(x[0].transpose(1, 2, 3, 0))).to(torch.bfloat16).cuda()
    wd = torch.from_numpy(relayout(w)).to(torch.bfloat16).cuda()
    bd = torch.from_numpy(b).cuda()
    out = torch.empty((F, H, W, Cout), device="cuda")
    gpu_ctx.op_conv3d(xd, wd, bd, out, causal=causal)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), ref)


@pytest.mark.parametrize("F,H,W,Cin,Cout", [(6, 96, 192, 64, 128), (6, 96, 96, 64, 256), (6, 95, 190, 64, 128)])  # the last: 565 row tiles, ragged
def test_conv3d_split_tail_integer_exact(gpu_ctx, oracle, F, H, W, Cin, Cout):
    """576 (565) output tiles = two full rounds of 256 + 64 (53): the launcher runs the remainder as a split-K launch over a tile window
    (csrc/gemm.hip, conv branch of launch_gemm_bf16). Small integers: every summation order gives the same f32."""
    rng = np.random.default_rng(F + H + W + Cout)
    x = rng.integers(-2, 3, (1, Cin, F, H, W)).astype(np.float32)
    w = rng.integers(-2, 3, (Cout, Cin, 3, 3, 3)).astype(np.float32)
    b = rng.integers(-4, 5, (Cout,)).astype(np.float32)
    ref = oracle.conv3d_full(x, w, b, causal=False)[0].transpose(1, 2, 3, 0)
    xd = torch.from_numpy(np.ascontiguousarray(x[0].transpose(1, 2, 3, 0))).to(torch.bfloat16).cuda()
    wd = torch.from_numpy(relayout(w)).to(torch.bfloat16).cuda()
    out = torch.full((F, H, W, Cout), float("nan"), device="cuda")
    gpu_ctx.op_conv3d(xd, wd, torch.from_numpy(b).cuda(), out)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), ref)


def test_conv3d_random_shapes_integer_exact(gpu_ctx):
    """Seeded random geometries through the conv launcher's three routes (one launch; whole rounds + split-K tail window; narrow
    outputs) against torch's conv3d on the same integer data - exact in f32 whatever the summation order. Ragged tile counts,
    image rows that straddle tiles, single-frame inputs."""
    import torch.nn.functional as F_

    rng = np.random.default_rng(20260)
    cases = [(1, 2, 2), (1, 31, 45), (3, 17, 23), (2, 64, 96), (5, 95, 101), (7, 97, 193), (4, 128, 192), (9, 64, 191)]
    for F, H, W in cases:
        Cin = int(rng.choice([64, 128]))
        Cout = int(rng.choice([48, 128, 256]))
        if F * H * W * Cin * Cout * 27 > 3.5e12:  # keep the torch reference to a few seconds
            Cin, Cout = 64, 128
        x = torch.from_numpy(rng.integers(-2, 3, (1, Cin, F, H, W)).astype(np.float32)).cuda()
        w = torch.from_numpy(rng.integers(-2, 3, (Cout, Cin, 3, 3, 3)).astype(np.float32)).cuda()
        b = torch.from_numpy(rng.integers(-4, 5, (Cout,)).astype(np.float32)).cuda()
        xp = F_.pad(x, (1, 1, 1, 1, 0, 0), mode="reflect")
        xp = torch.cat([xp[:, :, :1], xp, xp[:, :, -1:]], 2)  # replicate in T
        ref = F_.conv3d(xp.double(), w.double(), b.double())[0].permute(1, 2, 3, 0).float()
        xd = x[0].permute(1, 2, 3, 0).contiguous().to(torch.bfloat16)
        wd = torch.from_numpy(relayout(w.cpu().numpy())).to(torch.bfloat16).cuda()
        out = torch.full((F, H, W, Cout), float("nan"), device="cuda")
        gpu_ctx.op_conv3d(xd, wd, b, out)
        torch.cuda.synchronize()
        assert torch.equal(out, ref), (F, H, W, Cin, Cout, float((out - ref).abs().max()))


@pytest.mark.parametrize("F,H,W,Cin,Cout", [(9, 64, 96, 64, 128), (11, 100, 48, 64, 128), (5, 64, 96, 64, 256), (9, 121, 48, 64, 96), (2, 140, 192, 128, 128)])
def test_conv3d_halo_persistent_launches_integer_exact(gpu_ctx, F, H, W, Cin, Cout):
    """Round 4: above one round of tiles the halo-staged kernel is persistent (one workgroup per CU walks its XCD's chunk of the tile order
    and requests the next tile's operands from inside the epilogue). Tile counts just above the CU count, with chunk remainders: 288 tiles
    in the blocked order, 275 in the plain order (25 tiles per frame), 320 over two column tiles (supertile order), 273 with a ragged last
    tile and 96 of 128 columns, 280 whole-row tiles at two channel halves - against torch's conv3d on integer data, exactly."""
    import torch.nn.functional as F_

    rng = np.random.default_rng(F * 1000 + H * 100 + W + Cin + Cout)
    x = torch.from_numpy(rng.integers(-2, 3, (1, Cin, F, H, W)).astype(np.float32)).cuda()
    w = torch.from_numpy(rng.integers(-2, 3, (Cout, Cin, 3, 3, 3)).astype(np.float32)).cuda()
    b = torch.from_numpy(rng.integers(-4, 5, (Cout,)).astype(np.float32)).cuda()
    xp = F_.pad(x, (1, 1, 1, 1, 0, 0), mode="reflect")
    xp = torch.cat([xp[:, :, :1], xp, xp[:, :, -1:]], 2)
    ref = F_.conv3d(xp.double(), w.double(), b.double())[0].permute(1, 2, 3, 0).float()
    xd = x[0].permute(1, 2, 3, 0).contiguous().to(torch.bfloat16)
    wd = torch.from_numpy(relayout(w.cpu().numpy())).to(torch.bfloat16).cuda()
    assert ((F * H * W + 191) // 192) * ((Cout + 127) // 128) > 256
    for _ in range(2):  # twice: the second launch finds the first one's lines in the caches
        out = torch.full((F, H, W, Cout), float("nan"), device="cuda")
        gpu_ctx.op_conv3d(xd, wd, b, out)
        torch.cuda.synchronize()
        assert torch.equal(out, ref), (F, H, W, Cin, Cout, float((out - ref).abs().max()))


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("F,H,W,Cin,Cout", [(3, 7, 48, 64, 128), (2, 5, 64, 128, 96), (3, 6, 96, 192, 128), (2, 3, 192, 64, 256),
                                            (1, 4, 384, 128, 128), (5, 4, 48, 64, 128), (1, 2, 192, 64, 128), (2, 2, 576, 64, 80)])
def test_conv3d_halo_staged_geometries_integer_exact(gpu_ctx, F, H, W, Cin, Cout, causal):
    """The geometries the halo-staged conv kernel takes (csrc/conv_halo.inc: an image row is a whole number of 192-position tiles or a
    tile a whole number of image rows, W >= 48): segments of 48 / 64 / 96 positions, whole and half rows of 192 / 384 / 576, a last tile
    that ends before its 192 rows (7 x 48, 5 x 64 positions per frame), fewer than 128 output channels, one- and two-half channel
    counts, causal and replicated time padding - against torch's conv3d on integer data (exact whatever the order of the sums).
    VideoConvolution.swift:202-348 is the reference's conv (three conv2d summed)."""
    import torch.nn.functional as F_

    rng = np.random.default_rng(F * 1000 + H * 100 + W + Cin + Cout + causal)
    x = torch.from_numpy(rng.integers(-2, 3, (1, Cin, F, H, W)).astype(np.float32)).cuda()
    w = torch.from_numpy(rng.integers(-2, 3, (Cout, Cin, 3, 3, 3)).astype(np.float32)).cuda()
    b = torch.from_numpy(rng.integers(-4, 5, (Cout,)).astype(np.float32)).cuda()
    xp = F_.pad(x, (1, 1, 1, 1, 0, 0), mode="reflect")
    xp = torch.cat([xp[:, :, :1], xp[:, :, :1], xp], 2) if causal else torch.cat([xp[:, :, :1], xp, xp[:, :, -1:]], 2)
    ref = F_.conv3d(xp.double(), w.double(), b.double())[0].permute(1, 2, 3, 0).float()
    xd = x[0].permute(1, 2, 3, 0).contiguous().to(torch.bfloat16)
    wd = torch.from_numpy(relayout(w.cpu().numpy())).to(torch.bfloat16).cuda()
    out = torch.full((F, H, W, Cout), float("nan"), device="cuda")
    gpu_ctx.op_conv3d(xd, wd, b, out, causal=causal)
    torch.cuda.synchronize()
    assert torch.equal(out, ref), float((out - ref).abs().max())


@pytest.fixture(scope="module")
def vae_model(ltx, oracle, gpu_ctx, tmp_path_factory):
    from safetensors.torch import save_file

    w = oracle.synth_vae_weights(seed=5, timestep_conditioning=True)
    d = tmp_path_factory.mktemp("vae")
    path = d / "diffusion_pytorch_model.safetensors"
    save_file({k: torch.from_numpy(np.ascontiguousarray(v)).to(torch.bfloat16 if v.ndim == 5 else torch.float32)
               for k, v in oracle.vae_file_keys(w).items()}, str(path))
    (d / "config.json").write_text(json.dumps({"timestep_conditioning": False}))
    gpu_ctx.vae_load(path)
    rep = gpu_ctx.load_report()
    assert rep["unmatched"] == 0 and rep["missing"] == 0, rep
    assert gpu_ctx.vae_timestep_conditioning is False
    return w, d


def _cmp(got, ref_raw):
    ref = np.clip((ref_raw + 1) / 2, 0, 1)
    assert got.shape == ref.shape
    err = np.abs(got - ref).max()
    rel = np.linalg.norm(got - ref) / max(1e-9, np.linalg.norm(ref - ref.mean()))
    return err, rel


def test_vae_decode_parity(ltx, oracle, gpu_ctx, vae_model):
    w, _ = vae_model
    rng = np.random.default_rng(1)
    lat = rng.standard_normal((1, 128, 2, 2, 3)).astype(np.float32)
    got = gpu_ctx.vae_decode(lat)
    assert got.shape == (9, 64, 96, 3)
    raw = oracle.decode_video(w, lat, return_raw=True)
    err, rel = _cmp(got, raw)
    assert err <= 2e-2 and rel <= 3e-2, (err, rel)


def test_vae_decode_tiled_parity(ltx, oracle, gpu_ctx, vae_model):
    """Temporal tiling: tile walk, per-tile decode, linear blend over 8*overlap frames, frame count."""
    w, _ = vae_model
    rng = np.random.default_rng(2)
    lat = rng.standard_normal((1, 128, 4, 2, 2)).astype(np.float32)
    tiles, nf = ltx.vae_tile_plan(4, 2, 1)
    assert tiles == [(0, 2), (1, 3), (2, 4)] and nf == 9 + 9 + 9 - 16
    got = gpu_ctx.vae_decode(lat, tile=2, overlap=1)
    assert got.shape == (nf, 64, 64, 3)
    raw = oracle.decode_video(w, lat, tile=2, overlap=1, return_raw=True)
    err, rel = _cmp(got, raw)
    assert err <= 2e-2 and rel <= 3e-2, (err, rel)


def test_vae_decode_timestep_conditioned(ltx, oracle, gpu_ctx, vae_model):
    w, d = vae_model
    (d / "config.json").write_text(json.dumps({"timestep_conditioning": True}))
    gpu_ctx.vae_load(d / "diffusion_pytorch_model.safetensors")
    assert gpu_ctx.vae_timestep_conditioning is True
    rng = np.random.default_rng(3)
    lat = rng.standard_normal((1, 128, 2, 2, 2)).astype(np.float32)
    noise = rng.standard_normal(lat.shape).astype(np.float32)
    got = gpu_ctx.vae_decode(lat, timestep=0.05, noise=noise)
    raw = oracle.decode_video(w, lat, timestep=0.05, noise=noise, return_raw=True)
    err, rel = _cmp(got, raw)
    assert err <= 2e-2 and rel <= 3e-2, (err, rel)
    with pytest.raises(ltx.LTXError):  # noise is an explicit input
        gpu_ctx.vae_decode(lat, timestep=0.05, noise=None)


def test_conv3d_vs_golden_fixture(gpu_ctx):
    import os

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "conv3d_small.npz"))
    x, w, b, y = g["x"], g["w"], g["b"], g["y"]
    xd = torch.from_numpy(np.ascontiguousarray(x[0].transpose(1, 2, 3, 0))).to(torch.bfloat16).cuda()
    wd = torch.from_numpy(relayout(w)).to(torch.bfloat16).cuda()
    out = torch.empty(xd.shape[:3] + (w.shape[0],), device="cuda")
    gpu_ctx.op_conv3d(xd, wd, torch.from_numpy(b).cuda(), out)
    torch.cuda.synchronize()
    ref = y[0].transpose(1, 2, 3, 0)
    assert np.abs(out.cpu().numpy() - ref).max() <= 1e-4  # inputs are bf16-exact: accumulation order only


@pytest.mark.parametrize("F,H,W,Cin,Cout,causal", [(1, 2, 192, 128, 128, False), (2, 4, 192, 64, 128, True), (3, 6, 192, 128, 128, False),
                                                    (5, 130, 192, 128, 128, True), (2, 258, 192, 192, 128, False), (2, 6, 192, 128, 256, False),
                                                    (1, 4, 96, 128, 128, False), (2, 8, 96, 256, 256, True), (3, 132, 96, 128, 384, False),
                                                    (2, 516, 96, 128, 256, True), (2, 5, 384, 128, 128, False), (3, 131, 384, 64, 256, True)])
def test_tall_conv_tile_integer_exact(ltx, gpu_ctx, F, H, W, Cin, Cout, causal):
    """conv_halo2.inc (round 5): convs with W == 192 (W == 96) run as 384 x 128 tiles of TWO (FOUR) image rows - four (six) staged rows
    per (frame tap, channel half) serve the tile's (row, dy) pairs, the K loop runs on across a workgroup's tiles, row slots rotate from
    group to group (W == 96); with W == 384 the tile is ONE image row in two alternating slots (seven row pieces per wave, an odd H). Small integers: any summation order gives the same f32, so equality with torch's conv3d is exact. Shapes:
    one tile (every row its own reflection), reflect at the top AND bottom of a frame inside one tile walk, causal and non-causal frame
    clamps, 64 / 128 / 192 / 256 channels (1 - 4 halves; W == 96 needs an even number), several column tiles (the weights change from
    tile to tile while the K loop runs on), 325 tall tiles (persistent walk with uneven XCD chunks), 258 row tiles and 2 x 516 / 4 x 2 column
    tiles = 516: whole-round launches + a 192-row tail window through conv_halo.inc. The same launch with option conv_tall = 0 (the
    192-row kernel) must give the same integers."""
    import torch.nn.functional as F_

    g = torch.Generator(device="cuda").manual_seed(F * 1000 + H)
    x = torch.randint(-2, 3, (1, Cin, F, H, W), generator=g, device="cuda", dtype=torch.int8).float()
    w = torch.randint(-2, 3, (Cout, Cin, 3, 3, 3), generator=g, device="cuda", dtype=torch.int8).float()
    b = torch.randint(-4, 5, (Cout,), generator=g, device="cuda", dtype=torch.int8).float()
    xd = x[0].permute(1, 2, 3, 0).contiguous().to(torch.bfloat16)
    wd = torch.from_numpy(relayout(w.cpu().numpy())).to(torch.bfloat16).cuda()
    out = torch.full((F, H, W, Cout), float("nan"), device="cuda")
    with ltx.options(conv_tall=3):   # 3: the tall kernel for every qualifying shape (by default only launches of more than half a round take it)
        gpu_ctx.op_conv3d(xd, wd, b, out, causal=causal)
        torch.cuda.synchronize()
    xp = F_.pad(x, (1, 1, 1, 1, 0, 0), mode="reflect")
    xp = torch.cat([xp[:, :, :1], xp[:, :, :1], xp], 2) if causal else torch.cat([xp[:, :, :1], xp, xp[:, :, -1:]], 2)
    ref = F_.conv3d(xp.double(), w.double(), b.double())[0].permute(1, 2, 3, 0).float()
    assert torch.equal(out, ref), float((out - ref).abs().max())
    auto = torch.full_like(out, float("nan"))
    gpu_ctx.op_conv3d(xd, wd, b, auto, causal=causal)   # the launcher's own choice
    torch.cuda.synchronize()
    assert torch.equal(auto, ref)
    old = torch.full_like(out, float("nan"))
    with ltx.options(conv_tall=0):
        gpu_ctx.op_conv3d(xd, wd, b, old, causal=causal)
        torch.cuda.synchronize()
    assert torch.equal(old, ref)
