"""VAE encoder: HIP path vs the oracle restatement of VideoEncoder.swift, thin channel ladder
(base 64 -> 64..1024 channels; the reference's is 128..2048) so the numpy oracle finishes in seconds.

Tolerance: 30 bf16 x bf16 convolutions in a row with an f32 residual stream: rel-L2 <= 3e-2, cosine >= 0.999. The index
logic (patchify channel order, space-to-depth channel order and front padding of odd frame counts, group-mean residual) is pinned
separately by integer-valued fixtures through the same kernels (exact)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BASE = 64


@pytest.fixture(scope="module")
def enc(ltx, oracle, gpu_ctx, tmp_path_factory):
    from safetensors.torch import save_file

    w = oracle.synth_vae_encoder_weights(base=BASE, seed=8)
    d = tmp_path_factory.mktemp("vaeenc")
    path = d / "vae.safetensors"
    tensors = {k: torch.from_numpy(np.ascontiguousarray(v)).to(torch.bfloat16 if v.ndim == 5 else torch.float32)
               for k, v in oracle.vae_encoder_file_keys(w).items()}
    tensors["decoder.conv_in.conv.bias"] = torch.zeros(4)  # the VAE file also carries the decoder
    save_file(tensors, str(path))
    gpu_ctx.vae_encoder_load(path, BASE)
    rep = gpu_ctx.load_report()
    assert rep["missing"] == 0 and rep["unmatched"] == 0 and rep["loaded"] == len(w), rep
    yield w
    gpu_ctx.vae_encoder_unload()


def _rel(a, b):
    return np.linalg.norm(a - b) / max(1e-9, np.linalg.norm(b))


@pytest.mark.parametrize("T,H,W", [(1, 64, 96), (1, 32, 32), (9, 64, 64), (5, 32, 64)])
def test_vae_encode_parity(ltx, oracle, gpu_ctx, enc, T, H, W):
    w = enc
    rng = np.random.default_rng(T * 1000 + H + W)
    px = rng.uniform(-1, 1, (1, 3, T, H, W)).astype(np.float32)
    got = gpu_ctx.vae_encode(px)
    ref = oracle.vae_encode(w, px, base=BASE)
    assert got.shape == ref.shape == (1, 128, ltx.vae_encoder_latent_frames(T), H // 32, W // 32)
    cos = float((got * ref).sum() / (np.linalg.norm(got) * np.linalg.norm(ref)))
    assert _rel(got, ref) <= 3e-2 and cos >= 0.999, (_rel(got, ref), cos)


def test_vae_encode_normalised_latent_feeds_i2v(ltx, oracle, gpu_ctx, enc):
    """encodeImage (LTXPipeline.swift:1902-1932): (latent - mean_of_means) / std_of_means with the DECODER's statistics; the result
    has the [1,128,1,H/32,W/32] shape ltx_denoise_options.cond_latent takes."""
    w = enc
    rng = np.random.default_rng(3)
    px = rng.uniform(-1, 1, (1, 3, 1, 64, 64)).astype(np.float32)
    with pytest.raises(ltx.LTXError):  # statistics come from the decoder
        gpu_ctx.vae_unload()
        gpu_ctx.vae_encode(px, normalize=True)
    gpu_ctx.vae_init_synthetic(seed=77)
    raw = gpu_ctx.vae_encode(px)
    nrm = gpu_ctx.vae_encode(px, normalize=True)
    assert nrm.shape == (1, 128, 1, 2, 2)
    # synthetic decoder statistics are mean 0 / std 1 -> identical; a loaded file exercises the arithmetic in
    # test_vae_gpu's decoder tests (same buffers)
    assert np.allclose(nrm, raw)
    gpu_ctx.vae_unload()


def test_encoder_index_logic_exact(ltx, oracle, gpu_ctx):
    """Patchify / space-to-depth / group-mean residual order on integer data through an encoder whose convolutions are zero:
    with all conv weights and biases 0, every resblock is the identity and each downsampler returns exactly the group mean of
    space_to_depth(x) - so the output latent is a pure function of the index permutations."""
    from safetensors.torch import save_file
    import tempfile, os

    shapes = oracle.vae_encoder_param_shapes(base=BASE)
    w = {k: np.zeros(s, np.float32) for k, s in shapes.items()}
    # conv_in: identity-like tap so that the patchified pixels reach the stream: out channel o reads input channel o % 48, centre
    # tap of the CURRENT frame (causal kernel: temporal index 2), weight 1
    for o in range(BASE):
        w["conv_in.conv.weight"][o, o % 48, 2, 1, 1] = 1.0
    # conv_out: pass-through of the first 128 stream channels (centre tap)
    for o in range(128):
        w["conv_out.conv.weight"][o, o, 2, 1, 1] = 1.0
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "z.safetensors")
        save_file({k: torch.from_numpy(v).to(torch.bfloat16 if v.ndim == 5 else torch.float32)
                   for k, v in oracle.vae_encoder_file_keys(w).items()}, path)
        gpu_ctx.vae_encoder_load(path, BASE)
    rng = np.random.default_rng(1)
    px = rng.integers(-4, 5, (1, 3, 3, 32, 64)).astype(np.float32)
    got = gpu_ctx.vae_encode(px)
    ref = oracle.vae_encode(w, px, base=BASE)
    gpu_ctx.vae_encoder_unload()
    # the only inexact ops left are pixel-norm + SiLU before conv_out (bf16 conv input): compare at bf16 resolution
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 2.0 ** -7 * max(1.0, np.abs(ref).max()), np.abs(got - ref).max()
    assert _rel(got, ref) <= 4e-3


def test_vae_encode_vs_golden_fixture(ltx, oracle, gpu_ctx, tmp_path):
    """HIP path vs tests/golden/vae_encoder_tiny.npz (oracle output cross-checked against an independent torch implementation)."""
    import os
    from safetensors.torch import save_file

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vae_encoder_tiny.npz"))
    base = int(g["base"])
    w = oracle.synth_vae_encoder_weights(base=base, seed=int(g["seed"]))
    path = tmp_path / "e.safetensors"
    save_file({k: torch.from_numpy(np.ascontiguousarray(v)).to(torch.bfloat16 if v.ndim == 5 else torch.float32)
               for k, v in oracle.vae_encoder_file_keys(w).items()}, str(path))
    gpu_ctx.vae_encoder_load(path, base)
    got = gpu_ctx.vae_encode(g["pixels"])
    gpu_ctx.vae_encoder_unload()
    assert got.shape == g["latent"].shape and _rel(got, g["latent"]) <= 3e-2
