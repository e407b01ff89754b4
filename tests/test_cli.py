"""`ltx-video` CLI (C++ host mirror over the C ABI): flag surface, defaults, dry-run and validation messages of the
reference CLI (LTXVideoCLI.swift:21-209) on CPU; one end-to-end generate on the GPU."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "ltx-video-swift-mlx_amd", "csrc", "build", "ltx-video")


def run(*args):
    p = subprocess.run([CLI, *args], capture_output=True, text=True, timeout=600)
    return p.returncode, p.stdout, p.stderr


def test_cli_info_and_version(ltx):
    rc, out, _ = run()
    assert rc == 0 and "0.1.0" in out  # default subcommand = info
    rc, out, _ = run("--version")
    assert rc == 0 and out.strip() == "0.1.0"


def test_cli_dry_run_defaults(ltx):
    rc, out, _ = run("generate", "a beaver building a dam", "--dry-run")
    assert rc == 0
    # defaults of the reference CLI: 512x512, 25 frames, distilled, output.mp4 (LTXVideoCLI.swift:29-50)
    assert "Resolution: 512x512" in out and "Frames: 25" in out and "Model: distilled" in out and "Output: output.mp4" in out
    assert out.strip().endswith("Validation passed (dry run mode)")
    rc, out, _ = run("generate", "x", "-w", "768", "-h", "512", "-f", "25", "--two-stage", "--distilled-lora", "--dry-run",
                     "--transformer-quant", "qint8", "--seed", "42")
    assert rc == 0 and "Model: dev" in out and "Two-stage pipeline: 384x256 -> upscale 2x -> 768x512" in out
    assert "Distilled LoRA: will fuse into dev model (8 steps, no CFG)" in out and "Seed: 42" in out


@pytest.mark.parametrize("args,msg", [
    (["-f", "24"], "Frame count must be 8n+1 (e.g., 9, 17, 25, 33, ...). Got 24"),
    (["-w", "500"], "Width and height must be divisible by 32. Got 500x512"),
    (["--transformer-quant", "fp8"], "Invalid transformer quantization: fp8. Use: bf16, qint8, or int4"),
    (["-m", "turbo"], "Invalid model: turbo. Use: distilled or dev"),
    (["-w", "800", "--two-stage"], "Two-stage requires width and height divisible by 64. Got 800x512"),
    (["--noise-rng", "philox"], "Invalid noise generator: philox. Use: native or mlx"),
])
def test_cli_validation_messages(ltx, args, msg):
    rc, _, err = run("generate", "x", "--dry-run", *args)
    assert rc != 0 and msg in err


def test_cli_lists_and_validates_library_options(ltx):
    """Round 5: the library's launcher switches are reachable from the compiled host only through the ABI (`ltx-video options` prints
    ltx_option_info's table; `--hip-option name=value` calls ltx_ctx_set_option before anything else) - no GPU needed for either."""
    rc, out, _ = run("options")
    assert rc == 0 and "ABI revision 2" in out
    names = [ln.split("=")[0].replace("*", "").strip() for ln in out.splitlines()[1:]]
    assert names == [t[0] for t in ltx.option_table()] and "* qk_f32 = 0" in out and "  finish_norm = 1" in out
    rc, out, _ = run("generate", "x", "--dry-run", "--hip-option", "qk_f32=1", "--hip-option", "conv_tall=0")
    assert rc == 0 and out.strip().endswith("Validation passed (dry run mode)")
    for bad, msg in (("no_such=1", "unknown option"), ("finish_rows=9", "outside"), ("qk_f32", "expects name=value")):
        rc, _, err = run("generate", "x", "--dry-run", "--hip-option", bad)
        assert rc == 64 and msg in err, (bad, err)


@pytest.mark.gpu
def test_cli_generate_end_to_end(ltx, oracle, tmp_path):
    """generate with a reduced-depth DiT, statistics-only VAE file and explicit embeddings: plumbing + file contract."""
    from safetensors.numpy import save_file

    from test_dit_gpu import write_dit_file

    ocfg = oracle.DiTConfig(num_layers=2, num_heads=2, caption_channels=128)
    write_dit_file(oracle, oracle.synth_dit_weights(ocfg, seed=1), tmp_path / "dit.safetensors")
    save_file({"latents_mean": np.zeros(128, np.float32), "latents_std": np.ones(128, np.float32)}, str(tmp_path / "vae.safetensors"))
    rng = np.random.default_rng(0)
    save_file({"prompt_embeddings": rng.standard_normal((1, 24, 128)).astype(np.float32),
               "prompt_mask": np.ones((1, 24), np.int32)}, str(tmp_path / "emb.safetensors"))
    out = tmp_path / "out.raw"
    rc, so, se = run("generate", "test", "-w", "64", "-h", "64", "-f", "9", "--seed", "7", "-o", str(out),
                     "--ltx-weights", str(tmp_path / "dit.safetensors"), "--vae-weights", str(tmp_path / "vae.safetensors"),
                     "--embeddings", str(tmp_path / "emb.safetensors"), "--num-layers", "2", "--num-heads", "2",
                     "--caption-channels", "128", "--profile", "--png-dir", str(tmp_path))
    assert rc == 0, se
    assert "Step 8/8" in so and "Generated 9 frames (64x64)" in so
    # --profile: the reference's per-step diagnostics line (LTXPipeline.swift:951), one per step, sigma schedule = the distilled table
    import re
    diag = re.findall(r"  Step (\d): σ=([0-9.]+)→([0-9.]+), vel mean=(-?[0-9.]+), std=([0-9.]+), latent mean=(-?[0-9.]+), std=([0-9.]+)", so)
    assert [int(d[0]) for d in diag] == list(range(8)), so
    assert diag[0][1] == "1.0000" and diag[7][2] == "0.0000" and all(float(d[4]) > 0 and float(d[6]) > 0 for d in diag)
    meta = json.loads(open(str(out) + ".json").read())
    assert meta == {"frames": 9, "height": 64, "width": 64, "channels": 3, "dtype": "float32", "range": [0, 1], "seed": 7}
    frames = np.fromfile(out, np.float32).reshape(9, 64, 64, 3)
    assert np.isfinite(frames).all() and frames.min() >= 0 and frames.max() <= 1
    # --png-dir: one PNG per frame holding uint8(clip(x,0,1)*255) (VideoExporter.swift:563-580)
    import struct
    import zlib
    raw = (tmp_path / "frame_0003.png").read_bytes()
    pos, idat = 8, b""
    while pos < len(raw):
        n, typ = struct.unpack(">I4s", raw[pos:pos + 8])
        if typ == b"IDAT":
            idat += raw[pos + 8:pos + 8 + n]
        pos += 12 + n
    px = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(64, 64 * 3 + 1)[:, 1:].reshape(64, 64, 3)
    assert np.array_equal(px, (frames[3] * np.float32(255)).astype(np.uint8))
    assert (tmp_path / "frame_0008.png").exists() and not (tmp_path / "frame_0009.png").exists()


@pytest.mark.gpu
def test_cli_bench_runs_without_python_on_the_data_path():
    """`ltx-video bench`: the headline loop driven by the compiled host alone - it links libltxhip.so only (no Python, no PyTorch) and
    uses the C ABI with host pointers. Two layers and a small clip here; the full-size line is quoted in DESIGN.md section 5."""
    rc, so, se = run("bench", "-w", "256", "-h", "256", "-f", "9", "--steps", "3", "--warmup", "1", "--text-keys", "64", "--num-layers", "2",
                     "--decodes", "1")
    assert rc == 0, se
    line = json.loads(so.strip().splitlines()[-1])
    assert line["unit"] == "steps/s" and line["value"] > 0 and line["tokens"] == 128 and line["layers"] == 2 and line["pcie_inclusive"] is True
    assert line["vae_decode_ms_host_pointers"] > 0
    import subprocess
    ldd = subprocess.run(["ldd", CLI], capture_output=True, text=True).stdout
    assert "libltxhip" in ldd and "torch" not in ldd and "python" not in ldd.lower()


@pytest.mark.gpu
def test_cli_connector_and_image_inputs_are_validated(ltx, oracle, tmp_path):
    """--gemma-hidden-states / --image-tensor (text-embedding connector and image-to-video inputs of this build): shape contracts
    are checked before any model work; the full-size models behind them are covered by test_connector_gpu / test_vae_encoder_gpu."""
    from safetensors.numpy import save_file

    from test_dit_gpu import write_dit_file

    ocfg = oracle.DiTConfig(num_layers=1, num_heads=2, caption_channels=128)
    write_dit_file(oracle, oracle.synth_dit_weights(ocfg, seed=1), tmp_path / "dit.safetensors")
    save_file({"latents_mean": np.zeros(128, np.float32), "latents_std": np.ones(128, np.float32)}, str(tmp_path / "vae.safetensors"))
    save_file({"prompt_embeddings": np.zeros((1, 8, 128), np.float32)}, str(tmp_path / "emb.safetensors"))
    save_file({"pixels": np.zeros((1, 3, 1, 32, 32), np.float32)}, str(tmp_path / "img.safetensors"))
    base = ["generate", "x", "-w", "64", "-h", "64", "-f", "9", "-o", str(tmp_path / "o.raw"), "--ltx-weights", str(tmp_path / "dit.safetensors"),
            "--vae-weights", str(tmp_path / "vae.safetensors"), "--num-layers", "1", "--num-heads", "2", "--caption-channels", "128"]
    rc, _, err = run(*base, "--embeddings", str(tmp_path / "emb.safetensors"), "--image-tensor", str(tmp_path / "img.safetensors"))
    assert rc != 0 and "pixels must be [1][3][1][height][width]" in err
    rc, _, err = run(*base)  # neither embeddings nor hidden states
    assert rc != 0 and "--embeddings or --gemma-hidden-states" in err
    rc, _, err = run(*base, "--embeddings", str(tmp_path / "emb.safetensors"), "--image", "cat.png")
    assert rc != 0 and "--image-tensor" in err
