#!/usr/bin/env python3
"""Pinning kit: a seed-defined generation case that the REFERENCE can run on a Mac and this repo can run here,
compared on the numbers the reference already prints.

The reference ships no golden vectors and cannot run in this image, so parity here is pinned only by this repo's own oracle
("parity unpinned"). What CAN be done is to make the comparison a one-command job for whoever has the reference
running: with `--profile` its denoise loop logs, per step,
    "  Step i: σ=a→b, vel mean=…, std=…, latent mean=…, std=…"                       (LTXPipeline.swift:945-951, four decimals)
and `ltx-video generate … --profile` of this repo prints the same line from `ltx_denoise_options.step_stats`. This script writes the
inputs both sides need, generated with numpy only (no GPU, no MLX, no torch: it also runs on the Mac), and the lines the oracle expects:

  <out>/ltx_transformer.safetensors   synthetic transformer in the unified checkpoint's key naming, stored as bf16 (`--layers 48` is the
                                      reference architecture: 26 GB and ~10 min of numpy; fewer layers only for a reference build patched to
                                      accept them - this repo's CLI takes --num-layers / --num-heads / --caption-channels)
  <out>/case.safetensors              prompt_embeddings [1,S,C] (bf16-representable f32), prompt_mask [1,S] int32, noise [1,128,F',H',W'] f32
                                      N(0,1), sigmas [9] f32 (the distilled table)
  <out>/expected_oracle.txt           the per-step lines as oracle.denoise gives them (numpy: f32 activations x bf16-valued weights) and the
                                      final latent's mean / std; `expected_oracle.json` holds the same numbers unrounded
  <out>/expected_hip.txt / .json      (--hip, on an MI355X) the same from libltxhip.so

Reference side (swift/PinningHarness/PinParityTests.swift, source-only): load the tensors, build `PrecomputedEmbeddings`, pass the noise
(one optional parameter added to `generateVideo`: the harness file says where), run with `profile: true`, diff the log against
expected_oracle.txt. This repo's HIP path agrees with the oracle within 6e-4 on every column
(tests/test_denoise_gpu.py::test_denoise_step_diagnostics_match_the_oracle, tests/test_pinning_kit_gpu.py).

    python tools/make_pinning_case.py --out /tmp/pin --layers 48 [--width 256 --height 256 --frames 9] [--seed 42] [--hip]
"""
import argparse
import json
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ltx_oracle as o  # noqa: E402


def line(i, sg, sn, st):
    return f"  Step {i}: σ={sg:.4f}→{sn:.4f}, vel mean={st[0]:.4f}, std={st[1]:.4f}, latent mean={st[2]:.4f}, std={st[3]:.4f}"


def save_safetensors(path, tensors):
    """Minimal safetensors writer (8-byte little-endian header length, JSON header, raw little-endian data) that can store bf16 without
    torch or ml_dtypes: `tensors` maps a name to (array, "BF16" | "F32" | "I32"); BF16 entries are f32 arrays of bf16-representable values
    and are written as their upper 16 bits. Streams tensor by tensor."""
    size = {"BF16": 2, "F32": 4, "I32": 4}
    header, off = {}, 0
    for name, (arr, dt) in tensors.items():
        n = int(np.prod(arr.shape)) * size[dt]
        header[name] = {"dtype": dt, "shape": [int(s) for s in arr.shape], "data_offsets": [off, off + n]}
        off += n
    hj = json.dumps(header, separators=(",", ":")).encode()
    hj += b" " * (-len(hj) % 8)
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(hj)))
        f.write(hj)
        for name, (arr, dt) in tensors.items():
            if dt == "BF16":
                bits = o.f32_to_bf16_bits(arr)
                assert np.array_equal(o.bf16_bits_to_f32(bits), np.asarray(arr, np.float32)), name + ": not bf16-representable"
                f.write(bits.astype("<u2").tobytes())
            elif dt == "F32":
                f.write(np.ascontiguousarray(arr, "<f4").tobytes())
            else:
                f.write(np.ascontiguousarray(arr, "<i4").tobytes())


def build_case(layers, heads, caption, width, height, frames, text_keys, seed):
    """-> (oracle config, module-key weights, latent shape, embeddings, mask, noise, sigmas); everything derives from `seed`."""
    cfg = o.DiTConfig(num_layers=layers, num_heads=heads, caption_channels=caption)
    w = o.synth_dit_weights(cfg, seed=seed)
    F, H, W = o.latent_shape(width, height, frames)
    rng = np.random.default_rng(seed + 1)
    emb = o.bf16_round(rng.standard_normal((1, text_keys, caption)).astype(np.float32))
    mask = np.ones((1, text_keys), np.int32)
    noise = rng.standard_normal((1, 128, F, H, W)).astype(np.float32)
    sig = np.asarray(o.sigmas(True, 8, F * H * W), np.float32)
    return cfg, w, (F, H, W), emb, mask, noise, sig


def oracle_run(cfg, w, fhw, emb, mask, noise, sig):
    F, H, W = fhw
    stats = []
    final = o.denoise(w, cfg, noise * sig[0], sig, emb, mask, F, H, W, step_stats=stats)
    return np.asarray(stats, np.float64), final


def report(stats, sig, final):
    lines = [line(i, float(sig[i]), float(sig[i + 1]), stats[i]) for i in range(len(stats))]
    # the reference's "[DIAG] Final latent: mean=, std=, min=, max=" (LTXPipeline.swift:960-964; it prints Swift's shortest float form)
    lines.append(f"  [DIAG] Final latent: mean={float(final.mean(dtype=np.float64)):.4f}, std={float(final.std(dtype=np.float64)):.4f}, "
                 f"min={float(final.min()):.4f}, max={float(final.max()):.4f}")
    return lines


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--layers", type=int, default=48)
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--caption", type=int, default=3840)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--frames", type=int, default=9)
    ap.add_argument("--text-keys", type=int, default=256)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--hip", action="store_true", help="also run the case through libltxhip.so (needs an MI355X)")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    cfg, w, fhw, emb, mask, noise, sig = build_case(a.layers, a.heads, a.caption, a.width, a.height, a.frames, a.text_keys, a.seed)
    F, H, W = fhw
    files = o.dit_file_keys(w)
    # Linear weights as bf16 (what the checkpoint holds); vectors (biases, norm weights, scale-shift tables) too - all are bf16 values
    save_safetensors(os.path.join(a.out, "ltx_transformer.safetensors"), {k: (np.asarray(v, np.float32), "BF16") for k, v in files.items()})
    save_safetensors(os.path.join(a.out, "case.safetensors"),
                     {"prompt_embeddings": (emb, "F32"), "prompt_mask": (mask, "I32"), "noise": (noise, "F32"), "sigmas": (sig, "F32")})
    stats, final = oracle_run(cfg, w, fhw, emb, mask, noise, sig)
    lines = report(stats, sig, final)
    open(os.path.join(a.out, "expected_oracle.txt"), "w").write("\n".join(lines) + "\n")
    meta = {"layers": a.layers, "heads": a.heads, "caption_channels": a.caption, "width": a.width, "height": a.height, "frames": a.frames,
            "latent": [F, H, W], "text_keys": a.text_keys, "seed": a.seed, "sigmas": [float(s) for s in sig]}
    json.dump(dict(meta, step_stats=stats.tolist(), final_mean=float(final.mean(dtype=np.float64)), final_std=float(final.std(dtype=np.float64))),
              open(os.path.join(a.out, "expected_oracle.json"), "w"), indent=1)
    print("\n".join(lines))
    if a.hip:
        import importlib

        sys.path.insert(0, ROOT)
        ltx = importlib.import_module("ltx-video-swift-mlx_amd")
        ctx = ltx.Context(0)
        tcfg = ltx.default_transformer_config(num_layers=a.layers, num_attention_heads=a.heads, cross_attention_dim=a.heads * 128,
                                              caption_channels=a.caption)
        ctx.dit_load(os.path.join(a.out, "ltx_transformer.safetensors"), tcfg)
        hs = np.zeros((len(sig) - 1, 4), np.float32)
        got = ctx.denoise(noise * sig[0], sig, ltx.f32_to_bf16_bits(emb), mask, F, H, W, step_stats=hs)
        hl = report(hs.astype(np.float64), sig, got)
        open(os.path.join(a.out, "expected_hip.txt"), "w").write("\n".join(hl) + "\n")
        json.dump(dict(meta, step_stats=hs.tolist(), final_mean=float(got.mean(dtype=np.float64)), final_std=float(got.std(dtype=np.float64)),
                       max_abs_diff_to_oracle=float(np.abs(hs - stats).max())), open(os.path.join(a.out, "expected_hip.json"), "w"), indent=1)
        print("libltxhip.so:\n" + "\n".join(hl) + f"\nmax |difference| to the oracle over all columns: {float(np.abs(hs - stats).max()):.2e}")
        ctx.close()


if __name__ == "__main__":
    main()
