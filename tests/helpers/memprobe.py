import importlib, sys, os, pathlib
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle")); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import torch
ltx = importlib.import_module("ltx-video-swift-mlx_amd")
import ltx_oracle as oracle
cfg = ltx.default_transformer_config(num_layers=2, num_attention_heads=8, cross_attention_dim=1024, caption_channels=256)
def run(tag, pre):
    ctx = ltx.Context(0)
    ctx.dit_init_synthetic(cfg, seed=5)
    pre(ctx)
    torch.cuda.synchronize(); f2 = torch.cuda.mem_get_info()[0]
    ctx.dit_quantize(8)
    torch.cuda.synchronize(); f3 = torch.cuda.mem_get_info()[0]; print(tag, "after quantize", f3 - f2, flush=True)
    ctx.close()
run("plain", lambda c: None)
run("export1", lambda c: c.dit_export_param("proj_out.weight"))
def many(c):
    keys = ["transformer_blocks.1.attn1.to_q.weight", "transformer_blocks.1.attn1.to_k.weight", "transformer_blocks.0.ff.project_in.proj.weight",
            "transformer_blocks.1.ff.project_out.weight", "patchify_proj.weight", "adaln_single.linear.weight", "proj_out.weight"]
    return {k: c.dit_export_param(k) for k in keys}
run("export7", many)
import test_lora_quant_gpu as t
try:
    t.test_quantised_storage_is_real_and_matches_the_oracle_rule(ltx, oracle, pathlib.Path("/tmp"), 8)
    print("test function passed when called directly")
except AssertionError as e:
    print("test function failed when called directly:", e)
