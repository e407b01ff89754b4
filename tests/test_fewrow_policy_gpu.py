"""Few-row launches (config 1: 256x256x9 -> 128 tokens; LTXTransformerBlock.swift:187-232 is the graph being run): the weight loads'
cache policy is an A/B switch of the library (option "b_nt" of ltx_ctx_set_option; an environment hook until round 4) and must not change
a bit of the forward - one forward per setting, same process, same context."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def six_layer(ltx):
    ctx = ltx.Context(0)
    ctx.dit_init_synthetic(ltx.default_transformer_config(num_layers=6), seed=1234)
    yield ctx
    ctx.close()


@pytest.mark.parametrize("F,H,W", [(2, 8, 8), (1, 5, 7)])  # 128 tokens (config 1); 35 tokens (ragged against every tile)
def test_non_temporal_weight_loads_keep_the_bits(ltx, six_layer, F, H, W):
    ctx = six_layer
    T, S = F * H * W, 1024
    lat = torch.empty((1, T, 128), dtype=torch.bfloat16, device="cuda")
    ctx.op_fill_normal_bf16(lat, seed=3)
    c = torch.empty((1, S, 3840), dtype=torch.bfloat16, device="cuda")
    ctx.op_fill_normal_bf16(c, seed=4)
    ts = torch.full((1,), 0.7, dtype=torch.float32, device="cuda")

    def run(**opts):
        vel = torch.empty((1, T, 128), dtype=torch.float32, device="cuda")
        with ctx.options(**opts):
            ctx.dit_forward_dev(lat, c, ts, None, F, H, W, vel, ctx_version=5, mask_all_ones=True)
            torch.cuda.synchronize()
        assert bool(torch.isfinite(vel).all())
        return vel

    base = run()
    assert torch.equal(base, run(b_nt=0))
    assert torch.equal(base, run(b_nt=1))
