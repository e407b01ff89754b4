// pipeline.cpp - see pipeline.h.
#include "pipeline.h"
#include <math.h>

#include "dist.h"
#include "dit.h"
#include "elementwise.h"

// Cache keys of the projected text context (dit.cpp prepare_context matches entries on (key, B, S)). Every pass of this loop
// derives its key the same way from the caller's version, in a namespace of its own (top bit set), so that no two passes with
// different contents can meet on one key whatever small counters a caller uses: kind 0 = the [neg,pos] batch, 1 = the negative
// context alone, 2 = the positive context alone (plain forward, sharded positive branch and the STG pass share this entry).
// Cache-key kinds of the passes of one denoise step (DiTForwardArgs::ctx_kind). The caller's version travels unchanged beside them:
// any non-zero 64-bit value is a valid version (a hash of the prompt, say), and a raw ltx_dit_forward (kind 0) can never land on a
// denoise entry whatever its bits.
enum { CTX_PAIR = 1, CTX_NEG = 2, CTX_POS = 3 };

void denoise_run(ltx_ctx* ctx, const DenoiseParams& p) {
    DiTModel* m = ctx->dit;
    if (!m) LTX_THROW(LTXS_MODEL_NOT_LOADED, "Model component not loaded: transformer");
    LTX_REQUIRE(p.latent && p.sigmas && p.n_sigmas >= 2 && p.context && p.S >= 1, "denoise: bad arguments");
    const int C = m->cfg.in_channels;
    LTX_REQUIRE(m->cfg.out_channels == C, "denoise: in/out channels differ");
    const int T = p.F * p.H * p.W;
    const long n = (long)C * T;
    const bool use_cfg = p.cfg_scale > 1.0f;
    const int world = dist_world(ctx), rank = dist_rank(ctx);
    LTX_REQUIRE(p.shard == SHARD_NONE || p.shard == SHARD_CFG || p.shard == SHARD_SEQUENCE, "denoise: unknown shard mode %d", p.shard);
    LTX_REQUIRE(p.shard == SHARD_NONE || ctx->dist, "denoise: a sharded loop needs ltx_dist_init / ltx_dist_set_transport on this context");
    const bool cfg_sharded = p.shard == SHARD_CFG && use_cfg;
    if (p.shard == SHARD_CFG) {
        LTX_REQUIRE(use_cfg, "denoise: CFG sharding needs cfg_scale > 1 (got %g)", (double)p.cfg_scale);
        LTX_REQUIRE(world == 2 || (world == 3 && p.stg_scale > 0.f),
                    "denoise: the CFG pair shards over 2 ranks (3 with an STG pass); this context's group has %d", world);
    }
    const bool seq = p.shard == SHARD_SEQUENCE && world > 1;
    if (seq) LTX_REQUIRE(T % world == 0 && (T / world) % 8 == 0, "denoise: %d tokens do not split over %d ranks into multiples of 8", T, world);
    const int Tn = seq ? T / world : T;
    const int tok0 = seq ? rank * Tn : 0;
    const int B = (use_cfg && !cfg_sharded && !seq) ? 2 : 1;
    hipStream_t st = ctx->stream;
    const long cap = m->cfg.caption_channels;

    ctx->dn_tokens.ensure((size_t)2 * n * 2);
    ctx->dn_vel_tok.ensure((size_t)2 * n * 4);
    ctx->dn_vel.ensure((size_t)3 * n * 4);
    ctx->dn_vel2.ensure((size_t)n * 4);
    ctx->dn_vel3.ensure((size_t)n * 4);
    ctx->dn_prev.ensure((size_t)n * 4);
    ctx->dn_ts.ensure(8 * 4);
    if (seq) ctx->dn_vel_slice.ensure((size_t)Tn * C * 4);
    const bool i2v = p.cond_latent != nullptr;
    const int HW = p.H * p.W;
    const int G = i2v ? 2 : 1;  // timestep groups per batch element: 0 = sigma, 1 = frame-0 tokens at 0
    if (i2v) {
        LTX_REQUIRE(p.F >= 2, "denoise: image-to-video needs at least two latent frames");
        ctx->dn_rowmap.ensure((size_t)2 * T * 4);
        launch_i2v_rowmap(ctx->dn_rowmap.as<int32_t>(), 2, T, HW, G, st);
        launch_set_frame0(p.latent, p.cond_latent, nullptr, 0.f, 0.f, C, p.F, HW, st);  // LTXPipeline.swift:2092-2094
    }
    ctx->dn_stats.ensure(16 * 4);
    bf16_t* tokens = ctx->dn_tokens.as<bf16_t>();
    float* vel_tok = ctx->dn_vel_tok.as<float>();
    float* vel = ctx->dn_vel.as<float>();   // [2 or 3][C][T]: negative, positive (, STG-perturbed)
    float* v = ctx->dn_vel2.as<float>();    // guided velocity
    float* vp = ctx->dn_vel3.as<float>();   // STG perturbed velocity / this rank's branch before the exchange
    float* prev = ctx->dn_prev.as<float>();
    bool have_prev = false;

    // context slices: with CFG the batch order is [neg, pos]
    const bf16_t* ctx_pos = use_cfg ? p.context + (size_t)p.S * cap : p.context;
    const int32_t* mask_pos = (use_cfg && p.mask) ? p.mask + p.S : p.mask;
    const uint64_t ver = p.ctx_version;

    auto forward = [&](const bf16_t* tok, const bf16_t* c, const int32_t* mk, int b, int kind, float* out_tok) {
        DiTForwardArgs a;
        a.context = c;
        a.timesteps = ctx->dn_ts.as<float>();
        a.mask = mk;
        a.mask_all_ones = p.mask_all_ones;
        a.B = b; a.F = p.F; a.H = p.H; a.W = p.W; a.S = p.S;
        a.ctx_version = ver;
        a.ctx_kind = ver ? kind : 0;
        if (i2v) {
            a.n_groups = G;
            a.row_map = ctx->dn_rowmap.as<int32_t>() + tok0;
        }
        if (!seq) {
            a.latent = tok;
            a.velocity = out_tok;
            dit_forward(ctx, m, a);
            return;
        }
        // this rank's token rows only; the [T/N][C] velocity slices of all ranks land in global token order
        a.latent = tok + (size_t)tok0 * C;
        a.velocity = ctx->dn_vel_slice.as<float>();
        a.sp_rank = rank;
        a.sp_world = world;
        dit_forward(ctx, m, a);
        dist_allgather(ctx, a.velocity, out_tok, (long)Tn * C * 4);
    };
    auto set_stg = [&](bool on) {
        if (on) {
            for (int j = 0; j < p.n_stg; ++j) {
                const int i = p.stg_blocks[j];
                if (i >= 0 && i < m->L) m->blocks[i].skip_attn = true;
            }
        } else {
            for (auto& b : m->blocks) b.skip_attn = b.skip_ff = false;  // clearSTGSkipFlags
        }
    };

    const int steps = p.n_sigmas - 1;
    float* step_stats_dev = nullptr;
    if (p.step_stats) {
        ctx->dn_step_stats.ensure((size_t)steps * 16);
        step_stats_dev = ctx->dn_step_stats.as<float>();
    }
    for (int step = 0; step < steps; ++step) {
        const float sigma = p.sigmas[step], sigma_next = p.sigmas[step + 1];
        if (p.progress) p.progress(step, steps, sigma, p.user);  // before the forward (LTXPipeline.swift:805-810)
        if (i2v) {
            // re-noise the conditioned frame (LTXPipeline.swift:2225-2229), then per-token timesteps sigma*(1-mask) (:2237-2252)
            if (p.image_cond_noise_scale > 0.f && sigma > 0.f && p.cond_noise)
                launch_set_frame0(p.latent, p.cond_latent, p.cond_noise + (size_t)step * C * HW, p.image_cond_noise_scale, sigma * sigma, C,
                                  p.F, HW, st);
            float* ts = ctx->dn_ts.as<float>();  // [b][group] = {sigma, 0, sigma, 0}
            launch_fill_const_f32(ts, 4, sigma, st);
            launch_fill_const_f32(ts + 1, 1, 0.f, st);
            launch_fill_const_f32(ts + 3, 1, 0.f, st);
        } else {
            launch_fill_const_f32(ctx->dn_ts.as<float>(), 2, sigma, st);
        }
        // patchify + .asType(.bfloat16) (LTXPipeline.swift:815)
        launch_patchify_bf16(p.latent, tokens, 1, C, T, st);
        if (B == 2) HIP_CHECK(hipMemcpyAsync(tokens + n, tokens, (size_t)n * 2, hipMemcpyDeviceToDevice, st));

        bool have_stg = false;  // vel + 2n already holds the STG-perturbed velocity (three-rank group)
        if (!use_cfg) {
            forward(tokens, p.context, p.mask, 1, CTX_POS, vel_tok);
            launch_unpatchify_f32(vel_tok, v, 1, C, T, st);
        } else {
            if (cfg_sharded) {
                // this rank's pass only; one all-gather per step brings the others' velocities over xGMI
                if (rank == 0) {
                    forward(tokens, p.context, p.mask, 1, CTX_NEG, vel_tok);
                } else {
                    if (rank == 2) set_stg(true);
                    forward(tokens, ctx_pos, mask_pos, 1, CTX_POS, vel_tok);
                    if (rank == 2) set_stg(false);
                }
                launch_unpatchify_f32(vel_tok, vp, 1, C, T, st);
                dist_allgather(ctx, vp, vel, n * 4);  // vel = [neg | pos (| stg)]
                have_stg = world == 3;
            } else if (seq) {
                // two sequence-parallel B=1 forwards (LTXPipeline.swift:829-848)
                forward(tokens, p.context, p.mask, 1, CTX_NEG, vel_tok);
                launch_unpatchify_f32(vel_tok, vel, 1, C, T, st);
                forward(tokens, ctx_pos, mask_pos, 1, CTX_POS, vel_tok);
                launch_unpatchify_f32(vel_tok, vel + n, 1, C, T, st);
            } else {
                forward(tokens, p.context, p.mask, 2, CTX_PAIR, vel_tok);
                launch_unpatchify_f32(vel_tok, vel, 2, C, T, st);
            }
            const float* uncond = vel;
            const float* cond = vel + n;
            launch_cfg_combine(uncond, cond, p.cfg_scale, v, n, st);  // f32 (LTXPipeline.swift:860-865)
            if (p.guidance_rescale > 0.f) {
                float* stats = ctx->dn_stats.as<float>();
                launch_mean_var(v, n, 1, stats, st);
                launch_mean_var(cond, n, 1, stats + 2, st);
                launch_guidance_rescale(v, stats, stats + 2, p.guidance_rescale, n, 1, st);
            }
        }
        // STG: extra cond-only pass with self-attention skipped on stg_blocks (LTXPipeline.swift:897-921)
        if (p.stg_scale > 0.f) {
            const float* pert = vp;
            if (have_stg) {
                pert = vel + 2 * n;
            } else {
                set_stg(true);
                forward(tokens, ctx_pos, mask_pos, 1, CTX_POS, vel_tok);
                set_stg(false);
                launch_unpatchify_f32(vel_tok, vp, 1, C, T, st);
            }
            launch_axpby(v, pert, p.stg_scale, 1.0f, v, n, st);  // v + s*(v - vp)
        }
        // GE velocity correction (LTXPipeline.swift:924-927): v = g*(v - prev) + prev ; prev <- v
        if (p.ge_gamma > 0.f && have_prev) launch_ge(v, prev, p.ge_gamma, v, n, st);
        if (p.ge_gamma > 0.f) {
            HIP_CHECK(hipMemcpyAsync(prev, v, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
            have_prev = true;
        }
        launch_euler_step(p.latent, v, sigma, sigma_next, n, st, i2v ? HW : 0, p.F);  // I2V: frames 1+ only (:2344-2357)
        if (step_stats_dev) {  // velocity.mean / sqrt(variance), latent.mean / sqrt(variance) (LTXPipeline.swift:945-951)
            launch_mean_var(v, n, 1, step_stats_dev + 4 * step, st);
            launch_mean_var(p.latent, n, 1, step_stats_dev + 4 * step + 2, st);
        }
    }
    if (step_stats_dev) {
        HIP_CHECK(hipMemcpyAsync(p.step_stats, step_stats_dev, (size_t)steps * 16, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        for (int i = 0; i < steps; ++i) {  // variance -> standard deviation
            p.step_stats[4 * i + 1] = sqrtf(p.step_stats[4 * i + 1]);
            p.step_stats[4 * i + 3] = sqrtf(p.step_stats[4 * i + 3]);
        }
    }
}
