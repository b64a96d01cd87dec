"""Host statement of the fp8 weight format (pevit_amd/fp8.py): CPU-only properties."""
import torch

from pevit_amd import fp8


def test_scales_are_powers_of_two_and_codes_never_saturate():
    g = torch.Generator().manual_seed(0)
    w = torch.randn(300, 256, generator=g) * torch.logspace(-6, 3, 300)[:, None]
    w[7] = 0.0
    codes, s = fp8.quantize_rows(w)
    m, _ = torch.frexp(s)
    assert torch.all(m == 0.5)                                  # exact powers of two
    q = codes.view(torch.float8_e4m3fn).float()
    assert torch.isfinite(q).all() and float(q.abs().max()) <= fp8.E4M3_MAX
    amax = q.abs().amax(dim=1)
    assert torch.all((amax > 0.5 * fp8.E4M3_MAX * 0.9) | (w.abs().amax(dim=1) == 0))   # the top binade is used
    deq = fp8.dequantize_rows(codes, s)
    rel = ((deq - w).abs().amax(dim=1) / w.abs().amax(dim=1).clamp_min(1e-30))
    assert float(rel.max()) < 2 ** -4                           # half a step of a 3-bit mantissa at the row maximum


def test_dequantised_weights_are_a_fixed_point_and_exact_in_bf16():
    g = torch.Generator().manual_seed(1)
    w = torch.randn(64, 384, generator=g) * 0.05
    deq = fp8.dequantize_rows(*fp8.quantize_rows(w))
    assert torch.equal(fp8.dequantize_rows(*fp8.quantize_rows(deq)), deq)
    assert torch.equal(deq.to(torch.bfloat16).float(), deq)
    # scaling a row by a power of two (the 1/8 folded into the q rows at load) commutes with the quantiser
    assert torch.equal(fp8.dequantize_rows(*fp8.quantize_rows(deq * 0.125)), deq * 0.125)


def test_kperm_is_a_permutation_with_contiguous_lane_runs():
    p = fp8.kperm(256)
    assert sorted(p.tolist()) == list(range(256))
    # the 32 codes a lane half needs of one 64-wide k-tile are contiguous: k = 64*par + 16*ks + 8*half + j
    for par in range(2):
        for half in range(2):
            ks_ = [64 * par + 16 * ks + 8 * half + j for ks in range(4) for j in range(8)]
            pos = p[ks_].tolist()
            assert pos == list(range(pos[0], pos[0] + 32)) and pos[0] % 32 == 0


def test_dequantized_state_dict_touches_only_the_block_products():
    sd = {"visual.transformer.resblocks.0.attn.in_proj_weight": torch.randn(384, 128),
          "visual.transformer.resblocks.0.attn.in_proj_bias": torch.randn(384),
          "visual.transformer.resblocks.0.mlp.c_fc.weight": torch.randn(512, 128),
          "visual.proj": torch.randn(128, 64)}
    out = fp8.dequantized_state_dict(sd)
    assert torch.equal(out["visual.proj"], sd["visual.proj"])
    assert torch.equal(out["visual.transformer.resblocks.0.attn.in_proj_bias"], sd["visual.transformer.resblocks.0.attn.in_proj_bias"])
    for k in ("visual.transformer.resblocks.0.attn.in_proj_weight", "visual.transformer.resblocks.0.mlp.c_fc.weight"):
        assert not torch.equal(out[k], sd[k]) and torch.allclose(out[k], sd[k], rtol=0.07, atol=1e-3)
